timeout 900 python -m pytest -x -q -m gpu tests/test_hip_kernels.py -k "split_sample or gae or uniform_ratio_and_no_wait or prioritized" 2>&1 | tail -40
