"""Runs a fast, self-contained subset of the REFERENCE's own test files against pfrl_amd through
tools/run_reference_tests.py (``import pfrl`` -> ``pfrl_amd``).  Only where the reference is
mounted (the build container); skipped on the GPU box.  The full selection and its accounting are
in COVERAGE.md (b)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = os.environ.get("PFRL_REFERENCE", "/root/reference")

# no gym environments, no worker processes, no git: nothing environmental can fail in these
# (deselected: actor-learner mode, outside SURVEY 8; the PPO dataset equivalence test, which the
# reference itself fails here at rtol 1e-7)
FILES = [
    "tests/collections_tests/test_random_access_queue.py",
    "tests/collections_tests/test_persistent_collections.py",
    "tests/collections_tests/test_prioritized.py",
    "tests/replay_buffers_test/test_replay_buffer.py",
    "tests/replay_buffers_test/test_persistent_replay_buffer.py",
    "tests/utils_tests/test_random.py",
    "tests/utils_tests/test_batch_states.py",
    "tests/utils_tests/test_copy_param.py",
    "tests/utils_tests/test_recurrent.py",
    "tests/utils_tests/test_mode_of_distribution.py",
    "tests/utils_tests/test_conjugate_gradient.py",
    "tests/utils_tests/test_is_return_code_zero.py",
    "tests/utils_tests/test_stoppable_thread.py",
    "tests/utils_tests/test_clip_l2_grad_norm.py",
    "tests/utils_tests/test_contexts.py",
    "tests/nn_tests/test_recurrent_sequential.py",
    "tests/nn_tests/test_recurrent_branched.py",
    "tests/nn_tests/test_empirical_normalization.py",
    "tests/nn_tests/test_noisy_linear.py",
    "tests/explorers_tests/test_epsilon_greedy.py",
    "tests/explorers_tests/test_boltzmann.py",
    "tests/explorers_tests/test_additive_ou.py",
    "tests/experiments_tests/test_train_agent.py",
    "tests/experiments_tests/test_train_agent_batch.py",
    "tests/experiments_tests/test_hooks.py",
    "tests/test_agent.py",
    "tests/test_action_value.py",
    "tests/agents_tests/test_dqn.py",
    "tests/agents_tests/test_double_dqn.py",
    "tests/agents_tests/test_categorical_dqn.py",
    "tests/agents_tests/test_double_categorical_dqn.py",
    "tests/agents_tests/test_iqn.py",
    "tests/agents_tests/test_al.py",
    "tests/agents_tests/test_pal.py",
    "tests/agents_tests/test_double_pal.py",
    "tests/agents_tests/test_dpp.py",
    "tests/agents_tests/test_ppo.py",
    "tests/agents_tests/test_soft_actor_critic.py",
    "tests/agents_tests/test_td3.py",
    "tests/agents_tests/test_ddpg.py",
    "tests/agents_tests/test_a2c.py",
]


def _run(args, log):
    cmd = [sys.executable, os.path.join(ROOT, "tools", "run_reference_tests.py"),
           "--timeout", "300"] + args
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    with open(log, "w") as out:       # a file, not a pipe: see the runner's docstring
        proc = subprocess.run(cmd, stdout=out, stderr=subprocess.STDOUT, env=env,
                              cwd=os.path.dirname(str(log)), timeout=1500, start_new_session=True)
    text = open(log).read()
    summary = [line for line in text.splitlines() if re.search(r"\d+ (passed|failed)", line)]
    assert summary, text[-3000:]
    return proc.returncode, summary[-1], text


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "tests")),
                    reason="the reference is not mounted here")
def test_reference_tests_pass_against_pfrl_amd(tmp_path):
    code, summary, text = _run(
        ["-m", "not slow and not gpu",
         "-k", "not actor_learner and not non_recurrent_equivalence"]
        + FILES, tmp_path / "reference_tests.log")
    passed = int(re.search(r"(\d+) passed", summary).group(1))
    failed = re.findall(r"^FAILED (\S+)", text, flags=re.M)
    # A few of the reference's tests are statistical and unseeded (e.g. the NoisyNet
    # "randomness" check fails about once in ten runs, against the reference itself too):
    # whatever failed has to pass when it is run again on its own.
    assert len(failed) <= 2 and " error" not in summary, text[-4000:]
    if failed:
        code, again, text2 = _run(failed, tmp_path / "rerun.log")
        assert code == 0, text2[-4000:]
        passed += int(re.search(r"(\d+) passed", again).group(1))
    assert passed >= 518, summary
