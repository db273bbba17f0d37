"""CPU tests of the host-side integer bookkeeping (no kernels involved) and of
the C-ABI library's exported symbols."""
import glob
import os
import re

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "pbuf_trace_*.npz"))),
                         ids=os.path.basename)
def test_tree_frame_matches_reference_bounds(path):
    """TreeFrame reproduces TreeQueue.bounds / length after every operation
    (pfrl/collections/prioritized.py:207-242)."""
    from pfrl_amd.collections.tree_frame import TreeFrame, smax_log2_for_capacity

    g = np.load(path)
    cap = int(g["meta"][1])
    f = TreeFrame()
    max_log2 = 0
    for k, kind in enumerate(g["op_kind"]):
        if kind in (0, 1):
            if cap >= 0 and f.length == cap:
                f.popleft()
            f.append()
        elif kind == 4:
            f.popleft()
        assert f.length == g["length"][k]
        if f.length:
            assert f.bounds == (g["ixl"][k], g["ixr"][k]), k
            assert f.base <= f.head < f.base + f.size
            assert f.head + f.length <= f.base + f.size
            max_log2 = max(max_log2, f.log2_size)
            # every live level is aligned to the frame start
            for l in range(1, f.log2_size + 1):
                assert (f.base - f.origin[l]) % (1 << l) == 0
    if cap >= 0:
        assert max_log2 <= smax_log2_for_capacity(cap)


def test_c_abi_exports_every_declared_symbol():
    """The shared library loads without a GPU and exports every function that
    include/pfrl_amd.h declares (no compute calls here)."""
    from pfrl_amd import _native

    if not _native.available():
        _native.build()
    lib = _native.lib()
    header = open(os.path.join(ROOT, "include", "pfrl_amd.h")).read()
    declared = set(re.findall(r"\b(pfrl_[a-z0-9_]+)\s*\(", header))
    declared = {d for d in declared if not d.endswith("_t")}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), "library does not export %s" % name
    assert set(_native.EXPORTS) == declared
    assert lib.pfrl_amd_version() >= 100


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under pfrl_amd/ may import, link
    or load it (the device path must fail loudly instead of falling back)."""
    import glob

    bad = []
    for path in glob.glob(os.path.join(ROOT, "pfrl_amd", "**", "*.py"), recursive=True):
        src = open(path).read()
        if re.search(r"^\s*(import|from)\s+oracle\b", src, re.M) or "libpfrl_oracle" in src:
            bad.append(path)
    for path in glob.glob(os.path.join(ROOT, "pfrl_amd", "csrc", "*")):
        if "oracle" in open(path).read():
            bad.append(path)
    assert not bad, bad


def test_device_paths_fail_loudly_without_the_hip_library(monkeypatch):
    from pfrl_amd import _native

    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", os.path.join(ROOT, "pfrl_amd", "lib", "missing.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _native.lib()


def test_multiprocess_vector_env_matches_serial():
    from pfrl_amd.envs import MultiprocessVectorEnv, SerialVectorEnv

    envs_a = MultiprocessVectorEnv([_CountingEnv, _CountingEnv])
    envs_b = SerialVectorEnv([_CountingEnv(), _CountingEnv()])
    try:
        assert envs_a.num_envs == 2
        assert list(envs_a.reset()) == list(envs_b.reset())
        for actions in ([1, 2], [3, 4], [0, 0]):
            ra, rb = envs_a.step(actions), envs_b.step(actions)
            assert [list(x) for x in ra[:3]] == [list(x) for x in rb[:3]]
        mask = [True, False]
        assert list(envs_a.reset(mask)) == list(envs_b.reset(mask))
    finally:
        envs_a.close()
    with pytest.raises(AssertionError):
        envs_a.step([0, 0])


class _CountingEnv:
    def __init__(self):
        self.t = 0

    def reset(self):
        self.t = 0
        return 0

    def step(self, a):
        self.t += a + 1
        return self.t, float(a), self.t > 6, {}

    def seed(self, s):
        return [s]

    def close(self):
        pass


def test_bench_result_line_contract(monkeypatch):
    """bench.py's JSON line (driver contract + roofline object) assembled from
    synthetic timings: every required key, the dominant launch shape, the aggregate."""
    import json
    import sys

    sys.path.insert(0, ROOT)
    import bench

    monkeypatch.setattr(sys, "argv", ["bench.py", "--steps", "20", "--warmup", "5"])
    args = bench.parse_args()
    assert (args.gpus, args.algo, args.num_envs) == (1, "dqn", 256)
    # 20 steps: one 256-entry and one 1792-entry gather each, plus act gathers (kind 1)
    us = [20.0, 100.0] * 20 + [9.5] * 20
    units = [256, 1792] * 20 + [1024] * 20
    kinds = [0, 0] * 20 + [1] * 20
    roof = bench.compute_roofline("dqn", us, units, kinds)
    assert roof["kernel"] == "k_batch_experiences" and roof["bound"] == "hbm"
    assert roof["entries_per_launch"] == 1792 and roof["launches_timed"] == 20
    per_entry = 2 * 4 * 5 * 84 * 84
    assert roof["bytes_per_launch"] == 1792 * per_entry
    np.testing.assert_allclose(roof["achieved"], 1792 * per_entry / 100e-6 / 1e9, rtol=1e-3)
    np.testing.assert_allclose(roof["frac"], roof["achieved"] / roof["peak"], atol=1e-4)
    np.testing.assert_allclose(roof["all_launches"]["achieved"],
                               2048 * per_entry / 120e-6 / 1e9, rtol=1e-3)
    assert roof["all_launches"]["launches"] == 40
    assert roof["traffic"] is None or roof["traffic"] > roof["bytes_per_launch"] * 0.9
    ppo = bench.compute_roofline("ppo", us, units, kinds)
    assert ppo["kernel"] == "k_batch_states_u8" and ppo["frames_per_launch"] == 1024
    assert bench.compute_roofline("dqn", [], [], []) is None
    out = bench.assemble_result(args, 2, 256, 0.4, 1280, 8.3, "workload text", roof)
    line = json.loads(json.dumps(out))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline"):
        assert key in line, key
    assert line["value"] == 2 * 256 * 20 / 0.4 and line["n_gpus"] == 2
    assert line["ms_per_step"] == 20.0 and line["scaling"] == "strong"   # 256 envs sharded N/G
    # the step as a whole against the HBM roofline (not the gather kernel's fraction)
    step_bytes = bench.algorithmic_bytes_per_step("dqn", 256, 32, 4)
    assert step_bytes == 256 * (7056 + 141120 + 8 * 282240)      # 2,406,096 B / env-step
    np.testing.assert_allclose(line["roofline"]["step_frac"], step_bytes / 20e-3 / 8e12, rtol=1e-3)
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    dflt = bench.parse_args()
    assert dflt.steps >= 200 and dflt.scaling == "strong"
    assert line["vs_baseline"] is None and line["higher_is_better"] is True
    assert line["config"]["workload"] == "workload text" and "model" not in line["config"]
    assert bench.PROFILE_BATCH_EXPERIENCES == 0 and bench.PROFILE_BATCH_STATES_U8 == 1


def test_integration_doc_names_every_entry_point():
    """INTEGRATION.md maps every C entry point of include/pfrl_amd.h to the reference
    code it replaces."""
    hdr = open(os.path.join(ROOT, "include", "pfrl_amd.h")).read()
    names = set(re.findall(r"^\w[\w\s\*]*?\b(pfrl_\w+)\(", hdr, flags=re.M))
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert len(names) >= 30
    assert not [n for n in sorted(names) if n not in doc]


def test_python_boundary_signatures_match_the_reference():
    """a user switches ``import pfrl`` to ``import pfrl_amd as pfrl``.  For the 325
    callables of tests/_api_surface.py (constructors, functions and the methods drivers call), every
    parameter the reference declares exists here with the same name, kind, order and literal
    default (tests/golden/api_signatures.json, recorded from the reference); pfrl_amd may add
    parameters, but only optional ones."""
    import json
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _api_surface import describe_api

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                           "api_signatures.json")) as f:
        ref = json.load(f)
    mine = describe_api("pfrl_amd")
    assert len(ref) > 250
    problems = []
    for key, want in sorted(ref.items()):
        if key not in mine:
            problems.append("%s: missing" % key)
            continue
        got = mine[key]
        if want is None or got is None:
            if (want is None) != (got is None):
                problems.append("%s: signature introspection differs" % key)
            continue
        by_name = {p[0]: p for p in got}
        order = [p[0] for p in got]
        last = -1
        declared = set()
        for name, kind, default in want:
            declared.add(name)
            if kind in ("VAR_KEYWORD", "VAR_POSITIONAL"):
                continue
            if name not in by_name:
                problems.append("%s: no parameter %r" % (key, name))
                continue
            if by_name[name][1] != kind or by_name[name][2] != default:
                problems.append("%s: %r is %s=%s, reference %s=%s" % (
                    key, name, by_name[name][1], by_name[name][2], kind, default))
            if order.index(name) < last:
                problems.append("%s: %r out of order" % (key, name))
            last = order.index(name)
        for name, kind, default in got:
            if name not in declared and default == "<required>" and kind not in (
                    "VAR_KEYWORD", "VAR_POSITIONAL"):
                problems.append("%s: extra required parameter %r" % (key, name))
    assert not problems, "\n".join(problems)


def test_sample_n_k_equals_the_reference_walk_including_the_stream_position():
    """pfrl_amd.utils.random.sample_n_k visits only the duplicate positions; the reference
    (pfrl/utils/random.py:4-28, restated literally here) walks all k with a set.  Same indices
    and same position of the global NumPy stream afterwards, also where duplicates are
    frequent (k^2 / 2n ~ 0.3 for the SAC minibatch of 256 out of 10^5), where a spare is itself a
    duplicate, where it equals the value of a later position, and where the spares run out."""
    import numpy as np

    from pfrl_amd.utils.random import sample_n_k

    def reference(n, k):
        if 3 * k >= n:
            return np.random.choice(n, k, replace=False)
        result = np.random.choice(n, 2 * k)
        selected = set()
        j = k
        for i in range(k):
            x = result[i]
            while x in selected:
                x = result[i] = result[j]
                j += 1
                if j == 2 * k:
                    result[k:] = np.random.choice(n, k)
                    j = k
            selected.add(x)
        return result[:k]

    for n, k in ((100000, 256), (1000, 256), (800, 256), (10 ** 6, 32), (300, 90), (50, 10), (7, 2),
                 (31, 10), (4, 1)):
        for seed in range(60):
            np.random.seed(seed)
            want = reference(n, k)
            pos_ref = np.random.randint(1 << 30)
            np.random.seed(seed)
            got = sample_n_k(n, k)
            pos = np.random.randint(1 << 30)
            assert np.array_equal(got, want) and pos == pos_ref, (n, k, seed)


def test_ppo_minibatch_positions_drawn_up_front_consume_the_same_random_stream():
    """PPO's minibatch order comes from Python's ``random`` (reference ppo.py:247-257).  The device
    path keeps minibatches as arrays and, from 1 024 elements up, draws the permutations natively
    on the module's own MT19937 state (pfrl_pyrandom_permutation, CPython's pool algorithm): same
    positions, same stream position afterwards -- also right after a generator block boundary and
    with a Gaussian draw pending (``gauss_next`` is carried through setstate)."""
    import random

    from pfrl_amd.agents.ppo import _all_minibatch_positions, _yield_minibatch_positions

    for n, mb, ep in [(100, 32, 3), (64, 64, 2), (65, 16, 4), (10, 3, 5), (2048, 512, 4),
                      (65536, 16384, 2), (5000, 1250, 3), (1025, 41, 2)]:
        random.seed(5)
        a = [list(x) for x in _yield_minibatch_positions(n, mb, ep)]
        sa = random.getstate()
        random.seed(5)
        b = [[int(v) for v in x] for x in _all_minibatch_positions(n, mb, ep)]
        assert a == b and sa == random.getstate(), (n, mb, ep)
    from pfrl_amd.agents.ppo import _random_permutation

    for warm in (0, 1, 623, 624, 625, 1000):
        random.seed(11)
        random.gauss(0.0, 1.0)                      # leaves gauss_next set
        for _ in range(warm):
            random.getrandbits(32)
        a = random.sample(range(4099), k=4099)
        sa = random.getstate()
        random.seed(11)
        random.gauss(0.0, 1.0)
        for _ in range(warm):
            random.getrandbits(32)
        b = _random_permutation(4099)
        assert a == [int(v) for v in b] and sa == random.getstate(), warm


def test_appends_keep_frame_never_misses_a_frame_change():
    """``appends_keep_frame`` (what lets DQN._batch_observe_train_per look one update point ahead)
    against the frame bookkeeping itself: whenever it says the next m appends -- with the popleft a
    full buffer does first -- keep the frame, performing them must not move ``epoch``."""
    import random

    from pfrl_amd.collections.tree_frame import TreeFrame, appends_keep_frame

    rnd = random.Random(7)
    said_yes = said_no = 0
    for capacity in (5, 8, 33, 64, 100, 257):
        f = TreeFrame()
        for step in range(6 * capacity):
            m = rnd.randint(1, 4)
            pops = f.length + m > capacity
            ok = appends_keep_frame(f, m, pops)
            before = f.epoch
            for _ in range(m):
                if f.length == capacity:
                    f.popleft()
                f.append()
            if ok:
                assert f.epoch == before, (capacity, step, m)
                said_yes += 1
            else:
                said_no += 1
    assert said_yes > 10 * said_no > 0      # (and it is not vacuous: mostly yes, sometimes no)


