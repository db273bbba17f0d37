"""ctypes binding of the C ABI declared in include/pfrl_amd.h.

The shared library is built in-tree by :func:`build` (hipcc, gfx950) as
``pfrl_amd/lib/libpfrl_amd.so``.  There is no CPU fallback: any attempt to use
a device-resident buffer without the library raises ``RuntimeError``.
"""
import ctypes
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libpfrl_amd.so")
# measurement hook (tools/build_variant.sh): another build of the same sources, e.g. other compiler
# flags, loaded instead of the in-tree library.  Never set by the package, the tests or bench.py.
_LIB_OVERRIDE = os.environ.get("PFRL_AMD_LIB")
SOURCES = ["frames.hip", "replay.hip", "sumtree.hip", "rollout.hip", "optim.hip", "tdloss.hip", "bias_act.hip", "noisy.hip", "c51.hip", "dueling.hip", "qnet.hip", "actor.hip", "hostplan.hip", "philox.hip"]
HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    # the parity contract is one correctly rounded IEEE op per source op
    "-ffp-contract=off", "-fno-fast-math",
    # leading scalar / pointer kernel arguments (up to 16 dwords) arrive in SGPRs at wave launch
    # instead of through a scalar load from the kernarg segment: ~0.25 us less at the head of every
    # launch of the B = 32 update chain (10 launches: 91.8 -> 89.3 us)
    "-mllvm", "-amdgpu-kernarg-preload-count=16",
    # MFMA accumulators in VGPRs (gfx950 has one unified register file): the default heuristics put
    # the accumulators of the tile programs in AGPRs and then copy all of them to VGPRs and back
    # around every pipeline stage (16 v_accvgpr_read + 16 v_accvgpr_write per stage of the 64 x 64
    # forward program); with this form the copies are gone and no kernel loses occupancy
    # (tools/build_variant.sh A/B on one box: update 85.98 -> 85.06 us, large forwards 1-3 %)
    "-mllvm", "-amdgpu-mfma-vgpr-form",
]

MAX_LEVELS = 40
MAX_NSTEP = 16
MAX_STACK = 8


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; cannot build pfrl_amd HIP kernels")


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link libpfrl_amd.so (in-tree)."""
    os.makedirs(LIB_DIR, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "powf_glibc.h"), os.path.join(CSRC, "rms_update.h"),
                   os.path.join(CSRC, "nhwc.h"),
                   os.path.join(_HERE, "..", "include", "pfrl_amd.h"),
                   os.path.abspath(__file__)]      # (the compiler flags live in this file)
    if not force and os.path.exists(LIB_PATH):
        if all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps):
            return LIB_PATH
    hipcc = _hipcc()
    objs = []
    for s in srcs:
        o = os.path.join(LIB_DIR, os.path.basename(s).replace(".hip", ".o"))
        cmd = [hipcc] + HIPCC_FLAGS + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(o)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


class TableDesc(ctypes.Structure):
    """pfrl_table_t"""

    _fields_ = [
        ("t_state_ref", ctypes.c_void_p),
        ("t_next_ref", ctypes.c_void_p),
        ("t_action", ctypes.c_void_p),
        ("t_reward", ctypes.c_void_p),
        ("t_terminal", ctypes.c_void_p),
        ("e_tids", ctypes.c_void_p),
        ("e_len", ctypes.c_void_p),
        ("k", ctypes.c_int32),
        ("n", ctypes.c_int32),
        ("act_dim", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
    ]


class HostStoreDesc(ctypes.Structure):
    """pfrl_host_store_t"""

    _fields_ = [
        ("h_state_ref", ctypes.c_void_p),
        ("h_next_ref", ctypes.c_void_p),
        ("h_reward", ctypes.c_void_p),
        ("h_terminal", ctypes.c_void_p),
        ("h_min_fseq", ctypes.c_void_p),
        ("h_e_tids", ctypes.c_void_p),
        ("h_e_len", ctypes.c_void_p),
        ("h_e_min_fseq", ctypes.c_void_p),
        ("R", ctypes.c_int64),
        ("E", ctypes.c_int64),
        ("maxlen", ctypes.c_int64),
        ("bound", ctypes.c_int64),
        ("k", ctypes.c_int32),
        ("n", ctypes.c_int32),
    ]


class OptTask(ctypes.Structure):
    """pfrl_opt_task_t"""

    _fields_ = [
        ("p", ctypes.c_void_p), ("sq", ctypes.c_void_p), ("ga", ctypes.c_void_p),
        ("src", ctypes.c_void_p), ("out", ctypes.c_void_p), ("mask", ctypes.c_void_p),
        ("x", ctypes.c_void_p),
        ("numel", ctypes.c_int64), ("slab_stride", ctypes.c_int64),
        ("n_slabs", ctypes.c_int32), ("mode", ctypes.c_int32),
        ("M", ctypes.c_int32), ("F", ctypes.c_int32), ("K", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
    ]


class TreeDesc(ctypes.Structure):
    """pfrl_tree_t"""

    _fields_ = [
        ("sum_val", ctypes.c_void_p),
        ("sum_tag", ctypes.c_void_p),
        ("min_val", ctypes.c_void_p),
        ("min_tag", ctypes.c_void_p),
        ("maxp_val", ctypes.c_void_p),
        ("maxp_tag", ctypes.c_void_p),
        ("level_off", ctypes.c_int64 * MAX_LEVELS),
        ("origin", ctypes.c_int64 * MAX_LEVELS),
        ("base", ctypes.c_int64),
        ("head", ctypes.c_int64),
        ("length", ctypes.c_int64),
        ("log2_size", ctypes.c_int32),
        ("log2_smax", ctypes.c_int32),
    ]


EXPORTS = {
    # name: (restype, argtypes)
    "pfrl_amd_version": (ctypes.c_int, []),
    "pfrl_amd_last_error": (ctypes.c_char_p, []),
    "pfrl_frames_scatter": (ctypes.c_int, "pqppqp"),
    "pfrl_frames_synth_u8": (ctypes.c_int, "pqpqQqqp"),
    "pfrl_frames_synth_u8_ring": (ctypes.c_int, "pqqqqQqqp"),
    "pfrl_synth_reward_done": (ctypes.c_int, "Qqqqdpp"),
    "pfrl_select_actions": (ctypes.c_int, "pippqp"),
    "pfrl_plan_sample_n_k": (ctypes.c_int, "pqip"),
    "pfrl_pyrandom_permutation": (ctypes.c_int, "pqp"),
    "pfrl_plan_eps_greedy": (ctypes.c_int, "pqdqp"),
    "pfrl_plan_dqn_range": (ctypes.c_int64, "HpqpppppPqqqiiqppqp"),
    "pfrl_batch_states_u8": (ctypes.c_int, "pqpqfpp"),
    "pfrl_batch_states_u8_nhwc4": (ctypes.c_int, "pqpqfpp"),
    "pfrl_batch_states_u8_raw_nhwc4": (ctypes.c_int, "pqpqpp"),
    "pfrl_conv2d_u8nhwc4_fwd": (ctypes.c_int, "pfpppiiiiiiiiip"),
    "pfrl_conv2d_u8nhwc4_bwd_weight": (ctypes.c_int, "pppfppqqiiiiiiiip"),
    "pfrl_batch_states_f32": (ctypes.c_int, "pqpqpp"),
    "pfrl_table_append": (ctypes.c_int, "Tqppppppp"),
    "pfrl_entries_append": (ctypes.c_int, "Tqpppp"),
    "pfrl_batch_experiences": (ctypes.c_int, "Tpqifpqpppppppp"),
    "pfrl_batch_experiences_nhwc4": (ctypes.c_int, "Tpqfpqpppppppp"),
    "pfrl_batch_episodes": (ctypes.c_int, "Tpqifpppiiqqfppppppp"),
    "pfrl_tree_write": (ctypes.c_int, "Rqppppp"),
    "pfrl_tree_write_sum": (ctypes.c_int, "Rqpppppp"),
    "pfrl_tree_update_errors_write_sample": (ctypes.c_int, "Rqppifdifdddiiqppppqpppppppppidqpp"),
    "pfrl_tree_sample": (ctypes.c_int, "Rqpppppppppidqpp"),
    "pfrl_tree_update_errors_f32": (ctypes.c_int, "Rqppifdifdddiip"),
    "pfrl_tree_update_errors_write_f32": (ctypes.c_int, "Rqppifdifdddiiqppppp"),
    "pfrl_h2d_async": (ctypes.c_int, "ppqp"),
    "pfrl_powf_host": (ctypes.c_int, "ipfpq"),
    "pfrl_powf_host_variant": (ctypes.c_int, "fq"),
    "pfrl_powf_device": (ctypes.c_int, "ipfpqp"),
    "pfrl_tree_set_priorities": (ctypes.c_int, "Rqpppip"),
    "pfrl_gae_scan": (ctypes.c_int, "qqpppppddippp"),
    "pfrl_a2c_returns": (ctypes.c_int, "qqppppddip"),
    "pfrl_adv_stats": (ctypes.c_int, "pqppp"),
    "pfrl_ppo_minibatch": (ctypes.c_int, "qpppipppppippppppp"),
    "pfrl_ppo_act_head": (ctypes.c_int, "pppppppppppiiipp"),
    "pfrl_ppo_loss": (ctypes.c_int, "pppppppiiffffppppp"),
    "pfrl_ppo_head_loss": (ctypes.c_int, "ppppppppppiiiffffppippp"),
    "pfrl_rmsprop_step": (ctypes.c_int, "ipppppffffip"),
    "pfrl_rmsprop_fused_step": (ctypes.c_int, "ipffffip"),
    "pfrl_dqn_td_loss": (ctypes.c_int, "ppppppppqiiippppp"),
    "pfrl_dqn_head_td_loss": (ctypes.c_int, "ppppppppppiiiiipppppiqpppfp"),
    "pfrl_bias_relu_fwd": (ctypes.c_int, "pppqiqp"),
    "pfrl_bias_relu_bwd": (ctypes.c_int, "ppppppqiiqp"),
    "pfrl_c51_loss": (ctypes.c_int, "pppppppppiiiippppp"),
    "pfrl_dueling_softmax_fwd": (ctypes.c_int, "pppqiip"),
    "pfrl_dueling_softmax_bwd": (ctypes.c_int, "ppppqiip"),
    "pfrl_noisy_weights_fwd": (ctypes.c_int, "pppppppqqp"),
    "pfrl_noisy_weights_bwd": (ctypes.c_int, "pppppqqp"),
    "pfrl_conv2d_nhwc_fwd": (ctypes.c_int, "ppppiiiiiiiiiiip"),
    "pfrl_conv2d_nhwc_bwd_data": (ctypes.c_int, "pppppiiiiiiiiiip"),
    "pfrl_conv2d_nhwc_bwd_weight": (ctypes.c_int, "pppppqqiiiiiiiiip"),
    "pfrl_conv2d_nhwc_bwd_weight_ride": (ctypes.c_int, "pppppqqiiiiiiiiiipppppffffip"),
    "pfrl_conv2d_nhwc_bwd": (ctypes.c_int, "ppppppppqqiiiiiiiiiiip"),
    "pfrl_ride_set": (ctypes.c_int, "ippppppp" + "ffffi"),
    "pfrl_splitk_reduce": (ctypes.c_int, "ippppppppp"),
    "pfrl_splitk_reduce_noisy": (ctypes.c_int, "ippppppppppp"),
    "pfrl_splitk_group": (ctypes.c_int, "pqiiipqp"),
    "pfrl_clip_grad_norm": (ctypes.c_int, "ippfppp"),
    "pfrl_linear_noisy_fwd": (ctypes.c_int, "pppppppiiiiip"),
    "pfrl_linear_noisy_fwd_pair": (ctypes.c_int, "pippppppiipip"),
    "pfrl_linear_fwd": (ctypes.c_int, "ppppiiiiip"),
    "pfrl_linear_bwd_weight": (ctypes.c_int, "pppppqqiiiip"),
    "pfrl_linear_small_fwd": (ctypes.c_int, "ppppiiip"),
    "pfrl_linear_small_bwd": (ctypes.c_int, "ppppppiiip"),
    "pfrl_dqn_act_head": (ctypes.c_int, "pppppppiiip"),
    "pfrl_qnet_plan_images": (ctypes.c_int, "i"),
    "pfrl_squashed_gaussian_fwd": (ctypes.c_int, "pqpqppppiip"),
    "pfrl_squashed_gaussian_bwd": (ctypes.c_int, "pppppqppiip"),
    "pfrl_squashed_head_fwd": (ctypes.c_int, "pqffippppiip"),
    "pfrl_squashed_head_bwd": (ctypes.c_int, "pppppqffipiip"),
    "pfrl_soft_update": (ctypes.c_int, "ipppdp"),
    "pfrl_adam_step": (ctypes.c_int, "ippppppdddddpp"),
    "pfrl_adam_step_ex": (ctypes.c_int, "ippppppppdpdddddpp"),
    "pfrl_sac_temperature_loss": (ctypes.c_int, "ppfpip"),
    "pfrl_sac_temperature_step": (ctypes.c_int, "ppfppppdddddip"),
    "pfrl_sac_target_q": (ctypes.c_int, "pppppppfpip"),
    "pfrl_half_mse_fwd": (ctypes.c_int, "pppip"),
    "pfrl_half_mse_bwd": (ctypes.c_int, "ppppip"),
    "pfrl_half_mse_twin_fwd": (ctypes.c_int, "ppppip"),
    "pfrl_half_mse_twin_bwd": (ctypes.c_int, "ppppip"),
    "pfrl_linear_fwd_twin": (ctypes.c_int, "ppipppiiiip"),
    "pfrl_linear_bwd_twin": (ctypes.c_int, "pppppipppqqiiiip"),
    "pfrl_linear_small_fwd_twin": (ctypes.c_int, "ppppiiip"),
    "pfrl_linear_small_bwd_twin": (ctypes.c_int, "ppppppiiip"),
    "pfrl_twin_input_grad": (ctypes.c_int, "pppiiipiip"),
    "pfrl_sac_policy_loss_fwd": (ctypes.c_int, "ppppfppppip"),
    "pfrl_sac_policy_loss_bwd": (ctypes.c_int, "ppppfpppip"),
    "pfrl_philox_normal": (ctypes.c_int, "QQipppppip"),
    "pfrl_profile_enable": (ctypes.c_int, "i"),
    "pfrl_profile_collect": (ctypes.c_int64, "pppq"),
}

_CODES = {
    "p": ctypes.c_void_p, "q": ctypes.c_int64, "Q": ctypes.c_uint64, "i": ctypes.c_int,
    "f": ctypes.c_float, "d": ctypes.c_double,
    "T": ctypes.POINTER(TableDesc), "R": ctypes.POINTER(TreeDesc),
    "H": ctypes.POINTER(HostStoreDesc), "P": ctypes.c_void_p,
}

_lib = None


def lib():
    """Load libpfrl_amd.so; raise loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "pfrl_amd: %s is missing.  The device replay path has no CPU fallback; "
            "run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc)." % LIB_PATH
        )
    L = ctypes.CDLL(_LIB_OVERRIDE or LIB_PATH)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(L, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = [_CODES[c] for c in args] if isinstance(args, str) else args
    _lib = L
    return L


def available():
    return os.path.exists(LIB_PATH)


def check(rc, what=""):
    if rc != 0:
        msg = lib().pfrl_amd_last_error().decode()
        raise RuntimeError("pfrl_amd %s failed (code %d): %s" % (what, rc, msg))


class timed_calls:
    """``with timed_calls() as rec:`` -- every entry point of the library that launches on a stream
    (last argument of its signature: the stream) is bracketed by a pair of timing events on the
    current torch stream while the block runs; ``rec.results()`` -> [(entry point, microseconds)]
    in call order.  A measurement hook (bench.py's per-launch table of one update run eagerly):
    the launches are the ones a captured graph replays, their durations include ~1 us of event
    bracketing each.  Not re-entrant; restores the plain entry points on exit."""

    def __init__(self):
        self.calls = []
        self._saved = {}

    def __enter__(self):
        import torch

        L = lib()
        for name, (_, args) in EXPORTS.items():
            if not (isinstance(args, str) and args.endswith("p")) or name.startswith("pfrl_plan_"):
                continue
            orig = getattr(L, name)
            self._saved[name] = orig

            def wrapped(*a, _orig=orig, _name=name):
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = _orig(*a)
                e1.record()
                self.calls.append((_name, e0, e1))
                return rc

            setattr(L, name, wrapped)
        return self

    def __exit__(self, *exc):
        L = lib()
        for name, orig in self._saved.items():
            setattr(L, name, orig)
        self._saved = {}
        return False

    def results(self):
        import torch

        torch.cuda.synchronize()
        return [(n, e0.elapsed_time(e1) * 1e3) for n, e0, e1 in self.calls]
