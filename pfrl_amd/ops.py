"""Thin torch-tensor front-end of the C ABI (include/pfrl_amd.h).

Tensors only supply device pointers and the current HIP stream; every
computation happens in the hand-written gfx950 kernels of pfrl_amd/csrc.
"""
import ctypes
import os

import numpy as np
import torch

from pfrl_amd import _native
from pfrl_amd._native import TableDesc, check


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    # (torch.cuda.current_stream() builds a Stream object through several Python layers, ~7 us a
    # call and a dozen calls per update; the raw handle of the current device's current stream
    # is one C call)
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device-contiguous tensor required"
    return ctypes.c_void_p(t.data_ptr())


def frame_bytes_of(frames):
    return frames[0].numel() * frames.element_size()


def frames_scatter(frames, src, slots):
    """frames[slots[i]] <- src[i]."""
    n = slots.numel()
    check(_native.lib().pfrl_frames_scatter(_ptr(frames), frame_bytes_of(frames), _ptr(src),
                                            _ptr(slots), n, _stream()), "frames_scatter")


def frames_synth_u8(frames, slots, seed, env_id0, step):
    check(_native.lib().pfrl_frames_synth_u8(_ptr(frames), frame_bytes_of(frames), _ptr(slots),
                                             slots.numel(), seed, env_id0, step, _stream()),
          "frames_synth_u8")


def frames_synth_u8_ring(frames, seq0, n, seed, env_id0, step):
    """n fresh frames at ring positions seq0 .. seq0 + n - 1 (mod n_slots): no slot upload."""
    check(_native.lib().pfrl_frames_synth_u8_ring(_ptr(frames), frame_bytes_of(frames),
                                                  frames.shape[0], int(seq0), int(n), seed,
                                                  env_id0, step, _stream()), "frames_synth_u8_ring")


def select_actions(greedy, choice, out=None):
    """out[i] = choice[i] >= 0 ? choice[i] : greedy[i] (epsilon-greedy with the host's draws)."""
    assert greedy.dtype in (torch.int32, torch.int64) and choice.dtype == torch.int32
    if out is None:
        out = torch.empty(greedy.shape, dtype=torch.int64, device=greedy.device)
    check(_native.lib().pfrl_select_actions(_ptr(greedy), int(greedy.dtype == torch.int32),
                                            _ptr(choice), _ptr(out), greedy.numel(), _stream()),
          "select_actions")
    return out


def dqn_act_head(h, weight, bias, choice=None, want_q=False, out=None):
    """The narrow Q head of the acting path with argmax and the epsilon-greedy decision in one
    launch (pfrl_dqn_act_head; reference pfrl/agents/dqn.py:490-507): ``h`` [M, K] f32, ``weight``
    [A, K], ``choice`` int32 [M] (the host's draws, < 0 = greedy) or None.  Returns (actions int64
    [M], q [M, A] or None); the action values are those of ``pfrl_linear_small_fwd`` bit for bit."""
    M, K = h.shape
    A = weight.shape[0]
    assert h.dtype == torch.float32 and h.is_contiguous() and weight.is_contiguous() and 1 <= A <= 16
    assert choice is None or (choice.dtype == torch.int32 and choice.numel() == M)
    if out is None:
        out = torch.empty(M, dtype=torch.int64, device=h.device)
    q = torch.empty((M, A), dtype=torch.float32, device=h.device) if want_q else None
    check(_native.lib().pfrl_dqn_act_head(_ptr(h), _ptr(weight), _ptr(bias) if bias is not None else None,
                                          _ptr(choice) if choice is not None else None,
                                          _ptr(q) if q is not None else None, None, _ptr(out), M, K, A,
                                          _stream()), "dqn_act_head")
    return out, q


def _spatial_hw(fshape):
    """(H, W) if the frame is one 2-D plane ((H, W) or (1, H, W)), else None."""
    dims = [d for d in fshape]
    while len(dims) > 2 and dims[0] == 1:
        dims = dims[1:]
    return tuple(dims) if len(dims) == 2 else None


def channels_last_supported(frames, k):
    """Stacks of four u8 planes can be emitted directly in torch.channels_last."""
    return (frames.dtype == torch.uint8 and k == 4 and _spatial_hw(tuple(frames.shape[1:]))
            is not None)


def empty_channels_last(M, hw, device):
    """f32 [M, 4, H, W] tensor whose memory is [M][H][W][4]."""
    return torch.empty((M, hw[0], hw[1], 4), dtype=torch.float32,
                       device=device).permute(0, 3, 1, 2)


def batch_states_nhwc4(frames, refs, divisor=255.0, out=None):
    """refs: int32 [M, 4] -> f32 [M, 4, H, W] in channels_last memory format."""
    M, k = refs.shape
    assert channels_last_supported(frames, k)
    hw = _spatial_hw(tuple(frames.shape[1:]))
    if out is None:
        out = empty_channels_last(M, hw, frames.device)
    assert out.is_contiguous(memory_format=torch.channels_last)
    check(_native.lib().pfrl_batch_states_u8_nhwc4(
        _ptr(frames), frame_bytes_of(frames), _ptr(refs), M, float(divisor),
        ctypes.c_void_p(out.data_ptr()), _stream()), "batch_states_u8_nhwc4")
    return out


class U8Pixels:
    """A minibatch of observations as u8 NHWC4 pixels -- ``data``: uint8 [M, H, W, 4], byte c of
    a pixel = stacked frame c -- together with the divisor of the feature extractor
    ``phi(x) = float32(x) / divisor`` that has NOT been applied yet.  What
    :func:`batch_states_raw_nhwc4` returns and the MFMA trunk's first convolution consumes
    (nn/mfma_trunk.py): phi is evaluated in that kernel's operand loader, so the fp32 copy of the
    minibatch (4 x the bytes, written by the gather and read twice by the layer) never exists."""

    __slots__ = ("data", "divisor")

    def __init__(self, data, divisor):
        self.data, self.divisor = data, float(divisor)

    @property
    def shape(self):            # (as the fp32 network input would have it: [M, 4, H, W])
        M, H, W, C = self.data.shape
        return torch.Size((M, C, H, W))

    @property
    def device(self):
        return self.data.device

    def float(self):
        """The fp32 channels_last tensor this stands for (any consumer without a u8 loader):
        the 256 values of phi from NumPy's float32 division (IEEE; torch's device division is
        not correctly rounded on this stack), looked up per byte."""
        lut = torch.from_numpy(np.arange(256, dtype=np.float32) / np.float32(self.divisor))
        return lut.to(self.data.device)[self.data.long()].permute(0, 3, 1, 2)


_U8_DIV_OK = {}


def u8_division_exact(divisor):
    """True when ``q = x * r; q + fma(-q, d, x) * r`` with ``r = fl(1 / d)`` (the u8 operand
    loaders of csrc/qnet.hip: ``u8_over``) equals IEEE ``float32(x) / float32(d)`` for EVERY byte
    value x -- checked here in exact rational arithmetic, once per divisor (255 and 1 pass)."""
    d32 = np.float32(divisor)
    key = float(d32)
    hit = _U8_DIV_OK.get(key)
    if hit is not None:
        return hit
    from fractions import Fraction as Fr

    def rn(fr):          # round-to-nearest-even of an exact value to float32
        f = np.float32(float(fr))       # (a double-rounded guess; the neighbours decide)
        best = None
        for c in (np.nextafter(f, np.float32(-np.inf)), f, np.nextafter(f, np.float32(np.inf))):
            err = abs(Fr(float(c)) - fr)
            tie = int(np.float32(c).view(np.uint32)) & 1
            if best is None or (err, tie) < best[:2]:
                best = (err, tie, c)
        return np.float32(best[2])

    ok = bool(np.isfinite(d32) and d32 > 0)
    if ok:
        d = Fr(float(d32))
        r = Fr(float(rn(1 / d)))
        for x in range(256):
            q = Fr(float(rn(x * r)))
            e = Fr(float(rn(x - q * d)))
            if float(rn(q + e * r)) != float(np.float32(x) / d32):
                ok = False
                break
    _U8_DIV_OK[key] = ok
    return ok


def batch_states_raw_nhwc4(frames, refs, divisor=255.0, out=None):
    """refs: int32 [M, 4] into u8 frames -> :class:`U8Pixels` ([M, H, W, 4] bytes, phi pending)."""
    M, k = refs.shape
    assert channels_last_supported(frames, k) and frames.dtype == torch.uint8
    hw = _spatial_hw(tuple(frames.shape[1:]))
    if out is None:
        out = torch.empty((M, hw[0], hw[1], 4), dtype=torch.uint8, device=frames.device)
    check(_native.lib().pfrl_batch_states_u8_raw_nhwc4(
        _ptr(frames), frame_bytes_of(frames), _ptr(refs), M, ctypes.c_void_p(out.data_ptr()),
        _stream()), "batch_states_u8_raw_nhwc4")
    return U8Pixels(out, divisor)


def batch_states(frames, refs, divisor=255.0, out=None):
    """refs: int32 [M, k] -> f32 [M, k, *frame_shape]."""
    M, k = refs.shape
    fshape = tuple(frames.shape[1:])
    if out is None:
        out = torch.empty((M, k) + fshape, dtype=torch.float32, device=frames.device)
    fb = frame_bytes_of(frames)
    if frames.dtype == torch.uint8:
        check(_native.lib().pfrl_batch_states_u8(_ptr(frames), fb, _ptr(refs), M * k,
                                                 float(divisor), _ptr(out), _stream()),
              "batch_states_u8")
    elif frames.dtype == torch.float32:
        check(_native.lib().pfrl_batch_states_f32(_ptr(frames), fb, _ptr(refs), M * k, _ptr(out),
                                                  _stream()), "batch_states_f32")
    else:
        raise TypeError("frame store dtype must be uint8 or float32, got %s" % frames.dtype)
    return out


def make_table_desc(t_state_ref, t_next_ref, t_action, t_reward, t_terminal, e_tids, e_len, k, n,
                    act_dim):
    d = TableDesc()
    d.t_state_ref = t_state_ref.data_ptr()
    d.t_next_ref = t_next_ref.data_ptr()
    d.t_action = t_action.data_ptr()
    d.t_reward = t_reward.data_ptr()
    d.t_terminal = t_terminal.data_ptr()
    d.e_tids = e_tids.data_ptr()
    d.e_len = e_len.data_ptr()
    d.k, d.n, d.act_dim, d.reserved = k, n, act_dim, 0
    return d


def table_append(desc, t_slots, state_ref, next_ref, action, reward, terminal):
    check(_native.lib().pfrl_table_append(ctypes.byref(desc), t_slots.numel(), _ptr(t_slots),
                                          _ptr(state_ref), _ptr(next_ref), _ptr(action),
                                          _ptr(reward), _ptr(terminal), _stream()), "table_append")


def entries_append(desc, e_slots, tids, lens):
    check(_native.lib().pfrl_entries_append(ctypes.byref(desc), e_slots.numel(), _ptr(e_slots),
                                            _ptr(tids), _ptr(lens), _stream()), "entries_append")


_PROFILE_ON = [False]


def profile_enable(on):
    """Attach a hipEvent pair to every fused-gather dispatch (bench.py roofline)."""
    check(_native.lib().pfrl_profile_enable(int(bool(on))), "profile_enable")
    _PROFILE_ON[0] = bool(on)


class profile_paused:
    """No event pairs while a stream capture records launches (events created inside a capture
    would belong to the capture, and a replay never signals them)."""

    def __enter__(self):
        self.was = _PROFILE_ON[0]
        if self.was:
            profile_enable(False)

    def __exit__(self, *exc):
        if self.was:
            profile_enable(True)


PROFILE_BATCH_EXPERIENCES = 0
PROFILE_BATCH_STATES_U8 = 1
PROFILE_GAE_SCAN = 2
PROFILE_ADV_STATS = 3
PROFILE_BATCH_STATES_U8_RAW = 4     # (pfrl_batch_states_u8_raw_nhwc4: 2 bytes per frame byte)


def profile_collect(kind=PROFILE_BATCH_EXPERIENCES, cap=1 << 16):
    """-> (durations_us, units) of the launches of ``kind`` timed since
    profile_enable(True); kind=None returns (durations_us, units, kinds) of all."""
    us = (ctypes.c_double * cap)()
    ent = (ctypes.c_int64 * cap)()
    kinds = (ctypes.c_int32 * cap)()
    n = _native.lib().pfrl_profile_collect(us, ent, kinds, cap)
    if kind is None:
        return list(us[:n]), list(ent[:n]), list(kinds[:n])
    sel = [i for i in range(n) if kinds[i] == kind]
    return [us[i] for i in sel], [ent[i] for i in sel]


def batch_experiences(desc, frames, divisor, entry_slots, gamma_pow, out):
    """Fused n-step collapse + state/next_state gathers.  ``out`` is a dict of
    preallocated tensors: state, next_state, action, reward, is_state_terminal,
    discount."""
    B = entry_slots.numel()
    gp = (ctypes.c_double * len(gamma_pow))(*gamma_pow)
    st = out["state"]
    if st.dim() == 4 and not st.is_contiguous() and \
            st.is_contiguous(memory_format=torch.channels_last):
        # channels_last minibatch buffers: the kernel writes [B][H][W][4] directly
        assert channels_last_supported(frames, desc.k)
        assert out["next_state"].is_contiguous(memory_format=torch.channels_last)
        check(_native.lib().pfrl_batch_experiences_nhwc4(
            ctypes.byref(desc), _ptr(frames), frame_bytes_of(frames), float(divisor),
            _ptr(entry_slots), B, ctypes.cast(gp, ctypes.c_void_p),
            ctypes.c_void_p(st.data_ptr()), ctypes.c_void_p(out["next_state"].data_ptr()),
            _ptr(out["action"]), _ptr(out["reward"]), _ptr(out["is_state_terminal"]),
            _ptr(out["discount"]), _stream()), "batch_experiences_nhwc4")
        return out
    check(_native.lib().pfrl_batch_experiences(
        ctypes.byref(desc), _ptr(frames), frame_bytes_of(frames),
        int(frames.dtype == torch.float32), float(divisor), _ptr(entry_slots), B,
        ctypes.cast(gp, ctypes.c_void_p), _ptr(out["state"]), _ptr(out["next_state"]),
        _ptr(out["action"]), _ptr(out["reward"]), _ptr(out["is_state_terminal"]),
        _ptr(out["discount"]), _stream()), "batch_experiences")
    return out


def batch_episodes(desc, frames, divisor, ep_first, ep_row0, row_start, n_eps, T, rows, entry_ring,
                   gamma, out):
    """Ragged gather of sampled episode windows (pfrl_batch_episodes); ``out`` as for
    batch_experiences, state / next_state episode-major, scalars time-major packed."""
    check(_native.lib().pfrl_batch_episodes(
        ctypes.byref(desc), _ptr(frames), frame_bytes_of(frames),
        int(frames.dtype == torch.float32), float(divisor), _ptr(ep_first), _ptr(ep_row0),
        _ptr(row_start), int(n_eps), int(T), int(rows), int(entry_ring), float(gamma),
        _ptr(out["state"]), _ptr(out["next_state"]), _ptr(out["action"]), _ptr(out["reward"]),
        _ptr(out["is_state_terminal"]), _ptr(out["discount"]), _stream()), "batch_episodes")
    return out


def tree_write(desc, x, val, tag, use_maxp):
    check(_native.lib().pfrl_tree_write(ctypes.byref(desc), x.numel(), _ptr(x), _ptr(val),
                                        _ptr(tag), _ptr(use_maxp), _stream()), "tree_write")


def tree_write_sum(desc, x, val=None, tag=None, old_val=None, old_tag=None):
    """TreeQueue._write on the sum tree only (pfrl_tree_write_sum): ``val`` / ``tag`` None writes
    Python-float zeros; ``old_val`` / ``old_tag`` receive the leaves' previous contents."""
    check(_native.lib().pfrl_tree_write_sum(
        ctypes.byref(desc), x.numel(), _ptr(x), _ptr(val) if val is not None else None,
        _ptr(tag) if tag is not None else None, _ptr(old_val) if old_val is not None else None,
        _ptr(old_tag) if old_tag is not None else None, _stream()), "tree_write_sum")


def tree_sample(desc, u01, out, normalize, beta, slot_mod=0):
    B = u01.numel()
    check(_native.lib().pfrl_tree_sample(
        ctypes.byref(desc), B, _ptr(u01), _ptr(out["x"]), _ptr(out["pri"]), _ptr(out["pri_tag"]),
        _ptr(out["prob"]), _ptr(out["weight"]), _ptr(out["total"]), _ptr(out["total_tag"]),
        _ptr(out["min_prob"]), int(normalize), float(beta), int(slot_mod), _ptr(out.get("slot")),
        _stream()), "tree_sample")
    return out


POW_CORRECTLY_ROUNDED, POW_GLIBC, POW_GLIBC_FMA = 0, 1, 2   # PFRL_POW_* of include/pfrl_amd.h
_powf_variant_cache = {}


def powf_host_variant(alpha, n_probe=1 << 18):
    """Which restatement of glibc's powf (POW_GLIBC / POW_GLIBC_FMA) reproduces THIS host's
    libm bit for bit -- i.e. what ``np.float32(x) ** alpha`` gives here -- or None if neither
    does (a libm that is not glibc's).  Probed once per alpha."""
    key = float(np.float32(alpha))
    if key not in _powf_variant_cache:
        v = _native.lib().pfrl_powf_host_variant(key, int(n_probe))
        _powf_variant_cache[key] = None if v < 0 else int(v)
    return _powf_variant_cache[key]


def powf_host(x, alpha, pow_mode):
    """The restated powf on a host float32 array (tests)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    check(_native.lib().pfrl_powf_host(int(pow_mode), x.ctypes.data, float(np.float32(alpha)),
                                       out.ctypes.data, x.size), "powf_host")
    return out


def powf_device(x, alpha, pow_mode):
    """The update kernel's power on a float32 device tensor (tests)."""
    x = x.contiguous()
    out = torch.empty_like(x)
    check(_native.lib().pfrl_powf_device(int(pow_mode), _ptr(x), float(np.float32(alpha)), _ptr(out),
                                         x.numel(), _stream()), "powf_device")
    return out


def tree_update_errors_f32(desc, x, err, error_min, pri_at_min, error_max, pri_at_max, eps, alpha,
                           dedupe=True, pow_mode=POW_CORRECTLY_ROUNDED):
    check(_native.lib().pfrl_tree_update_errors_f32(
        ctypes.byref(desc), x.numel(), _ptr(x), _ptr(err),
        int(error_min is not None), float(error_min or 0.0), float(pri_at_min or 0.0),
        int(error_max is not None), float(error_max or 0.0), float(pri_at_max or 0.0),
        float(eps), float(alpha), int(dedupe), int(pow_mode), _stream()), "tree_update_errors_f32")


def tree_update_errors_write_f32(desc, x, err, error_min, pri_at_min, error_max, pri_at_max, eps,
                                 alpha, writes, dedupe=True, pow_mode=POW_CORRECTLY_ROUNDED):
    """``tree_update_errors_f32`` followed by ``tree_write(*writes)`` as one launch."""
    wx, wv, wt, wm = writes
    check(_native.lib().pfrl_tree_update_errors_write_f32(
        ctypes.byref(desc), x.numel(), _ptr(x), _ptr(err),
        int(error_min is not None), float(error_min or 0.0), float(pri_at_min or 0.0),
        int(error_max is not None), float(error_max or 0.0), float(pri_at_max or 0.0),
        float(eps), float(alpha), int(dedupe), int(pow_mode), wx.numel(), _ptr(wx), _ptr(wv),
        _ptr(wt), _ptr(wm), _stream()), "tree_update_errors_write_f32")


def tree_update_errors_write_sample(desc, x, err, error_min, pri_at_min, error_max, pri_at_max, eps,
                                    alpha, writes, u01, out, normalize, beta, slot_mod=0, dedupe=True,
                                    pow_mode=POW_CORRECTLY_ROUNDED):
    """``tree_update_errors_write_f32`` (``writes`` = (x, val, tag, use_maxp) or None) followed by
    ``tree_sample`` as one launch (pfrl_tree_update_errors_write_sample)."""
    if writes is not None:
        wx, wv, wt, wm = writes
        n = wx.numel()
    else:
        wx = wv = wt = wm = None
        n = 0
    check(_native.lib().pfrl_tree_update_errors_write_sample(
        ctypes.byref(desc), x.numel(), _ptr(x), _ptr(err),
        int(error_min is not None), float(error_min or 0.0), float(pri_at_min or 0.0),
        int(error_max is not None), float(error_max or 0.0), float(pri_at_max or 0.0),
        float(eps), float(alpha), int(dedupe), int(pow_mode), n,
        _ptr(wx) if n else None, _ptr(wv) if n else None, _ptr(wt) if n else None,
        _ptr(wm) if n else None, u01.numel(), _ptr(u01), _ptr(out["x"]), _ptr(out["pri"]),
        _ptr(out["pri_tag"]), _ptr(out["prob"]), _ptr(out["weight"]), _ptr(out["total"]),
        _ptr(out["total_tag"]), _ptr(out["min_prob"]), int(normalize), float(beta), int(slot_mod),
        _ptr(out.get("slot")), _stream()), "tree_update_errors_write_sample")
    return out


def tree_set_priorities(desc, x, val, tag, dedupe=True):
    check(_native.lib().pfrl_tree_set_priorities(ctypes.byref(desc), x.numel(), _ptr(x), _ptr(val),
                                                 _ptr(tag), int(dedupe), _stream()),
          "tree_set_priorities")


def gae_scan(reward, v_pred, next_v_pred, nonterminal, cut, gamma, lambd, mode=0):
    """All inputs [T, N] on the device; reward float64, v f32, flags uint8."""
    T, N = reward.shape
    adv = torch.empty((T, N), dtype=torch.float32, device=reward.device)
    vt = torch.empty_like(adv)
    check(_native.lib().pfrl_gae_scan(T, N, _ptr(reward), _ptr(v_pred), _ptr(next_v_pred),
                                      _ptr(nonterminal), _ptr(cut), float(gamma), float(lambd),
                                      int(mode), _ptr(adv), _ptr(vt), _stream()), "gae_scan")
    return adv, vt


def ppo_loss(logits, value, action, adv, log_prob_old, v_pred_old, v_teacher, clip_eps, clip_eps_vf,
             value_func_coef, entropy_coef):
    """PPO._lossfun (reference pfrl/agents/ppo.py:634-671) and its gradient with respect to the
    logits [M, A] and values [M(, 1)] of a minibatch in one launch (pfrl_ppo_loss).  Returns (out4 =
    [loss, loss_policy, loss_value, mean entropy], dlogits [M, A], dvalue shaped like ``value``);
    the caller starts backward at the logits / values with these gradients."""
    M, A = logits.shape
    logits, v = logits.detach().contiguous(), value.detach().reshape(-1).contiguous()
    dev = logits.device
    dlogits = torch.empty_like(logits)
    dvalue = torch.empty(M, dtype=torch.float32, device=dev)
    ws = torch.empty(3 * ((M + 255) // 256), dtype=torch.float64, device=dev)
    out = torch.empty(4, dtype=torch.float32, device=dev)
    check(_native.lib().pfrl_ppo_loss(
        _ptr(logits), _ptr(v), _ptr(action), _ptr(adv.reshape(-1)), _ptr(log_prob_old.reshape(-1)),
        _ptr(v_pred_old.reshape(-1)) if clip_eps_vf is not None else None, _ptr(v_teacher.reshape(-1)),
        M, A, float(clip_eps), -1.0 if clip_eps_vf is None else float(clip_eps_vf),
        float(value_func_coef), float(entropy_coef), _ptr(dlogits), _ptr(dvalue), _ptr(ws), _ptr(out),
        _stream()), "ppo_loss")
    return out, dlogits, dvalue.view(value.shape)


def ppo_head_loss_ok(h, w_policy):
    """Shapes the one-launch heads + loss + heads' backward covers (pfrl_ppo_head_loss)."""
    return (h.is_cuda and h.dim() == 2 and h.dtype == torch.float32 and h.shape[1] in (256, 512)
            and 1 <= w_policy.shape[0] <= 9)


def ppo_head_loss(h, w_policy, b_policy, w_value, b_value, action, adv, log_prob_old, v_pred_old,
                  v_teacher, clip_eps, clip_eps_vf, value_func_coef, entropy_coef):
    """The two narrow heads on the body's output h [M, K], PPO._lossfun and the heads' backward in one
    launch (pfrl_ppo_head_loss) + the fold of the per-workgroup gradient slabs.  Returns (out4 =
    [loss, loss_policy, loss_value, mean entropy], dh [M, K], (dWp, dbp, dWv, dbv)): backward of the
    body starts at h with dh; the four head gradients are final."""
    from pfrl_amd.nn import mfma_trunk as _t

    M, K = h.shape
    A = w_policy.shape[0]
    h = h.detach().contiguous()
    dev = h.device
    NO = A + 1
    blocks = min(512, (M + 7) // 8)
    stride = NO * K + (NO + 3) // 4 * 4
    dh = torch.empty_like(h)
    part = torch.empty(blocks * stride, dtype=torch.float32, device=dev)
    ws = torch.empty(3 * blocks, dtype=torch.float64, device=dev)
    out = torch.empty(4, dtype=torch.float32, device=dev)
    check(_native.lib().pfrl_ppo_head_loss(
        _ptr(h), _ptr(w_policy), _ptr(b_policy), _ptr(w_value), _ptr(b_value), _ptr(action),
        _ptr(adv.reshape(-1)), _ptr(log_prob_old.reshape(-1)),
        _ptr(v_pred_old.reshape(-1)) if clip_eps_vf is not None else None, _ptr(v_teacher.reshape(-1)),
        M, K, A, float(clip_eps), -1.0 if clip_eps_vf is None else float(clip_eps_vf),
        float(value_func_coef), float(entropy_coef), _ptr(dh), _ptr(part), blocks, _ptr(ws), _ptr(out),
        _stream()), "ppo_head_loss")
    dwp = torch.empty_like(w_policy)
    dbp = torch.empty_like(b_policy)
    dwv = torch.empty_like(w_value)
    dbv = torch.empty_like(b_value)
    _t._reduce([(part, dwp, None, stride, A * K, blocks, 4, 0),
                (part[A * K:], dwv, None, stride, K, blocks, 4, 0),
                (part[NO * K:], dbp, None, stride, A, blocks, 1, 0),
                (part[NO * K + A:], dbv, None, stride, 1, blocks, 1, 0)])
    return out, dh, (dwp, dbp, dwv, dbv)


def ppo_act_head(h, w_policy, b_policy, w_value, b_value, u01, want_log_prob=False, into=None):
    """The two narrow heads of the PPO example network + Categorical sample / entropy in one launch
    (pfrl_ppo_act_head).  h [N, K] f32; returns (action i64 [N], entropy [N], value [N][, log_prob]).
    ``into`` = (action column i64 [T, N], stats ring f32 [R, 2, N], rows int32[2] on the device):
    the launch writes the action into row rows[0] of the column and (entropy, value) into slot
    rows[1] of the ring -- the indices are read on the device, so a captured graph keeps fixed
    arguments -- and the returned tensors are the BASES (index them with the host's copy of rows)."""
    N, K = h.shape
    A = w_policy.shape[0]
    dev = h.device
    rows = None
    if into is not None:
        col, ring, rows = into
        assert col.dtype == torch.int64 and col.is_contiguous() and col.shape[1] == N
        assert ring.dtype == torch.float32 and ring.is_contiguous() and tuple(ring.shape[1:]) == (2, N)
        assert rows.dtype == torch.int32 and rows.numel() == 2
        action, entropy, value = col, ring, ring.view(-1)[N:]
    else:
        action = torch.empty(N, dtype=torch.int64, device=dev)
        entropy = torch.empty(N, dtype=torch.float32, device=dev)
        value = torch.empty(N, dtype=torch.float32, device=dev)
    logp = torch.empty(N, dtype=torch.float32, device=dev) if want_log_prob else None
    check(_native.lib().pfrl_ppo_act_head(_ptr(h), _ptr(w_policy), _ptr(b_policy), _ptr(w_value),
                                          _ptr(b_value), _ptr(u01), None, _ptr(action), _ptr(entropy),
                                          _ptr_dense(value), _ptr(logp) if logp is not None else None,
                                          N, K, A, _ptr(rows) if rows is not None else None, _stream()),
          "ppo_act_head")
    return (action, entropy, value, logp) if want_log_prob else (action, entropy, value)


def ppo_value_head(h, w_policy, b_policy, w_value, b_value, actions, out_log_prob, out_value):
    """The same launch without a draw: log pi(actions | s) (actions = None: values only) and V(s)
    written into ``out_log_prob`` / ``out_value`` [N] (the value pass of a PPO update)."""
    N, K = h.shape
    check(_native.lib().pfrl_ppo_act_head(
        _ptr(h), _ptr(w_policy), _ptr(b_policy), _ptr(w_value), _ptr(b_value), None,
        _ptr(actions) if actions is not None else _ptr(_zeros_i64(N, h.device)), None, None,
        _ptr(out_value), _ptr(out_log_prob) if actions is not None else None, N, K,
        w_policy.shape[0], None, _stream()), "ppo_value_head")


_ZI64 = {}


def _zeros_i64(n, device):
    t = _ZI64.get(device)
    if t is None or t.numel() < n:
        t = _ZI64[device] = torch.zeros(max(n, 1 << 16), dtype=torch.int64, device=device)
    return t


def a2c_returns(rewards, masks, value_preds, returns, gamma, tau, use_gae):
    T, N = rewards.shape
    check(_native.lib().pfrl_a2c_returns(T, N, _ptr(rewards), _ptr(masks), _ptr(value_preds),
                                         _ptr(returns), float(gamma), float(tau), int(use_gae),
                                         _stream()), "a2c_returns")
    return returns


_ws_cache = {}


def adv_stats(adv):
    """-> f32 tensor [2] = (mean, std(unbiased=False)), left on the device."""
    key = adv.device
    ws = _ws_cache.get(key)
    if ws is None:
        ws = _ws_cache[key] = torch.empty(2 * 1024, dtype=torch.float64, device=adv.device)
    out = torch.empty(2, dtype=torch.float32, device=adv.device)
    check(_native.lib().pfrl_adv_stats(_ptr(adv), adv.numel(), _ptr(out), _ptr(ws), _stream()),
          "adv_stats")
    return out


def ppo_minibatch(idx, adv, mean_std, standardize, log_prob, v_pred, v_teacher, action,
                  state_refs):
    M = idx.numel()
    k = state_refs.shape[1]
    dev = adv.device
    out = dict(
        adv=torch.empty(M, dtype=torch.float32, device=dev),
        log_prob=torch.empty(M, dtype=torch.float32, device=dev),
        v_pred=torch.empty(M, dtype=torch.float32, device=dev),
        v_teacher=torch.empty(M, dtype=torch.float32, device=dev),
        action=torch.empty(M, dtype=torch.int64, device=dev),
        refs=torch.empty((M, k), dtype=torch.int32, device=dev),
    )
    check(_native.lib().pfrl_ppo_minibatch(
        M, _ptr(idx), _ptr(adv), _ptr(mean_std), int(standardize), _ptr(log_prob), _ptr(v_pred),
        _ptr(v_teacher), _ptr(action), _ptr(state_refs), k, _ptr(out["adv"]),
        _ptr(out["log_prob"]), _ptr(out["v_pred"]), _ptr(out["v_teacher"]), _ptr(out["action"]),
        _ptr(out["refs"]), _stream()), "ppo_minibatch")
    return out


class _DQNTDLoss(torch.autograd.Function):
    """loss = sum/mean_b w_b L(Q(s)[a] - target); one HIP launch computes the loss,
    its gradient w.r.t. Q(s), the selected Q values and |TD error|."""

    @staticmethod
    def forward(ctx, q, action, target_q, next_q_online, reward, discount, terminal, weights,
                clip_delta, mean):
        B, A = q.shape
        qc = q.detach().contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=q.device)
        grad_q = torch.empty((B, A), dtype=torch.float32, device=q.device)
        y = torch.empty(B, dtype=torch.float32, device=q.device)
        delta = torch.empty(B, dtype=torch.float32, device=q.device)
        check(_native.lib().pfrl_dqn_td_loss(
            _ptr(qc), _ptr(action.contiguous()), _ptr(target_q.contiguous()),
            _ptr(next_q_online.contiguous()) if next_q_online is not None else None,
            _ptr(reward), _ptr(discount), _ptr(terminal),
            _ptr(weights.contiguous()) if weights is not None else None, B, A, int(clip_delta),
            int(mean), _ptr(loss), _ptr(grad_q), _ptr(y), _ptr(delta), _stream()), "dqn_td_loss")
        ctx.save_for_backward(grad_q)
        ctx.mark_non_differentiable(y, delta)
        ctx.set_materialize_grads(False)   # no zero-fill kernels for the unused outputs
        return loss.view(()), y, delta

    @staticmethod
    def backward(ctx, g_loss, g_y, g_delta):
        (grad_q,) = ctx.saved_tensors
        return (grad_q * g_loss,) + (None,) * 9


def dqn_td_loss(q, action, target_q, next_q_online, reward, discount, terminal, weights,
                clip_delta, mean):
    """-> (loss scalar with grad, y [B], |y - t| [B])"""
    return _DQNTDLoss.apply(q, action, target_q, next_q_online, reward, discount, terminal,
                            weights, clip_delta, mean)


class _DQNHeadTDLoss(torch.autograd.Function):
    """The narrow head ``q = h W^T + b``, the TD loss of ``_DQNTDLoss`` and the head's backward
    in one launch (pfrl_dqn_head_td_loss, a wave per row).  The sums over the batch (dL/dW,
    dL/db, the loss) leave the launch as per-row partials; ``defer=True`` queues their fold for
    the fold launch that ends the MFMA trunk's backward (``mfma_trunk.defer_fold``), otherwise
    it is launched here."""

    @staticmethod
    def forward(ctx, h, w, b, action, target_q, next_q_online, reward, discount, terminal, weights,
                clip_delta, mean, defer, h_fold=None):
        from pfrl_amd.nn import mfma_trunk

        B, K = h.shape
        A = w.shape[0]
        hc = h.detach().contiguous()
        if h_fold is not None:
            # h is still the hidden layer's split-K slabs: this launch folds them row by row and
            # fills h itself (the tensor the trunk saved for its backward pass)
            f_part, f_bias, f_stride, f_splits = h_fold
            assert hc.data_ptr() == h.data_ptr()
            fold_args = (_ptr(f_part), int(f_splits), int(f_stride), _ptr(f_bias), _ptr(hc))
        else:
            fold_args = (None, 0, 0, None, None)
        dev = h.device
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        y = torch.empty(B, dtype=torch.float32, device=dev)
        delta = torch.empty(B, dtype=torch.float32, device=dev)
        dh = torch.empty((B, K), dtype=torch.float32, device=dev)
        dw = torch.empty((A, K), dtype=torch.float32, device=dev)
        db = torch.empty((A,), dtype=torch.float32, device=dev)
        stride = A * K + 32
        slabs = (B + 3) // 4          # one partial slab per workgroup of four rows
        part = torch.empty(slabs * stride, dtype=torch.float32, device=dev)
        # data parallel: the masked, 1 / world-scaled copy of dh the hidden layer's low-rank exchange
        # all-gathers comes out of this launch too (two elementwise launches less per update)
        dh_masked, dh_scale = None, 1.0
        mfma_trunk.MASKED_DH.clear()
        d = torch.distributed
        if (d.is_available() and d.is_initialized()
                and os.environ.get("PFRL_DP_FUSED_MASK", "1") != "0"):
            world = d.get_world_size()
            dh_masked = torch.empty((B, K), dtype=torch.float32, device=dev)
            dh_scale = 1.0 / world
            mfma_trunk.MASKED_DH[dh.data_ptr()] = (dh_masked, world)
        check(_native.lib().pfrl_dqn_head_td_loss(
            _ptr(hc), _ptr(w.detach()), _ptr(b.detach()), _ptr(action.contiguous()),
            _ptr(target_q.contiguous()),
            _ptr(next_q_online.contiguous()) if next_q_online is not None else None,
            _ptr(reward), _ptr(discount), _ptr(terminal),
            _ptr(weights.contiguous()) if weights is not None else None, B, K, A, int(clip_delta),
            int(mean), _ptr(y), _ptr(delta), _ptr(dh), _ptr(part), *fold_args,
            _ptr(dh_masked) if dh_masked is not None else None, float(dh_scale), _stream()),
            "dqn_head_td_loss")
        tasks = [(part, dw, None, stride, A * K, slabs, 4, 0),
                 (part[A * K:], db, None, stride, A, slabs, 4, 0),
                 (part[A * K + 16:], loss, None, stride, 1, slabs, 4, 0)]
        if defer:
            mfma_trunk.defer_fold(tasks)
        else:
            mfma_trunk._reduce(tasks)
        ctx.save_for_backward(dh, dw, db)
        ctx.mark_non_differentiable(y, delta)
        ctx.set_materialize_grads(False)
        return loss.view(()), y, delta

    @staticmethod
    def backward(ctx, g_loss, g_y, g_delta):
        dh, dw, db = ctx.saved_tensors
        return (dh * g_loss, dw * g_loss, db * g_loss) + (None,) * 11


def dqn_head_td_loss_supported(h, w, b):
    return (h.is_cuda and h.dim() == 2 and h.dtype == torch.float32 and w.dtype == torch.float32
            and b is not None and w.is_contiguous() and 1 <= w.shape[0] <= 16
            and h.shape[1] in (256, 512) and 1 <= h.shape[0] <= 4096 and _native.available())


def dqn_head_td_loss(h, w, b, action, target_q, next_q_online, reward, discount, terminal, weights,
                     clip_delta, mean, defer=False, h_fold=None):
    """-> (loss scalar with grad w.r.t. h, w, b; y [B]; |y - t| [B]).  With ``defer`` the loss
    value and the head's gradients are final only after the deferred fold has run (the trunk's
    backward or ``mfma_trunk.flush_deferred_folds()``).  ``h_fold`` = (part, bias, stride,
    splits): ``h`` has not been folded from the hidden layer's split-K slabs yet
    (``mfma_trunk.FWD_FOLD_SINK``) -- this launch does it and fills ``h``."""
    return _DQNHeadTDLoss.apply(h, w, b, action, target_q, next_q_online, reward, discount, terminal,
                                weights, clip_delta, mean, defer, h_fold)


_bias_relu_ws = {}


def _bias_relu_plan(rows, C):
    """Workgroups of the backward launch: ~8 rows per row-lane; at most 256 for the
    small minibatches of the replay agents (the last workgroup folds blocks * C
    partials), up to 2048 for rollout-sized batches, which need the whole chip."""
    rstep = 1024 // C                 # row lanes of a workgroup (float4 per thread)
    cap = 256 if rows <= (1 << 18) else 2048
    return max(1, min(cap, -(-rows // (rstep * 8))))


def _bias_relu_workspace(device, C, blocks):
    key = (device, C, blocks)
    ws = _bias_relu_ws.get(key)
    if ws is None:
        ws = _bias_relu_ws[key] = (
            torch.zeros(blocks * C, dtype=torch.int64, device=device),
            torch.zeros(2, dtype=torch.int64, device=device))
    return ws


class _BiasReLU(torch.autograd.Function):
    """y = relu(x + bias[c]) for channels_last conv outputs / [N, C] matrices.  With
    ``planar`` (4-D channels_last x only) y comes out as a plain contiguous NCHW tensor,
    so that a following flatten is a view; the gradient is then expected in NCHW too."""

    @staticmethod
    def forward(ctx, x, bias, planar=False):
        C = bias.numel()
        rows = x.numel() // C
        hw = 0
        if planar:
            hw = x.shape[2] * x.shape[3]
            y = torch.empty(x.shape, dtype=torch.float32, device=x.device)   # NCHW
        else:
            y = torch.empty_like(x)   # preserves the (dense) layout of x
        check(_native.lib().pfrl_bias_relu_fwd(_ptr_dense(x), _ptr(bias), _ptr_dense(y), rows, C,
                                               hw, _stream()), "bias_relu_fwd")
        ctx.save_for_backward(y)
        ctx.C = C
        ctx.hw = hw
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        C = ctx.C
        rows = y.numel() // C
        if gy.stride() != y.stride():
            gy = gy.contiguous(memory_format=torch.channels_last) \
                if (y.dim() == 4 and not ctx.hw) else gy.contiguous()
        if ctx.hw:
            gx = torch.empty(y.shape, dtype=torch.float32, device=y.device).contiguous(
                memory_format=torch.channels_last)
        else:
            gx = torch.empty_like(y)
        gb = torch.empty(C, dtype=torch.float32, device=y.device)
        if _bias_relu_small(y, C):
            check(_native.lib().pfrl_bias_relu_bwd(_ptr_dense(gy), _ptr_dense(y), _ptr_dense(gx),
                                                   _ptr(gb), None, None, rows, C, 0, 0,
                                                   _stream()), "bias_relu_bwd")
            return gx, gb, None
        blocks = _bias_relu_plan(rows, C)
        ws, counters = _bias_relu_workspace(y.device, C, blocks)
        check(_native.lib().pfrl_bias_relu_bwd(_ptr_dense(gy), _ptr_dense(y), _ptr_dense(gx),
                                               _ptr(gb), _ptr(ws), _ptr(counters), rows, C,
                                               blocks, ctx.hw, _stream()), "bias_relu_bwd")
        return gx, gb, None


def _ptr_dense(t):
    assert t.is_cuda
    return ctypes.c_void_p(t.data_ptr())


def _bias_relu_small(x, C):
    """2-D activations with few rows (a hidden linear layer at minibatch size): the
    single-workgroup backward, any C % 4 == 0."""
    return x.dim() == 2 and x.shape[0] <= 256 and C % 4 == 0


def bias_relu_supported(x, bias):
    """Row-major [rows][C] view available?  (channels_last 4-D or contiguous 2-D)"""
    if not (x.is_cuda and x.dtype == torch.float32 and bias is not None):
        return False
    C = bias.numel()
    if x.dim() == 2 and x.shape[1] == C and x.is_contiguous() and _bias_relu_small(x, C):
        return True
    if C % 4 or 256 % C:
        return False
    if x.dim() == 4:
        return x.shape[1] == C and x.is_contiguous(memory_format=torch.channels_last)
    if x.dim() == 2:
        return x.shape[1] == C and x.is_contiguous()
    return False


def bias_relu(x, bias, planar=False):
    """``planar=True``: NCHW-contiguous result from a channels_last ``x`` (see _BiasReLU)."""
    return _BiasReLU.apply(x, bias, bool(planar and x.dim() == 4))


class _NoisyWeights(torch.autograd.Function):
    """(W, b) of a factorised noisy layer from (mu, sigma, r) in one launch;
    b is None for layers without bias."""

    @staticmethod
    def forward(ctx, mu_w, sigma_w, mu_b, sigma_b, r):
        out_f, in_f = sigma_w.shape
        w = torch.empty((out_f, in_f), dtype=torch.float32, device=sigma_w.device)
        hasbias = mu_b is not None
        b = torch.empty(out_f, dtype=torch.float32, device=sigma_w.device) if hasbias else None
        check(_native.lib().pfrl_noisy_weights_fwd(
            _ptr(mu_w), _ptr(sigma_w), _ptr(mu_b) if hasbias else None,
            _ptr(sigma_b) if hasbias else None, _ptr(r), _ptr(w), _ptr(b) if hasbias else None,
            out_f, in_f, _stream()), "noisy_weights_fwd")
        ctx.save_for_backward(r)
        ctx.shape = (out_f, in_f)
        ctx.hasbias = hasbias
        ctx.set_materialize_grads(False)
        return w, b

    @staticmethod
    def backward(ctx, g_w, g_b):
        (r,) = ctx.saved_tensors
        out_f, in_f = ctx.shape
        if g_w is None:
            g_w = torch.zeros((out_f, in_f), dtype=torch.float32, device=r.device)
        g_w = g_w.contiguous()
        want_b = ctx.hasbias and g_b is not None
        if want_b:
            g_b = g_b.contiguous()
        g_sw = torch.empty_like(g_w)
        g_sb = torch.empty_like(g_b) if want_b else None
        check(_native.lib().pfrl_noisy_weights_bwd(
            _ptr(g_w), _ptr(g_b) if want_b else None, _ptr(r), _ptr(g_sw),
            _ptr(g_sb) if want_b else None, out_f, in_f, _stream()), "noisy_weights_bwd")
        return g_w, g_sw, (g_b if want_b else None), g_sb, None


def noisy_weights_supported(sigma_w):
    return sigma_w.is_cuda and sigma_w.dtype == torch.float32 and sigma_w.is_contiguous()


def noisy_weights(mu_w, sigma_w, mu_b, sigma_b, r):
    """Perturbed (weight, bias) of a FactorizedNoisyLinear; ``r`` = in + out unit
    Gaussians in the reference's order (eps_x first)."""
    return _NoisyWeights.apply(mu_w, sigma_w, mu_b, sigma_b, r)


class _C51Loss(torch.autograd.Function):
    """Categorical DQN loss (projection + cross entropy) in one HIP launch; the
    same launch produces d loss / d q_dist, Q(s, a) and the per-sample KL."""

    @staticmethod
    def forward(ctx, q_dist, action, next_dist, next_select, z_values, reward, discount, terminal,
                weights, mean):
        B, A, Z = q_dist.shape
        dev = q_dist.device
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        grad_q = torch.empty((B, A, Z), dtype=torch.float32, device=dev)
        qsa = torch.empty(B, dtype=torch.float32, device=dev)
        delta = torch.empty(B, dtype=torch.float32, device=dev)
        check(_native.lib().pfrl_c51_loss(
            _ptr(q_dist.detach().contiguous()), _ptr(action.contiguous()),
            _ptr(next_dist.contiguous()),
            _ptr(next_select.contiguous()) if next_select is not None else None,
            _ptr(z_values.contiguous()), _ptr(reward), _ptr(discount), _ptr(terminal),
            _ptr(weights.contiguous()) if weights is not None else None, B, A, Z, int(mean),
            _ptr(loss), _ptr(grad_q), _ptr(qsa), _ptr(delta), _stream()), "c51_loss")
        ctx.save_for_backward(grad_q)
        ctx.mark_non_differentiable(qsa, delta)
        ctx.set_materialize_grads(False)
        return loss.view(()), qsa, delta

    @staticmethod
    def backward(ctx, g_loss, g_q, g_delta):
        (grad_q,) = ctx.saved_tensors
        return (grad_q * g_loss,) + (None,) * 9


def c51_loss_supported(q_dist):
    return (q_dist.is_cuda and q_dist.dtype == torch.float32 and q_dist.ndim == 3
            and 2 <= q_dist.shape[2] <= 64 and q_dist.shape[0] <= 4096)


def c51_loss(q_dist, action, next_dist, next_select, z_values, reward, discount, terminal, weights,
             mean):
    """-> (loss scalar with grad, Q(s, a) [B], per-sample cross entropy [B])"""
    return _C51Loss.apply(q_dist, action, next_dist, next_select, z_values, reward, discount,
                          terminal, weights, mean)


class _DuelingSoftmax(torch.autograd.Function):
    """q = softmax_atoms(ya - mean_actions(ya) + ys) for the distributional dueling head."""

    @staticmethod
    def forward(ctx, ya, ys, n_actions, n_atoms):
        B = ya.shape[0]
        q = torch.empty((B, n_actions, n_atoms), dtype=torch.float32, device=ya.device)
        check(_native.lib().pfrl_dueling_softmax_fwd(_ptr(ya.contiguous()), _ptr(ys.contiguous()),
                                                     _ptr(q), B, n_actions, n_atoms, _stream()),
              "dueling_softmax_fwd")
        ctx.save_for_backward(q)
        return q

    @staticmethod
    def backward(ctx, gq):
        (q,) = ctx.saved_tensors
        B, A, Z = q.shape
        g_ya = torch.empty((B, A * Z), dtype=torch.float32, device=q.device)
        g_ys = torch.empty((B, Z), dtype=torch.float32, device=q.device)
        check(_native.lib().pfrl_dueling_softmax_bwd(_ptr(gq.contiguous()), _ptr(q), _ptr(g_ya),
                                                     _ptr(g_ys), B, A, Z, _stream()),
              "dueling_softmax_bwd")
        return g_ya, g_ys, None, None


def dueling_softmax_supported(ya, n_atoms):
    return ya.is_cuda and ya.dtype == torch.float32 and 1 <= n_atoms <= 64


def dueling_softmax(ya, ys, n_actions, n_atoms):
    """ya [B, A*Z] (or [B, A, Z]), ys [B, Z] -> q [B, A, Z]."""
    return _DuelingSoftmax.apply(ya, ys, n_actions, n_atoms)


# ---------------------------------------------------------------------------------------------
# torch.randn on the device generator as ONE launch for several draws (csrc/philox.hip)
# ---------------------------------------------------------------------------------------------
_PHILOX_VARIANT = {}


def _default_generator(device):
    torch.cuda.init()          # (the tuple of default generators is filled by the lazy init)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    return torch.cuda.default_generators[idx]


def _randn_grid(numel, device):
    """Blocks of 256 threads torch launches for a float normal_ of ``numel`` elements
    (ATen/native/cuda/DistributionTemplates.h calc_execution_policy)."""
    prop = torch.cuda.get_device_properties(device)
    cap = prop.multi_processor_count * (prop.max_threads_per_multi_processor // 256)
    return max(1, min(cap, (numel + 255) // 256))


def _randn_increment(numel, grid):
    """By how much that call advances the generator's Philox offset."""
    inc = ((numel - 1) // (256 * grid * 4) + 1) * 4
    return (inc + 3) // 4 * 4


class RandnPlan:
    """``[torch.randn(n, device=device) for n in sizes]`` on the device's default generator as a
    prepared launch sequence: the argument arrays are built once, ``run()`` reads the generator's
    (seed, offset) on the host, launches (one launch per 16 draws) and advances the offset as the
    torch calls would.  ``views[i]`` is draw i inside ``out`` (16-byte aligned offsets)."""

    def __init__(self, sizes, device, out=None, variant=None):
        device = torch.device(device)
        if variant is None:
            variant = philox_variant(device)
            assert variant is not None, "pfrl_philox_normal does not reproduce torch.randn on this stack"
        self.variant, self.device = int(variant), device
        self.gen = _default_generator(device)
        self.sizes = [int(n) for n in sizes]
        rel, grids, outs = [], [], []
        pos = off = 0
        for numel in self.sizes:
            g = _randn_grid(numel, device)
            rel.append(off)
            grids.append(g)
            outs.append(pos)
            off += _randn_increment(numel, g)
            pos += (numel + 3) & ~3
        self.total_increment = off
        if out is None:
            out = torch.empty(pos, dtype=torch.float32, device=device)
        assert out.numel() >= pos and out.dtype == torch.float32 and out.is_contiguous()
        self.out = out
        self.views = [out[o:o + k] for o, k in zip(outs, self.sizes)]
        self.launches = []
        for lo in range(0, len(self.sizes), 16):
            hi = min(len(self.sizes), lo + 16)
            m = hi - lo
            self.launches.append((m, (ctypes.c_uint64 * m)(*rel[lo:hi]), (ctypes.c_int64 * m)(*self.sizes[lo:hi]),
                                  (ctypes.c_int64 * m)(*outs[lo:hi]), (ctypes.c_int32 * m)(*grids[lo:hi])))
        self._out_ptr = _ptr(out)
        self._fn = _native.lib().pfrl_philox_normal

    def run(self):
        gen = self.gen
        seed, base = gen.initial_seed(), gen.get_offset()
        stream = _stream()
        for m, rel, numel, outs, grids in self.launches:
            check(self._fn(seed, base, m, rel, numel, outs, grids, self._out_ptr, self.variant, stream),
                  "philox_normal")
        gen.set_offset(base + self.total_increment)
        return self.views


def randn_calls(sizes, device, out=None, variant=None):
    """One-shot form of :class:`RandnPlan`: returns the list of views into ``out``.  Eager only:
    the generator's (seed, offset) are read and advanced on the host."""
    return RandnPlan(sizes, device, out=out, variant=variant).run()


def philox_variant(device):
    """Which restatement of rocRAND's Box-Muller arithmetic (0: separate multiply and add, 1:
    contracted to an fma) reproduces ``torch.randn`` of THIS PyTorch build on this device bit for
    bit, generator offsets included -- or None if neither does (callers then keep their
    torch.randn calls).  Probed once per device; the generator is left as it was found."""
    device = torch.device(device)
    key = (device.index, torch.__version__)
    if key in _PHILOX_VARIANT:
        return _PHILOX_VARIANT[key]
    gen = _default_generator(device)
    state = gen.get_state()
    found = None
    sizes = [4160, 51, 563, 1, 70001]
    try:
        gen.manual_seed(0x5EED5)
        gen.set_offset(8)
        want = [torch.randn(k, device=device) for k in sizes]
        end = gen.get_offset()
        for v in (1, 0):
            gen.manual_seed(0x5EED5)
            gen.set_offset(8)
            got = randn_calls(sizes, device, variant=v)
            if gen.get_offset() == end and all(torch.equal(a, b) for a, b in zip(got, want)):
                found = v
                break
    finally:
        gen.set_state(state)
    _PHILOX_VARIANT[key] = found
    return found
