#!/usr/bin/env python
"""Per-layer, per-direction time of the csrc/qnet.hip tile programs at the batch sizes of the
PPO rollout / update (512, 16384) and of the DQN update (32), against the f32 MFMA peak and the
HBM time of the layer's unique bytes.  Runs ON THE GPU BOX:

    python tools/layer_bench.py [--batches 16384,512] [--iters 10] [--only fwd,dgrad,wgrad]

Every row: kernel time (hipEvents around `iters` back-to-back launches), the FLOPs the layer
needs (useful: taps outside the output are not counted), TFLOP/s and the fraction of the 155
TFLOP/s f32 MFMA peak; `hbm_us` = unique bytes of the layer / 6.29 TB/s (what the launch would
take if it were a pure copy of its operands).
"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pfrl_amd import _native   # noqa: E402
from pfrl_amd._native import check   # noqa: E402

PEAK_TF = 155.0
HBM_TBS = 6.29

# (name, C, Cout, R, ST, H) of train_ppo_ale.py:247-264 / atari_cnn.py:17-47; the linear layer is 1x1
LAYERS = [("conv1", 4, 32, 8, 4, 84), ("conv2", 32, 64, 4, 2, 20), ("conv3", 64, 64, 3, 1, 9),
          ("fc", 3136, 512, 1, 1, 1)]


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def time_us(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def wgrad_splits(M, Cout, K, cps=None, cap=None):
    from pfrl_amd.nn import mfma_trunk as mt

    old = mt._WGRAD_CPS, mt._WGRAD_MAX_SPLITS
    if cps is not None:
        mt._WGRAD_CPS, mt._WGRAD_MAX_SPLITS = cps, cap
    try:
        return mt._wgrad_splits(M, Cout, K)
    finally:
        mt._WGRAD_CPS, mt._WGRAD_MAX_SPLITS = old


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="16384,512,32")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only", default="fwd,dgrad,wgrad")
    ap.add_argument("--layers", default="conv1,conv2,conv3,fc")
    ap.add_argument("--splits", default="",
                    help="comma list of (chunks per split):(max splits) pairs for the weight-gradient "
                         "split rule, e.g. 16:1024,8:2048,32:1024 -- timed with the default programs")
    ap.add_argument("--sweep", action="store_true",
                    help="every tile program that fits the layer (PFRL_QNET_FWD / _DGRAD / _WGRAD), each "
                         "checked bit for bit against the round-3 program of the same layer")
    args = ap.parse_args()
    only = set(args.only.split(","))
    lib = _native.lib()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    print("%-6s %-16s %6s %10s %9s %8s %6s %9s" % ("layer", "dir", "B", "us", "GFLOP", "TF/s", "frac",
                                                   "hbm_us"))
    for B in [int(b) for b in args.batches.split(",")]:
        for name, C, Co, R, ST, H in LAYERS:
            if name not in args.layers.split(","):
                continue
            OH = (H - R) // ST + 1
            x = torch.randn(B, H, H, C, device=dev)
            w = torch.randn(Co, R, R, C, device=dev) * 0.05
            b = torch.randn(Co, device=dev)
            y = torch.empty(B, OH, OH, Co, device=dev)
            dy = torch.randn(B, OH, OH, Co, device=dev)
            dx = torch.empty_like(x)
            M = B * OH * OH
            K = R * R * C
            flop = 2.0 * M * K * Co
            bx, by, bw = x.numel() * 4, y.numel() * 4, w.numel() * 4
            rows = []

            def run_fwd():
                check(lib.pfrl_conv2d_nhwc_fwd(_p(x), _p(w), _p(b), _p(y), B, H, H, C, Co, R, R, ST, 1, 0, 1,
                                               _stream()), "fwd")

            def run_dgrad():
                # (LAYER_BENCH_NO_MASK=1: without the ReLU mask of the layer below -- what its 4 bytes per
                # element of extra reads cost)
                check(lib.pfrl_conv2d_nhwc_bwd_data(_p(dy), None, _p(w),
                                                    None if os.environ.get("LAYER_BENCH_NO_MASK") else _p(x),
                                                    _p(dx), B, H, H, C, Co, R, R, ST, 0, 0, _stream()), "dgrad")

            splits = wgrad_splits(M, Co, K)
            stride = w.numel() + Co
            part = torch.empty(splits * stride, device=dev)

            def run_wgrad():
                check(lib.pfrl_conv2d_nhwc_bwd_weight(_p(dy), None, _p(x), _p(part), _p(part[w.numel():]),
                                                      stride, stride, B, H, H, C, Co, R, R, ST, splits,
                                                      _stream()), "wgrad")

            wide = Co % 64 == 0
            plans = [("fwd", "PFRL_QNET_FWD", run_fwd, lambda: y, bx + by + bw,
                      ([2, 7] if wide else []) + [3, 8] if args.sweep else [None]),
                     ("dgrad", "PFRL_QNET_DGRAD", run_dgrad, lambda: dx, 2 * bx + by + bw,
                      ([0] if C % 64 == 0 else []) + [1] + ([6, 7, 9] if ST > 1 and 64 % C == 0 else []) + ([8] if ST == 1 and R > 1 and C % 64 == 0 else []) if args.sweep else [None]),
                     ("wgrad/%d" % splits, "PFRL_QNET_WGRAD", run_wgrad, lambda: part,
                      bx + by + splits * stride * 4,
                      [0] + ([2] if wide and K % 64 == 0 else []) + ([3] if wide and K % 128 == 0 else [])
                      + ([4] if K % 128 == 0 else []) + ([5] if K % 256 == 0 else [])
                      if args.sweep else [None])]
            for d, env, fn, out, nbytes, progs in plans:
                if d.split("/")[0] not in only or (d == "dgrad" and name == "conv1"):
                    continue
                ref = None
                for prog in progs:
                    if prog is None:
                        os.environ.pop(env, None)
                    else:
                        os.environ[env] = str(prog)
                    t = time_us(fn, args.iters)
                    tag = d if prog is None else "%s#%d" % (d, prog)
                    if args.sweep:
                        got = out().clone()
                        if ref is None:
                            ref = got
                        else:
                            diff = float((got - ref).abs().max())
                            tag += " ==" if torch.equal(got, ref) else " d=%.1e" % diff
                    rows.append((tag, t, flop, nbytes))
                os.environ.pop(env, None)
            if args.splits and "wgrad" in only:
                for pair in args.splits.split(","):
                    cps, cap = [int(v) for v in pair.split(":")]
                    sp = wgrad_splits(M, Co, K, cps, cap)
                    part2 = torch.empty(sp * stride, device=dev)

                    def run_w2():
                        check(lib.pfrl_conv2d_nhwc_bwd_weight(
                            _p(dy), None, _p(x), _p(part2), _p(part2[w.numel():]), stride, stride, B, H,
                            H, C, Co, R, R, ST, sp, _stream()), "wgrad")

                    rows.append(("wgrad/%d(%s)" % (sp, pair), time_us(run_w2, args.iters), flop,
                                 bx + by + sp * stride * 4))
                    del part2
            if name == "conv1" and B >= 256 and "fwd" in only:
                # the fp32 form of the direct kernel (DQN's acting / target passes) beside its tile program
                os.environ["PFRL_CONV1_DIRECT"] = "0"
                t_tiles = time_us(run_fwd, args.iters)
                os.environ.pop("PFRL_CONV1_DIRECT")
                tf = flop / t_tiles * 1e-6
                print("%-6s %-16s %6d %10.1f %9.2f %8.1f %6.3f %9.1f" % (
                    name, "fwd (f32, tiles)", B, t_tiles, flop * 1e-9, tf, tf / PEAK_TF, (bx + by + bw) / HBM_TBS * 1e-6))
            if name == "conv1" and B >= 64:
                # the forms PPO runs since round 5: the layer reads u8 NHWC4 pixels itself
                px = torch.randint(0, 256, (B, H, H, 4), dtype=torch.uint8, device=dev)

                def run_fwd_u8():
                    check(lib.pfrl_conv2d_u8nhwc4_fwd(_p(px), 255.0, _p(w), _p(b), _p(y), B, H, H, Co, R, R,
                                                      ST, 1, 0, _stream()), "fwd_u8")

                def run_wgrad_u8():
                    check(lib.pfrl_conv2d_u8nhwc4_bwd_weight(_p(dy), None, _p(px), 255.0, _p(part),
                                                             _p(part[w.numel():]), stride, stride, B, H, H,
                                                             Co, R, R, ST, splits, _stream()), "wgrad_u8")

                try:
                    if "fwd" in only:
                        rows.append(("fwd (u8 input)", time_us(run_fwd_u8, args.iters), flop, bx // 4 + by + bw))
                        # the implicit-GEMM tile program the direct kernel replaced (round 6)
                        os.environ["PFRL_CONV1_DIRECT"] = "0"
                        rows.append(("fwd (u8, tiles)", time_us(run_fwd_u8, args.iters), flop, bx // 4 + by + bw))
                        os.environ.pop("PFRL_CONV1_DIRECT")
                    if "wgrad" in only:
                        rows.append(("wgrad/%d (u8)" % splits, time_us(run_wgrad_u8, args.iters), flop,
                                     bx // 4 + by + splits * stride * 4))
                        os.environ["PFRL_CONV1_DIRECT"] = "0"
                        rows.append(("wgrad (u8,tiles)", time_us(run_wgrad_u8, args.iters), flop,
                                     bx // 4 + by + splits * stride * 4))
                        os.environ.pop("PFRL_CONV1_DIRECT")
                except RuntimeError as e:
                    print("# conv1 u8 forms at B = %d: %s" % (B, str(e)[:120]))
                del px
            for d, t, f, nbytes in rows:
                tf = f / t * 1e-6
                print("%-6s %-16s %6d %10.1f %9.2f %8.1f %6.3f %9.1f" % (
                    name, d, B, t, f * 1e-9, tf, tf / PEAK_TF, nbytes / HBM_TBS * 1e-6))
            del x, w, y, dy, dx
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
