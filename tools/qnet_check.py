#!/usr/bin/env python
"""Diagnostic for csrc/qnet.hip on the GPU box: per-kernel max error against stock
PyTorch (CPU fp32 / fp64) and per-kernel time next to the MIOpen / hipBLASLt route.

    python tools/qnet_check.py [--batches 32,256,2048] [--no-time]
"""
import argparse
import os
import sys
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pfrl_amd.nn import mfma_trunk as mt   # noqa: E402
import pfrl_amd   # noqa: E402


def err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    d = (a - b).abs()
    i = int(d.argmax())
    return "max|d|=%.3e (ref %.3e at flat %d) rel=%.3e" % (
        float(d.max()), float(b.flatten()[i]), i, float(d.max() / (b.abs().max() + 1e-30)))


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def graph_time(fn, n=50):
    """us per call when replayed from a HIP graph (no host launch cost)."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (n * 10) * 1e6


def make_model(dev):
    torch.manual_seed(0)
    m = pfrl_amd.nn.LargeAtariCNN()
    head = nn.Linear(512, 6)
    ref = nn.Sequential(m, head)
    import copy

    dut = copy.deepcopy(ref).to(dev).to(memory_format=torch.channels_last)
    mt.accelerate_heads(dut)
    return ref, dut


def check_layers(dev, B):
    print("== per-layer checks, B=%d" % B)
    torch.manual_seed(1)
    geoms = [(4, 32, 8, 4, 84), (32, 64, 4, 2, 20), (64, 64, 3, 1, 9)]
    for C, Co, R, ST, H in geoms:
        conv = nn.Conv2d(C, Co, R, stride=ST)
        x = torch.rand(B, C, H, H)
        y_ref = F.relu(conv(x))
        cg = nn.Conv2d(C, Co, R, stride=ST).to(dev)
        cg.load_state_dict(conv.state_dict())
        cg = cg.to(memory_format=torch.channels_last)
        xg = x.to(dev).contiguous(memory_format=torch.channels_last)
        sp = mt.ConvSpec(cg, H, H)
        y = mt.conv_fwd(xg, cg.weight, cg.bias, sp, B, relu=True, planar=False)
        print(" conv %dx%d/%d C%d->%d fwd NHWC : %s" % (R, R, ST, C, Co,
              err(y.permute(0, 3, 1, 2), y_ref)))
        y = mt.conv_fwd(xg, cg.weight, cg.bias, sp, B, relu=True, planar=True)
        print("   planar                     : %s" % err(y.view(B, Co, sp.OH, sp.OW), y_ref))
        # backward pieces against autograd on the CPU
        xr = x.clone().requires_grad_(True)
        yr = F.relu(conv(xr))
        gy = torch.randn_like(yr)
        gy_masked = gy * (yr > 0)
        yr.backward(gy)
        dy = gy_masked.permute(0, 2, 3, 1).contiguous().to(dev)     # NHWC, already masked
        lib = mt._native.lib()
        # wgrad
        nW = cg.weight.numel()
        M = B * sp.OH * sp.OW
        splits = mt._wgrad_splits(M, Co, R * R * C)
        stride = nW + Co
        part = torch.empty(splits * stride, device=dev)
        dw = torch.empty_like(cg.weight)
        db = torch.empty(Co, device=dev)
        mt.check(lib.pfrl_conv2d_nhwc_bwd_weight(mt._p(dy), None, mt._p(xg), mt._p(part), mt._p(part[nW:]),
                                                  stride, stride, B, H, H, C, Co, R, R, ST, splits,
                                                  mt._stream()), "wgrad")
        mt._reduce([(part, dw, None, stride, nW, splits, 4, 0), (part[nW:], db, None, stride, Co, splits, 4, 0)])
        print("   wgrad (splits %2d)          : %s" % (splits, err(dw, conv.weight.grad)))
        print("   bgrad                      : %s" % err(db, conv.bias.grad))
        if C % 16 == 0:
            aprev = torch.rand(B, H, H, C, device=dev) - 0.3
            dx = torch.empty(B, H, H, C, device=dev)
            mt.check(lib.pfrl_conv2d_nhwc_bwd_data(mt._p(dy), None, mt._p(cg.weight), mt._p(aprev), mt._p(dx),
                                                   B, H, H, C, Co, R, R, ST, 0, 0, mt._stream()), "dgrad")
            want = xr.grad.permute(0, 2, 3, 1) * (aprev.cpu() > 0)
            print("   dgrad (+mask)              : %s" % err(dx, want))
    # linear 3136 -> 512
    lin = nn.Linear(3136, 512)
    x = torch.rand(B, 3136)
    xr = x.clone().requires_grad_(True)
    yr = F.relu(lin(xr))
    lg = nn.Linear(3136, 512).to(dev)
    lg.load_state_dict(lin.state_dict())
    y = mt.linear_fwd(x.to(dev), lg.weight, lg.bias, relu=True)
    print(" linear 3136->512 fwd (splits %d): %s" % (mt._fwd_splits(B, 512, 3136), err(y, yr)))
    head = nn.Linear(512, 6)
    hg = nn.Linear(512, 6).to(dev)
    hg.load_state_dict(head.state_dict())
    h = torch.randn(B, 512)
    hr = h.clone().requires_grad_(True)
    q = head(hr)
    gq = torch.randn_like(q)
    q.backward(gq)
    hgx = h.to(dev).requires_grad_(True)
    qg = mt.small_linear(hgx, hg)
    qg.backward(gq.to(dev))
    print(" head 512->6 fwd: %s" % err(qg, q))
    print("   dx: %s" % err(hgx.grad, hr.grad))
    print("   dw: %s" % err(hg.weight.grad, head.weight.grad))
    print("   db: %s" % err(hg.bias.grad, head.bias.grad))


def check_trunk(dev, B):
    print("== whole trunk, B=%d" % B)
    ref, dut = make_model(dev)
    x = torch.rand(B, 4, 84, 84)
    xg = x.to(dev).contiguous(memory_format=torch.channels_last)
    q_ref = ref(x)
    q = dut(xg)
    print(" q            : %s" % err(q, q_ref))
    g = torch.randn_like(q_ref)
    q_ref.backward(g)
    q.backward(g.to(dev))
    for (n, p), (_, pr) in zip(dut.named_parameters(), ref.named_parameters()):
        print(" grad %-22s: %s" % (n, err(p.grad, pr.grad)))
        assert p.grad.stride() == p.stride(), n
    return ref, dut


def time_trunk(dev, batches):
    print("== timing (us per call, HIP-graph replay of 10 calls)")
    ref, dut = make_model(dev)
    torch.backends.cudnn.benchmark = True
    for B in batches:
        xg = torch.rand(B, 4, 84, 84, device=dev).contiguous(memory_format=torch.channels_last)
        trunk = dut[0]

        def fwd_native():
            with torch.no_grad():
                return dut(xg)

        def fwd_stock():
            with torch.no_grad():
                h = xg
                for i, layer in enumerate(trunk.layers):
                    h = pfrl_amd.nn.atari_cnn.conv_activation(layer, h, F.relu, planar_out=(i == 2))
                return F.linear(F.relu(F.linear(h.reshape(B, -1), trunk.output.weight, trunk.output.bias)),
                                dut[1].weight, dut[1].bias)

        def fb_native():
            for p in dut.parameters():
                p.grad = None
            dut(xg).sum().backward()

        def fb_stock():
            for p in dut.parameters():
                p.grad = None
            h = xg
            for i, layer in enumerate(trunk.layers):
                h = pfrl_amd.nn.atari_cnn.conv_activation(layer, h, F.relu, planar_out=(i == 2))
            q = F.linear(F.relu(F.linear(h.reshape(B, -1), trunk.output.weight, trunk.output.bias)),
                         dut[1].weight, dut[1].bias)
            q.sum().backward()

        print(" B=%5d  fwd native %8.1f  stock %8.1f" % (B, graph_time(fwd_native), graph_time(fwd_stock)))
        if B <= 512:
            print("          fwd+bwd native %8.1f  stock %8.1f" % (graph_time(fb_native), graph_time(fb_stock)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="32,256,2048")
    ap.add_argument("--no-time", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.backends.cuda.matmul.allow_tf32 = False
    for B in (32, 5):
        check_layers(dev, B)
    for B in (32, 7, 256):
        check_trunk(dev, B)
    if not args.no_time:
        time_trunk(dev, [int(b) for b in args.batches.split(",")])


if __name__ == "__main__":
    main()
