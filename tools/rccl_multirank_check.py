"""Functional check of the data plane with MORE THAN ONE rank (pfrl_amd/rccl.py + distributed.py).

Run on a box that shows >= 2 HIP devices (an 8-GPU node, or ONE MI355X put into CPX compute
partitioning -- 8 XCD partitions, each its own device), or with --shared-device on a one-GPU box
(the lease of round 5 refused CPX: profiles/r05_cpx_refused.txt):

    python tools/rccl_multirank_check.py --world 2 [--shared-device] --out gpurun_out/r05/rccl_multirank.json

Every rank: unique-id broadcast over the gloo control plane, ncclCommInitRank(nranks = world),
eager all-reduce / all-gather / grouped all-gather, the capture probe, ONE captured graph holding
an all-reduce on the main stream and a grouped all-gather forked to the side stream, replayed
100 times on changing data, then the low-rank exchange of a large Linear layer's gradient against
the flat all-reduce plan and against one process on the concatenated batch (the assertions of
tests/test_distributed.py::_lowrank_worker, with the parameters on the device).  A functional test
only: partitions of one GPU say nothing about xGMI bandwidth.
"""
import argparse
import copy
import json
import os
import socket
import sys
import time

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def worker(rank, world, port, out_dir, shared=False, pattern="D"):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if shared:
        os.environ["PFRL_RCCL_SHARED_DEVICE"] = "1"     # (rccl.py: one NCCL_HOSTID per rank)
    import faulthandler

    import torch.distributed as dist

    from pfrl_amd import distributed, rccl

    faulthandler.enable()
    res = {"rank": rank, "world": world, "steps": []}

    def step(name, t0):
        torch.cuda.synchronize()
        res["steps"].append({"name": name, "s": round(time.time() - t0, 3)})
        sys.stderr.write("[rank %d] %s ok (%.2f s)\n" % (rank, name, time.time() - t0))
        sys.stderr.flush()

    distributed.init_process_group_from_env()            # gloo control plane (default)
    assert dist.get_backend() == "gloo"
    dev = torch.device("cuda", rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    t0 = time.time()
    comm = rccl.default_comm(dev)
    assert comm is not None and comm.world == world and comm.rank == rank
    step("comm_init", t0)

    # eager collectives
    t0 = time.time()
    a = torch.full((1 << 16,), float(rank + 1), device=dev)
    comm.all_reduce(a, average=False)
    assert bool((a == world * (world + 1) / 2).all()), "all_reduce sum"
    a = torch.arange(1000, dtype=torch.float32, device=dev) * (rank + 1)
    comm.all_reduce(a, average=True)
    want = torch.arange(1000, dtype=torch.float32, device=dev) * ((world + 1) / 2)
    assert torch.allclose(a, want, rtol=1e-6), "all_reduce avg"
    inp = torch.full((257,), float(rank), device=dev)
    out = torch.empty(world * 257, device=dev)
    comm.all_gather(out, inp)
    assert torch.equal(out.view(world, 257)[:, 0].cpu(), torch.arange(world, dtype=torch.float32))
    o1, o2 = torch.empty(world * 257, device=dev), torch.empty(world * 64, device=dev)
    i2 = torch.full((64,), float(10 + rank), device=dev)
    with comm.group():
        comm.all_gather(o1, inp)
        comm.all_gather(o2, i2)
    torch.cuda.synchronize()
    assert torch.equal(o1, out) and torch.equal(
        o2.view(world, 64)[:, 3].cpu(), torch.arange(world, dtype=torch.float32) + 10)
    step("eager_collectives", t0)

    # the probe every captured plan is gated by
    t0 = time.time()
    res["captured_collectives_work"] = bool(distributed.captured_collectives_work(dev))
    step("capture_probe", t0)

    # one graph holding collectives, in the shape `pattern` names:
    #   A  all-reduce on the capture stream
    #   B  A + a grouped pair of all-gathers, all on the capture stream
    #   C  fork: two separate all-gathers on the side stream, all-reduce on the capture stream, join
    #   D  fork: GROUPED all-gathers on the side stream, all-reduce on the capture stream, join
    #   E  fork: grouped all-gathers on the side stream only, join
    #   F  fork: grouped all-gathers AND the all-reduce on the side stream, join
    if res["captured_collectives_work"] and pattern != "none":
        t0 = time.time()
        x = torch.zeros(4096, device=dev)
        red = torch.zeros(4096, device=dev)
        gi = torch.zeros(512, device=dev)
        go1, go2 = torch.zeros(world * 512, device=dev), torch.zeros(world * 512, device=dev)
        s = torch.cuda.Stream(dev)
        s.wait_stream(torch.cuda.current_stream(dev))

        def gathers(stream, grouped):
            if grouped:
                with comm.group():
                    comm.all_gather(go1, gi, stream=stream)
                    comm.all_gather(go2, gi, stream=stream)
            else:
                comm.all_gather(go1, gi, stream=stream)
                comm.all_gather(go2, gi, stream=stream)

        with torch.cuda.stream(s):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
                x.add_(float(rank + 1))
                gi.copy_(x[:512])
                red.copy_(x)
                if pattern == "A":
                    comm.all_reduce(red, average=False)
                elif pattern == "B":
                    comm.all_reduce(red, average=False)
                    gathers(None, True)
                else:
                    comm.side.wait_stream(torch.cuda.current_stream(dev))
                    with torch.cuda.stream(comm.side):
                        gathers(comm.side, pattern in ("D", "E", "F"))
                        if pattern == "F":
                            comm.all_reduce(red, average=False, stream=comm.side)
                    if pattern in ("C", "D"):
                        comm.all_reduce(red, average=False)
                    torch.cuda.current_stream(dev).wait_stream(comm.side)
            sys.stderr.write("[rank %d] pattern %s captured\n" % (rank, pattern))
            sys.stderr.flush()
            for _ in range(100):
                g.replay()
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize()
        if pattern != "E":
            assert bool((red == 100.0 * world * (world + 1) / 2).all()), "captured all_reduce"
        if pattern != "A":
            want = (torch.arange(world, dtype=torch.float32) + 1) * 100.0
            assert torch.equal(go1.view(world, 512)[:, 0].cpu(), want), "captured all_gather"
            assert torch.equal(go1, go2)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s):
            ev0.record(s)
            for _ in range(200):
                g.replay()
            ev1.record(s)
        torch.cuda.synchronize()
        res["captured_graph_pattern"] = pattern
        res["captured_graph_us_per_replay"] = round(ev0.elapsed_time(ev1) * 1e3 / 200, 2)
        step("captured_graph_%s_100_replays" % pattern, t0)

    # low-rank exchange == flat all-reduce == one process on the concatenated batch
    t0 = time.time()
    torch.manual_seed(5)
    base = torch.nn.Sequential(torch.nn.Linear(16, 256), torch.nn.ReLU(), torch.nn.Linear(256, 512),
                               torch.nn.ReLU(), torch.nn.Linear(512, 3)).to(dev)
    M = 4
    assert distributed.lowrank_pays(M, 512, 256, world)
    nets = {k: copy.deepcopy(base) for k in ("lowrank", "flat", "one")}
    red = {"lowrank": distributed.GradientAllReducer(nets["lowrank"], early_bytes=100_000),
           "flat": distributed.GradientAllReducer(nets["flat"], early_bytes=0)}
    assert red["lowrank"]._comm is comm and len(red["lowrank"]._lowrank_modules) == 1
    opts = {k: torch.optim.RMSprop(n.parameters(), lr=1e-3, alpha=0.95, eps=1e-2, centered=True)
            for k, n in nets.items()}
    taken = []
    orig = red["lowrank"].lowrank_ready
    red["lowrank"].lowrank_ready = lambda *a: (taken.append(orig(*a)), taken[-1])[1]
    for it in range(20):
        gen = torch.Generator().manual_seed(1000 + it)
        xs = torch.randn(world * M, 16, generator=gen).to(dev)
        ys = torch.randn(world * M, 3, generator=gen).to(dev)
        mine = slice(rank * M, (rank + 1) * M)
        for k in ("lowrank", "flat"):
            opts[k].zero_grad(set_to_none=True)
            torch.nn.functional.mse_loss(nets[k](xs[mine]), ys[mine], reduction="sum").backward()
            red[k].all_reduce()
            opts[k].step()
        opts["one"].zero_grad(set_to_none=True)
        (torch.nn.functional.mse_loss(nets["one"](xs), ys, reduction="sum") / world).backward()
        opts["one"].step()
    torch.cuda.synchronize()
    assert taken == [True] * 20
    flat = {k: torch.cat([p.detach().reshape(-1) for p in n.parameters()]).cpu() for k, n in nets.items()}
    scale = float(flat["one"].abs().max())
    res["lowrank_vs_flat"] = float((flat["lowrank"] - flat["flat"]).abs().max()) / scale
    res["lowrank_vs_one_process"] = float((flat["lowrank"] - flat["one"]).abs().max()) / scale
    assert res["lowrank_vs_flat"] <= 2e-6 and res["lowrank_vs_one_process"] <= 2e-6
    np.save(os.path.join(out_dir, "lowrank%d.npy" % rank), flat["lowrank"].numpy())
    step("lowrank_equals_flat_equals_one_process", t0)

    res["ok"] = True
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as f:
        json.dump(res, f)
    dist.barrier()
    torch.cuda.synchronize()
    # (no ncclCommDestroy: it waits for every captured graph holding a collective of the
    # communicator to be released; the process simply leaves)
    sys.stderr.flush()
    os._exit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=2)
    ap.add_argument("--out", default=None)
    ap.add_argument("--graph-pattern", default="D", choices=["A", "B", "C", "D", "E", "F", "none"],
                    help="shape of the captured graph (see worker()); D = the data-parallel update's")
    ap.add_argument("--shared-device", action="store_true",
                    help="all ranks on ONE device: every rank poses as its own host (NCCL_HOSTID), so "
                         "RCCL's duplicate-GPU check passes and the ranks talk over its socket "
                         "transport on loopback -- real multi-rank communicators, graphs and "
                         "exchanges on a one-GPU box")
    args = ap.parse_args()
    import tempfile

    n_dev = torch.cuda.device_count()
    summary = {"devices_visible": n_dev, "world": args.world, "shared_device": args.shared_device,
               "graph_pattern": args.graph_pattern}
    if n_dev < args.world and not args.shared_device:
        summary["ok"] = False
        summary["reason"] = "only %d HIP device(s) visible" % n_dev
    else:
        d = tempfile.mkdtemp()
        try:
            mp.spawn(worker, args=(args.world, _free_port(), d, args.shared_device, args.graph_pattern), nprocs=args.world, join=True)
            ranks = [json.load(open(os.path.join(d, "rank%d.json" % r))) for r in range(args.world)]
            p = [np.load(os.path.join(d, "lowrank%d.npy" % r)) for r in range(args.world)]
            summary["replicas_identical"] = bool(all(np.array_equal(p[0], q) for q in p[1:]))
            summary["ranks"] = ranks
            summary["ok"] = all(r.get("ok") for r in ranks) and summary["replicas_identical"]
        except Exception as e:      # report, do not hide
            summary["ok"] = False
            summary["reason"] = repr(e)[:2000]
    text = json.dumps(summary, indent=1)
    print(text)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            f.write(text + "\n")
    sys.exit(0 if summary.get("ok") else 1)


if __name__ == "__main__":
    main()
