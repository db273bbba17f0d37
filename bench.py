#!/usr/bin/env python
"""Headline benchmark: env-steps/s of the batched DQN hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input: 256
Atari-shaped envs step once (84x84 uint8 frames generated on the device into
the HBM frame ring) -> batch_act (gather u8->f32, Q forward, eps-greedy) ->
batch_observe (256 appends, 64 updates at the reference schedule: sample 32,
fused batch_experiences gather, Huber loss, backward, centered RMSprop) ->
env.reset(not_end).  BASELINE.json config[1]: DQN, Nature CNN,
ReplayBuffer(10**6) prefilled to capacity, B=32, update_interval=4,
batch_accumulator='sum', fp32 network.  Nothing is skipped inside the timed
region.  N GPUs = N ranks, the metric's 256 envs sharded 256 / N per rank (strong
scaling, the default; --scaling weak keeps 256 per rank), per-GPU-local replay,
per update one flat RCCL all-reduce of the small gradients + an all-gather of the
hidden layer's batch matrices (pfrl_amd/distributed.py).

Prints ONE JSON line on rank 0 (see DESIGN.md section "Measurement").
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


from benchkit.baselines import (cpu_baseline, data_path_only, reference_baseline,  # noqa: E402,F401
                                reference_baseline_other, reference_baseline_ppo)
from benchkit.roofline import (HBM_PEAK_GBS, MFMA_F32_PEAK_TFLOPS, NATURE_CONV1_FLOPS,  # noqa: E402,F401
                               NATURE_FWD_FLOPS, PROFILE_ADV_STATS, PROFILE_BATCH_EXPERIENCES,
                               PROFILE_BATCH_STATES_U8, PROFILE_BATCH_STATES_U8_RAW, PROFILE_GAE_SCAN,
                               algorithmic_bytes_per_step, compute_roofline,
                               gather_sources_sha16, launches_per_update, mfma_per_launch,
                               step_flops_dqn, step_flops_ppo)
from benchkit.supervisor import (_WATCHDOG, DP_PLANS, _StallWatchdog, _tick,  # noqa: E402,F401
                                 also_in_own_process, supervise)
from benchkit.workloads import (build_agent, build_ppo, build_rainbow, build_sac,  # noqa: E402,F401
                                one_step, prefill, workload_description)


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--algo", choices=["dqn", "rainbow", "ppo", "sac"], default="dqn",
                   help="dqn = BASELINE configs[1] (headline); rainbow = configs[2]; ppo = configs[3]; "
                        "sac = configs[4]")
    p.add_argument("--host-env", action="store_true",
                   help="dqn only: frames are produced on the HOST (numpy) and ingested over PCIe "
                        "through the pinned staging ring -- the PCIe-inclusive rate of DESIGN.md, "
                        "never the headline value")
    p.add_argument("--steps", type=int, default=None,
                   help="timed batched env steps (default 200; 128 = one rollout + update for ppo)")
    p.add_argument("--warmup", type=int, default=None,
                   help="untimed steps (default 5; 128 = one full rollout + update for ppo)")
    p.add_argument("--num-envs", type=int, default=None, help="default 256 (512 for ppo)")
    p.add_argument("--capacity", type=int, default=10 ** 6)
    p.add_argument("--minibatch", type=int, default=32)
    p.add_argument("--update-interval", type=int, default=4)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--replay-start", type=int, default=None,
                   help="replay_start_size (default: the example scripts' 5e4 dqn / 2e4 rainbow / 1e4 sac)")
    p.add_argument("--frame-slots", type=int, default=None,
                   help="slots of the HBM frame ring (default: capacity + room for the windows of all envs)")
    p.add_argument("--slack", type=int, default=None,
                   help="spare rows of the entry / transition rings (default 65536)")
    p.add_argument("--prefill", type=int, default=None,
                   help="transitions to prefill (default: capacity, i.e. full buffer)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--allow-lib-override", action="store_true",
                   help="accept PFRL_AMD_LIB (an A/B build of the native library); recorded on the line")
    p.add_argument("--no-data-path-only", action="store_true",
                   help="dqn: skip the second measurement with a zero-FLOP q_function")
    p.add_argument("--ppo-reuse-next-values", type=int, default=0, choices=[0, 1],
                   help="ppo: 0 (default, the package's default) = the reference's value pass over "
                        "states AND next_states (pfrl/agents/ppo.py:110-142; rows the two passes "
                        "share are evaluated once where that is bit-identical); 1 = the opt-in "
                        "shortcut without that guarantee (PPO(reuse_next_values=True))")
    p.add_argument("--no-also", action="store_true",
                   help="dqn: do not append the PPO configs[3] measurement ('also') to the line")
    p.add_argument("--scaling", choices=["weak", "strong"], default=None,
                   help="N > 1: weak = --num-envs envs on EVERY GPU; strong = --num-envs envs "
                        "sharded N/G per GPU as BASELINE.json's metric and SURVEY.md 8(e) describe "
                        "(default: strong)")
    p.add_argument("--cpu-baseline-seconds", type=float, default=14.0,
                   help="end-to-end sample of the reference on this box's host cores")
    p.add_argument("--cpu-baseline-threads", type=int, default=16,
                   help="torch CPU threads for the baseline's network (capped by the host; at "
                        "B=32 the Nature CNN is slower with all 128 threads than with 16)")
    p.add_argument("--no-cudnn-benchmark", dest="cudnn_benchmark", action="store_false",
                   help="do not let MIOpen search conv algorithms")
    p.add_argument("--nchw", dest="channels_last", action="store_false",
                   help="keep the network in NCHW (default: channels_last)")
    p.add_argument("--blas", choices=["default", "rocblas", "tunable"], default=None,
                   help="GEMM back-end for the PyTorch side: torch default (hipBLASLt), rocBLAS, "
                        "or torch's TunableOp (times every rocBLAS / hipBLASLt solution once per "
                        "shape and keeps the fastest); default: tunable for dqn and sac "
                        "(measured +3 % and 2.5x), torch default for rainbow and ppo (no gain)")
    p.add_argument("--chunks", type=str, default=None,
                   help="dqn: env-range cut points of the step-fused path as fractions, e.g. "
                        "'0.125' (default) or '' for one range")
    p.add_argument("--priority-pow", choices=["device", "device_cr", "host_libm"], default="device",
                   help="rainbow: where (clip(err) + eps) ** alpha is evaluated.  device = one "
                        "launch, glibc's powf restated on the device (bit-exact priority trees); "
                        "device_cr = one launch, correctly rounded power (<= 1 ulp from NumPy's "
                        "powf in <1 %% of inputs; rounds 1-2); host_libm = this host's libm as "
                        "NumPy does (one D2H per update)")
    p.add_argument("--torch-optimizer", action="store_true",
                   help="stock torch.optim.RMSprop instead of the fused HIP step")
    args = p.parse_args()
    if args.scaling is None:
        args.scaling = "strong"
    if args.steps is None:
        args.steps = 128 if args.algo == "ppo" else 200
    if args.warmup is None:
        args.warmup = 128 if args.algo == "ppo" else 5
    if args.algo == "ppo":
        # large-batch convs: MIOpen's default choices are already good and the
        # exhaustive search costs minutes of GPU time for ~-8 % throughput here
        args.cudnn_benchmark = False
    if args.blas is None:
        args.blas = "tunable" if args.algo in ("dqn", "sac") else "default"
    if args.num_envs is None:
        args.num_envs = {"ppo": 512, "sac": 64}.get(args.algo, 256)
    if args.algo == "sac" and args.minibatch == 32:
        args.minibatch = 256
    return args


def assemble_result(args, world, N, elapsed, n_updates, t_fill, workload, roofline):
    """The ONE JSON line of the driver contract (cpu_baseline etc. are added by the caller)."""
    ms = elapsed / args.steps * 1e3
    if roofline is not None:
        # the step as a whole against the same roofline: the kernel fraction above is for
        # the dominant gather alone and must not be read as the end-to-end figure
        nvp = getattr(args, "_ppo_next_value_pass", None) or {}
        vp = 1.0 + (nvp.get("evaluated", nvp.get("of", 1)) / max(1, nvp.get("of", 1)))
        step_bytes = algorithmic_bytes_per_step(args.algo, N, args.minibatch, args.update_interval,
                                                vp)
        roofline["step_algorithmic_bytes"] = int(step_bytes)
        roofline["step_frac"] = round(step_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
        if args.algo == "dqn":
            # the step against the OTHER roofline: its arithmetic / the f32 MFMA peak (the data
            # path is HBM-bound, the step as a whole is bound by the B = 32 update chain)
            fl = step_flops_dqn(N, args.minibatch, args.update_interval)
            tf = fl / (ms * 1e-3) / 1e12
            roofline["mfma"] = {"step_flops": int(fl), "achieved": round(tf, 2),
                                "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4)}
            # what fp32 arithmetic + the reference's schedule allow at all: the N / update_interval
            # updates of a batched step are DEPENDENT (each reads the parameters the previous one
            # wrote, pfrl/agents/dqn.py:516-549, pfrl/replay_buffer.py:329-356), so even with
            # every launch at the f32 MFMA peak a step takes step_flops / peak
            per_update = (fl - N * NATURE_FWD_FLOPS) / max(1, N // args.update_interval)
            floor_us = per_update / (MFMA_F32_PEAK_TFLOPS * 1e12) * 1e6
            ceiling = N / (fl / (MFMA_F32_PEAK_TFLOPS * 1e12))
            roofline["mfma"].update({
                "flop_per_update": int(per_update), "floor_us_per_update": round(floor_us, 2),
                "parity_ceiling_env_steps_s": int(ceiling),
                "frac_of_parity_ceiling": round(N / (ms * 1e-3) / ceiling, 4),
                "ceiling_what": "env-steps/s if every launch of the step ran at the f32 MFMA peak under "
                                "the reference's schedule (B = %d, %d dependent updates per %d-env step, "
                                "fp32 as the parity contract demands); the north star's 1 M env-steps/s "
                                "is above it" % (args.minibatch, N // args.update_interval, N)})
        if args.algo == "ppo":
            # PPO is bound by the f32 MFMA trunk at update size (B = 16384), not by the gather the
            # HBM block above describes (3 % of the device time): rollout FLOPs / time / peak
            fl = step_flops_ppo(N, value_passes=vp)
            tf = fl / (ms * 1e-3) / 1e12
            roofline["mfma"] = {"step_flops": int(fl), "achieved": round(tf, 2),
                                "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4),
                                "value_passes": round(vp, 4), "next_value_pass": nvp or None,
                                "what": "acting + value pass(es) + 4 epochs x (forward + backward) of the "
                                        "rollout, per env step (FLOPs executed); per-layer fractions at B = 16384: "
                                        "profiles/r06_layer_final.txt (tools/layer_bench.py)"}
            roofline["captured_launches_not_timed"] = (
                "the 65536-frame minibatch gathers run inside the captured update graph "
                "(agents/ppo.py::_minibatch_step) and carry no event pair; the timed launches are the "
                "value pass's gathers of the same kernel")
    per = "per GPU" if args.scaling == "weak" else "sharded over the GPUs"
    return {
        "metric": "env-steps/sec whole node (%s %d envs %s)" % (args.algo.upper(), N if args.scaling == "weak" else N * world, per),
        "value": round(world * N * args.steps / elapsed, 1), "unit": "env-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 3), "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None, "dtype": "f32",
        "data": "synthetic (host frames over PCIe)" if args.host_env else "synthetic",
        "config": {
            "workload": workload,
            "global_envs": world * N, "envs_per_gpu": N, "updates_in_timed_region": n_updates,
            "parallelism": "env-sharded dp%d, per-GPU-local replay" % world,
            "prefill_s": round(t_fill, 1),
        },
        "roofline": roofline,
    }


def collective_microbench(agent, device, iters=50):
    """Per-update exchange of the data-parallel update, timed on its own (hipEvents around ``iters``
    back-to-back calls, every rank takes part): the flat bucket's all-reduce and the grouped
    all-gather of the large Linear layer's batch matrices at this agent's sizes.  The in-graph cost
    is part of ``update_us``; these are the collectives' stand-alone latencies."""
    red = getattr(agent, "grad_reducer", None)
    comm = getattr(red, "_comm", None)
    if red is None or not red.active():
        return None
    out = {}
    bucket = red.current_bucket() if red.current_bucket() is not None else red._flat
    try:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        if bucket is not None:
            buf = torch.zeros_like(bucket)
            for _ in range(3):
                red.reduce_flat(buf)
            ev[0].record()
            for _ in range(iters):
                red.reduce_flat(buf)
            ev[1].record()
            torch.cuda.synchronize()
            out["flat_all_reduce"] = round(ev[0].elapsed_time(ev[1]) * 1e3 / iters, 2)
            out["flat_bytes"] = int(buf.numel() * 4)
        from pfrl_amd import distributed

        B = getattr(agent, "minibatch_size", 32)
        if comm is not None and distributed.lowrank_pays(B, 512, 3136, comm.world):
            G = comm.world
            dy, x = torch.zeros(B * 512, device=device), torch.zeros(B * 3136, device=device)
            dya, xa = torch.zeros(G * B * 512, device=device), torch.zeros(G * B * 3136, device=device)
            for i in range(iters + 3):
                if i == 3:
                    ev[2].record()
                with comm.group():
                    comm.all_gather(dya, dy)
                    comm.all_gather(xa, x)
            ev[3].record()
            torch.cuda.synchronize()
            out["lowrank_all_gather_pair"] = round(ev[2].elapsed_time(ev[3]) * 1e3 / iters, 2)
            out["lowrank_bytes_per_rank"] = int((dy.numel() + x.numel()) * 4)
    except Exception as e:      # (evidence only: never costs the line)
        out["note"] = "not measured: %s" % (str(e)[:120],)
    return out


def run_workload(args, device, rank, world, result_extras=True):
    """Build, prefill, warm up and time one workload; returns the result dict (rank 0) or None."""
    from pfrl_amd import ops

    _tick("build %s" % args.algo)
    agent, env, rbuf = build_agent(args, device, rank)
    N = args.num_envs
    obss = env.reset()
    _tick("prefill %s" % args.algo)
    t_fill = time.perf_counter()
    if rbuf is not None:
        # every replay workload runs at its stated size: the buffer is FULL when the timed region
        # starts (configs[4]: 1M fp32 transitions = 3.1 GB of rows, far outside the 256 MiB
        # Infinity Cache; round 3 timed SAC on a 10 % fill)
        target = args.prefill if args.prefill is not None else args.capacity
        target = max(min(target, args.capacity),
                     getattr(agent, "replay_start_size", None)
                     or agent.replay_updater.replay_start_size)
        obss = prefill(agent, env, obss, N, target)
    torch.cuda.synchronize()
    t_fill = time.perf_counter() - t_fill

    def barrier():
        # (the process group is the control plane -- gloo by default, pfrl_amd/distributed.py -- so the
        # barrier is a host rendezvous: the device is drained before AND after it)
        torch.cuda.synchronize()
        if torch.distributed.is_initialized():
            torch.distributed.barrier()

    # One-time work of the first updating steps (HIP-graph capture of every minibatch
    # buffer, MIOpen solver search, TunableOp GEMM tuning) must not fall into the timed
    # region even if the caller asks for fewer than 3 warm-up steps; for PPO the same
    # holds for the first rollout + update.
    need = (128 if args.algo == "ppo" else 3) - args.warmup
    for _ in range(max(0, need)):
        _tick("warm-up %s" % args.algo)
        obss = one_step(agent, env, obss, N)
    for _ in range(args.warmup):
        _tick("warm-up %s" % args.algo)
        obss = one_step(agent, env, obss, N)
    # (everything allocated so far -- networks, graphs, staging rings -- moved out of the collector's
    # generations: a full collection over them landing in the timed region is a ~10 ms host pause,
    # seen once as a 34 ms instead of a 22.5 ms acting phase of the PPO leg.  The collector stays on.)
    import gc

    gc.collect()
    gc.freeze()

    def updates_done():
        for name in ("optim_t", "n_updates", "n_policy_updates"):
            if hasattr(agent, name):
                return getattr(agent, name)
        return 0

    optim_before = updates_done()
    ops.profile_collect(kind=None)      # drop launches timed by an earlier workload
    ops.profile_enable(True)
    graphed = getattr(agent, "_graphed", None)
    if graphed is not None and hasattr(graphed, "run_range"):
        graphed.time_ranges = []
    barrier()
    torch.cuda.synchronize()
    # (one event per step boundary: where the device time of the timed region goes, step by step)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(min(args.steps, 1024) + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        _tick("timed steps %s" % args.algo)
        obss = one_step(agent, env, obss, N)
        if i + 1 < len(marks):
            marks[i + 1].record()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    step_ms = [a.elapsed_time(b) for a, b in zip(marks[:-1], marks[1:])]
    _tick("after timed steps %s" % args.algo)
    ops.profile_enable(False)
    n_updates = updates_done() - optim_before

    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64)
        if torch.distributed.get_backend() == "nccl":
            tmax = tmax.to(device)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tmax.item())

    all_us, all_units, all_kinds = ops.profile_collect(kind=None)
    roofline = compute_roofline(args.algo, all_us, all_units, all_kinds)
    args._ppo_next_value_pass = getattr(agent, "next_value_pass", None)
    out = assemble_result(args, world, N, elapsed, n_updates, t_fill,
                          workload_description(args, N, rbuf), roofline)
    if graphed is not None and getattr(graphed, "time_ranges", None):
        timed, graphed.time_ranges = graphed.time_ranges, None
        big_u = max(u for u, _, _ in timed)
        us = [e0.elapsed_time(e1) * 1e3 / u for u, e0, e1 in timed if u == big_u]
        if roofline is not None and us:
            roofline.setdefault("mfma", {})["update_us"] = round(sum(us) / len(us), 2)
            roofline["mfma"]["update_us_what"] = (
                "device time of ONE optimizer update (forward, TD loss, backward, optimizer step) "
                "inside the captured %d-update range graph, hipEvents around the replay" % big_u)
            try:
                if args.algo != "dqn" or torch.distributed.is_initialized():
                    raise RuntimeError("measured for the single-process DQN update only")
                pl = mfma_per_launch(agent, rbuf, args.minibatch)
                roofline["mfma"]["per_launch"] = pl
                roofline["mfma"]["launches_per_update"] = {"value": pl["n_launches"], "source": pl["source"]}
            except Exception as e:      # (the measurement must not cost the line)
                roofline["mfma"]["per_launch"] = {"note": "not available: %s" % e}
                roofline["mfma"]["launches_per_update"] = launches_per_update()
    out["config"]["ranks_seen"] = world
    if step_ms:
        srt = sorted(step_ms)
        split = {"median_step_ms": round(srt[len(srt) // 2], 4), "max_step_ms": round(srt[-1], 4),
                 "what": "device time between step boundaries (hipEvents), timed region"}
        if args.algo == "ppo" and len(srt) >= 8:
            # a rollout = (steps - 1) acting steps + the step that also runs the update
            n_up = max(1, args.steps * N // (N * 128))
            split["act_phase_ms"] = round(sum(srt[:-n_up]), 3)
            split["update_phase_ms"] = round(sum(srt[-n_up:]) - n_up * srt[len(srt) // 2], 3)
        out["config"]["phase_split"] = split
    if torch.distributed.is_initialized():
        from pfrl_amd import rccl

        out["config"].update(rccl.status())
        out["config"]["collective_us"] = collective_microbench(agent, device)
    if args.algo == "rainbow":
        out["config"]["priority_pow"] = rbuf.priority_pow
    if result_extras and args.algo == "dqn" and not args.host_env and not args.no_data_path_only:
        # every rank runs it (the agents are symmetric); rank 0 reports its own
        dpo, obss = data_path_only(args, device, agent, env, rbuf, obss, min(args.steps, 100))
        out["data_path_only"] = dpo
    del agent, env, rbuf
    torch.cuda.empty_cache()
    return out if rank == 0 else None


def rank_shape_legs(args, out, keys):
    """What ONE rank of the 8-GPU job runs, measured on this GPU under a single-rank process group
    (PFRL_DIST_ALWAYS=1: RCCL communicator, data-parallel form of the update with its collectives
    as single-rank launches, the control-plane exchanges): 256 / 8 = 32 envs for DQN, 512 / 8 = 64
    envs x 128 steps with minibatch 2 048 for PPO.  ``projected_speedup_g8`` = 8 x the rank-shaped
    value / the 1-GPU value of the same line: what env sharding gives BEFORE any byte crosses a
    link (the collective's wire time is not in it; DESIGN.md section 6 prices that).  No hardware
    scaling curve is claimed here: the driver measures that itself."""
    dp_env = {"PFRL_DIST_ALWAYS": "1", "PFRL_FORCE_SPLIT_GRAPH": "1", "PFRL_DP_LOWRANK": "force",
              "MASTER_ADDR": "127.0.0.1"}
    legs = (("dqn_rank_shape_g8", out.get("value"),
             ["--algo", "dqn", "--num-envs", "32", "--steps", "160", "--warmup", "40", "--scaling", "weak"]),
            ("ppo_rank_shape_g8", (out["also"].get("ppo") or {}).get("value"),
             ["--algo", "ppo", "--num-envs", "64", "--steps", "128", "--warmup", "128", "--scaling", "weak"]))
    import socket

    for k, (name, full, argv) in enumerate(legs):
        with socket.socket() as sock:       # (a free rendezvous port for the leg's single-rank group)
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        env = dict(dp_env, MASTER_PORT=str(port))
        r = also_in_own_process(args, argv, limit_s=300, extra_env=env)
        if r is None:
            out["also"][name] = {"value": None, "note": "the rank-shaped run did not complete"}
            continue
        leg = {k_: r[k_] for k_ in keys if k_ in r}
        leg["metric"] = "env-steps/sec of ONE rank of the 8-GPU job (its env shard, data-parallel update)"
        if full:
            leg["projected_speedup_g8"] = round(8.0 * r["value"] / full, 3)
            leg["projection_what"] = ("8 x this value / the 1-GPU value of the full workload on this "
                                      "line; collectives run as single-rank launches, wire time "
                                      "excluded (DESIGN.md section 6)")
        out["also"][name] = leg


def main():
    args = parse_args()
    if (int(os.environ.get("WORLD_SIZE", "1")) > 1 and os.environ.get("PFRL_BENCH_CHILD") != "1"
            and os.environ.get("PFRL_BENCH_SUPERVISE", "1") != "0"):
        return supervise(args)
    # stdout carries exactly ONE line, the JSON result.  Native libraries (RCCL's
    # version banner, MIOpen notes) write to fd 1 through C stdio at times of their
    # own choosing: park fd 1 on stderr for the whole run and keep the real stdout
    # for the result line.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    from pfrl_amd import _native
    from pfrl_amd.distributed import init_process_group_from_env

    rank, world, local = init_process_group_from_env()
    assert torch.cuda.is_available(), "bench.py needs the MI355X"
    if world != args.gpus:
        assert world == 1 and args.gpus == 1, "--gpus must equal WORLD_SIZE (use torchrun for N>1)"
    device = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(device)
    # the line is about the library the parity tests validated: the in-tree build, nothing else
    assert args.allow_lib_override or not os.environ.get("PFRL_AMD_LIB"), (
        "PFRL_AMD_LIB is set: bench.py measures pfrl_amd/lib/libpfrl_amd.so only (A/B builds of "
        "tools/build_variant.sh: --allow-lib-override, and the line says so)")
    _native.lib()
    if world > 1:
        # one process per GPU: keep each rank's host-side torch ops on its share of cores
        torch.set_num_threads(max(1, (os.cpu_count() or world) // world))
    if args.scaling == "strong":
        # the metric's envs are sharded, GPU g owns envs [g N/G, (g+1) N/G)
        assert args.num_envs % world == 0, "--num-envs must divide by the number of GPUs"
        args.total_envs = args.num_envs
        args.num_envs //= world

    if args.blas == "rocblas":
        torch.backends.cuda.preferred_blas_library("cublas")
    elif args.blas == "tunable":
        torch.cuda.tunable.enable(True)
        torch.cuda.tunable.tuning_enable(True)
        torch.cuda.tunable.set_max_tuning_duration(10)
        torch.cuda.tunable.set_filename("/tmp/pfrl_tunableop_rank%d.csv" % rank)

    if world > 1:
        _WATCHDOG[0] = _StallWatchdog(args, rank, world, result_fd,
                                      float(os.environ.get("PFRL_BENCH_STALL_S", "300")))
    run_guarded = run_workload
    if torch.distributed.is_initialized():
        from pfrl_amd import rccl

        def run_guarded(*a, **kw):
            """The data plane must not cost the line: an RCCL error code raised on this rank (the
            same call fails on its peers) retires the direct data plane -- the ranks confirm it to
            each other over the control plane -- and the workload is built and run again with the
            gradients carried by the process group (config.dp_plan = "fallback:...")."""
            try:
                return run_workload(*a, **kw)
            except rccl.DataPlaneError as e:
                sys.stderr.write("bench.py: %s -- retrying on the process-group path\n" % (e,))
                why = str(e)[:160]
            torch.distributed.monitored_barrier(timeout=__import__("datetime").timedelta(seconds=120)) \
                if torch.distributed.get_backend() == "gloo" else torch.distributed.barrier()
            rccl.retire(why)
            return run_workload(*a, **kw)

    out = run_guarded(args, device, rank, world)
    if args.algo == "dqn" and not args.host_env and not args.no_also:
        # the other half of BASELINE.json's metric ("DQN 256 envs, PPO 512 envs"): the PPO
        # configs[3] workload, timed by the same process right after
        import copy

        pargs = copy.copy(args)
        pargs.algo, pargs.steps, pargs.warmup = "ppo", 128, 128
        pargs.num_envs = 512 if args.scaling == "weak" else 512 // world
        pargs.cudnn_benchmark = False
        torch.backends.cudnn.benchmark = False
        keys = ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "scaling", "config",
                "roofline")
        own = world == 1 and os.environ.get("PFRL_BENCH_ALSO_IN_PROCESS") != "1"
        torch.cuda.empty_cache()
        also = also_in_own_process(args, ["--algo", "ppo", "--steps", "128", "--warmup", "128",
                                          "--num-envs", "512"]) if own else None
        if also is None:
            also = run_guarded(pargs, device, rank, world, result_extras=False)
        if rank == 0:
            out["also"] = {"ppo": {k: also[k] for k in keys}}
        if world == 1:
            # (also.ppo IS the reference's value pass since round 6 -- PPO(reuse_next_values=False)
            # is the package default; round 5's separate also.ppo_reference_semantics leg is gone)
            # configs[2] and configs[4], short: every GPU config of BASELINE.json on one line
            for algo, n_envs, mb, blas in (("rainbow", 256, 32, "default"), ("sac", 64, 256, "tunable")):
                # (warm-up: every double-buffered minibatch set captures its graphs)
                r2 = also_in_own_process(args, ["--algo", algo, "--steps", "50", "--warmup", "20",
                                                "--num-envs", str(n_envs), "--minibatch", str(mb),
                                                "--blas", blas]) if own else None
                if r2 is None:
                    a2 = copy.copy(args)
                    a2.algo, a2.steps, a2.warmup = algo, 50, 20
                    a2.num_envs, a2.minibatch, a2.blas = n_envs, mb, blas
                    a2.cudnn_benchmark = True
                    torch.backends.cudnn.benchmark = True
                    if blas == "tunable":
                        torch.cuda.tunable.enable(True)
                        torch.cuda.tunable.tuning_enable(True)
                    else:
                        torch.cuda.tunable.enable(False)
                    r2 = run_workload(a2, device, rank, world, result_extras=False)
                if rank == 0:
                    out["also"][algo] = {k: r2[k] for k in keys}
            if own and rank == 0:
                rank_shape_legs(args, out, keys)
    if rank == 0:
        if not args.no_cpu_baseline and world == 1 and args.algo == "dqn":
            port = cpu_baseline(args, min(args.cpu_baseline_seconds, 8.0))
            ref = reference_baseline(args)
            if ref is not None:
                ref["port"] = port       # the C oracle + torch-CPU port, as rounds 1-2 reported
                out["cpu_baseline"] = ref
            else:
                out["cpu_baseline"] = port
            if "also" in out and "ppo" in out["also"]:
                # the PPO half of the metric gets its own reference baseline
                pref = reference_baseline_ppo(args)
                if pref is not None:
                    out["also"]["ppo"]["cpu_baseline"] = pref
            for algo, n_envs in (("rainbow", 256), ("sac", 64)):
                if algo in out.get("also", {}):
                    oref = reference_baseline_other(args, algo, n_envs)
                    if oref is not None:
                        out["also"][algo]["cpu_baseline"] = oref
        if not args.no_cpu_baseline and world == 1 and args.algo in ("rainbow", "sac"):
            oref = reference_baseline_other(args, args.algo, args.num_envs)
            if oref is not None:
                out["cpu_baseline"] = oref
        if not args.no_cpu_baseline and world == 1 and args.algo == "ppo":
            pref = reference_baseline_ppo(args, num_envs=args.num_envs)
            if pref is not None:
                out["cpu_baseline"] = pref
    if rank == 0 and out is not None:
        import hashlib

        lib_path = os.environ.get("PFRL_AMD_LIB") or _native.LIB_PATH
        with open(lib_path, "rb") as f:
            out["config"]["native_lib"] = {
                "path": os.path.relpath(lib_path, os.path.dirname(os.path.abspath(__file__))),
                "sha256_16": hashlib.sha256(f.read()).hexdigest()[:16],
                "override": bool(os.environ.get("PFRL_AMD_LIB"))}
    # the line first: teardown of a communicator must not be able to cost it
    sys.stdout.flush()
    if rank == 0:
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if _WATCHDOG[0] is not None:
        _WATCHDOG[0].printed = True
    _tick("teardown")
    if torch.distributed.is_initialized() and os.environ.get("PFRL_BENCH_SOFT_EXIT") != "1":
        # (PFRL_BENCH_SOFT_EXIT=1: a profiler's exit handlers must run -- rocprofv3 writes its
        # trace at exit -- single-rank groups only)
        # The communicator is NOT destroyed: ncclCommDestroy waits for every captured graph that
        # holds one of its collectives to be released first (round 5, two live ranks: the check
        # tool sat in it until its timeout with a graph still referenced), and nothing here needs
        # an orderly RCCL shutdown -- the device is drained, the ranks meet at a barrier, and the
        # process leaves without running finalizers.
        torch.cuda.synchronize()
        try:
            torch.distributed.barrier()
        except Exception:       # noqa: BLE001 -- a peer that already left must not cost the status
            pass
        os.close(result_fd)
        sys.stderr.flush()
        os._exit(0)
    if _WATCHDOG[0] is not None:
        _WATCHDOG[0].done = True
    os.close(result_fd)


if __name__ == "__main__":
    main()
