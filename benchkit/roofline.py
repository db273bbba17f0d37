"""Algorithmic bytes / FLOPs of a step and the ``roofline`` object of the line (bench.py)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec

PROFILE_BATCH_EXPERIENCES, PROFILE_BATCH_STATES_U8, PROFILE_GAE_SCAN, PROFILE_ADV_STATS = 0, 1, 2, 3   # pfrl_amd.ops constants
PROFILE_BATCH_STATES_U8_RAW = 4


def compute_roofline(algo, all_us, all_units, all_kinds):
    """``roofline`` object for the dominant HIP kernel of the path: the fused
    batch_experiences gather for the replay agents, the batch_states gather (value
    pass + minibatches) for PPO.  Inputs: per-launch durations (us), unit counts
    and kinds as returned by ``ops.profile_collect(kind=None)``."""
    k, fb = 4, 84 * 84
    if algo == "ppo":
        kind, kname, unit_name = PROFILE_BATCH_STATES_U8, "k_batch_states_u8", "frames"
        # per gathered frame: fb bytes read as u8, 4*fb written as f32
        per_unit = fb + 4 * fb
        if PROFILE_BATCH_STATES_U8_RAW in all_kinds:
            # round 5: the network reads u8 NHWC4 pixels (phi in the first convolution's operand
            # loader, agents/ppo.py _u8_pixels), so the gather writes one byte per frame byte: the
            # path's gather IS this kernel, priced at what it has to move (2 bytes per frame byte;
            # SURVEY 8d's 5 bytes assume the fp32 copy that no longer exists)
            kind, kname, per_unit = PROFILE_BATCH_STATES_U8_RAW, "k_batch_states_u8_raw", fb + fb
    elif algo == "sac":
        kind, kname, unit_name = PROFILE_BATCH_EXPERIENCES, "k_batch_experiences", "entries"
        # per sampled entry: state + next_state f32[376] read and written, action f32[17]
        # read and written, reward/terminal/discount
        per_unit = 2 * (2 * 376 * 4) + 2 * 17 * 4 + 2 * 12
    else:
        kind, kname, unit_name = PROFILE_BATCH_EXPERIENCES, "k_batch_experiences", "entries"
        # per sampled entry: state + next_state, each k frames read as u8, written as f32
        per_unit = 2 * k * (fb + 4 * fb)
    k_us = [u for u, kd in zip(all_us, all_kinds) if kd == kind]
    k_units = [n for n, kd in zip(all_units, all_kinds) if kd == kind]
    if not k_us:
        return None
    scan = {}
    for skind, sname, sbytes, swhat in (
            (PROFILE_GAE_SCAN, "k_gae_scan_lds", 8 + 4 + 4 + 1 + 1 + 4 + 4,
             "per (t, env): reward f64 + v + next_v f32 + nonterminal + cut u8 read, adv + v_teacher "
             "f32 written"),
            (PROFILE_ADV_STATS, "k_adv_partial", 4, "per advantage: one f32 read")):
        s_us = [u for u, kd in zip(all_us, all_kinds) if kd == skind]
        s_units = [n for n, kd in zip(all_units, all_kinds) if kd == skind]
        if s_us:
            gbs = sbytes * sum(s_units) / (sum(s_us) * 1e-6) / 1e9
            scan[sname] = {"achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(gbs / HBM_PEAK_GBS, 5), "launches_timed": len(s_us),
                           "avg_launch_us": round(sum(s_us) / len(s_us), 2),
                           "elements_per_launch": int(s_units[0]), "bytes_per_element": sbytes,
                           "what": swhat}
    # The kernel is launched in a few shapes (DQN: a small and a large env range per
    # step; PPO: acting, value pass and minibatch gathers).  The roofline object
    # describes the shape that moves the most bytes; the aggregate over every timed
    # launch of the kernel is reported next to it.
    classes = {}
    for u, n in zip(k_us, k_units):
        c = classes.setdefault(n, [0, 0.0])
        c[0] += 1
        c[1] += u
    main_units = max(classes, key=lambda n: n * classes[n][0])
    n_main, us_main = classes[main_units]
    bytes_main = per_unit * main_units
    achieved = bytes_main * n_main / (us_main * 1e-6) / 1e9
    tot_bytes = sum(per_unit * b for b in k_units)
    tot_s = sum(k_us) * 1e-6
    roofline = {
        "bound": "hbm", "kernel": kname,
        "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
        "bytes_per_launch": int(bytes_main),
        # (the pricing convention, explicit -- ADVICE r5: what one unit is charged, in bytes)
        "priced_bytes_per_unit": int(per_unit), "unit_name": unit_name,
        "%s_per_launch" % unit_name: int(main_units),
        "avg_launch_us": round(us_main / n_main, 2), "launches_timed": n_main,
        "share_of_kernel_bytes": round(bytes_main * n_main / tot_bytes, 4),
        "all_launches": {
            "achieved": round(tot_bytes / tot_s / 1e9, 1), "launches": len(k_us),
            "shapes": {str(n): {"launches": c[0], "avg_launch_us": round(c[1] / c[0], 2)}
                       for n, c in sorted(classes.items())}},
        "timing": "hipEvent pair attached to each dispatch (hipExtLaunchKernelGGL) on the "
                  "launch stream, inside the timed region",
    }
    if roofline["frac"] > 1.0:
        # (algorithmic bytes count every observation's k stacked frames as read; a launch whose
        # frames AND output fit the 256 MiB Infinity Cache -- a 64-env rank's rollout -- is served
        # from there and can exceed the HBM peak: say so instead of leaving a fraction above one)
        roofline["note"] = ("above the HBM peak: the launch's working set is Infinity-Cache resident "
                            "(consecutive observations share k - 1 frames); not an HBM-bound measurement")
    if scan:
        # the north star's named scan / reduction kernels: one launch each per rollout, a few MB --
        # latency-bound (the launch, not the bytes), reported against the same HBM roofline
        roofline["scan_kernels"] = scan
    # HBM traffic cannot be sampled from inside the process: it is taken from the
    # committed rocprofv3 --pmc passes of this same command
    # (profiles/rNN_pmc_gather.json, tools/pmc_gather.py), per launch shape.
    prof_dir = os.path.join(ROOT, "profiles")
    names = sorted((n for n in os.listdir(prof_dir) if n.endswith(("_pmc_gather.json", "_pmc_ppo.json", "_pmc_rainbow.json", "_pmc_sac.json"))),
                   reverse=True)      # newest round first
    # A PMC pass describes the build it was taken on: it carries the hash of the gather kernels'
    # sources (tools/pmc_gather.py: "kernel_sources_sha16") and is attached only while those files
    # are unchanged; otherwise traffic stays null and the line says which pass went stale.
    current = gather_sources_sha16()
    for name in names:
        try:
            pmc = json.load(open(os.path.join(prof_dir, name)))
            kk = pmc["kernels"].get("%s (%d %s)" % (kname, main_units, unit_name))
            if kk and "traffic_bytes_per_launch" in kk:
                taken_on = pmc.get("kernel_sources_sha16")
                if taken_on != current:
                    roofline["traffic_source"] = (
                        "none: profiles/%s was taken on gather sources %s, this build is %s"
                        % (name, taken_on or "of an untagged earlier round", current))
                    break
                roofline["traffic"] = kk["traffic_bytes_per_launch"]
                roofline["traffic_source"] = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / " \
                                             "WRITE_SIZE, separate passes, corrected; taken on " \
                                             "gather sources %s = this build)" % (name, taken_on)
                break
        except Exception:
            pass
    return roofline


def gather_sources_sha16():
    """sha256 (first 16 hex digits) over the sources of the gather kernels the roofline object
    describes: what a PMC pass is valid for."""
    import hashlib

    h = hashlib.sha256()
    for rel in ("pfrl_amd/csrc/replay.hip", "pfrl_amd/csrc/nhwc.h", "pfrl_amd/csrc/common.h"):
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


NATURE_FWD_FLOPS = 2 * (20 * 20 * 32 * 8 * 8 * 4 + 9 * 9 * 64 * 4 * 4 * 32 + 7 * 7 * 64 * 3 * 3 * 64
                        + 3136 * 512 + 512 * 6)      # per observation, pfrl/nn/atari_cnn.py:17-47
NATURE_CONV1_FLOPS = 2 * 20 * 20 * 32 * 8 * 8 * 4
MFMA_F32_PEAK_TFLOPS = 155.0    # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, measured


def step_flops_dqn(N, minibatch, update_interval):
    """Arithmetic of one batched DQN step: acting forward on N observations + per update the
    online forward / backward on B (backward = 2 x forward minus conv1's input gradient, which
    does not exist) and the target forward on B."""
    updates = N // update_interval
    per_update = minibatch * (NATURE_FWD_FLOPS + 2 * NATURE_FWD_FLOPS - NATURE_CONV1_FLOPS
                              + NATURE_FWD_FLOPS)
    return N * NATURE_FWD_FLOPS + updates * per_update


def step_flops_ppo(N, n_actions=6, value_passes=2.0):
    """Arithmetic EXECUTED by one batched PPO step (512 envs), the rollout's passes amortised per
    env step: acting forward, the value pass(es) over the rollout (``value_passes``: 2 = states and
    all next_states; 1 + the fraction of next_states actually evaluated when shared rows are taken
    from the first pass), and 4 epochs of forward + backward (backward = 2 x forward minus conv1's
    input gradient); the two narrow heads (512 -> A, 512 -> 1) counted with the trunk."""
    heads = 2 * 512 * (n_actions + 1)
    fwd = NATURE_FWD_FLOPS + heads
    return N * (fwd + value_passes * fwd + 4 * (3 * fwd - NATURE_CONV1_FLOPS))


def mfma_per_launch(agent, rbuf, B=32, n_actions=6):
    """Per launch of ONE DQN update, MEASURED in this run: the launches a captured update replays
    (``GraphedUpdate.measure_launches``: the same Python run eagerly on a fresh minibatch, every
    library entry point bracketed by a pair of timing events, median of 5) and the arithmetic each
    performs (the Nature CNN of pfrl/nn/atari_cnn.py:17-47 at minibatch B: forward per layer; a
    backward launch = input gradient + weight gradient of its layer = 2 x forward, conv1 has no
    input gradient; the hidden layer's RMSprop step rides in the last backward launch), as a
    fraction of the f32 MFMA peak.  Eager durations carry ~1 us of event bracketing each and no
    graph-internal boundary: their sum is not ``update_us`` (that is the range graph's own clock)."""
    conv1 = 2 * 20 * 20 * 32 * 8 * 8 * 4
    conv2 = 2 * 9 * 9 * 64 * 4 * 4 * 32
    conv3 = 2 * 7 * 7 * 64 * 3 * 3 * 64
    hidden = 2 * 3136 * 512
    head = 2 * 512 * n_actions
    fwd = iter([(conv1, "conv1 fwd"), (conv2, "conv2 fwd"), (conv3, "conv3 fwd"), (hidden, "hidden fwd")])
    bwd = iter([(2 * hidden, "hidden bwd (input + weight gradient)"), (2 * conv3, "conv3 bwd"),
                (2 * conv2, "conv2 bwd")])
    seqs = [rbuf.lookahead_sample(B)]
    big = rbuf.fetch_many(seqs, agent.phi, agent.gamma)
    ns = big["next_state"]
    raw = agent._precompute_target_raw(ns.view((B,) + tuple(ns.shape[2:])))
    big["target_next_raw"] = raw.view((1, B) + tuple(raw.shape[1:]))
    calls = agent._graphed.measure_launches({k: v[0] for k, v in big.items()})
    out = []
    for name, us in calls:
        if name == "pfrl_conv2d_nhwc_fwd":
            f, w = next(fwd, (0, "forward"))
        elif name == "pfrl_dqn_head_td_loss":
            f, w = 3 * head, "hidden-layer fold + head + TD loss + head bwd"
        elif name == "pfrl_conv2d_nhwc_bwd":
            f, w = next(bwd, (0, "backward"))
        elif name == "pfrl_conv2d_nhwc_bwd_weight_ride":
            f, w = conv1, "conv1 wgrad (+ the hidden layer's RMSprop step riding)"
        elif name == "pfrl_rmsprop_fused_step":
            f, w = 0, "RMSprop (slab folds + step of the convolutions and the head)"
        else:
            f, w = 0, "-"
        gf = f * B / 1e9
        out.append({"entry": name, "what": w, "us": round(us, 2), "gflop": round(gf, 4),
                    "frac": round(gf / 1e3 / (us * 1e-6) / MFMA_F32_PEAK_TFLOPS, 4) if us > 0 and f else None})
    return {"source": "measured in this run (hipEvent pair around each launch of one eager update)",
            "launches": out, "n_launches": len(out),
            "sum_us": round(sum(o["us"] for o in out), 1)}


def launches_per_update():
    """Kernel launches of one update, from the committed rocprofv3 timeline of this build
    (profiles/rNN_dqn_update_timeline.txt, tools/update_timeline.py), newest round first."""
    import re

    prof_dir = os.path.join(ROOT, "profiles")
    for name in sorted(os.listdir(prof_dir), reverse=True):
        if name.endswith("_dqn_update_timeline.txt"):
            m = re.search(r"kernels (\d+),", open(os.path.join(prof_dir, name)).read())
            if m:
                return {"value": int(m.group(1)), "source": "profiles/" + name}
    return None


def algorithmic_bytes_per_step(algo, N, minibatch, update_interval, value_passes=2.0):
    """SURVEY.md 8(d) per env-step figures x envs per batched step."""
    fb, k = 84 * 84, 4
    if algo in ("dqn", "rainbow"):
        rho = minibatch / update_interval
        return N * (fb + (k * fb + 4 * k * fb) + rho * 2 * (k * fb + 4 * k * fb))
    if algo == "ppo":
        # act 141,120 + ring 7,056 + value pass + 4 epochs x 141,120 + GAE 24 +
        # adv-norm 12.  The reference evaluates V on states AND next_states (2 x 141,120: 994,932 B);
        # priced here are the bytes the build MOVES: next_states that are the next step's state are
        # not gathered again (value_passes = 1 + evaluated fraction: the 0.854 MB variant SURVEY
        # says to flag -- flagged in config.workload; the VALUES are the full second pass's).
        return N * (141120 + 7056 + value_passes * 141120 + 4 * 141120 + 24 + 12)
    return N * (2 * minibatch * 3084 + 3084)   # sac
