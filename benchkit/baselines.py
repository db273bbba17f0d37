"""``cpu_baseline`` legs and the zero-FLOP data-path measurement (bench.py)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from benchkit.supervisor import _tick
from benchkit.workloads import build_agent, one_step

def cpu_baseline(args, seconds):
    """The same workload through the CPU oracle (oracle/pfrl_oracle.c = plain C
    restatement of the reference's data path) plus the same network in torch
    CPU, on this box's host cores, for a bounded sample.  kind = "port"."""
    N, B = args.num_envs, args.minibatch
    avail = torch.get_num_threads()
    cores = max(1, min(avail, args.cpu_baseline_threads))
    torch.set_num_threads(cores)
    try:
        return _cpu_baseline_run(args, seconds, N, B, cores)
    finally:
        torch.set_num_threads(avail)


def _cpu_baseline_run(args, seconds, N, B, cores):
    import oracle
    import pfrl_amd as pfrl
    from pfrl_amd.agents.dqn import compute_value_loss
    from pfrl_amd.initializers import init_chainer_default
    from pfrl_amd.q_functions import DiscreteActionValueHead
    from pfrl_amd.utils.random import sample_n_k

    rs = np.random.RandomState(0)
    F = 20000
    frames = rs.randint(0, 256, size=(F, 84 * 84)).astype(np.uint8)
    cap = 100000  # host memory bound, stated in the sample description
    t_state = rs.randint(0, F, size=(cap, 4)).astype(np.int32)
    t_next = rs.randint(0, F, size=(cap, 4)).astype(np.int32)
    rewards = rs.choice([-1.0, 0.0, 1.0], size=cap)
    terms = (rs.rand(cap) < 0.002).astype(np.uint8)
    actions = rs.randint(0, 6, size=cap)
    torch.manual_seed(0)
    q = torch.nn.Sequential(pfrl.nn.LargeAtariCNN(),
                            init_chainer_default(torch.nn.Linear(512, 6)),
                            DiscreteActionValueHead())
    tq = torch.nn.Sequential(pfrl.nn.LargeAtariCNN(), torch.nn.Linear(512, 6),
                             DiscreteActionValueHead())
    tq.load_state_dict(q.state_dict())
    opt = torch.optim.RMSprop(q.parameters(), lr=2.5e-4, alpha=0.95, eps=1e-2, centered=True)
    n_updates_per_step = N // args.update_interval
    t0 = time.perf_counter()
    updates = 0
    data_s = 0.0
    done = False
    while not done:
        d0 = time.perf_counter()
        refs = rs.randint(0, F, size=(N, 4)).astype(np.int32)
        x = oracle.batch_states_u8(frames, refs, 255.0).reshape(N, 4, 84, 84)
        data_s += time.perf_counter() - d0
        with torch.no_grad():
            q(torch.from_numpy(x)).greedy_actions.numpy()
        for _ in range(n_updates_per_step):
            d0 = time.perf_counter()
            idx = sample_n_k(cap, B)
            ents = [[int(i)] for i in idx]
            sc = oracle.batch_experiences_scalars(ents, rewards, terms, 0.99, 1)
            s = oracle.batch_states_u8(frames, t_state[idx], 255.0).reshape(B, 4, 84, 84)
            ns = oracle.batch_states_u8(frames, t_next[idx], 255.0).reshape(B, 4, 84, 84)
            data_s += time.perf_counter() - d0
            qout = q(torch.from_numpy(s))
            y = qout.evaluate_actions(torch.from_numpy(actions[idx]))
            with torch.no_grad():
                nq = tq(torch.from_numpy(ns)).max
                t = (torch.from_numpy(sc["reward"]) + torch.from_numpy(sc["discount"])
                     * (1.0 - torch.from_numpy(sc["is_state_terminal"])) * nq)
            loss = compute_value_loss(y, t, True, "sum")
            opt.zero_grad()
            loss.backward()
            opt.step()
            updates += 1
            # the sample is bounded by time, at update granularity: a batched step is
            # 64 updates (several seconds on the host), so fractions of a step count
            if time.perf_counter() - t0 >= seconds:
                done = True
                break
    el = time.perf_counter() - t0
    steps = updates / n_updates_per_step
    return {
        "value": round(steps * N / el, 2), "unit": "env-steps/s", "cores": cores, "kind": "port",
        "data_path_only_value": round(steps * N / max(data_s, 1e-9), 2),
        "sample": "%.2f batched steps of %d envs (%d updates of B=%d) in %.1f s; oracle C data "
                  "path (single thread) + torch-CPU Nature CNN (%d threads); replay capacity 1e5 on "
                  "the host" % (steps, N, updates, B, el, cores),
    }


class _ZeroFlopQ(torch.nn.Module):
    """q_function stand-in for the data-path-only figure: Q-values that do
    not depend on the observation, one learnable row, so that every replay / gather / loss /
    optimizer launch of the step still happens and the network costs nothing."""

    def __init__(self, n_actions):
        super().__init__()
        self.q = torch.nn.Parameter(torch.zeros(1, n_actions))

    def forward(self, x):
        from pfrl_amd.action_value import DiscreteActionValue

        return DiscreteActionValue(self.q.expand(x.shape[0], self.q.shape[1]))


def data_path_only(args, device, agent, env, rbuf, obss, steps):
    """The same batched step over the same (full) replay buffer and env with a zero-FLOP
    q_function: appends, index draws, the fused gathers, TD loss and optimizer step remain."""
    from pfrl_amd import agents
    from pfrl_amd.optimizers import FusedRMSprop

    N = args.num_envs
    q = _ZeroFlopQ(6)
    opt = FusedRMSprop(q.parameters(), lr=2.5e-4, alpha=0.95, eps=1e-2, centered=True)
    stub = agents.DQN(q, opt, rbuf, gpu=device.index, gamma=0.99, explorer=agent.explorer,
                      replay_start_size=agent.replay_start_size,
                      target_update_interval=3 * 10 ** 4, clip_delta=True,
                      update_interval=args.update_interval, minibatch_size=args.minibatch,
                      batch_accumulator="sum", phi=agent.phi)
    stub.step_fused_chunks = ()   # nothing to overlap host preparation with: one range
    stub.t = agent.t
    for _ in range(3):
        obss = one_step(stub, env, obss, N)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        obss = one_step(stub, env, obss, N)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    out = {"value": round(N * steps / el, 1), "unit": "env-steps/s", "steps": steps,
           "ms_per_step": round(el / steps * 1e3, 3),
           "what": "the same step with a zero-FLOP q_function (SURVEY.md 8d): env frames, "
                   "act gather, appends, index draws, fused minibatch gathers of the full "
                   "schedule, TD loss, optimizer step on one row"}
    # ... and with the per-update launches gone too (the zero-FLOP network still costs 4 launches
    # per update, 256 per step, which is all that bounds the figure above): what the replay side
    # ALONE sustains -- env frames, the acting gather + action select, the native planner, one
    # transfer, appends and the 2 048-entry gather of the step's whole schedule.

    class _NoUpdates:
        graphs = {("range",): None}
        pipeline = False

        def range_capturable(self):
            return True

        zeros = {}

        def run_range(self, big):
            U, B = big["reward"].shape[:2]
            z = self.zeros.get((U, B))
            if z is None:
                z = self.zeros[(U, B)] = (torch.zeros(U, device=device),
                                          torch.zeros(U * B, device=device))
            return z

    single = not (torch.distributed.is_available() and torch.distributed.is_initialized()
                  and torch.distributed.get_world_size() > 1)
    if not (single and stub.use_graphs and stub.range_graphs):
        return out, obss       # (the range-graph path is what the stand-in below replaces)
    try:
        stub._graphed = _NoUpdates()
        stub.batch_target_pass = False
        stub.target_update_interval = 10 ** 12   # (no sync inside a range: every range is "one graph")
        for _ in range(3):
            obss = one_step(stub, env, obss, N)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            obss = one_step(stub, env, obss, N)
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t0
        out["without_update_launches"] = {
            "value": round(N * steps / el2, 1), "ms_per_step": round(el2 / steps * 1e3, 3),
            "what": "the replay side alone: env frames, acting gather + action select, native "
                    "planner + one transfer, appends, the fused gather of all 64 minibatches of the "
                    "step; no per-update launch"}
        data_path_only.last_stub = stub          # (tools/data_path_phases.py times its phases)
    except Exception as e:      # an extra figure must never cost the line its numbers
        sys.stderr.write("data_path_only.without_update_launches failed: %r\n" % (e,))
    return out, obss


def reference_baseline(args):
    """pfnet/pfrl ITSELF (gpu=-1) on the same synthetic workload, timed on THIS box's host cores
    by tools/reference_cpu_baseline.py in a subprocess that imports the reference from
    oracle/_ref/ (its modules compiled to .pyc by oracle/build_ref.py; /root/reference does not
    exist on the GPU box).  A bounded sample: BASELINE.md section 3's full protocol (>= 2e4
    env-steps, 3 seeds) is profiles/r03_reference_cpu_baseline_gpubox.json."""
    import subprocess

    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    if not os.path.isdir(os.path.join(ref_dir, "pfrl")):
        return None
    env = dict(os.environ, PFRL_REFERENCE=ref_dir, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    cmd = [sys.executable, os.path.join(ROOT, "tools", "reference_cpu_baseline.py"),
           "--seconds", str(args.cpu_baseline_seconds), "--dp-seconds",
           str(max(2.0, args.cpu_baseline_seconds * 0.4)), "--prefill", "5120",
           "--threads", str(args.cpu_baseline_threads), "--num-envs", str(args.num_envs)]
    try:
        out = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=600)
        d = json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as e:     # the baseline must never cost the line its GPU numbers
        sys.stderr.write("reference cpu baseline failed: %r\n" % (e,))
        return None
    return {
        "value": d["end_to_end"]["value"], "unit": "env-steps/s", "cores": d["cores"],
        "kind": "reference",
        "data_path_only_value": d["data_path_only"]["value"],
        "host_cores": d["host_cores"],
        "sample": "pfnet/pfrl itself (compiled from /root/reference into oracle/_ref), gpu=-1, "
                  "train loop of pfrl/agents/dqn.py on %d in-process synthetic Atari-shaped envs, "
                  "ReplayBuffer(1e5) holding %d transitions at the start: %d env-steps end to end "
                  "in %.0f s with %d torch threads, %d env-steps with a zero-FLOP q_function; "
                  "full protocol (>= 2e4 env-steps, 3 seeds, median): profiles/"
                  "r03_reference_cpu_baseline_gpubox.json"
                  % (d["num_envs"], d["replay_len_at_start"],
                     d["end_to_end"]["env_steps_per_sample"][0], args.cpu_baseline_seconds,
                     d["cores"], d["data_path_only"]["env_steps_per_sample"][0]),
    }


def reference_baseline_ppo(args, num_envs=512, steps=16):
    """The reference's PPO (pfrl/agents/ppo.py:465-532, gpu=-1, the model and hyperparameters of
    examples/atari/train_ppo_ale.py:247-264) on this box's host cores, by the same tool and the same
    oracle/_ref/ copy as :func:`reference_baseline`.  Bounded: rollouts of ``steps`` steps instead of
    128 (update_interval and minibatch scaled with them: every transition still gets one acting
    forward, one value pass and 4 epochs), one untimed + one timed rollout INCLUDING its update.
    The full-size figure (128-step rollouts) is profiles/r04_reference_cpu_baseline_ppo_gpubox.json."""
    import subprocess

    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    if not os.path.isdir(os.path.join(ref_dir, "pfrl")):
        return None
    env = dict(os.environ, PFRL_REFERENCE=ref_dir, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    cmd = [sys.executable, os.path.join(ROOT, "tools", "reference_cpu_baseline.py"), "--algo", "ppo",
           "--num-envs", str(num_envs), "--ppo-steps", str(steps), "--ppo-rollouts", "1",
           "--threads", str(args.cpu_baseline_threads)]
    try:
        out = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=600)
        d = json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as e:     # the baseline must never cost the line its GPU numbers
        sys.stderr.write("reference PPO cpu baseline failed: %r\n" % (e,))
        return None
    return {
        "value": d["end_to_end"]["value"], "unit": "env-steps/s", "cores": d["cores"],
        "kind": "reference", "host_cores": d["host_cores"],
        "sample": "pfnet/pfrl itself (oracle/_ref), gpu=-1, pfrl/agents/ppo.py on %d in-process "
                  "synthetic Atari-shaped envs: one %d-step rollout INCLUDING its update "
                  "(update_interval=%d, minibatch=%d, 4 epochs; BASELINE's rollout is 128 steps, the "
                  "per-transition work is the same): %d env-steps in %.0f s with %d torch threads; "
                  "full size: profiles/r04_reference_cpu_baseline_ppo_gpubox.json"
                  % (d["num_envs"], d["rollout_steps"], d["update_interval"], d["minibatch"],
                     d["end_to_end"]["env_steps"], d["end_to_end"]["seconds"], d["cores"]),
    }


def reference_baseline_other(args, algo, num_envs, seconds=8.0):
    """The reference's Rainbow / SAC (gpu=-1; tools/reference_cpu_baseline.py --algo rainbow|sac, the
    constructions of train_rainbow.py:110-159 / train_soft_actor_critic.py:172-243) on this box's host
    cores from the same oracle/_ref/ copy: a bounded sample of whole env steps with their updates."""
    import subprocess

    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    if not os.path.isdir(os.path.join(ref_dir, "pfrl")):
        return None
    env = dict(os.environ, PFRL_REFERENCE=ref_dir, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    cmd = [sys.executable, os.path.join(ROOT, "tools", "reference_cpu_baseline.py"), "--algo", algo,
           "--num-envs", str(num_envs), "--seconds", str(seconds), "--prefill", "5120",
           "--threads", str(args.cpu_baseline_threads)]
    try:
        out = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=600)
        d = json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as e:     # the baseline must never cost the line its GPU numbers
        sys.stderr.write("reference %s cpu baseline failed: %r\n" % (algo, e))
        return None
    return {
        "value": d["end_to_end"]["value"], "unit": "env-steps/s", "cores": d["cores"],
        "kind": "reference", "host_cores": d["host_cores"],
        "sample": "pfnet/pfrl itself (oracle/_ref), gpu=-1, %s on %d in-process synthetic envs, replay "
                  "capacity 1e5 holding %d transitions at the start: %d env-steps (%d updates) in %.0f s "
                  "with %d torch threads"
                  % (d["what"].split("pfrl ")[1].split(" (")[0], d["num_envs"], d["replay_len_at_start"],
                     d["end_to_end"]["env_steps"], d["end_to_end"]["updates"],
                     d["end_to_end"]["seconds"], d["cores"]),
    }
