"""The per-env loop of the vector-observation agents through the native step planner.

``SoftActorCritic.batch_observe`` / ``TD3`` / ``DDPG`` (reference
pfrl/agents/soft_actor_critic.py:354-374, td3.py:283-303, ddpg.py:207-227) walk the envs of a
batched step one by one: ``replay_buffer.append``, ``stop_current_episode`` at episode ends, and
-- when ``t`` reaches a multiple of ``update_interval`` with enough transitions stored --
``sample_n_k(len, B)`` on NumPy's global stream for each due update.  With a uniform one-step
device ReplayBuffer none of it needs the interpreter per env:

* the observations of the range (f32 vectors from a host env) enter the frame ring in one pass
  and one stacked transfer (``DeviceReplayStore.ingest_vectors``; observation identity is
  resolved through the same cache as the per-observation ``ingest``, so ``state`` of this step
  is found where ``next_state`` of the last one left it);
* the appends (transition rows, one-transition entries, host mirrors, RandomAccessQueue head)
  and every index set the loop would draw between them come from ONE call of the planner the
  DQN device step uses (``pfrl_plan_dqn_range``, csrc/hostplan.hip: it is generic in the number
  of frames per observation) on NumPy's OWN generator -- same stream position afterwards as the
  Python loop;
* rows, entries, the float action rows and the sampled entry slots cross PCIe in ONE pinned
  block; ``pfrl_table_append`` / ``pfrl_entries_append`` write the tables, the fused gather
  reads the slots where they landed.

Everything here is a fast path with the general code behind it: any condition that does not hold
(prioritized / n-step / episodic replay, per-transition extras, observations that are not plain
arrays, an arbitrary ``phi`` applied at ingest, the dense ``sample_n_k`` regime of a tiny
buffer) returns None and ``ReplayActorCritic._observe_range_fused`` runs its Python loop.
``PFRL_VECTOR_PLANNER=0`` switches the path off (A/B, tests).
"""
import ctypes
import os

import numpy as np
import torch

from pfrl_amd import _native, host_plan
from pfrl_amd.staging import StagingRing, on_stream


def _align16(x):
    return (x + 15) & ~15


def enabled():
    return os.environ.get("PFRL_VECTOR_PLANNER", "1") != "0"


def plan_range(agent, lo, hi, batch_obs, batch_reward, batch_done, batch_reset):
    """Appends + index draws of envs [lo, hi).  Returns (U, slots_dev) -- the number of updates
    due in the range and their sampled entry slots on the device ([U * B] int32, None when
    U == 0) -- or None when the Python loop has to run (nothing has been touched then, except
    that observations may already sit in the frame ring, where ``ingest`` will find them)."""
    rbuf, up = agent.replay_buffer, agent.replay_updater
    if not (enabled() and getattr(rbuf, "vector_range_append_supported", None) is not None
            and rbuf.vector_range_append_supported()):
        return None
    st = rbuf.store
    last_obs, last_act = agent.batch_last_obs, agent.batch_last_action
    m = hi - lo
    pairs = []
    for i in range(lo, hi):
        s = last_obs[i]
        if s is None or last_act[i] is None:
            return None
        pairs.append(s)
        pairs.append(batch_obs[i])
    try:
        actions = np.stack([np.asarray(last_act[i], dtype=np.float32).reshape(-1)
                            for i in range(lo, hi)])
    except ValueError:
        return None
    A = st.act_dim
    if actions.shape != (m, A):
        return None
    got = st.ingest_vectors(pairs)
    if got is None:
        return None
    refs, seqs = got
    s_refs, n_refs = np.ascontiguousarray(refs[0::2]), np.ascontiguousarray(refs[1::2])
    s_seq, n_seq = np.ascontiguousarray(seqs[0::2]), np.ascontiguousarray(seqs[1::2])
    reward = np.ascontiguousarray(batch_reward[lo:hi], dtype=np.float64)
    done = np.ascontiguousarray(batch_done[lo:hi], dtype=np.uint8)
    ended = np.logical_or(np.asarray(batch_done[lo:hi], dtype=bool),
                          np.asarray(batch_reset[lo:hi], dtype=bool))

    planner = agent.__dict__.get("_vec_planner")
    if planner is None or not planner.valid_for(rbuf):
        planner = agent._vec_planner = host_plan.DQNRangePlanner(rbuf, float_actions=True)
    B = up.batchsize
    plan_bytes = _align16(planner.block_bytes(m, (m // up.update_interval + 1) * up.n_times_update, B))
    need = plan_bytes + 4 * m * A
    ring = agent.__dict__.get("_vec_ring")
    if ring is None or ring.slot_bytes < need:
        ring = agent._vec_ring = StagingRing(agent.device, slot_bytes=max(1 << 16, need), n_slots=8)
    st.flush()                  # rows the Python path may have left pending go first
    host, tok = ring.reserve()
    n_trans0 = st.n_trans
    U = planner.plan(s_refs, s_seq, n_refs, n_seq, reward, done, agent.t, up.replay_start_size,
                     up.update_interval, up.n_times_update, B, host[:plan_bytes])
    if U == host_plan.PLAN_DENSE:
        return None
    # the action rows: host mirror (checkpoints, transition views) and, behind the planner's
    # part of the block, the copy the table kernel reads
    tslots = (n_trans0 + np.arange(m, dtype=np.int64)) % st.R
    st.h_action[tslots] = actions
    host[plan_bytes:need].view(np.float32)[:] = actions.reshape(-1)
    o = planner.offs
    lib = _native.lib()
    V = ctypes.c_void_p
    with on_stream(st.side_stream):
        dev = ring.commit(tok, need)
        base = dev.data_ptr()
        stream = V(torch.cuda.current_stream().cuda_stream)
        _native.check(lib.pfrl_table_append(ctypes.byref(st.desc), m, V(base + int(o[0])),
                                            V(base + int(o[1])), V(base + int(o[2])),
                                            V(base + plan_bytes), V(base + int(o[3])),
                                            V(base + int(o[4])), stream), "table_append")
        _native.check(lib.pfrl_entries_append(ctypes.byref(st.desc), m, V(base + int(o[5])),
                                              V(base + int(o[6])), V(base + int(o[7])), stream),
                      "entries_append")
    # episode ends: what _append does after the append of env i (stop_current_episode of a
    # one-step buffer only empties the env's window, which this path never fills)
    for j in np.flatnonzero(ended):
        i = lo + int(j)
        last_obs[i] = None
        last_act[i] = None
        rbuf.stop_current_episode(env_id=i)
    if U == 0:
        return 0, None
    slots_dev = dev[int(o[8]):int(o[8]) + 4 * U * B].view(torch.int32)
    return U, slots_dev
