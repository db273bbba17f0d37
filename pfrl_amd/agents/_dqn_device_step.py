"""The batched DQN step with NO host round trip (device env + uniform one-step device replay).

``DQN.batch_act`` / ``DQN._batch_observe_train`` (reference pfrl/agents/dqn.py:490-549) spend
their time on the host: a per-env epsilon-greedy loop that has to wait for the greedy actions
(``.cpu().numpy()``, :497), a per-env append loop and one ``sample_n_k`` per update.  None of it
needs the GPU's results:

* the epsilon-greedy draws do not depend on the network -- the native planner
  (``pfrl_amd.host_plan``, csrc/hostplan.hip) makes them on NumPy's global stream, the GPU
  resolves ``action = draw fired ? random action : argmax Q`` (``pfrl_select_actions``) and
  the actions stay in HBM as :class:`DeviceActions`;
* the appends of an env range, the queue bookkeeping and every index set the reference's loop
  would draw between them come from ONE planner call, written into ONE pinned block that
  crosses PCIe in one transfer; the transition rows take their action column from the device
  tensor.

So the host never blocks on the device inside a step: it runs ahead (bounded by ``RUN_AHEAD``
steps) and the GPU queue stays full.  Order of appends, NumPy draws, target syncs and updates is
the reference's (tests/test_host_plan.py: stream position after every call;
tests/test_bench_path_parity.py, tests/test_agent_parity.py: whole traces).

Everything here is a fast path with the general code behind it: any condition that does not
hold (another explorer, a host env, prioritized or n-step replay, recurrent model, the dense
``sample_n_k`` regime of a tiny buffer) falls back to ``DQN``'s own methods.
"""
import collections
import ctypes
import logging
import os

import numpy as np
import torch

from pfrl_amd import _native, host_plan, ops
from pfrl_amd.device_store import DeviceActions, DeviceObsBatch
from pfrl_amd.staging import StagingRing
from pfrl_amd.utils.contexts import evaluating

RUN_AHEAD = 3       # batched steps the host may be ahead of the GPU


class LastBatch:
    """``agent.batch_last_obs`` / ``batch_last_action`` over a device batch: indexable like the
    lists the reference keeps (pfrl/agents/dqn.py:503-505), entries can be set to None, nothing
    is materialised unless somebody looks."""

    def __init__(self, batch):
        self.batch = batch
        self.cleared = None

    def __len__(self):
        return len(self.batch)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if self.cleared is not None and self.cleared[i]:
            return None
        return self.batch[i]

    def __setitem__(self, i, value):
        assert value is None
        if self.cleared is None:
            self.cleared = np.zeros(len(self), dtype=bool)
        self.cleared[i] = True

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def all_set(self, lo, hi):
        return self.cleared is None or not self.cleared[lo:hi].any()


def _align16(x):
    return (x + 15) & ~15


def act_split(agent):
    """(body, head layer) when the model is ``Sequential(..., Linear(K, A <= 16),
    DiscreteActionValueHead())`` -- the example Q-network (examples/atari/train_dqn_batch_ale.py:35-41):
    its narrow head, the argmax and the epsilon-greedy decision then run as ONE launch
    (pfrl_dqn_act_head) instead of head + torch argmax + cast + select.  ``body`` is a view of the model
    without its last two children (same class, children and parameters: a fused trunk stays fused)."""
    model = agent.model
    hit = agent.__dict__.get("_act_split_cache")
    if hit is not None and hit[0] is model and hit[2] == tuple(id(m) for m in model.children()):
        return hit[1]
    from pfrl_amd.q_functions import DiscreteActionValueHead

    out = None
    nn = torch.nn
    if (os.environ.get("PFRL_DQN_ACT_HEAD", "1") != "0" and isinstance(model, nn.Sequential)
            and len(model) >= 3 and type(model[len(model) - 1]) is DiscreteActionValueHead):
        lin = model[len(model) - 2]
        if (isinstance(lin, nn.Linear) and 1 <= lin.out_features <= 16
                and lin.weight.dtype == torch.float32 and lin.weight.is_contiguous()):
            body = object.__new__(type(model))
            body.__dict__ = dict(model.__dict__)
            body._modules = collections.OrderedDict(list(model._modules.items())[:-2])
            out = (body, lin)
    agent._act_split_cache = (model, out, tuple(id(m) for m in model.children()))
    return out


class ActGraph:
    """The device side of ``DQN.batch_act`` on the device-step path -- observation gather, trunk,
    Q head + argmax + epsilon-greedy decision -- as ONE captured HIP graph per batch shape, fed by
    the step's single staging transfer (frame slots + the host's draws) landing in the graph's own
    input block (reference pfrl/agents/dqn.py:490-507).  Two instances alternate, so the actions
    of a step stay valid while the next step's are computed.  The graph reads the frame ring and
    the parameters in place: optimizer steps, target syncs and new frames need no re-capture."""

    def __init__(self, agent):
        self.agent = agent
        self.entries = {}
        self.pool = None

    @staticmethod
    def applicable(agent):
        return (os.environ.get("PFRL_DQN_ACT_GRAPH", "1") != "0" and agent.use_graphs
                and act_split(agent) is not None)

    def _body(self, agent, batch_obs, inp, N, k, off_choice, out):
        refs_bytes = 4 * N * k
        batch_obs._refs_dev = inp[:refs_bytes].view(torch.int32).view(N, k)
        choice = inp[off_choice:off_choice + 4 * N].view(torch.int32)
        return _act_launches(agent, batch_obs, choice, out)

    def _capture(self, agent, batch_obs, N, k, off_choice, need):
        from pfrl_amd.agents.graphed_update import _capturing

        dev = agent.device
        pair = []
        for _ in range(2):
            inp = torch.zeros(_align16(need), dtype=torch.uint8, device=dev)
            inp[:4 * N * k].view(torch.int32).copy_(
                torch.from_numpy(np.ascontiguousarray(batch_obs.refs.reshape(-1))).to(dev))
            inp[off_choice:off_choice + 4 * N].view(torch.int32).fill_(-1)
            out = torch.empty(N, dtype=torch.int64, device=dev)
            cur = torch.cuda.current_stream(dev)
            side = torch.cuda.Stream(dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for _ in range(2):
                    if self._body(agent, batch_obs, inp, N, k, off_choice, out) is None:
                        cur.wait_stream(side)
                        return None
            cur.wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with ops.profile_paused(), _capturing(g, self.pool):
                self._body(agent, batch_obs, inp, N, k, off_choice, out)
            if self.pool is None:
                self.pool = g.pool()
            pair.append((g, inp, out))
        return [pair, 0]

    def run(self, agent, batch_obs, ring, tok, N, k, off_choice, need):
        """Ships pinned slot ``tok`` into the graph's input block and replays; returns the
        actions (int64 [N], owned by the graph instance) or None when the model cannot take the
        fused head (the caller then commits the slot to the ring and launches eagerly)."""
        model = agent.model
        fr = batch_obs.store
        key = (N, k, id(model), tuple(id(m) for m in model.modules()), id(fr),
               getattr(fr, "emit_channels_last", None), id(agent.phi))
        e = self.entries.get(key)
        if e is None:
            if len(self.entries) > 8:
                self.entries.clear()
            e = self.entries[key] = self._capture(agent, batch_obs, N, k, off_choice, need) or False
        if e is False:
            return None
        pair, at = e
        e[1] = at ^ 1
        g, inp, out = pair[at]
        ring.commit_to(tok, need, inp)
        batch_obs._refs_dev = inp[:4 * N * k].view(torch.int32).view(N, k)
        g.replay()
        return out


def _act_launches(agent, batch_obs, choice, out=None):
    """gather -> trunk -> fused head: the actions (int64 [N]), or None when the hidden activations
    are not what pfrl_dqn_act_head reads."""
    split = act_split(agent)
    with torch.no_grad(), evaluating(agent.model):
        if split is None:
            greedy = agent._evaluate_model(batch_obs).greedy_actions.detach()
            return ops.select_actions(greedy, choice, out=out)
        body, lin = split
        agent._route_observation_layout(batch_obs)
        h = body(agent.batch_states(batch_obs, agent.device, agent.phi))
        if not (torch.is_tensor(h) and h.dim() == 2 and h.dtype == torch.float32 and h.is_contiguous()
                and h.shape[1] == lin.in_features):
            if out is not None:
                return None
            greedy = list(agent.model._modules.values())[-1](lin(h)).greedy_actions.detach()
            return ops.select_actions(greedy, choice)
        return ops.dqn_act_head(h, lin.weight, lin.bias, choice, out=out)[0]


def act(agent, batch_obs):
    """``DQN.batch_act`` in training mode for a DeviceObsBatch; None = conditions not met."""
    if not (agent.training and isinstance(batch_obs, DeviceObsBatch) and not agent.recurrent
            and agent.device.type == "cuda" and agent._explorer_draws_before_greedy()):
        return None
    ex = agent.explorer
    n_act = host_plan.recognise_randint(ex.random_action_func)
    if n_act is None or ex.logger.isEnabledFor(logging.DEBUG):
        return None
    N, k = batch_obs.refs.shape
    refs_bytes = 4 * N * k
    off_choice = _align16(refs_bytes)
    need = off_choice + 4 * N
    ring = agent.__dict__.get("_act_ring")
    if ring is None or ring.slot_bytes < need:
        ring = agent._act_ring = StagingRing(agent.device, slot_bytes=max(1 << 14, need + 64),
                                             n_slots=2 * RUN_AHEAD + 2)
        agent._step_events = collections.deque()
    # bounded run-ahead: wait for the step RUN_AHEAD steps back
    evs = agent._step_events
    while len(evs) >= RUN_AHEAD:
        evs.popleft().synchronize()
    host, tok = ring.reserve()
    host[:refs_bytes].view(np.int32)[:] = batch_obs.refs.reshape(-1)
    choice = host[off_choice:need].view(np.int32)
    ex.epsilon = eps = ex.compute_epsilon(agent.t)
    host_plan.eps_greedy(N, eps, n_act, out=choice)
    actions = None
    if ActGraph.applicable(agent):
        graph = agent.__dict__.get("_act_graph")
        if graph is None:
            graph = agent._act_graph = ActGraph(agent)
        actions = graph.run(agent, batch_obs, ring, tok, N, k, off_choice, need)
    if actions is None:
        dev = ring.commit(tok, need)
        batch_obs._refs_dev = dev[:refs_bytes].view(torch.int32).view(N, k)
        actions = _act_launches(agent, batch_obs, dev[off_choice:need].view(torch.int32))
    ev = torch.cuda.Event()
    ev.record()
    evs.append(ev)
    agent._last_obs_batch = batch_obs
    agent._last_actions_dev = actions
    agent.batch_last_obs = LastBatch(batch_obs)
    agent.batch_last_action = LastBatch(DeviceActions(actions))
    return agent.batch_last_action.batch


def begin_observe(agent, batch_obs, batch_reward, batch_done, batch_reset):
    """Per-step preparation of the native append path: contiguous reward / done columns."""
    agent._obs_cols = None
    acts = agent.__dict__.get("_last_actions_dev")
    prev = agent.__dict__.get("_last_obs_batch")
    rbuf = agent.replay_buffer
    if not (acts is not None and isinstance(batch_obs, DeviceObsBatch)
            and isinstance(prev, DeviceObsBatch) and prev.store is batch_obs.store
            and len(prev) == len(batch_obs) == acts.shape[0]
            and isinstance(agent.batch_last_obs, LastBatch)
            and agent.batch_last_obs.batch is prev
            and not agent.recurrent and rbuf.batch_append_supported(batch_obs)):
        return
    agent._obs_cols = (np.ascontiguousarray(batch_reward, dtype=np.float64),
                       np.ascontiguousarray(batch_done, dtype=np.uint8),
                       np.logical_or(np.asarray(batch_done, dtype=bool),
                                     np.asarray(batch_reset, dtype=bool)))


def plan_range(agent, lo, hi, batch_obs):
    """Appends + index draws of envs [lo, hi) through the native planner.  Returns
    (plan_env, slots_dev, U) -- the env after whose append each update runs, the sampled entry
    slots on the device, the number of updates -- or None when the Python path has to run."""
    cols = agent.__dict__.get("_obs_cols")
    if cols is None or not agent.batch_last_obs.all_set(lo, hi):
        return None
    rbuf, up = agent.replay_buffer, agent.replay_updater
    st = rbuf.store
    planner = agent.__dict__.get("_planner")
    if planner is None or not planner.valid_for(rbuf):
        planner = agent._planner = host_plan.DQNRangePlanner(rbuf)
    prev = agent._last_obs_batch
    N = len(prev)
    B = up.batchsize
    m = hi - lo
    need = planner.block_bytes(N, (N // up.update_interval + 1) * up.n_times_update, B)
    ring = agent.__dict__.get("_obs_ring")
    if ring is None or ring.slot_bytes < need:
        ring = agent._obs_ring = StagingRing(agent.device, slot_bytes=max(1 << 16, need),
                                             n_slots=4 * RUN_AHEAD + 4)
    st.flush()                  # rows the Python path may have left pending go first
    reward, done, ended = cols
    host, tok = ring.reserve()
    t0 = agent.t
    U = planner.plan(prev.refs[lo:hi], prev.min_seq[lo:hi], batch_obs.refs[lo:hi],
                     batch_obs.min_seq[lo:hi], reward[lo:hi], done[lo:hi], t0,
                     up.replay_start_size, up.update_interval, up.n_times_update, B, host)
    if U == host_plan.PLAN_DENSE:
        return None
    o = planner.offs
    dev = ring.commit(tok, int(o[9]))
    base = dev.data_ptr()
    lib = _native.lib()
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    V = ctypes.c_void_p
    a_ptr = agent._last_actions_dev.data_ptr() + 8 * lo
    _native.check(lib.pfrl_table_append(ctypes.byref(st.desc), m, V(base + int(o[0])),
                                        V(base + int(o[1])), V(base + int(o[2])), V(a_ptr),
                                        V(base + int(o[3])), V(base + int(o[4])), stream),
                  "table_append")
    _native.check(lib.pfrl_entries_append(ctypes.byref(st.desc), m, V(base + int(o[5])),
                                          V(base + int(o[6])), V(base + int(o[7])), stream),
                  "entries_append")
    st.h_action_stale = True     # the action column exists on the device only
    if ended[lo:hi].any():
        for i in np.flatnonzero(ended[lo:hi]):
            agent.batch_last_obs[lo + int(i)] = None
            agent.batch_last_action[lo + int(i)] = None
    if U == 0:
        return [], None, 0
    ui = up.update_interval
    first = (-(t0 + 1)) % ui          # offset of the first env whose t is a multiple of ui
    # (updates only start once len >= replay_start_size: the LAST U // n_times due envs)
    due = [lo + j for j in range(first, m, ui)]
    due = due[len(due) - U // up.n_times_update:]
    plan_env = [i for i in due for _ in range(up.n_times_update)]
    slots_dev = dev[int(o[8]):int(o[8]) + 4 * U * B].view(torch.int32)
    return plan_env, slots_dev, U
