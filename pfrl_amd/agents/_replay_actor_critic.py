"""Shared plumbing of the continuous-action replay agents (DDPG, TD3).

Everything these agents have in common with the reference's own copies of the
same code (pfrl/agents/ddpg.py:253-303, td3.py:262-317): burn-in / explorer
acting, the per-env append + ``stop_current_episode`` + ``update_if_necessary``
loop, device-resident statistics -- plus what is specific to this package: the
HBM replay store binding and HIP-graph capture of the whole update
(:class:`pfrl_amd.agents.graphed_update.CapturedStep`).
"""
import torch

from pfrl_amd.agent import AttributeSavingMixin, BatchAgent
from pfrl_amd.agents.dqn import _DeviceRecord, _mean_or_nan
from pfrl_amd.replay_buffer import ReplayUpdater, batch_experiences
from pfrl_amd.utils.contexts import evaluating


class ReplayActorCritic(AttributeSavingMixin, BatchAgent):
    # name -> window of the statistics the update produces (subclass)
    _STATS = ()

    def _setup(self, modules, gpu, replay_buffer, phi, gamma, explorer, batch_states, logger,
               burnin_action_func, minibatch_size, replay_start_size, update_interval,
               n_times_update, use_graphs):
        if gpu is not None and gpu >= 0:
            assert torch.cuda.is_available()
            self.device = torch.device("cuda:{}".format(gpu))
            from pfrl_amd.nn import accelerate_mlp

            for m in modules:
                m.to(self.device)
                # nn.Linear (+ ReLU) at minibatch size on the MFMA kernels; the target
                # networks are deep copies made after this and inherit it
                accelerate_mlp(m)
        else:
            self.device = torch.device("cpu")
        self.gpu = gpu
        self.replay_buffer = replay_buffer
        if hasattr(replay_buffer, "bind"):
            replay_buffer.bind(self.device, phi)
        self.phi = phi
        self.gamma = gamma
        self.explorer = explorer
        self.batch_states = batch_states
        self.logger = logger
        self.burnin_action_func = burnin_action_func
        self.replay_updater = ReplayUpdater(
            replay_buffer=replay_buffer, update_func=self.update, batchsize=minibatch_size,
            episodic_update=False, n_times_update=n_times_update,
            replay_start_size=replay_start_size, update_interval=update_interval)
        self.t = 0
        self.batch_last_obs = []
        self.batch_last_action = []
        self._records = {name: _DeviceRecord(win) for name, win in self._STATS}
        self._stat_sink = None
        from pfrl_amd import distributed

        on_gpu = (self.device.type == "cuda" and getattr(replay_buffer, "is_device", False)
                  and distributed.world_size() == 1)
        self.use_graphs = on_gpu if use_graphs is None else bool(use_graphs and on_gpu)
        self._captured = None

    # -- statistics ---------------------------------------------------------------
    def _stat(self, **tensors):
        if self._stat_sink is not None:
            self._stat_sink.update({k: v.detach() for k, v in tensors.items()})
        else:
            self._record_stats(tensors)

    def _record_stats(self, st):
        for name, t in st.items():
            self._records[name].extend(t)

    def _mean_stat(self, name):
        return _mean_or_nan(self._records[name].values())

    # -- hooks for subclasses -------------------------------------------------------
    def _policy(self):
        raise NotImplementedError

    def _burnin_over(self):
        raise NotImplementedError

    def _variant(self):
        """Hashable description of what the NEXT update will do (one graph each)."""
        return None

    def _update_impl(self, batch, variant):
        raise NotImplementedError

    def _after_update(self, variant):
        pass

    def _graph_modules(self):
        raise NotImplementedError

    def _graph_optimizers(self):
        raise NotImplementedError

    def _on_env_step(self):
        pass

    # -- learning -----------------------------------------------------------------------
    def _update_core(self, batch, variant=None):
        """The captured step: statistics come back as one flat device vector."""
        self._stat_sink = {}
        try:
            self._update_impl(batch, variant)
            sink = self._stat_sink
        finally:
            self._stat_sink = None
        names = sorted(sink)
        # (the statistics stay separate tensors inside the graph: packing them with a cat was one
        # launch per update -- 64 per env step for SAC -- and the caller packs a whole range with ONE)
        return {"parts": [sink[k].reshape(-1).float() for k in names],
                "names": names, "sizes": [sink[k].numel() for k in names]}

    def update(self, experiences, errors_out=None):
        batch = batch_experiences(experiences, self.device, self.phi, self.gamma)
        variant = self._variant()
        state = batch.get("state")
        if self.use_graphs and isinstance(state, torch.Tensor) and state.is_cuda:
            if self._captured is None:
                from pfrl_amd.agents.graphed_update import CapturedStep

                self._captured = CapturedStep(self._update_core, self._graph_modules(),
                                              self._graph_optimizers(), self.device)
            tensors = {k: v for k, v in batch.items() if isinstance(v, torch.Tensor)}
            try:
                out = self._captured.run(tensors, variant)
            except Exception:
                self.logger.exception("HIP-graph capture of the update failed; running eager")
                self.use_graphs = False
                self._captured = None
                return self.update(experiences, errors_out)
            flat = torch.cat(out["parts"])   # (new memory: the graph owns and overwrites its outputs)
            self._record_stats(dict(zip(out["names"], torch.split(flat, out["sizes"]))))
        else:
            self._update_impl(batch, variant)
        self._after_update(variant)

    # -- acting / observing -----------------------------------------------------------------
    def _batch_select_actions(self, batch_obs):
        with torch.no_grad(), evaluating(self._policy()):
            xs = self.batch_states(batch_obs, self.device, self.phi)
            return self._policy()(xs).sample().cpu().numpy()

    def batch_select_onpolicy_action(self, batch_obs):
        """The policy's own action for every observation, without exploration noise, as a list
        (reference td3.py:261-265, ddpg.py ``_batch_select_greedy_actions``)."""
        return list(self._batch_select_actions(batch_obs))

    def batch_act(self, batch_obs):
        if not self.training:
            return self._batch_select_actions(batch_obs)
        if self.burnin_action_func is not None and not self._burnin_over():
            actions = [self.burnin_action_func() for _ in range(len(batch_obs))]
        else:
            greedy = self._batch_select_actions(batch_obs)
            actions = [self.explorer.select_action(self.t, lambda i=i: greedy[i])
                       for i in range(len(greedy))]
        self.batch_last_obs = list(batch_obs)
        self.batch_last_action = list(actions)
        return actions

    # -- step-fused path (uniform device replay) ---------------------------------------------
    step_fused = True
    # fractions of the env batch at which the fused step is cut into ranges (see
    # _batch_observe_fused); () = one range
    step_fused_chunks = (0.125,)

    def _step_fusable(self):
        """All updates of one batched env step from ONE gather launch and ONE captured
        graph: the index sets of uniform replay depend only on len(buffer) and the NumPy
        stream, in order (the same argument as DQN._batch_observe_train_fused), and nothing
        else may have to happen on the host between the updates."""
        rbuf = self.replay_buffer
        return (self.step_fused and self.use_graphs
                and getattr(rbuf, "supports_lookahead", False)
                and type(self)._on_env_step is ReplayActorCritic._on_env_step)

    def _append(self, i, batch_obs, batch_reward, batch_done, batch_reset):
        rbuf = self.replay_buffer
        if self.batch_last_obs[i] is not None:
            assert self.batch_last_action[i] is not None
            rbuf.append(state=self.batch_last_obs[i], action=self.batch_last_action[i],
                        reward=batch_reward[i], next_state=batch_obs[i], next_action=None,
                        is_state_terminal=batch_done[i], env_id=i)
            if batch_reset[i] or batch_done[i]:
                self.batch_last_obs[i] = None
                self.batch_last_action[i] = None
                rbuf.stop_current_episode(env_id=i)

    def _batch_observe_fused(self, batch_obs, batch_reward, batch_done, batch_reset):
        """The per-env loop below (reference soft_actor_critic.py:354-374, td3.py:283-303)
        reorganised for the device: appends and index draws in the reference's order, one
        fused gather of every minibatch of the step (U * B entries: HBM-bound instead of U
        small launches), then the U updates replayed as one graph."""
        # A small env range first, the rest second: while the GPU runs the first range's
        # updates the host prepares the second (appends, index draws, gather and graph launch)
        # instead of the GPU idling through the whole preparation.  Appends, draws and updates
        # keep the reference's order.
        n_env = len(batch_obs)
        cuts = sorted({0, n_env} | {int(n_env * f) for f in self.step_fused_chunks})
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            self._observe_range_fused(lo, hi, batch_obs, batch_reward, batch_done, batch_reset)

    def _observe_range_fused(self, lo, hi, batch_obs, batch_reward, batch_done, batch_reset):
        rbuf, up = self.replay_buffer, self.replay_updater
        t0 = self.t
        from pfrl_amd.agents import _vector_device_step

        native = _vector_device_step.plan_range(self, lo, hi, batch_obs, batch_reward, batch_done,
                                                batch_reset)
        if native is not None:
            # appends, queue bookkeeping and every index set of the range from ONE planner call
            # and ONE transfer (agents/_vector_device_step.py); same NumPy stream as the loop
            self.t = t0 + (hi - lo)
            n_plan, slots_dev = native
            if n_plan == 0:
                return
            plan = [None] * n_plan
            big = rbuf.store.fetch_many_slots(slots_dev, n_plan, up.batchsize, self.phi,
                                              self.gamma, alternate=False)
        else:
            plan = []
            for i in range(lo, hi):
                self._append(i, batch_obs, batch_reward, batch_done, batch_reset)
                if len(rbuf) >= up.replay_start_size and (t0 + (i - lo) + 1) % up.update_interval == 0:
                    for _ in range(up.n_times_update):
                        plan.append(rbuf.lookahead_sample(up.batchsize))
            self.t = t0 + (hi - lo)
            if not plan:
                return
            big = rbuf.fetch_many(plan, self.phi, self.gamma)
        variants = []
        for _ in plan:          # host counters only: what each update is going to be
            v = self._variant()
            variants.append(v)
            self._after_update(v)
        if self._captured is None:
            from pfrl_amd.agents.graphed_update import CapturedStep

            self._captured = CapturedStep(self._update_core, self._graph_modules(),
                                          self._graph_optimizers(), self.device)
        tensors = {k: v for k, v in big.items() if isinstance(v, torch.Tensor)}
        try:
            outs = self._captured.run_range(tensors, variants)
        except Exception:
            # as in update(): a step that cannot be captured runs eagerly.  The variants (host
            # counters) of this range are already decided, so the same updates run one by one.
            if self._captured.graphs:
                raise
            self.logger.exception("HIP-graph capture of the env range failed; running eager")
            self.use_graphs = False
            self._captured = None
            for p, v in enumerate(variants):
                self._update_impl({k: t[p] for k, t in tensors.items()}, v)
            return
        # The graph owns (and overwrites) its outputs.  One stacked copy per run of updates
        # with the same statistics layout (64 separate clones were 64 x 32 us of host-paced
        # copies per step), recorded name by name in update order.
        i = 0
        while i < len(outs):
            j = i + 1
            while (j < len(outs) and outs[j]["names"] == outs[i]["names"]
                   and outs[j]["sizes"] == outs[i]["sizes"]):
                j += 1
            flat = torch.cat([p for o in outs[i:j] for p in o["parts"]]).view(j - i, -1)   # new memory
            off = 0
            st = {}
            for name, size in zip(outs[i]["names"], outs[i]["sizes"]):
                st[name] = flat[:, off:off + size].reshape(-1)
                off += size
            self._record_stats(st)
            i = j

    def batch_observe(self, batch_obs, batch_reward, batch_done, batch_reset):
        if not self.training:
            return
        if self._step_fusable():
            return self._batch_observe_fused(batch_obs, batch_reward, batch_done, batch_reset)
        rbuf = self.replay_buffer
        for i in range(len(batch_obs)):
            self.t += 1
            self._on_env_step()
            if self.batch_last_obs[i] is not None:
                assert self.batch_last_action[i] is not None
                rbuf.append(state=self.batch_last_obs[i], action=self.batch_last_action[i],
                            reward=batch_reward[i], next_state=batch_obs[i], next_action=None,
                            is_state_terminal=batch_done[i], env_id=i)
                if batch_reset[i] or batch_done[i]:
                    self.batch_last_obs[i] = None
                    self.batch_last_action[i] = None
                    rbuf.stop_current_episode(env_id=i)
            self.replay_updater.update_if_necessary(self.t)
