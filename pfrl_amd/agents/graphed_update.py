"""HIP-graph capture of one optimizer update.

The DQN update at minibatch 32 is ~150 tiny kernels (MIOpen convs, GEMMs,
elementwise, foreach optimizer): eager dispatch costs ~1.7 ms of host time for
~0.6 ms of device time.  Capturing loss -> backward -> (clip) -> optimizer.step
once and replaying it removes the dispatcher from the loop (hipGraphLaunch,
~15 us).  Capture is keyed by the *addresses* of the minibatch tensors: the
replay store writes minibatches into persistent HBM buffers, so the same key
recurs every step.

Numerics are unchanged: the graph replays exactly the kernels eager mode
launches.  The warm-up iterations PyTorch needs before capture run on a
snapshot that is restored afterwards, so no extra optimizer step leaks into
training.
"""
import collections
import gc
import logging
import os

import torch

from pfrl_amd import distributed


def _capture_kwargs(pool):
    """Arguments of ``torch.cuda.graph``: the shared memory pool, and thread-local capture error
    mode -- in the default global mode a HIP call from ANY other thread while this one captures
    (a process group's watchdog polling events, a runtime helper thread) aborts the process."""
    kw = {} if pool is None else {"pool": pool}
    kw["capture_error_mode"] = "thread_local"
    return kw


class _capturing:
    """``with _capturing(g, pool):`` -- ``torch.cuda.graph`` with the collector paused.  A
    collection that starts mid-capture runs finalizers on the capturing thread (graphs, pinned
    buffers and events of an agent that went out of scope): their HIP calls are not allowed in a
    capture, the error is thrown inside a destructor and the process aborts -- seen as a rare
    "Fatal Python error: Aborted ... Garbage-collecting" when several agents are built in one
    process.  ``torch.cuda.graph`` collects once on entry, before the capture begins."""

    def __init__(self, g, pool):
        self.ctx = torch.cuda.graph(g, **_capture_kwargs(pool))

    def __enter__(self):
        self.was_enabled = gc.isenabled()
        r = self.ctx.__enter__()
        gc.disable()
        return r

    def __exit__(self, *exc):
        try:
            return self.ctx.__exit__(*exc)
        finally:
            if self.was_enabled:
                gc.enable()


def _optimizer_tensors(optimizer):
    out = []
    for st in optimizer.state.values():
        for k, v in st.items():
            if isinstance(v, torch.Tensor):
                out.append((st, k, v))
    return out


def _make_capturable(optimizer, device):
    """Switch a stock torch optimizer to its graph-capturable code path."""
    ok = True
    for group in optimizer.param_groups:
        if "capturable" in group:
            group["capturable"] = True
        elif "foreach" in group or "fused" in group:
            # optimizers without a step counter dependency (SGD) capture as is
            pass
        else:
            ok = False
    for st in optimizer.state.values():
        step = st.get("step")
        if isinstance(step, torch.Tensor) and step.device != device:
            st["step"] = step.to(device)
    return ok


class _GraphCache(collections.OrderedDict):
    """Captured graphs by key, least recently used first.  When ``max_graphs`` entries exist the
    oldest is dropped to admit a new one (its graph is released; the shared pool keeps the
    memory for the next capture) instead of failing the update."""

    def __init__(self, max_graphs):
        super().__init__()
        self.max_graphs = max_graphs

    def lookup(self, key):
        entry = self.get(key)
        if entry is not None:
            self.move_to_end(key)
        return entry

    def admit(self, key, entry):
        while len(self) >= self.max_graphs:
            self.popitem(last=False)
        self[key] = entry


class _DeviceLR:
    """Learning rates of stock ``torch.optim.Adam`` optimizers held in DEVICE scalars while a step
    is captured, so that the graph reads them from memory instead of baking them in: a schedule
    that moves ``param_group["lr"]`` every step (the reference's ``LinearInterpolationHook`` on the
    learning rate, examples/atari/train_ppo_ale.py:307-311) then costs one scalar fill per change
    instead of a new capture per rollout.  Users keep seeing Python floats in
    ``param_groups``: the tensors are installed around warm-up / capture only."""

    def __init__(self, optimizers, device):
        self.slots = []
        self.ok = bool(optimizers)
        for opt in optimizers:
            if type(opt) is not torch.optim.Adam:
                self.ok = False
                return
            for g in opt.param_groups:
                if isinstance(g["lr"], torch.Tensor) or g.get("amsgrad") or not g.get("fused"):
                    self.ok = False
                    return
                self.slots.append([g, torch.tensor(float(g["lr"]), dtype=torch.float32, device=device),
                                   float(g["lr"])])

    def install(self):
        for g, t, last in self.slots:
            g["lr"] = t

    def restore(self):
        for g, t, last in self.slots:
            g["lr"] = last

    def sync(self):
        """Before a replay: whatever a hook wrote into the groups since goes to the device."""
        for slot in self.slots:
            g, t, last = slot
            v = float(g["lr"])
            if v != last:
                t.fill_(v)
                slot[2] = v


def _hyper_signature(optimizers, skip=()):
    """What a captured optimizer launch bakes in as kernel arguments: the Python-number
    hyperparameters of every parameter group.  Part of the graph key, so that a changed
    learning rate (a schedule hook) captures anew instead of replaying the old value."""
    sig = []
    for opt in optimizers:
        if opt is None:
            continue
        for g in opt.param_groups:
            # (flags such as `capturable` are switched by the capture itself: numbers only)
            sig.append(tuple((k, v if isinstance(v, (int, float, tuple)) else id(v))
                             for k, v in sorted(g.items())
                             if k != "params" and k not in skip and not isinstance(v, bool)
                             and isinstance(v, (int, float, tuple, torch.Tensor))))
    return tuple(sig)


class GraphedUpdate:
    """Caches one captured graph per minibatch-buffer address set."""

    def __init__(self, agent, max_graphs=256):
        self.agent = agent
        self.graphs = _GraphCache(max_graphs)
        self.pool = None
        self.enabled = True
        self.max_graphs = max_graphs
        self.logger = logging.getLogger(__name__)
        self._capturable_done = False
        self.split_for_allreduce = (distributed.world_size() > 1
                                    or os.environ.get("PFRL_FORCE_SPLIT_GRAPH") == "1")
        # pipeline=True: the forward/loss part is its own graph, so that the caller
        # can hand the TD errors to the replay stream (priority update, next
        # sample, next gather) while backward + optimizer step still run here
        self.pipeline = False
        # RCCL collectives are capturable: with the nccl backend they go INTO the graph (one
        # replay per update -- or per env range -- and no eager launches in between; the early
        # all-reduce / low-rank all-gather of the large layer run on RCCL's stream beside the
        # convolution backward).  Default for a real process group, but only after
        # distributed.captured_collectives_work() has captured, replayed and checked a small
        # all-reduce on this group under a timeout and every rank has agreed; otherwise, and on
        # any capture error, the split plan: graph -> eager collective -> graph.
        # PFRL_GRAPH_COLLECTIVE=0 forces the split plan, =1 skips the probe.
        self.graph_collective = os.environ.get("PFRL_GRAPH_COLLECTIVE", "auto")

    def _key(self, exp_batch):
        return (tuple(sorted((k, v.data_ptr(), tuple(v.shape)) for k, v in exp_batch.items()
                             if isinstance(v, torch.Tensor))),
                _hyper_signature([self.agent.optimizer]))

    # the work that gets captured ------------------------------------------------
    def _forward(self, exp_batch, want_errors):
        ag = self.agent
        ag._analytic_backward = None
        # (pipeline mode hands the loss outputs over between forward and backward: no deferral)
        ag._defer_head_fold = not self.pipeline
        from pfrl_amd.nn import mfma_trunk

        del mfma_trunk._DEFERRED_FOLDS[:]      # (leftovers of an update that raised)
        try:
            return ag._compute_loss(exp_batch, want_errors=want_errors, record=False)
        finally:
            ag._defer_head_fold = False

    def _optimizer_finishes_gradients(self):
        """The optimizer step takes split-K slabs and the hidden layer's batch matrices as they
        are (FusedRMSprop.step_from_sources, csrc/optim.hip k_rmsprop_fused): one launch instead
        of fold + step, and the hidden layer's weight gradient is never materialised.  Only when
        nothing else reads the gradients: no clipping, no collective, one parameter group."""
        ag = self.agent
        opt = ag.optimizer
        # (data parallel: the slabs are folded into the flat bucket instead -- pack_sources)
        return (os.environ.get("PFRL_FUSED_OPT", "1") != "0" and ag.max_grad_norm is None
                and not self.pipeline
                and hasattr(opt, "step_from_sources") and opt.accepts_sources()
                and self._optimizer_owns_every_parameter()
                and (not self.split_for_allreduce
                     or os.environ.get("PFRL_DP_SOURCES", "1") != "0"))

    def _optimizer_owns_every_parameter(self):
        """Sources are keyed by parameter: a trainable parameter outside the optimizer's groups
        (a frozen-by-omission layer, a partial optimizer) would have nobody to hand its slabs to."""
        ag = self.agent
        owned = {p.data_ptr() for g in ag.optimizer.param_groups for p in g["params"]}
        return all(p.data_ptr() in owned for p in ag.model.parameters() if p.requires_grad)

    def _backward(self, loss):
        ab = getattr(self.agent, "_analytic_backward", None)
        self._sources, self._folds = None, ()
        if ab is not None and ab[1] is not None:
            # fused TD loss: the gradient w.r.t. Q(s) (or w.r.t. the head's input, with the
            # head's own gradients beside it) came out of the same launch
            from pfrl_amd.nn import mfma_trunk

            head = ab[2] if len(ab) > 2 else ()
            defer = bool(head) and bool(mfma_trunk._DEFERRED_FOLDS) and \
                self._optimizer_finishes_gradients()
            red = self.agent.grad_reducer
            if defer:
                mfma_trunk.OPT_SOURCES = {}
                mfma_trunk.RIDE_ALONG = self.agent.optimizer
                # (the head's per-row partials may ride in a backward launch too: which parameter
                # each of the queued folds belongs to)
                mfma_trunk.RIDE_HEAD = {g.data_ptr(): p for p, g in head if p.requires_grad}
                # data parallel: a layer whose gradient is exchanged as its batch matrices is
                # stepped where the product is formed, on the communicator's side stream
                red.lowrank_step = self._step_on_side_stream
            try:
                if ab[0].requires_grad:
                    torch.autograd.backward([ab[0]], [ab[1]])
                sources = mfma_trunk.OPT_SOURCES
            finally:
                mfma_trunk.OPT_SOURCES = None
                mfma_trunk.RIDE_ALONG = None
                mfma_trunk.RIDE_HEAD = None
                red.lowrank_step = None
            if defer and sources:
                # the head's per-row partials (queued by the loss launch for "the fold that ends
                # the trunk's backward") become sources / folds of the optimizer launch too
                from pfrl_amd.optimizers import GradSource

                queued = list(mfma_trunk._DEFERRED_FOLDS)
                del mfma_trunk._DEFERRED_FOLDS[:]
                by_out = {t[1].data_ptr(): t for t in queued}
                for p, g in head:
                    t = by_out.pop(g.data_ptr(), None)
                    if t is None:
                        continue        # (its step rode in a backward launch: marked done there)
                    sources[p.data_ptr()] = GradSource.slabs(t[0], t[3], t[5])
                self._folds = [(t[0], t[1], t[3], t[5]) for t in by_out.values()]
                self._sources = sources
                return
            mfma_trunk.flush_deferred_folds()     # (no-op when the trunk's backward took them)
            for p, g in head:
                if p.requires_grad:
                    p.grad = g if p.grad is None else p.grad + g
        else:
            loss.backward()

    def _step_on_side_stream(self, weight, bias, dw, db):
        from pfrl_amd.nn import mfma_trunk
        from pfrl_amd.optimizers import GradSource

        if os.environ.get("PFRL_DP_SIDE_STEP", "1") == "0" or mfma_trunk.OPT_SOURCES is None:
            return False
        pairs = [(weight, dw)] + ([(bias, db)] if bias is not None else [])
        if not self.agent.optimizer.step_pairs(pairs):
            return False
        for p, _ in pairs:
            mfma_trunk.OPT_SOURCES[p.data_ptr()] = GradSource.done()
        return True

    def _forward_backward(self, exp_batch, want_errors):
        loss, delta = self._forward(exp_batch, want_errors)
        self._backward(loss)
        return loss, delta

    _sources, _folds = None, ()

    def _sources_by_param(self):
        ag = self.agent
        by_ptr = {p.data_ptr(): p for g in ag.optimizer.param_groups for p in g["params"]}
        return {by_ptr[ptr]: src for ptr, src in self._sources.items()}

    def _reduce(self, part):
        """The data-parallel exchange of one update, in the three parts the plans place inside or
        between their graphs: "pack" (gradients / slabs -> flat bucket), "collective" (the flat
        all-reduce), "finish" (join the early exchanges; flat -> gradients unless they alias)."""
        red = self.agent.grad_reducer
        if not red.active():
            return
        if self._sources is not None:
            if part == "pack":
                self._sources = {p.data_ptr(): s for p, s in self._pack_sources(red).items()}
            elif part == "collective":
                red.reduce_flat()
            else:
                red.finish_sources()
        elif part == "pack":
            red.pack()
        elif part == "collective":
            red.reduce_flat()
        else:
            red.unpack()

    def _pack_sources(self, red):
        sources = self._sources_by_param()
        red.pack_sources(sources)
        return sources

    def _step(self):
        ag = self.agent
        if self._sources is not None:
            sources = self._sources_by_param()
            self._sources = None
            ag.optimizer.step_from_sources(sources, self._folds)
            self._folds = ()
            return
        if ag.max_grad_norm is not None:
            torch.nn.utils.clip_grad_norm_(ag.model.parameters(), float(ag.max_grad_norm))
        ag.optimizer.step()

    def _reduce_and_step(self):
        self._reduce("pack")
        self._reduce("collective")
        self._reduce("finish")
        self._step()

    def _snapshot(self):
        ag = self.agent
        params = [p.detach().clone() for p in ag.model.parameters()]
        bufs = [b.detach().clone() for b in ag.model.buffers()]
        had_state = len(ag.optimizer.state) > 0
        opt = [(st, k, v.detach().clone()) for st, k, v in _optimizer_tensors(ag.optimizer)]
        rng = torch.cuda.get_rng_state(ag.device)
        return params, bufs, had_state, opt, rng

    def _restore(self, snap):
        ag = self.agent
        params, bufs, had_state, opt, rng = snap
        with torch.no_grad():
            for p, s in zip(ag.model.parameters(), params):
                p.copy_(s)
            for b, s in zip(ag.model.buffers(), bufs):
                b.copy_(s)
            if had_state:
                for st, k, v in opt:
                    st[k].copy_(v)
            else:
                # state was created by the warm-up: reset it to its initial value
                for st, k, v in _optimizer_tensors(ag.optimizer):
                    v.zero_()
        torch.cuda.set_rng_state(rng, ag.device)

    # -- the NoisyNet draws of an update as ONE launch in front of its graph ---------------------
    def _noise_feed_ok(self):
        """The update's networks hold factorised-noise layers on the GPU and csrc/philox.hip
        reproduces this PyTorch build's torch.randn bit for bit (probed): the draws then leave the
        captured graph -- one eager launch per replay writes all of them, in the layers' call
        order, from the device generator (PFRL_NOISE_FEED=0: torch.randn calls inside the graph)."""
        if os.environ.get("PFRL_NOISE_FEED", "1") == "0" or self.agent.device.type != "cuda":
            return False
        from pfrl_amd import ops
        from pfrl_amd.nn.noisy_linear import FactorizedNoisyLinear

        if not any(isinstance(m, FactorizedNoisyLinear) for m in self.agent.model.modules()):
            return False
        return ops.philox_variant(self.agent.device) is not None

    def _noise_buffers(self, sizes, repeat=1):
        sizes = list(sizes) * repeat
        offs, pos = [], 0
        for n in sizes:
            offs.append(pos)
            pos += (n + 3) & ~3
        from pfrl_amd import ops

        buf = torch.zeros(pos, dtype=torch.float32, device=self.agent.device)
        plan = ops.RandnPlan(sizes, self.agent.device, out=buf)      # (argument arrays built once)
        return {"sizes": sizes, "buf": buf, "views": plan.views, "plan": plan}

    def _fill_noise(self, entry):
        noise = entry.get("noise")
        if noise is not None:
            if noise.pop("filled", False):
                return              # (prefill() has drawn this replay's noise already)
            noise["plan"].run()

    def prefill(self, exp_batch, want_errors):
        """Draw the noise of the NEXT ``run(exp_batch, want_errors)`` now: a caller that is about
        to make its stream wait for the minibatch (the replay stream's gather) enqueues the draws
        in front of that wait -- they depend on nothing but the generator.  The host order of
        generator consumers is unchanged (nothing may draw between this call and that run)."""
        entry = self.graphs.lookup((self._key(exp_batch), bool(want_errors)))
        if entry is not None and entry.get("noise") is not None and not entry["noise"].get("filled"):
            entry["noise"]["plan"].run()
            entry["noise"]["filled"] = True

    def _collective_capturable(self):
        d = torch.distributed
        if not (d.is_available() and d.is_initialized()):
            return False
        if self.agent.grad_reducer._comm is None and d.get_backend() != "nccl":
            return False            # (gloo on host buffers: nothing a HIP graph could hold)
        if self.graph_collective == "auto":
            # decided once, by a probe every rank takes part in (never from inside a capture)
            self.graph_collective = "1" if distributed.captured_collectives_work(
                self.agent.device) else "0"
        return self.graph_collective not in ("0", False)

    def _capture(self, exp_batch, want_errors):
        if self.split_for_allreduce and self._collective_capturable():
            try:
                return self._capture_plan(exp_batch, want_errors, collective_in_graph=True)
            except Exception as e:
                self.logger.warning("capturing the RCCL all-reduce inside the update graph "
                                    "failed (%s); keeping it eager between two graphs", e)
                self.graph_collective = "0"
        return self._capture_plan(exp_batch, want_errors, collective_in_graph=False)

    def _capture_plan(self, exp_batch, want_errors, collective_in_graph):
        ag = self.agent
        dev = ag.device
        if not self._capturable_done:
            if not _make_capturable(ag.optimizer, dev):
                raise RuntimeError("optimizer %s has no capturable mode" % type(ag.optimizer))
            self._capturable_done = True
        snap = self._snapshot()
        try:
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        except AttributeError:
            pass
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(dev)
        side.wait_stream(cur)
        from pfrl_amd.nn.noisy_linear import NoiseFeed, noise_feed

        noise = None
        rec = NoiseFeed() if self._noise_feed_ok() else None
        try:
            with torch.cuda.stream(side):
                for it in range(2):
                    if it == 1 and rec is not None and rec.sizes:
                        # the first warm-up drew through torch.randn and noted the sizes: from here
                        # on the layers read views of one static buffer (filled once for the rest of
                        # the warm-up; the RNG state is restored below either way)
                        noise = self._noise_buffers(rec.sizes)
                        self._fill_noise({"noise": noise})
                    feed = rec if it == 0 else (NoiseFeed(noise["views"]) if noise else None)
                    with noise_feed(feed):
                        ag.optimizer.zero_grad(set_to_none=True)
                        self._forward_backward(exp_batch, want_errors)
                        self._reduce_and_step()
            cur.wait_stream(side)
            _make_capturable(ag.optimizer, dev)  # state created by the warm-up
            ag.optimizer.zero_grad(set_to_none=True)
            distributed._CAPTURE_COLLECTIVES[0] = bool(collective_in_graph)
            try:
                with noise_feed(NoiseFeed(noise["views"]) if noise else None):
                    entry = self._capture_graphs(exp_batch, want_errors, collective_in_graph)
                entry["noise"] = noise
            finally:
                distributed._CAPTURE_COLLECTIVES[0] = False
        finally:
            cur.wait_stream(side)
            self._restore(snap)
        return entry

    def _capture_graphs(self, exp_batch, want_errors, collective_in_graph):
        ag = self.agent
        entry = {}

        def graph_of(fn):
            g = torch.cuda.CUDAGraph()
            with _capturing(g, self.pool):
                r = fn()
            if self.pool is None:
                self.pool = g.pool()
            return g, r

        # plan: graphs interleaved with the two things that cannot be captured --
        # the caller's hand-over after the forward pass ("after_forward", pipeline
        # mode) and the eager RCCL all-reduce (data parallel); pack / finish (the fold into
        # the flat bucket, the low-rank product) live inside the graphs
        reduce_and_step = self._reduce_and_step

        plan = []
        if self.pipeline:
            g, (loss, delta) = graph_of(lambda: self._forward(exp_batch, want_errors))
            plan += [g, "after_forward"]
            if collective_in_graph:
                plan += [graph_of(lambda: (self._backward(loss), reduce_and_step()))[0]]
            elif self.split_for_allreduce:
                plan += [graph_of(lambda: (self._backward(loss), self._reduce("pack")))[0],
                         "all_reduce",
                         graph_of(lambda: (self._reduce("finish"), self._step()))[0]]
            else:
                plan += [graph_of(lambda: (self._backward(loss), self._step()))[0]]
        elif collective_in_graph:
            def whole_dp():
                r = self._forward_backward(exp_batch, want_errors)
                reduce_and_step()
                return r

            g, (loss, delta) = graph_of(whole_dp)
            plan += [g]
        elif self.split_for_allreduce:
            # data parallel: graph(fwd+bwd) -> eager RCCL all-reduce -> graph(step)
            def fwd_bwd_pack():
                r = self._forward_backward(exp_batch, want_errors)
                self._reduce("pack")
                return r

            g, (loss, delta) = graph_of(fwd_bwd_pack)
            plan += [g, "all_reduce",
                     graph_of(lambda: (self._reduce("finish"), self._step()))[0]]
        else:
            def whole():
                r = self._forward_backward(exp_batch, want_errors)
                self._step()
                return r

            g, (loss, delta) = graph_of(whole)
            plan += [g]
        entry["plan"] = plan
        entry["bucket"] = ag.grad_reducer.current_bucket()   # what the eager collective reduces
        entry["loss"] = loss
        entry["delta"] = delta
        entry["y"] = ag._last_y
        return entry

    # -- a whole env range as ONE graph ---------------------------------------------
    # set to a list by bench.py: (updates in the range, start event, end event) per replay
    time_ranges = None

    def range_capturable(self):
        """Several consecutive updates can share one graph when nothing has to happen
        between them on the host: no eager collective, no hand-over to a replay stream."""
        return not self.pipeline and (not self.split_for_allreduce or self._collective_capturable())

    def measure_launches(self, exp_batch, repeats=5):
        """The launches of ONE update -- exactly the Python a capture records (``_forward_backward``
        + ``_reduce_and_step``) run eagerly on ``exp_batch`` -- each bracketed by timing events
        (``_native.timed_calls``): [(entry point, median microseconds over ``repeats``)] in launch
        order.  Parameters, optimizer state and generator are restored afterwards."""
        from pfrl_amd import _native

        ag = self.agent
        snap = self._snapshot()
        runs = []
        try:
            for _ in range(repeats + 1):
                with _native.timed_calls() as rec:
                    ag.optimizer.zero_grad(set_to_none=True)
                    self._forward_backward(exp_batch, False)
                    self._reduce_and_step()
                runs.append(rec.results())
        finally:
            ag.optimizer.zero_grad(set_to_none=True)
            self._restore(snap)
        runs = [r for r in runs[1:] if [n for n, _ in r] == [n for n, _ in runs[-1]]]
        out = []
        for i, (name, _) in enumerate(runs[-1]):
            us = sorted(r[i][1] for r in runs)
            out.append((name, us[len(us) // 2]))
        return out

    def run_range(self, big):
        """``big``: dict of tensors with a leading update axis U (the step-fused gather's
        buffers).  Replays ONE graph holding the U updates back to back (each: forward,
        loss, backward, clip, optimizer step -- the same launches as U single replays,
        without U - 1 graph launches and the idle gaps between them).  Returns
        (losses [U], ys [U * B]) owned by the graph."""
        key = ("range", self._key(big))
        entry = self.graphs.lookup(key)
        if entry is None:
            entry = self._capture_range(big)
            self.graphs.admit(key, entry)
        self._fill_noise(entry)
        timing = self.time_ranges
        if timing is not None:
            # bench.py: device time of a whole range graph (events on the launch stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            entry["graph"].replay()
            e1.record()
            timing.append((int(entry["losses"].shape[0]), e0, e1))
        else:
            entry["graph"].replay()
        return entry["losses"], entry["ys"]

    def _capture_range(self, big):
        ag = self.agent
        dev = ag.device
        U = next(iter(big.values())).shape[0]
        if not self._capturable_done:
            if not _make_capturable(ag.optimizer, dev):
                raise RuntimeError("optimizer %s has no capturable mode" % type(ag.optimizer))
            self._capturable_done = True

        def body():
            losses, ys = [], []
            for p in range(U):
                ag.optimizer.zero_grad(set_to_none=True)
                loss, _ = self._forward_backward({k: v[p] for k, v in big.items()}, False)
                self._reduce_and_step()
                losses.append(loss.reshape(()))
                ys.append(ag._last_y.reshape(-1))
            return torch.stack(losses), torch.cat(ys)

        snap = self._snapshot()
        try:
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        except AttributeError:
            pass
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(dev)
        side.wait_stream(cur)
        from pfrl_amd.nn.noisy_linear import NoiseFeed, noise_feed

        noise = None
        rec = NoiseFeed() if self._noise_feed_ok() else None
        try:
            with torch.cuda.stream(side):
                # warm-up on the first update of the range only (same kernels for all U)
                for it in range(2):
                    if it == 1 and rec is not None and rec.sizes:
                        noise = self._noise_buffers(rec.sizes, repeat=U)    # (every update draws anew)
                        self._fill_noise({"noise": noise})
                    feed = rec if it == 0 else (NoiseFeed(noise["views"]) if noise else None)
                    with noise_feed(feed):
                        ag.optimizer.zero_grad(set_to_none=True)
                        self._forward_backward({k: v[0] for k, v in big.items()}, False)
                        self._reduce_and_step()
            cur.wait_stream(side)
            _make_capturable(ag.optimizer, dev)
            ag.optimizer.zero_grad(set_to_none=True)
            g = torch.cuda.CUDAGraph()
            distributed._CAPTURE_COLLECTIVES[0] = self.split_for_allreduce
            cap_feed = NoiseFeed(noise["views"]) if noise else None
            try:
                with _capturing(g, self.pool), noise_feed(cap_feed):
                    losses, ys = body()
            finally:
                distributed._CAPTURE_COLLECTIVES[0] = False
            # (the captured range must have consumed exactly the draws recorded x U)
            assert cap_feed is None or cap_feed.at == len(noise["views"]), \
                "the captured updates drew %d normals' tensors, %d were recorded" % (
                    cap_feed.at, len(noise["views"]))
            if self.pool is None:
                self.pool = g.pool()
        finally:
            cur.wait_stream(side)
            self._restore(snap)
        return {"graph": g, "losses": losses, "ys": ys, "noise": noise}

    def _replay_items(self, items, bucket):
        for item in items:
            if item == "all_reduce":
                self.agent.grad_reducer.reduce_flat(bucket)
            elif item != "after_forward":
                item.replay()

    def run(self, exp_batch, want_errors, after_forward=None, late=None):
        """Returns (loss, delta, y) tensors owned by the graph (static).  In
        pipeline mode ``after_forward(delta)`` is called between the forward
        graph and the backward/step graph.  ``late``: a list that receives a
        callable replaying everything after the hand-over, instead of it being
        replayed here -- the caller launches the next minibatch's replay-side work
        first (that chain, not backward + step, is what the next forward pass waits
        for) and must call it before anything else touches the model."""
        key = (self._key(exp_batch), bool(want_errors))
        entry = self.graphs.lookup(key)
        if entry is None:
            entry = self._capture(exp_batch, want_errors)
            self.graphs.admit(key, entry)
        plan = entry["plan"]
        self._fill_noise(entry)
        if after_forward is not None and "after_forward" in plan:
            cut = plan.index("after_forward")
            self._replay_items(plan[:cut], entry["bucket"])
            after_forward(entry["delta"])
            rest, bucket = plan[cut + 1:], entry["bucket"]
            if late is not None:
                late.append(lambda: self._replay_items(rest, bucket))
            else:
                self._replay_items(rest, bucket)
        else:
            self._replay_items(plan, entry["bucket"])
            if after_forward is not None:
                after_forward(entry["delta"])
        return entry["loss"], entry["delta"], entry["y"]


class _no_distribution_validation:
    """torch.distributions validates constructor arguments and samples with a
    blocking ``bool(tensor)``: a D2H sync per distribution, and illegal while a
    stream is capturing.  Inside a captured step the check is switched off (it
    can only ever raise, never change a result).

    ``Normal.sample`` is ``torch.normal(loc, scale)``, which checks ``scale.min()
    >= 0`` on the host; ATen then computes ``normal_(0, 1) * scale + loc``.  The
    stand-in below draws the same ``normal_`` and applies the same two roundings
    without the host check."""

    def __enter__(self):
        from torch.distributions import Distribution, Normal
        from torch.distributions.utils import _standard_normal

        self.saved = (Distribution._validate_args, Normal.sample)
        Distribution.set_default_validate_args(False)

        def sample(dist, sample_shape=torch.Size()):
            shape = dist._extended_shape(sample_shape)
            with torch.no_grad():
                eps = _standard_normal(shape, dtype=dist.loc.dtype, device=dist.loc.device)
                return eps * dist.scale.expand(shape) + dist.loc.expand(shape)

        if Normal.sample.__module__ == "torch.distributions.normal":
            Normal.sample = sample   # leave a test's own stand-in alone

    def __exit__(self, *exc):
        from torch.distributions import Distribution, Normal

        Distribution.set_default_validate_args(self.saved[0])
        Normal.sample = self.saved[1]


class CapturedStep:
    """HIP-graph capture of an arbitrary training step ``fn(batch) -> dict of
    tensors`` that touches several modules / optimizers (SAC: two Q updates, the
    policy update, the temperature update and the soft target sync).

    Same contract as :class:`GraphedUpdate`: one graph per set of minibatch
    buffer addresses, warm-up on a snapshot that is restored (parameters,
    buffers, optimizer state, RNG), stock optimizers switched to their
    capturable code path.  The returned tensors are owned by the graph.
    """

    def __init__(self, fn, modules, optimizers, device, max_graphs=64, lr_on_device=False):
        self.fn = fn
        self.modules = [m for m in modules if m is not None]
        self.optimizers = [o for o in optimizers if o is not None]
        self.device = device
        self.graphs = _GraphCache(max_graphs)
        self.pool = None
        self.max_graphs = max_graphs
        self.device_lr = None
        if lr_on_device and os.environ.get("PFRL_LR_ON_DEVICE", "1") != "0":
            d = _DeviceLR(self.optimizers, device)
            self.device_lr = d if d.ok else None

    @staticmethod
    def _key(batch):
        return tuple(sorted((k, v.data_ptr(), tuple(v.shape)) for k, v in batch.items()
                            if isinstance(v, torch.Tensor)))

    def _tensors(self):
        out = []
        for m in self.modules:
            out.extend(p for p in m.parameters())
            out.extend(b for b in m.buffers())
        return out

    def _capture(self, batch, variant=None, step=None, warm=None, repeat=1, prewarm=None):
        dev = self.device
        if step is None:
            step = (lambda: self.fn(batch)) if variant is None else (lambda: self.fn(batch, variant))
        if warm is None:
            warm = step
        for opt in self.optimizers:
            if not _make_capturable(opt, dev):
                raise RuntimeError("optimizer %s has no capturable mode" % type(opt))
        saved = [t.detach().clone() for t in self._tensors()]
        had_state = [len(o.state) > 0 for o in self.optimizers]
        opt_saved = [[(st, k, v.detach().clone()) for st, k, v in _optimizer_tensors(o)]
                     for o in self.optimizers]
        rng = torch.cuda.get_rng_state(dev)
        try:
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        except AttributeError:
            pass

        def restore():
            # undo the warm-up steps (also when the capture itself fails)
            with torch.no_grad():
                for t, s in zip(self._tensors(), saved):
                    t.copy_(s)
                for o, had, sv in zip(self.optimizers, had_state, opt_saved):
                    if had:
                        for st, k, v in sv:
                            st[k].copy_(v)
                    else:
                        for st, k, v in _optimizer_tensors(o):
                            v.zero_()
            torch.cuda.set_rng_state(rng, dev)

        from pfrl_amd.nn.noisy_linear import NoiseFeed, noise_feed

        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(dev)
        side.wait_stream(cur)
        noise = None
        rec = NoiseFeed() if self._noise_feed_ok() else None
        if self.device_lr is not None:
            self.device_lr.sync()
            self.device_lr.install()
        try:
            with torch.cuda.stream(side), _no_distribution_validation():
                if prewarm is not None:
                    prewarm()
                # first warm-up: the step draws its normals as usual and their sizes are noted;
                # from then on they are views of one buffer that ONE launch fills per replay
                with noise_feed(rec):
                    warm()
                if rec is not None and rec.sizes:
                    noise = self._noise_buffers(rec.sizes, repeat)
                    noise["plan"].run()
                with noise_feed(NoiseFeed(noise["views"]) if noise else None):
                    warm()
            cur.wait_stream(side)
            for opt in self.optimizers:
                _make_capturable(opt, dev)  # state created by the warm-up
                opt.zero_grad(set_to_none=True)
            g = torch.cuda.CUDAGraph()
            with _capturing(g, self.pool), _no_distribution_validation(), \
                    noise_feed(NoiseFeed(noise["views"]) if noise else None):
                out = step()
            if self.pool is None:
                self.pool = g.pool()
        finally:
            cur.wait_stream(side)
            restore()
            if self.device_lr is not None:
                self.device_lr.restore()
        return g, out, noise

    def _noise_feed_ok(self):
        if os.environ.get("PFRL_NOISE_FEED", "1") == "0" or torch.device(self.device).type != "cuda":
            return False
        from pfrl_amd import ops

        return ops.philox_variant(self.device) is not None

    def _noise_buffers(self, sizes, repeat):
        from pfrl_amd import ops

        sizes = list(sizes) * repeat
        pos = sum((n + 3) & ~3 for n in sizes)
        buf = torch.zeros(pos, dtype=torch.float32, device=self.device)
        plan = ops.RandnPlan(sizes, self.device, out=buf)
        return {"sizes": sizes, "buf": buf, "views": plan.views, "plan": plan}

    def run_range(self, big, variants):
        """``big``: dict of tensors with a leading update axis U (one fused gather for all
        updates of a batched env step); ``variants``: the U variants in order.  ONE graph
        holds the U steps back to back.  Returns the list of the U ``fn`` results (tensors
        owned by the graph)."""
        U = len(variants)
        key = ("range", self._key(big), tuple(variants), _hyper_signature(self.optimizers))
        entry = self.graphs.lookup(key)
        if entry is None:

            def slice_of(p):
                return {k: v[p] for k, v in big.items()}

            def call(p):
                v = variants[p]
                return self.fn(slice_of(p)) if v is None else self.fn(slice_of(p), v)

            # warm-up: the first update of EACH variant in the range (TD3's delayed policy
            # update only runs in every second one; lazily created state of the modules it
            # alone reaches -- e.g. BoundByTanh's bounds on the device -- must exist before
            # the capture begins, where a host->device copy is not permitted)
            others = [variants.index(v) for v in dict.fromkeys(variants)][1:]
            entry = self._capture(None, None, step=lambda: [call(p) for p in range(U)],
                                  warm=lambda: call(0), repeat=U,
                                  prewarm=(lambda: [call(p) for p in others]) if others else None)
            self.graphs.admit(key, entry)
        if entry[2] is not None:
            entry[2]["plan"].run()          # every normal of the U steps: one launch per 16 draws
        entry[0].replay()
        return entry[1]

    def run(self, batch, variant=None, baked=None):
        """``variant`` (hashable) selects between differently shaped steps over the
        same buffers (TD3: with / without the delayed policy update); it is passed
        to ``fn`` as a second argument when not None.  ``baked`` (hashable): every Python-side
        number ``fn`` reads while it is captured (loss coefficients, clip ranges, ...) -- they
        become kernel arguments of the graph, so a changed value must capture anew."""
        skip = ("lr",) if self.device_lr is not None else ()
        key = (self._key(batch), variant, _hyper_signature(self.optimizers, skip), baked)
        entry = self.graphs.lookup(key)
        if entry is None:
            entry = self._capture(batch, variant)
            self.graphs.admit(key, entry)
        if self.device_lr is not None:
            self.device_lr.sync()
        if entry[2] is not None:
            entry[2]["plan"].run()
        entry[0].replay()
        return entry[1]
