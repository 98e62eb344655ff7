"""The training step as captured hipGraphs.

``KPCNInterface._forward_backward`` (two PathNet forwards, input assembly, KPCN forward, losses, both
backward passes) is ~2,400 kernel launches whose shapes never change during training
(``train_kpcn.py:45`` feeds fixed-size batches).  Eagerly the host needs ~40 ms to enqueue them from
Python -- four times as long as the MI355X needs to run them -- so the step is captured once with
``torch.cuda.graph`` (HIP stream capture) and replayed.  Two forms:

* ``two_stream=True`` (what ``capture_validated`` builds for the launcher and the benchmark): the diffuse and the specular
  half of the step as two linear graphs replayed on two streams that ``ops.concurrent_stream_pair`` has shown to sit on
  different hardware queues, between a head graph (step counter, the shared split of ``paths``) and a tail graph (radiance
  metrics, guard, clip + Adam).  The pairings of ``FeatureMSE(rng='device')`` are drawn inside the graphs.
* ``two_stream=False``: one forked graph (rounds 2-4); the overlap of its halves is up to hipGraphInstantiate.

What stays eager is what talks to the host or other ranks: reading the non-finite-loss flags (``interfaces.py:254-257``; one
step late with ``defer_check``), CPU-generator pairings (``rng='cpu'``) and the RCCL gradient all-reduces, around which the
multi-rank step is split into graph(s) A and graph B.
"""
import torch

from . import ops


class GraphedTrainStep:
    """``step = GraphedTrainStep(itf, example_batch); step(batch)`` == ``itf.preprocess(batch); itf.train_batch(batch)``."""

    def __init__(self, itf, batch, warmup=2, side_stream=True, capture_optimizer=True, defer_check=False, overlap_allreduce=False,
                 cut_backward=False, two_stream=False):
        """``two_stream`` (``train_branches`` steps): the diffuse and the specular half of the step -- PathNet, the branch's conv
        stack, kernel apply, losses, the whole backward: ``KPCNInterface._half_forward_backward`` -- are captured as TWO hipGraphs
        and replayed on two explicit streams, followed by a third graph with the recombined radiance's metrics and the optimiser
        tail.  The single forked graph leaves the overlap of the halves to the stream assignment hipGraphInstantiate makes for
        its parallel branches, which differs from capture to capture (round 4: 602-624 patches/s in two of four processes
        against 690); two linear graphs on two streams overlap by construction.  Results are bit-identical (same kernels, same
        order inside each half).

        ``defer_check`` (captured optimiser only): the non-finite-loss check of step t -- the step's one host sync -- is made
        after step t + 1 has been enqueued, so the host prepares the next batch while the GPU runs (a loader-fed loop gains
        what the sync-then-prepare gap cost).  The device guard still skips the update of a non-finite step at once; the
        reference's error (``interfaces.py:254-257``) is raised one call later, or by ``flush()``, which the epoch loop calls
        after its last step.

        ``overlap_allreduce`` (several ranks, PathNets in use; build ``FusedClipAdam(order=('dncnn', ...))``): the backward is cut at
        the P-buffers -- graph A1 ends when the gradients of ``dncnn`` are complete, their bucket's all-reduce is issued, graph A2
        (the PathNets' backward) runs while it is on the wire, then the PathNet buckets follow, then graph B.  SURVEY 8e's
        "bucketed in backward order, overlapped with the remaining backward"; bit-identical to the two-graph step
        (``tests/test_gpu_models.py::test_collective_branch_...``).  Off by default: what it hides (half of 46.8 MB over xGMI) has
        never been measured against what the extra graph boundary costs (both halves of the step join there).
        ``cut_backward``: the same cut inside ONE graph (an experiment switch: it buys nothing once both branch losses share one
        engine run, ``KPCNInterface._backward``)."""
        self.itf = itf
        self.static = {k: v.clone() for k, v in batch.items() if isinstance(v, torch.Tensor)}
        self.keys = list(self.static)                     # (PathNet stashes a converted copy of `paths` in the dict)
        self.fm = itf.loss_funcs.get('l_manif') if itf.manif_learn and itf.train_branches else None
        if self.fm is not None:
            self.fm.static_perms = None                   # (a re-capture on the same interface starts like a first one)
        dev = next(iter(self.static.values())).device
        cur = torch.cuda.current_stream()
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(cur)
        with torch.cuda.stream(s):                        # warm-up off the default stream (torch.cuda.graph rule)
            for _ in range(warmup):
                self.static.pop('_wcmc_paths_nhwc', None)
                warm = itf._forward_backward(self.static)
        cur.wait_stream(s)
        torch.cuda.synchronize()
        if self.fm is not None:
            # the step calls FeatureMSE twice (diffuse, specular): two static pairs of device permutations
            ip, ib = self.fm.last_perms
            self.perm_sizes = (ip.numel(), ib.numel() if ib is not None else 0)
            self.perms = [(torch.empty(self.perm_sizes[0], dtype=torch.int64, device=dev),
                           torch.empty(self.perm_sizes[1], dtype=torch.int64, device=dev) if ib is not None else None)
                          for _ in range(2)]
            self._draw()
            self.fm.static_perms, self.fm._static_i, self.fm.check_finite = self.perms, 0, False
            # rng='device': the draws are PART of the captured step -- [seed, step counter] on the device, the counter advanced by the
            # graph's first node, every permutation keyed with (seed, counter, slot): no host draw, no eager launch between replays
            self.dev_keys = self.fm.rng == 'device'
            self.key_state = torch.zeros(2, dtype=torch.int64, device=dev)
            self._seeded = False
        side = ops.USE_SIDE_STREAM
        ops.USE_SIDE_STREAM = side and side_stream        # forked capture: wgrad branches run beside dgrad in the graph
        self.static.pop('_wcmc_paths_nhwc', None)
        # (Tried: the step as a SEQUENCE of three hipGraphs sharing one memory pool -- PathNet forwards | KPCN forward and losses |
        # backward passes -- so that the host launches the later segments while the GPU runs the first: hipGraphLaunch of
        # the ~410-node step costs the host 0.94 ms, scripts/diag_step_host.py.  Bit-identical, and no faster: 397.7 vs 400.9
        # patches/s -- the runtime already feeds the GPU while it is still submitting.)
        # One rank, fused optimiser: the step's tail -- finite check, loss sums, gradient gather, clip + Adam -- is captured too
        # (a dozen small launches the host used to enqueue behind its sync on the losses: 0.5-0.7 ms per step with an idle
        # GPU, profiles/r03_step_trace_gaps.txt).  The update sits behind a DEVICE guard (all losses finite); the host reads the
        # flags after the replay and raises the reference's error (interfaces.py:254-257) -- the update was then skipped.
        # Several ranks (or FusedClipAdam(force_collective=True)): the same tail as TWO captured pieces around the eager bucket
        # all-reduces -- graph A ends with the gradient gather and this rank's guard flag, graph B holds the global guard, the
        # loss sums and scale -> clip -> Adam (no RCCL kernel inside a capture; the host launches graph A, three collectives,
        # graph B and reads the flags)
        fo = getattr(itf, 'fused_optim', None)
        coll = fo is not None and getattr(fo, 'collective', fo.world > 1)
        self.tail_captured = (capture_optimizer and fo is not None and not coll and itf.grad_sync is None)
        self.tail_split = (capture_optimizer and coll and itf.grad_sync is None)
        self.overlap = bool(overlap_allreduce) and self.tail_split and itf.use_llpm_buf
        if self.overlap:
            assert next(iter(fo.flats)) == 'dncnn', "overlap_allreduce: build FusedClipAdam(order=('dncnn', ...)) -- its bucket goes first"
        self.cut = self.overlap or bool(cut_backward)
        self.two_stream = bool(two_stream)
        if self.two_stream:
            assert itf.halves_supported() and not self.cut, "two_stream: a train_branches step of sbmc.KPCN, no backward cut"
        self.defer_check = bool(defer_check) and (self.tail_captured or self.tail_split)
        self._pending, self._flag_bufs, self._n_calls = None, None, 0
        if self.tail_captured or self.tail_split:
            assert warmup >= 1
            fo.prepare_capture(itf.optims)                            # (adopts optimiser state loaded since construction: not in the capture)
            # 1 until a step's losses were non-finite, then 0 -- ANDed into every later guard -- until the host has raised the
            # reference's error: with defer_check step t + 1 is already enqueued when step t's flags are read, and it must not
            # update parameters the reference would never have stepped (interfaces.py:254-257 aborts before optim.step)
            self.ok = torch.ones(1, device=dev)
            self.sums = torch.zeros(len(warm), device=dev)            # one slot per loss key (the warm-up's result has them)
            self._sum_views = [self.sums[i] for i in range(self.sums.numel())]
        self.graph = torch.cuda.CUDAGraph()
        try:
            pool = self._capture_halves(dev) if self.two_stream else None
            # thread_local: a loader thread (support/loader.py: pinned staging buffers, device allocations, H2D copies on its own
            # stream) may allocate while this thread captures -- in the default 'global' mode a hipHostMalloc / hipMalloc from
            # ANY thread invalidates the capture
            with torch.cuda.graph(self.graph, pool=pool, capture_error_mode="thread_local"):
                if not self.two_stream and self.fm is not None and self.dev_keys:
                    ops.step_counter_advance(self.key_state)
                    self._draw_captured(0)
                    self._draw_captured(1)
                if self.two_stream:     # (the halves are graphs of their own: what is left is the radiance, its metrics and the tail)
                    self.losses = itf._finish_halves(self.static, self._half_out[0][0], self._half_out[1][0],
                                                     self._half_out[0][1], self._half_out[1][1])
                else:
                    self.losses = itf._forward_backward(self.static, cut=self.cut)
                if self.cut and not self.overlap:
                    itf._backward_stage2()                            # (same graph: the PathNets' backward as one more engine run)
                if self.tail_captured:
                    # finite flags, guard (AND the poison flag), poison update and the running sums of interfaces.py:263-267 (in
                    # place on one persistent tensor: itf.m_losses holds views) in ONE launch
                    self.loss_keys = list(self.losses)
                    fused_guard = all(v.is_cuda and v.dtype == torch.float32 and v.numel() == 1 for v in self.losses.values()) \
                        and len(self.loss_keys) <= 16
                    if fused_guard:
                        self._loss_refs = [self.losses[k].reshape(()) for k in self.loss_keys]
                        self.flags = torch.empty(len(self.loss_keys) + 1, device=dev)
                        ops.step_guard_(self._loss_refs, self.ok, self.sums, self.flags)
                        self.guard = self.flags[len(self.loss_keys):]
                    else:
                        vals = torch.stack([self.losses[k].reshape(()) for k in self.loss_keys])
                        finite = torch.isfinite(vals)
                        self.guard = finite.all().to(torch.float32).reshape(1) * self.ok
                        self.ok.copy_(self.guard)
                        self.sums.add_(torch.where(self.guard > 0, vals, torch.zeros_like(vals)))
                        self.flags = torch.cat([finite.to(torch.float32), self.guard])
                    fo.capture_step(itf.models, itf.optims, self.guard)
                elif self.tail_split:
                    self.loss_keys = list(self.losses)
                    fused_guard = all(v.is_cuda and v.dtype == torch.float32 and v.numel() == 1 for v in self.losses.values()) \
                        and len(self.loss_keys) <= 16
                    if fused_guard:                                   # finite flags + this rank's flag-slot entry in one launch
                        self._loss_refs = [self.losses[k].reshape(()) for k in self.loss_keys]
                        self.flags = torch.empty(len(self.loss_keys) + 1, device=dev)
                        fo.capture_gather(itf.models, itf.optims, None, names=('dncnn',) if self.overlap else None)
                        ops.step_guard_local_(self._loss_refs, self.ok, self.flags, fo.flag_slot())
                    else:
                        vals = torch.stack([self.losses[k].reshape(()) for k in self.loss_keys])
                        finite = torch.isfinite(vals)
                        local = finite.all().to(torch.float32).reshape(1) * self.ok
                        fo.capture_gather(itf.models, itf.optims, local, names=('dncnn',) if self.overlap else None)
            if self.overlap:
                self._rest = tuple(n for n in fo.flats if n != 'dncnn')
                self.graph_a2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_a2, pool=self.graph.pool(), capture_error_mode="thread_local"):
                    itf._backward_stage2()                            # through the PathNets, from the P-buffers' gradients
                    fo.capture_gather(itf.models, itf.optims, None, names=self._rest)
                fo._gather_open = False
            if self.tail_split:
                self.graph_b = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_b, pool=self.graph.pool(), capture_error_mode="thread_local"):
                    if fused_guard:                                   # global guard, poison flag, loss sums in one launch
                        ops.step_guard_global_(self._loss_refs, fo.flag_slot(), self.ok, self.sums, self.flags)
                        self.guard = fo.capture_update(self.flags[len(self.loss_keys):])
                    else:
                        self.guard = fo.capture_update()              # 1 when NO rank saw a non-finite loss
                        self.ok.copy_(self.guard)
                        self.sums.add_(torch.where(self.guard > 0, vals, torch.zeros_like(vals)))
                        self.flags = torch.cat([finite.to(torch.float32), self.guard])
        finally:
            ops.USE_SIDE_STREAM = side
        if fo is not None:
            fo.leave_grads = False                        # .grad must keep pointing at the captured buffers

    def _capture_halves(self, dev):
        """Graphs of the two-stream step: ``graph_h`` (the shared split of ``paths``, on the launch stream), ``graph_d`` /
        ``graph_s`` (the halves, each on a stream and in a memory pool of its own: they run concurrently).  Returns the pool the
        tail graph shares (it runs after both halves, on the launch stream)."""
        itf = self.itf
        self.half_streams = ops.concurrent_stream_pair(dev)      # two streams on different hardware queues, probed once per process
        branch = ops.USE_BRANCH_STREAM
        ops.USE_BRANCH_STREAM = False                      # a half is linear: nothing forks inside it
        try:
            self.graph_h = None
            pre = getattr(itf.models.get('backbone_diffuse'), '_paths_nhwc', None) if itf.use_llpm_buf else None
            keys = self.fm is not None and self.dev_keys
            if pre is not None or keys:
                self.graph_h = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_h, capture_error_mode="thread_local"):
                    if keys:
                        ops.step_counter_advance(self.key_state)
                    if pre is not None:
                        pre(self.static)                   # NHWC / split copy of `paths`, read by both halves
            self._half_out, self.half_graphs = [], []
            for i, (br, st) in enumerate(zip(('diffuse', 'specular'), self.half_streams)):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
                    if keys:
                        self._draw_captured(i)             # this half's pairings (FeatureMSE reads perms[i])
                    self._half_out.append(itf._half_forward_backward(self.static, br))
                self.half_graphs.append(g)
        finally:
            ops.USE_BRANCH_STREAM = branch
        return self.half_graphs[0].pool()

    def _replay(self):
        """One replay of the captured step up to (and including) ``self.graph``."""
        if self.two_stream:
            main = torch.cuda.current_stream()
            if self.graph_h is not None:
                self.graph_h.replay()
            for g, st in zip(self.half_graphs, self.half_streams):
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    g.replay()
            for st in self.half_streams:
                main.wait_stream(st)
        self.graph.replay()

    def time_replays(self, n=10):
        """Milliseconds per replay of the captured step, measured with the optimiser held back by the device guard (``ok`` = 0:
        parameters, moments, step counters and running sums stay as they are), so that a fresh capture can be judged before it
        is used (``capture_validated``).  Captured single-rank tail only; returns None otherwise."""
        if not self.tail_captured:
            return None
        self.ok.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self._replay()                                     # (first replay: uploads the executable graph)
        e0.record()
        for _ in range(n):
            self._replay()
        e1.record()
        e1.synchronize()
        self.ok.fill_(1.0)
        return e0.elapsed_time(e1) / n

    def _draw_captured(self, half):
        """Under capture: the two permutations of one half (slot 2 * half: patch, 2 * half + 1: batch), keyed from ``key_state``."""
        ip, ib = self.perms[half]
        ops.random_permutation_dev(ip, self.key_state, 2 * half)
        if ib is not None:
            ops.random_permutation_dev(ib, self.key_state, 2 * half + 1)

    def reseed(self, seed=None):
        """Key the captured draws: seed (default: one draw from torch's CPU generator, so ``torch.manual_seed`` fixes the whole
        sequence of pairings) and counter 0.  Done by the first call of the step if nobody did it before."""
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        self.key_state.copy_(torch.tensor([int(seed), 0], dtype=torch.int64), non_blocking=False)
        self._seeded = True

    def _draw(self):
        """Fresh pairings, in the reference's call order (diffuse: patch, batch; specular: patch, batch)."""
        if self.fm.rng == 'device':                       # written in place: no sort, no copy; the keys of the step's
            outs = [t for pair in self.perms for t in pair if t is not None]      # permutations in ONE draw (the same stream
            seeds = torch.randint(0, 2 ** 62, (len(outs),)).tolist()              # of keys as one draw per permutation)
            for t, seed in zip(outs, seeds):
                ops.random_permutation(t.numel(), t.device, out=t, seed=seed)
            return
        for ip, ib in self.perms:
            ip.copy_(torch.randperm(self.perm_sizes[0]), non_blocking=True)
            if ib is not None:
                ib.copy_(torch.randperm(self.perm_sizes[1]), non_blocking=True)

    def __call__(self, batch):
        itf = self.itf
        itf.preprocess(batch)                             # key asserts + iters += 1
        dst, src = [], []
        for k in self.keys:
            v, b = self.static[k], batch[k]
            if b.data_ptr() != v.data_ptr():
                dst.append(v); src.append(b)
        if dst:                                           # one multi-tensor launch where torch can fuse it (same device / dtype)
            if all(b.is_cuda and b.dtype == v.dtype and b.shape == v.shape for v, b in zip(dst, src)):
                torch._foreach_copy_(dst, src)
            else:
                for v, b in zip(dst, src):
                    v.copy_(b, non_blocking=True)
        if self.fm is not None:
            if self.dev_keys:
                if not self._seeded:
                    self.reseed()
            else:
                self._draw()
            self.fm._static_i = 0
        if not (self.tail_captured or self.tail_split):
            self._replay()
            itf._logging(self.losses)
            itf._optimization()
            return
        fo = itf.fused_optim
        for i, k in enumerate(self.loss_keys):            # itf.m_losses['m_<key>'] are views of self.sums; get_epoch_summary
            cur = itf.m_losses.get('m_' + k)              # replaces them with fresh zeros (interfaces.py:320-333): adopt those
            if cur is not self._sum_views[i]:
                self.sums[i].copy_(cur) if cur is not None else self.sums[i].zero_()
                itf.m_losses['m_' + k] = self._sum_views[i]
        fo.refresh_hyper(itf.optims)
        self._replay()
        if self.tail_split:
            ev = getattr(self, 'tail_events', None)       # (bench.py: a list that receives (start, end) events of the tail)
            if ev is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            if self.overlap:
                works = fo.allreduce_async(('dncnn',))    # on the wire while the PathNets' backward runs
                self.graph_a2.replay()
                works += fo.allreduce_async(self._rest)
                for w in works:
                    w.wait()
            else:
                fo.allreduce()                            # eager: three asynchronous RCCL sums, the launch stream waits for them
            self.graph_b.replay()
            if ev is not None:
                e1.record()
                ev.append((e0, e1))
        itf.last_loss_dict = self.losses
        cb = getattr(self, 'after_enqueue', None)         # (a loader's ``kick``: the step is enqueued, the host is about to wait)
        if cb is not None:
            cb()
        if self.defer_check:
            if self._flag_bufs is None:
                self._flag_bufs = [(torch.empty(self.flags.numel(), dtype=torch.float32).pin_memory(), torch.cuda.Event()) for _ in range(2)]
            slot = self._n_calls & 1
            self._n_calls += 1
            host, ev = self._flag_bufs[slot]
            host.copy_(self.flags, non_blocking=True)
            ev.record()
            fo.after_replay(True)                         # optimistic: taken back by _check if the guard skipped the update
            fo.last_guard = self.guard
            prev, self._pending = self._pending, slot
            if prev is not None:
                self._check(prev, after=1)
            return
        self._raise_unless_finite(self.flags.tolist())    # the step's one sync

    def _raise_unless_finite(self, flags, rollback=0):
        """rollback = 0: the counters of the step have not been advanced yet; k > 0: they were, optimistically, for this step
        and the k - 1 steps enqueued behind it (which the poisoned guard skipped as well)."""
        fo = self.itf.fused_optim
        ok = flags[-1] != 0
        if rollback:
            if not ok:
                fo.rollback(rollback)
        else:
            fo.after_replay(ok)
            fo.last_guard = self.guard
        if not ok:                                        # (the guard kept the sums, the moments and the parameters as they were)
            self._pending = None                          # (a step enqueued behind this one was skipped: its flags say nothing)
            self.ok.fill_(1.0)                            # the error is being raised: later steps may update again
            for k, f in zip(self.loss_keys, flags[:-1]):
                if not f:
                    raise RuntimeError("%s: Non-finite loss at train time." % (k))
            raise RuntimeError("Non-finite loss at train time on another rank.")     # (every rank skipped the update)

    def _check(self, slot, after=0):
        host, ev = self._flag_bufs[slot]
        ev.synchronize()
        self._raise_unless_finite(host.tolist(), rollback=1 + after)

    def close(self):
        """Release the captured graphs (last captured first), their memory pool and the static batch NOW instead of whenever the
        last reference to this object goes: a process that builds one graphed step after another (a test session, a sweep over
        configurations) keeps at most one alive.  The object cannot be called afterwards."""
        torch.cuda.synchronize()
        fo = getattr(self.itf, "fused_optim", None) if self.itf is not None else None
        if fo is not None and getattr(fo, "last_guard", None) is self.__dict__.get("guard"):
            fo.last_guard = None                          # (a tensor of the pool that is about to go)
        graphs = [self.__dict__.pop(name, None) for name in ("graph_b", "graph_a2", "graph")]
        graphs += list(reversed(self.__dict__.pop("half_graphs", []) or [])) + [self.__dict__.pop("graph_h", None)]
        for g in graphs:                                  # (last captured first)
            if g is not None:
                g.reset()
        for name in ("losses", "static", "flags", "guard", "sums", "_sum_views", "perms", "_loss_refs", "_flag_bufs", "_pending", "ok",
                     "after_enqueue", "tail_events", "_half_out", "half_streams", "key_state"):
            self.__dict__.pop(name, None)
        if self.fm is not None and getattr(self.fm, "static_perms", None) is not None:
            self.fm.static_perms, self.fm.check_finite = None, True       # (the eager loss draws and checks for itself again)
        self.itf.last_loss_dict = None
        self.itf.last_out = None
        self.itf = None
        torch.cuda.synchronize()

    def flush(self):
        """The deferred check of the last step (``defer_check=True``); a no-op otherwise."""
        prev, self._pending = self._pending, None
        if prev is not None:
            self._check(prev)


# ---- capture validation ------------------------------------------------------------------------------------------------------
# The speed of a captured step is decided at capture / instantiate time (hipGraphInstantiate maps the graph's parallel branches
# onto streams of its own choosing) and stays what it is for the life of the graph: a capture whose halves ended up in series
# costs 9-12 % for a whole run.  ``capture_validated`` times every fresh capture and re-captures the slow ones.
_BEST_MS = {}


def capture_validated(itf, batch, attempts=3, min_attempts=2, tol=0.05, replays=10, **kw):
    """``GraphedTrainStep(itf, batch, **kw)`` whose replay time is within ``tol`` of the best this process has seen for the same
    configuration: a capture that is slower is closed (never two live steps) and made again, at most ``attempts`` times; the
    first time a configuration is captured in a process, at least ``min_attempts`` captures are compared (there is no earlier
    time to judge the first one by).  The accepted step carries ``capture_attempts`` and ``capture_ms`` (every attempt's time)."""
    key = (ops.PRECISION, bool(kw.get("two_stream")), bool(ops.USE_BRANCH_STREAM), bool(ops.USE_SIDE_STREAM),
           tuple(sorted((k, tuple(v.shape)) for k, v in batch.items() if isinstance(v, torch.Tensor))),
           tuple(sorted((n, sum(p.numel() for p in m.parameters())) for n, m in itf.models.items())))
    tried = []
    rng0 = torch.get_rng_state()          # (every attempt's warm-up draws pairings from the CPU generator: all attempts start from the
    for a in range(max(1, attempts)):     # same state, so the generator ends up where ONE capture would have left it)
        torch.set_rng_state(rng0)
        step = GraphedTrainStep(itf, batch, **kw)
        t = step.time_replays(replays)
        if t is None:                                      # (nothing to validate by: multi-rank tail, eager optimiser)
            step.capture_attempts, step.capture_ms = 1, []
            return step
        tried.append(round(t, 4))
        seen = key in _BEST_MS
        best = min(_BEST_MS.get(key, t), t)
        _BEST_MS[key] = best
        last = a == max(1, attempts) - 1
        if last or (t <= best * (1.0 + tol) and (seen or len(tried) >= min_attempts)):
            step.capture_attempts, step.capture_ms = len(tried), tried
            return step
        step.close()
