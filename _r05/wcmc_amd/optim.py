"""Fused ``clip_grad_value_(1.0)`` + ``Adam.step()`` (+ the cross-rank gradient average).

Replaces ``support/interfaces.py:260-261`` and ``:269-271`` (optimisers built at
``train_kpcn.py:274-277``): per model one flat parameter buffer, one flat gradient bucket, one
RCCL all-reduce per bucket when a process group is given, one ``wcmc_clip_adam`` launch.

Multi-rank order of operations (``train_kpcn.py:266-269``: ``nn.DataParallel`` sums the replica
gradients, ``interfaces.py:260-261`` clips the sum's mean afterwards): buckets are issued in the
order the backward passes complete them (``interfaces.py:237-238``: diffuse PathNet, KPCN, specular
PathNet) as asynchronous all-reduces (SUM) on the communicator's stream; the launch stream waits for
bucket *i* only, so the clip + Adam of bucket *i* runs while buckets *i+1..* are still on the wire.
The 1/world mean is folded into the kernel (``grad_scale``) -- reduce, then scale, then clip.

The non-finite guard is global: every rank appends ``1 - guard`` to its first bucket, so after the
sum every rank holds the number of ranks whose loss was not finite and either all ranks skip the
update and raise (``interfaces.py:254-257``), or none does.

The ``torch.optim.Adam`` objects the caller built stay the source of truth for hyper-parameters
(``param_groups[0]['lr'|'betas'|'eps']`` are read every step) and keep a regular ``state``
(``step`` / ``exp_avg`` / ``exp_avg_sq`` as views of the flat buffers) so that
``optim.state_dict()`` -- which the reference pickles into its checkpoints
(``train_kpcn.py:110-118``) -- stays meaningful.
"""
import torch

from . import ops

_ALIGN = 4                      # floats: every parameter starts on a 16-byte boundary of the flat buffers
BUCKET_ORDER = ("backbone_diffuse", "dncnn", "backbone_specular")


def _round_up(n, a):
    return (n + a - 1) // a * a


class _Flat:
    def __init__(self, model, optim):
        params = [p for p in model.parameters()]
        group_params = [p for g in optim.param_groups for p in g["params"]]
        assert len(optim.param_groups) == 1 and len(group_params) == len(params) and \
            all(a is b for a, b in zip(params, group_params)), \
            "FusedClipAdam expects optim.Adam(model.parameters()) with a single param group"
        g0 = optim.param_groups[0]
        if g0.get("weight_decay", 0) != 0 or g0.get("amsgrad", False) or g0.get("maximize", False):
            raise NotImplementedError("FusedClipAdam: only plain Adam (train_kpcn.py:277)")
        self.params = params
        self.sizes = [p.numel() for p in params]
        self.offsets, off = [], 0
        for n in self.sizes:
            self.offsets.append(off)
            off += _round_up(n, _ALIGN)
        self.total = off                                   # multiple of _ALIGN; the gaps hold zeros forever
        dev = params[0].device
        self.flat = torch.zeros(self.total, device=dev, dtype=torch.float32)
        self.m = torch.zeros(self.total, device=dev, dtype=torch.float32)
        self.v = torch.zeros(self.total, device=dev, dtype=torch.float32)
        # gradient bucket: [total gradients | flag slot (number of ranks with a non-finite loss) | 3 x pad]
        self.g = torch.zeros(self.total + _ALIGN, device=dev, dtype=torch.float32)
        self.steps = 0                                     # updates of this model so far
        self.psteps = [0] * len(params)                    # Adam's step count PER PARAMETER, as torch.optim.Adam keeps it
        self.step_t = torch.tensor(0.0)                    # optim.state[p]['step'] of every parameter whose count == steps
        self.stepped = False                               # did the last FusedClipAdam.step() touch this model
        for p, n, o in zip(params, self.sizes, self.offsets):
            self.flat[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + n].view(p.shape)
        self._adopt_state(optim)

    def _views(self, buf):
        return [buf[o:o + n].view(p.shape) for p, n, o in zip(self.params, self.sizes, self.offsets)]

    def grad_views(self):
        """Views of the gradient bucket, one per parameter -- built once (a hundred slice + view calls per model and step
        were ~170 us of host time during which the GPU had nothing to do) and registered as the parameters' gradient sinks
        (``ops.register_grad_sinks``: the weight-gradient kernels then write into the bucket directly)."""
        if getattr(self, "_gviews", None) is None:
            self._gviews = self._views(self.g)
            if self.params[0].is_cuda:
                ops.register_grad_sinks(self.params, self._gviews)
        return self._gviews

    def gather(self, have):
        """Copy the gradients that are NOT already in the bucket (a gradient a sink received IS its bucket view)."""
        gv = self.grad_views()
        ops.release_grad_sinks(self.params)                 # (the accumulation window of these gradients ends here)
        idx = [j for j, h in enumerate(have) if h and self.params[j].grad.data_ptr() != gv[j].data_ptr()]
        if idx:
            torch._foreach_copy_([gv[j] for j in idx], [self.params[j].grad for j in idx])
        return len(idx)

    def _adopt_state(self, optim):
        """(Re)bind optim.state to views of the flat moments, importing loaded checkpoints."""
        mv, vv = self._views(self.m), self._views(self.v)
        self.optim = optim
        self._uniform_published = False                     # (a load_state_dict() brings state dicts with 'step' tensors of their own)
        for i, (p, m, v) in enumerate(zip(self.params, mv, vv)):
            st = optim.state[p]
            if "exp_avg" in st and st["exp_avg"].data_ptr() != m.data_ptr():
                m.copy_(st["exp_avg"])
                v.copy_(st["exp_avg_sq"])
                if "step" in st:
                    self.psteps[i] = int(st["step"])
            st["exp_avg"], st["exp_avg_sq"] = m, v
        self.steps = max(self.psteps) if self.psteps else 0
        self._publish_steps(optim)

    def _publish_steps(self, optim):
        """optim.state[p]['step']: the shared tensor for parameters that took part in every update, an own one otherwise."""
        self.step_t.fill_(float(self.steps))
        uniform = min(self.psteps) == self.steps
        if uniform and getattr(self, "_uniform_published", False):
            return                                          # (the usual case: every state already points at step_t)
        for p, n in zip(self.params, self.psteps):
            st = optim.state[p]
            if n == self.steps:
                st["step"] = self.step_t
            elif not isinstance(st.get("step"), torch.Tensor) or st["step"] is self.step_t or float(st["step"]) != n:
                st["step"] = torch.tensor(float(n))
        self._uniform_published = uniform

    def bound(self, optim):
        st = optim.state.get(self.params[0], {})
        return "exp_avg" in st and st["exp_avg"].data_ptr() == self.m.data_ptr()

    def segments(self, have):
        """Runs of consecutive parameters that have a gradient AND the same Adam step count, as (first offset, end offset,
        step count of this update) of the flat buffers.  ``torch.optim.Adam`` skips a parameter whose ``.grad`` is None (no
        moment decay, no update, no step) and bias-corrects every parameter with ITS OWN count: so do we -- one launch per run,
        i.e. one per model while all its parameters train together."""
        out, start, cur = [], None, None
        for i, h in enumerate(have):
            n = self.psteps[i] + 1
            if h and start is not None and n != cur:
                out.append((start, self.offsets[i], cur))
                start = None
            if h and start is None:
                start, cur = self.offsets[i], n
            if not h and start is not None:
                out.append((start, self.offsets[i], cur))
                start = None
        if start is not None:
            out.append((start, self.total, cur))
        return out


class FusedClipAdam:
    def __init__(self, models, optims, process_group=None, clip=1.0, force_collective=False, order=None):
        """force_collective: run the bucket all-reduces even on a ONE-rank process group (a sum over one rank is the identity:
        results are those of the world-1 path bit for bit) -- the multi-rank code path, exercised where only one GPU is at hand
        (tests/test_gpu_models.py, the `multi_rank_path` leg of bench.py)."""
        self.clip = clip
        self.leave_grads = True      # re-point p.grad at the clipped flat gradient like clip_grad_value_ leaves it
        self.group = process_group
        self.world = torch.distributed.get_world_size(process_group) if process_group is not None else 1
        self.collective = self.world > 1 or (bool(force_collective) and process_group is not None)
        # order: the bucket order (default BUCKET_ORDER: the order the reference's two backward passes complete the models); a
        # step whose backward is cut at the P-buffers (GraphedTrainStep(overlap_allreduce=True)) finishes `dncnn` first
        base = tuple(order) if order is not None else BUCKET_ORDER
        order = [n for n in base if n in models] + [n for n in models if n not in base]
        self.flats = {name: _Flat(models[name], optims["optim_" + name]) for name in order}
        for fl in self.flats.values():
            fl.grad_views()          # (built now: registers the gradient sinks before the first backward)
        self.last_guard = None       # device float: 1 when every rank's losses were finite at the last step

    def step(self, models, optims, guard=None):
        """guard: optional device float of THIS rank; 0 turns the update of this step into a no-op -- on every
        rank.  Returns the global guard (device float) or None when no guard was given."""
        work = []
        first = True
        for name, fl in self.flats.items():
            optim = optims["optim_" + name]
            if not fl.bound(optim):
                fl._adopt_state(optim)
            have = [p.grad is not None for p in fl.params]
            fl.stepped = any(have)
            if not fl.stepped:                              # e.g. a frozen model: Adam.step() does nothing
                work.append(None)
                continue
            # (several ranks must show the same pattern of gradients: the buckets are collectives, one per stepped model)
            fl.gather(have)                                  # (device copies of whatever a sink did not receive)
            n_msg = fl.total
            if first and guard is not None:
                fl.g[fl.total:fl.total + 1].copy_((1.0 - guard).reshape(1))
                n_msg = fl.total + _ALIGN
            if self.collective:                              # RCCL sum on the communicator's stream
                work.append(torch.distributed.all_reduce(fl.g[:n_msg], group=self.group, async_op=True))
            else:
                work.append(None)
            if first:
                first_fl = fl
            first = False
        gguard = None
        for (name, fl), w in zip(self.flats.items(), work):
            if not fl.stepped:
                continue
            if w is not None:
                w.wait()                                     # launch stream waits for THIS bucket only
            if guard is not None and gguard is None:
                gguard = (first_fl.g[first_fl.total] == 0).to(torch.float32)
            optim = optims["optim_" + name]
            g0 = optim.param_groups[0]
            have = [p.grad is not None for p in fl.params]
            for a, b, nstep in fl.segments(have):
                ops.clip_adam_(fl.flat[a:b], fl.g[a:b], fl.m[a:b], fl.v[a:b], nstep, float(g0["lr"]),
                               float(g0["betas"][0]), float(g0["betas"][1]), float(g0["eps"]), clip=self.clip,
                               grad_scale=1.0 / self.world, guard=gguard)
            fl.steps += 1
            fl.psteps = [n + 1 if h else n for n, h in zip(fl.psteps, have)]
            fl.last_have = have
            # leave the (averaged, clipped) gradients behind as the reference does
            if self.leave_grads:
                for p, gview, h in zip(fl.params, fl.grad_views(), have):
                    if h:
                        p.grad = gview
            fl._publish_steps(optim)
        self.last_guard = gguard
        return gguard

    # ---- the same step as part of a captured hipGraph (one rank: no collective inside the capture)
    def capture_step(self, models, optims, guard):
        """Enqueue gather + clip + Adam of every model ONCE, under stream capture (``wcmc_amd.graph.GraphedTrainStep``):
        the kernels read step size / bias correction from ``self.hyper`` (device), which ``refresh_hyper`` fills before every
        replay.  The set of parameters that have a gradient is frozen with the capture (as the graph itself is)."""
        assert not self.collective, "a collective cannot be captured: capture_gather / allreduce / capture_update split the tail around it"
        assert getattr(self, "hyper", None) is not None, "prepare_capture() first (no allocation of pinned memory inside a capture)"
        self._captured = []
        for i, (name, fl) in enumerate(self.flats.items()):
            optim = optims["optim_" + name]
            # (adopting a loaded optimiser state here would RECORD its moment copies into the graph: every replay would reset
            # the moments to the checkpoint's -- prepare_capture(optims) adopts before the capture begins)
            if not fl.bound(optim):
                raise RuntimeError("FusedClipAdam.capture_step: the state of optim_%s is not bound to the flat buffers (it was loaded "
                                   "after FusedClipAdam was built): call prepare_capture(optims) before the capture" % name)
            have = [p.grad is not None for p in fl.params]
            fl.stepped = any(have)
            self._captured.append((name, fl, fl.stepped, have))
            if not fl.stepped:
                continue
            assert len({n for n, h in zip(fl.psteps, have) if h}) == 1, \
                "a captured optimiser bias-corrects a model's parameters with ONE step count: they must have trained together"
            fl.gather(have)
            for a, b, _ in fl.segments(have):
                ops.clip_adam_dev_(fl.flat[a:b], fl.g[a:b], fl.m[a:b], fl.v[a:b], self.hyper[i], clip=self.clip,
                                   grad_scale=1.0, guard=guard)

    # ---- the multi-rank step as TWO captured pieces around the eager collectives (no RCCL kernel inside a capture):
    #   graph A  ... backward, capture_gather: gradients -> buckets, this rank's (1 - guard) -> the flag slot of the first bucket
    #   eager    allreduce(): the buckets summed over the ranks, asynchronously, in backward order; the launch stream waits
    #   graph B  capture_update: global guard from the summed flag slot, then scale (1 / world) -> clip -> Adam per bucket
    def capture_gather(self, models, optims, guard, names=None):
        """names: the models whose gradients are complete at this point of the capture (None: all).  The first call starts the list
        of captured buckets (its first stepped bucket carries the flag slot), later calls append to it."""
        assert getattr(self, "hyper", None) is not None, "prepare_capture() first"
        first = names is None or not getattr(self, "_gather_open", False)
        if first:
            self._captured = []
        self._gather_open = names is not None
        for name, fl in self.flats.items():
            if names is not None and name not in names:
                continue
            optim = optims["optim_" + name]
            if not fl.bound(optim):
                raise RuntimeError("FusedClipAdam.capture_gather: call prepare_capture(optims) before the capture (optim_%s)" % name)
            have = [p.grad is not None for p in fl.params]
            fl.stepped = any(have)
            self._captured.append((name, fl, fl.stepped, have))
            if not fl.stepped:
                continue
            assert len({n for n, h in zip(fl.psteps, have) if h}) == 1, \
                "a captured optimiser bias-corrects a model's parameters with ONE step count: they must have trained together"
            fl.gather(have)
            fl.n_msg = fl.total
            if first:
                if guard is not None:                        # (None: the caller fills flag_slot() itself -- ops.step_guard_local_)
                    fl.g[fl.total:fl.total + 1].copy_((1.0 - guard).reshape(1))
                fl.n_msg = fl.total + _ALIGN
                self._first_fl = fl
            first = False

    def allreduce_async(self, names=None):
        """One asynchronous all-reduce (SUM) per stepped bucket (of `names`) on the communicator's stream; returns the work handles."""
        return [torch.distributed.all_reduce(fl.g[:fl.n_msg], group=self.group, async_op=True)
                for name, fl, stepped, _ in self._captured if stepped and (names is None or name in names)]

    def allreduce(self):
        """Between the two graphs: one asynchronous all-reduce (SUM) per stepped bucket on the communicator's stream, issued in
        backward order; the launch stream then waits for all of them (graph B reads every bucket)."""
        for w in self.allreduce_async():
            w.wait()

    def flag_slot(self):
        """The float behind the first stepped bucket that carries 1 - guard of every rank through that bucket's all-reduce."""
        return self._first_fl.g[self._first_fl.total:self._first_fl.total + 1]

    def capture_update(self, gguard=None):
        """Graph B; returns the global guard (device float: 1 when no rank saw a non-finite loss).  gguard: that guard when the caller
        has already formed it (ops.step_guard_global_)."""
        if gguard is None:
            gguard = (self._first_fl.g[self._first_fl.total] == 0).to(torch.float32).reshape(1)
        for i, (name, fl, stepped, have) in enumerate(self._captured):
            if not stepped:
                continue
            for a, b, _ in fl.segments(have):
                ops.clip_adam_dev_(fl.flat[a:b], fl.g[a:b], fl.m[a:b], fl.v[a:b], self.hyper[i], clip=self.clip,
                                   grad_scale=1.0 / self.world, guard=gguard)
        return gguard

    def prepare_capture(self, optims=None):
        """Buffers of the captured step, allocated BEFORE the capture begins (hipHostMalloc invalidates a stream capture), and
        the adoption of optimiser state loaded since construction (its copies must run now, not be recorded into the graph)."""
        if optims is not None:
            for name, fl in self.flats.items():
                if not fl.bound(optims["optim_" + name]):
                    fl._adopt_state(optims["optim_" + name])
        dev = next(iter(self.flats.values())).flat.device
        self.hyper = torch.zeros(len(self.flats), 8, device=dev, dtype=torch.float32)
        # (a ring: with GraphedTrainStep(defer_check=True) the host prepares step t + 1 while step t's copy may still be queued)
        self._hyper_host = [torch.zeros(len(self.flats), 8, dtype=torch.float32).pin_memory() for _ in range(3)]
        self._hyper_slot = 0
        for fl in self.flats.values():
            fl.grad_views()

    def refresh_hyper(self, optims):
        """Before a replay: this step's scalars (``optim.param_groups[0]`` is read every step, like the eager path)."""
        self._hyper_slot = (self._hyper_slot + 1) % len(self._hyper_host)
        host = self._hyper_host[self._hyper_slot]
        for i, (name, fl, stepped, have) in enumerate(self._captured):
            if not stepped:
                continue
            g0 = optims["optim_" + name].param_groups[0]
            nstep = next(n for n, h in zip(fl.psteps, have) if h) + 1
            h = ops.clip_adam_hyper(nstep, float(g0["lr"]), float(g0["betas"][0]), float(g0["betas"][1]), float(g0["eps"]))
            host[i, :7] = torch.tensor(h)
        self.hyper.copy_(host, non_blocking=True)

    def after_replay(self, updated):
        """Host bookkeeping of a replayed step: the counters advance unless the device guard skipped the update."""
        for name, fl, stepped, have in self._captured:
            fl.stepped = stepped and updated
            if fl.stepped:
                fl.steps += 1
                fl.psteps = [n + 1 if h else n for n, h in zip(fl.psteps, have)]
                fl.last_have = have
                fl._publish_steps(fl.optim)               # optim.state[p]['step'] of every parameter, uniform or not

    def rollback(self, n=1):
        """The guard turned the last step (the last ``n`` steps: a deferred check finds step t non-finite after step t + 1 was
        enqueued behind the same, now poisoned, guard) into a no-op: take the step counters back, as the reference never
        reaches ``optim.step()`` in that case (``interfaces.py:254-271``)."""
        for fl in self.flats.values():
            if fl.stepped:
                fl.steps -= n
                fl.psteps = [k - n if h else k for k, h in zip(fl.psteps, fl.last_have)]
                fl._publish_steps(fl.optim)
                fl.stepped = False
