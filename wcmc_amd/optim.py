"""Fused ``clip_grad_value_(1.0)`` + ``Adam.step()`` (+ the cross-rank gradient average).

Replaces ``support/interfaces.py:260-261`` and ``:269-271`` (optimisers built at
``train_kpcn.py:274-277``): per model one flat parameter buffer, one flat gradient gather, one
RCCL all-reduce when a process group is given, one ``wcmc_clip_adam`` launch.

The ``torch.optim.Adam`` objects the caller built stay the source of truth for hyper-parameters
(``param_groups[0]['lr'|'betas'|'eps']`` are read every step) and keep a regular ``state``
(``step`` / ``exp_avg`` / ``exp_avg_sq`` as views of the flat buffers) so that
``optim.state_dict()`` -- which the reference pickles into its checkpoints
(``train_kpcn.py:110-118``) -- stays meaningful.
"""
import torch

from . import ops


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
        total = sum(self.sizes)
        dev = params[0].device
        self.flat = torch.empty(total, device=dev, dtype=torch.float32)
        self.m = torch.zeros(total, device=dev, dtype=torch.float32)
        self.v = torch.zeros(total, device=dev, dtype=torch.float32)
        self.steps = 0
        off = 0
        for p, n in zip(params, self.sizes):
            self.flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + n].view(p.shape)
            off += n
        self._adopt_state(optim)

    def _views(self, buf):
        out, off = [], 0
        for p, n in zip(self.params, self.sizes):
            out.append(buf[off:off + n].view(p.shape))
            off += n
        return out

    def _adopt_state(self, optim):
        """(Re)bind optim.state to views of the flat moments, importing loaded checkpoints."""
        mv, vv = self._views(self.m), self._views(self.v)
        for p, m, v in zip(self.params, mv, vv):
            st = optim.state[p]
            if "exp_avg" in st and st["exp_avg"].data_ptr() != m.data_ptr():
                m.copy_(st["exp_avg"])
                v.copy_(st["exp_avg_sq"])
                self.steps = int(st["step"]) if "step" in st else self.steps
            st["exp_avg"], st["exp_avg_sq"] = m, v
            st["step"] = torch.tensor(float(self.steps))

    def bound(self, optim):
        st = optim.state.get(self.params[0], {})
        return "exp_avg" in st and st["exp_avg"].data_ptr() == self.m.data_ptr()


class FusedClipAdam:
    def __init__(self, models, optims, process_group=None, clip=1.0):
        self.clip = clip
        self.leave_grads = True      # re-point p.grad at the clipped flat gradient like clip_grad_value_ leaves it
        self.group = process_group
        self.world = torch.distributed.get_world_size(process_group) if process_group is not None else 1
        self.flats = {name: _Flat(models[name], optims["optim_" + name]) for name in models}

    def step(self, models, optims, guard=None):
        """guard: optional device float; 0 turns every update of this step into a no-op."""
        for name, fl in self.flats.items():
            optim = optims["optim_" + name]
            if not fl.bound(optim):
                fl._adopt_state(optim)
            grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in fl.params]
            flat_g = torch.cat([g.reshape(-1) for g in grads])          # one gather (device copy)
            if self.world > 1:
                torch.distributed.all_reduce(flat_g, group=self.group)  # RCCL sum; the mean is folded below
            g0 = optim.param_groups[0]
            fl.steps += 1
            ops.clip_adam_(fl.flat, flat_g, fl.m, fl.v, fl.steps, float(g0["lr"]), float(g0["betas"][0]),
                           float(g0["betas"][1]), float(g0["eps"]), clip=self.clip, grad_scale=1.0 / self.world,
                           guard=guard)
            # leave the (averaged, clipped) gradients behind as the reference does
            if self.leave_grads:
                for p, gv in zip(fl.params, fl._views(flat_g)):
                    p.grad = gv
            for p in fl.params:
                optim.state[p]["step"] = torch.tensor(float(fl.steps))
