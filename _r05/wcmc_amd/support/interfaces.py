"""``support/interfaces.py:18-333`` of the reference: ``BaseInterface`` / ``KPCNInterface``.

Same constructor signature, attributes (``models, optims, loss_funcs, iters, m_losses,
best_err``), method names and error behaviour, so ``train_kpcn.train_epoch_kpcn`` /
``validate_kpcn`` / ``train`` drive it unchanged.  Differences, all on purpose:

  * the P-buffer statistics + input assembly (``interfaces.py:165-180``) is one HIP kernel
    (``ops.pbuffer_cat``) instead of var/mean/cat;
  * gradients may be averaged across ranks (``grad_sync``) before the clip, which is where
    ``nn.DataParallel``'s reduce sits in the reference (``train_kpcn.py:266-269``);
  * clip + Adam may run as the fused HIP kernel (``wcmc_amd.optim.FusedClipAdam``);
  * the P-buffer PNG dump at ``iters % 1000 == 1`` (``interfaces.py:130-137``) is skipped when
    ``../LLPM_results`` does not exist instead of raising FileNotFoundError.
"""
import os
from abc import ABCMeta, abstractmethod

import torch
import torch.nn as nn

from .. import ops as _ops
from .utils import crop_like


def _is_plain_l1(fn):
    """torch.nn.L1Loss with the default mean reduction: what train_kpcn.py:299-304 passes for every image loss."""
    return type(fn) is nn.L1Loss and fn.reduction == 'mean'


def _l1(fn, x, ref):
    """``fn(x, ref)``; the reference's plain L1Loss on a CUDA image runs as the HIP loss op (one pass, K8)."""
    if _is_plain_l1(fn) and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and x.shape == ref.shape:
        return _ops.l1_mean(x, ref)
    return fn(x, ref)


def _abstract(name):
    def missing(self, *args, **kwargs):
        raise NotImplementedError(name)
    missing.__name__ = name
    return abstractmethod(missing)


class BaseInterface(metaclass=ABCMeta):
    """``interfaces.py:18-77``: the state every interface carries (models / optimisers / losses by name, the
    iteration counter, running loss sums, best validation error) and the eleven methods a training script calls
    or a subclass must provide."""

    # name -> abstract method; the reference spells them out one by one
    for _name in ("to_train_mode", "preprocess", "train_batch", "_manifold_forward", "_regress_forward", "_backward",
                  "_logging", "_optimization", "to_eval_mode", "validate_batch", "get_epoch_summary"):
        locals()[_name] = _abstract(_name)
    del _name

    def __init__(self, models, optims, loss_funcs, args, visual=False, use_llpm_buf=False, manif_learn=False,
                 w_manif=0.1):
        self.models, self.optims, self.loss_funcs, self.args = models, optims, loss_funcs, args
        self.visual, self.use_llpm_buf, self.manif_learn, self.w_manif = visual, use_llpm_buf, manif_learn, w_manif
        self.iters, self.m_losses, self.best_err, self.fixed_batch = 0, {}, 1e10, None


_BATCH_KEYS = ('target_total', 'target_diffuse', 'target_specular', 'kpcn_diffuse_in', 'kpcn_specular_in',
               'kpcn_diffuse_buffer', 'kpcn_specular_buffer', 'kpcn_albedo')
_OPTIONS = ('m11r11', 'm10r01', 'm11r01', 'm10r11')


class KPCNInterface(BaseInterface):

    def __init__(self, models, optims, loss_funcs, args, visual=False, use_llpm_buf=False, manif_learn=False,
                 w_manif=0.1, train_branches=True, disentanglement_option="m11r11"):
        if manif_learn:
            assert 'backbone_diffuse' in models, "argument `models` dictionary should contain `'backbone_diffuse'` key."
            assert 'backbone_specular' in models, "argument `models` dictionary should contain `'backbone_specular'` key."
        assert 'dncnn' in models, "argument `models` dictionary should contain `'dncnn'` key."
        if train_branches:
            assert 'l_diffuse' in loss_funcs
            assert 'l_specular' in loss_funcs
        if manif_learn:
            assert 'l_manif' in loss_funcs
        assert 'l_recon' in loss_funcs
        assert 'l_test' in loss_funcs
        assert disentanglement_option in _OPTIONS

        super(KPCNInterface, self).__init__(models, optims, loss_funcs, args, visual, use_llpm_buf, manif_learn,
                                            w_manif)
        self.train_branches = train_branches
        self.disentanglement_option = disentanglement_option
        # build-specific hooks (None = the reference's behaviour)
        self.grad_sync = None       # callable(models) -> None, averages .grad across ranks IN PLACE
        self.fused_optim = None     # wcmc_amd.optim.FusedClipAdam
        self.last_out = None        # {'radiance','diffuse','specular'} of the last training forward (detached)

    def __str__(self):
        return 'KPCNInterface'

    def to_train_mode(self):
        for model_name in self.models:
            self.models[model_name].train()
            assert 'optim_' + model_name in self.optims, \
                '`optim_%s`: an optimization algorithm is not defined.' % (model_name)

    def preprocess(self, batch=None):
        for key in _BATCH_KEYS:
            assert key in batch
        if self.use_llpm_buf:
            assert 'paths' in batch
            batch.pop('_wcmc_paths_nhwc', None)     # one NHWC conversion of `paths` per step, not per run
        self.iters += 1

    # ------------------------------------------------------------------ forward pieces
    def _split(self, p_buffers, train):
        """Feature disentanglement (train interfaces.py:139-163, val :284-291)."""
        c = next(iter(p_buffers.values())).shape[2]
        assert c >= 2
        opt = self.disentanglement_option
        lo = {k: v[:, :, :c // 2, ...] for k, v in p_buffers.items()}
        hi = {k: v[:, :, c // 2:, ...] for k, v in p_buffers.items()}
        if not train:
            return None, (lo if opt in ('m10r01', 'm11r01') else p_buffers)
        if opt == 'm11r11':
            return p_buffers, p_buffers
        if opt == 'm10r01':
            return hi, lo
        if opt == 'm11r01':
            return p_buffers, lo
        return hi, p_buffers                      # m10r11

    @staticmethod
    def _assemble(batch, p_regress):
        """interfaces.py:165-180: cat([in, mean_s P, var_s P .mean_c / S (detached)]) in one kernel."""
        new_batch = {k: batch[k] for k in _BATCH_KEYS}
        new_batch['kpcn_diffuse_in'] = _ops.pbuffer_cat(batch['kpcn_diffuse_in'], p_regress['diffuse'])
        new_batch['kpcn_specular_in'] = _ops.pbuffer_cat(batch['kpcn_specular_in'], p_regress['specular'])
        return new_batch

    def _dump_pbuffers(self, p_buffers):
        """interfaces.py:130-137 (debug PNGs every 1000 iterations; forces a device sync)."""
        if not os.path.isdir('../LLPM_results'):
            return
        import numpy as np
        import matplotlib.pyplot as plt
        for br in ('diffuse', 'specular'):
            pimg = np.mean(np.transpose(p_buffers[br].detach().cpu().numpy()[0, :, :3, ...], (2, 3, 0, 1)), 2)
            plt.imsave('../LLPM_results/pbuf_%s_%s.png' % (self.args.model_name, br), np.clip(pimg, 0.0, 1.0))

    def _forward_backward(self, batch, cut=False):
        """Everything of ``train_batch`` up to (not including) ``_logging``: no host sync inside, so
        ``wcmc_amd.graph.GraphedTrainStep`` can capture it into one hipGraph.

        cut=True (``GraphedTrainStep(overlap_allreduce=True)``): the backward passes stop at the P-buffers -- the gradients of
        ``dncnn`` are complete and the P-buffers' gradients are kept; ``_backward_stage2()`` continues through the PathNets.  The
        ``dncnn`` gradient bucket can then cross the wire while the PathNets' backward still runs (buckets in backward
        order, overlapped with the remaining backward; the reference's two backward calls are ``interfaces.py:237-238``)."""
        out_manif = None
        dev = batch['kpcn_diffuse_in'].device
        _ops.fork_all_streams(dev)
        self._p_raw = None

        if self.use_llpm_buf:
            self.models['backbone_diffuse'].zero_grad()
            self.models['backbone_specular'].zero_grad()
            p_buffers = self._manifold_forward(batch)
            if cut:
                self._p_raw = p_buffers

            if self.iters % 1000 == 1 and not torch.cuda.is_current_stream_capturing():
                self._dump_pbuffers(p_buffers)

            out_manif, p_regress = self._split(p_buffers, train=True)
            batch = self._assemble(batch, p_regress)

        self.models['dncnn'].zero_grad()
        out = self._regress_forward(batch)
        self.last_out = {k: v.detach() for k, v in out.items()}      # denoised patches of the step (parity tests)

        loss_dict = self._backward(batch, out, out_manif)
        _ops.join_all_streams(dev)
        return loss_dict

    # ---- the step as two independent halves (``GraphedTrainStep(two_stream=True)``) ----------------------------------
    # With ``train_branches`` the diffuse and the specular half of the step -- PathNet, input assembly, the branch's nine
    # convolutions, kernel apply, L1 (+ manifold) loss and the whole backward -- share no parameter, no activation and no
    # autograd node (``interfaces.py:122-238``: two models' worth of PathNet, two ConvChains of ``sbmc.KPCN``, two
    # ``backward()`` calls); only the logged metrics of the recombined radiance (``:240-249``) see both.  Each half can
    # therefore be captured as a hipGraph of its own and replayed on a stream of its own.
    def halves_supported(self):
        return bool(self.train_branches) and hasattr(self.models['dncnn'], '_branch') and self.grad_sync is None

    def _half_forward_backward(self, batch, br):
        """One half (``br`` = 'diffuse' | 'specular') of ``_forward_backward``; returns (denoised branch output, its loss scalars)."""
        losses, out_manif = {}, None
        x = batch['kpcn_%s_in' % br]
        if self.use_llpm_buf:
            net = self.models['backbone_' + br]
            net.zero_grad()
            p = net(batch)
            out_manif, p_regress = self._split({br: p}, train=True)
            x = _ops.pbuffer_cat(x, p_regress[br])
        kpcn = self.models['dncnn']
        chain = getattr(kpcn, br)
        for q in chain.parameters():
            q.grad = None
        r = kpcn._branch(chain, x, batch['kpcn_%s_buffer' % br])
        tgt = crop_like(batch['target_' + br], r)
        loss = _l1(self.loss_funcs['l_' + br], r, tgt)
        if self.manif_learn:
            l_manif = self.loss_funcs['l_manif'](crop_like(out_manif[br], r), tgt)
            loss = loss + l_manif * self.w_manif
            losses['l_manif_' + br] = l_manif.detach()
        losses['l_' + br] = loss.detach()             # (L1 + w * manifold: the reference's aliasing quirk, see _backward)
        with self._defer_scope():                     # (the small layers' slab reductions: one launch at the end)
            torch.autograd.backward([loss])
        return r.detach(), losses

    def _finish_halves(self, batch, r_diffuse, r_specular, l_diffuse, l_specular):
        """What is left of the step's forward once both halves are done: the recombined radiance and its two logged metrics
        (``sbmc.KPCN.forward``'s last line, ``interfaces.py:240-249``); returns ``loss_dict`` in the reference's key order."""
        with torch.no_grad():
            albedo = crop_like(batch['kpcn_albedo'], r_diffuse)
            total = _ops.recombine(albedo, r_diffuse, r_specular)
            self.last_out = dict(radiance=total, diffuse=r_diffuse, specular=r_specular)
            tgt_total = crop_like(batch['target_total'], total)
            loss_dict = {}
            for k in ('l_manif_diffuse', 'l_manif_specular', 'l_diffuse', 'l_specular'):
                src = l_diffuse if k.endswith('diffuse') else l_specular
                if k in src:
                    loss_dict[k] = src[k]
            if self._fused_metrics(total, tgt_total):
                loss_dict['l_total'], loss_dict['rmse'] = _ops.image_metrics(total, tgt_total, self.loss_funcs['l_test'].eps)
            else:
                loss_dict['l_total'] = self.loss_funcs['l_recon'](total, tgt_total).detach()
                loss_dict['rmse'] = self.loss_funcs['l_test'](total, tgt_total).detach()
        return loss_dict

    def train_batch(self, batch, grad_hook_mode=False):
        loss_dict = self._forward_backward(batch)

        if grad_hook_mode:  # do not update this model
            return

        self._logging(loss_dict)

        self._optimization()

    def _manifold_forward(self, batch):
        pre = getattr(self.models['backbone_diffuse'], '_paths_nhwc', None)
        if pre is not None:
            pre(batch)                       # shared NHWC copy of `paths`, made before the streams fork
        with _ops.on_branch(batch['paths'].device) as br:
            p_specular = self.models['backbone_specular'](batch)
        p_diffuse = self.models['backbone_diffuse'](batch)
        br.join(p_specular)
        return {'diffuse': p_diffuse, 'specular': p_specular}

    def _regress_forward(self, batch):
        return self.models['dncnn'](batch)

    def _backward(self, batch, out, p_buffers):
        assert 'radiance' in out
        assert 'diffuse' in out
        assert 'specular' in out

        total, diffuse, specular = out['radiance'], out['diffuse'], out['specular']
        loss_dict = {}
        tgt_total = crop_like(batch['target_total'], total)

        if self.train_branches:  # training diffuse and specular branches
            tgt_diffuse = crop_like(batch['target_diffuse'], diffuse)
            L_diffuse = _l1(self.loss_funcs['l_diffuse'], diffuse, tgt_diffuse)

            tgt_specular = crop_like(batch['target_specular'], specular)
            L_specular = _l1(self.loss_funcs['l_specular'], specular, tgt_specular)

            if self.manif_learn:
                p_buffer_diffuse = crop_like(p_buffers['diffuse'], diffuse)
                L_manif_diffuse = self.loss_funcs['l_manif'](p_buffer_diffuse, tgt_diffuse)
                L_diffuse = L_diffuse + L_manif_diffuse * self.w_manif

                p_buffer_specular = crop_like(p_buffers['specular'], specular)
                L_manif_specular = self.loss_funcs['l_manif'](p_buffer_specular, tgt_specular)
                L_specular = L_specular + L_manif_specular * self.w_manif

                loss_dict['l_manif_diffuse'] = L_manif_diffuse.detach()
                loss_dict['l_manif_specular'] = L_manif_specular.detach()

            # The reference logs `L_diffuse.detach()` and THEN adds the manifold term in place
            # (interfaces.py:221,227): the detached alias sees the add, so what it accumulates in
            # m_l_diffuse is L1 + w_manif * manifold.  Reproduced here on purpose.
            loss_dict['l_diffuse'] = L_diffuse.detach()
            loss_dict['l_specular'] = L_specular.detach()

            # the two no-grad metrics of the step (interfaces.py:240-249) depend on the forward only: their two launches are
            # enqueued BEFORE the backward passes, where they run beside them, instead of behind the last backward kernel in
            # front of the optimiser (20 us of every step's serial tail); the dictionary keeps the reference's key order
            metrics = None
            with torch.no_grad():
                if self._fused_metrics(total, tgt_total):      # l_total and rmse of the step in one pass
                    metrics = _ops.image_metrics(total, tgt_total, self.loss_funcs['l_test'].eps)

            # ONE engine run for both branch losses (the reference calls L_diffuse.backward() and then L_specular.backward(),
            # interfaces.py:237-238: the same gradients -- the two losses share no parameter and no graph node).  Two calls
            # SERIALISE the halves on the GPU: the second call's first nodes run on the launch stream behind everything the first
            # call enqueued, and the branch stream then waits for that point, so the specular backward started only when the
            # diffuse backward had finished; in one run the engine feeds both streams alternately and the halves overlap
            # like their forwards do (round 4: 12.9 -> 12.3 ms per step together with the P-buffer cut below, bit-identical)
            self._run_backward(L_diffuse, L_specular)

            with torch.no_grad():
                if metrics is not None:
                    loss_dict['l_total'], loss_dict['rmse'] = metrics
                    return loss_dict
                L_total = self.loss_funcs['l_recon'](total, tgt_total)
                loss_dict['l_total'] = L_total.detach()
        else:  # post-training the entire system (no manifold term: interfaces.py:243-246)
            L_total = _l1(self.loss_funcs['l_recon'], total, tgt_total)
            loss_dict['l_total'] = L_total.detach()
            self._run_backward(L_total)

        with torch.no_grad():
            loss_dict['rmse'] = self.loss_funcs['l_test'](total, tgt_total).detach()

        return loss_dict

    def _defer_scope(self):
        """``ops.deferred_wgrad_reduce()`` when every model declares that each of its parameters feeds exactly ONE autograd node per
        step (``single_use_parameters``: this package's KPCN and PathNet) -- a weight gradient whose reduction is deferred must not be
        summed with another producer's by the engine before the reduction has run; any other model keeps the reduction behind its GEMM."""
        import contextlib
        if all(getattr(m, 'single_use_parameters', False) for m in self.models.values()):
            return _ops.deferred_wgrad_reduce()
        return contextlib.nullcontext()

    def _run_backward(self, *losses):
        """``loss.backward()`` of every given loss in one engine run (interfaces.py:237-238, 246), or -- with the P-buffer cut of
        ``_forward_backward(cut=True)`` -- its first stage: down to the parameters of ``dncnn`` and to the PathNets' outputs,
        whose gradients stay in their ``.grad``."""
        raw = getattr(self, '_p_raw', None)
        if raw is None:
            with self._defer_scope():                   # (the small layers' slab reductions: one launch per stream at the end)
                if os.environ.get('WCMC_JOINT_BACKWARD', '1') == '0':      # A/B switch: the reference's one engine run per loss
                    for loss in losses:
                        loss.backward()
                else:
                    torch.autograd.backward(list(losses))
            return
        ins = [p for p in self.models['dncnn'].parameters() if p.requires_grad] + [t for t in raw.values() if t.requires_grad]
        # (no retain_graph: the engine frees what it walks, and with `inputs` it walks only the nodes above the P-buffers -- the
        # PathNets' part keeps its saved tensors for stage 2.  A retained graph would also keep the walked part's buffers, which
        # live in the hipGraph's memory pool, until Python's cycle collector gets to them: during some later capture)
        torch.autograd.backward(list(losses), inputs=ins)

    def _backward_stage2(self):
        """The second stage of a cut backward: through the PathNets, from the gradients stage 1 left on their outputs."""
        raw = getattr(self, '_p_raw', None)
        if raw is None:
            return
        ts = [t for t in raw.values() if t.grad is not None]
        if ts:
            dev = ts[0].device
            _ops.fork_all_streams(dev)
            torch.autograd.backward(ts, [t.grad for t in ts])
            _ops.join_all_streams(dev)
        self._p_raw = None

    def _fused_metrics(self, total, tgt_total):
        from .losses import RelativeMSE
        return (_is_plain_l1(self.loss_funcs['l_recon']) and type(self.loss_funcs['l_test']) is RelativeMSE and
                total.is_cuda and total.dim() == 4 and total.dtype == torch.float32 and total.shape == tgt_total.shape)

    def _logging(self, loss_dict):
        """ error handling """
        self.last_loss_dict = loss_dict
        keys = list(loss_dict)
        finite = torch.isfinite(torch.stack([loss_dict[k].reshape(()) for k in keys]))
        vals = [loss_dict[k] for k in keys]
        if self.fused_optim is not None:
            # Deferred: the fused optimiser is enqueued behind a device-side guard (no update when a
            # loss is non-finite) and the host check -- the only sync of the step -- comes after it.
            self._pending_finite = (keys, finite)
            ok = finite.all()                            # the reference raises BEFORE it logs (interfaces.py:254-267): a
            vals = [torch.where(ok, v, torch.zeros_like(v)) for v in vals]      # step that will raise adds nothing to the sums
        else:
            if self.grad_sync is not None and torch.distributed.is_available() and torch.distributed.is_initialized():
                # every rank must reach the gradient all-reduce or none: agree on the flag first (a rank that raised alone
                # would leave the others waiting in the collective)
                flag = finite.all().to(torch.float32).reshape(1)
                torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
                self._raise_if_nonfinite(keys, finite)
                if flag.item() == 0:
                    raise RuntimeError("Non-finite loss at train time on another rank.")
            else:
                self._raise_if_nonfinite(keys, finite)

        if self.grad_sync is not None:
            self.grad_sync(self.models)

        if self.fused_optim is None:
            for model_name in self.models:
                nn.utils.clip_grad_value_(self.models[model_name].parameters(), clip_value=1.0)

        """ logging """
        for key in loss_dict:
            if 'm_' + key not in self.m_losses:
                self.m_losses['m_' + key] = torch.tensor(0.0, device=loss_dict[key].device)
        sums = [self.m_losses['m_' + k] for k in keys]
        if all(a.shape == b.shape and a.device == b.device and a.dtype == b.dtype for a, b in zip(sums, vals)):
            torch._foreach_add_(sums, vals)              # the running sums in one launch (seven tiny ones otherwise)
        else:
            for a, b in zip(sums, vals):
                a += b

    @staticmethod
    def _raise_if_nonfinite(keys, finite):
        for key, ok in zip(keys, finite.tolist()):                   # one sync
            if not ok:
                raise RuntimeError("%s: Non-finite loss at train time." % (key))

    def _optimization(self):
        if self.fused_optim is not None:
            keys, finite = self._pending_finite
            guard = finite.all().to(torch.float32)
            # clip_grad_value_(1.0) + Adam, fused; the guard comes back reduced over the ranks (all skip or none)
            gguard = self.fused_optim.step(self.models, self.optims, guard=guard)
            if gguard is None:                       # no model had a gradient (all frozen): nothing was reduced or updated
                gguard = guard
            flags = torch.cat([finite.to(torch.float32), gguard.reshape(1)]).tolist()          # the step's one sync
            if flags[-1] == 0:
                self.fused_optim.rollback()          # the reference never reaches optim.step() (interfaces.py:254-271)
            for key, ok in zip(keys, flags[:-1]):
                if not ok:
                    raise RuntimeError("%s: Non-finite loss at train time." % (key))
            if flags[-1] == 0:
                raise RuntimeError("Non-finite loss at train time on another rank.")
            return
        for model_name in self.models:
            self.optims['optim_' + model_name].step()

    def to_eval_mode(self):
        for model_name in self.models:
            self.models[model_name].eval()
        self.m_losses['m_val'] = torch.tensor(0.0)

    def validate_batch(self, batch):
        p_buffers = None

        if self.use_llpm_buf:
            batch.pop('_wcmc_paths_nhwc', None)
            p_buffers = self._manifold_forward(batch)
            _, p_buffers = self._split(p_buffers, train=False)
            batch = self._assemble(batch, p_buffers)

        out = self._regress_forward(batch)

        return self._score_validation(batch, out), p_buffers

    def _score_validation(self, batch, out):
        """interfaces.py:296-300: running sum of the test loss (RelativeMSE) of the denoised radiance against the
        cropped target; the accumulator moves to the loss's device on first use."""
        radiance = out['radiance']
        err = self.loss_funcs['l_test'](radiance, crop_like(batch['target_total'], radiance)).detach()
        acc = self.m_losses['m_val']
        if acc == 0.0 and acc.device != err.device:
            acc = torch.tensor(0.0, device=err.device)
        self.m_losses['m_val'] = acc + err
        return radiance

    def get_epoch_summary(self, mode, norm):
        if mode == 'train':
            print('[][][]', end=' ')
            for key in self.m_losses:
                if key == 'm_val':
                    continue
                tr_l_tmp = self.m_losses[key] / (norm * 2)
                tr_l_tmp *= 1000
                print('%s: %.3fE-3' % (key, tr_l_tmp), end='\t')
                self.m_losses[key] = torch.tensor(0.0, device=self.m_losses[key].device)
            print('')
            return -1.0
        else:
            return self.m_losses['m_val'].item() / (norm * 2)


class KPCNRefInterface(KPCNInterface):
    """``interfaces.py:526-585``: vanilla KPCN whose inputs are extended by the clean per-branch targets
    (the "reference features" upper bound of the paper).  Same kernels as ``KPCNInterface``; only the batch
    assembly differs."""

    def __init__(self, models, optims, loss_funcs, args, visual=False, use_llpm_buf=False, manif_learn=False,
                 w_manif=0.1, train_branches=True):
        assert not use_llpm_buf
        assert not manif_learn
        super(KPCNRefInterface, self).__init__(models, optims, loss_funcs, args, visual, use_llpm_buf,
                                               manif_learn, w_manif, train_branches)

    def __str__(self):
        return 'KPCNRefInterface'

    @staticmethod
    def _with_targets(batch):
        new_batch = {k: batch[k] for k in _BATCH_KEYS}
        # 34 + 3 channels of leaf data (no gradient, 19 MB at the benchmark shape): a plain torch.cat
        new_batch['kpcn_diffuse_in'] = torch.cat([batch['kpcn_diffuse_in'], batch['target_diffuse']], 1)
        new_batch['kpcn_specular_in'] = torch.cat([batch['kpcn_specular_in'], batch['target_specular']], 1)
        return new_batch

    def _forward_backward(self, batch):
        dev = batch['kpcn_diffuse_in'].device
        _ops.fork_all_streams(dev)
        batch = self._with_targets(batch)
        self.models['dncnn'].zero_grad()
        out = self._regress_forward(batch)
        loss_dict = self._backward(batch, out, None)
        _ops.join_all_streams(dev)
        return loss_dict

    def validate_batch(self, batch):
        batch = self._with_targets(batch)
        out = self._regress_forward(batch)
        return self._score_validation(batch, out), None


class KPCNPreInterface(KPCNInterface):
    """``interfaces.py:588-750``: two-phase training.  ``manif_learn=True`` pre-trains the two PathNets on the
    manifold loss alone (full-size P-buffers against the full-size targets, no KPCN forward);
    ``manif_learn=False`` trains KPCN on top of the frozen PathNets (their gradients are still produced,
    as in the reference, but neither clipped nor stepped)."""

    def __init__(self, models, optims, loss_funcs, args, visual=False, manif_learn=False, w_manif=0.1,
                 train_branches=True):
        super(KPCNPreInterface, self).__init__(models, optims, loss_funcs, args, visual, True, manif_learn,
                                               w_manif, train_branches)

    def __str__(self):
        return 'KPCNPreInterface'

    def _trained(self, model_name):
        return ('backbone' in model_name) if self.manif_learn else ('dncnn' in model_name)

    def to_train_mode(self):
        for model_name in self.models:
            if 'dncnn' in model_name or 'backbone' in model_name:
                self.models[model_name].train(self._trained(model_name))
            assert 'optim_' + model_name in self.optims, \
                '`optim_%s`: an optimization algorithm is not defined.' % (model_name)

    def _forward_backward(self, batch):
        dev = batch['kpcn_diffuse_in'].device
        _ops.fork_all_streams(dev)
        self.models['backbone_diffuse'].zero_grad()
        self.models['backbone_specular'].zero_grad()
        if self.manif_learn:
            p_buffers = self._manifold_forward(batch)
            if self.iters % 1000 == 1 and not torch.cuda.is_current_stream_capturing():
                self._dump_pbuffers(p_buffers)
            loss_dict = self._backward(batch, None, p_buffers)
        else:
            self.models['dncnn'].zero_grad()
            p_buffers = self._manifold_forward(batch)
            batch = self._assemble(batch, p_buffers)           # interfaces.py:647-663 (no disentanglement here)
            out = self._regress_forward(batch)
            loss_dict = self._backward(batch, out, None)
        _ops.join_all_streams(dev)
        return loss_dict

    def _backward(self, batch, out, p_buffers):
        assert not out or 'radiance' in out
        assert not out or 'diffuse' in out
        assert not out or 'specular' in out
        loss_dict = {}
        if out:
            total, diffuse, specular = out['radiance'], out['diffuse'], out['specular']
            tgt_total = crop_like(batch['target_total'], total)

        if self.manif_learn:
            L_manif_diffuse = self.loss_funcs['l_manif'](p_buffers['diffuse'], batch['target_diffuse']) * self.w_manif
            L_manif_specular = self.loss_funcs['l_manif'](p_buffers['specular'], batch['target_specular']) * self.w_manif
            loss_dict['l_manif_diffuse'] = L_manif_diffuse.detach() / self.w_manif
            loss_dict['l_manif_specular'] = L_manif_specular.detach() / self.w_manif
            torch.autograd.backward([L_manif_diffuse, L_manif_specular])      # one engine run: the halves overlap (KPCNInterface._backward)
        elif self.train_branches:
            Ls = []
            for br, img in (('diffuse', diffuse), ('specular', specular)):      # interfaces.py:702-712, in that order
                L = _l1(self.loss_funcs['l_' + br], img, crop_like(batch['target_' + br], img))
                loss_dict['l_' + br] = L.detach()
                Ls.append(L)
            torch.autograd.backward(Ls)
            with torch.no_grad():
                loss_dict['l_total'] = self.loss_funcs['l_recon'](total, tgt_total).detach()
        else:
            L_total = _l1(self.loss_funcs['l_recon'], total, tgt_total)
            loss_dict['l_total'] = L_total.detach()
            L_total.backward()
        return loss_dict

    def _logging(self, loss_dict):
        keys = list(loss_dict)
        finite = torch.isfinite(torch.stack([loss_dict[k].reshape(()) for k in keys]))
        self._raise_if_nonfinite(keys, finite)
        if self.grad_sync is not None:
            self.grad_sync({n: m for n, m in self.models.items() if self._trained(n)})
        for model_name in self.models:                       # interfaces.py:729-735: only the phase's models
            if self._trained(model_name):
                nn.utils.clip_grad_value_(self.models[model_name].parameters(), clip_value=1.0)
        for key in loss_dict:
            if 'm_' + key not in self.m_losses:
                self.m_losses['m_' + key] = torch.tensor(0.0, device=loss_dict[key].device)
            self.m_losses['m_' + key] += loss_dict[key]

    def _optimization(self):
        for model_name in self.models:
            if self._trained(model_name):
                self.optims['optim_' + model_name].step()


# ------------------------------------------------------------------------------------------------------------------
# the glue of the two sample-based denoisers (train_sbmc.py / train_lbmc.py).  The base
# denoisers themselves (sbmc.Multisteps, layerdenoise's LayerNet) are external to the reference tree; what lives
# here is everything interfaces.py:336-523 and :753-839 do around them: the single PathNet backbone, the
# disentanglement slicing, the per-sample feature assembly (one HIP kernel, ops.sample_features_cat), the
# reconstruction + manifold loss, gradient-NORM clipping (1000 / 250) and the running sums.
class SBMCInterface(BaseInterface):
    GRAD_NORM_CLIP = 1000.0                   # interfaces.py:452-456

    def __init__(self, models, optims, loss_funcs, args, visual=False, use_llpm_buf=False, manif_learn=False,
                 w_manif=0.1, use_sbmc_buf=True, disentangle="m11r11"):
        if manif_learn:
            assert 'backbone' in models, "argument `models` dictionary should contain `'backbone'` key."
        assert 'dncnn' in models, "argument `models` dictionary should contain `'dncnn'` key."
        if manif_learn:
            assert 'l_manif' in loss_funcs
        assert 'l_recon' in loss_funcs
        assert 'l_test' in loss_funcs
        assert disentangle in _OPTIONS
        super(SBMCInterface, self).__init__(models, optims, loss_funcs, args, visual, use_llpm_buf, manif_learn, w_manif)
        self.disentangle, self.use_sbmc_buf = disentangle, use_sbmc_buf

    def __str__(self):
        return 'SBMCInterface'

    def to_train_mode(self):
        for name, model in self.models.items():
            model.train()
            assert 'optim_' + name in self.optims, '`optim_%s`: an optimization algorithm is not defined.' % (name)

    def preprocess(self, batch=None):
        for key in ('target_image', 'radiance', 'features') + (('paths',) if self.use_llpm_buf else ()):
            assert key in batch
        self.iters += 1

    def _manifold_forward(self, batch):
        return self.models['backbone'](batch)

    def _regress_forward(self, batch):
        return self.models['dncnn'](batch)

    def _split(self, p_buffer, train):
        """interfaces.py:378-388 / :484-489: (what the manifold loss sees, what the denoiser sees)."""
        c = p_buffer.shape[2]
        assert c >= 2
        lo, hi = p_buffer[:, :, :c // 2, ...], p_buffer[:, :, c // 2:, ...]
        regress = lo if self.disentangle in ('m10r01', 'm11r01') else p_buffer
        if not train:
            return None, regress
        return (p_buffer if self.disentangle in ('m11r11', 'm11r01') else hi), regress

    @staticmethod
    def _assemble(batch, p_regress):
        """interfaces.py:390-403: features' = cat([features, P, P.var(1).mean(1) / S (detached, per sample)], 2)."""
        return {'target_image': batch['target_image'], 'radiance': batch['radiance'],
                'features': _ops.sample_features_cat(batch['features'], p_regress)}

    def _dump_pbuffer(self, p_buffer):
        if not os.path.isdir('../LLPM_results'):     # interfaces.py:371-374 (debug PNG every 1000 iterations)
            return
        import numpy as np
        import matplotlib.pyplot as plt
        pimg = np.mean(np.transpose(p_buffer.detach().cpu().numpy()[0, :, :3, ...], (2, 3, 0, 1)), 2)
        plt.imsave('../LLPM_results/pbuf_%s.png' % (self.args.model_name), np.clip(pimg, 0.0, 1.0))

    def train_batch(self, batch, grad_hook_mode=False):
        out_manif = None
        if self.use_llpm_buf:
            self.models['backbone'].zero_grad()
            p_buffer = self._manifold_forward(batch)
            if self.iters % 1000 == 1:
                self._dump_pbuffer(p_buffer)
            out_manif, p_regress = self._split(p_buffer, train=True)
            batch = self._assemble(batch, p_regress)
        self.models['dncnn'].zero_grad()
        out = self._regress_forward(batch)
        loss_dict = self._backward(batch, out, out_manif)
        if grad_hook_mode:  # do not update this model
            return
        self._logging(loss_dict)
        self._optimization()

    def _backward(self, batch, out, p_buffer):
        loss_dict = {}
        tgt_total = crop_like(batch['target_image'], out)
        L_total = self.loss_funcs['l_recon'](out, tgt_total)
        if self.manif_learn:
            L_manif = self.loss_funcs['l_manif'](crop_like(p_buffer, out), tgt_total)
            loss_dict['l_manif'], loss_dict['l_recon'] = L_manif.detach(), L_total.detach()
            # in place, as interfaces.py:427: `l_recon` above aliases this tensor (detach shares storage), so the
            # reference's logged l_recon equals l_total -- kept
            L_total += L_manif * self.w_manif
        loss_dict['l_total'] = L_total.detach()
        L_total.backward()
        with torch.no_grad():
            loss_dict['rmse'] = self.loss_funcs['l_test'](out, tgt_total).detach()
        return loss_dict

    def _logging(self, loss_dict):
        for key in loss_dict:
            if not torch.isfinite(loss_dict[key]).all():
                raise RuntimeError("%s: Non-finite loss at train time." % (key))
        for name, model in self.models.items():
            params = [p for p in model.parameters() if p.grad is not None]
            if params and all(p.grad.is_cuda and p.grad.dtype == torch.float32 and p.grad.is_contiguous() for p in params) and len(params) <= 96:
                actual = _ops.clip_grad_norm_(params, self.GRAD_NORM_CLIP)          # three HIP launches (wcmc_grad_norm_clip)
            else:
                actual = nn.utils.clip_grad_norm_(model.parameters(), max_norm=self.GRAD_NORM_CLIP)
            if actual > self.GRAD_NORM_CLIP:
                print("Clipped %s gradients %f -> %f" % (name, self.GRAD_NORM_CLIP, actual))
        for key in loss_dict:
            if 'm_' + key not in self.m_losses:
                self.m_losses['m_' + key] = torch.tensor(0.0, device=loss_dict[key].device)
            self.m_losses['m_' + key] += loss_dict[key]

    def _optimization(self):
        for name in self.models:
            self.optims['optim_' + name].step()

    def to_eval_mode(self):
        for model in self.models.values():
            model.eval()
        self.m_losses['m_val'] = torch.tensor(0.0)

    def validate_batch(self, batch):
        p_buffer = None
        if self.use_llpm_buf:
            _, p_buffer = self._split(self._manifold_forward(batch), train=False)
            batch = self._assemble(batch, p_buffer)
        out = self._regress_forward(batch)
        self._score(self.loss_funcs['l_test'](out, crop_like(batch['target_image'], out)))
        return out, p_buffer

    def _score(self, L_total):
        if self.m_losses['m_val'] == 0.0 and self.m_losses['m_val'].device != L_total.device:
            self.m_losses['m_val'] = torch.tensor(0.0, device=L_total.device)
        self.m_losses['m_val'] += L_total.detach()

    def get_epoch_summary(self, mode, norm):
        if mode != 'train':
            return self.m_losses['m_val'].item() / (norm * 2)
        print('[][][]', end=' ')
        for key in self.m_losses:
            if key == 'm_val':
                continue
            print('%s: %.3fE-3' % (key, self.m_losses[key] / (norm * 2) * 1000), end='\t')
            self.m_losses[key] = torch.tensor(0.0, device=self.m_losses[key].device)
        print('')
        return -1.0


class LBMCInterface(SBMCInterface):
    """interfaces.py:753-839: SBMCInterface without the visual / sbmc-buffer switches and with the layer-based
    denoiser's clamp, GRADIENT_CLAMP_N = 0.25 * 1000."""
    GRAD_NORM_CLIP = 250.0

    def __init__(self, models, optims, loss_funcs, args, use_llpm_buf=False, manif_learn=False, w_manif=0.1,
                 disentangle='m11r11'):
        super(LBMCInterface, self).__init__(models, optims, loss_funcs, args, False, use_llpm_buf, manif_learn,
                                            w_manif, False, disentangle)

    def __str__(self):
        return 'LBMCInterface'
