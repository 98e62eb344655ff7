"""``support/losses.py`` of the reference on the MI355X path.

``FeatureMSE`` (the path-disentangling loss, ``losses.py:9-113``) runs as the HIP op
``wcmc_feature_mse_*``; the pairing permutations are drawn exactly as the reference draws
them -- ``torch.randperm`` on the global CPU generator, patch then batch
(``losses.py:35,50,105-109``) -- and handed to the kernel as explicit inputs, so
``torch.manual_seed`` reproduces the reference's pairs.  ``RelativeMSE`` (``losses.py:245-264``)
is evaluated on the (B,3,92,92) outputs.
"""
import torch

from .. import ops

__all__ = ["GlobalRelativeSimilarityLoss", "FeatureMSE", "RelativeMSE", "SMAPE", "TonemappedMSE",
           "TonemappedRelativeMSE"]


class FeatureMSE(torch.nn.Module):
    """Feature Mean-Squared Error. Path disentangling loss"""

    def __init__(self, color='rgb', non_local=True, rng='cpu', pairing='local', process_group=None):
        """pairing (one process per GPU; the reference is single-process ``nn.DataParallel``, ``train_kpcn.py:266-269``, whose
           losses see the GATHERED global batch on one GPU):
             'local'  -- every rank pairs inside its own patches (default; the same estimator on other sample pairs);
             'global' -- the reference's semantics: the cropped P-buffers and references of all ranks are all-gathered
                         (7.3 MB per rank at the benchmark shape) and the loss of the GLOBAL batch is evaluated -- the
                         intra-batch permutation spans ``world * B * S * H * W`` rows (``losses.py:48-61``) -- with one set
                         of permutations for all ranks (rank 0's draw, broadcast).  The value returned is the global loss on
                         every rank; its gradient reaches this rank's rows only and is scaled by ``world`` so that the
                         gradient MEAN over ranks (``FusedClipAdam``, ``average_gradients``) is the gradient of the global
                         loss.  Every rank evaluates all ``world * N`` rows (the op is HBM-bound: ~0.1 ms per call and rank).
        rng: where the pairing permutations are drawn.
             'cpu'    -- torch.randperm on the global CPU generator, patch then batch: bit-identical pairs
                         to the reference under torch.manual_seed (losses.py:35,50); costs ~23 ms of host
                         time per 541,696-row permutation on the MI355X host.
             'device' -- a keyed bijection generated on the GPU (ops.random_permutation, no sort): random pairs from another
                         random stream."""
        super(FeatureMSE, self).__init__()
        if color != 'rgb':
            raise NotImplementedError("FeatureMSE(color=%r): only 'rgb' is on the KPCN-Manifold path "
                                      "(no caller of the reference uses 'hls')" % (color,))
        assert rng in ('cpu', 'device') and pairing in ('local', 'global')
        self.color = color
        self.non_local = non_local
        self.rng = rng
        self.pairing = pairing
        self.group = process_group
        self.last_perms = None
        self.static_perms = None        # queue of (idx_patch, idx_batch) device tensors (graph replay)
        self.check_finite = True
        print('FeatureMSE locality: %s' % ('Non-local' if non_local else 'Local'))

    def draw_permutations(self, b, s, h, w, device=None):
        if self.rng == 'device':
            return (ops.random_permutation(s * h * w, device),
                    ops.random_permutation(b * s * h * w, device) if self.non_local else None)
        idx_patch = torch.randperm(s * h * w)
        idx_batch = torch.randperm(b * s * h * w) if self.non_local else None
        return idx_patch, idx_batch

    def _world(self):
        import torch.distributed as dist
        return dist.get_world_size(self.group) if (self.pairing == 'global' and dist.is_available() and dist.is_initialized()) else 1

    def _forward_global(self, p_buffer, ref, perms, world):
        """The loss of the gathered global batch (see ``pairing``)."""
        import torch.distributed as dist
        b, s, c, h, w = p_buffer.shape
        dev = p_buffer.device
        rank = dist.get_rank(self.group)
        pl, rl = p_buffer.contiguous(), ref.contiguous()
        with torch.no_grad():
            pg = [torch.empty_like(pl) for _ in range(world)]
            rg = [torch.empty_like(rl) for _ in range(world)]
            dist.all_gather(pg, pl.detach(), group=self.group)
            dist.all_gather(rg, rl.detach(), group=self.group)
        pg[rank] = pl                                      # this rank's rows stay in the autograd graph
        p_all, r_all = torch.cat(pg, 0), torch.cat(rg, 0)
        if perms is None:                                  # one draw for all ranks: rank 0's
            if self.rng == 'device':
                seeds = torch.randint(0, 2 ** 62, (2,), dtype=torch.int64)
                seeds = seeds.to(dev) if dist.get_backend(self.group) == 'nccl' else seeds
                dist.broadcast(seeds, 0, group=self.group)
                sd = seeds.tolist()
                perms = (ops.random_permutation(s * h * w, dev, seed=sd[0]),
                         ops.random_permutation(world * b * s * h * w, dev, seed=sd[1]) if self.non_local else None)
            else:
                ip = torch.randperm(s * h * w)
                ib = torch.randperm(world * b * s * h * w) if self.non_local else None
                on = dev if dist.get_backend(self.group) == 'nccl' else torch.device('cpu')
                ip = ip.to(on)
                dist.broadcast(ip, 0, group=self.group)
                if ib is not None:
                    ib = ib.to(on)
                    dist.broadcast(ib, 0, group=self.group)
                perms = (ip, ib)
        self.last_perms = perms
        ip = perms[0].to(dev, non_blocking=True)
        ib = perms[1].to(dev, non_blocking=True) if perms[1] is not None else None
        loss = ops.feature_mse(p_all, r_all, ip, ib)
        # value: the global loss; gradient: world x d(global loss)/d(this rank's rows) -- the ranks' gradients are averaged later
        return loss.detach() + float(world) * (loss - loss.detach())

    def forward(self, p_buffer, ref, perms=None):
        """p_buffer (B,S,C,H,W) embedded paths, ref (B,3,H,W) reference radiance -> 0-d loss."""
        b, s, c, h, w = p_buffer.shape
        dev = p_buffer.device
        world = self._world()
        if world > 1:
            loss = self._forward_global(p_buffer, ref, perms, world)
            if self.check_finite and not torch.isfinite(loss.detach()):
                raise RuntimeError("Infinite loss at train time.")
            return loss
        if self.static_perms is not None:                 # pre-drawn device permutations, call order preserved
            idx_patch, idx_batch = self.static_perms[self._static_i % len(self.static_perms)]
            self._static_i += 1
        elif perms is not None:
            idx_patch, idx_batch = perms
        else:
            idx_patch, idx_batch = self.draw_permutations(b, s, h, w, dev)
        self.last_perms = (idx_patch, idx_batch)
        ip = idx_patch.to(dev, non_blocking=True)
        ib = idx_batch.to(dev, non_blocking=True) if idx_batch is not None else None
        loss = ops.feature_mse(p_buffer, ref, ip, ib)
        # A non-finite P or reference poisons every displacement it takes part in, so the check
        # the reference makes on the inputs (losses.py:99-102) is made on the scalar instead.
        if self.check_finite and not torch.isfinite(loss.detach()):
            raise RuntimeError("Infinite loss at train time.")
        return loss

    _static_i = 0


class GlobalRelativeSimilarityLoss(torch.nn.Module):
    """Global Relative Similarity Loss (``losses.py:116-211``, ``--manif_loss GRS``): same pairings and
    displacements as FeatureMSE, ``(logsumexp(alpha*[d_p, d_b, -d_p, -d_b, 0]) - log(1+4N)) / sqrt(alpha)``.
    Permutations are drawn like the reference: patch then batch on the global CPU generator."""

    def __init__(self, alpha=2, color='rgb', rng='cpu'):
        super(GlobalRelativeSimilarityLoss, self).__init__()
        assert rng in ('cpu', 'device')
        self.color = color
        self.alpha = alpha
        self.rng = rng
        self.last_perms = None

    def forward(self, p_buffer, ref, perms=None):
        b, s, c, h, w = p_buffer.shape
        dev = p_buffer.device
        if perms is None:
            if self.rng == 'device':
                perms = (ops.random_permutation(s * h * w, dev), ops.random_permutation(b * s * h * w, dev))
            else:
                perms = (torch.randperm(s * h * w), torch.randperm(b * s * h * w))
        self.last_perms = perms
        loss = ops.grs_loss(p_buffer, ref, perms[0].to(dev, non_blocking=True), perms[1].to(dev, non_blocking=True),
                            float(self.alpha))
        if not torch.isfinite(loss.detach()):       # losses.py:192-195, checked on the scalar
            raise RuntimeError("Infinite loss at train time.")
        return loss


class RelativeMSE(torch.nn.Module):
    """0.5 * mean((im - ref)^2 / (ref^2 + eps))  (``losses.py:245-264``)."""

    def __init__(self, eps=1e-2):
        super(RelativeMSE, self).__init__()
        self.eps = eps

    def forward(self, im, ref):
        # the score of a validation batch / the logged rmse of a step carries no gradient: one HIP pass
        # (wcmc_image_loss_fwd); with a gradient (nobody trains on it in the reference) the torch expression below
        if im.is_cuda and im.dim() == 4 and im.dtype == torch.float32 and ref.shape == im.shape and \
                not (torch.is_grad_enabled() and (im.requires_grad or ref.requires_grad)):
            return ops.relative_mse(im, ref, self.eps)
        mse = torch.pow(im - ref, 2)
        return 0.5 * torch.mean(mse / (torch.pow(ref, 2) + self.eps))


def _hip_image(im, ref):
    """The HIP loss kernels take fp32 (N,C,H,W) CUDA images whose reference carries no gradient (every call site of the reference:
    ``interfaces.py:418-421, 815-818``); anything else -- the CPU host-logic tests -- takes the torch expression."""
    return (im.is_cuda and im.dim() == 4 and im.dtype == torch.float32 and ref.shape == im.shape and ref.dtype == torch.float32
            and not ref.requires_grad)


def _reinhard(im):
    """``losses.py:234-242``: Reinhard tone map of the clamped image."""
    im = torch.clamp(im, min=0)
    return im / (1 + im)


class SMAPE(torch.nn.Module):
    """mean(|im - ref| / (eps + |im| + |ref|)) with a gradient-free denominator (``losses.py:267-284``; LBMC's loss)."""

    def __init__(self, eps=1e-2):
        super(SMAPE, self).__init__()
        self.eps = eps

    def forward(self, im, ref):
        if _hip_image(im, ref):
            return ops.image_loss2(im, ref, "smape", self.eps)
        scale = self.eps + im.detach().abs() + ref.detach().abs()
        return torch.mean((im - ref).abs() / scale)


class TonemappedMSE(torch.nn.Module):
    """0.5 * mean((T(im) - T(ref))^2), T = Reinhard (``losses.py:287-302``)."""

    def __init__(self, eps=1e-2):
        super(TonemappedMSE, self).__init__()
        self.eps = eps

    def forward(self, im, ref):
        if _hip_image(im, ref):
            return ops.image_loss2(im, ref, "tonemapped_mse", self.eps)
        return 0.5 * torch.mean(torch.pow(_reinhard(im) - _reinhard(ref), 2))


class TonemappedRelativeMSE(torch.nn.Module):
    """RelativeMSE of the tone-mapped images (``losses.py:305-320``; SBMC's loss)."""

    def __init__(self, eps=1e-2):
        super(TonemappedRelativeMSE, self).__init__()
        self.eps = eps

    def forward(self, im, ref):
        if _hip_image(im, ref):
            return ops.image_loss2(im, ref, "tonemapped_relative_mse", self.eps)
        im, ref = _reinhard(im), _reinhard(ref)
        return 0.5 * torch.mean(torch.pow(im - ref, 2) / (torch.pow(ref, 2) + self.eps))
