"""The host loop of ``train_kpcn.py`` over the MI355X interfaces.

Same entry points, arguments and side effects as the reference script so that its callers keep working:

  * ``train_epoch_kpcn`` / ``validate_kpcn`` / ``train``  -- ``train_kpcn.py:37-161``: one epoch over
    ``dataloaders['train']`` (``preprocess`` + ``train_batch`` per interface), the per-epoch ``latest_<name>.pth``, the
    validation pass every ``val_epoch`` epochs with checkpoint-on-best (``best_err``), the scheduler step;
  * ``init_model``                                         -- ``train_kpcn.py:191-338``: the interface (KPCN / Ref / Pre), one per
    point of the ``lr_pnet x pnet_out_size x w_manif`` grid, weights and optimiser state restored from ``<save>/<name>.pth``
    when ``--start_epoch`` is not 0 (``support/checkpoint.py``: the reference's file format);
  * ``build_parser`` / ``check_args``                      -- the flag surface of ``train_kpcn.py:376-442`` +
    ``support/utils.py:69-100`` (``BasicArgumentParser``), with the same defaults and the same argument errors.

What differs, on purpose:

  * one process per GPU (``torch.distributed`` over RCCL, ``wcmc_amd.distributed``) instead of ``nn.DataParallel``:
    launch with ``python -m torch.distributed.run --nproc-per-node N -m wcmc_amd.train_kpcn ...``; every rank steps on
    its own shard, gradients are summed inside the fused clip + Adam (``wcmc_amd.optim.FusedClipAdam``), rank 0 saves;
  * ``--graph`` replays one hipGraph per step (``wcmc_amd.graph.GraphedTrainStep``) when the loader keeps its shapes;
  * the reading of the authors' dataset files (``support/datasets.py:MSDenoiseDataset``) is out of scope (SURVEY.md 8,
    DESIGN.md section 1): ``init_data`` feeds ``--synthetic N`` batches per epoch with the dataset's schema
    (``wcmc_amd.synthetic``), or any iterable of batch dictionaries handed to ``train`` by the caller;
  * no visdom (``--visual`` is accepted and ignored), no tqdm.
"""
import argparse
import itertools
import os
import time

import torch

from . import KPCN
from . import distributed as wd
from .optim import FusedClipAdam
from .support import checkpoint as ckpt
from .support.interfaces import KPCNInterface, KPCNPreInterface, KPCNRefInterface
from .support.losses import FeatureMSE, GlobalRelativeSimilarityLoss, RelativeMSE
from .support.networks import PathNet
from .synthetic import make_batch

BS_VAL = 4          # validation batch size (train_kpcn.py:374)
DNCNN_IN = 34       # kpcn_*_in channels of the vanilla buffers (datasets.py:215-216)
PNET_IN = 36        # path-descriptor channels (datasets.py:343-347)


# ------------------------------------------------------------------------------------------------- epoch functions
def _to_device(batch, device):
    for k in batch:
        if isinstance(batch[k], torch.Tensor):
            batch[k] = batch[k].to(device, non_blocking=True)
    return batch


def train_epoch_kpcn(epoch, interfaces, dataloaders, params, args):
    assert 'train' in dataloaders, "argument `dataloaders` dictionary should contain `'train'` key."
    assert 'data_device' in params, "argument `params` dictionary should contain `'data_device'` key."
    print('[][] Epoch %d' % (epoch))
    for itf in interfaces:
        itf.to_train_mode()
    steps = params.setdefault('graphed_steps', {})
    n = 0
    for batch in dataloaders['train']:
        batch = _to_device(batch, params['data_device'])
        for i, itf in enumerate(interfaces):
            if getattr(args, 'graph', False):
                if i not in steps:
                    from .graph import capture_validated
                    overlap = getattr(args, 'overlap_allreduce', False)
                    two = bool(getattr(itf, 'halves_supported', lambda: False)()) and not overlap and not getattr(args, 'one_graph', False)
                    # (each capture is timed and re-made if it is more than 5 % slower than the best this process has seen)
                    steps[i] = capture_validated(itf, batch, defer_check=getattr(args, 'defer_check', True),
                                                 overlap_allreduce=overlap, two_stream=two)
                    kick = getattr(dataloaders['train'], 'kick', None)      # support/loader.py: pace the producer thread
                    if kick is not None and i == len(interfaces) - 1:
                        steps[i].after_enqueue = kick
                steps[i](batch)
            else:
                itf.preprocess(batch)
                itf.train_batch(batch)
        n += 1
    for step in steps.values():
        step.flush()                                      # (--defer_check: the last step's non-finite check)
    if not args.visual:
        for itf in interfaces:
            itf.get_epoch_summary(mode='train', norm=n)


def validate_kpcn(epoch, interfaces, dataloaders, params, args):
    assert 'val' in dataloaders, "argument `dataloaders` dictionary should contain `'val'` key."
    assert 'data_device' in params, "argument `params` dictionary should contain `'data_device'` key."
    print('[][] Validation (epoch %d)' % (epoch))
    for itf in interfaces:
        itf.to_eval_mode()
    n = 0
    with torch.no_grad():
        for batch in dataloaders['val']:
            batch = _to_device(batch, params['data_device'])
            for itf in interfaces:
                itf.validate_batch(batch)
            n += 1
    return [itf.get_epoch_summary(mode='eval', norm=n) for itf in interfaces]


def train(interfaces, dataloaders, params, args):
    print('[] Experiment: `{}`'.format(args.desc))
    print('[] # of interfaces : %d' % (len(interfaces)))
    print('[] Model training start...')
    rank0 = params.get('rank', 0) == 0
    for epoch in range(args.start_epoch, args.num_epoch):
        if len(interfaces) != 1:
            raise NotImplementedError('Multiple interfaces')          # as train_kpcn.py:96-99
        save_fn = args.model_name + '.pth'
        start_time = time.time()
        train_epoch_kpcn(epoch, interfaces, dataloaders, params, args)
        print('[][] Elapsed time: %d' % (time.time() - start_time))
        for itf in interfaces:
            if not args.not_save and rank0:
                ckpt.save_checkpoint(os.path.join(args.save, 'latest_' + save_fn), itf, epoch, args, _picklable(params))
        if epoch % args.val_epoch == args.val_epoch - 1:
            print('[][] Validation')
            summaries = validate_kpcn(epoch, interfaces, dataloaders, params, args)
            for i, itf in enumerate(interfaces):
                if summaries[i] < itf.best_err:
                    itf.best_err = summaries[i]
                    if not args.not_save and rank0:
                        ckpt.save_checkpoint(os.path.join(args.save, save_fn), itf, epoch, args, _picklable(params))
                        print('[][] Model %s saved at epoch %d.' % (save_fn, epoch))
                print('[][] Model {} RelMSE: {:.3f}e-3 \t Best RelMSE: {:.3f}e-3'.format(
                    save_fn, summaries[i] * 1000, itf.best_err * 1000))
        for key in params:
            if 'sched_' in key:
                params[key].step()
    print('[] Training complete!')


def _picklable(params):
    return {k: v for k, v in params.items() if k not in ('graphed_steps', 'group')}


# ------------------------------------------------------------------------------------------------- data
class SyntheticLoader:
    """``len``-able iterable of batch dictionaries with the dataset's schema (``wcmc_amd.synthetic.make_batch``): stands in
    for ``DataLoader(MSDenoiseDataset(...))`` (``train_kpcn.py:167-189``), whose file reading is out of scope."""

    def __init__(self, n_batches, batch_size, patch, spp, use_llpm, seed, device):
        self.n, self.args, self.seed = n_batches, (batch_size, spp, patch), seed
        self.use_llpm, self.device = use_llpm, device

    def __len__(self):
        return self.n

    def __iter__(self):
        b, s, h = self.args
        for i in range(self.n):
            yield make_batch(b, s, h, seed=self.seed + i, device=self.device, use_llpm=self.use_llpm)


def init_data(args, device, rank=0, world=1):
    n = max(1, args.synthetic)
    train = SyntheticLoader(n, args.batch_size, args.patch_size, 8, args.use_llpm_buf, 1000 * (rank + 1), device)
    val = SyntheticLoader(max(1, n // 4), BS_VAL, args.patch_size, 8, args.use_llpm_buf, 7_000_000 + 1000 * rank, device)
    sizes = {'dncnn_in_size': DNCNN_IN + (1 + 3 + 1 if args.use_llpm_buf else 0), 'pnet_in_size': PNET_IN, 'pnet_out_size': 3}
    return sizes, {'train': train, 'val': val}


# ------------------------------------------------------------------------------------------------- models
def init_model(sizes, args, device, group=None):
    """``train_kpcn.py:191-338``.  sizes: ``dncnn_in_size`` / ``pnet_in_size`` / ``pnet_out_size`` of the dataset."""
    interfaces = []
    grid = list(itertools.product(args.lr_pnet, args.pnet_out_size, args.w_manif))
    for lr_pnet, pnet_out_size, w_manif in grid:
        models = {}
        if len(grid) == 1:
            model_fn = os.path.join(args.save, args.model_name + '.pth')
        else:
            model_fn = os.path.join(args.save, '%s_lp%f_pos%d_wgt%f.pth' % (args.model_name, lr_pnet, pnet_out_size, w_manif))
        assert args.start_epoch != 0 or not os.path.isfile(model_fn), 'Model %s already exists.' % (model_fn)
        is_pretrained = args.start_epoch != 0 and os.path.isfile(model_fn)
        ck = ckpt.load_checkpoint(model_fn) if is_pretrained else None
        if ck is not None:
            ckpt.precision_note(ck)
        # upstream sbmc's ConvChain defaults to weight normalisation (which sbmc.KPCN switches off and PathNet,
        # support/networks.py:18-24, does not): PathNets are built that way (`weight_g` / `weight_v` per layer) unless
        # --no_pathnet_weight_norm is given; on a resume the checkpoint's own layout decides
        wn = bool(getattr(args, 'pathnet_weight_norm', True))
        if ck is not None and 'state_dict_backbone_diffuse' in ck:
            ck_wn = any(k.endswith('weight_g') for k in ck['state_dict_backbone_diffuse'])
            if ck_wn != wn:
                print('The checkpoint holds %s PathNets: building them that way.'
                      % ('weight-normalised (weight_g / weight_v)' if ck_wn else 'plain-weight (weight_norm=False)'))
            wn = ck_wn
        if args.use_llpm_buf:
            half = args.disentangle in ('m10r01', 'm11r01')
            n_in = sizes['dncnn_in_size'] - sizes['pnet_out_size'] + (pnet_out_size // 2 if half else pnet_out_size)
            models['dncnn'] = KPCN(n_in)
            print('Initialize KPCN for path descriptors (# of input channels: %d).' % (n_in))
            models['backbone_diffuse'] = PathNet(ic=sizes['pnet_in_size'], outc=pnet_out_size, weight_norm=wn)
            models['backbone_specular'] = PathNet(ic=sizes['pnet_in_size'], outc=pnet_out_size, weight_norm=wn)
        else:
            n_in = sizes['dncnn_in_size'] + (3 if args.kpcn_ref else 0)
            models['dncnn'] = KPCN(n_in)
            print('Initialize KPCN for vanilla buffers (# of input channels: %d).' % (n_in))
        if is_pretrained:
            ckpt.restore_models(ck, models)
            print('Pretraining weights are loaded.')
        else:
            print('Train models from scratch.')
        for name in models:
            models[name] = models[name].to(device)
        lrs = {'optim_' + name: (args.lr_dncnn if name == 'dncnn' else lr_pnet) for name in models}
        optims = {key: torch.optim.Adam(models[key[len('optim_'):]].parameters(), lr=lr) for key, lr in lrs.items()}
        if is_pretrained:
            ckpt.restore_optims(ck, optims, lrs, lr_ckpt=args.lr_ckpt)
        loss_funcs = {'l_diffuse': torch.nn.L1Loss(), 'l_specular': torch.nn.L1Loss(), 'l_recon': torch.nn.L1Loss(),
                      'l_test': RelativeMSE()}
        if args.manif_learn:
            if args.manif_loss == 'FMSE':
                loss_funcs['l_manif'] = FeatureMSE(non_local=not args.local, rng=args.pairing_rng, pairing=args.pairing,
                                                   process_group=group)
                print('Manifold loss: FeatureMSE')
            else:
                loss_funcs['l_manif'] = GlobalRelativeSimilarityLoss(rng=args.pairing_rng)
                print('Manifold loss: Global Relative Similarity')
        else:
            print('Manifold loss: None (i.e., ablation study)')
        if args.kpcn_ref:
            itf = KPCNRefInterface(models, optims, loss_funcs, args, train_branches=args.train_branches)
        elif args.kpcn_pre:
            itf = KPCNPreInterface(models, optims, loss_funcs, args, manif_learn=args.manif_learn,
                                   train_branches=args.train_branches)
        else:
            itf = KPCNInterface(models, optims, loss_funcs, args, visual=args.visual, use_llpm_buf=args.use_llpm_buf,
                                manif_learn=args.manif_learn, w_manif=w_manif, train_branches=args.train_branches,
                                disentanglement_option=args.disentangle)
            order = ('dncnn', 'backbone_diffuse', 'backbone_specular') if getattr(args, 'overlap_allreduce', False) else None
            itf.fused_optim = FusedClipAdam(models, optims, process_group=group, order=order)      # clip + Adam (+ RCCL sum), fused
            if group is not None:
                for fl in itf.fused_optim.flats.values():
                    torch.distributed.broadcast(fl.flat, 0, group=group)              # every rank starts from rank 0's weights
        if itf.fused_optim is None and group is not None:
            wd.broadcast_parameters(models, group=group)
            itf.grad_sync = lambda ms, g=group: wd.average_gradients(ms, group=g)
        if is_pretrained and args.best_err is not None:
            print('Use the checkpoint best error %.3e' % (args.best_err))
            itf.best_err = args.best_err
        interfaces.append(itf)
    if not os.path.isdir(args.save):
        os.makedirs(args.save, exist_ok=True)
    return interfaces, {'plots': {}, 'data_device': device}


# ------------------------------------------------------------------------------------------------- command line
def build_parser():
    p = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    # support/utils.py:69-100 (BasicArgumentParser)
    p.add_argument('--sbmc', action='store_true')
    p.add_argument('--p_buf', action='store_true')
    p.add_argument('--model_name', type=str, default='tSUNet', help='name of the model.')
    p.add_argument('--data_dir', type=str, default='./data', help='directory of dataset (unused: see --synthetic)')
    p.add_argument('--visual', action='store_true', help='accepted for compatibility; there is no visdom here')
    p.add_argument('-b', '--batch_size', type=int, default=64, help='batch size (per GPU).')
    p.add_argument('-e', '--num_epoch', type=int, default=100, help='number of epochs.')
    p.add_argument('-v', '--val_epoch', type=int, default=1, help='validate the model every val_epoch epoch.')
    p.add_argument('--vis_iter', type=int, default=4)
    p.add_argument('--start_epoch', type=int, default=0, help='from which epoch to start.')
    p.add_argument('--num_samples', type=int, default=8)
    p.add_argument('--save', type=str, default='./weights', help='directory to save the model.')
    p.add_argument('--overfit', action='store_true')
    # train_kpcn.py:376-425
    p.add_argument('--desc', type=str, required=True, help='short description of the current experiment.')
    p.add_argument('--lr_dncnn', type=float, default=1e-4, help='learning rate of KPCN.')
    p.add_argument('--lr_pnet', type=float, nargs='+', default=[0.0001], help='learning rate of PathNet.')
    p.add_argument('--lr_ckpt', action='store_true', help='keep the learning rate stored in the checkpoint.')
    p.add_argument('--best_err', type=float, required=False)
    p.add_argument('--pnet_out_size', type=int, nargs='+', default=[3], help='# of channels of outputs of PathNet.')
    p.add_argument('--manif_loss', type=str, required=False, help='`FMSE` or `GRS`')
    p.add_argument('--train_branches', action='store_true', help='train the diffuse and specular branches independently.')
    p.add_argument('--use_llpm_buf', action='store_true', help='use the llpm-specific buffer.')
    p.add_argument('--manif_learn', action='store_true', help='use the manifold learning loss.')
    p.add_argument('--w_manif', type=float, nargs='+', default=[0.1],
                   help='ratio of the manifold learning loss to the reconstruction loss.')
    p.add_argument('--disentangle', type=str, default='m11r11', help='`m11r11`, `m10r01`, `m10r11`, or `m11r01`')
    p.add_argument('--single_gpu', action='store_true', help='accepted for compatibility (one process drives one GPU)')
    p.add_argument('--device_id', type=int, default=0, help='device id (single process)')
    p.add_argument('--kpcn_ref', action='store_true', help='train KPCN-Ref model.')
    p.add_argument('--kpcn_pre', action='store_true', help='train KPCN-Pre model.')
    p.add_argument('--not_save', action='store_true', help='do not save checkpoint (debugging purpose).')
    p.add_argument('--local', action='store_true')
    # this build
    p.add_argument('--synthetic', type=int, default=16, help='synthetic batches per epoch (the dataset reader is out of scope)')
    p.add_argument('--patch_size', type=int, default=128)
    p.add_argument('--graph', action='store_true', help='one hipGraph replay per training step')
    p.add_argument('--defer_check', dest='defer_check', action='store_true', default=True,
                   help="with --graph (the default there): check a step's losses for non-finite values after the NEXT step has "
                        "been enqueued (the device guard still skips the update at once; the error is raised one step later) -- "
                        "the host prepares the next batch while the GPU runs")
    p.add_argument('--sync_check', dest='defer_check', action='store_false',
                   help="with --graph: read the non-finite flags of every step before the next one is enqueued (one host sync per step)")
    p.add_argument('--one_graph', action='store_true',
                   help="with --graph: the step as ONE forked hipGraph instead of two half-step graphs on two streams + a tail graph")
    p.add_argument('--overlap_allreduce', action='store_true',
                   help="with --graph on several ranks and --use_llpm_buf: cut the backward at the P-buffers and put the dncnn "
                        "gradient bucket on the wire while the PathNets' backward runs (three graphs; bit-identical; not measured "
                        "on a multi-GPU node yet, hence off by default)")
    p.add_argument('--pairing_rng', choices=('cpu', 'device'), default='cpu',
                   help="FeatureMSE pairings: the reference's CPU randperm stream, or a keyed permutation on the GPU")
    p.add_argument('--pathnet_weight_norm', dest='pathnet_weight_norm', action='store_true', default=True,
                   help="weight-normalised PathNet layers (w = g * v / ||v||: upstream sbmc's ConvChain default, which "
                        "support/networks.py:18-24 does not switch off).  The default; a restored checkpoint's layout wins")
    p.add_argument('--no_pathnet_weight_norm', dest='pathnet_weight_norm', action='store_false',
                   help="plain nn.Conv2d weights in the PathNets (the parametrisation of this build's rounds 1-4)")
    p.add_argument('--pairing', choices=('local', 'global'), default='local',
                   help="FeatureMSE intra-batch pairing under several ranks: inside a rank's patches (default), or over the "
                        "all-gathered GLOBAL batch as nn.DataParallel's gathered loss does (train_kpcn.py:266-269; not with --graph)")
    return p


def check_args(args):
    """The argument errors of ``train_kpcn.py:427-441``."""
    if args.manif_learn and not args.use_llpm_buf:
        raise RuntimeError('The manifold learning module requires a llpm-specific buffer.')
    if args.manif_learn and not args.manif_loss:
        raise RuntimeError('The manifold learning module requires a manifold loss.')
    if not args.manif_learn and args.manif_loss:
        raise RuntimeError('A manifold loss is not necessary when the manifold learning module is opted out.')
    if args.manif_learn and args.manif_loss not in ['GRS', 'FMSE']:
        raise RuntimeError('Argument `manif_loss` should be either `FMSE` or `GRS`')
    if args.disentangle not in ['m11r11', 'm10r01', 'm10r11', 'm11r01']:
        raise RuntimeError('Argument `disentangle` should be either `m11r11`, `m10r01`, `m10r11`, or `m11r01`')
    for s in args.pnet_out_size:
        if args.disentangle != 'm11r11' and s % 2 != 0:
            raise RuntimeError('Argument `pnet_out_size` should be a list of even numbers')
    if getattr(args, 'pairing', 'local') == 'global' and getattr(args, 'graph', False):
        raise RuntimeError('`--pairing global` all-gathers inside the loss: not capturable, run it without `--graph`')
    return args


def main(argv=None):
    args = check_args(build_parser().parse_args(argv))
    rank, world, local = wd.init('nccl')
    device = torch.device('cuda', local if world > 1 else args.device_id)
    torch.cuda.set_device(device)
    torch.manual_seed(0)                                               # train_kpcn.py:346-348
    group = torch.distributed.group.WORLD if world > 1 else None
    sizes, dataloaders = init_data(args, device, rank, world)
    interfaces, params = init_model(sizes, args, device, group)
    params['rank'] = rank
    train(interfaces, dataloaders, params, args)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
