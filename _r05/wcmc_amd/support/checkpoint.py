"""The checkpoint format of ``train_kpcn.py``, so that runs can move between the reference
scripts and this library in either direction.

Written at ``train_kpcn.py:106-124`` (``latest_<name>.pth`` every epoch) and ``:134-152`` (best model): one
``torch.save``d dict

    description, start_epoch, model (``str(models['dncnn'])``), params, optims (the optimiser OBJECTS, by name
    ``optim_<model>``), args, best_err, and ``state_dict_<model>`` for every model of the interface.

Read back at ``train_kpcn.py:240-296``: weights by ``state_dict_<model>`` (a ``module.`` prefix left by
``nn.DataParallel`` is stripped on a key mismatch), optimiser state from ``ck['optims']`` or, in older files, from
``ck['params']``; the learning rate of the command line replaces the stored one unless ``--lr_ckpt``.

The file is a pickle of arbitrary objects (``argparse.Namespace``, ``torch.optim.Adam``): loading needs
``torch.load(..., weights_only=False)`` and therefore a trusted file, exactly as with the reference.
"""
import os
from collections import OrderedDict

import torch


def make_checkpoint(itf, epoch, args, params=None):
    """The dict of ``train_kpcn.py:110-121`` for interface ``itf`` after epoch ``epoch`` (0-based)."""
    tmp_params = dict(params) if params is not None else {}
    tmp_params['vis'] = None                                  # the visdom handle does not pickle (train_kpcn.py:108)
    state = {
        'description': getattr(args, 'desc', None),
        'start_epoch': epoch + 1,
        'model': str(itf.models['dncnn']),
        'params': tmp_params,
        'optims': itf.optims,
        'args': args,
        'best_err': itf.best_err,
        # not a reference key (the reference's loader ignores it): the conv arithmetic the run trained in (wcmc_amd.ops.MODES) --
        # the modes differ in the rounding of the backward GEMMs, and a resume under another one is a (legitimate) change of the
        # optimisation's noise that should not happen unnoticed
        'wcmc_precision': _precision(),
    }
    for name, model in itf.models.items():
        state['state_dict_' + name] = model.state_dict()
    return state


def _precision():
    from .. import ops
    return ops.PRECISION


def precision_note(ck, log=print):
    """Say so when a checkpoint is resumed under another conv arithmetic than it was written in; returns the stored
    mode (None for files of the reference or of rounds 1-3)."""
    stored = ck.get('wcmc_precision')
    now = _precision()
    if stored is not None and stored != now:
        log("Note: the checkpoint was trained with conv arithmetic '%s'; this run uses '%s' (WCMC_PRECISION / ops.set_precision): "
            "same model and optimiser state, another rounding of the GEMMs." % (stored, now))
    return stored


def save_checkpoint(path, itf, epoch, args, params=None):
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    torch.save(make_checkpoint(itf, epoch, args, params), path)


def load_checkpoint(path, map_location='cpu'):
    return torch.load(path, map_location=map_location, weights_only=False)


def restore_models(ck, models):
    """``train_kpcn.py:241-250``."""
    for name, model in models.items():
        sd = ck['state_dict_' + name]
        try:
            model.load_state_dict(sd)
        except RuntimeError:                                  # saved from inside nn.DataParallel: 'module.<key>'
            model.load_state_dict(OrderedDict((k[7:], v) for k, v in sd.items()))
    return ck.get('start_epoch', 0), ck.get('best_err', 1e10)


def restore_optims(ck, optims, lrs, lr_ckpt=False, log=print):
    """``train_kpcn.py:279-296``.  optims: ``{'optim_<model>': Adam}`` freshly built over the restored models;
    lrs: ``{'optim_<model>': lr}`` from the command line, which win unless ``lr_ckpt``."""
    for key, optim in optims.items():
        name = key[len('optim_'):]
        if 'optims' in ck:
            state = ck['optims'][key].state_dict()
        elif key in ck.get('params', {}):
            state = ck['params'][key].state_dict()
        else:
            log('No state for the optimizer for %s, use the initial optimizer and learning rate.' % (name))
            continue
        if not lr_ckpt:
            log('Set the new learning rate %.3e for %s.' % (lrs[key], name))
            state['param_groups'][0]['lr'] = lrs[key]
        else:
            log('Use the checkpoint learning rate for %s.' % (name))
        optim.load_state_dict(state)
