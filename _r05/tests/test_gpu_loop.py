"""Loop-level GPU tests: the epoch loop with checkpoint / resume through the launcher's own
functions, and tiled full-image inference through the HIP ``validate_batch``."""
import os
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import step as ostep                             # noqa: E402
from oracle.models import KPCN as OKPCN                      # noqa: E402
from oracle.networks import PathNet as OPathNet              # noqa: E402

DEV = "cuda"


def _args(tmp, **kw):
    from wcmc_amd import train_kpcn as tk
    argv = ["--desc", "loop test", "--model_name", "m", "--save", str(tmp), "--batch_size", "2", "--patch_size", "48",
            "--synthetic", "3", "--num_epoch", "2", "--use_llpm_buf", "--manif_learn", "--manif_loss", "FMSE", "--train_branches",
            "--lr_dncnn", "1e-3", "--lr_pnet", "1e-3"]
    for k, v in kw.items():
        argv += ["--" + k] + ([] if v is True else [str(v)])
    return tk.check_args(tk.build_parser().parse_args(argv))


def _params_vector(itf):
    return torch.cat([p.detach().reshape(-1) for m in itf.models.values() for p in m.parameters()]).cpu()


@pytest.mark.parametrize("graph", [False, True, "defer"])
def test_epoch_loop_checkpoint_resume_gives_the_same_next_epoch(tmp_path, graph, monkeypatch):
    """train_kpcn.py:87-161 on the MI355X path: two epochs in one go == one epoch, ``latest_m.pth`` written by the loop,
    a NEW process state restored from it by ``init_model`` (train_kpcn.py:240-296: weights, Adam moments and step counts,
    learning rates), then the second epoch -- parameters bit for bit, epoch summaries equal.  Small KPCN (the launcher's
    ``KPCN(n_in)`` is swapped for a 3-layer one to keep the test short); without ``--graph``, with it, and with ``--defer_check``."""
    from wcmc_amd import KPCN, train_kpcn as tk
    monkeypatch.setattr(tk, "KPCN", lambda n_in: KPCN(n_in, ksize=21, depth=3, width=24))
    monkeypatch.setattr(tk, "PathNet", lambda ic, outc, weight_norm=True: __import__("wcmc_amd.support.networks", fromlist=["PathNet"]).PathNet(ic, intermc=16, outc=outc, weight_norm=weight_norm))
    dev = torch.device(DEV, 0)
    extra = {"graph": True} if graph else {}
    if graph == "defer":
        extra["defer_check"] = True                           # (--defer_check: the epoch loop flushes the last step's check)

    def run(save_dir, start_epoch, num_epoch):
        args = _args(save_dir, **extra)
        args.start_epoch, args.num_epoch, args.val_epoch = start_epoch, num_epoch, 1
        if start_epoch:
            args.model_name = "latest_m"                      # resume from the per-epoch file, like `--model_name latest_<name>`
        torch.manual_seed(0)
        sizes, loaders = tk.init_data(args, dev)
        itfs, params = tk.init_model(sizes, args, dev)
        torch.manual_seed(123 + start_epoch)                  # FeatureMSE pairings of the epochs run here
        tk.train(itfs, loaders, params, args)
        return itfs[0]

    a = tmp_path / "a"
    b = tmp_path / "b"
    os.makedirs(a), os.makedirs(b)
    # reference run: epoch 0, reseed, epoch 1 -- in one process state
    args = _args(a, **extra)
    args.num_epoch, args.val_epoch = 1, 1
    torch.manual_seed(0)
    sizes, loaders = tk.init_data(args, dev)
    itfs, params = tk.init_model(sizes, args, dev)
    torch.manual_seed(123)
    tk.train(itfs, loaders, params, args)
    p_epoch0 = _params_vector(itfs[0])
    args.start_epoch, args.num_epoch = 1, 2
    params.get("graphed_steps", {}).clear()                  # (capture again, as the resumed process will: its warm-up draws pairings)
    torch.manual_seed(124)
    tk.train(itfs, loaders, params, args)
    p_cont = _params_vector(itfs[0])
    assert not torch.equal(p_epoch0, p_cont)
    ck = torch.load(str(a / "latest_m.pth"), weights_only=False)
    assert ck["start_epoch"] == 2 and set(k for k in ck if k.startswith("state_dict_")) == {
        "state_dict_dncnn", "state_dict_backbone_diffuse", "state_dict_backbone_specular"}
    # resumed run: epoch 0 in one process state, epoch 1 in a fresh one restored from latest_m.pth
    first = run(b, 0, 1)
    assert torch.equal(_params_vector(first), p_epoch0)
    del first
    resumed = run(b, 1, 2)
    st = resumed.optims["optim_dncnn"].state[next(resumed.models["dncnn"].parameters())]
    assert float(st["step"]) == 6.0                           # 3 batches x 2 epochs: the Adam step count came through the file
    assert torch.equal(_params_vector(resumed), p_cont), "resume from the checkpoint changed the trajectory"


def test_tiled_inference_through_hip_validate_batch_matches_oracle_tiles():
    """SURVEY.md 8f rank 4 (test_models.py:49-101,217-232; tiling datasets.py:1276-1299): a 256x256 synthetic frame cut into
    128x128 tiles with 32 px overlap, every tile denoised by the HIP ``validate_batch`` (KPCN-Manifold: PathNets + input
    assembly + KPCN + kernel apply), stitched by ``support.inference.inference``; the stitched interior must equal the CPU
    oracle run on the same tiles to 1e-3, and the P-buffers of the validation split too."""
    from wcmc_amd import KPCN
    from wcmc_amd.support import inference as inf
    from wcmc_amd.support.interfaces import KPCNInterface
    from wcmc_amd.support.losses import FeatureMSE, RelativeMSE
    from wcmc_amd.support.networks import PathNet
    from wcmc_amd.synthetic import make_batch
    H = W = 256
    P, PAD, S = 128, 32, 2
    torch.manual_seed(3)
    kw = dict(ksize=21, depth=9, width=16)                     # the real geometry (128 -> 92 -> 72 px valid core), narrow layers
    # m11r01 with 4 P-buffer channels: the denoiser sees the low half (2 channels): 35 + 2 + 1 = 38 inputs
    omods = {"dncnn": OKPCN(38, **kw), "backbone_diffuse": OPathNet(36, intermc=16, outc=4),
             "backbone_specular": OPathNet(36, intermc=16, outc=4)}
    hmods = {"dncnn": KPCN(38, **kw), "backbone_diffuse": PathNet(36, intermc=16, outc=4),
             "backbone_specular": PathNet(36, intermc=16, outc=4)}
    g = torch.Generator().manual_seed(4)
    for k in omods:
        with torch.no_grad():
            for n, p in omods[k].named_parameters():
                if n.endswith("bias"):
                    p.copy_(torch.rand(p.shape, generator=g) * 0.2 - 0.1)
        hmods[k].load_state_dict(omods[k].state_dict())
        hmods[k].to(DEV)
    frame = make_batch(1, S, H, seed=77, device="cpu")          # one full frame with the dataset's schema
    coords = inf.tile_coords(H, W, P, PAD)
    assert len(coords) == 9

    def tiles(bs):
        for k in range(0, len(coords), bs):
            cs = coords[k:k + bs]
            batch = {name: torch.cat([t[..., c[4]:c[4] + P, c[5]:c[5] + P] for c in cs], 0) for name, t in frame.items()}
            yield (batch, *[torch.tensor([c[q] for c in cs]) for q in range(6)])

    lf = {"l_diffuse": torch.nn.L1Loss(), "l_specular": torch.nn.L1Loss(), "l_recon": torch.nn.L1Loss(),
          "l_test": RelativeMSE(), "l_manif": FeatureMSE(non_local=True)}
    opt = {"optim_" + k: torch.optim.Adam(m.parameters(), lr=1e-4) for k, m in hmods.items()}
    itf = KPCNInterface(hmods, opt, lf, types.SimpleNamespace(model_name="t"), use_llpm_buf=True, manif_learn=True,
                        w_manif=0.1, train_branches=True, disentanglement_option="m11r01")

    def dev_tiles():
        for batch, *idx in tiles(4):
            yield ({k: v.to(DEV) for k, v in batch.items()}, *idx)

    with torch.no_grad():
        rad, pbuf = inf.inference(itf, dev_tiles(), H, W, P)
    assert rad.shape == (3, H, W) and pbuf["diffuse"].shape == (S, 2, H, W)
    with pytest.raises(RuntimeError, match="input channels"):      # a denoiser built for another split fails loudly
        KPCNInterface(hmods, opt, lf, types.SimpleNamespace(model_name="t"), use_llpm_buf=True, manif_learn=True,
                      disentanglement_option="m11r11").validate_batch({k: v[:, ..., :128, :128].to(DEV) for k, v in frame.items()})

    class OracleItf:                                            # the same stitching around the CPU oracle's validation forward
        def to_eval_mode(self):
            pass

        def validate_batch(self, batch):
            cfg = dict(use_llpm_buf=True, manif_learn=False, disentanglement_option="m11r01")
            out, p_regress, _ = ostep.forward_losses(omods, batch, cfg, None, train=False)
            return out["radiance"], p_regress

    with torch.no_grad():
        rad_o, pbuf_o = inf.inference(OracleItf(), tiles(3), H, W, P)
    core = (slice(None), slice(PAD, H - PAD), slice(PAD, W - PAD))
    err = ((rad.cpu()[core] - rad_o[core]).abs().max() / rad_o[core].abs().max()).item()
    assert err <= 1e-3, "stitched radiance: %.3e" % err
    errf = ((rad.cpu() - rad_o).abs().max() / rad_o.abs().max()).item()
    assert errf <= 1e-3, "stitched radiance incl. the replicate-padded ring: %.3e" % errf
    for br in ("diffuse", "specular"):
        e = ((pbuf[br].cpu() - pbuf_o[br]).abs().max() / pbuf_o[br].abs().max().clamp_min(1e-30)).item()
        assert e <= 1e-3, "stitched %s P-buffer: %.3e" % (br, e)
    val = itf.get_epoch_summary(mode="eval", norm=len(coords))
    assert np.isfinite(val) and val > 0


def test_launcher_main_runs_an_epoch_end_to_end(tmp_path, capsys):
    """``python -m wcmc_amd.train_kpcn`` with the README's KPCN-Manifold command line (full-width KPCN and PathNets, 64x64
    synthetic patches): init_data -> init_model -> train (one epoch, validation, both checkpoint files) in one call."""
    from wcmc_amd import train_kpcn as tk
    tk.main(["--single_gpu", "--batch_size", "2", "--val_epoch", "1", "--model_name", "KPCN_manifold_FMSE", "--desc",
             "KPCN manifold FMSE", "--num_epoch", "1", "--manif_loss", "FMSE", "--lr_dncnn", "1e-4", "--lr_pnet", "1e-4",
             "--use_llpm_buf", "--manif_learn", "--w_manif", "0.1", "--train_branches", "--save", str(tmp_path),
             "--synthetic", "2", "--patch_size", "64", "--graph", "--pairing_rng", "device"])
    out = capsys.readouterr().out
    assert "[] Training complete!" in out and "Model KPCN_manifold_FMSE.pth saved at epoch 0." in out
    assert "m_l_manif_diffuse" in out and "m_rmse" in out
    ck = torch.load(str(tmp_path / "KPCN_manifold_FMSE.pth"), weights_only=False)
    assert ck["start_epoch"] == 1 and 0 < ck["best_err"] < 1e9 and ck["args"].w_manif == [0.1]
    assert os.path.isfile(str(tmp_path / "latest_KPCN_manifold_FMSE.pth"))
    assert ck["model"].startswith("KPCN(")
