"""Driver entry points: ``build()`` compiles the HIP library for gfx950 (no GPU needed),
``smoke()`` runs one tiny KPCN-Manifold training step on ``cuda:0`` and checks it against the
CPU oracle."""
import os
import subprocess
import sys
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build():
    """hipcc --offload-arch=gfx950 -> wcmc_amd/libwcmc_hip.so, then import the package and bind the ABI.

    The oracle is Python (PyTorch CPU) -- nothing to compile; the reference is Python too and cannot
    travel, so there is no oracle/_ref (goldens under tests/golden/ pin the oracle instead)."""
    env = dict(os.environ)
    csrc = os.path.join(ROOT, "wcmc_amd", "csrc")
    if os.environ.get("WCMC_CLEAN_BUILD", "0") not in ("", "0"):
        # from the sources alone: the objects and the .so are git-ignored and travel prebuilt with a gpurun snapshot, so an
        # incremental make on such a box compiles nothing -- this switch proves the tree builds the binary (~2 min, 4 jobs)
        subprocess.run(["make", "-C", csrc, "clean"], check=True, env=env)
    subprocess.run(["make", "-C", csrc, "-j4"], check=True, env=env)
    # the debug build of the same library (kernel A/B switches, timing-only ablations): what the kernel-against-kernel tests load in a
    # child process (tests/test_gpu_ops.py::test_kernel_cross_checks_run_against_the_debug_build_in_a_subprocess)
    subprocess.run(["make", "-C", csrc, "-j4", "debug"], check=True, env=env)
    import wcmc_amd  # noqa: F401
    from wcmc_amd._lib import SIGNATURES, lib
    h = lib()
    for name in SIGNATURES:
        getattr(h, name)
    assert h.wcmc_abi_version() == 2


def smoke():
    """One small KPCN-Manifold step (2 PathNets + KPCN + FeatureMSE + clip + Adam) on cuda:0 vs the oracle."""
    import torch
    from oracle import step as ostep
    from oracle.models import KPCN as OKPCN
    from oracle.networks import PathNet as OPathNet
    from wcmc_amd import KPCN
    from wcmc_amd.optim import FusedClipAdam
    from wcmc_amd.support.interfaces import KPCNInterface
    from wcmc_amd.support.losses import FeatureMSE, RelativeMSE
    from wcmc_amd.support.networks import PathNet
    from wcmc_amd.synthetic import make_batch

    assert torch.cuda.is_available(), "smoke() needs the MI355X"
    dev = "cuda:0"
    torch.cuda.set_device(0)
    torch.manual_seed(0)
    B, S, H = 2, 4, 48
    kw = dict(ksize=21, depth=3, width=24)        # 48 -> 36, 21x21 apply
    omods = {"dncnn": OKPCN(39, **kw), "backbone_diffuse": OPathNet(36, intermc=16),
             "backbone_specular": OPathNet(36, intermc=16)}
    hmods = {"dncnn": KPCN(39, **kw), "backbone_diffuse": PathNet(36, intermc=16),
             "backbone_specular": PathNet(36, intermc=16)}
    for k in omods:
        hmods[k].load_state_dict(omods[k].state_dict())
        hmods[k].to(dev)
    start = {mn: {k: v.detach().clone() for k, v in m.named_parameters()} for mn, m in omods.items()}
    batch = make_batch(B, S, H, seed=1, device="cpu")
    cfg = dict(use_llpm_buf=True, manif_learn=True, train_branches=True, disentanglement_option="m11r11",
               w_manif=0.1)
    h_out = H - 12
    torch.manual_seed(2)
    perms = [ostep.draw_perms(B, S, h_out, h_out), ostep.draw_perms(B, S, h_out, h_out)]
    oopt = {"optim_" + k: torch.optim.Adam(m.parameters(), lr=1e-4) for k, m in omods.items()}
    loss_o, _ = ostep.train_step(omods, oopt, batch, cfg, perms)

    hopt = {"optim_" + k: torch.optim.Adam(m.parameters(), lr=1e-4) for k, m in hmods.items()}
    lf = {"l_diffuse": torch.nn.L1Loss(), "l_specular": torch.nn.L1Loss(), "l_recon": torch.nn.L1Loss(),
          "l_test": RelativeMSE(), "l_manif": FeatureMSE(non_local=True)}
    itf = KPCNInterface(hmods, hopt, lf, types.SimpleNamespace(model_name="smoke"), use_llpm_buf=True,
                        manif_learn=True, w_manif=0.1, train_branches=True)
    itf.fused_optim = FusedClipAdam(hmods, hopt)
    itf.iters = 1
    itf.to_train_mode()
    dbatch = {k: v.to(dev) for k, v in batch.items()}
    torch.manual_seed(2)
    itf.preprocess(dbatch)
    itf.train_batch(dbatch)
    torch.cuda.synchronize()
    for k, v in loss_o.items():
        got = itf.m_losses["m_" + k].item()
        assert abs(got - v.item()) <= 1e-3 * abs(v.item()) + 1e-7, (k, got, v.item())
    # The update itself: parameter DELTAS against the oracle's.  Adam's first step moves every entry by
    # lr * g / (|g| + eps) ~ lr * sign(g), so "no update" or "wrong sign" is a full lr (or two) away; entries whose
    # gradient is well conditioned (> 5 % of the tensor's rms) must agree to 1 % of lr -- all but the <= 1 % of them
    # that a ReLU unit landing on the other side of zero may turn in these narrow test networks.
    lr = 1e-4
    worst = [0.0]
    for mn in omods:
        for (k, p), (_, q) in zip(hmods[mn].named_parameters(), omods[mn].named_parameters()):
            d_h = p.detach().cpu() - start[mn][k]
            d_o = q.detach() - start[mn][k]
            well = q.grad.abs() > 0.05 * q.grad.pow(2).mean().sqrt()
            if not bool(well.any()):
                continue
            bad = float(((d_h - d_o)[well].abs() > 1e-2 * lr).float().mean())
            worst[0] = max(worst[0], bad)
            assert bad <= 0.02, (mn, k, "fraction of well-conditioned entries whose update differs", bad)
            # moved, and by as much as the oracle's entry did (lr * |g| / (|g| + eps): a full lr unless |g| ~ eps)
            assert bool((d_h[well].abs() >= 0.5 * d_o[well].abs()).all()) and float(d_o[well].abs().max()) > 0.5 * lr, \
                (mn, k, "parameters not updated")
    print("smoke ok:", {k: round(itf.m_losses["m_" + k].item(), 6) for k in loss_o},
          "| worst fraction of well-conditioned entries whose Adam update differs from the oracle's: %.4f" % worst[0])


if __name__ == "__main__":
    build()
    if "--smoke" in sys.argv:
        smoke()
