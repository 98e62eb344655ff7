"""The fused PathNet chains (csrc/pathnet_fused.hip) at the benchmark shape, alone: forward and backward of the embedding
(B*S = 64 images of 128x128, 36 -> 64 -> 64 -> 64 + spp mean) and of the final chain (64 + 64 -> 128 -> 3), fused and
layer-by-layer, with the bytes each direction has to move.
   python3 scripts/time_pathnet_fused.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wcmc_amd import ops as o
dev = "cuda"
B, S, H = 8, 8, 128
torch.manual_seed(0)
def timeit(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
x = o.presplit_shared(torch.randn(B * S, 36, H, H, device=dev))
pe = [torch.randn(64, 36, 1, 1, device=dev) * 0.3, torch.zeros(64, device=dev), torch.randn(64, 64, 1, 1, device=dev) * 0.2, torch.zeros(64, device=dev),
      torch.randn(64, 64, 1, 1, device=dev) * 0.2, torch.zeros(64, device=dev)]
pf = [torch.randn(128, 128, 1, 1, device=dev) * 0.15, torch.zeros(128, device=dev), torch.randn(3, 128, 1, 1, device=dev) * 0.15, torch.zeros(3, device=dev)]
for t in pe + pf: t.requires_grad_(True)
gy = o.to_nhwc_raw(torch.randn(B * S, 64, H, H, device=dev)); gm = o.to_nhwc_raw(torch.randn(B, 64, H, H, device=dev))
prop = o.to_nhwc_raw(torch.randn(B, 64, H, H, device=dev)).requires_grad_(True)
gout = o.to_nhwc_raw(torch.randn(B * S, 3, H, H, device=dev))
M = B * S * H * H
for fused in (True, False):
    o.FUSE_EMBED = o.FUSE_FINAL = fused
    def emb_f():
        with torch.no_grad():
            return o.conv_chain_spp_mean(x, S, 1, 0, ["relu", "relu", "linear"], pe)
    def emb_fb():
        y, m = o.conv_chain_spp_mean(x, S, 1, 0, ["relu", "relu", "linear"], pe)
        torch.autograd.backward([y, m], [gy, gm])
    y, _ = emb_f()
    yl = y.detach().requires_grad_(True)
    def fin_f():
        with torch.no_grad():
            return o.cat_broadcast_chain(yl, prop, S, 1, 0, ["relu", "relu"], pf)
    def fin_fb():
        out = o.cat_broadcast_chain(yl, prop, S, 1, 0, ["relu", "relu"], pf)
        out.backward(gout)
    tf, tfb, uf, ufb = timeit(emb_f), timeit(emb_fb), timeit(fin_f), timeit(fin_fb)
    print("%-15s embedding: forward (+ spp mean) %6.1f us, backward %6.1f us | final: forward %6.1f us, backward %6.1f us" %
          ("fused" if fused else "layer by layer", tf, tfb - tf, uf, ufb - uf))
print("bytes (algorithmic): embedding fwd %.0f MB, bwd %.0f MB; final fwd %.0f MB, bwd %.0f MB" %
      (M * (160 + 256) / 1e6, M * (160 + 256 + 32) / 1e6, M * (256 + 32 + 16) / 1e6, M * (256 + 32 + 16 + 256 + 32) / 1e6))
