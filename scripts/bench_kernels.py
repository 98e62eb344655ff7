"""Micro-bench of the dominant kernels at the benchmark shapes (for rocprofv3 --pmc passes).
   python3 scripts/bench_kernels.py [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wcmc_amd import ops as o
it = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = "cuda"
torch.manual_seed(0)
n, c, h = 8, 100, 116
x = o.to_nhwc_raw(torch.randn(n, c, h, h, device=dev))
w = torch.randn(100, 100, 5, 5, device=dev) * 0.02
b = torch.zeros(100, device=dev)
dy = o.to_nhwc_raw(torch.randn(n, 100, h - 4, h - 4, device=dev))
xs, dys = o.split_raw(x), o.split_raw(dy)
wp, wpt, wpt2 = o._pack_x(w, 0), o._pack_x(w, 1), o._pack_x(w, 2)
h3 = 100                                   # 96x96 outputs: the 12x16-tile instance of the 5x5 forward (every other KPCN height runs the
                                           # 16-row instance since round 6: 116 -> 112 here, and the 116x116 outputs of the data gradients below = 5 x 16 + 3 x 12)
x3s = o.split_raw(o.to_nhwc_raw(torch.randn(n, c, h3, h3, device=dev)))
x1s = o.split_raw(o.to_nhwc_raw(torch.randn(64, 64, 128, 128, device=dev)))
w1p, b1 = o._pack_x(torch.randn(64, 64, 1, 1, device=dev) * 0.1, 0), torch.zeros(64, device=dev)
logits = o.to_nhwc_raw(torch.randn(8, 441, 92, 92, device=dev))
data = torch.rand(8, 3, 92, 92, device=dev)
g = torch.randn(8, 3, 92, 92, device=dev)
# the fused PathNet chains at the benchmark shape (B*S = 64 images of 128x128)
xe = o.presplit_shared(torch.randn(64, 36, 128, 128, device=dev))
pe = [torch.randn(64, 36, 1, 1, device=dev) * 0.3, torch.zeros(64, device=dev), torch.randn(64, 64, 1, 1, device=dev) * 0.2, torch.zeros(64, device=dev),
      torch.randn(64, 64, 1, 1, device=dev) * 0.2, torch.zeros(64, device=dev)]
pf = [torch.randn(128, 128, 1, 1, device=dev) * 0.15, torch.zeros(128, device=dev), torch.randn(3, 128, 1, 1, device=dev) * 0.15, torch.zeros(3, device=dev)]
# the KPCN output layer (100 -> 441 logits, 96^2 -> 92^2): three terms (default), one bf16 term ("bf16x321o"), one fp16 term ("bf16x321h")
xo = o.split_raw(o.to_nhwc_raw(torch.relu(torch.randn(n, 100, 96, 96, device=dev))))
wo = torch.randn(441, 100, 5, 5, device=dev) * 0.02
bo = torch.zeros(441, device=dev)
wo0, wo3, wo4 = o._pack_x(wo, 0), o._pack_x(wo, 3), o._pack_x(wo, 4)
for t in pe + pf: t.requires_grad_(True)
gy = o.to_nhwc_raw(torch.randn(64, 64, 128, 128, device=dev)); gm = o.to_nhwc_raw(torch.randn(8, 64, 128, 128, device=dev))
prop = o.to_nhwc_raw(torch.randn(8, 64, 128, 128, device=dev)).requires_grad_(True)
gout = o.to_nhwc_raw(torch.randn(64, 3, 128, 128, device=dev))
# the U-Net's 3x3 layers (conv_halo3_bf16x3_kernel): 64 -> 64 at 128^2, forward (three terms) and data gradient (two)
xu = o.split_raw(o.to_nhwc_raw(torch.relu(torch.randn(8, 64, 128, 128, device=dev))))
wu = torch.randn(64, 64, 3, 3, device=dev) * 0.05
bu = torch.zeros(64, device=dev)
wu0, wu2 = o._pack_x(wu, 0), o._pack_x(wu, 2)
dyu = o.split_raw(o.to_nhwc_raw(torch.randn(8, 64, 128, 128, device=dev)))
mu = (torch.rand(8 * 128 * 128 * 8, device=dev) * 255).to(torch.uint8)
def run():
    o.conv2d_x_raw(xu, (8, 64, 128, 128), wu0, bu, 64, 3, 1, "relu", out_split=True, mask_out=True)                     # conv_halo3<2, 2, 2>
    o.conv2d_x_raw(dyu, (8, 64, 128, 128), wu2, None, 64, 3, 1, "linear", out_split=True, gate_mask=mu, gate_act="relu",
                   colsum=True, terms=2)                                                                                 # conv_halo3<1, 2, 2>
    ye, me = o.conv_chain_spp_mean(xe, 8, 1, 0, ["relu", "relu", "linear"], pe)              # embed3_fwd / embed3_bwd
    torch.autograd.backward([ye, me], [gy, gm])
    yl = ye.detach().requires_grad_(True)
    o.cat_broadcast_chain(yl, prop, 8, 1, 0, ["relu", "relu"], pf).backward(gout)             # final2_kernel<false> / <true>
    y = o.conv2d_x_raw(xs, (n, c, h, h), wp, b, 100, 5, 0, "relu", out_split=True)          # bf16x3 fwd
    o.conv2d_x_raw(dys, (n, 100, h - 4, h - 4), wpt, None, 100, 5, 4, "linear", out_split=True, gate=xs, gate_act="relu")   # three-term dgrad (bf16x3 mode)
    o.conv2d_x_raw(dys, (n, 100, h - 4, h - 4), wpt2, None, 100, 5, 4, "linear", out_split=True, gate=xs, gate_act="relu", terms=2)   # two-term dgrad, 16x16 tiles
    o.conv2d_x_raw(x3s, (n, c, h3, h3), wp, b, 100, 5, 0, "relu", out_split=True)
    # (the three-term output layer shares its kernel name with the hidden 12x16-tile layers above: it stays out of this micro-bench so
    # that the per-kernel counter means keep describing ONE shape)
    o.conv2d_x_raw(xo, (n, 100, 96, 96), wo3, bo, 441, 5, 0, "linear", out_split=False, terms=1)        # ... one bf16 term
    o.conv2d_out_f16_raw(xo, (n, 100, 96, 96), wo4, bo, 441, 5, 0)                                      # ... one fp16 term
    o.conv2d_wgrad_x_raw(xs, (n, c, h, h), dys, 100, 5, 0, (100, 100, 5, 5), terms=1)     # one-term weight gradient (default mode)
    o.conv2d_wgrad_x_raw(xs, (n, c, h, h), dys, 100, 5, 0, (100, 100, 5, 5), terms=3)
    o.conv2d_x_raw(x1s, (64, 64, 128, 128), w1p, b1, 64, 1, 0, "relu", out_split=True)       # PathNet 1x1 layer (HBM-bound)
    ld = logits.clone().requires_grad_(True)
    out = o.kernel_apply(data, ld)
    out.backward(g)
for _ in range(it):
    run()
torch.cuda.synchronize()
print("done")
