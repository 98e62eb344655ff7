"""GPU diagnostic: deep-chain gradient error vs fp64 across seeds, max-norm and L2, plus ReLU mask flips."""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wcmc_amd import ops as o
from wcmc_amd.modules import ConvChain
from oracle import modules as om
torch.set_num_threads(16)
def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max()).item(), ((a - b).norm() / b.norm()).item()
for seed in range(4):
    torch.manual_seed(seed)
    ref = om.ConvChain(34, 441, ksize=5, width=100, depth=9, pad=False, weight_norm=False).double()
    mod = ConvChain(34, 441, ksize=5, width=100, depth=9, pad=False, weight_norm=False)
    mod.load_state_dict({k: v.float() for k, v in ref.state_dict().items()}); mod.cuda()
    x = torch.rand(2, 34, 64, 64) - 0.3
    # flips: run the oracle layer by layer in fp64 and the HIP chain truncated to l layers
    flips = []
    h = x.double()
    for l in range(8):
        h = torch.relu(ref.layers[l](h))
        sub = ConvChain(34, 100, ksize=5, width=100, depth=l + 1, pad=False, output_type="relu", weight_norm=False)
        sub.layers = torch.nn.ModuleList(list(mod.layers[:l + 1])); sub.depth = l + 1
        with torch.no_grad():
            hh = sub(x.cuda())
        flips.append(int(((hh.cpu() > 0) != (h > 0)).sum()))
    xr = x.double(); yr = ref(xr); g = torch.rand(yr.shape, generator=torch.Generator().manual_seed(9)) - 0.5
    (yr * g.double()).sum().backward()
    y = mod(x.cuda()); (y * g.cuda()).sum().backward()
    e0 = rel(mod.layers[0].weight.grad, ref.layers[0].weight.grad)
    e4 = rel(mod.layers[4].weight.grad, ref.layers[4].weight.grad)
    e8 = rel(mod.layers[8].weight.grad, ref.layers[8].weight.grad)
    print("seed %d flips/layer %s  out max %.2e | dW0 max %.2e l2 %.2e | dW4 max %.2e l2 %.2e | dW8 max %.2e l2 %.2e"
          % (seed, flips, rel(y, yr)[0], *e0, *e4, *e8))
