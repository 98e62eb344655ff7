"""One process per GPU; gradients are summed with RCCL (``torch.distributed`` backend "nccl" on ROCm)
over xGMI.  The reference's multi-GPU is single-process ``nn.DataParallel``
(``train_kpcn.py:256-271``): replicas see disjoint slices of the batch and the gradients of the
replicas are summed before the clip.  Here each rank holds full replicas of KPCN + 2 PathNets,
steps on its own shard of patches (no data-path collective), and the per-model flat gradient is
all-reduced once per step (46.8 MB in 3 messages) before the fused clip + Adam.
"""
import os

import torch
import torch.distributed as dist


def init(backend=None):
    """Initialise from the torchrun environment; returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def broadcast_parameters(models, src=0, group=None):
    """Make every rank start from rank ``src``'s weights (DataParallel replicates module 0)."""
    for name in sorted(models):
        for p in models[name].parameters():
            dist.broadcast(p.data, src, group=group)


def shard_seed(base_seed, rank):
    """Per-rank data seed: ranks draw disjoint synthetic patches (weak scaling, 8 patches per GPU)."""
    return base_seed + rank


def average_gradients(models, group=None):
    """Flat per-model all-reduce + mean, written back INTO the existing ``p.grad`` tensors (the un-fused path).

    In place on purpose: under ``wcmc_amd.graph.GraphedTrainStep`` ``p.grad`` are the buffers the captured
    backward writes on every replay; rebinding ``p.grad`` to views of a fresh flat buffer would leave every later
    step clipping and stepping on step 1's gradients."""
    world = dist.get_world_size(group)
    for name in sorted(models):
        params = [p for p in models[name].parameters() if p.grad is not None]
        if not params:
            continue
        flat = torch.cat([p.grad.reshape(-1) for p in params])
        dist.all_reduce(flat, group=group)
        flat.div_(world)
        off, views = 0, []
        for p in params:
            n = p.numel()
            views.append(flat[off:off + n].view(p.shape))
            off += n
        torch._foreach_copy_([p.grad for p in params], views)


def max_over_ranks(value, device):
    """The bench clock: MAX of a python float over ranks."""
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()


def gather_floats(value, device):
    """One python float per rank, on every rank (world 1: [value])."""
    if not dist.is_initialized():
        return [float(value)]
    t = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]


def time_allreduce(fused_optim, group, device, iters=10):
    """The step's gradient exchange alone: the per-model buckets of ``FusedClipAdam`` (46.8 MB in 3 messages) all-reduced
    back to back ``iters`` times, HIP events on the launch stream around them (a synchronous collective runs on the
    communicator's stream and the launch stream waits for it, so the events bracket the collectives).  Returns a dict for the
    bench line: milliseconds per step's worth of buckets, bytes, algorithmic bandwidth (MAX over ranks)."""
    bufs = [fl.g.clone() for fl in fused_optim.flats.values()]
    nbytes = sum(b.numel() * 4 for b in bufs)
    for b in bufs:                                         # warm-up (communicator set-up, channel allocation)
        dist.all_reduce(b, group=group)
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        for b in bufs:
            dist.all_reduce(b, group=group)
    e1.record()
    torch.cuda.synchronize(device)
    ms = max_over_ranks(e0.elapsed_time(e1) / iters, device)
    return {"ms_per_step": round(ms, 4), "bytes_per_rank": nbytes, "messages": len(bufs),
            "algbw_GBs": round(nbytes / (ms * 1e-3) / 1e9, 1), "ranks": dist.get_world_size(group),
            "backend": dist.get_backend(group),
            "note": "the three gradient buckets alone, back to back, outside the timed region; inside a step they are issued "
                    "asynchronously in backward order between the step's two graphs (eager steps: the clip + Adam of bucket i runs "
                    "while buckets i+1.. are on the wire)"}
