"""Pixel sharding across ranks (one process per GPU) and the global reductions.

The reference is single-process; its "collectives" are serial `sum()` intrinsics over
full-sky vectors (e.g. chi^2, src/dang_data_mod.f90:523-524).  With pixels sharded as
contiguous RING ranges these become scalar all-reduces (RCCL over xGMI when the
process group's backend is "nccl"; gloo on CPU in the tests).  Every BASELINE config
is per-pixel independent, so there is no other data-path collective.
"""
import os

import torch
import torch.distributed as td


def _single():
    """One rank: nothing to exchange -- unless DANGX_DIST_SINGLE=1 asks for the collectives anyway (a one-GPU rehearsal of the
    nccl = RCCL branches below: tests/test_gpu_multirank.py)."""
    return td.get_world_size() == 1 and os.environ.get("DANGX_DIST_SINGLE", "0") != "1"


def active():
    """a process group exists and its collectives are to be called"""
    return td.is_available() and td.is_initialized() and not _single()


def world():
    if td.is_available() and td.is_initialized():
        return td.get_rank(), td.get_world_size()
    return 0, 1


def shard_range(npix_global, rank, nranks, bounds=None):
    """Contiguous range [pix0, pix0+npix) of RING pixels owned by `rank`: equal ranges, or the ranges between the
    given boundaries (balanced_bounds_run / balanced_bounds_mask)."""
    if bounds is not None:
        return int(bounds[rank]), int(bounds[rank + 1] - bounds[rank])
    base, rem = divmod(int(npix_global), int(nranks))
    pix0 = rank * base + min(rank, rem)
    return pix0, base + (1 if rank < rem else 0)


# a masked pixel costs a kernel almost nothing (its lanes leave at once); 32 : 1 is close enough to balance by
_W_UNMASKED, _W_MASKED = 32, 1


def balanced_bounds_run(npix_global, nranks, m0, m1):
    """Shard boundaries b[0..nranks] of CONTIGUOUS RING ranges with equal WORK instead of equal pixel counts, for a mask that
    is one run of masked pixels [m0, m1) (a Galactic band in RING order): equal ranges leave the ranks that hold the masked
    band with little to do and make the others 1/(unmasked fraction) slower than the average -- 11 % at 8 ranks with a 10 % mask.
    Results do not depend on the boundaries (random streams are keyed by the global pixel)."""
    n, m0, m1 = int(npix_global), int(m0), int(m1)

    def cum(i):   # weight of pixels [0, i)
        over = min(max(i - m0, 0), m1 - m0)
        return _W_UNMASKED * (i - over) + _W_MASKED * over

    total = cum(n)
    bounds = [0]
    for r in range(1, nranks):
        target = (total * r) // nranks
        lo, hi = bounds[-1], n
        while lo < hi:          # smallest i with cum(i) >= target
            mid = (lo + hi) // 2
            if cum(mid) >= target:
                hi = mid
            else:
                lo = mid + 1
        bounds.append(lo)
    bounds.append(n)
    return bounds


def balanced_bounds_mask(mask, nranks):
    """The same for any mask (1-D array over the global sky, 0 / missing value = masked)."""
    import numpy as np
    m = np.asarray(mask)
    w = np.where((m == 0.0) | (m == -1.6375e30), _W_MASKED, _W_UNMASKED).astype(np.int64)
    c = np.cumsum(w)
    total = int(c[-1])
    bounds = [0] + [int(np.searchsorted(c, (total * r) // nranks, side="left")) + 1 for r in range(1, nranks)] + [m.size]
    for r in range(1, len(bounds)):
        bounds[r] = max(bounds[r], bounds[r - 1])
    return bounds


def allreduce_sum_float(x, device=None):
    """Sum a Python float over all ranks (returns the same value on every rank)."""
    if not active():
        return float(x)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if td.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    td.all_reduce(t, op=td.ReduceOp.SUM)
    return float(t.item())


def allreduce_sum_inplace_host(a):
    """Sum a small host float64 array over all ranks in place: the callback behind dangx_set_allreduce (dot products
    of the device CG, global-amplitude rows of template groups).  nccl process groups reduce on the device (RCCL)."""
    if not active():
        return a
    if td.get_backend() == "nccl":
        t = torch.from_numpy(a).to(torch.device("cuda", torch.cuda.current_device()))
        td.all_reduce(t, op=td.ReduceOp.SUM)
        a[...] = t.cpu().numpy()
    else:
        t = torch.from_numpy(a)     # shares memory
        td.all_reduce(t, op=td.ReduceOp.SUM)
    return a


def bcast_from_rank0(values):
    """Broadcast a short list of floats from rank 0 (e.g. c%indices(0, k, :), which lives on the first shard)."""
    if not active():
        return list(values)
    device = torch.device("cuda", torch.cuda.current_device()) if td.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    td.broadcast(t, src=0)
    return [float(v) for v in t.tolist()]


def allreduce_sum_(t):
    """In-place sum of a tensor over all ranks (device tensors go through RCCL)."""
    if active():
        td.all_reduce(t, op=td.ReduceOp.SUM)
    return t


def gather_maps(local, npix_global, dst=0, bounds=None):
    """Gather pixel-sharded maps [..., npix_local] to `dst` as [..., npix_global] (map output; what the reference's write_maps
    consumes, src/dang_data_mod.f90:573-664).  `bounds`: the shard boundaries b[0..nranks] the sky was split by (shard_range's
    argument: balanced_bounds_run / balanced_bounds_mask, synth.make_sky's meta["bounds"]); None = equal ranges."""
    rank, n = world()
    if not active():
        return local
    if bounds is not None and (len(bounds) != n + 1 or int(bounds[0]) != 0 or int(bounds[-1]) != int(npix_global)):
        raise ValueError("gather_maps: bounds must be the %d shard boundaries of a sky of %d pixels" % (n + 1, npix_global))
    sizes = [shard_range(npix_global, r, n, bounds)[1] for r in range(n)]
    if local.shape[-1] != sizes[rank]:
        raise ValueError("gather_maps: rank %d holds %d pixels, its shard has %d (wrong bounds?)" % (rank, local.shape[-1], sizes[rank]))
    if local.is_cuda and td.get_backend() != "nccl":
        local = local.cpu()     # gloo gathers host tensors only (rehearsals of the N-rank path on one GPU)
    lead = local.shape[:-1]
    # gloo and nccl both want equal-size tensors: shards differ in size (by one pixel when the rank count does not divide the
    # sky, by the masked run with work-balanced boundaries), so every shard is padded to the largest and trimmed on dst
    big = max(sizes)
    send = local.contiguous()
    if send.shape[-1] < big:
        send = torch.cat([send, send.new_zeros(*lead, big - send.shape[-1])], dim=-1)
    if rank == dst:
        parts = [torch.empty(*lead, big, dtype=local.dtype, device=local.device) for _ in sizes]
    else:
        parts = None
    td.gather(send, parts, dst=dst)
    return torch.cat([p[..., :s] for p, s in zip(parts, sizes)], dim=-1) if rank == dst else None
