"""World-size-2 gloo test of the N>1 path: pixel sharding, the chi^2 all-reduce and the map gather.
Per-rank compute is done by the oracle (CPU); on the GPU box the same host code drives libdangx."""
import os
import socket

import numpy as np
import torch
import torch.distributed as td
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_ffi as O
    from dang_amd import dist, synth, stream_id
    dpar, ddata, bands, comps, meta = synth.make_sky("C2", nside=4, rank=rank, nranks=world, start="truth")
    orc = O.Oracle(bands, comps, ddata, pix0=meta["pix0"])
    orc.amp_sample_direct(1, 1, "sample", dpar.seed, stream_id(1, 0, 1, 0, 1), "reference")
    orc.sample_index_mh(1, 0, 1, 5, "sample", dpar.seed, stream_id(2, 1, 1, 0, 1))
    local, _ = orc.chisq(1, 3, 1.0)                     # nump=1: un-normalised local sum / nbands
    total = dist.allreduce_sum_float(local) / ddata.nump
    amp = dist.gather_maps(torch.from_numpy(orc.amplitude(1)), meta["npix_global"], dst=0)
    idx = dist.gather_maps(torch.from_numpy(orc.indices(1)), meta["npix_global"], dst=0)
    if rank == 0:
        np.savez(out, chisq=total, amp=amp.numpy(), idx=idx.numpy())
    td.barrier()
    td.destroy_process_group()


def test_two_rank_sharded_path_matches_single_rank(tmp_path):
    import oracle_ffi as O
    from dang_amd import synth, stream_id
    out = str(tmp_path / "r0.npz")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    dpar, ddata, bands, comps, meta = synth.make_sky("C2", nside=4, start="truth")
    orc = O.Oracle(bands, comps, ddata)
    orc.amp_sample_direct(1, 1, "sample", dpar.seed, stream_id(1, 0, 1, 0, 1), "reference")
    orc.sample_index_mh(1, 0, 1, 5, "sample", dpar.seed, stream_id(2, 1, 1, 0, 1))
    chisq, _ = orc.chisq(1, 3, ddata.nump)
    assert np.array_equal(got["amp"], orc.amplitude(1))     # RNG keyed by global pixel: bitwise
    assert np.array_equal(got["idx"], orc.indices(1))
    assert abs(float(got["chisq"]) - chisq) <= 1e-12 * chisq


def _ragged_worker(rank, world, port, out):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("gloo", rank=rank, world_size=world)
    from dang_amd import dist
    npix = 12 * 4 * 4 + 1 - 1   # 192 pixels over 5 ranks: shards of 39, 39, 38, 38, 38
    p0, n = dist.shard_range(npix, rank, world)
    local = torch.arange(p0, p0 + n, dtype=torch.float64).repeat(2, 3, 1) + torch.tensor([0.0, 1000.0]).view(2, 1, 1)
    full = dist.gather_maps(local, npix, dst=0)
    if rank == 0:
        np.save(out, full.numpy())
    td.barrier()
    td.destroy_process_group()


def test_gather_maps_with_a_rank_count_that_does_not_divide_the_sky(tmp_path):
    """shard_range gives shards that differ by one pixel when nranks does not divide npix; gather_maps pads and trims."""
    out = str(tmp_path / "full.npy")
    mp.spawn(_ragged_worker, args=(5, _free_port(), out), nprocs=5, join=True)
    full = np.load(out)
    want = np.arange(192, dtype=np.float64)[None, None, :] + np.array([0.0, 1000.0])[:, None, None] + np.zeros((2, 3, 1))
    assert full.shape == (2, 3, 192) and np.array_equal(full, want)


def test_balanced_shard_boundaries():
    """dist.balanced_bounds_run / balanced_bounds_mask: contiguous ranges that cover the sky once, with equal unmasked pixel
    counts to within a fraction of a percent, and the two forms agree on a one-run mask."""
    import numpy as np
    from dang_amd import dist
    npix = 12 * 64 * 64
    m0, m1 = int(0.45 * npix) + 1, int(0.55 * npix) + 1
    mask = np.ones(npix)
    mask[m0:m1] = 0.0
    for n in (2, 3, 8):
        b = dist.balanced_bounds_run(npix, n, m0, m1)
        assert b[0] == 0 and b[-1] == npix and all(x <= y for x, y in zip(b, b[1:])) and len(b) == n + 1
        un = [mask[b[r]:b[r + 1]].sum() for r in range(n)]
        assert max(un) - min(un) <= 0.01 * np.mean(un) + (m1 - m0) / 32 + 2, (n, un)
        b2 = dist.balanced_bounds_mask(mask, n)
        assert all(abs(x - y) <= 1 for x, y in zip(b, b2)), (b, b2)
        assert [dist.shard_range(npix, r, n, b) for r in range(n)] == [(b[r], b[r + 1] - b[r]) for r in range(n)]
    # equal ranges when nothing is masked
    assert dist.balanced_bounds_run(1000, 4, 0, 0) == [0, 250, 500, 750, 1000]


def _balanced_worker(rank, world, port, out):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    td.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_ffi as O
    from dang_amd import dist, synth, stream_id
    # a mask off the centre of the RING order: the work-balanced shards are then all of different size
    dpar, ddata, bands, comps, meta = synth.make_sky("C2", nside=8, rank=rank, nranks=world, start="truth", balance=True,
                                                     mask_frac=(0.15, 0.40))
    assert meta["bounds"] is not None and meta["npix"] == meta["bounds"][rank + 1] - meta["bounds"][rank]
    orc = O.Oracle(bands, comps, ddata, pix0=meta["pix0"])
    orc.amp_sample_direct(1, 1, "sample", dpar.seed, stream_id(1, 0, 1, 0, 1), "reference")
    orc.sample_index_mh(1, 0, 1, 5, "sample", dpar.seed, stream_id(2, 1, 1, 0, 1))
    amp = dist.gather_maps(torch.from_numpy(orc.amplitude(1)), meta["npix_global"], dst=0, bounds=meta["bounds"])
    idx = dist.gather_maps(torch.from_numpy(orc.indices(1)), meta["npix_global"], dst=0, bounds=meta["bounds"])
    err = None
    b = meta["bounds"]
    if all(b[r + 1] - b[r] != dist.shard_range(meta["npix_global"], r, world)[1] for r in range(world)):
        # the equal-range sizes fit NO shard here (so no rank enters the collective): refused, not trimmed
        try:
            dist.gather_maps(torch.from_numpy(orc.amplitude(1)), meta["npix_global"], dst=0)
        except ValueError as e:
            err = str(e)
    sizes = [None] * world
    td.all_gather_object(sizes, (meta["npix"], err is not None))
    if rank == 0:
        np.savez(out, amp=amp.numpy(), idx=idx.numpy(), sizes=np.array([s[0] for s in sizes]), refused=np.array([s[1] for s in sizes]))
    td.barrier()
    td.destroy_process_group()


def test_gather_maps_with_work_balanced_shard_boundaries(tmp_path):
    """north_star's "gather for map output" with the boundaries bench.py and the Fortran MPI path shard by (equal unmasked pixel
    count, dist.balanced_bounds_run): 5 ranks, shards of five different sizes; the gathered maps are the one-rank maps bit for
    bit, and gather_maps refuses shards that do not fit the boundaries it was given."""
    import oracle_ffi as O
    from dang_amd import synth, stream_id
    out = str(tmp_path / "bal.npz")
    mp.spawn(_balanced_worker, args=(5, _free_port(), out), nprocs=5, join=True)
    got = np.load(out)
    assert len(set(got["sizes"].tolist())) >= 3 and got["sizes"].sum() == 12 * 8 * 8, got["sizes"]
    assert got["refused"].all()
    dpar, ddata, bands, comps, meta = synth.make_sky("C2", nside=8, start="truth", mask_frac=(0.15, 0.40))
    orc = O.Oracle(bands, comps, ddata)
    orc.amp_sample_direct(1, 1, "sample", dpar.seed, stream_id(1, 0, 1, 0, 1), "reference")
    orc.sample_index_mh(1, 0, 1, 5, "sample", dpar.seed, stream_id(2, 1, 1, 0, 1))
    assert np.array_equal(got["amp"], orc.amplitude(1))
    assert np.array_equal(got["idx"], orc.indices(1))
