"""Edge cases of the path on the GPU vs the oracle: ragged pixel counts (not a multiple of the wavefront / block),
an all-masked sky, one band, the maximum number of bands, every CG-group size 1..8, nmaps = 1 with a T-only model,
unusual band counts (generic LDS-form kernels), non-SPD blocks."""
import numpy as np
import pytest

import dang_amd as da
from dang_amd import _lib as L
from dang_amd import synth
from dang_amd.api import BandInfo, DangCGGroup, DangComps, DangData, DangParams

import oracle_ffi as O
from util import MAPN, assert_amps_close, assert_indices_close, make_case, pair

pytestmark = pytest.mark.gpu


def _run_iteration(case, it=2):
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    for g in dpar.cg_groups:
        for f in g.pol_flag:
            s = da.stream_id(it, 0, g.cg_group, 0, f)
            _, bad = eng.amp_sample(g.cg_group, f, "sample", dpar.seed, s)
            obad = orc.amp_sample_direct(g.cg_group, f, "sample", dpar.seed, s, "reference")
            assert bad == obad
    for l, c in enumerate(comps):
        for j in range(c.nindices):
            if c.sample_index[j]:
                for f in c.pol_flag[j]:
                    s = da.stream_id(it, 1, l, j, f)
                    ag = eng.index_sample(l, j, MAPN[f], dpar.nsample, "sample", dpar.seed, s)
                    ao = orc.sample_index_mh(l, j, MAPN[f], dpar.nsample, "sample", dpar.seed, s)
                    assert ag == ao
    assert_amps_close(eng, orc, len(comps), 1e-9)
    assert_indices_close(eng, orc, comps, 1e-12)
    return eng, orc


def _custom(npix, nb, names, nmaps=1, seed=3):
    """A small hand-built T-only problem with an arbitrary pixel count (not a HEALPix size)."""
    rng = np.random.default_rng(seed)
    freqs = synth.band_freqs_ghz(nb)
    bands = [BandInfo("b%d" % j, freqs[j]) for j in range(nb)]
    comps = []
    for name in names:
        ctype, nu_ref, asig, idx = synth.PHYS[name]
        nind = len(idx)
        ind = np.stack([np.full((nmaps, npix), m) + (s * rng.standard_normal((nmaps, npix)) if s > 0 else 0.0)
                        for (_, m, s, _) in idx]) if nind else None
        comps.append(DangComps(label=name, type=ctype, nu_ref=nu_ref, cg_group=1, nindices=nind,
                               ind_label=[i[0] for i in idx], sample_index=[i[3] for i in idx], index_mode=[2] * nind,
                               lnl_type=["chisq"] * nind, prior_type=["gaussian" if i[3] else "uniform" for i in idx],
                               gauss_prior=[[i[1], i[2] if i[2] > 0 else 1.0] for i in idx],
                               uni_prior=[[i[1] - 10 * max(i[2], 1.0), i[1] + 10 * max(i[2], 1.0)] for i in idx],
                               step_size=[0.5 * i[2] for i in idx], pol_flag=[[L.FLAG_T]] * nind,
                               amplitude=asig * rng.standard_normal((nmaps, npix)), indices=ind))
    o = O.Oracle(bands, comps, DangData(sig_map=np.zeros((nb, nmaps, npix)), rms_map=np.ones((nb, nmaps, npix)), masks=np.ones((nmaps, npix))))
    sky, _ = o.sky_model()
    rms = (0.5 + rng.uniform(size=sky.shape))
    mask = np.ones((nmaps, npix))
    mask[0, ::7] = 0.0
    dd = DangData(sig_map=sky + rms * rng.standard_normal(sky.shape), rms_map=rms, masks=mask, pol_type=[1],
                  nump=float((mask[0] != 0).sum()))
    dpar = DangParams(cg_groups=[DangCGGroup(1, pol_flag=[L.FLAG_T])])
    return dpar, dd, bands, comps, dict(npix=npix, npix_global=npix, pix0=0, nbands=nb, nmaps=nmaps)


@pytest.mark.parametrize("npix", [1, 63, 65, 257, 1000])
def test_ragged_pixel_counts(built, npix):
    eng, orc = _run_iteration(_custom(npix, 3, ["synch", "dust"]))
    a, b = eng.sky_model_chisq(1, 1), orc.chisq(1, 1, 1.0)[0] * 3
    assert abs(a - b) <= 1e-10 * max(abs(b), 1e-300)


@pytest.mark.parametrize("nb", [1, 2, 4, 7, 9, 11, 32])
def test_band_counts_including_generic_and_maximum(built, nb):
    names = ["synch"] if nb < 3 else ["synch", "dust"]
    _run_iteration(_custom(300, nb, names))


@pytest.mark.parametrize("ng", [1, 2, 3, 4, 5, 6, 7, 8])
def test_every_group_size(built, ng):
    names = ["cmb", "synch", "dust", "ff", "ame", "dust2", "synch", "dust"][:ng]
    case = _custom(200, 12, names)
    for q, c in enumerate(case[3]):       # distinct reference frequencies keep the blocks non-singular
        c.label = "%s%d" % (c.label, q)
        c.nu_ref = c.nu_ref * (1.0 + 0.07 * q)
        if c.nindices:
            c.indices = c.indices + (0.4 if q >= 6 else 0.05) * q   # the repeated synch / dust need clearly different SEDs
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    _, bad = eng.amp_sample(1, L.FLAG_T, "sample", 5, 6)
    obad = orc.amp_sample_direct(1, L.FLAG_T, "sample", 5, 6, "reference")
    assert bad == obad
    assert_amps_close(eng, orc, ng, 1e-7)   # 7-8 component blocks are ill-conditioned: condition number x eps


def test_all_masked_sky(built):
    case = make_case("C2", nside=4)
    case[1].masks[:] = 0.0
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    before = [eng.get_amplitude(l).copy() for l in range(len(comps))]
    for g in dpar.cg_groups:
        for f in g.pol_flag:
            _, bad = eng.amp_sample(g.cg_group, f, "sample", 1, 2)
            assert bad == 0
    for l in range(len(comps)):
        assert np.array_equal(eng.get_amplitude(l), before[l])      # nothing to solve: amplitudes untouched
    eng.index_sample(1, 0, 1, 10, "sample", 1, 3)
    assert np.all(eng.get_indices(1)[0, 0] == 0.0)                   # masked pixels of a swept index map become 0
    assert eng.sky_model_chisq(1, 3) == 0.0
    assert eng.chisq_cached(1, 1, 1) == 0.0


def test_non_spd_blocks_are_reported_and_left_unchanged(built):
    """Two components with identical SEDs make every block singular: counted, amplitudes untouched."""
    case = _custom(130, 4, ["synch", "synch"])
    for c in case[3]:
        c.indices[:] = -3.0
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    before = eng.get_amplitude(0).copy()
    _, bad = eng.amp_sample(1, L.FLAG_T, "optimize", 1, 2)
    obad = orc.amp_sample_direct(1, L.FLAG_T, "optimize", 1, 2, "reference")
    # an exactly singular block yields a pivot that is 0 up to rounding: whether it comes out <= 0 differs between
    # implementations, so only the presence of flagged units (and finite amplitudes elsewhere) is comparable
    assert bad > 0 and obad > 0
    assert np.isfinite(eng.get_amplitude(0)).all()
    assert np.array_equal(eng.get_amplitude(0)[:, ddata.masks[0] == 0], before[:, ddata.masks[0] == 0])


def test_bad_arguments_fail_loudly(built):
    case = make_case("C1", nside=2)
    dpar, ddata, bands, comps, meta = case
    eng, _ = pair(case)
    with pytest.raises(da.DangxError):
        eng.amp_sample(7, L.FLAG_T, "sample", 1, 2)                  # no such CG group
    with pytest.raises(da.DangxError):
        eng.amp_sample(1, L.FLAG_QU, "sample", 1, 2)                 # polarisation flag with nmaps == 1
    with pytest.raises(da.DangxError):
        eng.amp_sample(1, 3, "sample", 1, 2)                         # not a single poltype bit
    with pytest.raises(da.DangxError):
        eng.index_sample(0, 1, 1, 10, "sample", 1, 2)                # synch has one index
    with pytest.raises(da.DangxError):
        eng.index_sample(0, 0, 2, 10, "sample", 1, 2)                # map 2 with nmaps == 1
    with pytest.raises(da.DangxError):
        eng.sky_model_chisq(1, 3)


def test_host_windows_into_full_sky_arrays(built):
    """dangx_set_host_stride: a pixel-shard context reads and writes WINDOWS of the driver's full-sky arrays (plane q of
    the shard starts plane_stride doubles after plane q-1) -- what fortran/dangx_multi_mod.f90 does for every GPU.  Upload,
    put / get of amplitudes and indices and the sky / res / chi_map outputs through windows equal the packed path."""
    import ctypes as C
    from dang_amd import synth
    full = synth.make_sky("C2", nside=4, start="truth")
    dpar, ddata, bands, comps, meta = full
    npix = meta["npix_global"]
    shard = synth.make_sky("C2", nside=4, start="truth", rank=1, nranks=3)
    p0, n = shard[4]["pix0"], shard[4]["npix"]
    packed = da.Engine(shard[2], shard[3], shard[1], npix_global=npix, pix0=p0, device=0)
    # the same shard fed from windows of the full-sky arrays
    import copy
    sh = copy.deepcopy(shard)
    win = da.Engine(sh[2], sh[3], sh[1], npix_global=npix, pix0=p0, device=0)
    lib, h = win.lib, win.h
    assert lib.dangx_set_host_stride(h, npix) == 0

    def at(a):                      # address of pixel p0 of plane 0 of a full-sky array [..., npix]
        a = np.ascontiguousarray(a, dtype=np.float64)
        return a, a.ctypes.data + 8 * p0
    keep = []
    s, ps = at(ddata.sig_map); r, pr = at(ddata.rms_map); m, pm = at(ddata.masks)
    keep += [s, r, m]
    assert lib.dangx_upload_data(h, ps, pr, pm) == 0
    for l, c in enumerate(comps):
        a, pa = at(c.amplitude); keep.append(a)
        assert lib.dangx_put_amplitude(h, l, pa) == 0
        if c.nindices:
            x, px = at(c.indices); keep.append(x)
            assert lib.dangx_put_indices(h, l, px) == 0
    for e in (packed, win):
        e.amp_sample(1, L.FLAG_T, "sample", 3, 4)
        e.index_sample(1, 0, 1, 5, "sample", 3, 5)
    # read back through windows into full-sky arrays pre-filled with a sentinel: only the window changes
    for l, c in enumerate(comps):
        out = np.full((meta["nmaps"], npix), -7.0)
        assert lib.dangx_get_amplitude(h, l, out.ctypes.data + 8 * p0) == 0
        assert np.array_equal(out[:, p0:p0 + n], packed.get_amplitude(l))
        assert np.all(out[:, :p0] == -7.0) and np.all(out[:, p0 + n:] == -7.0)
        if c.nindices:
            outx = np.full((c.nindices, meta["nmaps"], npix), -7.0)
            assert lib.dangx_get_indices(h, l, outx.ctypes.data + 8 * p0) == 0
            assert np.array_equal(outx[:, :, p0:p0 + n], packed.get_indices(l))
            assert np.all(outx[:, :, :p0] == -7.0)
    nb = meta["nbands"]
    sky, res, chi = (np.full((nb, 3, npix), -7.0), np.full((nb, 3, npix), -7.0), np.full((3, npix), -7.0))
    cs = C.c_double(0.0)
    assert lib.dangx_sky_model_chisq(h, 1, 3, C.byref(cs), sky.ctypes.data + 8 * p0, res.ctypes.data + 8 * p0, chi.ctypes.data + 8 * p0) == 0
    s2, sky2, res2, chi2 = packed.sky_model_chisq(1, 3, want_maps=True)
    assert cs.value == s2 and np.array_equal(sky[:, :, p0:p0 + n], sky2, equal_nan=True) and np.array_equal(chi[:, p0:p0 + n], chi2)
    assert np.all(sky[:, :, :p0] == -7.0) and np.all(res[:, :, p0 + n:] == -7.0)
    assert lib.dangx_set_host_stride(h, n - 1) != 0          # a stride smaller than the shard is refused


def test_convert_maps_scales_adopted_device_buffers_in_place(built):
    import torch
    from dang_amd import synth
    dev = torch.device("cuda", 0)
    dpar, ddata, bands, comps, meta = synth.make_sky("C2", nside=4, device=dev, as_numpy=False, start="truth")
    sig0, rms0 = ddata.sig_map.clone(), ddata.rms_map.clone()
    eng = da.initialize(bands, comps, ddata, npix_global=meta["npix_global"], device=0)
    conv = da.convert_maps(ddata, ["uK_RJ", "uK_cmb", "MJy/sr", "uK_RJ", "uK_cmb"])
    eng.synchronize()
    for j in range(meta["nbands"]):
        assert torch.equal(ddata.sig_map[j], sig0[j] * conv[j]) and torch.equal(ddata.rms_map[j], rms0[j] * conv[j])
    assert conv[0] == 1.0 and conv[1] == 1.0 / eng.unit_conversion(1, "a2t") and conv[2] == 1.0 / eng.unit_conversion(2, "a2f")
