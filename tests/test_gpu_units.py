"""SURVEY 8f rank 3 (remainder): a2t / a2f / f2t, normalize_bandpass and convert_maps behind the ABI
(src/dang_bp_mod.f90:62-81, 181-274; src/dang_data_mod.f90:429-463) against the oracle's restatement."""
import numpy as np
import pytest

import dang_amd as da
from dang_amd import _lib as L

import oracle_ffi as O
from util import make_case, pair

pytestmark = pytest.mark.gpu


def _bp_case():
    def tweak(dpar, ddata, bands, comps):
        rng = np.random.default_rng(3)
        for b in bands[1::2]:                       # every second band carries a (raw, un-normalised) bandpass
            nu = b.nu_c * 1e9 * np.linspace(0.85, 1.15, 9)
            nu[0] = 0.0                             # a sample the reference skips (`if (bp%nu0(i) == 0.d0) cycle`)
            b.id, b.nu0 = "HFI_cmb", nu
            b.tau0 = da.normalize_bandpass(rng.uniform(0.1, 1.0, nu.size))
    return make_case("C2", nside=4, start="truth", tweak=tweak, gain=None, offset=[0.3, -0.2, 0.0, 1.5, 0.7])


def test_normalize_bandpass():
    t = np.array([0.5, 1.25, 3.0, 0.25])
    out = da.normalize_bandpass(t)
    want = np.empty(4)
    O.lib().dgo_normalize_bandpass(O._p(t), 4, O._p(want))
    assert np.array_equal(out, want) and abs(out.sum() - 1.0) < 1e-15


def test_unit_conversions_match_oracle(built):
    case = _bp_case()
    eng, orc = pair(case)
    nb = case[4]["nbands"]
    for j in range(nb):
        for which, ref in (("a2t", orc.a2t(j)), ("a2f", orc.a2f(j)), ("f2t", orc.f2t(j))):
            got = eng.unit_conversion(j, which)
            assert abs(got - ref) <= 4e-16 * abs(ref), (j, which, got, ref)
        # the three tie together: a2f * f2t = a2t * (1e14 as real(4)) / 1e14 for a delta band
        if case[2][j].id == "delta":
            r = eng.unit_conversion(j, "a2f") * eng.unit_conversion(j, "f2t") / eng.unit_conversion(j, "a2t")
            assert abs(r - float(np.float32(1e14)) / 1e14) <= 1e-14
    # T_CMB enters a2t and f2t
    eng.set_tcmb(2.9)
    orc2 = O.Oracle(case[2], case[3], case[1], tcmb=2.9)
    assert abs(eng.unit_conversion(1, "a2t") - orc2.a2t(1)) <= 4e-16 * orc2.a2t(1)
    assert abs(eng.unit_conversion(0, "f2t") - orc2.f2t(0)) <= 4e-16 * orc2.f2t(0)


def test_convert_maps_on_resident_maps_matches_oracle(built):
    from test_oracle_templates_cpu import add_globals
    case = _bp_case()
    dpar, ddata, bands, comps, meta = case
    add_globals(dpar, ddata, bands, comps, ("monopole",), 1, fit_bands=[0, 2])
    units = ["uK_RJ", "uK_cmb", "MJy/sr", "uK_cmb", "MJy/sr"]
    cg_map = [False, False, False, True, False]          # band 3 is a swapped-in map: left alone (:435)
    eng, orc = pair(case)                                # the oracle copies sig/rms: converted independently below
    # handing a monopole's (still zero) amplitudes to the device set the band offsets, as update_sky_model does
    # (src/dang_data_mod.f90:357-361); the reference converts BEFORE its first update_sky_model, with the offsets it read
    # from file -- restore them, then convert
    eng.set_calibration(ddata.gain, ddata.offset)
    conv = da.convert_maps(ddata, units, cg_map)
    oconv = orc.convert_maps(units, cg_map)
    assert np.array_equal(conv, oconv) and conv[0] == 1.0 and conv[3] == 1.0 and conv[1] != 1.0
    assert np.array_equal(ddata.offset, orc.offset)
    # the resident maps were scaled: the sky model / chi^2 of both sides still agree, and so does an amplitude solve
    s, sky, res, chi = eng.sky_model_chisq(1, 3, want_maps=True)
    osky, ores = orc.sky_model()
    good = ddata.masks[0] != 0
    assert np.abs(res - ores)[:, :, good].max() <= 1e-11 * np.abs(ores[:, :, good]).max()
    ochisq, _ = orc.chisq(1, 3, ddata.nump, osky)
    assert abs(s / meta["nbands"] / ddata.nump - ochisq) <= 1e-10 * ochisq
    l = len(comps) - 1                                   # the monopole took the converted offsets (:453-457)
    assert np.array_equal(eng.get_template_amplitudes(l)[0], orc.template_amplitudes(l)[0])
    assert np.array_equal(eng.get_template_amplitudes(l)[0], ddata.offset)
    with pytest.raises(da.DangxError, match="Not a unit"):
        eng.convert_maps(["uK_RJ", "K_cmb", "uK_RJ", "uK_RJ", "uK_RJ"])


def test_refresh_host_state_gives_the_output_side_current_arrays(built):
    """write_maps reads c%amplitude / c%indices / ddata%sky_model / res_map / chi_map from the host
    (src/dang_data_mod.f90:573-664): refresh_host_state makes them current at the map-output cadence."""
    case = make_case("C2", nside=4)
    dpar, ddata, bands, comps, meta = case
    eng, orc = pair(case)
    for it in (1, 2):
        da.sample_cg_groups(dpar, ddata, it=it)
        for g in dpar.cg_groups:
            f = g.pol_flag[0]
            orc.amp_sample_direct(g.cg_group, f, "sample", dpar.seed, da.stream_id(it, 0, g.cg_group, 0, f), "reference")
    da.refresh_host_state(ddata)
    osky, ores = orc.sky_model()
    ochisq, ochi = orc.chisq(1, 3, ddata.nump, osky)
    good = ddata.masks[0] != 0
    scale = np.abs(osky[:, :, good]).max()
    assert np.abs(ddata.sky_model - osky)[:, :, good].max() <= 1e-11 * scale
    assert np.abs(ddata.res_map - ores)[:, :, good].max() <= 1e-11 * scale
    assert np.abs(ddata.chi_map - ochi).max() <= 1e-11 * max(ochi.max(), 1.0)
    assert abs(ddata.chisq - ochisq) <= 1e-10 * ochisq
    for l, c in enumerate(comps):
        assert np.abs(c.amplitude - orc.amplitude(l)).max() <= 1e-9 * max(np.abs(orc.amplitude(l)).max(), 1.0)
