"""The whole Gibbs loop as a sampler (a property no single-call parity test sees): with the textbook fluctuation
term (`fluct_mode='correct'`: nb independent normals per unit, every component's own SED) the chain draws from the
posterior, and the chi^2 of a posterior DRAW has nb degrees of freedom per (pixel, plane) -- (nb - nc) from the
noise left in the data plus nc from the draw -- i.e. ddata%chisq -> 1.  With the reference's compute_sample_vector
(`fluct_mode='reference'`, SURVEY quirks 2 and 3: one eta per unit for all bands, written to the first component's
slot with the last component's SED, src/dang_cg_mod.f90:1008-1040) the amplitude draws are not posterior draws and
chi^2 settles an order of magnitude higher; that is the reference's behaviour and the default."""
import numpy as np
import pytest

import dang_amd as da
from dang_amd import synth

pytestmark = pytest.mark.gpu


def _run(fluct, niter, nside=32):
    dpar, ddata, bands, comps, meta = synth.make_sky("C3", nside=nside, fluct_mode=fluct)
    eng = da.initialize(bands, comps, ddata, npix_global=meta["npix_global"], device=0)
    trace = []
    for it in range(1, niter + 1):
        da.sample_cg_groups(dpar, ddata, it=it, defer_chisq=(it > 1))
        if it > 1:
            da.sample_spectral_parameters(dpar, ddata, it=it)
        trace.append(ddata.chisq)
    return np.array(trace), eng, comps, ddata


def test_textbook_sampler_converges_to_unit_chisq(built):
    trace, eng, comps, ddata = _run("correct", 200)
    tail = trace[-50:]
    assert np.all(np.isfinite(trace))
    assert abs(tail.mean() - 1.0) < 0.03, tail.mean()          # 12288 px x 3 planes x 10 bands: sigma ~ 2e-3 per iteration
    assert trace[0] > 5.0 and trace[-1] < trace[4] < trace[0]  # started far away (amplitudes 0, indices at the prior mean)
    # the sampled indices stay inside their hard bounds and scatter around the truth with about the prior width
    m = np.asarray(ddata.masks)[0] != 0
    for l, c in enumerate(comps):
        for j in range(c.nindices):
            if c.sample_index[j]:
                planes = [0] if c.pol_flag[j][0] == 1 else [1, 2]
                x = eng.get_indices(l)[j][planes][:, m]
                assert x.min() >= c.uni_prior[j][0] and x.max() <= c.uni_prior[j][1]
                assert abs(x.mean() - c.gauss_prior[j][0]) < 0.5 * c.gauss_prior[j][1]


def test_reference_fluctuation_term_is_not_a_posterior_draw(built):
    trace, *_ = _run("reference", 60)
    assert np.all(np.isfinite(trace))
    assert trace[-20:].mean() > 5.0   # measured ~13: the documented consequence of SURVEY quirks 2-3, reproduced on purpose
