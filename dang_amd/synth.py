"""Deterministic synthetic HEALPix-shaped sky for tests and bench.py (SURVEY 8d recipe).

No HEALPix library is needed: every per-pixel random field is a counter hash of the GLOBAL
RING pixel index (splitmix64 + Box-Muller), so any pixel shard of any rank reproduces the same
sky.  Works on CPU (numpy-like, via torch CPU tensors) and on the GPU (torch cuda tensors, used
by bench.py to build the inputs directly in HBM).

Components follow the reference's model: power-law synchrotron, modified-blackbody dust, CMB,
free-free, log-normal AME (src/dang_component_mod.f90:886-1040).  IQU configurations use one
component set for T (CG group 1, flag T) and one for polarisation (CG group 2, flag Q+U),
which is how the reference has to be configured (one poltype flag per CG group).
"""
import math

import numpy as np
import torch

from . import _lib as L
from .api import BandInfo, DangCGGroup, DangComps, DangData, DangParams

H = 1.0545726691251021e-34 * 2.0 * 3.141592653589793238462643383279502884197
K_B = 1.3806503e-23
T_CMB = 2.7255

CONFIGS = {
    # name: nside, nbands, physical components, nmaps
    "C1": dict(nside=64, nbands=3, comps=["synch", "dust"], nmaps=1),
    "C2": dict(nside=256, nbands=5, comps=["cmb", "synch", "dust"], nmaps=3),
    "C3": dict(nside=1024, nbands=10, comps=["cmb", "synch", "dust", "ff"], nmaps=3),
    "C5": dict(nside=2048, nbands=20, comps=["cmb", "synch", "dust", "ff", "ame", "dust2"], nmaps=3),
}

# label -> (type, nu_ref GHz, amplitude sigma uK_RJ, [(index label, mean, sigma, sampled)])
PHYS = {
    "cmb": ("cmb", 100.0, 60.0, []),
    "synch": ("power-law", 30.0, 20.0, [("beta", -3.1, 0.1, True)]),
    "dust": ("mbb", 353.0, 100.0, [("beta", 1.6, 0.1, True), ("T", 19.6, 1.5, True)]),
    "ff": ("freefree", 40.0, 10.0, [("T_e", 7000.0, 0.0, False)]),
    "ame": ("lognormal", 22.0, 15.0, [("nu_p", 21.0, 2.0, True), ("w", 0.5, 0.0, False)]),
    "dust2": ("mbb", 857.0, 20.0, [("beta", 2.0, 0.0, False), ("T", 30.0, 0.0, False)]),
}
FIELD_IDS = {"amp": 1, "idx": 2, "noise": 3, "rms": 4}
_MASK64 = (1 << 64) - 1


def _i64(v):
    v &= _MASK64
    return v - (1 << 64) if v >= (1 << 63) else v


def _lsr(x, s):
    return (x >> s) & ((1 << (64 - s)) - 1)


def _splitmix64(x):
    x = x + _i64(0x9E3779B97F4A7C15)
    x = (x ^ _lsr(x, 30)) * _i64(0xBF58476D1CE4E5B9)
    x = (x ^ _lsr(x, 27)) * _i64(0x94D049BB133111EB)
    return x ^ _lsr(x, 31)


def _uniform(pix, seed, field, a=0, b=0, c=0):
    """u in (0,1) for int64 tensor of global pixel indices and a field label."""
    key = _i64(seed * 0x9E3779B97F4A7C15 + field * 0xD1B54A32D192ED03 + a * 0x8CB92BA72F3D8DD7 + b * 0xABC98388FB8FAC03 + c * 0x2545F4914F6CDD1D)
    h = _splitmix64(_splitmix64(pix ^ key) + key)
    return (_lsr(h, 11).to(torch.float64) + 0.5) * (1.0 / 9007199254740992.0)


def _normal(pix, seed, field, a=0, b=0, c=0):
    u1 = _uniform(pix, seed, field, a, b, 2 * c)
    u2 = _uniform(pix, seed, field, a, b, 2 * c + 1)
    return torch.sqrt(-2.0 * torch.log(u1)) * torch.cos(2.0 * math.pi * u2)


def band_freqs_ghz(nb):
    if nb == 1:
        return [20.0]
    return [20.0 * (857.0 / 20.0) ** (j / (nb - 1)) for j in range(nb)]


def sed_torch(ctype, nu_hz, nu_ref_hz, th0, th1):
    """SED in torch fp64, same formulas as src/dang_component_mod.f90:886-1040 (delta bandpass)."""
    if ctype == "power-law":
        return (nu_hz / nu_ref_hz) ** th0
    if ctype == "mbb":
        z = H / (K_B * th1)
        return (torch.exp(z * nu_ref_hz) - 1.0) / (torch.exp(z * nu_hz) - 1.0) * (nu_hz / nu_ref_hz) ** (th0 + 1.0)
    if ctype == "freefree":
        t15 = (th0 / 1.0e4) ** (-1.5)
        g = lambda nu: torch.log(torch.exp(5.960 - math.sqrt(3.0) / math.pi * torch.log(nu / 1.0e9 * t15)) + 2.71828)
        return g(torch.as_tensor(nu_hz, dtype=torch.float64, device=th0.device)) / \
            g(torch.as_tensor(nu_ref_hz, dtype=torch.float64, device=th0.device)) * (nu_hz / nu_ref_hz) ** (-2)
    if ctype == "lognormal":
        return torch.exp(-0.5 * (torch.log(nu_hz / (th0 * 1e9)) / th1) ** 2) * (nu_ref_hz / nu_hz) ** 2
    if ctype == "cmb":
        y = H * nu_hz / (K_B * T_CMB)
        return torch.as_tensor(1.0 / ((math.exp(y) - 1.0) ** 2 / (y * y * math.exp(y))), dtype=torch.float64)
    raise ValueError(ctype)


def make_sky(config="C1", nside=None, nbands=None, comps=None, nmaps=None, device="cpu", rank=0, nranks=1,
             seed_data=20240601, seed_sampler=1234, nsample=10, mask_frac=(0.45, 0.55), start="prior",
             as_numpy=None, solver="direct", fluct_mode="reference", gain=None, offset=None, balance=False):
    """Build (dpar, ddata, bands, component_list, truth) for one pixel shard.

    start = 'prior' (amplitudes 0, indices at the prior mean) or 'truth'.
    as_numpy: return numpy arrays (default when device == 'cpu').
    balance: shard boundaries by work (unmasked pixels) instead of by pixel count.
    """
    from .dist import balanced_bounds_run, shard_range

    cfg = dict(CONFIGS[config]) if config else {}
    nside = nside or cfg["nside"]
    nbands = nbands or cfg["nbands"]
    phys = comps or cfg["comps"]
    nmaps = nmaps or cfg["nmaps"]
    if as_numpy is None:
        as_numpy = (str(device) == "cpu")
    dev = torch.device(device)
    npix_global = 12 * nside * nside
    bounds = None
    if balance and nranks > 1:   # shards of equal work: the masked band is one run of RING pixels (dist.balanced_bounds_run)
        bounds = balanced_bounds_run(npix_global, nranks, math.ceil(mask_frac[0] * npix_global), math.ceil(mask_frac[1] * npix_global))
    pix0, npix = shard_range(npix_global, rank, nranks, bounds)
    pix = torch.arange(pix0, pix0 + npix, dtype=torch.int64, device=dev)

    freqs = band_freqs_ghz(nbands)
    bands = [BandInfo(label="band%03d" % (j + 1), nu_c=freqs[j]) for j in range(nbands)]

    # component sets: (suffix, group, flag, planes)
    sets = [("", 1, L.FLAG_T, [1])]
    if nmaps == 3:
        sets.append(("_P", 2, L.FLAG_QU, [2, 3]))
    component_list, truth = [], []
    sig = torch.zeros(nbands, nmaps, npix, dtype=torch.float64, device=dev)
    for si, (suf, group, flag, planes) in enumerate(sets):
        for ci, name in enumerate(phys):
            ctype, nu_ref, asig, idx = PHYS[name]
            nind = len(idx)
            amp_true = torch.zeros(nmaps, npix, dtype=torch.float64, device=dev)
            for k in planes:
                amp_true[k - 1] = asig * _normal(pix, seed_data, FIELD_IDS["amp"], si, ci, k)
            ind_true = torch.zeros(max(nind, 1), nmaps, npix, dtype=torch.float64, device=dev)
            ind_start = torch.zeros(max(nind, 1), nmaps, npix, dtype=torch.float64, device=dev)
            for q, (_, mean, s, _) in enumerate(idx):
                ind_true[q] = mean
                ind_start[q] = mean
                if s > 0:
                    val = mean + s * _normal(pix, seed_data, FIELD_IDS["idx"], si, ci, q)
                    for k in planes:  # Q and U share the index value (sampled jointly, map_n=-1)
                        ind_true[q, k - 1] = val
            for k in planes:
                th0 = ind_true[0, k - 1]
                th1 = ind_true[1, k - 1] if nind > 1 else th0
                for j in range(nbands):
                    sig[j, k - 1] += amp_true[k - 1] * sed_torch(ctype, freqs[j] * 1e9, nu_ref * 1e9, th0, th1)
            c = DangComps(
                label=name + suf, type=ctype, nu_ref=nu_ref, cg_group=group, sample_amplitude=True, nindices=nind,
                ind_label=[i[0] for i in idx], sample_index=[i[3] for i in idx], index_mode=[2] * nind,
                lnl_type=["chisq"] * nind, prior_type=["gaussian" if i[3] else "uniform" for i in idx],
                gauss_prior=[[i[1], i[2] if i[2] > 0 else 1.0] for i in idx],
                uni_prior=[[i[1] - 10 * (i[2] if i[2] > 0 else 1.0), i[1] + 10 * (i[2] if i[2] > 0 else 1.0)] for i in idx],
                step_size=[0.5 * i[2] for i in idx], pol_flag=[[flag]] * nind)
            a0 = amp_true.clone() if start == "truth" else torch.zeros_like(amp_true)
            i0 = (ind_true if start == "truth" else ind_start)[:nind].clone() if nind else None
            c.amplitude, c.indices = a0, i0
            component_list.append(c)
            truth.append(dict(amplitude=amp_true, indices=ind_true[:nind] if nind else None))

    rms = torch.empty_like(sig)
    for j in range(nbands):
        s0 = 0.5 + 0.1 * (j + 1)
        for k in range(1, nmaps + 1):
            rms[j, k - 1] = s0 * (0.5 + _uniform(pix, seed_data, FIELD_IDS["rms"], j, k))
            sig[j, k - 1] += rms[j, k - 1] * _normal(pix, seed_data, FIELD_IDS["noise"], j, k)
    frac = pix.to(torch.float64) / float(npix_global)
    m1 = torch.where((frac >= mask_frac[0]) & (frac < mask_frac[1]), 0.0, 1.0).to(torch.float64)
    masks = m1.unsqueeze(0).repeat(nmaps, 1).contiguous()
    # nump over the GLOBAL sky (an input to the path): unmasked pixels x planes
    gfrac_lo, gfrac_hi = mask_frac
    n_masked = len(range(math.ceil(gfrac_lo * npix_global), math.ceil(gfrac_hi * npix_global)))
    nump = float((npix_global - n_masked) * nmaps)

    g = np.ones(nbands) if gain is None else np.asarray(gain, dtype=np.float64)
    o = np.zeros(nbands) if offset is None else np.asarray(offset, dtype=np.float64)
    if gain is not None or offset is not None:  # data = gain*sky + offset on the T plane
        for j in range(nbands):
            sig[j, 0] = sig[j, 0] * g[j] + o[j]

    conv = (lambda t: t.cpu().numpy()) if as_numpy else (lambda t: t.contiguous())
    for c in component_list:
        c.amplitude = conv(c.amplitude)
        c.indices = conv(c.indices) if c.indices is not None else None
    ddata = DangData(sig_map=conv(sig), rms_map=conv(rms), masks=conv(masks), gain=g, offset=o,
                     pol_type=list(range(1, nmaps + 1)), nump=nump)
    groups = [DangCGGroup(cg_group=grp, i_max=100, converge=1e-8, sample=True, pol_flag=[flag])
              for (_, grp, flag, _) in sets]
    dpar = DangParams(ml_mode="sample", nsample=nsample, cg_groups=groups, seed=seed_sampler, solver=solver,
                      fluct_mode=fluct_mode)
    meta = dict(nside=nside, npix_global=npix_global, pix0=pix0, npix=npix, nbands=nbands, nmaps=nmaps,
                ncomp=len(component_list), freqs_ghz=freqs, truth=truth, phys=phys,
                bounds=bounds)   # shard boundaries (None = equal ranges): what dist.gather_maps needs to reassemble the sky
    return dpar, ddata, bands, component_list, meta


def add_qu_template(ddata, comps, meta, fit_bands=(7, 8, 9), amplitudes=(2.0, -1.5, 0.7), cg_group=2, seed=3):
    """A Q/U template (unit-variance Gaussian maps, zero on T) fitted at `fit_bands` with ONE amplitude per band for Q and U
    (src/dang_cg_mod.f90:1380-1382), its signal added to the data: the model class of SURVEY 8f rank 1 -- diffuse components beside a
    dust template -- on top of any make_sky configuration with torch maps.  Appends the component to `comps`; returns its index."""
    import torch
    from .api import DangComps
    npix, nb = meta["npix"], meta["nbands"]
    dev = ddata.sig_map.device
    g = torch.Generator(device="cpu").manual_seed(seed)
    tmpl = torch.zeros(3, npix, dtype=torch.float64)
    tmpl[1:] = torch.randn(2, npix, generator=g, dtype=torch.float64)
    corr = [j in tuple(fit_bands) for j in range(nb)]
    for k in (1, 2):
        for j, a in zip(fit_bands, amplitudes):
            ddata.sig_map[j, k] += a * tmpl[k].to(dev)
    comps.append(DangComps(label="tmpl", type="template", nu_ref=100.0, cg_group=cg_group, nindices=0, nfit=len(tuple(fit_bands)), corr=corr,
                           template=tmpl.numpy(), template_amplitudes=np.zeros((3, nb)),
                           amplitude=torch.zeros(3, npix, dtype=torch.float64, device=dev)))
    return len(comps) - 1


def add_monopole(ddata, comps, meta, fit_bands=(0, 8, 9), amplitudes=(3.0, -2.0, 5.0), cg_group=1):
    """A monopole (template = 1 on T, src/dang_component_mod.f90:593-595) fitted at `fit_bands` in the T group, its signal added to
    the data; its amplitudes become the band offsets (update_sky_model, src/dang_data_mod.f90:357-361).  Appends the component."""
    import torch
    from .api import DangComps
    npix, nb = meta["npix"], meta["nbands"]
    tmpl = np.zeros((3, npix))
    tmpl[0] = 1.0
    corr = [j in tuple(fit_bands) for j in range(nb)]
    for j, a in zip(fit_bands, amplitudes):
        ddata.sig_map[j, 0] += a
    comps.append(DangComps(label="mono", type="monopole", nu_ref=100.0, cg_group=cg_group, nindices=0, nfit=len(tuple(fit_bands)), corr=corr,
                           template=tmpl, template_amplitudes=np.zeros((3, nb)),
                           amplitude=torch.zeros(3, npix, dtype=torch.float64, device=ddata.sig_map.device)))
    return len(comps) - 1
