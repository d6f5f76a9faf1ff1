"""ctypes wrapper of the CPU oracle (oracle/libdang_oracle.so).  TEST INFRASTRUCTURE ONLY.

Builds a dgo_ctx from the same host-side objects the product path takes
(dang_amd.api.BandInfo / DangComps / DangData with numpy arrays) so that the parity tests
feed identical inputs to both.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libdang_oracle.so")

TYPE_CODES = {"power-law": 1, "mbb": 2, "freefree": 3, "lognormal": 4, "cmb": 5, "T_cmb": 6, "template": 7, "monopole": 8,
              "hi_fit": 9}
LNL_CODES = {"chisq": 1, "marginal": 2, "prior": 3}
PRIOR_CODES = {"gaussian": 1, "uniform": 2, "jeffreys": 3}
ML_CODES = {"sample": 1, "optimize": 2}
FLUCT_CODES = {"correct": 0, "reference": 1}
_D = C.POINTER(C.c_double)


class Band(C.Structure):
    _fields_ = [("nu_c", C.c_double), ("n", C.c_int), ("nu0", _D), ("tau0", _D)]


class Comp(C.Structure):
    _fields_ = [("type", C.c_int), ("is_synch", C.c_int), ("nu_ref", C.c_double), ("nindices", C.c_int),
                ("cg_group", C.c_int), ("sample_amplitude", C.c_int), ("amplitude", _D), ("indices", _D),
                ("lnl_type", C.c_int * 2), ("prior_type", C.c_int * 2), ("gauss_prior", (C.c_double * 2) * 2),
                ("uni_prior", (C.c_double * 2) * 2), ("step_size", C.c_double * 2),
                ("nfit", C.c_int), ("corr", C.POINTER(C.c_int)), ("tmpl", _D), ("template_amplitudes", _D)]


class Ctx(C.Structure):
    _fields_ = [("npix", C.c_int), ("nmaps", C.c_int), ("nbands", C.c_int), ("ncomp", C.c_int),
                ("pix0", C.c_int64), ("sig", _D), ("rms", _D), ("mask", _D), ("gain", _D), ("offset", _D),
                ("bands", C.POINTER(Band)), ("comps", C.POINTER(Comp)), ("T_CMB", C.c_double), ("nthreads", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        src = [os.path.join(ORACLE_DIR, f) for f in ("dang_oracle.c", "dang_oracle.h")]
        if not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src if os.path.exists(s)):
            subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True)
        l = C.CDLL(LIB_PATH)
        l.dgo_const_h.restype = C.c_double
        l.dgo_const_kB.restype = C.c_double
        l.dgo_missval.restype = C.c_double
        l.dgo_rand_normal.restype = C.c_double
        l.dgo_rand_normal.argtypes = [C.c_double] * 4
        l.dgo_eval_normal_prior.restype = C.c_double
        l.dgo_eval_normal_prior.argtypes = [C.c_double] * 3
        l.dgo_uniform2.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, _D]
        l.dgo_B_nu.restype = C.c_double
        l.dgo_B_nu.argtypes = [C.c_double, C.c_double]
        l.dgo_bnu_prime_RJ.restype = C.c_double
        l.dgo_bnu_prime_RJ.argtypes = [C.c_double]
        l.dgo_a2t.restype = C.c_double
        l.dgo_a2t.argtypes = [C.POINTER(Ctx), C.c_int]
        l.dgo_eval_sed.restype = C.c_double
        l.dgo_eval_sed.argtypes = [C.POINTER(Ctx), C.c_int, C.c_int, C.c_int, C.c_int, _D]
        l.dgo_eval_signal.restype = C.c_double
        l.dgo_eval_signal.argtypes = [C.POINTER(Ctx), C.c_int, C.c_int, C.c_int, C.c_int, _D]
        l.dgo_group_size.restype = C.c_int64
        l.dgo_group_size.argtypes = [C.POINTER(Ctx), C.c_int, C.c_int, C.POINTER(C.c_int)]
        for name in ("dgo_compute_rhs",):
            getattr(l, name).argtypes = [C.POINTER(Ctx), C.c_int, C.c_int, _D]
        l.dgo_compute_Ax.argtypes = [C.POINTER(Ctx), C.c_int, C.c_int, _D, _D]
        l.dgo_compute_sample_vector.argtypes = [C.POINTER(Ctx), C.c_int, C.c_int, _D, _D]
        l.dgo_initialize_x.argtypes = [C.POINTER(Ctx), C.c_int, C.c_int, _D]
        l.dgo_unpack_amplitudes.argtypes = [C.POINTER(Ctx), C.c_int, C.c_int, _D]
        l.dgo_draw_eta.argtypes = [C.POINTER(Ctx), C.c_int, C.c_uint64, C.c_uint64, _D]
        l.dgo_cg_search.restype = C.c_int
        l.dgo_cg_search.argtypes = [C.POINTER(Ctx), C.c_int, C.c_int, _D, C.c_int, _D, _D, C.c_int, C.c_double, _D]
        l.dgo_amp_sample_cg.restype = C.c_int
        l.dgo_amp_sample_cg.argtypes = [C.POINTER(Ctx), C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_int,
                                        C.c_double, _D]
        l.dgo_amp_sample_direct.restype = C.c_int
        l.dgo_amp_sample_direct.argtypes = [C.POINTER(Ctx), C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64,
                                            C.c_uint64, C.POINTER(C.c_int64)]
        l.dgo_update_sky_model.argtypes = [C.POINTER(Ctx), _D, _D]
        l.dgo_compute_chisq.restype = C.c_double
        l.dgo_compute_chisq.argtypes = [C.POINTER(Ctx), _D, C.c_int, C.c_int, C.c_double, _D]
        l.dgo_evaluate_lnL.restype = C.c_double
        l.dgo_evaluate_lnL.argtypes = [C.c_int, C.c_int, C.c_int, _D, _D, _D, C.c_int64, C.c_int64, C.c_int, C.c_double]
        l.dgo_evaluate_marginal_lnL.restype = C.c_double
        l.dgo_evaluate_marginal_lnL.argtypes = [C.c_int, C.c_int, C.c_int, _D, _D, _D, C.c_int64, C.c_int64, C.c_int]
        l.dgo_sample_index_mh.restype = C.c_int64
        l.dgo_sample_index_mh.argtypes = [C.POINTER(Ctx), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64,
                                          C.c_uint64]
        l.dgo_nest2ring.restype = C.c_int64
        l.dgo_nest2ring.argtypes = [C.c_int, C.c_int64]
        l.dgo_udgrade.restype = None
        l.dgo_udgrade.argtypes = [C.c_int, _D, C.c_int, _D, C.c_int]
        l.dgo_sample_index_mh_coarse.restype = C.c_int64
        l.dgo_sample_index_mh_coarse.argtypes = [C.POINTER(Ctx), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64,
                                                 C.c_uint64, C.c_int, C.c_int]
        l.dgo_sample_index_fullsky.restype = C.c_int64
        l.dgo_sample_index_fullsky.argtypes = [C.POINTER(Ctx), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64,
                                               C.c_uint64, C.POINTER(C.c_int)]
        l.dgo_sample_index_fullsky_coarse.restype = C.c_int64
        l.dgo_sample_index_fullsky_coarse.argtypes = [C.POINTER(Ctx), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64,
                                                      C.c_uint64, C.POINTER(C.c_int), C.c_int, C.c_int]
        l.dgo_bnu_prime.restype = C.c_double
        l.dgo_bnu_prime.argtypes = [C.c_double, C.c_double]
        for name in ("dgo_a2f", "dgo_f2t"):
            getattr(l, name).restype = C.c_double
            getattr(l, name).argtypes = [C.POINTER(Ctx), C.c_int]
        l.dgo_normalize_bandpass.restype = None
        l.dgo_normalize_bandpass.argtypes = [_D, C.c_int, _D]
        l.dgo_convert_maps.restype = C.c_int
        l.dgo_convert_maps.argtypes = [C.POINTER(Ctx), C.POINTER(C.c_int), C.POINTER(C.c_int), _D]
        l.dgo_tune_perpixel.restype = None
        l.dgo_tune_perpixel.argtypes = [C.POINTER(Ctx), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64,
                                        C.POINTER(C.c_int)]
        l.dgo_fit_band_gain.restype = C.c_double
        l.dgo_fit_band_gain.argtypes = [C.POINTER(Ctx), _D, _D, C.c_int, C.c_int, C.c_uint64, C.c_uint64]
        _lib = l
    return _lib


def _p(a):
    return a.ctypes.data_as(_D)


class Oracle:
    """The reference's state (bp, component_list, ddata) as the oracle sees it.

    Amplitude / index maps are COPIED in; `amplitude(l)` / `indices(l)` read the oracle's state.
    """

    def __init__(self, bands, component_list, ddata, pix0=0, tcmb=2.7255, nthreads=1):
        self.L = lib()
        self.sig = np.ascontiguousarray(ddata.sig_map, dtype=np.float64)
        self.rms = np.ascontiguousarray(ddata.rms_map, dtype=np.float64)
        self.mask = np.ascontiguousarray(ddata.masks, dtype=np.float64)
        nb, nmaps, npix = self.sig.shape
        self.nb, self.nmaps, self.npix, self.ncomp = nb, nmaps, npix, len(component_list)
        self.gain = np.ones(nb) if ddata.gain is None else np.ascontiguousarray(ddata.gain, dtype=np.float64)
        self.offset = np.zeros(nb) if ddata.offset is None else np.ascontiguousarray(ddata.offset, dtype=np.float64)
        self._bands = (Band * nb)()
        self._bp = []
        for j, b in enumerate(bands):
            nu = float(b.nu_c)
            self._bands[j].nu_c = nu * 1e9 if nu < 1e9 else nu  # src/dang_bp_mod.f90:35-37
            if b.id == "delta" or b.nu0 is None:
                self._bands[j].n = 0
            else:
                nu0 = np.ascontiguousarray(b.nu0, dtype=np.float64)
                tau0 = np.ascontiguousarray(b.tau0, dtype=np.float64)
                self._bp += [nu0, tau0]
                self._bands[j].n = len(nu0)
                self._bands[j].nu0 = _p(nu0)
                self._bands[j].tau0 = _p(tau0)
        self._comps = (Comp * self.ncomp)()
        self.amp, self.idx = [], []
        self._glob = {}
        for l, c in enumerate(component_list):
            cc = self._comps[l]
            cc.type = TYPE_CODES[c.type]
            cc.is_synch = 1 if c.label.strip() == "synch" else 0
            nu_ref = float(c.nu_ref)
            cc.nu_ref = nu_ref * 1e9 if nu_ref < 1e7 else nu_ref  # src/dang_param_mod.f90:571-573
            cc.nindices = c.nindices
            cc.cg_group = c.cg_group
            cc.sample_amplitude = 1 if c.sample_amplitude else 0
            a = np.zeros((nmaps, npix)) if c.amplitude is None else np.array(c.amplitude, dtype=np.float64, copy=True)
            x = np.zeros((max(c.nindices, 1), nmaps, npix)) if c.indices is None else np.array(c.indices, dtype=np.float64, copy=True)
            self.amp.append(np.ascontiguousarray(a))
            self.idx.append(np.ascontiguousarray(x))
            cc.amplitude = _p(self.amp[l])
            cc.indices = _p(self.idx[l])
            if c.type in ("template", "monopole", "hi_fit"):
                corr = np.ascontiguousarray(np.asarray(c.corr, dtype=bool).astype(np.int32))
                tm = np.ascontiguousarray(c.template, dtype=np.float64)
                ta = np.zeros((nmaps, nb)) if c.template_amplitudes is None else np.array(c.template_amplitudes, dtype=np.float64, copy=True)
                ta = np.ascontiguousarray(ta)
                self._glob[l] = (corr, tm, ta)
                cc.nfit = int(c.nfit)
                cc.corr = corr.ctypes.data_as(C.POINTER(C.c_int))
                cc.tmpl = _p(tm)
                cc.template_amplitudes = _p(ta)
            for q in range(c.nindices):
                cc.lnl_type[q] = LNL_CODES[c.lnl_type[q]] if q < len(c.lnl_type) else 1
                cc.prior_type[q] = PRIOR_CODES[c.prior_type[q]] if q < len(c.prior_type) else 2
                gp = c.gauss_prior[q] if q < len(c.gauss_prior) else [0.0, 1.0]
                up = c.uni_prior[q] if q < len(c.uni_prior) else [-1e300, 1e300]
                cc.gauss_prior[q][0], cc.gauss_prior[q][1] = gp
                cc.uni_prior[q][0], cc.uni_prior[q][1] = up
                cc.step_size[q] = c.step_size[q] if q < len(c.step_size) else 0.0
        self.ctx = Ctx(npix, nmaps, nb, self.ncomp, pix0, _p(self.sig), _p(self.rms), _p(self.mask), _p(self.gain),
                       _p(self.offset), self._bands, self._comps, tcmb, nthreads)

    @property
    def c(self):
        return C.byref(self.ctx)

    def amplitude(self, l):
        return self.amp[l]

    def template_amplitudes(self, l):
        return self._glob[l][2]

    def indices(self, l):
        return self.idx[l]

    def eval_sed_map(self, comp, band, map_n):
        return np.array([self.L.dgo_eval_sed(self.c, comp, band, i, map_n, None) for i in range(self.npix)])

    def eval_sed(self, comp, band, theta):
        th = np.ascontiguousarray(theta, dtype=np.float64)
        return self.L.dgo_eval_sed(self.c, comp, band, 0, 1, _p(th))

    def group_size(self, group, flag):
        return self.L.dgo_group_size(self.c, group, flag, None)

    def compute_rhs(self, group, flag):
        b = np.empty(self.group_size(group, flag))
        self.L.dgo_compute_rhs(self.c, group, flag, _p(b))
        return b

    def compute_Ax(self, group, flag, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        r = np.empty_like(x)
        self.L.dgo_compute_Ax(self.c, group, flag, _p(x), _p(r))
        return r

    def compute_sample_vector(self, group, flag, eta):
        eta = np.ascontiguousarray(eta, dtype=np.float64)
        r = np.empty(self.group_size(group, flag))
        self.L.dgo_compute_sample_vector(self.c, group, flag, _p(eta), _p(r))
        return r

    def draw_eta(self, flag, seed, stream):
        eta = np.empty((2 if flag == 8 else 1) * self.npix)
        self.L.dgo_draw_eta(self.c, flag, seed, stream, _p(eta))
        return eta

    def initialize_x(self, group, flag):
        x = np.empty(self.group_size(group, flag))
        self.L.dgo_initialize_x(self.c, group, flag, _p(x))
        return x

    def cg_search(self, group, flag, b, ml_mode, eta, x, i_max, converge):
        trace = np.full(i_max + 1, np.nan)
        x = np.ascontiguousarray(x, dtype=np.float64).copy()
        eta = np.zeros(1) if eta is None else np.ascontiguousarray(eta, dtype=np.float64)
        it = self.L.dgo_cg_search(self.c, group, flag, _p(np.ascontiguousarray(b)), ML_CODES[ml_mode], _p(eta), _p(x),
                                  i_max, converge, _p(trace))
        return it, x, trace

    def amp_sample_cg(self, group, flag, ml_mode, seed, stream, i_max=100, converge=1e-8):
        return self.L.dgo_amp_sample_cg(self.c, group, flag, ML_CODES[ml_mode], seed, stream, i_max, converge, None)

    def amp_sample_direct(self, group, flag, ml_mode, seed, stream, fluct_mode="reference"):
        bad = C.c_int64(0)
        rc = self.L.dgo_amp_sample_direct(self.c, group, flag, ML_CODES[ml_mode], FLUCT_CODES[fluct_mode], seed, stream,
                                          C.byref(bad))
        assert rc == 0
        return bad.value

    def sky_model(self):
        sky = np.empty((self.nb, self.nmaps, self.npix))
        res = np.empty_like(sky)
        self.L.dgo_update_sky_model(self.c, _p(sky), _p(res))
        return sky, res

    def chisq(self, pol_lo, pol_hi, nump, sky=None):
        if sky is None:
            sky, _ = self.sky_model()
        chi = np.empty((self.nmaps, self.npix))
        v = self.L.dgo_compute_chisq(self.c, _p(np.ascontiguousarray(sky)), pol_lo, pol_hi, nump, _p(chi))
        return v, chi

    def sample_index_mh_coarse(self, comp, nind, map_n, nsample, ml_mode, seed, stream, nside, sample_nside):
        return self.L.dgo_sample_index_mh_coarse(self.c, comp, nind, map_n, nsample, ML_CODES[ml_mode], seed, stream,
                                                 nside, sample_nside)

    def sample_index_fullsky(self, comp, nind, map_n, nsample, ml_mode, seed, stream, tuned=True):
        t = C.c_int(1 if tuned else 0)
        acc = self.L.dgo_sample_index_fullsky(self.c, comp, nind, map_n, nsample, ML_CODES[ml_mode], seed, stream, C.byref(t))
        return acc, bool(t.value), self._comps[comp].step_size[nind]

    def a2t(self, band):
        return self.L.dgo_a2t(self.c, band)

    def a2f(self, band):
        return self.L.dgo_a2f(self.c, band)

    def f2t(self, band):
        return self.L.dgo_f2t(self.c, band)

    def convert_maps(self, units, cg_map=None):
        u = (C.c_int * self.nb)(*[{"uK_RJ": 0, "uK_cmb": 1, "MJy/sr": 2}.get(x, 99) for x in units])
        cg = None if cg_map is None else (C.c_int * self.nb)(*[int(bool(x)) for x in cg_map])
        conv = np.ones(self.nb)
        rc = self.L.dgo_convert_maps(self.c, u, cg, _p(conv))
        assert rc == 0, "Not a unit"
        return conv

    def tune_perpixel(self, comp, nind, map_n, nsample, ml_mode, seed, stream, tuned=False):
        t = C.c_int(1 if tuned else 0)
        self.L.dgo_tune_perpixel(self.c, comp, nind, map_n, nsample, ML_CODES[ml_mode], seed, stream, C.byref(t))
        return bool(t.value), self._comps[comp].step_size[nind]

    def sample_index_fullsky_coarse(self, comp, nind, map_n, nsample, ml_mode, seed, stream, nside, sample_nside, tuned=True):
        t = C.c_int(1 if tuned else 0)
        acc = self.L.dgo_sample_index_fullsky_coarse(self.c, comp, nind, map_n, nsample, ML_CODES[ml_mode], seed, stream,
                                                     C.byref(t), nside, sample_nside)
        return acc, bool(t.value), self._comps[comp].step_size[nind]

    def fit_band_gain(self, band, ml_mode, seed, stream):
        sky, res = self.sky_model()
        return self.L.dgo_fit_band_gain(self.c, _p(sky), _p(res), band, ML_CODES[ml_mode], seed, stream)

    def sample_index_mh(self, comp, nind, map_n, nsample, ml_mode, seed, stream):
        return self.L.dgo_sample_index_mh(self.c, comp, nind, map_n, nsample, ML_CODES[ml_mode], seed, stream)


def nest2ring(nside, ipnest):
    return int(lib().dgo_nest2ring(nside, ipnest))


def udgrade(mode, m, nside_in, nside_out):
    m = np.ascontiguousarray(m, dtype=np.float64)
    out = np.empty(12 * nside_out * nside_out)
    lib().dgo_udgrade(mode, m.ctypes.data_as(_D), nside_in, out.ctypes.data_as(_D), nside_out)
    return out


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().dgo_philox4x32_10(c, k, o)
    return list(o)


def uniform2(seed, stream, pix, draw):
    u = (C.c_double * 2)()
    lib().dgo_uniform2(seed, stream, pix, draw, u)
    return u[0], u[1]
