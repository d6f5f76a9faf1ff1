"""Host-side mirror of the reference's module API for the Gibbs inner loop.

The reference is a Fortran program whose hot path is reached through two calls,
`call sample_cg_groups(dpar, ddata)` (src/dang.f90:101, src/dang_cg_mod.f90:142-177) and
`call sample_spectral_parameters(dpar, ddata)` (src/dang.f90:106, src/dang_sample_mod.f90:21-86).
This module keeps those names, argument meanings and the state they read/write
(`dang_params`, `dang_data`, `dang_comps`, `dang_cg_group`, `bandinfo`), and routes the work
through the C ABI of libdangx.so (include/dangx.h).  It is the Python twin of
fortran/dangx_mod.f90 + the wrapper shown in INTEGRATION.md; tests and bench.py use it.

Arrays keep the Fortran memory order, i.e. numpy/torch shape [band][map][pix] for
sig_map/rms_map, [map][pix] for masks/amplitude and [ind][map][pix] for indices.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

import math

from . import _lib as L
from . import dist as _dist

MISSVAL = -1.6375e30  # src/dang_util_mod.f90:19


# --------------------------------------------------------------------------- state types

@dataclass
class BandInfo:
    """type bandinfo, src/dang_bp_mod.f90:7-12."""
    label: str
    nu_c: float                      # GHz (<1e9) or Hz, src/dang_bp_mod.f90:35-37
    id: str = "delta"
    nu0: Optional[np.ndarray] = None  # Hz
    tau0: Optional[np.ndarray] = None


@dataclass
class DangComps:
    """type dang_comps, src/dang_component_mod.f90:12-48 (fields the hot path reads/writes)."""
    label: str
    type: str                         # 'power-law' | 'mbb' | 'freefree' | 'lognormal' | 'cmb'
    nu_ref: float                     # GHz (<1e7) or Hz, src/dang_param_mod.f90:571-573
    cg_group: int = 1
    sample_amplitude: bool = True
    nindices: int = 0
    ind_label: List[str] = field(default_factory=list)
    sample_index: List[bool] = field(default_factory=list)
    index_mode: List[int] = field(default_factory=list)      # 2 = per-pixel (the built path)
    lnl_type: List[str] = field(default_factory=list)        # 'chisq' | 'marginal' | 'prior'
    prior_type: List[str] = field(default_factory=list)      # 'gaussian' | 'uniform' | 'jeffreys'
    gauss_prior: List[List[float]] = field(default_factory=list)  # [mean, std]
    uni_prior: List[List[float]] = field(default_factory=list)    # [low, high]
    step_size: List[float] = field(default_factory=list)
    pol_flag: List[List[int]] = field(default_factory=list)  # per index: list of poltype bit flags
    tuned: List[bool] = field(default_factory=list)          # c%tuned (empty = all tuned)
    sample_nside: List[int] = field(default_factory=list)    # c%sample_nside(nind) (empty / 0 = nside)
    # global-amplitude types ('template', 'monopole', 'hi_fit'), src/dang_component_mod.f90:17-33
    nfit: int = 0
    corr: List[bool] = field(default_factory=list)           # c%corr(j): is band j fitted?
    template: Optional[np.ndarray] = None                    # c%template, [nmaps][npix]
    template_amplitudes: Optional[np.ndarray] = None         # c%template_amplitudes(band,map) stored [map][band]
    amplitude: Optional[np.ndarray] = None                   # [nmaps][npix]
    indices: Optional[np.ndarray] = None                     # [nindices][nmaps][npix]


@dataclass
class DangCGGroup:
    """type dang_cg_group, src/dang_cg_mod.f90:16-42."""
    cg_group: int
    i_max: int = 100
    converge: float = 1e-8
    sample: bool = True
    pol_flag: List[int] = field(default_factory=lambda: [L.FLAG_T])


@dataclass
class DangParams:
    """The dang_params fields that drive the hot path (src/dang_param_mod.f90:370-400)."""
    ml_mode: str = "sample"           # ML_MODE
    nsample: int = 10                 # NUMSAMPLE
    cg_groups: List[DangCGGroup] = field(default_factory=list)
    # builder-added knobs (the reference has no seed: RANDOM_SEED() is unseeded, src/dang.f90:67)
    seed: int = 1234
    solver: str = "direct"            # 'direct' (MI355X block solve) | 'cg' (reference algorithm on device)
    fluct_mode: str = "reference"     # 'reference' reproduces compute_sample_vector's quirks | 'correct'


@dataclass
class DangData:
    """type dang_data, src/dang_data_mod.f90:9-61 (hot-path fields)."""
    sig_map: object                   # [nbands][nmaps][npix]  numpy (host) or torch cuda tensor
    rms_map: object
    masks: object                     # [nmaps][npix]
    gain: np.ndarray = None
    offset: np.ndarray = None
    pol_type: List[int] = field(default_factory=lambda: [1])
    nump: float = 0.0                 # number of unmasked (pixel, map) entries; an INPUT
    chisq: float = 0.0
    chisq_after_amp: float = 0.0      # chi^2 of the state the amplitude phase left (when deferred)
    fit_gain: List[bool] = field(default_factory=list)   # ddata%fit_gain(:)
    conversion: np.ndarray = None     # ddata%conversion(:), set by convert_maps
    sky_model: object = None          # refreshed by refresh_host_state (map-output cadence)
    res_map: object = None
    chi_map: object = None
    engine: object = None


def return_poltype_flag(string):
    """return_poltype_flag, src/dang_util_mod.f90:228-292: 'T', 'Q', 'U', 'Q+U' -> bit flags 1, 2, 4, 8, one list
    entry per flag.  As in the reference 'T+Q+U' sets local_flag = 0, which no `iand(flag, 0)` test can ever
    match, so it yields no usable flag."""
    local_flag, nflag = 0, 0
    for tok in string.split(","):
        tok = tok.strip()
        if tok == "T":
            local_flag += 1; nflag += 1
        elif tok == "Q":
            local_flag += 2; nflag += 1
        elif tok == "U":
            local_flag += 4; nflag += 1
        elif tok == "Q+U":
            local_flag += 8; nflag += 1
        elif tok == "T+Q+U":
            local_flag = 0; nflag += 1
    return [1 << j for j in range(4) if local_flag & (1 << j)]


def stream_id(it, phase, a=0, b=0, c=0):
    """64-bit random-stream label: Gibbs iteration, phase (0 amp / 1 index), and up to three small ids."""
    return ((int(it) & 0xFFFFFFFF) << 32) | ((phase & 0xF) << 28) | ((a & 0xFFF) << 16) | ((b & 0xFF) << 8) | (c & 0xFF)


# --------------------------------------------------------------------------- engine

def _is_torch(x):
    return type(x).__module__.startswith("torch")


class DangxError(RuntimeError):
    pass


class Engine:
    """Owns one dangx context = one pixel shard on one MI355X."""

    def __init__(self, bands, component_list, ddata, npix_global=None, pix0=0, device=-1, stream=None, tcmb=None):
        self.lib = L.load()
        self.bands = bands
        self.component_list = component_list
        self.ddata = ddata
        sig = ddata.sig_map
        nb, nmaps, npix = (int(s) for s in sig.shape)
        self.npix, self.nmaps, self.nbands, self.ncomp = npix, nmaps, nb, len(component_list)
        self.pix0 = int(pix0)
        self.npix_global = int(npix_global if npix_global is not None else npix)
        self._keep = []
        dims = L.Dims(npix, nmaps, nb, self.ncomp, self.pix0, self.npix_global, device, 0)
        h = C.c_void_p()
        rc = self.lib.dangx_create(C.byref(h), C.byref(dims))
        if rc != 0:
            raise DangxError("dangx_create failed with status %d (3 = no HIP device: the product path has no CPU fallback)" % rc)
        self.h = h
        if stream is not None:
            self._chk(self.lib.dangx_set_stream(self.h, C.c_void_p(stream)))
        for j, b in enumerate(bands):
            if b.id == "delta" or b.nu0 is None:
                self._chk(self.lib.dangx_set_band(self.h, j, float(b.nu_c), 0, None, None))
            else:
                nu0 = np.ascontiguousarray(b.nu0, dtype=np.float64)
                tau0 = np.ascontiguousarray(b.tau0, dtype=np.float64)
                self._chk(self.lib.dangx_set_band(self.h, j, float(b.nu_c), len(nu0), nu0.ctypes.data, tau0.ctypes.data))
        if tcmb is not None:
            self._chk(self.lib.dangx_set_tcmb(self.h, float(tcmb)))
        for l, c in enumerate(component_list):
            self._chk(self.lib.dangx_set_component(self.h, l, C.byref(comp_desc(c))))
        gain = np.ones(nb) if ddata.gain is None else np.ascontiguousarray(ddata.gain, dtype=np.float64)
        off = np.zeros(nb) if ddata.offset is None else np.ascontiguousarray(ddata.offset, dtype=np.float64)
        self._chk(self.lib.dangx_set_calibration(self.h, gain.ctypes.data, off.ctypes.data))
        if _is_torch(sig):
            for t in (ddata.sig_map, ddata.rms_map, ddata.masks):
                assert t.is_cuda and t.is_contiguous() and t.dtype.is_floating_point and t.element_size() == 8
            self._keep += [ddata.sig_map, ddata.rms_map, ddata.masks]
            self._chk(self.lib.dangx_adopt_device_data(self.h, ddata.sig_map.data_ptr(), ddata.rms_map.data_ptr(),
                                                       ddata.masks.data_ptr()))
        else:
            s, r, m = (np.ascontiguousarray(a, dtype=np.float64) for a in (ddata.sig_map, ddata.rms_map, ddata.masks))
            self._chk(self.lib.dangx_upload_data(self.h, s.ctypes.data, r.ctypes.data, m.ctypes.data))
        for l, c in enumerate(component_list):
            if c.type in ("template", "monopole", "hi_fit"):
                tm = np.ascontiguousarray(c.template, dtype=np.float64)
                corr = np.ascontiguousarray(np.asarray(c.corr, dtype=bool).astype(np.int32))
                assert tm.shape == (nmaps, npix) and corr.size == nb
                self._chk(self.lib.dangx_set_template(self.h, l, tm.ctypes.data, corr.ctypes.data, int(c.nfit)))
                if c.template_amplitudes is not None:
                    self.put_template_amplitudes(l, c.template_amplitudes)
        self._adopted = {}
        for l, c in enumerate(component_list):
            if c.amplitude is not None and _is_torch(c.amplitude) and c.amplitude.is_cuda:
                # torch owns the HBM buffers; the library works on them in place
                amp = c.amplitude
                idx = c.indices if c.nindices > 0 else None
                assert amp.is_contiguous() and tuple(amp.shape) == (nmaps, npix) and amp.element_size() == 8
                if idx is not None:
                    assert idx.is_cuda and idx.is_contiguous() and tuple(idx.shape) == (c.nindices, nmaps, npix)
                self._adopted[l] = (amp, idx)
                self._chk(self.lib.dangx_adopt_device_state(self.h, l, amp.data_ptr(),
                                                            idx.data_ptr() if idx is not None else None))
                continue
            if c.amplitude is not None:
                self.put_amplitude(l, c.amplitude)
            if c.nindices > 0 and c.indices is not None:
                self.put_indices(l, c.indices)
        self._allreduce_cb = None
        rank, nranks = _dist.world()
        if _dist.active():
            self.set_allreduce(_dist.allreduce_sum_inplace_host, is_root=(rank == 0))

    def set_allreduce(self, fn, is_root=True):
        """Pixel-sharded runs: `fn(numpy float64 array)` sums the array over all ranks in place (see
        dangx_set_allreduce in include/dangx.h).  Installed automatically when torch.distributed is initialised
        with more than one rank; fn=None returns to single-rank behaviour."""
        if fn is None:
            self._allreduce_cb = None
            self._chk(self.lib.dangx_set_allreduce(self.h, None, None, 1))
            return

        def cb(_user, buf, n):
            try:
                fn(np.ctypeslib.as_array(buf, shape=(int(n),)))
                return 0
            except Exception as e:  # an exception must not unwind through the C frames
                import sys
                print("dangx all-reduce callback failed: %r" % (e,), file=sys.stderr)
                return 1
        self._allreduce_cb = L.ALLREDUCE_FN(cb)   # keep the trampoline alive as long as the context
        self._chk(self.lib.dangx_set_allreduce(self.h, C.cast(self._allreduce_cb, C.c_void_p), None, 1 if is_root else 0))

    # -- plumbing
    def _chk(self, rc):
        if rc != 0:
            raise DangxError(self.lib.dangx_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.dangx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        self._chk(self.lib.dangx_synchronize(self.h))

    def put_amplitude(self, l, a):
        if _is_torch(a):
            a = a.cpu().numpy()
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.shape == (self.nmaps, self.npix)
        self._chk(self.lib.dangx_put_amplitude(self.h, l, a.ctypes.data))

    def get_amplitude(self, l):
        if l in self._adopted:
            self.synchronize()
            return self._adopted[l][0].cpu().numpy()
        a = np.empty((self.nmaps, self.npix))
        self._chk(self.lib.dangx_get_amplitude(self.h, l, a.ctypes.data))
        return a

    def put_indices(self, l, x):
        if _is_torch(x):
            x = x.cpu().numpy()
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert x.shape == (self.component_list[l].nindices, self.nmaps, self.npix)
        self._chk(self.lib.dangx_put_indices(self.h, l, x.ctypes.data))

    def get_indices(self, l):
        if l in self._adopted:
            self.synchronize()
            return self._adopted[l][1].cpu().numpy()
        x = np.empty((self.component_list[l].nindices, self.nmaps, self.npix))
        self._chk(self.lib.dangx_get_indices(self.h, l, x.ctypes.data))
        return x

    def put_template_amplitudes(self, l, ta):
        ta = np.ascontiguousarray(ta, dtype=np.float64)
        assert ta.shape == (self.nmaps, self.nbands)
        self._chk(self.lib.dangx_put_template_amplitudes(self.h, l, ta.ctypes.data))

    def get_template_amplitudes(self, l):
        ta = np.empty((self.nmaps, self.nbands))
        self._chk(self.lib.dangx_get_template_amplitudes(self.h, l, ta.ctypes.data))
        return ta

    def pull_state(self):
        """Copy the resident amplitude / index maps back into component_list (before output)."""
        for l, c in enumerate(self.component_list):
            c.amplitude = self.get_amplitude(l)
            if c.nindices > 0:
                c.indices = self.get_indices(l)
            if c.type in ("template", "monopole", "hi_fit"):
                c.template_amplitudes = self.get_template_amplitudes(l)

    # -- hot path
    def amp_sample(self, group, flag, ml_mode, seed, stream, solver="direct", fluct_mode="reference",
                   i_max=100, converge=1e-8, want_counts=True):
        it, bad = C.c_int(0), C.c_int64(0)
        self._chk(self.lib.dangx_amp_sample(
            self.h, group, flag, L.ML_CODES[ml_mode], L.SOLVER_CG if solver == "cg" else L.SOLVER_DIRECT,
            L.FLUCT_REFERENCE if fluct_mode == "reference" else L.FLUCT_CORRECT, seed, stream, i_max, converge,
            C.byref(it) if want_counts else None, C.byref(bad) if want_counts else None))
        return it.value, bad.value

    def schur_info(self):
        """((global-row residual relative to b, relative to the size of the row's terms), refinement steps) of the last
        direct solve of a template group."""
        r, n = (C.c_double * 2)(), C.c_int(0)
        self._chk(self.lib.dangx_schur_info(self.h, r, C.byref(n)))
        return (r[0], r[1]), n.value

    def amp_residual(self, group, flag, ml_mode, seed, stream):
        """(|b - A x| / |b|, largest relative global-row residual) of the reference's system at the current amplitudes."""
        out = (C.c_double * 2)()
        self._chk(self.lib.dangx_amp_residual(self.h, group, flag, L.ML_CODES[ml_mode], seed, stream, out))
        return out[0], out[1]

    def index_sample(self, comp, nind, map_n, nsample, ml_mode, seed, stream, want_counts=True):
        acc = C.c_int64(0)
        self._chk(self.lib.dangx_index_sample(self.h, comp, nind, map_n, nsample, L.ML_CODES[ml_mode], seed, stream,
                                              C.byref(acc) if want_counts else None))
        return acc.value

    def index_sample_pair(self, comp, nind, map_n, nsample, ml_mode, seed, stream_first, stream_second, want_counts=True):
        """index_sample(comp, nind, ...) followed by index_sample(comp, nind + 1, ...) on the same planes -- one kernel
        launch where the register chain covers both indices; bit for bit the two calls.  Returns both accepted counts."""
        a1, a2 = C.c_int64(0), C.c_int64(0)
        self._chk(self.lib.dangx_index_sample_pair(self.h, comp, nind, map_n, nsample, L.ML_CODES[ml_mode], seed,
                                                   stream_first, stream_second,
                                                   C.byref(a1) if want_counts else None, C.byref(a2) if want_counts else None))
        return a1.value, a2.value

    def amp_index_sample(self, group, flag, ml_mode, seed_amp, stream_amp, comp, nind, map_n, nsample, seed_index,
                         stream_index, solver="direct", fluct_mode="reference", want_counts=True, i_max=100, converge=1e-8):
        """amp_sample(group, flag, ...) followed by index_sample(comp, nind, map_n, ...) on the same planes -- one kernel
        launch when the model allows it, the two calls otherwise; bit for bit the same result either way.
        Returns (units whose block was not positive definite, accepted proposals)."""
        bad, acc = C.c_int64(0), C.c_int64(0)
        self._chk(self.lib.dangx_amp_index_sample(
            self.h, group, flag, L.ML_CODES[ml_mode], L.SOLVER_CG if solver == "cg" else L.SOLVER_DIRECT,
            L.FLUCT_REFERENCE if fluct_mode == "reference" else L.FLUCT_CORRECT, seed_amp, stream_amp, i_max, converge,
            comp, nind, map_n, nsample, seed_index, stream_index,
            None, C.byref(bad) if want_counts else None, C.byref(acc) if want_counts else None))
        return bad.value, acc.value

    def plane_set_sample(self, group, flag, ml_mode, seed_amp, stream_amp, sweeps, nsample, seed_index, solver="direct",
                         fluct_mode="reference", want_counts=True, i_max=100, converge=1e-8):
        """amp_sample(group, flag, ...) followed by index_sample(comp, nind, map_n of the flag, ...) for every (comp, nind, stream)
        of `sweeps` (the sweeps on the group's planes, in the reference's order) -- dangx_plane_set_sample: ONE launch for models
        with many bands and members (the members' SED columns stay in LDS across the sweeps), otherwise those calls through the
        two-step fusions.  Returns (units whose block was not positive definite, [accepted proposals per sweep])."""
        n = len(sweeps)
        comp = np.ascontiguousarray([s[0] for s in sweeps], dtype=np.int32)
        nind = np.ascontiguousarray([s[1] for s in sweeps], dtype=np.int32)
        strm = np.ascontiguousarray([s[2] for s in sweeps], dtype=np.uint64)
        bad, acc = C.c_int64(0), np.zeros(n, dtype=np.int64)
        self._chk(self.lib.dangx_plane_set_sample(
            self.h, group, flag, L.ML_CODES[ml_mode], L.SOLVER_CG if solver == "cg" else L.SOLVER_DIRECT,
            L.FLUCT_REFERENCE if fluct_mode == "reference" else L.FLUCT_CORRECT, seed_amp, stream_amp, i_max, converge,
            n, comp.ctypes.data, nind.ctypes.data, strm.ctypes.data, nsample, seed_index, None,
            C.byref(bad) if want_counts else None, acc.ctypes.data if want_counts else None))
        return bad.value, [int(a) for a in acc]

    def plane_sweeps_sample(self, flag, sweeps, nsample, ml_mode, seed, want_counts=True):
        """index_sample(comp, nind, map_n of the flag, ...) for every (comp, nind, stream) of `sweeps` -- the passes of
        sample_spectral_parameters on ONE plane set, in the reference's order (dangx_plane_sweeps_sample: one launch on the
        amplitudes in memory where the plane-set kernel covers the model, those calls otherwise).  Returns accepted counts."""
        n = len(sweeps)
        comp = np.ascontiguousarray([s[0] for s in sweeps], dtype=np.int32)
        nind = np.ascontiguousarray([s[1] for s in sweeps], dtype=np.int32)
        strm = np.ascontiguousarray([s[2] for s in sweeps], dtype=np.uint64)
        acc = np.zeros(n, dtype=np.int64)
        self._chk(self.lib.dangx_plane_sweeps_sample(self.h, flag, n, comp.ctypes.data, nind.ctypes.data, strm.ctypes.data, nsample,
                                                     L.ML_CODES[ml_mode], seed, acc.ctypes.data if want_counts else None))
        return [int(a) for a in acc]

    def sky_model_chisq(self, pol_lo, pol_hi, want_maps=False):
        s = C.c_double(0.0)
        if want_maps:
            sky = np.empty((self.nbands, self.nmaps, self.npix))
            res = np.empty_like(sky)
            chi = np.empty((self.nmaps, self.npix))
            self._chk(self.lib.dangx_sky_model_chisq(self.h, pol_lo, pol_hi, C.byref(s), sky.ctypes.data,
                                                     res.ctypes.data, chi.ctypes.data))
            return s.value, sky, res, chi
        self._chk(self.lib.dangx_sky_model_chisq(self.h, pol_lo, pol_hi, C.byref(s), None, None, None))
        return s.value

    def chisq_cached(self, which, pol_lo, pol_hi):
        """chi^2 sum fused into the index sweeps (which: 0 = state left by the amplitude phase, 1 = current);
        returns None when a plane was not covered by a sweep (caller falls back to sky_model_chisq)."""
        s = C.c_double(0.0)
        rc = self.lib.dangx_chisq_cached(self.h, which, pol_lo, pol_hi, C.byref(s))
        if rc == 2:
            return None
        self._chk(rc)
        return s.value

    def chisq_current(self, pol_lo, pol_hi):
        """the local chi^2 sum of the current state: cached sums where a sweep wrote the plane last, one pass over the others"""
        s = C.c_double(0.0)
        self._chk(self.lib.dangx_chisq_current(self.h, pol_lo, pol_hi, C.byref(s)))
        return s.value

    def index_masked_sums(self, entries):
        """[(comp, nind, map_n), ...] (at most 16) -> (sums, counts) of mask_avg's numerator and denominator, one launch"""
        n = len(entries)
        i32 = lambda k: np.ascontiguousarray([e[k] for e in entries], dtype=np.int32)
        c, j, m = i32(0), i32(1), i32(2)
        sums, counts = np.zeros(n), np.zeros(n, dtype=np.int64)
        self._chk(self.lib.dangx_index_masked_sums(self.h, n, c.ctypes.data, j.ctypes.data, m.ctypes.data, sums.ctypes.data, counts.ctypes.data))
        return sums, counts

    def chisq_cached_dev(self, which, pol_lo, pol_hi, out_tensor):
        rc = self.lib.dangx_chisq_cached_dev(self.h, which, pol_lo, pol_hi, out_tensor.data_ptr())
        if rc == 2:
            return False
        self._chk(rc)
        return True

    def sky_model_chisq_dev(self, pol_lo, pol_hi, out_tensor):
        """Asynchronous local chi^2 sum into a 1-element cuda fp64 tensor (for the RCCL all-reduce)."""
        self._chk(self.lib.dangx_sky_model_chisq_dev(self.h, pol_lo, pol_hi, out_tensor.data_ptr()))

    # -- full-sky index mode / gain fit primitives (local sums; the caller all-reduces)
    def fullsky_prepare(self, comp, map_n):
        self._chk(self.lib.dangx_fullsky_prepare(self.h, comp, map_n))

    def fullsky_prepare_coarse(self, comp, map_n, sample_nside):
        self._chk(self.lib.dangx_fullsky_prepare_coarse(self.h, comp, map_n, self.nside, int(sample_nside)))

    def fullsky_sums(self, what, theta, nrows):
        th = (C.c_double * 2)(float(theta[0]), float(theta[1]) if len(theta) > 1 else 0.0)
        out = (C.c_double * nrows)()
        self._chk(self.lib.dangx_fullsky_sums(self.h, what, th, out, nrows))
        return np.array(out[:nrows])

    def fill_index(self, comp, nind, map_n, value):
        self._chk(self.lib.dangx_fill_index(self.h, comp, nind, map_n, float(value)))

    def index_sample_coarse(self, comp, nind, map_n, nsample, ml_mode, seed, stream, sample_nside, want_counts=True):
        """sample_index_mh with sample_nside < nside (one whole-sky context); see dangx_index_sample_coarse."""
        acc = C.c_int64(0)
        self._chk(self.lib.dangx_index_sample_coarse(self.h, comp, nind, map_n, nsample, L.ML_CODES[ml_mode], seed, stream,
                                                     self.nside, int(sample_nside), C.byref(acc) if want_counts else None))
        return acc.value

    # the three phases of a coarse-Nside sweep on a pixel shard (one process driving several contexts adds the buffers)
    def coarse_sizes(self, map_n, sample_nside):
        a, b = C.c_int64(0), C.c_int64(0)
        self._chk(self.lib.dangx_coarse_sizes(self.h, map_n, int(sample_nside), C.byref(a), C.byref(b)))
        return a.value, b.value

    def coarse_partials(self, comp, map_n, sample_nside):
        n, _ = self.coarse_sizes(map_n, sample_nside)
        buf = np.empty(n)
        self._chk(self.lib.dangx_coarse_partials(self.h, comp, map_n, self.nside, int(sample_nside), buf.ctypes.data))
        return buf

    def coarse_chains(self, comp, nind, map_n, nsample, ml_mode, seed, stream, sample_nside, partials_sum):
        _, n = self.coarse_sizes(map_n, sample_nside)
        p = np.ascontiguousarray(partials_sum, dtype=np.float64)
        out = np.empty(n)
        self._chk(self.lib.dangx_coarse_chains(self.h, comp, nind, map_n, nsample, L.ML_CODES[ml_mode], seed, stream, self.nside,
                                               int(sample_nside), p.ctypes.data, out.ctypes.data))
        return out

    def coarse_writeback(self, comp, nind, map_n, sample_nside, index_sum):
        x = np.ascontiguousarray(index_sum, dtype=np.float64)
        self._chk(self.lib.dangx_coarse_writeback(self.h, comp, nind, map_n, self.nside, int(sample_nside), x.ctypes.data))

    def udgrade(self, mode, m, nside_in, nside_out):
        """udgrade_ring (0) / udgrade_rms (1) / udgrade_mask (2) of one RING map on the device."""
        m = np.ascontiguousarray(m, dtype=np.float64)
        out = np.empty(12 * nside_out * nside_out)
        self._chk(self.lib.dangx_udgrade(self.h, mode, m.ctypes.data, nside_in, out.ctypes.data, nside_out))
        return out

    @property
    def nside(self):
        n = int(round(math.sqrt(self.npix_global / 12.0)))
        if 12 * n * n != self.npix_global:
            raise DangxError("npix_global is not 12*nside^2")
        return n

    def index_masked_sum(self, comp, nind, map_n):
        s, n = C.c_double(0.0), C.c_int64(0)
        self._chk(self.lib.dangx_index_masked_sum(self.h, comp, nind, map_n, C.byref(s), C.byref(n)))
        return s.value, n.value

    def index_plain_sum(self, comp, nind, map_n):
        """(sum over every local pixel of c%indices(:,map_n,nind), sum of masks(:,1)): the per-pixel tuner's start."""
        s, m = C.c_double(0.0), C.c_double(0.0)
        self._chk(self.lib.dangx_index_plain_sum(self.h, comp, nind, map_n, C.byref(s), C.byref(m)))
        return s.value, m.value

    def peek_indices(self, comp, map_n, pix=0):
        n = self.component_list[comp].nindices
        out = (C.c_double * max(n, 1))()
        self._chk(self.lib.dangx_peek_indices(self.h, comp, map_n, pix, out))
        return [out[q] for q in range(n)]

    def gain_sums(self, band):
        out = (C.c_double * 2)()
        self._chk(self.lib.dangx_gain_sums(self.h, band, out))
        return out[0], out[1]

    def unit_conversion(self, band, which):
        """a2t / a2f / f2t of a band (src/dang_bp_mod.f90:181-274); which in 'a2t', 'a2f', 'f2t'."""
        out = C.c_double(0.0)
        self._chk(self.lib.dangx_unit_conversion(self.h, band, {"a2t": L.A2T, "a2f": L.A2F, "f2t": L.F2T}[which], C.byref(out)))
        return out.value

    def convert_maps(self, units, cg_map=None):
        """convert_maps (src/dang_data_mod.f90:429-463) on the resident maps; returns ddata%conversion."""
        u = np.ascontiguousarray([L.UNIT_CODES[x] if x in L.UNIT_CODES else 99 for x in units], dtype=np.int32)
        cg = None if cg_map is None else np.ascontiguousarray(np.asarray(cg_map, dtype=bool).astype(np.int32))
        conv = np.ones(self.nbands)
        self._chk(self.lib.dangx_convert_maps(self.h, u.ctypes.data, None if cg is None else cg.ctypes.data, conv.ctypes.data))
        return conv

    def set_tcmb(self, T):
        """The global T_CMB (src/dang_util_mod.f90:15): enters a2t of the 'cmb' component."""
        self._chk(self.lib.dangx_set_tcmb(self.h, float(T)))

    def set_calibration(self, gain, offset):
        g = np.ascontiguousarray(gain, dtype=np.float64)
        o = np.ascontiguousarray(offset, dtype=np.float64)
        self._chk(self.lib.dangx_set_calibration(self.h, g.ctypes.data, o.ctypes.data))

    # -- secondary seams
    def group_size(self, group, flag):
        n = self.lib.dangx_group_size(self.h, group, flag)
        if n < 0:
            raise DangxError(self.lib.dangx_last_error(self.h).decode())
        return n

    def compute_rhs(self, group, flag):
        b = np.empty(self.group_size(group, flag))
        self._chk(self.lib.dangx_compute_rhs(self.h, group, flag, b.ctypes.data))
        return b

    def compute_Ax(self, group, flag, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        res = np.empty_like(x)
        assert x.size == self.group_size(group, flag)
        self._chk(self.lib.dangx_compute_Ax(self.h, group, flag, x.ctypes.data, res.ctypes.data))
        return res

    def compute_sample_vector(self, group, flag, eta):
        eta = np.ascontiguousarray(eta, dtype=np.float64)
        res = np.empty(self.group_size(group, flag))
        assert eta.size == (2 if flag == L.FLAG_QU else 1) * self.npix
        self._chk(self.lib.dangx_compute_sample_vector(self.h, group, flag, eta.ctypes.data, res.ctypes.data))
        return res

    def eval_sed(self, comp, band, map_n):
        out = np.empty(self.npix)
        self._chk(self.lib.dangx_eval_sed(self.h, comp, band, map_n, out.ctypes.data))
        return out

    def rtc_kernels(self):
        """template-ids of the kernels this context obtained by run-time specialisation (csrc/dangx_rtc.hip)"""
        n, buf = C.c_int(0), C.create_string_buffer(16384)
        self._chk(self.lib.dangx_rtc_kernels(self.h, C.byref(n), buf, len(buf)))
        names = [x for x in buf.value.decode().split("\n") if x]
        assert len(names) == n.value or len(buf.value) >= len(buf) - 1
        return names

    # -- profiling
    def profile(self, on=True):
        self._chk(self.lib.dangx_profile_enable(self.h, 1 if on else 0))
        self._chk(self.lib.dangx_profile_reset(self.h))

    def profile_get(self, by_planes=False):
        """{kernel family: total_ms, launches, avg_ms}; by_planes: keys (family, planes) for the launches whose plane count the
        library records (1 = T / Q / U alone, 2 = Q+U): the T and Q+U instances of a kernel are different code objects."""
        out = {}
        for kid, name in L.KERNEL_NAMES.items():
            for pl in ((1, 2) if by_planes else (0,)):
                ms, n = C.c_double(0.0), C.c_int64(0)
                if pl:
                    self._chk(self.lib.dangx_profile_get_planes(self.h, kid, pl, C.byref(ms), C.byref(n)))
                else:
                    self._chk(self.lib.dangx_profile_get(self.h, kid, C.byref(ms), C.byref(n)))
                if n.value:
                    out[(name, pl) if pl else name] = {"total_ms": ms.value, "launches": n.value, "avg_ms": ms.value / n.value}
        return out


def comp_desc(c: DangComps):
    d = L.CompDesc()
    if c.type not in L.TYPE_CODES:
        raise DangxError("Error - unrecognized component type '%s'" % c.type)
    d.type = L.TYPE_CODES[c.type]
    d.is_synch = 1 if c.label.strip() == "synch" else 0
    d.nindices = c.nindices
    d.cg_group = c.cg_group
    d.sample_amplitude = 1 if c.sample_amplitude else 0
    d.nu_ref = float(c.nu_ref)
    for q in range(c.nindices):
        d.lnl_type[q] = L.LNL_CODES[c.lnl_type[q]] if q < len(c.lnl_type) else L.LNL_CHISQ
        d.prior_type[q] = L.PRIOR_CODES[c.prior_type[q]] if q < len(c.prior_type) else L.PRIOR_UNIFORM
        gp = c.gauss_prior[q] if q < len(c.gauss_prior) else [0.0, 1.0]
        up = c.uni_prior[q] if q < len(c.uni_prior) else [-1e300, 1e300]
        d.gauss_prior[q][0], d.gauss_prior[q][1] = gp
        d.uni_prior[q][0], d.uni_prior[q][1] = up
        d.step_size[q] = c.step_size[q] if q < len(c.step_size) else 0.0
    return d


# --------------------------------------------------------------------------- the two entry points

def initialize(bands, component_list, ddata, **kw):
    """What the driver does once after src/dang.f90:73: hand the static state to the device."""
    ddata.engine = Engine(bands, component_list, ddata, **kw)
    return ddata.engine


def index_means(ddata, map_n):
    """The per-iteration numbers of write_data (src/dang_data_mod.f90:716-731): mask_avg(c%indices(:,map_n,j), masks)
    for every sampled index, from device reductions (no map leaves the GPU).  {(label, ind_label): mean}"""
    eng = ddata.engine
    out = {}
    for l, c in enumerate(eng.component_list):
        for j in range(c.nindices):
            if c.sample_index[j]:
                s, n = eng.index_masked_sum(l, j, map_n)
                s, n = _dist.allreduce_sum_float(s), _dist.allreduce_sum_float(float(n))
                out[(c.label, c.ind_label[j] if c.ind_label else str(j))] = s / n if n else float("nan")
    return out


def mask_hi_threshold(ddata, c, thresh):
    """ddata%mask_hi_threshold(dpar), src/dang_data_mod.f90:398-427 -- what the driver does once BEFORE
    `initialize` when a 'hi_fit' component is present (src/dang.f90:74): pixels whose HI template exceeds the
    threshold are masked, missing / zero-rms pixels too, and the template is normalised by the threshold."""
    m = np.asarray(ddata.masks)
    t = np.asarray(c.template)
    m0 = m[0]
    bad = (t[0] > thresh) | (m0 == MISSVAL) | (m0 == 0.0) | (np.asarray(ddata.rms_map)[0, 0] == 0.0)
    m[0] = np.where(bad, 0.0, 1.0)
    c.template = t / thresh
    return ddata.masks


def normalize_bandpass(tau_in):
    """normalize_bandpass, src/dang_bp_mod.f90:62-81."""
    t = np.ascontiguousarray(tau_in, dtype=np.float64)
    out = np.empty_like(t)
    if L.load().dangx_normalize_bandpass(t.ctypes.data, t.size, out.ctypes.data):
        raise DangxError("normalize_bandpass: empty bandpass")
    return out


def convert_maps(ddata, units, cg_map=None):
    """ddata%convert_maps (src/dang_data_mod.f90:429-463) with the maps already resident: sets ddata.conversion, scales
    the device copies of sig_map / rms_map and the offsets; the host copy of ddata.offset follows."""
    eng = ddata.engine
    ddata.conversion = eng.convert_maps(units, cg_map)
    off = np.zeros(eng.nbands) if ddata.offset is None else np.asarray(ddata.offset, dtype=np.float64)
    skip = np.zeros(eng.nbands, dtype=bool) if cg_map is None else np.asarray(cg_map, dtype=bool)
    ddata.offset = np.where(skip, off, off * ddata.conversion)
    return ddata.conversion


def refresh_host_state(ddata):
    """What the output side of the Gibbs loop reads from the host (write_maps, src/dang_data_mod.f90:573-664): the
    components' amplitude / index maps and template amplitudes, ddata.sky_model / res_map / chi_map / chisq, and the
    band offsets a fitted monopole sets.  Call at the map-output cadence; nothing else moves maps off the device."""
    eng = ddata.engine
    eng.pull_state()
    for c in eng.component_list:
        if c.type == "monopole":
            ddata.offset = np.array(c.template_amplitudes[0], dtype=np.float64)
    s, sky, res, chi = eng.sky_model_chisq(ddata.pol_type[0], ddata.pol_type[-1], want_maps=True)
    ddata.sky_model, ddata.res_map, ddata.chi_map = sky, res, chi
    ddata.chisq = _dist.allreduce_sum_float(s) / eng.nbands / ddata.nump
    return ddata


def index_sample_coarse_multi(engines, comp, nind, map_n, nsample, ml_mode, seed, stream, sample_nside):
    """sample_index_mh with sample_nside < nside over several pixel-shard contexts of ONE process (e.g. one per GPU):
    the three phases of dangx_index_sample_coarse with the shards' buffers added in shard order between them.
    Returns the number of accepted proposals."""
    part = None
    for e in engines:
        b = e.coarse_partials(comp, map_n, sample_nside)
        part = b if part is None else part + b
    idx = None
    for e in engines:
        b = e.coarse_chains(comp, nind, map_n, nsample, ml_mode, seed, stream, sample_nside, part)
        idx = b if idx is None else idx + b
    for e in engines:
        e.coarse_writeback(comp, nind, map_n, sample_nside, idx)
    return int(idx[-1])


def sky_amp_sample(engines, group, flag, ml_mode, seed, stream, solver="direct", fluct_mode="reference", i_max=100, converge=1e-8):
    """One (group, flag) pass of sample_cg_groups over several pixel-shard contexts of ONE process (dangx_sky_amp_sample): a
    group with template / monopole / hi_fit members shares its Schur rows over the contexts.  Returns (cg_iters, n_not_spd)."""
    arr, n = _ctx_array(engines)
    it, bad = C.c_int(0), C.c_int64(0)
    engines[0]._chk(engines[0].lib.dangx_sky_amp_sample(
        arr, n, group, flag, L.ML_CODES[ml_mode], L.SOLVER_CG if solver == "cg" else L.SOLVER_DIRECT,
        L.FLUCT_REFERENCE if fluct_mode == "reference" else L.FLUCT_CORRECT, seed, stream, i_max, converge, C.byref(it), C.byref(bad)))
    return it.value, bad.value


def sky_plane_set_sample(engines, group, flag, ml_mode, seed_amp, stream_amp, sweeps, nsample, seed_index, solver="direct",
                         fluct_mode="reference", i_max=100, converge=1e-8, want_counts=True):
    """Engine.plane_set_sample over several pixel-shard contexts of ONE process (dangx_sky_plane_set_sample): a group with template
    members shares its Schur rows over the contexts, and where the plane-set kernel covers the model its back-substitution runs in
    the launch that does the sweeps.  Returns (cg_iters, n_not_spd, [accepted proposals per sweep])."""
    arr, n = _ctx_array(engines)
    ns = len(sweeps)
    comp = np.ascontiguousarray([s[0] for s in sweeps], dtype=np.int32)
    nind = np.ascontiguousarray([s[1] for s in sweeps], dtype=np.int32)
    strm = np.ascontiguousarray([s[2] for s in sweeps], dtype=np.uint64)
    it, bad, acc = C.c_int(0), C.c_int64(0), np.zeros(ns, dtype=np.int64)
    engines[0]._chk(engines[0].lib.dangx_sky_plane_set_sample(
        arr, n, group, flag, L.ML_CODES[ml_mode], L.SOLVER_CG if solver == "cg" else L.SOLVER_DIRECT,
        L.FLUCT_REFERENCE if fluct_mode == "reference" else L.FLUCT_CORRECT, seed_amp, stream_amp, i_max, converge,
        ns, comp.ctypes.data, nind.ctypes.data, strm.ctypes.data, nsample, seed_index, C.byref(it),
        C.byref(bad) if want_counts else None, acc.ctypes.data if want_counts else None))
    return it.value, bad.value, [int(a) for a in acc]


def compute_chisq(ddata):
    """update_sky_model + compute_chisq (src/dang_data_mod.f90:339-396, 494-526); all-reduced over shards."""
    eng = ddata.engine
    # the sums the last solve / sweep on each plane left behind (by-products of those launches), one pass over the planes that
    # have none (dangx_chisq_current); Engine.sky_model_chisq is the explicit pass with the map outputs
    s = eng.chisq_current(ddata.pol_type[0], ddata.pol_type[-1])
    s = _dist.allreduce_sum_float(s)
    ddata.chisq = s / eng.nbands / ddata.nump
    return ddata.chisq


def _plain_sweep(eng, c, j):
    """index j of component c is an ordinary per-pixel sweep at the map resolution with nothing to tune first"""
    coarse = c.sample_nside[j] if c.sample_nside else 0
    return (c.sample_index[j] and not (c.index_mode and c.index_mode[j] == 1) and not (coarse and coarse != eng.nside)
            and not (c.tuned and not c.tuned[j]))


def fusable_first_sweeps(dpar, eng):
    """{(cg_group, flag): (component, index)} -- the solves that may be issued TOGETHER with the first index sweep on their
    planes without changing the result of the main loop (the reference runs every solve of sample_cg_groups before any sweep
    of sample_spectral_parameters).  The rule lives behind the ABI, once for every host: dangx_plan_fusion
    (dang_amd/csrc/dangx_sky.hip); here the two lists are put together in the reference's loop orders."""
    comps = eng.component_list
    pairs = [(g.cg_group, f) for g in dpar.cg_groups if g.sample for f in g.pol_flag]
    sweeps = [(l, j, f, 1 if _plain_sweep(eng, c, j) else 0) for l, c in enumerate(comps) for j in range(c.nindices)
              if c.sample_index[j] for f in c.pol_flag[j]]
    if not pairs or not sweeps:
        return {}
    i32 = lambda v: np.ascontiguousarray(v, dtype=np.int32)
    pg, pf = i32([p[0] for p in pairs]), i32([p[1] for p in pairs])
    sc, sn, sf, sp = (i32([s[q] for s in sweeps]) for q in range(4))
    first = np.full(len(pairs), -1, dtype=np.int32)
    eng._chk(eng.lib.dangx_plan_fusion(eng.h, len(pairs), pg.ctypes.data, pf.ctypes.data, len(sweeps), sc.ctypes.data, sn.ctypes.data,
                                       sf.ctypes.data, sp.ctypes.data, L.SOLVER_CG if dpar.solver == "cg" else L.SOLVER_DIRECT,
                                       first.ctypes.data))
    return {pairs[p]: sweeps[e][:2] for p, e in enumerate(first) if e >= 0}


def _planes(flag):
    return {L.FLAG_T: 1, L.FLAG_Q: 2, L.FLAG_U: 4, L.FLAG_QU: 6}.get(flag, 7)


def plan_plane_sets(dpar, eng):
    """{(cg_group, flag): [(component, index), ...]} -- for every solve that may be issued together with sweeps
    (fusable_first_sweeps), the sweeps that go with it through ONE entry point (Engine.plane_set_sample): every sampled index
    on the group's planes in the reference's order when all of them carry this flag and no sampled index of ANOTHER flag touches
    one of the planes (a Q sweep beside a Q+U group would otherwise be reordered against src/dang_sample_mod.f90:40-75);
    else only the first sweep.  The one place this rule lives for the Python hosts (sample_cg_groups, bench.py)."""
    comps = eng.component_list
    out = {}
    for (grp, f), first in fusable_first_sweeps(dpar, eng).items():
        same = [(l, j) for l, c in enumerate(comps) for j in range(c.nindices) if c.sample_index[j] and f in c.pol_flag[j]]
        foreign = any(c.sample_index[j] and f2 != f and (_planes(f2) & _planes(f))
                      for c in comps for j in range(c.nindices) for f2 in c.pol_flag[j])
        if foreign or not same or same[0] != tuple(first):
            same = [tuple(first)]        # another flag shares a plane: only the first sweep goes with the solve
        out[(grp, f)] = same
    return out


def sample_cg_groups(dpar: DangParams, ddata: DangData, it=1, verbose=False, defer_chisq=False, fuse_first=None, it_index=None,
                     want_counts=True):
    """sample_cg_groups(dpar, ddata), src/dang_cg_mod.f90:142-177.

    defer_chisq=True skips the chi^2 pass after the amplitude phase; sample_spectral_parameters then
    reports it (ddata.chisq_after_amp) from the value its first sweeps compute as a by-product.
    fuse_first: a set; a group's solve is then issued together with the first index sweep that
    sample_spectral_parameters (iteration it_index) would run on the same planes -- where fusable_first_sweeps says the
    main loop's result does not change (Engine.amp_index_sample: one kernel launch where the model allows it, bit for bit
    the two calls) -- and (component, index, flag) of that sweep is added to the set: pass it to
    sample_spectral_parameters(skip=...).  See gibbs_iteration."""
    eng = ddata.engine
    info = []
    plane_sets = plan_plane_sets(dpar, eng) if fuse_first is not None else {}
    for g in dpar.cg_groups:
        if not g.sample:
            continue
        has_global = any(c.cg_group == g.cg_group and c.type in ("template", "monopole", "hi_fit") for c in eng.component_list)
        for f in g.pol_flag:
            same = plane_sets.get((g.cg_group, f))
            if same is not None:
                # the solve with EVERY sweep on its planes, in the reference's order (all of them are plain per-pixel sweeps and
                # nothing else touches these planes, so pulling them forward leaves the loop's result unchanged): one entry
                # point -- one launch for many-band, many-member models, the two-step fusions otherwise
                iti = it if it_index is None else it_index
                bad, accs = eng.plane_set_sample(g.cg_group, f, dpar.ml_mode, dpar.seed, stream_id(it, 0, g.cg_group, 0, f),
                                                 [(l, j, stream_id(iti, 1, l, j, f)) for l, j in same], dpar.nsample, dpar.seed,
                                                 solver=dpar.solver, fluct_mode=dpar.fluct_mode, i_max=g.i_max, converge=g.converge,
                                                 want_counts=want_counts)
                for l, j in same:
                    fuse_first.add((l, j, f))
                info.append((g.cg_group, f, 0, bad))
                continue
            cg_it, bad = eng.amp_sample(g.cg_group, f, dpar.ml_mode, dpar.seed, stream_id(it, 0, g.cg_group, 0, f),
                                        solver=dpar.solver, fluct_mode=dpar.fluct_mode,
                                        i_max=g.i_max, converge=g.converge, want_counts=want_counts or dpar.solver == "cg")
            info.append((g.cg_group, f, cg_it, bad))
        if has_global:  # update_sky_model: self%offset = c%template_amplitudes(:,1) of a monopole (src/dang_data_mod.f90:357-361)
            for l, c in enumerate(eng.component_list):
                if c.type == "monopole" and c.cg_group == g.cg_group:
                    ddata.offset = eng.get_template_amplitudes(l)[0].copy()
        if not defer_chisq:
            compute_chisq(ddata)  # update_sky_model + write_stats_to_term, :172-173
            if verbose:
                print("%6d - Chisq: %16.5E" % (it, ddata.chisq))
    return info


_MAPN = {L.FLAG_T: 1, L.FLAG_Q: 2, L.FLAG_U: 3, L.FLAG_QU: -1}  # src/dang_sample_mod.f90:53-64


def sample_spectral_parameters(dpar: DangParams, ddata: DangData, it=2, verbose=False, skip=(), want_counts=True):
    """sample_spectral_parameters(dpar, ddata), src/dang_sample_mod.f90:21-86 (per-pixel index_mode).

    Consecutive plain per-pixel sweeps of one plane set go through Engine.plane_sweeps_sample (one launch where the model
    allows it; inside it two consecutive indices of one component -- dust beta, dust T -- travel together).  skip: (component,
    index, flag) sweeps already done together with their group's solve (sample_cg_groups(fuse_first=...))."""
    eng = ddata.engine
    sampled = False
    info = []
    comps = eng.component_list
    # the sweeps of the iteration as a flat list in the loop's order (component, index, flag): consecutive plain per-pixel sweeps
    # with ONE flag whose components belong to one CG group go through Engine.plane_sweeps_sample -- one launch on the plane set
    # where the model allows it (sweeps on disjoint planes are independent, so collecting a plane set's sweeps changes nothing)
    entries = [(l, j, f) for l, c in enumerate(comps) for j in range(c.nindices) if c.sample_index[j] for f in c.pol_flag[j]]
    plain = [(f in _MAPN and _plain_sweep(eng, comps[l], j) and comps[l].sample_amplitude and (l, j, f) not in skip) for l, j, f in entries]
    done = set()
    for e, (l, j, f) in enumerate(entries):
        c = comps[l]
        sampled = True
        if (l, j, f) in skip or e in done:
            continue
        if plain[e]:
            n = 1
            while e + n < len(entries) and plain[e + n] and entries[e + n][2] == f and comps[entries[e + n][0]].cg_group == c.cg_group:
                n += 1
            if n >= 2:
                run = entries[e:e + n]
                accs = eng.plane_sweeps_sample(f, [(l2, j2, stream_id(it, 1, l2, j2, f)) for l2, j2, _ in run], dpar.nsample, dpar.ml_mode, dpar.seed,
                                               want_counts=want_counts)
                info += [(l2, j2, f, a) for (l2, j2, _), a in zip(run, accs)]
                done.update(range(e + 1, e + n))
                continue
        if f not in _MAPN:
            raise DangxError("There is something wrong with the poltype flag for component " + c.label)
        coarse = c.sample_nside[j] if c.sample_nside else 0
        if c.index_mode and c.index_mode[j] == 1:
            acc = sample_index_mh_fullsky(dpar, ddata, l, j, _MAPN[f], stream_id(it, 1, l, j, f),
                                          sample_nside=coarse if (coarse and coarse != eng.nside) else None)
        elif coarse and coarse != eng.nside:
            if c.tuned and not c.tuned[j]:
                raise DangxError("step-size tuning with sample_nside /= nside is not built")
            acc = eng.index_sample_coarse(l, j, _MAPN[f], dpar.nsample, dpar.ml_mode, dpar.seed,
                                          stream_id(it, 1, l, j, f), coarse)
        else:
            if c.tuned and not c.tuned[j]:       # 'Tuning!', src/dang_sample_mod.f90:341-346
                tune_perpixel(dpar, ddata, l, j, _MAPN[f], stream_id(it, 1, l, j, f))
            acc = eng.index_sample(l, j, _MAPN[f], dpar.nsample, dpar.ml_mode, dpar.seed,
                                   stream_id(it, 1, l, j, f))
        info.append((l, j, f, acc))
        # "Update the global variable T_CMB": T_CMB = c%indices(0,1,1), :75-78 (it enters a2t of 'cmb'), after the component's last sweep
        if c.type == "T_cmb" and not any(l2 == l for l2, _, _ in entries[e + 1:]):
            arr, n = _ctx_array([eng])
            eng._chk(eng.lib.dangx_update_tcmb(arr, n, l, None))
    if sampled:
        lo, hi = ddata.pol_type[0], ddata.pol_type[-1]
        before, after = eng.chisq_cached(0, lo, hi), eng.chisq_cached(1, lo, hi)
        if before is not None:
            ddata.chisq_after_amp = _dist.allreduce_sum_float(before) / eng.nbands / ddata.nump
        if after is not None:  # every plane was swept: chi^2 came for free
            ddata.chisq = _dist.allreduce_sum_float(after) / eng.nbands / ddata.nump
        else:
            compute_chisq(ddata)  # :81-84
        if verbose:
            print("%6d - Chisq: %16.5E" % (it, ddata.chisq))
    return info


def gibbs_iteration(dpar: DangParams, ddata: DangData, it, verbose=False, want_counts=True):
    """sample_cg_groups followed by sample_spectral_parameters (one pass of the main loop, src/dang.f90:87-126, for
    iterations in which both run) with every group's solve issued together with the first sweep on its planes: the same
    state, bit for bit, as the two calls with the same `it`, in fewer kernel launches.  chi^2 after the amplitude phase
    comes from the sweeps' by-product (ddata.chisq_after_amp).
    want_counts=False: the plane-set launches do not count accepted proposals / non-SPD blocks (the reference reports neither;
    reading them back makes the host wait for every launch and costs the kernel ~4 %) -- the info lists carry zeros there."""
    done = set()
    a = sample_cg_groups(dpar, ddata, it=it, verbose=False, defer_chisq=True, fuse_first=done, it_index=it, want_counts=want_counts)
    b = sample_spectral_parameters(dpar, ddata, it=it, verbose=verbose, skip=done, want_counts=want_counts)
    return a, b


# --------------------------------------------------------------------------- full-sky mode, tuner, calibrators
# The chains themselves live behind the C ABI (dang_amd/csrc/dangx_sky.hip: dangx_fullsky_sample, dangx_tune_perpixel,
# dangx_fit_band_gain, dangx_update_tcmb) -- one implementation for this mirror, the Fortran layers and any other host.
# What is left here is marshalling: the component's `tuned` flags and step size travel through in/out arguments.

def _ctx_array(engines):
    arr = (C.c_void_p * len(engines))(*[e.h.value for e in engines])
    return arr, len(engines)


def _engines_of(ddata, engines=None):
    return list(engines) if engines is not None else [ddata.engine]


def _tuned_array(c):
    n = max(c.nindices, 1)
    if not c.tuned:
        c.tuned = [True] * n
    return (C.c_int32 * n)(*[1 if t else 0 for t in c.tuned])


def tune_perpixel(dpar, ddata, l, nind, map_n, stream, engines=None):
    """The step-size tuning of the per-pixel branch, src/dang_sample_mod.f90:341-346 (dangx_tune_perpixel): updates
    c.step_size[nind] and c.tuned; the contexts' descriptors follow inside the call."""
    engs = _engines_of(ddata, engines)
    c = engs[0].component_list[l]
    arr, n = _ctx_array(engs)
    tuned, step = _tuned_array(c), C.c_double(0.0)
    engs[0]._chk(engs[0].lib.dangx_tune_perpixel(arr, n, l, nind, map_n, dpar.nsample, L.ML_CODES[dpar.ml_mode], dpar.seed, stream,
                                                 tuned, C.byref(step)))
    c.tuned = [bool(t) for t in tuned]
    c.step_size[nind] = step.value
    return step.value


def sample_index_mh_fullsky(dpar, ddata, l, nind, map_n, stream, sample_nside=None, engines=None):
    """sample_index_mh, index_mode == 1 (src/dang_sample_mod.f90:229-329): one spectral index for the whole sky, through
    dangx_fullsky_sample (tuner included).  sample_nside (/= nside): the chain's sums run over the degraded maps (:199-217).
    engines: the pixel-shard contexts of this process in shard order (default: ddata.engine).  Returns accepted proposals."""
    engs = _engines_of(ddata, engines)
    c = engs[0].component_list[l]
    arr, n = _ctx_array(engs)
    tuned, step, acc = _tuned_array(c), C.c_double(0.0), C.c_int64(0)
    engs[0]._chk(engs[0].lib.dangx_fullsky_sample(arr, n, l, nind, map_n, dpar.nsample, L.ML_CODES[dpar.ml_mode], dpar.seed, stream,
                                                  engs[0].nside if sample_nside else 0, int(sample_nside or 0), tuned,
                                                  C.byref(step), None, C.byref(acc)))
    c.tuned = [bool(t) for t in tuned]
    if nind < len(c.step_size):
        c.step_size[nind] = step.value
    return acc.value


def fit_band_gain(dpar, ddata, band, it=1, engines=None):
    """fit_band_gain(ddata, 1, band), src/dang_sample_mod.f90:570-621 (band 0-based here), through dangx_fit_band_gain."""
    engs = _engines_of(ddata, engines)
    arr, n = _ctx_array(engs)
    g = C.c_double(0.0)
    engs[0]._chk(engs[0].lib.dangx_fit_band_gain(arr, n, band, L.ML_CODES[dpar.ml_mode], dpar.seed, stream_id(it, 2, 0, 0, 0), C.byref(g)))
    ddata.gain[band] = g.value
    return g.value


def sample_calibrators(dpar, ddata, it=2, verbose=False):
    """sample_calibrators(ddata), src/dang_sample_mod.f90:487-518."""
    sampled = False
    for j, fit in enumerate(ddata.fit_gain or []):
        if fit:
            sampled = True
            fit_band_gain(dpar, ddata, j, it=it)
    if sampled:
        compute_chisq(ddata)
        if verbose:
            print("%6d - Chisq: %16.5E" % (it, ddata.chisq))
    return sampled
