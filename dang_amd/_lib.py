"""ctypes binding of libdangx.so (the C ABI in include/dangx.h).

There is NO CPU fallback: if the HIP library is missing or no GPU is visible the
product path raises.  The CPU oracle lives in oracle/ and is test infrastructure only.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DANGX_LIB selects another build of the same library (A/B timing of kernel variants on one GPU)
LIB_PATH = os.environ.get("DANGX_LIB") or os.path.join(_HERE, "lib", "libdangx.so")

MAX_BANDS, MAX_COMPS, MAX_IND, MAX_GROUP = 32, 16, 2, 8

# enums (include/dangx.h)
POWERLAW, MBB, FREEFREE, LOGNORMAL, CMB, TCMB, TEMPLATE, MONOPOLE, HIFIT = 1, 2, 3, 4, 5, 6, 7, 8, 9
LNL_CHISQ, LNL_MARGINAL, LNL_PRIOR = 1, 2, 3
PRIOR_GAUSSIAN, PRIOR_UNIFORM, PRIOR_JEFFREYS = 1, 2, 3
ML_SAMPLE, ML_OPTIMIZE = 1, 2
FLAG_T, FLAG_Q, FLAG_U, FLAG_QU = 1, 2, 4, 8
SOLVER_DIRECT, SOLVER_CG = 0, 1
FLUCT_CORRECT, FLUCT_REFERENCE = 0, 1
A2T, A2F, F2T = 0, 1, 2
UNIT_CODES = {"uK_RJ": 0, "uK_cmb": 1, "MJy/sr": 2}
K_AMP_DIRECT, K_INDEX_MH, K_SKY_CHISQ, K_REDUCE, K_CG_AX, K_CG_VEC, K_AMP_INDEX = range(7)
KERNEL_NAMES = {K_AMP_DIRECT: "k_amp_direct", K_INDEX_MH: "k_index_mh", K_SKY_CHISQ: "k_sky_chisq",
                K_REDUCE: "k_reduce", K_CG_AX: "k_Ax", K_CG_VEC: "k_cg_vec", K_AMP_INDEX: "k_amp_index"}

TYPE_CODES = {"power-law": POWERLAW, "mbb": MBB, "freefree": FREEFREE, "lognormal": LOGNORMAL, "cmb": CMB, "T_cmb": TCMB,
              "template": TEMPLATE, "monopole": MONOPOLE, "hi_fit": HIFIT}
LNL_CODES = {"chisq": LNL_CHISQ, "marginal": LNL_MARGINAL, "prior": LNL_PRIOR}
PRIOR_CODES = {"gaussian": PRIOR_GAUSSIAN, "uniform": PRIOR_UNIFORM, "jeffreys": PRIOR_JEFFREYS}
ML_CODES = {"sample": ML_SAMPLE, "optimize": ML_OPTIMIZE}


class Dims(C.Structure):
    _fields_ = [("npix", C.c_int32), ("nmaps", C.c_int32), ("nbands", C.c_int32), ("ncomp", C.c_int32),
                ("pix0", C.c_int64), ("npix_global", C.c_int64), ("device", C.c_int32), ("reserved", C.c_int32)]


class CompDesc(C.Structure):
    _fields_ = [("type", C.c_int32), ("is_synch", C.c_int32), ("nindices", C.c_int32), ("cg_group", C.c_int32),
                ("sample_amplitude", C.c_int32), ("reserved", C.c_int32), ("nu_ref", C.c_double),
                ("lnl_type", C.c_int32 * MAX_IND), ("prior_type", C.c_int32 * MAX_IND),
                ("gauss_prior", (C.c_double * 2) * MAX_IND), ("uni_prior", (C.c_double * 2) * MAX_IND),
                ("step_size", C.c_double * MAX_IND)]


# every symbol include/dangx.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_D = C.POINTER(C.c_double)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int64)  # dangx_allreduce_fn

SYMBOLS = {
    "dangx_create": (C.c_int, [C.POINTER(_P), C.POINTER(Dims)]),
    "dangx_destroy": (C.c_int, [_P]),
    "dangx_last_error": (C.c_char_p, [_P]),
    "dangx_version": (C.c_char_p, []),
    "dangx_set_stream": (C.c_int, [_P, _P]),
    "dangx_synchronize": (C.c_int, [_P]),
    "dangx_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "dangx_set_host_stride": (C.c_int, [_P, C.c_int64]),
    "dangx_set_allreduce": (C.c_int, [_P, _P, _P, C.c_int]),
    "dangx_index_sample_coarse": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64,
                                            C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    "dangx_coarse_sizes": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "dangx_coarse_partials": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "dangx_coarse_chains": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_int, _P, _P]),
    "dangx_coarse_writeback": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "dangx_udgrade": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, C.c_int]),
    "dangx_index_masked_sum": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "dangx_index_plain_sum": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "dangx_unit_conversion": (C.c_int, [_P, C.c_int, C.c_int, _D]),
    "dangx_normalize_bandpass": (C.c_int, [_P, C.c_int, _P]),
    "dangx_convert_maps": (C.c_int, [_P, _P, _P, _P]),
    "dangx_set_band": (C.c_int, [_P, C.c_int, C.c_double, C.c_int, _P, _P]),
    "dangx_set_component": (C.c_int, [_P, C.c_int, C.POINTER(CompDesc)]),
    "dangx_set_tcmb": (C.c_int, [_P, C.c_double]),
    "dangx_set_calibration": (C.c_int, [_P, _P, _P]),
    "dangx_upload_data": (C.c_int, [_P, _P, _P, _P]),
    "dangx_adopt_device_data": (C.c_int, [_P, _P, _P, _P]),
    "dangx_put_amplitude": (C.c_int, [_P, C.c_int, _P]),
    "dangx_get_amplitude": (C.c_int, [_P, C.c_int, _P]),
    "dangx_put_indices": (C.c_int, [_P, C.c_int, _P]),
    "dangx_get_indices": (C.c_int, [_P, C.c_int, _P]),
    "dangx_set_template": (C.c_int, [_P, C.c_int, _P, _P, C.c_int]),
    "dangx_put_template_amplitudes": (C.c_int, [_P, C.c_int, _P]),
    "dangx_get_template_amplitudes": (C.c_int, [_P, C.c_int, _P]),
    "dangx_adopt_device_state": (C.c_int, [_P, C.c_int, _P, _P]),
    "dangx_amplitude_devptr": (_P, [_P, C.c_int]),
    "dangx_indices_devptr": (_P, [_P, C.c_int]),
    "dangx_amp_sample": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64,
                                   C.c_int, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "dangx_schur_info": (C.c_int, [_P, _D, C.POINTER(C.c_int)]),
    "dangx_amp_residual": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, _D]),
    "dangx_index_sample": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64,
                                     C.POINTER(C.c_int64)]),
    "dangx_index_sample_pair": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64,
                                          C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "dangx_amp_index_sample": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64,
                                         C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64,
                                         C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "dangx_plane_set_sample": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_double,
                                         C.c_int, _P, _P, _P, C.c_int, C.c_uint64, C.POINTER(C.c_int), C.POINTER(C.c_int64), _P]),
    "dangx_plane_sweeps_sample": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, C.c_int, C.c_int, C.c_uint64, _P]),
    "dangx_sky_plane_set_sample": (C.c_int, [C.POINTER(_P), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_int,
                                             C.c_double, C.c_int, _P, _P, _P, C.c_int, C.c_uint64, C.POINTER(C.c_int), C.POINTER(C.c_int64), _P]),
    "dangx_sky_model_chisq": (C.c_int, [_P, C.c_int, C.c_int, _D, _P, _P, _P]),
    "dangx_sky_model_chisq_dev": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "dangx_chisq_cached": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _D]),
    "dangx_chisq_current": (C.c_int, [_P, C.c_int, C.c_int, _D]),
    "dangx_index_masked_sums": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P]),
    "dangx_chisq_cached_dev": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P]),
    "dangx_fullsky_prepare": (C.c_int, [_P, C.c_int, C.c_int]),
    "dangx_fullsky_prepare_coarse": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int]),
    "dangx_fullsky_sums": (C.c_int, [_P, C.c_int, _P, _P, C.c_int]),
    "dangx_fill_index": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_double]),
    "dangx_gain_sums": (C.c_int, [_P, C.c_int, _P]),
    "dangx_peek_indices": (C.c_int, [_P, C.c_int, C.c_int, C.c_longlong, _P]),
    "dangx_fullsky_finish_coarse": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    # the sky-wide steps, chain included: (ctxs, nctx, ...) = the contexts of this process in shard order
    "dangx_fullsky_sample": (C.c_int, [C.POINTER(_P), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64,
                                       C.c_int, C.c_int, C.POINTER(C.c_int32), _D, _D, C.POINTER(C.c_int64)]),
    "dangx_tune_step_size": (C.c_int, [C.POINTER(_P), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, _D,
                                       C.POINTER(C.c_uint32), C.POINTER(C.c_int32), _D]),
    "dangx_tune_perpixel": (C.c_int, [C.POINTER(_P), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64,
                                      C.POINTER(C.c_int32), _D]),
    "dangx_fit_band_gain": (C.c_int, [C.POINTER(_P), C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, _D]),
    "dangx_update_tcmb": (C.c_int, [C.POINTER(_P), C.c_int, C.c_int, _D]),
    "dangx_plan_fusion": (C.c_int, [_P, C.c_int, _P, _P, C.c_int, _P, _P, _P, _P, C.c_int, _P]),
    "dangx_sky_amp_sample": (C.c_int, [C.POINTER(_P), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64,
                                       C.c_int, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "dangx_group_size": (C.c_int64, [_P, C.c_int, C.c_int]),
    "dangx_compute_rhs": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "dangx_compute_Ax": (C.c_int, [_P, C.c_int, C.c_int, _P, _P]),
    "dangx_compute_sample_vector": (C.c_int, [_P, C.c_int, C.c_int, _P, _P]),
    "dangx_eval_sed": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P]),
    "dangx_rtc_kernels": (C.c_int, [_P, C.POINTER(C.c_int), C.c_char_p, C.c_int]),
    "dangx_rtc_compile": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]),
    "dangx_profile_enable": (C.c_int, [_P, C.c_int]),
    "dangx_profile_reset": (C.c_int, [_P]),
    "dangx_profile_get": (C.c_int, [_P, C.c_int, _D, C.POINTER(C.c_int64)]),
    "dangx_profile_get_planes": (C.c_int, [_P, C.c_int, C.c_int, _D, C.POINTER(C.c_int64)]),
}

_lib = None


def load():
    """Load libdangx.so and attach prototypes.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libdangx.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'`; "
            "there is no CPU fallback for the product path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
