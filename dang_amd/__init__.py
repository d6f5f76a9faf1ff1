"""dang_amd -- MI355X-native Gibbs inner loop for the `dang` component-separation sampler.

Only the hot path lives here: HIP kernels + C ABI (csrc/, lib/libdangx.so), the host-side
mirror of the reference's module API (api.py), pixel sharding / reductions (dist.py) and the
synthetic-sky generator used by tests and bench.py (synth.py).
"""
from . import _lib  # noqa: F401
from .api import (BandInfo, DangCGGroup, DangComps, DangData, DangParams, DangxError, Engine, mask_hi_threshold,  # noqa: F401
                  compute_chisq, convert_maps, fit_band_gain, gibbs_iteration, index_sample_coarse_multi, normalize_bandpass, refresh_host_state, index_means, initialize, return_poltype_flag, sample_calibrators, sample_cg_groups,
                  sample_index_mh_fullsky, sample_spectral_parameters, stream_id, tune_perpixel, fusable_first_sweeps, plan_plane_sets, sky_amp_sample, sky_plane_set_sample)

__version__ = "0.1.0"
