! dangx_mod.f90 -- ISO_C_BINDING interface to libdangx.so (include/dangx.h).
!
! This is the thin Fortran layer the north star asks for: the dang driver keeps its
! parameter files, HEALPix/CFITSIO I/O and dang_component_mod API and reaches the
! MI355X path through these bind(C) interfaces.  The wrapper with the reference's
! own signatures, sample_cg_groups_gpu(dpar, ddata) / sample_spectral_parameters_gpu
! (dpar, ddata), lives in fortran/reference_side/dang_gpu_mod.f90 (it `use`s the
! reference's modules and is compiled inside the reference's build).
!
! Array arguments are the reference's arrays passed as-is (c_loc of the first element):
!   sig_map/rms_map(0:npix-1,nmaps,nbands), masks(0:npix-1,nmaps),
!   c%amplitude(0:npix-1,nmaps), c%indices(0:npix-1,nmaps,nindices).
module dangx_mod
  use, intrinsic :: iso_c_binding
  implicit none

  integer(c_int), parameter :: DANGX_POWERLAW = 1, DANGX_MBB = 2, DANGX_FREEFREE = 3, &
       DANGX_LOGNORMAL = 4, DANGX_CMB = 5, DANGX_TCMB = 6, DANGX_TEMPLATE = 7, DANGX_MONOPOLE = 8, DANGX_HIFIT = 9
  integer(c_int), parameter :: DANGX_LNL_CHISQ = 1, DANGX_LNL_MARGINAL = 2, DANGX_LNL_PRIOR = 3
  integer(c_int), parameter :: DANGX_PRIOR_GAUSSIAN = 1, DANGX_PRIOR_UNIFORM = 2, DANGX_PRIOR_JEFFREYS = 3
  integer(c_int), parameter :: DANGX_ML_SAMPLE = 1, DANGX_ML_OPTIMIZE = 2
  integer(c_int), parameter :: DANGX_FLAG_T = 1, DANGX_FLAG_Q = 2, DANGX_FLAG_U = 4, DANGX_FLAG_QU = 8
  integer(c_int), parameter :: DANGX_SOLVER_DIRECT = 0, DANGX_SOLVER_CG = 1
  integer(c_int), parameter :: DANGX_FLUCT_CORRECT = 0, DANGX_FLUCT_REFERENCE = 1
  integer(c_int), parameter :: DANGX_A2T = 0, DANGX_A2F = 1, DANGX_F2T = 2
  integer(c_int), parameter :: DANGX_UNIT_UK_RJ = 0, DANGX_UNIT_UK_CMB = 1, DANGX_UNIT_MJY_SR = 2

  type, bind(C) :: dangx_dims
     integer(c_int32_t) :: npix, nmaps, nbands, ncomp
     integer(c_int64_t) :: pix0, npix_global
     integer(c_int32_t) :: device, reserved
  end type dangx_dims

  ! C: double gauss_prior[ind][2]  <->  Fortran gauss_prior(2, ind)  (mean/std fastest)
  type, bind(C) :: dangx_comp_desc
     integer(c_int32_t) :: type, is_synch, nindices, cg_group, sample_amplitude, reserved
     real(c_double)     :: nu_ref
     integer(c_int32_t) :: lnl_type(2), prior_type(2)
     real(c_double)     :: gauss_prior(2,2), uni_prior(2,2), step_size(2)
  end type dangx_comp_desc

  interface
     integer(c_int) function dangx_create(ctx, dims) bind(C, name='dangx_create')
       import :: c_int, c_ptr, dangx_dims
       type(c_ptr), intent(out) :: ctx
       type(dangx_dims), intent(in) :: dims
     end function
     integer(c_int) function dangx_destroy(ctx) bind(C, name='dangx_destroy')
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
     end function
     type(c_ptr) function dangx_last_error(ctx) bind(C, name='dangx_last_error')
       import :: c_ptr
       type(c_ptr), value :: ctx
     end function
     integer(c_int) function dangx_set_band(ctx, band, nu_c, n, nu0, tau0) bind(C, name='dangx_set_band')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: band, n
       real(c_double), value :: nu_c
       type(c_ptr), value :: nu0, tau0
     end function
     integer(c_int) function dangx_set_component(ctx, comp, desc) bind(C, name='dangx_set_component')
       import :: c_int, c_ptr, dangx_comp_desc
       type(c_ptr), value :: ctx
       integer(c_int), value :: comp
       type(dangx_comp_desc), intent(in) :: desc
     end function
     integer(c_int) function dangx_set_tcmb(ctx, T) bind(C, name='dangx_set_tcmb')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), value :: T
     end function
     integer(c_int) function dangx_set_calibration(ctx, gain, offset) bind(C, name='dangx_set_calibration')
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx, gain, offset
     end function
     integer(c_int) function dangx_upload_data(ctx, sig, rms, mask) bind(C, name='dangx_upload_data')
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx, sig, rms, mask
     end function
     integer(c_int) function dangx_put_amplitude(ctx, comp, amp) bind(C, name='dangx_put_amplitude')
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx, amp
       integer(c_int), value :: comp
     end function
     integer(c_int) function dangx_get_amplitude(ctx, comp, amp) bind(C, name='dangx_get_amplitude')
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx, amp
       integer(c_int), value :: comp
     end function
     integer(c_int) function dangx_put_indices(ctx, comp, ind) bind(C, name='dangx_put_indices')
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx, ind
       integer(c_int), value :: comp
     end function
     integer(c_int) function dangx_get_indices(ctx, comp, ind) bind(C, name='dangx_get_indices')
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx, ind
       integer(c_int), value :: comp
     end function
     integer(c_int) function dangx_amp_sample(ctx, group, flag, ml_mode, solver, fluct_mode, seed, stream, &
          i_max, converge, cg_iters, n_not_spd) bind(C, name='dangx_amp_sample')
       import :: c_int, c_ptr, c_double, c_int64_t
       type(c_ptr), value :: ctx
       integer(c_int), value :: group, flag, ml_mode, solver, fluct_mode, i_max
       integer(c_int64_t), value :: seed, stream
       real(c_double), value :: converge
       ! absent optional = C null pointer (F2018 15.3.7): without its count outputs the call only enqueues work on the
       ! context's device and returns, so one host thread can keep several devices busy
       integer(c_int), intent(out), optional :: cg_iters
       integer(c_int64_t), intent(out), optional :: n_not_spd
     end function
     integer(c_int) function dangx_schur_info(ctx, rel_residual, refinements) bind(C, name='dangx_schur_info')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(out) :: rel_residual(2)
       integer(c_int), intent(out) :: refinements
     end function
     integer(c_int) function dangx_amp_residual(ctx, group, flag, ml_mode, seed, stream, out) bind(C, name='dangx_amp_residual')
       import :: c_int, c_ptr, c_double, c_int64_t
       type(c_ptr), value :: ctx
       integer(c_int), value :: group, flag, ml_mode
       integer(c_int64_t), value :: seed, stream
       real(c_double), intent(out) :: out(2)
     end function
     integer(c_int) function dangx_index_sample(ctx, comp, nind, map_n, nsample, ml_mode, seed, stream, accepted) &
          bind(C, name='dangx_index_sample')
       import :: c_int, c_ptr, c_int64_t
       type(c_ptr), value :: ctx
       integer(c_int), value :: comp, nind, map_n, nsample, ml_mode
       integer(c_int64_t), value :: seed, stream
       integer(c_int64_t), intent(out), optional :: accepted
     end function
     integer(c_int) function dangx_index_sample_pair(ctx, comp, nind, map_n, nsample, ml_mode, seed, stream_first, stream_second, &
          accepted_first, accepted_second) bind(C, name='dangx_index_sample_pair')
       import :: c_int, c_ptr, c_int64_t
       type(c_ptr), value :: ctx
       integer(c_int), value :: comp, nind, map_n, nsample, ml_mode
       integer(c_int64_t), value :: seed, stream_first, stream_second
       integer(c_int64_t), intent(out), optional :: accepted_first, accepted_second
     end function
     integer(c_int) function dangx_amp_index_sample(ctx, group, flag, ml_mode, solver, fluct_mode, seed_amp, stream_amp, &
          i_max, converge, comp, nind, map_n, nsample, seed_index, stream_index, cg_iters, n_not_spd, accepted) &
          bind(C, name='dangx_amp_index_sample')
       import :: c_int, c_ptr, c_int64_t, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: group, flag, ml_mode, solver, fluct_mode, i_max, comp, nind, map_n, nsample
       real(c_double), value :: converge
       integer(c_int64_t), value :: seed_amp, stream_amp, seed_index, stream_index
       integer(c_int), intent(out), optional :: cg_iters
       integer(c_int64_t), intent(out), optional :: n_not_spd, accepted
     end function
     integer(c_int) function dangx_plane_set_sample(ctx, group, flag, ml_mode, solver, fluct_mode, seed_amp, stream_amp, i_max, converge, &
          nsweeps, comp, nind, stream, nsample, seed_index, cg_iters, n_not_spd, accepted) bind(C, name='dangx_plane_set_sample')
       import :: c_int, c_ptr, c_int32_t, c_int64_t, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: group, flag, ml_mode, solver, fluct_mode, i_max, nsweeps, nsample
       real(c_double), value :: converge
       integer(c_int64_t), value :: seed_amp, stream_amp, seed_index
       integer(c_int32_t), intent(in) :: comp(*), nind(*)          ! 0-based, in sample_spectral_parameters' order
       integer(c_int64_t), intent(in) :: stream(*)
       integer(c_int), intent(out), optional :: cg_iters
       integer(c_int64_t), intent(out), optional :: n_not_spd
       integer(c_int64_t), intent(out), optional :: accepted(*)
     end function
     integer(c_int) function dangx_plane_sweeps_sample(ctx, flag, nsweeps, comp, nind, stream, nsample, ml_mode, seed, accepted) &
          bind(C, name='dangx_plane_sweeps_sample')
       import :: c_int, c_ptr, c_int32_t, c_int64_t
       type(c_ptr), value :: ctx
       integer(c_int), value :: flag, nsweeps, nsample, ml_mode
       integer(c_int64_t), value :: seed
       integer(c_int32_t), intent(in) :: comp(*), nind(*)          ! 0-based, in sample_spectral_parameters' order
       integer(c_int64_t), intent(in) :: stream(*)
       integer(c_int64_t), intent(out), optional :: accepted(*)
     end function
     integer(c_int) function dangx_set_template(ctx, comp, tmpl, corr, nfit) bind(C, name='dangx_set_template')
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx, tmpl, corr        ! corr: integer(c_int32_t)(nbands), 1 = fitted band
       integer(c_int), value :: comp, nfit
     end function
     integer(c_int) function dangx_put_template_amplitudes(ctx, comp, ta) bind(C, name='dangx_put_template_amplitudes')
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx, ta                ! real(c_double)(nbands, nmaps) == C [map][band]
       integer(c_int), value :: comp
     end function
     integer(c_int) function dangx_get_template_amplitudes(ctx, comp, ta) bind(C, name='dangx_get_template_amplitudes')
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx, ta
       integer(c_int), value :: comp
     end function
     integer(c_int) function dangx_chisq_cached(ctx, which, pol_lo, pol_hi, chisq_sum) bind(C, name='dangx_chisq_cached')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: which, pol_lo, pol_hi
       real(c_double), intent(out) :: chisq_sum
     end function
     integer(c_int) function dangx_chisq_current(ctx, pol_lo, pol_hi, chisq_sum) bind(C, name='dangx_chisq_current')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: pol_lo, pol_hi
       real(c_double), intent(out) :: chisq_sum
     end function
     integer(c_int) function dangx_index_masked_sums(ctx, n, comp, nind, map_n, sums, counts) bind(C, name='dangx_index_masked_sums')
       import :: c_int, c_ptr, c_int32_t, c_int64_t, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: n
       integer(c_int32_t), intent(in) :: comp(*), nind(*), map_n(*)      ! comp / nind 0-based, map_n = 1..nmaps
       real(c_double), intent(out) :: sums(*)
       integer(c_int64_t), intent(out) :: counts(*)
     end function
     integer(c_int) function dangx_fullsky_prepare(ctx, comp, map_n) bind(C, name='dangx_fullsky_prepare')
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: comp, map_n
     end function
     integer(c_int) function dangx_fullsky_prepare_coarse(ctx, comp, map_n, nside, sample_nside) &
          bind(C, name='dangx_fullsky_prepare_coarse')
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: comp, map_n, nside, sample_nside
     end function
     integer(c_int) function dangx_fullsky_sums(ctx, what, theta, out, nout) bind(C, name='dangx_fullsky_sums')
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx, theta, out
       integer(c_int), value :: what, nout
     end function
     integer(c_int) function dangx_fill_index(ctx, comp, nind, map_n, value) bind(C, name='dangx_fill_index')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: comp, nind, map_n
       real(c_double), value :: value
     end function
     integer(c_int) function dangx_gain_sums(ctx, band, out) bind(C, name='dangx_gain_sums')
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx, out
       integer(c_int), value :: band
     end function
     integer(c_int) function dangx_index_sample_coarse(ctx, comp, nind, map_n, nsample, ml_mode, seed, stream, &
          nside, sample_nside, accepted) bind(C, name='dangx_index_sample_coarse')
       import :: c_int, c_ptr, c_int64_t
       type(c_ptr), value :: ctx
       integer(c_int), value :: comp, nind, map_n, nsample, ml_mode, nside, sample_nside
       integer(c_int64_t), value :: seed, stream
       integer(c_int64_t), intent(out) :: accepted
     end function
     ! coarse-Nside sampling on a pixel shard, in three phases (include/dangx.h): the caller adds the buffers of all shards
     integer(c_int) function dangx_coarse_sizes(ctx, map_n, sample_nside, n_partials, n_index) bind(C, name='dangx_coarse_sizes')
       import :: c_int, c_ptr, c_int64_t
       type(c_ptr), value :: ctx
       integer(c_int), value :: map_n, sample_nside
       integer(c_int64_t), intent(out) :: n_partials, n_index
     end function
     integer(c_int) function dangx_coarse_partials(ctx, comp, map_n, nside, sample_nside, buf) bind(C, name='dangx_coarse_partials')
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx, buf
       integer(c_int), value :: comp, map_n, nside, sample_nside
     end function
     integer(c_int) function dangx_coarse_chains(ctx, comp, nind, map_n, nsample, ml_mode, seed, stream, nside, sample_nside, &
          partials_sum, index_out) bind(C, name='dangx_coarse_chains')
       import :: c_int, c_ptr, c_int64_t
       type(c_ptr), value :: ctx, partials_sum, index_out
       integer(c_int), value :: comp, nind, map_n, nsample, ml_mode, nside, sample_nside
       integer(c_int64_t), value :: seed, stream
     end function
     integer(c_int) function dangx_coarse_writeback(ctx, comp, nind, map_n, nside, sample_nside, index_sum) &
          bind(C, name='dangx_coarse_writeback')
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx, index_sum
       integer(c_int), value :: comp, nind, map_n, nside, sample_nside
     end function
     integer(c_int) function dangx_udgrade(ctx, mode, map_in, nside_in, map_out, nside_out) bind(C, name='dangx_udgrade')
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx, map_in, map_out
       integer(c_int), value :: mode, nside_in, nside_out
     end function
     integer(c_int) function dangx_peek_indices(ctx, comp, map_n, pix, out) bind(C, name='dangx_peek_indices')
       import :: c_int, c_ptr, c_int64_t
       type(c_ptr), value :: ctx, out
       integer(c_int), value :: comp, map_n
       integer(c_int64_t), value :: pix
     end function
     integer(c_int) function dangx_index_masked_sum(ctx, comp, nind, map_n, sum, count) bind(C, name='dangx_index_masked_sum')
       import :: c_int, c_ptr, c_double, c_int64_t
       type(c_ptr), value :: ctx
       integer(c_int), value :: comp, nind, map_n
       real(c_double), intent(out) :: sum
       integer(c_int64_t), intent(out) :: count
     end function
     integer(c_int) function dangx_device_count(n) bind(C, name='dangx_device_count')
       import :: c_int
       integer(c_int), intent(out) :: n
     end function
     integer(c_int) function dangx_set_host_stride(ctx, plane_stride) bind(C, name='dangx_set_host_stride')
       import :: c_int, c_ptr, c_int64_t
       type(c_ptr), value :: ctx
       integer(c_int64_t), value :: plane_stride
     end function
     integer(c_int) function dangx_synchronize(ctx) bind(C, name='dangx_synchronize')
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
     end function
     integer(c_int) function dangx_unit_conversion(ctx, band, which, out) bind(C, name='dangx_unit_conversion')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: band, which          ! which: DANGX_A2T / DANGX_A2F / DANGX_F2T
       real(c_double), intent(out) :: out
     end function
     integer(c_int) function dangx_normalize_bandpass(tau_in, n, tau_out) bind(C, name='dangx_normalize_bandpass')
       import :: c_int, c_double
       integer(c_int), value :: n
       real(c_double), intent(in) :: tau_in(n)
       real(c_double), intent(out) :: tau_out(n)
     end function
     integer(c_int) function dangx_convert_maps(ctx, unit, cg_map, conversion) bind(C, name='dangx_convert_maps')
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx, unit, cg_map, conversion   ! integer(c_int32_t)(nbands) x2 (cg_map may be c_null_ptr), real(c_double)(nbands)
     end function
     integer(c_int) function dangx_index_plain_sum(ctx, comp, nind, map_n, sum_index, sum_mask) &
          bind(C, name='dangx_index_plain_sum')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: comp, nind, map_n
       real(c_double), intent(out) :: sum_index, sum_mask
     end function
     ! pixel-sharded (MPI) runs: fn = c_funloc of a bind(C) function that does
     ! MPI_Allreduce(MPI_IN_PLACE, buf, n, MPI_DOUBLE_PRECISION, MPI_SUM, comm) and returns 0
     integer(c_int) function dangx_set_allreduce(ctx, fn, user, is_root) bind(C, name='dangx_set_allreduce')
       import :: c_int, c_ptr, c_funptr
       type(c_ptr), value :: ctx, user
       type(c_funptr), value :: fn
       integer(c_int), value :: is_root
     end function
     ! ---- the sky-wide steps, chain included (dang_amd/csrc/dangx_sky.hip): ctxs = the contexts of this process in shard order
     integer(c_int) function dangx_fullsky_sample(ctxs, nctx, comp, nind, map_n, nsample, ml_mode, seed, stream, nside, &
          sample_nside, tuned, step_size, value, accepted) bind(C, name='dangx_fullsky_sample')
       import :: c_int, c_ptr, c_int32_t, c_int64_t, c_double
       type(c_ptr), intent(in) :: ctxs(*)
       integer(c_int), value :: nctx, comp, nind, map_n, nsample, ml_mode, nside, sample_nside
       integer(c_int64_t), value :: seed, stream
       integer(c_int32_t), intent(inout), optional :: tuned(*)       ! c%tuned(1:nindices), 1 = tuned
       real(c_double), intent(out), optional :: step_size, value
       integer(c_int64_t), intent(out), optional :: accepted
     end function
     integer(c_int) function dangx_tune_step_size(ctxs, nctx, comp, nind, nsample, ml_mode, seed, stream, theta_init, draw, &
          tuned, step_size) bind(C, name='dangx_tune_step_size')
       import :: c_int, c_ptr, c_int32_t, c_int64_t, c_double
       type(c_ptr), intent(in) :: ctxs(*)
       integer(c_int), value :: nctx, comp, nind, nsample, ml_mode
       integer(c_int64_t), value :: seed, stream
       real(c_double), intent(in) :: theta_init(2)
       integer(c_int32_t), intent(inout) :: draw                    ! running draw counter (unsigned in C; small here)
       integer(c_int32_t), intent(inout) :: tuned(*)
       real(c_double), intent(out), optional :: step_size
     end function
     integer(c_int) function dangx_tune_perpixel(ctxs, nctx, comp, nind, map_n, nsample, ml_mode, seed, stream, tuned, step_size) &
          bind(C, name='dangx_tune_perpixel')
       import :: c_int, c_ptr, c_int32_t, c_int64_t, c_double
       type(c_ptr), intent(in) :: ctxs(*)
       integer(c_int), value :: nctx, comp, nind, map_n, nsample, ml_mode
       integer(c_int64_t), value :: seed, stream
       integer(c_int32_t), intent(inout) :: tuned(*)
       real(c_double), intent(out), optional :: step_size
     end function
     integer(c_int) function dangx_fit_band_gain(ctxs, nctx, band, ml_mode, seed, stream, gain) bind(C, name='dangx_fit_band_gain')
       import :: c_int, c_ptr, c_int64_t, c_double
       type(c_ptr), intent(in) :: ctxs(*)
       integer(c_int), value :: nctx, band, ml_mode
       integer(c_int64_t), value :: seed, stream
       real(c_double), intent(out), optional :: gain
     end function
     integer(c_int) function dangx_update_tcmb(ctxs, nctx, comp, tcmb) bind(C, name='dangx_update_tcmb')
       import :: c_int, c_ptr, c_double
       type(c_ptr), intent(in) :: ctxs(*)
       integer(c_int), value :: nctx, comp
       real(c_double), intent(out), optional :: tcmb
     end function
     integer(c_int) function dangx_sky_amp_sample_c(ctxs, nctx, group, flag, ml_mode, solver, fluct_mode, seed, stream, &
          i_max, converge, cg_iters, n_not_spd) bind(C, name='dangx_sky_amp_sample')
       import :: c_int, c_ptr, c_double, c_int64_t
       type(c_ptr), intent(in) :: ctxs(*)
       integer(c_int), value :: nctx, group, flag, ml_mode, solver, fluct_mode, i_max
       integer(c_int64_t), value :: seed, stream
       real(c_double), value :: converge
       integer(c_int), intent(out), optional :: cg_iters
       integer(c_int64_t), intent(out), optional :: n_not_spd
     end function
     integer(c_int) function dangx_sky_plane_set_sample_c(ctxs, nctx, group, flag, ml_mode, solver, fluct_mode, seed_amp, stream_amp, &
          i_max, converge, nsweeps, comp, nind, stream, nsample, seed_index, cg_iters, n_not_spd, accepted) &
          bind(C, name='dangx_sky_plane_set_sample')
       import :: c_int, c_ptr, c_int32_t, c_int64_t, c_double
       type(c_ptr), intent(in) :: ctxs(*)
       integer(c_int), value :: nctx, group, flag, ml_mode, solver, fluct_mode, i_max, nsweeps, nsample
       real(c_double), value :: converge
       integer(c_int64_t), value :: seed_amp, stream_amp, seed_index
       integer(c_int32_t), intent(in) :: comp(*), nind(*)          ! 0-based, in sample_spectral_parameters' order
       integer(c_int64_t), intent(in) :: stream(*)
       integer(c_int), intent(out), optional :: cg_iters
       integer(c_int64_t), intent(out), optional :: n_not_spd
       integer(c_int64_t), intent(out), optional :: accepted(*)
     end function
     integer(c_int) function dangx_plan_fusion(ctx, npairs, pair_group, pair_flag, nsweeps, sweep_comp, sweep_nind, sweep_flag, &
          sweep_plain, solver, first_sweep) bind(C, name='dangx_plan_fusion')
       import :: c_int, c_ptr, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int), value :: npairs, nsweeps, solver
       integer(c_int32_t), intent(in) :: pair_group(*), pair_flag(*), sweep_comp(*), sweep_nind(*), sweep_flag(*), sweep_plain(*)
       integer(c_int32_t), intent(out) :: first_sweep(*)    ! 0-based position in the sweep list, or -1
     end function
     integer(c_int) function dangx_fullsky_finish_coarse(ctx, comp, map_n, nside, sample_nside, partials_sum) &
          bind(C, name='dangx_fullsky_finish_coarse')
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx, partials_sum
       integer(c_int), value :: comp, map_n, nside, sample_nside
     end function
     ! ---- the rest of the ABI (device-resident buffers, secondary seams, profiling)
     type(c_ptr) function dangx_version() bind(C, name='dangx_version')
       import :: c_ptr
     end function
     integer(c_int) function dangx_set_stream(ctx, hip_stream) bind(C, name='dangx_set_stream')
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx, hip_stream
     end function
     integer(c_int) function dangx_adopt_device_data(ctx, sig_dev, rms_dev, mask_dev) bind(C, name='dangx_adopt_device_data')
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx, sig_dev, rms_dev, mask_dev
     end function
     integer(c_int) function dangx_adopt_device_state(ctx, comp, amp_dev, idx_dev) bind(C, name='dangx_adopt_device_state')
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx, amp_dev, idx_dev
       integer(c_int), value :: comp
     end function
     type(c_ptr) function dangx_amplitude_devptr(ctx, comp) bind(C, name='dangx_amplitude_devptr')
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: comp
     end function
     type(c_ptr) function dangx_indices_devptr(ctx, comp) bind(C, name='dangx_indices_devptr')
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: comp
     end function
     integer(c_int) function dangx_sky_model_chisq_dev(ctx, pol_lo, pol_hi, chisq_sum_dev) bind(C, name='dangx_sky_model_chisq_dev')
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx, chisq_sum_dev
       integer(c_int), value :: pol_lo, pol_hi
     end function
     integer(c_int) function dangx_chisq_cached_dev(ctx, which, pol_lo, pol_hi, chisq_sum_dev) bind(C, name='dangx_chisq_cached_dev')
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx, chisq_sum_dev
       integer(c_int), value :: which, pol_lo, pol_hi
     end function
     integer(c_int64_t) function dangx_group_size(ctx, group, flag) bind(C, name='dangx_group_size')
       import :: c_int, c_ptr, c_int64_t
       type(c_ptr), value :: ctx
       integer(c_int), value :: group, flag
     end function
     integer(c_int) function dangx_compute_rhs(ctx, group, flag, b) bind(C, name='dangx_compute_rhs')      ! src/dang_cg_mod.f90:326
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx, b
       integer(c_int), value :: group, flag
     end function
     integer(c_int) function dangx_compute_Ax(ctx, group, flag, x, res) bind(C, name='dangx_compute_Ax')   ! :598
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx, x, res
       integer(c_int), value :: group, flag
     end function
     integer(c_int) function dangx_compute_sample_vector(ctx, group, flag, eta, res) bind(C, name='dangx_compute_sample_vector')  ! :913
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx, eta, res
       integer(c_int), value :: group, flag
     end function
     integer(c_int) function dangx_eval_sed(ctx, comp, band, map_n, out) bind(C, name='dangx_eval_sed')
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx, out
       integer(c_int), value :: comp, band, map_n
     end function
     integer(c_int) function dangx_rtc_kernels(ctx, n, names, names_len) bind(C, name='dangx_rtc_kernels')
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx, names                 ! names: character buffer or c_null_ptr
       integer(c_int), intent(out) :: n
       integer(c_int), value :: names_len
     end function
     integer(c_int) function dangx_rtc_compile(header, name_expr, log, log_len) bind(C, name='dangx_rtc_compile')
       import :: c_int, c_char
       character(kind=c_char), intent(in) :: header(*), name_expr(*)     ! null-terminated
       character(kind=c_char), intent(out) :: log(*)
       integer(c_int), value :: log_len
     end function
     integer(c_int) function dangx_profile_enable(ctx, on) bind(C, name='dangx_profile_enable')
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: on
     end function
     integer(c_int) function dangx_profile_reset(ctx) bind(C, name='dangx_profile_reset')
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
     end function
     integer(c_int) function dangx_profile_get(ctx, kernel_id, total_ms, launches) bind(C, name='dangx_profile_get')
       import :: c_int, c_ptr, c_double, c_int64_t
       type(c_ptr), value :: ctx
       integer(c_int), value :: kernel_id
       real(c_double), intent(out) :: total_ms
       integer(c_int64_t), intent(out) :: launches
     end function
     integer(c_int) function dangx_profile_get_planes(ctx, kernel_id, nplanes, total_ms, launches) bind(C, name='dangx_profile_get_planes')
       import :: c_int, c_ptr, c_double, c_int64_t
       type(c_ptr), value :: ctx
       integer(c_int), value :: kernel_id, nplanes
       real(c_double), intent(out) :: total_ms
       integer(c_int64_t), intent(out) :: launches
     end function
     integer(c_int) function dangx_sky_model_chisq(ctx, pol_lo, pol_hi, chisq_sum, sky, res, chi_map) &
          bind(C, name='dangx_sky_model_chisq')
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: pol_lo, pol_hi
       real(c_double), intent(out) :: chisq_sum
       type(c_ptr), value :: sky, res, chi_map
     end function
  end interface

contains

  ! the reference's error convention is print + stop (e.g. src/dang_cg_mod.f90:97-101)
  subroutine dangx_check(ctx, status, where)
    type(c_ptr), intent(in) :: ctx
    integer(c_int), intent(in) :: status
    character(len=*), intent(in) :: where
    character(kind=c_char), pointer :: msg(:)
    integer :: n
    if (status == 0) return
    write(*,*) 'dangx error in ', where, ' status = ', status
    if (c_associated(ctx)) then
       call c_f_pointer(dangx_last_error(ctx), msg, [512])
       n = 1
       do while (n < 512 .and. msg(n) /= c_null_char)
          n = n + 1
       end do
       write(*,*) msg(1:n-1)
    end if
    stop 1
  end subroutine dangx_check

  ! 64-bit random-stream label: Gibbs iteration, phase (0 amp / 1 index), three small ids
  ! (same packing as dang_amd.api.stream_id)
  function dangx_stream_id(iter, phase, a, b, c) result(s)
    integer, intent(in) :: iter, phase, a, b, c
    integer(c_int64_t) :: s
    s = ior(ior(ior(ior(shiftl(int(iter, c_int64_t), 32), shiftl(int(iand(phase, 15), c_int64_t), 28)), &
         shiftl(int(iand(a, 4095), c_int64_t), 16)), shiftl(int(iand(b, 255), c_int64_t), 8)), int(iand(c, 255), c_int64_t))
  end function dangx_stream_id

end module dangx_mod
