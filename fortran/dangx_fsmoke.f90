! dangx_fsmoke.f90 -- Fortran smoke driver for libdangx.so through dangx_mod.
!
! Reads a small problem (arrays shaped exactly like the reference's: sig_map(0:npix-1,nmaps,nbands)
! ...) from a raw little-endian file written by tests/test_fortran_gpu.py, runs one amplitude pass,
! one index sweep and the chi^2 evaluation, and writes the results back for comparison with the
! Python/ctypes path.  Two components: 'synch' power-law + 'dust' mbb, temperature only.
program dangx_fsmoke
  use, intrinsic :: iso_c_binding
  use dangx_mod
  implicit none
  integer(c_int32_t) :: npix, nmaps, nbands, nsample
  integer(c_int64_t) :: seed, nbad, nacc
  integer(c_int) :: st, iters
  real(c_double), allocatable, target :: sig(:,:,:), rms(:,:,:), mask(:,:), freqs(:)
  real(c_double), allocatable, target :: amp1(:,:), amp2(:,:), ind1(:,:,:), ind2(:,:,:)
  real(c_double) :: chisq_sum
  type(c_ptr) :: ctx
  type(dangx_dims) :: dims
  type(dangx_comp_desc) :: d
  character(len=512) :: fin, fout
  integer :: j, u
  integer, save :: ncalls = 0

  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  open(newunit=u, file=trim(fin), access='stream', form='unformatted', status='old')
  read(u) npix, nmaps, nbands, nsample, seed
  allocate(sig(0:npix-1,nmaps,nbands), rms(0:npix-1,nmaps,nbands), mask(0:npix-1,nmaps), freqs(nbands))
  allocate(amp1(0:npix-1,nmaps), amp2(0:npix-1,nmaps), ind1(0:npix-1,nmaps,1), ind2(0:npix-1,nmaps,2))
  read(u) freqs, sig, rms, mask, amp1, amp2, ind1, ind2
  close(u)

  dims = dangx_dims(npix, nmaps, nbands, 2, 0_c_int64_t, int(npix, c_int64_t), -1, 0)
  ctx = c_null_ptr
  call dangx_check(ctx, dangx_create(ctx, dims), 'dangx_create')
  do j = 1, nbands
     call dangx_check(ctx, dangx_set_band(ctx, j-1, freqs(j), 0, c_null_ptr, c_null_ptr), 'dangx_set_band')
  end do
  d = dangx_comp_desc(DANGX_POWERLAW, 1, 1, 1, 1, 0, 30.d0, [DANGX_LNL_CHISQ, 0], [DANGX_PRIOR_GAUSSIAN, 0], &
       reshape([-3.1d0, 0.1d0, 0.d0, 1.d0], [2,2]), reshape([-4.1d0, -2.1d0, 0.d0, 0.d0], [2,2]), [0.05d0, 0.d0])
  call dangx_check(ctx, dangx_set_component(ctx, 0, d), 'dangx_set_component(synch)')
  d = dangx_comp_desc(DANGX_MBB, 0, 2, 1, 1, 0, 353.d0, [DANGX_LNL_CHISQ, DANGX_LNL_CHISQ], &
       [DANGX_PRIOR_GAUSSIAN, DANGX_PRIOR_GAUSSIAN], reshape([1.6d0, 0.1d0, 19.6d0, 1.5d0], [2,2]), &
       reshape([0.6d0, 2.6d0, 4.6d0, 34.6d0], [2,2]), [0.05d0, 0.75d0])
  call dangx_check(ctx, dangx_set_component(ctx, 1, d), 'dangx_set_component(dust)')
  call dangx_check(ctx, dangx_upload_data(ctx, c_loc(sig), c_loc(rms), c_loc(mask)), 'dangx_upload_data')
  call dangx_check(ctx, dangx_put_amplitude(ctx, 0, c_loc(amp1)), 'put_amplitude')
  call dangx_check(ctx, dangx_put_amplitude(ctx, 1, c_loc(amp2)), 'put_amplitude')
  call dangx_check(ctx, dangx_put_indices(ctx, 0, c_loc(ind1)), 'put_indices')
  call dangx_check(ctx, dangx_put_indices(ctx, 1, c_loc(ind2)), 'put_indices')

  ! sample_cg_groups: group 1, flag T
  st = dangx_amp_sample(ctx, 1, DANGX_FLAG_T, DANGX_ML_SAMPLE, DANGX_SOLVER_DIRECT, DANGX_FLUCT_REFERENCE, seed, &
       dangx_stream_id(1, 0, 1, 0, DANGX_FLAG_T), 100, 1.d-8, iters, nbad)
  call dangx_check(ctx, st, 'dangx_amp_sample')
  ! sample_spectral_parameters: synch beta (comp 0, index 0), dust T (comp 1, index 1), map_n = 1
  call dangx_check(ctx, dangx_index_sample(ctx, 0, 0, 1, nsample, DANGX_ML_SAMPLE, seed, &
       dangx_stream_id(2, 1, 0, 0, DANGX_FLAG_T), nacc), 'dangx_index_sample')
  call dangx_check(ctx, dangx_index_sample(ctx, 1, 1, 1, nsample, DANGX_ML_SAMPLE, seed, &
       dangx_stream_id(2, 1, 1, 1, DANGX_FLAG_T), nacc), 'dangx_index_sample')
  call dangx_check(ctx, dangx_sky_model_chisq(ctx, 1, 1, chisq_sum, c_null_ptr, c_null_ptr, c_null_ptr), 'chisq')
  call dangx_check(ctx, dangx_get_amplitude(ctx, 0, c_loc(amp1)), 'get_amplitude')
  call dangx_check(ctx, dangx_get_amplitude(ctx, 1, c_loc(amp2)), 'get_amplitude')
  call dangx_check(ctx, dangx_get_indices(ctx, 0, c_loc(ind1)), 'get_indices')
  call dangx_check(ctx, dangx_get_indices(ctx, 1, c_loc(ind2)), 'get_indices')
  ! the pixel-sharded hook: the device CG hands its dot products to a callback written in the driver's language
  ! (an MPI_Allreduce in a real driver; one rank here, so the sum over ranks is the identity)
  ncalls = 0
  call dangx_check(ctx, dangx_set_allreduce(ctx, c_funloc(smoke_allreduce), c_null_ptr, 1), 'dangx_set_allreduce')
  st = dangx_amp_sample(ctx, 1, DANGX_FLAG_T, DANGX_ML_OPTIMIZE, DANGX_SOLVER_CG, DANGX_FLUCT_REFERENCE, seed, &
       dangx_stream_id(3, 0, 1, 0, DANGX_FLAG_T), 4, 0.d0, iters, nbad)
  call dangx_check(ctx, st, 'dangx_amp_sample(cg)')
  call dangx_check(ctx, dangx_destroy(ctx), 'dangx_destroy')

  open(newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace')
  write(u) chisq_sum, amp1, amp2, ind1, ind2
  close(u)
  write(*,'(a,es24.16,a,i0)') 'dangx_fsmoke ok: chisq_sum = ', chisq_sum, '  not_spd = ', nbad
  write(*,'(a,i0,a,i0)') 'allreduce callback calls = ', ncalls, '  cg iterations = ', iters

contains

  integer(c_int) function smoke_allreduce(user, buf, n) bind(C)
    type(c_ptr), value :: user
    integer(c_int64_t), value :: n
    real(c_double) :: buf(n)
    ncalls = ncalls + 1
    if (n < 1 .or. buf(1) /= buf(1)) then   ! touch the buffer: a NaN or an empty call is an error
       smoke_allreduce = 1
    else
       smoke_allreduce = 0
    end if
  end function smoke_allreduce

end program dangx_fsmoke
