! dang_gpu_drive.f90 -- RUNS the reference-side wrapper (dang_gpu_mod.f90) on a GPU.
!
! dang_gpu_mod `use`s the reference's modules, which cannot be built in this image (HEALPix-F90 / CFITSIO / MPI).  Here it is
! compiled against the mock modules of stubs/stubs.f90 -- builder-owned declarations of the derived types and globals the
! wrapper touches, nothing of the reference -- and this program plays `program dang` (src/dang.f90:43-126): it fills
! dang_params / dang_data / component_list / cg_groups / bp from a problem file written by tests/test_refside_gpu.py or
! bench.py, calls dangx_init, runs the Gibbs loop through the wrapper's entry points and writes the state back.
!
!   dang_gpu_drive <problem> <result> <nctx> <mode> [tile]
!     mode  twocall : call sample_cg_groups_gpu ; sample_spectral_parameters_gpu ; sample_calibrators_gpu   (src/dang.f90:101-110)
!           fused   : call gibbs_iteration_gpu (iterations > 1) ; sample_calibrators_gpu
!     tile  every pixel of the problem's maps stands for `tile` consecutive pixels (a timing run at Nside 1024 from an Nside 8
!           file; the masked region stays one run of pixels, as a Galactic mask is in RING order; the random streams are keyed by
!           the global pixel, so every pixel still runs its own chain); no maps are written back
! Every iteration ends with write_data_gpu (the ASCII traces, into the directory of <result>) as in src/dang.f90:116-118; the
! run ends with dangx_refresh_host_state (what precedes write_maps, :119-121).
subroutine mpi_allreduce(sendbuf, recvbuf, count, datatype, op, comm, ierror)
  ! the wrapper's MPI branch is never taken here (numprocs = 1); the symbol only has to exist
  integer :: sendbuf, recvbuf(*), count, datatype, op, comm, ierror
  ierror = 0
end subroutine mpi_allreduce

program dang_gpu_drive
  use, intrinsic :: iso_c_binding
  use healpix_types
  use dang_util_mod
  use dang_param_mod
  use dang_bp_mod
  use dang_component_mod
  use dang_data_mod
  use dang_cg_mod
  use dangx_mod
  use dangx_multi_mod
  use dang_gpu_mod
  implicit none
  type(dang_params) :: dpar
  type(dang_data), target :: ddata
  type(dang_comps), pointer :: cc
  type(dangx_comp_desc) :: d
  integer(c_int32_t) :: hdr(8), mlm, si(2), pf(2), im(2), sn(2), tn(2), nfit_in, g4(4)
  integer(c_int32_t), allocatable :: icorr(:), ifit(:)
  integer(c_int64_t) :: seed
  real(c_double) :: nump_in
  real(c_double), allocatable :: freqs(:), small3(:,:,:), small2(:,:)
  character(len=32), allocatable :: blabel(:)
  character(len=16) :: clabel, ilabel(2)
  character(len=512) :: fin, fout, arg, mode
  integer :: u, i, j, l, k, npix0, niter, ngroups, nctx, tile, t, it_first
  integer(i8b) :: c0, c1, crate
  real(dp) :: secs

  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  call get_command_argument(3, arg); read(arg, *) nctx
  call get_command_argument(4, mode)
  tile = 1
  if (command_argument_count() >= 5) then
     call get_command_argument(5, arg); read(arg, *) tile
  end if

  open(newunit=u, file=trim(fin), access='stream', form='unformatted', status='old')
  read(u) hdr, seed, nump_in, mlm
  npix0 = hdr(1); nmaps = hdr(2); nbands = hdr(3); ncomp = hdr(4); nsample = hdr(5); niter = hdr(6); ngroups = hdr(7)
  npix = npix0*tile
  nside = hdr(8)
  if (tile > 1) nside = nint(sqrt(npix/12.d0))
  nump = nint(nump_in)*tile
  ncg_groups = ngroups
  rank = 0; numprocs = 1
  tqu(1) = 'T'; tqu(2) = 'Q'; tqu(3) = 'U'
  ml_mode = merge('sample    ', 'optimize  ', mlm == 1)
  dpar%ml_mode = ml_mode
  dpar%outdir = fout(1:index(fout, '/', back=.true.))
  gpu_seed = seed
  allocate(freqs(nbands), blabel(nbands), ifit(nbands), icorr(nbands))
  allocate(ddata%gain(nbands), ddata%offset(nbands), ddata%fit_gain(nbands), ddata%label(nbands), ddata%conversion(nbands))
  read(u) freqs, ddata%gain, ddata%offset, ifit, blabel
  ddata%fit_gain = ifit /= 0
  ddata%label = blabel
  ddata%conversion = 1.d0
  allocate(ddata%pol_type(nmaps)); ddata%pol_type = [(k, k = 1, nmaps)]
  allocate(dpar%pol_type(nmaps)); dpar%pol_type = ddata%pol_type
  allocate(bp(nbands))
  do j = 1, nbands
     bp(j)%id = 'delta'; bp(j)%nu_c = freqs(j); bp(j)%n = 0
  end do
  allocate(component_list(ncomp))
  do l = 1, ncomp
     allocate(component_list(l)%p)
     cc => component_list(l)%p
     read(u) d, clabel, ilabel, si, pf, im, sn, tn, nfit_in, icorr
     cc%label = clabel; cc%type = type_name(d%type)
     cc%nindices = d%nindices; cc%cg_group = d%cg_group; cc%sample_amplitude = d%sample_amplitude /= 0
     cc%nu_ref = d%nu_ref; cc%nfit = nfit_in
     allocate(cc%corr(nbands)); cc%corr = icorr /= 0
     k = max(cc%nindices, 1)
     allocate(cc%ind_label(k), cc%sample_index(k), cc%tuned(k), cc%sample_nside(k), cc%step_size(k), cc%index_mode(k), &
          cc%lnl_type(k), cc%prior_type(k), cc%gauss_prior(k,2), cc%uni_prior(k,2), cc%nflag(k), cc%pol_flag(k,1))
     cc%sample_index = .false.; cc%tuned = .true.; cc%sample_nside = nside; cc%step_size = 0.d0; cc%index_mode = 2
     cc%nflag = 1; cc%pol_flag = 0; cc%lnl_type = 'chisq'; cc%prior_type = 'uniform'; cc%gauss_prior = 0.d0; cc%uni_prior = 0.d0
     do j = 1, cc%nindices
        cc%ind_label(j) = ilabel(j); cc%sample_index(j) = si(j) /= 0; cc%pol_flag(j,1) = pf(j); cc%index_mode(j) = im(j)
        cc%sample_nside(j) = merge(int(sn(j)), nside, sn(j) > 0 .and. tile == 1); cc%tuned(j) = tn(j) /= 0
        cc%step_size(j) = d%step_size(j)
        cc%gauss_prior(j,:) = d%gauss_prior(:,j); cc%uni_prior(j,:) = d%uni_prior(:,j)
        cc%lnl_type(j) = lnl_name(d%lnl_type(j)); cc%prior_type(j) = prior_name(d%prior_type(j))
     end do
  end do
  allocate(cg_groups(ngroups))
  do i = 1, ngroups
     allocate(cg_groups(i)%p)
     read(u) g4                                       ! group number, flag, sample, number of template-type members
     cg_groups(i)%p%cg_group = g4(1); cg_groups(i)%p%nflag = 1
     allocate(cg_groups(i)%p%pol_flag(1)); cg_groups(i)%p%pol_flag(1) = g4(2)
     cg_groups(i)%p%sample = g4(3) /= 0; cg_groups(i)%p%ntemp = g4(4)
     cg_groups(i)%p%i_max = 100; cg_groups(i)%p%converge = 1.d-8
  end do
  allocate(ddata%sig_map(0:npix-1,nmaps,nbands), ddata%rms_map(0:npix-1,nmaps,nbands), ddata%masks(0:npix-1,nmaps))
  allocate(ddata%sky_model(0:npix-1,nmaps,nbands), ddata%res_map(0:npix-1,nmaps,nbands), ddata%chi_map(0:npix-1,nmaps))
  allocate(small3(0:npix0-1,nmaps,nbands), small2(0:npix0-1,nmaps))
  read(u) small3; call tile3(small3, ddata%sig_map)
  read(u) small3; call tile3(small3, ddata%rms_map)
  read(u) small2; call tile2(small2, ddata%masks)
  do l = 1, ncomp
     cc => component_list(l)%p
     allocate(cc%amplitude(0:npix-1,nmaps))
     read(u) small2; call tile2(small2, cc%amplitude)
     if (cc%nindices > 0) then
        allocate(cc%indices(0:npix-1,nmaps,cc%nindices))
        do j = 1, cc%nindices
           read(u) small2; call tile2(small2, cc%indices(:,:,j))
        end do
     end if
     if (trim(cc%type) == 'template' .or. trim(cc%type) == 'monopole' .or. trim(cc%type) == 'hi_fit') then
        allocate(cc%template(0:npix-1,nmaps), cc%template_amplitudes(nbands,nmaps), cc%temp_norm(nmaps))
        read(u) small2; call tile2(small2, cc%template)
        read(u) cc%template_amplitudes, cc%temp_norm
     end if
  end do
  close(u)
  deallocate(small3, small2)

  ! ---- program dang from here on (src/dang.f90:79-126), through the wrapper
  call dangx_init(dpar, ddata, nctx)
  secs = 0.d0; it_first = 3
  do iter = 1, niter
     if (iter == it_first + 1) then                       ! time iterations it_first+1 .. niter: the first two full ones warm up (index maps that start spatially constant take the generic launches once, kernels specialised at run time are compiled on first use)
        call dangx_sky_wait(gpu_sky)
        call system_clock(c0, crate)
     end if
     if (trim(mode) == 'fused' .and. iter > 1) then
        call gibbs_iteration_gpu(dpar, ddata)
        call sample_calibrators_gpu(ddata)
     else
        call sample_cg_groups_gpu(dpar, ddata)
        if (iter > 1) then
           call sample_spectral_parameters_gpu(dpar, ddata)
           call sample_calibrators_gpu(ddata)
        end if
     end if
     if (tile == 1) then
        do k = dpar%pol_type(1), dpar%pol_type(size(dpar%pol_type))
           call write_data_gpu(ddata, dpar, k)
        end do
     end if
  end do
  call dangx_sky_wait(gpu_sky)
  if (niter > it_first) then
     call system_clock(c1)
     secs = real(c1 - c0, dp)/crate/(niter - it_first)
  end if
  if (tile == 1) call dangx_refresh_host_state(ddata)

  open(newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace')
  write(u) ddata%chisq, T_CMB, secs, ddata%gain, ddata%offset
  do l = 1, ncomp
     cc => component_list(l)%p
     write(u) real(cc%step_size(1), c_double), real(cc%step_size(size(cc%step_size)), c_double), &
          merge(1.d0, 0.d0, cc%tuned(1)), merge(1.d0, 0.d0, cc%tuned(size(cc%tuned)))
  end do
  if (tile == 1) then
     write(u) ddata%sky_model, ddata%res_map, ddata%chi_map
     do l = 1, ncomp
        cc => component_list(l)%p
        write(u) cc%amplitude
        if (cc%nindices > 0) write(u) cc%indices
        if (allocated(cc%template_amplitudes)) write(u) cc%template_amplitudes
     end do
  end if
  close(u)
  write(*,'(a,a,a,i0,a,i0,a,es24.16)') 'dang_gpu_drive ok: mode = ', trim(mode), '  contexts = ', nctx, '  npix = ', npix, &
       '  chisq = ', ddata%chisq
  if (secs > 0.d0) write(*,'(a,f12.6,a,f10.4)') 'drive seconds per iteration = ', secs, '  it/s = ', 1.d0/secs
  call dangx_sky_destroy(gpu_sky)

contains

  character(len=16) function type_name(code)
    integer(c_int32_t), intent(in) :: code
    character(len=16), parameter :: names(9) = [character(len=16) :: 'power-law', 'mbb', 'freefree', 'lognormal', 'cmb', 'T_cmb', &
         'template', 'monopole', 'hi_fit']
    type_name = names(code)
  end function type_name
  character(len=16) function lnl_name(code)
    integer(c_int32_t), intent(in) :: code
    character(len=16), parameter :: names(3) = [character(len=16) :: 'chisq', 'marginal', 'prior']
    lnl_name = names(max(1, min(3, code)))
  end function lnl_name
  character(len=16) function prior_name(code)
    integer(c_int32_t), intent(in) :: code
    character(len=16), parameter :: names(3) = [character(len=16) :: 'gaussian', 'uniform', 'jeffreys']
    prior_name = names(max(1, min(3, code)))
  end function prior_name

  subroutine tile2(small, big)
    real(c_double), intent(in)  :: small(0:,:)
    real(c_double), intent(out) :: big(0:,:)
    integer :: p, q
    do q = 1, size(small, 2)
       do p = 0, npix0-1
          big(p*tile:(p+1)*tile-1, q) = small(p, q)
       end do
    end do
  end subroutine tile2
  subroutine tile3(small, big)
    real(c_double), intent(in)  :: small(0:,:,:)
    real(c_double), intent(out) :: big(0:,:,:)
    integer :: p, q, r
    do r = 1, size(small, 3)
       do q = 1, size(small, 2)
          do p = 0, npix0-1
             big(p*tile:(p+1)*tile-1, q, r) = small(p, q, r)
          end do
       end do
    end do
  end subroutine tile3

end program dang_gpu_drive
