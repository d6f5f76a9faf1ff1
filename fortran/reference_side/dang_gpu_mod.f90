! dang_gpu_mod.f90 -- the REFERENCE-SIDE wrapper a dang maintainer adds to src/ (see INTEGRATION.md).
!
! It `use`s the reference's own modules (dang_util_mod, dang_param_mod, dang_bp_mod,
! dang_component_mod, dang_data_mod, dang_cg_mod), so it is compiled inside the reference's build
! (mpif90 + HEALPix), not in this repository: those modules need HEALPix-F90/CFITSIO/MPI, which this
! image lacks.  The bind(C) layer it calls (fortran/dangx_mod.f90) IS compiled and tested here
! (fortran/dangx_fsmoke.f90, tests/test_fortran_gpu.py).
!
! Drop-in use in src/dang.f90:
!     call dangx_init(dpar, ddata)                      ! once, after line 79
!     call sample_cg_groups_gpu(dpar, ddata)            ! instead of line 101
!     call sample_spectral_parameters_gpu(dpar, ddata)  ! instead of line 106
!     call dangx_pull_state()                           ! before write_data / write_maps (116-121)
module dang_gpu_mod
  use, intrinsic :: iso_c_binding
  use healpix_types
  use dang_util_mod
  use dang_param_mod
  use dang_bp_mod
  use dang_component_mod
  use dang_data_mod
  use dang_cg_mod
  use dangx_mod
  implicit none

  type(c_ptr), save  :: gpu_ctx = c_null_ptr
  integer(c_int64_t) :: gpu_seed = 1234_c_int64_t   ! the reference calls RANDOM_SEED() unseeded (src/dang.f90:67)

contains

  integer(c_int) function type_code(c)
    type(dang_comps), intent(in) :: c
    select case (trim(c%type))
    case ('power-law'); type_code = DANGX_POWERLAW
    case ('mbb');       type_code = DANGX_MBB
    case ('freefree');  type_code = DANGX_FREEFREE
    case ('lognormal'); type_code = DANGX_LOGNORMAL
    case ('cmb');       type_code = DANGX_CMB
    case ('T_cmb');     type_code = DANGX_TCMB
    case ('template');  type_code = DANGX_TEMPLATE
    case ('monopole');  type_code = DANGX_MONOPOLE
    case ('hi_fit');    type_code = DANGX_HIFIT
    case default
       write(*,*) 'dang_gpu_mod: component type not on the GPU path: ', trim(c%type)
       stop
    end select
  end function type_code

  subroutine dangx_init(dpar, ddata)
    type(dang_params)        :: dpar
    type(dang_data), target  :: ddata
    type(dang_comps), pointer :: c
    type(dangx_dims)      :: dims
    type(dangx_comp_desc) :: d
    real(c_double), allocatable, target :: gain(:), offs(:)
    integer(c_int32_t), allocatable, target :: icorr(:)
    integer :: i, j

    dims = dangx_dims(npix, nmaps, nbands, ncomp, 0_c_int64_t, int(npix, c_int64_t), -1, 0)
    call dangx_check(gpu_ctx, dangx_create(gpu_ctx, dims), 'dangx_create')
    do j = 1, nbands
       if (trim(bp(j)%id) == 'delta') then
          call dangx_check(gpu_ctx, dangx_set_band(gpu_ctx, j-1, bp(j)%nu_c, 0, c_null_ptr, c_null_ptr), 'set_band')
       else
          call dangx_check(gpu_ctx, dangx_set_band(gpu_ctx, j-1, bp(j)%nu_c, bp(j)%n, c_loc(bp(j)%nu0), &
               c_loc(bp(j)%tau0)), 'set_band')
       end if
    end do
    call dangx_check(gpu_ctx, dangx_set_tcmb(gpu_ctx, T_CMB), 'set_tcmb')
    do i = 1, ncomp
       c => component_list(i)%p
       d%type = type_code(c)
       d%is_synch = merge(1, 0, trim(c%label) == 'synch')
       d%nindices = c%nindices
       d%cg_group = c%cg_group
       d%sample_amplitude = merge(1, 0, c%sample_amplitude)
       d%nu_ref = c%nu_ref
       d%lnl_type = 0; d%prior_type = 0; d%gauss_prior = 0.d0; d%uni_prior = 0.d0; d%step_size = 0.d0
       do j = 1, c%nindices
          select case (trim(c%lnl_type(j)))
          case ('chisq');    d%lnl_type(j) = DANGX_LNL_CHISQ
          case ('marginal'); d%lnl_type(j) = DANGX_LNL_MARGINAL
          case ('prior');    d%lnl_type(j) = DANGX_LNL_PRIOR
          end select
          select case (trim(c%prior_type(j)))
          case ('gaussian'); d%prior_type(j) = DANGX_PRIOR_GAUSSIAN
          case ('uniform');  d%prior_type(j) = DANGX_PRIOR_UNIFORM
          case ('jeffreys'); d%prior_type(j) = DANGX_PRIOR_JEFFREYS
          end select
          d%gauss_prior(:, j) = c%gauss_prior(j, :)      ! reference (nind,2) -> C [ind][2]
          d%uni_prior(:, j)   = c%uni_prior(j, :)
          d%step_size(j)      = c%step_size(j)
       end do
       call dangx_check(gpu_ctx, dangx_set_component(gpu_ctx, i-1, d), 'set_component')
       if (d%type >= DANGX_TEMPLATE) then     ! c%template(0:npix-1,nmaps), c%corr(nbands), c%nfit
          allocate(icorr(nbands)); icorr = merge(1_c_int32_t, 0_c_int32_t, c%corr)
          call dangx_check(gpu_ctx, dangx_set_template(gpu_ctx, i-1, c_loc(c%template), c_loc(icorr), c%nfit), 'set_template')
          deallocate(icorr)
          ! c%template_amplitudes(nbands,nmaps) is already [map][band] in memory
          call dangx_check(gpu_ctx, dangx_put_template_amplitudes(gpu_ctx, i-1, c_loc(c%template_amplitudes)), 'put_tamp')
       end if
    end do
    allocate(gain(nbands), offs(nbands)); gain = ddata%gain; offs = ddata%offset
    call dangx_check(gpu_ctx, dangx_set_calibration(gpu_ctx, c_loc(gain), c_loc(offs)), 'set_calibration')
    call dangx_check(gpu_ctx, dangx_upload_data(gpu_ctx, c_loc(ddata%sig_map), c_loc(ddata%rms_map), &
         c_loc(ddata%masks)), 'upload_data')
    call dangx_push_state()
  end subroutine dangx_init

  subroutine dangx_push_state()
    type(dang_comps), pointer :: c
    integer :: i
    do i = 1, ncomp
       c => component_list(i)%p
       call dangx_check(gpu_ctx, dangx_put_amplitude(gpu_ctx, i-1, c_loc(c%amplitude)), 'put_amplitude')
       if (c%nindices > 0) call dangx_check(gpu_ctx, dangx_put_indices(gpu_ctx, i-1, c_loc(c%indices)), 'put_indices')
    end do
  end subroutine dangx_push_state

  subroutine dangx_pull_state()
    type(dang_comps), pointer :: c
    integer :: i
    do i = 1, ncomp
       c => component_list(i)%p
       call dangx_check(gpu_ctx, dangx_get_amplitude(gpu_ctx, i-1, c_loc(c%amplitude)), 'get_amplitude')
       if (c%nindices > 0) call dangx_check(gpu_ctx, dangx_get_indices(gpu_ctx, i-1, c_loc(c%indices)), 'get_indices')
       if (trim(c%type) == 'template' .or. trim(c%type) == 'monopole' .or. trim(c%type) == 'hi_fit') &
            call dangx_check(gpu_ctx, dangx_get_template_amplitudes(gpu_ctx, i-1, c_loc(c%template_amplitudes)), 'get_tamp')
    end do
  end subroutine dangx_pull_state

  subroutine gpu_chisq(ddata)
    ! update_sky_model + compute_chisq, src/dang_data_mod.f90:339-396, 494-526
    type(dang_data) :: ddata
    real(c_double)  :: s
    call dangx_check(gpu_ctx, dangx_sky_model_chisq(gpu_ctx, ddata%pol_type(1), ddata%pol_type(size(ddata%pol_type)), &
         s, c_null_ptr, c_null_ptr, c_null_ptr), 'sky_model_chisq')
    ddata%chisq = s/nbands/nump
    write(*,fmt='(i6,a,E16.5)') iter, " - Chisq: ", ddata%chisq
  end subroutine gpu_chisq

  subroutine sample_cg_groups_gpu(dpar, ddata)
    ! same signature and effect as sample_cg_groups, src/dang_cg_mod.f90:142-177
    type(dang_data)   :: ddata
    type(dang_params) :: dpar
    integer(i4b) :: i, f, k, mode
    integer(c_int) :: iters, solver
    integer(c_int64_t) :: nbad
    mode = merge(DANGX_ML_SAMPLE, DANGX_ML_OPTIMIZE, trim(dpar%ml_mode) == 'sample')
    do i = 1, ncg_groups
       if (cg_groups(i)%p%sample) then
          write(*,fmt='(a,i4)') "Computing a CG search of CG group ", i
          ! a group with template / monopole / hi_fit members is a coupled system: the reference's CG on the device
          solver = DANGX_SOLVER_DIRECT
          if (cg_groups(i)%p%ntemp > 0) then
             do k = 1, cg_groups(i)%p%ncg_components
                select case (trim(cg_groups(i)%p%cg_component(k)%p%type))
                case ('template', 'monopole', 'hi_fit'); solver = DANGX_SOLVER_CG
                end select
             end do
          end if
          do f = 1, cg_groups(i)%p%nflag
             call dangx_check(gpu_ctx, dangx_amp_sample(gpu_ctx, i, cg_groups(i)%p%pol_flag(f), mode, &
                  solver, DANGX_FLUCT_REFERENCE, gpu_seed, &
                  dangx_stream_id(iter, 0, i, 0, cg_groups(i)%p%pol_flag(f)), cg_groups(i)%p%i_max, &
                  cg_groups(i)%p%converge, iters, nbad), 'amp_sample')
             if (nbad > 0) write(*,*) 'warning: ', nbad, ' non-SPD pixel blocks left unchanged'
          end do
          call gpu_chisq(ddata)
       end if
    end do
  end subroutine sample_cg_groups_gpu

  subroutine sample_spectral_parameters_gpu(dpar, ddata)
    ! same signature and effect as sample_spectral_parameters, src/dang_sample_mod.f90:21-86
    type(dang_data)   :: ddata
    type(dang_params) :: dpar
    type(dang_comps), pointer :: c
    integer(i4b) :: i, j, k, map_n, mode
    integer(c_int64_t) :: nacc
    logical(lgt) :: sampled
    sampled = .false.
    mode = merge(DANGX_ML_SAMPLE, DANGX_ML_OPTIMIZE, trim(dpar%ml_mode) == 'sample')
    do i = 1, ncomp
       c => component_list(i)%p
       if (c%nindices == 0) cycle
       if (.not. any(c%sample_index)) cycle
       sampled = .true.
       do j = 1, c%nindices
          if (.not. c%sample_index(j)) cycle
          do k = 1, c%nflag(j)
             if (iand(c%pol_flag(j,k),1) .ne. 0) then
                map_n = 1
             else if (iand(c%pol_flag(j,k),2) .ne. 0) then
                map_n = 2
             else if (iand(c%pol_flag(j,k),4) .ne. 0) then
                map_n = 3
             else if (iand(c%pol_flag(j,k),8) .ne. 0) then
                map_n = -1
             else
                write(*,*) "There is something wrong with the poltype flag"
                cycle
             end if
             call dangx_check(gpu_ctx, dangx_index_sample(gpu_ctx, i-1, j-1, map_n, nsample, mode, gpu_seed, &
                  dangx_stream_id(iter, 1, i-1, j-1, c%pol_flag(j,k)), nacc), 'index_sample')
          end do
       end do
    end do
    if (sampled) call gpu_chisq(ddata)
  end subroutine sample_spectral_parameters_gpu

end module dang_gpu_mod
