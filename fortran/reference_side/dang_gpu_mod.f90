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
!     call dangx_pull_state()                           ! before write_maps (119-121); write_data (116-118) needs scalars only:
!                                                       ! gpu_index_mean(), ddata%chisq, template amplitudes
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
          ! DIRECT: per-pixel block solve; groups with template / monopole / hi_fit members: Schur-complement solve
          ! (iters = -k reports k global-amplitude directions absorbed by the diffuse members, left at their value).
          ! DANGX_SOLVER_CG would run the reference's cg_search on the device instead.
          solver = DANGX_SOLVER_DIRECT
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
    real(c_double), target :: tpeek(2)
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
             if (c%index_mode(j) == 1) then
                ! full-sky value: keep the reference's chain (src/dang_sample_mod.f90:229-329) and take its three
                ! evaluate_* / eval_jeffreys_prior calls from dangx_fullsky_sums (INTEGRATION.md section 4)
                call sample_index_mh_fullsky_gpu(ddata, c, i-1, j, map_n)
             else if (c%sample_nside(j) /= nside) then
                call dangx_check(gpu_ctx, dangx_index_sample_coarse(gpu_ctx, i-1, j-1, map_n, nsample, mode, gpu_seed, &
                     dangx_stream_id(iter, 1, i-1, j-1, c%pol_flag(j,k)), nside, c%sample_nside(j), nacc), 'index_sample_coarse')
             else
                call dangx_check(gpu_ctx, dangx_index_sample(gpu_ctx, i-1, j-1, map_n, nsample, mode, gpu_seed, &
                     dangx_stream_id(iter, 1, i-1, j-1, c%pol_flag(j,k)), nacc), 'index_sample')
             end if
          end do
       end do
       ! "Update the global variable T_CMB" (src/dang_sample_mod.f90:75-78)
       if (trim(c%type) == 'T_cmb') then
          call dangx_check(gpu_ctx, dangx_peek_indices(gpu_ctx, i-1, 1, 0_c_int64_t, c_loc(tpeek)), 'peek_indices')
          T_CMB = tpeek(1)
          call dangx_check(gpu_ctx, dangx_set_tcmb(gpu_ctx, T_CMB), 'set_tcmb')
       end if
    end do
    if (sampled) call gpu_chisq(ddata)
  end subroutine sample_spectral_parameters_gpu

  ! sample_index_mh, index_mode == 1 (src/dang_sample_mod.f90:229-329): the reference's chain, with its sky-wide
  ! evaluate_lnL / evaluate_marginal_lnL / eval_jeffreys_prior replaced by dangx_fullsky_sums (one memory-bound pass
  ! each) and `index_full_res(:,...) = sample(nind)` by dangx_fill_index.  comp0 is 0-based, nind 1-based as in the
  ! reference.  (The step-size tuner, :272-275, is the same substitution inside tune_spectral_parameter_length.)
  subroutine sample_index_mh_fullsky_gpu(ddata, c, comp0, nind, map_n)
    type(dang_data)            :: ddata
    type(dang_comps), pointer  :: c
    integer(i4b), intent(in)   :: comp0, nind, map_n
    real(c_double), target     :: sample(2), theta(2), rows(4*nbands)
    real(dp)                   :: lnl, lnl_old, lnl_new, diff, ratio, num
    integer(i4b)               :: l, s1, sp
    logical(lgt)               :: sample_it

    s1 = merge(2, map_n, map_n == -1); sp = merge(2, 1, map_n == -1)
    call dangx_check(gpu_ctx, dangx_fullsky_prepare(gpu_ctx, comp0, map_n), 'fullsky_prepare')       ! :173-196
    sample = 0.d0
    call dangx_check(gpu_ctx, dangx_peek_indices(gpu_ctx, comp0, s1, 0_c_int64_t, c_loc(sample)), 'peek_indices')  ! :240-242
    theta = sample
    lnl = 0.d0; sample_it = .true.
    if (trim(c%lnl_type(nind)) == 'prior') then                                                        ! :255-257
       sample_it = .false.
       sample(nind) = rand_normal(c%gauss_prior(nind,1), c%gauss_prior(nind,2))
    else
       lnl = fullsky_lnl(sample)
    end if
    lnl_old = lnl + fullsky_prior(sample(nind))
    if (sample_it) then
       do l = 1, nsample                                                                               ! :282-324
          theta(nind) = sample(nind) + rand_normal(0.d0, c%step_size(nind))
          if (theta(nind) < c%uni_prior(nind,1) .or. theta(nind) > c%uni_prior(nind,2)) cycle
          lnl_new = fullsky_lnl(theta) + fullsky_prior(theta(nind))
          diff  = lnl_new - lnl_old
          ratio = exp(diff)
          if (trim(ml_mode) == 'optimize') then
             if (ratio > 1.d0) then
                sample(nind) = theta(nind); lnl_old = lnl_new
             end if
          else
             call RANDOM_NUMBER(num)
             if (ratio > num) then
                sample(nind) = theta(nind); lnl_old = lnl_new
             end if
          end if
       end do
    end if
    call dangx_check(gpu_ctx, dangx_fill_index(gpu_ctx, comp0, nind-1, map_n, sample(nind)), 'fill_index')   ! :329, :483

  contains

    function fullsky_lnl(th) result(v)
      real(c_double), target, intent(in) :: th(2)
      real(dp) :: v
      integer(i4b) :: q
      v = 0.d0
      if (trim(c%lnl_type(nind)) == 'chisq') then
         call dangx_check(gpu_ctx, dangx_fullsky_sums(gpu_ctx, 0, c_loc(th), c_loc(rows), 1), 'fullsky_sums')
         v = rows(1)
      else if (trim(c%lnl_type(nind)) == 'marginal') then                 ! -0.5*TNd*invTNT*TNd per (band, map), j outer
         call dangx_check(gpu_ctx, dangx_fullsky_sums(gpu_ctx, 1, c_loc(th), c_loc(rows), 2*nbands*sp), 'fullsky_sums')
         do q = 1, nbands*sp
            v = v - 0.5d0*rows(2*q-1)*(1.d0/rows(2*q))*rows(2*q-1)
         end do
      end if
    end function fullsky_lnl

    function fullsky_prior(val) result(v)
      real(dp), intent(in) :: val
      real(dp) :: v
      real(c_double), target :: th(2)
      v = 0.d0
      if (trim(c%prior_type(nind)) == 'gaussian') then
         v = log(eval_normal_prior(val, c%gauss_prior(nind,1), c%gauss_prior(nind,2)))
      else if (trim(c%prior_type(nind)) == 'jeffreys') then
         th = [val, 0.d0]
         call dangx_check(gpu_ctx, dangx_fullsky_sums(gpu_ctx, 2, c_loc(th), c_loc(rows), 1), 'fullsky_sums')
         v = log(sqrt(rows(1)))
      end if
    end function fullsky_prior

  end subroutine sample_index_mh_fullsky_gpu

  ! mask_avg(c%indices(:,map_n,j), ddata%masks(:,1)) for write_data (src/dang_data_mod.f90:716-731) without pulling the map
  function gpu_index_mean(comp, j, map_n) result(avg)
    integer(i4b), intent(in) :: comp, j, map_n
    real(dp) :: avg
    real(c_double) :: s
    integer(c_int64_t) :: n
    call dangx_check(gpu_ctx, dangx_index_masked_sum(gpu_ctx, comp-1, j-1, map_n, s, n), 'index_masked_sum')
    avg = s/n
  end function gpu_index_mean

end module dang_gpu_mod
