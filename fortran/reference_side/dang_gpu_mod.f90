! dang_gpu_mod.f90 -- the REFERENCE-SIDE wrapper a dang maintainer adds to src/ (see INTEGRATION.md).
!
! It `use`s the reference's own modules (dang_util_mod, dang_param_mod, dang_bp_mod, dang_component_mod,
! dang_data_mod, dang_cg_mod), so it is compiled inside the reference's build (mpif90 + HEALPix), not in this
! repository: those modules need HEALPix-F90 / CFITSIO / MPI, which this image lacks.  What IS done here:
!   * the layers below it -- fortran/dangx_mod.f90 (bind(C)) and fortran/dangx_multi_mod.f90 (one sky over several
!     GPUs from one host thread) -- are compiled by flang and run on a GPU in the tests (fortran/dangx_fsmoke.f90);
!   * this file is type-checked by flang against builder-owned stub modules that declare only the names it touches
!     (fortran/reference_side/stubs/, `dang_amd/_build.py:check_reference_side`): that catches syntax and type
!     errors, it says nothing about the reference.
!
! Drop-in use in src/dang.f90 (the edits are listed in INTEGRATION.md section 2):
!     call dangx_init(dpar, ddata)                          once, after line 79
!     call sample_cg_groups_gpu(dpar, ddata)                instead of line 101
!     call sample_spectral_parameters_gpu(dpar, ddata)      instead of line 106
!     call sample_calibrators_gpu(ddata)                    instead of line 110
!     call write_data_gpu(ddata, dpar, k)                   instead of ddata%write_data(dpar, k), line 117
!     call dangx_refresh_host_state(ddata)                  before ddata%write_maps(dpar), line 120
! Between outputs the amplitude and index maps live in HBM only; the host copies (c%amplitude, c%indices,
! ddata%sky_model / res_map / chi_map) are refreshed by dangx_refresh_host_state, i.e. at the map-output cadence.
! write_data_gpu writes the same files in the same formats as write_data (src/dang_data_mod.f90:666-761) from
! device reductions: the reference's own write_data would recompute chi^2 from the (stale) host sky model and the
! index means from the (stale) host index maps every iteration.
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
  use dangx_multi_mod
  implicit none

  type(dangx_sky), save :: gpu_sky
  integer(c_int64_t)    :: gpu_seed = 1234_c_int64_t   ! the reference calls RANDOM_SEED() unseeded (src/dang.f90:67)
  integer(i4b)          :: gpu_pix0 = 0                ! first pixel of this PROCESS (0 unless the driver is run under MPI)

contains

  integer(c_int) function type_code(cc)
    type(dang_comps), intent(in) :: cc
    select case (trim(cc%type))
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
       write(*,*) 'dang_gpu_mod: component type not on the GPU path: ', trim(cc%type)
       stop
    end select
  end function type_code

  subroutine fill_desc(cc, d)
    type(dang_comps), intent(in)       :: cc
    type(dangx_comp_desc), intent(out) :: d
    integer :: j
    d%type = type_code(cc)
    d%is_synch = merge(1, 0, trim(cc%label) == 'synch')
    d%nindices = cc%nindices
    d%cg_group = cc%cg_group
    d%sample_amplitude = merge(1, 0, cc%sample_amplitude)
    d%reserved = 0
    d%nu_ref = cc%nu_ref
    d%lnl_type = 0; d%prior_type = 0; d%gauss_prior = 0.d0; d%uni_prior = 0.d0; d%step_size = 0.d0
    do j = 1, cc%nindices
       select case (trim(cc%lnl_type(j)))
       case ('chisq');    d%lnl_type(j) = DANGX_LNL_CHISQ
       case ('marginal'); d%lnl_type(j) = DANGX_LNL_MARGINAL
       case ('prior');    d%lnl_type(j) = DANGX_LNL_PRIOR
       end select
       select case (trim(cc%prior_type(j)))
       case ('gaussian'); d%prior_type(j) = DANGX_PRIOR_GAUSSIAN
       case ('uniform');  d%prior_type(j) = DANGX_PRIOR_UNIFORM
       case ('jeffreys'); d%prior_type(j) = DANGX_PRIOR_JEFFREYS
       end select
       d%gauss_prior(:, j) = cc%gauss_prior(j, :)      ! reference (nind,2) -> C [ind][2]
       d%uni_prior(:, j)   = cc%uni_prior(j, :)
       d%step_size(j)      = cc%step_size(j)
    end do
  end subroutine fill_desc

  ! sum over the MPI ranks of an MPI-enabled driver (one process per GPU); the reference itself is a single process
  integer(c_int) function dang_allreduce(user, buf, n) bind(C)
    type(c_ptr), value :: user
    integer(c_int64_t), value :: n
    real(c_double) :: buf(n)
    integer :: ierr_l
    call mpi_allreduce(MPI_IN_PLACE, buf, int(n), MPI_DOUBLE_PRECISION, MPI_SUM, MPI_COMM_WORLD, ierr_l)
    dang_allreduce = ierr_l
  end function dang_allreduce

  real(dp) function rank_sum(v)
    real(dp), intent(in) :: v
    real(dp) :: b(1)
    integer :: ierr_l
    b(1) = v
    if (numprocs > 1) call mpi_allreduce(MPI_IN_PLACE, b, 1, MPI_DOUBLE_PRECISION, MPI_SUM, MPI_COMM_WORLD, ierr_l)
    rank_sum = b(1)
  end function rank_sum

  ! once, after initialize_cg_groups (src/dang.f90:79): hand the static state and the maps to the device(s).
  ! Single process (the reference as it is): one context per visible GPU, each on its window of the full-sky arrays.
  ! Under MPI (numprocs > 1: one process per GPU, every process holding the full-sky arrays as the reference reads them):
  ! one context per process on the process's own pixel range; in-solve sums go through dang_allreduce.
  subroutine dangx_init(dpar, ddata, ngpu)
    type(dang_params)             :: dpar
    type(dang_data), target       :: ddata
    integer, intent(in), optional :: ngpu          ! contexts of this process (default: every visible device)
    type(dang_comps), pointer     :: cc
    type(dangx_comp_desc)         :: d
    real(c_double), allocatable, target     :: gain(:), offs(:)
    integer(c_int32_t), allocatable, target :: icorr(:)
    integer(c_int) :: ndev
    integer(c_int64_t) :: p0, np
    integer :: i, j, nctx, r
    integer, allocatable :: iseed(:)

    call dangx_check(c_null_ptr, dangx_device_count(ndev), 'dangx_device_count')
    nctx = ndev
    if (present(ngpu)) nctx = ngpu
    do i = 1, ncomp                                  ! coupled groups cannot be split over contexts of ONE host thread
       cc => component_list(i)%p
       if (type_code(cc) >= DANGX_TEMPLATE .and. cc%sample_amplitude) nctx = 1
    end do
    if (numprocs > 1) then
       ! one context on this rank's range [p0, p0+np) of the sky: a one-context sky whose window starts at p0
       call dangx_shard_range(int(npix, c_int64_t), rank, numprocs, p0, np)
       gpu_pix0 = int(p0, i4b)
       gpu_sky%nctx = 1; gpu_sky%npix_global = npix; gpu_sky%nmaps = nmaps; gpu_sky%nbands = nbands; gpu_sky%ncomp = ncomp
       gpu_sky%pix0(1) = p0; gpu_sky%npix(1) = np
       call dangx_check(c_null_ptr, dangx_create(gpu_sky%ctx(1), dangx_dims(int(np, c_int32_t), nmaps, nbands, ncomp, p0, &
            int(npix, c_int64_t), mod(rank, ndev), 0)), 'dangx_create')
       call dangx_check(gpu_sky%ctx(1), dangx_set_host_stride(gpu_sky%ctx(1), int(npix, c_int64_t)), 'dangx_set_host_stride')
       call dangx_check(gpu_sky%ctx(1), dangx_set_allreduce(gpu_sky%ctx(1), c_funloc(dang_allreduce), c_null_ptr, &
            merge(1, 0, rank == master)), 'dangx_set_allreduce')
       ! the sky-wide chains (full-sky index mode, tuner, gain draw) run on the host of EVERY rank: they must draw the
       ! same numbers, so the intrinsic generator gets one seed on all ranks (the reference leaves it unseeded)
       call random_seed(size=r)
       allocate(iseed(r)); iseed = 20240601
       call random_seed(put=iseed)
       deallocate(iseed)
    else
       call dangx_sky_create(gpu_sky, int(npix, c_int64_t), nmaps, nbands, ncomp, nctx)
    end if

    do j = 1, nbands
       if (trim(bp(j)%id) == 'delta') then
          call dangx_sky_set_band(gpu_sky, j-1, bp(j)%nu_c, 0, c_null_ptr, c_null_ptr)
       else
          call dangx_sky_set_band(gpu_sky, j-1, bp(j)%nu_c, bp(j)%n, c_loc(bp(j)%nu0), c_loc(bp(j)%tau0))
       end if
    end do
    call dangx_sky_set_tcmb(gpu_sky, T_CMB)
    do i = 1, ncomp
       cc => component_list(i)%p
       call fill_desc(cc, d)
       call dangx_sky_set_component(gpu_sky, i-1, d)
       if (d%type >= DANGX_TEMPLATE) then     ! c%template(0:npix-1,nmaps), c%corr(nbands), c%nfit
          allocate(icorr(nbands)); icorr = merge(1_c_int32_t, 0_c_int32_t, cc%corr)
          ! c%template_amplitudes(nbands,nmaps) is already [map][band] in memory
          call dangx_sky_set_template(gpu_sky, i-1, c_loc(cc%template), c_loc(icorr), cc%nfit, c_loc(cc%template_amplitudes))
          deallocate(icorr)
       end if
    end do
    allocate(gain(nbands), offs(nbands)); gain = ddata%gain; offs = ddata%offset
    call dangx_sky_set_calibration(gpu_sky, c_loc(gain), c_loc(offs))
    call dangx_sky_upload_data(gpu_sky, c_loc(ddata%sig_map), c_loc(ddata%rms_map), c_loc(ddata%masks))
    call dangx_push_state()
  end subroutine dangx_init

  ! host -> device: c%amplitude, c%indices (after the driver changed them, e.g. at start-up)
  subroutine dangx_push_state()
    type(dang_comps), pointer :: cc
    integer :: i
    do i = 1, ncomp
       cc => component_list(i)%p
       if (cc%nindices > 0) then
          call dangx_sky_put_state(gpu_sky, i-1, c_loc(cc%amplitude), c_loc(cc%indices))
       else
          call dangx_sky_put_state(gpu_sky, i-1, c_loc(cc%amplitude), c_null_ptr)
       end if
    end do
  end subroutine dangx_push_state

  ! device -> host: c%amplitude, c%indices, c%template_amplitudes.  Under MPI every rank fills its own pixel range
  ! and the ranges are merged with an all-reduce of (zero elsewhere) arrays, so that rank 0 can write the maps.
  subroutine dangx_pull_state()
    type(dang_comps), pointer :: cc
    integer :: i, ierr_l
    integer(i4b) :: lo, hi
    do i = 1, ncomp
       cc => component_list(i)%p
       if (numprocs > 1) then
          lo = gpu_pix0; hi = gpu_pix0 + int(gpu_sky%npix(1), i4b) - 1
          if (lo > 0) cc%amplitude(0:lo-1, :) = 0.d0
          if (hi < npix-1) cc%amplitude(hi+1:npix-1, :) = 0.d0
          if (cc%nindices > 0) then
             if (lo > 0) cc%indices(0:lo-1, :, :) = 0.d0
             if (hi < npix-1) cc%indices(hi+1:npix-1, :, :) = 0.d0
          end if
       end if
       if (cc%nindices > 0) then
          call dangx_sky_get_state(gpu_sky, i-1, c_loc(cc%amplitude), c_loc(cc%indices))
       else
          call dangx_sky_get_state(gpu_sky, i-1, c_loc(cc%amplitude), c_null_ptr)
       end if
       if (numprocs > 1) then
          call mpi_allreduce(MPI_IN_PLACE, cc%amplitude, size(cc%amplitude), MPI_DOUBLE_PRECISION, MPI_SUM, MPI_COMM_WORLD, ierr_l)
          if (cc%nindices > 0) call mpi_allreduce(MPI_IN_PLACE, cc%indices, size(cc%indices), MPI_DOUBLE_PRECISION, MPI_SUM, &
               MPI_COMM_WORLD, ierr_l)
       end if
    end do
    call pull_template_amplitudes(ddata_offset_only=.false.)
  end subroutine dangx_pull_state

  ! c%template_amplitudes of the global-amplitude components (a few numbers, replicated on every context)
  subroutine pull_template_amplitudes(ddata_offset_only)
    logical, intent(in) :: ddata_offset_only
    type(dang_comps), pointer :: cc
    integer :: i
    do i = 1, ncomp
       cc => component_list(i)%p
       if (ddata_offset_only .and. trim(cc%type) /= 'monopole') cycle
       if (trim(cc%type) == 'template' .or. trim(cc%type) == 'monopole' .or. trim(cc%type) == 'hi_fit') &
            call dangx_check(gpu_sky%ctx(1), dangx_get_template_amplitudes(gpu_sky%ctx(1), i-1, c_loc(cc%template_amplitudes)), &
            'get_template_amplitudes')
    end do
  end subroutine pull_template_amplitudes

  ! update_sky_model's `self%offset = c%template_amplitudes(:,1)` for a monopole (src/dang_data_mod.f90:357-361)
  subroutine refresh_offsets(ddata)
    type(dang_data) :: ddata
    type(dang_comps), pointer :: cc
    integer :: i
    call pull_template_amplitudes(ddata_offset_only=.true.)
    do i = 1, ncomp
       cc => component_list(i)%p
       if (trim(cc%type) == 'monopole') ddata%offset = cc%template_amplitudes(:, 1)
    end do
  end subroutine refresh_offsets

  ! ddata%chisq as update_sky_model + compute_chisq leave it (src/dang_data_mod.f90:339-396, 494-526): from the sums the
  ! index sweeps produced as a by-product when every plane was swept since its last change, otherwise one explicit pass
  subroutine gpu_chisq(ddata, announce)
    type(dang_data) :: ddata
    logical, intent(in) :: announce
    logical :: ok
    integer :: lo, hi
    lo = ddata%pol_type(1); hi = ddata%pol_type(size(ddata%pol_type))
    ddata%chisq = dangx_sky_chisq_cached(gpu_sky, 1, lo, hi, real(nump, c_double), ok)
    if (.not. ok) ddata%chisq = dangx_sky_chisq(gpu_sky, lo, hi, real(nump, c_double))
    ddata%chisq = rank_sum(ddata%chisq)
    if (announce .and. rank == master) call write_stats_gpu(ddata)
  end subroutine gpu_chisq

  ! write_stats_to_term (src/dang_data_mod.f90:528-570): the chi^2 line and the masked index means, from the device
  subroutine write_stats_gpu(ddata)
    type(dang_data) :: ddata
    type(dang_comps), pointer :: cc
    integer :: i, j, k
    write(*,fmt='(a)') '---------------------------------------------'
    write(*,fmt='(i6,a,E16.5)') iter, " - Chisq: ", ddata%chisq
    do i = 1, ncomp
       cc => component_list(i)%p
       do j = 1, cc%nindices
          if (.not. cc%sample_index(j)) cycle
          do k = 1, cc%nflag(j)
             if (iand(cc%pol_flag(j,k),1) .ne. 0) then
                write(*,fmt='(a,a,a,a,a,f12.5)') '     ', trim(cc%label), ' ', trim(cc%ind_label(j)), ' I mean:   ', gpu_index_mean(i, j, 1)
             else if (iand(cc%pol_flag(j,k),2) .ne. 0) then
                write(*,fmt='(a,a,a,a,a,f12.5)') '     ', trim(cc%label), ' ', trim(cc%ind_label(j)), ' Q mean:   ', gpu_index_mean(i, j, 2)
             else if (iand(cc%pol_flag(j,k),4) .ne. 0) then
                write(*,fmt='(a,a,a,a,a,f12.5)') '     ', trim(cc%label), ' ', trim(cc%ind_label(j)), ' U mean:   ', gpu_index_mean(i, j, 3)
             else if (iand(cc%pol_flag(j,k),8) .ne. 0) then
                write(*,fmt='(a,a,a,a,a,f12.5)') '     ', trim(cc%label), ' ', trim(cc%ind_label(j)), ' Q+U mean:   ', gpu_index_mean(i, j, 2)
             end if
          end do
       end do
    end do
    write(*,fmt='(a)') '---------------------------------------------'
  end subroutine write_stats_gpu

  ! mask_avg(c%indices(:,map_n,j), ddata%masks(:,1)) (src/dang_util_mod.f90:186-206) without pulling the map;
  ! comp and j are 1-based as in the reference
  function gpu_index_mean(comp, j, map_n) result(avg)
    integer(i4b), intent(in) :: comp, j, map_n
    real(dp) :: avg, s, stot
    integer(c_int64_t) :: n
    real(dp) :: ntot
    integer :: r
    if (numprocs == 1) then
       avg = dangx_sky_index_mean(gpu_sky, comp-1, j-1, map_n)
    else
       stot = 0.d0; ntot = 0.d0
       do r = 1, gpu_sky%nctx
          call dangx_check(gpu_sky%ctx(r), dangx_index_masked_sum(gpu_sky%ctx(r), comp-1, j-1, map_n, s, n), 'index_masked_sum')
          stot = stot + s; ntot = ntot + n
       end do
       avg = rank_sum(stot)/rank_sum(ntot)
    end if
  end function gpu_index_mean

  subroutine sample_cg_groups_gpu(dpar, ddata)
    ! same signature and effect as sample_cg_groups, src/dang_cg_mod.f90:142-177
    type(dang_data)   :: ddata
    type(dang_params) :: dpar
    integer(i4b) :: i, f, mode
    integer(c_int64_t) :: nbad
    integer(c_int) :: refinements
    real(c_double) :: resid(2)
    logical :: coupled
    mode = merge(DANGX_ML_SAMPLE, DANGX_ML_OPTIMIZE, trim(dpar%ml_mode) == 'sample')
    do i = 1, ncg_groups
       if (cg_groups(i)%p%sample) then
          write(*,fmt='(a,i4)') "Computing a CG search of CG group ", i
          coupled = cg_groups(i)%p%ntemp > 0
          ! DIRECT: per-pixel block solve instead of the CG iteration; groups with template / monopole / hi_fit members:
          ! Schur-complement solve with iterative refinement of the global rows.  (DANGX_SOLVER_CG through dangx_amp_sample
          ! would run the reference's cg_search on the device instead.)
          do f = 1, cg_groups(i)%p%nflag
             call dangx_sky_amp_sample(gpu_sky, i, cg_groups(i)%p%pol_flag(f), mode, DANGX_FLUCT_REFERENCE, gpu_seed, &
                  dangx_stream_id(iter, 0, i, 0, cg_groups(i)%p%pol_flag(f)), nbad)
             if (nbad > 0) write(*,*) 'warning: ', nbad, ' non-SPD pixel blocks left unchanged'
             if (coupled) then
                call dangx_check(gpu_sky%ctx(1), dangx_schur_info(gpu_sky%ctx(1), resid, refinements), 'schur_info')
                write(*,fmt='(a,es10.2,a,i2,a)') '  global rows: |b - A x| / |b| = ', resid(1), ' after ', refinements, ' refinement(s)'
             end if
          end do
          if (coupled) call refresh_offsets(ddata)       ! update_sky_model: offset <- monopole amplitudes
          call gpu_chisq(ddata, .true.)                   ! update_sky_model + write_stats_to_term, :172-173
       end if
    end do
  end subroutine sample_cg_groups_gpu

  subroutine sample_spectral_parameters_gpu(dpar, ddata)
    ! same signature and effect as sample_spectral_parameters, src/dang_sample_mod.f90:21-86
    type(dang_data)   :: ddata
    type(dang_params) :: dpar
    type(dang_comps), pointer :: cc
    integer(i4b) :: i, j, k, map_n, mode
    integer(c_int64_t) :: nacc
    real(c_double), target :: tpeek(2)
    type(dangx_comp_desc) :: d
    logical(lgt) :: sampled, pair_done, pairable
    sampled = .false.
    mode = merge(DANGX_ML_SAMPLE, DANGX_ML_OPTIMIZE, trim(dpar%ml_mode) == 'sample')
    do i = 1, ncomp
       cc => component_list(i)%p
       if (cc%nindices == 0) cycle
       if (.not. any(cc%sample_index)) cycle
       sampled = .true.
       pair_done = .false.
       do j = 1, cc%nindices
          if (pair_done) then                            ! this index went with the one before it (one launch)
             pair_done = .false.
             cycle
          end if
          if (.not. cc%sample_index(j)) cycle
          do k = 1, cc%nflag(j)
             if (iand(cc%pol_flag(j,k),1) .ne. 0) then
                map_n = 1
             else if (iand(cc%pol_flag(j,k),2) .ne. 0) then
                map_n = 2
             else if (iand(cc%pol_flag(j,k),4) .ne. 0) then
                map_n = 3
             else if (iand(cc%pol_flag(j,k),8) .ne. 0) then
                map_n = -1
             else
                write(*,*) "There is something wrong with the poltype flag"
                cycle
             end if
             if (cc%index_mode(j) == 1) then
                write(*,*) 'Sampling fullsky'
                if (cc%sample_nside(j) /= nside .and. (gpu_sky%nctx > 1 .or. numprocs > 1)) then
                   write(*,*) 'dang_gpu_mod: sample_nside /= nside needs the whole sky in one context (dangx_init(..., ngpu=1))'
                   stop
                end if
                call sample_index_mh_fullsky_gpu(cc, i-1, j, map_n)
             else if (cc%sample_nside(j) /= nside) then
                write(*,fmt='(a,i4)') 'Sampling per-pixel at nside ', cc%sample_nside(j)
                if (numprocs > 1) then      ! one context per process: the sums over the shards go through dang_allreduce
                   call dangx_check(gpu_sky%ctx(1), dangx_index_sample_coarse(gpu_sky%ctx(1), i-1, j-1, map_n, nsample, mode, &
                        gpu_seed, dangx_stream_id(iter, 1, i-1, j-1, cc%pol_flag(j,k)), nside, cc%sample_nside(j), nacc), &
                        'index_sample_coarse')
                else                        ! one process, one or several contexts
                   call dangx_sky_index_sample_coarse(gpu_sky, i-1, j-1, map_n, nsample, mode, gpu_seed, &
                        dangx_stream_id(iter, 1, i-1, j-1, cc%pol_flag(j,k)), nside, cc%sample_nside(j), nacc)
                end if
             else
                write(*,fmt='(a,i4)') 'Sampling per-pixel at nside ', cc%sample_nside(j)
                if (.not. cc%tuned(j)) then              ! 'Tuning!', src/dang_sample_mod.f90:341-346
                   write(*,*) 'Tuning!'
                   call tune_perpixel_gpu(cc, i-1, j, map_n)
                   call fill_desc(cc, d)
                   call dangx_sky_set_component(gpu_sky, i-1, d)   ! the new step size
                end if
                ! two consecutive plain per-pixel sweeps of this component on the same planes (dust beta, dust T): one entry
                ! point, one kernel launch where the register chain covers both -- the same numbers as the two sweeps
                pairable = .false.
                if (j < cc%nindices .and. cc%nflag(j) == 1) then
                   if (cc%sample_index(j+1) .and. cc%nflag(j+1) == 1) then
                      pairable = cc%pol_flag(j+1,1) == cc%pol_flag(j,k) .and. cc%index_mode(j+1) /= 1 .and. &
                           cc%sample_nside(j+1) == nside .and. cc%tuned(j+1)
                   end if
                end if
                if (pairable) then
                   write(*,fmt='(a,i4)') 'Sampling per-pixel at nside ', cc%sample_nside(j+1)
                   call dangx_sky_index_sample_pair(gpu_sky, i-1, j-1, map_n, nsample, mode, gpu_seed, &
                        dangx_stream_id(iter, 1, i-1, j-1, cc%pol_flag(j,k)), dangx_stream_id(iter, 1, i-1, j, cc%pol_flag(j,k)))
                   pair_done = .true.
                else
                   call dangx_sky_index_sample(gpu_sky, i-1, j-1, map_n, nsample, mode, gpu_seed, &
                        dangx_stream_id(iter, 1, i-1, j-1, cc%pol_flag(j,k)))
                end if
             end if
          end do
       end do
       ! "Update the global variable T_CMB" (src/dang_sample_mod.f90:75-78): pixel 0 lives on the first shard
       if (trim(cc%type) == 'T_cmb') then
          tpeek = 0.d0
          if (gpu_pix0 == 0) call dangx_sky_peek_first(gpu_sky, i-1, 1, tpeek)
          T_CMB = rank_sum(tpeek(1))
          call dangx_sky_set_tcmb(gpu_sky, T_CMB)
       end if
    end do
    if (sampled) call gpu_chisq(ddata, .true.)            ! update_sky_model + write_stats_to_term, :81-84
  end subroutine sample_spectral_parameters_gpu

  ! ---- the sky-wide chain of index_mode == 1 and of the tuner: the reference's own statements with the three
  ! evaluate_* / eval_jeffreys_prior calls replaced by device passes
  function fullsky_lnl(cc, nind, sp, th) result(v)
    type(dang_comps), pointer :: cc
    integer(i4b), intent(in)  :: nind, sp
    real(c_double), intent(in) :: th(2)
    real(dp) :: v
    real(c_double) :: rows(4*nbands)
    integer(i4b) :: q
    v = 0.d0
    if (trim(cc%lnl_type(nind)) == 'chisq') then                       ! evaluate_lnL, src/dang_lnl_mod.f90:126-182
       call dangx_sky_fullsky_sums(gpu_sky, 0, th, rows, 1)
       v = rank_sum(rows(1))
    else if (trim(cc%lnl_type(nind)) == 'marginal') then              ! evaluate_marginal_lnL, :47-124: j outer, k inner
       call dangx_sky_fullsky_sums(gpu_sky, 1, th, rows, 2*nbands*sp)
       do q = 1, nbands*sp
          v = v - 0.5d0*rank_sum(rows(2*q-1))*(1.d0/rank_sum(rows(2*q)))*rank_sum(rows(2*q-1))
       end do
    end if
  end function fullsky_lnl

  function fullsky_prior(cc, nind, val) result(v)
    type(dang_comps), pointer :: cc
    integer(i4b), intent(in)  :: nind
    real(dp), intent(in)      :: val
    real(dp) :: v
    real(c_double) :: th(2), rows(1)
    v = 0.d0
    if (trim(cc%prior_type(nind)) == 'gaussian') then
       v = log(eval_normal_prior(val, cc%gauss_prior(nind,1), cc%gauss_prior(nind,2)))
    else if (trim(cc%prior_type(nind)) == 'jeffreys') then             ! eval_jeffreys_prior, src/dang_lnl_mod.f90:242-304
       th = [val, 0.d0]
       call dangx_sky_fullsky_sums(gpu_sky, 2, th, rows, 1)
       v = log(sqrt(rank_sum(rows(1))))
    end if
  end function fullsky_prior

  ! tune_spectral_parameter_length, src/dang_sample_mod.f90:623-717, on data prepared by dangx_sky_fullsky_prepare
  subroutine tune_gpu(cc, nind, sp, theta_init)
    type(dang_comps), pointer :: cc
    integer(i4b), intent(in)  :: nind, sp
    real(dp), intent(in)      :: theta_init(2)
    real(c_double) :: theta(2), sample(2)
    real(dp) :: accept, lnl, lnl_new, lnl_old, diff, ratio, num
    integer(i4b) :: l
    lnl = 0.d0; lnl_new = 0.d0; lnl_old = 0.d0
    sample = theta_init; theta = theta_init
    if (trim(cc%lnl_type(nind)) == 'prior') then
       sample(nind) = rand_normal(cc%gauss_prior(nind,1), cc%gauss_prior(nind,2))
    else
       lnl = fullsky_lnl(cc, nind, sp, sample)
    end if
    if (trim(cc%prior_type(nind)) == 'gaussian') then
       lnl_old = lnl + log(eval_normal_prior(sample(nind), cc%gauss_prior(nind,1), cc%gauss_prior(nind,2)))
    else if (trim(cc%prior_type(nind)) == 'uniform') then
       lnl_old = lnl
    end if
    do while (.not. cc%tuned(nind))
       accept = 0.d0
       do l = 1, nsample
          theta(nind) = sample(nind) + rand_normal(0.d0, cc%step_size(nind))
          if (theta(nind) .lt. cc%uni_prior(nind,1) .or. theta(nind) .gt. cc%uni_prior(nind,2)) cycle
          if (trim(cc%lnl_type(nind)) == 'chisq' .or. trim(cc%lnl_type(nind)) == 'marginal') lnl = fullsky_lnl(cc, nind, sp, theta)
          if (trim(cc%prior_type(nind)) == 'gaussian') then
             lnl_new = lnl + log(eval_normal_prior(theta(nind), cc%gauss_prior(nind,1), cc%gauss_prior(nind,2)))
          else if (trim(cc%prior_type(nind)) == 'uniform') then
             lnl_new = lnl
          end if
          diff  = lnl_new - lnl_old
          ratio = exp(diff)
          if (trim(ml_mode) == 'optimize') then
             if (ratio > 1.d0) then
                sample(nind) = theta(nind); lnl_old = lnl_new; accept = accept + 1
             end if
          else if (trim(ml_mode) == 'sample') then
             call RANDOM_NUMBER(num)
             if (ratio > num) then
                sample(nind) = theta(nind); lnl_old = lnl_new; accept = accept + 1
             end if
          end if
          lnl = 0.d0
       end do
       if (accept/l .lt. 0.4) then
          cc%step_size(nind) = cc%step_size(nind) - 0.5*cc%step_size(nind)
       else if (accept/l .gt. 0.6) then
          cc%step_size(nind) = cc%step_size(nind) + 0.5*cc%step_size(nind)
       else
          cc%tuned = .true.
       end if
       write(*,*) accept/l, cc%step_size(nind)
    end do
  end subroutine tune_gpu

  ! the 'Tuning!' block of the per-pixel branch, src/dang_sample_mod.f90:341-346 (comp0 0-based, nind 1-based)
  subroutine tune_perpixel_gpu(cc, comp0, nind, map_n)
    type(dang_comps), pointer :: cc
    integer(i4b), intent(in)  :: comp0, nind, map_n
    real(dp) :: sample(2)
    integer(i4b) :: l, s1, sp
    s1 = merge(2, map_n, map_n == -1); sp = merge(2, 1, map_n == -1)
    call dangx_sky_fullsky_prepare(gpu_sky, comp0, map_n)             ! data_raw minus every other component, :173-196
    sample = 0.d0
    do l = 1, cc%nindices
       ! sum(c%indices(:,map_inds(1),l))/sum(mask(:,1)): every pixel, the mask's values
       sample(l) = dangx_sky_index_plain_mean(gpu_sky, comp0, l-1, s1)
       call tune_gpu(cc, nind, sp, sample)
    end do
  end subroutine tune_perpixel_gpu

  ! sample_index_mh, index_mode == 1 (src/dang_sample_mod.f90:229-329).  comp0 is 0-based, nind 1-based.
  subroutine sample_index_mh_fullsky_gpu(cc, comp0, nind, map_n)
    type(dang_comps), pointer  :: cc
    integer(i4b), intent(in)   :: comp0, nind, map_n
    real(c_double), target     :: sample(2), theta(2)
    real(dp)                   :: lnl, lnl_old, lnl_new, diff, ratio, num
    integer(i4b)               :: l, s1, sp
    logical(lgt)               :: sample_it
    type(dangx_comp_desc)      :: d

    s1 = merge(2, map_n, map_n == -1); sp = merge(2, 1, map_n == -1)
    if (cc%sample_nside(nind) /= nside) then                                                        ! :173-217, degraded maps
       call dangx_check(gpu_sky%ctx(1), dangx_fullsky_prepare_coarse(gpu_sky%ctx(1), comp0, map_n, nside, cc%sample_nside(nind)), &
            'fullsky_prepare_coarse')
    else
       call dangx_sky_fullsky_prepare(gpu_sky, comp0, map_n)                                       ! :173-196
    end if
    sample = 0.d0
    if (gpu_pix0 == 0) call dangx_sky_peek_first(gpu_sky, comp0, s1, sample)                       ! :240-242
    sample(1) = rank_sum(sample(1)); sample(2) = rank_sum(sample(2))
    theta = sample
    lnl = 0.d0; sample_it = .true.
    if (trim(cc%lnl_type(nind)) == 'prior') then                                                    ! :255-257
       sample_it = .false.
       sample(nind) = rand_normal(cc%gauss_prior(nind,1), cc%gauss_prior(nind,2))
    else
       lnl = fullsky_lnl(cc, nind, sp, sample)
    end if
    lnl_old = lnl + fullsky_prior(cc, nind, sample(nind))
    if (sample_it) then
       if (.not. cc%tuned(nind)) then                                                               ! :272-275
          call tune_gpu(cc, nind, sp, sample)
          call fill_desc(cc, d)
          call dangx_sky_set_component(gpu_sky, comp0, d)
       end if
       do l = 1, nsample                                                                            ! :282-324
          theta(nind) = sample(nind) + rand_normal(0.d0, cc%step_size(nind))
          if (theta(nind) < cc%uni_prior(nind,1) .or. theta(nind) > cc%uni_prior(nind,2)) cycle
          lnl_new = fullsky_lnl(cc, nind, sp, theta) + fullsky_prior(cc, nind, theta(nind))
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
    call dangx_sky_fill_index(gpu_sky, comp0, nind-1, map_n, sample(nind))                          ! :329, :483
  end subroutine sample_index_mh_fullsky_gpu

  ! sample_calibrators + fit_band_gain (src/dang_sample_mod.f90:487-518, 570-621): the two sky-wide sums come from the
  ! device, the draw stays here, the new gains go back with dangx_set_calibration
  subroutine sample_calibrators_gpu(ddata)
    type(dang_data), intent(inout) :: ddata
    real(c_double), allocatable, target :: gain(:), offs(:)
    real(dp) :: mu, sigma
    integer(i4b) :: j
    logical(lgt) :: sampled
    sampled = .false.
    if (any(ddata%fit_gain(:))) then
       write(*,*) "Sampling band calibrators"
       sampled = .true.
    end if
    do j = 1, nbands
       if (ddata%fit_gain(j)) then
          call dangx_sky_gain_sums(gpu_sky, j-1, mu, sigma)
          mu = rank_sum(mu); sigma = rank_sum(sigma)
          mu = mu / sigma
          sigma = sqrt(1.d0 / sigma)
          if (trim(ml_mode) == 'optimize') then
             ddata%gain(j) = mu
          else
             ddata%gain(j) = mu + sigma * rand_normal(0.d0, 1.d0)
          end if
          ! each band's fit sees the gains already drawn for the bands before it only through res_map of ITS OWN band,
          ! so one upload after the loop would do; uploading per band keeps the device state equal to the host's
          allocate(gain(nbands), offs(nbands)); gain = ddata%gain; offs = ddata%offset
          call dangx_sky_set_calibration(gpu_sky, c_loc(gain), c_loc(offs))
          deallocate(gain, offs)
       end if
    end do
    if (sampled) call gpu_chisq(ddata, .true.)            ! update_sky_model + write_stats_to_term, :513-516
  end subroutine sample_calibrators_gpu

  ! before ddata%write_maps(dpar) (src/dang.f90:119-121): everything write_maps reads from host arrays --
  ! c%amplitude, c%indices, c%template_amplitudes, ddata%sky_model / res_map / chi_map / chisq / offset
  ! (src/dang_data_mod.f90:573-664; its own compute_chisq then finds a current sky model)
  subroutine dangx_refresh_host_state(ddata)
    type(dang_data), target :: ddata
    integer :: ierr_l
    integer(i4b) :: lo, hi
    call dangx_pull_state()
    call refresh_offsets(ddata)
    if (numprocs > 1) then
       lo = gpu_pix0; hi = gpu_pix0 + int(gpu_sky%npix(1), i4b) - 1
       ddata%sky_model = 0.d0; ddata%res_map = 0.d0
    end if
    ddata%chi_map = 0.d0
    ddata%chisq = dangx_sky_chisq(gpu_sky, ddata%pol_type(1), ddata%pol_type(size(ddata%pol_type)), real(nump, c_double), &
         c_loc(ddata%sky_model), c_loc(ddata%res_map), c_loc(ddata%chi_map))
    ddata%chisq = rank_sum(ddata%chisq)
    if (numprocs > 1) then
       call mpi_allreduce(MPI_IN_PLACE, ddata%sky_model, size(ddata%sky_model), MPI_DOUBLE_PRECISION, MPI_SUM, MPI_COMM_WORLD, ierr_l)
       call mpi_allreduce(MPI_IN_PLACE, ddata%res_map, size(ddata%res_map), MPI_DOUBLE_PRECISION, MPI_SUM, MPI_COMM_WORLD, ierr_l)
       call mpi_allreduce(MPI_IN_PLACE, ddata%chi_map, size(ddata%chi_map), MPI_DOUBLE_PRECISION, MPI_SUM, MPI_COMM_WORLD, ierr_l)
    end if
  end subroutine dangx_refresh_host_state

  ! write_data (src/dang_data_mod.f90:666-761) with the numbers taken from the device: same files, same formats.
  ! Every iteration; moves no map.
  subroutine write_data_gpu(ddata, dpar, map_n)
    type(dang_data)           :: ddata
    type(dang_params)         :: dpar
    integer(i4b), intent(in)  :: map_n
    type(dang_comps), pointer :: cc
    integer(i4b)              :: j, n, unit
    character(len=128)        :: fmt
    character(len=512)        :: fname
    character(len=1)          :: nmaps_str
    character(len=4)          :: nband_str
    character(len=5)          :: it_str
    real(dp), allocatable     :: means(:,:)

    call gpu_chisq(ddata, .false.)                       ! write_data's compute_chisq(self), :694
    call pull_template_amplitudes(ddata_offset_only=.false.)
    allocate(means(2, ncomp)); means = 0.d0              ! sky-wide sums: every rank takes part, the master writes
    do n = 1, ncomp
       cc => component_list(n)%p
       do j = 1, cc%nindices
          if (cc%sample_index(j)) means(j, n) = gpu_index_mean(n, j, map_n)   ! mask_avg(c%indices(:,map_n,j), self%masks(:,1))
       end do
    end do
    if (rank /= master) return
    write(*,*) 'Output data files'
    write(nband_str, '(i4)') nbands
    write(it_str, '(i0.5)') iter
    write(nmaps_str, '(i1)') nmaps

    fname = trim(dpar%outdir) // 'total_chisq_' // trim(tqu(map_n)) // '.dat'
    inquire(file=fname, exist=exist)
    if (exist) then
       open(33, file=fname, status="old", position="append", action="write")
    else
       open(33, file=fname, status="new", action="write")
    end if
    write(33,*) ddata%chisq
    close(33)

    do n = 1, ncomp
       cc => component_list(n)%p
       if (trim(cc%type) == 'template' .or. trim(cc%type) == 'hi_fit') then
          unit = getlun()
          fname = trim(dpar%outdir) // trim(cc%label) // '_' // trim(tqu(map_n)) // '_amplitudes.dat'
          inquire(file=fname, exist=exist)
          if (exist) then
             open(unit, file=fname, status="old", position="append", action="write")
          else
             open(unit, file=fname, status="new", action="write")
             write(unit, fmt='('//trim(nband_str)//'(A17))') ddata%label
          end if
          write(unit, fmt='('//trim(nband_str)//'(E17.8))') cc%template_amplitudes(:,map_n)/cc%temp_norm(map_n)
          close(unit)
       end if
       do j = 1, cc%nindices
          if (cc%sample_index(j)) then
             fmt = '('//nmaps_str//'(f12.8))'
             unit = getlun()
             fname = trim(dpar%outdir) // trim(cc%label) // '_' // trim(cc%ind_label(j)) // '_mean_' // trim(tqu(map_n)) // '.dat'
             inquire(file=fname, exist=exist)
             if (exist) then
                open(unit, file=fname, status="old", position="append", action="write")
             else
                open(unit, file=fname, status="new", action="write")
             end if
             write(unit, fmt=fmt) means(j, n)
             close(unit)
          end if
       end do
    end do

    fmt = '(a12,E16.8)'
    fname = trim(dpar%outdir)//'band_gains_k'//it_str//'.dat'
    inquire(file=fname, exist=exist)
    if (exist) then
       open(37, file=fname, status="old", position="append", action="write")
    else
       open(37, file=fname, status="new", action="write")
    end if
    do j = 1, nbands
       write(37, fmt=fmt) trim(ddata%label(j)), ddata%gain(j)
    end do
    close(37)

    fname = trim(dpar%outdir)//'band_offsets_k'//it_str//'.dat'
    inquire(file=fname, exist=exist)
    if (exist) then
       open(38, file=fname, status="old", position="append", action="write")
    else
       open(38, file=fname, status="new", action="write")
    end if
    do j = 1, nbands
       write(38, fmt=fmt) trim(ddata%label(j)), ddata%offset(j)/ddata%conversion(j)
    end do
    close(38)
  end subroutine write_data_gpu

end module dang_gpu_mod
