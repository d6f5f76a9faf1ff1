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
    integer(c_int64_t), allocatable :: bnds(:)
    integer :: i, j, nctx, r

    call dangx_check(c_null_ptr, dangx_device_count(ndev), 'dangx_device_count')
    nctx = ndev
    if (present(ngpu)) nctx = ngpu
    if (numprocs > 1) then
       ! one context on this rank's range [p0, p0+np) of the sky: a one-context sky whose window starts at p0
       allocate(bnds(0:numprocs))                           ! shards of equal work (unmasked pixels), the same on every rank
       call dangx_balanced_bounds(ddata%masks(:,1), numprocs, bnds)
       p0 = bnds(rank); np = bnds(rank+1) - bnds(rank)
       deallocate(bnds)
       gpu_pix0 = int(p0, i4b)
       gpu_sky%nctx = 1; gpu_sky%npix_global = npix; gpu_sky%nmaps = nmaps; gpu_sky%nbands = nbands; gpu_sky%ncomp = ncomp
       gpu_sky%pix0(1) = p0; gpu_sky%npix(1) = np
       call dangx_check(c_null_ptr, dangx_create(gpu_sky%ctx(1), dangx_dims(int(np, c_int32_t), nmaps, nbands, ncomp, p0, &
            int(npix, c_int64_t), mod(rank, ndev), 0)), 'dangx_create')
       call dangx_check(gpu_sky%ctx(1), dangx_set_host_stride(gpu_sky%ctx(1), int(npix, c_int64_t)), 'dangx_set_host_stride')
       call dangx_check(gpu_sky%ctx(1), dangx_set_allreduce(gpu_sky%ctx(1), c_funloc(dang_allreduce), c_null_ptr, &
            merge(1, 0, rank == master)), 'dangx_set_allreduce')
       ! the sky-wide chains (full-sky index mode, tuner, gain draw) run behind the ABI on EVERY rank with keyed random
       ! numbers: same draws everywhere, nothing to seed here
    else
       call dangx_sky_create(gpu_sky, int(npix, c_int64_t), nmaps, nbands, ncomp, nctx, mask=ddata%masks(:,1))
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

  ! ddata%chisq as update_sky_model + compute_chisq leave it (src/dang_data_mod.f90:339-396, 494-526): plane by plane from the
  ! sums the index sweeps left behind, or from one explicit pass over a plane that has changed since (dangx_chisq_current)
  subroutine gpu_chisq(ddata, announce)
    type(dang_data) :: ddata
    logical, intent(in) :: announce
    integer :: lo, hi
    lo = ddata%pol_type(1); hi = ddata%pol_type(size(ddata%pol_type))
    ddata%chisq = rank_sum(dangx_sky_chisq_current(gpu_sky, lo, hi, real(nump, c_double)))
    if (announce .and. rank == master) call write_stats_gpu(ddata)
  end subroutine gpu_chisq

  ! write_stats_to_term (src/dang_data_mod.f90:528-570): the chi^2 line and the masked index means -- every mean of the list
  ! from ONE device reduction
  subroutine write_stats_gpu(ddata)
    type(dang_data) :: ddata
    type(dang_comps), pointer :: cc
    integer(c_int32_t), allocatable :: lc(:), ln(:), lk(:)
    character(len=48), allocatable  :: what(:)
    real(c_double), allocatable     :: avg(:)
    integer :: i, j, k, n, f
    n = 0
    do i = 1, ncomp                                       ! one entry per sampled (index, flag): sized from the model
       cc => component_list(i)%p
       do j = 1, cc%nindices
          if (cc%sample_index(j)) n = n + cc%nflag(j)
       end do
    end do
    allocate(lc(n), ln(n), lk(n), what(n), avg(n))
    n = 0
    do i = 1, ncomp
       cc => component_list(i)%p
       do j = 1, cc%nindices
          if (.not. cc%sample_index(j)) cycle
          do k = 1, cc%nflag(j)
             f = cc%pol_flag(j,k)
             if (iand(f, 15) == 0) cycle
             n = n + 1
             lc(n) = i-1; ln(n) = j-1
             if (iand(f,1) .ne. 0) then
                lk(n) = 1; what(n) = trim(cc%label)//' '//trim(cc%ind_label(j))//' I mean:   '
             else if (iand(f,2) .ne. 0) then
                lk(n) = 2; what(n) = trim(cc%label)//' '//trim(cc%ind_label(j))//' Q mean:   '
             else if (iand(f,4) .ne. 0) then
                lk(n) = 3; what(n) = trim(cc%label)//' '//trim(cc%ind_label(j))//' U mean:   '
             else
                lk(n) = 2; what(n) = trim(cc%label)//' '//trim(cc%ind_label(j))//' Q+U mean:   '
             end if
          end do
       end do
    end do
    if (n > 0) call gpu_index_means(n, lc, ln, lk, avg)
    write(*,fmt='(a)') '---------------------------------------------'
    write(*,fmt='(i6,a,E16.5)') iter, " - Chisq: ", ddata%chisq
    do i = 1, n
       write(*,fmt='(a,a,f12.5)') '     ', what(i)(1:len_trim(what(i))+3), avg(i)
    end do
    write(*,fmt='(a)') '---------------------------------------------'
  end subroutine write_stats_gpu

  ! mask_avg(c%indices(:,map_n,j), ddata%masks(:,1)) (src/dang_util_mod.f90:186-206) for a LIST of index maps without pulling a
  ! map: comp0 / nind0 0-based.  One launch and one wait per context; under MPI the sums and counts are added over the ranks.
  subroutine gpu_index_means(n, comp0, nind0, map_n, avg)
    integer, intent(in) :: n
    integer(c_int32_t), intent(in) :: comp0(n), nind0(n), map_n(n)
    real(c_double), intent(out) :: avg(n)
    real(c_double) :: s(n)
    integer(c_int64_t) :: cnt(n)
    integer :: e, b0, nb
    do b0 = 1, n, 16                    ! dangx_index_masked_sums takes at most 16 maps per call: any number, in batches
       nb = min(16, n - b0 + 1)
       if (numprocs == 1) then
          call dangx_sky_index_means(gpu_sky, nb, comp0(b0:b0+nb-1), nind0(b0:b0+nb-1), map_n(b0:b0+nb-1), avg(b0:b0+nb-1))
       else
          call dangx_check(gpu_sky%ctx(1), dangx_index_masked_sums(gpu_sky%ctx(1), nb, comp0(b0:b0+nb-1), nind0(b0:b0+nb-1), &
               map_n(b0:b0+nb-1), s(b0:b0+nb-1), cnt(b0:b0+nb-1)), 'index_masked_sums')
          do e = b0, b0 + nb - 1
             avg(e) = rank_sum(s(e))/rank_sum(real(cnt(e), dp))
          end do
       end if
    end do
  end subroutine gpu_index_means

  integer(i4b) function map_of_flag(flag)
    ! src/dang_sample_mod.f90:53-64: T -> 1, Q -> 2, U -> 3, Q+U -> -1; 0 = "something wrong with the poltype flag"
    integer(i4b), intent(in) :: flag
    if (iand(flag,1) .ne. 0) then
       map_of_flag = 1
    else if (iand(flag,2) .ne. 0) then
       map_of_flag = 2
    else if (iand(flag,4) .ne. 0) then
       map_of_flag = 3
    else if (iand(flag,8) .ne. 0) then
       map_of_flag = -1
    else
       map_of_flag = 0
    end if
  end function map_of_flag

  integer(i4b) function planes_of_flag(flag)
    integer(i4b), intent(in) :: flag                       ! bit set of the map planes a poltype flag works on
    planes_of_flag = 7
    if (flag == 1) planes_of_flag = 1
    if (flag == 2) planes_of_flag = 2
    if (flag == 4) planes_of_flag = 4
    if (flag == 8) planes_of_flag = 6
  end function planes_of_flag

  ! an ordinary per-pixel sweep at the map resolution with nothing to tune first
  logical function plain_sweep(cc, j)
    type(dang_comps), intent(in) :: cc
    integer(i4b), intent(in)     :: j
    plain_sweep = cc%index_mode(j) /= 1 .and. cc%sample_nside(j) == nside .and. cc%tuned(j)
  end function plain_sweep

  ! ---- the amplitude phase.  owner(e) = p: sweep number e of the list (sw_*) is issued together with the p-th (group, flag)
  ! pass, through dangx_plane_set_sample -- the solve and the sweeps on its planes in one launch where the model allows it.
  subroutine run_solves(dpar, ddata, per_group_stats, owner, sw_comp, sw_nind)
    type(dang_data)   :: ddata
    type(dang_params) :: dpar
    logical, intent(in) :: per_group_stats
    integer(c_int32_t), intent(in), optional :: owner(:), sw_comp(:), sw_nind(:)
    integer(c_int32_t), allocatable :: lc(:), ln(:)
    integer(c_int64_t), allocatable :: ls(:)
    integer(i4b) :: i, f, mode, p, e, q, nl, flag, nullity
    integer(c_int64_t) :: nbad
    integer(c_int) :: refinements
    real(c_double) :: resid(2)
    logical :: coupled
    mode = merge(DANGX_ML_SAMPLE, DANGX_ML_OPTIMIZE, trim(dpar%ml_mode) == 'sample')
    p = 0
    do i = 1, ncg_groups
       if (.not. cg_groups(i)%p%sample) cycle
       write(*,fmt='(a,i4)') "Computing a CG search of CG group ", i
       coupled = cg_groups(i)%p%ntemp > 0
       ! DIRECT: per-pixel block solve instead of the CG iteration; groups with template / monopole / hi_fit members:
       ! Schur-complement solve over all contexts with iterative refinement of the global rows
       do f = 1, cg_groups(i)%p%nflag
          flag = cg_groups(i)%p%pol_flag(f)
          p = p + 1
          nl = 0
          if (present(owner)) nl = count(owner == p)
          if (nl > 0) then
             allocate(lc(nl), ln(nl), ls(nl))
             q = 0
             do e = 1, size(owner)
                if (owner(e) /= p) cycle
                q = q + 1
                lc(q) = sw_comp(e); ln(q) = sw_nind(e)
                ls(q) = dangx_stream_id(iter, 1, int(sw_comp(e)), int(sw_nind(e)), flag)
             end do
             if (coupled) then   ! the Schur solve's account, as on the two-call path (the call waits for the small system anyway)
                call dangx_sky_plane_set_sample(gpu_sky, i, flag, mode, DANGX_FLUCT_REFERENCE, gpu_seed, dangx_stream_id(iter, 0, i, 0, flag), &
                     nl, lc, ln, ls, nsample, gpu_seed, n_not_spd=nbad, nullity=nullity)
                if (nbad > 0) write(*,*) 'warning: ', nbad, ' non-SPD pixel blocks left unchanged'
                call dangx_check(gpu_sky%ctx(1), dangx_schur_info(gpu_sky%ctx(1), resid, refinements), 'schur_info')
                write(*,fmt='(a,es10.2,a,i2,a)') '  global rows: |b - A x| / |b| = ', resid(1), ' after ', refinements, ' refinement(s)'
             else
                call dangx_sky_plane_set_sample(gpu_sky, i, flag, mode, DANGX_FLUCT_REFERENCE, gpu_seed, dangx_stream_id(iter, 0, i, 0, flag), &
                     nl, lc, ln, ls, nsample, gpu_seed)
             end if
             deallocate(lc, ln, ls)
          else if (coupled .or. per_group_stats) then     ! with the count: the chi^2 pass below waits for the device anyway
             call dangx_sky_amp_sample(gpu_sky, i, flag, mode, DANGX_FLUCT_REFERENCE, gpu_seed, &
                  dangx_stream_id(iter, 0, i, 0, flag), nbad, nullity)
             if (nbad > 0) write(*,*) 'warning: ', nbad, ' non-SPD pixel blocks left unchanged'
             if (coupled) then
                call dangx_check(gpu_sky%ctx(1), dangx_schur_info(gpu_sky%ctx(1), resid, refinements), 'schur_info')
                write(*,fmt='(a,es10.2,a,i2,a)') '  global rows: |b - A x| / |b| = ', resid(1), ' after ', refinements, ' refinement(s)'
             end if
          else
             call dangx_sky_amp_sample(gpu_sky, i, flag, mode, DANGX_FLUCT_REFERENCE, gpu_seed, dangx_stream_id(iter, 0, i, 0, flag))
          end if
       end do
       if (coupled) call refresh_offsets(ddata)           ! update_sky_model: offset <- monopole amplitudes
       if (per_group_stats) call gpu_chisq(ddata, .true.) ! update_sky_model + write_stats_to_term, :172-173
    end do
  end subroutine run_solves

  subroutine sample_cg_groups_gpu(dpar, ddata)
    ! same signature and effect as sample_cg_groups, src/dang_cg_mod.f90:142-177 (chi^2 and the index means printed after
    ! every group, which costs one pass over the maps per group: gibbs_iteration_gpu prints them once per iteration)
    type(dang_data)   :: ddata
    type(dang_params) :: dpar
    call run_solves(dpar, ddata, .true.)
  end subroutine sample_cg_groups_gpu

  ! ---- the index phase: the loops of sample_spectral_parameters (src/dang_sample_mod.f90:32-79); sweep number e of the
  ! iteration is skipped when skip(e) is set (it went with its group's solve)
  subroutine run_sweeps(dpar, ddata, skip, sampled)
    type(dang_data)   :: ddata
    type(dang_params) :: dpar
    logical, intent(in), optional :: skip(:)
    logical(lgt), intent(out)     :: sampled
    type(dang_comps), pointer :: cc
    integer(i4b) :: i, j, k, e, map_n, mode, flag, ns, n, q
    integer(c_int64_t) :: nacc, stream
    integer(c_int32_t) :: tuned(2)
    real(c_double) :: step, val
    logical(lgt) :: pair_done, pairable, skip_next
    integer(c_int32_t), allocatable :: fc(:), fn(:), ff(:), fg(:)
    integer(c_int64_t), allocatable :: fs(:)
    logical, allocatable :: fplain(:), done(:)
    sampled = .false.
    mode = merge(DANGX_ML_SAMPLE, DANGX_ML_OPTIMIZE, trim(dpar%ml_mode) == 'sample')
    ! the sweeps of the iteration as a flat list in the loop's order: consecutive plain per-pixel sweeps with ONE flag whose
    ! components belong to one CG group go through dangx_plane_sweeps_sample -- one launch on the plane set where the model allows it
    ns = 0
    do i = 1, ncomp
       cc => component_list(i)%p
       do j = 1, cc%nindices
          if (cc%sample_index(j)) ns = ns + cc%nflag(j)
       end do
    end do
    allocate(fc(ns), fn(ns), ff(ns), fg(ns), fs(ns), fplain(ns), done(ns))
    ns = 0
    do i = 1, ncomp
       cc => component_list(i)%p
       do j = 1, cc%nindices
          if (.not. cc%sample_index(j)) cycle
          do k = 1, cc%nflag(j)
             ns = ns + 1
             fc(ns) = i-1; fn(ns) = j-1; ff(ns) = cc%pol_flag(j,k); fg(ns) = cc%cg_group
             fs(ns) = dangx_stream_id(iter, 1, i-1, j-1, cc%pol_flag(j,k))
             fplain(ns) = plain_sweep(cc, j) .and. cc%sample_amplitude .and. map_of_flag(cc%pol_flag(j,k)) /= 0
             if (present(skip)) fplain(ns) = fplain(ns) .and. .not. skip(ns)
          end do
       end do
    end do
    done = .false.
    e = 0
    do i = 1, ncomp
       cc => component_list(i)%p
       if (cc%nindices == 0) cycle
       if (.not. any(cc%sample_index)) cycle
       sampled = .true.
       pair_done = .false.
       do j = 1, cc%nindices
          if (.not. cc%sample_index(j)) cycle
          do k = 1, cc%nflag(j)
             e = e + 1
             if (done(e)) then                            ! went with an earlier sweep of its plane set (one launch)
                write(*,fmt='(a,i4)') 'Sampling per-pixel at nside ', cc%sample_nside(j)
                cycle
             end if
             if (pair_done) then                          ! this index went with the one before it (one launch)
                pair_done = .false.
                cycle
             end if
             if (present(skip)) then
                if (skip(e)) cycle
             end if
             if (fplain(e)) then                          ! how many sweeps from here on share this plane set?
                n = 1
                do while (e + n <= ns)
                   if (.not. (fplain(e+n) .and. ff(e+n) == ff(e) .and. fg(e+n) == fg(e))) exit
                   n = n + 1
                end do
                if (n >= 2) then
                   write(*,fmt='(a,i4)') 'Sampling per-pixel at nside ', cc%sample_nside(j)
                   call dangx_sky_plane_sweeps_sample(gpu_sky, int(ff(e)), n, fc(e:e+n-1), fn(e:e+n-1), fs(e:e+n-1), nsample, mode, gpu_seed)
                   done(e+1:e+n-1) = .true.
                   cycle
                end if
             end if
             flag = cc%pol_flag(j,k)
             map_n = map_of_flag(flag)
             if (map_n == 0) then
                write(*,*) "There is something wrong with the poltype flag"
                cycle
             end if
             stream = dangx_stream_id(iter, 1, i-1, j-1, flag)
             tuned = 1
             tuned(1:cc%nindices) = merge(1_c_int32_t, 0_c_int32_t, cc%tuned(1:cc%nindices))
             if (cc%index_mode(j) == 1) then              ! one index for the whole sky: the chain runs behind the ABI
                write(*,*) 'Sampling fullsky'
                call dangx_sky_fullsky_sample(gpu_sky, i-1, j-1, map_n, nsample, mode, gpu_seed, stream, nside, &
                     cc%sample_nside(j), tuned, step, val, nacc)
                cc%tuned(1:cc%nindices) = tuned(1:cc%nindices) /= 0
                cc%step_size(j) = step
             else if (cc%sample_nside(j) /= nside) then
                write(*,fmt='(a,i4)') 'Sampling per-pixel at nside ', cc%sample_nside(j)
                if (numprocs > 1) then      ! one context per process: the sums over the shards go through dang_allreduce
                   call dangx_check(gpu_sky%ctx(1), dangx_index_sample_coarse(gpu_sky%ctx(1), i-1, j-1, map_n, nsample, mode, &
                        gpu_seed, stream, nside, cc%sample_nside(j), nacc), 'index_sample_coarse')
                else                        ! one process, one or several contexts
                   call dangx_sky_index_sample_coarse(gpu_sky, i-1, j-1, map_n, nsample, mode, gpu_seed, stream, nside, &
                        cc%sample_nside(j), nacc)
                end if
             else
                write(*,fmt='(a,i4)') 'Sampling per-pixel at nside ', cc%sample_nside(j)
                if (.not. cc%tuned(j)) then               ! 'Tuning!', src/dang_sample_mod.f90:341-346
                   write(*,*) 'Tuning!'
                   call dangx_sky_tune_perpixel(gpu_sky, i-1, j-1, map_n, nsample, mode, gpu_seed, stream, tuned, step)
                   cc%tuned(1:cc%nindices) = tuned(1:cc%nindices) /= 0
                   cc%step_size(j) = step
                end if
                ! two consecutive plain sweeps of this component on the same planes (dust beta, dust T): one entry point,
                ! one kernel launch where the register chain covers both -- the same numbers as the two sweeps
                pairable = .false.
                if (j < cc%nindices .and. cc%nflag(j) == 1) then
                   if (cc%sample_index(j+1) .and. cc%nflag(j+1) == 1) then
                      skip_next = .false.
                      if (present(skip)) skip_next = skip(e+1)
                      pairable = cc%pol_flag(j+1,1) == flag .and. plain_sweep(cc, j+1) .and. .not. skip_next
                   end if
                end if
                if (pairable) then
                   write(*,fmt='(a,i4)') 'Sampling per-pixel at nside ', cc%sample_nside(j+1)
                   call dangx_sky_index_sample_pair(gpu_sky, i-1, j-1, map_n, nsample, mode, gpu_seed, stream, &
                        dangx_stream_id(iter, 1, i-1, j, flag))
                   pair_done = .true.
                else
                   call dangx_sky_index_sample(gpu_sky, i-1, j-1, map_n, nsample, mode, gpu_seed, stream)
                end if
             end if
          end do
       end do
       ! "Update the global variable T_CMB" (src/dang_sample_mod.f90:75-78)
       if (trim(cc%type) == 'T_cmb') T_CMB = dangx_sky_update_tcmb(gpu_sky, i-1)
    end do
  end subroutine run_sweeps

  subroutine sample_spectral_parameters_gpu(dpar, ddata)
    ! same signature and effect as sample_spectral_parameters, src/dang_sample_mod.f90:21-86
    type(dang_data)   :: ddata
    type(dang_params) :: dpar
    logical(lgt) :: sampled
    call run_sweeps(dpar, ddata, sampled=sampled)
    if (sampled) call gpu_chisq(ddata, .true.)            ! update_sky_model + write_stats_to_term, :81-84
  end subroutine sample_spectral_parameters_gpu

  ! One pass of the main loop for an iteration in which both phases run (src/dang.f90:101-106, iter > 1):
  !     call sample_cg_groups_gpu(dpar, ddata); call sample_spectral_parameters_gpu(dpar, ddata)
  ! with the same state at the end, bit for bit, in fewer passes over the maps: every group's solve is issued together with
  ! the first sweep on its planes where dangx_plan_fusion says the loop's result does not change, consecutive sweeps of a
  ! component go in pairs, no call waits for a count, and chi^2 comes from the sums the sweeps leave behind -- the statistics
  ! are printed once, after the index phase.  (The two-call form prints them after every group, one more pass each.)
  subroutine gibbs_iteration_gpu(dpar, ddata)
    type(dang_data)   :: ddata
    type(dang_params) :: dpar
    type(dang_comps), pointer :: cc
    integer(c_int32_t), allocatable :: pg(:), pf(:), sc(:), sn(:), sf(:), sp(:), first(:), owner(:)
    logical, allocatable :: skip(:)
    logical :: foreign
    integer(i4b) :: i, j, k, f, e, np, ns
    logical(lgt) :: sampled
    np = 0; ns = 0
    do i = 1, ncg_groups
       if (cg_groups(i)%p%sample) np = np + cg_groups(i)%p%nflag
    end do
    do i = 1, ncomp
       cc => component_list(i)%p
       do j = 1, cc%nindices
          if (cc%sample_index(j)) ns = ns + cc%nflag(j)
       end do
    end do
    allocate(pg(np), pf(np), first(np), sc(ns), sn(ns), sf(ns), sp(ns), skip(ns), owner(ns))
    np = 0; ns = 0
    do i = 1, ncg_groups                                  ! the (group, flag) passes in sample_cg_groups' order
       if (.not. cg_groups(i)%p%sample) cycle
       do f = 1, cg_groups(i)%p%nflag
          np = np + 1; pg(np) = i; pf(np) = cg_groups(i)%p%pol_flag(f)
       end do
    end do
    do i = 1, ncomp                                       ! the sweeps in sample_spectral_parameters' order
       cc => component_list(i)%p
       do j = 1, cc%nindices
          if (.not. cc%sample_index(j)) cycle
          do k = 1, cc%nflag(j)
             ns = ns + 1; sc(ns) = i-1; sn(ns) = j-1; sf(ns) = cc%pol_flag(j,k)
             sp(ns) = merge(1, 0, plain_sweep(cc, j))
          end do
       end do
    end do
    first = -1
    if (np > 0 .and. ns > 0) call dangx_check(gpu_sky%ctx(1), dangx_plan_fusion(gpu_sky%ctx(1), np, pg, pf, ns, sc, sn, sf, sp, &
         DANGX_SOLVER_DIRECT, first), 'dangx_plan_fusion')
    ! a solve that may take its first sweep along takes EVERY sweep on its planes (they are plain per-pixel sweeps and nothing
    ! else touches these planes); if a sweep with another flag shares a plane, only the first one goes with the solve
    owner = 0
    do i = 1, np
       if (first(i) < 0) cycle
       foreign = .false.
       do e = 1, ns
          if (sf(e) /= pf(i) .and. iand(planes_of_flag(int(sf(e))), planes_of_flag(int(pf(i)))) /= 0) foreign = .true.
       end do
       if (foreign) then
          owner(first(i)+1) = i
       else
          where (sf == pf(i)) owner = i
       end if
    end do
    skip = owner > 0
    call run_solves(dpar, ddata, .false., owner, sc, sn)
    call run_sweeps(dpar, ddata, skip, sampled)
    call gpu_chisq(ddata, .true.)
  end subroutine gibbs_iteration_gpu

  ! sample_calibrators + fit_band_gain (src/dang_sample_mod.f90:487-518, 570-621) through dangx_fit_band_gain: the sums, the
  ! draw and ddata%gain(band) on every context behind the ABI
  subroutine sample_calibrators_gpu(ddata)
    type(dang_data), intent(inout) :: ddata
    integer(i4b) :: j, mode
    logical(lgt) :: sampled
    sampled = .false.
    if (any(ddata%fit_gain(:))) then
       write(*,*) "Sampling band calibrators"
       sampled = .true.
    end if
    mode = merge(DANGX_ML_SAMPLE, DANGX_ML_OPTIMIZE, trim(ml_mode) == 'sample')
    do j = 1, nbands
       if (ddata%fit_gain(j)) ddata%gain(j) = dangx_sky_fit_band_gain(gpu_sky, j-1, mode, gpu_seed, dangx_stream_id(iter, 2, 0, 0, 0))
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

  ! a trace file of the run: opened for appending, created by the first iteration that writes to it
  subroutine open_trace(unit, fname, created)
    integer(i4b), intent(in)      :: unit
    character(len=*), intent(in)  :: fname
    logical, intent(out)          :: created
    logical :: there
    inquire(file=trim(fname), exist=there)
    created = .not. there
    if (there) then
       open(unit, file=trim(fname), status='old', position='append', action='write')
    else
       open(unit, file=trim(fname), status='new', action='write')
    end if
  end subroutine open_trace

  ! The per-iteration ASCII traces of write_data (src/dang_data_mod.f90:666-761) -- total_chisq_<S>.dat,
  ! <comp>_<S>_amplitudes.dat, <comp>_<index>_mean_<S>.dat, band_gains_k<iter>.dat, band_offsets_k<iter>.dat, each with the
  ! edit descriptors of the reference's statements (:695, :710-712, :719-729, :735-759) so that scripts/ keeps reading them --
  ! with every number taken from the device: chi^2 from the sweeps' sums (or one pass), the index means from masked device
  ! reductions, the template amplitudes from the context.  Runs every iteration and moves no map.
  subroutine write_data_gpu(ddata, dpar, map_n)
    type(dang_data)           :: ddata
    type(dang_params)         :: dpar
    integer(i4b), intent(in)  :: map_n
    type(dang_comps), pointer :: cc
    real(dp), allocatable     :: avg(:,:)
    character(len=512)        :: dir
    character(len=80)         :: stokes
    character(len=16)         :: nb_edit, it5
    integer(i4b)              :: j, n, u, nl
    integer(c_int32_t), allocatable :: lc(:), ln(:), lk(:)
    real(c_double), allocatable     :: la(:)
    logical                   :: created

    call gpu_chisq(ddata, .false.)
    call pull_template_amplitudes(ddata_offset_only=.false.)
    allocate(avg(2, ncomp)); avg = 0.d0                  ! sky-wide sums: every rank takes part, the master writes
    nl = 0
    do n = 1, ncomp                                       ! every sampled index of the model: sized from it
       nl = nl + count(component_list(n)%p%sample_index(1:component_list(n)%p%nindices))
    end do
    allocate(lc(nl), ln(nl), lk(nl), la(nl))
    nl = 0
    do n = 1, ncomp
       cc => component_list(n)%p
       do j = 1, cc%nindices
          if (cc%sample_index(j)) then
             nl = nl + 1; lc(nl) = n-1; ln(nl) = j-1; lk(nl) = map_n
          end if
       end do
    end do
    if (nl > 0) call gpu_index_means(nl, lc, ln, lk, la)     ! mask_avg(c%indices(:,map_n,j), self%masks(:,1)) for all of them
    do j = 1, nl
       avg(ln(j)+1, lc(j)+1) = la(j)
    end do
    if (rank /= master) return
    write(*,*) 'Output data files'
    dir = dpar%outdir; stokes = tqu(map_n)
    write(nb_edit, '(i4)') nbands
    write(it5, '(i0.5)') iter

    call open_trace(33, trim(dir)//'total_chisq_'//trim(stokes)//'.dat', created)
    write(33,*) ddata%chisq
    close(33)

    do n = 1, ncomp
       cc => component_list(n)%p
       if (trim(cc%type) == 'template' .or. trim(cc%type) == 'hi_fit') then
          u = getlun()
          call open_trace(u, trim(dir)//trim(cc%label)//'_'//trim(stokes)//'_amplitudes.dat', created)
          if (created) write(u, fmt='('//trim(nb_edit)//'(A17))') ddata%label
          write(u, fmt='('//trim(nb_edit)//'(E17.8))') cc%template_amplitudes(:,map_n)/cc%temp_norm(map_n)
          close(u)
       end if
       do j = 1, cc%nindices
          if (.not. cc%sample_index(j)) cycle
          u = getlun()
          call open_trace(u, trim(dir)//trim(cc%label)//'_'//trim(cc%ind_label(j))//'_mean_'//trim(stokes)//'.dat', created)
          write(u, fmt='('//achar(48+nmaps)//'(f12.8))') avg(j, n)
          close(u)
       end do
    end do

    call open_trace(37, trim(dir)//'band_gains_k'//trim(it5)//'.dat', created)
    write(37, fmt='(a12,E16.8)') (trim(ddata%label(j)), ddata%gain(j), j = 1, nbands)
    close(37)
    call open_trace(38, trim(dir)//'band_offsets_k'//trim(it5)//'.dat', created)
    write(38, fmt='(a12,E16.8)') (trim(ddata%label(j)), ddata%offset(j)/ddata%conversion(j), j = 1, nbands)
    close(38)
  end subroutine write_data_gpu

end module dang_gpu_mod
