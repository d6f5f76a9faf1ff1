! dangx_multi_mod.f90 -- one sky, several pixel-shard contexts, driven by ONE host thread.
!
! The reference is a single process (numprocs is never used, src/dang_util_mod.f90:48-57); a node has eight GPUs.
! This layer gives a single-process driver all of them: the sky's RING pixels are cut into contiguous ranges, one
! dangx context per device, every context works on the driver's FULL-SKY host arrays through a window
! (dangx_set_host_stride) -- no copy, no re-packing on the Fortran side -- and every per-pixel call is enqueued on
! all devices before the first result is awaited (the count outputs of dangx_amp_sample / dangx_index_sample are
! nullable: without them the call does not synchronise).  Sky-wide sums (chi^2, index means, gain sums, the sums of
! the full-sky Metropolis chain) are added over the contexts in shard order, which makes them independent of timing.
!
! Only plain arrays and ISO_C_BINDING here: the wrapper with the reference's derived types is
! fortran/reference_side/dang_gpu_mod.f90; fortran/dangx_fsmoke.f90 drives this module on a GPU in the tests.
module dangx_multi_mod
  use, intrinsic :: iso_c_binding
  use dangx_mod
  implicit none

  integer, parameter :: DANGX_MAX_CTX = 16

  type :: dangx_sky
     integer :: nctx = 0
     type(c_ptr) :: ctx(DANGX_MAX_CTX) = c_null_ptr
     integer(c_int64_t) :: pix0(DANGX_MAX_CTX) = 0, npix(DANGX_MAX_CTX) = 0
     integer(c_int64_t) :: npix_global = 0
     integer(c_int32_t) :: nmaps = 0, nbands = 0, ncomp = 0
  end type dangx_sky

contains

  ! contiguous RING range of shard r (0-based) of n, as dang_amd/dist.py:shard_range
  subroutine dangx_shard_range(npix_global, r, n, pix0, npix)
    integer(c_int64_t), intent(in)  :: npix_global
    integer, intent(in)             :: r, n
    integer(c_int64_t), intent(out) :: pix0, npix
    integer(c_int64_t) :: base, rem
    base = npix_global/n
    rem  = mod(npix_global, int(n, c_int64_t))
    pix0 = r*base + min(int(r, c_int64_t), rem)
    npix = base + merge(1_c_int64_t, 0_c_int64_t, r < rem)
  end subroutine dangx_shard_range

  ! Boundaries bounds(0:n) of n contiguous RING ranges of equal WORK: a masked pixel costs a kernel almost nothing, so equal
  ! ranges leave the contexts that hold a Galactic mask's band with little to do (dang_amd/dist.py: balanced_bounds_mask;
  ! weights 32 : 1).  Results do not depend on the boundaries (the random streams are keyed by the global pixel).
  subroutine dangx_balanced_bounds(mask, n, bounds)
    real(c_double), intent(in)      :: mask(0:)
    integer, intent(in)             :: n
    integer(c_int64_t), intent(out) :: bounds(0:n)
    integer(c_int64_t) :: total, acc, i, np
    integer :: r
    np = size(mask, kind=c_int64_t)
    total = 0
    do i = 0, np-1
       total = total + merge(1_c_int64_t, 32_c_int64_t, mask(i) == 0.d0 .or. mask(i) == -1.6375d30)
    end do
    bounds(0) = 0; bounds(n) = np
    acc = 0; r = 1
    do i = 0, np-1
       if (r >= n) exit
       acc = acc + merge(1_c_int64_t, 32_c_int64_t, mask(i) == 0.d0 .or. mask(i) == -1.6375d30)
       do while (r < n)
          if (acc < (total*r)/n) exit
          bounds(r) = i + 1; r = r + 1
       end do
    end do
    do while (r < n)
       bounds(r) = np; r = r + 1
    end do
  end subroutine dangx_balanced_bounds

  ! nctx contexts over the sky; context r runs on device devices(r) (or r modulo the device count).  mask (the full-sky mask,
  ! masks(:,1)) present: shards of equal work instead of equal pixel count.
  subroutine dangx_sky_create(sky, npix_global, nmaps, nbands, ncomp, nctx, devices, mask)
    type(dangx_sky), intent(out)   :: sky
    integer(c_int64_t), intent(in) :: npix_global
    integer, intent(in)            :: nmaps, nbands, ncomp, nctx
    integer, intent(in), optional  :: devices(:)
    real(c_double), intent(in), optional :: mask(0:)
    type(dangx_dims) :: dims
    integer(c_int) :: ndev
    integer :: r, dev
    integer(c_int64_t) :: bounds(0:DANGX_MAX_CTX)
    if (nctx < 1 .or. nctx > DANGX_MAX_CTX) then
       write(*,*) 'dangx_sky_create: bad number of contexts ', nctx
       stop 1
    end if
    call dangx_check(c_null_ptr, dangx_device_count(ndev), 'dangx_device_count')
    if (ndev < 1) then
       write(*,*) 'dangx_sky_create: no HIP device (the GPU path has no CPU fallback)'
       stop 1
    end if
    sky%nctx = nctx; sky%npix_global = npix_global
    sky%nmaps = nmaps; sky%nbands = nbands; sky%ncomp = ncomp
    if (present(mask) .and. nctx > 1) call dangx_balanced_bounds(mask, nctx, bounds(0:nctx))
    do r = 1, nctx
       if (present(mask) .and. nctx > 1) then
          sky%pix0(r) = bounds(r-1); sky%npix(r) = bounds(r) - bounds(r-1)
       else
          call dangx_shard_range(npix_global, r-1, nctx, sky%pix0(r), sky%npix(r))
       end if
       dev = mod(r-1, ndev)
       if (present(devices)) dev = devices(r)
       dims = dangx_dims(int(sky%npix(r), c_int32_t), nmaps, nbands, ncomp, sky%pix0(r), npix_global, dev, 0)
       call dangx_check(c_null_ptr, dangx_create(sky%ctx(r), dims), 'dangx_create')
       ! host arrays are the driver's full-sky arrays: plane stride = full-sky pixel count
       call dangx_check(sky%ctx(r), dangx_set_host_stride(sky%ctx(r), npix_global), 'dangx_set_host_stride')
    end do
  end subroutine dangx_sky_create

  subroutine dangx_sky_destroy(sky)
    type(dangx_sky), intent(inout) :: sky
    integer :: r
    do r = 1, sky%nctx
       call dangx_check(sky%ctx(r), dangx_destroy(sky%ctx(r)), 'dangx_destroy')
       sky%ctx(r) = c_null_ptr
    end do
    sky%nctx = 0
  end subroutine dangx_sky_destroy

  ! address of pixel pix0 of a full-sky map array whose first dimension is the pixel (0-based)
  function at_pix(base, pix0) result(p)
    type(c_ptr), intent(in) :: base
    integer(c_int64_t), intent(in) :: pix0
    type(c_ptr) :: p
    p = transfer(transfer(base, 0_c_intptr_t) + 8_c_intptr_t*pix0, p)
  end function at_pix

  ! ---- static description, identical on every context
  subroutine dangx_sky_set_band(sky, band, nu_c, n, nu0, tau0)
    type(dangx_sky), intent(in) :: sky
    integer, intent(in) :: band, n
    real(c_double), intent(in) :: nu_c
    type(c_ptr), intent(in) :: nu0, tau0
    integer :: r
    do r = 1, sky%nctx
       call dangx_check(sky%ctx(r), dangx_set_band(sky%ctx(r), band, nu_c, n, nu0, tau0), 'dangx_set_band')
    end do
  end subroutine dangx_sky_set_band

  subroutine dangx_sky_set_component(sky, comp, d)
    type(dangx_sky), intent(in) :: sky
    integer, intent(in) :: comp
    type(dangx_comp_desc), intent(in) :: d
    integer :: r
    do r = 1, sky%nctx
       call dangx_check(sky%ctx(r), dangx_set_component(sky%ctx(r), comp, d), 'dangx_set_component')
    end do
  end subroutine dangx_sky_set_component

  subroutine dangx_sky_set_tcmb(sky, T)
    type(dangx_sky), intent(in) :: sky
    real(c_double), intent(in) :: T
    integer :: r
    do r = 1, sky%nctx
       call dangx_check(sky%ctx(r), dangx_set_tcmb(sky%ctx(r), T), 'dangx_set_tcmb')
    end do
  end subroutine dangx_sky_set_tcmb

  subroutine dangx_sky_set_calibration(sky, gain, offset)
    type(dangx_sky), intent(in) :: sky
    type(c_ptr), intent(in) :: gain, offset
    integer :: r
    do r = 1, sky%nctx
       call dangx_check(sky%ctx(r), dangx_set_calibration(sky%ctx(r), gain, offset), 'dangx_set_calibration')
    end do
  end subroutine dangx_sky_set_calibration

  ! ---- maps: c_loc of the FULL-SKY arrays (first element); every context takes its window
  subroutine dangx_sky_upload_data(sky, sig, rms, mask)
    type(dangx_sky), intent(in) :: sky
    type(c_ptr), intent(in) :: sig, rms, mask
    integer :: r
    do r = 1, sky%nctx
       call dangx_check(sky%ctx(r), dangx_upload_data(sky%ctx(r), at_pix(sig, sky%pix0(r)), at_pix(rms, sky%pix0(r)), &
            at_pix(mask, sky%pix0(r))), 'dangx_upload_data')
    end do
  end subroutine dangx_sky_upload_data

  subroutine dangx_sky_put_state(sky, comp, amp, ind)
    type(dangx_sky), intent(in) :: sky
    integer, intent(in) :: comp
    type(c_ptr), intent(in) :: amp, ind      ! ind = c_null_ptr for a component without indices
    integer :: r
    do r = 1, sky%nctx
       call dangx_check(sky%ctx(r), dangx_put_amplitude(sky%ctx(r), comp, at_pix(amp, sky%pix0(r))), 'dangx_put_amplitude')
       if (c_associated(ind)) call dangx_check(sky%ctx(r), dangx_put_indices(sky%ctx(r), comp, at_pix(ind, sky%pix0(r))), &
            'dangx_put_indices')
    end do
  end subroutine dangx_sky_put_state

  subroutine dangx_sky_get_state(sky, comp, amp, ind)
    type(dangx_sky), intent(in) :: sky
    integer, intent(in) :: comp
    type(c_ptr), intent(in) :: amp, ind
    integer :: r
    do r = 1, sky%nctx
       call dangx_check(sky%ctx(r), dangx_get_amplitude(sky%ctx(r), comp, at_pix(amp, sky%pix0(r))), 'dangx_get_amplitude')
       if (c_associated(ind)) call dangx_check(sky%ctx(r), dangx_get_indices(sky%ctx(r), comp, at_pix(ind, sky%pix0(r))), &
            'dangx_get_indices')
    end do
  end subroutine dangx_sky_get_state

  subroutine dangx_sky_set_template(sky, comp, tmpl, corr, nfit, ta)
    type(dangx_sky), intent(in) :: sky
    integer, intent(in) :: comp, nfit
    type(c_ptr), intent(in) :: tmpl, corr, ta
    integer :: r
    do r = 1, sky%nctx
       call dangx_check(sky%ctx(r), dangx_set_template(sky%ctx(r), comp, at_pix(tmpl, sky%pix0(r)), corr, nfit), 'dangx_set_template')
       call dangx_check(sky%ctx(r), dangx_put_template_amplitudes(sky%ctx(r), comp, ta), 'dangx_put_template_amplitudes')
    end do
  end subroutine dangx_sky_set_template

  ! ---- the two per-pixel phases: enqueue on every device, then wait for all
  subroutine dangx_sky_wait(sky)
    type(dangx_sky), intent(in) :: sky
    integer :: r
    do r = 1, sky%nctx
       call dangx_check(sky%ctx(r), dangx_synchronize(sky%ctx(r)), 'dangx_synchronize')
    end do
  end subroutine dangx_sky_wait

  ! one (group, flag) pass of sample_cg_groups (src/dang_cg_mod.f90:166-171) through dangx_sky_amp_sample: diffuse groups
  ! (every BASELINE configuration) are enqueued on every device; groups with template / monopole / hi_fit members share
  ! their Schur rows over the contexts (pass 1 everywhere, one small solve, pass 2 everywhere).  Without n_not_spd a
  ! diffuse group's call does not wait for any device.
  subroutine dangx_sky_amp_sample(sky, group, flag, ml_mode, fluct_mode, seed, stream, n_not_spd, nullity)
    type(dangx_sky), intent(in) :: sky
    integer, intent(in) :: group, flag, ml_mode, fluct_mode
    integer(c_int64_t), intent(in) :: seed, stream
    integer(c_int64_t), intent(out), optional :: n_not_spd
    integer, intent(out), optional :: nullity          ! directions of the global amplitudes left at their current value
    integer(c_int) :: iters
    integer(c_int64_t) :: nbad
    if (present(n_not_spd) .or. present(nullity)) then
       call dangx_check(sky%ctx(1), dangx_sky_amp_sample_c(sky%ctx, sky%nctx, group, flag, ml_mode, DANGX_SOLVER_DIRECT, &
            fluct_mode, seed, stream, 0, 0.d0, iters, nbad), 'dangx_sky_amp_sample')
       if (present(n_not_spd)) n_not_spd = nbad
       if (present(nullity)) nullity = -iters
    else
       call dangx_check(sky%ctx(1), dangx_sky_amp_sample_c(sky%ctx, sky%nctx, group, flag, ml_mode, DANGX_SOLVER_DIRECT, &
            fluct_mode, seed, stream, 0, 0.d0), 'dangx_sky_amp_sample')
    end if
  end subroutine dangx_sky_amp_sample

  ! sample_index_mh, per-pixel branch (src/dang_sample_mod.f90:332-483), for (comp, nind, map_n), 0-based comp / nind
  subroutine dangx_sky_index_sample(sky, comp, nind, map_n, nsample, ml_mode, seed, stream, accepted)
    type(dangx_sky), intent(in) :: sky
    integer, intent(in) :: comp, nind, map_n, nsample, ml_mode
    integer(c_int64_t), intent(in) :: seed, stream
    integer(c_int64_t), intent(out), optional :: accepted
    integer(c_int64_t) :: nacc
    integer :: r
    if (present(accepted)) then
       accepted = 0
       do r = 1, sky%nctx
          call dangx_check(sky%ctx(r), dangx_index_sample(sky%ctx(r), comp, nind, map_n, nsample, ml_mode, seed, stream, nacc), &
               'dangx_index_sample')
          accepted = accepted + nacc
       end do
    else
       do r = 1, sky%nctx
          call dangx_check(sky%ctx(r), dangx_index_sample(sky%ctx(r), comp, nind, map_n, nsample, ml_mode, seed, stream), &
               'dangx_index_sample')
       end do
    end if
  end subroutine dangx_sky_index_sample

  ! two consecutive indices (nind, nind + 1) of one component on the same planes through dangx_index_sample_pair: one launch
  ! per context where the register chain covers both, the two calls' result bit for bit everywhere
  subroutine dangx_sky_index_sample_pair(sky, comp, nind, map_n, nsample, ml_mode, seed, stream_first, stream_second, &
       accepted_first, accepted_second)
    type(dangx_sky), intent(in) :: sky
    integer, intent(in) :: comp, nind, map_n, nsample, ml_mode
    integer(c_int64_t), intent(in) :: seed, stream_first, stream_second
    integer(c_int64_t), intent(out), optional :: accepted_first, accepted_second
    integer(c_int64_t) :: n1, n2
    integer :: r
    if (present(accepted_first) .or. present(accepted_second)) then
       if (present(accepted_first)) accepted_first = 0
       if (present(accepted_second)) accepted_second = 0
       do r = 1, sky%nctx
          call dangx_check(sky%ctx(r), dangx_index_sample_pair(sky%ctx(r), comp, nind, map_n, nsample, ml_mode, seed, &
               stream_first, stream_second, n1, n2), 'dangx_index_sample_pair')
          if (present(accepted_first)) accepted_first = accepted_first + n1
          if (present(accepted_second)) accepted_second = accepted_second + n2
       end do
    else
       do r = 1, sky%nctx
          call dangx_check(sky%ctx(r), dangx_index_sample_pair(sky%ctx(r), comp, nind, map_n, nsample, ml_mode, seed, &
               stream_first, stream_second), 'dangx_index_sample_pair')
       end do
    end if
  end subroutine dangx_sky_index_sample_pair

  ! dangx_sky_amp_sample(group, flag, ...) directly followed by dangx_sky_index_sample(comp, nind, map_n, ...) on the same
  ! planes, through dangx_amp_index_sample: one kernel launch per context where the model allows it, the two calls'
  ! result bit for bit everywhere (the first sampled index of a CG group's components follows the group's solve this way)
  subroutine dangx_sky_amp_index_sample(sky, group, flag, ml_mode, fluct_mode, seed_amp, stream_amp, &
       comp, nind, map_n, nsample, seed_index, stream_index, n_not_spd, accepted)
    type(dangx_sky), intent(in) :: sky
    integer, intent(in) :: group, flag, ml_mode, fluct_mode, comp, nind, map_n, nsample
    integer(c_int64_t), intent(in) :: seed_amp, stream_amp, seed_index, stream_index
    integer(c_int64_t), intent(out), optional :: n_not_spd, accepted
    integer(c_int64_t) :: nbad, nacc
    integer :: r
    if (present(n_not_spd) .or. present(accepted)) then
       if (present(n_not_spd)) n_not_spd = 0
       if (present(accepted)) accepted = 0
       do r = 1, sky%nctx
          call dangx_check(sky%ctx(r), dangx_amp_index_sample(sky%ctx(r), group, flag, ml_mode, DANGX_SOLVER_DIRECT, fluct_mode, &
               seed_amp, stream_amp, 0, 0.d0, comp, nind, map_n, nsample, seed_index, stream_index, n_not_spd=nbad, accepted=nacc), &
               'dangx_amp_index_sample')
          if (present(n_not_spd)) n_not_spd = n_not_spd + nbad
          if (present(accepted)) accepted = accepted + nacc
       end do
    else
       do r = 1, sky%nctx
          call dangx_check(sky%ctx(r), dangx_amp_index_sample(sky%ctx(r), group, flag, ml_mode, DANGX_SOLVER_DIRECT, fluct_mode, &
               seed_amp, stream_amp, 0, 0.d0, comp, nind, map_n, nsample, seed_index, stream_index), 'dangx_amp_index_sample')
       end do
    end if
  end subroutine dangx_sky_amp_index_sample

  ! dangx_sky_amp_sample(group, flag, ...) followed by dangx_sky_index_sample(comp(s), nind(s), ...) for every sweep on the group's
  ! planes, in the reference's order, through dangx_sky_plane_set_sample: ONE launch per context for models with many bands and
  ! members (the members' SED columns stay in LDS across the sweeps); a group with template members shares its Schur rows over the
  ! contexts and, where the plane-set kernel covers the model, back-substitutes inside the launch that runs the sweeps
  subroutine dangx_sky_plane_set_sample(sky, group, flag, ml_mode, fluct_mode, seed_amp, stream_amp, nsweeps, comp, nind, stream, &
       nsample, seed_index, n_not_spd, accepted, nullity)
    type(dangx_sky), intent(in) :: sky
    integer, intent(in) :: group, flag, ml_mode, fluct_mode, nsweeps, nsample
    integer(c_int32_t), intent(in) :: comp(nsweeps), nind(nsweeps)      ! 0-based
    integer(c_int64_t), intent(in) :: seed_amp, stream_amp, seed_index, stream(nsweeps)
    integer(c_int64_t), intent(out), optional :: n_not_spd, accepted(nsweeps)
    integer, intent(out), optional :: nullity
    integer(c_int64_t) :: nbad, nacc(nsweeps)
    integer(c_int) :: iters
    iters = 0
    if (present(n_not_spd) .or. present(accepted)) then
       call dangx_check(sky%ctx(1), dangx_sky_plane_set_sample_c(sky%ctx, sky%nctx, group, flag, ml_mode, DANGX_SOLVER_DIRECT, fluct_mode, &
            seed_amp, stream_amp, 0, 0.d0, nsweeps, comp, nind, stream, nsample, seed_index, iters, nbad, nacc), 'dangx_sky_plane_set_sample')
       if (present(n_not_spd)) n_not_spd = nbad
       if (present(accepted)) accepted = nacc
    else
       call dangx_check(sky%ctx(1), dangx_sky_plane_set_sample_c(sky%ctx, sky%nctx, group, flag, ml_mode, DANGX_SOLVER_DIRECT, fluct_mode, &
            seed_amp, stream_amp, 0, 0.d0, nsweeps, comp, nind, stream, nsample, seed_index, iters), 'dangx_sky_plane_set_sample')
    end if
    if (present(nullity)) nullity = -iters
  end subroutine dangx_sky_plane_set_sample

  ! dangx_sky_index_sample(comp(s), nind(s), ...) for the sweeps of ONE plane set, in the reference's order, through
  ! dangx_plane_sweeps_sample: one launch per context where the plane-set kernel covers the model, those calls otherwise
  subroutine dangx_sky_plane_sweeps_sample(sky, flag, nsweeps, comp, nind, stream, nsample, ml_mode, seed, accepted)
    type(dangx_sky), intent(in) :: sky
    integer, intent(in) :: flag, nsweeps, nsample, ml_mode
    integer(c_int32_t), intent(in) :: comp(nsweeps), nind(nsweeps)      ! 0-based
    integer(c_int64_t), intent(in) :: seed, stream(nsweeps)
    integer(c_int64_t), intent(out), optional :: accepted(nsweeps)
    integer(c_int64_t) :: nacc(nsweeps)
    integer :: r
    if (present(accepted)) accepted = 0
    do r = 1, sky%nctx
       if (present(accepted)) then
          call dangx_check(sky%ctx(r), dangx_plane_sweeps_sample(sky%ctx(r), flag, nsweeps, comp, nind, stream, nsample, ml_mode, seed, nacc), &
               'dangx_plane_sweeps_sample')
          accepted = accepted + nacc
       else
          call dangx_check(sky%ctx(r), dangx_plane_sweeps_sample(sky%ctx(r), flag, nsweeps, comp, nind, stream, nsample, ml_mode, seed), &
               'dangx_plane_sweeps_sample')
       end if
    end do
  end subroutine dangx_sky_plane_sweeps_sample

  ! sample_index_mh with sample_nside /= nside (src/dang_sample_mod.f90:199-217, 332-483) over the contexts: the three
  ! phases of dangx_index_sample_coarse, the shards' buffers added in shard order between them
  subroutine dangx_sky_index_sample_coarse(sky, comp, nind, map_n, nsample, ml_mode, seed, stream, nside, sample_nside, accepted)
    type(dangx_sky), intent(in) :: sky
    integer, intent(in) :: comp, nind, map_n, nsample, ml_mode, nside, sample_nside
    integer(c_int64_t), intent(in) :: seed, stream
    integer(c_int64_t), intent(out) :: accepted
    integer(c_int64_t) :: np, ni
    real(c_double), allocatable, target :: part(:), psum(:), idx(:), isum(:)
    integer :: r
    if (sky%nctx == 1 .and. sky%npix(1) == sky%npix_global) then     ! one whole-sky context: the direct form
       call dangx_check(sky%ctx(1), dangx_index_sample_coarse(sky%ctx(1), comp, nind, map_n, nsample, ml_mode, seed, stream, &
            nside, sample_nside, accepted), 'dangx_index_sample_coarse')
       return
    end if
    call dangx_check(sky%ctx(1), dangx_coarse_sizes(sky%ctx(1), map_n, sample_nside, np, ni), 'dangx_coarse_sizes')
    allocate(part(np), psum(np), idx(ni), isum(ni))
    psum = 0.d0; isum = 0.d0
    do r = 1, sky%nctx
       call dangx_check(sky%ctx(r), dangx_coarse_partials(sky%ctx(r), comp, map_n, nside, sample_nside, c_loc(part)), &
            'dangx_coarse_partials')
       psum = psum + part
    end do
    do r = 1, sky%nctx
       call dangx_check(sky%ctx(r), dangx_coarse_chains(sky%ctx(r), comp, nind, map_n, nsample, ml_mode, seed, stream, nside, &
            sample_nside, c_loc(psum), c_loc(idx)), 'dangx_coarse_chains')
       isum = isum + idx
    end do
    do r = 1, sky%nctx
       call dangx_check(sky%ctx(r), dangx_coarse_writeback(sky%ctx(r), comp, nind, map_n, nside, sample_nside, c_loc(isum)), &
            'dangx_coarse_writeback')
    end do
    accepted = int(isum(ni), c_int64_t)
  end subroutine dangx_sky_index_sample_coarse

  ! ---- sky-wide numbers: sums over the contexts in shard order
  ! update_sky_model + compute_chisq (src/dang_data_mod.f90:339-396, 494-526): chisq = sum/nbands/nump.  With the three
  ! optional full-sky host arrays (c_loc of sky_model / res_map / chi_map) it also refreshes them -- the state
  ! write_maps reads (:573-664).
  function dangx_sky_chisq(sky, pol_lo, pol_hi, nump, sky_model, res_map, chi_map) result(chisq)
    type(dangx_sky), intent(in) :: sky
    integer, intent(in) :: pol_lo, pol_hi
    real(c_double), intent(in) :: nump
    type(c_ptr), intent(in), optional :: sky_model, res_map, chi_map
    real(c_double) :: chisq, s
    type(c_ptr) :: ps, pr, pc
    integer :: r
    chisq = 0.d0
    do r = 1, sky%nctx
       ps = c_null_ptr; pr = c_null_ptr; pc = c_null_ptr
       if (present(sky_model)) ps = at_pix(sky_model, sky%pix0(r))
       if (present(res_map))   pr = at_pix(res_map, sky%pix0(r))
       if (present(chi_map))   pc = at_pix(chi_map, sky%pix0(r))
       call dangx_check(sky%ctx(r), dangx_sky_model_chisq(sky%ctx(r), pol_lo, pol_hi, s, ps, pr, pc), 'dangx_sky_model_chisq')
       chisq = chisq + s
    end do
    chisq = chisq/sky%nbands/nump
  end function dangx_sky_chisq

  ! the same number from the sums the index sweeps leave behind (which = 0: state after the amplitude phase, 1: now);
  ! ok = .false. when some plane was not swept since its last change (the caller then uses dangx_sky_chisq)
  function dangx_sky_chisq_cached(sky, which, pol_lo, pol_hi, nump, ok) result(chisq)
    type(dangx_sky), intent(in) :: sky
    integer, intent(in) :: which, pol_lo, pol_hi
    real(c_double), intent(in) :: nump
    logical, intent(out) :: ok
    real(c_double) :: chisq, s
    integer(c_int) :: st
    integer :: r
    chisq = 0.d0; ok = .true.
    do r = 1, sky%nctx
       st = dangx_chisq_cached(sky%ctx(r), which, pol_lo, pol_hi, s)
       if (st == 2) then
          ok = .false.
          return
       end if
       call dangx_check(sky%ctx(r), st, 'dangx_chisq_cached')
       chisq = chisq + s
    end do
    chisq = chisq/sky%nbands/nump
  end function dangx_sky_chisq_cached

  ! ddata%chisq for the current state at the least cost (dangx_chisq_current on every context): cached plane sums where the
  ! last sweeps left them, an explicit pass over the other planes only
  function dangx_sky_chisq_current(sky, pol_lo, pol_hi, nump) result(chisq)
    type(dangx_sky), intent(in) :: sky
    integer, intent(in) :: pol_lo, pol_hi
    real(c_double), intent(in) :: nump
    real(c_double) :: chisq, s
    integer :: r
    chisq = 0.d0
    do r = 1, sky%nctx
       call dangx_check(sky%ctx(r), dangx_chisq_current(sky%ctx(r), pol_lo, pol_hi, s), 'dangx_chisq_current')
       chisq = chisq + s
    end do
    chisq = chisq/sky%nbands/nump
  end function dangx_sky_chisq_current

  ! n masked index means at once (one launch and one wait per context): comp / nind 0-based, map_n = 1..nmaps
  subroutine dangx_sky_index_means(sky, n, comp, nind, map_n, avg)
    type(dangx_sky), intent(in) :: sky
    integer, intent(in) :: n
    integer(c_int32_t), intent(in) :: comp(n), nind(n), map_n(n)
    real(c_double), intent(out) :: avg(n)
    real(c_double) :: s(n), tot(n)
    integer(c_int64_t) :: cnt(n), ntot(n)
    integer :: r
    tot = 0.d0; ntot = 0
    do r = 1, sky%nctx
       call dangx_check(sky%ctx(r), dangx_index_masked_sums(sky%ctx(r), n, comp, nind, map_n, s, cnt), 'dangx_index_masked_sums')
       tot = tot + s; ntot = ntot + cnt
    end do
    avg = tot/ntot
  end subroutine dangx_sky_index_means

  ! mask_avg(c%indices(:,map_n,nind), masks(:,1)) (src/dang_util_mod.f90:186-206) without moving a map
  function dangx_sky_index_mean(sky, comp, nind, map_n) result(avg)
    type(dangx_sky), intent(in) :: sky
    integer, intent(in) :: comp, nind, map_n
    real(c_double) :: avg, s, tot
    integer(c_int64_t) :: n, ntot
    integer :: r
    tot = 0.d0; ntot = 0
    do r = 1, sky%nctx
       call dangx_check(sky%ctx(r), dangx_index_masked_sum(sky%ctx(r), comp, nind, map_n, s, n), 'dangx_index_masked_sum')
       tot = tot + s; ntot = ntot + n
    end do
    avg = tot/ntot
  end function dangx_sky_index_mean

  ! ---- the sky-wide steps of the loop: the chains run behind the C ABI (dang_amd/csrc/dangx_sky.hip), over all contexts
  ! sample_index_mh with index_mode == 1 (src/dang_sample_mod.f90:229-329), tuner included; comp / nind 0-based.
  ! tuned(1:nindices) = c%tuned as 0 / 1, step_size = c%step_size(nind) after the call.
  subroutine dangx_sky_fullsky_sample(sky, comp, nind, map_n, nsample, ml_mode, seed, stream, nside, sample_nside, tuned, &
       step_size, value, accepted)
    type(dangx_sky), intent(in) :: sky
    integer, intent(in) :: comp, nind, map_n, nsample, ml_mode, nside, sample_nside
    integer(c_int64_t), intent(in) :: seed, stream
    integer(c_int32_t), intent(inout) :: tuned(*)
    real(c_double), intent(out) :: step_size, value
    integer(c_int64_t), intent(out) :: accepted
    call dangx_check(sky%ctx(1), dangx_fullsky_sample(sky%ctx, sky%nctx, comp, nind, map_n, nsample, ml_mode, seed, stream, &
         nside, sample_nside, tuned, step_size, value, accepted), 'dangx_fullsky_sample')
  end subroutine dangx_sky_fullsky_sample

  ! the 'Tuning!' block of the per-pixel branch (src/dang_sample_mod.f90:337-346)
  subroutine dangx_sky_tune_perpixel(sky, comp, nind, map_n, nsample, ml_mode, seed, stream, tuned, step_size)
    type(dangx_sky), intent(in) :: sky
    integer, intent(in) :: comp, nind, map_n, nsample, ml_mode
    integer(c_int64_t), intent(in) :: seed, stream
    integer(c_int32_t), intent(inout) :: tuned(*)
    real(c_double), intent(out) :: step_size
    call dangx_check(sky%ctx(1), dangx_tune_perpixel(sky%ctx, sky%nctx, comp, nind, map_n, nsample, ml_mode, seed, stream, &
         tuned, step_size), 'dangx_tune_perpixel')
  end subroutine dangx_sky_tune_perpixel

  ! fit_band_gain(ddata, 1, band) (src/dang_sample_mod.f90:570-621), band 0-based; the new gain is on every context
  function dangx_sky_fit_band_gain(sky, band, ml_mode, seed, stream) result(gain)
    type(dangx_sky), intent(in) :: sky
    integer, intent(in) :: band, ml_mode
    integer(c_int64_t), intent(in) :: seed, stream
    real(c_double) :: gain
    call dangx_check(sky%ctx(1), dangx_fit_band_gain(sky%ctx, sky%nctx, band, ml_mode, seed, stream, gain), 'dangx_fit_band_gain')
  end function dangx_sky_fit_band_gain

  ! "Update the global variable T_CMB" (src/dang_sample_mod.f90:75-78) from the 'T_cmb' component comp (0-based)
  function dangx_sky_update_tcmb(sky, comp) result(T)
    type(dangx_sky), intent(in) :: sky
    integer, intent(in) :: comp
    real(c_double) :: T
    call dangx_check(sky%ctx(1), dangx_update_tcmb(sky%ctx, sky%nctx, comp, T), 'dangx_update_tcmb')
  end function dangx_sky_update_tcmb

end module dangx_multi_mod
