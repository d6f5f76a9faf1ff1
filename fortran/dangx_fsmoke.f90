! dangx_fsmoke.f90 -- Fortran smoke driver for libdangx.so through dangx_mod / dangx_multi_mod.
!
! Reads a problem written by tests/test_fortran_gpu.py (arrays shaped exactly like the reference's:
! sig_map(0:npix-1,nmaps,nbands), c%amplitude(0:npix-1,nmaps), c%indices(0:npix-1,nmaps,nindices); component
! descriptors as raw dangx_comp_desc records), cuts the sky into `nctx` pixel shards -- one context each, all on the
! visible device(s), working on windows of the FULL-SKY host arrays -- and runs `niter` Gibbs iterations the way
! program dang does (src/dang.f90:87-126): sample_cg_groups for every group, from the second iteration on one
! Metropolis sweep of every sampled index.  Then it does what the output side of the loop needs: pulls the state,
! refreshes sky_model / res_map / chi_map / chisq (update_sky_model + compute_chisq) and the masked index means of
! write_data, and writes everything back for comparison with the oracle and with a run on a different shard count.
program dangx_fsmoke
  use, intrinsic :: iso_c_binding
  use dangx_mod
  use dangx_multi_mod
  implicit none
  integer, parameter :: MAXC = 16
  integer(c_int32_t) :: npix, nmaps, nbands, ncomp, nsample, niter, ngroups, nctx
  integer(c_int64_t) :: seed, nbad, nacc, nacc2, nacc_tot
  integer(c_int) :: st, iters
  real(c_double), allocatable, target :: sig(:,:,:), rms(:,:,:), mask(:,:), freqs(:)
  real(c_double), allocatable, target :: sky_model(:,:,:), res_map(:,:,:), chi_map(:,:)
  type amap
     real(c_double), allocatable :: amp(:,:), ind(:,:,:)
  end type amap
  type(amap), target :: cm(MAXC)
  type(dangx_comp_desc) :: desc(MAXC)
  integer(c_int32_t) :: sample_index(2, MAXC), pol_flag(2, MAXC), grp(MAXC), gflag(MAXC)
  real(c_double) :: nump, chisq_explicit, chisq_amp, chisq_idx, means(2*MAXC)
  logical :: ok_amp, ok_idx, paired
  type(dangx_sky) :: sky
  character(len=512) :: fin, fout, arg
  integer :: i, j, k, l, f, it, u, map_n, nmeans, lf, jf, fused_l(MAXC), fused_j(MAXC)
  integer, save :: ncalls = 0

  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  call get_command_argument(3, arg)
  read(arg, *) nctx
  open(newunit=u, file=trim(fin), access='stream', form='unformatted', status='old')
  read(u) npix, nmaps, nbands, ncomp, nsample, niter, ngroups, seed, nump
  allocate(freqs(nbands))
  read(u) freqs
  do l = 1, ncomp
     read(u) desc(l), sample_index(:, l), pol_flag(:, l)
  end do
  read(u) grp(1:ngroups), gflag(1:ngroups)
  allocate(sig(0:npix-1,nmaps,nbands), rms(0:npix-1,nmaps,nbands), mask(0:npix-1,nmaps))
  allocate(sky_model(0:npix-1,nmaps,nbands), res_map(0:npix-1,nmaps,nbands), chi_map(0:npix-1,nmaps))
  read(u) sig, rms, mask
  do l = 1, ncomp
     allocate(cm(l)%amp(0:npix-1,nmaps), cm(l)%ind(0:npix-1,nmaps,max(desc(l)%nindices,1)))
     read(u) cm(l)%amp
     if (desc(l)%nindices > 0) read(u) cm(l)%ind
  end do
  close(u)

  ! ---- dangx_init: static description + maps, every context takes its window of the full-sky arrays
  call dangx_sky_create(sky, int(npix, c_int64_t), nmaps, nbands, ncomp, nctx)
  do j = 1, nbands
     call dangx_sky_set_band(sky, j-1, freqs(j), 0, c_null_ptr, c_null_ptr)
  end do
  do l = 1, ncomp
     call dangx_sky_set_component(sky, l-1, desc(l))
  end do
  call dangx_sky_upload_data(sky, c_loc(sig), c_loc(rms), c_loc(mask))
  do l = 1, ncomp
     if (desc(l)%nindices > 0) then
        call dangx_sky_put_state(sky, l-1, c_loc(cm(l)%amp), c_loc(cm(l)%ind))
     else
        call dangx_sky_put_state(sky, l-1, c_loc(cm(l)%amp), c_null_ptr)
     end if
  end do

  ! ---- the Gibbs loop of program dang (src/dang.f90:87-126) without the output calls
  nacc_tot = 0
  chisq_amp = 0.d0; chisq_idx = 0.d0; ok_amp = .false.; ok_idx = .false.
  do it = 1, niter
     fused_l = 0; fused_j = 0
     do i = 1, ngroups                                    ! sample_cg_groups, src/dang_cg_mod.f90:142-177
        ! the first sampled index of the group's components on the group's planes: its sweep directly follows this solve
        ! and nothing else touches those planes in between, so the pair goes through dangx_amp_index_sample (one launch
        ! where the model allows it; the Python host of tests/test_fortran_gpu.py makes the two calls -- same bits)
        lf = 0; jf = 0
        if (it > 1) then
           find: do l = 1, ncomp
              if (desc(l)%cg_group /= grp(i)) cycle
              do j = 1, desc(l)%nindices
                 if (sample_index(j, l) /= 0 .and. pol_flag(j, l) == gflag(i)) then
                    lf = l; jf = j
                    exit find
                 end if
              end do
           end do find
        end if
        if (lf > 0) then
           if (iand(gflag(i), 1) /= 0) then
              map_n = 1
           else if (iand(gflag(i), 2) /= 0) then
              map_n = 2
           else if (iand(gflag(i), 4) /= 0) then
              map_n = 3
           else
              map_n = -1
           end if
           fused_l(i) = lf; fused_j(i) = jf
           if (it == niter) then
              call dangx_sky_amp_index_sample(sky, grp(i), gflag(i), DANGX_ML_SAMPLE, DANGX_FLUCT_REFERENCE, seed, &
                   dangx_stream_id(it, 0, grp(i), 0, gflag(i)), lf-1, jf-1, map_n, nsample, seed, &
                   dangx_stream_id(it, 1, lf-1, jf-1, gflag(i)), nbad, nacc)
              if (nbad /= 0) stop 3
              nacc_tot = nacc_tot + nacc
           else
              call dangx_sky_amp_index_sample(sky, grp(i), gflag(i), DANGX_ML_SAMPLE, DANGX_FLUCT_REFERENCE, seed, &
                   dangx_stream_id(it, 0, grp(i), 0, gflag(i)), lf-1, jf-1, map_n, nsample, seed, &
                   dangx_stream_id(it, 1, lf-1, jf-1, gflag(i)))
           end if
           cycle
        end if
        if (it == niter) then                             ! with the count: synchronises per context
           call dangx_sky_amp_sample(sky, grp(i), gflag(i), DANGX_ML_SAMPLE, DANGX_FLUCT_REFERENCE, seed, &
                dangx_stream_id(it, 0, grp(i), 0, gflag(i)), nbad)
           if (nbad /= 0) stop 3
        else                                              ! enqueue on every context, no wait
           call dangx_sky_amp_sample(sky, grp(i), gflag(i), DANGX_ML_SAMPLE, DANGX_FLUCT_REFERENCE, seed, &
                dangx_stream_id(it, 0, grp(i), 0, gflag(i)))
        end if
     end do
     if (it > 1) then                                     ! sample_spectral_parameters, src/dang_sample_mod.f90:21-86
        do l = 1, ncomp
           paired = .false.
           do j = 1, desc(l)%nindices
              if (paired) then                            ! went with the index before it
                 paired = .false.
                 cycle
              end if
              if (sample_index(j, l) == 0) cycle
              f = pol_flag(j, l)
              if (any(fused_l(1:ngroups) == l .and. fused_j(1:ngroups) == j .and. gflag(1:ngroups) == f)) cycle  ! done with its solve
              ! two consecutive sampled indices of one component on the same planes: one entry point (one launch where the
              ! register chain covers both); the Python host of the test makes the two calls -- same bits
              if (j < desc(l)%nindices) then
                 paired = sample_index(j+1, l) /= 0 .and. pol_flag(j+1, l) == f
              end if
              if (iand(f, 1) /= 0) then
                 map_n = 1
              else if (iand(f, 2) /= 0) then
                 map_n = 2
              else if (iand(f, 4) /= 0) then
                 map_n = 3
              else
                 map_n = -1
              end if
              if (paired) then
                 if (it == niter) then
                    call dangx_sky_index_sample_pair(sky, l-1, j-1, map_n, nsample, DANGX_ML_SAMPLE, seed, &
                         dangx_stream_id(it, 1, l-1, j-1, f), dangx_stream_id(it, 1, l-1, j, f), nacc, nacc2)
                    nacc_tot = nacc_tot + nacc + nacc2
                 else
                    call dangx_sky_index_sample_pair(sky, l-1, j-1, map_n, nsample, DANGX_ML_SAMPLE, seed, &
                         dangx_stream_id(it, 1, l-1, j-1, f), dangx_stream_id(it, 1, l-1, j, f))
                 end if
              else if (it == niter) then
                 call dangx_sky_index_sample(sky, l-1, j-1, map_n, nsample, DANGX_ML_SAMPLE, seed, &
                      dangx_stream_id(it, 1, l-1, j-1, f), nacc)
                 nacc_tot = nacc_tot + nacc
              else
                 call dangx_sky_index_sample(sky, l-1, j-1, map_n, nsample, DANGX_ML_SAMPLE, seed, &
                      dangx_stream_id(it, 1, l-1, j-1, f))
              end if
           end do
        end do
        ! the two chi^2 values the reference prints per iteration, from the sweeps' by-products (no pass over the maps)
        chisq_amp = dangx_sky_chisq_cached(sky, 0, 1, int(nmaps), nump, ok_amp)
        chisq_idx = dangx_sky_chisq_cached(sky, 1, 1, int(nmaps), nump, ok_idx)
     end if
  end do
  call dangx_sky_wait(sky)

  ! ---- output side: what write_data (every iteration) and write_maps (every iter_out) read
  do l = 1, ncomp                                         ! dangx_pull_state
     if (desc(l)%nindices > 0) then
        call dangx_sky_get_state(sky, l-1, c_loc(cm(l)%amp), c_loc(cm(l)%ind))
     else
        call dangx_sky_get_state(sky, l-1, c_loc(cm(l)%amp), c_null_ptr)
     end if
  end do
  chi_map = 0.d0
  chisq_explicit = dangx_sky_chisq(sky, 1, int(nmaps), nump, c_loc(sky_model), c_loc(res_map), c_loc(chi_map))
  nmeans = 0
  means = 0.d0
  do l = 1, ncomp                                         ! mask_avg(c%indices(:,map_n,j), masks(:,1)), write_data :716-731
     do j = 1, desc(l)%nindices
        if (sample_index(j, l) == 0) cycle
        k = 1
        if (iand(pol_flag(j, l), 1) == 0) k = 2
        nmeans = nmeans + 1
        means(nmeans) = dangx_sky_index_mean(sky, l-1, j-1, k)
     end do
  end do

  ! ---- the pixel-sharded hook of an MPI driver: the device CG hands its dot products to a callback written in the
  ! driver's language (one rank here, so the sum over ranks is the identity); only meaningful on a single context
  iters = 0
  if (nctx == 1) then
     ncalls = 0
     call dangx_check(sky%ctx(1), dangx_set_allreduce(sky%ctx(1), c_funloc(smoke_allreduce), c_null_ptr, 1), 'dangx_set_allreduce')
     st = dangx_amp_sample(sky%ctx(1), grp(1), gflag(1), DANGX_ML_OPTIMIZE, DANGX_SOLVER_CG, DANGX_FLUCT_REFERENCE, seed, &
          dangx_stream_id(99, 0, grp(1), 0, gflag(1)), 4, 0.d0, iters, nbad)
     call dangx_check(sky%ctx(1), st, 'dangx_amp_sample(cg)')
  end if
  call dangx_sky_destroy(sky)

  open(newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace')
  write(u) chisq_explicit, chisq_amp, chisq_idx, real(nacc_tot, c_double), real(nmeans, c_double), means(1:2*MAXC)
  write(u) sky_model, res_map, chi_map
  do l = 1, ncomp
     write(u) cm(l)%amp
     if (desc(l)%nindices > 0) write(u) cm(l)%ind
  end do
  close(u)
  write(*,'(a,i0,a,es24.16,2(a,l1))') 'dangx_fsmoke ok: contexts = ', nctx, '  chisq = ', chisq_explicit, &
       '  cached after amp: ', ok_amp, '  after index: ', ok_idx
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
