! Mock modules for ../dang_gpu_mod.f90 (see README.md): the NAMES of the reference's modules with just the declarations the
! wrapper touches (types, globals, getlun).  Builder-owned test scaffolding for the wrapper -- no reference code, and nothing
! of the reference is compiled with them: ../dang_gpu_drive.f90 fills these types from a problem file and RUNS the wrapper.
module healpix_types                       ! HEALPix-F90 kinds
  integer, parameter :: i4b = selected_int_kind(9), i8b = selected_int_kind(16)
  integer, parameter :: sp = selected_real_kind(5,30), dp = selected_real_kind(12,200), lgt = kind(.true.)
end module healpix_types

module mpi                                 ! the few MPI names dang_util_mod re-exports (src/dang_util_mod.f90:7)
  integer, parameter :: MPI_COMM_WORLD = 0, MPI_DOUBLE_PRECISION = 1, MPI_SUM = 2, mpi_status_size = 5
  integer :: MPI_IN_PLACE
end module mpi

module dang_util_mod                       ! src/dang_util_mod.f90:12-44, 100-136
  use healpix_types
  use mpi
  implicit none
  real(dp)           :: T_CMB = 2.7255d0, missval = -1.6375d30
  integer(i4b)       :: ierr, rank, numprocs
  integer(i4b)       :: nbands, npix, nmaps, nside
  integer(i4b)       :: ncomp, ncg_groups, nsample
  integer(i4b)       :: iter
  integer(i4b)       :: master = 0
  integer(i4b)       :: nump
  logical(lgt)       :: exist
  character(len=80), dimension(3) :: tqu
  character(len=10)  :: ml_mode
contains
  function getlun()
    integer(i4b) :: getlun
    logical :: busy
    do getlun = 40, 99                      ! a free unit, as the reference's getlun hands out
       inquire(unit=getlun, opened=busy)
       if (.not. busy) return
    end do
  end function getlun
end module dang_util_mod

module dang_param_mod                      ! src/dang_param_mod.f90:7-24
  use healpix_types
  implicit none
  type dang_params
     character(len=512) :: outdir
     character(len=16)  :: ml_mode
     integer(i4b), allocatable, dimension(:) :: pol_type
  end type dang_params
end module dang_param_mod

module dang_bp_mod                         ! src/dang_bp_mod.f90:7-15
  use healpix_types
  implicit none
  type bandinfo
     character(len=20) :: label, id, unit
     integer(i4b)      :: n
     real(dp)          :: nu_c
     real(dp), allocatable, dimension(:) :: nu0, nu, tau0, tau
  end type bandinfo
  type(bandinfo), allocatable, dimension(:), target :: bp
end module dang_bp_mod

module dang_component_mod                  ! src/dang_component_mod.f90:12-65
  use healpix_types
  implicit none
  type :: dang_comps                       ! the fields of the reference's type that dang_gpu_mod.f90 reads or writes, by kind
     character(len=16) :: label, type
     integer(i4b)      :: cg_group, nfit, nindices
     real(dp)          :: nu_ref
     logical(lgt)      :: sample_amplitude
     logical(lgt),      allocatable :: corr(:), sample_index(:), tuned(:)
     character(len=16), allocatable :: ind_label(:), lnl_type(:), prior_type(:)
     integer(i4b),      allocatable :: sample_nside(:), index_mode(:), nflag(:), pol_flag(:,:)
     real(dp),          allocatable :: step_size(:), temp_norm(:)
     real(dp),          allocatable :: amplitude(:,:), template(:,:), template_amplitudes(:,:), gauss_prior(:,:), uni_prior(:,:)
     real(dp),          allocatable :: indices(:,:,:)
  end type dang_comps
  type component_pointer
     type(dang_comps), pointer :: p => null()
  end type component_pointer
  type(component_pointer), allocatable, dimension(:) :: component_list
end module dang_component_mod

module dang_data_mod                       ! src/dang_data_mod.f90:9-61
  use healpix_types
  implicit none
  type :: dang_data
     real(dp)                                     :: chisq
     character(len=32), allocatable, dimension(:) :: label
     integer(i4b), allocatable, dimension(:)      :: pol_type
     real(dp), allocatable, dimension(:,:,:)      :: sig_map, rms_map, res_map, sky_model
     real(dp), allocatable, dimension(:,:)        :: chi_map, masks
     real(dp), allocatable, dimension(:)          :: conversion, gain, offset
     logical(lgt), allocatable, dimension(:)      :: fit_gain
  end type dang_data
end module dang_data_mod

module dang_cg_mod                         ! src/dang_cg_mod.f90:16-52
  use healpix_types
  implicit none
  type :: dang_cg_group
     integer(i4b) :: cg_group, i_max, nflag, ntemp
     real(dp)     :: converge
     logical(lgt) :: sample
     integer(i4b), allocatable, dimension(:) :: pol_flag
  end type dang_cg_group
  type cg_pointer
     type(dang_cg_group), pointer :: p => null()
  end type cg_pointer
  type(cg_pointer), allocatable, dimension(:) :: cg_groups
end module dang_cg_mod
