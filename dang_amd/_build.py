"""Build helpers: compile the HIP library (gfx950), the Fortran binding module and the oracle.

Everything is built IN-TREE (dang_amd/lib/, oracle/) so that the shared objects travel
with the source snapshot.  hipcc cross-compiles gfx950 without a GPU.
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dang_amd", "csrc")
LIBDIR = os.path.join(ROOT, "dang_amd", "lib")
LIB = os.path.join(LIBDIR, "libdangx.so")
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "libdang_oracle.so")
FORTRAN_DIR = os.path.join(ROOT, "fortran")
FLANG = "/opt/rocm/lib/llvm/bin/flang"


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd, **kw):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, **kw)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise RuntimeError("command failed: " + " ".join(cmd))
    return r.stdout


def hip_sources():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))] + [os.path.join(ROOT, "include", "dangx.h")]


UNITS = [("dangx_core", "dangx_core.hip", []), ("dangx_sky", "dangx_sky.hip", []), ("dangx_amp", "dangx_amp.hip", []), ("dangx_ampreg", "dangx_ampreg.hip", []), ("dangx_mixed", "dangx_mixed.hip", []), ("dangx_schur", "dangx_schur.hip", []),
         ("dangx_mh", "dangx_mh.hip", []),
         ("dangx_mhreg", "dangx_mhreg.hip", [])] + \
        [("dangx_mhreg_m%d" % m, "dangx_mhreg.hip", ["-DDX_REG_MODE=%d" % m]) for m in (1, 2, 3, 4, 5)] + \
        [("dangx_fused", "dangx_fused.hip", [])] + \
        [("dangx_fused_m%d" % m, "dangx_fused.hip", ["-DDX_REG_MODE=%d" % m]) for m in (1, 2, 3)]


def build_hip(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> dang_amd/lib/libdangx.so (+ kernel resource report).
    The translation units are compiled in parallel (objects under dang_amd/lib/obj/) and linked."""
    os.makedirs(LIBDIR, exist_ok=True)
    if not force and not _newer(LIB, hip_sources()):
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    common = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
              "-Rpass-analysis=kernel-resource-usage"]

    headers = [f for f in hip_sources() if f.endswith(".h")]

    def one(unit):
        name, srcfile, defs = unit
        obj, log = os.path.join(objdir, name + ".o"), os.path.join(objdir, name + ".log")
        src = os.path.join(CSRC, srcfile)
        if not force and os.path.exists(log) and not _newer(obj, [src] + headers):   # this unit is up to date
            with open(log) as f:
                return obj, f.read()
        out = _run(common + defs + ["-c", "-o", obj, src])
        with open(log, "w") as f:
            f.write(out)
        return obj, out

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        results = list(ex.map(one, UNITS))
    _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + [o for o, _ in results])
    out = "".join(t for _, t in results)
    with open(os.path.join(LIBDIR, "kernel_resource_usage.txt"), "w") as f:
        f.write(out)
    if verbose:
        print(out)
    return LIB


def build_oracle(force=False):
    """gcc -> oracle/libdang_oracle.so (the CPU checker; test infrastructure only)."""
    src = [os.path.join(ORACLE_DIR, "dang_oracle.c"), os.path.join(ORACLE_DIR, "dang_oracle.h")]
    if not force and not _newer(ORACLE_LIB, src):
        return ORACLE_LIB
    _run(["make", "-C", ORACLE_DIR, "-s"])
    return ORACLE_LIB


def build_fortran(force=False):
    """flang -> fortran/dangx_fsmoke (ISO_C_BINDING module + smoke driver), if flang is present."""
    if not os.path.exists(FLANG) or not os.path.isdir(FORTRAN_DIR):
        return None
    exe = os.path.join(FORTRAN_DIR, "dangx_fsmoke")
    src = [os.path.join(FORTRAN_DIR, f) for f in ("dangx_mod.f90", "dangx_multi_mod.f90", "dangx_fsmoke.f90")]
    if not all(os.path.exists(s) for s in src):
        return None
    if not force and not _newer(exe, src + [LIB]):
        return exe
    _run([FLANG, "-O2", "-J", FORTRAN_DIR, "-o", exe] + src +
         ["-L" + LIBDIR, "-ldangx", "-Wl,-rpath," + LIBDIR])
    return exe


def check_reference_side():
    """Type-check fortran/reference_side/dang_gpu_mod.f90 -- the wrapper with the reference's own signatures, which can
    only be COMPILED inside the reference's tree -- with flang against stub modules that declare just the names it
    touches (fortran/reference_side/stubs/).  Catches syntax / type errors; says nothing about the reference.
    Returns True when the check ran and passed, None without flang; raises on errors."""
    import tempfile
    ref = os.path.join(FORTRAN_DIR, "reference_side")
    if not os.path.exists(FLANG) or not os.path.isdir(ref):
        return None
    with tempfile.TemporaryDirectory() as tmp:
        for f in ("dangx_mod.f90", "dangx_multi_mod.f90"):
            _run([FLANG, "-c", "-J", tmp, os.path.join(FORTRAN_DIR, f), "-o", os.path.join(tmp, f + ".o")])
        _run([FLANG, "-c", "-J", tmp, os.path.join(ref, "stubs", "stubs.f90"), "-o", os.path.join(tmp, "stubs.o")])
        _run([FLANG, "-fsyntax-only", "-J", tmp, os.path.join(ref, "dang_gpu_mod.f90")])
    return True


def build_reference_drive(force=False):
    """flang -> fortran/reference_side/dang_gpu_drive: the reference-side wrapper (dang_gpu_mod.f90) compiled against the mock
    modules of stubs/stubs.f90 and linked with libdangx.so, plus the driver that plays `program dang` for it -- so that the
    wrapper RUNS on a GPU in the tests and in bench.py (`fortran_seam`).  Nothing of the reference is compiled."""
    ref = os.path.join(FORTRAN_DIR, "reference_side")
    if not os.path.exists(FLANG) or not os.path.isdir(ref):
        return None
    exe = os.path.join(ref, "dang_gpu_drive")
    src = [os.path.join(FORTRAN_DIR, "dangx_mod.f90"), os.path.join(FORTRAN_DIR, "dangx_multi_mod.f90"),
           os.path.join(ref, "stubs", "stubs.f90"), os.path.join(ref, "dang_gpu_mod.f90"), os.path.join(ref, "dang_gpu_drive.f90")]
    if not force and not _newer(exe, src + [LIB]):
        return exe
    moddir = os.path.join(ref, "mod")
    os.makedirs(moddir, exist_ok=True)
    _run([FLANG, "-O2", "-J", moddir, "-o", exe] + src + ["-L" + LIBDIR, "-ldangx", "-Wl,-rpath," + LIBDIR])
    return exe


def build_ubench(force=False):
    """hipcc -> dang_amd/lib/ub/{isa_rate,stream_planes}: the microbenchmarks behind profiles/r02_isa_rate.txt and
    profiles/r02_stream_planes.txt (measurement tools, not part of the library)."""
    out = os.path.join(LIBDIR, "ub")
    os.makedirs(out, exist_ok=True)
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    built = []
    for name in ("isa_rate", "stream_planes"):
        src = os.path.join(ROOT, "tools", "ubench", name + ".hip")
        exe = os.path.join(out, name)
        if os.path.exists(src) and (force or _newer(exe, [src])):
            _run([hipcc, "-O3", "--offload-arch=gfx950", "-o", exe, src])
        built.append(exe)
    return built


def build_all(force=False):
    build_hip(force)
    build_ubench(force)
    build_oracle(force)
    build_fortran(force)
    check_reference_side()
    build_reference_drive(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
    print("built:", LIB, ORACLE_LIB)
