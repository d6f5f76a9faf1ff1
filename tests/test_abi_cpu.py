"""The C-ABI shared library loads and exports every symbol include/dangx.h declares; the product
path fails loudly (no CPU fallback) when no GPU is present."""
import ctypes
import os
import re

import pytest

from dang_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    txt = open(os.path.join(ROOT, "include", "dangx.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dangx_[a-zA-Z_0-9]+)\s*\(", txt)))


def test_header_and_binding_agree(built):
    names = header_functions()
    assert len(names) >= 30
    assert sorted(L.SYMBOLS) == names


def test_fortran_module_binds_every_declared_symbol():
    """fortran/dangx_mod.f90 (the ISO_C_BINDING layer of the drop-in) binds exactly the header's exports."""
    f90 = open(os.path.join(ROOT, "fortran", "dangx_mod.f90")).read()
    bound = sorted(set(re.findall(r"bind\(C,\s*name='(dangx_\w+)'\)", f90)))
    assert bound == header_functions()


def test_library_exports_every_declared_symbol(built):
    lib = ctypes.CDLL(L.LIB_PATH)
    for n in header_functions():
        assert hasattr(lib, n), "libdangx.so does not export " + n
    L.load()
    assert b"gfx950" in L.load().dangx_version()


def test_run_time_specialisation_compiles_without_a_device(built):
    """hiprtc builds the register chain / the fused kernel for a band count the library has no instantiation of, from the
    headers embedded in the .so (compile only: no GPU needed)."""
    lib = L.load()
    for hdr, name in (("dx_kern_chain.h", "dxk::k_index_mh_reg<1, 1, 9, 1>"), ("dx_kern_chain.h", "dxk::k_index_mh_pair<2, 2, 7, 1>"),
                      ("dx_kern_fused.h", "dxk::k_amp_index<1, 2, 9, 4, 1>")):
        log = ctypes.create_string_buffer(8192)
        rc = lib.dangx_rtc_compile(hdr.encode(), name.encode(), log, len(log))
        assert rc == 0 and log.value.startswith(b"_ZN3dxk"), (name, log.value.decode())
    log = ctypes.create_string_buffer(8192)
    assert lib.dangx_rtc_compile(b"dx_kern_chain.h", b"dxk::no_such_kernel<1>", log, len(log)) != 0 and log.value


def test_struct_layouts_match_header():
    # dangx_dims: 4 x i32, 2 x i64, 2 x i32 ; dangx_comp_desc: 6 x i32, f64, 2x2 i32, 2x(2x2) f64, 2 f64
    assert ctypes.sizeof(L.Dims) == 4 * 4 + 2 * 8 + 2 * 4
    assert ctypes.sizeof(L.CompDesc) == 6 * 4 + 8 + 2 * 4 + 2 * 4 + 4 * 8 + 4 * 8 + 2 * 8


def test_library_is_hip_code_object(built):
    """The shipped .so carries a gfx950 code object (hipcc fat binary), i.e. it is the HIP path."""
    blob = open(L.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"k_amp_direct" in blob and b"k_index_mh" in blob


def test_no_cpu_fallback_without_gpu(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import dang_amd as da
    from dang_amd import synth
    dpar, ddata, bands, comps, meta = synth.make_sky("C1", nside=1)
    with pytest.raises(da.DangxError):
        da.Engine(bands, comps, ddata)


def test_product_does_not_touch_the_oracle():
    """Nothing under dang_amd/ (or bench's timed path) imports, links or opens oracle/."""
    for d, _, files in os.walk(os.path.join(ROOT, "dang_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")) and f != "_build.py":
                src = open(os.path.join(d, f), errors="ignore").read()
                assert "oracle_ffi" not in src and "dang_oracle" not in src, os.path.join(d, f)
