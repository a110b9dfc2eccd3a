"""CPU-only: the C-ABI library loads and exports every symbol include/gpupoly.h declares."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "gpupoly.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(gpu_[a-zA-Z0-9_]+|gpupoly_[a-zA-Z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    from mxx_amd import _ffi

    lib = C.CDLL(_ffi.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 45
    for s in syms:
        assert hasattr(lib, s), f"libgpupoly.so does not export {s}"
    # and the python binding table covers exactly the header
    assert sorted(_ffi.SIGNATURES) == syms


def test_reference_ffi_surface_is_covered():
    """Every function the reference's Rust side binds (src/poly/dcrt/gpu.rs:69-240)."""
    bound = """gpu_context_create gpu_context_destroy gpu_context_get_N gpu_event_set_wait gpu_event_set_destroy
    gpu_matrix_create gpu_matrix_destroy gpu_matrix_copy gpu_matrix_load_rns_batch gpu_matrix_store_rns_batch
    gpu_matrix_store_const_coeff_batch gpu_matrix_store_compact_bytes gpu_matrix_load_compact_bytes gpu_matrix_add
    gpu_matrix_add_block gpu_matrix_sub gpu_matrix_mul gpu_matrix_equal gpu_matrix_mul_scalar gpu_matrix_copy_block
    gpu_matrix_fill_gadget gpu_matrix_fill_small_gadget gpu_matrix_fill_small_decomposed_identity_chunk
    gpu_matrix_decompose_base gpu_matrix_decompose_base_small gpu_matrix_gauss_samp_gq_arb_base
    gpu_matrix_create_p1_covariance_cache gpu_matrix_destroy_p1_covariance_cache gpu_matrix_sample_p1_full_cached
    gpu_matrix_sample_distribution gpu_matrix_sample_distribution_columns gpu_matrix_ntt_all gpu_matrix_intt_all
    gpu_device_synchronize gpu_device_count gpu_device_mem_info gpu_last_error gpu_pinned_alloc gpu_pinned_free""".split()
    syms = set(header_symbols())
    for b in bound:
        assert b in syms, b


def test_errors_without_gpu_are_reported_not_thrown():
    from mxx_amd import _ffi

    lib = _ffi.lib()
    n = C.c_int(-1)
    rc = lib.gpu_device_count(C.byref(n))
    if rc != 0 or n.value == 0:
        # no device here: context creation must fail with a message, never crash
        raw = C.c_void_p()
        mod = (C.c_uint64 * 1)(12289)
        ids = (C.c_int * 1)(0)
        assert lib.gpu_context_create(2, 0, 1, mod, 1, ids, 1, C.byref(raw)) != 0
        assert _ffi.last_error_string()
    assert lib.gpu_set_last_error(b"hello") != 0
    assert _ffi.last_error_string() == "hello"
    assert lib.gpu_context_create(2, 0, 1, None, 0, None, 0, None) != 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from mxx_amd import _ffi

    monkeypatch.setattr(_ffi, "_lib", None)
    monkeypatch.setattr(_ffi, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_ffi.GpuPolyError):
        _ffi.lib()


def test_header_compiles_as_c_and_links(tmp_path):
    """include/gpupoly.h is a plain C header (what a cgo / bindgen / FFI consumer parses): a C translation unit that
    takes the address of every declared entry point compiles with gcc -std=c99 -pedantic and links against libgpupoly."""
    import subprocess

    from mxx_amd import _ffi

    syms = header_symbols()
    src = tmp_path / "link_all.c"
    body = "\n".join(f"    p[{i}] = (void (*)(void))&{s};" for i, s in enumerate(syms))
    src.write_text('#include "gpupoly.h"\n#include <stdio.h>\nint main(void) {\n    void (*p[%d])(void);\n%s\n'
                   '    GpuBatchOp op; GpuRngSeed seed; op.kind = GPUPOLY_OP_ADD; seed.words[0] = 1; (void)op; (void)seed;\n'
                   '    printf("%%d symbols, version %%s\\n", %d, gpupoly_version());\n    return p[0] == 0;\n}\n' % (len(syms), body, len(syms)))
    exe = tmp_path / "link_all"
    libdir = os.path.dirname(_ffi.LIB_PATH)
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-Wno-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
           "-L", libdir, "-lgpupoly", "-L/opt/rocm/lib", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, env=dict(os.environ, LD_LIBRARY_PATH=f"{libdir}:/opt/rocm/lib"))
    assert run.returncode == 0 and f"{len(syms)} symbols" in run.stdout, (run.stdout, run.stderr[-2000:])
