"""CPU: the C-ABI library loads and exports every symbol include/vpt.h declares (no compute calls without a GPU);
the N-API addon builds against it when node headers are present."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "vpt.h")).read()
    return re.findall(r"VPT_API\s+[\w\s\*]+?\b(vpt_\w+)\s*\(", txt)


def test_library_exports_every_declared_symbol():
    from vpt_amd import _native as N
    lib_path = N.LIB_PATH
    if not os.path.exists(lib_path):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "vpt_amd", "csrc")])
    lib = C.CDLL(lib_path)
    names = declared_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), "libvpt_hip.so does not export %s" % n
    # the Python binding lists exactly the header's symbols
    assert sorted(N.SYMBOLS) == sorted(names)


def test_header_is_plain_c():
    """the boundary is a C ABI: compiles as C11 with gcc, no C++/torch types"""
    src = '#include "vpt.h"\nint main(void){ vpt_uniforms u; (void)u; return sizeof(vpt_uniforms) == 128 ? 0 : 1; }\n'
    exe = "/tmp/vpt_abi_check"
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-x", "c", "-", "-o", exe],
                   input=src.encode(), check=True)
    assert subprocess.call([exe]) == 0


def test_uniforms_struct_layout_matches_binding():
    from vpt_amd import _native as N
    assert C.sizeof(N.Uniforms) == 128
    assert N.Uniforms.rand_seed.offset == 64 and N.Uniforms.light_direction.offset == 92 and N.Uniforms.blur.offset == 108
    assert N.Uniforms.isovalue.offset == 112 and N.Uniforms.gradient_step.offset == 116 and N.Uniforms.threshold.offset == 120


def test_version_and_error_strings_without_gpu():
    from vpt_amd import _native as N
    L = N.lib()
    assert b"vpt" in L.vpt_version()
    assert isinstance(L.vpt_last_error(), bytes)
    # argument validation happens before any device call
    assert L.vpt_context_create(0, None) == -1
    assert b"null" in L.vpt_last_error()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from vpt_amd import _native as N
    monkeypatch.setattr(N, "_lib", None)
    monkeypatch.setattr(N, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        N.lib()
