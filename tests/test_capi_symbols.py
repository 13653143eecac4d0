"""CPU suite: the C-ABI library loads and exports every symbol include/acx.h declares
(no compute calls — those need a GPU)."""
import ctypes
import os
import re

from pyahocorasick_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "acx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    names = re.findall(r"\b(acx_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_header_functions_all_exported_and_bound():
    names = declared_functions()
    assert len(names) >= 35
    dll = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(dll, n), "libacx.so does not export %s" % n
    assert sorted(_lib.SIGNATURES) == names, "ctypes binding table out of sync with include/acx.h"


def test_library_exports_the_c_abi_and_nothing_else():
    """-fvisibility=hidden + the visibility pragma of include/acx.h: no kernel launcher, no C++ internal is linkable"""
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode().split("\n")
    syms = [l.split()[-1] for l in out if l.strip()]
    exported = sorted(s for s in syms if not s.startswith(("__hip_", "_init", "_fini", "__bss", "_edata", "_end")))
    assert exported == declared_functions(), sorted(set(exported) ^ set(declared_functions()))


def test_library_loads_and_reports_abi():
    l = _lib.lib()
    assert l.acx_abi_version() == 3


def test_struct_layout_matches_header(tmp_path):
    """compile a C probe against include/acx.h and compare sizeof/offsetof with the ctypes mirror"""
    import subprocess
    fields = [f for f, _ in _lib.ScanParams._fields_]
    prog = '#include <stdio.h>\n#include <stddef.h>\n#include "acx.h"\nint main(void){\n'
    prog += 'printf("%zu %zu\\n", sizeof(acx_scan_params), sizeof(acx_match_t));\n'
    for f in fields:
        prog += 'printf("%%zu\\n", offsetof(acx_scan_params, %s));\n' % f
    prog += "return 0;}\n"
    src = tmp_path / "probe.c"
    src.write_text(prog)
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split()
    assert int(out[0]) == ctypes.sizeof(_lib.ScanParams) and int(out[1]) == ctypes.sizeof(_lib.Match) == 8
    for f, off in zip(fields, out[2:]):
        assert getattr(_lib.ScanParams, f).offset == int(off), f


def test_scan_without_gpu_fails_loudly(have_gpu):
    """the product path never falls back to a CPU implementation"""
    if have_gpu:
        return
    import pytest
    import pyahocorasick_amd as acx
    A = acx.Automaton(acx.STORE_INTS)
    A.add_word(b"he", 1)
    A.make_automaton()
    with pytest.raises(acx.AcxError):
        list(A.iter(b"she"))
    with pytest.raises(acx.AcxError):
        A.iter_batch([b"she"])
