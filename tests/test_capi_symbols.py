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
    assert l.acx_abi_version() == 4


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
    """no GPU scan is ever answered by a CPU implementation behind the caller's back: with the walk over the host trie
    switched off (limit -1) a process without a device raises; a batch beyond what that walk takes (1 MiB) raises whatever
    the limit is — batches belong to the GPU"""
    if have_gpu:
        return
    import numpy as np
    import pytest
    import pyahocorasick_amd as acx
    A = acx.Automaton(acx.STORE_INTS)
    A.add_word(b"he", 1)
    A.make_automaton()
    l = _lib.lib()
    l.acx_set_host_walk_bytes(-1)
    try:
        assert l.acx_host_walk_applies(3) == 0
        with pytest.raises(acx.AcxError):
            list(A.iter(b"she"))
        with pytest.raises(acx.AcxError):
            A.iter_batch([b"she"])
    finally:
        l.acx_set_host_walk_bytes(2048)
    assert l.acx_host_walk_applies(3) == 1 and l.acx_host_walk_applies((1 << 20) + 1) == 0
    big = np.full((1 << 20) + 64, ord("x"), dtype=np.uint8)
    with pytest.raises(acx.AcxError):
        A.scan_batch(big, [0, big.size])
    r = C_scan_host_too_large(A, big)
    assert r == _lib.ACX_E_UNSUPPORTED


def C_scan_host_too_large(A, big):
    import ctypes as C
    import numpy as np
    off = np.array([0, big.size], dtype=np.int64)
    res = C.c_void_p()
    rc = _lib.lib().acx_trie_scan_host(A._trie, 0, big.ctypes.data, off.ctypes.data, 1, None, None, None, None, 0, 0, C.byref(res))
    if res:
        _lib.lib().acx_result_free(res)
    return rc
