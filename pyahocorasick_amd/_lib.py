"""ctypes binding of libacx.so — the C-ABI declared in include/acx.h.

The library is loaded lazily and LOUDLY: if libacx.so is missing or does not export a
symbol the header declares, importing users get an ImportError that says how to build it.
There is no Python/CPU fallback for any scan entry point.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libacx.so")

ACX_OK = 0
ACX_E_INVAL, ACX_E_NOMEM, ACX_E_STATE, ACX_E_HIP = -1, -2, -3, -4
ACX_E_UNSUPPORTED, ACX_E_FORMAT, ACX_E_NODEVICE = -5, -6, -7
ACX_SCAN_ALL, ACX_SCAN_LONG = 0, 1
ACX_SCAN_ASYNC = 1
ACX_SCAN_SKIP_WS = 2        # white space (0x09..0x0D, 0x20) never touches the automaton; indices stay those of the original bytes
ACX_BLOB_HEADER_BYTES = 256
# layout options of acx_flatten_ex (include/acx.h)
ACX_FLATTEN_NO_PPM, ACX_FLATTEN_WIDE, ACX_FLATTEN_NO_ITOP, ACX_FLATTEN_TABLE_HOST, ACX_FLATTEN_TABLE_DEVICE, ACX_FLATTEN_HOT12 = 1, 2, 4, 8, 16, 32


class AcxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libacx error %d: %s" % (code, msg))
        self.code = code


class AcxNoDevice(AcxError):
    pass


class Match(C.Structure):
    _fields_ = [("end_index", C.c_int32), ("value", C.c_int32)]


class ScanParams(C.Structure):
    _fields_ = [
        ("struct_bytes", C.c_uint32), ("mode", C.c_int32),
        ("dev_hay", C.c_void_p), ("hay_capacity", C.c_int64),
        ("dev_off", C.c_void_p), ("stride", C.c_int64), ("n_hay", C.c_int64),
        ("dev_init_state", C.c_void_p), ("dev_index_base", C.c_void_p),
        ("want_final_state", C.c_int32), ("timing", C.c_int32),
        ("variant", C.c_int32), ("flags", C.c_int32),
        ("min_hay_len", C.c_int32), ("reserved0", C.c_int32),
        ("dev_skip", C.c_void_p),
    ]


_P = C.c_void_p
_PP = C.POINTER(C.c_void_p)
_u8p = C.c_char_p
# name -> (restype, argtypes).  Must list EVERY function of include/acx.h
# (tests/test_capi_symbols.py parses the header and checks this table against it).
SIGNATURES = {
    "acx_last_error": (C.c_char_p, []),
    "acx_abi_version": (C.c_int, []),
    "acx_async_streams": (C.c_int, []),
    "acx_trie_new": (C.c_int, [_PP]),
    "acx_trie_free": (None, [_P]),
    "acx_trie_add_word": (C.c_int, [_P, _u8p, C.c_size_t, C.c_int64, C.POINTER(C.c_int)]),
    "acx_trie_add_words": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int, C.POINTER(C.c_int64)]),
    "acx_trie_get": (C.c_int, [_P, _u8p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "acx_trie_remove_word": (C.c_int, [_P, _u8p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "acx_trie_longest_prefix": (C.c_int, [_P, _u8p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "acx_trie_clear": (None, [_P]),
    "acx_trie_make_automaton": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "acx_trie_kind": (C.c_int, [_P]),
    "acx_trie_num_keys": (C.c_int64, [_P]),
    "acx_trie_num_nodes": (C.c_int64, [_P]),
    "acx_trie_longest_word": (C.c_int64, [_P]),
    "acx_trie_version": (C.c_int64, [_P]),
    "acx_trie_items": (C.c_int, [_P, _u8p, C.c_size_t, _u8p, C.c_size_t, C.c_int, C.c_int, _PP,
                                 C.POINTER(C.POINTER(C.c_int64)), C.POINTER(C.POINTER(C.c_int64)), C.POINTER(C.c_int64)]),
    "acx_trie_stats": (C.c_int, [_P] + [C.POINTER(C.c_int64)] * 6),
    "acx_trie_from_ref_pickle": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_int, C.c_int64, C.c_int,
                                           C.c_int, _PP, C.POINTER(C.c_int64)]),
    "acx_trie_to_ref_pickle": (C.c_int, [_P, C.c_int, C.c_size_t, C.c_int, C.c_int, _PP, C.POINTER(C.POINTER(C.c_size_t)),
                                         C.POINTER(C.c_size_t)]),
    "acx_trie_eow_values": (C.c_int, [_P, C.c_int, C.POINTER(C.POINTER(C.c_int64)), C.POINTER(C.c_int64)]),
    "acx_trie_from_ref_savefile": (C.c_int, [_P, C.c_size_t, C.c_int, _PP, _P, C.POINTER(C.POINTER(C.c_int64)),
                                             C.POINTER(C.POINTER(C.c_int64))]),
    "acx_trie_to_ref_savefile": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), _PP,
                                           C.POINTER(C.c_size_t)]),
    "acx_flatten": (C.c_int, [_P, _PP, C.POINTER(C.c_size_t)]),
    "acx_flatten_ex": (C.c_int, [_P, C.c_uint32, _PP, C.POINTER(C.c_size_t)]),
    "acx_set_host_group_bytes": (None, [C.c_int64]),
    "acx_blob_free": (None, [_P]),
    "acx_blob_long_trie": (C.c_int, [_P, C.c_size_t, _PP, _PP, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "acx_blob_validate": (C.c_int, [_P, C.c_size_t]),
    "acx_image_upload": (C.c_int, [_P, C.c_size_t, _PP]),
    "acx_image_adopt": (C.c_int, [_P, C.c_size_t, _P, _PP]),
    "acx_image_free": (None, [_P]),
    "acx_image_num_states": (C.c_int64, [_P]),
    "acx_image_num_classes": (C.c_int64, [_P]),
    "acx_image_nbytes": (C.c_size_t, [_P]),
    "acx_image_dev_ptr": (C.c_void_p, [_P]),
    "acx_image_itop_depth": (C.c_int, [_P]),
    "acx_image_table_dev_ptr": (C.c_void_p, [_P]),
    "acx_scan_batch": (C.c_int, [_P, C.POINTER(ScanParams), _PP, _P]),
    "acx_scan_plan": (C.c_int, [_P, C.POINTER(ScanParams)]),
    "acx_image_broadcast": (C.c_int, [_P, C.c_size_t, _P, C.c_int, C.c_int, _P, _PP]),
    "acx_result_wait": (C.c_int, [_P]),
    "acx_result_num_matches": (C.c_int64, [_P]),
    "acx_result_offsets_dev": (C.c_void_p, [_P]),
    "acx_result_matches_dev": (C.c_void_p, [_P]),
    "acx_result_final_state_dev": (C.c_void_p, [_P]),
    "acx_result_fetch_host": (C.c_int, [_P, _PP, _PP, _PP]),
    "acx_result_timing": (C.c_int, [_P] + [C.POINTER(C.c_float)] * 4),
    "acx_result_free": (None, [_P]),
    "acx_scan_host": (C.c_int, [_P, C.c_int, _P, _P, C.c_int64, _P, _P, _PP]),
    "acx_scan_host_nofinal": (C.c_int, [_P, C.c_int, _P, _P, C.c_int64, _P, _PP]),
    "acx_scan_host_ctx": (C.c_int, [_P, _P, _P, C.c_int64, _P, _P, _P, C.c_int32, _PP]),
    "acx_blob_long_pack": (C.c_int, [_P, C.c_size_t, _PP, C.POINTER(C.c_size_t)]),
    "acx_image_set_long": (C.c_int, [_P, _P, C.c_size_t, C.c_int]),
    "acx_image_long_state": (C.c_int, [_P]),
    "acx_trie_scan_host": (C.c_int, [_P, C.c_int, _P, _P, C.c_int64, _P, _P, _P, _P, C.c_int32, C.c_int, _PP]),
    "acx_set_host_walk_bytes": (None, [C.c_int64]),
    "acx_host_walk_bytes": (C.c_int64, []),
    "acx_host_walk_applies": (C.c_int, [C.c_int64]),
    "acx_host_walk_calls": (C.c_int64, []),
    "acx_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "acx_device_set": (C.c_int, [C.c_int]),
    "acx_dev_malloc": (C.c_int, [_PP, C.c_size_t]),
    "acx_dev_free": (None, [_P]),
    "acx_memcpy_h2d": (C.c_int, [_P, _P, C.c_size_t]),
    "acx_memcpy_d2h": (C.c_int, [_P, _P, C.c_size_t]),
    "acx_device_sync": (C.c_int, []),
}

_lib = None


def lib():
    """Return the loaded library (loading it on first use)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "pyahocorasick_amd: %s is missing. Build it with "
            "`python -m pyahocorasick_amd.build` (needs hipcc; cross-compiles for gfx950 without a GPU). "
            "There is no CPU fallback." % LIB_PATH)
    # ONE HIP runtime per process.  PyTorch wheels bundle their own libamdhip64 (same SONAME
    # as /opt/rocm's).  If torch is imported first, the dynamic loader resolves our
    # DT_NEEDED libamdhip64.so.7 to torch's already-loaded copy and both share one runtime
    # (device pointers and streams are then interchangeable).  Loaded the other way round the
    # process would end up with two runtimes — so when torch is installed but not yet
    # imported and the caller asked for it (ACX_WITH_TORCH=1), import it here first.
    import sys
    if os.environ.get("ACX_WITH_TORCH") == "1" and "torch" not in sys.modules:
        import torch  # noqa: F401
    try:
        l = C.CDLL(LIB_PATH)
    except OSError as e:
        raise ImportError("pyahocorasick_amd: cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(l, name)
        except AttributeError:
            raise ImportError("pyahocorasick_amd: %s does not export %s (stale build? rebuild with "
                              "`python -m pyahocorasick_amd.build --force`)" % (LIB_PATH, name))
        fn.restype = res
        fn.argtypes = args
    if l.acx_abi_version() != 4:
        raise ImportError("pyahocorasick_amd: libacx ABI version %d, binding expects 4" % l.acx_abi_version())
    _lib = l
    return l


def last_error():
    m = lib().acx_last_error()
    return m.decode("utf-8", "replace") if m else ""


def check(rc):
    """Raise the Python exception that matches the reference's error convention."""
    if rc == ACX_OK:
        return
    msg = last_error()
    if rc == ACX_E_NOMEM:
        raise MemoryError(msg)
    if rc == ACX_E_NODEVICE:
        raise AcxNoDevice(rc, msg)
    raise AcxError(rc, msg)


def device_count():
    n = C.c_int(0)
    rc = lib().acx_device_count(C.byref(n))
    return n.value if rc == ACX_OK else 0
