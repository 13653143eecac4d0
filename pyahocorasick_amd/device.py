"""Device-resident batch scanning: thin Python handles over the C-ABI for callers that keep
their haystacks in HBM (bench.py, pipelines that receive reads straight into device
buffers, torch users passing `tensor.data_ptr()`).

Nothing here computes: it only owns handles and forwards raw pointers to libacx.
"""
import ctypes as C

import numpy as np

from ._lib import ACX_SCAN_ALL, ACX_SCAN_ASYNC, ACX_SCAN_SKIP_WS, ACX_BLOB_HEADER_BYTES, ScanParams, check, lib


class DeviceBuffer:
    """hipMalloc'ed bytes owned by libacx (no torch needed)."""

    def __init__(self, nbytes):
        self.ptr = C.c_void_p()
        self.nbytes = int(nbytes)
        check(lib().acx_dev_malloc(C.byref(self.ptr), self.nbytes))

    @classmethod
    def from_numpy(cls, arr, pad=0):
        arr = np.ascontiguousarray(arr)
        b = cls(arr.nbytes + pad)
        if arr.nbytes:
            check(lib().acx_memcpy_h2d(b.ptr, arr.ctypes.data, arr.nbytes))
        return b

    def to_numpy(self, dtype, count):
        out = np.empty(count, dtype=dtype)
        if out.nbytes:
            check(lib().acx_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            lib().acx_dev_free(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def long_pack(blob_bytes):
    """the dictionary of the position-parallel iter_long for the automaton in `blob_bytes`, as one relocatable pack
    (acx_blob_long_pack: host only) — what rides behind the blob in the set-up broadcast"""
    buf = (C.c_char * len(blob_bytes)).from_buffer_copy(blob_bytes) if not isinstance(blob_bytes, C.Array) else blob_bytes
    out, n = C.c_void_p(), C.c_size_t()
    check(lib().acx_blob_long_pack(buf, len(blob_bytes), C.byref(out), C.byref(n)))
    try:
        return bytes((C.c_char * n.value).from_address(out.value))
    finally:
        lib().acx_blob_free(out)


class Image:
    """A flat automaton resident in HBM.  Build from an Automaton (`Image.from_automaton`),
    from blob bytes, or adopt a device buffer that already holds the blob (the receive
    side of the RCCL broadcast)."""

    def __init__(self, handle, keepalive=None):
        self.handle = handle
        self._keepalive = keepalive

    @classmethod
    def from_blob(cls, blob_bytes):
        h = C.c_void_p()
        buf = (C.c_char * len(blob_bytes)).from_buffer_copy(blob_bytes)
        check(lib().acx_image_upload(buf, len(blob_bytes), C.byref(h)))
        return cls(h)

    @classmethod
    def from_automaton(cls, automaton, flags=None):
        """flatten + upload without routing the (possibly tens of GB) blob through Python bytes"""
        blob, nbytes, h = C.c_void_p(), C.c_size_t(), C.c_void_p()
        check(lib().acx_flatten_ex(automaton._trie, automaton.flatten_flags if flags is None else flags, C.byref(blob), C.byref(nbytes)))
        try:
            check(lib().acx_image_upload(blob, nbytes.value, C.byref(h)))
        finally:
            lib().acx_blob_free(blob)
        return cls(h)

    @classmethod
    def from_file(cls, path):
        """load a flat image written by Automaton.save_image() (the blob is relocatable: the file
        IS the image, include/acx_blob.h) — SURVEY §8f N3: no pointer trie is rebuilt"""
        with open(path, "rb") as f:
            data = f.read()
        buf = (C.c_char * len(data)).from_buffer_copy(data)
        check(lib().acx_blob_validate(buf, len(data)))
        h = C.c_void_p()
        check(lib().acx_image_upload(buf, len(data), C.byref(h)))
        return cls(h)

    @classmethod
    def adopt(cls, dev_ptr, nbytes, host_header, keepalive=None):
        h = C.c_void_p()
        hdr = (C.c_char * ACX_BLOB_HEADER_BYTES).from_buffer_copy(bytes(host_header[:ACX_BLOB_HEADER_BYTES]))
        check(lib().acx_image_adopt(C.c_void_p(dev_ptr), nbytes, hdr, C.byref(h)))
        return cls(h, keepalive)

    def set_long(self, pack, pack_bytes=None, on_device=False):
        """install the iter_long dictionary built once by long_pack() (acx_image_set_long): `pack` = host bytes, or — on_device — the
        device address of a pack that the caller keeps alive (the tail of the broadcast's receive buffer)"""
        if on_device:
            check(lib().acx_image_set_long(self.handle, C.c_void_p(int(pack)), int(pack_bytes), 1))
        else:
            buf = (C.c_char * len(pack)).from_buffer_copy(pack)
            check(lib().acx_image_set_long(self.handle, buf, len(pack), 0))

    @property
    def long_state(self):
        """1: the iter_long dictionary is installed, -1: the position-parallel form does not apply, 0: not built yet"""
        return lib().acx_image_long_state(self.handle)

    @property
    def num_states(self):
        return lib().acx_image_num_states(self.handle)

    @property
    def num_classes(self):
        return lib().acx_image_num_classes(self.handle)

    @property
    def nbytes(self):
        return lib().acx_image_nbytes(self.handle)

    @property
    def itop_depth(self):
        return lib().acx_image_itop_depth(self.handle)

    def ppm_kernel(self, stride=0, has_offsets=False, variant=0, dev_hay=0x1000, n_hay=1, min_hay_len=0, mode=ACX_SCAN_ALL):
        """which kernel family a scan of such a batch takes (acx_scan_plan): None = the serial walks, "scan" = k_ppm_scan,
        "stream" = k_ppm_stream, "stream4" = k_ppm_stream4 (four letters; fixed stride or offsets); mode ACX_SCAN_LONG: the family that
        scans the dictionary of the position-parallel iter_long (acx_long.cpp), None = the serial walk"""
        p = ScanParams()
        p.struct_bytes = C.sizeof(ScanParams)
        p.mode = mode
        p.dev_hay = dev_hay
        p.hay_capacity = max(1, int(stride)) * n_hay
        p.dev_off = 0x1000 if has_offsets else None
        p.stride = int(stride)
        p.n_hay = n_hay
        p.variant = int(variant)
        p.min_hay_len = int(min_hay_len)
        return {0: None, 1: "scan", 2: "stream", 3: "stream4", 11: "scan", 12: "stream", 13: "stream4"}.get(lib().acx_scan_plan(self.handle, C.byref(p)))

    def download_table(self):
        """the dense transition table the scans read, as uint32[n_states, n_classes] (tests/tools)"""
        n, K = self.num_states, self.num_classes
        out = np.empty((n, K), dtype=np.uint32)
        check(lib().acx_memcpy_d2h(out.ctypes.data, C.c_void_p(lib().acx_image_table_dev_ptr(self.handle)), out.nbytes))
        return out

    def free(self):
        if self.handle:
            lib().acx_image_free(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Scanner:
    """Reusable scan context: owns one acx_result_t (device + pinned buffers are reused
    across calls, so the steady state allocates nothing)."""

    def __init__(self, image):
        self.image = image
        self._res = C.c_void_p()
        self.n_hay = 0

    def scan(self, dev_hay, hay_capacity, n_hay, dev_off=None, stride=0, mode=ACX_SCAN_ALL,
             dev_init_state=None, dev_index_base=None, want_final_state=False, timing=False,
             variant=0, stream=None, asynchronous=False, min_hay_len=0, dev_skip=None, skip_white_space=False):
        """All pointer arguments are raw device addresses (int / c_void_p / None).
        asynchronous=True: return as soon as the kernels are queued on `stream` (returns None);
        `wait()`, `num_matches()`, `fetch()` complete the scan.
        skip_white_space=True: ACX_SCAN_SKIP_WS — white space never touches the automaton, indices stay those of the
        bytes given (iter(..., ignore_white_space=True) for a whole batch)."""
        p = ScanParams()
        p.struct_bytes = C.sizeof(ScanParams)
        p.mode = mode
        p.dev_hay = _addr(dev_hay)
        p.hay_capacity = int(hay_capacity)
        p.dev_off = _addr(dev_off)
        p.stride = int(stride)
        p.n_hay = int(n_hay)
        p.dev_init_state = _addr(dev_init_state)
        p.dev_index_base = _addr(dev_index_base)
        p.want_final_state = 1 if want_final_state else 0
        p.timing = 2 if (timing == 2 and timing is not True) else (1 if timing else 0)   # 2: events around the walk only
        p.variant = int(variant)
        p.min_hay_len = int(min_hay_len)
        p.dev_skip = _addr(dev_skip)                 # streams: context bytes in front of every haystack (include/acx.h)
        p.flags = (ACX_SCAN_ASYNC if asynchronous else 0) | (ACX_SCAN_SKIP_WS if skip_white_space else 0)
        check(lib().acx_scan_batch(self.image.handle, C.byref(p), C.byref(self._res), _addr(stream)))
        self.n_hay = int(n_hay)
        return None if asynchronous else lib().acx_result_num_matches(self._res)

    def wait(self):
        if self._res:
            check(lib().acx_result_wait(self._res))

    def timing_ms(self):
        w, s, e, t = C.c_float(), C.c_float(), C.c_float(), C.c_float()
        check(lib().acx_result_timing(self._res, C.byref(w), C.byref(s), C.byref(e), C.byref(t)))
        return {"walk": w.value, "scan": s.value, "expand": e.value, "total": t.value}

    def num_matches(self):
        return lib().acx_result_num_matches(self._res)

    def fetch(self):
        """-> (match_off int64[n+1], end_index int32[], value int32[], final_state or None) on the host"""
        p_off, p_m, p_f = C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(lib().acx_result_fetch_host(self._res, C.byref(p_off), C.byref(p_m), C.byref(p_f)))
        n, total = self.n_hay, self.num_matches()
        off = np.ctypeslib.as_array(C.cast(p_off, C.POINTER(C.c_int64)), shape=(n + 1,)).copy()
        if total:
            m = np.ctypeslib.as_array(C.cast(p_m, C.POINTER(C.c_int32)), shape=(total, 2)).copy()
        else:
            m = np.zeros((0, 2), dtype=np.int32)
        fin = None
        if p_f and n:
            fin = np.ctypeslib.as_array(C.cast(p_f, C.POINTER(C.c_int32)), shape=(n,)).copy()
        return off, m[:, 0], m[:, 1], fin

    def free(self):
        if self._res:
            lib().acx_result_free(self._res)
            self._res = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _addr(x):
    if x is None:
        return None
    if isinstance(x, C.c_void_p):
        return x
    if isinstance(x, DeviceBuffer):
        return x.ptr
    return C.c_void_p(int(x))
