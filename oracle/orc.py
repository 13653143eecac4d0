"""orc.py — Python handle on the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Wraps oracle/liboracle.so (ac_oracle.c + flat_walk.c, built by oracle/Makefile) with the
call shapes the tests need; `load_reference()` imports the REFERENCE ITSELF from
oracle/_ref/ (compiled from /root/reference by the same Makefile) when it is present.
"""
import ctypes as C
import importlib.util
import os
import subprocess
import sysconfig

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "ahocorasick" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))

_lib = None


def build(quiet=True):
    """make -C oracle: liboracle.so always, _ref/ only when /root/reference is present."""
    subprocess.check_call(["make", "-C", HERE, "all"], stdout=subprocess.DEVNULL if quiet else None)


def lib():
    global _lib
    if _lib is None:
        src_newer = (not os.path.exists(LIB)) or any(
            os.path.getmtime(os.path.join(HERE, f)) > os.path.getmtime(LIB)
            for f in ("ac_oracle.c", "flat_walk.c", "ppm_walk.c", os.path.join("..", "include", "acx_blob.h")))
        if src_newer:
            build()
        l = C.CDLL(LIB)
        P, I64, I32p = C.c_void_p, C.c_int64, C.POINTER(C.c_int32)
        l.orc_new.restype = P
        l.orc_free.argtypes = [P]
        l.orc_add_word.argtypes = [P, C.c_char_p, C.c_size_t, I64]
        l.orc_make_automaton.argtypes = [P]
        l.orc_kind.argtypes = [P]
        l.orc_count.argtypes = [P]
        l.orc_num_nodes.argtypes = [P]
        l.orc_longest_word.argtypes = [P]
        l.orc_get.argtypes = [P, C.c_char_p, C.c_size_t, C.POINTER(I64)]
        l.orc_iter.restype = I64
        l.orc_iter.argtypes = [P, P, I64, I64, C.c_int, I32p, I64, P, P, I64]
        l.orc_iter_long.restype = I64
        l.orc_iter_long.argtypes = [P, P, I64, I64, I64, P, P, I64]
        l.orc_iter_batch_count.restype = I64
        l.orc_iter_batch_count.argtypes = [P, P, P, I64, C.c_int]
        l.orc_iter_batch.restype = I64
        l.orc_iter_batch.argtypes = [P, P, P, I64, C.c_int, P, P, P, I64]
        l.flat_iter.restype = I64
        l.flat_iter.argtypes = [P, P, I64, I32p, I64, P, P, I64]
        l.flat_iter_itop.restype = I64
        l.flat_iter_itop.argtypes = [P, P, I64, I64, I32p, P, P, I64]
        l.ppm_iter.restype = I64
        l.ppm_iter.argtypes = [P, P, I64, C.c_int32, P, P, I64]
        l.ppm_check_hot.restype = I64
        l.ppm_check_hot.argtypes = [P]
        l.ppm_iter_fill.restype = I64
        l.ppm_iter_fill.argtypes = [P, P, I64, C.c_int32, P, P, I64, C.c_uint32]
        l.flat_iter_long.restype = I64
        l.flat_iter_long.argtypes = [P, P, I64, I64, P, P, I64]
        _lib = l
    return _lib


class Oracle:
    """The reference's algorithm restated (ac_oracle.c).  Values are integers."""

    def __init__(self):
        self._a = C.c_void_p(lib().orc_new())

    def __del__(self):
        try:
            lib().orc_free(self._a)
        except Exception:
            pass

    def add_word(self, key, value):
        return lib().orc_add_word(self._a, key, len(key), int(value)) == 1

    def make_automaton(self):
        return lib().orc_make_automaton(self._a)

    @property
    def kind(self):
        return lib().orc_kind(self._a)

    def __len__(self):
        return lib().orc_count(self._a)

    def num_nodes(self):
        return lib().orc_num_nodes(self._a)

    def _run(self, fn, hay, cap_guess):
        cap = cap_guess
        while True:
            e = np.empty(cap, dtype=np.int32)
            v = np.empty(cap, dtype=np.int32)
            n = fn(e.ctypes.data, v.ctypes.data, cap)
            if n < 0:
                raise AttributeError("oracle: automaton not finalised (code %d)" % n)
            if n <= cap:
                return e[:n], v[:n]
            cap = int(n)

    def iter_arrays(self, hay, start=0, end=None, ignore_ws=False, state=0, shift=0):
        """-> (end_index int32[], value int32[], final_state)"""
        end = len(hay) if end is None else end
        st = C.c_int32(state)
        buf = C.create_string_buffer(hay, len(hay))

        def fn(pe, pv, cap):
            st.value = state
            return lib().orc_iter(self._a, C.cast(buf, C.c_void_p), start, end, int(ignore_ws),
                                  C.byref(st), shift, pe, pv, cap)
        e, v = self._run(fn, hay, max(16, 2 * len(hay)))
        return e, v, st.value

    def iter(self, hay, start=0, end=None, **kw):
        e, v, _ = self.iter_arrays(hay, start, end, **kw)
        return list(zip(e.tolist(), v.tolist()))

    def iter_long(self, hay, start=0, end=None, shift=0):
        end = len(hay) if end is None else end
        buf = C.create_string_buffer(hay, len(hay))

        def fn(pe, pv, cap):
            return lib().orc_iter_long(self._a, C.cast(buf, C.c_void_p), start, end, shift, pe, pv, cap)
        e, v = self._run(fn, hay, max(16, len(hay) + 1))
        return list(zip(e.tolist(), v.tolist()))

    def batch_count(self, data, offsets, mode=0):
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        buf = np.frombuffer(data, dtype=np.uint8)
        return lib().orc_iter_batch_count(self._a, buf.ctypes.data, off.ctypes.data, len(off) - 1, mode)

    def batch_records(self, data, offsets, mode=0, threads=None):
        """-> (match_off int64[n+1], end int32[], value int32[]) for the whole batch, computed in C
        (ac_oracle.c:orc_iter_batch) on `threads` host threads (ctypes releases the GIL; the automaton
        is only read).  What the full-size parity tests compare the GPU result with."""
        from concurrent.futures import ThreadPoolExecutor
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        buf = np.frombuffer(data, dtype=np.uint8)
        n = len(off) - 1
        threads = threads or min(64, os.cpu_count() or 1)
        cuts = np.linspace(0, n, min(threads, max(1, n)) + 1).astype(np.int64)

        def work(k):
            a, b = int(cuts[k]), int(cuts[k + 1])
            sub = np.ascontiguousarray(off[a:b + 1] - off[a])
            base = buf[off[a]:off[b]] if off[b] > off[a] else np.zeros(1, dtype=np.uint8)
            base = np.ascontiguousarray(base)
            cap = max(1024, int(off[b] - off[a]) // 4)
            while True:
                mo = np.empty(b - a + 1, dtype=np.int64)
                e = np.empty(cap, dtype=np.int32)
                v = np.empty(cap, dtype=np.int32)
                t = lib().orc_iter_batch(self._a, base.ctypes.data, sub.ctypes.data, b - a, mode,
                                         mo.ctypes.data, e.ctypes.data, v.ctypes.data, cap)
                if t < 0:
                    raise AttributeError("oracle: automaton not finalised (code %d)" % t)
                if t <= cap:
                    return mo, e[:t], v[:t]
                cap = int(t)

        with ThreadPoolExecutor(max_workers=len(cuts) - 1) as ex:
            parts = list(ex.map(work, range(len(cuts) - 1)))
        mos, base = [], 0
        for mo, e, v in parts:
            mos.append(mo[:-1] + base)
            base += int(mo[-1])
        match_off = np.concatenate(mos + [np.array([base], dtype=np.int64)])
        return match_off, np.concatenate([p[1] for p in parts]), np.concatenate([p[2] for p in parts])

    def batch(self, data, offsets, mode=0):
        """-> (match_off int64[n+1], end int32[], value int32[]) for the whole batch."""
        off = np.asarray(offsets, dtype=np.int64)
        mv = memoryview(data)
        es, vs, mo = [], [], [0]
        for k in range(len(off) - 1):
            h = bytes(mv[off[k]:off[k + 1]])
            if mode == 0:
                e, v, _ = self.iter_arrays(h)
            else:
                r = self.iter_long(h)
                e = np.array([x[0] for x in r], dtype=np.int32)
                v = np.array([x[1] for x in r], dtype=np.int32)
            es.append(e)
            vs.append(v)
            mo.append(mo[-1] + len(e))
        cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, dtype=np.int32)
        return np.array(mo, dtype=np.int64), cat(es), cat(vs)


def flat_iter(blob, hay, state=0, index_base=0):
    """walk the FLAT image on the CPU (flat_walk.c) -> (list of (end, value), final_state)"""
    st = C.c_int32(state)
    cap = max(16, 2 * len(hay))
    while True:
        e = np.empty(cap, dtype=np.int32)
        v = np.empty(cap, dtype=np.int32)
        st.value = state
        n = lib().flat_iter(blob, hay, len(hay), C.byref(st), index_base, e.ctypes.data, v.ctypes.data, cap)
        if n < 0:
            raise RuntimeError("flat_iter failed: %d" % n)
        if n <= cap:
            return list(zip(e[:n].tolist(), v[:n].tolist())), st.value
        cap = int(n)


def flat_iter_itop(blob, hay, index_base=0):
    """walk the flat image with the implicit top-of-trie (flat_walk.c:flat_iter_itop);
    -> (list of (end, value), final_state).  Raises on any disagreement with the explicit table."""
    st = C.c_int32(0)
    cap = max(16, 2 * len(hay))
    while True:
        e = np.empty(cap, dtype=np.int32)
        v = np.empty(cap, dtype=np.int32)
        n = lib().flat_iter_itop(blob, hay, len(hay), index_base, C.byref(st), e.ctypes.data, v.ctypes.data, cap)
        if n < 0:
            raise RuntimeError("flat_iter_itop failed: %d" % n)
        if n <= cap:
            return list(zip(e[:n].tolist(), v[:n].tolist())), st.value
        cap = int(n)


def ppm_check_hot(blob):
    """cells of the position-parallel image whose 8-byte hot form disagrees with the 32-byte form (0 = consistent)"""
    return int(lib().ppm_check_hot(blob))


def ppm_iter(blob, hay, index_base=0, fill=0):
    """scan with the position-parallel image on the CPU (ppm_walk.c) -> list of (end, value);
    None when the blob carries no ppm section.  fill != 0: symbols that do not exist at a position (before the
    haystack, behind a byte of no key) read as pseudo-random ones in the filter and cell codes, as on the GPU."""
    cap = max(16, 2 * len(hay))
    while True:
        e = np.empty(cap, dtype=np.int32)
        v = np.empty(cap, dtype=np.int32)
        n = lib().ppm_iter_fill(blob, hay, len(hay), index_base, e.ctypes.data, v.ctypes.data, cap, fill)
        if n == -2:
            return None
        if n < 0:
            raise RuntimeError("ppm_iter failed: %d" % n)
        if n <= cap:
            return list(zip(e[:n].tolist(), v[:n].tolist()))
        cap = int(n)


def flat_iter_long(blob, hay, index_base=0):
    cap = len(hay) + 16
    e = np.empty(cap, dtype=np.int32)
    v = np.empty(cap, dtype=np.int32)
    n = lib().flat_iter_long(blob, hay, len(hay), index_base, e.ctypes.data, v.ctypes.data, cap)
    if n < 0:
        raise RuntimeError("flat_iter_long failed: %d" % n)
    return list(zip(e[:n].tolist(), v[:n].tolist()))


def load_reference():
    """Import the reference's own `ahocorasick` module from oracle/_ref/, or None if absent."""
    if not os.path.exists(REF_SO):
        return None
    spec = importlib.util.spec_from_file_location("ahocorasick", REF_SO)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
