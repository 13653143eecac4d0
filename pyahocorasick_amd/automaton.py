"""`Automaton` — host-side mirror of the reference's `ahocorasick.Automaton` (bytes build)
for the accelerated path: add_word / make_automaton / iter / iter_long / find_all, plus
the batch entry points the reference lacks (iter_batch / scan_batch).

Same names, argument meaning and error behaviour as the reference so that the parity
tests read like the reference's own tests:
  * constructor            src/Automaton.c:96-181
  * add_word               src/Automaton.c:201-300   (value rules per `store`)
  * make_automaton         src/Automaton.c:560-649   (None when built, False otherwise)
  * iter                   src/Automaton.c:875-966 + src/AutomatonSearchIter.c
  * iter_long              src/Automaton.c:969-1041 + src/AutomatonSearchIterLong.c
  * find_all               src/Automaton.c:652-719
  * [start, [end]] parsing src/utils.c:293-359 (pymod_parse_start_end)

The trie and its failure links are built on the CPU inside libacx (C++).  Every search
runs on the GPU through the C-ABI (include/acx.h); there is NO CPU search path here —
without a GPU, iter()/iter_long()/find_all()/iter_batch() raise.
"""
import ctypes as C
import functools
import os
import threading

import numpy as np

from . import _lib
from ._lib import ACX_SCAN_ALL, ACX_SCAN_LONG, ACX_SCAN_SKIP_WS, AcxError, lib, check

# constants of the reference module, src/pyahocorasick.c:113-134, src/Automaton.h:16-41
EMPTY, TRIE, AHOCORASICK = 0, 1, 2
STORE_INTS, STORE_LENGTH, STORE_ANY = 10, 20, 30
KEY_STRING, KEY_SEQUENCE = 100, 200
MATCH_EXACT_LENGTH, MATCH_AT_MOST_PREFIX, MATCH_AT_LEAST_PREFIX = 0, 1, 2
unicode = 0   # this engine implements the bytes build (one letter = one byte)


def _parse_start_end(args, lo, hi):
    """pymod_parse_start_end, src/utils.c:293-359 (including its `len-1+end` quirk)."""
    start, end = lo, hi
    if len(args) >= 1:
        start = args[0].__index__()
        if start < 0:
            start = hi + start
        if start < lo or start >= hi:
            raise IndexError("start index not in range %d..%d" % (lo, hi))
    if len(args) >= 2:
        end = args[1].__index__()
        if end < 0:
            end = hi - 1 + end
        if end < lo or end > hi:
            raise IndexError("end index not in range %d..%d" % (lo, hi))
    return start, end


class _DeviceImage:
    """The flat automaton uploaded to HBM, tagged with the trie version it was built from."""

    def __init__(self, trie, version, flags=0):
        blob = C.c_void_p()
        nbytes = C.c_size_t()
        check(lib().acx_flatten_ex(trie, flags, C.byref(blob), C.byref(nbytes)))
        self.flags = flags
        self.handle = C.c_void_p()
        try:
            check(lib().acx_image_upload(blob, nbytes.value, C.byref(self.handle)))
        finally:
            lib().acx_blob_free(blob)
        self.version = version
        self.nbytes = nbytes.value

    def close(self):
        if self.handle:
            lib().acx_image_free(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BatchResult:
    """CSR result of a batch scan: matches of haystack k are rows offsets[k]:offsets[k+1].

    `end_index` / `value` are int32 arrays exactly as the reference emits them
    (Py_BuildValue("ii"), src/AutomatonSearchIter.c:180-184); for STORE_ANY automata
    `value` holds value ids — use `objects()` or `tolists()` to get the stored objects.
    """

    def __init__(self, offsets, end_index, value, final_state, values_table):
        self.offsets = offsets
        self.end_index = end_index
        self.value = value
        self.final_state = final_state
        self._values = values_table

    def __len__(self):
        return len(self.offsets) - 1

    def num_matches(self):
        return int(self.offsets[-1])

    def objects(self):
        if self._values is None:
            return self.value
        return [self._values[i] for i in self.value.tolist()]

    def tolists(self):
        ends = self.end_index.tolist()
        vals = self.objects() if self._values is not None else self.value.tolist()
        off = self.offsets.tolist()
        return [list(zip(ends[off[k]:off[k + 1]], vals[off[k]:off[k + 1]])) for k in range(len(off) - 1)]


def _locked(fn):
    """one mutation or scan at a time per Automaton (ctypes drops the GIL inside libacx calls)"""
    @functools.wraps(fn)
    def wrapper(self, *a, **kw):
        with self._lock:
            return fn(self, *a, **kw)
    return wrapper


def _extension():
    """the CPython extension `ahocorasick` (dropin/, bytes build): the product's host side"""
    import importlib.util
    from .build import dropin_path
    path = dropin_path()
    if not os.path.exists(path):
        raise ImportError("pyahocorasick_amd: %s is missing; build it with `python -m pyahocorasick_amd.build`" % path)
    spec = importlib.util.spec_from_file_location("ahocorasick", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class Automaton:
    """The ctypes mirror of the drop-in's `ahocorasick.Automaton` (csrc/ahocorasick_module.cpp is the product's host side; tests and
    bench.py drive this class for its numpy-level batch API: scan_batch, flat_image_bytes, flatten_flags, save_image).

    KEY_SEQUENCE automata are the EXTENSION's: `Automaton(store, KEY_SEQUENCE)` returns an instance of the extension's type, not of this
    class — isinstance(A, pyahocorasick_amd.Automaton) is False for it, a subclass of this class loses its type there, and the
    batch methods named above do not exist on it (iter / iter_long / find_all / keys / pickling do, as in the reference).  It needs the
    built extension (`python -m pyahocorasick_amd.build`; ImportError otherwise).  tests/test_key_sequence.py pins all of this."""

    def __new__(cls, *args):
        # KEY_SEQUENCE (src/utils.c:238-289: keys and haystacks are tuples of integers) lives in ONE place, the extension:
        # this class hands such automata to it (the numpy-level batch methods below are for byte automata)
        kt = args[3] if len(args) == 7 else (args[1] if len(args) >= 2 and isinstance(args[0], int) and isinstance(args[1], int) else KEY_STRING)
        if kt == KEY_SEQUENCE and (len(args) == 7 or args[0] in (STORE_INTS, STORE_LENGTH, STORE_ANY)):
            return _extension().Automaton(*args)
        return super().__new__(cls)

    def __init__(self, *args):
        """Automaton([store, [key_type]]) — or the 7-tuple of `__reduce__`
        (bytes_list, kind, store, key_type, count, longest_word, values), which is how pickles of
        this class AND of the reference's bytes build are loaded (src/Automaton.c:97-181)."""
        self._trie = C.c_void_p()
        self._image = None
        self.flatten_flags = 0                               # layout options of the flat image (acx_flatten_ex, ACX_FLATTEN_*): 0 = the library's choice
        self._result = C.c_void_p()                          # reusable device/pinned buffers
        self._lock = threading.RLock()
        self._free_slots = []                                # STORE_ANY: value slots freed by remove_word / pop
        pickled = None
        if len(args) == 7:
            pickled = args
            store, key_type = args[2], args[3]
            if not all(isinstance(a, int) for a in args[1:6]):
                raise ValueError("Unable to load from pickle.")
            if args[1] not in (EMPTY, TRIE, AHOCORASICK):
                raise ValueError("kind value must be one of ahocorasick.EMPTY, TRIE or AHOCORASICK")
        else:
            store, key_type = STORE_ANY, KEY_STRING
            if len(args) >= 1 and isinstance(args[0], int):
                store = args[0]
                if len(args) >= 2 and isinstance(args[1], int):
                    key_type = args[1]
        if store not in (STORE_INTS, STORE_LENGTH, STORE_ANY):
            raise ValueError("store value must be one of ahocorasick.STORE_LENGTH, STORE_INTS or STORE_ANY")
        if key_type not in (KEY_STRING, KEY_SEQUENCE):
            raise ValueError("key_type must have value KEY_STRING or KEY_SEQUENCE")
        self._store = store
        self._key_type = key_type
        self._values = [] if store == STORE_ANY else None   # STORE_ANY: value id -> object
        if pickled is not None and pickled[1] != EMPTY:
            self._load_pickle(*pickled)
        else:
            check(lib().acx_trie_new(C.byref(self._trie)))

    # ---- persistence in the reference's formats (csrc/acx_persist.cpp) ----------
    def _load_pickle(self, bytes_list, kind, store, key_type, count, longest_word, values):
        if type(bytes_list) is not list:
            raise TypeError("Expected list")
        for k, b in enumerate(bytes_list):
            if type(b) is not bytes:
                raise ValueError("Item #%d on the bytes list is not a bytes object" % k)
        any_ = store == STORE_ANY
        if any_ and not isinstance(values, list):
            raise ValueError("Unable to load from pickle.")
        n = len(bytes_list)
        ptrs = (C.c_void_p * n)(*[C.cast(C.c_char_p(b), C.c_void_p) for b in bytes_list])
        sizes = (C.c_size_t * n)(*[len(b) for b in bytes_list])
        n_eow = C.c_int64(0)
        try:
            check(lib().acx_trie_from_ref_pickle(ptrs, sizes, n, 1 if any_ else 0, longest_word, 2, 0, C.byref(self._trie), C.byref(n_eow)))
        except AcxError as e:
            raise ValueError(str(e)) from None
        if any_:
            if len(values) < n_eow.value:
                raise IndexError("list index out of range")          # PyList_GetItem in automaton_unpickle
            self._values = list(values[:n_eow.value])
        if kind == AHOCORASICK:
            self.make_automaton()

    def __reduce__(self):
        """same tuple as the reference (src/Automaton_pickle.c:192-285): loadable by either side"""
        if len(self) == 0:
            return (type(self), ())
        any_ = self._store == STORE_ANY
        buf, sizes, n = C.c_void_p(), C.POINTER(C.c_size_t)(), C.c_size_t()
        check(lib().acx_trie_to_ref_pickle(self._trie, 1 if any_ else 0, 0, 2, 0, C.byref(buf), C.byref(sizes), C.byref(n)))
        try:
            chunks, at = [], 0
            for k in range(n.value):
                chunks.append(C.string_at(buf.value + at, sizes[k]))
                at += sizes[k]
        finally:
            lib().acx_blob_free(buf)
            lib().acx_blob_free(C.cast(sizes, C.c_void_p))
        values = [self._values[i] for i in self._eow_values()] if any_ else None
        return (type(self), (chunks, self.kind, self._store, self._key_type, len(self),
                             lib().acx_trie_longest_word(self._trie), values))

    def _eow_values(self):
        vals, n = C.POINTER(C.c_int64)(), C.c_int64()
        check(lib().acx_trie_eow_values(self._trie, 0, C.byref(vals), C.byref(n)))
        try:
            return [vals[i] for i in range(n.value)]
        finally:
            lib().acx_blob_free(C.cast(vals, C.c_void_p))

    def save(self, *args):
        """save(path[, serializer]) in the reference's file format (src/custompickle/save/automaton_save.c)"""
        any_ = self._store == STORE_ANY
        if len(args) != (2 if any_ else 1):                            # src/custompickle/pyhelpers.c:8-18
            raise ValueError("expected exactly two arguments" if any_ else "expected exactly one argument")
        if not isinstance(args[0], str):
            raise TypeError("the first argument must be a string")
        if any_ and not callable(args[1]):
            raise TypeError("the second argument must be a callable object")
        payloads = None
        if any_:
            payloads = []
            for i in self._eow_values():
                b = args[1](self._values[i])
                if type(b) is not bytes:
                    raise TypeError("serializer must return bytes object")
                payloads.append(b)
        n = len(payloads) if payloads else 0
        ptrs = (C.c_void_p * max(n, 1))(*[C.cast(C.c_char_p(b), C.c_void_p) for b in (payloads or [])])
        sizes = (C.c_size_t * max(n, 1))(*[len(b) for b in (payloads or [])])
        buf, nbytes = C.c_void_p(), C.c_size_t()
        check(lib().acx_trie_to_ref_savefile(self._trie, self._store, self._key_type, 2, 0, ptrs, sizes, C.byref(buf), C.byref(nbytes)))
        try:
            with open(args[0], "wb") as f:
                f.write(C.string_at(buf, nbytes.value))
        finally:
            lib().acx_blob_free(buf)

    def __del__(self):
        try:
            if self._image is not None:
                self._image.close()
            if self._result:
                lib().acx_result_free(self._result)
            if self._trie:
                lib().acx_trie_free(self._trie)
        except Exception:
            pass

    # ---- attributes (src/Automaton.c:1237-1256) ---------------------------------------
    @property
    def kind(self):
        return lib().acx_trie_kind(self._trie)

    @property
    def store(self):
        return self._store

    @property
    def _version(self):
        return lib().acx_trie_version(self._trie)

    def _longest_word(self):
        return int(lib().acx_trie_longest_word(self._trie))

    def __len__(self):
        return lib().acx_trie_num_keys(self._trie)

    def __contains__(self, key):
        return self.exists(key)

    # ---- trie API (CPU, inside libacx) ------------------------------------------------
    @staticmethod
    def _key(key, what="bytes expected"):
        if not isinstance(key, bytes):
            raise TypeError(what)
        return key

    @_locked
    def add_word(self, key, *value):
        key = self._key(key)
        if self._store == STORE_ANY:
            if not value:
                raise ValueError("A value object is required as second argument.")
            found = C.c_int(0)
            old = C.c_int64(0)
            check(lib().acx_trie_get(self._trie, key, len(key), C.byref(found), C.byref(old)))
            if len(key) == 0:
                return False
            if found.value:
                vid = old.value
                self._values[vid] = value[0]
            elif self._free_slots:                     # reuse a slot that remove_word / pop freed
                vid = self._free_slots.pop()
                self._values[vid] = value[0]
            else:
                vid = len(self._values)
                self._values.append(value[0])
            v = vid
        elif self._store == STORE_INTS:
            if value:
                if not hasattr(value[0], "__index__") and not isinstance(value[0], (int, float)):
                    raise TypeError("An integer value is required as second argument.")
                v = int(value[0])
            else:
                v = len(self) + 1              # src/Automaton.c:238-242
        else:
            v = len(key)                       # STORE_LENGTH, src/Automaton.c:245-247
        is_new = C.c_int(0)
        v &= 0xFFFFFFFFFFFFFFFF
        if v >= 1 << 63:
            v -= 1 << 64
        check(lib().acx_trie_add_word(self._trie, key, len(key), v, C.byref(is_new)))
        return bool(is_new.value)

    @_locked
    def add_words(self, keys, values=None):
        """Many keys in one call (not in the reference): the same as add_word(k, v) for every pair in order, without
        a Python call per key — one acx_trie_add_words for STORE_INTS / STORE_LENGTH.  Returns how many keys were new."""
        keys = keys if isinstance(keys, list) else list(keys)
        if not all(map(bytes.__instancecheck__, keys)):
            raise TypeError("bytes expected")
        if values is not None and not isinstance(values, (np.ndarray, range)):
            values = values if isinstance(values, list) else list(values)   # (a generator is read once; what fails below must not leave it half consumed)
        if values is not None and len(values) != len(keys):
            raise ValueError("add_words: %d keys, %d values" % (len(keys), len(values)))
        if self._store == STORE_ANY:
            if values is None:
                raise ValueError("A value object is required as second argument.")
            return sum(1 for k, v in zip(keys, values) if self.add_word(k, v))
        off = np.zeros(len(keys) + 1, dtype=np.int64)
        np.cumsum(np.fromiter(map(len, keys), dtype=np.int64, count=len(keys)), out=off[1:])
        buf = np.frombuffer(b"".join(keys), dtype=np.uint8) if off[-1] else np.zeros(1, dtype=np.uint8)
        vals = None
        if self._store == STORE_INTS and values is not None:
            try:                                # (a range, an array, a list of ints that fit 64 bits: no Python arithmetic per value)
                vals = np.ascontiguousarray(values if isinstance(values, np.ndarray) else np.fromiter(values, dtype=np.int64), dtype=np.int64)
            except (OverflowError, TypeError, ValueError):
                vals = np.array([((int(v) + (1 << 63)) % (1 << 64)) - (1 << 63) for v in values], dtype=np.int64)   # wraps like the reference's C long
            if len(vals) != len(keys):
                raise ValueError("add_words: %d keys, %d values" % (len(keys), len(vals)))
        n_new = C.c_int64(0)
        check(lib().acx_trie_add_words(self._trie, buf.ctypes.data, off.ctypes.data, vals.ctypes.data if vals is not None else None,
                                       len(keys), 1 if self._store == STORE_INTS else 2, C.byref(n_new)))
        return int(n_new.value)

    def exists(self, key):
        key = self._key(key)
        found = C.c_int(0)
        check(lib().acx_trie_get(self._trie, key, len(key), C.byref(found), None))
        return bool(found.value)

    _NO_DEFAULT = object()

    def get(self, key, default=_NO_DEFAULT):
        key = self._key(key)
        found = C.c_int(0)
        val = C.c_int64(0)
        check(lib().acx_trie_get(self._trie, key, len(key), C.byref(found), C.byref(val)))
        if not found.value:
            if default is Automaton._NO_DEFAULT:
                raise KeyError(key)
            return default
        return self._values[val.value] if self._store == STORE_ANY else val.value

    def longest_prefix(self, key):
        key = self._key(key)
        n = C.c_size_t(0)
        check(lib().acx_trie_longest_prefix(self._trie, key, len(key), C.byref(n)))
        return n.value

    @_locked
    def _remove(self, key):
        key = self._key(key)
        found = C.c_int(0)
        val = C.c_int64(0)
        check(lib().acx_trie_remove_word(self._trie, key, len(key), C.byref(found), C.byref(val)))
        if not found.value:
            return False, None
        if self._store == STORE_ANY:
            obj = self._values[val.value]
            self._values[val.value] = None
            self._free_slots.append(val.value)
            return True, obj
        return True, val.value

    def remove_word(self, key):
        return self._remove(key)[0]

    def pop(self, key):
        ok, v = self._remove(key)
        if not ok:
            raise KeyError(key)
        return v

    @_locked
    def clear(self):
        lib().acx_trie_clear(self._trie)
        if self._values is not None:
            self._values = []
            self._free_slots = []
        self._drop_image()

    @_locked
    def make_automaton(self):
        changed = C.c_int(0)
        check(lib().acx_trie_make_automaton(self._trie, C.byref(changed)))
        return None if changed.value else False    # src/Automaton.c:574-575, 645

    # ---- device image -----------------------------------------------------------------
    def _drop_image(self):
        if self._image is not None:
            self._image.close()
            self._image = None

    def _ensure_image(self):
        v = self._version
        if self._image is None or self._image.version != v or self._image.flags != self.flatten_flags:
            self._drop_image()                      # invalidated by version (SURVEY §7 "Invalidation")
            self._image = _DeviceImage(self._trie, v, self.flatten_flags)
        return self._image

    def flat_image_bytes(self, flags=None):
        """The flat image (include/acx_blob.h) as bytes — what a single RCCL broadcast replicates.  flags: layout options
        of acx_flatten_ex (ACX_FLATTEN_*; default: this automaton's `flatten_flags`)."""
        blob = C.c_void_p()
        nbytes = C.c_size_t()
        check(lib().acx_flatten_ex(self._trie, self.flatten_flags if flags is None else flags, C.byref(blob), C.byref(nbytes)))
        try:
            return bytes((C.c_char * nbytes.value).from_address(blob.value))   # (string_at takes a C int)
        finally:
            lib().acx_blob_free(blob)

    def save_image(self, path):
        """write the flat image to a file; pyahocorasick_amd.device.Image.from_file(path) scans with it
        without rebuilding a trie.  Integer values only (STORE_INTS / STORE_LENGTH): object values of a
        STORE_ANY automaton are ids into this process's value list and do not travel."""
        if self._store == STORE_ANY:
            raise ValueError("save_image() needs STORE_INTS or STORE_LENGTH: object values cannot be stored in the image")
        with open(path, "wb") as f:
            f.write(self.flat_image_bytes())

    # ---- batch scan (NEW: the reference scans one haystack per iterator) ---------------
    def scan_batch(self, data, offsets, mode=ACX_SCAN_ALL, init_state=None, index_base=None, context=None, skip_white_space=False,
                   final_states=True):
        """Scan haystacks data[offsets[k]:offsets[k+1]] on the GPU; returns a BatchResult.

        final_states=False (no init_state): the result carries no final states (acx_scan_host_nofinal) — what a caller that
        does not continue the walk asks for; ACX_SCAN_LONG then runs position-parallel.

        data: bytes-like; offsets: int64[n+1] with offsets[0] == 0.
        context (ACX_SCAN_ALL): (bytes-like, int64[n+1]) — the bytes a stream delivered in front of every haystack
        (the last longest_word - 1 matter): matches may begin in there, none that ends in there is reported
        (acx_scan_host_ctx); this is how iter().set() continues without a state.
        skip_white_space (ACX_SCAN_ALL): iter(..., ignore_white_space=True) for the whole batch — the device takes the
        white space out before the scan and maps the end indices back (ACX_SCAN_SKIP_WS).
        """
        if self.kind != AHOCORASICK:
            raise AttributeError("Not an Aho-Corasick automaton yet: call add_word to add some keys and call "
                                 "make_automaton to convert the trie to an automaton.")
        buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        n = len(off) - 1
        if n < 0 or off[0] != 0 or (n and off[-1] > buf.size):
            raise ValueError("bad offsets")
        init = None if init_state is None else np.ascontiguousarray(init_state, dtype=np.int32)
        base = None if index_base is None else np.ascontiguousarray(index_base, dtype=np.int32)
        if (init is not None and init.shape != (n,)) or (base is not None and base.shape != (n,)):
            raise ValueError("init_state and index_base need one entry per haystack")
        # ctypes drops the GIL inside every libacx call, and the image and the result buffers belong to this
        # Automaton: one scan (image refresh, kernels, fetch) at a time per object.  The reference holds the
        # GIL for the whole of every call, so sharing an automaton between threads is safe there too.
        if skip_white_space and (mode != ACX_SCAN_ALL or init is not None):
            raise ValueError("skip_white_space is for ACX_SCAN_ALL without carried states")
        ctx = None
        if context is None and skip_white_space:
            context = (b"", np.zeros(n + 1, dtype=np.int64))
        if context is not None:
            if mode != ACX_SCAN_ALL or init is not None:
                raise ValueError("context is for ACX_SCAN_ALL without carried states")
            cbuf = np.frombuffer(context[0], dtype=np.uint8) if not isinstance(context[0], np.ndarray) else np.ascontiguousarray(context[0], dtype=np.uint8)
            coff = np.ascontiguousarray(context[1], dtype=np.int64)
            if coff.shape != (n + 1,) or coff[0] != 0 or coff[-1] > cbuf.size:
                raise ValueError("bad context offsets")
            ctx = (cbuf, coff)
        with self._lock:
            return self._scan_locked(buf, off, n, mode, init, base, ctx, ACX_SCAN_SKIP_WS if skip_white_space else 0,
                                     final_states=final_states or init is not None)

    def _host_walk(self, n, off, mode, init):
        """Does this scan go to the walk over the host trie (acx_trie_scan_host: BASELINE config 1, no device, haystacks
        that do not pay a launch)?  A carried state decides alone: negative = a node of the host trie, positive = a
        state of the device image.  ACX_SCAN_ALL with carried states is a device-only form (contexts replace it)."""
        if init is not None and init.size and (init != 0).any():
            return mode == ACX_SCAN_LONG and bool((init <= 0).all())
        return bool(lib().acx_host_walk_applies(int(off[n]) if n else 0))

    def _scan_locked(self, buf, off, n, mode, init, base, ctx=None, flags=0, final_states=True):
        if self._host_walk(n, off, mode, init):
            check(lib().acx_trie_scan_host(self._trie, mode, buf.ctypes.data if buf.size else None, off.ctypes.data, n,
                                           (ctx[0].ctypes.data if ctx[0].size else C.c_void_p(1)) if ctx is not None else None,
                                           ctx[1].ctypes.data if ctx is not None else None,
                                           init.ctypes.data if (init is not None and mode == ACX_SCAN_LONG) else None,
                                           base.ctypes.data if base is not None else None, flags,
                                           1 if final_states else 0, C.byref(self._result)))
            return self._fetch_result(n)
        img = self._ensure_image()
        if ctx is None and not final_states:
            check(lib().acx_scan_host_nofinal(img.handle, mode, buf.ctypes.data if buf.size else None, off.ctypes.data, n,
                                              base.ctypes.data if base is not None else None, C.byref(self._result)))
        elif ctx is not None:
            check(lib().acx_scan_host_ctx(img.handle, buf.ctypes.data if buf.size else None, off.ctypes.data, n,
                                          ctx[0].ctypes.data if ctx[0].size else C.c_void_p(1), ctx[1].ctypes.data,
                                          base.ctypes.data if base is not None else None, flags, C.byref(self._result)))
        else:
            check(lib().acx_scan_host(img.handle, mode, buf.ctypes.data if buf.size else None, off.ctypes.data, n,
                                      init.ctypes.data if init is not None else None,
                                      base.ctypes.data if base is not None else None,
                                      C.byref(self._result)))
        return self._fetch_result(n)

    def _fetch_result(self, n):
        p_off, p_m, p_f = C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(lib().acx_result_fetch_host(self._result, C.byref(p_off), C.byref(p_m), C.byref(p_f)))
        total = lib().acx_result_num_matches(self._result)
        r_off = np.ctypeslib.as_array(C.cast(p_off, C.POINTER(C.c_int64)), shape=(n + 1,)).copy()
        if total:
            m = np.ctypeslib.as_array(C.cast(p_m, C.POINTER(C.c_int32)), shape=(total, 2)).copy()
        else:
            m = np.zeros((0, 2), dtype=np.int32)
        fin = None
        if p_f and n:
            fin = np.ctypeslib.as_array(C.cast(p_f, C.POINTER(C.c_int32)), shape=(n,)).copy()
        return BatchResult(r_off, m[:, 0], m[:, 1], fin, self._values)

    def iter_batch(self, haystacks, long=False):
        """list of bytes -> list of lists of (end_index, value); == [list(A.iter(h)) for h in haystacks]."""
        lens = np.fromiter((len(h) for h in haystacks), dtype=np.int64, count=len(haystacks))
        off = np.zeros(len(haystacks) + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        for h in haystacks:
            if not isinstance(h, bytes):
                raise TypeError("bytes required")
        return self.scan_batch(b"".join(haystacks), off, ACX_SCAN_LONG if long else ACX_SCAN_ALL, final_states=False).tolists()

    # ---- dict-like enumeration (src/Automaton.c:722-873, src/AutomatonItemsIter.c) ---------------
    def _items_iter(self, args, what):
        if len(args) > 3:
            raise TypeError("at most 3 arguments: [prefix, [wildcard, [how]]]")
        pattern = b""
        if len(args) >= 1:
            pattern = self._key(args[0])
        use_wildcard, wildcard = False, 0
        if len(args) >= 2:
            w = self._key(args[1])
            if len(w) != 1:
                raise ValueError("Wildcard must be a single character.")
            use_wildcard, wildcard = True, w[0]
        how = MATCH_EXACT_LENGTH if use_wildcard else MATCH_AT_LEAST_PREFIX
        if len(args) >= 3:
            how = args[2].__index__()
            if how not in (MATCH_EXACT_LENGTH, MATCH_AT_MOST_PREFIX, MATCH_AT_LEAST_PREFIX):
                raise ValueError("The optional how third argument must be one of: "
                                 "MATCH_EXACT_LENGTH, MATCH_AT_LEAST_PREFIX or MATCH_AT_LEAST_PREFIX")
        return AutomatonItemsIter(self, pattern, use_wildcard, wildcard, how, what)

    def keys(self, *args):
        return self._items_iter(args, "keys")

    def values(self, *args):
        return self._items_iter(args, "values")

    def items(self, *args):
        return self._items_iter(args, "items")

    def __iter__(self):
        return self._items_iter((), "keys")

    def match(self, key):
        """True iff `key` is a prefix of some key (src/Automaton.c:460-479)"""
        key = self._key(key)
        if self.kind == EMPTY:             # trie_find on a NULL root (src/trie.c:136-152)
            return False
        return self.longest_prefix(key) == len(key)

    def dump(self):
        """(nodes, edges, fail) like the reference (src/Automaton.c:1098-1180): nodes = [(id, eow)],
        edges = [(id, letter, child id)], fail = [(id, fail id)], in pre-order; ids are 1-based dump
        numbers here (the reference prints node addresses)"""
        if self.kind == EMPTY:
            return None
        import struct
        chunks = self.__reduce__()[1][0] if len(self) else self._trie_chunks()
        nodes, edges, fail, nid = [], [], [], 0
        for chunk in chunks:
            n_nodes, = struct.unpack_from("<q", chunk, 0)
            at = 8
            for _ in range(n_nodes):
                nid += 1
                _out, f, n, eow = struct.unpack_from("<QQIB", chunk, at)
                at += 24
                nodes.append((nid, int(eow)))
                for j in range(n):
                    letter, child = struct.unpack_from("<HQ", chunk, at)
                    at += 10
                    edges.append((nid, bytes([letter & 0xFF]), child))
                if f:
                    fail.append((nid, f))
        return nodes, edges, fail

    def _trie_chunks(self):
        buf, sizes, n = C.c_void_p(), C.POINTER(C.c_size_t)(), C.c_size_t()
        check(lib().acx_trie_to_ref_pickle(self._trie, 0, 0, 2, 0, C.byref(buf), C.byref(sizes), C.byref(n)))
        try:
            out, at = [], 0
            for k in range(n.value):
                out.append(C.string_at(buf.value + at, sizes[k]))
                at += sizes[k]
            return out
        finally:
            lib().acx_blob_free(buf)
            lib().acx_blob_free(C.cast(sizes, C.c_void_p))

    def __sizeof__(self):
        return object.__sizeof__(self) + (self.get_stats()["total_size"] if self.kind != EMPTY else 0)

    def get_stats(self):
        v = [C.c_int64() for _ in range(6)]
        check(lib().acx_trie_stats(self._trie, *[C.byref(x) for x in v]))
        names = ("nodes_count", "words_count", "longest_word", "links_count", "sizeof_node", "total_size")
        return {k: x.value for k, x in zip(names, v)}

    # ---- reference search API, GPU-backed ---------------------------------------------
    def iter(self, string, start=-1, end=-1, ignore_white_space=False):
        if self.kind != AHOCORASICK:
            raise AttributeError("Not an Aho-Corasick automaton yet: call add_word to add some keys and call "
                                 "make_automaton to convert the trie to an automaton.")
        if not isinstance(string, bytes):
            raise TypeError("bytes required")
        # -1 means "default" for both (src/Automaton.c:893-956).  The reference does not
        # validate the range (out-of-range is undefined behaviour there); here it is clamped.
        s = 0 if start == -1 else start
        e = len(string) if end == -1 else end
        s = max(0, min(s, len(string)))
        e = max(s, min(e, len(string)))
        return AutomatonSearchIter(self, string, s, e, bool(ignore_white_space))

    def iter_long(self, string, *args):
        if self.kind != AHOCORASICK:
            raise AttributeError("not an automaton yet; add some words and call make_automaton")
        if not isinstance(string, bytes):
            raise TypeError("bytes required")
        s, e = _parse_start_end(args, 0, len(string))
        return AutomatonSearchIterLong(self, string, s, e)

    def find_all(self, string, callback, *args):
        if self.kind != AHOCORASICK:
            return None                                   # src/Automaton.c:666-667
        if not isinstance(string, bytes):
            raise TypeError("bytes expected")
        if not callable(callback):
            raise TypeError("The callback argument must be a callable such as a function.")
        s, e = _parse_start_end(args, 0, len(string))
        res = self.scan_batch(string[s:e], [0, max(0, e - s)], ACX_SCAN_ALL, index_base=[s])
        for idx, val in res.tolists()[0]:
            callback(idx, val)                            # an exception aborts, src/Automaton.c:705-708
        return None


_WS = np.zeros(256, dtype=bool)
_WS[[9, 10, 11, 12, 13, 32]] = True      # iswspace() over the letters the bytes build can produce


class AutomatonSearchIter:
    """iterator of (end_index, value); mirrors src/AutomatonSearchIter.c and, with long_mode,
    src/AutomatonSearchIterLong.c (same rules as the CPython extension, csrc/ahocorasick_module.cpp).

    The whole range is scanned on the GPU at construction / set(); next() hands the tuples
    out one by one and re-checks the automaton version like the reference (:247-250)."""

    _long = False

    def __init__(self, automaton, string, start, end, ignore_white_space=False):
        self._a = automaton
        self._version = automaton._version
        self._ws = ignore_white_space
        self._state = 0
        self._ctx = b""                  # iter: what the stream delivered before this chunk (its last longest_word - 1 letters)
        self._shift = 0
        self._load(string, start, end)

    def _letters(self, string, start, end):
        """the letters of string[start:end] that touch the automaton (ignore_white_space skips the others without
        touching the state, src/AutomatonSearchIter.c:269-274) and where each came from"""
        chunk = string[start:end]
        if not self._ws:
            return chunk, None
        arr = np.frombuffer(chunk, dtype=np.uint8)
        remap = np.flatnonzero(~_WS[arr])
        return arr[remap].tobytes(), remap

    def _tail(self, ctx, walked):
        keep = max(0, self._a._longest_word() - 1)
        t = ctx + walked
        return t[len(t) - keep:] if keep and len(t) > keep else (t if keep else b"")

    def _scan(self, string, start, end, state, shift):
        chunk = string[start:end]
        if self._long:
            res = self._a.scan_batch(chunk, [0, len(chunk)], ACX_SCAN_LONG, init_state=[state] if state else None,
                                     index_base=[start + shift])
        else:
            # iter continues a stream from the bytes before it, not from a state: every match that ends in this chunk
            # depends on the previous longest_word - 1 letters at most (the position-parallel kernels take it).
            # ignore_white_space: the device skips the white space and reports indices of the original bytes.
            res = self._a.scan_batch(chunk, [0, len(chunk)], ACX_SCAN_ALL, context=(self._ctx, [0, len(self._ctx)]),
                                     index_base=[start + shift], skip_white_space=self._ws)
        return res.tolists()[0], (int(res.final_state[0]) if res.final_state is not None else 0)

    def _load(self, string, start, end):
        self._src, self._start, self._state0 = string, start, self._state
        self._pending = []
        if self._version == self._a._version:             # a stale iterator raises from next(); its state id belongs to an old image
            self._pending, self._state = self._scan(string, start, end, self._state, self._shift)
        self._pos = 0
        self._end = end
        self._exhausted = False          # StopIteration seen: the reference has walked the whole chunk
        self._ref_index = start - 1     # the reference's iter->index (src/AutomatonSearchIter.c:123)

    def __iter__(self):
        return self

    def __next__(self):
        if self._version != self._a._version:
            raise ValueError("underlaying automaton has changed, iterator is not valid anymore")
        if self._pos >= len(self._pending):
            self._ref_index = self._end
            self._exhausted = True
            raise StopIteration
        item = self._pending[self._pos]
        self._pos += 1
        self._ref_index = item[0] - self._shift
        return item

    def set(self, string, reset=False):
        """src/AutomatonSearchIter.c:303-368 / src/AutomatonSearchIterLong.c:156-216: continue on a new chunk
        (keep the state, accumulate the shift) or reset to the root."""
        if not isinstance(string, bytes):
            raise TypeError("bytes expected")
        if reset:
            self._state = 0
            self._ctx = b""
            self._shift = 0
        else:
            # What the reference has walked of the old chunk: all of it after StopIteration, else up to the last match it
            # returned.  iter: those letters, behind the old context, are the new context.  iter_long: it is at the root
            # after every match (src/AutomatonSearchIterLong.c:101-110), or where the exhausted walk ended.
            if self._version == self._a._version:
                upto = self._end if self._exhausted else max(self._ref_index + 1, self._start)
                if self._long:
                    if not self._exhausted:
                        self._state = 0 if self._ref_index >= self._start else self._state0   # (nothing returned yet: untouched)
                else:
                    self._ctx = self._tail(self._ctx, self._letters(self._src, self._start, upto)[0])
            self._shift += self._ref_index if self._ref_index >= 0 else 0   # :344-352
        self._load(string, 0, len(string))


class AutomatonSearchIterLong(AutomatonSearchIter):
    """mirrors src/AutomatonSearchIterLong.c (longest, non-overlapping; see the oracle for quirks)."""

    _long = True

    def __init__(self, automaton, string, start, end):
        AutomatonSearchIter.__init__(self, automaton, string, start, end, False)


class _RefMeta(C.Structure):
    _fields_ = [("kind", C.c_int32), ("store", C.c_int32), ("key_type", C.c_int32), ("reserved", C.c_int32),
                ("count", C.c_int64), ("longest_word", C.c_int64), ("n_nodes", C.c_int64), ("n_eow", C.c_int64)]


def load(*args):
    """ahocorasick.load(path, deserializer): read a file written by Automaton.save — the
    reference's (bytes build) or this package's (src/custompickle/load/module_automaton_load.c).
    Both arguments are always required, as in the reference."""
    if len(args) != 2:
        raise ValueError("expected exactly two arguments")
    path, deserializer = args
    if not isinstance(path, str):
        raise TypeError("the first argument must be a string")
    if not callable(deserializer):
        raise TypeError("the second argument must be a callable object")
    with open(path, "rb") as f:                                   # IOError as in loadbuffer_open
        data = f.read()
    meta = _RefMeta()
    trie, poff, plen = C.c_void_p(), C.POINTER(C.c_int64)(), C.POINTER(C.c_int64)()
    try:
        check(lib().acx_trie_from_ref_savefile(data, len(data), 2, C.byref(trie), C.byref(meta), C.byref(poff), C.byref(plen)))
    except AcxError as e:
        raise ValueError(str(e)) from None
    try:
        if meta.store not in (STORE_INTS, STORE_LENGTH, STORE_ANY) or meta.key_type not in (KEY_STRING, KEY_SEQUENCE) \
                or meta.kind not in (EMPTY, TRIE, AHOCORASICK):
            raise ValueError("invalid header")
        A = Automaton(meta.store, meta.key_type)
        if trie:
            lib().acx_trie_free(A._trie)
            A._trie, trie = trie, C.c_void_p()
            if meta.store == STORE_ANY:
                A._values = [deserializer(data[poff[k]:poff[k] + plen[k]]) for k in range(meta.n_eow)]
            if meta.kind == AHOCORASICK:
                A.make_automaton()
        return A
    finally:
        if trie:
            lib().acx_trie_free(trie)
        if poff:
            lib().acx_blob_free(C.cast(poff, C.c_void_p))
        if plen:
            lib().acx_blob_free(C.cast(plen, C.c_void_p))


class AutomatonItemsIter:
    """keys() / values() / items() iterator (src/AutomatonItemsIter.c): the enumeration is done in
    libacx in the reference's order; like the reference's, the iterator dies when the automaton changes"""

    def __init__(self, automaton, pattern, use_wildcard, wildcard, how, what):
        self._a = automaton
        self._version = automaton._version
        self._what = what
        keys, koff, vals, n = C.c_void_p(), C.POINTER(C.c_int64)(), C.POINTER(C.c_int64)(), C.c_int64()
        wc = bytes([wildcard]) if use_wildcard else b""
        check(lib().acx_trie_items(automaton._trie, pattern, len(pattern), wc, len(wc), how, 0,
                                   C.byref(keys), C.byref(koff), C.byref(vals), C.byref(n)))
        try:
            self._n = n.value
            off = [koff[i] for i in range(self._n + 1)]
            blob = C.string_at(keys, off[-1]) if self._n else b""
            self._keys = [blob[off[i]:off[i + 1]] for i in range(self._n)]
            self._vals = [vals[i] for i in range(self._n)]
        finally:
            lib().acx_blob_free(keys)
            lib().acx_blob_free(C.cast(koff, C.c_void_p))
            lib().acx_blob_free(C.cast(vals, C.c_void_p))
        self._pos = 0

    def __iter__(self):
        return self

    def __next__(self):
        if self._version != self._a._version:
            raise ValueError("The underlying automaton has changed: this iterator is no longer valid.")
        if self._pos >= self._n:
            raise StopIteration
        i = self._pos
        self._pos += 1
        if self._what == "keys":
            return self._keys[i]
        v = self._vals[i]
        if self._a._values is not None:
            v = self._a._values[v]
        else:
            v = int(C.c_int32(v).value)                      # Py_BuildValue("i"): the low 32 bits
        return v if self._what == "values" else (self._keys[i], v)
