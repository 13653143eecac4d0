// ahocorasick_module.cpp — CPython extension `ahocorasick`: the drop-in host side.
//
// Exports PyInit_ahocorasick and reproduces the Python-visible surface of the reference
// module for the accelerated path (reference src/pyahocorasick.c:67-137,
// method table src/Automaton.c:1206-1229): Automaton(store, key_type), add_word, exists, get,
// longest_prefix, remove_word, pop, clear, make_automaton, iter (+ .set), iter_long, find_all,
// len(), `in`, attributes kind / store, the module constants — plus the batch entry the
// reference lacks (iter_batch).  Same argument meaning and error behaviour as the reference
// (bytes build: keys and haystacks are `bytes`, ahocorasick.unicode == 0).
//
// Everything below the Python objects goes through the C-ABI of libacx (include/acx.h): the
// trie and its failure links live in acx_trie_t (CPU), every search is acx_scan_host (GPU).
// There is no CPU search path: without a GPU the search methods raise RuntimeError.
//
// Not a port: the reference's node graph, input widening, generator state machines and
// pickling are not here; iterators hold the finished match list of one GPU scan.
#define PY_SSIZE_T_CLEAN
#include <Python.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "acx.h"

namespace {

enum { K_EMPTY = 0, K_TRIE = 1, K_AHOCORASICK = 2 };
enum { STORE_INTS = 10, STORE_LENGTH = 20, STORE_ANY = 30 };
enum { KEY_STRING = 100, KEY_SEQUENCE = 200 };
enum { MATCH_EXACT_LENGTH = 0, MATCH_AT_MOST_PREFIX = 1, MATCH_AT_LEAST_PREFIX = 2 };

const char* NOT_AUTOMATON_MSG =
    "Not an Aho-Corasick automaton yet: call add_word to add some keys and call make_automaton to "
    "convert the trie to an automaton.";

struct AutomatonObject {
    PyObject_HEAD
    acx_trie_t* trie;
    int store;
    int key_type;
    PyObject* values;          // list: value id -> object (STORE_ANY only)
    struct ImageRef* image;    // device image, valid for image_version; shared with the scans in flight
    int64_t image_version;
    std::vector<acx_result_t*>* results;   // idle result objects (device / pinned buffers), reused by the next scan
    std::vector<Py_ssize_t>* free_slots;   // STORE_ANY: slots of `values` freed by remove_word / pop, reused by add_word
};

// A scan runs WITHOUT the GIL: other Python threads may meanwhile change the automaton, which replaces
// the device image.  The image a scan uses is therefore counted: it is freed by whoever drops the last use.  Every
// scan has a result object of its own (taken from the automaton's idle list, handed back when its records have been
// copied out), so two threads scanning one automaton share nothing but the immutable image.  All counters and lists
// are touched with the GIL held.
struct ImageRef { acx_image_t* img; int users; bool dead; };
inline void image_drop(ImageRef* r) { if (r && r->users == 0 && r->dead) { acx_image_free(r->img); delete r; } }
inline void image_retire(AutomatonObject* a) {
    if (!a->image) return;
    a->image->dead = true;
    image_drop(a->image);
    a->image = nullptr;
}
struct ScanLease {                 // what one scan holds until its caller has copied the records out
    AutomatonObject* a = nullptr; ImageRef* ref = nullptr; acx_result_t* res = nullptr;
    void done() {
        if (ref) { ref->users--; image_drop(ref); ref = nullptr; }
        if (res) {
            if (a->results && a->results->size() < 4) a->results->push_back(res); else acx_result_free(res);
            res = nullptr;
        }
    }
    ~ScanLease() { done(); }
};

PyObject* set_acx_error(int rc) {
    if (rc == ACX_E_NOMEM) return PyErr_NoMemory();
    PyErr_SetString(PyExc_RuntimeError, acx_last_error());
    return nullptr;
}

// ---- helpers -------------------------------------------------------------------------------
// ---- letters --------------------------------------------------------------------------------
// Two flavours of this module are built from this file, like the reference (src/common.h:50-67):
//   bytes build   (default)              keys and haystacks are `bytes`, one letter = one byte;
//   unicode build (-DACX_UNICODE_BUILD)   keys and haystacks are `str`, one letter = one code point.
// The engine underneath is the byte automaton in both: the unicode build feeds it UTF-8.  UTF-8 is
// prefix-free and self-synchronising, so a key's bytes can only match on code-point boundaries
// and the byte trie's failure links mirror the code-point trie's: same matches, and an index in
// bytes converts to an index in letters by counting the bytes that start a character.
#ifndef ACX_UNICODE_BUILD
#define ACX_UNICODE_BUILD 0
#endif

// dump layout of this flavour (acx_persist.cpp): uint16 letters (bytes build) or uint32 (unicode build);
// multi-byte letters (str keys, KEY_SEQUENCE) are decoded on the way out and re-encoded on the way in
#define ACX_LETTER_BYTES (ACX_UNICODE_BUILD ? 4 : 2)
inline int letters_multibyte(int key_type) { return (ACX_UNICODE_BUILD || key_type == KEY_SEQUENCE) ? 1 : 0; }

struct Text {
    const uint8_t* data;     // the bytes the engine sees
    Py_ssize_t nbytes;
    Py_ssize_t nchars;       // letters as the reference counts them
    std::vector<uint8_t> own;   // storage when the bytes had to be made (KEY_SEQUENCE); never copy a filled Text
    bool ascii() const { return nbytes == nchars; }
};

inline bool is_char_start(uint8_t b) { return (b & 0xC0) != 0x80; }

// one letter as a self-synchronising byte sequence: UTF-8, continued to 31 bits the way UTF-8 was
// first specified (5- and 6-byte forms) so that KEY_SEQUENCE letters fit as well
inline void encode_letter(uint32_t v, std::vector<uint8_t>& out) {
    if (v < 0x80) { out.push_back((uint8_t)v); return; }
    int n = v < 0x800 ? 2 : v < 0x10000 ? 3 : v < 0x200000 ? 4 : v < 0x4000000 ? 5 : 6;
    static const uint8_t lead[7] = {0, 0, 0xC0, 0xE0, 0xF0, 0xF8, 0xFC};
    out.push_back((uint8_t)(lead[n] | (v >> (6 * (n - 1)))));
    for (int k = n - 2; k >= 0; k--) out.push_back((uint8_t)(0x80 | ((v >> (6 * k)) & 0x3F)));
}

inline const uint8_t* decode_letter(const uint8_t* p, const uint8_t* e, uint32_t* v) {
    const uint8_t b = *p++;
    if (b < 0x80) { *v = b; return p; }
    int n = b >= 0xFC ? 6 : b >= 0xF8 ? 5 : b >= 0xF0 ? 4 : b >= 0xE0 ? 3 : 2;
    uint32_t x = b & (0xFF >> (n + 1));
    for (int k = 1; k < n && p < e; k++) x = (x << 6) | (*p++ & 0x3F);
    *v = x;
    return p;
}

// KEY_SEQUENCE (src/utils.c:238-289): a tuple of integers, one letter each
bool get_sequence(PyObject* o, Text* t, bool haystack) {
    if (!PyTuple_Check(o)) {
        PyErr_SetString(PyExc_TypeError, haystack ? "tuple required" : "argument is not a supported sequence type");
        return false;
    }
    const Py_ssize_t n = PyTuple_GET_SIZE(o);
    t->own.clear();
    t->own.reserve((size_t)n + 8);
    for (Py_ssize_t i = 0; i < n; i++) {
        const Py_ssize_t v = PyNumber_AsSsize_t(PyTuple_GET_ITEM(o, i), PyExc_ValueError);
        if (v == -1 && PyErr_Occurred()) { PyErr_Format(PyExc_ValueError, "item #%zd is not a number", i); return false; }
        // the reference's range and wording (src/utils.c:258-268); letters are stored in 31 bits here
        const unsigned long max_val = ACX_UNICODE_BUILD ? 4294967295ul : 65535ul;
        if (v < 0 || (unsigned long)v > max_val) {
            PyErr_Format(PyExc_ValueError, "item #%zd: value %zd outside range [%d..%lu]", i, v, 0, max_val);
            return false;
        }
        if ((unsigned long)v > 2147483647ul) {
            PyErr_Format(PyExc_ValueError, "item #%zd: value %zd is beyond the 31 bits this build stores per sequence letter", i, v);
            return false;
        }
        encode_letter((uint32_t)v, t->own);
    }
    t->own.push_back(0);                                             // never an empty vector: data() stays valid
    t->data = t->own.data(); t->nbytes = (Py_ssize_t)t->own.size() - 1; t->nchars = n;
    return true;
}

// haystack = true: the wording iter()/iter_long() use for a wrong argument type
bool get_text(PyObject* o, Text* t, bool haystack = false, int key_type = KEY_STRING) {
    if (key_type == KEY_SEQUENCE) return get_sequence(o, t, haystack);
#if ACX_UNICODE_BUILD
    if (!PyUnicode_Check(o)) { PyErr_SetString(PyExc_TypeError, haystack ? "string required" : "string expected"); return false; }
    Py_ssize_t n = 0;
    const char* u = PyUnicode_AsUTF8AndSize(o, &n);              // cached in the str object: lives as long as `o`
    if (!u) return false;
    t->data = (const uint8_t*)u; t->nbytes = n; t->nchars = PyUnicode_GET_LENGTH(o);
#else
    if (!PyBytes_Check(o)) { PyErr_SetString(PyExc_TypeError, haystack ? "bytes required" : "bytes expected"); return false; }
    t->data = (const uint8_t*)PyBytes_AS_STRING(o); t->nbytes = PyBytes_GET_SIZE(o); t->nchars = t->nbytes;
#endif
    return true;
}

// the prefix pattern / wildcard of keys() & co: always bytes (str in the unicode build), also for
// KEY_SEQUENCE automata (pymod_get_string, src/Automaton.c:746, 765) — whose letters are then the
// reference's widening of each byte (bytes build: sign-extended to uint16)
bool get_pattern(PyObject* o, Text* t, int key_type) {
    if (!get_text(o, t)) return false;
    if (key_type != KEY_SEQUENCE) return true;
    std::vector<uint8_t> enc;
    const uint8_t *p = t->data, *e = t->data + t->nbytes;
    Py_ssize_t n = 0;
    while (p < e) {
        uint32_t v;
#if ACX_UNICODE_BUILD
        p = decode_letter(p, e, &v);
#else
        v = (uint16_t)(int16_t)(int8_t)*p++;
#endif
        encode_letter(v, enc);
        n++;
    }
    enc.push_back(0);
    t->own.swap(enc);
    t->data = t->own.data(); t->nbytes = (Py_ssize_t)t->own.size() - 1; t->nchars = n;
    return true;
}

// byte offset of letter `ci` (0 <= ci <= nchars)
Py_ssize_t char_to_byte(const Text& t, Py_ssize_t ci) {
    if (t.ascii()) return ci;
    if (ci >= t.nchars) return t.nbytes;
    Py_ssize_t seen = -1;
    for (Py_ssize_t i = 0; i < t.nbytes; i++) {
        if (is_char_start(t.data[i]) && ++seen == ci) return i;
    }
    return t.nbytes;
}

// number of letters in the first `nb` bytes
Py_ssize_t chars_in(const uint8_t* p, Py_ssize_t nb) {
    Py_ssize_t c = 0;
    for (Py_ssize_t i = 0; i < nb; i++) c += is_char_start(p[i]);
    return c;
}

PyObject* make_key_object(const uint8_t* p, Py_ssize_t n, int key_type = KEY_STRING) {
    if (key_type == KEY_SEQUENCE) {
        // what the reference's keys() gives for a sequence automaton: its letter buffer as a string —
        // bytes build: the low byte of each letter (char_buffer, src/AutomatonItemsIter.c:211-216);
        // unicode build: a str of the letters taken as code points
        std::vector<uint32_t> letters;
        for (const uint8_t *q = p, *e = p + n; q < e;) { uint32_t v; q = decode_letter(q, e, &v); letters.push_back(v); }
#if ACX_UNICODE_BUILD
        return PyUnicode_FromKindAndData(PyUnicode_4BYTE_KIND, letters.data(), (Py_ssize_t)letters.size());
#else
        std::vector<char> low(letters.size() + 1);
        for (size_t i = 0; i < letters.size(); i++) low[i] = (char)(letters[i] & 0xFF);
        return PyBytes_FromStringAndSize(low.data(), (Py_ssize_t)letters.size());
#endif
    }
#if ACX_UNICODE_BUILD
    return PyUnicode_DecodeUTF8((const char*)p, n, "strict");
#else
    return PyBytes_FromStringAndSize((const char*)p, n);
#endif
}

// [start, [end]] exactly as pymod_parse_start_end does (src/utils.c:293-359), quirks included
bool parse_start_end(PyObject* args, Py_ssize_t i0, Py_ssize_t i1, Py_ssize_t lo, Py_ssize_t hi,
                     Py_ssize_t* start, Py_ssize_t* end) {
    *start = lo; *end = hi;
    if (PyTuple_GET_SIZE(args) > i0) {
        PyObject* o = PyNumber_Index(PyTuple_GET_ITEM(args, i0));
        if (!o) return false;
        Py_ssize_t v = PyNumber_AsSsize_t(o, PyExc_IndexError);
        Py_DECREF(o);
        if (v == -1 && PyErr_Occurred()) return false;
        if (v < 0) v = hi + v;
        if (v < lo || v >= hi) { PyErr_Format(PyExc_IndexError, "start index not in range %zd..%zd", lo, hi); return false; }
        *start = v;
    } else return true;
    if (PyTuple_GET_SIZE(args) > i1) {
        PyObject* o = PyNumber_Index(PyTuple_GET_ITEM(args, i1));
        if (!o) return false;
        Py_ssize_t v = PyNumber_AsSsize_t(o, PyExc_IndexError);
        Py_DECREF(o);
        if (v == -1 && PyErr_Occurred()) return false;
        if (v < 0) v = hi - 1 + v;
        if (v < lo || v > hi) { PyErr_Format(PyExc_IndexError, "end index not in range %zd..%zd", lo, hi); return false; }
        *end = v;
    }
    return true;
}

// flatten + upload when the trie version moved (device image is tagged like iterators are,
// src/AutomatonSearchIter.c:247-250)
bool gpu_sync(AutomatonObject* a) {
    const int64_t v = acx_trie_version(a->trie);
    if (a->image && a->image_version == v) return true;
    image_retire(a);
    void* blob = nullptr; size_t nbytes = 0;
    int rc = acx_flatten(a->trie, &blob, &nbytes);                // (reads the host trie: the GIL stays held)
    if (rc) { set_acx_error(rc); return false; }
    acx_image_t* img = nullptr;
    rc = acx_image_upload(blob, nbytes, &img);
    acx_blob_free(blob);
    if (rc) { set_acx_error(rc); return false; }
    a->image = new (std::nothrow) ImageRef{img, 0, false};
    if (!a->image) { acx_image_free(img); PyErr_NoMemory(); return false; }
    a->image_version = v;
    return true;
}

// one GPU scan of n haystacks given as (data, offsets).  The GIL is released around H2D, kernels and D2H; the
// records stay valid until lease->done() (or its destructor).  `data` must not be a buffer that another Python
// thread can free meanwhile: the callers pass memory of objects they hold a reference to, or their own copies.
// ctx != nullptr (ACX_SCAN_ALL, one haystack): the bytes a stream delivered before `data` — matches may begin in there,
// none that ends in there is reported (acx_scan_host_ctx).  ACX_SCAN_ALL asks for no final states either way.
bool run_scan(AutomatonObject* a, int mode, const uint8_t* data, const int64_t* off, int64_t n,
              const int32_t* init_state, const int32_t* index_base,
              const int64_t** moff, const acx_match_t** m, const int32_t** fin, ScanLease* lease,
              const uint8_t* ctx = nullptr, int64_t ctx_len = 0, int32_t flags = 0) {
    // What does not pay a launch — BASELINE config 1, a process without a device, a haystack below the crossover — is a walk
    // over the host trie (acx_trie_scan_host, include/acx.h §4b).  A carried iter_long state decides alone: a negative one is
    // a node of the host trie, a positive one a state of the device image.
    bool host;
    if (init_state && n == 1 && init_state[0] != 0) host = mode == ACX_SCAN_LONG && init_state[0] < 0;
    else host = !init_state && acx_host_walk_applies(off[n]) != 0;
    if (host) {
        lease->a = a;
        if (a->results && !a->results->empty()) { lease->res = a->results->back(); a->results->pop_back(); }
        const int64_t c_off[2] = {0, ctx_len};
        // (the GIL stays held — the walk reads the trie that add_word of another thread would grow —: microseconds for what the walk is
        //  for (at most 2 KiB with a device), 20 ns per byte for a process without a device or a stream that carries a host state)
        int rc = acx_trie_scan_host(a->trie, mode, data, off, n, (mode == ACX_SCAN_ALL && ctx && n == 1) ? ctx : nullptr,
                                    (mode == ACX_SCAN_ALL && ctx && n == 1) ? c_off : nullptr,
                                    mode == ACX_SCAN_LONG ? init_state : nullptr, index_base, flags, 1, &lease->res);
        if (!rc) rc = acx_result_fetch_host(lease->res, moff, m, fin);
        if (rc) { set_acx_error(rc); return false; }
        return true;
    }
    if (!gpu_sync(a)) return false;
    lease->a = a; lease->ref = a->image; lease->ref->users++;
    if (a->results && !a->results->empty()) { lease->res = a->results->back(); a->results->pop_back(); }
    acx_image_t* const img = lease->ref->img;
    int rc;
    char err[512];
    err[0] = 0;
    const int64_t ctx_off[2] = {0, ctx_len};
    static const uint8_t no_ctx = 0;
    Py_BEGIN_ALLOW_THREADS
    if (mode == ACX_SCAN_ALL && !init_state)
        rc = acx_scan_host_ctx(img, data, off, n, ctx ? ctx : (n == 1 ? &no_ctx : nullptr), n == 1 ? ctx_off : nullptr, index_base, flags, &lease->res);
    else rc = acx_scan_host(img, mode, data, off, n, init_state, index_base, &lease->res);
    if (!rc) rc = acx_result_fetch_host(lease->res, moff, m, fin);
    if (rc) { strncpy(err, acx_last_error(), sizeof err - 1); err[sizeof err - 1] = 0; }     // (the message is thread-local: keep it across the switch)
    Py_END_ALLOW_THREADS
    if (rc) {
        if (rc == ACX_E_NOMEM) PyErr_NoMemory(); else PyErr_SetString(PyExc_RuntimeError, err);
        return false;
    }
    return true;
}

PyObject* make_pair(AutomatonObject* a, int32_t index, int32_t value) {
    if (a->store == STORE_ANY) {
        PyObject* o = PyList_GetItem(a->values, value);           // borrowed; "O" increfs
        if (!o) return nullptr;
        return Py_BuildValue("iO", (int)index, o);                // src/AutomatonSearchIter.c:186-188
    }
    return Py_BuildValue("ii", (int)index, (int)value);           // src/AutomatonSearchIter.c:181-184
}

// ---- Automaton -----------------------------------------------------------------------------
bool check_store_key(int store, int key_type) {
    if (store != STORE_INTS && store != STORE_LENGTH && store != STORE_ANY) {
        PyErr_SetString(PyExc_ValueError, "store value must be one of ahocorasick.STORE_LENGTH, STORE_INTS or STORE_ANY");
        return false;
    }
    if (key_type != KEY_STRING && key_type != KEY_SEQUENCE) {
        PyErr_SetString(PyExc_ValueError, "key_type must have value KEY_STRING or KEY_SEQUENCE");
        return false;
    }
    return true;
}

AutomatonObject* automaton_alloc(PyTypeObject* type, int store, int key_type) {
    AutomatonObject* a = (AutomatonObject*)type->tp_alloc(type, 0);
    if (!a) return nullptr;
    a->trie = nullptr; a->values = nullptr; a->image = nullptr; a->image_version = -1;
    a->results = new (std::nothrow) std::vector<acx_result_t*>();
    a->free_slots = new (std::nothrow) std::vector<Py_ssize_t>();
    a->store = store; a->key_type = key_type;
    return a;
}

// the 7-tuple of __reduce__ (ours or the reference's bytes build): src/Automaton.c:107-149,
// automaton_unpickle src/Automaton_pickle.c:326-488; parsing in libacx (acx_persist.cpp)
PyObject* automaton_from_pickle(PyTypeObject* type, PyObject* args) {
    PyObject *bytes_list = nullptr, *values = nullptr;
    int kind, store, key_type, count, longest;
    if (!PyArg_ParseTuple(args, "OiiiiiO", &bytes_list, &kind, &store, &key_type, &count, &longest, &values)) {
        PyErr_SetString(PyExc_ValueError, "Unable to load from pickle.");
        return nullptr;
    }
    if (!check_store_key(store, key_type)) return nullptr;
    if (kind != K_EMPTY && kind != K_TRIE && kind != K_AHOCORASICK) {
        PyErr_SetString(PyExc_ValueError, "kind value must be one of ahocorasick.EMPTY, TRIE or AHOCORASICK");
        return nullptr;
    }
    if (!PyList_CheckExact(bytes_list)) { PyErr_SetString(PyExc_TypeError, "Expected list"); return nullptr; }
    AutomatonObject* a = automaton_alloc(type, store, key_type);
    if (!a) return nullptr;
    int rc;
    if (kind == K_EMPTY) {
        rc = acx_trie_new(&a->trie);
        if (rc) { Py_DECREF(a); return set_acx_error(rc); }
        if (store == STORE_ANY && !(a->values = PyList_New(0))) { Py_DECREF(a); return nullptr; }
        return (PyObject*)a;
    }
    const Py_ssize_t n = PyList_GET_SIZE(bytes_list);
    std::vector<const void*> ptrs((size_t)n);
    std::vector<size_t> sizes((size_t)n);
    for (Py_ssize_t k = 0; k < n; k++) {
        PyObject* b = PyList_GET_ITEM(bytes_list, k);
        if (!PyBytes_CheckExact(b)) {
            PyErr_Format(PyExc_ValueError, "Item #%zd on the bytes list is not a bytes object", k);
            Py_DECREF(a);
            return nullptr;
        }
        ptrs[(size_t)k] = PyBytes_AS_STRING(b);
        sizes[(size_t)k] = (size_t)PyBytes_GET_SIZE(b);
    }
    int64_t n_eow = 0;
    rc = acx_trie_from_ref_pickle(ptrs.data(), sizes.data(), (size_t)n, store == STORE_ANY, longest, ACX_LETTER_BYTES,
                                  key_type == KEY_SEQUENCE, &a->trie, &n_eow);
    if (rc) {
        Py_DECREF(a);
        if (rc == ACX_E_NOMEM) return PyErr_NoMemory();
        PyErr_SetString(PyExc_ValueError, acx_last_error());
        return nullptr;
    }
    if (store == STORE_ANY) {
        if (!PyList_Check(values) || PyList_GET_SIZE(values) < (Py_ssize_t)n_eow) {
            PyErr_SetString(PyExc_IndexError, "list index out of range");
            Py_DECREF(a);
            return nullptr;
        }
        if (!(a->values = PyList_GetSlice(values, 0, (Py_ssize_t)n_eow))) { Py_DECREF(a); return nullptr; }
    }
    if (kind == K_AHOCORASICK) {
        rc = acx_trie_make_automaton(a->trie, nullptr);
        if (rc) { Py_DECREF(a); return set_acx_error(rc); }
    }
    return (PyObject*)a;
}

PyObject* automaton_new(PyTypeObject* type, PyObject* args, PyObject*) {
    // the 7-tuple of __reduce__: a pickle written by the reference build of this flavour, or by this module
    if (PyTuple_GET_SIZE(args) == 7) return automaton_from_pickle(type, args);
    int store = STORE_ANY, key_type = KEY_STRING;
    if (!PyArg_ParseTuple(args, "|ii", &store, &key_type)) return nullptr;
    if (!check_store_key(store, key_type)) return nullptr;
    AutomatonObject* a = automaton_alloc(type, store, key_type);
    if (!a) return nullptr;
    int rc = acx_trie_new(&a->trie);
    if (rc) { Py_DECREF(a); return set_acx_error(rc); }
    if (store == STORE_ANY) { a->values = PyList_New(0); if (!a->values) { Py_DECREF(a); return nullptr; } }
    return (PyObject*)a;
}

void automaton_dealloc(AutomatonObject* a) {
    image_retire(a);
    if (a->results) { for (acx_result_t* r : *a->results) acx_result_free(r); delete a->results; }
    if (a->trie) acx_trie_free(a->trie);
    Py_XDECREF(a->values);
    delete a->free_slots;
    Py_TYPE(a)->tp_free((PyObject*)a);
}

Py_ssize_t automaton_len(AutomatonObject* a) { return (Py_ssize_t)acx_trie_num_keys(a->trie); }

PyObject* automaton_add_word(AutomatonObject* a, PyObject* args) {
    const Py_ssize_t na = PyTuple_GET_SIZE(args);
    if (na < 1) { PyErr_SetString(PyExc_TypeError, "add_word() takes a key"); return nullptr; }
    Text kt;
    if (!get_text(PyTuple_GET_ITEM(args, 0), &kt, false, a->key_type)) return nullptr;
    const uint8_t* key = kt.data; const Py_ssize_t len = kt.nbytes;
    int64_t v = 0;
    PyObject* obj = nullptr;
    Py_ssize_t slot = -1;
    if (a->store == STORE_ANY) {                                   // src/Automaton.c:216-223
        if (na < 2) { PyErr_SetString(PyExc_ValueError, "A value object is required as second argument."); return nullptr; }
        obj = PyTuple_GET_ITEM(args, 1);
        if (len == 0) Py_RETURN_FALSE;
        int found = 0; int64_t old = 0;
        int rc = acx_trie_get(a->trie, key, (size_t)len, &found, &old);
        if (rc) return set_acx_error(rc);
        if (found) slot = (Py_ssize_t)old;
        else if (a->free_slots && !a->free_slots->empty()) { slot = a->free_slots->back(); a->free_slots->pop_back(); }
        else {
            slot = PyList_GET_SIZE(a->values);
            if (PyList_Append(a->values, Py_None) < 0) return nullptr;
        }
        v = slot;
    } else if (a->store == STORE_INTS) {                           // src/Automaton.c:225-243
        if (na >= 2) {
            PyObject* o = PyTuple_GET_ITEM(args, 1);
            if (!PyNumber_Check(o)) { PyErr_SetString(PyExc_TypeError, "An integer value is required as second argument."); return nullptr; }
            Py_ssize_t iv = PyNumber_AsSsize_t(o, PyExc_ValueError);
            if (iv == -1 && PyErr_Occurred()) return nullptr;
            v = (int64_t)iv;
        } else v = acx_trie_num_keys(a->trie) + 1;
    } else v = (int64_t)kt.nchars;                                 // STORE_LENGTH: letters
    int is_new = 0;
    int rc = acx_trie_add_word(a->trie, key, (size_t)len, v, &is_new);
    if (rc) return set_acx_error(rc);
    if (obj && len > 0) {
        Py_INCREF(obj);
        if (PyList_SetItem(a->values, slot, obj) < 0) return nullptr;   // steals; drops the old value
    }
    if (is_new) Py_RETURN_TRUE;
    Py_RETURN_FALSE;
}

// add_words(keys[, values]) -> number of new keys.  Not in the reference (an extension, like iter_batch): add_word for
// every (key, value) pair in order, one Python call for the lot — a million signatures cost seconds of interpreter
// time otherwise.
PyObject* automaton_add_words(AutomatonObject* a, PyObject* args) {
    PyObject* keys; PyObject* values = nullptr;
    if (!PyArg_ParseTuple(args, "O|O", &keys, &values)) return nullptr;
    if (values == Py_None) values = nullptr;
    PyObject* kf = PySequence_Fast(keys, "add_words() takes a sequence of keys");
    if (!kf) return nullptr;
    PyObject* vf = values ? PySequence_Fast(values, "add_words(): values must be a sequence") : nullptr;
    if (values && !vf) { Py_DECREF(kf); return nullptr; }
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(kf);
    if (vf && PySequence_Fast_GET_SIZE(vf) != n) {
        Py_DECREF(kf); Py_DECREF(vf);
        PyErr_SetString(PyExc_ValueError, "add_words(): as many values as keys");
        return nullptr;
    }
    Py_ssize_t fresh = 0;
    for (Py_ssize_t i = 0; i < n; i++) {
        PyObject* pair = vf ? PyTuple_Pack(2, PySequence_Fast_GET_ITEM(kf, i), PySequence_Fast_GET_ITEM(vf, i))
                            : PyTuple_Pack(1, PySequence_Fast_GET_ITEM(kf, i));
        PyObject* r = pair ? automaton_add_word(a, pair) : nullptr;
        Py_XDECREF(pair);
        if (!r) { Py_DECREF(kf); Py_XDECREF(vf); return nullptr; }
        fresh += r == Py_True;
        Py_DECREF(r);
    }
    Py_DECREF(kf); Py_XDECREF(vf);
    return PyLong_FromSsize_t(fresh);
}

bool lookup(AutomatonObject* a, PyObject* keyobj, int* found, int64_t* value) {
    Text kt;
    if (!get_text(keyobj, &kt, false, a->key_type)) return false;
    int rc = acx_trie_get(a->trie, kt.data, (size_t)kt.nbytes, found, value);
    if (rc) { set_acx_error(rc); return false; }
    return true;
}

PyObject* automaton_exists(AutomatonObject* a, PyObject* args) {
    PyObject* k; if (!PyArg_ParseTuple(args, "O", &k)) return nullptr;
    int found; int64_t v;
    if (!lookup(a, k, &found, &v)) return nullptr;
    if (found) Py_RETURN_TRUE;
    Py_RETURN_FALSE;
}

int automaton_contains(AutomatonObject* a, PyObject* k) {
    int found; int64_t v;
    if (!lookup(a, k, &found, &v)) return -1;
    return found;
}

PyObject* value_object(AutomatonObject* a, int64_t v) {
    if (a->store == STORE_ANY) { PyObject* o = PyList_GetItem(a->values, (Py_ssize_t)v); Py_XINCREF(o); return o; }
    return PyLong_FromLongLong((long long)v);
}

PyObject* automaton_get(AutomatonObject* a, PyObject* args) {
    PyObject* k; PyObject* dflt = nullptr;
    if (!PyArg_ParseTuple(args, "O|O", &k, &dflt)) return nullptr;
    int found; int64_t v;
    if (!lookup(a, k, &found, &v)) return nullptr;
    if (found) return value_object(a, v);
    if (dflt) { Py_INCREF(dflt); return dflt; }
    PyErr_SetObject(PyExc_KeyError, k);
    return nullptr;
}

PyObject* automaton_longest_prefix(AutomatonObject* a, PyObject* args) {
    PyObject* k; if (!PyArg_ParseTuple(args, "O", &k)) return nullptr;
    Text kt;
    if (!get_text(k, &kt, false, a->key_type)) return nullptr;
    const uint8_t* key = kt.data; const Py_ssize_t len = kt.nbytes;
    size_t n = 0;
    int rc = acx_trie_longest_prefix(a->trie, key, (size_t)len, &n);
    if (rc) return set_acx_error(rc);
    // one byte is one letter for bytes keys (KEY_STRING of the bytes build): bytes 0x80..0xBF are letters
    // like any other.  Multi-byte letters (str build, KEY_SEQUENCE): a prefix that ends inside a letter
    // still counts only whole letters.
    if (!ACX_UNICODE_BUILD && a->key_type != KEY_SEQUENCE) return PyLong_FromSize_t(n);
    while (n > 0 && n < (size_t)len && !is_char_start(key[n])) n--;
    return PyLong_FromSize_t((size_t)chars_in(key, (Py_ssize_t)n));
}

// 1 removed (value in *out, new ref), 0 absent, -1 error
int remove_common(AutomatonObject* a, PyObject* args, PyObject** out) {
    PyObject* k; if (!PyArg_ParseTuple(args, "O", &k)) return -1;
    Text kt;
    if (!get_text(k, &kt, false, a->key_type)) return -1;
    int found = 0; int64_t v = 0;
    int rc = acx_trie_remove_word(a->trie, kt.data, (size_t)kt.nbytes, &found, &v);
    if (rc) { set_acx_error(rc); return -1; }
    if (!found) return 0;
    if (a->store == STORE_ANY) {
        PyObject* o = PyList_GetItem(a->values, (Py_ssize_t)v);
        if (!o) return -1;
        Py_INCREF(o);
        Py_INCREF(Py_None);
        PyList_SetItem(a->values, (Py_ssize_t)v, Py_None);
        if (a->free_slots) a->free_slots->push_back((Py_ssize_t)v);
        *out = o;
    } else *out = PyLong_FromLongLong((long long)v);
    return 1;
}

PyObject* automaton_remove_word(AutomatonObject* a, PyObject* args) {
    PyObject* v = nullptr;
    int r = remove_common(a, args, &v);
    if (r < 0) return nullptr;
    Py_XDECREF(v);
    if (r) Py_RETURN_TRUE;
    Py_RETURN_FALSE;
}

PyObject* automaton_pop(AutomatonObject* a, PyObject* args) {
    PyObject* v = nullptr;
    int r = remove_common(a, args, &v);
    if (r < 0) return nullptr;
    if (!r) { PyErr_SetNone(PyExc_KeyError); return nullptr; }     // src/Automaton.c:354-356
    return v;
}

PyObject* automaton_clear(AutomatonObject* a, PyObject*) {
    acx_trie_clear(a->trie);
    if (a->values) { if (PyList_SetSlice(a->values, 0, PyList_GET_SIZE(a->values), nullptr) < 0) return nullptr; }
    if (a->free_slots) a->free_slots->clear();
    image_retire(a);
    Py_RETURN_NONE;
}

PyObject* automaton_make_automaton(AutomatonObject* a, PyObject*) {
    int changed = 0;
    const int rc = acx_trie_make_automaton(a->trie, &changed);       // (GIL held: the trie is being rewritten)
    if (rc) return set_acx_error(rc);
    if (changed) Py_RETURN_NONE;
    Py_RETURN_FALSE;                                               // src/Automaton.c:574-575
}

PyObject* automaton_get_kind(AutomatonObject* a, void*) { return PyLong_FromLong(acx_trie_kind(a->trie)); }
PyObject* automaton_get_store(AutomatonObject* a, void*) { return PyLong_FromLong(a->store); }

// ---- search iterators ----------------------------------------------------------------------
struct SearchIterObject {
    PyObject_HEAD
    AutomatonObject* automaton;
    int64_t version;
    std::vector<acx_match_t>* pending;
    size_t pos;
    int32_t state;          // iter_long: carried across set()
    std::vector<uint8_t>* ctx;   // iter: what the stream delivered before the current chunk (its last longest_word - 1 letters)
    Py_ssize_t shift;
    Py_ssize_t ref_index;   // the reference's iter->index, for set()'s shift arithmetic
    Py_ssize_t end;
    bool ignore_ws;
    bool is_long;
    // what set() needs when it is called before the iterator is exhausted: the state the reference holds
    // then is the one after the last yielded position, found by scanning that prefix of this chunk again
    PyObject* src;          // the object being scanned (strong reference)
    Py_ssize_t start;       // first letter of the scanned slice
    int32_t state0;         // state before this chunk
    bool exhausted;         // StopIteration seen: the reference has walked the whole chunk
    bool loaded;            // the chunk has been scanned (lazily: at the first next(), as the reference walks nothing in iter())
    bool busy;              // a scan of this iterator is running without the GIL: next() / set() from another thread raise
};

// The scan behind next() runs without the GIL and holds pointers into the iterator's source and context; a second
// thread that calls next() or set() on the SAME iterator meanwhile would free or move them.  The reference holds the
// GIL throughout (sharing an iterator is memory-safe there); here re-entry raises, as it does for a generator.
struct IterBusy {
    SearchIterObject* it; bool ok;
    explicit IterBusy(SearchIterObject* i) : it(i), ok(!i->busy) { if (ok) it->busy = true; else PyErr_SetString(PyExc_ValueError, "iterator already executing"); }
    ~IterBusy() { if (ok) it->busy = false; }
};

extern PyTypeObject SearchIterType;

inline bool is_cspace(uint8_t b) { return b == ' ' || (b >= '\t' && b <= '\r'); }    // iswspace over bytes-build letters

// Scan letters [start, end) of `t`; matches come back with end_index in LETTERS of the whole
// string plus `index_shift`.  ignore_ws: white-space letters are skipped without touching the state
// (src/AutomatonSearchIter.c:269-274).  state_io: automaton state carried in and out (iter.set()).
// ctx (ACX_SCAN_ALL): the letters the stream delivered before this slice (iter().set()): the scan continues from them,
// not from a state.
bool scan_text(AutomatonObject* a, int mode, const Text& t, Py_ssize_t start, Py_ssize_t end, bool ignore_ws,
               int32_t* state_io, Py_ssize_t index_shift, std::vector<acx_match_t>* out,
               const std::vector<uint8_t>* ctx = nullptr) {
    const Py_ssize_t bs = char_to_byte(t, start), be = char_to_byte(t, end);
    const uint8_t* src = t.data + bs;
    const Py_ssize_t nb = be - bs;
    int64_t off[2] = {0, (int64_t)nb};
    // ignore_ws: the device takes the white space out and maps the end indices back (ACX_SCAN_SKIP_WS)
    std::vector<int32_t> cob;                                     // letter (relative to `start`) of each byte of the slice
    if (!t.ascii()) {
        cob.resize((size_t)nb);
        int32_t c = -1;
        for (Py_ssize_t i = 0; i < nb; i++) { c += is_char_start(src[i]); cob[(size_t)i] = c; }
    }
    const int64_t* moff; const acx_match_t* m; const int32_t* fin;
    int32_t init = state_io ? *state_io : 0;
    // (the root needs no init_state array: such a scan may take the position-parallel kernels)
    ScanLease lease;
    if (!run_scan(a, mode, src, off, 1, (state_io && init != 0) ? &init : nullptr, nullptr, &moff, &m, &fin, &lease,
                  ctx && !ctx->empty() ? ctx->data() : nullptr, ctx ? (int64_t)ctx->size() : 0,
                  (ignore_ws && mode == ACX_SCAN_ALL) ? ACX_SCAN_SKIP_WS : 0)) return false;
    out->assign(m, m + moff[1]);
    for (auto& r : *out) {
        const int32_t byte_off = r.end_index;
        const int32_t letter = t.ascii() ? byte_off : cob[(size_t)byte_off];
        r.end_index = (int32_t)(start + letter + index_shift);
    }
    if (state_io && fin) *state_io = fin[0];
    lease.done();
    return true;
}

// the context after the letters [start, upto) of `t` have been walked behind `ctx`: their bytes (white space left out
// when the iterator ignores it: it does not touch the state, src/AutomatonSearchIter.c:269-274), the last
// longest_word - 1 of the lot
void ctx_advance(AutomatonObject* a, std::vector<uint8_t>* ctx, const Text& t, Py_ssize_t start, Py_ssize_t upto, bool ignore_ws) {
    const Py_ssize_t bs = char_to_byte(t, start), be = char_to_byte(t, upto);
    for (Py_ssize_t i = bs; i < be; i++) if (!ignore_ws || !is_cspace(t.data[i])) ctx->push_back(t.data[i]);
    const int64_t lw = acx_trie_longest_word(a->trie);
    const size_t keep = lw > 1 ? (size_t)(lw - 1) : 0;
    if (ctx->size() > keep) ctx->erase(ctx->begin(), ctx->end() - (std::ptrdiff_t)keep);
}

// A chunk is scanned when its first match is asked for, not when iter() / set() hands it over: the reference walks
// nothing there either (src/AutomatonSearchIter.c:65-131), and code that creates iterators it never drains
// (tests/test_issue_9.py: two million of them) must not pay a GPU scan each.
bool iter_load(SearchIterObject* it, const Text& t, Py_ssize_t start, Py_ssize_t end) {
    (void)t;
    it->pending->clear();
    it->start = start; it->state0 = it->state;
    it->pos = 0; it->exhausted = false; it->loaded = false;
    it->end = end;
    it->ref_index = start - 1;                                     // src/AutomatonSearchIter.c:123
    return true;
}

bool iter_ensure_loaded(SearchIterObject* it) {
    if (it->loaded) return true;
    Text t;
    if (!get_text(it->src, &t, true, it->automaton->key_type)) return false;
    Py_ssize_t end = it->end > t.nchars ? t.nchars : it->end;
    if (!scan_text(it->automaton, it->is_long ? ACX_SCAN_LONG : ACX_SCAN_ALL, t, it->start, end, it->ignore_ws,
                   it->is_long ? &it->state : nullptr, it->shift, it->pending, it->is_long ? nullptr : it->ctx)) return false;
    it->pos = 0; it->loaded = true;
    return true;
}

PyObject* search_iter_create(AutomatonObject* a, PyObject* srcobj, const Text& t, Py_ssize_t start, Py_ssize_t end, bool ws, bool is_long) {
    SearchIterObject* it = PyObject_New(SearchIterObject, &SearchIterType);
    if (!it) return nullptr;
    it->automaton = a; Py_INCREF(a);
    it->version = acx_trie_version(a->trie);
    it->pending = new std::vector<acx_match_t>();
    it->ctx = new std::vector<uint8_t>();
    it->pos = 0; it->state = 0; it->shift = 0; it->ref_index = -1; it->end = 0; it->ignore_ws = ws; it->is_long = is_long;
    Py_INCREF(srcobj); it->src = srcobj; it->start = 0; it->state0 = 0; it->exhausted = false; it->loaded = false; it->busy = false;
    if (!iter_load(it, t, start, end)) { Py_DECREF(it); return nullptr; }
    return (PyObject*)it;
}

void search_iter_dealloc(SearchIterObject* it) {
    Py_XDECREF(it->automaton);
    Py_XDECREF(it->src);
    delete it->pending;
    delete it->ctx;
    PyObject_Del(it);
}

PyObject* search_iter_iter(PyObject* self) { Py_INCREF(self); return self; }

PyObject* search_iter_next(SearchIterObject* it) {
    if (it->version != acx_trie_version(it->automaton->trie)) {    // src/AutomatonSearchIter.c:247-250
        PyErr_SetString(PyExc_ValueError, "underlaying automaton has changed, iterator is not valid anymore");
        return nullptr;
    }
    {
        IterBusy guard(it);
        if (!guard.ok || !iter_ensure_loaded(it)) return nullptr;
    }
    if (it->pos >= it->pending->size()) { it->ref_index = it->end; it->exhausted = true; return nullptr; }   // StopIteration
    const acx_match_t r = (*it->pending)[it->pos++];
    it->ref_index = (Py_ssize_t)r.end_index - it->shift;
    return make_pair(it->automaton, r.end_index, r.value);
}

PyObject* search_iter_set(SearchIterObject* it, PyObject* args) {  // src/AutomatonSearchIter.c:303-368
    PyObject* s; int reset = 0;
    if (!PyArg_ParseTuple(args, "O|p", &s, &reset)) return nullptr;
    Text t;
    if (!get_text(s, &t, true, it->automaton->key_type)) return nullptr;
    IterBusy guard(it);
    if (!guard.ok) return nullptr;
    if (reset) { it->state = 0; it->shift = 0; it->ctx->clear(); }
    else {
        // What the reference has walked of the old chunk: all of it after StopIteration, else up to the last match it
        // returned (nothing, when no match was asked for).  iter: those letters go behind the old context; no scan.
        // iter_long: it is at the root after every match it returned (src/AutomatonSearchIterLong.c:101-110), where it
        // was before the chunk when it returned none, and where the walk ended after StopIteration.
        if (it->version == acx_trie_version(it->automaton->trie)) {
            if (it->is_long) {
                if (!it->loaded) it->state = it->state0;
                else if (!it->exhausted) it->state = it->ref_index >= it->start ? 0 : it->state0;
            } else if (it->src) {
                Text old;
                if (!get_text(it->src, &old, true, it->automaton->key_type)) return nullptr;
                Py_ssize_t upto = it->exhausted ? it->end : (it->ref_index + 1 > it->start ? it->ref_index + 1 : it->start);
                if (upto > old.nchars) upto = old.nchars;
                ctx_advance(it->automaton, it->ctx, old, it->start, upto, it->ignore_ws);
            }
        }
        it->shift += it->ref_index >= 0 ? it->ref_index : 0;
    }
    Py_INCREF(s); Py_XSETREF(it->src, s);
    if (!iter_load(it, t, 0, t.nchars)) return nullptr;
    Py_RETURN_NONE;
}

PyMethodDef search_iter_methods[] = {
    {"set", (PyCFunction)search_iter_set, METH_VARARGS, "set(string, reset=False): continue the search on a new chunk"},
    {nullptr, nullptr, 0, nullptr}};

PyTypeObject SearchIterType = {PyVarObject_HEAD_INIT(nullptr, 0) "ahocorasick.AutomatonSearchIter"};

// ---- search methods ------------------------------------------------------------------------
PyObject* automaton_iter(AutomatonObject* a, PyObject* args, PyObject* kw) {
    static const char* kwlist[] = {"string", "start", "end", "ignore_white_space", nullptr};
    if (acx_trie_kind(a->trie) != K_AHOCORASICK) { PyErr_SetString(PyExc_AttributeError, NOT_AUTOMATON_MSG); return nullptr; }
    PyObject* s; int start = -1, end = -1, ws = -1;
    if (!PyArg_ParseTupleAndKeywords(args, kw, "O|iii", (char**)kwlist, &s, &start, &end, &ws)) return nullptr;
    Text t;
    if (!get_text(s, &t, true, a->key_type)) return nullptr;
    const Py_ssize_t n = t.nchars;
    // -1 = default for both (src/Automaton.c:893-956).  The reference does not validate the
    // range (out of range is undefined behaviour there); here it is clamped to the haystack.
    Py_ssize_t st = start == -1 ? 0 : start, en = end == -1 ? n : end;
    if (st < 0) st = 0;
    if (st > n) st = n;
    if (en > n) en = n;
    if (en < st) en = st;
    return search_iter_create(a, s, t, st, en, ws == 1, false);
}

PyObject* automaton_iter_long(AutomatonObject* a, PyObject* args) {
    if (acx_trie_kind(a->trie) != K_AHOCORASICK) {
        PyErr_SetString(PyExc_AttributeError, "not an automaton yet; add some words and call make_automaton");
        return nullptr;
    }
    if (PyTuple_GET_SIZE(args) < 1) { PyErr_SetString(PyExc_TypeError, "iter_long() takes a string"); return nullptr; }
    Text t;
    if (!get_text(PyTuple_GET_ITEM(args, 0), &t, true, a->key_type)) return nullptr;
    Py_ssize_t st, en;
    if (!parse_start_end(args, 1, 2, 0, t.nchars, &st, &en)) return nullptr;
    if (en < st) en = st;
    return search_iter_create(a, PyTuple_GET_ITEM(args, 0), t, st, en, false, true);
}

PyObject* automaton_find_all(AutomatonObject* a, PyObject* args) {
    if (acx_trie_kind(a->trie) != K_AHOCORASICK) Py_RETURN_NONE;   // src/Automaton.c:666-667
    if (PyTuple_GET_SIZE(args) < 2) { PyErr_SetString(PyExc_TypeError, "find_all() takes a string and a callback"); return nullptr; }
    PyObject* s = PyTuple_GET_ITEM(args, 0);
    PyObject* cb = PyTuple_GET_ITEM(args, 1);
    Text t;
    if (!get_text(s, &t, false, a->key_type)) return nullptr;
    if (!PyCallable_Check(cb)) {
        PyErr_SetString(PyExc_TypeError, "The callback argument must be a callable such as a function.");
        return nullptr;
    }
    Py_ssize_t st, en;
    if (!parse_start_end(args, 2, 3, 0, t.nchars, &st, &en)) return nullptr;
    if (en < st) en = st;
    std::vector<acx_match_t> copy;                                 // the callback may re-enter this automaton
    if (!scan_text(a, ACX_SCAN_ALL, t, st, en, false, nullptr, 0, &copy)) return nullptr;
    for (const acx_match_t& r : copy) {
        PyObject* pair = make_pair(a, r.end_index, r.value);
        if (!pair) return nullptr;
        PyObject* ret = PyObject_CallObject(cb, pair);
        Py_DECREF(pair);
        if (!ret) return nullptr;                                  // an exception aborts, src/Automaton.c:705-708
        Py_DECREF(ret);
    }
    Py_RETURN_NONE;
}

// NEW: iter_batch(list_of_bytes, long=False) -> [list(A.iter(h)) for h in list]
PyObject* automaton_iter_batch(AutomatonObject* a, PyObject* args, PyObject* kw) {
    static const char* kwlist[] = {"haystacks", "long", nullptr};
    if (acx_trie_kind(a->trie) != K_AHOCORASICK) { PyErr_SetString(PyExc_AttributeError, NOT_AUTOMATON_MSG); return nullptr; }
    PyObject* seq; int is_long = 0;
    if (!PyArg_ParseTupleAndKeywords(args, kw, "O|p", (char**)kwlist, &seq, &is_long)) return nullptr;
    PyObject* fast = PySequence_Fast(seq, "iter_batch() takes a sequence of bytes");
    if (!fast) return nullptr;
    const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
    std::vector<int64_t> off((size_t)n + 1, 0);
    std::vector<Text> texts((size_t)n);
    bool all_ascii = true;
    for (Py_ssize_t i = 0; i < n; i++) {
        if (!get_text(PySequence_Fast_GET_ITEM(fast, i), &texts[(size_t)i], true, a->key_type)) { Py_DECREF(fast); return nullptr; }
        off[(size_t)i + 1] = off[(size_t)i] + texts[(size_t)i].nbytes;
        all_ascii = all_ascii && texts[(size_t)i].ascii();
    }
    std::vector<uint8_t> data((size_t)off[(size_t)n]);
    for (Py_ssize_t i = 0; i < n; i++) memcpy(data.data() + off[(size_t)i], texts[(size_t)i].data, (size_t)texts[(size_t)i].nbytes);
    Py_DECREF(fast);
    const int64_t* moff; const acx_match_t* m0; const int32_t* fin;
    ScanLease lease;                                              // (its destructor hands the buffers back on every return path)
    if (!run_scan(a, is_long ? ACX_SCAN_LONG : ACX_SCAN_ALL, data.data(), off.data(), n, nullptr, nullptr, &moff, &m0, &fin, &lease))
        return nullptr;
    std::vector<acx_match_t> conv;                                // unicode build: byte indices -> letters
    const acx_match_t* m = m0;
    if (!all_ascii) {
        conv.assign(m0, m0 + moff[n]);
        for (Py_ssize_t i = 0; i < n; i++) {
            if (texts[(size_t)i].ascii()) continue;
            const uint8_t* p = data.data() + off[(size_t)i];
            int64_t k = moff[i];
            int32_t c = -1;
            for (Py_ssize_t bi = 0; bi < texts[(size_t)i].nbytes && k < moff[i + 1]; bi++) {   // matches are sorted by end index
                c += is_char_start(p[bi]);
                while (k < moff[i + 1] && conv[(size_t)k].end_index == (int32_t)bi) conv[(size_t)k++].end_index = c;
            }
        }
        m = conv.data();
    }
    PyObject* out = PyList_New(n);
    if (!out) return nullptr;
    for (Py_ssize_t i = 0; i < n; i++) {
        const int64_t lo = moff[i], hi = moff[i + 1];
        PyObject* lst = PyList_New((Py_ssize_t)(hi - lo));
        if (!lst) { Py_DECREF(out); return nullptr; }
        for (int64_t k = lo; k < hi; k++) {
            PyObject* pair = make_pair(a, m[k].end_index, m[k].value);
            if (!pair) { Py_DECREF(lst); Py_DECREF(out); return nullptr; }
            PyList_SET_ITEM(lst, (Py_ssize_t)(k - lo), pair);
        }
        PyList_SET_ITEM(out, i, lst);
    }
    return out;
}

// ---- persistence in the reference's formats (parsing/writing in acx_persist.cpp) ----
// objects of the keys in dump (pre-order) order: the `values` list of a STORE_ANY pickle
PyObject* eow_objects(AutomatonObject* a) {
    int64_t* ids = nullptr; int64_t n = 0;
    int rc = acx_trie_eow_values(a->trie, letters_multibyte(a->key_type), &ids, &n);
    if (rc) return set_acx_error(rc);
    PyObject* list = PyList_New((Py_ssize_t)n);
    for (int64_t k = 0; list && k < n; k++) {
        PyObject* o = PyList_GetItem(a->values, (Py_ssize_t)ids[k]);  // borrowed
        if (!o) { Py_CLEAR(list); break; }
        Py_INCREF(o);
        PyList_SET_ITEM(list, (Py_ssize_t)k, o);
    }
    acx_blob_free(ids);
    return list;
}

// same tuple as the reference (src/Automaton_pickle.c:192-285): loadable by either side
PyObject* automaton_reduce(AutomatonObject* a, PyObject*) {
    if (acx_trie_num_keys(a->trie) == 0) return Py_BuildValue("O()", Py_TYPE(a));
    void* buf = nullptr; size_t* sizes = nullptr; size_t n = 0;
    int rc = acx_trie_to_ref_pickle(a->trie, a->store == STORE_ANY, 0, ACX_LETTER_BYTES, letters_multibyte(a->key_type), &buf, &sizes, &n);
    if (rc) return set_acx_error(rc);
    PyObject* chunks = PyList_New((Py_ssize_t)n);
    size_t at = 0;
    for (size_t k = 0; chunks && k < n; k++) {
        PyObject* b = PyBytes_FromStringAndSize((const char*)buf + at, (Py_ssize_t)sizes[k]);
        if (!b) { Py_CLEAR(chunks); break; }
        PyList_SET_ITEM(chunks, (Py_ssize_t)k, b);
        at += sizes[k];
    }
    acx_blob_free(buf); acx_blob_free(sizes);
    if (!chunks) return nullptr;
    PyObject* values;
    if (a->store == STORE_ANY) { values = eow_objects(a); if (!values) { Py_DECREF(chunks); return nullptr; } }
    else { values = Py_None; Py_INCREF(values); }
    return Py_BuildValue("O(NiiiiiN)", Py_TYPE(a), chunks, acx_trie_kind(a->trie), a->store, a->key_type,
                         (int)acx_trie_num_keys(a->trie), (int)acx_trie_longest_word(a->trie), values);
}

// argument rules of save()/load(): src/custompickle/pyhelpers.c:4-59
bool parse_path_callback(int store, PyObject* args, const char** path, PyObject** callback) {
    if (store == STORE_ANY) {
        if (PyTuple_GET_SIZE(args) != 2) { PyErr_SetString(PyExc_ValueError, "expected exactly two arguments"); return false; }
    } else if (PyTuple_GET_SIZE(args) != 1) { PyErr_SetString(PyExc_ValueError, "expected exactly one argument"); return false; }
    PyObject* s = PyTuple_GET_ITEM(args, 0);
    if (!PyUnicode_Check(s)) { PyErr_SetString(PyExc_TypeError, "the first argument must be a string"); return false; }
    *callback = nullptr;
    if (store == STORE_ANY) {
        *callback = PyTuple_GET_ITEM(args, 1);
        if (!PyCallable_Check(*callback)) { PyErr_SetString(PyExc_TypeError, "the second argument must be a callable object"); return false; }
    }
    *path = PyUnicode_AsUTF8(s);
    return *path != nullptr;
}

// Automaton.save(path[, serializer]): src/custompickle/save/automaton_save.c:13-138
PyObject* automaton_save(AutomatonObject* a, PyObject* args) {
    const char* path; PyObject* serializer;
    if (!parse_path_callback(a->store, args, &path, &serializer)) return nullptr;
    std::vector<PyObject*> keep;                 // serialized values, kept alive until written
    std::vector<const void*> ptrs;
    std::vector<size_t> sizes;
    auto release = [&]() { for (PyObject* o : keep) Py_DECREF(o); };
    if (a->store == STORE_ANY) {
        PyObject* objs = eow_objects(a);
        if (!objs) return nullptr;
        for (Py_ssize_t k = 0; k < PyList_GET_SIZE(objs); k++) {
            PyObject* b = PyObject_CallFunctionObjArgs(serializer, PyList_GET_ITEM(objs, k), nullptr);
            if (b && !PyBytes_CheckExact(b)) { Py_DECREF(b); b = nullptr; PyErr_SetString(PyExc_TypeError, "serializer must return bytes object"); }
            if (!b) { Py_DECREF(objs); release(); return nullptr; }
            keep.push_back(b); ptrs.push_back(PyBytes_AS_STRING(b)); sizes.push_back((size_t)PyBytes_GET_SIZE(b));
        }
        Py_DECREF(objs);
    }
    void* buf = nullptr; size_t nbytes = 0;
    int rc = acx_trie_to_ref_savefile(a->trie, a->store, a->key_type, ACX_LETTER_BYTES, letters_multibyte(a->key_type),
                                      ptrs.data(), sizes.data(), &buf, &nbytes);
    release();
    if (rc) return set_acx_error(rc);
    FILE* f = fopen(path, "wb");
    const bool ok = f && fwrite(buf, 1, nbytes, f) == nbytes;
    if (f) fclose(f);
    acx_blob_free(buf);
    if (!ok) return PyErr_SetFromErrnoWithFilename(PyExc_IOError, path);
    Py_RETURN_NONE;
}

PyObject* automaton_get_stats(AutomatonObject* a, PyObject*) {       // src/Automaton.c:1077-1096
    int64_t v[6] = {0, 0, 0, 0, 0, 0};
    int rc = acx_trie_stats(a->trie, &v[0], &v[1], &v[2], &v[3], &v[4], &v[5]);
    if (rc) return set_acx_error(rc);
    return Py_BuildValue("{s:L,s:L,s:L,s:L,s:i,s:L}", "nodes_count", (long long)v[0], "words_count", (long long)v[1],
                         "longest_word", (long long)v[2], "links_count", (long long)v[3], "sizeof_node", (int)v[4],
                         "total_size", (long long)v[5]);
}

// match(key): True iff key is a prefix of some key (src/Automaton.c:460-479)
PyObject* automaton_match(AutomatonObject* a, PyObject* args) {
    if (PyTuple_GET_SIZE(args) < 1) { PyErr_SetString(PyExc_TypeError, "match() takes a key"); return nullptr; }
    Text kt;
    if (!get_text(PyTuple_GET_ITEM(args, 0), &kt, false, a->key_type)) return nullptr;
    if (acx_trie_kind(a->trie) == K_EMPTY) Py_RETURN_FALSE;         // trie_find on a NULL root (src/trie.c:136-152)
    size_t n = 0;
    int rc = acx_trie_longest_prefix(a->trie, kt.data, (size_t)kt.nbytes, &n);
    if (rc) return set_acx_error(rc);
    if (n == (size_t)kt.nbytes) Py_RETURN_TRUE;
    Py_RETURN_FALSE;
}

// dump() -> (nodes, edges, fail) like src/Automaton.c:1098-1180, in pre-order; ids are 1-based dump
// numbers (the reference prints node addresses).  Read off the pickle records libacx writes.
PyObject* automaton_dump(AutomatonObject* a, PyObject*) {
    if (acx_trie_kind(a->trie) == K_EMPTY) Py_RETURN_NONE;
    void* buf = nullptr; size_t* sizes = nullptr; size_t n_chunks = 0;
    int rc = acx_trie_to_ref_pickle(a->trie, 0, 0, 2, 0, &buf, &sizes, &n_chunks);
    if (rc) return set_acx_error(rc);
    PyObject *nodes = PyList_New(0), *edges = PyList_New(0), *fail = PyList_New(0);
    bool ok = nodes && edges && fail;
    const uint8_t* p = (const uint8_t*)buf;
    long long id = 0;
    auto append = [&](PyObject* list, PyObject* tuple) { if (!tuple || PyList_Append(list, tuple) < 0) ok = false; Py_XDECREF(tuple); };
    for (size_t c = 0; ok && c < n_chunks; c++) {
        const uint8_t* q = p;
        int64_t cnt; memcpy(&cnt, q, 8); q += 8;
        for (int64_t i = 0; ok && i < cnt; i++) {
            id++;
            uint64_t f; uint32_t nch; memcpy(&f, q + 8, 8); memcpy(&nch, q + 16, 4);
            const int eow = q[20];
            q += 24;
            append(nodes, Py_BuildValue("Li", id, eow));
            for (uint32_t j = 0; ok && j < nch; j++, q += 10) {
                uint16_t letter; uint64_t child; memcpy(&letter, q, 2); memcpy(&child, q + 2, 8);
                const char ch = (char)(letter & 0xFF);
                append(edges, Py_BuildValue("Ly#L", id, &ch, (Py_ssize_t)1, (long long)child));
            }
            if (f) append(fail, Py_BuildValue("LL", id, (long long)f));
        }
        p += sizes[c];
    }
    acx_blob_free(buf); acx_blob_free(sizes);
    if (!ok) { Py_XDECREF(nodes); Py_XDECREF(edges); Py_XDECREF(fail); return nullptr; }
    return Py_BuildValue("NNN", nodes, edges, fail);
}

PyObject* automaton_sizeof(AutomatonObject* a, PyObject*) {          // src/Automaton.c:1183-1198
    int64_t total = 0;
    if (acx_trie_kind(a->trie) != K_EMPTY) (void)acx_trie_stats(a->trie, nullptr, nullptr, nullptr, nullptr, nullptr, &total);
    return PyLong_FromLongLong((long long)sizeof(AutomatonObject) + total);
}

// ---- keys() / values() / items() / __iter__ (src/Automaton.c:722-873, src/AutomatonItemsIter.c) -------
// The enumeration is done in libacx (acx_items.cpp) in the reference's order; the iterator hands the
// items out one by one and, like the reference's, dies when the automaton changes.
enum { ITER_KEYS = 0, ITER_VALUES = 1, ITER_ITEMS = 2 };

struct ItemsIterObject {
    PyObject_HEAD
    AutomatonObject* automaton;
    int64_t version;
    int what;
    uint8_t* keys; int64_t* key_off; int64_t* values;
    int64_t n, pos;
};

void items_iter_dealloc(ItemsIterObject* it) {
    acx_blob_free(it->keys); acx_blob_free(it->key_off); acx_blob_free(it->values);
    Py_XDECREF(it->automaton);
    Py_TYPE(it)->tp_free((PyObject*)it);
}

PyObject* items_iter_iter(PyObject* self) { Py_INCREF(self); return self; }

PyObject* items_iter_next(ItemsIterObject* it) {
    if (it->version != acx_trie_version(it->automaton->trie)) {
        PyErr_SetString(PyExc_ValueError, "The underlying automaton has changed: this iterator is no longer valid.");
        return nullptr;
    }
    if (it->pos >= it->n) return nullptr;                             // StopIteration
    const int64_t i = it->pos++;
    PyObject* key = nullptr;
    if (it->what != ITER_VALUES) {
        key = make_key_object(it->keys + it->key_off[i], (Py_ssize_t)(it->key_off[i + 1] - it->key_off[i]), it->automaton->key_type);
        if (!key || it->what == ITER_KEYS) return key;
    }
    PyObject* val;
    if (it->automaton->store == STORE_ANY) {
        val = PyList_GetItem(it->automaton->values, (Py_ssize_t)it->values[i]);      // borrowed
        Py_XINCREF(val);
    } else {
        val = Py_BuildValue("i", (int)it->values[i]);               // as the reference: the low 32 bits
    }
    if (!val) { Py_XDECREF(key); return nullptr; }
    if (it->what == ITER_VALUES) return val;
    return Py_BuildValue("(NN)", key, val);
}

PyTypeObject ItemsIterType = {PyVarObject_HEAD_INIT(nullptr, 0) "ahocorasick.AutomatonItemsIter"};

// argument rules of automaton_items_create, src/Automaton.c:722-850
PyObject* automaton_items_create(AutomatonObject* a, PyObject* args, int what) {
    const Py_ssize_t na = args ? PyTuple_GET_SIZE(args) : 0;
    Text pt, wt;
    const uint8_t* pat = nullptr; Py_ssize_t plen = 0;
    if (na >= 1) {
        if (!get_pattern(PyTuple_GET_ITEM(args, 0), &pt, a->key_type)) return nullptr;
        pat = pt.data; plen = pt.nbytes;
    }
    int use_wildcard = 0;
    const uint8_t* wild = nullptr; Py_ssize_t wild_len = 0;
    if (na >= 2) {
        if (!get_pattern(PyTuple_GET_ITEM(args, 1), &wt, a->key_type)) return nullptr;
        if (wt.nchars != 1) { PyErr_SetString(PyExc_ValueError, "Wildcard must be a single character."); return nullptr; }
        use_wildcard = 1; wild = wt.data; wild_len = wt.nbytes;
    }
    int how = use_wildcard ? MATCH_EXACT_LENGTH : MATCH_AT_LEAST_PREFIX;
    if (na >= 3) {
        const Py_ssize_t v = PyNumber_AsSsize_t(PyTuple_GET_ITEM(args, 2), PyExc_OverflowError);
        if (v == -1 && PyErr_Occurred()) return nullptr;
        if (v != MATCH_EXACT_LENGTH && v != MATCH_AT_MOST_PREFIX && v != MATCH_AT_LEAST_PREFIX) {
            PyErr_SetString(PyExc_ValueError, "The optional how third argument must be one of: "
                                              "MATCH_EXACT_LENGTH, MATCH_AT_LEAST_PREFIX or MATCH_AT_LEAST_PREFIX");
            return nullptr;
        }
        how = (int)v;
    }
    ItemsIterObject* it = (ItemsIterObject*)ItemsIterType.tp_alloc(&ItemsIterType, 0);
    if (!it) return nullptr;
    it->automaton = nullptr; it->keys = nullptr; it->key_off = nullptr; it->values = nullptr; it->n = 0; it->pos = 0;
    it->what = what;
    // (str build: the trie holds UTF-8; libacx then enumerates letter by letter)
    int rc = acx_trie_items(a->trie, pat, (size_t)plen, wild, (size_t)wild_len, how, ACX_UNICODE_BUILD || a->key_type == KEY_SEQUENCE,
                            &it->keys, &it->key_off, &it->values, &it->n);
    if (rc) { Py_DECREF(it); return set_acx_error(rc); }
    Py_INCREF(a);
    it->automaton = a;
    it->version = acx_trie_version(a->trie);
    return (PyObject*)it;
}

PyObject* automaton_keys(AutomatonObject* a, PyObject* args) { return automaton_items_create(a, args, ITER_KEYS); }
PyObject* automaton_values(AutomatonObject* a, PyObject* args) { return automaton_items_create(a, args, ITER_VALUES); }
PyObject* automaton_items(AutomatonObject* a, PyObject* args) { return automaton_items_create(a, args, ITER_ITEMS); }
PyObject* automaton_tp_iter(PyObject* a) { return automaton_items_create((AutomatonObject*)a, nullptr, ITER_KEYS); }

PyMethodDef automaton_methods[] = {
    {"add_word", (PyCFunction)automaton_add_word, METH_VARARGS, "add_word(key, [value]) -> bool"},
    {"add_words", (PyCFunction)automaton_add_words, METH_VARARGS, "add_words(keys, [values]) -> number of new keys (extension: add_word for every pair)"},
    {"exists", (PyCFunction)automaton_exists, METH_VARARGS, "exists(key) -> bool"},
    {"get", (PyCFunction)automaton_get, METH_VARARGS, "get(key[, default])"},
    {"longest_prefix", (PyCFunction)automaton_longest_prefix, METH_VARARGS, "longest_prefix(key) -> int"},
    {"remove_word", (PyCFunction)automaton_remove_word, METH_VARARGS, "remove_word(key) -> bool"},
    {"pop", (PyCFunction)automaton_pop, METH_VARARGS, "pop(key) -> value"},
    {"clear", (PyCFunction)automaton_clear, METH_NOARGS, "clear()"},
    {"make_automaton", (PyCFunction)automaton_make_automaton, METH_NOARGS, "make_automaton()"},
    {"iter", (PyCFunction)automaton_iter, METH_VARARGS | METH_KEYWORDS, "iter(string, [start, [end]], ignore_white_space=False)"},
    {"iter_long", (PyCFunction)automaton_iter_long, METH_VARARGS, "iter_long(string, [start, [end]])"},
    {"find_all", (PyCFunction)automaton_find_all, METH_VARARGS, "find_all(string, callback, [start, [end]])"},
    {"iter_batch", (PyCFunction)automaton_iter_batch, METH_VARARGS | METH_KEYWORDS, "iter_batch(haystacks, long=False) -> list of lists (GPU batch scan)"},
    {"get_stats", (PyCFunction)automaton_get_stats, METH_NOARGS, "get_stats() -> dict"},
    {"match", (PyCFunction)automaton_match, METH_VARARGS, "match(key) -> bool: key is a prefix of some key"},
    {"dump", (PyCFunction)automaton_dump, METH_NOARGS, "dump() -> (nodes, edges, fail)"},
    {"__sizeof__", (PyCFunction)automaton_sizeof, METH_NOARGS, "size in bytes (as the reference's pointer trie would take)"},
    {"keys", (PyCFunction)automaton_keys, METH_VARARGS, "keys([prefix, [wildcard, [how]]]) -> iterator"},
    {"values", (PyCFunction)automaton_values, METH_VARARGS, "values([prefix, [wildcard, [how]]]) -> iterator"},
    {"items", (PyCFunction)automaton_items, METH_VARARGS, "items([prefix, [wildcard, [how]]]) -> iterator"},
    {"__reduce__", (PyCFunction)automaton_reduce, METH_NOARGS, "pickle support: the reference's (bytes build) payload"},
    {"save", (PyCFunction)automaton_save, METH_VARARGS, "save(path[, serializer]): the reference's file format"},
    {nullptr, nullptr, 0, nullptr}};

PyGetSetDef automaton_getset[] = {
    {"kind", (getter)automaton_get_kind, nullptr, "EMPTY, TRIE or AHOCORASICK", nullptr},
    {"store", (getter)automaton_get_store, nullptr, "STORE_INTS, STORE_LENGTH or STORE_ANY", nullptr},
    {nullptr, nullptr, nullptr, nullptr, nullptr}};

PySequenceMethods automaton_as_sequence = {};

PyTypeObject AutomatonType = {PyVarObject_HEAD_INIT(nullptr, 0) "ahocorasick.Automaton"};

// ahocorasick.load(path, deserializer): src/custompickle/load/module_automaton_load.c:12-36
PyObject* module_load(PyObject*, PyObject* args) {
    const char* path; PyObject* deserializer;
    if (!parse_path_callback(STORE_ANY, args, &path, &deserializer)) return nullptr;   // both arguments, always
    FILE* f = fopen(path, "rb");
    if (!f) return PyErr_SetFromErrnoWithFilename(PyExc_IOError, path);
    std::vector<uint8_t> data;
    uint8_t chunk[1 << 16];
    for (size_t got; (got = fread(chunk, 1, sizeof chunk, f)) > 0;) data.insert(data.end(), chunk, chunk + got);
    fclose(f);
    acx_trie_t* trie = nullptr; acx_ref_meta_t meta; int64_t *poff = nullptr, *plen = nullptr;
    int rc = acx_trie_from_ref_savefile(data.data(), data.size(), ACX_UNICODE_BUILD ? 4 : 2, &trie, &meta, &poff, &plen);
    if (rc) {
        if (rc == ACX_E_NOMEM) return PyErr_NoMemory();
        PyErr_SetString(PyExc_ValueError, acx_last_error());
        return nullptr;
    }
    AutomatonObject* a = nullptr;
    if ((meta.kind == K_EMPTY || meta.kind == K_TRIE || meta.kind == K_AHOCORASICK) && check_store_key(meta.store, meta.key_type))
        a = automaton_alloc(&AutomatonType, meta.store, meta.key_type);
    else if (!PyErr_Occurred()) PyErr_SetString(PyExc_ValueError, "invalid header");
    bool ok = a != nullptr;
    if (ok) {
        a->trie = trie; trie = nullptr;
        if (!a->trie) ok = acx_trie_new(&a->trie) == ACX_OK;
        if (ok && meta.store == STORE_ANY) {
            a->values = PyList_New((Py_ssize_t)meta.n_eow);
            ok = a->values != nullptr;
            for (int64_t k = 0; ok && k < meta.n_eow; k++) {
                PyObject* b = PyBytes_FromStringAndSize((const char*)data.data() + poff[k], (Py_ssize_t)plen[k]);
                PyObject* v = b ? PyObject_CallFunctionObjArgs(deserializer, b, nullptr) : nullptr;
                Py_XDECREF(b);
                if (!v) { ok = false; break; }
                PyList_SET_ITEM(a->values, (Py_ssize_t)k, v);
            }
            if (!ok && a->values) {                       // fill the holes so that dealloc is safe
                for (Py_ssize_t k = 0; k < PyList_GET_SIZE(a->values); k++)
                    if (!PyList_GET_ITEM(a->values, k)) { Py_INCREF(Py_None); PyList_SET_ITEM(a->values, k, Py_None); }
            }
        }
        if (ok && meta.kind == K_AHOCORASICK && acx_trie_num_keys(a->trie) > 0) {
            rc = acx_trie_make_automaton(a->trie, nullptr);
            if (rc) { set_acx_error(rc); ok = false; }
        }
    }
    if (trie) acx_trie_free(trie);
    acx_blob_free(poff); acx_blob_free(plen);
    if (!ok) { Py_XDECREF(a); if (!PyErr_Occurred()) PyErr_SetString(PyExc_RuntimeError, acx_last_error()); return nullptr; }
    return (PyObject*)a;
}

// NEW (not in the reference): which scans are walked over the host trie instead of launched (include/acx.h §4b)
PyObject* module_set_host_walk_bytes(PyObject*, PyObject* args) {
    long long n;
    if (!PyArg_ParseTuple(args, "L", &n)) return nullptr;
    acx_set_host_walk_bytes((int64_t)n);
    Py_RETURN_NONE;
}
PyObject* module_host_walk_bytes(PyObject*, PyObject*) { return PyLong_FromLongLong((long long)acx_host_walk_bytes()); }
PyObject* module_host_walk_calls(PyObject*, PyObject*) { return PyLong_FromLongLong((long long)acx_host_walk_calls()); }

PyMethodDef module_methods[] = {
    {"load", (PyCFunction)module_load, METH_VARARGS, "load(path, deserializer) -> Automaton: read a file written by Automaton.save"},
    {"set_host_walk_bytes", (PyCFunction)module_set_host_walk_bytes, METH_VARARGS,
     "set_host_walk_bytes(n): searches of at most n bytes walk the host trie instead of launching a GPU scan (default 2048; "
     "-1: never, a process without a GPU then raises; with no GPU every search of up to 1 MiB is walked on the host)"},
    {"host_walk_bytes", (PyCFunction)module_host_walk_bytes, METH_NOARGS, "the limit set by set_host_walk_bytes"},
    {"host_walk_calls", (PyCFunction)module_host_walk_calls, METH_NOARGS, "searches of this process that were walked on the host"},
    {nullptr, nullptr, 0, nullptr}};

PyModuleDef module_def = {PyModuleDef_HEAD_INIT, "ahocorasick",
                          "MI355X-native Aho-Corasick scan engine behind the pyahocorasick Automaton API (bytes build)",
                          -1, module_methods, nullptr, nullptr, nullptr, nullptr};

}  // namespace

PyMODINIT_FUNC PyInit_ahocorasick(void) {
    AutomatonType.tp_basicsize = sizeof(AutomatonObject);
    AutomatonType.tp_flags = Py_TPFLAGS_DEFAULT;
    AutomatonType.tp_doc = "Automaton(value_type=STORE_ANY, key_type=KEY_STRING)";
    AutomatonType.tp_new = automaton_new;
    AutomatonType.tp_dealloc = (destructor)automaton_dealloc;
    AutomatonType.tp_methods = automaton_methods;
    AutomatonType.tp_getset = automaton_getset;
    automaton_as_sequence.sq_length = (lenfunc)automaton_len;
    automaton_as_sequence.sq_contains = (objobjproc)automaton_contains;
    AutomatonType.tp_as_sequence = &automaton_as_sequence;
    AutomatonType.tp_iter = automaton_tp_iter;
    if (PyType_Ready(&AutomatonType) < 0) return nullptr;

    ItemsIterType.tp_basicsize = sizeof(ItemsIterObject);
    ItemsIterType.tp_flags = Py_TPFLAGS_DEFAULT;
    ItemsIterType.tp_dealloc = (destructor)items_iter_dealloc;
    ItemsIterType.tp_iter = items_iter_iter;
    ItemsIterType.tp_iternext = (iternextfunc)items_iter_next;
    if (PyType_Ready(&ItemsIterType) < 0) return nullptr;

    SearchIterType.tp_basicsize = sizeof(SearchIterObject);
    SearchIterType.tp_flags = Py_TPFLAGS_DEFAULT;
    SearchIterType.tp_dealloc = (destructor)search_iter_dealloc;
    SearchIterType.tp_iter = search_iter_iter;
    SearchIterType.tp_iternext = (iternextfunc)search_iter_next;
    SearchIterType.tp_methods = search_iter_methods;
    if (PyType_Ready(&SearchIterType) < 0) return nullptr;

    PyObject* m = PyModule_Create(&module_def);
    // ACX_HOST_WALK_BYTES: the limit of set_host_walk_bytes for processes that cannot call it first (the GPU test-suite runs the
    // reference's own tests against this module with ACX_HOST_WALK_BYTES=-1: every search there is a GPU scan)
    if (const char* e = getenv("ACX_HOST_WALK_BYTES")) { if (*e) acx_set_host_walk_bytes((int64_t)atoll(e)); }
    if (!m) return nullptr;
    Py_INCREF(&AutomatonType);
    if (PyModule_AddObject(m, "Automaton", (PyObject*)&AutomatonType) < 0) return nullptr;
#define ADD_INT(name, value) if (PyModule_AddIntConstant(m, name, value) < 0) return nullptr
    ADD_INT("TRIE", K_TRIE); ADD_INT("AHOCORASICK", K_AHOCORASICK); ADD_INT("EMPTY", K_EMPTY);
    ADD_INT("STORE_LENGTH", STORE_LENGTH); ADD_INT("STORE_INTS", STORE_INTS); ADD_INT("STORE_ANY", STORE_ANY);
    ADD_INT("KEY_STRING", KEY_STRING); ADD_INT("KEY_SEQUENCE", KEY_SEQUENCE);
    ADD_INT("MATCH_EXACT_LENGTH", MATCH_EXACT_LENGTH); ADD_INT("MATCH_AT_MOST_PREFIX", MATCH_AT_MOST_PREFIX);
    ADD_INT("MATCH_AT_LEAST_PREFIX", MATCH_AT_LEAST_PREFIX);
    ADD_INT("unicode", ACX_UNICODE_BUILD);
#undef ADD_INT
    return m;
}
