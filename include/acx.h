/*
 * acx.h — C-ABI of the MI355X-native Aho-Corasick scan engine (libacx.so).
 *
 * This is the drop-in boundary for ONE hot path of WojciechMula/pyahocorasick:
 * the per-byte goto/fail/output walk of AutomatonSearchIter / AutomatonSearchIterLong
 * over a batch of haystacks.  The reference has no FFI of its own (everything is
 * `static` inside one CPython translation unit), so each entry point below names the
 * reference function(s) whose work it replaces (paths relative to the reference
 * repository root).  INTEGRATION.md shows the binding a maintainer of the reference
 * would add to src/Automaton.c to call these.
 *
 * Conventions
 *   - plain C, plain pointers and sizes; no Python, no torch types.
 *   - every function returns ACX_OK (0) or a negative acx_status; acx_last_error()
 *     returns a thread-local human-readable message for the last failure.
 *   - "host" pointers are ordinary process memory, "dev" pointers are HIP device
 *     memory on the device that was current when the image was created.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *   - integer widths follow the reference: end_index and value are C `int`
 *     (Py_BuildValue("ii"), src/AutomatonSearchIter.c:180-184), i.e. int32.
 */
#ifndef ACX_H_INCLUDED
#define ACX_H_INCLUDED

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* libacx.so is built with -fvisibility=hidden: the entry points declared in this header — and nothing else, no kernel
 * launcher, no C++ internals — are what the library exports (tests/test_capi_symbols.py reads `nm -D`). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define ACX_ABI_VERSION 4          /* 3: acx_scan_params.dev_skip, acx_scan_host_ctx, acx_trie_add_words, ACX_SCAN_SKIP_WS
                                    * 4: acx_trie_scan_host, acx_set_host_walk_bytes / acx_host_walk_applies / acx_host_walk_calls, acx_blob_long_pack,
                                    *    acx_image_set_long / acx_image_long_state, ACX_FLATTEN_HOT12, acx_async_streams; the values that acx_blob_long_trie
                                    *    returns carry kind 3 and, for dictionaries below 2^18 entries, `below` in bits 18-23 (section 3b) */

typedef enum acx_status {
    ACX_OK            =  0,
    ACX_E_INVAL       = -1,  /* bad argument */
    ACX_E_NOMEM       = -2,  /* host or device allocation failed (reference: MemoryError) */
    ACX_E_STATE       = -3,  /* wrong automaton kind, e.g. scan before make_automaton
                                (reference: AttributeError, src/Automaton.c:886-891) */
    ACX_E_HIP         = -4,  /* a HIP runtime call failed */
    ACX_E_UNSUPPORTED = -5,  /* automaton does not fit the device layout of this build */
    ACX_E_FORMAT      = -6,  /* malformed flat image */
    ACX_E_NODEVICE    = -7   /* no usable GPU: the product path never falls back to CPU */
} acx_status;

const char* acx_last_error(void);
int         acx_abi_version(void);

/* ------------------------------------------------------------------------------------
 * 1. Host trie (CPU).  Stays on the CPU exactly as in the reference.
 *    Replaces: trie_add_word (src/trie.c:14-63), automaton_add_word value rules
 *    (src/Automaton.c:201-300), trie_find (src/trie.c:136-152),
 *    automaton_make_automaton (src/Automaton.c:560-649).
 *    Keys are byte strings (the reference's bytes build, KEY_STRING); a letter is one
 *    byte (src/utils.c:191-205 widens each signed char to uint16 — injective on bytes).
 * ---------------------------------------------------------------------------------- */
typedef struct acx_trie acx_trie_t;

enum { ACX_KIND_EMPTY = 0, ACX_KIND_TRIE = 1, ACX_KIND_AHOCORASICK = 2 };  /* src/Automaton.h:16-20 */

int  acx_trie_new(acx_trie_t** out);
void acx_trie_free(acx_trie_t* t);
/* add (or overwrite the value of) a key.  *is_new = 1 if the key was not present.
 * len == 0 is accepted and ignored (*is_new = 0), as src/Automaton.c:257 does. */
int  acx_trie_add_word(acx_trie_t* t, const uint8_t* key, size_t len, int64_t value, int* is_new);
/* Many keys in one call (no counterpart in the reference, which adds one word per Python call: src/Automaton.c:201-300
 * — 4.4 s of interpreter time for a million signatures).  Key i = keys[key_off[i] .. key_off[i+1]); its value is
 * values[i], or with values == NULL what add_word's defaults give: value_mode 1 = the number of keys after the
 * insertion (STORE_INTS default, src/Automaton.c:238-242), 2 = the key's length in bytes (STORE_LENGTH, :245-247).
 * Same semantics as n calls of acx_trie_add_word, in order; *n_new = how many keys were not present before. */
int  acx_trie_add_words(acx_trie_t* t, const uint8_t* keys, const int64_t* key_off, const int64_t* values, int64_t n,
                        int value_mode, int64_t* n_new);
/* exact lookup; *found = 1 and *value set when key is present (trie_find + eow test) */
int  acx_trie_get(const acx_trie_t* t, const uint8_t* key, size_t len, int* found, int64_t* value);
/* remove a key (trie_remove_word, src/trie.c:66-133); *found = 0 when absent */
int  acx_trie_remove_word(acx_trie_t* t, const uint8_t* key, size_t len, int* found, int64_t* value);
/* length of the longest prefix of `key` present as a path in the trie (trie_longest) */
int  acx_trie_longest_prefix(const acx_trie_t* t, const uint8_t* key, size_t len, size_t* out_len);
void acx_trie_clear(acx_trie_t* t);
/* BFS failure links.  Returns ACX_OK and *changed = 1 when links were (re)built,
 * *changed = 0 when kind != TRIE (reference returns False, src/Automaton.c:574-575). */
int  acx_trie_make_automaton(acx_trie_t* t, int* changed);
int      acx_trie_kind(const acx_trie_t* t);
int64_t  acx_trie_num_keys(const acx_trie_t* t);
int64_t  acx_trie_num_nodes(const acx_trie_t* t);
int64_t  acx_trie_longest_word(const acx_trie_t* t);
/* bumped by add_word(new key) / remove_word / clear / make_automaton
 * (src/Automaton.c:284,342,364,413,640); device images are tagged with it. */
int64_t  acx_trie_version(const acx_trie_t* t);

/* Enumeration (dict-like methods; acx_items.cpp): the keys — and their values —
 * that Automaton.keys/values/items([prefix, [wildcard, [how]]]) iterate over, in the reference's
 * order (automaton_items_iter_next, src/AutomatonItemsIter.c:124-209).  how: 0 exact length,
 * 1 at most, 2 at least the pattern's length (src/Automaton.h:43-47).  keys are returned back to
 * back, key i = keys[key_off[i] .. key_off[i+1]); the three arrays are malloc'd (acx_blob_free).
 * wildcard: the bytes of the one wildcard letter, wlen = 0 for none.  letters_utf8 = 1 (str build):
 * the trie holds UTF-8 and a letter is one whole sequence — depth, pattern and wildcard count
 * characters, children are visited in the order their letters were first added. */
int acx_trie_items(const acx_trie_t* t, const uint8_t* pattern, size_t plen, const uint8_t* wildcard, size_t wlen,
                   int how, int letters_utf8, uint8_t** keys, int64_t** key_off, int64_t** values, int64_t* n);
/* Automaton.get_stats (src/Automaton.c:1044-1096); sizes are those of the reference's pointer trie */
int acx_trie_stats(const acx_trie_t* t, int64_t* nodes, int64_t* words, int64_t* longest, int64_t* links,
                   int64_t* sizeof_node, int64_t* total_size);

/* ------------------------------------------------------------------------------------
 * 1b. The reference's persistence formats <-> the trie (acx_persist.cpp).
 *     LP64.  Dumps of the bytes build (letters stored as uint16) are read and written; dumps of the
 *     unicode build (uint32 code points) are read: their keys are re-inserted as UTF-8.
 *     Loading a bytes-build dump rebuilds the trie node for node (same child order);
 *     fail links are not taken from the dump: call acx_trie_make_automaton afterwards when
 *     the dump was an automaton.  Out-arrays are malloc'd: release with acx_blob_free.
 *     values_by_position (STORE_ANY): the values travel outside the dump, one per key in
 *     dump order; a loaded trie then stores that position as the key's value.  Dumps are written
 *     in node creation order (parents first; the reference writes pre-order, reads any order).
 * ---------------------------------------------------------------------------------- */
enum { ACX_STORE_LENGTH = 20, ACX_STORE_INTS = 10, ACX_STORE_ANY = 30 };    /* src/Automaton.h:22-27 */
enum { ACX_KEY_STRING = 100, ACX_KEY_SEQUENCE = 200 };                      /* src/Automaton.h:29-32 */
typedef struct acx_ref_meta {       /* header of a save file, src/custompickle/custompickle.h:5-17 */
    int32_t kind, store, key_type, reserved;
    int64_t count, longest_word, n_nodes, n_eow;
} acx_ref_meta_t;
/* Automaton.__reduce__ payload: the list of byte chunks (src/Automaton_pickle.c:128-285 writes it,
 * automaton_unpickle :326-488 reads it) */
int acx_trie_from_ref_pickle(const void* const* chunks, const size_t* chunk_bytes, size_t n_chunks,
                             int values_by_position, int64_t longest_word /* of the tuple; never lowered by remove_word */,
                             int letter_bytes /* 2: bytes build; 4: unicode build (code points -> UTF-8 keys) */,
                             int sequence /* 1: a KEY_SEQUENCE dump: letters are integers, re-encoded like the hosts do */,
                             acx_trie_t** out, int64_t* n_eow);
/* letter_bytes: 2 writes the bytes build's layout (uint16 letters), 4 the unicode build's (uint32).
 * letters_multibyte = 0: every byte of the trie is a letter (byte keys).  1: the trie stores letters
 * as multi-byte sequences (str flavour, KEY_SEQUENCE): they are decoded on the way out, so the dump
 * is a genuine dump of the reference build that `letter_bytes` names. */
int acx_trie_to_ref_pickle(const acx_trie_t* t, int values_by_position, size_t chunk_limit, int letter_bytes,
                           int letters_multibyte, void** buf, size_t** chunk_bytes, size_t* n_chunks);   /* chunks back to back in buf */
/* values of the keys in the order to_ref_pickle / to_ref_savefile write them (what the `values` list
 * of a STORE_ANY pickle is ordered by) */
int acx_trie_eow_values(const acx_trie_t* t, int letters_multibyte, int64_t** values, int64_t* n);
/* Automaton.save file (src/custompickle/save/automaton_save.c:36-138; loader
 * src/custompickle/load/module_automaton_load.c).  payload_off/len: byte range of the
 * serialized value of each key, in dump order (STORE_ANY only, else NULL); *out is NULL
 * for the file of an empty automaton. */
int acx_trie_from_ref_savefile(const void* data, size_t nbytes, int letter_bytes, acx_trie_t** out, acx_ref_meta_t* meta,
                               int64_t** payload_off, int64_t** payload_len);
int acx_trie_to_ref_savefile(const acx_trie_t* t, int store, int key_type, int letter_bytes, int letters_multibyte,
                             const void* const* payloads, const size_t* payload_bytes, void** buf, size_t* nbytes);

/* ------------------------------------------------------------------------------------
 * 2. Flat image.  One contiguous, relocatable little-endian blob (layout: acx_blob.h):
 *    class map, dense fail-resolved transition table, fail vector, CSR output lists.
 *    Built from a finalised trie on the CPU; it is what gets uploaded to HBM and what a
 *    single RCCL broadcast replicates to the other GPUs of the node.
 *    Replaces the pointer graph of src/trienode.h:19-42 as the thing the scan reads.
 * ---------------------------------------------------------------------------------- */
int  acx_flatten(const acx_trie_t* t, void** blob, size_t* nbytes);   /* malloc'd */
/* The same with layout options (0 = what acx_flatten chooses): the image is valid whatever is asked for, only laid out
 * otherwise — the test-suite walks every layout on small automata with them. */
enum { ACX_FLATTEN_NO_PPM       = 1,    /* no position-parallel section: ACX_SCAN_ALL takes the serial walk kernels */
       ACX_FLATTEN_WIDE         = 2,    /* the wide entry layout (27-bit states) although the narrow one would fit */
       ACX_FLATTEN_NO_ITOP      = 4,    /* no implicit top-of-trie structures */
       ACX_FLATTEN_TABLE_HOST   = 8,    /* the dense table is built on the host and travels in the blob */
       ACX_FLATTEN_TABLE_DEVICE = 16,   /* the blob carries the sparse form only: the table is built in HBM (default above 64 MiB) */
       ACX_FLATTEN_HOT12        = 32 }; /* four-letter alphabets: k_ppm_stream4's hot cells as 12 bytes — the value of the shallowest key AND the id of
                                           the depth-C node — instead of 8 + cid[] (include/acx_blob.h).  For dictionaries in which the cells that send
                                           a walk deeper mostly end a key as well: the dictionaries of iter_long (acx_blob_long_pack uses it) */
int  acx_flatten_ex(const acx_trie_t* t, uint32_t flags, void** blob, size_t* nbytes);
void acx_blob_free(void* blob);
int  acx_blob_validate(const void* blob, size_t nbytes);             /* host blob */

/* The dictionary that ACX_SCAN_LONG scans position-parallel (acx_long.cpp): iter_long — automaton_search_iter_long_next,
 * src/AutomatonSearchIterLong.c:89-153, a serial state machine — only ever reports nodes that end a key (E) or whose fail
 * node does while they do not (FE), and whether it stops at one depends on the next LONGER path of the trie that ends
 * with it (U: the nodes whose fail node is an E or FE node).  A trie of exactly these node strings, each with the value
 * index | length << 24 | kind << 30  (kind 0 U, 1 E, 2 FE, 3 an E node with no E / FE node below it in the trie: reported the
 * moment it is reached; real_vals[index] = what iter_long reports for the node; when the dictionary has fewer than
 * 2^ACX_LONG_SMALL_BITS entries the index takes that many bits and the six above it say how many letters below the node the deepest
 * E / FE node of its subtree lies: how far the sweep looks ahead along a remembered node's path),
 * finalised.  ACX_SCAN_ALL over it gives, per haystack, the records a single sweep turns into iter_long's output; the
 * device side builds it from an image on the first ACX_SCAN_LONG scan.  *n = 0 (and no trie) when the form does not apply:
 * a node deeper than 63 letters, 2^24 nodes or more.  Host only; the trie is the caller's (acx_trie_free), real_vals is
 * malloc'd (acx_blob_free). */
#define ACX_LONG_SMALL_BITS 18
int  acx_blob_long_trie(const void* blob, size_t nbytes, acx_trie_t** out_trie, int32_t** real_vals, int64_t* n, int32_t* longest);

/* The same dictionary as a relocatable PACK (a 256-byte header, the dictionary's flat image, the reported values), built once from the
 * host blob: with N GPUs the pack travels behind the blob in the ONE broadcast of the set-up (pyahocorasick_amd/parallel.py
 * broadcast_image(long_pack=True)) and acx_image_set_long installs it on every rank — without it every rank's first ACX_SCAN_LONG scan
 * copies its whole image back to the host and builds the dictionary there.  A pack whose header says "does not apply" is valid: the
 * image then keeps the serial walk.  malloc'd (acx_blob_free). */
int  acx_blob_long_pack(const void* blob, size_t nbytes, void** pack, size_t* pack_bytes);

typedef struct acx_image acx_image_t;
/* copy a host blob to the current HIP device */
int  acx_image_upload(const void* blob, size_t nbytes, acx_image_t** out);
/* adopt a blob that is ALREADY in device memory (e.g. the receive buffer of the RCCL
 * broadcast).  The image does not own `dev_blob`; the caller keeps it alive.
 * `host_header` = the first ACX_BLOB_HEADER_BYTES of the blob in host memory. */
int  acx_image_adopt(void* dev_blob, size_t nbytes, const void* host_header, acx_image_t** out);
/* Replicate a flat image to every rank of an RCCL communicator with ONE ncclBroadcast of the blob over xGMI
 * (plus an 8-byte one for its size) and adopt it in place: the multi-GPU set-up step of SURVEY.md §8(e) for
 * hosts that have no torch (the reference's own C extension).  The scan itself needs no
 * collective: every rank scans its own shard of the haystacks.
 *   host_blob, nbytes : the blob from acx_flatten; read on `root` only (NULL / 0 elsewhere)
 *   nccl_comm         : the caller's ncclComm_t (RCCL), passed as void*; this rank's device must be current
 *   stream            : hipStream_t the broadcast is queued on (NULL = default); the call returns after it completed
 * RCCL is resolved at run time from the process (the library that created `nccl_comm`), else librccl.so is
 * opened: libacx has no link-time dependency on it.  The image owns its device copy. */
int  acx_image_broadcast(const void* host_blob, size_t nbytes, void* nccl_comm, int root, int rank, void* stream, acx_image_t** out);
void acx_image_free(acx_image_t* img);
/* install the pack of acx_blob_long_pack on an image that has not scanned in ACX_SCAN_LONG mode yet.  on_device = 0: `pack` is host
 * memory (the dictionary is uploaded); 1: it is device memory — e.g. the tail of the broadcast's receive buffer — and is adopted IN
 * PLACE: the caller keeps it alive as long as the image. */
int  acx_image_set_long(acx_image_t* img, const void* pack, size_t pack_bytes, int on_device);
int  acx_image_long_state(const acx_image_t* img);     /* 1: dictionary present, -1: the form does not apply (serial walk), 0: not built yet */
int64_t acx_image_num_states(const acx_image_t* img);
int64_t acx_image_num_classes(const acx_image_t* img);
size_t  acx_image_nbytes(const acx_image_t* img);
void*   acx_image_dev_ptr(const acx_image_t* img);
/* depth D of the implicit top-of-trie the image carries (0 = none): states shallower than D
 * are walked from LDS-resident k-gram bitmaps instead of table rows (include/acx_blob.h) */
int     acx_image_itop_depth(const acx_image_t* img);
/* device address of the dense transition table the scans read: inside the blob, or the table
 * built in HBM by acx_image_upload/adopt when the blob carries only the sparse form */
const void* acx_image_table_dev_ptr(const acx_image_t* img);

/* ------------------------------------------------------------------------------------
 * 3. Batch scan — THE hot path.
 *    mode ACX_SCAN_ALL  : every match, reference order (position ascending; within a
 *                         position the state first, then its fail chain = longest key
 *                         first).  Replaces automaton_search_iter_next +
 *                         automaton_build_output + ahocorasick_next
 *                         (src/AutomatonSearchIter.c:157-197, 243-300; src/trie.c:177-194)
 *                         and the inlined loop of automaton_find_all
 *                         (src/Automaton.c:693-714).
 *    mode ACX_SCAN_LONG : the exact state machine of automaton_search_iter_long_next
 *                         (src/AutomatonSearchIterLong.c:89-153).
 *
 *    Input: n_hay haystacks concatenated in device buffer `dev_hay`;
 *           haystack h = bytes [off[h], off[h+1])  (dev_off: int64[n_hay+1], device; off[0] = 0, monotone:
 *           bytes in front of off[0] would belong to no haystack — the kernels report nothing for them),
 *           or, when dev_off == NULL, bytes [h*stride, (h+1)*stride) (fixed-length reads).
 *           hay_capacity = number of readable bytes at dev_hay (>= the last offset).
 *    Optional per-haystack device arrays (NULL = absent):
 *           dev_init_state  int32[n_hay]  start state  (streaming continuation; the
 *                           state that AutomatonSearchIter.set(chunk, reset=False) keeps,
 *                           src/AutomatonSearchIter.c:344-352).  State ids are image ids; an id
 *                           that the image does not have is taken as the root.
 *           dev_index_base  int32[n_hay]  added to every reported end_index (`shift`,
 *                           src/AutomatonSearchIter.c:178, or the `start` of a slice).
 *    Output (acx_result_t, device resident, optionally copied to pinned host memory):
 *           match_off int64[n_hay+1]; matches acx_match_t[match_off[n_hay]] in haystack
 *           order; final_state int32[n_hay] (ACX_SCAN_LONG: the trie node the walk stands on when the
 *           haystack ends — what AutomatonSearchIterLong.set(chunk, reset=False) continues from,
 *           src/AutomatonSearchIterLong.c:194-211).
 * ---------------------------------------------------------------------------------- */
typedef struct acx_match {
    int32_t end_index;   /* index of the LAST byte of the match within its haystack */
    int32_t value;       /* low 32 bits of the stored integer / value id */
} acx_match_t;

enum { ACX_SCAN_ALL = 0, ACX_SCAN_LONG = 1 };

typedef struct acx_scan_params {
    uint32_t struct_bytes;     /* = sizeof(acx_scan_params); ABI growth guard */
    int32_t  mode;             /* ACX_SCAN_ALL | ACX_SCAN_LONG */
    const uint8_t* dev_hay;
    int64_t  hay_capacity;
    const int64_t* dev_off;    /* NULL => fixed stride */
    int64_t  stride;
    int64_t  n_hay;
    const int32_t* dev_init_state;
    const int32_t* dev_index_base;
    int32_t  want_final_state; /* 1 => fill final_state */
    int32_t  timing;           /* 1 => record HIP events around each kernel; 2 => around the walk only
                                  (an event between two kernels costs a few microseconds of idle GPU) */
    int32_t  variant;          /* 0 = default; >0 selects an alternative kernel (bench/tuning) */
    int32_t  flags;            /* ACX_SCAN_ASYNC | ACX_SCAN_SKIP_WS or 0 */
    int32_t  min_hay_len;      /* offsets batches: a lower bound on the haystack lengths that the caller vouches for
                                  (0 = unknown).  With >= 8 the position-parallel stream kernel takes the batch (it
                                  keeps room for one haystack start per eight positions of a tile).  A batch that breaks
                                  the promise is noticed on the device (a flag, like the pool overflow) and scanned
                                  again on the general kernels when the result completes: slower, never wrong.
                                  acx_scan_host derives it from the host offsets. */
    int32_t  reserved0;
    const int32_t* dev_skip;   /* ACX_SCAN_ALL, optional int32[n_hay]: the first dev_skip[h] bytes of haystack h are CONTEXT —
                                  the tail of what a stream delivered before (at most longest_word - 1 bytes matter).
                                  Matches that end inside it are not reported, and end_index counts from the first
                                  byte behind it (then dev_index_base is added).  This is how a stream continues on the
                                  position-parallel kernels: what AutomatonSearchIter.set(chunk, reset=False) does with
                                  a carried state (src/AutomatonSearchIter.c:303-368) follows from the previous
                                  longest_word - 1 bytes alone, so the caller keeps those instead of a state id and
                                  no dense table is needed.  Not for ACX_SCAN_LONG (its restarts need the state). */
} acx_scan_params;
/* Return as soon as the kernels are queued on `stream`.  The result completes (wait for THIS
 * scan's completion event — later scans queued on the same stream keep running —, total read,
 * expand re-run if the match buffer was too small) in acx_result_wait or in the first accessor.
 * Lets one host thread keep several batches in flight, on one stream (the host's bookkeeping
 * overlaps the next batch) or on several (the expand of batch i then overlaps the walk of
 * batch i+1): bench.py --pipeline / --streams.
 * An asynchronous scan is complete when acx_result_wait (or an accessor) returns — NOT when `stream` has
 * drained: the library queues the final copy of the records on a low-priority side stream of its own (a pool of three per
 * device, made at the device's first asynchronous scan: `iter` results share one, `iter_long` results take them in turn), so
 * that it overlaps the next scan kernel that the caller queues on `stream`.
 * Queues.  How much of that overlap a process gets depends on how the HIP runtime maps its streams onto hardware queues: by default
 * it has FOUR per process, and a stream gets one at its first use.  Three scan streams + the library's side stream are four busy
 * queues; a process that also holds RCCL's streams or streams of its own shares queues between them, and a gather that shares a queue
 * with a scan stream runs behind that stream's kernels instead of in the seams between them (config 2 of bench.py: 590-607 GB/s with
 * the side stream on a queue of its own, 565-577 without; profiles/r5_experiments.md section 10).  The environment variable
 * GPU_MAX_HW_QUEUES (read by the HIP runtime when it starts, i.e. it must be set before the first HIP call of the process; bench.py
 * sets 8) raises the number; the library cannot set it for a process whose runtime is already up.  acx_async_streams() says what
 * the library itself holds. */
enum { ACX_SCAN_ASYNC = 1,
/* White space never touches the automaton: AutomatonSearchIter with ignore_white_space=True steps over every letter
 * that iswspace() accepts without changing its state, and reports end indices of the original string
 * (src/AutomatonSearchIter.c:269-274).  Over the letters of a bytes build that is 0x09..0x0D and 0x20.  On the device:
 * the batch is compacted (white space out, the original position of every kept byte remembered), the compacted batch is
 * scanned by the same kernels as any other batch, and the end indices of the records are mapped back — end_index,
 * dev_skip and dev_index_base all count bytes of the batch as the caller gave it.  Batches below 4 GiB. */
       ACX_SCAN_SKIP_WS = 2 };

typedef struct acx_result acx_result_t;

/* `*result` may be NULL (a new result object is created) or a previous result whose
 * device buffers are then reused/grown — the steady-state path allocates nothing.
 * Reusing a result that is still in flight waits for it first. */
int  acx_scan_batch(acx_image_t* img, const acx_scan_params* p, acx_result_t** result, void* stream);
/* Which kernels acx_scan_batch would run for these parameters (nothing is launched): 0 = the serial walks
 * (one lane per haystack or chunk: k_walk_itop / k_walk_all / k_walk_chunks / k_walk_long), 1 = the general
 * position-parallel kernel k_ppm_scan, 2 = the position-parallel stream kernel k_ppm_stream, 3 = its
 * specialisation for fixed-length haystacks over a four-letter alphabet, k_ppm_stream4; -1 on bad arguments.  ACX_SCAN_LONG:
 * 10 + that number for the scan over the dictionary of acx_blob_long_trie when iter_long takes the position-parallel form
 * (the call builds that dictionary's image if it does not exist yet), 0 for the serial walk k_walk_long_sel.  bench.py
 * names the dominant kernel of its roofline entry with it. */
int  acx_scan_plan(const acx_image_t* img, const acx_scan_params* p);
int  acx_result_wait(acx_result_t* r);
/* the side streams the library makes per device for the follow-up work of asynchronous scans (the pool's size; 0: none made, asynchronous
 * scans finish on the caller's stream) */
int  acx_async_streams(void);

/* All accessors below complete the scan (acx_result_wait) as needed. */
int64_t            acx_result_num_matches(acx_result_t* r);
const int64_t*     acx_result_offsets_dev(acx_result_t* r);
const acx_match_t* acx_result_matches_dev(acx_result_t* r);
const int32_t*     acx_result_final_state_dev(acx_result_t* r);
/* D2H into library-owned pinned buffers, valid until the result is reused or freed */
int  acx_result_fetch_host(acx_result_t* r, const int64_t** off, const acx_match_t** matches,
                           const int32_t** final_state);
/* kernel timing of the last scan (ms): walk, scan(prefix sum), expand, total GPU span.
 * Needs params.timing = 1 (or 2: then only walk_ms is measured, scan_ms and expand_ms are 0 and
 * total_ms = walk_ms).  Measured with hipEvents on the scan's stream. */
int  acx_result_timing(acx_result_t* r, float* walk_ms, float* scan_ms, float* expand_ms, float* total_ms);
void acx_result_free(acx_result_t* r);

/* ------------------------------------------------------------------------------------
 * 4. Convenience: scan host buffers end to end (H2D, scan, D2H).  `off` is int64[n_hay+1].
 *    This is what a CPython binding for Automaton.iter()/find_all() on a large haystack,
 *    or a new Automaton.iter_batch(), calls.  PCIe-inclusive by construction.
 *    A batch larger than one launch can stage (4 GiB of haystack) is scanned in groups of whole
 *    haystacks; a large batch of equally long haystacks is scanned as a pipeline of groups whose records are written
 *    straight into the result's pinned host buffers while the next group is uploaded.  Either way the result of these
 *    entry points is what acx_result_fetch_host returns; the *_dev accessors are for acx_scan_batch.
 * ---------------------------------------------------------------------------------- */
int  acx_scan_host(acx_image_t* img, int mode, const uint8_t* hay, const int64_t* off, int64_t n_hay,
                   const int32_t* init_state, const int32_t* index_base,
                   acx_result_t** result);
/* The same for a caller that does not continue the walk (Automaton.iter_batch): no carried-in states, no final states.
 * ACX_SCAN_LONG then takes its position-parallel form (acx_blob_long_trie above) where that applies, and an
 * ACX_SCAN_ALL image never builds its dense table.  acx_result_fetch_host returns final_state = NULL. */
int  acx_scan_host_nofinal(acx_image_t* img, int mode, const uint8_t* hay, const int64_t* off, int64_t n_hay,
                           const int32_t* index_base, acx_result_t** result);
/* acx_scan_host / acx_scan_host_ctx scan a batch larger than this many bytes in groups of haystacks, one launch each, and
 * assemble one result (default, and 0: 4 GiB — what one launch stages).  Process-wide. */
void acx_set_host_group_bytes(int64_t bytes);
/* ACX_SCAN_ALL over streams: haystack h is scanned with the context bytes ctx[ctx_off[h] .. ctx_off[h+1]) in front of it
 * (dev_skip above); the library stages context and chunk side by side, the caller copies nothing.  ctx == NULL: as
 * acx_scan_host without states.  No final states are computed (the next chunk's context is the caller's: the last
 * longest_word - 1 bytes of context + chunk). */
int  acx_scan_host_ctx(acx_image_t* img, const uint8_t* hay, const int64_t* off, int64_t n_hay,
                       const uint8_t* ctx, const int64_t* ctx_off, const int32_t* index_base,
                       int32_t flags /* ACX_SCAN_SKIP_WS or 0: iter(..., ignore_white_space=True) */,
                       acx_result_t** result);

/* ------------------------------------------------------------------------------------
 * 4b. The walk over the HOST trie (acx_hostwalk.cpp) — BASELINE.json's config 1 ("iter() over a 1 KB haystack on CPU:
 *     plumbing, no GPU"), a process without a device, and haystacks that do not pay a launch.  The reference walks its
 *     pointer trie at 18-26 us per KB; a GPU scan of 1 KB is ~90 us of launch and copies.  Same semantics
 *     as acx_scan_host_ctx (ACX_SCAN_ALL: ctx, ACX_SCAN_SKIP_WS) and acx_scan_host (ACX_SCAN_LONG: states), on the arena
 *     trie itself: automaton_search_iter_next / automaton_build_output / ahocorasick_next
 *     (src/AutomatonSearchIter.c:243-300, :157-197; src/trie.c:177-194), automaton_search_iter_long_next
 *     (src/AutomatonSearchIterLong.c:89-153).  No image is needed.  At most ACX_HOSTWALK_MAX_BYTES of haystack per call:
 *     batches belong to the GPU.
 *       init_node / final states (ACX_SCAN_LONG): a state of the host walk is  -(arena node) - 1  (0 = root), so that it
 *       is never mistaken for a state id of a device image (those are > 0); a stream that carries a negative state
 *       continues on the host walk, one that carries a positive one on the device.
 *     The result is read with acx_result_fetch_host / acx_result_num_matches (it has no device side).
 *     Which scans come here is the HOST SIDE's decision, by acx_host_walk_applies(total bytes):
 *       limit < 0   never (a process without a device then fails with ACX_E_NODEVICE as before)
 *       limit >= 0  when the process has no device, or the scan is of at most `limit` bytes (default 2048: below the
 *                   measured crossover of a GPU scan)
 *     acx_host_walk_calls() counts the walks of this process: the GPU test-suite switches the walk off
 *     (tests/conftest.py: limit -1; the drop-in module also reads ACX_HOST_WALK_BYTES at import) and asserts the counter
 *     did not move — no GPU parity result can come from the host.
 * ---------------------------------------------------------------------------------- */
#define ACX_HOSTWALK_MAX_BYTES       (1ll << 20)
#define ACX_HOST_WALK_DEFAULT_BYTES  2048
int     acx_trie_scan_host(const acx_trie_t* t, int mode, const uint8_t* hay, const int64_t* off, int64_t n_hay,
                           const uint8_t* ctx, const int64_t* ctx_off, const int32_t* init_node, const int32_t* index_base,
                           int32_t flags, int want_final, acx_result_t** result);
void    acx_set_host_walk_bytes(int64_t limit);
int64_t acx_host_walk_bytes(void);
int     acx_host_walk_applies(int64_t total_bytes);
int64_t acx_host_walk_calls(void);

/* device helpers used by bindings that have no HIP runtime of their own */
int  acx_device_count(int* n);
int  acx_device_set(int dev);
int  acx_dev_malloc(void** p, size_t nbytes);
void acx_dev_free(void* p);
int  acx_memcpy_h2d(void* dst_dev, const void* src_host, size_t nbytes);
int  acx_memcpy_d2h(void* dst_host, const void* src_dev, size_t nbytes);
int  acx_device_sync(void);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif
#endif /* ACX_H_INCLUDED */
