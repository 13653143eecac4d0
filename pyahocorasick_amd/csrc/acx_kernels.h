// acx_kernels.h — launch interface between the C-ABI (acx_capi.hip) and the HIP kernels
// (acx_kernels.hip).  Internal to libacx.
#ifndef ACX_KERNELS_H_INCLUDED
#define ACX_KERNELS_H_INCLUDED

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "acx_ppm_layout.h"

struct acx_walk_args {
    // input batch
    const uint8_t* hay;        // concatenated haystacks (device)
    int64_t        hay_cap;    // readable bytes at `hay`
    const int64_t* off;        // int64[n_hay+1] or nullptr (fixed stride)
    int64_t        stride;
    int64_t        n_hay;
    const int32_t* init_state; // nullable
    const int32_t* index_base; // nullable
    // image
    const uint8_t*  cls;       // uint8[256]
    const uint32_t* table;     // uint32[n_states*K]
    const uint32_t* out_off;   // uint32[n_states+1]
    uint32_t        row_bytes; // K*4
    uint32_t        state_bits; // entry layout: 24 (narrow) or 27 (wide), include/acx_blob.h
    uint32_t        n_states;   // init_state entries at or beyond it are taken as the root
    // outputs
    int32_t* counts;           // matches per haystack
    int32_t* nev;              // events per haystack
    uint2*   events;           // staging: one slot per haystack byte; haystack h starts at slot off[h]
    int32_t* final_state;      // nullable
    // iter_long: matches do not overlap, so a haystack of len bytes reports at most len / shortest_key + 1 of them: its
    // events start at slot (byte offset >> ev_shift) + h instead of one slot per byte (0: one slot per byte)
    int32_t  ev_shift;
};

// One unit of work of the chunked scan (long haystacks): walk bytes [start, start+len) of the
// concatenated buffer from the root (or from init_state when the chunk begins its haystack),
// report only matches ending at or after byte start+emit.  emit = the left halo,
// longest_word-1 bytes: every match ending at i depends only on the longest_word bytes
// up to i, so chunks are independent (SURVEY.md §5 "chunked streaming").
struct acx_chunk_desc {
    int64_t start;    // global byte offset where the walk starts
    int32_t emit;     // first reported position, relative to start
    int32_t len;      // bytes to walk
    int32_t idx0;     // end_index reported for byte `start` (position in haystack + index_base)
    int32_t hay;      // owning haystack
    int32_t flags;    // bit0: chunk starts at the haystack start, bit1: chunk ends the haystack
    int32_t pad;
};

struct acx_chunk_args {
    const int64_t* off; int64_t stride; int64_t n_hay;
    const int32_t* index_base;     // nullable
    int32_t chunk_bytes;           // CH
    int32_t halo;                  // longest_word - 1
    int32_t* nck;                  // int32[n_hay]     chunks per haystack
    int64_t* ck_first;             // int64[n_hay+1]   first chunk of each haystack (exclusive scan of nck)
    acx_chunk_desc* ck;            // [n_chunks]
};

struct acx_expand_args {
    const int64_t* off;  int64_t stride;  int64_t n_hay;
    const int32_t* nev;
    const uint2*   events;
    const int64_t* match_off;  // int64[n_hay+1]
    const uint32_t* out_off;
    const int32_t*  out_val;
    const int32_t*  first_val; // int32[n_states]: out_val[out_off[s]]
    uint2*   matches;          // acx_match_t[capacity]
    int64_t  capacity;
    int32_t  long_mode;        // 1: every event is exactly one match (iter_long)
    int32_t  ev_shift;         // as in acx_walk_args
    uint32_t state_bits;       // entry layout of the image
    // chunked scans: items are chunks; event base = ck[c].start + ck[c].emit; the number of
    // items lives in device memory (n_items_dev) and n_hay is only an upper bound for the grid
    const acx_chunk_desc* ck;  // nullable
    const int64_t* n_items_dev;
};

// variant: 0 = default.  See acx_kernels.hip for the list.
hipError_t acx_launch_walk_all(const acx_walk_args& a, bool has_escape, int variant, hipStream_t s);
hipError_t acx_launch_walk_long(const acx_walk_args& a, int variant, hipStream_t s);
// exclusive prefix sum: counts int32[n] -> match_off int64[n+1]; partials = int64[ceil(n/4096)+1] scratch
hipError_t acx_launch_scan(const int32_t* counts, int64_t n, int64_t* match_off, int64_t* partials, hipStream_t s);
hipError_t acx_launch_expand(const acx_expand_args& a, int variant, hipStream_t s);
// chunked scan (ACX_SCAN_ALL on long haystacks)
hipError_t acx_launch_chunk_count(const acx_chunk_args& c, hipStream_t s);          // fills nck
hipError_t acx_launch_chunk_fill(const acx_chunk_args& c, int64_t n_chunks_bound, hipStream_t s);   // needs ck_first
hipError_t acx_launch_walk_chunks(const acx_walk_args& a, const acx_chunk_desc* ck, const int64_t* n_chunks_dev,
                                  int64_t n_chunks_bound, bool has_escape, hipStream_t s);
// walk with the implicit top-of-trie in LDS (narrow images with itop_depth > 0, no init_state);
// items are haystacks (ck == nullptr) or chunks
hipError_t acx_launch_walk_itop(const acx_walk_args& a, const acx_chunk_desc* ck, const int64_t* n_chunks_dev,
                                int64_t n_items_bound, bool has_escape, const uint32_t* itop_lds, uint32_t itop_words,
                                const uint32_t* itop_entry, const uint32_t* itop_ebits, const void* itop_cells,
                                const uint32_t* tflags, uint32_t cell_bytes, uint32_t itop_flags, int tune, hipStream_t s);
// per-haystack match offsets from per-chunk ones: match_off[h] = ck_match_off[ck_first[h]]
hipError_t acx_launch_hay_offsets(const int64_t* ck_first, const int64_t* ck_match_off, int64_t n_hay,
                                  int64_t* match_off, hipStream_t s);
int64_t acx_scan_num_partials(int64_t n);

// ---- position-parallel scan (acx_ppm_kernels.hip; image: include/acx_blob.h "ppm") ----------------
// Work items are TILES of ACX_PPM_TILE end positions: tile i of the concatenated buffer (fixed-stride
// batches: ck == nullptr) or one chunk of one haystack (ck != nullptr, chunks of at most
// ACX_PPM_TILE bytes; their number lives in device memory).
struct acx_ppm_args {
    const uint8_t* hay; int64_t hay_cap;
    int64_t stride; int64_t n_hay; uint64_t stride_magic;    // ceil(2^64 / stride); 0 for stride 1
    const int32_t* index_base;                               // nullable (fixed-stride batches)
    const int32_t* skip;                                     // nullable: context bytes in front of every haystack (acx_scan_params.dev_skip)
    const acx_chunk_desc* ck; const int64_t* n_items_dev;
    int64_t n_items;                                         // tiles of a fixed-stride batch
    // image
    const uint8_t* cls; const uint32_t* g; const uint32_t* cells; const int32_t* top_val;
    const uint32_t* kids; const uint32_t* chains; uint32_t n_branch;
    const uint32_t* hot;     // k_ppm_stream: 8-byte hot cells
    const uint32_t* gh;      // k_ppm_stream, filter in global memory: its hashed copy for LDS (include/acx_blob.h ACX_PPM_GH_*; nullptr: ask G for every position)
    uint32_t hot12;          // k_ppm_stream4: hot4's cells are 12 bytes, the id in the third word (no cid[])
    const uint32_t* hot4; const uint32_t* cid;   // k_ppm_stream4: its hot cells and the ids of the depth-C nodes (include/acx_blob.h; nullptr: absent or not wanted)
    const uint8_t* symtab;   // byte -> symbol, 0xFF = a byte of no key
    uint32_t sym_arith, sym_lut;   // K == 4: symbol = (byte >> (sym_arith - 1)) & 3, sym_lut = the four key bytes (0: table only)
    uint32_t K, sym_bits, pow2, C, F, g_words, has_other, longest, min_len;
    uint32_t top_base[ACX_PPM_MAX_C + 2];
    acx_ppm_lds lds;
    uint32_t fast;           // 1: k_ppm_stream (fixed stride >= 4, aligned buffer, bit-field codes, halo_pos <= 256)
#ifdef ACX_PPM_DEV
    uint32_t dbg;            // development builds only (k_ppm_stream's phase switches: 2 = no record writes, 4 = no rounds, 8 = no queue, 16 = no filter)
#endif
    uint32_t nsub;           // k_ppm_stream: sub-steps of 256 positions per tile (4 or 8)
    uint32_t share_a, share_b;   // k_ppm_stream4: the unequal runs of a block's waves (acx_ppm_slot_first_tile; 0, 0: equal)
    uint32_t m24;            // k_ppm_stream: ceil(2^23 / stride) for strides below 2048 (a 24-bit multiply divides), else 0
    const int64_t* off; const int64_t* first_h;    // k_ppm_stream on an offsets batch: offsets, first haystack at or after every tile
    const uint32_t* start_bits;                    // k_ppm_stream4 on an offsets batch: bit p = a haystack starts at byte p (two tiles of zero words behind the last)
    uint32_t g_global;       // the filter bitmap is read from global memory (not copied to LDS)
    const uint8_t* deep_base; uint32_t row_off, single_off;   // k_ppm_stream: rows and singles as 32-bit offsets from one base
    uint32_t* wave_desc;     // k_ppm_stream: per wave {records, grants, 16 x base, 16 x count}
    uint32_t halo_pos;       // k_ppm_stream: staged halo positions (multiple of 32 / sym_bits and of 4, >= longest - 1)
    // outputs
    int32_t*  counts;        // matches per tile
    uint32_t* scr_off;       // where the tile's records start in `scratch` (0xFFFFFFFF: none)
    uint2*    scratch;       // record pool: 8 sub-pools of pool_records each, one bump pointer per sub-pool
    unsigned long long* heads;
    uint32_t  n_pools;       // min(8, blocks): block b bumps heads[b % n_pools]
    const uint32_t* g2;      // k_ppm_stream: second-level filter bitmap (global), asked about F2 symbols; nullptr: none
    uint32_t  F2;
    uint32_t  reserve_cus;   // k_ppm_stream: CUs the grid leaves free (asynchronous scans: the gather of the previous batch runs there)
    uint64_t  pool_records;
    int32_t*  overflow;      // set when a sub-pool ran out: the host grows the pool and scans again
    unsigned long long* phase_out;   // development builds: 8 clock sums over all waves (stage+filter, push, -, fetch, tops, deep, place+records, rest)
    int32_t*  short_hay;     // k_ppm_stream on an offsets batch: set when a tile holds more haystack starts than min_hay_len >= 8 allows
    uint32_t* block_sum;     // fixed-stride stream scans: += the records of every wave of block b (zero before the scan); NULL: not kept
    int32_t*  hay_local;     // fixed-stride batches: tile-local record offset of every haystack start
};
struct acx_ppm_compact_args {
    const int32_t* counts; const uint32_t* scr_off; const uint2* scratch;
    const int64_t* item_off;       // exclusive prefix sum of counts, [n_items + 1]
    int64_t n_items; const int64_t* n_items_dev;
    uint2* matches; int64_t capacity;
    // fixed-stride batches: match_off[h] = item_off[h * stride / TILE] + hay_local[h]
    const int32_t* hay_local; int64_t* match_off; int64_t n_hay; int64_t stride;
};
#define ACX_PPM_MAX_BLOCKS 1024     // blocks of one k_ppm_stream launch at most (one or two per CU)
struct acx_ppm_gather_args {       // k_ppm_stream results -> final place
    const uint32_t* wave_desc; const int64_t* wave_off; int64_t n_waves;
    // k_ppm_gather_pos: the block sums of this scan, those of the result's next scan (zeroed here), the control words
    // (flags read, then zeroed here) and the host's pinned words (total, overflow, short_hay)
    const uint32_t* block_sum; uint32_t* block_sum_next; unsigned long long* ctl; long long* host_words;
    const uint2* scratch; uint2* matches; int64_t capacity;
    const int32_t* hay_local; int64_t* match_off; int64_t n_hay; int64_t stride;
    const int64_t* off;            // offsets batch (else nullptr: fixed stride — records carry global positions)
    uint64_t used_words;           // (their number: a multiple of 4)
    uint32_t* used_bits;           // that case: the scan's start bitmap — every wave's block here zeroes the words of its run of tiles (the result's next scan finds it clean: no memset in front of its scan kernel)
    int32_t pos_records;           // 1: an offsets batch scanned by k_ppm_stream4 (records carry global positions all the same) — k_ppm_gather_pos<true> instead of k_ppm_wave_scan + k_ppm_gather
    int64_t tile_pos, tpw;         // positions per tile, tiles per wave
    uint64_t stride_magic;         // fixed stride: ceil(2^64 / stride) (0 for stride 1)
    const int32_t* index_base;     // fixed stride: added to every index of haystack h (nullable)
    const int32_t* skip;           // fixed stride: context bytes of haystack h, taken off every index (nullable)
    int64_t off_base;              // k_ppm_gather_pos: added to every offset written (the records of the groups in front: acx_scan_host's pipeline)
    uint32_t share_a, share_b;     // k_ppm_gather_pos: the runs of the scan kernel's waves (acx_ppm_slot_first_tile)
};
#define ACX_PPM_DESC_WORDS 40
hipError_t acx_launch_ppm_first_h(const int64_t* off, int64_t n_hay, int64_t n_tiles, int64_t tile_pos, int64_t* first_h, hipStream_t s);
hipError_t acx_launch_ppm_gather(const uint32_t* wave_desc, int64_t n_waves, int64_t* wave_off, const acx_ppm_gather_args& c, hipStream_t s);
hipError_t acx_launch_ppm_scan(const acx_ppm_args& a, int64_t n_items_bound, hipStream_t s);
// k_ppm_stream4 (acx_ppm_stream4.hip): fixed-stride batches over four-letter alphabets; eligible() says whether it takes the scan
bool acx_ppm_stream4_eligible(const acx_ppm_args& a);
bool acx_ppm_stream4_offs_ok(const acx_ppm_args& a);      // an offsets batch it would take once start_bits exists
// start_bits for k_ppm_stream4's offsets form: bit p of the bitmap = a haystack starts at byte p (n_words words; zero_first: zeroed here first —
// otherwise they ARE zero: k_ppm_gather_pos<true> clears what its scan used, acx_ppm_gather_args.used_bits)
hipError_t acx_launch_ppm_start_bits(const int64_t* off, int64_t n_hay, uint32_t* bits, size_t n_words, bool zero_first, hipStream_t s);
hipError_t acx_launch_ppm_stream4(const acx_ppm_args& a, int64_t blocks, hipStream_t s);
hipError_t acx_launch_ppm_compact(const acx_ppm_compact_args& c, int64_t n_items_bound, hipStream_t s);
int64_t acx_ppm_grid_blocks(const acx_ppm_lds& lds, int64_t n_items_bound, uint32_t reserve_cus = 0);   // blocks of a k_ppm_scan launch
// final_state of an ACX_SCAN_ALL scan when the matches came from the position-parallel kernels: the
// state after a haystack = the state after its last longest_word bytes walked from the root
hipError_t acx_launch_tail_state(const acx_walk_args& a, int32_t longest, hipStream_t s);
int acx_num_cus();
// dev_skip for the kernel families that do not know it (serial walks, k_ppm_scan): the records of the context — a prefix
// of every haystack's records, they are sorted by end_index — are dropped and the rest rebased.  kept[h] and new offsets
// come from k_skip_count + scan, then k_skip_move copies into `dst`.
hipError_t acx_launch_skip_count(const int64_t* match_off, const uint2* matches, const int32_t* skip, const int32_t* index_base,
                                 int64_t n_hay, int32_t* kept, hipStream_t s);
hipError_t acx_launch_skip_move(const int64_t* match_off, const uint2* matches, const int32_t* skip, const int32_t* kept,
                                const int64_t* new_off, int64_t n_hay, uint2* dst, hipStream_t s);
// ACX_SCAN_SKIP_WS (acx_ws.hip): white space out of the haystack buffer before the scan (tile counts -> scan ->
// compacted bytes + the original position of each -> offsets and context lengths of the compacted batch), and the end
// indices of the records back to original positions afterwards
struct acx_ws_remap_args {
    uint2* matches; const int64_t* match_off; int64_t n_hay; int64_t total;
    const int64_t* off; int64_t stride; const int32_t* skip; const int32_t* index_base;     // the batch as the caller gave it
    const int64_t* c_off; const int32_t* c_skip; const uint32_t* map;                       // the compacted batch
};
int64_t acx_ws_num_tiles(int64_t total);
hipError_t acx_launch_ws_count(const uint8_t* hay, int64_t total, int32_t* tile_count, hipStream_t s);
hipError_t acx_launch_ws_move(const uint8_t* hay, int64_t total, const int64_t* tile_off, uint8_t* out_hay, uint32_t* out_map, hipStream_t s);
hipError_t acx_launch_ws_offsets(const int64_t* off, int64_t stride, int64_t n_hay, const int32_t* skip, const uint32_t* map,
                                 const int64_t* n_kept, int64_t* c_off, int32_t* c_skip, hipStream_t s);
hipError_t acx_launch_ws_remap(const acx_ws_remap_args& a, hipStream_t s);
// build the dense transition table in HBM from the sparse form (acx_build.hip)
hipError_t acx_launch_build_table(uint32_t* table, const int32_t* fail, const uint32_t* edge_off, const uint8_t* edge_cls,
                                  const uint32_t* edge_dst, const uint32_t* tflags, const uint32_t* lvl_first_host,
                                  uint32_t n_levels, uint32_t K, uint32_t state_bits, hipStream_t s);

#endif
