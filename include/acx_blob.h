/*
 * acx_blob.h — layout of the flat automaton image ("blob").
 *
 * One contiguous little-endian buffer; every section starts on a 256-byte boundary and
 * is addressed by a byte offset from the start of the blob, so the blob can be copied,
 * broadcast (one RCCL call) or written to disk verbatim.  It replaces, for scanning,
 * the pointer graph of the reference (TrieNode/Pair, src/trienode.h:19-42).
 *
 * States are renumbered in BFS order (root = 0, shallow states first) so the rows that
 * random text visits most are contiguous and stay cache resident.
 *
 * Transition table: uint32 table[n_states][n_classes], fail links pre-resolved
 * (table[s][c] is ahocorasick_next(s, c), src/trie.c:177-194, for every s and c), with
 * per-TARGET facts packed into the spare bits so the scan needs ONE 4-byte gather per
 * input byte and nothing else unless a match is reported:
 *
 *   bits  0..23  next state t                                   (ACX_ENTRY_STATE_MASK)
 *   bit   24     EDGE     s has a real trie edge labelled c     (iter_long needs it:
 *                         src/AutomatonSearchIterLong.c:117 uses trienode_get_next only)
 *   bit   25     EOW      t->eow                                (…IterLong.c:119)
 *   bit   26     FAILEOW  t->fail != root && t->fail->eow       (…IterLong.c:123)
 *   bits 27..31  CNT      min(|out(t)|, 31), |out(t)| = number of eow nodes on the chain
 *                         t, fail(t), fail(fail(t)), …  (what automaton_build_output walks,
 *                         src/AutomatonSearchIter.c:161-168).  31 = escape: read out_off.
 *
 * Byte classes: bytes that occur in no key can never leave the root, so all of them
 * share class 0; every byte that occurs in some key gets its own class.  DNA keys give
 * 5 classes (20-byte rows) instead of 256 (1 KiB rows).
 */
#ifndef ACX_BLOB_H_INCLUDED
#define ACX_BLOB_H_INCLUDED

#include <stdint.h>

/*
 * Implicit top-of-trie ("itop"), for key alphabets of at most 16 symbols.  The top D levels of
 * the trie are dense (every k-gram over the used bytes tends to exist), so a state at depth
 * d <= D is representable as the pair (d, code) where code = its last d symbols, b bits each
 * (symbol = class - 1): no row, no id.  The walk keeps the rolling history `hist` of the last D
 * symbols and reads, from a table that is resident in LDS, where that history puts it:
 *
 *   ND4[hist]  4 bits:  bits 0..1  delta = D - depth of the longest k-gram node that is a
 *                                  suffix of the history (0, 1, 2; 3 = deeper fall: probe E)
 *                       bits 2..3  output class of that node: 0 none, 1 exactly one output (it is
 *                                  reported as pseudo state n_states + x, see first_val), 2 more
 *                                  (its packed entry is itop_entry[x])
 *
 * ND4 is exact once D symbols have been seen since the last reset (haystack start or a byte
 * that occurs in no key); before that the slow path probes the existence bitmap E level by
 * level.  ND4 only speaks about nodes of depth <= D (the last D symbols of a deeper state need
 * not be a node), so:
 *   - a lane at depth < D steps with ND4 alone;
 *   - a lane at depth == D first asks whether its node has a child on the new symbol: one 4- or
 *     8-byte CELL per level-D code (itop_cells, global, code-indexed: 1 MB for DNA instead of
 *     the 1.7 MB of level-D rows, and the only thing 2/3 of all table accesses need).  Children
 *     of a level-D node are numbered consecutively in symbol order (levels 1..D+1 are numbered in
 *     code order), so child id = first_child + popcount(mask below the symbol).  No child: ND4.
 *   - a lane at depth > D holds an explicit state and gathers its table entry like the plain
 *     walk; when the target is not deeper than D it drops back to (depth from ND4, hist).
 *
 *   cell, 4 bytes (<= 4 symbols):  first_child[0..23] | child mask[24..27] | child-has-outputs[28..31]
 *   cell, 8 bytes (<= 16 symbols): first_child (low word); child mask[0..15] | child-has-outputs[16..31] (high word)
 *   (a child with outputs takes its packed entry from tflags[id], which carries the id)
 *
 * Levels 0..D share ONE index space: node (d, code) is  x = (1 << b*d) | code  ("sentinel bit"
 * above the code; the root is x = 1).  E (global) and itop_entry (global) are indexed by x.
 *
 * LDS image (uint32 words): [0]=b [1]=D [2]=first id of level D+1 (ids from here on are "deep")
 * [3]=cell bytes (4 or 8) [5]=1 if class 0 is "other" [6]=total words [7]=mask(D) = 2^(bD)-1
 * [8]=ND4 word offset [11]=shift up to which levels are complete (every k-gram exists: no probe
 * needed) [12]=shift of the shallowest level that has outputs;  then ND4 (2^(bD) x 4 bits).  b*D >= 5.
 */
#define ACX_ITOP_MAX_LEVELS   15
#define ACX_ITOP_HDR_WORDS    16
#define ACX_ITOP_FLAG_NOESC   1u
#define ACX_ITOP_FLAG_TFLAGS_ID 2u

#define ACX_BLOB_MAGIC        0x31424F4C42584341ull   /* "ACXBLOB1" */
#define ACX_BLOB_VERSION      5u          /* 5: "PPM6" (hot4: the path below a one-child cell); 4: "PPM5" (deep records of up to 48 / sym_bits symbols); 3: the "PPM4" section (PPM3 + hot4 / cid for four-letter alphabets); 2: the checksum below, "PPM3" */
#define ACX_BLOB_HEADER_BYTES 256u
#define ACX_BLOB_ALIGN        256u

/* Entry layout, parameterised by the number of state bits SB.
 *   SB = 24 (narrow): < 2^24 states and a table < 4 GiB: v_mad_u32_u24 + 32-bit offset; CNT has 5 bits
 *   SB = 27 (wide)  : < 2^27 states, any table size: 64-bit addressing; CNT has 2 bits
 * (bit positions in the comment above are for SB = 24) */
#define ACX_STATE_BITS_NARROW 24
#define ACX_STATE_BITS_WIDE   27
#define ACX_ENTRY_STATE_MASK(SB)  ((1u << (SB)) - 1u)
#define ACX_ENTRY_EDGE(SB)        (1u << (SB))
#define ACX_ENTRY_EOW(SB)         (1u << ((SB) + 1))
#define ACX_ENTRY_FAILEOW(SB)     (1u << ((SB) + 2))
#define ACX_ENTRY_CNT_SHIFT(SB)   ((SB) + 3)
#define ACX_ENTRY_CNT_ESCAPE(SB)  ((1u << (32 - ((SB) + 3))) - 1u)

typedef struct acx_blob_header {
    uint64_t magic;
    uint32_t version;
    uint32_t header_bytes;
    uint64_t total_bytes;
    uint32_t n_states;
    uint32_t n_classes;      /* K: entries per table row */
    uint32_t n_keys;
    uint32_t longest_word;   /* halo size for chunked scans = longest_word - 1 */
    uint32_t max_out_count;  /* max |out(t)| over all states */
    uint32_t has_escape;     /* 1 if some state has |out(t)| >= ACX_ENTRY_CNT_ESCAPE(SB) */
    uint64_t n_out;          /* total entries of out_val */
    uint64_t trie_version;   /* acx_trie_version() at flatten time */
    /* section offsets (bytes from blob start) */
    uint64_t off_cls;        /* uint8  [256]            byte -> class                    */
    uint64_t off_table;      /* uint32 [n_states * K]   packed entries (see above)       */
    uint64_t off_fail;       /* int32  [n_states]       BFS id of fail(s); root: -1      */
    uint64_t off_node_val;   /* int32  [n_states]       low 32 bits of s's value if eow  */
    uint64_t off_node_flags; /* uint8  [n_states]       bit0 = eow                       */
    uint64_t off_out_off;    /* uint32 [n_states + 1]   CSR offsets into out_val         */
    uint64_t off_out_val;    /* int32  [n_out]          values in fail-chain order       */
    uint64_t fnv1a64;        /* checksum of bytes [header_bytes, total_bytes): FNV-1a's xor-multiply step (offset basis
                                0xcbf29ce484222325 + k, prime 0x100000001b3) applied to little-endian 8-byte WORDS in 8
                                interleaved lanes k = 0..7 (word i of every 64-byte block goes to lane i), then plain
                                byte-wise FNV-1a over the 64 bytes of the lanes (lane 0 first, low byte first) and over
                                the tail bytes that do not fill a 64-byte block (acx_fnv1a64, acx_trie.cpp) */
    uint64_t off_first_val;  /* int32  [n_states]       out_val[out_off[s]] (first output of s:
                                the value iter_long reports, and the only one when CNT == 1) */
    uint32_t state_bits;     /* SB of the entry layout: 24 or 27                         */
    uint32_t itop_depth;     /* D of the implicit top-of-trie (0 = absent), see above    */
    uint64_t off_itop_lds;   /* uint32 [itop_lds_bytes/4]  LDS image: levels, bitmaps, ranks */
    uint64_t off_itop_entry; /* uint32 [sum over levels 1..D of 2^(b*d)]  entry of node (d, code) */
    uint32_t itop_lds_bytes;
    uint32_t itop_bits;      /* b: bits per symbol                                       */
    /* Sparse form of the transitions (always present): enough to (re)build `table` on the
     * device — row(s) = row(fail(s)) with EDGE cleared, then s's own edges — level by level.
     * When table_in_blob == 0 the blob carries NO table section (off_table = 0): the image is
     * ~K x smaller on the wire (one RCCL broadcast of 1 GB instead of 35 GB for config 4) and
     * acx_image_upload/adopt build the table in HBM with k_build_rows_* (SURVEY §8f N2). */
    uint64_t off_edge_off;   /* uint32 [n_states + 1]   CSR offsets of the trie edges of s   */
    uint64_t off_edge_cls;   /* uint8  [n_edges]        class of the edge label              */
    uint64_t off_edge_dst;   /* uint32 [n_edges]        child state                          */
    uint64_t off_tflags;     /* uint32 [n_states]       per-target bits of an entry (EOW, FAILEOW, CNT) | s */
    uint64_t off_lvl_first;  /* uint32 [n_levels + 1]   first state id of each depth         */
    uint32_t n_levels;       /* max depth + 1                                                */
    uint32_t n_edges;
    uint32_t table_in_blob;  /* 1: `table` section present; 0: build it on the device        */
    uint32_t itop_cell_bytes; /* 4 or 8 (0 without itop)                                      */
    uint64_t off_itop_ebits; /* uint32 [2^(bD+1)/32]  existence bitmap E, sentinel-indexed (global; slow path) */
    uint64_t off_itop_cells; /* uint32|uint64 [2^(bD)]  child cell of the level-D node with that code   */
    uint32_t itop_flags;     /* bit 0: the depth field of ND4 never escapes once D symbols have been
                                seen (levels up to D-2 are complete): walk without the probe path;
                                bit 1: tflags[s] carries s in its state field (a whole packed entry):
                                blobs without it walk with the plain kernels */
    uint32_t reserved0;
    uint64_t off_ppm;        /* acx_ppm_header + its sections (position-parallel scan, below); 0 = absent */
} acx_blob_header;

/*
 * Position-parallel scan image ("ppm"), ACX_SCAN_ALL.
 *
 * automaton_search_iter_next (src/AutomatonSearchIter.c:243-300) reports, at every position e, the
 * keys that are suffixes of text[..e], longest first (the state, then its fail chain:
 * automaton_build_output, :157-197).  That set depends only on the last longest_word bytes, so it
 * can be computed for every position independently: lane e walks the trie of REVERSED keys
 * backwards from byte e (text[e], text[e-1], ...) and every end-of-word node it passes is one
 * match ending at e.  No serial state, no fail links, no output lists: haystack bytes are read
 * coalesced and staged in LDS as packed symbols, matches are placed with a wave prefix sum.
 *
 * symbol  = symtab[byte] (0xFF = "other": a byte no key contains; no match spans it); K symbols.  Symbols are numbered
 *           in byte order, except for alphabets of exactly four bytes that some shift tells apart ((byte >> s) & 3 is
 *           distinct: ACGT with s = 1): there symbol = (byte >> s) & 3, so that the stream kernel turns four bytes into
 *           four symbols with a shift and a mask instead of four table lookups (sym_arith, sym_lut).
 * code_d  = Horner value of the d newest symbols, newest first, radix K:
 *           code_d = code_{d-1} * K + sym(e - d + 1)           (code_0 = 0)
 * Levels 1..C of the reversed trie are direct-indexed by code (K^C <= 2^18):
 *   G    filter bitmap over code_F (F = C + 1 when K^(C+1) bits fit LDS, else C; g_global = 1: F = C + 1 and the
 *        bitmap is read from global memory — 8-bit symbols only): "look at the cell".  Set iff the depth-F node
 *        exists (F = C + 1) or the cell of code_C is non-empty.
 *   cell[code_C] (global, 32 bytes = 8 words):
 *       [0]  eowmask: bit d-1 set iff the d newest symbols are a key (d <= C)
 *       [1]  deep id of the depth-C node (0: absent or childless)
 *       [2]  K <= 4 only: children of that node: mask[0..3] | child is a key[4..7] |
 *            grandchild (s1*4+s2) exists [8..23]; else 0
 *       [3..7] the values of the first five keys of eowmask, shortest first (= the order matches are produced in)
 *   hot[code_C] (global, 8 bytes: 2 MiB for 2^18 cells, resident in every XCD's 4 MiB L2 — the 32-byte cells are not):
 *       what the stream kernel asks first about a position that passed G; most positions need nothing else.
 *       [0]  bits 0..C-1  eowmask, as in the cell.
 *            sym_bits == 2 (C <= 12): bits 12..15 child s1 of the depth-C node is a key, bits 16..31 its grandchild
 *            (s1 * 4 + s2) exists — the walk goes deeper iff one of the two bits of the next two symbols is set;
 *            (bits 12..31) != 0 iff the node has children.  Wider symbols: bit 31 = the node has children.
 *       [1]  the node has children: its deep id (as cell[1]); else the value of the shallowest key of eowmask (else 0).
 *   hot4[code_C] (global, 8 bytes; sym_bits == 2 and C + 2 <= 16 only — off_hot4 = 0: absent) and cid[code_C] (4 bytes):
 *       the hot cell as k_ppm_stream4 (four-letter alphabets, fixed stride) reads it.  A position that ends a key — most
 *       candidates that are no false alarm — needs the VALUE of its shallowest key, not the id of the depth-C node:
 *       [0]  bits 0..C-1  eowmask; bits 16..31  go[s1 * 4 + s2]: the child s1 of the depth-C node is a key or its
 *            grandchild (s1, s2) exists — the walk goes deeper iff the bit of the next two symbols is set (a leaf is a
 *            key: bits 16..31 != 0 iff the node has children)
 *       [1]  eowmask != 0: the value of its shallowest key; else the deep id of the depth-C node (0: none)
 *       cid[code_C] = the deep id of the depth-C node (0: absent or childless): read by the walks that go deeper from a
 *       cell whose second word holds a value.
 *       off_hot4 != 0 with off_cid == 0 (ACX_FLATTEN_HOT12): the cells are 12 bytes and there is no cid section —
 *       [0] as above, [1] the value of the shallowest key of eowmask (0: none), [2] the deep id of the depth-C node (0: none).
 *       For dictionaries whose walks mostly start from cells that also end a key (the dictionaries of iter_long, acx_long.cpp):
 *       with 8-byte cells every such walker waits for cid[] before its first record — one more round trip in every pass.
 *   gh (optional, g_global images): a hashed copy of G for LDS, see ACX_PPM_GH_* below.
 *   G2[code_F2] (global, L2 resident, at most 2^25 bits; optional): the same question as G asked with F2 > F symbols,
 *       put to the positions that passed G before they become candidates: set iff the depth-F2 node exists or a key
 *       shorter than F2 ends here.  For alphabets whose filter passes many positions that end no key (text: a 32-bit
 *       window holds four 8-bit symbols, F2 = 5 takes one more from the symbol array).
 *   top_val[top_base[d] + code_d] (global): value of the key that is node (d, code_d), d <= C (the sixth key on).
 * Deeper: the walk stands on a node that has children and takes one 16-byte record per step
 *       { label, len | is_key << 8 | exists << 9 | more label << 16, value, next id }
 *   a node with two or more children owns a ROW of K records (section `kids`; its id is the index of the row's first
 *   record, row number x K, so that the device adds the symbol and never multiplies), indexed by the next
 *   symbol; a node with exactly one child owns a SINGLE record (section `chains`; its id carries bit 31).  A record
 *   consumes 1 + len symbols (row) or len symbols (single): it follows the unbranched, key-free path below its first
 *   edge for up to 48 / sym_bits symbols (label: the first 32 / sym_bits of them, first one in the top bits; the top half
 *   of the second word: the 16 / sym_bits behind those, the same way); `next` is the id of the node it ends on if that
 *   node has children.
 * All section offsets are relative to the start of the acx_ppm_header.
 */
#define ACX_PPM_MAGIC 0x364D5050u   /* "PPM6" */
#define ACX_PPM_MAX_C 20
/* gh: a filter that lives in global memory (g_global: 2^24 bits for three 8-bit symbols) costs one L2 request per position.
 * gh is a hashed copy of it that LDS can hold: bit acx_ppm_gh_index(code_F) is set for every code whose bit of G is set, so a
 * position whose bit of gh is clear is rejected without asking G (no false negatives); the others ask G as before.  Written
 * only when it rejects enough (the builder measures: at most 3 of 4 random codes pass). */
#ifndef ACX_PPM_GH_BITS               /* (development builds try other sizes: tools/build_variant.sh -DACX_PPM_GH_BITS=...) */
#define ACX_PPM_GH_BITS  (15u << 16)  /* 120 KiB: what the staging of 1024-position tiles of 8-bit symbols (halo up to 256) leaves of a CU's LDS */
#endif
#define ACX_PPM_GH_WORDS (ACX_PPM_GH_BITS / 32u)
#define ACX_PPM_GH_MUL   0x9E3779B1u
/* index of a code in gh: the high word of (code * MUL mod 2^32) * BITS */
#define ACX_PPM_GH_INDEX(code) ((uint32_t)(((uint64_t)(uint32_t)((uint32_t)(code) * ACX_PPM_GH_MUL) * ACX_PPM_GH_BITS) >> 32))
#define ACX_PPM_TILE  256           /* end positions per wave and tile */
typedef struct acx_ppm_header {
    uint32_t magic;
    uint32_t K;              /* symbols (bytes that occur in keys), 1..256 */
    uint32_t sym_bits;       /* bits per symbol in the LDS staging: 2, 4 or 8 */
    uint32_t pow2;           /* 1: K == 1 << sym_bits, codes are plain bit fields */
    uint32_t C, F;
    uint32_t g_global;       /* 1: G has more bits than LDS holds and is read from global memory (stream kernel only) */
    uint32_t F2;             /* second-level filter (global memory): symbols it is asked about (0: none) */
    uint32_t sym_arith;      /* K == 4 only: 1 + s when symbol(byte) = (byte >> s) & 3 for the four key bytes (0: table only) */
    uint32_t g_words, g2_words;
    uint32_t sym_lut;        /* sym_arith != 0: byte j = the key byte whose symbol is j (a byte b is "other" iff lut[(b >> s) & 3] != b) */
    uint32_t has_other;
    uint32_t longest;        /* longest key */
    uint32_t n_deep;         /* rows: deep ids K, 2K, .. n_deep * K */
    uint32_t n_top;          /* entries of top_val */
    uint32_t min_len;        /* shortest key */
    uint32_t n_chain;        /* single records: ids 0x80000000 | 1..n_chain */
    uint64_t total_bytes;    /* header + sections */
    uint64_t off_g, off_g2;
    uint64_t off_symtab;     /* uint8 [256]: byte -> symbol, 0xFF = a byte no key contains */
    uint64_t off_cells, off_top_val, off_kids /* rows */;
    uint64_t off_hot;        /* uint32 [2 * K^C]: the 8-byte hot cells (stream kernel) */
    uint32_t top_base[ACX_PPM_MAX_C + 2];
    uint64_t off_chains;     /* singles */
    uint64_t off_hot4;       /* uint32 [2 * K^C]: hot cells of k_ppm_stream4 (0: absent); [3 * K^C] when off_cid == 0 */
    uint64_t off_cid;        /* uint32 [K^C]: deep id of every depth-C node (with off_hot4; 0 with 12-byte cells) */
    uint64_t off_gh;         /* uint32 [ACX_PPM_GH_WORDS]: hashed copy of a global filter for LDS (g_global images; 0: absent) */
} acx_ppm_header;

#endif
