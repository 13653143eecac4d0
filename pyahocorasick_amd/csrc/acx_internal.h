// acx_internal.h — shared declarations inside libacx (not part of the C-ABI).
#ifndef ACX_INTERNAL_H_INCLUDED
#define ACX_INTERNAL_H_INCLUDED

#include "acx.h"
#include "acx_blob.h"

static_assert(sizeof(acx_blob_header) == ACX_BLOB_HEADER_BYTES, "acx_blob_header must be exactly 256 bytes");

extern "C" {
// records a thread-local message for acx_last_error() and returns `code`
int acx_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
uint64_t acx_fnv1a64(const uint8_t* p, size_t n);
int acx_blob_check_header(const acx_blob_header* h, size_t nbytes);
}


/* Tuning hooks (A/B switches, size caps of the builders, launch shapes) exist in builds made with -DACX_TUNING only.  The release
 * library reads ACX_HOST_THREADS (acx_trie_impl.h) and nothing else; the drop-in module reads ACX_HOST_WALK_BYTES at import
 * (include/acx.h §4b).  Layout options are flags of acx_flatten_ex. */
#include <stdlib.h>
static inline const char* acx_tune_env(const char* name) {
#ifdef ACX_TUNING
    return getenv(name);
#else
    (void)name;
    return (const char*)0;
#endif
}

#endif
