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

#endif
