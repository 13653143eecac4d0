// acx_capi.hip — the C-ABI of libacx (include/acx.h): device image, batch scan driver,
// result objects.  Host trie + flattener live in acx_trie.cpp, kernels in acx_kernels.hip.
//
// There is NO CPU fallback anywhere in this file: without a usable HIP device every scan
// entry point fails with ACX_E_NODEVICE / ACX_E_HIP.
#include "acx_internal.h"
#include "acx_kernels.h"
#include "acx_long.h"

#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <thread>
#include <new>
#include <vector>
#include <algorithm>
#include <cmath>

// ------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

extern "C" int acx_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* acx_last_error(void) { return g_err; }
extern "C" int acx_abi_version(void) { return ACX_ABI_VERSION; }

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return acx_fail(_e == hipErrorOutOfMemory ? ACX_E_NOMEM                            \
                            : (_e == hipErrorNoDevice || _e == hipErrorInvalidDevice) ? ACX_E_NODEVICE \
                                                                                      : ACX_E_HIP, \
                            "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

// ------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------
extern "C" int acx_device_count(int* n) {
    if (!n) return acx_fail(ACX_E_INVAL, "acx_device_count: NULL");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *n = 0; return acx_fail(ACX_E_NODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *n = c;
    return ACX_OK;
}
extern "C" int acx_device_set(int dev) { HIP_TRY(hipSetDevice(dev)); return ACX_OK; }
extern "C" int acx_dev_malloc(void** p, size_t nbytes) {
    if (!p) return acx_fail(ACX_E_INVAL, "acx_dev_malloc: NULL");
    HIP_TRY(hipMalloc(p, nbytes ? nbytes : 1));
    return ACX_OK;
}
extern "C" void acx_dev_free(void* p) { if (p) (void)hipFree(p); }
extern "C" int acx_memcpy_h2d(void* d, const void* s, size_t n) { HIP_TRY(hipMemcpy(d, s, n, hipMemcpyHostToDevice)); return ACX_OK; }
extern "C" int acx_memcpy_d2h(void* d, const void* s, size_t n) { HIP_TRY(hipMemcpy(d, s, n, hipMemcpyDeviceToHost)); return ACX_OK; }
extern "C" int acx_device_sync(void) { HIP_TRY(hipDeviceSynchronize()); return ACX_OK; }

// ------------------------------------------------------------------------------------
// image
// ------------------------------------------------------------------------------------
struct acx_image {
    acx_blob_header h;
    uint8_t* dev = nullptr;     // base of the blob in device memory
    size_t nbytes = 0;
    bool owns = false;
    int device = 0;
    // resolved section pointers
    const uint8_t* cls = nullptr;
    const uint32_t* table = nullptr;
    const uint32_t* out_off = nullptr;
    const int32_t* out_val = nullptr;
    const int32_t* first_val = nullptr;
    const uint32_t* itop_lds = nullptr;     // nullptr when the image has no implicit top
    const uint32_t* itop_entry = nullptr;
    const uint32_t* itop_ebits = nullptr;
    const void* itop_cells = nullptr;
    const uint32_t* tflags = nullptr;
    uint32_t* built_table = nullptr;        // table built in HBM (blob without a table section); owned
    std::vector<uint32_t> lvl_host;         // level boundaries, kept for a table that is built on first use
    std::mutex table_mu;                    // the dense table is built on first use: one builder, and everybody who needs the table asks under it
    // position-parallel scan image (include/acx_blob.h "ppm"); ppm_g == nullptr: absent
    acx_ppm_header ppm;
    const uint32_t* ppm_g = nullptr;
    const uint32_t* ppm_cells = nullptr;
    const uint32_t* ppm_g2 = nullptr;          // second-level filter (nullptr: none)
    const int32_t*  ppm_top_val = nullptr;
    const uint32_t* ppm_kids = nullptr;
    const uint32_t* ppm_hot = nullptr;         // 8-byte hot cells (stream kernel)
    const uint8_t*  ppm_symtab = nullptr;      // byte -> symbol, 0xFF = a byte of no key
    const uint32_t* ppm_chains = nullptr;
    const uint32_t* ppm_hot4 = nullptr;        // hot cells and depth-C ids of k_ppm_stream4 (four-letter alphabets; nullptr: absent)
    const uint32_t* ppm_cid = nullptr;
    bool ppm_hot12 = false;                    // hot4's cells are 12 bytes, the id inline (ACX_FLATTEN_HOT12)
    const uint32_t* ppm_gh = nullptr;          // hashed copy of a global filter for LDS (nullptr: absent)
    // ACX_SCAN_LONG position-parallel (acx_long.cpp): a second image, over the dictionary D = E + FE + U of this one, built on
    // the first such scan; long_state 0: not tried yet, 1: there, -1: does not apply (the serial walk stays)
    std::mutex long_mu;
    int long_state = 0;
    acx_image* long_img = nullptr;
    int32_t* long_real = nullptr;              // device: what iter_long reports for dictionary entry i
    uint32_t long_longest = 0; int64_t long_n_real = 0; bool long_real_owned = true;
};

// The steady-state step of the itop walk uses 32-bit offsets: table and cells from the lower of
// the two, tflags and the implicit entries likewise.  An image where either pair spans 4 GiB or
// more walks with the plain kernels instead.
static void image_check_itop_reach(acx_image* img) {
    if (!img->itop_lds) return;
    auto span = [](const void* a, size_t na, const void* b, size_t nb) {
        const uint8_t* lo = (const uint8_t*)a < (const uint8_t*)b ? (const uint8_t*)a : (const uint8_t*)b;
        const uint8_t* ea = (const uint8_t*)a + na;
        const uint8_t* eb = (const uint8_t*)b + nb;
        return (size_t)((ea > eb ? ea : eb) - lo);
    };
    const size_t n = (size_t)img->h.n_states;
    const size_t tbytes = n * img->h.n_classes * 4 + 16;
    const size_t cbytes = (size_t)img->h.itop_cell_bytes << (img->h.itop_bits * img->h.itop_depth);
    const size_t ebytes = (size_t)8 << (img->h.itop_bits * img->h.itop_depth);      // 2^(bD+1) entries
    const size_t lim = (size_t)1 << 32;
    if (!(img->h.itop_flags & ACX_ITOP_FLAG_TFLAGS_ID)) { img->itop_lds = nullptr; return; }
    if (span(img->table, tbytes, img->itop_cells, cbytes) >= lim || span(img->tflags, n * 4, img->itop_entry, ebytes) >= lim)
        img->itop_lds = nullptr;
}

// resolve section pointers; when the blob carries no table, build it in HBM from the sparse
// form (acx_build.hip).  lvl_host = host copy of the level boundaries, or nullptr to fetch it.
static int image_build_table(acx_image* img, const uint32_t* lvl_host);

static int image_resolve(acx_image* img, const uint32_t* lvl_host) {
    img->cls = img->dev + img->h.off_cls;
    if (img->h.off_ppm) {
        HIP_TRY(hipMemcpy(&img->ppm, img->dev + img->h.off_ppm, sizeof img->ppm, hipMemcpyDeviceToHost));
        const acx_ppm_header& ph = img->ppm;
        // a corrupt or foreign section must not make the kernels read out of bounds: every sub-section lies inside it
        auto pw = [&](uint64_t base, uint32_t e) -> uint64_t { uint64_t p = 1; for (uint32_t i = 0; i < e; i++) { p *= base; if (p > ((uint64_t)1 << 40)) return (uint64_t)1 << 40; } return p; };
        const uint64_t tb = ph.total_bytes;
        auto inside = [&](uint64_t off, uint64_t len) { return off >= sizeof(acx_ppm_header) && off <= tb && len <= tb - off; };
        bool ok = ph.magic == ACX_PPM_MAGIC && img->h.off_ppm <= img->nbytes && tb <= img->nbytes - img->h.off_ppm && tb >= sizeof(acx_ppm_header) &&
                  ph.C >= 1 && ph.C <= 16 && ph.C <= ACX_PPM_MAX_C && ph.F >= ph.C && ph.F <= ph.C + 1 &&
                  (ph.sym_bits == 2 || ph.sym_bits == 4 || ph.sym_bits == 8) && ph.K >= 1 && ph.K <= (1u << ph.sym_bits) &&
                  (ph.sym_bits != 2 || ph.C <= 12) && ph.longest >= 1;
        if (ok) {
            const uint64_t nC = pw(ph.K, ph.C), nF = pw(ph.K, ph.F), nF2 = ph.F2 ? pw(ph.K, ph.F2) : 0;
            ok = nC <= ((uint64_t)1 << 22) && nF <= ((uint64_t)1 << 32) && ph.g_words == (uint32_t)((nF + 31) / 32) &&
                 (ph.F2 == 0 || (ph.F2 > ph.F && nF2 <= ((uint64_t)1 << 32) && ph.g2_words == (uint32_t)((nF2 + 31) / 32))) &&
                 inside(ph.off_g, (uint64_t)ph.g_words * 4) && (ph.F2 == 0 || inside(ph.off_g2, (uint64_t)ph.g2_words * 4)) &&
                 inside(ph.off_symtab, 256) && inside(ph.off_cells, nC * 32) && inside(ph.off_hot, nC * 8) &&
                 inside(ph.off_top_val, (uint64_t)ph.n_top * 4) && inside(ph.off_kids, ((uint64_t)ph.n_deep + 1) * ph.K * 16) &&
                 inside(ph.off_chains, ((uint64_t)ph.n_chain + 1) * 16) &&
                 (ph.off_gh == 0 || (ph.g_global && inside(ph.off_gh, (uint64_t)ACX_PPM_GH_WORDS * 4))) &&
                 ((ph.off_hot4 == 0 && ph.off_cid == 0) || (ph.sym_bits == 2 && inside(ph.off_hot4, (nC + 1) * 8) && inside(ph.off_cid, (nC + 1) * 4)) ||
                  (ph.sym_bits == 2 && ph.off_cid == 0 && inside(ph.off_hot4, (nC + 1) * 12)));      // (12-byte cells: no cid section)
            uint64_t tbase = 0;
            for (uint32_t d = 0; ok && d <= ph.C; d++) { ok = ph.top_base[d] == tbase; tbase += pw(ph.K, d); }
            ok = ok && ph.n_top == tbase && (ph.sym_arith == 0 || (ph.K == 4 && ph.sym_arith <= 7));
        }
        if (!ok) return acx_fail(ACX_E_FORMAT, "image: malformed ppm section");
        const uint8_t* sec = img->dev + img->h.off_ppm;
        img->ppm_g = (const uint32_t*)(sec + ph.off_g);
        img->ppm_cells = (const uint32_t*)(sec + ph.off_cells);
        img->ppm_g2 = ph.F2 ? (const uint32_t*)(sec + ph.off_g2) : nullptr;
        img->ppm_top_val = (const int32_t*)(sec + ph.off_top_val);
        img->ppm_kids = (const uint32_t*)(sec + ph.off_kids);
        img->ppm_hot = (const uint32_t*)(sec + ph.off_hot);
        img->ppm_symtab = sec + ph.off_symtab;
        img->ppm_chains = (const uint32_t*)(sec + ph.off_chains);
        img->ppm_hot4 = ph.off_hot4 ? (const uint32_t*)(sec + ph.off_hot4) : nullptr;
        img->ppm_cid = ph.off_cid ? (const uint32_t*)(sec + ph.off_cid) : nullptr;
        img->ppm_hot12 = ph.off_hot4 != 0 && ph.off_cid == 0;
        img->ppm_gh = ph.off_gh ? (const uint32_t*)(sec + ph.off_gh) : nullptr;
        if (!ph.g_global && acx_ppm_lds_layout(ph.g_words, ph.sym_bits, ph.longest).total_words * 4 > ACX_PPM_LDS_BYTES) img->ppm_g = nullptr;
    }
    img->out_off = (const uint32_t*)(img->dev + img->h.off_out_off);
    img->out_val = (const int32_t*)(img->dev + img->h.off_out_val);
    img->first_val = (const int32_t*)(img->dev + img->h.off_first_val);
    if (img->h.itop_depth > 0 && img->h.state_bits == ACX_STATE_BITS_NARROW) {
        img->itop_lds = (const uint32_t*)(img->dev + img->h.off_itop_lds);
        img->itop_entry = (const uint32_t*)(img->dev + img->h.off_itop_entry);
        img->itop_ebits = (const uint32_t*)(img->dev + img->h.off_itop_ebits);
        img->itop_cells = img->dev + img->h.off_itop_cells;
        img->tflags = (const uint32_t*)(img->dev + img->h.off_tflags);
    }
    if (img->h.table_in_blob) {
        img->table = (const uint32_t*)(img->dev + img->h.off_table);
        image_check_itop_reach(img);
        return ACX_OK;
    }
    // The blob carries only the sparse form.  An image that has the position-parallel structures builds the dense
    // table on first use by a serial walk (iter_long, carried-in states, final states): 34.8 GB and most of the
    // set-up time for the 1M-signature dictionary, which an ACX_SCAN_ALL-only user never needs.
    if (img->ppm_g && !acx_tune_env("ACX_EAGER_TABLE")) {
        if (lvl_host) img->lvl_host.assign(lvl_host, lvl_host + (size_t)img->h.n_levels + 1);
        return ACX_OK;
    }
    return image_build_table(img, lvl_host);
}

// build the dense transition table in HBM from the sparse form (acx_build.hip); lvl_host = host copy of the level
// boundaries, or nullptr to fetch it
static int image_build_table(acx_image* img, const uint32_t* lvl_host) {
    const size_t tbytes = (size_t)img->h.n_states * img->h.n_classes * 4;
    // the itop walk addresses the table and the cells with 32-bit offsets from one base (and may
    // read one entry past the end of the table): a table built here gets its own copy of the cells
    const size_t cells_at = (tbytes + 16 + 255) & ~(size_t)255;
    const size_t cbytes = img->itop_lds ? (size_t)img->h.itop_cell_bytes << (img->h.itop_bits * img->h.itop_depth) : 0;
    uint32_t* built = nullptr;
    HIP_TRY(hipMalloc((void**)&built, cells_at + cbytes));
    auto fail = [&](hipError_t e, const char* what) {
        (void)hipFree(built);
        return acx_fail(e == hipErrorOutOfMemory ? ACX_E_NOMEM : ACX_E_HIP, "building the dense table: %s failed: %s", what, hipGetErrorString(e));
    };
    hipError_t e = hipSuccess;
    if (cbytes) { e = hipMemcpy((uint8_t*)built + cells_at, img->itop_cells, cbytes, hipMemcpyDeviceToDevice); if (e != hipSuccess) return fail(e, "hipMemcpy(cells)"); }
    std::vector<uint32_t> lvl;
    if (!lvl_host && !img->lvl_host.empty()) lvl_host = img->lvl_host.data();
    if (!lvl_host) {
        lvl.resize((size_t)img->h.n_levels + 1);
        e = hipMemcpy(lvl.data(), img->dev + img->h.off_lvl_first, lvl.size() * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return fail(e, "hipMemcpy(levels)");
        lvl_host = lvl.data();
    }
    e = acx_launch_build_table(built, (const int32_t*)(img->dev + img->h.off_fail),
                               (const uint32_t*)(img->dev + img->h.off_edge_off), img->dev + img->h.off_edge_cls,
                               (const uint32_t*)(img->dev + img->h.off_edge_dst), (const uint32_t*)(img->dev + img->h.off_tflags),
                               lvl_host, img->h.n_levels, img->h.n_classes, img->h.state_bits, nullptr);
    if (e != hipSuccess) return fail(e, "the build kernels");
    e = hipDeviceSynchronize();
    if (e != hipSuccess) return fail(e, "hipDeviceSynchronize");
    // publish: everything else first, the table pointer last (readers ask image_ensure_table, under the same mutex)
    img->built_table = built;
    if (cbytes) img->itop_cells = (const uint8_t*)built + cells_at;
    img->table = built;
    image_check_itop_reach(img);
    return ACX_OK;
}

// the serial walks read the dense table: make sure it exists.  Thread-safe: two host threads that share an image and
// both need the table find one builder; scans that never need it (ACX_SCAN_ALL on the position-parallel kernels
// without carried states) do not come here.
static int image_ensure_table(acx_image* img) {
    std::lock_guard<std::mutex> g(img->table_mu);
    if (img->table) return ACX_OK;
    return image_build_table(img, nullptr);
}

extern "C" int acx_image_upload(const void* blob, size_t nbytes, acx_image_t** out) {
    if (!blob || !out) return acx_fail(ACX_E_INVAL, "acx_image_upload: NULL argument");
    acx_blob_header h;
    if (nbytes < sizeof h) return acx_fail(ACX_E_FORMAT, "image: truncated");
    memcpy(&h, blob, sizeof h);
    int rc = acx_blob_check_header(&h, nbytes);
    if (rc) return rc;
    acx_image* img = new (std::nothrow) acx_image();
    if (!img) return acx_fail(ACX_E_NOMEM, "acx_image_upload: out of memory");
    img->h = h; img->nbytes = nbytes; img->owns = true;
    hipError_t e = hipGetDevice(&img->device);
    if (e == hipSuccess) e = hipMalloc((void**)&img->dev, nbytes);
    if (e == hipSuccess) e = hipMemcpy(img->dev, blob, nbytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (img->dev) (void)hipFree(img->dev);
        delete img;
        return acx_fail(e == hipErrorOutOfMemory ? ACX_E_NOMEM : ACX_E_HIP, "acx_image_upload: %s", hipGetErrorString(e));
    }
    rc = image_resolve(img, (const uint32_t*)((const uint8_t*)blob + h.off_lvl_first));
    if (rc) { acx_image_free(img); return rc; }
    *out = img;
    return ACX_OK;
}

extern "C" int acx_image_adopt(void* dev_blob, size_t nbytes, const void* host_header, acx_image_t** out) {
    if (!dev_blob || !host_header || !out) return acx_fail(ACX_E_INVAL, "acx_image_adopt: NULL argument");
    acx_blob_header h;
    memcpy(&h, host_header, sizeof h);
    int rc = acx_blob_check_header(&h, nbytes);
    if (rc) return rc;
    acx_image* img = new (std::nothrow) acx_image();
    if (!img) return acx_fail(ACX_E_NOMEM, "acx_image_adopt: out of memory");
    img->h = h; img->nbytes = nbytes; img->owns = false; img->dev = (uint8_t*)dev_blob;
    (void)hipGetDevice(&img->device);
    rc = image_resolve(img, nullptr);
    if (rc) { acx_image_free(img); return rc; }
    *out = img;
    return ACX_OK;
}

// ---- one RCCL broadcast of the blob (no link-time dependency on librccl: resolved from the process) ----
#include <dlfcn.h>
namespace {
typedef int (*nccl_bcast_fn)(const void*, void*, size_t, int /*ncclDataType_t*/, int, void* /*ncclComm_t*/, hipStream_t);
typedef const char* (*nccl_errstr_fn)(int);
nccl_bcast_fn g_nccl_bcast = nullptr;
nccl_errstr_fn g_nccl_errstr = nullptr;
bool resolve_rccl() {
    if (g_nccl_bcast) return true;
    void* sym = dlsym(RTLD_DEFAULT, "ncclBroadcast");               // the RCCL already in the process (it made the communicator)
    void* lib = nullptr;
    if (!sym) {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (lib) sym = dlsym(lib, "ncclBroadcast");
    }
    if (!sym) return false;
    g_nccl_bcast = (nccl_bcast_fn)sym;
    g_nccl_errstr = (nccl_errstr_fn)(lib ? dlsym(lib, "ncclGetErrorString") : dlsym(RTLD_DEFAULT, "ncclGetErrorString"));
    return true;
}
}  // namespace

extern "C" int acx_image_broadcast(const void* host_blob, size_t nbytes, void* nccl_comm, int root, int rank, void* stream_v, acx_image_t** out) {
    if (!nccl_comm || !out || root < 0 || rank < 0) return acx_fail(ACX_E_INVAL, "acx_image_broadcast: bad argument");
    if (rank == root && (!host_blob || nbytes < ACX_BLOB_HEADER_BYTES)) return acx_fail(ACX_E_INVAL, "acx_image_broadcast: the root rank needs the blob");
    if (!resolve_rccl()) return acx_fail(ACX_E_UNSUPPORTED, "acx_image_broadcast: no RCCL in this process and librccl.so cannot be opened");
    hipStream_t s = (hipStream_t)stream_v;
    enum { NCCL_UINT8 = 1, NCCL_UINT64 = 5 };                       // ncclDataType_t, rccl.h
    auto nccl_try = [&](int rc, const char* what) -> int {
        if (rc == 0) return ACX_OK;
        return acx_fail(ACX_E_HIP, "acx_image_broadcast: %s failed: %s", what, g_nccl_errstr ? g_nccl_errstr(rc) : "RCCL error");
    };
    unsigned long long* d_size = nullptr;
    HIP_TRY(hipMalloc((void**)&d_size, sizeof *d_size));
    unsigned long long h_size = rank == root ? (unsigned long long)nbytes : 0;
    hipError_t e = hipMemcpyAsync(d_size, &h_size, sizeof h_size, hipMemcpyHostToDevice, s);
    int rc = e == hipSuccess ? nccl_try(g_nccl_bcast(d_size, d_size, 1, NCCL_UINT64, root, nccl_comm, s), "ncclBroadcast(size)") : acx_fail(ACX_E_HIP, "acx_image_broadcast: %s", hipGetErrorString(e));
    if (!rc) { e = hipMemcpyAsync(&h_size, d_size, sizeof h_size, hipMemcpyDeviceToHost, s); if (e == hipSuccess) e = hipStreamSynchronize(s); if (e != hipSuccess) rc = acx_fail(ACX_E_HIP, "acx_image_broadcast: %s", hipGetErrorString(e)); }
    (void)hipFree(d_size);
    if (rc) return rc;
    if (h_size < ACX_BLOB_HEADER_BYTES) return acx_fail(ACX_E_FORMAT, "acx_image_broadcast: the root announced %llu bytes", h_size);
    uint8_t* d_blob = nullptr;
    HIP_TRY(hipMalloc((void**)&d_blob, (size_t)h_size));
    if (rank == root) e = hipMemcpyAsync(d_blob, host_blob, (size_t)h_size, hipMemcpyHostToDevice, s); else e = hipSuccess;
    if (e == hipSuccess) rc = nccl_try(g_nccl_bcast(d_blob, d_blob, (size_t)h_size, NCCL_UINT8, root, nccl_comm, s), "ncclBroadcast(blob)");
    else rc = acx_fail(ACX_E_HIP, "acx_image_broadcast: %s", hipGetErrorString(e));
    acx_blob_header hdr;
    if (!rc) { e = hipMemcpyAsync(&hdr, d_blob, sizeof hdr, hipMemcpyDeviceToHost, s); if (e == hipSuccess) e = hipStreamSynchronize(s); if (e != hipSuccess) rc = acx_fail(ACX_E_HIP, "acx_image_broadcast: %s", hipGetErrorString(e)); }
    if (!rc) rc = acx_image_adopt(d_blob, (size_t)h_size, &hdr, out);
    if (rc) { (void)hipFree(d_blob); return rc; }
    (*out)->owns = true;                                             // the image keeps (and frees) its device copy
    return ACX_OK;
}

extern "C" void acx_image_free(acx_image_t* img) {
    if (!img) return;
    if (img->owns && img->dev) (void)hipFree(img->dev);
    if (img->built_table) (void)hipFree(img->built_table);
    if (img->long_img) acx_image_free(img->long_img);
    if (img->long_real && img->long_real_owned) (void)hipFree(img->long_real);
    delete img;
}
extern "C" int64_t acx_image_num_states(const acx_image_t* img) { return img ? img->h.n_states : 0; }
extern "C" int64_t acx_image_num_classes(const acx_image_t* img) { return img ? img->h.n_classes : 0; }
extern "C" size_t  acx_image_nbytes(const acx_image_t* img) { return img ? img->nbytes : 0; }
extern "C" void*   acx_image_dev_ptr(const acx_image_t* img) { return img ? img->dev : nullptr; }
extern "C" const void* acx_image_table_dev_ptr(const acx_image_t* img) {
    if (!img || image_ensure_table(const_cast<acx_image_t*>(img))) return nullptr;
    return img->table;
}
extern "C" int     acx_image_itop_depth(const acx_image_t* img) { return (img && img->itop_lds) ? (int)img->h.itop_depth : 0; }

// ------------------------------------------------------------------------------------
// result
// ------------------------------------------------------------------------------------
template <typename T>
struct DevBuf {
    T* p = nullptr; size_t cap = 0;   // elements
    int ensure(size_t n) {
        if (n <= cap) return ACX_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = n + n / 8 + 64;
        hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
        if (e != hipSuccess) { p = nullptr; return acx_fail(ACX_E_NOMEM, "device allocation of %zu bytes failed: %s", want * sizeof(T), hipGetErrorString(e)); }
        cap = want;
        return ACX_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};
template <typename T>
struct PinBuf {
    T* p = nullptr; size_t cap = 0;
    int ensure(size_t n) {
        if (n <= cap) return ACX_OK;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        size_t want = n + n / 8 + 64;
        hipError_t e = hipHostMalloc((void**)&p, want * sizeof(T), hipHostMallocDefault);
        if (e != hipSuccess) { p = nullptr; return acx_fail(ACX_E_NOMEM, "pinned allocation of %zu bytes failed: %s", want * sizeof(T), hipGetErrorString(e)); }
        cap = want;
        return ACX_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

struct acx_result {
    DevBuf<int32_t> counts, nev, final_state;
    DevBuf<int64_t> match_off, partials;
    DevBuf<uint2> events, matches;
    // chunked scans
    DevBuf<int32_t> nck; DevBuf<int64_t> ck_first, ck_match_off; DevBuf<acx_chunk_desc> ck;
    // position-parallel scans: record pool, where each tile's records start, pool heads + overflow flag
    DevBuf<uint2> scratch; DevBuf<uint32_t> scr_off; DevBuf<unsigned long long> ppm_ctl; DevBuf<int32_t> hay_local;
    bool ppm = false;           // the pending scan is a position-parallel one
    acx_ppm_args pend_pa; acx_ppm_compact_args pend_ca; int64_t pend_items = 0;
    const int32_t* pend_counts = nullptr; int64_t* pend_item_off = nullptr;
    acx_chunk_args pend_cka; acx_walk_args pend_tail; bool ppm_chunk = false, ppm_tail = false;
    bool ppm_stream = false; acx_ppm_gather_args pend_ga; DevBuf<uint32_t> wave_desc, wave_aux;
    // ACX_SCAN_ASYNC scans: the gather (memory bound) runs on a stream of the result's own, so that the next
    // scan kernel (instruction bound) of another result on the caller's stream overlaps it
    bool use_side = false; hipStream_t side = nullptr; int side_device = -1; hipEvent_t ev_scan = nullptr;      // (side: one of side_device's pool, not owned)
    bool ctl_zero = false;      // ppm_ctl is known to be all zero (the gather of the last fixed-stride stream scan cleaned up)
    // acx_scan_host, pipelined (scan_host_pipelined): the gather of a fixed-stride stream scan writes records and offsets
    // straight into the result's pinned host buffers (device-mapped pointers) instead of r->matches / r->match_off
    uint2* ext_matches = nullptr; int64_t ext_capacity = 0; int64_t* ext_match_off = nullptr; int64_t ext_off_base = 0;
    acx_result* long_inner = nullptr;           // ACX_SCAN_LONG position-parallel: the result of the scan over the dictionary D
    bool is_long_inner = false;                 // ... and this IS such a result (which side stream it takes: side_stream_from_pool)
    bool long_pending = false;                  // ... asynchronous: the sweep over the inner scan's records is queued behind its gather (long_complete)
    // the sweep straight from the record pool (acx_long.h acx_long_fuse_args): on the INNER result `fuse` points at the outer one's arguments — its
    // stream scan then queues k_long_gather_sweep where it would queue k_ppm_gather_pos, into `long_out` —, `fused` says whether it did
    const acx_long_fuse_args* fuse = nullptr; bool fused = false; DevBuf<uint2> long_out; acx_ppm_gather_args fused_ga;
    acx_long_fuse_args fuse_args; DevBuf<uint32_t> long_aux; bool long_nofuse = false; uint32_t fail_seen = 0;
    DevBuf<uint32_t> start_bits; size_t start_bits_words = 0; bool start_bits_clean = false;         // k_ppm_stream4 on an offsets batch: bit p = a haystack starts at byte p
    DevBuf<uint32_t> long_gtot;                                        // reports per group (k_long_sweep_raw -> k_long_place)
    acx_image* long_img = nullptr; uint32_t reruns = 0;   // reruns: scans of this result that were issued again at completion (pool too small, a broken promise)
    hipStream_t copy_stream = nullptr;
    bool ppm_self = false;      // the pending stream scan is a fixed-stride one: block sums, totals and clean-up in k_ppm_gather_pos
    int bs_parity = 0;          // which half of wave_aux the next such scan sums into
    acx_image* pend_img = nullptr;
    acx_scan_params pend_params;                // the scan as it was asked for (a stream scan that must be issued again)
    PinBuf<int64_t> h_off;
    PinBuf<acx_match_t> h_matches;
    PinBuf<int32_t> h_final;
    PinBuf<int64_t> h_total;
    // staging used only by acx_scan_host
    DevBuf<uint8_t> in_hay; DevBuf<int64_t> in_off; DevBuf<int32_t> in_init, in_base, in_skip;
    PinBuf<uint8_t> h_stage;                    // acx_scan_host_ctx: context and chunk of every haystack side by side
    // dev_skip on the kernel families that do not know it: the context's records are dropped after the scan
    const int32_t* skip_after = nullptr; const int32_t* skip_base = nullptr;
    DevBuf<int32_t> skip_kept; DevBuf<int64_t> skip_off; DevBuf<uint2> matches2;
    // ACX_SCAN_SKIP_WS: the compacted batch that was scanned, and the batch as the caller gave it (for the way back)
    bool ws_active = false;
    DevBuf<uint8_t> ws_hay; DevBuf<uint32_t> ws_map; DevBuf<int32_t> ws_cnt, ws_skip; DevBuf<int64_t> ws_tile_off, ws_off, ws_partials;
    const int64_t* ws_o_off = nullptr; int64_t ws_o_stride = 0; const int32_t* ws_o_skip = nullptr; const int32_t* ws_o_base = nullptr;
    int64_t n_hay = 0;
    int64_t total = 0;
    bool has_final = false;
    bool host_valid = false;
    // acx_trie_scan_host (acx_hostwalk.cpp): the result of a walk over the host trie lives in ordinary memory (a process
    // without a device cannot pin any); the device accessors return NULL for it
    bool host_walk = false;
    std::vector<int64_t> hw_off; std::vector<acx_match_t> hw_m; std::vector<int32_t> hw_fin;
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t done = nullptr;  // recorded after the last operation of a scan: completion of THIS result, not of the whole stream
    bool timed = false;
    bool timed_all = false;     // events around scan and expand too (params.timing == 1)
    float t_walk = 0, t_scan = 0, t_expand = 0, t_total = 0;
    // ACX_SCAN_ASYNC: kernels queued, not yet completed (see result_complete)
    bool pending = false;
    acx_expand_args pend_ea;
    int pend_variant = 0;
    int device = 0;
    ~acx_result() {
        if (pending) (void)hipStreamSynchronize(stream);
        counts.release(); nev.release(); final_state.release(); match_off.release(); partials.release();
        nck.release(); ck_first.release(); ck_match_off.release(); ck.release();
        scratch.release(); scr_off.release(); ppm_ctl.release(); hay_local.release(); wave_desc.release(); wave_aux.release();
        events.release(); matches.release(); h_off.release(); h_matches.release(); h_final.release(); h_total.release();
        in_hay.release(); in_off.release(); in_init.release(); in_base.release(); in_skip.release(); h_stage.release();
        long_out.release(); long_aux.release(); long_gtot.release(); start_bits.release();
        skip_kept.release(); skip_off.release(); matches2.release();
        ws_hay.release(); ws_map.release(); ws_cnt.release(); ws_skip.release(); ws_tile_off.release(); ws_off.release(); ws_partials.release();
        for (auto& e : ev) if (e) (void)hipEventDestroy(e);
        if (done) (void)hipEventDestroy(done);
        if (ev_scan) (void)hipEventDestroy(ev_scan);
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
        delete long_inner;
    }
};

// The side streams of asynchronous scans belong to the device, not to the results: ACX_SIDE_STREAMS of them, made at the first asynchronous
// scan on the device and kept for the life of the library.  (A stream per result — rounds 3 and 4 — makes the number of HIP streams, and of the hardware queues under
// them, grow with the number of Scanner objects a process keeps around, and WHICH queues a scan's work sits on is worth 5–10 % of its
// throughput: profiles/r5_experiments.md §10.)
//   * `iter` results (what follows their scan kernel is ONE gather of ~45 us) all take stream 0: with the caller's three scan streams that
//     is four busy queues, and the gathers of consecutive results follow one another on the chip anyway (a gather does not run beside a
//     scan kernel — registers — so it runs in the seams between them).  Config 2: 605–607 GB/s against 572–577 with a stream per result;
//   * the inner results of `iter_long` (gather + sweep + prefix sum + move: 40 % of a step) take the streams round robin, so that the tail
//     of one haystack batch overlaps the tail of the next: 227 GB/s on three streams, 216–218 on two, 201 on one.
// The lowest priority: this work fills the CUs that the scan kernels of the caller's streams leave idle, it must not take a CU before them
// (a scan block needs a whole CU: LDS and registers).
#ifndef ACX_SIDE_STREAMS
#define ACX_SIDE_STREAMS 3
#endif
#define ACX_MAX_DEVICES 64
extern "C" int acx_async_streams(void) { return (int)ACX_SIDE_STREAMS; }
// `device`: the IMAGE's device — the streams are made on it (the caller's current device is restored) and kept under its number: a
// caller whose current device is another one gets streams of the device its scan runs on.
static hipError_t side_stream_from_pool(int device, bool round_robin, hipStream_t* out) {
    static std::mutex mu;
    static hipStream_t pool[ACX_MAX_DEVICES][ACX_SIDE_STREAMS];
    static unsigned next[ACX_MAX_DEVICES];
    if (device < 0 || device >= ACX_MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> g(mu);
    int cur_dev = -1;
    { hipError_t e = hipGetDevice(&cur_dev); if (e != hipSuccess) return e; }
    struct DeviceGuard {                                       // makes `device` current while the pool's streams are made, then the caller's again
        int back = -1;
        ~DeviceGuard() { if (back >= 0) (void)hipSetDevice(back); }
    } guard;
    if (!pool[device][0] && cur_dev != device) {
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) return e;
        guard.back = cur_dev;
    }
    if (!pool[device][0]) {
        // ALL streams of the pool are made, and used once, at the first request: which queues they get then does not depend on what the
        // process scans first (made on first use, `iter_long` after an `iter` measurement in the same process ran at 211.5–213.5 GB/s where
        // it runs at 224.4–224.7 with the pool made at once; the `iter` scans in front: 590–594 either way — tools/r5_eager.sh)
        int lo_pri = 0, hi_pri = 0;
        const bool plain = acx_tune_env("ACX_SIDE_DEFAULT_PRIORITY") || hipDeviceGetStreamPriorityRange(&lo_pri, &hi_pri) != hipSuccess || lo_pri == hi_pri;
        hipStream_t made[ACX_SIDE_STREAMS];
        for (unsigned j = 0; j < ACX_SIDE_STREAMS; j++) {
            hipError_t e = plain ? hipStreamCreateWithFlags(&made[j], hipStreamNonBlocking)
                                 : hipStreamCreateWithPriority(&made[j], hipStreamNonBlocking, acx_tune_env("ACX_SIDE_HIGH_PRIORITY") ? hi_pri : lo_pri);
            if (e != hipSuccess) { for (unsigned i = 0; i < j; i++) (void)hipStreamDestroy(made[i]); return e; }
            hipEvent_t ev;                                    // (one packet through it: the stream has its queue now)
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) { (void)hipEventRecord(ev, made[j]); (void)hipEventSynchronize(ev); (void)hipEventDestroy(ev); }
        }
        for (unsigned j = 0; j < ACX_SIDE_STREAMS; j++) pool[device][j] = made[j];
    }
    *out = pool[device][round_robin ? next[device]++ % ACX_SIDE_STREAMS : 0u];
    return hipSuccess;
}

static int ppm_enqueue(acx_result* r, acx_image* img, const acx_chunk_args* ca, const acx_walk_args* tail, hipStream_t s);
static int ppm_size_pool(acx_result* r, size_t records);
enum { ACX_HOST_RETRY = 1 };            // internal: the pipelined host scan could not be used; scan_host_once takes the staged path
static int scan_batch_inner(acx_image_t* img, const acx_scan_params* p, acx_result_t** result, void* stream_v);

// dev_skip after a scan on kernels that do not know it (the serial walks, k_ppm_scan): drop the records of every
// haystack's context (a prefix of its records) and rebase the rest; the result's buffers are swapped for the new ones
static int skip_compact(acx_result* r) {
    if (!r->skip_after) return ACX_OK;
    const int32_t* skip = r->skip_after;
    r->skip_after = nullptr;
    const size_t n = (size_t)r->n_hay;
    hipStream_t s = r->stream;
    int rc;
    if ((rc = r->skip_kept.ensure(n + 1))) return rc;
    if ((rc = r->skip_off.ensure(n + 1))) return rc;
    if ((rc = r->matches2.ensure((size_t)r->total + 1))) return rc;
    if ((rc = r->partials.ensure((size_t)acx_scan_num_partials((int64_t)n) + 2))) return rc;
    HIP_TRY(acx_launch_skip_count(r->match_off.p, r->matches.p, skip, r->skip_base, (int64_t)n, r->skip_kept.p, s));
    HIP_TRY(acx_launch_scan(r->skip_kept.p, (int64_t)n, r->skip_off.p, r->partials.p, s));
    HIP_TRY(acx_launch_skip_move(r->match_off.p, r->matches.p, skip, r->skip_kept.p, r->skip_off.p, (int64_t)n, r->matches2.p, s));
    int64_t total = 0;
    HIP_TRY(hipMemcpyAsync(&total, r->skip_off.p + n, sizeof total, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    std::swap(r->matches.p, r->matches2.p); std::swap(r->matches.cap, r->matches2.cap);
    std::swap(r->match_off.p, r->skip_off.p); std::swap(r->match_off.cap, r->skip_off.cap);
    r->total = total;
    return ACX_OK;
}

// ACX_SCAN_SKIP_WS, last step of a scan: the records hold end indices of the compacted batch; send them back to the
// positions of the batch the caller gave (context off, the caller's base on)
static int ws_remap(acx_result* r) {
    if (!r->ws_active) return ACX_OK;
    r->ws_active = false;
    if (r->total <= 0) return ACX_OK;
    acx_ws_remap_args a;
    a.matches = r->matches.p; a.match_off = r->match_off.p; a.n_hay = r->n_hay; a.total = r->total;
    a.off = r->ws_o_off; a.stride = r->ws_o_stride; a.skip = r->ws_o_skip; a.index_base = r->ws_o_base;
    a.c_off = r->ws_off.p; a.c_skip = r->ws_o_skip ? r->ws_skip.p : nullptr; a.map = r->ws_map.p;
    HIP_TRY(acx_launch_ws_remap(a, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    return ACX_OK;
}

static int finish_records(acx_result* r) {
    const int rc = skip_compact(r);
    return rc ? rc : ws_remap(r);
}

// position-parallel scan: the record pool ran out (grow it and scan again) or the match buffer
// is too small (grow it and copy again: the pool is intact)
// the second half of a stream scan: k_ppm_gather_pos / k_ppm_gather — or, for the scan over iter_long's dictionary (r->fuse), the sweep straight
// from the record pool where it applies: a fixed-stride scan whose haystacks are no longer than a tile, positions a 32-bit multiply-high from
// their haystack, no stream contexts
static int launch_gather_or_fused(acx_result* r, hipStream_t g) {
    r->fused = false;
    const acx_ppm_gather_args& ga = r->pend_ga;
    if (r->fuse && r->ppm_self && !ga.off && !ga.skip && ga.stride > 0 && ga.stride <= ga.tile_pos) {
        const uint64_t span = ((uint64_t)ga.tpw + ga.share_a) * (uint64_t)ga.tile_pos, st = (uint64_t)ga.stride;
        if ((span + 2 * st) * st < ((uint64_t)1 << 32) && (uint64_t)ga.n_hay * st + st < ((uint64_t)1 << 32)) {
            int rc = r->long_out.ensure((size_t)ga.capacity + (size_t)ga.n_waves * ACX_LONG_WAVE_SLACK + 64);
            if (rc) return rc;
            r->fused_ga = ga; r->fused_ga.matches = r->long_out.p;
            HIP_TRY(acx_launch_long_gather_sweep(r->fused_ga, *r->fuse, g));
            r->fused = true;
            return ACX_OK;
        }
    }
    HIP_TRY(acx_launch_ppm_gather(r->pend_pa.wave_desc, r->pend_ga.n_waves, r->pend_item_off, r->pend_ga, g));
    if (r->pend_ga.pos_records && r->pend_ga.used_bits) r->start_bits_clean = true;       // (k_ppm_gather_pos<true> zeroes the start bitmap its scan used)
    return ACX_OK;
}

static int ppm_complete(acx_result* r) {
    hipStream_t s = r->stream;
    for (int attempt = 0;; attempt++) {
        if (r->done) HIP_TRY(hipEventSynchronize(r->done)); else HIP_TRY(hipStreamSynchronize(s));
        if (r->ppm_stream && (int32_t)r->h_total.p[2] != 0) {
            // An offsets batch held a haystack shorter than its min_hay_len promised (acx.h): the stream kernel saw more
            // starts in a tile than it has room for and said so.  Its result is void; the same scan is issued again
            // without the promise (the general kernel or the serial walks take it).
            acx_scan_params p = r->pend_params;
            p.min_hay_len = 0; p.flags &= ~(int32_t)ACX_SCAN_ASYNC;
            acx_result* self = r;
            r->reruns++;
            return scan_batch_inner(r->pend_img, &p, &self, (void*)s);
        }
        r->total = r->h_total.p[0];
        const bool overflow = (int32_t)r->h_total.p[1] != 0;
        const bool ext = r->ppm_stream && r->ppm_self && r->ext_matches;
        const bool small = r->total > (ext ? r->ext_capacity : (int64_t)r->matches.cap);
        if (!overflow && !small) break;
        if (ext && small && !overflow) return ACX_HOST_RETRY;          // (the caller's buffer: it falls back to the staged path)
        if (attempt >= 4) return acx_fail(ACX_E_NOMEM, "position-parallel scan: record pool still too small after %d attempts", attempt);
        int rc;
        r->reruns++;
        if (small && !ext) {
            if ((rc = r->matches.ensure((size_t)r->total))) return rc;
            r->pend_ca.matches = r->matches.p; r->pend_ca.capacity = (int64_t)r->matches.cap;
            r->pend_ga.matches = r->matches.p; r->pend_ga.capacity = (int64_t)r->matches.cap;
        }
        if (overflow) {
            const size_t have = r->scratch.cap, need = (size_t)r->total;
            if ((rc = ppm_size_pool(r, (need > have ? need : have) + need / 2))) return rc;
            if ((rc = ppm_enqueue(r, r->pend_img, r->ppm_chunk ? &r->pend_cka : nullptr, r->ppm_tail ? &r->pend_tail : nullptr, s))) return rc;
        } else {
            hipStream_t g = (r->ppm_stream && r->use_side && r->side) ? r->side : s;
            if (r->ppm_stream) { if ((rc = launch_gather_or_fused(r, g))) return rc; }
            else HIP_TRY(acx_launch_ppm_compact(r->pend_ca, r->pend_items, g));
            HIP_TRY(hipEventRecord(r->done, g));
        }
    }
    if (r->timed) {
        HIP_TRY(hipEventElapsedTime(&r->t_walk, r->ev[0], r->ev[1]));
        r->t_scan = 0.f; r->t_expand = 0.f; r->t_total = r->t_walk;
        if (r->timed_all) {
            HIP_TRY(hipEventElapsedTime(&r->t_scan, r->ev[1], r->ev[2]));
            HIP_TRY(hipEventElapsedTime(&r->t_expand, r->ev[2], r->ev[3]));
            HIP_TRY(hipEventElapsedTime(&r->t_total, r->ev[0], r->ev[3]));
        }
    }
    return finish_records(r);
}

// Finish a scan whose kernels are queued: wait, read the total, and if the speculative expand did
// not fit the match buffer grow it and run expand again (the events are intact).
static int long_complete(acx_result* r);
static int result_complete(acx_result* r) {
    if (!r || !r->pending) return ACX_OK;
    r->pending = false;
    if (r->long_pending) { r->long_pending = false; return long_complete(r); }
    if (r->ppm) { r->ppm = false; return ppm_complete(r); }
    hipStream_t s = r->stream;
    // wait for this scan only: later scans queued on the same stream (other result objects) keep running
    if (r->done) HIP_TRY(hipEventSynchronize(r->done)); else HIP_TRY(hipStreamSynchronize(s));
    r->total = r->h_total.p[0];
    if (r->timed) {
        HIP_TRY(hipEventElapsedTime(&r->t_walk, r->ev[0], r->ev[1]));
        r->t_scan = 0.f; r->t_expand = 0.f; r->t_total = r->t_walk;
        if (r->timed_all) {
            HIP_TRY(hipEventElapsedTime(&r->t_scan, r->ev[1], r->ev[2]));
            HIP_TRY(hipEventElapsedTime(&r->t_expand, r->ev[2], r->ev[3]));
            HIP_TRY(hipEventElapsedTime(&r->t_total, r->ev[0], r->ev[3]));
        }
    }
    if (r->total > (int64_t)r->matches.cap) {
        // first call / larger batch than ever seen: grow and run expand again
        int rc;
        if ((rc = r->matches.ensure((size_t)r->total))) return rc;
        acx_expand_args ea = r->pend_ea;
        ea.matches = r->matches.p; ea.capacity = (int64_t)r->matches.cap;
        if (r->timed_all) HIP_TRY(hipEventRecord(r->ev[2], s));
        HIP_TRY(acx_launch_expand(ea, r->pend_variant, s));
        if (r->timed_all) HIP_TRY(hipEventRecord(r->ev[3], s));
        HIP_TRY(hipStreamSynchronize(s));
        if (r->timed_all) {
            HIP_TRY(hipEventElapsedTime(&r->t_expand, r->ev[2], r->ev[3]));
            r->t_total = r->t_walk + r->t_scan + r->t_expand;
        }
    }
    return finish_records(r);
}

extern "C" int acx_result_wait(acx_result_t* r) {
    if (!r) return acx_fail(ACX_E_INVAL, "acx_result_wait: NULL result");
    return result_complete(r);
}

extern "C" void acx_result_free(acx_result_t* r) { delete r; }

// ------------------------------------------------------------------------------------
// scan driver
// ------------------------------------------------------------------------------------
static const int64_t ACX_MAX_LAUNCH_BYTES = (int64_t)4 << 30;   // event staging = 8 B per haystack byte


// ------------------------------------------------------------------------------------
// position-parallel scan driver (kernels: acx_ppm_kernels.hip)
// ------------------------------------------------------------------------------------
// (Re-)issue everything a position-parallel scan queues on its stream, from the arguments kept in
// the result: a scan whose record pool ran out is issued again with a larger pool.
static int ppm_enqueue(acx_result* r, acx_image* img, const acx_chunk_args* ca, const acx_walk_args* tail, hipStream_t s) {
    const acx_ppm_args& pa = r->pend_pa;
    const int64_t ni = r->pend_items;
    // (a stream scan leaves the control words zeroed behind it: only the first one, and one after another kernel family, clears them)
    // (the flags of the host's pinned words are cleared BEFORE anything is queued: the kernels of a small batch write them — through the
    //  device-mapped pointer — sooner than this function returns)
    r->h_total.p[1] = 0; r->h_total.p[2] = 0;
    const bool self = r->ppm_stream && r->ppm_self;
    const bool words = r->ppm_stream && r->pend_ga.host_words && r->pend_ga.ctl;      // the scan's second half writes total and flags to the host and zeroes the control words
    if (!(words && r->ctl_zero)) HIP_TRY(hipMemsetAsync(r->ppm_ctl.p, 0, 16 * sizeof(unsigned long long), s));
    r->ctl_zero = words;
    if (self) {                                         // this scan's block sums (zeroed by the gather of the scan before), the next scan's
        r->pend_pa.block_sum = r->wave_aux.p + (size_t)r->bs_parity * ACX_PPM_MAX_BLOCKS;
        r->pend_ga.block_sum = r->pend_pa.block_sum;
        r->bs_parity ^= 1;
        r->pend_ga.block_sum_next = r->wave_aux.p + (size_t)r->bs_parity * ACX_PPM_MAX_BLOCKS;
    }
    if (ca) {
        HIP_TRY(hipMemsetAsync(r->counts.p, 0, ((size_t)ni + 1) * sizeof(int32_t), s));   // tiles beyond the real chunk count read as empty
        HIP_TRY(acx_launch_chunk_count(*ca, s));
        HIP_TRY(acx_launch_scan(r->nck.p, ca->n_hay, r->ck_first.p, r->partials.p, s));
        HIP_TRY(acx_launch_chunk_fill(*ca, ni, s));
    }
    if (r->ppm_stream && pa.off && !pa.start_bits) HIP_TRY(acx_launch_ppm_first_h(pa.off, pa.n_hay, ni, (int64_t)pa.nsub * 256, (int64_t*)pa.first_h, s));
    if (r->ppm_stream && pa.off && pa.start_bits) {
        HIP_TRY(acx_launch_ppm_start_bits(pa.off, pa.n_hay, (uint32_t*)pa.start_bits, r->start_bits_words, !r->start_bits_clean, s));
        r->start_bits_clean = false;                                    // (until this scan's gather is queued)
    }
    if (r->timed) HIP_TRY(hipEventRecord(r->ev[0], s));
    HIP_TRY(acx_launch_ppm_scan(pa, ni, s));
    if (r->timed) HIP_TRY(hipEventRecord(r->ev[1], s));
    if (tail) HIP_TRY(acx_launch_tail_state(*tail, (int32_t)img->ppm.longest, s));
    hipStream_t g = s;                                 // where the rest of this scan is queued
    if (r->ppm_stream) {
        if (r->use_side) {
            if (!r->side || r->side_device != img->device) {              // (a result that moves to an image of another device takes that device's)
                HIP_TRY(side_stream_from_pool(img->device, r->is_long_inner, &r->side));
                r->side_device = img->device;
            }
            if (!r->ev_scan) HIP_TRY(hipEventCreateWithFlags(&r->ev_scan, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(r->ev_scan, s));
            HIP_TRY(hipStreamWaitEvent(r->side, r->ev_scan, 0));
            g = r->side;
        }
        if (r->timed_all) HIP_TRY(hipEventRecord(r->ev[2], g));
        { int rcg = launch_gather_or_fused(r, g); if (rcg) return rcg; }
    } else {
        HIP_TRY(acx_launch_scan(r->pend_counts, ni, r->pend_item_off, r->partials.p, s));
        if (r->timed_all) HIP_TRY(hipEventRecord(r->ev[2], s));
        HIP_TRY(acx_launch_ppm_compact(r->pend_ca, ni, s));
        if (ca) HIP_TRY(acx_launch_hay_offsets(r->ck_first.p, r->ck_match_off.p, ca->n_hay, r->match_off.p, s));
    }
    if (r->timed_all) HIP_TRY(hipEventRecord(r->ev[3], g));
    if (!words) {                                       // (k_ppm_gather_pos / k_ppm_wave_scan write total and flags into h_total themselves)
        HIP_TRY(hipMemcpyAsync(r->h_total.p, r->pend_item_off + (r->ppm_stream ? r->pend_ga.n_waves : ni), sizeof(int64_t), hipMemcpyDeviceToHost, g));
        HIP_TRY(hipMemcpyAsync(r->h_total.p + 1, r->ppm_ctl.p + 8, sizeof(int32_t), hipMemcpyDeviceToHost, g));
        HIP_TRY(hipMemcpyAsync(r->h_total.p + 2, r->ppm_ctl.p + 9, sizeof(int32_t), hipMemcpyDeviceToHost, g));
    }
    if (!r->done) HIP_TRY(hipEventCreateWithFlags(&r->done, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(r->done, g));
    return ACX_OK;
}

// size the record pool: `records` of capacity plus one grant of slack per wave, cut into n_pools
static int ppm_size_pool(acx_result* r, size_t records) {
    acx_ppm_args& pa = r->pend_pa;
    const int64_t blocks = acx_ppm_grid_blocks(pa.lds, r->pend_items, pa.reserve_cus);
    pa.n_pools = (uint32_t)(blocks < 8 ? blocks : 8);
    const size_t slack = (size_t)blocks * ACX_PPM_WAVES * 1024u;
    size_t want = records + records / 2 + slack + 1024;
    if (want >= 0xFFFFFFF0ull) want = 0xFFFFFFF0ull;                 // tile offsets into the pool are 32-bit
    int rc = r->scratch.ensure(want);
    if (rc) return rc;
    size_t cap = r->scratch.cap < 0xFFFFFFF0ull ? r->scratch.cap : 0xFFFFFFF0ull;
    pa.scratch = r->scratch.p; pa.pool_records = cap / pa.n_pools;
    r->pend_ca.scratch = r->scratch.p; r->pend_ga.scratch = r->scratch.p;
    return ACX_OK;
}

// which kernels an ACX_SCAN_ALL scan takes: 0 the serial walks, 1 k_ppm_scan (general position-parallel), 2 k_ppm_stream
static uint32_t ppm_halo_pos(const acx_ppm_header& ph) {         // whole words of the symbol and "no key" bitmaps
    return ph.longest > 1 ? ((ph.longest - 1 + 31u) / 32u) * 32u : 32u;
}
// sub-steps of 256 positions per tile: the most that fit LDS for this image (0: none).  Larger tiles fill the rounds
// better (a tile's candidates are worked off before the next one is staged).
static uint32_t ppm_stream_nsub(const acx_ppm_header& ph, uint32_t halo_pos, bool offs, bool with_gh = false) {
    const uint32_t gw = ph.g_global ? (with_gh ? ACX_PPM_GH_WORDS : 0u) : ph.g_words;
    static const uint32_t forced = [] { const char* v = acx_tune_env("ACX_PPM_NSUB"); const int x = v ? atoi(v) : 0; return (x == 4 || x == 8) ? (uint32_t)x : 0u; }();   // tuning hook
    // (8-bit symbols with the filter in LDS and no second-level filter: 2048-position tiles measured slower than 1024,
    //  203 vs 218 GB/s — too many candidates per tile for the queue; with the second level: 353 vs 340)
    const uint32_t top = forced ? forced : ((ph.sym_bits == 8 && !ph.g_global && !ph.F2) ? 4u : 8u);
    if (ph.pow2 && ph.sym_bits * ph.F < 5) return 0;               // (the stream kernel's probe: a code of at least 5 bits)
    for (uint32_t nsub = top; nsub >= 4; nsub >>= 1) {
        if (acx_ppm_stream_layout(gw, ph.sym_bits, halo_pos, nsub, offs).total_words * 4 <= ACX_PPM_LDS_BYTES) return nsub;
    }
    return 0;
}

static int ppm_plan(const acx_image* img, const acx_scan_params* p) {
    if (p->mode != ACX_SCAN_ALL || !img->ppm_g || p->dev_init_state || p->n_hay <= 0 || ((p->variant >> 23) & 1)) return 0;
    if (!p->dev_off && p->stride <= 0) return 0;
    const acx_ppm_header& ph = img->ppm;
    // k_ppm_stream: fixed stride, aligned buffer, codes that are bit fields, a halo of at most one sub-step,
    // rows and singles within 4 GiB of each other (variant bit 24: the general kernel instead, A/B)
    const uint8_t* rows = (const uint8_t*)img->ppm_kids; const uint8_t* sing = (const uint8_t*)img->ppm_chains;
    const uint8_t* lo = rows < sing ? rows : sing;
    const uint64_t re = (uint64_t)(rows - lo) + ((uint64_t)ph.n_deep + 1) * ph.K * 16, se = (uint64_t)(sing - lo) + ((uint64_t)ph.n_chain + 1) * 16;
    const uint32_t hp = ppm_halo_pos(ph);
    const bool offs = p->dev_off != nullptr;
    const int64_t total = offs ? p->hay_capacity : p->n_hay * p->stride;
    if ((offs ? p->min_hay_len >= 8 : p->stride >= 8) && ((uintptr_t)p->dev_hay & 3u) == 0 && hp <= ACX_PPM_TILE &&
        total <= 0xFFFFF000ll && !((p->variant >> 24) & 1) && (re > se ? re : se) < ((uint64_t)1 << 32) && ppm_stream_nsub(ph, hp, offs))
        return 2;
    if (ph.g_global) return 0;                     // (only the stream kernel reads the filter from global memory)
    // The general kernel (k_ppm_scan) pays ~4 instructions per position more than the serial walks when their table
    // rows are cache resident; it wins when the dense table is far beyond the caches (measured: 200k binary
    // signatures, 6.9 GB table: 77 vs 62 GB/s; 100k text keys, 185 MB: 110 vs 168).  variant bit 28 forces it.
    if (((p->variant >> 28) & 1) || (uint64_t)img->h.n_states * img->h.n_classes * 4 > ((uint64_t)1 << 30)) return 1;
    return 0;
}

// k_ppm_stream4 instead of k_ppm_stream (plan 2): what acx_ppm_stream4_eligible asks of the arguments scan_ppm fills, asked of
// the image and the parameters (keep the two in step)
static bool ppm_plan_stream4(const acx_image* img, const acx_scan_params* p) {
    const acx_ppm_header& ph = img->ppm;
    const uint32_t f2 = (img->ppm_g2 && !((p->variant >> 20) & 1)) ? ph.F2 : 0u;       // (as scan_ppm sets acx_ppm_args.F2: no second level without its bitmap)
    const bool image_ok = img->ppm_hot4 && (img->ppm_cid || img->ppm_hot12) && !((p->variant >> 19) & 1) && !p->dev_skip &&
           ph.sym_bits == 2 && ph.pow2 && ph.sym_arith != 0 && ph.K == 4 && !ph.g_global && !f2 &&
           ph.C == 9 && ph.F == 10 && ppm_halo_pos(ph) == 32 && ph.longest <= 33 && ph.g_words * 4u == (128u << 10);
    if (!image_ok) return false;
    if (p->dev_off) return p->hay_capacity < ((int64_t)1 << 32) - 4096 && ppm_stream_nsub(ph, 32, true) == 8;      // (an offsets batch: the starts as a bitmap, scan_ppm)
    return p->stride >= 8 && p->stride < 2048 && ppm_stream_nsub(ph, 32, false) == 8;
}

static acx_image* image_long(acx_image* img);
extern "C" int acx_scan_plan(const acx_image_t* img, const acx_scan_params* p) {
    if (!img || !p || p->struct_bytes != sizeof(acx_scan_params)) return -1;
    if (p->flags & ACX_SCAN_SKIP_WS) {                      // what scan_batch_ws hands to the kernels: offsets, aligned, a promise of 8 at most
        acx_scan_params q = *p;
        q.dev_hay = nullptr; q.hay_capacity = p->dev_off ? p->hay_capacity : p->n_hay * p->stride;
        q.dev_off = (const int64_t*)(uintptr_t)8; q.stride = 0;
        q.min_hay_len = (p->dev_off ? p->min_hay_len : (int32_t)(p->stride > INT32_MAX ? INT32_MAX : p->stride)) >= 8 ? 8 : 0;
        return ppm_plan(img, &q);
    }
    if (p->mode == ACX_SCAN_LONG) {                         // 10 + the plan of the scan over the dictionary D (builds D's image on first use)
        if (p->dev_init_state || p->want_final_state || p->n_hay <= 0 || ((p->variant >> 25) & 1) || !(p->dev_off || p->stride > 0)) return 0;
        acx_image* li = image_long(const_cast<acx_image_t*>(img));
        if (!li) return 0;
        acx_scan_params q = *p;
        q.mode = ACX_SCAN_ALL; q.want_final_state = 0; q.dev_skip = nullptr;
        const int lp = ppm_plan(li, &q);
        return lp ? 10 + (lp == 2 && ppm_plan_stream4(li, &q) ? 3 : lp) : 0;
    }
    const int plan = ppm_plan(img, p);
    return plan == 2 && ppm_plan_stream4(img, p) ? 3 : plan;
}

static int scan_ppm(acx_image_t* img, const acx_scan_params* p, acx_result* r, hipStream_t s, int plan) {
    const acx_ppm_header& ph = img->ppm;
    const size_t n = (size_t)p->n_hay;
    const bool chunked = p->dev_off != nullptr;
    const int64_t n_items = chunked ? p->n_hay + p->hay_capacity / ACX_PPM_TILE + 1
                                    : (p->n_hay * p->stride + ACX_PPM_TILE - 1) / ACX_PPM_TILE;
    const size_t ni = (size_t)n_items;
    int rc;
    if ((rc = r->counts.ensure(ni + 1))) return rc;
    if ((rc = r->scr_off.ensure(ni + 1))) return rc;
    if ((rc = r->match_off.ensure(n + 1))) return rc;
    if ((rc = r->ck_match_off.ensure(ni + 1))) return rc;
    if ((rc = r->partials.ensure((size_t)acx_scan_num_partials((int64_t)(ni > n ? ni : n)) + 2))) return rc;
    if ((rc = r->ppm_ctl.ensure(16))) return rc;
    if ((rc = r->h_total.ensure(3))) return rc;
    if (r->has_final && (rc = r->final_state.ensure(n + 1))) return rc;
    if (r->matches.cap == 0 && (rc = r->matches.ensure((size_t)(p->hay_capacity / 8) + 1024))) return rc;
    if (chunked) {
        if ((rc = r->nck.ensure(n + 1))) return rc;
        if ((rc = r->ck_first.ensure(n + 1))) return rc;
        if ((rc = r->ck.ensure(ni + 1))) return rc;
    } else if ((rc = r->hay_local.ensure(n + 1))) return rc;
    if (r->timed) for (auto& e : r->ev) if (!e) HIP_TRY(hipEventCreate(&e));

    acx_ppm_args& pa = r->pend_pa;
    memset(&pa, 0, sizeof pa);
    pa.hay = p->dev_hay; pa.hay_cap = p->hay_capacity; pa.stride = p->stride; pa.n_hay = p->n_hay;
    pa.stride_magic = p->stride > 1 ? ~0ull / (uint64_t)p->stride + 1 : 0;           // ceil(2^64 / stride) (stride is no power of two, or the +1 is still right)
    if (p->stride > 1 && (p->stride & (p->stride - 1)) == 0) pa.stride_magic = ((uint64_t)1 << 63) / (uint64_t)p->stride * 2;
    pa.index_base = chunked ? nullptr : p->dev_index_base;
    pa.skip = plan == 2 ? p->dev_skip : nullptr;
    pa.ck = chunked ? r->ck.p : nullptr; pa.n_items_dev = chunked ? r->ck_first.p + p->n_hay : nullptr;
    pa.n_items = n_items;
    pa.cls = img->cls; pa.g = img->ppm_g; pa.cells = img->ppm_cells; pa.top_val = img->ppm_top_val;
    pa.kids = img->ppm_kids; pa.chains = img->ppm_chains; pa.n_branch = ph.n_deep;
    pa.hot4 = ((p->variant >> 19) & 1) ? nullptr : img->ppm_hot4; pa.cid = img->ppm_cid; pa.hot12 = img->ppm_hot12 ? 1u : 0u;        // (variant bit 19: the general stream kernel instead of k_ppm_stream4, A/B)
    pa.hot = img->ppm_hot; pa.symtab = img->ppm_symtab; pa.sym_arith = ph.sym_arith; pa.sym_lut = ph.sym_lut;
    pa.K = ph.K; pa.sym_bits = ph.sym_bits; pa.pow2 = ph.pow2; pa.C = ph.C; pa.F = ph.F; pa.g_words = ph.g_words;
    pa.g2 = ((p->variant >> 20) & 1) ? nullptr : img->ppm_g2; pa.F2 = pa.g2 ? ph.F2 : 0u;      // (variant bit 20: without the second-level filter, A/B)
    pa.has_other = ph.has_other; pa.longest = ph.longest; pa.min_len = ph.min_len ? ph.min_len : 1;
    memcpy(pa.top_base, ph.top_base, sizeof pa.top_base);
    pa.lds = acx_ppm_lds_layout(ph.g_words, ph.sym_bits, ph.longest);
    pa.counts = r->counts.p; pa.scr_off = r->scr_off.p;
    pa.heads = r->ppm_ctl.p; pa.overflow = (int32_t*)(r->ppm_ctl.p + 8); pa.short_hay = (int32_t*)(r->ppm_ctl.p + 9);
    pa.hay_local = chunked ? nullptr : r->hay_local.p;
#ifdef ACX_PPM_DEV
    pa.dbg = 0;
    if (const char* e = acx_tune_env("ACX_PPM_DBG")) pa.dbg = (uint32_t)atoi(e);
#endif
    static unsigned long long* g_phase = nullptr;
    if (acx_tune_env("ACX_PPM_PHASES")) {
        if (!g_phase) { if (hipMalloc((void**)&g_phase, 64) != hipSuccess) g_phase = nullptr; }
        if (g_phase) {
            unsigned long long hph[8];
            if (hipMemcpy(hph, g_phase, 64, hipMemcpyDeviceToHost) == hipSuccess && hph[7])
                fprintf(stderr, "[phases of the previous scan, clock ticks summed over waves] %llu %llu %llu %llu %llu %llu %llu %llu (k_ppm_stream: stage+filter, push, fetch, tops, deep, place+records, -, rest; k_ppm_stream4: top+deep, convert, push+fetch, place+records, stage, -, -, loop)\n", hph[0], hph[1], hph[2], hph[3], hph[4], hph[5], hph[6], hph[7]);
            (void)hipMemset(g_phase, 0, 64);
        }
    }
#ifdef ACX_S4_WAVETIME
    {   // development: start and end of every wave of the previous k_ppm_stream4 launch (100 MHz ticks)
        static unsigned long long* g_wt = nullptr;
        static std::vector<unsigned long long> h_wt;
        const size_t W = 8 + 2 * 4096;
        if (!g_wt) { if (hipMalloc((void**)&g_wt, W * 8) != hipSuccess) g_wt = nullptr; else (void)hipMemset(g_wt, 0, W * 8); }
        if (g_wt) {
            h_wt.resize(W);
            if (hipMemcpy(h_wt.data(), g_wt, W * 8, hipMemcpyDeviceToHost) == hipSuccess && h_wt[8 + 1]) {
                unsigned long long t0 = ~0ull, t1 = 0; double sum_end = 0, sum_busy = 0; int n = 0;
                for (int w = 0; w < 4096; w++) { const unsigned long long b = h_wt[8 + 2 * w], e = h_wt[8 + 2 * w + 1]; if (!e) continue; if (b < t0) t0 = b; if (e > t1) t1 = e; n++; }
                std::vector<double> ends, blk_end(256, 0.0), starts;
                for (int w = 0; w < 4096; w++) { const unsigned long long b = h_wt[8 + 2 * w], e = h_wt[8 + 2 * w + 1]; if (!e) continue;
                    const double en = (double)(e - t0) / 100.0, st = (double)(b - t0) / 100.0; ends.push_back(en); starts.push_back(st); sum_end += en; sum_busy += en - st; if (en > blk_end[w / 16]) blk_end[w / 16] = en; }
                std::sort(ends.begin(), ends.end()); std::sort(starts.begin(), starts.end()); std::sort(blk_end.begin(), blk_end.end());
                fprintf(stderr, "[wave times of the previous k_ppm_stream4 launch, us from the first wave's start] waves %d  start p50 %.1f max %.1f | end min %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f mean %.1f | busy mean %.1f | block end min %.1f p50 %.1f max %.1f\n",
                        n, starts[starts.size() / 2], starts.back(), ends.front(), ends[ends.size() / 10], ends[ends.size() / 2], ends[ends.size() * 9 / 10], ends[ends.size() * 99 / 100], ends.back(), sum_end / n, sum_busy / n,
                        blk_end.front(), blk_end[128], blk_end.back());
                {   // where the spread comes from: by wave slot of the block, by XCD (block % 8), between and within blocks
                    double by_wid[16] = {0}, by_xcd[8] = {0}; int n_wid[16] = {0}, n_xcd[8] = {0};
                    std::vector<double> bmean(256, 0.0); std::vector<int> bn(256, 0);
                    for (int w = 0; w < 4096; w++) { const unsigned long long e = h_wt[8 + 2 * w + 1]; if (!e) continue; const double en = (double)(e - t0) / 100.0;
                        by_wid[w % 16] += en; n_wid[w % 16]++; by_xcd[(w / 16) % 8] += en; n_xcd[(w / 16) % 8]++; bmean[w / 16] += en; bn[w / 16]++; }
                    fprintf(stderr, "  mean end by wave slot:"); for (int i = 0; i < 16; i++) fprintf(stderr, " %.0f", by_wid[i] / (n_wid[i] ? n_wid[i] : 1));
                    fprintf(stderr, "\n  mean end by XCD:"); for (int i = 0; i < 8; i++) fprintf(stderr, " %.0f", by_xcd[i] / (n_xcd[i] ? n_xcd[i] : 1));
                    double gm = 0; int gb = 0; for (int b = 0; b < 256; b++) if (bn[b]) { bmean[b] /= bn[b]; gm += bmean[b]; gb++; } gm /= gb ? gb : 1;
                    double vb = 0, vw = 0; int nw = 0;
                    for (int b = 0; b < 256; b++) if (bn[b]) vb += (bmean[b] - gm) * (bmean[b] - gm);
                    for (int w = 0; w < 4096; w++) { const unsigned long long e = h_wt[8 + 2 * w + 1]; if (!e) continue; const double en = (double)(e - t0) / 100.0; vw += (en - bmean[w / 16]) * (en - bmean[w / 16]); nw++; }
                    fprintf(stderr, "\n  sd of block means %.1f us, sd within blocks %.1f us\n", sqrt(vb / (gb ? gb : 1)), sqrt(vw / (nw ? nw : 1)));
                }
            }
            (void)hipMemset(g_wt, 0, W * 8);
        }
        g_phase = g_wt;
    }
#endif
    pa.phase_out = g_phase;                      // development builds (-DACX_PPM_DEV): phase switches, timing only
    pa.halo_pos = ppm_halo_pos(ph);
    pa.fast = plan == 2;
    r->ppm_stream = pa.fast != 0;
    int64_t stream_tiles = 0;
    if (pa.fast) {   // rows and singles are addressed with 32-bit offsets from the lower of the two
        const uint8_t* rows = (const uint8_t*)img->ppm_kids; const uint8_t* sing = (const uint8_t*)img->ppm_chains;
        pa.deep_base = rows < sing ? rows : sing;
        pa.row_off = (uint32_t)(rows - pa.deep_base); pa.single_off = (uint32_t)(sing - pa.deep_base);
    }
    if (pa.fast) {
        pa.nsub = ppm_stream_nsub(ph, pa.halo_pos, chunked);
        pa.g_global = ph.g_global;
        // A hashed copy of a global filter goes into LDS whenever a tile size exists beside which it fits (variant bit 18:
        // without, A/B) — also when that means tiles of 1024 positions instead of 2048: config 4 (a million signatures) 239
        // GB/s with a 96 KiB copy beside tiles of 2048, 251 with the same copy beside tiles of 1024, 268 with 122 KiB beside
        // tiles of 1024 (profiles/experiments/README.md).
        pa.gh = nullptr;
        if (ph.g_global && img->ppm_gh && !((p->variant >> 18) & 1)) {
            const uint32_t ns = ppm_stream_nsub(ph, pa.halo_pos, chunked, true);
            if (ns) { pa.nsub = ns; pa.gh = img->ppm_gh; }
        }
        pa.lds = acx_ppm_stream_layout(ph.g_global ? (pa.gh ? ACX_PPM_GH_WORDS : 0u) : ph.g_words, ph.sym_bits, pa.halo_pos, pa.nsub, chunked);
    }
    if (pa.fast) {
        const int64_t tpos = (int64_t)pa.nsub * 256;
        const int64_t total = chunked ? p->hay_capacity : p->n_hay * p->stride;
        stream_tiles = (total + tpos - 1) / tpos;
        pa.m24 = (!chunked && p->stride < 2048) ? (uint32_t)(((1u << 23) + (uint32_t)p->stride - 1) / (uint32_t)p->stride) : 0u;
        // An asynchronous scan finishes on a side stream.  Leaving CUs free for that copy while the NEXT batch is
        // scanned did not pay (config 2, 4 / 8 / 16 CUs: 364 / 363 / 355 GB/s against 375 with none): hook only.
        if ((p->flags & ACX_SCAN_ASYNC) && !acx_tune_env("ACX_NO_SIDE_STREAM")) {
            static const int env_res = [] { const char* v = acx_tune_env("ACX_PPM_RESERVE_CUS"); return v ? atoi(v) : 0; }();
            pa.reserve_cus = env_res > 0 ? (uint32_t)env_res : 0u;
        }
        const int64_t blocks = acx_ppm_grid_blocks(pa.lds, stream_tiles, pa.reserve_cus);
        const int64_t n_waves = blocks * ACX_PPM_WAVES;
        if ((rc = r->wave_desc.ensure((size_t)n_waves * ACX_PPM_DESC_WORDS))) return rc;
        if ((rc = r->ck_match_off.ensure((size_t)n_waves + 1))) return rc;
        if ((rc = r->hay_local.ensure(n + 1))) return rc;
        pa.wave_desc = r->wave_desc.p;
        pa.block_sum = nullptr;
        // an offsets batch that k_ppm_stream4 takes (variant bit 19: the general stream kernel, A/B): the haystacks' starts as a bitmap for the
        // scan, records with global positions, k_ppm_gather_pos<true> behind it — everything below as for a fixed stride
        pa.off = chunked ? p->dev_off : nullptr;
        const bool s4o = chunked && acx_ppm_stream4_offs_ok(pa) && p->hay_capacity < ((int64_t)1 << 32) - 4096;
        pa.off = nullptr;
        if (s4o) {
            const size_t words = ((size_t)stream_tiles + 3) * 64 + 64;
            if (r->start_bits.cap < words) r->start_bits_clean = false;   // (a new buffer; the one in use is cleaned by every scan's gather)
            if ((rc = r->start_bits.ensure(words))) return rc;
            r->start_bits_words = words;
            pa.start_bits = r->start_bits.p;
        }
        if (!chunked || s4o) {
            // fixed stride: block sums instead of a prefix-sum launch, totals and clean-up by k_ppm_gather_pos (acx_kernels.h);
            // two sets of sums: the gather of one scan zeroes the set of the result's next scan
            if (blocks > ACX_PPM_MAX_BLOCKS) return acx_fail(ACX_E_UNSUPPORTED, "position-parallel scan: %lld blocks", (long long)blocks);
            if (r->wave_aux.cap < 2 * ACX_PPM_MAX_BLOCKS) {
                if ((rc = r->wave_aux.ensure(2 * ACX_PPM_MAX_BLOCKS))) return rc;
                HIP_TRY(hipMemset(r->wave_aux.p, 0, 2 * ACX_PPM_MAX_BLOCKS * sizeof(uint32_t)));
                r->bs_parity = 0;
            }
            if ((rc = r->h_total.ensure(4))) return rc;
        }
        pa.hay_local = r->hay_local.p;
        pa.n_items = stream_tiles;
        pa.ck = nullptr; pa.n_items_dev = nullptr;
        pa.index_base = p->dev_index_base;
        if (chunked) {                              // offsets batch: first haystack at or after every tile (one binary search per tile)
            if ((rc = r->ck_first.ensure((size_t)stream_tiles + 2))) return rc;
            pa.off = p->dev_off; pa.first_h = r->ck_first.p;
        }
        acx_ppm_gather_args& ga = r->pend_ga;
        memset(&ga, 0, sizeof ga);
        ga.wave_desc = r->wave_desc.p; ga.wave_off = r->ck_match_off.p; ga.n_waves = n_waves;
        ga.matches = r->matches.p; ga.capacity = (int64_t)r->matches.cap;
        ga.hay_local = r->hay_local.p; ga.match_off = r->match_off.p;
        if (!chunked && r->ext_matches) { ga.matches = r->ext_matches; ga.capacity = r->ext_capacity; ga.match_off = r->ext_match_off; ga.off_base = r->ext_off_base; }
        ga.n_hay = p->n_hay; ga.stride = p->stride;
        ga.off = chunked ? p->dev_off : nullptr;
        ga.pos_records = s4o ? 1 : 0; ga.used_bits = s4o ? r->start_bits.p : nullptr; ga.used_words = s4o ? (uint64_t)(r->start_bits_words & ~(size_t)3) : 0;
        ga.tile_pos = tpos; ga.tpw = (stream_tiles + n_waves - 1) / n_waves;
        // unequal runs for the waves of a block (acx_ppm_layout.h; variant bit 17: equal runs, A/B)
        pa.share_a = pa.share_b = 0;
        if (!((p->variant >> 17) & 1) && ga.tpw >= 8 && ga.tpw < 100000) {
            uint32_t pa_ = 160, pb_ = 55;                               // per mille of tpw
            if (const char* e = acx_tune_env("ACX_S4_SHARE")) { int x = 0, y = 0; if (sscanf(e, "%d,%d", &x, &y) == 2 && x >= 0 && y >= 0 && x < 900 && y <= x) { pa_ = (uint32_t)x; pb_ = (uint32_t)y; } }   // tuning hook
            pa.share_a = (uint32_t)((ga.tpw * pa_ + 500) / 1000); pa.share_b = (uint32_t)((ga.tpw * pb_ + 500) / 1000);
        }
        ga.share_a = pa.share_a; ga.share_b = pa.share_b;
        ga.stride_magic = pa.stride_magic; ga.index_base = (chunked && !s4o) ? nullptr : p->dev_index_base; ga.skip = chunked ? nullptr : p->dev_skip;
        {   // (both gathers of the stream scans report through the host's pinned words and clean the control words up: k_ppm_gather_pos, and —
            //  offsets batches — k_ppm_wave_scan in front of k_ppm_gather)
            void* dp = nullptr;
            HIP_TRY(hipHostGetDevicePointer(&dp, r->h_total.p, 0));
            ga.host_words = (long long*)dp; ga.ctl = r->ppm_ctl.p;
        }
        r->ppm_self = !chunked || s4o;
    }

    acx_ppm_compact_args& ca = r->pend_ca;            // (the general kernel: per-tile counts, scan, compact)
    memset(&ca, 0, sizeof ca);
    ca.counts = r->counts.p; ca.scr_off = r->scr_off.p;
    ca.item_off = r->ck_match_off.p; ca.n_items = n_items; ca.n_items_dev = nullptr;
    ca.matches = r->matches.p; ca.capacity = (int64_t)r->matches.cap;
    ca.hay_local = chunked ? nullptr : r->hay_local.p; ca.match_off = r->match_off.p; ca.n_hay = p->n_hay; ca.stride = p->stride;
    r->pend_items = pa.fast ? stream_tiles : n_items; r->pend_counts = r->counts.p; r->pend_item_off = r->ck_match_off.p;
    if ((rc = ppm_size_pool(r, r->matches.cap))) return rc;

    r->ppm_chunk = chunked && !pa.fast;
    if (r->ppm_chunk) {
        acx_chunk_args& cka = r->pend_cka;
        cka.off = p->dev_off; cka.stride = p->stride; cka.n_hay = p->n_hay; cka.index_base = p->dev_index_base;
        cka.chunk_bytes = ACX_PPM_TILE; cka.halo = ph.longest > 0 ? (int32_t)ph.longest - 1 : 0;
        cka.nck = r->nck.p; cka.ck_first = r->ck_first.p; cka.ck = r->ck.p;
    }
    r->ppm_tail = r->has_final;
    if (r->has_final) {
        if ((rc = image_ensure_table(img))) return rc;
        acx_walk_args& wa = r->pend_tail;
        memset(&wa, 0, sizeof wa);
        wa.hay = p->dev_hay; wa.hay_cap = p->hay_capacity; wa.off = p->dev_off; wa.stride = p->stride; wa.n_hay = p->n_hay;
        wa.cls = img->cls; wa.table = img->table; wa.row_bytes = img->h.n_classes * 4u; wa.state_bits = img->h.state_bits; wa.n_states = img->h.n_states;
        wa.final_state = r->final_state.p;
    }
    r->pend_img = img; r->pend_params = *p;
    // acx_scan_plan (what bench.py names its roofline kernel by) and the launcher must agree on k_ppm_stream4: the plan asks the image
    // and the parameters, the launcher the arguments filled above — a scan on which they differ is refused, not mislabelled
    if (pa.fast && acx_ppm_stream4_eligible(pa) != ppm_plan_stream4(img, p))
        return acx_fail(ACX_E_STATE, "internal: acx_scan_plan and the launcher disagree on k_ppm_stream4 for this scan (ppm_plan_stream4 / acx_ppm_stream4_eligible)");
    r->use_side = (p->flags & ACX_SCAN_ASYNC) != 0 && r->ppm_stream && !acx_tune_env("ACX_NO_SIDE_STREAM");
    if ((rc = ppm_enqueue(r, img, r->ppm_chunk ? &r->pend_cka : nullptr, r->has_final ? &r->pend_tail : nullptr, s))) return rc;
    r->pending = true; r->ppm = true;
    if (p->flags & ACX_SCAN_ASYNC) return ACX_OK;
    return result_complete(r);
}

// ---- ACX_SCAN_LONG, position-parallel (acx_long.cpp, acx_long.hip) ---------------------------------------------------------
// The image over the dictionary D of `img`, built on first use: the blob comes back from the device (the sparse form of the
// transitions, fail links and key flags are all in it), acx_blob_long_trie makes the trie of D, flatten + upload as for any
// automaton.  nullptr: the form does not apply (no position-parallel section, nodes deeper than 63, D has no such section).
static acx_image* image_long(acx_image* img) {
    std::lock_guard<std::mutex> g(img->long_mu);
    if (img->long_state) return img->long_state > 0 ? img->long_img : nullptr;
    // long_state = -1 only where the form does not APPLY (no position-parallel section, a node deeper than 63 letters, 2^24 entries or
    // more, a dictionary that gets no such section itself).  A failure for want of memory or of the device leaves it 0 — the next
    // ACX_SCAN_LONG scan tries again — and says why through acx_last_error(); this scan takes the serial walk.
    if (!img->ppm_g || !img->dev) { img->long_state = -1; return nullptr; }
    acx_trie_t* t = nullptr; int32_t* real = nullptr; int64_t n = 0; int32_t longest = 0;
    {
        std::vector<uint8_t> host;
        try { host.resize(img->nbytes); } catch (const std::bad_alloc&) { (void)acx_fail(ACX_E_NOMEM, "iter_long: no host memory for the image (%zu bytes): the serial walk takes this scan", img->nbytes); return nullptr; }
        if (hipMemcpy(host.data(), img->dev, img->nbytes, hipMemcpyDeviceToHost) != hipSuccess) { (void)acx_fail(ACX_E_HIP, "iter_long: the image did not come back from the device: the serial walk takes this scan"); return nullptr; }
        const int rcb = acx_blob_long_trie(host.data(), img->nbytes, &t, &real, &n, &longest);
        if (rcb == ACX_E_NOMEM) return nullptr;                        // (the message is acx_blob_long_trie's)
        if (rcb != ACX_OK || n == 0 || !t) { img->long_state = -1; return nullptr; }
    }
    void* blob2 = nullptr; size_t nb2 = 0;
    int rc = acx_flatten_ex(t, ACX_FLATTEN_NO_ITOP | ACX_FLATTEN_TABLE_DEVICE | ACX_FLATTEN_HOT12, &blob2, &nb2);
    acx_trie_free(t);
    acx_image* li = nullptr;
    if (!rc) { rc = acx_image_upload(blob2, nb2, &li); acx_blob_free(blob2); }
    if (rc || !li) { if (li) acx_image_free(li); free(real); if (rc != ACX_E_NOMEM && rc != ACX_E_HIP) img->long_state = -1; return nullptr; }
    if (!li->ppm_g) { acx_image_free(li); free(real); img->long_state = -1; return nullptr; }
    li->long_state = -1;                                              // (no dictionary of the dictionary)
    int32_t* d_real = nullptr;
    if (hipMalloc((void**)&d_real, (size_t)n * 4) != hipSuccess || hipMemcpy(d_real, real, (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess) {
        if (d_real) (void)hipFree(d_real);
        acx_image_free(li); free(real);
        (void)acx_fail(ACX_E_NOMEM, "iter_long: no device memory for the dictionary's values: the serial walk takes this scan");
        return nullptr;
    }
    free(real);
    img->long_img = li; img->long_real = d_real; img->long_n_real = n; img->long_longest = (uint32_t)longest; img->long_state = 1;
    return li;
}

// ---- the dictionary of iter_long as a travelling companion of the blob (multi-GPU: SURVEY §8e) ----------------------------------------
// image_long above builds the dictionary per image, from the image itself: on N ranks that is N device-to-host copies of the whole
// blob and N host builds.  acx_blob_long_pack builds it ONCE, from the host blob, into a relocatable pack that rides behind the blob
// in the ONE broadcast (pyahocorasick_amd/parallel.py broadcast_image(long_pack=True)); acx_image_set_long installs it — a pack already
// in device memory is adopted in place.
struct acx_long_pack_header {       // 256 bytes, little-endian
    uint64_t magic;                 // "ACXLONG1"
    uint64_t total_bytes;           // of the pack
    uint64_t d_off, d_bytes;        // the dictionary's flat image (0, 0: the position-parallel form does not apply to this automaton)
    uint64_t real_off, n_real;      // int32[n_real]: what iter_long reports for dictionary entry i
    uint32_t longest, reserved0;
    uint64_t trie_version;          // of the blob the pack was made from
    uint8_t pad[256 - 64];
};
static_assert(sizeof(acx_long_pack_header) == 256, "acx_long_pack_header is 256 bytes");
static const uint64_t ACX_LONG_PACK_MAGIC = 0x31474E4F4C584341ull;     // "ACXLONG1"

extern "C" int acx_blob_long_pack(const void* blob, size_t nbytes, void** pack_out, size_t* pack_bytes) {
    if (!blob || !pack_out || !pack_bytes) return acx_fail(ACX_E_INVAL, "acx_blob_long_pack: NULL argument");
    *pack_out = nullptr; *pack_bytes = 0;
    acx_trie_t* t = nullptr; int32_t* real = nullptr; int64_t n = 0; int32_t longest = 0;
    int rc = acx_blob_long_trie(blob, nbytes, &t, &real, &n, &longest);
    if (rc) return rc;
    void* d = nullptr; size_t dn = 0;
    if (n > 0 && t) {
        rc = acx_flatten_ex(t, ACX_FLATTEN_NO_ITOP | ACX_FLATTEN_TABLE_DEVICE | ACX_FLATTEN_HOT12, &d, &dn);
        acx_trie_free(t);
        if (rc) { free(real); return rc; }
        acx_blob_header dh; memcpy(&dh, d, sizeof dh);
        if (!dh.off_ppm) { acx_blob_free(d); d = nullptr; dn = 0; n = 0; }     // (a dictionary without a position-parallel section: the form does not apply)
    } else if (t) acx_trie_free(t);
    acx_long_pack_header h;
    memset(&h, 0, sizeof h);
    h.magic = ACX_LONG_PACK_MAGIC;
    h.d_off = dn ? 256 : 0; h.d_bytes = dn;
    h.real_off = dn ? 256 + ((dn + 255) & ~(size_t)255) : 0; h.n_real = dn ? (uint64_t)n : 0;
    h.longest = (uint32_t)longest;
    { acx_blob_header bh; memcpy(&bh, blob, sizeof bh); h.trie_version = bh.trie_version; }
    h.total_bytes = dn ? h.real_off + (((size_t)n * 4 + 255) & ~(size_t)255) : 256;
    uint8_t* out = (uint8_t*)calloc(1, (size_t)h.total_bytes);
    if (!out) { if (d) acx_blob_free(d); free(real); return acx_fail(ACX_E_NOMEM, "acx_blob_long_pack: out of memory"); }
    memcpy(out, &h, sizeof h);
    if (dn) { memcpy(out + h.d_off, d, dn); memcpy(out + h.real_off, real, (size_t)n * 4); }
    if (d) acx_blob_free(d);
    free(real);
    *pack_out = out; *pack_bytes = (size_t)h.total_bytes;
    return ACX_OK;
}

extern "C" int acx_image_set_long(acx_image_t* img, const void* pack, size_t pack_bytes, int on_device) {
    if (!img || !pack || pack_bytes < sizeof(acx_long_pack_header)) return acx_fail(ACX_E_INVAL, "acx_image_set_long: bad argument");
    std::lock_guard<std::mutex> g(img->long_mu);
    if (img->long_state) return acx_fail(ACX_E_STATE, "acx_image_set_long: the image already has its iter_long dictionary (or knows it gets none)");
    acx_long_pack_header h;
    if (on_device) HIP_TRY(hipMemcpy(&h, pack, sizeof h, hipMemcpyDeviceToHost)); else memcpy(&h, pack, sizeof h);
    // (every bound by subtraction — a sum of two fields of a malformed header may wrap — and the alignments the device reads rely on:
    //  the dictionary's image is adopted IN PLACE when the pack is in device memory, its values are read as 32-bit words)
    const bool empty = h.d_bytes == 0;
    if (h.magic != ACX_LONG_PACK_MAGIC || h.total_bytes > pack_bytes || h.total_bytes < sizeof(acx_long_pack_header) ||
        (!empty && (h.d_off < 256 || h.d_off % 256 != 0 || h.d_off > h.total_bytes || h.d_bytes > h.total_bytes - h.d_off || h.d_bytes < sizeof(acx_blob_header) ||
                    h.real_off % 4 != 0 || h.real_off > h.total_bytes || h.real_off < h.d_off + h.d_bytes || h.n_real > (h.total_bytes - h.real_off) / 4)))
        return acx_fail(ACX_E_FORMAT, "acx_image_set_long: not a pack of acx_blob_long_pack");
    if (h.trie_version != img->h.trie_version) return acx_fail(ACX_E_STATE, "acx_image_set_long: the pack was made from another version of the automaton");
    if (!h.d_bytes || !img->ppm_g) { img->long_state = -1; return ACX_OK; }       // (does not apply: the serial walk stays)
    const uint8_t* base = (const uint8_t*)pack;
    acx_image* li = nullptr;
    int32_t* d_real = nullptr;
    int rc;
    if (on_device) {
        acx_blob_header dh;
        HIP_TRY(hipMemcpy(&dh, base + h.d_off, sizeof dh, hipMemcpyDeviceToHost));
        if ((rc = acx_image_adopt((void*)(base + h.d_off), (size_t)h.d_bytes, &dh, &li))) return rc;     // (in place: the caller keeps the pack alive, as it keeps the blob)
        d_real = (int32_t*)(base + h.real_off);
        img->long_real_owned = false;
    } else {
        if ((rc = acx_image_upload(base + h.d_off, (size_t)h.d_bytes, &li))) return rc;
        if (hipMalloc((void**)&d_real, (size_t)h.n_real * 4 + 4) != hipSuccess || hipMemcpy(d_real, base + h.real_off, (size_t)h.n_real * 4, hipMemcpyHostToDevice) != hipSuccess) {
            if (d_real) (void)hipFree(d_real);
            acx_image_free(li);
            return acx_fail(ACX_E_NOMEM, "acx_image_set_long: no device memory for the dictionary's values");
        }
        img->long_real_owned = true;
    }
    if (!li->ppm_g) { acx_image_free(li); if (img->long_real_owned && d_real) (void)hipFree(d_real); img->long_state = -1; return ACX_OK; }
    li->long_state = -1;
    img->long_img = li; img->long_real = d_real; img->long_n_real = (int64_t)h.n_real; img->long_longest = h.longest; img->long_state = 1;
    return ACX_OK;
}
// 1: the image has its dictionary, -1: it gets none (the serial walk), 0: not built yet (the first ACX_SCAN_LONG scan builds it)
extern "C" int acx_image_long_state(const acx_image_t* img) { return img ? img->long_state : 0; }

enum { ACX_LONG_FALLBACK = 2 };         // internal: not this batch (the serial walk takes it)
// the sweep over the records of the scan over D (in->matches / in->match_off, on the device), a prefix sum, the move, the total into
// the result's pinned word: queued on `g`
static int long_enqueue_sweep(acx_result* r, acx_image* img, hipStream_t g) {
    acx_result* in = r->long_inner;
    const size_t n = (size_t)r->n_hay;
    int rc;
    if ((rc = r->h_total.ensure(4))) return rc;
    if (in->fused) {
        // the inner scan's second half WAS the sweep (k_long_gather_sweep, straight from the record pool): counts per haystack and the packed
        // reports of every scan wave are there — a prefix sum and the move remain
        if ((rc = r->matches.ensure(in->matches.cap + 1))) return rc;
        if (r->timed) HIP_TRY(hipEventRecord(r->ev[2], g));
        HIP_TRY(acx_launch_scan(r->counts.p, (int64_t)n, r->match_off.p, r->partials.p, g));
        HIP_TRY(acx_launch_long_move_waves(in->fused_ga, r->fuse_args, r->match_off.p, img->long_real, r->matches.p, g));
        if (r->timed) HIP_TRY(hipEventRecord(r->ev[3], g));
        HIP_TRY(hipMemcpyAsync(r->h_total.p + 3, r->fuse_args.fail, sizeof(uint32_t), hipMemcpyDeviceToHost, g));
    } else {
        if ((rc = r->matches.ensure(in->matches.cap + 1))) return rc;      // (no more records than the scan over D can hold)
        acx_long_args la;
        la.rec = in->matches.p; la.off = in->match_off.p; la.n_hay = r->n_hay; la.index_base = r->pend_params.dev_index_base;
        la.longest = img->long_longest; la.counts = r->counts.p;
        // (queued behind a scan that may turn out incomplete — more records than its buffer holds: its offsets then point beyond the
        //  buffer — the kernels look at the scan's total first and leave; entry indices are checked against the dictionary's size)
        la.rec_capacity = (int64_t)in->matches.cap; la.n_real = img->long_n_real;
        la.gtot = nullptr;
        la.scan_words = (in->ppm_stream && in->pend_ga.host_words && in->pend_ga.ctl) ? in->pend_ga.host_words : nullptr;   // (what the inner scan's gather reports: see acx_long_args)
        la.compact = (r->pend_params.variant >> 27) & 1;                  // (A/B and the multi-way tests: the compact form of round 5)
        if (r->timed) HIP_TRY(hipEventRecord(r->ev[2], g));
        if (!la.compact && !((r->pend_params.variant >> 29) & 1)) {
            // the prefix sum and the move in one launch (acx_long.hip: k_long_place)
            const size_t groups = (n + 63) / 64;
            if ((rc = r->long_gtot.ensure(groups + 1))) return rc;
            la.gtot = r->long_gtot.p;
            HIP_TRY(acx_launch_long_sweep(la, g));
            HIP_TRY(acx_launch_long_place(la, r->match_off.p, img->long_real, r->matches.p, g));
        } else {
            // (variant bit 29, and the compact form: sweep in place, a three-launch prefix sum over the counts, a move; A/B and the multi-way tests)
            HIP_TRY(acx_launch_long_sweep(la, g));
            HIP_TRY(acx_launch_scan(r->counts.p, (int64_t)n, r->match_off.p, r->partials.p, g));
            HIP_TRY(acx_launch_long_move(la, r->match_off.p, img->long_real, r->matches.p, g));
        }
        if (r->timed) HIP_TRY(hipEventRecord(r->ev[3], g));
    }
    HIP_TRY(hipMemcpyAsync(r->h_total.p, r->match_off.p + n, sizeof(int64_t), hipMemcpyDeviceToHost, g));
    if (!r->done) HIP_TRY(hipEventCreateWithFlags(&r->done, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(r->done, g));
    return ACX_OK;
}
static int scan_long_ppm(acx_image* img, acx_image* li, const acx_scan_params* p, acx_result* r, hipStream_t s);
// completion of an ACX_SCAN_LONG scan in its position-parallel form: the scan over D first (a pool that was too small is scanned
// again there — the sweep then ran over nothing useful and is issued again), then the sweep's own event
static int long_complete(acx_result* r) {
    acx_result* in = r->long_inner;
    acx_image* img = r->long_img;
    const uint32_t before = in->reruns;
    int rc = result_complete(in);
    if (rc) return rc;
    if (in->reruns != before) {                                       // (the sweep that was queued saw an incomplete scan and left at once, or swept stale records)
        HIP_TRY(hipEventSynchronize(r->done));
        if ((rc = long_enqueue_sweep(r, img, r->stream))) return rc;
    }
    HIP_TRY(hipEventSynchronize(r->done));
    if (in->fused && (uint32_t)r->h_total.p[3] != r->fail_seen) {
        // a batch of 64 haystacks held more records than a wave's LDS (or a wave's reports outgrew its region): this result sweeps the
        // GATHERED records from now on (k_ppm_gather_pos + k_long_sweep, which takes haystacks of any size), starting with this scan
        r->fail_seen = (uint32_t)r->h_total.p[3];
        r->long_nofuse = true;
        acx_scan_params p = r->pend_params;
        p.flags &= ~(int32_t)ACX_SCAN_ASYNC;
        return scan_long_ppm(img, img->long_img, &p, r, r->stream);
    }
    r->total = r->h_total.p[0]; r->has_final = false; r->ppm = false;
    if (r->timed) {
        float t_in_walk = 0, t_in_scan = 0, t_in_exp = 0, t_in_total = 0, t_sweep = 0;
        if (acx_result_timing(in, &t_in_walk, &t_in_scan, &t_in_exp, &t_in_total) != ACX_OK) { t_in_walk = t_in_total = 0; }
        HIP_TRY(hipEventElapsedTime(&t_sweep, r->ev[2], r->ev[3]));
        r->t_walk = t_in_walk; r->t_scan = t_in_total > t_in_walk ? t_in_total - t_in_walk : 0.f; r->t_expand = t_sweep;
        r->t_total = (t_in_total > 0 ? t_in_total : t_in_walk) + t_sweep;
    }
    return finish_records(r);                                         // (ACX_SCAN_SKIP_WS: the indices go back to the batch the caller gave)
}
// the scan over D (any position-parallel plan), then one sweep per haystack over its records, a prefix sum, a move.  With
// ACX_SCAN_ASYNC everything is queued — the scan kernel on the caller's stream, its gather and the sweep behind it on the inner
// result's side stream, beside the caller's next scan kernel — and completes in acx_result_wait / the first accessor.
static int scan_long_ppm(acx_image* img, acx_image* li, const acx_scan_params* p, acx_result* r, hipStream_t s) {
    acx_scan_params q = *p;
    q.mode = ACX_SCAN_ALL; q.want_final_state = 0; q.dev_skip = nullptr; q.dev_init_state = nullptr;
    if (!ppm_plan(li, &q)) return ACX_LONG_FALLBACK;
    if (!r->long_inner) {
        r->long_inner = new (std::nothrow) acx_result();
        if (!r->long_inner) return acx_fail(ACX_E_NOMEM, "acx_scan_batch: out of memory");
        r->long_inner->is_long_inner = true;
    }
    acx_result* in = r->long_inner;
    const size_t n = (size_t)p->n_hay;
    int rc;
    if ((rc = r->counts.ensure(n + 1))) return rc;
    if ((rc = r->match_off.ensure(n + 1))) return rc;
    if ((rc = r->partials.ensure((size_t)acx_scan_num_partials((int64_t)n) + 2))) return rc;
    // the sweep straight from the record pool — the inner scan launches it in place of its gather — is OPT-IN (variant bit 26: A/B and the four-way
    // tests): measured, it loses (profiles/r5_experiments.md §7: k_long_gather_sweep 239 us where k_ppm_gather_pos + k_long_sweep take 65 + 168 —
    // reading the pool's grants costs what the gather cost, and the staging that finds haystack boundaries in a stream of positions is dearer)
    in->fuse = nullptr;
    if (!r->long_nofuse && ((p->variant >> 26) & 1)) {
        const size_t need = 64 + (size_t)ACX_PPM_MAX_BLOCKS * 16;
        if (r->long_aux.cap < need) { if ((rc = r->long_aux.ensure(need))) return rc; HIP_TRY(hipMemset(r->long_aux.p, 0, 64 * sizeof(uint32_t))); r->fail_seen = 0; }
        r->fuse_args.counts = r->counts.p; r->fuse_args.fail = r->long_aux.p; r->fuse_args.wave_base = r->long_aux.p + 64;
        r->fuse_args.longest = img->long_longest; r->fuse_args.n_real = img->long_n_real;
        in->fuse = &r->fuse_args;
    }
    in->fused = false;
    rc = scan_batch_inner(li, &q, &in, (void*)s);
    if (rc) return rc;
    if (r->timed) for (auto& e : r->ev) if (!e) HIP_TRY(hipEventCreate(&e));
    r->pend_params = *p; r->long_img = img;
    // where the inner scan's last kernel (its gather) was queued: the sweep goes behind it
    hipStream_t g = (in->pending && in->ppm_stream && in->use_side && in->side) ? in->side : s;
    if ((rc = long_enqueue_sweep(r, img, g))) return rc;
    r->pending = true; r->long_pending = true; r->ppm = false;
    if (p->flags & ACX_SCAN_ASYNC) return ACX_OK;
    return result_complete(r);
}

static int scan_batch_inner(acx_image_t* img, const acx_scan_params* p, acx_result_t** result, void* stream_v) {
    if (!img || !p || !result) return acx_fail(ACX_E_INVAL, "acx_scan_batch: NULL argument");
    if (p->struct_bytes != sizeof(acx_scan_params))
        return acx_fail(ACX_E_INVAL, "acx_scan_batch: params.struct_bytes = %u, library expects %zu", p->struct_bytes, sizeof(acx_scan_params));
    if (p->mode != ACX_SCAN_ALL && p->mode != ACX_SCAN_LONG) return acx_fail(ACX_E_INVAL, "acx_scan_batch: bad mode %d", p->mode);
    if (p->n_hay < 0 || p->hay_capacity < 0) return acx_fail(ACX_E_INVAL, "acx_scan_batch: negative size");
    if (p->n_hay > 0 && !p->dev_hay && p->hay_capacity > 0) return acx_fail(ACX_E_INVAL, "acx_scan_batch: dev_hay is NULL");
    if (!p->dev_off) {
        if (p->stride < 0 || p->stride > INT32_MAX) return acx_fail(ACX_E_INVAL, "acx_scan_batch: stride out of range");
        if (p->n_hay * p->stride > p->hay_capacity) return acx_fail(ACX_E_INVAL, "acx_scan_batch: n_hay*stride exceeds hay_capacity");
    }
    if (p->dev_skip && (p->mode != ACX_SCAN_ALL || p->dev_init_state))
        return acx_fail(ACX_E_INVAL, "acx_scan_batch: dev_skip is for ACX_SCAN_ALL without carried states (the context replaces them)");
    if (p->hay_capacity > ACX_MAX_LAUNCH_BYTES)
        return acx_fail(ACX_E_UNSUPPORTED, "acx_scan_batch: %lld haystack bytes in one call; split the batch into calls of <= %lld bytes",
                        (long long)p->hay_capacity, (long long)ACX_MAX_LAUNCH_BYTES);

    hipStream_t s = (hipStream_t)stream_v;
    acx_result* r = *result;
    if (!r) {
        r = new (std::nothrow) acx_result();
        if (!r) return acx_fail(ACX_E_NOMEM, "acx_scan_batch: out of memory");
        *result = r;
    }
    if (r->pending) { int rcw = result_complete(r); if (rcw) return rcw; }     // still in flight on its old stream
    r->stream = s; r->n_hay = p->n_hay; r->total = 0; r->host_valid = false;
    r->skip_after = nullptr; r->skip_base = p->dev_index_base;
    r->has_final = p->want_final_state != 0; r->timed = p->timing != 0; r->timed_all = p->timing == 1;

    // position-parallel kernels: ACX_SCAN_ALL on an image that carries the structures, no carried-in state
    // (variant bit 23 turns them off: A/B against the serial walks)
    if (p->mode == ACX_SCAN_ALL && img->ppm_g && !p->dev_init_state && p->n_hay > 0 && !((p->variant >> 23) & 1) &&
        (p->dev_off || p->stride > 0)) {
        const int plan = ppm_plan(img, p);
        if (plan == 1) r->skip_after = p->dev_skip;                 // (k_ppm_scan does not know dev_skip: its records are dropped afterwards)
        if (plan) return scan_ppm(img, p, r, s, plan);
    }
    // iter_long as a position-parallel scan over the dictionary of acx_long.cpp + one sweep (variant bit 25: the serial walk, A/B)
    if (p->mode == ACX_SCAN_LONG && !p->dev_init_state && !p->want_final_state && p->n_hay > 0 && !((p->variant >> 25) & 1) &&
        (p->dev_off || p->stride > 0) && !(p->flags & ACX_SCAN_SKIP_WS)) {
        acx_image* li = image_long(img);
        if (li) { const int rcl = scan_long_ppm(img, li, p, r, s); if (rcl != ACX_LONG_FALLBACK) return rcl; }
    }
    r->skip_after = p->dev_skip;                                      // (neither do the serial walks)

    const size_t n = (size_t)p->n_hay;
    int rc;
    if ((rc = image_ensure_table(img))) return rc;           // (an image with position-parallel structures builds it on first use)

    // ---- work decomposition -----------------------------------------------------------
    // direct : one lane per haystack (fixed-length short reads — config 2/5)
    // chunked: ACX_SCAN_ALL over ragged or long haystacks: fixed-size chunks with a left halo of
    //          longest_word-1 bytes, one lane per chunk (exact for `iter`).
    //          Offsets that live in device memory are never inspected by the host, so every
    //          dev_off batch takes this path: no single lane can be handed a huge haystack.
    const int64_t halo = img->h.longest_word > 0 ? (int64_t)img->h.longest_word - 1 : 0;
    int64_t CH = 0;
    if (p->mode == ACX_SCAN_ALL && p->n_hay > 0 && !((p->variant >> 13) & 1)) {
        int64_t want = 256;                                   // aim for >= ~1M chunks, 256 B .. 4 KiB
        while (want < 4096 && p->hay_capacity / want > ((int64_t)1 << 20)) want <<= 1;
        if (want < 8 * (halo + 1)) want = 8 * (halo + 1);     // keep the halo re-walk <= 1/8 of the work
        if (p->dev_off || p->stride > want) CH = want;
    }
    const bool chunked = CH > 0;
    int64_t n_items = p->n_hay;                               // work items the walk/expand kernels see
    if (chunked) n_items = p->dev_off ? p->n_hay + p->hay_capacity / CH + 1
                                      : p->n_hay * ((p->stride + CH - 1) / CH < 1 ? 1 : (p->stride + CH - 1) / CH);
    const size_t ni = (size_t)n_items;

    if ((rc = r->counts.ensure(ni + 1))) return rc;
    if ((rc = r->nev.ensure(ni + 1))) return rc;
    if ((rc = r->match_off.ensure(n + 1))) return rc;
    if ((rc = r->partials.ensure((size_t)acx_scan_num_partials(n_items > p->n_hay ? n_items : p->n_hay) + 2))) return rc;
    // iter_long: matches do not overlap (the walk restarts behind every match it reports), each is a key, so a haystack
    // of len bytes has at most len / shortest_key + 1: its event slots are packed by that factor (a power of two)
    int ev_shift = 0;
    if (p->mode == ACX_SCAN_LONG && !chunked && img->h.off_ppm && img->ppm.magic == ACX_PPM_MAGIC) {
        const uint32_t m = img->ppm.min_len;
        ev_shift = m >= 16 ? 4 : (m >= 8 ? 3 : (m >= 4 ? 2 : (m >= 2 ? 1 : 0)));
    }
    if ((rc = r->events.ensure(ev_shift ? ((size_t)p->hay_capacity >> ev_shift) + n + 2 : (size_t)p->hay_capacity + 1))) return rc;
    if (r->has_final && (rc = r->final_state.ensure(n + 1))) return rc;
    if ((rc = r->h_total.ensure(1))) return rc;
    if (r->matches.cap == 0 && (rc = r->matches.ensure((size_t)(p->hay_capacity / 8) + 1024))) return rc;
    if (chunked) {
        if ((rc = r->nck.ensure(n + 1))) return rc;
        if ((rc = r->ck_first.ensure(n + 1))) return rc;
        if ((rc = r->ck.ensure(ni + 1))) return rc;
        if ((rc = r->ck_match_off.ensure(ni + 1))) return rc;
    }
    if (r->timed) for (auto& e : r->ev) if (!e) HIP_TRY(hipEventCreate(&e));

    acx_walk_args wa;
    wa.hay = p->dev_hay; wa.hay_cap = p->hay_capacity; wa.off = p->dev_off; wa.stride = p->stride; wa.n_hay = p->n_hay;
    wa.init_state = p->dev_init_state; wa.index_base = p->dev_index_base;
    wa.cls = img->cls; wa.table = img->table; wa.out_off = img->out_off; wa.row_bytes = img->h.n_classes * 4u; wa.state_bits = img->h.state_bits;
    wa.n_states = img->h.n_states;
    wa.counts = r->counts.p; wa.nev = r->nev.p; wa.events = r->events.p;
    wa.final_state = r->has_final ? r->final_state.p : nullptr;
    wa.ev_shift = ev_shift;

    int64_t* item_match_off = chunked ? r->ck_match_off.p : r->match_off.p;
    // implicit top-of-trie kernel: ACX_SCAN_ALL, narrow image that carries the structures, no carried-in
    // state (an arbitrary shallow state id has no k-gram history).  variant bit 16 turns it off (A/B).
    const bool use_itop = p->mode == ACX_SCAN_ALL && img->itop_lds && !p->dev_init_state && !((p->variant >> 16) & 1);

    const bool async = (p->flags & ACX_SCAN_ASYNC) != 0;
    if (r->timed) HIP_TRY(hipEventRecord(r->ev[0], s));
    if (chunked) {
        acx_chunk_args ca;
        ca.off = p->dev_off; ca.stride = p->stride; ca.n_hay = p->n_hay; ca.index_base = p->dev_index_base;
        ca.chunk_bytes = (int32_t)CH; ca.halo = (int32_t)halo;
        ca.nck = r->nck.p; ca.ck_first = r->ck_first.p; ca.ck = r->ck.p;
        // items beyond the real chunk count (the host only knows a bound) must read as empty
        HIP_TRY(hipMemsetAsync(r->counts.p, 0, (ni + 1) * sizeof(int32_t), s));
        HIP_TRY(hipMemsetAsync(r->nev.p, 0, (ni + 1) * sizeof(int32_t), s));
        HIP_TRY(acx_launch_chunk_count(ca, s));
        HIP_TRY(acx_launch_scan(r->nck.p, p->n_hay, r->ck_first.p, r->partials.p, s));
        HIP_TRY(acx_launch_chunk_fill(ca, n_items, s));
        if (use_itop) HIP_TRY(acx_launch_walk_itop(wa, r->ck.p, r->ck_first.p + p->n_hay, n_items, img->h.has_escape != 0,
                                                   img->itop_lds, img->h.itop_lds_bytes / 4, img->itop_entry, img->itop_ebits,
                                                   img->itop_cells, img->tflags, img->h.itop_cell_bytes, img->h.itop_flags, (int)((p->variant >> 17) & 0x3f), s));
        else          HIP_TRY(acx_launch_walk_chunks(wa, r->ck.p, r->ck_first.p + p->n_hay, n_items, img->h.has_escape != 0, s));
    } else if (p->mode == ACX_SCAN_ALL) {
        if (use_itop) HIP_TRY(acx_launch_walk_itop(wa, nullptr, nullptr, p->n_hay, img->h.has_escape != 0,
                                                   img->itop_lds, img->h.itop_lds_bytes / 4, img->itop_entry, img->itop_ebits,
                                                   img->itop_cells, img->tflags, img->h.itop_cell_bytes, img->h.itop_flags, (int)((p->variant >> 17) & 0x3f), s));
        else          HIP_TRY(acx_launch_walk_all(wa, img->h.has_escape != 0, p->variant, s));
    } else {
        HIP_TRY(acx_launch_walk_long(wa, p->variant, s));
    }
    if (r->timed) HIP_TRY(hipEventRecord(r->ev[1], s));
    HIP_TRY(acx_launch_scan(r->counts.p, n_items, item_match_off, r->partials.p, s));
    if (r->timed_all) HIP_TRY(hipEventRecord(r->ev[2], s));

    acx_expand_args ea;
    ea.off = p->dev_off; ea.stride = p->stride; ea.n_hay = n_items; ea.nev = r->nev.p; ea.events = r->events.p;
    ea.match_off = item_match_off; ea.out_off = img->out_off; ea.out_val = img->out_val; ea.first_val = img->first_val;
    ea.long_mode = p->mode == ACX_SCAN_LONG ? 1 : 0; ea.state_bits = img->h.state_bits; ea.ev_shift = ev_shift;
    ea.ck = chunked ? r->ck.p : nullptr; ea.n_items_dev = nullptr;
    // Speculative launch with the capacity we already have: no host round trip between
    // scan and expand in the steady state.  The kernel is a no-op if the total does not fit.
    ea.matches = r->matches.p; ea.capacity = (int64_t)r->matches.cap;
    HIP_TRY(acx_launch_expand(ea, p->variant, s));
    if (chunked) HIP_TRY(acx_launch_hay_offsets(r->ck_first.p, r->ck_match_off.p, p->n_hay, r->match_off.p, s));
    if (r->timed_all) HIP_TRY(hipEventRecord(r->ev[3], s));
    HIP_TRY(hipMemcpyAsync(r->h_total.p, item_match_off + n_items, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    if (!r->done) HIP_TRY(hipEventCreateWithFlags(&r->done, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(r->done, s));
    r->pending = true; r->pend_ea = ea; r->pend_variant = p->variant;
    if (async) return ACX_OK;
    return result_complete(r);
}

// ACX_SCAN_SKIP_WS: compact the batch (acx_ws.hip), scan the compacted one, map the records back when the scan completes
static int scan_batch_ws(acx_image_t* img, const acx_scan_params* p, acx_result* r, hipStream_t s) {
    const int64_t total = p->dev_off ? p->hay_capacity : p->n_hay * p->stride;
    if (!p->dev_off && (p->stride < 0 || p->n_hay * p->stride > p->hay_capacity)) return acx_fail(ACX_E_INVAL, "acx_scan_batch: n_hay*stride exceeds hay_capacity");
    if (total > 0xFFFFFFFFll) return acx_fail(ACX_E_UNSUPPORTED, "acx_scan_batch: ACX_SCAN_SKIP_WS takes batches below 4 GiB");
    if (total > 0 && !p->dev_hay) return acx_fail(ACX_E_INVAL, "acx_scan_batch: dev_hay is NULL");
    const size_t n = (size_t)p->n_hay;
    const int64_t nt = acx_ws_num_tiles(total);
    int rc;
    if ((rc = r->ws_hay.ensure((size_t)total + 64))) return rc;
    if ((rc = r->ws_map.ensure((size_t)total + 1))) return rc;
    if ((rc = r->ws_cnt.ensure((size_t)nt + 1))) return rc;
    if ((rc = r->ws_tile_off.ensure((size_t)nt + 2))) return rc;
    if ((rc = r->ws_partials.ensure((size_t)acx_scan_num_partials(nt) + 2))) return rc;
    if ((rc = r->ws_off.ensure(n + 2))) return rc;
    if (p->dev_skip && (rc = r->ws_skip.ensure(n + 1))) return rc;
    if (nt > 0) {
        HIP_TRY(acx_launch_ws_count(p->dev_hay, total, r->ws_cnt.p, s));
        HIP_TRY(acx_launch_scan(r->ws_cnt.p, nt, r->ws_tile_off.p, r->ws_partials.p, s));
        HIP_TRY(acx_launch_ws_move(p->dev_hay, total, r->ws_tile_off.p, r->ws_hay.p, r->ws_map.p, s));
    } else HIP_TRY(hipMemsetAsync(r->ws_tile_off.p, 0, sizeof(int64_t), s));
    HIP_TRY(acx_launch_ws_offsets(p->dev_off, p->stride, p->n_hay, p->dev_skip, r->ws_map.p, r->ws_tile_off.p + nt, r->ws_off.p,
                                  p->dev_skip ? r->ws_skip.p : nullptr, s));
    r->ws_o_off = p->dev_off; r->ws_o_stride = p->stride; r->ws_o_skip = p->dev_skip; r->ws_o_base = p->dev_index_base;
    acx_scan_params q = *p;
    q.flags &= ~(int32_t)ACX_SCAN_SKIP_WS;
    q.dev_hay = r->ws_hay.p; q.hay_capacity = total; q.dev_off = r->ws_off.p; q.stride = 0;
    q.dev_index_base = nullptr; q.dev_skip = p->dev_skip ? r->ws_skip.p : nullptr;
    // what is left of a haystack may be shorter than what the caller promised for the whole of it; the stream kernel
    // notices a broken promise and the batch is scanned again on the general kernels (acx.h), so the promise stays
    q.min_hay_len = (p->dev_off ? p->min_hay_len : (p->stride > INT32_MAX ? INT32_MAX : (int32_t)p->stride)) >= 8 ? 8 : 0;
    r->ws_active = true;
    acx_result* self = r;
    rc = scan_batch_inner(img, &q, &self, (void*)s);
    if (rc) r->ws_active = false;
    return rc;
}

extern "C" int acx_scan_batch(acx_image_t* img, const acx_scan_params* p, acx_result_t** result, void* stream_v) {
    if (!img || !p || !result) return acx_fail(ACX_E_INVAL, "acx_scan_batch: NULL argument");
    if (p->struct_bytes != sizeof(acx_scan_params))
        return acx_fail(ACX_E_INVAL, "acx_scan_batch: params.struct_bytes = %u, library expects %zu", p->struct_bytes, sizeof(acx_scan_params));
    if (*result && (*result)->pending) { int rcw = result_complete(*result); if (rcw) return rcw; }     // still in flight on its old stream
    if (*result) { (*result)->ws_active = false; (*result)->host_walk = false; }
    if (!(p->flags & ACX_SCAN_SKIP_WS) || p->n_hay <= 0) return scan_batch_inner(img, p, result, stream_v);
    if (p->n_hay < 0 || p->hay_capacity < 0) return acx_fail(ACX_E_INVAL, "acx_scan_batch: negative size");
    if (p->hay_capacity > ACX_MAX_LAUNCH_BYTES)
        return acx_fail(ACX_E_UNSUPPORTED, "acx_scan_batch: %lld haystack bytes in one call; split the batch into calls of <= %lld bytes",
                        (long long)p->hay_capacity, (long long)ACX_MAX_LAUNCH_BYTES);
    if (!*result) {
        *result = new (std::nothrow) acx_result();
        if (!*result) return acx_fail(ACX_E_NOMEM, "acx_scan_batch: out of memory");
    }
    (*result)->stream = (hipStream_t)stream_v;
    return scan_batch_ws(img, p, *result, (hipStream_t)stream_v);
}

extern "C" int64_t acx_result_num_matches(acx_result_t* r) {
    if (r && r->host_walk) return (int64_t)r->hw_m.size();
    return (r && result_complete(r) == ACX_OK) ? r->total : 0;
}
// (the result of a host walk has no device side: acx_result_fetch_host is its accessor)
static bool no_dev_side(acx_result_t* r) { if (r && r->host_walk) { (void)acx_fail(ACX_E_STATE, "the result of acx_trie_scan_host has no device buffers: use acx_result_fetch_host"); return true; } return false; }
extern "C" const int64_t* acx_result_offsets_dev(acx_result_t* r) { return (r && !no_dev_side(r) && result_complete(r) == ACX_OK) ? r->match_off.p : nullptr; }
extern "C" const acx_match_t* acx_result_matches_dev(acx_result_t* r) {
    return (r && !no_dev_side(r) && result_complete(r) == ACX_OK) ? (const acx_match_t*)r->matches.p : nullptr;
}
extern "C" const int32_t* acx_result_final_state_dev(acx_result_t* r) {
    return (r && !no_dev_side(r) && r->has_final && result_complete(r) == ACX_OK) ? r->final_state.p : nullptr;
}

extern "C" int acx_result_fetch_host(acx_result_t* r, const int64_t** off, const acx_match_t** matches, const int32_t** final_state) {
    if (!r) return acx_fail(ACX_E_INVAL, "acx_result_fetch_host: NULL result");
    if (r->host_walk) {
        if (off) *off = r->hw_off.data();
        if (matches) *matches = r->hw_m.data();
        if (final_state) *final_state = r->has_final ? r->hw_fin.data() : nullptr;
        return ACX_OK;
    }
    { int rcw = result_complete(r); if (rcw) return rcw; }
    if (!r->host_valid) {
        int rc;
        const size_t n = (size_t)r->n_hay;
        if ((rc = r->h_off.ensure(n + 1))) return rc;
        if ((rc = r->h_matches.ensure((size_t)r->total + 1))) return rc;
        HIP_TRY(hipMemcpyAsync(r->h_off.p, r->match_off.p, (n + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, r->stream));
        if (r->total) HIP_TRY(hipMemcpyAsync(r->h_matches.p, r->matches.p, (size_t)r->total * sizeof(acx_match_t), hipMemcpyDeviceToHost, r->stream));
        if (r->has_final) {
            if ((rc = r->h_final.ensure(n + 1))) return rc;
            if (n) HIP_TRY(hipMemcpyAsync(r->h_final.p, r->final_state.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, r->stream));
        }
        HIP_TRY(hipStreamSynchronize(r->stream));
        r->host_valid = true;
    }
    if (off) *off = r->h_off.p;
    if (matches) *matches = r->h_matches.p;
    if (final_state) *final_state = r->has_final ? r->h_final.p : nullptr;
    return ACX_OK;
}

extern "C" int acx_result_timing(acx_result_t* r, float* walk_ms, float* scan_ms, float* expand_ms, float* total_ms) {
    if (!r) return acx_fail(ACX_E_INVAL, "acx_result_timing: NULL result");
    if (r->host_walk) return acx_fail(ACX_E_STATE, "acx_result_timing: the result of a host walk has no kernel timing");
    { int rcw = result_complete(r); if (rcw) return rcw; }
    if (!r->timed) return acx_fail(ACX_E_STATE, "acx_result_timing: the last scan was not run with params.timing = 1");
    if (walk_ms) *walk_ms = r->t_walk;
    if (scan_ms) *scan_ms = r->t_scan;
    if (expand_ms) *expand_ms = r->t_expand;
    if (total_ms) *total_ms = r->t_total;
    return ACX_OK;
}

// A fixed-stride batch, host to host, as a pipeline of groups of haystacks: the copy engine uploads group k + 1 while
// group k is scanned and its gather writes records and offsets STRAIGHT into the result's pinned host buffers (the
// kernel stores go over PCIe themselves).  Why this shape: on this box two copies in opposite directions crawl (15 GB/s
// each way), a kernel that writes pinned host memory beside an upload does not (43 GB/s each way;
// profiles/r3_pcie_probe.txt).  Returns ACX_HOST_RETRY when it does not apply (not the stream kernel's batch, or the
// records outgrew the host buffer: the staged path grows it); the result is then untouched as far as callers see.
#ifdef ACX_HOST_TRACE             // development builds: where a host-to-host call spends its time (stderr)
#include <chrono>
static double host_now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double g_host_t0 = 0;
#define HOST_MARK(...) do { fprintf(stderr, "[host %8.3f] ", host_now_ms() - g_host_t0); fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); } while (0)
#else
#define HOST_MARK(...) do { } while (0)
#endif

// `verify`: the caller's offsets when they have only been SAMPLED so far (scan_host_impl): every one of them is compared
// with h * L while the first group travels, and a batch that turns out not to be equally spaced goes back untouched.
static int scan_host_pipelined(acx_image_t* img, const uint8_t* hay, int64_t n_hay, int64_t L, const int32_t* index_base,
                               acx_result* r, int32_t flags, const int64_t* verify) {
    const int64_t total_bytes = n_hay * L;
    if (flags || total_bytes < ((int64_t)16 << 20) || n_hay < 4096) return ACX_HOST_RETRY;
    // Groups: the uploads are the bottleneck (they run back to back on the copy engine), every scan + gather hides behind the
    // next group's upload, and what does not hide is the LAST group's scan + gather: many small groups make that tail short,
    // few large ones keep the number of waits down.  About 16 MB each, 2 .. 12.
    int G = (int)(total_bytes >> 24);
    G = G < 2 ? 2 : (G > 12 ? 12 : G);
    if (const char* e = acx_tune_env("ACX_HOST_GROUPS")) { const int v = atoi(e); if (v >= 1 && v <= 64) G = v; }   // tuning hook
    int rc;
    if ((rc = r->in_hay.ensure((size_t)total_bytes + 64))) return rc;
    {
        acx_scan_params q;
        memset(&q, 0, sizeof q);
        q.struct_bytes = sizeof q; q.mode = ACX_SCAN_ALL; q.dev_hay = r->in_hay.p; q.hay_capacity = total_bytes / G; q.stride = L; q.n_hay = n_hay / G;
        if (ppm_plan(img, &q) != 2) return ACX_HOST_RETRY;
    }
    if (r->pending) { int rcw = result_complete(r); if (rcw) return rcw; }
    r->host_walk = false;
    if ((rc = r->h_off.ensure((size_t)n_hay + 2))) return rc;
    {   // room for the records: what earlier calls needed, or one per eight bytes
        const size_t want = (size_t)(total_bytes / 8) + 1024;
        if (r->h_matches.cap < want && (rc = r->h_matches.ensure(want))) return rc;
    }
    if (index_base) {
        if ((rc = r->in_base.ensure((size_t)n_hay + 1))) return rc;
        HIP_TRY(hipMemcpy(r->in_base.p, index_base, (size_t)n_hay * 4, hipMemcpyHostToDevice));
    }
    if (!r->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&r->copy_stream, hipStreamNonBlocking));
    HOST_MARK("pipelined: %d groups, buffers ready", G);
    void* d_m = nullptr; void* d_o = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&d_m, r->h_matches.p, 0));
    HIP_TRY(hipHostGetDevicePointer(&d_o, r->h_off.p, 0));
    auto first = [&](int g) { return g >= G ? n_hay : (n_hay * g / G) & ~(int64_t)15; };   // (a group starts 16-byte aligned: 16 haystacks of any length)
    // The uploads run on a thread of their own: the caller's buffer is pageable, and a copy from pageable memory keeps
    // the calling thread until it is done — queued from this thread they would all be over before the first scan starts.
    std::atomic<int> uploaded{0}, up_err{0}, up_stop{0};
    int dev_id = 0;
    HIP_TRY(hipGetDevice(&dev_id));
    std::thread uploader([&] {
        if (hipSetDevice(dev_id) != hipSuccess) { up_err.store(1); return; }
        for (int g = 0; g < G && !up_stop.load(std::memory_order_relaxed); g++) {
            const int64_t h0 = first(g), h1 = first(g + 1);
            hipError_t e = hipMemcpyAsync(r->in_hay.p + h0 * L, hay + h0 * L, (size_t)((h1 - h0) * L), hipMemcpyHostToDevice, r->copy_stream);
            if (e == hipSuccess) e = hipStreamSynchronize(r->copy_stream);
            if (e != hipSuccess) { up_err.store(1); return; }
            HOST_MARK("  upload %d done", g);
            uploaded.store(g + 1, std::memory_order_release);
        }
    });
    struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{uploader};
    if (verify) {
        // (one pass the compiler can vectorise, hidden behind the first upload; a million offsets took 1.2 ms of every call
        //  when two scalar passes looked at them before anything was sent)
        uint64_t bad = 0;
        for (int64_t h = 0; h < n_hay; h++) bad |= (uint64_t)((verify[h + 1] - verify[h]) ^ L);
        HOST_MARK("  offsets verified (%s)", bad ? "NOT equally spaced" : "equally spaced");
        if (bad) { up_stop.store(1); uploader.join(); return ACX_HOST_RETRY; }
    }
    int64_t done = 0;
    int out = ACX_OK;
    for (int g = 0; g < G && out == ACX_OK; g++) {
        const int64_t h0 = first(g), h1 = first(g + 1);
        while (uploaded.load(std::memory_order_acquire) <= g && !up_err.load()) std::this_thread::yield();
        if (up_err.load()) { out = acx_fail(ACX_E_HIP, "acx_scan_host: the upload of a group failed"); break; }
        acx_scan_params q;
        memset(&q, 0, sizeof q);
        q.struct_bytes = sizeof q; q.mode = ACX_SCAN_ALL;
        q.dev_hay = r->in_hay.p + h0 * L; q.hay_capacity = (h1 - h0) * L; q.stride = L; q.n_hay = h1 - h0;
        q.dev_index_base = index_base ? r->in_base.p + h0 : nullptr;
        r->ext_matches = (uint2*)d_m + done; r->ext_capacity = (int64_t)r->h_matches.cap - done; r->ext_match_off = (int64_t*)d_o + h0;
        r->ext_off_base = done;                                        // (the gather writes offsets that count from the batch's first record)
        acx_result* self = r;
        HOST_MARK("  scan %d issued", g);
        out = scan_batch_inner(img, &q, &self, nullptr);              // (synchronous: the gather has written when it returns)
        HOST_MARK("  scan %d complete (%lld records)", g, (long long)r->total);
        if (out == ACX_OK && !(r->ppm_stream && r->ppm_self)) out = ACX_HOST_RETRY;      // (not the kernels this path is for)
        if (out != ACX_OK) break;
        done += r->total;
    }
    r->ext_matches = nullptr; r->ext_capacity = 0; r->ext_match_off = nullptr; r->ext_off_base = 0;
    if (out != ACX_OK) up_stop.store(1);
    uploader.join();                                                   // (the uploads read the caller's buffer)
    if (out != ACX_OK) {
        if (out == ACX_HOST_RETRY) (void)r->h_matches.ensure(r->h_matches.cap * 2);     // the staged path would need it as well
        return out;
    }
    r->h_off.p[n_hay] = done;
    r->n_hay = n_hay; r->total = done; r->has_final = false; r->host_valid = true;
    HOST_MARK("pipelined: done");
    return ACX_OK;
}

// one group of haystacks that fits a launch: H2D, scan; `off` starts at 0
static int scan_host_once(acx_image_t* img, int mode, const uint8_t* hay, const int64_t* off, int64_t n_hay,
                          const int32_t* init_state, const int32_t* index_base, acx_result_t** result, int want_final, int32_t flags) {
    acx_result* r = *result;
    const int64_t total_bytes = off[n_hay];
    int rc;
    // one pass over the offsets: the shortest haystack (what the stream kernel is promised), and whether they are all
    // equally long — then the batch is a fixed-stride one: no offsets travel, and the scan takes the faster layout
    int64_t shortest = INT32_MAX;
    bool uniform = n_hay > 0;
    const int64_t L0 = n_hay > 0 ? off[1] - off[0] : 0;
    for (int64_t h = 0; h < n_hay; h++) {
        const int64_t l = off[h + 1] - off[h];
        if (l < shortest) shortest = l;
        uniform = uniform && l == L0;
    }
    const bool as_stride = uniform && L0 > 0 && L0 <= INT32_MAX;
    HOST_MARK("offsets looked at twice");
    if (as_stride && mode == ACX_SCAN_ALL && !init_state && !want_final) {
        rc = scan_host_pipelined(img, hay, n_hay, L0, index_base, r, flags, nullptr);
        if (rc != ACX_HOST_RETRY) return rc;
    }
    if ((rc = r->in_hay.ensure((size_t)total_bytes + 64))) return rc;
    if (total_bytes) HIP_TRY(hipMemcpyAsync(r->in_hay.p, hay, (size_t)total_bytes, hipMemcpyHostToDevice, nullptr));
    if (!as_stride) {
        if ((rc = r->in_off.ensure((size_t)n_hay + 1))) return rc;
        HIP_TRY(hipMemcpyAsync(r->in_off.p, off, ((size_t)n_hay + 1) * sizeof(int64_t), hipMemcpyHostToDevice, nullptr));
    }
    if (init_state) {
        if ((rc = r->in_init.ensure((size_t)n_hay + 1))) return rc;
        if (n_hay) HIP_TRY(hipMemcpyAsync(r->in_init.p, init_state, (size_t)n_hay * 4, hipMemcpyHostToDevice, nullptr));
    }
    if (index_base) {
        if ((rc = r->in_base.ensure((size_t)n_hay + 1))) return rc;
        if (n_hay) HIP_TRY(hipMemcpyAsync(r->in_base.p, index_base, (size_t)n_hay * 4, hipMemcpyHostToDevice, nullptr));
    }
    acx_scan_params p;
    memset(&p, 0, sizeof p);
    p.struct_bytes = sizeof p; p.mode = mode;
    p.dev_hay = r->in_hay.p; p.hay_capacity = total_bytes; p.n_hay = n_hay;
    if (as_stride) { p.dev_off = nullptr; p.stride = L0; } else { p.dev_off = r->in_off.p; p.stride = 0; }
    p.dev_init_state = init_state ? r->in_init.p : nullptr;
    p.dev_index_base = index_base ? r->in_base.p : nullptr;
    p.want_final_state = want_final;
    p.min_hay_len = n_hay > 0 ? (int32_t)shortest : 0;
    p.flags = flags;
    return acx_scan_batch(img, &p, result, nullptr);
}

static int scan_host_impl(acx_image_t* img, int mode, const uint8_t* hay, const int64_t* off, int64_t n_hay,
                          const int32_t* init_state, const int32_t* index_base, acx_result_t** result, int want_final, int32_t flags);

// streams: context and chunk of every haystack staged side by side; the context's records are never reported
extern "C" int acx_scan_host_ctx(acx_image_t* img, const uint8_t* hay, const int64_t* off, int64_t n_hay,
                                 const uint8_t* ctx, const int64_t* ctx_off, const int32_t* index_base, int32_t flags,
                                 acx_result_t** result) {
    if (!img || !off || !result || n_hay < 0) return acx_fail(ACX_E_INVAL, "acx_scan_host_ctx: bad argument");
    if (flags & ~(int32_t)ACX_SCAN_SKIP_WS) return acx_fail(ACX_E_INVAL, "acx_scan_host_ctx: flags = %d (ACX_SCAN_SKIP_WS or 0)", flags);
    if (!ctx || !ctx_off) return scan_host_impl(img, ACX_SCAN_ALL, hay, off, n_hay, nullptr, index_base, result, 0, flags);
    if (off[0] != 0 || ctx_off[0] != 0) return acx_fail(ACX_E_INVAL, "acx_scan_host_ctx: off[0] and ctx_off[0] must be 0");
    int64_t total = 0, shortest = INT32_MAX;
    for (int64_t h = 0; h < n_hay; h++) {
        const int64_t l = off[h + 1] - off[h], c = ctx_off[h + 1] - ctx_off[h];
        if (l < 0 || c < 0 || l + c > INT32_MAX) return acx_fail(ACX_E_INVAL, "acx_scan_host_ctx: bad lengths at haystack %lld", (long long)h);
        total += l + c;
        if (l + c < shortest) shortest = l + c;
    }
    if (off[n_hay] > 0 && !hay) return acx_fail(ACX_E_INVAL, "acx_scan_host_ctx: hay is NULL");
    if (total > ACX_MAX_LAUNCH_BYTES) return acx_fail(ACX_E_UNSUPPORTED, "acx_scan_host_ctx: %lld bytes in one call; split the batch", (long long)total);
    acx_result* r = *result;
    if (!r) {
        r = new (std::nothrow) acx_result();
        if (!r) return acx_fail(ACX_E_NOMEM, "acx_scan_host_ctx: out of memory");
        *result = r;
    }
    if (r->pending) { int rcw = result_complete(r); if (rcw) return rcw; }
    int rc;
    if ((rc = r->h_stage.ensure((size_t)total + 64))) return rc;
    if ((rc = r->h_off.ensure((size_t)n_hay + 1))) return rc;           // (reused as staging for the new offsets; fetch_host rewrites it)
    if ((rc = r->h_final.ensure((size_t)n_hay + 1))) return rc;         // (staging for the context lengths)
    int64_t at = 0;
    for (int64_t h = 0; h < n_hay; h++) {
        const int64_t l = off[h + 1] - off[h], c = ctx_off[h + 1] - ctx_off[h];
        r->h_off.p[h] = at;
        r->h_final.p[h] = (int32_t)c;
        if (c) memcpy(r->h_stage.p + at, ctx + ctx_off[h], (size_t)c);
        if (l) memcpy(r->h_stage.p + at + c, hay + off[h], (size_t)l);
        at += l + c;
    }
    r->h_off.p[n_hay] = at;
    if ((rc = r->in_hay.ensure((size_t)total + 64))) return rc;
    if ((rc = r->in_off.ensure((size_t)n_hay + 1))) return rc;
    if ((rc = r->in_skip.ensure((size_t)n_hay + 1))) return rc;
    if (total) HIP_TRY(hipMemcpy(r->in_hay.p, r->h_stage.p, (size_t)total, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(r->in_off.p, r->h_off.p, ((size_t)n_hay + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    if (n_hay) HIP_TRY(hipMemcpy(r->in_skip.p, r->h_final.p, (size_t)n_hay * 4, hipMemcpyHostToDevice));
    if (index_base) {
        if ((rc = r->in_base.ensure((size_t)n_hay + 1))) return rc;
        if (n_hay) HIP_TRY(hipMemcpy(r->in_base.p, index_base, (size_t)n_hay * 4, hipMemcpyHostToDevice));
    }
    acx_scan_params p;
    memset(&p, 0, sizeof p);
    p.struct_bytes = sizeof p; p.mode = ACX_SCAN_ALL;
    p.dev_hay = r->in_hay.p; p.hay_capacity = total; p.dev_off = r->in_off.p; p.stride = 0; p.n_hay = n_hay;
    p.dev_index_base = index_base ? r->in_base.p : nullptr;
    p.dev_skip = r->in_skip.p;
    p.min_hay_len = n_hay > 0 ? (int32_t)shortest : 0;
    p.flags = flags;
    return acx_scan_batch(img, &p, result, nullptr);
}

// ---- the walk over the host trie (acx_hostwalk.cpp): BASELINE config 1, processes without a device, haystacks below the launch crossover ----
int acxi_hostwalk_batch(const acx_trie_t* t, int mode, const uint8_t* hay, const int64_t* off, int64_t n_hay,
                       const uint8_t* ctx, const int64_t* ctx_off, const int32_t* init_node, const int32_t* index_base,
                       int32_t flags, std::vector<int64_t>* moff, std::vector<acx_match_t>* m, std::vector<int32_t>* fin);
static std::atomic<int64_t> g_host_walk_bytes{ACX_HOST_WALK_DEFAULT_BYTES};
static std::atomic<int64_t> g_host_walk_calls{0};
extern "C" void acx_set_host_walk_bytes(int64_t bytes) { g_host_walk_bytes.store(bytes < 0 ? -1 : bytes); }
extern "C" int64_t acx_host_walk_bytes(void) { return g_host_walk_bytes.load(); }
extern "C" int64_t acx_host_walk_calls(void) { return g_host_walk_calls.load(); }
extern "C" int acx_host_walk_applies(int64_t total_bytes) {
    const int64_t lim = g_host_walk_bytes.load();
    if (lim < 0 || total_bytes < 0 || total_bytes > ACX_HOSTWALK_MAX_BYTES) return 0;
    static const int n_dev = [] { int c = 0; return hipGetDeviceCount(&c) == hipSuccess ? c : 0; }();
    return (n_dev == 0 || total_bytes <= lim) ? 1 : 0;
}
extern "C" int acx_trie_scan_host(const acx_trie_t* t, int mode, const uint8_t* hay, const int64_t* off, int64_t n_hay,
                                  const uint8_t* ctx, const int64_t* ctx_off, const int32_t* init_node, const int32_t* index_base,
                                  int32_t flags, int want_final, acx_result_t** result) {
    if (!t || !off || !result || n_hay < 0) return acx_fail(ACX_E_INVAL, "acx_trie_scan_host: bad argument");
    if (off[0] != 0 || (ctx && (!ctx_off || ctx_off[0] != 0))) return acx_fail(ACX_E_INVAL, "acx_trie_scan_host: off[0] and ctx_off[0] must be 0");
    // (a stream that carries a node of the HOST trie — iter_long().set() behind a chunk that the walk took — goes on here whatever the
    //  chunk's size: the state means nothing to the device image, and refusing the chunk would break a stream that has begun)
    bool carried = false;
    if (init_node && mode == ACX_SCAN_LONG) for (int64_t i = 0; i < n_hay; i++) if (init_node[i] < 0) { carried = true; break; }
    if (off[n_hay] > ACX_HOSTWALK_MAX_BYTES && !carried)
        return acx_fail(ACX_E_UNSUPPORTED, "acx_trie_scan_host: %lld bytes: the host walk is for what does not pay a launch (<= %lld bytes); batches go to acx_scan_host",
                        (long long)off[n_hay], (long long)ACX_HOSTWALK_MAX_BYTES);
    acx_result* r = *result;
    if (!r) {
        r = new (std::nothrow) acx_result();
        if (!r) return acx_fail(ACX_E_NOMEM, "acx_trie_scan_host: out of memory");
        *result = r;
    }
    if (r->pending) { int rcw = result_complete(r); if (rcw) return rcw; }
    r->host_walk = false; r->host_valid = false;
    const bool fin = mode == ACX_SCAN_LONG && want_final;
    int rc = acxi_hostwalk_batch(t, mode, hay, off, n_hay, ctx, ctx_off, init_node, index_base, flags, &r->hw_off, &r->hw_m, fin ? &r->hw_fin : nullptr);
    if (rc) return rc;
    r->host_walk = true; r->n_hay = n_hay; r->total = (int64_t)r->hw_m.size(); r->has_final = fin && n_hay > 0;
    g_host_walk_calls.fetch_add(1);
    return ACX_OK;
}

static std::atomic<int64_t> g_host_group_bytes{0};
extern "C" void acx_set_host_group_bytes(int64_t bytes) { g_host_group_bytes.store(bytes > 0 ? bytes : 0); }
static int64_t max_launch_bytes() {
    const int64_t x = g_host_group_bytes.load();
    return x > 0 && x < ACX_MAX_LAUNCH_BYTES ? x : ACX_MAX_LAUNCH_BYTES;
}

extern "C" int acx_scan_host(acx_image_t* img, int mode, const uint8_t* hay, const int64_t* off, int64_t n_hay,
                             const int32_t* init_state, const int32_t* index_base, acx_result_t** result) {
    return scan_host_impl(img, mode, hay, off, n_hay, init_state, index_base, result, 1, 0);
}
extern "C" int acx_scan_host_nofinal(acx_image_t* img, int mode, const uint8_t* hay, const int64_t* off, int64_t n_hay,
                                     const int32_t* index_base, acx_result_t** result) {
    return scan_host_impl(img, mode, hay, off, n_hay, nullptr, index_base, result, 0, 0);
}

// want_final = 0: no final states (an image with the position-parallel structures then never builds its dense table
// for ACX_SCAN_ALL: acx_scan_host_ctx, what the iterators and find_all call)
static int scan_host_impl(acx_image_t* img, int mode, const uint8_t* hay, const int64_t* off, int64_t n_hay,
                          const int32_t* init_state, const int32_t* index_base, acx_result_t** result, int want_final, int32_t flags) {
    if (!img || !off || !result || n_hay < 0) return acx_fail(ACX_E_INVAL, "acx_scan_host: bad argument");
    if (off[0] != 0) return acx_fail(ACX_E_INVAL, "acx_scan_host: off[0] must be 0");
#ifdef ACX_HOST_TRACE
    g_host_t0 = host_now_ms();
#endif
    HOST_MARK("scan_host_impl: %lld haystacks", (long long)n_hay);
    // A large batch whose offsets LOOK equally spaced (both ends and a few in between) goes to the pipeline of groups at
    // once: it compares every offset while the first group is on its way, and hands the batch back if one differs.
    if (mode == ACX_SCAN_ALL && !init_state && !want_final && !flags && n_hay >= 4096 && hay) {
        const int64_t L0 = off[1] - off[0];
        bool looks = L0 > 0 && L0 <= INT32_MAX && off[n_hay] == n_hay * L0 && off[n_hay] <= max_launch_bytes();
        for (int64_t k = 1; k < 16 && looks; k++) { const int64_t h = n_hay / 16 * k; looks = off[h] == h * L0; }
        if (looks) {
            acx_result* r0 = *result;
            if (!r0) {
                r0 = new (std::nothrow) acx_result();
                if (!r0) return acx_fail(ACX_E_NOMEM, "acx_scan_host: out of memory");
                *result = r0;
            }
            const int rc0 = scan_host_pipelined(img, hay, n_hay, L0, index_base, r0, flags, off);
            if (rc0 != ACX_HOST_RETRY) return rc0;
        }
    }
    {   // (one branch-free pass: a batch of a million reads spends as long here as in its scan kernel otherwise)
        int64_t lo = 0, hi = 0;
        for (int64_t h = 0; h < n_hay; h++) { const int64_t l = off[h + 1] - off[h]; lo = l < lo ? l : lo; hi = l > hi ? l : hi; }
        if (lo < 0) return acx_fail(ACX_E_INVAL, "acx_scan_host: offsets not monotone");
        if (hi > INT32_MAX) return acx_fail(ACX_E_INVAL, "acx_scan_host: a haystack longer than INT_MAX");
    }
    const int64_t total_bytes = off[n_hay];
    if (total_bytes > 0 && !hay) return acx_fail(ACX_E_INVAL, "acx_scan_host: hay is NULL");
    acx_result* r = *result;
    if (!r) {
        r = new (std::nothrow) acx_result();
        if (!r) return acx_fail(ACX_E_NOMEM, "acx_scan_host: out of memory");
        *result = r;
    }
    r->host_walk = false;
    int64_t limit = max_launch_bytes();
    if ((flags & ACX_SCAN_SKIP_WS) && limit > 0xFFFFFFFFll) limit = 0xFFFFFFFFll;     // (positions of the compacted batch's map are 32-bit)
    if (total_bytes <= limit) return scan_host_once(img, mode, hay, off, n_hay, init_state, index_base, result, want_final, flags);

    // More than one launch can stage (8 B of event scratch per haystack byte): scan groups of whole
    // haystacks one after the other and assemble the host-side result; the device-side accessors then
    // only see the last group, acx_result_fetch_host sees everything.
    int rc;
    std::vector<int64_t> all_off;
    std::vector<acx_match_t> all_m;
    std::vector<int32_t> all_fin;
    std::vector<int64_t> goff;
    try {
        all_off.reserve((size_t)n_hay + 1);
        all_off.push_back(0);
        for (int64_t g0 = 0; g0 < n_hay;) {
            int64_t g1 = g0 + 1;                                      // at least one haystack per group (each is < 2 GiB)
            while (g1 < n_hay && off[g1 + 1] - off[g0] <= limit) g1++;
            const int64_t gn = g1 - g0;
            goff.resize((size_t)gn + 1);
            for (int64_t k = 0; k <= gn; k++) goff[(size_t)k] = off[g0 + k] - off[g0];
            if (goff[(size_t)gn] > ACX_MAX_LAUNCH_BYTES)
                return acx_fail(ACX_E_UNSUPPORTED, "acx_scan_host: haystack %lld alone exceeds one launch", (long long)g0);
            rc = scan_host_once(img, mode, hay + off[g0], goff.data(), gn, init_state ? init_state + g0 : nullptr,
                                index_base ? index_base + g0 : nullptr, result, want_final, flags);
            if (rc) return rc;
            const int64_t* moff; const acx_match_t* m; const int32_t* fin;
            if ((rc = acx_result_fetch_host(r, &moff, &m, &fin))) return rc;
            const int64_t base = all_off.back();
            for (int64_t k = 1; k <= gn; k++) all_off.push_back(base + moff[k]);
            all_m.insert(all_m.end(), m, m + moff[gn]);
            if (fin) all_fin.insert(all_fin.end(), fin, fin + gn);
            g0 = g1;
        }
    } catch (const std::bad_alloc&) {
        return acx_fail(ACX_E_NOMEM, "acx_scan_host: out of memory");
    }
    if ((rc = r->h_off.ensure((size_t)n_hay + 1))) return rc;
    if ((rc = r->h_matches.ensure(all_m.size() + 1))) return rc;
    memcpy(r->h_off.p, all_off.data(), all_off.size() * sizeof(int64_t));
    if (!all_m.empty()) memcpy(r->h_matches.p, all_m.data(), all_m.size() * sizeof(acx_match_t));
    r->has_final = (int64_t)all_fin.size() == n_hay && n_hay > 0;
    if (r->has_final) {
        if ((rc = r->h_final.ensure((size_t)n_hay + 1))) return rc;
        memcpy(r->h_final.p, all_fin.data(), all_fin.size() * sizeof(int32_t));
    }
    r->n_hay = n_hay; r->total = (int64_t)all_m.size(); r->host_valid = true;
    return ACX_OK;
}
