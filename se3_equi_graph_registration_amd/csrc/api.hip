// C-ABI glue that is not a kernel: error text, ABI version, parameter-block layout queries.
#include "common.h"

#include <atomic>
#include <stdlib.h>
#include <string.h>

namespace eq {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// Chunk size of the 16-edge kernels (egnn_mfma16.h): 64 edges amortise the one-lane-per-edge front over four sub-tiles; small
// problems take smaller chunks so that every SIMD gets work.  equigspr_set_edge_chunk / EQUIGSPR_EDGE_CHUNK override it.
static std::atomic<int> g_edge_chunk{-1};
int edge_chunk_for(int n_edges, int waves_resident) {
    int forced = g_edge_chunk.load(std::memory_order_relaxed);
    if (forced < 0) {
        const char* v = getenv("EQUIGSPR_EDGE_CHUNK");
        forced = v ? atoi(v) : 0;
        g_edge_chunk.store(forced, std::memory_order_relaxed);
    }
    if (forced == 16 || forced == 32 || forced == 48 || forced == 64) return forced;
    if ((long long)n_edges >= 64ll * 2 * waves_resident) return 64;
    if ((long long)n_edges >= 32ll * 2 * waves_resident) return 32;
    return 16;
}
// Arithmetic of the edge kernels' 32 x 32 products (egnn_mfma16.h): 0 = split-bf16 (default), 1 = fp32.  -1 = not yet read from the environment.
static std::atomic<int> g_edge_fp32{-1};
bool edge_products_fp32() {
    int v = g_edge_fp32.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("EQUIGSPR_EDGE_PRODUCTS");
        v = (e && (!strcmp(e, "fp32") || !strcmp(e, "1"))) ? 1 : 0;
        g_edge_fp32.store(v, std::memory_order_relaxed);
    }
    return v == 1;
}
// Upper bound on the workgroups of the 16-edge kernels (developer / test knob; 0 = the kernel's default)
static std::atomic<int> g_edge_cap{-1};
int edge_grid_cap(int dflt) {
    int v = g_edge_cap.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("EQUIGSPR_EDGE_MAXBLOCKS");
        v = e ? atoi(e) : 0;
        if (v < 0) v = 0;
        g_edge_cap.store(v, std::memory_order_relaxed);
    }
    return (v > 0 && v < dflt) ? v : dflt;
}
}  // namespace eq

extern "C" int equigspr_set_edge_products(int32_t fp32) {
    EQ_CHECK_ARG(fp32 == 0 || fp32 == 1, EQUIGSPR_EINVAL, "set_edge_products: 0 (split-bf16) or 1 (fp32)");
    eq::g_edge_fp32.store(fp32, std::memory_order_relaxed);
    return 0;
}
extern "C" int equigspr_set_edge_grid_cap(int32_t max_blocks) {
    EQ_CHECK_ARG(max_blocks >= 0, EQUIGSPR_EINVAL, "set_edge_grid_cap: >= 0 (0 = default)");
    eq::g_edge_cap.store(max_blocks, std::memory_order_relaxed);
    return 0;
}

extern "C" int equigspr_set_edge_chunk(int32_t chunk) {
    EQ_CHECK_ARG(chunk == 0 || chunk == 16 || chunk == 32 || chunk == 48 || chunk == 64, EQUIGSPR_EINVAL,
                 "set_edge_chunk: 0 (automatic), 16, 32, 48 or 64");
    eq::g_edge_chunk.store(chunk, std::memory_order_relaxed);
    return 0;
}

extern "C" int equigspr_abi_version(void) { return EQUIGSPR_ABI_VERSION; }
extern "C" const char* equigspr_last_error(void) { return eq::g_err; }
// The HIP runtime keeps the last error of the calling thread until someone reads it, and every launch wrapper here reads it
// (EQ_CHECK_LAUNCH): after an ABORTED hipGraph capture (a collective that cannot be captured, say) the next, perfectly good launch
// would report that stale error.  A caller that has handled such a failure clears it with this.  Returns the code it cleared.
extern "C" int equigspr_clear_device_error(void) {
    const hipError_t e = hipGetLastError();
    (void)hipGetLastError();
    return (int)e;
}

extern "C" int64_t equigspr_egnn_param_count(const equigspr_egnn_config* c) {
    if (!eq::config_supported(c)) return -1;
    return eq::make_model_layout(c->in_nf, c->hidden_nf, c->out_nf, c->in_edge_nf, c->n_layers, c->num_heads).total;
}

extern "C" int64_t equigspr_egnn_layer_offset(const equigspr_egnn_config* c, int32_t layer) {
    if (!eq::config_supported(c) || layer < 0 || layer > c->n_layers) return -1;
    const eq::ModelLayout M = eq::make_model_layout(c->in_nf, c->hidden_nf, c->out_nf, c->in_edge_nf, c->n_layers, c->num_heads);
    return (int64_t)M.layer0 + (int64_t)layer * M.layer_size;
}
