// Shared host-side plumbing for the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/equigspr.h"
#include "egnn_math.h"

// Developer trace build (make trace): lane 0 of every wave drains its counters and stores s_memtime into slot k.
#ifdef EQ_TRACE
#define EQ_TRACE_SLOTS 16
static __device__ unsigned long long* eq_trace_ptr = nullptr;      // one copy per translation unit
#define EQ_TRACE_WAVES 16384                                        // waves per kernel id
#define EQ_STAMPK(kid, k)                                                                                   \
    do {                                                                                                    \
        __builtin_amdgcn_s_waitcnt(0);                                                                      \
        if ((threadIdx.x & 63) == 0 && ::eq_trace_ptr)                                                      \
            ::eq_trace_ptr[((size_t)(kid) * EQ_TRACE_WAVES + (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6))) * EQ_TRACE_SLOTS + (k)] = \
                __builtin_amdgcn_s_memtime();                                                               \
    } while (0)
#define EQ_STAMP(k) EQ_STAMPK(0, k)
#else
#define EQ_STAMP(k)
#define EQ_STAMPK(kid, k)
#endif

namespace eq {

void set_error(const char* fmt, ...);   // api.hip
int edge_chunk_for(int n_edges, int waves_resident);   // api.hip: chunk size of the 16-edge kernels
bool edge_products_fp32();                              // api.hip: equigspr_set_edge_products / EQUIGSPR_EDGE_PRODUCTS
int edge_grid_cap(int dflt);                            // api.hip: equigspr_set_edge_grid_cap / EQUIGSPR_EDGE_MAXBLOCKS

#define EQ_CHECK_ARG(cond, code, ...)            \
    do {                                         \
        if (!(cond)) {                           \
            ::eq::set_error(__VA_ARGS__);        \
            return (code);                       \
        }                                        \
    } while (0)

#define EQ_CHECK_LAUNCH(what)                                                     \
    do {                                                                          \
        hipError_t _e = hipGetLastError();                                        \
        if (_e != hipSuccess) {                                                   \
            ::eq::set_error("%s: %s", (what), hipGetErrorString(_e));            \
            return (int)_e;                                                       \
        }                                                                         \
    } while (0)

constexpr int kWave = 64;
static inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

// padded feature stride used by every internal [rows x F] fp32 buffer (16-byte rows)
static inline int fpad(int F) { return round_up(F, 4); }

inline bool config_supported(const equigspr_egnn_config* c) {
    if (!c) return false;
    const int hd = c->num_heads;
    const bool h_ok = hd == 1 || hd == 2 || hd == 4 || hd == 8;
    const bool fp_width = c->hidden_nf == 16 || c->hidden_nf == 48 || c->hidden_nf == 64;        // egnn_edge_fp.h
    const bool f_ok = (c->hidden_nf == 32 && h_ok) || (c->hidden_nf == 33 && hd == 1) || (fp_width && h_ok);
    return f_ok && c->in_nf >= 1 && c->in_nf <= 64 && c->out_nf >= 1 && c->out_nf <= 64 &&
           (c->in_edge_nf == 0 || c->in_edge_nf == 1) && c->n_layers >= 1 && c->n_layers <= 16 &&
           (c->edge_basis == EQUIGSPR_BASIS_FRAMES || c->edge_basis == EQUIGSPR_BASIS_IRREPS);
}

}  // namespace eq
