// What exactly can overwrite a source register of v_mfma_f32_16x16x32_bf16 before the instruction has read it?
// The split-bf16 products of the edge kernels (csrc/egnn_mfma16.h::mv16_bf) were corrupted in round 5 by a timing-dependent
// overwrite; the fix fences the instruction groups and tools/isa_lint.py checks the fence in the shipped objects.  This probe
// measures the hazard itself, in hand-placed assembly the compiler cannot reorder:
//
//     [ chain of PRE products into the same accumulator (dependent), or into different ones (independent) ]
//     v_mfma_f32_16x16x32_bf16 D, A, B, C          <- the probed instruction
//     s_nop (DIST - 1)                             <- DIST idle issue slots
//     overwrite of A, B or C:  v_mov_b32 | ds_read_b128 | global_load_dwordx4
//     wait for everything, read D
//
// with 0 or 1 further wave on the same SIMD keeping the matrix pipe busy.  A = B = 1 (bf16), C = 1: every element of D must be
// 32 * (PRE + 1) + 1 in the dependent form, 33 in the independent one.  The overwrite writes 1e30-class junk.
// Output: one line per (form, target, writer, DIST, PRE, hog): corrupted results per million probes.
//     hipcc -O2 --offload-arch=gfx950 tools/dev/mfma_war_probe.hip -o /tmp/war && /tmp/war
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// fixed registers of the hand-written block: A v[20:23], B v[24:27], C v[32:35], D v[36:39], independent-form outputs v[40:63]
#define REG_A "v[20:23]"
#define REG_B "v[24:27]"
#define REG_C "v[32:35]"
#define CLOBBERS "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", \
                 "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", \
                 "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "memory"

#define SETUP \
    "v_mov_b32 v20, %4\n v_mov_b32 v21, %4\n v_mov_b32 v22, %4\n v_mov_b32 v23, %4\n" \
    "v_mov_b32 v24, %4\n v_mov_b32 v25, %4\n v_mov_b32 v26, %4\n v_mov_b32 v27, %4\n" \
    "v_mov_b32 v32, 1.0\n v_mov_b32 v33, 1.0\n v_mov_b32 v34, 1.0\n v_mov_b32 v35, 1.0\n" \
    "s_nop 7\n s_nop 7\n"
#define MF(d, c) "v_mfma_f32_16x16x32_bf16 " d ", " REG_A ", " REG_B ", " c "\n"
// dependent chains: the first product takes C, the rest accumulate into D
#define DEP0  MF("v[36:39]", REG_C)
#define DEP2  MF("v[36:39]", REG_C) MF("v[36:39]", "v[36:39]") MF("v[36:39]", "v[36:39]")
#define DEP5  DEP2 MF("v[36:39]", "v[36:39]") MF("v[36:39]", "v[36:39]") MF("v[36:39]", "v[36:39]")
#define DEP11 DEP5 MF("v[36:39]", "v[36:39]") MF("v[36:39]", "v[36:39]") MF("v[36:39]", "v[36:39]") MF("v[36:39]", "v[36:39]") MF("v[36:39]", "v[36:39]") MF("v[36:39]", "v[36:39]")
// independent: PRE products into other registers, all reading the same A, B, C; the probed one last
#define IND5  MF("v[40:43]", REG_C) MF("v[44:47]", REG_C) MF("v[48:51]", REG_C) MF("v[52:55]", REG_C) MF("v[56:59]", REG_C) MF("v[36:39]", REG_C)
#define FINISH \
    "s_waitcnt vmcnt(0) lgkmcnt(0)\n s_nop 7\n s_nop 7\n s_nop 7\n" \
    "v_mov_b32 %0, v36\n v_mov_b32 %1, v37\n v_mov_b32 %2, v38\n v_mov_b32 %3, v39\n"

#define W_NONE(t)   ""
#define W_VMOV(t)   "v_mov_b32 " t ", %5\n"
#define W_LDS(t)    "ds_read_b128 " t ", %6\n"
#define W_GLOBAL(t) "global_load_dwordx4 " t ", %7, off\n"
// v_mov writes one register: name the first of the four
#define FIRST_A "v20"
#define FIRST_B "v24"
#define FIRST_C "v32"

#define BLOCK(chain, nop, writer) \
    asm volatile(SETUP chain nop writer FINISH \
                 : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) \
                 : "v"(ones), "v"(junk1), "v"(lds_addr), "v"(gptr) \
                 : CLOBBERS)

enum { T_A = 0, T_B = 1, T_C = 2 };
enum { WR_NONE = 0, WR_VMOV = 1, WR_LDS = 2, WR_GLOBAL = 3 };
enum { CH_DEP0 = 0, CH_DEP2 = 1, CH_DEP5 = 2, CH_DEP11 = 3, CH_IND5 = 4 };

template <int CH, int TGT, int WR, int NOP>   // NOP: -1 none, else the s_nop operand (NOP + 1 idle slots)
__device__ __forceinline__ void one_probe(float& d0, float& d1, float& d2, float& d3, unsigned ones, float junk1, unsigned lds_addr, const float* gptr) {
#define NOPSTR(n) "s_nop " #n "\n"
#define WITH_WRITER(chain, nop) \
    if constexpr (WR == WR_NONE) BLOCK(chain, nop, ""); \
    else if constexpr (WR == WR_VMOV && TGT == T_A) BLOCK(chain, nop, W_VMOV(FIRST_A)); \
    else if constexpr (WR == WR_VMOV && TGT == T_B) BLOCK(chain, nop, W_VMOV(FIRST_B)); \
    else if constexpr (WR == WR_VMOV && TGT == T_C) BLOCK(chain, nop, W_VMOV(FIRST_C)); \
    else if constexpr (WR == WR_LDS && TGT == T_A) BLOCK(chain, nop, W_LDS(REG_A)); \
    else if constexpr (WR == WR_LDS && TGT == T_B) BLOCK(chain, nop, W_LDS(REG_B)); \
    else if constexpr (WR == WR_LDS && TGT == T_C) BLOCK(chain, nop, W_LDS(REG_C)); \
    else if constexpr (WR == WR_GLOBAL && TGT == T_A) BLOCK(chain, nop, W_GLOBAL(REG_A)); \
    else if constexpr (WR == WR_GLOBAL && TGT == T_B) BLOCK(chain, nop, W_GLOBAL(REG_B)); \
    else BLOCK(chain, nop, W_GLOBAL(REG_C));
#define WITH_NOP(chain) \
    if constexpr (NOP < 0) { WITH_WRITER(chain, "") } \
    else if constexpr (NOP == 0) { WITH_WRITER(chain, NOPSTR(0)) } \
    else if constexpr (NOP == 1) { WITH_WRITER(chain, NOPSTR(1)) } \
    else if constexpr (NOP == 3) { WITH_WRITER(chain, NOPSTR(3)) } \
    else if constexpr (NOP == 7) { WITH_WRITER(chain, NOPSTR(7)) } \
    else { WITH_WRITER(chain, NOPSTR(15)) }
    if constexpr (CH == CH_DEP0) { WITH_NOP(DEP0) }
    else if constexpr (CH == CH_DEP2) { WITH_NOP(DEP2) }
    else if constexpr (CH == CH_DEP5) { WITH_NOP(DEP5) }
    else if constexpr (CH == CH_DEP11) { WITH_NOP(DEP11) }
    else { WITH_NOP(IND5) }
}

// Block of 512 threads: waves 0..3 probe (one per SIMD), waves 4..7 share their SIMDs and, if `hog`, run products back to back.
template <int CH, int TGT, int WR, int NOP>
__global__ __launch_bounds__(512) void probe_kernel(const float* __restrict__ gjunk, unsigned long long* __restrict__ bad, int iters, int hog, float expect) {
    __shared__ float lds_junk[64 * 4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x < 256) lds_junk[threadIdx.x] = 3.0e30f;
    __syncthreads();
    if (wave >= 4) {
        if (!hog) return;
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * lane); b[j] = (__bf16)1.0f; }
        f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
        for (int it = 0; it < iters * 6; ++it) {       // two independent chains: the pipe never idles while the probes run
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc1, 0, 0, 0);
        }
        if (acc0[0] + acc1[0] == 12345.678f) bad[1] = 1;   // keep the chain
        return;
    }
    const unsigned ones = 0x3f803f80u;                     // two bf16 1.0
    const unsigned lds_addr = (unsigned)(size_t)(&lds_junk[0]) + lane * 16;   // LDS offsets are the low 32 bits of the generic address on gfx9
    const float* gptr = gjunk + lane * 4;
    unsigned long long wrong = 0;
    for (int it = 0; it < iters; ++it) {
        float d0, d1, d2, d3;
        one_probe<CH, TGT, WR, NOP>(d0, d1, d2, d3, ones, 3.0e30f, lds_addr, gptr);
        wrong += (d0 != expect) || (d1 != expect) || (d2 != expect) || (d3 != expect);
    }
    // a probe is wrong if any lane saw a wrong element
    const unsigned long long any = __ballot(wrong != 0) ? 1 : 0;
    unsigned long long tot = wrong;
    for (int o = 32; o; o >>= 1) tot += __shfl_xor(tot, o);
    if (lane == 0) { atomicAdd(bad, tot); (void)any; }
}

static float* g_junk;
static unsigned long long* g_bad;

template <int CH, int TGT, int WR, int NOP>
static void run(const char* chain, const char* tgt, const char* wr, int dist, float expect) {
    const int iters = 20000, blocks = 256;
    for (int hog = 0; hog < 2; ++hog) {
        CHECK(hipMemset(g_bad, 0, 16));
        hipLaunchKernelGGL((probe_kernel<CH, TGT, WR, NOP>), dim3(blocks), dim3(512), 0, 0, g_junk, g_bad, iters, hog, expect);
        CHECK(hipDeviceSynchronize());
        unsigned long long bad = 0;
        CHECK(hipMemcpy(&bad, g_bad, 8, hipMemcpyDeviceToHost));
        const double lanes = (double)iters * blocks * 4 * 64;
        printf("%-6s %-2s %-7s dist %2d  hog %d   wrong lane-results per million: %12.3f\n", chain, tgt, wr, dist, hog, 1e6 * (double)bad / lanes);
    }
}

template <int CH, int TGT, int WR>
static void sweep(const char* chain, const char* tgt, const char* wr, float expect) {
    run<CH, TGT, WR, -1>(chain, tgt, wr, 0, expect);
    run<CH, TGT, WR, 0>(chain, tgt, wr, 1, expect);
    run<CH, TGT, WR, 1>(chain, tgt, wr, 2, expect);
    run<CH, TGT, WR, 3>(chain, tgt, wr, 4, expect);
    run<CH, TGT, WR, 7>(chain, tgt, wr, 8, expect);
    run<CH, TGT, WR, 15>(chain, tgt, wr, 16, expect);
}

template <int CH>
static void chain_sweep(const char* chain, float expect) {
    run<CH, T_C, WR_NONE, -1>(chain, "-", "none", 0, expect);
    sweep<CH, T_C, WR_VMOV>(chain, "C", "v_mov", expect);
    sweep<CH, T_A, WR_VMOV>(chain, "A", "v_mov", expect);
    sweep<CH, T_C, WR_LDS>(chain, "C", "ds_read", expect);
    sweep<CH, T_A, WR_LDS>(chain, "A", "ds_read", expect);
    sweep<CH, T_B, WR_LDS>(chain, "B", "ds_read", expect);
    sweep<CH, T_C, WR_GLOBAL>(chain, "C", "global", expect);
    sweep<CH, T_A, WR_GLOBAL>(chain, "A", "global", expect);
}


// ---- control: the overwrite BEFORE the chain must be seen by every probe (the checker works) ----
__global__ __launch_bounds__(256) void control_kernel(unsigned long long* __restrict__ bad, int iters, float expect) {
    __shared__ float lds_junk[64 * 4];
    const int lane = threadIdx.x & 63;
    lds_junk[threadIdx.x] = 3.0e30f;
    __syncthreads();
    const unsigned ones = 0x3f803f80u;
    const unsigned lds_addr = (unsigned)(size_t)(&lds_junk[0]) + lane * 16;
    unsigned long long wrong = 0;
    for (int it = 0; it < iters; ++it) {
        float d0, d1, d2, d3;
        asm volatile(SETUP "ds_read_b128 " REG_A ", %5\n s_waitcnt lgkmcnt(0)\n" DEP2 FINISH
                     : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(ones), "v"(lds_addr) : CLOBBERS);
        wrong += (d0 != expect) || (d1 != expect) || (d2 != expect) || (d3 != expect);
    }
    for (int o = 32; o; o >>= 1) wrong += __shfl_xor(wrong, o);
    if (lane == 0) atomicAdd(bad, wrong);
}

// ---- the instruction order of the failing build (egnn_bwd without the closed group, 0x725E8 .. 0x726B4 of its listing) ----
// fragments come from LDS, the next fragment is read into the registers of the one just consumed, in the middle of the chain:
//     A1, A2 <- old (1.0);  acc += A1 B1;  acc += A1 B2;  A1 <- new (2.0);  acc += A2 B1;  acc += A2 B2;  A2 <- new (2.0);
//     wait(1);  acc2 += A1' B2;  wait(0);  acc2 += A2' B1
// acc must be 32 * (PRE + 4) (an early overwrite makes it larger), acc2 must be 128 (a late one makes it smaller).
// All eight waves of the workgroup run it (two per SIMD, as in the product); `jitter` shifts their phases against each other.
#define RMF(d, a, b) "v_mfma_f32_16x16x32_bf16 " d ", " a ", " b ", " d "\n"
#define R_A1 "v[20:23]"
#define R_A2 "v[44:47]"
#define R_B1 "v[24:27]"
#define R_B2 "v[28:31]"
#define R_ACC "v[36:39]"
#define R_ACC2 "v[40:43]"
#define REAL_SETUP \
    "v_mov_b32 v24, %8\n v_mov_b32 v25, %8\n v_mov_b32 v26, %8\n v_mov_b32 v27, %8\n" \
    "v_mov_b32 v28, %8\n v_mov_b32 v29, %8\n v_mov_b32 v30, %8\n v_mov_b32 v31, %8\n" \
    "v_mov_b32 v36, 0\n v_mov_b32 v37, 0\n v_mov_b32 v38, 0\n v_mov_b32 v39, 0\n" \
    "v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n" \
    "ds_read_b128 " R_A1 ", %9\n ds_read_b128 " R_A2 ", %9\n s_waitcnt lgkmcnt(0)\n s_nop 7\n"
#define REAL_BODY \
    RMF(R_ACC, R_A1, R_B1) RMF(R_ACC, R_A1, R_B2) "ds_read_b128 " R_A1 ", %9 offset:4096\n" \
    RMF(R_ACC, R_A2, R_B1) RMF(R_ACC, R_A2, R_B2) "ds_read_b128 " R_A2 ", %9 offset:4096\n" \
    "s_waitcnt lgkmcnt(1)\n" RMF(R_ACC2, R_A1, R_B2) "s_waitcnt lgkmcnt(0)\n" RMF(R_ACC2, R_A2, R_B1)
#define REAL_FINISH \
    "s_nop 7\n s_nop 7\n s_nop 7\n" \
    "v_mov_b32 %0, v36\n v_mov_b32 %1, v37\n v_mov_b32 %2, v38\n v_mov_b32 %3, v39\n" \
    "v_mov_b32 %4, v40\n v_mov_b32 %5, v41\n v_mov_b32 %6, v42\n v_mov_b32 %7, v43\n"
#define REAL_PRE4 RMF(R_ACC, R_A2, R_B1) RMF(R_ACC, R_A2, R_B1) RMF(R_ACC, R_A2, R_B1) RMF(R_ACC, R_A2, R_B1)

template <int PRE>
__global__ __launch_bounds__(512) void real_kernel(unsigned long long* __restrict__ bad, int iters, int jitter) {
    __shared__ float frag[2 * 1024];           // [old | new], 64 lanes x 16 bytes each
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = threadIdx.x; t < 2048; t += 512) frag[t] = __uint_as_float(t < 1024 ? 0x3f803f80u : 0x40004000u);
    __syncthreads();
    const unsigned ones = 0x3f803f80u;
    const unsigned addr = (unsigned)(size_t)(&frag[0]) + lane * 16;
    const float e1 = 32.0f * (PRE + 4), e2 = 128.0f;
    unsigned long long early = 0, late = 0;
    for (int it = 0; it < iters; ++it) {
        if (jitter) { const int n = (it * (wave + 3)) & 15; for (int j = 0; j < n; ++j) __builtin_amdgcn_s_sleep(1); }
        float a0, a1, a2, a3, b0, b1, b2, b3;
        if constexpr (PRE == 0)
            asm volatile(REAL_SETUP REAL_BODY REAL_FINISH : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(b0), "=v"(b1), "=v"(b2), "=v"(b3) : "v"(ones), "v"(addr) : CLOBBERS);
        else
            asm volatile(REAL_SETUP REAL_PRE4 REAL_BODY REAL_FINISH : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(b0), "=v"(b1), "=v"(b2), "=v"(b3) : "v"(ones), "v"(addr) : CLOBBERS);
        early += (a0 != e1) || (a1 != e1) || (a2 != e1) || (a3 != e1);
        late += (b0 != e2) || (b1 != e2) || (b2 != e2) || (b3 != e2);
    }
    for (int o = 32; o; o >>= 1) { early += __shfl_xor(early, o); late += __shfl_xor(late, o); }
    if (lane == 0) { atomicAdd(bad, early); atomicAdd(bad + 1, late); }
}

template <int PRE>
static void run_real(int blocks) {
    const int iters = 20000;
    for (int jitter = 0; jitter < 2; ++jitter) {
        CHECK(hipMemset(g_bad, 0, 16));
        hipLaunchKernelGGL((real_kernel<PRE>), dim3(blocks), dim3(512), 0, 0, g_bad, iters, jitter);
        CHECK(hipDeviceSynchronize());
        unsigned long long bad[2] = {0, 0};
        CHECK(hipMemcpy(bad, g_bad, 16, hipMemcpyDeviceToHost));
        const double lanes = (double)iters * blocks * 8 * 64;
        printf("real   pre %d  blocks %4d  jitter %d   per million lane-results:  consumed the NEXT fragment %12.3f   consumed the OLD fragment %12.3f\n",
               PRE, blocks, jitter, 1e6 * bad[0] / lanes, 1e6 * bad[1] / lanes);
    }
}

// ---- read-after-write: a source written by a VALU instruction (or an LDS read + s_waitcnt) DIST slots BEFORE the matrix instruction.
// A = B = 1 except the freshly written register (2.0 pairs), C = 1 except a fresh 3.0: a stale read gives the old sum.
//   fresh A/B register (k = 0, 1 of each lane group):  D = 4 * (2 * 2 + 6) + 1 = 41 (stale: 33);   fresh C[0]: D0 = 32 + 3 = 35 (stale: 33)
#define RAW_BLOCK(writer, nop) \
    asm volatile(SETUP writer nop DEP0 FINISH : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) \
                 : "v"(ones), "v"(twos), "v"(lds_addr), "v"(sel) : CLOBBERS)
template <int TGT, int WR, int NOP>    // WR 0: v_mov_b32, 1: v_perm_b32, 2: ds_read_b32 + s_waitcnt
__global__ __launch_bounds__(512) void raw_kernel(unsigned long long* __restrict__ bad, int iters) {
    __shared__ float lds_two[64];
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 64) lds_two[threadIdx.x] = __uint_as_float(TGT == T_C ? 0x40400000u : 0x40004000u);
    __syncthreads();
    const unsigned ones = 0x3f803f80u, twos = TGT == T_C ? 0x40400000u : 0x40004000u, sel = TGT == T_C ? 0x07060504u : 0x07060302u;
    const unsigned lds_addr = (unsigned)(size_t)(&lds_two[0]) + lane * 4;
    const float expect0 = TGT == T_C ? 35.0f : 41.0f, expect = TGT == T_C ? 33.0f : 41.0f;
    unsigned long long wrong = 0;
    for (int it = 0; it < iters; ++it) {
        float d0, d1, d2, d3;
#define RAW_T(t) \
        if constexpr (WR == 0) { RAW_N("v_mov_b32 " t ", %5\n") } \
        else if constexpr (WR == 1) { RAW_N("v_perm_b32 " t ", %5, %5, %7\n") } \
        else { RAW_N("ds_read_b32 " t ", %6\n s_waitcnt lgkmcnt(0)\n") }
#define RAW_N(w) \
        if constexpr (NOP < 0) RAW_BLOCK(w, ""); \
        else if constexpr (NOP == 0) RAW_BLOCK(w, "s_nop 0\n"); \
        else if constexpr (NOP == 1) RAW_BLOCK(w, "s_nop 1\n"); \
        else RAW_BLOCK(w, "s_nop 3\n");
        if constexpr (TGT == T_A) { RAW_T("v20") } else if constexpr (TGT == T_B) { RAW_T("v24") } else { RAW_T("v32") }
        wrong += (d0 != expect0) || (d1 != expect) || (d2 != expect) || (d3 != expect);
    }
    for (int o = 32; o; o >>= 1) wrong += __shfl_xor(wrong, o);
    if (lane == 0) atomicAdd(bad, wrong);
}
template <int TGT, int WR, int NOP>
static void run_raw(const char* tgt, const char* wr, int dist) {
    const int iters = 20000, blocks = 256;
    CHECK(hipMemset(g_bad, 0, 16));
    hipLaunchKernelGGL((raw_kernel<TGT, WR, NOP>), dim3(blocks), dim3(512), 0, 0, g_bad, iters);
    CHECK(hipDeviceSynchronize());
    unsigned long long bad = 0;
    CHECK(hipMemcpy(&bad, g_bad, 8, hipMemcpyDeviceToHost));
    printf("raw    %-2s written by %-8s %2d idle slots before the instruction   stale lane-results per million: %12.3f\n", tgt, wr, dist,
           1e6 * (double)bad / ((double)iters * blocks * 512));
}
template <int TGT, int WR>
static void sweep_raw(const char* tgt, const char* wr) {
    run_raw<TGT, WR, -1>(tgt, wr, 0); run_raw<TGT, WR, 0>(tgt, wr, 1); run_raw<TGT, WR, 1>(tgt, wr, 2); run_raw<TGT, WR, 3>(tgt, wr, 4);
}

// ---- which instructions count as wait states between a VALU write of B and the matrix instruction that reads it?
// (the compiler's hazard recogniser counts every issued instruction as one; two are required, see the raw table)
#define FILL_BLOCK(fill) \
    asm volatile(SETUP "v_mov_b32 v24, %5\n" fill DEP0 FINISH : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) \
                 : "v"(ones), "v"(twos), "v"(lds_addr), "v"(gptr) : CLOBBERS, "s40", "s41")
template <int F>
__global__ __launch_bounds__(512) void fill_kernel(const float* __restrict__ gjunk, unsigned long long* __restrict__ bad, int iters) {
    __shared__ float lds_x[64];
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 64) lds_x[threadIdx.x] = 0.0f;
    __syncthreads();
    const unsigned ones = 0x3f803f80u, twos = 0x40004000u;
    const unsigned lds_addr = (unsigned)(size_t)(&lds_x[0]) + lane * 4;
    const float* gptr = gjunk + lane;
    unsigned long long wrong = 0;
    for (int it = 0; it < iters; ++it) {
        float d0, d1, d2, d3;
        if constexpr (F == 0) FILL_BLOCK("");
        else if constexpr (F == 1) FILL_BLOCK("s_nop 0\n s_nop 0\n");
        else if constexpr (F == 2) FILL_BLOCK("s_nop 1\n");
        else if constexpr (F == 3) FILL_BLOCK("s_waitcnt lgkmcnt(0)\n s_waitcnt lgkmcnt(0)\n");
        else if constexpr (F == 4) FILL_BLOCK("ds_read_b32 v50, %6\n s_waitcnt lgkmcnt(1)\n");
        else if constexpr (F == 5) FILL_BLOCK("ds_read_b32 v50, %6\n ds_read_b32 v51, %6\n");
        else if constexpr (F == 6) FILL_BLOCK("s_mov_b32 s40, 0\n s_mov_b32 s41, 0\n");
        else if constexpr (F == 7) FILL_BLOCK("v_mov_b32 v50, 0\n v_mov_b32 v51, 0\n");
        else if constexpr (F == 8) FILL_BLOCK("v_mov_b32 v50, 0\n s_waitcnt lgkmcnt(0)\n");
        else if constexpr (F == 9) FILL_BLOCK("global_load_dword v50, %7, off\n s_waitcnt vmcnt(1)\n");
        else if constexpr (F == 10) FILL_BLOCK("v_mfma_f32_16x16x32_bf16 v[44:47], v[52:55], v[56:59], v[44:47]\n");
        else if constexpr (F == 11) FILL_BLOCK("ds_read_b32 v50, %6\n s_waitcnt lgkmcnt(1)\n s_waitcnt lgkmcnt(1)\n");
        else if constexpr (F == 12) FILL_BLOCK("ds_read_b32 v50, %6\n ds_read_b32 v51, %6\n s_waitcnt lgkmcnt(2)\n");
        else if constexpr (F == 13) FILL_BLOCK("v_mov_b32 v50, 0\n");
        else if constexpr (F == 14) FILL_BLOCK("v_mov_b32 v50, 0\n ds_read_b32 v51, %6\n");
        else FILL_BLOCK("s_nop 0\n ds_read_b32 v51, %6\n");
        wrong += (d0 != 41.0f) || (d1 != 41.0f) || (d2 != 41.0f) || (d3 != 41.0f);
    }
    for (int o = 32; o; o >>= 1) wrong += __shfl_xor(wrong, o);
    if (lane == 0) atomicAdd(bad, wrong);
}
template <int F>
static void run_fill(const char* what) {
    const int iters = 20000, blocks = 256;
    CHECK(hipMemset(g_bad, 0, 16));
    hipLaunchKernelGGL((fill_kernel<F>), dim3(blocks), dim3(512), 0, 0, g_junk, g_bad, iters);
    CHECK(hipDeviceSynchronize());
    unsigned long long bad = 0;
    CHECK(hipMemcpy(&bad, g_bad, 8, hipMemcpyDeviceToHost));
    printf("fill   v_mov B;  %-62s;  v_mfma    stale lane-results per million: %12.3f\n", what, 1e6 * (double)bad / ((double)iters * blocks * 512));
}

// ---- write-after-read by a VALU instruction, per register of the operand: the matrix instruction reads its sources over several
// cycles; which of the four registers of A / B / C can a v_mov_b32 right behind it still change?
#define WARV_BLOCK(chain, nop, reg) \
    asm volatile(SETUP chain nop "v_mov_b32 " reg ", %5\n" FINISH : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) \
                 : "v"(ones), "v"(junk1) : CLOBBERS)
template <int REG, int NOP, int PRE>     // REG: 20..23 A, 24..27 B, 32..35 C
__global__ __launch_bounds__(512) void warv_kernel(unsigned long long* __restrict__ bad, int iters) {
    const int lane = threadIdx.x & 63;
    const unsigned ones = 0x3f803f80u;
    const float junk1 = 3.0e30f, expect = PRE ? 97.0f : 33.0f;
    unsigned long long wrong = 0;
    for (int it = 0; it < iters; ++it) {
        float d0, d1, d2, d3;
#define WARV_R(chain, nop) \
        if constexpr (REG == 20) WARV_BLOCK(chain, nop, "v20"); else if constexpr (REG == 21) WARV_BLOCK(chain, nop, "v21"); \
        else if constexpr (REG == 22) WARV_BLOCK(chain, nop, "v22"); else if constexpr (REG == 23) WARV_BLOCK(chain, nop, "v23"); \
        else if constexpr (REG == 24) WARV_BLOCK(chain, nop, "v24"); else if constexpr (REG == 25) WARV_BLOCK(chain, nop, "v25"); \
        else if constexpr (REG == 26) WARV_BLOCK(chain, nop, "v26"); else if constexpr (REG == 27) WARV_BLOCK(chain, nop, "v27"); \
        else if constexpr (REG == 32) WARV_BLOCK(chain, nop, "v32"); else if constexpr (REG == 33) WARV_BLOCK(chain, nop, "v33"); \
        else if constexpr (REG == 34) WARV_BLOCK(chain, nop, "v34"); else WARV_BLOCK(chain, nop, "v35");
#define WARV_N(chain) \
        if constexpr (NOP < 0) { WARV_R(chain, "") } else if constexpr (NOP == 0) { WARV_R(chain, "s_nop 0\n") } \
        else if constexpr (NOP == 1) { WARV_R(chain, "s_nop 1\n") } else if constexpr (NOP == 2) { WARV_R(chain, "s_nop 2\n") } \
        else if constexpr (NOP == 3) { WARV_R(chain, "s_nop 3\n") } else { WARV_R(chain, "s_nop 7\n") }
        if constexpr (PRE) { WARV_N(DEP2) } else { WARV_N(DEP0) }
        wrong += (d0 != expect) || (d1 != expect) || (d2 != expect) || (d3 != expect);
    }
    for (int o = 32; o; o >>= 1) wrong += __shfl_xor(wrong, o);
    if (lane == 0) atomicAdd(bad, wrong);
}
template <int REG, int NOP, int PRE>
static double run_warv() {
    const int iters = 5000, blocks = 256;
    CHECK(hipMemset(g_bad, 0, 16));
    hipLaunchKernelGGL((warv_kernel<REG, NOP, PRE>), dim3(blocks), dim3(512), 0, 0, g_bad, iters);
    CHECK(hipDeviceSynchronize());
    unsigned long long bad = 0;
    CHECK(hipMemcpy(&bad, g_bad, 8, hipMemcpyDeviceToHost));
    return 1e6 * (double)bad / ((double)iters * blocks * 512);
}
template <int REG, int PRE>
static void sweep_warv(const char* name) {
    printf("warv   %s overwritten by v_mov_b32 behind %s; idle slots 0 1 2 3 4 8:  %10.1f %10.1f %10.1f %10.1f %10.1f %10.1f   (wrong lane-results per million)\n", name,
           PRE ? "3 dependent instr." : "1 instruction     ", run_warv<REG, -1, PRE>(), run_warv<REG, 0, PRE>(), run_warv<REG, 1, PRE>(), run_warv<REG, 2, PRE>(), run_warv<REG, 3, PRE>(), run_warv<REG, 7, PRE>());
}

// ---- result read: how many wait states behind the matrix instruction may a VALU instruction read D, and which instructions count?
// (software-managed: the compiler's hazard recogniser places s_nop / counts the instructions in between, one wait state each)
//   D starts as 0.5 everywhere (stale value), the product writes 33.
#define RD_BLOCK(fill) \
    asm volatile(SETUP "v_mov_b32 v36, 0.5\n v_mov_b32 v37, 0.5\n v_mov_b32 v38, 0.5\n v_mov_b32 v39, 0.5\n s_nop 7\n" \
                 "v_mfma_f32_16x16x32_bf16 v[36:39], v[20:23], v[24:27], v[32:35]\n" fill \
                 "v_mov_b32 %0, v36\n v_mov_b32 %1, v37\n v_mov_b32 %2, v38\n v_mov_b32 %3, v39\n s_waitcnt vmcnt(0) lgkmcnt(0)\n s_nop 7\n s_nop 7\n" \
                 : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(ones), "v"(lds_addr) : CLOBBERS, "s40", "s41")
#define N1(x) x
#define N2(x) x x
#define N3(x) x x x
#define N4(x) x x x x
#define N5(x) N4(x) x
#define N6(x) N4(x) x x
#define N7(x) N4(x) x x x
#define N8(x) N4(x) N4(x)
#define N9(x) N8(x) x
#define N10(x) N8(x) x x
#define N12(x) N8(x) N4(x)
template <int KIND, int N>   // KIND 0: s_nop 0, 1: s_waitcnt (satisfied), 2: independent v_mov, 3: ds_read_b32 (unrelated), 4: s_mov_b32
__global__ __launch_bounds__(512) void rd_kernel(unsigned long long* __restrict__ bad, int iters) {
    __shared__ float lds_x[64];
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 64) lds_x[threadIdx.x] = 0.0f;
    __syncthreads();
    const unsigned ones = 0x3f803f80u;
    const unsigned lds_addr = (unsigned)(size_t)(&lds_x[0]) + lane * 4;
    unsigned long long wrong = 0;
    for (int it = 0; it < iters; ++it) {
        float d0, d1, d2, d3;
#define RD_KIND(NN) \
        if constexpr (KIND == 0) RD_BLOCK(NN("s_nop 0\n")); else if constexpr (KIND == 1) RD_BLOCK(NN("s_waitcnt lgkmcnt(0)\n")); \
        else if constexpr (KIND == 2) RD_BLOCK(NN("v_mov_b32 v50, 0\n")); else if constexpr (KIND == 3) RD_BLOCK(NN("ds_read_b32 v51, %5\n")); \
        else RD_BLOCK(NN("s_mov_b32 s40, 0\n"));
        if constexpr (N == 0) RD_BLOCK("");
        else if constexpr (N == 1) { RD_KIND(N1) } else if constexpr (N == 2) { RD_KIND(N2) } else if constexpr (N == 3) { RD_KIND(N3) }
        else if constexpr (N == 4) { RD_KIND(N4) } else if constexpr (N == 5) { RD_KIND(N5) } else if constexpr (N == 6) { RD_KIND(N6) }
        else if constexpr (N == 7) { RD_KIND(N7) } else if constexpr (N == 8) { RD_KIND(N8) } else if constexpr (N == 9) { RD_KIND(N9) }
        else if constexpr (N == 10) { RD_KIND(N10) } else { RD_KIND(N12) }
        wrong += (d0 != 33.0f) || (d1 != 33.0f) || (d2 != 33.0f) || (d3 != 33.0f);
    }
    for (int o = 32; o; o >>= 1) wrong += __shfl_xor(wrong, o);
    if (lane == 0) atomicAdd(bad, wrong);
}
template <int KIND, int N>
static double run_rd(int threads) {
    const int iters = 5000, blocks = 256;
    CHECK(hipMemset(g_bad, 0, 16));
    hipLaunchKernelGGL((rd_kernel<KIND, N>), dim3(blocks), dim3(threads), 0, 0, g_bad, iters);
    CHECK(hipDeviceSynchronize());
    unsigned long long bad = 0;
    CHECK(hipMemcpy(&bad, g_bad, 8, hipMemcpyDeviceToHost));
    return 1e6 * (double)bad / ((double)iters * blocks * threads);
}
template <int KIND>
static void sweep_rd(const char* what) {
    for (int threads = 256; threads <= 512; threads += 256) {
        printf("read   D read by a VALU instruction behind N x %-28s (%d wave/SIMD)  N = 0 1 2 3 4 5 6 7 8 9 10 12:", what, threads / 256);
        printf(" %8.0f %8.0f %8.0f %8.0f %8.0f %8.0f %8.0f %8.0f %8.0f %8.0f %8.0f %8.0f   stale per million\n",
               run_rd<KIND, 0>(threads), run_rd<KIND, 1>(threads), run_rd<KIND, 2>(threads), run_rd<KIND, 3>(threads), run_rd<KIND, 4>(threads), run_rd<KIND, 5>(threads),
               run_rd<KIND, 6>(threads), run_rd<KIND, 7>(threads), run_rd<KIND, 8>(threads), run_rd<KIND, 9>(threads), run_rd<KIND, 10>(threads), run_rd<KIND, 12>(threads));
    }
}

// ---- interleaved accumulation chains: acc1 and acc2 alternate (each instruction depends on the one TWO back), optionally with the
// filler instructions the compiler puts between them, optionally against an fp32 16x16x4 chain.  Every bf16 instruction adds 32, every fp32 one 4.
#define IB1 "v_mfma_f32_16x16x32_bf16 v[36:39], v[20:23], v[24:27], v[36:39]\n"
#define IB2 "v_mfma_f32_16x16x32_bf16 v[40:43], v[20:23], v[24:27], v[40:43]\n"
#define IF3 "v_mfma_f32_16x16x4_f32 v[44:47], v60, v61, v[44:47]\n"
#define SETUP6 \
    "v_mov_b32 v20, %6\n v_mov_b32 v21, %6\n v_mov_b32 v22, %6\n v_mov_b32 v23, %6\n" \
    "v_mov_b32 v24, %6\n v_mov_b32 v25, %6\n v_mov_b32 v26, %6\n v_mov_b32 v27, %6\n s_nop 7\n"
#define IL_BLOCK(body) \
    asm volatile(SETUP6 "v_mov_b32 v36, 0\n v_mov_b32 v37, 0\n v_mov_b32 v38, 0\n v_mov_b32 v39, 0\n v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n" \
                 "v_mov_b32 v44, 0\n v_mov_b32 v45, 0\n v_mov_b32 v46, 0\n v_mov_b32 v47, 0\n v_mov_b32 v60, 1.0\n v_mov_b32 v61, 1.0\n s_nop 7\n" body \
                 "s_waitcnt vmcnt(0) lgkmcnt(0)\n s_nop 7\n s_nop 7\n s_nop 7\n" \
                 "v_mov_b32 %0, v36\n v_mov_b32 %1, v39\n v_mov_b32 %2, v40\n v_mov_b32 %3, v43\n v_mov_b32 %4, v44\n v_mov_b32 %5, v47\n" \
                 : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5) : "v"(ones), "v"(lds_addr) : CLOBBERS, "s40")
#define FV "v_mov_b32 v50, 0\n"
#define FW "s_waitcnt lgkmcnt(0)\n"
#define FD "ds_read_b32 v51, %7\n"
template <int V>
__global__ __launch_bounds__(512) void il_kernel(unsigned long long* __restrict__ bad, int iters, int jitter) {
    __shared__ float lds_x[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 64) lds_x[threadIdx.x] = 0.0f;
    __syncthreads();
    const unsigned ones = 0x3f803f80u;
    const unsigned lds_addr = (unsigned)(size_t)(&lds_x[0]) + lane * 4;
    float e1, e2, e3;
    unsigned long long wrong = 0;
    for (int it = 0; it < iters; ++it) {
        if (jitter) { const int n = (it * (wave + 3)) & 15; for (int j = 0; j < n; ++j) __builtin_amdgcn_s_sleep(1); }
        float d0, d1, d2, d3, d4, d5;
        if constexpr (V == 0) { IL_BLOCK(IB1 IB2 IB1 IB2 IB1 IB2 IB1 IB2 IB1 IB2 IB1 IB2); e1 = 192; e2 = 192; e3 = 0; }
        else if constexpr (V == 1) { IL_BLOCK(IB1 FV FW IB2 FD IB1 FW IB2 FV IB1 FD FW IB2 IB1 FV FV IB2 FW IB1 IB2 FD IB1 IB2); e1 = 192; e2 = 192; e3 = 0; }
        else if constexpr (V == 2) { IL_BLOCK(IB1 IF3 IB1 IF3 IB1 IF3 IB1 IF3 IB1 IF3 IB1 IF3); e1 = 192; e2 = 0; e3 = 24; }
        else if constexpr (V == 3) { IL_BLOCK(IF3 IF3 IB1 IF3 IF3 IB1 IB2 IF3 IB1 IB2 IF3 IF3 IB1 IB2 IF3 IB2); e1 = 128; e2 = 128; e3 = 32; }
        else if constexpr (V == 4) { IL_BLOCK(IB1 FV IF3 FW IB1 FD IF3 IB2 FV IB1 IF3 FW IB2 IB1 FD IF3 FV IB2 IF3 IB1 FW IF3); e1 = 160; e2 = 96; e3 = 28; }
        else { IL_BLOCK(IB1 IB1 IB1 IB1 IB1 IB1 IB2 IB2 IB2 IB2 IB2 IB2); e1 = 192; e2 = 192; e3 = 0; }
        wrong += (d0 != e1) || (d1 != e1) || (d2 != e2) || (d3 != e2) || (d4 != e3) || (d5 != e3);
    }
    for (int o = 32; o; o >>= 1) wrong += __shfl_xor(wrong, o);
    if (lane == 0) atomicAdd(bad, wrong);
}
template <int V>
static void run_il(const char* what) {
    const int iters = 10000;
    for (int blocks = 1; blocks <= 1024; blocks *= 32)
        for (int jitter = 0; jitter < 2; ++jitter) {
            CHECK(hipMemset(g_bad, 0, 16));
            hipLaunchKernelGGL((il_kernel<V>), dim3(blocks), dim3(512), 0, 0, g_bad, iters, jitter);
            CHECK(hipDeviceSynchronize());
            unsigned long long bad = 0;
            CHECK(hipMemcpy(&bad, g_bad, 8, hipMemcpyDeviceToHost));
            printf("chains %-58s blocks %4d jitter %d   wrong lane-results per million: %12.3f\n", what, blocks, jitter, 1e6 * (double)bad / ((double)iters * blocks * 512));
        }
}

// ---- the same result read with a second matrix instruction among the wait states (as the compiler schedules it when matrix
// instructions may cross the product's fences): N x s_nop 0, then one independent fp32 16x16x4 or bf16 instruction, then M x s_nop 0.
#define RDM_BLOCK(pre, mid, post) \
    asm volatile(SETUP "v_mov_b32 v36, 0.5\n v_mov_b32 v37, 0.5\n v_mov_b32 v38, 0.5\n v_mov_b32 v39, 0.5\n v_mov_b32 v60, 1.0\n v_mov_b32 v61, 1.0\n s_nop 7\n" \
                 "v_mfma_f32_16x16x32_bf16 v[36:39], v[20:23], v[24:27], v[32:35]\n" pre mid post \
                 "v_mov_b32 %0, v36\n v_mov_b32 %1, v37\n v_mov_b32 %2, v38\n v_mov_b32 %3, v39\n s_waitcnt vmcnt(0) lgkmcnt(0)\n s_nop 7\n s_nop 7\n" \
                 : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(ones) : CLOBBERS)
#define MID_F32 "v_mfma_f32_16x16x4_f32 v[44:47], v60, v61, v[44:47]\n"
#define MID_BF  "v_mfma_f32_16x16x32_bf16 v[44:47], v[20:23], v[24:27], v[44:47]\n"
#define NOP0 "s_nop 0\n"
template <int MID, int PRE, int POST>
__global__ __launch_bounds__(512) void rdm_kernel(unsigned long long* __restrict__ bad, int iters) {
    const int lane = threadIdx.x & 63;
    const unsigned ones = 0x3f803f80u;
    unsigned long long wrong = 0;
    for (int it = 0; it < iters; ++it) {
        float d0, d1, d2, d3;
#define RDM_POST(pre, mid) \
        if constexpr (POST == 0) RDM_BLOCK(pre, mid, ""); else if constexpr (POST == 1) RDM_BLOCK(pre, mid, N1(NOP0)); else if constexpr (POST == 2) RDM_BLOCK(pre, mid, N2(NOP0)); \
        else if constexpr (POST == 3) RDM_BLOCK(pre, mid, N3(NOP0)); else if constexpr (POST == 4) RDM_BLOCK(pre, mid, N4(NOP0)); else if constexpr (POST == 5) RDM_BLOCK(pre, mid, N5(NOP0)); \
        else if constexpr (POST == 6) RDM_BLOCK(pre, mid, N6(NOP0)); else RDM_BLOCK(pre, mid, N7(NOP0));
#define RDM_MID(pre) if constexpr (MID == 0) { RDM_POST(pre, MID_F32) } else if constexpr (MID == 1) { RDM_POST(pre, MID_BF) } else { RDM_POST(pre, MID_F32 MID_F32) }
        if constexpr (PRE == 0) { RDM_MID("") } else if constexpr (PRE == 1) { RDM_MID(N1(NOP0)) } else if constexpr (PRE == 2) { RDM_MID(N2(NOP0)) }
        else if constexpr (PRE == 3) { RDM_MID(N3(NOP0)) } else if constexpr (PRE == 4) { RDM_MID(N4(NOP0)) } else { RDM_MID(N5(NOP0)) }
        wrong += (d0 != 33.0f) || (d1 != 33.0f) || (d2 != 33.0f) || (d3 != 33.0f);
    }
    for (int o = 32; o; o >>= 1) wrong += __shfl_xor(wrong, o);
    if (lane == 0) atomicAdd(bad, wrong);
}
template <int MID, int PRE, int POST>
static double run_rdm() {
    const int iters = 5000, blocks = 256;
    CHECK(hipMemset(g_bad, 0, 16));
    hipLaunchKernelGGL((rdm_kernel<MID, PRE, POST>), dim3(blocks), dim3(512), 0, 0, g_bad, iters);
    CHECK(hipDeviceSynchronize());
    unsigned long long bad = 0;
    CHECK(hipMemcpy(&bad, g_bad, 8, hipMemcpyDeviceToHost));
    return 1e6 * (double)bad / ((double)iters * blocks * 512);
}
template <int MID, int PRE>
static void sweep_rdm(const char* what) {
    printf("readm  D read behind %d x s_nop 0, %-22s, M x s_nop 0;  M = 0 1 2 3 4 5 6 7: %8.0f %8.0f %8.0f %8.0f %8.0f %8.0f %8.0f %8.0f   stale per million\n", PRE, what,
           run_rdm<MID, PRE, 0>(), run_rdm<MID, PRE, 1>(), run_rdm<MID, PRE, 2>(), run_rdm<MID, PRE, 3>(), run_rdm<MID, PRE, 4>(), run_rdm<MID, PRE, 5>(), run_rdm<MID, PRE, 6>(), run_rdm<MID, PRE, 7>());
}

// ---- accumulation chains whose destination differs from the accumulator source (D != C: the compiler does this when the incoming
// sum is still needed or the allocator moves the accumulator), back to back and with another matrix instruction / idle slots between
#define HB(d, c) "v_mfma_f32_16x16x32_bf16 " d ", v[20:23], v[24:27], " c "\n"
#define H36 "v[36:39]"
#define H40 "v[40:43]"
#define H48 "v[48:51]"
#define HF "v_mfma_f32_16x16x4_f32 v[44:47], v60, v61, v[44:47]\n"
#define HBI "v_mfma_f32_16x16x32_bf16 v[52:55], v[20:23], v[24:27], v[52:55]\n"
template <int V>
__global__ __launch_bounds__(512) void hop_kernel(unsigned long long* __restrict__ bad, int iters, int jitter) {
    __shared__ float lds_x[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 64) lds_x[threadIdx.x] = 0.0f;
    __syncthreads();
    const unsigned ones = 0x3f803f80u;
    const unsigned lds_addr = (unsigned)(size_t)(&lds_x[0]) + lane * 4;
    unsigned long long wrong = 0;
    float e1 = 0, e2 = 0, e3 = 0;
    for (int it = 0; it < iters; ++it) {
        if (jitter) { const int n = (it * (wave + 3)) & 15; for (int j = 0; j < n; ++j) __builtin_amdgcn_s_sleep(1); }
        float d0, d1, d2, d3, d4, d5;
        // every form ends in v[36:39]; six products: 192
        if constexpr (V == 0) { IL_BLOCK(HB(H40, H36) HB(H36, H40) HB(H40, H36) HB(H36, H40) HB(H40, H36) HB(H36, H40)); e1 = 192; e2 = 160; e3 = 0; }
        else if constexpr (V == 1) { IL_BLOCK(HB(H40, H36) HF HB(H36, H40) HF HB(H40, H36) HF HB(H36, H40) HF HB(H40, H36) HF HB(H36, H40)); e1 = 192; e2 = 160; e3 = 20; }
        else if constexpr (V == 2) { IL_BLOCK(HB(H40, H36) HBI HB(H36, H40) HBI HB(H40, H36) HBI HB(H36, H40) HBI HB(H40, H36) HBI HB(H36, H40)); e1 = 192; e2 = 160; e3 = 0; }
        else if constexpr (V == 3) { IL_BLOCK(HB(H40, H36) FV HB(H36, H40) FV FV HB(H40, H36) FW HB(H36, H40) FD HB(H40, H36) FV FW HB(H36, H40)); e1 = 192; e2 = 160; e3 = 0; }
        else if constexpr (V == 4) { IL_BLOCK(HB(H36, H36) HB(H36, H36) HB(H36, H36) HF HB(H40, H36) HF HB(H36, H40) HB(H36, H36)); e1 = 192; e2 = 128; e3 = 8; }
        else if constexpr (V == 5) { IL_BLOCK(HB(H40, H36) "s_nop 0\n" HB(H36, H40) "s_nop 1\n" HB(H40, H36) "s_nop 2\n" HB(H36, H40) "s_nop 3\n" HB(H40, H36) "s_nop 4\n" HB(H36, H40)); e1 = 192; e2 = 160; e3 = 0; }
        else { IL_BLOCK(HB(H40, H36) HF HF HB(H36, H40) HF HBI HB(H40, H36) HBI HF HB(H36, H40) HF FV HB(H40, H36) FV HF HB(H36, H40)); e1 = 192; e2 = 160; e3 = 24; }
        wrong += (d0 != e1) || (d1 != e1) || (d2 != e2) || (d3 != e2) || (d4 != e3) || (d5 != e3);
    }
    for (int o = 32; o; o >>= 1) wrong += __shfl_xor(wrong, o);
    if (lane == 0) atomicAdd(bad, wrong);
}
template <int V>
static void run_hop(const char* what) {
    const int iters = 10000;
    for (int blocks = 1; blocks <= 1024; blocks *= 32)
        for (int jitter = 0; jitter < 2; ++jitter) {
            CHECK(hipMemset(g_bad, 0, 16));
            hipLaunchKernelGGL((hop_kernel<V>), dim3(blocks), dim3(512), 0, 0, g_bad, iters, jitter);
            CHECK(hipDeviceSynchronize());
            unsigned long long bad = 0;
            CHECK(hipMemcpy(&bad, g_bad, 8, hipMemcpyDeviceToHost));
            printf("hops   %-66s blocks %4d jitter %d   wrong lane-results per million: %12.3f\n", what, blocks, jitter, 1e6 * (double)bad / ((double)iters * blocks * 512));
        }
}

// ---- the same chains with DIFFERENT data in every wave (A = 1 + wave / 2 in wave `wave`, B = 1): a result that picked up anything from
// the other wave of the SIMD (all the probes above run identical data in all waves and cannot see that) differs from 32 * 6 * A.
#define XW_SETUP \
    "v_mov_b32 v20, %6\n v_mov_b32 v21, %6\n v_mov_b32 v22, %6\n v_mov_b32 v23, %6\n" \
    "v_mov_b32 v24, %8\n v_mov_b32 v25, %8\n v_mov_b32 v26, %8\n v_mov_b32 v27, %8\n s_nop 7\n"
#define XW_BLOCK(body) \
    asm volatile(XW_SETUP "v_mov_b32 v36, 0\n v_mov_b32 v37, 0\n v_mov_b32 v38, 0\n v_mov_b32 v39, 0\n v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n" \
                 "v_mov_b32 v44, 0\n v_mov_b32 v45, 0\n v_mov_b32 v46, 0\n v_mov_b32 v47, 0\n v_mov_b32 v60, %9\n v_mov_b32 v61, 1.0\n s_nop 7\n" body \
                 "s_waitcnt vmcnt(0) lgkmcnt(0)\n s_nop 7\n s_nop 7\n s_nop 7\n" \
                 "v_mov_b32 %0, v36\n v_mov_b32 %1, v39\n v_mov_b32 %2, v40\n v_mov_b32 %3, v43\n v_mov_b32 %4, v44\n v_mov_b32 %5, v47\n" \
                 : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5) : "v"(abits), "v"(lds_addr), "v"(ones), "v"(af) : CLOBBERS, "s40")
template <int V>
__global__ __launch_bounds__(512) void xw_kernel(unsigned long long* __restrict__ bad, int iters, int jitter) {
    __shared__ float lds_x[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 64) lds_x[threadIdx.x] = 0.0f;
    __syncthreads();
    const float af = 1.0f + 0.5f * wave;                              // exact in bf16
    const unsigned hi = __float_as_uint(af) >> 16, abits = hi | (hi << 16), ones = 0x3f803f80u;
    const unsigned lds_addr = (unsigned)(size_t)(&lds_x[0]) + lane * 4;
    unsigned long long wrong = 0;
    float e1 = 0, e2 = 0, e3 = 0;
    for (int it = 0; it < iters; ++it) {
        if (jitter) { const int n = (it * (wave + 3)) & 15; for (int j = 0; j < n; ++j) __builtin_amdgcn_s_sleep(1); }
        float d0, d1, d2, d3, d4, d5;
        if constexpr (V == 0) { XW_BLOCK(IB1 IB1 IB1 IB1 IB1 IB1); e1 = 192 * af; }
        else if constexpr (V == 1) { XW_BLOCK(IB1 "s_nop 0\n" IB1 "s_nop 1\n" IB1 "s_nop 3\n" IB1 "s_nop 7\n" IB1 "s_nop 2\n" IB1); e1 = 192 * af; }
        else if constexpr (V == 2) { XW_BLOCK(IB1 FV IB1 FD IB1 FW IB1 FV FV IB1 FD FW IB1); e1 = 192 * af; }
        else if constexpr (V == 3) { XW_BLOCK(IB1 IB2 IB1 IB2 IB1 IB2 IB1 IB2 IB1 IB2 IB1 IB2); e1 = 192 * af; e2 = 192 * af; }
        else if constexpr (V == 4) { XW_BLOCK(IB1 IF3 IB1 IF3 "s_nop 1\n" IB1 FD IF3 IB1 FW IF3 IB1 IF3 FV IB1 IF3); e1 = 192 * af; e3 = 24 * af; }
        else if constexpr (V == 5) { XW_BLOCK(HB(H40, H36) "s_nop 1\n" HB(H36, H40) FD HB(H40, H36) FW HB(H36, H40) "s_nop 5\n" HB(H40, H36) FV HB(H36, H40)); e1 = 192 * af; e2 = 160 * af; }
        else { XW_BLOCK(IB1 "s_nop 7\n" IB1 "s_nop 7\n" IB1 "s_nop 7\n" IB1 "s_nop 7\n" IB1 "s_nop 7\n" IB1); e1 = 192 * af; }
        wrong += (d0 != e1) || (d1 != e1) || (d2 != e2) || (d3 != e2) || (d4 != e3) || (d5 != e3);
    }
    for (int o = 32; o; o >>= 1) wrong += __shfl_xor(wrong, o);
    if (lane == 0) atomicAdd(bad, wrong);
}
template <int V>
static void run_xw(const char* what) {
    const int iters = 20000;
    for (int blocks = 1; blocks <= 1024; blocks *= 32)
        for (int jitter = 0; jitter < 2; ++jitter) {
            CHECK(hipMemset(g_bad, 0, 16));
            hipLaunchKernelGGL((xw_kernel<V>), dim3(blocks), dim3(512), 0, 0, g_bad, iters, jitter);
            CHECK(hipDeviceSynchronize());
            unsigned long long bad = 0;
            CHECK(hipMemcpy(&bad, g_bad, 8, hipMemcpyDeviceToHost));
            printf("xwave  %-62s blocks %4d jitter %d   wrong lane-results per million: %12.3f\n", what, blocks, jitter, 1e6 * (double)bad / ((double)iters * blocks * 512));
        }
}

// ---- transcendental unit: v_rcp_f32 / v_sqrt_f32 results consumed by the next VALU instructions.  The unit runs beside the VALU;
// the compiler's rule for this family is ONE wait state between a TRANS result and a non-TRANS VALU use.  Stale lanes are counted
// per quarter of the wave (the unit writes a wave64 result in four passes of 16 lanes).
#define TR_BLOCK(op, fill) \
    asm volatile("v_mov_b32 v50, 0x42c80000\n v_mov_b32 v51, %1\n v_mov_b32 v53, 1.0\n s_nop 7\n s_nop 7\n" \
                 op " v50, v51\n" fill "v_mul_f32 %0, v50, v53\n s_nop 7\n" : "=&v"(r) : "v"(x) : "v50", "v51", "v53", "v60", "v61", "s40", "memory")
template <int OP, int F>     // OP 0: v_rcp_f32, 1: v_sqrt_f32, 2: v_exp_f32;  F: filler between the TRANS instruction and its reader
__global__ __launch_bounds__(512) void trans_kernel(unsigned long long* __restrict__ bad, int iters, int busy_waves) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= 4 && wave - 4 >= busy_waves) return;                // waves 4..7 are the siblings of 0..3 on their SIMDs
    const float x = 4.0f;
    const float expect = OP == 0 ? 0.25f : OP == 1 ? 2.0f : 16.0f;
    unsigned long long wrong = 0;
    for (int it = 0; it < iters; ++it) {
        float r;
#define TR_F(op) \
        if constexpr (F == 0) TR_BLOCK(op, ""); else if constexpr (F == 1) TR_BLOCK(op, "s_nop 0\n"); else if constexpr (F == 2) TR_BLOCK(op, "s_nop 1\n"); \
        else if constexpr (F == 3) TR_BLOCK(op, "v_mov_b32 v60, 0\n"); else if constexpr (F == 4) TR_BLOCK(op, "v_mov_b32 v60, 0\n v_mov_b32 v61, 0\n"); \
        else if constexpr (F == 5) TR_BLOCK(op, "s_mov_b32 s40, 0\n"); else if constexpr (F == 6) TR_BLOCK(op, "s_waitcnt lgkmcnt(0)\n"); \
        else if constexpr (F == 7) TR_BLOCK(op, "v_rcp_f32 v60, v53\n"); else TR_BLOCK(op, "s_nop 3\n");
        if constexpr (OP == 0) { TR_F("v_rcp_f32") } else if constexpr (OP == 1) { TR_F("v_sqrt_f32") } else { TR_F("v_exp_f32") }
        wrong += (r != expect);
    }
    // per quarter of the wave
    if (wave < 4 && wrong) atomicAdd(bad + (lane >> 4), wrong);
}
static unsigned long long* g_bad4;
template <int OP, int F>
static void run_trans(const char* op, const char* fill) {
    const int iters = 20000, blocks = 256;
    for (int busy = 0; busy <= 4; busy += 4) {
        CHECK(hipMemset(g_bad4, 0, 32));
        hipLaunchKernelGGL((trans_kernel<OP, F>), dim3(blocks), dim3(512), 0, 0, g_bad4, iters, busy);
        CHECK(hipDeviceSynchronize());
        unsigned long long b[4];
        CHECK(hipMemcpy(b, g_bad4, 32, hipMemcpyDeviceToHost));
        const double per = (double)iters * blocks * 4 * 16;
        printf("trans  %-10s; %-26s; v_mul (reads it)   sibling wave %s   stale per million, lanes 0-15 | 16-31 | 32-47 | 48-63: %10.1f %10.1f %10.1f %10.1f\n", op, fill,
               busy ? "busy" : "idle", 1e6 * b[0] / per, 1e6 * b[1] / per, 1e6 * b[2] / per, 1e6 * b[3] / per);
    }
}
template <int OP>
static void sweep_trans(const char* op) {
    run_trans<OP, 0>(op, "(nothing)"); run_trans<OP, 1>(op, "s_nop 0"); run_trans<OP, 2>(op, "s_nop 1"); run_trans<OP, 8>(op, "s_nop 3");
    run_trans<OP, 3>(op, "v_mov_b32 (independent)"); run_trans<OP, 4>(op, "v_mov_b32 x 2"); run_trans<OP, 5>(op, "s_mov_b32");
    run_trans<OP, 6>(op, "s_waitcnt (satisfied)"); run_trans<OP, 7>(op, "v_rcp_f32 (independent)");
}

// ---- the same with the transcendental unit already busy: N independent TRANS instructions right in front of the probed one
#define TQ_BLOCK(pre, fill) \
    asm volatile("v_mov_b32 v50, 0x42c80000\n v_mov_b32 v51, %1\n v_mov_b32 v53, 1.0\n v_mov_b32 v54, 3.0\n s_nop 7\n s_nop 7\n" \
                 pre "v_rcp_f32 v50, v51\n" fill "v_mul_f32 %0, v50, v53\n s_nop 7\n" : "=&v"(r) : "v"(x) : "v50", "v51", "v53", "v54", "v60", "v61", "v62", "v63", "memory")
#define TQ1 "v_rcp_f32 v60, v54\n"
#define TQ2 "v_rcp_f32 v60, v54\n v_sqrt_f32 v61, v54\n"
#define TQ4 "v_rcp_f32 v60, v54\n v_sqrt_f32 v61, v54\n v_exp_f32 v62, v54\n v_rcp_f32 v63, v54\n"
#define TQ8 TQ4 TQ4
template <int PRE, int F>
__global__ __launch_bounds__(512) void transq_kernel(unsigned long long* __restrict__ bad, int iters, int busy_waves) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= 4 && wave - 4 >= busy_waves) return;
    const float x = 4.0f;
    unsigned long long wrong = 0;
    for (int it = 0; it < iters; ++it) {
        float r;
#define TQ_F(pre) \
        if constexpr (F == 0) TQ_BLOCK(pre, ""); else if constexpr (F == 1) TQ_BLOCK(pre, "s_nop 0\n"); else if constexpr (F == 2) TQ_BLOCK(pre, "v_mov_b32 v60, 0\n"); \
        else if constexpr (F == 3) TQ_BLOCK(pre, "s_nop 1\n"); else TQ_BLOCK(pre, "s_nop 3\n");
        if constexpr (PRE == 1) { TQ_F(TQ1) } else if constexpr (PRE == 2) { TQ_F(TQ2) } else if constexpr (PRE == 4) { TQ_F(TQ4) } else { TQ_F(TQ8) }
        wrong += (r != 0.25f);
    }
    if (wave < 4 && wrong) atomicAdd(bad + (lane >> 4), wrong);
}
template <int PRE, int F>
static void run_transq(const char* fill) {
    const int iters = 20000, blocks = 256;
    for (int busy = 0; busy <= 4; busy += 4) {
        CHECK(hipMemset(g_bad4, 0, 32));
        hipLaunchKernelGGL((transq_kernel<PRE, F>), dim3(blocks), dim3(512), 0, 0, g_bad4, iters, busy);
        CHECK(hipDeviceSynchronize());
        unsigned long long b[4];
        CHECK(hipMemcpy(b, g_bad4, 32, hipMemcpyDeviceToHost));
        const double per = (double)iters * blocks * 4 * 16;
        printf("transq %d TRANS in front; v_rcp_f32; %-12s; v_mul   sibling %s   stale per million, lanes 0-15 | 16-31 | 32-47 | 48-63: %10.1f %10.1f %10.1f %10.1f\n", PRE, fill,
               busy ? "busy" : "idle", 1e6 * b[0] / per, 1e6 * b[1] / per, 1e6 * b[2] / per, 1e6 * b[3] / per);
    }
}
template <int PRE>
static void sweep_transq() { run_transq<PRE, 0>("(nothing)"); run_transq<PRE, 1>("s_nop 0"); run_transq<PRE, 2>("v_mov_b32"); run_transq<PRE, 3>("s_nop 1"); run_transq<PRE, 4>("s_nop 3"); }

// ---- write-after-read on a TRANS source: v_rcp_f32 v50, v51 followed (after `fill`) by a VALU write of v51.  If the transcendental
// unit reads its source pass by pass (16 lanes at a time) and the wave's other VALU instructions run on beside it, the late lanes see the new value.
// Sibling waves (4..7) optionally keep the unit busy with their own TRANS instructions.
#define TW_BLOCK(fill) \
    asm volatile("v_mov_b32 v50, 0\n v_mov_b32 v51, %1\n v_mov_b32 v53, 0x42c80000\n s_nop 7\n s_nop 7\n" \
                 "v_rcp_f32 v50, v51\n" fill "v_mov_b32 v51, v53\n s_nop 7\n s_nop 7\n v_mov_b32 %0, v50\n" : "=&v"(r) : "v"(x) : "v50", "v51", "v53", "v60", "v61", "s40", "memory")
template <int F>
__global__ __launch_bounds__(512) void transwar_kernel(unsigned long long* __restrict__ bad, int iters, int busy_waves, float* __restrict__ sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= 4) {
        if (wave - 4 >= busy_waves) return;
        float a = 1.0f + lane * 0.001f;
        for (int it = 0; it < iters * 12; ++it) { a = __builtin_amdgcn_rcpf(a) + 0.5f; a = __builtin_amdgcn_sqrtf(a) + 0.25f; }
        if (a == 123.456f) sink[0] = a;
        return;
    }
    const float x = 4.0f;
    unsigned long long wrong = 0;
    for (int it = 0; it < iters; ++it) {
        float r;
        if constexpr (F == 0) TW_BLOCK(""); else if constexpr (F == 1) TW_BLOCK("s_nop 0\n"); else if constexpr (F == 2) TW_BLOCK("v_mov_b32 v60, 0\n");
        else if constexpr (F == 3) TW_BLOCK("v_mov_b32 v60, 0\n v_mov_b32 v61, 0\n"); else if constexpr (F == 4) TW_BLOCK("s_nop 3\n"); else TW_BLOCK("s_nop 7\n");
        wrong += (r != 0.25f);
    }
    if (wrong) atomicAdd(bad + (lane >> 4), wrong);
}
template <int F>
static void run_transwar(const char* fill) {
    const int iters = 20000, blocks = 256;
    float* sink; CHECK(hipMalloc(&sink, 4));
    for (int busy = 0; busy <= 4; busy += 4) {
        CHECK(hipMemset(g_bad4, 0, 32));
        hipLaunchKernelGGL((transwar_kernel<F>), dim3(blocks), dim3(512), 0, 0, g_bad4, iters, busy, sink);
        CHECK(hipDeviceSynchronize());
        unsigned long long b[4];
        CHECK(hipMemcpy(b, g_bad4, 32, hipMemcpyDeviceToHost));
        const double per = (double)iters * blocks * 4 * 16;
        printf("twar   v_rcp_f32 v50, v51; %-14s; v_mov v51 (overwrite)   sibling %-22s wrong per million, lanes 0-15 | 16-31 | 32-47 | 48-63: %10.1f %10.1f %10.1f %10.1f\n", fill,
               busy ? "running TRANS chains" : "idle", 1e6 * b[0] / per, 1e6 * b[1] / per, 1e6 * b[2] / per, 1e6 * b[3] / per);
    }
    CHECK(hipFree(sink));
}

// ---- VALU compare -> s_and_saveexec: the lane mask a compare writes, consumed by the scalar unit right behind it (every divergent
// `if` compiles to this).  All lanes satisfy the condition: every lane must execute the guarded v_mov.  PRE: what produced the compared value.
#define SX_BLOCK(pre) \
    asm volatile("v_mov_b32 v50, 0\n v_mov_b32 v51, %1\n v_mov_b32 v52, 1.0\n s_mov_b64 vcc, 0\n s_nop 7\n s_nop 7\n" \
                 pre "v_cmp_lt_f32 vcc, v52, v51\n s_and_saveexec_b64 s[40:41], vcc\n v_mov_b32 v50, 1.0\n s_or_b64 exec, exec, s[40:41]\n s_nop 7\n v_mov_b32 %0, v50\n" \
                 : "=&v"(r) : "v"(x) : "v50", "v51", "v52", "v60", "v61", "v62", "v63", "s40", "s41", "vcc", "memory")
template <int PRE>
__global__ __launch_bounds__(512) void saveexec_kernel(unsigned long long* __restrict__ bad, int iters, int busy_waves, float* __restrict__ sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= 4) {
        if (wave - 4 >= busy_waves) return;
        float a = 1.0f + lane * 0.001f;
        for (int it = 0; it < iters * 12; ++it) { a = __builtin_amdgcn_rcpf(a) + 0.5f; a = __builtin_amdgcn_sqrtf(a) + 0.25f; }
        if (a == 123.456f) sink[0] = a;
        return;
    }
    const float x = 4.0f;
    unsigned long long wrong = 0;
    for (int it = 0; it < iters; ++it) {
        float r;
        if constexpr (PRE == 0) SX_BLOCK("");
        else if constexpr (PRE == 1) SX_BLOCK("v_add_f32 v51, v51, v52\n");                                   // VALU result compared at once
        else if constexpr (PRE == 2) SX_BLOCK("v_sqrt_f32 v51, v51\n s_nop 0\n v_add_f32 v51, v51, v52\n");    // TRANS -> VALU -> compare
        else if constexpr (PRE == 3) SX_BLOCK("v_pk_mul_f32 v[60:61], v[50:51], v[50:51]\n v_fma_f32 v51, v51, v52, v52\n v_add_f32 v51, v51, v52\n");
        else if constexpr (PRE == 4) SX_BLOCK("v_rcp_f32 v60, v52\n v_sqrt_f32 v61, v52\n v_exp_f32 v62, v52\n v_add_f32 v51, v51, v52\n");   // TRANS unit busy beside
        else SX_BLOCK("v_sqrt_f32 v51, v51\n s_nop 0\n");                                                      // TRANS result compared (one wait state)
        wrong += (r != 1.0f);
    }
    if (wrong) atomicAdd(bad + (lane >> 4), wrong);
}
template <int PRE>
static void run_saveexec(const char* what) {
    const int iters = 20000, blocks = 256;
    float* sink; CHECK(hipMalloc(&sink, 4));
    for (int busy = 0; busy <= 4; busy += 4) {
        CHECK(hipMemset(g_bad4, 0, 32));
        hipLaunchKernelGGL((saveexec_kernel<PRE>), dim3(blocks), dim3(512), 0, 0, g_bad4, iters, busy, sink);
        CHECK(hipDeviceSynchronize());
        unsigned long long b[4];
        CHECK(hipMemcpy(b, g_bad4, 32, hipMemcpyDeviceToHost));
        const double per = (double)iters * blocks * 4 * 16;
        printf("sexec  %-44s; v_cmp; s_and_saveexec; v_mov   sibling %-5s lanes that skipped the guarded instruction, per million, 0-15 | 16-31 | 32-47 | 48-63: %9.1f %9.1f %9.1f %9.1f\n", what,
               busy ? "busy" : "idle", 1e6 * b[0] / per, 1e6 * b[1] / per, 1e6 * b[2] / per, 1e6 * b[3] / per);
    }
    CHECK(hipFree(sink));
}

// ---- vector memory loads: s_waitcnt vmcnt(0) and a VALU read of the destination right behind it, per quarter of the wave.
// Rows of 12 bytes gathered per lane (as the coordinates are: global_load_dwordx3), the same rows every iteration (cache hits).
#define LD_BLOCK(ld, fill) \
    asm volatile("v_mov_b32 v20, -1.0\n v_mov_b32 v21, -1.0\n v_mov_b32 v22, -1.0\n v_mov_b32 v23, -1.0\n s_nop 7\n" \
                 ld "\n s_waitcnt vmcnt(0)\n" fill "v_add_f32 %0, v20, v20\n v_add_f32 %1, v22, v22\n s_nop 7\n" : "=&v"(r0), "=&v"(r2) : "v"(ptr) : "v20", "v21", "v22", "v23", "memory")
template <int W, int F>
__global__ __launch_bounds__(512) void vload_kernel(const float* __restrict__ src, unsigned long long* __restrict__ bad, int iters, int busy_waves, float* __restrict__ sink, int spread) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= 4) {
        if (wave - 4 >= busy_waves) return;
        float a = 1.0f + lane * 0.001f;
        for (int it = 0; it < iters * 12; ++it) { a = a * 1.0001f + 0.5f; a = a * 0.999f - 0.25f; }
        if (a == 123.456f) sink[0] = a;
        return;
    }
    unsigned long long wrong = 0;
    for (int it = 0; it < iters; ++it) {
        const int row = ((blockIdx.x * 4 + wave) * 64 + lane) * spread + (it & 3);      // 12-byte rows, every value = 2.0
        const float* ptr = src + (size_t)row * 3;
        float r0, r2;
#define LD_F(ld) if constexpr (F == 0) LD_BLOCK(ld, ""); else if constexpr (F == 1) LD_BLOCK(ld, "s_nop 0\n"); else LD_BLOCK(ld, "s_nop 3\n");
        if constexpr (W == 3) { LD_F("global_load_dwordx3 v[20:22], %2, off") }
        else if constexpr (W == 1) { LD_F("global_load_dword v20, %2, off\n global_load_dword v22, %2, off offset:8") }
        else { LD_F("global_load_dwordx2 v[20:21], %2, off\n global_load_dword v22, %2, off offset:8") }
        wrong += (r0 != 4.0f) || (r2 != 4.0f);
    }
    if (wrong) atomicAdd(bad + (lane >> 4), wrong);
}
template <int W, int F>
static void run_vload(const float* src, const char* what, const char* fill) {
    const int iters = 20000, blocks = 256;
    float* sink; CHECK(hipMalloc(&sink, 4));
    for (int spread = 1; spread <= 7; spread += 6)
        for (int busy = 0; busy <= 4; busy += 4) {
            CHECK(hipMemset(g_bad4, 0, 32));
            hipLaunchKernelGGL((vload_kernel<W, F>), dim3(blocks), dim3(512), 0, 0, src, g_bad4, iters, busy, sink, spread);
            CHECK(hipDeviceSynchronize());
            unsigned long long b[4];
            CHECK(hipMemcpy(b, g_bad4, 32, hipMemcpyDeviceToHost));
            const double per = (double)iters * blocks * 4 * 16;
            printf("vload  %-28s; s_waitcnt vmcnt(0); %-8s; v_add   rows %s, sibling %-4s  stale per million, lanes 0-15 | 16-31 | 32-47 | 48-63: %9.1f %9.1f %9.1f %9.1f\n", what, fill,
                   spread == 1 ? "adjacent " : "scattered", busy ? "busy" : "idle", 1e6 * b[0] / per, 1e6 * b[1] / per, 1e6 * b[2] / per, 1e6 * b[3] / per);
        }
    CHECK(hipFree(sink));
}

// ---- packed fp32 on a register pair whose low half came from a vector-memory load and whose high half from a VALU move -- the
// instruction form of the failing coordinate difference (v_pk_add_f32 v[a:b], v[92:93], v[86:87] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1])
template <int V>
__global__ __launch_bounds__(512) void pk_kernel(const float* __restrict__ src, unsigned long long* __restrict__ bad, int iters, int busy_waves, float* __restrict__ sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= 4) {
        if (wave - 4 >= busy_waves) return;
        float a = 1.0f + lane * 0.001f, b = 0.5f;
        for (int it = 0; it < iters * 12; ++it) { a = a * 1.0001f + 0.5f; b = __builtin_amdgcn_rcpf(a) + b; a = a * 0.999f - 0.25f; }
        if (a + b == 123.456f) sink[0] = a;
        return;
    }
    unsigned long long wrong = 0;
    for (int it = 0; it < iters; ++it) {
        const int row = ((blockIdx.x * 4 + wave) * 4 + (lane >> 4)) * 7 + (it & 3);      // 16 lanes share a row (as the 16 edges of a tile share their centre)
        const float* ptr = src + (size_t)row * 3;
        float r0, r1;
        if constexpr (V == 0)
            asm volatile("v_mov_b32 v20, -1.0\n v_mov_b32 v21, -1.0\n v_mov_b32 v22, -1.0\n v_mov_b32 v23, 3.0\n v_mov_b32 v26, 5.0\n v_mov_b32 v27, 7.0\n s_nop 7\n"
                         "global_load_dwordx3 v[20:22], %2, off\n v_mov_b32 v23, 3.0\n s_waitcnt vmcnt(0)\n"
                         "v_pk_add_f32 v[28:29], v[22:23], v[26:27] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n s_nop 7\n v_mov_b32 %0, v28\n v_mov_b32 %1, v29\n"
                         : "=&v"(r0), "=&v"(r1) : "v"(ptr) : "v20", "v21", "v22", "v23", "v26", "v27", "v28", "v29", "memory");
        else
            asm volatile("v_mov_b32 v20, -1.0\n v_mov_b32 v21, -1.0\n v_mov_b32 v22, -1.0\n v_mov_b32 v23, 3.0\n v_mov_b32 v26, 5.0\n v_mov_b32 v27, 7.0\n s_nop 7\n"
                         "global_load_dwordx3 v[20:22], %2, off\n v_mov_b32 v23, 3.0\n s_waitcnt vmcnt(0)\n"
                         "v_pk_mov_b32 v[30:31], v[22:23], v[26:27] op_sel:[1,0]\n v_pk_add_f32 v[28:29], v[22:23], v[26:27] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
                         "v_pk_add_f32 v[32:33], v[20:21], v[30:31] neg_lo:[0,1] neg_hi:[0,1]\n s_nop 7\n v_mov_b32 %0, v28\n v_mov_b32 %1, v32\n"
                         : "=&v"(r0), "=&v"(r1) : "v"(ptr) : "v20", "v21", "v22", "v23", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "memory");
        // memory holds 2.0 everywhere: lo = v22 - v27 = 2 - 7 = -5;  V 0: hi = v23 - v26 = 3 - 5 = -2;  V 1: v32 = v20 - v30 = 2 - v23 = -1
        wrong += (r0 != -5.0f) || (r1 != (V == 0 ? -2.0f : -1.0f));
    }
    if (wrong) atomicAdd(bad + (lane >> 4), wrong);
}
template <int V>
static void run_pk(const float* src, const char* what) {
    const int iters = 50000, blocks = 256;
    float* sink; CHECK(hipMalloc(&sink, 4));
    for (int busy = 0; busy <= 4; busy += 4) {
        CHECK(hipMemset(g_bad4, 0, 32));
        hipLaunchKernelGGL((pk_kernel<V>), dim3(blocks), dim3(512), 0, 0, src, g_bad4, iters, busy, sink);
        CHECK(hipDeviceSynchronize());
        unsigned long long b[4];
        CHECK(hipMemcpy(b, g_bad4, 32, hipMemcpyDeviceToHost));
        const double per = (double)iters * blocks * 4 * 16;
        printf("pk     %-52s sibling %-4s  wrong per million, lanes 0-15 | 16-31 | 32-47 | 48-63: %9.2f %9.2f %9.2f %9.2f\n", what, busy ? "busy" : "idle",
               1e6 * b[0] / per, 1e6 * b[1] / per, 1e6 * b[2] / per, 1e6 * b[3] / per);
    }
    CHECK(hipFree(sink));
}

// ---- packed fp32 beside the sibling wave's matrix instructions: waves 0..3 run v_pk_add_f32 / v_pk_fma_f32 on per-lane data and
// check them against the scalar instructions; waves 4..7 (same SIMDs) run split-bf16 product chains with LDS reads inside, as the
// unfenced kernels do.
template <int HOG>
__global__ __launch_bounds__(512) void pkmfma_kernel(unsigned long long* __restrict__ bad, int iters, float* __restrict__ sink) {
    __shared__ float lds_f[64 * 4 * 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = threadIdx.x; t < 64 * 4 * 8; t += 512) lds_f[t] = __uint_as_float(0x3f803f80u);
    __syncthreads();
    if (wave >= 4) {
        if (HOG == 0) return;
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
        if (HOG >= 3) {
            f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
            float a = 1.0f + lane * 0.001f, b2 = 0.5f;
            for (int it = 0; it < iters * 4; ++it) {
                if (HOG == 3) { c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b2, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b2, a, c1, 0, 0, 0); a = a * 0.999f + c0[0] * 1e-9f; }
                else if (HOG == 4) { a += __shfl_xor(a, 16); b2 += __shfl_xor(b2, 32); a = a * 0.25f + 1.0f; b2 = b2 * 0.25f + 0.5f; }
                else if (HOG == 5) { asm volatile("v_pk_fma_f32 %0, %0, %1, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "+v"(*(double*)&c0) : "v"(*(double*)&c1)); c0[0] = c0[0] * 1e-3f + 1.0f; }
                else if (HOG == 6) { lds_f[(lane * 4 + it) & 2047] = a; a = lds_f[(lane * 4 + 7 * it) & 2047] * 0.5f + 1.0f; }
                else { a = __builtin_amdgcn_rcpf(a) + 0.5f; a = __builtin_amdgcn_sqrtf(a) + 0.25f; a = __expf(-a) + 1.0f; }
            }
            if (a + b2 + c0[0] + c1[0] == 123.456f) sink[0] = a;
            return;
        }
        const bf16x8* f = reinterpret_cast<const bf16x8*>(lds_f);
        bf16x8 b;
        for (int j = 0; j < 8; ++j) b[j] = (__bf16)(1.0f + 0.01f * j);
        f32x4 acc = {0, 0, 0, 0};
        for (int it = 0; it < iters * 3; ++it) {
            if (HOG == 1) {                       // closed groups: reads first, six instructions back to back
                const bf16x8 a0 = f[lane], a1 = f[64 + lane], a2 = f[128 + lane];
                __builtin_amdgcn_sched_barrier(0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b, acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            } else {                              // reads between the instructions
                bf16x8 a = f[lane];
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0); a = f[64 + lane];
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0); a = f[128 + lane];
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0); a = f[192 + lane];
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0); a = f[256 + lane];
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0); a = f[320 + lane];
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
            }
            acc[0] *= 1e-3f;
        }
        if (acc[0] == 123.456f) sink[0] = acc[0];
        return;
    }
    unsigned long long wrong = 0;
    float x0 = 1.0f + lane * 0.25f, x1 = 2.0f + lane * 0.5f, y0 = 0.125f * (lane + 3), y1 = 7.0f - lane;
    for (int it = 0; it < iters; ++it) {
        float lo, hi, lo2, hi2;
        asm volatile("v_mov_b32 v22, %4\n v_mov_b32 v23, %5\n v_mov_b32 v26, %6\n v_mov_b32 v27, %7\n s_nop 1\n"
                     "v_pk_add_f32 v[28:29], v[22:23], v[26:27] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
                     "v_pk_fma_f32 v[30:31], v[22:23], v[26:27], v[22:23]\n s_nop 3\n"
                     "v_mov_b32 %0, v28\n v_mov_b32 %1, v29\n v_mov_b32 %2, v30\n v_mov_b32 %3, v31\n"
                     : "=&v"(lo), "=&v"(hi), "=&v"(lo2), "=&v"(hi2) : "v"(x0), "v"(x1), "v"(y0), "v"(y1) : "v22", "v23", "v26", "v27", "v28", "v29", "v30", "v31");
        wrong += (lo != x0 - y1) || (hi != x1 - y0) || (lo2 != fmaf(x0, y0, x0)) || (hi2 != fmaf(x1, y1, x1));
        x0 += 0.5f; y1 -= 0.25f;
    }
    if (wrong) atomicAdd(bad + (lane >> 4), wrong);
}
template <int HOG>
static void run_pkmfma(const char* what) {
    const int iters = 100000;
    float* sink; CHECK(hipMalloc(&sink, 4));
    for (int blocks = 1; blocks <= 256; blocks *= 256) {
        CHECK(hipMemset(g_bad4, 0, 32));
        hipLaunchKernelGGL((pkmfma_kernel<HOG>), dim3(blocks), dim3(512), 0, 0, g_bad4, iters, sink);
        CHECK(hipDeviceSynchronize());
        unsigned long long b[4];
        CHECK(hipMemcpy(b, g_bad4, 32, hipMemcpyDeviceToHost));
        const double per = (double)iters * blocks * 4 * 16;
        printf("pkmfma packed fp32 beside %-46s blocks %3d  wrong per million, lanes 0-15 | 16-31 | 32-47 | 48-63: %9.2f %9.2f %9.2f %9.2f\n", what, blocks,
               1e6 * b[0] / per, 1e6 * b[1] / per, 1e6 * b[2] / per, 1e6 * b[3] / per);
    }
    CHECK(hipFree(sink));
}

// ---- packed fp32 right behind an EXEC restore (the failing instruction sits two slots behind `s_or_b64 exec, exec, s[..]` that closes a divergent region)
template <int MASKLANES, int F>
__global__ __launch_bounds__(512) void pkexec_kernel(unsigned long long* __restrict__ bad, int iters, int busy_waves, float* __restrict__ sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= 4) {
        if (wave - 4 >= busy_waves) return;
        float a = 1.0f + lane * 0.001f;
        for (int it = 0; it < iters * 10; ++it) { a = a * 1.0001f + 0.5f; a = __builtin_amdgcn_rcpf(a) + 0.25f; }
        if (a == 123.456f) sink[0] = a;
        return;
    }
    unsigned long long wrong = 0;
    float x0 = 1.0f + lane * 0.25f, x1 = 2.0f + lane * 0.5f, y0 = 0.125f * (lane + 3), y1 = 7.0f - lane;
    const unsigned lim = MASKLANES;
    for (int it = 0; it < iters; ++it) {
        float lo, hi;
#define PKX(fill) \
        asm volatile("v_mov_b32 v22, %2\n v_mov_b32 v23, %3\n v_mov_b32 v26, %4\n v_mov_b32 v27, %5\n v_mov_b32 v28, -1.0\n v_mov_b32 v29, -1.0\n s_nop 3\n" \
                     "v_cmp_gt_u32 vcc, %6, %7\n s_and_saveexec_b64 s[40:41], vcc\n v_mov_b32 v60, 1.0\n v_add_f32 v60, v60, v60\n s_or_b64 exec, exec, s[40:41]\n" fill \
                     "v_pk_add_f32 v[28:29], v[22:23], v[26:27] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n s_nop 3\n v_mov_b32 %0, v28\n v_mov_b32 %1, v29\n" \
                     : "=&v"(lo), "=&v"(hi) : "v"(x0), "v"(x1), "v"(y0), "v"(y1), "s"(lim), "v"(lane) : "v22", "v23", "v26", "v27", "v28", "v29", "v60", "s40", "s41", "vcc")
        if constexpr (F == 0) PKX(""); else if constexpr (F == 1) PKX("s_waitcnt lgkmcnt(0)\n"); else PKX("s_waitcnt lgkmcnt(0)\n v_pk_mov_b32 v[30:31], v[22:23], v[26:27] op_sel:[1,0]\n");
        wrong += (lo != x0 - y1) || (hi != x1 - y0);
        x0 += 0.5f; y1 -= 0.25f;
    }
    if (wrong) atomicAdd(bad + (lane >> 4), wrong);
}
template <int MASKLANES, int F>
static void run_pkexec(const char* what) {
    const int iters = 50000, blocks = 256;
    float* sink; CHECK(hipMalloc(&sink, 4));
    for (int busy = 0; busy <= 4; busy += 4) {
        CHECK(hipMemset(g_bad4, 0, 32));
        hipLaunchKernelGGL((pkexec_kernel<MASKLANES, F>), dim3(blocks), dim3(512), 0, 0, g_bad4, iters, busy, sink);
        CHECK(hipDeviceSynchronize());
        unsigned long long b[4];
        CHECK(hipMemcpy(b, g_bad4, 32, hipMemcpyDeviceToHost));
        const double per = (double)iters * blocks * 4 * 16;
        printf("pkexec %-64s sibling %-4s  wrong per million, lanes 0-15 | 16-31 | 32-47 | 48-63: %9.2f %9.2f %9.2f %9.2f\n", what, busy ? "busy" : "idle",
               1e6 * b[0] / per, 1e6 * b[1] / per, 1e6 * b[2] / per, 1e6 * b[3] / per);
    }
    CHECK(hipFree(sink));
}

// ---- the SAME wave alternating split-bf16 product chains and packed fp32 on per-lane data (what the stripped failing kernel still does:
// front phase with packed fp32 -> sub-tiles with bf16 products -> back phase -> next front phase), 8 waves per workgroup, two per SIMD
template <int NMFMA, int GAP, int REGS = 0>
__global__ __launch_bounds__(512) void alt_kernel(unsigned long long* __restrict__ bad, int iters, float* __restrict__ sink) {
    __shared__ float lds_f[64 * 4 * 8];
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = threadIdx.x; t < 64 * 4 * 8; t += 512) lds_f[t] = __uint_as_float(0x3f803f80u);
    __syncthreads();
    const bf16x8* f = reinterpret_cast<const bf16x8*>(lds_f);
    bf16x8 b;
    for (int j = 0; j < 8; ++j) b[j] = (__bf16)(1.0f + 0.01f * j);
    f32x4 acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
    unsigned long long wrong = 0;
    float x0 = 1.0f + lane * 0.25f, x1 = 2.0f + lane * 0.5f, y0 = 0.125f * (lane + 3), y1 = 7.0f - lane;
    for (int it = 0; it < iters; ++it) {
        // phase A: matrix chains with LDS reads between
        for (int r = 0; r < NMFMA; ++r) {
            bf16x8 a = f[((r * 3 + wave) & 7) * 64 + lane];
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
            a = f[((r * 5 + 1) & 7) * 64 + lane];
            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc2, 0, 0, 0);
        }
        if constexpr (REGS != 2) { acc[0] *= 1e-3f; acc2[1] *= 1e-3f; }      // REGS == 2: the products' results are not read: the packed ops run while they execute
        for (int j = 0; j < GAP; ++j) __builtin_amdgcn_s_sleep(1);
        // phase B: packed fp32 with operand selects, against the scalar results
        float lo, hi, lo2, hi2, lo3, hi3;
        if constexpr (REGS == 0 || REGS == 2)
        asm volatile("v_mov_b32 v22, %6\n v_mov_b32 v23, %7\n v_mov_b32 v26, %8\n v_mov_b32 v27, %9\n s_nop 1\n"
                     "v_pk_mov_b32 v[32:33], v[22:23], v[26:27] op_sel:[1,0]\n"
                     "v_pk_add_f32 v[28:29], v[22:23], v[26:27] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
                     "v_pk_add_f32 v[34:35], v[22:23], v[32:33] neg_lo:[0,1] neg_hi:[0,1]\n"
                     "v_pk_fma_f32 v[30:31], v[28:29], v[28:29], v[22:23] op_sel_hi:[1,1,0]\n s_nop 3\n"
                     "v_mov_b32 %0, v28\n v_mov_b32 %1, v29\n v_mov_b32 %2, v30\n v_mov_b32 %3, v31\n v_mov_b32 %4, v34\n v_mov_b32 %5, v35\n"
                     : "=&v"(lo), "=&v"(hi), "=&v"(lo2), "=&v"(hi2), "=&v"(lo3), "=&v"(hi3) : "v"(x0), "v"(x1), "v"(y0), "v"(y1)
                     : "v22", "v23", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35");
        else        // the failing build's own registers (v[92:93] - v[86:87] -> v[122:123]: all four VGPR banks in one instruction)
        asm volatile("v_mov_b32 v92, %6\n v_mov_b32 v93, %7\n v_mov_b32 v86, %8\n v_mov_b32 v87, %9\n v_mov_b32 v90, %6\n v_mov_b32 v91, %7\n s_nop 1\n"
                     "v_pk_mov_b32 v[96:97], v[92:93], v[86:87] op_sel:[1,0]\n"
                     "v_pk_add_f32 v[122:123], v[92:93], v[86:87] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n"
                     "v_pk_add_f32 v[116:117], v[90:91], v[96:97] neg_lo:[0,1] neg_hi:[0,1]\n"
                     "v_pk_fma_f32 v[94:95], v[122:123], v[122:123], v[92:93] op_sel_hi:[1,1,0]\n s_nop 3\n"
                     "v_mov_b32 %0, v122\n v_mov_b32 %1, v123\n v_mov_b32 %2, v94\n v_mov_b32 %3, v95\n v_mov_b32 %4, v116\n v_mov_b32 %5, v117\n"
                     : "=&v"(lo), "=&v"(hi), "=&v"(lo2), "=&v"(hi2), "=&v"(lo3), "=&v"(hi3) : "v"(x0), "v"(x1), "v"(y0), "v"(y1)
                     : "v86", "v87", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v116", "v117", "v122", "v123");
        const float e_lo = x0 - y1, e_hi = x1 - y0;
        wrong += (lo != e_lo) || (hi != e_hi) || (lo2 != fmaf(e_lo, e_lo, x0)) || (hi2 != fmaf(e_hi, e_hi, x0)) || (lo3 != x0 - x1) || (hi3 != x1 - y0);
        x0 += 0.5f; y1 -= 0.25f;
    }
    if (acc[0] + acc2[1] == 123.456f) sink[0] = acc[0];
    if (wrong) atomicAdd(bad + (lane >> 4), wrong);
}
template <int NMFMA, int GAP, int REGS = 0>
static void run_alt(const char* what) {
    const int iters = 30000;
    float* sink; CHECK(hipMalloc(&sink, 4));
    for (int blocks = 1; blocks <= 256; blocks *= 256) {
        CHECK(hipMemset(g_bad4, 0, 32));
        hipLaunchKernelGGL((alt_kernel<NMFMA, GAP, REGS>), dim3(blocks), dim3(512), 0, 0, g_bad4, iters, sink);
        CHECK(hipDeviceSynchronize());
        unsigned long long b[4];
        CHECK(hipMemcpy(b, g_bad4, 32, hipMemcpyDeviceToHost));
        const double per = (double)iters * blocks * 8 * 16;
        printf("alt    %-58s blocks %3d  wrong per million, lanes 0-15 | 16-31 | 32-47 | 48-63: %9.2f %9.2f %9.2f %9.2f\n", what, blocks,
               1e6 * b[0] / per, 1e6 * b[1] / per, 1e6 * b[2] / per, 1e6 * b[3] / per);
    }
    CHECK(hipFree(sink));
}

int main() {
    CHECK(hipMalloc(&g_junk, 64 * 4 * sizeof(float)));
    CHECK(hipMalloc(&g_bad, 16));
    std::vector<float> h(256, 3.0e30f);
    CHECK(hipMemcpy(g_junk, h.data(), 256 * sizeof(float), hipMemcpyHostToDevice));
    printf("# v_mfma_f32_16x16x32_bf16 source overwrite probe: chain form, overwritten source, writer, idle slots between, matrix-pipe hog on the SIMD\n");
    {
        CHECK(hipMemset(g_bad, 0, 16));
        hipLaunchKernelGGL(control_kernel, dim3(64), dim3(256), 0, 0, g_bad, 1000, 97.0f);
        CHECK(hipDeviceSynchronize());
        unsigned long long bad = 0;
        CHECK(hipMemcpy(&bad, g_bad, 8, hipMemcpyDeviceToHost));
        printf("control (overwrite BEFORE the chain): wrong lane-results per million: %12.3f   (must be 1000000)\n", 1e6 * (double)bad / (1000.0 * 64 * 256));
    }
    sweep_raw<T_A, 0>("A", "v_mov"); sweep_raw<T_B, 0>("B", "v_mov"); sweep_raw<T_C, 0>("C", "v_mov");
    sweep_raw<T_A, 1>("A", "v_perm"); sweep_raw<T_B, 1>("B", "v_perm"); sweep_raw<T_C, 1>("C", "v_perm");
    sweep_raw<T_A, 2>("A", "ds_read"); sweep_raw<T_B, 2>("B", "ds_read"); sweep_raw<T_C, 2>("C", "ds_read");
    CHECK(hipMalloc(&g_bad4, 32));
    run_alt<6, 0, 2>("12 bf16 products IN FLIGHT, packed fp32 right behind them"); run_alt<2, 0, 2>("4 products in flight, packed fp32 behind"); run_alt<24, 0, 2>("48 products in flight, packed fp32 behind");
    run_alt<6, 0, 1>("12 bf16 products, then packed fp32 in the failing build's registers"); run_alt<24, 2, 1>("48 products, sleep, the failing build's registers");
    run_alt<6, 0>("12 bf16 products, then packed fp32, same wave"); run_alt<24, 0>("48 bf16 products, then packed fp32"); run_alt<6, 3>("12 products, short sleep, packed fp32"); run_alt<1, 0>("2 products, packed fp32");
    run_pkexec<48, 0>("region over lanes 0..47; s_or exec; v_pk_add"); run_pkexec<48, 1>("region over lanes 0..47; s_or exec; s_waitcnt; v_pk_add");
    run_pkexec<48, 2>("region over lanes 0..47; s_or exec; s_waitcnt; v_pk_mov; v_pk_add"); run_pkexec<64, 2>("region over all lanes; s_or exec; s_waitcnt; v_pk_mov; v_pk_add");
    run_pkexec<0, 2>("region over no lane (exec 0); s_or exec; s_waitcnt; v_pk_mov; v_pk_add"); run_pkexec<16, 0>("region over lanes 0..15; s_or exec; v_pk_add");
    run_pkmfma<3>("a sibling running fp32 16x16x4 chains"); run_pkmfma<4>("a sibling running cross-lane shuffles"); run_pkmfma<5>("a sibling running packed fp32");
    run_pkmfma<6>("a sibling running LDS traffic"); run_pkmfma<7>("a sibling running TRANS chains");
    run_pkmfma<0>("an idle sibling"); run_pkmfma<1>("a sibling running closed product groups"); run_pkmfma<2>("a sibling running products with LDS reads inside");
    {
        const size_t nrow = (size_t)256 * 4 * 64 * 7 + 8;
        float* src; CHECK(hipMalloc(&src, nrow * 3 * sizeof(float)));
        std::vector<float> hs(nrow * 3, 2.0f);
        CHECK(hipMemcpy(src, hs.data(), nrow * 3 * sizeof(float), hipMemcpyHostToDevice));
        run_pk<0>(src, "dwordx3 load; v_pk_add (op_sel) on the pair"); run_pk<1>(src, "dwordx3 load; v_pk_mov; v_pk_add; v_pk_add (as compiled)");
        run_vload<3, 0>(src, "global_load_dwordx3", "(none)"); run_vload<3, 1>(src, "global_load_dwordx3", "s_nop 0"); run_vload<3, 2>(src, "global_load_dwordx3", "s_nop 3");
        run_vload<1, 0>(src, "global_load_dword x 2", "(none)"); run_vload<2, 0>(src, "global_load_dwordx2 + dword", "(none)");
        CHECK(hipFree(src));
    }
    run_saveexec<0>("(compared value ready)"); run_saveexec<1>("v_add (produces the compared value)"); run_saveexec<2>("v_sqrt; s_nop 0; v_add"); run_saveexec<3>("v_pk_mul; v_fma; v_add");
    run_saveexec<4>("v_rcp; v_sqrt; v_exp (independent); v_add"); run_saveexec<5>("v_sqrt; s_nop 0 (TRANS result compared)");
    run_transwar<0>("(nothing)"); run_transwar<1>("s_nop 0"); run_transwar<2>("v_mov_b32"); run_transwar<3>("v_mov_b32 x 2"); run_transwar<4>("s_nop 3"); run_transwar<5>("s_nop 7");
    sweep_transq<1>(); sweep_transq<2>(); sweep_transq<4>(); sweep_transq<8>();
    sweep_trans<0>("v_rcp_f32"); sweep_trans<1>("v_sqrt_f32"); sweep_trans<2>("v_exp_f32");
    run_xw<0>("per-wave data: chain back to back"); run_xw<1>("per-wave data: chain with 1..8 idle slots between"); run_xw<2>("per-wave data: chain with VALU / LDS / waitcnt between");
    run_xw<3>("per-wave data: two chains alternating"); run_xw<4>("per-wave data: bf16 chain mixed with an fp32 chain + fillers"); run_xw<5>("per-wave data: D != C hops with fillers"); run_xw<6>("per-wave data: chain with 8 idle slots between");
    run_hop<0>("D != C hops, back to back"); run_hop<1>("D != C hops, an fp32 16x16x4 instruction between"); run_hop<2>("D != C hops, an independent bf16 instruction between");
    run_hop<3>("D != C hops, VALU / waitcnt / LDS between"); run_hop<4>("D == C then hops with fp32 between"); run_hop<5>("D != C hops, 1..5 idle slots between"); run_hop<6>("D != C hops, two instructions of mixed kinds between");
    run_il<5>("two bf16 chains one after the other (control)"); run_il<0>("two bf16 chains alternating"); run_il<1>("two bf16 chains alternating, VALU / LDS / waitcnt between");
    run_il<2>("bf16 chain alternating with an fp32 16x16x4 chain"); run_il<3>("two bf16 chains and an fp32 chain mixed"); run_il<4>("the same with VALU / LDS / waitcnt between");
    sweep_rdm<0, 0>("one fp32 16x16x4"); sweep_rdm<0, 2>("one fp32 16x16x4"); sweep_rdm<0, 4>("one fp32 16x16x4"); sweep_rdm<0, 5>("one fp32 16x16x4");
    sweep_rdm<1, 0>("one bf16 16x16x32"); sweep_rdm<1, 2>("one bf16 16x16x32"); sweep_rdm<1, 4>("one bf16 16x16x32");
    sweep_rdm<2, 0>("two fp32 16x16x4"); sweep_rdm<2, 2>("two fp32 16x16x4"); sweep_rdm<2, 4>("two fp32 16x16x4");
    sweep_rd<0>("s_nop 0"); sweep_rd<1>("s_waitcnt (satisfied)"); sweep_rd<2>("v_mov_b32 (independent)"); sweep_rd<3>("ds_read_b32 (unrelated)"); sweep_rd<4>("s_mov_b32");
    sweep_warv<20, 0>("A[0]"); sweep_warv<21, 0>("A[1]"); sweep_warv<22, 0>("A[2]"); sweep_warv<23, 0>("A[3]");
    sweep_warv<24, 0>("B[0]"); sweep_warv<25, 0>("B[1]"); sweep_warv<26, 0>("B[2]"); sweep_warv<27, 0>("B[3]");
    sweep_warv<32, 0>("C[0]"); sweep_warv<33, 0>("C[1]"); sweep_warv<34, 0>("C[2]"); sweep_warv<35, 0>("C[3]");
    sweep_warv<20, 1>("A[0]"); sweep_warv<23, 1>("A[3]"); sweep_warv<24, 1>("B[0]"); sweep_warv<27, 1>("B[3]");
    run_fill<0>("(nothing)"); run_fill<1>("s_nop 0; s_nop 0"); run_fill<2>("s_nop 1"); run_fill<3>("s_waitcnt lgkmcnt(0) x 2 (nothing outstanding)");
    run_fill<4>("ds_read_b32 (unrelated); s_waitcnt lgkmcnt(1) (satisfied)"); run_fill<5>("ds_read_b32 x 2 (unrelated)"); run_fill<6>("s_mov_b32 x 2");
    run_fill<7>("v_mov_b32 x 2 (unrelated)"); run_fill<8>("v_mov_b32 (unrelated); s_waitcnt (satisfied)"); run_fill<9>("global_load_dword (unrelated); s_waitcnt vmcnt(1) (satisfied)");
    run_fill<10>("v_mfma (independent)"); run_fill<11>("ds_read_b32; s_waitcnt (satisfied) x 2"); run_fill<12>("ds_read_b32 x 2; s_waitcnt (satisfied)");
    run_fill<13>("v_mov_b32 (unrelated)"); run_fill<14>("v_mov_b32 (unrelated); ds_read_b32"); run_fill<15>("s_nop 0; ds_read_b32");
    run_real<0>(1); run_real<0>(256); run_real<0>(2048);
    run_real<4>(1); run_real<4>(256); run_real<4>(2048);
    chain_sweep<CH_DEP0>("dep0", 33.0f);
    chain_sweep<CH_DEP2>("dep2", 97.0f);
    chain_sweep<CH_DEP5>("dep5", 193.0f);
    chain_sweep<CH_DEP11>("dep11", 385.0f);
    chain_sweep<CH_IND5>("ind5", 33.0f);
    return 0;
}
