// k-NN graph build + inverse-index (CSR) build for gfx950.
//
// knn: one 64-lane wavefront per query point, brute force over the query's cloud (N <= 8192
// -> <= 128 candidates per lane held in VGPRs).  Selection is exact and deterministic:
//   fast path : tau = k-th smallest of the 64 per-lane minima (an upper bound of the k-th
//               distance), compact the (usually ~1.2k) candidates <= tau through LDS, rank them
//               by counting on their (d2,index) keys (one key per lane up to 64 candidates, two up to
//               128), the k smallest write their output slots.
//   slow path : k rounds of wave-wide min extraction (taken when more than 128 candidates
//               tie under tau, e.g. heavily duplicated points; also forced by tests).
// Both paths produce the (d2, index)-lexicographic order of oracle.knn_graph bit for bit.
//
// Replaces torch_cluster.knn_graph at src/3dmatch_train_egnn_with_batch.py:1005-1006.
#include "common.h"
#include "scan.h"

namespace eq {

typedef unsigned long long u64;

// k-th smallest (1-based) of the wave's 64 non-negative floats given as bit patterns (unsigned order = float order): bisection on
// the value with one ballot + popcount per step -- one vector instruction per step against ~300 for a 64-lane bitonic sort.
__device__ __forceinline__ unsigned wave_kth_smallest_bits(unsigned bits, int k) {
    unsigned lo = 0u, hi = 0x7f800000u;                 // +inf: every lane counts
    while (lo < hi) {                                   // wave-uniform
        const unsigned mid = lo + ((hi - lo) >> 1);
        const int c = __popcll(__ballot(bits <= mid));
        if (c >= k) hi = mid;
        else lo = mid + 1;
    }
    return lo;
}

__device__ __forceinline__ u64 wave_min_u64(u64 v) {
#pragma unroll
    for (int j = 32; j > 0; j >>= 1) {
        const u64 o = __shfl_xor(v, j, 64);
        v = o < v ? o : v;
    }
    return v;
}

__device__ __forceinline__ u64 pack_key(float d2, int idx) {
    return ((u64)__float_as_uint(d2) << 32) | (unsigned)idx;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Squared distances of TWO (query, candidate) pairs at once in the arithmetic contract of oracle.knn_graph --
// d2 = fl(fl(fl(dx dx) + fl(dy dy)) + fl(dz dz)), no contraction -- as packed fp32 instructions (v_pk_add_f32 / v_pk_mul_f32: two
// IEEE results per instruction, issued in the time of one; tools/dev/valu_cost.hip).  The difference is formed as p + (-q):
// (p - q)^2 and (q - p)^2 are the same number.  `nq*` = the NEGATED query coordinates.
__device__ __forceinline__ f32x2 dist2_pk(f32x2 px, f32x2 py, f32x2 pz, f32x2 nqx, f32x2 nqy, f32x2 nqz) {
#pragma clang fp contract(off)
    const f32x2 dx = px + nqx, dy = py + nqy, dz = pz + nqz;
    const f32x2 xx = dx * dx, yy = dy * dy, zz = dz * dz;
    const f32x2 s = xx + yy;
    return s + zz;
}
// One pair, the same contract.  (NOT __fsub_rn / __fmul_rn / __fadd_rn: in this toolchain they are plain `x - y`, `x * y`, `x + y`
// (__clang_hip_math.h), which hipcc's default -ffp-contract=fast fuses into FMAs once inlined -- keys an ulp away from the contract.)
__device__ __forceinline__ float dist2_scalar(float px, float py, float pz, float qx, float qy, float qz) {
#pragma clang fp contract(off)
    const float dx = px - qx, dy = py - qy, dz = pz - qz;
    const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    const float s = xx + yy;
    return s + zz;
}
// position of this lane among the set bits of a ballot mask (= popcount of the mask below the lane): two v_mbcnt
__device__ __forceinline__ int rank_in_mask(u64 mask) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// The cnt (<= 64) compacted candidates of a query, already in LDS: lane l < cnt takes key l and counts the keys smaller than its
// own (uniform-address LDS reads = broadcasts; keys are distinct: the index is part of the key).  A lane whose rank is < k owns
// output slot `rank` -- nothing is physically sorted (a 64-lane bitonic network on 64-bit keys is ~250 vector instructions and
// 42 cross-lane moves; this is ~2 per candidate).  Returns the rank (>= 64 for lanes without a key).
__device__ __forceinline__ int rank_by_counting(const u64* list, int cnt, int lane, u64& mine) {
    mine = lane < cnt ? list[lane] : ~0ull;
    int rank = lane < cnt ? 0 : 64;
    int j = 0;
    for (; j + 4 <= cnt; j += 4) {                      // cnt is wave-uniform
        const u64 a = list[j], b = list[j + 1], c = list[j + 2], d = list[j + 3];
        rank += (a < mine ? 1 : 0) + (b < mine ? 1 : 0) + (c < mine ? 1 : 0) + (d < mine ? 1 : 0);
    }
    for (; j < cnt; ++j) rank += list[j] < mine ? 1 : 0;
    return rank;
}

// 64 < cnt <= 128 compacted candidates (k = 32 of 64 lane minima leaves ~44 candidates on average and a tail beyond 64 in a few
// queries of every other 8 192-point cloud -- each of which used to take the slow path, ~0.6 ms): lane l ranks keys l and l + 64.
__device__ __forceinline__ void rank_by_counting2(const u64* list, int cnt, int lane, u64& k0, u64& k1, int& r0, int& r1) {
    k0 = list[lane];                                    // cnt > 64: every lane has a first key
    k1 = lane + 64 < cnt ? list[lane + 64] : ~0ull;
    r0 = 0;
    r1 = lane + 64 < cnt ? 0 : 128;
    for (int j = 0; j < cnt; ++j) {                     // cnt is wave-uniform
        const u64 a = list[j];
        r0 += a < k0 ? 1 : 0;
        r1 += a < k1 ? 1 : 0;
    }
}
constexpr int kKnnList = 128;                           // compaction list entries per query

// kLoop: the query itself is a candidate (loop=True, every reference call site); kFull: n == 64 NPL (no index guards)
template <int NPL, bool kLoop, bool kFull>
// (32 candidates per lane: register budget of EXACTLY five waves per SIMD -- 95 instead of 97 registers, nothing spilled: sixteen 2 048-point pairs 88.5 -> 80.5 us per launch;
// six waves spill five registers: 94 us)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NPL >= 32 ? 5 : 1, NPL >= 32 ? 5 : 8))) void knn_kernel(const float* __restrict__ x, int n_clouds, int n, int k,
                                                 long long* __restrict__ edges,
                                                 int* __restrict__ row32, int* __restrict__ col32,
                                                 int* __restrict__ row_hist, int* __restrict__ arrival, int force_slow) {
    static_assert(NPL % 2 == 0, "candidates are processed in pairs");
    __shared__ u64 list[kKnnList];
    const int lane = threadIdx.x;
    const int qg = blockIdx.x;                 // query id over all clouds
    const int cloud = qg / n;
    const int q = qg - cloud * n;
    const float* __restrict__ xc = x + (size_t)cloud * n * 3;
    const float qx = xc[q * 3 + 0], qy = xc[q * 3 + 1], qz = xc[q * 3 + 2];
    const f32x2 nqx = {-qx, -qx}, nqy = {-qy, -qy}, nqz = {-qz, -qz};
    const float INF = __uint_as_float(0x7f800000u);

    float d[NPL];
    float lmin = INF;
#pragma unroll
    for (int i = 0; i < NPL; i += 2) {
        const int i0 = i * 64 + lane, i1 = i0 + 64;
        const int j0 = kFull ? i0 : min(i0, n - 1), j1 = kFull ? i1 : min(i1, n - 1);
        const f32x2 v = dist2_pk(f32x2{xc[j0 * 3 + 0], xc[j1 * 3 + 0]}, f32x2{xc[j0 * 3 + 1], xc[j1 * 3 + 1]},
                                 f32x2{xc[j0 * 3 + 2], xc[j1 * 3 + 2]}, nqx, nqy, nqz);
        float v0 = v[0], v1 = v[1];
        if (!kFull) { v0 = i0 < n ? v0 : INF; v1 = i1 < n ? v1 : INF; }
        if (!kLoop) { v0 = i0 == q ? INF : v0; v1 = i1 == q ? INF : v1; }
        d[i] = v0; d[i + 1] = v1;
        lmin = fminf(lmin, fminf(v0, v1));
    }

    int out_idx = 0;
    // ---- fast path: tau = k-th smallest lane minimum (at least k candidates are <= tau); count and compact in one sweep
    const unsigned tau_bits = wave_kth_smallest_bits(__float_as_uint(lmin), k);
    const float tau = __uint_as_float(tau_bits);
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const bool pred = d[i] <= tau;
        const u64 mask = __ballot(pred);
        if (mask) {                                     // wave-uniform: most 64-candidate rows hold no candidate <= tau
            const int at = cnt + rank_in_mask(mask);
            if (pred && at < kKnnList) list[at] = pack_key(d[i], i * 64 + lane);
            cnt += __popcll(mask);
        }
    }
    const bool fast = (cnt <= kKnnList) && !force_slow;   // wave-uniform
    int slot = lane;                                // output slot of this lane's out_idx
    int slot2 = 128, out_idx2 = 0;                  // second (slot, index) of a lane when 64 < cnt <= 128
    if (fast) {
        __syncthreads();   // single-wave block: orders the LDS writes before the reads below
        if (cnt <= 64) {
            u64 key;
            slot = rank_by_counting(list, cnt, lane, key);
            out_idx = (int)(unsigned)(key & 0xffffffffu);
        } else {
            u64 k0, k1;
            rank_by_counting2(list, cnt, lane, k0, k1, slot, slot2);
            out_idx = (int)(unsigned)(k0 & 0xffffffffu);
            out_idx2 = (int)(unsigned)(k1 & 0xffffffffu);
        }
    } else {
        // ---- slow path: k rounds of "smallest key strictly greater than the previous one"
        u64 last = 0;
        bool have_last = false;
        for (int t = 0; t < k; ++t) {
            u64 best = ~0ull;
#pragma unroll
            for (int i = 0; i < NPL; ++i) {
                const u64 key = pack_key(d[i], i * 64 + lane);
                const bool ok = !have_last || key > last;
                best = (ok && key < best) ? key : best;
            }
            best = wave_min_u64(best);
            if (lane == t) out_idx = (int)(unsigned)(best & 0xffffffffu);
            last = best;
            have_last = true;
        }
    }
    auto emit = [&](int slot_, int idx_) {
        if (slot_ >= k) return;
        const size_t e = (size_t)q * k + slot_;
        const size_t epc = (size_t)n * k;
        if (edges) {
            edges[(size_t)cloud * 2 * epc + e] = idx_;
            edges[(size_t)cloud * 2 * epc + epc + e] = q;
        }
        if (row32) row32[(size_t)cloud * epc + e] = cloud * n + idx_;
        if (col32) col32[(size_t)cloud * epc + e] = cloud * n + q;
        if (row_hist) {                                  // the key histogram build_csr(key = row) starts from; its old value = this edge's slot
            const int slot_in_row = atomicAdd(&row_hist[cloud * n + idx_], 1);
            if (arrival) arrival[(size_t)cloud * epc + e] = slot_in_row;
        }
    };
    emit(slot, out_idx);
    emit(slot2, out_idx2);
}

// ---------------------------------------------------------------------------------------
// Tiled multi-query form (used for n > 2 048): the per-query kernel above re-reads the whole cloud from L2 for every query
// (and its 128-candidates-per-lane variant spills).  Here a workgroup of 4 waves owns 16 (32) queries of one cloud; candidate tiles
// of 2 048 points are staged through LDS once per workgroup and pass.  Same lane <-> candidate assignment (candidate 64 s + l of a
// tile belongs to lane l), same arithmetic, same key order as knn_kernel, so the results are bit-identical.
//   pass 1: every candidate read from LDS is tested against the wave's 4 (8) queries, two queries per packed instruction; a lane
//           keeps the minimum of each 16-candidate CHUNK it owns (2 per tile: s < 16, s >= 16) -> lane minima -> tau.
//   pass 2: only the chunks whose minimum is <= tau can hold a candidate <= tau: ~1 chunk per candidate, ~30-45 of the cloud's
//           256-512.  Per query, the live chunks of a tile (two ballots) are visited four per step (16 lanes each) and their
//           candidates <= tau compacted -- ~8-14 steps per query instead of re-scanning all 64-128 (with the point-wise staging
//           below: two 8 192-point clouds, k = 32: 102 -> 76 us per launch; sixteen 4 096-point clouds, k = 20: 175 -> 136 us).
// The tile is stored lane-major and rotated: candidate 64 s + l at point 32 l + ((s + l) & 31) -- pass 1 (lane l reads its s-th
// candidate) and pass 2 (16 lanes read the 16 candidates of a chunk) are both free of LDS bank conflicts without padding (one
// padded point per lane had cost the 8-queries-per-wave variant a workgroup per CU).
// The (rare) slow path of a query re-scans the cloud from global memory with its own wave.
// ---------------------------------------------------------------------------------------
constexpr int kKnnTile = 2048;
__device__ __forceinline__ int knn_tile_point(int l, int s) { return 32 * l + ((s + l) & 31); }      // of candidate 64 s + l
// kKnnQpw = queries per wave (even: two per packed instruction): 8 when that still leaves >= 8 workgroups per CU (measured:
// KITTI shard 197 -> 171 us), 4 otherwise (two 8 192-point clouds: 99 us against 122 with 8 and 109 with 2)
// kCap = entries of a query's compaction list: 64 (one key per lane) while k <= 24 -- the 24th of 64 lane minima leaves ~30 candidates,
// > 64 practically never -- and 128 (two keys per lane, rank_by_counting2) above, where a k = 32 query keeps ~44 on average and a few
// of every 16 384 exceed 64.  A template parameter because the list is LDS: 128 entries for the 32 queries of the 8-per-wave variant
// cost it half its occupancy (KITTI shard, k = 20: 173 -> 264 us when every launch paid for them).
// kTiles = tiles of the cloud (2: n <= 4 096, 4: n <= 8 192): the chunk minima are registers, 2 kTiles per query.
template <bool kLoop, int kKnnQpw, int kCap, int kTiles>
__global__ __launch_bounds__(256) void knn_tiled_kernel(const float* __restrict__ x, int n_clouds, int n, int k,
                                                        long long* __restrict__ edges, int* __restrict__ row32,
                                                        int* __restrict__ col32, int* __restrict__ row_hist,
                                                        int* __restrict__ arrival, int force_slow) {
    __shared__ float tile[kKnnTile * 3];
    __shared__ u64 list[4 * kKnnQpw][kCap];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform BY CONSTRUCTION: the query ids and pass 2's chunk masks stay scalar
    const int bpc = (n + 4 * kKnnQpw - 1) / (4 * kKnnQpw);      // workgroups per cloud
    const int cloud = blockIdx.x / bpc;
    const int q0 = (blockIdx.x - cloud * bpc) * 4 * kKnnQpw + wv * kKnnQpw;
    const float* __restrict__ xc = x + (size_t)cloud * n * 3;
    const float INF = __uint_as_float(0x7f800000u);
    float qx[kKnnQpw], qy[kKnnQpw], qz[kKnnQpw];
    int qi[kKnnQpw];
#pragma unroll
    for (int u = 0; u < kKnnQpw; ++u) {
        qi[u] = q0 + u;
        const int qq = min(qi[u], n - 1);
        qx[u] = xc[qq * 3 + 0]; qy[u] = xc[qq * 3 + 1]; qz[u] = xc[qq * 3 + 2];
    }
    static_assert(kKnnQpw % 2 == 0, "two queries per packed instruction");
    f32x2 nqx[kKnnQpw / 2], nqy[kKnnQpw / 2], nqz[kKnnQpw / 2];
#pragma unroll
    for (int h = 0; h < kKnnQpw / 2; ++h) {
        nqx[h] = f32x2{-qx[2 * h], -qx[2 * h + 1]}; nqy[h] = f32x2{-qy[2 * h], -qy[2 * h + 1]}; nqz[h] = f32x2{-qz[2 * h], -qz[2 * h + 1]};
    }
    // the squared distances of one candidate to the wave's queries (queries 2h, 2h + 1 packed)
    auto dist4 = [&](float px, float py, float pz, float (&v)[kKnnQpw]) {
        const f32x2 p2x = {px, px}, p2y = {py, py}, p2z = {pz, pz};
#pragma unroll
        for (int h = 0; h < kKnnQpw / 2; ++h) {
            const f32x2 a = dist2_pk(p2x, p2y, p2z, nqx[h], nqy[h], nqz[h]);
            v[2 * h] = a[0]; v[2 * h + 1] = a[1];
        }
    };
    auto dist2 = [&](int u, float px, float py, float pz) {   // scalar form of the same contract (pass 2, slow path)
        return dist2_scalar(px, py, pz, qx[u], qy[u], qz[u]);
    };
    // a thread stages whole points: candidate j = 256 q + threadIdx.x of the tile has lane id j & 63 = this thread's lane and
    // s = j >> 6 = 4 q + wave -- no divisions, and all 24 loads of a thread are in flight before the first LDS write (the
    // element-wise loop this replaces exposed the L2 latency several times per tile: 30 us of a 93 us launch at two 8 192-point clouds)
    auto stage = [&](int t0, int cnt) {
        __syncthreads();
        const float* src = xc + (size_t)t0 * 3;
        float px[8], py[8], pz[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int j = min(256 * q + (int)threadIdx.x, cnt - 1);
            px[q] = src[3 * j]; py[q] = src[3 * j + 1]; pz[q] = src[3 * j + 2];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (256 * q + (int)threadIdx.x < cnt) {
                float* dst = tile + 3 * knn_tile_point(lane, 4 * q + wv);
                dst[0] = px[q]; dst[1] = py[q]; dst[2] = pz[q];
            }
        }
        __syncthreads();
    };
    // ---- pass 1: per-lane minima, kept per 16-candidate chunk
    float cmin[kTiles][2][kKnnQpw];
#pragma unroll
    for (int t = 0; t < kTiles; ++t)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int u = 0; u < kKnnQpw; ++u) cmin[t][hf][u] = INF;
#pragma unroll
    for (int t = 0; t < kTiles; ++t) {
        const int t0 = t * kKnnTile;
        if (t0 < n) {                                       // uniform over the grid's workgroups of a launch
            const int cnt = min(kKnnTile, n - t0);
            stage(t0, cnt);
            const bool whole = cnt == kKnnTile;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
#pragma unroll 4
                for (int s = 16 * hf; s < 16 * hf + 16; ++s) {
                    if (whole || 64 * s + lane < cnt) {
                        float v[kKnnQpw];
                        const float* pt = tile + 3 * knn_tile_point(lane, s);
                        dist4(pt[0], pt[1], pt[2], v);
#pragma unroll
                        for (int u = 0; u < kKnnQpw; ++u) {
                            if (!kLoop && t0 + 64 * s + lane == qi[u]) v[u] = INF;
                            cmin[t][hf][u] = fminf(cmin[t][hf][u], v[u]);
                        }
                    }
                }
            }
        }
    }
    float tau[kKnnQpw];
#pragma unroll
    for (int u = 0; u < kKnnQpw; ++u) {
        float lmin = INF;
#pragma unroll
        for (int t = 0; t < kTiles; ++t) lmin = fminf(lmin, fminf(cmin[t][0][u], cmin[t][1][u]));
        tau[u] = __uint_as_float(wave_kth_smallest_bits(__float_as_uint(lmin), k));
    }
    // ---- pass 2: the candidates <= tau of the live chunks, compacted (the first kCap of them)
    int base[kKnnQpw];
#pragma unroll
    for (int u = 0; u < kKnnQpw; ++u) base[u] = 0;
#pragma unroll
    for (int t = 0; t < kTiles; ++t) {
        const int t0 = t * kKnnTile;
        if (t0 < n) {
            const int cnt = min(kKnnTile, n - t0);
            stage(t0, cnt);
#pragma unroll
            for (int u = 0; u < kKnnQpw; ++u) {
                // chunk id = 64 * half + lane.  The live chunks of the query in this tile, in rank order, WITHOUT scalar bit loops (popping
                // four set bits per step from the two ballot masks cost ~50 scalar instructions a step, and the CU's one scalar unit
                // was the kernel's bottleneck: 35 us of pass 2 at two 8 192-point clouds): every lane sends its id to the lane of its
                // rank -- live lanes to ranks 0 .. n0 - 1, the others behind them, a bijection, so one ds_permute per half and no LDS
                // memory -- and a step's four chunk ids are one ds_bpermute pull.
                if (qi[u] >= n) continue;                  // wave-uniform: a query past the cloud's end has no output
                const bool p0 = cmin[t][0][u] <= tau[u], p1 = cmin[t][1][u] <= tau[u];
                const u64 m0 = __ballot(p0), m1 = __ballot(p1);
                const int n0 = __popcll(m0), tot = n0 + __popcll(m1);
                if (tot == 0) continue;
                const int k0 = rank_in_mask(m0), k1 = rank_in_mask(m1);
                const int ids0 = __builtin_amdgcn_ds_permute(4 * (p0 ? k0 : n0 + (lane - k0)), lane);                 // ids0[r] = r-th live lane of half 0
                const int ids1 = __builtin_amdgcn_ds_permute(4 * (p1 ? k1 : (tot - n0) + (lane - k1)), 64 + lane);     // ... of half 1 (as chunk ids)
                // rank r of the query's list: half 0's n0 chunks, then half 1's
                const int lo = __builtin_amdgcn_ds_bpermute(4 * ((lane - n0) & 63), ids1);
                const int ids_a = lane < n0 ? ids0 : lo;                                                                // ranks 0 .. 63
                const int ids_b = __builtin_amdgcn_ds_bpermute(4 * ((64 + lane - n0) & 63), ids1);                      // ranks 64 .. 127 (tot > 64 only: n0 > 0 there)
                for (int r0 = 0; r0 < tot; r0 += 4) {
                    const int rr = r0 + (lane >> 4);
                    const int from_a = __builtin_amdgcn_ds_bpermute(4 * (rr & 63), ids_a);
                    int my = from_a;
                    if (tot > 64) {                        // wave-uniform, practically never
                        const int from_b = __builtin_amdgcn_ds_bpermute(4 * (rr & 63), ids_b);
                        my = rr < 64 ? from_a : from_b;
                    }
                    my = rr < tot ? my : -1;
                    const int l = my & 63, sft = ((my >> 6) << 4) + (lane & 15);
                    const int j = 64 * sft + l;
                    const bool in = my >= 0 && j < cnt;
                    const float* pt = tile + 3 * (in ? knn_tile_point(l, sft) : 0);
                    float vu = dist2(u, pt[0], pt[1], pt[2]);
                    const int idx = t0 + j;
                    vu = in ? vu : INF;
                    if (!kLoop) vu = idx == qi[u] ? INF : vu;
                    const bool pred = in && vu <= tau[u];
                    const u64 mask = __ballot(pred);
                    if (mask) {
                        const int at = base[u] + rank_in_mask(mask);
                        if (pred && at < kCap) list[wv * kKnnQpw + u][at] = pack_key(vu, idx);
                        base[u] += __popcll(mask);
                    }
                }
            }
        }
    }
    __syncthreads();                                       // LDS list writes before the reads below (and the tile is free)
#pragma unroll
    for (int u = 0; u < kKnnQpw; ++u) {
        if (qi[u] >= n) continue;                          // wave-uniform
        const int cnt = base[u];
        int out_idx = 0, slot = lane, out_idx2 = 0, slot2 = 128;
        if ((cnt <= 64) && (cnt >= k) && !force_slow) {
            u64 key;
            slot = rank_by_counting(list[wv * kKnnQpw + u], cnt, lane, key);
            out_idx = (int)(unsigned)(key & 0xffffffffu);
        } else if (kCap > 64 && (cnt <= kCap) && (cnt >= k) && !force_slow) {
            u64 k0, k1;
            rank_by_counting2(list[wv * kKnnQpw + u], cnt, lane, k0, k1, slot, slot2);
            out_idx = (int)(unsigned)(k0 & 0xffffffffu);
            out_idx2 = (int)(unsigned)(k1 & 0xffffffffu);
        } else {
            // slow path: k rounds of "smallest key strictly greater than the previous one", distances recomputed from global
            u64 last = 0;
            bool have_last = false;
            for (int t = 0; t < k; ++t) {
                u64 best = ~0ull;
                for (int idx = lane; idx < n; idx += 64) {
                    float v = dist2(u, xc[idx * 3 + 0], xc[idx * 3 + 1], xc[idx * 3 + 2]);
                    if (!kLoop && idx == qi[u]) v = INF;
                    const u64 key = pack_key(v, idx);
                    const bool ok = !have_last || key > last;
                    best = (ok && key < best) ? key : best;
                }
                best = wave_min_u64(best);
                if (lane == t) out_idx = (int)(unsigned)(best & 0xffffffffu);
                last = best;
                have_last = true;
            }
        }
        auto emit = [&](int slot_, int idx_) {
            if (slot_ >= k) return;
            const int q = qi[u];
            const size_t e = (size_t)q * k + slot_;
            const size_t epc = (size_t)n * k;
            if (edges) {
                edges[(size_t)cloud * 2 * epc + e] = idx_;
                edges[(size_t)cloud * 2 * epc + epc + e] = q;
            }
            if (row32) row32[(size_t)cloud * epc + e] = cloud * n + idx_;
            if (col32) col32[(size_t)cloud * epc + e] = cloud * n + q;
            if (row_hist) {
                const int slot_in_row = atomicAdd(&row_hist[cloud * n + idx_], 1);
                if (arrival) arrival[(size_t)cloud * epc + e] = slot_in_row;
            }
        };
        emit(slot, out_idx);
        emit(slot2, out_idx2);
    }
}

// ---------------------------------------------------------------------------------------
// CSR build: stable counting sort of edge ids by key
// ---------------------------------------------------------------------------------------
__global__ void csr_count_kernel(const int* __restrict__ key, int n_edges, int* __restrict__ cnt) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n_edges) atomicAdd(&cnt[key[e]], 1);
}

// exclusive scan of cnt[0..n) -> ptr[0..n], cursor[0..n) = ptr[0..n) ; single block of 1024
__global__ void csr_fill_kernel(const int* __restrict__ key, int n_edges, int* __restrict__ cursor,
                                int* __restrict__ perm) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n_edges) perm[atomicAdd(&cursor[key[e]], 1)] = e;
}
// the same with every edge's slot inside its segment already known (equigspr_knn_graph's `arrival`): a scatter, no atomics
__global__ void csr_fill_ranked_kernel(const int* __restrict__ key, const int* __restrict__ arrival, int n_edges,
                                       const int* __restrict__ ptr, int* __restrict__ perm) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n_edges) perm[ptr[key[e]] + arrival[e]] = e;
}

// The ranked fill with the scan INSIDE (<= 8 192 nodes: one or two 2 048-point pairs): every workgroup recomputes the exclusive
// prefix of the row counts in LDS -- <= 32 KB from L2, all workgroups at once -- instead of waiting behind a single-workgroup
// scan launch (5.8 us + a kernel boundary for ten barrier-separated steps of one workgroup); workgroup 0 also writes ptr.
// The count array is left as it is: the sort pass hands it back zeroed.
constexpr int kFusedScanMaxNodes = 8192;
__global__ __launch_bounds__(256) void csr_scan_fill_ranked_kernel(const int* __restrict__ cnt, int n_nodes, const int* __restrict__ key,
                                                                   const int* __restrict__ arrival, int n_edges, int* __restrict__ ptr,
                                                                   int* __restrict__ perm) {
    __shared__ int pre[kFusedScanMaxNodes + 1];
    __shared__ int wsum[4];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int per = (n_nodes + 255) >> 8;                 // <= 32 consecutive counts per thread
    const int lo = t * per;
    int v[32];
    int s = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        v[i] = (i < per && lo + i < n_nodes) ? cnt[lo + i] : 0;
        s += v[i];
    }
    int inc = s;                                          // inclusive scan over the wave, then over the four waves
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(inc, o, 64);
        if (lane >= o) inc += u;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int run = inc - s;
    for (int w = 0; w < wv; ++w) run += wsum[w];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        if (i < per && lo + i < n_nodes) pre[lo + i] = run;
        run += v[i];
    }
    if (t == 255) pre[n_nodes] = run;
    __syncthreads();
    if (blockIdx.x == 0)
        for (int i = t; i <= n_nodes; i += 256) ptr[i] = pre[i];
    const int e = blockIdx.x * 256 + t;
    if (e < n_edges) perm[pre[key[e]] + arrival[e]] = e;
}

// each node's segment back into ascending edge order (atomics filled it in arbitrary order):
// one wave per node, <= 64 entries sorted with a wave bitonic network, longer segments serially.  The same wave writes the
// inverse permutation of its segment (inv[perm[p]] = p, optional) and hands the node's cursor word back as zero, so the
// scratch array leaves build_csr the way it must enter it.
constexpr int kCsrLongSeg = 1024;
__global__ __launch_bounds__(256) void csr_sort_kernel(const int* __restrict__ ptr, int n_nodes, int* __restrict__ perm,
                                                       int* __restrict__ inv, int* __restrict__ scratch) {
    __shared__ __attribute__((aligned(16))) int long_seg[4][kCsrLongSeg];
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= n_nodes) return;                     // wave-uniform
    const int lo = ptr[j], hi = ptr[j + 1];
    const int len = hi - lo;
    if (lane == 0) scratch[j] = 0;
    if (len <= 64) {
        // distinct edge ids: a lane's final position is the number of smaller ids in the segment -- `len` broadcast steps of one
        // v_readlane + compare + add (a k = 20 neighbourhood: ~60 instructions) instead of the 21-stage 64-lane bitonic network
        const int key = lane < len ? perm[lo + lane] : 0x7fffffff;
        int rank = 0;
        for (int t = 0; t < len; ++t) rank += (__builtin_amdgcn_readlane(key, t) < key) ? 1 : 0;      // t is wave-uniform
        if (lane < len) {
            if (len > 1) perm[lo + rank] = key;
            if (inv) inv[key] = lo + rank;
        }
    } else if (len <= kCsrLongSeg) {
        // a hub (a point in more than 64 neighbour lists: one or two per few clouds of uniform synthetic points, in-degree up to several
        // hundred): the segment goes to LDS once and every lane ranks its entries by counting against broadcast reads, four keys per
        // ds_read_b128 -- len^2 / 64 compares per lane.  (The first version insertion-sorted such a segment on ONE lane in global
        // memory: 0.6 ms for an in-degree of ~150, 4 ms for ~450 -- found when the bench moved from one fixed pair to distinct pairs.)
        int* sg = long_seg[threadIdx.x >> 6];
        for (int a = lane; a < len; a += 64) sg[a] = perm[lo + a];
        for (int a = len + lane; a < ((len + 3) & ~3); a += 64) sg[a] = 0x7fffffff;      // pad to whole int4 reads
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int4* sg4 = reinterpret_cast<const int4*>(sg);
        const int n4 = (len + 3) >> 2;
        for (int a = lane; a < len; a += 64) {
            const int key = sg[a];
            int rank = 0;
            for (int t = 0; t < n4; ++t) {
                const int4 v = sg4[t];
                rank += (v.x < key ? 1 : 0) + (v.y < key ? 1 : 0) + (v.z < key ? 1 : 0) + (v.w < key ? 1 : 0);
            }
            perm[lo + rank] = key;
            if (inv) inv[key] = lo + rank;
        }
    } else {
        if (lane == 0) {                                  // beyond the LDS budget (in-degree > 1024): serial, as a last resort
            for (int a = lo + 1; a < hi; ++a) {
                const int v = perm[a];
                int b = a - 1;
                while (b >= lo && perm[b] > v) { perm[b + 1] = perm[b]; --b; }
                perm[b + 1] = v;
            }
        }
        if (inv) {
            __threadfence_block();
            __builtin_amdgcn_wave_barrier();
            for (int a = lo + lane; a < hi; a += 64) inv[perm[a]] = a;
        }
    }
}

__global__ void edges_i64_to_i32_kernel(const long long* __restrict__ edges, int n_edges, int off,
                                        int* __restrict__ row32, int* __restrict__ col32) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n_edges) {
        row32[e] = (int)edges[e] + off;
        col32[e] = (int)edges[(size_t)n_edges + e] + off;
    }
}

// ---------------------------------------------------------------------------------------
// General form (k > 64 or n > 8 192: the sizes torch_cluster.knn_graph accepts and the kernels above do not): a wave per query,
// the cloud streamed from global memory (L2) once per output slot -- round t extracts the smallest (d2, index) key strictly greater
// than round t - 1's, so the output is the same (d2, index)-lexicographic order, bit for bit, for any k <= n and any n.  O(k n)
// distance evaluations per query: a compatibility path (k = 100 over 2 048 points: ~0.2 G evaluations per cloud), not a tuned one.
// ---------------------------------------------------------------------------------------
template <bool kLoop>
__global__ __launch_bounds__(64) void knn_stream_kernel(const float* __restrict__ x, int n_clouds, int n, int k,
                                                        long long* __restrict__ edges, int* __restrict__ row32, int* __restrict__ col32,
                                                        int* __restrict__ row_hist, int* __restrict__ arrival) {
    const int lane = threadIdx.x;
    const long long qg = blockIdx.x;
    const int cloud = (int)(qg / n);
    const int q = (int)(qg - (long long)cloud * n);
    const float* __restrict__ xc = x + (size_t)cloud * n * 3;
    const float qx = xc[q * 3 + 0], qy = xc[q * 3 + 1], qz = xc[q * 3 + 2];
    const size_t epc = (size_t)n * k;
    u64 last = 0;
    bool have_last = false;
    for (int t = 0; t < k; ++t) {
        u64 best = ~0ull;
        for (int j = lane; j < n; j += 64) {
            if (!kLoop && j == q) continue;
            const u64 key = pack_key(dist2_scalar(xc[j * 3 + 0], xc[j * 3 + 1], xc[j * 3 + 2], qx, qy, qz), j);
            const bool ok = !have_last || key > last;
            best = (ok && key < best) ? key : best;
        }
        best = wave_min_u64(best);
        last = best;
        have_last = true;
        if (lane == 0) {
            const int idx = (int)(unsigned)(best & 0xffffffffu);
            const size_t e = (size_t)q * k + t;
            if (edges) {
                edges[(size_t)cloud * 2 * epc + e] = idx;
                edges[(size_t)cloud * 2 * epc + epc + e] = q;
            }
            if (row32) row32[(size_t)cloud * epc + e] = cloud * n + idx;
            if (col32) col32[(size_t)cloud * epc + e] = cloud * n + q;
            if (row_hist) {
                const int slot_in_row = atomicAdd(&row_hist[cloud * n + idx], 1);
                if (arrival) arrival[(size_t)cloud * epc + e] = slot_in_row;
            }
        }
    }
}

}  // namespace eq

// Largest cloud the wave-per-query kernel takes.  Alone the two forms tie at 2 048 points (16 us); since round 6 the graph build runs on a
// second stream BESIDE the previous step's kernels (train.GraphedStep build_ahead), and there the tiled form -- one sixteenth of the L2
// reads: every workgroup stages the cloud once for its 16 queries -- disturbs them less although it takes longer itself (35 vs 22 us
// under contention): one pair training 0.3242 -> 0.3182 ms, inference pair 5 842-6 420 -> 6 188-6 826 pairs/s, 16 pairs +0.6 %
// (three alternating runs each, one box).  2 048: the round-5 choice.
#ifndef EQ_KNN_PER_QUERY_MAX
#define EQ_KNN_PER_QUERY_MAX 1024
#endif
extern "C" int equigspr_knn_graph(const float* x, int32_t n_clouds, int32_t n, int32_t k, int32_t loop,
                                  int64_t* edges_i64, int32_t* row32, int32_t* col32, int32_t* row_hist, int32_t* arrival,
                                  int32_t force_slow_path, void* stream) {
    using namespace eq;
    EQ_CHECK_ARG(x != nullptr, EQUIGSPR_EINVAL, "knn_graph: x is null");
    EQ_CHECK_ARG(n_clouds >= 1 && n >= 1, EQUIGSPR_EINVAL, "knn_graph: need n_clouds>=1, n>=1");
    EQ_CHECK_ARG(k >= 1, EQUIGSPR_EINVAL, "knn_graph: k=%d < 1", k);
    EQ_CHECK_ARG((long long)n_clouds * n < (1ll << 31), EQUIGSPR_EINVAL, "knn_graph: node count overflows int32");
    EQ_CHECK_ARG(k <= (loop ? n : n - 1), EQUIGSPR_EINVAL, "knn_graph: k=%d exceeds candidates (n=%d, loop=%d)", k, n, loop);
    EQ_CHECK_ARG((long long)n_clouds * n * k < (1ll << 31), EQUIGSPR_EINVAL, "knn_graph: edge count overflows int32");
    EQ_CHECK_ARG(!arrival || row_hist, EQUIGSPR_EINVAL, "knn_graph: arrival needs row_hist");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(n_clouds * n), block(64);
    long long* e64 = reinterpret_cast<long long*>(edges_i64);
    if (k > 64 || n > 8192) {      // beyond the tuned kernels' register / LDS plans: the streaming form (same order, any k <= n, any n)
        if (loop) hipLaunchKernelGGL(knn_stream_kernel<true>, grid, block, 0, s, x, n_clouds, n, k, e64, row32, col32, row_hist, arrival);
        else hipLaunchKernelGGL(knn_stream_kernel<false>, grid, block, 0, s, x, n_clouds, n, k, e64, row32, col32, row_hist, arrival);
        EQ_CHECK_LAUNCH("knn_stream_kernel");
        return 0;
    }
#define EQ_KNN_LAUNCH(NPL, LOOP, FULL) \
    hipLaunchKernelGGL((knn_kernel<NPL, LOOP, FULL>), grid, block, 0, s, x, n_clouds, n, k, e64, row32, col32, row_hist, arrival, force_slow_path)
#define EQ_KNN_PICK(NPL) \
    do { const bool full = n == 64 * NPL; \
         if (loop) { if (full) EQ_KNN_LAUNCH(NPL, true, true); else EQ_KNN_LAUNCH(NPL, true, false); } \
         else { if (full) EQ_KNN_LAUNCH(NPL, false, true); else EQ_KNN_LAUNCH(NPL, false, false); } } while (0)
    if (n <= 256) {
        EQ_KNN_PICK(4);
    } else if (n <= 1024) {
        EQ_KNN_PICK(16);
    } else if (n <= EQ_KNN_PER_QUERY_MAX) {      // measured: the per-query kernel is as fast as the tiled one up to 2 048 points
        EQ_KNN_PICK(32);
    } else {                     // 4 096 / 8 192 points (no 128-value register arrays)
        const bool wide = (long long)n_clouds * ceil_div(n, 32) >= 2048;
        const int bpc = ceil_div(n, wide ? 32 : 16);
#define EQ_KNN_TILED(LOOP, QPW, CAP) \
    do { if (n <= 2 * kKnnTile) hipLaunchKernelGGL((knn_tiled_kernel<LOOP, QPW, CAP, 2>), dim3(n_clouds * bpc), dim3(256), 0, s, x, n_clouds, n, k, e64, row32, col32, row_hist, arrival, force_slow_path); \
         else hipLaunchKernelGGL((knn_tiled_kernel<LOOP, QPW, CAP, 4>), dim3(n_clouds * bpc), dim3(256), 0, s, x, n_clouds, n, k, e64, row32, col32, row_hist, arrival, force_slow_path); } while (0)
#define EQ_KNN_TILED_K(LOOP, QPW) do { if (k <= 24) EQ_KNN_TILED(LOOP, QPW, 64); else EQ_KNN_TILED(LOOP, QPW, 128); } while (0)
        if (loop) { if (wide) EQ_KNN_TILED_K(true, 8); else EQ_KNN_TILED_K(true, 4); }
        else { if (wide) EQ_KNN_TILED_K(false, 8); else EQ_KNN_TILED_K(false, 4); }
#undef EQ_KNN_TILED_K
#undef EQ_KNN_TILED
    }
#undef EQ_KNN_PICK
#undef EQ_KNN_LAUNCH
    EQ_CHECK_LAUNCH("knn_kernel");
    return 0;
}

extern "C" int equigspr_build_csr(const int32_t* key, int32_t n_edges, int32_t n_nodes, int32_t* ptr,
                                  int32_t* perm, int32_t* inv, int32_t* scratch, int32_t scratch_state, const int32_t* arrival,
                                  void* stream) {
    using namespace eq;
    EQ_CHECK_ARG(ptr && scratch && (n_edges == 0 || (key && perm)), EQUIGSPR_EINVAL, "build_csr: null pointer");
    EQ_CHECK_ARG(n_edges >= 0 && n_nodes >= 1, EQUIGSPR_EINVAL, "build_csr: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    // (a single-block LDS count+scan was tried instead of clear + count + scan: 25 us vs ~12 us -- one block cannot hide
    // the key-load latency.)  The clear is skipped for a scratch the caller knows to be zero: the sort pass re-zeroes every
    // word the count/fill passes touched, so a buffer can be handed from call to call (hipMemsetAsync of N+1 words is
    // two fill launches -- an aligned body and a tail -- i.e. ~10 us of a 0.5 ms step)
    EQ_CHECK_ARG(scratch_state >= 0 && scratch_state <= 2, EQUIGSPR_EINVAL, "build_csr: scratch_state must be 0, 1 or 2");
    EQ_CHECK_ARG(!arrival || scratch_state == EQUIGSPR_SCRATCH_HISTOGRAM, EQUIGSPR_EINVAL, "build_csr: arrival goes with EQUIGSPR_SCRATCH_HISTOGRAM");
    if (scratch_state == EQUIGSPR_SCRATCH_DIRTY) {
        hipError_t err = hipMemsetAsync(scratch, 0, sizeof(int) * (size_t)(n_nodes + 1), s);
        if (err != hipSuccess) { set_error("build_csr memset: %s", hipGetErrorString(err)); return (int)err; }
    }
    if (n_edges > 0 && scratch_state != EQUIGSPR_SCRATCH_HISTOGRAM)      // (equigspr_knn_graph can leave the histogram behind)
        hipLaunchKernelGGL(csr_count_kernel, dim3(ceil_div(n_edges, 256)), dim3(256), 0, s, key, n_edges, scratch);
    if (arrival && n_edges > 0 && n_nodes <= kFusedScanMaxNodes) {      // the k-NN path of one or two pairs: scan inside the fill
        hipLaunchKernelGGL(csr_scan_fill_ranked_kernel, dim3(ceil_div(n_edges, 256)), dim3(256), 0, s, scratch, n_nodes, key, arrival, n_edges, ptr, perm);
        hipLaunchKernelGGL(csr_sort_kernel, dim3(ceil_div(n_nodes, 4)), dim3(256), 0, s, ptr, n_nodes, perm, inv, scratch);
        EQ_CHECK_LAUNCH("build_csr");
        return 0;
    }
    // large N: two-level scan; its chunk totals borrow the head of `perm`, which is only written by the fill below
    const int chunks = ceil_div(n_nodes, kScanChunk);
    launch_exclusive_scan(scratch, n_nodes, ptr, (n_edges >= chunks && chunks <= 1024) ? perm : nullptr, s);
    if (n_edges > 0) {
        if (arrival)
            hipLaunchKernelGGL(csr_fill_ranked_kernel, dim3(ceil_div(n_edges, 256)), dim3(256), 0, s, key, arrival, n_edges, ptr, perm);
        else
            hipLaunchKernelGGL(csr_fill_kernel, dim3(ceil_div(n_edges, 256)), dim3(256), 0, s, key, n_edges, scratch, perm);
        hipLaunchKernelGGL(csr_sort_kernel, dim3(ceil_div(n_nodes, 4)), dim3(256), 0, s, ptr, n_nodes, perm, inv, scratch);
    }
    EQ_CHECK_LAUNCH("build_csr");
    return 0;
}

namespace eq {
__global__ void invert_perm_kernel(const int* __restrict__ perm, int n, int* __restrict__ inv) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) inv[perm[p]] = p;
}
}  // namespace eq

extern "C" int equigspr_invert_perm(const int32_t* perm, int32_t n, int32_t* inv, void* stream) {
    using namespace eq;
    if (n == 0) return 0;
    EQ_CHECK_ARG(perm && inv && n > 0, EQUIGSPR_EINVAL, "invert_perm: bad arguments");
    hipLaunchKernelGGL(invert_perm_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, perm, n, inv);
    EQ_CHECK_LAUNCH("invert_perm");
    return 0;
}

extern "C" int equigspr_edges_i64_to_i32(const int64_t* edges_i64, int32_t n_edges, int32_t node_offset,
                                         int32_t* row32, int32_t* col32, void* stream) {
    using namespace eq;
    if (n_edges == 0) return 0;                      // an empty edge list has no storage to point at
    EQ_CHECK_ARG(edges_i64 && row32 && col32 && n_edges > 0, EQUIGSPR_EINVAL, "edges_i64_to_i32: bad args");
    hipLaunchKernelGGL(edges_i64_to_i32_kernel, dim3(ceil_div(n_edges, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const long long*>(edges_i64), n_edges, node_offset, row32, col32);
    EQ_CHECK_LAUNCH("edges_i64_to_i32");
    return 0;
}
