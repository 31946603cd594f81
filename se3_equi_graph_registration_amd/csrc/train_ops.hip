// Small fused training-step kernels: the pose loss of the reference training loop and a flat-bucket Adam.
//   pose loss : pose_loss() of src/3dmatch_train_egnn_with_batch.py:896-962 (geodesic rotation angle :952,
//               translation-direction angle :955-960), forward + backward, one thread per pair;
//   Adam      : torch.optim.Adam semantics (lr, betas, eps; no weight decay / amsgrad -- train :1412) over up
//               to four flat fp32 buckets in ONE launch; the step counter lives on the device so the whole
//               optimizer step is hipGraph-capturable.
#include <cstdlib>
#include "common.h"
#include "pose_loss_device.h"

namespace eq {

__global__ void pose_loss_fwd_kernel(const float* __restrict__ R, const float* __restrict__ t, const float* __restrict__ gt_pose,
                                     int B, float* __restrict__ rot_loss, float* __restrict__ trans_loss) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    pose_loss_pair(R + (size_t)b * 9, t + (size_t)b * 3, gt_pose + (size_t)b * 16, rot_loss[b], trans_loss[b]);
}

__global__ void pose_loss_bwd_kernel(const float* __restrict__ R, const float* __restrict__ t, const float* __restrict__ gt_pose,
                                     const float* __restrict__ g_rot, const float* __restrict__ g_trans, int B,
                                     float* __restrict__ gR, float* __restrict__ gt) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    pose_loss_grad_pair(R + (size_t)b * 9, t + (size_t)b * 3, gt_pose + (size_t)b * 16, g_rot ? g_rot[b] : 0.0f, g_trans ? g_trans[b] : 0.0f,
                        gR + (size_t)b * 9, gt + (size_t)b * 3);
}

// Step loss of the training loop (train :1120): loss = corr_loss + mean_b(rot_loss) + mean_b(trans_loss) in ONE launch
// (the three means, two adds and their autograd nodes are ~10 ATen launches otherwise).  One block; fixed-order sums.
__global__ __launch_bounds__(256) void step_loss_fwd_kernel(const float* __restrict__ R, const float* __restrict__ t,
                                                            const float* __restrict__ gt_pose, const float* __restrict__ closs, int B,
                                                            float* __restrict__ loss, float* __restrict__ rot_loss,
                                                            float* __restrict__ trans_loss) {
    for (int b = threadIdx.x; b < B; b += 256)
        pose_loss_pair(R + (size_t)b * 9, t + (size_t)b * 3, gt_pose + (size_t)b * 16, rot_loss[b], trans_loss[b]);
    __syncthreads();
    step_loss_sum(rot_loss, trans_loss, closs ? closs[0] : 0.0f, B, loss);
}

// d loss / d(R, t) = pose-loss gradients scaled by g/B (g = upstream scalar gradient, 1 when NULL)
__global__ void step_loss_bwd_kernel(const float* __restrict__ R, const float* __restrict__ t, const float* __restrict__ gt_pose,
                                     const float* __restrict__ g, int B, float* __restrict__ gR, float* __restrict__ gt) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float up = (g ? g[0] : 1.0f) / (float)B;
    pose_loss_grad_pair(R + (size_t)b * 9, t + (size_t)b * 3, gt_pose + (size_t)b * 16, up, up, gR + (size_t)b * 9, gt + (size_t)b * 3);
}

struct AdamBuckets {
    float* p[4]; float* g[4]; float* m[4]; float* v[4]; long long n[4]; int count;
};

template <bool kClear>
__global__ __launch_bounds__(256) void adam_kernel(AdamBuckets bk, float* __restrict__ step, float lr, float b1, float b2,
                                                   float eps) {
    // every thread reads the SAME pre-increment step value; the LAST workgroup to finish (ticket counter in the second word,
    // zero between launches) publishes step+1 for the next launch, so no workgroup can observe the new value
    const float tcur = step[0] + 1.0f;
    const float bc1 = 1.0f - powf(b1, tcur), bc2s = sqrtf(1.0f - powf(b2, tcur));
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x, stride = (long long)gridDim.x * 256;
    for (int k = 0; k < bk.count; ++k) {
        for (long long i = gid; i < bk.n[k]; i += stride) {
            const float g = bk.g[k][i];
            const float m = b1 * bk.m[k][i] + (1.0f - b1) * g;
            const float v = b2 * bk.v[k][i] + (1.0f - b2) * g * g;
            bk.m[k][i] = m;
            bk.v[k][i] = v;
            bk.p[k][i] -= (lr / bc1) * m / (sqrtf(v) / bc2s + eps);
            if (kClear) bk.g[k][i] = 0.0f;       // the optimizer step consumes the gradient: no separate zero_grad fill
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned* done = reinterpret_cast<unsigned*>(step + 1);
        __threadfence();
        if (atomicAdd(done, 1u) == gridDim.x - 1) { step[0] = tcur; *done = 0u; }
    }
}


// The guarded step of the graphed training loop (train.GraphedStep): the reference loop's host decision "loss and every gradient
// finite, else skip / stop" (train :1124 set_detect_anomaly; run_training_batches) evaluated on the device, so that a captured step
// needs no host round trip.  Every workgroup scans ALL gradient buckets (+ `extra`, + the loss): all reach the same verdict without
// any inter-workgroup wait.  Verdict finite: the plain Adam update, gradients cleared as consumed.  Verdict non-finite: NOTHING is
// modified until every workgroup has drawn its ticket (a cleared entry could otherwise change a slower workgroup's verdict); the
// last one then clears all gradients and `extra`.  The last workgroup also keeps the loop's running totals in `stats`:
//   [0] sum of loss * hyper[4] over the applied steps   [1] applied steps   [2] skipped steps
//   [3] 1 + index (count of guarded calls before it) of the FIRST skipped step, 0 = none   [4] the last loss   [5] guarded calls
//   [6] void steps (extra[1] not finite: some rank had no data; not counted as skipped)
// progress_host (pinned host memory, may be NULL) receives stats[5] after every call.
// hyper (device): lr, beta1, beta2, eps, loss weight -- read at run time, so a scheduler's new lr reaches a captured graph.
struct AdamGuard {
    const float* hyper; const float* loss; float* stats; float* extra; int n_extra; float* progress_host;
};

#ifndef EQ_ADAM_ROTATE
#define EQ_ADAM_ROTATE 1
#endif
constexpr int kGuardThreads = 1024;
__global__ __launch_bounds__(kGuardThreads) void adam_guarded_kernel(AdamBuckets bk, float* __restrict__ step, AdamGuard gd) {
    constexpr int NT = kGuardThreads;
    __shared__ int s_bad;
    // the running totals, requested now by the thread that may turn out to be the launch's last: they were written by the previous
    // launch, and fetching them only after the ticket cost the tail of the kernel half a dozen serial memory round trips
    float4 sA = make_float4(0.0f, 0.0f, 0.0f, 0.0f), sB = sA;
    float e1 = 0.0f;
    if (threadIdx.x == 0) {
        s_bad = 0;
        sA = reinterpret_cast<const float4*>(gd.stats)[0];
        sB = reinterpret_cast<const float4*>(gd.stats)[1];
        if (gd.n_extra > 1) e1 = gd.extra[1];
    }
    // everything the update needs is requested before the scan, so that its latency hides under the scan's: the hyper-parameters,
    // the step count, and the operands of the thread's first element of every bucket (all of them, for models up to blocks x 1024
    // parameters per bucket)
    const float lr = gd.hyper[0], b1 = gd.hyper[1], b2 = gd.hyper[2], eps = gd.hyper[3];
    const float tcur = step[0] + 1.0f;
    const long long gid = (long long)blockIdx.x * NT + threadIdx.x, stride = (long long)gridDim.x * NT;
    float pg[4], pm[4], pv[4], pp[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool on = k < bk.count && gid < bk.n[k];
        pg[k] = on ? bk.g[k][gid] : 0.0f; pm[k] = on ? bk.m[k][gid] : 0.0f; pv[k] = on ? bk.v[k][gid] : 0.0f; pp[k] = on ? bk.p[k][gid] : 0.0f;
    }
    __syncthreads();
    // largest |bit pattern| seen: >= 0x7f800000 <=> an inf or a NaN (one AND + one unsigned MAX per element).  (The first version
    // waited for every load before issuing the next -- 55 dependent round trips, 13 us for the 56 k gradients of the 4-layer model.)
    unsigned top = 0u;
    auto see = [&](float v) { const unsigned u = __float_as_uint(v) & 0x7fffffffu; top = u > top ? u : top; };
    for (int k = 0; k < bk.count; ++k) {
        // sixteen 16-byte requests in flight per thread: each bucket of the 4-layer model (40 k + 16 k gradients) is one round trip
        const long long n4 = bk.n[k] >> 2;
        const float4* g4 = reinterpret_cast<const float4*>(bk.g[k]);          // buckets are 16-byte aligned (checked on the host)
        // every workgroup reads the whole bucket (the maximum does not care about the order): each starts at its own offset, so that the
        // workgroups do not all pull the same L2 lines at the same moment (the typed network's 249 k gradients: 34.4 -> 27.8 us)
        const long long rot = EQ_ADAM_ROTATE ? (n4 / gridDim.x) * blockIdx.x : 0;
        for (long long base = threadIdx.x; base < n4; base += 16 * NT) {
            float4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const long long i = base + (long long)u * NT;
                long long j = i + rot;
                j = j >= n4 ? j - n4 : j;
                v[u] = i < n4 ? g4[j] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) { see(v[u].x); see(v[u].y); see(v[u].z); see(v[u].w); }
        }
        for (long long i = (n4 << 2) + threadIdx.x; i < bk.n[k]; i += NT) see(bk.g[k][i]);
    }
    for (int i = threadIdx.x; i < gd.n_extra; i += NT) see(gd.extra[i]);
    const float loss = gd.loss ? gd.loss[0] : 0.0f;
    see(loss);
    int bad = top >= 0x7f800000u;
    if (bad) s_bad = 1;                      // benign race: every writer stores 1
    __syncthreads();
    bad = s_bad;
    if (!bad) {
        const float bc1 = 1.0f - powf(b1, tcur), bc2s = sqrtf(1.0f - powf(b2, tcur));
        auto update = [&](int k, long long i, float g, float m0, float v0, float p0) {
            const float m = b1 * m0 + (1.0f - b1) * g;
            const float v = b2 * v0 + (1.0f - b2) * g * g;
            bk.m[k][i] = m;
            bk.v[k][i] = v;
            bk.p[k][i] = p0 - (lr / bc1) * m / (sqrtf(v) / bc2s + eps);
            bk.g[k][i] = 0.0f;
        };
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < bk.count) {
                if (gid < bk.n[k]) update(k, gid, pg[k], pm[k], pv[k], pp[k]);
                for (long long i = gid + stride; i < bk.n[k]; i += stride) update(k, i, bk.g[k][i], bk.m[k][i], bk.v[k][i], bk.p[k][i]);
            }
        }
    }
    __syncthreads();
    __shared__ int s_last;
    if (threadIdx.x == 0) {
        unsigned* done = reinterpret_cast<unsigned*>(step + 1);
        __threadfence();
        s_last = atomicAdd(done, 1u) == gridDim.x - 1;
        if (s_last) {
            *done = 0u;
            // extra[1] non-finite = "a rank had no data for this step" (train.GraphedStep): a void step, not a rejected one
            const bool voided = (__float_as_uint(e1) & 0x7fffffffu) >= 0x7f800000u;
            if (!bad) { step[0] = tcur; sA.x += loss * gd.hyper[4]; sA.y += 1.0f; }
            else if (voided) sB.z += 1.0f;
            else { sA.z += 1.0f; if (sA.w == 0.0f) sA.w = sB.y + 1.0f; }
            sB.x = loss;
            sB.y += 1.0f;
            reinterpret_cast<float4*>(gd.stats)[0] = sA;
            reinterpret_cast<float4*>(gd.stats)[1] = sB;
            // pinned HOST word: the launching thread sees "step n is done" without a HIP event.  A RELAXED system-scope store: the host
            // reads nothing but this word (it only needs to know that the step's kernels -- all earlier in the stream -- no longer read
            // their input slot), so no system-scope release, i.e. no write-back of the whole L2, is wanted here
            if (gd.progress_host) __hip_atomic_store(gd.progress_host, sB.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    __syncthreads();
    if (bad && s_last) {                     // every workgroup has finished reading: now the rejected gradients can go
        __threadfence();
        for (int k = 0; k < bk.count; ++k)
            for (long long i = threadIdx.x; i < bk.n[k]; i += NT) bk.g[k][i] = 0.0f;
        for (int i = threadIdx.x; i < gd.n_extra; i += NT) gd.extra[i] = 0.0f;
    }
}

}  // namespace eq

extern "C" int equigspr_pose_loss_forward(const float* R, const float* t, const float* gt_pose, int32_t B, float* rot_loss,
                                          float* trans_loss, void* stream) {
    using namespace eq;
    EQ_CHECK_ARG(R && t && gt_pose && rot_loss && trans_loss && B >= 1, EQUIGSPR_EINVAL, "pose_loss_forward: bad arguments");
    hipLaunchKernelGGL(pose_loss_fwd_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream, R, t, gt_pose, B, rot_loss, trans_loss);
    EQ_CHECK_LAUNCH("pose_loss_forward");
    return 0;
}

extern "C" int equigspr_pose_loss_backward(const float* R, const float* t, const float* gt_pose, const float* g_rot,
                                           const float* g_trans, int32_t B, float* grad_R, float* grad_t, void* stream) {
    using namespace eq;
    EQ_CHECK_ARG(R && t && gt_pose && grad_R && grad_t && B >= 1, EQUIGSPR_EINVAL, "pose_loss_backward: bad arguments");
    hipLaunchKernelGGL(pose_loss_bwd_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream, R, t, gt_pose, g_rot, g_trans, B,
                       grad_R, grad_t);
    EQ_CHECK_LAUNCH("pose_loss_backward");
    return 0;
}

extern "C" int equigspr_step_loss_forward(const float* R, const float* t, const float* gt_pose, const float* corr_loss, int32_t B,
                                          float* loss, float* rot_loss, float* trans_loss, void* stream) {
    using namespace eq;
    EQ_CHECK_ARG(R && t && gt_pose && loss && rot_loss && trans_loss && B >= 1, EQUIGSPR_EINVAL, "step_loss_forward: bad arguments");
    hipLaunchKernelGGL(step_loss_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, R, t, gt_pose, corr_loss, B, loss, rot_loss, trans_loss);
    EQ_CHECK_LAUNCH("step_loss_forward");
    return 0;
}

extern "C" int equigspr_step_loss_backward(const float* R, const float* t, const float* gt_pose, const float* grad_loss, int32_t B,
                                           float* grad_R, float* grad_t, void* stream) {
    using namespace eq;
    EQ_CHECK_ARG(R && t && gt_pose && grad_R && grad_t && B >= 1, EQUIGSPR_EINVAL, "step_loss_backward: bad arguments");
    hipLaunchKernelGGL(step_loss_bwd_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream, R, t, gt_pose, grad_loss, B, grad_R, grad_t);
    EQ_CHECK_LAUNCH("step_loss_backward");
    return 0;
}

extern "C" int equigspr_adam_step(int32_t n_buckets, float* const* params, float* const* grads, float* const* exp_avg,
                                  float* const* exp_avg_sq, const int64_t* sizes /* host arrays */, float* step_dev /* [2] */,
                                  float lr, float beta1, float beta2, float eps, int32_t clear_grads, void* stream) {
    using namespace eq;
    EQ_CHECK_ARG(n_buckets >= 1 && n_buckets <= 4 && params && grads && exp_avg && exp_avg_sq && sizes && step_dev,
                 EQUIGSPR_EINVAL, "adam_step: 1..4 buckets, non-null arrays");
    AdamBuckets bk;
    bk.count = n_buckets;
    long long tot = 0;
    for (int k = 0; k < n_buckets; ++k) {
        bk.p[k] = params[k]; bk.g[k] = grads[k]; bk.m[k] = exp_avg[k]; bk.v[k] = exp_avg_sq[k]; bk.n[k] = sizes[k];
        tot += sizes[k];
    }
    int blocks = ceil_div(tot, 256 * 4);
    blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
    hipStream_t s = (hipStream_t)stream;
    if (clear_grads) hipLaunchKernelGGL(adam_kernel<true>, dim3(blocks), dim3(256), 0, s, bk, step_dev, lr, beta1, beta2, eps);
    else hipLaunchKernelGGL(adam_kernel<false>, dim3(blocks), dim3(256), 0, s, bk, step_dev, lr, beta1, beta2, eps);
    EQ_CHECK_LAUNCH("adam_step");
    return 0;
}

extern "C" int equigspr_adam_step_guarded(int32_t n_buckets, float* const* params, float* const* grads, float* const* exp_avg,
                                          float* const* exp_avg_sq, const int64_t* sizes /* host arrays */, float* step_dev /* [2] */,
                                          const float* hyper_dev /* [5] */, const float* loss_dev /* [1] or NULL */,
                                          float* stats_dev /* [8] */, float* extra_dev, int32_t n_extra, float* progress_host,
                                          void* stream) {
    using namespace eq;
    EQ_CHECK_ARG(n_buckets >= 1 && n_buckets <= 4 && params && grads && exp_avg && exp_avg_sq && sizes && step_dev && hyper_dev && stats_dev,
                 EQUIGSPR_EINVAL, "adam_step_guarded: 1..4 buckets, non-null arrays, hyper and stats on the device");
    EQ_CHECK_ARG(((uintptr_t)stats_dev & 15) == 0, EQUIGSPR_EINVAL, "adam_step_guarded: stats_dev must be 16-byte aligned");
    EQ_CHECK_ARG(n_extra >= 0 && (n_extra == 0 || extra_dev), EQUIGSPR_EINVAL, "adam_step_guarded: extra_dev is NULL with n_extra > 0");
    AdamBuckets bk;
    bk.count = n_buckets;
    long long tot = 0;
    for (int k = 0; k < n_buckets; ++k) {
        EQ_CHECK_ARG(((uintptr_t)grads[k] & 15) == 0, EQUIGSPR_EINVAL, "adam_step_guarded: gradient buckets must be 16-byte aligned");
        bk.p[k] = params[k]; bk.g[k] = grads[k]; bk.m[k] = exp_avg[k]; bk.v[k] = exp_avg_sq[k]; bk.n[k] = sizes[k];
        tot += sizes[k];
    }
    // every workgroup scans all `tot` gradients: bound the redundant reads (blocks * tot <= 16 M floats, i.e. 64 MB from L2)
    long long blocks = ceil_div(tot, kGuardThreads * 2);
    const long long cap = (16ll << 20) / (tot > 0 ? tot : 1);
    if (blocks > cap) blocks = cap;
    blocks = blocks < 8 ? 8 : (blocks > 512 ? 512 : blocks);
#ifdef EQ_ADAM_BLOCKS_ENV
    if (const char* e = getenv("EQUIGSPR_ADAM_BLOCKS")) blocks = atoi(e);      // developer A/B
#endif
    AdamGuard gd{hyper_dev, loss_dev, stats_dev, extra_dev, n_extra, progress_host};
    hipLaunchKernelGGL(adam_guarded_kernel, dim3((int)blocks), dim3(kGuardThreads), 0, (hipStream_t)stream, bk, step_dev, gd);
    EQ_CHECK_LAUNCH("adam_step_guarded");
    return 0;
}

// ---------------------------------------------------------------------------------------
// N2: evaluation metrics on the device (tools/evaluation_metrics.py:14-43, eval loop eval :1245-1296)
//   out[b] = { rotation error [deg], translation error [cm], recall = sqrt(TP/n), precision = TP/n, F1 }
//   TP = #{ i : || R_pred src_i + t_pred - tgt_i || < tau }.  fp64 arithmetic, as the numpy reference.
// ---------------------------------------------------------------------------------------
namespace eq {
__global__ __launch_bounds__(256) void eval_metrics_kernel(const float* __restrict__ R, const float* __restrict__ t,
                                                           const float* __restrict__ gt_pose, const float* __restrict__ src,
                                                           const float* __restrict__ tgt, int n, double tau, double* __restrict__ out) {
    __shared__ int cnt[256];
    const int b = blockIdx.x;
    const float* Rb = R + 9 * b;
    const float* tb = t + 3 * b;
    const float* G = gt_pose + 16 * b;
    double r[9], tr[3];
    for (int i = 0; i < 9; ++i) r[i] = (double)Rb[i];
    for (int i = 0; i < 3; ++i) tr[i] = (double)tb[i];
    int c = 0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float* s = src + ((size_t)b * n + i) * 3;
        const float* g = tgt + ((size_t)b * n + i) * 3;
        double d2 = 0.0;
        for (int a = 0; a < 3; ++a) {
            const double v = r[3 * a] * (double)s[0] + r[3 * a + 1] * (double)s[1] + r[3 * a + 2] * (double)s[2] + tr[a] - (double)g[a];
            d2 += v * v;
        }
        c += sqrt(d2) < tau ? 1 : 0;
    }
    cnt[threadIdx.x] = c;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) cnt[threadIdx.x] += cnt[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double te = 0.0, trace = 0.0;
        for (int a = 0; a < 3; ++a) {
            const double d = (double)G[4 * a + 3] - tr[a];
            te += d * d;
            for (int k = 0; k < 3; ++k) trace += (double)G[4 * k + a] * r[3 * k + a];        // trace(G_R^T R)
        }
        double cosv = (trace - 1.0) * 0.5;
        cosv = cosv < -1.0 ? -1.0 : (cosv > 1.0 ? 1.0 : cosv);
        const double tp = (double)cnt[0];
        const double recall = sqrt(tp / (double)n), precision = n > 0 ? tp / (double)n : 0.0;
        double* o = out + 5 * b;
        o[0] = acos(cosv) * (180.0 / 3.14159265358979323846);
        o[1] = sqrt(te) * 100.0;
        o[2] = recall;
        o[3] = precision;
        o[4] = 2.0 * (precision * recall) / (precision + recall + 1e-6);                      // eval :1276
    }
}
}  // namespace eq

extern "C" int equigspr_eval_metrics(const float* R, const float* t, const float* gt_pose, const float* src, const float* tgt,
                                     int32_t n_pairs, int32_t n, double tau, double* out, void* stream) {
    using namespace eq;
    EQ_CHECK_ARG(R && t && gt_pose && src && tgt && out && n_pairs >= 1 && n >= 1, EQUIGSPR_EINVAL, "eval_metrics: bad arguments");
    hipLaunchKernelGGL(eval_metrics_kernel, dim3(n_pairs), dim3(256), 0, (hipStream_t)stream, R, t, gt_pose, src, tgt, n, tau, out);
    EQ_CHECK_LAUNCH("eval_metrics");
    return 0;
}
