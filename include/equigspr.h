/*
 * equigspr.h -- C-ABI of the MI355X-native Equi-GSPR registration hot path.
 *
 * libequigspr.so (built from the .hip files in se3_equi_graph_registration_amd/csrc for gfx950) exports
 * exactly the entry points below.  Plain pointers and sizes only: no torch types, no C++
 * types.  Every pointer is a DEVICE pointer unless it says "host".  All floating point is
 * fp32; all functions enqueue on `stream` (a hipStream_t passed as void*) and never
 * synchronise the host, never allocate (hipGraph-capturable: callers own every buffer and
 * size the workspaces with the *_size() helpers).
 *
 * The reference has no FFI: its hot path is Python calling ATen ops.  Each entry point
 * therefore cites the reference Python interface (file:line under /root/reference) that
 * it replaces; INTEGRATION.md shows the ctypes binding a reference maintainer would add.
 *
 * Return value: 0 on success; a positive hipError_t for a HIP failure; a negative
 * EQUIGSPR_E* for an argument error.  equigspr_last_error() gives the text (thread-local).
 */
#ifndef EQUIGSPR_H
#define EQUIGSPR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EQUIGSPR_EINVAL   (-1)  /* bad shape / size / null pointer                         */
#define EQUIGSPR_ENOTSUP  (-2)  /* configuration outside the compiled kernel set             */
#define EQUIGSPR_ABI_VERSION 14

/* EGNN architecture.  Mirrors EGNN.__init__ (src/3dmatch_train_egnn_with_batch.py:294-326)
 * plus num_heads (E_GCL.__init__ :186; the shipped checkpoints need 4).
 * Supported kernel set: hidden_nf in {16, 32, 33, 48, 64} (32 / 33: the 16-edge MFMA edge kernels of csrc/egnn_mfma16.h; 16 / 48 / 64:
 * the feature-parallel MFMA edge kernels of csrc/egnn_edge_fp.h), num_heads in {1, 2, 4, 8} (1 at hidden_nf 33: the other divisors
 * of 33 are not instantiated),
 * in_nf/out_nf <= 64, in_edge_nf in {0, 1}, 1 <= n_layers <= 16.
 * Arithmetic: fp32 storage and accumulation everywhere.  At hidden_nf 32 / 33 the two 32 x 32 products of the edge MLP (and their
 * transposes in the hidden_nf 32 backward) are evaluated as six bf16 x bf16 -> fp32 matrix products on 3-way split operands
 * (x = hi + mid + lo): within 1e-6 relative of the fp32 product, bitwise reproducible run to run, independent of the batch. */
typedef struct equigspr_egnn_config {
    int32_t in_nf;       /* in_node_nf  */
    int32_t hidden_nf;   /* hidden_nf   */
    int32_t out_nf;      /* out_node_nf */
    int32_t in_edge_nf;  /* in_edge_nf (edge_attr width) */
    int32_t n_layers;
    int32_t num_heads;
    int32_t edge_basis;  /* EQUIGSPR_BASIS_FRAMES: the reference's 12 geometry inputs (radial, dist, dot, frame vectors; train :128-181).
                            EQUIGSPR_BASIS_IRREPS: rotation invariants from the degree <= 1 irrep basis kernels
                            of tools/basis.py::get_basis contracted with the endpoint coordinates (csrc/egnn_math.h) */
} equigspr_egnn_config;
#define EQUIGSPR_BASIS_FRAMES 0
#define EQUIGSPR_BASIS_IRREPS 1

/* Graph in the device format every layer reuses (built once per batch of clouds).
 * row = message/aggregation index (= the NEIGHBOUR for torch_cluster.knn_graph output),
 * col = the other endpoint (the centre); both are global node ids over all clouds.
 * rptr/rperm: for node j the edges with row==j, ascending edge id (reproduces the CPU
 * scatter_add_ summation order of unsorted_segment_sum, :343-348).  cptr/cperm: same by col. */
typedef struct equigspr_graph {
    int32_t n_nodes;
    int32_t n_edges;
    const int32_t* row;    /* [E] */
    const int32_t* col;    /* [E] */
    const int32_t* rptr;   /* [N+1] */
    const int32_t* rperm;  /* [E] */
    const int32_t* cptr;   /* [N+1] */
    const int32_t* cperm;  /* [E] */
    const int32_t* rinv;   /* [E] inverse of rperm (equigspr_invert_perm): position of edge e in the by-row order.  The edge
                              kernels store per-edge rows AT that position, so the node kernels stream each node's segment
                              [rptr[j], rptr[j+1]) contiguously instead of gathering 128-byte rows through rperm. */
} equigspr_graph;

int         equigspr_abi_version(void);
const char* equigspr_last_error(void);
/* reads and thereby clears the HIP runtime's sticky per-thread "last error" (returns it): for a caller that has just handled a failed
 * operation of its own -- an aborted hipGraph capture around these launches -- and goes on using the library (train.GraphedStep) */
int         equigspr_clear_device_error(void);

/* ---- parameter block ---------------------------------------------------------------
 * All EGNN parameters live in ONE flat fp32 buffer (and their gradients in another of the
 * same layout -- the single bucket the data-parallel all-reduce sends over xGMI).  Order =
 * the reference module's named_parameters() order, each tensor in PyTorch [out,in] layout:
 *   embedding_in.weight [hid,in], embedding_in.bias, embedding_out.weight [out,hid], .bias,
 *   per layer i: for head j: edge_mlps.j.0.weight [hid/H, 2*hid+12+in_edge_nf], .0.bias,
 *                            edge_mlps.j.2.weight [hid/H, hid/H], .2.bias;
 *                layer_norm.weight, .bias; node_mlp.0.weight [hid,2*hid], .0.bias,
 *                node_mlp.2.weight [hid,hid], .2.bias; coord_mlp.0.weight [hid,hid], .0.bias,
 *                coord_mlp.2.weight [1,hid].
 * (state_dict keys: src/3dmatch_train_egnn_with_batch.py:202-229,:320-325.) */
int64_t equigspr_egnn_param_count(const equigspr_egnn_config* cfg);
/* offset (in floats) of layer i's block inside the flat buffer; i == n_layers -> total */
int64_t equigspr_egnn_layer_offset(const equigspr_egnn_config* cfg, int32_t layer);

/* ---- A1: k-NN graph -----------------------------------------------------------------
 * Replaces torch_cluster.knn_graph(x, k, loop) as called per cloud at
 * src/3dmatch_train_egnn_with_batch.py:1005-1006 (eval :1156-1157, kitti :975-976), for
 * n_clouds equally sized clouds in one launch.
 *   x          [n_clouds*n, 3]
 *   edges_i64  [n_clouds, 2, n*k] or NULL: per cloud [neighbour; centre], LOCAL indices,
 *              centre-major, ascending (d2, index) -- the reference LongTensor layout.
 *   row32/col32 [n_clouds*n*k] or NULL: same edges as GLOBAL int32 node ids (device format).
 *   row_hist   [n_clouds*n] int32 or NULL, +=: row_hist[i] += number of edges whose row (neighbour) is node i -- the
 *              histogram equigspr_build_csr(key = row32) starts from (pass EQUIGSPR_SCRATCH_HISTOGRAM there).
 *   arrival    [n_clouds*n*k] int32 or NULL (ABI 10; needs row_hist): arrival[e] = the value row_hist[row32[e]] had when edge e
 *              was counted -- a unique slot 0 .. deg-1 inside its row segment, in arrival order.  Handed to equigspr_build_csr
 *              it turns the index fill into a plain scatter: the returning atomics (the slow part of the fill: scattered
 *              returning atomics run at ~25 G/s chip-wide) are issued here, under the search's arithmetic.
 * Distance contract: d2 = fl(fl(fl(dx*dx)+fl(dy*dy))+fl(dz*dz)), ties by lower index.
 * Limits: 1 <= k <= n (k <= n-1 when loop==0); n_clouds*n*k < 2^31.  k <= 64 and n <= 8192 run on the tuned kernels (a wave per
 * query up to 2 048 points, LDS candidate tiles above); larger k or n (torch_cluster accepts any: src/gcnn.py:31) take a streaming
 * kernel with the same output order (ABI 14; O(k n) distance evaluations per query). */
int equigspr_knn_graph(const float* x, int32_t n_clouds, int32_t n, int32_t k, int32_t loop,
                       int64_t* edges_i64, int32_t* row32, int32_t* col32, int32_t* row_hist, int32_t* arrival,
                       int32_t force_slow_path, void* stream);

/* ---- graph index build ---------------------------------------------------------------
 * Builds rptr/rperm (key=row) or cptr/cperm (key=col): stable counting sort of edge ids.
 * Replaces the implicit scatter_add_ indexing of unsorted_segment_sum (:343-348).
 *   key [E] global ids in [0,N);  ptr [N+1];  perm [E];  inv [E] or NULL: the inverse of perm (inv[perm[p]] = p, the
 *   rinv field of equigspr_graph), written by the same pass that orders the segments;  scratch: int32 [N+1].
 *   scratch_state says what scratch holds on entry: EQUIGSPR_SCRATCH_DIRTY (anything: it is cleared first),
 *   EQUIGSPR_SCRATCH_ZERO (all-zero: the clear is skipped) or EQUIGSPR_SCRATCH_HISTOGRAM (scratch[i] = number of edges
 *   with key i, e.g. the row_hist output of equigspr_knn_graph for key = row32: clear and count are skipped).  On return
 *   scratch is all-zero again, so one buffer can be handed from call to call on the same stream.
 *   arrival [E] or NULL (ABI 10; with EQUIGSPR_SCRATCH_HISTOGRAM): a unique slot of every edge inside its key's segment (the
 *   `arrival` output of equigspr_knn_graph): the fill is then perm[ptr[key[e]] + arrival[e]] = e, no atomics; the ordering pass
 *   that follows makes the result the same stable sort either way. */
#define EQUIGSPR_SCRATCH_DIRTY 0
#define EQUIGSPR_SCRATCH_ZERO 1
#define EQUIGSPR_SCRATCH_HISTOGRAM 2
int equigspr_build_csr(const int32_t* key, int32_t n_edges, int32_t n_nodes, int32_t* ptr, int32_t* perm,
                       int32_t* inv, int32_t* scratch, int32_t scratch_state, const int32_t* arrival, void* stream);
/* inv[perm[p]] = p for p in [0, n): the rinv field of equigspr_graph from rperm */
int equigspr_invert_perm(const int32_t* perm, int32_t n, int32_t* inv, void* stream);
/* int64 reference edge list [2,E] (one cloud, local ids) -> int32 global ids (+node_offset) */
int equigspr_edges_i64_to_i32(const int64_t* edges_i64, int32_t n_edges, int32_t node_offset,
                              int32_t* row32, int32_t* col32, void* stream);

/* ---- A3-A9: EGNN forward ---------------------------------------------------------------
 * Replaces EGNN.forward(h, x, edges, edge_attr) (src/3dmatch_train_egnn_with_batch.py:328-340)
 * = embedding_in, n_layers x E_GCL.forward (:280-289: coord2radial :271, edge_model :231,
 * coord_model :262, node_model :252), embedding_out -- for ALL clouds of a batch at once.
 *   params    flat parameter block (above)
 *   h_in      [N, in_nf]     x_in [N,3]     edge_attr [E, in_edge_nf] or NULL (= all ones)
 *   h_out     [N, out_nf]    x_out [N,3]
 *   saved     float[equigspr_egnn_saved_size]   activations kept for backward (NULL = inference): per layer the node
 *             states h, x, agg.  Nothing is kept per edge: the edge backward recomputes the edge forward from the node
 *             states (flash-style), so the training-mode forward runs the same edge kernel as inference and the size does
 *             not depend on n_edges (the argument is kept for ABI stability)
 *   work      float[equigspr_egnn_work_size]    scratch */
int64_t equigspr_egnn_saved_size(const equigspr_egnn_config* cfg, int32_t n_nodes, int32_t n_edges);
int64_t equigspr_egnn_work_size(const equigspr_egnn_config* cfg, int32_t n_nodes, int32_t n_edges);
int equigspr_egnn_forward(const equigspr_egnn_config* cfg, const float* params,
                          const equigspr_graph* g, const float* h_in, const float* x_in,
                          const float* edge_attr, float* h_out, float* x_out,
                          float* saved, float* work, void* stream);

/* ---- A14: EGNN backward --------------------------------------------------------------
 * Replaces autograd through EGNN.forward (loss.backward(), :1126).  Recomputes the per-edge
 * activations from `saved`.
 *   grad_h_out [N,out_nf], grad_x_out [N,3]    incoming
 *   grad_params flat, same layout as params: ACCUMULATED into (+=) (caller zeroes)
 *   grad_h_in [N,in_nf] or NULL, grad_x_in [N,3] or NULL   (overwritten)
 *   work  float[equigspr_egnn_bwd_work_size] */
int64_t equigspr_egnn_bwd_work_size(const equigspr_egnn_config* cfg, int32_t n_nodes, int32_t n_edges);
int equigspr_egnn_backward(const equigspr_egnn_config* cfg, const float* params,
                           const equigspr_graph* g, const float* h_in, const float* edge_attr,
                           const float* saved, const float* grad_h_out, const float* grad_x_out,
                           float* grad_params, float* grad_h_in, float* grad_x_in,
                           float* work, void* stream);

/* ---- A12: batched weighted Kabsch / 3x3 SVD pose head ---------------------------------
 * Replaces the per-sample loop body at src/3dmatch_train_egnn_with_batch.py:726-758
 * (eval :786-818): weighted centroids, H = sum w (s-cs)(t-ct)^T + 1e-6 I, SVD, R = V U^T with
 * the det<0 fix on the smallest singular direction, t = ct - R cs.  Zero-weight points do
 * not contribute (masking by weight); all-zero weights give R = I, t = 0 (:708-711).
 *   src,tgt [B,n,3]   w [B,n]   R [B,9] row-major   t [B,3]   aux float[B*EQUIGSPR_KABSCH_AUX]
 * Backward: closed form through the polar factor (1/(s_i+s_j) terms), finite for repeated
 * singular values. */
#define EQUIGSPR_KABSCH_AUX 320
int equigspr_kabsch_forward(const float* src, const float* tgt, const float* w, int32_t batch,
                            int32_t n, float* R, float* t, float* aux, void* stream);
int equigspr_kabsch_backward(const float* src, const float* tgt, const float* w, const float* aux,
                             const float* grad_R, const float* grad_t, int32_t batch, int32_t n,
                             float* grad_src, float* grad_tgt, float* grad_w, void* stream);

/* ---- A10: fused training pose head ------------------------------------------------------
 * Replaces the body of CrossAttentionPoseRegression.forward AFTER the EGNN calls
 * (src/3dmatch_train_egnn_with_batch.py:677-796) except the Kabsch solve (equigspr_kabsch_*):
 * sim = <h_s,h_t> (:681) and raw_sim (:682), top-K by sim (:684, ties by lower index), gather +
 * Linear-ReLU-Linear-ReLU-Linear logits (:760-770), BCEWithLogits (:772-773), the standardised
 * similarity MSE (:777-781, batch-wide mean / unbiased std), egnn_equi_loss (:860-893), and the
 * label-masked softmax Kabsch weights (:717-724; invalid points get weight 0).
 *   hs,ht [B,N,F]  raw_hs,raw_ht [B,N,F_raw]  xs,xt [B,N,3]  labels [B,N] (0/1)  gt_pose [B,16]
 *   mlp: flat [w1 (F x 2F), b1, w2 (F/2 x F), b2, w3 (F/2), b3]  (equigspr_head_mlp_size floats)
 * Outputs kept for backward: sim,rsim,cosv [B,N], pos int32 [B,N], wts [B,N], top_idx int64 [B,K],
 * logits [B,K], a1 [B,K,F], a2 [B,K,F/2]; out_scalars (16 floats): [0] = corr_loss+sim_loss, [1] = equi
 * loss, [2..9] = statistics reused by the backward.  work: equigspr_head_work_bytes bytes (8-byte aligned),
 * must be passed unchanged to the backward.  Limits: F <= 64, K <= N <= 8192 (K = N: every point is scored, as the KITTI script's
 * top_k = 2048 of 2 048 points does, src/kitti_train_egnn_with_batch.py:661).
 * flags: EQUIGSPR_HEAD_NO_SIM_LOSS = the KITTI script's forward, which has no similarity-MSE term and returns corr_loss alone
 * (kitti :766); pass the same flags to the backward.
 * kabsch_R [B,9], kabsch_t [B,3], kabsch_aux [B*EQUIGSPR_KABSCH_AUX] (all three or all NULL): also performs
 * equigspr_kabsch_forward(xs, xt, wts, ...) -- identical results and aux record -- inside the launch that finishes the
 * loss scalars, instead of two more launches.  Launches: three (per-node similarities; per pair the selection beside the weight /
 * Kabsch blocks; the MLP on the selected rows), or two when B * (1 + N/256 + K/8) <= 512 workgroups and kabsch_aux is given: the
 * MLP workgroups then ride in the second launch and wait for their pair's selection (csrc/head.hip, head_mlp_block).
 * Backward: g_corr_sim / g_equi are DEVICE scalars (may be NULL = 0); gw = dL/d wts from
 * equigspr_kabsch_backward -- or, with kabsch_aux / grad_R / grad_t given (ABI 8), that Kabsch backward runs inside this call's
 * first launch beside the MLP backward: gw is then an OUTPUT (scratch [B*N]) and gxs, gxt are written first (grad_R / grad_t may
 * be NULL = zero).  step_loss (forward, ABI 9; needs kabsch_aux): [1 + 2B] floats = the training loop's scalar loss
 * corr_loss + sim_loss + mean rot_loss + mean trans_loss (train :1117-1120), then rot_loss[B], trans_loss[B] -- evaluated inside the
 * head's launches instead of equigspr_step_loss_forward; g_step_loss (backward, device scalar, needs kabsch_t) is the upstream
 * gradient of that scalar: its pose terms join grad_R / grad_t, its corr/sim term joins g_corr_sim.  writes ghs,ght; ADDS the equi-loss terms into gxs,gxt (which hold the
 * Kabsch coordinate gradients); ADDS into grad_mlp. */
#define EQUIGSPR_HEAD_NO_SIM_LOSS 1
int64_t equigspr_head_mlp_size(int32_t F);
int64_t equigspr_head_work_bytes(int32_t B, int32_t N);
int64_t equigspr_head_slab_count(int32_t B, int32_t K);   /* mlp_slab = slab_count * head_mlp_size floats */
int equigspr_head_forward(const float* hs, const float* ht, const float* raw_hs, const float* raw_ht,
                          const float* xs, const float* xt, const float* labels, const float* gt_pose,
                          const float* mlp, int32_t B, int32_t N, int32_t F, int32_t F_raw, int32_t K, int32_t flags,
                          float* sim, float* rsim, float* cosv, int32_t* pos, float* wts, int64_t* top_idx,
                          float* logits, float* a1_save, float* a2_save, void* work, float* out_scalars,
                          float* kabsch_R, float* kabsch_t, float* kabsch_aux, float* step_loss, void* stream);
int equigspr_head_backward(const float* hs, const float* ht, const float* xs, const float* xt,
                           const float* labels, const float* gt_pose, const float* mlp, const float* sim,
                           const float* rsim, const float* cosv, const int32_t* pos, const float* wts,
                           const int64_t* top_idx, const float* logits, const float* a1_save,
                           const float* a2_save, const float* fin_scalars, const float* g_corr_sim,
                           const float* g_equi, float* gw, const float* kabsch_aux, const float* grad_R, const float* grad_t,
                           const float* kabsch_t, const float* g_step_loss, int32_t B, int32_t N, int32_t F, int32_t K, int32_t flags,
                           void* work, float* dinp, float* mlp_slab, float* grad_mlp, float* ghs, float* ght,
                           float* gxs, float* gxt, void* stream);

/* ---- A11: fused evaluation pair model ---------------------------------------------------------
 * Replaces the body of the EVALUATION CrossAttentionPoseRegression.forward after the EGNN calls
 * (src/eval_egnn_metrics.py:687-818): egnn_equi_loss on the EGNN outputs (:687), sim0 = <h0_s, h0_t> on the INPUT features
 * (:691), top-K of sim0 (:694, ties by lower index), the correspondence MLP on the gathered EGNN features (:706-742), the
 * conditional substitution of :758-768 INCLUDING its [K] vs [1,K,1] broadcast (the `where` result is [1,K,K] and scatter_ with
 * an index of shape [1,K,1] consumes column 0: every top-K slot i receives where(cond(p[0], sim0[idx_i]), p[0], sim0[idx_i]),
 * p[0] = the MLP score of the best-ranked row), W / (sum W + 1e-6) (:771), softmax over the N points (:774), / (sum + 1e-6)
 * (:783), and the weighted Kabsch on the RAW coordinates of all N points (:786-818).  Inference only (the reference never
 * back-propagates through this head).  Three launches.
 *   hs, ht [B,N,F], xs, xt [B,N,3]: EGNN outputs;  raw_hs, raw_ht [B,N,F_raw], raw_xs, raw_xt [B,N,3]: the model inputs
 *   outputs: sim [B,N] (<h_s,h_t>, unused by this head), sim0 [B,N], cosv [B,N], pos int32 [B,N] (rank in the top-K or -1),
 *            wts [B,N] (the Kabsch weights), top_idx int64 [B,K], kabsch_R [B,9], kabsch_t [B,3], kabsch_aux [B,EQUIGSPR_KABSCH_AUX],
 *            out_scalars [16] device: [1] = equi loss, [10 + min(b,5)] = p[0] of pair b
 *   work: equigspr_head_work_bytes(B, N) bytes.  The reference runs this head at B = 1 only (:719,:773); pairs are independent here. */
int equigspr_head_eval_forward(const float* hs, const float* ht, const float* raw_hs, const float* raw_ht,
                               const float* xs, const float* xt, const float* raw_xs, const float* raw_xt,
                               const float* labels, const float* gt_pose, const float* mlp, int32_t B, int32_t N,
                               int32_t F, int32_t F_raw, int32_t K, float* sim, float* sim0, float* cosv, int32_t* pos,
                               float* wts, int64_t* top_idx, void* work, float* out_scalars, float* kabsch_R,
                               float* kabsch_t, float* kabsch_aux, void* stream);

/* head_fuses_mlp: 1 when equigspr_head_forward (with its own Kabsch) runs the MLP blocks of this shape as extra grid columns of
 * its pair launch -- only when the whole fused grid is resident on the current device at the launch's LDS size (occupancy query,
 * cached), so that the columns' wait for their pair's selection rests on co-residency and not on dispatch order; 0: the MLP is its
 * own launch.  head_wait_failures: how many such bounded waits have ever run out on this process's current device (they poison the
 * affected logits and the loss with NaN; 0 on a healthy device, -1 on error); synchronises.  (train :677-796 has no counterpart:
 * these describe the launch structure that replaces its ~230 ATen launches.) */
int     equigspr_head_fuses_mlp(int32_t B, int32_t N, int32_t F, int32_t K);
int64_t equigspr_head_wait_failures(void);

/* ---- A13 (pose part) + optimizer: small fused training-step kernels ------------------------
 * pose_loss: replaces pose_loss() (src/3dmatch_train_egnn_with_batch.py:896-962): rot_loss[b] =
 * arccos(clamp((trace(R^T R_gt)-1)/2)), trans_loss[b] = arccos(clamp(cos(t, t_gt))).  R [B,9], t [B,3],
 * gt_pose [B,16]; backward takes dL/drot_loss, dL/dtrans_loss [B] (NULL = 0) and OVERWRITES grad_R, grad_t.
 * The gradient is -inf at zero pose error, exactly as in the reference.
 * adam_step: torch.optim.Adam(lr, betas, eps) semantics (train :1412) over 1..4 flat fp32 buckets in one
 * launch.  The pointer/size arrays are HOST arrays of n_buckets entries; step_dev is DEVICE memory, two zero-
 * initialised 32-bit words (the step count as a float, and the ticket with which the last workgroup publishes
 * step+1), so the call is hipGraph-capturable.  clear_grads != 0: every gradient is zeroed as it is consumed
 * (optimizer.zero_grad() fused into the step: the fill launch of the next iteration disappears). */
int equigspr_pose_loss_forward(const float* R, const float* t, const float* gt_pose, int32_t B, float* rot_loss,
                               float* trans_loss, void* stream);
int equigspr_pose_loss_backward(const float* R, const float* t, const float* gt_pose, const float* g_rot,
                                const float* g_trans, int32_t B, float* grad_R, float* grad_t, void* stream);
int equigspr_adam_step(int32_t n_buckets, float* const* params, float* const* grads, float* const* exp_avg,
                       float* const* exp_avg_sq, const int64_t* sizes, float* step_dev, float lr, float beta1,
                       float beta2, float eps, int32_t clear_grads, void* stream);
/* adam_step_guarded: the reference loop's per-step decision -- stop / skip when the loss or any gradient is not finite
 * (train :1124 set_detect_anomaly; here train.run_training_batches) -- taken ON THE DEVICE inside the optimizer launch, so that a
 * captured training step (train.GraphedStep; train_one_epoch :964-1140) replays without a host round trip.  Same update as
 * adam_step with clear_grads = 1 when everything is finite; otherwise parameters, moments and the step count stay as they are
 * and the gradients (and `extra`) are cleared.  All of gradient buckets, `extra_dev[n_extra]` (e.g. the spare words of a joint
 * all-reduce bucket through which ranks signal each other) and `loss_dev[0]` (NULL = not examined) enter the verdict.
 *   hyper_dev [5] device: lr, beta1, beta2, eps, loss weight (read at run time: schedulers act on a captured graph)
 *   stats_dev [8] device, 16-byte aligned, zero-initialised by the caller: [0] sum of loss * weight over applied steps, [1] applied steps,
 *             [2] skipped steps, [3] 1 + number of guarded calls before the first skipped one (0 = none), [4] last loss,
 *             [5] guarded calls, [6] void steps: a non-finite extra_dev[1] (when n_extra > 1) marks a step for which some rank had
 *             no data -- rejected like any other, but counted here instead of [2] / [3].  Gradient buckets must be 16-byte aligned.
 *   progress_host: NULL, or one float of PINNED HOST memory (device-accessible) that receives stats[5] with a system-scope release
 *             store at the end of every call: the launching thread learns "step n has completed" from plain memory -- no HIP event
 *             on the compute stream (an event record after every hipGraph launch measured +5 us per step at one pair). */
int equigspr_adam_step_guarded(int32_t n_buckets, float* const* params, float* const* grads, float* const* exp_avg,
                               float* const* exp_avg_sq, const int64_t* sizes, float* step_dev, const float* hyper_dev,
                               const float* loss_dev, float* stats_dev, float* extra_dev, int32_t n_extra,
                               float* progress_host, void* stream);

/* ---- A15: SO3TensorProductLayer.forward (src/clean_so3_model.py:34-52) ------------------------
 * y = W2 relu(W1 vec(D x x^T) + b1) + b2, D = input_dim (the bmm of the two broadcast views sums D identical
 * products).  x [n_rows, D], w1 [2*out, D*D], w2 [out, 2*out], y [n_rows, out]; w1t_scratch: D*D*2*out floats.
 * Compiled for input_dim = output_dim = 32 (the reference's instantiation, :85); inference only. */
int equigspr_so3_tensor_product_forward(const float* x, int32_t n_rows, int32_t input_dim, int32_t output_dim,
                                        const float* w1, const float* b1, const float* w2, const float* b2,
                                        float* w1t_scratch, float* y, void* stream);
/* Backward (ABI 14): grad_y [n_rows, 32] -> grad_x [n_rows, 32], grad_w1 [64, 1024], grad_b1 [64], grad_w2 [32, 64], grad_b2 [32] (each may be
 * NULL; written, not accumulated).  The [n_rows, D*D] tensor product is not materialised here either: the hidden layer is recomputed by the
 * forward's first product, dW1 forms its B operand D x_i x_j on the fly (K = rows, 32x32x2 MFMA), grad_x contracts G = W1^T dh block by
 * block in registers.  work: equigspr_so3_tensor_product_backward_work_size(n_rows) floats.  Deterministic. */
int64_t equigspr_so3_tensor_product_backward_work_size(int32_t n_rows);
int equigspr_so3_tensor_product_backward(const float* x, const float* grad_y, int32_t n_rows, int32_t input_dim, int32_t output_dim,
                                         const float* w1, const float* b1, const float* w2, float* work, float* grad_x,
                                         float* grad_w1, float* grad_b1, float* grad_w2, float* grad_b2, void* stream);

/* ---- N4: typed irrep-basis message passing (ABI 13: one kernel path, forward AND backward) -----------
 * The layer the reference's basis (tools/basis.py::get_basis :153-205, tools/irr_repr.py:103-104,
 * tools/spherical_harmonics.py:75-123) was written for and its scripts never build: degree-0 features f0 [N, C0i] and degree-1
 * features f1 [N, C1i, 3] (components in the basis' own real-SH order), pairwise kernels K = sum_J R_J(|r|) K_J(r / |r|) for the
 * four (d_in, d_out) paths of degrees <= 1, messages from the features at `col` aggregated at `row` (train :252-256, :270-278),
 * plus a per-degree channel-mixing self-interaction.  Definition: oracle/irreps_oracle.py.
 *   r = x[row] - x[col], d = |r|, u = r / d;   hid = SiLU(w_r1 d + b_r1) [H];   R = W_r2 hid + b_r2, rows laid out
 *   R00 [C0o, C0i] | R01 [C1o, C0i] | R10 [C0o, C1i] | R11 [C1o, C1i, 3 (J)]   (n_w = C0o C0i + C1o C0i + C0o C1i + 3 C1o C1i);
 *   m0[e, o]    = sum_i R00[o, i] K00 f0[col, i]    + sum_i R10[o, i] (K10 . f1[col, i, :])
 *   m1[e, o, a] = sum_i R01[o, i] K01[a] f0[col, i] + sum_i sum_J R11[o, i, J] (K11[a, :, J] . f1[col, i, :])
 *   out0[n] = W_s0 f0[n] + sum_{e: row = n} m0[e],   out1[n] = W_s1 f1[n] + sum_{e: row = n} m1[e].
 * The basis of degrees <= 1 is evaluated in CLOSED FORM inside the kernels (no [E, 27] tensors):
 *   K00 = k00;  K01[a] = k01 u_a;  K10[b] = k10 u_b;  K11[a, b, 0] = k11_0 delta_ab;  K11[a, b, 1] = k11_1 eps_abc u_c;
 *   K11[a, b, 2] = k11_2 (u_a u_b - delta_ab / 3);  on an edge with r = 0 (self loops) the J >= 1 kernels are zero.
 * The six constants belong to the config: the host fits them to its own get_basis (Q_J is defined up to sign). */
typedef struct equigspr_irrep_config {
    int32_t c0_in, c1_in, c0_out, c1_out;   /* 1..32 channels per degree */
    int32_t hidden;                          /* radial hidden width H: 16 or 32 */
    float k00, k01, k10, k11_0, k11_1, k11_2;
} equigspr_irrep_config;

/* floats of what a forward leaves for the backward: the per-node moments M [N, Lc, H] | Mb [N, Lc] and the per-edge geometry table
 * (unit vector, length and source node of every edge in aggregation order: 20 bytes per edge, computed once per call and read by
 * every kernel instead of the index -> source -> coordinate chain),
 * Lc = 4 ceil16(C0i) + 10 pad(C1i), pad = 8 / 16 / 32;  M[n, j, h] = sum_{e: row = n} c[e, j] hid[e, h], Mb = the same with hid = 1,
 * c[e] = [K00 f0 | K01[a] f0 (a-major) | K10 . f1 | K11[a, :, J] . f1 ((a, J)-major)] (every slot padded to its channel padding). */
int64_t equigspr_irrep_moments_size(const equigspr_irrep_config* cfg, int32_t n_nodes, int32_t n_edges);
/* floats of scratch equigspr_irrep_conv_backward needs */
int64_t equigspr_irrep_bwd_work_size(const equigspr_irrep_config* cfg, int32_t n_nodes, int32_t n_edges);

/* Forward.  keep_moments != 0 (training): two launches (moments: closed-form basis x gathered features x radial hidden
 * activations, aggregated per node in ascending edge order on the matrix cores; node: the radial weights applied to the moments +
 * bias + self-interaction).  keep_moments == 0 (inference): the same two launches, `moments` is scratch (always pass it).
 *   w_r1 [H], b_r1 [H] (radial.0), w_r2 [n_w, H], b_r2 [n_w] (radial.2), w_self0 [C0o, C0i], w_self1 [C1o, C1i];
 *   g: row / col / rptr / rperm (cptr / cperm for the backward); f0 [N, C0i], f1 [N, C1i, 3], x [N, 3];
 *   out0 [N, C0o], out1 [N, C1o, 3]; moments: equigspr_irrep_moments_size floats (kept by the caller for the backward).
 * Deterministic (no atomics), hipGraph-capturable (no allocation, no synchronisation). */
int equigspr_irrep_conv_forward(const equigspr_irrep_config* cfg, const float* w_r1, const float* b_r1, const float* w_r2,
                                const float* b_r2, const float* w_self0, const float* w_self1, const equigspr_graph* g,
                                const float* f0, const float* f1, const float* x, float* out0, float* out1, float* moments,
                                int32_t keep_moments, void* stream);
/* Backward for EVERY input: the features (d_f0, d_f1: the adjoint convolution = the same two kernels on the reversed edges with
 * transposed weights), the geometry (d_x [N, 3], ACCUMULATED: a stack of layers on fixed coordinates sums into one buffer) and
 * all six parameter tensors (OVERWRITTEN).  `moments` = what the forward of the same inputs left.  d_f0 / d_f1 / d_x may be NULL
 * (not computed).  work: equigspr_irrep_bwd_work_size floats. */
int equigspr_irrep_conv_backward(const equigspr_irrep_config* cfg, const float* w_r1, const float* b_r1, const float* w_r2,
                                 const float* b_r2, const float* w_self0, const float* w_self1, const equigspr_graph* g,
                                 const float* f0, const float* f1, const float* x, const float* moments, const float* g_out0,
                                 const float* g_out1, float* d_f0, float* d_f1, float* d_x, float* d_w_r1, float* d_b_r1,
                                 float* d_w_r2, float* d_b_r2, float* d_w_self0, float* d_w_self1, float* work, void* stream);

/* ---- the typed network: a stack of the layer above behind the EGNN interface (irreps.IrrepEGNN; N4, ABI 13) ----------------
 * (h [N, in_nf], x [N, 3]) -> (h_out [N, out_nf], x_out [N, 3]):  f0 = h, f1 = 0 [N, c1, 3];  per layer l = 0 .. L-1:
 *   (p0, p1) = irrep_conv_l(f0, f1, x)  with degree-0 widths in_nf -> hidden_nf -> ... -> out_nf and c1 degree-1 channels throughout;
 *   l < L-1:  f0 = SiLU(p0),  f1 = p1 * sigmoid(W_g p0 + b_g)[:, :, None]   (an invariant gate per degree-1 channel);
 *   h_out = p0 of the last layer,  x_out = x + sum_i w_x[i] p1[:, i, :]       (equivariant coordinate update, cf. train :262-268).
 * The coordinates are fixed through the stack (the graph is built once, train :1005), so d x sums over the layers.
 * Parameter block (fp32, 16-byte aligned; every tensor starts on a 16-byte boundary -- equigspr_irrep_gnn_param_offset): per layer
 *   radial.0.weight [H] | radial.0.bias [H] | radial.2.weight [n_w, H] | radial.2.bias [n_w] | self0.weight | self1.weight,
 * then per gate  weight [c1, c0] | bias [c1],  then the coordinate head w_x [c1]  (which = 0..5 per layer, 6 / 7 gate, 8 head). */
typedef struct equigspr_irrep_gnn_config {
    int32_t in_nf, hidden_nf, out_nf, c1, n_layers, radial_hidden;   /* 1..32 channels, 1..16 layers, radial_hidden 16 or 32 */
    float k00, k01, k10, k11_0, k11_1, k11_2;                        /* closed form of the degree <= 1 basis (see above) */
} equigspr_irrep_gnn_config;
int64_t equigspr_irrep_gnn_param_count(const equigspr_irrep_gnn_config* cfg);     /* floats, padding included */
int64_t equigspr_irrep_gnn_param_offset(const equigspr_irrep_gnn_config* cfg, int32_t layer, int32_t which);
int64_t equigspr_irrep_gnn_saved_size(const equigspr_irrep_gnn_config* cfg, int32_t n_nodes, int32_t n_edges);
int64_t equigspr_irrep_gnn_work_size(const equigspr_irrep_gnn_config* cfg, int32_t n_nodes, int32_t n_edges);
/* forward: 3 launches per layer (moments, node, gate / coordinate head; training == 0: the fused convolution where the shape allows
 * it, 2 launches per layer); `saved` (saved_size floats) holds the layer inputs, the pre-gate outputs and the moments -- the
 * backward's inputs, and the forward's own buffers (always required). */
int equigspr_irrep_gnn_forward(const equigspr_irrep_gnn_config* cfg, const float* params, const equigspr_graph* g, const float* h_in,
                               const float* x, float* h_out, float* x_out, float* saved, int32_t training, void* stream);
/* backward: parameter gradients are ACCUMULATED into `grads` (same layout as `params`; the flat gradient bucket of the optimizer and
 * the all-reduce), d_x [N, 3] is ACCUMULATED (may be NULL), d_h_in [N, in_nf] is overwritten (may be NULL). */
int equigspr_irrep_gnn_backward(const equigspr_irrep_gnn_config* cfg, const float* params, const equigspr_graph* g, const float* h_in,
                                const float* x, const float* saved, const float* g_h, const float* g_x, float* d_h_in, float* d_x,
                                float* grads, float* work, void* stream);

/* ---- A17: gcnLayer (src/gcnLayer.py) ------------------------------------------------------------
 * build_graph :58-67 (_get_weights :28-46, get_normalize_adj :49-56): w_ij = exp(-d_ij^2 / sigma2), sigma2 =
 * mean_i(d(i, k-th neighbour))^2, no self loops, symmetrised by max, D^-1/2 A D^-1/2 -- returned as CSR by node
 * with LOCAL column ids (fp64 arithmetic, fp32 storage) instead of the reference's scipy [N,N] matrix.
 *   coords [n_clouds*n,3]; nbr/col = row32/col32 of equigspr_knn_graph(loop=1) (k includes the point itself, as
 *   cKDTree.query(k) does); rptr/rperm = equigspr_build_csr(key=nbr).  sigma2 double[n_clouds]; cnt_scratch int32[N+1];
 *   dinv float[N]; adj_ptr int32[N+1]; adj_idx/adj_val [2*N*(k-1)] (upper bound on the symmetrised edge count). */
int equigspr_gcn_build_graph(const float* coords, int32_t n_clouds, int32_t n, int32_t k, const int32_t* nbr,
                             const int32_t* col, const int32_t* rptr, const int32_t* rperm, double* sigma2,
                             int32_t* cnt_scratch, float* dinv, int32_t* adj_ptr, int32_t* adj_idx, float* adj_val,
                             void* stream);
/* scatter the CSR into a zero-initialised dense [n_clouds, n, n] (parity with the reference matrix) */
int equigspr_gcn_adjacency_to_dense(const int32_t* adj_ptr, const int32_t* adj_idx, const float* adj_val,
                                    int32_t n_clouds, int32_t n, float* dense_zeroed, void* stream);
/* GraphConvolution.forward :77-80 = Conv1d_1x1(bmm(x, adj)):  x [B,C_in,n] -> y [B,C_out,n].
 * adj_ptr != NULL: fused sparse gather over the CSR (adj_clouds = 1 shares one adjacency, = B one per sample);
 * adj_ptr == NULL: x already holds bmm(x, adj) (dense adjacency multiplied by the caller's library GEMM). */
int equigspr_gcn_conv_forward(const float* x, int32_t n_batch, int32_t c_in, int32_t c_out, int32_t n,
                              const int32_t* adj_ptr, const int32_t* adj_idx, const float* adj_val, int32_t adj_clouds,
                              const float* weight, const float* bias, float* y, void* stream);
/* GlobalPooling.forward :90-94: x [B,C,n] -> out [B,2C] = [max over n | mean over n] */
int equigspr_gcn_global_pool(const float* x, int32_t n_batch, int32_t channels, int32_t n, float* out, void* stream);
/* Backward of the two launches above (ABI 14; the reference never trains these layers -- gcnLayer.py:100-107 is a demo -- here they are
 * trainable modules).  Weights: grad_w [c_out, c_in] = sum_{b,i} grad_y[b,o,i] agg[b,c,i] with agg = x aggregated over the adjacency
 * (recomputed into agg_scratch [n_batch, c_in, n]; dense-adjacency callers pass x @ adj there and adj_ptr = NULL), grad_b [c_out] (or NULL).
 * The INPUT gradient is equigspr_gcn_conv_forward itself on grad_y with the transposed weight and no bias (symmetric adjacency).
 * Pooling: grad_x[b,c,i] = grad_out[b, C + c] / n + (x == max ? grad_out[b, c] / #ties : 0)  (torch.amax's split between ties). */
int equigspr_gcn_conv_backward_weights(const float* x, const float* grad_y, int32_t n_batch, int32_t c_in, int32_t c_out, int32_t n,
                                       const int32_t* adj_ptr, const int32_t* adj_idx, const float* adj_val, int32_t adj_clouds,
                                       float* agg_scratch, float* grad_w, float* grad_b, void* stream);
int equigspr_gcn_global_pool_backward(const float* x, const float* grad_out, int32_t n_batch, int32_t channels, int32_t n,
                                      float* grad_x, void* stream);

/* ---- A16: real spherical harmonics and the equivariant kernel basis (tools/) --------------------------
 * equigspr_sh_forward: all degrees 0..max_l in one pass; Y row layout [(max_l+1)^2], degree l at [l^2, (l+1)^2),
 * m = -l..l (tools/spherical_harmonics.py:107-123).  `in` is float or double (in_is_f64) and holds
 *   mode 0: cartesian [n,3]   (basis.py:57-95 axis convention; also writes sph [n,3] = radius, alpha, beta if non-NULL)
 *   mode 1: (theta, phi) [n,2]                       (get_spherical_harmonics(l, theta, phi))
 *   mode 2: spherical [n,3] = (radius, alpha, beta)  (precompute_sh: theta = pi - beta, phi = alpha; irr_repr.py:103-104)
 * fp64 arithmetic; y (fp32) and/or y64 (fp64) receive the result.  max_l <= 8. */
int equigspr_sh_forward(const void* in, int32_t in_is_f64, int64_t n, int32_t mode, int32_t max_l, float* sph, float* y,
                        double* y64, void* stream);
/* get_basis inner product (tools/basis.py:186-198) for one (d_in, d_out):
 *   out [n, (2 d_out+1)(2 d_in+1), nJ] = stack_J( Y_J Q_J^T ),  J = |d_in-d_out| .. d_in+d_out,
 *   q [nJ, (2 d_out+1)(2 d_in+1), 2 (d_in+d_out) + 1] (rows of Q_J, zero padded); y from equigspr_sh_forward (max_l). */
int equigspr_sh_basis_forward(const float* y, int64_t n, int32_t max_l, int32_t d_in, int32_t d_out, const float* q,
                              float* out, void* stream);

/* ---- training-step loss (src/3dmatch_train_egnn_with_batch.py:1120) ------------------------------------------
 * loss = corr_loss + mean_b(rot_loss_b) + mean_b(trans_loss_b) with pose_loss() of :896-962, in one launch;
 * corr_loss: device scalar (the head's BCE + similarity-MSE) or NULL; rot_loss/trans_loss [B] are also returned.
 * backward: grad_R [B,9], grad_t [B,3] = d loss / d(R, t) * grad_loss (device scalar, NULL = 1). */
int equigspr_step_loss_forward(const float* R, const float* t, const float* gt_pose, const float* corr_loss, int32_t B,
                               float* loss, float* rot_loss, float* trans_loss, void* stream);
int equigspr_step_loss_backward(const float* R, const float* t, const float* gt_pose, const float* grad_loss, int32_t B,
                                float* grad_R, float* grad_t, void* stream);

/* ---- N2: evaluation metrics (tools/evaluation_metrics.py:14-43; eval loop src/eval_egnn_metrics.py:1245-1296) ----
 * Per pair b: out[5b..] = { rotation error [deg] = acos(clip((tr(G_R^T R) - 1)/2)), translation error [cm],
 * recall = sqrt(TP/n), precision = TP/n, F1 = 2 P R / (P + R + 1e-6) }, TP = #{ ||R src_i + t - tgt_i|| < tau }.
 * R [B,9] row-major, t [B,3], gt_pose [B,16] row-major 4x4, src/tgt [B,n,3]; fp64 arithmetic; out double[5B]. */
int equigspr_eval_metrics(const float* R, const float* t, const float* gt_pose, const float* src, const float* tgt,
                          int32_t n_pairs, int32_t n, double tau, double* out, void* stream);

/* ---- N1: PointNet encoder (src/3dmatch_train_egnn_with_batch.py:52-96) --------------------------------
 * PointNetLayer.forward :61-68 with aggr='max': out_i = max_{e: tgt(e)=i} Linear2(ReLU(Linear1([h_src(e), pos_src(e) - pos_i])))
 * (0 for a node without incoming edges; relu_out != 0 applies the ReLU PointNet.forward puts after each layer, :89-92).
 *   h [N,c_in], pos [N,3]; src [E] = edge_index[0] (global ids); tptr [N+1] / tperm [E] = equigspr_build_csr(key = edge_index[1])
 *   (tperm NULL: edges already grouped by target in tptr order); w1 [c_out, c_in+3], w2 [c_out, c_out];
 *   work float[2*N*c_out]; out [N,c_out].  c_in, c_out <= 64. */
int equigspr_pointnet_layer_forward(const float* h, const float* pos, int32_t n_nodes, int32_t c_in, int32_t c_out,
                                    const int32_t* src, const int32_t* tptr, const int32_t* tperm, const float* w1,
                                    const float* b1, const float* w2, const float* b2, int32_t relu_out, float* work,
                                    float* out, void* stream);
/* torch_geometric.nn.global_max_pool(x, batch) (:95) for an ascending batch vector: out [n_batches, channels] */
int equigspr_global_max_pool(const float* x, const int64_t* batch_sorted, int32_t n_nodes, int32_t channels,
                             int32_t n_batches, float* out, void* stream);
/* Backward of the two entries above (ABI 14; the reference builds a fresh, untrained PointNet per batch at train :1034 and never
 * optimises it -- here the modules are trainable).  Layer: src / tptr / tperm as in the forward, sptr / sperm = the edges sorted by
 * SOURCE (DeviceGraph.rptr / rperm); grad_out [n, c_out]; grad_h [n, c_in], grad_pos [n, 3], grad_w1 [c_out, c_in + 3], grad_b1 [c_out],
 * grad_w2 [c_out, c_out], grad_b2 [c_out]: each may be NULL, written (not accumulated).  The maximum's gradient goes to ONE incoming edge
 * per channel -- the first maximum in the target's segment, as torch_geometric's max aggregation routes it.  work:
 * equigspr_pointnet_layer_backward_work_size floats.  Pool: grad_x[i][f] = grad_out[b][f] / #ties where x[i][f] is its graph's maximum. */
int64_t equigspr_pointnet_layer_backward_work_size(int32_t n_nodes, int32_t n_edges, int32_t c_in, int32_t c_out);
int equigspr_pointnet_layer_backward(const float* h, const float* pos, int32_t n_nodes, int32_t n_edges, int32_t c_in, int32_t c_out,
                                     const int32_t* src, const int32_t* tptr, const int32_t* tperm, const int32_t* sptr,
                                     const int32_t* sperm, const float* w1, const float* b1, const float* w2, const float* b2,
                                     int32_t relu_out, const float* grad_out, float* work, float* grad_h, float* grad_pos,
                                     float* grad_w1, float* grad_b1, float* grad_w2, float* grad_b2, void* stream);
int equigspr_global_max_pool_backward(const float* x, const int64_t* batch_sorted, int32_t n_nodes, int32_t channels,
                                      int32_t n_batches, const float* grad_out, float* grad_x, void* stream);

/* ---- N3: correspondence construction (data_preprocess/3DMatch_Feature.py:157-196, test split :331-351) ----------
 * nearest_idx[i] = argmin_j sqrt(2 - 2 <src_desc_i, tgt_desc_j> + 1e-6) (first minimum, as np.argmin; unit-norm descriptors),
 * nearest_dist[i] (optional) that minimum; fused MFMA dot products + running argmin, the N x M matrix is never stored.
 * src_desc [n_src, dim], tgt_desc [n_tgt, dim], dim <= 64.  Call it with the roles swapped for the mutual check. */
int equigspr_match_descriptors(const float* src_desc, int32_t n_src, const float* tgt_desc, int32_t n_tgt, int32_t dim,
                               int64_t* nearest_idx, float* nearest_dist, void* stream);
/* labels[c] = || R_gt src_pts[corr[c][0]] + t_gt - tgt_pts[corr[c][1]] || < inlier_threshold  (:186-190);
 * corr int64 [n_corr, 2], gt_pose row-major 4x4, labels int32 [n_corr], dist (optional) float [n_corr]. */
int equigspr_correspondence_labels(const float* src_pts, const float* tgt_pts, const int64_t* corr, int32_t n_corr,
                                   const float* gt_pose, float inlier_threshold, int32_t* labels, float* dist, void* stream);

/* ---- measurement hooks (bench.py's roofline leg; NOT part of the drop-in path) -----------
 * Average duration in ms (written to HOST memory) of the EGNN message-passing edge kernel,
 * measured with HIP events on `stream` over `iters` back-to-back launches (these two calls
 * synchronise the host).  fwd: layer 0's edge kernel after embedding_in.  bwd: the last
 * layer's edge backward kernel; `saved` from a training-mode equigspr_egnn_forward. */
/* Developer / test knob: chunk size (edges per wave pass) of the hidden_nf = 32 edge kernels: 16, 32, 48, 64, or 0 = automatic
 * (by problem size).  Process-wide; results are bit-identical run to run for any fixed value. */
int equigspr_set_edge_chunk(int32_t chunk);
/* Arithmetic of the two 32 x 32 products of the edge MLP (W2, W6; in the backward also their transposes) in the hidden_nf 32 / 33
 * edge kernels -- where the reference computes `nn.Linear` in fp32 (train :231-268):
 *   0 = split-bf16 (default): six bf16 x bf16 -> fp32 matrix products on 3-way split operands, <= 1e-6 relative from the fp32 product;
 *   1 = fp32 matrix products (v_mfma_f32_16x16x4_f32: an exact fmaf chain), 5-9 % slower at batch scale.
 * Process-wide, takes effect at the next launch (hipGraphs captured before keep the kernels they captured).  The environment
 * variable EQUIGSPR_EDGE_PRODUCTS=fp32|bf16x3 sets the initial value.  Both forms are bit-reproducible run to run. */
int equigspr_set_edge_products(int32_t fp32);
/* Developer / test knob: upper bound on the workgroups of the 16-edge kernels (0 = the default, one persistent workgroup per CU slot).
 * A small cap makes every wave run many units on a small graph: the regime the reproducibility tests of the split products need
 * (tests/test_gpu_parity.py::test_split_products_match_fp32_products_tile_by_tile).  EQUIGSPR_EDGE_MAXBLOCKS sets the initial value. */
int equigspr_set_edge_grid_cap(int32_t max_blocks);
int equigspr_time_edge_fwd(const equigspr_egnn_config* cfg, const float* params, const equigspr_graph* g,
                           const float* h_in, const float* x_in, const float* edge_attr, float* work,
                           int32_t iters, float* ms_out_host, void* stream);
int equigspr_time_edge_bwd(const equigspr_egnn_config* cfg, const float* params, const equigspr_graph* g,
                           const float* edge_attr, const float* saved, float* bwd_work, int32_t iters,
                           float* ms_out_host, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EQUIGSPR_H */
