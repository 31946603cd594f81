// placeholder until the backward kernels land (next commit)
#include "common.h"
extern "C" int64_t equigspr_egnn_bwd_work_size(const equigspr_egnn_config* cfg, int32_t n_nodes, int32_t n_edges) { return 64; }
extern "C" int equigspr_egnn_backward(const equigspr_egnn_config* cfg, const float* params, const equigspr_graph* g,
                                      const float* h_in, const float* edge_attr, const float* saved,
                                      const float* grad_h_out, const float* grad_x_out, float* grad_params,
                                      float* grad_h_in, float* grad_x_in, float* work, void* stream) {
    eq::set_error("egnn_backward: not built yet");
    return EQUIGSPR_ENOTSUP;
}
