// C-ABI glue that is not a kernel: error text, ABI version, parameter-block layout queries.
#include "common.h"

namespace eq {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace eq

extern "C" int equigspr_abi_version(void) { return EQUIGSPR_ABI_VERSION; }
extern "C" const char* equigspr_last_error(void) { return eq::g_err; }

extern "C" int64_t equigspr_egnn_param_count(const equigspr_egnn_config* c) {
    if (!eq::config_supported(c)) return -1;
    return eq::make_model_layout(c->in_nf, c->hidden_nf, c->out_nf, c->in_edge_nf, c->n_layers, c->num_heads).total;
}

extern "C" int64_t equigspr_egnn_layer_offset(const equigspr_egnn_config* c, int32_t layer) {
    if (!eq::config_supported(c) || layer < 0 || layer > c->n_layers) return -1;
    const eq::ModelLayout M = eq::make_model_layout(c->in_nf, c->hidden_nf, c->out_nf, c->in_edge_nf, c->n_layers, c->num_heads);
    return (int64_t)M.layer0 + (int64_t)layer * M.layer_size;
}
