// TORCH_LIBRARY(equigspr, ...) registration of the C-ABI entry points ("a TORCH_LIBRARY(equigspr, ...) HIP
// extension ... each also reachable as extern "C" int equigspr_<op>(...)").  Host-only C++: every op checks its tensors, allocates
// its outputs through ATen on the inputs' device and calls the SAME C-ABI function of libequigspr.so the ctypes binding uses, on
// the current HIP stream -- so the ops are visible to the dispatcher (torch.ops.equigspr.*), usable from TorchScript / C++ front
// ends, and there is exactly one implementation underneath.  Built by csrc/Makefile into ../libequigspr_torch.so (g++; no device
// code here) and loaded with torch.ops.load_library (se3_equi_graph_registration_amd/torch_library.py).
#include <ATen/ATen.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/autograd.h>
#include <torch/library.h>

#include "../../include/equigspr.h"

namespace {

void* cur_stream(const at::Tensor& t) { return (void*)c10::hip::getCurrentHIPStream(t.device().index()).stream(); }
// every op runs on the device of its first tensor (the C-ABI launches on the CURRENT device: a tensor of another GPU must switch it)
// (c10::DeviceGuard resolves the guard implementation registered for the tensor's device type: on a ROCm build of PyTorch the HIP
// device masquerades as DeviceType::CUDA, which the un-masqueraded c10::hip::HIPGuard rejects)
using DeviceGuard = c10::DeviceGuard;
void same_device(const at::Tensor& a, const at::Tensor& b, const char* name) {
    TORCH_CHECK(a.device() == b.device(), "equigspr: ", name, " lives on ", b.device(), ", expected ", a.device());
}

void check(int rc, const char* what) {
    TORCH_CHECK(rc == 0, "equigspr::", what, ": ", equigspr_last_error(), " (code ", rc, ")");
}
void need(const at::Tensor& t, const char* name, at::ScalarType dt, int64_t dim) {
    TORCH_CHECK(t.is_cuda(), "equigspr: ", name, " must live on the GPU (there is no CPU path)");
    TORCH_CHECK(t.scalar_type() == dt, "equigspr: ", name, " has the wrong dtype");
    TORCH_CHECK(dim < 0 || t.dim() == dim, "equigspr: ", name, " must be ", dim, "-dimensional");
    TORCH_CHECK(t.is_contiguous(), "equigspr: ", name, " must be contiguous");
}

// knn_graph(x [n, 3], k, loop) -> int64 [2, n k]   (torch_cluster.knn_graph as called at train :1005-1006)
at::Tensor knn_graph(const at::Tensor& x, int64_t k, bool loop) {
    need(x, "x", at::kFloat, 2);
    TORCH_CHECK(x.size(1) == 3, "equigspr::knn_graph: x must be [n, 3]");
    const int64_t n = x.size(0);
    TORCH_CHECK(k >= 1 && k <= n, "equigspr::knn_graph: need 1 <= k <= n");
    DeviceGuard guard(x.device());
    auto out = at::empty({1, 2, n * k}, x.options().dtype(at::kLong));
    check(equigspr_knn_graph(x.data_ptr<float>(), 1, (int32_t)n, (int32_t)k, loop ? 1 : 0, out.data_ptr<int64_t>(), nullptr, nullptr,
                             nullptr, nullptr, 0, cur_stream(x)), "knn_graph");
    return out[0];
}

// build_inverse_index(key int32 [E], n_nodes) -> (ptr int32 [N + 1], perm int32 [E], inv int32 [E])   (stable counting sort)
std::tuple<at::Tensor, at::Tensor, at::Tensor> build_inverse_index(const at::Tensor& key, int64_t n_nodes) {
    need(key, "key", at::kInt, 1);
    TORCH_CHECK(n_nodes >= 1, "equigspr::build_inverse_index: n_nodes must be positive");
    DeviceGuard guard(key.device());
    const int64_t e = key.size(0);
    auto o = key.options();
    auto ptr = at::empty({n_nodes + 1}, o), perm = at::empty({e}, o), inv = at::empty({e}, o), scratch = at::empty({n_nodes + 1}, o);
    check(equigspr_build_csr(key.data_ptr<int32_t>(), (int32_t)e, (int32_t)n_nodes, ptr.data_ptr<int32_t>(), perm.data_ptr<int32_t>(),
                             inv.data_ptr<int32_t>(), scratch.data_ptr<int32_t>(), EQUIGSPR_SCRATCH_DIRTY, nullptr, cur_stream(key)), "build_csr");
    return {ptr, perm, inv};
}

// egnn_forward(params, h [N, in], x [N, 3], row int32 [E], col int32 [E], edge_attr [E] or empty, cfg ints) -> (h_out, x_out)
// Inference form of EGNN.forward (train :328-340); the graph's inverse indices are built here.
std::tuple<at::Tensor, at::Tensor> egnn_forward(const at::Tensor& params, const at::Tensor& h, const at::Tensor& x, const at::Tensor& row,
                                                const at::Tensor& col, const c10::optional<at::Tensor>& edge_attr, int64_t hidden_nf,
                                                int64_t out_nf, int64_t n_layers, int64_t num_heads, int64_t edge_basis) {
    need(params, "params", at::kFloat, 1); need(h, "h", at::kFloat, 2); need(x, "x", at::kFloat, 2);
    need(row, "row", at::kInt, 1); need(col, "col", at::kInt, 1);
    const int64_t n = h.size(0), e = row.size(0);
    TORCH_CHECK(x.size(0) == n && x.size(1) == 3, "equigspr::egnn_forward: x must be [", n, ", 3]");
    TORCH_CHECK(col.numel() == e, "equigspr::egnn_forward: row and col must have the same length");
    TORCH_CHECK(!edge_attr.has_value() || edge_attr->numel() == e, "equigspr::egnn_forward: edge_attr must hold one value per edge");
    same_device(h, params, "params"); same_device(h, x, "x"); same_device(h, row, "row"); same_device(h, col, "col");
    if (edge_attr.has_value()) same_device(h, *edge_attr, "edge_attr");
    DeviceGuard guard(h.device());
    equigspr_egnn_config cfg{(int32_t)h.size(1), (int32_t)hidden_nf, (int32_t)out_nf, edge_attr.has_value() ? 1 : 0, (int32_t)n_layers,
                             (int32_t)num_heads, (int32_t)edge_basis};
    const int64_t want = equigspr_egnn_param_count(&cfg);
    TORCH_CHECK(want > 0, "equigspr::egnn_forward: unsupported configuration (in_nf ", h.size(1), ", hidden_nf ", hidden_nf, ", out_nf ", out_nf,
                ", heads ", num_heads, ", layers ", n_layers, "): ", equigspr_last_error());
    TORCH_CHECK(want == params.numel(), "equigspr::egnn_forward: parameter block has ", params.numel(), " floats, the configuration needs ", want);
    if (edge_attr.has_value()) need(*edge_attr, "edge_attr", at::kFloat, -1);
    auto io = row.options();
    auto rptr = at::empty({n + 1}, io), rperm = at::empty({e}, io), rinv = at::empty({e}, io), scratch = at::empty({n + 1}, io);
    void* s = cur_stream(h);
    check(equigspr_build_csr(row.data_ptr<int32_t>(), (int32_t)e, (int32_t)n, rptr.data_ptr<int32_t>(), rperm.data_ptr<int32_t>(),
                             rinv.data_ptr<int32_t>(), scratch.data_ptr<int32_t>(), EQUIGSPR_SCRATCH_DIRTY, nullptr, s), "build_csr");
    equigspr_graph g{(int32_t)n, (int32_t)e, row.data_ptr<int32_t>(), col.data_ptr<int32_t>(), rptr.data_ptr<int32_t>(),
                     rperm.data_ptr<int32_t>(), nullptr, nullptr, rinv.data_ptr<int32_t>()};
    auto fo = h.options();
    auto h_out = at::empty({n, out_nf}, fo), x_out = at::empty({n, 3}, fo);
    auto work = at::empty({equigspr_egnn_work_size(&cfg, (int32_t)n, (int32_t)e)}, fo);
    check(equigspr_egnn_forward(&cfg, params.data_ptr<float>(), &g, h.data_ptr<float>(), x.data_ptr<float>(),
                                edge_attr.has_value() ? edge_attr->data_ptr<float>() : nullptr, h_out.data_ptr<float>(),
                                x_out.data_ptr<float>(), nullptr, work.data_ptr<float>(), s), "egnn_forward");
    return {h_out, x_out};
}

// weighted_kabsch_forward(src [B, n, 3], tgt [B, n, 3], w [B, n]) -> (R [B, 3, 3], t [B, 3], aux)   (train :726-758)
std::tuple<at::Tensor, at::Tensor, at::Tensor> weighted_kabsch_forward(const at::Tensor& src, const at::Tensor& tgt, const at::Tensor& w) {
    need(src, "src", at::kFloat, 3); need(tgt, "tgt", at::kFloat, 3); need(w, "weights", at::kFloat, 2);
    const int64_t b = src.size(0), n = src.size(1);
    TORCH_CHECK(src.size(2) == 3 && tgt.sizes() == src.sizes() && w.size(0) == b && w.size(1) == n,
                "equigspr::weighted_kabsch_forward: need src, tgt [B, n, 3] and weights [B, n]");
    same_device(src, tgt, "tgt"); same_device(src, w, "weights");
    DeviceGuard guard(src.device());
    auto R = at::empty({b, 3, 3}, src.options()), t = at::empty({b, 3}, src.options()), aux = at::empty({b, EQUIGSPR_KABSCH_AUX}, src.options());
    check(equigspr_kabsch_forward(src.data_ptr<float>(), tgt.data_ptr<float>(), w.data_ptr<float>(), (int32_t)b, (int32_t)n,
                                  R.data_ptr<float>(), t.data_ptr<float>(), aux.data_ptr<float>(), cur_stream(src)), "kabsch_forward");
    return {R, t, aux};
}

// weighted_kabsch_backward(src, tgt, w, aux, grad_R, grad_t) -> (grad_src, grad_tgt, grad_w)
std::tuple<at::Tensor, at::Tensor, at::Tensor> weighted_kabsch_backward(const at::Tensor& src, const at::Tensor& tgt, const at::Tensor& w,
                                                                         const at::Tensor& aux, const at::Tensor& gR, const at::Tensor& gt) {
    need(src, "src", at::kFloat, 3); need(tgt, "tgt", at::kFloat, 3); need(w, "weights", at::kFloat, 2); need(aux, "aux", at::kFloat, 2);
    need(gR, "grad_R", at::kFloat, 3); need(gt, "grad_t", at::kFloat, 2);
    const int64_t b = src.size(0), n = src.size(1);
    TORCH_CHECK(src.size(2) == 3 && tgt.sizes() == src.sizes() && w.size(0) == b && w.size(1) == n && aux.size(0) == b &&
                    aux.size(1) == EQUIGSPR_KABSCH_AUX && gR.size(0) == b && gR.size(1) == 3 && gR.size(2) == 3 && gt.size(0) == b && gt.size(1) == 3,
                "equigspr::weighted_kabsch_backward: need src, tgt [B, n, 3], weights [B, n], aux [B, ", EQUIGSPR_KABSCH_AUX, "], grad_R [B, 3, 3], grad_t [B, 3]");
    same_device(src, tgt, "tgt"); same_device(src, w, "weights"); same_device(src, aux, "aux"); same_device(src, gR, "grad_R"); same_device(src, gt, "grad_t");
    DeviceGuard guard(src.device());
    auto gs = at::empty_like(src), gtt = at::empty_like(tgt), gw = at::empty_like(w);
    check(equigspr_kabsch_backward(src.data_ptr<float>(), tgt.data_ptr<float>(), w.data_ptr<float>(), aux.data_ptr<float>(),
                                   gR.data_ptr<float>(), gt.data_ptr<float>(), (int32_t)b, (int32_t)n, gs.data_ptr<float>(),
                                   gtt.data_ptr<float>(), gw.data_ptr<float>(), cur_stream(src)), "kabsch_backward");
    return {gs, gtt, gw};
}

int64_t abi_version() { return equigspr_abi_version(); }

// ---- Autograd key.  weighted_kabsch_forward and egnn_forward are differentiable through the dispatcher: their backward is the C-ABI's
// (closed-form Kabsch backward; equigspr_egnn_backward on the activations the training-form forward kept).
struct KabschFn : torch::autograd::Function<KabschFn> {
    static torch::autograd::variable_list forward(torch::autograd::AutogradContext* ctx, const at::Tensor& src, const at::Tensor& tgt,
                                                  const at::Tensor& w) {
        at::AutoDispatchBelowADInplaceOrView below;
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("equigspr::weighted_kabsch_forward", "")
                             .typed<std::tuple<at::Tensor, at::Tensor, at::Tensor>(const at::Tensor&, const at::Tensor&, const at::Tensor&)>();
        auto out = op.call(src, tgt, w);
        ctx->save_for_backward({src, tgt, w, std::get<2>(out)});
        ctx->mark_non_differentiable({std::get<2>(out)});
        return {std::get<0>(out), std::get<1>(out), std::get<2>(out)};
    }
    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list g) {
        const auto sv = ctx->get_saved_variables();
        const auto gR = g[0].defined() ? g[0].contiguous() : at::zeros({sv[0].size(0), 3, 3}, sv[0].options());
        const auto gt = g[1].defined() ? g[1].contiguous() : at::zeros({sv[0].size(0), 3}, sv[0].options());
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("equigspr::weighted_kabsch_backward", "")
                             .typed<std::tuple<at::Tensor, at::Tensor, at::Tensor>(const at::Tensor&, const at::Tensor&, const at::Tensor&,
                                                                                  const at::Tensor&, const at::Tensor&, const at::Tensor&)>();
        auto out = op.call(sv[0], sv[1], sv[2], sv[3], gR, gt);
        return {std::get<0>(out), std::get<1>(out), std::get<2>(out)};
    }
};
std::tuple<at::Tensor, at::Tensor, at::Tensor> weighted_kabsch_forward_autograd(const at::Tensor& src, const at::Tensor& tgt, const at::Tensor& w) {
    auto out = KabschFn::apply(src, tgt, w);
    return {out[0], out[1], out[2]};
}
bool wants_grad(const at::Tensor& t) { return at::GradMode::is_enabled() && t.defined() && t.requires_grad(); }

equigspr_egnn_config make_cfg(const at::Tensor& h, bool has_attr, int64_t hidden_nf, int64_t out_nf, int64_t n_layers, int64_t num_heads,
                              int64_t edge_basis) {
    return equigspr_egnn_config{(int32_t)h.size(1), (int32_t)hidden_nf, (int32_t)out_nf, has_attr ? 1 : 0, (int32_t)n_layers, (int32_t)num_heads,
                                (int32_t)edge_basis};
}

// egnn_backward(params, h, row, col, rptr, rperm, rinv, cptr, cperm, edge_attr?, saved, grad_h_out, grad_x_out, cfg ints)
//   -> (grad_params, grad_h_in, grad_x_in)     (equigspr_egnn_backward: loss.backward() through EGNN.forward, train :1126)
std::tuple<at::Tensor, at::Tensor, at::Tensor> egnn_backward(const at::Tensor& params, const at::Tensor& h, const at::Tensor& row,
                                                             const at::Tensor& col, const at::Tensor& rptr, const at::Tensor& rperm,
                                                             const at::Tensor& rinv, const at::Tensor& cptr, const at::Tensor& cperm,
                                                             const c10::optional<at::Tensor>& edge_attr, const at::Tensor& saved,
                                                             const at::Tensor& gh, const at::Tensor& gx, int64_t hidden_nf, int64_t out_nf,
                                                             int64_t n_layers, int64_t num_heads, int64_t edge_basis) {
    need(params, "params", at::kFloat, 1); need(h, "h", at::kFloat, 2); need(saved, "saved", at::kFloat, 1);
    need(gh, "grad_h_out", at::kFloat, 2); need(gx, "grad_x_out", at::kFloat, 2);
    for (const at::Tensor* t : {&row, &col, &rptr, &rperm, &rinv, &cptr, &cperm}) need(*t, "graph index", at::kInt, 1);
    const int64_t n = h.size(0), e = row.size(0);
    TORCH_CHECK(col.numel() == e && rperm.numel() >= e && rinv.numel() >= e && cperm.numel() >= e && rptr.numel() == n + 1 && cptr.numel() == n + 1,
                "equigspr::egnn_backward: graph index sizes do not match N = ", n, ", E = ", e);
    TORCH_CHECK(gh.size(0) == n && gh.size(1) == out_nf && gx.size(0) == n && gx.size(1) == 3, "equigspr::egnn_backward: grad_h_out [N, out_nf], grad_x_out [N, 3]");
    if (edge_attr.has_value()) { need(*edge_attr, "edge_attr", at::kFloat, -1); TORCH_CHECK(edge_attr->numel() == e, "equigspr::egnn_backward: edge_attr must hold one value per edge"); }
    // raw pointers of all of these go to the kernels under h's device guard: a tensor of another GPU would be an illegal access, not an error
    same_device(h, params, "params"); same_device(h, row, "row"); same_device(h, col, "col"); same_device(h, rptr, "rptr");
    same_device(h, rperm, "rperm"); same_device(h, rinv, "rinv"); same_device(h, cptr, "cptr"); same_device(h, cperm, "cperm");
    same_device(h, saved, "saved"); same_device(h, gh, "grad_h_out"); same_device(h, gx, "grad_x_out");
    if (edge_attr.has_value()) same_device(h, *edge_attr, "edge_attr");
    DeviceGuard guard(h.device());
    auto cfg = make_cfg(h, edge_attr.has_value(), hidden_nf, out_nf, n_layers, num_heads, edge_basis);
    const int64_t want = equigspr_egnn_param_count(&cfg);
    TORCH_CHECK(want > 0 && want == params.numel(), "equigspr::egnn_backward: configuration / parameter block mismatch");
    TORCH_CHECK(saved.numel() >= equigspr_egnn_saved_size(&cfg, (int32_t)n, (int32_t)e), "equigspr::egnn_backward: `saved` is too small");
    equigspr_graph g{(int32_t)n, (int32_t)e, row.data_ptr<int32_t>(), col.data_ptr<int32_t>(), rptr.data_ptr<int32_t>(), rperm.data_ptr<int32_t>(),
                     cptr.data_ptr<int32_t>(), cperm.data_ptr<int32_t>(), rinv.data_ptr<int32_t>()};
    auto gp = at::zeros_like(params);
    auto ghi = at::empty_like(h), gxi = at::empty({n, 3}, h.options());
    auto work = at::empty({equigspr_egnn_bwd_work_size(&cfg, (int32_t)n, (int32_t)e)}, h.options());
    check(equigspr_egnn_backward(&cfg, params.data_ptr<float>(), &g, h.data_ptr<float>(), edge_attr.has_value() ? edge_attr->data_ptr<float>() : nullptr,
                                 saved.data_ptr<float>(), gh.data_ptr<float>(), gx.data_ptr<float>(), gp.data_ptr<float>(), ghi.data_ptr<float>(),
                                 gxi.data_ptr<float>(), work.data_ptr<float>(), cur_stream(h)), "egnn_backward");
    return {gp, ghi, gxi};
}

// Autograd kernel of egnn_forward: the training form of equigspr_egnn_forward (node-level activations kept in `saved`, both inverse
// indices built), backward = equigspr_egnn_backward -- EGNN.forward differentiable through the dispatcher (params, h and x).
struct EgnnFn : torch::autograd::Function<EgnnFn> {
    static torch::autograd::variable_list forward(torch::autograd::AutogradContext* ctx, const at::Tensor& params, const at::Tensor& h_,
                                                  const at::Tensor& x_, const at::Tensor& row, const at::Tensor& col,
                                                  const c10::optional<at::Tensor>& edge_attr, int64_t hidden_nf, int64_t out_nf, int64_t n_layers,
                                                  int64_t num_heads, int64_t edge_basis) {
        at::AutoDispatchBelowADInplaceOrView below;
        // dtype / device before anything is detached or made contiguous (a CPU or other-GPU tensor must be a TORCH_CHECK error)
        for (const auto* t : {&params, &h_, &x_}) {
            TORCH_CHECK(t->is_cuda(), "equigspr::egnn_forward: params, h and x must live on the GPU (there is no CPU path)");
            TORCH_CHECK(t->scalar_type() == at::kFloat, "equigspr::egnn_forward: params, h and x must be float32");
        }
        same_device(h_, params, "params"); same_device(h_, x_, "x"); same_device(h_, row, "row"); same_device(h_, col, "col");
        if (edge_attr.has_value()) same_device(h_, *edge_attr, "edge_attr");
        const at::Tensor p = params.detach().contiguous(), h = h_.detach().contiguous(), x = x_.detach().contiguous();
        need(p, "params", at::kFloat, 1); need(h, "h", at::kFloat, 2); need(x, "x", at::kFloat, 2);
        need(row, "row", at::kInt, 1); need(col, "col", at::kInt, 1);
        const int64_t n = h.size(0), e = row.size(0);
        TORCH_CHECK(x.size(0) == n && x.size(1) == 3 && col.numel() == e, "equigspr::egnn_forward: x must be [N, 3], row and col of one length");
        if (edge_attr.has_value()) { need(*edge_attr, "edge_attr", at::kFloat, -1); TORCH_CHECK(edge_attr->numel() == e, "equigspr::egnn_forward: edge_attr must hold one value per edge"); }
        DeviceGuard guard(h.device());
        auto cfg = make_cfg(h, edge_attr.has_value(), hidden_nf, out_nf, n_layers, num_heads, edge_basis);
        const int64_t want = equigspr_egnn_param_count(&cfg);
        TORCH_CHECK(want > 0, "equigspr::egnn_forward: unsupported configuration: ", equigspr_last_error());
        TORCH_CHECK(want == p.numel(), "equigspr::egnn_forward: parameter block has ", p.numel(), " floats, the configuration needs ", want);
        auto io = row.options();
        auto rptr = at::empty({n + 1}, io), rperm = at::empty({std::max<int64_t>(e, 1)}, io), rinv = at::empty({std::max<int64_t>(e, 1)}, io);
        auto cptr = at::empty({n + 1}, io), cperm = at::empty({std::max<int64_t>(e, 1)}, io), scratch = at::empty({n + 1}, io);
        void* s = cur_stream(h);
        check(equigspr_build_csr(row.data_ptr<int32_t>(), (int32_t)e, (int32_t)n, rptr.data_ptr<int32_t>(), rperm.data_ptr<int32_t>(),
                                 rinv.data_ptr<int32_t>(), scratch.data_ptr<int32_t>(), EQUIGSPR_SCRATCH_DIRTY, nullptr, s), "build_csr(row)");
        check(equigspr_build_csr(col.data_ptr<int32_t>(), (int32_t)e, (int32_t)n, cptr.data_ptr<int32_t>(), cperm.data_ptr<int32_t>(), nullptr,
                                 scratch.data_ptr<int32_t>(), EQUIGSPR_SCRATCH_ZERO, nullptr, s), "build_csr(col)");
        equigspr_graph g{(int32_t)n, (int32_t)e, row.data_ptr<int32_t>(), col.data_ptr<int32_t>(), rptr.data_ptr<int32_t>(), rperm.data_ptr<int32_t>(),
                         cptr.data_ptr<int32_t>(), cperm.data_ptr<int32_t>(), rinv.data_ptr<int32_t>()};
        auto fo = h.options();
        auto h_out = at::empty({n, out_nf}, fo), x_out = at::empty({n, 3}, fo);
        auto saved = at::empty({equigspr_egnn_saved_size(&cfg, (int32_t)n, (int32_t)e)}, fo);
        auto work = at::empty({equigspr_egnn_work_size(&cfg, (int32_t)n, (int32_t)e)}, fo);
        check(equigspr_egnn_forward(&cfg, p.data_ptr<float>(), &g, h.data_ptr<float>(), x.data_ptr<float>(),
                                    edge_attr.has_value() ? edge_attr->data_ptr<float>() : nullptr, h_out.data_ptr<float>(), x_out.data_ptr<float>(),
                                    saved.data_ptr<float>(), work.data_ptr<float>(), s), "egnn_forward");
        ctx->save_for_backward({p, h, row, col, rptr, rperm, rinv, cptr, cperm, saved, edge_attr.has_value() ? *edge_attr : at::Tensor()});
        ctx->saved_data["cfg"] = std::vector<int64_t>{hidden_nf, out_nf, n_layers, num_heads, edge_basis};
        return {h_out, x_out};
    }
    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list g) {
        const auto sv = ctx->get_saved_variables();
        const auto c = ctx->saved_data["cfg"].toIntVector();
        const int64_t n = sv[1].size(0);
        const auto gh = g[0].defined() ? g[0].contiguous() : at::zeros({n, c[1]}, sv[1].options());
        const auto gx = g[1].defined() ? g[1].contiguous() : at::zeros({n, 3}, sv[1].options());
        const c10::optional<at::Tensor> ea = sv[10].defined() ? c10::optional<at::Tensor>(sv[10]) : c10::nullopt;
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("equigspr::egnn_backward", "")
                             .typed<std::tuple<at::Tensor, at::Tensor, at::Tensor>(const at::Tensor&, const at::Tensor&, const at::Tensor&, const at::Tensor&,
                                                                                  const at::Tensor&, const at::Tensor&, const at::Tensor&, const at::Tensor&,
                                                                                  const at::Tensor&, const c10::optional<at::Tensor>&, const at::Tensor&,
                                                                                  const at::Tensor&, const at::Tensor&, int64_t, int64_t, int64_t, int64_t,
                                                                                  int64_t)>();
        auto out = op.call(sv[0], sv[1], sv[2], sv[3], sv[4], sv[5], sv[6], sv[7], sv[8], ea, sv[9], gh, gx, c[0], c[1], c[2], c[3], c[4]);
        return {std::get<0>(out), std::get<1>(out), std::get<2>(out), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(),
                at::Tensor(), at::Tensor()};
    }
};
std::tuple<at::Tensor, at::Tensor> egnn_forward_autograd(const at::Tensor& params, const at::Tensor& h, const at::Tensor& x, const at::Tensor& row,
                                                         const at::Tensor& col, const c10::optional<at::Tensor>& edge_attr, int64_t hidden_nf,
                                                         int64_t out_nf, int64_t n_layers, int64_t num_heads, int64_t edge_basis) {
    if (!wants_grad(params) && !wants_grad(h) && !wants_grad(x)) {          // inference form: nothing kept, no col index
        at::AutoDispatchBelowADInplaceOrView below;
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("equigspr::egnn_forward", "")
                             .typed<std::tuple<at::Tensor, at::Tensor>(const at::Tensor&, const at::Tensor&, const at::Tensor&, const at::Tensor&, const at::Tensor&,
                                                                        const c10::optional<at::Tensor>&, int64_t, int64_t, int64_t, int64_t, int64_t)>();
        return op.call(params, h, x, row, col, edge_attr, hidden_nf, out_nf, n_layers, num_heads, edge_basis);
    }
    auto out = EgnnFn::apply(params, h, x, row, col, edge_attr, hidden_nf, out_nf, n_layers, num_heads, edge_basis);
    return {out[0], out[1]};
}

// ---- one E_GCL layer through the dispatcher (egcl_forward / egcl_backward; reference E_GCL.forward, train :280-289).
// A layer is a one-layer EGNN with identity embeddings: the composite op builds that parameter block out of ATen ops around the
// caller's layer block, so autograd reaches `layer_params`, `h` and `x` through egnn_forward's own autograd kernel.
at::Tensor one_layer_block(const at::Tensor& layer_params, int64_t F, bool has_attr, int64_t num_heads, int64_t edge_basis) {
    equigspr_egnn_config cfg{(int32_t)F, (int32_t)F, (int32_t)F, has_attr ? 1 : 0, 1, (int32_t)num_heads, (int32_t)edge_basis};
    const int64_t total = equigspr_egnn_param_count(&cfg), off = equigspr_egnn_layer_offset(&cfg, 0);
    TORCH_CHECK(total > 0, "equigspr::egcl_forward: unsupported layer configuration (hidden_nf ", F, ", heads ", num_heads, "): ", equigspr_last_error());
    TORCH_CHECK(layer_params.dim() == 1 && layer_params.numel() == total - off,
                "equigspr::egcl_forward: the layer block has ", layer_params.numel(), " floats, hidden_nf ", F, " with ", num_heads, " head(s) needs ",
                total - off);
    TORCH_CHECK(off == 2 * (F * F + F), "equigspr::egcl_forward: unexpected parameter-block layout");
    auto eye = at::eye(F, layer_params.options()).reshape({-1});
    auto zero = at::zeros({F}, layer_params.options());
    return at::cat({eye, zero, eye, zero, layer_params});          // [embedding_in | embedding_out | gcl_0] (include/equigspr.h)
}
std::tuple<at::Tensor, at::Tensor> egcl_forward(const at::Tensor& layer_params, const at::Tensor& h, const at::Tensor& x, const at::Tensor& row,
                                                const at::Tensor& col, const c10::optional<at::Tensor>& edge_attr, int64_t num_heads, int64_t edge_basis) {
    TORCH_CHECK(h.dim() == 2, "equigspr::egcl_forward: h must be [N, hidden_nf]");
    const int64_t F = h.size(1);
    auto block = one_layer_block(layer_params, F, edge_attr.has_value(), num_heads, edge_basis);
    static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("equigspr::egnn_forward", "")
                         .typed<std::tuple<at::Tensor, at::Tensor>(const at::Tensor&, const at::Tensor&, const at::Tensor&, const at::Tensor&, const at::Tensor&,
                                                                    const c10::optional<at::Tensor>&, int64_t, int64_t, int64_t, int64_t, int64_t)>();
    return op.call(block, h, x, row, col, edge_attr, F, F, 1, num_heads, edge_basis);
}
// egcl_backward(layer_params, h, x, row, col, edge_attr?, grad_h_out, grad_x_out, heads, basis) -> (grad_layer_params, grad_h, grad_x):
// the layer's vector-Jacobian product as an explicit op (re-runs the training-form forward, then equigspr_egnn_backward)
std::tuple<at::Tensor, at::Tensor, at::Tensor> egcl_backward(const at::Tensor& layer_params, const at::Tensor& h, const at::Tensor& x,
                                                             const at::Tensor& row, const at::Tensor& col, const c10::optional<at::Tensor>& edge_attr,
                                                             const at::Tensor& gh, const at::Tensor& gx, int64_t num_heads, int64_t edge_basis) {
    at::AutoGradMode grad_on(true);
    auto p = layer_params.detach().clone().requires_grad_(true), hh = h.detach().clone().requires_grad_(true), xx = x.detach().clone().requires_grad_(true);
    auto out = egcl_forward(p, hh, xx, row, col, edge_attr, num_heads, edge_basis);
    auto g = torch::autograd::grad({std::get<0>(out), std::get<1>(out)}, {p, hh, xx}, {gh, gx}, /*retain_graph=*/false, /*create_graph=*/false,
                                   /*allow_unused=*/true);
    auto or_zero = [](const at::Tensor& t, const at::Tensor& like) { return t.defined() ? t : at::zeros_like(like); };
    return {or_zero(g[0], p), or_zero(g[1], hh), or_zero(g[2], xx)};
}

// ---- pose head and losses (inference forms; the training head's autograd node lives in the Python package, ops.py::_HeadFn)
// head_eval_forward(h_all [2BN,F], x_all [2BN,3], raw_hs, raw_ht [B,N,Fr], raw_xs, raw_xt [B,N,3], labels [B,N], gt_pose [B,4,4], mlp, K)
//   -> (R [B,3,3], t [B,3], equi_loss, weights [B,N], top_idx [B,K])        the evaluation pair model after the EGNN (eval :687-818)
std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor> head_eval_forward(
    const at::Tensor& h_all, const at::Tensor& x_all, const at::Tensor& raw_hs, const at::Tensor& raw_ht, const at::Tensor& raw_xs,
    const at::Tensor& raw_xt, const at::Tensor& labels, const at::Tensor& gt_pose, const at::Tensor& mlp, int64_t K) {
    need(h_all, "h_all", at::kFloat, 2); need(x_all, "x_all", at::kFloat, 2); need(raw_hs, "raw_h_src", at::kFloat, 3); need(raw_ht, "raw_h_tgt", at::kFloat, 3);
    need(raw_xs, "raw_x_src", at::kFloat, 3); need(raw_xt, "raw_x_tgt", at::kFloat, 3); need(labels, "labels", at::kFloat, 2);
    need(gt_pose, "gt_pose", at::kFloat, 3); need(mlp, "mlp", at::kFloat, 1);
    const int64_t B = raw_hs.size(0), N = raw_hs.size(1), Fr = raw_hs.size(2), F = h_all.size(1);
    TORCH_CHECK(h_all.size(0) == 2 * B * N && x_all.size(0) == 2 * B * N && x_all.size(1) == 3 && raw_ht.sizes() == raw_hs.sizes() &&
                    raw_xs.size(0) == B && raw_xs.size(1) == N && raw_xs.size(2) == 3 && raw_xt.sizes() == raw_xs.sizes() && labels.size(0) == B &&
                    labels.size(1) == N && gt_pose.size(0) == B && gt_pose.size(1) == 4 && gt_pose.size(2) == 4,
                "equigspr::head_eval_forward: shapes do not describe B pairs of N points");
    TORCH_CHECK(K >= 1 && K <= N && F <= 64 && Fr <= 64 && N <= 8192 && mlp.numel() == equigspr_head_mlp_size((int32_t)F),
                "equigspr::head_eval_forward: need 1 <= K <= N <= 8192, widths <= 64 and the MLP block Linear(2F,F)-ReLU-Linear(F,F/2)-ReLU-Linear(F/2,1)");
    for (const at::Tensor* t : {&x_all, &raw_hs, &raw_ht, &raw_xs, &raw_xt, &labels, &gt_pose, &mlp}) same_device(h_all, *t, "an argument");
    DeviceGuard guard(h_all.device());
    auto fo = h_all.options();
    auto sim = at::empty({B, N}, fo), sim0 = at::empty({B, N}, fo), cosv = at::empty({B, N}, fo), wts = at::empty({B, N}, fo);
    auto pos = at::empty({B, N}, fo.dtype(at::kInt)), top_idx = at::empty({B, K}, fo.dtype(at::kLong));
    auto work = at::empty({equigspr_head_work_bytes((int32_t)B, (int32_t)N) / 8 + 1}, fo.dtype(at::kDouble));
    auto fin = at::empty({16}, fo), R = at::empty({B, 3, 3}, fo), t = at::empty({B, 3}, fo), aux = at::empty({B, EQUIGSPR_KABSCH_AUX}, fo);
    const float* hs = h_all.data_ptr<float>();
    const float* xs = x_all.data_ptr<float>();
    check(equigspr_head_eval_forward(hs, hs + B * N * F, raw_hs.data_ptr<float>(), raw_ht.data_ptr<float>(), xs, xs + B * N * 3, raw_xs.data_ptr<float>(),
                                     raw_xt.data_ptr<float>(), labels.data_ptr<float>(), gt_pose.data_ptr<float>(), mlp.data_ptr<float>(), (int32_t)B,
                                     (int32_t)N, (int32_t)F, (int32_t)Fr, (int32_t)K, sim.data_ptr<float>(), sim0.data_ptr<float>(), cosv.data_ptr<float>(),
                                     pos.data_ptr<int32_t>(), wts.data_ptr<float>(), top_idx.data_ptr<int64_t>(), work.data_ptr<double>(),
                                     fin.data_ptr<float>(), R.data_ptr<float>(), t.data_ptr<float>(), aux.data_ptr<float>(), cur_stream(h_all)),
          "head_eval_forward");
    return {R, t, fin[1], wts, top_idx};
}
// pose_loss_forward(R [B,3,3], t [B,3], gt_pose [B,4,4]) -> (rot_loss [B], trans_loss [B]); pose_loss_backward -> (grad_R, grad_t)   (train :896-962)
std::tuple<at::Tensor, at::Tensor> pose_loss_forward(const at::Tensor& R, const at::Tensor& t, const at::Tensor& gt_pose) {
    need(R, "pred_rot", at::kFloat, 3); need(t, "pred_translation", at::kFloat, 2); need(gt_pose, "gt_pose", at::kFloat, 3);
    const int64_t B = R.size(0);
    TORCH_CHECK(R.size(1) == 3 && R.size(2) == 3 && t.size(0) == B && t.size(1) == 3 && gt_pose.size(0) == B && gt_pose.size(1) == 4 && gt_pose.size(2) == 4,
                "equigspr::pose_loss_forward: need R [B,3,3], t [B,3], gt_pose [B,4,4]");
    same_device(R, t, "pred_translation"); same_device(R, gt_pose, "gt_pose");
    DeviceGuard guard(R.device());
    auto rot = at::empty({B}, R.options()), tr = at::empty({B}, R.options());
    check(equigspr_pose_loss_forward(R.data_ptr<float>(), t.data_ptr<float>(), gt_pose.data_ptr<float>(), (int32_t)B, rot.data_ptr<float>(),
                                     tr.data_ptr<float>(), cur_stream(R)), "pose_loss_forward");
    return {rot, tr};
}
std::tuple<at::Tensor, at::Tensor> pose_loss_backward(const at::Tensor& R, const at::Tensor& t, const at::Tensor& gt_pose, const at::Tensor& g_rot,
                                                      const at::Tensor& g_trans) {
    need(R, "pred_rot", at::kFloat, 3); need(t, "pred_translation", at::kFloat, 2); need(gt_pose, "gt_pose", at::kFloat, 3);
    need(g_rot, "grad_rot_loss", at::kFloat, 1); need(g_trans, "grad_trans_loss", at::kFloat, 1);
    const int64_t B = R.size(0);
    TORCH_CHECK(t.size(0) == B && gt_pose.size(0) == B && g_rot.numel() == B && g_trans.numel() == B, "equigspr::pose_loss_backward: batch sizes differ");
    for (const at::Tensor* q : {&t, &gt_pose, &g_rot, &g_trans}) same_device(R, *q, "an argument");
    DeviceGuard guard(R.device());
    auto gR = at::empty_like(R), gt = at::empty_like(t);
    check(equigspr_pose_loss_backward(R.data_ptr<float>(), t.data_ptr<float>(), gt_pose.data_ptr<float>(), g_rot.data_ptr<float>(), g_trans.data_ptr<float>(),
                                      (int32_t)B, gR.data_ptr<float>(), gt.data_ptr<float>(), cur_stream(R)), "pose_loss_backward");
    return {gR, gt};
}
struct PoseLossFn : torch::autograd::Function<PoseLossFn> {
    static torch::autograd::variable_list forward(torch::autograd::AutogradContext* ctx, const at::Tensor& R, const at::Tensor& t, const at::Tensor& gt) {
        at::AutoDispatchBelowADInplaceOrView below;
        auto Rc = R.detach().contiguous(), tc = t.detach().contiguous(), gc = gt.detach().contiguous();
        auto out = pose_loss_forward(Rc, tc, gc);
        ctx->save_for_backward({Rc, tc, gc});
        return {std::get<0>(out), std::get<1>(out)};
    }
    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list g) {
        const auto sv = ctx->get_saved_variables();
        const auto zero = at::zeros({sv[0].size(0)}, sv[0].options());
        auto out = pose_loss_backward(sv[0], sv[1], sv[2], g[0].defined() ? g[0].contiguous() : zero, g[1].defined() ? g[1].contiguous() : zero);
        return {std::get<0>(out), std::get<1>(out), at::Tensor()};
    }
};
std::tuple<at::Tensor, at::Tensor> pose_loss_forward_autograd(const at::Tensor& R, const at::Tensor& t, const at::Tensor& gt) {
    auto out = PoseLossFn::apply(R, t, gt);
    return {out[0], out[1]};
}

// ---- N4: typed irrep-basis message passing (equigspr_irrep_conv_forward / _backward; no reference layer exists -- the consumer the basis
// of tools/basis.py:153-205 was written for).  basis = the six closed-form constants (k00, k01, k10, k11_0, k11_1, k11_2).
struct IrrepGraph {
    at::Tensor rptr, rperm, rinv, cptr, cperm, scratch;
    equigspr_graph g;
};
IrrepGraph irrep_graph(const at::Tensor& row, const at::Tensor& col, int64_t n, bool need_col, void* s) {
    IrrepGraph G;
    const int64_t e = row.size(0);
    auto io = row.options();
    G.rptr = at::empty({n + 1}, io); G.rperm = at::empty({std::max<int64_t>(e, 1)}, io); G.rinv = at::empty({std::max<int64_t>(e, 1)}, io);
    G.scratch = at::empty({n + 1}, io);
    check(equigspr_build_csr(row.data_ptr<int32_t>(), (int32_t)e, (int32_t)n, G.rptr.data_ptr<int32_t>(), G.rperm.data_ptr<int32_t>(),
                             G.rinv.data_ptr<int32_t>(), G.scratch.data_ptr<int32_t>(), EQUIGSPR_SCRATCH_DIRTY, nullptr, s), "build_csr(row)");
    if (need_col) {
        G.cptr = at::empty({n + 1}, io); G.cperm = at::empty({std::max<int64_t>(e, 1)}, io);
        check(equigspr_build_csr(col.data_ptr<int32_t>(), (int32_t)e, (int32_t)n, G.cptr.data_ptr<int32_t>(), G.cperm.data_ptr<int32_t>(),
                                 nullptr, G.scratch.data_ptr<int32_t>(), EQUIGSPR_SCRATCH_DIRTY, nullptr, s), "build_csr(col)");
    }
    G.g = equigspr_graph{(int32_t)n, (int32_t)e, row.data_ptr<int32_t>(), col.data_ptr<int32_t>(), G.rptr.data_ptr<int32_t>(),
                         G.rperm.data_ptr<int32_t>(), need_col ? G.cptr.data_ptr<int32_t>() : nullptr,
                         need_col ? G.cperm.data_ptr<int32_t>() : nullptr, G.rinv.data_ptr<int32_t>()};
    return G;
}
equigspr_irrep_config irrep_cfg(const at::Tensor& f0, const at::Tensor& f1, const at::Tensor& w_r1, const at::Tensor& ws0, const at::Tensor& ws1,
                                c10::ArrayRef<double> basis) {
    TORCH_CHECK(basis.size() == 6, "equigspr::irrep_conv: basis = (k00, k01, k10, k11_0, k11_1, k11_2)");
    TORCH_CHECK(ws0.dim() == 2 && ws1.dim() == 2 && ws0.size(1) == f0.size(1) && ws1.size(1) == f1.size(1), "equigspr::irrep_conv: w_self0 [C0o, C0i], w_self1 [C1o, C1i]");
    return equigspr_irrep_config{(int32_t)f0.size(1), (int32_t)f1.size(1), (int32_t)ws0.size(0), (int32_t)ws1.size(0), (int32_t)w_r1.numel(),
                                 (float)basis[0], (float)basis[1], (float)basis[2], (float)basis[3], (float)basis[4], (float)basis[5]};
}
void irrep_check(const at::Tensor& f0, const at::Tensor& f1, const at::Tensor& x, const at::Tensor& row, const at::Tensor& col,
                 std::initializer_list<const at::Tensor*> params) {
    need(f0, "f0", at::kFloat, 2); need(f1, "f1", at::kFloat, 3); need(x, "x", at::kFloat, 2); need(row, "row", at::kInt, 1); need(col, "col", at::kInt, 1);
    const int64_t n = x.size(0);
    TORCH_CHECK(x.size(1) == 3 && f0.size(0) == n && f1.size(0) == n && f1.size(2) == 3 && col.numel() == row.numel(),
                "equigspr::irrep_conv: f0 [N, C0], f1 [N, C1, 3], x [N, 3], row / col [E]");
    same_device(f0, f1, "f1"); same_device(f0, x, "x"); same_device(f0, row, "row"); same_device(f0, col, "col");
    for (const at::Tensor* p : params) { need(*p, "parameter", at::kFloat, -1); same_device(f0, *p, "parameter"); }
}
// -> (out0 [N, C0o], out1 [N, C1o, 3], moments)
std::tuple<at::Tensor, at::Tensor, at::Tensor> irrep_conv_forward(const at::Tensor& f0, const at::Tensor& f1, const at::Tensor& x, const at::Tensor& row,
                                                                  const at::Tensor& col, const at::Tensor& w_r1, const at::Tensor& b_r1,
                                                                  const at::Tensor& w_r2, const at::Tensor& b_r2, const at::Tensor& ws0,
                                                                  const at::Tensor& ws1, c10::ArrayRef<double> basis) {
    irrep_check(f0, f1, x, row, col, {&w_r1, &b_r1, &w_r2, &b_r2, &ws0, &ws1});
    DeviceGuard guard(f0.device());
    auto cfg = irrep_cfg(f0, f1, w_r1, ws0, ws1, basis);
    const int64_t n = x.size(0), e = row.size(0);
    const int64_t msz = equigspr_irrep_moments_size(&cfg, (int32_t)n, (int32_t)e);
    TORCH_CHECK(msz >= 0, "equigspr::irrep_conv_forward: unsupported shape (1..32 channels per degree, radial hidden width 16 or 32)");
    const int64_t n_w = (int64_t)cfg.c0_out * cfg.c0_in + (int64_t)cfg.c1_out * cfg.c0_in + (int64_t)cfg.c0_out * cfg.c1_in + 3LL * cfg.c1_out * cfg.c1_in;
    TORCH_CHECK(b_r1.numel() == cfg.hidden && w_r2.numel() == n_w * cfg.hidden && b_r2.numel() == n_w, "equigspr::irrep_conv_forward: radial parameter sizes");
    void* s = cur_stream(f0);
    auto G = irrep_graph(row, col, n, false, s);
    auto o = f0.options();
    auto out0 = at::empty({n, cfg.c0_out}, o), out1 = at::empty({n, cfg.c1_out, 3}, o), moments = at::empty({msz}, o);
    check(equigspr_irrep_conv_forward(&cfg, w_r1.data_ptr<float>(), b_r1.data_ptr<float>(), w_r2.data_ptr<float>(), b_r2.data_ptr<float>(),
                                      ws0.data_ptr<float>(), ws1.data_ptr<float>(), &G.g, f0.data_ptr<float>(), f1.data_ptr<float>(), x.data_ptr<float>(),
                                      out0.data_ptr<float>(), out1.data_ptr<float>(), moments.data_ptr<float>(), 1, s), "irrep_conv_forward");
    return {out0, out1, moments};
}
// -> (d_f0, d_f1, d_x, d_w_r1, d_b_r1, d_w_r2, d_b_r2, d_w_self0, d_w_self1)
std::vector<at::Tensor> irrep_conv_backward(const at::Tensor& f0, const at::Tensor& f1, const at::Tensor& x, const at::Tensor& row, const at::Tensor& col,
                                            const at::Tensor& w_r1, const at::Tensor& b_r1, const at::Tensor& w_r2, const at::Tensor& b_r2,
                                            const at::Tensor& ws0, const at::Tensor& ws1, c10::ArrayRef<double> basis, const at::Tensor& moments,
                                            const at::Tensor& g0, const at::Tensor& g1) {
    irrep_check(f0, f1, x, row, col, {&w_r1, &b_r1, &w_r2, &b_r2, &ws0, &ws1, &moments, &g0, &g1});
    DeviceGuard guard(f0.device());
    auto cfg = irrep_cfg(f0, f1, w_r1, ws0, ws1, basis);
    const int64_t n = x.size(0), e = row.size(0);
    TORCH_CHECK(g0.numel() == n * cfg.c0_out && g1.numel() == n * cfg.c1_out * 3, "equigspr::irrep_conv_backward: g0 [N, C0o], g1 [N, C1o, 3]");
    TORCH_CHECK(moments.numel() == equigspr_irrep_moments_size(&cfg, (int32_t)n, (int32_t)e), "equigspr::irrep_conv_backward: `moments` is not what the forward of these shapes left");
    void* s = cur_stream(f0);
    auto G = irrep_graph(row, col, n, true, s);
    auto o = f0.options();
    auto d_f0 = at::empty_like(f0), d_f1 = at::empty_like(f1), d_x = at::zeros_like(x);
    auto dw1 = at::empty_like(w_r1), db1 = at::empty_like(b_r1), dw2 = at::empty_like(w_r2), db2 = at::empty_like(b_r2), ds0 = at::empty_like(ws0), ds1 = at::empty_like(ws1);
    auto work = at::empty({equigspr_irrep_bwd_work_size(&cfg, (int32_t)n, (int32_t)e)}, o);
    check(equigspr_irrep_conv_backward(&cfg, w_r1.data_ptr<float>(), b_r1.data_ptr<float>(), w_r2.data_ptr<float>(), b_r2.data_ptr<float>(),
                                       ws0.data_ptr<float>(), ws1.data_ptr<float>(), &G.g, f0.data_ptr<float>(), f1.data_ptr<float>(), x.data_ptr<float>(),
                                       moments.data_ptr<float>(), g0.data_ptr<float>(), g1.data_ptr<float>(), d_f0.data_ptr<float>(), d_f1.data_ptr<float>(),
                                       d_x.data_ptr<float>(), dw1.data_ptr<float>(), db1.data_ptr<float>(), dw2.data_ptr<float>(), db2.data_ptr<float>(),
                                       ds0.data_ptr<float>(), ds1.data_ptr<float>(), work.data_ptr<float>(), s), "irrep_conv_backward");
    return {d_f0, d_f1, d_x, dw1, db1, dw2, db2, ds0, ds1};
}
struct IrrepConvFn : torch::autograd::Function<IrrepConvFn> {
    static torch::autograd::variable_list forward(torch::autograd::AutogradContext* ctx, const at::Tensor& f0, const at::Tensor& f1, const at::Tensor& x,
                                                  const at::Tensor& row, const at::Tensor& col, const at::Tensor& w_r1, const at::Tensor& b_r1,
                                                  const at::Tensor& w_r2, const at::Tensor& b_r2, const at::Tensor& ws0, const at::Tensor& ws1,
                                                  std::vector<double> basis) {
        at::AutoDispatchBelowADInplaceOrView below;
        auto out = irrep_conv_forward(f0, f1, x, row, col, w_r1, b_r1, w_r2, b_r2, ws0, ws1, basis);
        ctx->save_for_backward({f0, f1, x, row, col, w_r1, b_r1, w_r2, b_r2, ws0, ws1, std::get<2>(out)});
        ctx->saved_data["basis"] = basis;
        ctx->mark_non_differentiable({std::get<2>(out)});
        return {std::get<0>(out), std::get<1>(out), std::get<2>(out)};
    }
    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list g) {
        const auto sv = ctx->get_saved_variables();
        const auto basis = ctx->saved_data["basis"].toDoubleVector();
        const auto g0 = g[0].defined() ? g[0].contiguous() : at::zeros({sv[0].size(0), sv[9].size(0)}, sv[0].options());
        const auto g1 = g[1].defined() ? g[1].contiguous() : at::zeros({sv[0].size(0), sv[10].size(0), 3}, sv[0].options());
        auto d = irrep_conv_backward(sv[0], sv[1], sv[2], sv[3], sv[4], sv[5], sv[6], sv[7], sv[8], sv[9], sv[10], basis, sv[11], g0, g1);
        return {d[0], d[1], d[2], at::Tensor(), at::Tensor(), d[3], d[4], d[5], d[6], d[7], d[8], at::Tensor()};
    }
};
std::tuple<at::Tensor, at::Tensor, at::Tensor> irrep_conv_forward_autograd(const at::Tensor& f0, const at::Tensor& f1, const at::Tensor& x, const at::Tensor& row,
                                                                           const at::Tensor& col, const at::Tensor& w_r1, const at::Tensor& b_r1,
                                                                           const at::Tensor& w_r2, const at::Tensor& b_r2, const at::Tensor& ws0,
                                                                           const at::Tensor& ws1, c10::ArrayRef<double> basis) {
    auto out = IrrepConvFn::apply(f0, f1, x, row, col, w_r1, b_r1, w_r2, b_r2, ws0, ws1, basis.vec());
    return {out[0], out[1], out[2]};
}
}  // namespace

TORCH_LIBRARY(equigspr, m) {
    m.def("abi_version() -> int", &abi_version);
    m.def("knn_graph(Tensor x, int k, bool loop=True) -> Tensor");
    m.def("build_inverse_index(Tensor key, int n_nodes) -> (Tensor, Tensor, Tensor)");
    m.def("egnn_forward(Tensor params, Tensor h, Tensor x, Tensor row, Tensor col, Tensor? edge_attr, int hidden_nf, int out_nf, "
          "int n_layers, int num_heads=1, int edge_basis=0) -> (Tensor, Tensor)");
    m.def("egnn_backward(Tensor params, Tensor h, Tensor row, Tensor col, Tensor rptr, Tensor rperm, Tensor rinv, Tensor cptr, Tensor cperm, "
          "Tensor? edge_attr, Tensor saved, Tensor grad_h_out, Tensor grad_x_out, int hidden_nf, int out_nf, int n_layers, int num_heads=1, "
          "int edge_basis=0) -> (Tensor, Tensor, Tensor)");
    m.def("weighted_kabsch_forward(Tensor src, Tensor tgt, Tensor weights) -> (Tensor, Tensor, Tensor)");
    m.def("weighted_kabsch_backward(Tensor src, Tensor tgt, Tensor weights, Tensor aux, Tensor grad_R, Tensor grad_t) -> (Tensor, Tensor, Tensor)");
    // composite ops (ATen glue around egnn_forward: autograd follows): one E_GCL layer, reference E_GCL.forward train :280-289
    m.def("egcl_forward(Tensor layer_params, Tensor h, Tensor x, Tensor row, Tensor col, Tensor? edge_attr, int num_heads=1, int edge_basis=0) "
          "-> (Tensor, Tensor)", &egcl_forward);
    m.def("egcl_backward(Tensor layer_params, Tensor h, Tensor x, Tensor row, Tensor col, Tensor? edge_attr, Tensor grad_h_out, Tensor grad_x_out, "
          "int num_heads=1, int edge_basis=0) -> (Tensor, Tensor, Tensor)", &egcl_backward);
    m.def("head_eval_forward(Tensor h_all, Tensor x_all, Tensor raw_h_src, Tensor raw_h_tgt, Tensor raw_x_src, Tensor raw_x_tgt, Tensor labels, "
          "Tensor gt_pose, Tensor mlp, int K) -> (Tensor, Tensor, Tensor, Tensor, Tensor)");
    m.def("pose_loss_forward(Tensor pred_rot, Tensor pred_translation, Tensor gt_pose) -> (Tensor, Tensor)");
    m.def("pose_loss_backward(Tensor pred_rot, Tensor pred_translation, Tensor gt_pose, Tensor grad_rot_loss, Tensor grad_trans_loss) -> (Tensor, Tensor)");
    // N4: typed irrep-basis message passing; differentiable in f0, f1, x and all six parameter tensors (HIP backward)
    m.def("irrep_conv_forward(Tensor f0, Tensor f1, Tensor x, Tensor row, Tensor col, Tensor w_r1, Tensor b_r1, Tensor w_r2, Tensor b_r2, "
          "Tensor w_self0, Tensor w_self1, float[] basis) -> (Tensor, Tensor, Tensor)");
    m.def("irrep_conv_backward(Tensor f0, Tensor f1, Tensor x, Tensor row, Tensor col, Tensor w_r1, Tensor b_r1, Tensor w_r2, Tensor b_r2, "
          "Tensor w_self0, Tensor w_self1, float[] basis, Tensor moments, Tensor g_out0, Tensor g_out1) -> Tensor[]");
}

TORCH_LIBRARY_IMPL(equigspr, Autograd, m) {
    m.impl("weighted_kabsch_forward", &weighted_kabsch_forward_autograd);
    m.impl("egnn_forward", &egnn_forward_autograd);
    m.impl("pose_loss_forward", &pose_loss_forward_autograd);
    m.impl("irrep_conv_forward", &irrep_conv_forward_autograd);
}

TORCH_LIBRARY_IMPL(equigspr, CUDA, m) {       // the HIP backend key of a ROCm build of PyTorch
    m.impl("knn_graph", &knn_graph);
    m.impl("build_inverse_index", &build_inverse_index);
    m.impl("egnn_forward", &egnn_forward);
    m.impl("egnn_backward", &egnn_backward);
    m.impl("weighted_kabsch_forward", &weighted_kabsch_forward);
    m.impl("weighted_kabsch_backward", &weighted_kabsch_backward);
    m.impl("head_eval_forward", &head_eval_forward);
    m.impl("pose_loss_forward", &pose_loss_forward);
    m.impl("pose_loss_backward", &pose_loss_backward);
    m.impl("irrep_conv_forward", &irrep_conv_forward);
    m.impl("irrep_conv_backward", &irrep_conv_backward);
}
