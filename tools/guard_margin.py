#!/usr/bin/env python3
"""Tile-by-tile exactness and run-to-run reproducibility of the split-bf16 edge kernels per library variant (fenced / unfenced products,
with / without packed fp32 instructions, ISA-patched forms: profiles/r06_mfma_hazard_probe.txt), measured on the GPU.

For every library variant given (built here with `make -C csrc variant_edge NAME=.. DEFS=..`; the default library is always included)
the harness of profiles/r05_bf16x3_edge_kernels.txt part D runs in a CHILD process (EQUIGSPR_LIB selects the library): one EGNN layer,
forward + backward, on a ring graph whose node n owns the 16-edge tile n -- a wrong tile of the edge kernels shows at nodes n / n + 1
only -- at several workgroup caps (1 workgroup = 8 waves, two per SIMD, ~n/32 units per wave; 8; 64; uncapped), each `--runs` times.
Reference = the same library's fp32 products (equigspr_set_edge_products(1): v_mfma_f32_16x16x4_f32, an exact fmaf chain).
Reported per (variant, graph, cap): nodes whose gradient differs from the fp32-product run by > 2e-5 of the largest gradient (a lost
split level is ~1e-2), worst difference, and whether the runs are bit-identical to each other.

    python tools/guard_margin.py --out gpurun_out/r06_guard_margin.json [--runs 5] [name=path/to/libequigspr_name.so ...]
"""
import argparse, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import json, os, sys
sys.path.insert(0, %(root)r)
import torch
import se3_equi_graph_registration_amd as pk
from se3_equi_graph_registration_amd import _lib
L = _lib.lib()
runs = int(sys.argv[1])
def run(n, feat, fp32, cap):
    _lib.check(L.equigspr_set_edge_products(1 if fp32 else 0), "set_edge_products")
    _lib.check(L.equigspr_set_edge_grid_cap(cap), "set_edge_grid_cap")
    torch.manual_seed(2)
    m = pk.EGNN(feat, feat, feat, in_edge_nf=1, device="cuda", n_layers=1, num_heads=1)
    g = torch.Generator().manual_seed(5)
    h = torch.randn(n, feat, generator=g).cuda(); x = (torch.rand(n, 3, generator=g) * 2).cuda()
    row = torch.arange(n).repeat_interleave(16); col = (row + 1) %% n
    ea = torch.rand(row.numel(), 1, generator=g).cuda()
    gh = torch.randn(n, feat, generator=g).cuda(); gx = torch.randn(n, 3, generator=g).cuda()
    hc, xc = h.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ho, xo = m(hc, xc, [row.cuda(), col.cuda()], ea)
    m.zero_grad()
    ((ho * gh).sum() + (xo * gx).sum()).backward()
    torch.cuda.synchronize()
    return [t.detach().double().cpu() for t in (ho, xo, hc.grad, xc.grad)], m.flat_grads.detach().double().cpu()
out = []
for feat in [int(f) for f in sys.argv[2].split(',')]:
    for n in (3000, 40000):
        ref, refp = run(n, feat, True, 0)
        for cap in (1, 8, 64, 0):
            res = [run(n, feat, False, cap) for _ in range(runs)]
            bad, worst, same, per = 0, 0.0, True, [0, 0, 0, 0]
            for (ts, p) in res:
                for q, (a, b) in enumerate(zip(ts, ref)):
                    d = (a - b).abs().reshape(n, -1).max(dim=1).values / b.abs().max()
                    bad = max(bad, int((d > 2e-5).sum())); worst = max(worst, float(d.max())); per[q] = max(per[q], int((d > 2e-5).sum()))
                worst_p = float(((p - refp).abs().max() / refp.abs().max()))
                same = same and all(torch.equal(a, b) for a, b in zip(ts, res[0][0])) and torch.equal(p, res[0][1])
            out.append({"hidden_nf": feat, "nodes": n, "tiles": n, "grid_cap": cap, "runs": runs, "wrong_nodes_max": bad, "worst_rel_diff": worst,
                        "param_grad_rel_diff": worst_p, "bit_identical_runs": same,
                        "wrong_nodes_by_tensor": dict(zip(("h_out", "x_out", "grad_h", "grad_x"), per))})
print("RESULT " + json.dumps(out))
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="*", help="name=path pairs")
    ap.add_argument("--runs", type=int, default=5)
    ap.add_argument("--feats", default="32,33", help="hidden_nf values to run")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    libs = [("default", os.path.join(ROOT, "se3_equi_graph_registration_amd", "libequigspr.so"))] + [tuple(v.split("=", 1)) for v in a.variants]
    report = {}
    for name, path in libs:
        env = dict(os.environ, EQUIGSPR_LIB=os.path.abspath(path))
        p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}, str(a.runs), a.feats], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
        if p.returncode != 0 or not line:
            report[name] = {"error": p.stderr[-1500:]}
            print(name, "FAILED", p.stderr[-600:])
            continue
        report[name] = json.loads(line[-1][7:])
        for r in report[name]:
            print(f"{name:10s} F={r['hidden_nf']} tiles={r['tiles']:6d} cap={r['grid_cap']:3d}: wrong nodes {r['wrong_nodes_max']:5d}  worst {r['worst_rel_diff']:.2e}  "
                  f"params {r['param_grad_rel_diff']:.2e}  bit-identical runs: {r['bit_identical_runs']}  by tensor {r.get('wrong_nodes_by_tensor')}")
    if a.out:
        json.dump(report, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
