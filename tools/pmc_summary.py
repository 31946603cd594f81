#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter-collection CSVs (two separate passes) -> profiles/rNN_pmc_edge_kernels.json.
   python tools/pmc_summary.py <fetch_csv> <write_csv> <out_json> [edges_per_launch] [feat] [round-tag]"""
import csv, json, sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
edges = int(sys.argv[4]) if len(sys.argv) > 4 else 81920
feat = int(sys.argv[5]) if len(sys.argv) > 5 else 32
tag = sys.argv[6] if len(sys.argv) > 6 else "round 2"
out = {
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes), bench.py --steps 5 --warmup 2 --no-graph "
              f"--no-cpu-baseline --min-seconds 0, MI355X, {tag}",
    "workload": {"pairs_per_gpu": 1, "points": 2048, "k": 20, "feat": feat, "edges_per_launch": edges},
    "units": "KB per launch (rocprofv3 FETCH_SIZE/WRITE_SIZE), averaged over the launches of the run",
    "gfx950_correction": "FETCH_SIZE reports 1/2 of wide (16 B/lane) reads (MI355X_MICROARCH.md HBM section): traffic = 2*FETCH + WRITE",
}
# short name -> kernel-name prefixes (first match wins): the hidden_nf = 32 kernels are the 16-edge formulation
for short, pats in (("edge_fwd_kernel", ("eq::edge_fwd16_kernel", "eq::edge_fwd_mfma_kernel")),
                    ("edge_bwd_kernel", ("eq::edge_bwd16_kernel", "eq::edge_bwd_mfma_kernel")),
                    ("node_fwd_kernel", ("eq::node_fwd_kernel",)), ("gather_node_bwd_kernel", ("eq::gather_node_bwd_kernel",)),
                    ("knn_kernel", ("eq::knn_kernel",))):
    for pat in pats:
        f = [v for k, v in fetch.items() if k.startswith(pat)]
        w = [v for k, v in write.items() if k.startswith(pat)]
        if f and w:
            out[short] = {"kernel": pat, "FETCH_SIZE_KB": round(f[0], 1), "WRITE_SIZE_KB": round(w[0], 1),
                          "traffic_bytes": int((2 * f[0] + w[0]) * 1024)}
            break
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
