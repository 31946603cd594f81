#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter-collection CSVs (two separate passes) -> profiles/rNN_pmc_<workload>.json.
   python tools/pmc_summary.py <fetch_csv> <write_csv> <out_json> <edges_per_launch> <feat> <tag> [workload text]
Per kernel: FETCH_SIZE / WRITE_SIZE in KB per launch (averaged over the launches of the run) and the HBM traffic in bytes per launch,
2 * FETCH + WRITE (MI355X_MICROARCH.md, HBM section: FETCH_SIZE counts half of the wide reads on gfx950).  `edge_fwd_kernel` /
`edge_bwd_kernel` are the entries bench.py quotes as roofline.traffic; every node-level kernel is listed under its own name."""
import csv, json, sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "").strip()].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
edges, feat, tag = int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
what = sys.argv[7] if len(sys.argv) > 7 else ""
out = {
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes), bench.py --steps 5 --warmup 2 --no-graph "
              f"--no-cpu-baseline --no-workloads --min-seconds 0 {what}, MI355X, {tag}",
    "workload": {"edges_per_launch": edges, "feat": feat, "flags": what},
    "units": "KB per launch (rocprofv3 FETCH_SIZE/WRITE_SIZE), averaged over the launches of the run",
    "gfx950_correction": "FETCH_SIZE reports 1/2 of wide (16 B/lane) reads (MI355X_MICROARCH.md HBM section): traffic = 2*FETCH + WRITE",
}


def entry(name):
    f, nf = fetch[name]
    w, _ = write.get(name, (0.0, 0))
    return {"kernel": name, "launches": nf, "FETCH_SIZE_KB": round(f, 1), "WRITE_SIZE_KB": round(w, 1), "traffic_bytes": int((2 * f + w) * 1024)}


for short, pats in (("edge_fwd_kernel", ("eq::edge_fwd16_kernel", "eq::edge_fwd_fp_kernel")),
                    ("edge_bwd_kernel", ("eq::edge_bwd16_kernel", "eq::edge_bwd_fp_kernel"))):
    for pat in pats:
        names = [k for k in fetch if k.startswith(pat)]
        if names:
            out[short] = entry(names[0])
            break
for name in sorted(fetch):
    if any(t in name for t in ("node_", "gather_", "embed_", "knn_", "csr_", "head_", "layer_grad_reduce", "adam_", "irrep_")):
        out[name.replace("eq::", "")] = entry(name)
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if isinstance(v, dict) and "traffic_bytes" in v}, indent=1)[:3000])
