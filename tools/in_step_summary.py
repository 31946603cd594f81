#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats CSVs of bench.py runs (hipGraph replays of the product loop) -> profiles/rNN_in_step.json: the
average duration of the edge / node kernels INSIDE the step, which bench.py quotes as roofline.in_step_* next to its back-to-back
HIP-event figure.   python tools/in_step_summary.py <out_json> <tag> <csv>:<edges>:<feat>:<training 0|1>:<flags> ..."""
import csv, json, sys

out = {"source": f"rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-workloads --min-seconds 0 <flags>, MI355X, {sys.argv[2]}",
       "workloads": []}
for spec in sys.argv[3:]:
    path, edges, feat, training, flags = spec.split(":", 4)
    flags = flags.replace("+", " ")
    kernels = {}
    for r in csv.DictReader(open(path)):
        name = r["Name"].split("(")[0].replace("void ", "").replace("eq::", "").replace("ir::", "").strip()
        short = name.split("<")[0]
        if any(t in short for t in ("edge_", "node_", "gather_", "knn_", "head_", "csr_", "embed_", "layer_grad_reduce", "adam_", "irrep_")):
            e = kernels.setdefault(short, {"avg_us": 0.0, "calls": 0, "variants": {}})
            calls, avg = int(r["Calls"]), float(r["AverageNs"]) / 1e3
            e["avg_us"] = (e["avg_us"] * e["calls"] + avg * calls) / (e["calls"] + calls)
            e["calls"] += calls
            e["variants"][name] = {"avg_us": round(avg, 2), "calls": calls}
    for e in kernels.values():
        e["avg_us"] = round(e["avg_us"], 2)
    out["workloads"].append({"edges_per_launch": int(edges), "feat": int(feat), "training": bool(int(training)), "source": f"bench.py {flags}", "kernels": kernels})
json.dump(out, open(sys.argv[1], "w"), indent=1)
for w in out["workloads"]:
    print(w["source"], {k: v["avg_us"] for k, v in w["kernels"].items() if "edge_" in k or "node_" in k})
