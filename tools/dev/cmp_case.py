"""developer: max relative difference of every array of two tools/dev/dump_case.py dumps:  python tools/dev/cmp_case.py a.npz b.npz"""
import sys
import numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
grp = {}
for k in a.files:
    d = float(np.abs(a[k].astype(np.float64) - b[k]).max() / max(1e-30, np.abs(a[k]).max()))
    grp[k if not k.startswith("g.") else "params"] = max(grp.get(k if not k.startswith("g.") else "params", 0.0), d)
print("  ".join(f"{k} {v:.1e}" for k, v in grp.items()))
