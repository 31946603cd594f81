import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
for k in ("grad_h", "grad_x"):
    d = np.abs(a[k].astype(np.float64) - b[k]).max(axis=1) / np.abs(a[k]).max()
    bad = np.nonzero(d > 2e-5)[0]
    print(k, "max", d.max(), "bad nodes", len(bad), "of", len(d), ":", bad[:80].tolist())
k = "grad_h"
d = np.abs(a[k].astype(np.float64) - b[k]).max(axis=1) / np.abs(a[k]).max()
bad = np.nonzero(d > 2e-5)[0]
np.set_printoptions(precision=4, linewidth=200, suppress=True)
for n in bad[:4]:
    print("node", n, "ref ", a[k][n][:16]); print("          got ", b[k][n][:16]); print("          diff", (b[k][n] - a[k][n])[:32])
