"""Hand-derived backward of the EGNN in exactly the decomposition the HIP kernels use
(per-node projections P, per-edge recompute, node-level gathers through the row / col
inverse indices).  Pure torch fp64, vectorised; TEST INFRASTRUCTURE: it documents and
validates the formulas (tests/test_backward_derivation.py checks it against the golden
gradients of the real reference) before they are trusted on the GPU."""
import torch
import torch.nn.functional as F


def silu_grad(v):
    s = torch.sigmoid(v)
    return s * (1 + v * (1 - s))


def cross(a, b):
    return torch.cross(a, b, dim=1)


def geometry(xr, xc):
    d = xr - xc
    radial = (d * d).sum(1, keepdim=True)
    dist = radial.sqrt()
    dot = (xr * xc).sum(1, keepdim=True)
    inv_na = 1 / (dist + 1e-8)
    a = d * inv_na
    p = cross(xr, xc)
    pn = p.norm(dim=1, keepdim=True)
    inv_nb = 1 / (pn + 1e-8)
    b = p * inv_nb
    c = cross(a, b)
    deg = (a.norm(dim=1) < 1e-6) | (b.norm(dim=1) < 1e-6) | (c.norm(dim=1) < 1e-6)
    so3 = torch.stack([a, b, c], dim=2).reshape(-1, 9)
    eye = torch.eye(3, dtype=xr.dtype).reshape(1, 9).expand_as(so3)
    so3 = torch.where(deg[:, None], eye, so3)
    geo = torch.cat([radial, dist, dot, so3], dim=1)
    return dict(d=d, geo=geo, dist=dist, inv_na=inv_na, inv_nb=inv_nb, p=p, pn=pn, deg=deg)


def geometry_backward(xr, xc, g, gg, gd):
    """Mirror of eq::edge_geometry_backward (csrc/egnn_math.h)."""
    d, dist = g["d"], g["dist"]
    inv_dist = torch.where(dist > 0, 1 / dist.clamp_min(1e-300), torch.zeros_like(dist))
    gdt = gd + 2 * d * gg[:, 0:1] + d * inv_dist * gg[:, 1:2]
    gxr = xc * gg[:, 2:3]
    gxc = xr * gg[:, 2:3]
    a, b = d * g["inv_na"], g["p"] * g["inv_nb"]
    gs = gg[:, 3:].reshape(-1, 3, 3)
    ga, gb, gc = gs[:, :, 0], gs[:, :, 1], gs[:, :, 2]
    ga = ga + cross(b, gc)
    gb = gb + cross(gc, a)
    ka = (d * ga).sum(1, keepdim=True) * inv_dist * g["inv_na"] ** 2
    gd_f = ga * g["inv_na"] - d * ka
    inv_pn = torch.where(g["pn"] > 0, 1 / g["pn"].clamp_min(1e-300), torch.zeros_like(g["pn"]))
    kb = (g["p"] * gb).sum(1, keepdim=True) * inv_pn * g["inv_nb"] ** 2
    gp = gb * g["inv_nb"] - g["p"] * kb
    live = (~g["deg"])[:, None].to(xr.dtype)
    gdt = gdt + live * gd_f
    gxr = gxr + live * cross(xc, gp)
    gxc = gxc + live * cross(gp, xr)
    return gxr + gdt, gxc - gdt


def seg_sum(data, idx, n):
    out = data.new_zeros((n, data.shape[1]))
    out.index_add_(0, idx, data)
    return out


def egnn_forward_backward(w, h_in, x_in, row, col, attr, gh_out, gx_out):
    """Returns (h_out, x_out, grads dict keyed like the state_dict, grad_h_in, grad_x_in)."""
    L = 0
    while f"gcl_{L}.layer_norm.weight" in w:
        L += 1
    H = 0
    while f"gcl_0.edge_mlps.{H}.0.weight" in w:
        H += 1
    n = h_in.shape[0]
    Fd = w["embedding_in.weight"].shape[0]

    def stacked(l):
        pre = f"gcl_{l}."
        W1 = torch.cat([w[f"{pre}edge_mlps.{j}.0.weight"] for j in range(H)], 0)      # [F, 2F+12+A]
        b1 = torch.cat([w[f"{pre}edge_mlps.{j}.0.bias"] for j in range(H)], 0)
        W2 = torch.block_diag(*[w[f"{pre}edge_mlps.{j}.2.weight"] for j in range(H)])  # [F,F] block diagonal
        b2 = torch.cat([w[f"{pre}edge_mlps.{j}.2.bias"] for j in range(H)], 0)
        return W1, b1, W2, b2

    # ---------------- forward, saving what the kernels save: h_l, x_l, agg_l
    hs, xs, aggs = [F.linear(h_in, w["embedding_in.weight"], w["embedding_in.bias"])], [x_in], []
    for l in range(L):
        pre = f"gcl_{l}."
        W1, b1, W2, b2 = stacked(l)
        h, x = hs[l], xs[l]
        Pr, Pc = h @ W1[:, :Fd].T, h @ W1[:, Fd:2 * Fd].T + b1
        g = geometry(x[row], x[col])
        t1 = Pr[row] + Pc[col] + g["geo"] @ W1[:, 2 * Fd:2 * Fd + 12].T
        if attr is not None:
            t1 = t1 + attr * W1[:, 2 * Fd + 12][None, :]
        u = F.silu(t1) @ W2.T + b2
        m = F.layer_norm(u, (Fd,), w[pre + "layer_norm.weight"], w[pre + "layer_norm.bias"], 1e-5)
        s = F.silu(m @ w[pre + "coord_mlp.0.weight"].T + w[pre + "coord_mlp.0.bias"]) @ w[pre + "coord_mlp.2.weight"].T
        agg = seg_sum(m, row, n)
        aggs.append(agg)
        xs.append(x + seg_sum(g["d"] * s, row, n))
        v = torch.cat([h, agg], 1) @ w[pre + "node_mlp.0.weight"].T + w[pre + "node_mlp.0.bias"]
        hs.append(h + F.silu(v) @ w[pre + "node_mlp.2.weight"].T + w[pre + "node_mlp.2.bias"])
    h_out = F.linear(hs[L], w["embedding_out.weight"], w["embedding_out.bias"])
    x_out = xs[L]

    # ---------------- backward
    G = {}
    G["embedding_out.weight"] = gh_out.T @ hs[L]
    G["embedding_out.bias"] = gh_out.sum(0)
    Gh = gh_out @ w["embedding_out.weight"]          # dL/dh_L
    Gx = gx_out.clone()                              # dL/dx_L
    for l in reversed(range(L)):
        pre = f"gcl_{l}."
        W1, b1, W2, b2 = stacked(l)
        W4, W5, W6 = w[pre + "node_mlp.0.weight"], w[pre + "node_mlp.2.weight"], w[pre + "coord_mlp.0.weight"]
        w7 = w[pre + "coord_mlp.2.weight"]
        gam = w[pre + "layer_norm.weight"]
        h, x, agg = hs[l], xs[l], aggs[l]
        # (a) node backward
        vin = torch.cat([h, agg], 1)
        v = vin @ W4.T + w[pre + "node_mlp.0.bias"]
        gact = F.silu(v)
        G[pre + "node_mlp.2.weight"] = Gh.T @ gact
        G[pre + "node_mlp.2.bias"] = Gh.sum(0)
        dv = (Gh @ W5) * silu_grad(v)
        G[pre + "node_mlp.0.weight"] = dv.T @ vin
        G[pre + "node_mlp.0.bias"] = dv.sum(0)
        dvin = dv @ W4
        dh = Gh + dvin[:, :Fd]
        dagg = dvin[:, Fd:]
        # (b) edge backward (recompute)
        Pr, Pc = h @ W1[:, :Fd].T, h @ W1[:, Fd:2 * Fd].T + b1
        g = geometry(x[row], x[col])
        Wg = W1[:, 2 * Fd:2 * Fd + 12]
        t1 = Pr[row] + Pc[col] + g["geo"] @ Wg.T
        if attr is not None:
            t1 = t1 + attr * W1[:, 2 * Fd + 12][None, :]
        g1 = F.silu(t1)
        u = g1 @ W2.T + b2
        mean = u.mean(1, keepdim=True)
        rstd = 1 / torch.sqrt(((u - mean) ** 2).mean(1, keepdim=True) + 1e-5)
        xhat = (u - mean) * rstd
        m = xhat * gam + w[pre + "layer_norm.bias"]
        q = m @ W6.T + w[pre + "coord_mlp.0.bias"]
        p = F.silu(q)
        s = p @ w7.T
        Gxr = Gx[row]
        ds = (g["d"] * Gxr).sum(1, keepdim=True)
        dd = s * Gxr
        G[pre + "coord_mlp.2.weight"] = (p * ds).sum(0, keepdim=True)
        dq = (ds * w7) * silu_grad(q)
        G[pre + "coord_mlp.0.weight"] = dq.T @ m
        G[pre + "coord_mlp.0.bias"] = dq.sum(0)
        dm = dagg[row] + dq @ W6
        G[pre + "layer_norm.weight"] = (dm * xhat).sum(0)
        G[pre + "layer_norm.bias"] = dm.sum(0)
        dxh = dm * gam
        du = rstd * (dxh - dxh.mean(1, keepdim=True) - xhat * (dxh * xhat).mean(1, keepdim=True))
        dW2 = du.T @ g1
        db2 = du.sum(0)
        dt1 = (du @ W2) * silu_grad(t1)
        dWg = dt1.T @ g["geo"]
        dgeo = dt1 @ Wg
        gxr_e, gxc_e = geometry_backward(x[row], x[col], g, dgeo, dd)
        # (c) node-level gathers
        n_e = row.shape[0]
        dPr = seg_sum(dt1, row, n)
        dPc = seg_sum(dt1, col, n)
        dW1 = torch.zeros_like(W1)
        dW1[:, :Fd] = dPr.T @ h
        dW1[:, Fd:2 * Fd] = dPc.T @ h
        dW1[:, 2 * Fd:2 * Fd + 12] = dWg
        if attr is not None:
            dW1[:, 2 * Fd + 12] = (dt1 * attr).sum(0)
        db1 = dPc.sum(0)
        FH = Fd // H
        for j in range(H):
            sl = slice(j * FH, (j + 1) * FH)
            G[f"{pre}edge_mlps.{j}.0.weight"] = dW1[sl]
            G[f"{pre}edge_mlps.{j}.0.bias"] = db1[sl]
            G[f"{pre}edge_mlps.{j}.2.weight"] = dW2[sl, sl]
            G[f"{pre}edge_mlps.{j}.2.bias"] = db2[sl]
        Gh = dh + dPr @ W1[:, :Fd] + dPc @ W1[:, Fd:2 * Fd]
        Gx = Gx + seg_sum(gxr_e, row, n) + seg_sum(gxc_e, col, n)
    G["embedding_in.weight"] = Gh.T @ h_in
    G["embedding_in.bias"] = Gh.sum(0)
    grad_h_in = Gh @ w["embedding_in.weight"]
    return h_out, x_out, G, grad_h_in, Gx
