"""LFR-like synthetic benchmark graphs: power-law degrees (tau1), power-law community
sizes (tau2), mixing parameter mu; stubs are matched configuration-model style inside communities and
across them.  numpy only (networkx is not available); used by bench/probe scripts and tests."""
import numpy as np


def _powerlaw_ints(rng, n, lo, hi, tau):
    u = rng.random(n)
    a = 1.0 - tau
    x = ((hi ** a - lo ** a) * u + lo ** a) ** (1.0 / a)
    return np.clip(np.floor(x).astype(np.int64), lo, hi)


def _match(rng, stubs):
    """random perfect matching of a stub array → (src, dst) without self loops"""
    p = rng.permutation(stubs)
    if len(p) % 2:
        p = p[:-1]
    s, d = p[0::2], p[1::2]
    keep = s != d
    return s[keep], d[keep]


def lfr_like(n, avg_deg=40, max_deg=200, mu=0.3, tau1=2.5, tau2=1.5, min_comm=50, max_comm=1000, seed=42):
    rng = np.random.default_rng(seed)
    # degrees: power law on [kmin, max_deg] with kmin tuned to hit avg_deg
    kmin = max(2, int(avg_deg / 2))
    for _ in range(40):
        deg = _powerlaw_ints(rng, n, kmin, max_deg, tau1)
        if deg.mean() < avg_deg * 0.97:
            kmin += 1
        elif deg.mean() > avg_deg * 1.03 and kmin > 2:
            kmin -= 1
        else:
            break
    # community sizes
    sizes = []
    tot = 0
    while tot < n:
        s = int(_powerlaw_ints(rng, 1, min_comm, max_comm, tau2)[0])
        s = min(s, n - tot) if n - tot >= min_comm else n - tot
        sizes.append(s)
        tot += s
    comm = np.repeat(np.arange(len(sizes)), sizes)[:n]
    comm = comm[rng.permutation(n)]
    k_in = np.round(deg * (1.0 - mu)).astype(np.int64)
    k_out = deg - k_in
    # internal edges per community
    order = np.argsort(comm, kind="stable")
    bounds = np.searchsorted(comm[order], np.arange(len(sizes) + 1))
    ss, dd = [], []
    for c in range(len(sizes)):
        nodes = order[bounds[c]:bounds[c + 1]]
        stubs = np.repeat(nodes, np.minimum(k_in[nodes], len(nodes) - 1))
        s, d = _match(rng, stubs)
        ss.append(s)
        dd.append(d)
    s, d = _match(rng, np.repeat(np.arange(n), k_out))
    keep = comm[s] != comm[d]
    ss.append(s[keep])
    dd.append(d[keep])
    s = np.concatenate(ss)
    d = np.concatenate(dd)
    # drop multi-edges
    lo, hi = np.minimum(s, d), np.maximum(s, d)
    key = lo.astype(np.int64) * n + hi
    _, first = np.unique(key, return_index=True)
    first = np.sort(first)
    p = rng.permutation(len(first))
    return s[first][p].astype(np.int32), d[first][p].astype(np.int32), comm.astype(np.int32)
