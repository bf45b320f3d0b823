"""Ad-hoc perf probe (not the bench contract): build N x D on device, time batched search."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import muninn_amd
pkg = muninn_amd.pkg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 768
order = pkg.ORDER_WAVE if (len(sys.argv) > 3 and sys.argv[3] == "wave") else pkg.ORDER_SSE
NQ = int(sys.argv[4]) if len(sys.argv) > 4 else 10000
maxb = int(sys.argv[5]) if len(sys.argv) > 5 else 8192
rng = np.random.default_rng(42)
X = rng.standard_normal((N, D), dtype=np.float32)
ids = np.arange(1, N + 1, dtype=np.int64)
Q = np.random.default_rng(43).standard_normal((NQ, D), dtype=np.float32)
g = pkg.HnswIndex(D, "cosine", 16, 200, order=order)
t = time.time()
# chunked build with progress
pos = 0
while pos < N:
    b = min(max(1, g.node_count // 16), maxb, N - pos)
    assert g.insert_batch(ids[pos:pos + b], X[pos:pos + b], pkg.BUILD_BATCHED) == 0
    pos += b
    if (pos // maxb) % 8 == 0 or pos == N:
        st = g.last_launch()
        print(f"  built {pos} in {time.time()-t:.1f}s  (last search kernel {st['last_kernel_ms']:.1f} ms, b={b}, n_dist/node={st['last_n_dist']/max(b,1):.0f})", flush=True)
tb = time.time() - t
print(f"build {N}x{D}: {tb:.1f}s = {N/tb:.0f} vec/s  max_level={g.max_level}", flush=True)
dq = g.dev_malloc(Q.nbytes); g.dev_upload(dq, Q)
k = 10
d_ids = g.dev_malloc(NQ * k * 8); d_ds = g.dev_malloc(NQ * k * 4); d_c = g.dev_malloc(NQ * 4)
for ef in (64, 128, 256):
    for rep in range(3):
        t = time.time()
        g.search_batch_dev(dq, NQ, k, ef, d_ids, d_ds, d_c); g.sync()
        dt = time.time() - t
    st = g.last_launch()
    ndist = st["last_n_dist"]; nexp = st["last_n_expanded"]
    byt = ndist * D * 4 + nexp * 32 * 4 + ndist * 4
    out = np.empty((NQ, k), np.int64); g.dev_download(out, d_ids)
    nbf = min(NQ, 500)
    truth = g.bruteforce_topk(dq, nbf, k)
    rec = np.mean([len(set(out[i]) & set(truth[i])) / k for i in range(nbf)])
    print(f"ef={ef}: wall {dt*1e3:.1f} ms kernel {st['last_kernel_ms']:.1f} ms -> {NQ/dt:.0f} q/s; n_dist/q={ndist/NQ:.0f} exp/q={nexp/NQ:.0f} "
          f"alg GB/s={byt/st['last_kernel_ms']/1e6:.0f} recall@10={rec:.3f} ovf={st['last_n_overflow']}", flush=True)
