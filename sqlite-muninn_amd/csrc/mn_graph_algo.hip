// mn_graph_algo.hip — the reference's remaining edge-list algorithms on the device, gfx950:
//   mn_graph_pagerank     run_pagerank  (src/graph_tvf.c:1631-1797)
//   mn_graph_components   run_components (src/graph_tvf.c:1314-1366, union-find :1231-1273)
//
// PageRank.  The reference PUSHES: for i = 0..N-1 in order, rank_new[target] += share_i for every out-edge of i, and a
// node without out-edges adds its share to EVERY node.  f64 addition is not associative, so the value of rank_new[j] is
// defined by that order: teleport, then the shares of j's in-neighbours and of all dangling nodes interleaved by
// ascending source index (edges of one source carry the same share, so their mutual order is immaterial).  The device
// PULLS the same sequence: one thread owns a target j and walks its in-list (sources ascending, one entry per edge —
// multi-edges count as often as in the reference) merged with the dangling list, which a workgroup stages through LDS
// 256 entries at a time (every thread consumes the same dangling entries in the same order).  Bit-identical ranks.
// HBM traffic per iteration: E·(4 + 8) B of in-list gathers + N·16 B; the N·D dangling term lives in LDS.
//
// Components.  The reference's component_id is the union-find root, which depends on the order of the unions (union by
// rank, path halving): MN_COMPONENTS_EXACT replays that sequence (one lane, latency-bound — the mode the SQL surface
// uses for inputs of the reference's own test sizes); MN_COMPONENTS_FAST hooks roots in parallel (atomicMin) until
// nothing changes and compresses: the same partition and sizes, component_id = smallest node index of the component.
#include "../../include/muninn_hip.h"
#include "mn_guard.hpp"
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

static thread_local std::string g_aerr;
static void aset_err(const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_aerr = buf;
}
extern "C" const char *mn_graph_algo_last_error(void) { return g_aerr.c_str(); }

#define ACHK(expr)                                                                                 \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess) {                                                                   \
            aset_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return -1;                                                                             \
        }                                                                                          \
    } while (0)

namespace {
struct Bufs { // frees on scope exit
    std::vector<void *> p;
    template <typename T> T *alloc(size_t n) {
        void *q = nullptr;
        if (hipMalloc(&q, (n ? n : 1) * sizeof(T)) != hipSuccess)
            return nullptr;
        p.push_back(q);
        return static_cast<T *>(q);
    }
    ~Bufs() {
        for (void *q : p)
            (void)hipFree(q);
    }
};
} // namespace

// ───────────────────────── PageRank ─────────────────────────

// share_i (:1692-1704): damping * rank / N for a dangling node, damping * rank / out_count otherwise
__global__ void k_pr_share(const double *rank, const int *outc, int n, double damping, double *share) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const int oc = outc[i];
    share[i] = damping * rank[i] / (double)(oc == 0 ? n : oc);
}

#define PR_CHUNK 256
#define DEVI_PR __device__ __forceinline__
__global__ void __launch_bounds__(PR_CHUNK)
    k_pr_pull(const int *in_off, const int *in_src, const double *share, const int *dang, int n_dang, int n, double teleport,
              double *rank_new) {
    __shared__ int d_idx[PR_CHUNK];
    __shared__ double d_sh[PR_CHUNK];
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    int p = 0, pe = 0;
    if (j < n) {
        p = in_off[j];
        pe = in_off[j + 1];
    }
    double acc = teleport; // :1689
    int next_src = p < pe ? in_src[p] : 0x7fffffff;
    for (int base = 0; base < n_dang; base += PR_CHUNK) {
        __syncthreads();
        if (base + (int)threadIdx.x < n_dang) {
            const int di = dang[base + threadIdx.x];
            d_idx[threadIdx.x] = di;
            d_sh[threadIdx.x] = share[di];
        }
        __syncthreads();
        const int cnt = n_dang - base < PR_CHUNK ? n_dang - base : PR_CHUNK;
        for (int q = 0; q < cnt; q++) {
            const int di = d_idx[q];
            while (next_src < di) { // in-neighbours with a smaller index come first
                acc += share[next_src];
                p++;
                next_src = p < pe ? in_src[p] : 0x7fffffff;
            }
            acc += d_sh[q];
        }
    }
    // what is left of the in-list (all of it when the graph has no dangling node): the additions stay in list order, but eight
    // sources and then their eight shares are requested together — the one-by-one walk was two dependent round trips per edge
    // (round 3: 7.7 % of the HBM roofline at 1M nodes / 20M rows)
    for (; p + 8 <= pe; p += 8) {
        int sidx[8];
        double sv[8];
#pragma unroll
        for (int u = 0; u < 8; u++)
            sidx[u] = in_src[p + u];
#pragma unroll
        for (int u = 0; u < 8; u++)
            sv[u] = share[sidx[u]];
#pragma unroll
        for (int u = 0; u < 8; u++)
            acc += sv[u];
    }
    while (p < pe) {
        acc += share[in_src[p]];
        p++;
    }
    if (j < n)
        rank_new[j] = acc;
}

// Round 4, graphs without dangling nodes whose share[] outgrows one XCD's L2 (4 MB): the in-list of a target is ascending by
// source, so cutting the SOURCES into ranges of 2^PR_TILE_LOG2 nodes cuts every list into consecutive pieces — one launch per range adds
// a target's piece to its running sum, in list order, and parks the sum in rank_new[] for the next range (an f64 through memory
// keeps its bits).  Every gather of a launch then falls into the same 2 MB of share[], which stays in each XCD's L2 instead of
// coming over the fabric a 128-byte line per 8-byte gather.
#define PR_TILE_LOG2 18
template <bool NT> DEVI_PR int pr_ld(const int *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT>
__global__ void __launch_bounds__(PR_CHUNK)
    k_pr_pull_tile(const int *lo, const int *hi, const int *in_src, const double *share, int n, double teleport, int first,
                   double *rank_new) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n)
        return;
    // NT (MN_PR_NT=1, off): the streamed operands loaded non-temporal so that they leave share[] in the L2 — measured 40 % SLOWER
    // (profiles/r04_ab_pagerank_source_ranges.txt); smaller ranges lose too: every launch re-reads the cuts and the running sums
    int p = pr_ld<NT>(lo + j);
    const int pe = pr_ld<NT>(hi + j);
    double acc = first ? teleport : rank_new[j]; // :1689
    for (; p + 4 <= pe; p += 4) {
        int sidx[4];
        double sv[4];
#pragma unroll
        for (int u = 0; u < 4; u++)
            sidx[u] = pr_ld<NT>(in_src + p + u);
#pragma unroll
        for (int u = 0; u < 4; u++)
            sv[u] = share[sidx[u]];
#pragma unroll
        for (int u = 0; u < 4; u++)
            acc += sv[u];
    }
    if (p < pe) { // up to three left: requested together, added in order
        const int s0 = pr_ld<NT>(in_src + p), s1 = p + 1 < pe ? pr_ld<NT>(in_src + p + 1) : s0, s2 = p + 2 < pe ? pr_ld<NT>(in_src + p + 2) : s0;
        const double v0 = share[s0], v1 = share[s1], v2 = share[s2];
        acc += v0;
        if (p + 1 < pe)
            acc += v1;
        if (p + 2 < pe)
            acc += v2;
    }
    rank_new[j] = acc;
}

// Balanced gathers (round 4, last): with one lane per target a wavefront issues as many gather instructions as its LONGEST piece has
// entries (≈ 12 where the mean is 5), and the kernel's time follows the gather instructions issued.  The pieces of a range are
// therefore stored range-major (tsrc / toff, built with the in-lists on the host), so that the pieces of a wavefront's 64 targets are
// one contiguous run: its lanes request entry base + k·64 + lane — coalesced sources, every lane busy — park the shares in LDS, and
// each lane then adds ITS piece from LDS in list order.  The additions and their order are unchanged.
#define PR_FLAT_CAP 512
__global__ void __launch_bounds__(PR_CHUNK)
    k_pr_pull_flat(const int *toff, const int *tsrc, const double *share, int n, double teleport, int first, double *rank_new) {
    __shared__ double val[PR_CHUNK / 64][PR_FLAT_CAP];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j0 = blockIdx.x * blockDim.x + (wv << 6); // first target of this wavefront
    if (j0 >= n)
        return; // (the whole wavefront)
    const int j = j0 + lane;
    const bool live = j < n;
    int p = toff[live ? j : n];
    const int pe = toff[live ? j + 1 : n];
    double acc = live ? (first ? teleport : rank_new[j]) : 0.0; // :1689
    const int wbeg = __shfl(p, 0), wend = toff[j0 + 64 < n ? j0 + 64 : n];
    double *v = val[wv];
    for (int base = wbeg; base < wend; base += PR_FLAT_CAP) {
        const int cnt = wend - base < PR_FLAT_CAP ? wend - base : PR_FLAT_CAP;
        for (int k0 = 0; k0 < cnt; k0 += 256) {
            int sidx[4];
            double sv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = k0 + u * 64 + lane;
                sidx[u] = i < cnt ? tsrc[base + i] : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
                sv[u] = sidx[u] >= 0 ? share[sidx[u]] : 0.0;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = k0 + u * 64 + lane;
                if (i < cnt)
                    v[i] = sv[u];
            }
        }
        __builtin_amdgcn_wave_barrier();
        const int hi = pe < base + cnt ? pe : base + cnt;
        for (; p < hi; p++)
            acc += v[p - base];
        __builtin_amdgcn_wave_barrier();
    }
    if (live)
        rank_new[j] = acc;
}

extern "C" int mn_graph_pagerank(int n, int64_t n_edges, const int *src, const int *dst, double damping, int iterations,
                                 int device, double *rank_out, mn_graph_algo_stats *stats) try {
    if (stats)
        memset(stats, 0, sizeof(*stats));
    if (n < 0 || n_edges < 0 || (n_edges && (!src || !dst)) || n_edges > 0x7fffffffLL) {
        aset_err("mn_graph_pagerank: bad arguments");
        return -1;
    }
    if (n == 0)
        return 0; // :1671-1674
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        aset_err("mn_graph_pagerank: HIP device %d not available (no CPU fallback)", device);
        return -1;
    }
    ACHK(hipSetDevice(device));
    // host preparation of the device input (as the reference builds its adjacency lists on the host): out-degrees, the
    // dangling list, and the in-lists with sources ascending — a stable counting sort of the edges by source, then by target
    const int E = (int)n_edges;
    std::vector<int> outc((size_t)n, 0), in_off((size_t)n + 1, 0), in_src((size_t)(E ? E : 1));
    for (int e = 0; e < E; e++) {
        if (src[e] < 0 || src[e] >= n || dst[e] < 0 || dst[e] >= n) {
            aset_err("mn_graph_pagerank: edge %d names a node outside [0, %d)", e, n);
            return -1;
        }
        outc[src[e]]++;
        in_off[dst[e] + 1]++;
    }
    for (int i = 0; i < n; i++)
        in_off[i + 1] += in_off[i];
    {
        std::vector<int> out_off((size_t)n + 1, 0), by_src((size_t)(E ? E : 1)), cur((size_t)n);
        for (int i = 0; i < n; i++)
            out_off[i + 1] = out_off[i] + outc[i];
        std::copy(out_off.begin(), out_off.begin() + n, cur.begin());
        for (int e = 0; e < E; e++)
            by_src[cur[src[e]]++] = dst[e];
        std::copy(in_off.begin(), in_off.begin() + n, cur.begin());
        for (int i = 0; i < n; i++)
            for (int x = out_off[i]; x < out_off[i + 1]; x++)
                in_src[cur[by_src[x]]++] = i;
    }
    std::vector<int> dang;
    for (int i = 0; i < n; i++)
        if (outc[i] == 0)
            dang.push_back(i);
    // source ranges (see k_pr_pull_tile): cut[t][j] = first position of j's list whose source is >= t * PR_TILE
    int tile_log2 = PR_TILE_LOG2, nt = 0;
    if (const char *e = getenv("MN_PR_TILE_LOG2"))
        tile_log2 = std::max(10, std::min(30, atoi(e)));
    if (const char *e = getenv("MN_PR_NT"))
        nt = atoi(e) != 0;
    const long long PR_TILE = 1LL << tile_log2;
    int tiles = dang.empty() && n > PR_TILE ? (int)((n + PR_TILE - 1) / PR_TILE) : 1;
    if (const char *e = getenv("MN_PR_TILES"))
        if (atoi(e) == 0)
            tiles = 1;
    std::vector<int> cut;
    if (tiles > 1) {
        cut.resize((size_t)(tiles + 1) * n);
        for (int j = 0; j < n; j++) {
            int p = in_off[j];
            for (int t = 0; t <= tiles; t++) {
                const long long bound = (long long)t * PR_TILE;
                while (p < in_off[j + 1] && in_src[p] < bound)
                    p++;
                cut[(size_t)t * n + j] = t == tiles ? in_off[j + 1] : p;
            }
        }
    }
    int flat = 1;
    if (const char *e = getenv("MN_PR_FLAT"))
        flat = atoi(e) != 0;
    std::vector<int> toff;
    if (tiles > 1 && flat) { // range-major pieces (k_pr_pull_flat); in_src is then only needed in this order
        toff.resize((size_t)tiles * ((size_t)n + 1));
        std::vector<int> tsrc((size_t)(E ? E : 1));
        int run = 0;
        for (int t = 0; t < tiles; t++) {
            for (int j = 0; j < n; j++) {
                toff[(size_t)t * (n + 1) + j] = run;
                for (int x = cut[(size_t)t * n + j]; x < cut[(size_t)(t + 1) * n + j]; x++)
                    tsrc[run++] = in_src[x];
            }
            toff[(size_t)t * (n + 1) + n] = run;
        }
        in_src.swap(tsrc);
        cut.swap(toff); // (uploaded through d_cut below)
        toff.assign(1, 0);
    }
    const bool use_flat = tiles > 1 && flat;
    Bufs b;
    int *d_cut = tiles > 1 ? b.alloc<int>(cut.size()) : nullptr;
    if (tiles > 1 && !d_cut) {
        aset_err("mn_graph_pagerank: out of device memory");
        return -1;
    }
    if (tiles > 1)
        ACHK(hipMemcpy(d_cut, cut.data(), cut.size() * sizeof(int), hipMemcpyHostToDevice));
    int *d_outc = b.alloc<int>(n), *d_inoff = b.alloc<int>((size_t)n + 1), *d_insrc = b.alloc<int>(E), *d_dang = b.alloc<int>(dang.size());
    double *d_r0 = b.alloc<double>(n), *d_r1 = b.alloc<double>(n), *d_share = b.alloc<double>(n);
    if (!d_outc || !d_inoff || !d_insrc || !d_dang || !d_r0 || !d_r1 || !d_share) {
        aset_err("mn_graph_pagerank: out of device memory");
        return -1;
    }
    ACHK(hipMemcpy(d_outc, outc.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice));
    ACHK(hipMemcpy(d_inoff, in_off.data(), ((size_t)n + 1) * sizeof(int), hipMemcpyHostToDevice));
    if (E)
        ACHK(hipMemcpy(d_insrc, in_src.data(), (size_t)E * sizeof(int), hipMemcpyHostToDevice));
    if (!dang.empty())
        ACHK(hipMemcpy(d_dang, dang.data(), dang.size() * sizeof(int), hipMemcpyHostToDevice));
    std::vector<double> init((size_t)n, 1.0 / n); // :1684-1686
    ACHK(hipMemcpy(d_r0, init.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    ACHK(hipEventCreate(&e0));
    ACHK(hipEventCreate(&e1));
    const double teleport = (1.0 - damping) / n; // :1689
    const int nb = (n + PR_CHUNK - 1) / PR_CHUNK;
    ACHK(hipEventRecord(e0, nullptr));
    double *cur = d_r0, *nxt = d_r1;
    for (int it = 0; it < iterations; it++) {
        hipLaunchKernelGGL(k_pr_share, dim3(nb), dim3(PR_CHUNK), 0, nullptr, cur, d_outc, n, damping, d_share);
        if (use_flat)
            for (int t = 0; t < tiles; t++)
                hipLaunchKernelGGL(k_pr_pull_flat, dim3(nb), dim3(PR_CHUNK), 0, nullptr, d_cut + (size_t)t * (n + 1), d_insrc, d_share, n, teleport,
                                   t == 0, nxt);
        else if (tiles > 1)
            for (int t = 0; t < tiles; t++)
                hipLaunchKernelGGL(nt ? k_pr_pull_tile<true> : k_pr_pull_tile<false>, dim3(nb), dim3(PR_CHUNK), 0, nullptr, d_cut + (size_t)t * n,
                                   d_cut + (size_t)(t + 1) * n, d_insrc, d_share, n, teleport, t == 0, nxt);
        else
            hipLaunchKernelGGL(k_pr_pull, dim3(nb), dim3(PR_CHUNK), 0, nullptr, d_inoff, d_insrc, d_share, d_dang, (int)dang.size(), n,
                               teleport, nxt);
        std::swap(cur, nxt);
    }
    ACHK(hipEventRecord(e1, nullptr));
    ACHK(hipGetLastError());
    ACHK(hipMemcpy(rank_out, cur, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (stats) {
        stats->device_ms = ms;
        stats->iterations = iterations;
        stats->aux = (int64_t)dang.size();
    }
    return 0;
} MN_GUARD_END(aset_err, MN_NOTHING, -1)

// ───────────────────────── connected components ─────────────────────────

// uf_find (:1249-1256) / uf_union (:1258-1273), verbatim, one lane: the root ids depend on this very sequence
__global__ void k_uf_seq(const int *src, const int *dst, long long n_edges, int *parent, int *rnk) {
    if (threadIdx.x != 0 || blockIdx.x != 0)
        return;
    for (long long e = 0; e < n_edges; e++) {
        int a = src[e], bq = dst[e];
        while (parent[a] != a) { // path halving
            parent[a] = parent[parent[a]];
            a = parent[a];
        }
        while (parent[bq] != bq) {
            parent[bq] = parent[parent[bq]];
            bq = parent[bq];
        }
        if (a == bq)
            continue;
        if (rnk[a] < rnk[bq]) { // union by rank
            const int t = a;
            a = bq;
            bq = t;
        }
        parent[bq] = a;
        if (rnk[a] == rnk[bq])
            rnk[a]++;
    }
}

__global__ void k_uf_init(int *parent, int *rnk, int *size, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        parent[i] = i;
        rnk[i] = 0;
        size[i] = 0;
    }
}

// EXACT without walking every edge through one lane (round 4: 158 ms for 10 000 nodes / 199 000 rows, 4.6x a 100-iteration
// PageRank of the same graph).  The reference's component_id is the union-find ROOT, and roots and ranks change only at an
// EFFECTIVE union — an edge whose ends are in different trees when its turn comes; every other edge only halves paths, which
// moves no root.  Trees only ever merge, so an edge whose ends share a root at the start of a block of edges is ineffective
// whatever happens inside the block: one workgroup tests 1 024 edges at once with read-only walks to the root and only the
// survivors — a few per cent of the rows of a connected graph — are replayed in row order, verbatim (find with path halving,
// union by rank, :1249-1273), by one wavefront.  Small graphs keep parent[] and rank[] in LDS (LDSV).
#define UF_BLOCK 1024
template <bool LDSV>
__global__ void __launch_bounds__(UF_BLOCK) k_uf_blocks(const int *src, const int *dst, long long n_edges, int *parent_g, int *rnk_g, int n) {
    extern __shared__ __align__(16) int uf_sm[];
    int *ea = uf_sm, *eb = uf_sm + UF_BLOCK;                                   // the block's edges
    unsigned long long *flag = reinterpret_cast<unsigned long long *>(uf_sm + 2 * UF_BLOCK); // one ballot per wavefront
    int *parent = LDSV ? uf_sm + 2 * UF_BLOCK + 2 * (UF_BLOCK / 64) : parent_g;
    unsigned char *rnk_l = reinterpret_cast<unsigned char *>(parent + (LDSV ? n : 0));
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    // (the table in global memory is read and written past the L1: the replaying wavefront's stores must be what the whole
    //  workgroup's next look sees)
    auto ldp = [&](int i) -> int { return LDSV ? parent[i] : __hip_atomic_load(parent + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto stp = [&](int i, int v) {
        if (LDSV)
            parent[i] = v;
        else
            __hip_atomic_store(parent + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    if (LDSV) {
        for (int i = tid; i < n; i += UF_BLOCK) {
            parent[i] = i;
            rnk_l[i] = 0;
        }
    }
    __syncthreads();
    for (long long base = 0; base < n_edges; base += UF_BLOCK) {
        const long long e = base + tid;
        int a = 0, b = 0;
        bool differ = false;
        if (e < n_edges) {
            a = src[e];
            b = dst[e];
            int ra = a, rb = b; // read-only walks: nothing is written while the whole workgroup looks
            for (int p = ldp(ra); p != ra; p = ldp(ra))
                ra = p;
            for (int p = ldp(rb); p != rb; p = ldp(rb))
                rb = p;
            differ = ra != rb;
        }
        ea[tid] = a;
        eb[tid] = b;
        const unsigned long long m = __ballot(differ);
        if (lane == 0)
            flag[wv] = m;
        __syncthreads();
        if (wv == 0) { // the survivors, in row order; every lane runs the same steps on the same words
            for (int w = 0; w < UF_BLOCK / 64; w++) {
                unsigned long long mm = flag[w];
                while (mm) {
                    const int i = __ffsll((long long)mm) - 1;
                    mm &= mm - 1;
                    int x = ea[w * 64 + i], y = eb[w * 64 + i];
                    for (int px = ldp(x); px != x; px = ldp(x)) { // uf_find: path halving (:1249-1256)
                        const int gp = ldp(px);
                        stp(x, gp);
                        x = gp;
                    }
                    for (int py = ldp(y); py != y; py = ldp(y)) {
                        const int gp = ldp(py);
                        stp(y, gp);
                        y = gp;
                    }
                    if (x == y)
                        continue;
                    int rx = LDSV ? rnk_l[x] : rnk_g[x], ry = LDSV ? rnk_l[y] : rnk_g[y];
                    if (rx < ry) { // uf_union: by rank (:1258-1273)
                        const int t = x;
                        x = y;
                        y = t;
                        const int tr = rx;
                        rx = ry;
                        ry = tr;
                    }
                    stp(y, x);
                    if (rx == ry) {
                        if (LDSV)
                            rnk_l[x] = (unsigned char)(rx + 1);
                        else
                            rnk_g[x] = rx + 1;
                    }
                    __builtin_amdgcn_s_waitcnt(0);
                }
            }
        }
        __threadfence_block();
        __syncthreads();
    }
    if (LDSV)
        for (int i = tid; i < n; i += UF_BLOCK)
            parent_g[i] = parent[i];
}
static size_t uf_lds_bytes(int n, bool ldsv) {
    return (size_t)(2 * UF_BLOCK + 2 * (UF_BLOCK / 64)) * sizeof(int) + (ldsv ? (size_t)n * 5 + 16 : 0);
}
size_t mn_lds_optin_limit();                          // mn_kernels.hip: dynamic LDS a workgroup may ask for (64 KB, or what the device grants)
bool mn_lds_grant(const void *kernel, size_t bytes);

static __device__ __forceinline__ int uf_root(const int *parent, int x) { // read-only walk to the root
    while (true) {
        const int p = __hip_atomic_load(parent + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p == x)
            return x;
        x = p;
    }
}

// FAST: hook the larger root under the smaller one (atomicMin keeps the forest acyclic: parents only ever decrease)
__global__ void k_cc_hook(const int *src, const int *dst, long long n_edges, int *parent, int *changed) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_edges)
        return;
    int a = uf_root(parent, src[e]), b = uf_root(parent, dst[e]);
    while (a != b) {
        const int hi = a > b ? a : b, lo = a > b ? b : a;
        const int old = atomicMin(parent + hi, lo);
        if (old == hi) { // hooked a root
            *changed = 1;
            break;
        }
        // someone hooked `hi` meanwhile: carry on from where it points now
        a = uf_root(parent, old);
        b = lo;
    }
}

// One atomic per distinct root of a wavefront (round 4): a graph that is one giant component sent a million atomicAdds to one
// address — 5.7 ms of the 14 ms of a 1M-node run.  Counts are integers: the order they arrive in does not matter.
__global__ void k_cc_root(const int *parent, int n, int *root, int *size) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    int r = -1;
    if (live) {
        r = uf_root(parent, i);
        root[i] = r;
    }
    unsigned long long todo = __ballot(live); // (wave-uniform: every lane takes every turn of the loop)
    while (todo) {
        const int lead = __ffsll((long long)todo) - 1;
        const int r0 = __shfl(r, lead);
        const unsigned long long same = __ballot(live && r == r0);
        if ((int)(threadIdx.x & 63) == lead)
            atomicAdd(size + r0, __popcll(same));
        todo &= ~same;
    }
}

__global__ void k_cc_out(const int *root, const int *size, int n, int *comp_size) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        comp_size[i] = size[root[i]];
}

extern "C" int mn_graph_components(int n, int64_t n_edges, const int *src, const int *dst, int mode, int device, int *component_id,
                                   int *component_size, mn_graph_algo_stats *stats) try {
    if (stats)
        memset(stats, 0, sizeof(*stats));
    if (n < 0 || n_edges < 0 || (n_edges && (!src || !dst)) || (mode != MN_COMPONENTS_EXACT && mode != MN_COMPONENTS_FAST)) {
        aset_err("mn_graph_components: bad arguments");
        return -1;
    }
    if (n == 0)
        return 0;
    for (int64_t e = 0; e < n_edges; e++)
        if (src[e] < 0 || src[e] >= n || dst[e] < 0 || dst[e] >= n) {
            aset_err("mn_graph_components: edge %lld names a node outside [0, %d)", (long long)e, n);
            return -1;
        }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        aset_err("mn_graph_components: HIP device %d not available (no CPU fallback)", device);
        return -1;
    }
    ACHK(hipSetDevice(device));
    Bufs b;
    int *d_src = b.alloc<int>((size_t)n_edges), *d_dst = b.alloc<int>((size_t)n_edges), *d_parent = b.alloc<int>(n), *d_rank = b.alloc<int>(n),
        *d_size = b.alloc<int>(n), *d_root = b.alloc<int>(n), *d_out = b.alloc<int>(n), *d_changed = b.alloc<int>(1);
    if (!d_src || !d_dst || !d_parent || !d_rank || !d_size || !d_root || !d_out || !d_changed) {
        aset_err("mn_graph_components: out of device memory");
        return -1;
    }
    if (n_edges) {
        ACHK(hipMemcpy(d_src, src, (size_t)n_edges * sizeof(int), hipMemcpyHostToDevice));
        ACHK(hipMemcpy(d_dst, dst, (size_t)n_edges * sizeof(int), hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1;
    ACHK(hipEventCreate(&e0));
    ACHK(hipEventCreate(&e1));
    const int nb = (n + 255) / 256;
    ACHK(hipEventRecord(e0, nullptr));
    hipLaunchKernelGGL(k_uf_init, dim3(nb), dim3(256), 0, nullptr, d_parent, d_rank, d_size, n);
    int rounds = 0;
    if (mode == MN_COMPONENTS_EXACT) {
        const char *one = getenv("MN_COMPONENTS_ONE_LANE"); // (the round-3 replay, every edge through one lane: for A/B runs)
        const size_t lds_v = uf_lds_bytes(n, true);
        if (one && atoi(one) == 1) {
            hipLaunchKernelGGL(k_uf_seq, dim3(1), dim3(64), 0, nullptr, d_src, d_dst, (long long)n_edges, d_parent, d_rank);
        } else if (lds_v <= mn_lds_optin_limit() && mn_lds_grant(reinterpret_cast<const void *>(k_uf_blocks<true>), lds_v)) {
            hipLaunchKernelGGL((k_uf_blocks<true>), dim3(1), dim3(UF_BLOCK), lds_v, nullptr, d_src, d_dst, (long long)n_edges, d_parent,
                               d_rank, n);
        } else {
            hipLaunchKernelGGL((k_uf_blocks<false>), dim3(1), dim3(UF_BLOCK), uf_lds_bytes(n, false), nullptr, d_src, d_dst,
                               (long long)n_edges, d_parent, d_rank, n);
        }
        rounds = 1;
    } else if (n_edges) {
        for (int changed = 1; changed && rounds < 64; rounds++) { // (one round settles everything; the loop is the safety net)
            changed = 0;
            ACHK(hipMemsetAsync(d_changed, 0, sizeof(int), nullptr));
            hipLaunchKernelGGL(k_cc_hook, dim3((unsigned)((n_edges + 255) / 256)), dim3(256), 0, nullptr, d_src, d_dst, (long long)n_edges,
                               d_parent, d_changed);
            ACHK(hipMemcpy(&changed, d_changed, sizeof(int), hipMemcpyDeviceToHost));
        }
    }
    hipLaunchKernelGGL(k_cc_root, dim3(nb), dim3(256), 0, nullptr, d_parent, n, d_root, d_size); // :1347-1356
    hipLaunchKernelGGL(k_cc_out, dim3(nb), dim3(256), 0, nullptr, d_root, d_size, n, d_out);
    ACHK(hipEventRecord(e1, nullptr));
    ACHK(hipGetLastError());
    ACHK(hipMemcpy(component_id, d_root, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    ACHK(hipMemcpy(component_size, d_out, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (stats) {
        stats->device_ms = ms;
        stats->iterations = rounds;
    }
    return 0;
} MN_GUARD_END(aset_err, MN_NOTHING, -1)

// ───────────────────────── csr_apply_delta (src/graph_csr.c:175-325) ─────────────────────────
// The reference turns the CSR into per-node lists, replays the delta log in order (INSERT appends, DELETE removes the
// first occurrence by swapping the last element in) and rebuilds the CSR.  A node's final list depends only on ITS
// deltas in log order, so the replay is parallel over nodes: the log is grouped by source with a STABLE radix sort
// (log order kept inside a node), every node copies its old list into a scratch row sized old degree + its deltas,
// replays them, and an exclusive scan of the final lengths gives the new offsets.
__global__ void k_delta_keys(const mn_csr_delta *dl, int nd, int n_new, unsigned *keys, int *vals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nd)
        return;
    const int s = dl[i].src_idx, d = dl[i].dst_idx;
    const bool ok = s >= 0 && s < n_new && d >= 0 && d < n_new; // :221-222: others are skipped
    keys[i] = ok ? (unsigned)s : 0xffffffffu;
    vals[i] = i;
}

__global__ void k_delta_ranges(const unsigned *keys, int nd, const int *old_off, int n_old, int n_new, int *dstart, int *cap) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_new)
        return;
    int lo = 0, hi = nd; // first sorted delta with key >= v
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (keys[mid] < (unsigned)v)
            lo = mid + 1;
        else
            hi = mid;
    }
    int e = lo;
    while (e < nd && keys[e] == (unsigned)v)
        e++;
    dstart[v] = lo;
    const int deg = v < n_old ? old_off[v + 1] - old_off[v] : 0;
    cap[v] = deg + (e - lo);
}

__global__ void k_delta_apply(const mn_csr_delta *dl, const unsigned *keys, const int *order, int nd, const int *old_off,
                              const int *old_tgt, const double *old_w, int n_old, int n_new, const int *dstart, const int *tmp_off,
                              int *tmp_tgt, double *tmp_w, int *cnt) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_new)
        return;
    int *row = tmp_tgt + tmp_off[v];
    double *wrow = tmp_w ? tmp_w + tmp_off[v] : nullptr;
    int n = 0;
    if (v < n_old) {
        const int o = old_off[v];
        n = old_off[v + 1] - o;
        for (int j = 0; j < n; j++) {
            row[j] = old_tgt[o + j];
            if (wrow)
                wrow[j] = old_w ? old_w[o + j] : 0.0;
        }
    }
    for (int x = dstart[v]; x < nd && keys[x] == (unsigned)v; x++) { // this node's deltas in log order
        const mn_csr_delta d = dl[order[x]];
        if (d.op == 2) { // DELETE: first occurrence, swap with the last (:225-238)
            for (int j = 0; j < n; j++)
                if (row[j] == d.dst_idx) {
                    n--;
                    if (j < n) {
                        row[j] = row[n];
                        if (wrow)
                            wrow[j] = wrow[n];
                    }
                    break;
                }
        } else if (d.op == 1) { // INSERT: append (:239-261)
            row[n] = d.dst_idx;
            if (wrow)
                wrow[n] = d.weight;
            n++;
        }
    }
    cnt[v] = n;
}

__global__ void k_delta_compact(const int *tmp_off, const int *tmp_tgt, const double *tmp_w, const int *new_off, const int *cnt,
                                int n_new, int *out_tgt, double *out_w) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_new)
        return;
    const int a = tmp_off[v], b = new_off[v], n = cnt[v];
    for (int j = 0; j < n; j++) {
        out_tgt[b + j] = tmp_tgt[a + j];
        if (out_w)
            out_w[b + j] = tmp_w[a + j];
    }
}

extern "C" int mn_csr_apply_delta(int old_node_count, const int *old_offsets, const int *old_targets, const double *old_weights,
                                  int has_weights, const mn_csr_delta *deltas, int delta_count, int new_node_count, int device,
                                  int *new_offsets, int **new_targets, double **new_weights, int *new_edge_count) try {
    if (old_node_count < 0 || delta_count < 0 || !old_offsets || !new_offsets || !new_targets || !new_edge_count) {
        aset_err("mn_csr_apply_delta: bad arguments");
        return -1;
    }
    if (new_node_count < old_node_count) // :179-180
        new_node_count = old_node_count;
    *new_targets = nullptr;
    if (new_weights)
        *new_weights = nullptr;
    *new_edge_count = 0;
    const int n_old = old_node_count, n_new = new_node_count, nd = delta_count;
    if (n_new == 0) {
        new_offsets[0] = 0;
        return 0;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        aset_err("mn_csr_apply_delta: HIP device %d not available (no CPU fallback)", device);
        return -1;
    }
    ACHK(hipSetDevice(device));
    const int E = n_old ? old_offsets[n_old] : 0;
    Bufs b;
    int *d_off = b.alloc<int>((size_t)n_old + 1), *d_tgt = b.alloc<int>(E), *d_vals = b.alloc<int>(nd), *d_vals_s = b.alloc<int>(nd),
        *d_dstart = b.alloc<int>(n_new), *d_cap = b.alloc<int>(n_new), *d_tmpoff = b.alloc<int>((size_t)n_new + 1), *d_cnt = b.alloc<int>(n_new),
        *d_newoff = b.alloc<int>((size_t)n_new + 1);
    unsigned *d_keys = b.alloc<unsigned>(nd), *d_keys_s = b.alloc<unsigned>(nd);
    double *d_w = has_weights ? b.alloc<double>(E) : nullptr;
    mn_csr_delta *d_dl = b.alloc<mn_csr_delta>(nd);
    if (!d_off || !d_tgt || !d_vals || !d_vals_s || !d_dstart || !d_cap || !d_tmpoff || !d_cnt || !d_newoff || !d_keys || !d_keys_s ||
        !d_dl || (has_weights && !d_w)) {
        aset_err("mn_csr_apply_delta: out of device memory");
        return -1;
    }
    ACHK(hipMemcpy(d_off, old_offsets, ((size_t)n_old + 1) * sizeof(int), hipMemcpyHostToDevice));
    if (n_old == 0)
        ACHK(hipMemset(d_off, 0, sizeof(int)));
    if (E)
        ACHK(hipMemcpy(d_tgt, old_targets, (size_t)E * sizeof(int), hipMemcpyHostToDevice));
    if (has_weights && E && old_weights)
        ACHK(hipMemcpy(d_w, old_weights, (size_t)E * sizeof(double), hipMemcpyHostToDevice));
    if (nd)
        ACHK(hipMemcpy(d_dl, deltas, (size_t)nd * sizeof(mn_csr_delta), hipMemcpyHostToDevice));
    const int nbd = (nd + 255) / 256, nbn = (n_new + 255) / 256;
    const unsigned *keys_sorted = d_keys;
    const int *order = d_vals;
    void *tmp = nullptr;
    if (nd) {
        hipLaunchKernelGGL(k_delta_keys, dim3(nbd), dim3(256), 0, nullptr, d_dl, nd, n_new, d_keys, d_vals);
        size_t bytes = 0;
        if (rocprim::radix_sort_pairs(nullptr, bytes, d_keys, d_keys_s, d_vals, d_vals_s, (size_t)nd, 0, 32, nullptr) != hipSuccess) {
            aset_err("rocprim::radix_sort_pairs (size query) failed");
            return -1;
        }
        ACHK(hipMalloc(&tmp, bytes ? bytes : 16));
        b.p.push_back(tmp);
        if (rocprim::radix_sort_pairs(tmp, bytes, d_keys, d_keys_s, d_vals, d_vals_s, (size_t)nd, 0, 32, nullptr) != hipSuccess) { // stable
            aset_err("rocprim::radix_sort_pairs failed");
            return -1;
        }
        keys_sorted = d_keys_s;
        order = d_vals_s;
    }
    hipLaunchKernelGGL(k_delta_ranges, dim3(nbn), dim3(256), 0, nullptr, keys_sorted, nd, d_off, n_old, n_new, d_dstart, d_cap);
    auto scan = [&](int *in, int *out, int n) -> int { // exclusive scan; out[n] = total
        size_t bytes = 0;
        void *t2 = nullptr;
        if (rocprim::exclusive_scan(nullptr, bytes, in, out, 0, (size_t)n, rocprim::plus<int>(), nullptr) != hipSuccess)
            return -1;
        if (hipMalloc(&t2, bytes ? bytes : 16) != hipSuccess)
            return -1;
        b.p.push_back(t2);
        if (rocprim::exclusive_scan(t2, bytes, in, out, 0, (size_t)n, rocprim::plus<int>(), nullptr) != hipSuccess)
            return -1;
        int last_in = 0, last_out = 0;
        if (hipMemcpy(&last_in, in + n - 1, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(&last_out, out + n - 1, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)
            return -1;
        const int total = last_in + last_out;
        if (hipMemcpy(out + n, &total, sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
            return -1;
        return total;
    };
    const int tmp_total = scan(d_cap, d_tmpoff, n_new);
    if (tmp_total < 0) {
        aset_err("mn_csr_apply_delta: scan failed");
        return -1;
    }
    int *d_tmpt = b.alloc<int>(tmp_total);
    double *d_tmpw = has_weights ? b.alloc<double>(tmp_total) : nullptr;
    if (!d_tmpt || (has_weights && !d_tmpw)) {
        aset_err("mn_csr_apply_delta: out of device memory");
        return -1;
    }
    hipLaunchKernelGGL(k_delta_apply, dim3(nbn), dim3(256), 0, nullptr, d_dl, keys_sorted, order, nd, d_off, d_tgt, d_w, n_old, n_new,
                       d_dstart, d_tmpoff, d_tmpt, d_tmpw, d_cnt);
    const int total = scan(d_cnt, d_newoff, n_new);
    if (total < 0) {
        aset_err("mn_csr_apply_delta: scan failed");
        return -1;
    }
    int *d_outt = b.alloc<int>(total);
    double *d_outw = has_weights ? b.alloc<double>(total) : nullptr;
    if (!d_outt || (has_weights && !d_outw)) {
        aset_err("mn_csr_apply_delta: out of device memory");
        return -1;
    }
    hipLaunchKernelGGL(k_delta_compact, dim3(nbn), dim3(256), 0, nullptr, d_tmpoff, d_tmpt, d_tmpw, d_newoff, d_cnt, n_new, d_outt, d_outw);
    ACHK(hipGetLastError());
    ACHK(hipMemcpy(new_offsets, d_newoff, ((size_t)n_new + 1) * sizeof(int), hipMemcpyDeviceToHost));
    if (total) { // (the reference leaves targets / weights NULL for an empty graph, :282-291)
        *new_targets = (int *)malloc((size_t)total * sizeof(int));
        if (!*new_targets) {
            aset_err("mn_csr_apply_delta: out of memory");
            return -1;
        }
        ACHK(hipMemcpy(*new_targets, d_outt, (size_t)total * sizeof(int), hipMemcpyDeviceToHost));
        if (has_weights && new_weights) {
            *new_weights = (double *)malloc((size_t)total * sizeof(double));
            if (!*new_weights) {
                aset_err("mn_csr_apply_delta: out of memory");
                return -1;
            }
            ACHK(hipMemcpy(*new_weights, d_outw, (size_t)total * sizeof(double), hipMemcpyDeviceToHost));
        }
    }
    *new_edge_count = total;
    return 0;
} MN_GUARD_END(aset_err, MN_NOTHING, -1)

extern "C" void mn_host_free(void *p) { free(p); }
