// mn_graph.hip — Leiden community detection (src/graph_community.c:75-429) over device-resident CSR
// adjacency (src/graph_csr.h:27-34), gfx950.
//
// The reference sweeps nodes in order and applies every move immediately (:158-228); all arithmetic
// is f64 and every per-community sum is taken in adjacency-list order.  Device design:
//   best_move        one wavefront evaluates one node: its edges' (community, weight) pairs are staged
//                    in LDS in list order; lane e decides whether edge e is the first occurrence of
//                    its community, sums that community's weights in list order, computes the gain
//                    expression of :209-210 verbatim in f64 and a max-with-lowest-index reduction
//                    reproduces the strict `gain > best_gain` first-seen tie rule (:212)
//   k_leiden_seq     MN_LEIDEN_SEQUENTIAL: ONE wavefront walks v = 0..N-1 with in-place updates
//                    (agent-scope atomics for label/sum_tot: never a stale L1 line) — community
//                    assignment and Q bit-identical to the reference
//   k_leiden_eval/cmin/win/apply   MN_LEIDEN_BATCHED: a range of nodes evaluated in parallel against
//                    frozen state; a mover commits iff it is the smallest-index mover among the
//                    movers touching its old/target community and its moving neighbours → committed
//                    moves are pairwise independent, realise exactly their computed gain (Q strictly
//                    increases) and the result does not depend on execution order
// O(N) bookkeeping between phases (renumber :317-331, distinct counts :388-403, sum_tot rebuild
// :413-416, the final per-community accumulation of :128-139) is done by the host in the
// reference's order from arrays the kernels produce.
#include "../../include/muninn_hip.h"
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#define DEVI __device__ __forceinline__
#define LEI_CAP 1024 // edges of one node staged in LDS; larger nodes use the global scratch path

struct DevGraph {
    int n;
    const int *off_out, *tgt_out;
    const double *w_out; // null = 1.0
    const int *off_in, *tgt_in;
    const double *w_in;
};

template <bool COH> DEVI int ld_i(const int *p) {
    if (COH)
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <bool COH> DEVI double ld_d(const double *p) {
    if (COH)
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}

// Returns the community node v should move to (== its current one if no strictly positive gain).
// ec/ew/el: staging for the node's edges (community, weight, eligible) in list order — LDS for degree
// <= LEI_CAP, global scratch otherwise; both 16-byte aligned, capacity rounded up to a multiple of 4.
template <bool COH>
DEVI int best_move(const DevGraph &g, int v, const int *label, const double *sum_tot, const double *kdeg, double m,
                   double resolution, int use_both, const int *elig_part, int *ec, double *ew, unsigned char *el, int lane,
                   double *dk_out = nullptr) {
    const int o0 = g.off_out[v], d_out = g.off_out[v + 1] - o0;
    const int i0 = use_both ? g.off_in[v] : 0, d_in = use_both ? g.off_in[v + 1] - i0 : 0;
    const int d = d_out + d_in;
    const int d4 = (d + 3) & ~3;
    const int old = ld_i<COH>(label + v);
    const int mypart = elig_part ? elig_part[v] : 0;
    __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < d4; e += 64) {
        int c = -2; // padding never matches a community
        double w = 0.0;
        unsigned char ok = 0;
        if (e < d) {
            int t;
            if (e < d_out) {
                t = g.tgt_out[o0 + e];
                w = g.w_out ? g.w_out[o0 + e] : 1.0;
            } else {
                t = g.tgt_in[i0 + (e - d_out)];
                w = g.w_in ? g.w_in[i0 + (e - d_out)] : 1.0;
            }
            c = ld_i<COH>(label + t);
            ok = (!elig_part || elig_part[t] == mypart) ? 1 : 0;
        }
        ec[e] = c;
        ew[e] = w;
        el[e] = ok;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    const double k_v = kdeg[v];
    const int4 *ec4 = reinterpret_cast<const int4 *>(ec);
    const double2 *ew2 = reinterpret_cast<const double2 *>(ew);
    const uchar4 *el4 = reinterpret_cast<const uchar4 *>(el);
    double k_v_to_old = 0.0; // weight_to_community(v, old), :163 — list order
    for (int q = 0; q < (d4 >> 2); q++) {
        const int4 cj = ec4[q];
        const double2 wa = ew2[2 * q], wb = ew2[2 * q + 1];
        if (cj.x == old) k_v_to_old += wa.x;
        if (cj.y == old) k_v_to_old += wa.y;
        if (cj.z == old) k_v_to_old += wb.x;
        if (cj.w == old) k_v_to_old += wb.y;
    }
    const double st_old = ld_d<COH>(sum_tot + old);
    double best_gain = 0.0;
    int best = old;
    for (int base = 0; base < d; base += 64) {
        const int e = base + lane;
        const int c = e < d ? ec[e] : -3;
        bool cand = e < d && el[e] && c != old;
        // one pass over the list: the in-order weight sum of community c (weight_to_community, :206) and
        // "an eligible earlier edge already carries c" (the dedup scan of :173-199)
        double sacc = 0.0;
        bool dup = false;
        for (int q = 0; q < (d4 >> 2); q++) {
            const int4 cj = ec4[q];
            const double2 wa = ew2[2 * q], wb = ew2[2 * q + 1];
            const uchar4 ej = el4[q];
            const int j = q << 2;
            if (cj.x == c) { sacc += wa.x; dup |= (j < e) && ej.x; }
            if (cj.y == c) { sacc += wa.y; dup |= (j + 1 < e) && ej.y; }
            if (cj.z == c) { sacc += wb.x; dup |= (j + 2 < e) && ej.z; }
            if (cj.w == c) { sacc += wb.y; dup |= (j + 3 < e) && ej.w; }
        }
        cand = cand && !dup;
        double gain = -1.0, dk = 0.0;
        if (cand) {
            const double st_c = ld_d<COH>(sum_tot + c);
            dk = sacc - k_v_to_old;
            gain = (sacc - k_v_to_old) / m + resolution * k_v * (st_old - k_v - st_c) / (2.0 * m * m); // :209-210
            if (!(gain > 0.0))
                gain = -1.0; // also drops NaN: `gain > best_gain` is false for it
        }
        // max gain, lowest lane on ties == first candidate with the strictly largest gain
        double bg = gain;
        int bl = lane;
        for (int mk = 32; mk >= 1; mk >>= 1) {
            double og = __shfl_xor(bg, mk);
            int ol = __shfl_xor(bl, mk);
            if (og > bg || (og == bg && ol < bl)) {
                bg = og;
                bl = ol;
            }
        }
        if (bg > best_gain) { // strict: an equal gain in a later chunk does not replace (:212)
            best_gain = bg;
            best = __shfl(c, bl);
            double bdk = __shfl(dk, bl);
            if (dk_out)
                *dk_out = bdk;
        }
    }
    return best;
}

struct LeiArgs {
    DevGraph g;
    int *label;
    double *sum_tot;
    const double *kdeg;
    double m, resolution;
    int use_both;
    const int *elig_part; // refinement: phase-1 partition; null for local moving
    int *scratch_c;       // global scratch for nodes with more than LEI_CAP edges: [blocks][max_deg]
    double *scratch_w;
    unsigned char *scratch_e;
    int max_deg;
    int *out; // [0] moves [1] sweeps
    int max_sweeps;
    // batched
    int b0, b1;
    int *dec, *cmin;
    unsigned char *win;
    double *dk;
    unsigned long long *Jq, *Lq; // fixed-point (2^20) tallies of the movers' degrees per community
    int apply_on_device;         // 0: weighted graph → the host applies winners in node order
    int lds_cap;                 // k_leiden_eval: edges staged in LDS per node (multiple of 16, ≤ LEI_CAP)
};

#define LEI_FX 1048576.0
DEVI unsigned long long fx_up(double k) { return (unsigned long long)ceil(k * LEI_FX); }

DEVI int node_degree(const LeiArgs &a, int v) {
    return a.g.off_out[v + 1] - a.g.off_out[v] + (a.use_both ? a.g.off_in[v + 1] - a.g.off_in[v] : 0);
}

__global__ void __launch_bounds__(64) k_leiden_seq(LeiArgs a) {
    __shared__ __align__(16) double lds_w[LEI_CAP];
    __shared__ __align__(16) int lds_c[LEI_CAP];
    __shared__ __align__(16) unsigned char lds_e[LEI_CAP];
    const int lane = threadIdx.x;
    int total = 0, improved = 1, sweeps = 0;
    while (improved && sweeps < a.max_sweeps) { // :154-229
        improved = 0;
        sweeps++;
        for (int v = 0; v < a.g.n; v++) {
            const int old = ld_i<true>(a.label + v);
            int best; // two call sites: LDS staging keeps its address space (a runtime-selected pointer would be FLAT)
            if (node_degree(a, v) <= LEI_CAP)
                best = best_move<true>(a.g, v, a.label, a.sum_tot, a.kdeg, a.m, a.resolution, a.use_both, a.elig_part, lds_c,
                                       lds_w, lds_e, lane);
            else
                best = best_move<true>(a.g, v, a.label, a.sum_tot, a.kdeg, a.m, a.resolution, a.use_both, a.elig_part,
                                       a.scratch_c, a.scratch_w, a.scratch_e, lane);
            if (best != old) { // :220-227
                if (lane == 0) {
                    const double k_v = a.kdeg[v];
                    double so = ld_d<true>(a.sum_tot + old), sb = ld_d<true>(a.sum_tot + best);
                    __hip_atomic_store(a.sum_tot + old, so - k_v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(a.sum_tot + best, sb + k_v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(a.label + v, best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __builtin_amdgcn_s_waitcnt(0);
                improved = 1;
                total++;
            }
        }
    }
    if (lane == 0) {
        a.out[0] = total;
        a.out[1] = sweeps;
        a.out[2] = improved; // 1 → stopped by max_sweeps, not converged
    }
}

// LDS staging sized for the graph's largest node (a.lds_cap ≤ LEI_CAP edges, 13 B each) instead of LEI_CAP: with the
// full 13 KB only 12 wavefronts fit a CU and the kernel, which is a chain of dependent gathers, starves.
__global__ void __launch_bounds__(64) k_leiden_eval(LeiArgs a) {
    extern __shared__ __align__(16) unsigned char lei_smem[];
    double *lds_w = reinterpret_cast<double *>(lei_smem);
    int *lds_c = reinterpret_cast<int *>(lds_w + a.lds_cap);
    unsigned char *lds_e = reinterpret_cast<unsigned char *>(lds_c + a.lds_cap);
    const int v = a.b0 + blockIdx.x;
    if (v >= a.b1)
        return;
    double dk = 0.0;
    int best;
    if (node_degree(a, v) <= a.lds_cap) {
        best = best_move<false>(a.g, v, a.label, a.sum_tot, a.kdeg, a.m, a.resolution, a.use_both, a.elig_part, lds_c, lds_w,
                                lds_e, threadIdx.x, &dk);
    } else {
        const size_t o = (size_t)blockIdx.x * a.max_deg;
        best = best_move<false>(a.g, v, a.label, a.sum_tot, a.kdeg, a.m, a.resolution, a.use_both, a.elig_part, a.scratch_c + o,
                                a.scratch_w + o, a.scratch_e + o, threadIdx.x, &dk);
    }
    if (threadIdx.x == 0) {
        a.dec[v - a.b0] = best;
        a.dk[v - a.b0] = dk;
    }
}

__global__ void k_leiden_cmin(LeiArgs a) {
    const int v = a.b0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= a.b1)
        return;
    const int old = a.label[v], best = a.dec[v - a.b0];
    if (best == old)
        return;
    atomicMin(a.cmin + old, v);
    atomicMin(a.cmin + best, v);
    const unsigned long long q = fx_up(a.kdeg[v]);
    atomicAdd(a.Lq + old, q);
    atomicAdd(a.Jq + best, q);
}

__global__ void k_leiden_win(LeiArgs a) {
    const int v = a.b0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= a.b1)
        return;
    const int old = a.label[v], best = a.dec[v - a.b0];
    unsigned char win = 0;
    if (best != old) {
        int free_nb = 1;
        for (int pass = 0; free_nb && pass < (a.use_both ? 2 : 1); pass++) {
            const int *off = pass ? a.g.off_in : a.g.off_out;
            const int *tgt = pass ? a.g.tgt_in : a.g.tgt_out;
            for (int x = off[v]; x < off[v + 1]; x++) {
                const int w = tgt[x];
                if (w >= a.b0 && w < v && a.dec[w - a.b0] != a.label[w]) {
                    free_nb = 0;
                    break;
                }
            }
        }
        if (free_nb) {
            const double k_v = a.kdeg[v];
            const unsigned long long q = fx_up(k_v);
            const double Lo = (double)(a.Lq[old] - q) / LEI_FX, Jc = (double)(a.Jq[best] - q) / LEI_FX;
            const double gain2 = a.dk[v - a.b0] / a.m +
                                 a.resolution * k_v * (a.sum_tot[old] - Lo - k_v - a.sum_tot[best] - Jc) / (2.0 * a.m * a.m);
            const int strict = a.cmin[old] == v && a.cmin[best] == v;
            win = (unsigned char)((gain2 > 0.0 ? 1 : 0) | (strict ? 2 : 0));
            if (gain2 > 0.0)
                atomicAdd(a.out + 1, 1); // safe winners in this round
        }
    }
    a.win[v - a.b0] = win;
}

__global__ void k_leiden_apply(LeiArgs a) {
    const int v = a.b0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= a.b1)
        return;
    const int old = a.label[v], best = a.dec[v - a.b0];
    if (best == old)
        return;
    a.cmin[old] = 0x7fffffff;
    a.cmin[best] = 0x7fffffff;
    a.Lq[old] = 0;
    a.Jq[best] = 0;
    const int use_bit = a.out[1] > 0 ? 1 : 2;
    if (a.apply_on_device && (a.win[v - a.b0] & use_bit)) {
        // unweighted graph: degrees are integers, f64 atomic adds are exact → order-free
        atomicAdd(a.sum_tot + old, -a.kdeg[v]);
        atomicAdd(a.sum_tot + best, a.kdeg[v]);
        a.label[v] = best;
        atomicAdd(a.out, 1);
    }
}

// host-ordered application for weighted graphs: scatter the changed entries back
__global__ void k_scatter_d(double *dst, const int *idx, const double *val, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        dst[idx[i]] = val[i];
}
__global__ void k_scatter_i(int *dst, const int *idx, const int *val, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        dst[idx[i]] = val[i];
}

// weighted_degree (:95-104) and weight_to_community(v, community[v]) (:75-90), list order, f64
__global__ void k_wdeg(DevGraph g, int use_both, double *kdeg) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= g.n)
        return;
    double k = 0.0;
    for (int e = g.off_out[v]; e < g.off_out[v + 1]; e++)
        k += g.w_out ? g.w_out[e] : 1.0;
    if (use_both)
        for (int e = g.off_in[v]; e < g.off_in[v + 1]; e++)
            k += g.w_in ? g.w_in[e] : 1.0;
    kdeg[v] = k;
}

__global__ void k_w2c_self(DevGraph g, int use_both, const int *label, double *out) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= g.n)
        return;
    const int c = label[v];
    double s = 0.0;
    for (int e = g.off_out[v]; e < g.off_out[v + 1]; e++)
        if (label[g.tgt_out[e]] == c)
            s += g.w_out ? g.w_out[e] : 1.0;
    if (use_both)
        for (int e = g.off_in[v]; e < g.off_in[v + 1]; e++)
            if (label[g.tgt_in[e]] == c)
                s += g.w_in ? g.w_in[e] : 1.0;
    out[v] = s;
}

// ───────────────────────── host ─────────────────────────

static thread_local std::string g_gerr;
static void gset_err(const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_gerr = buf;
}
extern "C" const char *mn_graph_last_error(void) { return g_gerr.c_str(); }

#define GCHK(expr)                                                                                 \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess) {                                                                   \
            gset_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return -1;                                                                             \
        }                                                                                          \
    } while (0)

struct mn_graph {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int n = 0;
    long long e_out = 0, e_in = 0;
    int max_deg_both = 0, max_deg_out = 0;
    bool weighted = false;
    int *off_out = nullptr, *tgt_out = nullptr, *off_in = nullptr, *tgt_in = nullptr;
    double *w_out = nullptr, *w_in = nullptr;
    double last_ms = 0;
    mn_leiden_stats stats = {};
};

template <typename T> static int up(T **dst, const T *src, size_t n) {
    *dst = nullptr;
    GCHK(hipMalloc(dst, (n ? n : 1) * sizeof(T)));
    if (n)
        GCHK(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

extern "C" mn_graph *mn_graph_create(int n_nodes, const int *off_out, const int *tgt_out, const double *w_out,
                                     const int *off_in, const int *tgt_in, const double *w_in, int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        gset_err("mn_graph_create: HIP device %d not available (no CPU fallback)", device);
        return nullptr;
    }
    if (n_nodes < 0 || !off_out || !off_in) {
        gset_err("mn_graph_create: bad arguments");
        return nullptr;
    }
    mn_graph *g = new mn_graph();
    g->device = device;
    g->n = n_nodes;
    g->e_out = n_nodes ? off_out[n_nodes] : 0;
    g->e_in = n_nodes ? off_in[n_nodes] : 0;
    g->weighted = w_out != nullptr || w_in != nullptr;
    for (int v = 0; v < n_nodes; v++) {
        int dO = off_out[v + 1] - off_out[v], dI = off_in[v + 1] - off_in[v];
        if (dO > g->max_deg_out) g->max_deg_out = dO;
        if (dO + dI > g->max_deg_both) g->max_deg_both = dO + dI;
    }
    bool ok = hipSetDevice(device) == hipSuccess && hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreate(&g->ev0) == hipSuccess && hipEventCreate(&g->ev1) == hipSuccess;
    ok = ok && up(&g->off_out, off_out, (size_t)n_nodes + 1) == 0 && up(&g->tgt_out, tgt_out, (size_t)g->e_out) == 0 &&
         up(&g->off_in, off_in, (size_t)n_nodes + 1) == 0 && up(&g->tgt_in, tgt_in, (size_t)g->e_in) == 0;
    if (ok && w_out)
        ok = up(&g->w_out, w_out, (size_t)g->e_out) == 0;
    if (ok && w_in)
        ok = up(&g->w_in, w_in, (size_t)g->e_in) == 0;
    if (!ok) {
        mn_graph_destroy(g);
        return nullptr;
    }
    return g;
}

// One direction of a blocked CSR (the rows of "{t}_csr_fwd" / "{t}_csr_rev") straight into device buffers: per
// block, node count = offsets_bytes/4 - 1 and edge count = targets_bytes/4 (csr_deserialize, src/graph_csr.c:122-163);
// offsets are rebased by the running edge count, targets are already global (csr_merge_blocks, :402-477).
static int upload_blocks(const mn_csr_block *blk, int nb, int n_nodes, std::vector<int> &off, int **d_tgt, double **d_w,
                         long long *n_edges, bool *weighted) {
    off.assign((size_t)n_nodes + 1, 0);
    long long edges = 0;
    for (int b = 0; b < nb; b++) {
        if (!blk[b].offsets || blk[b].offsets_bytes < 4 || blk[b].targets_bytes < 0 || blk[b].targets_bytes % 4) {
            gset_err("mn_graph_create_blocked: block %d is malformed", b);
            return -1;
        }
        edges += blk[b].targets_bytes / 4;
    }
    if (edges > 0x7fffffffLL) {
        gset_err("mn_graph_create_blocked: %lld edges exceed int32 offsets", edges);
        return -1;
    }
    *weighted = nb > 0 && blk[0].weights != nullptr && blk[0].weights_bytes > 0; // has_weights of the first block (:415)
    *d_tgt = nullptr;
    *d_w = nullptr;
    GCHK(hipMalloc(d_tgt, (size_t)std::max<long long>(1, edges) * sizeof(int)));
    if (*weighted)
        GCHK(hipMalloc(d_w, (size_t)std::max<long long>(1, edges) * sizeof(double)));
    long long eoff = 0;
    int noff = 0;
    for (int b = 0; b < nb; b++) {
        const int bn = blk[b].offsets_bytes / 4 - 1, be = blk[b].targets_bytes / 4;
        const int *bo = static_cast<const int *>(blk[b].offsets);
        for (int i = 0; i < bn && noff + i < n_nodes; i++) {
            if (bo[i] < 0 || bo[i] > be) {
                gset_err("mn_graph_create_blocked: block %d offsets out of range", b);
                return -1;
            }
            off[(size_t)noff + i] = (int)(bo[i] + eoff);
        }
        if (be > 0 && blk[b].targets)
            GCHK(hipMemcpy(*d_tgt + eoff, blk[b].targets, (size_t)be * sizeof(int), hipMemcpyHostToDevice));
        if (*weighted && be > 0) {
            if (!blk[b].weights || blk[b].weights_bytes != be * 8) {
                gset_err("mn_graph_create_blocked: block %d weights do not match its targets", b);
                return -1;
            }
            GCHK(hipMemcpy(*d_w + eoff, blk[b].weights, (size_t)be * sizeof(double), hipMemcpyHostToDevice));
        }
        eoff += be;
        noff += bn;
    }
    for (int i = std::min(noff, n_nodes); i <= n_nodes; i++) // sentinel, and nodes the blocks did not cover (:470-475)
        off[(size_t)i] = (int)eoff;
    *n_edges = eoff;
    return 0;
}

extern "C" mn_graph *mn_graph_create_blocked(int n_nodes, const mn_csr_block *fwd, int n_fwd, const mn_csr_block *rev, int n_rev,
                                             int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        gset_err("mn_graph_create_blocked: HIP device %d not available (no CPU fallback)", device);
        return nullptr;
    }
    if (n_nodes < 0 || n_fwd < 0 || n_rev < 0 || (n_fwd && !fwd) || (n_rev && !rev)) {
        gset_err("mn_graph_create_blocked: bad arguments");
        return nullptr;
    }
    mn_graph *g = new mn_graph();
    g->device = device;
    g->n = n_nodes;
    std::vector<int> oo, oi;
    bool wo = false, wi = false;
    bool ok = hipSetDevice(device) == hipSuccess && hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreate(&g->ev0) == hipSuccess && hipEventCreate(&g->ev1) == hipSuccess;
    ok = ok && upload_blocks(fwd, n_fwd, n_nodes, oo, &g->tgt_out, &g->w_out, &g->e_out, &wo) == 0 &&
         upload_blocks(rev, n_rev, n_nodes, oi, &g->tgt_in, &g->w_in, &g->e_in, &wi) == 0;
    if (ok) {
        // out-of-range targets would fault in the kernels: the rows come from a file, so look before launching
        std::vector<int> chk;
        for (int dir = 0; dir < 2 && ok; dir++) {
            const long long ne = dir ? g->e_in : g->e_out;
            chk.resize((size_t)ne);
            if (ne)
                ok = hipMemcpy(chk.data(), dir ? g->tgt_in : g->tgt_out, (size_t)ne * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
            for (long long e = 0; e < ne && ok; e++)
                if (chk[(size_t)e] < 0 || chk[(size_t)e] >= n_nodes) {
                    gset_err("mn_graph_create_blocked: target %d out of range", chk[(size_t)e]);
                    ok = false;
                }
        }
    }
    if (ok) {
        g->weighted = wo || wi;
        for (int v = 0; v < n_nodes; v++) {
            int dO = oo[v + 1] - oo[v], dI = oi[v + 1] - oi[v];
            if (dO < 0 || dI < 0) {
                gset_err("mn_graph_create_blocked: offsets are not monotone at node %d", v);
                ok = false;
                break;
            }
            if (dO > g->max_deg_out) g->max_deg_out = dO;
            if (dO + dI > g->max_deg_both) g->max_deg_both = dO + dI;
        }
    }
    ok = ok && up(&g->off_out, oo.data(), (size_t)n_nodes + 1) == 0 && up(&g->off_in, oi.data(), (size_t)n_nodes + 1) == 0;
    if (!ok) {
        mn_graph_destroy(g);
        return nullptr;
    }
    return g;
}

extern "C" void mn_graph_destroy(mn_graph *g) {
    if (!g)
        return;
    (void)hipSetDevice(g->device);
    if (g->stream)
        (void)hipStreamSynchronize(g->stream);
    (void)hipFree(g->off_out); (void)hipFree(g->tgt_out); (void)hipFree(g->off_in); (void)hipFree(g->tgt_in);
    (void)hipFree(g->w_out); (void)hipFree(g->w_in);
    if (g->ev0) (void)hipEventDestroy(g->ev0);
    if (g->ev1) (void)hipEventDestroy(g->ev1);
    if (g->stream) (void)hipStreamDestroy(g->stream);
    delete g;
}

static int renumber(std::vector<int> &c) { // :317-331
    const int N = (int)c.size();
    std::vector<int> map((size_t)N, -1);
    int next = 0;
    for (int i = 0; i < N; i++) {
        if (map[c[i]] == -1)
            map[c[i]] = next++;
        c[i] = map[c[i]];
    }
    return next;
}

static int distinct(const std::vector<int> &c) {
    std::vector<unsigned char> seen(c.size(), 0);
    int n = 0;
    for (int x : c)
        if (!seen[x]) {
            seen[x] = 1;
            n++;
        }
    return n;
}

struct LeiDev {
    int *label = nullptr, *refined = nullptr, *out = nullptr, *dec = nullptr, *cmin = nullptr, *sc = nullptr, *sidx = nullptr,
        *sival = nullptr;
    unsigned char *win = nullptr, *se = nullptr;
    double *sum_tot = nullptr, *kdeg = nullptr, *tmp = nullptr, *sw = nullptr, *dk = nullptr, *sdval = nullptr;
    unsigned long long *Jq = nullptr, *Lq = nullptr;
    ~LeiDev() {
        (void)hipFree(label); (void)hipFree(refined); (void)hipFree(out); (void)hipFree(dec); (void)hipFree(cmin);
        (void)hipFree(win); (void)hipFree(sc); (void)hipFree(sum_tot); (void)hipFree(kdeg); (void)hipFree(tmp);
        (void)hipFree(sw); (void)hipFree(se); (void)hipFree(dk); (void)hipFree(Jq); (void)hipFree(Lq); (void)hipFree(sidx);
        (void)hipFree(sival); (void)hipFree(sdval);
    }
};

// host mirrors of the phase's label / sum_tot (weighted graphs: winners are applied here in node order)
struct HostState {
    std::vector<int> *label;
    std::vector<double> *sum_tot;
    const std::vector<double> *k;
    int *d_sidx, *d_sival;
    double *d_sdval;
};

// one phase (local moving when elig_part == nullptr, refinement otherwise); returns moves, -1 on error
static long long run_phase(mn_graph *g, LeiArgs a, int mode, int batch, int64_t *sweeps_out, HostState *hs) {
    hipStream_t st = g->stream;
    int out[3] = {0, 0, 0};
    if (mode == MN_LEIDEN_SEQUENTIAL) {
        GCHK(hipMemsetAsync(a.out, 0, 3 * sizeof(int), st));
        hipLaunchKernelGGL(k_leiden_seq, dim3(1), dim3(64), 0, st, a);
        GCHK(hipGetLastError());
        GCHK(hipMemcpyAsync(out, a.out, sizeof(out), hipMemcpyDeviceToHost, st));
        GCHK(hipStreamSynchronize(st));
        *sweeps_out += out[1];
        if (out[2]) {
            gset_err("mn_graph_leiden: sweeps did not converge within %d (asymmetric adjacency?)", a.max_sweeps);
            return -1;
        }
        return out[0];
    }
    long long total = 0;
    int improved = 1, sweeps = 0;
    std::vector<int> h_dec((size_t)batch), ch_idx, ch_ival, touched;
    std::vector<unsigned char> h_win((size_t)batch);
    std::vector<double> ch_dval;
    while (improved && sweeps < a.max_sweeps) {
        improved = 0;
        sweeps++;
        GCHK(hipMemsetAsync(a.out, 0, 3 * sizeof(int), st));
        long long sweep_moves = 0;
        for (int b = 0; b < g->n; b += batch) {
            a.b0 = b;
            a.b1 = b + batch < g->n ? b + batch : g->n;
            const int nb = a.b1 - a.b0;
            GCHK(hipMemsetAsync(a.out + 1, 0, sizeof(int), st)); // safe winners of this round
            hipLaunchKernelGGL(k_leiden_eval, dim3(nb), dim3(64), (size_t)a.lds_cap * 13, st, a);
            hipLaunchKernelGGL(k_leiden_cmin, dim3((nb + 255) / 256), dim3(256), 0, st, a);
            hipLaunchKernelGGL(k_leiden_win, dim3((nb + 255) / 256), dim3(256), 0, st, a);
            ch_idx.clear();
            ch_ival.clear();
            touched.clear();
            if (!a.apply_on_device) {
                // weighted graph: several winners may share a community and f64 addition is not
                // associative → apply them here, in node order, on the host mirrors
                GCHK(hipMemcpyAsync(h_dec.data(), a.dec, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost, st));
                GCHK(hipMemcpyAsync(h_win.data(), a.win, (size_t)nb, hipMemcpyDeviceToHost, st));
                GCHK(hipMemcpyAsync(out, a.out, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
                GCHK(hipStreamSynchronize(st));
                const int use_bit = out[1] > 0 ? 1 : 2;
                std::vector<int> &L = *hs->label;
                std::vector<double> &S = *hs->sum_tot;
                for (int v = a.b0; v < a.b1; v++) {
                    const int old = L[v], best = h_dec[v - a.b0];
                    if (best != old && (h_win[v - a.b0] & use_bit)) {
                        S[old] -= (*hs->k)[v];
                        S[best] += (*hs->k)[v];
                        L[v] = best;
                        ch_idx.push_back(v);
                        ch_ival.push_back(best);
                        touched.push_back(old);
                        touched.push_back(best);
                        sweep_moves++;
                    }
                }
            }
            hipLaunchKernelGGL(k_leiden_apply, dim3((nb + 255) / 256), dim3(256), 0, st, a); // resets tallies (+ applies)
            if (!ch_idx.empty()) {
                int nc = (int)ch_idx.size();
                GCHK(hipMemcpyAsync(hs->d_sidx, ch_idx.data(), (size_t)nc * sizeof(int), hipMemcpyHostToDevice, st));
                GCHK(hipMemcpyAsync(hs->d_sival, ch_ival.data(), (size_t)nc * sizeof(int), hipMemcpyHostToDevice, st));
                hipLaunchKernelGGL(k_scatter_i, dim3((nc + 255) / 256), dim3(256), 0, st, a.label, hs->d_sidx, hs->d_sival, nc);
                GCHK(hipStreamSynchronize(st)); // d_sidx is reused for the sum_tot scatter
                std::sort(touched.begin(), touched.end());
                touched.erase(std::unique(touched.begin(), touched.end()), touched.end());
                nc = (int)touched.size();
                ch_dval.resize((size_t)nc);
                for (int i = 0; i < nc; i++)
                    ch_dval[(size_t)i] = (*hs->sum_tot)[touched[(size_t)i]];
                GCHK(hipMemcpyAsync(hs->d_sidx, touched.data(), (size_t)nc * sizeof(int), hipMemcpyHostToDevice, st));
                GCHK(hipMemcpyAsync(hs->d_sdval, ch_dval.data(), (size_t)nc * sizeof(double), hipMemcpyHostToDevice, st));
                hipLaunchKernelGGL(k_scatter_d, dim3((nc + 255) / 256), dim3(256), 0, st, a.sum_tot, hs->d_sidx, hs->d_sdval, nc);
                GCHK(hipStreamSynchronize(st));
            }
        }
        GCHK(hipGetLastError());
        GCHK(hipMemcpyAsync(out, a.out, sizeof(int), hipMemcpyDeviceToHost, st));
        GCHK(hipStreamSynchronize(st));
        if (!a.apply_on_device)
            out[0] = (int)sweep_moves;
        if (out[0]) {
            improved = 1;
            total += out[0];
        }
    }
    *sweeps_out += sweeps;
    return total;
}

extern "C" int mn_graph_leiden(mn_graph *g, double resolution, int use_both, int mode, int batch, int *community_out,
                               double *modularity_out) {
    GCHK(hipSetDevice(g->device));
    const int N = g->n;
    memset(&g->stats, 0, sizeof(g->stats));
    if (modularity_out)
        *modularity_out = 0.0;
    if (N == 0)
        return 0;
    if (mode == MN_LEIDEN_BATCHED && batch <= 1) {
        // A round costs about the same wall time up to a few full waves of the chip (8192 resident wavefronts), and
        // larger rounds need more sweeps.  Measured on the cfg5 graph (N = 500k, <k> = 37): round size 6 250 → 205 ms,
        // 12 500 → 124 ms, 25 000 → 107 ms, 50 000 → 121 ms, 100 000 → 170 ms, modularity 0.668–0.674 throughout.
        batch = (int)std::min<long long>(32768, std::max<long long>(256, N / 16));
    }
    hipStream_t st = g->stream;
    DevGraph dg = {N, g->off_out, g->tgt_out, g->w_out, g->off_in, g->tgt_in, g->w_in};
    LeiDev d;
    const int max_deg = ((use_both ? g->max_deg_both : g->max_deg_out) + 7) & ~3; // int4-aligned scratch stride
    const int nslots = mode == MN_LEIDEN_SEQUENTIAL ? 1 : batch;
    GCHK(hipMalloc(&d.label, (size_t)N * sizeof(int)));
    GCHK(hipMalloc(&d.refined, (size_t)N * sizeof(int)));
    GCHK(hipMalloc(&d.sum_tot, (size_t)N * sizeof(double)));
    GCHK(hipMalloc(&d.kdeg, (size_t)N * sizeof(double)));
    GCHK(hipMalloc(&d.tmp, (size_t)N * sizeof(double)));
    GCHK(hipMalloc(&d.out, 4 * sizeof(int)));
    if (mode == MN_LEIDEN_BATCHED) {
        GCHK(hipMalloc(&d.dec, (size_t)batch * sizeof(int)));
        GCHK(hipMalloc(&d.dk, (size_t)batch * sizeof(double)));
        GCHK(hipMalloc(&d.win, (size_t)batch));
        GCHK(hipMalloc(&d.cmin, (size_t)N * sizeof(int)));
        GCHK(hipMalloc(&d.Jq, (size_t)N * sizeof(unsigned long long)));
        GCHK(hipMalloc(&d.Lq, (size_t)N * sizeof(unsigned long long)));
        GCHK(hipMalloc(&d.sidx, (size_t)(2 * batch + 2) * sizeof(int)));
        GCHK(hipMalloc(&d.sival, (size_t)(2 * batch + 2) * sizeof(int)));
        GCHK(hipMalloc(&d.sdval, (size_t)(2 * batch + 2) * sizeof(double)));
        GCHK(hipMemsetAsync(d.cmin, 0x7f, (size_t)N * sizeof(int), st)); // 0x7f7f7f7f > any node index
        GCHK(hipMemsetAsync(d.Jq, 0, (size_t)N * sizeof(unsigned long long), st));
        GCHK(hipMemsetAsync(d.Lq, 0, (size_t)N * sizeof(unsigned long long), st));
    }
    if (max_deg > LEI_CAP) {
        GCHK(hipMalloc(&d.sc, (size_t)nslots * max_deg * sizeof(int)));
        GCHK(hipMalloc(&d.sw, (size_t)nslots * max_deg * sizeof(double)));
        GCHK(hipMalloc(&d.se, (size_t)nslots * max_deg));
    }
    GCHK(hipEventRecord(g->ev0, st));
    // k[i], m (:344-350) — per-node sums on the device, the running total in node order on the host
    hipLaunchKernelGGL(k_wdeg, dim3((N + 255) / 256), dim3(256), 0, st, dg, use_both, d.kdeg);
    std::vector<double> k((size_t)N);
    GCHK(hipMemcpyAsync(k.data(), d.kdeg, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, st));
    GCHK(hipStreamSynchronize(st));
    double m = 0.0;
    for (int i = 0; i < N; i++)
        m += k[i];
    m /= 2.0;
    std::vector<int> community((size_t)N), refined((size_t)N);
    for (int i = 0; i < N; i++)
        community[i] = i;
    if (m <= 0.0) { // :351-356
        memcpy(community_out, community.data(), (size_t)N * sizeof(int));
        return 0;
    }
    std::vector<double> sum_tot(k);
    GCHK(hipMemcpyAsync(d.label, community.data(), (size_t)N * sizeof(int), hipMemcpyHostToDevice, st));
    GCHK(hipMemcpyAsync(d.sum_tot, sum_tot.data(), (size_t)N * sizeof(double), hipMemcpyHostToDevice, st));

    LeiArgs a;
    memset(&a, 0, sizeof(a));
    a.g = dg;
    a.kdeg = d.kdeg;
    a.m = m;
    a.resolution = resolution;
    a.use_both = use_both;
    a.scratch_c = d.sc;
    a.scratch_w = d.sw;
    a.scratch_e = d.se;
    a.max_deg = max_deg;
    a.out = d.out;
    a.max_sweeps = 100000;
    a.dec = d.dec;
    a.cmin = d.cmin;
    a.win = d.win;
    a.dk = d.dk;
    a.Jq = d.Jq;
    a.Lq = d.Lq;
    a.apply_on_device = g->weighted ? 0 : 1;
    a.lds_cap = std::min(LEI_CAP, std::max(64, (max_deg + 15) & ~15));
    std::vector<double> r_sum_tot;
    HostState hs = {&community, &sum_tot, &k, d.sidx, d.sival, d.sdval};

    for (int iter = 0; iter < 100; iter++) { // :368-417
        a.label = d.label;
        a.sum_tot = d.sum_tot;
        a.elig_part = nullptr;
        hs.label = &community;
        hs.sum_tot = &sum_tot;
        long long moves = run_phase(g, a, mode, batch, &g->stats.move_sweeps, &hs);
        if (moves < 0)
            return -1;
        g->stats.iterations++;
        g->stats.moves += moves;
        if (moves == 0)
            break;
        // refinement (:238-312): singletons, r_sum_tot = k
        GCHK(hipMemcpyAsync(community.data(), d.label, (size_t)N * sizeof(int), hipMemcpyDeviceToHost, st));
        for (int i = 0; i < N; i++)
            refined[i] = i;
        GCHK(hipMemcpyAsync(d.refined, refined.data(), (size_t)N * sizeof(int), hipMemcpyHostToDevice, st));
        GCHK(hipMemcpyAsync(d.tmp, d.kdeg, (size_t)N * sizeof(double), hipMemcpyDeviceToDevice, st));
        a.label = d.refined;
        a.sum_tot = d.tmp;
        a.elig_part = d.label;
        r_sum_tot = k;
        hs.label = &refined;
        hs.sum_tot = &r_sum_tot;
        if (run_phase(g, a, mode, batch, &g->stats.refine_sweeps, &hs) < 0)
            return -1;
        GCHK(hipMemcpyAsync(refined.data(), d.refined, (size_t)N * sizeof(int), hipMemcpyDeviceToHost, st));
        GCHK(hipStreamSynchronize(st));
        if (distinct(refined) <= distinct(community)) // :388-408
            community = refined;
        renumber(community);
        std::fill(sum_tot.begin(), sum_tot.end(), 0.0); // :413-416, node order
        for (int i = 0; i < N; i++)
            sum_tot[community[i]] += k[i];
        GCHK(hipMemcpyAsync(d.label, community.data(), (size_t)N * sizeof(int), hipMemcpyHostToDevice, st));
        GCHK(hipMemcpyAsync(d.sum_tot, sum_tot.data(), (size_t)N * sizeof(double), hipMemcpyHostToDevice, st));
    }
    GCHK(hipMemcpyAsync(community.data(), d.label, (size_t)N * sizeof(int), hipMemcpyDeviceToHost, st));
    GCHK(hipStreamSynchronize(st));
    const int K = renumber(community); // :420
    // compute_modularity (:109-142): per-node terms on the device, accumulation in node order here
    GCHK(hipMemcpyAsync(d.label, community.data(), (size_t)N * sizeof(int), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_w2c_self, dim3((N + 255) / 256), dim3(256), 0, st, dg, use_both, d.label, d.tmp);
    std::vector<double> w2c((size_t)N);
    GCHK(hipMemcpyAsync(w2c.data(), d.tmp, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, st));
    GCHK(hipEventRecord(g->ev1, st));
    GCHK(hipStreamSynchronize(st));
    std::vector<double> s_in((size_t)K, 0.0), s_tot((size_t)K, 0.0);
    for (int i = 0; i < N; i++) {
        s_tot[community[i]] += k[i];
        s_in[community[i]] += w2c[i];
    }
    double Q = 0.0;
    for (int c = 0; c < K; c++)
        if (s_tot[c] > 0)
            Q += s_in[c] / (2.0 * m) - resolution * (s_tot[c] / (2.0 * m)) * (s_tot[c] / (2.0 * m));
    float ms = 0;
    if (hipEventElapsedTime(&ms, g->ev0, g->ev1) == hipSuccess)
        g->stats.device_ms = ms;
    g->stats.n_communities = K;
    memcpy(community_out, community.data(), (size_t)N * sizeof(int));
    if (modularity_out)
        *modularity_out = Q;
    return 0;
}

extern "C" int mn_graph_leiden_stats(mn_graph *g, mn_leiden_stats *out) {
    *out = g->stats;
    return 0;
}
