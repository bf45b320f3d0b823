// mn_n2v.hip — Node2Vec biased walks + skip-gram negative sampling (src/node2vec.c:154-394,:486-551)
// on gfx950.
//
// The reference is ONE serial stochastic-gradient stream: a single xorshift32 state feeds the walk
// sampler and the negative sampler alike (:486,:514,:529), every pair updates the embedding rows in
// place, and every f32 dot product is a left-to-right chain over d (:372-375).
//   k_n2v_seq (MN_N2V_SEQUENTIAL)  one wavefront replays that stream exactly: lanes parallelise inside
//        a step (transition weights of the ≤deg neighbours, is_neighbor scans, the dim products, the
//        row updates) while every order-sensitive reduction (Σ weights in f64, the dot chain in f32)
//        is carried out in the reference's order.  Embedding rows are accessed with agent-scope
//        relaxed atomics (L2-served) because the wavefront re-reads rows it has just written.
//        Output bytes are identical to what the reference INSERTs into the output table.
//   k_n2v_normalize                L2 normalisation (:540-551), one wavefront per row, same chain order.
// Host side prepares exactly what the reference prepares serially before training: syn0 from the RNG
// stream (:323-325), the (deg+1)^0.75 negative table (:284-303, f64 pow) and the 1001-entry sigmoid
// LUT (:247-258, expf) — these are inputs of the hot loop, not part of it.
#include "../../include/muninn_hip.h"
#include "mn_guard.hpp"
#include <hip/hip_runtime.h>
#include <chrono>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#define DEVI __device__ __forceinline__
#define N2V_SIG_SIZE 1000
#define N2V_MAX_SIG 6.0f
#define N2V_NEG_TABLE 100000
#define N2V_LDS_DEG 2048
#define N2V_LDS_WALK 4096

DEVI unsigned xs32(unsigned &s) { // :27-34
    unsigned x = s;
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    s = x;
    return x;
}
DEVI double xs_rand(unsigned &s) { return (double)xs32(s) / (double)0xFFFFFFFFu; } // :36-38

DEVI float ldf(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// running-total array of one walk step: LDS when it fits, else global scratch through L2 (sc1)
DEVI double cum_ld(const double *lds, const double *glb, bool in_lds, int i) {
    return in_lds ? lds[i] : __hip_atomic_load(glb + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
DEVI void cum_st(double *lds, double *glb, bool in_lds, int i, double v) {
    if (in_lds)
        lds[i] = v;
    else
        __hip_atomic_store(glb + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
DEVI void stf(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct N2vArgs {
    int n;
    const int *off, *adj;
    float *syn0, *syn1neg;
    const int *neg_table;
    const float *sig_table;
    int dim, num_walks, walk_length, window, neg, epochs;
    double p, q, lr;
    unsigned rng;
    double *cum_scratch; // [max_deg] for nodes with more than N2V_LDS_DEG neighbours
    int *walk_scratch;   // [walk_length] when longer than N2V_LDS_WALK
    unsigned long long *out; // [0] pairs, [1] final rng
};

DEVI float fast_sigmoid(const float *tab, float x) { // :260-271
    if (x >= N2V_MAX_SIG)
        return 1.0f;
    if (x <= -N2V_MAX_SIG)
        return 0.0f;
    int idx = (int)((x + N2V_MAX_SIG) / (2.0f * N2V_MAX_SIG) * N2V_SIG_SIZE);
    if (idx < 0)
        idx = 0;
    if (idx > N2V_SIG_SIZE)
        idx = N2V_SIG_SIZE;
    return tab[idx];
}

__global__ void __launch_bounds__(64) k_n2v_seq(N2vArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    double *cum_l = reinterpret_cast<double *>(smem);            // [N2V_LDS_DEG]
    float *prod = reinterpret_cast<float *>(cum_l + N2V_LDS_DEG); // [dim]
    float *vc = prod + a.dim;                                     // [dim]
    float *neu = vc + a.dim;                                      // [dim]
    float *sig = neu + a.dim;                                     // [1001]
    int *walk_l = reinterpret_cast<int *>(sig + N2V_SIG_SIZE + 1);
    int *walk = walk_l; // LDS (walk_length <= N2V_LDS_WALK is enforced by the host)
    for (int i = lane; i <= N2V_SIG_SIZE; i += 64)
        sig[i] = a.sig_table[i];
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();

    unsigned rng = a.rng;
    const int dim = a.dim;
    const int total_words = a.n * a.num_walks * a.walk_length * a.epochs; // :503 — 32-bit, as the reference
    int word_count = 0;
    unsigned long long pairs = 0;
    for (int epoch = 0; epoch < a.epochs; epoch++)
        for (int w = 0; w < a.num_walks; w++)
            for (int n = 0; n < a.n; n++) {
                float lr = (float)(a.lr * (1.0 - (double)word_count / (double)total_words)); // :510-512
                if (lr < (float)(a.lr * 0.0001))
                    lr = (float)(a.lr * 0.0001);
                // ── biased_walk (:168-226) ──
                int wlen;
                {
                    if (lane == 0)
                        walk[0] = n;
                    const int s0 = a.off[n], deg0 = a.off[n + 1] - s0;
                    if (deg0 == 0) {
                        wlen = 1;
                    } else {
                        int idx = (int)(xs_rand(rng) * deg0);
                        if (idx >= deg0)
                            idx = deg0 - 1;
                        int cur = a.adj[s0 + idx], prev = n;
                        if (lane == 0)
                            walk[1] = cur;
                        wlen = a.walk_length;
                        for (int step = 2; step < a.walk_length; step++) {
                            const int c0 = a.off[cur], deg = a.off[cur + 1] - c0;
                            if (deg == 0) {
                                wlen = step;
                                break;
                            }
                            const bool in_lds = deg <= N2V_LDS_DEG;
                            const int p0 = a.off[prev], degp = a.off[prev + 1] - p0;
                            __builtin_amdgcn_wave_barrier();
                            for (int i = lane; i < deg; i += 64) { // transition weights, :186-195
                                const int x = a.adj[c0 + i];
                                double wt;
                                if (x == prev) {
                                    wt = 1.0 / a.p;
                                } else {
                                    bool nb = false;
                                    for (int j = 0; j < degp; j++)
                                        if (a.adj[p0 + j] == x) {
                                            nb = true;
                                            break;
                                        }
                                    wt = nb ? 1.0 : 1.0 / a.q;
                                }
                                cum_st(cum_l, a.cum_scratch, in_lds, i, wt);
                            }
                            __builtin_amdgcn_s_waitcnt(0);
                            __builtin_amdgcn_wave_barrier();
                            if (lane == 0) { // Σ in list order, f64 (:196, :213) → running totals
                                double t = 0.0;
                                for (int i = 0; i < deg; i++) {
                                    t += cum_ld(cum_l, a.cum_scratch, in_lds, i);
                                    cum_st(cum_l, a.cum_scratch, in_lds, i, t);
                                }
                            }
                            __builtin_amdgcn_s_waitcnt(0);
                            __builtin_amdgcn_wave_barrier();
                            const double total = cum_ld(cum_l, a.cum_scratch, in_lds, deg - 1);
                            const double r = xs_rand(rng) * total; // :200
                            int chosen_i = 0x7fffffff;
                            for (int i = lane; i < deg; i += 64)
                                if (r <= cum_ld(cum_l, a.cum_scratch, in_lds, i)) { // first i with r <= cumulative (:214-217)
                                    chosen_i = i;
                                    break;
                                }
                            for (int m = 32; m >= 1; m >>= 1) {
                                int o = __shfl_xor(chosen_i, m);
                                chosen_i = o < chosen_i ? o : chosen_i;
                            }
                            const int chosen = chosen_i == 0x7fffffff ? a.adj[c0] : a.adj[c0 + chosen_i]; // fallback :202
                            if (lane == 0)
                                walk[step] = chosen;
                            prev = cur;
                            cur = chosen;
                        }
                    }
                }
                __builtin_amdgcn_s_waitcnt(0);
                __builtin_amdgcn_wave_barrier();
                // ── skip-gram over the walk (:517-533) ──
                for (int pos = 0; pos < wlen; pos++) {
                    const int center = walk[pos];
                    int cs = pos - a.window, ce = pos + a.window;
                    if (cs < 0)
                        cs = 0;
                    if (ce >= wlen)
                        ce = wlen - 1;
                    float *rowc = a.syn0 + (size_t)center * dim;
                    for (int c = cs; c <= ce; c++) {
                        if (c == pos)
                            continue;
                        const int context = walk[c];
                        // sgns_train_pair (:345-394)
                        __builtin_amdgcn_wave_barrier();
                        for (int d = lane; d < dim; d += 64) {
                            vc[d] = ldf(rowc + d);
                            neu[d] = 0.0f;
                        }
                        for (int s = 0; s <= a.neg; s++) {
                            int target;
                            float label;
                            if (s == 0) {
                                target = context;
                                label = 1.0f;
                            } else {
                                target = a.neg_table[xs32(rng) % N2V_NEG_TABLE];
                                if (target == center || target == context)
                                    continue; // the draw is consumed first (:361-363)
                                label = 0.0f;
                            }
                            float *rowt = a.syn1neg + (size_t)target * dim;
                            __builtin_amdgcn_s_waitcnt(0);
                            __builtin_amdgcn_wave_barrier();
                            float vt_reg[16]; // dim <= 1024 → at most 16 elements per lane
                            for (int d = lane, r = 0; d < dim; d += 64, r++) {
                                vt_reg[r] = ldf(rowt + d);
                                prod[d] = __fmul_rn(vc[d], vt_reg[r]);
                            }
                            __builtin_amdgcn_s_waitcnt(0);
                            __builtin_amdgcn_wave_barrier();
                            float dot = 0.0f; // left-to-right chain (:372-375), every lane redundantly
                            for (int d = 0; d < dim; d++)
                                dot = __fadd_rn(dot, prod[d]);
                            const float sg = fast_sigmoid(sig, dot);
                            const float err = __fmul_rn(__fsub_rn(label, sg), lr);
                            for (int d = lane, r = 0; d < dim; d += 64, r++) {
                                neu[d] = __fadd_rn(neu[d], __fmul_rn(err, vt_reg[r]));  // :382-384
                                stf(rowt + d, __fadd_rn(vt_reg[r], __fmul_rn(err, vc[d]))); // :386-388
                            }
                        }
                        __builtin_amdgcn_s_waitcnt(0);
                        __builtin_amdgcn_wave_barrier();
                        for (int d = lane; d < dim; d += 64) // :391-393
                            stf(rowc + d, __fadd_rn(vc[d], neu[d]));
                        __builtin_amdgcn_s_waitcnt(0);
                        pairs++;
                    }
                    word_count++;
                }
            }
    if (lane == 0) {
        a.out[0] = pairs;
        a.out[1] = rng;
    }
}

// :540-551 — one wavefront per row; the Σ emb[d]² chain in d order
__global__ void __launch_bounds__(64) k_n2v_normalize(float *syn0, int n, int dim) {
    extern __shared__ float row[];
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= n)
        return;
    float *emb = syn0 + (size_t)i * dim;
    for (int d = lane; d < dim; d += 64)
        row[d] = emb[d];
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    float norm = 0.0f;
    for (int d = 0; d < dim; d++)
        norm = __fadd_rn(norm, __fmul_rn(row[d], row[d]));
    norm = sqrtf(norm);
    if (norm > 1e-10f)
        for (int d = lane; d < dim; d += 64)
            emb[d] = __fdiv_rn(row[d], norm);
}

// ───────────────────────── host ─────────────────────────

static thread_local std::string g_nerr;
static void nset_err(const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_nerr = buf;
}
extern "C" const char *mn_node2vec_last_error(void) { return g_nerr.c_str(); }

#define NCHK(expr)                                                                                 \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess) {                                                                   \
            nset_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return -1;                                                                             \
        }                                                                                          \
    } while (0)

static unsigned h_xs32(unsigned *s) {
    unsigned x = *s;
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    *s = x;
    return x;
}

#include "mn_n2v_batched.hpp"
#include "mn_comm.hpp"

static int valid_params(const mn_n2v_params *prm) { // src/node2vec.c:443-464
    return prm && prm->dim > 0 && prm->dim <= 1024 && prm->p > 0 && prm->q > 0 && prm->num_walks > 0 && prm->walk_length > 0 &&
           prm->window > 0 && prm->neg_samples > 0 && prm->learning_rate > 0 && prm->epochs > 0;
}

// uploads the graph and prepares what the reference prepares serially before its loop (sgns_create, :305-330)
static int session_init(mn_n2v_session *S, int n, const int *off, const int *adj, const mn_n2v_params *prm, int device,
                        bool batched) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        nset_err("node2vec: HIP device %d not available (no CPU fallback)", device);
        return -1;
    }
    NCHK(hipSetDevice(device));
    S->device = device;
    const int dim = prm->dim;
    const size_t nd = (size_t)n * dim;
    unsigned rng = 42; // :486
    std::vector<float> syn0(nd);
    for (size_t i = 0; i < nd; i++) // :323-325
        syn0[i] = ((float)((double)h_xs32(&rng) / (double)0xFFFFFFFFu) - 0.5f) / (float)dim;
    std::vector<int> neg(N2V_NEG_TABLE);
    {
        double total = 0.0; // :284-303
        for (int i = 0; i < n; i++)
            total += pow((double)(off[i + 1] - off[i] + 1), 0.75);
        int idx = 0;
        double cum = 0.0;
        for (int i = 0; i < n && idx < N2V_NEG_TABLE; i++) {
            cum += pow((double)(off[i + 1] - off[i] + 1), 0.75) / total;
            while (idx < N2V_NEG_TABLE && (double)idx / N2V_NEG_TABLE < cum)
                neg[idx++] = i;
        }
        while (idx < N2V_NEG_TABLE)
            neg[idx++] = n - 1;
    }
    std::vector<float> sig(N2V_SIG_SIZE + 1);
    for (int i = 0; i <= N2V_SIG_SIZE; i++) { // :247-258
        float x = (float)i / (float)N2V_SIG_SIZE * 2.0f * N2V_MAX_SIG - N2V_MAX_SIG;
        sig[i] = 1.0f / (1.0f + expf(-x));
    }
    S->max_deg = 0;
    for (int i = 0; i < n; i++)
        S->max_deg = std::max(S->max_deg, off[i + 1] - off[i]);
    const size_t ne = (size_t)off[n];
    NCHK(hipMalloc(&S->off, ((size_t)n + 1) * sizeof(int)));
    NCHK(hipMalloc(&S->adj, std::max<size_t>(1, ne) * sizeof(int)));
    NCHK(hipMalloc(&S->neg, N2V_NEG_TABLE * sizeof(int)));
    NCHK(hipMalloc(&S->sig, (N2V_SIG_SIZE + 1) * sizeof(float)));
    // (+ 64 rows: the data-parallel mode all-gathers equal row shards of up to 64 ranks in place, ceil(n / world) * world rows)
    NCHK(hipMalloc(&S->syn0, (nd + (size_t)64 * dim) * sizeof(float)));
    NCHK(hipMalloc(&S->syn1, (nd + (size_t)64 * dim) * sizeof(float)));
    NCHK(hipMalloc(&S->pairs, 2 * sizeof(unsigned long long)));
    NCHK(hipMemcpy(S->off, off, ((size_t)n + 1) * sizeof(int), hipMemcpyHostToDevice));
    if (ne)
        NCHK(hipMemcpy(S->adj, adj, ne * sizeof(int), hipMemcpyHostToDevice));
    NCHK(hipMemcpy(S->neg, neg.data(), N2V_NEG_TABLE * sizeof(int), hipMemcpyHostToDevice));
    NCHK(hipMemcpy(S->sig, sig.data(), sig.size() * sizeof(float), hipMemcpyHostToDevice));
    NCHK(hipMemcpy(S->syn0, syn0.data(), nd * sizeof(float), hipMemcpyHostToDevice));
    NCHK(hipStreamCreateWithFlags(&S->st, hipStreamNonBlocking));
    NCHK(hipMemsetAsync(S->syn1, 0, nd * sizeof(float), S->st));
    NCHK(hipMemsetAsync(S->pairs, 0, 2 * sizeof(unsigned long long), S->st));
    NCHK(hipEventCreate(&S->e0));
    NCHK(hipEventCreate(&S->e1));
    N2vArgs &a = S->a;
    memset(&a, 0, sizeof(a));
    a.n = n;
    a.off = S->off;
    a.adj = S->adj;
    a.syn0 = S->syn0;
    a.syn1neg = S->syn1;
    a.neg_table = S->neg;
    a.sig_table = S->sig;
    a.dim = dim;
    a.num_walks = prm->num_walks;
    a.walk_length = prm->walk_length;
    a.window = prm->window;
    a.neg = prm->neg_samples;
    a.epochs = prm->epochs;
    a.p = prm->p;
    a.q = prm->q;
    a.lr = prm->learning_rate;
    a.rng = rng;
    a.out = S->pairs;
    if (!batched) {
        NCHK(hipMalloc(&S->cum, (size_t)std::max(1, S->max_deg) * sizeof(double)));
        NCHK(hipMalloc(&S->s_center, (size_t)prm->walk_length * sizeof(int))); // walk scratch
        a.cum_scratch = S->cum;
        a.walk_scratch = S->s_center;
        return 0;
    }
    int B = prm->batch_walks;
    if (B <= 0) // ~75 contributions per row and batch at the default walk/window/neg settings
        B = std::min(16384, std::max(1, n / 64));
    B = std::min(B, n);
    S->B = B;
    S->cap = prm->walk_length * 2 * prm->window * (1 + prm->neg_samples);
    S->ns_max = (size_t)(B + 64) * S->cap; // + padding slots when a batch is split over up to 64 ranks
    if (S->ns_max > 0x7fffffffULL) {
        nset_err("node2vec: batch of %d walks x %d samples exceeds 2^31", B, S->cap);
        return -1;
    }
    NCHK(hipMalloc(&S->s_center, S->ns_max * sizeof(int)));
    NCHK(hipMalloc(&S->s_target, S->ns_max * sizeof(int)));
    NCHK(hipMalloc(&S->s_err, S->ns_max * sizeof(float)));
    S->np_max = (size_t)(B + 64) * prm->walk_length;
    NCHK(hipMalloc(&S->p_center, S->np_max * sizeof(int)));
    NCHK(hipMalloc(&S->p_neu, S->np_max * dim * sizeof(float)));
    NCHK(hipMalloc(&S->keys, S->ns_max * sizeof(int)));
    NCHK(hipMalloc(&S->vals, S->ns_max * sizeof(int)));
    NCHK(hipMalloc(&S->keys_s, S->ns_max * sizeof(int)));
    NCHK(hipMalloc(&S->vals_s, S->ns_max * sizeof(int)));
    NCHK(hipMalloc(&S->keys_c, S->ns_max * sizeof(int)));
    NCHK(hipMalloc(&S->seg, ((size_t)n + 1) * sizeof(int)));
    NCHK(hipMalloc(&S->seg_c, ((size_t)n + 1) * sizeof(int)));
    NCHK(hipMalloc(&S->nseg, sizeof(int)));
    NCHK(hipMalloc(&S->nseg_c, sizeof(int)));
    NCHK(hipMalloc(&S->staged, nd * sizeof(float)));
    if (S->max_deg > N2VB_LDS_DEG && !(a.p == 1.0 && a.q == 1.0))
        NCHK(hipMalloc(&S->cum, (size_t)B * S->max_deg * sizeof(double)));
    S->bits = 1;
    while ((1u << S->bits) <= (unsigned)n)
        S->bits++;
    if (rocprim::radix_sort_pairs(nullptr, S->tmp_bytes, S->keys, S->keys_s, S->vals, S->vals_s, S->ns_max, 0, S->bits, S->st) !=
        hipSuccess) {
        nset_err("rocprim::radix_sort_pairs (size query) failed");
        return -1;
    }
    NCHK(hipMalloc(&S->tmp, S->tmp_bytes));
    return 0;
}

extern "C" mn_n2v_session *mn_n2v_begin(int n, const int *off, const int *adj, const mn_n2v_params *prm, int device) try {
    if (n <= 0 || !valid_params(prm)) {
        nset_err("mn_n2v_begin: invalid parameters");
        return nullptr;
    }
    mn_n2v_session *S = new mn_n2v_session();
    if (session_init(S, n, off, adj, prm, device, true) != 0) {
        delete S;
        return nullptr;
    }
    (void)hipEventRecord(S->e0, S->st);
    return S;
} MN_GUARD_END(nset_err, MN_NOTHING, nullptr)
extern "C" int mn_n2v_batch_walks(mn_n2v_session *S) { return S->B; }
extern "C" int mn_n2v_sample_slots(mn_n2v_session *S) { return S->cap; }
extern "C" int mn_n2v_position_slots(mn_n2v_session *S) { return S->a.walk_length; }
extern "C" int mn_n2v_samples(mn_n2v_session *S, int epoch, int w, int lo, int hi, int *d_center, int *d_target, float *d_err,
                              int *d_pcenter, float *d_pneu) try {
    NCHK(hipSetDevice(S->device));
    if (lo < 0 || hi > S->a.n || hi <= lo || hi - lo > S->B) {
        nset_err("mn_n2v_samples: bad walk range [%d, %d)", lo, hi);
        return -1;
    }
    return n2v_samples(S, epoch, w, lo, hi, d_center, d_target, d_err, d_pcenter, d_pneu);
} MN_GUARD_END(nset_err, MN_NOTHING, -1)
extern "C" int mn_n2v_apply(mn_n2v_session *S, const int *d_center, const int *d_target, const float *d_err, int64_t ns,
                            const int *d_pcenter, const float *d_pneu, int64_t np) try {
    NCHK(hipSetDevice(S->device));
    return n2v_apply(S, d_center, d_target, d_err, ns, d_pcenter, d_pneu, np);
} MN_GUARD_END(nset_err, MN_NOTHING, -1)
extern "C" int mn_n2v_sync(mn_n2v_session *S) try {
    NCHK(hipSetDevice(S->device));
    NCHK(hipDeviceSynchronize());
    return 0;
} MN_GUARD_END(nset_err, MN_NOTHING, -1)
extern "C" int mn_n2v_finish(mn_n2v_session *S, float *out, mn_n2v_stats *stats) try {
    NCHK(hipSetDevice(S->device));
    const int n = S->a.n, dim = S->a.dim;
    hipLaunchKernelGGL(k_n2v_normalize, dim3(n), dim3(64), (size_t)dim * sizeof(float), S->st, S->syn0, n, dim);
    NCHK(hipEventRecord(S->e1, S->st));
    NCHK(hipDeviceSynchronize());
    unsigned long long o[2];
    NCHK(hipMemcpy(o, S->pairs, sizeof(o), hipMemcpyDeviceToHost));
    NCHK(hipMemcpy(out, S->syn0, (size_t)n * dim * sizeof(float), hipMemcpyDeviceToHost));
    if (stats) {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, S->e0, S->e1);
        stats->pairs = (int64_t)o[0];
        stats->device_ms = ms;
    }
    return n;
} MN_GUARD_END(nset_err, MN_NOTHING, -1)
// the same, with the normalised embeddings left where they are: *d_out = [n][dim] f32 in HBM on the session's device, valid
// until mn_n2v_end (config 4's "-> hnsw index" leg feeds them to mn_hnsw_build_dev without a host round trip)
extern "C" int mn_n2v_finish_dev(mn_n2v_session *S, const float **d_out, mn_n2v_stats *stats) try {
    NCHK(hipSetDevice(S->device));
    const int n = S->a.n, dim = S->a.dim;
    hipLaunchKernelGGL(k_n2v_normalize, dim3(n), dim3(64), (size_t)dim * sizeof(float), S->st, S->syn0, n, dim);
    NCHK(hipEventRecord(S->e1, S->st));
    NCHK(hipStreamSynchronize(S->st));
    unsigned long long o[2];
    NCHK(hipMemcpy(o, S->pairs, sizeof(o), hipMemcpyDeviceToHost));
    *d_out = S->syn0;
    if (stats) {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, S->e0, S->e1);
        stats->pairs = (int64_t)o[0];
        stats->device_ms = ms;
    }
    return n;
} MN_GUARD_END(nset_err, MN_NOTHING, -1)

extern "C" void mn_n2v_end(mn_n2v_session *S) {
    if (!S)
        return;
    (void)hipSetDevice(S->device);
    (void)hipDeviceSynchronize();
    delete S;
}

// MN_N2V_BATCHED over several GPUs (config 4), see muninn_hip.h.  The slices are contiguous and gathered in rank order, so
// the sample order every replica applies is the single-GPU order.
extern "C" int mn_node2vec_train_shared(mn_comm *c, int n, const int *off, const int *adj, const mn_n2v_params *prm, int device,
                                        float *out, mn_n2v_stats *stats) try {
    if (stats)
        memset(stats, 0, sizeof(*stats));
    if (n == 0)
        return 0;
    if (!valid_params(prm)) {
        nset_err("mn_node2vec_train_shared: invalid parameters (src/node2vec.c:443-464)");
        return -1;
    }
    const int world = c ? c->world : 1, rank = c ? c->rank : 0;
    mn_n2v_session *S = mn_n2v_begin(n, off, adj, prm, device);
    if (!S)
        return -1;
    const int B = S->B, cap = S->cap, pcap = prm->walk_length, dim = prm->dim;
    const int per_max = (B + world - 1) / world;
    const int rows_per = (n + world - 1) / world; // destination rows per rank (equal shards: the last one is padded)
    if (world > 64) {
        nset_err("mn_node2vec_train_shared: more than 64 ranks");
        mn_n2v_end(S);
        return -1;
    }
    int *lc = nullptr, *lt = nullptr, *lpc = nullptr, *gc = nullptr, *gt = nullptr, *gpc = nullptr, *send_s = nullptr, *recv_s = nullptr,
        *d_hist = nullptr;
    float *le = nullptr, *lpn = nullptr, *ge = nullptr, *gpn = nullptr, *send_p = nullptr, *recv_p = nullptr;
    long long *d_cnt = nullptr;
    auto cleanup = [&](int rc) {
        (void)hipDeviceSynchronize();
        void *ps[] = {lc, lt, lpc, gc, gt, gpc, le, lpn, ge, gpn, send_s, recv_s, send_p, recv_p, d_hist, d_cnt};
        for (void *q : ps)
            (void)hipFree(q);
        mn_n2v_end(S);
        return rc;
    };
    const size_t ls = (size_t)per_max * cap, lp = (size_t)per_max * pcap;
    bool ok = hipMalloc(&lc, ls * 4) == hipSuccess && hipMalloc(&lt, ls * 4) == hipSuccess && hipMalloc(&le, ls * 4) == hipSuccess &&
              hipMalloc(&lpc, lp * 4) == hipSuccess && hipMalloc(&lpn, lp * dim * 4) == hipSuccess;
    if (ok && world > 1) // the exchange by destination shard: packed buckets out, packed buckets in, then the unpacked arrays
        ok = hipMalloc(&send_s, ls * 12) == hipSuccess && hipMalloc(&recv_s, ls * world * 12) == hipSuccess &&
             hipMalloc(&send_p, lp * (dim + 1) * 4) == hipSuccess && hipMalloc(&recv_p, lp * world * (dim + 1) * 4) == hipSuccess &&
             hipMalloc(&gc, ls * world * 4) == hipSuccess && hipMalloc(&gt, ls * world * 4) == hipSuccess &&
             hipMalloc(&ge, ls * world * 4) == hipSuccess && hipMalloc(&gpc, lp * world * 4) == hipSuccess &&
             hipMalloc(&gpn, lp * world * dim * 4) == hipSuccess && hipMalloc(&d_hist, 2 * 65 * sizeof(int)) == hipSuccess &&
             hipMalloc(&d_cnt, (size_t)(world + 1) * 2 * world * sizeof(long long)) == hipSuccess;
    if (!ok) {
        nset_err("mn_node2vec_train_shared: out of device memory for the exchange buffers");
        return cleanup(-1);
    }
    int shard_bits = 1;
    while ((1 << shard_bits) <= world)
        shard_bits++;
    if (world > 1) { // the bucket sort's scratch (fewer key bits than the row sorts: its own size query)
        size_t need = 0;
        if (rocprim::radix_sort_pairs(nullptr, need, S->keys, S->keys_s, S->vals, S->vals_s, ls, 0, shard_bits, S->st) != hipSuccess) {
            nset_err("rocprim::radix_sort_pairs (size query) failed");
            return cleanup(-1);
        }
        if (need > S->tmp_bytes) {
            (void)hipFree(S->tmp);
            S->tmp = nullptr;
            if (hipMalloc(&S->tmp, need) != hipSuccess) {
                nset_err("mn_node2vec_train_shared: out of device memory for the sort scratch");
                return cleanup(-1);
            }
            S->tmp_bytes = need;
        }
    }
    std::vector<int> h_hist(2 * 65);
    std::vector<long long> h_cnt((size_t)2 * world * world), h_row((size_t)2 * world);
    // A rank whose local step fails may not leave the loop on its own: its peers are in — or about to enter — the batch's
    // all-gathers and would wait for ever.  `status` carries a local failure (this batch's samples, or the previous batch's
    // apply) into the status exchange at the head of the next collective; all ranks stop together (mn_comm_agree).
    int rc = 0, status = 0;
    std::string mine;
    auto agree = [&]() {
        if (world == 1) {
            if (status)
                rc = -1;
            return;
        }
        int failed = -1;
        const int ag = mn_comm_agree(c, status, S->st, &failed);
        if (ag == 0)
            return;
        if (ag < 0)
            nset_err("mn_node2vec_train_shared: %s", mn_comm_last_error_str());
        else if (failed == rank)
            nset_err("mn_node2vec_train_shared: rank %d failed: %s", rank, mine.c_str());
        else
            nset_err("mn_node2vec_train_shared: rank %d failed; all ranks stop", failed);
        rc = -1;
    };
    for (int epoch = 0; epoch < prm->epochs && rc == 0; epoch++)
        for (int w = 0; w < prm->num_walks && rc == 0; w++)
            for (int b0 = 0; b0 < n && rc == 0; b0 += B) {
                const int b1 = std::min(n, b0 + B);
                const int per = (b1 - b0 + world - 1) / world;
                const int lo = std::min(b1, b0 + rank * per), hi = std::min(b1, lo + per);
                if (!status) {
                    // unused slots carry -1 (ranks with a short or empty slice still contribute `per` walks' worth of slots)
                    if (hipMemsetAsync(lc, 0xFF, (size_t)per * cap * 4, S->st) != hipSuccess ||
                        hipMemsetAsync(lt, 0xFF, (size_t)per * cap * 4, S->st) != hipSuccess ||
                        hipMemsetAsync(le, 0, (size_t)per * cap * 4, S->st) != hipSuccess ||
                        hipMemsetAsync(lpc, 0xFF, (size_t)per * pcap * 4, S->st) != hipSuccess) {
                        nset_err("mn_node2vec_train_shared: memset failed");
                        status = 1;
                    } else if (hi > lo && n2v_samples(S, epoch, w, lo, hi, lc, lt, le, lpc, lpn)) {
                        status = 1;
                    }
                    if (const char *fi = getenv("MN_FAULT_INJECT")) { // test hook: "n2v_shared:<rank>:<first node of the batch>"
                        int fr = -1, fb = -1;
                        if (sscanf(fi, "n2v_shared:%d:%d", &fr, &fb) == 2 && fr == rank && fb == b0 && !status) {
                            nset_err("injected failure (MN_FAULT_INJECT)");
                            status = 1;
                        }
                    }
                    if (status)
                        mine = mn_node2vec_last_error();
                }
                agree();
                if (rc)
                    break;
                const int ls_n = per * cap, lp_n = per * pcap;
                if (world == 1) {
                    if (n2v_apply(S, lc, lt, le, ls_n, lpc, lpn, lp_n, 0, n)) {
                        status = 1;
                        mine = mn_node2vec_last_error();
                    }
                    continue;
                }
                // ── exchange by destination shard: a sample goes to the rank that owns its target row, a position to the one that
                //    owns its centre row; each rank then sorts and applies only what is meant for its rows ──
                bool xok = hipMemsetAsync(d_hist, 0, 2 * 65 * sizeof(int), S->st) == hipSuccess;
                size_t tb = S->tmp_bytes;
                if (xok) {
                    hipLaunchKernelGGL(k_n2v_shard_keys, dim3((ls_n + 255) / 256), dim3(256), 0, S->st, lt, ls_n, rows_per, world, S->keys,
                                       S->vals, d_hist);
                    xok = rocprim::radix_sort_pairs(S->tmp, tb, S->keys, S->keys_s, S->vals, S->vals_s, (size_t)ls_n, 0, shard_bits, S->st) ==
                          hipSuccess;
                }
                if (xok) {
                    xok = hipMemcpyAsync(h_hist.data(), d_hist, 2 * 65 * sizeof(int), hipMemcpyDeviceToHost, S->st) == hipSuccess &&
                          hipStreamSynchronize(S->st) == hipSuccess;
                }
                int nv_s = 0;
                if (xok) {
                    nv_s = ls_n - h_hist[(size_t)world];
                    if (nv_s > 0)
                        hipLaunchKernelGGL(k_n2v_pack_samples, dim3((nv_s + 255) / 256), dim3(256), 0, S->st, S->vals_s, nv_s, lc, lt, le, send_s);
                    hipLaunchKernelGGL(k_n2v_shard_keys, dim3((lp_n + 255) / 256), dim3(256), 0, S->st, lpc, lp_n, rows_per, world, S->keys,
                                       S->vals, d_hist + 65);
                    tb = S->tmp_bytes;
                    xok = rocprim::radix_sort_pairs(S->tmp, tb, S->keys, S->keys_c, S->vals, S->vals_s, (size_t)lp_n, 0, shard_bits, S->st) ==
                              hipSuccess &&
                          hipMemcpyAsync(h_hist.data() + 65, d_hist + 65, 65 * sizeof(int), hipMemcpyDeviceToHost, S->st) == hipSuccess &&
                          hipStreamSynchronize(S->st) == hipSuccess;
                }
                int nv_p = 0;
                if (xok) {
                    nv_p = lp_n - h_hist[(size_t)65 + world];
                    if (nv_p > 0)
                        hipLaunchKernelGGL(k_n2v_pack_positions, dim3(nv_p), dim3(64), 0, S->st, S->vals_s, lpc, lpn, dim, send_p);
                    xok = hipGetLastError() == hipSuccess;
                }
                // the bucket sizes of every rank; a rank whose preparation failed says so in its row (-1) instead of staying
                // away from the collective its peers are about to enter
                for (int p = 0; p < world; p++) {
                    h_row[(size_t)p] = xok ? h_hist[(size_t)p] : -1;
                    h_row[(size_t)world + p] = xok ? h_hist[(size_t)65 + p] : -1;
                }
                long long *d_row = d_cnt + (size_t)2 * world * world; // (send and receive areas do not overlap)
                if (hipMemcpyAsync(d_row, h_row.data(), (size_t)2 * world * sizeof(long long), hipMemcpyHostToDevice, S->st) != hipSuccess ||
                    hipStreamSynchronize(S->st) != hipSuccess ||
                    mn_comm_allgather_dev(c, d_row, d_cnt, (size_t)2 * world * sizeof(long long), S->st) ||
                    hipMemcpyAsync(h_cnt.data(), d_cnt, (size_t)2 * world * world * sizeof(long long), hipMemcpyDeviceToHost, S->st) != hipSuccess ||
                    hipStreamSynchronize(S->st) != hipSuccess) {
                    nset_err("mn_node2vec_train_shared: exchanging the bucket sizes failed (%s)", mn_comm_last_error_str());
                    rc = -1;
                    break;
                }
                int bad = -1;
                for (int sr = 0; sr < world && bad < 0; sr++)
                    if (h_cnt[(size_t)sr * 2 * world] < 0)
                        bad = sr;
                if (bad >= 0) {
                    nset_err("mn_node2vec_train_shared: rank %d could not prepare its buckets; all ranks stop", bad);
                    rc = -1;
                    break;
                }
                std::vector<long long> cs((size_t)world * world), cp((size_t)world * world);
                long long ns_recv = 0, np_recv = 0;
                for (int sr = 0; sr < world; sr++)
                    for (int p = 0; p < world; p++) {
                        cs[(size_t)sr * world + p] = h_cnt[(size_t)sr * 2 * world + p];
                        cp[(size_t)sr * world + p] = h_cnt[(size_t)sr * 2 * world + world + p];
                        if (p == rank) {
                            ns_recv += cs[(size_t)sr * world + p];
                            np_recv += cp[(size_t)sr * world + p];
                        }
                    }
                if (mn_comm_alltoallv_dev(c, send_s, recv_s, cs.data(), 12, S->st) ||
                    mn_comm_alltoallv_dev(c, send_p, recv_p, cp.data(), (size_t)(dim + 1) * 4, S->st)) {
                    nset_err("mn_node2vec_train_shared: %s", mn_comm_last_error_str());
                    rc = -1;
                    break;
                }
                if (ns_recv > 0)
                    hipLaunchKernelGGL(k_n2v_unpack_samples, dim3((unsigned)((ns_recv + 255) / 256)), dim3(256), 0, S->st, recv_s, (int)ns_recv,
                                       gc, gt, ge);
                if (np_recv > 0)
                    hipLaunchKernelGGL(k_n2v_unpack_positions, dim3((unsigned)np_recv), dim3(64), 0, S->st, recv_p, dim, gpc, gpn);
                // this rank's rows receive exactly the additions, in exactly the order, one GPU gives them; the updated shards of
                // both matrices are all-gathered in place
                if (n2v_apply(S, gc, gt, ge, ns_recv, gpc, gpn, np_recv, rank * rows_per, (rank + 1) * rows_per)) {
                    status = 1; // (reported to the peers at the head of the next batch, or below after the last one)
                    mine = mn_node2vec_last_error();
                }
                if (world > 1 &&
                    (mn_comm_allgather_dev(c, S->syn1 + (size_t)rank * rows_per * dim, S->syn1, (size_t)rows_per * dim * 4, S->st) ||
                     mn_comm_allgather_dev(c, S->syn0 + (size_t)rank * rows_per * dim, S->syn0, (size_t)rows_per * dim * 4, S->st))) {
                    nset_err("mn_node2vec_train_shared: %s", mn_comm_last_error_str());
                    rc = -1;
                    break;
                }
            }
    if (rc == 0)
        agree();
    if (rc == 0)
        rc = mn_n2v_finish(S, out, stats);
    return cleanup(rc < 0 ? -1 : n);
} MN_GUARD_END(nset_err, MN_NOTHING, -1)

// node2vec_train's compute followed by its output step (src/node2vec.c:540-583: INSERT every embedding into the output
// hnsw_index) with the embeddings never leaving HBM: trained and normalised on the index's device, then handed to
// mn_hnsw_build_dev with rowids first_rowid + i (the reference's rowid = first-seen index + 1).  host_out (or NULL) also
// receives the embeddings — one bulk copy, for a host that persists them (the extension's "{t}_nodes" shadow table).
// Same embedding bytes as mn_node2vec_train, same graph as mn_hnsw_build on them.  Returns n, or -1.
extern "C" int mn_node2vec_train_into(int n, const int *off, const int *adj, const mn_n2v_params *prm, int mode, mn_index *idx,
                                      int64_t first_rowid, float *host_out, mn_n2v_stats *stats, double *build_seconds) try {
    if (stats)
        memset(stats, 0, sizeof(*stats));
    if (build_seconds)
        *build_seconds = 0.0;
    if (n == 0)
        return 0;
    if (!valid_params(prm) || !idx) {
        nset_err("mn_node2vec_train_into: invalid parameters (src/node2vec.c:443-464)");
        return -1;
    }
    if (mode != MN_N2V_BATCHED) {
        nset_err("mn_node2vec_train_into: MN_N2V_BATCHED only (the serial stream is latency-bound: nothing to gain from HBM residency)");
        return -1;
    }
    const int device = mn_hnsw_device(idx);
    mn_n2v_session *S = mn_n2v_begin(n, off, adj, prm, device);
    if (!S)
        return -1;
    int rc = 0;
    for (int epoch = 0; epoch < prm->epochs && rc == 0; epoch++)
        for (int w = 0; w < prm->num_walks && rc == 0; w++)
            for (int b0 = 0; b0 < n && rc == 0; b0 += S->B) {
                const int b1 = std::min(n, b0 + S->B);
                rc = n2v_samples(S, epoch, w, b0, b1, S->s_center, S->s_target, S->s_err, S->p_center, S->p_neu);
                if (rc == 0)
                    rc = n2v_apply(S, S->s_center, S->s_target, S->s_err, (int64_t)(b1 - b0) * S->cap, S->p_center, S->p_neu,
                                   (int64_t)(b1 - b0) * prm->walk_length);
            }
    const float *d_emb = nullptr;
    if (rc == 0 && mn_n2v_finish_dev(S, &d_emb, stats) < 0)
        rc = -1;
    if (rc == 0) {
        std::vector<int64_t> ids((size_t)n);
        for (int i = 0; i < n; i++)
            ids[(size_t)i] = first_rowid + i;
        const auto t0 = std::chrono::steady_clock::now();
        if (mn_hnsw_build_dev(idx, ids.data(), d_emb, n, 0, 0) != 0 || mn_hnsw_sync(idx) != 0) {
            nset_err("mn_node2vec_train_into: %s", mn_last_error());
            rc = -1;
        }
        if (build_seconds)
            *build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    if (rc == 0 && host_out && hipMemcpy(host_out, d_emb, (size_t)n * prm->dim * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) {
        nset_err("mn_node2vec_train_into: download of the embeddings failed");
        rc = -1;
    }
    mn_n2v_end(S);
    return rc < 0 ? -1 : n;
} MN_GUARD_END(nset_err, MN_NOTHING, -1)

extern "C" int mn_node2vec_train(int n, const int *off, const int *adj, const mn_n2v_params *prm, int mode, int device,
                                 float *out, mn_n2v_stats *stats) try {
    if (stats)
        memset(stats, 0, sizeof(*stats));
    if (n == 0)
        return 0;
    if (!valid_params(prm)) {
        nset_err("mn_node2vec_train: invalid parameters (src/node2vec.c:443-464)");
        return -1;
    }
    if (mode == MN_N2V_BATCHED) {
        mn_n2v_session *S = mn_n2v_begin(n, off, adj, prm, device);
        if (!S)
            return -1;
        int rc = 0;
        for (int epoch = 0; epoch < prm->epochs && rc == 0; epoch++)
            for (int w = 0; w < prm->num_walks && rc == 0; w++)
                for (int b0 = 0; b0 < n && rc == 0; b0 += S->B) {
                    int b1 = std::min(n, b0 + S->B);
                    rc = n2v_samples(S, epoch, w, b0, b1, S->s_center, S->s_target, S->s_err, S->p_center, S->p_neu);
                    if (rc == 0)
                        rc = n2v_apply(S, S->s_center, S->s_target, S->s_err, (int64_t)(b1 - b0) * S->cap, S->p_center, S->p_neu,
                                       (int64_t)(b1 - b0) * prm->walk_length);
                }
        if (rc == 0)
            rc = mn_n2v_finish(S, out, stats);
        mn_n2v_end(S);
        return rc < 0 ? -1 : n;
    }
    if (prm->walk_length > N2V_LDS_WALK) {
        nset_err("mn_node2vec_train: walk_length %d exceeds %d", prm->walk_length, N2V_LDS_WALK);
        return -1;
    }
    if (mode != MN_N2V_SEQUENTIAL) {
        nset_err("mn_node2vec_train: mode %d not available", mode);
        return -1;
    }
    mn_n2v_session sess;
    mn_n2v_session *S = &sess;
    if (session_init(S, n, off, adj, prm, device, false) != 0)
        return -1;
    const int dim = prm->dim;
    size_t lds = N2V_LDS_DEG * sizeof(double) + 3 * (size_t)dim * sizeof(float) + (N2V_SIG_SIZE + 1) * sizeof(float) +
                 (size_t)std::min(prm->walk_length, N2V_LDS_WALK) * sizeof(int) + 64;
    NCHK(hipEventRecord(S->e0, S->st));
    hipLaunchKernelGGL(k_n2v_seq, dim3(1), dim3(64), lds, S->st, S->a);
    NCHK(hipGetLastError());
    return mn_n2v_finish(S, out, stats) < 0 ? -1 : n;
} MN_GUARD_END(nset_err, MN_NOTHING, -1)
