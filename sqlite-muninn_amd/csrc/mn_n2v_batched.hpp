// mn_n2v_batched.hpp — MN_N2V_BATCHED: the batch-synchronous Node2Vec schedule (included by mn_n2v.hip).
//
// Per (epoch, w) pass the start nodes are cut into batches of B walks (DESIGN.md §node2vec;
// oracle/mn_graph_oracle.c orc_node2vec_train_batched restates it):
//   k_n2v_walk_grad  one wavefront per walk: its own xorshift32 stream (seeded from epoch, w, n) draws the
//                    biased walk and then the negatives of its pairs in pair order; every sample's error is
//                    taken against the matrices frozen at batch start (centre row in registers, the target
//                    rows of one pair fetched together, lane-strided fmaf + xor butterfly, the reference's
//                    sigmoid LUT).  While a target row is in registers its contribution err·row is folded
//                    into the position's neu1e (src/node2vec.c:347,:383-385), written once per position.
//   sort             rocPRIM stable radix sort by destination row (plumbing): positions by centre,
//                    samples by target
//   k_n2v_apply_centers  one wavefront per centre row: staged = row + Σ neu1e of its positions, in walk order
//   k_n2v_apply      one wavefront per target row: row += Σ err · centre_row(old) in sample order (fmaf);
//                    then k_n2v_commit copies the staged centre rows back
// HBM traffic per pair: (1+neg) target rows for the dots + (1+neg) centre rows for the target updates; the
// centre-side update costs one row per walk position instead of (1+neg) rows per pair.
// No float atomics anywhere: embeddings are bit-reproducible and equal the CPU restatement's.
#pragma once
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#define N2VB_LDS_DEG 512

struct N2vBatchArgs {
    N2vArgs a;
    int epoch, w, b0, b1;
    int split; // wavefronts per walk
    double total_words;
    int cap; // sample slots per walk
    int *s_center, *s_target;
    float *s_err;
    int *p_center; // [walks][walk_length] centre of each walk position, -1 past the walk's end
    float *p_neu;  // [walks][walk_length][dim] Σ err · target_row(old) over the position's samples
    double *cum_scratch; // [walks][max_deg], used when a node has more than N2VB_LDS_DEG neighbours
    int max_deg;
    unsigned long long *pairs_out;
};

DEVI unsigned n2v_walk_seed(int epoch, int w, int n) {
    unsigned s = 42u ^ ((unsigned)epoch * 0x9E3779B9u) ^ ((unsigned)w * 0x85EBCA6Bu) ^ ((unsigned)n * 0xC2B2AE35u);
    s ^= s >> 15;
    s *= 0x2C1B3C6Du;
    s ^= s >> 12;
    return s ? s : 1u;
}

// biased_walk (src/node2vec.c:168-226) for one wavefront; walk[] in LDS.  With p == q == 1 every weight is
// 1.0, the running totals are the exact integers 1..deg and "first i with r <= i+1" is ceil(r)-1: the
// is_neighbor scans are skipped with identical results.
DEVI int gen_walk(const N2vArgs &a, int n, unsigned &rng, int *walk, double *cum_l, double *cum_g, int lane) {
    if (lane == 0)
        walk[0] = n;
    const int s0 = a.off[n], deg0 = a.off[n + 1] - s0;
    if (deg0 == 0)
        return 1;
    int idx = (int)(xs_rand(rng) * deg0);
    if (idx >= deg0)
        idx = deg0 - 1;
    int cur = a.adj[s0 + idx], prev = n;
    if (lane == 0)
        walk[1] = cur;
    const bool uniform = a.p == 1.0 && a.q == 1.0;
    for (int step = 2; step < a.walk_length; step++) {
        const int c0 = a.off[cur], deg = a.off[cur + 1] - c0;
        if (deg == 0)
            return step;
        int chosen;
        if (uniform) {
            const double r = xs_rand(rng) * (double)deg;
            int i = (int)ceil(r) - 1;
            if (i < 0)
                i = 0;
            if (i >= deg)
                i = deg - 1;
            chosen = a.adj[c0 + i];
        } else {
            const bool in_lds = deg <= N2VB_LDS_DEG;
            const int p0 = a.off[prev], degp = a.off[prev + 1] - p0;
            __builtin_amdgcn_wave_barrier();
            for (int i = lane; i < deg; i += 64) {
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
                cum_st(cum_l, cum_g, in_lds, i, wt);
            }
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
                double t = 0.0;
                for (int i = 0; i < deg; i++) {
                    t += cum_ld(cum_l, cum_g, in_lds, i);
                    cum_st(cum_l, cum_g, in_lds, i, t);
                }
            }
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_wave_barrier();
            const double total = cum_ld(cum_l, cum_g, in_lds, deg - 1);
            const double r = xs_rand(rng) * total;
            int ci = 0x7fffffff;
            for (int i = lane; i < deg; i += 64)
                if (r <= cum_ld(cum_l, cum_g, in_lds, i)) {
                    ci = i;
                    break;
                }
            for (int m = 32; m >= 1; m >>= 1) {
                int o = __shfl_xor(ci, m);
                ci = o < ci ? o : ci;
            }
            chosen = ci == 0x7fffffff ? a.adj[c0] : a.adj[c0 + ci];
        }
        if (lane == 0)
            walk[step] = chosen;
        prev = cur;
        cur = chosen;
    }
    return a.walk_length;
}

// Samples s0 .. s0+nd-1 of one (centre, context) pair (src/node2vec.c:353-386): draw the targets (the stream is
// consumed in sample order), fetch all their rows at once, then score them in order.  FULL (nd == PF) is the
// branch-free path: a rejected negative (== centre or context, :361-363) still has its row fetched, but leaves no
// sample and no contribution.  The sample slot is written before it is known to be kept; a rejected one is
// overwritten by the next sample or by the walk's -1 tail fill.
template <int NR, int PF, bool FULL>
DEVI void n2v_score_chunk(const N2vArgs &a, const N2vBatchArgs &b, unsigned &rng, int s0, int nd, int center, int context,
                          const float (&vc)[NR], float (&neu)[NR], float lr, size_t base, int &ns, int lane, const float *sig) {
    const int dim = a.dim;
    int tg[PF];
    bool ok[PF];
#pragma unroll
    for (int i = 0; i < PF; i++) {
        tg[i] = context;
        ok[i] = false;
        if (FULL || i < nd) {
            if (i == 0 && s0 == 0) {
                ok[i] = true; // the positive sample
            } else {
                tg[i] = a.neg_table[xs32(rng) % N2V_NEG_TABLE];
                ok[i] = tg[i] != center && tg[i] != context;
            }
        }
    }
    float tr[PF][NR];
#pragma unroll
    for (int i = 0; i < PF; i++)
        if (FULL || i < nd) {
            const float *rowt = a.syn1neg + (size_t)tg[i] * dim;
#pragma unroll
            for (int r = 0; r < NR; r++) {
                int d = lane + 64 * r;
                tr[i][r] = d < dim ? rowt[d] : 0.0f;
            }
        }
    float tot[PF];
#pragma unroll
    for (int i = 0; i < PF; i++)
        if (FULL || i < nd) {
            float acc = 0.0f;
#pragma unroll
            for (int r = 0; r < NR; r++)
                if (lane + 64 * r < dim)
                    acc = fmaf(vc[r], tr[i][r], acc);
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1)
                acc = __fadd_rn(acc, __shfl_xor(acc, m));
            tot[i] = acc;
        }
#pragma unroll
    for (int i = 0; i < PF; i++)
        if (FULL || i < nd) {
            const float acc = tot[i];
            const float label = (i == 0 && s0 == 0) ? 1.0f : 0.0f;
            const float err = __fmul_rn(__fsub_rn(label, fast_sigmoid(sig, acc)), lr);
            if (lane == 0) {
                b.s_center[base + ns] = center;
                b.s_target[base + ns] = tg[i];
                b.s_err[base + ns] = err;
            }
            ns += ok[i] ? 1 : 0;
#pragma unroll
            for (int r = 0; r < NR; r++)
                neu[r] = ok[i] ? fmaf(err, tr[i][r], neu[r]) : neu[r];
        }
}

// Round 4, measured and not kept (same box, one pass of 1M walks, scripts/ab_n2v.sh): (i) software-pipelining the chunks — the
// next pair's targets drawn and its rows requested before this pair is scored — doubles the live registers (125 VGPRs, 4
// wavefronts per SIMD instead of 8) and is twice as slow (18.7 vs 9.5 ms per batch); (ii) drawing only the next pair's targets
// ahead, so that the negative-table gather leaves the per-pair chain, still costs 84 VGPRs (5 wavefronts per SIMD): 1.61 vs
// 1.27 s; (iii) reducing a pair's six dot products together (a lane keeps half of its values at each of the first butterfly steps:
// 9 lane exchanges instead of 36, same association and bits) is 1.24 vs 1.26 s — the reductions are not what bounds it.  What
// this kernel lives on is wavefronts in flight, not a shorter chain per wavefront.
template <int NR> // NR = ceil(dim / 64) register slots per lane
__global__ void __launch_bounds__(64) k_n2v_walk_grad(N2vBatchArgs b) {
    extern __shared__ __align__(16) unsigned char smem[];
    const N2vArgs &a = b.a;
    const int lane = threadIdx.x;
    // `split` wavefronts share one walk: each regenerates it (cheap) and owns a contiguous range of positions, with
    // its own fixed range of sample slots, so the (walk, position, pair, sample) order of the output is unchanged
    const int wi = blockIdx.x / b.split, part = blockIdx.x % b.split;
    const int n = b.b0 + wi;
    if (n >= b.b1)
        return;
    const bool uniform = a.p == 1.0 && a.q == 1.0;
    double *cum_l = reinterpret_cast<double *>(smem); // only the biased walk needs it
    int *walk = reinterpret_cast<int *>(smem + (uniform ? 0 : N2VB_LDS_DEG * sizeof(double)));
    float *sig_l = reinterpret_cast<float *>(walk + ((a.walk_length + 3) & ~3));
    for (int i = lane; i <= N2V_SIG_SIZE; i += 64)
        sig_l[i] = a.sig_table[i];
    double *cum_g = b.cum_scratch ? b.cum_scratch + (size_t)blockIdx.x * b.max_deg : nullptr;
    unsigned rng = n2v_walk_seed(b.epoch, b.w, n);
    const double wc = ((double)(b.epoch * a.num_walks + b.w) * a.n + n) * a.walk_length;
    float lr = (float)(a.lr * (1.0 - wc / b.total_words));
    if (lr < (float)(a.lr * 0.0001))
        lr = (float)(a.lr * 0.0001);
    const int wlen = gen_walk(a, n, rng, walk, cum_l, cum_g, lane);
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    const int dim = a.dim;
    const size_t base = (size_t)wi * b.cap;
    const size_t pbase = (size_t)wi * a.walk_length;
    const int p0 = (int)((long long)part * a.walk_length / b.split);
    const int p1 = (int)((long long)(part + 1) * a.walk_length / b.split);
    const int pe = p1 < wlen ? p1 : wlen; // positions [p0, pe) are this wavefront's
    // pairs of the positions before p0 / before pe: every pair owns (1+neg) sample slots and consumes neg draws
    int pairs_p0 = 0, pairs_pe = 0;
    for (int pos = 0; pos < pe; pos++) {
        int cs = pos - a.window, ce = pos + a.window;
        if (cs < 0)
            cs = 0;
        if (ce >= wlen)
            ce = wlen - 1;
        if (pos < p0)
            pairs_p0 += ce - cs;
        pairs_pe += ce - cs;
    }
    for (int i = 0; i < pairs_p0 * a.neg; i++)
        xs32(rng);
    const int slot_end = part == b.split - 1 ? b.cap : pairs_pe * (1 + a.neg);
#ifdef MN_N2V_PF // timing experiments only
    constexpr int PF = MN_N2V_PF;
#else
    // target rows in flight per wavefront.  Measured on MI355X (1M nodes, 20M edges, dim 128, neg 5: device time of one
    // pass of 1M walks, everything else equal): PF 2: 1.37 s, 3: 1.34 s, 6: 1.28 s (one branch-free chunk per pair)
    constexpr int PF = NR <= 4 ? 6 : 2;
#endif
    int ns = pairs_p0 * (1 + a.neg);
    unsigned long long pairs = 0;
    for (int pos = p0; pos < pe; pos++) {
        const int center = walk[pos];
        int cs = pos - a.window, ce = pos + a.window;
        if (cs < 0)
            cs = 0;
        if (ce >= wlen)
            ce = wlen - 1;
        float vc[NR], neu[NR];
        const float *rowc = a.syn0 + (size_t)center * dim;
#pragma unroll
        for (int r = 0; r < NR; r++) {
            int d = lane + 64 * r;
            vc[r] = d < dim ? rowc[d] : 0.0f;
            neu[r] = 0.0f;
        }
        for (int c = cs; c <= ce; c++) {
            if (c == pos)
                continue;
            const int context = walk[c];
            pairs++;
            for (int s0 = 0; s0 <= a.neg; s0 += PF) {
                const int nd = a.neg + 1 - s0 < PF ? a.neg + 1 - s0 : PF;
                if (nd == PF)
                    n2v_score_chunk<NR, PF, true>(a, b, rng, s0, nd, center, context, vc, neu, lr, base, ns, lane, sig_l);
                else
                    n2v_score_chunk<NR, PF, false>(a, b, rng, s0, nd, center, context, vc, neu, lr, base, ns, lane, sig_l);
            }
        }
        if (lane == 0)
            b.p_center[pbase + pos] = center;
#ifndef MN_N2V_NO_PNEU // timing experiments only
        float *pn = b.p_neu + (pbase + pos) * dim;
#pragma unroll
        for (int r = 0; r < NR; r++) {
            int d = lane + 64 * r;
            if (d < dim)
                pn[d] = neu[r];
        }
#endif
    }
    for (int i = ns + lane; i < slot_end; i += 64) {
        b.s_center[base + i] = -1;
        b.s_target[base + i] = -1;
    }
    for (int i = (p0 > wlen ? p0 : wlen) + lane; i < p1; i += 64)
        b.p_center[pbase + i] = -1;
    if (lane == 0)
        atomicAdd(b.pairs_out, pairs);
}

__global__ void k_n2v_keys(const int *dest, int n_samples, int n_nodes, int *keys, int *vals) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n_samples)
        return;
    int d = dest[i];
    keys[i] = d < 0 ? n_nodes : d;
    vals[i] = (int)i;
}

// first sorted position of every destination row that has samples (seg_start preset to -1): no atomics,
// the row id itself is the segment index
__global__ void k_n2v_segments(const int *keys_sorted, int n_samples, int n_nodes, int *seg_start) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n_samples)
        return;
    int k = keys_sorted[i];
    if (k >= n_nodes)
        return;
    if (i == 0 || keys_sorted[i - 1] != k)
        seg_start[k] = (int)i;
}

// target row += Σ err · centre row (old) over the row's samples, in sorted (= sample) order.  Sample metadata is
// fetched 64 at a time (one per lane) and the source rows of 4 samples are in flight while the fmaf chain runs.
template <int NR>
__global__ void __launch_bounds__(64)
    k_n2v_apply(const int *keys_sorted, const int *vals_sorted, const int *seg_start, int n_samples, const int *other,
                const float *s_err, const float *dst_old, const float *src_mat, float *dst_out, int dim, int row0) {
    const int k = row0 + blockIdx.x; // destination row (row0: first row of this rank's shard)
    int j = seg_start[k];
    if (j < 0)
        return;
    const int lane = threadIdx.x;
    float acc[NR];
    const float *row = dst_old + (size_t)k * dim;
#pragma unroll
    for (int r = 0; r < NR; r++) {
        int d = lane + 64 * r;
        acc[r] = d < dim ? row[d] : 0.0f;
    }
    constexpr int U = NR <= 2 ? 8 : NR <= 4 ? 4 : 1;
    for (;;) {
        // this lane's sample of the next 64
        const int jj = j + lane;
        const bool mine = jj < n_samples && keys_sorted[jj] == k;
        int src_row = 0;
        float err = 0.0f;
        if (mine) {
            const int i = vals_sorted[jj];
            err = s_err[i];
            src_row = other[i];
        }
        const int cnt = __popcll(__ballot(mine)); // samples of row k are contiguous: lanes 0..cnt-1
        int t = 0;
        for (; t + U <= cnt; t += U) { // full groups: U source rows in flight, no branches
            float v[U][NR], e[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const float *src = src_mat + (size_t)__shfl(src_row, t + u) * dim;
                e[u] = __shfl(err, t + u);
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    int d = lane + 64 * r;
                    v[u][r] = d < dim ? src[d] : 0.0f;
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int r = 0; r < NR; r++)
                    acc[r] = fmaf(e[u], v[u][r], acc[r]);
        }
        for (; t < cnt; t++) {
            const float *src = src_mat + (size_t)__shfl(src_row, t) * dim;
            const float e1 = __shfl(err, t);
#pragma unroll
            for (int r = 0; r < NR; r++) {
                int d = lane + 64 * r;
                if (d < dim)
                    acc[r] = fmaf(e1, src[d], acc[r]);
            }
        }
        if (cnt < 64)
            break;
        j += 64;
    }
    float *out = dst_out + (size_t)k * dim;
#pragma unroll
    for (int r = 0; r < NR; r++) {
        int d = lane + 64 * r;
        if (d < dim)
            out[d] = acc[r];
    }
}

// centre row: staged = row + Σ neu1e of the walk positions centred on it, in (walk, position) order
template <int NR>
__global__ void __launch_bounds__(64)
    k_n2v_apply_centers(const int *keys_sorted, const int *vals_sorted, const int *seg_start, int n_pos, const float *p_neu,
                        const float *dst_old, float *dst_out, int dim, int row0) {
    const int k = row0 + blockIdx.x;
    int j = seg_start[k];
    if (j < 0)
        return;
    const int lane = threadIdx.x;
    float acc[NR];
    const float *row = dst_old + (size_t)k * dim;
#pragma unroll
    for (int r = 0; r < NR; r++) {
        int d = lane + 64 * r;
        acc[r] = d < dim ? row[d] : 0.0f;
    }
    for (; j < n_pos && keys_sorted[j] == k; j++) {
        const float *src = p_neu + (size_t)vals_sorted[j] * dim;
#pragma unroll
        for (int r = 0; r < NR; r++) {
            int d = lane + 64 * r;
            if (d < dim)
                acc[r] = __fadd_rn(acc[r], src[d]);
        }
    }
    float *out = dst_out + (size_t)k * dim;
#pragma unroll
    for (int r = 0; r < NR; r++) {
        int d = lane + 64 * r;
        if (d < dim)
            out[d] = acc[r];
    }
}

__global__ void __launch_bounds__(64)
    k_n2v_commit(const int *seg_start, const float *staged, float *dst, int dim, int row0) {
    const int k = row0 + blockIdx.x;
    if (seg_start[k] < 0)
        return;
    for (int d = threadIdx.x; d < dim; d += 64)
        dst[(size_t)k * dim + d] = staged[(size_t)k * dim + d];
}

// ───────────────────────── session: the batched schedule as begin / samples / apply / finish ─────────────────────────
// Single-GPU training is a loop over (samples, apply); multi-GPU training lets every rank produce the samples of
// its slice of a batch's walks, exchanges the (centre, target, err) triples (RCCL all-gather, rank order = walk
// order) and applies the full batch on every replica — the N-GPU result is bit-identical to the 1-GPU result.

struct mn_n2v_session {
    int device = 0;
    N2vArgs a;
    int B = 0, cap = 0, max_deg = 0, bits = 1;
    size_t ns_max = 0, np_max = 0, tmp_bytes = 0;
    // graph + model
    int *off = nullptr, *adj = nullptr, *neg = nullptr;
    float *syn0 = nullptr, *syn1 = nullptr, *sig = nullptr, *staged = nullptr;
    // batch scratch
    int *s_center = nullptr, *s_target = nullptr, *keys = nullptr, *vals = nullptr, *keys_s = nullptr, *vals_s = nullptr,
        *keys_c = nullptr, *seg = nullptr, *seg_c = nullptr, *nseg = nullptr, *nseg_c = nullptr;
    float *s_err = nullptr, *p_neu = nullptr;
    int *p_center = nullptr;
    double *cum = nullptr;
    void *tmp = nullptr;
    unsigned long long *pairs = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t st = nullptr; // the session's own stream: every launch, memset, sort and exchange of a training run
    ~mn_n2v_session() {
        (void)hipFree(off); (void)hipFree(adj); (void)hipFree(neg); (void)hipFree(syn0); (void)hipFree(syn1); (void)hipFree(sig);
        (void)hipFree(staged); (void)hipFree(s_center); (void)hipFree(s_target); (void)hipFree(keys); (void)hipFree(vals);
        (void)hipFree(keys_s); (void)hipFree(vals_s); (void)hipFree(keys_c); (void)hipFree(seg); (void)hipFree(seg_c);
        (void)hipFree(nseg); (void)hipFree(nseg_c); (void)hipFree(s_err); (void)hipFree(p_neu); (void)hipFree(p_center); (void)hipFree(cum); (void)hipFree(tmp); (void)hipFree(pairs);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (st) (void)hipStreamDestroy(st);
    }
};

template <int NR>
static int n2v_samples_t(mn_n2v_session *S, int epoch, int w, int lo, int hi, int *d_center, int *d_target, float *d_err,
                         int *d_pcenter, float *d_pneu) {
    N2vBatchArgs b;
    memset(&b, 0, sizeof(b));
    b.a = S->a;
    b.total_words = (double)S->a.n * S->a.num_walks * S->a.walk_length * S->a.epochs;
    b.cap = S->cap;
    b.s_center = d_center;
    b.s_target = d_target;
    b.s_err = d_err;
    b.p_center = d_pcenter;
    b.p_neu = d_pneu;
    b.cum_scratch = S->cum;
    b.max_deg = S->max_deg;
    b.pairs_out = S->pairs;
    b.epoch = epoch;
    b.w = w;
    b.b0 = lo;
    b.b1 = hi;
    const bool uniform = S->a.p == 1.0 && S->a.q == 1.0;
    // enough wavefronts for several full rounds of the chip (8192 resident); the biased walk is too dear to repeat
    int split = uniform ? (32768 + (hi - lo) - 1) / (hi - lo) : 1;
    split = std::max(1, std::min(split, std::min(8, S->a.walk_length)));
#ifdef MN_N2V_NO_SPLIT // timing experiments only
    split = 1;
#endif
    b.split = split;
    const size_t lds = (uniform ? 0 : N2VB_LDS_DEG * sizeof(double)) + (size_t)((S->a.walk_length + 3) & ~3) * sizeof(int) +
                       (N2V_SIG_SIZE + 1) * sizeof(float) + 64;
    hipLaunchKernelGGL((k_n2v_walk_grad<NR>), dim3((unsigned)(hi - lo) * split), dim3(64), lds, S->st, b);
    NCHK(hipGetLastError());
    return 0;
}

template <int NR>
static int n2v_apply_t(mn_n2v_session *S, const int *d_center, const int *d_target, const float *d_err, int64_t ns64,
                       const int *d_pcenter, const float *d_pneu, int64_t np64, int row0, int row1) {
    // rows [row0, row1): the destination rows this call updates (everything: 0, n; a rank of the data-parallel mode: its shard)
    const N2vArgs &a = S->a;
    const int N = a.n, dim = a.dim;
    row1 = std::min(row1, N);
    const unsigned nrows = (unsigned)std::max(0, row1 - row0);
    if (ns64 <= 0 && np64 <= 0)
        return 0;
    if ((size_t)std::max<int64_t>(ns64, 0) > S->ns_max || (size_t)std::max<int64_t>(np64, 0) > S->np_max) {
        nset_err("mn_n2v_apply: %lld samples / %lld positions exceed the session capacity %zu / %zu", (long long)ns64,
                 (long long)np64, S->ns_max, S->np_max);
        return -1;
    }
    const int ns = (int)std::max<int64_t>(ns64, 0), np = (int)std::max<int64_t>(np64, 0);
    const unsigned g256 = (unsigned)((ns + 255) / 256), p256 = (unsigned)((np + 255) / 256);
    // (a rank of the data-parallel mode may have received samples but no positions for its rows, or the reverse: two halves)
    if (np > 0) {
        // centres: syn0[c] + Σ neu1e(position)  → staged (targets below still read the old centres)
        hipLaunchKernelGGL(k_n2v_keys, dim3(p256), dim3(256), 0, S->st, d_pcenter, np, N, S->keys, S->vals);
        if (rocprim::radix_sort_pairs(S->tmp, S->tmp_bytes, S->keys, S->keys_c, S->vals, S->vals_s, (size_t)np, 0, S->bits, S->st) !=
            hipSuccess) {
            nset_err("rocprim::radix_sort_pairs failed");
            return -1;
        }
        NCHK(hipMemsetAsync(S->seg_c, 0xFF, (size_t)N * sizeof(int), S->st));
        hipLaunchKernelGGL(k_n2v_segments, dim3(p256), dim3(256), 0, S->st, S->keys_c, np, N, S->seg_c);
        if (nrows)
            hipLaunchKernelGGL((k_n2v_apply_centers<NR>), dim3(nrows), dim3(64), 0, S->st, S->keys_c, S->vals_s, S->seg_c, np, d_pneu,
                               a.syn0, S->staged, dim, row0);
    }
    if (ns > 0) {
        // targets: syn1neg[t] += Σ err · syn0_old[c]  (syn0 is still the old one) → in place
        hipLaunchKernelGGL(k_n2v_keys, dim3(g256), dim3(256), 0, S->st, d_target, ns, N, S->keys, S->vals);
        if (rocprim::radix_sort_pairs(S->tmp, S->tmp_bytes, S->keys, S->keys_s, S->vals, S->vals_s, (size_t)ns, 0, S->bits, S->st) !=
            hipSuccess) {
            nset_err("rocprim::radix_sort_pairs failed");
            return -1;
        }
        NCHK(hipMemsetAsync(S->seg, 0xFF, (size_t)N * sizeof(int), S->st));
        hipLaunchKernelGGL(k_n2v_segments, dim3(g256), dim3(256), 0, S->st, S->keys_s, ns, N, S->seg);
        if (nrows)
            hipLaunchKernelGGL((k_n2v_apply<NR>), dim3(nrows), dim3(64), 0, S->st, S->keys_s, S->vals_s, S->seg, ns, d_center, d_err,
                               a.syn1neg, a.syn0, a.syn1neg, dim, row0);
    }
    if (np > 0 && nrows)
        hipLaunchKernelGGL(k_n2v_commit, dim3(nrows), dim3(64), 0, S->st, S->seg_c, S->staged, a.syn0, dim, row0);
    NCHK(hipGetLastError());
    return 0;
}

// ── data-parallel exchange by destination shard (mn_node2vec_train_shared) ──
// A rank's samples are wanted by ONE rank each: the one that owns the sample's target row (positions: the centre row).  Instead of
// all-gathering everything to everybody, a rank sorts its slots by destination shard — a stable one-pass radix sort, so every
// bucket keeps the walk order —, packs the buckets and exchanges them all-to-all: 1 / world of the bytes on the links, and each
// rank sorts and applies 1 / world of the samples.  Received buckets stand in source-rank order = the batch's walk order, so a
// row's additions are the ones, in the order, that one GPU makes.
__global__ void k_n2v_shard_keys(const int *dest, int n, int rows_per, int world, int *keys, int *vals, int *hist) {
    __shared__ int h[65];
    for (int i = threadIdx.x; i <= world; i += blockDim.x)
        h[i] = 0;
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)n) {
        const int d = dest[i];
        const int k = d < 0 ? world : d / rows_per; // (unused slot: sorts to the end, is not sent)
        keys[i] = k;
        vals[i] = (int)i;
        atomicAdd(&h[k], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i <= world; i += blockDim.x)
        if (h[i])
            atomicAdd(&hist[i], h[i]);
}
__global__ void k_n2v_pack_samples(const int *order, int n, const int *c, const int *t, const float *e, int *out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n)
        return;
    const int j = order[i];
    out[3 * i] = c[j];
    out[3 * i + 1] = t[j];
    out[3 * i + 2] = __float_as_int(e[j]);
}
__global__ void k_n2v_unpack_samples(const int *in, int n, int *c, int *t, float *e) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n)
        return;
    c[i] = in[3 * i];
    t[i] = in[3 * i + 1];
    e[i] = __int_as_float(in[3 * i + 2]);
}
// a position travels as one record: its centre, then its neu1e vector
__global__ void __launch_bounds__(64) k_n2v_pack_positions(const int *order, const int *pc, const float *pn, int dim, float *out) {
    const int j = order[blockIdx.x];
    float *o = out + (size_t)blockIdx.x * (dim + 1);
    if (threadIdx.x == 0)
        o[0] = __int_as_float(pc[j]);
    for (int d = threadIdx.x; d < dim; d += 64)
        o[1 + d] = pn[(size_t)j * dim + d];
}
__global__ void __launch_bounds__(64) k_n2v_unpack_positions(const float *in, int dim, int *pc, float *pn) {
    const float *r = in + (size_t)blockIdx.x * (dim + 1);
    if (threadIdx.x == 0)
        pc[blockIdx.x] = __float_as_int(r[0]);
    for (int d = threadIdx.x; d < dim; d += 64)
        pn[(size_t)blockIdx.x * dim + d] = r[1 + d];
}

#define N2V_DISPATCH(fn, ...)                                         \
    do {                                                              \
        const int nr__ = (S->a.dim + 63) / 64;                        \
        if (nr__ <= 1) return fn<1>(__VA_ARGS__);                     \
        if (nr__ <= 2) return fn<2>(__VA_ARGS__);                     \
        if (nr__ <= 4) return fn<4>(__VA_ARGS__);                     \
        if (nr__ <= 8) return fn<8>(__VA_ARGS__);                     \
        return fn<16>(__VA_ARGS__);                                   \
    } while (0)

static int n2v_samples(mn_n2v_session *S, int epoch, int w, int lo, int hi, int *c, int *t, float *e, int *pc, float *pn) {
    N2V_DISPATCH(n2v_samples_t, S, epoch, w, lo, hi, c, t, e, pc, pn);
}
static int n2v_apply(mn_n2v_session *S, const int *c, const int *t, const float *e, int64_t ns, const int *pc, const float *pn,
                     int64_t np, int row0 = 0, int row1 = 0x7fffffff) {
    N2V_DISPATCH(n2v_apply_t, S, c, t, e, ns, pc, pn, np, row0, row1);
}
