// mn_seq.hip — hnsw_insert with the reference's exact sequential semantics (src/hnsw_algo.c:520-666)
// on the device: k_insert_seq, ONE wavefront walking a list of already-uploaded nodes in order.
//
// Every insert sees the graph left by all earlier ones (links are updated in place between
// searches), so the resulting graph is bit-identical to calling the reference's hnsw_insert in a
// loop — including the MN-RU tie-break, which here reads the LIVE neighbour lists exactly as
// count_mutual_neighbors does (:460-475).  The 64 lanes parallelise inside one insert: ≤64
// neighbour distances per expansion / prune, heap sifts, the row scans.  Latency-bound by design
// (≈1 ms per insert); the batch-synchronous schedule (mn_build.hip) is the throughput path.
#include "mn_beam.hpp"
#include "mn_prune.hpp"

struct MnSeqArgs {
    const int *slots; // nodes to insert, in order
    int n;
    int ef;
    int *state; // [0] entry slot, [1] max level — read at start, written back at the end
    unsigned *bitmap0;
    long long bm0_words;
    unsigned *bitmap_up;
    long long bmu_words;
    uint2 *cand_ovf;
    int cand_gcap;
    uint2 *res_ovf;
    int res_gcap;
    unsigned long long *counters;
    int SB; // LDS entries of the selected-list buffer (>= M0)
    int LW; // LDS entries of each prune array (>= longest list + 1, multiple of 64)
};

DEVI int *seq_row(const MnDevIndex &ix, int node, int level) {
    if (level == 0)
        return ix.links0 + (size_t)node * ix.W0;
    return ix.links_up + ((size_t)ix.up_off[node] + (level - 1)) * ix.WU;
}

template <int ORDER, int NCH, bool WIDE = false>
__global__ void __launch_bounds__(64) k_insert_seq(MnDevIndex ix, MnSeqArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    uint2 *cand_l = reinterpret_cast<uint2 *>(smem);
    uint2 *res_l = cand_l + MN_CAND_LDS;
    int *scratch = reinterpret_cast<int *>(res_l + MN_RES_LDS); // [64]
    int *selbuf = scratch + 64;                                 // [SB]
    int *list = selbuf + a.SB;                                  // [LW]
    float *nd = reinterpret_cast<float *>(list + a.LW);         // [LW]
    int *mn = reinterpret_cast<int *>(nd + a.LW);               // [LW]
    float *q = reinterpret_cast<float *>(mn + a.LW);            // [ld]
    float *tv = q + ix.ld;                                      // [ld]

    WaveCtx w;
    w.q = q;
    w.scratch = scratch;
    w.n_dist = 0;
    w.n_exp = 0;
    WHeap cand, res;
    cand.l = cand_l;
    cand.lcap = MN_CAND_LDS;
    cand.g = reinterpret_cast<unsigned long long *>(a.cand_ovf);
    cand.gcap = a.cand_gcap;
    cand.size = 0;
    cand.ovf = 0;
    res.l = res_l;
    res.lcap = MN_RES_LDS;
    res.g = reinterpret_cast<unsigned long long *>(a.res_ovf);
    res.gcap = a.res_gcap;
    res.size = 0;
    res.ovf = 0;

    int entry = a.state[0];
    int maxl = a.state[1];

    for (int it = 0; it < a.n; it++) {
        const int s = a.slots[it];
        const int level = ix.levels[s];
        const float *sv = ix.vectors + (size_t)s * ix.ld;
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < ix.ld; i += 64)
            q[i] = sv[i];
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        w.qnorm = ix.metric == 1 ? ix.norms[s] : 0.0f;

        int cur = entry;
        for (int l = maxl; l > level; l--) // :553-555
            cur = greedy_layer<ORDER, NCH, true, WIDE>(ix, w, cur, l, lane);

        const int start = level < maxl ? level : maxl;
        for (int l = start; l >= 0; l--) { // :572-653
            const int W = l == 0 ? ix.W0 : ix.WU;     // capacity of a row
            const int M_max = l == 0 ? ix.M0 : ix.MU; // what an over-full list is pruned back to
            unsigned *bm = l == 0 ? a.bitmap0 : a.bitmap_up;
            const long long words = l == 0 ? a.bm0_words : a.bmu_words;
            for (long long i = lane; i < words; i += 64)
                bm[i] = 0u;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __builtin_amdgcn_s_waitcnt(0);

            beam_layer<ORDER, NCH, true, WIDE>(ix, w, cand, res, bm, cur, l, a.ef, lane);
            const int count = res.size;
            const int nsel = count < M_max ? count : M_max; // :511
            int first = cur;
            for (int i = count - 1; i >= 0; i--) {
                uint2 itx = heap_pop(res, lane);
                if (i < nsel && lane == 0)
                    selbuf[i] = (int)itx.y;
                if (i == 0)
                    first = (int)itx.y;
            }
            __builtin_amdgcn_wave_barrier();
            int *srow = seq_row(ix, s, l);
            for (int i = 0; i < nsel; i++) { // :582-648
                const int t = selbuf[i];
                if (lane == 0) {
                    st_link(srow + i, t); // node_add_neighbor(new_node, l, selected[i])
                    ix.dirty[t] = 1;      // every neighbour of the new node is re-persisted (src/hnsw_vtab.c:761-768)
                }
                if (ix.levels[t] < l)      // :590
                    continue;
                int *trow = seq_row(ix, t, l);
                // the row, 64 links per pass (two passes when M > 32), staged in LDS in case it has to be pruned
                int cnt = 0;
                bool already = false;
                __builtin_amdgcn_wave_barrier();
                for (int c0 = 0; c0 < W; c0 += 64) {
                    const int v = c0 + lane < W ? ld_link<true>(trow + c0 + lane) : -1;
                    cnt += __popcll(__ballot(v >= 0));
                    already |= __ballot(v == s) != 0;
                    if (c0 + lane < W)
                        list[c0 + lane] = v;
                }
                if (already) // already a neighbour (:147-150)
                    continue;
                if (cnt < M_max) {
                    if (lane == 0)
                        st_link(trow + cnt, s);
                    continue;
                }
                // ── over-full: MN-RU prune of t's list (:601-646).  A list that a delete's reconnection (or a loaded
                //    database) left longer than M_max is cut back to M_max here, exactly as the reference does ──
                const int nc = cnt + 1;
                if (lane == 0)
                    list[cnt] = s;
                const float *tsrc = ix.vectors + (size_t)t * ix.ld;
                for (int e = lane; e < ix.ld; e += 64)
                    tv[e] = tsrc[e];
                __builtin_amdgcn_s_waitcnt(0);
                __builtin_amdgcn_wave_barrier();
                const float tnorm = ix.metric == 1 ? ix.norms[t] : 0.0f;
                prune_any<ORDER, NCH, true>(ix, tv, tnorm, list, nd, mn, nc, M_max, l, lane);
                w.n_dist += nc;
                for (int i = lane; i < cnt; i += 64)
                    st_link(trow + i, i < M_max ? list[i] : -1);
                __builtin_amdgcn_wave_barrier();
            }
            if (count > 0) // :651-652
                cur = first;
        }
        if (level > maxl) { // :660-663
            entry = s;
            maxl = level;
        }
    }
    if (lane == 0) {
        a.state[0] = entry;
        a.state[1] = maxl;
        atomicAdd(&a.counters[0], w.n_dist);
        atomicAdd(&a.counters[1], w.n_exp);
        if (cand.ovf || res.ovf)
            atomicAdd(&a.counters[2], 1ull);
    }
}

size_t mn_insert_seq_lds_bytes(const MnDevIndex &ix) {
    int SB = (ix.M0 + 63) & ~63, LW = (ix.WX + 1 + 63) & ~63;
    if (LW < 192)
        LW = 192;
    return (size_t)(MN_CAND_LDS + MN_RES_LDS) * sizeof(uint2) + (size_t)(64 + SB + 3 * LW) * sizeof(int) +
           2 * (size_t)ix.ld * sizeof(float);
}

static int pick_nch_s(int ld) {
    int need = (ld + 255) / 256;
    if (need <= 1) return 1;
    if (need <= 2) return 2;
    if (need <= 3) return 3;
    if (need <= 4) return 4;
    if (need <= 6) return 6;
    if (need <= 8) return 8;
    return 0;
}

void mn_launch_insert_seq(const MnDevIndex &ix, const int *d_slots, int n, int ef, int *d_state, unsigned *bitmap0,
                          long long bm0_words, unsigned *bitmap_up, long long bmu_words, uint2 *cand_ovf, int cand_gcap,
                          uint2 *res_ovf, int res_gcap, unsigned long long *counters, hipStream_t st) {
    MnSeqArgs a;
    a.slots = d_slots;
    a.n = n;
    a.ef = ef;
    a.state = d_state;
    a.bitmap0 = bitmap0;
    a.bm0_words = bm0_words;
    a.bitmap_up = bitmap_up;
    a.bmu_words = bmu_words;
    a.cand_ovf = cand_ovf;
    a.cand_gcap = cand_gcap;
    a.res_ovf = res_ovf;
    a.res_gcap = res_gcap;
    a.counters = counters;
    a.SB = (ix.M0 + 63) & ~63;
    a.LW = (ix.WX + 1 + 63) & ~63;
    if (a.LW < 192)
        a.LW = 192;
    const size_t lds = mn_insert_seq_lds_bytes(ix);
#define MN_SQ(O, N)                                                                            \
    do {                                                                                       \
        if (ix.WX > 64)                                                                        \
            hipLaunchKernelGGL((k_insert_seq<O, N, true>), dim3(1), dim3(64), lds, st, ix, a); \
        else                                                                                   \
            hipLaunchKernelGGL((k_insert_seq<O, N>), dim3(1), dim3(64), lds, st, ix, a);       \
    } while (0)
    if (ix.order == MN_ORDER_SSE_V) {
        MN_SQ(MN_ORDER_SSE_V, 0);
        return;
    }
    switch (pick_nch_s(ix.ld)) {
    case 1: MN_SQ(MN_ORDER_WAVE_V, 1); break;
    case 2: MN_SQ(MN_ORDER_WAVE_V, 2); break;
    case 3: MN_SQ(MN_ORDER_WAVE_V, 3); break;
    case 4: MN_SQ(MN_ORDER_WAVE_V, 4); break;
    case 6: MN_SQ(MN_ORDER_WAVE_V, 6); break;
    case 8: MN_SQ(MN_ORDER_WAVE_V, 8); break;
    default: MN_SQ(MN_ORDER_WAVE_V, 0); break;
    }
#undef MN_SQ
}
