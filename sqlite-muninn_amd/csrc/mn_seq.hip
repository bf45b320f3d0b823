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
    // change log of ONE insert (n == 1; null otherwise): which edges it added and removed, so that a host that persists the
    // graph edge by edge ("{t}_edges" of the SQLite extension) rewrites only those rows.  chlog[0] = entries (or -1: the log
    // cannot describe this insert — a row wider than 64 links — and the host uses the persist set instead), then entries of
    // MN_CH_INTS ints {op (1 add, 2 delete), source slot, level, target slot, distance bits}
    int *chlog;
    int chcap;
    // link phase: the distances the prunes of one (insert, layer) need — every full target row against its owner — are
    // computed ahead by all wavefronts side by side (targets are distinct rows and a row changes only at its own step), so
    // that the steps themselves, which stay in list order, only rank and write.  pre_rows = targets covered (0: off),
    // pre_w = floats per target.
    int pre_rows, pre_w;
    // every wavefront's own LDS scratch, wave_floats each: the owner vector of a precomputed prune, and (SSE order) the
    // mn_lat_tile_floats(ld, lat_tile_rows) tile of its share of a search's distance request (sse_rows_lat_tiled) — never both at once
    int wave_floats, lat_tile_rows;
    int no_spec_rows; // MN_SPEC_ROWS=0: see MnSearchArgs
};
#define MN_CH_INTS 5

DEVI void seq_log(const MnSeqArgs &a, int &nlog, int op, int src, int level, int dst, unsigned dist_bits) { // lane 0 only
    if (nlog < 0)
        return;
    if (nlog >= a.chcap) {
        nlog = -1;
        return;
    }
    int *e = a.chlog + 1 + (size_t)nlog * MN_CH_INTS;
    e[0] = op;
    e[1] = src;
    e[2] = level;
    e[3] = dst;
    e[4] = (int)dist_bits;
    nlog++;
}

DEVI int *seq_row(const MnDevIndex &ix, int node, int level) {
    if (level == 0)
        return ix.links0 + (size_t)node * ix.W0;
    return ix.links_up + ((size_t)ix.up_off[node] + (level - 1)) * ix.WU;
}

// Does any list of node t hold a soft-deleted node?  persist_node writes such a link with distance 0 (hnsw_get_node is NULL
// for it, src/hnsw_vtab.c:272-273) whenever it rewrites t, so an insert that touches t changes a row no edge of the insert
// names: the change log gives up and the host rewrites t whole.
DEVI bool seq_links_deleted(const MnDevIndex &ix, int t, int lane) {
    bool bad = false;
    const int top = ix.levels[t];
    for (int lv = 0; lv <= top; lv++) {
        const int *row = seq_row(ix, t, lv);
        const int W = lv == 0 ? ix.W0 : ix.WU;
        for (int c0 = 0; c0 < W; c0 += 64) {
            const int v = c0 + lane < W ? ld_link<true>(row + c0 + lane) : -1;
            bad |= v >= 0 && ix.deleted[v] != 0;
        }
    }
    return __ballot(bad) != 0;
}

// this wavefront's share of the link phase's distances (see MnSeqArgs::pre_rows): targets wv, wv + nw, ...
template <int ORDER, int NCH>
DEVI void seq_pre_share(const MnDevIndex &ix, const MnSeqArgs &a, const CoopCtx &c, const int *selbuf, float *pre_nd,
                        int *pre_cnt, float *tvw, int lane) {
    const int l = c.n[2], nsel = c.n[3];
    const int W = l == 0 ? ix.W0 : ix.WU, M_max = l == 0 ? ix.M0 : ix.MU;
    for (int i = c.wv; i < nsel; i += c.nw) {
        int ok = 0;
        const int t = selbuf[i];
        if (i < a.pre_rows && W <= 64 && ix.levels[t] >= l) {
            const int *trow = seq_row(ix, t, l);
            const int v = lane < W ? ld_link<true>(trow + lane) : -1;
            const int cnt = __popcll(__ballot(v >= 0));
            if (cnt >= M_max && cnt + 1 <= a.pre_w) { // a row that will be pruned (if the new node is not in it already)
                const float *tsrc = ix.vectors + (size_t)t * ix.ld;
                __builtin_amdgcn_wave_barrier();
                for (int e = lane; e < ix.ld; e += 64)
                    tvw[e] = tsrc[e];
                __builtin_amdgcn_s_waitcnt(0);
                __builtin_amdgcn_wave_barrier();
                const float tnorm = ix.metric == 1 ? ix.norms[t] : 0.0f;
                const float d = rows_distance<ORDER, NCH>(ix, tvw, tnorm, lane < cnt ? v : 0, cnt, lane);
                if (lane < cnt)
                    pre_nd[i * a.pre_w + lane] = d;
                ok = cnt;
            }
        }
        if (lane == 0)
            pre_cnt[i] = ok;
    }
}

// the helpers' life in k_insert_seq: distance requests of the searches (n >= 0), link-phase requests (-2), the end (-1)
template <int ORDER, int NCH>
DEVI void seq_helper(const MnDevIndex &ix, const MnSeqArgs &a, const float *q, const CoopCtx &c, const int *selbuf,
                     float *pre_nd, int *pre_cnt, float *tvw, int lane) {
    for (;;) {
        __syncthreads();
        const int n = *c.n;
        if (n == -1)
            break;
        if (n == -2)
            seq_pre_share<ORDER, NCH>(ix, a, c, selbuf, pre_nd, pre_cnt, tvw, lane);
        else
            coop_share<ORDER, NCH>(ix, q, c, n, lane);
        __syncthreads();
    }
}

// The inserts are one wavefront's work (the reference's loop, in its order); the other wavefronts of the workgroup only
// stand by for the distance step of its searches (CoopCtx, mn_beam.hpp: a 32-row step is eight round trips to memory for one
// wavefront and one for eight) and leave when it is done.
#define MN_SEQ_WAVES 8
#define MN_SEQ_COOP_BYTES ((4 + 64 + 64) * sizeof(int))
template <int ORDER, int NCH, bool WIDE = false>
__global__ void __launch_bounds__(MN_SEQ_WAVES * 64) k_insert_seq(MnDevIndex ix, MnSeqArgs a, size_t base_lds) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    CoopCtx coop;
    coop.n = reinterpret_cast<int *>(smem + base_lds);
    coop.qnorm = reinterpret_cast<float *>(coop.n + 1);
    coop.list = coop.n + 4;
    coop.dist = reinterpret_cast<float *>(coop.list + 64);
    coop.nw = blockDim.x >> 6;
    coop.wv = threadIdx.x >> 6;
    uint2 *cand_l = reinterpret_cast<uint2 *>(smem);
    uint2 *res_l = cand_l + MN_CAND_LDS;
    int *scratch = reinterpret_cast<int *>(res_l + MN_RES_LDS); // [64]
    int *selbuf = scratch + 64;                                 // [SB]
    unsigned *seldist = reinterpret_cast<unsigned *>(selbuf + a.SB); // [SB] distance bits of the selected neighbours
    int *list = reinterpret_cast<int *>(seldist + a.SB);        // [LW]
    float *nd = reinterpret_cast<float *>(list + a.LW);         // [LW]
    int *mn = reinterpret_cast<int *>(nd + a.LW);               // [LW]
    float *q = reinterpret_cast<float *>(mn + a.LW);            // [ld]
    float *tv = q + ix.ld;                                      // [ld]
    // behind the request area: the precomputed prune distances, their row counts, one owner vector per wavefront
    float *pre_nd = reinterpret_cast<float *>(smem + base_lds + MN_SEQ_COOP_BYTES);
    int *pre_cnt = reinterpret_cast<int *>(pre_nd + (size_t)a.pre_rows * a.pre_w);
    float *tvw = reinterpret_cast<float *>(pre_cnt + ((a.pre_rows + 3) & ~3)) + (size_t)coop.wv * a.wave_floats;
    if (a.lat_tile_rows > 0) {
        coop.tile = tvw;
        coop.tile_rows = a.lat_tile_rows;
    }
    if (coop.wv != 0) {
        seq_helper<ORDER, NCH>(ix, a, q, coop, selbuf, pre_nd, pre_cnt, tvw, lane);
        return;
    }

    WaveCtx w;
    w.coop = &coop; // (works alone too: with one wavefront a request is simply its own share)
    w.no_spec_rows = a.no_spec_rows;
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
    int nlog = a.chlog ? 0 : -1; // (lane 0's count is the one written out)

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
        if (lane == 0)
            *coop.qnorm = w.qnorm; // (the helpers read it, and the new query, behind the first barrier of a request)

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

            beam_layer_auto<ORDER, NCH, true, WIDE>(ix, w, cand, res, bm, words, cur, l, a.ef, lane);
#ifdef MN_PHASE_TIMING
            const unsigned long long ph_link0 = __builtin_amdgcn_s_memrealtime();
#endif
            const int count = res.size;
            const int nsel = count < M_max ? count : M_max; // :511
            int first = cur;
            for (int i = count - 1; i >= 0; i--) {
                uint2 itx = res_take(res, i, count, lane);
                if (i < nsel && lane == 0) {
                    selbuf[i] = (int)itx.y;
                    seldist[i] = itx.x ^ 0x80000000u; // (the result heap holds negated distances: flip the sign back)
                }
                if (i == 0)
                    first = (int)itx.y;
            }
            __builtin_amdgcn_wave_barrier();
            const bool pre_on = a.pre_rows > 0 && nsel >= 4 && W <= 64; // uniform
            if (pre_on) {
                if (lane == 0) {
                    coop.n[2] = l;
                    coop.n[3] = nsel;
                    *coop.n = -2;
                }
                __syncthreads();
                seq_pre_share<ORDER, NCH>(ix, a, coop, selbuf, pre_nd, pre_cnt, tvw, lane);
                __syncthreads();
            }
            int *srow = seq_row(ix, s, l);
            for (int i = 0; i < nsel; i++) { // :582-648
                const int t = selbuf[i];
                if (lane == 0) {
                    st_link(srow + i, t); // node_add_neighbor(new_node, l, selected[i])
                    ix.dirty[t] = 1;      // every neighbour of the new node is re-persisted (src/hnsw_vtab.c:761-768)
                    if (a.chlog)
                        seq_log(a, nlog, 1, s, l, t, seldist[i]);
                }
                if (a.chlog && ix.has_deleted && seq_links_deleted(ix, t, lane))
                    nlog = -1;
                if (ix.levels[t] < l)      // :590
                    continue;
                int *trow = seq_row(ix, t, l);
                // the row, 64 links per pass (two passes when M > 32), staged in LDS in case it has to be pruned
                int cnt = 0, vold = -1;
                bool already = false;
                __builtin_amdgcn_wave_barrier();
                for (int c0 = 0; c0 < W; c0 += 64) {
                    const int v = c0 + lane < W ? ld_link<true>(trow + c0 + lane) : -1;
                    cnt += __popcll(__ballot(v >= 0));
                    already |= __ballot(v == s) != 0;
                    if (c0 + lane < W)
                        list[c0 + lane] = v;
                    vold = v; // (the log below is only kept for rows of at most 64 links: one pass)
                }
                if (already) // already a neighbour (:147-150)
                    continue;
                if (cnt < M_max) {
                    if (lane == 0) {
                        st_link(trow + cnt, s);
                        if (a.chlog) // d(t, s) = d(s, t) bit for bit in all three metrics (src/vec_math.c:78-143)
                            seq_log(a, nlog, 1, t, l, s, seldist[i]);
                    }
                    continue;
                }
                // ── over-full: MN-RU prune of t's list (:601-646).  A list that a delete's reconnection (or a loaded
                //    database) left longer than M_max is cut back to M_max here, exactly as the reference does ──
                const int nc = cnt + 1;
                if (lane == 0)
                    list[cnt] = s;
                const float tnorm = ix.metric == 1 ? ix.norms[t] : 0.0f;
                if (pre_on && i < a.pre_rows && pre_cnt[i] == cnt) { // the distances are there (the row is as it was: its count)
                    __builtin_amdgcn_wave_barrier();
                    prune_any<ORDER, NCH, true>(ix, tv, tnorm, list, nd, mn, nc, M_max, l, lane, true,
                                                __uint_as_float(seldist[i]), pre_nd + (size_t)i * a.pre_w);
                } else {
                    const float *tsrc = ix.vectors + (size_t)t * ix.ld;
                    for (int e = lane; e < ix.ld; e += 64)
                        tv[e] = tsrc[e];
                    __builtin_amdgcn_s_waitcnt(0);
                    __builtin_amdgcn_wave_barrier();
                    prune_any<ORDER, NCH, true>(ix, tv, tnorm, list, nd, mn, nc, M_max, l, lane, true, __uint_as_float(seldist[i]));
                }
                w.n_dist += nc;
                if (a.chlog) {
                    // what the prune changed: every old link that is not among the M_max kept ones was removed; the new
                    // node was added iff it is among them
                    if (W > 64) {
                        nlog = -1;
                    } else {
                        bool kept = false, s_kept = false;
                        for (int j = 0; j < M_max; j++) {
                            const int lj = list[j];
                            kept |= lj == vold;
                            s_kept |= lj == s;
                        }
                        unsigned long long gone = __ballot(vold >= 0 && !kept);
                        while (gone) { // (one link, unless the row had outgrown M_max)
                            const int gl = __ffsll((long long)gone) - 1;
                            gone &= gone - 1;
                            const int u = __shfl(vold, gl);
                            if (lane == 0)
                                seq_log(a, nlog, 2, t, l, u, 0u);
                        }
                        if (s_kept && lane == 0)
                            seq_log(a, nlog, 1, t, l, s, seldist[i]);
                    }
                }
                for (int i2 = lane; i2 < cnt; i2 += 64)
                    st_link(trow + i2, i2 < M_max ? list[i2] : -1);
                __builtin_amdgcn_wave_barrier();
            }
            if (count > 0) // :651-652
                cur = first;
#ifdef MN_PHASE_TIMING
            if (lane == 0)
                atomicAdd(&mn_phase[6], __builtin_amdgcn_s_memrealtime() - ph_link0);
#endif
        }
        if (level > maxl) { // :660-663
            entry = s;
            maxl = level;
        }
    }
    if (lane == 0)
        *coop.n = -1;
    __syncthreads(); // releases the helpers
    if (lane == 0) {
        if (a.chlog)
            a.chlog[0] = nlog;
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
    const size_t own = (size_t)(MN_CAND_LDS + MN_RES_LDS) * sizeof(uint2) + (size_t)(64 + 2 * SB + 3 * LW) * sizeof(int) +
                       2 * (size_t)ix.ld * sizeof(float);
    return ((own + 15) & ~(size_t)15) + MN_SEQ_COOP_BYTES; // + the request area the helper wavefronts watch
}
// + the link phase's precomputed distances (rows x (row width + 1) floats and a count per row) and every wavefront's scratch
static size_t seq_pre_bytes(const MnDevIndex &ix, int *rows, int *w) {
    *rows = ix.M0;
    *w = (std::max(ix.W0, ix.WU) + 1 + 3) & ~3;
    if (std::max(ix.W0, ix.WU) > 64)
        return 0;
    return (size_t)*rows * *w * sizeof(float) + (size_t)((*rows + 3) & ~3) * sizeof(int);
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
                          uint2 *res_ovf, int res_gcap, unsigned long long *counters, hipStream_t st, int *chlog, int chcap) {
    MnSeqArgs a;
    a.chlog = n == 1 ? chlog : nullptr;
    a.chcap = chcap;
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
    size_t lds = mn_insert_seq_lds_bytes(ix);
    const size_t base = lds - MN_SEQ_COOP_BYTES;
    // optional LDS behind the request area, by what fits (64 KB, or what the device grants on request): the link phase's
    // precomputed distances, and per wavefront an owner vector for them / a distance tile of 4 or 2 rows (SSE order)
    int pre_rows = 0, pre_w = 0;
    size_t pre = seq_pre_bytes(ix, &pre_rows, &pre_w);
    const char *pe = getenv("MN_SEQ_PRE"); // MN_SEQ_PRE=0: every prune computes its own distances
    if (pe && atoi(pe) == 0)
        pre = 0;
    a.pre_rows = a.pre_w = a.wave_floats = a.lat_tile_rows = 0;
    {
        const char *sr = getenv("MN_SPEC_ROWS"); // (see prepare_search_ws, mn_index.hip: only for indexes beyond the Infinity Cache)
        const bool big = (size_t)ix.n_slots * ix.ld * sizeof(float) > ((size_t)256 << 20);
        a.no_spec_rows = sr ? (atoi(sr) == 0 ? 1 : 0) : (big ? 0 : 1);
    }
    const size_t lds0 = lds;
    const char *te = getenv("MN_LAT_TILE"); // MN_LAT_TILE=0: no distance tiles
    for (int rows = ix.order == MN_ORDER_SSE_V && ix.ld >= 256 && !(te && atoi(te) == 0) ? 4 : 0; rows >= 0; rows = rows == 4 ? 2 : rows == 2 ? 0 : -1) {
        const int wf = std::max(pre ? ix.ld : 0, rows ? mn_lat_tile_floats(ix.ld, rows) : 0);
        const size_t need = lds0 + pre + (size_t)MN_SEQ_WAVES * wf * sizeof(float);
        if (rows == 0 && !pre)
            break;
        if (need <= (rows ? mn_lds_optin_limit() : (size_t)MN_LDS_LIMIT)) {
            lds = need;
            a.wave_floats = wf;
            a.lat_tile_rows = rows;
            if (pre) {
                a.pre_rows = pre_rows;
                a.pre_w = pre_w;
            }
            break;
        }
    }
    const char *co = getenv("MN_COOP"); // MN_COOP=0: the inserting wavefront alone
    const dim3 blk(co && atoi(co) == 0 ? 64 : MN_SEQ_WAVES * 64);
#define MN_SQ(O, N)                                                                                                          \
    do {                                                                                                                     \
        if (lds > 64 * 1024) {                                                                                               \
            const void *kp = ix.WX > 64 ? reinterpret_cast<const void *>(k_insert_seq<O, N, true>)                           \
                                        : reinterpret_cast<const void *>(k_insert_seq<O, N, false>);                         \
            if (!mn_lds_grant(kp, lds)) { /* not granted: without the tile (and, if that is still too much, without the rest) */ \
                a.lat_tile_rows = 0;                                                                                         \
                a.wave_floats = a.pre_rows ? ix.ld : 0;                                                                      \
                lds = lds0 + pre + (size_t)MN_SEQ_WAVES * a.wave_floats * sizeof(float);                                     \
                if (lds > 64 * 1024) {                                                                                       \
                    a.pre_rows = a.pre_w = a.wave_floats = 0;                                                                \
                    lds = lds0;                                                                                              \
                }                                                                                                            \
                if (lds > 64 * 1024)                                                                                         \
                    (void)mn_lds_grant(kp, lds); /* the kernel's own arrays alone pass 64 KB: refused = the launch fails */  \
            }                                                                                                                \
        }                                                                                                                    \
        if (ix.WX > 64)                                                                                                      \
            hipLaunchKernelGGL((k_insert_seq<O, N, true>), dim3(1), blk, lds, st, ix, a, base);                              \
        else                                                                                                                 \
            hipLaunchKernelGGL((k_insert_seq<O, N>), dim3(1), blk, lds, st, ix, a, base);                                    \
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

#ifdef MN_PHASE_TIMING
extern "C" int mn_debug_phase_seq(unsigned long long *out, int reset) { // probe builds only (scripts/probe_phases.sh)
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mn_phase), 8 * sizeof(unsigned long long)) != hipSuccess)
        return -1;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(mn_phase), z, sizeof(z)) != hipSuccess)
            return -1;
    }
    return 0;
}
#endif

// HIP loads a translation unit's code object on the first use of one of its kernels (several milliseconds for these units): an
// index asks for all of them when it is created (mn_index.hip), so that the first query or insert of a process does not pay.
void mn_module_touch_seq() {
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&k_insert_seq<MN_ORDER_SSE_V, 0, false>));
}
