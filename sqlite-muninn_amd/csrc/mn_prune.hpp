// MN-RU prune of one over-full neighbour row (src/hnsw_algo.c:601-646), shared by the exact sequential
// insert (mn_seq.hip), its speculative variant (mn_spec.hip) and the batch link step (mn_build.hip).  One wavefront;
// the list's nc entries sit in LDS.  Up to 192 entries (any M ≤ 64 whose lists never outgrew M_max) lane i owns
// entries i, i+64 and i+128 in registers (prune_row); longer lists — M > 64, or lists that a delete's reconnection or
// a loaded database grew past M_max, which the reference allows without bound (:142-163,:775-782) — take the same
// decisions from LDS alone (prune_row_long).  prune_any picks.
#pragma once
#include "mn_dist.hpp"

template <bool COH> DEVI int prune_ld(const int *p) {
    if (COH)
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}

DEVI const int *prune_row_ptr(const MnDevIndex &ix, int node, int level) {
    if (level == 0)
        return ix.links0 + (size_t)node * ix.W0;
    return ix.links_up + ((size_t)ix.up_off[node] + (level - 1)) * ix.WU;
}

// MN(t, nn) = |list ∩ N(nn)| (src/hnsw_algo.c:460-475); list in LDS, nn's row from HBM (two links per lane when it has
// at most 128, otherwise 64 at a time: a row holds no duplicates, so an entry is found in at most one pass)
template <bool COH> DEVI int prune_mutual(const MnDevIndex &ix, const int *list, int nc, int nn, int level, int lane) {
    if (ix.levels[nn] < level)
        return 0;
    const int W = level == 0 ? ix.W0 : ix.WU;
    const int *row = prune_row_ptr(ix, nn, level);
    if (W > 128) {
        int c = 0;
        for (int c0 = 0; c0 < W; c0 += 64) {
            const int mine = c0 + lane < W ? prune_ld<COH>(row + c0 + lane) : -1;
            if (!__ballot(mine >= 0))
                break; // rows are packed: nothing after the first empty pass
            for (int i = 0; i < nc; i++)
                if (__ballot(mine >= 0 && mine == list[i]))
                    c++;
        }
        return c;
    }
    const int mine0 = lane < W ? prune_ld<COH>(row + lane) : -1;
    const int mine1 = lane + 64 < W ? prune_ld<COH>(row + lane + 64) : -1;
    int c = 0;
    for (int i = 0; i < nc; i++) {
        int a = list[i];
        if (__ballot((mine0 >= 0 && mine0 == a) || (mine1 >= 0 && mine1 == a)))
            c++;
    }
    return c;
}

template <bool COH>
DEVI void prune_ties(const MnDevIndex &ix, int *list, float *nd, int *mn, int nc, int keep, int level, int lane);

// list[0..nc) → list[0..keep) = the kept neighbours in the reference's order. tq = the row owner's vector
// (LDS), tnorm its cached |t|². nd/mn: LDS scratch of ≥ 192 entries each; lane i owns entries i, i+64, i+128
// (nc ≤ 129: a row of ≤ 128 links plus the new one).  COH: read neighbour rows with agent-scope loads (the
// sequential kernel edits them in the same launch).
// TIES = false: a distance tie (where the outcome depends on neighbours' rows) is not resolved; the list is left
// untouched and 1 is returned, so that the caller can redo the step where those rows are stable.  Returns 0 otherwise.
// last_known: the distance of the LAST candidate (the node being inserted) to the row's owner is already known — the search
// that selected the owner computed it, and every metric gives d(t, s) = d(s, t) bit for bit (src/vec_math.c:78-143; checked by
// the edge-log tests) — so it is not computed again: a 33-candidate list is two passes of 16 rows instead of three.
// pre: all distances but the last were computed ahead (k_insert_seq's helper wavefronts, mn_seq.hip).
template <int ORDER, int NCH, bool COH, bool TIES = true>
DEVI int prune_row(const MnDevIndex &ix, const float *tq, float tnorm, int *list, float *nd, int *mn, int nc, int keep,
                   int level, int lane, bool last_known = false, float last_d = 0.0f, const float *pre = nullptr) {
    constexpr int NS = 3;
    bool has[NS];
    int sl[NS];
    float dd[NS];
#pragma unroll
    for (int k = 0; k < NS; k++) {
        has[k] = lane + 64 * k < nc;
        sl[k] = has[k] ? list[lane + 64 * k] : 0;
        dd[k] = 0.0f;
    }
#pragma unroll
    for (int k = 0; k < NS; k++) {
        const int nk = nc - 64 * k < 64 ? nc - 64 * k : 64;
        const bool holds_last = last_known && nc - 1 >= 64 * k && nc - 1 < 64 * (k + 1); // uniform
        const int ncomp = holds_last ? nk - 1 : nk;
        if (pre) { // (LDS) the distances were computed ahead, by the same code, against the same row
            if (has[k])
                dd[k] = pre[lane + 64 * k];
        } else if (ncomp > 0) { // uniform
            dd[k] = rows_distance<ORDER, NCH>(ix, tq, tnorm, sl[k], ncomp, lane);
        }
        if (holds_last && lane + 64 * k == nc - 1)
            dd[k] = last_d;
        if (has[k] && ix.deleted[sl[k]])
            dd[k] = 1e30f; // :610-612
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < NS; k++)
        if (has[k])
            nd[lane + 64 * k] = dd[k];
    __builtin_amdgcn_wave_barrier();
    // all distances distinct and ordered? then the selection sort is an ascending sort: rank and scatter.
    bool clash = false;
    int rk[NS] = {0, 0, 0};
    for (int x = 0; x < nc; x++) {
        const float o = nd[x];
#pragma unroll
        for (int k = 0; k < NS; k++) {
            if (has[k] && x != lane + 64 * k && !(o < dd[k]) && !(dd[k] < o))
                clash = true; // equal or unordered (NaN)
            rk[k] += o < dd[k];
        }
    }
    if (!__ballot(clash)) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < NS; k++)
            if (has[k] && rk[k] < keep)
                list[rk[k]] = sl[k];
        __builtin_amdgcn_wave_barrier();
        return 0;
    }
    if (!TIES)
        return 1;
    prune_ties<COH>(ix, list, nd, mn, nc, keep, level, lane);
    return 0;
}

// tie path: mutual-neighbour counts (:613-616), then the reference's selection sort verbatim (:620-639)
template <bool COH>
DEVI void prune_ties(const MnDevIndex &ix, int *list, float *nd, int *mn, int nc, int keep, int level, int lane) {
    for (int j = 0; j < nc; j++) {
        const int nn = list[j];
        const int c = ix.deleted[nn] ? -1 : prune_mutual<COH>(ix, list, nc, nn, level, lane);
        if (lane == 0)
            mn[j] = c;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        for (int x = 0; x < keep && x < nc; x++) {
            int bi = x;
            for (int y = x + 1; y < nc; y++)
                if (nd[y] < nd[bi] || (nd[y] == nd[bi] && mn[y] > mn[bi]))
                    bi = y;
            if (bi != x) {
                float td = nd[x];
                nd[x] = nd[bi];
                nd[bi] = td;
                int tm = mn[x];
                mn[x] = mn[bi];
                mn[bi] = tm;
                int ti = list[x];
                list[x] = list[bi];
                list[bi] = ti;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// The same prune for a list of any length (nd / mn: LDS scratch of nc entries rounded up to 64).
template <int ORDER, int NCH, bool COH, bool TIES = true>
DEVI int prune_row_long(const MnDevIndex &ix, const float *tq, float tnorm, int *list, float *nd, int *mn, int nc, int keep,
                        int level, int lane) {
    for (int c0 = 0; c0 < nc; c0 += 64) { // :606-612 distances from the row's owner, 64 rows per pass
        const int n = nc - c0 < 64 ? nc - c0 : 64;
        const int sl = lane < n ? list[c0 + lane] : 0;
        float d = rows_distance<ORDER, NCH>(ix, tq, tnorm, sl, n, lane);
        if (lane < n && ix.deleted[sl])
            d = 1e30f;
        if (lane < n)
            nd[c0 + lane] = d;
    }
    __builtin_amdgcn_wave_barrier();
    bool clash = false; // all distinct and ordered → the selection sort is an ascending sort: rank and scatter
    for (int c0 = 0; c0 < nc; c0 += 64) {
        const int e = c0 + lane;
        if (e < nc) {
            const float de = nd[e];
            int r = 0;
            for (int x = 0; x < nc; x++) {
                const float o = nd[x];
                if (x != e && !(o < de) && !(de < o))
                    clash = true;
                r += o < de;
            }
            mn[e] = r;
        }
    }
    if (!__ballot(clash)) {
        __builtin_amdgcn_wave_barrier();
        int *old = reinterpret_cast<int *>(nd); // the distances are no longer needed: keep the unsorted list there
        for (int e = lane; e < nc; e += 64)
            old[e] = list[e];
        __builtin_amdgcn_wave_barrier();
        for (int e = lane; e < nc; e += 64)
            if (mn[e] < keep)
                list[mn[e]] = old[e];
        __builtin_amdgcn_wave_barrier();
        return 0;
    }
    if (!TIES)
        return 1;
    __builtin_amdgcn_wave_barrier();
    prune_ties<COH>(ix, list, nd, mn, nc, keep, level, lane);
    return 0;
}

template <int ORDER, int NCH, bool COH, bool TIES = true>
DEVI int prune_any(const MnDevIndex &ix, const float *tq, float tnorm, int *list, float *nd, int *mn, int nc, int keep,
                   int level, int lane, bool last_known = false, float last_d = 0.0f, const float *pre = nullptr) {
    if (nc <= 192) // uniform
        return prune_row<ORDER, NCH, COH, TIES>(ix, tq, tnorm, list, nd, mn, nc, keep, level, lane, last_known, last_d, pre);
    return prune_row_long<ORDER, NCH, COH, TIES>(ix, tq, tnorm, list, nd, mn, nc, keep, level, lane);
}
