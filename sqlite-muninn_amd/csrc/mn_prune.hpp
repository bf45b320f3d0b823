// MN-RU prune of one over-full neighbour row (src/hnsw_algo.c:601-646), shared by the exact sequential
// insert (mn_seq.hip) and the batch link step (mn_build.hip). One wavefront; the row's nc = W+1 entries
// (W ≤ 64, so nc ≤ 65) sit in LDS, lane i owns entries i and i+64.
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

// MN(t, nn) = |list ∩ N(nn)| (src/hnsw_algo.c:460-475); list in LDS, nn's row read from HBM
template <bool COH> DEVI int prune_mutual(const MnDevIndex &ix, const int *list, int nc, int nn, int level, int lane) {
    if (ix.levels[nn] < level)
        return 0;
    const int W = level == 0 ? ix.W0 : ix.WU;
    const int *row = prune_row_ptr(ix, nn, level);
    int mine = lane < W ? prune_ld<COH>(row + lane) : -1;
    int c = 0;
    for (int i = 0; i < nc; i++) {
        int a = list[i];
        if (__ballot(mine >= 0 && mine == a))
            c++;
    }
    return c;
}

// list[0..nc) → list[0..keep) = the kept neighbours in the reference's order. tq = the row owner's vector
// (LDS), tnorm its cached |t|². nd/mn: LDS scratch of ≥ 128 entries each. COH: read neighbour rows with
// agent-scope loads (the sequential kernel edits them in the same launch).
// TIES = false: a distance tie (where the outcome depends on neighbours' rows) is not resolved; the list is left
// untouched and 1 is returned, so that the caller can redo the step where those rows are stable.  Returns 0 otherwise.
template <int ORDER, int NCH, bool COH, bool TIES = true>
DEVI int prune_row(const MnDevIndex &ix, const float *tq, float tnorm, int *list, float *nd, int *mn, int nc, int keep,
                   int level, int lane) {
    const bool has0 = lane < nc, has1 = lane + 64 < nc;
    const int s0 = has0 ? list[lane] : 0;
    const int s1 = has1 ? list[lane + 64] : 0;
    const int n0 = nc < 64 ? nc : 64;
    float d0 = rows_distance<ORDER, NCH>(ix, tq, tnorm, s0, n0, lane);
    float d1 = 0.0f;
    if (nc > 64)
        d1 = rows_distance<ORDER, NCH>(ix, tq, tnorm, s1, nc - 64, lane);
    if (has0 && ix.deleted[s0])
        d0 = 1e30f; // :610-612
    if (has1 && ix.deleted[s1])
        d1 = 1e30f;
    __builtin_amdgcn_wave_barrier();
    if (has0)
        nd[lane] = d0;
    if (has1)
        nd[lane + 64] = d1;
    __builtin_amdgcn_wave_barrier();
    // all distances distinct and ordered? then the selection sort is an ascending sort: rank and scatter.
    bool clash = false;
    int r0 = 0, r1 = 0;
    for (int x = 0; x < nc; x++) {
        const float o = nd[x];
        if (has0 && x != lane && !(o < d0) && !(d0 < o))
            clash = true; // equal or unordered (NaN)
        if (has1 && x != lane + 64 && !(o < d1) && !(d1 < o))
            clash = true;
        r0 += o < d0;
        r1 += o < d1;
    }
    if (!__ballot(clash)) {
        __builtin_amdgcn_wave_barrier();
        if (has0 && r0 < keep)
            list[r0] = s0;
        if (has1 && r1 < keep)
            list[r1] = s1;
        __builtin_amdgcn_wave_barrier();
        return 0;
    }
    if (!TIES)
        return 1;
    // tie path: mutual-neighbour counts (:613-616), then the reference's selection sort verbatim (:620-639)
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
    return 0;
}
