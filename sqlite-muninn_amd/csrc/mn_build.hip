// mn_build.hip — link half of hnsw_insert (src/hnsw_algo.c:581-648) for a whole batch, gfx950.
//
// Batch-synchronous schedule (oracle/mn_oracle.c orc_hnsw_insert_batch restates
// it on the CPU).  For one layer l, after k_beam produced selected_l(j) for every batch node j:
//   k_link_forward  new node rows ← selected lists; one (target, j) tuple per selected neighbour;
//                   count[target]++ and a list of touched targets
//   k_link_offsets  bin offset per touched target
//   k_link_scatter  tuples → bins
//   k_link_reverse  one wavefront per touched target: its sources in batch order are appended to a
//                   copy of its row; every overflow is pruned by MN-RU (:601-646): 33 distances
//                   from the target, mutual-neighbour counts only when two distances tie (the count
//                   is a tie-break only, :623-624); rows of OTHER nodes are read as they stood
//                   before any reverse edge of this batch (nothing is written in place)
//   k_link_commit   new rows → links; scratch counters back to zero
// No float atomics, no order-dependent results: every target is owned by one wavefront and its
// sources are sorted by batch index.
#include "mn_dist.hpp"
#include "mn_prune.hpp"

DEVI int *row_ptr(const MnDevIndex &ix, int node, int level) {
    if (level == 0)
        return ix.links0 + (size_t)node * ix.W0;
    return ix.links_up + ((size_t)ix.up_off[node] + (level - 1)) * ix.WU;
}

__global__ void k_link_forward(MnDevIndex ix, MnLinkArgs a, int max_tuples) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.nq)
        return;
    int s = a.query_slots[j];
    if (ix.levels[s] < a.level || a.level >= a.nlev)
        return;
    const int W = a.level == 0 ? ix.W0 : ix.WU;
    int n = a.nsel[(size_t)j * a.nlev + a.level];
    const int *sel = a.sel + ((size_t)j * a.nlev + a.level) * ix.M0;
    int *row = row_ptr(ix, s, a.level);
    for (int i = 0; i < W; i++)
        row[i] = i < n ? sel[i] : -1;
    for (int i = 0; i < n; i++)
        ix.dirty[sel[i]] = 1; // every neighbour of a new node is re-persisted (src/hnsw_vtab.c:761-768)
    for (int i = 0; i < n; i++) {
        int t = sel[i];
        if (ix.levels[t] < a.level) // src/hnsw_algo.c:590
            continue;
        int p = atomicAdd(&a.counters[0], 1);
        if (p >= max_tuples)
            continue;
        a.t_target[p] = t;
        a.t_src[p] = j;
        int old = atomicAdd(&a.count[t], 1);
        if (old == 0) {
            int q = atomicAdd(&a.counters[1], 1);
            a.touched[q] = t;
        }
    }
}

__global__ void k_link_offsets(MnLinkArgs a) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.counters[1])
        return;
    int t = a.touched[i];
    a.binoff[t] = atomicAdd(&a.counters[2], a.count[t]);
    a.fill[t] = 0;
}

__global__ void k_link_scatter(MnLinkArgs a, int max_tuples) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int n = a.counters[0] < max_tuples ? a.counters[0] : max_tuples;
    if (i >= n)
        return;
    int t = a.t_target[i];
    int p = atomicAdd(&a.fill[t], 1);
    a.bins[a.binoff[t] + p] = a.t_src[i];
}

template <int ORDER, int NCH>
__global__ void __launch_bounds__(64) k_link_reverse(MnDevIndex ix, MnLinkArgs a, int LW) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    if ((int)blockIdx.x >= a.counters[1])
        return;
    const int t = a.touched[blockIdx.x];
    const int W = a.level == 0 ? ix.W0 : ix.WU;
    const int M_max = a.M_max;
    int *list = reinterpret_cast<int *>(smem);        // [LW]  (the row, up to its full capacity, + the new link)
    float *nd = reinterpret_cast<float *>(list + LW); // [LW]
    int *mn = reinterpret_cast<int *>(nd + LW);       // [LW]
    float *q = reinterpret_cast<float *>(mn + LW);    // [ld]

    const float *tv = ix.vectors + (size_t)t * ix.ld;
    for (int i = lane; i < ix.ld; i += 64)
        q[i] = tv[i];
    const int *row = row_ptr(ix, t, a.level);
    int nc = 0;
    for (int c0 = 0; c0 < W; c0 += 64) { // 64 links per pass
        const int cur = c0 + lane < W ? row[c0 + lane] : -1;
        nc += __popcll(__ballot(cur >= 0));
        list[c0 + lane] = cur;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    const float tnorm = ix.metric == 1 ? ix.norms[t] : 0.0f;

    const int nb = a.count[t];
    const int *bin = a.bins + a.binoff[t];
    int last = -1;
    for (int it = 0; it < nb; it++) {
        // next source in batch order: smallest j > last
        int best = 0x7fffffff;
        for (int i = lane; i < nb; i += 64) {
            int j = bin[i];
            if (j > last && j < best)
                best = j;
        }
        for (int m = 32; m >= 1; m >>= 1) {
            int o = __shfl_xor(best, m);
            best = o < best ? o : best;
        }
        last = best;
        const int s = a.query_slots[best];
        // node_add_neighbor (src/hnsw_algo.c:142-163): skip if already present
        bool present = false;
        for (int c0 = 0; c0 < nc; c0 += 64)
            present |= __ballot(c0 + lane < nc && list[c0 + lane] == s) != 0;
        if (present)
            continue;
        __builtin_amdgcn_wave_barrier();
        if (lane == 0)
            list[nc] = s;
        nc++;
        __builtin_amdgcn_wave_barrier();
        if (nc <= M_max)
            continue;
        // ── prune to M_max (:601-646) ──
        prune_any<ORDER, NCH, false>(ix, q, tnorm, list, nd, mn, nc, M_max, a.level, lane);
        nc = M_max;
    }
    // stage the finished row
    int *out = a.newrows + (size_t)blockIdx.x * ix.WX;
    for (int i = lane; i < W; i += 64)
        out[i] = i < nc ? list[i] : -1;
}

__global__ void k_link_commit(MnDevIndex ix, MnLinkArgs a) {
    int i = blockIdx.x;
    if (i >= a.counters[1])
        return;
    int t = a.touched[i];
    const int W = a.level == 0 ? ix.W0 : ix.WU;
    int *row = row_ptr(ix, t, a.level);
    const int *src = a.newrows + (size_t)i * ix.WX;
    for (int j = threadIdx.x; j < W; j += blockDim.x)
        row[j] = src[j];
    if (threadIdx.x == 0)
        a.count[t] = 0;
}

static int pick_nch_b(int ld) {
    int need = (ld + 255) / 256;
    if (need <= 1) return 1;
    if (need <= 2) return 2;
    if (need <= 3) return 3;
    if (need <= 4) return 4;
    if (need <= 6) return 6;
    if (need <= 8) return 8;
    return 0;
}

template <int ORDER, int NCH>
static void launch_reverse(const MnDevIndex &ix, const MnLinkArgs &a, int max_tuples, hipStream_t st) {
    int LW = (ix.WX + 64 + 63) & ~63; // the row is staged 64 links at a time, then one more is appended
    if (LW < 192)
        LW = 192;
    size_t lds = (size_t)LW * 3 * sizeof(int) + (size_t)ix.ld * sizeof(float);
    hipLaunchKernelGGL((k_link_reverse<ORDER, NCH>), dim3(max_tuples), dim3(64), lds, st, ix, a, LW);
}

void mn_launch_link(const MnDevIndex &ix, const MnLinkArgs &a, int max_tuples, hipStream_t st) {
    if (a.nq <= 0)
        return;
    hipMemsetAsync(a.counters, 0, 3 * sizeof(int), st);
    hipLaunchKernelGGL(k_link_forward, dim3((a.nq + 255) / 256), dim3(256), 0, st, ix, a, max_tuples);
    hipLaunchKernelGGL(k_link_offsets, dim3((max_tuples + 255) / 256), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_link_scatter, dim3((max_tuples + 255) / 256), dim3(256), 0, st, a, max_tuples);
    if (ix.order == MN_ORDER_SSE_V) {
        launch_reverse<MN_ORDER_SSE_V, 0>(ix, a, max_tuples, st);
    } else {
        switch (pick_nch_b(ix.ld)) {
        case 1: launch_reverse<MN_ORDER_WAVE_V, 1>(ix, a, max_tuples, st); break;
        case 2: launch_reverse<MN_ORDER_WAVE_V, 2>(ix, a, max_tuples, st); break;
        case 3: launch_reverse<MN_ORDER_WAVE_V, 3>(ix, a, max_tuples, st); break;
        case 4: launch_reverse<MN_ORDER_WAVE_V, 4>(ix, a, max_tuples, st); break;
        case 6: launch_reverse<MN_ORDER_WAVE_V, 6>(ix, a, max_tuples, st); break;
        case 8: launch_reverse<MN_ORDER_WAVE_V, 8>(ix, a, max_tuples, st); break;
        default: launch_reverse<MN_ORDER_WAVE_V, 0>(ix, a, max_tuples, st); break;
        }
    }
    hipLaunchKernelGGL(k_link_commit, dim3(max_tuples), dim3(64), 0, st, ix, a);
}
