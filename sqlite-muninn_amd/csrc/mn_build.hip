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

// Sources of a target in batch order: a bitmap over the batch indices (LDS, nq bits) is walked from the lowest set bit up —
// the O(sources²) "smallest j above the last one" scan of round 1 made a hub with thousands of reverse edges a straggler.
DEVI int next_source(const unsigned *bm, int words, int after, int lane) { // smallest set bit > after, or -1 (uniform)
    int w0 = (after + 1) >> 5;
    if (after + 1 < 0)
        w0 = 0;
    for (int base = w0; base < words; base += 64) {
        const int wi = base + lane;
        unsigned v = wi < words ? bm[wi] : 0u;
        if (wi == w0 && after >= 0)
            v &= ~((((after + 1) & 31) == 0) ? 0u : ((1u << ((after + 1) & 31)) - 1u));
        const unsigned long long any = __ballot(v != 0);
        if (any) {
            const int l = __ffsll((long long)any) - 1;
            const unsigned vv = __shfl(v, l);
            return ((base + l) << 5) + (__ffs((int)vv) - 1);
        }
    }
    return -1;
}

// One wavefront per touched target replays the reference's steps in batch order: append (node_add_neighbor, :142-163), and on
// every overflow the MN-RU prune (:601-646).  Round 3 ran each prune from scratch — 33 distances from the target, a ranking
// pass — which made a HUB a straggler: node2vec embeddings concentrate, most of a batch's 8 192 nodes select the same few
// dozen neighbours, and each of those received thousands of reverse edges, i.e. thousands of full prunes one after the other
// (92 ms for one launch, 9.4 of the 10.3 s of the 1M x 128 index build in round 3's profiles).  The replay is now incremental,
// with the same outcome at every step:
//   * a distance is computed ONCE per candidate (the row's own entries at the first overflow, the sources 64 at a time): a
//     prune recomputes d(target, x) from the same operands each time, so the value it would get is the value kept here;
//   * after a prune without equal distances the row is its candidates in ascending order, so the next overflow is a sorted
//     insert: a source strictly farther than the last entry is what the prune would drop, any other one lands between its
//     neighbours in distance and the last entry falls off;
//   * equal (or unordered) distances are where the mutual-neighbour count and the selection sort's mechanics decide
//     (:613-639): such a step runs the reference's procedure verbatim (prune_ties) on the M_max + 1 candidates, and keeps
//     doing so while equal distances remain in the row.
#ifdef MN_LINK_DEBUG // diagnostics of the link step (scripts/probe builds only): where a straggler's time goes
static __device__ unsigned long long mn_link_dbg[16];
#define LD_ADD(k, v) (dbg[k] += (v))
#else
#define LD_ADD(k, v)
#endif
template <int ORDER, int NCH>
__global__ void __launch_bounds__(64) k_link_reverse(MnDevIndex ix, MnLinkArgs a, int LW, int bm_words) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    if ((int)blockIdx.x >= a.counters[1])
        return;
#ifdef MN_LINK_DEBUG
    unsigned long long dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
#endif
    const int t = a.touched[blockIdx.x];
    if (a.world > 1 && t % a.world != a.rank)
        return; // (divided link step: another rank replays this target and sends the row)
    const int W = a.level == 0 ? ix.W0 : ix.WU;
    const int M_max = a.M_max;
    int *list = reinterpret_cast<int *>(smem);        // [LW]  (the row, up to its full capacity, + the new link)
    float *nd = reinterpret_cast<float *>(list + LW); // [LW]  distances of list[] from the target, once known
    int *mn = reinterpret_cast<int *>(nd + LW);       // [LW]
    int *tid = mn + LW;                               // [LW]  scatter scratch
    float *tdd = reinterpret_cast<float *>(tid + LW); // [LW]
    unsigned *bm = reinterpret_cast<unsigned *>(tdd + LW); // [bm_words] sources of this target by batch index
    float *q = reinterpret_cast<float *>(bm + bm_words);   // [ld]

    const float *tv = ix.vectors + (size_t)t * ix.ld;
    for (int i = lane; i < ix.ld; i += 64)
        q[i] = tv[i];
    const int *row = row_ptr(ix, t, a.level);
    const float tnorm = ix.metric == 1 ? ix.norms[t] : 0.0f;
    const int nb = a.count[t];
    const int *bin = a.bins + a.binoff[t];
    for (int i = lane; i < bm_words; i += 64)
        bm[i] = 0u;
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < nb; i += 64) {
        const int j = bin[i];
        atomicOr(&bm[j >> 5], 1u << (j & 31));
    }
    int nc = 0;
    for (int c0 = 0; c0 < W; c0 += 64) { // 64 links per pass
        const int cur = c0 + lane < W ? row[c0 + lane] : -1;
        nc += __popcll(__ballot(cur >= 0));
        list[c0 + lane] = cur;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();

    const bool may_overflow = nc + nb > M_max; // otherwise every source is appended and no distance is ever needed
    const int n_init = nc;                     // the row's own entries: their distances are computed at the first overflow
    bool have_nd = false, clean = false;       // clean: list[0..M_max) ascending by distance, all different, none unordered
    bool settled = false; // equal distances remain in the row, but it is in the order the prune gives it from ITS OWN entries
    float worst = 0.0f;
    int last = -1;
    for (int c0 = 0; c0 < nb; c0 += 64) {
        const int n = nb - c0 < 64 ? nb - c0 : 64;
        // the next n sources in batch order, one per lane, and their distances in one pass
        int myj = 0;
        for (int k = 0; k < n; k++) {
            last = next_source(bm, bm_words, last, lane);
            if (lane == k)
                myj = last;
        }
        const int svec = lane < n ? a.query_slots[myj] : 0;
        float dvec = 0.0f;
        if (may_overflow) {
            dvec = rows_distance<ORDER, NCH>(ix, q, tnorm, svec, n, lane);
            if (lane < n && ix.deleted[svec])
                dvec = 1e30f; // :610-612
        }
        for (int k = 0; k < n; k++) {
            const int s = __builtin_amdgcn_readlane(svec, k);
            const float ds = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dvec), k));
            // node_add_neighbor (src/hnsw_algo.c:142-163): skip if already present
            bool present = false;
            for (int x0 = 0; x0 < nc; x0 += 64)
                present |= __ballot(x0 + lane < nc && list[x0 + lane] == s) != 0;
            if (present) {
                LD_ADD(0, 1);
                continue;
            }
            if (nc < M_max) { // room left: appended, nothing pruned
                LD_ADD(1, 1);
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) {
                    list[nc] = s;
                    nd[nc] = ds;
                }
                nc++;
                __builtin_amdgcn_wave_barrier();
                continue;
            }
            if (!have_nd) { // first overflow: the distances of the row's own entries (later ones came with their source)
                for (int x0 = 0; x0 < n_init; x0 += 64) {
                    const int m = n_init - x0 < 64 ? n_init - x0 : 64;
                    const int sl = lane < m ? list[x0 + lane] : 0;
                    float d = rows_distance<ORDER, NCH>(ix, q, tnorm, sl, m, lane);
                    if (lane < m && ix.deleted[sl])
                        d = 1e30f;
                    if (lane < m)
                        nd[x0 + lane] = d;
                }
                __builtin_amdgcn_wave_barrier();
                have_nd = true;
            }
            // A source strictly farther than every entry is what the prune drops.  With all distances different that is all the
            // prune does.  With equal distances in the row the prune also re-sorts the tied entries by their mutual-neighbour
            // counts |candidates ∩ N(entry)| (:613-639) — but a source is a node of this batch, and no row read here names one
            // (new rows hold neighbours found in the frozen graph), so it adds nothing to any count: once the row has been put
            // in order by a prune whose dropped candidate was such a source (`settled`), further drops find it in that order
            // and leave it alone — one real prune per change of the row, not one per source of a hub.
            if (M_max <= 64 && ds == ds && (clean || settled) && ds > worst) {
                LD_ADD(2, 1);
                continue;
            }
            if (clean && M_max <= 64 && ds == ds) {
                const float di = lane < M_max ? nd[lane] : 3.0e38f;
                const unsigned long long lt = __ballot(lane < M_max && di < ds), eq = __ballot(lane < M_max && di == ds);
                if (!eq) { // lands between its neighbours in distance; the last entry falls off
                    const int pos = __popcll(lt);
                    const int idi = lane < M_max ? list[lane] : 0;
                    __builtin_amdgcn_wave_barrier();
                    if (lane >= pos && lane + 1 < M_max) {
                        list[lane + 1] = idi;
                        nd[lane + 1] = di;
                    }
                    if (lane == 0) {
                        list[pos] = s;
                        nd[pos] = ds;
                    }
                    __builtin_amdgcn_wave_barrier();
                    worst = nd[M_max - 1];
                    LD_ADD(3, 1);
                    continue;
                }
            }
            // ── the prune itself (:601-646) on the nc + 1 candidates, distances known ──
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
                list[nc] = s;
                nd[nc] = ds;
            }
            nc++;
            __builtin_amdgcn_wave_barrier();
            bool cl = false; // all distances different and ordered → the selection sort is an ascending sort: rank and scatter
            for (int e0 = 0; e0 < nc; e0 += 64) {
                const int e = e0 + lane;
                if (e < nc) {
                    const float de = nd[e];
                    int r = 0;
                    for (int x = 0; x < nc; x++) {
                        const float o = nd[x];
                        if (x != e && !(o < de) && !(de < o))
                            cl = true;
                        r += o < de;
                    }
                    mn[e] = r;
                    tid[e] = list[e];
                    tdd[e] = de;
                }
            }
            if (!__ballot(cl)) {
                __builtin_amdgcn_wave_barrier();
                for (int e = lane; e < nc; e += 64)
                    if (mn[e] < M_max) {
                        list[mn[e]] = tid[e];
                        nd[mn[e]] = tdd[e];
                    }
                __builtin_amdgcn_wave_barrier();
                clean = true;
                settled = false;
                LD_ADD(4, 1);
            } else { // equal or unordered distances: mutual-neighbour counts and the reference's selection sort, verbatim
                LD_ADD(5, 1);
                __builtin_amdgcn_wave_barrier();
                prune_ties<false>(ix, list, nd, mn, nc, M_max, a.level, lane);
                bool dirty = false; // do equal / unordered distances remain among the kept entries?
                for (int e = lane; e + 1 < M_max; e += 64)
                    dirty |= !(nd[e] < nd[e + 1]);
                clean = !__ballot(dirty);
                settled = !clean && list[M_max] == s; // (the one candidate left over sits right behind the kept ones)
            }
            nc = M_max;
            worst = nd[M_max - 1];
        }
    }
    // stage the finished row
    int *out = a.newrows + (size_t)blockIdx.x * ix.WX;
    if (a.world > 1) { // a record {target, row}: any order inside this rank's segment — every record names its row
        int slot = 0;
        if (lane == 0)
            slot = atomicAdd(a.rec_count, 1);
        slot = __builtin_amdgcn_readfirstlane(slot);
        out = a.records + (size_t)slot * (1 + ix.WX);
        if (lane == 0)
            out[0] = t;
        out++;
    }
    for (int i = lane; i < W; i += 64)
        out[i] = i < nc ? list[i] : -1;
#ifdef MN_LINK_DEBUG
    if (lane == 0) {
        const unsigned long long dt = __builtin_amdgcn_s_memrealtime() - t_start;
        for (int k = 0; k < 6; k++)
            atomicAdd(&mn_link_dbg[k], dbg[k]);
        atomicAdd(&mn_link_dbg[6], 1ull);                          // targets
        atomicAdd(&mn_link_dbg[7], (unsigned long long)nb);        // sources
        atomicMax(&mn_link_dbg[8], (unsigned long long)nb);        // most sources of one target
        atomicMax(&mn_link_dbg[9], dt);                            // slowest target (100 MHz ticks)
        atomicAdd(&mn_link_dbg[10], dt);
        if (nb > 1000)
            atomicAdd(&mn_link_dbg[11], 1ull);
    }
#endif
}

#ifdef MN_LINK_DEBUG
extern "C" int mn_debug_link_stats(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mn_link_dbg), 16 * sizeof(unsigned long long)) != hipSuccess)
        return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(mn_link_dbg), z, sizeof(z));
    }
    return 0;
}
#endif

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

// divided link step: touched targets per residue class (the all-gather's segment size), and the commit from the gathered records
__global__ void k_link_classes(MnLinkArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.counters[1])
        return;
    atomicAdd(&a.cls_count[a.touched[i] % a.world], 1);
}
__global__ void k_link_commit_records(MnDevIndex ix, MnLinkArgs a, const int *all, int seg) {
    const int r = blockIdx.y, k = blockIdx.x; // record k of rank r's segment
    if (k >= a.cls_count[r])
        return;
    const int *rec = all + ((size_t)r * seg + k) * (1 + ix.WX);
    const int t = rec[0];
    const int W = a.level == 0 ? ix.W0 : ix.WU;
    int *row = row_ptr(ix, t, a.level);
    for (int j = threadIdx.x; j < W; j += blockDim.x)
        row[j] = rec[1 + j];
}
__global__ void k_link_reset_counts(MnLinkArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.counters[1])
        a.count[a.touched[i]] = 0;
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
    int LW = (ix.WX + 64 + 63) & ~63; // the row is staged 64 links at a time, then one more / a chunk of 64 sources is appended
    if (LW < 192)
        LW = 192;
    const int bm_words = ((a.nq + 31) / 32 + 3) & ~3; // sources of one target by batch index
    size_t lds = (size_t)LW * 5 * sizeof(int) + (size_t)bm_words * sizeof(unsigned) + (size_t)ix.ld * sizeof(float);
    hipLaunchKernelGGL((k_link_reverse<ORDER, NCH>), dim3(max_tuples), dim3(64), lds, st, ix, a, LW, bm_words);
}

static void launch_reverse_any(const MnDevIndex &ix, const MnLinkArgs &a, int max_tuples, hipStream_t st) {
    if (ix.order == MN_ORDER_SSE_V) {
        launch_reverse<MN_ORDER_SSE_V, 0>(ix, a, max_tuples, st);
        return;
    }
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

void mn_launch_link(const MnDevIndex &ix, const MnLinkArgs &a, int max_tuples, hipStream_t st) {
    if (a.nq <= 0)
        return;
    hipMemsetAsync(a.counters, 0, 3 * sizeof(int), st);
    hipLaunchKernelGGL(k_link_forward, dim3((a.nq + 255) / 256), dim3(256), 0, st, ix, a, max_tuples);
    hipLaunchKernelGGL(k_link_offsets, dim3((max_tuples + 255) / 256), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_link_scatter, dim3((max_tuples + 255) / 256), dim3(256), 0, st, a, max_tuples);
    launch_reverse_any(ix, a, max_tuples, st);
    hipLaunchKernelGGL(k_link_commit, dim3(max_tuples), dim3(64), 0, st, ix, a);
}

// divided link step, first half: everything up to this rank's share of the replay (a.world > 1; rec_count and cls_count zeroed here)
void mn_launch_link_first(const MnDevIndex &ix, const MnLinkArgs &a, int max_tuples, hipStream_t st) {
    if (a.nq <= 0)
        return;
    hipMemsetAsync(a.counters, 0, 3 * sizeof(int), st);
    hipMemsetAsync(a.rec_count, 0, sizeof(int), st);
    hipMemsetAsync(a.cls_count, 0, (size_t)a.world * sizeof(int), st);
    hipLaunchKernelGGL(k_link_forward, dim3((a.nq + 255) / 256), dim3(256), 0, st, ix, a, max_tuples);
    hipLaunchKernelGGL(k_link_offsets, dim3((max_tuples + 255) / 256), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_link_scatter, dim3((max_tuples + 255) / 256), dim3(256), 0, st, a, max_tuples);
    hipLaunchKernelGGL(k_link_classes, dim3((max_tuples + 255) / 256), dim3(256), 0, st, a);
    launch_reverse_any(ix, a, max_tuples, st);
}
void mn_launch_link_commit_records(const MnDevIndex &ix, const MnLinkArgs &a, int max_tuples, const int *all_records, int seg,
                                   hipStream_t st) {
    if (a.nq <= 0)
        return;
    if (seg > 0)
        hipLaunchKernelGGL(k_link_commit_records, dim3((unsigned)seg, (unsigned)a.world), dim3(64), 0, st, ix, a, all_records, seg);
    hipLaunchKernelGGL(k_link_reset_counts, dim3((max_tuples + 255) / 256), dim3(256), 0, st, a);
}

// HIP loads a translation unit's code object on the first use of one of its kernels (several milliseconds for these units): an
// index asks for all of them when it is created (mn_index.hip), so that the first query or insert of a process does not pay.
void mn_module_touch_build() {
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&k_link_offsets));
}
