// mn_beam.hpp — wave-uniform binary heaps and the greedy/beam search device functions shared by
// k_beam (mn_kernels.hip) and k_insert_seq (mn_seq.hip).  gfx950; see mn_kernels.hip for the model.
#pragma once
#include "mn_dist.hpp"

// Link rows are read-only while k_beam runs (plain loads).  k_insert_seq mutates rows while it
// searches, so there every link access is an agent-scope relaxed atomic (sc1: served by L2, never
// by a stale L1 line) — COH = true.
template <bool COH> DEVI int ld_link(const int *p) {
    if (COH)
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
DEVI void st_link(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ───────────────────────── wave-uniform binary heap (src/priority_queue.c) ─────────────────────────
// Item = (distance bits, slot).  1-based; index i < lcap lives in LDS, the rest in global memory
// (agent-scope relaxed atomics there: L1 is bypassed so lanes of the wave see each other's stores).

struct WHeap {
    uint2 *l;
    unsigned long long *g;
    int lcap, gcap;
    int size;
    int ovf;
    int sorted = 0; // (result queue, beam_layer_regs) the array is in descending key order: it can be read instead of popped
};

DEVI uint2 hget(const WHeap &h, int i) {
    if (i < h.lcap)
        return h.l[i];
    unsigned long long v = __hip_atomic_load(&h.g[i - h.lcap], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_uint2((unsigned)(v & 0xffffffffull), (unsigned)(v >> 32));
}
// executed by exactly the lanes that should write
DEVI void hset(const WHeap &h, int i, uint2 v) {
    if (i < h.lcap)
        h.l[i] = v;
    else
        __hip_atomic_store(&h.g[i - h.lcap], (unsigned long long)v.x | ((unsigned long long)v.y << 32), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}

// pq_push + sift_up (src/priority_queue.c:18-26,56-70).  The new item climbs while its parent is
// strictly greater; lane j inspects ancestor j, one ballot finds where it stops.
DEVI void heap_push(WHeap &h, int slot, float d, int lane) {
    if (h.size + 1 >= h.lcap + h.gcap) {
        h.ovf = 1;
        return;
    }
    h.size++;
    const int idx = h.size;
    const int depth = 31 - __clz(idx);
    const bool anc = lane >= 1 && lane <= depth;
    const int p = anc ? (idx >> lane) : idx;
    const uint2 x = make_uint2(f2u(d), (unsigned)slot);
    uint2 it = x;
    if (anc)
        it = hget(h, p);
    unsigned long long stopm = __ballot(anc && (u2f(it.x) <= d));
    const int stop = stopm ? (__ffsll((long long)stopm) - 1) : depth + 1;
    if (lane >= 1 && lane < stop)
        hset(h, idx >> (lane - 1), it);
    if (lane == 0)
        hset(h, idx >> (stop - 1), x);
    __builtin_amdgcn_wave_barrier();
}

// pq_pop + sift_down (src/priority_queue.c:28-42,72-80): strict <, left child first.  Wave-uniform.
DEVI uint2 heap_pop(WHeap &h, int lane) {
    uint2 top = hget(h, 1);
    top.x = rflu(top.x);
    top.y = rflu(top.y);
    uint2 x = hget(h, h.size);
    x.x = rflu(x.x);
    x.y = rflu(x.y);
    h.size--;
    if (h.size > 0) {
        int idx = 1;
        const float xd = u2f(x.x);
        for (;;) {
            int left = 2 * idx;
            if (left > h.size)
                break;
            uint2 L, Rt;
            if (left + 1 < h.lcap) { // both children in LDS, 16-byte aligned pair
                uint4 c = *reinterpret_cast<const uint4 *>(&h.l[left]);
                L = make_uint2(c.x, c.y);
                Rt = make_uint2(c.z, c.w);
            } else {
                L = hget(h, left);
                Rt = (left + 1 <= h.size) ? hget(h, left + 1) : L;
            }
            L.x = rflu(L.x);
            L.y = rflu(L.y);
            Rt.x = rflu(Rt.x);
            Rt.y = rflu(Rt.y);
            int smallest = idx;
            float sd = xd;
            uint2 sv = x;
            if (u2f(L.x) < sd) {
                smallest = left;
                sd = u2f(L.x);
                sv = L;
            }
            if (left + 1 <= h.size && u2f(Rt.x) < sd) {
                smallest = left + 1;
                sv = Rt;
            }
            if (smallest == idx)
                break;
            if (lane == 0)
                hset(h, idx, sv);
            idx = smallest;
        }
        if (lane == 0)
            hset(h, idx, x);
    }
    __builtin_amdgcn_wave_barrier();
    return top;
}

// ───────────────────────── beam search ─────────────────────────

// A query served by a workgroup of several wavefronts (k_beam_coop): wavefront 0 runs the search exactly as the
// one-wavefront kernel does; whenever it needs the distances of a candidate list it posts the list here and every
// wavefront of the group takes every nw-th candidate.  A row's distance is computed by one wavefront with the same
// code either way, so the results are bit-identical; only the latency of the ≤ 64-row distance step shrinks.
struct CoopCtx {
    int *n;       // LDS [1]: candidates of the current request; < 0 = the search is over
    int *list;    // LDS [64] candidate slots
    float *dist;  // LDS [64] their distances
    float *qnorm; // LDS [1]
    int nw, wv;   // wavefronts in the group, this wavefront's index
    float *tile = nullptr; // LDS, this wavefront's own: tile_rows x (ld + 4) floats for sse_rows_lat_tiled, or null
    int tile_rows = 0;
};

// -DMN_PHASE_TIMING (scripts/probe_phases.sh builds a separate library with it; never the product): where one search's
// latency chain goes — per expansion: heap pop, link row + visited probe, distances, heap pushes (s_memrealtime, 100 MHz)
#ifdef MN_PHASE_TIMING
static __device__ unsigned long long mn_phase[8];
#define PH_DECL unsigned long long ph_t = __builtin_amdgcn_s_memrealtime()
#define PH_ADD(w, k)                                                                                                             \
    do {                                                                                                                         \
        const unsigned long long ph_n = __builtin_amdgcn_s_memrealtime();                                                        \
        (w).ph[k] += ph_n - ph_t;                                                                                                \
        ph_t = ph_n;                                                                                                             \
    } while (0)
#define PH_CNT(w, k, n) (w).ph[k] += (n)
#else
#define PH_DECL
#define PH_ADD(w, k)
#define PH_CNT(w, k, n)
#endif

struct WaveCtx {
#ifdef MN_PHASE_TIMING
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    float *tile = nullptr; // LDS staging tile for the coalesced SSE-order loads (k_beam), or null
    const float *q;  // LDS query, zero padded to ld
    float qnorm;
    int *scratch;    // LDS, 64 ints
    unsigned long long n_dist, n_exp;
    // speculative exact build (mn_spec.hip): every link row this search reads is logged, so that the commit
    // step can tell whether an earlier insert of the same window rewrote one of them.  Level-0 rows are logged
    // as the node's slot, upper rows as -(pool row) - 2.
    int *rlog = nullptr;
    int rcap = 0, nr = 0;
    CoopCtx *coop = nullptr;
    int no_spec_rows = 0; // MN_SPEC_ROWS=0 (host): the helpers' rows are requested only once the visited probe has answered (A/B runs)
};
DEVI bool getenv_spec_off(const WaveCtx &w) { return (w.no_spec_rows & 1) != 0; }

// A log entry is MN_RLOG_INTS ints: [0] the row (level 0: the node's slot, upper: -(pool row) - 2); [4] what read it.
// [4] = 0, a beam search: [1] the worst result's distance when the row was opened (float bits: what a neighbour had to beat to be
// pushed; +inf while the results had room); [2..3] the positions of the row whose neighbour was new AND nearer than that —
// everything that could have been pushed.  [4] = 2: the same from the search with the reference's heaps (beam_layer: equal keys
// are about — the ORDER of the pushes counts as well).  [4] = 1, a greedy descent (greedy_layer): [1] the distance to beat, [2] the position
// the scan started from, [3] the position of the neighbour it moved to, -1 if none improved.
// The commit step of a speculative window (mn_spec.hip) uses them to decide whether a rewrite of the row by an earlier insert of
// the window would have changed this search at all.  The defaults written here say "anything would have" (-inf, all positions):
// rows of more than 64 links keep them.
DEVI void log_row_read(const MnDevIndex &ix, WaveCtx &w, int node, int level, int lane) {
    if (!w.rlog)
        return;
    if (lane == 0 && w.nr < w.rcap) {
        int *e = w.rlog + (size_t)w.nr * MN_RLOG_INTS;
        e[0] = level == 0 ? node : -(ix.up_off[node] + level - 1) - 2;
        e[1] = (int)0xff800000u; // -inf
        e[2] = e[3] = -1;
        e[4] = 0;
    }
    w.nr++;
}
// ... and of a greedy step
DEVI void log_greedy_detail(WaveCtx &w, float to_beat, int from, int moved_to, int lane) {
    if (!w.rlog)
        return;
    if (lane == 0 && w.nr >= 1 && w.nr <= w.rcap) {
        int *e = w.rlog + (size_t)(w.nr - 1) * MN_RLOG_INTS;
        e[1] = __float_as_int(to_beat);
        e[2] = from;
        e[3] = moved_to;
        e[4] = 1;
    }
}
// the detail of the entry log_row_read has just written (beam_layer_regs, a row of one chunk)
DEVI void log_row_detail(WaveCtx &w, float worst, unsigned long long could_push, int lane, int kind = 0) {
    if (!w.rlog)
        return;
    if (lane == 0 && w.nr >= 1 && w.nr <= w.rcap) {
        int *e = w.rlog + (size_t)(w.nr - 1) * MN_RLOG_INTS;
        e[1] = __float_as_int(worst);
        e[2] = (int)(unsigned)(could_push & 0xffffffffull);
        e[3] = (int)(unsigned)(could_push >> 32);
        e[4] = kind;
    }
}

// this wavefront's share of the posted candidate list: entries wv, wv+nw, wv+2nw, ...
#define MN_COOP_NO_LEADER 0x10000 // flag in a request's count: the leader is busy elsewhere, the helpers divide the list among themselves
template <int ORDER, int NCH>
DEVI void coop_share(const MnDevIndex &ix, const float *q, const CoopCtx &c, int n_req, int lane) {
    const bool alone = (n_req & MN_COOP_NO_LEADER) != 0; // (never seen by the leader itself)
    const int n = n_req & (MN_COOP_NO_LEADER - 1);
    const int nw = alone ? c.nw - 1 : c.nw, wv = alone ? c.wv - 1 : c.wv;
    const int cnt = n > wv ? (n - wv + nw - 1) / nw : 0;
    if (cnt == 0)
        return;
    const int myslot = lane < cnt ? c.list[wv + lane * nw] : 0;
    const float d = rows_distance<ORDER, NCH, true>(ix, q, *c.qnorm, myslot, cnt, lane, c.tile, c.tile_rows);
    if (lane < cnt)
        c.dist[wv + lane * nw] = d;
}

// the helpers' whole life: wait for a request, do the share, repeat until the leader signals the end
template <int ORDER, int NCH>
DEVI void coop_helper(const MnDevIndex &ix, const float *q, const CoopCtx &c, int lane) {
    for (;;) {
        __syncthreads();
        const int n = *c.n;
        if (n < 0)
            break;
        coop_share<ORDER, NCH>(ix, q, c, n, lane);
        __syncthreads();
    }
}

template <int ORDER, int NCH>
DEVI float ctx_distance(const MnDevIndex &ix, WaveCtx &w, int myslot, int n, int lane) {
    if (!w.coop || n <= 2) // (helpers only reach a barrier when a request is posted)
        return rows_distance<ORDER, NCH>(ix, w.q, w.qnorm, myslot, n, lane, w.tile);
    const CoopCtx &c = *w.coop;
    if (lane < n)
        c.list[lane] = myslot;
    if (lane == 0)
        *c.n = n;
    __syncthreads();
    coop_share<ORDER, NCH>(ix, w.q, c, n, lane);
    __syncthreads();
    return lane < n ? c.dist[lane] : 0.0f;
}

DEVI const int *link_row(const MnDevIndex &ix, int node, int level, int &W) {
    if (level == 0) {
        W = ix.W0;
        return ix.links0 + (size_t)node * ix.W0;
    }
    W = ix.WU;
    return ix.links_up + ((size_t)ix.up_off[node] + (level - 1)) * ix.WU;
}

// src/hnsw_algo.c:257-282 incl. its quirk: after `current` is re-pointed the for-loop carries on at
// index i+1 of the NEW node's list.
// WIDE: rows may hold more than 64 links (M > 32) and are walked in 64-link chunks; with WIDE = false the chunk
// logic folds away at compile time and the code is the single-pass one the throughput kernels were tuned with.
// LOG: the search feeds the read log of a speculative window (only k_beam_coop<BUILD>: every other kernel compiles the log away)
template <int ORDER, int NCH, bool COH = false, bool WIDE = false, bool LOG = false>
DEVI int greedy_layer(const MnDevIndex &ix, WaveCtx &w, int entry, int level, int lane) {
    int cur = entry;
    float cur_d = ctx_distance<ORDER, NCH>(ix, w, cur, 1, lane);
    cur_d = __shfl(cur_d, 0);
    w.n_dist += 1;
    int changed = 1;
    int guard = 0;
    while (changed && guard < (1 << 20)) {
        changed = 0;
        int i0 = 0;
        bool fresh = true; // a row of more than 64 links (M > 32) is walked in 64-link chunks; count the node once
        for (;;) {
            guard++;
            int W;
            const int *row = link_row(ix, cur, level, W);
            const int c0 = WIDE ? (i0 & ~63) : 0;
            if (fresh) {
                w.n_exp++;
                if (LOG)
                    log_row_read(ix, w, cur, level, lane);
            }
            const int pos = c0 + lane;
            int nb = (pos < W) ? ld_link<COH>(row + pos) : -1;
            bool valid = pos >= i0 && nb >= 0 && !(ix.has_deleted && ix.deleted[nb >= 0 ? nb : 0]);
            unsigned long long m = __ballot(valid);
            int n = __popcll(m);
            if (n == 0) {
                if (WIDE && c0 + 64 < W) { // nothing left in this chunk: on to the next one of the same list
                    i0 = c0 + 64;
                    fresh = false;
                    continue;
                }
                if (LOG && !WIDE)
                    log_greedy_detail(w, cur_d, i0, -1, lane);
                break;
            }
            int rank = __popcll(m & ((1ull << lane) - 1ull));
            __builtin_amdgcn_wave_barrier();
            if (valid)
                w.scratch[rank] = nb;
            __builtin_amdgcn_wave_barrier();
            int myslot = lane < n ? w.scratch[lane] : 0;
            float d = ctx_distance<ORDER, NCH>(ix, w, myslot, n, lane);
            w.n_dist += n;
            unsigned long long better = __ballot(lane < n && d < cur_d);
            if (!better) {
                if (WIDE && c0 + 64 < W) {
                    i0 = c0 + 64;
                    fresh = false;
                    continue;
                }
                if (LOG && !WIDE)
                    log_greedy_detail(w, cur_d, i0, -1, lane);
                break;
            }
            int c = __ffsll((long long)better) - 1; // first compact index that improves
            const float beaten = cur_d;
            const int from = i0;
            cur_d = __shfl(d, c);
            cur = __shfl(myslot, c);
            // list position of compact index c = position of the (c+1)-th set bit of m
            int pos_of_me = pos; // lanes with valid hold their own list position
            __builtin_amdgcn_wave_barrier();
            if (valid)
                w.scratch[rank] = pos_of_me;
            __builtin_amdgcn_wave_barrier();
            i0 = w.scratch[c] + 1;
            i0 = rfl(i0);
            cur = rfl(cur);
            if (LOG && !WIDE)
                log_greedy_detail(w, beaten, from, i0 - 1, lane);
            changed = 1;
            fresh = true;
        }
    }
    return cur;
}

// src/hnsw_algo.c:347-448.  Results are left in the result heap; the caller drains it.
template <int ORDER, int NCH, bool COH = false, bool WIDE = false, bool LOG = false>
DEVI void beam_layer(const MnDevIndex &ix, WaveCtx &w, WHeap &cand, WHeap &res, unsigned *bitmap, int entry, int level,
                     int ef, int lane) {
    cand.size = 0;
    res.size = 0;
    if (!(ix.has_deleted && ix.deleted[entry])) { // :360-366
        float d = ctx_distance<ORDER, NCH>(ix, w, entry, 1, lane);
        d = __shfl(d, 0);
        w.n_dist += 1;
        heap_push(cand, entry, d, lane);
        heap_push(res, entry, -d, lane);
        if (lane == 0) {
            int vi = level == 0 ? entry : ix.up_off[entry];
            atomicOr(&bitmap[vi >> 5], 1u << (vi & 31));
        }
    }
    int patience_max = ef / 4; // :372-375
    if (patience_max < 10)
        patience_max = 10;
    int stale = 0;
    int guard = 0;
    while (cand.size > 0 && guard < (1 << 24)) {
        guard++;
        PH_DECL;
        uint2 c = heap_pop(cand, lane);
        const float cd = u2f(c.x);
        if (res.size >= ef) { // :382-386
            float worst = -u2f(rflu(hget(res, 1).x));
            if (cd > worst)
                break;
        }
        if (stale >= patience_max && res.size >= ef) // :391
            break;
        const int node = (int)c.y;
        int W;
        const int *row = link_row(ix, node, level, W);
        w.n_exp++;
        if (LOG)
            log_row_read(ix, w, node, level, lane);
        PH_ADD(w, 0);
        PH_CNT(w, 5, 1);
        int improved = 0;
        const int nchunk = WIDE ? (W + 63) >> 6 : 1; // one pass unless the row has more than 64 links: list order is kept
        for (int ch = 0; ch < nchunk; ch++) {
        const int c0 = WIDE ? ch << 6 : 0;
        int nb = (c0 + lane < W) ? ld_link<COH>(row + c0 + lane) : -1;
        bool todo = false;
        if (nb >= 0) { // :403-409 — mark visited first, then drop deleted
            int vi = level == 0 ? nb : ix.up_off[nb];
            unsigned bit = 1u << (vi & 31);
            unsigned old = atomicOr(&bitmap[vi >> 5], bit);
            todo = !(old & bit) && !(ix.has_deleted && ix.deleted[nb]);
        }
        unsigned long long m = __ballot(todo);
        int n = __popcll(m);
        PH_ADD(w, 1);
        if (n > 0) {
            int rank = __popcll(m & ((1ull << lane) - 1ull));
            __builtin_amdgcn_wave_barrier();
            if (todo)
                w.scratch[rank] = nb;
            __builtin_amdgcn_wave_barrier();
            int myslot = lane < n ? w.scratch[lane] : 0;
            float d = ctx_distance<ORDER, NCH>(ix, w, myslot, n, lane);
            w.n_dist += n;
            PH_ADD(w, 2);
            // :413-425, in list order.  Once the result set is full an element can only be accepted
            // if it beats the worst AT THAT MOMENT, which never exceeds the worst now: pre-filter.
            unsigned long long am;
            float worst_log = __builtin_inff();
            if (res.size >= ef) {
                float worst0 = -u2f(rflu(hget(res, 1).x));
                am = __ballot(lane < n && d < worst0);
                worst_log = worst0;
            } else {
                am = __ballot(lane < n);
            }
            if (LOG && !WIDE && w.rlog) // (speculative windows: what this row's expansion depended on, by list position)
                log_row_detail(w, worst_log, __ballot(todo && ((am >> rank) & 1ull)), lane, 2);
            PH_CNT(w, 4, __popcll(am));
            while (am) {
                int i = __ffsll((long long)am) - 1;
                am &= am - 1;
                float di = __shfl(d, i);
                int si = __shfl(myslot, i);
                di = u2f(rflu(f2u(di)));
                si = rfl(si);
                if (res.size < ef) {
                    heap_push(cand, si, di, lane);
                    heap_push(res, si, -di, lane);
                    improved = 1;
                } else {
                    float worst = -u2f(rflu(hget(res, 1).x));
                    if (di < worst) {
                        heap_push(cand, si, di, lane);
                        heap_pop(res, lane);
                        heap_push(res, si, -di, lane);
                        improved = 1;
                    }
                }
            }
            PH_ADD(w, 3);
        } else if (LOG && !WIDE && w.rlog) {
            log_row_detail(w, res.size >= ef ? -u2f(rflu(hget(res, 1).x)) : __builtin_inff(), 0ull, lane, 2);
        }
        } // chunks of the row
        stale = improved ? 0 : stale + 1; // :428-432
    }
    if (guard >= (1 << 24))
        cand.ovf = 1;
#ifdef MN_PHASE_TIMING
    if (lane == 0)
        for (int k = 0; k < 8; k++) {
            atomicAdd(&mn_phase[k], w.ph[k]);
            w.ph[k] = 0;
        }
#endif
}



// ───────────────────────── the same search with its queues in registers ─────────────────────────
// One search is a latency chain, and with the binary heaps above most of a link of that chain is heap maintenance
// (MN_PHASE_TIMING, 10k x 128: of 7.7 us per expansion, 1.5 us is the candidates' pop — the heap's deeper levels live
// in global memory — and 4.6 us are ≈ 4 × {push, pop, push}, against 1.3 us for the distances).  Where latency is what
// counts — one query per launch (the SQL surface), one insert, a window of speculative inserts — the queues live in the
// wavefront's registers.
//
// Round 4: ONE sorted array serves as both queues.  A priority queue's answers do not depend on its shape as long as no
// two keys in it are equal (the reference's binary heap decides ties by its shape, src/priority_queue.c:56-80), so while
// keys are distinct the loop of src/hnsw_algo.c:376-433 can be restated on sets:
//   * the results W are the ef nearest of everything evaluated so far;
//   * a candidate that is not in W is farther than W's worst, and the worst only falls: popping it could only ever end
//     the loop (:382-386) — it is "dead".  The live candidates are exactly the members of W that have not been expanded.
// So W is kept sorted by distance, rank r in register r / 64 of lane r % 64 (≤ 4 registers per lane: ef ≤ 256), with an
// "expanded" flag in the sign bit of the slot word:
//   pop the nearest candidate = the first member without the flag (one ballot per register in use); none → the loop ends
//                               as the reference's does (candidates empty, or only dead ones left);
//   a row's new distances     = ONE merge for all of them instead of a push / pop / push per element: every new distance
//                               below the worst result is counted against the array (two compares and two popcounts per
//                               register in use: members before it, members after it — which shift up by one) and against
//                               the other new ones; then members and newcomers are written to their new ranks in LDS and
//                               read back rank-major.  Whatever lands at rank ≥ ef is gone (dead, see above).
// "before + after ≠ everybody" means an equal key (or a NaN): the layer is then searched again from its start by
// beam_layer with the real heaps (beam_layer_auto), exactly as before.  The patience counter (:428-432) needs to know
// whether ANY push happened in a row: the first new distance below the row's initial worst is always pushed, and if there
// is none nothing is, so "any" = "a merge took place".
// Measured against round 3's two unsorted arrays with cached extremes (same box, interleaved): DESIGN §4.

#define MN_SA_R 4
#define MN_SA_CAP (64 * MN_SA_R)
#define MN_SA_EACH(X) X(0, k0, v0, s0) X(1, k1, v1, s1) X(2, k2, v2, s2) X(3, k3, v3, s3)
#define MN_SA_DONE ((int)0x80000000)

// (named registers, not arrays: an array indexed by anything but a literal ends up in scratch memory)
struct SQueue {
    float k0, k1, k2, k3;
    int v0, v1, v2, v3; // slot, sign bit = expanded; an empty place is -1 (counts as expanded: never popped)
    int rn;             // members, ranks 0..rn-1
    float wk;           // the worst member's key once rn == ef
};

DEVI void sq_init(SQueue &s) {
    s.k0 = s.k1 = s.k2 = s.k3 = __builtin_nanf("");
    s.v0 = s.v1 = s.v2 = s.v3 = -1;
    s.rn = 0;
    s.wk = 0.0f;
}

// the first member that has not been expanded (the nearest candidate), or -1; MARK: it becomes expanded
template <bool MARK> DEVI int sq_first(SQueue &s, int lane) {
    unsigned long long pm = __builtin_amdgcn_ballot_w64(s.v0 >= 0);
    int pt = 0;
    if (!pm && s.rn > 64) {
        pm = __builtin_amdgcn_ballot_w64(s.v1 >= 0);
        pt = 1;
    }
    if (!pm && s.rn > 128) {
        pm = __builtin_amdgcn_ballot_w64(s.v2 >= 0);
        pt = 2;
    }
    if (!pm && s.rn > 192) {
        pm = __builtin_amdgcn_ballot_w64(s.v3 >= 0);
        pt = 3;
    }
    if (!pm)
        return -1;
    const int pl = __builtin_amdgcn_readfirstlane(__ffsll((long long)pm) - 1);
    int node = 0;
#define MN_SA_POP(T, K, V, S)                                                                                                    \
    if (pt == T) {                                                                                                               \
        node = __builtin_amdgcn_readlane(s.V, pl);                                                                               \
        if (MARK)                                                                                                                \
            s.V = lane == pl ? (s.V | MN_SA_DONE) : s.V;                                                                         \
    }
    MN_SA_EACH(MN_SA_POP)
#undef MN_SA_POP
    return node;
}

// (:413-425) the new distances of a row (lanes with `fresh`: d, slot — any lanes, the merge is a set operation) enter the array
// in one merge.  false: an equal key somewhere (the heaps' business).  `any`: something was pushed.
DEVI bool sq_merge(SQueue &s, uint2 *perm, float d, int myslot, bool fresh, int ef, int lane, int &any,
                   unsigned long long &entering) {
    // what can enter: everything while there is room, else what is nearer than the worst result
    const bool acc = fresh && (s.rn < ef || d < s.wk);
    const unsigned long long am = __ballot(acc);
    entering = am;
    const int na = __popcll(am);
    if (na == 0)
        return true;
    any = 1;
    const int rn = s.rn;
    int s0 = 0, s1 = 0, s2 = 0, s3 = 0; // how far each member moves up
    int mypos = 0;                      // (an entering lane) the rank of its distance
    const float dm = acc ? d : __builtin_nanf(""); // (a NaN compares false both ways: lanes that do not enter count nowhere)
    int pairs = 0; // every (new, member) and (new, new) pair must be ordered one way or the other: counted, checked once
    unsigned long long rem = am;
    while (rem) {
        const int j = __builtin_amdgcn_readfirstlane(__ffsll((long long)rem) - 1);
        rem &= rem - 1;
        const float y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), j));
        int before = 0;
#define MN_SA_CNT(T, K, V, S)                                                                                                    \
    if (rn > T * 64) {                                                                                                           \
        const bool gt = y < s.K;                                                                                                 \
        before += __popcll(__builtin_amdgcn_ballot_w64(s.K < y));                                                                \
        pairs += __popcll(__builtin_amdgcn_ballot_w64(gt));                                                                      \
        S += gt ? 1 : 0;                                                                                                         \
    }
        MN_SA_EACH(MN_SA_CNT)
#undef MN_SA_CNT
        before += __popcll(__builtin_amdgcn_ballot_w64(dm < y));
        pairs += before + __popcll(__builtin_amdgcn_ballot_w64(y < dm));
        mypos = lane == j ? before : mypos;
    }
    if (pairs != na * (rn + na - 1)) // an equal key somewhere
        return false;
    __builtin_amdgcn_wave_barrier();
#define MN_SA_OUT(T, K, V, S)                                                                                                    \
    if (rn > T * 64) {                                                                                                           \
        const int r = T * 64 + lane, p = r + S;                                                                                  \
        if (r < rn && p < ef)                                                                                                    \
            perm[p] = make_uint2(f2u(s.K), (unsigned)s.V);                                                                       \
    }
    MN_SA_EACH(MN_SA_OUT)
#undef MN_SA_OUT
    if (acc && mypos < ef)
        perm[mypos] = make_uint2(f2u(d), (unsigned)myslot);
    __builtin_amdgcn_wave_barrier();
    const int nn = rn + na < ef ? rn + na : ef;
    s.rn = nn;
#define MN_SA_IN(T, K, V, S)                                                                                                     \
    if (nn > T * 64) {                                                                                                           \
        const int r = T * 64 + lane;                                                                                             \
        const uint2 it = perm[r < nn ? r : 0];                                                                                   \
        s.K = r < nn ? u2f(it.x) : __builtin_nanf("");                                                                           \
        s.V = r < nn ? (int)it.y : -1;                                                                                           \
    }
    MN_SA_EACH(MN_SA_IN)
#undef MN_SA_IN
    s.wk = u2f(rflu(perm[nn - 1].x));
    __builtin_amdgcn_wave_barrier();
    return true;
}

// this lane's neighbour (or -1) of a row that is being expanded: the visited probe (:403-409); true = new and alive
template <bool COH> DEVI bool probe_new(const MnDevIndex &ix, unsigned *bitmap, int level, int nb) {
    if (nb < 0)
        return false;
    const int vi = level == 0 ? nb : ix.up_off[nb];
    const unsigned bit = 1u << (vi & 31);
    const unsigned old = atomicOr(&bitmap[vi >> 5], bit);
    return !(old & bit) && !(ix.has_deleted && ix.deleted[nb]);
}

// beam_layer with the queues in registers.  true: `res` holds the results as beam_layer would have left them (and
// res.sorted = 1: position i is the (size - i)-th nearest).  false: a tie (or something else the array cannot decide) came
// up; nothing but the bitmap, the read log and the counters has been touched, and the caller redoes the layer with
// beam_layer.  `perm`: LDS, MN_SA_CAP items (the unused candidate heap).
template <int ORDER, int NCH, bool COH, bool WIDE, bool LOG = false>
DEVI bool beam_layer_regs(const MnDevIndex &ix, WaveCtx &w, WHeap &res, uint2 *perm, unsigned *bitmap, int entry, int level,
                          int ef, int lane) {
    SQueue s;
    sq_init(s);
    bool ok = true;
    if (!(ix.has_deleted && ix.deleted[entry])) { // :360-366
        float d = ctx_distance<ORDER, NCH>(ix, w, entry, 1, lane);
        d = u2f(rflu(f2u(d)));
        w.n_dist += 1;
        ok = d == d;
        if (lane == 0) {
            s.k0 = d;
            s.v0 = entry;
        }
        s.rn = 1;
        s.wk = d;
        if (lane == 0) {
            int vi = level == 0 ? entry : ix.up_off[entry];
            atomicOr(&bitmap[vi >> 5], 1u << (vi & 31));
        }
    }
    int patience_max = ef / 4; // :372-375
    if (patience_max < 10)
        patience_max = 10;
    int stale = 0;
#ifdef MN_PHASE_TIMING
    int ph_second = -1;
#endif
    while (ok) {
        PH_DECL;
        const int node = sq_first<true>(s, lane); // the nearest candidate = the first member that has not been expanded
        if (node < 0)                             // no candidate, or only dead ones: :377 / :382-386
            break;
#ifdef MN_PHASE_TIMING
        { // (probe builds) counter 7 = pops of the previous pop's runner-up
            if (ph_second == node)
                PH_CNT(w, 7, 1);
            ph_second = sq_first<false>(s, lane);
        }
#endif
        if (stale >= patience_max && s.rn >= ef) // :391
            break;
        w.n_exp++;
        if (LOG)
            log_row_read(ix, w, node, level, lane);
        int improved = 0;
        int W;
        const int *row = link_row(ix, node, level, W);
        PH_ADD(w, 0);
        PH_CNT(w, 5, 1);
        const int nchunk = WIDE ? (W + 63) >> 6 : 1;
        for (int ch = 0; ch < nchunk && ok; ch++) {
            const int c0 = WIDE ? ch << 6 : 0;
            int nb = (c0 + lane < W) ? ld_link<COH>(row + c0 + lane) : -1;
            bool todo = false;
            unsigned long long m;
            int n, myslot = 0;
            float d = 0.0f;
            bool compacted = false, nb_todo = false; // (for the read log: where the new neighbours sit)
            int nb_rank = 0;
            const unsigned long long m_all = __ballot(nb >= 0);
            const int n_all = __popcll(m_all);
            if (w.coop && w.coop->nw > 1 && n_all > 2 && (!getenv_spec_off(w) || !w.coop->tile)) {
                // Round 4: with helper wavefronts the rows of ALL the listed neighbours are requested before it is known which of
                // them are new — the visited probe (a returning atomic: one more round trip) then runs next to the row loads
                // instead of in front of them.  Long rows in a cache-resident index are the exception (a helper's tile pass takes
                // 4 rows: the extra rows cost a second pass, 10k x 768: 0.58 -> 0.61 ms; MN_SPEC_ROWS forces either way).  With
                // more than two wavefronts the leader leaves the rows to the helpers altogether (≤ 5 short rows each: one pass):
                // a lone search is bound by the LEADER's instruction stream (≈ 1 000 instructions per expansion; its
                // instruction-cache hit rate is 99.9 %, profiles/r04_lone_query_counters.txt), and its share of the rows was a
                // fifth of that.  Nor are the new ones compacted any more: the merge is a set operation, it takes them where they
                // are.  Same distances for the rows that count, same sets: same bits.
                const CoopCtx &c = *w.coop;
                const int rank_all = __popcll(m_all & ((1ull << lane) - 1ull));
                const bool alone = c.nw > 2 && !c.tile;
                __builtin_amdgcn_wave_barrier();
                if (nb >= 0)
                    c.list[rank_all] = nb;
                if (lane == 0)
                    *c.n = n_all | (alone ? MN_COOP_NO_LEADER : 0);
                __syncthreads(); // the helpers start on their shares
                todo = probe_new<COH>(ix, bitmap, level, nb);
                if (!alone)
                    coop_share<ORDER, NCH>(ix, w.q, c, n_all, lane);
                __syncthreads();
                d = nb >= 0 ? c.dist[rank_all] : 0.0f;
                myslot = nb;
                m = __ballot(todo);
                n = __popcll(m);
                PH_ADD(w, 1);
            } else {
                todo = probe_new<COH>(ix, bitmap, level, nb);
                m = __ballot(todo);
                n = __popcll(m);
                PH_ADD(w, 1);
                if (n > 0) {
                    int rank = __popcll(m & ((1ull << lane) - 1ull));
                    __builtin_amdgcn_wave_barrier();
                    if (todo)
                        w.scratch[rank] = nb;
                    __builtin_amdgcn_wave_barrier();
                    myslot = lane < n ? w.scratch[lane] : 0;
                    d = ctx_distance<ORDER, NCH>(ix, w, myslot, n, lane);
                    compacted = true;
                    nb_todo = todo;
                    nb_rank = rank;
                    todo = lane < n; // (compacted: the rows whose distances were asked for)
                }
            }
            if (n > 0) {
                w.n_dist += n;
                PH_ADD(w, 2);
                if (__ballot(todo && !(d == d))) {
                    ok = false; // a NaN distance: the heaps' business
                    break;
                }
                const float worst0 = s.rn >= ef ? s.wk : __builtin_inff();
                unsigned long long entering;
                ok = sq_merge(s, perm, d, myslot, todo, ef, lane, improved, entering);
                if (LOG && !WIDE && w.rlog) { // (speculative windows) what this row's expansion depended on
                    // lanes → positions of the row: the compacted path numbered the new neighbours in list order
                    const bool e_here = compacted ? (nb_todo && ((entering >> nb_rank) & 1ull)) : ((entering >> lane) & 1ull) != 0;
                    log_row_detail(w, worst0, __ballot(e_here), lane);
                }
                PH_ADD(w, 3);
            } else if (LOG && !WIDE && w.rlog) {
                log_row_detail(w, s.rn >= ef ? s.wk : __builtin_inff(), 0ull, lane); // nothing new in the row: nothing could be pushed
            }
        }
        stale = improved ? 0 : stale + 1; // :428-432
    }
#ifdef MN_PHASE_TIMING
    if (lane == 0)
        for (int k = 0; k < 8; k++) {
            atomicAdd(&mn_phase[k], w.ph[k]);
            w.ph[k] = 0;
        }
#endif
    if (!ok)
        return false;
    // hand the results over as the heap array the callers expect: keys -distance, root (position 1) = the worst
    const int n = s.rn;
    __builtin_amdgcn_wave_barrier();
    res.size = n;
    res.ovf = 0;
    res.sorted = 1;
#define MN_SA_RES(T, K, V, S)                                                                                                    \
    if (T * 64 + lane < n)                                                                                                       \
        hset(res, n - (T * 64 + lane), make_uint2(f2u(-s.K), (unsigned)(s.V & 0x7fffffff)));
    MN_SA_EACH(MN_SA_RES)
#undef MN_SA_RES
    __builtin_amdgcn_wave_barrier();
    return true;
}

// the layer search of the latency-bound paths: registers first, the reference's heaps when a tie has to be decided
template <int ORDER, int NCH, bool COH, bool WIDE, bool LOG = false>
DEVI void beam_layer_auto(const MnDevIndex &ix, WaveCtx &w, WHeap &cand, WHeap &res, unsigned *bitmap, long long bm_words,
                          int entry, int level, int ef, int lane) {
    if (ef <= MN_SA_CAP && ef >= 1 && ef + 1 < res.lcap + res.gcap && cand.lcap >= MN_SA_CAP) {
        const unsigned long long nd0 = w.n_dist, ne0 = w.n_exp;
        const int nr0 = w.nr;
        if (beam_layer_regs<ORDER, NCH, COH, WIDE, LOG>(ix, w, res, cand.l, bitmap, entry, level, ef, lane))
            return;
        w.n_dist = nd0;
        w.n_exp = ne0;
        w.nr = nr0;
        for (long long i = lane; i < bm_words; i += 64)
            bitmap[i] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
    }
    res.sorted = 0;
    beam_layer<ORDER, NCH, COH, WIDE, LOG>(ix, w, cand, res, bitmap, entry, level, ef, lane);
}

// the callers' drain (:436-441): entry i of the results in ascending distance, i counted down from size - 1 to 0 — a pop of
// the heap, or a plain read when the register search left the array sorted
DEVI uint2 res_take(WHeap &res, int i, int count, int lane) {
    if (res.sorted) {
        uint2 it = hget(res, count - i);
        it.x = rflu(it.x);
        it.y = rflu(it.y);
        return it;
    }
    return heap_pop(res, lane);
}
