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
};

struct WaveCtx {
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
};

DEVI void log_row_read(const MnDevIndex &ix, WaveCtx &w, int node, int level, int lane) {
    if (!w.rlog)
        return;
    if (lane == 0 && w.nr < w.rcap)
        w.rlog[w.nr] = level == 0 ? node : -(ix.up_off[node] + level - 1) - 2;
    w.nr++;
}

// this wavefront's share of the posted candidate list: entries wv, wv+nw, wv+2nw, ...
template <int ORDER, int NCH>
DEVI void coop_share(const MnDevIndex &ix, const float *q, const CoopCtx &c, int n, int lane) {
    const int cnt = n > c.wv ? (n - c.wv + c.nw - 1) / c.nw : 0;
    if (cnt == 0)
        return;
    const int myslot = lane < cnt ? c.list[c.wv + lane * c.nw] : 0;
    const float d = rows_distance<ORDER, NCH>(ix, q, *c.qnorm, myslot, cnt, lane);
    if (lane < cnt)
        c.dist[c.wv + lane * c.nw] = d;
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
template <int ORDER, int NCH, bool COH = false, bool WIDE = false>
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
                break;
            }
            int c = __ffsll((long long)better) - 1; // first compact index that improves
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
            changed = 1;
            fresh = true;
        }
    }
    return cur;
}

// src/hnsw_algo.c:347-448.  Results are left in the result heap; the caller drains it.
template <int ORDER, int NCH, bool COH = false, bool WIDE = false>
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
        log_row_read(ix, w, node, level, lane);
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
        if (n > 0) {
            int rank = __popcll(m & ((1ull << lane) - 1ull));
            __builtin_amdgcn_wave_barrier();
            if (todo)
                w.scratch[rank] = nb;
            __builtin_amdgcn_wave_barrier();
            int myslot = lane < n ? w.scratch[lane] : 0;
            float d = ctx_distance<ORDER, NCH>(ix, w, myslot, n, lane);
            w.n_dist += n;
            // :413-425, in list order.  Once the result set is full an element can only be accepted
            // if it beats the worst AT THAT MOMENT, which never exceeds the worst now: pre-filter.
            unsigned long long am;
            if (res.size >= ef) {
                float worst0 = -u2f(rflu(hget(res, 1).x));
                am = __ballot(lane < n && d < worst0);
            } else {
                am = __ballot(lane < n);
            }
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
        }
        } // chunks of the row
        stale = improved ? 0 : stale + 1; // :428-432
    }
    if (guard >= (1 << 24))
        cand.ovf = 1;
}

