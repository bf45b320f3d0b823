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
DEVI bool getenv_spec_off(const WaveCtx &w) { return w.no_spec_rows != 0; }

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
    const float d = rows_distance<ORDER, NCH, true>(ix, q, *c.qnorm, myslot, cnt, lane, c.tile, c.tile_rows);
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
            if (res.size >= ef) {
                float worst0 = -u2f(rflu(hget(res, 1).x));
                am = __ballot(lane < n && d < worst0);
            } else {
                am = __ballot(lane < n);
            }
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



// ───────────────────────── the same search with its two queues in registers ─────────────────────────
// One search is a latency chain, and with the binary heaps above most of a link of that chain is heap maintenance
// (MN_PHASE_TIMING, 10k x 128: of 7.7 us per expansion, 1.5 us is the candidates' pop — the heap's deeper levels live
// in global memory — and 4.6 us are ≈ 4 × {push, pop, push}, against 1.3 us for the distances).  Where latency is what
// counts — one query per launch (the SQL surface), one insert, a window of speculative inserts — the two queues are
// kept as UNSORTED arrays spread over the wavefront's registers (element e in register e / 64 of lane e % 64, empty
// places hold NaN) with the one element each queue is asked for cached: the nearest candidate, the worst result.
//   insert          = one v_writelane per word at the end of the array (+ compare with the cached element)
//   pop the nearest = move the last element into its place, then a DPP min-reduction over the wavefront finds the next
//   replace worst   = overwrite it in place, then a max-reduction finds the next worst
// No memory is touched.  A priority queue's answers do not depend on its shape as long as no two keys in it are equal;
// the reference's binary heap decides ties by its shape (src/priority_queue.c:56-80).  So every insert first tests the
// queue for an equal key (four compares) and for NaN: at the first one the layer is searched again from the start by
// beam_layer with the real heaps.  Results are ranked once at the end and handed over as the result heap the callers
// drain (a descending array is a heap; `sorted` tells them they may read it directly).
// ef ≤ 256 (4 registers per lane and queue).  When more than 256 candidates are alive the farthest is overwritten,
// which is exact: with more than ef candidates alive it is farther than the worst result and could only ever end the
// loop (checked; otherwise → the heaps).

#define MN_SA_R 4
#define MN_SA_CAP (64 * MN_SA_R)

// (named registers, not arrays: an array indexed by anything but a literal ends up in scratch memory)
struct UArr {
    float k0, k1, k2, k3;
    int v0, v1, v2, v3;
    int n;    // elements 0..n-1 are in use
    float bk; // the cached element's key (candidates: smallest, results: largest) ...
    int bp;   // ... and its place
};
#define MN_SA_EACH(X) X(0, k0, v0) X(1, k1, v1) X(2, k2, v2) X(3, k3, v3)

DEVI void ua_init(UArr &a, float none) {
    a.k0 = a.k1 = a.k2 = a.k3 = __builtin_nanf("");
    a.v0 = a.v1 = a.v2 = a.v3 = 0;
    a.n = 0;
    a.bk = none;
    a.bp = 0;
}

DEVI void ua_get(const UArr &a, int e, float &key, int &val) { // e uniform
    const int sl = __builtin_amdgcn_readfirstlane(e >> 6), l = __builtin_amdgcn_readfirstlane(e & 63);
    // (the empty asm statements keep the compiler from turning the selection into a table on the stack)
    int kk = __float_as_int(a.k0), vv = a.v0;
    if (sl == 1) {
        kk = __float_as_int(a.k1);
        vv = a.v1;
    }
    asm volatile("" : "+v"(kk), "+v"(vv));
    if (sl == 2) {
        kk = __float_as_int(a.k2);
        vv = a.v2;
    }
    asm volatile("" : "+v"(kk), "+v"(vv));
    if (sl == 3) {
        kk = __float_as_int(a.k3);
        vv = a.v3;
    }
    asm volatile("" : "+v"(kk), "+v"(vv));
    key = __int_as_float(__builtin_amdgcn_readlane(kk, l));
    val = __builtin_amdgcn_readlane(vv, l);
}

DEVI void ua_put(UArr &a, int e, float key, int val, int lane) { // e, key, val uniform
    const int sl = __builtin_amdgcn_readfirstlane(e >> 6);
    const bool here = lane == (e & 63);
    // (the empty asm statements keep the compiler from turning the four cases into an indexed array on the stack)
#define MN_SA_PUT(T, K, V)                                                                                                       \
    if (sl == T) {                                                                                                               \
        a.K = here ? key : a.K;                                                                                                  \
        a.V = here ? val : a.V;                                                                                                  \
    }                                                                                                                            \
    asm volatile("" : "+v"(a.K), "+v"(a.V));
    MN_SA_EACH(MN_SA_PUT)
#undef MN_SA_PUT
}

DEVI bool ua_holds(const UArr &a, float key) { // an equal key is in the queue (empty places are NaN: never equal)
    const unsigned long long m = __builtin_amdgcn_ballot_w64(a.k0 == key) | __builtin_amdgcn_ballot_w64(a.k1 == key) |
                                 __builtin_amdgcn_ballot_w64(a.k2 == key) | __builtin_amdgcn_ballot_w64(a.k3 == key);
    return m != 0;
}

// wavefront-wide min / max by DPP: xor 1, xor 2 inside the quads, the mirrored half row, the mirrored row, then row 0 → 1,
// 2 → 3 and rows 0-1 → 2-3 by row broadcast; lane 63 ends up with the result.  NaN operands are ignored (v_min / v_max).
template <bool MAXQ> DEVI float ua_op(float x, float y) { return MAXQ ? __builtin_fmaxf(x, y) : __builtin_fminf(x, y); }
template <bool MAXQ> DEVI float wave_extreme(float x) {
#define MN_DPP(ctrl, rmask) __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), ctrl, rmask, 0xf, false))
    x = ua_op<MAXQ>(x, MN_DPP(0xB1, 0xf));
    x = ua_op<MAXQ>(x, MN_DPP(0x4E, 0xf));
    x = ua_op<MAXQ>(x, MN_DPP(0x141, 0xf)); // row_half_mirror
    x = ua_op<MAXQ>(x, MN_DPP(0x140, 0xf)); // row_mirror
    x = ua_op<MAXQ>(x, MN_DPP(0x142, 0xa)); // row_bcast:15 into rows 1 and 3
    x = ua_op<MAXQ>(x, MN_DPP(0x143, 0xc)); // row_bcast:31 into rows 2 and 3
#undef MN_DPP
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}

// recompute the cached element (n > 0, keys pairwise different)
template <bool MAXQ> DEVI void ua_refresh(UArr &a) {
    const float m = ua_op<MAXQ>(ua_op<MAXQ>(a.k0, a.k1), ua_op<MAXQ>(a.k2, a.k3));
    const float ext = wave_extreme<MAXQ>(m);
    a.bk = ext;
    int pos = 0;
#define MN_SA_FIND(T, K, V)                                                                                                      \
    {                                                                                                                            \
        const unsigned long long mm = __builtin_amdgcn_ballot_w64(a.K == ext);                                                   \
        if (mm)                                                                                                                  \
            pos = T * 64 + (__ffsll((long long)mm) - 1);                                                                         \
    }
    MN_SA_EACH(MN_SA_FIND)
#undef MN_SA_FIND
    a.bp = __builtin_amdgcn_readfirstlane(pos);
}

// take the cached element out (the last element moves into its place)
template <bool MAXQ> DEVI void ua_remove_best(UArr &a, int lane) {
    const int last = a.n - 1;
    float lk;
    int lv;
    ua_get(a, last, lk, lv);
    ua_put(a, last, __builtin_nanf(""), 0, lane);
    if (a.bp != last)
        ua_put(a, a.bp, lk, lv, lane);
    a.n = last;
    if (last > 0)
        ua_refresh<MAXQ>(a);
    else
        a.bk = MAXQ ? -__builtin_inff() : __builtin_inff();
}

// beam_layer with the queues in registers.  true: `res` holds the results as beam_layer would have left them (and
// res.sorted = 1: position i is the (size - i)-th nearest).  false: a tie (or something else the arrays cannot decide) came
// up; nothing but the bitmap, the read log and the counters has been touched, and the caller redoes the layer with
// beam_layer.  `tmp`: LDS, MN_SA_CAP floats (the unused candidate heap).
template <int ORDER, int NCH, bool COH, bool WIDE>
DEVI bool beam_layer_regs(const MnDevIndex &ix, WaveCtx &w, WHeap &res, float *tmp, unsigned *bitmap, int entry, int level,
                          int ef, int lane) {
    UArr ca, ra; // candidates (cached: the nearest), results (cached: the worst)
    ua_init(ca, __builtin_inff());
    ua_init(ra, -__builtin_inff());
    bool ok = true;
    if (!(ix.has_deleted && ix.deleted[entry])) { // :360-366
        float d = ctx_distance<ORDER, NCH>(ix, w, entry, 1, lane);
        d = u2f(rflu(f2u(d)));
        w.n_dist += 1;
        ok = d == d;
        ua_put(ca, 0, d, entry, lane);
        ua_put(ra, 0, d, entry, lane);
        ca.n = ra.n = 1;
        ca.bk = ra.bk = d;
        if (lane == 0) {
            int vi = level == 0 ? entry : ix.up_off[entry];
            atomicOr(&bitmap[vi >> 5], 1u << (vi & 31));
        }
    }
    int patience_max = ef / 4; // :372-375
    if (patience_max < 10)
        patience_max = 10;
    int stale = 0;
    while (ok && ca.n > 0) {
        PH_DECL;
        const float cd = ca.bk;
        float ck_;
        int node;
        ua_get(ca, ca.bp, ck_, node);
        ua_remove_best<false>(ca, lane);
        if (ra.n >= ef && cd > ra.bk) // :382-386
            break;
        if (stale >= patience_max && ra.n >= ef) // :391
            break;
        int W;
        const int *row = link_row(ix, node, level, W);
        w.n_exp++;
        log_row_read(ix, w, node, level, lane);
        PH_ADD(w, 0);
        PH_CNT(w, 5, 1);
        int improved = 0;
        const int nchunk = WIDE ? (W + 63) >> 6 : 1;
        for (int ch = 0; ch < nchunk && ok; ch++) {
            const int c0 = WIDE ? ch << 6 : 0;
            int nb = (c0 + lane < W) ? ld_link<COH>(row + c0 + lane) : -1;
            bool todo = false;
            unsigned long long m;
            int n, myslot = 0;
            float d = 0.0f;
            const unsigned long long m_all = __ballot(nb >= 0);
            const int n_all = __popcll(m_all);
            if (w.coop && w.coop->nw > 1 && n_all > 2 && !getenv_spec_off(w)) {
                // Round 4: with helper wavefronts the rows of ALL the listed neighbours are requested before it is known which of
                // them are new — the visited probe (a returning atomic: one more round trip) then runs next to the row loads
                // instead of in front of them.  A wavefront's share is ≤ 4 rows either way (one pass of its tile), so the
                // distances of already-visited neighbours cost no time; they are simply not looked at.  Same distances for the
                // rows that count, same order.
                const CoopCtx &c = *w.coop;
                const int rank_all = __popcll(m_all & ((1ull << lane) - 1ull));
                __builtin_amdgcn_wave_barrier();
                if (nb >= 0)
                    c.list[rank_all] = nb;
                if (lane == 0)
                    *c.n = n_all;
                __syncthreads(); // the helpers start on their shares
                unsigned old = 0, bit = 0;
                if (nb >= 0) { // :403-409
                    int vi = level == 0 ? nb : ix.up_off[nb];
                    bit = 1u << (vi & 31);
                    old = atomicOr(&bitmap[vi >> 5], bit);
                }
                coop_share<ORDER, NCH>(ix, w.q, c, n_all, lane);
                __syncthreads();
                todo = nb >= 0 && !(old & bit) && !(ix.has_deleted && ix.deleted[nb]);
                const float d_mine = nb >= 0 ? c.dist[rank_all] : 0.0f;
                m = __ballot(todo);
                n = __popcll(m);
                PH_ADD(w, 1);
                if (n > 0) { // compact the new ones, in list order, with their distances
                    const int rank = __popcll(m & ((1ull << lane) - 1ull));
                    float *dtmp = reinterpret_cast<float *>(c.list); // (the helpers are done with the list)
                    __builtin_amdgcn_wave_barrier();
                    if (todo) {
                        w.scratch[rank] = nb;
                        dtmp[rank] = d_mine;
                    }
                    __builtin_amdgcn_wave_barrier();
                    myslot = lane < n ? w.scratch[lane] : 0;
                    d = lane < n ? dtmp[lane] : 0.0f;
                    __builtin_amdgcn_wave_barrier();
                }
            } else {
                if (nb >= 0) { // :403-409
                    int vi = level == 0 ? nb : ix.up_off[nb];
                    unsigned bit = 1u << (vi & 31);
                    unsigned old = atomicOr(&bitmap[vi >> 5], bit);
                    todo = !(old & bit) && !(ix.has_deleted && ix.deleted[nb]);
                }
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
                }
            }
            if (n > 0) {
                w.n_dist += n;
                PH_ADD(w, 2);
                unsigned long long am; // (worst only falls while the row is worked through: pre-filter, as beam_layer does)
                if (ra.n >= ef)
                    am = __ballot(lane < n && d < ra.bk);
                else
                    am = __ballot(lane < n);
                if (__ballot(lane < n && !(d == d)))
                    ok = false; // a NaN distance: the heaps' business
                PH_CNT(w, 4, __popcll(am));
                while (am && ok) { // :413-425, in list order
                    int i = __ffsll((long long)am) - 1;
                    am &= am - 1;
                    i = __builtin_amdgcn_readfirstlane(i);
                    const float di = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), i));
                    const int si = __builtin_amdgcn_readlane(myslot, i);
                    const bool full = ra.n >= ef;
                    if (full && !(di < ra.bk))
                        continue;
                    if (ua_holds(ca, di) || ua_holds(ra, di)) {
                        ok = false;
                        break;
                    }
                    // results first: either one more, or the worst is overwritten and the next worst looked up
                    if (!full) {
                        ua_put(ra, ra.n, di, si, lane);
                        if (di > ra.bk) {
                            ra.bk = di;
                            ra.bp = ra.n;
                        }
                        ra.n++;
                    } else {
                        ua_put(ra, ra.bp, di, si, lane);
                        ua_refresh<true>(ra);
                    }
                    if (ca.n == MN_SA_CAP) { // more candidates alive than the array holds: the farthest goes — it must be dead
                        UArr far = ca;
                        ua_refresh<true>(far);
                        if (!(ra.n >= ef && far.bk > ra.bk)) {
                            ok = false;
                            break;
                        }
                        ua_put(ca, far.bp, di, si, lane);
                        if (di < ca.bk) {
                            ca.bk = di;
                            ca.bp = far.bp;
                        }
                    } else {
                        ua_put(ca, ca.n, di, si, lane);
                        if (di < ca.bk) {
                            ca.bk = di;
                            ca.bp = ca.n;
                        }
                        ca.n++;
                    }
                    improved = 1;
                }
                PH_ADD(w, 3);
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
    // rank the results (keys pairwise different): element e goes to heap position n - rank(e); keys -distance, root = worst
    const int n = ra.n;
    __builtin_amdgcn_wave_barrier();
    tmp[lane] = ra.k0;
    tmp[64 + lane] = ra.k1;
    tmp[128 + lane] = ra.k2;
    tmp[192 + lane] = ra.k3;
    __builtin_amdgcn_wave_barrier();
    int r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    for (int j = 0; j < n; j += 4) { // (broadcast reads; NaN = empty never counts)
        const float4 kk = *reinterpret_cast<const float4 *>(tmp + j);
        r0 += (kk.x < ra.k0) + (kk.y < ra.k0) + (kk.z < ra.k0) + (kk.w < ra.k0);
        if (n > 64)
            r1 += (kk.x < ra.k1) + (kk.y < ra.k1) + (kk.z < ra.k1) + (kk.w < ra.k1);
        if (n > 128) {
            r2 += (kk.x < ra.k2) + (kk.y < ra.k2) + (kk.z < ra.k2) + (kk.w < ra.k2);
            r3 += (kk.x < ra.k3) + (kk.y < ra.k3) + (kk.z < ra.k3) + (kk.w < ra.k3);
        }
    }
    __builtin_amdgcn_wave_barrier();
    res.size = n;
    res.ovf = 0;
    res.sorted = 1;
    if (lane < n)
        hset(res, n - r0, make_uint2(f2u(-ra.k0), (unsigned)ra.v0));
    if (64 + lane < n)
        hset(res, n - r1, make_uint2(f2u(-ra.k1), (unsigned)ra.v1));
    if (128 + lane < n)
        hset(res, n - r2, make_uint2(f2u(-ra.k2), (unsigned)ra.v2));
    if (192 + lane < n)
        hset(res, n - r3, make_uint2(f2u(-ra.k3), (unsigned)ra.v3));
    __builtin_amdgcn_wave_barrier();
    return true;
}

// the layer search of the latency-bound paths: registers first, the reference's heaps when a tie has to be decided
template <int ORDER, int NCH, bool COH, bool WIDE>
DEVI void beam_layer_auto(const MnDevIndex &ix, WaveCtx &w, WHeap &cand, WHeap &res, unsigned *bitmap, long long bm_words,
                          int entry, int level, int ef, int lane) {
    if (ef <= MN_SA_CAP && ef >= 1 && ef + 1 < res.lcap + res.gcap) {
        const unsigned long long nd0 = w.n_dist, ne0 = w.n_exp;
        const int nr0 = w.nr;
        if (beam_layer_regs<ORDER, NCH, COH, WIDE>(ix, w, res, reinterpret_cast<float *>(cand.l), bitmap, entry, level, ef, lane))
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
    beam_layer<ORDER, NCH, COH, WIDE>(ix, w, cand, res, bitmap, entry, level, ef, lane);
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
