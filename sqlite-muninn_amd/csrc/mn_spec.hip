// mn_spec.hip — speculative execution of the reference's one-at-a-time insert (src/hnsw_algo.c:520-666).
//
// hnsw_insert is a chain: insert j searches the graph that inserts 0..j-1 left behind.  k_insert_seq (mn_seq.hip)
// honours that with a single wavefront.  Here a WINDOW of consecutive inserts has its searches run at once against
// the graph as it stands (k_beam<BUILD>, one wavefront each, every link row read is logged), and one workgroup then
// commits the window in order:
//   insert j is valid  ⇐  none of the link rows its search read was rewritten by inserts 0..j-1 of the window
//                          (rows carry the epoch of their last rewrite; the first insert of a window is always valid)
//                      or (round 4) every such rewrite provably leaves the search as it was: see spec_rewrite_is_harmless
// A valid insert's search result is exactly what the sequential algorithm would have computed, so its links are
// applied to the live rows exactly as hnsw_insert applies them (targets in list order; MN-RU prune, :601-646).  At the
// first invalid insert the window stops: the host restarts from there, so the prefix property — and with it the
// reference's graph, bit for bit — is preserved.  On a large index most of a window commits (a search reads ≈ 300 of
// N rows, an insert rewrites ≈ 33); on a small one the scheme degrades to one insert per round.
//
// Inside a commit, the targets of one (insert, layer) are distinct rows: the wavefronts of the workgroup prune them
// side by side.  Only a distance tie makes a prune look at other rows (mutual-neighbour counts, :613-616); that
// (insert, layer) is then redone by one wavefront in list order.
#include "mn_beam.hpp"
#include "mn_prune.hpp"

#define MN_SPEC_MAX_WAVES 8

struct MnSpecArgs {
    const int *slots; // [W] the window's nodes in insertion order
    int W;
    int nlev;        // frozen max_level + 1
    const int *sel;  // [W][nlev][M0]  first min(found, M_max) search results per layer (used only while W0 == M0 <= 64)
    const int *nsel; // [W][nlev]
    const int *readlog; // [W][readcap]
    int readcap;
    const int *nread; // [W]
    int *stamp0;      // [n_slots]     epoch of the last rewrite of the node's layer-0 row
    int *stampU;      // [n_pool_rows] same for upper-layer rows
    int *sidx0;       // [n_slots]     valid while stamp == epoch: where in saved_rows the row's list of the window's start is kept
    int *sidxU;       // [n_pool_rows] (-1: not kept, the save buffer was full)
    int *saved_rows;  // [MN_SPEC_SAVE_CAP][64]
    int epoch;
    // decisions taken ahead for every (insert, layer, target) by k_spec_prepare against the rows of the window's start — good for
    // as long as the target's row has not been rewritten in this window; index ((jj * nlev + l) * W0 + i)
    int *pre_act;     // 0 nothing, 1 append, 2 pruned row, 3 a tie came up
    int *pre_cnt;     // append position
    int *pre_row;     // [..][64] the pruned row
    int *why;         // [8] or null (MN_SPEC_TRACE): why inserts were judged stale — [1] log overflow, [2] more than 64 rewritten rows
                      // read, [3..6] see spec_rewrite_is_harmless; [7] windows that committed whole
    int *ncommit;     // out: inserts committed (>= 1)
    size_t wave_bytes; // LDS per wavefront
};

DEVI int *spec_row(const MnDevIndex &ix, int node, int level) {
    if (level == 0)
        return ix.links0 + (size_t)node * ix.W0;
    return ix.links_up + ((size_t)ix.up_off[node] + (level - 1)) * ix.WU;
}

// A row is about to be rewritten (one wavefront; v = its list as it stands, lane < W): the first time in a window its list
// — the one every search of the window read — is kept, then the row is stamped.
DEVI void spec_stamp(const MnDevIndex &ix, const MnSpecArgs &a, int node, int level, int v, int W, int *nsaved, int lane) {
    const size_t r = level == 0 ? (size_t)node : (size_t)(ix.up_off[node] + level - 1);
    int *p = (level == 0 ? a.stamp0 : a.stampU) + r;
    int *q = (level == 0 ? a.sidx0 : a.sidxU) + r;
    if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.epoch)
        return; // (uniform) rewritten before in this window: the kept list is the older one
    int k = 0;
    if (lane == 0)
        k = W <= 64 ? atomicAdd(nsaved, 1) : MN_SPEC_SAVE_CAP;
    k = __builtin_amdgcn_readfirstlane(k);
    if (k < MN_SPEC_SAVE_CAP) {
        if (lane < 64)
            a.saved_rows[(size_t)k * 64 + lane] = lane < W ? v : -1;
    } else {
        k = -1;
    }
    if (lane == 0) {
        __hip_atomic_store(q, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Round 4.  One log entry of insert jj's search names a row that an earlier insert of the window has rewritten since.  The
// search ran on the row's OLD list; the sequential algorithm would have run on the new one.  With distinct keys a beam search is
// a computation on sets (mn_beam.hpp, beam_layer_regs), and by induction over its expansions the two runs stay identical if at
// this row
//   * no neighbour that was REMOVED from the list could have been pushed when the search opened it (the log entry carries the
//     positions that were new and nearer than the worst result of that moment): a neighbour that was already visited, or was
//     evaluated and rejected, leaves no trace — if it turns up again in another row the worst result has only come nearer;
//   * every neighbour that was ADDED (one of the window's new nodes: nothing else is ever added) would have been rejected: its
//     distance to this insert's vector, computed here by the search's own code, is not below that worst result.
// A greedy descent's step (src/hnsw_algo.c:257-282) is simply taken again on the new list: from the same place, against the same
// distance to beat, it must move to the same neighbour at the same position (the scan of the next row starts after it), or
// nowhere as before.  Anything else — a search that met equal keys and went back to the heaps, a row of more than 64 links, a
// list that was not kept — carries the log's defaults (-inf / all positions) and invalidates the insert as before.
// One wavefront; qv = the insert's vector (LDS), qnorm as its search had it.  tmp: LDS, 64 ints.
// Returns 0 = harmless, else why not (MN_SPEC_TRACE counts them): 3 the log's defaults, 4 the old list was not kept or the row is
// wider than 64, 5 a removed neighbour could have been pushed, 6 an added node would have been pushed, 7 a greedy step would
// have gone elsewhere.
template <int ORDER, int NCH>
DEVI int spec_rewrite_is_harmless(const MnDevIndex &ix, const MnSpecArgs &a, const int *e, const float *qv, float qnorm, int *tmp,
                                  int lane) {
    const int r = e[0];
    if (e[4] == 1) { // a greedy step: take it again on the row as it is now — it must move to the same neighbour at the same place
        const int k = __hip_atomic_load((r >= 0 ? a.sidx0 + r : a.sidxU + (-r - 2)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int W = r >= 0 ? ix.W0 : ix.WU;
        if (k < 0 || W > 64)
            return 4;
        const float to_beat = __int_as_float(e[1]);
        const int from = e[2], moved_to = e[3];
        const int *cur_row = r >= 0 ? ix.links0 + (size_t)r * ix.W0 : ix.links_up + (size_t)(-r - 2) * ix.WU;
        const int c = lane < W ? ld_link<true>(cur_row + lane) : -1;
        const bool valid = lane >= from && c >= 0 && !(ix.has_deleted && ix.deleted[c >= 0 ? c : 0]);
        const unsigned long long m = __ballot(valid);
        const int n = __popcll(m);
        int now = -1;
        if (n > 0) {
            const int rank = __popcll(m & ((1ull << lane) - 1ull));
            __builtin_amdgcn_wave_barrier();
            if (valid) {
                tmp[rank] = c;
                tmp[64 + rank] = lane;
            }
            __builtin_amdgcn_wave_barrier();
            const int myslot = lane < n ? tmp[lane] : 0;
            const int mypos = lane < n ? tmp[64 + lane] : 0;
            __builtin_amdgcn_wave_barrier();
            const float d = rows_distance<ORDER, NCH>(ix, qv, qnorm, myslot, n, lane);
            const unsigned long long better = __ballot(lane < n && d < to_beat);
            if (better)
                now = __builtin_amdgcn_readlane(mypos, __ffsll((long long)better) - 1);
        }
        if (now != moved_to)
            return 7;
        if (now >= 0 && a.saved_rows[(size_t)k * 64 + now] != __builtin_amdgcn_readlane(c, now))
            return 7;
        return 0;
    }
    const float worst = __int_as_float(e[1]);
    const unsigned long long could = (unsigned long long)(unsigned)e[2] | ((unsigned long long)(unsigned)e[3] << 32);
    if (!(worst > -__builtin_inff())) // the defaults (or a NaN)
        return 3;
    const int k = __hip_atomic_load((r >= 0 ? a.sidx0 + r : a.sidxU + (-r - 2)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (k < 0)
        return 4;
    const int W = r >= 0 ? ix.W0 : ix.WU;
    if (W > 64)
        return 4;
    const int *cur_row = r >= 0 ? ix.links0 + (size_t)r * ix.W0 : ix.links_up + (size_t)(-r - 2) * ix.WU;
    const int o = lane < W ? a.saved_rows[(size_t)k * 64 + lane] : -1;
    const int c = lane < W ? ld_link<true>(cur_row + lane) : -1;
    bool o_stays = false, c_was = false;
    int o_now = 0; // where this lane's old neighbour sits in the list now
    for (int p = 0; p < W; p++) { // (lists of ≤ 64: all pairs)
        const int cp = __builtin_amdgcn_readlane(c, p), op = __builtin_amdgcn_readlane(o, p);
        if (o == cp) {
            o_stays = true;
            o_now = p;
        }
        c_was |= c == op;
    }
    const unsigned long long removed = __ballot(o >= 0 && !o_stays);
    if (removed & could)
        return 5;
    if (e[4] == 2) { // the heaps' search: the neighbours that could be pushed must still come in the order they came in
        const bool mine = (could >> lane) & 1ull;
        const int rk = __popcll(could & ((1ull << lane) - 1ull));
        __builtin_amdgcn_wave_barrier();
        if (mine)
            tmp[rk] = o_now;
        __builtin_amdgcn_wave_barrier();
        const bool bad = mine && rk > 0 && tmp[rk - 1] >= o_now;
        __builtin_amdgcn_wave_barrier();
        if (__ballot(bad))
            return 5;
    }
    const bool added = c >= 0 && !c_was;
    const unsigned long long am = __ballot(added);
    const int na = __popcll(am);
    if (na == 0)
        return 0;
    const int rank = __popcll(am & ((1ull << lane) - 1ull));
    __builtin_amdgcn_wave_barrier();
    if (added)
        tmp[rank] = c;
    __builtin_amdgcn_wave_barrier();
    const int myslot = lane < na ? tmp[lane] : 0;
    __builtin_amdgcn_wave_barrier();
    const float d = rows_distance<ORDER, NCH>(ix, qv, qnorm, myslot, na, lane);
    return __ballot(lane < na && !(d >= worst)) == 0 ? 0 : 6; // (a NaN distance fails too)
}

// What adding node s to target t's layer-l list does (src/hnsw_algo.c:590-646), decided by one wavefront from the row as it stands,
// nothing written: 0 nothing (t has no such layer, or s is in the list), 1 append at *where, 2 the list is full and the MN-RU
// prune's choice is in out_row[0..W), 3 the prune met a distance tie (it would have to look at other rows: list order decides).
template <int ORDER, int NCH>
DEVI int spec_decide(const MnDevIndex &ix, int s, int t, int l, int W, int *list, float *nd, int *mn, float *tv, int lane,
                     int *where, int *out_row) {
    if (ix.levels[t] < l) // :590
        return 0;
    int *trow = spec_row(ix, t, l);
    const int v = lane < W ? ld_link<true>(trow + lane) : -1;
    const int cnt = __popcll(__ballot(v >= 0));
    if (__ballot(v == s)) // already a neighbour (:147-150)
        return 0;
    if (cnt < W) {
        *where = cnt;
        return 1;
    }
    // over-full: MN-RU prune of t's list (:601-646)
    __builtin_amdgcn_wave_barrier();
    if (lane < W)
        list[lane] = v;
    if (lane == 0)
        list[W] = s;
    const float *tsrc = ix.vectors + (size_t)t * ix.ld;
    for (int e = lane; e < ix.ld; e += 64)
        tv[e] = tsrc[e];
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    const float tnorm = ix.metric == 1 ? ix.norms[t] : 0.0f;
    if (prune_row<ORDER, NCH, true, false>(ix, tv, tnorm, list, nd, mn, W + 1, W, l, lane))
        return 3;
    if (lane < W)
        out_row[lane] = list[lane];
    return 2;
}

// Round 4: the decisions of a whole window at once, one workgroup per insert — the prunes are most of a commit's time, the
// inserts of a window mostly touch different rows, and a decision depends on nothing but the target's row and the vectors.  The
// commit kernel takes a decision from here when the target's row still is what it was, and decides itself when it is not.
template <int ORDER, int NCH>
__global__ void __launch_bounds__(MN_SPEC_MAX_WAVES * 64) k_spec_prepare(MnDevIndex ix, MnSpecArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, NW = blockDim.x >> 6;
    unsigned char *pw = smem + (size_t)wv * a.wave_bytes;
    int *list = reinterpret_cast<int *>(pw);
    float *nd = reinterpret_cast<float *>(list + 128);
    int *mn = reinterpret_cast<int *>(nd + 128);
    float *tv = reinterpret_cast<float *>(mn + 128);
    const int jj = blockIdx.x;
    const int s = a.slots[jj];
    const int level = ix.levels[s];
    const int start = level < a.nlev - 1 ? level : a.nlev - 1;
    for (int l = start; l >= 0; l--) {
        const int W = l == 0 ? ix.W0 : ix.WU;
        const int ns = a.nsel[jj * a.nlev + l];
        const size_t base = ((size_t)jj * a.nlev + l) * ix.W0;
        const int *sel = a.sel + base;
        for (int i = wv; i < ns; i += NW) {
            int where = 0;
            const int what = spec_decide<ORDER, NCH>(ix, s, sel[i], l, W, list, nd, mn, tv, lane, &where, a.pre_row + (base + i) * 64);
            if (lane == 0) {
                a.pre_act[base + i] = what;
                a.pre_cnt[base + i] = where;
            }
        }
    }
}

template <int ORDER, int NCH>
__global__ void __launch_bounds__(MN_SPEC_MAX_WAVES * 64) k_spec_commit(MnDevIndex ix, MnSpecArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, NW = blockDim.x >> 6;
    int *flag = reinterpret_cast<int *>(smem); // [0] invalid insert  [1] tie in this (insert, layer)  [2] rewritten rows read  [3] rows kept
    int *act = flag + 16;                      // [64] per target: 0 nothing, 1 append, 2 pruned row
    int *cntA = act + 64;                      // [64] append position
    int *hot = cntA + 64;                      // [64] log entries of the insert under test whose row has been rewritten
    int *newrow = hot + 64;                    // [64][64]
    unsigned char *pw = reinterpret_cast<unsigned char *>(newrow + 64 * 64) + (size_t)wv * a.wave_bytes;
    int *list = reinterpret_cast<int *>(pw);            // [128]
    float *nd = reinterpret_cast<float *>(list + 128);  // [128]
    int *mn = reinterpret_cast<int *>(nd + 128);        // [128]
    float *tv = reinterpret_cast<float *>(mn + 128);    // [ld]

    int done = a.W;
    if (tid == 0)
        flag[3] = 0;
    __syncthreads();
    for (int jj = 0; jj < a.W; jj++) {
        const int s = a.slots[jj];
        // ── did an earlier insert of this window rewrite a row this search read? ──
        if (tid == 0)
            flag[0] = 0;
        __syncthreads();
        if (tid == 0)
            flag[2] = 0; // rewritten rows among those the search read
        __syncthreads();
        if (jj > 0) {
            const int nr = a.nread[jj];
            if (nr > a.readcap) {
                if (tid == 0) {
                    flag[0] = 1; // incomplete log: cannot be trusted
                    if (a.why)
                        atomicAdd(&a.why[1], 1);
                }
            } else {
                const int *log = a.readlog + (size_t)jj * a.readcap * MN_RLOG_INTS;
                for (int i = tid; i < nr; i += blockDim.x) {
                    const int r = log[(size_t)i * MN_RLOG_INTS];
                    const int *p = r >= 0 ? a.stamp0 + r : a.stampU + (-r - 2);
                    if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.epoch) {
                        const int h = atomicAdd(&flag[2], 1);
                        if (h < 64)
                            hot[h] = i;
                        else {
                            flag[0] = 1; // (more than the check looks at)
                            if (a.why && h == 64)
                                atomicAdd(&a.why[2], 1);
                        }
                    }
                }
            }
        }
        __syncthreads();
        if (!flag[0] && flag[2] > 0) { // (uniform) would those rewrites have changed this search?
            const int nh = flag[2];
            const int *log = a.readlog + (size_t)jj * a.readcap * MN_RLOG_INTS;
            if (wv < nh) {
                const float *src = ix.vectors + (size_t)s * ix.ld;
                for (int e = lane; e < ix.ld; e += 64)
                    tv[e] = src[e];
                __builtin_amdgcn_s_waitcnt(0);
                __builtin_amdgcn_wave_barrier();
                const float qnorm = ix.metric == 1 ? ix.norms[s] : 0.0f;
                for (int h = wv; h < nh; h += NW) {
                    const int why = spec_rewrite_is_harmless<ORDER, NCH>(ix, a, log + (size_t)hot[h] * MN_RLOG_INTS, tv, qnorm, list, lane);
                    if (why) {
                        if (lane == 0) {
                            flag[0] = 1;
                            if (a.why)
                                atomicAdd(&a.why[why == 7 ? 0 : why], 1); // ([7] counts the windows that committed whole)
                        }
                        break;
                    }
                }
            }
            __syncthreads();
        }
        if (flag[0]) { // uniform over the workgroup
            done = jj;
            break;
        }
        const int level = ix.levels[s];
        const int start = level < a.nlev - 1 ? level : a.nlev - 1;
        for (int l = start; l >= 0; l--) { // src/hnsw_algo.c:572-653
            const int W = l == 0 ? ix.W0 : ix.WU;
            const int ns = a.nsel[jj * a.nlev + l];
            const int *sel = a.sel + ((size_t)jj * a.nlev + l) * ix.W0;
            int *srow = spec_row(ix, s, l);
            if (tid == 0)
                flag[1] = 0;
            __syncthreads();
            // pass 1: every target's new row — taken from k_spec_prepare while the target's row is the one it saw, decided here
            // otherwise; nothing is written yet
            for (int i = wv; i < ns; i += NW) {
                const int t = sel[i];
                int what = 0, where = 0;
                bool taken = false;
                if (a.pre_act && ix.levels[t] >= l) {
                    const size_t r = l == 0 ? (size_t)t : (size_t)(ix.up_off[t] + l - 1);
                    if (__hip_atomic_load((l == 0 ? a.stamp0 : a.stampU) + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.epoch) {
                        const size_t idx = ((size_t)jj * a.nlev + l) * ix.W0 + i;
                        what = a.pre_act[idx];
                        where = a.pre_cnt[idx];
                        if (what == 2 && lane < W)
                            newrow[i * 64 + lane] = a.pre_row[idx * 64 + lane];
                        taken = true;
                    }
                }
                if (!taken)
                    what = spec_decide<ORDER, NCH>(ix, s, t, l, W, list, nd, mn, tv, lane, &where, newrow + i * 64);
                if (what == 3) {
                    what = 0;
                    if (lane == 0)
                        flag[1] = 1; // a tie: this layer is redone in list order below
                }
                if (lane == 0) {
                    act[i] = what;
                    cntA[i] = where;
                }
            }
            __syncthreads();
            if (!flag[1]) {
                // pass 2: apply
                for (int i = wv; i < ns; i += NW) {
                    const int t = sel[i];
                    if (lane == 0) {
                        st_link(srow + i, t); // node_add_neighbor(new_node, l, selected[i])
                        ix.dirty[t] = 1;      // persist set (src/hnsw_vtab.c:761-768)
                    }
                    if (act[i] == 0)
                        continue;
                    int *trow = spec_row(ix, t, l);
                    spec_stamp(ix, a, t, l, lane < W ? ld_link<true>(trow + lane) : -1, W, &flag[3], lane);
                    if (act[i] == 1) {
                        if (lane == 0)
                            st_link(trow + cntA[i], s);
                    } else if (lane < W) {
                        st_link(trow + lane, newrow[i * 64 + lane]);
                    }
                }
            } else if (wv == 0) {
                // the layer in list order by one wavefront, as k_insert_seq does it
                for (int i = 0; i < ns; i++) {
                    const int t = sel[i];
                    if (lane == 0) {
                        st_link(srow + i, t);
                        ix.dirty[t] = 1;
                    }
                    if (ix.levels[t] < l)
                        continue;
                    int *trow = spec_row(ix, t, l);
                    const int v = lane < W ? ld_link<true>(trow + lane) : -1;
                    const int cnt = __popcll(__ballot(v >= 0));
                    if (__ballot(v == s))
                        continue;
                    if (cnt < W) {
                        spec_stamp(ix, a, t, l, v, W, &flag[3], lane);
                        if (lane == 0)
                            st_link(trow + cnt, s);
                        continue;
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (lane < W)
                        list[lane] = v;
                    if (lane == 0)
                        list[W] = s;
                    const float *tsrc = ix.vectors + (size_t)t * ix.ld;
                    for (int e = lane; e < ix.ld; e += 64)
                        tv[e] = tsrc[e];
                    __builtin_amdgcn_s_waitcnt(0);
                    __builtin_amdgcn_wave_barrier();
                    const float tnorm = ix.metric == 1 ? ix.norms[t] : 0.0f;
                    prune_row<ORDER, NCH, true, true>(ix, tv, tnorm, list, nd, mn, W + 1, W, l, lane);
                    spec_stamp(ix, a, t, l, v, W, &flag[3], lane);
                    if (lane < W)
                        st_link(trow + lane, list[lane]);
                    __builtin_amdgcn_s_waitcnt(0);
                    __builtin_amdgcn_wave_barrier();
                }
            }
            __threadfence();
            __syncthreads();
        }
    }
    if (tid == 0) {
        *a.ncommit = done;
        if (a.why && done == a.W)
            atomicAdd(&a.why[7], 1);
    }
}

static int pick_nch_p(int ld) {
    int need = (ld + 255) / 256;
    if (need <= 1) return 1;
    if (need <= 2) return 2;
    if (need <= 3) return 3;
    if (need <= 4) return 4;
    if (need <= 6) return 6;
    if (need <= 8) return 8;
    return 0;
}

// one workgroup; as many wavefronts as 60 KB of LDS allow (≤ 8)
void mn_launch_spec_commit(const MnDevIndex &ix, const int *d_slots, int W, int nlev, const int *sel, const int *nsel,
                           const int *readlog, int readcap, const int *nread, int *stamp0, int *stampU, int *sidx0, int *sidxU,
                           int *saved_rows, int *pre_act, int *pre_cnt, int *pre_row, int *why, int epoch, int *d_ncommit,
                           hipStream_t st) {
    MnSpecArgs a;
    a.slots = d_slots;
    a.W = W;
    a.nlev = nlev;
    a.sel = sel;
    a.nsel = nsel;
    a.readlog = readlog;
    a.readcap = readcap;
    a.nread = nread;
    a.stamp0 = stamp0;
    a.stampU = stampU;
    a.sidx0 = sidx0;
    a.sidxU = sidxU;
    a.saved_rows = saved_rows;
    a.pre_act = pre_act;
    a.pre_cnt = pre_cnt;
    a.pre_row = pre_row;
    a.why = why;
    a.epoch = epoch;
    a.ncommit = d_ncommit;
    a.wave_bytes = (3 * 128 * sizeof(int) + (size_t)ix.ld * sizeof(float) + 15) & ~(size_t)15;
    const size_t shared = (16 + 64 + 64 + 64 + 64 * 64) * sizeof(int);
    int nw = (int)((60 * 1024 - shared) / a.wave_bytes);
    nw = nw < 1 ? 1 : (nw > MN_SPEC_MAX_WAVES ? MN_SPEC_MAX_WAVES : nw);
    const size_t lds = shared + (size_t)nw * a.wave_bytes;
    const size_t lds_pre = (size_t)nw * a.wave_bytes;
#define MN_SP(O, N)                                                                                                              \
    do {                                                                                                                         \
        if (a.pre_act)                                                                                                           \
            hipLaunchKernelGGL((k_spec_prepare<O, N>), dim3(W), dim3(nw * 64), lds_pre, st, ix, a);                              \
        hipLaunchKernelGGL((k_spec_commit<O, N>), dim3(1), dim3(nw * 64), lds, st, ix, a);                                       \
    } while (0)
    if (ix.order == MN_ORDER_SSE_V) {
        MN_SP(MN_ORDER_SSE_V, 0);
        return;
    }
    switch (pick_nch_p(ix.ld)) {
    case 1: MN_SP(MN_ORDER_WAVE_V, 1); break;
    case 2: MN_SP(MN_ORDER_WAVE_V, 2); break;
    case 3: MN_SP(MN_ORDER_WAVE_V, 3); break;
    case 4: MN_SP(MN_ORDER_WAVE_V, 4); break;
    case 6: MN_SP(MN_ORDER_WAVE_V, 6); break;
    case 8: MN_SP(MN_ORDER_WAVE_V, 8); break;
    default: MN_SP(MN_ORDER_WAVE_V, 0); break;
    }
#undef MN_SP
}

// HIP loads a translation unit's code object on the first use of one of its kernels (several milliseconds for these units): an
// index asks for all of them when it is created (mn_index.hip), so that the first query or insert of a process does not pay.
void mn_module_touch_spec() {
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&k_spec_commit<MN_ORDER_SSE_V, 0>));
}
