// mn_graph.hip — Leiden community detection (src/graph_community.c:75-429) over device-resident CSR
// adjacency (src/graph_csr.h:27-34), gfx950.
//
// The reference sweeps nodes in order and applies every move immediately (:158-228); all arithmetic
// is f64 and every per-community sum is taken in adjacency-list order.  Device design:
//   best_move        one wavefront evaluates one node: its edges' (community, weight) pairs are staged
//                    in LDS in list order; lane e decides whether edge e is the first occurrence of
//                    its community, sums that community's weights in list order, computes the gain
//                    expression of :209-210 verbatim in f64 and a max-with-lowest-index reduction
//                    reproduces the strict `gain > best_gain` first-seen tie rule (:212)
//   k_leiden_seq     MN_LEIDEN_SEQUENTIAL: ONE wavefront walks v = 0..N-1 with in-place updates
//                    (agent-scope atomics for label/sum_tot: never a stale L1 line) — community
//                    assignment and Q bit-identical to the reference
//   k_leiden_eval_sg / _big / win / apply   MN_LEIDEN_BATCHED: a range of nodes evaluated in parallel against
//                    frozen state; a mover commits iff it is the smallest-index mover among the
//                    movers touching its old/target community and its moving neighbours → committed
//                    moves are pairwise independent, realise exactly their computed gain (Q strictly
//                    increases) and the result does not depend on execution order.
//                    The evaluation is a chain of dependent gathers per node (offsets → targets → labels →
//                    sum_tot), so what it needs is nodes in flight: a wavefront evaluates 64/SG nodes at once in
//                    SG-lane sub-groups (3.3 KB of LDS per wavefront whatever the largest degree is); the few
//                    nodes with more than LEI_SG_CAP edges take a whole wavefront each (k_leiden_eval_big over a
//                    precomputed list).  The movers' per-community tallies are folded into the evaluation.
// O(N) bookkeeping between phases (renumber :317-331, distinct counts :388-403, sum_tot rebuild
// :413-416, m :344-350, the per-community accumulation of :128-139) stays on the device for unweighted graphs —
// every such sum is then a sum of integers below 2^53, exact in any order, so atomics give the reference's bits;
// first-seen renumbering = atomicMin of the first index per label + a prefix sum over the first-occurrence flags.
// Weighted graphs keep those sums on the host in the reference's node order (f64 addition is not associative).
// All device buffers are allocated once per graph (LeiWork) and reused by later calls.
#include "../../include/muninn_hip.h"
#include "mn_guard.hpp"
#include "mn_comm.hpp"
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#define DEVI __device__ __forceinline__
#define LEI_CAP 1024 // edges of one node staged in LDS; larger nodes use the global scratch path
#ifndef LEI_SG_CAP
#define LEI_SG_CAP 64 // sub-group evaluation: edges staged per node (more → the node takes a whole wavefront); power of two
#endif
// ints per sub-group: table of 1 << log2h entries (key, count, first position) + counter (4 ints) + 16-bit slot list (one per
// edge at most).  log2h is LEI_SG_LOG2H (two entries per edge) while most labels are still different — the first sweeps of a
// phase — and one less afterwards (round 4): the table then takes half the LDS and 32 wavefronts fit a CU instead of 20.  Only
// communities other than the node's own enter it, at most one per edge, so it cannot overflow at either size.
#define LEI_SG_AREA_OF(log2h) (3 * (1 << (log2h)) + 4 + LEI_SG_CAP / 2)
#define LEI_WPB 4 // wavefronts per k_leiden_eval workgroup (at most; power of two)
#define LEI_SG_LOG2H (LEI_SG_CAP == 32 ? 6 : LEI_SG_CAP == 64 ? 7 : LEI_SG_CAP == 128 ? 8 : 9) // table of 2 * LEI_SG_CAP entries

struct DevGraph {
    int n;
    const int *off_out, *tgt_out;
    const double *w_out; // null = 1.0
    const int *off_in, *tgt_in;
    const double *w_in;
};

template <bool COH> DEVI int ld_i(const int *p) {
    if (COH)
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <bool COH> DEVI double ld_d(const double *p) {
    if (COH)
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}

// Returns the community node v should move to (== its current one if no strictly positive gain).
// ec/ew/el: staging for the node's edges (community, weight, eligible) in list order — LDS for degree
// <= LEI_CAP, global scratch otherwise; both 16-byte aligned, capacity rounded up to a multiple of 4.
template <bool COH>
DEVI int best_move(const DevGraph &g, int v, const int *label, const double *sum_tot, const double *kdeg, double m,
                   double resolution, int use_both, const int *elig_part, int *ec, double *ew, unsigned char *el, int lane,
                   double *dk_out = nullptr, int pickless = 0) {
    const int o0 = g.off_out[v], d_out = g.off_out[v + 1] - o0;
    const int i0 = use_both ? g.off_in[v] : 0, d_in = use_both ? g.off_in[v + 1] - i0 : 0;
    const int d = d_out + d_in;
    const int d4 = (d + 3) & ~3;
    const int old = ld_i<COH>(label + v);
    const int mypart = elig_part ? elig_part[v] : 0;
    __builtin_amdgcn_wave_barrier();
    if (COH) {
        for (int e = lane; e < d4; e += 64) {
            int c = -2; // padding never matches a community
            double w = 0.0;
            unsigned char ok = 0;
            if (e < d) {
                int t;
                if (e < d_out) {
                    t = g.tgt_out[o0 + e];
                    w = g.w_out ? g.w_out[o0 + e] : 1.0;
                } else {
                    t = g.tgt_in[i0 + (e - d_out)];
                    w = g.w_in ? g.w_in[i0 + (e - d_out)] : 1.0;
                }
                c = ld_i<COH>(label + t);
                ok = (!elig_part || elig_part[t] == mypart) ? 1 : 0;
            }
            ec[e] = c;
            ew[e] = w;
            el[e] = ok;
        }
    } else {
        // parallel rounds: four edges per lane in flight — every load unconditional (index clamped to the last edge), the four
        // targets and weights go out back to back, then the four labels (and partitions): two round trips per 256 edges
        // instead of two per 64 (the guarded one-edge-per-lane loop above is one s_waitcnt per load)
        const bool has_w = g.w_out != nullptr || g.w_in != nullptr;
        for (int e0 = 0; e0 < d4; e0 += 256) {
            int t[4], c[4], part[4];
            double w[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int ecl = max(0, min(e0 + j * 64 + lane, d - 1));
                const bool out = ecl < d_out;
                t[j] = *(out ? g.tgt_out + o0 + ecl : g.tgt_in + i0 + (ecl - d_out));
                w[j] = 1.0;
                if (has_w) {
                    const double *pw = out ? g.w_out : g.w_in;
                    w[j] = pw ? pw[(out ? o0 + ecl : i0 + (ecl - d_out))] : 1.0;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++)
                c[j] = label[t[j]];
            if (elig_part) {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    part[j] = elig_part[t[j]];
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int e = e0 + j * 64 + lane;
                if (e >= d4)
                    continue;
                const bool in = e < d;
                ec[e] = in ? c[j] : -2;
                ew[e] = in ? w[j] : 0.0;
                el[e] = in && (!elig_part || part[j] == mypart) ? 1 : 0;
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    const double k_v = kdeg[v];
    const int4 *ec4 = reinterpret_cast<const int4 *>(ec);
    const double2 *ew2 = reinterpret_cast<const double2 *>(ew);
    const uchar4 *el4 = reinterpret_cast<const uchar4 *>(el);
    double k_v_to_old = 0.0; // weight_to_community(v, old), :163 — list order
    for (int q = 0; q < (d4 >> 2); q++) {
        const int4 cj = ec4[q];
        const double2 wa = ew2[2 * q], wb = ew2[2 * q + 1];
        if (cj.x == old) k_v_to_old += wa.x;
        if (cj.y == old) k_v_to_old += wa.y;
        if (cj.z == old) k_v_to_old += wb.x;
        if (cj.w == old) k_v_to_old += wb.y;
    }
    const double st_old = ld_d<COH>(sum_tot + old);
    double best_gain = 0.0;
    int best = old;
    for (int base = 0; base < d; base += 64) {
        const int e = base + lane;
        const int c = e < d ? ec[e] : -3;
        bool cand = e < d && el[e] && c != old && !(pickless && c > old);
        // one pass over the list: the in-order weight sum of community c (weight_to_community, :206) and
        // "an eligible earlier edge already carries c" (the dedup scan of :173-199)
        double sacc = 0.0;
        bool dup = false;
        for (int q = 0; q < (d4 >> 2); q++) {
            const int4 cj = ec4[q];
            const double2 wa = ew2[2 * q], wb = ew2[2 * q + 1];
            const uchar4 ej = el4[q];
            const int j = q << 2;
            if (cj.x == c) { sacc += wa.x; dup |= (j < e) && ej.x; }
            if (cj.y == c) { sacc += wa.y; dup |= (j + 1 < e) && ej.y; }
            if (cj.z == c) { sacc += wb.x; dup |= (j + 2 < e) && ej.z; }
            if (cj.w == c) { sacc += wb.y; dup |= (j + 3 < e) && ej.w; }
        }
        cand = cand && !dup;
        double gain = -1.0, dk = 0.0;
        if (cand) {
            const double st_c = ld_d<COH>(sum_tot + c);
            dk = sacc - k_v_to_old;
            gain = (sacc - k_v_to_old) / m + resolution * k_v * (st_old - k_v - st_c) / (2.0 * m * m); // :209-210
            if (!(gain > 0.0))
                gain = -1.0; // also drops NaN: `gain > best_gain` is false for it
        }
        // max gain, lowest lane on ties == first candidate with the strictly largest gain
        double bg = gain;
        int bl = lane;
        for (int mk = 32; mk >= 1; mk >>= 1) {
            double og = __shfl_xor(bg, mk);
            int ol = __shfl_xor(bl, mk);
            if (og > bg || (og == bg && ol < bl)) {
                bg = og;
                bl = ol;
            }
        }
        if (bg > best_gain) { // strict: an equal gain in a later chunk does not replace (:212)
            best_gain = bg;
            best = __shfl(c, bl);
            double bdk = __shfl(dk, bl);
            if (dk_out)
                *dk_out = bdk;
        }
    }
    return best;
}


// ── unweighted graphs: O(degree) evaluation ──
// Every weight is 1.0, so weight_to_community(v, c) (:75-90) is the NUMBER of v's edges into c — an integer, exact in
// any order — and the O(deg · #neighbour communities) rescans of the reference (and of best_move above, which keeps
// them because weighted sums must be taken in list order) collapse into one pass: each edge is dropped into a small
// open-addressing table in LDS keyed by community (count, position of the first eligible edge).  The candidates are
// the occupied entries; the strict-gain first-seen rule (:212) = highest gain, lowest first position.  Same decisions
// as best_move, bit for bit (the gain expression is evaluated on the same f64 operands).
#define LEI_EMPTY (-1)
DEVI unsigned lei_hash(int c, int log2h) { return ((unsigned)c * 2654435761u) >> (32 - log2h); }

// SG lanes (lane = absolute lane, sl = lane % SG) evaluate node v; tk/tc/tp: the group's table of H = 1 << log2h entries
// everything about node v whose address depends on v alone: requested together, one round trip
struct LeiHead {
    int o0, d_out, i0, d_in, old, mypart;
    double k_v;
};
DEVI LeiHead lei_head(const DevGraph &g, int v, const int *label, const double *kdeg, int use_both, const int *elig_part) {
    LeiHead h;
    const int o0 = g.off_out[v], o1 = g.off_out[v + 1];
    const int i0 = g.off_in[use_both ? v : 0], i1 = g.off_in[use_both ? v + 1 : 0]; // (unconditional loads)
    h.old = label[v];
    h.mypart = (elig_part ? elig_part : label)[v];
    h.k_v = kdeg[v];
    h.o0 = o0;
    h.d_out = o1 - o0;
    h.i0 = use_both ? i0 : 0;
    h.d_in = use_both ? i1 - i0 : 0;
    return h;
}

// Lane exchange for the reductions inside a sub-group, without the LDS crossbar (__shfl_xor = ds_bpermute, and the SQ counters of
// round 4 put this kernel's LDS pipe at ≈ 88 % busy): DPP inside a row of 16 lanes — quad exchanges, then the mirrors (any pairing
// that ends with every lane holding the row's result serves a commutative, associative reduction) —, shuffles only above 16.
template <int STEP> DEVI int lei_peer(int v) {
    if (STEP == 0) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
    if (STEP == 1) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);  // quad_perm [2,3,0,1]
    if (STEP == 2) return __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false); // row_half_mirror
    if (STEP == 3) return __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false); // row_mirror
    return __shfl_xor(v, STEP == 4 ? 16 : 32);
}
template <int STEP> DEVI double lei_peer(double v) {
    return __hiloint2double(lei_peer<STEP>(__double2hiint(v)), lei_peer<STEP>(__double2loint(v)));
}
struct LeiBest {
    double gain, dk;
    int pos, c;
};
template <int STEP> DEVI void lei_best_step(LeiBest &b) {
    const double og = lei_peer<STEP>(b.gain), od = lei_peer<STEP>(b.dk);
    const int op = lei_peer<STEP>(b.pos), oc = lei_peer<STEP>(b.c);
    if (og > b.gain || (og == b.gain && op < b.pos)) {
        b.gain = og;
        b.pos = op;
        b.c = oc;
        b.dk = od;
    }
}

// Round 4 (the kernel is bound by the LDS pipe): (i) only the KEYS are cleared, four per ds_write_b128 — whoever inserts a key
// initialises its count and position (LDS operations of one wavefront execute in program order, and the additions come after the
// probe loop); (ii) edges into the node's own community, most of them once communities have formed, never touch the table:
// weight_to_community(v, old) is counted in registers and reduced across the sub-group by DPP; (iii) the closing max-reduction
// runs on DPP as well.  Same decisions, bit for bit.
template <int SG>
DEVI int best_move_hash(const DevGraph &g, const LeiHead &hd, const int *label, const double *sum_tot, double m,
                        double resolution, const int *elig_part, int *tk, int *tc, int *tp, int *cl, int log2h, int lane,
                        int sl, double *dk_out, int pickless) {
    const int H = 1 << log2h;
    const int o0 = hd.o0, d_out = hd.d_out, i0 = hd.i0, d_in = hd.d_in;
    const int d = d_out + d_in;
    const int old = hd.old;
    const int mypart = elig_part ? hd.mypart : 0;
    // issued now, consumed after the table is built: this round trip overlaps the targets → labels chain
    const double k_v = hd.k_v;
    const double st_old = sum_tot[old];
    for (int j = 4 * sl; j < H; j += 4 * SG)
        *reinterpret_cast<int4 *>(tk + j) = make_int4(LEI_EMPTY, LEI_EMPTY, LEI_EMPTY, LEI_EMPTY);
    // cl[0]: number of occupied entries; then their slots (16-bit), appended by whoever inserts a key — the candidate
    // scan below visits the occupied entries only (a handful once communities have formed), not the whole table
    unsigned short *clist = reinterpret_cast<unsigned short *>(cl + 4);
    if (sl == 0)
        cl[0] = 0;
    __builtin_amdgcn_wave_barrier();
    int n_old = 0; // this lane's edges into the node's own community
    // Four edges per lane at a time, every load unconditional (index clamped to the last edge): the four targets go out
    // back to back, then their four labels (and partitions) — two round trips per 4·SG edges instead of two per SG edges.
    for (int e0 = 0; e0 < d; e0 += 4 * SG) {
        int t[4], c[4], part[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int ec = min(e0 + j * SG + sl, d - 1);
            const int *pt = ec < d_out ? g.tgt_out + o0 + ec : g.tgt_in + i0 + (ec - d_out);
            t[j] = *pt;
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
            c[j] = label[t[j]];
        if (elig_part) {
#pragma unroll
            for (int j = 0; j < 4; j++)
                part[j] = elig_part[t[j]];
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++)
                part[j] = mypart;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int e = e0 + j * SG + sl;
            if (e >= d)
                continue;
            if (c[j] == old) {
                n_old++;
                continue;
            }
            // refinement: a target in another phase-1 community carries a refined label that no candidate of this node can
            // have (a refined community lies inside one phase-1 community, and candidates need an eligible edge) and that is
            // not `old` either: it changes no count that is looked at — it does not enter the table
            if (part[j] != mypart)
                continue;
            unsigned h = lei_hash(c[j], log2h);
            for (;;) {
                int prev = tk[h]; // (a plain read first: most edges find their community already inserted)
                if (prev == LEI_EMPTY) {
                    prev = atomicCAS(&tk[h], LEI_EMPTY, c[j]);
                    if (prev == LEI_EMPTY) {
                        tc[h] = 0;
                        tp[h] = 0x7fffffff;
                        clist[atomicAdd(&cl[0], 1)] = (unsigned short)h;
                    }
                }
                if (prev == LEI_EMPTY || prev == c[j])
                    break;
                h = (h + 1) & (H - 1);
            }
            atomicAdd(&tc[h], 1);
            atomicMin(&tp[h], e);
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    n_old += lei_peer<0>(n_old);
    n_old += lei_peer<1>(n_old);
    n_old += lei_peer<2>(n_old);
    n_old += lei_peer<3>(n_old);
    if (SG > 16)
        n_old += lei_peer<4>(n_old);
    if (SG > 32)
        n_old += lei_peer<5>(n_old);
    const double k_v_to_old = (double)n_old; // weight_to_community(v, old), :163
    LeiBest b = {-1.0, 0.0, 0x7fffffff, old};
    // candidates = the occupied entries, one per lane per pass (max gain, ties to the lowest first position: any order)
    const int ncand = cl[0];
    for (int i = sl; i < ncand; i += SG) {
        const int slot = clist[i];
        const int c = tk[slot], pos = tp[slot];
        if (pos == 0x7fffffff || (pickless && c > old))
            continue;
        const double sacc = (double)tc[slot];
        const double st_c = sum_tot[c];
        double gain = (sacc - k_v_to_old) / m + resolution * k_v * (st_old - k_v - st_c) / (2.0 * m * m); // :209-210
        if (!(gain > 0.0))
            continue;
        if (gain > b.gain || (gain == b.gain && pos < b.pos)) {
            b.gain = gain;
            b.pos = pos;
            b.c = c;
            b.dk = sacc - k_v_to_old;
        }
    }
    lei_best_step<0>(b);
    lei_best_step<1>(b);
    lei_best_step<2>(b);
    lei_best_step<3>(b);
    if (SG > 16)
        lei_best_step<4>(b);
    if (SG > 32)
        lei_best_step<5>(b);
    *dk_out = b.gain > 0.0 ? b.dk : 0.0;
    return b.gain > 0.0 ? b.c : old;
}

// ── weighted graphs: O(degree · distinct communities / lanes) evaluation (round 4) ──
// weight_to_community(v, c) (:75-90) is an f64 sum in LIST ORDER, so the hash-and-count shortcut of the unweighted path does not
// apply — but the sum does not have to be taken once per EDGE (best_move: every lane sums the community of its own edge, most of
// them the same few communities: O(degree²) compare-and-add steps, 70 % of a weighted run).  Here every edge is hashed to its
// community's table slot (as in best_move_hash; the slot remembers the first eligible position), communities are numbered in
// the order they were inserted, and lane p OWNS community number p: one walk over the staged list in order, adding the weights
// of the edges whose community number is p — the reference's additions in the reference's order, once per distinct community.
// Sixteen lanes cover sixteen communities per walk; once communities have formed a node's neighbours lie in a handful.
DEVI size_t lei_wslots_bytes(int cap, int log2h) { return (size_t)20 * cap + (size_t)10 * (1 << log2h) + 16; }
template <int SG>
DEVI int best_move_wslots(const DevGraph &g, const LeiHead &hd, const int *label, const double *sum_tot, double m, double resolution,
                          const int *elig_part, unsigned char *area, int cap, int log2h, int lane, int sl, double *dk_out,
                          int pickless) {
    const int H = 1 << log2h;
    double *ew = reinterpret_cast<double *>(area);           // [cap] weights in list order (padding 0.0)
    double *ssum = ew + cap;                                  // [cap] in-order weight sum of community number p
    int *tk = reinterpret_cast<int *>(ssum + cap);            // [H] community of a slot
    int *tp = tk + H;                                         // [H] first eligible position
    unsigned short *tpos = reinterpret_cast<unsigned short *>(tp + H); // [H] slot -> community number
    unsigned short *spos = tpos + H;                          // [cap] community number -> slot
    unsigned short *es = spos + cap;                          // [cap] edge -> slot, then edge -> community number (padding 0xFFFF)
    int *cnt = reinterpret_cast<int *>(es + cap);
    const int o0 = hd.o0, d_out = hd.d_out, i0 = hd.i0, d_in = hd.d_in;
    const int d = d_out + d_in;
    const int d4 = (d + 3) & ~3;
    const int old = hd.old;
    const int mypart = elig_part ? hd.mypart : 0;
    const double k_v = hd.k_v;
    *dk_out = 0.0;
    if (d == 0)
        return old;
    const double st_old = sum_tot[old];
    for (int j = 4 * sl; j < H; j += 4 * SG)
        *reinterpret_cast<int4 *>(tk + j) = make_int4(LEI_EMPTY, LEI_EMPTY, LEI_EMPTY, LEI_EMPTY);
    if (sl == 0)
        cnt[0] = 0;
    __builtin_amdgcn_wave_barrier();
    for (int e0 = 0; e0 < d4; e0 += 4 * SG) { // four edges per lane in flight: targets + weights, then labels (+ partitions)
        int t[4], c[4], part[4];
        double w[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int ecl = min(e0 + j * SG + sl, d - 1);
            const bool out = ecl < d_out;
            t[j] = *(out ? g.tgt_out + o0 + ecl : g.tgt_in + i0 + (ecl - d_out));
            const double *pw = out ? g.w_out : g.w_in;
            w[j] = pw ? pw[out ? o0 + ecl : i0 + (ecl - d_out)] : 1.0;
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
            c[j] = label[t[j]];
#pragma unroll
        for (int j = 0; j < 4; j++)
            part[j] = elig_part ? elig_part[t[j]] : mypart;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int e = e0 + j * SG + sl;
            if (e >= d4)
                continue;
            if (e >= d) {
                ew[e] = 0.0;
                es[e] = 0xFFFF;
                continue;
            }
            ew[e] = w[j];
            unsigned h = lei_hash(c[j], log2h);
            for (;;) {
                int prev = tk[h];
                if (prev == LEI_EMPTY) {
                    prev = atomicCAS(&tk[h], LEI_EMPTY, c[j]);
                    if (prev == LEI_EMPTY) { // (the inserter numbers the community; LDS operations of a wavefront run in order)
                        const int p = atomicAdd(&cnt[0], 1);
                        tpos[h] = (unsigned short)p;
                        spos[p] = (unsigned short)h;
                        tp[h] = 0x7fffffff;
                    }
                }
                if (prev == LEI_EMPTY || prev == c[j])
                    break;
                h = (h + 1) & (H - 1);
            }
            es[e] = (unsigned short)h;
            if (part[j] == mypart)
                atomicMin(&tp[h], e);
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    for (int e = sl; e < d; e += SG) // slot -> community number
        es[e] = tpos[es[e]];
    __builtin_amdgcn_wave_barrier();
    const int ncomm = cnt[0];
    const ushort4 *es4 = reinterpret_cast<const ushort4 *>(es);
    const double2 *ew2 = reinterpret_cast<const double2 *>(ew);
    for (int p0 = 0; p0 < ncomm; p0 += SG) { // lane p walks the list for community number p: list-order f64 sum (:75-90)
        const int p = p0 + sl;
        const unsigned short my = p < ncomm ? (unsigned short)p : (unsigned short)0xFFFE;
        double acc = 0.0;
        for (int q = 0; q < (d4 >> 2); q++) {
            const ushort4 e4 = es4[q];
            const double2 wa = ew2[2 * q], wb = ew2[2 * q + 1];
            if (e4.x == my) acc += wa.x;
            if (e4.y == my) acc += wa.y;
            if (e4.z == my) acc += wb.x;
            if (e4.w == my) acc += wb.y;
        }
        if (p < ncomm)
            ssum[p] = acc;
    }
    __builtin_amdgcn_wave_barrier();
    double k_v_to_old = 0.0; // weight_to_community(v, old), :163
    {
        unsigned h = lei_hash(old, log2h);
        for (int probe = 0; probe < H; probe++) {
            const int key = tk[h];
            if (key == old) {
                k_v_to_old = ssum[tpos[h]];
                break;
            }
            if (key == LEI_EMPTY)
                break;
            h = (h + 1) & (H - 1);
        }
    }
    LeiBest b = {-1.0, 0.0, 0x7fffffff, old};
    for (int p = sl; p < ncomm; p += SG) { // max gain, ties to the lowest first eligible position = the first-seen rule (:212)
        const int h = spos[p];
        const int c = tk[h], pos = tp[h];
        if (c == old || pos == 0x7fffffff || (pickless && c > old))
            continue;
        const double sacc = ssum[p];
        const double st_c = sum_tot[c];
        const double gain = (sacc - k_v_to_old) / m + resolution * k_v * (st_old - k_v - st_c) / (2.0 * m * m); // :209-210
        if (!(gain > 0.0))
            continue;
        if (gain > b.gain || (gain == b.gain && pos < b.pos)) {
            b.gain = gain;
            b.pos = pos;
            b.c = c;
            b.dk = sacc - k_v_to_old;
        }
    }
    lei_best_step<0>(b);
    lei_best_step<1>(b);
    lei_best_step<2>(b);
    lei_best_step<3>(b);
    if (SG > 16)
        lei_best_step<4>(b);
    if (SG > 32)
        lei_best_step<5>(b);
    *dk_out = b.gain > 0.0 ? b.dk : 0.0;
    return b.gain > 0.0 ? b.c : old;
}

struct LeiArgs {
    DevGraph g;
    int *label;
    double *sum_tot;
    const double *kdeg;
    double m, resolution;
    int use_both;
    const int *elig_part; // refinement: phase-1 partition; null for local moving
    int *scratch_c;       // global scratch for nodes with more than LEI_CAP edges: [blocks][max_deg]
    double *scratch_w;
    unsigned char *scratch_e;
    int max_deg;
    int *out; // [0] moves [1] sweeps
    int max_sweeps;
    // batched
    int b0, b1;
    int *dec, *cmin;
    unsigned char *win;
    unsigned char *mv;           // [round slot] 1 = the node wants to move (k_leiden_win scans these instead of dec + label)
    double *dk;
    unsigned long long *Jq, *Lq; // fixed-point (2^20) tallies of the movers' degrees per community
    int apply_on_device;         // 0: weighted graph → the host applies winners in node order
    int lds_cap;                 // k_leiden_eval_big: edges staged in LDS per node (multiple of 16, ≤ LEI_CAP)
    const int *biglist;          // nodes with more than LEI_SG_CAP edges, ascending
    const long long *bigoff;     // [biglist index] offset of the node's region in scratch_c/w/e (nodes with more than LEI_CAP edges)
    int big0, big1;              // the slice of biglist inside [b0, b1)
    int parity;                  // round parity: out[1 + parity] counts this round's safe winners
    int big_log2h;               // hash table size of a wide node (unweighted): 2^big_log2h >= 2 * lds_cap
    int sync;                    // 1: whole-graph synchronous sweep — every positive-gain mover applies, no tallies (k_leiden_apply_sync)
    int pickless;                // this sweep only allows moves to a community with a smaller id
    int sg_log2h;                // log2 of a sub-group's table size (unweighted): LEI_SG_LOG2H, or one less once labels have merged
};

#define LEI_FX 1048576.0
#define LEI_GROW 4       // tail rule of the batched schedule: round size factor ...
#define LEI_GROW_DIV 256 // ... once a sweep commits fewer than N / 256 moves
#define LEI_GROW_MAX 64  // (upper bound of the MN_LEIDEN_GROW tuning knob)
DEVI unsigned long long fx_up(double k) { return (unsigned long long)ceil(k * LEI_FX); }

DEVI int node_degree(const LeiArgs &a, int v) {
    return a.g.off_out[v + 1] - a.g.off_out[v] + (a.use_both ? a.g.off_in[v + 1] - a.g.off_in[v] : 0);
}

__global__ void __launch_bounds__(64) k_leiden_seq(LeiArgs a) {
    __shared__ __align__(16) double lds_w[LEI_CAP];
    __shared__ __align__(16) int lds_c[LEI_CAP];
    __shared__ __align__(16) unsigned char lds_e[LEI_CAP];
    const int lane = threadIdx.x;
    int total = 0, improved = 1, sweeps = 0;
    while (improved && sweeps < a.max_sweeps) { // :154-229
        improved = 0;
        sweeps++;
        for (int v = 0; v < a.g.n; v++) {
            const int old = ld_i<true>(a.label + v);
            int best; // two call sites: LDS staging keeps its address space (a runtime-selected pointer would be FLAT)
            if (node_degree(a, v) <= LEI_CAP)
                best = best_move<true>(a.g, v, a.label, a.sum_tot, a.kdeg, a.m, a.resolution, a.use_both, a.elig_part, lds_c,
                                       lds_w, lds_e, lane);
            else
                best = best_move<true>(a.g, v, a.label, a.sum_tot, a.kdeg, a.m, a.resolution, a.use_both, a.elig_part,
                                       a.scratch_c, a.scratch_w, a.scratch_e, lane);
            if (best != old) { // :220-227
                if (lane == 0) {
                    const double k_v = a.kdeg[v];
                    double so = ld_d<true>(a.sum_tot + old), sb = ld_d<true>(a.sum_tot + best);
                    __hip_atomic_store(a.sum_tot + old, so - k_v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(a.sum_tot + best, sb + k_v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(a.label + v, best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __builtin_amdgcn_s_waitcnt(0);
                improved = 1;
                total++;
            }
        }
    }
    if (lane == 0) {
        a.out[0] = total;
        a.out[1] = sweeps;
        a.out[2] = improved; // 1 → stopped by max_sweeps, not converged
    }
}

// a mover's tallies: smallest mover index per touched community, and the movers' degrees leaving / joining it
DEVI void lei_tally(const LeiArgs &a, int v, int old, int best, double dk) {
    a.dec[v - a.b0] = best;
    if (a.sync) // synchronous sweep: the decision is all k_leiden_apply_sync needs
        return;
    a.dk[v - a.b0] = dk;
    a.mv[v - a.b0] = best != old;
    if (best == old)
        return;
    atomicMin(a.cmin + old, v);
    atomicMin(a.cmin + best, v);
    const unsigned long long q = fx_up(a.kdeg[v]);
    atomicAdd(a.Lq + old, q);
    atomicAdd(a.Jq + best, q);
}

// One launch evaluates a whole round: blocks [0, nsmall) take 64/SG nodes each in SG-lane sub-groups (nodes with more
// than LEI_SG_CAP edges are skipped there), blocks [nsmall, nsmall + big1 - big0) take one such wide node each
// (biglist).  HASH = unweighted graph → best_move_hash; otherwise the list-order f64 sums of best_move(_sg).
// Dynamic LDS: max(sub-group area, wide-node area) — sized by the host (lei_eval_lds).
template <int SG, bool HASH>
__global__ void __launch_bounds__(64 * LEI_WPB) k_leiden_eval(LeiArgs a, int nsmall, unsigned wave_lds) {
    // LEI_WPB independent wavefronts per workgroup (no workgroup barrier anywhere): a quarter of the workgroups to dispatch
    extern __shared__ __align__(16) unsigned char lei_smem_all[];
    unsigned char *lei_smem = lei_smem_all + (threadIdx.x >> 6) * wave_lds;
    const int vblock = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); // the wavefront's index in the launch
    constexpr int NG = 64 / SG;
    const int lane = threadIdx.x & 63;
    if (vblock < nsmall) {
        const int grp = lane / SG, sl = lane % SG;
        const int v = a.b0 + vblock * NG + grp;
        if (v >= a.b1)
            return;
        double dk = 0.0;
        int best, old;
        if (HASH) {
            const LeiHead hd = lei_head(a.g, v, a.label, a.kdeg, a.use_both, a.elig_part);
            if (hd.d_out + hd.d_in > LEI_SG_CAP)
                return;
            old = hd.old;
            const int H = 1 << a.sg_log2h;
            int *tk = reinterpret_cast<int *>(lei_smem) + grp * LEI_SG_AREA_OF(a.sg_log2h);
            best = best_move_hash<SG>(a.g, hd, a.label, a.sum_tot, a.m, a.resolution, a.elig_part, tk, tk + H, tk + 2 * H, tk + 3 * H,
                                      a.sg_log2h, lane, sl, &dk, a.pickless);
        } else {
            const LeiHead hd = lei_head(a.g, v, a.label, a.kdeg, a.use_both, a.elig_part);
            if (hd.d_out + hd.d_in > LEI_SG_CAP)
                return;
            old = hd.old;
            unsigned char *area = lei_smem + (size_t)grp * lei_wslots_bytes(LEI_SG_CAP, LEI_SG_LOG2H - 1);
            best = best_move_wslots<SG>(a.g, hd, a.label, a.sum_tot, a.m, a.resolution, a.elig_part, area, LEI_SG_CAP, LEI_SG_LOG2H - 1,
                                        lane, sl, &dk, a.pickless);
        }
        if (sl == 0)
            lei_tally(a, v, old, best, dk);
        return;
    }
    const int bi = vblock - nsmall;
    if (bi >= a.big1 - a.big0)
        return;
    const int v = a.biglist[a.big0 + bi];
    const int deg = node_degree(a, v);
    double dk = 0.0;
    int best;
    if (HASH && deg <= a.lds_cap) {
        int *tk = reinterpret_cast<int *>(lei_smem);
        const int H = 1 << a.big_log2h;
        const LeiHead hd = lei_head(a.g, v, a.label, a.kdeg, a.use_both, a.elig_part);
        int lg = 8; // the table is sized to this node (≥ its degree: one entry per edge at most), not to the widest one
        while ((1 << lg) < deg)
            lg++;
        best = best_move_hash<64>(a.g, hd, a.label, a.sum_tot, a.m, a.resolution, a.elig_part, tk, tk + H, tk + 2 * H, tk + 3 * H,
                                  lg, lane, lane, &dk, a.pickless);
    } else if (deg <= a.lds_cap) {
        const LeiHead hd = lei_head(a.g, v, a.label, a.kdeg, a.use_both, a.elig_part);
        best = best_move_wslots<64>(a.g, hd, a.label, a.sum_tot, a.m, a.resolution, a.elig_part, lei_smem, a.lds_cap, a.big_log2h, lane,
                                    lane, &dk, a.pickless);
    } else { // more edges than fit in LDS: global scratch, list-order sums
        const size_t o = (size_t)a.bigoff[a.big0 + bi]; // this node's own region (only nodes past LEI_CAP have one)
        best = best_move<false>(a.g, v, a.label, a.sum_tot, a.kdeg, a.m, a.resolution, a.use_both, a.elig_part, a.scratch_c + o,
                                a.scratch_w + o, a.scratch_e + o, lane, &dk, a.pickless);
    }
    if (lane == 0)
        lei_tally(a, v, a.label[v], best, dk);
}

static size_t lei_wslots_bytes_h(int cap, int log2h) { return (size_t)20 * cap + (size_t)10 * ((size_t)1 << log2h) + 16; }
// LDS bytes of one k_leiden_eval workgroup
static size_t lei_eval_lds(int sg, bool hash, int lds_cap, int big_log2h, int sg_log2h) {
    const int ng = 64 / sg;
    const size_t small = hash ? (size_t)ng * LEI_SG_AREA_OF(sg_log2h) * sizeof(int) : (size_t)ng * lei_wslots_bytes_h(LEI_SG_CAP, LEI_SG_LOG2H - 1);
    // table (3 ints per entry) + occupied-entry list: a counter (16 B) and one 16-bit slot per edge
    const size_t big = hash ? (size_t)3 * ((size_t)1 << big_log2h) * sizeof(int) + 16 + (((size_t)lds_cap * 2 + 15) & ~(size_t)15)
                            : lei_wslots_bytes_h(lds_cap, big_log2h);
    return small > big ? small : big;
}

// movers whose smaller-index neighbours in the round do not move ("free"), gain re-checked in the worst order of
// application; 8 lanes share a node's adjacency scan
__global__ void __launch_bounds__(256) k_leiden_win(LeiArgs a) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int v = a.b0 + (tid >> 3), sl = tid & 7;
    if (v >= a.b1)
        return;
    // (a chain of dependent gathers: everything whose address is known is requested at once)
    const int old = a.label[v], best = a.dec[v - a.b0];
    const int xo0 = a.g.off_out[v], xo1 = a.g.off_out[v + 1];
    const int xi0 = a.use_both ? a.g.off_in[v] : 0, xi1 = a.use_both ? a.g.off_in[v + 1] : 0;
    const double k_v = a.kdeg[v], dkv = a.dk[v - a.b0];
    unsigned char win = 0;
    if (best != old) { // (uniform over the node's 8 lanes)
        const unsigned long long Lq = a.Lq[old], Jq = a.Jq[best];
        const double st_o = a.sum_tot[old], st_b = a.sum_tot[best];
        const int cm_o = a.cmin[old], cm_b = a.cmin[best];
        int blocked = 0;
        // four targets per lane in flight, then their four mover flags: every load is unconditional (clamped index), so the
        // compiler issues them back to back instead of one guarded load + wait per edge
        for (int pass = 0; pass < (a.use_both ? 2 : 1); pass++) {
            const int *tgt = pass ? a.g.tgt_in : a.g.tgt_out;
            const int x1 = pass ? xi1 : xo1;
            for (int x = (pass ? xi0 : xo0) + sl; x < x1; x += 32) {
                int w[4], f[4];
#pragma unroll
                for (int j = 0; j < 4; j++)
                    w[j] = tgt[min(x + 8 * j, x1 - 1)]; // (a repeated last edge changes nothing)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const bool inr = w[j] >= a.b0 && w[j] < v;
                    f[j] = a.mv[inr ? w[j] - a.b0 : 0] & (inr ? 1 : 0);
                }
                blocked |= f[0] | f[1] | f[2] | f[3];
            }
        }
        blocked |= __shfl_xor(blocked, 1);
        blocked |= __shfl_xor(blocked, 2);
        blocked |= __shfl_xor(blocked, 4);
        if (!blocked && sl == 0) {
            const unsigned long long q = fx_up(k_v);
            const double Lo = (double)(Lq - q) / LEI_FX, Jc = (double)(Jq - q) / LEI_FX;
            const double gain2 = dkv / a.m + a.resolution * k_v * (st_o - Lo - k_v - st_b - Jc) / (2.0 * a.m * a.m);
            const int strict = cm_o == v && cm_b == v;
            win = (unsigned char)((gain2 > 0.0 ? 1 : 0) | (strict ? 2 : 0));
        }
    }
    if (sl == 0)
        a.win[v - a.b0] = win;
    // "this round has a safe winner" is a flag, not a count: a plain store (thousands of atomics on one address cost a
    // mover-heavy round ≈ 10 ns each, even one per wavefront)
    if (win & 1)
        a.out[1 + a.parity] = 1;
}

// resets the round's tallies and applies its winners
__global__ void __launch_bounds__(256) k_leiden_apply(LeiArgs a) {
    __shared__ int blk_moves;
    const int v = a.b0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (threadIdx.x == 0)
        blk_moves = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0)
        a.out[1 + (a.parity ^ 1)] = 0; // the next round's flag (this round reads the other one)
    __syncthreads();
    int applied = 0;
    if (v < a.b1) {
        const int old = a.label[v], best = a.dec[v - a.b0];
        const unsigned char wbits = a.win[v - a.b0];
        const int safe = a.out[1 + a.parity];
        const double k_v = a.kdeg[v];
        if (best != old) {
            a.cmin[old] = 0x7fffffff;
            a.cmin[best] = 0x7fffffff;
            a.Lq[old] = 0;
            a.Jq[best] = 0;
            const int use_bit = safe > 0 ? 1 : 2;
            if (wbits & use_bit) {
                if (a.apply_on_device) {
                    // unweighted graph: degrees are integers, f64 atomic adds are exact → order-free
                    atomicAdd(a.sum_tot + old, -k_v);
                    atomicAdd(a.sum_tot + best, k_v);
                } // (weighted: sum_tot was brought up to date in the reference's addition order by k_leiden_apply_ops)
                a.label[v] = best;
                applied = 1;
            }
        }
    }
    // the sweep's move count: one atomic per workgroup
    const unsigned long long ba = __ballot(applied);
    if (ba && (threadIdx.x & 63) == __ffsll((long long)ba) - 1)
        atomicAdd(&blk_moves, __popcll(ba));
    __syncthreads();
    if (threadIdx.x == 0 && blk_moves)
        atomicAdd(a.out, blk_moves);
}

// Synchronous sweep: every mover applies.  Unweighted graphs: degrees are integers, the f64 atomic adds are exact and
// order-free.  Weighted graphs: sum_tot was brought up to date in node order by k_leiden_apply_ops, only labels move here.
__global__ void __launch_bounds__(256) k_leiden_apply_sync(LeiArgs a) {
    __shared__ int blk_moves;
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (threadIdx.x == 0)
        blk_moves = 0;
    __syncthreads();
    int applied = 0;
    if (v < a.b1) {
        const int old = a.label[v], best = a.dec[v];
        if (best != old) {
            if (a.apply_on_device) {
                const double k_v = a.kdeg[v];
                atomicAdd(a.sum_tot + old, -k_v);
                atomicAdd(a.sum_tot + best, k_v);
            }
            a.label[v] = best;
            applied = 1;
        }
    }
    const unsigned long long ba = __ballot(applied);
    if (ba && (threadIdx.x & 63) == __ffsll((long long)ba) - 1)
        atomicAdd(&blk_moves, __popcll(ba));
    __syncthreads();
    if (threadIdx.x == 0 && blk_moves)
        atomicAdd(a.out, blk_moves);
}

// Weighted graphs: a round's winners applied on the device in the reference's order.  f64 addition is not associative, so
// sum_tot[c] must receive its additions exactly as the sequential loop makes them: winners in node order, each first
// "sum_tot[old] -= k" then "sum_tot[new] += k" (:220-223).  Operation 2·slot is the subtraction, 2·slot + 1 the addition;
// k_leiden_ops keys them by community (non-winners: key = n), a STABLE radix sort groups a community's operations without
// reordering them, and k_leiden_apply_ops lets the lane that holds a community's first operation replay its run.
__global__ void __launch_bounds__(256) k_leiden_ops(LeiArgs a, int n_nodes, int *keys) {
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= a.b1 - a.b0)
        return;
    const int v = a.b0 + slot;
    const int old = a.label[v], best = a.dec[slot]; // (labels are written by k_leiden_apply, after this)
    bool winner = best != old; // synchronous sweep: every mover
    if (!a.sync) {
        const int use_bit = a.out[1 + a.parity] > 0 ? 1 : 2;
        winner = winner && (a.win[slot] & use_bit);
    }
    keys[2 * slot] = winner ? old : n_nodes;
    keys[2 * slot + 1] = winner ? best : n_nodes;
}
__global__ void k_leiden_apply_ops(LeiArgs a, int n_nodes, const int *keys_sorted, const int *ops_sorted, int n_ops) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_ops)
        return;
    const int c = keys_sorted[j];
    if (c >= n_nodes || (j > 0 && keys_sorted[j - 1] == c))
        return;
    double S = a.sum_tot[c];
    for (int i = j; i < n_ops && keys_sorted[i] == c; i++) {
        const int op = ops_sorted[i];
        const double kv = a.kdeg[a.b0 + (op >> 1)];
        S = (op & 1) ? S + kv : S - kv;
    }
    a.sum_tot[c] = S;
}

// weighted_degree (:95-104) and weight_to_community(v, community[v]) (:75-90), list order, f64
__global__ void k_wdeg(DevGraph g, int use_both, double *kdeg) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= g.n)
        return;
    double k = 0.0;
    for (int e = g.off_out[v]; e < g.off_out[v + 1]; e++)
        k += g.w_out ? g.w_out[e] : 1.0;
    if (use_both)
        for (int e = g.off_in[v]; e < g.off_in[v + 1]; e++)
            k += g.w_in ? g.w_in[e] : 1.0;
    kdeg[v] = k;
}

__global__ void k_w2c_self(DevGraph g, int use_both, const int *label, double *out, int v0, int v1) {
    const int v = v0 + blockIdx.x * blockDim.x + threadIdx.x; // nodes [v0, v1): one rank's share of the modularity's per-node terms
    if (v >= v1)
        return;
    const int c = label[v];
    double s = 0.0;
    for (int e = g.off_out[v]; e < g.off_out[v + 1]; e++)
        if (label[g.tgt_out[e]] == c)
            s += g.w_out ? g.w_out[e] : 1.0;
    if (use_both)
        for (int e = g.off_in[v]; e < g.off_in[v + 1]; e++)
            if (label[g.tgt_in[e]] == c)
                s += g.w_in ? g.w_in[e] : 1.0;
    out[v] = s;
}

// ───────────────────────── host ─────────────────────────

static thread_local std::string g_gerr;
static void gset_err(const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_gerr = buf;
}
extern "C" const char *mn_graph_last_error(void) { return g_gerr.c_str(); }

#define GCHK(expr)                                                                                 \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess) {                                                                   \
            gset_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return -1;                                                                             \
        }                                                                                          \
    } while (0)

struct mn_graph {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int n = 0;
    long long e_out = 0, e_in = 0;
    int max_deg_both = 0, max_deg_out = 0;
    bool weighted = false;
    int *off_out = nullptr, *tgt_out = nullptr, *off_in = nullptr, *tgt_in = nullptr;
    double *w_out = nullptr, *w_in = nullptr;
    double last_ms = 0;
    mn_leiden_stats stats = {};
    std::vector<int> h_off_out, h_off_in; // host copies of the offsets (degrees: list of wide nodes, scratch sizing)
    struct LeiWork *work = nullptr;       // run_leiden's device buffers, allocated on first use and kept
};

template <typename T> static int up(T **dst, const T *src, size_t n) {
    *dst = nullptr;
    GCHK(hipMalloc(dst, (n ? n : 1) * sizeof(T)));
    if (n)
        GCHK(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

extern "C" mn_graph *mn_graph_create(int n_nodes, const int *off_out, const int *tgt_out, const double *w_out,
                                     const int *off_in, const int *tgt_in, const double *w_in, int device) try {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        gset_err("mn_graph_create: HIP device %d not available (no CPU fallback)", device);
        return nullptr;
    }
    if (n_nodes < 0 || !off_out || !off_in) {
        gset_err("mn_graph_create: bad arguments");
        return nullptr;
    }
    mn_graph *g = new mn_graph();
    g->device = device;
    g->n = n_nodes;
    g->e_out = n_nodes ? off_out[n_nodes] : 0;
    g->e_in = n_nodes ? off_in[n_nodes] : 0;
    g->weighted = w_out != nullptr || w_in != nullptr;
    for (int v = 0; v < n_nodes; v++) {
        int dO = off_out[v + 1] - off_out[v], dI = off_in[v + 1] - off_in[v];
        if (dO > g->max_deg_out) g->max_deg_out = dO;
        if (dO + dI > g->max_deg_both) g->max_deg_both = dO + dI;
    }
    bool ok = hipSetDevice(device) == hipSuccess && hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreate(&g->ev0) == hipSuccess && hipEventCreate(&g->ev1) == hipSuccess;
    ok = ok && up(&g->off_out, off_out, (size_t)n_nodes + 1) == 0 && up(&g->tgt_out, tgt_out, (size_t)g->e_out) == 0 &&
         up(&g->off_in, off_in, (size_t)n_nodes + 1) == 0 && up(&g->tgt_in, tgt_in, (size_t)g->e_in) == 0;
    if (ok && w_out)
        ok = up(&g->w_out, w_out, (size_t)g->e_out) == 0;
    if (ok && w_in)
        ok = up(&g->w_in, w_in, (size_t)g->e_in) == 0;
    if (!ok) {
        mn_graph_destroy(g);
        return nullptr;
    }
    g->h_off_out.assign(off_out, off_out + n_nodes + 1);
    g->h_off_in.assign(off_in, off_in + n_nodes + 1);
    return g;
} MN_GUARD_END(gset_err, MN_NOTHING, nullptr)

// One direction of a blocked CSR (the rows of "{t}_csr_fwd" / "{t}_csr_rev") straight into device buffers: per
// block, node count = offsets_bytes/4 - 1 and edge count = targets_bytes/4 (csr_deserialize, src/graph_csr.c:122-163);
// offsets are rebased by the running edge count, targets are already global (csr_merge_blocks, :402-477).
static int upload_blocks(const mn_csr_block *blk, int nb, int n_nodes, std::vector<int> &off, int **d_tgt, double **d_w,
                         long long *n_edges, bool *weighted) {
    off.assign((size_t)n_nodes + 1, 0);
    long long edges = 0;
    for (int b = 0; b < nb; b++) {
        if (!blk[b].offsets || blk[b].offsets_bytes < 4 || blk[b].targets_bytes < 0 || blk[b].targets_bytes % 4) {
            gset_err("mn_graph_create_blocked: block %d is malformed", b);
            return -1;
        }
        edges += blk[b].targets_bytes / 4;
    }
    if (edges > 0x7fffffffLL) {
        gset_err("mn_graph_create_blocked: %lld edges exceed int32 offsets", edges);
        return -1;
    }
    *weighted = nb > 0 && blk[0].weights != nullptr && blk[0].weights_bytes > 0; // has_weights of the first block (:415)
    *d_tgt = nullptr;
    *d_w = nullptr;
    GCHK(hipMalloc(d_tgt, (size_t)std::max<long long>(1, edges) * sizeof(int)));
    if (*weighted)
        GCHK(hipMalloc(d_w, (size_t)std::max<long long>(1, edges) * sizeof(double)));
    long long eoff = 0;
    int noff = 0;
    for (int b = 0; b < nb; b++) {
        const int bn = blk[b].offsets_bytes / 4 - 1, be = blk[b].targets_bytes / 4;
        const int *bo = static_cast<const int *>(blk[b].offsets);
        for (int i = 0; i < bn && noff + i < n_nodes; i++) {
            if (bo[i] < 0 || bo[i] > be) {
                gset_err("mn_graph_create_blocked: block %d offsets out of range", b);
                return -1;
            }
            off[(size_t)noff + i] = (int)(bo[i] + eoff);
        }
        if (be > 0 && blk[b].targets)
            GCHK(hipMemcpy(*d_tgt + eoff, blk[b].targets, (size_t)be * sizeof(int), hipMemcpyHostToDevice));
        if (*weighted && be > 0) {
            if (!blk[b].weights || blk[b].weights_bytes != be * 8) {
                gset_err("mn_graph_create_blocked: block %d weights do not match its targets", b);
                return -1;
            }
            GCHK(hipMemcpy(*d_w + eoff, blk[b].weights, (size_t)be * sizeof(double), hipMemcpyHostToDevice));
        }
        eoff += be;
        noff += bn;
    }
    for (int i = std::min(noff, n_nodes); i <= n_nodes; i++) // sentinel, and nodes the blocks did not cover (:470-475)
        off[(size_t)i] = (int)eoff;
    *n_edges = eoff;
    return 0;
}

extern "C" mn_graph *mn_graph_create_blocked(int n_nodes, const mn_csr_block *fwd, int n_fwd, const mn_csr_block *rev, int n_rev,
                                             int device) try {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        gset_err("mn_graph_create_blocked: HIP device %d not available (no CPU fallback)", device);
        return nullptr;
    }
    if (n_nodes < 0 || n_fwd < 0 || n_rev < 0 || (n_fwd && !fwd) || (n_rev && !rev)) {
        gset_err("mn_graph_create_blocked: bad arguments");
        return nullptr;
    }
    mn_graph *g = new mn_graph();
    g->device = device;
    g->n = n_nodes;
    std::vector<int> oo, oi;
    bool wo = false, wi = false;
    bool ok = hipSetDevice(device) == hipSuccess && hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreate(&g->ev0) == hipSuccess && hipEventCreate(&g->ev1) == hipSuccess;
    ok = ok && upload_blocks(fwd, n_fwd, n_nodes, oo, &g->tgt_out, &g->w_out, &g->e_out, &wo) == 0 &&
         upload_blocks(rev, n_rev, n_nodes, oi, &g->tgt_in, &g->w_in, &g->e_in, &wi) == 0;
    if (ok) {
        // out-of-range targets would fault in the kernels: the rows come from a file, so look before launching
        std::vector<int> chk;
        for (int dir = 0; dir < 2 && ok; dir++) {
            const long long ne = dir ? g->e_in : g->e_out;
            chk.resize((size_t)ne);
            if (ne)
                ok = hipMemcpy(chk.data(), dir ? g->tgt_in : g->tgt_out, (size_t)ne * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
            for (long long e = 0; e < ne && ok; e++)
                if (chk[(size_t)e] < 0 || chk[(size_t)e] >= n_nodes) {
                    gset_err("mn_graph_create_blocked: target %d out of range", chk[(size_t)e]);
                    ok = false;
                }
        }
    }
    if (ok) {
        g->weighted = wo || wi;
        for (int v = 0; v < n_nodes; v++) {
            int dO = oo[v + 1] - oo[v], dI = oi[v + 1] - oi[v];
            if (dO < 0 || dI < 0) {
                gset_err("mn_graph_create_blocked: offsets are not monotone at node %d", v);
                ok = false;
                break;
            }
            if (dO > g->max_deg_out) g->max_deg_out = dO;
            if (dO + dI > g->max_deg_both) g->max_deg_both = dO + dI;
        }
    }
    ok = ok && up(&g->off_out, oo.data(), (size_t)n_nodes + 1) == 0 && up(&g->off_in, oi.data(), (size_t)n_nodes + 1) == 0;
    if (!ok) {
        mn_graph_destroy(g);
        return nullptr;
    }
    g->h_off_out.swap(oo);
    g->h_off_in.swap(oi);
    return g;
} MN_GUARD_END(gset_err, MN_NOTHING, nullptr)

// ───────────────────────── run_leiden workspace (one per graph, reused) ─────────────────────────

struct LeiWork {
    int n = 0, batch_cap = 0, big_mode = -1;
    int *label = nullptr, *refined = nullptr, *out = nullptr, *dec = nullptr, *cmin = nullptr, *sc = nullptr,
        *first = nullptr, *flag = nullptr, *rank = nullptr, *biglist = nullptr, *counts = nullptr;
    unsigned char *win = nullptr, *mv = nullptr, *se = nullptr;
    double *sum_tot = nullptr, *kdeg = nullptr, *tmp = nullptr, *sw = nullptr, *dk = nullptr, *scal = nullptr,
           *s_in = nullptr;
    unsigned long long *Jq = nullptr, *Lq = nullptr;
    long long *bigoff = nullptr;
    // weighted graphs: the round's operations keyed by community, their stable sort (k_leiden_ops / k_leiden_apply_ops)
    int *okeys = nullptr, *okeys_s = nullptr, *oiota = nullptr, *oops_s = nullptr;
    void *osort_tmp = nullptr;
    size_t osort_bytes = 0;
    int osort_bits = 0, ocap = 0;
    size_t scratch_need = 0, scratch_have = 0; // entries: one region per node with more than LEI_CAP edges
    void *scan_tmp = nullptr;
    size_t scan_bytes = 0;
    int *h_out = nullptr;          // pinned: per-sweep move counts read back without stalling the launch queue
    hipEvent_t ev_rd[2] = {nullptr, nullptr};
    std::vector<int> h_big; // nodes with more than LEI_SG_CAP edges (for big_mode = use_both)
    void release() {
        void *ps[] = {label, refined, out, dec, cmin, sc, first, flag, rank, biglist, counts, win, mv, se, sum_tot, kdeg,
                      tmp, sw, dk, scal, s_in, Jq, Lq, scan_tmp, bigoff, okeys, okeys_s, oiota, oops_s, osort_tmp};
        for (void *q : ps)
            (void)hipFree(q);
        if (h_out)
            (void)hipHostFree(h_out);
        for (hipEvent_t e : ev_rd)
            if (e)
                (void)hipEventDestroy(e);
    }
};

extern "C" void mn_graph_destroy(mn_graph *g) {
    if (!g)
        return;
    (void)hipSetDevice(g->device);
    if (g->stream)
        (void)hipStreamSynchronize(g->stream);
    (void)hipFree(g->off_out); (void)hipFree(g->tgt_out); (void)hipFree(g->off_in); (void)hipFree(g->tgt_in);
    (void)hipFree(g->w_out); (void)hipFree(g->w_in);
    if (g->work) {
        g->work->release();
        delete g->work;
    }
    if (g->ev0) (void)hipEventDestroy(g->ev0);
    if (g->ev1) (void)hipEventDestroy(g->ev1);
    if (g->stream) (void)hipStreamDestroy(g->stream);
    delete g;
}

static int renumber(std::vector<int> &c) { // :317-331
    const int N = (int)c.size();
    std::vector<int> map((size_t)N, -1);
    int next = 0;
    for (int i = 0; i < N; i++) {
        if (map[c[i]] == -1)
            map[c[i]] = next++;
        c[i] = map[c[i]];
    }
    return next;
}

static int distinct(const std::vector<int> &c) {
    std::vector<unsigned char> seen(c.size(), 0);
    int n = 0;
    for (int x : c)
        if (!seen[x]) {
            seen[x] = 1;
            n++;
        }
    return n;
}

template <typename T> static int wmalloc(T **p, size_t n) {
    if (*p)
        (void)hipFree(*p);
    *p = nullptr;
    GCHK(hipMalloc(p, (n ? n : 1) * sizeof(T)));
    return 0;
}

__global__ void k_iota(int *p, int n);

// (re)size the workspace for this call; everything is kept for the next one
static int lei_prepare(mn_graph *g, int mode, int batch, int use_both, int max_deg) {
    if (!g->work)
        g->work = new LeiWork();
    LeiWork &w = *g->work;
    if (!w.h_out) {
        GCHK(hipHostMalloc(&w.h_out, 8 * sizeof(int)));
        GCHK(hipEventCreateWithFlags(&w.ev_rd[0], hipEventDisableTiming));
        GCHK(hipEventCreateWithFlags(&w.ev_rd[1], hipEventDisableTiming));
    }
    const int N = g->n;
    hipStream_t st = g->stream;
    if (w.n != N) {
        if (wmalloc(&w.label, (size_t)N) || wmalloc(&w.refined, (size_t)N) || wmalloc(&w.sum_tot, (size_t)N) ||
            wmalloc(&w.kdeg, (size_t)N) || wmalloc(&w.tmp, (size_t)N + 64) || wmalloc(&w.out, 16) || wmalloc(&w.cmin, (size_t)N) ||
            wmalloc(&w.Jq, (size_t)N) || wmalloc(&w.Lq, (size_t)N) || wmalloc(&w.first, (size_t)N) || wmalloc(&w.flag, (size_t)N) ||
            wmalloc(&w.rank, (size_t)N) || wmalloc(&w.counts, 8) || wmalloc(&w.scal, 8) || wmalloc(&w.s_in, (size_t)N))
            return -1;
        size_t bytes = 0;
        if (rocprim::exclusive_scan(nullptr, bytes, w.flag, w.rank, 0, (size_t)N, rocprim::plus<int>(), st) != hipSuccess) {
            gset_err("rocprim::exclusive_scan (size query) failed");
            return -1;
        }
        if (w.scan_tmp)
            (void)hipFree(w.scan_tmp);
        w.scan_tmp = nullptr;
        GCHK(hipMalloc(&w.scan_tmp, bytes ? bytes : 16));
        w.scan_bytes = bytes;
        w.n = N;
        w.big_mode = -1;
    }
    // every round leaves the tallies clean (k_leiden_apply); a call that failed half-way may not have
    GCHK(hipMemsetAsync(w.cmin, 0x7f, (size_t)N * sizeof(int), st)); // 0x7f7f7f7f > any node index
    GCHK(hipMemsetAsync(w.Jq, 0, (size_t)N * sizeof(unsigned long long), st));
    GCHK(hipMemsetAsync(w.Lq, 0, (size_t)N * sizeof(unsigned long long), st));
    if (w.big_mode != use_both) { // nodes the sub-group kernel leaves to the one-wavefront-per-node kernel
        w.h_big.clear();
        std::vector<long long> h_off;
        size_t need = 0; // scratch entries: Σ degree (int4-aligned) over the nodes that do not fit in LDS — at most 2E + 4N
        for (int v = 0; v < N; v++) {
            const int d = g->h_off_out[v + 1] - g->h_off_out[v] + (use_both ? g->h_off_in[v + 1] - g->h_off_in[v] : 0);
            if (d > LEI_SG_CAP) {
                w.h_big.push_back(v);
                h_off.push_back((long long)need);
                if (d > LEI_CAP)
                    need += ((size_t)d + 7) & ~(size_t)3;
            }
        }
        w.scratch_need = need;
        if (wmalloc(&w.biglist, w.h_big.size()) || wmalloc(&w.bigoff, w.h_big.size()))
            return -1;
        if (!w.h_big.empty()) {
            GCHK(hipMemcpyAsync(w.biglist, w.h_big.data(), w.h_big.size() * sizeof(int), hipMemcpyHostToDevice, st));
            GCHK(hipMemcpyAsync(w.bigoff, h_off.data(), h_off.size() * sizeof(long long), hipMemcpyHostToDevice, st));
            GCHK(hipStreamSynchronize(st)); // (h_off is a local)
        }
        w.big_mode = use_both;
    }
    if (mode == MN_LEIDEN_BATCHED && batch > w.batch_cap) {
        if (wmalloc(&w.dec, (size_t)batch + 64) || wmalloc(&w.dk, (size_t)batch) || wmalloc(&w.win, (size_t)batch) || wmalloc(&w.mv, (size_t)batch))
            return -1;
        w.batch_cap = batch;
    }
    if (mode == MN_LEIDEN_BATCHED && g->weighted && (batch > w.ocap || !w.okeys)) {
        const size_t n_ops = (size_t)2 * batch;
        if (wmalloc(&w.okeys, n_ops) || wmalloc(&w.okeys_s, n_ops) || wmalloc(&w.oiota, n_ops) || wmalloc(&w.oops_s, n_ops))
            return -1;
        hipLaunchKernelGGL(k_iota, dim3((unsigned)((n_ops + 255) / 256)), dim3(256), 0, st, w.oiota, (int)n_ops);
        w.osort_bits = 1;
        while (w.osort_bits < 31 && (1ll << w.osort_bits) <= (long long)N) // keys 0..N (N = "not a winner")
            w.osort_bits++;
        size_t bytes = 0;
        if (rocprim::radix_sort_pairs(nullptr, bytes, w.okeys, w.okeys_s, w.oiota, w.oops_s, n_ops, 0, w.osort_bits, st) != hipSuccess) {
            gset_err("rocprim::radix_sort_pairs (size query) failed");
            return -1;
        }
        if (w.osort_tmp)
            (void)hipFree(w.osort_tmp);
        w.osort_tmp = nullptr;
        GCHK(hipMalloc(&w.osort_tmp, bytes ? bytes : 16));
        w.osort_bytes = bytes;
        w.ocap = batch;
    }
    // global scratch for nodes whose edges do not fit in LDS: the sequential kernel reuses one region of max_deg entries,
    // the batched rounds give every such node its own (bigoff) — sized by those nodes' degrees, not by the round
    const size_t want = std::max<size_t>(w.scratch_need, (size_t)max_deg);
    if (max_deg > LEI_CAP && want > w.scratch_have) {
        if (wmalloc(&w.sc, want) || wmalloc(&w.sw, want) || wmalloc(&w.se, want))
            return -1;
        w.scratch_have = want;
    }
    return 0;
}

// ───────────────────────── device bookkeeping (unweighted graphs: all sums are exact integers) ─────────────────────────

__global__ void k_iota(int *p, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        p[i] = i;
}
__global__ void k_sum_d(const double *x, int n, double *out) { // integer-valued terms: exact in any order
    __shared__ double sh[256];
    double acc = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        acc += x[i];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if ((int)threadIdx.x < s2)
            sh[threadIdx.x] += sh[threadIdx.x + s2];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        atomicAdd(out, sh[0]);
}
__global__ void k_first_seen(const int *label, int n, int *first) { // first[c] = smallest i with label[i] == c
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        atomicMin(first + label[i], i);
}
__global__ void k_first_flag(const int *label, const int *first, int n, int *flag, int *count) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int f = 0;
    if (i < n) {
        f = first[label[i]] == i;
        flag[i] = f;
    }
    const unsigned long long b = __ballot(f);
    if (count && (threadIdx.x & 63) == 0 && b)
        atomicAdd(count, __popcll(b));
}
// renumber_communities (:317-331): new id = number of distinct labels first seen before this one's first member
__global__ void k_relabel(int *label, const int *first, const int *rank, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        label[i] = rank[first[label[i]]];
}
__global__ void k_scatter_add(const int *label, const double *val, int n, double *acc) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        atomicAdd(acc + label[i], val[i]);
}

// distinct labels of `label` → counts[slot] (device), leaves first[] / flag[] describing `label`
static int dev_distinct(mn_graph *g, const int *label, int slot) {
    LeiWork &w = *g->work;
    const int N = g->n, nb = (N + 255) / 256;
    hipStream_t st = g->stream;
    GCHK(hipMemsetAsync(w.first, 0x7f, (size_t)N * sizeof(int), st));
    GCHK(hipMemsetAsync(w.counts + slot, 0, sizeof(int), st));
    hipLaunchKernelGGL(k_first_seen, dim3(nb), dim3(256), 0, st, label, N, w.first);
    hipLaunchKernelGGL(k_first_flag, dim3(nb), dim3(256), 0, st, label, w.first, N, w.flag, w.counts + slot);
    return 0;
}
// renumber `label` in place in first-seen order (first[] / flag[] must describe it: dev_distinct)
static int dev_renumber(mn_graph *g, int *label) {
    LeiWork &w = *g->work;
    const int N = g->n, nb = (N + 255) / 256;
    hipStream_t st = g->stream;
    size_t bytes = w.scan_bytes;
    if (rocprim::exclusive_scan(w.scan_tmp, bytes, w.flag, w.rank, 0, (size_t)N, rocprim::plus<int>(), st) != hipSuccess) {
        gset_err("rocprim::exclusive_scan failed");
        return -1;
    }
    hipLaunchKernelGGL(k_relabel, dim3(nb), dim3(256), 0, st, label, w.first, w.rank, N);
    return 0;
}

// the tail rule's constants; MN_LEIDEN_GROW="factor,divisor" is a tuning knob (the oracle reads ORC_LEI_GROW the same way:
// other values give another — equally valid — schedule, so parity holds only when both sides are set alike)
static void lei_grow_setting(int *grow, int *grow_div) {
    *grow = LEI_GROW;
    *grow_div = LEI_GROW_DIV;
    if (const char *e = getenv("MN_LEIDEN_GROW"))
        sscanf(e, "%d,%d", grow, grow_div);
    *grow = std::min(std::max(*grow, 1), LEI_GROW_MAX);
    *grow_div = std::max(*grow_div, 1);
}

// evaluation of the nodes [a.b0, a.b1) (+ the wide nodes a.big0..a.big1 of that range) against the frozen state
static void lei_launch_eval(const LeiArgs &a, int nb, int sg, bool hashed, hipStream_t st) {
    const int nsmall = (nb + (64 / sg) - 1) / (64 / sg);
    const unsigned wlds = (unsigned)((lei_eval_lds(sg, hashed, a.lds_cap, a.big_log2h, a.sg_log2h) + 15) & ~(size_t)15);
    int wpb = LEI_WPB; // (wide nodes with large tables: fewer wavefronts per workgroup, 64 KB of dynamic LDS at most)
    while (wpb > 1 && (size_t)wlds * wpb > 60 * 1024)
        wpb >>= 1;
    const dim3 grid((unsigned)((nsmall + a.big1 - a.big0 + wpb - 1) / wpb)), blk(64 * wpb);
    const size_t lds = (size_t)wlds * wpb;
    if (sg == 32 && hashed)
        hipLaunchKernelGGL((k_leiden_eval<32, true>), grid, blk, lds, st, a, nsmall, wlds);
    else if (sg == 32)
        hipLaunchKernelGGL((k_leiden_eval<32, false>), grid, blk, lds, st, a, nsmall, wlds);
    else if (hashed)
        hipLaunchKernelGGL((k_leiden_eval<16, true>), grid, blk, lds, st, a, nsmall, wlds);
    else
        hipLaunchKernelGGL((k_leiden_eval<16, false>), grid, blk, lds, st, a, nsmall, wlds);
}
static int lei_sub_group(const mn_graph *g, int use_both) {
    int sg = (double)(use_both ? g->e_out + g->e_in : g->e_out) / std::max(1, g->n) > 48.0 ? 32 : 16;
    if (const char *e = getenv("MN_LEIDEN_SG")) // tuning knob: 16 or 32 lanes per node
        sg = atoi(e) == 16 ? 16 : 32;
    return sg;
}

// one phase (local moving when elig_part == nullptr, refinement otherwise); returns moves, -1 on error
static long long run_phase(mn_graph *g, LeiArgs a, int mode, int batch, int64_t *sweeps_out) {
    hipStream_t st = g->stream;
    int out[3] = {0, 0, 0};
    if (mode == MN_LEIDEN_SEQUENTIAL) {
        GCHK(hipMemsetAsync(a.out, 0, 3 * sizeof(int), st));
        hipLaunchKernelGGL(k_leiden_seq, dim3(1), dim3(64), 0, st, a);
        GCHK(hipGetLastError());
        GCHK(hipMemcpyAsync(out, a.out, sizeof(out), hipMemcpyDeviceToHost, st));
        GCHK(hipStreamSynchronize(st));
        *sweeps_out += out[1];
        if (out[2]) {
            gset_err("mn_graph_leiden: sweeps did not converge within %d (asymmetric adjacency?)", a.max_sweeps);
            return -1;
        }
        return out[0];
    }
    const std::vector<int> &big = g->work->h_big;
    long long total = 0;
    int improved = 1, sweeps = 0, parity = 0;
    const bool hashed = !g->weighted; // every weight 1.0 → counts (best_move_hash)
    const int sg = lei_sub_group(g, a.use_both);
    int *const out_base = a.out; // two counter blocks of 8 ints ([0] moves, [1..2] "has a safe winner" by round parity): sweep s uses block s & 1
    int pending = -1;            // sweep whose move count is still on its way to the host (device-applied moves only)
    // Tail rule (part of the schedule, restated in oracle/mn_graph_oracle.c batched_phase): once the sweep before the
    // previous one committed fewer than N / LEI_GROW_DIV moves, rounds are LEI_GROW times larger — few movers, few
    // conflicts, and a round's cost is mostly its three launches.  "Before the previous one" because the previous sweep's
    // count is still on its way to the host when this sweep is queued.
    int grow, grow_div;
    lei_grow_setting(&grow, &grow_div);
    const int batch0 = batch;
    long long moves_prev2 = -1; // sweep s-2 (as far as the host has seen it)
    while (improved && sweeps < a.max_sweeps) {
        improved = 0;
        sweeps++;
        batch = batch0;
        if (moves_prev2 >= 0 && moves_prev2 < g->n / grow_div)
            batch = (int)std::min<long long>((long long)batch0 * grow, std::max(batch0, g->n));
        const int blk = sweeps & 1;
        a.out = out_base + 8 * blk;
        GCHK(hipMemsetAsync(a.out, 0, 8 * sizeof(int), st));
        parity = 0;
        size_t bigpos = 0;
        for (int b = 0; b < g->n; b += batch) {
            a.b0 = b;
            a.b1 = b + batch < g->n ? b + batch : g->n;
            a.parity = parity;
            const int nb = a.b1 - a.b0;
            a.big0 = (int)bigpos;
            while (bigpos < big.size() && big[bigpos] < a.b1)
                bigpos++;
            a.big1 = (int)bigpos;
            lei_launch_eval(a, nb, sg, hashed, st);
            hipLaunchKernelGGL(k_leiden_win, dim3((nb * 8 + 255) / 256), dim3(256), 0, st, a);
            if (!a.apply_on_device) {
                // weighted graph: several winners may share a community and f64 addition is not associative → the round's
                // operations are grouped by community with a stable sort and every community replays its own in node order
                LeiWork &w = *g->work;
                const int n_ops = 2 * nb;
                hipLaunchKernelGGL(k_leiden_ops, dim3((nb + 255) / 256), dim3(256), 0, st, a, g->n, w.okeys);
                size_t bytes = w.osort_bytes;
                if (rocprim::radix_sort_pairs(w.osort_tmp, bytes, w.okeys, w.okeys_s, w.oiota, w.oops_s, (size_t)n_ops, 0, w.osort_bits,
                                              st) != hipSuccess) {
                    gset_err("rocprim::radix_sort_pairs failed");
                    return -1;
                }
                hipLaunchKernelGGL(k_leiden_apply_ops, dim3((n_ops + 255) / 256), dim3(256), 0, st, a, g->n, w.okeys_s, w.oops_s, n_ops);
            }
            hipLaunchKernelGGL(k_leiden_apply, dim3((nb + 255) / 256), dim3(256), 0, st, a); // resets tallies (+ applies)
            parity ^= 1;
        }
        GCHK(hipGetLastError());
        // Moves applied on the device: the count of this sweep travels to pinned memory behind the sweep, and the NEXT sweep
        // is queued before the host looks at the PREVIOUS one — the launch queue never drains.  If that previous sweep
        // moved nothing the state is a fixed point: the sweep just queued moves nothing either and is not counted.
        int *h = g->work->h_out + 4 * blk;
        GCHK(hipMemcpyAsync(h, a.out, sizeof(int), hipMemcpyDeviceToHost, st));
        GCHK(hipEventRecord(g->work->ev_rd[blk], st));
        improved = 1;
        if (pending >= 0) {
            const int pb = pending & 1;
            GCHK(hipEventSynchronize(g->work->ev_rd[pb]));
            const int mv = g->work->h_out[4 * pb];
            if (mv == 0) { // `pending` was the last real sweep; the one queued above is redundant
                sweeps = pending;
                improved = 0;
                pending = -1;
                break;
            }
            total += mv;
            moves_prev2 = mv; // `pending` is the sweep before the one just queued
        }
        pending = sweeps;
    }
    if (pending >= 0) { // stopped by max_sweeps
        GCHK(hipEventSynchronize(g->work->ev_rd[pending & 1]));
        total += g->work->h_out[4 * (pending & 1)];
    }
    a.out = out_base;
    GCHK(hipStreamSynchronize(st));
    *sweeps_out += sweeps;
    return total;
}

// Default schedule of MN_LEIDEN_BATCHED since round 4: WHOLE-GRAPH synchronous sweeps (oracle/mn_graph_oracle.c sync_phase is
// the restatement, bit for bit).  A sweep = k_leiden_eval over every node against the state frozen at its start +
// k_leiden_apply_sync applying EVERY positive-gain mover (weighted graphs: sum_tot in node order through the stable sort of
// k_leiden_ops).  Simultaneous moves can swap two nodes for ever, so every `period`-th sweep is "pick-less" (Naim et al., GPU
// Louvain): a node may only move to a community with a smaller id.  The phase ends with the first ordinary sweep that moves
// nothing (= the sequential loop's fixed point); Q is not monotone under simultaneous moves, so after LEI_SYNC_CAP sweeps the
// round schedule (run_phase: safe winners, Q strictly increasing) finishes the phase.  Config 5's graph: 16 + 16 sweeps of two
// launches instead of 856 rounds of three.
#define LEI_PICKLESS 3
#define LEI_SYNC_CAP 48
static int lei_round_default(int N) { return (int)std::min<long long>(16384, std::max<long long>(256, N / 32)); }
static long long run_phase_sync(mn_graph *g, mn_comm *c, LeiArgs a, int period, int64_t *sweeps_out) {
    hipStream_t st = g->stream;
    LeiWork &w = *g->work;
    const int N = g->n, nbN = (N + 255) / 256;
    const bool hashed = !g->weighted;
    const int sg = lei_sub_group(g, a.use_both);
    int cap = LEI_SYNC_CAP;
    if (const char *e = getenv("MN_LEIDEN_SYNC_CAP")) // tuning knob (the oracle reads ORC_LEI_SYNC_CAP the same way)
        cap = atoi(e);
    // Several GPUs (mn_graph_leiden_shared): every rank holds the graph and the whole state; the evaluation — four fifths of a
    // sweep — is divided by node range, the decisions are all-gathered (N ints per sweep) and every replica applies ALL of them,
    // so every rank walks through the same states as one GPU does and ends with its bits.
    const int world = c ? c->world : 1, rank = c ? c->rank : 0;
    const int per = (N + world - 1) / world;
    const int v0 = std::min(N, rank * per), v1 = std::min(N, v0 + per);
    const int bg0 = (int)(std::lower_bound(w.h_big.begin(), w.h_big.end(), v0) - w.h_big.begin());
    const int bg1 = (int)(std::lower_bound(w.h_big.begin(), w.h_big.end(), v1) - w.h_big.begin());
    int *const dec_all = a.dec;
    a.sync = 1;
    a.parity = 0;
    long long total = 0;
    int sweeps = 0;
    bool converged = false;
    while (sweeps < cap && sweeps < a.max_sweeps) {
        sweeps++;
        a.pickless = period > 0 && sweeps % period == 0;
        // (table size is a matter of speed only: full-size while most neighbours still carry labels of their own)
        a.sg_log2h = sweeps <= 3 || getenv("MN_LEIDEN_FULL_TABLES") ? LEI_SG_LOG2H : LEI_SG_LOG2H - 1;
        GCHK(hipMemsetAsync(a.out, 0, sizeof(int), st));
        a.b0 = v0;
        a.b1 = v1;
        a.big0 = bg0;
        a.big1 = bg1;
        a.dec = dec_all + v0; // (the evaluation stores decision v at dec[v - b0])
        if (v1 > v0)
            lei_launch_eval(a, v1 - v0, sg, hashed, st);
        if (world > 1) {
            int failed = -1;
            const int ag = mn_comm_agree(c, hipGetLastError() != hipSuccess ? 1 : 0, st, &failed);
            if (ag != 0) {
                if (ag < 0)
                    gset_err("mn_graph_leiden_shared: %s", mn_comm_last_error_str());
                else
                    gset_err("mn_graph_leiden_shared: rank %d failed; all ranks stop", failed);
                return -1;
            }
            if (mn_comm_allgather_dev(c, dec_all + (size_t)rank * per, dec_all, (size_t)per * sizeof(int), st)) {
                gset_err("mn_graph_leiden_shared: %s", mn_comm_last_error_str());
                return -1;
            }
        }
        a.b0 = 0;
        a.b1 = N;
        a.dec = dec_all;
        if (!a.apply_on_device) {
            const int n_ops = 2 * N;
            hipLaunchKernelGGL(k_leiden_ops, dim3(nbN), dim3(256), 0, st, a, N, w.okeys);
            size_t bytes = w.osort_bytes;
            if (rocprim::radix_sort_pairs(w.osort_tmp, bytes, w.okeys, w.okeys_s, w.oiota, w.oops_s, (size_t)n_ops, 0, w.osort_bits, st) !=
                hipSuccess) {
                gset_err("rocprim::radix_sort_pairs failed");
                return -1;
            }
            hipLaunchKernelGGL(k_leiden_apply_ops, dim3((n_ops + 255) / 256), dim3(256), 0, st, a, N, w.okeys_s, w.oops_s, n_ops);
        }
        hipLaunchKernelGGL(k_leiden_apply_sync, dim3(nbN), dim3(256), 0, st, a);
        GCHK(hipGetLastError());
        GCHK(hipMemcpyAsync(w.h_out, a.out, sizeof(int), hipMemcpyDeviceToHost, st));
        GCHK(hipStreamSynchronize(st));
        const int mv = w.h_out[0];
        total += mv;
        if (mv == 0 && !a.pickless) {
            converged = true;
            break;
        }
    }
    a.b0 = 0;
    a.b1 = N;
    a.big0 = 0;
    a.big1 = (int)w.h_big.size();
    a.dec = dec_all;
    *sweeps_out += sweeps;
    if (!converged) {
        a.sync = 0;
        a.pickless = 0;
        a.sg_log2h = LEI_SG_LOG2H;
        const long long more = run_phase(g, a, MN_LEIDEN_BATCHED, lei_round_default(N), sweeps_out);
        if (more < 0)
            return -1;
        total += more;
    }
    return total;
}

// compute_modularity's per-node terms weight_to_community(i, community[i]) (:131): each rank computes those of its node range
// and the ranges are all-gathered — the modularity partials of several GPUs.  (Gathered per NODE rather than reduced per
// community: the per-community f64 sums are then taken in the reference's node order on every rank, the same bits as one GPU.)
static int lei_w2c_all(mn_graph *g, mn_comm *c, const DevGraph &dg, int use_both, const int *label, double *tmp) {
    hipStream_t st = g->stream;
    const int N = g->n, world = c ? c->world : 1, rank = c ? c->rank : 0;
    const int per = (N + world - 1) / world;
    const int v0 = std::min(N, rank * per), v1 = std::min(N, v0 + per);
    if (v1 > v0)
        hipLaunchKernelGGL(k_w2c_self, dim3((v1 - v0 + 255) / 256), dim3(256), 0, st, dg, use_both, label, tmp, v0, v1);
    if (world > 1 && mn_comm_allgather_dev(c, tmp + (size_t)rank * per, tmp, (size_t)per * sizeof(double), st)) {
        gset_err("mn_graph_leiden_shared: %s", mn_comm_last_error_str());
        return -1;
    }
    return 0;
}

static int leiden_impl(mn_graph *g, mn_comm *c, double resolution, int use_both, int mode, int batch, int *community_out,
                       double *modularity_out) {
    GCHK(hipSetDevice(g->device));
    const int N = g->n;
    memset(&g->stats, 0, sizeof(g->stats));
    if (modularity_out)
        *modularity_out = 0.0;
    if (N == 0)
        return 0;
    // MN_LEIDEN_BATCHED: batch 0 / 1 = the default schedule, whole-graph synchronous sweeps with a pick-less sweep every
    // LEI_PICKLESS (run_phase_sync); batch < 0 = the same with period -batch; batch > 1 = rounds of `batch` nodes with the
    // safe-winner commit rule (run_phase) — also what finishes a synchronous phase that does not settle.
    int period = 0;
    if (mode == MN_LEIDEN_BATCHED && batch <= 1) {
        period = batch < 0 ? -batch : LEI_PICKLESS;
        if (const char *e = getenv("MN_LEIDEN_BATCH")) { // tuning knob: MN_LEIDEN_BATCH=<round size> selects the round schedule
            if (atoi(e) > 1) {
                period = 0;
                batch = std::max(256, atoi(e));
            }
        }
        if (period)
            batch = lei_round_default(N); // the rounds a synchronous phase falls back to
    }
    hipStream_t st = g->stream;
    DevGraph dg = {N, g->off_out, g->tgt_out, g->w_out, g->off_in, g->tgt_in, g->w_in};
    const int max_deg = ((use_both ? g->max_deg_both : g->max_deg_out) + 7) & ~3; // int4-aligned scratch stride
    // buffers hold the largest round of the schedule (run_phase's tail rule)
    int grow_cap, div_unused;
    lei_grow_setting(&grow_cap, &div_unused);
    const int round_cap = (int)std::min<long long>((long long)batch * grow_cap, std::max(batch, N));
    if (lei_prepare(g, mode, mode == MN_LEIDEN_BATCHED ? (period ? std::max(round_cap, N) : round_cap) : batch, use_both, max_deg))
        return -1;
    LeiWork &d = *g->work;
    const int nbN = (N + 255) / 256;
    // Unweighted: every per-community / per-graph sum is a sum of integers (exact in any order) → all bookkeeping on the
    // device.  Weighted: the reference's node-order f64 sums are kept on the host.
    const bool on_dev = !g->weighted && mode == MN_LEIDEN_BATCHED;
    GCHK(hipEventRecord(g->ev0, st));
    // k[i], m (:344-350)
    hipLaunchKernelGGL(k_wdeg, dim3(nbN), dim3(256), 0, st, dg, use_both, d.kdeg);
    std::vector<double> k;
    double m = 0.0;
    if (on_dev) {
        GCHK(hipMemsetAsync(d.scal, 0, sizeof(double), st));
        hipLaunchKernelGGL(k_sum_d, dim3(std::min(nbN, 1024)), dim3(256), 0, st, d.kdeg, N, d.scal);
        GCHK(hipMemcpyAsync(&m, d.scal, sizeof(double), hipMemcpyDeviceToHost, st));
        GCHK(hipStreamSynchronize(st));
    } else {
        k.resize((size_t)N);
        GCHK(hipMemcpyAsync(k.data(), d.kdeg, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, st));
        GCHK(hipStreamSynchronize(st));
        for (int i = 0; i < N; i++) // the running total in node order
            m += k[i];
    }
    m /= 2.0;
    std::vector<int> community, refined;
    if (m <= 0.0) { // :351-356
        for (int i = 0; i < N; i++)
            community_out[i] = i;
        return 0;
    }
    hipLaunchKernelGGL(k_iota, dim3(nbN), dim3(256), 0, st, d.label, N);
    GCHK(hipMemcpyAsync(d.sum_tot, d.kdeg, (size_t)N * sizeof(double), hipMemcpyDeviceToDevice, st));
    std::vector<double> sum_tot;
    if (!on_dev) {
        community.resize((size_t)N);
        refined.resize((size_t)N);
        for (int i = 0; i < N; i++)
            community[i] = i;
        sum_tot = k;
    }

    LeiArgs a;
    memset(&a, 0, sizeof(a));
    a.g = dg;
    a.kdeg = d.kdeg;
    a.m = m;
    a.resolution = resolution;
    a.use_both = use_both;
    a.scratch_c = d.sc;
    a.scratch_w = d.sw;
    a.scratch_e = d.se;
    a.max_deg = max_deg;
    a.out = d.out;
    a.max_sweeps = 100000;
    a.dec = d.dec;
    a.cmin = d.cmin;
    a.win = d.win;
    a.mv = d.mv;
    a.dk = d.dk;
    a.Jq = d.Jq;
    a.Lq = d.Lq;
    a.apply_on_device = g->weighted ? 0 : 1;
    a.lds_cap = std::min(LEI_CAP, std::max(64, (max_deg + 15) & ~15));
    a.big_log2h = 8; // a wide node's table: one entry per edge at most (only other communities enter it)
    while ((1 << a.big_log2h) < a.lds_cap)
        a.big_log2h++;
    a.sg_log2h = LEI_SG_LOG2H;
    a.biglist = d.biglist;
    a.bigoff = d.bigoff;

    for (int iter = 0; iter < 100; iter++) { // :368-417
        a.label = d.label;
        a.sum_tot = d.sum_tot;
        a.elig_part = nullptr;
        long long moves = period ? run_phase_sync(g, c, a, period, &g->stats.move_sweeps) : run_phase(g, a, mode, batch, &g->stats.move_sweeps);
        if (moves < 0)
            return -1;
        g->stats.iterations++;
        g->stats.moves += moves;
        if (moves == 0)
            break;
        // refinement (:238-312): singletons, r_sum_tot = k
        hipLaunchKernelGGL(k_iota, dim3(nbN), dim3(256), 0, st, d.refined, N);
        GCHK(hipMemcpyAsync(d.tmp, d.kdeg, (size_t)N * sizeof(double), hipMemcpyDeviceToDevice, st));
        a.label = d.refined;
        a.sum_tot = d.tmp;
        a.elig_part = d.label;
        if (!on_dev) {
            GCHK(hipMemcpyAsync(community.data(), d.label, (size_t)N * sizeof(int), hipMemcpyDeviceToHost, st));
            for (int i = 0; i < N; i++)
                refined[i] = i;
        }
        if ((period ? run_phase_sync(g, c, a, period, &g->stats.refine_sweeps) : run_phase(g, a, mode, batch, &g->stats.refine_sweeps)) < 0)
            return -1;
        if (on_dev) {
            // :388-408 adopt the refinement iff it has no more communities than phase 1; then renumber (:317-331)
            // and rebuild sum_tot (:413-416) — on the device
            int cnt[2] = {0, 0};
            if (dev_distinct(g, d.label, 0) || dev_distinct(g, d.refined, 1))
                return -1;
            GCHK(hipMemcpyAsync(cnt, d.counts, sizeof(cnt), hipMemcpyDeviceToHost, st));
            GCHK(hipStreamSynchronize(st));
            if (cnt[1] <= cnt[0]) { // first[] / flag[] describe `refined` (the later of the two calls)
                GCHK(hipMemcpyAsync(d.label, d.refined, (size_t)N * sizeof(int), hipMemcpyDeviceToDevice, st));
            } else if (dev_distinct(g, d.label, 0)) {
                return -1;
            }
            if (dev_renumber(g, d.label))
                return -1;
            GCHK(hipMemsetAsync(d.sum_tot, 0, (size_t)N * sizeof(double), st));
            hipLaunchKernelGGL(k_scatter_add, dim3(nbN), dim3(256), 0, st, d.label, d.kdeg, N, d.sum_tot);
        } else {
            GCHK(hipMemcpyAsync(refined.data(), d.refined, (size_t)N * sizeof(int), hipMemcpyDeviceToHost, st));
            GCHK(hipStreamSynchronize(st));
            if (distinct(refined) <= distinct(community)) // :388-408
                community = refined;
            renumber(community);
            std::fill(sum_tot.begin(), sum_tot.end(), 0.0); // :413-416, node order
            for (int i = 0; i < N; i++)
                sum_tot[community[i]] += k[i];
            GCHK(hipMemcpyAsync(d.label, community.data(), (size_t)N * sizeof(int), hipMemcpyHostToDevice, st));
            GCHK(hipMemcpyAsync(d.sum_tot, sum_tot.data(), (size_t)N * sizeof(double), hipMemcpyHostToDevice, st));
        }
    }
    int K = 0;
    std::vector<double> s_in, s_tot;
    if (on_dev) {
        int cnt = 0;
        if (dev_distinct(g, d.label, 0) || dev_renumber(g, d.label)) // :420
            return -1;
        // compute_modularity (:109-142): per-node terms and per-community sums (integers) on the device
        if (lei_w2c_all(g, c, dg, use_both, d.label, d.tmp))
            return -1;
        GCHK(hipMemsetAsync(d.s_in, 0, (size_t)N * sizeof(double), st));
        GCHK(hipMemsetAsync(d.sum_tot, 0, (size_t)N * sizeof(double), st));
        hipLaunchKernelGGL(k_scatter_add, dim3(nbN), dim3(256), 0, st, d.label, d.tmp, N, d.s_in);
        hipLaunchKernelGGL(k_scatter_add, dim3(nbN), dim3(256), 0, st, d.label, d.kdeg, N, d.sum_tot);
        GCHK(hipMemcpyAsync(&cnt, d.counts, sizeof(int), hipMemcpyDeviceToHost, st));
        GCHK(hipMemcpyAsync(community_out, d.label, (size_t)N * sizeof(int), hipMemcpyDeviceToHost, st));
        GCHK(hipStreamSynchronize(st));
        K = cnt;
        s_in.resize((size_t)K);
        s_tot.resize((size_t)K);
        GCHK(hipMemcpyAsync(s_in.data(), d.s_in, (size_t)K * sizeof(double), hipMemcpyDeviceToHost, st));
        GCHK(hipMemcpyAsync(s_tot.data(), d.sum_tot, (size_t)K * sizeof(double), hipMemcpyDeviceToHost, st));
        GCHK(hipEventRecord(g->ev1, st));
        GCHK(hipStreamSynchronize(st));
    } else {
        GCHK(hipMemcpyAsync(community.data(), d.label, (size_t)N * sizeof(int), hipMemcpyDeviceToHost, st));
        GCHK(hipStreamSynchronize(st));
        K = renumber(community); // :420
        // compute_modularity (:109-142): per-node terms on the device, accumulation in node order here
        GCHK(hipMemcpyAsync(d.label, community.data(), (size_t)N * sizeof(int), hipMemcpyHostToDevice, st));
        if (lei_w2c_all(g, c, dg, use_both, d.label, d.tmp))
            return -1;
        std::vector<double> w2c((size_t)N);
        GCHK(hipMemcpyAsync(w2c.data(), d.tmp, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, st));
        GCHK(hipEventRecord(g->ev1, st));
        GCHK(hipStreamSynchronize(st));
        s_in.assign((size_t)K, 0.0);
        s_tot.assign((size_t)K, 0.0);
        for (int i = 0; i < N; i++) {
            s_tot[community[i]] += k[i];
            s_in[community[i]] += w2c[i];
        }
        memcpy(community_out, community.data(), (size_t)N * sizeof(int));
    }
    double Q = 0.0; // :128-139, communities in order (K terms)
    for (int c = 0; c < K; c++)
        if (s_tot[c] > 0)
            Q += s_in[c] / (2.0 * m) - resolution * (s_tot[c] / (2.0 * m)) * (s_tot[c] / (2.0 * m));
    float ms = 0;
    if (hipEventElapsedTime(&ms, g->ev0, g->ev1) == hipSuccess)
        g->stats.device_ms = ms;
    g->stats.n_communities = K;
    if (modularity_out)
        *modularity_out = Q;
    return 0;
}

extern "C" int mn_graph_leiden(mn_graph *g, double resolution, int use_both, int mode, int batch, int *community_out,
                               double *modularity_out) try {
    return leiden_impl(g, nullptr, resolution, use_both, mode, batch, community_out, modularity_out);
} MN_GUARD_END(gset_err, MN_NOTHING, -1)

// run_leiden on the GPUs of a communicator (north_star: the local-move sweep "partitioned across the GPUs ... modularity
// partials").  Every rank holds the whole graph; the parallel schedule's sweeps are divided by node range
// (run_phase_sync) and so are the modularity's per-node terms (lei_w2c_all).  Every rank returns the communities and Q that
// mn_graph_leiden(MN_LEIDEN_BATCHED, batch) returns on one GPU, bit for bit.  batch as there (0: the default schedule).
extern "C" int mn_graph_leiden_shared(mn_graph *g, mn_comm *c, double resolution, int use_both, int batch, int *community_out,
                                      double *modularity_out) try {
    if (c && c->world > 64) {
        gset_err("mn_graph_leiden_shared: more than 64 ranks");
        return -1;
    }
    return leiden_impl(g, c, resolution, use_both, MN_LEIDEN_BATCHED, batch, community_out, modularity_out);
} MN_GUARD_END(gset_err, MN_NOTHING, -1)

extern "C" int mn_graph_leiden_stats(mn_graph *g, mn_leiden_stats *out) try {
    *out = g->stats;
    return 0;
} MN_GUARD_END(gset_err, MN_NOTHING, -1)

// ───────────────────────── Brandes betweenness (src/graph_centrality.c:260-505) ─────────────────────────
// The reference runs one single-source shortest-path pass per source — BFS for unweighted graphs, Dijkstra with a lazy binary
// heap for weighted ones, predecessor lists in discovery order — then the dependency accumulation in reverse stack order,
// and adds every source's result to CB / EB in source order (f64: the order of those additions is part of the result).
// Sources are independent, so the device runs them side by side: ONE LANE PER SOURCE replays the reference's pass verbatim
// on that source's own scratch rows (queue, stack, predecessor lists, heap — the same control flow, hence the same
// stack order, predecessor order and f64 operations), and k_brandes_accumulate then folds the sources of the chunk into
// CB[w] / EB[v][w] in source order, one lane per target w (a cell is only ever written by w's lane).  The dependency of w
// at the moment the reference pops it is its final value, so the flow (sigma[v] / sigma[w]) * (1 + delta[w]) is recomputed
// there from the stored sigma / delta with the same operands.  Divergent by construction (64 different traversals per
// wavefront): this trades SIMD efficiency for bit-exact reference semantics; throughput comes from thousands of sources
// in flight.
struct BrDpq {
    int node;
    double dist;
};
struct BrCell {
    double dist, sigma, delta;
    int pcnt, pad;
};
struct BrArgs {
    DevGraph g;
    int use_out, use_in, weighted;
    int n_src;           // sources in this chunk
    const int *sources;  // [n_src]
    const int *poff;     // [N+1] predecessor-list slots per node (static: one per incident traversed edge)
    long long P;         // poff[N]
    long long heap_cap;  // Dijkstra: entries per source
    BrCell *cell;        // [n_src][N]   what a pass keeps per node, side by side (one line per touch of a node instead of three)
    int *stack, *queue;  // [n_src][N]   (queue doubles as Dijkstra's settled flags)
    int *pitems;                  // [n_src][P]
    BrDpq *heap;                  // [n_src][heap_cap]
    int *overflow;
};

DEVI bool br_double_eq(double a, double b) { return fabs(a - b) < 1e-10 * fmax(1.0, fabs(b)); } // :215-217

__global__ void __launch_bounds__(64) k_brandes_sources(BrArgs a) {
    const int si = blockIdx.x * blockDim.x + threadIdx.x;
    if (si >= a.n_src)
        return;
    const int N = a.g.n, src = a.sources[si];
    BrCell *c = a.cell + (size_t)si * N;
    int *stack = a.stack + (size_t)si * N, *queue = a.queue + (size_t)si * N;
    int *pitems = a.pitems + (size_t)si * a.P;
    for (int i = 0; i < N; i++) {
        c[i] = BrCell{-1.0, 0.0, 0.0, 0, 0};
        if (a.weighted)
            queue[i] = 0; // (Dijkstra's settled flags; the BFS writes a queue position before it reads it)
    }
    c[src].dist = 0.0;
    c[src].sigma = 1.0;
    int ss = 0;
    if (!a.weighted) { // sssp_bfs, :263-315
        int qh = 0, qt = 0;
        queue[qt++] = src;
        while (qh < qt) {
            const int v = queue[qh++];
            stack[ss++] = v;
            for (int pass = 0; pass < 2; pass++) {
                if (pass == 0 ? !a.use_out : !a.use_in)
                    continue;
                const int *off = pass ? a.g.off_in : a.g.off_out, *tgt = pass ? a.g.tgt_in : a.g.tgt_out;
                for (int e = off[v]; e < off[v + 1]; e++) {
                    const int w = tgt[e];
                    if (c[w].dist < 0) {
                        c[w].dist = c[v].dist + 1.0;
                        queue[qt++] = w;
                    }
                    if (br_double_eq(c[w].dist, c[v].dist + 1.0)) {
                        const int pc = c[w].pcnt;
                        if (pc == 0 || pitems[a.poff[w] + pc - 1] != v) {
                            c[w].sigma += c[v].sigma;
                            pitems[a.poff[w] + pc] = v;
                            c[w].pcnt = pc + 1;
                        }
                    }
                }
            }
        }
    } else { // sssp_dijkstra, :321-378, with dpq_push / dpq_pop (:158-212) verbatim
        BrDpq *h = a.heap + (size_t)si * a.heap_cap;
        int hs = 0;
        int *settled = queue;
        h[hs].node = src;
        h[hs].dist = 0.0;
        hs++;
        while (hs > 0) {
            const BrDpq top = h[0];
            hs--;
            if (hs > 0) {
                h[0] = h[hs];
                int i = 0;
                for (;;) {
                    const int left = 2 * i + 1, right = 2 * i + 2;
                    int smallest = i;
                    if (left < hs && h[left].dist < h[smallest].dist)
                        smallest = left;
                    if (right < hs && h[right].dist < h[smallest].dist)
                        smallest = right;
                    if (smallest == i)
                        break;
                    const BrDpq t = h[i];
                    h[i] = h[smallest];
                    h[smallest] = t;
                    i = smallest;
                }
            }
            const int v = top.node;
            if (settled[v])
                continue;
            settled[v] = 1;
            stack[ss++] = v;
            for (int pass = 0; pass < 2; pass++) {
                if (pass == 0 ? !a.use_out : !a.use_in)
                    continue;
                const int *off = pass ? a.g.off_in : a.g.off_out, *tgt = pass ? a.g.tgt_in : a.g.tgt_out;
                const double *wt = pass ? a.g.w_in : a.g.w_out;
                for (int e = off[v]; e < off[v + 1]; e++) {
                    const int w = tgt[e];
                    const double nd = c[v].dist + (wt ? wt[e] : 1.0);
                    if (c[w].dist < 0 || nd < c[w].dist - 1e-10) {
                        c[w].dist = nd;
                        c[w].sigma = c[v].sigma;
                        pitems[a.poff[w]] = v;
                        c[w].pcnt = 1;
                        if (hs >= a.heap_cap) {
                            *a.overflow = 1;
                            return;
                        }
                        int i = hs++;
                        h[i].node = w;
                        h[i].dist = nd;
                        while (i > 0) {
                            const int parent = (i - 1) / 2;
                            if (h[parent].dist <= h[i].dist)
                                break;
                            const BrDpq t = h[parent];
                            h[parent] = h[i];
                            h[i] = t;
                            i = parent;
                        }
                    } else if (br_double_eq(nd, c[w].dist)) {
                        const int pc = c[w].pcnt;
                        if (pc == 0 || pitems[a.poff[w] + pc - 1] != v) {
                            if (pc >= a.poff[w + 1] - a.poff[w]) { // (cannot happen: one slot per incident edge)
                                *a.overflow = 1;
                                return;
                            }
                            c[w].sigma += c[v].sigma;
                            pitems[a.poff[w] + pc] = v;
                            c[w].pcnt = pc + 1;
                        }
                    }
                }
            }
        }
    }
    // dependency accumulation in reverse stack order (:448-462)
    while (ss > 0) { // (delta is zero from the start: the passes above never touch it)
        const int w = stack[--ss];
        const int pc = c[w].pcnt;
        for (int pi = 0; pi < pc; pi++) {
            const int v = pitems[a.poff[w] + pi];
            if (c[w].sigma > 0) {
                const double flow = (c[v].sigma / c[w].sigma) * (1.0 + c[w].delta);
                c[v].delta += flow;
            }
        }
    }
}

// CB[w] += delta_s[w] (w != s) and EB[v*N + w] += flow, sources of the chunk in order; one lane per target w
__global__ void k_brandes_accumulate(BrArgs a, double *CB, double *EB) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = a.g.n;
    if (w >= N)
        return;
    double cb = CB[w];
    for (int si = 0; si < a.n_src; si++) {
        const BrCell *c = a.cell + (size_t)si * N;
        const double dw = c[w].delta;
        if (EB) {
            const int pc = c[w].pcnt;
            const int *items = a.pitems + (size_t)si * a.P + a.poff[w];
            const double sw = c[w].sigma;
            for (int pi = 0; pi < pc; pi++) {
                const int v = items[pi];
                if (sw > 0)
                    EB[(size_t)v * N + w] += (c[v].sigma / sw) * (1.0 + dw);
            }
        }
        if (w != a.sources[si])
            cb += dw;
    }
    CB[w] = cb;
}

__global__ void k_scale_d(double *x, long long n, double mul, double div1, double div2) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    double v = x[i];
    if (mul != 1.0)
        v *= mul;
    if (div1 != 1.0)
        v /= div1;
    if (div2 != 1.0)
        v /= div2;
    x[i] = v;
}

extern "C" int mn_graph_betweenness(mn_graph *g, int direction, int auto_approx, int normalized, double *cb_out, double *eb_out) try {
    GCHK(hipSetDevice(g->device));
    const int N = g->n;
    if (N == 0)
        return 0;
    if (direction < 0 || direction > 2) {
        gset_err("mn_graph_betweenness: direction must be 0 (both), 1 (forward) or 2 (reverse)");
        return -1;
    }
    const int use_out = direction != 2, use_in = direction == 2 || direction == 0; // :281-282
    hipStream_t st = g->stream;
    // source set (:417-433)
    std::vector<int> sources;
    double scale = 1.0;
    if (auto_approx > 0 && N > auto_approx) {
        const int want = (int)ceil(sqrt((double)N));
        int n_sources = want < 1 ? 1 : want;
        int step = N / n_sources;
        if (step < 1)
            step = 1;
        for (int i = 0; i < N && (int)sources.size() < want; i += step)
            sources.push_back(i);
        scale = (double)N / (double)sources.size();
    } else {
        for (int i = 0; i < N; i++)
            sources.push_back(i);
    }
    // predecessor slots: one per traversed edge arriving at the node
    std::vector<int> tgt_o((size_t)g->e_out), tgt_i((size_t)g->e_in), poff((size_t)N + 1, 0);
    if (g->e_out)
        GCHK(hipMemcpy(tgt_o.data(), g->tgt_out, (size_t)g->e_out * sizeof(int), hipMemcpyDeviceToHost));
    if (g->e_in)
        GCHK(hipMemcpy(tgt_i.data(), g->tgt_in, (size_t)g->e_in * sizeof(int), hipMemcpyDeviceToHost));
    long long e_trav = 0;
    if (use_out)
        for (int x : tgt_o) {
            poff[(size_t)x + 1]++;
            e_trav++;
        }
    if (use_in)
        for (int x : tgt_i) {
            poff[(size_t)x + 1]++;
            e_trav++;
        }
    for (int i = 0; i < N; i++)
        poff[(size_t)i + 1] += poff[(size_t)i];
    const long long P = poff[(size_t)N] > 0 ? poff[(size_t)N] : 1;
    const long long heap_cap = g->weighted ? e_trav + 2 : 1;
    // chunk of sources that fits the scratch budget
    const size_t per_src = (size_t)N * (sizeof(BrCell) + 2 * sizeof(int)) + (size_t)P * sizeof(int) + (size_t)heap_cap * sizeof(BrDpq);
    // Half of what the device has free (round 4): the lanes of a launch are the only parallelism there is, and at the 8 GB of
    // rounds 2-3 a 20 000-node graph went through in six launches of 58 wavefronts each on a chip of 1 024 SIMDs.
    size_t budget = (size_t)8 << 30, free_b = 0, total_b = 0;
    const size_t eb_bytes = eb_out ? (size_t)N * N * sizeof(double) : 0; // (allocated after the scratch: leave it its room)
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
        budget = std::max<size_t>((size_t)256 << 20, (free_b > eb_bytes ? free_b - eb_bytes : 0) / 2);
    if (const char *e = getenv("MN_BRANDES_SCRATCH_MB"))
        budget = (size_t)atoll(e) << 20;
    int chunk = (int)std::max<size_t>(1, std::min<size_t>(sources.size(), budget / per_src));
    struct Scr {
        std::vector<void *> p;
        ~Scr() {
            for (void *q : p)
                (void)hipFree(q);
        }
        void *get(size_t bytes) {
            void *q = nullptr;
            if (hipMalloc(&q, bytes ? bytes : 16) != hipSuccess)
                return nullptr;
            p.push_back(q);
            return q;
        }
    } scr;
    BrArgs a;
    memset(&a, 0, sizeof(a));
    a.g = {N, g->off_out, g->tgt_out, g->w_out, g->off_in, g->tgt_in, g->w_in};
    a.use_out = use_out;
    a.use_in = use_in;
    a.weighted = g->weighted ? 1 : 0;
    a.P = P;
    a.heap_cap = heap_cap;
    int *d_sources = (int *)scr.get((size_t)chunk * sizeof(int)), *d_poff = (int *)scr.get(((size_t)N + 1) * sizeof(int));
    a.cell = (BrCell *)scr.get((size_t)chunk * N * sizeof(BrCell));
    a.stack = (int *)scr.get((size_t)chunk * N * sizeof(int));
    a.queue = (int *)scr.get((size_t)chunk * N * sizeof(int));
    a.pitems = (int *)scr.get((size_t)chunk * P * sizeof(int));
    a.heap = (BrDpq *)scr.get((size_t)chunk * heap_cap * sizeof(BrDpq));
    a.overflow = (int *)scr.get(sizeof(int));
    double *d_cb = (double *)scr.get((size_t)N * sizeof(double));
    double *d_eb = eb_out ? (double *)scr.get((size_t)N * N * sizeof(double)) : nullptr;
    if (!d_sources || !d_poff || !a.cell || !a.stack || !a.queue || !a.pitems || !a.heap ||
        !a.overflow || !d_cb || (eb_out && !d_eb)) {
        gset_err("mn_graph_betweenness: out of device memory (N = %d%s)", N, eb_out ? ", dense N x N edge matrix as in the reference" : "");
        return -1;
    }
    a.sources = d_sources;
    a.poff = d_poff;
    GCHK(hipMemcpyAsync(d_poff, poff.data(), ((size_t)N + 1) * sizeof(int), hipMemcpyHostToDevice, st));
    GCHK(hipMemsetAsync(d_cb, 0, (size_t)N * sizeof(double), st));
    GCHK(hipMemsetAsync(a.overflow, 0, sizeof(int), st));
    if (d_eb)
        GCHK(hipMemsetAsync(d_eb, 0, (size_t)N * N * sizeof(double), st));
    GCHK(hipEventRecord(g->ev0, st));
    for (size_t s0 = 0; s0 < sources.size(); s0 += (size_t)chunk) {
        a.n_src = (int)std::min<size_t>((size_t)chunk, sources.size() - s0);
        GCHK(hipMemcpyAsync(d_sources, sources.data() + s0, (size_t)a.n_src * sizeof(int), hipMemcpyHostToDevice, st));
        // One lane per source, and every lane a chain of dependent accesses to its own rows: what hides the latency is the
        // number of wavefronts, not their width.  Narrow workgroups (4 to 64 lanes) until the launch has some 4 096 of them —
        // 20 000 sources are 5 000 four-lane wavefronts, five per SIMD, instead of 313 full ones on a third of the SIMDs; a
        // divergent memory instruction also costs one address cycle per distinct line, so narrow wavefronts lose nothing.
        int lanes = 64;
        if (const char *e = getenv("MN_BRANDES_LANES"))
            lanes = std::max(1, std::min(64, atoi(e)));
        else
            while (lanes > 4 && (a.n_src + lanes - 1) / lanes < 4096)
                lanes >>= 1;
        hipLaunchKernelGGL(k_brandes_sources, dim3((a.n_src + lanes - 1) / lanes), dim3(lanes), 0, st, a);
        hipLaunchKernelGGL(k_brandes_accumulate, dim3((N + 255) / 256), dim3(256), 0, st, a, d_cb, d_eb);
        GCHK(hipStreamSynchronize(st)); // (the host vector `sources` chunk must outlive the copy; also bounds the queue)
    }
    // approximation scale, undirected halving, normalisation — in the reference's order (:466-498)
    const int undirected = direction == 0;
    const double half = undirected ? 2.0 : 1.0;
    double norm = 1.0;
    if (normalized && N > 2)
        norm = undirected ? (double)(N - 1) * (double)(N - 2) / 2.0 : (double)(N - 1) * (double)(N - 2);
    hipLaunchKernelGGL(k_scale_d, dim3((N + 255) / 256), dim3(256), 0, st, d_cb, (long long)N, scale, half, norm);
    if (d_eb)
        hipLaunchKernelGGL(k_scale_d, dim3((unsigned)(((long long)N * N + 255) / 256)), dim3(256), 0, st, d_eb, (long long)N * N, scale,
                           half, norm);
    GCHK(hipEventRecord(g->ev1, st));
    GCHK(hipGetLastError());
    int ovf = 0;
    GCHK(hipMemcpyAsync(&ovf, a.overflow, sizeof(int), hipMemcpyDeviceToHost, st));
    GCHK(hipMemcpyAsync(cb_out, d_cb, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, st));
    if (d_eb)
        GCHK(hipMemcpyAsync(eb_out, d_eb, (size_t)N * N * sizeof(double), hipMemcpyDeviceToHost, st));
    GCHK(hipStreamSynchronize(st));
    if (ovf) {
        gset_err("mn_graph_betweenness: scratch overflow");
        return -1;
    }
    float ms = 0;
    if (hipEventElapsedTime(&ms, g->ev0, g->ev1) == hipSuccess)
        g->last_ms = ms;
    return 0;
} MN_GUARD_END(gset_err, MN_NOTHING, -1)

extern "C" double mn_graph_last_ms(mn_graph *g) { return g->last_ms; }

// GraphData.out as the host sees it (graph_edge_betweenness emits its rows in this order, src/graph_centrality.c:1172-1182)
extern "C" long long mn_graph_out_edge_count(mn_graph *g) { return g->e_out; }
extern "C" int mn_graph_out_lists(mn_graph *g, int *off, int *tgt) try {
    GCHK(hipSetDevice(g->device));
    memcpy(off, g->h_off_out.data(), ((size_t)g->n + 1) * sizeof(int));
    if (g->e_out)
        GCHK(hipMemcpy(tgt, g->tgt_out, (size_t)g->e_out * sizeof(int), hipMemcpyDeviceToHost));
    return 0;
} MN_GUARD_END(gset_err, MN_NOTHING, -1)
