// mn_spec.hip — speculative execution of the reference's one-at-a-time insert (src/hnsw_algo.c:520-666).
//
// hnsw_insert is a chain: insert j searches the graph that inserts 0..j-1 left behind.  k_insert_seq (mn_seq.hip)
// honours that with a single wavefront.  Here a WINDOW of consecutive inserts has its searches run at once against
// the graph as it stands (k_beam<BUILD>, one wavefront each, every link row read is logged), and one workgroup then
// commits the window in order:
//   insert j is valid  ⇔  none of the link rows its search read was rewritten by inserts 0..j-1 of the window
//                          (rows carry the epoch of their last rewrite; the first insert of a window is always valid)
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
    int epoch;
    int *ncommit;     // out: inserts committed (>= 1)
    size_t wave_bytes; // LDS per wavefront
};

DEVI int *spec_row(const MnDevIndex &ix, int node, int level) {
    if (level == 0)
        return ix.links0 + (size_t)node * ix.W0;
    return ix.links_up + ((size_t)ix.up_off[node] + (level - 1)) * ix.WU;
}

DEVI void spec_stamp(const MnDevIndex &ix, const MnSpecArgs &a, int node, int level) {
    int *p = level == 0 ? a.stamp0 + node : a.stampU + (ix.up_off[node] + level - 1);
    __hip_atomic_store(p, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int ORDER, int NCH>
__global__ void __launch_bounds__(MN_SPEC_MAX_WAVES * 64) k_spec_commit(MnDevIndex ix, MnSpecArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, NW = blockDim.x >> 6;
    int *flag = reinterpret_cast<int *>(smem); // [0] invalid insert  [1] tie in this (insert, layer)
    int *act = flag + 16;                      // [64] per target: 0 nothing, 1 append, 2 pruned row
    int *cntA = act + 64;                      // [64] append position
    int *newrow = cntA + 64;                   // [64][64]
    unsigned char *pw = reinterpret_cast<unsigned char *>(newrow + 64 * 64) + (size_t)wv * a.wave_bytes;
    int *list = reinterpret_cast<int *>(pw);            // [128]
    float *nd = reinterpret_cast<float *>(list + 128);  // [128]
    int *mn = reinterpret_cast<int *>(nd + 128);        // [128]
    float *tv = reinterpret_cast<float *>(mn + 128);    // [ld]

    int done = a.W;
    for (int jj = 0; jj < a.W; jj++) {
        const int s = a.slots[jj];
        // ── did an earlier insert of this window rewrite a row this search read? ──
        if (tid == 0)
            flag[0] = 0;
        __syncthreads();
        if (jj > 0) {
            const int nr = a.nread[jj];
            if (nr > a.readcap) {
                if (tid == 0)
                    flag[0] = 1; // incomplete log: cannot be trusted
            } else {
                const int *log = a.readlog + (size_t)jj * a.readcap;
                for (int i = tid; i < nr; i += blockDim.x) {
                    const int r = log[i];
                    const int *p = r >= 0 ? a.stamp0 + r : a.stampU + (-r - 2);
                    if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.epoch)
                        flag[0] = 1;
                }
            }
        }
        __syncthreads();
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
            // pass 1: decide every target's new row, write nothing
            for (int i = wv; i < ns; i += NW) {
                const int t = sel[i];
                int what = 0, where = 0;
                if (ix.levels[t] >= l) { // :590
                    int *trow = spec_row(ix, t, l);
                    const int v = lane < W ? ld_link<true>(trow + lane) : -1;
                    const int cnt = __popcll(__ballot(v >= 0));
                    if (__ballot(v == s)) { // already a neighbour (:147-150)
                        what = 0;
                    } else if (cnt < W) {
                        what = 1;
                        where = cnt;
                    } else { // over-full: MN-RU prune of t's list (:601-646)
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
                        if (prune_row<ORDER, NCH, true, false>(ix, tv, tnorm, list, nd, mn, W + 1, W, l, lane)) {
                            if (lane == 0)
                                flag[1] = 1; // a tie: this layer is redone in list order below
                        } else {
                            what = 2;
                            if (lane < W)
                                newrow[i * 64 + lane] = list[lane];
                        }
                    }
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
                    if (act[i] == 1) {
                        if (lane == 0)
                            st_link(trow + cntA[i], s);
                    } else if (lane < W) {
                        st_link(trow + lane, newrow[i * 64 + lane]);
                    }
                    if (lane == 0)
                        spec_stamp(ix, a, t, l);
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
                        if (lane == 0) {
                            st_link(trow + cnt, s);
                            spec_stamp(ix, a, t, l);
                        }
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
                    if (lane < W)
                        st_link(trow + lane, list[lane]);
                    if (lane == 0)
                        spec_stamp(ix, a, t, l);
                    __builtin_amdgcn_s_waitcnt(0);
                    __builtin_amdgcn_wave_barrier();
                }
            }
            __threadfence();
            __syncthreads();
        }
    }
    if (tid == 0)
        *a.ncommit = done;
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
                           const int *readlog, int readcap, const int *nread, int *stamp0, int *stampU, int epoch, int *d_ncommit,
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
    a.epoch = epoch;
    a.ncommit = d_ncommit;
    a.wave_bytes = (3 * 128 * sizeof(int) + (size_t)ix.ld * sizeof(float) + 15) & ~(size_t)15;
    const size_t shared = (16 + 64 + 64 + 64 * 64) * sizeof(int);
    int nw = (int)((60 * 1024 - shared) / a.wave_bytes);
    nw = nw < 1 ? 1 : (nw > MN_SPEC_MAX_WAVES ? MN_SPEC_MAX_WAVES : nw);
    const size_t lds = shared + (size_t)nw * a.wave_bytes;
#define MN_SP(O, N) hipLaunchKernelGGL((k_spec_commit<O, N>), dim3(1), dim3(nw * 64), lds, st, ix, a)
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
