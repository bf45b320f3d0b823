// mn_kernels.hip — hand-written gfx950 kernels for sqlite-muninn's HNSW hot path.
//
//   k_norms        |v|² per row in the index's summation order (cosine)
//   k_dist_batch   vec_*_distance(query, rows[i])                       src/vec_math.c:78-143
//   k_beam         greedy descent + ef-bounded beam search, one 64-lane wavefront per query
//                  (search flavour: hnsw_search src/hnsw_algo.c:670-704; build flavour: the search
//                  half of hnsw_insert :550-579 for every node of a batch)
//
// Execution model: one wavefront owns one query.  The candidate min-heap and the result max-heap
// (negated distances) are the reference's 1-based binary heaps (src/priority_queue.c:18-80) held in
// LDS with a global-memory spill, driven wave-uniformly so that pop order among equal distances —
// and therefore the returned id set — is the reference's.  The visited set (src/hnsw_algo.c:299-338,
// a plain set) is a per-query bitmap in HBM probed with returning atomicOr (L2-coherent, one round
// trip for the whole neighbour row).  Candidate vectors are read straight from HBM into registers:
//   MN_ORDER_WAVE  each row as coalesced float4 per lane (1 KiB per wave-instruction), 8 rows in
//                  flight, fmaf chain per lane + xor butterfly
//   MN_ORDER_SSE   16 rows at a time, lane (r,j) walks accumulator j of row r in the reference's
//                  exact order (separate mul/add, ((t0+t1)+t2)+t3, scalar tail)
// Build with -ffp-contract=off; the WAVE path uses explicit fmaf, the SSE path explicit *_rn ops.
#include "mn_device.hpp"
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <unordered_map>

#include "mn_dist.hpp"

// ───────────────────────── k_norms ─────────────────────────

template <int ORDER>
__global__ void __launch_bounds__(64) k_norms(MnDevIndex ix, int first_slot, int n, float *norms_out) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x;
    int slot = first_slot + blockIdx.x;
    if (blockIdx.x >= n)
        return;
    const float *row = ix.vectors + (size_t)slot * ix.ld;
    for (int i = lane; i < ix.ld; i += 64)
        lds[i] = row[i];
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    float v = lds_self_norm<ORDER>(lds, ix.dim, ix.ld, lane);
    if (lane == 0)
        norms_out[slot] = v;
}

void mn_launch_norms(const MnDevIndex &ix, int first_slot, int n, float *norms_out, hipStream_t st) {
    if (n <= 0)
        return;
    size_t lds = (size_t)ix.ld * sizeof(float);
    if (ix.order == MN_ORDER_SSE_V)
        hipLaunchKernelGGL(k_norms<MN_ORDER_SSE_V>, dim3(n), dim3(64), lds, st, ix, first_slot, n, norms_out);
    else
        hipLaunchKernelGGL(k_norms<MN_ORDER_WAVE_V>, dim3(n), dim3(64), lds, st, ix, first_slot, n, norms_out);
}

// ───────────────────────── k_dist_batch ─────────────────────────
// One wavefront handles up to 64 consecutive rows (WAVE: 8 in flight; SSE: 16 at a time).

template <int ORDER, int NCH>
__global__ void __launch_bounds__(64)
    k_dist_batch(int metric, const float *query, const float *rows, long long n, int dim, int ld, float *out) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x;
    for (int i = lane; i < ld; i += 64)
        lds[i] = i < dim ? query[i] : 0.0f;
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    MnDevIndex ix = {};
    ix.vectors = rows;
    ix.dim = dim;
    ix.ld = ld;
    ix.metric = metric;
    float qnorm = 0.0f;
    if (metric == 1)
        qnorm = lds_self_norm<ORDER>(lds, dim, ld, lane);
    long long base = (long long)blockIdx.x * 64;
    int cnt = (int)((n - base) < 64 ? (n - base) : 64);
    if (cnt <= 0)
        return;
    // rows are addressed as slots relative to `rows`; slots fit int32 per block via rebasing
    ix.vectors = rows + (size_t)base * ld;
    int myslot = lane < cnt ? lane : 0;
    float d;
    if (metric == 0) {
        d = rows_accumulate<ORDER, NCH, true>(ix, lds, myslot, cnt, lane);
    } else {
        float dot = rows_accumulate<ORDER, NCH, false>(ix, lds, myslot, cnt, lane);
        if (metric == 2) {
            d = -dot;
        } else {
            // |row|² in the same order: recompute per row (no cached norms for loose rows)
            float nb = 0.0f;
            for (int t = 0; t < cnt; t++) {
                const float *row = ix.vectors + (size_t)t * ld;
                // stage row into the upper half of LDS, then self-norm
                float *tmp = lds + ld;
                for (int i = lane; i < ld; i += 64)
                    tmp[i] = row[i];
                __builtin_amdgcn_s_waitcnt(0);
                __builtin_amdgcn_wave_barrier();
                float v = lds_self_norm<ORDER>(tmp, dim, ld, lane);
                __builtin_amdgcn_wave_barrier();
                if (lane == t)
                    nb = v;
            }
            d = cosine_finish(dot, qnorm, nb);
        }
    }
    if (lane < cnt)
        out[base + lane] = d;
}

static int pick_nch(int ld) {
    int need = (ld + 255) / 256;
    if (need <= 1) return 1;
    if (need <= 2) return 2;
    if (need <= 3) return 3;
    if (need <= 4) return 4;
    if (need <= 6) return 6;
    if (need <= 8) return 8;
    return 0;
}

void mn_launch_dist_batch(int metric, int order, const float *d_query, const float *d_rows, long long n, int dim, int ld,
                          float *d_out, hipStream_t st) {
    if (n <= 0)
        return;
    dim3 grid((unsigned)((n + 63) / 64)), block(64);
    size_t lds = (size_t)ld * sizeof(float) * 2;
    if (order == MN_ORDER_SSE_V) {
        hipLaunchKernelGGL((k_dist_batch<MN_ORDER_SSE_V, 0>), grid, block, lds, st, metric, d_query, d_rows, n, dim, ld, d_out);
        return;
    }
#define MN_DB(N) \
    case N:      \
        hipLaunchKernelGGL((k_dist_batch<MN_ORDER_WAVE_V, N>), grid, block, lds, st, metric, d_query, d_rows, n, dim, ld, d_out); \
        break;
    switch (pick_nch(ld)) {
        MN_DB(1) MN_DB(2) MN_DB(3) MN_DB(4) MN_DB(6) MN_DB(8)
    default:
        hipLaunchKernelGGL((k_dist_batch<MN_ORDER_WAVE_V, 0>), grid, block, lds, st, metric, d_query, d_rows, n, dim, ld, d_out);
    }
#undef MN_DB
}

#include "mn_beam.hpp"

// one query, one (leading) wavefront; `coop` = the group's shared area when helpers stand by (k_beam_coop)
// LAT: a latency-bound launch (few queries, one workgroup each): the layer searches keep their queues in registers (beam_layer_auto)
template <int ORDER, int NCH, bool BUILD, bool WIDE, bool LAT = false>
DEVI void beam_query(const MnDevIndex &ix, const MnSearchArgs &a, const long long qi, const int lane, unsigned char *smem,
                     CoopCtx *coop, unsigned *lds_bitmap = nullptr) {
    // LDS carve: cand heap | result heap | scratch | query
    uint2 *cand_l = reinterpret_cast<uint2 *>(smem);
    uint2 *res_l = cand_l + MN_CAND_LDS;
    int *scratch = reinterpret_cast<int *>(res_l + MN_RES_LDS);
    float *q = reinterpret_cast<float *>(scratch + 64);
    float *tile = q + ix.ld; // SSE order only (mn_search_lds_bytes)

    int qslot = -1;
    const float *qsrc;
    if (BUILD) {
        qslot = a.query_slots[qi];
        qsrc = ix.vectors + (size_t)qslot * ix.ld;
    } else {
        qsrc = a.queries + (size_t)qi * ix.dim;
    }
    for (int i = lane; i < ix.ld; i += 64)
        q[i] = i < ix.dim ? qsrc[i] : 0.0f;
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();

    WaveCtx w;
    w.q = q;
    w.tile = (ORDER == MN_ORDER_SSE_V && a.use_tile) ? tile : nullptr;
    w.scratch = scratch;
    w.n_dist = 0;
    w.n_exp = 0;
    w.qnorm = 0.0f;
    if (ix.metric == 1)
        w.qnorm = BUILD ? ix.norms[qslot] : lds_self_norm<ORDER>(q, ix.dim, ix.ld, lane);
    if (coop) {
        w.coop = coop;
        w.no_spec_rows = a.no_spec_rows;
        if (lane == 0)
            *coop->qnorm = w.qnorm;
    }
    if (BUILD && a.readlog) {
        w.rlog = a.readlog + (size_t)qi * a.readcap * MN_RLOG_INTS;
        w.rcap = a.readcap;
    }

    WHeap cand, res;
    cand.l = cand_l;
    cand.lcap = MN_CAND_LDS;
    cand.g = reinterpret_cast<unsigned long long *>(a.cand_ovf + (size_t)qi * a.cand_gcap);
    cand.gcap = a.cand_gcap;
    cand.size = 0;
    cand.ovf = 0;
    res.l = res_l;
    res.lcap = MN_RES_LDS;
    res.g = reinterpret_cast<unsigned long long *>(a.res_ovf + (size_t)qi * a.res_gcap);
    res.gcap = a.res_gcap;
    res.size = 0;
    res.ovf = 0;

    unsigned *bm0 = lds_bitmap ? lds_bitmap : a.bitmap0 + (size_t)qi * a.bm0_words;
    int cur = a.entry_slot;

    if (!BUILD) {
        // hnsw_search, src/hnsw_algo.c:676-703
        for (int l = a.max_level; l > 0; l--)
            cur = greedy_layer<ORDER, NCH, false, WIDE>(ix, w, cur, l, lane);
        if (LAT)
            beam_layer_auto<ORDER, NCH, false, WIDE>(ix, w, cand, res, bm0, a.bm0_words, cur, 0, a.ef, lane);
        else
            beam_layer<ORDER, NCH, false, WIDE>(ix, w, cand, res, bm0, cur, 0, a.ef, lane);
        int count = res.size;
        int outn = count < a.k ? count : a.k;
        for (int i = count - 1; i >= 0; i--) { // :436-441
            uint2 it = res_take(res, i, count, lane);
            if (i < a.k && lane == 0) {
                a.out_ids[qi * a.k + i] = ix.ids[it.y];
                a.out_dists[qi * a.k + i] = -u2f(it.x);
            }
        }
        if (lane == 0) {
            for (int i = outn; i < a.k; i++) {
                a.out_ids[qi * a.k + i] = -1;
                a.out_dists[qi * a.k + i] = 0.0f;
            }
            a.out_counts[qi] = outn;
        }
    } else {
        // search half of hnsw_insert, src/hnsw_algo.c:550-579,:650-652, against the frozen graph
        const int level = ix.levels[qslot];
        for (int l = a.max_level; l > level; l--)
            cur = greedy_layer<ORDER, NCH, false, WIDE, LAT>(ix, w, cur, l, lane); // (LAT && BUILD: a window's search, logged)
        int start = level < a.max_level ? level : a.max_level;
        for (int l = start; l >= 0; l--) {
            unsigned *bm = bm0;
            if (l > 0)
                bm = a.bitmap_up + ((size_t)a.up_bm_index[qi] * a.max_level + (l - 1)) * a.bmu_words;
            if (LAT)
                beam_layer_auto<ORDER, NCH, false, WIDE, true>(ix, w, cand, res, bm, l > 0 ? a.bmu_words : a.bm0_words, cur, l, a.ef, lane);
            else
                beam_layer<ORDER, NCH, false, WIDE>(ix, w, cand, res, bm, cur, l, a.ef, lane);
            const int M_max = (l == 0) ? ix.M0 : ix.MU;
            int count = res.size;
            int keep = count < M_max ? count : M_max;
            int *sel = a.sel + ((size_t)qi * a.nlev + l) * ix.M0;
            int first = cur;
            for (int i = count - 1; i >= 0; i--) {
                uint2 it = res_take(res, i, count, lane);
                if (i < keep && lane == 0)
                    sel[i] = (int)it.y;
                if (i == 0)
                    first = (int)it.y;
            }
            if (lane == 0)
                a.nsel[qi * a.nlev + l] = keep;
            if (count > 0)
                cur = first;
        }
    }
    if (lane == 0) {
        if (BUILD && a.readlog)
            a.nread[qi] = LAT ? w.nr : a.readcap + 1; // > readcap: the log is incomplete and the commit step must not trust it
                                                      // (only the latency kernel's searches keep a log)
        if (a.q_counters) {
            a.q_counters[(size_t)qi * 4 + 0] = w.n_dist;
            a.q_counters[(size_t)qi * 4 + 1] = w.n_exp;
            a.q_counters[(size_t)qi * 4 + 2] = (cand.ovf || res.ovf) ? 1ull : 0ull;
        } else {
            atomicAdd(&a.counters[0], w.n_dist);
            atomicAdd(&a.counters[1], w.n_exp);
            if (cand.ovf || res.ovf)
                atomicAdd(&a.counters[2], 1ull);
        }
    }
}

template <int ORDER, int NCH, bool BUILD, bool WIDE = false>
__global__ void __launch_bounds__(64) k_beam(MnDevIndex ix, MnSearchArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    if ((long long)blockIdx.x >= a.nq)
        return;
    beam_query<ORDER, NCH, BUILD, WIDE>(ix, a, blockIdx.x, threadIdx.x, smem, nullptr);
}

// Few queries (a single xFilter, a window of speculative inserts): one WORKGROUP per query, see CoopCtx (mn_beam.hpp).
#define MN_COOP_WAVES 8
template <int ORDER, int NCH, bool BUILD, bool WIDE = false>
__global__ void __launch_bounds__(MN_COOP_WAVES * 64) k_beam_coop(MnDevIndex ix, MnSearchArgs a, size_t base_lds) {
    extern __shared__ __align__(16) unsigned char smem[];
    if ((long long)blockIdx.x >= a.nq)
        return;
    const int lane = threadIdx.x & 63;
    CoopCtx c;
    c.n = reinterpret_cast<int *>(smem + base_lds);
    c.qnorm = reinterpret_cast<float *>(c.n + 1);
    c.list = c.n + 4;
    c.dist = reinterpret_cast<float *>(c.list + 64);
    c.nw = blockDim.x >> 6;
    c.wv = threadIdx.x >> 6;
    if (a.lat_tile_rows > 0) {
        c.tile = reinterpret_cast<float *>(smem + a.lat_tile_off) + (size_t)c.wv * mn_lat_tile_floats(ix.ld, a.lat_tile_rows);
        c.tile_rows = a.lat_tile_rows;
    }
    unsigned *lbm = nullptr;
    if (!BUILD && a.lds_bitmap) { // (uniform) one query's visited bitmap in LDS, cleared by the whole workgroup
        lbm = reinterpret_cast<unsigned *>(smem + base_lds) + 4 + 64 + 64;
        for (long long i = threadIdx.x; i < a.bm0_words; i += blockDim.x)
            lbm[i] = 0u;
        __syncthreads();
    }
    if (c.wv == 0) {
        beam_query<ORDER, NCH, BUILD, WIDE, true>(ix, a, blockIdx.x, lane, smem, &c, lbm);
        if (lane == 0)
            *c.n = -1;
        __syncthreads(); // releases the helpers
    } else {
        const float *q = reinterpret_cast<const float *>(smem + (size_t)(MN_CAND_LDS + MN_RES_LDS) * sizeof(uint2) + 64 * sizeof(int));
        coop_helper<ORDER, NCH>(ix, q, c, lane);
    }
}

size_t mn_search_lds_bytes(int ld, bool tile) {
    size_t b = (size_t)(MN_CAND_LDS + MN_RES_LDS) * sizeof(uint2) + 64 * sizeof(int) + (size_t)ld * sizeof(float);
    if (tile)
        b += MN_TILE_FLOATS * sizeof(float);
    // tuning knob: extra (unused) LDS per wavefront lowers the waves resident per CU
    if (const char *pad = getenv("MN_LDS_PAD_BYTES"))
        b += (size_t)atoi(pad);
    return b;
}

// Dynamic LDS beyond 64 KB has to be asked for per kernel (hipFuncAttributeMaxDynamicSharedMemorySize); what the device would
// grant is asked once.  Anything that fails leaves the 64 KB every kernel gets.
size_t mn_lds_optin_limit() {
    // per device ordinal (an index lives on its own device and any host thread may serve it), under a mutex
    static std::mutex mu;
    static std::unordered_map<int, size_t> lims;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    auto it = lims.find(dev);
    if (it != lims.end())
        return it->second;
    size_t lim = 64 * 1024;
    const char *e = getenv("MN_LDS_OPTIN"); // MN_LDS_OPTIN=0: never ask
    int v = 0;
    if (!(e && atoi(e) == 0) && hipDeviceGetAttribute(&v, hipDeviceAttributeSharedMemPerBlockOptin, dev) == hipSuccess && v > 64 * 1024)
        lim = (size_t)v;
    (void)hipGetLastError();
    lims[dev] = lim;
    return lim;
}
// one request per kernel and size (the call is not free, and these kernels are launched once per query / insert)
bool mn_lds_grant(const void *kern, size_t bytes) {
    if (bytes <= 64 * 1024)
        return true;
    static std::mutex mu;
    struct Seen {
        size_t granted = 0;           // largest size granted
        size_t refused = (size_t)-1;  // smallest size refused: only requests below it are tried again
    };
    static std::unordered_map<unsigned long long, Seen> seen; // (kernel, device)
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long key = (unsigned long long)reinterpret_cast<uintptr_t>(kern) * 64ull + (unsigned)(dev & 63);
    std::lock_guard<std::mutex> lk(mu);
    Seen &sn = seen[key];
    if (sn.granted >= bytes)
        return true;
    if (bytes >= sn.refused)
        return false;
    const bool ok = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
    if (ok) {
        sn.granted = bytes;
    } else {
        (void)hipGetLastError();
        sn.refused = bytes;
    }
    return ok;
}
template <typename K> static bool lds_grant(K kern, size_t bytes) { return mn_lds_grant(reinterpret_cast<const void *>(kern), bytes); }

template <int ORDER, int NCH, bool BUILD, bool WIDE>
static void launch_coop(const MnDevIndex &ix, MnSearchArgs a, size_t base, size_t tot, hipStream_t st) {
    // SSE order: an LDS tile per wavefront for the distance step (4 rows, else 2, else none — by what fits)
    const char *te = getenv("MN_LAT_TILE"); // MN_LAT_TILE=0: no distance tiles
    if (ORDER == MN_ORDER_SSE_V && ix.ld >= 256 && !(te && atoi(te) == 0)) { // (short rows: the plain walk is as fast)
        const size_t off = (tot + 15) & ~(size_t)15;
        for (int rows = 4; rows >= 2; rows >>= 1) {
            const size_t need = off + (size_t)MN_COOP_WAVES * mn_lat_tile_floats(ix.ld, rows) * sizeof(float);
            if (need <= mn_lds_optin_limit() && lds_grant(k_beam_coop<ORDER, NCH, BUILD, WIDE>, need)) {
                a.lat_tile_rows = rows;
                a.lat_tile_off = (unsigned)off;
                tot = need;
                if (getenv("MN_LAT_DEBUG"))
                    fprintf(stderr, "[mn] k_beam_coop: distance tile of %d rows per wavefront, %zu bytes of LDS\n", rows, need);
                break;
            }
        }
    }
    hipLaunchKernelGGL((k_beam_coop<ORDER, NCH, BUILD, WIDE>), dim3((unsigned)a.nq), dim3(MN_COOP_WAVES * 64), tot, st, ix, a, base);
}

template <int ORDER, int NCH>
static void launch_beam(const MnDevIndex &ix, const MnSearchArgs &a, bool build, hipStream_t st) {
    dim3 grid((unsigned)a.nq), block(64);
    size_t lds = mn_search_lds_bytes(ix.ld, ORDER == MN_ORDER_SSE_V && a.use_tile);
    const bool wide = ix.WX > 64; // rows of more than 64 links (M > 32, or lists grown by deletes) are walked in 64-link passes
    const char *co = getenv("MN_COOP"); // MN_COOP=0: always one wavefront per query
    if (a.nq <= 128 && !(co && atoi(co) == 0)) {
        const size_t base = (lds + 15) & ~(size_t)15;
        const size_t tot = base + (4 + 64 + 64) * sizeof(int) + (!build && a.lds_bitmap ? (size_t)a.bm0_words * sizeof(unsigned) : 0);
        if (wide) {
            if (build)
                launch_coop<ORDER, NCH, true, true>(ix, a, base, tot, st);
            else
                launch_coop<ORDER, NCH, false, true>(ix, a, base, tot, st);
        } else if (build)
            launch_coop<ORDER, NCH, true, false>(ix, a, base, tot, st);
        else
            launch_coop<ORDER, NCH, false, false>(ix, a, base, tot, st);
        return;
    }
    if (wide) {
        if (build)
            hipLaunchKernelGGL((k_beam<ORDER, NCH, true, true>), grid, block, lds, st, ix, a);
        else
            hipLaunchKernelGGL((k_beam<ORDER, NCH, false, true>), grid, block, lds, st, ix, a);
    } else if (build)
        hipLaunchKernelGGL((k_beam<ORDER, NCH, true>), grid, block, lds, st, ix, a);
    else
        hipLaunchKernelGGL((k_beam<ORDER, NCH, false>), grid, block, lds, st, ix, a);
}

int mn_launch_search(const MnDevIndex &ix, const MnSearchArgs &a, bool build, hipStream_t st) {
    if (a.nq <= 0)
        return 0;
    if (ix.order == MN_ORDER_SSE_V) {
        // (NCH is the wave order's chunk count; in the reference's order it selects the row walk: 1 = float4 loads with the
        //  in-quad transpose, for rows long enough to be bound by the memory system — mn_dist.hpp MN_SSE_QUAD2)
        const char *qe = getenv("MN_SSE_QUAD"); // MN_SSE_QUAD=0: always the dword walk (A/B runs)
        if (MN_SSE_QUAD2 > 0 && (ix.dim >> 4) >= MN_SSE_QUAD2 && !(qe && atoi(qe) == 0))
            launch_beam<MN_ORDER_SSE_V, 1>(ix, a, build, st);
        else
            launch_beam<MN_ORDER_SSE_V, 0>(ix, a, build, st);
        return 0;
    }
    switch (pick_nch(ix.ld)) {
    case 1: launch_beam<MN_ORDER_WAVE_V, 1>(ix, a, build, st); break;
    case 2: launch_beam<MN_ORDER_WAVE_V, 2>(ix, a, build, st); break;
    case 3: launch_beam<MN_ORDER_WAVE_V, 3>(ix, a, build, st); break;
    case 4: launch_beam<MN_ORDER_WAVE_V, 4>(ix, a, build, st); break;
    case 6: launch_beam<MN_ORDER_WAVE_V, 6>(ix, a, build, st); break;
    case 8: launch_beam<MN_ORDER_WAVE_V, 8>(ix, a, build, st); break;
    default: launch_beam<MN_ORDER_WAVE_V, 0>(ix, a, build, st); break;
    }
    return 0;
}

// ───────────────────────── k_edge_rows ─────────────────────────
// persist_node's per-edge distances (src/hnsw_vtab.c:268-279) for a list of (node, level) rows.

template <int ORDER, int NCH>
__global__ void __launch_bounds__(64)
    k_edge_rows(MnDevIndex ix, const int *row_slot, const int *row_level, int n_rows, int *out_nbr, float *out_dist) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    const int r = blockIdx.x;
    if (r >= n_rows)
        return;
    float *q = reinterpret_cast<float *>(smem);
    const int s = row_slot[r], level = row_level[r];
    const float *sv = ix.vectors + (size_t)s * ix.ld;
    for (int i = lane; i < ix.ld; i += 64)
        q[i] = sv[i];
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    int W;
    const int *row = link_row(ix, s, level, W);
    for (int c0 = 0; c0 < ix.WX; c0 += 64) { // 64 links per pass (rows are packed: no gaps before the -1 padding)
        const int p = c0 + lane;
        const int nb = p < W ? row[p] : -1;
        const int n = __popcll(__ballot(nb >= 0));
        float d = 0.0f;
        if (n > 0) {
            const int myslot = lane < n ? nb : 0;
            d = rows_distance<ORDER, NCH>(ix, q, ix.metric == 1 ? ix.norms[s] : 0.0f, myslot, n, lane);
            if (lane < n && ix.deleted[myslot])
                d = 0.0f;
        }
        if (p < ix.WX) {
            out_nbr[(size_t)r * ix.WX + p] = nb;
            out_dist[(size_t)r * ix.WX + p] = d;
        }
    }
}

void mn_launch_edge_rows(const MnDevIndex &ix, const int *d_row_slot, const int *d_row_level, int n_rows, int *d_out_nbr,
                         float *d_out_dist, hipStream_t st) {
    if (n_rows <= 0)
        return;
    size_t lds = (size_t)ix.ld * sizeof(float);
    dim3 grid(n_rows), block(64);
#define MN_ER(O, N) hipLaunchKernelGGL((k_edge_rows<O, N>), grid, block, lds, st, ix, d_row_slot, d_row_level, n_rows, d_out_nbr, d_out_dist)
    if (ix.order == MN_ORDER_SSE_V) {
        MN_ER(MN_ORDER_SSE_V, 0);
        return;
    }
    switch (pick_nch(ix.ld)) {
    case 1: MN_ER(MN_ORDER_WAVE_V, 1); break;
    case 2: MN_ER(MN_ORDER_WAVE_V, 2); break;
    case 3: MN_ER(MN_ORDER_WAVE_V, 3); break;
    case 4: MN_ER(MN_ORDER_WAVE_V, 4); break;
    case 6: MN_ER(MN_ORDER_WAVE_V, 6); break;
    case 8: MN_ER(MN_ORDER_WAVE_V, 8); break;
    default: MN_ER(MN_ORDER_WAVE_V, 0); break;
    }
#undef MN_ER
}

// ───────────────────────── k_merge_topk ─────────────────────────
// The exchange step of a sharded index: every shard returned its k nearest in ascending order; the global k nearest
// are the first k of the merge.  Equal distances: lower shard rank first, then position in the shard's list (the same
// total order as a stable sort of the shard-major concatenation).  One thread per query; world <= 64.
__global__ void k_merge_topk(const long long *g_ids, const float *g_dists, const int *g_counts, int world, long long nq, int k,
                             long long *out_ids, float *out_dists, int *out_counts) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq)
        return;
    int pos[64];
    for (int w = 0; w < world; w++)
        pos[w] = 0;
    int n = 0;
    for (; n < k; n++) {
        int bw = -1;
        float bd = 0.0f;
        for (int w = 0; w < world; w++) {
            if (pos[w] >= g_counts[(size_t)w * nq + q])
                continue;
            const float d = g_dists[((size_t)w * nq + q) * k + pos[w]];
            if (bw < 0 || d < bd) { // strict: ties keep the lower shard
                bw = w;
                bd = d;
            }
        }
        if (bw < 0)
            break;
        out_ids[q * k + n] = g_ids[((size_t)bw * nq + q) * k + pos[bw]];
        out_dists[q * k + n] = bd;
        pos[bw]++;
    }
    out_counts[q] = n;
    for (int i = n; i < k; i++) {
        out_ids[q * k + i] = -1;
        out_dists[q * k + i] = 0.0f;
    }
}

void mn_launch_merge_topk(const long long *g_ids, const float *g_dists, const int *g_counts, int world, long long nq, int k,
                          long long *out_ids, float *out_dists, int *out_counts, hipStream_t st) {
    if (nq <= 0)
        return;
    hipLaunchKernelGGL(k_merge_topk, dim3((unsigned)((nq + 127) / 128)), dim3(128), 0, st, g_ids, g_dists, g_counts, world, nq, k,
                       out_ids, out_dists, out_counts);
}

#ifdef MN_PHASE_TIMING
extern "C" int mn_debug_phase_kernels(unsigned long long *out, int reset) { // probe builds only (scripts/probe_phases.sh)
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mn_phase), 8 * sizeof(unsigned long long)) != hipSuccess)
        return -1;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(mn_phase), z, sizeof(z)) != hipSuccess)
            return -1;
    }
    return 0;
}
#endif

// HIP loads a translation unit's code object on the first use of one of its kernels (several milliseconds for these units): an
// index asks for all of them when it is created (mn_index.hip), so that the first query or insert of a process does not pay.
void mn_module_touch_kernels() {
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&k_merge_topk));
}
