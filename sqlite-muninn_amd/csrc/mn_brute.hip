// mn_brute.hip — exact brute-force top-k over the device-resident index (gfx950).
// Measurement support only: produces the ground truth recall@k is computed against
// (benchmarks/harness/treatments/vss.py:96-102 in the reference does the same on the CPU).
//
//   k_brute_mfma   the batched query x row block as a GEMM on the f32 matrix cores (v_mfma_f32_32x32x2_f32: exact f32,
//                  a k-ordered fmaf chain) with the top-k selection fused into the epilogue — the one place of this
//                  path where a dense query x candidate block appears (north_star).  k <= 16.
//   k_bruteforce   VALU kernel using the index's own inner loop (any k <= 128); one 256-thread workgroup per query:
//                  4 wavefronts stride over the rows, each keeping a sorted top-k in LDS; wave 0 merges.
#include "mn_dist.hpp"

#define BF_KMAX 128

template <int ORDER, int NCH>
__global__ void __launch_bounds__(256) k_bruteforce(MnDevIndex ix, const float *queries, long long nq, int k,
                                                    long long *out_ids) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long long qi = blockIdx.x;
    float *q = reinterpret_cast<float *>(smem);                 // [ld]
    float *topd = q + ix.ld;                                    // [4][BF_KMAX]
    int *tops = reinterpret_cast<int *>(topd + 4 * BF_KMAX);    // [4][BF_KMAX]
    const float *qsrc = queries + (size_t)qi * ix.dim;
    for (int i = tid; i < ix.ld; i += 256)
        q[i] = i < ix.dim ? qsrc[i] : 0.0f;
    __syncthreads();
    float qnorm = 0.0f;
    if (ix.metric == 1)
        qnorm = lds_self_norm<ORDER>(q, ix.dim, ix.ld, lane);
    float *myd = topd + wv * BF_KMAX;
    int *mys = tops + wv * BF_KMAX;
    int cnt = 0; // wave-uniform
    for (int base = wv * 64; base < ix.n_slots; base += 256) {
        int n = ix.n_slots - base < 64 ? ix.n_slots - base : 64;
        MnDevIndex sub = ix;
        sub.vectors = ix.vectors + (size_t)base * ix.ld;
        sub.norms = ix.norms ? ix.norms + base : nullptr;
        int myslot = lane < n ? lane : 0;
        float d = rows_distance<ORDER, NCH>(sub, q, qnorm, myslot, n, lane);
        bool ok = lane < n && !ix.deleted[base + myslot];
        float worst = cnt >= k ? myd[k - 1] : 3.0e38f;
        unsigned long long m = __ballot(ok && (cnt < k || d < worst));
        while (m) {
            int i = __ffsll((long long)m) - 1;
            m &= m - 1;
            float di = __shfl(d, i);
            int si = base + i;
            if (cnt >= k && !(di < myd[k - 1]))
                continue;
            // insertion into the sorted list (ascending distance, then ascending slot)
            int pos = cnt < k ? cnt : k - 1;
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
                while (pos > 0 && (myd[pos - 1] > di)) {
                    myd[pos] = myd[pos - 1];
                    mys[pos] = mys[pos - 1];
                    pos--;
                }
                myd[pos] = di;
                mys[pos] = si;
            }
            __builtin_amdgcn_wave_barrier();
            if (cnt < k)
                cnt++;
        }
    }
    // publish counts through LDS: reuse slot BF_KMAX-1? keep separate
    __shared__ int wcnt[4];
    if (lane == 0)
        wcnt[wv] = cnt;
    __syncthreads();
    if (tid == 0) {
        int p[4] = {0, 0, 0, 0};
        for (int o = 0; o < k; o++) {
            int bw = -1;
            for (int w2 = 0; w2 < 4; w2++) {
                if (p[w2] >= wcnt[w2])
                    continue;
                float dd = topd[w2 * BF_KMAX + p[w2]];
                int ss = tops[w2 * BF_KMAX + p[w2]];
                if (bw < 0 || dd < topd[bw * BF_KMAX + p[bw]] ||
                    (dd == topd[bw * BF_KMAX + p[bw]] && ss < tops[bw * BF_KMAX + p[bw]]))
                    bw = w2;
            }
            if (bw < 0) {
                out_ids[qi * k + o] = -1;
            } else {
                out_ids[qi * k + o] = ix.ids[tops[bw * BF_KMAX + p[bw]]];
                p[bw]++;
            }
        }
    }
}

static int pick_nch_bf(int ld) {
    int need = (ld + 255) / 256;
    if (need <= 1) return 1;
    if (need <= 2) return 2;
    if (need <= 3) return 3;
    if (need <= 4) return 4;
    if (need <= 6) return 6;
    if (need <= 8) return 8;
    return 0;
}

void mn_launch_bruteforce(const MnDevIndex &ix, const float *d_queries, long long nq, int k, long long *d_out_ids,
                          float *, hipStream_t st) {
    if (nq <= 0 || k <= 0 || k > BF_KMAX)
        return;
    size_t lds = (size_t)ix.ld * sizeof(float) + 4 * BF_KMAX * (sizeof(float) + sizeof(int));
    dim3 grid((unsigned)nq), block(256);
#define MN_BF(O, N) hipLaunchKernelGGL((k_bruteforce<O, N>), grid, block, lds, st, ix, d_queries, nq, k, d_out_ids)
    if (ix.order == MN_ORDER_SSE_V) {
        MN_BF(MN_ORDER_SSE_V, 0);
        return;
    }
    switch (pick_nch_bf(ix.ld)) {
    case 1: MN_BF(MN_ORDER_WAVE_V, 1); break;
    case 2: MN_BF(MN_ORDER_WAVE_V, 2); break;
    case 3: MN_BF(MN_ORDER_WAVE_V, 3); break;
    case 4: MN_BF(MN_ORDER_WAVE_V, 4); break;
    case 6: MN_BF(MN_ORDER_WAVE_V, 6); break;
    case 8: MN_BF(MN_ORDER_WAVE_V, 8); break;
    default: MN_BF(MN_ORDER_WAVE_V, 0); break;
    }
#undef MN_BF
}


// ───────────────────────── k_brute_mfma ─────────────────────────
// S = Q · Xᵀ tile by tile: a workgroup (4 wavefronts) owns 128 queries and walks a chunk of the rows 128 at a time;
// wavefront w owns queries 32w..32w+31 against all 128 rows of the tile (four 32x32 accumulators, 1 A read + 4 B reads
// from LDS per four MFMAs).  Operands are staged k-major in LDS ([k][129]: conflict-free fragment reads), 32 k per
// stage, the next stage's global loads in flight while the current one is multiplied.  Epilogue: dot → distance
// (cached |x|², |q|²), compare with the query's current k-th best (LDS), survivors — rare after the first tiles — are
// inserted into the query's sorted list by the wavefront that owns it (no cross-wave traffic).  Each (query tile, row
// chunk) writes one partial list per query; k_brute_merge folds the chunks in row order (ties: lower row first).
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BM_Q 128
#define BM_R 128
#define BM_KC 32
#define BM_LD 129
#define BM_KMAX 16

struct MnBruteArgs {
    const float *q;  // [nq_pad][ld] zero padded (nq_pad multiple of 128)
    const float *qn; // [nq_pad] |q|²
    const float *xn; // [n_slots] |x|² (cosine, l2) or null
    long long nq;
    int k, rows_per_chunk, n_chunks;
    float *pd; // [n_chunks][nq][k]
    int *pi;   // [n_chunks][nq][k]
    int *pc;   // [n_chunks][nq]
};

__global__ void k_rows_sqnorm(const float *rows, long long n, int ld, float *out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const float *r = rows + (size_t)i * ld;
    float acc = 0.0f;
    for (int e = 0; e < ld; e++)
        acc = fmaf(r[e], r[e], acc);
    out[i] = acc;
}

template <int METRIC>
__global__ void __launch_bounds__(256) k_brute_mfma(MnDevIndex ix, MnBruteArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    float *As = reinterpret_cast<float *>(smem);           // [BM_KC][BM_LD]
    float *Bs = As + BM_KC * BM_LD;                        // [BM_KC][BM_LD]
    float *qn_s = Bs + BM_KC * BM_LD;                      // [128]
    float *thr = qn_s + BM_Q;                              // [128] current k-th best (3.4e38 until the list is full)
    int *cnt = reinterpret_cast<int *>(thr + BM_Q);        // [128]
    float *ld_ = reinterpret_cast<float *>(cnt + BM_Q);    // [128][k]
    int *li_ = reinterpret_cast<int *>(ld_ + BM_Q * a.k);  // [128][k]

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int K = a.k;
    const long long q0 = (long long)blockIdx.x * BM_Q;
    const int chunk = blockIdx.y;
    const int r_begin = chunk * a.rows_per_chunk;
    const int r_end = r_begin + a.rows_per_chunk < ix.n_slots ? r_begin + a.rows_per_chunk : ix.n_slots;
    if (tid < BM_Q) {
        qn_s[tid] = a.qn[q0 + tid];
        thr[tid] = 3.4e38f;
        cnt[tid] = 0;
    }
    __syncthreads();
    const int ld = ix.ld;
    const int nk = (ld + BM_KC - 1) / BM_KC;
    const int srow = tid >> 3, skq = tid & 7; // staging: 8 threads x float4 = 128 contiguous bytes of one row
    for (int rt = r_begin; rt < r_end; rt += BM_R) {
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int i = 0; i < 16; i++)
                acc[t][i] = 0.0f;
        float4 pa[4], pb[4];
        auto prefetch = [&](int kc) {
            const int kcol = kc * BM_KC + 4 * skq;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int row = srow + 32 * j;
                const bool kin = kcol < ld;
                pa[j] = kin ? *reinterpret_cast<const float4 *>(a.q + (size_t)(q0 + row) * ld + kcol) : make_float4(0, 0, 0, 0);
                pb[j] = (kin && rt + row < ix.n_slots)
                            ? *reinterpret_cast<const float4 *>(ix.vectors + (size_t)(rt + row) * ld + kcol)
                            : make_float4(0, 0, 0, 0);
            }
        };
        prefetch(0);
        for (int kc = 0; kc < nk; kc++) {
            __syncthreads(); // the previous stage has been consumed
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int row = srow + 32 * j;
                float *ap = As + (4 * skq) * BM_LD + row, *bp = Bs + (4 * skq) * BM_LD + row;
                ap[0] = pa[j].x; ap[BM_LD] = pa[j].y; ap[2 * BM_LD] = pa[j].z; ap[3 * BM_LD] = pa[j].w;
                bp[0] = pb[j].x; bp[BM_LD] = pb[j].y; bp[2 * BM_LD] = pb[j].z; bp[3 * BM_LD] = pb[j].w;
            }
            __syncthreads();
            if (kc + 1 < nk)
                prefetch(kc + 1);
            const float *ar = As + (lane >> 5) * BM_LD + 32 * w + (lane & 31);
            const float *br = Bs + (lane >> 5) * BM_LD + (lane & 31);
#pragma unroll
            for (int kk = 0; kk < BM_KC / 2; kk++) {
                const float av = ar[2 * kk * BM_LD];
#pragma unroll
                for (int t = 0; t < 4; t++)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, br[2 * kk * BM_LD + 32 * t], acc[t], 0, 0, 0);
            }
        }
        // ── epilogue: distances, threshold filter, insertion (this wavefront's 32 queries only) ──
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int row = rt + 32 * t + (lane & 31);
            const bool rv = row < r_end && !ix.deleted[row < ix.n_slots ? row : 0];
            const float xn = (a.xn && row < ix.n_slots) ? a.xn[row] : 0.0f;
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                const int ql = 32 * w + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); // C/D map: row of the tile = query
                const float dot = acc[t][reg];
                float d;
                if (METRIC == 1)
                    d = cosine_finish(dot, qn_s[ql], xn);
                else if (METRIC == 0)
                    d = __fadd_rn(__fsub_rn(qn_s[ql], __fmul_rn(2.0f, dot)), xn);
                else
                    d = __fsub_rn(0.0f, dot);
                unsigned long long m = __ballot(rv && q0 + ql < a.nq && d < thr[ql]);
                while (m) {
                    const int b = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const float nd = __shfl(d, b);
                    const int nrow = rt + 32 * t + (b & 31);
                    const int q = 32 * w + (reg & 3) + 8 * (reg >> 2) + 4 * (b >> 5);
                    if (!(nd < thr[q]))
                        continue;
                    const int c = cnt[q];
                    float cd = 3.4e38f;
                    int ci = -1;
                    if (lane < c) {
                        cd = ld_[q * K + lane];
                        ci = li_[q * K + lane];
                    }
                    const int pos = __popcll(__ballot(lane < c && cd <= nd)); // equal distances: the earlier row stays first
                    __builtin_amdgcn_wave_barrier();
                    if (lane >= pos && lane < c && lane + 1 < K) {
                        ld_[q * K + lane + 1] = cd;
                        li_[q * K + lane + 1] = ci;
                    }
                    if (lane == 0) {
                        ld_[q * K + pos] = nd;
                        li_[q * K + pos] = nrow;
                        cnt[q] = c < K ? c + 1 : K;
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (c + 1 >= K && lane == 0)
                        thr[q] = ld_[q * K + K - 1];
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    for (int i = 0; i < 32; i++) { // partial lists of this wavefront's queries
        const int ql = 32 * w + i;
        if (q0 + ql >= a.nq)
            break;
        const size_t o = ((size_t)chunk * a.nq + (q0 + ql));
        if (lane < K) {
            a.pd[o * K + lane] = ld_[ql * K + lane];
            a.pi[o * K + lane] = li_[ql * K + lane];
        }
        if (lane == 0)
            a.pc[o] = cnt[ql];
    }
}

__global__ void k_brute_merge(MnDevIndex ix, MnBruteArgs a, long long *out_ids) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= a.nq)
        return;
    const int K = a.k;
    float best_d[BM_KMAX];
    int best_i[BM_KMAX];
    int n = 0;
    for (int c = 0; c < a.n_chunks; c++) { // chunks in row order; a later equal distance never displaces an earlier one
        const size_t o = (size_t)c * a.nq + q;
        const int pc = a.pc[o];
        for (int j = 0; j < pc; j++) {
            const float d = a.pd[o * K + j];
            if (n == K && !(d < best_d[K - 1]))
                break; // the partial list is ascending
            int pos = n < K ? n : K - 1;
            while (pos > 0 && best_d[pos - 1] > d) {
                best_d[pos] = best_d[pos - 1];
                best_i[pos] = best_i[pos - 1];
                pos--;
            }
            best_d[pos] = d;
            best_i[pos] = a.pi[o * K + j];
            if (n < K)
                n++;
        }
    }
    for (int j = 0; j < K; j++)
        out_ids[q * K + j] = j < n ? ix.ids[best_i[j]] : -1;
}

size_t mn_brute_mfma_scratch_bytes(const MnDevIndex &ix, long long nq, int k, int *n_chunks_out, int *rows_per_chunk_out) {
    const long long qt = (nq + BM_Q - 1) / BM_Q;
    long long want = (1024 + qt - 1) / qt; // >= ~4 workgroups per CU
    const long long max_chunks = (ix.n_slots + 8 * BM_R - 1) / (8 * BM_R); // >= 8 tiles per chunk: the first tiles pay for list warm-up
    if (want > max_chunks)
        want = max_chunks;
    if (want < 1)
        want = 1;
    int rpc = (int)((ix.n_slots + want - 1) / want);
    rpc = (rpc + BM_R - 1) / BM_R * BM_R;
    const int nc = (ix.n_slots + rpc - 1) / rpc;
    *n_chunks_out = nc;
    *rows_per_chunk_out = rpc;
    const size_t nq_pad = (size_t)qt * BM_Q;
    return nq_pad * ix.ld * 4 + nq_pad * 4 + (size_t)ix.n_slots * 4 + (size_t)nc * nq * k * 8 + (size_t)nc * nq * 4 + 1024;
}

// scratch: one device allocation of mn_brute_mfma_scratch_bytes(); d_queries dense [nq][dim]
int mn_launch_bruteforce_mfma(const MnDevIndex &ix, const float *d_queries, long long nq, int k, long long *d_out_ids,
                              void *scratch, hipStream_t st) {
    if (nq <= 0 || k <= 0 || k > BM_KMAX || ix.n_slots <= 0)
        return -1;
    MnBruteArgs a;
    size_t bytes = mn_brute_mfma_scratch_bytes(ix, nq, k, &a.n_chunks, &a.rows_per_chunk);
    (void)bytes;
    const long long qt = (nq + BM_Q - 1) / BM_Q;
    const size_t nq_pad = (size_t)qt * BM_Q;
    unsigned char *p = static_cast<unsigned char *>(scratch);
    float *qpad = reinterpret_cast<float *>(p);
    p += nq_pad * ix.ld * 4;
    float *qn = reinterpret_cast<float *>(p);
    p += nq_pad * 4;
    float *xn = reinterpret_cast<float *>(p);
    p += (size_t)ix.n_slots * 4;
    a.pd = reinterpret_cast<float *>(p);
    p += (size_t)a.n_chunks * nq * k * 4;
    a.pi = reinterpret_cast<int *>(p);
    p += (size_t)a.n_chunks * nq * k * 4;
    a.pc = reinterpret_cast<int *>(p);
    if (hipMemsetAsync(qpad, 0, nq_pad * ix.ld * 4, st) != hipSuccess)
        return -1;
    if (hipMemcpy2DAsync(qpad, (size_t)ix.ld * 4, d_queries, (size_t)ix.dim * 4, (size_t)ix.dim * 4, (size_t)nq,
                         hipMemcpyDeviceToDevice, st) != hipSuccess)
        return -1;
    hipLaunchKernelGGL(k_rows_sqnorm, dim3((unsigned)((nq_pad + 255) / 256)), dim3(256), 0, st, qpad, (long long)nq_pad, ix.ld, qn);
    a.xn = nullptr;
    if (ix.metric == 1) {
        a.xn = ix.norms;
    } else if (ix.metric == 0) {
        hipLaunchKernelGGL(k_rows_sqnorm, dim3((unsigned)((ix.n_slots + 255) / 256)), dim3(256), 0, st, ix.vectors,
                           (long long)ix.n_slots, ix.ld, xn);
        a.xn = xn;
    }
    a.q = qpad;
    a.qn = qn;
    a.nq = nq;
    a.k = k;
    const size_t lds = (size_t)(2 * BM_KC * BM_LD + 2 * BM_Q) * 4 + BM_Q * 4 + (size_t)BM_Q * k * 8;
    const dim3 grid((unsigned)qt, (unsigned)a.n_chunks);
    if (ix.metric == 1)
        hipLaunchKernelGGL(k_brute_mfma<1>, grid, dim3(256), lds, st, ix, a);
    else if (ix.metric == 0)
        hipLaunchKernelGGL(k_brute_mfma<0>, grid, dim3(256), lds, st, ix, a);
    else
        hipLaunchKernelGGL(k_brute_mfma<2>, grid, dim3(256), lds, st, ix, a);
    hipLaunchKernelGGL(k_brute_merge, dim3((unsigned)((nq + 127) / 128)), dim3(128), 0, st, ix, a, d_out_ids);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
