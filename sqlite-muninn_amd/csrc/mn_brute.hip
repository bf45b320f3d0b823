// mn_brute.hip — exact brute-force top-k over the device-resident index (gfx950).
// Measurement support only: produces the ground truth recall@k is computed against
// (benchmarks/harness/treatments/vss.py:96-102 in the reference does the same on the CPU).
// Distances use the index's own inner loop, so "exact" means exact in that arithmetic.
// One 256-thread workgroup per query: 4 wavefronts stride over the rows (64 rows per step each,
// coalesced float4 loads), each keeping a sorted top-k in LDS; wave 0 merges.
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
