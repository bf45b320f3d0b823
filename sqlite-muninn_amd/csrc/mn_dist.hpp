// mn_dist.hpp — device-side distance inner loops shared by the search and link kernels (gfx950).
// See mn_kernels.hip for the execution model.  Build with -ffp-contract=off.
#pragma once
#include "mn_device.hpp"

#define DEVI __device__ __forceinline__

DEVI int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
DEVI unsigned rflu(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
DEVI float u2f(unsigned u) { return __uint_as_float(u); }
DEVI unsigned f2u(float f) { return __float_as_uint(f); }

// ───────────────────────── distance inner loops ─────────────────────────

DEVI float cosine_finish(float dot, float na, float nb) {
    // src/vec_math.c:122-125.  sqrtf and / are IEEE-correct here (hipcc default
    // -fhip-fp32-correctly-rounded-divide-sqrt); __fsqrt_rn would be the approximate native sqrt.
    float denom = __fmul_rn(sqrtf(na), sqrtf(nb));
    if (denom < 1e-30f)
        return 1.0f;
    return __fsub_rn(1.0f, __fdiv_rn(dot, denom));
}

DEVI float wave_butterfly(float acc) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
        acc = __fadd_rn(acc, __shfl_xor(acc, m));
    return acc;
}

// WAVE order, one row against a vector held in LDS.  Lane L owns elements 256k + 4L + j.
template <bool L2>
DEVI float wave_row_generic(const float *__restrict__ row, const float *q_lds, int ld, int lane) {
    float acc = 0.0f;
    for (int e = lane * 4; e < ld; e += 256) {
        float4 v = *reinterpret_cast<const float4 *>(row + e);
        float4 q = *reinterpret_cast<const float4 *>(q_lds + e);
        if (L2) {
            float d0 = __fsub_rn(q.x, v.x), d1 = __fsub_rn(q.y, v.y), d2 = __fsub_rn(q.z, v.z), d3 = __fsub_rn(q.w, v.w);
            acc = fmaf(d0, d0, acc);
            acc = fmaf(d1, d1, acc);
            acc = fmaf(d2, d2, acc);
            acc = fmaf(d3, d3, acc);
        } else {
            acc = fmaf(q.x, v.x, acc);
            acc = fmaf(q.y, v.y, acc);
            acc = fmaf(q.z, v.z, acc);
            acc = fmaf(q.w, v.w, acc);
        }
    }
    return wave_butterfly(acc);
}

// SSE order: the 4 lanes (lane&3 = j) of a row group walk accumulator j; every lane of the group
// returns the finished sum.  `row` may point to global or LDS memory.
template <bool L2>
DEVI float sse_row(const float *__restrict__ row, const float *q_lds, int dim, int lane) {
    const int j = lane & 3;
    const int steps = dim >> 2;
    float s = 0.0f;
#pragma unroll 8
    for (int c = 0; c < steps; c++) {
        float b = row[4 * c + j];
        float a = q_lds[4 * c + j];
        float p;
        if (L2) {
            float d = __fsub_rn(a, b);
            p = __fmul_rn(d, d);
        } else {
            p = __fmul_rn(a, b);
        }
        s = __fadd_rn(s, p);
    }
    const int g = lane & ~3;
    float s0 = __shfl(s, g), s1 = __shfl(s, g + 1), s2 = __shfl(s, g + 2), s3 = __shfl(s, g + 3);
    float sum = __fadd_rn(__fadd_rn(__fadd_rn(s0, s1), s2), s3);
    for (int i = steps * 4; i < dim; i++) {
        float a = q_lds[i], b = row[i], p;
        if (L2) {
            float d = __fsub_rn(a, b);
            p = __fmul_rn(d, d);
        } else {
            p = __fmul_rn(a, b);
        }
        sum = __fadd_rn(sum, p);
    }
    return sum;
}

// Raw accumulation (dot or Σd²) of `n` rows (slot per lane, lanes < n valid) against q_lds.
// Returns, in lane i < n, the value for row i.
template <int ORDER, int NCH, bool L2>
DEVI float rows_accumulate(const MnDevIndex &ix, const float *q_lds, int myslot, int n, int lane) {
    float mine = 0.0f;
    if (ORDER == MN_ORDER_SSE_V) {
        for (int t = 0; t < n; t += 16) {
            int r = t + (lane >> 2);
            int s = __shfl(myslot, r < n ? r : n - 1);
            float v = sse_row<L2>(ix.vectors + (size_t)s * ix.ld, q_lds, ix.dim, lane);
            // lane i in [t, t+16) fetches the sum from group (i - t)
            float got = __shfl(v, ((lane - t) & 15) << 2);
            if (lane >= t && lane < t + 16)
                mine = got;
        }
        return mine;
    }
    if (NCH == 0) { // generic: any ld, one row in flight
        for (int t = 0; t < n; t++) {
            int s = __builtin_amdgcn_readlane(myslot, t);
            float v = wave_row_generic<L2>(ix.vectors + (size_t)s * ix.ld, q_lds, ix.ld, lane);
            if (lane == t)
                mine = v;
        }
        return mine;
    }
    constexpr int NC = NCH > 0 ? NCH : 1;
    constexpr int R = (NC <= 3) ? 8 : ((NC <= 6) ? 4 : 2);
    for (int t = 0; t < n; t += R) {
        float4 v[R][NC];
#pragma unroll
        for (int r = 0; r < R; r++) {
            int src = t + r < n ? t + r : n - 1;
            int s = __builtin_amdgcn_readlane(myslot, src);
            const float4 *row = reinterpret_cast<const float4 *>(ix.vectors + (size_t)s * ix.ld);
#pragma unroll
            for (int k = 0; k < NC; k++) {
                int e4 = k * 64 + lane;
                v[r][k] = (e4 * 4 < ix.ld) ? row[e4] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < NC; k++) {
                int e4 = k * 64 + lane;
                if (e4 * 4 < ix.ld) {
                    float4 q = reinterpret_cast<const float4 *>(q_lds)[e4];
                    float4 b = v[r][k];
                    if (L2) {
                        float d0 = __fsub_rn(q.x, b.x), d1 = __fsub_rn(q.y, b.y), d2 = __fsub_rn(q.z, b.z),
                              d3 = __fsub_rn(q.w, b.w);
                        acc = fmaf(d0, d0, acc);
                        acc = fmaf(d1, d1, acc);
                        acc = fmaf(d2, d2, acc);
                        acc = fmaf(d3, d3, acc);
                    } else {
                        acc = fmaf(q.x, b.x, acc);
                        acc = fmaf(q.y, b.y, acc);
                        acc = fmaf(q.z, b.z, acc);
                        acc = fmaf(q.w, b.w, acc);
                    }
                }
            }
            acc = wave_butterfly(acc);
            if (lane == t + r)
                mine = acc;
        }
    }
    return mine;
}

// distances of rows myslot[0..n) to the query in q_lds (qnorm = |q|² for cosine)
template <int ORDER, int NCH>
DEVI float rows_distance(const MnDevIndex &ix, const float *q_lds, float qnorm, int myslot, int n, int lane) {
    if (ix.metric == 0) {
        return rows_accumulate<ORDER, NCH, true>(ix, q_lds, myslot, n, lane);
    }
    float dot = rows_accumulate<ORDER, NCH, false>(ix, q_lds, myslot, n, lane);
    if (ix.metric == 2)
        return -dot; // src/vec_math.c:142
    float nb = (lane < n) ? ix.norms[myslot] : 1.0f;
    return cosine_finish(dot, qnorm, nb);
}

// |v|² of the vector in LDS, in the index's order
template <int ORDER>
DEVI float lds_self_norm(const float *q_lds, int dim, int ld, int lane) {
    if (ORDER == MN_ORDER_SSE_V) {
        float v = sse_row<false>(q_lds, q_lds, dim, lane);
        return __shfl(v, 0);
    }
    return wave_row_generic<false>(q_lds, q_lds, ld, lane);
}

