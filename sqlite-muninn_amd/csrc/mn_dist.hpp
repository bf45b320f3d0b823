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
DEVI float quad_xor1(float v) { // lanes (0,1) and (2,3) of each quad exchange
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
}
DEVI float quad_xor2(float v) { // lanes (0,2) and (1,3) of each quad exchange
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));
}

#ifndef MN_SSE_UNROLL
#define MN_SSE_UNROLL 16
#endif
// chain positions c .. in whole batches of U: all U loads first (the scheduler may not move anything across the barrier — left
// to itself about half of the inlined copies of this loop came out as load-use-load-use), then the sums in order
template <bool L2, int U>
DEVI void sse_chunks(const float *__restrict__ row, const float *q_lds, int j, int steps, int &c, float &s) {
    for (; c + U <= steps; c += U) {
        float b[U];
#pragma unroll
        for (int u = 0; u < U; u++)
            b[u] = row[4 * (c + u) + j];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const float a = q_lds[4 * (c + u) + j];
            float p;
            if (L2) {
                float d = __fsub_rn(a, b[u]);
                p = __fmul_rn(d, d);
            } else {
                p = __fmul_rn(a, b[u]);
            }
            s = __fadd_rn(s, p);
        }
    }
}

#ifndef MN_SSE_QUAD2
#define MN_SSE_QUAD2 12 // float4 loads in flight per lane (0: the dword walk).  Same box, 1M x 768, ef 128, 10k queries per launch,
                        // three interleaved trials of each library (scripts/r04_call6.sh): 8 in flight 27.46 / 27.53 / 27.42 ms,
                        // 12 in flight 25.80 / 25.82 / 25.88 ms, 16 in flight 27.41 / 27.64 ms; the dword walk 27.8 - 29.9 ms
#endif
// MN_SSE_QUAD2 (round 4): the SSE-order walk with 16-byte loads.  Quad lane q of a row fetches the float4 of chain position
// 4t + q — the quad reads 64 contiguous bytes of the row with ONE load instruction where the dword walk needs four, and a
// quarter of the cache-line look-ups per byte —, forms its four products, and a 4 x 4 transpose inside the quad (two DPP
// exchange rounds) hands accumulator j its terms of positions 4t .. 4t+3, which it adds in that order: the reference's
// operations in the reference's order, same bits.  UB float4 loads are in flight per lane before the first use.
template <bool L2, int UB>
DEVI void sse_blocks_quad(const float *__restrict__ row, const float *q_lds, int j, int blocks, int &t, float &s) {
    const bool b0 = j & 1, b1 = j & 2;
    for (; t + UB <= blocks; t += UB) {
        float4 bv[UB];
#pragma unroll
        for (int u = 0; u < UB; u++)
            bv[u] = *reinterpret_cast<const float4 *>(row + 16 * (t + u) + 4 * j);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UB; u++) {
            const float4 a = *reinterpret_cast<const float4 *>(q_lds + 16 * (t + u) + 4 * j);
            const float4 b = bv[u];
            float p0, p1, p2, p3;
            if (L2) {
                const float d0 = __fsub_rn(a.x, b.x), d1 = __fsub_rn(a.y, b.y), d2 = __fsub_rn(a.z, b.z), d3 = __fsub_rn(a.w, b.w);
                p0 = __fmul_rn(d0, d0);
                p1 = __fmul_rn(d1, d1);
                p2 = __fmul_rn(d2, d2);
                p3 = __fmul_rn(d3, d3);
            } else {
                p0 = __fmul_rn(a.x, b.x);
                p1 = __fmul_rn(a.y, b.y);
                p2 = __fmul_rn(a.z, b.z);
                p3 = __fmul_rn(a.w, b.w);
            }
            // lane q holds the products of position 4t+q for accumulators 0..3; accumulator j wants positions 4t..4t+3 of j
            float r0 = quad_xor1(b0 ? p0 : p1), r1 = quad_xor1(b0 ? p2 : p3); // across lane bit 0: pairs (0,1), (2,3)
            if (b0) {
                p0 = r0;
                p2 = r1;
            } else {
                p1 = r0;
                p3 = r1;
            }
            r0 = quad_xor2(b1 ? p0 : p2); // across lane bit 1: pairs (0,2), (1,3)
            r1 = quad_xor2(b1 ? p1 : p3);
            if (b1) {
                p0 = r0;
                p1 = r1;
            } else {
                p2 = r0;
                p3 = r1;
            }
            s = __fadd_rn(s, p0);
            s = __fadd_rn(s, p1);
            s = __fadd_rn(s, p2);
            s = __fadd_rn(s, p3);
        }
    }
}

template <bool L2, bool LAT = false, bool QUAD = false>
DEVI float sse_row(const float *__restrict__ row, const float *q_lds, int dim, int lane) {
    const int j = lane & 3;
    const int steps = dim >> 2; // chain positions (one per group of 4 elements)
    float s = 0.0f;
    int c = 0;
#ifdef MN_SSE_QUAD // measured slower on gfx950 (VALU-bound transpose: 38.1 vs 31.5 ms per 10k queries); kept for reference
    // Blocks of 4 chain positions: quad lane q fetches the float4 of position 4t+q (the quad reads 64
    // contiguous bytes of the row), forms its four products, and a 4x4 transpose inside the quad (two
    // DPP exchange rounds) hands accumulator j its terms of positions 4t..4t+3, added in that order.
    const bool b0 = j & 1, b1 = j & 2;
    const int blocks = steps >> 2;
#pragma unroll 4
    for (int t = 0; t < blocks; t++) {
        const float4 b = *reinterpret_cast<const float4 *>(row + 16 * t + 4 * j);
        const float4 a = *reinterpret_cast<const float4 *>(q_lds + 16 * t + 4 * j);
        float p0, p1, p2, p3;
        if (L2) {
            float d0 = __fsub_rn(a.x, b.x), d1 = __fsub_rn(a.y, b.y), d2 = __fsub_rn(a.z, b.z), d3 = __fsub_rn(a.w, b.w);
            p0 = __fmul_rn(d0, d0);
            p1 = __fmul_rn(d1, d1);
            p2 = __fmul_rn(d2, d2);
            p3 = __fmul_rn(d3, d3);
        } else {
            p0 = __fmul_rn(a.x, b.x);
            p1 = __fmul_rn(a.y, b.y);
            p2 = __fmul_rn(a.z, b.z);
            p3 = __fmul_rn(a.w, b.w);
        }
        // round 1: exchange across lane bit 0 — register pairs (0,1) and (2,3)
        float r0 = quad_xor1(b0 ? p0 : p1), r1 = quad_xor1(b0 ? p2 : p3);
        if (b0) {
            p0 = r0;
            p2 = r1;
        } else {
            p1 = r0;
            p3 = r1;
        }
        // round 2: exchange across lane bit 1 — register pairs (0,2) and (1,3)
        r0 = quad_xor2(b1 ? p0 : p2);
        r1 = quad_xor2(b1 ? p1 : p3);
        if (b1) {
            p0 = r0;
            p1 = r1;
        } else {
            p2 = r0;
            p3 = r1;
        }
        // now p_u = term of accumulator j at chain position 4t+u
        s = __fadd_rn(s, p0);
        s = __fadd_rn(s, p1);
        s = __fadd_rn(s, p2);
        s = __fadd_rn(s, p3);
    }
    c = blocks << 2;
#endif
#if MN_SSE_QUAD2 > 0
    // QUAD is a property of the KERNEL (k_beam<SSE, 1>: the launcher picks it for rows of at least MN_SSE_QUAD2 blocks): the
    // twelve float4 registers cost occupancy (98 VGPRs, 16 wavefronts per CU), which pays at 768 floats and does not at 128,
    // where the kernels are latency-bound and keep the dword walk with 20 wavefronts per CU (1M x 128: 7.8 ms per 10k queries
    // against 9.0 with quads — and 9.1 when the quad code merely shared the kernel).  The lone-search kernels (LAT) keep their
    // LDS tiles.
    if (QUAD && !LAT) {
        const int blocks = steps >> 2;
        int t = 0;
        sse_blocks_quad<L2, MN_SSE_QUAD2>(row, q_lds, j, blocks, t, s);
        if (MN_SSE_QUAD2 > 8)
            sse_blocks_quad<L2, 8>(row, q_lds, j, blocks, t, s);
        if (MN_SSE_QUAD2 > 4)
            sse_blocks_quad<L2, 4>(row, q_lds, j, blocks, t, s);
        if (MN_SSE_QUAD2 > 2)
            sse_blocks_quad<L2, 2>(row, q_lds, j, blocks, t, s);
        sse_blocks_quad<L2, 1>(row, q_lds, j, blocks, t, s);
        c = blocks << 2;
    }
#endif
    // Whole batches first, with every load of a batch issued before its first use: the sums are one dependent chain, so the
    // loads are the only parallelism there is.  (Batches of 64 and 32 positions for the lone-search kernels were tried: 768-d
    // distances 7.7 -> 6.5 us, but the code they add slows every other step of those kernels by 5-8 % — they run one
    // wavefront against the instruction cache; the LDS tile of sse_rows_lat_tiled does the same job better.)
    // (round 4: issuing batch k+1's loads before batch k's sums — one round trip per 128-float row instead of two — made a lone
    // query SLOWER, 0.272 -> 0.303 ms at 3k x 128, same box, interleaved: these kernels are bound by instruction issue and fetch,
    // not by the loads; profiles/r04_ab_pipelined_loads.txt)
    sse_chunks<L2, MN_SSE_UNROLL>(row, q_lds, j, steps, c, s);
#pragma unroll 4
    for (; c < steps; c++) {
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

// SSE order with coalesced loads: 16 rows at a time, 128 elements of each per stage.  A wave-instruction
// fetches two rows' 512-B pieces as float4 per lane (full 128-B lines), the pieces are laid down in an
// LDS tile [16][128+4] and lane (r, j) then walks accumulator j of row r through the tile in the
// reference's order — same bits as sse_row, HBM access shape of the wave-order kernel.
#define MN_TILE_ROW 132 // floats per tile row: 128 + 4 pad → conflict-free b32 column reads
#define MN_TILE_FLOATS (16 * MN_TILE_ROW)

template <bool L2>
DEVI float sse_rows_tiled(const MnDevIndex &ix, const float *q_lds, float *tile, int myslot, int n, int lane) {
    float mine = 0.0f;
    const int j = lane & 3, r = lane >> 2;
    const int steps = ix.dim >> 2;
    const int half = lane >> 5, l32 = lane & 31;
    for (int t = 0; t < n; t += 16) {
        float s = 0.0f;
        const int nh = (ix.ld + 127) >> 7;
        // row base pointers of the two rows each load instruction serves (lanes 0-31 / 32-63)
        const float *rowp[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            int rr = t + 2 * i + half;
            int sl = __shfl(myslot, rr < n ? rr : n - 1);
            rowp[i] = ix.vectors + (size_t)sl * ix.ld + (l32 << 2);
        }
        float4 vn[8]; // software pipeline: stage h+1 is in flight while stage h is consumed from LDS
#pragma unroll
        for (int i = 0; i < 8; i++)
            vn[i] = (l32 << 2) < ix.ld ? *reinterpret_cast<const float4 *>(rowp[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int h = 0; h < nh; h++) {
            float4 v[8];
#pragma unroll
            for (int i = 0; i < 8; i++)
                v[i] = vn[i];
            const int en = ((h + 1) << 7) + (l32 << 2);
            if (h + 1 < nh) {
#pragma unroll
                for (int i = 0; i < 8; i++)
                    vn[i] = en < ix.ld ? *reinterpret_cast<const float4 *>(rowp[i] + ((h + 1) << 7))
                                       : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 8; i++)
                *reinterpret_cast<float4 *>(tile + (2 * i + half) * MN_TILE_ROW + (l32 << 2)) = v[i];
            __builtin_amdgcn_wave_barrier();
            const int c0 = h << 5; // chain positions of this stage: c0 .. c0+31
            const int cn = steps - c0 < 32 ? steps - c0 : 32;
            const float *trow = tile + r * MN_TILE_ROW + j;
            const float *qrow = q_lds + (h << 7) + j;
#pragma unroll 8
            for (int c = 0; c < cn; c++) {
                float b = trow[4 * c];
                float a = qrow[4 * c];
                float p;
                if (L2) {
                    float d = __fsub_rn(a, b);
                    p = __fmul_rn(d, d);
                } else {
                    p = __fmul_rn(a, b);
                }
                s = __fadd_rn(s, p);
            }
        }
        const int g = lane & ~3;
        float s0 = __shfl(s, g), s1 = __shfl(s, g + 1), s2 = __shfl(s, g + 2), s3 = __shfl(s, g + 3);
        float sum = __fadd_rn(__fadd_rn(__fadd_rn(s0, s1), s2), s3);
        if (steps * 4 < ix.dim) { // scalar tail (src/vec_math.c:91-94)
            int rr = t + r;
            int sl = __shfl(myslot, rr < n ? rr : n - 1);
            const float *row = ix.vectors + (size_t)sl * ix.ld;
            for (int i = steps * 4; i < ix.dim; i++) {
                float a = q_lds[i], b = row[i], p;
                if (L2) {
                    float d = __fsub_rn(a, b);
                    p = __fmul_rn(d, d);
                } else {
                    p = __fmul_rn(a, b);
                }
                sum = __fadd_rn(sum, p);
            }
        }
        float got = __shfl(sum, ((lane - t) & 15) << 2);
        if (lane >= t && lane < t + 16)
            mine = got;
    }
    return mine;
}

// A lone search's distance step, SSE order, with a per-wavefront LDS tile of R rows: the wavefront's few rows (a request
// is shared out over eight wavefronts, so 4 or so each) are fetched by ALL 64 lanes as coalesced float4s — every load of a
// pass in flight at once, one round trip to memory instead of dependent batches of dword gathers by four lanes per row — and
// stored TRANSPOSED: plane j of a row holds its elements 4c + j, c = 0, 1, ...  Quad g then walks row g's four chains exactly
// as sse_row walks them out of memory (same operations, same order, same bits), lane j reading plane j — and the query's
// plane j, transposed once per request — four chain positions per ds_read_b128.
__host__ __device__ inline int mn_lat_plane(int ld) { return (((ld >> 2) + 3) & ~3) + 4; }          // floats per plane
__host__ __device__ inline int mn_lat_row(int ld) { return 4 * mn_lat_plane(ld) + 8; }              // floats per tile row
__host__ __device__ inline int mn_lat_tile_floats(int ld, int rows) { return rows * mn_lat_row(ld) + 4 * mn_lat_plane(ld); }

template <bool L2, int R>
DEVI float sse_rows_lat_tiled(const MnDevIndex &ix, const float *q_lds, float *tile, int myslot, int n, int lane) {
    float mine = 0.0f;
    const int P = mn_lat_plane(ix.ld), RS = mn_lat_row(ix.ld);
    const int nf4 = ix.ld >> 2;
    const int j = lane & 3, g = lane >> 2;
    const int steps = ix.dim >> 2;
    float *qT = tile + R * RS;
    __builtin_amdgcn_wave_barrier();
    for (int e4 = lane; e4 < nf4; e4 += 64) { // the query's planes (q_lds is zero padded to ld)
        const float4 qv = *reinterpret_cast<const float4 *>(q_lds + e4 * 4);
        qT[e4] = qv.x;
        qT[P + e4] = qv.y;
        qT[2 * P + e4] = qv.z;
        qT[3 * P + e4] = qv.w;
    }
    for (int t = 0; t < n; t += R) {
        const int nr = n - t < R ? n - t : R;
        __builtin_amdgcn_wave_barrier(); // (the previous pass's chains are done with the tile)
        for (int cb = 0; cb < nf4; cb += 256) {
            float4 v[R][4];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int s = __builtin_amdgcn_readlane(myslot, t + (r < nr ? r : 0));
                const float4 *row = reinterpret_cast<const float4 *>(ix.vectors + (size_t)s * ix.ld);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int e4 = cb + lane + 64 * k;
                    v[r][k] = (r < nr && e4 < nf4) ? row[e4] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int e4 = cb + lane + 64 * k;
                    if (r < nr && e4 < nf4) {
                        float *tr = tile + r * RS + e4;
                        tr[0] = v[r][k].x;
                        tr[P] = v[r][k].y;
                        tr[2 * P] = v[r][k].z;
                        tr[3 * P] = v[r][k].w;
                    }
                }
        }
        __builtin_amdgcn_wave_barrier();
        float s = 0.0f;
        const int gr = g < nr ? g : 0;
        const float *tp = tile + gr * RS + j * P; // row gr, plane j
        const float *qp = qT + j * P;
        int c = 0;
        // (round 4: two register sets in turn, the LDS reads of the next 16 positions issued before the sums of these: a lone
        // query at 10k x 768 went from 0.538 to 0.580 ms, same box, interleaved — profiles/r04_ab_latency_variants.txt; not kept)
#pragma unroll 4
        for (; c + 4 <= steps; c += 4) {
            const float4 b = *reinterpret_cast<const float4 *>(tp + c);
            const float4 a = *reinterpret_cast<const float4 *>(qp + c);
            if (L2) {
                const float d0 = __fsub_rn(a.x, b.x), d1 = __fsub_rn(a.y, b.y), d2 = __fsub_rn(a.z, b.z), d3 = __fsub_rn(a.w, b.w);
                s = __fadd_rn(s, __fmul_rn(d0, d0));
                s = __fadd_rn(s, __fmul_rn(d1, d1));
                s = __fadd_rn(s, __fmul_rn(d2, d2));
                s = __fadd_rn(s, __fmul_rn(d3, d3));
            } else {
                s = __fadd_rn(s, __fmul_rn(a.x, b.x));
                s = __fadd_rn(s, __fmul_rn(a.y, b.y));
                s = __fadd_rn(s, __fmul_rn(a.z, b.z));
                s = __fadd_rn(s, __fmul_rn(a.w, b.w));
            }
        }
        for (; c < steps; c++) {
            const float b = tp[c], a = qp[c];
            float p;
            if (L2) {
                const float d = __fsub_rn(a, b);
                p = __fmul_rn(d, d);
            } else {
                p = __fmul_rn(a, b);
            }
            s = __fadd_rn(s, p);
        }
        const int gb = lane & ~3;
        const float s0 = __shfl(s, gb), s1 = __shfl(s, gb + 1), s2 = __shfl(s, gb + 2), s3 = __shfl(s, gb + 3);
        float sum = __fadd_rn(__fadd_rn(__fadd_rn(s0, s1), s2), s3);
        for (int i = steps * 4; i < ix.dim; i++) { // scalar tail (src/vec_math.c:91-94): element i is plane i & 3, place steps
            const float a = qT[(i & 3) * P + steps], b = tile[gr * RS + (i & 3) * P + steps];
            float p;
            if (L2) {
                const float d = __fsub_rn(a, b);
                p = __fmul_rn(d, d);
            } else {
                p = __fmul_rn(a, b);
            }
            sum = __fadd_rn(sum, p);
        }
        const float got = __shfl(sum, ((lane - t) & 15) << 2);
        if (lane >= t && lane < t + nr)
            mine = got;
    }
    return mine;
}

// Raw accumulation (dot or Σd²) of `n` rows (slot per lane, lanes < n valid) against q_lds.
// Returns, in lane i < n, the value for row i.
template <int ORDER, int NCH, bool L2, bool LAT = false>
DEVI float rows_accumulate(const MnDevIndex &ix, const float *q_lds, int myslot, int n, int lane, float *tile = nullptr,
                           int tile_rows = 0) {
    float mine = 0.0f;
    if (ORDER == MN_ORDER_SSE_V) {
        if (LAT && tile) { // (uniform)
            if (tile_rows >= 4)
                return sse_rows_lat_tiled<L2, 4>(ix, q_lds, tile, myslot, n, lane);
            if (tile_rows >= 2)
                return sse_rows_lat_tiled<L2, 2>(ix, q_lds, tile, myslot, n, lane);
        }
#ifdef MN_SSE_TILE_PATH // opt-in: same bits, same speed as the strided walk on gfx950 (both sit at the gather ceiling)
        if (tile)
            return sse_rows_tiled<L2>(ix, q_lds, tile, myslot, n, lane);
#endif
        for (int t = 0; t < n; t += 16) {
            int r = t + (lane >> 2);
            int s = __shfl(myslot, r < n ? r : n - 1);
            float v = sse_row<L2, LAT, (NCH == 1)>(ix.vectors + (size_t)s * ix.ld, q_lds, ix.dim, lane); // (SSE order: NCH 1 = quad loads)
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
template <int ORDER, int NCH, bool LAT = false>
DEVI float rows_distance(const MnDevIndex &ix, const float *q_lds, float qnorm, int myslot, int n, int lane,
                         float *tile = nullptr, int tile_rows = 0) {
    if (ix.metric == 0) {
        return rows_accumulate<ORDER, NCH, true, LAT>(ix, q_lds, myslot, n, lane, tile, tile_rows);
    }
    float dot = rows_accumulate<ORDER, NCH, false, LAT>(ix, q_lds, myslot, n, lane, tile, tile_rows);
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

