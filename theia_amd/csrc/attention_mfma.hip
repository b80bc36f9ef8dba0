// K4 (bf16 throughput path): multi-head self-attention forward / backward on the matrix cores.
//
// One workgroup (4 wave64) per (image, head); head_dim = 64; n <= 208 tokens (197 for DeiT), processed as 13 tiles of
// 16 query (resp. key) rows distributed round-robin over the waves.  The head's K,V (fwd, dQ) or Q,dO (dK/dV) live in
// LDS as natural [token][64] bf16 rows of pitch 144 B:
//   * "natural" operand fragments (16 rows x 8 consecutive d) are ds_read_b128 -- pitch 144 = 9*16 B makes the 16 rows
//     of a fragment hit 16 different 16-byte bank slots;
//   * "transposed" operand fragments (16 d-columns x 8 tokens) come from the SAME image through ds_read_b64_tr_b16.
// v_mfma_f32_16x16x32_bf16 everywhere.  The score tile is computed transposed (A = K rows, B = Q rows) so a lane owns
// ONE query row and 4 consecutive keys per 16-key tile: the softmax row reductions are lane-local + 2 shuffles, and
// P (resp. dS) feeds the next MFMA as its B operand straight from registers: two 16-key accumulator tiles are packed
// into one K=32 fragment whose k-slots are permuted as  slot(g, j<4) = 32s + 4g + j,  slot(g, j>=4) = 32s + 16 + 4g +
// (j-4);  the A operand (V^T / K^T / dO^T / Q^T via the transposing LDS read) uses the same permutation, so the
// contraction is unchanged.  Softmax statistics and all accumulation are f32.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 am_bf16x8;
typedef __attribute__((ext_vector_type(4))) float am_f32x4;
typedef __attribute__((ext_vector_type(4))) short am_s16x4;

constexpr int AM_PITCH = 144;   // bytes per LDS token row (64 bf16 + 16 B pad)
constexpr int AM_TILES = 13;    // 16-row tiles covering n <= 208
constexpr int AM_ROWS = 224;    // token rows kept in LDS (7 k-steps of 32)
constexpr int AM_KSTEPS = 7;

__device__ __forceinline__ void am_mma(am_f32x4& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(am_bf16x8, a), __builtin_bit_cast(am_bf16x8, b), acc, 0, 0, 0);
}

// keep `a` or zeros (component-wise: `c ? a : zero` on two uint4 lvalues is an lvalue select, i.e. a pointer, and sends the arrays to scratch)
__device__ __forceinline__ uint4 am_keep(bool c, const uint4& a) { return make_uint4(c ? a.x : 0u, c ? a.y : 0u, c ? a.z : 0u, c ? a.w : 0u); }

// Stage two [token][64] bf16 matrices (ROWS_A / ROWS_B token rows, zero beyond n) with the given global row strides.  All of a
// thread's 16-byte pieces of BOTH matrices are requested before the first is written to LDS: as a load -> wait -> store loop (what
// the compiler makes of the obvious form) the staging was 13 exposed memory latencies per workgroup, most of the forward kernel.
template <int ROWS_A, int ROWS_B, int NT>
__device__ __forceinline__ void am_stage2(const bf16_t* __restrict__ srcA, int64_t strideA, char* __restrict__ dstA,
                                          const bf16_t* __restrict__ srcB, int64_t strideB, char* __restrict__ dstB, int n) {
    constexpr int ITA = (ROWS_A * 8 + NT - 1) / NT, ITB = (ROWS_B * 8 + NT - 1) / NT;
    // Every load is UNCONDITIONAL, from a row clamped to [0, n) (rows beyond n become zeros by a select behind the load): a load under a
    // lane mask compiles to a branch, the compiler's wait-count bookkeeping turns conservative where the two sides meet
    // (s_waitcnt vmcnt(0)), and the "all requests first" order above degenerated into one exposed latency per matrix and iteration.
    uint4 xa[ITA], xb[ITB];
#pragma unroll
    for (int it = 0; it < ITA; ++it) {
        const int v = threadIdx.x + it * NT, r = min(v >> 3, n - 1), c = v & 7;
        xa[it] = *reinterpret_cast<const uint4*>(srcA + r * strideA + c * 8);
    }
#pragma unroll
    for (int it = 0; it < ITB; ++it) {
        const int v = threadIdx.x + it * NT, r = min(v >> 3, n - 1), c = v & 7;
        xb[it] = *reinterpret_cast<const uint4*>(srcB + r * strideB + c * 8);
    }
#pragma unroll
    for (int it = 0; it < ITA; ++it) {
        const int v = threadIdx.x + it * NT, r = v >> 3, c = v & 7;
        if ((ROWS_A * 8) % NT == 0 || r < ROWS_A) *reinterpret_cast<uint4*>(dstA + r * AM_PITCH + c * 16) = am_keep(r < n, xa[it]);
    }
#pragma unroll
    for (int it = 0; it < ITB; ++it) {
        const int v = threadIdx.x + it * NT, r = v >> 3, c = v & 7;
        if ((ROWS_B * 8) % NT == 0 || r < ROWS_B) *reinterpret_cast<uint4*>(dstB + r * AM_PITCH + c * 16) = am_keep(r < n, xb[it]);
    }
}

// natural fragment: row (row0 + lane&15), d = kk*32 + (lane>>4)*8 .. +8
__device__ __forceinline__ uint4 am_nat(const char* __restrict__ base, int row0, int kk, int lane) {
    return *reinterpret_cast<const uint4*>(base + (row0 + (lane & 15)) * AM_PITCH + (kk * 4 + (lane >> 4)) * 16);
}

// transposed fragment: column d = df*16 + (lane&15); tokens 32s + 4g + {0..3} and 32s + 16 + 4g + {0..3}
__device__ __forceinline__ uint4 am_tr(const char* __restrict__ base, int s, int df, int lane) {
    const int q = lane & 15, g = lane >> 4;
    const char* p = base + (s * 32 + g * 4 + (q >> 2)) * AM_PITCH + (df * 16 + (q & 3) * 4) * 2;
    const am_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) am_s16x4*)(p));
    const am_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) am_s16x4*)(p + 16 * AM_PITCH));
    const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
    return make_uint4(l2.x, l2.y, h2.x, h2.y);
}

__device__ __forceinline__ uint32_t am_pack2(float a, float b) { return pack2_bf16(a, b); }
constexpr float AM_C = 0.125f * 1.4426950408889634f;  // softmax scale 1/sqrt(64) folded with log2(e): exp(x/8 - m) = exp2(x*AM_C - m*log2e)
constexpr float AM_LOG2E = 1.4426950408889634f;
__device__ __forceinline__ float am_exp2(float x) { return __builtin_amdgcn_exp2f(x); }  // v_exp_f32 (inputs here are <= 0 or -inf)
__device__ __forceinline__ uint4 am_pack(const am_f32x4& t0, const am_f32x4& t1) {
    return make_uint4(am_pack2(t0[0], t0[1]), am_pack2(t0[2], t0[3]), am_pack2(t1[0], t1[1]), am_pack2(t1[2], t1[3]));
}
__device__ __forceinline__ float am_rsum(float v) {  // sum over the 4 lane groups (lanes l, l^16, l^32, l^48)
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float am_rmax(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}
__device__ __forceinline__ void am_store4(bf16_t* p, const am_f32x4& v, float scale) {
    *reinterpret_cast<uint2*>(p) = make_uint2(am_pack2(v[0] * scale, v[1] * scale), am_pack2(v[2] * scale, v[3] * scale));
}

// ================================================================================================ forward
// TAIL: n > 192, i.e. only the last 16-key tile straddles n and needs its keys masked (the per-element compare + select of the
// generic form was a quarter of the kernel's VALU instructions; DeiT: n = 197 / 196 / 204)
template <bool TAIL>
__global__ __launch_bounds__(256) void attn_fwd_mfma_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o,
                                                            float* __restrict__ lse, int n, int h) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    char* sK = sm;                          // [224][144] (rows 208 .. 223 are staged as zeros and never read: 224 * 8 pieces = 7 per thread,
    char* sV = sm + AM_ROWS * AM_PITCH;     // [224][144]  no partial pass -- in a partial pass the compiler sinks the load behind the lane mask)
    const int bh = blockIdx.x, bi = bh / h, hi = bh % h;
    const int D = h * 64;
    const int64_t rs = 3 * (int64_t)D;
    const bf16_t* base = qkv + (int64_t)bi * n * rs + hi * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q16 = lane & 15, g = lane >> 4;
    // The query fragments of a wave's NEXT tile are requested one tile ahead (the first one before K / V are staged): as a load at the
    // top of each iteration the wave sat out one global-memory latency per query tile, 3-4 times per head.
    // Unconditional loads from a clamped row (rows beyond n are zeroed at use): a load under a lane mask is a branch, and the compiler
    // drains the memory counter where its two sides meet -- the latency would be paid on the spot after all.
    uint4 qn[2];
    auto q_request = [&](int qt) {
        const int qrow = min(qt * 16 + q16, n - 1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) qn[kk] = *reinterpret_cast<const uint4*>(base + qrow * rs + kk * 32 + g * 8);
    };
    q_request(wave);
    am_stage2<AM_ROWS, AM_ROWS, 256>(base + D, rs, sK, base + 2 * D, rs, sV, n);
    // a use in front of the barrier (free: the staging has just waited for every younger load) -- without it the compiler sinks the
    // request to the fragments' first use behind the barrier, and the wave waits for it there
    asm volatile("" ::"v"(qn[0].x), "v"(qn[0].y), "v"(qn[0].z), "v"(qn[0].w), "v"(qn[1].x), "v"(qn[1].y), "v"(qn[1].z), "v"(qn[1].w));
    __syncthreads();
    for (int qt = wave; qt < AM_TILES; qt += 4) {
        const int qrow = qt * 16 + q16;
        const uint4 qf[2] = {am_keep(qrow < n, qn[0]), am_keep(qrow < n, qn[1])};
        q_request(qt + 4);
        am_f32x4 s[AM_TILES + 1];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < AM_TILES; ++kt) {
            am_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            am_mma(acc, am_nat(sK, kt * 16, 0, lane), qf[0]);
            am_mma(acc, am_nat(sK, kt * 16, 1, lane), qf[1]);
            if (!TAIL || kt == AM_TILES - 1) {  // keys beyond n: probability 0
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = kt * 16 + g * 4 + r < n ? acc[r] : -INFINITY;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, acc[r]);
            s[kt] = acc;  // raw q.k (the 1/8 scale is folded into the exponent below)
        }
        mx = am_rmax(mx);
        const float mc = mx * AM_C;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < AM_TILES; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = am_exp2(fmaf(s[kt][r], AM_C, -mc));
                s[kt][r] = p;
                sum += p;
            }
        s[AM_TILES] = (am_f32x4){0.f, 0.f, 0.f, 0.f};
        sum = am_rsum(sum);
        am_f32x4 oa[4];
#pragma unroll
        for (int df = 0; df < 4; ++df) oa[df] = (am_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < AM_KSTEPS; ++ks) {
            const uint4 pb = am_pack(s[2 * ks], s[2 * ks + 1]);
#pragma unroll
            for (int df = 0; df < 4; ++df) am_mma(oa[df], am_tr(sV, ks, df, lane), pb);
        }
        if (qrow < n) {
            const float inv = 1.0f / sum;
            bf16_t* orow = o + ((int64_t)bi * n + qrow) * D + hi * 64 + g * 4;
#pragma unroll
            for (int df = 0; df < 4; ++df) am_store4(orow + df * 16, oa[df], inv);
            if (g == 0) lse[(int64_t)bh * n + qrow] = mx * 0.125f + __logf(sum);
        }
    }
}

// ================================================================================================ backward: dQ (+ delta)
// 8 waves per workgroup (the backward kernels need 84 / 105 registers, the 62 KB of LDS allow two workgroups per CU): 4 waves per SIMD
// instead of 2 hide the MFMA -> exp2 -> pack -> MFMA chain of a step -- 198 -> 137 us for dQ + dK/dV at b=128 (the forward, rewritten
// the same way with its score tiles recomputed instead of held in 192 registers, stayed at 54-56 us and keeps its 4-wave form).
// 13 tiles over 8 waves: the 5 two-tile waves rotate with the block.
template <bool TAIL>
__global__ __launch_bounds__(512) void attn_bwd_dq_mfma_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                               const bf16_t* __restrict__ d_o, const float* __restrict__ lse,
                                                               bf16_t* __restrict__ dqkv, float* __restrict__ delta, int n, int h) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    char* sK = sm;                          // [224][144] (also read transposed)
    char* sV = sm + AM_ROWS * AM_PITCH;     // [208][144]
    const int bh = blockIdx.x, bi = bh / h, hi = bh % h;
    const int D = h * 64;
    const int64_t rs = 3 * (int64_t)D;
    const bf16_t* base = qkv + (int64_t)bi * n * rs + hi * 64;
    am_stage2<AM_ROWS, 208, 512>(base + D, rs, sK, base + 2 * D, rs, sV, n);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q16 = lane & 15, g = lane >> 4;
    for (int qt = (wave + bh) & 7; qt < AM_TILES; qt += 8) {
        const int qrow = qt * 16 + q16;
        const bool qok = qrow < n;
        const int64_t orow = ((int64_t)bi * n + qrow) * D + hi * 64;
        uint4 qf[2], gf[2];
        float dl = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            qf[kk] = gf[kk] = make_uint4(0, 0, 0, 0);
            if (qok) {
                qf[kk] = *reinterpret_cast<const uint4*>(base + qrow * rs + kk * 32 + g * 8);
                gf[kk] = *reinterpret_cast<const uint4*>(d_o + orow + kk * 32 + g * 8);
                float a8[8], b8[8];
                load8(d_o + orow + kk * 32 + g * 8, a8);
                load8(o + orow + kk * 32 + g * 8, b8);
#pragma unroll
                for (int j = 0; j < 8; ++j) dl += a8[j] * b8[j];
            }
        }
        dl = am_rsum(dl);
        const float lq2 = qok ? lse[(int64_t)bh * n + qrow] * AM_LOG2E : 0.f;
        // dS needs no row maximum (the forward's log-sum-exp is saved), so the 13 key tiles are streamed two at a time straight
        // into the dQ accumulators: nothing but one packed dS fragment lives across a step (the version that first computed all
        // 13 dS tiles needed 334 registers -> one wave per SIMD)
        am_f32x4 dq[4];
#pragma unroll
        for (int df = 0; df < 4; ++df) dq[df] = (am_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int ks = 0; ks < AM_KSTEPS; ++ks) {
            am_f32x4 dsv[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int kt = 2 * ks + t;
                am_f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                if (kt < AM_TILES) {  // wave-uniform (the 14th tile of the last step does not exist)
                    am_mma(st, am_nat(sK, kt * 16, 0, lane), qf[0]);
                    am_mma(st, am_nat(sK, kt * 16, 1, lane), qf[1]);
                    am_mma(dp, am_nat(sV, kt * 16, 0, lane), gf[0]);
                    am_mma(dp, am_nat(sV, kt * 16, 1, lane), gf[1]);
                }
                if (!TAIL || kt >= AM_TILES - 1) {  // keys beyond n get probability 0
#pragma unroll
                    for (int r = 0; r < 4; ++r) st[r] = kt * 16 + g * 4 + r < n ? st[r] : -INFINITY;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) dsv[t][r] = am_exp2(fmaf(st[r], AM_C, -lq2)) * (dp[r] - dl) * 0.125f;
            }
            const uint4 sb = am_pack(dsv[0], dsv[1]);
#pragma unroll
            for (int df = 0; df < 4; ++df) am_mma(dq[df], am_tr(sK, ks, df, lane), sb);
        }
        if (qok) {
            bf16_t* drow = dqkv + ((int64_t)bi * n + qrow) * rs + hi * 64 + g * 4;
#pragma unroll
            for (int df = 0; df < 4; ++df) am_store4(drow + df * 16, dq[df], 1.0f);
            if (g == 0) delta[(int64_t)bh * n + qrow] = dl;
        }
    }
}

// ================================================================================================ backward: dK, dV
__global__ __launch_bounds__(512) void attn_bwd_dkv_mfma_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ d_o,
                                                                const float* __restrict__ lse, const float* __restrict__ delta,
                                                                bf16_t* __restrict__ dqkv, int n, int h) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    char* sQ = sm;                                   // [224][144]
    char* sG = sm + AM_ROWS * AM_PITCH;              // [224][144]  dO
    float* sL = reinterpret_cast<float*>(sm + 2 * AM_ROWS * AM_PITCH);   // [224] lse (+inf beyond n)
    float* sD = sL + AM_ROWS;                                            // [224] delta
    const int bh = blockIdx.x, bi = bh / h, hi = bh % h;
    const int D = h * 64;
    const int64_t rs = 3 * (int64_t)D;
    const bf16_t* base = qkv + (int64_t)bi * n * rs + hi * 64;
    if (threadIdx.x < AM_ROWS) {  // requested ahead of the operand pieces below (one wait for everything)
        const int i = threadIdx.x;
        sL[i] = i < n ? lse[(int64_t)bh * n + i] * AM_LOG2E : INFINITY;  // log2-domain; +inf beyond n: probability 0
        sD[i] = i < n ? delta[(int64_t)bh * n + i] : 0.f;
    }
    am_stage2<AM_ROWS, AM_ROWS, 512>(base, rs, sQ, d_o + (int64_t)bi * n * D + hi * 64, (int64_t)D, sG, n);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k16 = lane & 15, g = lane >> 4;
    for (int kt = (wave + bh) & 7; kt < AM_TILES; kt += 8) {
        const int krow = kt * 16 + k16;
        const bool kok = krow < n;
        uint4 kf[2], vf[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            kf[kk] = vf[kk] = make_uint4(0, 0, 0, 0);
            if (kok) {
                kf[kk] = *reinterpret_cast<const uint4*>(base + krow * rs + D + kk * 32 + g * 8);
                vf[kk] = *reinterpret_cast<const uint4*>(base + krow * rs + 2 * D + kk * 32 + g * 8);
            }
        }
        am_f32x4 dk[4], dv[4];
#pragma unroll
        for (int df = 0; df < 4; ++df) dk[df] = dv[df] = (am_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int ks = 0; ks < AM_KSTEPS; ++ks) {
            am_f32x4 p[2], dsv[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int q0 = (2 * ks + t) * 16;
                am_f32x4 sa = {0.f, 0.f, 0.f, 0.f}, da = {0.f, 0.f, 0.f, 0.f};
                am_mma(sa, am_nat(sQ, q0, 0, lane), kf[0]);
                am_mma(sa, am_nat(sQ, q0, 1, lane), kf[1]);
                am_mma(da, am_nat(sG, q0, 0, lane), vf[0]);
                am_mma(da, am_nat(sG, q0, 1, lane), vf[1]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qi = q0 + g * 4 + r;
                    const float pv = am_exp2(fmaf(sa[r], AM_C, -sL[qi]));
                    p[t][r] = pv;
                    dsv[t][r] = pv * (da[r] - sD[qi]) * 0.125f;
                }
            }
            const uint4 pb = am_pack(p[0], p[1]), sb = am_pack(dsv[0], dsv[1]);
#pragma unroll
            for (int df = 0; df < 4; ++df) {
                am_mma(dv[df], am_tr(sG, ks, df, lane), pb);
                am_mma(dk[df], am_tr(sQ, ks, df, lane), sb);
            }
        }
        if (kok) {
            bf16_t* drow = dqkv + ((int64_t)bi * n + krow) * rs + hi * 64 + g * 4;
#pragma unroll
            for (int df = 0; df < 4; ++df) {
                am_store4(drow + D + df * 16, dk[df], 1.0f);
                am_store4(drow + 2 * D + df * 16, dv[df], 1.0f);
            }
        }
    }
}

// ================================================================================================ backward: one kernel
// dQ, dK and dV of a head from ONE evaluation of S, P, dP and dS (the two-kernel form evaluates them twice), bit-reproducible:
//   phase 0  Q and dO -> LDS, delta[q] = sum_d dO[q][d] * O[q][d] (8 lanes per row) -> LDS, lse -> LDS
//   phase 1  a wave owns a 16-key tile (13 of the 16 waves): the dK / dV loop of the two-kernel form, and every dS tile it
//            computes is also written to LDS as bf16 [key][query] (8-byte writes)
//   phase 2  K -> LDS over Q; a wave owns a 16-query tile: dQ = dS K with dS read back through the transposing LDS read as the
//            k-slot-permuted operand the dQ kernel builds in registers -- a fixed summation order, no atomics.
// LDS: Q, dO 2 x 31.5 KB + dS 208 x 464 B = 94.3 KB + lse / delta 1.8 KB = 159.0 KB: one 16-wave workgroup per CU (the same 16 waves
// per CU as two 8-wave workgroups of the two-kernel form).
__device__ __forceinline__ void am_unpack8(const uint4& a, float (&v)[8]) {
    const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(w[i] << 16);
        v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}
constexpr int AF_DS_PITCH = 464;   // bytes per dS row ([key][query]): 224 queries x 2 B + 16 (16 rows x 16 B of a half-wave's write: 64 distinct banks)
constexpr int AF_DS_ROWS = 208;
constexpr int AF_THREADS = 1024;

#ifdef AF_TRACE
__device__ unsigned long long af_trace_buf[4096 * 8];
#define AF_STAMP(i) do { if (threadIdx.x == 0) af_trace_buf[(size_t)bh * 8 + (i)] = wall_clock64(); } while (0)  // (bh: the head in flight)
extern "C" int theia_debug_attn_trace(unsigned long long* out, int nblocks) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(af_trace_buf), (size_t)nblocks * 8 * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#else
#define AF_STAMP(i) do { } while (0)
#endif
__global__ __launch_bounds__(AF_THREADS) void attn_bwd_fused_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                                    const bf16_t* __restrict__ d_o, const float* __restrict__ lse,
                                                                    bf16_t* __restrict__ dqkv, int n, int h, int nheads) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    char* sQ = sm;                                   // [224][144]; phase 2: K
    char* sG = sm + AM_ROWS * AM_PITCH;              // [224][144]  dO
    char* sDS = sm + 2 * AM_ROWS * AM_PITCH;         // [208 keys][464]  dS (bf16, scaled), [key][query]
    float* sL = reinterpret_cast<float*>(sDS + AF_DS_ROWS * AF_DS_PITCH);   // [224] lse * log2e (+inf beyond n)
    float* sD = sL + AM_ROWS;                                              // [224] delta
    const int D = h * 64;
    const int64_t rs = 3 * (int64_t)D;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, g = lane >> 4;
    // PERSISTENT: a workgroup walks over heads blockIdx.x, + gridDim.x, ... (one workgroup per CU: the LDS holds one head), and a wave requests
    // the NEXT head's Q / dO / O / lse pieces into registers as soon as its own phase-1 loop is done -- its phase-1 registers are dead then, the
    // loads fly while it waits for the slowest wave, through the K staging and phase 2.  As one-head workgroups every CU spent 4.5 us per head
    // waiting for these loads with the whole chip doing the same (the phase runs at HBM rate, then HBM idles for 12 us) + 2 us of workgroup
    // turnaround (AF_TRACE stamps).  (Round 4 measured a persistent form that requested the next head AHEAD of phase 1: 172 us against 134,
    // its 26 staging registers on top of phase 1's spilled.)
    uint4 xq[2], xg[2], xo[2];
    uint4 nkf[2], nvf[2];  // the head's K / V fragments of this wave's key tile (phase 1), requested with its Q / dO / O pieces
    float xl[2];
    // (Thread-index arithmetic inside the head loop starts from a value the compiler cannot see through: hoisted out of the loop, the per-thread
    //  row offsets of the five staged matrices lived in registers across phase 1 and were spilled to scratch -- 148 bytes per lane.)
    auto opaque_tid = [&]() {
        int t = tid;
        asm volatile("" : "+v"(t));
        return t;
    };
    // (Branch-free: every load unconditional from a row clamped into the head, the head index clamped into the batch -- the last iteration
    //  requests a head nobody uses.  Under a lane mask or a uniform `if` a load compiles to a branch, and where the sides meet the compiler
    //  drains the memory counter (s_waitcnt vmcnt(0)): the requests would be waited for on the spot.  Rows beyond n become zeros in phase 0.)
    auto request_head = [&](int bh_) {
        const int tid_ = opaque_tid();
        const bool more = bh_ < nheads;  // (behind the last head: every lane requests the first 16 bytes of a valid head -- one cache line per wave)
        bh_ = more ? bh_ : nheads - 1;
        const int bi_ = bh_ / h, hi_ = bh_ % h;
        const bf16_t* base_ = qkv + (int64_t)bi_ * n * rs + hi_ * 64;
        const bf16_t* gbase = d_o + (int64_t)bi_ * n * D + hi_ * 64;
        const bf16_t* obase = o + (int64_t)bi_ * n * D + hi_ * 64;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int v = more ? tid_ + it * AF_THREADS : 0, r = min(v >> 3, n - 1), c = v & 7;
            xq[it] = *reinterpret_cast<const uint4*>(base_ + r * rs + c * 8);
            xg[it] = *reinterpret_cast<const uint4*>(gbase + (int64_t)r * D + c * 8);
            xo[it] = *reinterpret_cast<const uint4*>(obase + (int64_t)r * D + c * 8);
            xl[it] = lse[(int64_t)bh_ * n + r];  // (scaled where it is used: a multiply here would wait for the load)
        }
        const int krow_ = more ? min(((wave + bh_) & 15) * 16 + (tid_ & 15), n - 1) : 0;  // (idle waves / keys beyond n: a clamped row, zeroed at use)
        const int g_ = more ? (tid_ & 63) >> 4 : 0;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            nkf[kk] = *reinterpret_cast<const uint4*>(base_ + krow_ * rs + D + kk * 32 + g_ * 8);
            nvf[kk] = *reinterpret_cast<const uint4*>(base_ + krow_ * rs + 2 * D + kk * 32 + g_ * 8);
        }
    };
    request_head(blockIdx.x);
    for (int bh = blockIdx.x; bh < nheads; bh += gridDim.x) {
        const int bi = bh / h, hi = bh % h;
        const bf16_t* base = qkv + (int64_t)bi * n * rs + hi * 64;
        AF_STAMP(0);
#ifdef AF_TRACE
        if (threadIdx.x == 0) {
            af_trace_buf[(size_t)bh * 8 + 6] = __builtin_amdgcn_s_getreg(63492);
            af_trace_buf[(size_t)bh * 8 + 7] = __builtin_amdgcn_s_getreg(63508);
        }
#endif
        // ---- phase 0: registers -> LDS, delta
        const int tid0 = opaque_tid();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int v = tid0 + it * AF_THREADS, r = v >> 3, c = v & 7;
            float a8[8], b8[8], part = 0.f;
            const bool live = r < n;  // (the pieces of rows beyond n were loaded from row n - 1)
            xq[it] = am_keep(live, xq[it]);
            xg[it] = am_keep(live, xg[it]);
            am_unpack8(xg[it], a8);
            am_unpack8(xo[it], b8);
#pragma unroll
            for (int j = 0; j < 8; ++j) part += a8[j] * b8[j];
            part += __shfl_xor(part, 1, 64);
            part += __shfl_xor(part, 2, 64);
            part += __shfl_xor(part, 4, 64);
            if (r < AM_ROWS) {
                *reinterpret_cast<uint4*>(sQ + r * AM_PITCH + c * 16) = xq[it];
                *reinterpret_cast<uint4*>(sG + r * AM_PITCH + c * 16) = xg[it];
                if (c == 0) {
                    sD[r] = part * 0.125f;   // (pre-scaled with the softmax scale, see phase 1) rows >= n: 0
                    sL[r] = live ? xl[it] * AM_LOG2E : INFINITY;  // rows >= n: +inf
                }
            }
        }
        __syncthreads();
        AF_STAMP(1);
        // this head's K pieces for phase 2, requested now
        uint4 xk[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int v = tid0 + it * AF_THREADS, r = v >> 3, c = v & 7;
            xk[it] = *reinterpret_cast<const uint4*>(base + min(r, n - 1) * rs + D + c * 8);  // (rows beyond n: zeroed when staged)
        }
        const int mytile = (wave + bh) & 15;  // 13 of the 16 waves own a tile; which three idle rotates with the head
        // ---- phase 1: dK, dV (+ dS -> LDS)
        if (mytile < AM_TILES) {
            const int krow = mytile * 16 + l16;
            const bool kok = krow < n;
            uint4 kf[2], vf[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                kf[kk] = am_keep(kok, nkf[kk]);
                vf[kk] = am_keep(kok, nvf[kk]);
            }
            // The per-score arithmetic is what bounds this loop (exp2 at quarter rate + 5 more issue slots per score, 8 scores per lane and
            // tile), so everything that can leave it does.  dS = P (dP - delta) / 8: the 1/8 goes into V (dP = dO V^T) and delta, both exact
            // (a power of two).  Keys beyond n need no select: their K and V fragments are zeros, so their P = exp2(-lse) and dS = -P delta / 8
            // are finite, their dK / dV rows are not stored, and in dQ = dS K they meet the zero rows K is staged with.
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                float v8[8];
                am_unpack8(vf[kk], v8);
                vf[kk] = make_uint4(am_pack2(v8[0] * 0.125f, v8[1] * 0.125f), am_pack2(v8[2] * 0.125f, v8[3] * 0.125f),
                                    am_pack2(v8[4] * 0.125f, v8[5] * 0.125f), am_pack2(v8[6] * 0.125f, v8[7] * 0.125f));
            }
            am_f32x4 dk[4], dv[4];
#pragma unroll
            for (int df = 0; df < 4; ++df) dk[df] = dv[df] = (am_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int ks = 0; ks < AM_KSTEPS; ++ks) {
                am_f32x4 p[2], dsv[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int q0 = (2 * ks + t) * 16;
                    if (t == 1 && 2 * ks + 1 >= AM_TILES) {  // the 14th query tile does not exist (uniform; rows 208 .. 223: P = 0)
                        p[1] = dsv[1] = (am_f32x4){0.f, 0.f, 0.f, 0.f};
                        continue;
                    }
                    am_f32x4 sa = {0.f, 0.f, 0.f, 0.f}, da = {0.f, 0.f, 0.f, 0.f};
                    am_mma(sa, am_nat(sQ, q0, 0, lane), kf[0]);
                    am_mma(sa, am_nat(sQ, q0, 1, lane), kf[1]);
                    am_mma(da, am_nat(sG, q0, 0, lane), vf[0]);
                    am_mma(da, am_nat(sG, q0, 1, lane), vf[1]);
                    // (the four rows' lse / delta as one 16-byte LDS read each: sixteen 4-byte reads per step were a third of the loop's LDS instructions)
                    const float4 l4 = *reinterpret_cast<const float4*>(sL + q0 + g * 4), d4 = *reinterpret_cast<const float4*>(sD + q0 + g * 4);
                    const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq4[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = am_exp2(fmaf(sa[r], AM_C, -lq[r]));
                        p[t][r] = pv;
                        dsv[t][r] = pv * (da[r] - dq4[r]);
                    }
                }
                const uint4 pb = am_pack(p[0], p[1]), sb = am_pack(dsv[0], dsv[1]);
                // dS for phase 2, stored [key][query]: this lane holds key krow x queries q0 + 4g + {0..3} (tile t = 0: sb.x/.y, t = 1: .z/.w) --
                // two 8-byte writes (as [query][key] they were eight 2-byte ones); phase 2 reads it back through the transposing LDS read
                {
                    char* w0 = sDS + krow * AF_DS_PITCH + (32 * ks + g * 4) * 2;
                    *reinterpret_cast<uint2*>(w0) = make_uint2(sb.x, sb.y);
                    if (2 * ks + 1 < AM_TILES) *reinterpret_cast<uint2*>(w0 + 32) = make_uint2(sb.z, sb.w);  // (the 14th query tile does not exist)
                }
#pragma unroll
                for (int df = 0; df < 4; ++df) {
                    am_mma(dv[df], am_tr(sG, ks, df, lane), pb);
                    am_mma(dk[df], am_tr(sQ, ks, df, lane), sb);
                }
            }
            if (kok) {
                bf16_t* drow = dqkv + ((int64_t)bi * n + krow) * rs + hi * 64 + g * 4;
#pragma unroll
                for (int df = 0; df < 4; ++df) {
                    am_store4(drow + D + df * 16, dk[df], 1.0f);
                    am_store4(drow + 2 * D + df * 16, dv[df], 1.0f);
                }
            }
        }
        request_head(bh + (int)gridDim.x);  // the next head's pieces: see the top of the loop
        AF_STAMP(5);
        __syncthreads();  // every wave has finished reading Q / dO and writing dS
        AF_STAMP(2);
        // ---- phase 2: K -> LDS (over Q), dQ = dS K
        const int tid2 = opaque_tid();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int v = tid2 + it * AF_THREADS, r = v >> 3, c = v & 7;
            if (r < AM_ROWS) *reinterpret_cast<uint4*>(sQ + r * AM_PITCH + c * 16) = am_keep(r < n, xk[it]);
        }
        __syncthreads();
        AF_STAMP(3);
        if (mytile < AM_TILES) {
            const int qrow = mytile * 16 + l16;
            am_f32x4 dq[4];
#pragma unroll
            for (int df = 0; df < 4; ++df) dq[df] = (am_f32x4){0.f, 0.f, 0.f, 0.f};
            // dS[query tile][keys 32 ks + 4g + {0..3}, 32 ks + 16 + 4g + {0..3}] out of the [key][query] table: the transposing read of am_tr with
            // the table's pitch (keys 208 .. 223 of the last step do not exist: zeros)
            const char* dcol = sDS + (g * 4 + (l16 >> 2)) * AF_DS_PITCH + (mytile * 16 + (l16 & 3) * 4) * 2;
#pragma unroll
            for (int ks = 0; ks < AM_KSTEPS; ++ks) {
                const am_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) am_s16x4*)(dcol + ks * 32 * AF_DS_PITCH));
                uint2 h2 = make_uint2(0u, 0u);
                if (ks * 32 + 16 < AF_DS_ROWS)
                    h2 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                                       (__attribute__((address_space(3))) am_s16x4*)(dcol + (ks * 32 + 16) * AF_DS_PITCH)));
                const uint2 l2 = __builtin_bit_cast(uint2, lo);
                const uint4 sb = make_uint4(l2.x, l2.y, h2.x, h2.y);
#pragma unroll
                for (int df = 0; df < 4; ++df) am_mma(dq[df], am_tr(sQ, ks, df, lane), sb);
            }
            if (qrow < n) {
                bf16_t* drow = dqkv + ((int64_t)bi * n + qrow) * rs + hi * 64 + g * 4;
#pragma unroll
                for (int df = 0; df < 4; ++df) am_store4(drow + df * 16, dq[df], 1.0f);
            }
        }
        AF_STAMP(4);
        __syncthreads();  // phase 2 has read K (over Q) and dS: the next head's phase 0 may overwrite them
    }
}

// ------------------------------------------------------------------------------------------------ launchers (called from attention.hip)
static void am_set_lds(const void* kern, int bytes) {
    static const void* seen[8];
    static int nseen = 0;
    for (int i = 0; i < nseen; ++i)
        if (seen[i] == kern) return;
    (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (nseen < 8) seen[nseen++] = kern;
}

int theia_attention_fwd_mfma(const void* qkv, void* o, float* lse, int b, int n, int h, hipStream_t s) {
    const int lds = 2 * AM_ROWS * AM_PITCH;
    am_set_lds(reinterpret_cast<const void*>(attn_fwd_mfma_kernel<true>), lds);
    am_set_lds(reinterpret_cast<const void*>(attn_fwd_mfma_kernel<false>), lds);
    if (n > 192) hipLaunchKernelGGL(attn_fwd_mfma_kernel<true>, dim3(b * h), dim3(256), lds, s, (const bf16_t*)qkv, (bf16_t*)o, lse, n, h);
    else hipLaunchKernelGGL(attn_fwd_mfma_kernel<false>, dim3(b * h), dim3(256), lds, s, (const bf16_t*)qkv, (bf16_t*)o, lse, n, h);
    THEIA_CHECK_LAUNCH("theia_attention_fwd(mfma)");
    return THEIA_OK;
}

int theia_attention_bwd_mfma(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv, float* delta,
                             int b, int n, int h, hipStream_t s) {
    static int fused = -1;  // THEIA_ATTN_BWD=split: the two-kernel form (A/B switch)
    if (fused < 0) {
        const char* e = getenv("THEIA_ATTN_BWD");
        fused = (e != nullptr && strcmp(e, "split") == 0) ? 0 : 1;
    }
    if (fused) {
        const int ldsf = 2 * AM_ROWS * AM_PITCH + AF_DS_ROWS * AF_DS_PITCH + 2 * AM_ROWS * (int)sizeof(float);
        am_set_lds(reinterpret_cast<const void*>(attn_bwd_fused_kernel), ldsf);
        const int cus = theia_compute_cus();
        hipLaunchKernelGGL(attn_bwd_fused_kernel, dim3(b * h < cus ? b * h : cus), dim3(AF_THREADS), ldsf, s, (const bf16_t*)qkv, (const bf16_t*)o,
                           (const bf16_t*)d_o, lse, (bf16_t*)dqkv, n, h, b * h);
        THEIA_CHECK_LAUNCH("theia_attention_bwd(fused mfma)");
        return THEIA_OK;
    }
    const int lds1 = (AM_ROWS + 208) * AM_PITCH;
    const int lds2 = 2 * AM_ROWS * AM_PITCH + 2 * AM_ROWS * (int)sizeof(float);
    am_set_lds(reinterpret_cast<const void*>(attn_bwd_dq_mfma_kernel<true>), lds1);
    am_set_lds(reinterpret_cast<const void*>(attn_bwd_dq_mfma_kernel<false>), lds1);
    am_set_lds(reinterpret_cast<const void*>(attn_bwd_dkv_mfma_kernel), lds2);
    if (n > 192)
        hipLaunchKernelGGL(attn_bwd_dq_mfma_kernel<true>, dim3(b * h), dim3(512), lds1, s, (const bf16_t*)qkv, (const bf16_t*)o,
                           (const bf16_t*)d_o, lse, (bf16_t*)dqkv, delta, n, h);
    else
        hipLaunchKernelGGL(attn_bwd_dq_mfma_kernel<false>, dim3(b * h), dim3(512), lds1, s, (const bf16_t*)qkv, (const bf16_t*)o,
                           (const bf16_t*)d_o, lse, (bf16_t*)dqkv, delta, n, h);
    THEIA_CHECK_LAUNCH("theia_attention_bwd(dq mfma)");
    hipLaunchKernelGGL(attn_bwd_dkv_mfma_kernel, dim3(b * h), dim3(512), lds2, s, (const bf16_t*)qkv, (const bf16_t*)d_o, lse,
                       delta, (bf16_t*)dqkv, n, h);
    THEIA_CHECK_LAUNCH("theia_attention_bwd(dkv mfma)");
    return THEIA_OK;
}
