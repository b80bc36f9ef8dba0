// LayerNorm kernels of the Theia hot path (HBM-bound, wavefront-reduced, 16-byte vector accesses).
//   row LayerNorm (ViT blocks, eps 1e-12)            transformers modeling_vit.py:261-262,348
//   whole-sample LayerNorm over (C,H,W) with affine   adapter_heads.py:306,309,312,318,321,324
#include "common.h"

// ================================================================================================
// row LayerNorm: one wave64 per row, row held in registers (D <= 2048, D % 8 == 0)
// ================================================================================================
constexpr int LN_MAXV = 4;  // 8-element vectors per lane



// NV = 8-element vectors per lane: 2 for D <= 1024 (the ViT widths), else NV -- the row and, in backward, the per-lane
// dgamma/dbeta accumulators live in registers, so NV sets the register count (backward: 226 VGPRs at NV = 4).
// Q8: also the e4m3 copy of y (theia_layernorm_fwd_q8).  A separate instantiation: with the side output as a run-time branch of the one kernel
// the compiler contracted the affine expression differently and the f32 parity mode's outputs moved by an ulp (3.5e-6 on the predictions --
// enough to push a sampled gradient of the G5 golden test from 2.9e-3 to 5.8e-3 of its tensor's RMS): the plain instantiation is the
// round-5 kernel, instruction for instruction.
template <typename T, int NV, bool Q8 = false>
__global__ __launch_bounds__(256) void ln_row_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, T* __restrict__ y,
                                                         float* __restrict__ mean, float* __restrict__ rstd, int64_t M,
                                                         int D, float eps, const theia_q8_out_t q8) {
    const int lane = threadIdx.x & 63;
    float qsc = 0.f, qam = 0.f;
    if constexpr (Q8) qsc = *q8.scale;
    const int nv = D >> 3;
    const float invD = 1.0f / (float)D;
    // Every load is unconditional, from a vector index clamped into the row (lanes past the row mask their contribution with a select),
    // and the row's x pieces AND the affine pieces are requested together at the top: under `if (vi < nv)` each piece was a branch with
    // its own load -> s_waitcnt vmcnt(0) -> use (the compiler drains the memory counter where the sides of a branch meet), and the
    // affine rows were fetched behind the statistics -- three exposed latencies per row for a wave that lives for one or two rows.
    int vc[NV];
    bool ok[NV];
    float g8[NV][8], b8[NV][8];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = lane + 64 * i;
        ok[i] = vi < nv;
        vc[i] = ok[i] ? vi : nv - 1;
    }
    bool first = true;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += (int64_t)gridDim.x * 4) {
        const T* xr = x + row * D;
        float v[NV][8];
#pragma unroll
        for (int i = 0; i < NV; ++i) load8(xr + vc[i] * 8, v[i]);
        if (first) {  // (wave-uniform; the affine pieces are the same for every row of the wave)
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                load8(gamma + vc[i] * 8, g8[i]);
                load8(beta + vc[i] * 8, b8[i]);
            }
            first = false;
        }
        __builtin_amdgcn_sched_barrier(0);  // the requests stay in front of the statistics (the scheduler moves loads down to their first use)
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) s += ok[i] ? v[i][j] : 0.f;
        const float mu = wave_sum(s) * invD;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = v[i][j] - mu;
                q += ok[i] ? d * d : 0.f;
            }
        const float var = wave_sum(q) * invD;
        const float rs = 1.0f / sqrtf(var + eps);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (v[i][j] - mu) * rs * g8[i][j] + b8[i][j];  // (outside the mask: a use under it would pull the loads in)
            if (ok[i]) store8(y + row * D + vc[i] * 8, o);
            if constexpr (Q8) {
                if (ok[i]) q8_store8(q8.out, row * D + vc[i] * 8, o, qsc, qam);
            }
        }
        if (lane == 0) {
            mean[row] = mu;
            rstd[row] = rs;
        }
    }
    if constexpr (Q8) q8_flush_wave(q8.amax, qam);
}

// D <= 256 (DeiT-tiny: 24 of a wave's 64 lanes hold a vector of the row): TWO rows per wave, one per 32-lane half -- half the loads,
// shuffles and stores per row; the sums run over the half's 32 lanes (another order of the same additions than the one-row form).
__device__ __forceinline__ float half_wave_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <typename T, bool Q8>
__global__ __launch_bounds__(256) void ln_row_fwd2_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, T* __restrict__ y,
                                                          float* __restrict__ mean, float* __restrict__ rstd, int64_t M,
                                                          int D, float eps, const theia_q8_out_t q8) {
    const int lane = threadIdx.x & 63, l32 = lane & 31, half = lane >> 5;
    float qsc = 0.f, qam = 0.f;
    if constexpr (Q8) qsc = *q8.scale;
    const int nv = D >> 3;
    const float invD = 1.0f / (float)D;
    const bool ok = l32 < nv;
    const int vc = ok ? l32 : nv - 1;
    float g8[8], b8[8];
    bool first = true;
    for (int64_t r0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2; r0 < M; r0 += (int64_t)gridDim.x * 8) {
        const bool rok = r0 + half < M;
        const int64_t row = rok ? r0 + half : M - 1;  // (the odd last row's partner: loads from a clamped row, stores nothing)
        float v[8];
        load8(x + row * D + vc * 8, v);
        if (first) {
            load8(gamma + vc * 8, g8);
            load8(beta + vc * 8, b8);
            first = false;
        }
        __builtin_amdgcn_sched_barrier(0);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += ok ? v[j] : 0.f;
        const float mu = half_wave_sum(s) * invD;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float d = v[j] - mu;
            q += ok ? d * d : 0.f;
        }
        const float var = half_wave_sum(q) * invD;
        const float rs = 1.0f / sqrtf(var + eps);
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (v[j] - mu) * rs * g8[j] + b8[j];
        if (ok && rok) store8(y + row * D + vc * 8, o);
        if constexpr (Q8) {
            if (ok && rok) q8_store8(q8.out, row * D + vc * 8, o, qsc, qam);
        }
        if (l32 == 0 && rok) {
            mean[row] = mu;
            rstd[row] = rs;
        }
    }
    if constexpr (Q8) q8_flush_wave(q8.amax, qam);
}
static bool ln_two_rows(int D) {
    static int on = -1;  // THEIA_LN_TWO_ROWS=0: the one-row-per-wave kernels for every D (A/B switch)
    if (on < 0) {
        const char* e = getenv("THEIA_LN_TWO_ROWS");
        on = (e != nullptr && e[0] == '0') ? 0 : 1;
    }
    return on != 0 && D <= 256;
}

extern "C" int theia_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                   float* rstd, int64_t M, int D, float eps, int dtype, void* stream) {
    THEIA_CHECK_ARG(x && gamma && beta && y && mean && rstd, "theia_layernorm_fwd: null pointer");
    THEIA_CHECK_ARG(M > 0 && D > 0 && D % 8 == 0 && D <= 8 * 64 * LN_MAXV, "theia_layernorm_fwd: unsupported D=%d", D);
    int blocks = (int)((M + 3) / 4);
    if (blocks > 8192) blocks = 8192;
    const theia_q8_out_t q8 = q8_take();
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // 8-element vectors per lane: 1 for D <= 512 (DeiT-small / -tiny: a second, fully masked vector would still be loaded -- the loads are
    // unconditional), 2 for D <= 1024, else LN_MAXV
    const int nvl = D <= 512 ? 1 : D <= 1024 ? 2 : LN_MAXV;
    if (ln_two_rows(D)) {
        int b2 = (int)((M + 7) / 8);
        if (b2 > 8192) b2 = 8192;
#define LN_FWD2_LAUNCH(TT, QQ) \
    hipLaunchKernelGGL((ln_row_fwd2_kernel<TT, QQ>), dim3(b2), dim3(256), 0, s, (const TT*)x, gamma, beta, (TT*)y, mean, rstd, M, D, eps, q8)
        if (dtype == THEIA_BF16 && q8.out != nullptr) LN_FWD2_LAUNCH(bf16_t, true);
        else if (dtype == THEIA_BF16) LN_FWD2_LAUNCH(bf16_t, false);
        else if (dtype == THEIA_F32) LN_FWD2_LAUNCH(float, false);
        else THEIA_CHECK_ARG(false, "theia_layernorm_fwd: bad dtype %d", dtype);
#undef LN_FWD2_LAUNCH
        THEIA_CHECK_LAUNCH("theia_layernorm_fwd");
        return THEIA_OK;
    }
#define LN_FWD_LAUNCH(TT, NVV, QQ)                                                                                                            \
    hipLaunchKernelGGL((ln_row_fwd_kernel<TT, NVV, QQ>), dim3(blocks), dim3(256), 0, s, (const TT*)x, gamma, beta, (TT*)y, mean, rstd, M, D, eps, q8)
    if (dtype == THEIA_BF16 && q8.out != nullptr) {
        if (nvl == 1) LN_FWD_LAUNCH(bf16_t, 1, true); else if (nvl == 2) LN_FWD_LAUNCH(bf16_t, 2, true); else LN_FWD_LAUNCH(bf16_t, LN_MAXV, true);
    } else if (dtype == THEIA_BF16) {
        if (nvl == 1) LN_FWD_LAUNCH(bf16_t, 1, false); else if (nvl == 2) LN_FWD_LAUNCH(bf16_t, 2, false); else LN_FWD_LAUNCH(bf16_t, LN_MAXV, false);
    } else if (dtype == THEIA_F32) {
        if (nvl == 1) LN_FWD_LAUNCH(float, 1, false); else if (nvl == 2) LN_FWD_LAUNCH(float, 2, false); else LN_FWD_LAUNCH(float, LN_MAXV, false);
    } else
        THEIA_CHECK_ARG(false, "theia_layernorm_fwd: bad dtype %d", dtype);
#undef LN_FWD_LAUNCH
    THEIA_CHECK_LAUNCH("theia_layernorm_fwd");
    return THEIA_OK;
}

extern "C" int theia_layernorm_fwd_q8(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int64_t M,
                                      int D, float eps, int dtype, const theia_q8_out_t* q8, void* stream) {
    Q8_FORWARD("theia_layernorm_fwd_q8", dtype, q8, theia_layernorm_fwd(x, gamma, beta, y, mean, rstd, M, D, eps, dtype, stream));
}

// backward: dx per row (wave); dgamma/dbeta accumulated per lane over the rows this wave visits, then
// block-reduced through LDS and written as one partial row per block; a second kernel sums the partials.
template <typename T, int NV, bool HAS_RES, bool Q8 = false>
__global__ __launch_bounds__(256) void ln_row_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                         const float* __restrict__ gamma, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, const T* __restrict__ dres,
                                                         T* __restrict__ dx, float* __restrict__ part, int64_t M, int D,
                                                         const theia_q8_out_t q8) {
    extern __shared__ float red[];  // [4][2*D]
    float qsc = 0.f, qam = 0.f;
    if constexpr (Q8) qsc = *q8.scale;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = D >> 3;
    const float invD = 1.0f / (float)D;
    // (unconditional loads from a clamped vector index, all of a row's pieces requested together: see ln_row_fwd_kernel)
    int vc[NV];
    bool ok[NV];
    float ag[NV][8], ab[NV][8], g8[NV][8];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = lane + 64 * i;
        ok[i] = vi < nv;
        vc[i] = ok[i] ? vi : nv - 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) ag[i][j] = ab[i][j] = 0.f;
        load8(gamma + vc[i] * 8, g8[i]);
    }
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < M; row += (int64_t)gridDim.x * 4) {
        float xv[NV][8], dv[NV][8], rr[HAS_RES ? NV : 1][8];  // rr: the residual-stream gradient, requested with the row (not after the
#pragma unroll                                                   // reduction: a second exposed memory latency per row)
        for (int i = 0; i < NV; ++i) {
            load8(x + row * D + vc[i] * 8, xv[i]);
            load8(dy + row * D + vc[i] * 8, dv[i]);
            if constexpr (HAS_RES) load8(dres + row * D + vc[i] * 8, rr[i]);
        }
        const float mu = mean[row], rs = rstd[row];
        __builtin_amdgcn_sched_barrier(0);  // every request of the row in front of the arithmetic
        float xh[NV][8], gy[NV][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                xh[i][j] = (xv[i][j] - mu) * rs;
                gy[i][j] = dv[i][j] * g8[i][j];
                s1 += ok[i] ? gy[i][j] : 0.f;
                s2 += ok[i] ? gy[i][j] * xh[i][j] : 0.f;
                ag[i][j] += ok[i] ? dv[i][j] * xh[i][j] : 0.f;
                ab[i][j] += ok[i] ? dv[i][j] : 0.f;
            }
        }
        const float m1 = wave_sum(s1) * invD, m2 = wave_sum(s2) * invD;
        if constexpr (HAS_RES) {  // a use outside the store mask (the data arrived long ago): otherwise the compiler sinks a piece's request into
#pragma unroll                    // the masked block that is its only user, and the wave waits for it there
            for (int i = 0; i < NV; ++i) asm volatile("" ::"v"(rr[i][0]));
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = rs * (gy[i][j] - m1 - xh[i][j] * m2);
            if constexpr (HAS_RES) {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] += rr[i][j];
            }
            if (ok[i]) store8(dx + row * D + vc[i] * 8, o);
            if constexpr (Q8) {
                if (ok[i]) q8_store8(q8.out, row * D + vc[i] * 8, o, qsc, qam);
            }
        }
    }
    if constexpr (Q8) q8_flush_wave(q8.amax, qam);
    float* mine = red + wave * 2 * D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = lane + 64 * i;
        if (vi < nv) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                mine[vi * 8 + j] = ag[i][j];
                mine[D + vi * 8 + j] = ab[i][j];
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * D; i += 256)
        part[(int64_t)blockIdx.x * 2 * D + i] = red[i] + red[2 * D + i] + red[4 * D + i] + red[6 * D + i];
}

// two rows per wave for D <= 256 (see ln_row_fwd2_kernel); the affine partials of a block are the sum over its 8 half-waves
template <typename T, bool HAS_RES, bool Q8>
__global__ __launch_bounds__(256) void ln_row_bwd2_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                          const float* __restrict__ gamma, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const T* __restrict__ dres,
                                                          T* __restrict__ dx, float* __restrict__ part, int64_t M, int D,
                                                          const theia_q8_out_t q8) {
    extern __shared__ float red[];  // [8][2*D]
    float qsc = 0.f, qam = 0.f;
    if constexpr (Q8) qsc = *q8.scale;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l32 = lane & 31, half = lane >> 5;
    const int nv = D >> 3;
    const float invD = 1.0f / (float)D;
    const bool ok = l32 < nv;
    const int vc = ok ? l32 : nv - 1;
    float ag[8], ab[8], g8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) ag[j] = ab[j] = 0.f;
    load8(gamma + vc * 8, g8);
    for (int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * 2; r0 < M; r0 += (int64_t)gridDim.x * 8) {
        const bool rok = r0 + half < M, use = ok && rok;
        const int64_t row = rok ? r0 + half : M - 1;
        float xv[8], dv[8], rr[8];
        load8(x + row * D + vc * 8, xv);
        load8(dy + row * D + vc * 8, dv);
        if constexpr (HAS_RES) load8(dres + row * D + vc * 8, rr);
        const float mu = mean[row], rs = rstd[row];
        __builtin_amdgcn_sched_barrier(0);
        float xh[8], gy[8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            xh[j] = (xv[j] - mu) * rs;
            gy[j] = dv[j] * g8[j];
            s1 += ok ? gy[j] : 0.f;
            s2 += ok ? gy[j] * xh[j] : 0.f;
            ag[j] += use ? dv[j] * xh[j] : 0.f;
            ab[j] += use ? dv[j] : 0.f;
        }
        const float m1 = half_wave_sum(s1) * invD, m2 = half_wave_sum(s2) * invD;
        if constexpr (HAS_RES) asm volatile("" ::"v"(rr[0]));
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = rs * (gy[j] - m1 - xh[j] * m2);
        if constexpr (HAS_RES) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += rr[j];
        }
        if (use) store8(dx + row * D + vc * 8, o);
        if constexpr (Q8) {
            if (use) q8_store8(q8.out, row * D + vc * 8, o, qsc, qam);
        }
    }
    if constexpr (Q8) q8_flush_wave(q8.amax, qam);
    float* mine = red + (wave * 2 + half) * 2 * D;
    if (ok) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            mine[l32 * 8 + j] = ag[j];
            mine[D + l32 * 8 + j] = ab[j];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * D; i += 256) {
        float t = red[i];
#pragma unroll
        for (int h = 1; h < 8; ++h) t += red[h * 2 * D + i];
        part[(int64_t)blockIdx.x * 2 * D + i] = t;
    }
}

// out[c] (+)= sum_p part[p][c]  for c in [0, ncol); deterministic order.
// COLS columns x (256 / COLS) part lanes per block: a thread-per-column loop over hundreds of partial rows is latency-bound,
// so the row-LN reduction (512 partial rows, 1536 columns) uses 16 x 16 (96 blocks, 32 loads per thread: 24 -> ~8 us) and the
// wide whole-sample reductions (few partial rows, millions of columns) 64 x 4.
template <int COLS>
__global__ __launch_bounds__(256) void partial_reduce_kernel(const float* __restrict__ part, int nparts, int ncol, int64_t pitch,
                                                             float* __restrict__ out0, float* __restrict__ out1, int split,
                                                             int accumulate) {
    constexpr int LANES = 256 / COLS;
    __shared__ float red[LANES][COLS];
    const int cl = threadIdx.x % COLS, pl = threadIdx.x / COLS;
    const int c = blockIdx.x * COLS + cl;
    float s = 0.f;
    if (c < ncol) {
        // 8 partial rows per pass, requested together (unconditionally, from a clamped row; rows past the end add 0): as a load -> add loop
        // this was one L2 latency per partial row -- 32 in a row for the row-LayerNorm reduction, 11 us for 3 MB.  Same order of additions.
        for (int p0 = pl; p0 < nparts; p0 += LANES * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int p = p0 + u * LANES;
                v[u] = part[(int64_t)(p < nparts ? p : nparts - 1) * pitch + c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += p0 + u * LANES < nparts ? v[u] : 0.f;
        }
    }
    red[pl][cl] = s;
    __syncthreads();
    if (pl == 0 && c < ncol) {
        s = 0.f;
#pragma unroll
        for (int q = 0; q < LANES; ++q) s += red[q][cl];
        float* dst = c < split ? out0 + c : out1 + (c - split);
        *dst = accumulate ? *dst + s : s;
    }
}

static int ln_bwd_blocks(int64_t M) {
    int64_t b = (M + 3) / 4;
    if (b > 512) b = 512;
    return (int)b;
}

extern "C" size_t theia_layernorm_bwd_workspace_bytes(int64_t M, int D) {
    return (size_t)ln_bwd_blocks(M) * 2 * D * sizeof(float);
}

extern "C" int theia_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                                   const float* rstd, const void* dresid, void* dx, float* dgamma, float* dbeta,
                                   float* workspace, int64_t M, int D, int accumulate, int dtype, void* stream) {
    THEIA_CHECK_ARG(dy && x && gamma && mean && rstd && dx && dgamma && dbeta && workspace, "theia_layernorm_bwd: null pointer");
    THEIA_CHECK_ARG(M > 0 && D > 0 && D % 8 == 0 && D <= 8 * 64 * LN_MAXV, "theia_layernorm_bwd: unsupported D=%d", D);
    const int blocks = ln_bwd_blocks(M);
    const size_t lds = 4 * 2 * D * sizeof(float);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool res = dresid != nullptr;
    const theia_q8_out_t q8 = q8_take();
    const int nvl = D <= 512 ? 1 : D <= 1024 ? 2 : LN_MAXV;  // (see theia_layernorm_fwd)
    if (ln_two_rows(D)) {
        const size_t lds2 = 8 * 2 * D * sizeof(float);
#define LN_BWD2_LAUNCH(TT, RR, QQ)                                                                                                              \
    hipLaunchKernelGGL((ln_row_bwd2_kernel<TT, RR, QQ>), dim3(blocks), dim3(256), lds2, s, (const TT*)dy, (const TT*)x, gamma, mean, rstd, \
                       (const TT*)dresid, (TT*)dx, workspace, M, D, q8)
        if (dtype == THEIA_BF16 && q8.out != nullptr) {
            if (res) LN_BWD2_LAUNCH(bf16_t, true, true); else LN_BWD2_LAUNCH(bf16_t, false, true);
        } else if (dtype == THEIA_BF16) {
            if (res) LN_BWD2_LAUNCH(bf16_t, true, false); else LN_BWD2_LAUNCH(bf16_t, false, false);
        } else if (dtype == THEIA_F32) {
            if (res) LN_BWD2_LAUNCH(float, true, false); else LN_BWD2_LAUNCH(float, false, false);
        } else
            THEIA_CHECK_ARG(false, "theia_layernorm_bwd: bad dtype %d", dtype);
#undef LN_BWD2_LAUNCH
        THEIA_CHECK_LAUNCH("theia_layernorm_bwd");
        hipLaunchKernelGGL(partial_reduce_kernel<16>, dim3((2 * D + 15) / 16), dim3(256), 0, s, workspace, blocks, 2 * D,
                           (int64_t)2 * D, dgamma, dbeta, D, accumulate);
        THEIA_CHECK_LAUNCH("theia_layernorm_bwd(reduce)");
        return THEIA_OK;
    }
#define LN_BWD_LAUNCH(TT, NVV, RR)                                                                                                      \
    if (q8.out != nullptr && sizeof(TT) == 2)                                                                                          \
        hipLaunchKernelGGL((ln_row_bwd_kernel<bf16_t, NVV, RR, true>), dim3(blocks), dim3(256), lds, s, (const bf16_t*)dy, (const bf16_t*)x, gamma, mean, rstd, \
                           (const bf16_t*)dresid, (bf16_t*)dx, workspace, M, D, q8);                                                \
    else                                                                                                                               \
    hipLaunchKernelGGL((ln_row_bwd_kernel<TT, NVV, RR, false>), dim3(blocks), dim3(256), lds, s, (const TT*)dy, (const TT*)x, gamma, mean, rstd, \
                       (const TT*)dresid, (TT*)dx, workspace, M, D, q8)
#define LN_BWD_NV(TT, RR)                                                     \
    do {                                                                      \
        if (nvl == 1) LN_BWD_LAUNCH(TT, 1, RR);                               \
        else if (nvl == 2) LN_BWD_LAUNCH(TT, 2, RR);                          \
        else LN_BWD_LAUNCH(TT, LN_MAXV, RR);                                  \
    } while (0)
    if (dtype == THEIA_BF16) {
        if (res) LN_BWD_NV(bf16_t, true); else LN_BWD_NV(bf16_t, false);
    } else if (dtype == THEIA_F32) {
        if (res) LN_BWD_NV(float, true); else LN_BWD_NV(float, false);
    } else
        THEIA_CHECK_ARG(false, "theia_layernorm_bwd: bad dtype %d", dtype);
#undef LN_BWD_NV
#undef LN_BWD_LAUNCH
    THEIA_CHECK_LAUNCH("theia_layernorm_bwd");
    hipLaunchKernelGGL(partial_reduce_kernel<16>, dim3((2 * D + 15) / 16), dim3(256), 0, s, workspace, blocks, 2 * D,
                       (int64_t)2 * D, dgamma, dbeta, D, accumulate);
    THEIA_CHECK_LAUNCH("theia_layernorm_bwd(reduce)");
    return THEIA_OK;
}
extern "C" int theia_layernorm_bwd_q8(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                      const void* dresid, void* dx, float* dgamma, float* dbeta, float* workspace, int64_t M, int D,
                                      int accumulate, int dtype, const theia_q8_out_t* q8, void* stream) {
    Q8_FORWARD("theia_layernorm_bwd_q8", dtype, q8,
               theia_layernorm_bwd(dy, x, gamma, mean, rstd, dresid, dx, dgamma, dbeta, workspace, M, D, accumulate, dtype, stream));
}

// ================================================================================================
// whole-sample LayerNorm over E = H*W*C elements, affine [E] (already permuted to NHWC order)
// ================================================================================================
constexpr int CHW_CHUNK = 8192;  // elements per block in the statistics passes (256 thr x 8 x 4)

static int chw_chunks(int64_t E) { return (int)((E + CHW_CHUNK - 1) / CHW_CHUNK); }
static int chw_groups(int b, int64_t E) {
    // batch groups for the affine-gradient reduction: enough blocks to fill the chip, at most b
    const int64_t col_blocks = (E / 8 + 255) / 256;
    int64_t g = (1024 + col_blocks - 1) / col_blocks;
    if (g > b) g = b;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" size_t theia_layernorm_chw_workspace_bytes(int b, int64_t E) {
    const size_t stats = (((size_t)b * chw_chunks(E) * 2 + 63) / 64) * 64;  // floats: per-chunk partial sums
    const size_t dstat = (((size_t)b * 2 + 63) / 64) * 64;                    // floats: per-sample backward means
    const size_t parts = (size_t)chw_groups(b, E) * 3 * E;                   // floats: per-group affine-grad partials (+ dx sums)
    const size_t csum = (size_t)chw_groups(b, E) * E + 64;  // floats: column-sum partials of the dx sums (<= min(256, rows) x C <= ng x E)
    return (stats + dstat + parts + csum) * sizeof(float);
}

// partial (sum a, sum b) per (sample, chunk).  MODE 0: a = x, b = x*x.  MODE 1: a = dy*g, b = dy*g*xhat.
// A block owns one chunk of the element axis and walks over a GROUP of samples (blockIdx.y = group): the f32 affine row of the
// chunk (MODE 1) is loaded once per block and kept in registers -- one block per (chunk, sample) re-read 4 B of gamma for every
// 4 B of x and dy (PMC, round 2: 3.3x the algorithmic HBM reads on the LayerNorm[C,H,W] kernels).
template <typename T, int MODE>
__global__ __launch_bounds__(256) void chw_partial_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                          const float* __restrict__ gamma, const float* __restrict__ stats,
                                                          float* __restrict__ part, int64_t E, int nchunks, int b, int ngroups) {
    __shared__ float red[8];
    constexpr int IT = CHW_CHUNK / (256 * 8);
    const int chunk = blockIdx.x, grp = blockIdx.y;
    const int per = (b + ngroups - 1) / ngroups;
    const int s0 = grp * per, s1 = min(b, s0 + per);
    float g8[MODE == 1 ? IT : 1][8];
    if (MODE == 1) {
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int64_t e = (int64_t)chunk * CHW_CHUNK + (it * 256 + threadIdx.x) * 8;
            if (e < E) load8(gamma + e, g8[it]);
        }
    }
    for (int sample = s0; sample < s1; ++sample) {
        const int64_t base = (int64_t)sample * E;
        float mu = 0.f, rs = 0.f;
        if (MODE == 1) {
            mu = stats[2 * sample];
            rs = stats[2 * sample + 1];
        }
        float sa = 0.f, sb = 0.f;
        // every piece of the sample's chunk requested first, unconditionally (pieces past E: from the last valid vector, masked below) -- under
        // `if (e < E)` each piece was a branch with its own load -> s_waitcnt vmcnt(0) -> use (see ln_row_fwd_kernel)
        float xv[IT][8], dv[MODE == 1 ? IT : 1][8];
        bool okv[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int64_t e = (int64_t)chunk * CHW_CHUNK + (it * 256 + threadIdx.x) * 8;
            okv[it] = e < E;
            const int64_t ec = okv[it] ? e : E - 8;
            load8(x + base + ec, xv[it]);
            if (MODE == 1) load8(dy + base + ec, dv[MODE == 1 ? it : 0]);
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    sa += okv[it] ? xv[it][j] : 0.f;
                    sb += okv[it] ? xv[it][j] * xv[it][j] : 0.f;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float gy = dv[MODE == 1 ? it : 0][j] * g8[MODE == 1 ? it : 0][j];
                    sa += okv[it] ? gy : 0.f;
                    sb += okv[it] ? gy * (xv[it][j] - mu) * rs : 0.f;
                }
            }
        }
        sa = block_sum<256>(sa, red);
        sb = block_sum<256>(sb, red + 4);
        if (threadIdx.x == 0) {
            part[((int64_t)sample * nchunks + chunk) * 2] = sa;
            part[((int64_t)sample * nchunks + chunk) * 2 + 1] = sb;
        }
    }
}

// sample groups of the statistics / apply passes: enough blocks to fill the chip (>= ~2048 blocks), each walking over b / groups samples
static int chw_sample_groups(int b, int64_t col_blocks) {
    int64_t g = (2048 + col_blocks - 1) / col_blocks;
    if (g > b) g = b;
    if (g < 1) g = 1;
    return (int)g;
}

// MODE 0: stats[s] = (mean, rstd).  MODE 1: stats_out[s] = (mean(dy*g), mean(dy*g*xhat)).
template <int MODE>
__global__ __launch_bounds__(64) void chw_finalize_kernel(const float* __restrict__ part, float* __restrict__ out, int b, int nchunks, int64_t E,
                                                          float eps) {
    // one wave per sample: the lanes stride over the sample's chunk partials (f64 accumulation, a fixed order: lane-strided sums, then a
    // butterfly) -- one thread per sample walked up to 384 partials with one L2 latency each (49 us for the 64 x 64 maps)
    const int s = blockIdx.x, lane = threadIdx.x;
    if (s >= b) return;
    double a = 0.0, q = 0.0;
    for (int c = lane; c < nchunks; c += 64) {
        const float2 v = *reinterpret_cast<const float2*>(part + ((int64_t)s * nchunks + c) * 2);
        a += (double)v.x;
        q += (double)v.y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o, 64);
        q += __shfl_xor(q, o, 64);
    }
    if (lane != 0) return;
    if (MODE == 0) {
        const double mu = a / (double)E;
        double var = q / (double)E - mu * mu;
        if (var < 0.0) var = 0.0;
        out[2 * s] = (float)mu;
        out[2 * s + 1] = (float)(1.0 / sqrt(var + (double)eps));
    } else {
        out[2 * s] = (float)(a / (double)E);
        out[2 * s + 1] = (float)(q / (double)E);
    }
}

// SUMS: `stats` holds the per-sample (sum, sum of squares) produced by the convolution's epilogue; mean / rstd are derived here
// (and written to stats_out for the backward pass) instead of by two more passes over x.
template <typename T, bool SUMS = false, bool Q8 = false>
__global__ __launch_bounds__(256) void chw_apply_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ stats,
                                                        T* __restrict__ y, int64_t E, int b, int ngroups, float* __restrict__ stats_out = nullptr,
                                                        float eps = 0.f, const theia_q8_out_t q8 = {nullptr, nullptr, nullptr}) {
    float qsc = 0.f, qam = 0.f;
    if constexpr (Q8) qsc = *q8.scale;
    // a block owns 2048 elements of the affine row (f32 gamma / beta: 8 B per element, in registers) and walks over its group of
    // samples -- the round-2 kernel (one block per sample) re-read those 8 B for every 2 B of x
    const int grp = blockIdx.y;
    const int per = (b + ngroups - 1) / ngroups;
    const int s0 = grp * per, s1 = min(b, s0 + per);
    const int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    const bool live = e < E;
    float g8[8], b8[8];
    if (live) {
        load8(gamma + e, g8);
        load8(beta + e, b8);
    }
    __shared__ float2 st[256];  // SUMS: (mean, rstd) of up to 256 samples of the group, derived once per block (one sample per thread) --
    // as part of the sample loop every lane repeated the f64 division and square root for every sample: more VALU work than the 8 elements
    for (int sample = s0; sample < s1; ++sample) {
        float mu, rs;
        if constexpr (SUMS) {  // 2^-24 fixed-point 64-bit sums (GEMM epilogues)
            const int w = (sample - s0) & 255;
            if (w == 0) {
                __syncthreads();  // (the previous window has been read)
                const int mine = sample + (int)threadIdx.x;
                if (mine < s1) {
                    const long long* fx = reinterpret_cast<const long long*>(stats);
                    const double m = (double)fx[2 * mine] * (1.0 / 16777216.0) / (double)E;
                    double var = (double)fx[2 * mine + 1] * (1.0 / 16777216.0) / (double)E - m * m;
                    if (var < 0.0) var = 0.0;
                    const float2 v = make_float2((float)m, (float)(1.0 / sqrt(var + (double)eps)));
                    st[threadIdx.x] = v;
                    if (blockIdx.x == 0) {
                        stats_out[2 * mine] = v.x;
                        stats_out[2 * mine + 1] = v.y;
                    }
                }
                __syncthreads();
            }
            mu = st[w].x;
            rs = st[w].y;
        } else {
            mu = stats[2 * sample];
            rs = stats[2 * sample + 1];
        }
        if (!live) continue;
        float xv[8], o[8];
        load8(x + (int64_t)sample * E + e, xv);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (xv[j] - mu) * rs * g8[j] + b8[j];
        store8(y + (int64_t)sample * E + e, o);
        if constexpr (Q8) q8_store8(q8.out, (int64_t)sample * E + e, o, qsc, qam);
    }
    if constexpr (Q8) q8_flush_wave(q8.amax, qam);
}

extern "C" int theia_layernorm_chw_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats,
                                       float* workspace, int b, int64_t E, float eps, int dtype, void* stream) {
    THEIA_CHECK_ARG(x && gamma && beta && y && stats && workspace, "theia_layernorm_chw_fwd: null pointer");
    THEIA_CHECK_ARG(b > 0 && E > 0 && E % 8 == 0, "theia_layernorm_chw_fwd: E=%lld must be a positive multiple of 8", (long long)E);
    THEIA_CHECK_ARG(dtype == THEIA_F32 || dtype == THEIA_BF16, "theia_layernorm_chw_fwd: bad dtype");
    const int nch = chw_chunks(E);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == THEIA_BF16)
        hipLaunchKernelGGL((chw_partial_kernel<bf16_t, 0>), dim3(nch, b), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)nullptr, (const float*)nullptr, (const float*)nullptr, workspace, E, nch, b, b);
    else
        hipLaunchKernelGGL((chw_partial_kernel<float, 0>), dim3(nch, b), dim3(256), 0, s, (const float*)x, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, workspace, E, nch, b, b);
    THEIA_CHECK_LAUNCH("theia_layernorm_chw_fwd(partial)");
    hipLaunchKernelGGL(chw_finalize_kernel<0>, dim3(b), dim3(64), 0, s, workspace, stats, b, nch, E, eps);
    THEIA_CHECK_LAUNCH("theia_layernorm_chw_fwd(finalize)");
    const int64_t colb = (E / 8 + 255) / 256;
    const int ng = chw_sample_groups(b, colb);
    const dim3 grid((unsigned)colb, ng);
    if (dtype == THEIA_BF16)
        hipLaunchKernelGGL((chw_apply_kernel<bf16_t, false>), grid, dim3(256), 0, s, (const bf16_t*)x, gamma, beta, (const float*)stats, (bf16_t*)y, E, b, ng, (float*)nullptr, 0.f);
    else
        hipLaunchKernelGGL((chw_apply_kernel<float, false>), grid, dim3(256), 0, s, (const float*)x, gamma, beta, (const float*)stats, (float*)y, E, b, ng, (float*)nullptr, 0.f);
    THEIA_CHECK_LAUNCH("theia_layernorm_chw_fwd(apply)");
    return THEIA_OK;
}

extern "C" int theia_layernorm_chw_fwd_sums(const void* x, const float* gamma, const float* beta, void* y, const float* sums,
                                            float* stats, int b, int64_t E, float eps, int dtype, void* stream) {
    THEIA_CHECK_ARG(x && gamma && beta && y && sums && stats, "theia_layernorm_chw_fwd_sums: null pointer");
    THEIA_CHECK_ARG(b > 0 && E > 0 && E % 8 == 0, "theia_layernorm_chw_fwd_sums: E=%lld must be a positive multiple of 8", (long long)E);
    THEIA_CHECK_ARG(dtype == THEIA_F32 || dtype == THEIA_BF16, "theia_layernorm_chw_fwd_sums: bad dtype");
    const theia_q8_out_t q8 = q8_take();
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int64_t colb = (E / 8 + 255) / 256;
    const int ng = chw_sample_groups(b, colb);
    const dim3 grid((unsigned)colb, ng);
    if (dtype == THEIA_BF16 && q8.out != nullptr)
        hipLaunchKernelGGL((chw_apply_kernel<bf16_t, true, true>), grid, dim3(256), 0, s, (const bf16_t*)x, gamma, beta, sums, (bf16_t*)y, E, b, ng, stats, eps, q8);
    else if (dtype == THEIA_BF16)
        hipLaunchKernelGGL((chw_apply_kernel<bf16_t, true>), grid, dim3(256), 0, s, (const bf16_t*)x, gamma, beta, sums, (bf16_t*)y, E, b, ng, stats, eps);
    else
        hipLaunchKernelGGL((chw_apply_kernel<float, true>), grid, dim3(256), 0, s, (const float*)x, gamma, beta, sums, (float*)y, E, b, ng, stats, eps);
    THEIA_CHECK_LAUNCH("theia_layernorm_chw_fwd_sums");
    return THEIA_OK;
}

extern "C" int theia_layernorm_chw_fwd_sums_q8(const void* x, const float* gamma, const float* beta, void* y, const float* sums, float* stats,
                                               int b, int64_t E, float eps, int dtype, const theia_q8_out_t* q8, void* stream) {
    Q8_FORWARD("theia_layernorm_chw_fwd_sums_q8", dtype, q8, theia_layernorm_chw_fwd_sums(x, gamma, beta, y, sums, stats, b, E, eps, dtype, stream));
}

// dx for a group of samples + partial affine gradients of that group
// DXSUM: also the group's sum over samples of dx per element (part3[grp][E]) -- the producing convolution's bias gradient is its
// sum over pixels, so the 805 MB re-read of dx by a column-sum kernel (64x64 maps) becomes a 12.6 MB one
template <typename T, bool DXSUM, bool Q8 = false>
__global__ __launch_bounds__(256) void chw_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                      const float* __restrict__ gamma, const float* __restrict__ stats,
                                                      const float* __restrict__ dstat, T* __restrict__ dx,
                                                      float* __restrict__ part, float* __restrict__ part3, int b, int64_t E, int ngroups,
                                                      int relu_mask, const theia_q8_out_t q8) {
    // (Q8: a separate instantiation, see ln_row_fwd_kernel; every lane of a wave must reach the reduction of the maxima, so that form
    // has no early return)
    const int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    float qsc = 0.f, qam = 0.f;
    if constexpr (Q8) qsc = *q8.scale;
    if constexpr (!Q8) {
        if (e >= E) return;
    }
    if (!Q8 || e < E) {
        const int grp = blockIdx.y;
        const int per = (b + ngroups - 1) / ngroups;
        const int s0 = grp * per, s1 = min(b, s0 + per);
        float g8[8], ag[8], ab[8], ac[8];
        load8(gamma + e, g8);
#pragma unroll
        for (int j = 0; j < 8; ++j) ag[j] = ab[j] = ac[j] = 0.f;
        for (int s = s0; s < s1; ++s) {
            const float mu = stats[2 * s], rs = stats[2 * s + 1];
            const float m1 = dstat[2 * s], m2 = dstat[2 * s + 1];
            float xv[8], dv[8], o[8];
            load8(x + (int64_t)s * E + e, xv);
            load8(dy + (int64_t)s * E + e, dv);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xh = (xv[j] - mu) * rs;
                ag[j] += dv[j] * xh;
                ab[j] += dv[j];
                float d = rs * (dv[j] * g8[j] - m1 - xh * m2);
                if (relu_mask && !(xv[j] > 0.f)) d = 0.f;
                o[j] = d;
            }
            store8(dx + (int64_t)s * E + e, o);
            if constexpr (Q8) q8_store8(q8.out, (int64_t)s * E + e, o, qsc, qam);
            if constexpr (DXSUM) {
#pragma unroll
                for (int j = 0; j < 8; ++j) ac[j] += sizeof(T) == 2 ? bf16_to_f32(f32_to_bf16(o[j])) : o[j];  // the value as stored
            }
        }
        float* pg = part + (int64_t)grp * 2 * E;
        store8(pg + e, ag);
        store8(pg + E + e, ab);
        if constexpr (DXSUM) store8(part3 + (int64_t)grp * E + e, ac);
    }
    if constexpr (Q8) q8_flush_wave(q8.amax, qam);
}

extern "C" int theia_layernorm_chw_bwd_colsum(const void* dy, const void* x, const float* gamma, const float* stats, void* dx,
                                              float* dgamma, float* dbeta, float* workspace, int b, int64_t E, int relu_mask,
                                              int accumulate, float* dxsum, int C, int dxsum_accumulate, int dtype, void* stream);
extern "C" int theia_layernorm_chw_bwd_colsum_q8(const void* dy, const void* x, const float* gamma, const float* stats, void* dx, float* dgamma,
                                                 float* dbeta, float* workspace, int b, int64_t E, int relu_mask, int accumulate, float* dxsum,
                                                 int C, int dxsum_accumulate, int dtype, const theia_q8_out_t* q8, void* stream) {
    Q8_FORWARD("theia_layernorm_chw_bwd_colsum_q8", dtype, q8,
               theia_layernorm_chw_bwd_colsum(dy, x, gamma, stats, dx, dgamma, dbeta, workspace, b, E, relu_mask, accumulate, dxsum, C, dxsum_accumulate, dtype, stream));
}
extern "C" int theia_layernorm_chw_bwd(const void* dy, const void* x, const float* gamma, const float* stats, void* dx,
                                       float* dgamma, float* dbeta, float* workspace, int b, int64_t E, int relu_mask,
                                       int accumulate, int dtype, void* stream) {
    return theia_layernorm_chw_bwd_colsum(dy, x, gamma, stats, dx, dgamma, dbeta, workspace, b, E, relu_mask, accumulate, nullptr, 0, 0, dtype, stream);
}
extern "C" int theia_layernorm_chw_bwd_colsum(const void* dy, const void* x, const float* gamma, const float* stats, void* dx,
                                              float* dgamma, float* dbeta, float* workspace, int b, int64_t E, int relu_mask,
                                              int accumulate, float* dxsum, int C, int dxsum_accumulate, int dtype, void* stream) {
    THEIA_CHECK_ARG(dy && x && gamma && stats && dx && dgamma && dbeta && workspace, "theia_layernorm_chw_bwd: null pointer");
    THEIA_CHECK_ARG(dxsum == nullptr || (C >= 8 && C % 8 == 0 && E % C == 0), "theia_layernorm_chw_bwd_colsum: C=%d must be a multiple of 8 dividing E", C);
    THEIA_CHECK_ARG(b > 0 && E > 0 && E % 8 == 0, "theia_layernorm_chw_bwd: bad E");
    THEIA_CHECK_ARG(dtype == THEIA_F32 || dtype == THEIA_BF16, "theia_layernorm_chw_bwd: bad dtype");
    const int nch = chw_chunks(E);
    const int ng = chw_groups(b, E);
    const theia_q8_out_t q8 = q8_take();
    // workspace layout (see theia_layernorm_chw_workspace_bytes): [partial stats | dstat | affine partials]
    float* part_stats = workspace;
    float* dstat = part_stats + (((size_t)b * nch * 2 + 63) / 64) * 64;
    float* parts = dstat + (((size_t)b * 2 + 63) / 64) * 64;
    float* parts3 = parts + (size_t)ng * 2 * E;   // [ng][E]: per-group sums of dx over the group's samples
    float* csum_ws = parts3 + (size_t)ng * E;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int ngp = chw_sample_groups(b, nch);
    if (dtype == THEIA_BF16)
        hipLaunchKernelGGL((chw_partial_kernel<bf16_t, 1>), dim3(nch, ngp), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)dy, gamma, stats, part_stats, E, nch, b, ngp);
    else
        hipLaunchKernelGGL((chw_partial_kernel<float, 1>), dim3(nch, ngp), dim3(256), 0, s, (const float*)x, (const float*)dy, gamma, stats, part_stats, E, nch, b, ngp);
    THEIA_CHECK_LAUNCH("theia_layernorm_chw_bwd(partial)");
    hipLaunchKernelGGL(chw_finalize_kernel<1>, dim3(b), dim3(64), 0, s, part_stats, dstat, b, nch, E, 0.f);
    THEIA_CHECK_LAUNCH("theia_layernorm_chw_bwd(finalize)");
    const dim3 grid((unsigned)((E / 8 + 255) / 256), ng);
    if (dtype == THEIA_BF16 && q8.out != nullptr) {
        if (dxsum) hipLaunchKernelGGL((chw_bwd_kernel<bf16_t, true, true>), grid, dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x, gamma, stats, dstat, (bf16_t*)dx, parts, parts3, b, E, ng, relu_mask, q8);
        else hipLaunchKernelGGL((chw_bwd_kernel<bf16_t, false, true>), grid, dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x, gamma, stats, dstat, (bf16_t*)dx, parts, parts3, b, E, ng, relu_mask, q8);
    } else if (dtype == THEIA_BF16) {
        if (dxsum) hipLaunchKernelGGL((chw_bwd_kernel<bf16_t, true>), grid, dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x, gamma, stats, dstat, (bf16_t*)dx, parts, parts3, b, E, ng, relu_mask, q8);
        else hipLaunchKernelGGL((chw_bwd_kernel<bf16_t, false>), grid, dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x, gamma, stats, dstat, (bf16_t*)dx, parts, parts3, b, E, ng, relu_mask, q8);
    } else {
        if (dxsum) hipLaunchKernelGGL((chw_bwd_kernel<float, true>), grid, dim3(256), 0, s, (const float*)dy, (const float*)x, gamma, stats, dstat, (float*)dx, parts, parts3, b, E, ng, relu_mask, q8);
        else hipLaunchKernelGGL((chw_bwd_kernel<float, false>), grid, dim3(256), 0, s, (const float*)dy, (const float*)x, gamma, stats, dstat, (float*)dx, parts, parts3, b, E, ng, relu_mask, q8);
    }
    THEIA_CHECK_LAUNCH("theia_layernorm_chw_bwd(dx)");
    if (dxsum) {  // [ng * H*W][C] f32 -> [C]
        const int rc = theia_colsum(parts3, (int64_t)ng * (E / C), C, C, dxsum, csum_ws, dxsum_accumulate, THEIA_F32, stream);
        if (rc) return rc;
    }
    // reduce partials: columns [0,E) -> dgamma, [E,2E) -> dbeta
    const int64_t ncol = 2 * E;
    hipLaunchKernelGGL(partial_reduce_kernel<64>, dim3((unsigned)((ncol + 63) / 64)), dim3(256), 0, s, parts, ng, (int)ncol,
                       (int64_t)2 * E, dgamma, dbeta, (int)E, accumulate);
    THEIA_CHECK_LAUNCH("theia_layernorm_chw_bwd(reduce)");
    return THEIA_OK;
}
