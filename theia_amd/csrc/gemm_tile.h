// Shared pieces of the MFMA GEMM kernels: MFMA wrappers, XCD-aware block remap, and the vectorised epilogue.
#pragma once
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 gt_bf16x8;
typedef __attribute__((ext_vector_type(4))) float gt_f32x4;

template <typename T> struct GtMma;
template <> struct GtMma<bf16_t> {
    __device__ static __forceinline__ void run(gt_f32x4& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(gt_bf16x8, a), __builtin_bit_cast(gt_bf16x8, b), acc, 0, 0, 0);
    }
};
template <> struct GtMma<float> {
    __device__ static __forceinline__ void run(gt_f32x4& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
    }
};

// hardware places block b on XCD b % 8; give each XCD a contiguous range of tiles (bijective for any grid size)
__device__ __forceinline__ int gt_xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + idx;
}

// fast exact-erf GELU pieces for the bf16 path: Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, far below bf16 resolution);
// the f32 (parity) path keeps libm erff.
__device__ __forceinline__ float gt_erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __frcp_rn(1.0f + 0.3275911f * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float r = 1.0f - poly * __expf(-ax * ax);
    return copysignf(r, x);
}
template <typename T> __device__ __forceinline__ float gt_gelu(float x) {
    if constexpr (sizeof(T) == 2) return 0.5f * x * (1.0f + gt_erf_fast(x * 0.70710678118654752440f));
    else return gelu_erf(x);
}
template <typename T> __device__ __forceinline__ float gt_gelu_grad(float x) {
    if constexpr (sizeof(T) == 2) {
        const float cdf = 0.5f * (1.0f + gt_erf_fast(x * 0.70710678118654752440f));
        return cdf + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
    } else {
        return gelu_erf_grad(x);
    }
}

// Epilogue of one wave's WM x WN accumulator tile.  acc[i][j] is the 16x16 fragment for n-frag i / m-frag j produced
// with the WEIGHT fragment as the MFMA A operand (lane: m = j*16 + lane&15, n = i*16 + (lane>>4)*4 .. +4).
// The tile goes through a wave-private LDS region (EPH rows at a time) and is re-read row-contiguously so that every
// global access of the epilogue (bias, row table, residual, aux, output) is a 16-byte vector access.
template <typename T, int WM, int WN>
__device__ __forceinline__ void gt_epilogue(gt_f32x4 (&acc)[WN / 16][WM / 16], float* ep /* wave-private LDS */,
                                            const theia_gemm_args_t& p, int m_wave0, int n_wave0, int lane) {
    constexpr int FM = WM / 16, FN = WN / 16;
    constexpr int EP_PITCH = WN + 4;
    constexpr int EPH = WM > 64 ? 64 : WM;
    constexpr int LPR = WN / 8, RPP = 64 / LPR;
    const theia_rowmap_t& mp = p.map;
    const int frow = lane & 15, fg = lane >> 4;
    const int R = mp.rows_h * mp.rows_w;
    T* __restrict__ O = reinterpret_cast<T*>(p.out);
    const T* __restrict__ RES = reinterpret_cast<const T*>(p.resid);
    const T* __restrict__ AUXI = reinterpret_cast<const T*>(p.aux_in);
    T* __restrict__ AUXO = reinterpret_cast<T*>(p.aux_out);
    const int col = (lane % LPR) * 8;
    const int n = n_wave0 + col;
    float bias8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bias8[j] = 0.f;
    if (p.bias != nullptr && n < p.N) load8(p.bias + n, bias8);
#pragma unroll
    for (int hf = 0; hf < WM / EPH; ++hf) {
        // (1) issue every global read of this pass group first (residual / aux / row table): their latency then overlaps
        //     the LDS round trip instead of being paid once per row pass
        constexpr int NPS = EPH / RPP;
        int64_t off[NPS];
        bool live[NPS];
        uint4 rpre[NPS];  // prefetched aux_in (ACT_MUL_*) or, otherwise, residual rows (bf16 path)
        const bool want_aux = p.act == THEIA_ACT_MUL_DGELU || p.act == THEIA_ACT_MUL_DRELU;
        const T* __restrict__ PRE = want_aux ? AUXI : RES;
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) {
            const int row = ps * RPP + lane / LPR;
            const int m = m_wave0 + hf * EPH + row;
            live[ps] = (m < p.M) && (n < p.N);
            const int mm = live[ps] ? m : 0;
            const int img = mm / R, rem = mm - img * R;
            const int ry = rem / mp.rows_w, rx = rem - ry * mp.rows_w;
            off[ps] = (int64_t)img * mp.out_batch_stride + mp.out_offset +
                      (int64_t)((ry * mp.out_sy + mp.out_y0) * mp.out_w + rx * mp.out_sx + mp.out_x0) * p.ldo + (live[ps] ? n : 0);
            rpre[ps] = make_uint4(0, 0, 0, 0);
            if (sizeof(T) == 2 && PRE != nullptr && live[ps]) rpre[ps] = *reinterpret_cast<const uint4*>(PRE + off[ps]);
        }
        // (2) accumulators -> wave-private LDS tile
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int jj = 0; jj < EPH / 16; ++jj) {
                const int j = hf * (EPH / 16) + jj;
                float* q = ep + (jj * 16 + frow) * EP_PITCH + i * 16 + fg * 4;
                *reinterpret_cast<float4*>(q) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
        // (3) row-contiguous passes
#pragma unroll
        for (int ps = 0; ps < NPS; ++ps) {
            const int row = ps * RPP + lane / LPR;
            if (!live[ps]) continue;
            const int64_t o = off[ps];
            float v[8];
            load8(ep + row * EP_PITCH + col, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += bias8[j];
            if (p.rowtab != nullptr) {
                const int m = m_wave0 + hf * EPH + row;
                float t8[8];
                load8(p.rowtab + (int64_t)(m % p.rowtab_period) * p.N + n, t8);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += t8[j];
            }
            float a8[8];
            if (want_aux) {
                if (sizeof(T) == 2) {
                    const uint32_t w4[4] = {rpre[ps].x, rpre[ps].y, rpre[ps].z, rpre[ps].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        a8[2 * j] = __uint_as_float(w4[j] << 16);
                        a8[2 * j + 1] = __uint_as_float(w4[j] & 0xffff0000u);
                    }
                } else {
                    load8(AUXI + o, a8);
                }
            }
            if (p.act == THEIA_ACT_GELU) {
                if (AUXO != nullptr) store8(AUXO + o, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = gt_gelu<T>(v[j]);
            } else if (p.act == THEIA_ACT_RELU) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
            } else if (p.act == THEIA_ACT_MUL_DGELU) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] *= gt_gelu_grad<T>(a8[j]);
            } else if (p.act == THEIA_ACT_MUL_DRELU) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = a8[j] > 0.f ? v[j] : 0.f;
            }
            if (RES != nullptr) {
                float r8[8];
                if (sizeof(T) == 2 && !want_aux) {
                    const uint32_t w4[4] = {rpre[ps].x, rpre[ps].y, rpre[ps].z, rpre[ps].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        r8[2 * j] = __uint_as_float(w4[j] << 16);
                        r8[2 * j + 1] = __uint_as_float(w4[j] & 0xffff0000u);
                    }
                } else {
                    load8(RES + o, r8);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += r8[j];
            }
            store8(O + o, v);
        }
        __builtin_amdgcn_wave_barrier();  // the next pass reuses the wave's LDS region
    }
}
