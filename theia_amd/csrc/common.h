// Shared device/host helpers for the theia_hip kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/theia_hip.h"

// ----------------------------------------------------------------------------------
// error plumbing: every C-ABI entry returns 0 or a negative code; message is thread-local
// ----------------------------------------------------------------------------------
void theia_set_error(const char* fmt, ...);

#define THEIA_CHECK_ARG(cond, ...)                 \
    do {                                           \
        if (!(cond)) {                             \
            theia_set_error(__VA_ARGS__);          \
            return THEIA_ERR_INVALID;              \
        }                                          \
    } while (0)

#define THEIA_CHECK_LAUNCH(name)                                                   \
    do {                                                                           \
        hipError_t e__ = hipGetLastError();                                        \
        if (e__ != hipSuccess) {                                                   \
            theia_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return THEIA_ERR_LAUNCH;                                               \
        }                                                                          \
    } while (0)

// ----------------------------------------------------------------------------------
// element types.  bf16 is carried as uint16_t bit patterns; conversions are explicit RNE.
// ----------------------------------------------------------------------------------
typedef uint16_t bf16_t;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// f32 -> bf16, round-to-nearest-even: gfx950's v_cvt_pk_bf16_f32 (one instruction per PAIR; the bit-twiddling form it replaces
// was ~6 VALU instructions per element and made the attention kernels and every bf16 epilogue VALU-bound)
typedef __attribute__((ext_vector_type(2))) __bf16 theia_bf16x2_v;
typedef __attribute__((ext_vector_type(2))) float theia_f32x2_v;
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((theia_f32x2_v){lo, hi}, theia_bf16x2_v));
}

// fp8 (OCP e4m3) operand bytes of the THEIA_FP8 GEMM path: a distinct 1-byte type so that templates can tell it from uint8 pixels
struct fp8_t { uint8_t bits; };
static_assert(sizeof(fp8_t) == 1, "fp8_t is one byte");

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int kPer16B = 4;
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int kPer16B = 8;
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// 8 consecutive elements <-> 8 floats (vector global access: 16 B for bf16, 2x16 B for f32)
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
    const uint4 a = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(w[i] << 16);
        v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
    *reinterpret_cast<uint4*>(p) = make_uint4(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7]));
}

// ----------------------------------------------------------------------------------
// wave64 / block reductions
// ----------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide sum for blockDim.x == NT (multiple of 64); `red` is NT/64 floats of LDS. All threads get the sum.
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) s += red[i];
    return s;
}

// ----------------------------------------------------------------------------------
// e4m3 copy of a pass's main output (theia_q8_out_t, the *_q8 entry points): 8 values at element offset `off`, quantised from their
// bf16-rounded form with the slot's scale; `am` collects max |value| (NaN -> +Inf, as theia_quantize_fp8 does)
// ----------------------------------------------------------------------------------
__device__ __forceinline__ void q8_store8(uint8_t* __restrict__ out, int64_t off, const float (&v)[8], float sc, float& am) {
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float r = bf16_to_f32(f32_to_bf16(v[j]));
        const bool nan = r != r;
        am = nan ? INFINITY : fmaxf(am, fabsf(r));
        x[j] = nan ? r : fminf(fmaxf(r * sc, -448.f), 448.f);
    }
    uint32_t lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(x[0], x[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(x[2], x[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(x[4], x[5], hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(x[6], x[7], hi, true);
    *reinterpret_cast<uint2*>(out + off) = make_uint2(lo, hi);
}
// one wave's maximum into the slot.  Device-scope atomics are performed at the memory side and serialise per address (~11 ns each):
// the atomic is issued only when the wave's maximum exceeds what the slot already holds (an agent-scope load: not a stale line of this
// XCD's L2), which after the first few waves of a launch is almost never
__device__ __forceinline__ void q8_flush_wave(float* amax, float am) {
    if (amax == nullptr) return;  // (uniform: the engine's fused passes run without a maximum, see engine.Fp8Scales.fused)
    am = wave_max(am);
    if ((threadIdx.x & 63) == 0) {
        const float cur = __hip_atomic_load(amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!(am <= cur)) atomicMax(reinterpret_cast<unsigned int*>(amax), __float_as_uint(am));  // non-negative floats order like their bits
    }
}

// The *_q8 entry points arm this for the plain entry point they forward to (same thread, next launch); the plain one takes it.
static thread_local theia_q8_out_t g_q8_armed = {nullptr, nullptr, nullptr};
static theia_q8_out_t q8_take() {
    const theia_q8_out_t q = g_q8_armed;
    g_q8_armed = {nullptr, nullptr, nullptr};
    return q;
}
#define Q8_FORWARD(who, dtype, q8, CALL)                                                                        \
    do {                                                                                                        \
        if ((q8) != nullptr && (q8)->out != nullptr) {                                                          \
            THEIA_CHECK_ARG((dtype) == THEIA_BF16 && (q8)->scale != nullptr, who ": e4m3 output: bf16 passes, with a scale"); \
            g_q8_armed = *(q8);                                                                                 \
        }                                                                                                       \
        const int rc_ = (CALL);                                                                                 \
        g_q8_armed = {nullptr, nullptr, nullptr};                                                               \
        return rc_;                                                                                             \
    } while (0)

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

static inline int cdiv_i(long a, long b) { return (int)((a + b - 1) / b); }
int theia_compute_cus();  // CU budget of the GEMM planners (misc.hip: theia_set_compute_cus)
