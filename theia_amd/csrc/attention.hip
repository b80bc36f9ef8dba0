// K4: multi-head self-attention for the ViT student (transformers modeling_vit.py:164-189), head_dim = 64,
// n <= 256 tokens (197 for DeiT).  Round-1 kernels: whole K/V (resp. Q/dO) of one (image, head) live in LDS as
// f32 rows of pitch 68 (conflict-free ds_read_b128), one wave64 per query (resp. key) row, f32 VALU math with
// exact softmax -- attention is 1.6 % of the step's FLOPs; an MFMA version is the planned upgrade.
//
//   fwd    : per query row i:  s_j = q_i.k_j / 8 ; p = softmax(s) ; o_i = sum_j p_j v_j ; lse_i
//   bwd dq : per query row i:  p_j = exp(s_j - lse_i) ; dp_j = do_i.v_j ; ds_j = p_j (dp_j - delta_i) / 8 ;
//            dq_i = sum_j ds_j k_j ;  delta_i = do_i.o_i is stored for the second kernel
//   bwd dkv: per key row j  :  (same p, ds with roles swapped)  dv_j = sum_i p_ij do_i ; dk_j = sum_i ds_ij q_i
#include "common.h"

// bf16 matrix-core kernels (attention_mfma.hip); the kernels in this file are the exact-f32 path and the fallback
int theia_attention_fwd_mfma(const void* qkv, void* o, float* lse, int b, int n, int h, hipStream_t s);
int theia_attention_bwd_mfma(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv, float* delta,
                             int b, int n, int h, hipStream_t s);

constexpr int AT_DH = 64;
constexpr int AT_PITCH = 68;    // floats per LDS row
constexpr int AT_MAXN = 256;
constexpr int AT_THREADS = 512; // 8 waves
constexpr int AT_WAVES = AT_THREADS / 64;

template <typename T>
__device__ __forceinline__ void stage_rows(const T* __restrict__ src, int64_t row_stride, int n, float* __restrict__ dst) {
    // src row r at src + r*row_stride (64 contiguous elements) -> dst[r][0..63] f32
    for (int v = threadIdx.x; v < n * 8; v += AT_THREADS) {
        const int r = v >> 3, c = (v & 7) * 8;
        float x[8];
        load8(src + r * row_stride + c, x);
        float* q = dst + r * AT_PITCH + c;
        *reinterpret_cast<float4*>(q) = make_float4(x[0], x[1], x[2], x[3]);
        *reinterpret_cast<float4*>(q + 4) = make_float4(x[4], x[5], x[6], x[7]);
    }
}

// dot of the wave's broadcast row `vec` (64 floats in LDS) with LDS row `mat[r]`
__device__ __forceinline__ float dot64(const float* __restrict__ vec, const float* __restrict__ matrow) {
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < AT_DH; c += 4) {
        const float4 a = *reinterpret_cast<const float4*>(vec + c);
        const float4 b = *reinterpret_cast<const float4*>(matrow + c);
        acc += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
    return acc;
}

template <typename T>
__global__ __launch_bounds__(AT_THREADS) void attn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ o, float* __restrict__ lse,
                                                              int n, int h, int nsplit) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sK = sm;                           // [n][68]
    float* sV = sK + n * AT_PITCH;            // [n][68]
    float* sQ = sV + n * AT_PITCH;            // [waves][64]
    float* sP = sQ + AT_WAVES * AT_DH;        // [waves][256]
    const int bh = blockIdx.x, bi = bh / h, hi = bh % h;
    const int D = h * AT_DH;
    const int64_t rs = 3 * (int64_t)D;
    const T* base = qkv + (int64_t)bi * n * rs + hi * AT_DH;
    stage_rows(base + D, rs, n, sK);
    stage_rows(base + 2 * D, rs, n, sV);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* myQ = sQ + wave * AT_DH;
    float* myP = sP + wave * AT_MAXN;
    const float scale = 0.125f;
    for (int i = blockIdx.y * AT_WAVES + wave; i < n; i += nsplit * AT_WAVES) {
        myQ[lane] = Elem<T>::ld(base + i * rs + lane);
        __builtin_amdgcn_wave_barrier();
        float s[4];
        float mx = -INFINITY;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int j = ps * 64 + lane;
            s[ps] = -INFINITY;
            if (j < n) s[ps] = dot64(myQ, sK + j * AT_PITCH) * scale;
            mx = fmaxf(mx, s[ps]);
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int j = ps * 64 + lane;
            const float p = j < n ? expf(s[ps] - mx) : 0.f;
            myP[j] = p;
            sum += p;
        }
        sum = wave_sum(sum);
        __builtin_amdgcn_wave_barrier();
        float acc = 0.f;
        const int n4 = (n + 3) & ~3;
        for (int j = 0; j < n4; j += 4) {
            const float4 p4 = *reinterpret_cast<const float4*>(myP + j);
            acc += p4.x * sV[(j + 0) * AT_PITCH + lane];
            if (j + 1 < n) acc += p4.y * sV[(j + 1) * AT_PITCH + lane];
            if (j + 2 < n) acc += p4.z * sV[(j + 2) * AT_PITCH + lane];
            if (j + 3 < n) acc += p4.w * sV[(j + 3) * AT_PITCH + lane];
        }
        Elem<T>::st(o + ((int64_t)bi * n + i) * D + hi * AT_DH + lane, acc / sum);
        if (lane == 0) lse[(int64_t)bh * n + i] = mx + logf(sum);
        __builtin_amdgcn_wave_barrier();
    }
}

// dq kernel (LDS: K, V);  also writes delta
template <typename T>
__global__ __launch_bounds__(AT_THREADS) void attn_bwd_dq_kernel(const T* __restrict__ qkv, const T* __restrict__ o,
                                                                 const T* __restrict__ d_o, const float* __restrict__ lse,
                                                                 T* __restrict__ dqkv, float* __restrict__ delta, int n, int h,
                                                                 int nsplit) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sK = sm;
    float* sV = sK + n * AT_PITCH;
    float* sQ = sV + n * AT_PITCH;            // [waves][64]  q_i
    float* sG = sQ + AT_WAVES * AT_DH;        // [waves][64]  do_i
    float* sP = sG + AT_WAVES * AT_DH;        // [waves][256] ds
    const int bh = blockIdx.x, bi = bh / h, hi = bh % h;
    const int D = h * AT_DH;
    const int64_t rs = 3 * (int64_t)D;
    const T* base = qkv + (int64_t)bi * n * rs + hi * AT_DH;
    stage_rows(base + D, rs, n, sK);
    stage_rows(base + 2 * D, rs, n, sV);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* myQ = sQ + wave * AT_DH;
    float* myG = sG + wave * AT_DH;
    float* myP = sP + wave * AT_MAXN;
    const float scale = 0.125f;
    for (int i = blockIdx.y * AT_WAVES + wave; i < n; i += nsplit * AT_WAVES) {
        const int64_t orow = ((int64_t)bi * n + i) * D + hi * AT_DH + lane;
        const float gi = Elem<T>::ld(d_o + orow);
        myQ[lane] = Elem<T>::ld(base + i * rs + lane);
        myG[lane] = gi;
        const float dl = wave_sum(gi * Elem<T>::ld(o + orow));
        const float l = lse[(int64_t)bh * n + i];
        if (lane == 0) delta[(int64_t)bh * n + i] = dl;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int j = ps * 64 + lane;
            float ds = 0.f;
            if (j < n) {
                const float p = expf(dot64(myQ, sK + j * AT_PITCH) * scale - l);
                const float dp = dot64(myG, sV + j * AT_PITCH);
                ds = p * (dp - dl) * scale;
            }
            myP[j] = ds;
        }
        __builtin_amdgcn_wave_barrier();
        float acc = 0.f;
        const int n4 = (n + 3) & ~3;
        for (int j = 0; j < n4; j += 4) {
            const float4 p4 = *reinterpret_cast<const float4*>(myP + j);
            acc += p4.x * sK[(j + 0) * AT_PITCH + lane];
            if (j + 1 < n) acc += p4.y * sK[(j + 1) * AT_PITCH + lane];
            if (j + 2 < n) acc += p4.z * sK[(j + 2) * AT_PITCH + lane];
            if (j + 3 < n) acc += p4.w * sK[(j + 3) * AT_PITCH + lane];
        }
        Elem<T>::st(dqkv + ((int64_t)bi * n + i) * rs + hi * AT_DH + lane, acc);
        __builtin_amdgcn_wave_barrier();
    }
}

// dk/dv kernel (LDS: Q, dO); one wave per key row j, lanes over query rows
template <typename T>
__global__ __launch_bounds__(AT_THREADS) void attn_bwd_dkv_kernel(const T* __restrict__ qkv, const T* __restrict__ d_o,
                                                                  const float* __restrict__ lse, const float* __restrict__ delta,
                                                                  T* __restrict__ dqkv, int n, int h, int nsplit) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sQm = sm;                          // [n][68] Q
    float* sGm = sQm + n * AT_PITCH;          // [n][68] dO
    float* sKj = sGm + n * AT_PITCH;          // [waves][64] k_j
    float* sVj = sKj + AT_WAVES * AT_DH;      // [waves][64] v_j
    float* sP = sVj + AT_WAVES * AT_DH;       // [waves][256] p_ij
    float* sS = sP + AT_WAVES * AT_MAXN;      // [waves][256] ds_ij
    const int bh = blockIdx.x, bi = bh / h, hi = bh % h;
    const int D = h * AT_DH;
    const int64_t rs = 3 * (int64_t)D;
    const T* base = qkv + (int64_t)bi * n * rs + hi * AT_DH;
    stage_rows(base, rs, n, sQm);
    stage_rows(d_o + (int64_t)bi * n * D + hi * AT_DH, (int64_t)D, n, sGm);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* myK = sKj + wave * AT_DH;
    float* myV = sVj + wave * AT_DH;
    float* myP = sP + wave * AT_MAXN;
    float* myS = sS + wave * AT_MAXN;
    const float scale = 0.125f;
    for (int j = blockIdx.y * AT_WAVES + wave; j < n; j += nsplit * AT_WAVES) {
        myK[lane] = Elem<T>::ld(base + j * rs + D + lane);
        myV[lane] = Elem<T>::ld(base + j * rs + 2 * D + lane);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int i = ps * 64 + lane;
            float p = 0.f, ds = 0.f;
            if (i < n) {
                p = expf(dot64(myK, sQm + i * AT_PITCH) * scale - lse[(int64_t)bh * n + i]);
                const float dp = dot64(myV, sGm + i * AT_PITCH);
                ds = p * (dp - delta[(int64_t)bh * n + i]) * scale;
            }
            myP[i] = p;
            myS[i] = ds;
        }
        __builtin_amdgcn_wave_barrier();
        float av = 0.f, ak = 0.f;
        const int n4 = (n + 3) & ~3;
        for (int i = 0; i < n4; i += 4) {
            const float4 p4 = *reinterpret_cast<const float4*>(myP + i);
            const float4 s4 = *reinterpret_cast<const float4*>(myS + i);
            av += p4.x * sGm[(i + 0) * AT_PITCH + lane];
            ak += s4.x * sQm[(i + 0) * AT_PITCH + lane];
            if (i + 1 < n) { av += p4.y * sGm[(i + 1) * AT_PITCH + lane]; ak += s4.y * sQm[(i + 1) * AT_PITCH + lane]; }
            if (i + 2 < n) { av += p4.z * sGm[(i + 2) * AT_PITCH + lane]; ak += s4.z * sQm[(i + 2) * AT_PITCH + lane]; }
            if (i + 3 < n) { av += p4.w * sGm[(i + 3) * AT_PITCH + lane]; ak += s4.w * sQm[(i + 3) * AT_PITCH + lane]; }
        }
        T* drow = dqkv + ((int64_t)bi * n + j) * rs + hi * AT_DH + lane;
        Elem<T>::st(drow + D, ak);
        Elem<T>::st(drow + 2 * D, av);
        __builtin_amdgcn_wave_barrier();
    }
}

static int attn_nsplit(int bh, int n) {
    int s = (768 + bh - 1) / bh;
    const int smax = (n + AT_WAVES - 1) / AT_WAVES;
    if (s > smax) s = smax;
    if (s > 8) s = 8;
    if (s < 1) s = 1;
    return s;
}

extern "C" size_t theia_attention_bwd_workspace_bytes(int b, int n, int h) { return (size_t)b * n * h * sizeof(float); }

// raise a kernel's dynamic-LDS limit to the largest supported footprint, once per kernel (keyed on its address)
template <typename K>
static void set_lds(K kern, size_t) {
    static const void* seen[16];
    static int nseen = 0;
    const void* p = reinterpret_cast<const void*>(kern);
    for (int i = 0; i < nseen; ++i)
        if (seen[i] == p) return;
    const size_t max_lds = ((size_t)2 * AT_MAXN * AT_PITCH + AT_WAVES * (2 * AT_DH + 2 * AT_MAXN)) * sizeof(float);
    (void)hipFuncSetAttribute(p, hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds);
    if (nseen < 16) seen[nseen++] = p;
}

extern "C" int theia_attention_fwd(const void* qkv, void* o, float* lse, int b, int n, int h, int dtype, void* stream) {
    THEIA_CHECK_ARG(qkv && o && lse, "theia_attention_fwd: null pointer");
    THEIA_CHECK_ARG(b > 0 && h > 0 && n > 0 && n <= AT_MAXN, "theia_attention_fwd: n=%d must be in [1,%d]", n, AT_MAXN);
    THEIA_CHECK_ARG(dtype == THEIA_F32 || dtype == THEIA_BF16, "theia_attention_fwd: bad dtype");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == THEIA_BF16 && n <= 208) return theia_attention_fwd_mfma(qkv, o, lse, b, n, h, s);  // matrix-core path
    const size_t lds = ((size_t)2 * n * AT_PITCH + AT_WAVES * (AT_DH + AT_MAXN)) * sizeof(float);
    const int ns = attn_nsplit(b * h, n);
    if (dtype == THEIA_BF16) {
        set_lds(attn_fwd_kernel<bf16_t>, lds);
        hipLaunchKernelGGL(attn_fwd_kernel<bf16_t>, dim3(b * h, ns), dim3(AT_THREADS), lds, s, (const bf16_t*)qkv, (bf16_t*)o, lse, n, h, ns);
    } else {
        set_lds(attn_fwd_kernel<float>, lds);
        hipLaunchKernelGGL(attn_fwd_kernel<float>, dim3(b * h, ns), dim3(AT_THREADS), lds, s, (const float*)qkv, (float*)o, lse, n, h, ns);
    }
    THEIA_CHECK_LAUNCH("theia_attention_fwd");
    return THEIA_OK;
}

extern "C" int theia_attention_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv,
                                   float* delta_ws, int b, int n, int h, int dtype, void* stream) {
    THEIA_CHECK_ARG(qkv && o && d_o && lse && dqkv && delta_ws, "theia_attention_bwd: null pointer");
    THEIA_CHECK_ARG(b > 0 && h > 0 && n > 0 && n <= AT_MAXN, "theia_attention_bwd: n=%d must be in [1,%d]", n, AT_MAXN);
    THEIA_CHECK_ARG(dtype == THEIA_F32 || dtype == THEIA_BF16, "theia_attention_bwd: bad dtype");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == THEIA_BF16 && n <= 208) return theia_attention_bwd_mfma(qkv, o, d_o, lse, dqkv, delta_ws, b, n, h, s);
    const size_t lds1 =((size_t)2 * n * AT_PITCH + AT_WAVES * (2 * AT_DH + AT_MAXN)) * sizeof(float);
    const size_t lds2 = ((size_t)2 * n * AT_PITCH + AT_WAVES * (2 * AT_DH + 2 * AT_MAXN)) * sizeof(float);
    const int ns = attn_nsplit(b * h, n);
    if (dtype == THEIA_BF16) {
        set_lds(attn_bwd_dq_kernel<bf16_t>, lds1);
        set_lds(attn_bwd_dkv_kernel<bf16_t>, lds2);
        hipLaunchKernelGGL(attn_bwd_dq_kernel<bf16_t>, dim3(b * h, ns), dim3(AT_THREADS), lds1, s, (const bf16_t*)qkv, (const bf16_t*)o, (const bf16_t*)d_o, lse, (bf16_t*)dqkv, delta_ws, n, h, ns);
        THEIA_CHECK_LAUNCH("theia_attention_bwd(dq)");
        hipLaunchKernelGGL(attn_bwd_dkv_kernel<bf16_t>, dim3(b * h, ns), dim3(AT_THREADS), lds2, s, (const bf16_t*)qkv, (const bf16_t*)d_o, lse, delta_ws, (bf16_t*)dqkv, n, h, ns);
    } else {
        set_lds(attn_bwd_dq_kernel<float>, lds1);
        set_lds(attn_bwd_dkv_kernel<float>, lds2);
        hipLaunchKernelGGL(attn_bwd_dq_kernel<float>, dim3(b * h, ns), dim3(AT_THREADS), lds1, s, (const float*)qkv, (const float*)o, (const float*)d_o, lse, (float*)dqkv, delta_ws, n, h, ns);
        THEIA_CHECK_LAUNCH("theia_attention_bwd(dq)");
        hipLaunchKernelGGL(attn_bwd_dkv_kernel<float>, dim3(b * h, ns), dim3(AT_THREADS), lds2, s, (const float*)qkv, (const float*)d_o, lse, delta_ws, (float*)dqkv, n, h, ns);
    }
    THEIA_CHECK_LAUNCH("theia_attention_bwd(dkv)");
    return THEIA_OK;
}
