// Shared pieces of the MFMA GEMM kernels: MFMA wrappers, XCD-aware block remap, and the vectorised epilogue.
#pragma once
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 gt_bf16x8;
typedef __attribute__((ext_vector_type(4))) float gt_f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int gt_u32x4;

// 8 elements per lane that the epilogue's dead lanes (rows >= M, columns >= N) store to instead of branching around the store
__device__ __attribute__((aligned(16))) static char g_gt_dump[64 * 32];

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

// the same on fragments held as 4 x u32 vectors (what the inline-asm LDS reads below produce)
template <typename T> __device__ __forceinline__ void gt_mma(gt_f32x4& acc, const gt_u32x4& a, const gt_u32x4& b) {
    if constexpr (sizeof(T) == 2) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(gt_bf16x8, a), __builtin_bit_cast(gt_bf16x8, b), acc, 0, 0, 0);
    } else if constexpr (sizeof(T) == 1) {
        // fp8 e4m3: a 16-byte chunk = two k-steps of 8 bytes per lane; both operands split their chunk the same way, so the
        // contraction is over the same 16 k whatever the instruction's internal k order
        const long a0 = (long)(((unsigned long)a[1] << 32) | a[0]), a1 = (long)(((unsigned long)a[3] << 32) | a[2]);
        const long b0 = (long)(((unsigned long)b[1] << 32) | b[0]), b1 = (long)(((unsigned long)b[3] << 32) | b[2]);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a0, b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a1, b1, acc, 0, 0, 0);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[k]), __uint_as_float(b[k]), acc, 0, 0, 0);
    }
}

// ds_read_b128 as inline asm.  A plain C++ LDS load in a loop that also issues LDS-DMA (global_load_lds) makes hipcc put
// `s_waitcnt vmcnt(0)` in front of the first load of every iteration (it cannot tell the ring slot being read from the slot being
// filled), which drains the whole prefetch pipeline once per k-step: measured as a kernel bound by the LATENCY of one operand
// tile.  The asm form is invisible to that bookkeeping; the kernels order it themselves with counted vmcnt waits + barriers, and
// tie the destination registers to their own `s_waitcnt lgkmcnt(0)` (gt_wait_lds) before the MFMAs read them.
template <int OFF> __device__ __forceinline__ void gt_ds_read128(gt_u32x4& dst, uint32_t lds_addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_addr), "n"(OFF));
}
__device__ __forceinline__ void gt_wait_lds(gt_u32x4 (&b)[4], gt_u32x4 (&a)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]),
                   "+v"(a[6]), "+v"(a[7])::"memory");
}
// LDS byte address of a pointer into the workgroup's dynamic shared memory
__device__ __forceinline__ uint32_t gt_lds_addr(const void* p) { return (uint32_t)reinterpret_cast<uintptr_t>(p); }

// hardware places block b on XCD b % 8; give each XCD a contiguous range of tiles (bijective for any grid size)
__device__ __forceinline__ int gt_xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + idx;
}

// floor(m / d) and remainder for 0 <= m, d >= 1 with rcp = 1.0f / d hoisted by the caller: exact below 2^24 (the float
// quotient is within one of the answer, two integer fix-ups), plain division above.  The row maps decode a GEMM row into
// (image, y, x) with two of these per row; the hardware integer division they replace is ~40 VALU instructions each.
__device__ __forceinline__ int gt_divmod(int m, int d, float rcp, int& rem) {
    if (m >= (1 << 24)) {
        const int q = m / d;
        rem = m - q * d;
        return q;
    }
    int q = (int)((float)m * rcp);
    int r = m - q * d;
    if (r < 0) { --q; r += d; }
    if (r >= d) { ++q; r -= d; }
    rem = r;
    return q;
}

// the same without the large-m branch (callers guarantee m < 2^24): no divergent control flow in the kernels' address set-up
__device__ __forceinline__ int gt_divmod24(int m, int d, float rcp, int& rem) {
    int q = (int)((float)m * rcp);
    int r = m - q * d;
    if (r < 0) { --q; r += d; }
    if (r >= d) { ++q; r -= d; }
    rem = r;
    return q;
}

// GELU pieces of the bf16 path.  The f32 (parity) path keeps libm erff.
// Phi(x) = 0.5 * (1 + erf(x / sqrt 2)) as a logistic function of an odd quintic (round 5; minimax fit on [-4.5, 4.5] with the
// saturation below included):   Phi(x) ~ 1 / (1 + exp(-(a x + b x^3 + c x^5))),  xc = clamp(x, +-4.5)
//     |x| * |error| <= 6.5e-5 over the whole real line, i.e. gelu(x) = x * Phi(x) is within 6.5e-5 ABSOLUTE of the erf form (bf16
//     resolves 3.9e-3 relative; the degree-9 polynomial in x^2 used in rounds 2-4 was within 1.25e-5 * |x| <= 5.6e-5).
// 10 VALU operations, two of them transcendental (v_exp_f32, v_rcp_f32: quarter rate) instead of 17: the GELU / GELU' epilogues of fc1
// forward / fc2 data-gradient are VALU-bound (~9k cycles of a 256 x 256 tile, profiles/r04_epilogue_lane_order.txt; the same ~10 us per
// tile in the one-wave-per-SIMD experiment of round 5).  History: Abramowitz-Stegun 7.1.26 (rcp + exp + 14 plain, round 1: ~15k cycles
// per tile), the clamped degree-9 polynomial (17 plain, rounds 2-4).
// Saturating form: stretching by (1 + 2 delta) around 1/2 with delta = 1.7e-5 and clamping to [0, 1] pins both tails to exactly 0 / 1
// (gelu(x) = 0 for x <= -4.5, = x for x >= 4.5: Phi(-4.5) = 3.4e-6 plus the fit error stay below delta).
__device__ __forceinline__ float gt_phi_sat(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -4.5f, 4.5f);
    const float s = xc * xc;
    // -(a + b s + c s^2) * log2(e): a = 1.594730384, b = 0.07449136885, c = -0.0008378073912
    float t = fmaf(1.2087006e-03f, s, -1.0746833e-01f);
    t = fmaf(t, s, -2.3007097e+00f);
    const float e = __builtin_amdgcn_exp2f(t * xc);
    const float r = __builtin_amdgcn_rcpf(1.0f + e);
    return __builtin_amdgcn_fmed3f(fmaf(r, 1.000034f, -1.7e-5f), 0.f, 1.f);
}
template <typename T> __device__ __forceinline__ float gt_gelu(float x) {
    if constexpr (sizeof(T) == 2) return x * gt_phi_sat(x);
    else return gelu_erf(x);
}
template <typename T> __device__ __forceinline__ float gt_gelu_grad(float x) {
    if constexpr (sizeof(T) == 2) {  // Phi(x) + x * phi(x), phi = exp(-x^2 / 2) / sqrt(2 pi) through one v_exp_f32
        const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);
        return fmaf(x * e, 0.39894228040143267794f, gt_phi_sat(x));
    } else {
        return gelu_erf_grad(x);
    }
}

// cycle stamps of block 0's epilogue for tools/pp_trace.hip (-DPP_TRACE); compiled out of the library
#ifdef PP_TRACE
__device__ unsigned long long g_ep_trace[8][16];
#define GT_EP_STAMP(k) \
    if (blockIdx.x == 0 && lane == 0) g_ep_trace[threadIdx.x >> 6][k] = __builtin_readcyclecounter();
#else
#define GT_EP_STAMP(k)
#endif

// Epilogue of one wave's WM x WN accumulator tile.  acc[i][j] is the 16x16 fragment for n-frag i / m-frag j produced
// with the WEIGHT fragment as the MFMA A operand (lane: m = j*16 + lane&15, n = i*16 + (lane>>4)*4 .. +4).
// The tile goes through a wave-private LDS region (EPH rows at a time) and is re-read row-contiguously so that every
// global access of the epilogue (bias, row table, residual, aux, output) is a 16-byte vector access.
// The row passes are a ROLLED loop over groups of passes: fully unrolled the epilogue was ~70 KB of straight-line code
// executed once per workgroup -- more than the 64 KB instruction cache, and the instruction fetch (not the stores) set its
// duration (29-36k cycles per 256x256 tile; tools/pp_trace.hip).
// SUMS: also accumulate p.ln_sums (a separate instantiation, launched only for the GEMMs that feed a whole-sample LayerNorm, so
// that every other launch pays nothing for it)
// SCALE: multiply the accumulators by (*p.a_scale_inv) * (*p.w_scale_inv) first (fp8 operands: T is then the OUTPUT type, bf16)
// PREF: 1 = the launch has a residual / aux_in row to prefetch (bf16), 0 = it has not (callers branch once, wave-uniformly, between
// the two instantiations: without a prefetch the loop carries no vector-memory wait and no prefetch registers), -1 = decide at
// run time inside whether to prefetch and wait at the end of every group regardless (the smaller kernels)
constexpr int GT_SUMS_SLOTS = 4;  // images a tile of <= 320 rows can touch when an image has >= 160 rows: 3 (+1: a wave's empty second image)
// One wave, after a workgroup barrier that follows every wave's epilogue of the tile: add the table to the per-image totals and clear it.
__device__ __forceinline__ void gt_flush_sums(unsigned long long* sums_tab, unsigned long long* lsum, int img_tile0, int R, int M, int lane) {
    if (lane < 2 * GT_SUMS_SLOTS) {
        const unsigned long long v = sums_tab[lane];
        sums_tab[lane] = 0ull;
        const int img = img_tile0 + (lane >> 1);
        if (v != 0ull && (int64_t)img * R < M) atomicAdd(lsum + 2 * img + (lane & 1), v);
    }
}

template <typename T, int WM, int WN, bool SUMS = false, bool SCALE = false, int GPMAX = 4, int PREF = -1>
__device__ __forceinline__ void gt_epilogue(gt_f32x4 (&acc)[WN / 16][WM / 16], float* ep /* wave-private LDS */,
                                            const theia_gemm_args_t& p, int m_wave0, int n_wave0, int lane, unsigned long long* sums_tab = nullptr, int img_tile0 = 0) {
    constexpr int FM = WM / 16, FN = WN / 16;
    constexpr int EP_PITCH = WN + 4;
    constexpr int EPH = WM > 64 ? 64 : WM;
    constexpr int LPR = WN / 8, RPP = 64 / LPR;
    constexpr int NPS = EPH / RPP;
    const theia_rowmap_t& mp = p.map;
    const int frow = lane & 15, fg = lane >> 4;
    const int R = mp.rows_h * mp.rows_w;
    const float rcp_R = 1.0f / (float)R, rcp_w = 1.0f / (float)mp.rows_w;
    T* __restrict__ O = reinterpret_cast<T*>(p.out);
    const T* __restrict__ RES = reinterpret_cast<const T*>(p.resid);
    const T* __restrict__ AUXI = reinterpret_cast<const T*>(p.aux_in);
    T* __restrict__ AUXO = reinterpret_cast<T*>(p.aux_out);
    const int col = (lane % LPR) * 8, lrow = lane / LPR;
    const int n = n_wave0 + col;
    const bool n_ok = n < p.N;
    float bias8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bias8[j] = 0.f;
    if (p.bias != nullptr && n_ok) load8(p.bias + n, bias8);
    const int act = p.act;
    float alpha = 1.0f;
    if constexpr (SCALE) alpha = (p.a_scale_inv != nullptr ? *p.a_scale_inv : 1.0f) * (p.w_scale_inv != nullptr ? *p.w_scale_inv : 1.0f);
    const bool want_aux = act == THEIA_ACT_MUL_DGELU || act == THEIA_ACT_MUL_DRELU;
    const T* __restrict__ PRE = want_aux ? AUXI : RES;  // the row that is prefetched (bf16 path)
    const bool pre_on = PREF == 0 ? false : (sizeof(T) == 2 && PRE != nullptr);
    // Output element offset of this lane's row in every pass.  The epilogue is bound by VALU issue (a wave64 instruction holds the
    // SIMD for 4 cycles and two waves share it: ~700 cycles per pass when decoding each row with two divisions and 64-bit
    // products, tools/pp_trace.hip), so the row (image, y, x) is decoded ONCE and then stepped by the RPP rows between passes:
    //   plain maps (one row per "image"):  offset += RPP * out_batch_stride
    //   image maps whose step wraps at most once in x and once in y (RPP / rows_w + 1 <= rows_h):  offset += d_step, + d_wx when
    //   x leaves the row, + d_wy when y leaves the image
    //   anything else: decode again.
    auto decode_row = [&](int m, int& ry, int& rx) -> int64_t {
        int rem;
        const int img = gt_divmod(m, R, rcp_R, rem);
        ry = gt_divmod(rem, mp.rows_w, rcp_w, rx);
        return (int64_t)img * mp.out_batch_stride + mp.out_offset +
               (int64_t)((ry * mp.out_sy + mp.out_y0) * mp.out_w + rx * mp.out_sx + mp.out_x0) * p.ldo;
    };
    const int step_qw = RPP / mp.rows_w, step_rw = RPP - step_qw * mp.rows_w;  // wave-uniform
    const int step_mode = R == 1 ? 0 : (step_qw + 1 <= mp.rows_h ? 1 : 2);
    const int64_t row_pitch = (int64_t)mp.out_sy * mp.out_w * p.ldo;             // offset of one map row down
    const int64_t d_step = step_mode == 0 ? (int64_t)RPP * mp.out_batch_stride : step_qw * row_pitch + (int64_t)step_rw * mp.out_sx * p.ldo;
    const int64_t d_wx = row_pitch - (int64_t)mp.rows_w * mp.out_sx * p.ldo;
    const int64_t d_wy = mp.out_batch_stride - mp.rows_h * row_pitch;
    int dead_ry, dead_rx;
    const int64_t off_dead = decode_row(0, dead_ry, dead_rx);  // where dead lanes (rows >= M, columns >= N) prefetch from: row 0, column 0
    struct row_cursor_t { int m, ry, rx; int64_t off; };
    auto first_row = [&]() {
        row_cursor_t c;
        c.m = m_wave0 + lrow;
        c.off = decode_row(c.m, c.ry, c.rx) + n;
        return c;
    };
    auto next_row = [&](row_cursor_t& c) {  // advance this lane's row by RPP
        c.m += RPP;
        if (step_mode == 2) {
            c.off = decode_row(c.m, c.ry, c.rx) + n;
            return;
        }
        c.off += d_step;
        if (step_mode == 1) {
            c.rx += step_rw;
            c.ry += step_qw;
            const bool wx = c.rx >= mp.rows_w;
            c.rx -= wx ? mp.rows_w : 0;
            c.ry += wx ? 1 : 0;
            c.off += wx ? d_wx : 0;
            const bool wy = c.ry >= mp.rows_h;
            c.ry -= wy ? mp.rows_h : 0;
            c.off += wy ? d_wy : 0;
        }
    };
    row_cursor_t cur = first_row();
    auto unpack8 = [](const gt_u32x4& u, float (&f)[8]) {
        const uint32_t w4[4] = {u[0], u[1], u[2], u[3]};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f[2 * j] = __uint_as_float(w4[j] << 16);
            f[2 * j + 1] = __uint_as_float(w4[j] & 0xffff0000u);
        }
    };
    // Passes are processed in groups of GP.  The residual / aux_in rows (bf16) of ALL passes are requested before the first store
    // of the tile and waited for once with vmcnt(0): loads and stores share the vmcnt counter but do not retire in order with
    // respect to each other, so a counted wait cannot tell a landed load from an acknowledged store (an earlier version requested
    // group g+1's rows ahead of group g's stores and waited with vmcnt(GP): wrong rows whenever the stores were acknowledged
    // first -- caught by ln_sums + resid on the 128x128 kernel), and a load issued behind stores waits for their acknowledgements
    // (~1.7-4k cycles under load).  Order: rows of the first LDS half -> stage that half's accumulators (their registers are then
    // free) -> rows of the second half -> one wait.  The loop below has no vector-memory wait at all; the stores just stream out.
    constexpr int GP = NPS < GPMAX ? NPS : GPMAX;      // passes per group (unrolled: static registers for the prefetched rows)
    constexpr int NG = WM / RPP / GP;          // groups per wave tile
    constexpr int GPH = NPS / GP;              // groups per LDS half
    static_assert(GP == 8 || GP == 4 || GP == 2, "explicit waits below are written for 2, 4 or 8 passes per group");
    static_assert(WM / EPH == 1 || WM / EPH == 2, "one or two LDS halves per wave tile");
    bool nlive[GP];
    int64_t noff[GP];
    constexpr int NPRE = PREF == 0 ? GP : NPS;
    gt_u32x4 pre_cur[NPRE], pre_nxt[NPRE];     // prefetched rows of the current / the second LDS half (slot 0 = next pass)
    // optional per-image (sum, sum of squares) of the stored values (theia_gemm_args_t.ln_sums): a wave tile of WM <= 128 rows
    // touches at most two images (rows per image >= 128, checked by the dispatch): slot 0 = the image of the tile's first row
    unsigned long long* const lsum = SUMS ? reinterpret_cast<unsigned long long*>(p.ln_sums) : nullptr;
    int img0, dummy_rem;
    img0 = gt_divmod(m_wave0 < p.M ? m_wave0 : 0, R, rcp_R, dummy_rem);
    const int m_split = (img0 + 1) * R;  // first GEMM row of the second image
    float ls0 = 0.f, lq0 = 0.f, ls1 = 0.f, lq1 = 0.f;
    // Every pass issues its output store unconditionally (dead lanes store to a dump page).
    auto fetch_group = [&](int g) {  // rows of group g: offsets and liveness (the prefetched data is already in pre_cur)
#pragma unroll
        for (int q = 0; q < GP; ++q) {
            nlive[q] = (cur.m < p.M) && n_ok;
            noff[q] = nlive[q] ? cur.off : off_dead;
            next_row(cur);
        }
    };
    // accumulators of EPH rows -> wave-private LDS tile (static register indices: one copy per half, selected by a wave-uniform
    // branch)
    auto stage_half = [&](int half) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int hf = 0; hf < WM / EPH; ++hf) {
            if (half != hf) continue;
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int jj = 0; jj < EPH / 16; ++jj) {
                    const int j = hf * (EPH / 16) + jj;
                    float* q = ep + (jj * 16 + frow) * EP_PITCH + i * 16 + fg * 4;
                    *reinterpret_cast<float4*>(q) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
    };
    GT_EP_STAMP(0)
#pragma unroll
    for (int i = 0; i < NPRE; ++i) pre_cur[i] = pre_nxt[i] = (gt_u32x4){0u, 0u, 0u, 0u};
    const bool prefetch = PREF == 1 || pre_on;  // wave-uniform
    row_cursor_t pc = first_row();
    auto request_half = [&](gt_u32x4 (&dst)[NPRE]) {  // dead lanes read row 0 (valid memory)
#pragma unroll
        for (int i = 0; i < NPS; ++i) {
            const bool lv = (pc.m < p.M) && n_ok;
            const int64_t po = lv ? pc.off : off_dead;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[PREF == 0 ? 0 : i]) : "v"(PRE + po) : "memory");
            next_row(pc);
        }
    };
    if (prefetch) request_half(pre_cur);
    stage_half(0);
    if (prefetch) {
        if (WM / EPH > 1) request_half(pre_nxt);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < NPS; ++i) {  // uses of the rows stay behind the wait
            asm volatile("" : "+v"(pre_cur[PREF == 0 ? 0 : i]));
            asm volatile("" : "+v"(pre_nxt[PREF == 0 ? 0 : i]));
        }
    }
    fetch_group(0);
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0), visible to the compiler: the bias row has landed too
    T* const dump = reinterpret_cast<T*>(g_gt_dump) + lane * 8;
    GT_EP_STAMP(1)
#pragma unroll 1
    for (int g = 0; g < NG; ++g) {
        GT_EP_STAMP(2 + 3 * (g & 3))
        if (g != 0 && g % GPH == 0) {  // second half (the first was staged above)
            stage_half(g / GPH);
            if constexpr (PREF != 0) {
#pragma unroll
                for (int i = 0; i < NPRE; ++i) pre_cur[i] = pre_nxt[i];
            }
        }
        GT_EP_STAMP(3 + 3 * (g & 3))
        bool live[GP];
        int64_t off[GP];
        gt_u32x4 pre[GP];
#pragma unroll
        for (int q = 0; q < GP; ++q) {
            live[q] = nlive[q];
            off[q] = noff[q];
            pre[q] = pre_cur[q];
        }
        if constexpr (PREF != 0) {  // the half's remaining rows move up one group
#pragma unroll
            for (int i = 0; i + GP < NPRE; ++i) pre_cur[i] = pre_cur[i + GP];
        }
        if (g + 1 < NG) fetch_group(g + 1);
#pragma unroll
        for (int q = 0; q < GP; ++q) {  // straight-line per pass: dead lanes compute on row 0 and skip only the stores
            const int64_t o = off[q];
            const int row = ((g % GPH) * GP + q) * RPP + lrow;
            float v[8];
            load8(ep + row * EP_PITCH + col, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (SCALE ? v[j] * alpha : v[j]) + bias8[j];
            if (p.rowtab != nullptr) {
                const int m = m_wave0 + lrow + (g * GP + q) * RPP;
                float t8[8];
                load8(p.rowtab + (int64_t)(m % p.rowtab_period) * p.N + (n_ok ? n : 0), t8);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += t8[j];
            }
            float a8[8];
            if (want_aux) {
                if (sizeof(T) == 2) unpack8(pre[q], a8);
                else load8(AUXI + o, a8);
            }
            if (act == THEIA_ACT_GELU) {
                if (AUXO != nullptr) store8(live[q] ? AUXO + o : dump, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = gt_gelu<T>(v[j]);
            } else if (act == THEIA_ACT_RELU) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j], 0.f);
            } else if (act == THEIA_ACT_MUL_DGELU) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] *= gt_gelu_grad<T>(a8[j]);
            } else if (act == THEIA_ACT_MUL_DRELU) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = a8[j] > 0.f ? v[j] : 0.f;
            }
            if (RES != nullptr) {
                float r8[8];
                if (sizeof(T) == 2 && !want_aux) unpack8(pre[q], r8);
                else load8(RES + o, r8);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += r8[j];
            }
#ifdef GT_EP_NOSTORE
            if (v[0] == 123456.f)
#endif
            store8(live[q] ? O + o : dump, v);
            if constexpr (SUMS) {
                float s = 0.f, sq = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float r = sizeof(T) == 2 ? bf16_to_f32(f32_to_bf16(v[j])) : v[j];  // the value as stored
                    s += r;
                    sq += r * r;
                }
                s = live[q] ? s : 0.f;
                sq = live[q] ? sq : 0.f;
                const bool first = m_wave0 + lrow + (g * GP + q) * RPP < m_split;
                ls0 += first ? s : 0.f;
                lq0 += first ? sq : 0.f;
                ls1 += first ? 0.f : s;
                lq1 += first ? 0.f : sq;
            }
        }
        // group g+1's rows have landed once at most the GP (or more) stores issued after them are outstanding.  Without a
        // prefetch there is nothing to wait for: the stores of all groups stream out back to back (a wait here would hold every
        // group until the previous group's stores are ACKNOWLEDGED, ~4k cycles each when all CUs are in their epilogues at once)
        GT_EP_STAMP(4 + 3 * (g & 3))
    }
    GT_EP_STAMP(14)
    if constexpr (SUMS) {
        ls0 = wave_sum(ls0);
        lq0 = wave_sum(lq0);
        ls1 = wave_sum(ls1);
        lq1 = wave_sum(lq1);
        // 2^-24 fixed point in 64-bit integers: integer addition is associative, so the totals do not depend on the order in
        // which the waves arrive (bit-reproducible steps; float atomics were not), and a wave's partial loses < 6e-8 absolute
        auto fx = [](float v) { return (unsigned long long)__double2ll_rn((double)v * 16777216.0); };
        if (sums_tab != nullptr) {  // workgroup-level combine in LDS, one wave flushes later (gt_flush_sums; see gemm_epi_direct.h)
            if (lane == 0 && m_wave0 < p.M) {
                unsigned long long* t0 = sums_tab + 2 * (img0 - img_tile0);
                atomicAdd(t0, fx(ls0));
                atomicAdd(t0 + 1, fx(lq0));
                if ((int64_t)(img0 + 1) * R < p.M && img0 + 1 - img_tile0 < GT_SUMS_SLOTS) {
                    atomicAdd(t0 + 2, fx(ls1));
                    atomicAdd(t0 + 3, fx(lq1));
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (lane == 0 && m_wave0 < p.M) {
            atomicAdd(lsum + 2 * img0, fx(ls0));
            atomicAdd(lsum + 2 * img0 + 1, fx(lq0));
            if ((int64_t)(img0 + 1) * R < p.M) {  // a second image exists (its sums are zero when the tile did not reach it)
                atomicAdd(lsum + 2 * (img0 + 1), fx(ls1));
                atomicAdd(lsum + 2 * (img0 + 1) + 1, fx(lq1));
            }
        }
    }
}
