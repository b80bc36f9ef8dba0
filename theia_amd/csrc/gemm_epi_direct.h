// LDS-free epilogue of the ping-pong GEMM kernels: the accumulators are converted and stored straight from the MFMA layout.
//
// gt_epilogue (gemm_tile.h) stages every wave tile through LDS to get 16-byte row-contiguous stores: two staging passes (~1.8k
// cycles each per 256x256 tile, tools/pp_trace.hip) and -- worse -- an LDS footprint that collides with the operand ring, so the
// next tile's operands cannot be requested before the epilogue is over.  Here the WEIGHT rows are permuted when they are staged
// (gd_wperm: LDS row rho of a wave's 64 weight rows holds weight row perm(rho); free, the LDS-DMA source address is per lane), so
// that the accumulator fragments of one lane cover CONTIGUOUS output columns:
//     acc[i][j][e]  <->  row m = j*16 + (lane & 15),  column n = (i >> 1)*32 + (lane >> 4)*8 + (i & 1)*4 + e
// i.e. fragments (2t, 2t+1) of a lane are 8 consecutive columns = one 16-byte bf16 store, and the 4 lanes of a row cover 32
// consecutive columns (64 B contiguous; the two t halves of a row complete its 128 B).  Bias, activation, residual, statistics
// are row-local and happen in registers.  No LDS, no lgkmcnt waits, no staging VALU.
#pragma once
#include <type_traits>
#include "gemm_tile.h"

// weight row (within a 64-row wave slice; higher bits pass through) held by LDS row rho
__device__ __forceinline__ int gd_wperm(int rho) { return (rho & ~31) | ((rho & 12) << 1) | ((rho & 16) >> 2) | (rho & 3); }

template <int I0, int I1, typename F> __device__ __forceinline__ void gd_static_for(F&& f) {
    if constexpr (I0 < I1) {
        f(std::integral_constant<int, I0>{});
        gd_static_for<I0 + 1, I1>(f);
    }
}

// Output-row addressing of a row map, stepped 16 rows at a time (one MFMA fragment row group): decode (image, y, x) once per lane,
// then add constants (see gt_epilogue for the derivation; RPP = 16 here).  Only maps whose 16-row step wraps at most once in x and
// once in y (16 / rows_w + 1 <= rows_h) or that have one row per "image" (plain matrices): gd_rows_steppable, checked by the dispatch.
__host__ __device__ inline bool gd_rows_steppable(const theia_rowmap_t& mp) {
    return mp.rows_h * mp.rows_w == 1 || 16 / mp.rows_w + 1 <= mp.rows_h;
}
struct gd_rows_t {
    int R, rows_w, rows_h, step_qw, step_rw;
    bool plain;
    float rcp_R, rcp_w;
    int64_t d_step, d_wx, d_wy;
    struct cursor_t { int m, ry, rx; int64_t off; };

    __device__ __forceinline__ gd_rows_t(const theia_gemm_args_t& p) {
        const theia_rowmap_t& mp = p.map;
        R = mp.rows_h * mp.rows_w;
        rows_w = mp.rows_w;
        rows_h = mp.rows_h;
        rcp_R = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(1.0f / (float)R)));
        rcp_w = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(1.0f / (float)mp.rows_w)));
        step_qw = 16 / mp.rows_w;
        step_rw = 16 - step_qw * mp.rows_w;
        plain = R == 1;
        const int64_t row_pitch = (int64_t)mp.out_sy * mp.out_w * p.ldo;
        d_step = plain ? (int64_t)16 * mp.out_batch_stride : step_qw * row_pitch + (int64_t)step_rw * mp.out_sx * p.ldo;
        d_wx = row_pitch - (int64_t)mp.rows_w * mp.out_sx * p.ldo;
        d_wy = mp.out_batch_stride - mp.rows_h * row_pitch;
    }
    __device__ __forceinline__ int64_t decode(const theia_gemm_args_t& p, int m, int& ry, int& rx) const {
        const theia_rowmap_t& mp = p.map;
        if (plain) {
            ry = rx = 0;
            return (int64_t)m * mp.out_batch_stride + mp.out_offset + (int64_t)(mp.out_y0 * mp.out_w + mp.out_x0) * p.ldo;
        }
        int rem;
        const int img = gt_divmod24(m, R, rcp_R, rem);
        ry = gt_divmod24(rem, rows_w, rcp_w, rx);
        return (int64_t)img * mp.out_batch_stride + mp.out_offset +
               (int64_t)((ry * mp.out_sy + mp.out_y0) * mp.out_w + rx * mp.out_sx + mp.out_x0) * p.ldo;
    }
    __device__ __forceinline__ cursor_t first(const theia_gemm_args_t& p, int m) const {
        cursor_t c;
        c.m = m;
        c.off = decode(p, m, c.ry, c.rx);
        return c;
    }
    __device__ __forceinline__ void next(cursor_t& c) const {  // advance by 16 rows
        c.m += 16;
        c.off += d_step;
        if (!plain) {
            c.rx += step_rw;
            c.ry += step_qw;
            const bool wx = c.rx >= rows_w;
            c.rx -= wx ? rows_w : 0;
            c.ry += wx ? 1 : 0;
            c.off += wx ? d_wx : 0;
            const bool wy = c.ry >= rows_h;
            c.ry -= wy ? rows_h : 0;
            c.off += wy ? d_wy : 0;
        }
    }
};

__device__ __forceinline__ void gd_unpack8(const gt_u32x4& u, float (&f)[8]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f[2 * j] = __uint_as_float(u[j] << 16);
        f[2 * j + 1] = __uint_as_float(u[j] & 0xffff0000u);
    }
}

// one 16-byte row piece, issued as inline asm: invisible to hipcc's waitcnt bookkeeping (the callers wait themselves)
__device__ __forceinline__ void gd_load16(gt_u32x4& dst, const void* ptr) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory");
}

// state shared by the activation-specialised bodies below
template <typename T> struct gd_ctx_t {
    T* O;
    const T* RES;
    const T* AUXI;
    T* AUXO;
    T* dump;
    int ncol[2];
    bool nok[2];
    float bias8[2][8];
    float alpha;
    uint8_t* O8;   // fp8 launches: e4m3 copy of the output (theia_gemm_args_t.out8), or NULL
    float sc8;
    int64_t off_dead;
    int m_split;
    float ls0, lq0, ls1, lq1;
};

template <int N> __device__ __forceinline__ void gd_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// 8 accumulator values of row group j, column half t -> bias / activation / residual applied (everything but the store)
template <typename T, int FM, bool SCALE, bool BIAS_IN_ACC, int ACT>
__device__ __forceinline__ void gd_apply(const gt_f32x4 (&acc)[4][FM], int j, int t, const gd_ctx_t<T>& cx, const float (&a8)[8], bool has_res,
                                         const float (&r8)[8], float (&v)[8], float (&pre)[8]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v[e] = acc[2 * t][j][e];
        v[4 + e] = acc[2 * t + 1][j][e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        if constexpr (SCALE) v[e] *= cx.alpha;
        if constexpr (!BIAS_IN_ACC) v[e] += cx.bias8[t][e];
        pre[e] = v[e];
    }
    if constexpr (ACT == THEIA_ACT_GELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = gt_gelu<T>(v[e]);
    } else if constexpr (ACT == THEIA_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
    } else if constexpr (ACT == THEIA_ACT_MUL_DGELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= gt_gelu_grad<T>(a8[e]);
    } else if constexpr (ACT == THEIA_ACT_MUL_DRELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = a8[e] > 0.f ? v[e] : 0.f;
    }
    if (ACT == THEIA_ACT_NONE && has_res) {  // (the dispatch keeps residual + activation launches off this kernel)
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += r8[e];
    }
}

// One wave tile with a fixed activation.
// PRE (bf16, 128-row wave tiles): the aux_in (MUL_D*) / residual rows are read with inline-asm loads in three phases --
//   A  all FM x 2 row pieces are requested (64 registers next to the 128 accumulators);
//   B  row group by row group: a COUNTED wait (no store has been issued yet, so the counter holds loads only, and loads retire in
//      order -- the next tile's operand fetches, requested earlier, retire first), the arithmetic, and the packed bf16 result is
//      written back INTO the registers the row piece arrived in;
//   C  all stores.
// Loads and stores share the vmcnt counter and retire out of order with respect to each other, so a wait between stores is always a
// wait for everything (the first version of this epilogue prefetched chunk c+1 around chunk c's stores and waited vmcnt(0) per chunk).
template <typename T, int FM, bool SUMS, bool SCALE, bool BIAS_IN_ACC, int ACT, bool PRE>
__device__ __forceinline__ void gd_epilogue_body(gt_f32x4 (&acc)[4][FM], const theia_gemm_args_t& p, const gd_rows_t& rw, int m_row0,
                                                 gd_ctx_t<T>& cx) {
    constexpr bool WANT_AUX = ACT == THEIA_ACT_MUL_DGELU || ACT == THEIA_ACT_MUL_DRELU;
    gd_rows_t::cursor_t cur = rw.first(p, m_row0);
    if constexpr (PRE) {
        static_assert(sizeof(T) == 2 && !SUMS, "row prefetch: bf16 outputs, no statistics");
        const T* __restrict__ PREP = WANT_AUX ? cx.AUXI : cx.RES;
        gt_u32x4 rows[FM][2];
        gd_rows_t::cursor_t pc = cur;
#pragma unroll
        for (int j = 0; j < FM; ++j) {  // phase A; dead lanes read row 0 (valid memory)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const bool lv = (pc.m < p.M) && cx.nok[t];
                gd_load16(rows[j][t], PREP + (lv ? pc.off + cx.ncol[t] : cx.off_dead));
            }
            rw.next(pc);
        }
        gd_static_for<0, FM>([&](auto J_C) {  // phase B
            constexpr int j = decltype(J_C)::value;
            gd_wait_vm<(FM - 1 - j) * 2>();
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                asm volatile("" : "+v"(rows[j][t]));
                float a8[8], v[8], pre[8];
                gd_unpack8(rows[j][t], a8);
                gd_apply<T, FM, SCALE, BIAS_IN_ACC, ACT>(acc, j, t, cx, a8, !WANT_AUX, a8, v, pre);
                rows[j][t] = (gt_u32x4){pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7])};
            }
        });
#pragma unroll
        for (int j = 0; j < FM; ++j) {  // phase C
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const bool live = (cur.m < p.M) && cx.nok[t];
                T* dst = live ? cx.O + cur.off + cx.ncol[t] : cx.dump;
                *reinterpret_cast<gt_u32x4*>(dst) = rows[j][t];
            }
            rw.next(cur);
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < FM; ++j) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const bool live = (cur.m < p.M) && cx.nok[t];
            const int64_t o = live ? cur.off + cx.ncol[t] : cx.off_dead;
            float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, r8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, v[8], pre[8];
            if constexpr (WANT_AUX) load8(cx.AUXI + o, a8);
            const bool has_res = ACT == THEIA_ACT_NONE && cx.RES != nullptr;
            if (has_res) load8(cx.RES + o, r8);
            gd_apply<T, FM, SCALE, BIAS_IN_ACC, ACT>(acc, j, t, cx, a8, has_res, r8, v, pre);
            if constexpr (ACT == THEIA_ACT_GELU) {
                if (cx.AUXO != nullptr) store8(live ? cx.AUXO + o : cx.dump, pre);
            }
            store8(live ? cx.O + o : cx.dump, v);
            if constexpr (SCALE && !SUMS) {  // (fp8 operands only: those launches take this plain loop for every activation)
                if (cx.O8 != nullptr && live) {
                    float am_unused = 0.f;
                    q8_store8(cx.O8, o, v, cx.sc8, am_unused);
                }
            }
            if constexpr (SUMS) {
                float s = 0.f, sq = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float r = sizeof(T) == 2 ? bf16_to_f32(f32_to_bf16(v[e])) : v[e];  // the value as stored
                    s += r;
                    sq += r * r;
                }
                s = live ? s : 0.f;
                sq = live ? sq : 0.f;
                const bool first = cur.m < cx.m_split;
                cx.ls0 += first ? s : 0.f;
                cx.lq0 += first ? sq : 0.f;
                cx.ls1 += first ? 0.f : s;
                cx.lq1 += first ? 0.f : sq;
                // pin the running sums here: left alone, the optimiser sinks the whole reduction behind the last block and keeps
                // every block's 8 stored values alive for it (128 registers: scratch spills)
                asm volatile("" : "+v"(cx.ls0), "+v"(cx.lq0), "+v"(cx.ls1), "+v"(cx.lq1));
            }
        }
        rw.next(cur);
    }
}

// acc: FM x 4 fragments of one wave (rows m_wave0 .. +16*FM, columns n_wave0 .. +64 in the permuted order above).
// T = output element type.  BIAS_IN_ACC: the kernel initialised the accumulators with the bias row (then it is not added again);
// resid_in_acc: the same for the residual rows (act == NONE only).  SCALE: fp8 operands, accumulators are rescaled first.
template <typename T, int FM, bool SUMS, bool SCALE, bool BIAS_IN_ACC>
__device__ __forceinline__ int gd_epilogue(gt_f32x4 (&acc)[4][FM], const theia_gemm_args_t& p, const gd_rows_t& rw, int m_wave0,
                                           int n_wave0, int lane, bool resid_in_acc, unsigned long long* sums_tab = nullptr,
                                           int m_tile0 = 0) {  // -> SUMS: the tile's first image (for gt_flush_sums), else 0
    const int frow = lane & 15, fg = lane >> 4;
    gd_ctx_t<T> cx;
    cx.O = reinterpret_cast<T*>(p.out);
    cx.RES = resid_in_acc ? nullptr : reinterpret_cast<const T*>(p.resid);
    cx.AUXI = reinterpret_cast<const T*>(p.aux_in);
    cx.AUXO = reinterpret_cast<T*>(p.aux_out);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        cx.ncol[t] = n_wave0 + t * 32 + fg * 8;
        cx.nok[t] = cx.ncol[t] < p.N;
#pragma unroll
        for (int e = 0; e < 8; ++e) cx.bias8[t][e] = 0.f;
        if constexpr (!BIAS_IN_ACC) {
            if (p.bias != nullptr && cx.nok[t]) load8(p.bias + cx.ncol[t], cx.bias8[t]);
        }
    }
    cx.O8 = nullptr;
    cx.sc8 = 0.f;
    if constexpr (SCALE && !SUMS) {
        cx.O8 = p.out8;
        cx.sc8 = p.out8 != nullptr ? *p.out8_scale : 0.f;
    }
    cx.alpha = 1.0f;
    if constexpr (SCALE) cx.alpha = (p.a_scale_inv != nullptr ? *p.a_scale_inv : 1.0f) * (p.w_scale_inv != nullptr ? *p.w_scale_inv : 1.0f);
    int dry, drx;
    cx.off_dead = rw.decode(p, 0, dry, drx);  // where dead lanes (rows >= M, columns >= N) read from: row 0, column 0
    cx.dump = reinterpret_cast<T*>(g_gt_dump) + lane * 8;
    // per-image (sum, sum of squares) of the stored values: see gt_epilogue
    unsigned long long* const lsum = SUMS ? reinterpret_cast<unsigned long long*>(p.ln_sums) : nullptr;
    int img0 = 0, img_tile0 = 0;
    cx.m_split = 0;
    if constexpr (SUMS) {
        int dummy;
        img0 = gt_divmod24(m_wave0 < p.M ? m_wave0 : 0, rw.R, rw.rcp_R, dummy);
        cx.m_split = (img0 + 1) * rw.R;
        img_tile0 = __builtin_amdgcn_readfirstlane(gt_divmod24(m_tile0 < p.M ? m_tile0 : 0, rw.R, rw.rcp_R, dummy));
    }
    cx.ls0 = cx.lq0 = cx.ls1 = cx.lq1 = 0.f;
    // the row prefetch costs 32 registers: the 160-row wave tile and the statistics epilogues (whose launches never carry an
    // aux_in / residual row: the dispatch sends that combination elsewhere) read such rows with plain loads instead
    constexpr bool CAN_PRE = sizeof(T) == 2 && FM <= 8 && !SUMS && BIAS_IN_ACC;  // (fp8 operands: 16 bias registers more, no room)
    const int m_row0 = m_wave0 + frow;
    if constexpr (SUMS) {  // the statistics launches are the convolutions in front of a LayerNorm[C,H,W]: no activation or ReLU
        if (p.act == THEIA_ACT_RELU) gd_epilogue_body<T, FM, SUMS, SCALE, BIAS_IN_ACC, THEIA_ACT_RELU, false>(acc, p, rw, m_row0, cx);
        else gd_epilogue_body<T, FM, SUMS, SCALE, BIAS_IN_ACC, THEIA_ACT_NONE, false>(acc, p, rw, m_row0, cx);
    } else {
        switch (p.act) {
            case THEIA_ACT_GELU: gd_epilogue_body<T, FM, SUMS, SCALE, BIAS_IN_ACC, THEIA_ACT_GELU, false>(acc, p, rw, m_row0, cx); break;
            case THEIA_ACT_RELU: gd_epilogue_body<T, FM, SUMS, SCALE, BIAS_IN_ACC, THEIA_ACT_RELU, false>(acc, p, rw, m_row0, cx); break;
            case THEIA_ACT_MUL_DGELU: gd_epilogue_body<T, FM, SUMS, SCALE, BIAS_IN_ACC, THEIA_ACT_MUL_DGELU, CAN_PRE>(acc, p, rw, m_row0, cx); break;
            case THEIA_ACT_MUL_DRELU: gd_epilogue_body<T, FM, SUMS, SCALE, BIAS_IN_ACC, THEIA_ACT_MUL_DRELU, CAN_PRE>(acc, p, rw, m_row0, cx); break;
            default:
                if (CAN_PRE && cx.RES != nullptr) gd_epilogue_body<T, FM, SUMS, SCALE, BIAS_IN_ACC, THEIA_ACT_NONE, CAN_PRE>(acc, p, rw, m_row0, cx);
                else gd_epilogue_body<T, FM, SUMS, SCALE, BIAS_IN_ACC, THEIA_ACT_NONE, false>(acc, p, rw, m_row0, cx);
                break;
        }
    }
    if constexpr (SUMS) {
        const float ls0 = wave_sum(cx.ls0), lq0 = wave_sum(cx.lq0), ls1 = wave_sum(cx.ls1), lq1 = wave_sum(cx.lq1);
        // 2^-24 fixed point in 64-bit integers: integer addition is associative, so the totals do not depend on the order in
        // which the waves arrive (bit-reproducible steps), and a wave's partial loses < 6e-8 absolute
        auto fx = [](float v) { return (unsigned long long)__double2ll_rn((double)v * 16777216.0); };
        // The waves' partials meet in a table in LDS (4 images x (sum, sum of squares); LDS integer atomics) and ONE wave adds the tile's
        // totals to global memory later (gd_flush_sums, after the workgroup's next barrier): 2-4 global atomics per tile instead of
        // 16-32.  Device-scope atomics are performed at the memory side (the per-XCD L2s are not coherent with each other) and
        // serialise per address: with one pair per wave they cost 70-105 us per launch on the 64x64 maps (285 -> 179 us at K = 768,
        // profiles/r03_ab_ln_sums_atomics.txt).
        if (lane == 0 && m_wave0 < p.M) {
            unsigned long long* t0 = sums_tab + 2 * (img0 - img_tile0);
            atomicAdd(t0, fx(ls0));
            atomicAdd(t0 + 1, fx(lq0));
            if ((int64_t)(img0 + 1) * rw.R < p.M && img0 + 1 - img_tile0 < GT_SUMS_SLOTS) {  // a second image exists
                atomicAdd(t0 + 2, fx(ls1));
                atomicAdd(t0 + 3, fx(lq1));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // performed before this wave reaches the barrier the flush waits behind
    }
    return img_tile0;
}
