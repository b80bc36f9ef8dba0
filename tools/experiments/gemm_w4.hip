// One-wave-per-SIMD persistent NT GEMM for the long-K plain-matrix launches (every nn.Linear of the ViT): 256 x 256 tiles, FOUR wave64
// (one per SIMD, 512 registers each), wave tile 128 x 128 = 4 x 4 fragments of v_mfma_f32_32x32x16_bf16 (256 accumulator registers).
//
// Why (round-4 review item 1; profiles/r04_epilogue_*.txt): the 8-wave ping-pong kernel (gemm_pp.hip) loses 12-23k idle-pipe cycles
// per 35k-cycle K = 768 tile to its epilogue (lane-order-bound stores, ~9k cycles of GELU VALU) and to the tile switch, and two waves
// per SIMD leave no registers to keep a finished tile around.  Here a wave owns the whole register file of its SIMD, so the finished
// tile stays in registers as PACKED bf16 (128 registers) while the accumulators start the next tile, and its epilogue -- activation,
// conversions, stores -- is issued in the shadow of the NEXT tile's MFMAs, one small chunk per k-step:
//
//   tile t:   |switch| 32 k-steps carrying the epilogue chunks of tile t-1 (half fragments) | plain k-steps ........ |
//   switch of tile t+1: fragment by fragment, pack acc -> bf16 (16 AGPR reads + 8 v_cvt_pk), then one MFMA of synthetic operands that
//   re-initialises the fragment with its bias (bias = bf16 hi + bf16 lo in two k slots against ones: 2^-17 relative, no VALU, no copies).
//
// Operands: 4-deep ring of half k-tiles (32 k of 256 + 256 rows, 64-byte rows, source-side XOR swizzle) filled by LDS-DMA
// (buffer_load_dwordx4 ... lds: one 32-bit offset register per piece, the k position in an SGPR), fragments double-buffered in
// registers: the reads of k-step s+1 are issued between the MFMAs of k-step s.  ONE barrier per half k-tile:
//     half-tile g:  [16 MFMA (g, s=0) || reads (g, s=1)]  wait(g+1 landed) BARRIER  [16 MFMA (g, s=1) || reads (g+1, s=0)]
// with the 8 DMA pieces of half-tile g+3 (into the slot of g-1: every wave had it in registers before the previous barrier) spread
// over the 32 MFMA gaps, one piece every 4th gap, the four waves one gap apart.
// Counted waits only; the epilogue's stores share the counter with the DMA loads and retire out of order with respect to them, which
// can only make a counted wait stricter (see gemm_pp.hip).
//
// The weight rows are permuted at staging so that a lane's 16 accumulators of a fragment are 2 x 8 CONSECUTIVE output columns of one
// row (two 16-byte stores): MFMA row r of a 32-row weight fragment holds column ((r>>4)&1)*16 + ((r>>2)&1)*8 + ((r>>3)&1)*4 + (r&3).
//
// Scope: bf16, single-tap plain maps (rows_h * rows_w == 1), K % 32 == 0 and K >= 512 (GELU: 768), N % 256 == 0, no rowtab / ln_sums;
// activations NONE / GELU (+ saved pre-activation) here; everything else stays on gemm_pp.hip.
#include <type_traits>
#include "../../theia_amd/csrc/gemm_epi_direct.h"

typedef __attribute__((ext_vector_type(16))) float w4_f32x16;

constexpr int W4_SLOT = 32768;                 // one half k-tile: 256 activation rows + 256 weight rows of 64 B
constexpr int W4_NSLOT = 4;
constexpr int W4_BIAS_LDS = W4_NSLOT * W4_SLOT;  // two bias rows (this tile's, the next tile's) of 256 floats
constexpr int W4_LDS = W4_BIAS_LDS + 2048;
constexpr int W4_NCH = 32;                     // epilogue chunks per tile (half fragments: 4 packed registers = one 16-byte store)
// gaps (MFMA slots) one epilogue chunk is spread over, and the unrolled half-tiles per tile that carry the 32 chunks (the minimum K / 32)
constexpr int w4_span(int act) { return act == THEIA_ACT_GELU ? 24 : 16; }
constexpr int w4_hunr(int act) { return W4_NCH * w4_span(act) / 32; }
// a buffer offset beyond every num_records (the launcher keeps every buffer of this kernel below 2^31 - 2^24 bytes, and nothing added to
// it -- row steps of the store base, column offsets -- reaches 2^31, so it never wraps back into range): loads return zeros, stores drop
constexpr uint32_t W4_OOB = 0x80000000u;

#ifdef W4_TRACE
__device__ unsigned long long g_w4_phase[4][16];
#define W4_PHASE(k) \
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (k) < 16) g_w4_phase[threadIdx.x >> 6][k] = __builtin_readcyclecounter();
#else
#define W4_PHASE(k)
#endif

__device__ __forceinline__ int w4_f(int row) { return (4 - ((row >> 2) & 3)) & 3; }
// weight row (within a 16-row staging piece; the piece's position inside its 32-row fragment adds 16) held by LDS row q
__device__ __forceinline__ int w4_nperm16(int q) { return ((q >> 2) & 1) * 8 + ((q >> 3) & 1) * 4 + (q & 3); }

template <int N> __device__ __forceinline__ void w4_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void w4_wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

typedef __attribute__((address_space(3))) void* w4_lds_ptr;
// uses of the fragment registers stay behind the wait in front of this
__device__ __forceinline__ void w4_pin8(gt_u32x4 (&a)[4], gt_u32x4 (&b)[4]) {
    asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
}
__device__ __forceinline__ void w4_pin4(gt_u32x4 (&a)[4]) { asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])); }

template <int ACT>
__global__ __launch_bounds__(256) void gemm_nt_w4_kernel(const theia_gemm_args_t p, const int ntiles, const int panel) {
    constexpr int SPAN = w4_span(ACT), HUNR = w4_hunr(ACT);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = w >> 1, wc = w & 1;      // wave tile: rows wr*128.., columns wc*128..
    const int hh = lane >> 5, m32 = lane & 31;
    const theia_rowmap_t& mp = p.map;
    const int tiles_n = p.N >> 8;
    const int nh = p.K >> 5;
    W4_PHASE(0)
    // ---------------------------------------------------------------- schedule: as gemm_pp.hip (static rounds, XCD-contiguous, column panels)
    const int grid = gridDim.x, bid = blockIdx.x;
    const int rounds = (ntiles + grid - 1) / grid;
    const int cnt_last = ntiles - (rounds - 1) * grid;
    const int my_tiles = rounds - 1 + (bid < cnt_last ? 1 : 0);
    auto tile_of = [&](int r) {
        const int t = r * grid + gt_xcd_remap(bid, r + 1 < rounds ? grid : cnt_last);
        if (panel <= 0) return t;
        const int tiles_m = ntiles / tiles_n;
        const int per = panel * tiles_m, full = tiles_n / panel;
        const int pi = min(t / per, full);
        const int cols = pi < full ? panel : tiles_n - full * panel;
        const int rem = t - pi * per;
        const int m = rem / cols, n = pi * panel + (rem - m * cols);
        return m * tiles_n + n;
    };
    // ---------------------------------------------------------------- buffers
    const uint32_t lda2 = (uint32_t)mp.in_batch_stride * 2u, ldw2 = (uint32_t)p.ldw * 2u, ldo2 = (uint32_t)mp.out_batch_stride * 2u;
    const bf16_t* Abase = reinterpret_cast<const bf16_t*>(p.a) + mp.in_offset;
    bf16_t* Obase = reinterpret_cast<bf16_t*>(p.out) + mp.out_offset + (int64_t)(mp.out_y0 * mp.out_w + mp.out_x0) * p.ldo;
    const int64_t aux_delta = mp.out_offset + (int64_t)(mp.out_y0 * mp.out_w + mp.out_x0) * p.ldo;
    const uint32_t out_bytes = (uint32_t)(p.M - 1) * ldo2 + (uint32_t)p.N * 2u;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Abase), 0, (int)((uint32_t)(p.M - 1) * lda2 + (uint32_t)p.K * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)((uint32_t)(p.N - 1) * ldw2 + (uint32_t)p.K * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(Obase, 0, (int)out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
        p.aux_out != nullptr ? reinterpret_cast<bf16_t*>(p.aux_out) + aux_delta : Obase, 0, (int)(p.aux_out != nullptr ? out_bytes : 0u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, (int)(p.bias != nullptr ? (uint32_t)p.N * 4u : 0u), 0x00020000);

    // ---------------------------------------------------------------- operand stream (LDS-DMA): 4 + 4 pieces of 16 rows per wave and half-tile
    uint32_t voffA[4], voffW[4], voffB;
    uint32_t koff = 0;  // byte offset of the stream's next half-tile inside its rows (SGPR)
    int pf_h = 0;
    int nx_par = 0, nx_m0 = 0, nx_n0 = 0;  // origin of the tile the stream enters next (m0 < 0: none -- every piece out of range: zeros, no traffic)
    auto set_stream_tile = [&](int m0, int n0) {
        const int q16 = lane >> 2;
        const uint32_t chunk_b = (uint32_t)(((lane & 3) ^ w4_f(q16)) << 4);
        if (m0 < 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) voffA[q] = voffW[q] = W4_OOB;
            voffB = W4_OOB;
            return;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = min(m0 + w * 64 + q * 16 + q16, p.M - 1);
            voffA[q] = (uint32_t)row * lda2 + chunk_b;
            const int n = n0 + w * 64 + q * 16 + w4_nperm16(q16);  // (q * 16: fragment (q >> 1), upper / lower 16 MFMA rows (q & 1))
            voffW[q] = (uint32_t)n * ldw2 + chunk_b;
        }
        voffB = (uint32_t)(n0 + lane * 4) * 4u;
    };
    auto dma_piece = [&](auto P_C, int slot) {
        constexpr int P = decltype(P_C)::value;
        if constexpr (P < 4) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (w4_lds_ptr)(smem + slot * W4_SLOT + (w * 64 + P * 16) * 64), 16, voffA[P], koff, 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (w4_lds_ptr)(smem + slot * W4_SLOT + 16384 + (w * 64 + (P - 4) * 16) * 64), 16,
                                                     voffW[P - 4], koff, 0, 0);
        }
    };
    auto dma_bias = [&](int par) {  // (every wave issues it: the same bytes to the same place -- keeps the counted waits wave-independent)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (w4_lds_ptr)(smem + W4_BIAS_LDS + par * 1024), 16, voffB, 0, 0, 0);
    };
    // after a DMA group: advance the stream; entering the next tile re-derives the 8 piece offsets and requests its bias row
    auto stream_advance = [&]() {
        ++pf_h;
        koff += 64;
        if (pf_h == nh) {  // wave-uniform, once per tile
            pf_h = 0;
            koff = 0;
            set_stream_tile(nx_m0, nx_n0);
            dma_bias(nx_par);
        }
    };

    // ---------------------------------------------------------------- fragments
    const uint32_t sb = gt_lds_addr(smem);
    const uint32_t c0 = (uint32_t)((hh ^ w4_f(m32)) << 4);
    const uint32_t laneA0 = sb + (wr * 128 + m32) * 64 + c0, laneW0 = sb + 16384 + (wc * 128 + m32) * 64 + c0;
    gt_u32x4 FW[2][4], FA[2][4];
    w4_f32x16 acc[4][4];
    uint32_t pk[16][8];
    // read k-step S of the half-tile in `slot` into buffer S: piece R (0..3 weight fragments, 4..7 activation fragments)
    auto frag_read = [&](auto S_C, auto R_C, int slot) {
        constexpr int S = decltype(S_C)::value, R = decltype(R_C)::value;
        const uint32_t so = (uint32_t)slot * W4_SLOT;
        if constexpr (R < 4) gt_ds_read128<R * 2048>(FW[S][R], ((S ? laneW0 ^ 32u : laneW0)) + so);
        else gt_ds_read128<(R - 4) * 2048>(FA[S][R - 4], ((S ? laneA0 ^ 32u : laneA0)) + so);
    };
    auto frag_pin = [&](auto S_C) { w4_pin8(FW[decltype(S_C)::value], FA[decltype(S_C)::value]); };
    // ---------------------------------------------------------------- epilogue of the PREVIOUS tile, one chunk per k-step
    // chunk c: fragment f = c >> 1 (i = f >> 2, j = f & 3), half q = c & 1: packed registers pk[f][4q .. 4q+3] = 8 columns of one row.
    // NONE: one 16-byte store.  GELU: pk holds the PRE-activation (bias included) rounded to bf16 -- it is stored as aux_out as is, and
    // gelu() of the rounded value is the output (what a bf16 Linear followed by a GELU computes); one value per two MFMA gaps.
    uint32_t st_base = W4_OOB;  // byte offset of (row m_prev0 + wr*128 + m32, column n_prev0 + wc*128 + hh*8); OOB: no previous tile
    float gx[2] = {0.f, 0.f}, gxc[2] = {0.f, 0.f}, gs[2] = {0.f, 0.f}, gp[2] = {0.f, 0.f}, ga[8];
    // GX = gap index counted from the tile's first k-step (16 per k-step); a chunk spans SPAN gaps: 16 (NONE: one store per k-step) or
    // 24 (GELU: one value per three gaps, 5-6 VALU instructions per gap -- what one wave hides beside a 32-cycle MFMA; the compiler
    // would otherwise pair values (v_pk_mul_f32) and drop both polynomials into one gap: the opaque asm statements pin each part)
    auto chunk_part = [&](auto GX_C) {
        constexpr int GX = decltype(GX_C)::value;
        constexpr int C = GX >= 0 ? GX / SPAN : -1, POS = GX >= 0 ? GX % SPAN : 0;
        if constexpr (C >= 0 && C < W4_NCH) {
            constexpr int F = C >> 1, Q = C & 1, I = F >> 2, J = F & 3;
            constexpr uint32_t coff = (I * 32 + Q * 16) * 2;
            if constexpr (ACT == THEIA_ACT_NONE) {
                if constexpr (POS == 8) {
                    const uint32_t vo = st_base + (uint32_t)(J * 32) * ldo2 + coff;
                    const gt_u32x4 d = {pk[F][Q * 4], pk[F][Q * 4 + 1], pk[F][Q * 4 + 2], pk[F][Q * 4 + 3]};
                    __builtin_amdgcn_raw_buffer_store_b128(d, rsO, vo, 0, 0);
                }
            } else {  // GELU: the two halves of one packed register at a time -- two independent dependency chains per gap
                constexpr int PAIR = POS / 6, PART = POS % 6;
                auto part = [&](float& x, float& xc, float& sq, float& pp, float& out, bool hi) {
                    if constexpr (PART == 0) {
                        const uint32_t u = pk[F][Q * 4 + PAIR];
                        x = __uint_as_float(hi ? (u & 0xffff0000u) : (u << 16));
                        xc = __builtin_amdgcn_fmed3f(x, -4.5f, 4.5f);
                        sq = xc * (1.0f / 4.5f);
                    } else if constexpr (PART == 1) {
                        sq = sq * sq;
                        pp = fmaf(-1.050371170e+00f, sq, 6.019040585e+00f);
                        pp = fmaf(pp, sq, -1.537067318e+01f);
                    } else if constexpr (PART == 2) {
                        pp = fmaf(pp, sq, 2.330106735e+01f);
                        pp = fmaf(pp, sq, -2.366455650e+01f);
                        pp = fmaf(pp, sq, 1.729463768e+01f);
                    } else if constexpr (PART == 3) {
                        pp = fmaf(pp, sq, -9.533602715e+00f);
                        pp = fmaf(pp, sq, 4.061982155e+00f);
                        pp = fmaf(pp, sq, -1.345344782e+00f);
                    } else if constexpr (PART == 4) {
                        pp = fmaf(pp, sq, 3.989298940e-01f);
                        pp = fmaf(xc, pp, 0.5f);
                        pp = __builtin_amdgcn_fmed3f(fmaf(pp, 1.000034f, -1.7e-5f), 0.f, 1.f);
                    } else {
                        out = x * pp;
                    }
                };
                part(gx[0], gxc[0], gs[0], gp[0], ga[2 * PAIR], false);
                part(gx[1], gxc[1], gs[1], gp[1], ga[2 * PAIR + 1], true);
                if constexpr (PART < 5) asm volatile("" : "+v"(gx[0]), "+v"(gxc[0]), "+v"(gs[0]), "+v"(gp[0]), "+v"(gx[1]), "+v"(gxc[1]), "+v"(gs[1]), "+v"(gp[1]));
                else asm volatile("" : "+v"(ga[2 * PAIR]), "+v"(ga[2 * PAIR + 1]));
                if constexpr (POS == SPAN - 1) {
                    const uint32_t vo = st_base + (uint32_t)(J * 32) * ldo2 + coff;
                    const gt_u32x4 d = {pk[F][Q * 4], pk[F][Q * 4 + 1], pk[F][Q * 4 + 2], pk[F][Q * 4 + 3]};
                    __builtin_amdgcn_raw_buffer_store_b128(d, rsX, vo, 0, 0);  // (no aux_out: num_records = 0, dropped)
                    const gt_u32x4 o = {pack2_bf16(ga[0], ga[1]), pack2_bf16(ga[2], ga[3]), pack2_bf16(ga[4], ga[5]), pack2_bf16(ga[6], ga[7])};
                    __builtin_amdgcn_raw_buffer_store_b128(o, rsO, vo, 0, 0);
                }
            }
        }
    };
    // The accumulators live in the AGPR half of the register file.  One asm statement per packed dword -- two AGPR reads and the
    // conversion -- so that exactly one VGPR result stays live (left to the compiler, the AGPR -> VGPR copies of all 256 accumulators are
    // placed at the top of the block, in front of the first scheduling barrier, and the conversions are sunk to the stores: 135-240 spills)
    auto pack_frag = [&](auto F_C) {
        constexpr int F = decltype(F_C)::value;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            uint32_t lo, hi;
            asm volatile("v_accvgpr_read_b32 %0, %2\n\tv_accvgpr_read_b32 %1, %3\n\tv_cvt_pk_bf16_f32 %0, %0, %1"
                         : "=&v"(lo), "=&v"(hi)
                         : "a"(acc[F >> 2][F & 3][2 * e]), "a"(acc[F >> 2][F & 3][2 * e + 1]));
            pk[F][e] = lo;
        }
    };
    // Bias: the accumulators are AGPRs and only whole MFMA tuples can be written without a detour through VGPRs, so the bias row enters
    // through the matrix pipe: the switch "k-step" multiplies a synthetic weight fragment -- k = 0: bf16(bias), k = 1: bf16(bias -
    // bf16(bias)), rest 0 -- with a synthetic activation fragment of ones in k = 0, 1: acc = bias to 2^-17 relative, C = 0, no VALU.
    // (16 MFMAs per tile that hide behind the packing's VALU.)
    gt_u32x4 Wb[4], Aone;
    auto bias_prep = [&](int par) {
        const int rho = m32;
        const int ncol = ((rho >> 4) & 1) * 16 + ((rho >> 2) & 1) * 8 + ((rho >> 3) & 1) * 4 + (rho & 3);  // column held by MFMA row rho
        const uint32_t a = sb + W4_BIAS_LDS + par * 1024 + (uint32_t)(wc * 128 + ncol) * 4u;
        uint32_t b0, b1, b2, b3;
        asm volatile("ds_read_b32 %0, %1 offset:0" : "=v"(b0) : "v"(a));
        asm volatile("ds_read_b32 %0, %1 offset:128" : "=v"(b1) : "v"(a));
        asm volatile("ds_read_b32 %0, %1 offset:256" : "=v"(b2) : "v"(a));
        asm volatile("ds_read_b32 %0, %1 offset:384" : "=v"(b3) : "v"(a));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3)::"memory");
        const uint32_t bb[4] = {b0, b1, b2, b3};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float f = __uint_as_float(bb[i]);
            const float fh = bf16_to_f32(f32_to_bf16(f));
            const uint32_t x = pack2_bf16(fh, f - fh);
            Wb[i] = (gt_u32x4){hh == 0 ? x : 0u, 0u, 0u, 0u};
        }
        Aone = (gt_u32x4){hh == 0 ? 0x3F803F80u : 0u, 0u, 0u, 0u};
    };
    // the tile switch: fragment by fragment, pack the finished accumulators, then re-initialise them with the bias
    auto switch_step = [&]() {
        gd_static_for<0, 16>([&](auto G_C) {
            constexpr int G = decltype(G_C)::value, I = G >> 2, J = G & 3;
            pack_frag(G_C);
            const w4_f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gt_bf16x8, Wb[I]), __builtin_bit_cast(gt_bf16x8, Aone), z, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    // ---------------------------------------------------------------- one k-step: 16 MFMAs (weight fragment i outer, activation fragment j
    // inner) on fragment buffer S; gap G = what is issued between MFMA G and MFMA G + 1
    //   S = 0: the reads of (this half-tile, s = 1) in gaps 0-7
    //   S = 1: the 8 DMA pieces of half-tile g + 4 into this half-tile's slot in gaps 0-7, the reads of (next half-tile, s = 0) in gaps 8-15
    //   SW:    switch step (S = 0, first k-step of a tile): pack fragment G, then its first MFMA of the new tile with C = bias fragment
    //   C:     epilogue chunk of the previous tile carried by this step (-1: none)
    auto kstep = [&](auto S_C, auto T_C, int slot_cur, int slot_next, int slot_dma) {
        constexpr int S = decltype(S_C)::value, T = decltype(T_C)::value;
        gd_static_for<0, 16>([&](auto G_C) {
            constexpr int G = decltype(G_C)::value, I = G >> 2, J = G & 3;
            acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gt_bf16x8, FW[S][I]), __builtin_bit_cast(gt_bf16x8, FA[S][J]),
                                                                acc[I][J], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#ifndef W4_NO_READS
            if constexpr (G < 8) {
                if constexpr (S == 0) frag_read(std::integral_constant<int, 1>{}, G_C, slot_cur);
                else frag_read(std::integral_constant<int, 0>{}, G_C, slot_next);
            }
#endif
            // One DMA piece per wave every 4th gap.  (Staggering the four waves by one gap with a wave-uniform branch per gap was
            // measured: the taken branches cost more than the address-unit contention they avoid.)
#ifndef W4_NO_DMA
            if constexpr ((G & 3) == 3) dma_piece(std::integral_constant<int, S * 4 + (G >> 2)>{}, slot_dma);
#endif
            chunk_part(std::integral_constant<int, T >= 0 ? T * 16 + G : -1>{});
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    int g = 0;  // half-tile counter of the workgroup: ring slot = g & 3
    auto half_tile = [&](auto H_C) {  // H: half-tile index inside the tile's unrolled part (-1: no epilogue chunk)
        constexpr int HT = decltype(H_C)::value;
        const std::integral_constant<int, HT >= 0 ? 2 * HT : -1> C0_C;
        const std::integral_constant<int, HT >= 0 ? 2 * HT + 1 : -1> C1_C;
        const int slot = g & 3, slot_n = (g + 1) & 3, slot_d = (g + 3) & 3;
        kstep(std::integral_constant<int, 0>{}, C0_C, slot, slot_n, slot_d);
        w4_wait_lgkm0();
        frag_pin(std::integral_constant<int, 1>{});
        w4_wait_vm<12>();  // half-tile g + 1 has landed (g + 2 and the first four pieces of g + 3 may be in flight)
#ifndef W4_NO_BARRIER
        __builtin_amdgcn_s_barrier();
#endif
        __builtin_amdgcn_sched_barrier(0);
        kstep(std::integral_constant<int, 1>{}, C1_C, slot, slot_n, slot_d);
        stream_advance();
        w4_wait_lgkm0();
        frag_pin(std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        ++g;
    };

    if (p.bias == nullptr) {  // no bias: the out-of-range bias fetches may or may not write zeros -- the two rows are zeroed here, once
        reinterpret_cast<uint64_t*>(smem + W4_BIAS_LDS)[threadIdx.x] = 0ull;
    }
    // ---------------------------------------------------------------- prologue: stream 4 half-tiles ahead, fragments of (0, s = 0), bias fragment 0
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#pragma unroll
    for (int f = 0; f < 16; ++f)
#pragma unroll
        for (int e = 0; e < 8; ++e) pk[f][e] = 0u;
    int tile = tile_of(0);
    set_stream_tile((tile / tiles_n) * 256, (tile % tiles_n) * 256);
    dma_bias(0);
    {
        const int t1 = my_tiles > 1 ? tile_of(1) : -1;
        nx_m0 = t1 < 0 ? -1 : (t1 / tiles_n) * 256;
        nx_n0 = t1 < 0 ? 0 : (t1 % tiles_n) * 256;
        nx_par = 1;
    }
#pragma unroll
    for (int s = 0; s < W4_NSLOT - 1; ++s) {
        gd_static_for<0, 8>([&](auto P_C) { dma_piece(P_C, s); });
        stream_advance();
    }
    W4_PHASE(1)
    w4_wait_vm<16>();  // the bias row and half-tile 0 have landed
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    gd_static_for<0, 8>([&](auto R_C) { frag_read(std::integral_constant<int, 0>{}, R_C, 0); });
    w4_wait_lgkm0();
    frag_pin(std::integral_constant<int, 0>{});
    W4_PHASE(2)

    for (int r = 0; r < my_tiles; ++r) {
        const int par = r & 1;
        W4_PHASE(3 + 3 * r)
        // the switch, then the unrolled part: 32 k-steps that carry the chunks of the previous tile's epilogue
        bias_prep(par);
        switch_step();
        gd_static_for<0, HUNR>([&](auto H_C) { half_tile(H_C); });
        W4_PHASE(4 + 3 * r)
        for (int h = HUNR; h < nh; ++h) half_tile(std::integral_constant<int, -1>{});
        W4_PHASE(5 + 3 * r)
        // the finished tile becomes "the previous tile": its store base; bias fragment 0 of the next tile (its row landed with the next
        // tile's first half-tile, waited for and fenced in the last half-tile above)
        {
            const int m0 = (tile / tiles_n) * 256, n0 = (tile % tiles_n) * 256;
            st_base = (uint32_t)(m0 + wr * 128 + m32) * ldo2 + (uint32_t)(n0 + wc * 128 + hh * 8) * 2u;
        }
        if (r + 1 < my_tiles) {
            tile = tile_of(r + 1);
            // (the stream entered tile r + 1 four half-tiles ago: from here on its next crossing goes to tile r + 2)
            const int t2 = r + 2 < my_tiles ? tile_of(r + 2) : -1;
            nx_m0 = t2 < 0 ? -1 : (t2 / tiles_n) * 256;
            nx_n0 = t2 < 0 ? 0 : (t2 % tiles_n) * 256;
            nx_par = par;
        }
    }
    // ---------------------------------------------------------------- drain: the last tile's epilogue, nothing to hide it behind
    gd_static_for<0, 16>([&](auto F_C) { pack_frag(F_C); });
    gd_static_for<0, W4_NCH * SPAN>([&](auto GX_C) { chunk_part(GX_C); });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the out-of-range tail fetches still write LDS)
    W4_PHASE(15)
}

// ---------------------------------------------------------------------------------------------------------------- launch
bool theia_gemm_nt_w4_supported(const theia_gemm_args_t* a, int dtype) {
    const theia_rowmap_t& mp = a->map;
    if (dtype != THEIA_BF16) return false;
    if (mp.ntaps != 1 || mp.rows_h * mp.rows_w != 1 || mp.dy[0] != 0 || mp.dx[0] != 0) return false;
    if (a->K % 32 != 0 || a->K < 32 * w4_hunr(a->act) || a->N % 256 != 0 || a->M < 1) return false;
    if (a->rowtab != nullptr || a->ln_sums != nullptr || a->resid != nullptr || a->aux_in != nullptr) return false;
    if (a->act != THEIA_ACT_NONE && a->act != THEIA_ACT_GELU) return false;
    const uint64_t lim = 0x7F000000ull;
    if (((uint64_t)a->M + 256) * (uint64_t)mp.in_batch_stride * 2 >= lim || ((uint64_t)a->M + 256) * (uint64_t)mp.out_batch_stride * 2 >= lim ||
        (uint64_t)a->N * (uint64_t)a->ldw * 2 >= lim)
        return false;
    return true;
}

int g_w4_grid_cap = 0;
template <int ACT> static int w4_launch_one(const theia_gemm_args_t* a, hipStream_t stream) {
    auto kern = gemm_nt_w4_kernel<ACT>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS);
        attr_set = true;
    }
    const int tiles = cdiv_i(a->M, 256) * (a->N / 256);
    const int cus = g_w4_grid_cap > 0 ? g_w4_grid_cap : theia_compute_cus();
    const int grid = tiles < cus ? tiles : cus;
    const int tn = a->N / 256;
    const long wbytes = (long)a->N * a->K * 2, colblock = 256L * a->K * 2;
    int panel = 0;
    if (tn >= 8 && wbytes > (3L << 20) && tiles > grid) {
        panel = (int)((1600L << 10) / colblock);
        if (panel < 1) panel = 1;
        if (panel >= tn) panel = 0;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), W4_LDS, stream, *a, tiles, panel);
    THEIA_CHECK_LAUNCH("theia_gemm_nt(w4)");
    return THEIA_OK;
}

int theia_gemm_nt_w4_launch(const theia_gemm_args_t* a, int dtype, hipStream_t stream) {
    if (!theia_gemm_nt_w4_supported(a, dtype)) {
        theia_set_error("theia_gemm_nt: tile 256004 (one wave per SIMD) takes bf16 plain matrices with K >= %d, N %% 256 == 0", 32 * w4_hunr(a->act));
        return THEIA_ERR_UNSUPPORTED;
    }
    if (a->act == THEIA_ACT_GELU) return w4_launch_one<THEIA_ACT_GELU>(a, stream);
    return w4_launch_one<THEIA_ACT_NONE>(a, stream);
}
