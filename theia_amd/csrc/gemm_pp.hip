// Ping-pong NT implicit GEMM for the 256x256 tile (8 wave64 = two groups of four, one wave of each group per SIMD).
//
// The two wave groups run the same program ONE BARRIER APART, and the program alternates two kinds of segments:
//     R(P): ds_read_b128 the operand fragments of phase P              |  M(P): 16 x v_mfma_f32_16x16x32_bf16
// so whenever group 0 is in an M segment group 1 is in an R segment and vice versa: the matrix pipe of every SIMD is fed by
// one wave while the SIMD's other wave fetches its next fragments (rocprofv3 on the lock-step kernels: 28 % MFMA busy, 42 %
// of wave cycles parked in s_waitcnt/s_barrier).  Operands arrive by LDS-DMA (global_load_lds_dwordx4) into a 4-deep ring
// of HALF k-tiles (32 bf16 / 16 f32 of k per row, 64-byte rows, source-side XOR swizzle); the four pieces of half-tile h+3
// are issued in R(h), right after the fragment reads (an LDS-DMA instruction costs ~60-100 issue cycles: between a group's
// own MFMAs that is lost matrix-pipe time, in the R segment it overlaps the OTHER group's MFMAs; +5 % measured), and only
// counted waits are used:
//     end of R(h):  s_waitcnt vmcnt(8)  -> half-tile h+1 has landed (h+2 and the just-issued h+3 stay in flight)
// One phase per half-tile: R(h) reads 4 B + 8 A fragments, M(h) issues 32 MFMAs (cycle trace, tools/pp_trace.hip: R is
// ~720 cycles, M ~630, so the R segment -- 12 ds_read_b128 + 4 LDS-DMA issues -- sets the interval).
//
// Hazards (interval k = time between barrier k and k+1; group 0 runs segment k in interval k, group 1 segment k-1):
//   WAR  ring slot (h+3)&3 = (h-1)&3 is refilled from R(h) on (interval >= 2h); its last readers are R(h-1) of group 0
//        (interval 2h-2) and of group 1 (interval 2h-1), both closed by s_waitcnt lgkmcnt(0) before their barrier.
//   RAW  half-tile h+1 is first read in interval 2h+2 (group 0, R(h+1)); every wave has waited for its own pieces of it at
//        the end of its R(h) (interval 2h / 2h+1), i.e. before barrier 2h+2.
#include <type_traits>
#include "gemm_tile.h"

// zeros read by out-of-range rows / taps: such a lane's source pointer is the page start and advances with the k offset
// inside a tap like every other lane's, so the page covers one tap's row (in_c elements <= 16 KiB, checked by the dispatch)
__device__ uint4 g_pp_zero_page[1024 + 1];

// Ablation switches for tools/pp_trace.hip (timing only, results are wrong): -DPP_EXP_NOMFMA / -DPP_EXP_NODMA / -DPP_EXP_NOLDS
// drop the MFMAs / the LDS-DMA operand stream / the fragment reads from the main loop.  One round of 255 tiles with
// K = 6912 (255 CUs busy): everything 220 us, LDS-DMA only 172 us, MFMA only 121 us, and LDS-DMA only on 24 CUs 120 us --
// the L2 -> LDS operand stream (32 KiB per CU per half-tile, ~10 TB/s over the chip at most, 64- or 128-byte rows alike),
// not the matrix pipe, bounds this kernel at full occupancy.
// Optional cycle trace (tools/pp_trace.hip builds this file with -DPP_TRACE): s_memtime stamps of block 0 at the segment
// boundaries of iterations PP_TRACE_H0 .. PP_TRACE_H0+3, one row per wave.  Compiled out of the library.
#ifdef PP_TRACE
__device__ unsigned long long g_pp_trace[8][4][2][5];
__device__ unsigned long long g_pp_phase[2][8][7];
#define PP_PHASE(k)                                                                  \
    if ((blockIdx.x & 255) == 0 && blockIdx.x < 512 && (threadIdx.x & 63) == 0)      \
        g_pp_phase[blockIdx.x >> 8][threadIdx.x >> 6][k] = __builtin_readcyclecounter();
#define PP_STAMP(k)                                                                                  \
    if (blockIdx.x == 0 && h >= PP_TRACE_H0 && h < PP_TRACE_H0 + 4 && lane == 0)                      \
        g_pp_trace[wave][h - PP_TRACE_H0][0][k] = __builtin_readcyclecounter();
#define PP_STAMP2(k)                                                                                 \
    if (blockIdx.x == 0 && h >= PP_TRACE_H0 && h < PP_TRACE_H0 + 4 && lane == 0)                      \
        g_pp_trace[wave][h - PP_TRACE_H0][1][k] = __builtin_readcyclecounter();
#else
#define PP_STAMP(k)
#define PP_STAMP2(k)
#define PP_PHASE(k)
#endif

__device__ __forceinline__ int pp_f(int row) { return (4 - ((row >> 2) & 3)) & 3; }

// CXXR: A/B switch (THEIA_PP_READS=cxx) -- fragment reads as plain C++ LDS loads, the form in which hipcc puts `s_waitcnt vmcnt(0)`
// in front of them (every LDS-DMA in flight is drained at the top of each iteration)
template <typename T, bool CXXR = false, bool SUMS = false>
__global__ __launch_bounds__(512) void gemm_nt_pp_kernel(const theia_gemm_args_t p) {
    constexpr int BM = 256, BN = 256, WAVES_N = 4;
    constexpr int NSTAGE = 4;
    constexpr int HKT = 64 / (int)sizeof(T);
    constexpr int EPC = 16 / (int)sizeof(T);
    constexpr int WM = 128, WN = 64, FM = 8, FN = 4;
    constexpr int SRP = 128;                    // rows staged per pass (512 threads x 16 B = 128 rows of 64 B)
    constexpr int NPA = BM / SRP, NPB = BN / SRP;
    constexpr int LPH = NPA + NPB;              // 4 LDS-DMA pieces per thread per half-tile
    constexpr int STAGE = (BM + BN) * 64;
    static_assert(LPH == 4, "the counted waits assume 4 pieces per thread per half-tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;  // wm = wave group (0: waves 0-3, 1: waves 4-7)
    const int ugroup = uwave >> 2;
    const theia_rowmap_t& mp = p.map;
    const int tiles_n = (p.N + BN - 1) / BN;
    PP_PHASE(0)
    const int tile = gt_xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const T* __restrict__ A = reinterpret_cast<const T*>(p.a);
    const T* __restrict__ W = reinterpret_cast<const T*>(p.w);

    const int st_chunk = tid & 3, st_row = tid >> 2;
    const int lchunk = st_chunk ^ pp_f(st_row);
    const int R = mp.rows_h * mp.rows_w;
    const float rcp_R = 1.0f / (float)R, rcp_w = 1.0f / (float)mp.rows_w;
    int64_t a_base[NPA];
    int a_iy0[NPA], a_ix0[NPA];
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
        const int m = m0 + st_row + SRP * i;
        if (m < p.M) {
            int rem, rx;
            const int img = gt_divmod(m, R, rcp_R, rem);
            const int ry = gt_divmod(rem, mp.rows_w, rcp_w, rx);
            a_base[i] = (int64_t)img * mp.in_batch_stride + mp.in_offset;
            a_iy0[i] = ry * mp.in_sy;
            a_ix0[i] = rx * mp.in_sx;
        } else {
            a_base[i] = 0;
            a_iy0[i] = -(1 << 28);
            a_ix0[i] = 0;
        }
    }
    int64_t w_base[NPB];
    bool w_ok[NPB];
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
        const int n = n0 + st_row + SRP * i;
        w_ok[i] = n < p.N;
        w_base[i] = (int64_t)n * p.ldw;
    }

    gt_f32x4 acc[FN][FM];  // zeroed below, behind the prologue's LDS-DMA issues (128 v_mov per wave: hidden in the operands' latency)

    const int nh = (p.K + HKT - 1) / HKT;   // host guarantees K % HKT == 0 for this kernel
    const uint64_t zp = reinterpret_cast<uint64_t>(g_pp_zero_page);
    // Per-tap source pointers of this thread's LDS-DMA pieces (recomputed only when the prefetch stream enters a new tap, a
    // wave-uniform event): inside the M segments a piece costs one masked 64-bit add + the LDS-DMA issue.
    uint64_t src_ptr[LPH];
    auto set_tap = [&](int tap) {
        const int dy = mp.dy[tap], dx = mp.dx[tap];
        const int64_t wcol = (int64_t)mp.wslot[tap] * mp.in_c + lchunk * EPC;
#pragma unroll
        for (int q = 0; q < NPA; ++q) {
            const int iy = a_iy0[q] + dy, ix = a_ix0[q] + dx;
            const bool ok = (iy >= 0) & (iy < mp.in_h) & (ix >= 0) & (ix < mp.in_w);
            const uint64_t pa = reinterpret_cast<uint64_t>(A + a_base[q] + (int64_t)(iy * mp.in_w + ix) * mp.in_c + lchunk * EPC);
            const uint64_t msk = 0ull - (uint64_t)ok;
            src_ptr[q] = (pa & msk) | (zp & ~msk);
        }
#pragma unroll
        for (int i = 0; i < NPB; ++i) {
            const uint64_t pw = reinterpret_cast<uint64_t>(W + w_base[i] + wcol);
            const uint64_t msk = 0ull - (uint64_t)w_ok[i];
            src_ptr[NPA + i] = (pw & msk) | (zp & ~msk);
        }
    };
    auto issue_piece = [&](int q, uint64_t coff, char* sa, char* sb) {
        const uint64_t src = src_ptr[q] + coff;
        char* dst = q < NPA ? sa + q * (SRP * 64) : sb + (q - NPA) * (SRP * 64);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    const int hpt = mp.in_c / HKT;  // half-tiles per tap
    PP_PHASE(1)
    int cur_tap = 0, next_tap_h = hpt;  // prefetch stream state: tap of half-tile hp, first half-tile of the next tap
    set_tap(0);
    // prologue: half-tiles 0..2 (clamped for very short K; the duplicates are never read)
#pragma unroll
    for (int h = 0; h < NSTAGE - 1; ++h) {
        const int hp = min(h, nh - 1);
        if (hp >= next_tap_h) {
            ++cur_tap;
            next_tap_h += hpt;
            set_tap(cur_tap);
        }
        const uint64_t coff = (uint64_t)(hp - cur_tap * hpt) * 64;
        char* sa = smem + h * STAGE + uwave * (16 * 64);
#pragma unroll
        for (int q = 0; q < LPH; ++q) issue_piece(q, coff, sa, sa + BM * 64);
    }
    PP_PHASE(2)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) {
            acc[i][j] = (gt_f32x4){0.f, 0.f, 0.f, 0.f};
            asm volatile("" : "+v"(acc[i][j]));  // materialise the zeros HERE (the compiler otherwise sinks them behind the wait)
        }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPH) : "memory");  // half-tile 0 landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (ugroup == 1) {  // group 1 runs one barrier behind group 0
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }

    PP_PHASE(3)
    const int frow = lane & 15, fg = lane >> 4;
    // Fragment addresses: rows 16 apart share the swizzle (pp_f looks at bits 2..3 of the row), so the 8 A fragments / 4 B
    // fragments of a wave are 1 KiB apart: one lane-dependent offset each + immediates.  The reads are inline asm (see
    // gt_ds_read128: a C++ load here would make the compiler drain every in-flight LDS-DMA at the top of each iteration).
    const uint32_t smem_base = gt_lds_addr(smem);
    const uint32_t lane_a = smem_base + (wm * WM + frow) * 64 + ((fg ^ pp_f(frow)) << 4);
    const uint32_t lane_b = smem_base + BM * 64 + (wn * WN + frow) * 64 + ((fg ^ pp_f(frow)) << 4);
    gt_u32x4 fb[FN], fa[FM];
#ifdef PP_EXP_NOLDS
    for (int j = 0; j < FM; ++j) fa[j] = (gt_u32x4){(unsigned)tid, (unsigned)tid, (unsigned)tid, (unsigned)tid};
    for (int i = 0; i < FN; ++i) fb[i] = (gt_u32x4){(unsigned)tid, (unsigned)tid, (unsigned)tid, (unsigned)tid};
#endif
    for (int h = 0; h < nh; ++h) {
        const int hp = min(h + NSTAGE - 1, nh - 1);  // half-tile prefetched during this iteration (clamped at the tail)
        if (hp >= next_tap_h) {                       // wave-uniform, once per in_c/HKT iterations
            ++cur_tap;
            next_tap_h += hpt;
            set_tap(cur_tap);
        }
        const uint64_t coff = (uint64_t)(hp - cur_tap * hpt) * 64;
        char* na = smem + ((h + NSTAGE - 1) & (NSTAGE - 1)) * STAGE + uwave * (16 * 64);
        char* nb = na + BM * 64;
        const uint32_t soff = (uint32_t)(h & (NSTAGE - 1)) * STAGE;
        {
            // ---------------- R(h): 12 fragment reads + the 4 LDS-DMA pieces of half-tile h+3
            PP_STAMP(0)
#ifndef PP_EXP_NOLDS
            const uint32_t ab = lane_b + soff, aa = lane_a + soff;
            if constexpr (CXXR) {
#pragma unroll
                for (int i = 0; i < FN; ++i) fb[i] = *reinterpret_cast<const gt_u32x4*>(smem + (ab - smem_base) + i * 1024);
#pragma unroll
                for (int j = 0; j < FM; ++j) fa[j] = *reinterpret_cast<const gt_u32x4*>(smem + (aa - smem_base) + j * 1024);
            } else {
            gt_ds_read128<0>(fb[0], ab);
            gt_ds_read128<1024>(fb[1], ab);
            gt_ds_read128<2048>(fb[2], ab);
            gt_ds_read128<3072>(fb[3], ab);
            gt_ds_read128<0>(fa[0], aa);
            gt_ds_read128<1024>(fa[1], aa);
            gt_ds_read128<2048>(fa[2], aa);
            gt_ds_read128<3072>(fa[3], aa);
            gt_ds_read128<4096>(fa[4], aa);
            gt_ds_read128<5120>(fa[5], aa);
            gt_ds_read128<6144>(fa[6], aa);
            gt_ds_read128<7168>(fa[7], aa);
            }
#endif
            PP_STAMP2(0)
#ifndef PP_GLDS_IN_M  // default: the LDS-DMA pieces are issued in the R segment (their ~60-100 issue cycles each overlap the
                      // OTHER group's MFMAs); -DPP_GLDS_IN_M puts them between this group's MFMAs (5% slower, 2 A/B runs)
#ifndef PP_EXP_NODMA
#pragma unroll
            for (int q = 0; q < LPH; ++q) issue_piece(q, coff, na, nb);
#endif
            PP_STAMP2(1)
            gt_wait_lds(fb, fa);
            PP_STAMP(1)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPH) : "memory");  // half-tile h+1 landed; h+2, h+3 may be in flight
#else
            gt_wait_lds(fb, fa);
            PP_STAMP(1)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPH) : "memory");  // half-tile h+1 landed; h+2 may still be in flight
#endif
            __builtin_amdgcn_sched_barrier(0);
            PP_STAMP(2)
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---------------- M(h): 32 MFMAs
            PP_STAMP(3)
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < FM; ++j) {
#pragma unroll
#ifdef PP_EXP_NOMFMA
                for (int i = 0; i < FN; ++i) asm volatile("" ::"v"(fb[i]), "v"(fa[j]));
#else
                for (int i = 0; i < FN; ++i) gt_mma<T>(acc[i][j], fb[i], fa[j]);
#endif
#ifdef PP_GLDS_IN_M
                if ((j & 1) == 0) issue_piece(j >> 1, coff, na, nb);
#endif
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            PP_STAMP(4)
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (ugroup == 0) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0), as a builtin so that the compiler KNOWS no LDS-DMA is pending in the epilogue  // drain the clamped tail prefetches before LDS is reused
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    PP_PHASE(4)
    float* ep = reinterpret_cast<float*>(smem) + wave * (64 * (WN + 4));
    using OutT = typename std::conditional<sizeof(T) == 1, bf16_t, T>::type;  // fp8 operands: bf16 out, accumulators rescaled
    const bool prefetch = sizeof(OutT) == 2 && (p.resid != nullptr || p.act == THEIA_ACT_MUL_DGELU || p.act == THEIA_ACT_MUL_DRELU);
    // (the statistics-emitting instantiation keeps ONE epilogue that decides at run time: with two the register allocation of its
    // main loop spills -- the stride-2 transposed convolutions ran 2x slower)
    if constexpr (SUMS) gt_epilogue<OutT, WM, WN, true, sizeof(T) == 1, 4, 0>(acc, ep, p, m0 + wm * WM, n0 + wn * WN, lane);
    else if (prefetch) gt_epilogue<OutT, WM, WN, false, sizeof(T) == 1, 4, 1>(acc, ep, p, m0 + wm * WM, n0 + wn * WN, lane);
    else gt_epilogue<OutT, WM, WN, false, sizeof(T) == 1, 4, 0>(acc, ep, p, m0 + wm * WM, n0 + wn * WN, lane);
    PP_PHASE(5)
#ifdef PP_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    PP_PHASE(6)
}

int theia_gemm_nt_pp_launch(const theia_gemm_args_t* a, int dtype, hipStream_t stream) {
    constexpr int stage_bytes = 4 * (256 + 256) * 64;
    constexpr int ep_bytes = 8 * 64 * (64 + 4) * 4;
    constexpr int lds = stage_bytes > ep_bytes ? stage_bytes : ep_bytes;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_pp_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_pp_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    const int tiles = cdiv_i(a->M, 256) * cdiv_i(a->N, 256);
    static int cxx_reads = -1;
    if (cxx_reads < 0) {
        const char* e = getenv("THEIA_PP_READS");
        cxx_reads = (e != nullptr && strcmp(e, "cxx") == 0) ? 1 : 0;
        if (cxx_reads) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_pp_kernel<bf16_t, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    }
    if (dtype == THEIA_FP8) {  // fp8 e4m3 operands (bytes), bf16 output
        static bool attr8 = false;
        if (!attr8) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_pp_kernel<fp8_t, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_pp_kernel<fp8_t, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            attr8 = true;
        }
        if (a->ln_sums != nullptr) hipLaunchKernelGGL((gemm_nt_pp_kernel<fp8_t, false, true>), dim3(tiles), dim3(512), lds, stream, *a);
        else hipLaunchKernelGGL((gemm_nt_pp_kernel<fp8_t, false, false>), dim3(tiles), dim3(512), lds, stream, *a);
    } else if (a->ln_sums != nullptr) {
        static bool attr2 = false;
        if (!attr2) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_pp_kernel<bf16_t, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_pp_kernel<float, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            attr2 = true;
        }
        if (dtype == THEIA_BF16) hipLaunchKernelGGL((gemm_nt_pp_kernel<bf16_t, false, true>), dim3(tiles), dim3(512), lds, stream, *a);
        else hipLaunchKernelGGL((gemm_nt_pp_kernel<float, false, true>), dim3(tiles), dim3(512), lds, stream, *a);
    } else if (dtype == THEIA_BF16 && cxx_reads) hipLaunchKernelGGL((gemm_nt_pp_kernel<bf16_t, true>), dim3(tiles), dim3(512), lds, stream, *a);
    else if (dtype == THEIA_BF16) hipLaunchKernelGGL(gemm_nt_pp_kernel<bf16_t>, dim3(tiles), dim3(512), lds, stream, *a);
    else hipLaunchKernelGGL(gemm_nt_pp_kernel<float>, dim3(tiles), dim3(512), lds, stream, *a);
    THEIA_CHECK_LAUNCH("theia_gemm_nt(pp)");
    return THEIA_OK;
}
