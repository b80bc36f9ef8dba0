// Pipelined NT implicit GEMM: 4-deep LDS ring of HALF k-tiles filled by LDS-DMA (global_load_lds_dwordx4), counted
// vmcnt waits and raw s_barrier, so three half-tiles of operands are always in flight while one is being multiplied.
//
// Why: the 2-stage kernel of gemm.hip issues tile k+1's loads at the top of iteration k and drains them (vmcnt(0)) at
// its bottom, so every iteration pays max(MFMA time, operand-fetch latency); rocprofv3 shows 42 % of wave cycles parked
// in s_waitcnt/s_barrier at 28 % MFMA utilisation (profiles/r01_gemm_pmc.txt).  Here a half-tile has three iterations
// to land before it is needed.
//
// LDS image of one ring slot: [BM + BN rows][64 B] (32 bf16 / 16 f32 of k per row); the 16-byte chunk index is XORed
// with f(row) = (4 - ((row>>2)&3)) & 3, which makes every ds_read_b128 lane group (the hardware's non-contiguous
// 16-lane groups) hit 16 distinct 16-byte bank slots.  LDS-DMA writes lane-linearly, so the swizzle is applied to the
// SOURCE chunk; out-of-image taps / out-of-range rows read a page of zeros.
#include "gemm_tile.h"

__device__ uint4 g_ring_zero_page[16];

__device__ __forceinline__ int ring_f(int row) { return (4 - ((row >> 2) & 3)) & 3; }

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void gemm_nt_ring_kernel(const theia_gemm_args_t p) {
    constexpr int NSTAGE = 4;
    constexpr int HKT = 64 / (int)sizeof(T);   // k elements per half-tile row
    constexpr int EPC = 16 / (int)sizeof(T);
    constexpr int NTHR = 64 * WAVES_M * WAVES_N;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int FM = WM / 16, FN = WN / 16;
    constexpr int SRP = NTHR / 4;               // rows staged per pass
    constexpr int NPA = BM / SRP, NPB = BN / SRP;
    constexpr int LPH = NPA + NPB;              // LDS-DMA instructions per thread per half-tile
    constexpr int STAGE = (BM + BN) * 64;
    static_assert(BM % SRP == 0 && BN % SRP == 0 && SRP % 16 == 0, "staging geometry");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const theia_rowmap_t& mp = p.map;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tile = gt_xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const T* __restrict__ A = reinterpret_cast<const T*>(p.a);
    const T* __restrict__ W = reinterpret_cast<const T*>(p.w);

    const int st_chunk = tid & 3, st_row = tid >> 2;
    const int lchunk = st_chunk ^ ring_f(st_row);
    const int R = mp.rows_h * mp.rows_w;
    int64_t a_base[NPA];
    int a_iy0[NPA], a_ix0[NPA];
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
        const int m = m0 + st_row + SRP * i;
        if (m < p.M) {
            const int img = m / R, rem = m - img * R;
            const int ry = rem / mp.rows_w, rx = rem - ry * mp.rows_w;
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

    gt_f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = (gt_f32x4){0.f, 0.f, 0.f, 0.f};

    const int nh = (p.K + HKT - 1) / HKT;
    const uint64_t zp = reinterpret_cast<uint64_t>(g_ring_zero_page);
    // issue piece q (0..LPH-1) of half-tile hh into ring slot `slot`; branch-free (bit-mask select of the source address)
    auto issue_piece = [&](int q, int c, bool cok, int dy, int dx, int64_t wcol, char* sa, char* sb) {
        uint64_t src;
        char* dst;
        if (q < NPA) {
            const int iy = a_iy0[q] + dy, ix = a_ix0[q] + dx;
            const bool ok = cok & (iy >= 0) & (iy < mp.in_h) & (ix >= 0) & (ix < mp.in_w);
            const uint64_t pa = reinterpret_cast<uint64_t>(A + a_base[q] + (int64_t)(iy * mp.in_w + ix) * mp.in_c + c);
            const uint64_t msk = 0ull - (uint64_t)ok;
            src = (pa & msk) | (zp & ~msk);
            dst = sa + q * (SRP * 64);
        } else {
            const int i = q - NPA;
            const bool ok = cok & w_ok[i];
            const uint64_t pw = reinterpret_cast<uint64_t>(W + w_base[i] + wcol);
            const uint64_t msk = 0ull - (uint64_t)ok;
            src = (pw & msk) | (zp & ~msk);
            dst = sb + i * (SRP * 64);
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    // prologue: half-tiles 0..2 (clamped: short K re-fetches its last half-tile, which is never read)
#pragma unroll
    for (int h = 0; h < NSTAGE - 1; ++h) {
        const int k0 = min(h, nh - 1) * HKT;
        const int tap = k0 / mp.in_c;
        const int c = k0 - tap * mp.in_c + lchunk * EPC;
        char* sa = smem + h * STAGE + uwave * (16 * 64);
#pragma unroll
        for (int q = 0; q < LPH; ++q)
            issue_piece(q, c, c < mp.in_c, mp.dy[tap], mp.dx[tap], (int64_t)mp.wslot[tap] * mp.in_c + c, sa, sa + BM * 64);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPH) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    const int frow = lane & 15, fg = lane >> 4;
    for (int h = 0; h < nh; ++h) {
        // prefetch half-tile h+3 (clamped) into the slot consumed in iteration h-1, one piece per MFMA group
        const int k0 = min(h + NSTAGE - 1, nh - 1) * HKT;
        const int tap = k0 / mp.in_c;
        const int c = k0 - tap * mp.in_c + lchunk * EPC;
        const bool cok = c < mp.in_c;
        const int dy = mp.dy[tap], dx = mp.dx[tap];
        const int64_t wcol = (int64_t)mp.wslot[tap] * mp.in_c + c;
        char* na = smem + ((h + NSTAGE - 1) & (NSTAGE - 1)) * STAGE + uwave * (16 * 64);
        char* nb = na + BM * 64;
        const char* sa = smem + (h & (NSTAGE - 1)) * STAGE;
        const char* sb = sa + BM * 64;
        uint4 fb[FN];
#pragma unroll
        for (int i = 0; i < FN; ++i) {
            const int row = wn * WN + i * 16 + frow;
            fb[i] = *reinterpret_cast<const uint4*>(sb + row * 64 + ((fg ^ ring_f(row)) << 4));
        }
#pragma unroll
        for (int j = 0; j < FM; ++j) {
            const int row = wm * WM + j * 16 + frow;
            const uint4 fa = *reinterpret_cast<const uint4*>(sa + row * 64 + ((fg ^ ring_f(row)) << 4));
#pragma unroll
            for (int i = 0; i < FN; ++i) GtMma<T>::run(acc[i][j], fb[i], fa);
            static_assert(LPH <= FM, "one LDS-DMA piece per MFMA group at most");
            if ((j * LPH) / FM != ((j + 1) * LPH) / FM) issue_piece((j * LPH) / FM, c, cok, dy, dx, wcol, na, nb);
        }
        // half-tile h+1 must have landed (the two younger ones stay in flight); then everyone is done reading slot h
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPH) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the clamped tail prefetches before LDS is reused
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    float* ep = reinterpret_cast<float*>(smem) + wave * ((WM > 64 ? 64 : WM) * (WN + 4));
    gt_epilogue<T, WM, WN>(acc, ep, p, m0 + wm * WM, n0 + wn * WN, lane);
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N>
static int launch_ring(const theia_gemm_args_t* a, hipStream_t stream) {
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int NTHR = 64 * WAVES_M * WAVES_N;
    constexpr int stage_bytes = 4 * (BM + BN) * 64;
    constexpr int ep_bytes = WAVES_M * WAVES_N * (WM > 64 ? 64 : WM) * (WN + 4) * 4;
    constexpr int lds = stage_bytes > ep_bytes ? stage_bytes : ep_bytes;
    auto kern = gemm_nt_ring_kernel<T, BM, BN, WAVES_M, WAVES_N>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    const int tiles = cdiv_i(a->M, BM) * cdiv_i(a->N, BN);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(NTHR), lds, stream, *a);
    THEIA_CHECK_LAUNCH("theia_gemm_nt(ring)");
    return THEIA_OK;
}

// called by theia_gemm_nt (gemm.hip) after argument validation; tile = BM*1000 + BN
int theia_gemm_nt_ring_launch(const theia_gemm_args_t* a, int dtype, int tile, hipStream_t stream) {
    if (dtype == THEIA_BF16) {
        if (tile == 256256) return launch_ring<bf16_t, 256, 256, 2, 4>(a, stream);
        if (tile == 128128) return launch_ring<bf16_t, 128, 128, 2, 2>(a, stream);
        return launch_ring<bf16_t, 128, 64, 2, 2>(a, stream);
    }
    if (tile == 128064) return launch_ring<float, 128, 64, 2, 2>(a, stream);
    return launch_ring<float, 128, 128, 2, 2>(a, stream);
}
