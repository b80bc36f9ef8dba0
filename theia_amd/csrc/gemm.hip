// MFMA GEMM kernels of the Theia hot path (gfx950 / CDNA4) and the C-ABI dispatch of theia_gemm_nt / theia_gemm_wgrad.
//
//   gemm_nt_kernel    out[m,n] = epi( sum_k A[m,k] W[n,k] )    A gathered through a theia_rowmap_t (implicit GEMM for
//                     Linear / Conv3x3 / ConvTranspose3x3 and every data-gradient).  This file holds the 2-stage
//                     kernel used for 128x128 / 128x64 tiles (f32 parity path, narrow N, small grids); 256x256 tiles go
//                     to the ping-pong kernel of gemm_pp.hip.
//   gemm_wgrad_kernel slab[s][n][k] = sum_{m in split s} dY[m,n] A[m,k]   (weight gradients, split over M): f32 path and
//                     shapes the ping-pong kernel of gemm_wgrad_pp.hip does not take.
//
// 2-stage NT kernel: 256 threads = 4 wave64; LDS tiles are [rows][128 B] (64 bf16 / 32 f32 of K per row) filled by LDS-DMA
// (global_load_lds_dwordx4, lane-linear destination) with the 16-byte chunk index XOR-swizzled by (row>>1)&7 on the SOURCE
// side, so every ds_read_b128 fragment read is bank-conflict free; out-of-image taps / out-of-range rows read a page of
// zeros (no predication, no divergent branches in the K loop).
// bf16: v_mfma_f32_16x16x32_bf16 (one 16-byte chunk per lane = 8 consecutive k).
// f32 : v_mfma_f32_16x16x4_f32, four per chunk (k order inside a tile is permuted identically for both operands, which
//       leaves the sum unchanged up to fp32 association).
// The MFMA is issued with the WEIGHT fragment as its A operand and the ACTIVATION fragment as its B operand, so a lane ends
// up with 4 consecutive n for one m: the accumulator tile goes to LDS with ds_write_b128 and is re-read row-contiguously,
// giving a fully vectorised epilogue (gemm_tile.h).
#include "gemm_epi_direct.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_v;
typedef gt_f32x4 f32x4_v;
typedef __attribute__((ext_vector_type(4))) short s16x4_v;

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    __device__ static __forceinline__ void run(f32x4_v& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_v, a), __builtin_bit_cast(bf16x8_v, b),
                                                      acc, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    __device__ static __forceinline__ void run(f32x4_v& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
    }
};

// XCD-aware bijective block remap: hardware places block b on XCD b % 8; give each XCD a contiguous range of tiles
// so that neighbouring tiles (which share operand panels) hit the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + idx;
}

__device__ __forceinline__ int swz_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// ================================================================================================
// NT implicit GEMM
// ================================================================================================
int theia_gemm_nt_pp_launch(const theia_gemm_args_t* a, int dtype, hipStream_t stream);               // gemm_pp.hip
int theia_gemm_nt_pp_bm(const theia_gemm_args_t* a, int dtype);                                        // rows per tile it would use
struct conv_taps_t;
bool theia_gemm_conv_pp_match(const theia_gemm_args_t* a, int dtype, conv_taps_t* out);                 // gemm_conv_pp.hip
int theia_gemm_conv_pp_launch(const theia_gemm_args_t* a, int dtype, hipStream_t stream);

__device__ uint4 g_zero_page[16];  // 256 B of zeros: source of out-of-range operand chunks

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, bool SUMS = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void gemm_nt_kernel(const theia_gemm_args_t p) {
    constexpr int KT = 128 / (int)sizeof(T);   // k elements per LDS row
    constexpr int EPC = 16 / (int)sizeof(T);   // elements per 16-byte chunk
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int FM = WM / 16, FN = WN / 16;
    constexpr int NTHR = 64 * WAVES_M * WAVES_N;  // 256 (128x128 / 128x64 tiles) or 512 (256x256 tile)
    constexpr int SRP = NTHR / 8;                  // rows staged per pass (8 x 16-byte chunks per row)
    constexpr int NPA = BM / SRP, NPB = BN / SRP;
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int EP_PITCH = WN + 4;  // floats
    static_assert(BM % SRP == 0 && BN % SRP == 0 && (SRP % 16) == 0, "staging geometry");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const theia_rowmap_t& mp = p.map;

    const int tiles_n = (p.N + BN - 1) / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;

    const T* __restrict__ A = reinterpret_cast<const T*>(p.a);
    const T* __restrict__ W = reinterpret_cast<const T*>(p.w);

    // ---- per-thread staging rows (fixed for the whole K loop) ----
    const int st_chunk = tid & 7, st_row = tid >> 3;
    const int R = mp.rows_h * mp.rows_w;
    int64_t a_base[NPA];
    int a_iy0[NPA], a_ix0[NPA];
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
        const int m = m0 + st_row + SRP * i;
        if (m < p.M) {
            int rem, rx;
            const int img = gt_divmod(m, R, 1.0f / (float)R, rem);
            const int ry = gt_divmod(rem, mp.rows_w, 1.0f / (float)mp.rows_w, rx);
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

    // LDS-DMA staging: global_load_lds_dwordx4 writes LDS lane-linearly (wave-uniform base + lane*16), so the XOR swizzle is
    // applied on the SOURCE side: the lane that fills slot s of row r fetches logical chunk s ^ ((r>>1)&7).  Rows that are
    // out of range / taps that fall outside the image read a 256-byte page of zeros instead (no predication).
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    const int lchunk = st_chunk ^ ((st_row >> 1) & 7);
    auto issue_tile = [&](int kt, int stage) {
        const int k0 = kt * KT;
        const int tap = k0 / mp.in_c;
        const int c = k0 - tap * mp.in_c + lchunk * EPC;
        const bool cok = c < mp.in_c;
        const int dy = mp.dy[tap], dx = mp.dx[tap];
        const int64_t wcol = (int64_t)mp.wslot[tap] * mp.in_c + c;
        char* sa = smem + stage * STAGE + uwave * (8 * 128);
        char* sb = sa + BM * 128;
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            const int iy = a_iy0[i] + dy, ix = a_ix0[i] + dx;
            const bool ok = cok && iy >= 0 && iy < mp.in_h && ix >= 0 && ix < mp.in_w;
            const T* src = ok ? A + a_base[i] + (int64_t)(iy * mp.in_w + ix) * mp.in_c + c : reinterpret_cast<const T*>(g_zero_page);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(sa + i * (SRP * 128)), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NPB; ++i) {
            const T* src = (cok && w_ok[i]) ? W + w_base[i] + wcol : reinterpret_cast<const T*>(g_zero_page);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(sb + i * (SRP * 128)), 16, 0, 0);
        }
    };

    f32x4_v acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = (f32x4_v){0.f, 0.f, 0.f, 0.f};

    const int nkt = (p.K + KT - 1) / KT;
    issue_tile(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int frow = lane & 15, fg = lane >> 4;
    {
        // Branch-free single-block main loop: the source address of every LDS-DMA piece is selected with bit masks (no
        // divergent branches around the loads) and the last iteration simply re-fetches the last tile into the idle slot
        // instead of branching, so the compiler keeps the accumulators in place across iterations.
        const uint64_t zp = reinterpret_cast<uint64_t>(g_zero_page);
        for (int kt = 0; kt < nkt; ++kt) {
            const int cur = kt & 1;
            const int ktn = min(kt + 1, nkt - 1);
            const int k0 = ktn * KT;
            const int tap = k0 / mp.in_c;
            const int c = k0 - tap * mp.in_c + lchunk * EPC;
            const bool cok = c < mp.in_c;
            const int dy = mp.dy[tap], dx = mp.dx[tap];
            const int64_t wcol = (int64_t)mp.wslot[tap] * mp.in_c + c;
            char* na = smem + (cur ^ 1) * STAGE + uwave * (8 * 128);
            char* nb = na + BM * 128;
            const char* sa = smem + cur * STAGE;
            const char* sb = sa + BM * 128;
            auto issue_piece = [&](int q) {
                uint64_t src;
                char* dst;
                if (q < NPA) {
                    const int iy = a_iy0[q] + dy, ix = a_ix0[q] + dx;
                    const bool ok = cok & (iy >= 0) & (iy < mp.in_h) & (ix >= 0) & (ix < mp.in_w);
                    const uint64_t pa = reinterpret_cast<uint64_t>(A + a_base[q] + (int64_t)(iy * mp.in_w + ix) * mp.in_c + c);
                    const uint64_t msk = 0ull - (uint64_t)ok;
                    src = (pa & msk) | (zp & ~msk);
                    dst = na + q * (SRP * 128);
                } else {
                    const int i = q - NPA;
                    const bool ok = cok & w_ok[i];
                    const uint64_t pw = reinterpret_cast<uint64_t>(W + w_base[i] + wcol);
                    const uint64_t msk = 0ull - (uint64_t)ok;
                    src = (pw & msk) | (zp & ~msk);
                    dst = nb + i * (SRP * 128);
                }
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            };
            // all pieces first: with two LDS stages the prefetch must land within this iteration, so it is issued as early
            // as possible (spreading it between the MFMA groups measured 30 % slower: the last piece's latency is exposed)
#pragma unroll
            for (int q = 0; q < NPA + NPB; ++q) issue_piece(q);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                uint4 fb[FN];
#pragma unroll
                for (int i = 0; i < FN; ++i) fb[i] = *reinterpret_cast<const uint4*>(sb + swz_off(wn * WN + i * 16 + frow, kk * 4 + fg));
#pragma unroll
                for (int j = 0; j < FM; ++j) {
                    const uint4 fa = *reinterpret_cast<const uint4*>(sa + swz_off(wm * WM + j * 16 + frow, kk * 4 + fg));
#pragma unroll
                    for (int i = 0; i < FN; ++i) Mma<T>::run(acc[i][j], fb[i], fa);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tile kt+1 has landed
            __syncthreads();
        }
    }
    // ---- epilogue (shared with the ping-pong kernel): wave-private LDS round trip, 16-byte vector global accesses ----
    float* ep = reinterpret_cast<float*>(smem) + wave * ((WM > 64 ? 64 : WM) * EP_PITCH);
    gt_epilogue<T, WM, WN, SUMS>(acc, ep, p, m0 + wm * WM, n0 + wn * WN, lane);
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, bool SUMS = false>
static int launch_gemm_nt(const theia_gemm_args_t* a, hipStream_t stream) {
    if constexpr (!SUMS) {
        if (a->ln_sums != nullptr) return launch_gemm_nt<T, BM, BN, WAVES_M, WAVES_N, true>(a, stream);
    }
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int NTHR = 64 * WAVES_M * WAVES_N;
    constexpr int stage_bytes = 2 * (BM + BN) * 128;
    constexpr int ep_bytes = WAVES_M * WAVES_N * (WM > 64 ? 64 : WM) * (WN + 4) * 4;
    constexpr int lds = stage_bytes > ep_bytes ? stage_bytes : ep_bytes;
    auto kern = gemm_nt_kernel<T, BM, BN, WAVES_M, WAVES_N, SUMS>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    const int tiles = cdiv_i(a->M, BM) * cdiv_i(a->N, BN);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(NTHR), lds, stream, *a);
    THEIA_CHECK_LAUNCH("theia_gemm_nt");
    return THEIA_OK;
}

static int check_rowmap(const theia_rowmap_t& m, int kt, const char* who) {
    THEIA_CHECK_ARG(m.ntaps >= 1 && m.ntaps <= THEIA_MAX_TAPS, "%s: ntaps=%d out of range", who, m.ntaps);
    THEIA_CHECK_ARG(m.rows_h >= 1 && m.rows_w >= 1 && m.in_h >= 1 && m.in_w >= 1 && m.out_w >= 1, "%s: bad row map grid", who);
    THEIA_CHECK_ARG(m.in_c >= 1, "%s: in_c=%d", who, m.in_c);
    THEIA_CHECK_ARG(m.ntaps == 1 || m.in_c % kt == 0, "%s: in_c=%d must be a multiple of %d for a multi-tap gather", who, m.in_c, kt);
    return THEIA_OK;
}

// Tile chooser (BM*1000 + BN).  128x64 when a 128-wide N tile would waste > 12 % of the MFMA work.  For bf16 the 256x256
// tile (8 waves, 1 block/CU) halves the L2->LDS bytes per flop of the 128x128 tile (2 blocks/CU); the kernel is bound by
// operand-fetch latency x bytes in flight, so the big tile wins whenever it does not waste the grid:
// score = useful fraction of the padded tile grid x fill of the last block wave x relative per-CU rate.
extern "C" int theia_gemm_nt_tile(int M, int N, int dtype) {
    const int t128 = cdiv_i(N, 128) * 128;
    const bool narrow = (t128 - N) * 8 > t128;  // a 128-wide N tile would waste > 12 % of the MFMA work: 128x64 among the 2-stage tiles
    if (dtype != THEIA_BF16) return narrow ? 128064 : 128128;
    static int force = -1;
    if (force < 0) {
        const char* e = getenv("THEIA_GEMM_TILE");
        force = e == nullptr ? 0 : atoi(e);
    }
    auto score = [&](int bm, int bn, int slots, double rate) {
        const double tm = cdiv_i(M, bm), tn = cdiv_i(N, bn);
        const double useful = ((double)M * N) / (tm * bm * tn * bn);
        const double blocks = tm * tn;
        const double fill = blocks / (cdiv_i((long)blocks, slots) * (double)slots);
        return useful * fill * rate;
    };
    // Relative per-CU rates of the kernels behind the tiles, from the bench tables: the persistent ping-pong kernel 1.0 (600-1200 TF on the
    // shapes of the step), the 2-stage 128x128 kernel 0.6 of that, the 2-stage 128x64 kernel 0.25 (DeiT-tiny's N = 192 launches ran it at
    // ~90 TF: 22 % of that model's step until round 4 -- a 256-wide tile that is 3/4 full beats it 3x).  Ties go to the smaller tile.
    const double s256 = score(256, 256, 256, 1.0);
    const double s2 = narrow ? score(128, 64, 512, 0.25) : score(128, 128, 512, 0.6);
    const bool big = force == 256 || (force == 0 && s256 > s2 * 1.0001);
    return big ? 256256 : (narrow ? 128064 : 128128);
}

// Which kernel theia_gemm_nt runs for these arguments: 128128 / 128064 = 2-stage kernel with that tile, 256000 = 2-stage kernel
// with the 256x256 tile (problems the ping-pong kernels do not take), 256256 = ping-pong kernel, 256009 = ping-pong kernel for
// 3x3 convolutions with one image per tile (gemm_conv_pp.hip); negative = error.
static int gemm_nt_plan(const theia_gemm_args_t* a, int dtype) {
    THEIA_CHECK_ARG(a->tile == 0 || a->tile == 128128 || a->tile == 128064 || a->tile == 256256 || a->tile == 320256 || a->tile == 256009,
                    "theia_gemm_nt: bad tile request %d", a->tile);
    const bool want_pp = a->tile == 256256 || a->tile == 320256;
    const int tile = a->tile == 320256 ? 256256 : a->tile != 0 ? a->tile : theia_gemm_nt_tile(a->M, a->N, dtype);
    static int use_pp = -1, use_conv = -1;  // THEIA_GEMM_KERNEL=std / THEIA_CONV_KERNEL=std: A/B switches for timing runs
    if (use_pp < 0) {
        const char* e = getenv("THEIA_GEMM_KERNEL");
        use_pp = (e != nullptr && strcmp(e, "std") == 0) ? 0 : 1;
        e = getenv("THEIA_CONV_KERNEL");
        use_conv = (e != nullptr && strcmp(e, "std") == 0) ? 0 : 1;
    }
    const int hkt = dtype == THEIA_FP8 ? 64 : dtype == THEIA_BF16 ? 32 : 16;  // k elements of one half-tile of the ping-pong kernels
    const int esz = dtype == THEIA_FP8 ? 1 : dtype == THEIA_BF16 ? 2 : 4;
    // what the persistent ping-pong kernel (gemm_pp.hip) takes: half-tile granularity of K and of a tap, one tap's row inside the zero
    // page, rows addressable by its stepping epilogue (gemm_epi_direct.h), no position row-table, a residual only without activation
    const bool has_aux = a->act == THEIA_ACT_MUL_DGELU || a->act == THEIA_ACT_MUL_DRELU;
    const bool pp_ok = a->K % hkt == 0 && a->map.in_c % hkt == 0 && (int64_t)a->map.in_c * esz <= 16384 && a->M < (1 << 24) &&
                       gd_rows_steppable(a->map) && a->rowtab == nullptr && (a->resid == nullptr || a->act == THEIA_ACT_NONE) &&
                       (a->ln_sums == nullptr || a->act == THEIA_ACT_NONE || a->act == THEIA_ACT_RELU) &&
                       (a->tile != 320256 || (dtype == THEIA_BF16 && (a->ln_sums == nullptr || a->map.rows_h * a->map.rows_w >= 160)));
    const bool conv_ok = dtype != THEIA_FP8 && theia_gemm_conv_pp_match(a, dtype, nullptr);
    // The ping-pong kernels' statistics-emitting instantiations carry no residual / aux_in prefetch (with it they spill registers):
    // a launch that wants both -- none of the reference's layers does -- runs on the 2-stage 128x128 kernel.
    const bool sums_and_prefetch = a->ln_sums != nullptr && (a->resid != nullptr || a->act == THEIA_ACT_MUL_DGELU || a->act == THEIA_ACT_MUL_DRELU);
    if (dtype == THEIA_FP8) {  // fp8 operands exist for the ping-pong kernel only
        if (sums_and_prefetch) {
            theia_set_error("theia_gemm_nt(fp8): ln_sums together with resid / aux_in is not available");
            return THEIA_ERR_UNSUPPORTED;
        }
        if (!pp_ok || (a->tile != 0 && a->tile != 256256)) {  // (no 320-row fp8 instantiation)
            theia_set_error("theia_gemm_nt(fp8): needs K and in_c multiples of 64 and the 256x256 ping-pong kernel (K=%d in_c=%d tile=%d)", a->K,
                            a->map.in_c, a->tile);
            return THEIA_ERR_UNSUPPORTED;
        }
        return 256256;
    }
    if (want_pp && !pp_ok) {
        theia_set_error("theia_gemm_nt: the ping-pong kernel needs K and in_c multiples of %d, in_c <= %d, no rowtab, a residual only with act "
                        "NONE, rows it can step through; 320-row tiles are bf16 only (K=%d in_c=%d tile=%d)", hkt,
                        dtype == THEIA_BF16 ? 8192 : 4096, a->K, a->map.in_c, a->tile);
        return THEIA_ERR_UNSUPPORTED;
    }
    if (a->tile == 256009 && !conv_ok) {
        theia_set_error("theia_gemm_nt: tile request 256009 needs a 3x3 stride-1 convolution row map with one 16x16 image per 256-row tile");
        return THEIA_ERR_UNSUPPORTED;
    }
    if (sums_and_prefetch && (a->tile == 256009 || want_pp)) {
        theia_set_error("theia_gemm_nt: ln_sums together with resid / aux_in is not available on the 256x256 ping-pong kernels (tile request %d)", a->tile);
        return THEIA_ERR_UNSUPPORTED;
    }
    if (sums_and_prefetch && tile == 256256) return 128128;
    if (a->tile == 256009) return 256009;
    if (want_pp) return theia_gemm_nt_pp_bm(a, dtype) == 320 ? 320256 : 256256;
    if (tile == 256256) {
        if (use_pp && use_conv && conv_ok) return 256009;
        if (use_pp && pp_ok) return theia_gemm_nt_pp_bm(a, dtype) == 320 ? 320256 : 256256;
        return 256000;
    }
    return tile;
}

extern "C" int theia_gemm_nt_plan(const theia_gemm_args_t* a, int dtype) {
    THEIA_CHECK_ARG(a != nullptr && (dtype == THEIA_F32 || dtype == THEIA_BF16 || dtype == THEIA_FP8), "theia_gemm_nt_plan: bad arguments");
    return gemm_nt_plan(a, dtype);
}

extern "C" int theia_gemm_nt(const theia_gemm_args_t* a, int dtype, void* stream) {
    THEIA_CHECK_ARG(a != nullptr, "theia_gemm_nt: null args");
    THEIA_CHECK_ARG(dtype == THEIA_F32 || dtype == THEIA_BF16 || dtype == THEIA_FP8, "theia_gemm_nt: bad dtype %d", dtype);
    THEIA_CHECK_ARG(a->a && a->w && a->out, "theia_gemm_nt: null operand pointer");
    THEIA_CHECK_ARG(a->M > 0 && a->N > 0 && a->K > 0, "theia_gemm_nt: bad shape M=%d N=%d K=%d", a->M, a->N, a->K);
    const int epc = dtype == THEIA_FP8 ? 16 : dtype == THEIA_BF16 ? 8 : 4;
    THEIA_CHECK_ARG(a->N % 8 == 0, "theia_gemm_nt: N=%d must be a multiple of 8", a->N);
    THEIA_CHECK_ARG(a->K % epc == 0 && a->ldw % epc == 0, "theia_gemm_nt: K=%d / ldw=%d must be multiples of %d", a->K, a->ldw, epc);
    THEIA_CHECK_ARG(a->K == a->map.ntaps * a->map.in_c, "theia_gemm_nt: K=%d != ntaps*in_c=%d", a->K, a->map.ntaps * a->map.in_c);
    THEIA_CHECK_ARG(a->ldo % 8 == 0 && a->map.out_offset % 8 == 0 && a->map.out_batch_stride % 8 == 0,
                    "theia_gemm_nt: output pitch/offset must be multiples of 8 elements");
    THEIA_CHECK_ARG(a->map.in_c % epc == 0 && a->map.in_offset % epc == 0 && a->map.in_batch_stride % epc == 0,
                    "theia_gemm_nt: input pitch/offset must be multiples of %d elements", epc);
    THEIA_CHECK_ARG(a->act >= 0 && a->act <= THEIA_ACT_MUL_DRELU, "theia_gemm_nt: bad act %d", a->act);
    THEIA_CHECK_ARG((a->act != THEIA_ACT_MUL_DGELU && a->act != THEIA_ACT_MUL_DRELU) || a->aux_in, "theia_gemm_nt: act needs aux_in");
    THEIA_CHECK_ARG(a->rowtab == nullptr || a->rowtab_period > 0, "theia_gemm_nt: rowtab_period");
    THEIA_CHECK_ARG(a->ln_sums == nullptr || a->map.rows_h * a->map.rows_w >= 128, "theia_gemm_nt: ln_sums needs >= 128 rows per image");
    THEIA_CHECK_ARG(a->out8 == nullptr || (dtype == THEIA_FP8 && a->out8_scale != nullptr && a->ln_sums == nullptr),
                    "theia_gemm_nt: out8 is an option of THEIA_FP8 launches without ln_sums (with out8_scale)");
    int rc = check_rowmap(a->map, dtype == THEIA_F32 ? 32 : 64, "theia_gemm_nt");
    if (rc) return rc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int plan = gemm_nt_plan(a, dtype);
    if (plan < 0) return plan;
    if (plan == 256009) return theia_gemm_conv_pp_launch(a, dtype, s);
    if (plan == 256256 || plan == 320256) return theia_gemm_nt_pp_launch(a, dtype, s);
    if (dtype == THEIA_BF16) {
        if (plan == 256000) return launch_gemm_nt<bf16_t, 256, 256, 2, 4>(a, s);
        return plan == 128064 ? launch_gemm_nt<bf16_t, 128, 64, 2, 2>(a, s) : launch_gemm_nt<bf16_t, 128, 128, 2, 2>(a, s);
    }
    return plan == 128064 ? launch_gemm_nt<float, 128, 64, 2, 2>(a, s) : launch_gemm_nt<float, 128, 128, 2, 2>(a, s);
}

// ================================================================================================
// weight-gradient GEMM (reduction over rows m; both operands are m-major so fragments need a transpose)
// ================================================================================================
template <typename T> struct WgTile;
template <> struct WgTile<bf16_t> {
    static constexpr int MS = 64;          // rows (m) per step
    static constexpr int PAD = 32;         // bytes of row padding
};
template <> struct WgTile<float> {
    static constexpr int MS = 32;
    static constexpr int PAD = 64;
};

// q = m / d, r = m % d for 0 <= m < 2^24 via a float reciprocal and one correction step (the row decode runs for every
// staged row of every step: integer division made the kernel VALU-bound, 17 VALU instructions per MFMA)
__device__ __forceinline__ void fast_divmod(int m, int d, float rcp, int& q, int& r) {
    q = (int)((float)m * rcp);
    r = m - q * d;
    if (r >= d) { ++q; r -= d; }
    if (r < 0) { --q; r += d; }
}

// BNN: n tile (dY columns), BC: c tile (A columns inside one tap)
template <typename T, int BNN, int BC>
__global__ __launch_bounds__(256) void gemm_wgrad_kernel(const theia_wgrad_args_t p) {
    constexpr int MS = WgTile<T>::MS;
    constexpr int EPC = 16 / (int)sizeof(T);
    constexpr int PY = BNN * (int)sizeof(T) + WgTile<T>::PAD;  // dY tile row pitch (bytes)
    constexpr int PX = BC * (int)sizeof(T) + WgTile<T>::PAD;   // A  tile row pitch
    constexpr int CY = BNN / EPC, CX = BC / EPC;               // chunks per row
    constexpr int NCY = MS * CY / 256, NCX = MS * CX / 256;    // chunks per thread
    constexpr int STAGE = MS * (PY + PX);
    constexpr int WNN = BNN / 2, WCC = BC / 2;                 // per-wave sub-tile (2 x 2 waves)
    constexpr int FNn = WNN / 16, FC = WCC / 16;
    static_assert(NCY >= 1 && NCX >= 1, "tile too small");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wnn = wave >> 1, wcc = wave & 1;
    const theia_rowmap_t& mp = p.map;
    const int tiles_n = (p.N + BNN - 1) / BNN;
    const int tiles_c_per_tap = mp.in_c / BC;
    const int tiles_k = mp.ntaps * tiles_c_per_tap;
    const int ntile = tiles_n * tiles_k;
    const int tile = blockIdx.x % ntile, split = blockIdx.x / ntile;
    const int tn = tile % tiles_n, tk = tile / tiles_n;
    const int tap = tk / tiles_c_per_tap, c0 = (tk - tap * tiles_c_per_tap) * BC;
    const int n0 = tn * BNN;
    const int dy = mp.dy[tap], dx = mp.dx[tap];

    const int nsteps = (p.M + MS - 1) / MS;
    const int per = (nsteps + p.splits - 1) / p.splits;
    const int s_begin = split * per, s_end = min(nsteps, s_begin + per);

    const T* __restrict__ DY = reinterpret_cast<const T*>(p.dy);
    const T* __restrict__ A = reinterpret_cast<const T*>(p.a);
    const int R = mp.rows_h * mp.rows_w;
    const float rcpR = 1.0f / (float)R, rcpW = 1.0f / (float)mp.rows_w;  // row decode without integer division (M < 2^24)

    uint4 ry_[NCY], rx_[NCX];
    auto load_step = [&](int s) {
        const int mbase = s * MS;
#pragma unroll
        for (int i = 0; i < NCY; ++i) {
            const int id = tid + i * 256;
            const int row = id / CY, ch = id - row * CY;
            const int m = mbase + row;
            const int n = n0 + ch * EPC;
            ry_[i] = make_uint4(0, 0, 0, 0);
            if (m < p.M && n < p.N) {
                int img, rem, ry, rx;
                fast_divmod(m, R, rcpR, img, rem);
                fast_divmod(rem, mp.rows_w, rcpW, ry, rx);
                const int64_t o = (int64_t)img * mp.out_batch_stride + mp.out_offset +
                                  (int64_t)((ry * mp.out_sy + mp.out_y0) * mp.out_w + rx * mp.out_sx + mp.out_x0) * p.ldo + n;
                ry_[i] = *reinterpret_cast<const uint4*>(DY + o);
            }
        }
#pragma unroll
        for (int i = 0; i < NCX; ++i) {
            const int id = tid + i * 256;
            const int row = id / CX, ch = id - row * CX;
            const int m = mbase + row;
            rx_[i] = make_uint4(0, 0, 0, 0);
            if (m < p.M) {
                int img, rem, ry, rx;
                fast_divmod(m, R, rcpR, img, rem);
                fast_divmod(rem, mp.rows_w, rcpW, ry, rx);
                const int iy = ry * mp.in_sy + dy, ix = rx * mp.in_sx + dx;
                if (iy >= 0 && iy < mp.in_h && ix >= 0 && ix < mp.in_w) {
                    const int64_t o = (int64_t)img * mp.in_batch_stride + mp.in_offset + (int64_t)(iy * mp.in_w + ix) * mp.in_c + c0 + ch * EPC;
                    rx_[i] = *reinterpret_cast<const uint4*>(A + o);
                }
            }
        }
    };
    auto store_step = [&](int stage) {
        char* sy = smem + stage * STAGE;
        char* sx = sy + MS * PY;
#pragma unroll
        for (int i = 0; i < NCY; ++i) {
            const int id = tid + i * 256;
            const int row = id / CY, ch = id - row * CY;
            *reinterpret_cast<uint4*>(sy + row * PY + ch * 16) = ry_[i];
        }
#pragma unroll
        for (int i = 0; i < NCX; ++i) {
            const int id = tid + i * 256;
            const int row = id / CX, ch = id - row * CX;
            *reinterpret_cast<uint4*>(sx + row * PX + ch * 16) = rx_[i];
        }
    };

    // D[i = c][j = n]: MFMA A operand = activation (rows c), B operand = dY (cols n)
    f32x4_v acc[FC][FNn];
#pragma unroll
    for (int i = 0; i < FC; ++i)
#pragma unroll
        for (int j = 0; j < FNn; ++j) acc[i][j] = (f32x4_v){0.f, 0.f, 0.f, 0.f};

    if (s_begin < s_end) {
        load_step(s_begin);
        store_step(0);
    }
    __syncthreads();
    const int q = lane & 15, g = lane >> 4;
    for (int s = s_begin; s < s_end; ++s) {
        const int cur = (s - s_begin) & 1;
        if (s + 1 < s_end) load_step(s + 1);
        const char* sy = smem + cur * STAGE;
        const char* sx = sy + MS * PY;
        if constexpr (sizeof(T) == 2) {
            // k-slot (g, j<4) <-> row g*4 + j ; (g, j>=4) <-> row 16 + g*4 + (j-4): a half-wave's 8 rows are contiguous
            // ds_read_b64_tr_b16: lanes 4r..4r+3 of a 16-lane group supply row r (4 x 8 B), lane i receives column i.
            const int tcol = (q & 3) * 4;
#pragma unroll
            for (int ks = 0; ks < MS / 32; ++ks) {
                const int trow = ks * 32 + g * 4 + (q >> 2);
                uint4 fx[FC], fy[FNn];
#pragma unroll
                for (int i = 0; i < FC; ++i) {
                    const char* b0 = sx + trow * PX + (wcc * WCC + i * 16 + tcol) * 2;
                    s16x4_v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_v*)(b0));
                    s16x4_v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_v*)(b0 + 16 * PX));
                    uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                    fx[i] = make_uint4(l2.x, l2.y, h2.x, h2.y);
                }
#pragma unroll
                for (int j = 0; j < FNn; ++j) {
                    const char* b0 = sy + trow * PY + (wnn * WNN + j * 16 + tcol) * 2;
                    s16x4_v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_v*)(b0));
                    s16x4_v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_v*)(b0 + 16 * PY));
                    uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                    fy[j] = make_uint4(l2.x, l2.y, h2.x, h2.y);
                }
#pragma unroll
                for (int i = 0; i < FC; ++i)
#pragma unroll
                    for (int j = 0; j < FNn; ++j) Mma<bf16_t>::run(acc[i][j], fx[i], fy[j]);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < MS / 4; ++ks) {
                float fx[FC], fy[FNn];
                const int row = ks * 4 + g;
#pragma unroll
                for (int i = 0; i < FC; ++i) fx[i] = *reinterpret_cast<const float*>(sx + row * PX + (wcc * WCC + i * 16 + q) * 4);
#pragma unroll
                for (int j = 0; j < FNn; ++j) fy[j] = *reinterpret_cast<const float*>(sy + row * PY + (wnn * WNN + j * 16 + q) * 4);
#pragma unroll
                for (int i = 0; i < FC; ++i)
#pragma unroll
                    for (int j = 0; j < FNn; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fx[i], fy[j], acc[i][j], 0, 0, 0);
            }
        }
        if (s + 1 < s_end) store_step(cur ^ 1);
        __syncthreads();
    }

    // lane holds n = q (+16 j), c = g*4 + r (+16 i): 4 consecutive k of one slab row -> float4 store
    const int64_t krow = (int64_t)p.kslots * mp.in_c;
    float* slab = p.slabs + (int64_t)split * p.N * krow;
#pragma unroll
    for (int i = 0; i < FC; ++i)
#pragma unroll
        for (int j = 0; j < FNn; ++j) {
            const int n = n0 + wnn * WNN + j * 16 + q;
            const int c = c0 + wcc * WCC + i * 16 + g * 4;
            if (n < p.N) {
                float* dst = slab + (int64_t)n * krow + (int64_t)mp.wslot[tap] * mp.in_c + c;
                *reinterpret_cast<float4*>(dst) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            }
        }
}

template <typename T, int BNN, int BC>
static int launch_wgrad(const theia_wgrad_args_t* a, hipStream_t stream) {
    constexpr int MS = WgTile<T>::MS;
    constexpr int PY = BNN * (int)sizeof(T) + WgTile<T>::PAD, PX = BC * (int)sizeof(T) + WgTile<T>::PAD;
    constexpr int lds = 2 * MS * (PY + PX);
    auto kern = gemm_wgrad_kernel<T, BNN, BC>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    const int tiles = cdiv_i(a->N, BNN) * a->map.ntaps * (a->map.in_c / BC);
    hipLaunchKernelGGL(kern, dim3(tiles * a->splits), dim3(256), lds, stream, *a);
    THEIA_CHECK_LAUNCH("theia_gemm_wgrad");
    return THEIA_OK;
}

int theia_gemm_wgrad_pp_launch(const theia_wgrad_args_t* a, hipStream_t stream);  // gemm_wgrad_pp.hip
bool theia_gemm_wgrad_pp_supported(const theia_wgrad_args_t* a);
int theia_gemm_wgrad_pp_mode(const theia_wgrad_args_t* a);
int theia_wgrad_pp_tiles_shape(int N, int in_c);  // output tiles per tap: 256 x 256, or 128 (n) x 384 (c) where that is less work
bool theia_wgrad_pp_shape_ok(int M, int N, int in_c);

static bool wgrad_use_pp() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("THEIA_WGRAD_KERNEL");
        v = (e != nullptr && strcmp(e, "std") == 0) ? 0 : 1;
    }
    return v == 1;
}

extern "C" int theia_wgrad_splits_taps(int M, int N, int kslots, int in_c) {
    const int Ktot = kslots * in_c;
    if (wgrad_use_pp() && theia_wgrad_pp_shape_ok(M, N, in_c)) {
        // ping-pong kernel: 256x256 output tiles (per tap; the last c tile of a tap may be partial), one workgroup per CU -> fill one
        // round of the CU budget as exactly as possible
        const int tiles = theia_wgrad_pp_tiles_shape(N, in_c) * kslots;
        // THEIA_WGRAD_CUS: workgroups (= CUs) a weight-gradient launch may fill -- fewer splits, each longer, and the rest of the chip stays
        // free for whatever the main stream runs beside it (A/B switch of the round-5 overlap experiments; default: the whole budget)
        static int wg_cus = -1;
        if (wg_cus < 0) {
            const char* e = getenv("THEIA_WGRAD_CUS");
            wg_cus = e != nullptr && atoi(e) > 0 ? atoi(e) : 0;
        }
        int budget = theia_compute_cus();
        if (wg_cus > 0 && wg_cus < budget) budget = wg_cus;
        int s = budget / (tiles > 0 ? tiles : 1);
        const int smax = cdiv_i(M, 32) / 8;
        if (s > smax) s = smax;
        const int cap = N < 128 ? 256 : 64;  // (the skinny, read-bound shapes: one or two tiles, a split per CU is still thousands of rows)
        if (s > cap) s = cap;
        return s < 1 ? 1 : s;
    }
    const int tiles = cdiv_i(N, 128) * cdiv_i(Ktot, 128);
    const int nsteps = cdiv_i(M, 64);
    int s = cdiv_i(768, tiles > 0 ? tiles : 1);
    const int smax = nsteps / 8 > 1 ? nsteps / 8 : 1;
    if (s > smax) s = smax;
    if (s > 64) s = 64;
    if (s < 1) s = 1;
    return s;
}
extern "C" int theia_wgrad_splits(int M, int N, int Ktot) { return theia_wgrad_splits_taps(M, N, 1, Ktot); }

// output tiles of one tap of a weight gradient with N output columns and in_c input channels on the ping-pong kernel (0: it does not take it)
extern "C" int theia_wgrad_tiles(int N, int in_c) {
    return wgrad_use_pp() && in_c % 64 == 0 && in_c >= 128 && N >= 128 ? theia_wgrad_pp_tiles_shape(N, in_c) : 0;
}
// M-splits of a GROUPED launch (theia_gemm_wgrad_group): `tiles` = sum over its problems of theia_wgrad_tiles(N, in_c), all with M rows
extern "C" int theia_wgrad_group_splits(int M, int tiles) {
    int s = theia_compute_cus() / (tiles > 0 ? tiles : 1);
    const int smax = cdiv_i(M, 32) / 8;
    if (s > smax) s = smax;
    if (s > 64) s = 64;
    return s < 1 ? 1 : s;
}

extern "C" int theia_gemm_wgrad_plan(const theia_wgrad_args_t* a, int dtype) {
    if (a == nullptr) return THEIA_ERR_INVALID;
    if (dtype == THEIA_BF16 && wgrad_use_pp() && theia_gemm_wgrad_pp_supported(a)) return 100 + theia_gemm_wgrad_pp_mode(a);
    return 0;
}

extern "C" int theia_wgrad_fuses_bias(const theia_wgrad_args_t* a, int dtype) {
    return a != nullptr && dtype == THEIA_BF16 && wgrad_use_pp() && theia_gemm_wgrad_pp_supported(a) ? 1 : 0;
}

static int wgrad_check_args(const theia_wgrad_args_t* a, int dtype) {
    THEIA_CHECK_ARG(a != nullptr, "theia_gemm_wgrad: null args");
    THEIA_CHECK_ARG(a->bias_out == nullptr || (a->bias_slabs != nullptr && theia_wgrad_fuses_bias(a, dtype)),
                    "theia_gemm_wgrad: bias_out needs bias_slabs and a kernel that fuses it (theia_wgrad_fuses_bias)");
    THEIA_CHECK_ARG(dtype == THEIA_F32 || dtype == THEIA_BF16, "theia_gemm_wgrad: bad dtype %d", dtype);
    THEIA_CHECK_ARG(a->dy && a->a && a->slabs, "theia_gemm_wgrad: null pointer");
    THEIA_CHECK_ARG(a->M > 0 && a->M < (1 << 24) && a->N > 0 && a->splits >= 1, "theia_gemm_wgrad: bad shape (M must be < 2^24)");
    THEIA_CHECK_ARG(a->N % 8 == 0, "theia_gemm_wgrad: N=%d must be a multiple of 8", a->N);
    THEIA_CHECK_ARG(a->map.in_c % 64 == 0, "theia_gemm_wgrad: in_c=%d must be a multiple of 64", a->map.in_c);
    THEIA_CHECK_ARG(a->ldo % 8 == 0 && a->map.out_offset % 8 == 0 && a->map.out_batch_stride % 8 == 0 &&
                        a->map.in_offset % 8 == 0 && a->map.in_batch_stride % 8 == 0,
                    "theia_gemm_wgrad: pitches/offsets must be multiples of 8 elements");
    for (int t = 0; t < a->map.ntaps; ++t)
        THEIA_CHECK_ARG(a->map.wslot[t] >= 0 && a->map.wslot[t] < a->kslots, "theia_gemm_wgrad: wslot out of range");
    return check_rowmap(a->map, 64, "theia_gemm_wgrad");
}

// n plain-matrix weight gradients (nn.Linear: one tap, one row per "image") in ONE launch of the ping-pong kernel -- see
// gemm_wgrad_pp.hip "Grouped launch".  Every problem carries its own splits / slabs / bias_slabs exactly as for theia_gemm_wgrad and is
// finished the same way (theia_wgrad_finish per problem).  THEIA_ERR_UNSUPPORTED (nothing launched): not bf16, the ping-pong kernel is
// switched off, n > 4, or a problem that is not a plain matrix -- the caller launches them one by one.
int theia_gemm_wgrad_pp_group_launch(const theia_wgrad_args_t* a, int n, hipStream_t stream);
extern "C" int theia_gemm_wgrad_group(const theia_wgrad_args_t* probs, int n, int dtype, void* stream) {
    THEIA_CHECK_ARG(probs != nullptr && n >= 1, "theia_gemm_wgrad_group: no problems");
    for (int k = 0; k < n; ++k) {
        const int rc = wgrad_check_args(&probs[k], dtype);
        if (rc) return rc;
    }
    if (dtype != THEIA_BF16 || !wgrad_use_pp()) return THEIA_ERR_UNSUPPORTED;
    for (int k = 0; k < n; ++k)
        if (!theia_gemm_wgrad_pp_supported(&probs[k])) return THEIA_ERR_UNSUPPORTED;
    return theia_gemm_wgrad_pp_group_launch(probs, n, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int theia_gemm_wgrad(const theia_wgrad_args_t* a, int dtype, void* stream) {
    int rc = wgrad_check_args(a, dtype);
    if (rc) return rc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == THEIA_BF16 && wgrad_use_pp()) {
        rc = theia_gemm_wgrad_pp_launch(a, s);
        if (rc != THEIA_ERR_UNSUPPORTED) return rc;
    }
    const bool n128 = a->N % 128 == 0, c128 = a->map.in_c % 128 == 0;
    if (dtype == THEIA_BF16) {
        if (n128 && c128) return launch_wgrad<bf16_t, 128, 128>(a, s);
        if (n128) return launch_wgrad<bf16_t, 128, 64>(a, s);
        if (c128) return launch_wgrad<bf16_t, 64, 128>(a, s);
        return launch_wgrad<bf16_t, 64, 64>(a, s);
    }
    if (n128 && c128) return launch_wgrad<float, 128, 128>(a, s);
    if (n128) return launch_wgrad<float, 128, 64>(a, s);
    if (c128) return launch_wgrad<float, 64, 128>(a, s);
    return launch_wgrad<float, 64, 64>(a, s);
}

// ------------------------------------------------------------------------------------------------
// slab reduction + permutation into the reference parameter layout
// ------------------------------------------------------------------------------------------------
__global__ void wgrad_reduce_kernel(const float* __restrict__ slabs, int splits, int N, int kslots, int C,
                                    float* __restrict__ out, int64_t sn, int64_t ss, int64_t sc, int accumulate) {
    const int64_t krow = (int64_t)kslots * C;
    const int64_t total = (int64_t)N * krow;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < splits; ++k) s += slabs[(int64_t)k * total + i];
        const int n = (int)(i / krow);
        const int r = (int)(i - (int64_t)n * krow);
        const int slot = r / C, ci = r - slot * C;
        const int64_t o = n * sn + slot * ss + ci * sc;
        out[o] = accumulate ? out[o] + s : s;
    }
}

extern "C" int theia_wgrad_reduce(const float* slabs, int splits, int N, int kslots, int C, float* out, int64_t sn,
                                  int64_t ss, int64_t sc, int accumulate, void* stream) {
    THEIA_CHECK_ARG(slabs && out && splits >= 1 && N > 0 && kslots > 0 && C > 0, "theia_wgrad_reduce: bad args");
    const int64_t total = (int64_t)N * kslots * C;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), slabs,
                       splits, N, kslots, C, out, sn, ss, sc, accumulate);
    THEIA_CHECK_LAUNCH("theia_wgrad_reduce");
    return THEIA_OK;
}

// ------------------------------------------------------------------------------------------------
// Fused finish of a weight-gradient GEMM: slab reduction + permutation into the reference parameter layout through an LDS tile
// (so that BOTH the slab reads and the parameter-gradient writes are coalesced whatever the permutation), and the bias
// partials of the same GEMM in the same launch.
//   weight tiles: TN rows n x all kslots x TC channels c.  Read order (slot, c fastest) = slab order; write order = the output's
//   own: its fastest dimension is the slot (stride 1) for the convolution layouts [co][ci][3][3] / [ci][co][3][3], followed by c
//   (stride 9: Conv2d, and ConvTranspose2d reduced over input pixels) or by n (stride 9: ConvTranspose2d stride 1), and c itself
//   for nn.Linear.  The plain wgrad_reduce_kernel wrote conv gradients with a 36-byte stride: 83 us for 85 MB (1 TB/s).
// ------------------------------------------------------------------------------------------------
// Sums over the split slabs, 8 slabs per pass requested together (a clamped slab index; slabs past the end add 0): as a load -> add loop a
// thread paid one L2 latency per slab -- with the 64 splits of DeiT-tiny's one-tile weight gradients the finish kernels took 17-48 us for
// a few MB.  The additions happen in the same order as before (k ascending).
__device__ __forceinline__ float4 wf_sum4(const float* __restrict__ p, int64_t stride, int k0, int splits, float4 s) {
    for (int k = k0; k < splits; k += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(p + (int64_t)(k + u < splits ? k + u : splits - 1) * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool in = k + u < splits;
            s.x += in ? v[u].x : 0.f;
            s.y += in ? v[u].y : 0.f;
            s.z += in ? v[u].z : 0.f;
            s.w += in ? v[u].w : 0.f;
        }
    }
    return s;
}
__device__ __forceinline__ float wf_sum1(const float* __restrict__ p, int64_t stride, int k0, int splits, float s) {
    for (int k = k0; k < splits; k += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(k + u < splits ? k + u : splits - 1) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += k + u < splits ? v[u] : 0.f;
    }
    return s;
}

__global__ __launch_bounds__(256) void wgrad_finish_kernel(const float* __restrict__ slabs, int splits, int N, int kslots, int C,
                                                           float* __restrict__ out, int64_t sn, int64_t ss, int64_t sc, int accumulate,
                                                           int TN, int TC, int wtiles, const float* __restrict__ bias_part,
                                                           float* __restrict__ bias_out, int bias_accumulate) {
    extern __shared__ float lds[];
    if ((int)blockIdx.x >= wtiles) {  // bias blocks: out[n] (+)= sum_s part[s*N + n], fixed order
        const int n = ((int)blockIdx.x - wtiles) * 256 + threadIdx.x;
        if (n < N) {
            const float s = wf_sum1(bias_part + n, N, 0, splits, 0.f);
            bias_out[n] = bias_accumulate ? bias_out[n] + s : s;
        }
        return;
    }
    // TN, TC are powers of two (host); kslots <= 16: float-reciprocal division (operands far below 2^24)
    const int LC = 31 - __builtin_clz(TC), LN = 31 - __builtin_clz(TN);
    const float rcp_k = 1.0f / (float)kslots;
    const int tiles_c = (C + TC - 1) >> LC;
    const int n0 = ((int)blockIdx.x / tiles_c) * TN, c0 = ((int)blockIdx.x % tiles_c) * TC;
    const int64_t krow = (int64_t)kslots * C, total = (int64_t)N * krow;
    const int pitch = TC + 1;
    const int per = TN * kslots * TC;
    const bool vec = (C & 3) == 0 && (reinterpret_cast<uint64_t>(slabs) & 15) == 0;  // c0, TC are multiples of 4
    if (vec) {  // 4 consecutive channels per lane: 16-byte slab reads
        for (int i = threadIdx.x * 4; i < per; i += 1024) {
            const int cc = i & (TC - 1), r = i >> LC;
            int slot;
            const int nn = gt_divmod(r, kslots, rcp_k, slot);
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n0 + nn < N && c0 + cc < C) {
                const int64_t idx = (int64_t)(n0 + nn) * krow + (int64_t)slot * C + c0 + cc;
                s = wf_sum4(slabs + idx, total, 0, splits, s);
            }
            float* d = lds + (nn * kslots + slot) * pitch + cc;
            d[0] = s.x; d[1] = s.y; d[2] = s.z; d[3] = s.w;
        }
    } else {
        for (int i = threadIdx.x; i < per; i += 256) {
            const int cc = i & (TC - 1), r = i >> LC;
            int slot;
            const int nn = gt_divmod(r, kslots, rcp_k, slot);
            float s = 0.f;
            if (n0 + nn < N && c0 + cc < C) {
                const int64_t idx = (int64_t)(n0 + nn) * krow + (int64_t)slot * C + c0 + cc;
                s = wf_sum1(slabs + idx, total, 0, splits, s);
            }
            lds[(nn * kslots + slot) * pitch + cc] = s;
        }
    }
    __syncthreads();
    // output order: slot fastest when ss == 1 (kslots > 1), then the smaller of (sc, sn)
    const bool c_mid = sc <= sn;  // [n][c][slot] (Conv2d) vs [c][n][slot] (ConvTranspose2d stride 1); Linear: kslots = 1, c_mid
    for (int i = threadIdx.x; i < per; i += 256) {
        int slot;
        const int r = gt_divmod(i, kslots, rcp_k, slot);
        int nn, cc;
        if (c_mid) { cc = r & (TC - 1); nn = r >> LC; } else { nn = r & (TN - 1); cc = r >> LN; }
        if (n0 + nn < N && c0 + cc < C) {
            const int64_t o = (int64_t)(n0 + nn) * sn + (int64_t)slot * ss + (int64_t)(c0 + cc) * sc;
            const float v = lds[(nn * kslots + slot) * pitch + cc];
            out[o] = accumulate ? out[o] + v : v;
        }
    }
}

// nn.Linear gradients (one slot, channels contiguous in the output: slab order IS output order): a plain 16-byte reduction, no LDS.
// 48 of the 69 finish launches of a DeiT-base step; the tiled kernel above spent them on 2-element-per-thread blocks.
__device__ __forceinline__ void wf_rows_body(int bid, const float* __restrict__ slabs, int splits, int N, int C, float* __restrict__ out, int64_t sn,
                                             int accumulate, int wblocks, const float* __restrict__ bias_part, float* __restrict__ bias_out,
                                             int bias_accumulate) {
    if (bid >= wblocks) {
        const int n = (bid - wblocks) * 256 + threadIdx.x;
        if (n < N) {
            const float s = wf_sum1(bias_part + n, N, 0, splits, 0.f);
            bias_out[n] = bias_accumulate ? bias_out[n] + s : s;
        }
        return;
    }
    const int c4n = C >> 2;
    const int64_t total4 = (int64_t)N * c4n, total = (int64_t)N * C;
    const float rcp = 1.0f / (float)c4n;
    for (int64_t i = (int64_t)bid * 256 + threadIdx.x; i < total4; i += (int64_t)wblocks * 256) {
        float4 s = *reinterpret_cast<const float4*>(slabs + i * 4);
        s = wf_sum4(slabs + i * 4, total, 1, splits, s);
        int c4;
        const int n = gt_divmod((int)i, c4n, rcp, c4);
        float4* d = reinterpret_cast<float4*>(out + (int64_t)n * sn + c4 * 4);
        if (accumulate) {
            const float4 o = *d;
            s.x = o.x + s.x; s.y = o.y + s.y; s.z = o.z + s.z; s.w = o.w + s.w;
        }
        *d = s;
    }
}

__global__ __launch_bounds__(256) void wgrad_finish_rows_kernel(const float* __restrict__ slabs, int splits, int N, int C, float* __restrict__ out,
                                                                int64_t sn, int accumulate, int wblocks, const float* __restrict__ bias_part,
                                                                float* __restrict__ bias_out, int bias_accumulate) {
    wf_rows_body((int)blockIdx.x, slabs, splits, N, C, out, sn, accumulate, wblocks, bias_part, bias_out, bias_accumulate);
}

// the reductions behind ONE grouped weight-gradient launch (theia_gemm_wgrad_group) as one launch: block ranges, one per problem
struct wf_group_t {
    int njobs, splits;
    int first[THEIA_WGRAD_FINISH_GROUP_MAX + 1];   // first[j] .. first[j + 1]: job j's blocks (its weight blocks, then its bias blocks)
    int wblocks[THEIA_WGRAD_FINISH_GROUP_MAX];
    theia_wgrad_finish_job_t job[THEIA_WGRAD_FINISH_GROUP_MAX];
};
__global__ __launch_bounds__(256) void wgrad_finish_rows_group_kernel(const wf_group_t g) {
    const int bid = (int)blockIdx.x;
    int j = 0;
#pragma unroll
    for (int q = 1; q < THEIA_WGRAD_FINISH_GROUP_MAX; ++q)
        if (q < g.njobs && bid >= g.first[q]) j = q;
    const theia_wgrad_finish_job_t& t = g.job[j];
    wf_rows_body(bid - g.first[j], t.slabs, g.splits, t.N, t.C, t.out, t.sn, t.accumulate, g.wblocks[j], t.bias_slabs, t.bias_out, t.bias_accumulate);
}

static bool wf_rows_ok(const float* slabs, const float* out, int N, int C, int64_t sn) {
    return (C & 3) == 0 && (sn & 3) == 0 && (int64_t)N * (C >> 2) < ((int64_t)1 << 24) &&
           ((reinterpret_cast<uint64_t>(slabs) | reinterpret_cast<uint64_t>(out)) & 15) == 0;
}
static int wf_rows_blocks(int N, int C) {
    int wblocks = (int)(((int64_t)N * (C >> 2) + 255) / 256);
    return wblocks > 4096 ? 4096 : wblocks;
}

extern "C" int theia_wgrad_finish_group(const theia_wgrad_finish_job_t* jobs, int n, int splits, void* stream) {
    THEIA_CHECK_ARG(jobs && n >= 1 && splits >= 1, "theia_wgrad_finish_group: bad args");
    if (n > THEIA_WGRAD_FINISH_GROUP_MAX) return THEIA_ERR_UNSUPPORTED;
    wf_group_t g;
    g.njobs = n;
    g.splits = splits;
    g.first[0] = 0;
    for (int j = 0; j < n; ++j) {
        const theia_wgrad_finish_job_t& t = jobs[j];
        THEIA_CHECK_ARG(t.slabs && t.out && t.N > 0 && t.C > 0, "theia_wgrad_finish_group: bad job");
        THEIA_CHECK_ARG((t.bias_out == nullptr) == (t.bias_slabs == nullptr), "theia_wgrad_finish_group: bias_slabs and bias_out go together");
        if (!wf_rows_ok(t.slabs, t.out, t.N, t.C, t.sn)) return THEIA_ERR_UNSUPPORTED;
        g.job[j] = t;
        g.wblocks[j] = wf_rows_blocks(t.N, t.C);
        g.first[j + 1] = g.first[j] + g.wblocks[j] + (t.bias_out != nullptr ? cdiv_i(t.N, 256) : 0);
    }
    for (int j = n; j < THEIA_WGRAD_FINISH_GROUP_MAX; ++j) {
        g.job[j] = g.job[0];
        g.wblocks[j] = 0;
        g.first[j + 1] = g.first[n];
    }
    hipLaunchKernelGGL(wgrad_finish_rows_group_kernel, dim3(g.first[n]), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g);
    THEIA_CHECK_LAUNCH("theia_wgrad_finish_group");
    return THEIA_OK;
}

extern "C" int theia_wgrad_finish(const float* slabs, int splits, int N, int kslots, int C, float* out, int64_t sn, int64_t ss,
                                  int64_t sc, int accumulate, const float* bias_slabs, float* bias_out, int bias_accumulate,
                                  void* stream) {
    THEIA_CHECK_ARG(slabs && out && splits >= 1 && N > 0 && kslots > 0 && kslots <= 16 && C > 0, "theia_wgrad_finish: bad args");
    THEIA_CHECK_ARG((bias_out == nullptr) == (bias_slabs == nullptr), "theia_wgrad_finish: bias_slabs and bias_out go together");
    const int btiles = bias_out != nullptr ? cdiv_i(N, 256) : 0;
    hipStream_t hs = reinterpret_cast<hipStream_t>(stream);
    if (kslots == 1 && sc == 1 && wf_rows_ok(slabs, out, N, C, sn)) {
        const int wblocks = wf_rows_blocks(N, C);
        hipLaunchKernelGGL(wgrad_finish_rows_kernel, dim3(wblocks + btiles), dim3(256), 0, hs, slabs, splits, N, C, out, sn, accumulate, wblocks,
                           bias_slabs, bias_out, bias_accumulate);
        THEIA_CHECK_LAUNCH("theia_wgrad_finish");
        return THEIA_OK;
    }
    // tile: many channels when c follows the slot in the output (long contiguous runs per n), square-ish when n does
    const bool c_mid = sc <= sn;
    const int TC = c_mid ? 64 : 32, TN = c_mid ? (kslots > 1 ? 4 : 8) : 32;
    const int wtiles = cdiv_i(N, TN) * cdiv_i(C, TC);
    const size_t lds = (size_t)TN * kslots * (TC + 1) * sizeof(float);
    hipLaunchKernelGGL(wgrad_finish_kernel, dim3(wtiles + btiles), dim3(256), lds, hs, slabs, splits, N,
                       kslots, C, out, sn, ss, sc, accumulate, TN, TC, wtiles, bias_slabs, bias_out, bias_accumulate);
    THEIA_CHECK_LAUNCH("theia_wgrad_finish");
    return THEIA_OK;
}

