// Ping-pong implicit GEMM for 3x3 stride-1 (transposed) convolutions whose 256-row M tile IS one 16x16 output image:
// Conv2d 3x3 p1 on 16x16 maps, its data-gradient, and the 14->16 "pad" ConvTranspose2d of the translator heads
// (adapter_heads.py:279-290,316-324).  Same output, epilogue and wave layout as gemm_nt_pp_kernel (gemm_pp.hip), different
// operand stream.
//
// The generic kernel gathers the activation operand tap by tap: per 32-channel slice it streams 9 x 16 KB of activations
// (the same <= 256 input pixels, shifted) plus 9 x 16 KB of weights from L2 into LDS, and that L2 -> LDS stream (~10 TB/s
// over the chip), not the matrix pipe, bounds it.  Here a slice of the input image is staged ONCE
// (in_h*in_w pixel rows of 64 B) and the 9 taps read their MFMA fragments at SHIFTED pixel rows: lane (x = lane & 15) of
// output row y reads pixel (y + dy, x + dx), or a row of zeros when that falls outside the image.  Per slice the stream is
// 16 KB + 9 x 16 KB instead of 18 x 16 KB, and the fragments of one dx serve its three dy taps (30 + 36 ds_read_b128 per
// slice and wave instead of 108).  Each image buffer carries 32 zero rows above and below the pixels (rows outside the image
// in y); lanes whose pixel column is outside the image read the zero pad instead (address select, once per kernel).
//
// Unit u = (slice s, tap r = 3*dxi + dyi), 32 MFMAs per wave -- the role a half k-tile plays in gemm_pp.hip; the two wave
// groups run one barrier apart and alternate R(u) (fragment reads + LDS-DMA issue) and M(u) (MFMAs).
//   B ring: NB = 6 weight tiles [256 n][64 B]; R(u) issues the tile of unit u + 5 into slot (u + 5) % 6 = (u - 1) % 6, whose
//           last readers were the R(u - 1) segments of both groups (closed by lgkmcnt(0) before their barrier).
//   A ring: 2 image slices; slice s + 1 is issued in R(9 s + 2) into the buffer slice s - 1 was last read from in unit 9 s - 3.
//   waits : end of R(u): counted s_waitcnt vmcnt(8), or (10) in the five units whose queue holds an image slice's two pieces
//           behind weight tile u + 1 -> the two pieces of weight tile u + 1 have landed (tiles u + 2 .. u + 5 stay in flight),
//           then the barrier makes every wave's pieces visible.  Image slice s + 1 sits in the queue right behind weight
//           tile 9 s + 7; the vmcnt(8) at the end of R(9 s + 7) retires it, two units before its first read in R(9 s + 9).
#include <type_traits>
#include "gemm_tile.h"

__device__ uint4 g_cv_zero_page[16];  // 256 B of zeros: source of out-of-range rows (never advanced)

struct conv_taps_t {
    int32_t dy0, dx0;       // smallest tap offsets; taps cover (dy0 + dyi, dx0 + dxi), dyi, dxi in 0..2
    int32_t wslot[9];       // weight slot of tap r = 3*dxi + dyi
};

__device__ __forceinline__ int cv_f(int row) { return (4 - ((row >> 2) & 3)) & 3; }

// BN: output columns per tile, 256 or 128.  The 128-wide instantiation exists for the launcher's column split (see
// theia_gemm_conv_pp_launch): 128 images x 768 columns are 384 tiles of 256 = 1.5 rounds of the 256 CUs, i.e. two rounds of time; as
// 256 tiles of 256 columns + 256 tiles of 128 columns they are one full round + one round of half tiles.
template <typename T, bool SUMS = false, int BN = 256>
__global__ __launch_bounds__(512) void gemm_conv_pp_kernel(const theia_gemm_args_t p, const conv_taps_t tp) {
    constexpr int BM = 256, WAVES_N = 4;
    constexpr int NB = 6, PD = NB - 1;
    constexpr int HKT = 64 / (int)sizeof(T);    // channels per slice (64-byte rows)
    constexpr int EPC = 16 / (int)sizeof(T);
    constexpr int WM = 128, WN = BN / WAVES_N, FM = 8, FN = WN / 16;
    constexpr int SRP = 128;                     // rows staged per piece (512 threads x 16 B)
    constexpr int NPB = BN / SRP;                // LDS-DMA pieces per thread per weight tile
    static_assert(BN == 256 || BN == 128, "weight tiles of one or two 128-row staging passes");
    constexpr int BTILE = BN * 64;
    constexpr int APAD = 32 * 64;                // 32 zero rows above and below the image: pixel rows y < 0 / y >= in_h read zeros
    constexpr int ATILE = BM * 64 + 2 * APAD;    // one image-slice buffer: [pad | 256 pixel rows | pad]
    constexpr int A_OFF = NB * BTILE;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int ugroup = uwave >> 2;
    const theia_rowmap_t& mp = p.map;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tile = gt_xcd_remap(blockIdx.x, gridDim.x);
    const int img = tile / tiles_n;
    const int m0 = img * BM, n0 = (tile % tiles_n) * BN;
    const T* __restrict__ A = reinterpret_cast<const T*>(p.a);
    const T* __restrict__ W = reinterpret_cast<const T*>(p.w);
    const uint64_t zp = reinterpret_cast<uint64_t>(g_cv_zero_page);

    if constexpr (SUMS) {  // the statistics table behind the ring and the epilogue's staging areas (see the end of the kernel)
        constexpr int ring_bytes_ = NB * BTILE + 2 * ATILE, ep_bytes_ = 8 * 64 * (WN + 4) * 4;
        if (tid < 2 * GT_SUMS_SLOTS) reinterpret_cast<unsigned long long*>(smem + (ring_bytes_ > ep_bytes_ ? ring_bytes_ : ep_bytes_))[tid] = 0ull;
    }
    // zero pads of the two image buffers (4 x 2 KiB = 512 x 16 B): never touched by the LDS-DMA, read by fragments whose pixel
    // row is above / below the image and by the lanes whose pixel column is outside it
    {
        const int pad = tid >> 7, o = (tid & 127) * 16;
        *reinterpret_cast<uint4*>(smem + A_OFF + (pad >> 1) * ATILE + (pad & 1) * (APAD + BM * 64) + o) = make_uint4(0, 0, 0, 0);
    }

    // ---- per-thread LDS-DMA sources: rows st_row and st_row + 128 of an image slice / of a weight tile ----
    // Swizzle of the 16-byte chunks inside a 64-byte row (bank-conflict-free ds_read_b128): weight rows by their row index,
    // image rows by the pixel's COLUMN x -- so that a fragment's address is (lane part) + (pixel row) * in_w * 64 for any in_w.
    const int st_chunk = tid & 3, st_row = tid >> 2;
    const int lchunk = st_chunk ^ cv_f(st_row);
    const int npix = mp.in_h * mp.in_w;
    const float rcp_w = 1.0f / (float)mp.in_w;
    uint64_t a_src[2], w_src[NPB];
    uint32_t w_live[NPB];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int pix = st_row + SRP * i;
        int px;
        (void)gt_divmod(pix, mp.in_w, rcp_w, px);
        const int achunk = st_chunk ^ cv_f(px);
        const uint64_t pa = reinterpret_cast<uint64_t>(A + (int64_t)img * mp.in_batch_stride + mp.in_offset + (int64_t)pix * mp.in_c + achunk * EPC);
        a_src[i] = pix < npix ? pa : 0;  // 0: the zero page, not advanced with the slice
    }
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
        const int n = n0 + st_row + SRP * i;
        w_live[i] = n < p.N ? 0xffffffffu : 0u;
        w_src[i] = n < p.N ? reinterpret_cast<uint64_t>(W + (int64_t)n * p.ldw + lchunk * EPC) : zp;
    }
    auto issue_b = [&](int slot, int r, int s) {  // weight tile of tap r, slice s
        const uint32_t off = (uint32_t)((tp.wslot[r] * mp.in_c + s * HKT) * (int)sizeof(T));
        char* dst = smem + slot * BTILE + uwave * (16 * 64);
#pragma unroll
        for (int i = 0; i < NPB; ++i) {
            const uint64_t src = w_src[i] + (off & w_live[i]);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + i * (SRP * 64)), 16, 0, 0);
        }
    };
    auto issue_a = [&](int s, int buf) {  // image slice s into image buffer buf
        const uint64_t off = (uint64_t)s * 64;
        char* dst = smem + A_OFF + buf * ATILE + APAD + uwave * (16 * 64);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint64_t src = a_src[i] != 0 ? a_src[i] + off : zp;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + i * (SRP * 64)), 16, 0, 0);
        }
    };

    gt_f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = (gt_f32x4){0.f, 0.f, 0.f, 0.f};

    const int nslice = mp.in_c / HKT;
    const int nunits = nslice * 9;
    // ---- prologue: image slice 0, weight tiles 0 .. PD-1 (clamped for very short K: duplicates are never read) ----
    issue_a(0, 0);
#pragma unroll
    for (int u = 0; u < PD; ++u) {
        const int uc = min(u, nunits - 1);
        issue_b(u, uc % 9, uc / 9);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPB * (PD - 1)) : "memory");  // image slice 0 and weight tile 0 landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    // the zero pads are written
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (ugroup == 1) {  // group 1 runs one barrier behind group 0
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }

    // ---- fragment addressing: lane (x = lane & 15, k-chunk fg) of output row y reads pixel (y + dy, x + dx) ----
    // All LDS reads of the main loop are inline asm (gt_ds_read128: C++ loads would make the compiler drain the LDS-DMA queue
    // at the top of every unit).  Weight fragments: one lane offset + immediates.  Image fragments of tap column dxi: address =
    // a_lane[dxi] + (pixel row) * a_step[dxi]; lanes whose pixel column is outside the image point at the zero pad with step 0.
    const int frow = lane & 15, fg = lane >> 4;
    const uint32_t smem_base = gt_lds_addr(smem);
    const uint32_t lane_b = smem_base + (wn * WN + frow) * 64 + ((fg ^ cv_f(frow)) << 4);
    const int y0 = wm * 8 + tp.dy0;  // first pixel row of this wave's 10-row window (8 output rows + 2 halo rows)
    uint32_t a_lane[3], a_step[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int ix = frow + tp.dx0 + d;
        const bool xok = (unsigned)ix < (unsigned)mp.in_w;
        a_lane[d] = smem_base + A_OFF + (xok ? APAD + (y0 * mp.in_w + ix) * 64 + ((fg ^ cv_f(ix)) << 4) : fg * 16);
        a_step[d] = xok ? mp.in_w * 64 : 0;
    }
    int bslot = 0;  // ring slot of the current unit's weight tile
    gt_u32x4 fb[FN], fa[FM];
    for (int s = 0; s < nslice; ++s) {
        const uint32_t abuf = (s & 1) * ATILE;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const int dxi = r / 3, dyi = r % 3;
            // ---------------- R(u): fragment reads + LDS-DMA issue
            {
                const uint32_t ab = lane_b + bslot * BTILE;
                gt_ds_read128<0>(fb[0], ab);
                gt_ds_read128<1024>(fb[1], ab);
                if constexpr (FN > 2) {
                    gt_ds_read128<2048>(fb[2 % FN], ab);
                    gt_ds_read128<3072>(fb[3 % FN], ab);
                }
            }
            {
                // The wave's 8 output rows of tap (dyi, dxi) read pixel rows y0 + dyi .. y0 + dyi + 7, shifted by dx.  Row q of
                // the 10-row window lives in fa[q % 8]: dyi = 0 loads rows 0..7, dyi = 1 replaces row 0 by row 8, dyi = 2 row 1
                // by row 9 (10 reads per dx, 8 fragment registers).
                const int q_lo = dyi == 0 ? 0 : FM + dyi - 1, q_hi = dyi == 0 ? FM : FM + dyi;
                const uint32_t aa = a_lane[dxi] + abuf;
#pragma unroll
                for (int q = q_lo; q < q_hi; ++q) gt_ds_read128<0>(fa[q % FM], aa + q * a_step[dxi]);
            }
            {
                // prefetch: weight tile of unit u + PD (clamped at the tail), image slice s + 1 once per slice
                int rp = r + PD, sp = s;
                if (rp >= 9) { rp -= 9; ++sp; }
                if (sp >= nslice) { sp = nslice - 1; rp = 8; }
                const int pslot = bslot == 0 ? NB - 1 : bslot - 1;  // (u + PD) % NB
                issue_b(pslot, rp, sp);
                // every slice issues exactly one image slice (the last one re-fetches itself into the idle buffer, never
                // read), so the number of LDS-DMA operations younger than a given weight tile is a compile-time constant
                if (r == 2) issue_a(min(s + 1, nslice - 1), (s + 1) & 1);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < FN; ++i) asm volatile("" : "+v"(fb[i]));
#pragma unroll
            for (int j = 0; j < FM; ++j) asm volatile("" : "+v"(fa[j]));
            // weight tile u + 1 landed: younger than it are tiles u + 2 .. u + 5 (NPB operations each) and, in the five units after
            // an image slice was queued (behind tile u_A + 5), that slice's two pieces
            if (r >= 2 && r <= 6) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPB * (PD - 1) + 2) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPB * (PD - 1)) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---------------- M(u): 32 MFMAs
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < FM; ++j)
#pragma unroll
                for (int i = 0; i < FN; ++i) gt_mma<T>(acc[i][j], fb[i], fa[(j + dyi) % FM]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            bslot = bslot == NB - 1 ? 0 : bslot + 1;
        }
    }
    if (ugroup == 0) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): drain the clamped tail prefetches before LDS is reused
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    float* ep = reinterpret_cast<float*>(smem) + wave * (64 * (WN + 4));
    const bool prefetch = sizeof(T) == 2 && (p.resid != nullptr || p.act == THEIA_ACT_MUL_DGELU || p.act == THEIA_ACT_MUL_DRELU);
    if constexpr (SUMS) {
        // the tile is one image: the eight waves' (sum, sum of squares) meet in LDS and one wave adds them to global memory -- two
        // device-scope atomics per tile instead of sixteen (they serialise per address at the memory side: gemm_epi_direct.h)
        constexpr int ring_bytes = NB * BTILE + 2 * ATILE, ep_bytes = 8 * 64 * (WN + 4) * 4;
        unsigned long long* const sums_tab = reinterpret_cast<unsigned long long*>(smem + (ring_bytes > ep_bytes ? ring_bytes : ep_bytes));
        gt_epilogue<T, WM, WN, true, false, 4, 0>(acc, ep, p, m0 + wm * WM, n0 + wn * WN, lane, sums_tab, img);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (uwave == 0) gt_flush_sums(sums_tab, reinterpret_cast<unsigned long long*>(p.ln_sums), img, BM, p.M, lane);
    }
    else if (prefetch) gt_epilogue<T, WM, WN, false, false, 4, 1>(acc, ep, p, m0 + wm * WM, n0 + wn * WN, lane);
    else gt_epilogue<T, WM, WN, false, false, 4, 0>(acc, ep, p, m0 + wm * WM, n0 + wn * WN, lane);
}

// Does the row map describe a convolution this kernel takes?  (one 16x16 output image per 256-row tile, stride 1, taps a
// full 3x3 grid of consecutive offsets, the input image within 256 pixels)  Fills the tap grid when it does.
bool theia_gemm_conv_pp_match(const theia_gemm_args_t* a, int dtype, conv_taps_t* out) {
    const theia_rowmap_t& m = a->map;
    const int hkt = dtype == THEIA_BF16 ? 32 : 16;
    if (m.ntaps != 9 || m.rows_h != 16 || m.rows_w != 16 || m.in_sy != 1 || m.in_sx != 1) return false;
    if (a->M % 256 != 0 || m.in_c % hkt != 0 || a->K != 9 * m.in_c) return false;
    if (m.in_h < 1 || m.in_w < 1 || m.in_h * m.in_w > 256) return false;
    int dy0 = m.dy[0], dx0 = m.dx[0];
    for (int t = 1; t < 9; ++t) {
        dy0 = m.dy[t] < dy0 ? m.dy[t] : dy0;
        dx0 = m.dx[t] < dx0 ? m.dx[t] : dx0;
    }
    conv_taps_t tp;
    tp.dy0 = dy0;
    tp.dx0 = dx0;
    for (int r = 0; r < 9; ++r) tp.wslot[r] = -1;
    for (int t = 0; t < 9; ++t) {
        const int dyi = m.dy[t] - dy0, dxi = m.dx[t] - dx0;
        if (dyi > 2 || dxi > 2 || tp.wslot[3 * dxi + dyi] != -1 || m.wslot[t] < 0) return false;
        tp.wslot[3 * dxi + dyi] = m.wslot[t];
    }
    if (out != nullptr) *out = tp;
    return true;
}

template <typename T, bool SUMS, int BN>
static void conv_pp_launch_one(const theia_gemm_args_t& a, const conv_taps_t& tp, hipStream_t stream) {
    constexpr int ring_bytes = 6 * BN * 64 + 2 * (256 * 64 + 2 * 32 * 64);
    constexpr int ep_bytes = 8 * 64 * (BN / 4 + 4) * 4;
    constexpr int lds = (ring_bytes > ep_bytes ? ring_bytes : ep_bytes) + 64;  // + the workgroup's statistics table (SUMS)
    auto kern = gemm_conv_pp_kernel<T, SUMS, BN>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    const int tiles = (a.M / 256) * cdiv_i(a.N, BN);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), lds, stream, a, tp);
}

template <int BN>
static void conv_pp_launch_bn(const theia_gemm_args_t& a, const conv_taps_t& tp, int dtype, hipStream_t stream) {
    const bool sums = a.ln_sums != nullptr;
    if (dtype == THEIA_BF16 && sums) conv_pp_launch_one<bf16_t, true, BN>(a, tp, stream);
    else if (dtype == THEIA_BF16) conv_pp_launch_one<bf16_t, false, BN>(a, tp, stream);
    else if (sums) conv_pp_launch_one<float, true, BN>(a, tp, stream);
    else conv_pp_launch_one<float, false, BN>(a, tp, stream);
}

int theia_gemm_conv_pp_launch(const theia_gemm_args_t* a, int dtype, hipStream_t stream) {
    conv_taps_t tp;
    if (!theia_gemm_conv_pp_match(a, dtype, &tp)) {
        theia_set_error("theia_gemm_nt: the row map is not a 3x3 stride-1 convolution with one 16x16 image per 256-row tile");
        return THEIA_ERR_UNSUPPORTED;
    }
    // Column split: when the grid of 256-column tiles ends in a round that is at most half full, the last 256 columns run as a second
    // launch of 128-column tiles -- (tiles_n - 1) * images full tiles + 2 * images half tiles.  A half tile costs ~0.6 of a full one
    // (its R segments carry the same image-fragment reads for half the MFMAs), so this pays when it removes a whole round:
    // 128 images x 768 columns: 2 rounds -> 1 + 0.6.  THEIA_CONV_SPLIT=0: A/B switch.
    static int split_ok = -1;
    if (split_ok < 0) {
        const char* e = getenv("THEIA_CONV_SPLIT");
        split_ok = (e != nullptr && strcmp(e, "0") == 0) ? 0 : 1;
    }
    const int cus = theia_compute_cus();
    const int images = a->M / 256, tn = cdiv_i(a->N, 256);
    const long full = (long)images * tn;
    const double cost_plain = (double)cdiv_i(full, cus);
    const double cost_split = (double)cdiv_i((long)images * (tn - 1), cus) + 0.6 * (double)cdiv_i((long)images * 2, cus);
    // N = 256 k + 128 (DeiT-small's 384 channels): the last 128 columns ARE one 128-column tile per image -- always split (the plain grid
    // would run them as half-empty 256-column tiles)
    const bool tail128 = tn >= 2 && a->N % 256 == 128;
    if (split_ok && tn >= 2 && (tail128 || (a->N % 256 == 0 && cost_split < cost_plain - 0.05))) {
        theia_gemm_args_t lo = *a, hi = *a;
        const int ncut = (tn - 1) * 256;
        const int esz = dtype == THEIA_BF16 ? 2 : 4;
        lo.N = ncut;
        hi.N = a->N - ncut;
        hi.w = static_cast<const char*>(a->w) + (size_t)ncut * a->ldw * esz;
        hi.out = static_cast<char*>(a->out) + (size_t)ncut * esz;
        if (a->bias != nullptr) hi.bias = a->bias + ncut;
        if (a->resid != nullptr) hi.resid = static_cast<const char*>(a->resid) + (size_t)ncut * esz;
        if (a->aux_in != nullptr) hi.aux_in = static_cast<const char*>(a->aux_in) + (size_t)ncut * esz;
        if (a->aux_out != nullptr) hi.aux_out = static_cast<char*>(a->aux_out) + (size_t)ncut * esz;
        conv_pp_launch_bn<256>(lo, tp, dtype, stream);
        THEIA_CHECK_LAUNCH("theia_gemm_nt(conv)");
        conv_pp_launch_bn<128>(hi, tp, dtype, stream);
        THEIA_CHECK_LAUNCH("theia_gemm_nt(conv, 128-column tiles)");
        return THEIA_OK;
    }
    conv_pp_launch_bn<256>(*a, tp, dtype, stream);
    THEIA_CHECK_LAUNCH("theia_gemm_nt(conv)");
    return THEIA_OK;
}
