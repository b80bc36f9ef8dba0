// Ping-pong weight-gradient GEMM (bf16):  slab[s][n][wslot*in_c + c] = sum_{m in split s} dY[m, n] * A[m, (tap, c)]
//
// Same two-wave-group schedule, 4-deep LDS-DMA ring and counted waits as gemm_pp.hip (read that header first); what differs:
//   * the contraction runs over ROWS m, so one ring slot holds 32 rows of dY (256 n-columns) and 32 rows of the gathered
//     activation (256 c-columns of one tap): [32][512 B] + [32][512 B];
//   * both MFMA operands are therefore m-major in LDS and their fragments are fetched with the transposing read
//     ds_read_b64_tr_b16 (two per fragment: rows 4g..4g+3 and 16+4g..16+4g+3 -- the k-slot permutation both operands share);
//   * bank conflicts: a half-wave's transposing read touches 8 rows x 32 B of one column block, so the 32-byte block index is
//     XORed with (row & 7) inside each 256-byte segment; LDS-DMA writes lane-linearly, hence the swizzle is applied to the
//     source column of each lane (each lane owns a fixed column for the whole kernel);
//   * each thread stages 2 rows per 32-row step (both operands); the (image, y, x) decode of a row is kept incrementally
//     (m advances by 32: float-reciprocal wrap, no integer division) inside the M segments;
//   * rows past the end of the split (or of M) read a page of zeros, so the tail needs no special case.
// Output tile 256 (c) x 256 (n) per workgroup; 8 waves, each 128 (c) x 64 (n); accumulators go straight to the f32 slab.
#include "gemm_tile.h"

typedef __attribute__((ext_vector_type(4))) short wp_s16x4;

__device__ uint4 g_wp_zero_page[16];
template <int N> __device__ __forceinline__ void wp_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ uint4 wp_tr_frag(const char* __restrict__ part, int lb, int lane) {
    const int q = lane & 15, g = lane >> 4;
    const int row = g * 4 + (q >> 2);
    const int pb = (lb & 8) | ((lb & 7) ^ (row & 7));
    const char* p = part + row * 512 + pb * 32 + (q & 3) * 8;
    const wp_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wp_s16x4*)(p));
    const wp_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wp_s16x4*)(p + 16 * 512));
    const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
    return make_uint4(l2.x, l2.y, h2.x, h2.y);
}

// The same for the 128 (n) x 384 (c) tile (NB = 8, round 6): a ring slot is four sub-parts [32 rows][256 B] -- 128 dY columns, then the three
// 128-column thirds of the activation tile -- with the 32-byte block index XORed with (row & 7) inside each 256-byte row.
__device__ __forceinline__ uint4 wp_tr_frag_v(const char* __restrict__ sub, int lb, int lane) {
    const int q = lane & 15, g = lane >> 4;
    const int row = g * 4 + (q >> 2);
    const char* p = sub + row * 256 + ((lb ^ (row & 7)) * 32) + (q & 3) * 8;
    const wp_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wp_s16x4*)(p));
    const wp_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wp_s16x4*)(p + 16 * 256));
    const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
    return make_uint4(l2.x, l2.y, h2.x, h2.y);
}

// FAST: constant-step row addressing (see below); the host picks the instantiation, so the hot loop carries one path only
// SPLIT_ISSUE: where the LDS-DMA of half-step h+NSTAGE-1 is issued (A/B switch THEIA_WGRAD_ISSUE=m: both rows in the M segment)
// NSTAGE: ring depth, 4 (128 KB of LDS) or 5 (all 160 KB: one more half-step of prefetch distance; THEIA_WGRAD_STAGES)
// MODE (of the FAST instantiation; THEIA_WGRAD_MODES=0: A/B switch, everything on mode 0):
//   0  stepping: per row (y, x) state, one address constant per step + one per pixel wrap + one per image wrap (any map that qualifies
//      for FAST)
//   1  plain row-major matrices (one row per "image", one tap at (0, 0): the nn.Linear launches) -- a staged row is valid iff it lies
//      inside the split and advances by one constant; no (y, x) state, no wrap selects
//   2  periodic: images whose pixel count is a multiple of 32 with at most 32 steps per image and whose width divides 32 (16x16 maps: 8
//      steps of two image rows).  A step then never wraps in x, wraps in y exactly when the step index inside the image returns to 0
//      -- for every row of the workgroup at once, so the wrap constant is a scalar select -- and whether the tap's input pixel exists
//      repeats with that period: each staged row carries a bit mask built once and the loop tests one bit of it
// Measured (profiles/r03_ab_wgrad_row_modes.txt): the ~35 VALU instructions per staged row of mode 0 sit in the R / M segments of every
// half-step; mode 1 took 19 % off the ViT weight-gradient launches.
// NB: 32-byte column blocks of dY per tile.  16 = the 256 (n) x 256 (c) tile above.  8 (round 6) = a 128 (n) x 384 (c) tile for channel
//   counts that are multiples of 128 but not of 256 -- DeiT-small: N = in_c = 384 is 2 x 2 tiles of 256 at 56 % use, but 3 x 1 of these at
//   100 %; eight waves of 96 (c) x 64 (n), 24 MFMAs per wave and half-step instead of 32 for the same 32 KB of operands.  Staging: ONE row per
//   thread (row 4 * wave + lane / 16), its dY piece and its three activation pieces (sub-parts of [32][256 B]: wp_tr_frag_v) -- half the
//   row state of the 256 x 256 form, the same four LDS-DMA operations per thread and half-step, two in each segment.
// hardware places block b on XCD b % 8.  The blocks [first, first + n) of a launch -> 0 .. n-1 such that each XCD's blocks get a contiguous
// range (gt_xcd_remap for a range that does not start at a multiple of 8: a problem inside a grouped launch)
__device__ __forceinline__ int wp_xcd_remap_range(int b, int first, int n) {
    const int x = b & 7;
    int start = 0;
#pragma unroll
    for (int y = 0; y < 8; ++y) {
        const int fy = first + ((y - first) & 7);                       // first block >= `first` on XCD y
        const int cy = fy < first + n ? (first + n - 1 - fy) / 8 + 1 : 0;  // blocks of the range on XCD y
        start += y < x ? cy : 0;
    }
    const int fx = first + ((x - first) & 7);
    return start + (b - fx) / 8;
}

// The body of one workgroup: block `hw_bid` (its blockIdx.x: hw_bid % 8 is the XCD it runs on) of the blocks [first, first + nblocks) that
// problem p owns in the launch (first = 0: a launch of its own).
template <bool FAST, bool SPLIT_ISSUE, int NSTAGE, int MODE, int NB = 16, bool BOTH_R = false>
__device__ __forceinline__ void wgrad_pp_body(const theia_wgrad_args_t& p, const int plain_order, const int hw_bid, const int nblocks, const int first = 0) {
    constexpr bool wgrad_split_issue = SPLIT_ISSUE;
    // BOTH_R (round 6): both rows of half-step h+AHEAD issued in the R segment, none between the MFMAs (what the NT kernel does).  Pays for
    // the stepping row mode only, whose ~35 VALU per staged row between the MFMAs held the matrix pipe up: the stride-2 launches 1273 ->
    // 1164 us isolated, the step -0.1 ms; the periodic mode got slower (304 -> 338 us: its R segment then outlasts the other group's M),
    // plain matrices the same, and the 128 x 384 tile (24 MFMAs per M segment) prefers the split too.
    constexpr bool both_r = BOTH_R && SPLIT_ISSUE;
    constexpr bool V = NB == 8;  // the 128 x 384 tile
    static_assert(NB == 16 || (NB == 8 && FAST && SPLIT_ISSUE), "the 128 x 384 tile: stepping instantiations with the split issue only");
    constexpr int TN = V ? 128 : 256, TC = V ? 384 : 256, WC = V ? 96 : 128;  // tile columns (n, c), c columns per wave
    constexpr int MS = 32, ROWB = 512, PART = MS * ROWB, STAGE = 2 * PART, SUB = MS * 256;
    constexpr int AHEAD = NSTAGE - 1;  // half-steps in flight ahead of the one being multiplied; 4 LDS-DMA operations per thread each
    constexpr int FM = V ? 6 : 8, FN = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    const int ugroup = uwave >> 2;
    const int wm = V ? wave >> 1 : wave >> 2, wn = V ? wave & 1 : wave & 3;   // wm: which WC c-columns, wn: which 64 n-columns
    const theia_rowmap_t& mp = p.map;
    const int tiles_n = (p.N + TN - 1) / TN;
    const int tiles_c = (mp.in_c + TC - 1) / TC;  // (the last c tile may be partial: in_c = 384, 192 -- columns past in_c read zeros and are not stored)
    const int ntile = tiles_n * mp.ntaps * tiles_c;
    // hardware places block b on XCD b % 8: give each XCD a contiguous range of (split, tile) so that the workgroups sharing a dY
    // column tile or an activation (tap, c) tile meet in one L2 (THEIA_WGRAD_XCD=0: A/B switch, plain order)
    // Order inside a split: tap slowest, then c tile, then n tile (plain_order & 2; rounds 2-6).  Round 6 tried the c tile slowest
    // (THEIA_WGRAD_XCD=ctile: the ~30 consecutive tiles of an XCD are then the taps and n tiles of ONE c tile, whose slice of the gathered
    // operand every tap reads a shifted copy of): the L2-side fetch of the 3x3 launches fell 18 % (414 -> 340 MB per launch), of the stride-2
    // launches 6 % (2.78 -> 2.62 GB: the ~30 workgroups of an XCD drift apart by more than the 4 MB L2 holds, the re-reads are served by
    // the memory-side cache either way) -- and both got 3 % SLOWER (303 vs 295 us, 1283 vs 1251 us, one box): with the tap slowest the
    // workgroups of an XCD share their dense operand rows, which are the larger stream.  Not the default.
    const int bid = (plain_order & 1) ? hw_bid - first : first == 0 ? gt_xcd_remap(hw_bid, nblocks) : wp_xcd_remap_range(hw_bid, first, nblocks);
    const int tile = bid % ntile, split = bid / ntile;
    const int tn = tile % tiles_n, tk = tile / tiles_n;
    const int tap = (plain_order & 2) ? tk / tiles_c : tk % mp.ntaps;
    const int c0 = ((plain_order & 2) ? tk - tap * tiles_c : tk / mp.ntaps) * TC, n0 = tn * TN;
    const int dy = mp.dy[tap], dx = mp.dx[tap];

    const int nsteps = (p.M + MS - 1) / MS;
    const int per = (nsteps + p.splits - 1) / p.splits;
    const int s_begin = split * per;
    const int s_end = min(nsteps, s_begin + per);
    const int nh = max(0, s_end - s_begin);
    const int m_limit = min(p.M, s_end * MS);

    const bf16_t* __restrict__ DY = reinterpret_cast<const bf16_t*>(p.dy);
    const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(p.a);
    const uint64_t zp = reinterpret_cast<uint64_t>(g_wp_zero_page);

    // ---- this thread's staging column (fixed) and its two rows' decode state
    // (128 x 384 tile: ONE row per thread, 4 * wave + lane / 16, and a 16-byte piece of each of the four 128-column sub-parts)
    const int srow = V ? uwave * 4 + (lane >> 4) : tid >> 5, s16 = V ? lane & 15 : tid & 31;
    const int pb = s16 >> 1;
    const int lb = V ? pb ^ (srow & 7) : (pb & 8) | ((pb & 7) ^ (srow & 7));
    const int col = (lb * 2 + (s16 & 1)) * 8;            // element column inside the 256-wide tile (the 128-wide sub-part)
    const bool n_ok = n0 + col < p.N;
    const bool c_ok = c0 + col < mp.in_c;
    const bool c_ok1 = c0 + 128 + col < mp.in_c, c_ok2 = c0 + 256 + col < mp.in_c;  // (V: the second and third activation sub-parts)
    const float rcpW = 1.0f / (float)mp.rows_w, rcpH = 1.0f / (float)mp.rows_h;
    int st_m[2], st_img[2], st_ry[2], st_rx[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = s_begin * MS + srow + (V ? 0 : 16 * i);
        const int R = mp.rows_h * mp.rows_w;
        st_m[i] = m;
        st_img[i] = m / R;
        const int rem = m - st_img[i] * R;
        st_ry[i] = rem / mp.rows_w;
        st_rx[i] = rem - st_ry[i] * mp.rows_w;
    }
    // Incremental row stepping (FAST instantiation, chosen by the host): a staged row advances by 32 GEMM rows per half-step,
    // i.e. by (32 / rows_w) image rows and (32 % rows_w) pixels, with at most one pixel wrap and one image wrap per step
    // (32 / rows_w + 1 <= rows_h), or by 32 / R whole images for plain matrices (R = 1).  Both source addresses therefore
    // move by a constant plus one constant per wrap -- ~25 VALU instructions per staged row instead of ~70 for the general
    // decode below (PMC, round 1: 5.8 VALU instructions per MFMA in this kernel, issued between the MFMAs of the M segment;
    // 3.6x the NT kernel's).
    const int R_img = mp.rows_h * mp.rows_w;
    const bool whole_images = (MS % R_img) == 0;
    constexpr bool fast = FAST;
    const int q32 = whole_images ? 0 : MS / mp.rows_w, r32 = whole_images ? 0 : MS % mp.rows_w;  // image rows / pixels per step
    // element steps of the dY (output side) and activation (input side) addresses: per step, per pixel wrap, per image wrap
    const int64_t oy_row = (int64_t)mp.out_sy * mp.out_w * p.ldo, oy_pix = (int64_t)mp.out_sx * p.ldo;
    const int64_t ox_row = (int64_t)mp.in_sy * mp.in_w * mp.in_c, ox_pix = (int64_t)mp.in_sx * mp.in_c;
    const int64_t oy_step = whole_images ? (int64_t)(MS / R_img) * mp.out_batch_stride : q32 * oy_row + r32 * oy_pix;
    const int64_t ox_step = whole_images ? (int64_t)(MS / R_img) * mp.in_batch_stride : q32 * ox_row + r32 * ox_pix;
    const int64_t oy_wx = oy_row - mp.rows_w * oy_pix, ox_wx = ox_row - mp.rows_w * ox_pix;                       // rx wrapped: one row down
    const int64_t oy_wy = mp.out_batch_stride - mp.rows_h * oy_row, ox_wy = mp.in_batch_stride - mp.rows_h * ox_row;  // ry wrapped: next image
    // byte steps as 32-bit values (the host takes this instantiation only when they fit), selected per lane with one v_cndmask each
    const int32_t by_step = (int32_t)(oy_step * 2), by_wx = (int32_t)(oy_wx * 2), by_wy = (int32_t)(oy_wy * 2);
    const int32_t bx_step = (int32_t)(ox_step * 2), bx_wx = (int32_t)(ox_wx * 2), bx_wy = (int32_t)(ox_wy * 2);
    // the tap is fixed per workgroup: "input pixel inside the image" is a range test on the row's (y, x) -- no multiplication
    const int ry_lo = dy >= 0 ? 0 : (-dy + mp.in_sy - 1) / mp.in_sy, ry_hi = dy >= mp.in_h ? 0 : (mp.in_h - dy + mp.in_sy - 1) / mp.in_sy;
    const int rx_lo = dx >= 0 ? 0 : (-dx + mp.in_sx - 1) / mp.in_sx, rx_hi = dx >= mp.in_w ? 0 : (mp.in_w - dx + mp.in_sx - 1) / mp.in_sx;
    const unsigned ry_span = ry_hi > ry_lo ? (unsigned)(ry_hi - ry_lo) : 0u, rx_span = rx_hi > rx_lo ? (unsigned)(rx_hi - rx_lo) : 0u;
    uint64_t f_py[2], f_px[2];   // addresses of the row's dY / activation piece (valid or not)
    uint32_t vmask[2] = {0u, 0u};  // MODE 2: bit k = the tap's input pixel exists for this row in step k of an image
    const int period = R_img / MS;
    int phase[2];                  // MODE 2: step inside the image of the row's NEXT issue (uniform)
    phase[0] = phase[1] = MODE == 2 ? s_begin % period : 0;
    if constexpr (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            for (int k = 0; k < period; ++k) {
                const int rem = srow + (V ? 0 : 16 * i) + MS * k;
                const int ry = rem / mp.rows_w, rx = rem - ry * mp.rows_w;
                const bool ok = ((unsigned)(ry - ry_lo) < ry_span) & ((unsigned)(rx - rx_lo) < rx_span);
                vmask[i] |= (ok ? 1u : 0u) << k;
            }
    }
    if constexpr (fast) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int64_t oy = (int64_t)st_img[i] * mp.out_batch_stride + mp.out_offset +
                               (int64_t)((st_ry[i] * mp.out_sy + mp.out_y0) * mp.out_w + st_rx[i] * mp.out_sx + mp.out_x0) * p.ldo + n0 + col;
            const int iy = st_ry[i] * mp.in_sy + dy, ix = st_rx[i] * mp.in_sx + dx;
            const int64_t ox = (int64_t)st_img[i] * mp.in_batch_stride + mp.in_offset + (int64_t)(iy * mp.in_w + ix) * mp.in_c + c0 + col;
            f_py[i] = reinterpret_cast<uint64_t>(DY + oy);
            f_px[i] = reinterpret_cast<uint64_t>(A + ox);
        }
    }
    // issue the dY piece and the activation piece of row i for the current half-step, then advance the row by 32
    auto issue_row = [&](int i, char* slot) {
        if constexpr (V) {
            // half 0 (R segment): the dY piece and the first activation piece; half 1 (M segment): the other two, then the row moves on
            const bool mok = st_m[0] < m_limit;
            bool xok = mok;
            if constexpr (MODE == 2) xok = mok & (((vmask[0] >> phase[0]) & 1u) != 0u);
            else if constexpr (MODE == 0) xok = mok & ((unsigned)(st_ry[0] - ry_lo) < ry_span) & ((unsigned)(st_rx[0] - rx_lo) < rx_span);
            char* dst = slot + uwave * (4 * 256);
            if (i == 0) {
                const uint64_t sy = (mok & n_ok) ? f_py[0] : zp;
                const uint64_t sx = (xok & c_ok) ? f_px[0] : zp;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sy, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sx, (__attribute__((address_space(3))) void*)(dst + SUB), 16, 0, 0);
                return;
            }
            const uint64_t s1 = (xok & c_ok1) ? f_px[0] + 256 : zp;
            const uint64_t s2 = (xok & c_ok2) ? f_px[0] + 512 : zp;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s1, (__attribute__((address_space(3))) void*)(dst + 2 * SUB), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s2, (__attribute__((address_space(3))) void*)(dst + 3 * SUB), 16, 0, 0);
            i = 0;  // ... and advance row 0 through the common code below
            st_m[0] += MS;
            if constexpr (MODE == 1) {
                f_py[0] += (uint64_t)(int64_t)by_step;
                f_px[0] += (uint64_t)(int64_t)bx_step;
                return;
            }
            if constexpr (MODE == 2) {
                const bool wrap = phase[0] + 1 == period;
                f_py[0] += (uint64_t)(int64_t)(wrap ? by_step + by_wy : by_step);
                f_px[0] += (uint64_t)(int64_t)(wrap ? bx_step + bx_wy : bx_step);
                phase[0] = wrap ? 0 : phase[0] + 1;
                return;
            }
            int rx = st_rx[0] + r32;
            const bool wx = rx >= mp.rows_w;
            rx = wx ? rx - mp.rows_w : rx;
            int ry = st_ry[0] + q32 + (wx ? 1 : 0);
            const bool wy = ry >= mp.rows_h;
            ry = wy ? ry - mp.rows_h : ry;
            st_rx[0] = rx;
            st_ry[0] = ry;
            f_py[0] += (uint64_t)(int64_t)(by_step + (wx ? by_wx : 0) + (wy ? by_wy : 0));
            f_px[0] += (uint64_t)(int64_t)(bx_step + (wx ? bx_wx : 0) + (wy ? bx_wy : 0));
            return;
        }
        const bool mok = st_m[i] < m_limit;
        uint64_t sy, sx;
        if constexpr (MODE == 1) {
            sy = (mok & n_ok) ? f_py[i] : zp;
            sx = (mok & c_ok) ? f_px[i] : zp;
        } else if constexpr (MODE == 2) {
            const bool xok = mok & c_ok & (((vmask[i] >> phase[i]) & 1u) != 0u);
            sy = (mok & n_ok) ? f_py[i] : zp;
            sx = xok ? f_px[i] : zp;
        } else if constexpr (fast) {
            const bool xok = mok & c_ok & ((unsigned)(st_ry[i] - ry_lo) < ry_span) & ((unsigned)(st_rx[i] - rx_lo) < rx_span);
            sy = (mok & n_ok) ? f_py[i] : zp;
            sx = xok ? f_px[i] : zp;
        } else {
            const int64_t oy = (int64_t)st_img[i] * mp.out_batch_stride + mp.out_offset +
                               (int64_t)((st_ry[i] * mp.out_sy + mp.out_y0) * mp.out_w + st_rx[i] * mp.out_sx + mp.out_x0) * p.ldo + n0 + col;
            const int iy = st_ry[i] * mp.in_sy + dy, ix = st_rx[i] * mp.in_sx + dx;
            const bool xok = mok & c_ok & (iy >= 0) & (iy < mp.in_h) & (ix >= 0) & (ix < mp.in_w);
            const int64_t ox = (int64_t)st_img[i] * mp.in_batch_stride + mp.in_offset + (int64_t)(iy * mp.in_w + ix) * mp.in_c + c0 + col;
            const uint64_t my = 0ull - (uint64_t)(mok & n_ok), mx = 0ull - (uint64_t)xok;
            sy = (reinterpret_cast<uint64_t>(DY + oy) & my) | (zp & ~my);
            sx = (reinterpret_cast<uint64_t>(A + ox) & mx) | (zp & ~mx);
        }
        char* dst = slot + (16 * i + uwave * 2) * ROWB;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sy, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sx, (__attribute__((address_space(3))) void*)(dst + PART), 16, 0, 0);
        st_m[i] += MS;
        if constexpr (MODE == 1) {
            f_py[i] += (uint64_t)(int64_t)by_step;
            f_px[i] += (uint64_t)(int64_t)bx_step;
            return;
        }
        if constexpr (MODE == 2) {  // phase is wave-uniform: the image wrap is a scalar select
            const bool wrap = phase[i] + 1 == period;
            f_py[i] += (uint64_t)(int64_t)(wrap ? by_step + by_wy : by_step);
            f_px[i] += (uint64_t)(int64_t)(wrap ? bx_step + bx_wy : bx_step);
            phase[i] = wrap ? 0 : phase[i] + 1;
            return;
        }
        if constexpr (fast) {  // constants per step; one more per wrap (never for whole_images: q32 = r32 = 0, R = 1 keeps rx = ry = 0)
            int rx = st_rx[i] + r32;
            const bool wx = rx >= mp.rows_w;
            rx = wx ? rx - mp.rows_w : rx;
            int ry = st_ry[i] + q32 + (wx ? 1 : 0);
            const bool wy = ry >= mp.rows_h;
            ry = wy ? ry - mp.rows_h : ry;
            st_rx[i] = rx;
            st_ry[i] = ry;
            f_py[i] += (uint64_t)(int64_t)(by_step + (wx ? by_wx : 0) + (wy ? by_wy : 0));
            f_px[i] += (uint64_t)(int64_t)(bx_step + (wx ? bx_wx : 0) + (wy ? bx_wy : 0));
            return;
        }
        // general advance: m += 32  ->  (rx, ry, img) with float-reciprocal wraps (operands < 2^12, one correction each)
        int rx = st_rx[i] + MS;
        int q = (int)((float)rx * rcpW);
        rx -= q * mp.rows_w;
        if (rx >= mp.rows_w) { rx -= mp.rows_w; ++q; }
        if (rx < 0) { rx += mp.rows_w; --q; }
        int ry = st_ry[i] + q;
        int q2 = (int)((float)ry * rcpH);
        ry -= q2 * mp.rows_h;
        if (ry >= mp.rows_h) { ry -= mp.rows_h; ++q2; }
        if (ry < 0) { ry += mp.rows_h; --q2; }
        st_rx[i] = rx;
        st_ry[i] = ry;
        st_img[i] += q2;
    };

    gt_f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = (gt_f32x4){0.f, 0.f, 0.f, 0.f};
    // fused bias gradient: the workgroups of the first (tap, c) tile column also sum dY over their rows -- the dY fragments
    // against a fragment of ones, 4 more MFMAs per half-step in the four waves that own the first 128 c columns
    const bool do_bias = p.bias_slabs != nullptr && tk == 0 && wm == 0;  // wave-uniform
    gt_f32x4 bacc[FN];
#pragma unroll
    for (int i = 0; i < FN; ++i) bacc[i] = (gt_f32x4){0.f, 0.f, 0.f, 0.f};
    const uint4 ones = make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u);

    // prologue: half-steps 0..AHEAD-1 (rows past the split end read zeros)
#pragma unroll
    for (int h = 0; h < AHEAD; ++h) {
        issue_row(0, smem + h * STAGE);
        issue_row(1, smem + h * STAGE);
    }
    wp_wait_vm<(AHEAD - 1) * 4>();  // half-step 0 landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (ugroup == 1) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }

    uint4 fb[FN];
    int cur = 0, nxt = AHEAD;  // ring slots of half-step h and of half-step h + AHEAD (NSTAGE need not be a power of two)
    for (int h = 0; h < nh; ++h) {
        char* nslot = smem + nxt * STAGE;
        const char* sy = smem + cur * STAGE;  // dY part; activation part follows
        const char* sx = sy + PART;
        cur = cur + 1 == NSTAGE ? 0 : cur + 1;
        nxt = nxt + 1 == NSTAGE ? 0 : nxt + 1;
        {
            // ---------------- R(h): 12 transposed fragments (24 ds_read_b64_tr_b16)
            uint4 fa[FM];
#pragma unroll
            for (int i = 0; i < FN; ++i) fb[i] = V ? wp_tr_frag_v(sy, wn * 4 + i, lane) : wp_tr_frag(sy, wn * 4 + i, lane);
#pragma unroll
            for (int j = 0; j < FM; ++j) {
                if constexpr (V) fa[j] = wp_tr_frag_v(sy + SUB * (1 + ((wm * 6 + j) >> 3)), (wm * 6 + j) & 7, lane);  // c block wm * 6 + j of 24
                else fa[j] = wp_tr_frag(sx, wm * 8 + j, lane);
            }
            // Row 0 of half-step h+AHEAD is issued here, row 1 between the MFMAs below: an issue_row is 2 LDS-DMA instructions (~100
            // issue cycles each) + ~40 VALU.  With both in the M segment it was ~1200 cycles against ~600 for R -- the matrix pipe 40 %
            // busy (PMC) -- since the other group's R segment cannot run longer than this group's M; one in each balances them.
            if (wgrad_split_issue) issue_row(0, nslot);
            if (both_r) issue_row(1, nslot);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // half-step h+1 landed; h+2 .. h+AHEAD-1 (4 operations each) and, when issued above, row 0 of h+AHEAD (2) may still be in flight
            if (both_r) wp_wait_vm<(AHEAD - 2) * 4 + 4>();
            else if (wgrad_split_issue) wp_wait_vm<(AHEAD - 2) * 4 + 2>();
            else wp_wait_vm<(AHEAD - 2) * 4>();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---------------- M(h): 32 MFMAs + row 1 (or both rows) of half-step h+AHEAD
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < FM; ++j) {
#pragma unroll
                for (int i = 0; i < FN; ++i) GtMma<bf16_t>::run(acc[i][j], fa[j], fb[i]);
                if (!wgrad_split_issue && j == 0) issue_row(0, nslot);
                if (!both_r && j == (wgrad_split_issue ? 2 : 4)) issue_row(1, nslot);
            }
            if (do_bias) {
#pragma unroll
                for (int i = 0; i < FN; ++i) GtMma<bf16_t>::run(bacc[i], ones, fb[i]);
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (ugroup == 0) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // lane holds n = nf*16 + (lane&15), c = kf*16 + (lane>>4)*4 .. +4  ->  one float4 per fragment
    const int64_t krow = (int64_t)p.kslots * mp.in_c;
    float* slab = p.slabs + (int64_t)split * p.N * krow;
    const int q = lane & 15, g = lane >> 4;
    if (do_bias && g == 0) {  // every c row of the ones-product holds the column sum: take row 0 (lanes 0-15, register 0)
#pragma unroll
        for (int i = 0; i < FN; ++i) {
            const int n = n0 + wn * 64 + i * 16 + q;
            if (n < p.N) p.bias_slabs[(int64_t)split * p.N + n] = bacc[i][0];
        }
    }
#pragma unroll
    for (int i = 0; i < FN; ++i) {
        const int n = n0 + wn * 64 + i * 16 + q;
        if (n >= p.N) continue;
        float* drow = slab + (int64_t)n * krow + (int64_t)mp.wslot[tap] * mp.in_c + c0 + wm * WC + g * 4;
        const int c_left = mp.in_c - (c0 + wm * WC + g * 4);  // columns of this lane's first float4 up to the end of the tap's c range
#pragma unroll
        for (int j = 0; j < FM; ++j)
            if (j * 16 < c_left) *reinterpret_cast<float4*>(drow + j * 16) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
}

template <bool FAST, bool SPLIT_ISSUE = true, int NSTAGE = 4, int MODE = 0, int NB = 16, bool BOTH_R = false>
__global__ __launch_bounds__(512) void gemm_wgrad_pp_kernel(const theia_wgrad_args_t p, const int plain_order) {
    wgrad_pp_body<FAST, SPLIT_ISSUE, NSTAGE, MODE, NB, BOTH_R>(p, plain_order, (int)blockIdx.x, (int)gridDim.x);
}

// Grouped launch (round 6): up to WGRAD_GROUP_MAX plain-matrix problems (mode 1: the nn.Linear weight gradients) in ONE grid.  Why: a
// launch wants one workgroup per CU, so the M-split count of a problem is CUs / tiles -- 28 for the 9 tiles of a [768, 768] gradient
// (o_proj), each split 900 rows long: 28 prologues, 28 f32 partials of every tile to write and to reduce again (66 MB for a 2.4 MB
// result), 713 TFLOP/s against 1.0-1.07 PFLOP/s for the 27 / 36-tile gradients.  o_proj's and the fused q/k/v gradient of a layer share M
// and are both at hand when the attention backward has run: together they are 36 tiles x 7 splits of 3602 rows -- the same shape of
// launch as fc1's and fc2's.  The problems' block ranges follow each other without gaps (wp_xcd_remap_range keeps each XCD's blocks of a
// problem on contiguous tiles wherever the range starts): 36 tiles x 7 splits are 252 workgroups, not 260.
constexpr int WGRAD_GROUP_MAX = 4;
struct wgrad_group_t {
    int nprob;
    int first[WGRAD_GROUP_MAX];   // first block of problem k
    int count[WGRAD_GROUP_MAX];   // its tiles x splits
    theia_wgrad_args_t prob[WGRAD_GROUP_MAX];
};
template <int NSTAGE, int NB = 16>
__global__ __launch_bounds__(512) void gemm_wgrad_pp_group_kernel(const wgrad_group_t g, const int plain_order) {
    int k = 0;
#pragma unroll
    for (int i = 1; i < WGRAD_GROUP_MAX; ++i)
        if (i < g.nprob && (int)blockIdx.x >= g.first[i]) k = i;
    wgrad_pp_body<true, true, NSTAGE, 1, NB>(g.prob[k], plain_order, (int)blockIdx.x, g.count[k], g.first[k]);
}

// out[n] (+)= sum_s part[s*N + n], fixed order
__global__ __launch_bounds__(256) void wgrad_bias_reduce_kernel(const float* __restrict__ part, int splits, int N, float* __restrict__ out,
                                                                int accumulate) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int q = 0; q < splits; ++q) s += part[(int64_t)q * N + n];
    out[n] = accumulate ? out[n] + s : s;
}

// in_c a multiple of 64 (the last of ceil(in_c / 256) c tiles may be partial: DeiT-small's 384, DeiT-tiny's 192), at least half a tile of n
// ... or a skinny output over very many rows (round 6: the Depth head's Linear(C, 32) over 64 x 64 maps, M = b * 4096): the launch is bound by
// reading the [M, in_c] activation once, which the row-streaming ring does at the copy rate with one workgroup per CU (the 2-stage kernel:
// 280 us for 805 MB); that most of the tile's n columns multiply zeros costs nothing there
bool theia_wgrad_pp_shape_ok(int M, int N, int in_c) { return in_c % 64 == 0 && in_c >= 128 && (N >= 128 || (N >= 32 && M >= 262144)); }
bool theia_gemm_wgrad_pp_supported(const theia_wgrad_args_t* a) { return theia_wgrad_pp_shape_ok(a->M, a->N, a->map.in_c); }

// Tile of a launch: 256 (n) x 256 (c), or 128 x 384 (returns 8: NB of the kernel) when that is less MFMA work -- tiles x 0.75 against tiles
// (N = in_c = 384: 3 x 0.75 against 4; N = 1152, in_c = 384: 9 x 0.75 against 10; multiples of 256 tie and stay).  THEIA_WGRAD_TILE=256 |
// 384: A/B switch (384 where the kernel can: stepping row modes, split issue, 4-deep ring).
int theia_gemm_wgrad_pp_mode(const theia_wgrad_args_t* a);
int theia_wgrad_pp_nb_shape(int N, int in_c) {
    static int force = -1;
    if (force < 0) {
        const char* e = getenv("THEIA_WGRAD_TILE");
        force = e == nullptr ? 0 : atoi(e);
        const char* e2 = getenv("THEIA_WGRAD_ISSUE");
        const char* e3 = getenv("THEIA_WGRAD_STAGES");
        if ((e2 != nullptr && strcmp(e2, "m") == 0) || (e3 != nullptr && atoi(e3) == 5)) force = 256;  // those A/B switches exist for the 256 x 256 tile only
    }
    if (force == 256) return 16;
    const long t256 = (long)cdiv_i(N, 256) * cdiv_i(in_c, 256) * 4, t384 = (long)cdiv_i(N, 128) * cdiv_i(in_c, 384) * 3;
    return force == 384 || t384 < t256 ? 8 : 16;
}
int theia_wgrad_pp_nb(const theia_wgrad_args_t* a) {
    const int mode = theia_gemm_wgrad_pp_mode(a);
    if (mode < 10) return 16;  // the per-row decode path (maps the stepping cannot take) has the 256 x 256 tile only
    return theia_wgrad_pp_nb_shape(a->N, a->map.in_c);
}
// output tiles of a launch per tap
int theia_wgrad_pp_tiles_shape(int N, int in_c) {
    return theia_wgrad_pp_nb_shape(N, in_c) == 8 ? cdiv_i(N, 128) * cdiv_i(in_c, 384) : cdiv_i(N, 256) * cdiv_i(in_c, 256);
}

// Which row addressing the launch will use (host logic only; exported through theia_gemm_wgrad_plan): 0 = per-row decode (the maps
// the stepping path cannot take), 10 = stepping (mode 0), 11 = plain matrices (mode 1), 12 = periodic (mode 2).  -1: not this kernel.
int theia_gemm_wgrad_pp_mode(const theia_wgrad_args_t* a) {
    if (!theia_gemm_wgrad_pp_supported(a)) return -1;
    static int allow_fast = -1, allow_modes = -1, issue_in_m = -1, stages = -1;
    if (allow_fast < 0) {
        const char* e = getenv("THEIA_WGRAD_STEP");      // =general: A/B switch for the row-stepping fast path
        allow_fast = (e != nullptr && strcmp(e, "general") == 0) ? 0 : 1;
        e = getenv("THEIA_WGRAD_MODES");                 // =0: everything on mode 0
        allow_modes = (e != nullptr && strcmp(e, "0") == 0) ? 0 : 1;
        e = getenv("THEIA_WGRAD_ISSUE");                 // =m: both LDS-DMA rows in the M segment
        issue_in_m = (e != nullptr && strcmp(e, "m") == 0) ? 1 : 0;
        e = getenv("THEIA_WGRAD_STAGES");                // =5: 5-deep ring
        stages = e != nullptr && atoi(e) == 5 ? 5 : 4;
    }
    const theia_rowmap_t& mp = a->map;
    const int R_img = mp.rows_h * mp.rows_w;
    // the stepping instantiation keeps its address steps (bytes) in 32 bits: one map row / one image of either operand below 1 GiB
    const int64_t step_lim = (int64_t)1 << 29;  // elements
    const bool steps_fit = (int64_t)mp.out_batch_stride * (R_img >= 32 ? 1 : 32 / (R_img > 0 ? R_img : 1)) < step_lim &&
                           (int64_t)mp.in_batch_stride * (R_img >= 32 ? 1 : 32 / (R_img > 0 ? R_img : 1)) < step_lim &&
                           (int64_t)mp.out_sy * mp.out_w * a->ldo * (mp.rows_h + 33) < step_lim &&
                           (int64_t)mp.in_sy * mp.in_w * mp.in_c * (mp.rows_h + 33) < step_lim;
    const bool fast = allow_fast && steps_fit && ((32 % R_img) == 0 || 32 / mp.rows_w + 1 <= mp.rows_h);
    if (!fast) return 0;
    const bool modes = allow_modes && !issue_in_m;
    if (modes && R_img == 1 && mp.ntaps == 1 && mp.dy[0] == 0 && mp.dx[0] == 0 && mp.in_h >= 1 && mp.in_w >= 1) return 11;
    // mode 2: no x wrap inside a step (the image width divides 32), whole steps per image, a period that fits the 32-bit mask
    // (R_img == 32 is the whole-images case of the stepping path: its step constant already is the image stride)
    if (modes && stages == 4 && R_img % 32 == 0 && R_img >= 64 && R_img / 32 <= 32 && 32 % mp.rows_w == 0) return 12;
    return 10;
}

static int wgrad_plain_order() {
    static int plain_order = -1;
    if (plain_order < 0) {
        const char* e = getenv("THEIA_WGRAD_XCD");
        plain_order = e == nullptr ? 2 : strcmp(e, "0") == 0 ? 3 : strcmp(e, "ctile") == 0 ? 0 : 2;
    }
    return plain_order;
}

// n <= WGRAD_GROUP_MAX plain-matrix problems (theia_gemm_wgrad_pp_mode == 11 each) in one launch; THEIA_ERR_UNSUPPORTED otherwise.
int theia_gemm_wgrad_pp_group_launch(const theia_wgrad_args_t* a, int n, hipStream_t stream) {
    if (n < 1 || n > WGRAD_GROUP_MAX) return THEIA_ERR_UNSUPPORTED;
    wgrad_group_t g;
    g.nprob = n;
    int next = 0;
    for (int k = 0; k < WGRAD_GROUP_MAX; ++k) {
        g.first[k] = next;
        g.count[k] = 0;
        if (k >= n) continue;
        if (theia_gemm_wgrad_pp_mode(&a[k]) != 11) return THEIA_ERR_UNSUPPORTED;
        if (theia_wgrad_pp_nb(&a[k]) != theia_wgrad_pp_nb(&a[0])) return THEIA_ERR_UNSUPPORTED;  // one tile form per launch
        g.prob[k] = a[k];
        g.count[k] = theia_wgrad_pp_tiles_shape(a[k].N, a[k].map.in_c) * a[k].splits;
        next += g.count[k];
    }
    constexpr int lds4 = 4 * 2 * 32 * 512, lds5 = 5 * 2 * 32 * 512;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wgrad_pp_group_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds4);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wgrad_pp_group_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, lds5);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>((gemm_wgrad_pp_group_kernel<4, 8>)), hipFuncAttributeMaxDynamicSharedMemorySize, lds4);
        attr_set = true;
    }
    static int stages = -1;
    if (stages < 0) {
        const char* e = getenv("THEIA_WGRAD_STAGES");
        stages = e != nullptr && atoi(e) == 5 ? 5 : 4;
    }
    const dim3 grid(g.first[n - 1] + g.count[n - 1]);
    if (theia_wgrad_pp_nb(&a[0]) == 8) hipLaunchKernelGGL((gemm_wgrad_pp_group_kernel<4, 8>), grid, dim3(512), lds4, stream, g, wgrad_plain_order());
    else if (stages == 5) hipLaunchKernelGGL(gemm_wgrad_pp_group_kernel<5>, grid, dim3(512), lds5, stream, g, wgrad_plain_order());
    else hipLaunchKernelGGL(gemm_wgrad_pp_group_kernel<4>, grid, dim3(512), lds4, stream, g, wgrad_plain_order());
    THEIA_CHECK_LAUNCH("theia_gemm_wgrad_group(pp)");
    for (int k = 0; k < n; ++k)
        if (a[k].bias_out != nullptr && a[k].defer_bias_reduce == 0) {
            hipLaunchKernelGGL(wgrad_bias_reduce_kernel, dim3(cdiv_i(a[k].N, 256)), dim3(256), 0, stream, a[k].bias_slabs, a[k].splits, a[k].N,
                               a[k].bias_out, a[k].bias_accumulate);
            THEIA_CHECK_LAUNCH("theia_gemm_wgrad_group(pp bias)");
        }
    return THEIA_OK;
}

// bf16 only; requires in_c % 64 == 0.  Returns THEIA_ERR_UNSUPPORTED when the shape does not qualify.
int theia_gemm_wgrad_pp_launch(const theia_wgrad_args_t* a, hipStream_t stream) {
    if (!theia_gemm_wgrad_pp_supported(a)) return THEIA_ERR_UNSUPPORTED;
    constexpr int lds4 = 4 * 2 * 32 * 512, lds5 = 5 * 2 * 32 * 512;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wgrad_pp_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds4);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wgrad_pp_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds4);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wgrad_pp_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds4);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wgrad_pp_kernel<true, true, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, lds5);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wgrad_pp_kernel<true, true, 4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds4);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wgrad_pp_kernel<true, true, 5, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds5);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wgrad_pp_kernel<true, true, 4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds4);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>((gemm_wgrad_pp_kernel<true, true, 4, 0, 8>)), hipFuncAttributeMaxDynamicSharedMemorySize, lds4);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>((gemm_wgrad_pp_kernel<true, true, 4, 0, 16, true>)), hipFuncAttributeMaxDynamicSharedMemorySize, lds4);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>((gemm_wgrad_pp_kernel<true, true, 4, 1, 8>)), hipFuncAttributeMaxDynamicSharedMemorySize, lds4);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>((gemm_wgrad_pp_kernel<true, true, 4, 2, 8>)), hipFuncAttributeMaxDynamicSharedMemorySize, lds4);
        attr_set = true;
    }
    static int stages = -1;  // THEIA_WGRAD_STAGES=4|5: ring depth of the stepping instantiation
    if (stages < 0) {
        const char* e = getenv("THEIA_WGRAD_STAGES");
        stages = e != nullptr && atoi(e) == 5 ? 5 : 4;
    }
    const int nb = theia_wgrad_pp_nb(a);
    const int tiles = (nb == 8 ? cdiv_i(a->N, 128) * cdiv_i(a->map.in_c, 384) : cdiv_i(a->N, 256) * cdiv_i(a->map.in_c, 256)) * a->map.ntaps;
    static int issue_in_m = -1;
    if (issue_in_m < 0) {
        const char* e = getenv("THEIA_WGRAD_ISSUE");
        issue_in_m = (e != nullptr && strcmp(e, "m") == 0) ? 1 : 0;
    }
    const int plain_order = wgrad_plain_order();
    const int mode = theia_gemm_wgrad_pp_mode(a);
    static int issue_r = -1;  // stepping row mode: both rows in the R segment (THEIA_WGRAD_ISSUE=split: A/B switch, one in each)
    if (issue_r < 0) {
        const char* e = getenv("THEIA_WGRAD_ISSUE");
        issue_r = (e != nullptr && strcmp(e, "split") == 0) ? 0 : 1;
    }
    const dim3 grid(tiles * a->splits);
    if (nb == 8 && mode == 11) hipLaunchKernelGGL((gemm_wgrad_pp_kernel<true, true, 4, 1, 8>), grid, dim3(512), lds4, stream, *a, plain_order);
    else if (nb == 8 && mode == 12) hipLaunchKernelGGL((gemm_wgrad_pp_kernel<true, true, 4, 2, 8>), grid, dim3(512), lds4, stream, *a, plain_order);
    else if (nb == 8) hipLaunchKernelGGL((gemm_wgrad_pp_kernel<true, true, 4, 0, 8>), grid, dim3(512), lds4, stream, *a, plain_order);
    else if (mode == 11 && stages == 5) hipLaunchKernelGGL((gemm_wgrad_pp_kernel<true, true, 5, 1>), grid, dim3(512), lds5, stream, *a, plain_order);
    else if (mode == 11) hipLaunchKernelGGL((gemm_wgrad_pp_kernel<true, true, 4, 1>), grid, dim3(512), lds4, stream, *a, plain_order);
    else if (mode == 12) hipLaunchKernelGGL((gemm_wgrad_pp_kernel<true, true, 4, 2>), grid, dim3(512), lds4, stream, *a, plain_order);
    else if (mode == 10 && issue_in_m) hipLaunchKernelGGL((gemm_wgrad_pp_kernel<true, false>), grid, dim3(512), lds4, stream, *a, plain_order);
    else if (mode == 10 && stages == 5) hipLaunchKernelGGL((gemm_wgrad_pp_kernel<true, true, 5>), grid, dim3(512), lds5, stream, *a, plain_order);
    else if (mode == 10 && issue_r) hipLaunchKernelGGL((gemm_wgrad_pp_kernel<true, true, 4, 0, 16, true>), grid, dim3(512), lds4, stream, *a, plain_order);
    else if (mode == 10) hipLaunchKernelGGL(gemm_wgrad_pp_kernel<true>, grid, dim3(512), lds4, stream, *a, plain_order);
    else hipLaunchKernelGGL(gemm_wgrad_pp_kernel<false>, grid, dim3(512), lds4, stream, *a, plain_order);
    THEIA_CHECK_LAUNCH("theia_gemm_wgrad(pp)");
    if (a->bias_out != nullptr && a->defer_bias_reduce == 0) {
        hipLaunchKernelGGL(wgrad_bias_reduce_kernel, dim3(cdiv_i(a->N, 256)), dim3(256), 0, stream, a->bias_slabs, a->splits, a->N,
                           a->bias_out, a->bias_accumulate);
        THEIA_CHECK_LAUNCH("theia_gemm_wgrad(pp bias)");
    }
    return THEIA_OK;
}
