// Persistent ping-pong NT implicit GEMM: BM x 256 tiles (BM = 256 or 320), 8 wave64 = two groups of four, one wave of each group
// per SIMD, ONE workgroup per CU that walks over its tiles.
//
// The two wave groups run the same program ONE BARRIER APART, and the program alternates two kinds of segments:
//     R(h): ds_read_b128 the operand fragments of half k-tile h          |  M(h): FM*4 x v_mfma_f32_16x16x32_bf16
// so whenever group 0 is in an M segment group 1 is in an R segment and vice versa: the matrix pipe of every SIMD is fed by
// one wave while the SIMD's other wave fetches its next fragments.  Operands arrive by LDS-DMA (global_load_lds_dwordx4) into a
// 4-deep ring of HALF k-tiles (32 bf16 / 16 f32 of k per row, 64-byte rows, source-side XOR swizzle); the pieces of half-tile
// g+3 are issued in R(g), right after the fragment reads, and only counted waits are used:
//     end of R(g):  s_waitcnt vmcnt(2 * pieces per half-tile)  -> half-tile g+1 has landed (g+2 and the just-issued g+3 in flight)
//
// What round 3 changed, and why (profiles/r02_pp_tile_phases.txt: a K = 768 tile was ~8k cycles of prologue + 35k of loop + 12-23k
// of epilogue, with nothing overlapping the two ends, and the N = 768 shapes of the ViT ran 297 tiles on 256 CUs = two rounds for
// 1.16 rounds of work):
//   * the half-tile stream is CONTINUOUS across a workgroup's tiles: the ring slot is (global half-tile counter) & 3, and the last
//     three iterations of a tile already request the first three half-tiles of the next one (they used to re-fetch the last
//     half-tile to keep the counted waits uniform).  The next tile's operands land while the epilogue runs;
//   * the epilogue is LDS-free (gemm_epi_direct.h: weight rows permuted at staging so that a lane's accumulators are contiguous
//     output columns) -- nothing in it collides with the ring, and the two LDS staging passes are gone;
//   * bias and (for act == NONE) the residual rows initialise the accumulators: the residual rows of a workgroup's first tile are
//     requested before the operand prologue and land behind it (they used to cost 8.5k cycles of the epilogue);
//   * BM = 320 (wave tile 160 x 64, 160 accumulator registers): 25216 rows = 79 tiles of 320, so the N = 768 launches are ONE
//     round of 237 tiles instead of two rounds of 256 + 41; 142 instead of 128 flop per L2->LDS byte;
//   * single-tap row maps (every nn.Linear) run an instantiation without the per-tap gather state (TAPS = false).
//
// Hazards (interval k = time between barrier k and k+1; group 0 runs segment k in interval k, group 1 segment k-1), g = global
// half-tile counter of the workgroup:
//   WAR  ring slot (g+3)&3 = (g-1)&3 is refilled from R(g) on (interval >= 2g); its last readers are R(g-1) of group 0
//        (interval 2g-2) and of group 1 (interval 2g-1), both closed by s_waitcnt lgkmcnt(0) before their barrier.  Across a tile
//        boundary the groups re-synchronise (one extra barrier each side of the epilogue), which only adds distance.
//   RAW  half-tile g+1 is first read in interval 2g+2 (group 0, R(g+1)); every wave has waited for its own pieces of it at
//        the end of its R(g), i.e. before barrier 2g+2.  The first half-tile of a later tile: s_waitcnt vmcnt(0) + barrier after
//        the epilogue (its stores share the counter with the LDS-DMA loads and retire out of order with respect to them, so no
//        counted wait is valid until they have drained; the epilogue's last stores are acknowledged within a few hundred cycles).
#include <type_traits>
#include "gemm_epi_direct.h"

// zeros read by out-of-range taps (multi-tap maps): such a lane's source pointer is the page start and advances with the k offset
// inside a tap like every other lane's, so the page covers one tap's row (in_c elements <= 16 KiB, checked by the dispatch)
__device__ uint4 g_pp_zero_page[1024 + 1];

// Optional cycle stamps (tools/pp_bench.hip builds this file with -DPP_TRACE): block 0 and the block that runs tile 1 of CU 0
#ifdef PP_TRACE
__device__ unsigned long long g_pp_phase[8][16];
#define PP_PHASE(k) \
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (k) < 16) g_pp_phase[threadIdx.x >> 6][k] = __builtin_readcyclecounter();
#else
#define PP_PHASE(k)
#endif

__device__ __forceinline__ int pp_f(int row) { return (4 - ((row >> 2) & 3)) & 3; }

// SGPRs the compiler may allocate: s100 / s101 stay out of its hands -- they carry the asynchronous tile draw of the work-conserving
// schedule from its issue to its wait (see "work-conserving schedule" in the kernel)
#define PP_NUM_SGPR 96

template <int N> __device__ __forceinline__ void pp_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// T: operand type.  BM: tile rows.  TAPS: multi-tap (3x3 gather) row maps; false = single-tap maps only.  SUMS: ln_sums epilogue.
template <typename T, int BM, bool TAPS, bool SUMS>
__global__ __launch_bounds__(512) __attribute__((amdgpu_num_sgpr(PP_NUM_SGPR))) void gemm_nt_pp_kernel(const theia_gemm_args_t p, const int ntiles,
                                                                                                     const int panel, unsigned* const sched,
                                                                                                     const int dephase) {
    constexpr int BN = 256, WAVES_N = 4;
    constexpr int NSTAGE = 4;
    constexpr int HKT = 64 / (int)sizeof(T);
    constexpr int EPC = 16 / (int)sizeof(T);
    constexpr int WM = BM / 2, WN = 64, FM = WM / 16, FN = 4;
    constexpr int SRP = 128;                      // rows staged per pass (512 threads x 16 B = 128 rows of 64 B)
    constexpr int NPA = (BM + SRP - 1) / SRP;      // A passes; the last one covers BM % 128 rows when BM is not a multiple of 128
    constexpr int A_TAIL_WAVES = (BM % SRP) / 16;  // waves that take part in the partial A pass (0 = every pass is full)
    constexpr int NPB = BN / SRP;
    constexpr int LPH_FULL = NPA + NPB;            // pieces per half-tile of a wave that takes part in every pass
    constexpr int LPH_PART = LPH_FULL - (A_TAIL_WAVES ? 1 : 0);
    constexpr int STAGE = (BM + BN) * 64;
    constexpr bool SCALE = sizeof(T) == 1;
    constexpr bool BIAS_IN_ACC = !SCALE;
    using OutT = typename std::conditional<sizeof(T) == 1, bf16_t, T>::type;  // fp8 operands: bf16 out, accumulators rescaled
    static_assert(BM % 32 == 0 && WM % 32 == 0, "two wave groups of whole fragment pairs");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int uwave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ugroup = uwave >> 2;
    const int wm = ugroup, wn = uwave & (WAVES_N - 1);  // wm = wave group (0: waves 0-3, 1: waves 4-7); scalars
    const bool tailw = A_TAIL_WAVES == 0 || uwave < A_TAIL_WAVES;  // this wave issues the partial A pass
    const theia_rowmap_t& mp = p.map;
    const int tiles_n = (p.N + BN - 1) / BN;
    PP_PHASE(0)
    // tiles of this workgroup: round r of the grid takes tiles [r * grid, r * grid + cnt), spread over the XCDs like a launch of cnt
    const int grid = gridDim.x, bid = blockIdx.x;
    const int rounds = (ntiles + grid - 1) / grid;
    const int cnt_last = ntiles - (rounds - 1) * grid;
    const int my_tiles = rounds - 1 + (bid < cnt_last ? 1 : 0);
    // Schedule position -> tile (row block * tiles_n + column block).  panel == 0: row-major -- a round's 32 consecutive positions of an
    // XCD are ~32 / tiles_n row blocks x every column block: each activation row block enters one L2 once, and the XCD streams the
    // whole weight matrix every round.  That is the right trade while the weights fit in the 4 MB L2 beside the activation stream; for
    // N = 3072, K = 768 (4.7 MB) they do not, and the PMC summary showed 2.4x the algorithmic fetch for those launches (the weights
    // re-fetched by every XCD in every round).  panel = G > 0: positions run down panels of G column blocks (row blocks fastest across a
    // panel's G columns), so an XCD keeps G weight column blocks hot across rounds and re-reads each activation row block once per panel.
    auto tile_of = [&](int r) {
        const int t = r * grid + gt_xcd_remap(bid, r + 1 < rounds ? grid : cnt_last);
        if (panel <= 0) return t;
        const int tiles_m = ntiles / tiles_n;
        const int per = panel * tiles_m, full = tiles_n / panel;
        const int pi = min(t / per, full);                       // panel index; the last one may be narrower
        const int cols = pi < full ? panel : tiles_n - full * panel;
        const int rem = t - pi * per;
        const int m = rem / cols, n = pi * panel + (rem - m * cols);
        return m * tiles_n + n;
    };
    // ---------------------------------------------------------------- work-conserving schedule (sched != nullptr; round 5)
    // With static ownership a workgroup that starts late -- its CU held by an RCCL channel, or by a weight-gradient workgroup of the side
    // stream -- still owns a full share of the tiles, so the launch takes as long as that CU stays away plus a whole share.  Dynamic
    // mode: the schedule positions above become eight QUEUES, one per XCD (position r * grid + q belongs to the XCD whose contiguous
    // range of round r holds q: the same L2 locality as the static rounds), and a workgroup draws its next item from the queue of the XCD
    // it runs on with a SCALAR atomic (s_atomic_add ... glc on a per-launch counter; tracked by lgkmcnt, which every R segment waits to
    // zero anyway; the counter lives in that XCD's L2 and only that XCD's workgroups touch it: tools/experiments/satomic_probe_not_kept.patch --
    // ~760 cycles per draw, every queue a permutation).  The draw for the next tile is issued by wave 0 six half-tiles before this tile
    // ends and read one half-tile later; the result reaches the other waves through an LDS word, in time for the operand stream to cross
    // into it.  Between issue and wait the returned value sits in s100, an SGPR the compiler cannot allocate (amdgpu_num_sgpr): as an
    // ordinary asm output it was copied into the loop-carried register right behind the issue, before the atomic had returned, and
    // keeping issue and wait in one straight-line block (a second copy of the M segment) cost 700 spills.
    // A launch degrades in proportion to the CUs it actually gets; a workgroup that finds its queue empty leaves.
    const bool dyn = sched != nullptr;
    // De-phasing (round 6; THEIA_PP_DEPHASE=0: off, =<units of ~2k cycles>[+65536][+131072]: forced; profiles/r06_dephase_experiment.txt):
    // in a multi-round launch every CU is in its MFMA phase (chip at the power cap, HBM idle) and then in its store burst (MFMA idle)
    // at the same time.  The workgroups that own one tile fewer than the others idle through the last round anyway: started late by
    // part of a tile (spread over (0, units], the host's estimate of 2/3 of a tile), their epilogues fall into the others' main loops
    // at no cost to the launch: fc2 data-gradient 171-181 -> 166-170 us, fc1 forward 166-170 -> 164-167, the step -0.15 ms; the
    // result is unaffected (same tiles, same order of additions).  +65536 (every odd workgroup as well: costs more than it gains).
    if (dephase != 0 && !dyn && rounds > 1 && (my_tiles < rounds || ((dephase >> 16) & (bid & 1)))) {
        int units = dephase & 0xffff;
        if ((dephase >> 17) & 1) units = my_tiles < rounds ? 1 + ((bid - cnt_last) * units) / (grid - cnt_last) : units;  // +131072: spread over (0, units]
        for (int i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(32);
    }
    int xcc = 0;
    if (dyn) {
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7;
    }
    unsigned* const my_counter = sched + xcc * 32;  // 128 bytes apart: one line per XCD
    // item k of XCD x's queue -> tile (or -1 past the end)
    auto item_tile = [&](int x, int k) {
        const int nfull = (grid >> 3) + (x < (grid & 7) ? 1 : 0), nlast = (cnt_last >> 3) + (x < (cnt_last & 7) ? 1 : 0);
        const int head = (rounds - 1) * nfull;
        int r, idx;
        if (k < head) {
            r = k / nfull;
            idx = k - r * nfull;
        } else {
            r = rounds - 1;
            idx = k - head;
            if (idx >= nlast) return -1;
        }
        const int t = r * grid + gt_xcd_remap(idx * 8 + x, r + 1 < rounds ? grid : cnt_last);
        if (panel <= 0) return t;
        const int tiles_m = ntiles / tiles_n;
        const int per = panel * tiles_m, full = tiles_n / panel;
        const int pi = min(t / per, full);
        const int cols = pi < full ? panel : tiles_n - full * panel;
        const int rem = t - pi * per;
        const int m = rem / cols, n = pi * panel + (rem - m * cols);
        return m * tiles_n + n;
    };
    int* const mailbox = reinterpret_cast<int*>(smem + NSTAGE * STAGE + 2048 + 64);
    auto claim_issue = [&]() {  // s100: 1 going in, the queue position coming back (pre-op value)
        asm volatile("s_mov_b32 s100, 1\n\ts_atomic_add s100, %0, 0x0 glc" ::"s"(my_counter) : "s100", "memory");
    };
    auto claim_publish = [&]() {  // the draw has returned: position -> tile, into the LDS word the other waves read
        int k;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, s100" : "=s"(k)::"memory");
        const int t = item_tile(xcc, k);
        asm volatile("ds_write_b32 %0, %1" ::"v"(gt_lds_addr(mailbox)), "v"(t) : "memory");
    };
    auto claim_read = [&]() {
        int t;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(gt_lds_addr(mailbox)) : "memory");
        return __builtin_amdgcn_readfirstlane(t);
    };
    const T* __restrict__ A = reinterpret_cast<const T*>(p.a);
    const T* __restrict__ W = reinterpret_cast<const T*>(p.w);
    const int R = mp.rows_h * mp.rows_w;
    const int nh = (p.K + HKT - 1) / HKT;   // host guarantees K % HKT == 0 for this kernel
    const int hpt = mp.in_c / HKT;          // half-tiles per tap
    const uint64_t zp = reinterpret_cast<uint64_t>(g_pp_zero_page);

    // ---------------------------------------------------------------- operand prefetch stream
    // State: the tile (round pf_r) and half-tile pf_h that the NEXT issue fetches, this thread's source pointers for that tile
    // (TAPS: for the current tap, recomputed when the stream enters a new tap), all wave-uniform except the pointers.
    uint64_t src_ptr[LPH_FULL];
    int pf_r = 0, pf_h = 0, cur_tap = 0, next_tap_h = hpt, pf_m0 = 0, pf_n0 = 0;
    int tile_next = -1;  // the tile after the one being computed (-1: none, or -- dynamic mode -- not drawn yet)
    // source pointers of tile (pf_m0, pf_n0) for tap `tap`.  Multi-tap maps: out-of-range taps read the zero page; the row decode is
    // redone at every tap switch (once per in_c / HKT half-tiles) rather than carried in registers across the whole kernel.
    // Single-tap maps (TAPS = false): rows beyond M / N are clamped to the last row (their products land in rows / columns that are
    // never stored), so no zero page.
    auto set_tap = [&](int tap) {
        // lane constants and reciprocals are re-derived HERE from an opaque copy of the thread index / row counts: hoisted to the top
        // of the kernel they are registers alive across the main loop, which has none to spare at BM = 320
        int tid_ = threadIdx.x, R_ = R, rows_w_ = mp.rows_w;
        asm volatile("" : "+v"(tid_));
        asm volatile("" : "+s"(R_), "+s"(rows_w_));
        const int st_chunk = tid_ & 3, st_row = tid_ >> 2;
        const int lchunk = st_chunk ^ pp_f(st_row);
        const float rcp_R = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(1.0f / (float)R_)));
        const float rcp_w = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(1.0f / (float)rows_w_)));
        const int dy = mp.dy[tap], dx = mp.dx[tap];
        const int64_t wcol = (int64_t)mp.wslot[tap] * mp.in_c + lchunk * EPC;
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            int m = pf_m0 + st_row + SRP * i;
            bool ok = true;
            if constexpr (TAPS) ok = m < p.M;
            m = min(m, p.M - 1);
            int64_t off;
            if (!TAPS && R == 1) {
                off = (int64_t)m * mp.in_batch_stride + mp.in_offset;
            } else {
                int rem, rx;
                const int img = gt_divmod24(m, R_, rcp_R, rem);
                const int ry = gt_divmod24(rem, rows_w_, rcp_w, rx);
                const int iy = ry * mp.in_sy + dy, ix = rx * mp.in_sx + dx;
                if constexpr (TAPS) ok = ok & (iy >= 0) & (iy < mp.in_h) & (ix >= 0) & (ix < mp.in_w);
                off = (int64_t)img * mp.in_batch_stride + mp.in_offset + (int64_t)(iy * mp.in_w + ix) * mp.in_c;
            }
            const uint64_t pa = reinterpret_cast<uint64_t>(A + off + lchunk * EPC);
            if constexpr (TAPS) {
                const uint64_t msk = 0ull - (uint64_t)ok;
                src_ptr[i] = (pa & msk) | (zp & ~msk);
            } else {
                src_ptr[i] = pa;
            }
        }
#pragma unroll
        for (int i = 0; i < NPB; ++i) {
            int n = pf_n0 + gd_wperm(st_row + SRP * i);
            const bool ok = n < p.N;
            n = min(n, p.N - 1);
            const uint64_t pw = reinterpret_cast<uint64_t>(W + (int64_t)n * p.ldw + wcol);
            if constexpr (TAPS) {
                const uint64_t msk = 0ull - (uint64_t)ok;
                src_ptr[NPA + i] = (pw & msk) | (zp & ~msk);
            } else {
                src_ptr[NPA + i] = pw;
            }
        }
    };
    auto setup_tile = [&](int tile) {
        pf_m0 = (tile / tiles_n) * BM;
        pf_n0 = (tile % tiles_n) * BN;
        cur_tap = 0;
        next_tap_h = hpt;
        set_tap(0);
    };
    // pf_issue: the pieces of half-tile (pf_r, pf_h) into ring slot `slot`, straight from src_ptr[] -- which always points at the NEXT
    // half-tile to fetch (inside its tap) and moves one 64-byte row chunk per issue: ONE live copy of each pointer (a base + offset form
    // made the compiler carry an incrementing copy next to the base through the loop).
    // pf_advance: everything that is not straight-line -- entering the next tap (multi-tap maps), the next tile (address set-up), or,
    // past the workgroup's last half-tile, stepping back to fetch it again (never read; keeps the counted waits uniform).  It runs
    // right AFTER an M segment: the fragment registers are dead there (56 of them in the 320-row instantiations), so the address
    // arithmetic of a tap / tile switch has room -- inside the R segment, between the asynchronous fragment reads and their wait, it
    // cost scratch spills (and a spilled or reused register with an LDS read in flight is garbage or a memory fault).
    auto pf_issue = [&](int slot) {
        char* sa = smem + slot * STAGE + uwave * (16 * 64);
        char* sb = sa + BM * 64;
#pragma unroll
        for (int q = 0; q < LPH_FULL; ++q) {
            if (A_TAIL_WAVES != 0 && q == NPA - 1 && !tailw) continue;  // wave-uniform
            char* dst = q < NPA ? sa + q * (SRP * 64) : sb + (q - NPA) * (SRP * 64);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_ptr[q],
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            src_ptr[q] += 64;
        }
        ++pf_h;
    };
    auto pf_advance = [&]() {
        if (pf_h == nh) {  // wave-uniform
            // static mode: the stream keeps its own round counter (with K of one to three half-tiles it crosses more than one tile
            // boundary per tile); dynamic mode (K >= 8 half-tiles): it enters the next tile at most once per tile, and that tile is known
            if (dyn ? tile_next >= 0 : pf_r + 1 < my_tiles) {
                ++pf_r;
                pf_h = 0;
                setup_tile(dyn ? tile_next : tile_of(pf_r));
            } else {
                pf_h = nh - 1;
#pragma unroll
                for (int q = 0; q < LPH_FULL; ++q) src_ptr[q] -= 64;
            }
        } else if constexpr (TAPS) {
            if (pf_h >= next_tap_h) {
                ++cur_tap;
                next_tap_h += hpt;
                set_tap(cur_tap);
            }
        }
    };
    // wait until at most `halves` half-tiles' worth of this wave's pieces are outstanding
    auto wait_halves2 = [&]() {
        if (A_TAIL_WAVES != 0 && !tailw) pp_wait_vm<2 * LPH_PART>();
        else pp_wait_vm<2 * LPH_FULL>();
    };

    // ---------------------------------------------------------------- accumulator initialisation: bias (+ residual) rows
    const OutT* __restrict__ RES = reinterpret_cast<const OutT*>(p.resid);
    const bool want_aux = p.act == THEIA_ACT_MUL_DGELU || p.act == THEIA_ACT_MUL_DRELU;
    // the residual rows can start the accumulation when nothing sits between the product and the addition (bf16 rows only)
    const bool resid_init = BIAS_IN_ACC && sizeof(OutT) == 2 && RES != nullptr && p.act == THEIA_ACT_NONE;
    gt_f32x4 acc[FN][FM];
    // Starts a tile: requests the bias row and (resid_init) the residual rows of the wave tile, waits, and initialises the
    // accumulators with them.  FIRST: the workgroup's first tile -- the rows are requested BEFORE the operand prologue (loads retire
    // in order: they have landed once only the prologue's pieces are outstanding); later tiles: after the previous epilogue, whose
    // stores share the counter, so everything is drained (the next tile's first half-tiles have been in flight since the last
    // iterations of the previous tile).  The row registers live only inside this function.
    auto start_tile = [&](auto FIRST_C, int tile_id, int m_wave0, int n_wave0) {
        constexpr bool FIRST = decltype(FIRST_C)::value;
        const int frow = threadIdx.x & 15, fg = (threadIdx.x >> 4) & 3;
        gt_u32x4 brow[2][2];  // bias: 2 column groups x 8 floats
        const bool has_bias = BIAS_IN_ACC && p.bias != nullptr;
        // The rows arrive by inline-asm loads (invisible to the compiler's waitcnt bookkeeping) -- and to its register allocator: between
        // such a load and the wait that covers it, the destination registers must not be spilled or reused.  The 256-row instantiations
        // are spill-free (tests/test_kernel_resources.py holds them to that), so their first tile requests the rows BEFORE the operand
        // prologue and lets them land behind it; the 320-row instantiations keep a few long-lived values in scratch around tile
        // boundaries, so there the request, the wait and the use are adjacent (a destination register reused for address arithmetic while
        // its load was in flight was a memory fault).
        constexpr bool EARLY_ROWS = FIRST && BM == 256;
        // EARLY_ROWS: inline-asm loads (waited for by hand).  Otherwise ordinary loads: the compiler tracks their registers and waits
        // (with the LDS-DMA queue in flight it waits for everything -- which this path does anyway).
        auto row_load = [&](auto& dst, const void* ptr) {
            if constexpr (EARLY_ROWS) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory");
            else dst = __builtin_bit_cast(typename std::remove_reference<decltype(dst)>::type, *reinterpret_cast<const gt_u32x4*>(ptr));
        };
        auto request_rows = [&]() {
            if constexpr (BIAS_IN_ACC) {
                if (has_bias) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int n = n_wave0 + t * 32 + fg * 8;
                        const float* bp = p.bias + (n < p.N ? n : 0);
                        row_load(brow[t][0], bp);
                        row_load(brow[t][1], bp + 4);
                    }
                }
                if (resid_init) {
                    const gd_rows_t rw(p);
                    int dry, drx;
                    const int64_t off_dead = rw.decode(p, 0, dry, drx);
                    gd_rows_t::cursor_t c = rw.first(p, m_wave0 + frow);
#pragma unroll
                    for (int j = 0; j < FM; ++j) {
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            const int n = n_wave0 + t * 32 + fg * 8;
                            const bool lv = (c.m < p.M) && (n < p.N);
                            // straight into the accumulator registers of fragment (2t, j): 8 bf16 = 4 dwords, expanded in place below
                            // (a separate row buffer would be 16 * FM more registers alive next to the full accumulator set)
                            row_load(acc[2 * t][j], RES + (lv ? c.off + n : off_dead));
                        }
                        rw.next(c);
                    }
                }
            }
        };
        if constexpr (EARLY_ROWS) request_rows();
        if constexpr (FIRST) {
            setup_tile(tile_id);
            PP_PHASE(1)
#pragma unroll
            for (int s = 0; s < NSTAGE - 1; ++s) {
                pf_issue(s);
                pf_advance();
            }
            PP_PHASE(2)
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (EARLY_ROWS) {  // loads retire in order: the rows have landed once only the prologue's pieces are outstanding
            if (A_TAIL_WAVES != 0 && !tailw) pp_wait_vm<3 * LPH_PART>();
            else pp_wait_vm<3 * LPH_FULL>();
        } else {
            request_rows();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if constexpr (BIAS_IN_ACC) {
            float b8[2][8];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const bool ok = has_bias && (n_wave0 + t * 32 + fg * 8) < p.N;
                if (has_bias) {
                    asm volatile("" : "+v"(brow[t][0]));
                    asm volatile("" : "+v"(brow[t][1]));
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        b8[t][e] = ok ? __uint_as_float(brow[t][0][e]) : 0.f;
                        b8[t][4 + e] = ok ? __uint_as_float(brow[t][1][e]) : 0.f;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) b8[t][e] = 0.f;
                }
            }
            if (resid_init) {
#pragma unroll
                for (int j = 0; j < FM; ++j)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        asm volatile("" : "+v"(acc[2 * t][j]));  // (use stays behind the wait above)
                        const gt_u32x4 raw = __builtin_bit_cast(gt_u32x4, acc[2 * t][j]);
                        float r8[8];
                        gd_unpack8(raw, r8);
                        acc[2 * t][j] = (gt_f32x4){b8[t][0] + r8[0], b8[t][1] + r8[1], b8[t][2] + r8[2], b8[t][3] + r8[3]};
                        acc[2 * t + 1][j] = (gt_f32x4){b8[t][4] + r8[4], b8[t][5] + r8[5], b8[t][6] + r8[6], b8[t][7] + r8[7]};
                    }
            } else {
#pragma unroll
                for (int j = 0; j < FM; ++j)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        acc[2 * t][j] = (gt_f32x4){b8[t][0], b8[t][1], b8[t][2], b8[t][3]};
                        acc[2 * t + 1][j] = (gt_f32x4){b8[t][4], b8[t][5], b8[t][6], b8[t][7]};
                    }
            }
        } else {
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < FM; ++j) acc[i][j] = (gt_f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j) asm volatile("" : "+v"(acc[i][j]));  // materialise HERE (not sunk behind the next wait)
    };

    // ---------------------------------------------------------------- first tile: init rows, then the operand prologue
    int tile;
    if (dyn) {  // the first tile is drawn like every other one: a workgroup that starts late must not own anything
        if (uwave == 0) {
            claim_issue();
            claim_publish();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        tile = claim_read();
        if (tile < 0) return;  // (wave-uniform) every item of this XCD's queue has been taken by workgroups that started earlier
    } else {
        tile = tile_of(0);
        tile_next = my_tiles > 1 ? tile_of(1) : -1;
    }
    int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    start_tile(std::true_type{}, tile, m0 + wm * WM, n0 + wn * WN);
    __builtin_amdgcn_sched_barrier(0);
    wait_halves2();  // half-tile 0 landed

    gt_u32x4 fb[FN], fa[FM];
    int g = 0;  // global half-tile counter of this workgroup: ring slot = g & 3

    // Tile boundaries: the groups re-join (group 0 executes one barrier more), run their epilogues CONCURRENTLY (one group's
    // conversions overlap the other's store issue: serialised -- group 0's epilogue beside group 1's last M segment and vice versa, no
    // re-join -- the two epilogues took 9.6k + 10k cycles instead of 14.5k together, measured), and stagger again.  What the boundary
    // does NOT do any more is drain the memory queue:
    //  * the ring invariant carries over (slot = g & 3 with g running across tiles; half-tile g+1 has been waited for and fenced by
    //    the barrier at the end of R(g), whichever tile it belongs to);
    //  * the epilogue's stores stay in flight into the next loop.  That is legal for the loop's counted waits: vmcnt(N) guarantees
    //    that at most N operations are outstanding, loads retire in order among themselves, so "the pieces of half-tile g+1 have
    //    landed" follows whatever the stores do -- a store still in flight only makes the wait stricter (it can never be satisfied by
    //    loads that have not retired).  What is NOT legal is waiting for loads issued after stores with a non-zero count; the only
    //    such loads are a next tile's residual rows (resid_init), and that path drains the queue (start_tile);
    //  * the next tile's bias row (256 floats = one 1 KiB LDS-DMA piece) is fetched by wave 0 into LDS behind the ring (two slots,
    //    alternating per tile: a wave may still be reading this tile's row) three iterations before the tile ends: the loop's own
    //    counted waits + barriers cover it (two iterations later at the latest), and no register holds it across the epilogue.
    //    K < 3 half-tiles: the draining path instead.
    constexpr int BIAS_LDS = NSTAGE * STAGE;
    const bool seamless = !resid_init && nh >= NSTAGE - 1;
    // SUMS: the waves' per-image partials of a tile meet in this table; wave 0 adds the previous tile's totals to global memory right
    // behind the barrier at the top of the next tile (every wave has left its epilogue by then) and clears the table -- the next
    // LDS additions are a whole main loop of barriers away
    unsigned long long* const sums_tab = reinterpret_cast<unsigned long long*>(smem + BIAS_LDS + 2048);
    int flush_img0 = -1;  // image base of the tile whose totals are still in the table
    if constexpr (SUMS) {
        if (threadIdx.x < 2 * GT_SUMS_SLOTS) sums_tab[threadIdx.x] = 0ull;  // (first use: behind every barrier of the first main loop)
    }
    const int h_claim = nh - 6;  // dynamic mode (host: nh >= 8): wave 0 draws at h_claim, publishes at h_claim + 1, every wave reads at nh - 4
    for (int r = 0;; ++r) {
        // Fragment addresses: rows 16 apart share the swizzle (pp_f looks at bits 2..3 of the row), so the FM A fragments / 4 B
        // fragments of a wave are 1 KiB apart: one lane-dependent offset each + immediates.  The reads are inline asm (see
        // gt_ds_read128: a C++ load here would make the compiler drain every in-flight LDS-DMA at the top of each iteration).
        // Recomputed per tile so that they do not occupy registers during the epilogue.
        const uint32_t smem_base = gt_lds_addr(smem);
        const int frow_ = threadIdx.x & 15, fg_ = (threadIdx.x >> 4) & 3;
        const uint32_t lane_a = smem_base + (wm * WM + frow_) * 64 + ((fg_ ^ pp_f(frow_)) << 4);
        const uint32_t lane_b = smem_base + BM * 64 + (wn * WN + frow_) * 64 + ((fg_ ^ pp_f(frow_)) << 4);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (SUMS) {
            if (uwave == 0 && flush_img0 >= 0)
                gt_flush_sums(sums_tab, reinterpret_cast<unsigned long long*>(p.ln_sums), flush_img0, p.map.rows_h * p.map.rows_w, p.M, threadIdx.x & 63);
        }
        if (ugroup == 1) {  // group 1 runs one barrier behind group 0
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        PP_PHASE(3 + 4 * r)
        // the tile after this one: static mode knows it now; dynamic mode learns it at h = nh - 4, in front of the pf_advance() that
        // enters it (the operand stream runs three half-tiles ahead) and of the bias fetch at nh - 3
        if (dyn) tile_next = -1;
        else tile_next = r + 1 < my_tiles ? tile_of(r + 1) : -1;
        for (int h = 0; h < nh; ++h, ++g) {
            const uint32_t soff = (uint32_t)(g & (NSTAGE - 1)) * STAGE;
            // three iterations before the tile ends: covered by the loop's own waits (wave-uniform condition)
            if (BIAS_IN_ACC && seamless && p.bias != nullptr && uwave == 0 && tile_next >= 0 && h == nh - (NSTAGE - 1)) {
                const int n0_next = (tile_next % tiles_n) * BN;
                const int n = min(n0_next + (int)(threadIdx.x & 63) * 4, p.N - 4);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.bias + n),
                                                 (__attribute__((address_space(3))) void*)(smem + BIAS_LDS + ((r + 1) & 1) * 1024), 16, 0, 0);
            }
            // ---------------- R(g): FN + FM fragment reads + the LDS-DMA pieces of half-tile g+3
            const uint32_t ab = lane_b + soff, aa = lane_a + soff;
            gd_static_for<0, FN>([&](auto I) { gt_ds_read128<decltype(I)::value * 1024>(fb[decltype(I)::value], ab); });
            gd_static_for<0, FM>([&](auto J) { gt_ds_read128<decltype(J)::value * 1024>(fa[decltype(J)::value], aa); });
            pf_issue((g + NSTAGE - 1) & (NSTAGE - 1));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < FN; ++i) asm volatile("" : "+v"(fb[i]));
#pragma unroll
            for (int j = 0; j < FM; ++j) asm volatile("" : "+v"(fa[j]));
            wait_halves2();  // half-tile g+1 landed; g+2, g+3 may be in flight
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---------------- M(g): FM * FN MFMAs
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < FM; ++j) {
#pragma unroll
                for (int i = 0; i < FN; ++i) gt_mma<T>(acc[i][j], fb[i], fa[j]);
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (dyn) {  // (wave-uniform branches; nothing here in static mode)
                if (h == h_claim) {
                    if (uwave == 0) claim_issue();
                } else if (h == h_claim + 1) {
                    if (uwave == 0) claim_publish();
                } else if (h == h_claim + 2) {
                    tile_next = claim_read();
                }
            }
            pf_advance();  // tap / tile switch of the prefetch stream, if the next fetch needs one (fragment registers are dead here)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ugroup == 0) {  // re-join the groups
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        PP_PHASE(4 + 4 * r)
        {
            const gd_rows_t rw(p);
            const int it0 = gd_epilogue<OutT, FM, SUMS, SCALE, BIAS_IN_ACC>(acc, p, rw, m0 + wm * WM, n0 + wn * WN, threadIdx.x & 63, resid_init, sums_tab, m0);
            if constexpr (SUMS) flush_img0 = it0;
        }
        PP_PHASE(5 + 4 * r)
        if (tile_next >= 0) {
            tile = tile_next;
            m0 = (tile / tiles_n) * BM;
            n0 = (tile % tiles_n) * BN;
            if (!seamless) {
                start_tile(std::false_type{}, tile, m0 + wm * WM, n0 + wn * WN);  // rows requested now, queue drained
            } else if constexpr (BIAS_IN_ACC) {  // accumulators = the bias row in LDS (landed and fenced: see above)
                const int fg_b = (threadIdx.x >> 4) & 3;
                float b8[2][8];
                if (p.bias != nullptr) {
                    gt_u32x4 bl[2][2];
                    const uint32_t ba = gt_lds_addr(smem) + BIAS_LDS + ((r + 1) & 1) * 1024 + (wn * WN + fg_b * 8) * 4;
                    gt_ds_read128<0>(bl[0][0], ba);
                    gt_ds_read128<16>(bl[0][1], ba);
                    gt_ds_read128<128>(bl[1][0], ba);
                    gt_ds_read128<144>(bl[1][1], ba);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        asm volatile("" : "+v"(bl[t][0]));
                        asm volatile("" : "+v"(bl[t][1]));
                        const bool ok = (n0 + wn * WN + t * 32 + fg_b * 8) < p.N;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            b8[t][e] = ok ? __uint_as_float(bl[t][0][e]) : 0.f;
                            b8[t][4 + e] = ok ? __uint_as_float(bl[t][1][e]) : 0.f;
                        }
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int e = 0; e < 8; ++e) b8[t][e] = 0.f;
                }
#pragma unroll
                for (int j = 0; j < FM; ++j)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        acc[2 * t][j] = (gt_f32x4){b8[t][0], b8[t][1], b8[t][2], b8[t][3]};
                        acc[2 * t + 1][j] = (gt_f32x4){b8[t][4], b8[t][5], b8[t][6], b8[t][7]};
                    }
#pragma unroll
                for (int i = 0; i < FN; ++i)
#pragma unroll
                    for (int j = 0; j < FM; ++j) asm volatile("" : "+v"(acc[i][j]));
            } else {
#pragma unroll
                for (int i = 0; i < FN; ++i)
#pragma unroll
                    for (int j = 0; j < FM; ++j) {
                        acc[i][j] = (gt_f32x4){0.f, 0.f, 0.f, 0.f};
                        asm volatile("" : "+v"(acc[i][j]));
                    }
            }
        }
        PP_PHASE(6 + 4 * r)
        if (tile_next < 0) break;
    }
    if constexpr (SUMS) {  // the last tile's totals
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (uwave == 0 && flush_img0 >= 0)
            gt_flush_sums(sums_tab, reinterpret_cast<unsigned long long*>(p.ln_sums), flush_img0, p.map.rows_h * p.map.rows_w, p.M, threadIdx.x & 63);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped tail fetches write LDS: let them land before the workgroup exits
}

// ---------------------------------------------------------------------------------------------------------------- launch
#ifndef PP_DYNAMIC_DEFAULT
#define PP_DYNAMIC_DEFAULT 0
#endif
int g_pp_grid_cap = 0;  // > 0: cap on the persistent grid (tools/pp_bench.hip: forces several tiles per workgroup on small problems)
static int pp_num_cus() {
    static int forced = -1;
    if (forced < 0) {
        const char* e = getenv("THEIA_PP_GRID");  // timing experiments: cap (or, with a huge value, lift) the persistent grid
        forced = e != nullptr && atoi(e) > 0 ? atoi(e) : 0;
    }
    return g_pp_grid_cap > 0 ? g_pp_grid_cap : forced > 0 ? forced : theia_compute_cus();
}

// ---- per-launch tile counters of the work-conserving schedule -------------------------------------------------------------------------
// One block of 8 counters (one 128-byte line per XCD) per launch, zero when the launch starts.  Blocks come from a per-stream arena that is
// cleared with ONE stream-ordered memset when it wraps (every PP_SCHED_BLOCKS launches): nothing in the kernel resets a counter, so a
// launch needs no "last workgroup out" protocol, and the stream order makes the clear safe (every earlier launch of the stream has finished,
// every later one starts behind it).  A launch recorded into a stream capture would replay with the counters of its first run: captured
// launches use the static schedule.  THEIA_PP_DYNAMIC=0/1: off / on.
#include <map>
#include <mutex>
#include <unordered_map>
#include <utility>
constexpr int PP_SCHED_BLOCKS = 4096, PP_SCHED_BLOCK_BYTES = 8 * 128;
struct pp_sched_arena_t { unsigned char* base = nullptr; int next = 0; };
static int pp_dynamic_mode() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("THEIA_PP_DYNAMIC");
        v = e == nullptr ? PP_DYNAMIC_DEFAULT : (atoi(e) != 0 ? 1 : 0);
    }
    return v;
}
int g_pp_dynamic_override = -1;  // -1 = environment, 0 / 1 = set through theia_set_gemm_schedule (or by tools/pp_bench.hip)
#ifndef PP_NO_ABI
extern "C" int theia_get_gemm_schedule(void) { return g_pp_dynamic_override >= 0 ? g_pp_dynamic_override : pp_dynamic_mode(); }
extern "C" int theia_set_gemm_schedule(int dynamic) {
    const int prev = theia_get_gemm_schedule();
    g_pp_dynamic_override = dynamic != 0 ? 1 : 0;
    return prev;
}
#endif
static int pp_device_xccs() {  // XCDs of the current device as a launch sees them (per device: the answer differs between partition modes)
    static std::mutex mu;
    static std::unordered_map<int, int> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(dev);
    if (it != cache.end()) return it->second;
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeNumberOfXccs, dev) != hipSuccess) { (void)hipGetLastError(); n = 0; }
    cache[dev] = n;
    return n;
}
static unsigned* pp_sched_block(hipStream_t stream) {
    static std::mutex mu;
    // keyed by (device, stream): the null / per-thread stream handles are the same value on every device
    static std::map<std::pair<int, hipStream_t>, pp_sched_arena_t> arenas;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return nullptr;
    }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    std::lock_guard<std::mutex> lock(mu);
    pp_sched_arena_t& ar = arenas[std::make_pair(dev, stream)];
    if (ar.base == nullptr) {
        if (hipMalloc(reinterpret_cast<void**>(&ar.base), (size_t)PP_SCHED_BLOCKS * PP_SCHED_BLOCK_BYTES) != hipSuccess) {
            (void)hipGetLastError();
            ar.base = nullptr;
            return nullptr;
        }
        ar.next = PP_SCHED_BLOCKS;  // (cleared on first use, in stream order)
    }
    if (ar.next >= PP_SCHED_BLOCKS) {
        if (hipMemsetAsync(ar.base, 0, (size_t)PP_SCHED_BLOCKS * PP_SCHED_BLOCK_BYTES, stream) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        ar.next = 0;
    }
    return reinterpret_cast<unsigned*>(ar.base + (size_t)(ar.next++) * PP_SCHED_BLOCK_BYTES);
}

template <typename T, int BM, bool TAPS, bool SUMS>
static int pp_launch_one(const theia_gemm_args_t* a, hipStream_t stream) {
    constexpr int lds = 4 * (BM + 256) * 64 + 2048 + 64 + 64;  // operand ring + two bias rows (this tile's, the next tile's) + statistics table + the schedule's word
    auto kern = gemm_nt_pp_kernel<T, BM, TAPS, SUMS>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    const int tiles = cdiv_i(a->M, BM) * cdiv_i(a->N, 256);
    const int grid = tiles < pp_num_cus() ? tiles : pp_num_cus();
    // column panels (see tile_of): only when the weight matrix does not fit an XCD's L2 beside the activation stream and there are enough
    // column blocks to group; G column blocks of ~1.6 MB together.  THEIA_PP_PANEL=0: off (A/B), =G: forced width
    static int panel_env = -2;
    if (panel_env == -2) {
        const char* e = getenv("THEIA_PP_PANEL");
        panel_env = e == nullptr ? -1 : atoi(e);
    }
    const int tn = cdiv_i(a->N, 256);
    const long wbytes = (long)a->N * a->K * (long)sizeof(T), colblock = 256L * a->K * (long)sizeof(T);
    int panel = 0;
    if (panel_env > 0) panel = panel_env < tn ? panel_env : 0;
    else if (panel_env < 0 && tn >= 8 && wbytes > (3L << 20) && tiles > grid) {
        panel = (int)((1600L << 10) / colblock);
        if (panel < 1) panel = 1;
        if (panel >= tn) panel = 0;
    }
    // work-conserving schedule: any launch whose K gives the draw six half-tiles of lead
    const int nh = a->K / (64 / (int)sizeof(T));
    // ... on a device that exposes eight XCDs to a launch of at least eight workgroups: a workgroup only drains the queue of the XCD it
    // runs on, so an XCD that receives no workgroup of the launch would leave its queue's tiles unwritten (CPX / DPX / QPX partitions,
    // 4- or 6-XCD parts, grids below 8: the static schedule there).  A CU-masked stream that hides a whole XCD cannot be detected from
    // here: do not combine CU masks with theia_set_gemm_schedule(1).
    const bool want_dyn = (g_pp_dynamic_override >= 0 ? g_pp_dynamic_override : pp_dynamic_mode()) != 0 && nh >= 8 && grid >= 8 && pp_device_xccs() == 8;
    unsigned* sched = want_dyn ? pp_sched_block(stream) : nullptr;
    static int dephase_env = -2;  // -1: automatic
    if (dephase_env == -2) {
        const char* e = getenv("THEIA_PP_DEPHASE");
        dephase_env = e == nullptr ? -1 : atoi(e) > 0 ? atoi(e) : 0;
    }
    int dephase = dephase_env;
    if (dephase < 0) {  // ~2/3 of a tile's duration in units of 2048 cycles: ~1400 cycles per half k-tile of a 256-row tile + ~12k of tile switch and epilogue
        const long cyc = (long)nh * 1400 * BM / 256 + 12000;
        const int units = (int)(cyc * 2 / 3 / 2048);
        dephase = tiles > grid ? 131072 + (units < 1 ? 1 : units > 24 ? 24 : units) : 0;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, *a, tiles, panel, sched, dephase);
    THEIA_CHECK_LAUNCH("theia_gemm_nt(pp)");
    return THEIA_OK;
}

// Rows per tile for a launch the ping-pong kernel takes: 320 when that saves whole rounds of the persistent grid (cost of a
// round ~ rows per tile), else 256.  bf16 only; ln_sums needs a wave tile (160 rows) inside two images.
int theia_gemm_nt_pp_bm(const theia_gemm_args_t* a, int dtype) {
    if (a->tile == 320256) return 320;
    if (a->tile == 256256) return 256;
    if (dtype != THEIA_BF16 && dtype != THEIA_FP8) return 256;
    if (dtype == THEIA_FP8 && (a->ln_sums != nullptr || a->map.ntaps > 1)) return 256;  // (fp8: 320-row tiles for the plain matrices only)
    if (a->ln_sums != nullptr && a->map.rows_h * a->map.rows_w < 160) return 256;
    static int force = -1;
    if (force < 0) {
        const char* e = getenv("THEIA_PP_BM");
        force = e == nullptr ? 0 : atoi(e);
    }
    if (force == 256 || force == 320) return force;
    const int cus = pp_num_cus(), tn = cdiv_i(a->N, 256);
    const double c256 = (double)cdiv_i((long)cdiv_i(a->M, 256) * tn, cus) * 256.0;
    const double c320 = (double)cdiv_i((long)cdiv_i(a->M, 320) * tn, cus) * 320.0;
    // ties go to 256 rows -- except for launches that store two tensors per tile (GELU output + saved pre-activation): fewer, taller
    // tiles mean fewer chip-wide store bursts (fc1 forward, 4 rounds of 320 vs 5 of 256: 163 vs 182 us, r03_ab_tile_order_and_height.txt)
    return c320 < c256 || (c320 == c256 && a->aux_out != nullptr) ? 320 : 256;
}

int theia_gemm_nt_pp_launch(const theia_gemm_args_t* a, int dtype, hipStream_t stream) {
    // the single-tap instantiation clamps rows instead of reading zeros: only for maps whose one tap never leaves the input
    const theia_rowmap_t& mp = a->map;
    const bool taps = mp.ntaps > 1 || mp.dy[0] != 0 || mp.dx[0] != 0 || (mp.rows_h - 1) * mp.in_sy >= mp.in_h || (mp.rows_w - 1) * mp.in_sx >= mp.in_w;
    const bool sums = a->ln_sums != nullptr;
#ifdef PP_ONE  // compile-time experiments: exactly one instantiation, e.g. -DPP_ONE="bf16_t, 256, true, true"
    return pp_launch_one<PP_ONE>(a, stream);
#else
#ifndef PP_QUICK  // -DPP_QUICK: only the two bf16 single-tap instantiations
    if (dtype == THEIA_FP8) {  // (round 6: the single-tap instantiation -- clamped rows, no zero page, no per-tap state -- for the plain matrices)
        if (sums) return pp_launch_one<fp8_t, 256, true, true>(a, stream);
        if (taps) return pp_launch_one<fp8_t, 256, true, false>(a, stream);
        return theia_gemm_nt_pp_bm(a, dtype) == 320 ? pp_launch_one<fp8_t, 320, false, false>(a, stream) : pp_launch_one<fp8_t, 256, false, false>(a, stream);
    }
    if (dtype == THEIA_F32) return sums ? pp_launch_one<float, 256, true, true>(a, stream) : pp_launch_one<float, 256, true, false>(a, stream);
#endif
    const int bm = theia_gemm_nt_pp_bm(a, dtype);
    if (bm == 320) {
#ifndef PP_QUICK
        if (sums) return pp_launch_one<bf16_t, 320, true, true>(a, stream);
        if (taps) return pp_launch_one<bf16_t, 320, true, false>(a, stream);
#endif
        return pp_launch_one<bf16_t, 320, false, false>(a, stream);
    }
#ifndef PP_QUICK
    if (sums) return pp_launch_one<bf16_t, 256, true, true>(a, stream);
    if (taps) return pp_launch_one<bf16_t, 256, true, false>(a, stream);
#endif
    return pp_launch_one<bf16_t, 256, false, false>(a, stream);
#endif
}
