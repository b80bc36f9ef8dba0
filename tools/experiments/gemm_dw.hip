// EXPERIMENT, NOT PART OF THE LIBRARY (round 4; measured and not kept: profiles/r04_dual_workgroup_kernel_not_kept.txt -- correct on every
// case of tools/pp_bench.hip, 10-40 % slower than gemm_pp.hip on every shape: two workgroups per CU cannot share an operand tile, the 1.5x
// L2->LDS bytes per flop and a ring that is only one half-tile ahead bound its main loop).  Build: hipcc ... -DWITH_DW tools/pp_bench.hip
//
// Persistent NT implicit GEMM, TWO INDEPENDENT WORKGROUPS PER CU ("dual-workgroup" kernel): BM x 256 tiles (BM = 128 or 160), 4 wave64
// per workgroup -- one per SIMD -- each owning all BM rows x 64 columns of the tile (the wave tile of gemm_pp.hip: 128 / 160 accumulator
// registers, FM + 4 fragment reads per 4 * FM MFMAs), and two such workgroups resident on every CU.
//
// Why (profiles/r03_pp_bench_tile_heights_and_phases.txt): gemm_pp.hip runs ONE 8-wave workgroup per CU whose two wave
// groups are locked one barrier apart.  Its main loop keeps the matrix pipe ~75 % busy, but at every tile boundary the whole CU leaves
// the loop together: epilogue (4.8-19 k cycles: all 256 CUs store their tiles in one burst that drains at the HBM write rate) + ~5 k to
// the next tile's first MFMA -- 15-35 % of a K = 768 tile with the matrix pipe idle, and the two wide-output launches of the MLP
// (fc1 forward with its saved pre-activation, fc2 data-gradient) at 630-730 TF.  Nothing inside one workgroup can overlap that:
// the tile's 128 accumulator registers are the data being stored, and a second set does not fit beside them.
//
// Here the two waves of a SIMD belong to DIFFERENT workgroups with their own tiles, LDS rings and barriers, so the hardware issues
// one workgroup's MFMAs while the other reads fragments, waits for operands -- or runs its epilogue: a workgroup's store burst and
// tile switch overlap the partner's main loop.  The ping-pong alternation is no longer scheduled by barriers, it falls out of
// in-order issue on a shared matrix pipe (MI355X_MICROARCH: a partner's MFMAs come straight out of your stream; an MFMA-only and a
// VALU / memory wave on one SIMD run concurrently).  To keep the two workgroups of a CU from running in phase (both in the loop, then
// both in the epilogue) one of them -- the one whose LDS allocation starts at 0 -- runs at a higher wave priority: it proceeds as if
// alone and its partner fills the gaps, which de-phases them after the first tile.
//
// Costs, accepted knowingly: (BM + 256) rows of operands per BM x 256 outputs = 85-98 flop per L2->LDS byte instead of 128-142
// (two workgroups do not share an operand tile), a 3-deep ring of half k-tiles instead of 4 (2 x 3 x (BM + 256) x 64 B + bias row
// = 148-160 KB of the CU's 160 KB), one workgroup barrier per half k-tile.
//
// Everything else is gemm_pp.hip's design and shares its code: LDS-DMA operand ring with a source-side XOR swizzle and counted vmcnt
// waits, ONE continuous prefetch stream across a workgroup's tiles, bias / residual rows initialising the accumulators, the LDS-free
// epilogue in the accumulator layout with permuted weight rows (gemm_epi_direct.h), LayerNorm statistics through an LDS table.
//
// Hazards (one wave group, barrier b(g) closes R(g)):
//   RAW  half-tile g is read in R(g); every wave waited for its own pieces of it at the end of R(g-1) (vmcnt <= pieces of g+1), before b(g-1).
//   WAR  R(g) refills slot (g+2) % 3 = (g-1) % 3, last read in R(g-1) by every wave before b(g-1) (lgkmcnt(0) precedes the barrier).
//   Tile boundary: the epilogue's stores stay in flight into the next loop; loads retire in order among loads, so a counted wait is only
//   made stricter by them (see gemm_pp.hip).  The next tile's first two half-tiles were requested by the last two iterations.
#include <type_traits>
#include "../../theia_amd/csrc/gemm_epi_direct.h"  // (the library's epilogue: accumulator lane order)

__device__ uint4 g_dw_zero_page[1024 + 1];

#ifdef PP_TRACE
__device__ unsigned long long g_dw_phase[8][16];
__device__ unsigned int g_dw_ids[512][2];
#define DW_PHASE(k) \
    if (blockIdx.x == DW_TRACE_BLOCK && (threadIdx.x & 63) == 0 && (k) < 16) g_dw_phase[threadIdx.x >> 6][k] = __builtin_readcyclecounter();
#ifndef DW_TRACE_BLOCK
#define DW_TRACE_BLOCK 0
#endif
#else
#define DW_PHASE(k)
#endif

__device__ __forceinline__ int dw_f(int row) { return (4 - ((row >> 2) & 3)) & 3; }
template <int N> __device__ __forceinline__ void dw_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// T: operand type (bf16).  FM: 16-row fragments per tile (8: 128 rows, 10: 160 rows).  TAPS: multi-tap row maps.  SUMS: ln_sums epilogue.
// prio_mode: 0 = both workgroups of a CU alike (priority 1 inside the MFMA segments only); 1 = the workgroup whose LDS allocation starts
// at 0 runs two priority levels above its partner.
template <typename T, int FM, bool TAPS, bool SUMS>
__global__ __launch_bounds__(256, 2) void gemm_nt_dw_kernel(const theia_gemm_args_t p, const int ntiles, const int panel, const int prio_mode) {
    constexpr int BM = FM * 16, BN = 256;
    constexpr int NSTAGE = 3;
    constexpr int HKT = 64 / (int)sizeof(T);
    constexpr int EPC = 16 / (int)sizeof(T);
    constexpr int FN = 4;
    constexpr int SRP = 64;                        // rows staged per pass (256 threads x 16 B = 64 rows of 64 B)
    constexpr int NPA = (BM + SRP - 1) / SRP;      // A passes; the last one covers BM % 64 rows when BM is not a multiple of 64
    constexpr int A_TAIL_WAVES = (BM % SRP) / 16;  // waves that take part in the partial A pass (0 = every pass is full)
    constexpr int NPB = BN / SRP;
    constexpr int LPH_FULL = NPA + NPB;
    constexpr int LPH_PART = LPH_FULL - (A_TAIL_WAVES ? 1 : 0);
    constexpr int STAGE = (BM + BN) * 64;
    constexpr bool SCALE = false;
    constexpr bool BIAS_IN_ACC = true;
    using OutT = T;
    static_assert(sizeof(T) == 2, "bf16 operands only");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int uwave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wn = uwave;
    const bool tailw = A_TAIL_WAVES == 0 || uwave < A_TAIL_WAVES;
    const theia_rowmap_t& mp = p.map;
    const int tiles_n = (p.N + BN - 1) / BN;
    DW_PHASE(0)
    // role of this workgroup on its CU: LDS_BASE of HW_REG_LDS_ALLOC (id 6, bits 0..11 here: base in allocation granules) is 0 for the
    // workgroup that got the CU's LDS first.  Wave priorities: lead 2 (3 inside MFMA segments), partner 0 (1).
    int prio_base = 0;
    if (prio_mode == 1) {
        const unsigned lds_alloc = __builtin_amdgcn_s_getreg((12 - 1) << 11 | 0 << 6 | 6);
        prio_base = lds_alloc == 0 ? 2 : 0;
    }
#ifdef PP_TRACE
    if (threadIdx.x == 0 && blockIdx.x < 512) {
        g_dw_ids[blockIdx.x][0] = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);   // HW_REG_HW_ID
        g_dw_ids[blockIdx.x][1] = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 6);   // HW_REG_LDS_ALLOC
    }
#endif
    auto setprio_lo = [&]() {
        if (prio_base == 0) __builtin_amdgcn_s_setprio(0);
        else __builtin_amdgcn_s_setprio(2);
    };
    auto setprio_hi = [&]() {
        if (prio_base == 0) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(3);
    };
    setprio_lo();

    // tiles of this workgroup: round r of the grid takes tiles [r * grid, r * grid + cnt), spread over the XCDs like a launch of cnt
    const int grid = gridDim.x, bid = blockIdx.x;
    const int rounds = (ntiles + grid - 1) / grid;
    const int cnt_last = ntiles - (rounds - 1) * grid;
    const int my_tiles = rounds - 1 + (bid < cnt_last ? 1 : 0);
    auto tile_of = [&](int r) {  // schedule position -> tile; column panels as in gemm_pp.hip
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
    const T* __restrict__ A = reinterpret_cast<const T*>(p.a);
    const T* __restrict__ W = reinterpret_cast<const T*>(p.w);
    const int R = mp.rows_h * mp.rows_w;
    const int nh = (p.K + HKT - 1) / HKT;   // host guarantees K % HKT == 0 for this kernel
    const int hpt = mp.in_c / HKT;          // half-tiles per tap
    const uint64_t zp = reinterpret_cast<uint64_t>(g_dw_zero_page);

    // ---------------------------------------------------------------- operand prefetch stream (see gemm_pp.hip)
    uint64_t src_ptr[LPH_FULL];
    int pf_r = 0, pf_h = 0, cur_tap = 0, next_tap_h = hpt, pf_m0 = 0, pf_n0 = 0;
    auto set_tap = [&](int tap) {
        int tid_ = threadIdx.x, R_ = R, rows_w_ = mp.rows_w;
        asm volatile("" : "+v"(tid_));
        asm volatile("" : "+s"(R_), "+s"(rows_w_));
        const int st_chunk = tid_ & 3, st_row = tid_ >> 2;
        const int lchunk = st_chunk ^ dw_f(st_row);
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
            if (pf_r + 1 < my_tiles) {
                ++pf_r;
                pf_h = 0;
                setup_tile(tile_of(pf_r));
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
    // wait until at most ONE half-tile's worth of this wave's pieces is outstanding
    auto wait_halves1 = [&]() {
        if (A_TAIL_WAVES != 0 && !tailw) dw_wait_vm<LPH_PART>();
        else dw_wait_vm<LPH_FULL>();
    };

    // ---------------------------------------------------------------- accumulator initialisation: bias (+ residual) rows
    const OutT* __restrict__ RES = reinterpret_cast<const OutT*>(p.resid);
    const bool resid_init = RES != nullptr && p.act == THEIA_ACT_NONE;
    gt_f32x4 acc[FN][FM];
    auto start_tile = [&](auto FIRST_C, int tile_id, int m_wave0, int n_wave0) {
        constexpr bool FIRST = decltype(FIRST_C)::value;
        const int frow = threadIdx.x & 15, fg = (threadIdx.x >> 4) & 3;
        gt_u32x4 brow[2][2];
        const bool has_bias = p.bias != nullptr;
        constexpr bool EARLY_ROWS = FIRST && FM == 8;  // (the 160-row instantiation keeps request, wait and use adjacent: see gemm_pp.hip)
        auto row_load = [&](auto& dst, const void* ptr) {
            if constexpr (EARLY_ROWS) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory");
            else dst = __builtin_bit_cast(typename std::remove_reference<decltype(dst)>::type, *reinterpret_cast<const gt_u32x4*>(ptr));
        };
        auto request_rows = [&]() {
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
                        row_load(acc[2 * t][j], RES + (lv ? c.off + n : off_dead));
                    }
                    rw.next(c);
                }
            }
        };
        if constexpr (EARLY_ROWS) request_rows();
        if constexpr (FIRST) {
            setup_tile(tile_id);
            DW_PHASE(1)
#pragma unroll
            for (int s = 0; s < NSTAGE - 1; ++s) {
                pf_issue(s);
                pf_advance();
            }
            DW_PHASE(2)
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (EARLY_ROWS) {  // loads retire in order: the rows have landed once only the prologue's pieces are outstanding
            if (A_TAIL_WAVES != 0 && !tailw) dw_wait_vm<(NSTAGE - 1) * LPH_PART>();
            else dw_wait_vm<(NSTAGE - 1) * LPH_FULL>();
        } else {
            request_rows();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
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
                    asm volatile("" : "+v"(acc[2 * t][j]));
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
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j) asm volatile("" : "+v"(acc[i][j]));
    };

    // ---------------------------------------------------------------- first tile: init rows, then the operand prologue
    int tile = tile_of(0);
    int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    start_tile(std::true_type{}, tile, m0, n0 + wn * 64);
    __builtin_amdgcn_sched_barrier(0);
    wait_halves1();  // half-tile 0 landed

    gt_u32x4 fb[FN], fa[FM];
    int slot = 0;  // ring slot of the half-tile the next R segment reads (= global half-tile counter % 3)

    constexpr int BIAS_LDS = NSTAGE * STAGE;  // one bias row (1 KiB), then the statistics table
    const bool seamless = !resid_init && nh >= NSTAGE - 1;
    unsigned long long* const sums_tab = reinterpret_cast<unsigned long long*>(smem + BIAS_LDS + 1024);
    int flush_img0 = -1;
    if constexpr (SUMS) {
        if (threadIdx.x < 2 * GT_SUMS_SLOTS) sums_tab[threadIdx.x] = 0ull;
    }
    for (int r = 0; r < my_tiles; ++r) {
        const uint32_t smem_base = gt_lds_addr(smem);
        const int frow_ = threadIdx.x & 15, fg_ = (threadIdx.x >> 4) & 3;
        const uint32_t lane_a = smem_base + frow_ * 64 + ((fg_ ^ dw_f(frow_)) << 4);
        const uint32_t lane_b = smem_base + BM * 64 + (wn * 64 + frow_) * 64 + ((fg_ ^ dw_f(frow_)) << 4);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (SUMS) {
            if (uwave == 0 && flush_img0 >= 0)
                gt_flush_sums(sums_tab, reinterpret_cast<unsigned long long*>(p.ln_sums), flush_img0, p.map.rows_h * p.map.rows_w, p.M, threadIdx.x & 63);
        }
        DW_PHASE(3 + 4 * r)
        const int tile_next = r + 1 < my_tiles ? tile_of(r + 1) : tile;
        const int m0_next = (tile_next / tiles_n) * BM, n0_next = (tile_next % tiles_n) * BN;
        const bool fetch_bias = seamless && p.bias != nullptr && uwave == 0 && r + 1 < my_tiles;  // wave-uniform
        for (int h = 0; h < nh; ++h) {
            const uint32_t soff = (uint32_t)slot * STAGE;
            const int pslot = slot == 0 ? NSTAGE - 1 : slot - 1;  // (g + 2) % 3: the slot read in the previous iteration
            if (fetch_bias && h == nh - (NSTAGE - 1)) {  // two iterations before the tile ends: covered by the loop's own waits
                const int n = min(n0_next + (int)(threadIdx.x & 63) * 4, p.N - 4);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.bias + n),
                                                 (__attribute__((address_space(3))) void*)(smem + BIAS_LDS), 16, 0, 0);
            }
            // ---------------- R(g): FN + FM fragment reads + the LDS-DMA pieces of half-tile g+2
            const uint32_t ab = lane_b + soff, aa = lane_a + soff;
            gd_static_for<0, FN>([&](auto I) { gt_ds_read128<decltype(I)::value * 1024>(fb[decltype(I)::value], ab); });
            gd_static_for<0, FM>([&](auto J) { gt_ds_read128<decltype(J)::value * 1024>(fa[decltype(J)::value], aa); });
            pf_issue(pslot);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < FN; ++i) asm volatile("" : "+v"(fb[i]));
#pragma unroll
            for (int j = 0; j < FM; ++j) asm volatile("" : "+v"(fa[j]));
            wait_halves1();  // half-tile g+1 landed; g+2 may be in flight
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---------------- M(g): FM * FN MFMAs
            setprio_hi();
#pragma unroll
            for (int j = 0; j < FM; ++j) {
#pragma unroll
                for (int i = 0; i < FN; ++i) gt_mma<T>(acc[i][j], fb[i], fa[j]);
            }
            setprio_lo();
            __builtin_amdgcn_sched_barrier(0);
            pf_advance();
            __builtin_amdgcn_sched_barrier(0);
            slot = slot == NSTAGE - 1 ? 0 : slot + 1;
        }
        DW_PHASE(4 + 4 * r)
        {
            const gd_rows_t rw(p);
            const int it0 = gd_epilogue<OutT, FM, SUMS, SCALE, BIAS_IN_ACC>(acc, p, rw, m0, n0 + wn * 64, threadIdx.x & 63, resid_init, sums_tab, m0);
            if constexpr (SUMS) flush_img0 = it0;
        }
        DW_PHASE(5 + 4 * r)
        if (r + 1 < my_tiles) {
            tile = tile_next;
            m0 = m0_next;
            n0 = n0_next;
            if (!seamless) {
                start_tile(std::false_type{}, tile, m0, n0 + wn * 64);  // rows requested now, queue drained
            } else {  // accumulators = the bias row in LDS (landed and fenced two iterations before the previous tile ended)
                const int fg_b = (threadIdx.x >> 4) & 3;
                float b8[2][8];
                if (p.bias != nullptr) {
                    gt_u32x4 bl[2][2];
                    const uint32_t ba = gt_lds_addr(smem) + BIAS_LDS + (wn * 64 + fg_b * 8) * 4;
                    gt_ds_read128<0>(bl[0][0], ba);
                    gt_ds_read128<16>(bl[0][1], ba);
                    gt_ds_read128<128>(bl[1][0], ba);
                    gt_ds_read128<144>(bl[1][1], ba);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        asm volatile("" : "+v"(bl[t][0]));
                        asm volatile("" : "+v"(bl[t][1]));
                        const bool ok = (n0 + wn * 64 + t * 32 + fg_b * 8) < p.N;
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
            }
        }
        DW_PHASE(6 + 4 * r)
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
int g_dw_grid_cap = 0;  // > 0: cap on the persistent grid (tools/dw_bench.hip: forces several tiles per workgroup on small problems)
static int dw_slots() {
    static int forced = -1;
    if (forced < 0) {
        const char* e = getenv("THEIA_DW_GRID");
        forced = e != nullptr && atoi(e) > 0 ? atoi(e) : 0;
    }
    return g_dw_grid_cap > 0 ? g_dw_grid_cap : forced > 0 ? forced : 2 * theia_compute_cus();
}

template <typename T, int FM, bool TAPS, bool SUMS>
static int dw_launch_one(const theia_gemm_args_t* a, hipStream_t stream) {
    constexpr int BM = FM * 16;
    constexpr int lds = 3 * (BM + 256) * 64 + 1024 + 64;  // operand ring + the next tile's bias row + statistics table
    static_assert(2 * lds <= 160 * 1024, "two workgroups per CU");
    auto kern = gemm_nt_dw_kernel<T, FM, TAPS, SUMS>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    static int prio_mode = -1, panel_env = -2;
    if (prio_mode < 0) {
        const char* e = getenv("THEIA_DW_PRIO");
        prio_mode = e == nullptr ? 1 : atoi(e);
        e = getenv("THEIA_PP_PANEL");
        panel_env = e == nullptr ? -1 : atoi(e);
    }
    const int tiles = cdiv_i(a->M, BM) * cdiv_i(a->N, 256);
    const int grid = tiles < dw_slots() ? tiles : dw_slots();
    const int tn = cdiv_i(a->N, 256);
    const long wbytes = (long)a->N * a->K * (long)sizeof(T), colblock = 256L * a->K * (long)sizeof(T);
    int panel = 0;
    if (panel_env > 0) panel = panel_env < tn ? panel_env : 0;
    else if (panel_env < 0 && tn >= 8 && wbytes > (3L << 20) && tiles > grid) {
        panel = (int)((1600L << 10) / colblock);
        if (panel < 1) panel = 1;
        if (panel >= tn) panel = 0;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, *a, tiles, panel, prio_mode);
    THEIA_CHECK_LAUNCH("theia_gemm_nt(dw)");
    return THEIA_OK;
}

// Rows per tile: 160 when that saves whole rounds of the 2-workgroups-per-CU grid, else 128 (same rule as theia_gemm_nt_pp_bm)
int theia_gemm_nt_dw_bm(const theia_gemm_args_t* a, int dtype) {
    if (a->tile == 160256) return 160;
    if (a->tile == 128256) return 128;
    if (a->ln_sums != nullptr && a->map.rows_h * a->map.rows_w < 160) return 128;
    static int force = -1;
    if (force < 0) {
        const char* e = getenv("THEIA_DW_BM");
        force = e == nullptr ? 0 : atoi(e);
    }
    if (force == 128 || force == 160) return force;
    const int slots = dw_slots(), tn = cdiv_i(a->N, 256);
    const double c128 = (double)cdiv_i((long)cdiv_i(a->M, 128) * tn, slots) * 128.0;
    const double c160 = (double)cdiv_i((long)cdiv_i(a->M, 160) * tn, slots) * 160.0;
    return c160 < c128 ? 160 : 128;
}

int theia_gemm_nt_dw_launch(const theia_gemm_args_t* a, int dtype, hipStream_t stream) {
    if (dtype != THEIA_BF16) {
        theia_set_error("theia_gemm_nt(dw): bf16 operands only");
        return THEIA_ERR_UNSUPPORTED;
    }
    const theia_rowmap_t& mp = a->map;
    const bool taps = mp.ntaps > 1 || mp.dy[0] != 0 || mp.dx[0] != 0 || (mp.rows_h - 1) * mp.in_sy >= mp.in_h || (mp.rows_w - 1) * mp.in_sx >= mp.in_w;
    const bool sums = a->ln_sums != nullptr;
#ifdef DW_ONE
    return dw_launch_one<DW_ONE>(a, stream);
#else
    const int bm = theia_gemm_nt_dw_bm(a, dtype);
    if (bm == 160) {
#ifndef DW_QUICK
        if (sums) return dw_launch_one<bf16_t, 10, true, true>(a, stream);
        if (taps) return dw_launch_one<bf16_t, 10, true, false>(a, stream);
#endif
        return dw_launch_one<bf16_t, 10, false, false>(a, stream);
    }
#ifndef DW_QUICK
    if (sums) return dw_launch_one<bf16_t, 8, true, true>(a, stream);
    if (taps) return dw_launch_one<bf16_t, 8, true, false>(a, stream);
#endif
    return dw_launch_one<bf16_t, 8, false, false>(a, stream);
#endif
}
