// Micro-benchmark: how fast can ONE CU store a tile from registers, and does it depend on how many other CUs store at the same time?
// (DESIGN §4 attributed the 14 B/clk/CU of the GEMM epilogues to a chip-wide HBM write burst; the alternative reading is a per-CU
// limit of the vector-store data path.)  Every block = 512 threads (8 waves) storing `per_wave` KiB per wave with dwordx4 stores in
// the epilogue's pattern (a wave instruction = 16 rows x 64 B, row pitch `ld` bytes) or fully contiguous (1 KiB per instruction).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/store_bench.hip -o build/store_bench && build/store_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

__global__ __launch_bounds__(512) void store_kernel(uint4* out, size_t block_stride_b, int ld_bytes, int n_inst, int pattern, int waves_active,
                                                    unsigned long long* cycles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= waves_active) return;
    char* base = reinterpret_cast<char*>(out) + (size_t)blockIdx.x * block_stride_b;
    uint4 v = make_uint4(threadIdx.x, blockIdx.x, lane, wave);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (pattern == 0) {  // epilogue pattern: lane (frow = lane & 15, fg = lane >> 4): row frow of a 16-row group, 16-byte piece fg of a 64-byte run
        char* p = base + (size_t)(wave * 16 + (lane & 15)) * ld_bytes + (lane >> 4) * 16;
        for (int i = 0; i < n_inst; ++i) {
            *reinterpret_cast<uint4*>(p + (size_t)(i >> 1) * 128 * ld_bytes + (i & 1) * 64) = v;
            v.x += 1;
        }
    } else if (pattern == 1) {  // contiguous: 64 lanes x 16 B = 1 KiB per instruction
        char* p = base + (size_t)wave * n_inst * 1024 + lane * 16;
        for (int i = 0; i < n_inst; ++i) {
            *reinterpret_cast<uint4*>(p + (size_t)i * 1024) = v;
            v.x += 1;
        }
    } else if (pattern == 3) {  // the same 8 full lines per instruction as pattern 2, but with the lanes of a row 8 / 16 apart (what a DPP
                                // exchange inside the MFMA accumulator layout gives): row = lane & 7, piece = ((lane >> 3) & 1) * 4 + (lane >> 4)
        char* p = base + (size_t)(wave * 8 + (lane & 7)) * ld_bytes + ((((lane >> 3) & 1) * 4 + (lane >> 4)) * 16);
        for (int i = 0; i < n_inst; ++i) {
            *reinterpret_cast<uint4*>(p + (size_t)i * 64 * ld_bytes) = v;
            v.x += 1;
        }
    } else if (pattern == 4) {  // half lines with ADJACENT lanes: row = lane >> 2 (16 rows), piece = lane & 3 (64 bytes per row)
        char* p = base + (size_t)(wave * 16 + (lane >> 2)) * ld_bytes + (lane & 3) * 16;
        for (int i = 0; i < n_inst; ++i) {
            *reinterpret_cast<uint4*>(p + (size_t)(i >> 1) * 128 * ld_bytes + (i & 1) * 64) = v;
            v.x += 1;
        }
    } else if (pattern == 5) {  // dwordx2, 16 adjacent lanes per row: 4 full lines per instruction (2 instructions per KiB)
        char* p = base + (size_t)(wave * 4 + (lane >> 4)) * ld_bytes + (lane & 15) * 8;
        uint2 w = make_uint2(v.x, v.y);
        for (int i = 0; i < 2 * n_inst; ++i) {
            *reinterpret_cast<uint2*>(p + (size_t)i * 32 * ld_bytes) = w;
            w.x += 1;
        }
    } else {  // full 128-byte lines: 8 lanes per row, 8 rows per instruction
        char* p = base + (size_t)(wave * 8 + (lane >> 3)) * ld_bytes + (lane & 7) * 16;
        for (int i = 0; i < n_inst; ++i) {
            *reinterpret_cast<uint4*>(p + (size_t)i * 64 * ld_bytes) = v;
            v.x += 1;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();  // all stores ISSUED (data has left the registers)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_readcyclecounter();  // all stores acknowledged
    if (lane == 0) {
        cycles[(blockIdx.x * 8 + wave) * 2] = t1 - t0;
        cycles[(blockIdx.x * 8 + wave) * 2 + 1] = t2 - t0;
    }
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const size_t cap = (size_t)1 << 30;
    uint4* out;
    unsigned long long* cyc;
    hipMalloc(&out, cap);
    hipMalloc(&cyc, 256 * 8 * 2 * 8);
    std::vector<unsigned long long> h(256 * 8 * 2);
    const int ld = 6144;  // row pitch of a [M, 3072] bf16 matrix
    printf("pattern 0 = 16 rows x 64 B per instruction, row = lane & 15 (GEMM epilogue), 2 = 8 rows x 128 B, 8 adjacent lanes per row, 3 = 8 rows x 128 B, lanes of a row 8/16 apart,\n4 = 16 rows x 64 B, 4 adjacent lanes per row, 5 = dwordx2, 4 rows x 128 B, 16 adjacent lanes per row\n");
    printf("%8s %6s %6s %8s | %10s %10s | %9s %9s\n", "pattern", "blocks", "waves", "KB/block", "issue cyc", "ack cyc", "B/clk iss", "B/clk ack");
    for (int pattern : {0, 2, 3, 4, 5})
        for (int waves : {8, 1})
            for (int blocks : {1, 256})
                for (int n_inst : {16, 32}) {
                    const size_t stride = (size_t)3 << 20;  // 3 MiB apart (regions of neighbouring blocks interleave, never the same bytes at pitch 6144)
                    for (int rep = 0; rep < 3; ++rep) {
                        hipLaunchKernelGGL(store_kernel, dim3(blocks), dim3(512), 0, 0, out, stride, ld, n_inst, pattern, waves, cyc);
                    }
                    hipDeviceSynchronize();
                    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
                    unsigned long long mi = 0, ma = 0;
                    for (int b = 0; b < blocks; ++b)
                        for (int w = 0; w < waves; ++w) {
                            mi = std::max(mi, h[(b * 8 + w) * 2]);
                            ma = std::max(ma, h[(b * 8 + w) * 2 + 1]);
                        }
                    const double kb = waves * n_inst * 1.0;
                    printf("%8d %6d %6d %8.0f | %10llu %10llu | %9.1f %9.1f\n", pattern, blocks, waves, kb, mi, ma, kb * 1024 / mi, kb * 1024 / ma);
                }
    return 0;
}
