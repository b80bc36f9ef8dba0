// Does gfx950 execute scalar-memory atomics (s_atomic_add ... glc), and are per-XCD counters consistent when every workgroup picks the
// counter of the XCD it runs on (HW_REG_XCC_ID)?  Each workgroup draws NDRAW tickets; the host checks that the tickets of every counter
// are a permutation of 0 .. total-1.   hipcc --offload-arch=gfx950 -O3 tools/experiments/satomic_probe.hip -o build/satomic_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
constexpr int NDRAW = 64;
__global__ void probe(unsigned* counters /* 8 x 16 dwords */, unsigned* tickets /* [grid][NDRAW] */, unsigned* xcc_of, unsigned long long* cyc, int pause) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    unsigned* c = counters + xcc * 16;
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) != 0) return;  // a scalar instruction executes once per WAVE: one wave draws
    unsigned long long spent = 0;
    for (int i = 0; i < NDRAW; ++i) {
        for (int k = 0; k < pause; ++k) __builtin_amdgcn_s_sleep(127);  // (pause > 0: draws far apart, as a tile scheduler would issue them)
        const unsigned long long t0 = __builtin_readcyclecounter();
        unsigned one = 1u, ret;
        // returns the pre-op value in the data register (glc)
        asm volatile("s_mov_b32 %0, %2\n\ts_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=&s"(ret) : "s"(c), "s"(one) : "memory");
        spent += __builtin_readcyclecounter() - t0;
        if (threadIdx.x == 0) tickets[blockIdx.x * NDRAW + i] = ret;
    }
    if (threadIdx.x == 0) { xcc_of[blockIdx.x] = xcc; cyc[blockIdx.x] = spent; }
}
int main() {
    const int grid = 256;
    unsigned *counters, *tickets, *xcc_of; unsigned long long* cyc;
    hipMalloc(&counters, 8 * 16 * 4); hipMalloc(&tickets, grid * NDRAW * 4); hipMalloc(&xcc_of, grid * 4); hipMalloc(&cyc, grid * 8);
    int bad_total = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(counters, 0, 8 * 16 * 4);
        hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 0, 0, counters, tickets, xcc_of, cyc, rep == 0 ? 0 : rep * 8);
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf("launch/sync failed: %s\n", hipGetErrorString(e)); return 2; }
        std::vector<unsigned> ht(grid * NDRAW), hx(grid), hc(8 * 16); std::vector<unsigned long long> hy(grid);
        hipMemcpy(ht.data(), tickets, ht.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(hx.data(), xcc_of, hx.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(hc.data(), counters, hc.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(hy.data(), cyc, hy.size() * 8, hipMemcpyDeviceToHost);
        int bad = 0, mism = 0; double cy = 0;
        for (int x = 0; x < 8; ++x) {
            std::vector<unsigned> t;
            int nb = 0;
            for (int b = 0; b < grid; ++b) if (hx[b] == (unsigned)x) { ++nb; if ((b & 7) != x) ++mism; for (int i = 0; i < NDRAW; ++i) t.push_back(ht[b * NDRAW + i]); }
            std::sort(t.begin(), t.end());
            int ok = 1;
            for (size_t i = 0; i < t.size(); ++i) if (t[i] != i) { ok = 0; break; }
            if (hc[x * 16] != t.size()) ok = 0;
            if (!ok) ++bad;
            printf("  rep %d xcc %d: %d blocks, %zu tickets, final counter %u: %s\n", rep, x, nb, t.size(), hc[x * 16], ok ? "permutation ok" : "BROKEN");
        }
        for (int b = 0; b < grid; ++b) cy += (double)hy[b];
        printf("rep %d: %d broken counters; blocks whose XCC_ID != blockIdx %% 8: %d; mean cycles per drawn ticket (incl. wait): %.0f\n", rep, bad, mism, cy / grid / NDRAW);
        bad_total += bad;
    }
    printf("%s\n", bad_total ? "FAIL" : "PASS");
    return bad_total ? 1 : 0;
}
