// Timing probe of the one-wave-per-SIMD NT kernel alone (no correctness): compile-time ablations -DW4_NO_DMA / -DW4_NO_READS / -DW4_NO_BARRIER
// tell where a half k-tile's cycles go.   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-D...] -I theia_amd/csrc tools/experiments/w4_probe.hip -o build/w4_probe
#include <stdarg.h>
#include <vector>
#include "gemm_w4.hip"
int theia_compute_cus() { return 256; }
void theia_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20;
    struct S { const char* name; int M, N, K, act; };
    const S shapes[] = {{"k3072 n768 1 round", 21760, 768, 3072, THEIA_ACT_NONE}, {"k768 n3072 4 rounds", 21760, 3072, 768, THEIA_ACT_NONE},
                        {"k768 n3072 gelu", 21760, 3072, 768, THEIA_ACT_GELU}, {"fc1 gelu", 25216, 3072, 768, THEIA_ACT_GELU}};
    bf16_t *a, *w, *o, *x; float* bias;
    hipMalloc(&a, (size_t)25216 * 3072 * 2); hipMalloc(&w, (size_t)3072 * 3072 * 2); hipMalloc(&o, (size_t)25216 * 3072 * 2); hipMalloc(&x, (size_t)25216 * 3072 * 2);
    hipMalloc(&bias, 4096 * 4);
    { std::vector<uint16_t> h((size_t)25216 * 3072); uint32_t r = 1; for (auto& v : h) { r = r * 1664525u + 1013904223u; v = (uint16_t)(0x3c00 + ((r >> 9) & 0x3ff) + ((r >> 31) << 15)); }
      hipMemcpy(a, h.data(), h.size() * 2, hipMemcpyHostToDevice); hipMemcpy(w, h.data(), (size_t)3072 * 3072 * 2, hipMemcpyHostToDevice); }
    hipMemset(bias, 0, 4096 * 4);
    for (const S& s : shapes) {
        theia_gemm_args_t g; memset(&g, 0, sizeof(g));
        g.a = a; g.w = w; g.out = o; g.M = s.M; g.N = s.N; g.K = s.K; g.ldw = s.K; g.ldo = s.N; g.bias = bias; g.act = s.act;
        g.map.ntaps = 1; g.map.rows_h = g.map.rows_w = g.map.in_h = g.map.in_w = g.map.out_w = 1; g.map.in_sy = g.map.in_sx = g.map.out_sy = g.map.out_sx = 1;
        g.map.in_c = s.K; g.map.in_batch_stride = s.K; g.map.out_batch_stride = s.N;
        if (s.act == THEIA_ACT_GELU) g.aux_out = x;
        for (int i = 0; i < 3; ++i) theia_gemm_nt_w4_launch(&g, THEIA_BF16, 0);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        for (int i = 0; i < iters; ++i) theia_gemm_nt_w4_launch(&g, THEIA_BF16, 0);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1000.0 / iters, tiles = (double)((s.M + 255) / 256) * (s.N / 256), rounds = ceil(tiles / 256.0);
#ifdef W4_TRACE
        {
            unsigned long long ph[4][16];
            hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_w4_phase), sizeof(ph));
            const int hunr = w4_hunr(s.act), nh = s.K / 32;
            const double unr = (double)(ph[0][4] - ph[0][3]) / hunr, rol = nh > hunr ? (double)(ph[0][5] - ph[0][4]) / (nh - hunr) : 0.0;
            const double tile_cyc = (double)(ph[0][6] - ph[0][3]);  // tile 0 start -> tile 1 start (when the workgroup has a second tile)
            printf("    block 0: %.0f cycles per unrolled half-tile (switch included), %.0f per rolled half-tile; entry -> tile 0 start %lld; tile 0 -> tile 1 %.0f; exit at %lld\n",
                   unr, rol, (long long)(ph[0][3] - ph[0][0]), rounds > 1 ? tile_cyc : 0.0, (long long)(ph[0][15] - ph[0][0]));
        }
#endif
        printf("%-22s %8.1f us %7.1f TF   %.0f tiles, %.0f rounds: %.2f us per tile-round = %.0f ns per half k-tile\n", s.name, us, 2.0 * s.M * s.N * s.K / us / 1e6, tiles, rounds,
               us / rounds, us / rounds / (s.K / 32) * 1000.0);
    }
    return 0;
}
