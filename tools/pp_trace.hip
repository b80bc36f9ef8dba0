// Standalone cycle-trace harness for the ping-pong GEMM kernel (not part of libtheia_hip.so):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPP_TRACE -DPP_TRACE_H0=40 tools/pp_trace.hip -o build/pp_trace && build/pp_trace
// Prints, for block 0 and every wave, the s_memtime stamps at: R start, ds_reads done, vmcnt done, M start (after the
// barrier), M end -- for four consecutive half-tiles.
#include <stdarg.h>
#include <vector>
#include "../theia_amd/csrc/gemm_pp.hip"

void theia_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }

int main(int argc, char** argv) {
    // pp_trace [M N K [act [resid]]] : defaults to the conv16-sized problem; act 2 = GELU (+ aux_out), resid 1 = residual add
    const int M = argc > 3 ? atoi(argv[1]) : 32768, N = argc > 3 ? atoi(argv[2]) : 768, K = argc > 3 ? atoi(argv[3]) : 6912;
    const int act = argc > 4 ? atoi(argv[4]) : 0, resid = argc > 5 ? atoi(argv[5]) : 0;
    bf16_t *a, *w, *o;
    hipMalloc(&a, (size_t)M * K * 2); hipMalloc(&w, (size_t)N * K * 2); hipMalloc(&o, (size_t)M * N * 2);
    std::vector<uint16_t> ha((size_t)M * 64), hw((size_t)N * K);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0x3c00 + (i * 2654435761u >> 20 & 0x1ff);
    hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    for (size_t r = 0; r < (size_t)M * K; r += hw.size()) hipMemcpy(a + r, hw.data(), std::min(hw.size(), (size_t)M * K - r) * 2, hipMemcpyHostToDevice);
    theia_gemm_args_t g; memset(&g, 0, sizeof(g));
    g.a = a; g.w = w; g.out = o; g.M = M; g.N = N; g.K = K; g.ldw = K; g.ldo = N;
    g.map.ntaps = 1; g.map.rows_h = g.map.rows_w = g.map.in_h = g.map.in_w = g.map.out_w = 1; g.map.in_sy = g.map.in_sx = 1;
    g.map.out_sy = g.map.out_sx = 1; g.map.in_c = K; g.map.in_batch_stride = K; g.map.out_batch_stride = N;
    bf16_t *aux = nullptr, *res = nullptr; float* bias = nullptr;
    hipMalloc(&aux, (size_t)M * N * 2); hipMalloc(&res, (size_t)M * N * 2); hipMalloc(&bias, N * 4);
    hipMemset(res, 0, (size_t)M * N * 2); hipMemset(bias, 0, N * 4);
    g.bias = bias; g.act = act;
    if (act == THEIA_ACT_GELU) g.aux_out = aux;
    if (resid) g.resid = res;
    for (int it = 0; it < 3; ++it) theia_gemm_nt_pp_launch(&g, THEIA_BF16, 0);
    hipDeviceSynchronize();
#ifdef PP_TRACE
    unsigned long long t[8][4][2][5];
    hipMemcpyFromSymbol(t, HIP_SYMBOL(g_pp_trace), sizeof(t));
    const unsigned long long t0 = t[0][0][0][0];
    // per half-tile: R start, fragment reads issued, LDS-DMA issued, fragments landed, own pieces landed, M start, M end
    for (int wv = 0; wv < 8; ++wv) {
        printf("wave %d:", wv);
        for (int h = 0; h < 4; ++h) {
            const long long v[7] = {(long long)(t[wv][h][0][0] - t0), (long long)(t[wv][h][1][0] - t0), (long long)(t[wv][h][1][1] - t0),
                                    (long long)(t[wv][h][0][1] - t0), (long long)(t[wv][h][0][2] - t0), (long long)(t[wv][h][0][3] - t0),
                                    (long long)(t[wv][h][0][4] - t0)};
            printf(" |");
            for (int k = 0; k < 7; ++k) printf(" %5lld", v[k]);
        }
        printf("\n");
    }
    unsigned long long et[8][16];
    hipMemcpyFromSymbol(et, HIP_SYMBOL(g_ep_trace), sizeof(et));
    printf("epilogue of block 0 (entry, bias/prefetch landed, then per group: start, staged, passes issued; end), cycles since the wave's entry\n");
    for (int wv = 0; wv < 8; wv += 4) {
        printf("wave %d:", wv);
        for (int k = 1; k < 15; ++k) printf(" %6lld", (long long)(et[wv][k] - et[wv][0]));
        printf("\n");
    }
    unsigned long long ph[2][8][7];
    hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_pp_phase), sizeof(ph));
    printf("phases (entry, addr setup done, prologue issued, loop start, loop end, epilogue issued, stores acknowledged), cycles since entry of block 0 wave 0\n");
    for (int b = 0; b < 2; ++b)
        for (int wv = 0; wv < 8; wv += 4) {
            printf("block %3d wave %d:", b * 256, wv);
            for (int k = 0; k < 7; ++k) printf(" %8lld", (long long)(ph[b][wv][k] - ph[0][0][0]));
            printf("\n");
        }
#endif
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int it = 0; it < 20; ++it) theia_gemm_nt_pp_launch(&g, THEIA_BF16, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("avg %.1f us per launch\n", ms * 1000 / 20);
    return 0;
}
