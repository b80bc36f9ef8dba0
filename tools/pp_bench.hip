// Standalone harness for the persistent ping-pong GEMM kernel (not part of libtheia_hip.so): correctness against a naive device
// GEMM on every epilogue flavour + ragged shapes, then timing of the hot-path shapes per tile height / grid mode.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pp_bench.hip -o build/pp_bench && build/pp_bench [check|time|all] [iters]
// With -DPP_TRACE it also prints block 0's cycle stamps (entry, prologue, loop start/end, epilogue end ... per tile).
#include <stdarg.h>
#include <math.h>
#include <vector>
#include <string>
#include "../theia_amd/csrc/gemm_pp.hip"
#ifdef WITH_W4  // the one-wave-per-SIMD experiment (tools/experiments/gemm_w4_not_kept.patch: git apply it first; tile 256004): not kept, not in the library
#include "experiments/gemm_w4.hip"
#endif
#ifdef WITH_DW  // the dual-workgroup experiment (tools/experiments/gemm_dw_not_kept.patch: git apply it first): not kept, not in the library
#include "experiments/gemm_dw.hip"
#endif

// tile request -> kernel: 256256 / 320256 = the 8-wave ping-pong kernel, 128256 / 160256 = the dual-workgroup kernel
static int launch_any(const theia_gemm_args_t* g, hipStream_t s) {
#ifdef WITH_W4
    if (g->tile == 256004) return theia_gemm_nt_w4_launch(g, THEIA_BF16, s);
#endif
#ifdef WITH_DW
    if (g->tile == 128256 || g->tile == 160256) return theia_gemm_nt_dw_launch(g, THEIA_BF16, s);
#endif
    return theia_gemm_nt_pp_launch(g, THEIA_BF16, s);
}
#ifndef WITH_DW
static int g_dw_grid_cap = 0;
static int theia_gemm_nt_dw_bm(const theia_gemm_args_t*, int) { return 0; }
#endif

int theia_compute_cus() {  // (misc.hip in the library)
    int v = 0;
    return hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, 0) == hipSuccess && v > 0 ? v : 256;
}
void theia_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }

static uint32_t rng_state = 12345u;
static inline float frand() { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xffff) / 32768.0f - 1.0f; }
static inline uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }

__device__ __forceinline__ float ref_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float ref_gelu_grad(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * expf(-0.5f * x * x);
}
// naive reference: one thread per output element (plain row-major maps only)
__global__ void ref_gemm(const bf16_t* a, const bf16_t* w, const float* bias, const bf16_t* resid, const bf16_t* aux_in, float* out,
                         float* pre_out, int M, int N, int K, int act) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N || m >= M) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += bf16_to_f32(a[(size_t)m * K + k]) * bf16_to_f32(w[(size_t)n * K + k]);
    if (bias) s += bias[n];
    if (pre_out) pre_out[(size_t)m * N + n] = s;
    if (act == THEIA_ACT_GELU) s = ref_gelu(s);
    else if (act == THEIA_ACT_RELU) s = fmaxf(s, 0.f);
    else if (act == THEIA_ACT_MUL_DGELU) s *= ref_gelu_grad(bf16_to_f32(aux_in[(size_t)m * N + n]));
    else if (act == THEIA_ACT_MUL_DRELU) s = bf16_to_f32(aux_in[(size_t)m * N + n]) > 0.f ? s : 0.f;
    if (resid) s += bf16_to_f32(resid[(size_t)m * N + n]);
    out[(size_t)m * N + n] = s;
}

struct Bufs {
    bf16_t *a, *w, *o, *res, *aux_in, *aux_out;
    float *bias, *ref, *ref_pre;
    size_t cap_a, cap_w, cap_o;
};

static void fill_bf16(bf16_t* d, size_t n, float scale) {
    std::vector<uint16_t> h(std::min(n, (size_t)1 << 22));
    for (auto& v : h) v = f2bf(frand() * scale);
    for (size_t o = 0; o < n; o += h.size()) hipMemcpy(d + o, h.data(), std::min(h.size(), n - o) * 2, hipMemcpyHostToDevice);
}

static theia_gemm_args_t make_args(const Bufs& b, int M, int N, int K, int act, bool bias, bool resid, int tile) {
    theia_gemm_args_t g;
    memset(&g, 0, sizeof(g));
    g.a = b.a; g.w = b.w; g.out = b.o; g.M = M; g.N = N; g.K = K; g.ldw = K; g.ldo = N;
    g.map.ntaps = 1; g.map.rows_h = g.map.rows_w = g.map.in_h = g.map.in_w = g.map.out_w = 1; g.map.in_sy = g.map.in_sx = 1;
    g.map.out_sy = g.map.out_sx = 1; g.map.in_c = K; g.map.in_batch_stride = K; g.map.out_batch_stride = N;
    g.act = act;
    g.tile = tile;
    if (bias) g.bias = b.bias;
    if (resid) g.resid = b.res;
    if (act == THEIA_ACT_GELU) g.aux_out = b.aux_out;
    if (act == THEIA_ACT_MUL_DGELU || act == THEIA_ACT_MUL_DRELU) g.aux_in = b.aux_in;
    return g;
}

static int check_one(const Bufs& b, int M, int N, int K, int act, bool bias, bool resid, int tile) {
    theia_gemm_args_t g = make_args(b, M, N, K, act, bias, resid, tile);
    hipMemset(b.o, 0xff, (size_t)M * N * 2);
    hipMemset(b.aux_out, 0xff, (size_t)M * N * 2);
    int rc = launch_any(&g, 0);
    if (rc) { printf("launch rc=%d\n", rc); return 1; }
    ref_gemm<<<dim3((N + 255) / 256, M), 256>>>(b.a, b.w, bias ? b.bias : nullptr, resid ? b.res : nullptr, g.aux_in ? b.aux_in : nullptr,
                                                 b.ref, b.ref_pre, M, N, K, act);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("sync: %s\n", hipGetErrorString(e)); return 1; }
    std::vector<uint16_t> ho((size_t)M * N), hp((size_t)M * N);
    std::vector<float> hr((size_t)M * N), hrp((size_t)M * N);
    hipMemcpy(ho.data(), b.o, ho.size() * 2, hipMemcpyDeviceToHost);
    hipMemcpy(hr.data(), b.ref, hr.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hp.data(), b.aux_out, hp.size() * 2, hipMemcpyDeviceToHost);
    hipMemcpy(hrp.data(), b.ref_pre, hrp.size() * 4, hipMemcpyDeviceToHost);
    double maxref = 0, maxerr = 0, maxerr_p = 0;
    size_t bad = 0, first_bad = 0;
    for (size_t i = 0; i < hr.size(); ++i) maxref = std::max(maxref, (double)fabsf(hr[i]));
    for (size_t i = 0; i < hr.size(); ++i) {
        uint32_t u = (uint32_t)ho[i] << 16; float v; memcpy(&v, &u, 4);
        const double err = fabs((double)v - hr[i]);
        if (!(err <= 0.01 * maxref + 0.01 * fabs(hr[i]))) { if (!bad) first_bad = i; ++bad; }
        maxerr = std::max(maxerr, err);
        if (act == THEIA_ACT_GELU) {
            uint32_t up = (uint32_t)hp[i] << 16; float vp; memcpy(&vp, &up, 4);
            maxerr_p = std::max(maxerr_p, fabs((double)vp - hrp[i]));
        }
    }
    const bool ok = bad == 0 && (act != THEIA_ACT_GELU || maxerr_p <= 0.02 * maxref + 0.05);
    printf("%s M=%d N=%d K=%d act=%d bias=%d resid=%d tile=%d: max|ref| %.3f max err %.4f%s", ok ? "ok  " : "FAIL", M, N, K, act, bias, resid,
           tile, maxref, maxerr, act == THEIA_ACT_GELU ? "" : "\n");
    if (act == THEIA_ACT_GELU) printf(" pre err %.4f\n", maxerr_p);
    if (bad) printf("     %zu bad elements, first at row %zu col %zu\n", bad, first_bad / N, first_bad % N);
    return ok ? 0 : 1;
}

static double time_one(const Bufs& b, int M, int N, int K, int act, bool bias, bool resid, int tile, int iters) {
    theia_gemm_args_t g = make_args(b, M, N, K, act, bias, resid, tile);
    for (int it = 0; it < 3; ++it) launch_any(&g, 0);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int it = 0; it < iters; ++it) launch_any(&g, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.0 / iters;
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const std::string what = argc > 1 ? argv[1] : "all";
    if (getenv("PPB_DYNAMIC")) g_pp_dynamic_override = atoi(getenv("PPB_DYNAMIC"));  // 1: work-conserving tile schedule, 0: static rounds
    printf("tile schedule: %s\n", g_pp_dynamic_override == 1 ? "dynamic (per-XCD queues)" : g_pp_dynamic_override == 0 ? "static" : "library default");
    const int iters = argc > 2 ? atoi(argv[2]) : 20;
    Bufs b;
    b.cap_a = (size_t)131072 * 3072; b.cap_w = (size_t)3072 * 3072; b.cap_o = (size_t)131072 * 768 > (size_t)25216 * 3072 ? (size_t)131072 * 768 : (size_t)25216 * 3072;
    hipMalloc(&b.a, b.cap_a * 2); hipMalloc(&b.w, b.cap_w * 2); hipMalloc(&b.o, b.cap_o * 2);
    hipMalloc(&b.res, b.cap_o * 2); hipMalloc(&b.aux_in, b.cap_o * 2); hipMalloc(&b.aux_out, b.cap_o * 2);
    hipMalloc(&b.bias, 4096 * 4); hipMalloc(&b.ref, (size_t)1200 * 3072 * 4); hipMalloc(&b.ref_pre, (size_t)1200 * 3072 * 4);
    fill_bf16(b.a, b.cap_a, 1.0f); fill_bf16(b.w, b.cap_w, 0.05f); fill_bf16(b.res, b.cap_o, 1.0f); fill_bf16(b.aux_in, b.cap_o, 1.5f);
    { std::vector<float> hb(4096); for (auto& v : hb) v = frand(); hipMemcpy(b.bias, hb.data(), hb.size() * 4, hipMemcpyHostToDevice); }
    int fails = 0;
    if (what == "check" || what == "all") {
        int tiles[4] = {256256, 320256, 128256, 160256};
        if (getenv("PPB_ONLY320")) tiles[0] = 320256;
        const int t_lo = getenv("PPB_ONLY_DW") ? 2 : 0;
#ifdef WITH_DW
        const int t_hi = 4;
#else
        const int t_hi = 2;
#endif
        for (int cap = 0; cap <= 5; cap += 5) {  // cap 5: every workgroup walks over several tiles, the last round is partial
            g_pp_grid_cap = cap;
            g_dw_grid_cap = cap;
            printf("-- persistent grid cap %d\n", cap);
            for (int ti = t_lo; ti < t_hi; ++ti) {
                const int t = tiles[ti];
                fails += check_one(b, 1000, 768, 768, THEIA_ACT_NONE, false, false, t);
                fails += check_one(b, 1000, 768, 768, THEIA_ACT_NONE, true, false, t);
                fails += check_one(b, 1000, 768, 768, THEIA_ACT_NONE, true, true, t);
                fails += check_one(b, 1187, 1000, 96, THEIA_ACT_NONE, true, false, t);     // ragged M, N; K = 3 half-tiles
                fails += check_one(b, 333, 264, 32, THEIA_ACT_NONE, true, true, t);        // K = one half-tile
                fails += check_one(b, 1111, 520, 64, THEIA_ACT_NONE, true, true, t);       // K = two half-tiles
                fails += check_one(b, 700, 3072, 768, THEIA_ACT_GELU, true, false, t);
                fails += check_one(b, 700, 768, 3072, THEIA_ACT_RELU, true, false, t);
                fails += check_one(b, 700, 3072, 768, THEIA_ACT_MUL_DGELU, false, false, t);
                fails += check_one(b, 650, 776, 64, THEIA_ACT_MUL_DRELU, false, false, t);
            }
        }
        g_pp_grid_cap = 0;
        g_dw_grid_cap = 0;
#ifdef WITH_W4
        if (!getenv("PPB_NO_W4")) {
            for (int cap = 0; cap <= 5; cap += 5) {
                g_w4_grid_cap = cap;
                printf("-- w4 (one wave per SIMD), persistent grid cap %d\n", cap);
                const int t = 256004;
                fails += check_one(b, 1000, 768, 768, THEIA_ACT_NONE, false, false, t);
                fails += check_one(b, 1000, 768, 768, THEIA_ACT_NONE, true, false, t);
                fails += check_one(b, 1187, 1024, 1024, THEIA_ACT_NONE, true, false, t);   // ragged M
                fails += check_one(b, 333, 256, 512, THEIA_ACT_NONE, true, false, t);      // the shortest K, one column tile
                fails += check_one(b, 700, 768, 3072, THEIA_ACT_NONE, true, false, t);
                fails += check_one(b, 700, 3072, 768, THEIA_ACT_GELU, true, false, t);
                fails += check_one(b, 1100, 1024, 1536, THEIA_ACT_GELU, true, false, t);    // GELU with plain k-steps behind the chunk steps
            }
            g_w4_grid_cap = 0;
        }
#endif
        printf("%d failures\n", fails);
    }
    if (what == "time" || what == "all") {
        struct S { const char* name; int M, N, K, act; bool bias, resid; };
        const S shapes[] = {
            {"fc1   ", 25216, 3072, 768, THEIA_ACT_GELU, true, false}, {"fc2   ", 25216, 768, 3072, THEIA_ACT_NONE, true, true},
            {"proj  ", 25216, 768, 768, THEIA_ACT_NONE, true, true},   {"qkv   ", 25216, 2304, 768, THEIA_ACT_NONE, true, false},
            {"qkv_d ", 25216, 768, 2304, THEIA_ACT_NONE, false, false}, {"fc2_d ", 25216, 3072, 768, THEIA_ACT_MUL_DGELU, false, false},
            {"fc1_d ", 25216, 768, 3072, THEIA_ACT_NONE, false, false}, {"proj_d", 25216, 768, 768, THEIA_ACT_NONE, false, false},
            {"up64c ", 131072, 768, 3072, THEIA_ACT_NONE, true, false}, {"lin16 ", 32768, 1024, 768, THEIA_ACT_NONE, true, false},
        };
        for (const S& s : shapes) {
            const double t256 = time_one(b, s.M, s.N, s.K, s.act, s.bias, s.resid, 256256, iters);
            const double t320 = time_one(b, s.M, s.N, s.K, s.act, s.bias, s.resid, 320256, iters);
#ifdef WITH_DW
            const double t128 = time_one(b, s.M, s.N, s.K, s.act, s.bias, s.resid, 128256, iters);
            const double t160 = time_one(b, s.M, s.N, s.K, s.act, s.bias, s.resid, 160256, iters);
#else
            const double t128 = 1e30, t160 = 1e30;
#endif
            const double fl = 2.0 * s.M * s.N * s.K;
            theia_gemm_args_t g = make_args(b, s.M, s.N, s.K, s.act, s.bias, s.resid, 0);
#ifdef WITH_W4
            {
                theia_gemm_args_t g4 = make_args(b, s.M, s.N, s.K, s.act, s.bias, s.resid, 256004);
                if (theia_gemm_nt_w4_supported(&g4, THEIA_BF16)) {
                    const double t4 = time_one(b, s.M, s.N, s.K, s.act, s.bias, s.resid, 256004, iters);
                    printf("%s M=%6d N=%4d K=%4d: W4    %7.1f us %7.1f TF\n", s.name, s.M, s.N, s.K, t4, fl / t4 / 1e6);
                }
            }
#endif
            printf("%s M=%6d N=%4d K=%4d: BM256 %7.1f us %7.1f TF | BM320 %7.1f us %7.1f TF | auto -> %d || dw128 %7.1f us %7.1f TF | dw160 %7.1f us %7.1f TF | auto -> %d\n",
                   s.name, s.M, s.N, s.K, t256, fl / t256 / 1e6, t320, fl / t320 / 1e6, theia_gemm_nt_pp_bm(&g, THEIA_BF16), t128, fl / t128 / 1e6, t160,
                   fl / t160 / 1e6, theia_gemm_nt_dw_bm(&g, THEIA_BF16));
        }
    }
#if defined(PP_TRACE) && defined(WITH_DW)
    {
        struct S { const char* name; int M, N, K, act; bool bias, resid; int tile; };
        const S dcfg[] = {{"fc1 gelu dw128", 25216, 3072, 768, THEIA_ACT_GELU, true, false, 128256},
                          {"fc1 gelu dw160", 25216, 3072, 768, THEIA_ACT_GELU, true, false, 160256},
                          {"qkv none dw160", 25216, 2304, 768, THEIA_ACT_NONE, true, false, 160256},
                          {"fc2_d dgelu dw128", 25216, 3072, 768, THEIA_ACT_MUL_DGELU, false, false, 128256},
                          {"fc2 resid dw160", 25216, 768, 3072, THEIA_ACT_NONE, true, true, 160256},
                          {"proj_d none dw160", 25216, 768, 768, THEIA_ACT_NONE, false, false, 160256}};
        for (const S& c : dcfg) {
            theia_gemm_args_t g = make_args(b, c.M, c.N, c.K, c.act, c.bias, c.resid, c.tile);
            for (int rep = 0; rep < 4; ++rep) launch_any(&g, 0);
            hipDeviceSynchronize();
            unsigned long long ph[8][16];
            unsigned int ids[512][2];
            hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_dw_phase), sizeof(ph));
            hipMemcpyFromSymbol(ids, HIP_SYMBOL(g_dw_ids), sizeof(ids));
            printf("%s: block 0 stamps (entry, addr set-up, prologue issued, then per tile: loop start, loop end, epilogue end, next tile ready)\n", c.name);
            for (int wv = 0; wv < 4; wv += 3) {
                printf("  wave %d:", wv);
                for (int k = 0; k < 16; ++k) printf(" %7lld", (long long)(ph[wv][k] - ph[0][0]));
                printf("\n");
            }
            if (&c == &dcfg[0]) {
                printf("  HW_ID / LDS_ALLOC of blocks 0..15, 256..271:");
                for (int k = 0; k < 16; ++k) printf(" %08x/%08x", ids[k][0], ids[k][1]);
                for (int k = 256; k < 272; ++k) printf(" %08x/%08x", ids[k][0], ids[k][1]);
                printf("\n");
                int lead = 0; for (int k = 0; k < 512; ++k) lead += (ids[k][1] & 0xfff) == 0;
                printf("  blocks with LDS base 0: %d of 512\n", lead);
            }
        }
    }
#endif
#if defined(W4_TRACE) && defined(WITH_W4)
    {
        struct S { const char* name; int M, N, K, act; bool bias, resid; };
        const S cfg[] = {{"fc1 gelu w4", 25216, 3072, 768, THEIA_ACT_GELU, true, false}, {"qkv none w4", 25216, 2304, 768, THEIA_ACT_NONE, true, false},
                         {"fc1_d none w4", 25216, 768, 3072, THEIA_ACT_NONE, false, false}, {"proj_d none w4", 25216, 768, 768, THEIA_ACT_NONE, false, false}};
        for (const S& c : cfg) {
            theia_gemm_args_t g = make_args(b, c.M, c.N, c.K, c.act, c.bias, c.resid, 256004);
            for (int rep = 0; rep < 4; ++rep) launch_any(&g, 0);
            hipDeviceSynchronize();
            unsigned long long ph[4][16];
            hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_w4_phase), sizeof(ph));
            printf("%s: block 0 stamps (entry, prologue issued, first fragments, then per tile: start, unrolled part done, tile done; [15] exit)\n", c.name);
            for (int wv = 0; wv < 4; wv += 3) {
                printf("  wave %d:", wv);
                for (int k = 0; k < 16; ++k) printf(" %7lld", (long long)(ph[wv][k] - ph[0][0]));
                printf("\n");
            }
        }
    }
#endif
#ifdef PP_TRACE
    {
        struct S { const char* name; int M, N, K, act; bool bias, resid; int tile; };
        const S cfg[] = {{"fc1 gelu BM256", 25216, 3072, 768, THEIA_ACT_GELU, true, false, 256256},
                         {"qkv none BM256", 25216, 2304, 768, THEIA_ACT_NONE, true, false, 256256},
                         {"fc2_d dgelu BM256", 25216, 3072, 768, THEIA_ACT_MUL_DGELU, false, false, 256256},
                         {"proj resid BM320", 25216, 768, 768, THEIA_ACT_NONE, true, true, 320256},
                         {"proj_d none BM320", 25216, 768, 768, THEIA_ACT_NONE, false, false, 320256},
                         {"proj resid BM256", 25216, 768, 768, THEIA_ACT_NONE, true, true, 256256}};
        for (const S& c : cfg) {
            theia_gemm_args_t g = make_args(b, c.M, c.N, c.K, c.act, c.bias, c.resid, c.tile);
            for (int rep = 0; rep < 4; ++rep) launch_any(&g, 0);  // back to back: the stamps are the last launch's
            hipDeviceSynchronize();
            unsigned long long ph[8][16];
            hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_pp_phase), sizeof(ph));
            printf("%s: block 0 stamps (entry, addr set-up, prologue issued, then per tile: loop start, loop end, epilogue end, next tile ready)\n", c.name);
            for (int wv = 0; wv < 8; wv += 4) {
                printf("  wave %d:", wv);
                for (int k = 0; k < 16; ++k) printf(" %7lld", (long long)(ph[wv][k] - ph[0][0]));
                printf("\n");
            }
        }
    }
#endif
    return fails ? 1 : 0;
}
