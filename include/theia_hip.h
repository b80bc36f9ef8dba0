/*
 * theia_hip.h -- C ABI of libtheia_hip.so: hand-written HIP (gfx950 / CDNA4) kernels for the Theia
 * distillation hot path (student ViT forward/backward, lconv translator heads, feature-matching loss).
 *
 * The reference (bdaiinstitute/theia) is pure Python and has NO native/FFI interface for this path
 * (SURVEY.md sec. 2b, 8b): the seam it offers is the nn.Module API (models/rvfm.py:94-185).  This header is
 * therefore the NEW boundary underneath that API; every entry point names the reference call site whose
 * ATen/cuDNN/cuBLAS kernels it replaces (paths relative to /root/reference/src/theia unless absolute).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only; all pointers are DEVICE pointers unless noted "host".
 *   - no allocation, no ownership transfer: the caller (PyTorch on the host side) owns every buffer.
 *   - every function enqueues work on `stream` (a hipStream_t passed as void*) and never synchronises.
 *   - return 0 on success, negative THEIA_ERR_* otherwise; theia_last_error() gives a thread-local message.
 *   - `dtype` selects the activation/operand element type: THEIA_F32 (exact-f32 MFMA path: parity mode) or
 *     THEIA_BF16 (bf16 operands, f32 accumulate: throughput mode).  Reductions / statistics are always f32.
 *   - activations are token-major / NHWC: [rows, channels] with channels contiguous.
 */
#ifndef THEIA_HIP_H_
#define THEIA_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define THEIA_ABI_VERSION 12

enum { THEIA_OK = 0, THEIA_ERR_INVALID = -1, THEIA_ERR_LAUNCH = -2, THEIA_ERR_UNSUPPORTED = -3 };
enum { THEIA_F32 = 0, THEIA_BF16 = 1,
       /* theia_gemm_nt only: operands A and W are OCP fp8 e4m3 bytes (quantised with theia_quantize_fp8, per-tensor scales),
        * accumulation is f32 on the fp8 matrix cores, the output / residual / aux tensors and every epilogue are bf16 */
       THEIA_FP8 = 2 };

int theia_abi_version(void);
const char* theia_last_error(void);
/* element size in bytes of a THEIA_* dtype */
int theia_dtype_size(int dtype);
/* Compute-unit budget of the GEMM planners (host-side state, read when a launch is enqueued).  The persistent NT kernel runs
 * min(tiles, budget) workgroups and the weight-gradient planner (theia_wgrad_splits) fills `budget` CUs per launch; both kernels
 * hold a whole CU per workgroup (130-150 KB of LDS, all of its registers), so a concurrent kernel that keeps k CUs -- RCCL runs one
 * workgroup per channel for the length of a collective -- would push k workgroups of a 256-workgroup launch into a second round and
 * double that launch's duration.  n = 0: every CU of the device (default); 0 < n: use n CUs (clamped to the device).  Replaces
 * nothing in the reference (DDP + cuBLAS leave this to the hardware scheduler): train_rvfm.py:125,258 is where the overlap arises.
 * Environment: THEIA_COMPUTE_CUS (initial value). */
int theia_set_compute_cus(int n);
int theia_get_compute_cus(void);
/* Tile schedule of the persistent NT GEMM (theia_gemm_nt, tiles 256256 / 320256; ABI v10).  0 = static rounds: workgroup w owns tiles w,
 * w + grid, ... -- the fastest schedule when the launch has the chip to itself (default).  1 = work-conserving: the schedule's positions
 * become one queue per XCD and a workgroup draws its next tile from the queue of the XCD it runs on (a scalar atomic on a per-launch
 * counter; the same L2 locality as the static rounds), so a launch that shares the chip -- RCCL channels at N > 1, another stream's
 * one-workgroup-per-CU kernels -- degrades in proportion to the CUs it actually gets instead of waiting for its late workgroups' whole
 * shares.  Measured alone the queues cost 3-8 % of a launch (profiles/r05_dynamic_tile_schedule.txt), hence opt-in.  Host-side state read
 * when a launch is enqueued; launches with K below 8 half k-tiles and launches recorded into a stream capture always use the static
 * schedule.  Returns the previous setting.  Environment: THEIA_PP_DYNAMIC (initial value).  Replaces nothing in the reference (cuBLAS /
 * DDP leave tile scheduling to the hardware): train_rvfm.py:125,258 is where the sharing arises. */
int theia_set_gemm_schedule(int dynamic);
int theia_get_gemm_schedule(void);

/* ------------------------------------------------------------------------------------------------
 * Row map: how GEMM row m of the (implicitly gathered) activation operand and of the output is found.
 *   rows per image R = rows_h*rows_w;  img = m / R;  (ry, rx) = divmod(m % R, rows_w)
 *   input pixel of tap t : (iy, ix) = (ry*in_sy + dy[t], rx*in_sx + dx[t]);  zero if outside in_h x in_w
 *   input address        : a + img*in_batch_stride + in_offset + (iy*in_w + ix)*in_c + ci
 *   K index              : k = t*in_c + ci     (weights use slot wslot[t]: column wslot[t]*in_c + ci)
 *   output pixel         : (oy, ox) = (ry*out_sy + out_y0, rx*out_sx + out_x0)
 *   output address       : out + img*out_batch_stride + out_offset + (oy*out_w + ox)*ldo + n
 * A plain row-major matrix is the degenerate case rows_h = rows_w = in_h = in_w = out_w = 1, ntaps = 1,
 * in_batch_stride = lda, out_batch_stride = ldo.
 * This one descriptor expresses nn.Linear, Conv2d 3x3 p1, ConvTranspose2d (stride 1, and stride 2 split in
 * output-parity classes), and all of their data-gradients  (adapter_heads.py:279-327).
 * ---------------------------------------------------------------------------------------------- */
#define THEIA_MAX_TAPS 9
typedef struct theia_rowmap {
    int32_t ntaps;
    int32_t dy[THEIA_MAX_TAPS];
    int32_t dx[THEIA_MAX_TAPS];
    int32_t wslot[THEIA_MAX_TAPS];
    int32_t rows_h, rows_w;
    int32_t in_h, in_w;
    int32_t in_sy, in_sx;
    int32_t in_c;
    int32_t out_w, out_sy, out_sx, out_y0, out_x0;
    int64_t in_batch_stride, in_offset;
    int64_t out_batch_stride, out_offset;
} theia_rowmap_t;

/* epilogue activation selector for theia_gemm_nt */
enum {
    THEIA_ACT_NONE = 0,
    THEIA_ACT_GELU = 1,      /* out = gelu_erf(v); if aux_out != NULL it receives v (pre-activation) */
    THEIA_ACT_RELU = 2,      /* out = max(v, 0) */
    THEIA_ACT_MUL_DGELU = 3, /* out = v * gelu_erf'(aux_in)   (backward of FC1's GELU) */
    THEIA_ACT_MUL_DRELU = 4  /* out = v * (aux_in > 0) */
};

/*
 * out[m, n] = act( sum_k A[m, k] * W[n, k] + bias[n] + rowtab[m % rowtab_period, n] ) + resid[m, n]
 *
 * A is gathered through `map` (see above); W is row-major [N, ldw] with K contiguous (the nn.Linear layout,
 * and the packed [co][slot][ci] layout produced by theia_pack_conv_weight for convolutions).
 * MFMA: v_mfma_f32_16x16x32_bf16 (bf16) / v_mfma_f32_16x16x4_f32 (f32), LDS-staged 128x128 tiles.
 * Replaces: aten::addmm / aten::linear of HF ViT (transformers modeling_vit.py:202-205,246-247),
 * aten::convolution of Conv2d / ConvTranspose2d and the head Linear (adapter_heads.py:282-288,307-326),
 * the patch-embedding conv (modeling_vit.py:60,69) and every data-gradient of those.
 * Requirements: K % 8 == 0 (bf16) / % 4 (f32); N % 8 == 0; in_c % 64 == 0 (bf16) / % 32 (f32) when ntaps > 1.
 */
typedef struct theia_gemm_args {
    const void* a;
    const void* w;
    void* out;
    const float* bias;    /* [N] or NULL */
    const void* resid;    /* dtype elements, output-indexed, or NULL */
    const void* aux_in;   /* dtype elements, output-indexed (ACT_MUL_*) or NULL */
    void* aux_out;        /* dtype elements, output-indexed (ACT_GELU pre-activation) or NULL */
    const float* rowtab;  /* f32 [rowtab_period, N] or NULL */
    int32_t rowtab_period;
    int32_t M, N, K;
    int32_t ldw;
    int32_t ldo;
    int32_t act;
    theia_rowmap_t map;
    /* tile request: 0 = let the launcher choose (theia_gemm_nt_tile / theia_gemm_nt_plan); 128128 / 128064 = the 2-stage kernel with
     * that tile; 256256 / 320256 = the persistent ping-pong kernel (gemm_pp.hip) with 256- / 320-row tiles (320: bf16 only; the
     * launcher picks it on its own when that saves whole rounds of the grid, e.g. 25216 x 768 outputs: 237 tiles = one round instead
     * of 297 = two); 256009 = the 256x256 ping-pong kernel for 3x3 stride-1 convolutions whose 256-row tile is one 16x16 output
     * image (the input slice is staged once, the 9 taps read shifted rows of it).
     * A request the problem does not meet is THEIA_ERR_UNSUPPORTED, never a fall-back, so a successful forced call proves which
     * kernel ran (the parity tests force every kernel on small shapes; bench.py cross-checks the automatic choice against a
     * forced one).  The ping-pong kernel takes: K and in_c multiples of 32 (bf16) / 16 (f32) / 64 (fp8), one tap's row <= 16 KiB,
     * M < 2^24, rows whose 16-row step wraps at most once per image row and per image (or plain matrices), no rowtab, a residual
     * only with act NONE, ln_sums only with act NONE / RELU and without residual / aux_in (such launches run the 2-stage kernels
     * when the choice is the launcher's).  256009: 9 taps on a full 3x3 grid, stride 1, rows_h = rows_w = 16, M % 256 == 0. */
    int32_t tile;
    int32_t reserved;
    /* optional: int64 [images][2] (declared float* for ABI stability: 16 bytes per image), accumulated atomically with the
     * 2^-24 fixed-point (sum, sum of squares) of the stored output values of each image -- integer adds, so the result does not
     * depend on arrival order -- (image = GEMM row / (rows_h*rows_w), needs rows_h*rows_w >= 128).  The whole-sample LayerNorm that follows a translator
     * convolution (adapter_heads.py:306-324) takes its statistics from here instead of re-reading the activation; the caller
     * zeroes the buffer before the first launch that writes the tensor (the 4 output-parity launches of a stride-2 transposed
     * convolution add into the same sums).  Together with resid / aux_in (no layer of the reference needs both) the launch runs
     * on the 128x128 kernel; tile requests 256256 / 256009 and THEIA_FP8 operands are then THEIA_ERR_UNSUPPORTED. */
    float* ln_sums;
    /* THEIA_FP8: device pointers to the de-quantisation factors 1/scale of the two operands (one f32 each); the accumulator is
     * multiplied by their product before the epilogue.  NULL = 1. */
    const float* a_scale_inv;
    const float* w_scale_inv;
    /* v11, THEIA_FP8 launches only (else THEIA_ERR_INVALID): `out` ALSO as e4m3, quantised from its bf16-rounded values with *out8_scale
     * (same pitch / row map as out) -- the next GEMM's operand without a quantisation pass of its own (fc1 -> fc2; fc2's data-gradient
     * -> fc1's).  NULL = off.  No maximum is recorded (the engine refreshes its delayed scales on the steps that run the separate
     * passes). */
    uint8_t* out8;
    const float* out8_scale;
} theia_gemm_args_t;

int theia_gemm_nt(const theia_gemm_args_t* args, int dtype, void* stream);
/* tile the launcher picks for an (M, N) problem when args->tile == 0: BM*1000 + BN (128128, 128064 or 256256) */
int theia_gemm_nt_tile(int M, int N, int dtype);
/* the kernel theia_gemm_nt would run for these arguments (no launch): 128128 / 128064 / 256000 (2-stage kernel, that tile),
 * 256256 (ping-pong), 256009 (ping-pong, 3x3 convolution with one image per tile); negative on a refused request */
int theia_gemm_nt_plan(const theia_gemm_args_t* args, int dtype);

/*
 * Weight gradient:  slab[s][n][wslot[t]*in_c + ci] = sum_{m in split s} dY[m, n] * A[m, (t, ci)]
 * dY is read through the OUTPUT side of `map` (where the forward wrote), A through the INPUT side.
 * `slabs` is f32 [splits][N][kslots*in_c]; reduce + layout-permute with theia_wgrad_reduce.
 * bf16 operands are transposed on the fly with ds_read_b64_tr_b16; f32 uses v_mfma_f32_16x16x4_f32.
 * Replaces: the weight-gradient half of convolution_backward / addmm backward (autograd of the above).
 * Requirements: N % 8 == 0, in_c % 64 == 0.
 */
typedef struct theia_wgrad_args {
    const void* dy;
    const void* a;
    float* slabs;
    int32_t M, N;
    int32_t ldo;     /* channels per output pixel of dY (= N for dense outputs) */
    int32_t kslots;  /* weight slots in the slab row: row length = kslots*in_c */
    int32_t splits;
    theia_rowmap_t map;
    /* optional fused bias gradient (the column sums of dY over the same rows; nn.Linear / Conv2d bias.grad):
     * bias_out[n] (+)= sum_m dY[m, n].  bias_slabs: scratch of splits*N floats.  Only when theia_wgrad_fuses_bias()
     * says the kernel that will run supports it (the ping-pong kernel feeds the dY fragments it already holds to one more
     * MFMA against a tile of ones); otherwise leave bias_out NULL and use theia_colsum. */
    float* bias_slabs;
    float* bias_out;
    int32_t bias_accumulate;
    int32_t defer_bias_reduce; /* 1: only write bias_slabs; the caller reduces them with theia_wgrad_finish (same launch as the weights) */
} theia_wgrad_args_t;

int theia_gemm_wgrad(const theia_wgrad_args_t* args, int dtype, void* stream);
int theia_wgrad_fuses_bias(const theia_wgrad_args_t* args, int dtype);
/* No launch: which kernel / row addressing a theia_gemm_wgrad call would use -- 0: the 2-stage kernel; the ping-pong kernel with
 * 100: per-row decode, 110: stepped rows, 111: plain row-major matrices, 112: periodic rows (16x16-like images).  < 0: error. */
int theia_gemm_wgrad_plan(const theia_wgrad_args_t* args, int dtype);
/* recommended number of M-splits for (M, N, kslots*in_c) so that the launch fills the CU budget (theia_get_compute_cus) */
int theia_wgrad_splits(int M, int N, int Ktot);
/* the same for a multi-tap map: the kernel's 256 x 256 output tiles do not straddle taps, so a launch has kslots * ceil(in_c / 256) of them
 * per 256 output columns (in_c = 384: 2 per tap, the second one half used) */
int theia_wgrad_splits_taps(int M, int N, int kslots, int in_c);
/* v11: n <= 4 plain-matrix weight gradients (nn.Linear: rm_plain maps) with the same M in ONE launch, e.g. a layer's o_proj and fused
 * q/k/v gradients (autograd of modeling_vit.py:207-238 under train_rvfm.py:125): 36 tiles x 7 M-splits instead of 9 x 28 and 27 x 9.
 * Each problem carries its own splits (theia_wgrad_group_splits(M, total tiles) for all of them), slabs and bias_slabs and is finished
 * with theia_wgrad_finish as after theia_gemm_wgrad.  Returns THEIA_ERR_UNSUPPORTED without launching anything when the problems do not
 * qualify (not bf16, not plain matrices, n > 4): launch them one by one then. */
int theia_gemm_wgrad_group(const theia_wgrad_args_t* probs, int n, int dtype, void* stream);
int theia_wgrad_group_splits(int M, int tiles);
/* output tiles (per tap) the ping-pong weight-gradient kernel cuts [N, in_c] into: 256 x 256, or 128 (n) x 384 (c) where that is less MFMA
 * work (N = in_c = 384: 3 instead of 4 at 56 % use); 0 when that kernel does not take the shape */
int theia_wgrad_tiles(int N, int in_c);

/* out[n*sn + slot*ss + ci*sc] (+)= sum_s slab[s][n][slot*C + ci]   (f32; permutes into the PyTorch layout) */
int theia_wgrad_reduce(const float* slabs, int splits, int N, int kslots, int C, float* out, int64_t sn,
                       int64_t ss, int64_t sc, int accumulate, void* stream);

/* The same reduction with BOTH sides coalesced (the permutation goes through an LDS tile) and, optionally, the bias partials of
 * the same weight-gradient GEMM (theia_wgrad_args_t.defer_bias_reduce = 1) reduced by the same launch:
 * bias_out[n] (+)= sum_s bias_slabs[s*N + n].  kslots <= 16. */
int theia_wgrad_finish(const float* slabs, int splits, int N, int kslots, int C, float* out, int64_t sn, int64_t ss, int64_t sc,
                       int accumulate, const float* bias_slabs, float* bias_out, int bias_accumulate, void* stream);

/* v12: the reductions behind one theia_gemm_wgrad_group launch as ONE launch (the small students run a layer's four nn.Linear gradients
 * as a group: 4 reductions of ~5 us each per layer were a tenth of the step's launches).  Each job is theia_wgrad_finish(slabs, splits, N, 1, C,
 * out, sn, 0, 1, accumulate, bias_slabs, bias_out, bias_accumulate): same summation order, bit-identical results.  THEIA_ERR_UNSUPPORTED
 * (nothing launched) when a job does not have the nn.Linear form that function's 16-byte path takes, or n > THEIA_WGRAD_FINISH_GROUP_MAX. */
#define THEIA_WGRAD_FINISH_GROUP_MAX 4
typedef struct {
    const float* slabs;
    float* out;
    const float* bias_slabs; /* NULL together with bias_out */
    float* bias_out;
    int64_t sn;              /* elements between output rows */
    int32_t N, C;
    int32_t accumulate, bias_accumulate;
} theia_wgrad_finish_job_t;
int theia_wgrad_finish_group(const theia_wgrad_finish_job_t* jobs, int n, int splits, void* stream);

/* out[n] (+)= sum_m x[m*ld + n]   (bias gradients; x has `dtype` elements, out f32)  */
int theia_colsum(const void* x, int64_t M, int N, int64_t ld, float* out, float* workspace, int accumulate,
                 int dtype, void* stream);
size_t theia_colsum_workspace_bytes(int64_t M, int N);

/* fp8 (OCP e4m3) quantisation with per-tensor delayed scaling, for THEIA_FP8 GEMM operands (BASELINE configs[3]):
 *   dst[r*C + c] = e4m3( clamp(src[r*ld + c] * *scale, +-448) ),   *amax = max(*amax, max |src|)   (src: THEIA_F32 / THEIA_BF16)
 * theia_fp8_update_scales turns the amax collected during a step into the scales of the next one:
 *   scale[i] = 448 / (amax[i] * margin) (unchanged when amax[i] == 0), inv_scale[i] = 1 / scale[i], amax[i] = 0. */
int theia_quantize_fp8(const void* src, int src_dtype, int64_t rows, int C, int64_t ld, uint8_t* dst, const float* scale, float* amax,
                       void* stream);
int theia_fp8_update_scales(float* amax, float* scale, float* inv_scale, int n, float margin, void* stream);
/* v11: the same quantisation for a whole table of contiguous bf16 tensors in ONE launch (the e4m3 copies of every GEMM weight operand after
 * an optimizer step).  theia_quantize_fp8_batch_plan fills first_block of a HOST table and returns the grid size (< 0: bad table); the table
 * is then copied to the device once and reused while pointers and sizes stay the same. */
typedef struct {
    const void* src;    /* bf16 [n], contiguous */
    uint8_t* dst;       /* e4m3 [n] */
    const float* scale; /* device scalar */
    float* amax;        /* device scalar, or NULL */
    int64_t n;          /* elements, a multiple of 8 */
    int32_t first_block;
    int32_t pad_;
} theia_quant_job_t;
int64_t theia_quantize_fp8_batch_plan(theia_quant_job_t* jobs_host, int njobs);
/* v11: quantisation fused into the producer.  The *_q8 forms of the HBM-bound passes below ALSO write their main output (y, dx, dpred) as
 * e4m3 with the slot's delayed scale and record max |value| -- the value as rounded to bf16, i.e. exactly what a theia_quantize_fp8 pass
 * over the bf16 output would produce: one extra byte per element written instead of a 3-byte-per-element pass.  q8 == NULL (or q8->out ==
 * NULL): the plain function.  bf16 only. */
typedef struct {
    uint8_t* out;       /* e4m3, same shape / pitch (in elements) as the bf16 output */
    const float* scale; /* device scalar */
    float* amax;        /* device scalar (atomic max), or NULL */
} theia_q8_out_t;
int theia_quantize_fp8_batch(const theia_quant_job_t* jobs_device, int njobs, int64_t total_blocks, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Parameter preparation (fp32 master weights in the reference state_dict layout -> operand layouts)
 * ---------------------------------------------------------------------------------------------- */
/* dst[i] = cast(src[i]) ; dst has `dtype` elements */
int theia_cast(const float* src, void* dst, int64_t n, int dtype, void* stream);
/* dst[c*ldd + r] = cast(src[r*C + c]) for a row-major [R, C] f32 matrix (W^T for data-gradients; ldd >= R lets
 * several transposes share one destination, e.g. q|k|v -> [D, 3D]) */
int theia_cast_transpose(const float* src, void* dst, int R, int C, int64_t ldd, int dtype, void* stream);
/* generic 3-D permuting cast: dst[(i*d1 + j)*d2 + k] = cast(src[i*s0 + j*s1 + k*s2])
 * (Conv2d [co,ci,3,3] / ConvTranspose2d [ci,co,3,3] -> packed [n][slot][c]; LN affine [C,H,W] -> [HW,C]) */
int theia_cast_permute3(const float* src, void* dst, int d0, int d1, int d2, int64_t s0, int64_t s1,
                        int64_t s2, int dtype, void* stream);
/* One launch for a whole table of permuting casts (every operand of the step is rebuilt from the fp32 master weights after
 * each optimizer update: ~250 casts / transposes / packs, launch-bound one by one).  Job j computes, for i < d0, j < d1,
 * k < d2:   dst[i*t0 + j*t1 + k] = cast(src[i*s0 + j*s1 + k*s2])   (dst elements: `dtype`, or f32 when dst_f32 != 0).
 * theia_cast_batch_plan fills the scheduling fields of a HOST table and returns the grid size; the table is then copied
 * to the device once and reused while pointers and shapes stay the same. */
typedef struct {
    const float* src;
    void* dst;
    int32_t d0, d1, d2;
    int32_t dst_f32;
    int64_t s0, s1, s2; /* source element strides */
    int64_t t0, t1;     /* destination element strides of i and j (k is contiguous) */
    /* filled by theia_cast_batch_plan */
    int64_t first_block;
    int32_t tiles1, tiles2;
    int32_t tile1, tile2;
} theia_cast_job_t;
int64_t theia_cast_batch_plan(theia_cast_job_t* jobs_host, int njobs);
int theia_cast_batch(const theia_cast_job_t* jobs_device, int njobs, int64_t total_blocks, int dtype, void* stream);
/* f32 -> f32 version writing with destination strides: dst[i*t0 + j*t1 + k*t2] (+)= src[(i*d1+j)*d2+k] */
int theia_unpermute3_f32(const float* src, float* dst, int d0, int d1, int d2, int64_t t0, int64_t t1,
                         int64_t t2, int accumulate, void* stream);

/* dst[c*R + r] (+)= src[r*C + c]  (f32 transpose, both sides coalesced: LayerNorm[C,H,W] affine gradients [HW][C] -> [C][HW]) */
int theia_transpose_acc_f32(const float* src, float* dst, int R, int C, int accumulate, void* stream);
/* v12: two matrices of one shape by one launch (a LayerNorm[C,H,W]'s weight and bias gradients, adapter_heads.py:306-324 under autograd) */
int theia_transpose_acc2_f32(const float* src0, float* dst0, int accumulate0, const float* src1, float* dst1, int accumulate1, int R, int C,
                             void* stream);

/* Image resize of the reference's HF image processor (models/backbones.py:337-339 -> Pillow Image.resize, two-pass 8-bit
 * resampling, Pillow 12.2.0 src/libImaging/Resample.c): src uint8 [b, in_h, in_w, 3] (channels_last) or [b, 3, in_h, in_w]
 * -> dst uint8 [b, out_h, out_w, 3].  The filter weights are host-computed 22-bit fixed-point tables (bounds: (first source
 * index, tap count) per output index; weights: [out][ksize]); the vertical tables are relative to first_row when a
 * horizontal pass runs (in_w != out_w), which covers source rows [first_row, first_row + tmp_rows) into tmp
 * (b*tmp_rows*out_w*3 bytes).  Bit-exact integer arithmetic. */
int theia_resize_u8(const uint8_t* src, uint8_t* dst, uint8_t* tmp, int b, int in_h, int in_w, int channels_last, int out_h,
                    int out_w, const int32_t* bounds_x, const int32_t* weights_x, int ksize_x, const int32_t* bounds_y,
                    const int32_t* weights_y, int ksize_y, int first_row, int tmp_rows, void* stream);
/* ------------------------------------------------------------------------------------------------
 * K1: image ingest.  uint8 [b,224,224,3] (channels_last=1) or [b,3,224,224] (0) -> patch matrix
 * [b*196, 768] with K index c*256 + ky*16 + kx, through a [3][256] f32 look-up table that reproduces the
 * HF processor's rescale+normalize arithmetic bit-exactly (backbones.py:337-339).
 * ---------------------------------------------------------------------------------------------- */
int theia_patchify_u8(const uint8_t* img, const float* lut, void* out, int b, int channels_last, int dtype,
                      void* stream);
/* the same for H x W images (interpolate_pos_encoding on non-224 inputs, backbones.py:314-341): [b, H, W, 3] / [b, 3, H, W] ->
 * [b * (H/16) * (W/16), 768]; pixels beyond the last whole 16x16 patch are not read (Conv2d stride 16) */
int theia_patchify_u8_hw(const uint8_t* img, const float* lut, void* out, int b, int H, int W, int channels_last, int dtype,
                         void* stream);
/* h[b, 0, :] = cls + pos[0]  (token 0 of every image; modeling_vit.py:148-149,159) */
int theia_write_cls(const float* cls, const float* pos, void* h, int b, int ntok, int D, int dtype, void* stream);
/* h[b, t0 + r, :] = tok[r, :] + pos[r, :], r < cnt  (tok, pos: f32 [cnt, D]): CLS token, and the register tokens appended
 * after the patches by the reg- students (backbones.py:196-205) */
int theia_write_tokens(const float* tok, const float* pos, void* h, int b, int ntok, int t0, int cnt, int D, int dtype,
                       void* stream);

/* ------------------------------------------------------------------------------------------------
 * K2: row LayerNorm over the last dim (eps 1e-12 in the ViT; modeling_vit.py:261-262,348)
 * ---------------------------------------------------------------------------------------------- */
int theia_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                        float* rstd, int64_t M, int D, float eps, int dtype, void* stream);
int theia_layernorm_fwd_q8(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int64_t M, int D,
                           float eps, int dtype, const theia_q8_out_t* q8, void* stream);
/* dx = LNbwd(dy) (+ dresid if non-NULL);  dgamma/dbeta (f32 [D]) (+)= reduction over rows.
 * workspace: theia_layernorm_bwd_workspace_bytes(M, D) */
int theia_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                        const void* dresid, void* dx, float* dgamma, float* dbeta, float* workspace,
                        int64_t M, int D, int accumulate, int dtype, void* stream);
int theia_layernorm_bwd_q8(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, const void* dresid,
                           void* dx, float* dgamma, float* dbeta, float* workspace, int64_t M, int D, int accumulate, int dtype,
                           const theia_q8_out_t* q8, void* stream);
size_t theia_layernorm_bwd_workspace_bytes(int64_t M, int D);

/* ------------------------------------------------------------------------------------------------
 * K11: whole-sample LayerNorm over (C,H,W) with elementwise affine (adapter_heads.py:306-324), NHWC.
 * x, y: [b, E] with E = H*W*C; gamma/beta f32 [E] already permuted to [HW, C]; stats f32 [b, 2] = (mean, rstd).
 * workspace: theia_layernorm_chw_workspace_bytes(b, E)
 * ---------------------------------------------------------------------------------------------- */
int theia_layernorm_chw_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats,
                            float* workspace, int b, int64_t E, float eps, int dtype, void* stream);
/* the same with the per-sample (sum, sum of squares) already known (theia_gemm_args_t.ln_sums of the producing GEMM: int64
 * fixed point, passed as const float*): one pass,
 * 4 B/element in bf16 instead of 6; writes stats (mean, rstd) for the backward pass */
int theia_layernorm_chw_fwd_sums(const void* x, const float* gamma, const float* beta, void* y, const float* sums, float* stats,
                                 int b, int64_t E, float eps, int dtype, void* stream);
int theia_layernorm_chw_fwd_sums_q8(const void* x, const float* gamma, const float* beta, void* y, const float* sums, float* stats, int b,
                                    int64_t E, float eps, int dtype, const theia_q8_out_t* q8, void* stream);
/* dx = LNbwd(dy) * (relu_mask ? (x > 0) : 1): the optional mask folds the backward of the ReLU that produced x.
 * dgamma/dbeta f32 [E] (+)= sum over the batch of dy*xhat / dy. */
int theia_layernorm_chw_bwd(const void* dy, const void* x, const float* gamma, const float* stats, void* dx,
                            float* dgamma, float* dbeta, float* workspace, int b, int64_t E, int relu_mask,
                            int accumulate, int dtype, void* stream);
/* The same, and dxsum[c] (+)= sum over samples and pixels of dx[s, pixel, c] (f32 [C], C a multiple of 8 dividing E, channels innermost):
 * the bias gradient of the convolution that produced x (adapter_heads.py:304-327: Conv2d / ConvTranspose2d -> LayerNorm) without a
 * second pass over dx.  dxsum == NULL: plain theia_layernorm_chw_bwd. */
int theia_layernorm_chw_bwd_colsum(const void* dy, const void* x, const float* gamma, const float* stats, void* dx,
                                   float* dgamma, float* dbeta, float* workspace, int b, int64_t E, int relu_mask,
                                   int accumulate, float* dxsum, int C, int dxsum_accumulate, int dtype, void* stream);
int theia_layernorm_chw_bwd_colsum_q8(const void* dy, const void* x, const float* gamma, const float* stats, void* dx, float* dgamma,
                                      float* dbeta, float* workspace, int b, int64_t E, int relu_mask, int accumulate, float* dxsum, int C,
                                      int dxsum_accumulate, int dtype, const theia_q8_out_t* q8, void* stream);
size_t theia_layernorm_chw_workspace_bytes(int b, int64_t E);

/* ------------------------------------------------------------------------------------------------
 * K4: multi-head self-attention softmax(Q K^T / sqrt(dh)) V, dh = 64, whole sequence per workgroup
 * (modeling_vit.py:164-189).  qkv: [b, n, 3*D] (q | k | v, heads contiguous inside each);  o: [b, n, D];
 * lse: f32 [b, h, n] (log-sum-exp of the scaled scores, saved for backward).  delta_ws: f32 [b, h, n] scratch of the backward pass
 * (theia_attention_bwd_workspace_bytes; the bf16 path for n <= 208 computes dQ, dK and dV in one kernel and does not touch it).
 * ---------------------------------------------------------------------------------------------- */
int theia_attention_fwd(const void* qkv, void* o, float* lse, int b, int n, int h, int dtype, void* stream);
int theia_attention_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, void* dqkv,
                        float* delta_ws, int b, int n, int h, int dtype, void* stream);
size_t theia_attention_bwd_workspace_bytes(int b, int n, int h);

/* ------------------------------------------------------------------------------------------------
 * K13: feature-matching losses of one teacher (models/rvfm.py:153-176): pred [b, E] (dtype), target [b, E] (f32).
 * fwd:  losses[0..2] = (mse, cos, smooth_l1);  coef f32 [b, 2] per-sample cosine-gradient coefficients.
 * bwd:  dpred = w[0]*d mse/dpred + w[1]*d cos/dpred + w[2]*d l1/dpred, w = 3 f32 on the device.
 * workspace: theia_distill_loss_workspace_bytes(b, E)
 * ---------------------------------------------------------------------------------------------- */
int theia_distill_loss_fwd(const void* pred, const float* target, float* losses, float* coef, float* workspace,
                           int b, int64_t E, int dtype, void* stream);
int theia_distill_loss_bwd(const void* pred, const float* target, const float* coef, const float* w, void* dpred,
                           int b, int64_t E, int dtype, void* stream);
size_t theia_distill_loss_workspace_bytes(int b, int64_t E);
/* v11: the same with the teacher features in `target_dtype`: THEIA_F32, or THEIA_BF16 beside bf16 predictions -- the reference's loader
 * produces them by bf16 arithmetic and widens them (dataset/data_utils.py:374-379), so a bf16 target holds the same values and the
 * results are bit-identical; each of the two passes reads 2 bytes less per element. */
int theia_distill_loss_fwd_t(const void* pred, const void* target, int target_dtype, float* losses, float* coef, float* workspace,
                             int b, int64_t E, int dtype, void* stream);
int theia_distill_loss_bwd_t(const void* pred, const void* target, int target_dtype, const float* coef, const float* w, void* dpred,
                             int b, int64_t E, int dtype, void* stream);
int theia_distill_loss_bwd_q8(const void* pred, const void* target, int target_dtype, const float* coef, const float* w, void* dpred,
                              int b, int64_t E, int dtype, const theia_q8_out_t* q8, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K15: token selection / pooling (models/utils.py:31-43).  x: [b, n, D]; mode 0: x[:, 1:n-disc] -> [b, n-1-disc, D];
 * 1: mean over those tokens -> [b, D]; 2: max -> [b, D]; 3: cls x[:,0] -> [b, D].  Output is f32.
 * ---------------------------------------------------------------------------------------------- */
int theia_token_select(const void* x, float* out, int b, int n, int D, int disc, int mode, int dtype, void* stream);

/* K14: bf16 teacher-feature normalisation with the reference's two bf16 roundings
 * (dataset/data_utils.py:342-355,374-379): x bf16 [rows, C] -> f32 [rows, C] */
int theia_feature_norm_bf16(const uint16_t* x, const float* mean, const float* std, float* out, int64_t rows,
                            int C, void* stream);
/* Teacher-feature ingest, whole batch: x_chw [b, C, HW] bf16 exactly as stored by the feature extractor (safetensors
 * "embedding", feature_extraction_core/models.py:55-97) -> out [b, HW, C] f32 = float(bf16(bf16(x - bf16(mean[c])) / bf16(std[c])))
 * (decode_sample's rearrange, dataset/data_utils.py:152-155; normalize_feature :342-355 with bf16 stats :374-379; .float()
 * scripts/train/train_rvfm.py:112-114).  mean == std == NULL: transpose + widen only. */
int theia_feature_ingest_bf16(const uint16_t* x_chw, const float* mean, const float* std, float* out, int b, int C, int HW,
                              void* stream);

/* ------------------------------------------------------------------------------------------------
 * elementwise helpers on `dtype` buffers
 * ---------------------------------------------------------------------------------------------- */
int theia_add_inplace(void* dst, const void* src, int64_t n, int dtype, void* stream); /* dst += src */
int theia_fill_zero(void* dst, int64_t bytes, void* stream);
/* strided copy/accumulate of token rows: dst[b, t0 + t, :] (+)= src[b, t, :] ; used for d(final LN output) */
int theia_scatter_tokens(const void* src, void* dst, int b, int nsrc, int ndst, int t0, int D, int accumulate,
                         int dtype, void* stream);

/* fused multi-tensor AdamW over a flat f32 parameter arena (train_rvfm.py:131; torch.optim.AdamW semantics):
 * p -= lr*(m_hat/(sqrt(v_hat)+eps) + wd*p) with decoupled weight decay; segments give per-range wd. */
int theia_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                     float eps, float weight_decay, float bias_c1, float bias_c2, float grad_scale, void* stream);
/* the same with the gradient scale read from device memory (*grad_scale_dev): the clip coefficient of theia_grad_clip_coef */
int theia_adamw_step_scaled(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                            float eps, float weight_decay, float bias_c1, float bias_c2, const float* grad_scale_dev, void* stream);

/* the same with the per-step scalars read from device memory: hyper_dev = {learning rate, 1 - beta1^t, 1 - beta2^t} (f32 x 3), and an
 * optional device-resident gradient scale (NULL: 1).  Kernel arguments are frozen when a launch is captured into a hipGraph; with the
 * scalars that change every step on the device, the optimizer step of train_rvfm.py:131 (+ the scheduler's learning rate, :133) is part
 * of a captured train step (theia_amd/train_graph.py): the host writes three floats per step and replays. */
int theia_adamw_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float beta1, float beta2, float eps,
                         float weight_decay, const float* hyper_dev, const float* grad_scale_dev, void* stream);

/* dst[i] = float(src_bf16[i]) * scale: widens a bf16 gradient-exchange buffer back into the fp32 gradient bucket after the RCCL
 * all-reduce (the optional bf16 exchange of theia_amd/parallel.py; replaces nothing in the reference, whose DDP exchanges fp32:
 * train_rvfm.py:258).  Both buffers 16-byte aligned. */
int theia_upcast_scale_bf16(const void* src_bf16, float* dst, int64_t n, float scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Gradient exchange over RCCL / xGMI through the C ABI: what DistributedDataParallel's reducer does for the reference
 * (scripts/train/train_rvfm.py:258 wraps the model, :125 backward() fires the bucket all-reduces; accelerate / torch.distributed
 * "nccl" underneath).  One communicator per process = per GPU; the caller moves the 128-byte id from rank 0 to the other ranks
 * out of band (a file, MPI, torch.distributed's store), exactly as with ncclGetUniqueId.  RCCL is bound when the first theia_comm_*
 * function is called (dlopen "librccl.so.1": the copy already in the process if there is one); THEIA_ERR_UNSUPPORTED if it cannot
 * be found.  Collectives are IN PLACE on `buf`, enqueued on `stream`, never synchronised; every rank must issue the same sequence.
 * The engine fills flat gradient buckets in backward-completion order (theia_amd/engine.py), so the host side is one
 * theia_comm_allreduce(bucket, average = 1) per finished bucket on a side stream (theia_amd/parallel.py, THEIA_DP_BACKEND=abi).
 * ------------------------------------------------------------------------------------------------ */
#define THEIA_COMM_ID_BYTES 128
/* rank 0: id_host[0 .. THEIA_COMM_ID_BYTES) (HOST memory) = a fresh communicator id */
int theia_comm_unique_id(void* id_host);
/* all ranks, collectively, each with its GPU current (hipSetDevice): *comm_out = this rank's communicator */
int theia_comm_init(void** comm_out, const void* id_host, int world, int rank);
/* buf[0 .. count) = sum (average = 0) or mean (average != 0: RCCL's AVG, no scale kernel) over the ranks; dtype THEIA_F32 / THEIA_BF16 */
int theia_comm_allreduce(void* comm, void* buf, int64_t count, int dtype, int average, void* stream);
/* buf of rank `root` to every rank: the parameter broadcast DDP's constructor performs (train_rvfm.py:258) */
int theia_comm_broadcast(void* comm, void* buf, int64_t count, int dtype, int root, void* stream);
int theia_comm_size(void* comm, int* world, int* rank);
/* collective; NULL is a no-op */
int theia_comm_destroy(void* comm);

/* Global-norm gradient clipping over flat f32 gradient ranges (nn.utils.clip_grad_norm_ at train_rvfm.py:126-130) without a host
 * round trip.  theia_grad_sumsq: partials[0 .. theia_grad_sumsq_blocks()) = partial sums of squares of g[0..n) (g 16-byte aligned;
 * fixed block order: bit-reproducible).  theia_grad_clip_coef: out2[0] = sqrt(sum of the nparts partials) = total norm,
 * out2[1] = min(1, max_norm / (total + 1e-6)) -- the factor torch multiplies every gradient with; here it goes to
 * theia_adamw_step_scaled, the gradient buffers themselves are left unscaled. */
int theia_grad_sumsq_blocks(void);
int theia_grad_sumsq(const float* g, int64_t n, float* partials, void* stream);
int theia_grad_clip_coef(const float* partials, int nparts, float max_norm, float* out2, void* stream);

/* hardware probe used by tests: transposed LDS read semantics of ds_read_b64_tr_b16 (out: 64 lanes x 4 u16) */
int theia_probe_tr16(const uint16_t* lds_image_1024, const int32_t* lane_byte_addr_64, uint16_t* out_256,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif /* THEIA_HIP_H_ */
