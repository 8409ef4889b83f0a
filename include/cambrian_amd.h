/*
 * cambrian_amd.h — C-ABI of the MI355X (gfx950) kernel library behind Cambrian-1's
 * vision-tower + Spatial-Vision-Aggregator (SVA) hot path.
 *
 * The reference (cambrian-mllm/cambrian) is 100 % Python and has no FFI of its own
 * (SURVEY.md §8b); every entry point below therefore cites the *library call site* in the
 * reference that it replaces (file:line relative to the reference repo root).  The Python
 * host side (cambrian_amd/ops.py) binds these symbols with ctypes and wraps them in
 * torch.autograd.Function objects; INTEGRATION.md shows the binding a reference maintainer
 * would add.
 *
 * Conventions
 *   - plain C: pointers, sizes, scalars, POD structs.  No torch / C++ types.
 *   - every buffer is caller-owned device memory (HBM) valid on the given stream.
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream).
 *   - every function returns 0 (CMB_OK) or a negative cmb_status; nothing throws, nothing
 *     synchronises the host, nothing allocates.
 *   - dtype codes: CMB_BF16 = 0 (bf16 storage, fp32 accumulate), CMB_F32 = 1 (exact fp32
 *     path on v_mfma_f32_32x32x2_f32; used by the parity tests).
 *   - "rowmap": a logical row index r of a [rows, cols] matrix is mapped to an element
 *     offset  (r / n1) * s0 + ((r % n1) / n2) * s1 + (r % n2) * s2 ; n1 == 0 means the
 *     plain row-major case r * s2.  This is how the window re-arrangement
 *     (cambrian_arch.py:271-287) and the in-LLM slice hidden[:, 91:691] viewed as
 *     [B,24,25,H][:, :, :24] (cambrian_llama.py:181-207) are folded into the consumer's
 *     loads instead of being materialised.
 */
#ifndef CAMBRIAN_AMD_H
#define CAMBRIAN_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum cmb_status {
  CMB_OK = 0,
  CMB_ERR_BAD_ARG = -1,      /* null pointer, negative size, unsupported combination */
  CMB_ERR_ALIGNMENT = -2,    /* pointer / leading dimension not 16-byte aligned */
  CMB_ERR_SHAPE = -3,        /* K not a multiple of the K-step, N not a multiple of 8, ... */
  CMB_ERR_WORKSPACE = -4,    /* workspace too small */
  CMB_ERR_LAUNCH = -5        /* hipLaunchKernel / hipFuncSetAttribute failed */
} cmb_status;

enum { CMB_BF16 = 0, CMB_F32 = 1 };
/* 8-bit operands of cmb_gemm (OCP e4m3fn, gfx950's native fp8; BASELINE configs[4] "fp8 MFMA projection GEMMs"):
 * A / B are uint8 [rows, K] produced by cmb_quantize_fp8_rows, C / residual / pre_out are bf16 (or fp32 C). */
enum { CMB_FP8_E4M3 = 3 };
enum { CMB_ACT_NONE = 0, CMB_ACT_GELU_ERF = 1, CMB_ACT_GELU_TANH = 2, CMB_ACT_QUICK_GELU = 3,
       CMB_ACT_SILU = 4,
       /* cmb_gemm only: the gated activation of a packed (gate | up) projection whose weight rows are INTERLEAVED (row 2j =
        * gate_j, row 2j + 1 = up_j): C[m, j] = silu(v[m, 2j]) * v[m, 2j + 1] with v = alpha * acc + bias — C has N / 2
        * columns (HF Dinov2SwiGLUFFN reached from dino_encoder.py:159: weights_in + silu(x1) * x2 in one launch, the
        * [M, N] intermediate never written).  N % 16 == 0; no colscale / residual / pre_out / split-K / fp32 C. */
       CMB_ACT_SWIGLU_PAIRS = 5 };

typedef struct cmb_rowmap {
  int64_t n1, n2;      /* n1 == 0: identity (offset = r * s2) */
  int64_t s0, s1, s2;  /* element strides */
} cmb_rowmap;

/* library / build identification ("cambrian_amd <version> gfx950"). */
const char* cmb_version(void);
/* ABI revision: bumped whenever an entry point's signature or a descriptor's layout changes (round 2's key_valid
 * arguments = 2, round 3's fold_kv workspace = 3, the batch fields of cmb_gemm_desc = 4,
 * the kernel-selection knobs of round 4 = 5, cmb_layernorm_fwd_multi = 9, cmb_ln_multi_desc.dx_out = 10).  Bindings must compare it with the revision they
 * were written against (CMB_ABI_VERSION; cambrian_amd/lib.py::load raises on a mismatch): every symbol of a stale
 * library still resolves, and a shifted argument list corrupts memory instead of failing. */
#define CMB_ABI_VERSION 10
int cmb_abi_version(void);

/* Run-time kernel-selection knobs: which of several kernels that compute the SAME function an entry point launches
 * (value 0 = the earlier kernel, kept for A/B runs; CMB_KNOB_DEFAULTS = what the library uses unless told otherwise).
 * Meant for same-process A/B measurements (tools/r04_lab.py, bench.py's `ab` block) and start-up calibration; none
 * changes an entry point's contract.  Not thread-safe against concurrent launches: set between steps.
 *   CMB_KNOB_LN_FWD     cmb_layernorm_fwd, affine rows without a position table: 0 = one row per wave at a time,
 *                       parameters re-read per row (rounds 1-3); 1 = parameters staged in LDS + next row prefetched,
 *                       rows / 16 workgroups up to 8192; > 1 = the same with that workgroup cap
 *   CMB_KNOB_DWCONV     cmb_dwconv7x7_nhwc (C % 64 == 0): 0 = LDS-tiled kernel (rounds 1-3); 1 = column-walking kernel with
 *                       register-resident taps, 64 rows per chunk for maps of >= 128 rows, else 32; > 1 = that many rows per chunk
 *   CMB_KNOB_VIT_ATTN   cmb_vit_attn_fwd (bf16): 0 = two barriers per key tile (round 3); 1 = two LDS tile buffers, one barrier
 *                       (both: register-staged K / V tiles, bit-identical); 2 = round 6: K / V tiles by LDS-DMA into swizzled
 *                       images, V by transposing reads, lazy rescale, XCD-grouped 1-D grid, 128 queries per workgroup; 3 = the
 *                       same with 256 queries per workgroup (2 / 3 agree with 0 / 1 to bf16 rounding, not bit for bit)
 *   CMB_KNOB_SVA_ABS    cmb_sva_abs_fwd / _bwd on bf16 operands: 0 = the MFMA kernels; 1 = the exact (plain fp32 arithmetic)
 *                       instantiation of the same algorithm that dtype CMB_F32 always runs (tests: one against the other)
 *   CMB_KNOB_LN_MULTI_CHUNK  cmb_layernorm_bwd_multi: layers per launch, 4 ... 7 (4, 5: two waves per SIMD; 6, 7: one).  13 layers of the
 *                       9216-token tower at 24 images: 4: 4684 us, 5: 4463, 6: 6431, 7: 6455 (profiles/r06_lab.md)
 *   CMB_KNOB_FLASH      cmb_flash_attn_fwd / _bwd, bit mask: 1 = forward, 2 = dQ on LDS-DMA operand tiles with transposing reads
 *                       (flash2.hip); 4 = the round-4 dK/dV kernel with the round-5 four-phase tile body; 16 = that kernel reading its
 *                       Q^T / dO^T fragments with transposing reads from swizzled row-major images (no transposed copies); 0 = the
 *                       round-4 kernels.  Default 23.  (Bit 8 selected a dK/dV kernel on LDS-DMA tiles that measured slower —
 *                       profiles/r05_lab.md — and was removed in round 6: values with it are rejected.)  dQ / dK / dV are
 *                       bit-identical across variants, the forward to fp32 rounding (tests/test_flash_bwd_gpu.py)
 *   CMB_KNOB_COLSUM_WGS cmb_colsum / cmb_colsum_scaled: 0 = 256 rows per workgroup whatever the input (rounds 1-5); n > 0 = rows per
 *                       workgroup chosen so that the launch has about n workgroups (16 ... 256 rows; same sums up to the order of
 *                       the fp32 atomics, which was never fixed; default 768: 13 824 x 1024 24 -> 14 us, scaled 42 -> 17 us)
 *   CMB_KNOB_LN_BWD_ROWS cmb_layernorm_bwd on inputs of < 65 536 rows per window position: rows per workgroup (4 ... 256; 16 = rounds
 *                       3-5, default 32: 13 824 x 1024 51 -> 39 us) — fewer workgroups = fewer end-of-block atomics on the parameter gradients, more = more rows in flight */
enum cmb_knob_id { CMB_KNOB_LN_FWD = 0, CMB_KNOB_DWCONV = 1, CMB_KNOB_VIT_ATTN = 2, CMB_KNOB_SVA_ABS = 3, CMB_KNOB_LN_MULTI_CHUNK = 4,
                   CMB_KNOB_FLASH = 5, CMB_KNOB_COLSUM_WGS = 6, CMB_KNOB_LN_BWD_ROWS = 7, CMB_KNOB_COUNT = 8 };
#define CMB_KNOB_DEFAULTS 1, 1, 2, 0, 5, 23, 768, 32
int cmb_knob_set(int32_t knob, int32_t value);   /* CMB_ERR_BAD_ARG for an unknown knob */
int cmb_knob_get(int32_t knob);                  /* -1 for an unknown knob */

/* ------------------------------------------------------------------------------------------
 * GEMM   C[M,N] = epilogue( alpha * A[M,K] · B[N,K]^T )            (nn.Linear layout)
 * replaces: every torch.nn.Linear / F.linear on the path —
 *   vision_sampler.py:170-175,187-189,232,240-245,254-255 (SVA projections),
 *   cambrian_arch.py:49,56,375,411 (mm_projector_aux_i / mm_projector),
 *   HF CLIPVisionModel / Dinov2Model / timm ViT + ConvNeXt linears reached from
 *   clip_encoder.py:104, siglip_encoder.py:97, dino_encoder.py:159, clip_convnext_encoder.py:133-136.
 * epilogue order:  v = alpha*acc + bias[n];  pre_out = v;  v = act(v);  v *= colscale[n];
 *                  v += residual[m,n];  v += beta*C_old[m,n] (fp32 C only);  C = v
 * requirements: K % (128 / sizeof(elem)) == 0; N % 8 == 0; 16-byte aligned rows.
 * dtype CMB_FP8_E4M3: v_mfma_f32_32x32x64_f8f6f4 (K = 64 per instruction, twice the bf16 MFMA rate), 128x128 tile,
 *   no split-K; see a_scale / b_scale.
 * two tile configurations: 128x128 / 4 waves, and for bf16 256x256 / 8 waves with the
 *   8-phase LDS-DMA schedule; `tile_hint` = 0 lets the library pick by grid fill.
 * split_k > 1: fp32 partial slabs go to `workspace` (split_k*M*N*4 bytes) and are reduced by a
 *   second kernel; only alpha/beta and out_dtype apply (used for weight gradients).
 * ---------------------------------------------------------------------------------------- */
typedef struct cmb_gemm_desc {
  int32_t dtype;       /* CMB_BF16 | CMB_F32 : element type of A, B, residual, pre_out */
  int32_t out_dtype;   /* CMB_BF16 | CMB_F32 : element type of C (bf16 requires dtype bf16) */
  int64_t M, N, K;
  const void* A;  cmb_rowmap a_map;
  const void* B;  int64_t ldb;
  void* C;        cmb_rowmap c_map;
  const float* bias;       /* [N] fp32 or NULL */
  const float* colscale;   /* [N] fp32 or NULL (LayerScale) */
  const void* residual; cmb_rowmap r_map;  /* NULL or [M,N] of `dtype` */
  void* pre_out;        cmb_rowmap p_map;  /* NULL or [M,N] of `dtype`: pre-activation copy */
  int32_t act;
  float alpha, beta;
  int32_t split_k;
  void* workspace; int64_t workspace_bytes;
  int32_t tile_hint;       /* 0 = choose by grid-fill cost model; 128 / 256 = force that block tile (bf16 only);
                              2560 / 2561 = 256 tile with schedule 0 (8-phase ping-pong, default) / 1 (in-wave pipeline);
                              2590 = 4-wave register-buffered 256x256 kernel (what 0 / 256 pick when N % 256 == 0 and there is
                              more than one round of tiles).  0 also consults the per-shape policy (cmb_gemm_policy_set) */
  const float* a_scale;    /* CMB_FP8_E4M3 only: [M] fp32 dequantisation factor of each A row (or NULL = 1) */
  const float* b_scale;    /* CMB_FP8_E4M3 only: [N] fp32 dequantisation factor of each B row (or NULL = 1);
                              the accumulator is multiplied by a_scale[m] * b_scale[n] before alpha / bias */
  int32_t batch;           /* > 1: `batch` independent problems of this shape in one launch (the sixteen per-head GEMMs of
                              the absorbed SVA projections, vision_sampler.py:187-189 restated per head): problem z uses
                              A + z * a_batch_stride, B + z * b_batch_stride, C + z * c_batch_stride (elements; 16-byte
                              multiples).  bf16 / fp32, 128x128 tile, no bias / colscale / pre_out / split-K; a residual (operand dtype)
                              is laid out as C: problem z adds residual + z * c_batch_stride through r_map */
  int64_t a_batch_stride, b_batch_stride, c_batch_stride;
  /* LayerNorm folded into the linear that follows it (round 6; the frozen towers' LN -> qkv / fc1 pairs: HF / timm blocks behind
   * clip_encoder.py:104, siglip_encoder.py:97, dino_encoder.py:159, clip_convnext_encoder.py:133-136):
   *   LN(x) W^T + b  =  rstd[m] * ( x W'^T - mean[m] * col_sum[n] ) + b'[n],   W' = W diag(gamma) (as stored in B),
   *   col_sum[n] = sum_k B[n,k] (of the stored, rounded values),  b' = b + W beta  (passed as `bias`)
   * so the GEMM reads the UN-normalised rows and cmb_row_stats replaces the LayerNorm kernel (no normalised copy is written).
   * With row_mean set the epilogue starts  v = row_rstd[m] * (acc - row_mean[m] * col_sum[n]) + bias[n]  (alpha must be 1,
   * bias / row_rstd / col_sum non-NULL; bf16 operands; no split-K, batch, fp8 scales); act, colscale, residual follow as usual. */
  const float* row_mean;   /* [M] fp32 or NULL */
  const float* row_rstd;   /* [M] fp32 */
  const float* col_sum;    /* [N] fp32 */
} cmb_gemm_desc;

int cmb_gemm(const cmb_gemm_desc* d, void* stream);
/* Two INDEPENDENT bf16 products in one call (round 6): the same results as cmb_gemm(d0) followed by cmb_gemm(d1), bit for bit.
 * When both would run on the persistent 256 x 256 kernel with their whole K, the same `act`, no batch / split_k / pre_out /
 * tile_hint / folded LayerNorm, and whole-round arithmetic says it pays, they leave as ONE launch whose workgroups are split
 * between the two problems — the frozen ViT towers' same-position linears (DINOv2 17520 x 1536 x {1536, 4096} beside SigLIP
 * 17496 x 1152 x {1152, 4352} at 24 images: 1.62 and 1.35 rounds of 256 workgroups each, 3.0 + 2.9 rounds on 138 + 118 side by
 * side; HF / timm blocks behind dino_encoder.py:156-165, siglip_encoder.py:95-99, towers independent: cambrian_arch.py:271-278).
 * Otherwise the two calls are made one after the other.  cmb_gemm_pair_last() = 1 if the calling thread's last cmb_gemm_pair
 * took the one-launch path.  CMB_GEMM_PAIR=0 in the environment forces two launches (A/B runs). */
int cmb_gemm_pair(const cmb_gemm_desc* d0, const cmb_gemm_desc* d1, void* stream);
int cmb_gemm_pair_last(void);
/* C[M,N] = alpha * At[K,M]^T * Bt[K,N] (+ beta * C): the weight-gradient product dW = g^T x of every trainable linear of
 * the path (autograd's addmm on a transposed view in the reference: the SVA projections vision_sampler.py:159-189, the
 * projectors cambrian_arch.py:49-56), with BOTH operands row-major over the contraction rows as the activations lie in
 * memory — no transposed copies.  Same descriptor as cmb_gemm, read as: A = At (row stride a_map.s2, a_map.n1 == 0),
 * B = Bt (row stride ldb), K = contraction rows (any count: rows beyond K contribute zero), M, N multiples of 8;
 * dtype CMB_BF16, out_dtype CMB_F32 | CMB_BF16; split_k / workspace and batch / *_batch_stride as cmb_gemm (both together
 * when the batch's results are contiguous: c_batch_stride == M * N, row stride N; workspace split_k * batch * M * N * 4 bytes);
 * bias, colscale, residual, pre_out, act must be unset.  cmb_gemm_last_kernel() reports 1281. */
int cmb_gemm_tn(const cmb_gemm_desc* d, void* stream);
/* Row-wise fp8 quantisation for cmb_gemm(CMB_FP8_E4M3): for every row r of x [rows, K] (dtype CMB_BF16 | CMB_F32, row
 * stride ldx elements):  amax = max|x[r,:]|;  s = 448 / amax (1 if amax == 0);
 * q[r,k] = e4m3fn_rne(clamp(x[r,k] * s, -448, 448));  inv_scale[r] = amax / 448 (1 if amax == 0).
 * One pass over x (a row lives in registers between the reduction and the cast).  K % 16 == 0, ldq % 16 == 0. */
int cmb_quantize_fp8_rows(int dtype, const void* x, int64_t ldx, int64_t rows, int64_t K, void* q, int64_t ldq,
                          float* inv_scale, void* stream);
/* block tile (128 or 256) cmb_gemm would run for this problem — lets the caller label profiles / rooflines per
 * kernel configuration.  dtype CMB_F32 always answers 128. */
int cmb_gemm_tile(int dtype, int64_t M, int64_t N, int32_t split_k, int32_t tile_hint);
/* which kernel the calling thread's most recent cmb_gemm launched (0 before the first): 128 = 128x128 tile kernel,
 * 256 = 8-wave 256x256 kernel (gemm256.hip), 2590 = 4-wave register-buffered 256x256 kernel (gemm_nt_p5_kernel,
 * gemm_p5.hip), 1281 = cmb_gemm_tn's 128x128 kernel (gemm_tn.hip), 64 = the batched K = 64 kernel (gemm_k64.hip), 32 = the
 * M <= 32 kernel (gemm_smallm.hip).  For labelling profiles and rooflines per kernel. */
int cmb_gemm_last_kernel(void);
/* Per-shape dispatch policy: bf16 launches of exactly (M, N, K, act) without a tile_hint and without split-K take
 * `kernel` (128 | 2560 = 8-wave 256x256 | 2590 = 4-wave 256x256; 0 removes the entry) instead of the built-in cost
 * model.  Meant for a start-up calibration on the device at hand (the candidates give bit-identical results; which is
 * faster for a shape varies from box to box — VERDICT r2 #4).  At most 64 entries (CMB_ERR_WORKSPACE beyond); not
 * thread-safe against concurrent cmb_gemm calls: set it between steps. */
int cmb_gemm_policy_set(int64_t M, int64_t N, int64_t K, int32_t act, int32_t kernel);
/* Tail split of bf16 launches without tile_hint / split-K (gemm.hip "Tail split"): when an M x N problem's 256 x 256 tile
 * count is a little more than a whole number of rounds of the device's CUs, rows [0, m1) run on the 256-tile kernel and
 * rows [m1, M) on the 128 x 128 kernel, two launches inside one cmb_gemm call.  Returns m1 (0 = launched whole).
 * CMB_GEMM_NO_TAIL_SPLIT=1 in the environment disables it (A/B runs). */
int64_t cmb_gemm_tail_rows(int64_t M, int64_t N);
int cmb_gemm_policy_clear(void);

/* out[C, R_pad] = in[R, C]^T, zero-filling columns R..R_pad-1 (R_pad >= R). Used to put the
 * reduction dimension innermost for weight-gradient GEMMs (autograd of the linears above). */
int cmb_transpose(int dtype, const void* in, int64_t R, int64_t C, int64_t ld_in,
                  void* out, int64_t R_pad, void* stream);

/* out[c] (fp32) += sum_r in[r, c]   — bias gradients (atomic accumulate; caller zero-fills). */
int cmb_colsum(int dtype, const void* in, int64_t R, int64_t C, int64_t ld_in,
               float* out, void* stream);
/* out[c] += sum_r row_scale[r * ld_scale + c / group] * in[r, c]  (fp32 accumulation; out must be zero-filled by the caller;
 * group % 8 == 0, C % group == 0).  The bias gradients of the absorbed SVA projections: d b_k[c] = sum_q d(cb)[q, h(c)] q[q, c],
 * d b_v[c] = sum_q m3[q, h(c)] d out[q, c] with group = 64 = one head (vision_sampler.py:187-189's k_proj / v_proj biases,
 * restated per head: DESIGN.md section 4.5) — one pass over the rows instead of a broadcast product + reduction. */
int cmb_colsum_scaled(int dtype, const void* in, int64_t R, int64_t C, int64_t ld_in, const float* row_scale,
                      int64_t ld_scale, int32_t group, float* out, void* stream);

/* y = (T)x elementwise casts between fp32 and bf16 (n elements). to_dtype/from_dtype are CMB_*. */
int cmb_cast(int from_dtype, const void* in, int to_dtype, void* out, int64_t n, void* stream);

/* Weight preparation: bf16 copy [rows, cols] (dst, row stride cols) and transposed bf16 copy [cols, rows_pad] (dst_t, zero
 * columns rows .. rows_pad - 1) of a 2-D weight (fp32 master or bf16; row stride ld_src) — what the linears of
 * vision_sampler.py:170-175,254-259 and cambrian_arch.py:49-56 need per step in bf16 compute: W for y = x W^T, W^T for
 * dx = g W.  cmb_weight_prep takes a table of jobs in DEVICE memory (tile0 = prefix sum of cmb_weight_prep_tiles over the
 * jobs before it) and covers every job in ONE launch; cmb_weight_prep_one takes one job by value (host memory).  Either
 * output pointer may be NULL.  cols, rows_pad, ld_src multiples of 8; 16-byte aligned pointers. */
typedef struct cmb_prep_job {
  const void* src;
  void* dst;
  void* dst_t;
  int64_t ld_src;
  int32_t src_dtype; /* CMB_F32 | CMB_BF16 */
  int32_t rows, cols, rows_pad;
  int32_t tile0;
  int32_t reserved;
} cmb_prep_job;
int64_t cmb_weight_prep_tiles(int64_t rows_pad, int64_t cols);
int cmb_weight_prep_one(const cmb_prep_job* job, void* stream);
int cmb_weight_prep(const cmb_prep_job* jobs_device, int32_t n_jobs, int64_t total_tiles, void* stream);

/* Row statistics of a LayerNorm without its output: mean[r], rstd[r] = 1 / sqrt(var + eps) of x [rows, D] (row stride ldx),
 * the two-pass fp32 arithmetic of cmb_layernorm_fwd — the operands of cmb_gemm_desc.row_mean / row_rstd.  D % 8 == 0, D <= 4096. */
int cmb_row_stats(int dtype, const void* x, int64_t rows, int64_t D, int64_t ldx, float eps, float* mean, float* rstd, void* stream);

/* ------------------------------------------------------------------------------------------
 * LayerNorm   y[r,:] = (x[r,:] (+ add[pos(r),:]) - mean) * rstd (* gamma + beta)
 * replaces: nn.LayerNorm at vision_sampler.py:170,173-174,261 and cambrian_arch.py:56, the
 *   pos_embed add at vision_sampler.py:304-309, and the HF/timm LayerNorms of the towers.
 * add: optional [grid_r*grid_r, D] fp32 table (SVA pos_embed_i); the row -> table-row map is
 *   the window-local position of token r on a square side×side token grid:
 *   pos = ((r % (side*side)) / side % grid_r) * grid_r + (r % side) % grid_r.
 * gamma/beta may be NULL (pure normalisation: the SVA path folds the K- and V-LayerNorm affines
 *   into the projection weights so that one normalised tensor serves both).
 * mean/rstd: optional fp32 [rows] outputs for the backward.
 * ---------------------------------------------------------------------------------------- */
int cmb_layernorm_fwd(int dtype, const void* x, int64_t rows, int64_t D, int64_t ldx,
                      const float* add, int32_t side, int32_t grid_r,
                      const float* gamma, const float* beta, float eps,
                      void* y, int64_t ldy, float* mean, float* rstd, void* stream);

/* dx (+)= LN backward.  dy: [rows,D]; xhat is recomputed from x (+add), mean, rstd.
 * dx_accumulate != 0: dx += result (gradient accumulation across the 13 SVA layers that share
 * the aux features, SURVEY.md §7 "hard parts").  dgamma/dbeta/dadd are fp32 and are
 * *accumulated into* (atomics), caller zero-fills; any may be NULL. */
int cmb_layernorm_bwd(int dtype, const void* dy, int64_t lddy, const void* x, int64_t ldx,
                      int64_t rows, int64_t D,
                      const float* add, int32_t side, int32_t grid_r,
                      const float* gamma, const float* mean, const float* rstd,
                      void* dx, int64_t lddx, int32_t dx_accumulate,
                      float* dgamma, float* dbeta, float* dadd, void* stream);

/* Backward of SEVERAL non-affine LayerNorms of one input in one pass (the 13 SVA layers each normalise the same aux feature
 * tensor with their own position table: vision_sampler.py:304-309 reached 13 times per step, cambrian_llama.py:168-207):
 *   dx[r,:] (+)= sum_l  rstd_l[r] * ( dy_l[r,:] - mean(dy_l[r,:]) - xh_l[r,:] * mean(dy_l[r,:] * xh_l[r,:]) ),
 *   xh_l[r,:] = (x[r,:] + add_l[pos(r),:] - mean_l[r]) * rstd_l[r],      dadd_l[pos,:] += the layer's term summed over rows
 * with mean_l / rstd_l as cmb_layernorm_fwd returned them (add = add_l, no gamma / beta) and pos(r) the window position
 * of row r on a side x side grid in grid_r x grid_r windows (as cmb_layernorm_fwd).  x is read once, every dy_l once
 * (dense rows: leading dimension D), dx (fp32, leading dimension lddx) is written once (`accumulate` != 0: added to).
 * add[l] may be NULL (no table for that layer; then dadd[l] is ignored); dadd[l] may be NULL (gradient not wanted);
 * caller zero-fills dadd.  layers <= CMB_LN_MULTI_MAX (run as launches of up to 7 layers: x is read once per launch),
 * D <= 1024. */
#define CMB_LN_MULTI_MAX 16
typedef struct cmb_ln_multi_desc {
  int32_t dtype;          /* CMB_BF16 | CMB_F32: element type of x and dy_l */
  int32_t layers;
  const void* x;  int64_t ldx;
  int64_t rows, D;
  int32_t side, grid_r;
  const void* dy[CMB_LN_MULTI_MAX];
  const float* add[CMB_LN_MULTI_MAX];
  const float* mean[CMB_LN_MULTI_MAX];
  const float* rstd[CMB_LN_MULTI_MAX];
  float* dadd[CMB_LN_MULTI_MAX];
  float* dx;      int64_t lddx;
  int32_t accumulate;
  int32_t reserved;
  void* dx_out;           /* optional (round 6): the call's LAST launch writes the finished sum here in x's dtype (dense rows,
                           * leading dimension D) instead of updating dx — the cast of the fp32 accumulator at the end of the
                           * deferred backward; dx then holds the sum without the last launch's layers */
} cmb_ln_multi_desc;
int cmb_layernorm_bwd_multi(const cmb_ln_multi_desc* d, void* stream);

/* Forward of SEVERAL non-affine LayerNorms of one input in one pass (round 6): y_l[r,:] = normalise(x[r,:] + add_l[pos(r),:]),
 * l < layers — the 13 SVA layers' normalisations of the windowed tower's tokens (vision_sampler.py:304-309 reached 13 times per
 * step on the SAME aux features; only the position table differs).  x is read once instead of `layers` times: 2 + 2 layers
 * instead of 4 layers bytes per element.  Every y_l / mean_l / rstd_l equals, bit for bit, what cmb_layernorm_fwd(add = add_l,
 * gamma = beta = NULL) returns.  add[l] may be NULL; y_l dense rows (leading dimension D); mean[l] / rstd[l] fp32 [rows],
 * required.  layers <= CMB_LN_MULTI_MAX, D <= 1024 (D % 8 == 0). */
typedef struct cmb_ln_fwd_multi_desc {
  int32_t dtype;          /* CMB_BF16 | CMB_F32: element type of x and y_l */
  int32_t layers;
  const void* x;  int64_t ldx;
  int64_t rows, D;
  int32_t side, grid_r;
  float eps;  int32_t reserved;
  const float* add[CMB_LN_MULTI_MAX];
  void* y[CMB_LN_MULTI_MAX];
  float* mean[CMB_LN_MULTI_MAX];
  float* rstd[CMB_LN_MULTI_MAX];
} cmb_ln_fwd_multi_desc;
int cmb_layernorm_fwd_multi(const cmb_ln_fwd_multi_desc* d, void* stream);

/* RMSNorm: y = (x * rsqrt(mean(x^2)+eps)) * w, fp32 math, weight multiplied BEFORE the down-cast
 * (the reference's patched LlamaRMSNorm: train_fsdp.py:1429-1438; phi3/modeling_phi3.py:83-97). */
int cmb_rmsnorm_fwd(int dtype, const void* x, int64_t rows, int64_t D, const float* w, float eps,
                    void* y, float* rstd, void* stream);
int cmb_rmsnorm_bwd(int dtype, const void* dy, const void* x, int64_t rows, int64_t D,
                    const float* w, const float* rstd, void* dx, float* dw, void* stream);

/* Fused residual add + RMSNorm of the decoder layer ("h = h + attn(...); x = post_attention_layernorm(h)"):
 * sum = x + res (stored in the compute dtype), y = rmsnorm(sum) * w, one pass; rstd fp32 [rows] for the backward. */
int cmb_add_rmsnorm_fwd(int dtype, const void* x, const void* res, int64_t rows, int64_t D, const float* w, float eps,
                        void* sum, void* y, float* rstd, void* stream);
/* dx = rmsnorm_backward(dy; x, w, rstd) + dadd (dadd may be NULL), single pass, frozen weight (no dw): the gradient of
 * the residual stream and of the normalised branch leave as one tensor. */
int cmb_rmsnorm_bwd_add(int dtype, const void* dy, const void* x, const void* dadd, int64_t rows, int64_t D,
                        const float* w, const float* rstd, void* dx, void* stream);

/* RoPE (rotate-half form): x[t,h,:] = x*cos + rotate_half(x)*sin with
 * cos/sin = cos/sin(position_ids[t] * base^(-2i/Dh)) evaluated in fp32
 * (phi3/modeling_phi3.py:114-141,257-281 and the HF Llama equivalent).
 * cmb_rope_table builds the [ntok, Dh/2] fp32 tables once per forward; cmb_rope_apply rotates
 * x [ntok, H, Dh] (token stride row_stride elements) in place; inverse != 0 applies the transposed
 * rotation (the backward). */
int cmb_rope_table(const int64_t* position_ids, int64_t ntok, int64_t Dh, float base,
                   float* cos_t, float* sin_t, void* stream);
int cmb_rope_apply(int dtype, void* x, const float* cos_t, const float* sin_t, int64_t ntok,
                   int64_t H, int64_t Dh, int64_t row_stride, int32_t inverse, void* stream);

/* ------------------------------------------------------------------------------------------
 * SVA windowed cross-attention core  (vision_sampler.py:193-230, F.scaled_dot_product_attention
 * at :215-220, with the window partition of cambrian_arch.py:271-287 done by index arithmetic).
 *   q   : [Bq, heads*hd]            Bq = B * qside*qside queries, one query row each
 *   kv_i: [B * (qside*r_i)^2, 2*heads*hd]  tower-token-major (NOT window-rearranged);
 *         K = columns [0, heads*hd), V = columns [heads*hd, 2*heads*hd)
 *   mask_i: uint8 [Bq, r_i*r_i] (1 = may attend), may be NULL (= all ones)
 *   out : [Bq, heads*hd];  lse: fp32 [Bq, heads] (log-sum-exp of the scaled scores)
 * query (b, qy, qx) attends, for every tower i, to tokens ((qy*r_i+ry)*qside*r_i + qx*r_i+rx).
 * Softmax in fp32 over the concatenation of all towers' keys, scale = 1/sqrt(hd).
 * ---------------------------------------------------------------------------------------- */
#define CMB_SVA_MAX_TOWERS 8
typedef struct cmb_sva_desc {
  int32_t dtype;
  int32_t B, qside, heads, hd, ntowers;
  int32_t window_major;  /* 0: kv_i tower-token-major (below); 1: kv_i is [Bq, r_i*r_i, 2*heads*hd] */
  int32_t r[CMB_SVA_MAX_TOWERS];
  const void* q;   int64_t ldq;
  const void* kv[CMB_SVA_MAX_TOWERS]; int64_t ldkv[CMB_SVA_MAX_TOWERS];
  const uint8_t* mask[CMB_SVA_MAX_TOWERS];
  void* out;       int64_t ldo;
  float* lse;
  /* backward only */
  const void* dout; int64_t lddo;
  void* dq;         int64_t lddq;
  void* dkv[CMB_SVA_MAX_TOWERS];   /* same layout as kv; every element written exactly once */
} cmb_sva_desc;

int cmb_sva_attn_fwd(const cmb_sva_desc* d, void* stream);
int cmb_sva_attn_bwd(const cmb_sva_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------
 * SVA cross-attention core with ONE windowed tower's K / V projections absorbed into the query side
 * (same reference lines as above: vision_sampler.py:187-230; heads = 16, hd = 64, feature width 1024; bf16 on the MFMA, or
 * fp32 through an exact instantiation of the same algorithm — `dtype`).
 * Every token of an s x s-window tower is seen by exactly one query, so instead of projecting K and V per token
 * (kv_i above) the caller supplies, for that tower,
 *   xhat : [B * (qside*ra)^2, 1024]  the LayerNorm-normalised tokens (affines folded into W_k, W_v, b_k, b_v)
 *   U    : [Bq, 16, 1024]            U[q,h,:] = W_k,h^T q_h           (cmb_gemm, batch = 16, K = 64)
 *   bk, bv : fp32 [1024]             the folded K / V biases: the kernel forms cb[q,h] = b_k,h . q_h itself
 * and receives, besides the joint-softmax probabilities P (fp32 [Bq, 16, 20]: columns [0, 4) the direct towers' keys,
 * [4, 20) the absorbed tower's window tokens, zero where absent or masked; saved for the backward),
 *   out  : [Bq, 1024]   sum over the DIRECT towers' keys of p * V, plus m3[q,h] b_v,h   (kv / mask / r as in cmb_sva_desc; r_i == 1)
 *   xbar : [Bq, 16, 1024]  Xb[q,h,:] = sum_t p[q,h,t] xhat_t     -> the layer's attention output is out + W_v,h Xb
 *   m3   : fp32 [Bq, 16]   sum_t p[q,h,t] over the absorbed tower's tokens (kept by the caller: d b_v = sum_q m3 d out)
 * with score[q,h,t] = (xhat_t . U[q,h,:] + cb[q,h]) / sqrt(hd) for the absorbed tokens and q_h . K_h / sqrt(hd) for the
 * direct ones.  The backward takes d(out), d(xbar) and writes dq (the direct towers' part and d(cb) b_k), dkv of the direct
 * towers, dU, d(cb) (fp32 [Bq, 16]: d b_k = sum_q d(cb) q) and d(xhat) (every element exactly once).  Neither K|V nor
 * dK|dV of the absorbed tower exist.
 * ---------------------------------------------------------------------------------------- */
typedef struct cmb_sva_abs_desc {
  int32_t B, qside, heads, hd;
  int32_t ntowers;       /* directly projected towers (may be 0), each with r_i == 1 */
  int32_t window_major;  /* layout of xhat: 0 tower-token-major, 1 [Bq, ra*ra, 1024] */
  int32_t r[CMB_SVA_MAX_TOWERS];
  const void* q;   int64_t ldq;
  const void* kv[CMB_SVA_MAX_TOWERS]; int64_t ldkv[CMB_SVA_MAX_TOWERS];
  const uint8_t* mask[CMB_SVA_MAX_TOWERS];
  int32_t ra;            /* window side of the absorbed tower, ra*ra <= 16 */
  int32_t dtype;         /* CMB_BF16 (= 0: the MFMA kernels) | CMB_F32 (the exact instantiation of the same algorithm: plain fp32
                            arithmetic, test speed): element type of q, kv, xhat, U, out, xbar and of every gradient */
  const void* xhat; int64_t ldx;
  const uint8_t* mask_a; /* uint8 [Bq, ra*ra] or NULL */
  const void* U;
  const float* bk;
  void* out;       int64_t ldo;
  void* xbar;
  float* m3;
  float* P;
  /* backward only */
  const void* dout; int64_t lddo;
  const void* dxbar;
  const float* bv;       /* (forward AND backward: see bk, bv above) */
  void* dq;         int64_t lddq;
  void* dkv[CMB_SVA_MAX_TOWERS];
  void* dU;
  float* dcb;
  void* dxhat;      int64_t lddx;
} cmb_sva_abs_desc;

int cmb_sva_abs_fwd(const cmb_sva_abs_desc* d, void* stream);
int cmb_sva_abs_bwd(const cmb_sva_abs_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------
 * Embedding merge (static layout): cambrian_arch.py:413-420 (newline column) + :457-490.
 *   out[b,t,:] = feat[b, i*side+j, :]   if t = p_b + i*(side+1) + j, j < side
 *              = newline[:]             if t = p_b + i*(side+1) + side
 *              = table[ids'[b,t], :]    otherwise (ids' = ids with image_token -> 0)
 * p_b = first index with ids[b,t] == image_token (rows without one are pure text).
 * Copies only: bit-exact.  `pos` (int32 [B]) receives p_b (or -1).
 * ---------------------------------------------------------------------------------------- */
int cmb_embed_splice_fwd(int dtype, const int64_t* ids, int64_t B, int64_t S, int64_t H,
                         int64_t image_token, const void* table, int64_t vocab,
                         const void* feat, int32_t side, const void* newline,
                         void* out, int32_t* pos, void* stream);
/* dfeat[b,i*side+j,:] = dout[b, p_b+i*(side+1)+j, :];  dnewline (fp32 [H], accumulated) +=
 * sum_{b,i} dout[b, p_b+i*(side+1)+side, :]. */
int cmb_embed_splice_bwd(int dtype, const void* dout, const int32_t* pos, int64_t B, int64_t S,
                         int64_t H, int32_t side, void* dfeat, float* dnewline, void* stream);

/* K|V weight folding of one tower of one SVA layer (vision_sampler.py:173-174,188-189: k_proj_i / v_proj_i = LayerNorm ->
 * Linear; the two LayerNorms share their statistics, so their affines are folded into one projection):
 *   w_out[n,:] = W[n,:] * gamma,  b_out[n] = W[n,:] . beta;  rows 0..H-1 from (wk, gk, bk), rows H..2H-1 from (wv, gv, bv).
 * All fp32 (master parameters); wk / wv [H, K] row-major, K % 4 == 0.  bwd: given d(w_out) [2H, K] and d(b_out) [2H]
 * writes dW = dw * gamma + db (x) beta, dgamma = colsum(dw o W), dbeta = W^T db for both halves.  Deterministic (two
 * passes through a caller-owned workspace of cmb_sva_fold_kv_bwd_workspace(H, K) bytes; CMB_ERR_WORKSPACE if smaller). */
int cmb_sva_fold_kv_fwd(const float* wk, const float* gk, const float* bk, const float* wv, const float* gv, const float* bv,
                        int64_t H, int64_t K, float* w_out, float* b_out, void* stream);
int cmb_sva_fold_kv_bwd(const float* dw_out, const float* db_out, const float* wk, const float* gk, const float* bk,
                        const float* wv, const float* gv, const float* bv, int64_t H, int64_t K, float* dwk, float* dgk,
                        float* dbk, float* dwv, float* dgv, float* dbv, void* workspace, int64_t workspace_bytes,
                        void* stream);
int64_t cmb_sva_fold_kv_bwd_workspace(int64_t H, int64_t K);

/* mean over tokens: out[b,:] = mean_t x[b,t,:]  (cambrian_arch.py:377); bwd broadcasts. */
int cmb_token_mean_fwd(int dtype, const void* x, int64_t B, int64_t T, int64_t D, void* out,
                       void* stream);
/* acc[b,t,:] (fp32) += g[b,:] / T  — backward of the mean, into the fp32 aux-feature accumulator. */
int cmb_token_mean_bwd(int dtype, const void* g, int64_t B, int64_t T, int64_t D, float* acc,
                       void* stream);

/* ------------------------------------------------------------------------------------------
 * Vision-tower kernels (frozen towers: forward only; SURVEY.md §8a T1–T4)
 * ---------------------------------------------------------------------------------------- */
/* Non-causal multi-head self-attention over packed QKV rows (HF CLIPAttention / Dinov2 /
 * timm Attention reached from clip_encoder.py:104, dino_encoder.py:159, siglip_encoder.py:97).
 * qkv: [B*N, 3*heads*hd] with q|k|v column blocks; out: [B*N, heads*hd]. hd in {64, 96}
 * (SigLIP's 72 is zero-padded to 96 by the weight packer). scale is applied to the scores.
 * bf16 with hd in {64,96} runs the MFMA flash kernel; fp32 (and force_simple != 0) runs the
 * one-wave-per-query VALU kernel used as the exact-fp32 parity path. */
int cmb_vit_attn_fwd(int dtype, const void* qkv, int64_t B, int64_t N, int32_t heads, int32_t hd,
                     float scale, void* out, int32_t force_simple, void* stream);

/* Patch gather for a stride==kernel conv (ViT patch-embed 14x14/14, ConvNeXt stem 4x4/4):
 * img NCHW [B,C,H,W] (fp32 or bf16) -> cols [B*(H/p)*(W/p), Kpad], column order (c, dy, dx),
 * zero padded to Kpad. */
int cmb_patchify_nchw(int in_dtype, const void* img, int64_t B, int64_t C, int64_t H, int64_t W,
                      int32_t p, int out_dtype, void* cols, int64_t Kpad, void* stream);
/* 2x2/2 patch gather on an NHWC map (ConvNeXt downsample): x [B,H,W,C] -> [B*(H/2)*(W/2), 4C],
 * column order (dy, dx, c). */
int cmb_patchify2x2_nhwc(int dtype, const void* x, int64_t B, int64_t H, int64_t W, int64_t C,
                         void* cols, void* stream);
/* Depthwise 7x7, pad 3, NHWC (timm ConvNeXtBlock.conv_dw): w [49, C] fp32 (tap-major), bias [C]. */
int cmb_dwconv7x7_nhwc(int dtype, const void* x, int64_t B, int64_t H, int64_t W, int64_t C,
                       const float* w, const float* bias, void* y, void* stream);
/* Weight gradient of that depthwise 7x7 (towers that train): every one of `slots` workgroup rows
 * writes the partial sum over the 8x8 tiles it walked to partial[slot][49][C] (fp32); the caller column-sums the slots
 * (cmb_colsum on [slots, 49*C]).  dX of the same layer is cmb_dwconv7x7_nhwc on dY with the taps reversed and a zero
 * bias; d(bias) is a column sum of dY.  C % 64 == 0. */
int cmb_dwconv7x7_wgrad(int dtype, const void* x, const void* dy, int64_t B, int64_t H, int64_t W, int64_t C,
                        float* partial, int32_t slots, void* stream);
/* Bilinear resample (align_corners=False, fp32 lerp) of a token grid, channels-last:
 * in [B, Hi*Wi, C] (row stride ld_in) -> out [B, Ho*Wo, ...] written at column offset into rows
 * of stride ld_out (so the 4 ConvNeXt stage maps land in one [B,9216,5760] buffer).
 * replaces F.interpolate at clip_encoder.py:83-88, siglip_encoder.py:80-85, dino_encoder.py:141-146,
 * clip_convnext_encoder.py:112-118 and the permutes around them. */
int cmb_resample_bilinear(int dtype, const void* in, int64_t B, int32_t Hi, int32_t Wi, int64_t C,
                          int64_t ld_in, int64_t batch_stride_in, void* out, int32_t Ho, int32_t Wo,
                          int64_t ld_out, int64_t batch_stride_out, void* stream);
/* Elementwise y = act(a) * b (SwiGLU: act = SiLU) / y = act(a); a,b,y [rows, D] with strides. */
int cmb_act_mul(int dtype, int32_t act, const void* a, int64_t lda, const void* b, int64_t ldb,
                int64_t rows, int64_t D, void* y, int64_t ldy, void* stream);
/* dx = dy * act'(pre)  (GELU backward for the SVA / projector MLPs). */
int cmb_act_bwd(int dtype, int32_t act, const void* dy, const void* pre, int64_t n, void* dx,
                void* stream);
/* dst[dst_map(r), 0:D] = src[src_map(r), 0:D] for r in [0,rows); src == NULL zero-fills.  Row gather /
 * scatter of the in-LLM SVA hook: hidden[:, p:p+600].view(B,24,25,H)[:, :, :24] <-> [B*576, H]
 * (cambrian_llama.py:181-207) and its backward. */
int cmb_copy_rows(int dtype, const void* src, const cmb_rowmap* src_map, void* dst,
                  const cmb_rowmap* dst_map, int64_t rows, int64_t D, void* stream);

/* dst[r*ld : r*ld+D] = src[0:D] for r in [0,nrows) — CLS-token rows of the ViT sequence buffers
 * (the position-embedding add itself rides in the patch-embed GEMM epilogue as a row-mapped residual). */
int cmb_bcast_rows(int dtype, void* dst, int64_t ld, int64_t nrows, int64_t D, const void* src,
                   void* stream);

/* ------------------------------------------------------------------------------------------
 * LLM-side HBM-bound kernels (llm_ops.hip)
 * ---------------------------------------------------------------------------------------- */
/* Shifted cross-entropy of cambrian_llama.py:411-422 without materialising logits.float():
 * logits [rows, V] (bf16 or fp32, row stride ld), labels int64 [rows] ALREADY shifted by the caller
 * (labels_shift[b, t] = labels[b, t+1], last position = ignore_index).  fwd: lse[row] = log sum_v exp(x_v) in fp32,
 * loss[row] = lse - x[label] (0 for ignored rows); the caller reduces mean over the non-ignored rows.
 * bwd: dlogits[row, v] = (exp(x_v - lse[row]) - [v == label]) * scale[0], zeros for ignored rows; dlogits may alias
 * logits (in-place).  Rows that are not 16-byte aligned (odd vocabularies) take a scalar path. */
int cmb_cross_entropy_fwd(int dtype, const void* logits, int64_t rows, int64_t V, int64_t ld,
                          const int64_t* labels, int64_t ignore_index, float* lse, float* loss, void* stream);
int cmb_cross_entropy_bwd(int dtype, const void* logits, int64_t rows, int64_t V, int64_t ld,
                          const int64_t* labels, int64_t ignore_index, const float* lse, const float* scale,
                          void* dlogits, int64_t ldd, void* stream);
/* Fused QKV hand-off of a decoder layer: packed [B*S, (nh + 2*nkv)*Dh] (q heads | k heads | v heads per token)
 * -> q [B,S,nh,Dh], k [B,S,nkv,Dh], v [B,S,nkv,Dh] (token-major) with RoPE (tables of cmb_rope_table) on q and k; merge != 0 is the
 * backward: (dq, dk, dv) -> d(packed) with the transposed rotation.  Replaces the q/k/v views, two rotary passes and
 * the transposes in front of the attention (HF LlamaAttention reached from cambrian_llama.py:157-166). Dh % 16 == 0. */
int cmb_qkv_rope(int dtype, int32_t merge, void* packed, const float* cos_t, const float* sin_t, int64_t B, int64_t S,
                 int32_t nh, int32_t nkv, int32_t Dh, void* q, void* k, void* v, void* stream);
/* Backward of causal self-attention with grouped KV heads (the decoder's F.scaled_dot_product_attention(is_causal=True,
 * enable_gqa=True); HF LlamaAttention reached from cambrian_llama.py:157-166).  bf16, head_dim 128, S % 128 == 0.
 * q / o / dout / dq are [B,S,H,128] and k / v / dk / dv [B,S,HKV,128] through (batch, token, head) element strides;
 * lse fp32 [B,H,S] = log sum_j exp(scale * q.k_j) from the forward; dvec fp32 [2,B,H,S] is scratch (ABI 6: [0] receives rowsum(dO*O), [1] lse * log2 e for the
 * LDS-DMA dK/dV kernel).
 * Two MFMA kernels (dQ; dK+dV), no atomics, bit-reproducible. */
/* Forward of the same attention: out [B,S,H,128] (strides of q), lse fp32 [B,H,S].
 * causal == 0 is the bidirectional form used when the vision towers train (HF CLIPAttention /
 * Dinov2SelfAttention / timm Attention reached from clip_encoder.py:104, dino_encoder.py:159, siglip_encoder.py:97):
 * every query sees keys [0, kv_len); rows [kv_len, S) are padding (S = kv_len rounded up to 128, zero-filled by the
 * caller, head_dim zero-padded to 128) whose outputs / gradients are don't-care / zero.
 * key_valid (causal only; may be NULL): the collator's key-padding mask (train_fsdp.py:1057-1085, 1089-1165 — padded rows /
 * columns of the 24 x 25 visual span, padded sequence tail), [B, S] bytes, non-zero = attendable.  Query q sees key k iff
 * k <= q and (key_valid[b][k] or k == q): HF's causal AND padding mask (cambrian_llama.py:142-166) with the diagonal
 * kept open so a padded query row is never empty.  Tiles made of padding only are skipped. */
int cmb_flash_attn_fwd(const void* q, const void* k, const void* v, int64_t B, int64_t S, int32_t H, int32_t HKV,
                       int32_t hd, int64_t q_sb, int64_t q_ss, int64_t q_sh, int64_t kv_sb, int64_t kv_ss, int64_t kv_sh,
                       float scale, int32_t causal, int64_t kv_len, const uint8_t* key_valid, void* out, float* lse,
                       void* stream);
int cmb_flash_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                       int64_t B, int64_t S, int32_t H, int32_t HKV, int32_t hd,
                       int64_t q_sb, int64_t q_ss, int64_t q_sh, int64_t kv_sb, int64_t kv_ss, int64_t kv_sh,
                       float scale, int32_t causal, int64_t kv_len, const uint8_t* key_valid, float* dvec, void* dq, void* dk,
                       void* dv, void* stream);
/* Backward of h = silu(g) * u (Llama MLP gate; forward is cmb_act_mul with CMB_ACT_SILU):
 * dg = dh * u * silu'(g), du = dh * silu(g); all [rows, D] with row strides. */
int cmb_swiglu_bwd(int dtype, const void* dh, int64_t lddh, const void* g, int64_t ldg, const void* u, int64_t ldu,
                   int64_t rows, int64_t D, void* dg, int64_t lddg, void* du, int64_t lddu, void* stream);

/* ------------------------------------------------------------------------------------------
 * Image pre-processing on the step boundary
 * replaces, per sample and per tower:  train_fsdp.py:985-1008 / mm_utils.py:153-165,183-201
 *     image_aux = expand2square(image, tuple(int(x*255) for x in processor.image_mean)).resize((R, R))
 *     image_aux = processor.preprocess(image_aux, return_tensors='pt')['pixel_values'][0]
 * i.e. letter-box to a square with the tower's mean colour, Pillow's two-pass fixed-point bicubic
 * resample (libImaging/Resample.c, 22-bit coefficients, uint8 intermediate), then the processor's pointwise
 * rescale + normalise, emitted planar [3,R,R].  Integer work: results are bit-identical to Pillow's; the
 * pointwise stage is a caller-built table lut[c][level] (fp32), so it reproduces whichever processor flavour
 * (HF rescale/normalize or torchvision ToTensor/Normalize) the tower uses exactly.
 * One launch pair handles a whole batch x all towers through a job table.
 * ---------------------------------------------------------------------------------------- */
enum { CMB_F16 = 2 };          /* extra output element type of cmb_image_preprocess only */

typedef struct cmb_image_job {
  int64_t src_off;     /* byte offset of this sample's uint8 [h, w, 3] pixels in `src` */
  int64_t tmp_off;     /* byte offset (multiple of 4) of this job's scratch planes [3, side, pitch] in `tmp`,
                          pitch = (out_side + 3) & ~3; unused when ksize == 0 */
  int64_t dst_off;     /* element offset of this job's [3, out_side, out_side] output in `dst` */
  int32_t w, h;        /* source size */
  int32_t side;        /* max(w, h): the letter-boxed square (mm_utils.py:153-165) */
  int32_t off_x, off_y;/* where the source sits inside the square: ((side-w)/2, 0) or (0, (side-h)/2) */
  int32_t out_side;    /* R */
  int32_t ksize;       /* taps per output; 0 = side == out_side, Image.resize returns a copy */
  int32_t coef_off;    /* int32 offset of this job's tap-major coefficients [ksize, out_side] in `coefs` */
  int32_t bounds_off;  /* int32 offset of this job's [out_side, 2] (first, count) in `bounds` */
  int32_t lut_off;     /* element offset of this tower's fp32 [3, 256] table in `lut` */
  uint32_t background; /* r | g << 8 | b << 16 */
  int32_t reserved;
} cmb_image_job;

/* HOST function (no device work): taps of Pillow's bicubic resample of `in_size` -> `out_size` samples over the
 * full box.  Returns ksize = 2*ceil(2*max(in/out,1)) + 1, or a negative status.  With non-null outputs fills
 * bounds [out_size, 2] = (first, count) and coefs [ksize, out_size] (tap-major, 22-bit fixed point, zero beyond
 * count).  Follows precompute_coeffs + normalize_coeffs_8bpc; double precision, no contraction. */
int cmb_resize_coeffs(int32_t in_size, int32_t out_size, int32_t* bounds, int32_t* coefs);
/* Runs the horizontal pass (src -> tmp) and the vertical + table pass (tmp -> dst) for n_jobs jobs.
 * jobs_dev / jobs_host are the same table in device and host memory (the host copy sizes the grids).
 * out_dtype: CMB_BF16 | CMB_F32 | CMB_F16 (table value rounded to nearest even). */
int cmb_image_preprocess(const cmb_image_job* jobs_dev, const cmb_image_job* jobs_host, int32_t n_jobs,
                         const uint8_t* src, const int32_t* bounds, const int32_t* coefs, const float* lut,
                         int32_t out_dtype, uint8_t* tmp, void* dst, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CAMBRIAN_AMD_H */
