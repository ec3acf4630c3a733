/* apla_hip.h — C-ABI of libapla_hip.so: the MI355X (gfx950) kernels behind the APLA fine-tuning hot path.
 *
 * Conventions:
 *   - every entry point is `extern "C" int fn(..., hipStream_t stream)`; 0 = OK, negative = error
 *     (APLA_EINVAL shape/alignment, APLA_ENOSYS unsupported configuration, APLA_EIO HIP launch error);
 *     a human-readable message for the calling thread is available from apla_last_error().
 *   - stateless and re-entrant: no allocation, no synchronisation, no global state (kernel-schedule choices used by the
 *     tests and the A/B benchmarks are per-call arguments of the *_ex entry points; what the library remembers is per DEVICE
 *     ordinal and idempotent: the CU count and "this kernel's dynamic-LDS limit was raised on this device"); no environment
 *     variable is read outside the diagnostic -DAPLA_ABL_* builds; all buffers (including workspaces) are owned by the caller;
 *     kernels are enqueued on `stream` only (hipGraph-capturable).
 *   - activations / frozen weights are bf16 (raw uint16 storage), trainable masters / statistics / gradients fp32,
 *     index vectors int32 on device.  "res" buffers (the residual stream and its gradient) are fp32 or bf16,
 *     selected by `res_dtype` (APLA_F32 / APLA_BF16).
 *   - matrices are row-major with an explicit leading dimension in ELEMENTS.
 *
 * Each declaration cites the reference interface it replaces (paths relative to /root/reference/src).
 */
#ifndef APLA_HIP_H
#define APLA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP_PLATFORM_AMD__
typedef struct ihipStream_t* hipStream_t;
#endif

enum { APLA_OK = 0, APLA_EIO = -5, APLA_EINVAL = -22, APLA_ENOSYS = -38 };
enum { APLA_BF16 = 0, APLA_F16 = 1, APLA_F32 = 2 };

/* The 16-bit operand type is fixed per build of the library: bf16 (libapla_hip.so, the default and the benchmarked
 * configuration) or IEEE fp16 (libapla_hip_f16.so, compiled with -DAPLA_FP16: the reference's own autocast dtype,
 * trainer.py:121-138, 8x smaller operand rounding).  Same entry points, same layouts, same MFMA rate; every pointer
 * documented as "bf16" below is fp16 in that build, and dtype arguments take APLA_F16 instead of APLA_BF16 (passing the
 * other build's code returns APLA_ENOSYS / APLA_EINVAL).  Returns APLA_BF16 or APLA_F16. */
int apla_operand_dtype(void);

/* GEMM epilogues */
enum {
  APLA_EPI_STORE = 0,     /* C = acc (+bias)                                 -> out_dtype (bf16|f32)            */
  APLA_EPI_GELU = 1,      /* a = acc+bias; C = gelu(a) bf16; aux_out = gelu'(a) bf16 (exact erf GELU)           */
  APLA_EPI_RESIDUAL = 2,  /* C = aux_in + acc (+bias); aux_in/C are residual-stream dtype (may alias)           */
  APLA_EPI_MUL = 3,       /* C = (acc (+bias)) * aux_in; aux_in bf16 [M,N]                      -> bf16         */
  APLA_EPI_SWIGLU = 4,    /* weight rows interleaved (x1_i,x2_i): C[:, i] = silu(x1)*x2 bf16 [M,N/2];
                             aux_out = (acc+bias) bf16 [M,N] (saved for backward)                               */
  APLA_EPI_SWIGLU_BWD = 5, /* acc = dh [M,N]; aux_in = saved x12 interleaved [M,2N];
                             C[:,2i] = dh*x2*silu'(x1), C[:,2i+1] = dh*silu(x1)       -> bf16 [M,2N]            */
  APLA_EPI_GELU_FWD = 6    /* C = gelu(acc+bias) bf16, nothing saved: the no-grad forward (evaluation, EMA teacher)          */
};

const char* apla_last_error(void);
int apla_version(void);

/* C[M,N] = A[M,K] · W[N,K]^T (+bias) with a fused epilogue; bf16 MFMA, fp32 accumulate.
 * Replaces every frozen nn.Linear on the path — qkv (apla/appla_attn.py:53), the merged APLA projection
 * (appla_attn.py:64-79, scatter folded into the natural-order weight), Mlp.fc1/fc2 (utils/transformers/vit.py:152-168),
 * SwiGLU w12/w3 (vit.py:108-149) — and their dX backward (same kernel on the transposed frozen weight).
 * Requires N % 128 == 0, K % 64 == 0, lda/ldw % 8 == 0, 16-byte aligned pointers. */
int apla_gemm_nt(const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc, int M, int N,
                 int K, int epilogue, int out_dtype, const void* aux_in, int ld_aux_in, void* aux_out,
                 int ld_aux_out, hipStream_t stream);

/* apla_gemm_nt with two extra choices packed into `flags`:
 *   bits 0-7   profiling tag: picks one of several identical instantiations of the bf16 STORE kernel, so that a
 *              rocprofv3 kernel trace names the call sites of the step (2 qkv, 3 proj, 4 fc2, 5 dfc1, 6 dproj, 7 dqkv,
 *              8 patch embedding; 0 = untagged).  No effect on results or speed.
 *   bits 8-15  kernel schedule: 0 = auto (what apla_gemm_nt does), for M >= 2048 and N % 256 == 0: the wide 4-wave kernel
 *              (160x256x32 tile, two workgroups per CU, gemm_w4.hip) for STORE with K <= 1024, the 8-wave ping-pong kernel
 *              (320x256x32 tile, one workgroup per CU, gemm_pp2.hip) for STORE with a longer K and for GELU above 40 000 rows, the
 *              tile-alternating kernel (160x256x32 tile, one 8-wave workgroup per CU whose two wave groups swap a compute and a
 *              service role per tile, gemm_tp.hip) for GELU_FWD from 8192 rows (K >= 704, K % 64 == 0); else the 4-wave
 *              persistent kernel (128/160 x 128 x 64 tile, gemm_nt.hip).  9 = ping-pong wherever instantiated; 16 = wide 4-wave
 *              wherever instantiated (16-bit STORE / GELU / GELU_FWD); 17 = tile-alternating wherever instantiated (16-bit
 *              STORE and GELU_FWD, and GELU with both outputs as images: bits 18 + 19); 14 / 15 = 4-wave
 *              persistent kernel with BM 128 / 160; 1 = the simple non-persistent 2-stage kernel (the round-1 starting point;
 *              an in-tree A/B baseline).  All schedules compute the same results (tests/test_kernels_gpu.py).
 *   bit 16     W is given as its K-PANEL IMAGE [K/32][N][32] (apla_pack_k_panels; `ldw` is not read), bit 17 the same for A
 *              ([K/32][M][32], `lda` not read).  The fill path of a CU works in 128-byte lines: an LDS-DMA instruction that
 *              takes 64 bytes (a 32-wide K-step) from each of 16 rows of a row-major operand moves half of what one reading
 *              whole lines does (64 vs 115-129 GB/s per CU, tools/dma_probe.hip); in the panel image the 16 rows' 64 bytes are
 *              1 KB contiguous.  Same results bit for bit; the six STORE shapes of config 2 run 2-9 % faster with W as an
 *              image and 2-9 % more with A too.  The two kernels with a 32-wide K-step only (ping-pong, wide 4-wave): apla_gemm_nt_panel_ok(M, N, K, epilogue, out_dtype)
 *              tells whether a problem is covered AND runs on one of them under the automatic schedule (an image passed anyway
 *              forces the kernel where it covers the problem and is an error where it does not).
 *   bit 18     the OUTPUT C [M, N] is written as its K-panel image [N/32][M][32] (`ldc` not read): 16-bit GELU / GELU_FWD / MUL
 *              epilogues (on the 4-wave persistent kernel: row-major operands only), so that fc1's h and dfc2's
 *              product reach the next GEMM (fc2, dfc1) as images without a conversion pass.
 *   bit 19     the epilogue's second operand (aux_out of GELU = gelu', aux_in of MUL) is an image [N/32][M][32] (`ld_aux_*` not
 *              read): private to those two epilogues, stored and loaded in whole lines.  Same kernels and conditions as bit 18.
 *   bits 20-27 CUs to leave FREE (0..191): the persistent kernels start one (ping-pong) or two (both 4-wave kernels) workgroups per CU on
 *              256 - n CUs instead of all 256.  The data-parallel step passes the CU budget of its collective here, so that the
 *              RCCL kernels of the overlapped gradient all-reduce (wrappers.py:182-183) run beside the GEMMs instead of taking a
 *              CU from a launch that wants them all.  At config 2 a reservation of up to 8 CUs costs the GEMMs nothing: their
 *              tile counts (237 / 711 tiles of 320 x 256, 3 792 of 160 x 128) leave that many CUs idle in the last round anyway. */
int apla_gemm_nt_panel_ok(int M, int N, int K, int epilogue, int out_dtype);
int apla_gemm_nt_out_image_ok(int M, int N, int K, int epilogue, int out_dtype);   /* bit 18: is the image store available AND on the kernel the automatic schedule picks? */
/* dst[(k / 32) * rows + r][k % 32] = src[r][k] for a 16-bit [rows, K] matrix with row pitch ld (K % 32 == 0): the K-panel image */
int apla_pack_k_panels(const void* src, long ld, void* dst, int rows, int K, hipStream_t stream);
int apla_gemm_nt_ex(const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc, int M, int N,
                    int K, int epilogue, int out_dtype, const void* aux_in, int ld_aux_in, void* aux_out,
                    int ld_aux_out, int flags, hipStream_t stream);

/* apla_gemm_nt_ex(..., APLA_EPI_GELU, ...) with Mlp.drop after the activation (utils/transformers/vit.py:164-165) INSIDE the epilogue (round 6):
 * C = drop(GELU(A W^T + b)) and aux_out = GELU'(.) through the SAME keep mask, so that the backward's dfc2 epilogue (dh * aux_out) needs no
 * change.  The keep decision of output (m, n) is that of apla_dropout_fwd for element i = m * N + n (row-major position, whatever layout
 * the outputs have: `flags` bits 18 / 19 = image outputs as in apla_gemm_nt_ex; bits 0-7 tag, 20-27 reserved CUs), with {seed, step} read
 * from device memory and offset = step * rng_stride + site as in apla_layernorm_fwd_drop.  Runs on the 4-wave persistent kernel
 * (N % 128 == 0, row-major operands). */
int apla_gemm_nt_gelu_drop(const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc, int M, int N, int K,
                           void* aux_out, int ld_aux_out, int flags, const unsigned long long* rng, unsigned long long rng_stride,
                           unsigned site, float p_drop, hipStream_t stream);

/* C[M,N] = A[M,K] @ W[N,K]^T (+ bias) for problems with FEW output tiles and a LONG K (the input gradient of the DINO head's prototype
 * layer, dino_head.py:12-40: M = a few thousand rows, N = 256, K = 65 536 prototypes — 35 tiles of the tiled kernels, 2 048 K-steps
 * each): the K axis is cut into S parts (apla_gemm_nt_splitk_workspace_bytes / (M*N*4)), every (tile, part) pair is a work item of
 * the wide 4-wave kernel writing an fp32 partial tile into `workspace`, and a second launch sums the parts in a fixed order (bitwise
 * reproducible).  16-bit operands, N % 256 == 0, K % 32 == 0, K >= 256; out_dtype APLA_F32 or the build's 16-bit type. */
long apla_gemm_nt_splitk_workspace_bytes(int M, int N, int K);
int apla_gemm_nt_splitk(const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc, int M, int N, int K,
                        int out_dtype, void* workspace, long workspace_bytes, hipStream_t stream);

/* The kernel apla_gemm_nt_ex runs this problem on, as a rocprofv3 kernel trace shows it (tools/summarize_prof.py spelling), e.g.
 * "gemm_persist_kernel<GELU,bf16,5>"; `flags` as for apla_gemm_nt_ex; buf needs >= 48 bytes.  Same decision code as the launch:
 * bench.py names the kernel of its roofline record from this, not from a literal. */
int apla_gemm_nt_kernel_name(int M, int N, int K, int epilogue, int out_dtype, int flags, char* buf, int buflen);

/* apla_gemm_nt for FEW rows (the CLS-only tail of the last block: projection / MLP / Q of Block.forward on the B rows that
 * reach x[:, 0], vit.py:279-288,416-419): the K axis is split over K/64 (K <= 1536) or K/128 slices so that a few hundred
 * workgroups run instead of N/128; fp32 partial tiles go to `workspace` (apla_gemm_small_workspace_bytes(M, N, K) bytes,
 * caller-owned) and a second launch sums them in a fixed order, adds the bias and applies the epilogue (STORE, RESIDUAL,
 * MUL, GELU with the argument meaning of apla_gemm_nt).  Needs N % 64 == 0, K % 128 == 0, M <= 4096.  Same result as
 * apla_gemm_nt up to the summation order of the fp32 accumulation. */
long apla_gemm_small_workspace_bytes(int M, int N, int K);
int apla_gemm_nt_small(const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc, int M, int N,
                       int K, int epilogue, int out_dtype, const void* aux_in, int ld_aux_in, void* aux_out,
                       int ld_aux_out, void* workspace, long workspace_bytes, hipStream_t stream);

/* y = LayerNorm(x [+ add_in])*gamma+beta (y_dtype bf16, or f32 for the classifier-head input), saving mean/rstd (fp32).
 * x rows are `x_row_stride` elements apart so the final-norm-on-CLS-rows case (vit.py:416-419) needs no gather.
 * If add_in != NULL (bf16 [M,D], the branch output of the preceding projection / fc2 GEMM) the residual update
 * x_new = x + add_in of Block.forward (vit.py:284-285) is fused: x_new is written to x_out (residual dtype; may alias x)
 * and normalised in the same pass.  Replaces nn.LayerNorm(eps=1e-6) (vit.py:251,261,554). */
int apla_layernorm_fwd(const void* x, int res_dtype, long x_row_stride, const float* gamma, const float* beta,
                       void* y, int y_dtype, int ldy, float* mean, float* rstd, int M, int D, float eps,
                       const void* add_in, long add_row_stride, void* x_out, long x_out_row_stride,
                       hipStream_t stream);

/* dx_out = dres_in + LN_backward_dx(dy; x, gamma, mean, rstd)   (gamma/beta frozen: apla/apla_vit.py:80-81).
 * x is the saved forward input (x_dtype); dres_in/dx_out are the residual-gradient stream (grad_dtype; dres_in may be
 * NULL (=0) and may alias dx_out; both use dx_row_stride).  dx_bf16_copy (optional) receives a bf16 copy of dx_out —
 * the GEMM operand when the gradient stream is fp32.  If `gather_out` != NULL also writes the APLA-trainable columns
 * of dx_out: gather_out[m, j] = dx_out[m, inds[j]] for j < r (bf16) — the only part of the projection's output
 * gradient that the column-masked dW1 needs (autograd of the scatter at appla_attn.py:70-74). */
int apla_layernorm_bwd(const void* dy, int dy_dtype, int lddy, const void* x, int x_dtype, long x_row_stride,
                       const float* gamma, const float* mean, const float* rstd, const void* dres_in, void* dx_out,
                       int grad_dtype, long dx_row_stride, void* dx_bf16_copy, long copy_row_stride,
                       const int32_t* inds, int r, void* gather_out, int M, int D, hipStream_t stream);

/* apla_layernorm_bwd with three more choices (what the fused step uses since round 3):
 *   mean == NULL        `x` holds the NORMALISED row xhat = (x - mean) * rstd as apla_layernorm_fwd wrote it with gamma == beta ==
 *                       NULL (16-bit, one buffer per LayerNorm of the model): the backward reads 2 bytes per element instead of
 *                       the 4 of the fp32 residual row, and the residual stream itself need not be kept per block;
 *   gamma == NULL       no affine part: gamma / beta of a frozen LayerNorm are folded into the weight / bias of the frozen Linear
 *                       that follows it (qkv: appla_attn.py:53, fc1 / w12: vit.py:152-168,131-149), W' = W diag(gamma),
 *                       b' = b + W beta, so dy already is the gradient of xhat;
 *   dres_row_period p   p > 1: dres_in is non-zero only in rows m with m % p == 0 and is not read elsewhere (the residual gradient
 *                       below the final norm lives in the CLS rows only, vit.py:416-419): no zero fill of the stream.
 * apla_layernorm_fwd accepts gamma == beta == NULL accordingly (y = xhat). */
int apla_layernorm_bwd_ex(const void* dy, int dy_dtype, int lddy, const void* x, int x_dtype, long x_row_stride,
                          const float* gamma, const float* mean, const float* rstd, const void* dres_in, int dres_row_period,
                          void* dx_out, int grad_dtype, long dx_row_stride, void* dx_bf16_copy, long copy_row_stride,
                          const int32_t* inds, int r, void* gather_out, int M, int D, hipStream_t stream);

/* Stochastic depth fused into the two LayerNorm kernels (round 6).  Replaces DropPath around both branches of a block
 * (utils/transformers/vit.py:74-93 drop_path / DropPath, :284-285 `x = x + self.drop_path(...)`; main.py --dpr): sample b keeps a branch with
 * factor scale[b] = floor(keep_prob + u_b) / keep_prob (0 or 1 / keep_prob, one float per sample, drawn by the caller).
 *   apla_layernorm_fwd_dp : apla_layernorm_fwd with x_new = x + add_scale[m / scale_period] * add_in (add_scale NULL: factor 1).
 *   apla_layernorm_bwd_dp : apla_layernorm_bwd_ex with dx_out = dres_in + dy_scale[m / scale_period] * LN_backward_dx(dy; ...) — the
 *                           branch's whole backward chain is linear in its factor, so the factor is applied once, where the chain
 *                           ends — and gather_out[m, j] = gather_scale[m / scale_period] * dx_out[m, inds[j]] (the factor of the branch
 *                           whose projection dW reads the gathered columns).  Either scale may be NULL (1).
 * scale_period = rows per sample (N tokens; 1 when the rows are one per sample, e.g. the CLS rows). */
int apla_layernorm_fwd_dp(const void* x, int res_dtype, long x_row_stride, const float* gamma, const float* beta,
                          void* y, int y_dtype, int ldy, float* mean, float* rstd, int M, int D, float eps,
                          const void* add_in, long add_row_stride, void* x_out, long x_out_row_stride,
                          const float* add_scale, int scale_period, hipStream_t stream);
int apla_layernorm_bwd_dp(const void* dy, int dy_dtype, int lddy, const void* x, int x_dtype, long x_row_stride,
                          const float* gamma, const float* mean, const float* rstd, const void* dres_in, int dres_row_period,
                          void* dx_out, int grad_dtype, long dx_row_stride, void* dx_bf16_copy, long copy_row_stride,
                          const int32_t* inds, int r, void* gather_out, int M, int D, const float* dy_scale,
                          const float* gather_scale, int scale_period, hipStream_t stream);

/* The element-wise dropout of a branch inside the same two kernels (round 6; proj_drop appla_attn.py:82, Mlp.drop after fc2
 * vit.py:166-167; main.py --dr).  The keep decision of element i is that of apla_dropout_fwd below — word (i & 3) of
 * Philox4x32-10(counter {i >> 2, offset}, key seed) >= p * 2^32 — with {seed, step} read from DEVICE memory (`rng`, two 64-bit words) and
 * offset = step * rng_stride + site: the step counter advances outside a captured launch sequence, the launches stay the same.
 *   apla_layernorm_fwd_drop : apla_layernorm_fwd_dp with x_new = x + add_scale * (keep ? add_in / (1 - p) : 0); element index of
 *                             add_in[m, c] = m * drop_row_stride + c (the row-major index in the branch's full [M, D] tensor, also when only
 *                             every N-th row is visited).  rng NULL: no dropout.
 *   apla_layernorm_bwd_drop : apla_layernorm_bwd_dp that also writes masked_out[m, c] = keep ? dx_out[m, c] / (1 - p) * mask_scale[sample] : 0
 *                             (16-bit; element index m * D + c) — dx_out through the mask of the branch that consumes it NEXT, i.e. the
 *                             operand of that branch's dX GEMM — and takes gather_out from masked_out.  16-bit normalised-row form only. */
int apla_layernorm_fwd_drop(const void* x, int res_dtype, long x_row_stride, const float* gamma, const float* beta,
                            void* y, int y_dtype, int ldy, float* mean, float* rstd, int M, int D, float eps,
                            const void* add_in, long add_row_stride, void* x_out, long x_out_row_stride,
                            const float* add_scale, int scale_period, const unsigned long long* rng, unsigned long long rng_stride,
                            unsigned site, float p, long drop_row_stride, hipStream_t stream);
int apla_layernorm_bwd_drop(const void* dy, int dy_dtype, int lddy, const void* x, int x_dtype, long x_row_stride,
                            const float* gamma, const float* mean, const float* rstd, const void* dres_in, int dres_row_period,
                            void* dx_out, int grad_dtype, long dx_row_stride, void* dx_bf16_copy, long copy_row_stride,
                            const int32_t* inds, int r, void* gather_out, int M, int D, const float* dy_scale,
                            const float* gather_scale, int scale_period, void* masked_out, long masked_row_stride,
                            const float* mask_scale, const unsigned long long* rng, unsigned long long rng_stride, unsigned site,
                            float p, hipStream_t stream);

/* Gather only (used when the projection output gradient is already materialised): out[m,j] = src[m,inds[j]] bf16. */
int apla_gather_cols(const void* src, int res_dtype, long src_row_stride, const int32_t* inds, int r, void* out,
                     int M, int D, hipStream_t stream);

/* Fused multi-head attention forward on the packed qkv activations [B*N, 3*H*64] (layout [.., 3, H, 64], exactly the
 * output of the qkv Linear): o[B*N, H*64] = softmax(q k^T * scale) v, lse[B,H,N] = log-sum-exp of the scaled scores.
 * Replaces appla_attn.py:53-60 without materialising attn[B,H,N,N].  head_dim must be 64. */
int apla_attn_fwd(const void* qkv, void* o, float* lse, int B, int N, int H, float scale, hipStream_t stream);

/* Block-diagonal attention over a PACKED batch of S variable-length sequences (the dinov2 nested-tensor path:
 * appla_attn_mem_eff.py:40-42 calls xformers memory_efficient_attention with a BlockDiagonalMask over the concatenated
 * crops, dinov2/layers/block.py:254-288).  qkv [total, 3*H*64], o / d_o [total, H*64], dqkv like qkv; cu_seqlens int32[S+1]
 * on the device (cu[0] = 0, cu[S] = total); lse / delta are [H, total] fp32; max_n = longest sequence (sizes the grid).
 * Token t of sequence s attends to the tokens of sequence s only.  Same kernels as the uniform-batch entry points. */
int apla_attn_varlen_fwd(const void* qkv, void* o, float* lse, const int32_t* cu_seqlens, int S, int total, int max_n,
                         int H, float scale, hipStream_t stream);
int apla_attn_varlen_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta, void* dqkv,
                         const int32_t* cu_seqlens, int S, int total, int max_n, int H, float scale, hipStream_t stream);

/* The attention entry points with an explicit kernel choice (`variant`; the plain entry points pass 0):
 *   0 = auto: uniform batches of up to 280 tokens with at least one head per CU use the persistent forward (resident workgroups
 *       walking their heads, K / V double-buffered by a loader wave, one row maximum per head: one wave per 32-row block up to 224
 *       tokens — several small workgroups per CU up to 96 —, four waves walking 8 or 9 blocks above) and, up to 256 tokens and at
 *       257 (eight blocks + the last token as rank-1 corrections: a ViT/14 at 224 pixels), the
 *       persistent backward (every load one phase ahead of its use; several small workgroups per CU for short sequences, down to
 *       one token); other batches of up to 288 tokens the one-workgroup-per-head
 *       kernels (whole K / V of a head in LDS); longer sequences the key-/query-blocked kernels;
 *   1 = always the blocked kernels; 2 = the one-workgroup-per-head kernels (never the persistent ones); 3 = the persistent
 *       kernels wherever they apply.  The backward kernels compute bitwise the same results (except the 257-token form of the
 *       persistent one: other products for the last token, equal to rounding), and so do the blocked and the
 *       one-workgroup-per-head forward; the persistent forward takes ONE maximum per row instead of a running one per 64 keys and
 *       agrees with them to rounding (tests/test_kernels_gpu.py).  Bits 8.. of `variant` are ignored. */
int apla_attn_fwd_ex(const void* qkv, void* o, float* lse, int B, int N, int H, float scale, int variant, hipStream_t stream);
/* Attention with dropout on the attention probabilities (appla_attn.py:56-58: attn = softmax(...); attn = self.attn_drop(attn); x = attn @ v;
 * settable from main.py:109-111 --adr, 0 in every shipped configuration): o = (keep ? attn / (1 - p) : 0) v.  The keep decision of
 * (b, h, q, key) is word (key & 3) of Philox4x32-10(counter {key >> 2, row (64 bit), offset}, key = seed) >= p * 2^32 with row =
 * (b * H + h) * N + q: counter-based, so forward and backward regenerate the same mask (nothing is stored) and the oracle reproduces
 * it bit for bit.  lse is the softmax's own (undropped).  Uniform batches, key- / query-blocked kernels.  0 < p < 1. */
int apla_attn_fwd_dropout(const void* qkv, void* o, float* lse, int B, int N, int H, float scale, float p,
                          unsigned long long seed, unsigned offset, hipStream_t stream);
int apla_attn_bwd_dropout(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta, void* dqkv, int B,
                          int N, int H, float scale, float p, unsigned long long seed, unsigned offset, hipStream_t stream);
/* Name of the kernel an attention launch is dispatched to (the same decision code as the launch itself): backward = 0 | 1,
 * packed = 1 for the block-diagonal entry points (N = max_n there).  Diagnostics / bench records; no reference counterpart. */
int apla_attn_kernel_name(int backward, int packed, int B, int N, int H, int variant, char* buf, int buflen);
int apla_attn_bwd_ex(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta, void* dqkv, int B, int N,
                     int H, float scale, int variant, hipStream_t stream);
int apla_attn_varlen_fwd_ex(const void* qkv, void* o, float* lse, const int32_t* cu_seqlens, int S, int total, int max_n,
                            int H, float scale, int variant, hipStream_t stream);
int apla_attn_varlen_bwd_ex(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta, void* dqkv,
                            const int32_t* cu_seqlens, int S, int total, int max_n, int H, float scale, int variant,
                            hipStream_t stream);

/* Attention backward from (qkv, o, do, lse): dqkv[B*N, 3*H*64].  `delta` is a caller workspace of B*H*N floats.
 * Deterministic (no atomics).  Autograd of appla_attn.py:53-60. */
int apla_attn_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta, void* dqkv, int B,
                  int N, int H, float scale, hipStream_t stream);

/* Attention forward for the CLS query only (last block: the head reads x[:, 0] after the final norm, vit.py:416-419, so
 * the other rows of this block's attention output are dead).  Same qkv layout as apla_attn_fwd; writes row b*N of o (all
 * heads) and lse[b, h, 0]; the other rows of o / entries of lse are left untouched. */
int apla_attn_fwd_cls(const void* qkv, void* o, float* lse, int B, int N, int H, float scale, hipStream_t stream);

/* Attention backward when only the CLS query (token 0 of every sequence) has a non-zero output gradient — the last ViT
 * block under `x[:,0]` pooling (utils/transformers/vit.py:416-419).  do_cls is compact [B, H*64] (the CLS rows of dO);
 * writes the full dqkv [B*N, 3*H*64] (dq is zero except token 0).  Same math as apla_attn_bwd, rank-1 per head. */
int apla_attn_bwd_cls(const void* qkv, const void* o, const void* do_cls, const float* lse, void* dqkv, int B, int N,
                      int H, float scale, hipStream_t stream);

/* Materialise attn[B,H,N,N] (fp32) on demand — the second return value of APLA_Attention.forward (appla_attn.py:83),
 * used only by Block.forward(return_attention=True) (vit.py:279-287). */
int apla_attn_probs(const void* qkv, const float* lse, float* attn, int B, int N, int H, float scale,
                    hipStream_t stream);
/* The same matrix AFTER the dropout of apla_attn_fwd_dropout with the same (p, seed, offset): kept entries / (1 - p), the others 0 —
 * what APLA_Attention.forward returns in training mode (appla_attn.py:56-58, 83). */
int apla_attn_probs_dropout(const void* qkv, const float* lse, float* attn, int B, int N, int H, float scale, float p,
                            unsigned long long seed, unsigned offset, hipStream_t stream);

/* The APLA output projection as one forward and one backward operator (appla_attn.py:62-79 and its autograd), composed of
 * the entry points of this header — no kernels of their own.
 *   apla_proj_fwd:  y[M,D] = x[M,D] @ Wnat^T + bnat, Wnat = the natural-order merged weight kept by apla_pack_proj_rows
 *                   (row inds[j] = proj_weight1[j] for j < r, the frozen rows elsewhere): the two F.linear + two scatter_
 *                   calls of the reference are one GEMM.  `inds` / `r` are accepted for symmetry and not read.
 *   apla_proj_bwd:  dx[M,D] = dy @ Wnat (through WnatT, the transposed copy; dx may be NULL), dW1[r,D] / db1[r] fp32 = the
 *                   gradient of the r trainable rows only (gather of dy[:, inds[:r]], then apla_proj_dw); any r
 *                   (non-multiples of 64 are padded inside the workspace; `accumulate` needs r % 64 == 0).
 * `inds` is the int32 device vector of the module's `inds` buffer (length D, trainable rows first).  Workspace:
 * apla_proj_workspace_bytes(M, D, r) bytes, caller-owned, 16-byte aligned.  16-bit activations, D % 128 == 0. */
long apla_proj_workspace_bytes(int M, int D, int r);
int apla_proj_fwd(const void* x, const void* Wnat, const float* bnat, const int32_t* inds, void* y, int M, int D, int r,
                  hipStream_t stream);
int apla_proj_bwd(const void* dy, const void* x, const void* WnatT, const int32_t* inds, void* dx, float* dW1, float* db1,
                  void* workspace, long workspace_bytes, int M, int D, int r, int accumulate, hipStream_t stream);

/* Column-masked weight gradient of the APLA projection:
 *   dW1[j,:] (+)= row_scale[j] * sum_m dyg[m,j] * x[m,:]     dW1 fp32 [r,D]
 *   db1[j]   (+)= row_scale[j] * sum_m dyg[m,j]              db1 fp32 [r]
 * dyg bf16 [M,r] (gathered trainable columns), x bf16 [M,D] (projection input = attention output).
 * The (D-r) frozen columns of dW are never formed.  Replaces autograd of F.linear(x, proj_weight1, proj_bias1)
 * (appla_attn.py:64).  `partial` is a caller workspace of apla_dw_workspace_bytes(M,r,D) bytes.
 * accumulate != 0 adds into dW1/db1.  row_scale may be NULL (=1).  Deterministic (slab reduction). */
long apla_dw_workspace_bytes(int M, int r, int D);
int apla_proj_dw(const void* dyg, const void* x, int ldx, const float* row_scale, float* dW1, float* db1,
                 void* partial, int M, int r, int D, int accumulate, hipStream_t stream);
/* The same gradient for nb (1..8) projections of equal shape in ONE launch pair — e.g. the blocks of one backward segment,
 * whose dyg / x the caller kept: the arrays are HOST arrays of nb device pointers (row_scale, or single entries of it, may be
 * NULL).  The token axis is cut into 1/nb as many slabs per layer, so each workgroup's pipeline is nb times longer and the
 * slab partials per layer shrink by nb (config 2, 11 blocks in batches of 6 + 5: 42 -> 22 us per block).  The result of a
 * layer differs from apla_proj_dw's only by the fp32 summation order over the token slabs (deterministic for a given nb).
 * `partial`: apla_dw_workspace_bytes_batched(M, r, D, nb) bytes. */
long apla_dw_workspace_bytes_batched(int M, int r, int D, int nb);
int apla_proj_dw_batched(int nb, const void* const* dyg, const void* const* x, int ldx, const float* const* row_scale,
                         float* const* dW1, float* const* db1, void* partial, int M, int r, int D, int accumulate,
                         hipStream_t stream);

/* Refresh the engine-layout copies of the projection after W1/b1 changed:
 *   Wnat[inds[j], :]  = bf16(gamma[inds[j]] * W1[j, :])         natural-order forward weight  [D,D]
 *   WnatT[:, inds[j]] = bf16(gamma[inds[j]] * W1[j, :])         transposed copy for dX         [D,D]
 *   bnat[inds[j]]     = gamma[inds[j]] * b1[j]                                                 [D] fp32
 * for j < r (gamma may be NULL).  This is where the reference's two scatter_ calls (appla_attn.py:70-79) go:
 * applied once per step to r rows of weights instead of every forward to M rows of activations. */
int apla_pack_proj_rows(const float* W1, const float* b1, const int32_t* inds, const float* gamma, void* Wnat,
                        void* WnatT, float* bnat, int r, int D, hipStream_t stream);

/* The same for all L blocks of a model in one launch (the per-step re-scatter after the optimizer): block l's fp32 masters
 * are W1 = flat + l*block_stride ([r,D]) followed by b1 ([r]) — the layout of the flat trainable buffer; inds_all [L,D],
 * gamma_all [L,D] or NULL, Wnat_all / WnatT_all [L,D,D] bf16, bnat_all [L,D]. */
int apla_pack_proj_rows_batched(const float* flat, long block_stride, const int32_t* inds_all, const float* gamma_all,
                                void* Wnat_all, void* WnatT_all, float* bnat_all, int L, int r, int D,
                                hipStream_t stream);
/* ... and, when the pointers are not NULL, the same r rows into the K-panel images of the two copies ([L][D/32][D][32] each;
 * the frozen rows of the images are the caller's, as they are for Wnat_all / WnatT_all) */
int apla_pack_proj_rows_batched_ex(const float* flat, long block_stride, const int32_t* inds_all, const float* gamma_all,
                                   void* Wnat_all, void* WnatT_all, float* bnat_all, void* Wnat_panels, void* WnatT_panels,
                                   int L, int r, int D, hipStream_t stream);

/* Fused global-norm clip + AdamW over the flat trainable buffer (defaults/trainer.py:127-138,
 * defaults/wrappers.py:205-221): grads are first multiplied by grad_scale (1/world for DDP mean), the global L2
 * norm is reduced on device (no host sync), clip coefficient = min(1, max_norm/(norm+1e-6)) (max_norm <= 0 disables),
 * weight decay applies to elements with decay_mask[i] != 0 (uint8).  `norm_ws` = 512 floats workspace:
 * [0] sum of squares, [1] resulting grad norm (pre-clip), [2..257] per-workgroup partials (deterministic reduction),
 * [260], [261] number of updates skipped so far because the norm was not finite (two slots: the call with step s reads
 * slot s&1 and writes slot (s+1)&1, so `step` must advance by one per call; zero the workspace before the first).  A skipped
 * update leaves params / moments untouched and does not count for the bias corrections (GradScaler.step semantics,
 * defaults/trainer.py:133-138): they use step - skipped.
 * Grads are overwritten with the scaled+clipped values (as clip_grad_norm_ does in place). */
int apla_adamw_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, const uint8_t* decay_mask,
                    long n, float lr, float weight_decay, float beta1, float beta2, float eps, int step,
                    float max_norm, float grad_scale, float* norm_ws, hipStream_t stream);

/* The two phases of apla_adamw_step on their own, for optimizers that keep a step count per tensor or skip tensors
 * (torch.optim.AdamW skips parameters whose .grad is None; self_supervised/dinov2/trainer.py:84-90 cancels the prototype
 * layer's gradients AFTER clip_grad_norm_ during the first epochs): apla_grad_sumsq leaves the 256 partial sums of
 * (grads*grad_scale)^2 over [0, n) in norm_ws; apla_adamw_apply re-reduces them, derives the same clip coefficient and
 * updates the n elements it is given (any sub-range of the buffer the norm was taken over) with bias corrections for
 * `step`.  apla_adamw_step(…) == apla_grad_sumsq(…) followed by apla_adamw_apply(…) on the same range. */
int apla_grad_sumsq(const float* grads, long n, float grad_scale, float* norm_ws, hipStream_t stream);
int apla_adamw_apply(float* params, float* grads, float* exp_avg, float* exp_avg_sq, const uint8_t* decay_mask,
                     long n, float lr, float weight_decay, float beta1, float beta2, float eps, int step,
                     float max_norm, float grad_scale, float* norm_ws, hipStream_t stream);

/* The EMA teacher of the self-supervised step (self_supervised/dinov2/models.py:443-453: torch._foreach_mul_(teacher, m) then
 * torch._foreach_add_(teacher, student, alpha = 1 - m)) over one flat fp32 range, in one pass: teacher[i] = fma(student[i], 1 - m,
 * round(teacher[i] * m)), the roundings of the two torch passes (m and 1 - m, taken in double, each cast to fp32 as torch casts its scalars). */
int apla_ema_update(float* teacher, const float* student, long n, double m, hipStream_t stream);

/* The same step under DYNAMIC loss scaling — torch.cuda.amp.GradScaler (defaults/trainer.py:129-138:
 * scaler.scale(loss).backward(); unscale_; clip; scaler.step; scaler.update) without a host round trip.  The gradients in
 * `grads` carry the current loss scale; `scaler` is a device float[8]: [0..2] and [3..5] are two slots of {scale,
 * growth_tracker, optimizer steps actually taken}; this call reads slot `parity` and writes slot parity^1 (the caller
 * alternates parity 0,1,0,… per call), [6] receives the scale the NEXT backward must apply to the loss gradient, [7] is set
 * to 1 if this call skipped the update.  Unscale by grad_scale/scale, clip as above; if the global norm is not finite the
 * parameters and moments are left untouched, scale *= backoff_factor and the tracker resets; otherwise the AdamW update
 * runs with bias corrections from the count of steps actually taken, and after growth_interval consecutive finite steps
 * scale *= growth_factor.  GradScaler defaults: scale 65536, growth 2, backoff 0.5, interval 2000. */
int apla_adamw_step_dynamic(float* params, float* grads, float* exp_avg, float* exp_avg_sq, const uint8_t* decay_mask,
                            long n, float lr, float weight_decay, float beta1, float beta2, float eps, float max_norm,
                            float grad_scale, float* scaler, int parity, float growth_factor, float backoff_factor,
                            int growth_interval, float* norm_ws, hipStream_t stream);

/* Patch embedding front end (vit.py:291-307, 387-396): images fp32 [B,3,S,S] -> im2col bf16 [B*Np, Kp] with
 * Kp = round_up(3*p*p, 64) (zero padded); then (after the GEMM) tokens[b,0] = cls+pos[0], tokens[b,1+t] =
 * patches[b,t] + pos[1+t] in residual dtype. */
int apla_patchify(const float* images, void* cols, int B, int S, int patch, int Kp, hipStream_t stream);
int apla_assemble_tokens(const void* patches, int ldp, const float* cls_token, const float* pos_embed, void* tokens,
                         int res_dtype, int B, int Np, int D, hipStream_t stream);
/* The same with iBOT masking (self_supervised/dinov2/dinov2_vits.py:210-222, prepare_tokens_with_masks): masked uint8 [B, Np]
 * (non-zero = masked), mask_token fp32 [D]; tokens[b,1+t] = (masked[b,t] ? mask_token : patches[b,t]) + pos[1+t].  Both NULL =
 * apla_assemble_tokens.  `tokens` may point into a larger buffer (the packed multi-crop batch): rows are written densely. */
int apla_assemble_tokens_masked(const void* patches, int ldp, const float* cls_token, const float* pos_embed,
                                const uint8_t* masked, const float* mask_token, void* tokens, int res_dtype, int B, int Np,
                                int D, hipStream_t stream);

/* torch.nn.utils.weight_norm(dim = 0) of the DINO head's prototype layer (self_supervised/dinov2/layers/dino_head.py:27-28):
 *   apla_weight_norm_fwd : W[i,:] = v[i,:] * g[i] / ||v[i,:]|| written in the build's 16-bit operand type, norm[i] = ||v[i,:]||
 *   apla_weight_norm_bwd : dv[i,:] = (g[i] / n) (dW[i,:] - v[i,:] (dW[i,:] . v[i,:]) / n^2),  dg[i] = (dW[i,:] . v[i,:]) / n   (dg may be NULL)
 * v, dW, dv fp32 [K, D] row-major, g / norm / dg fp32 [K]; D % 4 == 0. */
int apla_weight_norm_fwd(const float* v, const float* g, void* w_h16, float* norm, int K, int D, hipStream_t stream);
int apla_weight_norm_bwd(const float* dw, const float* v, const float* g, const float* norm, float* dv, float* dg, int K, int D,
                         hipStream_t stream);
/* apla_weight_norm_fwd with W^T [D, K] (16-bit, row-major) written beside W: what the layer's dX GEMM reads (dx = dy W through
 * apla_gemm_nt on W^T).  K % 64 == 0, D <= 1024; wt_h16 == NULL is apla_weight_norm_fwd. */
int apla_weight_norm_fwd_t(const float* v, const float* g, void* w_h16, void* wt_h16, float* norm, int K, int D, hipStream_t stream);

/* KoLeoLoss (self_supervised/dinov2/loss/koleo_loss.py:17-45) on G groups of B rows, as models.py:410-413 sums it over the two global
 * crops: xn = x / max(||x||, eps), j(i) = the other row of the group with the largest xn_i . xn_j, d_i = ||xn_i - xn_j(i) + 1e-8||,
 * loss_g = -mean_i log(d_i + eps).
 *   apla_koleo_fwd : out[g] = loss_g (g < G), out[G] = sum_g loss_g; nn_idx / dist / nrm (||x_i||) / terms: [G*B] saved for the backward
 *                    (terms is scratch); `counter`: one int32 that is 0 before the first call (the kernel leaves it 0).
 *   apla_koleo_bwd : dx = d(out[G]) / dx * gout[0]   (gout: one fp32 in device memory; dx in x's type)
 * x [G*B, D] row-major, fp32 or the build's 16-bit type (`dtype`), D % 4 == 0 (backward: D <= 4096, B <= 4096). */
int apla_koleo_fwd(const void* x, int dtype, int G, int B, int D, float eps, int* nn_idx, float* dist, float* nrm, float* terms,
                   float* out, int* counter, hipStream_t stream);
int apla_koleo_bwd(const void* x, int dtype, int G, int B, int D, float eps, const int* nn_idx, const float* dist, const float* nrm,
                   const float* gout, void* dx, hipStream_t stream);

/* Dropout and stochastic depth of the module path (utils/transformers/vit.py:74-93 DropPath, :152-168 Mlp.drop; apla/appla_attn.py:82
 * proj_drop; all shipped configurations use 0, main.py:101-111 can set them).
 *   apla_dropout_fwd : y = keep ? x / (1 - p) : 0 and keep[i] (one byte per element, for the backward).  Element i keeps iff word
 *                      (i & 3) of Philox4x32-10(counter {i >> 2, offset}, key seed) >= p * 2^32 — counter-based: independent of the
 *                      launch geometry, reproduced bit for bit by the oracle.  n % 8 == 0; dtype fp32 or the build's 16-bit type.
 *   apla_dropout_bwd : dx = keep ? dy / (1 - p) : 0
 *   apla_scale_samples: y[s, :] = x[s, :] * scale[s] (DropPath: scale[s] = floor(keep_prob + u_s) / keep_prob, drawn by the caller —
 *                      one number per sample); the same call is its backward. */
int apla_dropout_fwd(const void* x, int dtype, void* y, uint8_t* keep, long n, float p, unsigned long long seed,
                     unsigned long long offset, hipStream_t stream);
int apla_dropout_bwd(const void* dy, int dtype, const uint8_t* keep, void* dx, long n, float p, hipStream_t stream);
/* apla_dropout_fwd with {seed, step} read from DEVICE memory (`rng`, two 64-bit words; offset = step * rng_stride + site) and `keep`
 * optional (NULL: not written): the form a captured launch sequence replays with a new mask every step (round 6: pos_drop and the
 * gradient entering the last block of the fused step).  The same call applied to a gradient is the site's backward. */
int apla_dropout_fwd_dev(const void* x, int dtype, void* y, uint8_t* keep, long n, float p, const unsigned long long* rng,
                         unsigned long long rng_stride, unsigned site, hipStream_t stream);
int apla_scale_samples(const void* x, int dtype, void* y, const float* scale, long samples, long per_sample, hipStream_t stream);

/* Self-distillation losses of the DINOv2-APLA step (dinov2/loss/dino_clstoken_loss.py,
 * dinov2/loss/ibot_patch_loss.py) over rows of K prototype logits (K = 65 536 in the shipped config):
 *   apla_softmax_center : out[r,:] = softmax((x[r,:] - center[:]) * inv_temp)    teacher centering + sharpening
 *                         (DINOLoss.softmax_center_teacher, dino_clstoken_loss.py:29-32; x fp32 or 16-bit, out fp32)
 *   apla_distill_ce     : row_loss[r] = -w_r * sum_k t[r,k] * log_softmax(s[r,:] * inv_temp)[k] with w_r = weight *
 *                         (row_weight ? row_weight[r] : 1); dstudent[r,:] (+)= w_r * inv_temp * (softmax * sum_k t[r,k] - t[r,:])
 *                         (the double loop of DINOLoss.forward, :65-77, is a sum over targets: pass t = sum of the teacher
 *                         views; iBOTPatchLoss.forward_masked, ibot_patch_loss.py:103-121, is the row_weight form).
 *                         dstudent may be NULL (loss only); accumulate != 0 adds into it.  student fp32 or 16-bit. */
int apla_softmax_center(const void* x, int x_dtype, long ldx, const float* center, float inv_temp, float* out, long ldo,
                        int R, int K, hipStream_t stream);
int apla_distill_ce(const void* student, int s_dtype, long lds, const float* teacher_probs, long ldt, float inv_temp,
                    const float* row_weight, float weight, float* dstudent, long ldds, int accumulate, float* row_loss,
                    int R, int K, hipStream_t stream);
/* the same with the gradient written in `ds_dtype` (APLA_F32 or the build's 16-bit type: the student's own dtype saves the fp32
 * round trip of a [rows, 65 536] gradient; 16-bit needs K % 8 == 0 and 16-byte aligned rows) */
int apla_distill_ce_ex(const void* student, int s_dtype, long lds, const float* teacher_probs, long ldt, float inv_temp,
                       const float* row_weight, float weight, void* dstudent, int ds_dtype, long ldds, int accumulate,
                       float* row_loss, int R, int K, hipStream_t stream);
/* apla_distill_ce_ex with repeating targets: row r reads teacher_probs row r % t_rows (DINOLoss.forward, dino_clstoken_loss.py:65-77,
 * pairs every local crop of the batch with the same teacher rows: the 8 crops of the shipped recipe are one launch of 8 x 64 rows) */
int apla_distill_ce_bcast(const void* student, int s_dtype, long lds, const float* teacher_probs, long ldt, int t_rows, float inv_temp,
                          const float* row_weight, float weight, void* dstudent, int ds_dtype, long ldds, int accumulate,
                          float* row_loss, int R, int K, hipStream_t stream);
/* apla_softmax_center + apla_distill_ce_ex in one: the targets are softmax((teacher_logits - center) * inv_temp_t), computed on the
 * fly and never written (iBOTPatchLoss.softmax_center_teacher + forward_masked, ibot_patch_loss.py:46-55, 103-121: 4 879 rows of
 * 65 536 prototypes per iteration at config 4 — 10 bytes per logit instead of 22).  K % 8 == 0, 16-byte aligned rows. */
int apla_distill_ce_centered(const void* student, int s_dtype, long lds, const void* teacher_logits, int x_dtype, long ldx,
                             const float* center, float inv_temp_s, float inv_temp_t, const float* row_weight, float weight,
                             void* dstudent, int ds_dtype, long ldds, float* row_loss, int R, int K, hipStream_t stream);

/* Input side of the step (bases.py:69-231 ToTensor + Normalize + horizontal flip, utils/_utils.py:424-441
 * timm Mixup / CutMix applied by the collate function): decoded uint8 images already in device memory ->
 * the normalised fp32 batch the patch embedding reads, in one pass.  src uint8 [B,3,S,S] (hwc = 0) or [B,S,S,3] (hwc = 1);
 * dst fp32 [B,3,S,S]; mean3 / std3: HOST pointers to three floats in [0,1] units (ImageNet statistics in the reference);
 * flip uint8 [B] or NULL; perm int32 [B] partner sample or NULL; lam fp32 [B] Mixup weight of the own image or NULL (= 1);
 * box int32 [B,4] = {y0, y1, x0, x1} CutMix rectangle taken from the partner, or NULL (Mixup).  All but mean3/std3 are
 * device pointers.  The label side of Mixup is apla_cross_entropy_soft's probability targets. */
int apla_augment_images(const uint8_t* src, float* dst, const float* mean3, const float* std3, const uint8_t* flip,
                        const int32_t* perm, const float* lam, const int32_t* box, int B, int S, int hwc,
                        hipStream_t stream);

/* Classifier head on the normalised CLS features (defaults/models.py:64-65,86-87) and mean cross-entropy
 * (defaults/wrappers.py:312-316), fp32 throughout.  The head is [B,D]x[D,C] (0.2 GFLOP): plain fp32 FMA kernels.
 *   apla_sgemm_small : C[i,j] (+)= sum_k A[i*sai + k*sak] * B[k*sbk + j*sbj] (+ bias[j])   (any strides: NT/NN/TN)
 *   apla_cross_entropy: dlogits = (softmax(logits) - onehot(labels)) / B ; row_loss[b] ; loss = mean(row_loss)
 *   apla_colsum      : out[j] = sum_i X[i*ld + j]                                           (bias gradient) */
int apla_sgemm_small(const float* A, long sai, long sak, const float* Bm, long sbk, long sbj, const float* bias,
                     float* C, long ldc, int M, int N, int K, int accumulate, hipStream_t stream);
int apla_cross_entropy(const float* logits, int ldl, const int32_t* labels, float* dlogits, float* row_loss,
                       float* loss, int B, int C, hipStream_t stream);
/* The same with probability targets [B, C] (fp32, rows sum to 1): what nn.CrossEntropyLoss receives when the reference's
 * `advanced_aug` (timm Mixup / CutMix / label smoothing, utils/_utils.py:424-441, defaults/wrappers.py:137-139) is on.
 * row_loss[b] = -sum_c t_bc log softmax(logits_b)_c ; dlogits = (softmax * sum_c t_bc - t) / B ; loss = mean(row_loss). */
int apla_cross_entropy_soft(const float* logits, int ldl, const float* targets, int ldt, float* dlogits, float* row_loss,
                            float* loss, int B, int C, hipStream_t stream);
int apla_colsum(const float* X, long ld, float* out, int M, int N, hipStream_t stream);
/* the same for a 16-bit X (fp32 sums; N % 4 == 0): the iBOT centre's column mean over a few thousand teacher rows (ibot_patch_loss.py:123-135) */
int apla_colsum_h16(const void* X, long ld, float* out, int M, int N, hipStream_t stream);

/* Diagnostic, not on the product path: `workgroups` workgroups of `threads` threads with `lds_bytes` of LDS each stay resident
 * for `usec` microseconds.  tools/contention_probe.py launches it on the side stream where the data-parallel step launches its
 * RCCL all-reduces (wrappers.py:182-183), to measure what foreign resident workgroups cost the step's persistent kernels. */
int apla_probe_occupy(int workgroups, int threads, int lds_bytes, int usec, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* APLA_HIP_H */
