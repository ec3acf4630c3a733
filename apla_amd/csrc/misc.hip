// Front end (patchify / token assembly) and classifier-head kernels.  All are forward-only or tiny: the patch embedding
// never sees a gradient under APLA (first trainable leaf is block 0's projection) and the head is [B,D]x[D,C] in fp32.
#include "common.h"

namespace {

// images fp32 [B,3,S,S] -> cols bf16 [B*Np, Kp]; column order (c, py, px) == conv weight.reshape(D, 3*p*p).
// One workgroup per row of patches (b, gy): the 3 x patch image rows it covers are read as whole rows (S floats = contiguous
// lines), converted and parked in LDS, then every patch row of `cols` leaves as one contiguous run of Kp elements — in the
// element-per-thread form every load took 4 bytes from a 64-byte run of a row and the launch ran at half the rate of a copy.
// PATCH > 0: compile-time patch size (14 / 16: every division below is by a constant); PATCH == 0: any even patch size.
template <int PATCH>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, bf16* __restrict__ cols, int S,
                                                       int patch_rt, int grid_w, int Np, int Kp) {
  extern __shared__ __attribute__((aligned(16))) char pf_smem[];
  bf16* tile = (bf16*)pf_smem;                 // [3 * patch][S]
  const int patch = PATCH > 0 ? PATCH : patch_rt;
  const int b = blockIdx.x / grid_w, gy = blockIdx.x - b * grid_w;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rows = 3 * patch, W2 = S >> 1;     // S even (checked on the host; 518 = 37 * 14 is not a multiple of 4)
  for (int rr = wave; rr < rows; rr += 4) {    // one image row per wave and pass: S floats, contiguous
    const int c = rr / patch, py = rr - c * patch;
    const float* src = img + (((size_t)b * 3 + c) * S + gy * patch + py) * S;
    for (int x2 = lane; x2 < W2; x2 += 64) {
      const f32x2 v = *(const f32x2*)(src + x2 * 2);
      bf16x2 o;
      o[0] = (bf16)v[0]; o[1] = (bf16)v[1];
      *(bf16x2*)(tile + rr * S + x2 * 2) = o;
    }
  }
  __syncthreads();
  const int pp = patch * patch, K = 3 * pp;
  bf16* out = cols + ((size_t)b * Np + (size_t)gy * grid_w) * Kp;      // grid_w consecutive rows of Kp elements
  for (int gx = wave; gx < grid_w; gx += 4) {                          // one patch row (Kp elements, contiguous) per wave and pass
    bf16* orow = out + (size_t)gx * Kp;
    for (int k = lane * 2; k < Kp; k += 128) {
      bf16x2 v;
      v[0] = (bf16)0.f; v[1] = (bf16)0.f;
      if (k < K) {
        const int c = k / pp, rem = k - c * pp, py = rem / patch, px = rem - py * patch;
        v = *(const bf16x2*)(tile + (c * patch + py) * S + gx * patch + px);
      }
      *(bf16x2*)(orow + k) = v;
    }
  }
}

template <typename ResT>
__global__ __launch_bounds__(256) void assemble_tokens_kernel(const bf16* __restrict__ patches, int ldp,
                                                              const float* __restrict__ cls,
                                                              const float* __restrict__ pos, ResT* __restrict__ tok,
                                                              int Np, int D, const uint8_t* __restrict__ masked,
                                                              const float* __restrict__ mask_token) {
  const int row = blockIdx.x;  // b*(Np+1) + n
  const int N = Np + 1;
  const int b = row / N, n = row - b * N;
  // iBOT (dinov2_vits.py:213-214): a masked patch is replaced by the mask token BEFORE the position embedding is added
  const bool use_mask_token = masked != nullptr && n > 0 && masked[(size_t)b * Np + n - 1] != 0;
  for (int c4 = threadIdx.x; c4 < D / 4; c4 += 256) {
    f32x4 v = *(const f32x4*)(pos + (size_t)n * D + c4 * 4);
    if (n == 0) v += *(const f32x4*)(cls + c4 * 4);
    else if (use_mask_token) v += *(const f32x4*)(mask_token + c4 * 4);
    else v += Vec4IO<bf16>::load(patches + ((size_t)b * Np + n - 1) * ldp + c4 * 4);
    Vec4IO<ResT>::store(tok + (size_t)row * D + c4 * 4, v);
  }
}

// C[i,j] (+)= sum_k A[i*sai + k*sak] * B[k*sbk + j*sbj] (+ bias[j]) in exact fp32 on the matrix pipes: one 32 x 32 output tile per
// workgroup, the K axis split over its eight waves, each wave a chain of v_mfma_f32_32x32x2_f32 (an f32 fma chain in k order, no
// wider internal accumulation: MI355X guide, "FP32-input MFMA"); the eight partial tiles are summed through LDS in wave order,
// so the result does not depend on the launch.  The head GEMMs are tiny (0.2 GFLOP, operands L2-resident) and latency-bound: the
// scalar-FMA tile kernel this replaces spent 25 us per launch on its staging round trips (three launches per step).
// Operand maps of the instruction (lane l: i or j = l & 31, half h = l >> 5): A register s of an 8-deep chunk holds
// A[i][k0 + 4h + s], B register s holds B[k0 + 4h + s][j]: any assignment of k to (half, step) is valid as long as both operands use it.
constexpr int SG_WAVES = 8;
__global__ __launch_bounds__(64 * SG_WAVES) void sgemm_small_kernel(const float* __restrict__ A, long sai, long sak,
                                                          const float* __restrict__ Bm, long sbk, long sbj,
                                                          const float* __restrict__ bias, float* __restrict__ C,
                                                          long ldc, int M, int N, int K, int accumulate) {
  __shared__ float red[SG_WAVES][1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  const int li = lane & 31, h = lane >> 5;
  int ia = i0 + li, jb = j0 + li;
  ia = ia < M ? ia : M - 1;          // rows / columns past the edge re-read the last valid one; their results are never stored
  jb = jb < N ? jb : N - 1;
  const float* ap = A + (long)ia * sai;
  const float* bp = Bm + (long)jb * sbj;
  // this wave's K range: whole 8-deep chunks, dealt in contiguous runs
  const int chunks = (K + 7) >> 3, per = (chunks + SG_WAVES - 1) / SG_WAVES;
  const int c_lo = wave * per, c_hi = (c_lo + per) < chunks ? (c_lo + per) : chunks;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  // four chunks (32 k) per step; the loads of step s + 1 are issued before the products of step s (the loop is latency-bound)
  float av[4][4], bv[4][4], an[4][4], bn[4][4];
  // an operand whose K axis is the contiguous one is read 16 bytes per lane (its 4 consecutive k of a chunk): as four scalar loads
  // each instruction touched 64 different lines for 4 bytes apiece, and the launch was bound by the address path, not by latency
  const bool a_vec = sak == 1 && (sai & 3) == 0 && (((uintptr_t)A) & 15) == 0 && (K & 3) == 0;
  const bool b_vec = sbk == 1 && (sbj & 3) == 0 && (((uintptr_t)Bm) & 15) == 0 && (K & 3) == 0;
  auto fetch = [&](int c, float (&a_)[4][4], float (&b_)[4][4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k0 = (c + u) * 8 + 4 * h;
      const bool live = (c + u) < c_hi;
      if (a_vec) {
        const f32x4 v = (live && k0 < K) ? *(const f32x4*)(ap + k0) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t) a_[u][t] = v[t];
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) a_[u][t] = (live && k0 + t < K) ? ap[(long)(k0 + t) * sak] : 0.f;
      }
      if (b_vec) {
        const f32x4 v = (live && k0 < K) ? *(const f32x4*)(bp + k0) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t) b_[u][t] = v[t];
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) b_[u][t] = (live && k0 + t < K) ? bp[(long)(k0 + t) * sbk] : 0.f;
      }
    }
  };
  if (c_lo < c_hi) fetch(c_lo, av, bv);
  for (int c = c_lo; c < c_hi; c += 4) {
    if (c + 4 < c_hi) fetch(c + 4, an, bn);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][t], bv[u][t], acc, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t) { av[u][t] = an[u][t]; bv[u][t] = bn[u][t]; }
  }
  // C/D map of the 32x32 shapes: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int e = 0; e < 16; ++e) red[wave][((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + li] = acc[e];
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 1024 / (64 * SG_WAVES); ++u) {
    const int e = threadIdx.x + 64 * SG_WAVES * u, i = i0 + (e >> 5), j = j0 + (e & 31);
    if (i < M && j < N) {
      float v = red[0][e];
#pragma unroll
      for (int w = 1; w < SG_WAVES; ++w) v += red[w][e];
      if (bias != nullptr) v += bias[j];
      if (accumulate) v += C[(long)i * ldc + j];
      C[(long)i * ldc + j] = v;
    }
  }
}

// one workgroup per sample: softmax cross-entropy, dlogits = (softmax - onehot) / B, row loss
// labels == nullptr: soft targets [B, C] (rows sum to 1: timm Mixup / label smoothing feed nn.CrossEntropyLoss probabilities,
// utils/_utils.py:424-441): loss_b = lse - sum_c t_bc * logit_bc, dlogits = (softmax - t) / B
__global__ __launch_bounds__(256) void cross_entropy_kernel(const float* __restrict__ logits, int ldl,
                                                            const int32_t* __restrict__ labels,
                                                            const float* __restrict__ targets, int ldt,
                                                            float* __restrict__ dlogits, float* __restrict__ row_loss,
                                                            int B, int C) {
  __shared__ float red[4];
  __shared__ float bc;
  const int b = blockIdx.x;
  const float* lr = logits + (size_t)b * ldl;
  float mx = -INFINITY;
  for (int c = threadIdx.x; c < C; c += 256) mx = fmaxf(mx, lr[c]);
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) bc = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  mx = bc;
  float s = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) s += __expf(lr[c] - mx);
  s = wave_sum(s);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) bc = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  const float lse = mx + __logf(bc);
  const float invB = 1.0f / (float)B;
  if (labels != nullptr) {
    const int y = labels[b];
    if ((unsigned)y >= (unsigned)C) {
      // a class id outside [0, C) (wrong n_classes, an ignore_index of -1 …): torch's CrossEntropyLoss raises; a kernel cannot,
      // so the row contributes no gradient and its loss is NaN — the mean loss of the batch turns NaN and the caller sees it
      for (int c = threadIdx.x; c < C; c += 256) dlogits[(size_t)b * ldl + c] = 0.f;
      if (threadIdx.x == 0) row_loss[b] = __builtin_nanf("");
      return;
    }
    for (int c = threadIdx.x; c < C; c += 256) {
      const float pr = __expf(lr[c] - lse);
      dlogits[(size_t)b * ldl + c] = (pr - (c == y ? 1.0f : 0.0f)) * invB;
    }
    if (threadIdx.x == 0) row_loss[b] = lse - lr[y];
    return;
  }
  const float* tr = targets + (size_t)b * ldt;
  float dot = 0.f, tsum = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float t = tr[c];
    dot += t * lr[c];
    tsum += t;
  }
  dot = wave_sum(dot);
  tsum = wave_sum(tsum);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = dot; }
  __syncthreads();
  const float dall = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = tsum; }
  __syncthreads();
  const float tall = red[0] + red[1] + red[2] + red[3];  // 1 for proper probability rows; kept general like torch
  for (int c = threadIdx.x; c < C; c += 256) dlogits[(size_t)b * ldl + c] = (__expf(lr[c] - lse) * tall - tr[c]) * invB;
  if (threadIdx.x == 0) row_loss[b] = tall * lse - dall;  // -sum_c t_c log p_c
}

__global__ __launch_bounds__(256) void mean_kernel(const float* __restrict__ v, float* __restrict__ out, int n) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += v[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (red[0] + red[1] + red[2] + red[3]) / (float)n;
}

// 64 columns per workgroup, its four waves take the rows i = w, w + 4, ...; summed through LDS in wave order
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, long ld, float* __restrict__ out,
                                                     int M, int N) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (j < N)
    for (int i = wave; i < M; i += 4) s += X[(long)i * ld + j];
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && j < N) out[j] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
}

// 16-bit input, many rows (the iBOT centre: the column mean of a [~5000, 65536] teacher output): 256 columns per workgroup,
// four per lane and access; its sixteen waves take the rows i = w, w + 16, ... eight at a time (a CU holds one workgroup per
// column group: 16 waves x 8 loads of 512 B keep 64 KB in flight per CU); fp32 sums, combined in wave order
__global__ __launch_bounds__(1024) void colsum16_kernel(const bf16* __restrict__ X, long ld, float* __restrict__ out, int M, int N) {
  __shared__ f32x4 red[16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 256 + lane * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (j < N) {
    int i = wave;
    for (; i + 7 * 16 < M; i += 8 * 16) {
      bf16x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *(const bf16x4*)(X + (long)(i + 16 * u) * ld + j);
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] += (float)v[u][e];
    }
    for (; i < M; i += 16) {
      const bf16x4 v = *(const bf16x4*)(X + (long)i * ld + j);
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] += (float)v[e];
    }
  }
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && j < N) {
    f32x4 t = red[0][lane];
#pragma unroll
    for (int v = 1; v < 16; ++v) t += red[v][lane];
    *(f32x4*)(out + j) = t;
  }
}

}  // namespace

extern "C" int apla_patchify(const float* images, void* cols, int B, int S, int patch, int Kp, hipStream_t stream) {
  APLA_REQUIRE(images && cols && B > 0 && S > 0 && patch > 0 && S >= patch, "apla_patchify: bad arguments");
  APLA_REQUIRE(Kp >= 3 * patch * patch && Kp % 64 == 0, "apla_patchify: Kp must be >= 3*p*p and a multiple of 64");
  APLA_REQUIRE(S % 2 == 0 && patch % 2 == 0 && (size_t)3 * patch * S * 2 <= 64 * 1024 && apla_aligned16(images),
               "apla_patchify: need an even image side, an even patch size and 3*patch*S*2 bytes of LDS <= 64 KB (S=%d patch=%d)", S, patch);
  const int gw = S / patch, Np = gw * gw;
  const size_t lds = (size_t)3 * patch * S * 2;
  if (patch == 16) hipLaunchKernelGGL(patchify_kernel<16>, dim3(B * gw), dim3(256), lds, stream, images, (bf16*)cols, S, patch, gw, Np, Kp);
  else if (patch == 14) hipLaunchKernelGGL(patchify_kernel<14>, dim3(B * gw), dim3(256), lds, stream, images, (bf16*)cols, S, patch, gw, Np, Kp);
  else hipLaunchKernelGGL(patchify_kernel<0>, dim3(B * gw), dim3(256), lds, stream, images, (bf16*)cols, S, patch, gw, Np, Kp);
  APLA_CHECK_LAUNCH("apla_patchify");
  return APLA_OK;
}

extern "C" int apla_assemble_tokens_masked(const void* patches, int ldp, const float* cls_token, const float* pos_embed,
                                           const uint8_t* masked, const float* mask_token, void* tokens, int res_dtype, int B,
                                           int Np, int D, hipStream_t stream) {
  APLA_REQUIRE(patches && cls_token && pos_embed && tokens && B > 0 && Np > 0 && D % 4 == 0 && ldp % 4 == 0, "apla_assemble_tokens: bad arguments");
  APLA_REQUIRE((masked == nullptr) == (mask_token == nullptr), "apla_assemble_tokens: the mask and the mask token come together");
  APLA_REQUIRE((long)B * (Np + 1) < (1L << 31), "apla_assemble_tokens: too many rows for one launch");
  if (res_dtype == APLA_F32)
    hipLaunchKernelGGL(assemble_tokens_kernel<float>, dim3(B * (Np + 1)), dim3(256), 0, stream, (const bf16*)patches, ldp, cls_token, pos_embed, (float*)tokens, Np, D, masked, mask_token);
  else if (res_dtype == APLA_H16)
    hipLaunchKernelGGL(assemble_tokens_kernel<bf16>, dim3(B * (Np + 1)), dim3(256), 0, stream, (const bf16*)patches, ldp, cls_token, pos_embed, (bf16*)tokens, Np, D, masked, mask_token);
  else {
    apla_set_error("apla_assemble_tokens: bad res_dtype %d", res_dtype);
    return APLA_ENOSYS;
  }
  APLA_CHECK_LAUNCH("apla_assemble_tokens");
  return APLA_OK;
}

extern "C" int apla_assemble_tokens(const void* patches, int ldp, const float* cls_token, const float* pos_embed,
                                    void* tokens, int res_dtype, int B, int Np, int D, hipStream_t stream) {
  return apla_assemble_tokens_masked(patches, ldp, cls_token, pos_embed, nullptr, nullptr, tokens, res_dtype, B, Np, D, stream);
}

// ---- self-distillation losses (DINO CLS-token loss, iBOT patch loss): rows of K prototypes (65 536 in the shipped config) ----
// softmax_center: out[r,:] = softmax((x[r,:] - center) * inv_temp)   (teacher centering + sharpening; fp32 out)
// distill_ce    : loss_r = -w_r * sum_k t[r,k] * log_softmax(s[r,:] * inv_temp)[k],  ds[r,:] (+)= w_r * inv_temp * (softmax * sum_k t - t)
// One workgroup per row, three passes over the row (max, sum / target sums, write); the row stays in L2.
template <typename T> __device__ __forceinline__ float ldf(const T* p, long i) { return (float)p[i]; }

template <typename XT>
__global__ __launch_bounds__(256) void softmax_center_kernel(const XT* __restrict__ x, long ldx, const float* __restrict__ center,
                                                             float inv_temp, float* __restrict__ out, long ldo, int K) {
  __shared__ float red[4];
  __shared__ float bc;
  const XT* xr = x + (long)blockIdx.x * ldx;
  float* orow = out + (long)blockIdx.x * ldo;
  float mx = -INFINITY;
  for (int k = threadIdx.x; k < K; k += 256) mx = fmaxf(mx, (ldf(xr, k) - center[k]) * inv_temp);
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) bc = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  mx = bc;
  float sm = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) sm += __expf((ldf(xr, k) - center[k]) * inv_temp - mx);
  sm = wave_sum(sm);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sm;
  __syncthreads();
  const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
  for (int k = threadIdx.x; k < K; k += 256) orow[k] = __expf((ldf(xr, k) - center[k]) * inv_temp - mx) * inv;
}

template <typename ST>
__global__ __launch_bounds__(256) void distill_ce_kernel(const ST* __restrict__ s, long lds, const float* __restrict__ t, long ldt,
                                                         float inv_temp, const float* __restrict__ row_weight, float weight,
                                                         float* __restrict__ ds, long ldds, int accumulate,
                                                         float* __restrict__ row_loss, int K, int t_rows) {
  __shared__ float red[4];
  __shared__ float bc;
  const int r = blockIdx.x;
  const ST* sr = s + (long)r * lds;
  const float* tr = t + (long)(r % t_rows) * ldt;
  const float w = row_weight != nullptr ? row_weight[r] * weight : weight;
  float mx = -INFINITY;
  for (int k = threadIdx.x; k < K; k += 256) mx = fmaxf(mx, ldf(sr, k) * inv_temp);
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) bc = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  mx = bc;
  float sm = 0.f, dot = 0.f, tsum = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) {
    const float z = ldf(sr, k) * inv_temp, tv = tr[k];
    sm += __expf(z - mx);
    dot += tv * z;
    tsum += tv;
  }
  float vals[3] = {sm, dot, tsum};
  float tot[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const float v = wave_sum(vals[q]);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    tot[q] = red[0] + red[1] + red[2] + red[3];
  }
  const float lse = mx + __logf(tot[0]);
  if (threadIdx.x == 0) row_loss[r] = w * (tot[2] * lse - tot[1]);
  if (ds != nullptr) {
    float* dr = ds + (long)r * ldds;
    const float c = w * inv_temp;
    for (int k = threadIdx.x; k < K; k += 256) {
      const float g = c * (__expf(ldf(sr, k) * inv_temp - lse) * tot[2] - tr[k]);
      dr[k] = accumulate ? dr[k] + g : g;
    }
  }
}

// Wide versions of the two kernels above (K % 8 == 0, 16-byte aligned rows): 8 prototypes per thread and access, the row's
// statistics in ONE pass (running maximum + rescaled sum per thread, combined across the workgroup in a fixed order), and the row's
// output split over `S` workgroups — each of them walks the whole row for the statistics (it sits in L2 / the Infinity Cache after
// the first) and writes only its own slice.  With 64 student rows per call (DINO: one call per crop) one workgroup per row keeps
// 64 of 256 CUs busy with 2-byte loads: 312 us per call; S = 16 there.
template <typename T> __device__ __forceinline__ void ld8f(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void ld8f<float>(const float* p, float (&v)[8]) {
  const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
  for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
}
template <> __device__ __forceinline__ void ld8f<bf16>(const bf16* p, float (&v)[8]) {
  const bf16x8 a = *(const bf16x8*)p;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (float)a[e];
}
// (m, s) of the workgroup from every thread's running maximum m and sum s of exp(z - m); all threads return the same pair
__device__ __forceinline__ void block_max_sumexp(float& m, float& sm, float* red /* [8] */) {
  const float wm = wave_max(m);
  sm = wave_sum(m == -INFINITY ? 0.f : sm * __expf(m - wm));   // a thread (or a whole wave, K < 2048) that saw no element: m = -inf, sm = 0
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = wm; red[4 + (threadIdx.x >> 6)] = sm; }
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  sm = red[4] * __expf(red[0] - m) + red[5] * __expf(red[1] - m) + red[6] * __expf(red[2] - m) + red[7] * __expf(red[3] - m);
}
__device__ __forceinline__ float block_sum(float v, float* red /* [4] */) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

template <typename XT>
__global__ __launch_bounds__(256) void softmax_center_wide_kernel(const XT* __restrict__ x, long ldx, const float* __restrict__ center,
                                                                  float inv_temp, float* __restrict__ out, long ldo, int K, int slice) {
  __shared__ float red[8];
  const XT* xr = x + (long)blockIdx.y * ldx;
  float* orow = out + (long)blockIdx.y * ldo;
  float m = -INFINITY, sm = 0.f;
  for (int k = threadIdx.x * 8; k < K; k += 2048) {
    float v[8], c[8];
    ld8f<XT>(xr + k, v);
    ld8f<float>(center + k, c);
    float vm = -INFINITY;
#pragma unroll
    for (int e = 0; e < 8; ++e) { v[e] = (v[e] - c[e]) * inv_temp; vm = fmaxf(vm, v[e]); }
    if (vm > m) { sm *= __expf(m - vm); m = vm; }
#pragma unroll
    for (int e = 0; e < 8; ++e) sm += __expf(v[e] - m);
  }
  block_max_sumexp(m, sm, red);
  const float inv = 1.0f / sm;
  const int k0 = blockIdx.x * slice, k1 = k0 + slice < K ? k0 + slice : K;
  for (int k = k0 + threadIdx.x * 8; k < k1; k += 2048) {
    float v[8], c[8];
    ld8f<XT>(xr + k, v);
    ld8f<float>(center + k, c);
    f32x4 o0, o1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o0[e] = __expf((v[e] - c[e]) * inv_temp - m) * inv;
      o1[e] = __expf((v[4 + e] - c[4 + e]) * inv_temp - m) * inv;
    }
    *(f32x4*)(orow + k) = o0;
    *(f32x4*)(orow + k + 4) = o1;
  }
}

template <typename T> __device__ __forceinline__ void st8f(T* p, f32x4 a, f32x4 b, bool accumulate);
template <> __device__ __forceinline__ void st8f<float>(float* p, f32x4 a, f32x4 b, bool accumulate) {
  if (accumulate) { a += *(const f32x4*)p; b += *(const f32x4*)(p + 4); }
  *(f32x4*)p = a;
  *(f32x4*)(p + 4) = b;
}
template <> __device__ __forceinline__ void st8f<bf16>(bf16* p, f32x4 a, f32x4 b, bool accumulate) {
  if (accumulate) {
    const bf16x8 o = *(const bf16x8*)p;
#pragma unroll
    for (int e = 0; e < 4; ++e) { a[e] += (float)o[e]; b[e] += (float)o[4 + e]; }
  }
  bf16x8 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) { v[e] = (bf16)a[e]; v[4 + e] = (bf16)b[e]; }
  *(bf16x8*)p = v;
}

template <typename ST, typename GT>
__global__ __launch_bounds__(256) void distill_ce_wide_kernel(const ST* __restrict__ s, long lds, const float* __restrict__ t, long ldt,
                                                              float inv_temp, const float* __restrict__ row_weight, float weight,
                                                              GT* __restrict__ ds, long ldds, int accumulate,
                                                              float* __restrict__ row_loss, int K, int slice, int t_rows) {
  __shared__ float red[8];
  const int r = blockIdx.y;
  const ST* sr = s + (long)r * lds;
  const float* tr = t + (long)(r % t_rows) * ldt;   // t_rows < R: the targets repeat (the local crops of DINOLoss.forward share theirs)
  const float w = row_weight != nullptr ? row_weight[r] * weight : weight;
  float m = -INFINITY, sm = 0.f, dot = 0.f, tsum = 0.f;
  for (int k = threadIdx.x * 8; k < K; k += 2048) {
    float z[8], tv[8];
    ld8f<ST>(sr + k, z);
    ld8f<float>(tr + k, tv);
    float vm = -INFINITY;
#pragma unroll
    for (int e = 0; e < 8; ++e) { z[e] *= inv_temp; vm = fmaxf(vm, z[e]); }
    if (vm > m) { sm *= __expf(m - vm); m = vm; }
#pragma unroll
    for (int e = 0; e < 8; ++e) { sm += __expf(z[e] - m); dot += tv[e] * z[e]; tsum += tv[e]; }
  }
  block_max_sumexp(m, sm, red);
  dot = block_sum(dot, red);
  tsum = block_sum(tsum, red);
  const float lse = m + __logf(sm);
  if (blockIdx.x == 0 && threadIdx.x == 0) row_loss[r] = w * (tsum * lse - dot);
  if (ds != nullptr) {
    GT* dr = ds + (long)r * ldds;
    const float c = w * inv_temp;
    const int k0 = blockIdx.x * slice, k1 = k0 + slice < K ? k0 + slice : K;
    for (int k = k0 + threadIdx.x * 8; k < k1; k += 2048) {
      float z[8], tv[8];
      ld8f<ST>(sr + k, z);
      ld8f<float>(tr + k, tv);
      f32x4 g0, g1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        g0[e] = c * (__expf(z[e] * inv_temp - lse) * tsum - tv[e]);
        g1[e] = c * (__expf(z[4 + e] * inv_temp - lse) * tsum - tv[4 + e]);
      }
      st8f<GT>(dr + k, g0, g1, accumulate != 0);
    }
  }
}

// (m, s, d) of the workgroup from every thread's running maximum m, sum s of exp(z - m) and weighted sum d of exp(z - m) * y
__device__ __forceinline__ void block_max_sumexp2(float& m, float& sm, float& d, float* red /* [12] */) {
  const float wm = wave_max(m);
  const float sc = m == -INFINITY ? 0.f : __expf(m - wm);
  sm = wave_sum(sm * sc);
  d = wave_sum(d * sc);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = wm; red[4 + (threadIdx.x >> 6)] = sm; red[8 + (threadIdx.x >> 6)] = d; }
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float ws[4];
#pragma unroll
  for (int w = 0; w < 4; ++w) ws[w] = red[w] == -INFINITY ? 0.f : __expf(red[w] - m);
  sm = red[4] * ws[0] + red[5] * ws[1] + red[6] * ws[2] + red[7] * ws[3];
  d = red[8] * ws[0] + red[9] * ws[1] + red[10] * ws[2] + red[11] * ws[3];
}

// distill_ce with the teacher's centring + sharpening inside: t[r,:] = softmax((x[r,:] - center) * inv_temp_t) is never written.
// Pass 1 (whole row, every workgroup of the row): the student's (max, sum exp) and the teacher's (max, sum exp, sum exp * z_student),
// all running and rescaled; pass 2 (this workgroup's slice): ds = w * inv_temp_s * (softmax_s - t).  sum_k t = 1 by construction.
// iBOT at the shipped width moves 10 bytes per logit instead of 22 (softmax_center 2+2+4, distill_ce 2+4 and 2+4+2).
template <typename ST, typename XT, typename GT>
__global__ __launch_bounds__(256) void distill_ce_centered_kernel(const ST* __restrict__ s, long lds, const XT* __restrict__ x, long ldx,
                                                                  const float* __restrict__ center, float inv_temp_s, float inv_temp_t,
                                                                  const float* __restrict__ row_weight, float weight,
                                                                  GT* __restrict__ ds, long ldds, float* __restrict__ row_loss, int K, int slice) {
  __shared__ float red[12];
  const int r = blockIdx.y;
  const ST* sr = s + (long)r * lds;
  const XT* xr = x + (long)r * ldx;
  const float w = row_weight != nullptr ? row_weight[r] * weight : weight;
  float ms = -INFINITY, ss = 0.f, mt = -INFINITY, st = 0.f, dt = 0.f;
  for (int k = threadIdx.x * 8; k < K; k += 2048) {
    float z[8], y[8], c[8];
    ld8f<ST>(sr + k, z);
    ld8f<XT>(xr + k, y);
    ld8f<float>(center + k, c);
    float vs = -INFINITY, vt = -INFINITY;
#pragma unroll
    for (int e = 0; e < 8; ++e) { z[e] *= inv_temp_s; y[e] = (y[e] - c[e]) * inv_temp_t; vs = fmaxf(vs, z[e]); vt = fmaxf(vt, y[e]); }
    if (vs > ms) { ss *= __expf(ms - vs); ms = vs; }
    if (vt > mt) { const float f = __expf(mt - vt); st *= f; dt *= f; mt = vt; }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      ss += __expf(z[e] - ms);
      const float pe = __expf(y[e] - mt);
      st += pe;
      dt += pe * z[e];
    }
  }
  block_max_sumexp(ms, ss, red);
  block_max_sumexp2(mt, st, dt, red);
  const float lse = ms + __logf(ss), inv_t = 1.0f / st;
  if (blockIdx.x == 0 && threadIdx.x == 0) row_loss[r] = w * (lse - dt * inv_t);
  if (ds != nullptr) {
    GT* dr = ds + (long)r * ldds;
    const float cw = w * inv_temp_s;
    const int k0 = blockIdx.x * slice, k1 = k0 + slice < K ? k0 + slice : K;
    for (int k = k0 + threadIdx.x * 8; k < k1; k += 2048) {
      float z[8], y[8], c[8];
      ld8f<ST>(sr + k, z);
      ld8f<XT>(xr + k, y);
      ld8f<float>(center + k, c);
      f32x4 g0, g1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        g0[e] = cw * (__expf(z[e] * inv_temp_s - lse) - __expf((y[e] - c[e]) * inv_temp_t - mt) * inv_t);
        g1[e] = cw * (__expf(z[4 + e] * inv_temp_s - lse) - __expf((y[4 + e] - c[4 + e]) * inv_temp_t - mt) * inv_t);
      }
      st8f<GT>(dr + k, g0, g1, false);
    }
  }
}

// The same for 16-bit rows of at most 65 536 logits, ONE pass over HBM: a workgroup of 16 waves holds its row of student and
// teacher logits in registers (eight 16-byte pieces of each per thread) between the statistics and the gradient.  The two-pass
// kernel above re-reads both rows for the gradient; at iBOT's 4 879 x 65 536 the second read misses every cache (1.2 GB of rows
// in flight): 640 KB per row and 638 us, against 384 KB here.
template <typename HT>
__global__ __launch_bounds__(1024) void distill_ce_centered_row_kernel(const HT* __restrict__ s, long lds, const HT* __restrict__ x, long ldx,
                                                                       const float* __restrict__ center, float inv_temp_s, float inv_temp_t,
                                                                       const float* __restrict__ row_weight, float weight,
                                                                       HT* __restrict__ ds, long ldds, float* __restrict__ row_loss, int K) {
  __shared__ float red[5][16];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const HT* sr = s + (long)r * lds;
  const HT* xr = x + (long)r * ldx;
  const float w = row_weight != nullptr ? row_weight[r] * weight : weight;
  typedef HT h16x8 __attribute__((ext_vector_type(8)));
  h16x8 sv[8], xv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = (j * 1024 + tid) * 8;
    if (k < K) { sv[j] = *(const h16x8*)(sr + k); xv[j] = *(const h16x8*)(xr + k); }
  }
  float ms = -INFINITY, ss = 0.f, mt = -INFINITY, st = 0.f, dt = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = (j * 1024 + tid) * 8;
    if (k >= K) continue;
    float z[8], y[8], c[8];
    ld8f<float>(center + k, c);
    float vs = -INFINITY, vt = -INFINITY;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      z[e] = (float)sv[j][e] * inv_temp_s;
      y[e] = ((float)xv[j][e] - c[e]) * inv_temp_t;
      vs = fmaxf(vs, z[e]); vt = fmaxf(vt, y[e]);
    }
    if (vs > ms) { ss *= __expf(ms - vs); ms = vs; }
    if (vt > mt) { const float f = __expf(mt - vt); st *= f; dt *= f; mt = vt; }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      ss += __expf(z[e] - ms);
      const float pe = __expf(y[e] - mt);
      st += pe;
      dt += pe * z[e];
    }
  }
  // workgroup statistics: per wave, then the sixteen wave results in wave order (deterministic)
  {
    const float wms = wave_max(ms), wmt = wave_max(mt);
    const float fs = ms == -INFINITY ? 0.f : __expf(ms - wms), ft = mt == -INFINITY ? 0.f : __expf(mt - wmt);
    const float wss = wave_sum(ss * fs), wst = wave_sum(st * ft), wdt = wave_sum(dt * ft);
    if (lane == 0) { red[0][wave] = wms; red[1][wave] = wss; red[2][wave] = wmt; red[3][wave] = wst; red[4][wave] = wdt; }
    __syncthreads();
    ms = red[0][0]; mt = red[2][0];
#pragma unroll
    for (int v = 1; v < 16; ++v) { ms = fmaxf(ms, red[0][v]); mt = fmaxf(mt, red[2][v]); }
    ss = 0.f; st = 0.f; dt = 0.f;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const float a = red[0][v] == -INFINITY ? 0.f : __expf(red[0][v] - ms), b = red[2][v] == -INFINITY ? 0.f : __expf(red[2][v] - mt);
      ss += red[1][v] * a; st += red[3][v] * b; dt += red[4][v] * b;
    }
  }
  const float lse = ms + __logf(ss), inv_t = 1.0f / st;
  if (tid == 0) row_loss[r] = w * (lse - dt * inv_t);
  if (ds == nullptr) return;
  HT* dr = ds + (long)r * ldds;
  const float cw = w * inv_temp_s;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = (j * 1024 + tid) * 8;
    if (k >= K) continue;
    float c[8];
    ld8f<float>(center + k, c);
    f32x4 g0, g1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      g0[e] = cw * (__expf((float)sv[j][e] * inv_temp_s - lse) - __expf(((float)xv[j][e] - c[e]) * inv_temp_t - mt) * inv_t);
      g1[e] = cw * (__expf((float)sv[j][4 + e] * inv_temp_s - lse) - __expf(((float)xv[j][4 + e] - c[4 + e]) * inv_temp_t - mt) * inv_t);
    }
    st8f<HT>(dr + k, g0, g1, false);
  }
}

// workgroups per row: enough of them to fill the chip when the call has few rows; a slice is a multiple of one workgroup pass
static inline int wide_row_split(int R, int K, int* slice) {
  int S = R >= 512 ? 1 : (1024 + R - 1) / R;
  S = S > 16 ? 16 : S;
  int len = ((K + S - 1) / S + 2047) / 2048 * 2048;
  *slice = len;
  return (K + len - 1) / len;
}

extern "C" int apla_softmax_center(const void* x, int x_dtype, long ldx, const float* center, float inv_temp, float* out,
                                   long ldo, int R, int K, hipStream_t stream) {
  APLA_REQUIRE(x && center && out && R > 0 && K > 0 && ldx >= K && ldo >= K && inv_temp > 0.f, "apla_softmax_center: bad arguments");
  const bool wide = K % 8 == 0 && R <= 65535 && apla_aligned16(x) && apla_aligned16(center) && apla_aligned16(out) && ldo % 4 == 0 &&
                    ldx % (x_dtype == APLA_F32 ? 4 : 8) == 0 && (x_dtype == APLA_F32 || x_dtype == APLA_H16);
  if (wide) {
    int slice = 0;
    const int S = wide_row_split(R, K, &slice);
    if (x_dtype == APLA_F32) hipLaunchKernelGGL(softmax_center_wide_kernel<float>, dim3(S, R), dim3(256), 0, stream, (const float*)x, ldx, center, inv_temp, out, ldo, K, slice);
    else hipLaunchKernelGGL(softmax_center_wide_kernel<bf16>, dim3(S, R), dim3(256), 0, stream, (const bf16*)x, ldx, center, inv_temp, out, ldo, K, slice);
    APLA_CHECK_LAUNCH("apla_softmax_center");
    return APLA_OK;
  }
  if (x_dtype == APLA_F32) hipLaunchKernelGGL(softmax_center_kernel<float>, dim3(R), dim3(256), 0, stream, (const float*)x, ldx, center, inv_temp, out, ldo, K);
  else if (x_dtype == APLA_H16) hipLaunchKernelGGL(softmax_center_kernel<bf16>, dim3(R), dim3(256), 0, stream, (const bf16*)x, ldx, center, inv_temp, out, ldo, K);
  else { apla_set_error("apla_softmax_center: unsupported dtype %d", x_dtype); return APLA_ENOSYS; }
  APLA_CHECK_LAUNCH("apla_softmax_center");
  return APLA_OK;
}

extern "C" int apla_distill_ce_bcast(const void* student, int s_dtype, long lds, const float* teacher_probs, long ldt, int t_rows,
                                     float inv_temp, const float* row_weight, float weight, void* dstudent, int ds_dtype, long ldds,
                                     int accumulate, float* row_loss, int R, int K, hipStream_t stream) {
  APLA_REQUIRE(student && teacher_probs && row_loss && R > 0 && K > 0 && lds >= K && ldt >= K && inv_temp > 0.f && t_rows > 0 &&
               t_rows <= R && (dstudent == nullptr || ldds >= K), "apla_distill_ce: bad arguments");
  APLA_REQUIRE((s_dtype == APLA_F32 || s_dtype == APLA_H16) && (dstudent == nullptr || ds_dtype == APLA_F32 || ds_dtype == APLA_H16),
               "apla_distill_ce: unsupported dtype (student %d, gradient %d)", s_dtype, ds_dtype);
  const bool g16 = dstudent != nullptr && ds_dtype == APLA_H16;
  const bool wide = K % 8 == 0 && R <= 65535 && apla_aligned16(student) && apla_aligned16(teacher_probs) && ldt % 4 == 0 &&
                    (dstudent == nullptr || (apla_aligned16(dstudent) && ldds % (g16 ? 8 : 4) == 0)) &&
                    lds % (s_dtype == APLA_F32 ? 4 : 8) == 0;
  APLA_REQUIRE(wide || !g16, "apla_distill_ce_ex: a 16-bit gradient needs K %% 8 == 0 and 16-byte aligned rows");
  if (wide) {
    int slice = K;
    const int S = dstudent != nullptr ? wide_row_split(R, K, &slice) : 1;
#define DCE_WIDE(ST, GT) hipLaunchKernelGGL((distill_ce_wide_kernel<ST, GT>), dim3(S, R), dim3(256), 0, stream, (const ST*)student, lds, teacher_probs, ldt, inv_temp, row_weight, weight, (GT*)dstudent, ldds, accumulate, row_loss, K, slice, t_rows)
    if (s_dtype == APLA_F32) { if (g16) DCE_WIDE(float, bf16); else DCE_WIDE(float, float); }
    else { if (g16) DCE_WIDE(bf16, bf16); else DCE_WIDE(bf16, float); }
#undef DCE_WIDE
    APLA_CHECK_LAUNCH("apla_distill_ce");
    return APLA_OK;
  }
  if (s_dtype == APLA_F32) hipLaunchKernelGGL(distill_ce_kernel<float>, dim3(R), dim3(256), 0, stream, (const float*)student, lds, teacher_probs, ldt, inv_temp, row_weight, weight, (float*)dstudent, ldds, accumulate, row_loss, K, t_rows);
  else hipLaunchKernelGGL(distill_ce_kernel<bf16>, dim3(R), dim3(256), 0, stream, (const bf16*)student, lds, teacher_probs, ldt, inv_temp, row_weight, weight, (float*)dstudent, ldds, accumulate, row_loss, K, t_rows);
  APLA_CHECK_LAUNCH("apla_distill_ce");
  return APLA_OK;
}

extern "C" int apla_distill_ce_ex(const void* student, int s_dtype, long lds, const float* teacher_probs, long ldt, float inv_temp,
                                  const float* row_weight, float weight, void* dstudent, int ds_dtype, long ldds, int accumulate,
                                  float* row_loss, int R, int K, hipStream_t stream) {
  return apla_distill_ce_bcast(student, s_dtype, lds, teacher_probs, ldt, R, inv_temp, row_weight, weight, dstudent, ds_dtype, ldds,
                               accumulate, row_loss, R, K, stream);
}

extern "C" int apla_distill_ce(const void* student, int s_dtype, long lds, const float* teacher_probs, long ldt, float inv_temp,
                               const float* row_weight, float weight, float* dstudent, long ldds, int accumulate,
                               float* row_loss, int R, int K, hipStream_t stream) {
  return apla_distill_ce_ex(student, s_dtype, lds, teacher_probs, ldt, inv_temp, row_weight, weight, dstudent, APLA_F32, ldds, accumulate,
                            row_loss, R, K, stream);
}

extern "C" int apla_distill_ce_centered(const void* student, int s_dtype, long lds, const void* teacher_logits, int x_dtype, long ldx,
                                        const float* center, float inv_temp_s, float inv_temp_t, const float* row_weight, float weight,
                                        void* dstudent, int ds_dtype, long ldds, float* row_loss, int R, int K, hipStream_t stream) {
  APLA_REQUIRE(student && teacher_logits && center && row_loss && R > 0 && R <= 65535 && K > 0 && K % 8 == 0 && lds >= K && ldx >= K &&
               inv_temp_s > 0.f && inv_temp_t > 0.f && (dstudent == nullptr || ldds >= K), "apla_distill_ce_centered: bad arguments");
  APLA_REQUIRE((s_dtype == APLA_F32 || s_dtype == APLA_H16) && (x_dtype == APLA_F32 || x_dtype == APLA_H16) &&
               (dstudent == nullptr || ds_dtype == APLA_F32 || ds_dtype == APLA_H16), "apla_distill_ce_centered: unsupported dtype");
  const bool g16 = dstudent != nullptr && ds_dtype == APLA_H16;
  APLA_REQUIRE(apla_aligned16(student) && apla_aligned16(teacher_logits) && apla_aligned16(center) && lds % (s_dtype == APLA_F32 ? 4 : 8) == 0 &&
               ldx % (x_dtype == APLA_F32 ? 4 : 8) == 0 && (dstudent == nullptr || (apla_aligned16(dstudent) && ldds % (g16 ? 8 : 4) == 0)),
               "apla_distill_ce_centered: rows must be 16-byte aligned");
  if (s_dtype == APLA_H16 && x_dtype == APLA_H16 && (dstudent == nullptr || g16) && K >= 8192 && K <= 65536 && R >= 256) {
    hipLaunchKernelGGL(distill_ce_centered_row_kernel<bf16>, dim3(R), dim3(1024), 0, stream, (const bf16*)student, lds, (const bf16*)teacher_logits, ldx,
                       center, inv_temp_s, inv_temp_t, row_weight, weight, (bf16*)dstudent, ldds, row_loss, K);
    APLA_CHECK_LAUNCH("apla_distill_ce_centered");
    return APLA_OK;
  }
  int slice = K;
  const int S = dstudent != nullptr ? wide_row_split(R, K, &slice) : 1;
#define DCC(ST, XT, GT) hipLaunchKernelGGL((distill_ce_centered_kernel<ST, XT, GT>), dim3(S, R), dim3(256), 0, stream, (const ST*)student, lds, (const XT*)teacher_logits, ldx, center, inv_temp_s, inv_temp_t, row_weight, weight, (GT*)dstudent, ldds, row_loss, K, slice)
#define DCC_G(ST, XT) do { if (g16) DCC(ST, XT, bf16); else DCC(ST, XT, float); } while (0)
  if (s_dtype == APLA_F32) { if (x_dtype == APLA_F32) DCC_G(float, float); else DCC_G(float, bf16); }
  else { if (x_dtype == APLA_F32) DCC_G(bf16, float); else DCC_G(bf16, bf16); }
#undef DCC_G
#undef DCC
  APLA_CHECK_LAUNCH("apla_distill_ce_centered");
  return APLA_OK;
}

// ---- input side: uint8 images -> normalised fp32 batch, per-sample horizontal flip, Mixup / CutMix against a partner ----
// dst[b,c,y,x] = norm(src[b,c,y,fx_b]) mixed with the partner sample p = perm[b] (its own flip flag applies to it):
//   box == nullptr (Mixup):  lam_b * own + (1 - lam_b) * partner
//   box != nullptr (CutMix): partner inside [y0,y1) x [x0,x1) of box[b], own elsewhere
// One thread = four consecutive x of one (b, c, y) row; src is CHW (hwc = 0) or HWC (hwc = 1) uint8.
__global__ __launch_bounds__(256) void augment_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst,
                                                      float m0, float m1, float m2, float s0, float s1, float s2,
                                                      const uint8_t* __restrict__ flip, const int32_t* __restrict__ perm,
                                                      const float* __restrict__ lam, const int32_t* __restrict__ box,
                                                      int B, int S, int hwc) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int xq = S / 4;
  if (t >= (long)B * 3 * S * xq) return;
  const int x0 = (int)(t % xq) * 4;
  const int y = (int)((t / xq) % S), c = (int)((t / ((long)xq * S)) % 3), b = (int)(t / ((long)xq * S * 3));
  const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), inv = 1.0f / (255.0f * (c == 0 ? s0 : (c == 1 ? s1 : s2)));
  const float mo = mean / (c == 0 ? s0 : (c == 1 ? s1 : s2));
  auto px = [&](int bb, int x) {
    const int fx = (flip != nullptr && flip[bb]) ? S - 1 - x : x;
    const long idx = hwc ? (((long)bb * S + y) * S + fx) * 3 + c : (((long)bb * 3 + c) * S + y) * S + fx;
    return (float)src[idx] * inv - mo;
  };
  const int pb = perm != nullptr ? perm[b] : b;
  const float l = lam != nullptr ? lam[b] : 1.0f;
  f32x4 out;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int x = x0 + e;
    float v = px(b, x);
    if (perm != nullptr) {
      if (box != nullptr) {
        const int* bx = box + 4 * b;
        if (y >= bx[0] && y < bx[1] && x >= bx[2] && x < bx[3]) v = px(pb, x);
      } else if (l != 1.0f) {
        v = l * v + (1.0f - l) * px(pb, x);
      }
    }
    out[e] = v;
  }
  *(f32x4*)(dst + (((long)b * 3 + c) * S + y) * S + x0) = out;
}

extern "C" int apla_augment_images(const uint8_t* src, float* dst, const float* mean3, const float* std3,
                                   const uint8_t* flip, const int32_t* perm, const float* lam, const int32_t* box,
                                   int B, int S, int hwc, hipStream_t stream) {
  APLA_REQUIRE(src && dst && mean3 && std3 && B > 0 && S > 0 && S % 4 == 0, "apla_augment_images: bad arguments (S %% 4 == 0 required)");
  APLA_REQUIRE(apla_aligned16(dst) && (box == nullptr || perm != nullptr) && (lam == nullptr || perm != nullptr),
               "apla_augment_images: dst must be 16-byte aligned; lam / box need perm");
  APLA_REQUIRE(std3[0] > 0.f && std3[1] > 0.f && std3[2] > 0.f, "apla_augment_images: std must be positive");
  const long n = (long)B * 3 * S * (S / 4);
  hipLaunchKernelGGL(augment_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src, dst, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], flip, perm, lam, box, B, S, hwc);
  APLA_CHECK_LAUNCH("apla_augment_images");
  return APLA_OK;
}

extern "C" int apla_sgemm_small(const float* A, long sai, long sak, const float* Bm, long sbk, long sbj,
                                const float* bias, float* C, long ldc, int M, int N, int K, int accumulate,
                                hipStream_t stream) {
  APLA_REQUIRE(A && Bm && C && M > 0 && N > 0 && K > 0, "apla_sgemm_small: bad arguments");
  hipLaunchKernelGGL(sgemm_small_kernel, dim3((N + 31) / 32, (M + 31) / 32), dim3(64 * SG_WAVES), 0, stream, A, sai, sak, Bm, sbk, sbj, bias, C, ldc, M, N, K, accumulate);
  APLA_CHECK_LAUNCH("apla_sgemm_small");
  return APLA_OK;
}

extern "C" int apla_cross_entropy(const float* logits, int ldl, const int32_t* labels, float* dlogits,
                                  float* row_loss, float* loss, int B, int C, hipStream_t stream) {
  APLA_REQUIRE(logits && labels && dlogits && row_loss && loss && B > 0 && C > 0 && ldl >= C, "apla_cross_entropy: bad arguments");
  hipLaunchKernelGGL(cross_entropy_kernel, dim3(B), dim3(256), 0, stream, logits, ldl, labels, (const float*)nullptr, 0, dlogits, row_loss, B, C);
  APLA_CHECK_LAUNCH("apla_cross_entropy");
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, stream, (const float*)row_loss, loss, B);
  APLA_CHECK_LAUNCH("apla_cross_entropy[mean]");
  return APLA_OK;
}

extern "C" int apla_cross_entropy_soft(const float* logits, int ldl, const float* targets, int ldt, float* dlogits,
                                       float* row_loss, float* loss, int B, int C, hipStream_t stream) {
  APLA_REQUIRE(logits && targets && dlogits && row_loss && loss && B > 0 && C > 0 && ldl >= C && ldt >= C, "apla_cross_entropy_soft: bad arguments");
  hipLaunchKernelGGL(cross_entropy_kernel, dim3(B), dim3(256), 0, stream, logits, ldl, (const int32_t*)nullptr, targets, ldt, dlogits, row_loss, B, C);
  APLA_CHECK_LAUNCH("apla_cross_entropy_soft");
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, stream, (const float*)row_loss, loss, B);
  APLA_CHECK_LAUNCH("apla_cross_entropy_soft[mean]");
  return APLA_OK;
}

extern "C" int apla_colsum(const float* X, long ld, float* out, int M, int N, hipStream_t stream) {
  APLA_REQUIRE(X && out && M > 0 && N > 0, "apla_colsum: bad arguments");
  hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64), dim3(256), 0, stream, X, ld, out, M, N);
  APLA_CHECK_LAUNCH("apla_colsum");
  return APLA_OK;
}

// out[j] = sum_i X[i][j] for a 16-bit X (N % 4 == 0, rows 8-byte aligned) — fp32 accumulation
extern "C" int apla_colsum_h16(const void* X, long ld, float* out, int M, int N, hipStream_t stream) {
  APLA_REQUIRE(X && out && M > 0 && N > 0 && N % 4 == 0 && ld % 4 == 0 && ld >= N && (((uintptr_t)X) & 7) == 0 && apla_aligned16(out),
               "apla_colsum_h16: [M, N] 16-bit matrix with N %% 4 == 0 and 8-byte aligned rows");
  hipLaunchKernelGGL(colsum16_kernel, dim3((N + 255) / 256), dim3(1024), 0, stream, (const bf16*)X, ld, out, M, N);
  APLA_CHECK_LAUNCH("apla_colsum_h16");
  return APLA_OK;
}

// ------------------------------------------------------------------------------------------------ weight normalisation
// torch.nn.utils.weight_norm(dim = 0) of the DINO head's prototype layer (dinov2/layers/dino_head.py:27-28): W[i, :] = v[i, :] * g[i] / ||v[i, :]||.
// One wave per row.  Forward writes W in the GEMM's 16-bit operand type (the fp32 W of the torch route is only ever cast) and the row
// norms; backward turns dW (fp32, from the dW kernel) into dv = (g / n) (dW - v (dW . v) / n^2) and dg = (dW . v) / n — the formulas
// of torch's _weight_norm_interface_backward.  torch runs three element-wise kernels forward and about ten backward over the [65 536, 256]
// matrices of the shipped head.
__global__ __launch_bounds__(256) void weight_norm_fwd_kernel(const float* __restrict__ v, const float* __restrict__ g, bf16* __restrict__ w,
                                                              float* __restrict__ norm, int K, int D) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= K) return;
  const float* vr = v + (size_t)row * D;
  float ss = 0.f;
  for (int c = lane * 4; c < D; c += 256) {
    const f32x4 a = *(const f32x4*)(vr + c);
    ss += a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3];
  }
  const float n = sqrtf(wave_sum(ss));
  const float sc = g[row] / n;
  for (int c = lane * 4; c < D; c += 256) {
    const f32x4 a = *(const f32x4*)(vr + c);
    Vec4IO<bf16>::store(w + (size_t)row * D + c, a * sc);
  }
  if (lane == 0) norm[row] = n;
}

__global__ __launch_bounds__(256) void weight_norm_bwd_kernel(const float* __restrict__ dw, const float* __restrict__ v, const float* __restrict__ g,
                                                              const float* __restrict__ norm, float* __restrict__ dv, float* __restrict__ dg,
                                                              int K, int D) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= K) return;
  const float* vr = v + (size_t)row * D;
  const float* dr = dw + (size_t)row * D;
  float dot = 0.f;
  for (int c = lane * 4; c < D; c += 256) {
    const f32x4 a = *(const f32x4*)(vr + c), d = *(const f32x4*)(dr + c);
    dot += a[0] * d[0] + a[1] * d[1] + a[2] * d[2] + a[3] * d[3];
  }
  dot = wave_sum(dot);
  const float n = norm[row], gi = g[row];
  const float a1 = gi / n, a2 = gi * dot / (n * n * n);
  for (int c = lane * 4; c < D; c += 256) {
    const f32x4 a = *(const f32x4*)(vr + c), d = *(const f32x4*)(dr + c);
    *(f32x4*)(dv + (size_t)row * D + c) = d * a1 - a * a2;
  }
  if (lane == 0 && dg != nullptr) dg[row] = dot / n;
}

extern "C" int apla_weight_norm_fwd(const float* v, const float* g, void* w_h16, float* norm, int K, int D, hipStream_t stream) {
  APLA_REQUIRE(v && g && w_h16 && norm && K > 0 && D > 0 && D % 4 == 0, "apla_weight_norm_fwd: need D %% 4 == 0 (K=%d D=%d)", K, D);
  APLA_REQUIRE(apla_aligned16(v) && apla_aligned16(w_h16), "apla_weight_norm_fwd: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(weight_norm_fwd_kernel, dim3((K + 3) / 4), dim3(256), 0, stream, v, g, (bf16*)w_h16, norm, K, D);
  APLA_CHECK_LAUNCH("apla_weight_norm_fwd");
  return APLA_OK;
}

// The same with W^T [D, K] written beside W (the dX GEMM of the prototype layer reads W through its transposed copy; torch's
// .t().contiguous() of the [65 536, 256] matrix takes 73 us per iteration, this kernel's extra store ~10): a workgroup takes 64 rows,
// one wave per row as above with the 16-bit row also kept in LDS, then the tile leaves column by column in 128-byte runs of 64 rows.
__global__ __launch_bounds__(256) void weight_norm_fwd_t_kernel(const float* __restrict__ v, const float* __restrict__ g, bf16* __restrict__ w,
                                                                bf16* __restrict__ wt, float* __restrict__ norm, int K, int D) {
  extern __shared__ bf16 s_tile[];   // [64][D + 8]: rows 16 bytes apart modulo the banks
  const int ldt = D + 8, r0 = blockIdx.x * 64, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int rr = wave; rr < 64; rr += 4) {
    const int row = r0 + rr;
    const float* vr = v + (size_t)row * D;
    float ss = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
      const f32x4 a = *(const f32x4*)(vr + c);
      ss += a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3];
    }
    const float n = sqrtf(wave_sum(ss));
    const float sc = g[row] / n;
    for (int c = lane * 4; c < D; c += 256) {
      const f32x4 a = *(const f32x4*)(vr + c) * sc;
      const bf16x4 h = pack4(a[0], a[1], a[2], a[3]);
      *(bf16x4*)(w + (size_t)row * D + c) = h;
      *(bf16x4*)(s_tile + rr * ldt + c) = h;
    }
    if (lane == 0) norm[row] = n;
  }
  __syncthreads();
  const int sub = threadIdx.x & 7;   // 8 threads per column: rows sub * 8 .. sub * 8 + 7
  for (int c = threadIdx.x >> 3; c < D; c += 32) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = s_tile[(sub * 8 + e) * ldt + c];
    *(bf16x8*)(wt + (size_t)c * K + r0 + sub * 8) = o;
  }
}

extern "C" int apla_weight_norm_fwd_t(const float* v, const float* g, void* w_h16, void* wt_h16, float* norm, int K, int D, hipStream_t stream) {
  if (wt_h16 == nullptr) return apla_weight_norm_fwd(v, g, w_h16, norm, K, D, stream);
  APLA_REQUIRE(v && g && w_h16 && norm && K > 0 && K % 64 == 0 && D > 0 && D % 4 == 0 && D <= 1024,
               "apla_weight_norm_fwd_t: need K %% 64 == 0, D %% 4 == 0, D <= 1024 (K=%d D=%d)", K, D);
  APLA_REQUIRE(apla_aligned16(v) && apla_aligned16(w_h16) && apla_aligned16(wt_h16), "apla_weight_norm_fwd_t: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(weight_norm_fwd_t_kernel, dim3(K / 64), dim3(256), (size_t)64 * (D + 8) * 2, stream, v, g, (bf16*)w_h16, (bf16*)wt_h16,
                     norm, K, D);
  APLA_CHECK_LAUNCH("apla_weight_norm_fwd_t");
  return APLA_OK;
}

extern "C" int apla_weight_norm_bwd(const float* dw, const float* v, const float* g, const float* norm, float* dv, float* dg, int K, int D,
                                    hipStream_t stream) {
  APLA_REQUIRE(dw && v && g && norm && dv && K > 0 && D > 0 && D % 4 == 0, "apla_weight_norm_bwd: need D %% 4 == 0 (K=%d D=%d)", K, D);
  APLA_REQUIRE(apla_aligned16(dw) && apla_aligned16(v) && apla_aligned16(dv), "apla_weight_norm_bwd: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(weight_norm_bwd_kernel, dim3((K + 3) / 4), dim3(256), 0, stream, dw, v, g, norm, dv, dg, K, D);
  APLA_CHECK_LAUNCH("apla_weight_norm_bwd");
  return APLA_OK;
}

// ------------------------------------------------------------------------------------------------ KoLeo regulariser
// KoLeoLoss (self_supervised/dinov2/loss/koleo_loss.py:17-45) of G groups of B rows (the two global crops, models.py:410-413), fp32:
//   xn = x / max(||x||, eps);  j(i) = argmax_{j != i} xn_i . xn_j (within the group);  d_i = ||xn_i - xn_j(i) + 1e-8||;
//   loss_g = -mean_i log(d_i + eps);  out[g] = loss_g, out[G] = sum_g loss_g.
// torch runs ~30 small kernels per group forward and as many backward (a [B, B] product, a diagonal fill, a max, a gather, the pairwise
// distance, the log and the mean: 0.3 ms of a 42 ms iteration at 4 us each); here one workgroup per row does the row's search and its
// distance, the last workgroup to finish adds the terms in a fixed order (a ticket in `counter`, which it resets), and one backward
// launch turns d(out[G]) into dx: the gradient of row i collects its own term and those of the rows whose neighbour it is, then goes
// through the normalisation.
// The neighbour is the row at the smallest ||xn_i - xn_j||^2, taken directly: for unit vectors that IS the largest inner product the
// reference searches (d^2 = 2 - 2 xn_i . xn_j), but it keeps its meaning when the rows of a group nearly coincide — CLS tokens at
// initialisation sit 1e-4 apart, their fp32 inner products all round to 1 - {0, 1} ulp, and the reference's argmax then follows the
// rounding of its GEMM, not the geometry (no two implementations of that product agree there; away from that regime the choice is the same).
// (A row whose norm is under eps stays shorter than 1 after the division; its distance is corrected by 1 - ||xn_j||^2 so that the
// order is still the order of the inner products.)
// sum of squares of one row, one wave, the same roundings at every call site (a row's norm is taken in several places and must be ONE value)
template <typename T>
__device__ __forceinline__ float koleo_row_ss(const T* __restrict__ p, int D, int lane) {
#pragma clang fp contract(off)
  float ss = 0.f;
  for (int c = lane * 4; c < D; c += 256) {
    const f32x4 a = Vec4IO<T>::load(p + c);
    ss = ss + a[0] * a[0];
    ss = ss + a[1] * a[1];
    ss = ss + a[2] * a[2];
    ss = ss + a[3] * a[3];
  }
  return wave_sum(ss);
}

// xn_i - xn_j with each product rounded on its own (no fma contraction: identical rows must give exactly 0, as torch's separate passes do)
__device__ __forceinline__ f32x4 koleo_diff(f32x4 a, float ia, f32x4 b, float ib) {
#pragma clang fp contract(off)   // (hip's __fmul_rn / __fsub_rn are plain operators: they would still contract into one fma)
  const f32x4 pa = a * ia, pb = b * ib;
  return pa - pb;
}

template <typename T>
__global__ __launch_bounds__(256) void koleo_fwd_kernel(const T* __restrict__ x, int G, int B, int D, float eps, int* __restrict__ nn_idx,
                                                        float* __restrict__ dist, float* __restrict__ nrm, float* __restrict__ terms,
                                                        float* __restrict__ out, int* __restrict__ counter) {
  __shared__ float s_val[4];
  __shared__ int s_idx[4];
  __shared__ float s_red[4];
  __shared__ int s_last;
  const int row = blockIdx.x, g = row / B, i = row - g * B, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const T* xg = x + (size_t)g * B * D;
  const T* xi = xg + (size_t)i * D;
  const float ni = __fsqrt_rn(koleo_row_ss(xi, D, lane)), inv_i = __frcp_rn(fmaxf(ni, eps));   // (explicitly rounded: no rsqrt fusion)
  float best = 3.4e38f;
  int bj = (i == 0 && B > 1) ? 1 : 0;
  for (int j = wave; j < B; j += 4) {
    if (j == i) continue;
    const T* xj = xg + (size_t)j * D;
    const float ssj = koleo_row_ss(xj, D, lane), nj = __fsqrt_rn(ssj), inv_j = __frcp_rn(fmaxf(nj, eps));
    float q = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
      const f32x4 t = koleo_diff(Vec4IO<T>::load(xi + c), inv_i, Vec4IO<T>::load(xj + c), inv_j);
      q += t[0] * t[0] + t[1] * t[1] + t[2] * t[2] + t[3] * t[3];
    }
    q = wave_sum(q);
    if (!(nj > eps)) q += 1.f - ssj * inv_j * inv_j;   // a row under the clamp is not a unit vector: rank it by its inner product all the same
    if (q < best) { best = q; bj = j; }    // ascending j within the wave: the first minimum stays
  }
  if (lane == 0) { s_val[wave] = best; s_idx[wave] = bj; }
  __syncthreads();
  best = s_val[0]; bj = s_idx[0];
#pragma unroll
  for (int w = 1; w < 4; ++w)
    if (s_val[w] < best || (s_val[w] == best && s_idx[w] < bj)) { best = s_val[w]; bj = s_idx[w]; }
  if (B == 1) bj = 0;   // torch.max over the one-element row [-1] picks the row itself
  const T* xj = xg + (size_t)bj * D;
  const float inv_j = __frcp_rn(fmaxf(__fsqrt_rn(koleo_row_ss(xj, D, lane)), eps));
  float d2 = 0.f;
  for (int c = threadIdx.x * 4; c < D; c += 1024) {
    const f32x4 df = koleo_diff(Vec4IO<T>::load(xi + c), inv_i, Vec4IO<T>::load(xj + c), inv_j);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float t = __fadd_rn(df[e], 1e-8f);    // nn.PairwiseDistance adds its eps to the difference
      d2 += t * t;
    }
  }
  d2 = wave_sum(d2);
  if (lane == 0) s_red[wave] = d2;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float d = sqrtf(s_red[0] + s_red[1] + s_red[2] + s_red[3]);
    nn_idx[row] = bj;
    dist[row] = d;
    nrm[row] = ni;
    terms[row] = -logf(d + eps);
    __threadfence();
    s_last = atomicAdd(counter, 1) == (int)gridDim.x - 1;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  if (wave == 0) {   // the last workgroup: every term is written; one wave adds them group by group in a fixed order
    const volatile float* tv = terms;
    float total = 0.f;
    for (int gg = 0; gg < G; ++gg) {
      float a = 0.f;
      for (int r = lane; r < B; r += 64) a += tv[gg * B + r];
      a = wave_sum(a) / (float)B;
      if (lane == 0) out[gg] = a;
      total += a;
    }
    if (lane == 0) { out[G] = total; *counter = 0; }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void koleo_bwd_kernel(const T* __restrict__ x, int B, int D, float eps, const int* __restrict__ nn_idx,
                                                        const float* __restrict__ dist, const float* __restrict__ nrm,
                                                        const float* __restrict__ gout, T* __restrict__ dx) {
#pragma clang fp contract(off)       // each term rounded on its own: a row that is its own neighbour (B == 1) must get exactly 0, as in torch
  extern __shared__ float s_dyn[];   // per row of the group: coefficient -g / (B (d + eps) d), 1 / max(n, eps); then the neighbour index
  float* s_coef = s_dyn;
  float* s_inv = s_dyn + B;
  int* s_nn = (int*)(s_dyn + 2 * B);
  __shared__ float s_red[4];
  const int row = blockIdx.x, g = row / B, i = row - g * B, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float go = gout[0];
  for (int r = threadIdx.x; r < B; r += 256) {
    const float d = dist[g * B + r];
    s_coef[r] = -go / ((float)B * (d + eps) * d);
    s_inv[r] = __frcp_rn(fmaxf(nrm[g * B + r], eps));
    s_nn[r] = nn_idx[g * B + r];
  }
  __syncthreads();
  const T* xg = x + (size_t)g * B * D;
  const T* xi = xg + (size_t)i * D;
  const float inv_i = s_inv[i], ni = nrm[row];
  f32x4 acc[4];   // d(loss) / d(xn_i), D <= 4096
  float dot = 0.f;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int c = threadIdx.x * 4 + m * 1024;
    acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c >= D) continue;
    const f32x4 xa = Vec4IO<T>::load(xi + c);
    const f32x4 a = xa * inv_i;
    {
      const int j = s_nn[i];
      const f32x4 df = koleo_diff(xa, inv_i, Vec4IO<T>::load(xg + (size_t)j * D + c), s_inv[j]);
      acc[m] += (df + 1e-8f) * s_coef[i];
    }
    for (int k = 0; k < B; ++k) {
      if (s_nn[k] != i) continue;   // (k == i only when B == 1: the two terms of the row cancel, as in torch)
      const f32x4 df = koleo_diff(Vec4IO<T>::load(xg + (size_t)k * D + c), s_inv[k], xa, inv_i);
      acc[m] -= (df + 1e-8f) * s_coef[k];
    }
    dot += acc[m][0] * a[0] + acc[m][1] * a[1] + acc[m][2] * a[2] + acc[m][3] * a[3];
  }
  dot = wave_sum(dot);
  if (lane == 0) s_red[wave] = dot;
  __syncthreads();
  dot = s_red[0] + s_red[1] + s_red[2] + s_red[3];
  const bool clamped = !(ni > eps);     // F.normalize divides by clamp_min(n, eps): no gradient through a clamped norm
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int c = threadIdx.x * 4 + m * 1024;
    if (c >= D) continue;
    const f32x4 a = Vec4IO<T>::load(xi + c) * inv_i;
    Vec4IO<T>::store(dx + (size_t)row * D + c, clamped ? acc[m] * inv_i : (acc[m] - a * dot) * inv_i);
  }
}

extern "C" int apla_koleo_fwd(const void* x, int dtype, int G, int B, int D, float eps, int* nn_idx, float* dist, float* nrm, float* terms,
                              float* out, int* counter, hipStream_t stream) {
  APLA_REQUIRE(x && nn_idx && dist && nrm && terms && out && counter && G > 0 && B > 0 && D > 0 && D % 4 == 0 && (long)G * B <= (1 << 20),
               "apla_koleo_fwd: need G, B > 0 and D %% 4 == 0 (G=%d B=%d D=%d)", G, B, D);
  APLA_REQUIRE((((uintptr_t)x) & 15) == 0 && (dtype == APLA_F32 || dtype == APLA_H16), "apla_koleo_fwd: x fp32 or the build's 16-bit type, 16-byte aligned");
  if (dtype == APLA_F32)
    hipLaunchKernelGGL(koleo_fwd_kernel<float>, dim3(G * B), dim3(256), 0, stream, (const float*)x, G, B, D, eps, nn_idx, dist, nrm, terms, out, counter);
  else
    hipLaunchKernelGGL(koleo_fwd_kernel<bf16>, dim3(G * B), dim3(256), 0, stream, (const bf16*)x, G, B, D, eps, nn_idx, dist, nrm, terms, out, counter);
  APLA_CHECK_LAUNCH("apla_koleo_fwd");
  return APLA_OK;
}

extern "C" int apla_koleo_bwd(const void* x, int dtype, int G, int B, int D, float eps, const int* nn_idx, const float* dist, const float* nrm,
                              const float* gout, void* dx, hipStream_t stream) {
  APLA_REQUIRE(x && nn_idx && dist && nrm && gout && dx && G > 0 && B > 0 && B <= 4096 && D > 0 && D % 4 == 0 && D <= 4096,
               "apla_koleo_bwd: need G > 0, 0 < B <= 4096, D %% 4 == 0, D <= 4096 (G=%d B=%d D=%d)", G, B, D);
  APLA_REQUIRE((((uintptr_t)x) & 15) == 0 && (((uintptr_t)dx) & 15) == 0 && (dtype == APLA_F32 || dtype == APLA_H16),
               "apla_koleo_bwd: x / dx fp32 or the build's 16-bit type, 16-byte aligned");
  const size_t lds = (size_t)B * 12;
  if (dtype == APLA_F32)
    hipLaunchKernelGGL(koleo_bwd_kernel<float>, dim3(G * B), dim3(256), lds, stream, (const float*)x, B, D, eps, nn_idx, dist, nrm, gout, (float*)dx);
  else
    hipLaunchKernelGGL(koleo_bwd_kernel<bf16>, dim3(G * B), dim3(256), lds, stream, (const bf16*)x, B, D, eps, nn_idx, dist, nrm, gout, (bf16*)dx);
  APLA_CHECK_LAUNCH("apla_koleo_bwd");
  return APLA_OK;
}

// ------------------------------------------------------------------------------------------------ dropout / stochastic depth
// nn.Dropout (vit.py:152-168 Mlp.drop, appla_attn.py:82 proj_drop, pos_drop) and DropPath (vit.py:74-93) for the module path.  All
// shipped configurations use 0; main.py:101-111 can set them.  The keep decision of element i is word (i & 3) of
// Philox4x32-10(counter = {i >> 2 (64 bit), offset (64 bit)}, key = seed (64 bit)) compared with p * 2^32: counter-based, so the mask
// does not depend on the launch geometry and the oracle (oracle/apla_oracle.py:philox_keep_mask) reproduces it bit for bit.
// (philox4x32_10: common.h)

// y = keep ? x / (1 - p) : 0, keep bytes written for the backward.  One thread = 8 consecutive elements (two Philox blocks).
template <typename T>
__global__ __launch_bounds__(256) void dropout_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ keep, long n,
                                                          unsigned threshold, float inv_keep, unsigned long long seed,
                                                          unsigned long long offset, const unsigned long long* __restrict__ rng) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i >= n) return;
  if (rng != nullptr) {   // apla_dropout_fwd_dev: {seed, step} live in device memory; `seed` / `offset` carry stride and site
    const unsigned long long stride = seed, site = offset;
    seed = rng[0];
    offset = rng[1] * stride + site;
  }
  float v[8];
  ld8f<T>(x + i, v);
  unsigned char kb[8];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const unsigned long long blk = (unsigned long long)(i >> 2) + h;
    unsigned c[4] = {(unsigned)blk, (unsigned)(blk >> 32), (unsigned)offset, (unsigned)(offset >> 32)};
    philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool k = c[e] >= threshold;
      kb[4 * h + e] = k ? 1 : 0;
      v[4 * h + e] = k ? v[4 * h + e] * inv_keep : 0.f;
    }
  }
  f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
  st8f<T>(y + i, a, b, false);
  if (keep != nullptr) *(unsigned long long*)(keep + i) = *(const unsigned long long*)kb;
}

// dx = keep ? dy / (1 - p) : 0
template <typename T>
__global__ __launch_bounds__(256) void dropout_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ keep, T* __restrict__ dx,
                                                          long n, float inv_keep) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i >= n) return;
  float v[8];
  ld8f<T>(dy + i, v);
  const unsigned long long kw = *(const unsigned long long*)(keep + i);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = ((kw >> (8 * e)) & 0xff) ? v[e] * inv_keep : 0.f;
  f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
  st8f<T>(dx + i, a, b, false);
}

// y[s, :] = x[s, :] * scale[s] over S samples of `per` elements each (DropPath: scale[s] = floor(keep_prob + u_s) / keep_prob)
template <typename T>
__global__ __launch_bounds__(256) void scale_samples_kernel(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ scale,
                                                            long per, long n) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i >= n) return;
  float v[8];
  ld8f<T>(x + i, v);
  const float sc = scale[i / per];   // per % 8 == 0: the eight elements belong to one sample
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] *= sc;
  f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
  st8f<T>(y + i, a, b, false);
}

extern "C" int apla_dropout_fwd(const void* x, int dtype, void* y, uint8_t* keep, long n, float p, unsigned long long seed,
                                unsigned long long offset, hipStream_t stream) {
  APLA_REQUIRE(x && y && keep && n > 0 && n % 8 == 0 && p >= 0.f && p < 1.f, "apla_dropout_fwd: need n %% 8 == 0 and 0 <= p < 1 (n=%ld p=%f)", n, (double)p);
  APLA_REQUIRE(apla_aligned16(x) && apla_aligned16(y) && (((uintptr_t)keep) & 7) == 0, "apla_dropout_fwd: pointers must be 16-byte (keep: 8-byte) aligned");
  const double t = (double)p * 4294967296.0;
  const unsigned threshold = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
  const unsigned blocks = (unsigned)((n / 8 + 255) / 256);
  if (dtype == APLA_F32) hipLaunchKernelGGL(dropout_fwd_kernel<float>, dim3(blocks), dim3(256), 0, stream, (const float*)x, (float*)y, keep, n, threshold, 1.0f / (1.0f - p), seed, offset, (const unsigned long long*)nullptr);
  else if (dtype == APLA_H16) hipLaunchKernelGGL(dropout_fwd_kernel<bf16>, dim3(blocks), dim3(256), 0, stream, (const bf16*)x, (bf16*)y, keep, n, threshold, 1.0f / (1.0f - p), seed, offset, (const unsigned long long*)nullptr);
  else { apla_set_error("apla_dropout_fwd: unsupported dtype %d", dtype); return APLA_ENOSYS; }
  APLA_CHECK_LAUNCH("apla_dropout_fwd");
  return APLA_OK;
}

// the same pass with {seed, step} read from device memory (offset = step * rng_stride + site) and the keep bytes optional: what a
// captured launch sequence can replay with a new mask every step (include/apla_hip.h)
extern "C" int apla_dropout_fwd_dev(const void* x, int dtype, void* y, uint8_t* keep, long n, float p, const unsigned long long* rng,
                                    unsigned long long rng_stride, unsigned site, hipStream_t stream) {
  APLA_REQUIRE(x && y && rng && n > 0 && n % 8 == 0 && p >= 0.f && p < 1.f, "apla_dropout_fwd_dev: need rng, n %% 8 == 0 and 0 <= p < 1 (n=%ld p=%f)", n, (double)p);
  APLA_REQUIRE(apla_aligned16(x) && apla_aligned16(y) && (((uintptr_t)keep) & 7) == 0, "apla_dropout_fwd_dev: pointers must be 16-byte (keep: 8-byte) aligned");
  const unsigned threshold = apla_drop_threshold(p);
  const unsigned blocks = (unsigned)((n / 8 + 255) / 256);
  if (dtype == APLA_F32) hipLaunchKernelGGL(dropout_fwd_kernel<float>, dim3(blocks), dim3(256), 0, stream, (const float*)x, (float*)y, keep, n, threshold, 1.0f / (1.0f - p), rng_stride, (unsigned long long)site, rng);
  else if (dtype == APLA_H16) hipLaunchKernelGGL(dropout_fwd_kernel<bf16>, dim3(blocks), dim3(256), 0, stream, (const bf16*)x, (bf16*)y, keep, n, threshold, 1.0f / (1.0f - p), rng_stride, (unsigned long long)site, rng);
  else { apla_set_error("apla_dropout_fwd_dev: unsupported dtype %d", dtype); return APLA_ENOSYS; }
  APLA_CHECK_LAUNCH("apla_dropout_fwd_dev");
  return APLA_OK;
}

extern "C" int apla_dropout_bwd(const void* dy, int dtype, const uint8_t* keep, void* dx, long n, float p, hipStream_t stream) {
  APLA_REQUIRE(dy && dx && keep && n > 0 && n % 8 == 0 && p >= 0.f && p < 1.f, "apla_dropout_bwd: need n %% 8 == 0 and 0 <= p < 1");
  APLA_REQUIRE(apla_aligned16(dy) && apla_aligned16(dx) && (((uintptr_t)keep) & 7) == 0, "apla_dropout_bwd: pointers must be 16-byte (keep: 8-byte) aligned");
  const unsigned blocks = (unsigned)((n / 8 + 255) / 256);
  if (dtype == APLA_F32) hipLaunchKernelGGL(dropout_bwd_kernel<float>, dim3(blocks), dim3(256), 0, stream, (const float*)dy, keep, (float*)dx, n, 1.0f / (1.0f - p));
  else if (dtype == APLA_H16) hipLaunchKernelGGL(dropout_bwd_kernel<bf16>, dim3(blocks), dim3(256), 0, stream, (const bf16*)dy, keep, (bf16*)dx, n, 1.0f / (1.0f - p));
  else { apla_set_error("apla_dropout_bwd: unsupported dtype %d", dtype); return APLA_ENOSYS; }
  APLA_CHECK_LAUNCH("apla_dropout_bwd");
  return APLA_OK;
}

extern "C" int apla_scale_samples(const void* x, int dtype, void* y, const float* scale, long samples, long per_sample, hipStream_t stream) {
  APLA_REQUIRE(x && y && scale && samples > 0 && per_sample > 0 && per_sample % 8 == 0, "apla_scale_samples: need per_sample %% 8 == 0");
  APLA_REQUIRE(apla_aligned16(x) && apla_aligned16(y), "apla_scale_samples: pointers must be 16-byte aligned");
  const long n = samples * per_sample;
  const unsigned blocks = (unsigned)((n / 8 + 255) / 256);
  if (dtype == APLA_F32) hipLaunchKernelGGL(scale_samples_kernel<float>, dim3(blocks), dim3(256), 0, stream, (const float*)x, (float*)y, scale, per_sample, n);
  else if (dtype == APLA_H16) hipLaunchKernelGGL(scale_samples_kernel<bf16>, dim3(blocks), dim3(256), 0, stream, (const bf16*)x, (bf16*)y, scale, per_sample, n);
  else { apla_set_error("apla_scale_samples: unsupported dtype %d", dtype); return APLA_ENOSYS; }
  APLA_CHECK_LAUNCH("apla_scale_samples");
  return APLA_OK;
}

// ------------------------------------------------------------------------------------------------ K-panel operand image
// dst[(k / 32) * rows + r][k % 32] = src[r][k]: the image apla_gemm_nt_ex reads with flags bit 16 (W) / bit 17 (A).  One thread
// moves 16 bytes; a frozen weight is converted once, a trainable one by its own pack kernel.
__global__ __launch_bounds__(256) void pack_k_panels_kernel(const bf16* __restrict__ src, long ld, bf16* __restrict__ dst, int rows, int K) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int cpr = K / 8;  // 16-byte chunks per row
  if (t >= (long)rows * cpr) return;
  const int r = (int)(t / cpr), c = (int)(t - (long)r * cpr);
  const bf16x8 v = *(const bf16x8*)(src + (long)r * ld + c * 8);
  *(bf16x8*)(dst + ((long)(c >> 2) * rows + r) * 32 + (c & 3) * 8) = v;
}

extern "C" int apla_pack_k_panels(const void* src, long ld, void* dst, int rows, int K, hipStream_t stream) {
  APLA_REQUIRE(src && dst && rows > 0 && K > 0 && K % 32 == 0 && ld % 8 == 0 && ld >= K && apla_aligned16(src) && apla_aligned16(dst),
               "apla_pack_k_panels: [rows, K] 16-bit matrix with K %% 32 == 0, 16-byte aligned rows");
  const long n = (long)rows * (K / 8);
  hipLaunchKernelGGL(pack_k_panels_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const bf16*)src, ld, (bf16*)dst, rows, K);
  APLA_CHECK_LAUNCH("apla_pack_k_panels");
  return APLA_OK;
}

// ------------------------------------------------------------------------------------------------ CU-occupancy probe (diagnostic)
// `workgroups` workgroups of `threads` threads holding `lds_bytes` of LDS each spin for `usec` microseconds (constant 100 MHz
// clock; bounded).  Launched on a side stream next to the fused step it stands in for a collective's kernels: the persistent GEMM /
// attention kernels of the step launch one (or two) workgroups per CU and want all 256 CUs at once, so what a resident foreign
// kernel costs them is measured, not assumed (tools/contention_probe.py -> profiles/r03_contention.md).  Not on the product path.
__global__ void occupy_kernel(long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned n = 0;   // every wave leaves after `ticks` of the 100 MHz clock, or after 2^24 polls at the latest
  while ((long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks && n < (1u << 24)) { ++n; __builtin_amdgcn_s_sleep(8); }
}
extern "C" int apla_probe_occupy(int workgroups, int threads, int lds_bytes, int usec, hipStream_t stream) {
  APLA_REQUIRE(workgroups > 0 && workgroups <= 1024 && threads >= 64 && threads <= 1024 && threads % 64 == 0 && lds_bytes >= 0 &&
               lds_bytes <= 64 * 1024 && usec > 0 && usec <= 50000, "apla_probe_occupy: bad arguments");
  hipLaunchKernelGGL(occupy_kernel, dim3(workgroups), dim3(threads), (size_t)lds_bytes, stream, (long)usec * 100);
  APLA_CHECK_LAUNCH("apla_probe_occupy");
  return APLA_OK;
}
