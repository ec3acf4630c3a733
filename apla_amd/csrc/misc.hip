// Front end (patchify / token assembly) and classifier-head kernels.  All are forward-only or tiny: the patch embedding
// never sees a gradient under APLA (first trainable leaf is block 0's projection) and the head is [B,D]x[D,C] in fp32.
#include "common.h"

namespace {

// images fp32 [B,3,S,S] -> cols bf16 [B*Np, Kp]; column order (c, py, px) == conv weight.reshape(D, 3*p*p)
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, bf16* __restrict__ cols, int S,
                                                       int patch, int grid_w, int Np, int Kp) {
  const int row = blockIdx.x;  // b*Np + t
  const int b = row / Np, t = row - b * Np;
  const int gy = t / grid_w, gx = t - gy * grid_w;
  const int pp = patch * patch, K = 3 * pp;
  for (int k = threadIdx.x; k < Kp; k += 256) {
    float v = 0.f;
    if (k < K) {
      const int c = k / pp, rem = k - c * pp, py = rem / patch, px = rem - py * patch;
      v = img[(((size_t)b * 3 + c) * S + gy * patch + py) * S + gx * patch + px];
    }
    cols[(size_t)row * Kp + k] = (bf16)v;
  }
}

template <typename ResT>
__global__ __launch_bounds__(256) void assemble_tokens_kernel(const bf16* __restrict__ patches, int ldp,
                                                              const float* __restrict__ cls,
                                                              const float* __restrict__ pos, ResT* __restrict__ tok,
                                                              int Np, int D) {
  const int row = blockIdx.x;  // b*(Np+1) + n
  const int N = Np + 1;
  const int b = row / N, n = row - b * N;
  for (int c4 = threadIdx.x; c4 < D / 4; c4 += 256) {
    f32x4 v = *(const f32x4*)(pos + (size_t)n * D + c4 * 4);
    if (n == 0) v += *(const f32x4*)(cls + c4 * 4);
    else v += Vec4IO<bf16>::load(patches + ((size_t)b * Np + n - 1) * ldp + c4 * 4);
    Vec4IO<ResT>::store(tok + (size_t)row * D + c4 * 4, v);
  }
}

// C[i,j] (+)= sum_k A[i*sai + k*sak] * B[k*sbk + j*sbj] (+ bias[j]); fp32 FMA, 32x32 tile per workgroup, 128-deep K chunks.
// The head GEMMs are tiny (0.2 GFLOP) and latency-bound: the next chunk is fetched into registers while the current one
// is multiplied out of LDS, and the element -> (row, k) maps of the staging loads are fixed per thread (no index
// arithmetic inside the loop).
constexpr int SG_BK = 128;
__global__ __launch_bounds__(256) void sgemm_small_kernel(const float* __restrict__ A, long sai, long sak,
                                                          const float* __restrict__ Bm, long sbk, long sbj,
                                                          const float* __restrict__ bias, float* __restrict__ C,
                                                          long ldc, int M, int N, int K, int accumulate) {
  __shared__ float As[32][SG_BK + 1], Bs[SG_BK][33];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;  // 16 x 16 threads, 2x2 outputs each
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  // staging maps: element e = tid + 256*u of a 32 x 128 tile; the faster-varying global index follows the thread index
  const bool a_kfast = sak <= sai, b_jfast = sbj <= sbk;
  int ai[16], ak[16], bk[16], bj[16];
  long aoff[16], boff[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int e = threadIdx.x + 256 * u;
    ai[u] = a_kfast ? e >> 7 : e & 31;
    ak[u] = a_kfast ? e & 127 : e >> 5;
    bk[u] = b_jfast ? e >> 5 : e & 127;
    bj[u] = b_jfast ? e & 31 : e >> 7;
    aoff[u] = (long)(i0 + ai[u]) * sai + (long)ak[u] * sak;
    boff[u] = (long)bk[u] * sbk + (long)(j0 + bj[u]) * sbj;
  }
  float ar[16], br[16];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      ar[u] = (i0 + ai[u] < M && k0 + ak[u] < K) ? A[aoff[u] + (long)k0 * sak] : 0.f;
      br[u] = (j0 + bj[u] < N && k0 + bk[u] < K) ? Bm[boff[u] + (long)k0 * sbk] : 0.f;
    }
  };
  float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  fetch(0);
  for (int k0 = 0; k0 < K; k0 += SG_BK) {
    __syncthreads();  // previous chunk fully consumed
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      As[ai[u]][ak[u]] = ar[u];
      Bs[bk[u]][bj[u]] = br[u];
    }
    __syncthreads();
    if (k0 + SG_BK < K) fetch(k0 + SG_BK);
#pragma unroll 16
    for (int k = 0; k < SG_BK; ++k) {
      const float a0 = As[ty][k], a1 = As[ty + 16][k], b0 = Bs[k][tx], b1 = Bs[k][tx + 16];
      acc[0][0] += a0 * b0; acc[0][1] += a0 * b1; acc[1][0] += a1 * b0; acc[1][1] += a1 * b1;
    }
  }
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      const int i = i0 + ty + 16 * u, j = j0 + tx + 16 * w;
      if (i < M && j < N) {
        float v = acc[u][w];
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

__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, long ld, float* __restrict__ out,
                                                     int M, int N) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= N) return;
  float s = 0.f;
  for (int i = 0; i < M; ++i) s += X[(long)i * ld + j];
  out[j] = s;
}

}  // namespace

extern "C" int apla_patchify(const float* images, void* cols, int B, int S, int patch, int Kp, hipStream_t stream) {
  APLA_REQUIRE(images && cols && B > 0 && S > 0 && patch > 0 && S >= patch, "apla_patchify: bad arguments");
  APLA_REQUIRE(Kp >= 3 * patch * patch && Kp % 64 == 0, "apla_patchify: Kp must be >= 3*p*p and a multiple of 64");
  const int gw = S / patch, Np = gw * gw;
  hipLaunchKernelGGL(patchify_kernel, dim3(B * Np), dim3(256), 0, stream, images, (bf16*)cols, S, patch, gw, Np, Kp);
  APLA_CHECK_LAUNCH("apla_patchify");
  return APLA_OK;
}

extern "C" int apla_assemble_tokens(const void* patches, int ldp, const float* cls_token, const float* pos_embed,
                                    void* tokens, int res_dtype, int B, int Np, int D, hipStream_t stream) {
  APLA_REQUIRE(patches && cls_token && pos_embed && tokens && B > 0 && Np > 0 && D % 4 == 0 && ldp % 4 == 0, "apla_assemble_tokens: bad arguments");
  if (res_dtype == APLA_F32)
    hipLaunchKernelGGL(assemble_tokens_kernel<float>, dim3(B * (Np + 1)), dim3(256), 0, stream, (const bf16*)patches, ldp, cls_token, pos_embed, (float*)tokens, Np, D);
  else if (res_dtype == APLA_H16)
    hipLaunchKernelGGL(assemble_tokens_kernel<bf16>, dim3(B * (Np + 1)), dim3(256), 0, stream, (const bf16*)patches, ldp, cls_token, pos_embed, (bf16*)tokens, Np, D);
  else {
    apla_set_error("apla_assemble_tokens: bad res_dtype %d", res_dtype);
    return APLA_ENOSYS;
  }
  APLA_CHECK_LAUNCH("apla_assemble_tokens");
  return APLA_OK;
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
                                                         float* __restrict__ row_loss, int K) {
  __shared__ float red[4];
  __shared__ float bc;
  const int r = blockIdx.x;
  const ST* sr = s + (long)r * lds;
  const float* tr = t + (long)r * ldt;
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

extern "C" int apla_softmax_center(const void* x, int x_dtype, long ldx, const float* center, float inv_temp, float* out,
                                   long ldo, int R, int K, hipStream_t stream) {
  APLA_REQUIRE(x && center && out && R > 0 && K > 0 && ldx >= K && ldo >= K && inv_temp > 0.f, "apla_softmax_center: bad arguments");
  if (x_dtype == APLA_F32) hipLaunchKernelGGL(softmax_center_kernel<float>, dim3(R), dim3(256), 0, stream, (const float*)x, ldx, center, inv_temp, out, ldo, K);
  else if (x_dtype == APLA_H16) hipLaunchKernelGGL(softmax_center_kernel<bf16>, dim3(R), dim3(256), 0, stream, (const bf16*)x, ldx, center, inv_temp, out, ldo, K);
  else { apla_set_error("apla_softmax_center: unsupported dtype %d", x_dtype); return APLA_ENOSYS; }
  APLA_CHECK_LAUNCH("apla_softmax_center");
  return APLA_OK;
}

extern "C" int apla_distill_ce(const void* student, int s_dtype, long lds, const float* teacher_probs, long ldt, float inv_temp,
                               const float* row_weight, float weight, float* dstudent, long ldds, int accumulate,
                               float* row_loss, int R, int K, hipStream_t stream) {
  APLA_REQUIRE(student && teacher_probs && row_loss && R > 0 && K > 0 && lds >= K && ldt >= K && inv_temp > 0.f &&
               (dstudent == nullptr || ldds >= K), "apla_distill_ce: bad arguments");
  if (s_dtype == APLA_F32) hipLaunchKernelGGL(distill_ce_kernel<float>, dim3(R), dim3(256), 0, stream, (const float*)student, lds, teacher_probs, ldt, inv_temp, row_weight, weight, dstudent, ldds, accumulate, row_loss, K);
  else if (s_dtype == APLA_H16) hipLaunchKernelGGL(distill_ce_kernel<bf16>, dim3(R), dim3(256), 0, stream, (const bf16*)student, lds, teacher_probs, ldt, inv_temp, row_weight, weight, dstudent, ldds, accumulate, row_loss, K);
  else { apla_set_error("apla_distill_ce: unsupported dtype %d", s_dtype); return APLA_ENOSYS; }
  APLA_CHECK_LAUNCH("apla_distill_ce");
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
  hipLaunchKernelGGL(sgemm_small_kernel, dim3((N + 31) / 32, (M + 31) / 32), dim3(256), 0, stream, A, sai, sak, Bm, sbk, sbj, bias, C, ldc, M, N, K, accumulate);
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
  hipLaunchKernelGGL(colsum_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, X, ld, out, M, N);
  APLA_CHECK_LAUNCH("apla_colsum");
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
