// Few-row GEMM (M <= a few hundred rows) with the epilogues of apla_gemm_nt: the CLS-only tail of the last block
// (engine.py: projection / MLP / Q on [B, D] rows instead of [B*N, D]).
//
// At M = 128 the tiled kernels are latency-bound: N/128 = 6..24 workgroups walk the whole K axis alone (17 us at K = 768,
// 45 us at K = 3072 for a product whose operands are 5 MB).  Here the K axis is split over S slices so that a few hundred
// workgroups run: workgroup (n-tile of 64 columns, slice s, row group of 128) multiplies A[rows, slice] by W[cols, slice]^T
// straight from global memory — operand fragments of v_mfma_f32_16x16x32 are 16-byte row pieces, no LDS, no barrier — and
// writes an fp32 partial tile; a second launch sums the S partials in a fixed order (deterministic), adds the bias and applies
// the epilogue.  The partials live in a caller-provided workspace (apla_gemm_small_workspace_bytes).
#include "gemm_common.h"

namespace {

constexpr int SBN = 64, SBM = 128;

__host__ __device__ inline int small_kslice(int K) { return K <= 1536 ? 64 : 128; }

__global__ __launch_bounds__(256) void gemm_small_partial_kernel(const bf16* __restrict__ A, int lda, const bf16* __restrict__ W,
                                                                 int ldw, float* __restrict__ partial, int M, int N, int K,
                                                                 int kslice) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * SBN, s = blockIdx.y, m0 = blockIdx.z * SBM + wave * 32;
  if (m0 >= M) return;
  const int fr = lane & 15, kq = (lane >> 4) * 8;
  const int k0 = s * kslice;
  const bf16* ap[2];
  const bf16* wp[4];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int r = m0 + 16 * i + fr;
    r = r < M ? r : M - 1;   // rows past the end are computed from a clamped row and never stored
    ap[i] = A + (size_t)r * lda + k0 + kq;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) wp[j] = W + (size_t)(n0 + 16 * j + fr) * ldw + k0 + kq;
  f32x4 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int k = 0; k < kslice; k += 32) {
    bf16x8 af[2], wf[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) af[i] = *(const bf16x8*)(ap[i] + k);
#pragma unroll
    for (int j = 0; j < 4; ++j) wf[j] = *(const bf16x8*)(wp[j] + k);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = MFMA_F32_16x16x32_H16(af[i], wf[j], acc[i][j]);
  }
  // D[m][n]: lane holds column n = lane & 15, rows m = 4 * (lane >> 4) + reg
  float* P = partial + (size_t)s * M * N;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int m = m0 + 16 * i + 4 * (lane >> 4) + reg;
      if (m < M) {
#pragma unroll
        for (int j = 0; j < 4; ++j) P[(size_t)m * N + n0 + 16 * j + fr] = acc[i][j][reg];
      }
    }
}

template <int EPI, typename OutT>
__global__ __launch_bounds__(256) void gemm_small_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ bias,
                                                                OutT* __restrict__ C, int ldc, const void* __restrict__ aux_in,
                                                                int ld_in, bf16* __restrict__ aux_out, int ld_out, int M, int N,
                                                                int S) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int n4 = N / 4;
  if (t >= (long)M * n4) return;
  const int m = (int)(t / n4), n = (int)(t - (long)m * n4) * 4;
  const size_t stride = (size_t)M * N;
  const float* src = partial + (size_t)m * N + n;
  f32x4 v = *(const f32x4*)src;
  for (int s = 1; s < S; ++s) v += *(const f32x4*)(src + s * stride);   // fixed order: bitwise reproducible
  if (bias != nullptr) v += *(const f32x4*)(bias + n);
  if constexpr (EPI == APLA_EPI_STORE) {
    Vec4IO<OutT>::store(C + (size_t)m * ldc + n, v);
  } else if constexpr (EPI == APLA_EPI_RESIDUAL) {
    v += Vec4IO<OutT>::load((const OutT*)aux_in + (size_t)m * ld_in + n);
    Vec4IO<OutT>::store(C + (size_t)m * ldc + n, v);
  } else if constexpr (EPI == APLA_EPI_MUL) {
    v *= Vec4IO<bf16>::load((const bf16*)aux_in + (size_t)m * ld_in + n);
    Vec4IO<OutT>::store(C + (size_t)m * ldc + n, v);
  } else if constexpr (EPI == APLA_EPI_GELU_FWD) {
    f32x4 h;
    h = gelu_only4(v);
    Vec4IO<OutT>::store(C + (size_t)m * ldc + n, h);
  } else {  // GELU: h and gelu'
    f32x4 h, g;
    gelu_and_grad4(v, h, g);
    Vec4IO<OutT>::store(C + (size_t)m * ldc + n, h);
    Vec4IO<bf16>::store(aux_out + (size_t)m * ld_out + n, g);
  }
}

}  // namespace

extern "C" long apla_gemm_small_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0 || N % SBN != 0 || K % 128 != 0) return -1;
  return (long)(K / small_kslice(K)) * M * N * (long)sizeof(float);
}

extern "C" int apla_gemm_nt_small(const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc, int M,
                                  int N, int K, int epilogue, int out_dtype, const void* aux_in, int ld_aux_in,
                                  void* aux_out, int ld_aux_out, void* workspace, long workspace_bytes, hipStream_t stream) {
  APLA_REQUIRE(A && W && C && workspace && M > 0 && M <= 4096, "apla_gemm_nt_small: bad arguments (M=%d)", M);
  APLA_REQUIRE(N % SBN == 0 && K % 128 == 0, "apla_gemm_nt_small: need N%%64==0 and K%%128==0 (N=%d K=%d)", N, K);
  APLA_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && lda >= K && ldw >= K && ldc % 4 == 0 && ldc >= N, "apla_gemm_nt_small: bad leading dimensions");
  APLA_REQUIRE(apla_aligned16(A) && apla_aligned16(W) && apla_aligned16(C) && apla_aligned16(workspace) && (bias == nullptr || apla_aligned16(bias)),
               "apla_gemm_nt_small: pointers must be 16-byte aligned");
  APLA_REQUIRE(workspace_bytes >= apla_gemm_small_workspace_bytes(M, N, K), "apla_gemm_nt_small: workspace too small");
  const int ks = small_kslice(K), S = K / ks;
  hipLaunchKernelGGL(gemm_small_partial_kernel, dim3(N / SBN, S, (M + SBM - 1) / SBM), dim3(256), 0, stream, (const bf16*)A, lda,
                     (const bf16*)W, ldw, (float*)workspace, M, N, K, ks);
  APLA_CHECK_LAUNCH("apla_gemm_nt_small[partial]");
  const unsigned blocks = (unsigned)(((long)M * (N / 4) + 255) / 256);
#define SMALL_REDUCE(E, T)                                                                                                   \
  hipLaunchKernelGGL((gemm_small_reduce_kernel<E, T>), dim3(blocks), dim3(256), 0, stream, (const float*)workspace, bias,    \
                     (T*)C, ldc, aux_in, ld_aux_in, (bf16*)aux_out, ld_aux_out, M, N, S)
  const bool f32 = out_dtype == APLA_F32;
  if (!f32 && out_dtype != APLA_H16) { apla_set_error("apla_gemm_nt_small: unsupported out_dtype %d", out_dtype); return APLA_ENOSYS; }
  switch (epilogue) {
    case APLA_EPI_STORE:
      if (f32) SMALL_REDUCE(APLA_EPI_STORE, float); else SMALL_REDUCE(APLA_EPI_STORE, bf16);
      break;
    case APLA_EPI_RESIDUAL:
      APLA_REQUIRE(aux_in && apla_aligned16(aux_in) && ld_aux_in % 4 == 0 && ld_aux_in >= N, "apla_gemm_nt_small[residual]: aux_in [M,N] required");
      if (f32) SMALL_REDUCE(APLA_EPI_RESIDUAL, float); else SMALL_REDUCE(APLA_EPI_RESIDUAL, bf16);
      break;
    case APLA_EPI_MUL:
      APLA_REQUIRE(!f32 && aux_in && apla_aligned16(aux_in) && ld_aux_in % 4 == 0 && ld_aux_in >= N, "apla_gemm_nt_small[mul]: aux_in [M,N] bf16 required, 16-bit output");
      SMALL_REDUCE(APLA_EPI_MUL, bf16);
      break;
    case APLA_EPI_GELU_FWD:
      APLA_REQUIRE(!f32, "apla_gemm_nt_small[gelu_fwd]: 16-bit output");
      SMALL_REDUCE(APLA_EPI_GELU_FWD, bf16);
      break;
    case APLA_EPI_GELU:
      APLA_REQUIRE(!f32 && aux_out && apla_aligned16(aux_out) && ld_aux_out % 4 == 0 && ld_aux_out >= N, "apla_gemm_nt_small[gelu]: aux_out [M,N] bf16 required, 16-bit output");
      SMALL_REDUCE(APLA_EPI_GELU, bf16);
      break;
    default:
      apla_set_error("apla_gemm_nt_small: unsupported epilogue %d", epilogue);
      return APLA_ENOSYS;
  }
#undef SMALL_REDUCE
  APLA_CHECK_LAUNCH("apla_gemm_nt_small[reduce]");
  return APLA_OK;
}
