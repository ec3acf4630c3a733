// bf16 MFMA GEMM  C[M,N] = A[M,K] · W[N,K]^T  with fused epilogues, for gfx950.
//
// Every dense contraction on the APLA step is expressed in this "NT" form (both operands K-contiguous): forward
// linears use the frozen weight as stored by nn.Linear ([out,in]); the dX backward uses a transposed bf16 copy of the
// same frozen weight that the engine prepares once (288 GB of HBM makes the duplicate free), so no "NN" kernel exists.
//
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 tiles of v_mfma_f32_16x16x32_bf16.
//  * staging: __builtin_amdgcn_global_load_lds, 16 B per lane, straight into a double-buffered LDS image
//    (2 x (16 KB A + 16 KB W)).  The LDS destination of an LDS-DMA is lane-linear, so the bank-conflict swizzle
//    (16-byte chunk index XOR ((row>>1)&7)) is applied to the per-lane SOURCE address and again on the ds_read.
//  * the MFMA is issued with the WEIGHT fragment as the A operand and the ACTIVATION fragment as the B operand, i.e.
//    it computes C^T tiles: a lane then owns 4 consecutive output columns of one output row, which makes the epilogue
//    (bias, GELU, residual, multiplier) a vector op on 8/16-byte global accesses.
//  * blockIdx -> tile mapping is XCD-aware: each XCD walks a contiguous run of tiles with n fastest, so an A row
//    panel is fetched into one L2 and reused by all N/128 column tiles.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;  // 32 KB

struct GemmParams {
  const bf16* A; int lda;
  const bf16* W; int ldw;
  const float* bias;
  void* C; int ldc;
  const void* aux_in; int ld_aux_in;
  void* aux_out; int ld_aux_out;
  int M, N, K, tiles_n;
};

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ float gelu_f(float a) { return 0.5f * a * (1.0f + erff(a * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float a) {
  return 0.5f * (1.0f + erff(a * 0.70710678118654752f)) + a * __expf(-0.5f * a * a) * 0.3989422804014327f;
}
__device__ __forceinline__ float sigmoid_f(float a) { return 1.0f / (1.0f + __expf(-a)); }

template <int EPI, typename OutT>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = wg / p.tiles_n, tn = wg - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  // per-lane source rows / chunks for the 4 A pieces and 4 W pieces this wave stages per K-step
  const bf16* a_src[4];
  const bf16* w_src[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int piece = wave * 4 + it;
    const int row = piece * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    int gr = m0 + row;
    gr = gr < p.M ? gr : p.M - 1;  // tail rows: re-read the last valid row, never stored
    a_src[it] = p.A + (size_t)gr * p.lda + chunk * 8;
    w_src[it] = p.W + (size_t)(n0 + row) * p.ldw + chunk * 8;
  }

  auto stage = [&](int s, int k0) {
    char* base = smem + s * STAGE_BYTES;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int piece = wave * 4 + it;
      __builtin_amdgcn_global_load_lds(GLBP(a_src[it] + k0), LDSP(base + piece * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(GLBP(w_src[it] + k0), LDSP(base + BM * BK * 2 + piece * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fq = lane >> 4;
  const int nk = p.K / BK;
  stage(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, (kt + 1) * BK);
    const char* As = smem + cur * STAGE_BYTES;
    const char* Ws = As + BM * BK * 2;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], wf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *(const bf16x8*)(As + lds_off(wm * 64 + i * 16 + frow, ks * 4 + fq));
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[j] = *(const bf16x8*)(Ws + lds_off(wn * 64 + j * 16 + frow, ks * 4 + fq));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- epilogue: lane owns row m = .. + (lane&15), columns n = .. + 4*(lane>>4) + {0..3} of each 16x16 tile ----
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + frow;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + fq * 4;
      f32x4 v = acc[i][j];
      if (p.bias != nullptr) v += *(const f32x4*)(p.bias + n);
      if constexpr (EPI == APLA_EPI_STORE) {
        Vec4IO<OutT>::store((OutT*)p.C + (size_t)m * p.ldc + n, v);
      } else if constexpr (EPI == APLA_EPI_GELU) {
        f32x4 h, g;
#pragma unroll
        for (int e = 0; e < 4; ++e) { h[e] = gelu_f(v[e]); g[e] = gelu_grad_f(v[e]); }
        Vec4IO<bf16>::store((bf16*)p.C + (size_t)m * p.ldc + n, h);
        Vec4IO<bf16>::store((bf16*)p.aux_out + (size_t)m * p.ld_aux_out + n, g);
      } else if constexpr (EPI == APLA_EPI_RESIDUAL) {
        f32x4 r = Vec4IO<OutT>::load((const OutT*)p.aux_in + (size_t)m * p.ld_aux_in + n);
        Vec4IO<OutT>::store((OutT*)p.C + (size_t)m * p.ldc + n, r + v);
      } else if constexpr (EPI == APLA_EPI_MUL) {
        f32x4 g = Vec4IO<bf16>::load((const bf16*)p.aux_in + (size_t)m * p.ld_aux_in + n);
        Vec4IO<bf16>::store((bf16*)p.C + (size_t)m * p.ldc + n, v * g);
      } else if constexpr (EPI == APLA_EPI_SWIGLU) {
        // columns (n, n+1) = (x1_i, x2_i), (n+2, n+3) = (x1_{i+1}, x2_{i+1}); i = n/2
        Vec4IO<bf16>::store((bf16*)p.aux_out + (size_t)m * p.ld_aux_out + n, v);
        bf16x2 h;
        h[0] = (bf16)(v[0] * sigmoid_f(v[0]) * v[1]);
        h[1] = (bf16)(v[2] * sigmoid_f(v[2]) * v[3]);
        *(bf16x2*)((bf16*)p.C + (size_t)m * p.ldc + (n >> 1)) = h;
      } else if constexpr (EPI == APLA_EPI_SWIGLU_BWD) {
        // v = dh for hidden units n..n+3; saved x12 interleaved at columns 2n..2n+7
        const bf16* xs = (const bf16*)p.aux_in + (size_t)m * p.ld_aux_in + 2 * n;
        bf16x8 x12 = *(const bf16x8*)xs;
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x1 = (float)x12[2 * e], x2 = (float)x12[2 * e + 1];
          const float s = sigmoid_f(x1);
          o[2 * e] = (bf16)(v[e] * x2 * s * (1.0f + x1 * (1.0f - s)));
          o[2 * e + 1] = (bf16)(v[e] * x1 * s);
        }
        *(bf16x8*)((bf16*)p.C + (size_t)m * p.ldc + 2 * n) = o;
      }
    }
  }
}

template <int EPI, typename OutT>
int launch(const GemmParams& p, hipStream_t stream) {
  const int tiles_m = (p.M + BM - 1) / BM;
  dim3 grid(tiles_m * p.tiles_n), block(256);
  hipLaunchKernelGGL((gemm_nt_kernel<EPI, OutT>), grid, block, 0, stream, p);
  APLA_CHECK_LAUNCH("apla_gemm_nt");
  return APLA_OK;
}

}  // namespace

extern "C" int apla_gemm_nt(const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc, int M,
                            int N, int K, int epilogue, int out_dtype, const void* aux_in, int ld_aux_in,
                            void* aux_out, int ld_aux_out, hipStream_t stream) {
  APLA_REQUIRE(M > 0 && N > 0 && K > 0, "apla_gemm_nt: empty problem M=%d N=%d K=%d", M, N, K);
  APLA_REQUIRE(N % BN == 0 && K % BK == 0, "apla_gemm_nt: need N%%128==0 and K%%64==0 (N=%d K=%d)", N, K);
  APLA_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && lda >= K && ldw >= K, "apla_gemm_nt: bad lda/ldw (%d,%d) K=%d", lda, ldw, K);
  APLA_REQUIRE(apla_aligned16(A) && apla_aligned16(W) && apla_aligned16(C) && A && W && C, "apla_gemm_nt: pointers must be 16-byte aligned");
  APLA_REQUIRE(bias == nullptr || apla_aligned16(bias), "apla_gemm_nt: bias must be 16-byte aligned");
  APLA_REQUIRE(ldc % 4 == 0, "apla_gemm_nt: ldc %% 4 != 0");
  GemmParams p{(const bf16*)A, lda, (const bf16*)W, ldw, bias, C, ldc, aux_in, ld_aux_in, aux_out, ld_aux_out, M, N, K, N / BN};
  switch (epilogue) {
    case APLA_EPI_STORE:
      APLA_REQUIRE(ldc >= N, "apla_gemm_nt: ldc < N");
      if (out_dtype == APLA_BF16) return launch<APLA_EPI_STORE, bf16>(p, stream);
      if (out_dtype == APLA_F32) return launch<APLA_EPI_STORE, float>(p, stream);
      break;
    case APLA_EPI_GELU:
      APLA_REQUIRE(aux_out && apla_aligned16(aux_out) && ld_aux_out % 4 == 0 && ld_aux_out >= N && ldc >= N, "apla_gemm_nt[gelu]: aux_out [M,N] bf16 required");
      return launch<APLA_EPI_GELU, bf16>(p, stream);
    case APLA_EPI_RESIDUAL:
      APLA_REQUIRE(aux_in && apla_aligned16(aux_in) && ld_aux_in % 4 == 0 && ld_aux_in >= N && ldc >= N, "apla_gemm_nt[residual]: aux_in [M,N] required");
      if (out_dtype == APLA_BF16) return launch<APLA_EPI_RESIDUAL, bf16>(p, stream);
      if (out_dtype == APLA_F32) return launch<APLA_EPI_RESIDUAL, float>(p, stream);
      break;
    case APLA_EPI_MUL:
      APLA_REQUIRE(aux_in && apla_aligned16(aux_in) && ld_aux_in % 4 == 0 && ld_aux_in >= N && ldc >= N, "apla_gemm_nt[mul]: aux_in [M,N] bf16 required");
      return launch<APLA_EPI_MUL, bf16>(p, stream);
    case APLA_EPI_SWIGLU:
      APLA_REQUIRE(aux_out && apla_aligned16(aux_out) && ld_aux_out % 4 == 0 && ld_aux_out >= N && ldc % 2 == 0 && ldc >= N / 2, "apla_gemm_nt[swiglu]: aux_out [M,N] bf16 required");
      return launch<APLA_EPI_SWIGLU, bf16>(p, stream);
    case APLA_EPI_SWIGLU_BWD:
      APLA_REQUIRE(aux_in && apla_aligned16(aux_in) && ld_aux_in % 8 == 0 && ld_aux_in >= 2 * N && ldc % 8 == 0 && ldc >= 2 * N, "apla_gemm_nt[swiglu_bwd]: aux_in [M,2N] bf16 required");
      return launch<APLA_EPI_SWIGLU_BWD, bf16>(p, stream);
    default:
      break;
  }
  apla_set_error("apla_gemm_nt: unsupported epilogue %d / out_dtype %d", epilogue, out_dtype);
  return APLA_ENOSYS;
}
