// APLA-specific kernels: the column-masked projection weight gradient and the weight-side scatter.
//
// (1) apla_proj_dw:  dW1[j,:] = row_scale[j] * sum_m dyg[m,j] * x[m,:]   (TN contraction over the token axis)
//     Only the r trainable output features exist in dyg, so the frozen (D-r) rows of dW are never formed.
//     Both operands have the reduction index (m) as their slow axis; they are staged row-major into LDS and read
//     through ds_read_b64_tr_b16 as MFMA (32x32x16) fragments.  The token axis is split over S workgroup slabs that
//     write fp32 partial tiles; a second kernel sums the slabs in a fixed order (deterministic, no atomics), applies
//     row_scale (the LayerScale gamma of the trainable rows) and optionally accumulates into dW1/db1.
// (2) apla_pack_proj_rows: scatter the r trainable rows (scaled by gamma) into the natural-order bf16 weight, its
//     transposed copy and the natural-order bias.  This replaces the two activation-side scatter_ calls of the
//     reference forward (appla_attn.py:70-79) by a weight-side scatter of r*D elements per step.
#include "common.h"

namespace {

__device__ __forceinline__ int tile_off(int row, int chunk) {
  const int f = (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
  return row * 128 + ((chunk ^ f) << 4);
}

__device__ __forceinline__ bf16x8 tr_frag(const char* lds, int rbase, int c0, int lane) {
  const int g = lane >> 4, i = lane & 15;
  const int col = c0 + 16 * (g & 1) + 4 * (i & 3);
  const int r = rbase + 4 * (g >> 1) + (i >> 2);
  const int a0 = tile_off(r, col >> 3) + ((col & 4) << 1);
  const int a1 = tile_off(r + 8, col >> 3) + ((col & 4) << 1);
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(lds + a0));
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(lds + a1));
  bf16x8 out;
  out[0] = lo[0]; out[1] = lo[1]; out[2] = lo[2]; out[3] = lo[3];
  out[4] = hi[0]; out[5] = hi[1]; out[6] = hi[2]; out[7] = hi[3];
  return out;
}

constexpr int TJ = 64, TK = 128, TM = 64;  // output tile 64 (j) x 128 (k); 64 token rows per step

// grid = (tiles_k * tiles_j, S).  partial layout: [S][r][D] fp32, then [S][r] fp32 for the bias sums.
__global__ __launch_bounds__(256) void proj_dw_partial_kernel(const bf16* __restrict__ dyg, const bf16* __restrict__ x,
                                                              int ldx, float* __restrict__ partial, int M, int r,
                                                              int D, int rows_per_slab) {
  __shared__ __attribute__((aligned(16))) char smem[3 * 8192];
  char* Ys = smem;          // [64 m][64 j]
  char* Xs = smem + 8192;   // 2 x [64 m][64 k]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h2 = lane >> 5;
  const int tiles_k = D / TK;
  const int tj = blockIdx.x / tiles_k, tk = blockIdx.x - tj * tiles_k;
  const int j0 = tj * TJ, k0 = tk * TK;
  const int slab = blockIdx.y;
  const int m_begin = slab * rows_per_slab;
  int m_end = m_begin + rows_per_slab;
  m_end = m_end < M ? m_end : M;

  const int wj = (wave & 1) * 32;   // this wave's 32 output rows (j) inside the tile
  const int wk = (wave >> 1) * 64;  // and its 64 output columns (k): two 32-wide MFMA tiles
  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
  float bsum = 0.f;  // threads 0..63 of the tk==0 workgroups: column sum of dyg for j = j0 + tid

  for (int mb = m_begin; mb < m_end; mb += TM) {
    // stage: dyg tile 64x64 (2 chunks/thread), x tile 64x128 (4 chunks/thread); rows >= m_end are zero-filled
    bf16x8 yv[2], xv[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + i * 256, row = e >> 3, chunk = e & 7;
      const int m = mb + row;
      if (m < m_end) yv[i] = *(const bf16x8*)(dyg + (size_t)m * r + j0 + chunk * 8);
      else {
#pragma unroll
        for (int q = 0; q < 8; ++q) yv[i][q] = (bf16)0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * 256, row = (e >> 3) & 63, chunk = e & 7, sub = e >> 9;
      const int m = mb + row;
      if (m < m_end) xv[i] = *(const bf16x8*)(x + (size_t)m * ldx + k0 + sub * 64 + chunk * 8);
      else {
#pragma unroll
        for (int q = 0; q < 8; ++q) xv[i][q] = (bf16)0.f;
      }
    }
    __syncthreads();  // previous step's fragment reads are done
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + i * 256, row = e >> 3, chunk = e & 7;
      *(bf16x8*)(Ys + tile_off(row, chunk)) = yv[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * 256, row = (e >> 3) & 63, chunk = e & 7, sub = e >> 9;
      *(bf16x8*)(Xs + sub * 8192 + tile_off(row, chunk)) = xv[i];
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 a = tr_frag(Ys, 16 * ks, wj, lane);  // A[row = j][k' = m]
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int kc = wk + 32 * t;
        const bf16x8 b = tr_frag(Xs + (kc >> 6) * 8192, 16 * ks, kc & 63, lane);  // B[k' = m][col = k]
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
      }
    }
    if (tk == 0 && tid < 64) {
      const int chunk = tid >> 3, within = (tid & 7) * 2;
#pragma unroll 8
      for (int row = 0; row < 64; ++row) bsum += (float)*(const bf16*)(Ys + tile_off(row, chunk) + within);
    }
  }
  // D[j][k]: lane col = k (lane&31), rows j = acc_row(reg, h2)
  float* P = partial + (size_t)slab * r * D;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int j = j0 + wj + (reg & 3) + 8 * (reg >> 2) + 4 * h2;
      const int k = k0 + wk + 32 * t + (lane & 31);
      P[(size_t)j * D + k] = acc[t][reg];
    }
  if (tk == 0 && tid < 64) partial[(size_t)gridDim.y * r * D + (size_t)slab * r + j0 + tid] = bsum;
}

__global__ __launch_bounds__(256) void proj_dw_reduce_kernel(const float* __restrict__ partial,
                                                             const float* __restrict__ row_scale,
                                                             float* __restrict__ dW1, float* __restrict__ db1, int r,
                                                             int D, int S, int accumulate) {
  const long n4 = (long)r * D / 4;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int sl = 0; sl < S; ++sl) s += *(const f32x4*)(partial + (size_t)sl * r * D + i * 4);
    const int j = (int)(i * 4 / D);
    if (row_scale != nullptr) s *= row_scale[j];
    if (accumulate) s += *(const f32x4*)(dW1 + i * 4);
    *(f32x4*)(dW1 + i * 4) = s;
  }
  if (i < r) {
    const float* pb = partial + (size_t)S * r * D;
    float s = 0.f;
    for (int sl = 0; sl < S; ++sl) s += pb[(size_t)sl * r + i];
    if (row_scale != nullptr) s *= row_scale[i];
    if (accumulate) s += db1[i];
    db1[i] = s;
  }
}

inline int dw_slabs(int M, int r, int D) {
  const int tiles = (r / TJ) * (D / TK);
  int S = (448 + tiles - 1) / tiles;  // ~1.75 workgroups per CU: enough parallelism, fewer partial slabs to write and re-read
  const int max_s = (M + TM - 1) / TM;
  if (S > max_s) S = max_s;
  if (S < 1) S = 1;
  return S;
}

__global__ __launch_bounds__(256) void pack_proj_rows_kernel(const float* __restrict__ W1, const float* __restrict__ b1,
                                                             const int32_t* __restrict__ inds,
                                                             const float* __restrict__ gamma, bf16* __restrict__ Wnat,
                                                             bf16* __restrict__ WnatT, float* __restrict__ bnat, int r,
                                                             int D) {
  const int j = blockIdx.x;
  const int row = inds[j];
  const float g = gamma != nullptr ? gamma[row] : 1.0f;
  for (int k = threadIdx.x; k < D; k += 256) {
    const bf16 v = (bf16)(g * W1[(size_t)j * D + k]);
    Wnat[(size_t)row * D + k] = v;
    WnatT[(size_t)k * D + row] = v;
  }
  if (threadIdx.x == 0 && b1 != nullptr) bnat[row] = g * b1[j];
}

}  // namespace

extern "C" long apla_dw_workspace_bytes(int M, int r, int D) {
  if (M <= 0 || r <= 0 || D <= 0 || r % TJ != 0 || D % TK != 0) return -1;
  const long S = dw_slabs(M, r, D);
  return S * ((long)r * D + r) * (long)sizeof(float);
}

extern "C" int apla_proj_dw(const void* dyg, const void* x, int ldx, const float* row_scale, float* dW1, float* db1,
                            void* partial, int M, int r, int D, int accumulate, hipStream_t stream) {
  APLA_REQUIRE(dyg && x && dW1 && db1 && partial, "apla_proj_dw: null pointer");
  APLA_REQUIRE(M > 0 && r > 0 && r % TJ == 0 && D % TK == 0, "apla_proj_dw: need r%%64==0 and D%%128==0 (r=%d D=%d)", r, D);
  APLA_REQUIRE(ldx % 8 == 0 && ldx >= D && apla_aligned16(dyg) && apla_aligned16(x) && apla_aligned16(dW1) && apla_aligned16(partial), "apla_proj_dw: alignment");
  const int S = dw_slabs(M, r, D);
  int rows_per_slab = ((M + S - 1) / S + TM - 1) / TM * TM;
  hipLaunchKernelGGL(proj_dw_partial_kernel, dim3((r / TJ) * (D / TK), S), dim3(256), 0, stream, (const bf16*)dyg, (const bf16*)x, ldx, (float*)partial, M, r, D, rows_per_slab);
  APLA_CHECK_LAUNCH("apla_proj_dw[partial]");
  const long n4 = (long)r * D / 4;
  hipLaunchKernelGGL(proj_dw_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, (const float*)partial, row_scale, dW1, db1, r, D, S, accumulate);
  APLA_CHECK_LAUNCH("apla_proj_dw[reduce]");
  return APLA_OK;
}

extern "C" int apla_pack_proj_rows(const float* W1, const float* b1, const int32_t* inds, const float* gamma,
                                   void* Wnat, void* WnatT, float* bnat, int r, int D, hipStream_t stream) {
  APLA_REQUIRE(W1 && inds && Wnat && WnatT && r > 0 && r <= D, "apla_pack_proj_rows: bad arguments");
  APLA_REQUIRE(b1 == nullptr || bnat != nullptr, "apla_pack_proj_rows: bnat required with b1");
  hipLaunchKernelGGL(pack_proj_rows_kernel, dim3(r), dim3(256), 0, stream, W1, b1, inds, gamma, (bf16*)Wnat, (bf16*)WnatT, bnat, r, D);
  APLA_CHECK_LAUNCH("apla_pack_proj_rows");
  return APLA_OK;
}
