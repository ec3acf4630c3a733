// LayerNorm forward / backward-dX for gfx950.  HBM-bound row kernels: one wave per token row, the row lives in
// registers (<= 2048 features), 8/16-byte vector loads, two-pass statistics in fp32.
//
// Backward computes only dX (gamma/beta are frozen under APLA) and fuses (a) the residual-gradient add and (b) the
// gather of the r APLA-trainable columns of the result (bf16) that the column-masked dW1 kernel consumes.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int MAXC = 6;  // row chunks of 4 elements per lane: D <= 64*4*MAXC = 1536 = ViT-g, the widest BASELINE model (kernels are instantiated
                         // per chunk count; an 8-chunk backward spilled 448-896 registers to scratch and no configuration used it)
#ifndef APLA_LN_ROWS
#define APLA_LN_ROWS 4
#endif
constexpr int ROWS_PER_BLOCK = APLA_LN_ROWS;  // one wave per row
constexpr int LN_THREADS = 64 * ROWS_PER_BLOCK;

// Optional fused residual add: x_new = x + add (add = bf16 branch output of the previous GEMM) is formed in registers,
// written to xout and normalised in the same pass — the "x = x + branch" of vit.py:284-285 costs no kernel of its own and
// no fp32 read-modify-write in a GEMM epilogue.
template <typename ResT, typename YT, int NC, bool DROP = false>
__global__ __launch_bounds__(LN_THREADS) void ln_fwd_kernel(const ResT* x, long xs, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, YT* __restrict__ y, int ldy,
                                                     float* __restrict__ mean_o, float* __restrict__ rstd_o, int M,
                                                     int D, float eps, const bf16* __restrict__ add, long adds,
                                                     ResT* xout, long xouts, const float* __restrict__ add_scale, int scale_period,
                                                     DropArgsEw dr, long drop_row_stride) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunk = D >> 2;
  for (int m = blockIdx.x * ROWS_PER_BLOCK + wave; m < M; m += gridDim.x * ROWS_PER_BLOCK) {
    const ResT* xr = x + (size_t)m * xs;
    // stochastic depth (vit.py:74-93, :284-285): the branch of sample m / scale_period enters with its per-sample factor 0 or 1 / keep_prob
    const float bscale = add_scale != nullptr ? add_scale[m / scale_period] : 1.0f;
    f32x4 v[NC];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunk) {
        v[c] = Vec4IO<ResT>::load(xr + ch * 4);
        if (add != nullptr) {
          f32x4 bv = Vec4IO<bf16>::load(add + (size_t)m * adds + ch * 4) * bscale;
          if constexpr (DROP) {   // nn.Dropout on the branch (proj_drop appla_attn.py:82, Mlp.drop after fc2 vit.py:166-167): element m * stride + col
            unsigned w[4];
            drop_words4(dr, (((unsigned long long)m * (unsigned long long)drop_row_stride) >> 2) + ch, w);
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[e] = w[e] >= dr.threshold ? bv[e] * dr.inv_keep : 0.f;
          }
          v[c] += bv;
          Vec4IO<ResT>::store(xout + (size_t)m * xouts + ch * 4, v[c]);
          if constexpr (sizeof(ResT) == 2) {  // statistics of the value as stored (bf16-rounded), like a separate LN pass would see
#pragma unroll
            for (int e = 0; e < 4; ++e) v[c][e] = (float)(bf16)v[c][e];
          }
        }
        s += v[c][0] + v[c][1] + v[c][2] + v[c][3];
      }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunk) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[c][e] - mean; q += d * d; }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    if (lane == 0) { mean_o[m] = mean; rstd_o[m] = rstd; }
    YT* yr = y + (size_t)m * ldy;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int ch = lane + c * 64;
      if (ch < nchunk) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[c][e] - mean) * rstd;
        if (gamma != nullptr) {   // NULL: y = xhat (gamma / beta folded into the weights of the GEMM that reads y)
          const f32x4 g = *(const f32x4*)(gamma + ch * 4), b = *(const f32x4*)(beta + ch * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = o[e] * g[e] + b[e];
        }
        Vec4IO<YT>::store(yr + ch * 4, o);
      }
    }
  }
}

// dx_out = dres_in + ((dy*g) - mean(dy*g) - xhat*mean(dy*g*xhat)) * rstd ; optional bf16 copy of dx_out (GEMM operand
// when the gradient stream is fp32) and optional gather of the APLA-trainable columns.
template <typename XT, typename DYT, typename GT, bool GATHER, int NC>
__global__ __launch_bounds__(LN_THREADS) void ln_bwd_kernel(const DYT* __restrict__ dy, int lddy, const XT* __restrict__ x,
                                                     long xs, const float* __restrict__ gamma,
                                                     const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                     const GT* dres, GT* dx, long dxs, bf16* __restrict__ dxb, long dbs,
                                                     const int32_t* __restrict__ inds, int r, bf16* __restrict__ gout,
                                                     int M, int D, int dres_period, const float* __restrict__ dy_scale,
                                                     const float* __restrict__ g_scale, int scale_period) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* rowbuf = (float*)smem_raw;  // [ROWS_PER_BLOCK][D] when GATHER
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunk = D >> 2;
  for (int m0 = blockIdx.x * ROWS_PER_BLOCK; m0 < M; m0 += gridDim.x * ROWS_PER_BLOCK) {
    const int m = m0 + wave;
    if (m < M) {
      const XT* xr = x + (size_t)m * xs;
      const DYT* dyr = dy + (size_t)m * lddy;
      // mean_i == NULL: x holds the NORMALISED row (what the forward wrote with gamma == NULL), saved in 16 bits: the backward
      // then reads 2 bytes per element instead of the 4 of the fp32 residual row
      // stochastic depth: dy_scale = the per-sample factor of the branch this LayerNorm fed (its gradient is linear in the factor, so
      // it is applied here, at the end of the branch's backward chain); g_scale = the factor of the branch whose dW reads `gout`
      const float mean = mean_i != nullptr ? mean_i[m] : 0.f, rstd = rstd_i[m] * (dy_scale != nullptr ? dy_scale[m / scale_period] : 1.0f);
      const float xsc = mean_i != nullptr ? rstd_i[m] : 1.0f;
      const float gsc = g_scale != nullptr ? g_scale[m / scale_period] : 1.0f;
      // dres_period > 1: only every dres_period-th row of the incoming residual gradient is non-zero (the CLS rows below the
      // final norm, vit.py:416-419) and the rest is NOT read — no zero-fill of the stream, no read of zeros
      const bool has_res = dres != nullptr && (dres_period <= 1 || m % dres_period == 0);
      f32x4 xh[NC], w[NC];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int ch = lane + c * 64;
        if (ch < nchunk) {
          const f32x4 xv = Vec4IO<XT>::load(xr + ch * 4);
          const f32x4 dv = Vec4IO<DYT>::load(dyr + ch * 4);
          f32x4 g = f32x4{1.f, 1.f, 1.f, 1.f};
          if (gamma != nullptr) g = *(const f32x4*)(gamma + ch * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            xh[c][e] = (xv[e] - mean) * xsc;
            w[c][e] = dv[e] * g[e];
            s1 += w[c][e];
            s2 += w[c][e] * xh[c][e];
          }
        }
      }
      const float c1 = wave_sum(s1) / (float)D, c2 = wave_sum(s2) / (float)D;
      GT* dxr = dx + (size_t)m * dxs;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int ch = lane + c * 64;
        if (ch < nchunk) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (w[c][e] - c1 - xh[c][e] * c2) * rstd;
          if (has_res) o += Vec4IO<GT>::load(dres + (size_t)m * dxs + ch * 4);
          Vec4IO<GT>::store(dxr + ch * 4, o);
          if (dxb != nullptr) Vec4IO<bf16>::store(dxb + (size_t)m * dbs + ch * 4, o);
          if constexpr (GATHER) *(f32x4*)(rowbuf + wave * D + ch * 4) = o * gsc;
        }
      }
    }
    if constexpr (GATHER) {
      __syncthreads();
      if (m < M) {
        for (int j = lane; j < r; j += 64) gout[(size_t)m * r + j] = (bf16)rowbuf[wave * D + inds[j]];
      }
      __syncthreads();
    }
  }
}

// The backward as the fused step runs it (normalised row xhat in 16 bits, no affine part, 16-bit gradient stream): TWO rows per
// wave, a half-wave each, 8 elements (16 bytes) per lane and access.  The one-row-per-wave form above moves 8 bytes per lane and
// access for 16-bit operands and reached 4.3 TB/s on these shapes; rows of 768 / 1024 / 1536 elements are 3 / 4 / 6 whole chunks
// per lane here.  dx = dres + (dy - mean(dy) - xhat * mean(dy * xhat)) * rstd.
template <bool GATHER, int NC8, bool MASKED = false>
__global__ __launch_bounds__(LN_THREADS) void ln_bwd2_kernel(const bf16* __restrict__ dy, int lddy, const bf16* __restrict__ xh,
                                                      long xs, const float* __restrict__ rstd_i, const bf16* dres, bf16* dx,
                                                      long dxs, const int32_t* __restrict__ inds, int r,
                                                      bf16* __restrict__ gout, int M, int D, int dres_period,
                                                      const float* __restrict__ dy_scale, const float* __restrict__ g_scale, int scale_period,
                                                      bf16* __restrict__ masked, long mks, DropArgsEw dr, const float* __restrict__ mask_scale) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* rowbuf = (float*)smem_raw;  // [2 * ROWS_PER_BLOCK][D] when GATHER
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hl = lane & 31, half = lane >> 5;
  const int nchunk = D >> 3;
  const float invD = 1.0f / (float)D;
  for (int m0 = blockIdx.x * (2 * ROWS_PER_BLOCK); m0 < M; m0 += gridDim.x * (2 * ROWS_PER_BLOCK)) {
    const int m = m0 + 2 * wave + half;
    const bool live = m < M;
    const int mr = live ? m : M - 1;        // a dead half-wave re-reads the last row (keeps the cross-lane sums uniform), stores nothing
    const bf16* xr = xh + (size_t)mr * xs;
    const bf16* dyr = dy + (size_t)mr * lddy;
    const float rstd = rstd_i[mr] * (dy_scale != nullptr ? dy_scale[mr / scale_period] : 1.0f);     // (stochastic depth: see ln_bwd_kernel)
    const float gsc = g_scale != nullptr ? g_scale[mr / scale_period] : 1.0f;
    // `masked` (optional): a second copy of dx with the NEXT consumer's dropout mask (and its branch's per-sample factor) applied — the
    // operand of that branch's dX GEMM; the gathered columns of dW1 are then taken from it
    const float msc = (MASKED && mask_scale != nullptr) ? mask_scale[mr / scale_period] : 1.0f;
    const bool has_res = dres != nullptr && (dres_period <= 1 || mr % dres_period == 0);
    bf16x8 xv[NC8], dv[NC8], rv[NC8];
#pragma unroll
    for (int c = 0; c < NC8; ++c) {
      const int ch = hl + c * 32;
      if (ch < nchunk) {
        xv[c] = *(const bf16x8*)(xr + ch * 8);
        dv[c] = *(const bf16x8*)(dyr + ch * 8);
        if (has_res) rv[c] = *(const bf16x8*)(dres + (size_t)mr * dxs + ch * 8);
      }
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NC8; ++c) {
      if (hl + c * 32 < nchunk) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float w = (float)dv[c][e];
          s1 += w;
          s2 += w * (float)xv[c][e];
        }
      }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }   // within the half-wave
    const float c1 = s1 * invD, c2 = s2 * invD;
#pragma unroll
    for (int c = 0; c < NC8; ++c) {
      const int ch = hl + c * 32;
      if (ch < nchunk) {
        bf16x8 ov, mv;
        unsigned w[8];
        if constexpr (MASKED) {
          unsigned a[4], b[4];
          const unsigned long long blk = (((unsigned long long)mr * (unsigned long long)D) >> 2) + 2 * ch;
          drop_words4(dr, blk, a);
          drop_words4(dr, blk + 1, b);
#pragma unroll
          for (int e = 0; e < 4; ++e) { w[e] = a[e]; w[4 + e] = b[e]; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float o = ((float)dv[c][e] - c1 - (float)xv[c][e] * c2) * rstd;
          if (has_res) o += (float)rv[c][e];
          ov[e] = (bf16)o;
          float g = o * gsc;
          if constexpr (MASKED) {     // the mask acts on the value the stream holds (16-bit), as the stand-alone pass would
            g = w[e] >= dr.threshold ? (float)ov[e] * dr.inv_keep * msc : 0.f;
            mv[e] = (bf16)g;
            g = (float)mv[e];
          }
          if constexpr (GATHER) rowbuf[(2 * wave + half) * D + ch * 8 + e] = g;
        }
        if (live) *(bf16x8*)(dx + (size_t)m * dxs + ch * 8) = ov;
        if constexpr (MASKED) { if (live) *(bf16x8*)(masked + (size_t)m * mks + ch * 8) = mv; }
      }
    }
    if constexpr (GATHER) {
      __syncthreads();
      if (live)
        for (int j = hl; j < r; j += 32) gout[(size_t)m * r + j] = (bf16)rowbuf[(2 * wave + half) * D + inds[j]];
      __syncthreads();
    }
  }
}

template <typename ResT>
__global__ __launch_bounds__(LN_THREADS) void gather_cols_kernel(const ResT* __restrict__ src, long ss,
                                                          const int32_t* __restrict__ inds, int r,
                                                          bf16* __restrict__ out, int M) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int m = blockIdx.x * ROWS_PER_BLOCK + wave; m < M; m += gridDim.x * ROWS_PER_BLOCK)
    for (int j = lane; j < r; j += 64) out[(size_t)m * r + j] = (bf16)(float)src[(size_t)m * ss + inds[j]];
}

inline int ln_grid(int M) {
  int g = (M + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
#if defined(APLA_ABL_LNGRID)  // diagnostic build (tools/build_ablations.sh): grid cap from the environment
  const char* e = getenv("APLA_LN_GRID");
  const int cap = e ? atoi(e) : 4096;
#else
  constexpr int cap = 4096;
#endif
  return g < cap ? g : cap;
}

}  // namespace

extern "C" int apla_layernorm_fwd_drop(const void* x, int res_dtype, long x_row_stride, const float* gamma,
                                       const float* beta, void* y, int y_dtype, int ldy, float* mean, float* rstd, int M,
                                       int D, float eps, const void* add_in, long add_row_stride, void* x_out,
                                       long x_out_row_stride, const float* add_scale, int scale_period, const unsigned long long* rng,
                                       unsigned long long rng_stride, unsigned site, float p, long drop_row_stride, hipStream_t stream) {
  APLA_REQUIRE(add_scale == nullptr || (add_in != nullptr && scale_period >= 1), "apla_layernorm_fwd_dp: a per-sample scale needs add_in and scale_period >= 1");
  APLA_REQUIRE(rng == nullptr || (add_in != nullptr && p >= 0.f && p < 1.f && drop_row_stride >= D && drop_row_stride % 4 == 0),
               "apla_layernorm_fwd_drop: dropout needs add_in, 0 <= p < 1 and an index row stride >= D that is a multiple of 4");
  const DropArgsEw dr{rng, rng_stride, site, rng ? apla_drop_threshold(p) : 0u, rng ? 1.0f / (1.0f - p) : 1.0f};
  if (add_scale == nullptr) scale_period = 1;
  APLA_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && D <= 256 * MAXC, "apla_layernorm_fwd: need D%%4==0 and D<=1536 (D=%d)", D);
  APLA_REQUIRE(x && y && mean && rstd && ((gamma == nullptr) == (beta == nullptr)), "apla_layernorm_fwd: null pointer (gamma and beta are given together or not at all)");
  APLA_REQUIRE(x_row_stride % 4 == 0 && ldy % 4 == 0 && x_row_stride >= D && ldy >= D, "apla_layernorm_fwd: strides must be >= D and multiples of 4");
  APLA_REQUIRE(apla_aligned16(x) && (gamma == nullptr || (apla_aligned16(gamma) && apla_aligned16(beta))) && (((uintptr_t)y) & 7) == 0, "apla_layernorm_fwd: alignment");
  APLA_REQUIRE(add_in == nullptr || (x_out != nullptr && add_row_stride % 4 == 0 && add_row_stride >= D && x_out_row_stride % 4 == 0 && x_out_row_stride >= D),
               "apla_layernorm_fwd: fused residual add needs x_out and valid strides");
#define LN_FWD_NC(T, Y, NCV) if (rng != nullptr) LN_FWD_NC2(T, Y, NCV, true); else LN_FWD_NC2(T, Y, NCV, false)
#define LN_FWD_NC2(T, Y, NCV, DR) hipLaunchKernelGGL((ln_fwd_kernel<T, Y, NCV, DR>), dim3(ln_grid(M)), dim3(LN_THREADS), 0, stream, (const T*)x, x_row_stride, gamma, beta, (Y*)y, ldy, mean, rstd, M, D, eps, (const bf16*)add_in, add_row_stride, (T*)x_out, x_out_row_stride, add_scale, scale_period, dr, drop_row_stride)
#define LN_FWD(T, Y)                                    \
  do {                                                  \
    const int nc_ = (D + 255) / 256;                    \
    if (nc_ <= 1) { LN_FWD_NC(T, Y, 1); }               \
    else if (nc_ == 2) { LN_FWD_NC(T, Y, 2); }          \
    else if (nc_ == 3) { LN_FWD_NC(T, Y, 3); }          \
    else if (nc_ == 4) { LN_FWD_NC(T, Y, 4); }          \
    else { LN_FWD_NC(T, Y, 6); }                        \
  } while (0)
  if (res_dtype == APLA_F32 && y_dtype == APLA_H16) LN_FWD(float, bf16);
  else if (res_dtype == APLA_F32 && y_dtype == APLA_F32) LN_FWD(float, float);
  else if (res_dtype == APLA_H16 && y_dtype == APLA_H16) LN_FWD(bf16, bf16);
  else if (res_dtype == APLA_H16 && y_dtype == APLA_F32) LN_FWD(bf16, float);
  else {
    apla_set_error("apla_layernorm_fwd: bad res_dtype %d / y_dtype %d", res_dtype, y_dtype);
    return APLA_ENOSYS;
  }
#undef LN_FWD
#undef LN_FWD_NC
#undef LN_FWD_NC2
  APLA_CHECK_LAUNCH("apla_layernorm_fwd");
  return APLA_OK;
}

extern "C" int apla_layernorm_fwd_dp(const void* x, int res_dtype, long x_row_stride, const float* gamma,
                                     const float* beta, void* y, int y_dtype, int ldy, float* mean, float* rstd, int M,
                                     int D, float eps, const void* add_in, long add_row_stride, void* x_out,
                                     long x_out_row_stride, const float* add_scale, int scale_period, hipStream_t stream) {
  return apla_layernorm_fwd_drop(x, res_dtype, x_row_stride, gamma, beta, y, y_dtype, ldy, mean, rstd, M, D, eps, add_in, add_row_stride,
                                 x_out, x_out_row_stride, add_scale, scale_period, nullptr, 0, 0, 0.f, D, stream);
}

extern "C" int apla_layernorm_fwd(const void* x, int res_dtype, long x_row_stride, const float* gamma,
                                  const float* beta, void* y, int y_dtype, int ldy, float* mean, float* rstd, int M,
                                  int D, float eps, const void* add_in, long add_row_stride, void* x_out,
                                  long x_out_row_stride, hipStream_t stream) {
  return apla_layernorm_fwd_dp(x, res_dtype, x_row_stride, gamma, beta, y, y_dtype, ldy, mean, rstd, M, D, eps, add_in, add_row_stride,
                               x_out, x_out_row_stride, nullptr, 1, stream);
}

extern "C" int apla_layernorm_bwd_drop(const void* dy, int dy_dtype, int lddy, const void* x, int x_dtype, long x_row_stride,
                                       const float* gamma, const float* mean, const float* rstd, const void* dres_in,
                                       int dres_row_period, void* dx_out, int grad_dtype, long dx_row_stride, void* dx_bf16_copy,
                                       long copy_row_stride, const int32_t* inds, int r, void* gather_out, int M, int D,
                                       const float* dy_scale, const float* gather_scale, int scale_period, void* masked_out,
                                       long masked_row_stride, const float* mask_scale, const unsigned long long* rng,
                                       unsigned long long rng_stride, unsigned site, float p, hipStream_t stream) {
  APLA_REQUIRE((dy_scale == nullptr && gather_scale == nullptr && mask_scale == nullptr) || scale_period >= 1, "apla_layernorm_bwd_dp: scale_period >= 1");
  if (dy_scale == nullptr && gather_scale == nullptr && mask_scale == nullptr) scale_period = 1;
  const bool fast = mean == nullptr && gamma == nullptr && x_dtype == APLA_H16 && dy_dtype == APLA_H16 && grad_dtype == APLA_H16 &&
                    dx_bf16_copy == nullptr && D % 8 == 0;
  APLA_REQUIRE(masked_out == nullptr || (fast && rng != nullptr && p >= 0.f && p < 1.f && masked_row_stride >= D && masked_row_stride % 8 == 0 &&
                                         apla_aligned16(masked_out) && gather_scale == nullptr),
               "apla_layernorm_bwd_drop: the masked copy needs the 16-bit normalised-row form (mean = gamma = NULL, 16-bit operands and stream), rng, "
               "0 <= p < 1, an aligned buffer and no gather_scale (the branch's factor goes into mask_scale)");
  const DropArgsEw dr{masked_out ? rng : nullptr, rng_stride, site, masked_out ? apla_drop_threshold(p) : 0u, masked_out ? 1.0f / (1.0f - p) : 1.0f};
  APLA_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && D <= 256 * MAXC, "apla_layernorm_bwd: need D%%4==0 and D<=1536 (D=%d)", D);
  APLA_REQUIRE(dy && x && rstd && dx_out && dres_row_period >= 0, "apla_layernorm_bwd: null pointer");
  APLA_REQUIRE(lddy % 4 == 0 && x_row_stride % 4 == 0 && dx_row_stride % 4 == 0 && lddy >= D && x_row_stride >= D && dx_row_stride >= D, "apla_layernorm_bwd: bad strides");
  APLA_REQUIRE(dx_bf16_copy == nullptr || (copy_row_stride % 4 == 0 && copy_row_stride >= D), "apla_layernorm_bwd: bad copy stride");
  APLA_REQUIRE(gather_out == nullptr || (inds != nullptr && r > 0 && r <= D), "apla_layernorm_bwd: gather needs inds and 0<r<=D");
  const bool gather = gather_out != nullptr;
  // the fused step's case: xhat and dy in 16 bits, 16-bit gradient stream without a separate copy, no affine part, whole 8-element
  // chunks -> the two-rows-per-wave kernel
  if (mean == nullptr && gamma == nullptr && x_dtype == APLA_H16 && dy_dtype == APLA_H16 && grad_dtype == APLA_H16 &&
      dx_bf16_copy == nullptr && D % 8 == 0 && D <= 256 * 8 && lddy % 8 == 0 && x_row_stride % 8 == 0 && dx_row_stride % 8 == 0 &&
      apla_aligned16(dy) && apla_aligned16(x) && apla_aligned16(dx_out) && (dres_in == nullptr || apla_aligned16(dres_in))) {
    const int nc8 = (D / 8 + 31) / 32;
    const int g2 = (M + 2 * ROWS_PER_BLOCK - 1) / (2 * ROWS_PER_BLOCK);
    const dim3 grid2(g2 < 4096 ? g2 : 4096);
    const size_t lds2 = gather ? (size_t)2 * ROWS_PER_BLOCK * D * sizeof(float) : 0;
#define LN_BWD2(GA, NCV) do { if (masked_out != nullptr) LN_BWD2M(GA, NCV, true); else LN_BWD2M(GA, NCV, false); } while (0)
#define LN_BWD2M(GA, NCV, MK)                                                                                                      \
    hipLaunchKernelGGL((ln_bwd2_kernel<GA, NCV, MK>), grid2, dim3(LN_THREADS), lds2, stream, (const bf16*)dy, lddy, (const bf16*)x,      \
                       x_row_stride, rstd, (const bf16*)dres_in, (bf16*)dx_out, dx_row_stride, inds, r, (bf16*)gather_out, M, D,    \
                       dres_row_period, dy_scale, gather_scale, scale_period, (bf16*)masked_out, masked_row_stride, dr, mask_scale)
#define LN_BWD2_G(NCV) do { if (gather) LN_BWD2(true, NCV); else LN_BWD2(false, NCV); } while (0)
    if (nc8 <= 1) LN_BWD2_G(1);
    else if (nc8 == 2) LN_BWD2_G(2);
    else if (nc8 == 3) LN_BWD2_G(3);
    else if (nc8 == 4) LN_BWD2_G(4);
    else if (nc8 <= 6) LN_BWD2_G(6);
    else LN_BWD2_G(8);
#undef LN_BWD2_G
#undef LN_BWD2
#undef LN_BWD2M
    APLA_CHECK_LAUNCH("apla_layernorm_bwd");
    return APLA_OK;
  }
  APLA_REQUIRE(masked_out == nullptr, "apla_layernorm_bwd_drop: the masked copy exists on the 16-bit two-rows-per-wave form only (strides % 8, 16-byte aligned operands)");
  const size_t lds = gather ? (size_t)ROWS_PER_BLOCK * D * sizeof(float) : 0;
#define LN_BWD_NC(X, Y, G, GA, NCV)                                                                                    \
  hipLaunchKernelGGL((ln_bwd_kernel<X, Y, G, GA, NCV>), dim3(ln_grid(M)), dim3(LN_THREADS), lds, stream, (const Y*)dy, lddy,  \
                     (const X*)x, x_row_stride, gamma, mean, rstd, (const G*)dres_in, (G*)dx_out, dx_row_stride,       \
                     (bf16*)dx_bf16_copy, copy_row_stride, inds, r, (bf16*)gather_out, M, D, dres_row_period, dy_scale, gather_scale, scale_period)
#define LN_BWD(X, Y, G, GA)                             \
  do {                                                  \
    const int nc_ = (D + 255) / 256;                    \
    if (nc_ <= 1) LN_BWD_NC(X, Y, G, GA, 1);            \
    else if (nc_ == 2) LN_BWD_NC(X, Y, G, GA, 2);       \
    else if (nc_ == 3) LN_BWD_NC(X, Y, G, GA, 3);       \
    else if (nc_ == 4) LN_BWD_NC(X, Y, G, GA, 4);       \
    else LN_BWD_NC(X, Y, G, GA, 6);                     \
  } while (0)
#define LN_BWD_G(X, Y, G) do { if (gather) LN_BWD(X, Y, G, true); else LN_BWD(X, Y, G, false); } while (0)
#define LN_BWD_Y(X, G)                                           \
  do {                                                           \
    if (dy_dtype == APLA_H16) LN_BWD_G(X, bf16, G);             \
    else LN_BWD_G(X, float, G);                                  \
  } while (0)
  const bool ok = (x_dtype == APLA_F32 || x_dtype == APLA_H16) && (dy_dtype == APLA_F32 || dy_dtype == APLA_H16) &&
                  (grad_dtype == APLA_F32 || grad_dtype == APLA_H16);
  if (!ok) {
    apla_set_error("apla_layernorm_bwd: bad dtypes x=%d dy=%d grad=%d", x_dtype, dy_dtype, grad_dtype);
    return APLA_ENOSYS;
  }
  if (x_dtype == APLA_F32 && grad_dtype == APLA_F32) LN_BWD_Y(float, float);
  else if (x_dtype == APLA_F32 && grad_dtype == APLA_H16) LN_BWD_Y(float, bf16);
  else if (x_dtype == APLA_H16 && grad_dtype == APLA_F32) LN_BWD_Y(bf16, float);
  else LN_BWD_Y(bf16, bf16);
#undef LN_BWD_Y
#undef LN_BWD_G
#undef LN_BWD
#undef LN_BWD_NC
  APLA_CHECK_LAUNCH("apla_layernorm_bwd");
  return APLA_OK;
}

extern "C" int apla_layernorm_bwd_dp(const void* dy, int dy_dtype, int lddy, const void* x, int x_dtype, long x_row_stride,
                                     const float* gamma, const float* mean, const float* rstd, const void* dres_in,
                                     int dres_row_period, void* dx_out, int grad_dtype, long dx_row_stride, void* dx_bf16_copy,
                                     long copy_row_stride, const int32_t* inds, int r, void* gather_out, int M, int D,
                                     const float* dy_scale, const float* gather_scale, int scale_period, hipStream_t stream) {
  return apla_layernorm_bwd_drop(dy, dy_dtype, lddy, x, x_dtype, x_row_stride, gamma, mean, rstd, dres_in, dres_row_period, dx_out, grad_dtype,
                                 dx_row_stride, dx_bf16_copy, copy_row_stride, inds, r, gather_out, M, D, dy_scale, gather_scale, scale_period,
                                 nullptr, 0, nullptr, nullptr, 0, 0, 0.f, stream);
}

extern "C" int apla_layernorm_bwd_ex(const void* dy, int dy_dtype, int lddy, const void* x, int x_dtype, long x_row_stride,
                                     const float* gamma, const float* mean, const float* rstd, const void* dres_in,
                                     int dres_row_period, void* dx_out, int grad_dtype, long dx_row_stride, void* dx_bf16_copy,
                                     long copy_row_stride, const int32_t* inds, int r, void* gather_out, int M, int D,
                                     hipStream_t stream) {
  return apla_layernorm_bwd_dp(dy, dy_dtype, lddy, x, x_dtype, x_row_stride, gamma, mean, rstd, dres_in, dres_row_period, dx_out, grad_dtype,
                               dx_row_stride, dx_bf16_copy, copy_row_stride, inds, r, gather_out, M, D, nullptr, nullptr, 1, stream);
}

extern "C" int apla_layernorm_bwd(const void* dy, int dy_dtype, int lddy, const void* x, int x_dtype, long x_row_stride,
                                  const float* gamma, const float* mean, const float* rstd, const void* dres_in,
                                  void* dx_out, int grad_dtype, long dx_row_stride, void* dx_bf16_copy,
                                  long copy_row_stride, const int32_t* inds, int r, void* gather_out, int M, int D,
                                  hipStream_t stream) {
  APLA_REQUIRE(gamma && mean, "apla_layernorm_bwd: null pointer (apla_layernorm_bwd_ex takes a normalised row / no gamma)");
  return apla_layernorm_bwd_ex(dy, dy_dtype, lddy, x, x_dtype, x_row_stride, gamma, mean, rstd, dres_in, 0, dx_out, grad_dtype,
                               dx_row_stride, dx_bf16_copy, copy_row_stride, inds, r, gather_out, M, D, stream);
}

extern "C" int apla_gather_cols(const void* src, int res_dtype, long src_row_stride, const int32_t* inds, int r,
                                void* out, int M, int D, hipStream_t stream) {
  APLA_REQUIRE(src && inds && out && M > 0 && r > 0 && r <= D, "apla_gather_cols: bad arguments");
  if (res_dtype == APLA_F32)
    hipLaunchKernelGGL(gather_cols_kernel<float>, dim3(ln_grid(M)), dim3(LN_THREADS), 0, stream, (const float*)src, src_row_stride, inds, r, (bf16*)out, M);
  else if (res_dtype == APLA_H16)
    hipLaunchKernelGGL(gather_cols_kernel<bf16>, dim3(ln_grid(M)), dim3(LN_THREADS), 0, stream, (const bf16*)src, src_row_stride, inds, r, (bf16*)out, M);
  else {
    apla_set_error("apla_gather_cols: bad res_dtype %d", res_dtype);
    return APLA_ENOSYS;
  }
  APLA_CHECK_LAUNCH("apla_gather_cols");
  return APLA_OK;
}
