// Shared pieces of the bf16 MFMA GEMM kernels (gemm_nt.hip: 4-wave kernels, 128-wide tiles; gemm_pp2.hip: 8-wave ping-pong kernel,
// 320 x 256 tiles; gemm_small.hip: few-row split-K).
#pragma once
#include <cstdlib>
#include <type_traits>

#include "common.h"

struct GemmParams {
  const bf16* A; int lda;
  const bf16* W; int ldw;
  const float* bias;
  void* C; int ldc;
  const void* aux_in; int ld_aux_in;
  void* aux_out; int ld_aux_out;
  int M, N, K, tiles_n;
  int ngrp;  // n-tiles per column group of the tile walk (0 = plain row-major walk); see tile_coords
  int w_panel;  // operand images (apla_gemm_nt_ex flags bits 16 / 17): bit 0 = W is K-panel-major [K/32][N][32], bit 1 = A is
                // [K/32][M][32].  In that image the 64 bytes a row gives to a 32-wide K-step sit next to its neighbours' (whole
                // 128-byte lines per LDS-DMA instruction: tools/dma_probe.hip); ping-pong kernel only.  Bit 2 = the OUTPUT C is
                // written as its K-panel image [Nout/32][M][32] (16-bit, 4-wave persistent kernel: GELU / GELU_FWD / MUL and the
                // two SwiGLU epilogues, whose C is N/2 resp. 2N wide), ready to be the A operand of the next GEMM.  Bit 3 = the second operand of the epilogue (aux_out of GELU, aux_in of MUL: gelu'
                // saved by the forward for the backward) is an image [N/32][M][32] too: a tensor private to those two epilogues,
                // stored and loaded in whole lines instead of 64-byte row pieces.
  int reserve;  // CUs this launch leaves free (apla_gemm_nt_ex flags bits 20-27): the persistent kernels start 256 - reserve (ping-pong)
                // or 2 * (256 - reserve) (4-wave) workgroups, so that the kernels of a concurrent collective find CUs of their own
  int exp;   // experiment selector (apla_gemm_nt_ex flags bits 28-30; tools/gemm_bench.py): schedule variants of the 4-wave kernel's
             // K loop compiled beside the product one for A/B timing inside one process; 0 = product
  DropArgsEw drop;   // element-wise dropout of the GELU epilogue's two outputs (apla_gemm_nt_gelu_drop; rng == NULL: none)
  int tag;   // profiling tag (apla_gemm_nt_tagged): selects one of several identical kernel instantiations so that a rocprofv3
             // kernel trace tells the call sites of the step apart (qkv / proj / fc2 / dfc1 / dproj / dqkv …); 0 = untagged
};
constexpr int APLA_GEMM_TAGS = 9;   // tags 0 .. 8 (1 is not used: the profiler's demangler garbles that instantiation's name)

// 8-wave ping-pong kernel (gemm_pp2.hip); returns APLA_ENOSYS when the shape / (epilogue, dtype) is not covered there
int apla_gemm_pp2_launch(const GemmParams& p, int epilogue, int out_dtype, hipStream_t stream);
bool apla_gemm_pp2_covers(int M, int N, int K, long lda, long ldw, int epilogue, int out_dtype);
int apla_gemm_pp2_tile_rows(int M, int N, int epilogue, int out_dtype, int exp, int reserve);   // 320, or 256 where that takes fewer tile-rows of work
int apla_gemm_w4_launch(const GemmParams& p, int epilogue, int out_dtype, hipStream_t stream);   // gemm_w4.hip
bool apla_gemm_w4_covers(int M, int N, int K, long lda, long ldw, int epilogue, int out_dtype);
int apla_gemm_w4_tile_rows(int M, int N, int epilogue, int out_dtype, int exp, int reserve);    // 160, or 128
int apla_gemm_tp_launch(const GemmParams& p, int epilogue, int out_dtype, hipStream_t stream);   // gemm_tp.hip
bool apla_gemm_tp_covers(int M, int N, int K, long lda, long ldw, int epilogue, int out_dtype, int w_panel);
int apla_gemm_lw_launch(const GemmParams& p, int epilogue, int out_dtype, hipStream_t stream);   // gemm_lw.hip (loader-wave form of the 4-wave kernel)
bool apla_gemm_lw_covers(int M, int N, int K, long lda, long ldw, int epilogue, int out_dtype, int w_panel);

// Tile walk.  Linear tile ids are dealt to XCDs in contiguous runs (each XCD has its own 4 MB L2).  With n fastest, a run
// touches ALL column tiles, i.e. the whole weight matrix: fine while W fits next to the streaming A panels (N = 768:
// 1.2 MB), but for N = 2304 / 3072 (3.5 / 4.7 MB of W) the L2 thrashes and W is re-fetched from beyond L2 all the time
// (PMC: FETCH_SIZE 5x the algorithmic bytes).  So wide problems are walked in column groups of `ngrp` n-tiles (<= ~2 MB of
// W per group): inside a group m-major with n fastest, group after group.  Placement affects speed only.
__device__ __forceinline__ void tile_coords(int t, int tiles_m, int tiles_n, int ngrp, int& tm, int& tn) {
  if (ngrp <= 0 || ngrp >= tiles_n) { tm = t / tiles_n; tn = t - tm * tiles_n; return; }
  const int per = tiles_m * ngrp;
  const int nfull = tiles_n / ngrp;  // full groups; a smaller last group may follow
  const int g = t / per;
  if (g < nfull) {
    const int r = t - g * per;
    tm = r / ngrp;
    tn = g * ngrp + (r - tm * ngrp);
  } else {
    const int last = tiles_n - nfull * ngrp;
    const int r = t - nfull * per;
    tm = r / last;
    tn = nfull * ngrp + (r - tm * last);
  }
}
// host: n-tiles per group so that a group's weight slice (bn x K bf16 per tile) stays around 2 MB, groups equalised
static inline int pick_ngrp(int tiles_n, int bn, int K) {
#if defined(APLA_ABL_NGRP)  // diagnostic build (tools/build_ablations.sh): n-tiles per column group from the environment
  if (const char* e = getenv("APLA_NGRP")) {
    const int v = atoi(e);
    if (v >= 0) return v >= tiles_n ? 0 : v;
  }
#endif
  const long tile_bytes = (long)bn * K * 2;
  if ((long)tiles_n * tile_bytes <= (5L << 19)) return 0;  // <= 2.5 MB: plain walk
  long per = (2L << 20) / tile_bytes;
  if (per < 1) per = 1;
  const int groups = (int)((tiles_n + per - 1) / per);
  return (tiles_n + groups - 1) / groups;
}

// GELU (exact erf form) and its derivative from ONE exp2 and ONE rcp per element: Phi(a) via Abramowitz-Stegun 7.1.26
// (|erf error| < 1.5e-7, far below the bf16 output rounding) sharing E = exp(-a^2/2) with the Gaussian term of gelu'.
// Written on four values at a time with explicit elementwise FMAs: hipcc then emits v_pk_mul_f32 / v_pk_fma_f32 for everything
// but the two transcendentals, |a|, the compare and the select — 11.5 vector instructions per element instead of the ~24 of the
// scalar formulation (the epilogue's VALU work competes for issue slots with the co-resident workgroup's MFMAs).  The 0.5 of
// 0.5 * erfc is folded into the polynomial's coefficients (exact: a power of two).  Every GELU epilogue calls these two
// functions, so all kernel schedules give the same bits.
__device__ __forceinline__ f32x4 vrcp4(f32x4 v) {
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = __builtin_amdgcn_rcpf(v[e]);
  return o;
}
__device__ __forceinline__ f32x4 vexp2_4(f32x4 v) {
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = __builtin_amdgcn_exp2f(v[e]);
  return o;
}
__device__ __forceinline__ f32x4 half_erfc_poly4(f32x4 a) {   // 0.5 * poly(t) with t = 1 / (1 + p |a| / sqrt 2): q = this * E
  const f32x4 x = __builtin_elementwise_abs(a) * 0.70710678118654752f;
  const f32x4 t = vrcp4(__builtin_elementwise_fma(f32x4(0.3275911f), x, f32x4(1.0f)));
  f32x4 p = __builtin_elementwise_fma(t, f32x4(0.5f * 1.061405429f), f32x4(0.5f * -1.453152027f));
  p = __builtin_elementwise_fma(p, t, f32x4(0.5f * 1.421413741f));
  p = __builtin_elementwise_fma(p, t, f32x4(0.5f * -0.284496736f));
  p = __builtin_elementwise_fma(p, t, f32x4(0.5f * 0.254829592f));
  return p * t;
}
__device__ __forceinline__ void gelu_and_grad4(f32x4 a, f32x4& h, f32x4& g) {
  const f32x4 p = half_erfc_poly4(a);
  const f32x4 E = vexp2_4((a * a) * -0.72134752044448170f);  // exp(-a^2/2)
  const f32x4 q = p * E;                                       // = 1 - Phi(|a|)
  const f32x4 phi = (a >= 0.f) ? __builtin_elementwise_fma(-p, E, f32x4(1.0f)) : q;
  h = a * phi;
  g = __builtin_elementwise_fma(a * E, f32x4(0.3989422804014327f), phi);
}
// GELU alone (the no-grad forward): the same Phi as gelu_and_grad4, without the Gaussian term of the derivative
__device__ __forceinline__ f32x4 gelu_only4(f32x4 a) {
  const f32x4 p = half_erfc_poly4(a);
  const f32x4 E = vexp2_4((a * a) * -0.72134752044448170f);
  const f32x4 q = p * E;
  return a * ((a >= 0.f) ? __builtin_elementwise_fma(-p, E, f32x4(1.0f)) : q);
}
__device__ __forceinline__ void gelu_and_grad(float a, float& h, float& g) {
  f32x4 hv, gv;
  gelu_and_grad4(f32x4{a, a, a, a}, hv, gv);
  h = hv[0];
  g = gv[0];
}
__device__ __forceinline__ float gelu_only(float a) { return gelu_only4(f32x4{a, a, a, a})[0]; }
__device__ __forceinline__ float sigmoid_f(float a) { return 1.0f / (1.0f + __expf(-a)); }

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <typename T> struct Vec8IO;
template <> struct Vec8IO<float> {
  static __device__ __forceinline__ f32x4 pack(f32x4 lo, f32x4 hi) { return lo + hi; }
  static __device__ __forceinline__ void store(float* p, f32x4 lo, f32x4 hi) { *(f32x4*)p = lo; *(f32x4*)(p + 4) = hi; }
};
template <> struct Vec8IO<bf16> {
  static __device__ __forceinline__ bf16x8 pack(f32x4 lo, f32x4 hi) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = (bf16)lo[e]; o[4 + e] = (bf16)hi[e]; }
    return o;
  }
  static __device__ __forceinline__ void store(bf16* p, f32x4 lo, f32x4 hi) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = (bf16)lo[e]; o[4 + e] = (bf16)hi[e]; }
    *(bf16x8*)p = o;
  }
};

// 8 consecutive elements of an epilogue operand held in registers filled by inline-asm loads (invisible to hipcc's
// waitcnt pass; see persist_epilogue)
template <typename T> struct AuxRegs;
template <> struct AuxRegs<float> {
  f32x4 lo, hi;
  __device__ __forceinline__ f32x4 get_lo() const { return lo; }
  __device__ __forceinline__ f32x4 get_hi() const { return hi; }
};
template <> struct AuxRegs<bf16> {
  f32x4 lo;  // 8 bf16 in 4 VGPRs
  __device__ __forceinline__ f32x4 get_lo() const {
    const bf16x8 t = __builtin_bit_cast(bf16x8, lo);
    return f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
  }
  __device__ __forceinline__ f32x4 get_hi() const {
    const bf16x8 t = __builtin_bit_cast(bf16x8, lo);
    return f32x4{(float)t[4], (float)t[5], (float)t[6], (float)t[7]};
  }
};
template <typename T> __device__ __forceinline__ void asm_wait_pin(AuxRegs<T>& a);
template <> __device__ __forceinline__ void asm_wait_pin<float>(AuxRegs<float>& a) { asm volatile("" : "+v"(a.lo), "+v"(a.hi)); }
template <> __device__ __forceinline__ void asm_wait_pin<bf16>(AuxRegs<bf16>& a) { asm volatile("" : "+v"(a.lo)); }

// Output-column ownership in the persistent kernel: the W tile is staged into LDS in "MFMA order" — LDS row
// wn*64 + j*16 + r holds W row wn*64 + 32*(j>>1) + 8*(r>>2) + 4*(j&1) + (r&3) — so that after the MFMAs a lane owns
// EIGHT consecutive output columns per n-tile pair (j = 2u, 2u+1): columns 32u + 8*fq + 0..3 from acc[i][2u] and
// + 4..7 from acc[i][2u+1].  bf16 outputs then leave as one 16-byte store per (row, pair) instead of two 8-byte ones
// (the epilogue is store-issue bound: half the instructions, 64 contiguous bytes per row per instruction).
__device__ __forceinline__ int w_row_of_lds_row(int l) {  // any LDS row index; mapping is per 64-row block
  const int r = l & 15, j = (l >> 4) & 3;
  return (l & ~63) + 32 * (j >> 1) + 8 * (r >> 2) + 4 * (j & 1) + (r & 3);
}

template <int EPI, typename OutT, int MI> struct EpiStores {
  // store instructions per wave per (full) tile after the prefetch was issued
  static constexpr int PER_PAIR = (EPI == APLA_EPI_GELU) ? 2 : (EPI == APLA_EPI_SWIGLU) ? 2
                                  : (EPI == APLA_EPI_SWIGLU_BWD) ? 2 : (sizeof(OutT) == 4 ? 2 : 1);
  static constexpr int N = MI * 2 * PER_PAIR;
};

// 2 x (8 consecutive elements) of one row, 32 elements apart, loaded by inline asm (not tracked by hipcc's waitcnt pass)
__device__ __forceinline__ void asm_load_row2(AuxRegs<float>& a0, AuxRegs<float>& a1, const float* p) {
  asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\t"
               "global_load_dwordx4 %2, %4, off offset:128\n\tglobal_load_dwordx4 %3, %4, off offset:144"
               : "=&v"(a0.lo), "=&v"(a0.hi), "=&v"(a1.lo), "=&v"(a1.hi) : "v"(p) : "memory");
}
__device__ __forceinline__ void asm_load_2ptr(AuxRegs<bf16>& a0, AuxRegs<bf16>& a1, const bf16* p0, const bf16* p1) {
  asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %3, off"
               : "=&v"(a0.lo), "=&v"(a1.lo) : "v"(p0), "v"(p1) : "memory");
}
__device__ __forceinline__ void asm_load_2ptr(AuxRegs<float>&, AuxRegs<float>&, const float*, const float*) {}  // (no fp32 image)
__device__ __forceinline__ void asm_load_row2(AuxRegs<bf16>& a0, AuxRegs<bf16>& a1, const bf16* p) {
  asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:64"
               : "=&v"(a0.lo), "=&v"(a1.lo) : "v"(p) : "memory");
}
template <int EPI, typename OutT, int MI, bool DROP = false>
__device__ __forceinline__ void persist_epilogue(const GemmParams& p, f32x4 (&acc)[MI][4], const float* bias_lds,
                                                 int m0, int n0, int wm, int wn, int lane) {
  const int frow = lane & 15, fq = lane >> 4;
  const int ncol = wn * 64 + fq * 8;  // + 32*u : first of the 8 columns this lane owns in pair u (tile-local)
  // element (m, n) of the 16-bit output C: row-major, or its K-panel image (the 8 columns a lane owns never straddle a panel,
  // and the 16 rows of a fragment are 1 KB contiguous there)
  const bool c_img = (p.w_panel & 4) != 0;
  auto c_at = [&](int m, int n) -> bf16* {
    return (bf16*)p.C + (c_img ? ((size_t)(n >> 5) * p.M + m) * 32 + (n & 31) : (size_t)m * p.ldc + n);
  };
  int mrow[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) mrow[i] = m0 + wm * (MI * 16) + i * 16 + frow;
  if (p.bias != nullptr) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const f32x4 blo = *(const f32x4*)(bias_lds + ncol + 32 * u), bhi = *(const f32x4*)(bias_lds + ncol + 32 * u + 4);
#pragma unroll
      for (int i = 0; i < MI; ++i) { acc[i][2 * u] += blo; acc[i][2 * u + 1] += bhi; }
    }
  }
  if constexpr (EPI == APLA_EPI_RESIDUAL || EPI == APLA_EPI_MUL) {
    // Epilogue operands through inline-asm loads: an ordinary global load next to in-flight LDS-DMA makes hipcc's waitcnt
    // pass drain vmcnt(0) inside the K loop (de-pipelining it); asm loads are invisible to that pass, so we wait for them
    // ourselves: all loads are issued back to back, then ONE s_waitcnt, then every destination is pinned behind it.
    using AuxT = typename std::conditional<EPI == APLA_EPI_MUL, bf16, OutT>::type;
    AuxRegs<AuxT> aux[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int mr = mrow[i] < p.M ? mrow[i] : p.M - 1;
      if (EPI == APLA_EPI_MUL && (p.w_panel & 8)) {   // image: the two 8-column pieces of a row sit in consecutive panels
        const AuxT* q = (const AuxT*)p.aux_in + ((size_t)((n0 + ncol) >> 5) * p.M + mr) * 32 + ((n0 + ncol) & 31);
        asm_load_2ptr(aux[i][0], aux[i][1], q, q + (size_t)p.M * 32);
      } else {
        asm_load_row2(aux[i][0], aux[i][1], (const AuxT*)p.aux_in + (size_t)mr * p.ld_aux_in + n0 + ncol);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < MI; ++i) { asm_wait_pin<AuxT>(aux[i][0]); asm_wait_pin<AuxT>(aux[i][1]); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const f32x4 alo = aux[i][u].get_lo(), ahi = aux[i][u].get_hi();
        if constexpr (EPI == APLA_EPI_MUL) { acc[i][2 * u] *= alo; acc[i][2 * u + 1] *= ahi; }
        else { acc[i][2 * u] += alo; acc[i][2 * u + 1] += ahi; }
      }
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = mrow[i];
    if (m >= p.M) continue;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int n = n0 + ncol + 32 * u;
      const f32x4 lo = acc[i][2 * u], hi = acc[i][2 * u + 1];
      if constexpr (EPI == APLA_EPI_STORE || EPI == APLA_EPI_RESIDUAL) {
        Vec8IO<OutT>::store((OutT*)p.C + (size_t)m * p.ldc + n, lo, hi);
      } else if constexpr (EPI == APLA_EPI_MUL) {
        Vec8IO<bf16>::store(c_at(m, n), lo, hi);
      } else if constexpr (EPI == APLA_EPI_GELU) {
        f32x4 hl, hh, gl, gh;
#if defined(APLA_ABL_NOGELU)   // diagnostic build: both outputs are the accumulators (the two stores without the GELU arithmetic)
        hl = lo; hh = hi; gl = lo; gh = hi;
#else
        gelu_and_grad4(lo, hl, gl);
        gelu_and_grad4(hi, hh, gh);
#endif
        if constexpr (DROP) {
          // Mlp.drop after the activation (vit.py:164-165): h through the mask, and GELU' through the SAME mask — dfc2's epilogue
          // (dh * GELU') then needs no change.  Element index = row-major position of (m, n) in [M, N], whatever layout the outputs have.
          unsigned wa[4], wb[4];
          const unsigned long long blk = ((unsigned long long)m * (unsigned long long)p.N + (unsigned long long)n) >> 2;
          drop_words4(p.drop, blk, wa);
          drop_words4(p.drop, blk + 1, wb);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float ka = wa[e] >= p.drop.threshold ? p.drop.inv_keep : 0.f, kb = wb[e] >= p.drop.threshold ? p.drop.inv_keep : 0.f;
            hl[e] *= ka; gl[e] *= ka; hh[e] *= kb; gh[e] *= kb;
          }
        }
        Vec8IO<bf16>::store(c_at(m, n), hl, hh);
        Vec8IO<bf16>::store((bf16*)p.aux_out + ((p.w_panel & 8) ? ((size_t)(n >> 5) * p.M + m) * 32 + (n & 31) : (size_t)m * p.ld_aux_out + n), gl, gh);
      } else if constexpr (EPI == APLA_EPI_GELU_FWD) {
        const f32x4 hl = gelu_only4(lo), hh = gelu_only4(hi);
        Vec8IO<bf16>::store(c_at(m, n), hl, hh);
      } else if constexpr (EPI == APLA_EPI_SWIGLU) {
        // columns come in (x1_i, x2_i) pairs: 8 columns = 4 hidden units
        Vec8IO<bf16>::store((bf16*)p.aux_out + (size_t)m * p.ld_aux_out + n, lo, hi);
        bf16x4 h;
        h[0] = (bf16)(lo[0] * sigmoid_f(lo[0]) * lo[1]);
        h[1] = (bf16)(lo[2] * sigmoid_f(lo[2]) * lo[3]);
        h[2] = (bf16)(hi[0] * sigmoid_f(hi[0]) * hi[1]);
        h[3] = (bf16)(hi[2] * sigmoid_f(hi[2]) * hi[3]);
        *(bf16x4*)c_at(m, n >> 1) = h;    // 4 hidden units (n / 2 is a multiple of 4: they never straddle a 32-wide panel)
      } else if constexpr (EPI == APLA_EPI_SWIGLU_BWD) {
        // dh for hidden units n..n+7; saved x12 interleaved at columns 2n..2n+15
        const bf16* xs = (const bf16*)p.aux_in + (size_t)m * p.ld_aux_in + 2 * n;
        const bf16x8 xa = *(const bf16x8*)xs, xb = *(const bf16x8*)(xs + 8);
        bf16x8 oa, ob;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x1 = (float)xa[2 * e], x2 = (float)xa[2 * e + 1], sg = sigmoid_f(x1);
          oa[2 * e] = (bf16)(lo[e] * x2 * sg * (1.0f + x1 * (1.0f - sg)));
          oa[2 * e + 1] = (bf16)(lo[e] * x1 * sg);
          x1 = (float)xb[2 * e]; x2 = (float)xb[2 * e + 1]; sg = sigmoid_f(x1);
          ob[2 * e] = (bf16)(hi[e] * x2 * sg * (1.0f + x1 * (1.0f - sg)));
          ob[2 * e + 1] = (bf16)(hi[e] * x1 * sg);
        }
        *(bf16x8*)c_at(m, 2 * n) = oa;        // 2 n is a multiple of 16: both 8-element pieces stay inside one panel
        *(bf16x8*)c_at(m, 2 * n + 8) = ob;
      }
    }
  }
}


// ---- 80x128 wave tile (5 x 8 MFMA tiles, 4 column pairs): epilogue of gemm_pp2.hip ----
template <typename T> __device__ __forceinline__ void asm_load_row4(AuxRegs<T> (&a)[4], const T* p);
template <> __device__ __forceinline__ void asm_load_row4<float>(AuxRegs<float> (&a)[4], const float* p) {
  asm volatile(
      "global_load_dwordx4 %0, %8, off\n\tglobal_load_dwordx4 %1, %8, off offset:16\n\t"
      "global_load_dwordx4 %2, %8, off offset:128\n\tglobal_load_dwordx4 %3, %8, off offset:144\n\t"
      "global_load_dwordx4 %4, %8, off offset:256\n\tglobal_load_dwordx4 %5, %8, off offset:272\n\t"
      "global_load_dwordx4 %6, %8, off offset:384\n\tglobal_load_dwordx4 %7, %8, off offset:400"
      : "=&v"(a[0].lo), "=&v"(a[0].hi), "=&v"(a[1].lo), "=&v"(a[1].hi), "=&v"(a[2].lo), "=&v"(a[2].hi), "=&v"(a[3].lo), "=&v"(a[3].hi)
      : "v"(p) : "memory");
}
template <> __device__ __forceinline__ void asm_load_row4<bf16>(AuxRegs<bf16> (&a)[4], const bf16* p) {
  asm volatile(
      "global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:64\n\t"
      "global_load_dwordx4 %2, %4, off offset:128\n\tglobal_load_dwordx4 %3, %4, off offset:192"
      : "=&v"(a[0].lo), "=&v"(a[1].lo), "=&v"(a[2].lo), "=&v"(a[3].lo) : "v"(p) : "memory");
}

template <int EPI, typename OutT> struct WideEpi {
  using AuxT = typename std::conditional<EPI == APLA_EPI_MUL, bf16, OutT>::type;
  static constexpr bool HAS_AUX = (EPI == APLA_EPI_RESIDUAL || EPI == APLA_EPI_MUL);
  static constexpr int L = HAS_AUX ? (sizeof(AuxT) == 4 ? 8 : 4) : 0;                       // aux loads per row
  static constexpr int S = (EPI == APLA_EPI_GELU) ? 8 : ((EPI == APLA_EPI_MUL) ? 4 : (sizeof(OutT) == 4 ? 8 : 4));  // stores per row
  static constexpr int NST = 5 * S;
  static constexpr bool LINES = !HAS_AUX && sizeof(OutT) == 2 &&
                                (EPI == APLA_EPI_STORE || EPI == APLA_EPI_GELU || EPI == APLA_EPI_GELU_FWD);  // whole-line stores via LDS
};

// GELU and GELU' of 8 accumulator values, packed to bf16.  Four values at a time, each group's packed results pinned by an
// (empty) volatile asm: left alone hipcc computes all h first and keeps phi and E of every element for the g pass, and
// the epilogue — which starts with all 160 accumulators live — then spills lane constants that the K loop reloads with
// vmcnt(0) in front of every LDS-DMA.
__device__ __forceinline__ void gelu8(f32x4 lo, f32x4 hi, bf16x8& h, bf16x8& g) {
#if defined(APLA_ABL_NOGELU)  // diagnostic build: both outputs are the packed accumulators (store traffic without the VALU work)
  h = Vec8IO<bf16>::pack(lo, hi);
  g = h;
  return;
#endif
  typedef bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  unsigned hp[4], gp[4];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const f32x4 a = half ? hi : lo;
    f32x4 x, y;
    gelu_and_grad4(a, x, y);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      bf16x2_t hh, gg;
      hh[0] = (bf16)x[2 * e]; hh[1] = (bf16)x[2 * e + 1]; gg[0] = (bf16)y[2 * e]; gg[1] = (bf16)y[2 * e + 1];
      hp[2 * half + e] = __builtin_bit_cast(unsigned, hh);
      gp[2 * half + e] = __builtin_bit_cast(unsigned, gg);
    }
  }
  asm volatile("" : "+v"(hp[0]), "+v"(gp[0]), "+v"(hp[1]), "+v"(gp[1]), "+v"(hp[2]), "+v"(gp[2]), "+v"(hp[3]), "+v"(gp[3]));
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  h = __builtin_bit_cast(bf16x8, u32x4_t{hp[0], hp[1], hp[2], hp[3]});
  g = __builtin_bit_cast(bf16x8, u32x4_t{gp[0], gp[1], gp[2], gp[3]});
}

// GELU alone of 8 accumulator values, packed to bf16 (same structure and pinning as gelu8)
__device__ __forceinline__ void gelu8_fwd(f32x4 lo, f32x4 hi, bf16x8& h) {
  typedef bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  unsigned hp[4];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const f32x4 a = half ? hi : lo;
    const f32x4 x = gelu_only4(a);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      bf16x2_t hh;
      hh[0] = (bf16)x[2 * e]; hh[1] = (bf16)x[2 * e + 1];
      hp[2 * half + e] = __builtin_bit_cast(unsigned, hh);
    }
    asm volatile("" : "+v"(hp[2 * half]), "+v"(hp[2 * half + 1]));
  }
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  h = __builtin_bit_cast(bf16x8, u32x4_t{hp[0], hp[1], hp[2], hp[3]});
}

// bf16 outputs without a second operand go through a per-wave 2 KB LDS buffer `tbuf` (16 rows x 128 B, 16-B chunk c of row
// r at chunk c ^ (r >> 1): conflict-free both ways) so that a store instruction writes 8 rows x 128 B = whole cache lines.
// In the MFMA layout a wave instruction covers 16 rows x 64 B, and a CU then stores at ~23 GB/s whatever the rest of the
// chip does; with whole lines one CU reaches ~80 GB/s and only the chip-wide HBM rate bounds it (tools/store_probe.hip).
template <int EPI, typename OutT, int MI = 5>   // MI: 16-row fragments per wave (wave tile 16 MI x 128)
__device__ __forceinline__ void wide_epilogue(const GemmParams& p, f32x4 (&acc)[MI][8], const float* bias_lds, char* tbuf,
                                              int m0, int n0, int wm, int wn, int lane, bool full_tile) {
  using E = WideEpi<EPI, OutT>;
  using AuxT = typename E::AuxT;
  // every address below derives from this opaque copy of the lane id, so hipcc cannot hoist the (loop-invariant) address
  // arithmetic out of the persistent loop, where it would stay live across the MFMA phases and push other values to scratch
  asm volatile("" : "+v"(lane));
  const int frow = lane & 15, fq = lane >> 4;
  const int ncol = wn * 128 + fq * 8;  // + 32*u
  if (p.bias != nullptr) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const f32x4 blo = *(const f32x4*)(bias_lds + ncol + 32 * u), bhi = *(const f32x4*)(bias_lds + ncol + 32 * u + 4);
#pragma unroll
      for (int i = 0; i < MI; ++i) { acc[i][2 * u] += blo; acc[i][2 * u + 1] += bhi; }
    }
  }
  if constexpr (E::LINES) {
    char* wr0 = tbuf + frow * 128 + ((fq ^ (frow >> 1)) << 4);         // logical chunk fq
    char* wr1 = tbuf + frow * 128 + (((4 + fq) ^ (frow >> 1)) << 4);   // logical chunk 4 + fq
    const int rrow = lane >> 3, rc = lane & 7;
    const char* rd0 = tbuf + rrow * 128 + ((rc ^ (rrow >> 1)) << 4);
    const char* rd1 = tbuf + (rrow + 8) * 128 + ((rc ^ ((rrow + 8) >> 1)) << 4);
    const int mbase = m0 + wm * (16 * MI) + rrow;
    const size_t cbase = (size_t)n0 + wn * 128 + rc * 8;
    // One buffer, software-pipelined: LDS serves a wave's operations in order, so the writes of step k+1 may be issued
    // right behind the reads of step k; the global stores of step k go out while step k+1's round trip is in flight.
    // Full tiles take a branch-free path (per-row predication splits the stream into basic blocks hipcc cannot overlap).
    auto stage = [&](bf16x8 c0, bf16x8 c1, bf16x8& r0, bf16x8& r1) {
      *(bf16x8*)wr0 = c0;
      *(bf16x8*)wr1 = c1;
      r0 = *(const bf16x8*)rd0;
      r1 = *(const bf16x8*)rd1;
    };
    // `img`: dst is the K-panel image [N/32][M][32] of the output (GemmParams::w_panel bits 2 / 3): a lane's 8 columns never
    // straddle a panel, 8 consecutive rows are 512 contiguous bytes per panel: a store instruction writes two such runs
    auto commit = [&](auto FULL, bf16* dst, int ld, bool img, int i, int h, bf16x8 r0, bf16x8 r1) {
      const int ma = mbase + i * 16, mb = ma + 8;
      const size_t col = cbase + h * 64;
      auto at = [&](int m) -> bf16* { return img ? dst + ((col >> 5) * (size_t)p.M + m) * 32 + (col & 31) : dst + (size_t)m * ld + col; };
#if defined(APLA_ABL_NOSTORE)
      asm volatile("" :: "v"(r0), "v"(r1));
#else
      if (FULL.value || ma < p.M) *(bf16x8*)at(ma) = r0;
      if (FULL.value || mb < p.M) *(bf16x8*)at(mb) = r1;
#endif
    };
    const bool c_img = (p.w_panel & 4) != 0, x_img = (p.w_panel & 8) != 0;
    auto run = [&](auto FULL) {
      if constexpr (EPI == APLA_EPI_GELU_FWD) {   // like GELU below, one output: no cross-step registers
#pragma unroll
        for (int k = 0; k < 2 * MI; ++k) {
          const int i = k >> 1, h = k & 1;
          bf16x8 hc0, hc1, n0_, n1_;
          gelu8_fwd(acc[i][4 * h], acc[i][4 * h + 1], hc0);
          gelu8_fwd(acc[i][4 * h + 2], acc[i][4 * h + 3], hc1);
          stage(hc0, hc1, n0_, n1_);
          commit(FULL, (bf16*)p.C, p.ldc, c_img, i, h, n0_, n1_);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else if constexpr (EPI == APLA_EPI_STORE) {
        bf16x8 p0, p1;  // step k-1, read back and waiting to be stored
#pragma unroll
        for (int k = 0; k <= 2 * MI; ++k) {
          const int i = k >> 1, h = k & 1;
          bf16x8 n0_, n1_;
          if (k < 2 * MI)
            stage(Vec8IO<bf16>::pack(acc[i][4 * h], acc[i][4 * h + 1]), Vec8IO<bf16>::pack(acc[i][4 * h + 2], acc[i][4 * h + 3]), n0_, n1_);
          if (k > 0) commit(FULL, (bf16*)p.C, p.ldc, false, (k - 1) >> 1, (k - 1) & 1, p0, p1);
          p0 = n0_; p1 = n1_;
        }
      } else {  // GELU: the two outputs of a step overlap each other; no cross-step registers (the kernel has none to spare)
#pragma unroll
        for (int k = 0; k < 2 * MI; ++k) {
          const int i = k >> 1, h = k & 1;
          bf16x8 hc0, gc0, hc1, gc1, n0_, n1_, m0_, m1_;
          gelu8(acc[i][4 * h], acc[i][4 * h + 1], hc0, gc0);
          gelu8(acc[i][4 * h + 2], acc[i][4 * h + 3], hc1, gc1);
          stage(hc0, hc1, n0_, n1_);
          stage(gc0, gc1, m0_, m1_);
          commit(FULL, (bf16*)p.C, p.ldc, c_img, i, h, n0_, n1_);
          commit(FULL, (bf16*)p.aux_out, p.ld_aux_out, x_img, i, h, m0_, m1_);
          __builtin_amdgcn_sched_barrier(0);  // keep the steps apart: interleaving them costs registers, not time
        }
      }
    };
    if constexpr (EPI == APLA_EPI_STORE) {
      if (full_tile) run(std::true_type{}); else run(std::false_type{});
    } else {
      run(std::false_type{});  // one copy only: a second unrolled GELU epilogue costs registers the kernel does not have
    }
    return;
  }
  AuxRegs<AuxT> aux[2][4];  // two rows in flight
  auto row_ptr = [&](int i) {
    int mr = m0 + wm * (16 * MI) + i * 16 + frow;
    mr = mr < p.M ? mr : p.M - 1;
    return (const AuxT*)p.aux_in + (size_t)mr * p.ld_aux_in + n0 + ncol;
  };
  if constexpr (E::HAS_AUX) {
    asm_load_row4<AuxT>(aux[0], row_ptr(0));
    asm_load_row4<AuxT>(aux[1], row_ptr(1));
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    if constexpr (E::HAS_AUX) {
      // operations issued after row i's loads: row i+1's loads (if any) and, from i >= 1, the stores of row i-1
      constexpr int LN = E::L, SN = E::S;
      if (i == 0) wait_vmcnt<LN>();
      else if (i < MI - 1) { if (full_tile) wait_vmcnt<LN + SN>(); else wait_vmcnt<LN>(); }
      else { if (full_tile) wait_vmcnt<SN>(); else wait_vmcnt<0>(); }
#pragma unroll
      for (int u = 0; u < 4; ++u) asm_wait_pin<AuxT>(aux[i & 1][u]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const f32x4 alo = aux[i & 1][u].get_lo(), ahi = aux[i & 1][u].get_hi();
        if constexpr (EPI == APLA_EPI_MUL) { acc[i][2 * u] *= alo; acc[i][2 * u + 1] *= ahi; }
        else { acc[i][2 * u] += alo; acc[i][2 * u + 1] += ahi; }
      }
    }
    const int m = m0 + wm * (16 * MI) + i * 16 + frow;
    if (m < p.M) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int n = n0 + ncol + 32 * u;
        const f32x4 lo = acc[i][2 * u], hi = acc[i][2 * u + 1];
        if constexpr (EPI == APLA_EPI_STORE || EPI == APLA_EPI_RESIDUAL) {
          Vec8IO<OutT>::store((OutT*)p.C + (size_t)m * p.ldc + n, lo, hi);
        } else if constexpr (EPI == APLA_EPI_MUL) {
          Vec8IO<bf16>::store((bf16*)p.C + (size_t)m * p.ldc + n, lo, hi);
        } else if constexpr (EPI == APLA_EPI_GELU) {
          f32x4 hl, hh, gl, gh;
          gelu_and_grad4(lo, hl, gl);
          gelu_and_grad4(hi, hh, gh);
          Vec8IO<bf16>::store((bf16*)p.C + (size_t)m * p.ldc + n, hl, hh);
          Vec8IO<bf16>::store((bf16*)p.aux_out + (size_t)m * p.ld_aux_out + n, gl, gh);
        }
      }
    }
    asm volatile("" ::: "memory");
    if constexpr (E::HAS_AUX) {
      if (i + 2 < MI) asm_load_row4<AuxT>(aux[i & 1], row_ptr(i + 2));
    }
  }
}

