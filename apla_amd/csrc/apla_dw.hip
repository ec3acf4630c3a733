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
#include <stdlib.h>

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
  bf16x4 lo = DS_READ_TR16_B64_H16(lds + a0);
  bf16x4 hi = DS_READ_TR16_B64_H16(lds + a1);
  bf16x8 out;
  out[0] = lo[0]; out[1] = lo[1]; out[2] = lo[2]; out[3] = lo[3];
  out[4] = hi[0]; out[5] = hi[1]; out[6] = hi[2]; out[7] = hi[3];
  return out;
}

// Transposed fragments by inline asm with base + immediate addressing.  (a) Next to in-flight LDS-DMA hipcc puts
// s_waitcnt vmcnt(0) in front of every ds_read_b64_tr_b16 it can see (draining the ring each step); the asm form is waited
// for by hand (lgkmcnt).  (b) The swizzle term of tile_off does not depend on the 16-row step ks (rows move by 16) nor on
// the sub-tile, so a lane needs one address per (column half, row half) and everything else is an immediate: left to the
// compiler, the ~110 fragment addresses of a step were recomputed every step (~1000 VALU ops, 2 us per step).
__device__ __forceinline__ unsigned tr_base(const char* tile, int c0, int half, int lane) {
  const int g = lane >> 4, i = lane & 15;
  const int col = c0 + 16 * (g & 1) + 4 * (i & 3);
  const int r = 4 * (g >> 1) + (i >> 2) + 8 * half;
  return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)(tile + tile_off(r, col >> 3) + ((col & 4) << 1));
}
template <int OFF> __device__ __forceinline__ bf16x4 tr_read(unsigned addr) {
  bf16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ bf16x8 join8(bf16x4 lo, bf16x4 hi) {
  bf16x8 o;
  o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
  o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
  return o;
}

template <int I> struct IntC { static constexpr int value = I; };
template <int B, int E, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) { f(IntC<B>{}); static_for<B + 1, E>(f); }
}

constexpr int TJ = 64, TK = 128, TM = 64;  // j granule 64; output tile (64*NJ) x 128 (k); 64 token rows per step

// One workgroup: NJ*64 trainable rows (j) x 128 input features (k) over one slab of tokens.  The four waves split k
// (32 columns each) and keep all NJ*64 rows: with r = 192 = 3*64 the activation x is then read ONCE in total and dyg
// D/128 times (L2 hits), instead of r/64 and D/128 times.
// Operands stream global -> LDS by LDS-DMA (16 B per lane, 1 KB pieces of 8 rows x 128 B; the XOR swizzle of tile_off is
// applied on the SOURCE column because the LDS side of a DMA is lane-linear) through a 3-stage ring: two 64-token steps
// are in flight while one is multiplied, one s_barrier per step, counted vmcnt (never 0 inside the loop).
// Several layers' gradients can share ONE launch (apla_proj_dw_batched): a unit = (layer, token slab), U = nb * S units.  With
// nb layers the token axis is cut into 1/nb as many slabs for the same number of workgroups, so a workgroup's pipeline is nb
// times longer (prologue, the r x 128 partial tile it writes and the reduce pass are paid once per unit, not once per layer
// and slab) and the slab partials shrink from S x r x D to S/nb x r x D per layer.
// grid = 8 * ceil(U/8) * tiles_k * j_groups (1-D).  partial layout: [U][r][D] fp32, then [U][r] fp32 for the bias sums.
#define DW_LDSP(p) ((__attribute__((address_space(3))) void*)(p))
#define DW_GLBP(p) ((const __attribute__((address_space(1))) void*)(p))
constexpr int DW_MAX_BATCH = 8;
// Ring depth of proj_dw_partial_kernel: 3 stages, one workgroup per CU — or, up to NJ = 2 and from 16 output tiles (r = 256 at D = 1024,
// the 2048-wide layers of the DINO head), 2 stages and TWO workgroups per CU (see the kernel).  With few output tiles the token axis
// is cut into many slabs and doubling them doubles the fp32 partials: r = 128 at D = 768 (6 tiles) lost 18-30 % that way.  (A four-stage
// ring with one workgroup changed nothing in config 3's step; its counters show the batched launches fetching 2.4 x the algorithmic
// bytes — the 16 tiles of a unit drift apart over 200-500 steps and lose each other in the L2 — at the same time as the
// one-layer launches at 1.0 x: the kernel is bound by neither; profiles/r06_h_dispatch_checks.md.)
inline int dw_group(int r) { return r % 192 == 0 ? 3 : (r % 128 == 0 ? 2 : 1); }  // 64-row granules per workgroup
inline int dw_ring(int r, int D) {
#if defined(APLA_ABL_DWRING3)   // diagnostic build: round 5's form at every width
  return 3;
#else
  return (dw_group(r) <= 2 && (r / (TJ * dw_group(r))) * (D / TK) >= 16) ? 2 : 3;
#endif
}
struct DwIn { const bf16* dyg[DW_MAX_BATCH]; const bf16* x[DW_MAX_BATCH]; };
struct DwOut { float* dW[DW_MAX_BATCH]; float* db[DW_MAX_BATCH]; const float* row_scale[DW_MAX_BATCH]; };
template <int N> __device__ __forceinline__ void dw_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int NJ, int NS>
__global__ __launch_bounds__(256) void proj_dw_partial_kernel(DwIn in, int nb, int ldx, float* __restrict__ partial, int M, int r,
                                                              int D, int rows_per_slab, int S) {
  // Ring depth.  A wave of this kernel issues its own LDS-DMA (~100 blocking cycles per 1 KB piece, 8-10 pieces per step) next to 16-24
  // products and 40-56 transposed LDS reads, one wave per SIMD: the step is ISSUE-bound, not HBM-bound (round 6: 1.2-2.0 TB/s at
  // r = 256 / 128).  Up to NJ = 2 a two-stage ring (64 KB + the 66 KB epilogue tile) lets TWO workgroups share a CU, so that one's DMA
  // issue and barrier waits run under the other's products (NS = 2; chosen from 16 output tiles: dw_ring); NJ = 3 (r = 192: 40 KB
  // stages) keeps three stages and one workgroup.
  constexpr int STG = (NJ + 2) * 8192, PW = (NJ + 2) * 2;  // stage bytes; pieces per wave per step (NS: ring depth)
  constexpr int EPI_BYTES = 64 * NJ * 132 * 4;                               // the fp32 output tile of the epilogue
  __shared__ __attribute__((aligned(16))) char smem[NS * STG > EPI_BYTES ? NS * STG : EPI_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, h2 = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Workgroup ids are dealt round-robin to the 8 XCDs: all (tj, tk) tiles of one unit (layer, token slab) get ids of the same
  // residue mod 8, so the slab's dyg and x rows are fetched from HBM once and shared through that XCD's L2.
  const int tiles_k = D / TK, tiles = tiles_k * (r / (TJ * NJ));
  const int xcd = blockIdx.x & 7, n = blockIdx.x >> 3;
  const int U = nb * S;
  // (a problem with hundreds of output tiles and a single slab — r = 65 536 prototypes x a few thousand token rows — has nothing to
  // share through an L2: its tiles are dealt to the XCDs one by one; grid = tiles then)
  const bool flat = U == 1;
  const int tile = flat ? blockIdx.x : n % tiles, unit = flat ? 0 : (n / tiles) * 8 + xcd;
  if (unit >= U || tile >= tiles) return;
  const int layer = unit / S, slab = unit - layer * S;
  const bf16* __restrict__ dyg = in.dyg[layer];
  const bf16* __restrict__ x = in.x[layer];
  const int tj = tile / tiles_k, tk = tile - tj * tiles_k;
  const int j0 = tj * (TJ * NJ), k0 = tk * TK;
  const int m_begin = slab * rows_per_slab;
  int m_end = m_begin + rows_per_slab;
  m_end = m_end < M ? m_end : M;
  const int nsteps = m_end > m_begin ? (m_end - m_begin + TM - 1) / TM : 0;

  // piece p = wave*PW + it: sub-tile p>>3 (Y sub-tiles first), rows 8*(p&7) + (lane>>3), physical chunk lane&7
  auto issue = [&](int step) {
    char* st = smem + (step % NS) * STG;
    const int mb = m_begin + step * TM;
#pragma unroll
    for (int it = 0; it < PW; ++it) {
      const int pc = wave * PW + it, sub = pc >> 3;
      const int row = 8 * (pc & 7) + (lane >> 3);
      const int f = (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
      const int c = (lane & 7) ^ f;  // logical 16-byte chunk that lands at physical chunk lane&7 of this row
      int m = mb + row;
      m = m < M ? m : M - 1;  // tail rows are zeroed in LDS after they land (below)
      const bf16* src = sub < NJ ? dyg + (size_t)m * r + j0 + sub * 64 + c * 8
                                 : x + (size_t)m * ldx + k0 + (sub - NJ) * 64 + c * 8;
#if !defined(APLA_ABL_DW_NODMA)   // (diagnostic build: the step without its LDS-DMA)
      __builtin_amdgcn_global_load_lds(DW_GLBP(src), DW_LDSP(st + pc * 1024), 16, 0, 0);
#else
      asm volatile("" ::"v"(src));
#endif
    }
  };

  const int wk = wave * 32;  // this wave's 32 output columns (k)
  f32x16 acc[2 * NJ];
#pragma unroll
  for (int t = 0; t < 2 * NJ; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  float bsum[2 * NJ];  // tk==0 workgroups: column sums of dyg (the bias gradient), wave w sums the token rows of ks == w
#pragma unroll
  for (int t = 0; t < 2 * NJ; ++t) bsum[t] = 0.f;

  // fragment base addresses (stage 0, ks = 0, sub-tile 0): [column half of the 64-wide sub-tile][row half]
  unsigned yb[2][2], xb[2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    yb[0][hh] = tr_base(smem, 0, hh, lane);
    yb[1][hh] = tr_base(smem, 32, hh, lane);
    xb[hh] = tr_base(smem + NJ * 8192 + (wk >> 6) * 8192, wk & 63, hh, lane);
  }
  if (nsteps > 0) issue(0);
  if (NS > 2 && nsteps > 1) issue(1);
  // One 64-token step.  TAIL: token rows past the end exist in this step (only the last step of the last slab of a layer);
  // BIAS: this workgroup also sums dyg's columns (the tk == 0 tiles).  Both are compile-time so that the six products of a
  // 16-row step are issued back to back (as run-time tests they put two branches between every two MFMAs).
  auto step = [&](int s, auto TAILC, auto BIASC) {
    constexpr bool TAIL = TAILC.value != 0, BIAS = BIASC.value != 0;
      if (NS > 2 && s + 1 < nsteps) dw_wait_vmcnt<PW>(); else dw_wait_vmcnt<0>();   // stage s has landed (NS = 3: the next one may still fly)
    __builtin_amdgcn_s_barrier();  // stage s has landed for every wave; everyone is done reading stage s-1
    if (s + NS - 1 < nsteps) issue(s + NS - 1);
    const int valid = m_end - (m_begin + s * TM);  // token rows of this step that exist (< 64 only at the very end of M)
    // Rows past the end were fetched from a clamped address: their dyg fragment elements are forced to zero (element e
    // of a transposed fragment is token row 16*ks + 8*(e>>2) + 4*(lane>>5) + (e&3)).  No LDS writes here: a plain LDS
    // store next to in-flight LDS-DMA makes hipcc drain vmcnt(0) in front of the fragment reads of every step.
    const unsigned so = (unsigned)((s % NS) * STG);   // Ys = NJ x [64 m][64 j] at so, Xs = 2 x [64 m][64 k] behind it
    const unsigned xa0 = xb[0] + so, xa1 = xb[1] + so;
    const unsigned ya00 = yb[0][0] + so, ya01 = yb[0][1] + so, ya10 = yb[1][0] + so, ya11 = yb[1][1] + so;
    // The transposed fragment reads of 16-row step ks+1 are issued before the products of step ks (two register sets): with
    // one wave per SIMD nothing else hides the LDS latency.  LDS reads return in order, so lgkmcnt(2 + 4*NJ) = "all but the
    // reads just issued" retires exactly the set about to be multiplied.
    bf16x4 blo[2], bhi[2], alo[2][2 * NJ], ahi[2][2 * NJ];
    auto frags = [&](auto KS) {
      constexpr int ks = KS.value, st = ks & 1;
      blo[st] = tr_read<2048 * ks>(xa0);  // B[k' = m][col = k]
      bhi[st] = tr_read<2048 * ks>(xa1);
      static_for<0, 2 * NJ>([&](auto T) {  // A[row = j][k' = m]
        constexpr int t = T.value;
        alo[st][t] = tr_read<2048 * ks + 8192 * (t >> 1)>((t & 1) ? ya10 : ya00);
        ahi[st][t] = tr_read<2048 * ks + 8192 * (t >> 1)>((t & 1) ? ya11 : ya01);
      });
    };
#if defined(APLA_ABL_DW_NOMMA)     // (diagnostic build: DMA, waits and barriers only)
    if (s >= 0) return;
#endif
    frags(IntC<0>{});
    static_for<0, 4>([&](auto KS) {
      constexpr int ks = KS.value, st = ks & 1;
      if constexpr (ks + 1 < 4) {
        frags(IntC<ks + 1>{});
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 + 4 * NJ) : "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      const bf16x8 b = join8(blo[st], bhi[st]);
#pragma unroll
      for (int t = 0; t < 2 * NJ; ++t) {
        bf16x8 a = join8(alo[st][t], ahi[st][t]);
        if constexpr (TAIL) {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (16 * ks + 8 * (e >> 2) + 4 * h2 + (e & 3) >= valid) a[e] = (bf16)0.f;
        }
        acc[t] = MFMA_F32_32x32x16_H16(a, b, acc[t]);
        if constexpr (BIAS) {
          if (ks == wave) {  // lane holds dyg[8 token rows][j = 32t + (lane&31)]
            float sm = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) sm += (float)a[e];
            bsum[t] += sm;
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  const bool ragged = nsteps > 0 && m_end - (m_begin + (nsteps - 1) * TM) < TM;  // the last step holds fewer than 64 rows
  const int nfull = ragged ? nsteps - 1 : nsteps;
  if (tk == 0) {
    for (int s = 0; s < nfull; ++s) step(s, IntC<0>{}, IntC<1>{});
    if (ragged) step(nsteps - 1, IntC<1>{}, IntC<1>{});
  } else {
    for (int s = 0; s < nfull; ++s) step(s, IntC<0>{}, IntC<0>{});
    if (ragged) step(nsteps - 1, IntC<1>{}, IntC<0>{});
  }
  // D[j][k]: lane col = k (lane&31), rows j = acc_row(reg, h2).  The tile goes through LDS (the ring is idle now) so that
  // the partial slab is written with 16-byte stores of whole 512-byte rows instead of 32*NJ dword stores per wave.
#if defined(APLA_ABL_DW_NOEPI)     // (diagnostic build: no partial tile leaves the workgroup)
  if (acc[0][0] != 123.456f) return;
#endif
  __syncthreads();
  float* Ts = (float*)smem;  // [64*NJ][128] fp32, row pitch 132 floats (bank spread for the column-wise writes)
#pragma unroll
  for (int t = 0; t < 2 * NJ; ++t)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int j = 32 * t + (reg & 3) + 8 * (reg >> 2) + 4 * h2;
      Ts[j * 132 + wk + (lane & 31)] = acc[t][reg];
    }
  __syncthreads();
  float* P = partial + (size_t)unit * r * D;
  for (int e = tid; e < 64 * NJ * 32; e += 256) {
    const int j = e >> 5, c4 = (e & 31) * 4;
    *(f32x4*)(P + (size_t)(j0 + j) * D + k0 + c4) = *(const f32x4*)(Ts + j * 132 + c4);
  }
  if (tk == 0) {  // fold the two lane halves, then the four waves (fixed order), through LDS
    __syncthreads();
    float* Bs = (float*)smem;  // [4 waves][64*NJ]
#pragma unroll
    for (int t = 0; t < 2 * NJ; ++t) {
      const float v = bsum[t] + __shfl_xor(bsum[t], 32, 64);
      if (h2 == 0) Bs[wave * 64 * NJ + 32 * t + (lane & 31)] = v;
    }
    __syncthreads();
    if (tid < 64 * NJ)
      partial[(size_t)U * r * D + (size_t)unit * r + j0 + tid] =
          (Bs[tid] + Bs[64 * NJ + tid]) + (Bs[2 * 64 * NJ + tid] + Bs[3 * 64 * NJ + tid]);
  }
}

// Slab sum in a fixed order (bitwise reproducible).  Four lanes share one float4 of the output: lane q of the quad sums
// slabs q, q+4, q+8, ... (eight loads in flight), then the quad is folded 0+2, 1+3, (0+2)+(1+3).
__global__ __launch_bounds__(256) void proj_dw_reduce_kernel(const float* __restrict__ partial_all, DwOut out, int nb, int r,
                                                             int D, int S, int accumulate) {
  const int layer = blockIdx.y;  // this layer's S slabs are units layer*S .. layer*S + S-1
  const float* __restrict__ partial = partial_all + (size_t)layer * S * r * D;
  const float* __restrict__ row_scale = out.row_scale[layer];
  float* __restrict__ dW1 = out.dW[layer];
  float* __restrict__ db1 = out.db[layer];
  const long n4 = (long)r * D / 4;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long i = t >> 2;
  const int q = (int)(t & 3);
  const size_t stride = (size_t)r * D;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (i < n4) {
    const float* src = partial + i * 4;
    for (int sl = q; sl < S; sl += 64) {  // sixteen loads in flight; slabs past S contribute zero
      f32x4 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (sl + 4 * u < S) v[u] = *(const f32x4*)(src + (size_t)(sl + 4 * u) * stride);
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) s += v[u];
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    s[e] += __shfl_xor(s[e], 2, 64);
    s[e] += __shfl_xor(s[e], 1, 64);
  }
  if (i < n4 && q == 0) {
    const int j = (int)(i * 4 / D);
    if (row_scale != nullptr) s *= row_scale[j];
    if (accumulate) s += *(const f32x4*)(dW1 + i * 4);
    *(f32x4*)(dW1 + i * 4) = s;
  }
  if (t < r) {
    const float* pb = partial_all + (size_t)nb * S * stride + (size_t)layer * S * r;
    float b = 0.f;
    for (int sl = 0; sl < S; ++sl) b += pb[(size_t)sl * r + t];
    if (row_scale != nullptr) b *= row_scale[t];
    if (accumulate) b += db1[t];
    db1[t] = b;
  }
}

inline int dw_slabs(int M, int r, int D, int nb = 1) {
  const int tiles = (r / (TJ * dw_group(r))) * (D / TK);
  const int cap = dw_ring(r, D) == 2 ? 512 : 256;   // workgroups resident at once
  int S = cap / tiles / 8 * 8;  // a multiple of 8 (one slab set per XCD), as many workgroups as are resident at once
  if (tiles >= 256) return 1;   // more output tiles than CUs: no token split (every slab would cost an r x D fp32 partial)
  if (nb > 1) {                 // units = nb * S are dealt to the XCDs one by one: any S, still no more workgroups than are resident
    S = cap / (tiles * nb);
    const int max_s = (M + TM - 1) / TM;
    return S < 1 ? 1 : (S > max_s ? max_s : S);
  }
#if defined(APLA_ABL_DWSLABS)  // diagnostic build: slab count from the environment
  if (const char* e = getenv("APLA_DW_SLABS")) S = atoi(e);
#endif
  if (S < 8) S = 8;             // every slab costs an r x D fp32 partial to write and re-read
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

// all blocks of a model in one launch: block l's trainable rows sit at flat + l*block_stride (W1 [r,D] then b1 [r])
__global__ __launch_bounds__(256) void pack_proj_rows_batched_kernel(const float* __restrict__ flat, long block_stride,
                                                                     const int32_t* __restrict__ inds_all,
                                                                     const float* __restrict__ gamma_all,
                                                                     bf16* __restrict__ Wnat_all, bf16* __restrict__ WnatT_all,
                                                                     float* __restrict__ bnat_all, bf16* __restrict__ Wp_all,
                                                                     bf16* __restrict__ WTp_all, int r, int D) {
  const int j = blockIdx.x, l = blockIdx.y;
  const float* W1 = flat + (long)l * block_stride;
  const float* b1 = W1 + (long)r * D;
  const int row = inds_all[(long)l * D + j];
  const float g = gamma_all != nullptr ? gamma_all[(long)l * D + row] : 1.0f;
  bf16* Wnat = Wnat_all + (long)l * D * D;
  bf16* WnatT = WnatT_all + (long)l * D * D;
  bf16* Wp = Wp_all ? Wp_all + (long)l * D * D : nullptr;     // K-panel images of the two copies (apla_pack_k_panels layout)
  bf16* WTp = WTp_all ? WTp_all + (long)l * D * D : nullptr;
  for (int k = threadIdx.x; k < D; k += 256) {
    const bf16 v = (bf16)(g * W1[(size_t)j * D + k]);
    Wnat[(size_t)row * D + k] = v;
    WnatT[(size_t)k * D + row] = v;
    if (Wp) Wp[((size_t)(k >> 5) * D + row) * 32 + (k & 31)] = v;
    if (WTp) WTp[((size_t)(row >> 5) * D + k) * 32 + (row & 31)] = v;
  }
  if (threadIdx.x == 0) bnat_all[(long)l * D + row] = g * b1[j];
}

// The same re-scatter organised by PANELS of 32 natural rows (round 3).  The row-per-workgroup kernel above writes a column of
// the transposed copy (and of its K-panel image) per trainable row: 2-byte stores 1.5 KB apart, 3.5 million of them per step at
// config 2 (40 us).  Here a workgroup owns the 32 natural rows of one panel of one block: it finds the trainable rows that fall
// into its panel (a scan of the block's r indices), converts them once into LDS, writes their rows of the natural-order weight
// and of its image, and PATCHES the panel's 64-byte segments of the transposed weight and the panel's contiguous D x 64 B block
// of the transposed image: read 64 bytes, replace the trainable positions, write 64 bytes — whole sectors, coalesced over k.
// Frozen entries are rewritten with their own values.  Needs D % 32 == 0; LDS 32 * (D / gridDim.z) * 2 bytes.
__global__ __launch_bounds__(256) void pack_proj_panels_kernel(const float* __restrict__ flat, long block_stride,
                                                               const int32_t* __restrict__ inds_all,
                                                               const float* __restrict__ gamma_all, bf16* __restrict__ Wnat_all,
                                                               bf16* __restrict__ WnatT_all, float* __restrict__ bnat_all,
                                                               bf16* __restrict__ Wp_all, bf16* __restrict__ WTp_all, int r, int D) {
  extern __shared__ __attribute__((aligned(16))) char pk_smem[];
  bf16* tile = (bf16*)pk_smem;               // [32][Dz]: row (natural row & 31) of the panel, valid where owner[] >= 0; this
                                             // workgroup's quarter of the k axis (blockIdx.z): 4x the workgroups of a panel-per-
                                             // workgroup grid, which left the launch latency-bound (288 workgroups at config 2)
  const int Dz = D / (int)gridDim.z, kz = blockIdx.z * Dz;
  __shared__ int owner[32];                  // trainable index j of the panel's natural row, or -1
  const int p = blockIdx.x, l = blockIdx.y, tid = threadIdx.x;
  const float* W1 = flat + (long)l * block_stride;
  const float* b1 = W1 + (long)r * D;
  const int32_t* inds = inds_all + (long)l * D;
  if (tid < 32) owner[tid] = -1;
  __syncthreads();
  for (int j = tid; j < r; j += 256) {       // every natural row has at most one j: no two threads write one slot
    const int row = inds[j];
    if ((row >> 5) == p) owner[row & 31] = j;
  }
  __syncthreads();
  bf16* Wnat = Wnat_all + (long)l * D * D;
  bf16* WnatT = WnatT_all + (long)l * D * D;
  bf16* Wp = Wp_all ? Wp_all + (long)l * D * D : nullptr;
  bf16* WTp = WTp_all ? WTp_all + (long)l * D * D : nullptr;
  int any = 0;
  for (int i = 0; i < 32; ++i) {             // uniform over the workgroup
    const int j = owner[i];
    if (j < 0) continue;
    any = 1;
    const int row = 32 * p + i;
    const float g = gamma_all != nullptr ? gamma_all[(long)l * D + row] : 1.0f;
    for (int q4 = tid; q4 < Dz / 4; q4 += 256) {
      const int k4 = kz / 4 + q4;
      const f32x4 w = *(const f32x4*)(W1 + (size_t)j * D + k4 * 4);
      const bf16x4 v = pack4(g * w[0], g * w[1], g * w[2], g * w[3]);
      *(bf16x4*)(tile + i * Dz + q4 * 4) = v;
      *(bf16x4*)(Wnat + (size_t)row * D + k4 * 4) = v;
      if (Wp) *(bf16x4*)(Wp + ((size_t)(k4 >> 3) * D + row) * 32 + (k4 & 7) * 4) = v;
    }
    if (tid == 0 && blockIdx.z == 0) bnat_all[(long)l * D + row] = g * b1[j];
  }
  if (!any) return;
  __syncthreads();
  for (int kq = tid; kq < Dz; kq += 256) {   // the panel's 32 entries of transposed row k: 64 bytes in either layout
    const int k = kz + kq;
    bf16x8 seg[4];
    bf16* tp = WnatT + (size_t)k * D + 32 * p;
#pragma unroll
    for (int c = 0; c < 4; ++c) seg[c] = *(const bf16x8*)(tp + 8 * c);
#pragma unroll
    for (int i = 0; i < 32; ++i)
      if (owner[i] >= 0) seg[i >> 3][i & 7] = tile[i * Dz + kq];
#pragma unroll
    for (int c = 0; c < 4; ++c) *(bf16x8*)(tp + 8 * c) = seg[c];
    if (WTp) {
      bf16* ip = WTp + ((size_t)p * D + k) * 32;
#pragma unroll
      for (int c = 0; c < 4; ++c) *(bf16x8*)(ip + 8 * c) = seg[c];   // same 32 values: the image of the transposed weight
    }
  }
}

}  // namespace

extern "C" long apla_dw_workspace_bytes(int M, int r, int D) {
  if (M <= 0 || r <= 0 || D <= 0 || r % TJ != 0 || D % TK != 0) return -1;
  const long S = dw_slabs(M, r, D);
  return S * ((long)r * D + r) * (long)sizeof(float);
}

extern "C" long apla_dw_workspace_bytes_batched(int M, int r, int D, int nb) {
  if (M <= 0 || r <= 0 || D <= 0 || r % TJ != 0 || D % TK != 0 || nb < 1 || nb > DW_MAX_BATCH) return -1;
  const long U = (long)nb * dw_slabs(M, r, D, nb);
  return U * ((long)r * D + r) * (long)sizeof(float);
}

extern "C" int apla_proj_dw_batched(int nb, const void* const* dyg, const void* const* x, int ldx, const float* const* row_scale,
                                    float* const* dW1, float* const* db1, void* partial, int M, int r, int D, int accumulate,
                                    hipStream_t stream) {
  APLA_REQUIRE(nb >= 1 && nb <= DW_MAX_BATCH, "apla_proj_dw_batched: 1..%d layers per call (got %d)", DW_MAX_BATCH, nb);
  APLA_REQUIRE(dyg && x && dW1 && db1 && partial, "apla_proj_dw: null pointer");
  APLA_REQUIRE(M > 0 && r > 0 && r % TJ == 0 && D % TK == 0, "apla_proj_dw: need r%%64==0 and D%%128==0 (r=%d D=%d)", r, D);
  APLA_REQUIRE(ldx % 8 == 0 && ldx >= D && apla_aligned16(partial), "apla_proj_dw: alignment");
  DwIn in{};
  DwOut out{};
  for (int l = 0; l < nb; ++l) {
    APLA_REQUIRE(dyg[l] && x[l] && dW1[l] && db1[l], "apla_proj_dw: null pointer (layer %d of the batch)", l);
    APLA_REQUIRE(apla_aligned16(dyg[l]) && apla_aligned16(x[l]) && apla_aligned16(dW1[l]), "apla_proj_dw: alignment");
    in.dyg[l] = (const bf16*)dyg[l];
    in.x[l] = (const bf16*)x[l];
    out.dW[l] = dW1[l];
    out.db[l] = db1[l];
    out.row_scale[l] = row_scale ? row_scale[l] : nullptr;
  }
  const int S = dw_slabs(M, r, D, nb), U = nb * S;
  int rows_per_slab = ((M + S - 1) / S + TM - 1) / TM * TM;
  const int nj = dw_group(r);
  const dim3 grid(U == 1 ? (r / (TJ * nj)) * (D / TK) : 8 * ((U + 7) / 8) * (r / (TJ * nj)) * (D / TK));
  const int ns = dw_ring(r, D);
#define DW_LAUNCH(NJV, NSV) hipLaunchKernelGGL((proj_dw_partial_kernel<NJV, NSV>), grid, dim3(256), 0, stream, in, nb, ldx, (float*)partial, M, r, D, rows_per_slab, S)
  if (nj == 3) DW_LAUNCH(3, 3);
  else if (nj == 2) { if (ns == 2) DW_LAUNCH(2, 2); else DW_LAUNCH(2, 3); }
  else { if (ns == 2) DW_LAUNCH(1, 2); else DW_LAUNCH(1, 3); }
#undef DW_LAUNCH
  APLA_CHECK_LAUNCH("apla_proj_dw[partial]");
  const long n4 = (long)r * D / 4;
  hipLaunchKernelGGL(proj_dw_reduce_kernel, dim3((unsigned)((4 * n4 + 255) / 256), nb), dim3(256), 0, stream, (const float*)partial, out, nb, r, D, S, accumulate);
  APLA_CHECK_LAUNCH("apla_proj_dw[reduce]");
  return APLA_OK;
}

extern "C" int apla_proj_dw(const void* dyg, const void* x, int ldx, const float* row_scale, float* dW1, float* db1,
                            void* partial, int M, int r, int D, int accumulate, hipStream_t stream) {
  return apla_proj_dw_batched(1, &dyg, &x, ldx, &row_scale, &dW1, &db1, partial, M, r, D, accumulate, stream);
}

extern "C" int apla_pack_proj_rows(const float* W1, const float* b1, const int32_t* inds, const float* gamma,
                                   void* Wnat, void* WnatT, float* bnat, int r, int D, hipStream_t stream) {
  APLA_REQUIRE(W1 && inds && Wnat && WnatT && r > 0 && r <= D, "apla_pack_proj_rows: bad arguments");
  APLA_REQUIRE(b1 == nullptr || bnat != nullptr, "apla_pack_proj_rows: bnat required with b1");
  hipLaunchKernelGGL(pack_proj_rows_kernel, dim3(r), dim3(256), 0, stream, W1, b1, inds, gamma, (bf16*)Wnat, (bf16*)WnatT, bnat, r, D);
  APLA_CHECK_LAUNCH("apla_pack_proj_rows");
  return APLA_OK;
}

extern "C" int apla_pack_proj_rows_batched_ex(const float* flat, long block_stride, const int32_t* inds_all,
                                              const float* gamma_all, void* Wnat_all, void* WnatT_all, float* bnat_all,
                                              void* Wnat_panels, void* WnatT_panels, int L, int r, int D, hipStream_t stream) {
  APLA_REQUIRE(flat && inds_all && Wnat_all && WnatT_all && bnat_all && L > 0 && L <= 65535 && r > 0 && r <= D &&
               block_stride >= (long)r * D + r, "apla_pack_proj_rows_batched: bad arguments");
  APLA_REQUIRE((!Wnat_panels && !WnatT_panels) || D % 32 == 0, "apla_pack_proj_rows_batched_ex: K-panel images need D %% 32 == 0");
  if (D % 32 == 0 && (D % 128 == 0 ? D <= 4096 : D <= 1024) && apla_aligned16(flat) && block_stride % 4 == 0 && apla_aligned16(Wnat_all) && apla_aligned16(WnatT_all) &&
      apla_aligned16(Wnat_panels) && apla_aligned16(WnatT_panels)) {   // panel form (whole-sector writes): 32 * D * 2 <= 64 KB of LDS
    const int zs = (D % 128 == 0) ? 4 : 1;
    hipLaunchKernelGGL(pack_proj_panels_kernel, dim3(D / 32, L, zs), dim3(256), (size_t)32 * (D / zs) * 2, stream, flat, block_stride, inds_all, gamma_all,
                       (bf16*)Wnat_all, (bf16*)WnatT_all, bnat_all, (bf16*)Wnat_panels, (bf16*)WnatT_panels, r, D);
    APLA_CHECK_LAUNCH("apla_pack_proj_rows_batched");
    return APLA_OK;
  }
  hipLaunchKernelGGL(pack_proj_rows_batched_kernel, dim3(r, L), dim3(256), 0, stream, flat, block_stride, inds_all, gamma_all, (bf16*)Wnat_all, (bf16*)WnatT_all, bnat_all, (bf16*)Wnat_panels, (bf16*)WnatT_panels, r, D);
  APLA_CHECK_LAUNCH("apla_pack_proj_rows_batched");
  return APLA_OK;
}

extern "C" int apla_pack_proj_rows_batched(const float* flat, long block_stride, const int32_t* inds_all,
                                           const float* gamma_all, void* Wnat_all, void* WnatT_all, float* bnat_all,
                                           int L, int r, int D, hipStream_t stream) {
  return apla_pack_proj_rows_batched_ex(flat, block_stride, inds_all, gamma_all, Wnat_all, WnatT_all, bnat_all, nullptr, nullptr, L, r, D, stream);
}

// ------------------------------------------------------------------------------------------------ composite operator entry points
// The APLA projection as ONE forward and ONE backward call (the operator granularity SURVEY section 8b lists), composed of the
// entry points above and apla_gemm_nt / apla_gather_cols.  No additional kernels.
extern "C" int apla_gemm_nt(const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc, int M, int N,
                            int K, int epilogue, int out_dtype, const void* aux_in, int ld_aux_in, void* aux_out,
                            int ld_aux_out, hipStream_t stream);
extern "C" int apla_gather_cols(const void* src, int res_dtype, long src_row_stride, const int32_t* inds, int r, void* out, int M,
                                int D, hipStream_t stream);

static inline int proj_r_pad(int r) { return (r + TJ - 1) / TJ * TJ; }   // the dW kernel works on multiples of 64 rows

extern "C" long apla_proj_workspace_bytes(int M, int D, int r) {
  if (M <= 0 || D <= 0 || r <= 0 || r > D || D % TK != 0 || proj_r_pad(r) > D) return -1;
  const int rp = proj_r_pad(r);
  const long dw = apla_dw_workspace_bytes(M, rp, D);
  if (dw < 0) return -1;
  // [dyg M x rp 16-bit | dW rp x D fp32 | db rp fp32 | slab partials], each 256-byte aligned
  auto al = [](long v) { return (v + 255) / 256 * 256; };
  return al((long)M * rp * 2) + al((long)rp * D * 4) + al((long)rp * 4) + dw;
}

extern "C" int apla_proj_fwd(const void* x, const void* Wnat, const float* bnat, const int32_t* inds, void* y, int M, int D, int r,
                             hipStream_t stream) {
  (void)inds; (void)r;   // the natural-order weight already holds the scatter (apla_pack_proj_rows): one GEMM
  APLA_REQUIRE(x && Wnat && y && M > 0 && D > 0, "apla_proj_fwd: bad arguments");
  return apla_gemm_nt(x, D, Wnat, D, bnat, y, D, M, D, D, APLA_EPI_STORE, APLA_H16, nullptr, 0, nullptr, 0, stream);
}

extern "C" int apla_proj_bwd(const void* dy, const void* x, const void* WnatT, const int32_t* inds, void* dx, float* dW1, float* db1,
                             void* workspace, long workspace_bytes, int M, int D, int r, int accumulate, hipStream_t stream) {
  APLA_REQUIRE(dy && x && inds && dW1 && db1 && workspace && M > 0 && D > 0 && r > 0, "apla_proj_bwd: bad arguments");
  const long need = apla_proj_workspace_bytes(M, D, r);
  APLA_REQUIRE(need > 0 && workspace_bytes >= need && apla_aligned16(workspace), "apla_proj_bwd: workspace (need %ld bytes, 16-byte aligned)", need);
  const int rp = proj_r_pad(r);
  auto al = [](long v) { return (v + 255) / 256 * 256; };
  char* ws = (char*)workspace;
  void* dyg = ws;
  float* dWp = (float*)(ws + al((long)M * rp * 2));
  float* dbp = (float*)((char*)dWp + al((long)rp * D * 4));
  void* part = (char*)dbp + al((long)rp * 4);
  int rc = APLA_OK;
  if (dx != nullptr) {   // dX = dY * Wnat  (through the transposed copy: NT GEMM)
    APLA_REQUIRE(WnatT != nullptr, "apla_proj_bwd: WnatT required for dx");
    rc = apla_gemm_nt(dy, D, WnatT, D, nullptr, dx, D, M, D, D, APLA_EPI_STORE, APLA_H16, nullptr, 0, nullptr, 0, stream);
    if (rc != APLA_OK) return rc;
  }
  // the r trainable columns of dY (padded with the next, frozen, indices: their rows are dropped below), then the masked dW
  rc = apla_gather_cols(dy, APLA_H16, D, inds, rp, dyg, M, D, stream);
  if (rc != APLA_OK) return rc;
  const bool direct = rp == r;   // no padding: the dW kernel writes (or accumulates into) the caller's buffers
  rc = apla_proj_dw(dyg, x, D, nullptr, direct ? dW1 : dWp, direct ? db1 : dbp, part, M, rp, D, direct ? accumulate : 0, stream);
  if (rc != APLA_OK || direct) return rc;
  if (accumulate) { apla_set_error("apla_proj_bwd: accumulate needs r %% 64 == 0"); return APLA_ENOSYS; }
  hipError_t e = hipMemcpyAsync(dW1, dWp, (size_t)r * D * 4, hipMemcpyDeviceToDevice, stream);
  if (e == hipSuccess) e = hipMemcpyAsync(db1, dbp, (size_t)r * 4, hipMemcpyDeviceToDevice, stream);
  if (e != hipSuccess) { apla_set_error("apla_proj_bwd: copy failed: %s", hipGetErrorString(e)); return APLA_EIO; }
  return APLA_OK;
}
