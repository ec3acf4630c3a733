// Loader-wave form of the 4-wave persistent GEMM (gemm_nt.hip: gemm_persist_kernel), round 6: C[M,N] = A[M,K] W[N,K]^T with the
// epilogues of gemm_common.h:persist_epilogue, for gfx950.
//
// Same tile (32 MI x 128 x 64), same LDS image and swizzles, same tile walk and two workgroups per CU as gemm_persist_kernel, so
// the results are bit-identical.  What differs is WHO moves the operands: a workgroup has 4 compute waves + NL loader waves.  The
// compute waves issue no LDS-DMA and count no vmcnt — a K-step is one barrier, 18 fragment reads and 40 MFMAs; the loader waves
// issue every piece of the two-stage ring (20 A + 16 W pieces of 1 KB per K-step, the bias piece with a tile's first stage) from
// wave-uniform 64-bit bases in scalar registers plus one of two per-lane offsets (even / odd piece: the XOR swizzle depends on the
// piece's parity only), wait for their pieces (vmcnt(0)) and meet the compute waves at the K-step's barrier.
//
// Register budget: 2 workgroups x (4 + NL) waves = 10 or 12 waves per CU = 3 on a SIMD -> at most 168 VGPRs per lane; the compute
// path holds 80 accumulators + 36 fragment registers and fits without the DMA address registers of the 4-wave kernel.
#include <cstdio>

#include "gemm_common.h"

#if defined(APLA_ABL_CLOCK)  // diagnostic build: per-workgroup clock stamps (tools/gemm_clock.py)
__device__ unsigned long long apla_abl_clock_buf_lw[1024];
extern "C" int apla_abl_clock_lw(unsigned long long* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(apla_abl_clock_buf_lw), sizeof(apla_abl_clock_buf_lw)); }
#endif

namespace {

constexpr int BN = 128, BK = 64;

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ void dma_piece(unsigned lds_dst, unsigned voff, const char* sbase) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_dst), "v"(voff), "s"(sbase) : "memory");
}

template <int EPI, typename OutT, int MI, int NL, int EXP>
__global__ __launch_bounds__(64 * (4 + NL), 2) void gemm_lw_kernel(GemmParams p, int tiles_m) {
  constexpr int BMv = 32 * MI;
  constexpr int STG = (BMv + BN) * BK * 2;
  constexpr int NA = 4 * MI, NW = 16;     // 1 KB pieces (8 rows x 128 B) per stage
  __shared__ __attribute__((aligned(16))) char smem[2 * STG + 2048];  // + 2 x 1 KB bias pieces
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nk = p.K / BK;
  // tile range of this workgroup: XCD x (= blockIdx % 8 under round-robin dispatch; speed only) owns a contiguous run
  const int total = tiles_m * p.tiles_n;
  const int G = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int q = total >> 3, rr = total & 7;
  const int xbeg = xcd * q + (xcd < rr ? xcd : rr), xcnt = q + (xcd < rr ? 1 : 0);
  const int slots = (G >> 3) + ((G & 7) > xcd ? 1 : 0);
  if (slot >= xcnt) return;
#if defined(APLA_ABL_CLOCK)
  const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif

  if (wave >= 4) {
    // ---------------------------------------------------------------- loader wave(s) ----------------------------------------------
    __builtin_amdgcn_s_setprio(3);
    const int lw = wave - 4;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const bool has_bias = p.bias != nullptr;
    const int r8 = lane >> 3;
    unsigned a_lane[2], w_lane[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const int chunk = (lane & 7) ^ ((par * 4 + (lane >> 4)) & 7);
      a_lane[par] = (unsigned)((r8 * p.lda + chunk * 8) * 2);
      w_lane[par] = (unsigned)(((8 * (lane >> 5) + ((lane >> 3) & 3)) * p.ldw + chunk * 8) * 2);
    }
    const char* a_tile = nullptr;
    const char* w_tile = nullptr;
    const char* b_src = nullptr;                   // the tile's 128 bias values (wave-uniform); lane l reads 16 bytes at b_off
    const unsigned b_off = (lane & 31) * 16;       // lanes 32..63 duplicate (keeps EXEC full, stays in bounds)
    int last = 0;
    auto setup = [&](int tile) {
      int tm, tn;
      tile_coords(tile, tiles_m, p.tiles_n, p.ngrp, tm, tn);
      last = p.M - 1 - tm * BMv;    // last valid row of this tile (>= BMv - 1 except in the last row tile)
      a_tile = (const char*)(p.A + (size_t)tm * BMv * p.lda);
      w_tile = (const char*)(p.W + (size_t)tn * BN * p.ldw);
      if (has_bias) b_src = (const char*)(p.bias + tn * BN);
    };
    auto stage = [&](int sbuf, int k0) {
      const unsigned base = lds0 + sbuf * STG;
      const char* ak = a_tile + (size_t)k0 * 2;
      const char* wk = w_tile + (size_t)k0 * 2;
#pragma unroll
      for (int pc = 0; pc < NA + NW; ++pc) {
        if (pc % NL != lw) continue;
        if (pc < NA) {
          if (pc * 8 + 7 <= last) {
            dma_piece(base + pc * 1024, a_lane[pc & 1], ak + (size_t)(pc * 8) * p.lda * 2);
          } else {   // last row tile: rows past M - 1 re-read row M - 1 (never stored)
            const int row = pc * 8 + r8;
            const int rel = row < last ? row : last;
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            dma_piece(base + pc * 1024, (unsigned)((rel * p.lda + chunk * 8) * 2), ak);
          }
        } else {
          const int pw = pc - NA;
          // LDS rows 8 pw .. 8 pw + 7 hold W rows w_row_of_lds_row(.): the wave-uniform part of that permutation
          const int wrow = (pw >> 3) * 64 + 32 * ((pw >> 2) & 1) + 16 * (pw & 1) + 4 * ((pw >> 1) & 1);
          dma_piece(base + BMv * BK * 2 + pw * 1024, w_lane[pw & 1], wk + (size_t)wrow * p.ldw * 2);
        }
      }
    };
    int idx = slot;
    setup(xbeg + idx);
    stage(0, 0);
    if (has_bias && lw == 0) dma_piece(lds0 + 2 * STG, b_off, b_src);
    int cur = 0, bb = 0;
    while (true) {
      const int nidx = idx + slots;
      const bool has_next = nidx < xcnt;
      for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // stage (tile, kt) has landed; every compute wave has left buffer cur ^ 1
        if (kt + 1 < nk) {
          stage(cur ^ 1, (kt + 1) * BK);
        } else if (has_next) {
          setup(xbeg + nidx);
          stage(cur ^ 1, 0);
          if (has_bias && lw == 0) dma_piece(lds0 + 2 * STG + (bb ^ 1) * 1024, b_off, b_src);
        }
        cur ^= 1;
      }
      if (!has_next) break;
      bb ^= 1;
      idx = nidx;
    }
    return;
  }

  // ------------------------------------------------------------------ compute waves -----------------------------------------------
  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 15, fq = lane >> 4;
  int idx = slot;
  int cur = 0, bb = 0;
  while (true) {
    int tm, tn;
    tile_coords(xbeg + idx, tiles_m, p.tiles_n, p.ngrp, tm, tn);
    const int m0 = tm * BMv, n0 = tn * BN;
    const int nidx = idx + slots;
    f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nk; ++kt) {
      __builtin_amdgcn_s_barrier();
      const char* As = smem + cur * STG;
      const char* Ws = As + BMv * BK * 2;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 af[MI], wf[4];
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = *(const bf16x8*)(As + lds_off(wm * (MI * 16) + i * 16 + frow, ks * 4 + fq));
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[j] = *(const bf16x8*)(Ws + lds_off(wn * 64 + j * 16 + frow, ks * 4 + fq));
        if constexpr ((EXP & 2) != 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = MFMA_F32_16x16x32_H16(wf[j], af[i], acc[i][j]);
        if constexpr ((EXP & 2) != 0) __builtin_amdgcn_s_setprio(0);
      }
      cur ^= 1;
    }
    persist_epilogue<EPI, OutT, MI>(p, acc, (const float*)(smem + 2 * STG + bb * 1024), m0, n0, wm, wn, lane);
    asm volatile("" ::: "memory");
    if (nidx >= xcnt) break;
    bb ^= 1;
    idx = nidx;
  }
#if defined(APLA_ABL_CLOCK)
  if (tid == 0) {
    apla_abl_clock_buf_lw[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - clk0;
    apla_abl_clock_buf_lw[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - rt0;
  }
#endif
}

constexpr int RESIDENT_WGS = 512;  // 256 CUs x 2 workgroups

template <int EPI, typename OutT, int NL>
int launch_lw(const GemmParams& p_in, hipStream_t stream) {
  constexpr int MI = 5;
  GemmParams p = p_in;
  p.ngrp = pick_ngrp(p.tiles_n, BN, p.K);
  const int tiles_m = (p.M + 32 * MI - 1) / (32 * MI);
  const int total = tiles_m * p.tiles_n;
  const int resident = RESIDENT_WGS - 2 * (p.reserve > 0 && p.reserve < 192 ? p.reserve : 0);
  const int G = total < resident ? total : resident;
  constexpr int EXP = (EPI == APLA_EPI_GELU) ? 2 : 0;    // s_setprio around the MFMA clusters, as the 4-wave kernel runs this epilogue
  hipLaunchKernelGGL((gemm_lw_kernel<EPI, OutT, MI, NL, EXP>), dim3(G), dim3(64 * (4 + NL)), 0, stream, p, tiles_m);
  APLA_CHECK_LAUNCH("apla_gemm_nt[loader-wave]");
  return APLA_OK;
}

}  // namespace

// Covered: the 16-bit GELU / GELU_FWD / MUL epilogues and the plain 16-bit store on row-major operands (output / second-operand
// images as the 4-wave kernel takes them).
bool apla_gemm_lw_covers(int M, int N, int K, long lda, long ldw, int epilogue, int out_dtype, int w_panel) {
  if (M <= 0 || N % BN != 0 || K % BK != 0 || K < 2 * BK || (w_panel & 3) != 0 || out_dtype != APLA_H16) return false;
  if ((long)160 * lda * 2 >= (1L << 31) || (long)BN * ldw * 2 >= (1L << 31)) return false;   // 32-bit lane offsets
  return epilogue == APLA_EPI_GELU || epilogue == APLA_EPI_GELU_FWD || epilogue == APLA_EPI_MUL || epilogue == APLA_EPI_STORE;
}

int apla_gemm_lw_launch(const GemmParams& p, int epilogue, int out_dtype, hipStream_t stream) {
  if (!apla_gemm_lw_covers(p.M, p.N, p.K, p.lda, p.ldw, epilogue, out_dtype, p.w_panel)) {
    apla_set_error("apla_gemm_nt[loader-wave]: problem not covered (M=%d N=%d K=%d epilogue %d)", p.M, p.N, p.K, epilogue);
    return APLA_ENOSYS;
  }
  const bool two = (p.exp & 1) != 0;    // GemmParams::exp bit 0: two loader waves per workgroup (A/B)
  switch (epilogue) {
    case APLA_EPI_GELU: return two ? launch_lw<APLA_EPI_GELU, bf16, 2>(p, stream) : launch_lw<APLA_EPI_GELU, bf16, 1>(p, stream);
    case APLA_EPI_GELU_FWD: return two ? launch_lw<APLA_EPI_GELU_FWD, bf16, 2>(p, stream) : launch_lw<APLA_EPI_GELU_FWD, bf16, 1>(p, stream);
    case APLA_EPI_MUL: return two ? launch_lw<APLA_EPI_MUL, bf16, 2>(p, stream) : launch_lw<APLA_EPI_MUL, bf16, 1>(p, stream);
    default: return two ? launch_lw<APLA_EPI_STORE, bf16, 2>(p, stream) : launch_lw<APLA_EPI_STORE, bf16, 1>(p, stream);
  }
}
