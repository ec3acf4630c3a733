// bf16 MFMA GEMM  C[M,N] = A[M,K] · W[N,K]^T  with fused epilogues, for gfx950: dispatcher + the 4-wave kernels.
//
// Every dense contraction on the APLA step is expressed in this "NT" form (both operands K-contiguous): forward
// linears use the frozen weight as stored by nn.Linear ([out,in]); the dX backward uses a transposed bf16 copy of the
// same frozen weight that the engine prepares once (288 GB of HBM makes the duplicate free), so no "NN" kernel exists.
//
// Three schedules share the data layout, swizzles and epilogue code (gemm_common.h); the schedule is a per-call argument
// (apla_gemm_nt_ex flags, 0 = the automatic rule in `launch` below):
//   * gemm_pp2.hip — 8-wave ping-pong kernel, 320x256x32 tile, one workgroup per CU (STORE / GELU epilogues, large M);
//   * gemm_persist_kernel (here) — 4 waves, (128|160)x128x64 tile, 2 persistent workgroups per CU whose LDS ring runs
//     across tiles (no load prologue, epilogue overlapped with the next tile's LDS-DMA and with the co-resident
//     workgroup's main loop); carries the operand epilogues (RESIDUAL, MUL, SwiGLU) and all small problems;
//   * gemm_nt_kernel (here) — the simple non-persistent 128x128x64 kernel this work started from (A/B baseline).
// Common to all: staging by __builtin_amdgcn_global_load_lds (16 B per lane) into an LDS image whose bank-conflict
// swizzle is applied to the per-lane SOURCE address (an LDS-DMA destination is lane-linear) and again on the
// ds_read_b128; v_mfma_f32_16x16x32_bf16 with the WEIGHT fragment as the A operand, i.e. C^T tiles, so that a lane owns
// consecutive output columns of one output row and the epilogue is a vector op on 16-byte global accesses; XCD-aware
// tile walk (gemm_common.h:tile_coords).
#include <cstdio>

#include "gemm_common.h"

#if defined(APLA_ABL_CLOCK)  // diagnostic build (tools/build_ablations.sh CLOCK): per-workgroup clock stamps of the persistent kernel
__device__ unsigned long long apla_abl_clock_buf_nt[1024];
extern "C" int apla_abl_clock_nt(unsigned long long* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(apla_abl_clock_buf_nt), sizeof(apla_abl_clock_buf_nt)); }
#endif

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;  // 32 KB

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int EPI, typename OutT, int MI = 4>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x4 (&acc)[MI][4], int m0, int n0, int wm, int wn,
                                              int lane) {
  const int frow = lane & 15, fq = lane >> 4;
  // ---- epilogue: lane owns row m = .. + (lane&15), columns n = .. + 4*(lane>>4) + {0..3} of each 16x16 tile ----
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = m0 + wm * (MI * 16) + i * 16 + frow;
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
        gelu_and_grad4(v, h, g);
        Vec4IO<bf16>::store((bf16*)p.C + (size_t)m * p.ldc + n, h);
        Vec4IO<bf16>::store((bf16*)p.aux_out + (size_t)m * p.ld_aux_out + n, g);
      } else if constexpr (EPI == APLA_EPI_GELU_FWD) {
        f32x4 h;
        h = gelu_only4(v);
        Vec4IO<bf16>::store((bf16*)p.C + (size_t)m * p.ldc + n, h);
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

// BNv = 128: the 2 x 2 wave grid this work started from (wave tile 64 x 64).  BNv = 64 (round 6): a 128 x 64 tile, the four waves
// stacked along M (wave tile 32 x 64) — the instance behind widths that are multiples of 64 but not of 128 (the reference's vit_tiny:
// D = 192, utils/transformers/vit.py:511-525); every epilogue, row-major operands, no speed claim.
template <int EPI, typename OutT, int BNv = BN>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmParams p) {
  constexpr int MI = BNv == 128 ? 4 : 2;            // 16-row fragments per wave
  constexpr int WP = BNv * BK * 2 / 1024 / 4;       // W pieces per wave and K-step
  constexpr int STG = (BM + BNv) * BK * 2;
  __shared__ __attribute__((aligned(16))) char smem[2 * STG];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = BNv == 128 ? wave >> 1 : wave, wn = BNv == 128 ? wave & 1 : 0;
  const int tiles_n = p.N / BNv;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = wg / tiles_n, tn = wg - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BNv;

  // per-lane source rows / chunks for the 4 A pieces and WP W pieces this wave stages per K-step
  const bf16* a_src[4];
  const bf16* w_src[WP];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int piece = wave * 4 + it;
    const int row = piece * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    int gr = m0 + row;
    gr = gr < p.M ? gr : p.M - 1;  // tail rows: re-read the last valid row, never stored
    a_src[it] = p.A + (size_t)gr * p.lda + chunk * 8;
  }
#pragma unroll
  for (int it = 0; it < WP; ++it) {
    const int row = (wave * WP + it) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    w_src[it] = p.W + (size_t)(n0 + row) * p.ldw + chunk * 8;
  }

  auto stage = [&](int s, int k0) {
    char* base = smem + s * STG;
#pragma unroll
    for (int it = 0; it < 4; ++it) __builtin_amdgcn_global_load_lds(GLBP(a_src[it] + k0), LDSP(base + (wave * 4 + it) * 1024), 16, 0, 0);
#pragma unroll
    for (int it = 0; it < WP; ++it) __builtin_amdgcn_global_load_lds(GLBP(w_src[it] + k0), LDSP(base + BM * BK * 2 + (wave * WP + it) * 1024), 16, 0, 0);
  };

  f32x4 acc[MI][4];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fq = lane >> 4;
  const int nk = p.K / BK;
  stage(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, (kt + 1) * BK);
    const char* As = smem + cur * STG;
    const char* Ws = As + BM * BK * 2;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[MI], wf[4];
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = *(const bf16x8*)(As + lds_off(wm * (MI * 16) + i * 16 + frow, ks * 4 + fq));
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[j] = *(const bf16x8*)(Ws + lds_off(wn * 64 + j * 16 + frow, ks * 4 + fq));
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = MFMA_F32_16x16x32_H16(wf[j], af[i], acc[i][j]);
    }
    __syncthreads();
  }

  gemm_epilogue<EPI, OutT, MI>(p, acc, m0, n0, wm, wn, lane);
}


// ---- persistent variant -------------------------------------------------------------------------------------------
// 2 workgroups per CU stay resident and walk the tile list (each XCD owns a contiguous run of tiles, n fastest).  The
// 2-stage LDS ring runs ACROSS tiles: during the last K-step of a tile the first K-step of the workgroup's next tile is
// already being DMA'd into the free buffer, so a tile has no load prologue and its epilogue (VALU + global stores)
// overlaps that DMA; the stores themselves drain under the next main loop.  LDS-DMA completion is tracked with counted
// s_waitcnt vmcnt + raw s_barrier (vmcnt retires in issue order: the epilogue stores are younger than the prefetch, so
// the first wait of the next tile leaves exactly those stores outstanding).  The tile's 128 bias values travel with its
// first K-step as one more LDS-DMA piece and are read back with ds_read in the epilogue: an ordinary global load there
// would make hipcc drain vmcnt(0) before every store (measured: 16 serialised load->wait->store round trips per tile).
// Epilogue operands (residual / multiplier) are loaded up front, all at once.  BM = 32*MI (128/160) is picked on the
// host to minimise tile-count quantisation over the 512 resident workgroups.
template <int EPI, typename OutT, int MI, int EXP = 0, bool DROP = false>
__global__ __launch_bounds__(256, 2) void gemm_persist_kernel(GemmParams p, int tiles_m) {
  constexpr int BMv = 32 * MI;
  constexpr int STG = (BMv + BN) * BK * 2;
  __shared__ __attribute__((aligned(16))) char smem[2 * STG + 2048];  // + 2 x 1 KB bias pieces
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 15, fq = lane >> 4;
  const int nk = p.K / BK;
  // tile range of this workgroup: XCD x (= blockIdx % 8 under round-robin dispatch; speed only) owns a contiguous run
  const int total = tiles_m * p.tiles_n;
  const int G = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int q = total >> 3, rr = total & 7;
  const int xbeg = xcd * q + (xcd < rr ? xcd : rr), xcnt = q + (xcd < rr ? 1 : 0);
  const int slots = (G >> 3) + ((G & 7) > xcd ? 1 : 0);
  const bool has_bias = p.bias != nullptr;

  // LDS-DMA addressing (round 5): a piece's source is a wave-uniform 64-bit base (tile origin + K offset: scalar registers, two scalar
  // adds per K-step) plus a 32-bit lane offset that does not depend on the K-step — `global_load_lds_dwordx4 voff, s[base]`, its LDS
  // destination written to m0 from scalar arithmetic.  With per-lane 64-bit pointers (18 registers) hipcc issued per piece and K-step
  // a 64-bit vector add, a vector add + v_readfirstlane for m0 (the wave index was not known to be uniform) and a 64-bit register
  // copy: 36 vector instructions per wave and K-step beside 40 MFMAs — and vector and matrix work of the two waves of a SIMD do not
  // overlap in these kernels.  The W offsets are the same for every tile; the A offsets differ only in the
  // last row tile (rows >= M re-read row M - 1).
  const char* a_base = nullptr;
  const char* w_base = nullptr;
  unsigned a_off[MI], w_off[4];
  const char* b_src = nullptr;                   // the tile's 128 bias values (wave-uniform); lane l reads 16 bytes at b_off
  const unsigned b_off = (lane & 31) * 16;       // lanes 32..63 duplicate (keeps EXEC full, stays in bounds)
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = (wave * 4 + it) * 8 + (lane >> 3);   // LDS row; holds W row w_row_of_lds_row(row)
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    w_off[it] = (unsigned)((w_row_of_lds_row(row) * p.ldw + chunk * 8) * 2);
  }
  auto setup = [&](int tile) {
    int tm, tn;
    tile_coords(tile, tiles_m, p.tiles_n, p.ngrp, tm, tn);
    const int last = p.M - 1 - tm * BMv;    // last valid row of this tile (>= BMv - 1 except in the last row tile)
#pragma unroll
    for (int it = 0; it < MI; ++it) {
      const int row = (wave * MI + it) * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      const int rel = row < last ? row : last;
      a_off[it] = (unsigned)((rel * p.lda + chunk * 8) * 2);
    }
    a_base = (const char*)(p.A + (size_t)tm * BMv * p.lda);
    w_base = (const char*)(p.W + (size_t)tn * BN * p.ldw);
    if (has_bias) b_src = (const char*)(p.bias + tn * BN);
  };
  auto dma = [&](unsigned lds_dst, unsigned voff, const char* sbase) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_dst), "v"(voff), "s"(sbase) : "memory");
  };
#if defined(APLA_ABL_NODMA)   // diagnostic build: only the first two stages of a workgroup are loaded (both ring buffers then hold real
  int abl_dma_left = 2;       // operands); what the K loop costs once NO wave issues LDS-DMA = the ceiling of any loader-wave variant
#endif
  auto stage = [&](int sbuf, int k0) {
#if defined(APLA_ABL_NODMA)
    if (abl_dma_left <= 0) return;
    --abl_dma_left;
#endif
    const unsigned base = lds0 + sbuf * STG;
    const char* ak = a_base + (size_t)k0 * 2;
    const char* wk = w_base + (size_t)k0 * 2;
#pragma unroll
    for (int it = 0; it < MI; ++it) dma(base + (wave * MI + it) * 1024, a_off[it], ak);
#pragma unroll
    for (int it = 0; it < 4; ++it) dma(base + BMv * BK * 2 + (wave * 4 + it) * 1024, w_off[it], wk);
  };
  // the bias piece goes through the same helper as the operand pieces: every LDS-DMA of this kernel writes m0 itself, right in front
  // of its instruction (hipcc does not know that the asm changes m0; a builtin DMA beside it would rely on hipcc re-materialising it)
  auto stage_bias = [&](int bbuf) {
    if (has_bias && wave == 0) dma(lds0 + 2 * STG + bbuf * 1024, b_off, b_src);
  };

  int idx = slot;
  if (idx >= xcnt) return;
#if defined(APLA_ABL_CLOCK)  // diagnostic build: the core clock this workgroup ran at (tools/gemm_clock.py)
  const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif

  int tile = xbeg + idx;
  setup(tile);
  stage(0, 0);
  stage_bias(0);
  int cur = 0, bb = 0;
#if defined(APLA_ABL_NOREAD)
  bf16x8 abl_af[MI], abl_wf[4];
  bool abl_have = false;
#endif
  bool counted = false;  // may the first wait of this tile leave the previous tile's stores outstanding?
  while (true) {
    int tm, tn;
    tile_coords(tile, tiles_m, p.tiles_n, p.ngrp, tm, tn);
    const int m0 = tm * BMv, n0 = tn * BN;
    const int nidx = idx + slots;
    const bool has_next = nidx < xcnt;
    f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nk; ++kt) {
      if (kt == 0 && counted) wait_vmcnt<EpiStores<EPI, OutT, MI>::N>(); else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      const char* As = smem + cur * STG;
      const char* Ws = As + BMv * BK * 2;
      auto issue_next = [&]() {
        if (kt + 1 < nk) {
          stage(cur ^ 1, (kt + 1) * BK);
        } else if (has_next) {
          setup(xbeg + nidx);
          stage(cur ^ 1, 0);
          stage_bias(bb ^ 1);
        }
      };
      if constexpr ((EXP & 1) == 0) {
        issue_next();
        asm volatile("" ::: "memory");
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#if defined(APLA_ABL_NOREAD)   // diagnostic build: the fragments of the very first K-half stay in registers for the whole run
        bf16x8 (&af)[MI] = abl_af; bf16x8 (&wf)[4] = abl_wf;
        if (!abl_have) {
          abl_have = true;
#else
        bf16x8 af[MI], wf[4];
        {
#endif
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = *(const bf16x8*)(As + lds_off(wm * (MI * 16) + i * 16 + frow, ks * 4 + fq));
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[j] = *(const bf16x8*)(Ws + lds_off(wn * 64 + j * 16 + frow, ks * 4 + fq));
        }
        if constexpr ((EXP & 1) != 0) {
          // experiment 1: the first K-half's fragment reads are issued BEFORE the next stage's LDS-DMA, whose ~9 issue slots then
          // cover the reads' latency (as the ping-pong kernel's prepare phase does)
          if (ks == 0) {
            asm volatile("" ::: "memory");
            issue_next();
            asm volatile("" ::: "memory");
          }
        }
        if constexpr ((EXP & 2) != 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = MFMA_F32_16x16x32_H16(wf[j], af[i], acc[i][j]);
        if constexpr ((EXP & 2) != 0) __builtin_amdgcn_s_setprio(0);
      }
      cur ^= 1;
    }
    persist_epilogue<EPI, OutT, MI, DROP>(p, acc, (const float*)(smem + 2 * STG + bb * 1024), m0, n0, wm, wn, lane);
    asm volatile("" ::: "memory");
    if (!has_next) break;
    // a full (non-tail) tile issues exactly EpiStores::N store instructions per wave after the prefetch (epilogue operand
    // loads, if any, complete before the stores are issued); a tail tile may skip some, so its successor drains fully
    counted = (m0 + BMv <= p.M);
    bb ^= 1;
    idx = nidx;
    tile = xbeg + idx;
  }
#if defined(APLA_ABL_CLOCK)
  if (tid == 0) {
    apla_abl_clock_buf_nt[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - clk0;
    apla_abl_clock_buf_nt[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - rt0;
  }
#endif
}

// kernel schedule, a per-call argument (apla_gemm_nt_ex): 4 = auto, 9 = ping-pong, 14/15 = 128-wide persistent MI 4/5,
// 1 = the simple non-persistent kernel
constexpr int RESIDENT_WGS = 512;  // 256 CUs x 2 workgroups (64-80 KB LDS, <=256 VGPR)

template <int EPI, typename OutT, int MI>
int launch_persist(const GemmParams& p_in, hipStream_t stream) {
  GemmParams p = p_in;
  p.ngrp = pick_ngrp(p.tiles_n, BN, p.K);
  const int tiles_m = (p.M + 32 * MI - 1) / (32 * MI);
  const int total = tiles_m * p.tiles_n;
  const int resident = RESIDENT_WGS - 2 * (p.reserve > 0 && p.reserve < 192 ? p.reserve : 0);
  int G = total < resident ? total : resident;
  // K-loop schedule (the kernel's EXP bits): bit 0 = the first K-half's fragment reads are issued before the next stage's LDS-DMA,
  // bit 1 = s_setprio 1 around each MFMA cluster.  The two-output GELU epilogue runs with both (measured, tools/gemm_bench.py
  // GEMM_VARIANTS=0,...,7000: fc1 + GELU at M = 25216 180 -> 166 us back to back; with priority the co-resident workgroup's MFMA
  // clusters win the issue arbitration against this workgroup's epilogue VALU work instead of being spaced out by it); the other
  // epilogues do not move.  GemmParams::exp = 7 forces the plain loop (A/B), 1..3 force that schedule.
  constexpr int DEF = (EPI == APLA_EPI_GELU && MI == 5 && std::is_same<OutT, bf16>::value) ? 2 : 0;
  if constexpr (EPI == APLA_EPI_GELU && std::is_same<OutT, bf16>::value) {
    if (p.drop.rng != nullptr) {    // the two-output GELU with Mlp.drop inside (apla_gemm_nt_gelu_drop): the product schedule of this tile height
      hipLaunchKernelGGL((gemm_persist_kernel<EPI, OutT, MI, DEF, true>), dim3(G), dim3(256), 0, stream, p, tiles_m);
      APLA_CHECK_LAUNCH("apla_gemm_nt_gelu_drop");
      return APLA_OK;
    }
  }
  const int e = p.exp == 0 ? DEF : (p.exp == 7 ? 0 : p.exp);
  if constexpr (MI == 5 && std::is_same<OutT, bf16>::value && (EPI == APLA_EPI_GELU || EPI == APLA_EPI_MUL)) {
    if (e == 1) { hipLaunchKernelGGL((gemm_persist_kernel<EPI, OutT, MI, 1>), dim3(G), dim3(256), 0, stream, p, tiles_m); APLA_CHECK_LAUNCH("apla_gemm_nt"); return APLA_OK; }
    if (e == 2) { hipLaunchKernelGGL((gemm_persist_kernel<EPI, OutT, MI, 2>), dim3(G), dim3(256), 0, stream, p, tiles_m); APLA_CHECK_LAUNCH("apla_gemm_nt"); return APLA_OK; }
    if (e == 3) { hipLaunchKernelGGL((gemm_persist_kernel<EPI, OutT, MI, 3>), dim3(G), dim3(256), 0, stream, p, tiles_m); APLA_CHECK_LAUNCH("apla_gemm_nt"); return APLA_OK; }
  }
  hipLaunchKernelGGL((gemm_persist_kernel<EPI, OutT, MI>), dim3(G), dim3(256), 0, stream, p, tiles_m);
  APLA_CHECK_LAUNCH("apla_gemm_nt");
  return APLA_OK;
}

// BM = 32*MI minimising (rounds over the resident workgroups) x (tile height); ties -> larger tile (less W re-reads)
inline int pick_mi(int M, int tiles_n) {
  int best = 4;
  long best_cost = -1;
  for (int mi = 4; mi <= 5; ++mi) {
    const long tiles = (long)((M + 32 * mi - 1) / (32 * mi)) * tiles_n;
    const long rounds = (tiles + RESIDENT_WGS - 1) / RESIDENT_WGS;
    const long cost = rounds * 32 * mi;
    if (best_cost < 0 || cost < best_cost || (cost == best_cost && tiles >= RESIDENT_WGS)) { best = mi; best_cost = cost; }
  }
  return best;
}

// The schedule decision, separate from the launch so that apla_gemm_nt_kernel_name can report it (bench.py names the dominant
// kernel from the dispatch, not from a literal).  kind: 0 = simple non-persistent kernel, 1 = 4-wave persistent kernel with
// BM = 32 * mi, 2 = 8-wave ping-pong kernel.
struct Sched { int kind, mi; };
inline Sched pick_schedule(int epi, int out_dtype, int M, int N, int K, int lda, int ldw, int w_panel, int g_variant) {
  // auto (4): the 8-wave ping-pong kernel for large problems with the plain STORE epilogue, else the 128-wide persistent
  // kernel; 9 forces ping-pong wherever it is instantiated; 14/15 force the 128-wide persistent kernel with MI 4/5.
  // The GELU epilogues go to the persistent kernel although its main loop is the slower one (qkv shape: 786 vs 962 TFLOP/s):
  // its two workgroups per CU run one's epilogue under the other's main loop, and the GELU / GELU' arithmetic — a third of the
  // fc1 launch on the ping-pong kernel, whose eight waves reach their epilogues together — disappears: fc1 + GELU at
  // M = 25216: 185 -> 148 us, the forward-only GELU 147 -> 135 us (tools/gemm_bench.py, GEMM_VARIANTS=4,9,15, one process).
  // (measured limit of that rule: the two-output GELU epilogue at M = 58 496, the packed student batch of the self-supervised
  // step — 719 MB of stores per launch — runs 480 us on the persistent kernel against 423 us on the ping-pong kernel, while the
  // one-output GELU_FWD and MUL do not care: above 40 000 rows GELU goes back to the ping-pong kernel)
  // 16 forces the wide 4-wave kernel (gemm_w4.hip: 160 x 256 tiles, two workgroups per CU) wherever it is instantiated.  The
  // automatic rule gives it the plain STORE problems with a short K (qkv, proj, dproj, the patch embedding: K <= 1024) and the
  // forward-only GELU: measured back to back with W images, M = 25216 (tools/gemm_bench.py GEMM_VARIANTS=109,116): qkv 87.3 -> 85.4
  // us, proj 32.3 -> 31.4, GELU_FWD 139.9 (4-wave 128-wide kernel) -> 137.4; at K = 2304 / 3072 the ping-pong kernel stays ahead
  // (dqkv 75.5 vs 77.2, fc2 97.2 vs 100.0 us), the two-output GELU is a tie (153-155 us on either).
  const bool w4_ok = apla_gemm_w4_covers(M, N, K, lda, ldw, epi, out_dtype);
  // 17 forces the tile-alternating kernel (gemm_tp.hip: 160 x 256 tiles, one 8-wave workgroup per CU whose two wave groups swap the
  // compute and the service role per tile) wherever it is instantiated
  if (g_variant == 17 && apla_gemm_tp_covers(M, N, K, lda, ldw, epi, out_dtype, w_panel)) return {4, 0};
  // 18 forces the loader-wave form of the 4-wave kernel (gemm_lw.hip: the same 160 x 128 x 64 tiles, 4 compute + 1 or 2 loader waves)
  if (g_variant == 18 && apla_gemm_lw_covers(M, N, K, lda, ldw, epi, out_dtype, w_panel)) return {5, 5};
  // automatic: the forward-only GELU from 8192 rows (measured back to back, M = 25216, image outputs, one process: 141.4 us against
  // 149.2 us on the wide 4-wave kernel; the two-output GELU and the plain stores tie or lose there: profiles/r04_tp_*)
  if (g_variant == 4 && epi == APLA_EPI_GELU_FWD && M >= 8192 && apla_gemm_tp_covers(M, N, K, lda, ldw, epi, out_dtype, w_panel)) return {4, 0};
  if (g_variant == 16 && w4_ok) return {3, 0};
#if defined(APLA_ABL_NOW4STORE)   // (ablation build: the plain stores with a short K stay on the ping-pong kernel)
  const bool w4_auto = g_variant == 4 && w4_ok && (epi == APLA_EPI_GELU_FWD && M >= 8192);
#else
  const bool w4_auto = g_variant == 4 && w4_ok && ((epi == APLA_EPI_STORE && K <= 1024 && M >= 2048) || (epi == APLA_EPI_GELU_FWD && M >= 8192));
#endif
  if (w_panel & 12) {
    // output / second-operand image: the 4-wave persistent kernel's GELU / GELU_FWD / MUL / SwiGLU epilogues (checked by
    // gemm_nt_impl) — except the two-output GELU above 40 000 rows, which the automatic rule runs on the ping-pong kernel: its
    // line-store epilogue writes the images as well (round 3: config 3's fc2 read a row-major h, 552 us against dfc1's 474 us)
    // (round 6: at K <= 768 — the packed student batch of the self-supervised step, M = 58 496, N = 3072 — the tile-alternating kernel, whose
    // service waves run the epilogue under the partner group's products, beats the ping-pong kernel's exposed epilogue: 391 vs 432 us,
    // 437 on the 4-wave kernel; at K = 1024 (ViT-L, M = 65 792, N = 4096) the ping-pong kernel stays ahead: 693 vs 721 / 759 us)
    if (epi == APLA_EPI_GELU && M > 40000 && g_variant == 4 && K <= 768 && apla_gemm_tp_covers(M, N, K, lda, ldw, epi, out_dtype, w_panel))
      return {4, 0};
    if (epi == APLA_EPI_GELU && M > 40000 && (g_variant == 4 || g_variant == 9) &&
        apla_gemm_pp2_covers(M, N, K, lda, ldw, epi, out_dtype))
      return {2, 0};
    if (w4_auto) return {3, 0};
    return {1, 5};
  }
  if (w4_auto) return {3, 0};
  if (w_panel) return {2, 0};       // K-panel operand images exist on the 32-wide-K-step kernels only (gemm_nt_impl checked that one covers the problem)
  const bool pp2_auto = (epi == APLA_EPI_STORE) || (epi == APLA_EPI_GELU && M > 40000);
  if ((g_variant == 9 || (g_variant == 4 && pp2_auto && M >= 2048)) && apla_gemm_pp2_covers(M, N, K, lda, ldw, epi, out_dtype)) return {2, 0};
  if (g_variant >= 4) {
    // tile height: fewest (rounds x rows) over the resident workgroups — except for large problems with an epilogue that does
    // arithmetic or reads a second operand, where the taller tile is worth more than a few per cent of quantisation (the
    // self-supervised step's student fc1, M = 58 496: 128-row tiles win the round count by 2 % and ran 494 us against 368 us)
    const bool heavy = (epi == APLA_EPI_GELU || epi == APLA_EPI_GELU_FWD || epi == APLA_EPI_MUL);
    const int mi = (g_variant == 4 || g_variant == 9) ? ((heavy && M >= 8192) ? 5 : pick_mi(M, N / BN)) : g_variant - 10;
    return {1, mi == 5 ? 5 : 4};
  }
  return {0, 4};
}

template <int EPI, typename OutT>
int launch(const GemmParams& p, int g_variant, hipStream_t stream) {
  const int odt = std::is_same<OutT, float>::value ? APLA_F32 : APLA_H16;
  if (p.N % BN != 0) {       // a multiple of 64 only (gemm_nt_impl checked): the 128 x 64 instance of the simple kernel, whatever the schedule asked for
    dim3 grid(((p.M + BM - 1) / BM) * (p.N / 64)), block(256);
    hipLaunchKernelGGL((gemm_nt_kernel<EPI, OutT, 64>), grid, block, 0, stream, p);
    APLA_CHECK_LAUNCH("apla_gemm_nt");
    return APLA_OK;
  }
  Sched sc = pick_schedule(EPI, odt, p.M, p.N, p.K, (p.w_panel & 2) ? 32 : p.lda, (p.w_panel & 1) ? 32 : p.ldw, p.w_panel, g_variant);
  if (p.drop.rng != nullptr) sc = {1, (p.M >= 8192) ? 5 : pick_mi(p.M, p.N / BN)};   // dropout lives in the 4-wave persistent kernel's epilogue only
  if (sc.kind == 2) return apla_gemm_pp2_launch(p, EPI, odt, stream);
  if (sc.kind == 3) return apla_gemm_w4_launch(p, EPI, odt, stream);
  if (sc.kind == 4) return apla_gemm_tp_launch(p, EPI, odt, stream);
  if (sc.kind == 5) return apla_gemm_lw_launch(p, EPI, odt, stream);
  if (sc.kind == 1) {
    if (sc.mi == 5) return launch_persist<EPI, OutT, 5>(p, stream);
    return launch_persist<EPI, OutT, 4>(p, stream);
  }
  const int tiles_m = (p.M + BM - 1) / BM;
  dim3 grid(tiles_m * p.tiles_n), block(256);
  hipLaunchKernelGGL((gemm_nt_kernel<EPI, OutT>), grid, block, 0, stream, p);
  APLA_CHECK_LAUNCH("apla_gemm_nt");
  return APLA_OK;
}

}  // namespace

static int gemm_nt_impl(const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc, int M,
                        int N, int K, int epilogue, int out_dtype, const void* aux_in, int ld_aux_in,
                        void* aux_out, int ld_aux_out, int tag, int variant, hipStream_t stream, int w_panel = 0, int reserve = 0,
                        int exp = 0, DropArgsEw drop = DropArgsEw{nullptr, 0, 0, 0, 1.0f}) {
  APLA_REQUIRE(M > 0 && N > 0 && K > 0, "apla_gemm_nt: empty problem M=%d N=%d K=%d", M, N, K);
  APLA_REQUIRE(N % 64 == 0 && K % BK == 0, "apla_gemm_nt: need N%%64==0 and K%%64==0 (N=%d K=%d)", N, K);
  APLA_REQUIRE(N % BN == 0 || w_panel == 0, "apla_gemm_nt_ex: operand / output images need N%%128==0 (N=%d)", N);
  APLA_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ((w_panel & 2) || lda >= K) && ((w_panel & 1) || ldw >= K), "apla_gemm_nt: bad lda/ldw (%d,%d) K=%d", lda, ldw, K);
  if (w_panel & 12) {
    const bool sw = epilogue == APLA_EPI_SWIGLU || epilogue == APLA_EPI_SWIGLU_BWD;
    const bool on_pp2 = pick_schedule(epilogue, out_dtype, M, N, K, (w_panel & 2) ? 32 : lda, (w_panel & 1) ? 32 : ldw, w_panel, variant).kind >= 2;   // the two kernels with a 32-wide K-step
    APLA_REQUIRE(((w_panel & 3) == 0 || on_pp2) && out_dtype == APLA_H16 && N % 64 == 0 &&
                 (epilogue == APLA_EPI_GELU || epilogue == APLA_EPI_GELU_FWD || epilogue == APLA_EPI_MUL || sw),
                 "apla_gemm_nt_ex: an output / second-operand image needs a 16-bit GELU / GELU_FWD / MUL / SwiGLU epilogue and (on the 4-wave kernel) row-major operands");
    APLA_REQUIRE(!(w_panel & 8) || (epilogue != APLA_EPI_GELU_FWD && !sw), "apla_gemm_nt_ex: only GELU / MUL keep their second operand as an image");
  } else if (w_panel) {
    APLA_REQUIRE(pick_schedule(epilogue, out_dtype, M, N, K, (w_panel & 2) ? 32 : lda, (w_panel & 1) ? 32 : ldw, w_panel, variant).kind >= 2 &&
                 (apla_gemm_pp2_covers(M, N, K, (w_panel & 2) ? 32 : lda, (w_panel & 1) ? 32 : ldw, epilogue, out_dtype) ||
                  apla_gemm_w4_covers(M, N, K, (w_panel & 2) ? 32 : lda, (w_panel & 1) ? 32 : ldw, epilogue, out_dtype) ||
                  apla_gemm_tp_covers(M, N, K, (w_panel & 2) ? 32 : lda, (w_panel & 1) ? 32 : ldw, epilogue, out_dtype, w_panel)),
                 "apla_gemm_nt_ex: K-panel operand images need the ping-pong kernel (STORE / GELU, N %% 256 == 0, K %% 32 == 0, K >= 128): ask apla_gemm_nt_panel_ok first");
  }
  APLA_REQUIRE(apla_aligned16(A) && apla_aligned16(W) && apla_aligned16(C) && A && W && C, "apla_gemm_nt: pointers must be 16-byte aligned");
  APLA_REQUIRE(bias == nullptr || apla_aligned16(bias), "apla_gemm_nt: bias must be 16-byte aligned");
  APLA_REQUIRE((w_panel & 4) || ldc % 4 == 0, "apla_gemm_nt: ldc %% 4 != 0");
  if (w_panel & 8) { ld_aux_in = ld_aux_in ? N : 0; ld_aux_out = ld_aux_out ? N : 0; }   // not read for an image; keeps the row-major checks quiet
  GemmParams p{(const bf16*)A, lda, (const bf16*)W, ldw, bias, C, ldc, aux_in, ld_aux_in, aux_out, ld_aux_out, M, N, K, N / BN, 0, w_panel, reserve, exp,
               drop, (tag >= 0 && tag < APLA_GEMM_TAGS) ? tag : 0};
  switch (epilogue) {
    case APLA_EPI_STORE:
      APLA_REQUIRE(ldc >= N, "apla_gemm_nt: ldc < N");
      if (out_dtype == APLA_H16) return launch<APLA_EPI_STORE, bf16>(p, variant, stream);
      if (out_dtype == APLA_F32) return launch<APLA_EPI_STORE, float>(p, variant, stream);
      break;
    case APLA_EPI_GELU:
      APLA_REQUIRE(aux_out && apla_aligned16(aux_out) && ld_aux_out % 4 == 0 && ld_aux_out >= N && ldc >= N, "apla_gemm_nt[gelu]: aux_out [M,N] bf16 required");
      return launch<APLA_EPI_GELU, bf16>(p, variant, stream);
    case APLA_EPI_GELU_FWD:
      APLA_REQUIRE(ldc >= N && out_dtype == APLA_H16, "apla_gemm_nt[gelu_fwd]: 16-bit output [M,N]");
      return launch<APLA_EPI_GELU_FWD, bf16>(p, variant, stream);
    case APLA_EPI_RESIDUAL:
      APLA_REQUIRE(aux_in && apla_aligned16(aux_in) && ld_aux_in % 4 == 0 && ld_aux_in >= N && ldc >= N, "apla_gemm_nt[residual]: aux_in [M,N] required");
      if (out_dtype == APLA_H16) return launch<APLA_EPI_RESIDUAL, bf16>(p, variant, stream);
      if (out_dtype == APLA_F32) return launch<APLA_EPI_RESIDUAL, float>(p, variant, stream);
      break;
    case APLA_EPI_MUL:
      APLA_REQUIRE(aux_in && apla_aligned16(aux_in) && ld_aux_in % 4 == 0 && ld_aux_in >= N && ldc >= N, "apla_gemm_nt[mul]: aux_in [M,N] bf16 required");
      return launch<APLA_EPI_MUL, bf16>(p, variant, stream);
    case APLA_EPI_SWIGLU:
      APLA_REQUIRE(aux_out && apla_aligned16(aux_out) && ld_aux_out % 4 == 0 && ld_aux_out >= N && ((w_panel & 4) || (ldc % 2 == 0 && ldc >= N / 2)), "apla_gemm_nt[swiglu]: aux_out [M,N] bf16 required");
      return launch<APLA_EPI_SWIGLU, bf16>(p, variant, stream);
    case APLA_EPI_SWIGLU_BWD:
      APLA_REQUIRE(aux_in && apla_aligned16(aux_in) && ld_aux_in % 8 == 0 && ld_aux_in >= 2 * N && ((w_panel & 4) || (ldc % 8 == 0 && ldc >= 2 * N)), "apla_gemm_nt[swiglu_bwd]: aux_in [M,2N] bf16 required");
      return launch<APLA_EPI_SWIGLU_BWD, bf16>(p, variant, stream);
    default:
      break;
  }
  apla_set_error("apla_gemm_nt: unsupported epilogue %d / out_dtype %d", epilogue, out_dtype);
  return APLA_ENOSYS;
}

extern "C" int apla_gemm_nt(const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc, int M,
                            int N, int K, int epilogue, int out_dtype, const void* aux_in, int ld_aux_in,
                            void* aux_out, int ld_aux_out, hipStream_t stream) {
  return gemm_nt_impl(A, lda, W, ldw, bias, C, ldc, M, N, K, epilogue, out_dtype, aux_in, ld_aux_in, aux_out, ld_aux_out, 0, 4, stream);
}

// Should this problem be given K-panel operand images?  Yes where the ping-pong kernel covers it AND is what the automatic
// schedule runs it on (plain STORE from 2048 rows, the two-output GELU above 40 000 rows: see `launch`); an image passed anyway is
// accepted wherever the kernel covers the problem, and forces it.
extern "C" int apla_gemm_nt_panel_ok(int M, int N, int K, int epilogue, int out_dtype) {
  if (M <= 0 || !apla_gemm_pp2_covers(M, N, K, 32, 32, epilogue, out_dtype)) return 0;
  if (epilogue == APLA_EPI_STORE) return M >= 2048 ? 1 : 0;
  return (epilogue == APLA_EPI_GELU && M > 40000) ? 1 : 0;
}

// May (and should) this problem write its output as a K-panel image?  GELU, GELU_FWD, MUL and the SwiGLU epilogues: yes (the
// two-output GELU above 40 000 rows runs on the ping-pong kernel, whose line-store epilogue writes images since round 3).
extern "C" int apla_gemm_nt_out_image_ok(int M, int N, int K, int epilogue, int out_dtype) {
  if (M <= 0 || N % BN != 0 || K % BK != 0 || out_dtype != APLA_H16) return 0;
  if (epilogue == APLA_EPI_GELU_FWD || epilogue == APLA_EPI_MUL) return 1;
  if (epilogue == APLA_EPI_SWIGLU || epilogue == APLA_EPI_SWIGLU_BWD) return N % 64 == 0 ? 1 : 0;   // C is N/2 resp. 2N wide (ViT-g: vit.py:131-149)
  return epilogue == APLA_EPI_GELU ? 1 : 0;   // up to 40 000 rows on the 4-wave kernel, above on the ping-pong kernel (both write images)
}

// flags: bits 0-7 = profiling tag (GemmParams::tag), bits 8-15 = kernel schedule (0 = auto; see `launch`), bit 16 / 17 = W / A
// given as K-panel images, bits 18 / 19 output / second-operand image, bits 20-27 = CUs to leave free (GemmParams::reserve)
extern "C" int apla_gemm_nt_ex(const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc, int M,
                               int N, int K, int epilogue, int out_dtype, const void* aux_in, int ld_aux_in,
                               void* aux_out, int ld_aux_out, int flags, hipStream_t stream) {
  const int tag = flags & 0xff, v = (flags >> 8) & 0xff;
  APLA_REQUIRE(v == 0 || v == 1 || v == 4 || v == 9 || v == 14 || v == 15 || v == 16 || v == 17 || v == 18, "apla_gemm_nt_ex: unknown schedule %d", v);
  return gemm_nt_impl(A, lda, W, ldw, bias, C, ldc, M, N, K, epilogue, out_dtype, aux_in, ld_aux_in, aux_out, ld_aux_out, tag,
                      v == 0 ? 4 : (v == 1 ? 0 : v), stream, (flags >> 16) & 15, (flags >> 20) & 0xff, (flags >> 28) & 7);
}

// fc1 + GELU + GELU' with Mlp.drop (after the activation) inside the epilogue: include/apla_hip.h
extern "C" int apla_gemm_nt_gelu_drop(const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc, int M,
                                      int N, int K, void* aux_out, int ld_aux_out, int flags, const unsigned long long* rng,
                                      unsigned long long rng_stride, unsigned site, float p_drop, hipStream_t stream) {
  APLA_REQUIRE(rng != nullptr && p_drop >= 0.f && p_drop < 1.f, "apla_gemm_nt_gelu_drop: rng (device {seed, step}) and 0 <= p < 1 required");
  APLA_REQUIRE(N % BN == 0 && ((flags >> 16) & 3) == 0, "apla_gemm_nt_gelu_drop: N %% 128 == 0 and row-major operands (the 4-wave persistent kernel)");
  const DropArgsEw dr{rng, rng_stride, site, apla_drop_threshold(p_drop), 1.0f / (1.0f - p_drop)};
  return gemm_nt_impl(A, lda, W, ldw, bias, C, ldc, M, N, K, APLA_EPI_GELU, APLA_H16, nullptr, 0, aux_out, ld_aux_out, flags & 0xff, 4, stream,
                      (flags >> 16) & 12, (flags >> 20) & 0xff, 0, dr);
}

// Which kernel does apla_gemm_nt_ex run this problem on?  Writes e.g. "gemm_persist_kernel<GELU,bf16,5>" (the name a rocprofv3
// kernel trace shows, as tools/summarize_prof.py shortens it) into buf; same decision code as the launch.
extern "C" int apla_gemm_nt_kernel_name(int M, int N, int K, int epilogue, int out_dtype, int flags, char* buf, int buflen) {
  APLA_REQUIRE(buf && buflen >= 48 && M > 0 && N > 0 && K > 0, "apla_gemm_nt_kernel_name: bad arguments");
  static const char* const epi_names[] = {"STORE", "GELU", "RESIDUAL", "MUL", "SWIGLU", "SWIGLU_BWD", "GELU_FWD"};
  APLA_REQUIRE(epilogue >= 0 && epilogue <= 6, "apla_gemm_nt_kernel_name: unknown epilogue %d", epilogue);
  const int v = (flags >> 8) & 0xff, w_panel = (flags >> 16) & 15, reserve = (flags >> 20) & 0xff, exp = (flags >> 28) & 7;
  Sched sc = pick_schedule(epilogue, out_dtype, M, N, K, (w_panel & 2) ? 32 : K, (w_panel & 1) ? 32 : K, w_panel, v == 0 ? 4 : (v == 1 ? 0 : v));
  if (N % BN != 0) sc = {0, 2};     // the 128 x 64 instance of the simple kernel (launch())
#if defined(APLA_FP16)
  const char* ot = out_dtype == APLA_F32 ? "float" : "f16";
#else
  const char* ot = out_dtype == APLA_F32 ? "float" : "bf16";
#endif
  if (sc.kind == 4) snprintf(buf, buflen, "gemm_tp_kernel<%s,%s>", epi_names[epilogue], ot);
  else if (sc.kind == 5) snprintf(buf, buflen, "gemm_lw_kernel<%s,%s,%d>", epi_names[epilogue], ot, sc.mi);
  else if (sc.kind == 3 && apla_gemm_w4_tile_rows(M, N, epilogue, out_dtype, exp, reserve) == 128) snprintf(buf, buflen, "gemm_w4_kernel<%s,%s,128 rows>", epi_names[epilogue], ot);
  else if (sc.kind == 3) snprintf(buf, buflen, "gemm_w4_kernel<%s,%s>", epi_names[epilogue], ot);
  else if (sc.kind == 2 && apla_gemm_pp2_tile_rows(M, N, epilogue, out_dtype, exp, reserve) == 256) snprintf(buf, buflen, "gemm_pp2_kernel<%s,%s,256 rows>", epi_names[epilogue], ot);
  else if (sc.kind == 2) snprintf(buf, buflen, "gemm_pp2_kernel<%s,%s>", epi_names[epilogue], ot);
  else if (sc.kind == 1) snprintf(buf, buflen, "gemm_persist_kernel<%s,%s,%d>", epi_names[epilogue], ot, sc.mi);
  else if (sc.mi == 2) snprintf(buf, buflen, "gemm_nt_kernel<%s,%s,64>", epi_names[epilogue], ot);
  else snprintf(buf, buflen, "gemm_nt_kernel<%s,%s>", epi_names[epilogue], ot);
  return APLA_OK;
}
