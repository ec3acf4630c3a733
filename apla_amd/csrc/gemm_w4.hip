// 4-wave persistent bf16 MFMA GEMM with the ping-pong kernel's wave tile: tile 160(M) x 256(N) x 32(K), TWO workgroups per CU.
//
// Why a third schedule.  The heavy epilogues (GELU + GELU', the multiply by GELU') want two co-resident workgroups, so that one's
// epilogue arithmetic runs under the other's MFMAs (gemm_nt.hip's 4-wave kernel); the matrix pipes want a large wave tile, so that
// a K-step costs few LDS-DMA issues and fragment reads per MFMA (gemm_pp2.hip).  gemm_nt.hip's wave tile is 80 x 64: per 40 MFMAs
// a wave issues 9 LDS-DMA pieces and 18 fragment reads and the tile does 71 FLOP per staged byte, which puts the CU's LDS fill
// path beside its matrix pipes.  Here every wave owns 80 x 128 outputs (5 x 8 MFMA tiles, 160 accumulator registers,
// as in gemm_pp2.hip): per 40 MFMAs 6.5 pieces and 13 reads, 98 FLOP per staged byte; a 32-wide K-step keeps a stage at 26 KB, so
// a 3-stage ring (LDS-DMA two K-steps ahead) + two bias pieces are exactly 80 KB: two workgroups fill a CU's 160 KB (the epilogue's
// line buffers live in the stage that was read last).  The LDS image of a stage, the fragment
// maps and the epilogues are gemm_pp2.hip's (wide_epilogue in gemm_common.h): one workgroup here is one of its two wave groups.
#include "gemm_common.h"

#if defined(APLA_ABL_CLOCK)  // diagnostic build (tools/build_ablations.sh CLOCK, tools/gemm_clock.py): per-workgroup clock stamps
__device__ unsigned long long apla_abl_clock_buf[1024];
extern "C" int apla_abl_clock_w4(unsigned long long* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(apla_abl_clock_buf), sizeof(apla_abl_clock_buf)); }
#endif

namespace {

constexpr int VBN = 256, VBK = 32;
constexpr int VW_BYTES = VBN * VBK * 2;  // 16 KB

// VNS: stages of the ring (LDS-DMA runs VNS - 1 K-steps ahead).  3: 78 KB + two 1 KB bias pieces = exactly 80 KB, two workgroups
// fill a CU's 160 KB and the epilogue's four 2 KB line buffers live in the stage that was read last; 2: 62 KB with line buffers of
// their own (the default: measured back to back, STORE shapes ran 2-4 % slower on the deeper ring — qkv 85.4 vs 88.8 us — and the
// GELU epilogues the same on both).  PRIO: s_setprio 1 around the MFMA cluster (qkv 85.6 vs 93.8 us without).  TAG: profiling tag
// only (GemmParams::tag), as in gemm_pp2.hip.
// KSPLIT: the K axis is cut into `ksplit` equal parts and the work items are (tile, part) pairs, part fastest; part s writes its fp32
// partial tile into slice s of a workspace [ksplit][M][ldc] (p.C), summed afterwards in a fixed order (apla_gemm_nt_splitk: few
// tiles and a long K — the prototype layer's input gradient of the self-supervised step is 35 tiles x 2 048 K-steps).
// MI: 16-row fragments per wave: 5 = the 160-row tile; 4 = a 128-row tile (wave tile 64 x 128; 24 pieces per K-step, two A pieces per
// wave) for problems whose tile count falls just past a multiple of the resident workgroups (gemm_pp2.hip has the same choice).
template <int EPI, typename OutT, bool PRIO, int VNS, int TAG = 0, bool KSPLIT = false, int MI = 5>
__global__ __launch_bounds__(256, 2) void gemm_w4_kernel(GemmParams p, int tiles_m, int ksplit) {
  using E = WideEpi<EPI, OutT>;
  constexpr int VBM = 32 * MI;               // 160 / 128
  constexpr int VA_BYTES = VBM * VBK * 2;    // 10 / 8 KB
  constexpr int VSTG = VA_BYTES + VW_BYTES;
  constexpr int VGRP = MI == 5 ? 7 : 6;      // LDS-DMA issues per wave and K-step (MI = 5: 26 pieces over four waves, two duplicates)
  static_assert(MI == 5 || (MI == 4 && VNS == 2), "the 128-row tile is built for the two-stage ring");
  constexpr int VBIAS = VNS * VSTG;
  constexpr bool OWN_TBUF = VNS < 3;
  __shared__ __attribute__((aligned(16))) char smem[VBIAS + 2048 + (OWN_TBUF ? 4 * 2048 : 0)];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 15, fq = lane >> 4;
  const int nk = KSPLIT ? p.K / VBK / ksplit : p.K / VBK;
  const int tiles_n = p.N / VBN;
  const int total = KSPLIT ? tiles_m * tiles_n * ksplit : tiles_m * tiles_n;
  const int G = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int q = total >> 3, rr = total & 7;
  const int xbeg = xcd * q + (xcd < rr ? xcd : rr), xcnt = q + (xcd < rr ? 1 : 0);
  const int slots = (G >> 3) + ((G & 7) > xcd ? 1 : 0);
  if (slot >= xcnt) return;
  const int my_tiles = (xcnt - slot + slots - 1) / slots;
  const int s_total = my_tiles * nk;
  const bool has_bias = p.bias != nullptr;

  // ---- LDS-DMA stream (its own position d_*: two K-steps ahead of the products, across tile boundaries).  A piece is 16 LDS rows
  // x 64 B: lane i fills row 16*piece + (i>>2), physical chunk i&3, from logical chunk (i&3) ^ ((-(i>>4)) & 3) (gemm_pp2.hip's
  // image).  Wave w streams the W pieces 4w .. 4w+3 and the A pieces {0,1,2} {3,4,5} {6,7} {8,9}.
  const int srow = lane >> 2;
  const int koff = ((lane & 3) ^ ((-(srow >> 2)) & 3)) * 8;
  const unsigned wrow = (p.w_panel & 1) ? 32u : (unsigned)p.ldw;   // elements between consecutive W rows
  const size_t wkstep = (p.w_panel & 1) ? (size_t)p.N * 64 : (size_t)VBK * 2;  // bytes between consecutive K-steps of W
  const unsigned arow = (p.w_panel & 2) ? 32u : (unsigned)p.lda;
  const size_t akstep = (p.w_panel & 2) ? (size_t)p.M * 64 : (size_t)VBK * 2;
  const unsigned a_lane = ((unsigned)srow * arow + koff) * 2u;
  const unsigned w_lane = ((unsigned)(8 * (srow >> 2) + (srow & 3)) * wrow + koff) * 2u;
  const int a_first = MI == 4 ? 2 * wave : (wave < 2 ? 3 * wave : 6 + 2 * (wave - 2));
  int d_step = 0, d_k = 0, d_tile = 0, d_slot = 0, d_tm = 0, d_k0 = 0;
  bool d_edge = false;
  const char* a_base = nullptr;
  const char* w_base = nullptr;
  auto dma_issue = [&]() {
    if (d_step >= s_total) return;
#if defined(APLA_ABL_NODMA)  // diagnostic build: only the first stages are streamed (wrong results; timed by tools/gemm_clock.py)
    if (d_step > 4) { ++d_step; d_slot = d_slot == VNS - 1 ? 0 : d_slot + 1; if (++d_k == nk) { d_k = 0; ++d_tile; } return; }
#endif
    if (d_k == 0) {
      int tn, item = xbeg + slot + d_tile * slots;
      if constexpr (KSPLIT) { const int t = item / ksplit; d_k0 = (item - t * ksplit) * nk; item = t; }
      tile_coords(item, tiles_m, tiles_n, p.ngrp, d_tm, tn);
      d_edge = d_tm * VBM + VBM > p.M;
      a_base = (const char*)(p.A + (size_t)(d_tm * VBM) * arow);
      w_base = (const char*)(p.W + (size_t)(tn * VBN) * wrow);
      if (has_bias && (VNS >= 3 || wave == 0))   // deeper ring: every wave issues the piece (same bytes, same place), so that its waits count the same on all waves
        __builtin_amdgcn_global_load_lds(GLBP(p.bias + tn * VBN + lane * 4), LDSP(smem + VBIAS + (d_tile & 1) * 1024), 16, 0, 0);
    }
    char* base = smem + d_slot * VSTG;
#if defined(APLA_ABL_SAMEK)  // diagnostic build: every K-step streams the k = 0 slice again (an L2-resident source, same LDS-DMA count)
    const size_t ka = 0, kw = 0;
#else
    const size_t ka = (size_t)(d_k0 + d_k) * akstep, kw = (size_t)(d_k0 + d_k) * wkstep;
#endif
    // W piece pw fills LDS rows 16*pw + srow = (pw>>3)*128 + (j = pw&7)*16 + srow, which hold W row
    //   (pw>>3)*128 + 32*(j>>1) + 8*(srow>>2) + 4*(j&1) + (srow&3)        (MFMA order, see gemm_common.h)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int pw = 4 * wave + it, j = pw & 7;
      const unsigned off = (unsigned)((pw >> 3) * 128 + 32 * (j >> 1) + 4 * (j & 1)) * wrow * 2u;
      __builtin_amdgcn_global_load_lds(GLBP(w_base + kw + off + w_lane), LDSP(base + VA_BYTES + pw * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int it = 0; it < (MI == 4 ? 2 : 3); ++it) {
      int c = a_first + it;
      if (MI == 5 && wave >= 2 && it == 2) {   // waves 2 and 3 own two pieces.  The deeper ring counts its waits per wave, so there the third issue
        if (VNS < 3) continue;      // repeats the second (same bytes, same place); the two-stage ring waits for everything: no repeat
        c -= 1;
      }
      if (!d_edge) {
        __builtin_amdgcn_global_load_lds(GLBP(a_base + ka + (unsigned)(c * 16) * arow * 2u + a_lane), LDSP(base + c * 1024), 16, 0, 0);
      } else {  // A rows hang over the M edge: clamp them (reads stay inside A; those rows are never stored)
        int gr = d_tm * VBM + c * 16 + srow;
        gr = gr < p.M ? gr : p.M - 1;
        __builtin_amdgcn_global_load_lds(GLBP((const char*)p.A + ka + ((unsigned)gr * arow + koff) * 2u), LDSP(base + c * 1024), 16, 0, 0);
      }
    }
    ++d_step;
    d_slot = d_slot == VNS - 1 ? 0 : d_slot + 1;
    if (++d_k == nk) { d_k = 0; ++d_tile; }
  };

  // ---- fragments / accumulators
  bf16x8 af[MI], wf[8];
  f32x4 acc[MI][8];
  const int foff = frow * 64 + ((fq ^ ((-(frow >> 2)) & 3)) << 4);
  const int a_off = (wm * (16 * MI)) * 64 + foff;
  const int w_off = VA_BYTES + (wn * 128) * 64 + foff;
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

#if defined(APLA_ABL_CLOCK)  // diagnostic build (tools/build_ablations.sh CLOCK): the core clock this workgroup ran at
  const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
  for (int a = 0; a < VNS - 1; ++a) dma_issue();
  zero_acc();   // cleared here and after every epilogue: the MFMAs always update in place
  int cur = 0, kk = 0, ord = 0;
  int relaxed = 0;  // how many of the next waits may leave the previous tile's stores outstanding
  for (int s = 0; s < s_total; ++s) {
    // K-step s must have landed; the shares of the VNS - 2 K-steps after it (7 pieces each, 8 with a bias piece) may stay in
    // flight, and so may a full tile's epilogue stores, which were issued after the shares of the next tile's first VNS - 1 K-steps
    constexpr int YOUNGER = (VNS - 2) * VGRP;
    if (s + VNS - 2 >= s_total) wait_vmcnt<0>();
    else if (relaxed > 0) wait_vmcnt<YOUNGER + MI * E::S>();
    else if (VNS > 2 && kk == nk - 1 && has_bias) wait_vmcnt<YOUNGER + 1>();
    else wait_vmcnt<YOUNGER>();
    --relaxed;
    __builtin_amdgcn_s_barrier();
    const char* st = smem + cur * VSTG;
    dma_issue();   // K-step s + VNS - 1, into the stage that was read during K-step s-1
    asm volatile("" ::: "memory");
#if defined(APLA_ABL_NOREAD)  // diagnostic build: fragments are read for the first K-steps only
    if (s < 4)
#endif
    {
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = *(const bf16x8*)(st + a_off + i * 1024);
#pragma unroll
      for (int j = 0; j < 8; ++j) wf[j] = *(const bf16x8*)(st + w_off + j * 1024);
    }
#if defined(APLA_ABL_NOREAD)
#pragma unroll
    for (int i = 0; i < MI; ++i) asm volatile("" : "+v"(af[i]));
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(wf[j]));
#endif
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = MFMA_F32_16x16x32_H16(wf[j], af[i], acc[i][j]);
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    if (++kk == nk) {
      kk = 0;
      int tm, tn, item = xbeg + slot + ord * slots, part = 0;
      if constexpr (KSPLIT) { const int t = item / ksplit; part = item - t * ksplit; item = t; }
      tile_coords(item, tiles_m, tiles_n, p.ngrp, tm, tn);
      const int m0 = tm * VBM, n0 = tn * VBN;
      const bool full = m0 + VBM <= p.M;
      char* tbuf = smem + VBIAS + 2048 + wave * 2048;
      if constexpr (!OWN_TBUF) {
        // the stage just read is free (the K-steps in flight fill the other two): its first 8 KB are the epilogue's four line
        // buffers, once every wave has read its last fragments
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        tbuf = smem + cur * VSTG + wave * 2048;
      }
      if constexpr (KSPLIT) {
        GemmParams q = p;
        q.C = (char*)p.C + (size_t)part * p.M * p.ldc * sizeof(OutT);
        wide_epilogue<EPI, OutT, MI>(q, acc, (const float*)(smem + VBIAS + (ord & 1) * 1024), tbuf, m0, n0, wm, wn, lane, full);
      } else {
        wide_epilogue<EPI, OutT, MI>(p, acc, (const float*)(smem + VBIAS + (ord & 1) * 1024), tbuf, m0, n0, wm, wn, lane, full);
      }
      asm volatile("" ::: "memory");
      zero_acc();
      relaxed = full ? VNS - 1 : 0;
      ++ord;
    }
    cur = cur == VNS - 1 ? 0 : cur + 1;
  }
#if defined(APLA_ABL_CLOCK)
  if (tid == 0) {   // (core-clock ticks, 100 MHz ticks) of this workgroup
    apla_abl_clock_buf[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - clk0;
    apla_abl_clock_buf[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - rt0;
  }
#endif
}

}  // namespace

bool apla_gemm_w4_covers(int M, int N, int K, long lda, long ldw, int epilogue, int out_dtype) {
  if (N % VBN != 0 || K % VBK != 0 || K < 3 * VBK) return false;
  if ((size_t)M * lda >= (1ull << 30) || (size_t)N * ldw >= (1ull << 30)) return false;  // 32-bit operand offsets
  return (epilogue == APLA_EPI_GELU || epilogue == APLA_EPI_GELU_FWD || epilogue == APLA_EPI_STORE) && out_dtype == APLA_H16;
}

// Tile height (bf16 STORE only): 160 rows unless 128-row tiles finish in fewer tile-rows of work (rounds over the resident workgroups x
// height, 4 % charged to the smaller tile), as in gemm_pp2.hip; GemmParams::exp 5 / 6 force 128 / 160 (7 and 4 are this kernel's other
// A/B selectors and keep 160).
int apla_gemm_w4_tile_rows(int M, int N, int epilogue, int out_dtype, int exp, int reserve) {
  if (epilogue != APLA_EPI_STORE || out_dtype != APLA_H16 || exp == 6 || exp == 7 || exp == 4) return 160;
  if (exp == 5) return 128;
  const int resident = 512 - 2 * (reserve > 0 && reserve < 192 ? reserve : 0);
  auto rounds = [&](int vbm) { return ((long)((M + vbm - 1) / vbm) * (N / VBN) + resident - 1) / resident; };
  return (double)rounds(128) * 128 * 1.04 < (double)rounds(160) * 160 ? 128 : 160;
}

int apla_gemm_w4_launch(const GemmParams& p_in, int epilogue, int out_dtype, hipStream_t stream) {
  if (!apla_gemm_w4_covers(p_in.M, p_in.N, p_in.K, (p_in.w_panel & 2) ? 32 : p_in.lda, (p_in.w_panel & 1) ? 32 : p_in.ldw, epilogue, out_dtype))
    return APLA_ENOSYS;
  GemmParams p = p_in;
  p.ngrp = pick_ngrp(p.N / VBN, VBN, p.K);
  const int resident = 512 - 2 * (p.reserve > 0 && p.reserve < 192 ? p.reserve : 0);
  const int VBM = apla_gemm_w4_tile_rows(p.M, p.N, epilogue, out_dtype, p.exp, p.reserve);
  const bool small_tile = VBM == 128;
  const int tiles_m = (p.M + VBM - 1) / VBM;
  const int total = tiles_m * (p.N / VBN);
  const int G = total < resident ? total : resident;
  // GemmParams::exp (A/B runs, tools/gemm_bench.py): 0 = the default; 7 = no priority; 4 = the three-stage ring
#define W4_LAUNCH(...) hipLaunchKernelGGL((gemm_w4_kernel<__VA_ARGS__>), dim3(G), dim3(256), 0, stream, p, tiles_m, 1)
#define W4_AB(E)                                        \
  do {                                                  \
    if (p.exp == 7) W4_LAUNCH(E, bf16, false, 2);       \
    else if (p.exp == 4) W4_LAUNCH(E, bf16, true, 3);   \
    else W4_LAUNCH(E, bf16, true, 2);                   \
  } while (0)
  switch (epilogue) {
    case APLA_EPI_GELU: W4_AB(APLA_EPI_GELU); break;
    case APLA_EPI_GELU_FWD: W4_AB(APLA_EPI_GELU_FWD); break;
    case APLA_EPI_STORE:
      if (p.exp == 7 || p.exp == 4) { W4_AB(APLA_EPI_STORE); break; }
      if (small_tile) { W4_LAUNCH(APLA_EPI_STORE, bf16, true, 2, 0, false, 4); break; }
      switch (p.tag) {   // the step's call sites run through this one kernel under different names
        case 2: W4_LAUNCH(APLA_EPI_STORE, bf16, true, 2, 2); break;
        case 3: W4_LAUNCH(APLA_EPI_STORE, bf16, true, 2, 3); break;
        case 6: W4_LAUNCH(APLA_EPI_STORE, bf16, true, 2, 6); break;
        case 7: W4_LAUNCH(APLA_EPI_STORE, bf16, true, 2, 7); break;
        case 8: W4_LAUNCH(APLA_EPI_STORE, bf16, true, 2, 8); break;
        default: W4_LAUNCH(APLA_EPI_STORE, bf16, true, 2);
      }
      break;
    default: return APLA_ENOSYS;
  }
#undef W4_AB
#undef W4_LAUNCH
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { apla_set_error("apla_gemm_nt[w4]: launch failed: %s", hipGetErrorString(e)); return APLA_EIO; }
  return APLA_OK;
}

// ---- split-K: few tiles, long K -------------------------------------------------------------------------------------------------
namespace {
template <typename OutT>
__global__ __launch_bounds__(256) void w4_splitk_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ bias,
                                                               OutT* __restrict__ C, int ldc, int M, int N, int S) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int n4 = N / 4;
  if (t >= (long)M * n4) return;
  const int m = (int)(t / n4), n = (int)(t - (long)m * n4) * 4;
  const size_t stride = (size_t)M * N;
  const float* src = partial + (size_t)m * N + n;
  f32x4 v = *(const f32x4*)src;
  for (int s = 1; s < S; ++s) v += *(const f32x4*)(src + s * stride);   // fixed order: bitwise reproducible
  if (bias != nullptr) v += *(const f32x4*)(bias + n);
  Vec4IO<OutT>::store(C + (size_t)m * ldc + n, v);
}
// parts of the K axis: enough work items for two workgroups per CU, each part a whole number of 32-wide K-steps and at least 8 of them
inline int w4_pick_split(int M, int N, int K) {
  const long tiles = (long)((M + 160 - 1) / 160) * (N / VBN);
  int S = (int)((512 + tiles - 1) / tiles);
  S = S > 64 ? 64 : S;
  while (S > 1 && ((K / VBK) % S != 0 || K / VBK / S < 8)) --S;
  return S;
}
}  // namespace

extern "C" long apla_gemm_nt_splitk_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0 || N % VBN != 0 || K % VBK != 0) return -1;
  return (long)w4_pick_split(M, N, K) * M * N * (long)sizeof(float);
}

extern "C" int apla_gemm_nt_splitk(const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc, int M, int N,
                                   int K, int out_dtype, void* workspace, long workspace_bytes, hipStream_t stream) {
  APLA_REQUIRE(A && W && C && workspace && M > 0 && N > 0 && K > 0, "apla_gemm_nt_splitk: bad arguments");
  APLA_REQUIRE(N % VBN == 0 && K % VBK == 0 && K >= 8 * VBK, "apla_gemm_nt_splitk: need N %% 256 == 0, K %% 32 == 0, K >= 256 (N=%d K=%d)", N, K);
  APLA_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && lda >= K && ldw >= K && ldc % 4 == 0 && ldc >= N, "apla_gemm_nt_splitk: bad leading dimensions");
  APLA_REQUIRE((size_t)M * lda < (1ull << 30) && (size_t)N * ldw < (1ull << 30), "apla_gemm_nt_splitk: operand too large for 32-bit offsets");
  APLA_REQUIRE(apla_aligned16(A) && apla_aligned16(W) && apla_aligned16(C) && apla_aligned16(workspace) && (bias == nullptr || apla_aligned16(bias)),
               "apla_gemm_nt_splitk: pointers must be 16-byte aligned");
  APLA_REQUIRE(out_dtype == APLA_F32 || out_dtype == APLA_H16, "apla_gemm_nt_splitk: unsupported out_dtype %d", out_dtype);
  const int S = w4_pick_split(M, N, K);
  APLA_REQUIRE(workspace_bytes >= (long)S * M * N * (long)sizeof(float), "apla_gemm_nt_splitk: workspace too small (ask apla_gemm_nt_splitk_workspace_bytes)");
  GemmParams p{(const bf16*)A, lda, (const bf16*)W, ldw, nullptr, workspace, N, nullptr, 0, nullptr, 0, M, N, K, N / 128, 0, 0, 0, 0, DropArgsEw{nullptr, 0, 0, 0, 1.0f}, 0};
  p.ngrp = pick_ngrp(p.N / VBN, VBN, p.K);
  const int tiles_m = (M + 160 - 1) / 160;
  const long items = (long)tiles_m * (N / VBN) * S;
  const int G = items < 512 ? (int)items : 512;
  hipLaunchKernelGGL((gemm_w4_kernel<APLA_EPI_STORE, float, true, 2, 0, true>), dim3(G), dim3(256), 0, stream, p, tiles_m, S);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { apla_set_error("apla_gemm_nt_splitk: launch failed: %s", hipGetErrorString(e)); return APLA_EIO; }
  const unsigned blocks = (unsigned)(((long)M * (N / 4) + 255) / 256);
  if (out_dtype == APLA_F32) hipLaunchKernelGGL(w4_splitk_reduce_kernel<float>, dim3(blocks), dim3(256), 0, stream, (const float*)workspace, bias, (float*)C, ldc, M, N, S);
  else hipLaunchKernelGGL(w4_splitk_reduce_kernel<bf16>, dim3(blocks), dim3(256), 0, stream, (const float*)workspace, bias, (bf16*)C, ldc, M, N, S);
  e = hipGetLastError();
  if (e != hipSuccess) { apla_set_error("apla_gemm_nt_splitk[reduce]: launch failed: %s", hipGetErrorString(e)); return APLA_EIO; }
  return APLA_OK;
}
