// Ping-pong persistent bf16 MFMA GEMM, second generation: tile 320(M) x 256(N) x 32(K), 8 waves, one workgroup per CU.
//
// Role alternation of a ping-pong kernel (two groups of four waves, one wave of each group on every SIMD; while one group
// issues its 40 MFMAs the other issues LDS-DMA and reads fragments; one s_barrier per phase), with the tile reshaped so
// that the NON-compute phase gets cheaper than the compute phase (measured on gemm_pp: 7 LDS-DMA pieces + 18
// ds_read_b128 per prepare phase took ~1100 cycles against 640 cycles of MFMA):
//   * the groups split the tile along M (rows g*160 .. g*160+159) and every wave owns 80 x 128 outputs (5 x 8 MFMA tiles,
//     160 accumulator registers): per 40 MFMAs a wave now reads 13 fragments (was 18) and issues 5 LDS-DMA pieces (was 7);
//     the tile does 142 FLOP per staged byte (was 101).
//   * K-step 32 -> a stage is 36 KB and FOUR stages fit (144 KB): LDS-DMA runs three K-steps ahead (up to 108 KB in
//     flight per CU), which covers the 2-3 us loaded L2 latency.
// M = 25216 gives 79 row tiles: N = 768 -> 237 tiles = one round over 256 CUs at 92.6 %.
// Only the operand-free epilogues (STORE, GELU) are instantiated: with 160 accumulator registers per wave there is no
// room for residual / multiplier operands (tried: two-row asm pipeline -> spills and in-order vmcnt stalls; residual as
// accumulator init -> hipcc spills or drains vmcnt(0) inside the MFMA loop).  Those epilogues stay on gemm_nt.hip's 4-wave
// persistent kernel, whose two co-resident workgroups overlap one workgroup's epilogue with the other's main loop.
#include "gemm_common.h"

#if defined(APLA_ABL_CLOCK)  // diagnostic build (tools/build_ablations.sh CLOCK): per-workgroup clock stamps
__device__ unsigned long long apla_abl_clock_buf_pp2[512];
extern "C" int apla_abl_clock_pp2(unsigned long long* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(apla_abl_clock_buf_pp2), sizeof(apla_abl_clock_buf_pp2)); }
#endif

namespace {

constexpr int QBN = 256, QBK = 32;
constexpr int QW_BYTES = QBN * QBK * 2;  // 16 KB
constexpr int QNS = 4;
constexpr int QAHEAD = 3;

// TAG: profiling tag only (see GemmParams::tag) — the instantiations of one (EPI, OutT) are the same code under different names.
// MI: 16-row fragments per wave.  5 = the 320-row tile described above; 4 = a 256-row tile (wave tile 64 x 128, 128 accumulator
// registers, 32 pieces per K-step and no duplicates) for problems whose tile count falls just past a multiple of the CU count: the
// launch then takes the same number of rounds of tiles that are a fifth smaller (the self-supervised step's M = 58 496, N = 768:
// 549 tiles of 320 rows = 2.14 rounds -> 3; 687 tiles of 256 rows = 2.68 -> 3 rounds at 0.8 of the tile).
template <int EPI, typename OutT, int TAG = 0, int MI = 5>
__global__ __launch_bounds__(512, 2) void gemm_pp2_kernel(GemmParams p, int tiles_m) {
  using E = WideEpi<EPI, OutT>;
  constexpr int QBM = 64 * MI;                 // 320 / 256
  constexpr int QA_BYTES = QBM * QBK * 2;      // 20 / 16 KB
  constexpr int QSTG = QA_BYTES + QW_BYTES;
  constexpr int QGRP = MI;                     // LDS-DMA pieces per wave per K-step: MI = 5: 36 real (20 A + 16 W) + 4 duplicates over 8 waves
  constexpr int NPC = 4 * MI + 16;             // real pieces of a stage
  __shared__ __attribute__((aligned(16))) char smem[QNS * QSTG + 2048 + 4 * 2048];  // + two bias pieces (256 floats each) + four 2 KB epilogue line buffers
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, lw = wave & 3, wm = lw >> 1, wn = lw & 1;
  const int frow = lane & 15, fq = lane >> 4;
  const int nk = p.K / QBK;
  const int tiles_n = p.N / QBN;
  const int total = tiles_m * tiles_n;
  const int G = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int q = total >> 3, rr = total & 7;
  const int xbeg = xcd * q + (xcd < rr ? xcd : rr), xcnt = q + (xcd < rr ? 1 : 0);
  const int slots = (G >> 3) + ((G & 7) > xcd ? 1 : 0);
  if (slot >= xcnt) return;
  const int my_tiles = (xcnt - slot + slots - 1) / slots;
  const int s_total = my_tiles * nk;
  const bool has_bias = p.bias != nullptr;

  // ------------------------------------------------------------------ LDS-DMA stream
  // piece c = 5*wave + it of the combined list [A0..A19 | W0..W15 | 4 x W15 again]; a piece is 16 LDS rows x 64 B; lane i
  // fills row 16*piece + (i>>2), physical chunk i&3, from logical chunk (i&3) ^ ((-(i>>4)) & 3).
  // Addresses are (wave-uniform base in SGPRs) + (32-bit lane offset): two lane constants cover every piece of every
  // tile; only a tile that hangs over the M edge recomputes clamped row offsets (keeps VGPRs for the accumulators).
  const int srow = lane >> 2;
  const int koff = ((lane & 3) ^ ((-(srow >> 2)) & 3)) * 8;
  // waves 0-3 stream the 4 MI A pieces (c = MI * wave + it), waves 4-7 the 16 W pieces (MI = 5: + 4 duplicates of the last one)
  const bool a_wave = wave < 4;
  const unsigned wrow = (p.w_panel & 1) ? 32u : (unsigned)p.ldw;   // elements between consecutive W rows
  const size_t wkstep = (p.w_panel & 1) ? (size_t)p.N * 64 : (size_t)QBK * 2;  // bytes between consecutive K-steps of W
  const unsigned arow = (p.w_panel & 2) ? 32u : (unsigned)p.lda;   // experiment: A K-panel-major too ([K/32][M][32])
  const size_t akstep = (p.w_panel & 2) ? (size_t)p.M * 64 : (size_t)QBK * 2;
  const unsigned lane_off = a_wave ? ((unsigned)srow * arow + koff) * 2u
                                   : ((unsigned)(8 * (srow >> 2) + (srow & 3)) * wrow + koff) * 2u;
  int d_step = 0, d_k = 0, d_tile = 0, d_tm = 0, d_tn = 0, d_slot = 0;
  bool d_edge = false;
  // wave-uniform addressing: one operand base per tile (SGPR pair) + this wave's five piece offsets (fixed for the kernel)
  unsigned poff[QGRP];
#pragma unroll
  for (int it = 0; it < QGRP; ++it) {
    int c = wave * QGRP + it;
    c = c < NPC ? c : NPC - 1;
    // W piece pw = c - 4 MI fills LDS rows 16*pw + srow = (wn'=pw>>3)*128 + (j=pw&7)*16 + srow, which hold W row
    //   wn'*128 + 32*(j>>1) + 8*(srow>>2) + 4*(j&1) + (srow&3)     (MFMA order, see gemm_common.h)
    const int pw = c - 4 * MI, j = pw & 7;
    poff[it] = a_wave ? (unsigned)(c * 16) * arow * 2u
                      : (unsigned)((pw >> 3) * 128 + 32 * (j >> 1) + 4 * (j & 1)) * wrow * 2u;
  }
  const char* tbase = nullptr;
  auto dma_issue = [&]() {
    if (d_step >= s_total) return;
#if defined(APLA_ABL_NODMA)  // diagnostic build: only the first stages are streamed
    if (d_step > 8) { ++d_step; d_slot = (d_slot + 1) & (QNS - 1); if (++d_k == nk) { d_k = 0; ++d_tile; } return; }
#endif
    if (d_k == 0) {
      tile_coords(xbeg + slot + d_tile * slots, tiles_m, tiles_n, p.ngrp, d_tm, d_tn);
      d_edge = a_wave && d_tm * QBM + QBM > p.M;
      tbase = a_wave ? (const char*)(p.A + (size_t)(d_tm * QBM) * arow) : (const char*)(p.W + (size_t)(d_tn * QBN) * wrow);
    }
    char* base = smem + d_slot * QSTG + (a_wave ? wave * QGRP * 1024 : 0);
#if defined(APLA_ABL_SAMEK)  // diagnostic build: every K-step streams the k = 0 slice again (cache-resident source)
    const size_t kb = 0;
#else
    const size_t kb = a_wave ? (size_t)d_k * akstep : (size_t)d_k * wkstep;  // byte offset of this K-step
#endif
    if (!d_edge) {
#pragma unroll
      for (int it = 0; it < QGRP; ++it) {
        int c = wave * QGRP + it;
        c = c < NPC ? c : NPC - 1;
        char* dst = a_wave ? base + it * 1024 : base + c * 1024;
        __builtin_amdgcn_global_load_lds(GLBP(tbase + kb + poff[it] + lane_off), LDSP(dst), 16, 0, 0);
      }
    } else {  // A rows hang over the M edge: clamp them (reads stay inside A; those rows are never stored)
#pragma unroll
      for (int it = 0; it < QGRP; ++it) {
        const int c = wave * QGRP + it;
        int gr = d_tm * QBM + c * 16 + srow;
        gr = gr < p.M ? gr : p.M - 1;
        const unsigned off = ((unsigned)gr * arow + koff) * 2u;
        __builtin_amdgcn_global_load_lds(GLBP((const char*)p.A + kb + off), LDSP(base + it * 1024), 16, 0, 0);
      }
    }
    if (d_k == 0 && has_bias && wave == 0)
      __builtin_amdgcn_global_load_lds(GLBP(p.bias + d_tn * QBN + lane * 4), LDSP(smem + QNS * QSTG + (d_tile & 1) * 1024), 16, 0, 0);
    ++d_step;
    d_slot = (d_slot + 1) & (QNS - 1);
    if (++d_k == nk) { d_k = 0; ++d_tile; }
  };

  // ------------------------------------------------------------------ fragments / accumulators
  bf16x8 af[MI], wf[8];
  f32x4 acc[MI][8];
  const int foff = frow * 64 + ((fq ^ ((-(frow >> 2)) & 3)) << 4);
  const int a_off = (grp * (32 * MI) + wm * (16 * MI)) * 64 + foff;
  const int w_off = QA_BYTES + (wn * 128) * 64 + foff;
  int r_slot = 0;  // ring slot of the next K-step to read
  auto read_frags = [&]() {
    const char* st = smem + r_slot * QSTG;
    r_slot = (r_slot + 1) & (QNS - 1);
#if defined(APLA_ABL_NOREAD)  // diagnostic build: fragments are read once
    if (r_slot != 1 || d_step > 8) return;
#endif
#pragma unroll
    for (int i = 0; i < MI; ++i) af[i] = *(const bf16x8*)(st + a_off + i * 1024);
#pragma unroll
    for (int j = 0; j < 8; ++j) wf[j] = *(const bf16x8*)(st + w_off + j * 1024);
  };
  auto frags_landed = [&]() {  // fragment reads complete BEFORE the barrier: their stage is re-filled from the next phase on
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < MI; ++i) asm volatile("" : "+v"(af[i]));
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(wf[j]));
  };
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  zero_acc();  // accumulators are cleared here and at the end of every epilogue: the MFMAs always update in place
  auto compute = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = MFMA_F32_16x16x32_H16(wf[j], af[i], acc[i][j]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  int relaxed = 0;
  auto epilogue = [&](int ordinal) {
    const int tile = xbeg + slot + ordinal * slots;
    int tm, tn;
    tile_coords(tile, tiles_m, tiles_n, p.ngrp, tm, tn);
    const int m0 = tm * QBM + grp * (32 * MI), n0 = tn * QBN;
    const bool full = m0 + 32 * MI <= p.M;
    wide_epilogue<EPI, OutT, MI>(p, acc, (const float*)(smem + QNS * QSTG + (ordinal & 1) * 1024), smem + QNS * QSTG + 2048 + lw * 2048,
                              m0, n0, wm, wn, lane, full);
    asm volatile("" ::: "memory");
    zero_acc();
    relaxed = full ? 3 : 0;
  };
  // end of even phase 2s: K-step s+1 must have landed everywhere (group 0 reads it in phase 2s+1).  Every wave has issued
  // its shares up to K-step s+3 by now, so two younger groups may stay in flight (fewer at the very end).  A full tile's
  // epilogue issues its stores right AFTER the share of K-step s+3, so for the next three waits the stores sit behind the
  // K-step being waited for in the (in-order) vmcnt queue and may stay in flight too.
  auto end_even_phase = [&](int s) {
    if (s + 3 < s_total) {
      if (relaxed > 0) wait_vmcnt<2 * QGRP + MI * E::S>(); else wait_vmcnt<2 * QGRP>();
    } else if (s + 2 < s_total) {
      wait_vmcnt<QGRP>();
    } else {
      wait_vmcnt<0>();
    }
    --relaxed;
  };

#if defined(APLA_ABL_CLOCK)
  const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  // ------------------------------------------------------------------ prologue: K-steps 0..2 issued by everyone
  dma_issue();
  dma_issue();
  dma_issue();
  if (s_total > 2) wait_vmcnt<2 * QGRP>(); else wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  // A phase that prepares a K-step reads its fragments FIRST and does the LDS-DMA issue and the bookkeeping while those
  // reads are in flight; only at a tile boundary the order is DMA, epilogue, reads (the epilogue needs the fragment
  // registers, and its stores must queue behind the DMA share: see end_even_phase).
  int kk = 0, ord = 0;
  if (grp == 0) {
    read_frags();  // K-step 0
    dma_issue();   // share of K-step 3 (phase -1)
    frags_landed();
    __builtin_amdgcn_s_barrier();
    for (int s = 0; s < s_total; ++s) {
      compute();  // phase 2s
      end_even_phase(s);
      __builtin_amdgcn_s_barrier();
      const bool more = s + 1 < s_total;  // phase 2s+1: prepare K-step s+1, issue the share of K-step s+4
      const bool tile_end = ++kk == nk;
      if (tile_end) {
        kk = 0;
        dma_issue();
        epilogue(ord++);
      }
      if (more) read_frags();  // one call site: two would merge 52 fragment registers through PHI copies
      if (!tile_end) dma_issue();
      if (more) frags_landed();
      __builtin_amdgcn_s_barrier();
    }
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
  } else {
    __builtin_amdgcn_s_barrier();
    for (int s = 0; s < s_total; ++s) {
      const bool tile_end = kk == nk;  // phase 2s: prepare K-step s, issue the share of K-step s+3
      if (tile_end) {
        kk = 0;
        dma_issue();
        epilogue(ord++);
      }
      read_frags();
      if (!tile_end) dma_issue();
      ++kk;
      frags_landed();
      end_even_phase(s);
      __builtin_amdgcn_s_barrier();
      compute();  // phase 2s+1
      __builtin_amdgcn_s_barrier();
    }
    epilogue(ord);
    __builtin_amdgcn_s_barrier();
  }
#if defined(APLA_ABL_CLOCK)
  if (tid == 0) {
    apla_abl_clock_buf_pp2[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - clk0;
    apla_abl_clock_buf_pp2[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - rt0;
  }
#endif
}

}  // namespace

bool apla_gemm_pp2_covers(int M, int N, int K, long lda, long ldw, int epilogue, int out_dtype) {
  if (N % QBN != 0 || K % QBK != 0 || K < 4 * QBK) return false;
  if ((size_t)M * lda >= (1ull << 30) || (size_t)N * ldw >= (1ull << 30)) return false;  // 32-bit operand offsets
  if (epilogue == APLA_EPI_STORE) return out_dtype == APLA_H16 || out_dtype == APLA_F32;
  return (epilogue == APLA_EPI_GELU || epilogue == APLA_EPI_GELU_FWD) && out_dtype == APLA_H16;
}

// Tile height (bf16 STORE only): 320 rows unless 256-row tiles finish in fewer tile-rows of work — rounds over the CUs x height, with
// 4 % charged to the smaller tile for its leaner wave tile (12 fragment reads per 32 products instead of 13 per 40).
// GemmParams::exp 5 / 6 force 256 / 320 (A/B runs and tests).
int apla_gemm_pp2_tile_rows(int M, int N, int epilogue, int out_dtype, int exp, int reserve) {
  if (epilogue != APLA_EPI_STORE || out_dtype != APLA_H16 || exp == 6) return 320;
  if (exp == 5) return 256;
  const int cus = 256 - (reserve > 0 && reserve < 192 ? reserve : 0);
  auto rounds = [&](int qbm) { return ((long)((M + qbm - 1) / qbm) * (N / QBN) + cus - 1) / cus; };
  return (double)rounds(256) * 256 * 1.04 < (double)rounds(320) * 320 ? 256 : 320;
}

int apla_gemm_pp2_launch(const GemmParams& p_in, int epilogue, int out_dtype, hipStream_t stream) {
  // (operands given as K-panel images are addressed with a row pitch of 32 elements: the offset limit applies to that pitch)
  if (!apla_gemm_pp2_covers(p_in.M, p_in.N, p_in.K, (p_in.w_panel & 2) ? 32 : p_in.lda, (p_in.w_panel & 1) ? 32 : p_in.ldw, epilogue, out_dtype))
    return APLA_ENOSYS;
  GemmParams p = p_in;
  p.ngrp = pick_ngrp(p.N / QBN, QBN, p.K);
  // Round 6: with at most four column tiles (the N = 768 launches of ViT-B — fc2, dfc1, dqkv: 4.7 / 3.5 MB of W — and the N = 1024 ones
  // of ViT-L) the plain walk wins: the workgroups that share an A row tile run side by side on one XCD and the A panel is streamed ONCE,
  // where one-tile column groups streamed it three / four times (counters: 2.5-2.6 x the algorithmic bytes,
  // profiles/r06_a_bench_roofline.json).  Measured back to back with APLA_NGRP on the NGRP build (profiles/r06_b_ngrp_pp2.md): ViT-B
  // fc2 114 -> 108 us, dfc1 106 -> 104, dqkv 81.5 -> 77.5; ViT-L fc2 561 -> 533, dqkv 403 -> 387; wider outputs (qkv) do not move.
#if !defined(APLA_ABL_NGRP)
  if (p.N / QBN <= 4) p.ngrp = 0;
#endif
  const int cus = 256 - (p.reserve > 0 && p.reserve < 192 ? p.reserve : 0);
  const int QBM = apla_gemm_pp2_tile_rows(p.M, p.N, epilogue, out_dtype, p.exp, p.reserve);
  const bool small_tile = QBM == 256;
  const int tiles_m = (p.M + QBM - 1) / QBM;
  const int total = tiles_m * (p.N / QBN);
  const int G = total < cus ? total : cus;
#define PP2_LAUNCH(E, T)                                                                                 \
  do {                                                                                                   \
    hipLaunchKernelGGL((gemm_pp2_kernel<E, T>), dim3(G), dim3(512), 0, stream, p, tiles_m);              \
    hipError_t e__ = hipGetLastError();                                                                  \
    if (e__ != hipSuccess) { apla_set_error("apla_gemm_nt[pp2]: launch failed: %s", hipGetErrorString(e__)); return APLA_EIO; } \
    return APLA_OK;                                                                                      \
  } while (0)
  switch (epilogue) {
    case APLA_EPI_STORE:
      if (out_dtype == APLA_F32) PP2_LAUNCH(APLA_EPI_STORE, float);
      if (small_tile) {
        hipLaunchKernelGGL((gemm_pp2_kernel<APLA_EPI_STORE, bf16, 0, 4>), dim3(G), dim3(512), 0, stream, p, tiles_m);
        hipError_t e__ = hipGetLastError();
        if (e__ != hipSuccess) { apla_set_error("apla_gemm_nt[pp2]: launch failed: %s", hipGetErrorString(e__)); return APLA_EIO; }
        return APLA_OK;
      }
      switch (p.tag) {   // bf16 STORE: the step's six call-site shapes run through this one kernel
#define PP2_TAGGED(T)                                                                                                 \
        case T:                                                                                                        \
          hipLaunchKernelGGL((gemm_pp2_kernel<APLA_EPI_STORE, bf16, T>), dim3(G), dim3(512), 0, stream, p, tiles_m);  \
          break;
        PP2_TAGGED(2) PP2_TAGGED(3) PP2_TAGGED(4) PP2_TAGGED(5) PP2_TAGGED(6) PP2_TAGGED(7) PP2_TAGGED(8)
#undef PP2_TAGGED
        default:
          hipLaunchKernelGGL((gemm_pp2_kernel<APLA_EPI_STORE, bf16, 0>), dim3(G), dim3(512), 0, stream, p, tiles_m);
      }
      {
        hipError_t e__ = hipGetLastError();
        if (e__ != hipSuccess) { apla_set_error("apla_gemm_nt[pp2]: launch failed: %s", hipGetErrorString(e__)); return APLA_EIO; }
        return APLA_OK;
      }
    case APLA_EPI_GELU:
      PP2_LAUNCH(APLA_EPI_GELU, bf16);
    case APLA_EPI_GELU_FWD:
      PP2_LAUNCH(APLA_EPI_GELU_FWD, bf16);
    default:
      return APLA_ENOSYS;
  }
#undef PP2_LAUNCH
}
