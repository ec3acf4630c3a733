// Tile-alternating bf16 MFMA GEMM: tile 160(M) x 256(N) x 32(K), 8 waves, ONE workgroup per CU (round 4).
//
// Why a fourth schedule.  The two heavy epilogues of the step (fc1 + GELU + GELU', dfc2 * GELU') cost about as many vector
// instructions per output as the product costs matrix cycles.  The other schedules run an epilogue as ONE serial stretch of the wave
// that also feeds the matrix pipe: on the ping-pong kernel (gemm_pp2.hip) all eight waves reach it together (exposed), on the two
// 4-wave kernels (gemm_nt.hip, gemm_w4.hip) the co-resident workgroup covers it, but a wave alone on its SIMD cannot keep the pipe
// busy next to its own LDS-DMA issues (~100 cycles each, blocking the in-order wave) — MFMA busy 34 % on fc1 + GELU.
// Here the two 4-wave groups of the workgroup alternate roles per TILE instead of per K-step:
//   * the COMPUTE group (one wave per SIMD) runs a whole tile's K loop and does nothing but fragment reads and MFMAs: no LDS-DMA
//     issue, no stores, no vector arithmetic.  Fragments are software-pipelined in the wave itself (the A fragments of K-step s+1
//     are read at the top of K-step s into a second register set, W fragment j of s+1 right behind the five products that used
//     fragment j of s: 160 accumulators + 40 + 32 fragment registers).
//   * the SERVICE group (the other wave of every SIMD) meanwhile issues ALL the LDS-DMA of the ring (which runs R-1 K-steps ahead
//     and across tile boundaries) and runs the epilogue of the tile it computed in the previous period, cut into 20 slices of 8
//     values per lane, one per K-step: its vector work sits in the issue slots the partner's MFMAs leave free.
// The accumulators never move: the wave that computed a tile keeps it in registers and becomes the service wave.  One s_barrier
// per K-step for all eight waves (the barrier that publishes a landed stage); LDS-DMA completion by counted s_waitcnt vmcnt that
// leave the younger stages AND the slices' stores in flight.  Same LDS image, fragment maps, K order and epilogue arithmetic as
// gemm_pp2.hip / gemm_w4.hip: bit-identical results.
#include "gemm_common.h"

namespace {

constexpr int TBM = 160, TBN = 256, TBK = 32;
constexpr int TA_BYTES = TBM * TBK * 2;  // 10 KB
constexpr int TW_BYTES = TBN * TBK * 2;  // 16 KB
constexpr int TSTG = TA_BYTES + TW_BYTES;
constexpr int TGRP = 7;        // LDS-DMA issues per service wave and K-step (26 pieces over four waves: two duplicates)
constexpr int TSLICES = 20;    // epilogue half-slices (8 accumulator values per lane each), one per K-step: needs nk >= 22

template <int EPI> struct TpEpi {
  // store instructions of a B half-slice (every second K-step of the first 20 of a service period)
  static constexpr int S = (EPI == APLA_EPI_GELU) ? 4 : 2;
};

// Younger vector-memory operations than the stage that must have landed at the end of service K-step kk (full tile: every slice
// issues its stores): R = 5: the stage was issued two K-steps ago; R = 4: one K-step ago.
template <int R, int S> constexpr int tp_younger(int kk) {
  auto st = [](int x) { return (x >= 0 && x < TSLICES && (x & 1)) ? S : 0; };
  return R == 5 ? st(kk - 2) + TGRP + st(kk - 1) + TGRP + st(kk) : st(kk - 1) + TGRP + st(kk);
}

template <int N, int I = 0, typename F> __device__ __forceinline__ void tp_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    tp_static_for<N, I + 1>(f);
  }
}

template <int EPI, typename OutT, int R, int TAG = 0>
__global__ __launch_bounds__(512, 2) void gemm_tp_kernel(GemmParams p, int tiles_m) {
  static_assert(R == 4 || R == 5, "ring depth");
  constexpr int AHEAD = R - 1;
  constexpr int TBIAS = R * TSTG;          // four 1 KB bias pieces (tile t: slot t & 3)
  constexpr int TTBUF = TBIAS + 4096;      // eight 2 KB line buffers (one per wave) for the whole-line stores
  constexpr int S = TpEpi<EPI>::S;
  __shared__ __attribute__((aligned(16))) char smem[TTBUF + 8 * 2048];
  // the lane id is re-derived from the hardware (mbcnt) wherever a role needs it: no lane constant of the service role stays live
  // across a compute period (160 accumulators + 72 fragment registers + 2 fragment offsets there)
  // (volatile asm: hipcc must not hoist the value, or anything derived from it, out of the period loop)
  auto lane_id = []() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
  };
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int grp = wave >> 2, lw = wave & 3, wm = lw >> 1, wn = lw & 1;
  const int nk = p.K / TBK;
  const int tiles_n = p.N / TBN;
  const int total = tiles_m * tiles_n;
  const int G = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int q = total >> 3, rr = total & 7;
  const int xbeg = xcd * q + (xcd < rr ? xcd : rr), xcnt = q + (xcd < rr ? 1 : 0);
  const int slots = (G >> 3) + ((G & 7) > xcd ? 1 : 0);
  if (slot >= xcnt) return;
  const int my_tiles = (xcnt - slot + slots - 1) / slots;
  const int s_total = my_tiles * nk;
  const bool has_bias = p.bias != nullptr;

  // ---- LDS-DMA stream (service role).  Image and piece split of gemm_w4.hip: a piece is 16 LDS rows x 64 B; lane i fills row
  // 16*piece + (i>>2), physical chunk i&3, from logical chunk (i&3) ^ ((-(i>>4)) & 3).  Service wave lw streams the W pieces
  // 4lw .. 4lw+3 and the A pieces {0,1,2} {3,4,5} {6,7,7} {8,9,9}: seven issues per wave and K-step on every wave.
  const unsigned wrow = (p.w_panel & 1) ? 32u : (unsigned)p.ldw;
  const size_t wkstep = (p.w_panel & 1) ? (size_t)p.N * 64 : (size_t)TBK * 2;
  const unsigned arow = (p.w_panel & 2) ? 32u : (unsigned)p.lda;
  const size_t akstep = (p.w_panel & 2) ? (size_t)p.M * 64 : (size_t)TBK * 2;
  // lane constants of the service role: recomputed at the start of every service period from an opaque copy of the lane id, so
  // that they are not live across the compute periods (160 accumulators + 72 fragment registers there)
  int srow = 0, koff = 0, dlane = 0;
  unsigned a_lane = 0, w_lane = 0;
  auto svc_consts = [&]() {
    dlane = lane_id();
    srow = dlane >> 2;
    koff = ((dlane & 3) ^ ((-(srow >> 2)) & 3)) * 8;
    a_lane = ((unsigned)srow * arow + koff) * 2u;
    w_lane = ((unsigned)(8 * (srow >> 2) + (srow & 3)) * wrow + koff) * 2u;
  };
  const int a_first = lw < 2 ? 3 * lw : 6 + 2 * (lw - 2);
  int d_k = 0, d_tile = 0, d_tm = 0;
  bool d_edge = false;
  const char* a_base = nullptr;
  const char* w_base = nullptr;
  int ring = 0;   // ring slot of the CURRENT K-step s (= s mod R), kept by both roles
  auto dma_tile = [&](int t, bool with_bias) {   // operand bases of tile ordinal t (+ its bias piece)
    int tn;
    tile_coords(xbeg + slot + t * slots, tiles_m, tiles_n, p.ngrp, d_tm, tn);
    d_edge = d_tm * TBM + TBM > p.M;
    a_base = (const char*)(p.A + (size_t)(d_tm * TBM) * arow);
    w_base = (const char*)(p.W + (size_t)(tn * TBN) * wrow);
    if (with_bias && has_bias)   // every service wave issues the piece (same bytes, same place): the waits count the same on all
      __builtin_amdgcn_global_load_lds(GLBP(p.bias + tn * TBN + dlane * 4), LDSP(smem + TBIAS + (t & 3) * 1024), 16, 0, 0);
  };
  auto dma_stage = [&](int dslot) {   // the pieces of stage (d_tile, d_k) into ring slot dslot; advances (d_tile, d_k)
    if (d_tile >= my_tiles) return;
    if (d_k == 0) dma_tile(d_tile, true);
    char* base = smem + dslot * TSTG;
    const size_t ka = (size_t)d_k * akstep, kw = (size_t)d_k * wkstep;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int pw = 4 * lw + it, j = pw & 7;
      const unsigned off = (unsigned)((pw >> 3) * 128 + 32 * (j >> 1) + 4 * (j & 1)) * wrow * 2u;
      __builtin_amdgcn_global_load_lds(GLBP(w_base + kw + off + w_lane), LDSP(base + TA_BYTES + pw * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      int c = a_first + it;
      if (lw >= 2 && it == 2) c -= 1;   // waves 2 and 3 own two pieces: the third issue repeats the second (same bytes, same place)
      if (!d_edge) {
        __builtin_amdgcn_global_load_lds(GLBP(a_base + ka + (unsigned)(c * 16) * arow * 2u + a_lane), LDSP(base + c * 1024), 16, 0, 0);
      } else {  // A rows hang over the M edge: clamp them (reads stay inside A; those rows are never stored)
        int gr = d_tm * TBM + c * 16 + srow;
        gr = gr < p.M ? gr : p.M - 1;
        __builtin_amdgcn_global_load_lds(GLBP((const char*)p.A + ka + ((unsigned)gr * arow + koff) * 2u), LDSP(base + c * 1024), 16, 0, 0);
      }
    }
    if (++d_k == nk) { d_k = 0; ++d_tile; }
  };

  // ---- fragments / accumulators (compute role)
  bf16x8 aE[5], aO[5], wf[8];
  f32x4 acc[5][8];
  const int lane0 = lane_id();
  const int frow = lane0 & 15, fq = lane0 >> 4;
  const int foff = frow * 64 + ((fq ^ ((-(frow >> 2)) & 3)) << 4);
  const int a_off = (wm * 80) * 64 + foff;
  const int w_off = TA_BYTES + (wn * 128) * 64 + foff;
  auto load_all = [&](int rslot) {   // all 13 fragments of the stage in ring slot rslot (the first K-step of a tile)
    const char* st = smem + rslot * TSTG;
#pragma unroll
    for (int i = 0; i < 5; ++i) aE[i] = *(const bf16x8*)(st + a_off + i * 1024);
#pragma unroll
    for (int j = 0; j < 8; ++j) wf[j] = *(const bf16x8*)(st + w_off + j * 1024);
  };
  // one K-step of the compute role: products of the fragments in (cur, wf), fragments of the NEXT K-step read from `st` meanwhile
  auto kstep = [&](bf16x8 (&cur)[5], bf16x8 (&nxt)[5], const char* st) {
#pragma unroll
    for (int i = 0; i < 5; ++i) nxt[i] = *(const bf16x8*)(st + a_off + i * 1024);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int i = 0; i < 5; ++i) acc[i][j] = MFMA_F32_16x16x32_H16(wf[j], cur[i], acc[i][j]);
      wf[j] = *(const bf16x8*)(st + w_off + j * 1024);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto next_ring = [&]() { ring = ring == R - 1 ? 0 : ring + 1; };
  auto slot_after = [&](int r) { return r == R - 1 ? 0 : r + 1; };
  auto slot_before = [&](int r) { return r == 0 ? R - 1 : r - 1; };

  // ---- compute period: the K loop of one tile.  The wave was the service wave of the previous period: the stages it issued in
  // its last AHEAD-2 K-steps are waited for here, at the end of its first K-steps.
  auto compute_period = [&]() {
    __builtin_amdgcn_s_setprio(1);
    // (the accumulators are zero here — cleared by the slices — but hipcc must not know: it would peel the first K-step into
    // a copy with literal-zero accumulators whose results then travel to the loop's registers through 160 moves and scratch;
    // `first` is opaque for the same reason: a test on the loop counter gets the first iteration peeled)
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(acc[i][j]));
    int first = 1;
    asm volatile("" : "+s"(first));
    for (int kk = 0; kk < nk; kk += 2) {
      __builtin_amdgcn_s_barrier();
      kstep(aE, aO, smem + slot_after(ring) * TSTG);
      next_ring();
      if (first) { if constexpr (R == 5) wait_vmcnt<TGRP>(); else wait_vmcnt<0>(); }
      __builtin_amdgcn_s_barrier();
      kstep(aO, aE, smem + slot_after(ring) * TSTG);
      next_ring();
      if (first) { if constexpr (R == 5) wait_vmcnt<0>(); first = 0; }
    }
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- service period p: LDS-DMA of stage s + AHEAD in every K-step s of period p, the epilogue of tile p - 1 in slices
  // (has_epi), the fragments of this wave's next tile in the last K-step.
  char* tbuf = smem + TTBUF + wave * 2048;
  auto service_period = [&](int pd, bool has_epi) {
    // stage pd * nk + AHEAD belongs to tile pd (AHEAD < nk)
    svc_consts();
    d_tile = pd; d_k = AHEAD;
    if (d_tile < my_tiles) dma_tile(d_tile, false);
    int kk = 0;
    const int s0 = pd * nk;
    auto step_begin = [&]() {
      __builtin_amdgcn_s_barrier();
      dma_stage(slot_before(ring));   // stage s + AHEAD into the slot of stage s - 1 (its fragments were consumed in K-step s - 1)
      next_ring();
      asm volatile("" ::: "memory");
    };
    auto step_end = [&](auto NFULL, bool full) {
      if (s0 + kk + AHEAD >= s_total) wait_vmcnt<0>();                     // no stage was issued in this K-step: nothing may be assumed younger
      else if (full) wait_vmcnt<decltype(NFULL)::value>();
      else wait_vmcnt<(R == 5 ? 2 * TGRP : TGRP)>();                        // edge tile: its slices may have skipped stores
      ++kk;
    };
    if (has_epi) {
      const int ord = pd - 1;
      int tm, tn;
      tile_coords(xbeg + slot + ord * slots, tiles_m, tiles_n, p.ngrp, tm, tn);
      const int m0 = tm * TBM, n0 = tn * TBN;
      const bool full = m0 + TBM <= p.M;
      const float* bias_lds = (const float*)(smem + TBIAS + (ord & 3) * 1024);
      const int ln = lane_id();
      const int er = ln & 15, eq = ln >> 4;
      const int ncol = wn * 128 + eq * 8;  // + 32*u
      char* wr0 = tbuf + er * 128 + ((eq ^ (er >> 1)) << 4);
      char* wr1 = tbuf + er * 128 + (((4 + eq) ^ (er >> 1)) << 4);
      const int rrow = ln >> 3, rc = ln & 7;
      const char* rd0 = tbuf + rrow * 128 + ((rc ^ (rrow >> 1)) << 4);
      const char* rd1 = tbuf + (rrow + 8) * 128 + ((rc ^ ((rrow + 8) >> 1)) << 4);
      const int mbase = m0 + wm * 80 + rrow;
      const size_t cbase = (size_t)n0 + wn * 128 + rc * 8;
      const bool c_img = (p.w_panel & 4) != 0, x_img = (p.w_panel & 8) != 0;
      auto stage2 = [&](bf16x8 c0, bf16x8 c1, bf16x8& r0, bf16x8& r1) {
        *(bf16x8*)wr0 = c0;
        *(bf16x8*)wr1 = c1;
        r0 = *(const bf16x8*)rd0;
        r1 = *(const bf16x8*)rd1;
      };
      auto commit = [&](bf16* dst, int ld, bool img, int i, int h, bf16x8 r0, bf16x8 r1) {
        const int ma = mbase + i * 16, mb = ma + 8;
        const size_t col = cbase + h * 64;
        auto at = [&](int m) -> bf16* { return img ? dst + ((col >> 5) * (size_t)p.M + m) * 32 + (col & 31) : dst + (size_t)m * ld + col; };
        if (full || ma < p.M) *(bf16x8*)at(ma) = r0;
        if (full || mb < p.M) *(bf16x8*)at(mb) = r1;
      };
      auto add_bias = [&](f32x4& lo, f32x4& hi, int u) {
        if (has_bias) {
          lo += *(const f32x4*)(bias_lds + ncol + 32 * u);
          hi += *(const f32x4*)(bias_lds + ncol + 32 * u + 4);
        }
      };
      bf16x8 hA, gA;   // results of an A half-slice, staged and stored by the B half-slice that follows
      tp_static_for<TSLICES / 2>([&](auto KC) {
        constexpr int k = decltype(KC)::value;
        constexpr int i = k >> 1, h = k & 1;
        // ---- K-step 2k: columns 32 * (2h) .. + 31 of row block i
        step_begin();
        {
          f32x4 lo = acc[i][4 * h], hi = acc[i][4 * h + 1];
          acc[i][4 * h] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][4 * h + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
          add_bias(lo, hi, 2 * h);
          if constexpr (EPI == APLA_EPI_GELU) gelu8(lo, hi, hA, gA);
          else if constexpr (EPI == APLA_EPI_GELU_FWD) gelu8_fwd(lo, hi, hA);
          else hA = Vec8IO<bf16>::pack(lo, hi);
        }
        step_end(std::integral_constant<int, tp_younger<R, S>(2 * k)>{}, full);
        __builtin_amdgcn_sched_barrier(0);
        // ---- K-step 2k + 1: columns 32 * (2h + 1) .. + 31, then both halves leave as whole lines
        step_begin();
        {
          f32x4 lo = acc[i][4 * h + 2], hi = acc[i][4 * h + 3];
          acc[i][4 * h + 2] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][4 * h + 3] = f32x4{0.f, 0.f, 0.f, 0.f};
          add_bias(lo, hi, 2 * h + 1);
          bf16x8 hB, gB, r0, r1;
          if constexpr (EPI == APLA_EPI_GELU) gelu8(lo, hi, hB, gB);
          else if constexpr (EPI == APLA_EPI_GELU_FWD) gelu8_fwd(lo, hi, hB);
          else hB = Vec8IO<bf16>::pack(lo, hi);
          stage2(hA, hB, r0, r1);
          commit((bf16*)p.C, p.ldc, c_img, i, h, r0, r1);
          if constexpr (EPI == APLA_EPI_GELU) {
            stage2(gA, gB, r0, r1);
            commit((bf16*)p.aux_out, p.ld_aux_out, x_img, i, h, r0, r1);
          }
        }
        step_end(std::integral_constant<int, tp_younger<R, S>(2 * k + 1)>{}, full);
        __builtin_amdgcn_sched_barrier(0);
      });
      // ---- the K-steps after the slices (nk >= TSLICES + 2); the last one reads the fragments of this wave's next tile
      while (kk < nk - 1) {
        step_begin();
        if (kk < TSLICES + (R == 5 ? 2 : 1)) step_end(std::integral_constant<int, tp_younger<R, S>(TSLICES)>{}, full);
        else step_end(std::integral_constant<int, tp_younger<R, 0>(0)>{}, true);
      }
      step_begin();
      load_all(ring);   // (ring already names K-step s + 1 = the first K-step of this wave's next tile)
      step_end(std::integral_constant<int, tp_younger<R, 0>(0)>{}, true);
    } else {
      while (kk < nk - 1) {
        step_begin();
        step_end(std::integral_constant<int, tp_younger<R, 0>(0)>{}, true);
      }
      step_begin();
      load_all(ring);
      step_end(std::integral_constant<int, tp_younger<R, 0>(0)>{}, true);
    }
  };

  // ---- the last tile's epilogue (no K-steps left: no barriers, no LDS-DMA), same arithmetic as the slices
  auto final_epilogue = [&](int ord) {
    int tm, tn;
    tile_coords(xbeg + slot + ord * slots, tiles_m, tiles_n, p.ngrp, tm, tn);
    const int m0 = tm * TBM, n0 = tn * TBN;
    wide_epilogue<EPI, OutT>(p, acc, (const float*)(smem + TBIAS + (ord & 3) * 1024), tbuf, m0, n0, wm, wn, lane_id(), m0 + TBM <= p.M);
  };

  // ---- prologue: group 1 (the service group of period 0) issues the first AHEAD stages
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (grp == 1) {
    svc_consts();
    d_tile = 0; d_k = 0;
#pragma unroll
    for (int a = 0; a < AHEAD; ++a) dma_stage(a);
    wait_vmcnt<(AHEAD - 2) * TGRP>();   // stages 0 and 1 have landed
  }
  __builtin_amdgcn_s_barrier();
  if (grp == 0) load_all(0);

  // period pd: group pd & 1 computes tile pd, the other group serves it.  Both run nk barriers per period.
  int pd = 0;
  if (grp == 1) { service_period(0, false); pd = 1; }
  while (true) {
    // here: this wave computes tile pd (pd < my_tiles) or is done
    if (pd >= my_tiles) break;
    compute_period();
    ++pd;
    if (pd >= my_tiles) { final_epilogue(pd - 1); break; }
    service_period(pd, true);
    ++pd;
  }
}

}  // namespace

bool apla_gemm_tp_covers(int M, int N, int K, long lda, long ldw, int epilogue, int out_dtype) {
  if (N % TBN != 0 || K % (2 * TBK) != 0 || K / TBK < TSLICES + 2) return false;
  if ((size_t)M * lda >= (1ull << 30) || (size_t)N * ldw >= (1ull << 30)) return false;  // 32-bit operand offsets
  return (epilogue == APLA_EPI_GELU || epilogue == APLA_EPI_GELU_FWD || epilogue == APLA_EPI_STORE) && out_dtype == APLA_H16;
}

int apla_gemm_tp_launch(const GemmParams& p_in, int epilogue, int out_dtype, hipStream_t stream) {
  if (!apla_gemm_tp_covers(p_in.M, p_in.N, p_in.K, (p_in.w_panel & 2) ? 32 : p_in.lda, (p_in.w_panel & 1) ? 32 : p_in.ldw, epilogue, out_dtype))
    return APLA_ENOSYS;
  GemmParams p = p_in;
  p.ngrp = pick_ngrp(p.N / TBN, TBN, p.K);
  const int tiles_m = (p.M + TBM - 1) / TBM;
  const int total = tiles_m * (p.N / TBN);
  const int cus = 256 - (p.reserve > 0 && p.reserve < 192 ? p.reserve : 0);
  const int G = total < cus ? total : cus;
  // GemmParams::exp (A/B runs, tools/gemm_bench.py): 0 = five-stage ring; 4 = four-stage ring
#define TP_LAUNCH(...) hipLaunchKernelGGL((gemm_tp_kernel<__VA_ARGS__>), dim3(G), dim3(512), 0, stream, p, tiles_m)
#define TP_AB(E)                                  \
  do {                                            \
    if (p.exp == 4) TP_LAUNCH(E, bf16, 4);        \
    else TP_LAUNCH(E, bf16, 5);                   \
  } while (0)
  switch (epilogue) {
    case APLA_EPI_GELU: TP_AB(APLA_EPI_GELU); break;
    case APLA_EPI_GELU_FWD: TP_AB(APLA_EPI_GELU_FWD); break;
    case APLA_EPI_STORE: TP_AB(APLA_EPI_STORE); break;
    default: return APLA_ENOSYS;
  }
#undef TP_AB
#undef TP_LAUNCH
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { apla_set_error("apla_gemm_nt[tp]: launch failed: %s", hipGetErrorString(e)); return APLA_EIO; }
  return APLA_OK;
}
