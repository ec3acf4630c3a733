// Tile-alternating bf16 MFMA GEMM: tile 160(M) x 256(N) x 32(K), 8 waves, ONE workgroup per CU (round 4).
//
// Why a fourth schedule.  The two heavy epilogues of the step (fc1 + GELU + GELU', dfc2 * GELU') cost about as many vector
// instructions per output as the product costs matrix cycles.  The other schedules run an epilogue as ONE serial stretch of the wave
// that also feeds the matrix pipe: on the ping-pong kernel (gemm_pp2.hip) all eight waves reach it together (exposed), on the two
// 4-wave kernels (gemm_nt.hip, gemm_w4.hip) the co-resident workgroup covers it, but a wave alone on its SIMD cannot keep the pipe
// busy next to its own LDS-DMA issues (~100 cycles each, blocking the in-order wave) — MFMA busy 34 % on fc1 + GELU.
// Here the two 4-wave groups of the workgroup alternate roles per TILE instead of per K-step:
//   * the COMPUTE group (one wave per SIMD) runs a whole tile's K loop and does nothing but fragment reads and MFMAs: no LDS-DMA
//     issue, no stores, no vector arithmetic.  Fragments are software-pipelined in the wave itself (every fragment of K-step s+1
//     is read into its register right behind the last product of K-step s that used it: 160 accumulators + 52 fragment registers).
//   * the SERVICE group (the other wave of every SIMD) meanwhile issues ALL the LDS-DMA of the ring (which runs R-1 K-steps ahead
//     and across tile boundaries) and runs the epilogue of the tile it computed in the previous period, cut into 20 slices of 8
//     values per lane, one per K-step: its vector work sits in the issue slots the partner's MFMAs leave free.
// The accumulators never move: the wave that computed a tile keeps it in registers and becomes the service wave.  One s_barrier
// per K-step for all eight waves (the barrier that publishes a landed stage); LDS-DMA completion by counted s_waitcnt vmcnt that
// leave the younger stages AND the slices' stores in flight.  Same LDS image, fragment maps, K order and epilogue arithmetic as
// gemm_pp2.hip / gemm_w4.hip: bit-identical results.
#include "gemm_common.h"

#if defined(APLA_ABL_TPSTAMPS)  // diagnostic build (tools/build_ablations.sh TPSTAMPS, tools/tp_stamps.py): where the two roles spend their cycles
// per workgroup and wave group: [0] barrier waits as compute wave, [1] rest of the compute periods, [2] barrier waits as service
// wave, [3] LDS-DMA issue, [4] epilogue slices, [5] vmcnt waits, [6] whole run, [7] K-steps computed
__device__ unsigned long long apla_abl_tp_stamp_buf[256 * 2 * 8];
extern "C" int apla_abl_tp_stamps(unsigned long long* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(apla_abl_tp_stamp_buf), sizeof(apla_abl_tp_stamp_buf)); }
#define TP_NOW() __builtin_amdgcn_s_memtime()
#define TP_ADD(i, t0) (stamps[i] += __builtin_amdgcn_s_memtime() - (t0))
#else
#define TP_NOW() 0ull
#define TP_ADD(i, t0) ((void)(t0))
#endif

// The 160 accumulators of a wave live in the ACCUMULATOR half of the register file under LITERAL names: accumulator (i, j) of the
// 5 x 8 MFMA tiles is a[4 (8 i + j) .. + 3].  With the builtin, hipcc keeps them in ordinary VGPRs next to everything else and, at
// 256 registers, renames them between K-steps, copies them at period boundaries and spills (three formulations of this kernel's loop
// ended there); with an "a" constraint it splits the file 128 / 128.  Named literally (and clobbered once, which makes the kernel
// descriptor allocate them) they are out of the allocator's way: the kernel is capped at 96 ordinary VGPRs (amdgpu_num_vgpr) for
// the fragments and the service role, 96 + 160 = 256 = two waves per SIMD.  Audit after every edit (cdna guide 5.7 item 4):
// `.vgpr_spill_count 0`, no scratch, and no v_accvgpr_* outside these three helpers in the .s (a compiler spill into the AGPRs
// would be silent corruption; tests/test_cabi.py checks the code object's metadata).
// Hazards (inline asm is not padded by hipcc): the products read fragments that come from ds_read (hipcc inserts the waits for
// asm inputs) and accumulators last written >= 40 products or a whole service period earlier; the first reader of an accumulator
// after a compute period (v_accvgpr_read in a slice) sits behind an s_nop 15 + barrier (12 wait states needed for an 8-pass product);
// a cleared accumulator (v_accvgpr_write) is next read by a product a service period later.
#if defined(APLA_FP16)
#define TP_MFMA_OP "v_mfma_f32_16x16x32_f16"
#else
#define TP_MFMA_OP "v_mfma_f32_16x16x32_bf16"
#endif
template <int IDX> __device__ __forceinline__ void tp_mfma(const bf16x8& w, const bf16x8& a) {   // acc[IDX] += w x a
  asm volatile(TP_MFMA_OP " a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(w), "v"(a), "i"(4 * IDX), "i"(4 * IDX + 3));
}
template <int IDX> __device__ __forceinline__ f32x4 tp_acc_take() {   // returns acc[IDX] and clears it
  float x0, x1, x2, x3;
  asm volatile("v_accvgpr_read_b32 %0, a[%c4]\n\tv_accvgpr_read_b32 %1, a[%c5]\n\tv_accvgpr_read_b32 %2, a[%c6]\n\tv_accvgpr_read_b32 %3, a[%c7]\n\t"
               "v_accvgpr_write_b32 a[%c4], 0\n\tv_accvgpr_write_b32 a[%c5], 0\n\tv_accvgpr_write_b32 a[%c6], 0\n\tv_accvgpr_write_b32 a[%c7], 0"
               : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3) : "i"(4 * IDX), "i"(4 * IDX + 1), "i"(4 * IDX + 2), "i"(4 * IDX + 3));
  return f32x4{x0, x1, x2, x3};
}
template <int IDX> __device__ __forceinline__ void tp_acc_zero() {
  asm volatile("v_accvgpr_write_b32 a[%c0], 0\n\tv_accvgpr_write_b32 a[%c1], 0\n\tv_accvgpr_write_b32 a[%c2], 0\n\tv_accvgpr_write_b32 a[%c3], 0"
               ::"i"(4 * IDX), "i"(4 * IDX + 1), "i"(4 * IDX + 2), "i"(4 * IDX + 3));
}

namespace {

constexpr int TBM = 160, TBN = 256, TBK = 32;
constexpr int TA_BYTES = TBM * TBK * 2;  // 10 KB
constexpr int TW_BYTES = TBN * TBK * 2;  // 16 KB
constexpr int TSTG = TA_BYTES + TW_BYTES;
constexpr int TGRP = 7;        // LDS-DMA issues per service wave and K-step (26 pieces over four waves: two duplicates)
constexpr int AHEAD_MAX = 4;   // ring depth - 1 of the deepest instantiation (nk >= TSLICES + AHEAD_MAX: see dma_prepare)
constexpr int TSLICES = 20;    // epilogue half-slices (8 accumulator values per lane each), one per K-step: needs nk >= 22

// Store instructions of the half-slice in service K-step x of a period (x < 20).  IMG (the outputs are K-panel images: the
// accumulator layout itself covers whole lines there): one store per output in every half-slice; row-major outputs go through the
// wave's LDS line buffer in pairs of half-slices: two stores per output in every second K-step.
template <int EPI, bool IMG> constexpr int tp_stores(int x) {
  constexpr int outs = (EPI == APLA_EPI_GELU) ? 2 : 1;
  if (x < 0 || x >= TSLICES) return 0;
  return IMG ? outs : ((x & 1) ? 2 * outs : 0);
}
// Vector-memory operations younger than the LAST LDS-DMA piece of the stage that must have landed at the end of service K-step kk
// (full tile: every slice issues its stores; a K-step's last piece is issued behind its stores): all operations of the K-steps
// since — R = 5: the stage was issued two K-steps ago; R = 4: one K-step ago.
template <int R, int EPI, bool IMG> constexpr int tp_younger(int kk) {
  return R == 5 ? 2 * TGRP + tp_stores<EPI, IMG>(kk - 1) + tp_stores<EPI, IMG>(kk) : TGRP + tp_stores<EPI, IMG>(kk);
}

template <int N, int I = 0, typename F> __device__ __forceinline__ void tp_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    tp_static_for<N, I + 1>(f);
  }
}

template <int EPI, typename OutT, int R, bool IMG>
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_num_vgpr(96))) void gemm_tp_kernel(GemmParams p_arg, int tiles_m_arg) {
  static_assert(R == 4 || R == 5, "ring depth");
  constexpr int AHEAD = R - 1;
  constexpr int TBIAS = R * TSTG;          // four 1 KB bias pieces (tile t: slot t & 3)
  constexpr int TTBUF = TBIAS + 4096;      // eight 2 KB line buffers (one per wave) for the whole-line stores of row-major outputs
  __shared__ __attribute__((aligned(16))) char smem[TTBUF + 8 * 2048];
  // Register diet.  A compute period holds 160 accumulators + 72 fragment registers + 2 fragment offsets; a service K-step must
  // stay near 150 instructions (the in-order service wave issues ~one instruction per 4-5 cycles beside its partner's MFMAs, and a
  // K-step of the partner is 640 cycles).  So (a) nothing of the service role stays live across a compute period: the lane id is
  // re-derived from the hardware (mbcnt, volatile: hipcc must not hoist it), and the problem description is re-READ from the
  // kernel-argument segment through a laundered pointer at the start of every service period instead of living in ~60 SGPRs
  // (which spilled to VGPR lanes: 27 v_readlane per K-step); (b) the LDS-DMA of a K-step is seven instructions + seven M0 writes:
  // per-piece VGPR offsets and two running 64-bit operand pointers, no per-piece address arithmetic and no edge branches.
  auto lane_id = []() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
  };
  struct Args { GemmParams p; int tiles_m; };
  typedef const __attribute__((address_space(4))) Args* ArgsP;   // constant address space: scalar loads, wave-uniform values
  auto args = []() {   // the kernel arguments as they lie in memory (p_arg is the first argument, tiles_m_arg follows it)
    ArgsP a = (ArgsP)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(a));
    return a;
  };
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int grp = wave >> 2, lw = wave & 3, wm = lw >> 1, wn = lw & 1;
  const int nk = p_arg.K / TBK;
  int my_tiles, tile0, tstride;   // this workgroup's tiles: linear ids tile0 + t * tstride, t < my_tiles (XCD x owns a contiguous run)
  {
    const int tiles_n = p_arg.N / TBN;
    const int total = tiles_m_arg * tiles_n;
    const int G = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q = total >> 3, rr = total & 7;
    const int xbeg = xcd * q + (xcd < rr ? xcd : rr), xcnt = q + (xcd < rr ? 1 : 0);
    const int slots = (G >> 3) + ((G & 7) > xcd ? 1 : 0);
    if (slot >= xcnt) return;
    my_tiles = (xcnt - slot + slots - 1) / slots;
    tile0 = xbeg + slot;
    tstride = slots;
  }
  const int s_total = my_tiles * nk;
#if defined(APLA_ABL_TPSTAMPS)
  unsigned long long stamps[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long t_run = TP_NOW();
#endif
  auto coords = [&](ArgsP a, int t, int& tm, int& tn) {
    tile_coords(tile0 + t * tstride, a->tiles_m, a->p.N / TBN, a->p.ngrp, tm, tn);
  };

  // ---- LDS-DMA stream (service role).  Image and piece split of gemm_w4.hip: a piece is 16 LDS rows x 64 B; lane i fills row
  // 16*piece + (i>>2), physical chunk i&3, from logical chunk (i&3) ^ ((-(i>>4)) & 3).  Service wave lw streams the W pieces
  // 4lw .. 4lw+3 and the A pieces {0,1,2} {3,4,5} {6,7,7} {8,9,9}: seven issues per wave and K-step on every wave.
  // State of a service period (set up by svc_begin, dead outside):
  unsigned wv[4], av[3];             // per-piece lane offsets (bytes) from the running operand pointers
  const char* wp = nullptr;          // W tile base + K offset of the next stage to issue (wave-uniform)
  const char* ap = nullptr;          // A likewise (edge tiles: the row clamp is in av[])
  unsigned wkstep = 0, akstep = 0;   // bytes between consecutive K-steps of an operand
  int d_k = 0, d_tile = 0;           // the next stage to issue: K-step d_k of tile ordinal d_tile
  int lds_w = 0, lds_a = 0;          // LDS offsets of this wave's first W / A piece inside a stage
  int ring = 0;                      // ring slot of the CURRENT K-step s (= s mod R), kept by both roles
  auto new_tile = [&](ArgsP a, int t, int k, bool with_bias) {   // operand pointers / clamped A offsets of tile ordinal t at K-step k
    int tm, tn;
    coords(a, t, tm, tn);
    const auto& q = a->p;
    const unsigned wrow = (q.w_panel & 1) ? 32u : (unsigned)q.ldw, arow = (q.w_panel & 2) ? 32u : (unsigned)q.lda;
    wp = (const char*)(q.W + (size_t)(tn * TBN) * wrow) + (size_t)k * wkstep;
    ap = (const char*)(q.A + (size_t)(tm * TBM) * arow) + (size_t)k * akstep;
    const int dl = lane_id(), srow = dl >> 2;
    const int koff = ((dl & 3) ^ ((-(srow >> 2)) & 3)) * 8;
    const int last = q.M - 1 - tm * TBM;   // last valid row of the tile (rows beyond it re-read it; they are never stored)
    const int a_first = lw < 2 ? 3 * lw : 6 + 2 * (lw - 2);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int c = a_first + ((lw >= 2 && j == 2) ? 1 : j);   // waves 2 and 3 own two pieces: the third issue repeats the second
      int r = c * 16 + srow;
      r = r < last ? r : last;
      av[j] = ((unsigned)r * arow + koff) * 2u;
    }
    if (with_bias && q.bias != nullptr)   // every service wave issues the piece (same bytes, same place): the waits count the same on all
      __builtin_amdgcn_global_load_lds(GLBP(q.bias + tn * TBN + dl * 4), LDSP(smem + TBIAS + (t & 3) * 1024), 16, 0, 0);
  };
  auto svc_begin = [&](ArgsP a, int t, int k) {   // service state for a period whose first stage to issue is (tile t, K-step k)
    const auto& q = a->p;
    const unsigned wrow = (q.w_panel & 1) ? 32u : (unsigned)q.ldw;
    wkstep = (q.w_panel & 1) ? (unsigned)q.N * 64u : (unsigned)TBK * 2u;
    akstep = (q.w_panel & 2) ? (unsigned)q.M * 64u : (unsigned)TBK * 2u;
    const int dl = lane_id(), srow = dl >> 2;
    const int koff = ((dl & 3) ^ ((-(srow >> 2)) & 3)) * 8;
    const unsigned w_lane = ((unsigned)(8 * (srow >> 2) + (srow & 3)) * wrow + koff) * 2u;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      // W piece pw fills LDS rows 16*pw + srow = (pw>>3)*128 + (j = pw&7)*16 + srow, which hold W row
      //   (pw>>3)*128 + 32*(j>>1) + 8*(srow>>2) + 4*(j&1) + (srow&3)        (MFMA order, see gemm_common.h)
      const int pw = 4 * lw + it, j = pw & 7;
      wv[it] = (unsigned)((pw >> 3) * 128 + 32 * (j >> 1) + 4 * (j & 1)) * wrow * 2u + w_lane;
    }
    lds_w = TA_BYTES + 4 * lw * 1024;
    lds_a = (lw < 2 ? 3 * lw : 6 + 2 * (lw - 2)) * 1024;
    d_tile = t; d_k = k;
    if (d_tile < my_tiles) new_tile(a, t, k, k == 0);
  };
  // One K-step's share of the stream = dma_prepare + seven dma_piece calls + dma_advance, which the service role spreads over its
  // K-step between chunks of epilogue arithmetic: issued back to back by four waves at once, the 28 pieces queue up in the CU's one
  // address path and every issue blocks its in-order wave for ~100 cycles (tools/tp_stamps.py).
  int cur_lds = 0;
  bool cur_issue = false;
  // (a service period changes tile at its K-step nk - AHEAD >= TSLICES, nk >= 24: the slices' K-steps never do, and their 20 unrolled
  // copies stay free of the tile walk's code — SAME_TILE)
  auto dma_prepare = [&](ArgsP a, int dslot, auto SAME_TILE) {   // stage (d_tile, d_k) goes into ring slot dslot
    cur_issue = d_tile < my_tiles;
    if constexpr (!decltype(SAME_TILE)::value) { if (cur_issue && d_k == 0) new_tile(a, d_tile, 0, true); }
    cur_lds = dslot * TSTG;
  };
  auto dma_piece = [&](auto IT) {       // pieces 0-3: W, 4-6: A
    constexpr int it = decltype(IT)::value;
    if constexpr (it < 4) {
      __builtin_amdgcn_global_load_lds(GLBP(wp + wv[it]), LDSP(smem + cur_lds + lds_w + it * 1024), 16, 0, 0);
    } else {
      const int dup = (lw >= 2 && it == 6) ? 1024 : 0;
      __builtin_amdgcn_global_load_lds(GLBP(ap + av[it - 4]), LDSP(smem + cur_lds + lds_a + (it - 4) * 1024 - dup), 16, 0, 0);
    }
  };
  auto dma_range = [&](auto LO, auto HI) {   // pieces LO .. HI-1, pinned where they stand; HI == 7 ends the K-step's share
    if (cur_issue) {
      tp_static_for<decltype(HI)::value - decltype(LO)::value>([&](auto I) { dma_piece(std::integral_constant<int, decltype(LO)::value + decltype(I)::value>{}); });
      if constexpr (decltype(HI)::value == 7) {
        wp += wkstep; ap += akstep;
        if (++d_k == nk) { d_k = 0; ++d_tile; }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- fragments / accumulators (compute role)
  bf16x8 aE[5], wf[8];
  asm volatile("" ::: "a0", "a159");   // the accumulators a[0:159] belong to tp_mfma / tp_acc_take / tp_acc_zero (kernel descriptor: 160 AGPRs)
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
  // One K-step of the compute role: the products of the fragments in (af, wf); the fragments of the NEXT K-step are read from `st`
  // meanwhile into the same registers — W fragment j right behind the five products that used it, the five A fragments inside
  // the last group (A fragment i behind product (i, 7)): 160 accumulators + 52 fragment registers, no second set (a second A
  // set, 20 registers more, put hipcc's allocation on the edge: accumulators moved between K-steps and spilled).
  auto kstep = [&](const char* st) {
    tp_static_for<7>([&](auto J) {
      constexpr int j = decltype(J)::value;
      tp_static_for<5>([&](auto I) { tp_mfma<8 * decltype(I)::value + j>(wf[j], aE[decltype(I)::value]); });
      wf[j] = *(const bf16x8*)(st + w_off + j * 1024);
      __builtin_amdgcn_sched_barrier(0);
    });
    tp_static_for<5>([&](auto I) {
      constexpr int i = decltype(I)::value;
      tp_mfma<8 * i + 7>(wf[7], aE[i]);
      aE[i] = *(const bf16x8*)(st + a_off + i * 1024);
      __builtin_amdgcn_sched_barrier(0);
    });
    wf[7] = *(const bf16x8*)(st + w_off + 7 * 1024);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto next_ring = [&]() { ring = ring == R - 1 ? 0 : ring + 1; };
  auto slot_after = [&](int r) { return r == R - 1 ? 0 : r + 1; };
  auto slot_before = [&](int r) { return r == 0 ? R - 1 : r - 1; };

  // ---- compute period: the K loop of one tile.  The wave was the service wave of the previous period: the stages it issued in
  // its last AHEAD-2 K-steps are waited for here, at the end of its first K-steps.
  auto compute_period = [&]() {
    __builtin_amdgcn_s_setprio(1);
    // (`first` is opaque: a test on the loop counter gets the first iteration peeled)
    int first = 1;
    asm volatile("" : "+s"(first));
    [[maybe_unused]] const unsigned long long tc0 = TP_NOW();
    for (int kk = 0; kk < nk; kk += 2) {
      { [[maybe_unused]] const unsigned long long t0 = TP_NOW(); __builtin_amdgcn_s_barrier(); TP_ADD(0, t0); }
      kstep(smem + slot_after(ring) * TSTG);
      next_ring();
      if (first) { if constexpr (R == 5) wait_vmcnt<TGRP>(); else wait_vmcnt<0>(); }
      { [[maybe_unused]] const unsigned long long t0 = TP_NOW(); __builtin_amdgcn_s_barrier(); TP_ADD(0, t0); }
      kstep(smem + slot_after(ring) * TSTG);
      next_ring();
      if (first) { if constexpr (R == 5) wait_vmcnt<0>(); first = 0; }
    }
    TP_ADD(1, tc0);
#if defined(APLA_ABL_TPSTAMPS)
    stamps[7] += nk;
#endif
    asm volatile("s_nop 15" ::: "memory");   // the last products' results before any reader (see TP_MFMA)
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- service period p: LDS-DMA of stage s + AHEAD in every K-step s of period p, the epilogue of tile p - 1 in slices
  // (has_epi), the fragments of this wave's next tile in the last K-step.
  using I0 = std::integral_constant<int, 0>; using I2 = std::integral_constant<int, 2>; using I4 = std::integral_constant<int, 4>;
  using I6 = std::integral_constant<int, 6>; using I7 = std::integral_constant<int, 7>;
  // live = false: the epilogue of this workgroup's LAST tile, after the last period: the same slices without K-steps around them
  // (no barrier, no LDS-DMA, no wait; one code path: the slices are ~30 KB of instructions)
  auto service_period = [&](int pd, bool has_epi, bool live) {
    ArgsP a = args();
    const int stagger = a->p.exp;
    if (live) svc_begin(a, pd, AHEAD);   // stage pd * nk + AHEAD belongs to tile pd (AHEAD < nk)
    else cur_issue = false;
    int kk = 0;
    const int s_left = s_total - pd * nk - AHEAD;   // K-steps of this period in which a stage is still issued
    [[maybe_unused]] unsigned long long t_sl = 0;
    auto step_begin = [&](auto SAME_TILE) {   // barrier + bookkeeping; the seven pieces follow (dma_range)
      if (!live) return;
      { [[maybe_unused]] const unsigned long long t0 = TP_NOW(); __builtin_amdgcn_s_barrier(); TP_ADD(2, t0); }
      // the four service waves leave the barrier together and would issue their LDS-DMA pieces at the same instants of the K-step:
      // the CU's one address path takes a piece per ~16 cycles and every issue then blocks its wave for 4 x that.  Wave lw starts
      // its K-step lw * stagger * 16 cycles late (kept through the K-step: all four run the same instructions).
      for (int z = 0; z < lw * stagger; ++z) asm volatile("s_nop 15");
      dma_prepare(a, slot_before(ring), SAME_TILE);   // stage s + AHEAD into the slot of stage s - 1 (its fragments were consumed in K-step s - 1)
      next_ring();
      asm volatile("" ::: "memory");
      t_sl = TP_NOW();
    };
    auto step_end = [&](auto NFULL, bool full) {
      asm volatile("" ::: "memory");
      if (!live) return;
      TP_ADD(4, t_sl);
      [[maybe_unused]] const unsigned long long t0 = TP_NOW();
      if (kk >= s_left) wait_vmcnt<0>();                                    // no stage was issued in this K-step: nothing may be assumed younger
      else if (full) wait_vmcnt<decltype(NFULL)::value>();
      else wait_vmcnt<(R == 5 ? 2 * TGRP : TGRP)>();                        // edge tile: its slices may have skipped stores
      TP_ADD(5, t0);
      ++kk;
    };
    auto plain_step = [&](bool last) {
      step_begin(std::false_type{});
      dma_range(I0{}, I7{});
      if (last) load_all(ring);   // (ring already names K-step s + 1 = the first K-step of this wave's next tile)
      step_end(std::integral_constant<int, tp_younger<R, APLA_EPI_STORE, false>(TSLICES + 2)>{}, true);
    };
    if (has_epi) {
      const auto& q = a->p;
      const int ord = pd - 1;
      int tm, tn;
      coords(a, ord, tm, tn);
      const int m0 = tm * TBM, n0 = tn * TBN;
      const int Mrows = q.M;
      const bool full = m0 + TBM <= Mrows;
      const bool has_bias = q.bias != nullptr;
      const float* bias_lds = (const float*)(smem + TBIAS + (ord & 3) * 1024);
      const int ln = lane_id();
      const int er = ln & 15, eq = ln >> 4;
      const int ncol = wn * 128 + eq * 8;  // + 32*u
      auto add_bias = [&](f32x4& lo, f32x4& hi, int u) {
        if (has_bias) {
          lo += *(const f32x4*)(bias_lds + ncol + 32 * u);
          hi += *(const f32x4*)(bias_lds + ncol + 32 * u + 4);
        }
      };
      // GELU (and GELU') of 8 accumulator values in two halves of four (gelu8 / gelu8_fwd of gemm_common.h, same arithmetic and the
      // same pinning), `mid` between them: the service role's LDS-DMA pieces are spread between chunks of vector work
      typedef bf16 bf16x2_t __attribute__((ext_vector_type(2)));
      typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
      auto act8 = [&](f32x4 lo, f32x4 hi, bf16x8& h, bf16x8& g, auto&& mid) {
        if constexpr (EPI == APLA_EPI_STORE) {
          mid();
          h = Vec8IO<bf16>::pack(lo, hi);
        } else {
          unsigned hp[4], gp[4] = {0, 0, 0, 0};
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            const f32x4 x0 = half ? hi : lo;
            f32x4 x, y;
            if constexpr (EPI == APLA_EPI_GELU) gelu_and_grad4(x0, x, y); else x = gelu_only4(x0);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              bf16x2_t hh, gg;
              hh[0] = (bf16)x[2 * e]; hh[1] = (bf16)x[2 * e + 1];
              hp[2 * half + e] = __builtin_bit_cast(unsigned, hh);
              if constexpr (EPI == APLA_EPI_GELU) {
                gg[0] = (bf16)y[2 * e]; gg[1] = (bf16)y[2 * e + 1];
                gp[2 * half + e] = __builtin_bit_cast(unsigned, gg);
              }
            }
            if constexpr (EPI == APLA_EPI_GELU) asm volatile("" : "+v"(hp[2 * half]), "+v"(gp[2 * half]), "+v"(hp[2 * half + 1]), "+v"(gp[2 * half + 1]));
            else asm volatile("" : "+v"(hp[2 * half]), "+v"(hp[2 * half + 1]));
            if (half == 0) mid();
          }
          h = __builtin_bit_cast(bf16x8, u32x4_t{hp[0], hp[1], hp[2], hp[3]});
          if constexpr (EPI == APLA_EPI_GELU) g = __builtin_bit_cast(bf16x8, u32x4_t{gp[0], gp[1], gp[2], gp[3]});
        }
      };
      if constexpr (IMG) {
        // Outputs as K-panel images [N/32][M][32]: a lane's 8 columns are 16 contiguous bytes and the 16 rows of a fragment 1 KB:
        // the accumulator layout stores whole lines, no LDS transpose.  Address = (wave-uniform base of the tile's first panel
        // and row block, 64-bit) + (panel u, row block i: a scalar) + one lane offset that never changes.
        const unsigned voff = (unsigned)(er * 64 + eq * 16);
        const int row_lane = m0 + wm * 80 + er;   // + i * 16
        const size_t first = ((size_t)((n0 >> 5) + wn * 4) * (size_t)Mrows + (size_t)(m0 + wm * 80)) * 64;
        char* const cb = (char*)q.C + first;
        char* const xb = (EPI == APLA_EPI_GELU) ? (char*)q.aux_out + first : nullptr;
        const unsigned m64 = (unsigned)Mrows * 64u;   // bytes between consecutive panels
        tp_static_for<TSLICES>([&](auto KC) {
          constexpr int k = decltype(KC)::value;
          constexpr int i = k >> 2, u = k & 3;   // row block i, column pair u: columns 32 u .. 32 u + 31 of the wave's 128
          step_begin(std::true_type{});
          dma_range(I0{}, I2{});
          f32x4 lo = tp_acc_take<8 * i + 2 * u>(), hi = tp_acc_take<8 * i + 2 * u + 1>();   // (cleared: the products always accumulate in place)
          add_bias(lo, hi, u);
          bf16x8 hv, gv;
          act8(lo, hi, hv, gv, [&]() { __builtin_amdgcn_sched_barrier(0); dma_range(I2{}, I4{}); });
          __builtin_amdgcn_sched_barrier(0);
          dma_range(I4{}, I6{});
          const size_t so = (size_t)u * m64 + (size_t)(i * 1024);
          if (full || row_lane + i * 16 < Mrows) {
            *(bf16x8*)(cb + so + voff) = hv;
            if constexpr (EPI == APLA_EPI_GELU) *(bf16x8*)(xb + so + voff) = gv;
          }
          asm volatile("" ::: "memory");
          dma_range(I6{}, I7{});   // the K-step's last piece behind its stores (tp_younger)
          step_end(std::integral_constant<int, tp_younger<R, EPI, true>(k)>{}, full);
          __builtin_amdgcn_sched_barrier(0);
        });
      } else {
        char* tbuf = smem + TTBUF + wave * 2048;
        char* wr0 = tbuf + er * 128 + ((eq ^ (er >> 1)) << 4);
        char* wr1 = tbuf + er * 128 + (((4 + eq) ^ (er >> 1)) << 4);
        const int rrow = ln >> 3, rc = ln & 7;
        const char* rd0 = tbuf + rrow * 128 + ((rc ^ (rrow >> 1)) << 4);
        const char* rd1 = tbuf + (rrow + 8) * 128 + ((rc ^ ((rrow + 8) >> 1)) << 4);
        // row-major outputs: address = (wave-uniform base of the tile's first row / column of this wave, 64-bit) + (row block i,
        // column half h: a scalar) + a lane offset per output that never changes
        const int row_lane = m0 + wm * 80 + rrow;   // + i * 16 (+ 8)
        const int ldc = q.ldc, ldx = (EPI == APLA_EPI_GELU) ? q.ld_aux_out : 0;
        const unsigned voc = (unsigned)(rrow * ldc + rc * 8) * 2u, vox = (unsigned)(rrow * ldx + rc * 8) * 2u;
        char* const cb = (char*)q.C + ((size_t)(m0 + wm * 80) * (size_t)ldc + (size_t)(n0 + wn * 128)) * 2;
        char* const xb = (EPI == APLA_EPI_GELU) ? (char*)q.aux_out + ((size_t)(m0 + wm * 80) * (size_t)ldx + (size_t)(n0 + wn * 128)) * 2 : nullptr;
        auto stage2 = [&](bf16x8 c0, bf16x8 c1, bf16x8& r0, bf16x8& r1) {
          *(bf16x8*)wr0 = c0;
          *(bf16x8*)wr1 = c1;
          r0 = *(const bf16x8*)rd0;
          r1 = *(const bf16x8*)rd1;
        };
        auto commit = [&](char* base, int ld, unsigned voff, int i, int h, bf16x8 r0, bf16x8 r1) {
          const size_t so = ((size_t)(i * 16) * (size_t)ld + (size_t)(h * 64)) * 2;
          if (full || row_lane + i * 16 < Mrows) *(bf16x8*)(base + so + voff) = r0;
          if (full || row_lane + i * 16 + 8 < Mrows) *(bf16x8*)(base + so + (size_t)ld * 16 + voff) = r1;
        };
        bf16x8 hA, gA;   // results of an A half-slice, staged and stored by the B half-slice that follows
        tp_static_for<TSLICES / 2>([&](auto KC) {
          constexpr int k = decltype(KC)::value;
          constexpr int i = k >> 1, h = k & 1;
          // ---- K-step 2k: columns 32 * (2h) .. + 31 of row block i
          step_begin(std::true_type{});
          dma_range(I0{}, I2{});
          {
            f32x4 lo = tp_acc_take<8 * i + 4 * h>(), hi = tp_acc_take<8 * i + 4 * h + 1>();   // (cleared: the products always accumulate in place)
            add_bias(lo, hi, 2 * h);
            act8(lo, hi, hA, gA, [&]() { __builtin_amdgcn_sched_barrier(0); dma_range(I2{}, I4{}); });
          }
          __builtin_amdgcn_sched_barrier(0);
          dma_range(I4{}, I7{});
          step_end(std::integral_constant<int, tp_younger<R, EPI, false>(2 * k)>{}, full);
          __builtin_amdgcn_sched_barrier(0);
          // ---- K-step 2k + 1: columns 32 * (2h + 1) .. + 31, then both halves leave as whole lines
          step_begin(std::true_type{});
          dma_range(I0{}, I2{});
          {
            f32x4 lo = tp_acc_take<8 * i + 4 * h + 2>(), hi = tp_acc_take<8 * i + 4 * h + 3>();
            add_bias(lo, hi, 2 * h + 1);
            bf16x8 hB, gB, r0, r1;
            act8(lo, hi, hB, gB, [&]() { __builtin_amdgcn_sched_barrier(0); dma_range(I2{}, I4{}); });
            __builtin_amdgcn_sched_barrier(0);
            stage2(hA, hB, r0, r1);
            dma_range(I4{}, I6{});
            commit(cb, ldc, voc, i, h, r0, r1);
            if constexpr (EPI == APLA_EPI_GELU) {
              stage2(gA, gB, r0, r1);
              commit(xb, ldx, vox, i, h, r0, r1);
            }
          }
          asm volatile("" ::: "memory");
          dma_range(I6{}, I7{});   // the K-step's last piece behind its stores (tp_younger)
          step_end(std::integral_constant<int, tp_younger<R, EPI, false>(2 * k + 1)>{}, full);
          __builtin_amdgcn_sched_barrier(0);
        });
      }
      // ---- the K-steps after the slices (nk >= TSLICES + 2); the last one reads the fragments of this wave's next tile
      if (!live) return;
      while (kk < nk - 1) {
        step_begin(std::false_type{});
        dma_range(I0{}, I7{});
        if (kk == TSLICES) step_end(std::integral_constant<int, tp_younger<R, EPI, IMG>(TSLICES)>{}, full);
        else step_end(std::integral_constant<int, tp_younger<R, APLA_EPI_STORE, false>(TSLICES + 2)>{}, true);
      }
      plain_step(true);
    } else {
      while (kk < nk - 1) plain_step(false);
      plain_step(true);
    }
  };

  // ---- prologue: group 1 (the service group of period 0) issues the first AHEAD stages
  tp_static_for<40>([&](auto I) { tp_acc_zero<decltype(I)::value>(); });
  if (grp == 1) {
    ArgsP a = args();
    svc_begin(a, 0, 0);
#pragma unroll
    for (int s = 0; s < AHEAD; ++s) { dma_prepare(a, s, std::true_type{}); dma_range(I0{}, I7{}); }
    wait_vmcnt<(AHEAD - 2) * TGRP>();   // stages 0 and 1 have landed
  }
  __builtin_amdgcn_s_barrier();
  if (grp == 0) load_all(0);

  // period pd: group pd & 1 computes tile pd, the other group serves it.  Both run nk barriers per period; the group that computed
  // the last tile runs its epilogue afterwards on its own.
  int pd = grp;
  if (grp == 1) service_period(0, false, true);
  while (pd < my_tiles) {
    compute_period();   // tile pd
    ++pd;
    service_period(pd, true, pd < my_tiles);
    ++pd;
  }
#if defined(APLA_ABL_TPSTAMPS)
  stamps[6] = TP_NOW() - t_run;
  if (lane_id() == 0 && lw == 0 && blockIdx.x < 256)
    for (int i = 0; i < 8; ++i) apla_abl_tp_stamp_buf[(blockIdx.x * 2 + grp) * 8 + i] = stamps[i];
#endif
}

}  // namespace

// w_panel: GemmParams::w_panel (bits 2 / 3: output / second-operand image).  The two-output GELU exists for image outputs only (the
// training step's form; with row-major outputs its line-buffer path needs more than the 96 registers left beside the accumulators)
bool apla_gemm_tp_covers(int M, int N, int K, long lda, long ldw, int epilogue, int out_dtype, int w_panel) {
  const bool img = (w_panel & 4) && (epilogue != APLA_EPI_GELU || (w_panel & 8));
  if (!img && ((w_panel & 12) || epilogue == APLA_EPI_GELU)) return false;
  if (N % TBN != 0 || K % (2 * TBK) != 0 || K / TBK < TSLICES + AHEAD_MAX) return false;
  if ((size_t)M * lda >= (1ull << 30) || (size_t)N * ldw >= (1ull << 30)) return false;  // 32-bit operand offsets
  return (epilogue == APLA_EPI_GELU || epilogue == APLA_EPI_GELU_FWD || epilogue == APLA_EPI_STORE) && out_dtype == APLA_H16;
}

int apla_gemm_tp_launch(const GemmParams& p_in, int epilogue, int out_dtype, hipStream_t stream) {
  if (!apla_gemm_tp_covers(p_in.M, p_in.N, p_in.K, (p_in.w_panel & 2) ? 32 : p_in.lda, (p_in.w_panel & 1) ? 32 : p_in.ldw, epilogue, out_dtype, p_in.w_panel))
    return APLA_ENOSYS;
  GemmParams p = p_in;
  p.ngrp = pick_ngrp(p.N / TBN, TBN, p.K);
  const int tiles_m = (p.M + TBM - 1) / TBM;
  const int total = tiles_m * (p.N / TBN);
  const int cus = 256 - (p.reserve > 0 && p.reserve < 192 ? p.reserve : 0);
  const int G = total < cus ? total : cus;
  // image outputs (every output of the epilogue an image) take the store path without LDS transposes; row-major outputs go through the
  // waves' line buffers; one of each is not instantiated
  const bool img = (p.w_panel & 4) != 0;
#define TP_LAUNCH(...) hipLaunchKernelGGL((gemm_tp_kernel<__VA_ARGS__>), dim3(G), dim3(512), 0, stream, p, tiles_m)
  switch (epilogue) {
    case APLA_EPI_GELU: TP_LAUNCH(APLA_EPI_GELU, bf16, 5, true); break;
    case APLA_EPI_GELU_FWD: if (img) TP_LAUNCH(APLA_EPI_GELU_FWD, bf16, 5, true); else TP_LAUNCH(APLA_EPI_GELU_FWD, bf16, 5, false); break;
    case APLA_EPI_STORE: TP_LAUNCH(APLA_EPI_STORE, bf16, 5, false); break;
    default: return APLA_ENOSYS;
  }
#undef TP_LAUNCH
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { apla_set_error("apla_gemm_nt[tp]: launch failed: %s", hipGetErrorString(e)); return APLA_EIO; }
  return APLA_OK;
}
