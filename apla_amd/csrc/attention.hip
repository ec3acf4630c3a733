// Fused multi-head attention (head_dim 64) forward + deterministic backward for gfx950.
//
// Data layout: the packed qkv activation [B*N, 3*H*64] exactly as the qkv Linear produces it ([.., 3, H, 64]);
// o / do are [B*N, H*64] (heads already merged), so neither the reshape/permute copies of appla_attn.py:53-54 nor the
// transpose(1,2).reshape of :60 exist.  The [B,H,N,N] score tensor is never written.
//
// All products use v_mfma_f32_32x32x16_bf16 in the "reduction-axis on the lane's registers" orientation:
//   forward   S^T = K·Q^T (key rows, q on the lane) -> online softmax is lane-local (one lane^32 exchange per row op)
//             O^T += V^T · P^T   with P^T taken straight from the accumulator as the B operand
//   backward  kernel dQ   (one wave = 32 query rows): S^T, dP^T = V·dO^T, dQ^T += K^T · dS^T
//             kernel dKdV (one wave = 32 keys):       S = Q·K^T, dP = dO·V^T, dV^T += dO^T · P, dK^T += Q^T · dS
// The transposed operands (V^T, K^T, dO^T, Q^T) come from the SAME row-major LDS tile through ds_read_b64_tr_b16;
// tiles use one XOR swizzle that is conflict-free for both the ds_read_b128 row reads and the transposed reads.
// dQ and dK/dV are produced by separate kernels so that no cross-workgroup reduction (atomics) is needed: results are
// bitwise reproducible.  delta = rowsum(dO*O) is produced by the dQ kernel and consumed by the dKdV kernel.
#include <stdlib.h>

#include <type_traits>

#include <atomic>
#include <map>
#include <mutex>
#include <utility>

#include "common.h"

#define ATT_LDSP(p) ((__attribute__((address_space(3))) void*)(p))
#define ATT_GLBP(p) ((const __attribute__((address_space(1))) void*)(p))

#if defined(APLA_ATT_STAMPS)   // diagnostic build (tools/attn_stamps.py): per-wave cycle sums of the fused backward's segments
__device__ unsigned long long apla_att_dbg[4096 * 4 * 16];
extern "C" int apla_attn_debug_dump(unsigned long long* host_out, int n) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(apla_att_dbg), (size_t)n * 8, 0, hipMemcpyDeviceToHost);
}
#define STAMP_DECL unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long st_t = clock64(), st_n;
#define STAMP(k) do { __builtin_amdgcn_sched_barrier(0); st_n = clock64(); st_acc[k] += st_n - st_t; st_t = st_n; __builtin_amdgcn_sched_barrier(0); } while (0)
#define STAMP_FLUSH() do { if (lane == 0) { const int wg = blockIdx.y * gridDim.x + blockIdx.x; if (wg < 4096) for (int k_ = 0; k_ < 16; ++k_) apla_att_dbg[(wg * 4 + wave) * 16 + k_] = st_acc[k_]; } } while (0)
#else
#define STAMP_DECL
#define STAMP(k)
#define STAMP_FLUSH()
#endif

namespace {

#if defined(APLA_ABL_ATT_NOEXP)   // diagnostic build: no transcendental in the backward softmax recompute
#define ATT_EXP2(x) (x)
#else
#define ATT_EXP2(x) __builtin_amdgcn_exp2f(x)
#endif
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

// 64-column bf16 tile (128-byte rows), 16-byte chunk index XOR f(row): conflict-free for ds_read_b128 row fragments
// (32 rows x one chunk) and for ds_read_b64_tr_b16 blocks (4 rows x 64 bytes per half-wave).
__device__ __forceinline__ int tile_off(int row, int chunk) {
  const int f = (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
  return row * 128 + ((chunk ^ f) << 4);
}

// Stage a [ROWS x 64] bf16 tile: global (row stride ld elements, rows clamped to [0, nrows)) -> registers.
template <int ROWS>
struct TileRegs { bf16x8 v[ROWS * 8 / 256]; };

template <int ROWS>
__device__ __forceinline__ void tile_load(TileRegs<ROWS>& t, const bf16* base, long ld, int row0, int nrows, int tid) {
#pragma unroll
  for (int i = 0; i < ROWS * 8 / 256; ++i) {
    const int e = tid + i * 256, row = e >> 3, chunk = e & 7;
    int gr = row0 + row;
    gr = gr < nrows ? gr : nrows - 1;
    t.v[i] = *(const bf16x8*)(base + (long)gr * ld + chunk * 8);
  }
}
template <int ROWS>
__device__ __forceinline__ void tile_store(const TileRegs<ROWS>& t, char* lds, int tid) {
#pragma unroll
  for (int i = 0; i < ROWS * 8 / 256; ++i) {
    const int e = tid + i * 256, row = e >> 3, chunk = e & 7;
    *(bf16x8*)(lds + tile_off(row, chunk)) = t.v[i];
  }
}

// Row fragment (A operand, 32 rows x 16 k): lane l holds tile[row0 + (l&31)][16*ks + 8*(l>>5) + 0..7]
__device__ __forceinline__ bf16x8 row_frag(const char* lds, int row0, int ks, int lane) {
  return *(const bf16x8*)(lds + tile_off(row0 + (lane & 31), 2 * ks + (lane >> 5)));
}

// Transposed fragment (A operand = tile^T, 32 "columns of the tile" x 16 "rows of the tile"):
// element j of lane l = tile[rbase + 8*(j>>2) + 4*(l>>5) + (j&3)][c0 + (l&31)], matching the k-order in which a 32x32 f32
// accumulator is consumed as the other operand (see cdna guide: accumulator tile as next MFMA operand).
__device__ __forceinline__ bf16x8 tr_frag(const char* lds, int rbase, int c0, int lane) {
  const int g = lane >> 4, i = lane & 15;
  const int col = c0 + 16 * (g & 1) + 4 * (i & 3);  // this lane's address: 4 consecutive columns of one row
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

// The same fragment by inline asm, waited for by hand.  Next to in-flight LDS-DMA (global_load_lds) hipcc puts
// s_waitcnt vmcnt(0) in front of every ds_read_tr builtin it can see: in the key/query-blocked backward kernels that drained
// the NEXT tile's DMA in the middle of the current tile's products (no overlap of loading and computing at all).  The asm
// form is invisible to that pass: issue with tr_issue, then lds_landed() once per batch, which also pins the registers
// behind the wait so that no consumer is scheduled above it.
struct TrPair { bf16x4 lo, hi; };
__device__ __forceinline__ void tr_issue(TrPair& f, const char* lds, int rbase, int c0, int lane) {
  const int g = lane >> 4, i = lane & 15;
  const int col = c0 + 16 * (g & 1) + 4 * (i & 3);
  const int r = rbase + 4 * (g >> 1) + (i >> 2);
  const unsigned a0 = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)(lds + tile_off(r, col >> 3) + ((col & 4) << 1));
  const unsigned a1 = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)(lds + tile_off(r + 8, col >> 3) + ((col & 4) << 1));
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.lo) : "v"(a0));
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.hi) : "v"(a1));
}
template <int N>
__device__ __forceinline__ void lds_landed(TrPair (&f)[N]) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+v"(f[i].lo), "+v"(f[i].hi));
}
__device__ __forceinline__ bf16x8 tr_join(const TrPair& f) {
  bf16x8 out;
  out[0] = f.lo[0]; out[1] = f.lo[1]; out[2] = f.lo[2]; out[3] = f.lo[3];
  out[4] = f.hi[0]; out[5] = f.hi[1]; out[6] = f.hi[2]; out[7] = f.hi[3];
  return out;
}

// Lane-constant parts of the fragment addresses.  tile_off's swizzle looks at bits 1-3 of the row only, so for a fragment whose
// first row is a multiple of 16 the byte offset is (first row) * 128 + a value that depends on the lane alone: one VGPR per
// k-step (row fragments) or per (column half, row half) (transposed fragments) for the whole kernel, a single v_add per tile and
// instruction immediates for everything else.  Left to hipcc the swizzle arithmetic was redone for every fragment of every tile
// (18 / 30 of the 76 / 112 VALU instructions per tile in the two bodies of the fused backward).
struct FragOffs { unsigned rf[4], tr[4]; };
__device__ __forceinline__ FragOffs frag_offs(int lane) {
  FragOffs f;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) f.rf[ks] = (unsigned)tile_off(lane & 31, 2 * ks + (lane >> 5));
  const int g = lane >> 4, i = lane & 15;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int col = 32 * dt + 16 * (g & 1) + 4 * (i & 3);
      const int r = 4 * (g >> 1) + (i >> 2) + 8 * half;
      f.tr[2 * dt + half] = (unsigned)(tile_off(r, col >> 3) + ((col & 4) << 1));
    }
  return f;
}
__device__ __forceinline__ unsigned lds_addr(const char* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p; }
// transposed fragment pair (rows rbase .. rbase+15 of the tile at LDS byte address `tile`, columns 32*dt ..): base + immediate
template <int OFF>
__device__ __forceinline__ void tr_issue_at(TrPair& f, unsigned a_lo, unsigned a_hi) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.lo) : "v"(a_lo), "n"(OFF));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.hi) : "v"(a_hi), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ bf16x8 row_frag_at(unsigned a) {
  return *(const bf16x8*)((__attribute__((address_space(3))) const char*)(size_t)(a + OFF));
}

__device__ __forceinline__ bf16x8 acc_to_operand(const f32x16& a, int s) {
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16)a[8 * s + j];
  return o;
}

// row index (within a 32-row tile) of accumulator register `reg` for lane half h
__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// Register fragment straight from global (B operand with the matrix row on the lane):
// lane l holds M[row][16*ks + 8*(l>>5) + 0..7] for ks = 0..3
__device__ __forceinline__ void load_row_frags(bf16x8 (&f)[4], const bf16* rowptr, int lane) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) f[ks] = *(const bf16x8*)(rowptr + 16 * ks + 8 * (lane >> 5));
}

// Store a transposed accumulator pair (acc[dt][reg] = X[d = 32dt + acc_row(reg,h)][row on lane]) as bf16 to a row-major
// [rows, 64] slice: 8-byte pieces of 4 consecutive d.
__device__ __forceinline__ void store_acc_T(const f32x16 (&acc)[2], bf16* rowptr, int h, float mul) {
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 v = pack4(acc[dt][4 * g] * mul, acc[dt][4 * g + 1] * mul, acc[dt][4 * g + 2] * mul, acc[dt][4 * g + 3] * mul);
      *(bf16x4*)(rowptr + 32 * dt + 8 * g + 4 * h) = v;
    }
}

// store_acc_T of acc + coef * x^T (x: 64 fp32 values in LDS, one rank-1 term per lane's row): the term is added on the way out, the
// accumulator tuple itself is not modified (element-wise updates of an MFMA accumulator made hipcc copy the tuples: spills in a kernel
// that has no register to spare)
__device__ __forceinline__ void store_acc_T_rank1(const f32x16 (&acc)[2], bf16* rowptr, int h, float mul, float coef, const float* x) {
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 xv = *(const f32x4*)(x + 32 * dt + 8 * g + 4 * h);
      bf16x4 v = pack4((acc[dt][4 * g] + coef * xv[0]) * mul, (acc[dt][4 * g + 1] + coef * xv[1]) * mul,
                       (acc[dt][4 * g + 2] + coef * xv[2]) * mul, (acc[dt][4 * g + 3] + coef * xv[3]) * mul);
      *(bf16x4*)(rowptr + 32 * dt + 8 * g + 4 * h) = v;
    }
}

// The same store through a per-wave 4 KB LDS buffer [32 rows][64 columns], 16-byte chunk c of row r at chunk c ^ (r & 7): every
// global store instruction then writes 8 rows x 128 B (whole lines) with 16 bytes per lane — 4 instructions per 32 x 64 tile
// instead of 8 that touch 32-64 lines each.  In-kernel stamps showed the direct form costing 8-11k cycles per head in the fused
// backward (all waves of a workgroup reach their stores together and queue behind each other: store-issue bound).
// `row0` = sequence row of the tile's first row, rows >= N are not stored; dst points at row 0 of the sequence (+ head column).
__device__ __forceinline__ void store_acc_T_staged(const f32x16 (&acc)[2], char* buf, bf16* dst, long ld, int row0, int N, int lane, float mul) {
  // the addresses below derive from this opaque copy of the lane id: hipcc cannot hoist them out of the caller's loops, where
  // they would stay live across the product loops and push other lane constants to scratch (reloaded next to in-flight LDS-DMA)
  asm volatile("" : "+v"(lane));
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const bf16x4 v = pack4(acc[dt][4 * g] * mul, acc[dt][4 * g + 1] * mul, acc[dt][4 * g + 2] * mul, acc[dt][4 * g + 3] * mul);
      *(bf16x4*)(buf + r * 128 + (((4 * dt + g) ^ (r & 7)) << 4) + 8 * h) = v;   // columns 32dt + 8g + 4h .. +3 of row r
    }
  const int c = lane & 7;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int rr = (lane >> 3) + 8 * j;
    const bf16x8 v = *(const bf16x8*)(buf + rr * 128 + ((c ^ (rr & 7)) << 4));
    if (row0 + rr < N) *(bf16x8*)(dst + (long)(row0 + rr) * ld + c * 8) = v;
  }
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time even n in 0..8 (odd n wait like n - 1)
__device__ __forceinline__ void wait_vmcnt_upto8(int n) {
  n = __builtin_amdgcn_readfirstlane(n);
  if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (n >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (n >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// s_waitcnt vmcnt(n) for a wave-uniform run-time n in 0..4 (the counter is an instruction immediate)
__device__ __forceinline__ void wait_vmcnt_upto4(int n) {
  n = __builtin_amdgcn_readfirstlane(n);
  if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (n == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if (n == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Sequence geometry.  Uniform batch (cu == nullptr): sequence b holds tokens [b*N, (b+1)*N) and lse/delta are [B, H, N].
// Packed variable-length batch (block-diagonal attention, appla_attn_mem_eff.py:40-42 with a BlockDiagonalMask): sequence b
// holds tokens [cu[b], cu[b+1]) of one [total, ...] activation and lse/delta are [H, total].
struct Seq { int start, n; long stat; int stat_h; };  // stat index of (h, q) = stat + h*stat_h + q
__device__ __forceinline__ Seq seq_of(const int32_t* __restrict__ cu, int b, int N, int H, int total) {
  Seq s;
  if (cu != nullptr) { s.start = cu[b]; s.n = cu[b + 1] - s.start; s.stat = s.start; s.stat_h = total; }
  else { s.start = b * N; s.n = N; s.stat = (long)b * H * N; s.stat_h = N; }
  return s;
}

// One LDS-DMA piece = 8 rows x 128 B of a [rows][64] tile (row stride ld_ elements in global memory), piece `pr` (wave-uniform) to
// dst + pr KB.  The XOR swizzle of tile_off goes on the source column (the LDS side of a DMA is lane-linear).  For a piece of rows < N the
// address is (uniform piece base) + (one of two 32-bit lane offsets, by the piece's parity): computed per piece, the swizzle and the
// 64-bit row product cost ~240 cycles of a wave's time (round 5, stamps of the persistent forward's loader).  Pieces that hold rows >= N
// (they re-read row N - 1) take the long way.
struct PieceOffs { unsigned even, odd; };
__device__ __forceinline__ PieceOffs piece_offs(int lane, long ld_) {
  const int lr = lane >> 3;
  const int f_even = (((lr >> 1) & 1) << 2) | (lr >> 2);
  PieceOffs po;
  po.even = (unsigned)(lr * (int)ld_ * 2 + (((lane & 7) ^ f_even) << 4));
  po.odd = (unsigned)(lr * (int)ld_ * 2 + (((lane & 7) ^ f_even ^ 2) << 4));
  return po;
}
__device__ __forceinline__ void dma_piece(const bf16* src, long ld_, int pr, int N, int lane, const PieceOffs& po, char* dst) {
  if (pr * 8 + 8 <= N) {
    const char* pb = (const char*)src + (long)pr * 8 * ld_ * 2;
    __builtin_amdgcn_global_load_lds(ATT_GLBP(pb + ((pr & 1) ? po.odd : po.even)), ATT_LDSP(dst + pr * 1024), 16, 0, 0);
  } else {
    const int row = pr * 8 + (lane >> 3);
    const int f = (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
    const int gr = row < N ? row : N - 1;
    __builtin_amdgcn_global_load_lds(ATT_GLBP(src + (long)gr * ld_ + (((lane & 7) ^ f) << 3)), ATT_LDSP(dst + pr * 1024), 16, 0, 0);
  }
}

// Attention-probability dropout (appla_attn.py:58 `attn = self.attn_drop(attn)`, settable from main.py:109-111 --adr; 0 in every
// shipped configuration): attn_d = keep ? attn / (1 - p) : 0 between the softmax and the product with V.  Implemented in the key- /
// query-blocked kernels only (the module path routes there when attn_drop is active).  keep(row, key) = word (key & 3) of
// Philox4x32-10(counter {key >> 2, row (64 bit), offset}, key seed) >= p * 2^32 with row = the (b, h, q) index of lse — counter-based:
// the forward and the two backward kernels regenerate the same mask in their own tilings, nothing is stored, and the oracle
// (oracle/apla_oracle.py:philox_attn_keep_mask) reproduces it bit for bit.  Backward: with attn_d = attn o M / (1 - p),
// d attn = (dO V^T) o M / (1 - p), dS = attn o (d attn - delta) with delta = rowsum(dO o O) unchanged, dV = attn_d^T dO.
struct DropArgs { unsigned threshold; float inv_keep; unsigned long long seed; unsigned offset; };
// keep flags of keys 4*kg .. 4*kg+3 of one row: bit e = key 4*kg + e is kept
__device__ __forceinline__ unsigned drop_keep4(const DropArgs& da, unsigned long long row, unsigned kg) {
  unsigned c[4] = {kg, (unsigned)row, (unsigned)(row >> 32), da.offset};
  philox4x32_10(c, (unsigned)da.seed, (unsigned)(da.seed >> 32));
  return (c[0] >= da.threshold ? 1u : 0u) | (c[1] >= da.threshold ? 2u : 0u) | (c[2] >= da.threshold ? 4u : 0u) | (c[3] >= da.threshold ? 8u : 0u);
}

// ------------------------------------------------------------------------------------------------ forward
template <bool DROP>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ o,
                                                       float* __restrict__ lse, int Nmax, int H, float scale,
    const int32_t* __restrict__ cu, int total, DropArgs da) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 8192];
  char* Ks = smem;
  char* Vs = smem + 8192;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h2 = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 128;
  const Seq sq = seq_of(cu, b, Nmax, H, total);
  const int N = sq.n;
  if (q0 >= N) return;  // packed batches: this sequence is shorter than the grid's longest
  const int D = H * 64;
  const long ld = 3L * D;
  const bf16* base = qkv + (long)sq.start * ld + h * 64;
  const float c = scale * LOG2E;

  int q = q0 + wave * 32 + (lane & 31);
  const bool qvalid = q < N;
  const bool wave_active = q0 + wave * 32 < N;  // wave-uniform: waves past the sequence end only help staging K/V
  if (!qvalid) q = N - 1;
  bf16x8 qf[4];
  load_row_frags(qf, base + (long)q * ld, lane);

  f32x16 acc_o[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc_o[0][i] = 0.f; acc_o[1][i] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;

  const int nkb = (N + 63) / 64;
  TileRegs<64> kr, vr;
  tile_load<64>(kr, base + D, ld, 0, N, tid);
  tile_load<64>(vr, base + 2 * D, ld, 0, N, tid);
  for (int kb = 0; kb < nkb; ++kb) {
    __syncthreads();
    tile_store<64>(kr, Ks, tid);
    tile_store<64>(vr, Vs, tid);
    __syncthreads();
    if (kb + 1 < nkb) {
      tile_load<64>(kr, base + D, ld, (kb + 1) * 64, N, tid);
      tile_load<64>(vr, base + 2 * D, ld, (kb + 1) * 64, N, tid);
    }
    if (!wave_active) continue;
    const bool two = kb * 64 + 32 < N;  // second 32-key tile of this block has at least one valid key
    f32x16 s[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      if (kt == 1 && !two) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s[kt][i] = -INFINITY;
        continue;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) s[kt][i] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        s[kt] = MFMA_F32_32x32x16_H16(row_frag(Ks, kt * 32, ks, lane), qf[ks], s[kt]);
    }
    if (kb * 64 + 64 > N) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (kb * 64 + kt * 32 + acc_row(i, h2) >= N) s[kt][i] = -INFINITY;
    }
    float mx = s[0][0];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[kt][i]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
    const float mc = m_new * c;
    float rs = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        s[kt][i] = __builtin_amdgcn_exp2f(fmaf(s[kt][i], c, -mc));
        rs += s[kt][i];
      }
    l_run = l_run * alpha + rs;
    m_run = m_new;
    if constexpr (DROP) {   // the row sum above is the softmax's (undropped); the dropped probabilities go into the product with V
      const unsigned long long row = (unsigned long long)(sq.stat + (long)h * sq.stat_h + q);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const unsigned keep = drop_keep4(da, row, (unsigned)((kb * 64 + kt * 32 + 8 * g + 4 * h2) >> 2));
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (!((keep >> e) & 1)) s[kt][4 * g + e] = 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc_o[0][i] *= alpha; acc_o[1][i] *= alpha; }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      if (kt == 1 && !two) continue;
#pragma unroll
      for (int sk = 0; sk < 2; ++sk) {
        const bf16x8 pb = acc_to_operand(s[kt], sk);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
          acc_o[dt] = MFMA_F32_32x32x16_H16(tr_frag(Vs, kt * 32 + 16 * sk, 32 * dt, lane), pb, acc_o[dt]);
      }
    }
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (qvalid) {
    store_acc_T(acc_o, o + ((long)sq.start + q) * D + h * 64, h2, (DROP ? da.inv_keep : 1.0f) / l_tot);
    if (h2 == 0) lse[sq.stat + (long)h * sq.stat_h + q] = m_run * scale + __logf(l_tot);
  }
}

// ------------------------------------------------------------------------------------------------ forward, short sequences
// N <= 256 (ViT: 197 / 257 tokens do not need key blocking): ONE workgroup per (b, h) with ceil(N/32) waves.  The whole K
// and V of the head go global -> LDS once by LDS-DMA (1 KB pieces of 8 rows x 128 B, eight pieces per wave; the XOR
// swizzle of tile_off is applied on the source column since the LDS side of a DMA is lane-linear), one barrier, and then
// every wave walks the keys on its own — no per-block staging through registers, no further barriers.  Two workgroups
// per CU (56 KB of LDS each at N = 197).
constexpr int SMALL_MAX_ROWS = 288;   // up to nine 32-row blocks: also the 257-token sequences of ViT-*/14 at 224 px
constexpr int TINY_MAX_ROWS = 64;       // two 32-row blocks: the fused backward at two waves and 17 KB of LDS
constexpr int SMALL_MAX_ROWS_BWD = 288;  // the fused backward also takes the 257-token sequences of ViT-*/14 at 224 px (9 blocks of 32 rows)

__global__ __launch_bounds__(576, 4) void attn_fwd_small_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ o,
                                                                float* __restrict__ lse, int Nmax, int H, float scale,
    const int32_t* __restrict__ cu, int total) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // K rows [NP][64] then V rows [NP][64], NP = 32 * waves
  const int tid = threadIdx.x, lane = tid & 63, h2 = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = blockDim.x >> 6, NP = nw * 32;
  const int b = blockIdx.y, h = blockIdx.x;
  const Seq sq = seq_of(cu, b, Nmax, H, total);
  const int N = sq.n;
  if (N <= 0) return;
  const int D = H * 64;
  const long ld = 3L * D;
  const bf16* base = qkv + (long)sq.start * ld + h * 64;
  const float c = scale * LOG2E;
  char* Ks = smem;
  char* Vs = smem + NP * 128;

  // pieces: K has NP/8, V has NP/8; wave w issues pieces 8w .. 8w+7 of the combined list (NP/4 = 8 * nw pieces)
  const PieceOffs po = piece_offs(lane, ld);
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int pc = wave * 8 + it;          // wave-uniform
    const bool isv = pc >= NP / 8;
    const int pr = isv ? pc - NP / 8 : pc;  // piece inside K or V
    dma_piece(base + (isv ? 2 * D : D), ld, pr, N, lane, po, isv ? Vs : Ks);
  }

  int q = wave * 32 + (lane & 31);
  const bool qvalid = q < N;
  if (!qvalid) q = N - 1;
  bf16x8 qf[4];
  load_row_frags(qf, base + (long)q * ld, lane);

  f32x16 acc_o[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc_o[0][i] = 0.f; acc_o[1][i] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (wave * 32 >= N) return;  // packed batches: waves past this sequence's end only helped loading

  // One block = 64 keys = two 32-key tiles.  Only the sequence's LAST block can hold keys >= N: the full blocks run a body
  // without masking (as one predicated loop hipcc emitted compare + select for every score of every block: 64 of the ~300
  // VALU instructions per block).
  auto block = [&](int kb, auto MASKED) {
    constexpr bool masked = decltype(MASKED)::value;
    const bool two = !masked || kb * 64 + 32 < N;  // second 32-key tile of this block has at least one valid key
    f32x16 s[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      if (kt == 1 && !two) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s[kt][i] = -INFINITY;
        continue;
      }
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      s[kt] = MFMA_F32_32x32x16_H16(row_frag(Ks, kb * 64 + kt * 32, 0, lane), qf[0], zero);
#pragma unroll
      for (int ks = 1; ks < 4; ++ks)
        s[kt] = MFMA_F32_32x32x16_H16(row_frag(Ks, kb * 64 + kt * 32, ks, lane), qf[ks], s[kt]);
    }
    if constexpr (masked) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (kb * 64 + kt * 32 + acc_row(i, h2) >= N) s[kt][i] = -INFINITY;
    }
    float mx = s[0][0];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[kt][i]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
    const float mc = m_new * c;
    float rs = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        s[kt][i] = __builtin_amdgcn_exp2f(fmaf(s[kt][i], c, -mc));
        rs += s[kt][i];
      }
    l_run = l_run * alpha + rs;
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc_o[0][i] *= alpha; acc_o[1][i] *= alpha; }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      if (kt == 1 && !two) continue;
#pragma unroll
      for (int sk = 0; sk < 2; ++sk) {
        const bf16x8 pb = acc_to_operand(s[kt], sk);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
          acc_o[dt] = MFMA_F32_32x32x16_H16(tr_frag(Vs, kb * 64 + kt * 32 + 16 * sk, 32 * dt, lane), pb, acc_o[dt]);
      }
    }
  };
  const int nfull = N >> 6;   // blocks without a key >= N
#pragma unroll 1
  for (int kb = 0; kb < nfull; ++kb) block(kb, std::false_type{});
  if (N & 63) block(nfull, std::true_type{});
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (qvalid) {
    store_acc_T(acc_o, o + ((long)sq.start + q) * D + h * 64, h2, 1.0f / l_tot);
    if (h2 == 0) lse[sq.stat + (long)h * sq.stat_h + q] = m_run * scale + __logf(l_tot);
  }
}

// ------------------------------------------------------------------------------------------------ forward, persistent
// Uniform batches of short sequences (N <= 224) with at least one head per CU.  Ablations of the kernel above (round 5: all heads
// read from 256 cache-resident ones / key loop cut to its last block) put its 54 us per layer at ~25 us of skeleton — every
// workgroup issues its whole K / V, waits for it, computes, stores and exits, three rounds of that per CU —, ~12 us of exposed HBM
// latency and only ~17 us of products and softmax.  Here ONE workgroup per CU walks its heads, and every byte is requested a whole
// head ahead of its use:
//   waves  0 .. NT-1 own one 32-row query block each; wave NT is the loader: it issues every LDS-DMA piece (the compute waves issue
//          no K / V load at all) and waits for them before the head's only barrier.
//   LDS    K and V double-buffered, [NR rows][64] each with NR = N rounded up to 8, then 24 zero rows (the last 32-row tile of V
//          reads past NR: zeros times P = 0; the overrun of K lands in the next buffer and its scores are masked by select), then a
//          4 KB store buffer per compute wave (O leaves as whole lines).
//          LIMITATION (documented, not guarded): the last V tile of buffer V0 overruns into V1 — the NEXT head's V, possibly still
//          under DMA — and relies on P = 0 for those keys: finite values there contribute exactly 0, but an Inf / NaN in the
//          neighbouring head's V (an fp16 overflow upstream) gives 0 * Inf = NaN in an otherwise finite head.  The one-workgroup-per-
//          head kernels (variant 2) have no such coupling; 24 private zero rows per V buffer do not fit the 160 KB at 280 tokens.
//   head i barrier (K, V of head i landed; every wave is done with head i - 1) | loader: K, V of head i + 1 into the other buffers |
//          store O of head i - 1 (deferred, so that no wave ever waits for its own stores) | QK^T of ALL tiles: S stays in 16 * NT
//          registers, q rows of head i + 1 requested into the registers q just left | ONE row maximum, exp2, row sum, P rounded
//          to 16 bits | PV of all tiles | wait for the q rows.
// No online softmax: one maximum per row, no rescale of O.  The results agree with attn_fwd_small_kernel (four 64-key blocks with
// running maxima) to rounding, not bitwise.
template <int I, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < E) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, E>(f);
  }
}
struct QRegs { f32x4 q[4]; };
__device__ __forceinline__ void q_prefetch(QRegs& r, const bf16* qp) {
  asm volatile(
      "global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:32\n\t"
      "global_load_dwordx4 %2, %4, off offset:64\n\tglobal_load_dwordx4 %3, %4, off offset:96"
      : "=&v"(r.q[0]), "=&v"(r.q[1]), "=&v"(r.q[2]), "=&v"(r.q[3])
      : "v"(qp)
      : "memory");
}
__device__ __forceinline__ void q_pin(QRegs& r) {
  asm volatile("" : "+v"(r.q[0]), "+v"(r.q[1]), "+v"(r.q[2]), "+v"(r.q[3]));
  __builtin_amdgcn_sched_barrier(0);
}

constexpr int FWDP_PAD_ROWS = 24;
constexpr int FWDP_MAX_NT = 9;      // up to 7 blocks: one wave per block; 8 / 9: four waves walking the blocks
constexpr int FWDP_BLOCK_WAVES = 4;
constexpr int FWDP_MAX_ROWS = 280;   // 4 x 280 x 128 B of K, V + 3 KB of zero rows + 16 KB of store buffers = 159.0 KB of the CU's 160 KB
template <int NT>
__global__ __launch_bounds__(64 * (NT + 1), 2) void attn_fwd_persist_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ o,
                                                                          float* __restrict__ lse, int N, int H, float scale,
                                                                          int BH, int NR) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, h2 = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = H * 64;
  const long ld = 3L * D;
  const float c = scale * LOG2E;
  const int TB = NR * 128;                 // bytes per buffer; layout K0 | K1 | V0 | V1 | zero rows
  const int npc = NR >> 3;                 // 1 KB pieces (8 rows x 128 B) per buffer
  const FragOffs fo = frag_offs(lane);
  const unsigned k0a = lds_addr(smem), v0a = k0a + 2 * TB;

  // zero rows behind V1, and the first rows of V1 itself: the first head's last V tile overruns V0 into V1 before any DMA has
  // written there (whatever the LDS held times P = 0 must not be NaN)
  for (int i = tid; i < FWDP_PAD_ROWS * 8; i += 64 * (NT + 1)) {
    *(f32x4*)(smem + 4 * TB + i * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    *(f32x4*)(smem + 3 * TB + i * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  auto base_of = [&](int idx) { const int b = idx / H, h = idx - b * H; return qkv + (long)b * N * ld + h * 64; };
  // loader wave: K and V of one head, 2 * npc pieces of 8 rows x 128 B; the XOR swizzle of tile_off goes on the source column (the
  // LDS side of a DMA is lane-linear).  A piece's address is (wave-uniform piece base) + (a 32-bit lane offset that takes two values,
  // for even and odd pieces): nothing but scalar adds between two DMA instructions (computed per piece, the swizzle and the 64-bit
  // row products cost ~240 cycles per piece: 11.8k cycles per head, more than the compute waves need for the head itself).  Only
  // the buffer's last piece can hold rows >= N (they re-read row N - 1) and takes the long way.
  const int lr = lane >> 3;
  const int f_even = (((lr >> 1) & 1) << 2) | (lr >> 2);
  const unsigned loff_even = (unsigned)(lr * (int)ld * 2 + (((lane & 7) ^ f_even) << 4));
  const unsigned loff_odd = (unsigned)(lr * (int)ld * 2 + (((lane & 7) ^ f_even ^ 2) << 4));
  const int nfull = N >> 3;   // pieces without a row >= N
  auto stage_buf = [&](const bf16* src, char* dst) {
    const char* pb = (const char*)src;
    const long step = 8L * ld * 2;
    int pr = 0;
#pragma unroll 1
    for (; pr + 1 < nfull; pr += 2) {
      __builtin_amdgcn_global_load_lds(ATT_GLBP(pb + loff_even), ATT_LDSP(dst + pr * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(ATT_GLBP(pb + step + loff_odd), ATT_LDSP(dst + pr * 1024 + 1024), 16, 0, 0);
      pb += 2 * step;
    }
    for (; pr < npc; ++pr) {   // an odd full piece and / or the ragged last piece
      const int row = pr * 8 + lr;
      const int f = (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
      const int gr = row < N ? row : N - 1;
      __builtin_amdgcn_global_load_lds(ATT_GLBP(src + (long)gr * ld + (((lane & 7) ^ f) << 3)), ATT_LDSP(dst + pr * 1024), 16, 0, 0);
    }
  };
  auto stage_head = [&](const bf16* hb, int par) {
    stage_buf(hb + D, smem + par * TB);
    stage_buf(hb + 2 * D, smem + 2 * TB + par * TB);
  };

  int idx = blockIdx.x;
  if (idx >= BH) return;
  const int G = gridDim.x;
  STAMP_DECL
  if (wave == NT) {   // ------------------------------------------------------------------------------ loader wave
    stage_head(base_of(idx), 0);
    int par = 0;
    while (true) {
      STAMP(0);   // loader: issue
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      STAMP(1);   // loader: wait for the pieces
      __builtin_amdgcn_s_barrier();
      STAMP(2);   // loader: barrier
      const int next = idx + G;
      if (next >= BH) break;
      par ^= 1;
      stage_head(base_of(next), par);
      idx = next;
    }
#if defined(APLA_ATT_STAMPS)
    if (lane == 0 && blockIdx.x < 512) for (int k_ = 0; k_ < 16; ++k_) apla_att_dbg[(blockIdx.x * 8 + wave) * 16 + k_] = st_acc[k_];
#endif
    return;
  }
  // ---------------------------------------------------------------------------------------------------- compute waves
  int q = wave * 32 + (lane & 31);
  const bool qvalid = q < N;
  if (!qvalid) q = N - 1;
  // head coordinates (b, h) advance by G heads per step without a division
  const int gq = G / H, gr_ = G - gq * H;
  int hb_ = idx / H, hh_ = idx - hb_ * H;       // current head
  int pb_ = hb_, ph_ = hh_;                      // previous head (deferred store)
  const bf16* base = qkv + (long)hb_ * N * ld + hh_ * 64;
  QRegs qr;
  q_prefetch(qr, base + (long)q * ld + 8 * h2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  q_pin(qr);
  int par = 0, prev = -1;
  f32x16 acc_o[2];
  float m_fin = 0.f, l_fin = 1.f;
  char* wbuf = smem + 4 * TB + FWDP_PAD_ROWS * 128 + wave * 4096;   // this wave's 4 KB store buffer (whole-line stores: store_acc_T_staged)
  auto store_head = [&]() {   // head (pb_, ph_)
    const float l_tot = l_fin + __shfl_xor(l_fin, 32, 64);
    store_acc_T_staged(acc_o, wbuf, o + (long)pb_ * N * D + ph_ * 64, D, wave * 32, N, lane, 1.0f / l_tot);
    if (qvalid && h2 == 0) lse[((long)pb_ * H + ph_) * N + q] = m_fin * scale + __logf(l_tot);
  };
  while (true) {
    STAMP(5);   // q wait + loop overhead
    __builtin_amdgcn_s_barrier();   // K, V of head idx are in LDS (the loader waited for them); every wave is done with head prev
    asm volatile("" ::: "memory");
    STAMP(0);   // barrier
    if (prev >= 0) store_head();
    STAMP(1);   // deferred store
    const int next = idx + G;
    const bool has_next = next < BH;
    int nb_ = hb_ + gq, nh_ = hh_ + gr_;
    if (nh_ >= H) { nh_ -= H; ++nb_; }
    const bf16* nbase = has_next ? qkv + (long)nb_ * N * ld + nh_ * 64 : base;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = __builtin_bit_cast(bf16x8, qr.q[ks]);
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // per-head copies of the fragment bases behind an opaque move: every tile is then an instruction immediate away from four
    // registers; left visible, hipcc hoists one address per (tile, fragment) out of the head loop — 56 lane constants at seven tiles
    unsigned kr[4], vt[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      kr[k] = k0a + par * TB + fo.rf[k];
      vt[k] = v0a + par * TB + fo.tr[k];
      asm volatile("" : "+v"(kr[k]), "+v"(vt[k]));
    }
    // ---------------------------------------------------------------- S^T = K Q^T, all tiles
    f32x16 sc[NT];
    static_for<0, NT>([&](auto T) {
      constexpr int t = decltype(T)::value, OFF = t * 4096;
      sc[t] = MFMA_F32_32x32x16_H16(row_frag_at<OFF>(kr[0]), qf[0], zero);
#pragma unroll
      for (int ks = 1; ks < 4; ++ks) sc[t] = MFMA_F32_32x32x16_H16(row_frag_at<OFF>(kr[ks]), qf[ks], sc[t]);
    });
    asm volatile("" :: "v"(sc[NT - 1][15]));
    STAMP(2);   // S products
    // qf is dead: the q rows of the next head (the last head re-reads its own, never used)
    q_prefetch(qr, nbase + (long)q * ld + 8 * h2);
    if (N < 32 * NT) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if ((NT - 1) * 32 + acc_row(i, h2) >= N) sc[NT - 1][i] = -INFINITY;
    }
    // ---------------------------------------------------------------- softmax: one maximum per row
    float mx = sc[0][0];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sc[t][i]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mc = mx * c;
    float l = 0.f;
    bf16x8 pp[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        sc[t][i] = __builtin_amdgcn_exp2f(fmaf(sc[t][i], c, -mc));
        l += sc[t][i];
      }
      pp[t][0] = acc_to_operand(sc[t], 0);
      pp[t][1] = acc_to_operand(sc[t], 1);
    }
    asm volatile("" :: "v"(pp[NT - 1][1]));
    STAMP(3);   // softmax
    // ---------------------------------------------------------------- O^T = V^T P^T, all tiles
    static_for<0, NT>([&](auto T) {
      constexpr int t = decltype(T)::value, OFF = t * 4096;
      TrPair tf[4];   // [2 sk + dt]
      tr_issue_at<OFF>(tf[0], vt[0], vt[1]);
      tr_issue_at<OFF>(tf[1], vt[2], vt[3]);
      tr_issue_at<OFF + 2048>(tf[2], vt[0], vt[1]);
      tr_issue_at<OFF + 2048>(tf[3], vt[2], vt[3]);
      lds_landed(tf);
      if constexpr (t == 0) {
        acc_o[0] = MFMA_F32_32x32x16_H16(tr_join(tf[0]), pp[t][0], zero);
        acc_o[1] = MFMA_F32_32x32x16_H16(tr_join(tf[1]), pp[t][0], zero);
      } else {
        acc_o[0] = MFMA_F32_32x32x16_H16(tr_join(tf[0]), pp[t][0], acc_o[0]);
        acc_o[1] = MFMA_F32_32x32x16_H16(tr_join(tf[1]), pp[t][0], acc_o[1]);
      }
      acc_o[0] = MFMA_F32_32x32x16_H16(tr_join(tf[2]), pp[t][1], acc_o[0]);
      acc_o[1] = MFMA_F32_32x32x16_H16(tr_join(tf[3]), pp[t][1], acc_o[1]);
    });
    asm volatile("" :: "v"(acc_o[0][15]), "v"(acc_o[1][15]));
    STAMP(4);   // PV
    m_fin = mx;
    l_fin = l;
    // the q rows of the next head: requested a softmax and a PV ago
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    q_pin(qr);
    prev = idx;
    pb_ = hb_; ph_ = hh_;
    if (!has_next) break;
    idx = next;
    hb_ = nb_; hh_ = nh_;
    base = nbase;
    par ^= 1;
  }
  store_head();
#if defined(APLA_ATT_STAMPS)
  if (lane == 0 && blockIdx.x < 512) for (int k_ = 0; k_ < 16; ++k_) apla_att_dbg[(blockIdx.x * 8 + wave) * 16 + k_] = st_acc[k_];
#endif
}

// The same kernel for 8 or 9 blocks (225..288 tokens: the 257 of ViT-*/14 at 224 px).  16 NT score registers need two waves per SIMD,
// so NW = 4 compute waves (+ the loader) WALK the query blocks of a head — wave w takes blocks w, w + 4, (w + 8) — with the same
// pipeline per block as the kernel above has per head: the q rows of the wave's next block (of this head or of the next) are requested
// when the block's last score product has been issued, O of the previous block is stored at the start of the next one.  One barrier
// per head.  (A two-group online form with one wave per block needs three waves per SIMD = 168 registers and spills 41-57 under hipcc.)
template <int NT, int NW>
__global__ __launch_bounds__(64 * (NW + 1), 2) void attn_fwd_persist_blocks_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ o,
                                                                          float* __restrict__ lse, int N, int H, float scale,
                                                                          int BH, int NR) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, h2 = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = H * 64;
  const long ld = 3L * D;
  const float c = scale * LOG2E;
  const int TB = NR * 128;                 // bytes per buffer; layout K0 | K1 | V0 | V1 | zero rows
  const int npc = NR >> 3;                 // 1 KB pieces (8 rows x 128 B) per buffer
  const FragOffs fo = frag_offs(lane);
  const unsigned k0a = lds_addr(smem), v0a = k0a + 2 * TB;

  // zero rows behind V1, and the first rows of V1 itself: the first head's last V tile overruns V0 into V1 before any DMA has
  // written there (whatever the LDS held times P = 0 must not be NaN)
  for (int i = tid; i < FWDP_PAD_ROWS * 8; i += 64 * (NW + 1)) {
    *(f32x4*)(smem + 4 * TB + i * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    *(f32x4*)(smem + 3 * TB + i * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  auto base_of = [&](int idx) { const int b = idx / H, h = idx - b * H; return qkv + (long)b * N * ld + h * 64; };
  // loader wave: K and V of one head, 2 * npc pieces of 8 rows x 128 B; the XOR swizzle of tile_off goes on the source column (the
  // LDS side of a DMA is lane-linear).  A piece's address is (wave-uniform piece base) + (a 32-bit lane offset that takes two values,
  // for even and odd pieces): nothing but scalar adds between two DMA instructions (computed per piece, the swizzle and the 64-bit
  // row products cost ~240 cycles per piece: 11.8k cycles per head, more than the compute waves need for the head itself).  Only
  // the buffer's last piece can hold rows >= N (they re-read row N - 1) and takes the long way.
  const int lr = lane >> 3;
  const int f_even = (((lr >> 1) & 1) << 2) | (lr >> 2);
  const unsigned loff_even = (unsigned)(lr * (int)ld * 2 + (((lane & 7) ^ f_even) << 4));
  const unsigned loff_odd = (unsigned)(lr * (int)ld * 2 + (((lane & 7) ^ f_even ^ 2) << 4));
  const int nfull = N >> 3;   // pieces without a row >= N
  auto stage_buf = [&](const bf16* src, char* dst) {
    const char* pb = (const char*)src;
    const long step = 8L * ld * 2;
    int pr = 0;
#pragma unroll 1
    for (; pr + 1 < nfull; pr += 2) {
      __builtin_amdgcn_global_load_lds(ATT_GLBP(pb + loff_even), ATT_LDSP(dst + pr * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(ATT_GLBP(pb + step + loff_odd), ATT_LDSP(dst + pr * 1024 + 1024), 16, 0, 0);
      pb += 2 * step;
    }
    for (; pr < npc; ++pr) {   // an odd full piece and / or the ragged last piece
      const int row = pr * 8 + lr;
      const int f = (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
      const int gr = row < N ? row : N - 1;
      __builtin_amdgcn_global_load_lds(ATT_GLBP(src + (long)gr * ld + (((lane & 7) ^ f) << 3)), ATT_LDSP(dst + pr * 1024), 16, 0, 0);
    }
  };
  auto stage_head = [&](const bf16* hb, int par) {
    stage_buf(hb + D, smem + par * TB);
    stage_buf(hb + 2 * D, smem + 2 * TB + par * TB);
  };

  int idx = blockIdx.x;
  if (idx >= BH) return;
  const int G = gridDim.x;
  if (wave == NW) {   // ------------------------------------------------------------------------------ loader wave
    stage_head(base_of(idx), 0);
    int par = 0;
    while (true) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const int next = idx + G;
      if (next >= BH) break;
      par ^= 1;
      stage_head(base_of(next), par);
      idx = next;
    }
    return;
  }
  // ---------------------------------------------------------------------------------------------------- compute waves
  // head coordinates (b, h) advance by G heads per step without a division
  const int gq = G / H, gr_ = G - gq * H;
  int hb_ = idx / H, hh_ = idx - hb_ * H;       // current head
  const bf16* base = qkv + (long)hb_ * N * ld + hh_ * 64;
  QRegs qr;
  {
    const int q0 = wave * 32 + (lane & 31);
    q_prefetch(qr, base + (long)(q0 < N ? q0 : N - 1) * ld + 8 * h2);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  q_pin(qr);
  int par = 0;
  bool have_prev = false;
  int pb_ = hb_, ph_ = hh_, pblk = wave;        // the block whose O is still in registers (deferred store)
  f32x16 acc_o[2];
  float m_fin = 0.f, l_fin = 1.f;
  char* wbuf = smem + 4 * TB + FWDP_PAD_ROWS * 128 + wave * 4096;   // this wave's 4 KB store buffer (whole-line stores: store_acc_T_staged)
  auto store_prev = [&]() {   // block pblk of head (pb_, ph_)
    const float l_tot = l_fin + __shfl_xor(l_fin, 32, 64);
    store_acc_T_staged(acc_o, wbuf, o + (long)pb_ * N * D + ph_ * 64, D, pblk * 32, N, lane, 1.0f / l_tot);
    const int pq = pblk * 32 + (lane & 31);
    if (pq < N && h2 == 0) lse[((long)pb_ * H + ph_) * N + pq] = m_fin * scale + __logf(l_tot);
  };
  while (true) {
    __builtin_amdgcn_s_barrier();   // K, V of head idx are in LDS (the loader waited for them); every wave is done with the previous head
    asm volatile("" ::: "memory");
    const int next = idx + G;
    const bool has_next = next < BH;
    int nb_ = hb_ + gq, nh_ = hh_ + gr_;
    if (nh_ >= H) { nh_ -= H; ++nb_; }
    const bf16* nbase = has_next ? qkv + (long)nb_ * N * ld + nh_ * 64 : base;
#pragma unroll 1
    for (int blk = wave; blk < NT; blk += NW) {
    if (have_prev) store_prev();
    // the wave's next block: of this head, else its first block of the next head (the very last block re-reads its own rows, never used)
    const bool more = blk + NW < NT;
    const int nq = (more ? blk + NW : wave) * 32 + (lane & 31);
    const bf16* nqp = (more ? base : nbase) + (long)(nq < N ? nq : N - 1) * ld + 8 * h2;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = __builtin_bit_cast(bf16x8, qr.q[ks]);
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // per-head copies of the fragment bases behind an opaque move: every tile is then an instruction immediate away from four
    // registers; left visible, hipcc hoists one address per (tile, fragment) out of the head loop — 56 lane constants at seven tiles
    unsigned kr[4], vt[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      kr[k] = k0a + par * TB + fo.rf[k];
      vt[k] = v0a + par * TB + fo.tr[k];
      asm volatile("" : "+v"(kr[k]), "+v"(vt[k]));
    }
    // ---------------------------------------------------------------- S^T = K Q^T, all tiles
    f32x16 sc[NT];
    static_for<0, NT>([&](auto T) {
      constexpr int t = decltype(T)::value, OFF = t * 4096;
      sc[t] = MFMA_F32_32x32x16_H16(row_frag_at<OFF>(kr[0]), qf[0], zero);
#pragma unroll
      for (int ks = 1; ks < 4; ++ks) sc[t] = MFMA_F32_32x32x16_H16(row_frag_at<OFF>(kr[ks]), qf[ks], sc[t]);
    });
    asm volatile("" :: "v"(sc[NT - 1][15]));
    q_prefetch(qr, nqp);   // qf is dead
    if (N < 32 * NT) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if ((NT - 1) * 32 + acc_row(i, h2) >= N) sc[NT - 1][i] = -INFINITY;
    }
    // ---------------------------------------------------------------- softmax: one maximum per row
    float mx = sc[0][0];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sc[t][i]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mc = mx * c;
    float l = 0.f;
    bf16x8 pp[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        sc[t][i] = __builtin_amdgcn_exp2f(fmaf(sc[t][i], c, -mc));
        l += sc[t][i];
      }
      pp[t][0] = acc_to_operand(sc[t], 0);
      pp[t][1] = acc_to_operand(sc[t], 1);
    }
    asm volatile("" :: "v"(pp[NT - 1][1]));
    // ---------------------------------------------------------------- O^T = V^T P^T, all tiles
    static_for<0, NT>([&](auto T) {
      constexpr int t = decltype(T)::value, OFF = t * 4096;
      TrPair tf[4];   // [2 sk + dt]
      tr_issue_at<OFF>(tf[0], vt[0], vt[1]);
      tr_issue_at<OFF>(tf[1], vt[2], vt[3]);
      tr_issue_at<OFF + 2048>(tf[2], vt[0], vt[1]);
      tr_issue_at<OFF + 2048>(tf[3], vt[2], vt[3]);
      lds_landed(tf);
      if constexpr (t == 0) {
        acc_o[0] = MFMA_F32_32x32x16_H16(tr_join(tf[0]), pp[t][0], zero);
        acc_o[1] = MFMA_F32_32x32x16_H16(tr_join(tf[1]), pp[t][0], zero);
      } else {
        acc_o[0] = MFMA_F32_32x32x16_H16(tr_join(tf[0]), pp[t][0], acc_o[0]);
        acc_o[1] = MFMA_F32_32x32x16_H16(tr_join(tf[1]), pp[t][0], acc_o[1]);
      }
      acc_o[0] = MFMA_F32_32x32x16_H16(tr_join(tf[2]), pp[t][1], acc_o[0]);
      acc_o[1] = MFMA_F32_32x32x16_H16(tr_join(tf[3]), pp[t][1], acc_o[1]);
    });
    asm volatile("" :: "v"(acc_o[0][15]), "v"(acc_o[1][15]));
    m_fin = mx;
    l_fin = l;
    // the q rows of the wave's next block: requested a softmax and a PV ago
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    q_pin(qr);
    have_prev = true;
    pb_ = hb_; ph_ = hh_; pblk = blk;
    }
    if (!has_next) break;
    idx = next;
    hb_ = nb_; hh_ = nh_;
    base = nbase;
    par ^= 1;
  }
  if (have_prev) store_prev();
}

// ------------------------------------------------------------------------------------------------ backward: dQ (+delta)
template <bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ o,
                                                          const bf16* __restrict__ dout, const float* __restrict__ lse,
                                                          float* __restrict__ delta, bf16* __restrict__ dqkv, int Nmax, int H, float scale,
    const int32_t* __restrict__ cu, int total, DropArgs da) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 16384];  // two (K tile, V tile) buffers filled by LDS-DMA
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h2 = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 128;
  const Seq sq = seq_of(cu, b, Nmax, H, total);
  const int N = sq.n;
  if (q0 >= N) return;  // packed batches: this sequence is shorter than the grid's longest
  const int D = H * 64;
  const long ld = 3L * D;
  const bf16* base = qkv + (long)sq.start * ld + h * 64;
  const float c = scale * LOG2E;

  int q = q0 + wave * 32 + (lane & 31);
  const bool qvalid = q < N;
  if (!qvalid) q = N - 1;
  bf16x8 qf[4], dof[4];
  load_row_frags(qf, base + (long)q * ld, lane);
  load_row_frags(dof, dout + ((long)sq.start + q) * D + h * 64, lane);
  float dl = 0.f;
  {
    bf16x8 of[4];
    load_row_frags(of, o + ((long)sq.start + q) * D + h * 64, lane);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) dl += (float)dof[ks][j] * (float)of[ks][j];
  }
  dl += __shfl_xor(dl, 32, 64);
  const long statidx = sq.stat + (long)h * sq.stat_h + q;
  const bool wave_active = q0 + wave * 32 < N;
  if (qvalid && h2 == 0) delta[statidx] = dl;
  const float lse2 = lse[statidx] * LOG2E;

  f32x16 acc_dq[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc_dq[0][i] = 0.f; acc_dq[1][i] = 0.f; }

  const int nkb = (N + 63) / 64;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto issue = [&](int kb) {  // 16 pieces of 8 rows x 128 B (8 of K, 8 of V), four per wave; swizzle on the source column
    char* buf = smem + (kb & 1) * 16384;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int pc = wave_u * 4 + it;
      const bool isv = pc >= 8;
      const int pr = pc & 7;
      const int row = pr * 8 + (lane >> 3);
      const int f = (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
      const int ch = (lane & 7) ^ f;
      int gr = kb * 64 + row;
      gr = gr < N ? gr : N - 1;
      const bf16* src = base + (isv ? 2 * D : D) + (long)gr * ld + ch * 8;
      __builtin_amdgcn_global_load_lds(ATT_GLBP(src), ATT_LDSP(buf + (isv ? 8192 : 0) + pr * 1024), 16, 0, 0);
    }
  };
  issue(0);
  for (int kb = 0; kb < nkb; ++kb) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // tile kb is in LDS; every wave is done with tile kb-1
    if (kb + 1 < nkb) issue(kb + 1);
    const char* Ks = smem + (kb & 1) * 16384;
    const char* Vs = Ks + 8192;
    if (!wave_active) continue;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      if (kt == 1 && kb * 64 + 32 >= N) continue;  // fully masked key tile
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#if defined(APLA_ABL_ATT_NOS)   // diagnostic build: no S / dP products
      f32x16 s = zero, dp = zero;
      s[0] = (float)qf[0][0]; dp[0] = (float)dof[0][0];
#else
      f32x16 s = MFMA_F32_32x32x16_H16(row_frag(Ks, kt * 32, 0, lane), qf[0], zero);
      f32x16 dp = MFMA_F32_32x32x16_H16(row_frag(Vs, kt * 32, 0, lane), dof[0], zero);
#pragma unroll
      for (int ks = 1; ks < 4; ++ks) {
        s = MFMA_F32_32x32x16_H16(row_frag(Ks, kt * 32, ks, lane), qf[ks], s);
        dp = MFMA_F32_32x32x16_H16(row_frag(Vs, kt * 32, ks, lane), dof[ks], dp);
      }
#endif
      // K^T fragments [sk][dt] of the dQ product (hand-waited asm reads: see tr_issue), issued BEFORE the softmax arithmetic so
      // that they land behind it
      TrPair kt_[4];
#if defined(APLA_ABL_ATT_NOTR)   // diagnostic build: no transposed reads, no second-stage products
      for (int i = 0; i < 4; ++i) { kt_[i].lo = bf16x4{}; kt_[i].hi = bf16x4{}; }
#else
#pragma unroll
      for (int sk = 0; sk < 2; ++sk)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) tr_issue(kt_[2 * sk + dt], Ks, kt * 32 + 16 * sk, 32 * dt, lane);
#endif
      if constexpr (DROP) {   // d attn = (dO V^T) o M / (1 - p)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const unsigned keep = drop_keep4(da, (unsigned long long)statidx, (unsigned)((kb * 64 + kt * 32 + 8 * g + 4 * h2) >> 2));
#pragma unroll
          for (int e = 0; e < 4; ++e) dp[4 * g + e] = ((keep >> e) & 1) ? dp[4 * g + e] * da.inv_keep : 0.f;
        }
      }
      // dS^T (unscaled).  Only the sequence's last key tile needs the per-key mask: as one predicated loop hipcc emits the
      // compare/select pair for every element of every tile (45 % of this kernel's VALU instructions).
      if (kb * 64 + kt * 32 + 32 <= N) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = ATT_EXP2(fmaf(s[i], c, -lse2)) * (dp[i] - dl);
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float p = ATT_EXP2(fmaf(s[i], c, -lse2));
          if (kb * 64 + kt * 32 + acc_row(i, h2) >= N) p = 0.f;
          s[i] = p * (dp[i] - dl);
        }
      }
#if defined(APLA_ABL_ATT_NOTR)
      acc_dq[0][0] += s[0] + s[5]; acc_dq[1][0] += s[9] + s[15];
#else
      lds_landed(kt_);
#pragma unroll
      for (int sk = 0; sk < 2; ++sk) {
        const bf16x8 dsb = acc_to_operand(s, sk);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
          acc_dq[dt] = MFMA_F32_32x32x16_H16(tr_join(kt_[2 * sk + dt]), dsb, acc_dq[dt]);
      }
#endif
    }
  }
  if (qvalid) store_acc_T(acc_dq, dqkv + ((long)sq.start + q) * ld + h * 64, h2, scale);
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
template <bool DROP>
__global__ __launch_bounds__(256, (DROP ? 1 : 2)) void attn_bwd_dkv_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ dout,
                                                           const float* __restrict__ lse,
                                                           const float* __restrict__ delta, bf16* __restrict__ dqkv,
                                                           int Nmax, int H, float scale,
    const int32_t* __restrict__ cu, int total, DropArgs da) {
  // two (Q tile, dO tile) buffers filled by LDS-DMA (no staging registers: the kernel then fits three waves per SIMD),
  // then two (lse, delta) buffers
  __shared__ __attribute__((aligned(16))) char smem[2 * 16384 + 2 * 2 * 64 * 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h2 = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y, k0 = blockIdx.x * 128;
  const Seq sq = seq_of(cu, b, Nmax, H, total);
  const int N = sq.n;
  if (k0 >= N) return;  // packed batches: this sequence is shorter than the grid's longest
  const int D = H * 64;
  const long ld = 3L * D;
  const bf16* base = qkv + (long)sq.start * ld + h * 64;
  const bf16* dobase = dout + (long)sq.start * D + h * 64;
  const float* lsebase = lse + sq.stat + (long)h * sq.stat_h;
  const float* dlbase = delta + sq.stat + (long)h * sq.stat_h;
  const float c = scale * LOG2E;

  int key = k0 + wave * 32 + (lane & 31);
  const bool kvalid = key < N;
  const bool wave_active = k0 + wave * 32 < N;
  if (!kvalid) key = N - 1;
  bf16x8 kf[4], vf[4];
  load_row_frags(kf, base + D + (long)key * ld, lane);
  load_row_frags(vf, base + 2 * D + (long)key * ld, lane);

  f32x16 acc_dk[2], acc_dv[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc_dk[0][i] = 0.f; acc_dk[1][i] = 0.f; acc_dv[0][i] = 0.f; acc_dv[1][i] = 0.f; }

  const int nqb = (N + 63) / 64;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto issue = [&](int qb) {  // 16 pieces of 8 rows x 128 B (8 of Q, 8 of dO), four per wave; swizzle on the source column
    char* buf = smem + (qb & 1) * 16384;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int pc = wave_u * 4 + it;
      const bool isd = pc >= 8;
      const int pr = pc & 7;
      const int row = pr * 8 + (lane >> 3);
      const int f = (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
      const int ch = (lane & 7) ^ f;
      int gr = qb * 64 + row;
      gr = gr < N ? gr : N - 1;
      const bf16* src = isd ? dobase + (long)gr * D + ch * 8 : base + (long)gr * ld + ch * 8;
      __builtin_amdgcn_global_load_lds(ATT_GLBP(src), ATT_LDSP(buf + (isd ? 8192 : 0) + pr * 1024), 16, 0, 0);
    }
    {  // lse and delta of the 64 query rows: one 4-byte LDS-DMA per wave (lane-linear), no registers carried across the loop.
       // Branch-free on purpose (even waves fetch lse, odd waves delta, twice each): a conditional block here is sunk by hipcc
       // to the end of the iteration, right in front of the wait that needs it.
      int qq = qb * 64 + lane;
      qq = qq < N ? qq : N - 1;
      const int which = wave_u & 1;
      const float* sp = (which ? dlbase : lsebase) + qq;
      __builtin_amdgcn_global_load_lds(ATT_GLBP(sp), ATT_LDSP(smem + 2 * 16384 + (qb & 1) * 512 + which * 256), 4, 0, 0);
    }
  };
  const float* stats = (const float*)(smem + 2 * 16384);  // [2 buffers][lse 64 | delta 64]
  issue(0);
  for (int qb = 0; qb < nqb; ++qb) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // tile qb (and its lse/delta) is in LDS; every wave is done with tile qb-1
    if (qb + 1 < nqb) issue(qb + 1);
    const char* Qs = smem + (qb & 1) * 16384;
    const char* dOs = Qs + 8192;
    const float* lses = stats + (qb & 1) * 128;
    const float* dls = lses + 64;
    if (!wave_active) continue;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      if (qt == 1 && qb * 64 + 32 >= N) continue;  // fully masked query tile
      unsigned keepbits = 0;   // DROP: bit i = element i of this tile (this lane's key, query row acc_row(i, h2) of the tile) is kept
      if constexpr (DROP) {    // (a rolled loop in front of the products: sixteen unrolled Philox blocks beside 240 live registers spill)
        const long rowbase = sq.stat + (long)h * sq.stat_h + qb * 64 + qt * 32 + 4 * h2;
#pragma unroll 1
        for (int i = 0; i < 16; ++i) {
          const unsigned keep = drop_keep4(da, (unsigned long long)(rowbase + (i & 3) + 8 * (i >> 2)), (unsigned)key >> 2);
          keepbits |= ((keep >> (key & 3)) & 1u) << i;
        }
      }
      f32x16 s, dp;
#pragma unroll
      for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#if defined(APLA_ABL_ATT_NOS)
      s[0] = (float)kf[0][0]; dp[0] = (float)vf[0][0];
#else
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s = MFMA_F32_32x32x16_H16(row_frag(Qs, qt * 32, ks, lane), kf[ks], s);
        dp = MFMA_F32_32x32x16_H16(row_frag(dOs, qt * 32, ks, lane), vf[ks], dp);
      }
#endif
      // dO^T [sk][dt] and Q^T [sk][dt] of the dV / dK products (hand-waited asm reads: see tr_issue), issued BEFORE the softmax
      // arithmetic so that they land behind it
      TrPair tf[8];
#if defined(APLA_ABL_ATT_NOTR)
      for (int i = 0; i < 8; ++i) { tf[i].lo = bf16x4{}; tf[i].hi = bf16x4{}; }
#else
#pragma unroll
      for (int sk = 0; sk < 2; ++sk)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          tr_issue(tf[4 * sk + dt], dOs, qt * 32 + 16 * sk, 32 * dt, lane);
          tr_issue(tf[4 * sk + 2 + dt], Qs, qt * 32 + 16 * sk, 32 * dt, lane);
        }
#endif
      if constexpr (DROP) {   // d attn = (dO V^T) o M / (1 - p)
#pragma unroll
        for (int i = 0; i < 16; ++i) dp[i] = ((keepbits >> i) & 1) ? dp[i] * da.inv_keep : 0.f;
      }
      // P and dS (in place of dP).  Only the sequence's last query tile needs the per-row mask (see the dQ kernel).
      if (qb * 64 + qt * 32 + 32 <= N) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int r0 = qt * 32 + 8 * g + 4 * h2;
          const f32x4 l4 = *(const f32x4*)(lses + r0) * LOG2E, d4 = *(const f32x4*)(dls + r0);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int i = 4 * g + e;
            const float p = ATT_EXP2(fmaf(s[i], c, -l4[e]));
            s[i] = p;
            dp[i] = p * (dp[i] - d4[e]);
          }
        }
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int r0 = qt * 32 + 8 * g + 4 * h2;
          const f32x4 l4 = *(const f32x4*)(lses + r0) * LOG2E, d4 = *(const f32x4*)(dls + r0);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int i = 4 * g + e;
            float p = ATT_EXP2(fmaf(s[i], c, -l4[e]));
            if (qb * 64 + r0 + e >= N) p = 0.f;
            s[i] = p;
            dp[i] = p * (dp[i] - d4[e]);
          }
        }
      }
      if constexpr (DROP) {   // dV = attn_d^T dO: the P operand of that product is the dropped, rescaled one
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = ((keepbits >> i) & 1) ? s[i] * da.inv_keep : 0.f;
      }
#if defined(APLA_ABL_ATT_NOTR)
      acc_dv[0][0] += s[0] + s[7]; acc_dk[0][0] += dp[3] + dp[12];
      if (false)
#else
      lds_landed(tf);
#endif
#pragma unroll
      for (int sk = 0; sk < 2; ++sk) {
        const bf16x8 pb = acc_to_operand(s, sk), dsb = acc_to_operand(dp, sk);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          acc_dv[dt] = MFMA_F32_32x32x16_H16(tr_join(tf[4 * sk + dt]), pb, acc_dv[dt]);
          acc_dk[dt] = MFMA_F32_32x32x16_H16(tr_join(tf[4 * sk + 2 + dt]), dsb, acc_dk[dt]);
        }
      }
    }
  }
  if (kvalid) {
    bf16* orow = dqkv + ((long)sq.start + key) * ld + h * 64;
    store_acc_T(acc_dk, orow + D, h2, scale);
    store_acc_T(acc_dv, orow + 2 * D, h2, 1.0f);
  }
}

// ------------------------------------------------------------------------------------------------ backward, short sequences
// N <= 288: ONE workgroup per (b, h) computes dQ, dK and dV.  At these lengths the backward is bound by HBM traffic, not by
// its products (ablation builds: removing the exp, the S/dP products or the second-stage products changes the split kernels
// by 0 / 20 / 25 %, while their 2 x 232 MB per layer at 5 TB/s are 93 us of the 150 us they take): the split kernels read
// Q, K, V and dO of every head twice.  Here every operand of the head is read from HBM once:
//   phase 1  K, V -> LDS (LDS-DMA), lse by 4-byte LDS-DMA; each wave, for its query blocks: delta = rowsum(dO * O) (to LDS and
//            to the caller's buffer), then the dQ body (S^T, dP^T, dQ^T += K^T dS^T) over the resident key tiles;
//   phase 2  the SAME LDS buffer is refilled with Q, dO; each wave, for its key blocks: the dK/dV body (S, dP,
//            dV^T += dO^T P, dK^T += Q^T dS) over the resident query tiles.
// MAXR = rows the LDS layout holds, NW = waves per workgroup.  <288, 4> is the general form; <64, 2> the one for sequences of at
// most 64 tokens (the 50-token local crops of the multi-crop batch: 17 KB of LDS and two waves, four workgroups per CU, where the
// general form would hold 76 KB and leave two of its four waves without a block).
// Four waves per workgroup, wave w owns the 32-row blocks w, w+4 (and w+8): 58 KB of LDS at N = 197 and 256 threads, so TWO
// workgroups share a CU and one's loads / stores run under the other's products (a first version with one 7-wave workgroup
// per CU and all four tiles resident, 116 KB, serialised them: 139 us; the split kernels: 150 us).  Same seven products in
// the same order as the split kernels: bitwise equal to them.
template <int MAXR, int NW>
__global__ __launch_bounds__(64 * NW, 2) void attn_bwd_small_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ o,
                                                                const bf16* __restrict__ dout, const float* __restrict__ lse,
                                                                float* __restrict__ delta, bf16* __restrict__ dqkv, int Nmax,
                                                                int H, float scale, const int32_t* __restrict__ cu, int total,
                                                                int NP) {
  // two tiles at a FIXED distance (so that the second tile is an instruction immediate away from the first), then
  // lse * log2(e) and delta of the head's rows, both written by phase 1
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, h2 = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y, h = blockIdx.x;
  const Seq sq = seq_of(cu, b, Nmax, H, total);
  const int N = sq.n;
  if (N <= 0) return;
#if defined(APLA_ATT_PRIO)
  if (((blockIdx.y * gridDim.x + blockIdx.x) >> 8) & 1) __builtin_amdgcn_s_setprio(APLA_ATT_PRIO);
#endif
  const int D = H * 64;
  const long ld = 3L * D;
  const bf16* base = qkv + (long)sq.start * ld + h * 64;
  const bf16* dobase = dout + (long)sq.start * D + h * 64;
  const float c = scale * LOG2E;
  constexpr int TB_OFF = MAXR * 128;   // byte distance TA -> TB
  char* TA = smem;            // K, later Q
  char* TB = smem + TB_OFF;   // V, later dO
  float* lses2 = (float*)(smem + 2 * TB_OFF);
  float* dls = lses2 + MAXR;
  const int nt = (N + 31) / 32;   // 32-row blocks of this sequence
  const int npc = NP / 8;         // 1 KB pieces (8 rows x 128 B) per tile
  const FragOffs fo = frag_offs(lane);
  const unsigned ta0 = lds_addr(TA);
  STAMP_DECL

  // both tiles of a phase: 2 * npc pieces, dealt round-robin to the four waves; swizzle on the source column (the LDS side
  // of a DMA is lane-linear)
  auto stage = [&](const bf16* srcA, long ldA, const bf16* srcB, long ldB) {
    const PieceOffs poA = piece_offs(lane, ldA), poB = piece_offs(lane, ldB);
    for (int pc = wave; pc < 2 * npc; pc += NW) {     // wave-uniform
      if (pc >= npc) dma_piece(srcB, ldB, pc - npc, N, lane, poB, TB);
      else dma_piece(srcA, ldA, pc, N, lane, poA, TA);
    }
  };

  // ================================================================ phase 1: K, V resident; delta and dQ per query block
  stage(base + D, ld, base + 2 * D, ld);
  for (int blk = wave, first = 1; blk < MAXR / 32 + NW - 1; blk += NW, first = 0) {   // the barrier below is reached by every wave exactly once
    const bool active = blk < nt;
    int r = blk * 32 + (lane & 31);
    const bool rvalid = active && r < N;
    if (r >= N) r = N - 1;
    bf16x8 qf[4], dof[4];
    float dl = 0.f, lse2 = 0.f;
    if (active) {
      load_row_frags(qf, base + (long)r * ld, lane);
      load_row_frags(dof, dobase + (long)r * D, lane);
      bf16x8 of[4];
      load_row_frags(of, o + ((long)sq.start + r) * D + h * 64, lane);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) dl += (float)dof[ks][j] * (float)of[ks][j];
      dl += __shfl_xor(dl, 32, 64);
      const long statidx = sq.stat + (long)h * sq.stat_h + r;
      if (rvalid && h2 == 0) delta[statidx] = dl;
      lse2 = lse[statidx] * LOG2E;
      if (h2 == 0) { dls[blk * 32 + (lane & 31)] = dl; lses2[blk * 32 + (lane & 31)] = lse2; }   // for phase 2 (every row < NP)
    }
    STAMP(0);   // block start: row loads issued, delta
    if (first) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();   // K and V have landed
    }
    STAMP(1);   // wait for K, V (first block) / row loads
    if (!active) continue;
    f32x16 acc_dq[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc_dq[0][i] = 0.f; acc_dq[1][i] = 0.f; }
    for (int t = 0; t < nt; ++t) {
      const unsigned tb = ta0 + t * 4096;            // tile row 32t (uniform)
      unsigned ar[4], at[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { ar[k] = tb + fo.rf[k]; at[k] = tb + fo.tr[k]; }
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      f32x16 s = MFMA_F32_32x32x16_H16(row_frag_at<0>(ar[0]), qf[0], zero);
      f32x16 dp = MFMA_F32_32x32x16_H16(row_frag_at<TB_OFF>(ar[0]), dof[0], zero);
#pragma unroll
      for (int ks = 1; ks < 4; ++ks) {
        s = MFMA_F32_32x32x16_H16(row_frag_at<0>(ar[ks]), qf[ks], s);
        dp = MFMA_F32_32x32x16_H16(row_frag_at<TB_OFF>(ar[ks]), dof[ks], dp);
      }
      TrPair kt_[4];   // K^T fragments [sk][dt]: rows 32t + 16sk .., columns 32dt ..
      tr_issue_at<0>(kt_[0], at[0], at[1]);
      tr_issue_at<0>(kt_[1], at[2], at[3]);
      tr_issue_at<2048>(kt_[2], at[0], at[1]);
      tr_issue_at<2048>(kt_[3], at[2], at[3]);
      STAMP(2);   // P1: fragment reads + first-stage products issued
      if (t * 32 + 32 <= N) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = ATT_EXP2(fmaf(s[i], c, -lse2)) * (dp[i] - dl);
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float p = ATT_EXP2(fmaf(s[i], c, -lse2));
          if (t * 32 + acc_row(i, h2) >= N) p = 0.f;
          s[i] = p * (dp[i] - dl);
        }
      }
      asm volatile("" :: "v"(s[0]), "v"(s[15]));
      STAMP(3);   // P1: softmax arithmetic (starts with the wait for the products)
      lds_landed(kt_);
#pragma unroll
      for (int sk = 0; sk < 2; ++sk) {
        const bf16x8 dsb = acc_to_operand(s, sk);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
          acc_dq[dt] = MFMA_F32_32x32x16_H16(tr_join(kt_[2 * sk + dt]), dsb, acc_dq[dt]);
      }
      STAMP(4);   // P1: second-stage products issued
    }
    if (rvalid) store_acc_T(acc_dq, dqkv + ((long)sq.start + r) * ld + h * 64, h2, scale);
    STAMP(5);     // P1: dQ stores issued
  }
  __syncthreads();   // every wave is done with K, V; all delta / lse rows are in LDS
  STAMP(6);       // barrier at the end of phase 1

  // ================================================================ phase 2: Q, dO resident; dK and dV per key block
  stage(base, ld, dobase, D);
  for (int blk = wave, first = 1; blk < MAXR / 32 + NW - 1; blk += NW, first = 0) {
    const bool active = blk < nt;
    int r = blk * 32 + (lane & 31);
    const bool rvalid = active && r < N;
    if (r >= N) r = N - 1;
    bf16x8 kf[4], vf[4];
    if (active) {   // own key rows as B operands (row on the lane): just read by phase 1, L2 hits
      load_row_frags(kf, base + D + (long)r * ld, lane);
      load_row_frags(vf, base + 2 * D + (long)r * ld, lane);
    }
    STAMP(7);   // P2 block start: stage issue (first) + own k / v row loads issued
    if (first) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();   // Q, dO have landed
    }
    STAMP(8);   // wait for Q, dO
    if (!active) continue;
    f32x16 acc_dk[2], acc_dv[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc_dk[0][i] = 0.f; acc_dk[1][i] = 0.f; acc_dv[0][i] = 0.f; acc_dv[1][i] = 0.f; }
    for (int t = 0; t < nt; ++t) {
      const unsigned tb = ta0 + t * 4096;
      unsigned ar[4], at[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { ar[k] = tb + fo.rf[k]; at[k] = tb + fo.tr[k]; }
      f32x16 s, dp;
#pragma unroll
      for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s = MFMA_F32_32x32x16_H16(row_frag_at<0>(ar[ks]), kf[ks], s);
        dp = MFMA_F32_32x32x16_H16(row_frag_at<TB_OFF>(ar[ks]), vf[ks], dp);
      }
      TrPair tf[8];   // [4 sk + dt] = dO^T, [4 sk + 2 + dt] = Q^T
      tr_issue_at<TB_OFF>(tf[0], at[0], at[1]);
      tr_issue_at<0>(tf[2], at[0], at[1]);
      tr_issue_at<TB_OFF>(tf[1], at[2], at[3]);
      tr_issue_at<0>(tf[3], at[2], at[3]);
      tr_issue_at<TB_OFF + 2048>(tf[4], at[0], at[1]);
      tr_issue_at<2048>(tf[6], at[0], at[1]);
      tr_issue_at<TB_OFF + 2048>(tf[5], at[2], at[3]);
      tr_issue_at<2048>(tf[7], at[2], at[3]);
      STAMP(9);   // P2: fragment reads + first-stage products issued
      // P and dS (in place of dP).  Only the sequence's last query tile needs the per-row mask: as one predicated loop hipcc emits
      // the compare/select pair for every element of every tile.
      if (t * 32 + 32 <= N) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int r0 = t * 32 + 8 * g + 4 * h2;
          const f32x4 l4 = *(const f32x4*)(lses2 + r0), d4 = *(const f32x4*)(dls + r0);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int i = 4 * g + e;
            const float p = ATT_EXP2(fmaf(s[i], c, -l4[e]));
            s[i] = p;
            dp[i] = p * (dp[i] - d4[e]);
          }
        }
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int r0 = t * 32 + 8 * g + 4 * h2;
          const f32x4 l4 = *(const f32x4*)(lses2 + r0), d4 = *(const f32x4*)(dls + r0);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int i = 4 * g + e;
            float p = ATT_EXP2(fmaf(s[i], c, -l4[e]));
            if (r0 + e >= N) p = 0.f;
            s[i] = p;
            dp[i] = p * (dp[i] - d4[e]);
          }
        }
      }
      asm volatile("" :: "v"(s[0]), "v"(dp[15]));
      STAMP(10);  // P2: softmax arithmetic
      lds_landed(tf);
#pragma unroll
      for (int sk = 0; sk < 2; ++sk) {
        const bf16x8 pb = acc_to_operand(s, sk), dsb = acc_to_operand(dp, sk);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          acc_dv[dt] = MFMA_F32_32x32x16_H16(tr_join(tf[4 * sk + dt]), pb, acc_dv[dt]);
          acc_dk[dt] = MFMA_F32_32x32x16_H16(tr_join(tf[4 * sk + 2 + dt]), dsb, acc_dk[dt]);
        }
      }
      STAMP(11);  // P2: second-stage products issued
    }
    if (rvalid) {
      bf16* orow = dqkv + ((long)sq.start + r) * ld + h * 64;
      store_acc_T(acc_dk, orow + D, h2, scale);
      store_acc_T(acc_dv, orow + 2 * D, h2, 1.0f);
    }
    STAMP(12);    // P2: dK / dV stores issued
  }
  STAMP_FLUSH();
}

// ------------------------------------------------------------------------------------------------ backward, persistent
// Uniform batches of short sequences (N <= 256: at most eight 32-row blocks).  In-kernel stamps of the kernel above
// (tools/attn_stamps.py, -DAPLA_ATT_STAMPS) show that only half of a wave's time is spent in the two product loops: a quarter
// goes to the start of each query block (its q / dO / o rows come straight from HBM and are waited for on the spot), the rest to
// the two load phases and the stores — and a wave that waits leaves its SIMD to ONE other wave, which alone keeps it ~55 % busy.
// Here ONE 8-wave workgroup per CU walks its heads (b*H + h = blockIdx.x, + gridDim.x, ...) and every byte is requested one phase
// before it is needed:
//   LDS   set A = (K, V) tiles of head i, set B = (Q, dO) tiles of head i, lse * log2(e) and delta of head i.
//   P1(i) dQ:     A(i) resident; the LDS-DMA of B(i) runs under it.  Wave w owns query block w: its q / dO / o rows and lse are in
//                 registers already (inline-asm loads issued at the end of P2(i-1) and waited for, behind the dK / dV stores, before
//                 the loop's back edge: hipcc treats an asm load's destination as written at once and may copy it, so the wait sits in
//                 the same straight-line code); delta = rowsum(dO * o) and lse * log2(e) go to LDS for P2.
//   P2(i) dK/dV:  B(i) resident.  Wave w owns key block w: its k / v row fragments are READ FROM SET A (the A and B operand of
//                 v_mfma_f32_32x32x16 have the same lane layout) before the barrier that frees the set; the LDS-DMA of A(i+1) runs
//                 under the products; then the row loads of head i+1, then the dK / dV stores.
// Two barriers per head, each preceded by a counted vmcnt (never a drain of the stores).  Products, their order and the
// arithmetic between them are those of attn_bwd_small_kernel (and of the split kernels): the results are bitwise the same.
struct RowRegs { f32x4 q[4], d[4], o[4]; float lse; };   // 16-byte pieces = 8 packed 16-bit values each

__device__ __forceinline__ void rows_prefetch(RowRegs& r, const bf16* qp, const bf16* dp, const bf16* op, const float* lp) {
  asm volatile(
      "global_load_dwordx4 %0, %13, off\n\tglobal_load_dwordx4 %1, %13, off offset:32\n\t"
      "global_load_dwordx4 %2, %13, off offset:64\n\tglobal_load_dwordx4 %3, %13, off offset:96\n\t"
      "global_load_dwordx4 %4, %14, off\n\tglobal_load_dwordx4 %5, %14, off offset:32\n\t"
      "global_load_dwordx4 %6, %14, off offset:64\n\tglobal_load_dwordx4 %7, %14, off offset:96\n\t"
      "global_load_dwordx4 %8, %15, off\n\tglobal_load_dwordx4 %9, %15, off offset:32\n\t"
      "global_load_dwordx4 %10, %15, off offset:64\n\tglobal_load_dwordx4 %11, %15, off offset:96\n\t"
      "global_load_dword %12, %16, off"
      : "=&v"(r.q[0]), "=&v"(r.q[1]), "=&v"(r.q[2]), "=&v"(r.q[3]), "=&v"(r.d[0]), "=&v"(r.d[1]), "=&v"(r.d[2]), "=&v"(r.d[3]),
        "=&v"(r.o[0]), "=&v"(r.o[1]), "=&v"(r.o[2]), "=&v"(r.o[3]), "=&v"(r.lse)
      : "v"(qp), "v"(dp), "v"(op), "v"(lp)
      : "memory");
}
// wait until at most YOUNGER vector-memory operations of this wave are outstanding, then pin the prefetched registers behind the wait
__device__ __forceinline__ void rows_pin(RowRegs& r) {
  asm volatile("" : "+v"(r.q[0]), "+v"(r.q[1]), "+v"(r.q[2]), "+v"(r.q[3]), "+v"(r.d[0]), "+v"(r.d[1]), "+v"(r.d[2]), "+v"(r.d[3]),
               "+v"(r.o[0]), "+v"(r.o[1]), "+v"(r.o[2]), "+v"(r.o[3]), "+v"(r.lse));
  __builtin_amdgcn_sched_barrier(0);
}
template <int YOUNGER>
__device__ __forceinline__ void rows_landed(RowRegs& r) {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER) : "memory");
  asm volatile("" : "+v"(r.q[0]), "+v"(r.q[1]), "+v"(r.q[2]), "+v"(r.q[3]), "+v"(r.d[0]), "+v"(r.d[1]), "+v"(r.d[2]), "+v"(r.d[3]),
               "+v"(r.o[0]), "+v"(r.o[1]), "+v"(r.o[2]), "+v"(r.o[3]), "+v"(r.lse));
  __builtin_amdgcn_sched_barrier(0);
}

constexpr int PERSIST_MAX_ROWS = 256;

// ---- N = 32 NT + 1 (EXTRA): a ViT/14 at 224 pixels has 16 x 16 patches + the class token = 257 tokens, one more than eight blocks, and
// the block-granular kernels pay a ninth block for it (attn_bwd_small_kernel<288, 4>: 560 us against 394 us at 256 tokens, 4 092 heads).
// The persistent kernel takes the first 32 NT tokens through its blocks and the LAST token x as rank-1 corrections in plain vector
// arithmetic (attention is symmetric in the token order):
//   P1, wave = query block:  p[i,x], ds[i,x] of the lane's query i against key x (two 64-long dot products from the q / dO registers),
//                            dQ_i += ds[i,x] k_x;  partial sums over the block's 32 queries of ds[i,x] q_i (-> dK_x) and p[i,x] dO_i (-> dV_x)
//   Y,  wave = key block:    p[x,j], ds[x,j] of query x against the lane's key j (from the k / v registers read for P2),
//                            partial sum over the block's 32 keys of ds[x,j] k_j (-> dQ_x)
//   P2:                      dV_j += p[x,j] dO_x,  dK_j += ds[x,j] q_x
//   after barrier Y one wave adds the eight partial sums of each of dQ_x, dK_x, dV_x in a fixed order, the (x, x) term, and stores row x.
// Token x's q, k, v, dO (fp32), lse * log2(e) and delta live in a double-buffered 1 KB LDS record that wave 0 fills for the NEXT head
// while the current one is in P2.  p and ds are rounded to the 16-bit operand type where the blocks round theirs.
struct XRegs { float a, b, c; };
__device__ __forceinline__ void xrow_prefetch(XRegs& x, const void* pa, const void* pb, const void* pc) {
  asm volatile("global_load_dword %0, %3, off\n\tglobal_load_dword %1, %4, off\n\tglobal_load_dword %2, %5, off"
               : "=&v"(x.a), "=&v"(x.b), "=&v"(x.c) : "v"(pa), "v"(pb), "v"(pc) : "memory");
}
__device__ __forceinline__ void xrow_pin(XRegs& x) { asm volatile("" : "+v"(x.a), "+v"(x.b), "+v"(x.c)); }
__device__ __forceinline__ float h16_round(float v) { return (float)(bf16)v; }
// v[e], e < 32: one value per lane and index; afterwards lane l holds, in v[0], the sum over the 32 lanes of its half of index l & 31
// (butterfly: every step halves the indices a lane is responsible for; 31 exchanges instead of 160)
// the partner lane's value for the exchange distances of the butterfly: 16 = v_permlane16_swap (below), 8 = a rotation of the 16-lane
// row, 2 and 1 = quad permutations (data-parallel-primitive modifiers: vector ALU, no LDS crossbar — the kernel's LDS port is its
// scarce resource), 4 = ds_bpermute
template <int OFF>
__device__ __forceinline__ float lane_xor(float x) {
  if constexpr (OFF == 8) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xf, 0xf, false));   // row_ror:8
  else if constexpr (OFF == 2) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, false));   // quad_perm:[2,3,0,1]
  else if constexpr (OFF == 1) return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, false));   // quad_perm:[1,0,3,2]
  else return __shfl_xor(x, OFF, 64);
}
template <int OFF>
__device__ __forceinline__ void half_wave_reduce_step(float (&v)[32], int lane) {
  if constexpr (OFF == 16) {   // rows 1 / 3 of v[i] trade places with rows 0 / 2 of v[16 + i]: both halves of the exchange in one instruction
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float a = v[i], b = v[16 + i];   // (inline asm: with the builtin hipcc 7.2 added the first result to itself — seen in the .s)
      asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));   // (wait states around a lane swap whose operands a vector instruction just wrote: inline asm is not padded)
      v[i] = a + b;
    }
  } else {
    const bool up = (lane & OFF) != 0;
#pragma unroll
    for (int i = 0; i < OFF; ++i) {
      const float keep = up ? v[OFF + i] : v[i], send = up ? v[i] : v[OFF + i];
      v[i] = keep + lane_xor<OFF>(send);
    }
  }
}
__device__ __forceinline__ float half_wave_reduce32(float (&v)[32], int lane) {
  half_wave_reduce_step<16>(v, lane);
  half_wave_reduce_step<8>(v, lane);
  half_wave_reduce_step<4>(v, lane);
  half_wave_reduce_step<2>(v, lane);
  half_wave_reduce_step<1>(v, lane);
  return v[0];
}
constexpr int XA_FLOATS = 272;          // k[64] v[64] q[64] dO[64] lse2 delta (+ pad) per buffer, two buffers
constexpr int XA_BYTES = 2 * XA_FLOATS * 4 + 3 * 8 * 64 * 4;   // + the partial sums [3][8 waves][64]

// NT = 32-row blocks per sequence (compile time: the four tiles then sit at instruction-immediate distances of NT * 4 KB).
// STAGED: the dQ / dK / dV tiles leave through a per-wave 4 KB LDS buffer as whole lines (store_acc_T_staged); the eight buffers
// fit beside the four tiles up to seven blocks (N <= 224), the eight-block case stores directly.
constexpr int bwdp_waves(int nt) { return nt <= 7 ? nt + 1 : 8; }   // up to seven blocks: one wave per block + the loader wave
template <int NT, bool STAGED, bool EXTRA = false>
__global__ __launch_bounds__(64 * bwdp_waves(NT), 2) void attn_bwd_persist_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ o,
                                                                  const bf16* __restrict__ dout, const float* __restrict__ lse,
                                                                  float* __restrict__ delta, bf16* __restrict__ dqkv, int N,
                                                                  int H, float scale, int BH) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // K, V, Q, dO tiles [32 NT][64]; lse*log2e [256]; delta [256]; store buffers
  static_assert(!EXTRA || (NT == 8 && !STAGED), "the extra-token form exists for 8 blocks + 1 (257 tokens)");
  const int NSEQ = N;                  // tokens of a sequence (strides); below, N = the tokens the blocks cover
  if constexpr (EXTRA) N -= 1;
  const int tid = threadIdx.x, lane = tid & 63, h2 = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = H * 64;
  const long ld = 3L * D;
  const float c = scale * LOG2E;
  constexpr int T_OFF = NT * 4096;   // byte distance between consecutive tiles
  char* KA = smem;
  char* VA = smem + T_OFF;
  char* QB = smem + 2 * T_OFF;
  char* DB = smem + 3 * T_OFF;
  float* lses2 = (float*)(smem + 4 * T_OFF);
  float* dls = lses2 + PERSIST_MAX_ROWS;
  // store instructions per wave and 32 x 64 tile: 8 direct stores (all issued whenever the block has a valid row), or — staged —
  // one per 8 rows that hold a valid row (hipcc branches around a store no lane executes): the counted waits below must leave
  // exactly the stores that WERE issued in flight, or they return before older loads / LDS-DMA have landed
  const int tile_stores = STAGED ? ((N - wave * 32 + 7) >> 3 < 4 ? (N - wave * 32 + 7) >> 3 : 4) : 8;
  constexpr int nt = NT;             // wave w owns block w in both phases
  const bool active = wave < nt;
  int r = wave * 32 + (lane & 31);
  const bool rvalid = active && r < N;
  if (r >= N) r = N - 1;
  const FragOffs fo = frag_offs(lane);
  const unsigned ka0 = lds_addr(KA);
  char* wbuf = smem + 4 * T_OFF + 2 * PERSIST_MAX_ROWS * 4 + wave * 4096;   // STAGED: this wave's store buffer
  float* const xa_base = (float*)(smem + 4 * T_OFF + 2 * PERSIST_MAX_ROWS * 4);   // EXTRA: token x's record (two buffers), then the partial sums
  float* const xslots = xa_base + 2 * XA_FLOATS;                                  // [3: dQ_x, dK_x, dV_x][8][64]
  STAMP_DECL

  // A set of two tiles is 8 * nt pieces of 1 KB (8 rows x 128 B): wave w issues pieces [(w & 3) * nt, +nt) of the first (w < 4) or
  // second tile; the XOR swizzle of tile_off goes on the source column (the LDS side of a DMA is lane-linear)
  // (inside the product loops ONE piece is issued per tile iteration — nt iterations, nt pieces per wave —, so that the issue cost
  // of a set, 1.3-1.9k cycles per wave when issued in one go, disappears into the loop's stall slots)
  auto stage_piece = [&](int it, const bf16* srcA, long ldA, const bf16* srcB, long ldB, char* TA, char* TB) {
    const bool isb = wave >= 4;
    const bf16* src = isb ? srcB : srcA;
    const long lds_ = isb ? ldB : ldA;
    char* dst = isb ? TB : TA;
    const int pr = (wave & 3) * nt + it;
    const int row = pr * 8 + (lane >> 3);
    const int f = (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
    const int ch = (lane & 7) ^ f;
    const int gr = row < N ? row : N - 1;
    __builtin_amdgcn_global_load_lds(ATT_GLBP(src + (long)gr * lds_ + ch * 8), ATT_LDSP(dst + pr * 1024), 16, 0, 0);
  };
  auto stage = [&](const bf16* srcA, long ldA, const bf16* srcB, long ldB, char* TA, char* TB) {
#pragma unroll 1
    for (int it = 0; it < nt; ++it) stage_piece(it, srcA, ldA, srcB, ldB, TA, TB);
  };
  // Up to seven blocks wave 7 owns no block: it is the LOADER and issues every LDS-DMA piece of the kernel, the compute waves none.
  // (Round 5, stamps of the persistent forward's loader: computed per piece, the swizzle and the 64-bit row products cost ~240 cycles
  // per piece — here 14 pieces per compute wave and head inside the product loops.)  A piece's address is a wave-uniform piece base
  // plus a 32-bit lane offset that takes two values (even / odd pieces): scalar adds only between two DMA instructions.  Pieces
  // that hold rows >= N (they re-read row N - 1) take the long way: at most four per tile.
  constexpr bool LOADER = NT <= 7;
  constexpr int LW = bwdp_waves(NT) - 1;   // the loader wave (LOADER): the workgroup has NT + 1 waves, so that short sequences run
                                           // several workgroups per CU (8 waves of 238 registers fill a CU: 4 x 2, 2 x 3 or 2 x 4 waves)
  auto stage_tile_fast = [&](const bf16* src, long lds_, char* dst) {
    const int lr = lane >> 3;
    const int f_even = (((lr >> 1) & 1) << 2) | (lr >> 2);
    const unsigned loff_even = (unsigned)(lr * (int)lds_ * 2 + (((lane & 7) ^ f_even) << 4));
    const unsigned loff_odd = (unsigned)(lr * (int)lds_ * 2 + (((lane & 7) ^ f_even ^ 2) << 4));
    const int nfull = N >> 3;
    const char* pb = (const char*)src;
    const long step = 8L * lds_ * 2;
    int pr = 0;
#pragma unroll 1
    for (; pr + 1 < nfull; pr += 2) {
      __builtin_amdgcn_global_load_lds(ATT_GLBP(pb + loff_even), ATT_LDSP(dst + pr * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(ATT_GLBP(pb + step + loff_odd), ATT_LDSP(dst + pr * 1024 + 1024), 16, 0, 0);
      pb += 2 * step;
    }
#pragma unroll 1
    for (; pr < 4 * nt; ++pr) {
      const int row = pr * 8 + lr;
      const int f = (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
      const int gr = row < N ? row : N - 1;
      __builtin_amdgcn_global_load_lds(ATT_GLBP(src + (long)gr * lds_ + (((lane & 7) ^ f) << 3)), ATT_LDSP(dst + pr * 1024), 16, 0, 0);
    }
  };
  struct Head { const bf16* base; const bf16* dobase; const bf16* obase; const float* lsebase; float* dlbase; bf16* outbase; };
  auto head_of = [&](int idx) {
    const int b = idx / H, h = idx - b * H;
    Head hd;
    hd.base = qkv + (long)b * NSEQ * ld + h * 64;
    hd.dobase = dout + (long)b * NSEQ * D + h * 64;
    hd.obase = o + (long)b * NSEQ * D + h * 64;
    hd.lsebase = lse + (long)idx * NSEQ;
    hd.dlbase = delta + (long)idx * NSEQ;
    hd.outbase = dqkv + (long)b * NSEQ * ld + h * 64;
    return hd;
  };
  auto prefetch_rows = [&](RowRegs& rr, const Head& hd) {
    const int e = 8 * h2;
    rows_prefetch(rr, hd.base + (long)r * ld + e, hd.dobase + (long)r * D + e, hd.obase + (long)r * D + e, hd.lsebase + r);
  };
  // EXTRA, wave 0: token x of a head into registers (lanes 0-31: q | v | o pairs, lanes 32-63: k | dO pairs | lse), then — once landed —
  // into the record `buf` as fp32 with lse * log2(e) and delta = dO_x . o_x (also written to the delta output)
  auto xrow_request = [&](XRegs& xr, const Head& h) {
    const int pl = lane & 31;
    const bf16* rowq = h.base + (long)N * ld;           // row x = N (the blocks cover rows 0 .. N - 1)
    const void* pa = h2 ? (const void*)(rowq + D + 2 * pl) : (const void*)(rowq + 2 * pl);
    const void* pb = h2 ? (const void*)(h.dobase + (long)N * D + 2 * pl) : (const void*)(rowq + 2 * D + 2 * pl);
    const void* pc = h2 ? (const void*)(h.lsebase + N) : (const void*)(h.obase + (long)N * D + 2 * pl);
    xrow_prefetch(xr, pa, pb, pc);
  };
  auto xrow_publish = [&](const XRegs& xr, int buf, const Head& h) {
    float* xa = xa_base + buf * XA_FLOATS;
    const int pl = lane & 31;
    const bf16x2 pa = __builtin_bit_cast(bf16x2, xr.a), pb = __builtin_bit_cast(bf16x2, xr.b);
    // xa: k at 0, v at 64, q at 128, dO at 192
    float* da = xa + (h2 ? 0 : 128) + 2 * pl;
    float* db = xa + (h2 ? 192 : 64) + 2 * pl;
    da[0] = (float)pa[0]; da[1] = (float)pa[1];
    db[0] = (float)pb[0]; db[1] = (float)pb[1];
    const float oc = __shfl(xr.c, pl, 64);              // lanes 32-63 fetch the o pair of lane - 32
    const bf16x2 po = __builtin_bit_cast(bf16x2, oc);
    float dl = h2 ? (float)pb[0] * (float)po[0] + (float)pb[1] * (float)po[1] : 0.f;
#pragma unroll
    for (int o_ = 32; o_ > 0; o_ >>= 1) dl += __shfl_xor(dl, o_, 64);
    if (lane == 32) { xa[256] = xr.c * LOG2E; xa[257] = dl; h.dlbase[N] = dl; }
  };

  int idx = blockIdx.x;
  if (idx >= BH) return;
  Head hd = head_of(idx);
  if constexpr (LOADER) {
    if (wave == LW) {   // --------------------------------------------------------------------------- loader wave
      stage_tile_fast(hd.base + D, ld, KA);
      stage_tile_fast(hd.base + 2 * D, ld, VA);
      while (true) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // X: set A landed; every wave is done with set B
        stage_tile_fast(hd.base, ld, QB);
        stage_tile_fast(hd.dobase, D, DB);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // Y: set B landed; every wave is done with set A
        idx += gridDim.x;
        if (idx >= BH) break;
        hd = head_of(idx);
        stage_tile_fast(hd.base + D, ld, KA);
        stage_tile_fast(hd.base + 2 * D, ld, VA);
      }
      return;
    }
  }
  RowRegs rr;
  if (active) { prefetch_rows(rr, hd); rows_landed<0>(rr); }   // first head only: the rows are waited for on the spot
  int hb = 0;                        // EXTRA: which of the two records holds token x of the current head
  // p[x, j] and ds[x, j] of query x against this lane's key j, from the key block's k / v registers (at Y for the share of dQ_x, again
  // behind the P2 loop for dK_j / dV_j: two registers that would otherwise live through the loop, whose budget is spent)
  auto x_vs_key = [&](const bf16x8 (&kf_)[4], const bf16x8 (&vf_)[4], float& pk, float& dsk) {
    const float* xa = xa_base + hb * XA_FLOATS;
    float sx = 0.f, dpx = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const f32x4 q0 = *(const f32x4*)(xa + 128 + 16 * ks + 8 * h2), q1 = *(const f32x4*)(xa + 128 + 16 * ks + 8 * h2 + 4);
      const f32x4 d0 = *(const f32x4*)(xa + 192 + 16 * ks + 8 * h2), d1 = *(const f32x4*)(xa + 192 + 16 * ks + 8 * h2 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sx += (float)kf_[ks][j] * q0[j] + (float)kf_[ks][4 + j] * q1[j];
        dpx += (float)vf_[ks][j] * d0[j] + (float)vf_[ks][4 + j] * d1[j];
      }
      __builtin_amdgcn_sched_barrier(0);   // (one k-step's sixteen record values at a time: hoisted together they cost 64 registers)
    }
    sx += __shfl_xor(sx, 32, 64);
    dpx += __shfl_xor(dpx, 32, 64);
    const float p = ATT_EXP2(fmaf(sx, c, -xa[256]));
    pk = h16_round(p);
    dsk = h16_round(p * (dpx - xa[257]));
  };
  if constexpr (EXTRA) {
    XRegs xr;
    xrow_request(xr, hd);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    xrow_pin(xr);
    if (wave == 0) {
      xrow_publish(xr, 0, hd);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  if constexpr (!LOADER) stage(hd.base + D, ld, hd.base + 2 * D, ld, KA, VA);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  STAMP(0);   // first head: rows + set A
  while (true) {
    // ================================================================ X: set A (and, before the back edge, this wave's rows) landed
    __builtin_amdgcn_s_barrier();   // every wave is done with set B, lse and delta of the previous head
    asm volatile("" ::: "memory");
    STAMP(1);   // barrier X
    if constexpr (!LOADER) { if (!active) stage(hd.base, ld, hd.dobase, D, QB, DB); }   // a wave without a block issues its share of set B in one go
    if (active) {   // ---------------------------------------------- P1: delta and dQ of query block `wave`
      bf16x8 qf[4], dof[4];
      float dl = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        qf[ks] = __builtin_bit_cast(bf16x8, rr.q[ks]);
        dof[ks] = __builtin_bit_cast(bf16x8, rr.d[ks]);
        const bf16x8 of = __builtin_bit_cast(bf16x8, rr.o[ks]);
#pragma unroll
        for (int j = 0; j < 8; ++j) dl += (float)dof[ks][j] * (float)of[j];
      }
      dl += __shfl_xor(dl, 32, 64);
      if (rvalid && h2 == 0) hd.dlbase[r] = dl;
      const float lse2 = rr.lse * LOG2E;
      if (h2 == 0) { dls[wave * 32 + (lane & 31)] = dl; lses2[wave * 32 + (lane & 31)] = lse2; }
      f32x16 acc_dq[2];
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc_dq[0][i] = 0.f; acc_dq[1][i] = 0.f; }
      STAMP(2);   // P1 start: delta
#pragma unroll 1
      for (int t = 0; t < nt; ++t) {
        if constexpr (!LOADER) stage_piece(t, hd.base, ld, hd.dobase, D, QB, DB);   // set B, one piece per iteration
        const unsigned tb = ka0 + t * 4096;
        unsigned ar[4], at[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { ar[k] = tb + fo.rf[k]; at[k] = tb + fo.tr[k]; }
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        f32x16 s = MFMA_F32_32x32x16_H16(row_frag_at<0>(ar[0]), qf[0], zero);
        f32x16 dp = MFMA_F32_32x32x16_H16(row_frag_at<T_OFF>(ar[0]), dof[0], zero);
#pragma unroll
        for (int ks = 1; ks < 4; ++ks) {
          s = MFMA_F32_32x32x16_H16(row_frag_at<0>(ar[ks]), qf[ks], s);
          dp = MFMA_F32_32x32x16_H16(row_frag_at<T_OFF>(ar[ks]), dof[ks], dp);
        }
        TrPair kt_[4];
        tr_issue_at<0>(kt_[0], at[0], at[1]);
        tr_issue_at<0>(kt_[1], at[2], at[3]);
        tr_issue_at<2048>(kt_[2], at[0], at[1]);
        tr_issue_at<2048>(kt_[3], at[2], at[3]);
        if (t * 32 + 32 <= N) {
#pragma unroll
          for (int i = 0; i < 16; ++i) s[i] = ATT_EXP2(fmaf(s[i], c, -lse2)) * (dp[i] - dl);
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            float p = ATT_EXP2(fmaf(s[i], c, -lse2));
            if (t * 32 + acc_row(i, h2) >= N) p = 0.f;
            s[i] = p * (dp[i] - dl);
          }
        }
        lds_landed(kt_);
#pragma unroll
        for (int sk = 0; sk < 2; ++sk) {
          const bf16x8 dsb = acc_to_operand(s, sk);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)
            acc_dq[dt] = MFMA_F32_32x32x16_H16(tr_join(kt_[2 * sk + dt]), dsb, acc_dq[dt]);
        }
      }
      STAMP(3);   // P1 loop
      float pq = 0.f, dsq = 0.f;     // EXTRA: p[i, x], ds[i, x] of this lane's query
      if constexpr (EXTRA) {
        const float* xa = xa_base + hb * XA_FLOATS;
        float sx = 0.f, dpx = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const f32x4 k0 = *(const f32x4*)(xa + 16 * ks + 8 * h2), k1 = *(const f32x4*)(xa + 16 * ks + 8 * h2 + 4);
          const f32x4 v0 = *(const f32x4*)(xa + 64 + 16 * ks + 8 * h2), v1 = *(const f32x4*)(xa + 64 + 16 * ks + 8 * h2 + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            sx += (float)qf[ks][j] * k0[j] + (float)qf[ks][4 + j] * k1[j];
            dpx += (float)dof[ks][j] * v0[j] + (float)dof[ks][4 + j] * v1[j];
          }
        }
        sx += __shfl_xor(sx, 32, 64);
        dpx += __shfl_xor(dpx, 32, 64);
        const float p = ATT_EXP2(fmaf(sx, c, -lse2));
        pq = h16_round(p);
        dsq = h16_round(p * (dpx - dl));
      }
      if constexpr (STAGED) store_acc_T_staged(acc_dq, wbuf, hd.outbase, ld, wave * 32, N, lane, scale);
      else if constexpr (EXTRA) store_acc_T_rank1(acc_dq, hd.outbase + (long)r * ld, h2, scale, dsq, xa_base + hb * XA_FLOATS);   // dQ_i += ds[i,x] k_x
      else if (rvalid) store_acc_T(acc_dq, hd.outbase + (long)r * ld, h2, scale);
      STAMP(4);   // dQ stores
      if constexpr (EXTRA) {   // this block's share of dK_x = sum_i ds[i,x] q_i and dV_x = sum_i p[i,x] dO_i
        const int e_ = lane & 31, dd = 16 * (e_ >> 3) + 8 * h2 + (e_ & 7);
        float v[32];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int j = 0; j < 8; ++j) v[8 * ks + j] = dsq * (float)qf[ks][j];
        xslots[(8 + wave) * 64 + dd] = half_wave_reduce32(v, lane);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int j = 0; j < 8; ++j) v[8 * ks + j] = pq * (float)dof[ks][j];
        xslots[(16 + wave) * 64 + dd] = half_wave_reduce32(v, lane);
      }
    }
    // ================================================================ Y: own k / v rows out of set A, set B has landed
    bf16x8 kf[4], vf[4];
    if (active) {
      const unsigned a0 = ka0 + wave * 4096;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) { kf[ks] = row_frag_at<0>(a0 + fo.rf[ks]); vf[ks] = row_frag_at<T_OFF>(a0 + fo.rf[ks]); }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(kf[ks]), "+v"(vf[ks]));
      if constexpr (EXTRA) {   // query x against this lane's key: p[x,j], ds[x,j]; this block's share of dQ_x = sum_j ds[x,j] k_j
        float pxk, dsxk;
        x_vs_key(kf, vf, pxk, dsxk);
        const int e_ = lane & 31, dd = 16 * (e_ >> 3) + 8 * h2 + (e_ & 7);
        float v[32];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int j = 0; j < 8; ++j) v[8 * ks + j] = dsxk * (float)kf[ks][j];
        xslots[wave * 64 + dd] = half_wave_reduce32(v, lane);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the partial sums of this wave (P1's two as well) are in LDS before barrier Y
      }
      // the DMA of set B is older than the dQ stores (with a loader wave this wave has no DMA to wait for)
      if constexpr (!LOADER) { if constexpr (STAGED) wait_vmcnt_upto4(tile_stores); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    STAMP(5);   // wait for set B
    __builtin_amdgcn_s_barrier();   // every wave is done with set A; set B, lse and delta are complete
    asm volatile("" ::: "memory");
    STAMP(6);   // barrier Y
    if constexpr (EXTRA) {
      if (wave == NT - 1) {   // row x of dQ, dK, dV: the eight partial sums in wave order, the (x, x) term, one store each; lane = column
        const float* xa = xa_base + hb * XA_FLOATS;
        const float kq = xa[lane], vq = xa[64 + lane], qq = xa[128 + lane], dq_ = xa[192 + lane];
        float sxx = qq * kq, dpxx = dq_ * vq;
#pragma unroll
        for (int o_ = 32; o_ > 0; o_ >>= 1) { sxx += __shfl_xor(sxx, o_, 64); dpxx += __shfl_xor(dpxx, o_, 64); }
        const float p = ATT_EXP2(fmaf(sxx, c, -xa[256]));
        const float pb = h16_round(p), dsb = h16_round(p * (dpxx - xa[257]));
        float gq = dsb * kq, gk = dsb * qq, gv = pb * dq_;
#pragma unroll
        for (int w = 0; w < 8; ++w) { gq += xslots[w * 64 + lane]; gk += xslots[(8 + w) * 64 + lane]; gv += xslots[(16 + w) * 64 + lane]; }
        bf16* orow = hd.outbase + (long)N * ld + lane;
        orow[0] = (bf16)(gq * scale);
        orow[D] = (bf16)(gk * scale);
        orow[2 * D] = (bf16)gv;
      }
    }
    const int next = idx + gridDim.x;
    const bool has_next = next < BH;
    Head hn = hd;
    if (has_next) {
      hn = head_of(next);
      if constexpr (!LOADER) { if (!active) stage(hn.base + D, ld, hn.base + 2 * D, ld, KA, VA); }
    }
    if (active) {   // ---------------------------------------------- P2: dK and dV of key block `wave`
      f32x16 acc_dk[2], acc_dv[2];
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc_dk[0][i] = 0.f; acc_dk[1][i] = 0.f; acc_dv[0][i] = 0.f; acc_dv[1][i] = 0.f; }
      STAMP(7);   // P2 start
#pragma unroll 1
      for (int t = 0; t < nt; ++t) {
        if constexpr (!LOADER) { if (has_next) stage_piece(t, hn.base + D, ld, hn.base + 2 * D, ld, KA, VA); }   // set A of the next head, one piece per iteration
        const unsigned tb = ka0 + 2 * T_OFF + t * 4096;   // Q tile, row 32t; the dO tile is T_OFF further
        unsigned ar[4], at[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { ar[k] = tb + fo.rf[k]; at[k] = tb + fo.tr[k]; }
        f32x16 s, dp;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          s = MFMA_F32_32x32x16_H16(row_frag_at<0>(ar[ks]), kf[ks], s);
          dp = MFMA_F32_32x32x16_H16(row_frag_at<T_OFF>(ar[ks]), vf[ks], dp);
        }
        TrPair tf[8];   // [4 sk + dt] = dO^T, [4 sk + 2 + dt] = Q^T
        tr_issue_at<T_OFF>(tf[0], at[0], at[1]);
        tr_issue_at<0>(tf[2], at[0], at[1]);
        tr_issue_at<T_OFF>(tf[1], at[2], at[3]);
        tr_issue_at<0>(tf[3], at[2], at[3]);
        tr_issue_at<T_OFF + 2048>(tf[4], at[0], at[1]);
        tr_issue_at<2048>(tf[6], at[0], at[1]);
        tr_issue_at<T_OFF + 2048>(tf[5], at[2], at[3]);
        tr_issue_at<2048>(tf[7], at[2], at[3]);
        if (t * 32 + 32 <= N) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int r0 = t * 32 + 8 * g + 4 * h2;
            const f32x4 l4 = *(const f32x4*)(lses2 + r0), d4 = *(const f32x4*)(dls + r0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int i = 4 * g + e;
              const float p = ATT_EXP2(fmaf(s[i], c, -l4[e]));
              s[i] = p;
              dp[i] = p * (dp[i] - d4[e]);
            }
          }
        } else {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int r0 = t * 32 + 8 * g + 4 * h2;
            const f32x4 l4 = *(const f32x4*)(lses2 + r0), d4 = *(const f32x4*)(dls + r0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int i = 4 * g + e;
              float p = ATT_EXP2(fmaf(s[i], c, -l4[e]));
              if (r0 + e >= N) p = 0.f;
              s[i] = p;
              dp[i] = p * (dp[i] - d4[e]);
            }
          }
        }
        lds_landed(tf);
#pragma unroll
        for (int sk = 0; sk < 2; ++sk) {
          const bf16x8 pb = acc_to_operand(s, sk), dsb = acc_to_operand(dp, sk);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            acc_dv[dt] = MFMA_F32_32x32x16_H16(tr_join(tf[4 * sk + dt]), pb, acc_dv[dt]);
            acc_dk[dt] = MFMA_F32_32x32x16_H16(tr_join(tf[4 * sk + 2 + dt]), dsb, acc_dk[dt]);
          }
        }
      }
      STAMP(8);   // P2 loop
      // rows of the next head: requested BEFORE the dK / dV stores (since the loader wave took the DMA out of this loop its registers
      // have room for them), i.e. older than both tiles' stores in the vmcnt queue, and waited for behind them.  Unconditional (the
      // last head re-reads its own rows, never used): a conditional definition would keep the previous head's 49 registers alive
      // through P2.
      float pxk = 0.f, dsxk = 0.f;   // EXTRA: query x's share of this block's keys, added on the way out: dV_j += p[x,j] dO_x, dK_j += ds[x,j] q_x
      if constexpr (EXTRA) x_vs_key(kf, vf, pxk, dsxk);
      asm volatile("" ::: "memory");
      XRegs xr;
      if constexpr (EXTRA) xrow_request(xr, hn);   // token x of the next head (every wave asks, wave 0 uses it): older than the rows, landed with them
      prefetch_rows(rr, hn);
      if constexpr (STAGED) store_acc_T_staged(acc_dk, wbuf, hd.outbase + D, ld, wave * 32, N, lane, scale);
      else if constexpr (EXTRA) store_acc_T_rank1(acc_dk, hd.outbase + (long)r * ld + D, h2, scale, dsxk, xa_base + hb * XA_FLOATS + 128);
      else if (rvalid) store_acc_T(acc_dk, hd.outbase + (long)r * ld + D, h2, scale);
      if constexpr (STAGED) store_acc_T_staged(acc_dv, wbuf, hd.outbase + 2 * D, ld, wave * 32, N, lane, 1.0f);
      else if constexpr (EXTRA) store_acc_T_rank1(acc_dv, hd.outbase + (long)r * ld + 2 * D, h2, 1.0f, pxk, xa_base + hb * XA_FLOATS + 192);
      else if (rvalid) store_acc_T(acc_dv, hd.outbase + (long)r * ld + 2 * D, h2, 1.0f);
      STAMP(9);   // row prefetch + dK / dV stores issued
      // the rows (and, without a loader wave, set A of the next head: older still) have landed; the dK / dV stores stay in flight
      if constexpr (STAGED) { wait_vmcnt_upto8(2 * tile_stores); rows_pin(rr); } else rows_landed<16>(rr);
      STAMP(10);  // wait for rows / set A
      if constexpr (EXTRA) {
        xrow_pin(xr);
        if (wave == 0 && has_next) {   // into the OTHER record: the waves still in P2 read the current one
          xrow_publish(xr, hb ^ 1, hn);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
      }
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (!has_next) break;
    idx = next;
    hd = hn;
    if constexpr (EXTRA) hb ^= 1;
  }
#if defined(APLA_ATT_STAMPS)
  if (lane == 0 && blockIdx.x < 512) for (int k_ = 0; k_ < 16; ++k_) apla_att_dbg[(blockIdx.x * 8 + wave) * 16 + k_] = st_acc[k_];
#endif
}

// ------------------------------------------------------------------------------------------------ attention matrix on demand
// attn[b,h,q,k] = exp(q·k*scale - lse[q]).  Visualisation path only (Block.forward(return_attention=True)); plain VALU.
// DROP: the matrix the reference returns in training mode — after its dropout (appla_attn.py:56-58, 83): kept entries / (1 - p),
// the others 0, with the keep decision of the forward / backward kernels (drop_keep4 of (row, key group)).
template <bool DROP>
__global__ __launch_bounds__(256) void attn_probs_kernel(const bf16* __restrict__ qkv, const float* __restrict__ lse,
                                                         float* __restrict__ attn, int N, int H, float scale, DropArgs da) {
  const int b = blockIdx.z, h = blockIdx.y, q = blockIdx.x;
  const int D = H * 64;
  const long ld = 3L * D;
  const bf16* qrow = qkv + ((long)b * N + q) * ld + h * 64;
  __shared__ float qs[64];
  if (threadIdx.x < 64) qs[threadIdx.x] = (float)qrow[threadIdx.x];
  __syncthreads();
  const long row = ((long)b * H + h) * N + q;
  const float l = lse[row];
  for (int k = threadIdx.x; k < N; k += 256) {
    const bf16* krow = qkv + ((long)b * N + k) * ld + D + h * 64;
    float acc = 0.f;
#pragma unroll
    for (int d8 = 0; d8 < 8; ++d8) {
      const bf16x8 kv = *(const bf16x8*)(krow + d8 * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += qs[d8 * 8 + j] * (float)kv[j];
    }
    float pr = __expf(acc * scale - l);
    if constexpr (DROP) pr = ((drop_keep4(da, (unsigned long long)row, (unsigned)k >> 2) >> (k & 3)) & 1) ? pr * da.inv_keep : 0.f;
    attn[row * N + k] = pr;
  }
}

// ------------------------------------------------------------------------------------------------ forward, CLS query only
// Last block of the ViT: the classifier reads x[:, 0] after the final norm (vit.py:416-419), so of this block's attention
// output only the CLS row of every sequence is ever used.  One query against all keys: a rank-1, HBM-bound pass over K
// and V.  One workgroup per (b, h); lane = head-dim element; each wave walks keys wave, wave+4, ... with its own online
// softmax state, and the four states are merged in a fixed order.  Writes o[b*N] (this head's 64 columns) and lse[b, h, 0].
__global__ __launch_bounds__(256) void attn_fwd_cls_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ o,
                                                           float* __restrict__ lse, int N, int H, float scale) {
  __shared__ float sm[4], sl[4], sacc[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y, h = blockIdx.x;
  const int D = H * 64;
  const long ld = 3L * D;
  const bf16* base = qkv + (long)b * N * ld + h * 64;
  const float q0 = (float)base[lane] * scale;
  float m = -INFINITY, l = 0.f, acc = 0.f;
  auto fold = [&](float s, float vv) {      // online softmax update with one key
    const float mn = fmaxf(m, s);
    const float a = __expf(m - mn), pr = __expf(s - mn);
    l = l * a + pr;
    acc = acc * a + pr * vv;
    m = mn;
  };
  // four keys of this wave per iteration: their eight row loads are issued together and the four wave-wide sums are independent
  // chains (one key at a time, every load waited for the previous key's sum and exp: the pass was latency-bound, not HBM-bound)
  int k = wave;
  for (; k + 12 < N; k += 16) {
    float kv[4], vv[4], s[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      kv[u] = (float)base[(long)(k + 4 * u) * ld + D + lane];
      vv[u] = (float)base[(long)(k + 4 * u) * ld + 2 * D + lane];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) s[u] = q0 * kv[u];
#pragma unroll
    for (int o_ = 32; o_ > 0; o_ >>= 1)
#pragma unroll
      for (int u = 0; u < 4; ++u) s[u] += __shfl_xor(s[u], o_, 64);
#pragma unroll
    for (int u = 0; u < 4; ++u) fold(s[u], vv[u]);
  }
  for (; k < N; k += 4) {
    const float kv = (float)base[(long)k * ld + D + lane], vv = (float)base[(long)k * ld + 2 * D + lane];
    fold(wave_sum(q0 * kv), vv);
  }
  if (lane == 0) { sm[wave] = m; sl[wave] = l; }
  sacc[wave][lane] = acc;
  __syncthreads();
  if (wave == 0) {
    const float M = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    float L = 0.f, O = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float e = __expf(sm[w] - M);  // a wave that saw no key has m = -inf, l = 0: contributes nothing
      L += sl[w] * e;
      O += sacc[w][lane] * e;
    }
    o[(long)b * N * D + h * 64 + lane] = (bf16)(O / L);
    if (lane == 0) lse[((long)b * H + h) * N] = M + __logf(L);
  }
}

// ------------------------------------------------------------------------------------------------ backward, CLS-only dO
// Last block of the ViT: only the CLS query (row 0 of every sequence) carries gradient into the attention output (final
// norm + x[:,0], vit.py:416-419).  Then dV[k] = P[0,k] dO[0], dK[k] = scale dS[0,k] q[0], dQ[0] = scale sum_k dS[0,k] K[k]
// and dQ[q>0] = 0: a rank-1, HBM-bound pass over K and V instead of the 7-product flash backward.
// One workgroup per (b,h); lane = head-dim element; each wave walks keys wave, wave+4, ...
__global__ __launch_bounds__(256) void attn_bwd_cls_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ o,
                                                           const bf16* __restrict__ do_cls, const float* __restrict__ lse,
                                                           bf16* __restrict__ dqkv, int N, int H, float scale) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y, h = blockIdx.x;
  const int D = H * 64;
  const long ld = 3L * D;
  const bf16* base = qkv + (long)b * N * ld + h * 64;
  bf16* dbase = dqkv + (long)b * N * ld + h * 64;
  const float q0 = (float)base[lane];
  const float do0 = (float)do_cls[(long)b * D + h * 64 + lane];
  const float o0 = (float)o[(long)b * N * D + h * 64 + lane];
  const float delta = wave_sum(do0 * o0);
  const float l0 = lse[((long)b * H + h) * N];
  float dq = 0.f;
  auto one = [&](int k, float kv, float s, float dp) {
    const float pr = __expf(s * scale - l0);
    const float ds = pr * (dp - delta) * scale;
    dq += ds * kv;
    bf16* orow = dbase + (long)k * ld;
    orow[D + lane] = (bf16)(ds * q0);
    orow[2 * D + lane] = (bf16)(pr * do0);
    if (k > 0) orow[lane] = (bf16)0.f;
  };
  // four keys of this wave per iteration (eight loads in flight, eight independent wave-wide sums): see attn_fwd_cls_kernel
  int k = wave;
  for (; k + 12 < N; k += 16) {
    float kv[4], s[4], dp[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      kv[u] = (float)base[(long)(k + 4 * u) * ld + D + lane];
      const float vv = (float)base[(long)(k + 4 * u) * ld + 2 * D + lane];
      s[u] = q0 * kv[u];
      dp[u] = do0 * vv;
    }
#pragma unroll
    for (int o_ = 32; o_ > 0; o_ >>= 1)
#pragma unroll
      for (int u = 0; u < 4; ++u) { s[u] += __shfl_xor(s[u], o_, 64); dp[u] += __shfl_xor(dp[u], o_, 64); }
#pragma unroll
    for (int u = 0; u < 4; ++u) one(k + 4 * u, kv[u], s[u], dp[u]);
  }
  for (; k < N; k += 4) {
    const float kv = (float)base[(long)k * ld + D + lane], vv = (float)base[(long)k * ld + 2 * D + lane];
    one(k, kv, wave_sum(q0 * kv), wave_sum(do0 * vv));
  }
  red[wave][lane] = dq;
  __syncthreads();
  if (wave == 0) dbase[lane] = (bf16)(red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]);
}

}  // namespace

// Per-device facts.  A process may drive several devices (one engine per device): the CU count and the "dynamic LDS limit raised"
// marks are kept per device ordinal (hipFuncSetAttribute acts on the current device only), never process-wide.
constexpr int APLA_MAX_DEVICES = 64;
static int apla_current_device() {
  int dev = 0;
  return (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < APLA_MAX_DEVICES) ? dev : -1;
}
// number of CUs of the current device (persistent kernels launch one workgroup per CU); queried once per device
static int apla_num_cus() {
  static std::atomic<int> n[APLA_MAX_DEVICES];
  const int dev = apla_current_device();
  int v = dev >= 0 ? n[dev].load(std::memory_order_relaxed) : 0;
  if (v == 0) {
    int q = 0;
    v = (dev >= 0 && hipDeviceGetAttribute(&q, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && q > 0) ? q : 256;
    if (dev >= 0) n[dev].store(v, std::memory_order_relaxed);
  }
  return v;
}
// raise a kernel's dynamic-LDS limit once per device (`mask`: one bit per device ordinal, one mask per kernel)
static void apla_allow_lds(std::atomic<unsigned long long>& mask, const void* kern, int bytes) {
  const int dev = apla_current_device();
  if (dev >= 0 && (mask.load(std::memory_order_relaxed) >> dev) & 1ull) return;
  (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (dev >= 0) mask.fetch_or(1ull << dev, std::memory_order_relaxed);
}

// kernel choice, a per-call argument of the *_ex entry points: 0 = auto, 1 = always the blocked kernels, 2 = the
// one-workgroup-per-head short-sequence kernels (never the persistent one), 3 = the persistent backward wherever it applies

// B sequences of (at most) N tokens; cu == nullptr: uniform batch, else packed with cu[B+1] token offsets and `total` tokens
// Which kernel a launch takes (ONE decision code for the launch and for apla_attn_kernel_name).  `packed`: block-diagonal batch.
enum AttnKernel { ATTN_FWD_BLOCKED, ATTN_FWD_SMALL, ATTN_FWD_PERSIST, ATTN_BWD_SPLIT, ATTN_BWD_SMALL, ATTN_BWD_TINY, ATTN_BWD_PERSIST, ATTN_FWD_PERSIST_BLOCKS };
static const char* const ATTN_KERNEL_NAMES[] = {"attn_fwd_kernel", "attn_fwd_small_kernel", "attn_fwd_persist_kernel", "attn_bwd_dq_kernel + attn_bwd_dkv_kernel",
                                                "attn_bwd_small_kernel<288,4>", "attn_bwd_small_kernel<64,2>", "attn_bwd_persist_kernel", "attn_fwd_persist_blocks_kernel"};
static AttnKernel attn_fwd_choice(bool packed, int B, int N, int H, int variant) {
  // uniform batch of short sequences with at least one head per CU: the persistent kernel (variant 2 pins the
  // one-workgroup-per-head kernel, variant 3 the persistent one wherever it applies).  Measured against the one-workgroup-per-head
  // kernel at 6 144 heads: 50 tokens 47.1 -> 32.5 us, 37 tokens 36.1 -> 24.5, 65 tokens 68.9 -> 44.6, 96 tokens 83.2 -> 59.8 (up to
  // 96 tokens several workgroups share a CU: fwdp_grid; one per CU 45.0 / 42.1 / 55.6 / 64.5)
  if (!packed && N <= FWDP_MAX_ROWS && variant != 1 && variant != 2 && (variant == 3 || (long)B * H >= apla_num_cus()))
    return N <= 224 ? ATTN_FWD_PERSIST : ATTN_FWD_PERSIST_BLOCKS;   // one wave per 32-row block up to 7 blocks, four waves walking 8 or 9
  return (N <= SMALL_MAX_ROWS && variant != 1) ? ATTN_FWD_SMALL : ATTN_FWD_BLOCKED;
}
static AttnKernel attn_bwd_choice(bool packed, int B, int N, int H, int variant) {
  // uniform batch of short sequences with at least one head per CU: the persistent kernel (every load one phase ahead of its use);
  // variant 2 pins the one-workgroup-per-head kernels, variant 3 the persistent one wherever it applies (tests, A/B timing)
  if (!packed && (N <= PERSIST_MAX_ROWS || N == PERSIST_MAX_ROWS + 1) && variant != 1 && variant != 2 && (variant == 3 || (long)B * H >= apla_num_cus()))
    return ATTN_BWD_PERSIST;   // (257 tokens: eight blocks + the last token as rank-1 corrections, attn_bwd_persist_kernel<8, false, true>)
  if (N <= TINY_MAX_ROWS && variant != 1) return ATTN_BWD_TINY;
  if (N <= SMALL_MAX_ROWS_BWD && variant != 1) return ATTN_BWD_SMALL;
  return ATTN_BWD_SPLIT;
}

// Grid of a persistent launch (forward and backward): the workgroups that are resident at once (occupancy of the kernel at its LDS size x CUs),
// never more than heads.  The occupancy query is a host-side table lookup of the runtime (no stream operation: safe under capture);
// its answers are kept per (kernel, LDS size).
static int fwdp_grid(const void* kern, int threads, size_t lds, int BH) {
  static std::mutex mu;
  static std::map<std::pair<const void*, size_t>, int> per_cu;
  int k = 0;
  {
    std::lock_guard<std::mutex> g(mu);
    auto it = per_cu.find({kern, lds});
    if (it != per_cu.end()) k = it->second;
  }
  if (k == 0) {
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&k, kern, threads, lds) != hipSuccess || k < 1) { (void)hipGetLastError(); k = 1; }
    if (k > 8) k = 8;
    std::lock_guard<std::mutex> g(mu);
    per_cu[{kern, lds}] = k;
  }
  const long g = (long)apla_num_cus() * k;
  return (int)(BH < g ? BH : g);
}

// Dynamic LDS of a persistent-forward instantiation at the LONGEST sequence it serves (NR = N rounded up to 8 rows varies inside one
// block count: 197 and 224 tokens are both NT = 7).  The limit is raised once per (instantiation, device), so it must be the maximum.
static constexpr int fwdp_max_lds(int nt, int store_waves) {
  return (4 * (32 * nt < FWDP_MAX_ROWS ? 32 * nt : FWDP_MAX_ROWS) + FWDP_PAD_ROWS) * 128 + store_waves * 4096;
}

static int launch_attn_fwd(const void* qkv, void* o, float* lse, const int32_t* cu, int total, int B, int N, int H,
                           float scale, int g_attn_variant, hipStream_t stream, const char* who) {
  g_attn_variant &= 0xff;
  APLA_REQUIRE(g_attn_variant >= 0 && g_attn_variant <= 3, "%s: unknown kernel variant %d", who, g_attn_variant);
  const AttnKernel kchoice = attn_fwd_choice(cu != nullptr, B, N, H, g_attn_variant);
  if (kchoice == ATTN_FWD_PERSIST || kchoice == ATTN_FWD_PERSIST_BLOCKS) {
    const int nt = (N + 31) / 32, BH = B * H, NR = (N + 7) / 8 * 8;
    const size_t lds = (size_t)(4 * NR + FWDP_PAD_ROWS) * 128 + (size_t)(nt <= 7 ? nt : FWDP_BLOCK_WAVES) * 4096;   // K0 K1 V0 V1, zero rows, per-wave store buffers
    int G = 0;   // workgroups: as many per CU as fit (short sequences: up to eight small workgroups), each walking BH / G heads
#define APLA_FWDPB_CASE(NTV)                                                                                                         \
    case NTV: {                                                                                                                      \
      auto kern = attn_fwd_persist_blocks_kernel<NTV, FWDP_BLOCK_WAVES>;                                                             \
      static std::atomic<unsigned long long> lds_ok{0};                                                                              \
      apla_allow_lds(lds_ok, (const void*)kern, fwdp_max_lds(NTV, FWDP_BLOCK_WAVES));   /* the instantiation's maximum, not this call's */ \
      G = fwdp_grid((const void*)kern, 64 * (FWDP_BLOCK_WAVES + 1), lds, BH);                                                        \
      hipLaunchKernelGGL(kern, dim3(G), dim3(64 * (FWDP_BLOCK_WAVES + 1)), lds, stream, (const bf16*)qkv, (bf16*)o, lse, N, H, scale, BH, NR); \
    } break;
#define APLA_FWDP_CASE(NTV)                                                                                                          \
    case NTV: {                                                                                                                      \
      auto kern = attn_fwd_persist_kernel<NTV>;                                                                                      \
      static std::atomic<unsigned long long> lds_ok{0};                                                                              \
      apla_allow_lds(lds_ok, (const void*)kern, fwdp_max_lds(NTV, NTV));                                                             \
      G = fwdp_grid((const void*)kern, 64 * (NTV + 1), lds, BH);                                                                     \
      hipLaunchKernelGGL(kern, dim3(G), dim3(64 * (NTV + 1)), lds, stream, (const bf16*)qkv, (bf16*)o, lse, N, H, scale, BH, NR);     \
    } break;
    switch (nt) {
      APLA_FWDP_CASE(1) APLA_FWDP_CASE(2) APLA_FWDP_CASE(3) APLA_FWDP_CASE(4) APLA_FWDP_CASE(5) APLA_FWDP_CASE(6) APLA_FWDP_CASE(7) APLA_FWDPB_CASE(8) APLA_FWDPB_CASE(9)
      default: apla_set_error("%s: bad block count %d", who, nt); return APLA_EINVAL;
    }
#undef APLA_FWDP_CASE
#undef APLA_FWDPB_CASE
    APLA_CHECK_LAUNCH(who);
    return APLA_OK;
  }
  if (kchoice == ATTN_FWD_SMALL) {
    const int nw = (N + 31) / 32;
    static std::atomic<unsigned long long> lds_ok{0};   // 72 KB of dynamic LDS at nine blocks
    apla_allow_lds(lds_ok, (const void*)attn_fwd_small_kernel, SMALL_MAX_ROWS * 256);
    hipLaunchKernelGGL(attn_fwd_small_kernel, dim3(H, B), dim3(64 * nw), (size_t)nw * 32 * 256, stream, (const bf16*)qkv, (bf16*)o, lse, N, H, scale, cu, total);
  } else {
    hipLaunchKernelGGL(attn_fwd_kernel<false>, dim3((N + 127) / 128, H, B), dim3(256), 0, stream, (const bf16*)qkv, (bf16*)o, lse, N, H, scale, cu, total, DropArgs{});
  }
  APLA_CHECK_LAUNCH(who);
  return APLA_OK;
}

static int launch_attn_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta, void* dqkv,
                           const int32_t* cu, int total, int B, int N, int H, float scale, int g_attn_variant,
                           hipStream_t stream, const char* who) {
  g_attn_variant &= 0xff;   // (bits 8.. carried schedule experiments in round 5: static / alternating wave priorities and a half-tile
                            // stagger of waves 4-7 changed nothing, profiles/r05_attn_experiments.md)
  APLA_REQUIRE(g_attn_variant >= 0 && g_attn_variant <= 3, "%s: unknown kernel variant %d", who, g_attn_variant);
  const AttnKernel kchoice = attn_bwd_choice(cu != nullptr, B, N, H, g_attn_variant);
  if (kchoice == ATTN_BWD_PERSIST && N == PERSIST_MAX_ROWS + 1) {
    const int BH = B * H;
    const size_t lds = (size_t)8 * 4096 * 4 + 2 * PERSIST_MAX_ROWS * 4 + XA_BYTES;
    auto kern = attn_bwd_persist_kernel<8, false, true>;
    static std::atomic<unsigned long long> lds_ok{0};
    apla_allow_lds(lds_ok, (const void*)kern, (int)lds);
    const int G = fwdp_grid((const void*)kern, 64 * bwdp_waves(8), lds, BH);
    hipLaunchKernelGGL(kern, dim3(G), dim3(64 * bwdp_waves(8)), lds, stream, (const bf16*)qkv, (const bf16*)o, (const bf16*)d_o, lse, delta, (bf16*)dqkv, N, H, scale, BH);
    APLA_CHECK_LAUNCH(who);
    return APLA_OK;
  }
  if (kchoice == ATTN_BWD_PERSIST) {
    const int NP = (N + 31) / 32 * 32, BH = B * H;
    const int nt = NP / 32;
    const bool staged = nt <= 7;       // 4 tiles + 2 KB + 8 x 4 KB of store buffers: 162 KB at eight blocks, one CU has 160
    const size_t lds = (size_t)nt * 4096 * 4 + 2 * PERSIST_MAX_ROWS * 4 + (staged ? (size_t)nt * 4096 : 0);   // (a store buffer per compute wave)
    int G = 0;
#define APLA_PERSIST_CASE(NTV)                                                                                                       \
    case NTV: {                                                                                                                      \
      auto kern = attn_bwd_persist_kernel<NTV, (NTV <= 7)>;                                                                          \
      static std::atomic<unsigned long long> lds_ok{0};                                                                              \
      apla_allow_lds(lds_ok, (const void*)kern, (int)lds);                                                                           \
      G = fwdp_grid((const void*)kern, 64 * bwdp_waves(NTV), lds, BH);                                                               \
      hipLaunchKernelGGL(kern, dim3(G), dim3(64 * bwdp_waves(NTV)), lds, stream, (const bf16*)qkv, (const bf16*)o, (const bf16*)d_o, lse, delta, (bf16*)dqkv, N, H, scale, BH); \
    } break;
    switch (nt) {
      APLA_PERSIST_CASE(1) APLA_PERSIST_CASE(2) APLA_PERSIST_CASE(3) APLA_PERSIST_CASE(4)
      APLA_PERSIST_CASE(5) APLA_PERSIST_CASE(6) APLA_PERSIST_CASE(7) APLA_PERSIST_CASE(8)
      default: apla_set_error("%s: bad block count %d", who, nt); return APLA_EINVAL;
    }
#undef APLA_PERSIST_CASE
    APLA_CHECK_LAUNCH(who);
    return APLA_OK;
  }
  if (kchoice == ATTN_BWD_TINY) {       // the same kernel at 64 rows and two waves (see the kernel)
    const int NP = (N + 31) / 32 * 32;
    const size_t lds = (size_t)TINY_MAX_ROWS * (2 * 128 + 8);
    hipLaunchKernelGGL((attn_bwd_small_kernel<TINY_MAX_ROWS, 2>), dim3(H, B), dim3(128), lds, stream, (const bf16*)qkv, (const bf16*)o, (const bf16*)d_o, lse, delta, (bf16*)dqkv, N, H, scale, cu, total, NP);
    APLA_CHECK_LAUNCH(who);
    return APLA_OK;
  }
  if (kchoice == ATTN_BWD_SMALL) {  // one workgroup per head, operands read from HBM once (see the kernel)
    const int NP = (N + 31) / 32 * 32;
    const size_t lds = (size_t)SMALL_MAX_ROWS_BWD * (2 * 128 + 8);   // two tiles at a fixed distance + lse*log2e + delta
    static std::atomic<unsigned long long> lds_ok{0};   // > 64 KB of dynamic LDS
    auto kern = attn_bwd_small_kernel<SMALL_MAX_ROWS_BWD, 4>;
    apla_allow_lds(lds_ok, (const void*)kern, (int)lds);
    hipLaunchKernelGGL(kern, dim3(H, B), dim3(256), lds, stream, (const bf16*)qkv, (const bf16*)o, (const bf16*)d_o, lse, delta, (bf16*)dqkv, N, H, scale, cu, total, NP);
    APLA_CHECK_LAUNCH(who);
    return APLA_OK;
  }
  dim3 grid((N + 127) / 128, H, B);
  hipLaunchKernelGGL(attn_bwd_dq_kernel<false>, grid, dim3(256), 0, stream, (const bf16*)qkv, (const bf16*)o, (const bf16*)d_o, lse, delta, (bf16*)dqkv, N, H, scale, cu, total, DropArgs{});
  APLA_CHECK_LAUNCH(who);
  hipLaunchKernelGGL(attn_bwd_dkv_kernel<false>, grid, dim3(256), 0, stream, (const bf16*)qkv, (const bf16*)d_o, lse, delta, (bf16*)dqkv, N, H, scale, cu, total, DropArgs{});
  APLA_CHECK_LAUNCH(who);
  return APLA_OK;
}

extern "C" int apla_attn_kernel_name(int backward, int packed, int B, int N, int H, int variant, char* buf, int buflen) {
  APLA_REQUIRE(buf && buflen > 0 && B > 0 && N > 0 && H > 0, "apla_attn_kernel_name: bad arguments");
  variant &= 0xff;
  APLA_REQUIRE(variant >= 0 && variant <= 3, "apla_attn_kernel_name: unknown kernel variant %d", variant);
  const AttnKernel k = backward ? attn_bwd_choice(packed != 0, B, N, H, variant) : attn_fwd_choice(packed != 0, B, N, H, variant);
  snprintf(buf, (size_t)buflen, "%s", ATTN_KERNEL_NAMES[k]);
  return APLA_OK;
}

extern "C" int apla_attn_fwd_ex(const void* qkv, void* o, float* lse, int B, int N, int H, float scale, int variant,
                                hipStream_t stream) {
  APLA_REQUIRE(qkv && o && lse && B > 0 && N > 0 && H > 0, "apla_attn_fwd: bad arguments");
  APLA_REQUIRE(apla_aligned16(qkv) && apla_aligned16(o), "apla_attn_fwd: pointers must be 16-byte aligned");
  APLA_REQUIRE(B <= 65535 && H <= 65535, "apla_attn_fwd: B/H exceed grid limits");
  return launch_attn_fwd(qkv, o, lse, nullptr, B * N, B, N, H, scale, variant, stream, "apla_attn_fwd");
}
extern "C" int apla_attn_fwd(const void* qkv, void* o, float* lse, int B, int N, int H, float scale, hipStream_t stream) {
  return apla_attn_fwd_ex(qkv, o, lse, B, N, H, scale, 0, stream);
}

extern "C" int apla_attn_bwd_ex(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta,
                                void* dqkv, int B, int N, int H, float scale, int variant, hipStream_t stream) {
  APLA_REQUIRE(qkv && o && d_o && lse && delta && dqkv && B > 0 && N > 0 && H > 0, "apla_attn_bwd: bad arguments");
  APLA_REQUIRE(apla_aligned16(qkv) && apla_aligned16(o) && apla_aligned16(d_o) && apla_aligned16(dqkv), "apla_attn_bwd: pointers must be 16-byte aligned");
  APLA_REQUIRE(B <= 65535 && H <= 65535, "apla_attn_bwd: B/H exceed grid limits");
  return launch_attn_bwd(qkv, o, d_o, lse, delta, dqkv, nullptr, B * N, B, N, H, scale, variant, stream, "apla_attn_bwd");
}
extern "C" int apla_attn_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta,
                             void* dqkv, int B, int N, int H, float scale, hipStream_t stream) {
  return apla_attn_bwd_ex(qkv, o, d_o, lse, delta, dqkv, B, N, H, scale, 0, stream);
}

extern "C" int apla_attn_varlen_fwd_ex(const void* qkv, void* o, float* lse, const int32_t* cu_seqlens, int S, int total,
                                       int max_n, int H, float scale, int variant, hipStream_t stream) {
  APLA_REQUIRE(qkv && o && lse && cu_seqlens && S > 0 && total > 0 && max_n > 0 && H > 0, "apla_attn_varlen_fwd: bad arguments");
  APLA_REQUIRE(apla_aligned16(qkv) && apla_aligned16(o), "apla_attn_varlen_fwd: pointers must be 16-byte aligned");
  APLA_REQUIRE(S <= 65535 && H <= 65535, "apla_attn_varlen_fwd: S/H exceed grid limits");
  return launch_attn_fwd(qkv, o, lse, cu_seqlens, total, S, max_n, H, scale, variant, stream, "apla_attn_varlen_fwd");
}
extern "C" int apla_attn_varlen_fwd(const void* qkv, void* o, float* lse, const int32_t* cu_seqlens, int S, int total,
                                    int max_n, int H, float scale, hipStream_t stream) {
  return apla_attn_varlen_fwd_ex(qkv, o, lse, cu_seqlens, S, total, max_n, H, scale, 0, stream);
}

extern "C" int apla_attn_varlen_bwd_ex(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta,
                                       void* dqkv, const int32_t* cu_seqlens, int S, int total, int max_n, int H,
                                       float scale, int variant, hipStream_t stream) {
  APLA_REQUIRE(qkv && o && d_o && lse && delta && dqkv && cu_seqlens && S > 0 && total > 0 && max_n > 0 && H > 0, "apla_attn_varlen_bwd: bad arguments");
  APLA_REQUIRE(apla_aligned16(qkv) && apla_aligned16(o) && apla_aligned16(d_o) && apla_aligned16(dqkv), "apla_attn_varlen_bwd: pointers must be 16-byte aligned");
  APLA_REQUIRE(S <= 65535 && H <= 65535, "apla_attn_varlen_bwd: S/H exceed grid limits");
  return launch_attn_bwd(qkv, o, d_o, lse, delta, dqkv, cu_seqlens, total, S, max_n, H, scale, variant, stream, "apla_attn_varlen_bwd");
}
extern "C" int apla_attn_varlen_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta,
                                    void* dqkv, const int32_t* cu_seqlens, int S, int total, int max_n, int H,
                                    float scale, hipStream_t stream) {
  return apla_attn_varlen_bwd_ex(qkv, o, d_o, lse, delta, dqkv, cu_seqlens, S, total, max_n, H, scale, 0, stream);
}

static int drop_args(float p, unsigned long long seed, unsigned offset, DropArgs& da, const char* who) {
  APLA_REQUIRE(p > 0.f && p < 1.f, "%s: dropout probability must be in (0, 1), got %g", who, (double)p);
  const double t = (double)p * 4294967296.0;
  da.threshold = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
  da.inv_keep = 1.0f / (1.0f - p);
  da.seed = seed;
  da.offset = offset;
  return APLA_OK;
}
extern "C" int apla_attn_fwd_dropout(const void* qkv, void* o, float* lse, int B, int N, int H, float scale, float p,
                                     unsigned long long seed, unsigned offset, hipStream_t stream) {
  APLA_REQUIRE(qkv && o && lse && B > 0 && N > 0 && H > 0 && B <= 65535 && H <= 65535, "apla_attn_fwd_dropout: bad arguments");
  APLA_REQUIRE(apla_aligned16(qkv) && apla_aligned16(o), "apla_attn_fwd_dropout: pointers must be 16-byte aligned");
  DropArgs da;
  if (int rc = drop_args(p, seed, offset, da, "apla_attn_fwd_dropout")) return rc;
  hipLaunchKernelGGL(attn_fwd_kernel<true>, dim3((N + 127) / 128, H, B), dim3(256), 0, stream, (const bf16*)qkv, (bf16*)o, lse, N, H, scale,
                     (const int32_t*)nullptr, B * N, da);
  APLA_CHECK_LAUNCH("apla_attn_fwd_dropout");
  return APLA_OK;
}
extern "C" int apla_attn_bwd_dropout(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta, void* dqkv, int B,
                                     int N, int H, float scale, float p, unsigned long long seed, unsigned offset, hipStream_t stream) {
  APLA_REQUIRE(qkv && o && d_o && lse && delta && dqkv && B > 0 && N > 0 && H > 0 && B <= 65535 && H <= 65535, "apla_attn_bwd_dropout: bad arguments");
  APLA_REQUIRE(apla_aligned16(qkv) && apla_aligned16(o) && apla_aligned16(d_o) && apla_aligned16(dqkv), "apla_attn_bwd_dropout: pointers must be 16-byte aligned");
  DropArgs da;
  if (int rc = drop_args(p, seed, offset, da, "apla_attn_bwd_dropout")) return rc;
  const dim3 grid((N + 127) / 128, H, B);
  hipLaunchKernelGGL(attn_bwd_dq_kernel<true>, grid, dim3(256), 0, stream, (const bf16*)qkv, (const bf16*)o, (const bf16*)d_o, lse, delta, (bf16*)dqkv, N, H, scale,
                     (const int32_t*)nullptr, B * N, da);
  APLA_CHECK_LAUNCH("apla_attn_bwd_dropout");
  hipLaunchKernelGGL(attn_bwd_dkv_kernel<true>, grid, dim3(256), 0, stream, (const bf16*)qkv, (const bf16*)d_o, lse, delta, (bf16*)dqkv, N, H, scale,
                     (const int32_t*)nullptr, B * N, da);
  APLA_CHECK_LAUNCH("apla_attn_bwd_dropout");
  return APLA_OK;
}

extern "C" int apla_attn_probs_dropout(const void* qkv, const float* lse, float* attn, int B, int N, int H, float scale, float p,
                                       unsigned long long seed, unsigned offset, hipStream_t stream) {
  APLA_REQUIRE(qkv && lse && attn && B > 0 && N > 0 && H > 0 && B <= 65535 && H <= 65535, "apla_attn_probs_dropout: bad arguments");
  DropArgs da;
  if (int rc = drop_args(p, seed, offset, da, "apla_attn_probs_dropout")) return rc;
  hipLaunchKernelGGL(attn_probs_kernel<true>, dim3(N, H, B), dim3(256), 0, stream, (const bf16*)qkv, lse, attn, N, H, scale, da);
  APLA_CHECK_LAUNCH("apla_attn_probs_dropout");
  return APLA_OK;
}

extern "C" int apla_attn_fwd_cls(const void* qkv, void* o, float* lse, int B, int N, int H, float scale,
                                 hipStream_t stream) {
  APLA_REQUIRE(qkv && o && lse && B > 0 && N > 0 && H > 0 && B <= 65535, "apla_attn_fwd_cls: bad arguments");
  hipLaunchKernelGGL(attn_fwd_cls_kernel, dim3(H, B), dim3(256), 0, stream, (const bf16*)qkv, (bf16*)o, lse, N, H, scale);
  APLA_CHECK_LAUNCH("apla_attn_fwd_cls");
  return APLA_OK;
}

extern "C" int apla_attn_probs(const void* qkv, const float* lse, float* attn, int B, int N, int H, float scale,
                               hipStream_t stream) {
  APLA_REQUIRE(qkv && lse && attn && B > 0 && N > 0 && H > 0 && B <= 65535 && H <= 65535, "apla_attn_probs: bad arguments");
  hipLaunchKernelGGL(attn_probs_kernel<false>, dim3(N, H, B), dim3(256), 0, stream, (const bf16*)qkv, lse, attn, N, H, scale, DropArgs{});
  APLA_CHECK_LAUNCH("apla_attn_probs");
  return APLA_OK;
}

extern "C" int apla_attn_bwd_cls(const void* qkv, const void* o, const void* do_cls, const float* lse, void* dqkv, int B,
                                 int N, int H, float scale, hipStream_t stream) {
  APLA_REQUIRE(qkv && o && do_cls && lse && dqkv && B > 0 && N > 0 && H > 0 && B <= 65535, "apla_attn_bwd_cls: bad arguments");
  hipLaunchKernelGGL(attn_bwd_cls_kernel, dim3(H, B), dim3(256), 0, stream, (const bf16*)qkv, (const bf16*)o, (const bf16*)do_cls, lse, (bf16*)dqkv, N, H, scale);
  APLA_CHECK_LAUNCH("apla_attn_bwd_cls");
  return APLA_OK;
}
