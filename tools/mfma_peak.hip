// MFMA peak probe for MI355X (gfx950): register-only matrix-core loops on random operands over every CU (SURVEY §8d asks
// for a measured peak beside the 2.5 PFLOP/s datasheet figure that `roofline.peak` quotes).
//
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak && tools/mfma_peak [seconds_per_variant]
//
// Each workgroup is four waves (one per SIMD; "x2" variants launch two workgroups per CU = two waves per SIMD).  A wave keeps
// 4 A and 4 B fragments in registers and issues the 16 independent products of them back to back; nothing touches LDS or
// memory inside the loop (the MFMAs are inline asm on pinned registers: with the builtins hipcc moved the 16x16x32 accumulators
// through AGPR copies and s_nops and that loop reached 57 % of its issue rate), so the figure is the matrix pipes at the clock the chip holds under that load (it lowers its
// clock under a dense MFMA stream: MI355X_MICROARCH.md "DVFS give-back").  Prints ONE JSON line; bench.py embeds it as
// roofline.peak_measured.  `clock_ghz` is the in-kernel clock (s_memtime ticks / s_memrealtime at 100 MHz) of the median
// workgroup.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// SHAPE 0: v_mfma_f32_16x16x32 (16 accumulators of 4 registers), SHAPE 1: v_mfma_f32_32x32x16 (4 x 4 accumulators of 16
// registers would need 256 registers: 2 x 4 are kept instead, 8 independent chains); F16 picks the IEEE-half opcode.
template <int SHAPE, bool F16>
__global__ __launch_bounds__(256) void mfma_loop(const uint4* __restrict__ src, float* __restrict__ sink, int iters,
                                                 unsigned long long* __restrict__ stamps) {
  const int lane_id = blockIdx.x * 256 + threadIdx.x;
  u32x4 ra[4], rb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ra[i] = __builtin_bit_cast(u32x4, src[(lane_id * 8 + i) & 0xFFFFF]);
    rb[i] = __builtin_bit_cast(u32x4, src[(lane_id * 8 + 4 + i) & 0xFFFFF]);
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float total = 0.f;
  if constexpr (SHAPE == 0) {
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (F16)
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(ra[i]), "v"(rb[j]));
          else
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(ra[i]), "v"(rb[j]));
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) total += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  } else {
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (F16)
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(ra[i]), "v"(rb[j]));
          else
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(ra[i]), "v"(rb[j]));
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) total += acc[i][j][e];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  sink[lane_id] = total;  // keeps the products alive; never read back
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

struct Result { double tflops, clock_ghz; };

template <int SHAPE, bool F16>
static Result run(int wgs_per_cu, double seconds, const uint4* src, float* sink, unsigned long long* stamps, int cus) {
  const int grid = cus * wgs_per_cu;
  const int iters = 20000;
  // FLOP per wave per iteration: SHAPE 0: 16 x (2*16*16*32); SHAPE 1: 8 x (2*32*32*16)
  const double flop_iter = SHAPE == 0 ? 16.0 * 2 * 16 * 16 * 32 : 8.0 * 2 * 32 * 32 * 16;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  auto launch = [&]() { hipLaunchKernelGGL((mfma_loop<SHAPE, F16>), dim3(grid), dim3(256), 0, 0, src, sink, iters, stamps); };
  launch();
  CHECK(hipDeviceSynchronize());
  // warm the clock governor: back-to-back launches for `seconds`, then time the last quarter
  CHECK(hipEventRecord(e0));
  launch();
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms1 = 0.f;
  CHECK(hipEventElapsedTime(&ms1, e0, e1));
  int n = (int)(seconds * 1e3 / (ms1 > 0.01f ? ms1 : 0.01f));
  n = n < 8 ? 8 : n;
  for (int i = 0; i < n - n / 4; ++i) launch();
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < n / 4; ++i) launch();
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double tf = flop_iter * iters * 4.0 * grid * (n / 4) / (ms * 1e-3) / 1e12;
  std::vector<unsigned long long> h(2 * grid);
  CHECK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> clk(grid);
  for (int b = 0; b < grid; ++b) clk[b] = h[2 * b + 1] ? (double)h[2 * b] / (double)h[2 * b + 1] * 0.1 : 0.0;  // ticks per 10 ns -> GHz
  std::sort(clk.begin(), clk.end());
  return {tf, clk[grid / 2]};
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 0.6;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const size_t nsrc = 1 << 20;
  std::vector<unsigned> h(nsrc * 4);
  unsigned s = 12345u;
  for (auto& v : h) {  // two random 16-bit floats in about [-1, 1): sign, exponent 0x3c..0x3f region, random mantissa
    unsigned w = 0;
    for (int k = 0; k < 2; ++k) {
      s = s * 1664525u + 1013904223u;
      const unsigned r = s >> 8;
      // bf16 0x3f00..0x3f7f = [0.5, 1): keep exponent near there, random sign and mantissa; as fp16 bits the same pattern is
      // a normal number of magnitude ~1.75..2 — both are full-entropy mantissas, which is what sets the power draw
      const unsigned half = ((r & 1u) << 15) | 0x3f00u | ((r >> 1) & 0x7fu) | (((r >> 9) & 1u) ? 0u : 0x0080u);
      w |= half << (16 * k);
    }
    v = w;
  }
  uint4* src;
  float* sink;
  unsigned long long* stamps;
  CHECK(hipMalloc(&src, nsrc * 16));
  CHECK(hipMalloc(&sink, (size_t)cus * 2 * 256 * 4));
  CHECK(hipMalloc(&stamps, (size_t)cus * 2 * 16));
  CHECK(hipMemcpy(src, h.data(), nsrc * 16, hipMemcpyHostToDevice));
  const Result a1 = run<0, false>(1, seconds, src, sink, stamps, cus);
  const Result a2 = run<0, false>(2, seconds, src, sink, stamps, cus);
  const Result b1 = run<1, false>(1, seconds, src, sink, stamps, cus);
  const Result b2 = run<1, false>(2, seconds, src, sink, stamps, cus);
  const Result c1 = run<0, true>(1, seconds, src, sink, stamps, cus);
  printf("{\"device\": \"%s\", \"cus\": %d, \"unit\": \"TFLOP/s\", \"data\": \"random sign+mantissa\", "
         "\"bf16_16x16x32_1wave_per_simd\": %.1f, \"bf16_16x16x32_2waves_per_simd\": %.1f, "
         "\"bf16_32x32x16_1wave_per_simd\": %.1f, \"bf16_32x32x16_2waves_per_simd\": %.1f, \"f16_16x16x32_1wave_per_simd\": %.1f, "
         "\"clock_ghz\": {\"bf16_16x16x32\": %.3f, \"bf16_16x16x32_x2\": %.3f, \"bf16_32x32x16\": %.3f, \"bf16_32x32x16_x2\": %.3f, \"f16_16x16x32\": %.3f}}\n",
         prop.gcnArchName, cus, a1.tflops, a2.tflops, b1.tflops, b2.tflops, c1.tflops, a1.clock_ghz, a2.clock_ghz, b1.clock_ghz,
         b2.clock_ghz, c1.clock_ghz);
  return 0;
}
