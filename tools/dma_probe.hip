// LDS-DMA fill-rate probe for MI355X (gfx950): how fast does a CU pull row segments into LDS with global_load_lds_dwordx4, as a
// function of the SEGMENT a row contributes to one instruction (64 B = a 32-wide bf16 K-step, 128 B = a 64-wide one = one cache
// line)?  The GEMM kernels' K-step width decides this pattern; tools/ring_stamps.py showed the issue of these instructions to be
// the largest share of a K-step.
//
//   hipcc --offload-arch=gfx950 -O3 tools/dma_probe.hip -o tools/dma_probe && tools/dma_probe
//
// Every workgroup (256 threads) streams the same 2048 rows x 1536 B (3 MB: stays in each XCD's L2) K-step by K-step into a
// 32 KB LDS ring; a wave waits with vmcnt(0) after every 8 instructions.  Prints GB/s per CU and cycles per instruction.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))
#define GLBP(p) ((const __attribute__((address_space(1))) void*)(p))

constexpr int ROW_BYTES = 1536, ROWS = 2048;

// SEG: bytes of a row per instruction (64 | 128 | 256).  One instruction = 1 KB = (1024 / SEG) rows.
template <int SEG>
__global__ __launch_bounds__(256) void dma_kernel(const char* __restrict__ src, int iters, unsigned long long* __restrict__ stamps,
                                                   float* __restrict__ sink) {
  __shared__ __attribute__((aligned(16))) char smem[32768];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int LPR = SEG / 16;            // lanes per row
  constexpr int RPI = 64 / LPR;            // rows per instruction
  const int r_in = lane / LPR, c_in = lane % LPR;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int row = (blockIdx.x * 37 + wave * RPI * 8) % ROWS;
  int seg = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int r = (row + k * RPI + r_in) % ROWS;
      __builtin_amdgcn_global_load_lds(GLBP(src + (size_t)r * ROW_BYTES + seg * SEG + c_in * 16), LDSP(smem + wave * 8192 + k * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    seg += 1;
    if (seg * SEG >= ROW_BYTES) { seg = 0; row = (row + 8 * RPI * 4) % ROWS; }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  if (threadIdx.x == 0) { stamps[blockIdx.x] = t1 - t0; sink[blockIdx.x] = *(float*)(smem + (t1 & 1023) * 4); }
}

template <int SEG>
static void run(const char* src, int wgs_per_cu, int cus, unsigned long long* stamps, float* sink) {
  const int iters = 4000, grid = cus * wgs_per_cu;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(dma_kernel<SEG>, dim3(grid), dim3(256), 0, 0, src, iters, stamps, sink);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(dma_kernel<SEG>, dim3(grid), dim3(256), 0, 0, src, iters, stamps, sink);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long h = 0;
  CHECK(hipMemcpy(&h, stamps, 8, hipMemcpyDeviceToHost));
  const double bytes = (double)grid * 4 * iters * 8 * 1024;
  printf("segment %3d B, %d workgroup(s)/CU: %6.1f GB/s per CU, %5.2f TB/s chip, %6.1f ticks per instruction and wave (workgroup 0)\n", SEG,
         wgs_per_cu, bytes / (ms * 1e-3) / cus / 1e9, bytes / (ms * 1e-3) / 1e12, (double)h / (iters * 8.0));
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  char* src;
  unsigned long long* stamps;
  float* sink;
  CHECK(hipMalloc(&src, (size_t)ROWS * ROW_BYTES + 4096));
  CHECK(hipMemset(src, 1, (size_t)ROWS * ROW_BYTES + 4096));
  CHECK(hipMalloc(&stamps, 8 * 2048));
  CHECK(hipMalloc(&sink, 4 * 2048));
  for (int w = 1; w <= 2; ++w) {
    run<64>(src, w, cus, stamps, sink);
    run<128>(src, w, cus, stamps, sink);
    run<256>(src, w, cus, stamps, sink);
  }
  return 0;
}
