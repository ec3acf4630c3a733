// Where does the dispatcher put the workgroups of a persistent launch?  (gfx950; performance study only, nothing may DEPEND on it.)
// Launches G workgroups of 256 threads with LDS_KB of LDS (so that two fit a CU, like gemm_persist_kernel), keeps each resident for
// ~50 us so that the whole grid is co-resident, and records HW_REG_XCC_ID / HW_REG_HW_ID per workgroup.  Prints, per XCD label
// (blockIdx % 8), which slots (blockIdx / 8) share a CU.  Build: hipcc --offload-arch=gfx950 -O2 tools/placement_probe.hip -o
// tools/placement_probe ; run on the GPU box: tools/placement_probe [G=512] .
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#ifndef LDS_KB
#define LDS_KB 74
#endif

__global__ __launch_bounds__(256, 2) void where_kernel(unsigned* out, long long spin_ticks) {
  __shared__ char smem[LDS_KB * 1024];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  smem[threadIdx.x] = (char)hw;  // keep the allocation
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc + (unsigned)smem[1] * 0u;
  }
}

int main(int argc, char** argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 512;
  unsigned* d;
  hipMalloc(&d, G * 8);
  std::vector<unsigned> h(2 * G);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(where_kernel, dim3(G), dim3(256), 0, 0, d, 5000LL);  // 100 MHz wall clock: 50 us
    hipEventRecord(e1, 0);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%d workgroups x %d KB of LDS, 50 us each: launch took %.1f us (one round if all are co-resident)\n", G, LDS_KB, ms * 1e3f);
  }
  hipMemcpy(h.data(), d, G * 8, hipMemcpyDeviceToHost);
  int label_matches = 0, pair_adjacent = 0, pair_half = 0, pair_other = 0, singles = 0;
  for (int x = 0; x < 8; ++x) {
    std::map<unsigned, std::vector<int>> cu;  // (xcc, se, sh, cu) -> slots
    unsigned xcc0 = h[2 * x + 1] & 15;
    const int slots = G / 8 + (G % 8 > x ? 1 : 0);
    for (int s = 0; s < slots; ++s) {
      const unsigned hw = h[2 * (8 * s + x)], xcc = h[2 * (8 * s + x) + 1] & 15;
      label_matches += xcc == xcc0;
      cu[(xcc << 16) | (hw & 0xff00)].push_back(s);  // CU_ID [11:8], SH_ID [12], SE_ID [15:13]
    }
    printf("label %d (XCC_ID of slot 0: %u): %zu distinct CUs for %d slots;", x, xcc0, cu.size(), slots);
    int shown = 0;
    for (auto& kv : cu) {
      auto& v = kv.second;
      if (v.size() == 1) { ++singles; continue; }
      for (size_t i = 0; i + 1 < v.size(); ++i) {
        const int dlt = v[i + 1] - v[i];
        if (dlt == 1) ++pair_adjacent; else if (dlt == slots / 2) ++pair_half; else ++pair_other;
      }
      if (shown++ < 6) { printf(" ["); for (int s : v) printf("%d ", s); printf("]"); }
    }
    printf("\n");
  }
  printf("workgroups whose XCC_ID equals their label's: %d of %d\n", label_matches, G);
  printf("co-resident slot pairs: adjacent (s, s+1): %d, half a round apart (s, s+slots/2): %d, other: %d; CUs with one workgroup: %d\n",
         pair_adjacent, pair_half, pair_other, singles);
  hipFree(d);
  return 0;
}
