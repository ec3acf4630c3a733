// Store-pattern probe for gfx950: how fast can 8 waves per CU write a 320x256 bf16 output tile with 16-B stores, as a
// function of how a wave instruction's 64 lanes are laid over rows?  (GEMM epilogue design input; see DESIGN.md.)
// Build: hipcc --offload-arch=gfx950 -O3 tools/store_probe.hip -o tools/store_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// wave tile 80 rows x 128 columns (256 B per row); 20 store instructions per wave
template <int PAT>
__global__ __launch_bounds__(512) void store_kernel(char* out, int M, int N, int tiles_m, int tiles_n) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
  const size_t ld = (size_t)N * 2;
  f32x4 v = {(float)lane, 1.f, 2.f, 3.f};
  for (int t = blockIdx.x; t < tiles_m * tiles_n; t += gridDim.x) {
    const int tm = t / tiles_n, tn = t % tiles_n;
    const int r0 = tm * 320 + grp * 160 + wm * 80;
    char* base = out + (size_t)tn * 512 + wn * 256;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int row, chunk;
        if (PAT == 0) { row = i * 16 + (lane & 15); chunk = 4 * u + (lane >> 4); }          // 16 rows x 64 B  (MFMA layout)
        else if (PAT == 1) { row = i * 16 + 4 * u + (lane >> 4); chunk = lane & 15; }        // 4 rows x 256 B
        else if (PAT == 2) { row = i * 16 + 8 * (u >> 1) + (lane >> 3); chunk = (lane & 7) + 8 * (u & 1); }  // 8 rows x 128 B
        else { row = i * 16 + (lane & 15); chunk = 4 * (lane >> 4) + u; }                    // 16 rows x 4 scattered 16-B pieces
        const int r = r0 + row;
        if (r < M) *(f32x4*)(base + (size_t)r * ld + chunk * 16) = v;
      }
    }
  }
}

int main(int argc, char** argv) {
  const int M = 25216;
  const int G = argc > 1 ? atoi(argv[1]) : 256;  // workgroups (= CUs) storing; the tile list is cut to G tiles x 3 rounds
  for (int N : {2304, 3072}) {
    const int tiles_m = (M + 319) / 320, tiles_n = N / 256;
    const int tm_used = G >= 256 ? tiles_m : (3 * G + tiles_n - 1) / tiles_n;
    const double bytes = (double)(tm_used < tiles_m ? tm_used * 320 : M) * N * 2;
    char* out;
    hipMalloc(&out, (size_t)M * N * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pat = 0; pat < 4; ++pat) {
      float best = 1e9f;
      for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        for (int it = 0; it < 10; ++it) {
          switch (pat) {
            case 0: hipLaunchKernelGGL(store_kernel<0>, dim3(G), dim3(512), 0, 0, out, M, N, tm_used, tiles_n); break;
            case 1: hipLaunchKernelGGL(store_kernel<1>, dim3(G), dim3(512), 0, 0, out, M, N, tm_used, tiles_n); break;
            case 2: hipLaunchKernelGGL(store_kernel<2>, dim3(G), dim3(512), 0, 0, out, M, N, tm_used, tiles_n); break;
            default: hipLaunchKernelGGL(store_kernel<3>, dim3(G), dim3(512), 0, 0, out, M, N, tm_used, tiles_n); break;
          }
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms / 10 < best) best = ms / 10;
      }
      printf("G=%3d N=%4d pattern %d: %7.1f us  %6.2f TB/s  %6.1f GB/s per CU\n", G, N, pat, best * 1e3, bytes / (best * 1e-3) / 1e12, bytes / (best * 1e-3) / 1e9 / G);
    }
    hipFree(out);
  }
  return 0;
}
