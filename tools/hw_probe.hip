// Hardware-semantics probe for gfx950: verifies the MFMA fragment maps, the LDS-DMA destination rule
// and the transposed LDS read that the kernels in apla_amd/csrc rely on.  Build: hipcc --offload-arch=gfx950
// tools/hw_probe.hip -o tools/hw_probe ; run on the GPU box.  Prints PASS/FAIL per probe.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))
#define GLBP(p) ((const __attribute__((address_space(1))) void*)(p))

__global__ void p1_mfma16(const __bf16* A, const __bf16* B, float* out) {  // A[16][32], B[32][16]
  int l = threadIdx.x; bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = A[(l & 15) * 32 + 8 * (l >> 4) + j]; b[j] = B[(8 * (l >> 4) + j) * 16 + (l & 15)]; }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
__global__ void p2_mfma32(const __bf16* A, const __bf16* B, float* out) {  // A[32][16], B[16][32]
  int l = threadIdx.x; bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = A[(l & 31) * 16 + 8 * (l >> 5) + j]; b[j] = B[(8 * (l >> 5) + j) * 32 + (l & 31)]; }
  f32x16 c = {};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) out[l * 16 + r] = c[r];
}
// X = A1[32][16] * B1[16][32]; Y = A2[32][32] * X  with X taken from the accumulator as B operand;
// Z = X^T * B3[32][32] with X taken as A operand.
__global__ void p3_acc_as_operand(const __bf16* A1, const __bf16* B1, const __bf16* A2, const __bf16* B3, float* outY, float* outZ) {
  int l = threadIdx.x, h = l >> 5; bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = A1[(l & 31) * 16 + 8 * h + j]; b[j] = B1[(8 * h + j) * 32 + (l & 31)]; }
  f32x16 x = {};
  x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, x, 0, 0, 0);
  f32x16 y = {}, z = {};
  for (int s = 0; s < 2; ++s) {
    bf16x8 xb, a2, b3;
    for (int j = 0; j < 8; ++j) {
      xb[j] = (__bf16)x[8 * s + j];
      int k = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
      a2[j] = A2[(l & 31) * 32 + k];
      b3[j] = B3[k * 32 + (l & 31)];
    }
    y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, xb, y, 0, 0, 0);
    z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, b3, z, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) { outY[l * 16 + r] = y[r]; outZ[l * 16 + r] = z[r]; }
}
// LDS-DMA: 4 waves, wave w writes piece (w) at smem + 2048 + w*1024; lane i source = src + perm(i)*8 elements
__global__ void p4_glds(const __bf16* src, __bf16* dump) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int t = threadIdx.x, w = t >> 6, i = t & 63;
  int srcchunk = w * 64 + (i ^ 5);
  __builtin_amdgcn_global_load_lds(GLBP(src + srcchunk * 8), LDSP(smem + 2048 + w * 1024), 16, 0, 0);
  __syncthreads();
  const __bf16* s = (const __bf16*)(smem + 2048);
  for (int j = 0; j < 8; ++j) dump[t * 8 + j] = s[t * 8 + j];
}
// transposed read: LDS holds M[64][64] bf16 row-major (128 B rows).  For block (r0,c0): lane 4q+p of 16-group supplies
// &M[r0+q][c0+4p]; expects lane i gets M[r0+0..3][c0+i].
__global__ void p5_tr(const __bf16* M, __bf16* out) {
  __shared__ __attribute__((aligned(16))) __bf16 sm[64 * 64];
  int t = threadIdx.x;
  for (int e = t; e < 64 * 64; e += 64) sm[e] = M[e];
  __syncthreads();
  int g = t >> 4, i = t & 15, q = i >> 2, p = i & 3;
  int r0 = 4 * g + 8, c0 = 16 * (g & 1) + 16;
  bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(&sm[(r0 + q) * 64 + c0 + 4 * p]));
  for (int e = 0; e < 4; ++e) out[t * 4 + e] = v[e];
}
static float bf(float x) { return x; }
int main() {
  int fails = 0;
  auto rnd = [](int m) { return (float)((rand() % (2 * m + 1)) - m); };
  {  // P1
    std::vector<__bf16> A(16 * 32), B(32 * 16); std::vector<float> Af(16 * 32), Bf(32 * 16);
    for (int i = 0; i < 16 * 32; ++i) { Af[i] = rnd(4); A[i] = (__bf16)Af[i]; Bf[i] = rnd(4); B[i] = (__bf16)Bf[i]; }
    __bf16 *dA, *dB; float* dO; hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dO, 64 * 4 * 4);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    p1_mfma16<<<1, 64>>>(dA, dB, dO); std::vector<float> o(256); hipMemcpy(o.data(), dO, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) { int row = 4 * (l >> 4) + r, col = l & 15; float ref = 0; for (int k = 0; k < 32; ++k) ref += Af[row * 32 + k] * Bf[k * 16 + col]; if (ref != o[l * 4 + r]) ++bad; }
    printf("P1 mfma16x16x32 layout: %s (%d bad)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
  }
  {  // P2
    std::vector<__bf16> A(32 * 16), B(16 * 32); std::vector<float> Af(512), Bf(512);
    for (int i = 0; i < 512; ++i) { Af[i] = rnd(4); A[i] = (__bf16)Af[i]; Bf[i] = rnd(4); B[i] = (__bf16)Bf[i]; }
    __bf16 *dA, *dB; float* dO; hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dO, 64 * 16 * 4);
    hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
    p2_mfma32<<<1, 64>>>(dA, dB, dO); std::vector<float> o(1024); hipMemcpy(o.data(), dO, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) { int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31; float ref = 0; for (int k = 0; k < 16; ++k) ref += Af[row * 16 + k] * Bf[k * 32 + col]; if (ref != o[l * 16 + r]) ++bad; }
    printf("P2 mfma32x32x16 layout: %s (%d bad)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
  }
  {  // P3
    std::vector<__bf16> A1(512), B1(512), A2(1024), B3(1024); std::vector<float> A1f(512), B1f(512), A2f(1024), B3f(1024);
    for (int i = 0; i < 512; ++i) { A1f[i] = rnd(2); A1[i] = (__bf16)A1f[i]; B1f[i] = rnd(2); B1[i] = (__bf16)B1f[i]; }
    for (int i = 0; i < 1024; ++i) { A2f[i] = rnd(2); A2[i] = (__bf16)A2f[i]; B3f[i] = rnd(2); B3[i] = (__bf16)B3f[i]; }
    __bf16 *d1, *d2, *d3, *d4; float *dY, *dZ; hipMalloc(&d1, 1024); hipMalloc(&d2, 1024); hipMalloc(&d3, 2048); hipMalloc(&d4, 2048); hipMalloc(&dY, 4096); hipMalloc(&dZ, 4096);
    hipMemcpy(d1, A1.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(d2, B1.data(), 1024, hipMemcpyHostToDevice);
    hipMemcpy(d3, A2.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(d4, B3.data(), 2048, hipMemcpyHostToDevice);
    p3_acc_as_operand<<<1, 64>>>(d1, d2, d3, d4, dY, dZ); std::vector<float> y(1024), z(1024);
    hipMemcpy(y.data(), dY, 4096, hipMemcpyDeviceToHost); hipMemcpy(z.data(), dZ, 4096, hipMemcpyDeviceToHost);
    std::vector<float> X(1024);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 16; ++k) s += A1f[i * 16 + k] * B1f[k * 32 + j]; X[i * 32 + j] = s; }
    int badY = 0, badZ = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) { int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31; float ry = 0, rz = 0;
      for (int k = 0; k < 32; ++k) { ry += A2f[row * 32 + k] * X[k * 32 + col]; rz += X[k * 32 + row] * B3f[k * 32 + col]; }
      if (ry != y[l * 16 + r]) ++badY; if (rz != z[l * 16 + r]) ++badZ; }
    printf("P3 acc-as-B (A*X): %s (%d bad); acc-as-A (X^T*B): %s (%d bad)\n", badY ? "FAIL" : "PASS", badY, badZ ? "FAIL" : "PASS", badZ); fails += (badY != 0) + (badZ != 0);
  }
  {  // P4
    std::vector<__bf16> S(256 * 8); for (int i = 0; i < 2048; ++i) S[i] = (__bf16)(float)(i % 251);
    __bf16 *dS, *dD; hipMalloc(&dS, 4096); hipMalloc(&dD, 4096); hipMemcpy(dS, S.data(), 4096, hipMemcpyHostToDevice);
    p4_glds<<<1, 256, 8192>>>(dS, dD); std::vector<__bf16> D(2048); hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; ++t) { int w = t >> 6, i = t & 63; int sc = w * 64 + (i ^ 5); for (int j = 0; j < 8; ++j) if ((float)D[t * 8 + j] != (float)S[sc * 8 + j]) ++bad; }
    printf("P4 global_load_lds dest = base + lane*16: %s (%d bad)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
  }
  {  // P5
    std::vector<__bf16> M(4096); for (int i = 0; i < 4096; ++i) M[i] = (__bf16)(float)((i / 64) * 64 + (i % 64) % 64 == 0 ? 0 : ((i * 7) % 255));
    for (int i = 0; i < 4096; ++i) M[i] = (__bf16)(float)((i * 7 + i / 64) % 255);
    __bf16 *dM, *dO; hipMalloc(&dM, 8192); hipMalloc(&dO, 64 * 4 * 2); hipMemcpy(dM, M.data(), 8192, hipMemcpyHostToDevice);
    p5_tr<<<1, 64>>>(dM, dO); std::vector<__bf16> o(256); hipMemcpy(o.data(), dO, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 64; ++t) { int g = t >> 4, i = t & 15; int r0 = 4 * g + 8, c0 = 16 * (g & 1) + 16; for (int e = 0; e < 4; ++e) if ((float)o[t * 4 + e] != (float)M[(r0 + e) * 64 + c0 + i]) ++bad; }
    printf("P5 ds_read_b64_tr_b16: %s (%d bad)\n", bad ? "FAIL" : "PASS", bad); fails += bad != 0;
  }
  hipDeviceSynchronize();
  hipError_t e = hipGetLastError(); printf("last hip error: %s\n", hipGetErrorString(e));
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0); printf("device: %s CUs=%d clock=%d MHz mem=%.1f GB\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000, prop.totalGlobalMem / 1e9);
  (void)bf;
  return fails;
}
