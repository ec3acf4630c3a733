// Shared device/host helpers for the APLA gfx950 kernels (CDNA4 only; no other target is supported).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/apla_hip.h"

// `bf16` is this build's 16-bit operand type: __bf16 by default, IEEE half with -DAPLA_FP16 (libapla_hip_f16.so).  The
// kernels are written once against the name; the three type-specific builtins are behind the macros below.
#if defined(APLA_FP16)
typedef _Float16 bf16;
#define APLA_H16 APLA_F16
#define MFMA_F32_16x16x32_H16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define MFMA_F32_32x32x16_H16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
typedef __fp16 apla_v4fp16 __attribute__((__vector_size__(4 * sizeof(__fp16))));
#define DS_READ_TR16_B64_H16(p) \
  __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) apla_v4fp16*)(p)))
#else
typedef __bf16 bf16;
#define APLA_H16 APLA_BF16
#define MFMA_F32_16x16x32_H16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define MFMA_F32_32x32x16_H16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define DS_READ_TR16_B64_H16(p) __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p))
#endif
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))
#define GLBP(p) ((const __attribute__((address_space(1))) void*)(p))

// ---- host-side error plumbing (thread-local message, C-ABI returns negative errno-style codes) ----
void apla_set_error(const char* fmt, ...);

#define APLA_REQUIRE(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      apla_set_error(__VA_ARGS__);         \
      return APLA_EINVAL;                  \
    }                                      \
  } while (0)

#define APLA_CHECK_LAUNCH(name)                                                  \
  do {                                                                           \
    hipError_t e__ = hipGetLastError();                                          \
    if (e__ != hipSuccess) {                                                     \
      apla_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
      return APLA_EIO;                                                           \
    }                                                                            \
  } while (0)

static inline bool apla_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// ---- device helpers ----
__device__ __forceinline__ float bf2f(bf16 x) { return (float)x; }
__device__ __forceinline__ bf16 f2bf(float x) { return (bf16)x; }

__device__ __forceinline__ bf16x4 pack4(float a, float b, float c, float d) {
  bf16x4 r;
  r[0] = (bf16)a; r[1] = (bf16)b; r[2] = (bf16)c; r[3] = (bf16)d;
  return r;
}

// residual-stream element access: ResT is float or bf16; 4 consecutive elements
template <typename T> struct Vec4IO;
template <> struct Vec4IO<float> {
  static __device__ __forceinline__ f32x4 load(const float* p) { return *(const f32x4*)p; }
  static __device__ __forceinline__ void store(float* p, f32x4 v) { *(f32x4*)p = v; }
};
template <> struct Vec4IO<bf16> {
  static __device__ __forceinline__ f32x4 load(const bf16* p) {
    bf16x4 t = *(const bf16x4*)p;
    f32x4 r = {(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
    return r;
  }
  static __device__ __forceinline__ void store(bf16* p, f32x4 v) { *(bf16x4*)p = pack4(v[0], v[1], v[2], v[3]); }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// bijective XCD-aware block remap: blocks b and b+8 share an XCD (round-robin dispatch), so give each
// XCD a contiguous run of logical tile ids -> neighbouring tiles (same A panel) hit the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Philox4x32-10 (Salmon et al., SC'11): the counter-based generator behind every dropout mask of the library (misc.hip: nn.Dropout /
// DropPath; attention.hip: attention-probability dropout); restated in oracle/apla_oracle.py and pinned there by the Random123
// known-answer vectors.
__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0], p1 = (unsigned long long)0xCD9E8D57u * c[2];
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (unsigned)p1; c[3] = (unsigned)p0; c[0] = n0; c[2] = n2;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

// Dropout inside other kernels (round 6): the keep decision of element i of a tensor is word (i & 3) of Philox4x32-10(counter {i >> 2,
// offset}, key seed) >= threshold — the convention of apla_dropout_fwd (misc.hip), so a mask drawn inside a LayerNorm or GEMM epilogue
// is the one the stand-alone pass and the oracle (oracle/apla_oracle.py:philox_keep_mask) draw.  `rng` points at {seed, step} in
// device memory (the step counter advances outside the captured launches); offset = step * stride + site.
struct DropArgsEw {
  const unsigned long long* rng;   // NULL: no dropout
  unsigned long long stride;       // offsets per step (the engine's 4 L + 8)
  unsigned site;                   // which dropout of the step
  unsigned threshold;              // keep iff word >= threshold  (= p * 2^32)
  float inv_keep;                  // 1 / (1 - p)
};
__device__ __forceinline__ void drop_words4(const DropArgsEw& d, unsigned long long blk, unsigned (&c)[4]) {
  const unsigned long long seed = d.rng[0], off = d.rng[1] * d.stride + d.site;
  c[0] = (unsigned)blk; c[1] = (unsigned)(blk >> 32); c[2] = (unsigned)off; c[3] = (unsigned)(off >> 32);
  philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
}
static inline unsigned apla_drop_threshold(float p) {
  const double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 4294967295u : (unsigned)t;
}
