// Fused global-norm clip + AdamW over the flat APLA-trainable buffer (2L+2 tensors laid out back to back).
// Two launches, no atomics, no host synchronisation: (1) 256 workgroups write partial sums of squares of
// grad*grad_scale; (2) every workgroup re-reduces those 256 partials in a fixed order (deterministic), derives the
// clip coefficient on device and applies the AdamW update in the exact operation order of torch.optim.AdamW.
#include "common.h"

namespace {
constexpr int NPART = 256;

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, float grad_scale,
                                                    float* __restrict__ ws) {
  __shared__ float red[4];
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)NPART * 256) {
    const float v = g[i] * grad_scale;
    s += v * v;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) ws[2 + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, const uint8_t* __restrict__ decay, long n,
                                                    float lr, float wd, float b1, float b2, float eps, float bc1,
                                                    float bc2_sqrt, float max_norm, float grad_scale,
                                                    float* __restrict__ ws) {
  __shared__ float coef_s;
  if (threadIdx.x < 64) {
    float s = 0.f;
    for (int i = threadIdx.x; i < NPART; i += 64) s += ws[2 + i];
    s = wave_sum(s);
    if (threadIdx.x == 0) {
      const float norm = sqrtf(s);
      float coef = 1.0f;
      if (max_norm > 0.f) { coef = max_norm / (norm + 1e-6f); coef = coef < 1.0f ? coef : 1.0f; }
      // a non-finite norm (fp16 gradients that overflowed under the loss scale) skips the update, like GradScaler.step
      coef_s = isfinite(norm) ? coef * grad_scale : __builtin_nanf("");
      if (blockIdx.x == 0) { ws[0] = s; ws[1] = norm; }
    }
  }
  __syncthreads();
  const float coef = coef_s;
  if (coef != coef) return;
  const float step_size = lr / bc1;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float gi = g[i] * coef;
    float pi = p[i];
    if (decay[i]) pi *= (1.0f - lr * wd);
    const float mi = m[i] * b1 + gi * (1.0f - b1);
    const float vi = v[i] * b2 + gi * gi * (1.0f - b2);
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= step_size * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi; g[i] = gi;
  }
}
}  // namespace

extern "C" int apla_adamw_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                               const uint8_t* decay_mask, long n, float lr, float weight_decay, float beta1,
                               float beta2, float eps, int step, float max_norm, float grad_scale, float* norm_ws,
                               hipStream_t stream) {
  APLA_REQUIRE(params && grads && exp_avg && exp_avg_sq && decay_mask && norm_ws && n > 0 && step >= 1, "apla_adamw_step: bad arguments");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(sumsq_kernel, dim3(NPART), dim3(256), 0, stream, grads, n, grad_scale, norm_ws);
  APLA_CHECK_LAUNCH("apla_adamw_step[sumsq]");
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, params, grads, exp_avg, exp_avg_sq, decay_mask, n, lr, weight_decay, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2), max_norm, grad_scale, norm_ws);
  APLA_CHECK_LAUNCH("apla_adamw_step[update]");
  return APLA_OK;
}
