// Fused global-norm clip + AdamW over the flat APLA-trainable buffer (2L+2 tensors laid out back to back).
// Two launches, no atomics, no host synchronisation: (1) 256 workgroups write partial sums of squares of
// grad*grad_scale; (2) every workgroup re-reduces those 256 partials in a fixed order (deterministic), derives the
// clip coefficient on device and applies the AdamW update in the exact operation order of torch.optim.AdamW.
#include "common.h"

namespace {
constexpr int NPART = 256;

// One thread's share of sum (g[i] * scale)^2 over [0, n): 16-byte loads, four of them in flight per trip (a one-float-per-trip loop
// is a chain of dependent HBM round trips: 146 us for the 97 MB of the self-supervised step's gradients, 0.66 TB/s).  The order of
// the additions is fixed by (n, grid), so the result is deterministic.
__device__ inline float sumsq_strided(const float* __restrict__ g, long n, float scale) {
  const long tid = (long)blockIdx.x * 256 + threadIdx.x, nthr = (long)NPART * 256;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  long n4 = 0;
  if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
    n4 = n / 4;
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
    long i = tid;
    for (; i + 3 * nthr < n4; i += 4 * nthr) {
      const f32x4 a = g4[i], b = g4[i + nthr], c = g4[i + 2 * nthr], d = g4[i + 3 * nthr];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float va = a[e] * scale, vb = b[e] * scale, vc = c[e] * scale, vd = d[e] * scale;
        s0 += va * va; s1 += vb * vb; s2 += vc * vc; s3 += vd * vd;
      }
    }
    for (; i < n4; i += nthr) {
      const f32x4 a = g4[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float v = a[e] * scale; s0 += v * v; }
    }
  }
  for (long i = n4 * 4 + tid; i < n; i += nthr) {
    const float v = g[i] * scale;
    s1 += v * v;
  }
  return (s0 + s1) + (s2 + s3);
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, float grad_scale,
                                                    float* __restrict__ ws) {
  __shared__ float red[4];
  float s = sumsq_strided(g, n, grad_scale);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) ws[2 + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, const uint8_t* __restrict__ decay, long n,
                                                    float lr, float wd, float b1, float b2, float eps, float bc1,
                                                    float bc2_sqrt, float max_norm, float grad_scale,
                                                    float* __restrict__ ws, int track_step, float log_b1, float log_b2) {
  // track_step > 0 (apla_adamw_step): ws[260 + (step & 1)] holds the number of updates SKIPPED so far (non-finite gradient
  // norm); this call reads that slot and block 0 writes the new count into the other slot, which the next call (step + 1)
  // reads — no workgroup reads what another writes in the same launch.  The bias corrections then use the count of updates
  // actually applied, as GradScaler + torch.optim.AdamW do (a skipped step does not advance `step`).
  __shared__ float coef_s, bc1_s, bc2s_s;
  if (threadIdx.x < 64) {
    float s = 0.f;
    for (int i = threadIdx.x; i < NPART; i += 64) s += ws[2 + i];
    s = wave_sum(s);
    if (threadIdx.x == 0) {
      const float norm = sqrtf(s);
      float coef = 1.0f;
      if (max_norm > 0.f) { coef = max_norm / (norm + 1e-6f); coef = coef < 1.0f ? coef : 1.0f; }
      // a non-finite norm (fp16 gradients that overflowed under the loss scale) skips the update, like GradScaler.step
      const bool ok = isfinite(norm);
      coef_s = ok ? coef * grad_scale : __builtin_nanf("");
      float c1 = bc1, c2s = bc2_sqrt;
      if (track_step > 0) {
        const float skipped = ws[260 + (track_step & 1)];
        if (skipped > 0.f) {
          const float t = fmaxf((float)track_step - skipped, 1.0f);
          // 1 - b^t as -expm1(t log b) with log b from the host in double: `1 - powf(b, t)` cancels (1 - 0.999^t loses ~2e-5
          // at small t), and the host path of the unskipped case uses double pow
          c1 = -expm1f(t * log_b1);
          c2s = sqrtf(-expm1f(t * log_b2));
        }
        if (blockIdx.x == 0) ws[260 + ((track_step + 1) & 1)] = skipped + (ok ? 0.f : 1.f);
      }
      bc1_s = c1; bc2s_s = c2s;
      if (blockIdx.x == 0) { ws[0] = s; ws[1] = norm; }
    }
  }
  __syncthreads();
  const float coef = coef_s;
  if (coef != coef) return;
  const float step_size = lr / bc1_s;
  const float bc2_sqrt_ = bc2s_s;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float gi = g[i] * coef;
    float pi = p[i];
    if (decay[i]) pi *= (1.0f - lr * wd);
    const float mi = m[i] * b1 + gi * (1.0f - b1);
    const float vi = v[i] * b2 + gi * gi * (1.0f - b2);
    const float denom = sqrtf(vi) / bc2_sqrt_ + eps;
    pi -= step_size * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi; g[i] = gi;
  }
}
// ---- dynamic loss scaling (torch.cuda.amp.GradScaler semantics, defaults/trainer.py:129-138) kept on the device ----
// scaler[0..2] / scaler[3..5]: two slots of {scale, growth_tracker, optimizer steps taken}; a call reads slot `parity` and
// writes slot parity^1 (no workgroup ever reads what another one writes in the same launch); scaler[6] = the scale the NEXT
// backward must use (read by the step's loss-gradient scaling), scaler[7] = 1 if this call skipped the update.
__global__ __launch_bounds__(256) void sumsq_dyn_kernel(const float* __restrict__ g, long n, float grad_scale,
                                                        const float* __restrict__ scaler, int parity,
                                                        float* __restrict__ ws) {
  __shared__ float red[4];
  const float gs = grad_scale / scaler[3 * parity];
  float s = sumsq_strided(g, n, gs);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) ws[2 + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void adamw_dyn_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, const uint8_t* __restrict__ decay, long n,
                                                        float lr, float wd, float b1, float b2, float eps, float max_norm,
                                                        float grad_scale, float* __restrict__ scaler, int parity,
                                                        float growth, float backoff, int interval,
                                                        float* __restrict__ ws, float log_b1, float log_b2) {
  __shared__ float coef_s, bc1_s, bc2s_s;
  if (threadIdx.x < 64) {
    float s = 0.f;
    for (int i = threadIdx.x; i < NPART; i += 64) s += ws[2 + i];
    s = wave_sum(s);
    if (threadIdx.x == 0) {
      const float* cur = scaler + 3 * parity;
      const float scale = cur[0], tracker = cur[1], steps = cur[2];
      const float norm = sqrtf(s);
      const bool ok = isfinite(norm);
      float coef = 1.0f;
      if (max_norm > 0.f) { coef = max_norm / (norm + 1e-6f); coef = coef < 1.0f ? coef : 1.0f; }
      coef_s = ok ? coef * grad_scale / scale : __builtin_nanf("");
      const float t = steps + 1.0f;  // torch AdamW bias corrections with the count of steps actually taken
      bc1_s = -expm1f(t * log_b1);            // 1 - b^t without the cancellation of 1 - powf(b, t)
      bc2s_s = sqrtf(-expm1f(t * log_b2));
      if (blockIdx.x == 0) {
        float* nxt = scaler + 3 * (parity ^ 1);
        float nscale = scale, ntr = tracker + 1.0f;
        if (!ok) { nscale = scale * backoff; ntr = 0.f; }
        else if (ntr >= (float)interval) { nscale = scale * growth; ntr = 0.f; }
        nxt[0] = nscale; nxt[1] = ntr; nxt[2] = ok ? t : steps;
        scaler[6] = nscale;
        scaler[7] = ok ? 0.f : 1.f;
        ws[0] = s; ws[1] = norm;
      }
    }
  }
  __syncthreads();
  const float coef = coef_s;
  if (coef != coef) return;
  const float step_size = lr / bc1_s, bc2_sqrt = bc2s_s;
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

extern "C" int apla_adamw_step_dynamic(float* params, float* grads, float* exp_avg, float* exp_avg_sq,
                                       const uint8_t* decay_mask, long n, float lr, float weight_decay, float beta1,
                                       float beta2, float eps, float max_norm, float grad_scale, float* scaler,
                                       int parity, float growth_factor, float backoff_factor, int growth_interval,
                                       float* norm_ws, hipStream_t stream) {
  APLA_REQUIRE(params && grads && exp_avg && exp_avg_sq && decay_mask && norm_ws && scaler && n > 0, "apla_adamw_step_dynamic: bad arguments");
  APLA_REQUIRE((parity == 0 || parity == 1) && growth_factor >= 1.f && backoff_factor > 0.f && backoff_factor <= 1.f && growth_interval >= 1,
               "apla_adamw_step_dynamic: bad scaler settings");
  hipLaunchKernelGGL(sumsq_dyn_kernel, dim3(NPART), dim3(256), 0, stream, grads, n, grad_scale, scaler, parity, norm_ws);
  APLA_CHECK_LAUNCH("apla_adamw_step_dynamic[sumsq]");
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adamw_dyn_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, params, grads, exp_avg, exp_avg_sq, decay_mask, n, lr, weight_decay, beta1, beta2, eps, max_norm, grad_scale, scaler, parity, growth_factor, backoff_factor, growth_interval, norm_ws, (float)log((double)beta1), (float)log((double)beta2));
  APLA_CHECK_LAUNCH("apla_adamw_step_dynamic[update]");
  return APLA_OK;
}

// The two phases of apla_adamw_step as separate entry points: one global norm over the whole trainable buffer, then the
// update applied per contiguous range with its own step count (torch.optim.AdamW keeps `step` per tensor and skips tensors
// whose .grad is None — the DINOv2 trainer cancels the prototype layer's gradients during the first epoch AFTER clipping,
// self_supervised/dinov2/trainer.py:84-90,127-130).
extern "C" int apla_grad_sumsq(const float* grads, long n, float grad_scale, float* norm_ws, hipStream_t stream) {
  APLA_REQUIRE(grads && norm_ws && n > 0, "apla_grad_sumsq: bad arguments");
  hipLaunchKernelGGL(sumsq_kernel, dim3(NPART), dim3(256), 0, stream, grads, n, grad_scale, norm_ws);
  APLA_CHECK_LAUNCH("apla_grad_sumsq");
  return APLA_OK;
}

extern "C" int apla_adamw_apply(float* params, float* grads, float* exp_avg, float* exp_avg_sq, const uint8_t* decay_mask,
                                long n, float lr, float weight_decay, float beta1, float beta2, float eps, int step,
                                float max_norm, float grad_scale, float* norm_ws, hipStream_t stream) {
  APLA_REQUIRE(params && grads && exp_avg && exp_avg_sq && decay_mask && norm_ws && n > 0 && step >= 1, "apla_adamw_apply: bad arguments");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, params, grads, exp_avg, exp_avg_sq, decay_mask, n, lr, weight_decay, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2), max_norm, grad_scale, norm_ws, 0, 0.f, 0.f);
  APLA_CHECK_LAUNCH("apla_adamw_apply");
  return APLA_OK;
}

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
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, params, grads, exp_avg, exp_avg_sq, decay_mask, n, lr, weight_decay, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2), max_norm, grad_scale, norm_ws, step, (float)log((double)beta1), (float)log((double)beta2));
  APLA_CHECK_LAUNCH("apla_adamw_step[update]");
  return APLA_OK;
}

// ------------------------------------------------------------------------------------------------ EMA teacher
// teacher = m * teacher + (1 - m) * student over one flat range (self_supervised/dinov2/models.py:443-453 runs torch._foreach_mul_ then
// torch._foreach_add_(alpha = 1 - m) over the parameter lists: two passes over the teacher, 126 us at config 4; one pass here).  The
// rounding is theirs: the product t * m rounded to fp32 (the first pass stores it), then one fused multiply-add with the student.
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ t, const float* __restrict__ s, long n, float m, float om) {
  const long stride = (long)gridDim.x * 1024;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n) {
      f32x4 a = *(const f32x4*)(t + i);
      const f32x4 b = *(const f32x4*)(s + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] = __fmaf_rn(b[e], om, __fmul_rn(a[e], m));
      *(f32x4*)(t + i) = a;
    } else {
      for (long k = i; k < n; ++k) t[k] = __fmaf_rn(s[k], om, __fmul_rn(t[k], m));
    }
  }
}

extern "C" int apla_ema_update(float* teacher, const float* student, long n, double m, hipStream_t stream) {
  APLA_REQUIRE(teacher && student && n > 0 && apla_aligned16(teacher) && apla_aligned16(student), "apla_ema_update: two 16-byte aligned fp32 ranges expected");
  long blocks = (n + 1023) / 1024;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(ema_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, teacher, student, n, (float)m, (float)(1.0 - m));   // torch casts the two Python scalars m and 1 - m
  APLA_CHECK_LAUNCH("apla_ema_update");
  return APLA_OK;
}
