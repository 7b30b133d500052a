// Fused AdamW over a flat fp32 parameter bucket (the optimizer step that follows the path;
// the reference uses bitsandbytes' AdamW8bit, train.py:208-212 - this is the plain fp32-state form, same update rule
// as torch.optim.AdamW): one launch per bucket instead of torch's multi-tensor launches over ~170 small tensors.
//   p *= 1 - lr*wd;  m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;  p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// The step count t is read from device memory (so that a captured HIP graph replays with an advancing t), gradients can be
// unscaled on the fly (g * *inv_scale) and the update skipped when *found_inf != 0 (torch.amp.GradScaler semantics,
// train.py:283-289); pb_grad_check sets found_inf when a gradient is not finite.
#include "pb_common.h"

namespace {

struct AdamArgs {
  float* p;
  const float* g;
  float* m;
  float* v;
  int64_t n;
  float lr, b1, b2, eps, wd;
  const float* step;       // device scalar, already incremented for this step
  const float* inv_scale;  // optional
  const float* found_inf;  // optional
  const float* lr_wd;      // optional device float[2] = {lr, weight_decay}: overrides the by-value pair, so that a step
                           // replayed from a HIP graph follows a learning-rate schedule
};

__global__ __launch_bounds__(256) void adamw_flat_kernel(AdamArgs a) {
  if (a.found_inf && *a.found_inf != 0.f) return;
  const float t = *a.step;
  const float lr = a.lr_wd ? a.lr_wd[0] : a.lr, wd = a.lr_wd ? a.lr_wd[1] : a.wd;
  const float bc1 = 1.f - powf(a.b1, t), bc2 = 1.f - powf(a.b2, t);
  const float step_size = lr / bc1, inv_sqrt_bc2 = rsqrtf(bc2), decay = 1.f - lr * wd;
  const float gs = a.inv_scale ? *a.inv_scale : 1.f;
  const int64_t n4 = a.n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 p = reinterpret_cast<f32x4*>(a.p)[i], m = reinterpret_cast<f32x4*>(a.m)[i], v = reinterpret_cast<f32x4*>(a.v)[i];
    const f32x4 g = reinterpret_cast<const f32x4*>(a.g)[i] * gs;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      m[e] = a.b1 * m[e] + (1.f - a.b1) * g[e];
      v[e] = a.b2 * v[e] + (1.f - a.b2) * g[e] * g[e];
      p[e] = p[e] * decay - step_size * m[e] / (sqrtf(v[e]) * inv_sqrt_bc2 + a.eps);
    }
    reinterpret_cast<f32x4*>(a.p)[i] = p;
    reinterpret_cast<f32x4*>(a.m)[i] = m;
    reinterpret_cast<f32x4*>(a.v)[i] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {               // tail (n % 4 elements)
    const int64_t i = (n4 << 2) + threadIdx.x;
    const float g = a.g[i] * gs;
    const float m = a.b1 * a.m[i] + (1.f - a.b1) * g, v = a.b2 * a.v[i] + (1.f - a.b2) * g * g;
    a.m[i] = m; a.v[i] = v;
    a.p[i] = a.p[i] * decay - step_size * m / (sqrtf(v) * inv_sqrt_bc2 + a.eps);
  }
}

__global__ __launch_bounds__(256) void grad_check_kernel(const float* g, int64_t n, float* found_inf) {
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float x = g[i];
    bad |= !(fabsf(x) <= 3.4028234e38f);                          // inf or nan
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) *found_inf = 1.f;
}

}  // namespace

extern "C" int pb_adamw_flat(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                             float eps, float weight_decay, const float* step, const float* inv_scale,
                             const float* found_inf, const float* lr_wd, int device, void* stream) {
  PB_CHECK_ARG(p && g && m && v && step, "pb_adamw_flat: null pointer");
  PB_CHECK_ARG(n >= 0, "pb_adamw_flat: bad size");
  PB_CHECK_ARG(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 && ((uintptr_t)v % 16) == 0,
               "pb_adamw_flat: buffers must be 16-byte aligned");
  if (n == 0) return 0;
  PB_ENTER(device);
  AdamArgs a = {p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, inv_scale, found_inf, lr_wd};
  int64_t blocks = pb_cdiv(pb_cdiv(n, 4), 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adamw_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  PB_LAUNCH_CHECK();
  return 0;
}

extern "C" int pb_grad_check(const float* g, int64_t n, float* found_inf, int device, void* stream) {
  PB_CHECK_ARG(g && found_inf && n >= 0, "pb_grad_check: bad arguments");
  if (n == 0) return 0;
  PB_ENTER(device);
  int64_t blocks = pb_cdiv(n, 256 * 8);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(grad_check_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, n, found_inf);
  PB_LAUNCH_CHECK();
  return 0;
}
