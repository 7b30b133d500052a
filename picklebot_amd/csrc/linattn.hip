// MobileViT-V2 separable ("linear") self-attention generalised to 3-D patches: the part of
// mobilevitv2.py:28-55 between qkv_proj and out_proj.  The reference's tensor is (batch, channels, pixels-in-patch,
// patches); here the tokens of the 3-D unfold (token_ops.hip) are rows [G = batch*pixels-in-patch][n = patches] and
// the projection output keeps the reference's column order  [ q (1) | k (D) | v (D) ]  (mobilevitv2.py:37).
//   scores[g,i] = softmax_i(q[g,i])                                   (:41)
//   ctx[g,d]    = sum_i scores[g,i] * k[g,i,d]                        (:46-48)
//   out[g,i,d]  = relu(v[g,i,d]) * ctx[g,d]                           (:52)
// Every sum runs in a fixed order (per-thread strided partials, then an LDS tree): results repeat bit for bit.
// All of it is HBM-bound row streaming: ~(1+2D)+D elements per token forward.
#include "pb_common.h"

namespace {

constexpr int LA_T = 256;

__device__ __forceinline__ float block_tree(float v, float* red, bool is_max) {
  const int tid = threadIdx.x;
  red[tid] = v;
  __syncthreads();
  for (int s = LA_T / 2; s > 0; s >>= 1) {
    if (tid < s) red[tid] = is_max ? fmaxf(red[tid], red[tid + s]) : red[tid] + red[tid + s];
    __syncthreads();
  }
  const float r = red[0];
  __syncthreads();
  return r;
}

// per-group column sums  acc[d] = sum_i wrow(i) * f(src[i][d])  with the 256 threads split into (segments of rows) x (DQ
// columns); the segment partials are added in segment order through LDS.
template <int MODE, typename T_>   // MODE 0: scores * k        MODE 1: dout * relu(v)
__device__ __forceinline__ void group_colsum(const T_* qkv, const T_* dout, const float* scores, float* dst, float* red,
                                             int64_t row0, int n, int D, int L) {
  const int tid = threadIdx.x;
  int DQ = 64;
  while (DQ < D) DQ <<= 1;                 // D <= 256
  const int nseg = LA_T / DQ, seg = tid / DQ, d = tid % DQ;
  float acc0 = 0.f, acc1 = 0.f;
  if (d < D) {
    int i = seg;
    for (; i + nseg < n; i += 2 * nseg) {
      const int64_t r0 = row0 + i, r1 = r0 + nseg;
      if constexpr (MODE == 0) {
        acc0 += scores[r0] * to_f32(qkv[r0 * L + 1 + d]);
        acc1 += scores[r1] * to_f32(qkv[r1 * L + 1 + d]);
      } else {
        acc0 += to_f32(dout[r0 * D + d]) * fmaxf(to_f32(qkv[r0 * L + 1 + D + d]), 0.f);
        acc1 += to_f32(dout[r1 * D + d]) * fmaxf(to_f32(qkv[r1 * L + 1 + D + d]), 0.f);
      }
    }
    if (i < n) {
      const int64_t r0 = row0 + i;
      if constexpr (MODE == 0) acc0 += scores[r0] * to_f32(qkv[r0 * L + 1 + d]);
      else acc0 += to_f32(dout[r0 * D + d]) * fmaxf(to_f32(qkv[r0 * L + 1 + D + d]), 0.f);
    }
  }
  red[tid] = acc0 + acc1;
  __syncthreads();
  if (tid < DQ && tid < D) {
    float s = red[tid];
    for (int k = 1; k < nseg; ++k) s += red[k * DQ + tid];
    dst[tid] = s;
  }
  __syncthreads();
}

template <typename T_>
__global__ __launch_bounds__(LA_T) void linattn_ctx_kernel(const T_* qkv, float* scores, float* ctx, int n, int D) {
  __shared__ float red[LA_T];
  const int tid = threadIdx.x, L = 1 + 2 * D;
  const int64_t row0 = (int64_t)blockIdx.x * n;
  float m = -INFINITY;
  for (int i = tid; i < n; i += LA_T) m = fmaxf(m, to_f32(qkv[(row0 + i) * L]));
  m = block_tree(m, red, true);
  float s = 0.f;
  for (int i = tid; i < n; i += LA_T) s += __expf(to_f32(qkv[(row0 + i) * L]) - m);
  s = block_tree(s, red, false);
  const float inv = 1.f / s;
  for (int i = tid; i < n; i += LA_T) scores[row0 + i] = __expf(to_f32(qkv[(row0 + i) * L]) - m) * inv;
  __syncthreads();                         // this workgroup's own global stores, read back below
  group_colsum<0, T_>(qkv, nullptr, scores, ctx + (int64_t)blockIdx.x * D, red, row0, n, D, L);
}

template <typename T_>
__global__ __launch_bounds__(LA_T) void linattn_apply_kernel(const T_* qkv, const float* ctx, T_* out, int64_t rows, int n,
                                                             int D) {
  const int L = 1 + 2 * D;
  const int64_t total = rows * D;
  for (int64_t e = (int64_t)blockIdx.x * LA_T + threadIdx.x; e < total; e += (int64_t)gridDim.x * LA_T) {
    const int64_t r = e / D;
    const int d = (int)(e - r * D);
    out[e] = from_f32<T_>(fmaxf(to_f32(qkv[r * L + 1 + D + d]), 0.f) * ctx[(r / n) * D + d]);
  }
}

template <typename T_>
__global__ __launch_bounds__(LA_T) void linattn_bwd_ctx_kernel(const T_* qkv, const T_* dout, float* dctx, int n, int D) {
  __shared__ float red[LA_T];
  group_colsum<1, T_>(qkv, dout, nullptr, dctx + (int64_t)blockIdx.x * D, red, (int64_t)blockIdx.x * n, n, D, 1 + 2 * D);
}

// one wavefront per token row:  dv = dout * ctx * [v > 0],  dk = score * dctx,  dscore = <k, dctx>
template <typename T_>
__global__ __launch_bounds__(LA_T) void linattn_bwd_rows_kernel(const T_* qkv, const float* scores, const float* ctx,
                                                                const T_* dout, const float* dctx, float* dscore, T_* dqkv,
                                                                int64_t rows, int n, int D) {
  const int lane = threadIdx.x & 63, L = 1 + 2 * D;
  for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
    const int64_t g = r / n;
    const float a = scores[r];
    float dot = 0.f;
    for (int d = lane; d < D; d += 64) {
      const float dc = dctx[g * D + d];
      const float v = to_f32(qkv[r * L + 1 + D + d]);
      dot += to_f32(qkv[r * L + 1 + d]) * dc;
      dqkv[r * L + 1 + d] = from_f32<T_>(a * dc);
      dqkv[r * L + 1 + D + d] = from_f32<T_>(v > 0.f ? to_f32(dout[r * D + d]) * ctx[g * D + d] : 0.f);
    }
    dot = wave_sum(dot);
    if (lane == 0) dscore[r] = dot;
  }
}

// softmax backward over the patches of one group:  dq = score * (dscore - sum_j score_j dscore_j)
template <typename T_>
__global__ __launch_bounds__(LA_T) void linattn_bwd_q_kernel(const float* scores, const float* dscore, T_* dqkv, int n, int D) {
  __shared__ float red[LA_T];
  const int tid = threadIdx.x, L = 1 + 2 * D;
  const int64_t row0 = (int64_t)blockIdx.x * n;
  float s = 0.f;
  for (int i = tid; i < n; i += LA_T) s += scores[row0 + i] * dscore[row0 + i];
  s = block_tree(s, red, false);
  for (int i = tid; i < n; i += LA_T) dqkv[(row0 + i) * L] = from_f32<T_>(scores[row0 + i] * (dscore[row0 + i] - s));
}

template <typename T_>
int fwd_t(const void* qkv, float* scores, float* ctx, void* out, int G, int n, int D, hipStream_t st) {
  const int64_t rows = (int64_t)G * n;
  hipLaunchKernelGGL(linattn_ctx_kernel<T_>, dim3(G), dim3(LA_T), 0, st, (const T_*)qkv, scores, ctx, n, D);
  PB_LAUNCH_CHECK();
  int64_t gx = pb_cdiv(rows * D, (int64_t)LA_T * 4);
  if (gx > 8192) gx = 8192;
  hipLaunchKernelGGL(linattn_apply_kernel<T_>, dim3((unsigned)gx), dim3(LA_T), 0, st, (const T_*)qkv, (const float*)ctx,
                     (T_*)out, rows, n, D);
  PB_LAUNCH_CHECK();
  return 0;
}

template <typename T_>
int bwd_t(const void* qkv, const float* scores, const float* ctx, const void* dout, float* dctx, float* dscore, void* dqkv,
          int G, int n, int D, hipStream_t st) {
  const int64_t rows = (int64_t)G * n;
  hipLaunchKernelGGL(linattn_bwd_ctx_kernel<T_>, dim3(G), dim3(LA_T), 0, st, (const T_*)qkv, (const T_*)dout, dctx, n, D);
  PB_LAUNCH_CHECK();
  int64_t gx = pb_cdiv(rows, 4);
  if (gx > 16384) gx = 16384;
  hipLaunchKernelGGL(linattn_bwd_rows_kernel<T_>, dim3((unsigned)gx), dim3(LA_T), 0, st, (const T_*)qkv, scores, ctx,
                     (const T_*)dout, (const float*)dctx, dscore, (T_*)dqkv, rows, n, D);
  PB_LAUNCH_CHECK();
  hipLaunchKernelGGL(linattn_bwd_q_kernel<T_>, dim3(G), dim3(LA_T), 0, st, scores, (const float*)dscore, (T_*)dqkv, n, D);
  PB_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int pb_linattn_fwd(const void* qkv, float* scores, float* ctx, void* out, int G, int n, int D, pb_dtype dtype,
                              int device, void* stream) {
  PB_CHECK_ARG(qkv && scores && ctx && out, "pb_linattn_fwd: null pointer");
  PB_CHECK_ARG(G >= 0 && n > 0 && D > 0 && D <= 256, "pb_linattn_fwd: need n > 0 and 0 < D <= 256 (G=%d n=%d D=%d)", G, n, D);
  PB_CHECK_ARG(dtype == PB_F32 || dtype == PB_BF16, "pb_linattn_fwd: bad dtype");
  if (G == 0) return 0;
  PB_ENTER(device);
  return dtype == PB_BF16 ? fwd_t<bf16_t>(qkv, scores, ctx, out, G, n, D, (hipStream_t)stream)
                          : fwd_t<float>(qkv, scores, ctx, out, G, n, D, (hipStream_t)stream);
}

extern "C" int pb_linattn_bwd(const void* qkv, const float* scores, const float* ctx, const void* dout, float* dctx,
                              float* dscore, void* dqkv, int G, int n, int D, pb_dtype dtype, int device, void* stream) {
  PB_CHECK_ARG(qkv && scores && ctx && dout && dctx && dscore && dqkv, "pb_linattn_bwd: null pointer");
  PB_CHECK_ARG(G >= 0 && n > 0 && D > 0 && D <= 256, "pb_linattn_bwd: need n > 0 and 0 < D <= 256 (G=%d n=%d D=%d)", G, n, D);
  PB_CHECK_ARG(dtype == PB_F32 || dtype == PB_BF16, "pb_linattn_bwd: bad dtype");
  if (G == 0) return 0;
  PB_ENTER(device);
  return dtype == PB_BF16 ? bwd_t<bf16_t>(qkv, scores, ctx, dout, dctx, dscore, dqkv, G, n, D, (hipStream_t)stream)
                          : bwd_t<float>(qkv, scores, ctx, dout, dctx, dscore, dqkv, G, n, D, (hipStream_t)stream);
}
