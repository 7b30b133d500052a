// Squeeze-excite gate (SEBlock3D, mobilenet.py:11-26): the two 1x1x1 convolutions on the pooled
// [N][C] vector are tiny GEMVs - one workgroup per clip, wavefront-shuffle dot products.
#include "pb_common.h"

namespace {

constexpr int SE_MAX_C = 1 << 20;

// One wavefront per output element, lanes over the reduction index (coalesced weight rows,
// shuffle reduction): G*Cm + G*C wavefronts instead of G workgroups walking the weights serially.
//   MODE 0: hidden[g][j] = relu(b1[j] + W1[j,:] . (pool[g,:]*inv_count))          (out = Cm, red = C)
//   MODE 1: scale[g][c]  = hardsigmoid(b2[c] + W2[c,:] . hidden[g,:])              (out = C,  red = Cm)
template <int MODE>
__global__ __launch_bounds__(256) void se_gemv_kernel(const float* vec, float vscale, const float* w, const float* bias,
                                                      float* out, int G, int nout, int nred) {
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (wid >= (int64_t)G * nout) return;
  const int g = (int)(wid / nout), o = (int)(wid - (int64_t)g * nout);
  const float* wr = w + (int64_t)o * nred;
  const float* v = vec + (int64_t)g * nred;
  float s = 0.f;
  for (int c = lane; c < nred; c += 64) s += wr[c] * v[c];
  s = wave_sum(s) * vscale;
  if (lane == 0) {
    s += bias ? bias[o] : 0.f;
    out[wid] = MODE == 0 ? (s > 0.f ? s : 0.f) : fminf(fmaxf(s * (1.f / 6.f) + 0.5f, 0.f), 1.f);
  }
}

// Backward vectors.  The two matrix-transpose-vector products read the weight matrix along its
// contiguous dimension (lanes over outputs), split the reduction over the 4 waves of the workgroup
// with 4 independent accumulators each (16 loads in flight per lane), and combine in LDS:
//   out[g][o] = oscale * omask(g,o) * sum_r W[r][o] * v[g][r]
//   dh      (O = Cm, R = C ): W = W2 [C][Cm], v = dpre2,  omask = hidden[g][o] > 0
//   g_shift (O = C,  R = Cm): W = W1 [Cm][C], v = dh,     oscale = inv_count
constexpr int SE_RS = 16;     // reduction slices (waves) per workgroup: 60-240 dependent steps per thread with 4 were 12 us
__global__ __launch_bounds__(64 * SE_RS) void se_gemvT_kernel(const float* W, const float* v, int64_t vstride, float* out,
                                                              int64_t ostride, const float* omask_src, float oscale, int O,
                                                              int R) {
  __shared__ float part[SE_RS][64];
  const int ol = threadIdx.x & 63, rs = threadIdx.x >> 6;
  const int oblocks = (O + 63) / 64;
  const int g = blockIdx.x / oblocks, ob = blockIdx.x - g * oblocks;
  const int o = ob * 64 + ol;
  const float* vg = v + (int64_t)g * vstride;
  const int rq = (R + SE_RS - 1) / SE_RS, r0 = rs * rq, r1 = min(R, r0 + rq);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (o < O) {
    int r = r0;
    for (; r + 4 <= r1; r += 4) {
      s0 += W[(int64_t)r * O + o] * vg[r];
      s1 += W[(int64_t)(r + 1) * O + o] * vg[r + 1];
      s2 += W[(int64_t)(r + 2) * O + o] * vg[r + 2];
      s3 += W[(int64_t)(r + 3) * O + o] * vg[r + 3];
    }
    for (; r < r1; ++r) s0 += W[(int64_t)r * O + o] * vg[r];
  }
  part[rs][ol] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rs == 0 && o < O) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < SE_RS; ++i) s += part[i][ol];
    if (omask_src && !(omask_src[(int64_t)g * O + o] > 0.f)) s = 0.f;
    out[(int64_t)g * ostride + o] = s * oscale;
  }
}

// dpre2[g][c] = dscale * hardsigmoid'(.)  -> work[g][0..C)
__global__ __launch_bounds__(256) void se_dpre2_kernel(const float* dscale, const float* scale, float* work, int G, int C,
                                                       int Cm) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)G * C) return;
  const int g = (int)(idx / C), c = (int)(idx - (int64_t)g * C);
  const float sc = scale[idx];
  work[(int64_t)g * (C + Cm) + c] = (sc > 0.f && sc < 1.f) ? dscale[idx] * (1.f / 6.f) : 0.f;
}

// weight gradients: loops over the G clips, one thread per weight element (deterministic)
__global__ __launch_bounds__(256) void se_fc_bwd_w_kernel(const float* work, const float* pool, float inv_count,
                                                          const float* hidden, float* dw1, float* db1, float* dw2,
                                                          float* db2, int G, int C, int Cm) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t nW = (int64_t)C * Cm;
  const int stride = C + Cm;
  if (idx < nW) {                       // dW2[c][j] = sum_g dpre2[g][c] * h[g][j]
    const int c = (int)(idx / Cm), j = (int)(idx - (int64_t)c * Cm);
    float s = 0.f;
#pragma unroll 8
    for (int g = 0; g < G; ++g) s += work[(int64_t)g * stride + c] * hidden[(int64_t)g * Cm + j];
    dw2[idx] += s;
  } else if (idx < 2 * nW) {            // dW1[j][c] = sum_g dh[g][j] * mean[g][c]
    const int64_t i = idx - nW;
    const int j = (int)(i / C), c = (int)(i - (int64_t)j * C);
    float s = 0.f;
#pragma unroll 8
    for (int g = 0; g < G; ++g) s += work[(int64_t)g * stride + C + j] * pool[(int64_t)g * C + c] * inv_count;
    dw1[i] += s;
  } else if (idx < 2 * nW + C) {        // db2
    const int c = (int)(idx - 2 * nW);
    float s = 0.f;
#pragma unroll 8
    for (int g = 0; g < G; ++g) s += work[(int64_t)g * stride + c];
    if (db2) db2[c] += s;
  } else if (idx < 2 * nW + C + Cm) {   // db1
    const int j = (int)(idx - 2 * nW - C);
    float s = 0.f;
#pragma unroll 8
    for (int g = 0; g < G; ++g) s += work[(int64_t)g * stride + C + j];
    if (db1) db1[j] += s;
  }
}

}  // namespace

extern "C" int pb_se_fc_fwd(const float* pool, float inv_count, const float* w1, const float* b1, const float* w2,
                            const float* b2, float* scale, float* hidden, int G, int C, int Cm, int device,
                            void* stream) {
  PB_CHECK_ARG(pool && w1 && w2 && scale && hidden, "pb_se_fc_fwd: null pointer");
  PB_CHECK_ARG(G > 0 && C > 0 && Cm > 0 && C <= SE_MAX_C && Cm <= SE_MAX_C / 4, "pb_se_fc_fwd: bad sizes C=%d Cm=%d", C, Cm);
  PB_ENTER(device);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(se_gemv_kernel<0>, dim3((unsigned)pb_cdiv((int64_t)G * Cm, 4)), dim3(256), 0, st, pool, inv_count,
                     w1, b1, hidden, G, Cm, C);
  hipLaunchKernelGGL(se_gemv_kernel<1>, dim3((unsigned)pb_cdiv((int64_t)G * C, 4)), dim3(256), 0, st,
                     (const float*)hidden, 1.f, w2, b2, scale, G, C, Cm);
  PB_LAUNCH_CHECK();
  return 0;
}

extern "C" int pb_se_fc_bwd(const float* dscale, const float* pool, float inv_count, const float* scale,
                            const float* hidden, const float* w1, const float* w2, float* g_shift, float* dw1,
                            float* db1, float* dw2, float* db2, float* work, int G, int C, int Cm, int device,
                            void* stream) {
  PB_CHECK_ARG(dscale && pool && scale && hidden && w1 && w2 && g_shift && dw1 && dw2 && work, "pb_se_fc_bwd: null pointer");
  PB_CHECK_ARG(G > 0 && C > 0 && Cm > 0 && C <= SE_MAX_C && Cm <= SE_MAX_C / 4, "pb_se_fc_bwd: bad sizes C=%d Cm=%d", C, Cm);
  PB_ENTER(device);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(se_dpre2_kernel, dim3((unsigned)pb_cdiv((int64_t)G * C, 256)), dim3(256), 0, st, dscale, scale, work,
                     G, C, Cm);
  hipLaunchKernelGGL(se_gemvT_kernel, dim3((unsigned)(G * pb_cdiv(Cm, 64))), dim3(64 * SE_RS), 0, st, w2, (const float*)work,
                     (int64_t)(C + Cm), work + C, (int64_t)(C + Cm), hidden, 1.f, Cm, C);
  hipLaunchKernelGGL(se_gemvT_kernel, dim3((unsigned)(G * pb_cdiv(C, 64))), dim3(64 * SE_RS), 0, st, w1,
                     (const float*)(work + C), (int64_t)(C + Cm), g_shift, (int64_t)C, (const float*)nullptr, inv_count,
                     C, Cm);
  const int64_t total = 2LL * C * Cm + C + Cm;
  hipLaunchKernelGGL(se_fc_bwd_w_kernel, dim3((unsigned)pb_cdiv(total, 256)), dim3(256), 0, st, work, pool,
                     inv_count, hidden, dw1, db1, dw2, db2, G, C, Cm);
  PB_LAUNCH_CHECK();
  return 0;
}
