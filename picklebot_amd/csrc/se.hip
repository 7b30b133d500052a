// Squeeze-excite gate (SEBlock3D, mobilenet.py:11-26): the two 1x1x1 convolutions on the pooled
// [N][C] vector are tiny GEMVs - one workgroup per clip, wavefront-shuffle dot products.
#include "pb_common.h"

namespace {

constexpr int SE_MAX_C = 2048;   // pooled vector / hidden vector kept in LDS

// scale[g][c] = hardsigmoid(b2[c] + W2[c,:] . relu(b1 + W1 . (pool[g]*inv_count)))
__global__ __launch_bounds__(256) void se_fc_fwd_kernel(const float* pool, float inv_count, const float* w1,
                                                        const float* b1, const float* w2, const float* b2,
                                                        float* scale, float* hidden, int C, int Cm) {
  __shared__ float sm[SE_MAX_C];
  __shared__ float sh[SE_MAX_C / 4 + 8];
  const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int c = tid; c < C; c += 256) sm[c] = pool[(int64_t)g * C + c] * inv_count;
  __syncthreads();
  for (int j = wave; j < Cm; j += 4) {
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += w1[(int64_t)j * C + c] * sm[c];
    s = wave_sum(s);
    if (lane == 0) {
      s += b1 ? b1[j] : 0.f;
      s = s > 0.f ? s : 0.f;
      sh[j] = s;
      hidden[(int64_t)g * Cm + j] = s;
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float s = b2 ? b2[c] : 0.f;
    for (int j = 0; j < Cm; ++j) s += w2[(int64_t)c * Cm + j] * sh[j];
    scale[(int64_t)g * C + c] = fminf(fmaxf(s * (1.f / 6.f) + 0.5f, 0.f), 1.f);
  }
}

// per clip: dpre2 = dscale * hardsigmoid'(.), dh = relu'(h) * W2^T dpre2, g_shift = W1^T dh * inv_count
// work[g] = [dpre2 (C) | dh (Cm)]
__global__ __launch_bounds__(256) void se_fc_bwd_vec_kernel(const float* dscale, const float* scale,
                                                            const float* hidden, const float* w1, const float* w2,
                                                            float inv_count, float* g_shift, float* work, int C, int Cm) {
  __shared__ float sd[SE_MAX_C];
  __shared__ float sh[SE_MAX_C / 4 + 8];
  const int g = blockIdx.x, tid = threadIdx.x;
  float* wk = work + (int64_t)g * (C + Cm);
  for (int c = tid; c < C; c += 256) {
    const float s = scale[(int64_t)g * C + c];
    const float d = (s > 0.f && s < 1.f) ? dscale[(int64_t)g * C + c] * (1.f / 6.f) : 0.f;
    sd[c] = d;
    wk[c] = d;
  }
  __syncthreads();
  for (int j = tid; j < Cm; j += 256) {
    float s = 0.f;
    if (hidden[(int64_t)g * Cm + j] > 0.f)
      for (int c = 0; c < C; ++c) s += w2[(int64_t)c * Cm + j] * sd[c];
    sh[j] = s;
    wk[C + j] = s;
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float s = 0.f;
    for (int j = 0; j < Cm; ++j) s += w1[(int64_t)j * C + c] * sh[j];
    g_shift[(int64_t)g * C + c] = s * inv_count;
  }
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
    for (int g = 0; g < G; ++g) s += work[(int64_t)g * stride + c] * hidden[(int64_t)g * Cm + j];
    dw2[idx] += s;
  } else if (idx < 2 * nW) {            // dW1[j][c] = sum_g dh[g][j] * mean[g][c]
    const int64_t i = idx - nW;
    const int j = (int)(i / C), c = (int)(i - (int64_t)j * C);
    float s = 0.f;
    for (int g = 0; g < G; ++g) s += work[(int64_t)g * stride + C + j] * pool[(int64_t)g * C + c] * inv_count;
    dw1[i] += s;
  } else if (idx < 2 * nW + C) {        // db2
    const int c = (int)(idx - 2 * nW);
    float s = 0.f;
    for (int g = 0; g < G; ++g) s += work[(int64_t)g * stride + c];
    if (db2) db2[c] += s;
  } else if (idx < 2 * nW + C + Cm) {   // db1
    const int j = (int)(idx - 2 * nW - C);
    float s = 0.f;
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
  hipLaunchKernelGGL(se_fc_fwd_kernel, dim3(G), dim3(256), 0, (hipStream_t)stream, pool, inv_count, w1, b1, w2, b2,
                     scale, hidden, C, Cm);
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
  hipLaunchKernelGGL(se_fc_bwd_vec_kernel, dim3(G), dim3(256), 0, st, dscale, scale, hidden, w1, w2, inv_count,
                     g_shift, work, C, Cm);
  const int64_t total = 2LL * C * Cm + C + Cm;
  hipLaunchKernelGGL(se_fc_bwd_w_kernel, dim3((unsigned)pb_cdiv(total, 256)), dim3(256), 0, st, work, pool,
                     inv_count, hidden, dw1, db1, dw2, db2, G, C, Cm);
  PB_LAUNCH_CHECK();
  return 0;
}
