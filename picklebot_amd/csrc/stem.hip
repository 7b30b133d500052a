// Dense 3-D convolution with a tiny input-channel count (the RGB stems).  The clip is read in
// place through element strides (NDHWC view of the dataloader tensor, or NCDHW), optionally
// straight from uint8 with the /255 folded in; the output is NDHWC in the compute dtype.
// K = Cin*kt*kh*kw <= 127 and Cout = 16: far too small for MFMA tiles to matter - the kernel is
// bound by reading the clip and writing the 16-channel map, so it is a plain VALU stencil with
// the (wave-uniform) filter broadcast from LDS.
#include "pb_common.h"

namespace {

constexpr int OCB = 16;          // output channels per thread / per block column
constexpr int MAX_CT = 127;      // Cin * taps

struct StemArgs {
  const void* x;
  float x_scale;
  const float* w;
  const float* bias;
  void* y;         // forward output, or dy in backward
  float* stats;
  float* dw;
  float* dbias;
  pb_stem_shape s;
  int64_t npix;
};

template <typename XT> __device__ __forceinline__ float ldx(const XT* p) { return (float)*p; }
template <> __device__ __forceinline__ float ldx<bf16_t>(const bf16_t* p) { return (float)*p; }

template <typename XT, typename T>
__global__ __launch_bounds__(256) void stem_fwd_kernel(StemArgs a) {
  __shared__ __attribute__((aligned(16))) float wl[MAX_CT * OCB];
  __shared__ float red[2 * OCB];
  const pb_stem_shape& s = a.s;
  const int tid = threadIdx.x;
  const int taps = s.kt * s.kh * s.kw, CT = s.Cin * taps;
  const int oc0 = blockIdx.y * OCB;
  for (int i = tid; i < CT * OCB; i += 256) {
    const int ct = i / OCB, oc = i - ct * OCB;
    wl[i] = (oc0 + oc < s.Cout) ? a.w[(int64_t)(oc0 + oc) * CT + ct] : 0.f;
  }
  if (tid < 2 * OCB) red[tid] = 0.f;
  __syncthreads();
  const int64_t pix = (int64_t)blockIdx.x * 256 + tid;
  const bool valid = pix < a.npix;
  float acc[OCB];
#pragma unroll
  for (int oc = 0; oc < OCB; ++oc) acc[oc] = (a.bias && oc0 + oc < s.Cout) ? a.bias[oc0 + oc] : 0.f;
  if (valid) {
    int64_t r = pix;
    const int wo = (int)(r % s.Wo); r /= s.Wo;
    const int ho = (int)(r % s.Ho); r /= s.Ho;
    const int to = (int)(r % s.To);
    const int n = (int)(r / s.To);
    const XT* xb = (const XT*)a.x + (int64_t)n * s.xs_n;
    for (int ci = 0; ci < s.Cin; ++ci)
      for (int at = 0; at < s.kt; ++at) {
        const int ti = to * s.st - s.pt + at;
        if (ti < 0 || ti >= s.Ti) continue;
        for (int ah = 0; ah < s.kh; ++ah) {
          const int hi = ho * s.sh - s.ph + ah;
          if (hi < 0 || hi >= s.Hi) continue;
          for (int aw = 0; aw < s.kw; ++aw) {
            const int wi = wo * s.sw - s.pw + aw;
            if (wi < 0 || wi >= s.Wi) continue;
            const float xv = ldx<XT>(xb + ci * s.xs_c + ti * s.xs_t + hi * s.xs_h + wi * s.xs_w) * a.x_scale;
            const float* wp = wl + ((ci * s.kt + at) * s.kh + ah) * s.kw * OCB + aw * OCB;
#pragma unroll
            for (int q = 0; q < OCB / 4; ++q) {
              const f32x4 wv = *reinterpret_cast<const f32x4*>(wp + 4 * q);
              acc[4 * q + 0] += xv * wv[0];
              acc[4 * q + 1] += xv * wv[1];
              acc[4 * q + 2] += xv * wv[2];
              acc[4 * q + 3] += xv * wv[3];
            }
          }
        }
      }
    T* yp = (T*)a.y + pix * s.Cout + oc0;
    constexpr int VEC = ElemTraits<T>::VEC;
    if (oc0 + OCB <= s.Cout && (s.Cout % VEC) == 0) {
#pragma unroll
      for (int q = 0; q < OCB / VEC; ++q) {
        Vec16<T> v;
#pragma unroll
        for (int i = 0; i < VEC; ++i) v.set(i, acc[q * VEC + i]);
        vec_store<T>(yp + q * VEC, v);
      }
    } else {
      for (int oc = 0; oc < OCB; ++oc)
        if (oc0 + oc < s.Cout) yp[oc] = from_f32<T>(acc[oc]);
    }
  }
  if (a.stats) {
    if (valid) {
#pragma unroll
      for (int oc = 0; oc < OCB; ++oc) {
        const float v = to_f32(from_f32<T>(acc[oc]));
        atomicAdd(&red[oc], v);
        atomicAdd(&red[OCB + oc], v * v);
      }
    }
    __syncthreads();
    if (tid < OCB && oc0 + tid < s.Cout) {
      atomicAdd(a.stats + oc0 + tid, red[tid]);
      atomicAdd(a.stats + s.Cout + oc0 + tid, red[OCB + tid]);
    }
  }
}

// Weight gradient: a 128-lane group owns the (ci,tap) columns (+1 lane for the bias) and walks
// output pixels; dy's 16 channels are uniform over the group, x is one gather per lane.
template <typename XT, typename T>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(StemArgs a) {
  const pb_stem_shape& s = a.s;
  const int tid = threadIdx.x;
  const int grp = tid >> 7, col = tid & 127;
  const int taps = s.kt * s.kh * s.kw, CT = s.Cin * taps;
  const int oc0 = blockIdx.y * OCB;
  const bool is_w = col < CT, is_b = col == CT;
  const int ci = is_w ? col / taps : 0;
  const int tap = is_w ? col - ci * taps : 0;
  const int aw = tap % s.kw, ah = (tap / s.kw) % s.kh, at = tap / (s.kw * s.kh);
  const T* dy = (const T*)a.y;
  float acc[OCB];
#pragma unroll
  for (int oc = 0; oc < OCB; ++oc) acc[oc] = 0.f;
  for (int64_t pix = (int64_t)blockIdx.x * 2 + grp; pix < a.npix; pix += (int64_t)gridDim.x * 2) {
    int64_t r = pix;
    const int wo = (int)(r % s.Wo); r /= s.Wo;
    const int ho = (int)(r % s.Ho); r /= s.Ho;
    const int to = (int)(r % s.To);
    const int n = (int)(r / s.To);
    float xv = is_b ? 1.f : 0.f;
    if (is_w) {
      const int ti = to * s.st - s.pt + at, hi = ho * s.sh - s.ph + ah, wi = wo * s.sw - s.pw + aw;
      if (ti >= 0 && ti < s.Ti && hi >= 0 && hi < s.Hi && wi >= 0 && wi < s.Wi)
        xv = ldx<XT>((const XT*)a.x + (int64_t)n * s.xs_n + ci * s.xs_c + ti * s.xs_t + hi * s.xs_h + wi * s.xs_w) * a.x_scale;
    }
    const T* dp = dy + pix * s.Cout + oc0;
#pragma unroll
    for (int oc = 0; oc < OCB; ++oc) {
      const float g = (oc0 + oc < s.Cout) ? to_f32(dp[oc]) : 0.f;
      acc[oc] += g * xv;
    }
  }
#pragma unroll
  for (int oc = 0; oc < OCB; ++oc) {
    if (oc0 + oc >= s.Cout) continue;
    if (is_w) atomicAdd(a.dw + ((int64_t)(oc0 + oc) * s.Cin + ci) * taps + tap, acc[oc]);
    else if (is_b && a.dbias) atomicAdd(a.dbias + oc0 + oc, acc[oc]);
  }
}

int check_stem(const pb_stem_shape* s, int x_dtype, pb_dtype dtype, const char* who) {
  PB_CHECK_ARG(s, "%s: null shape", who);
  PB_CHECK_ARG(s->N > 0 && s->Cin > 0 && s->Cout > 0 && s->Ti > 0 && s->Hi > 0 && s->Wi > 0 && s->To > 0 &&
               s->Ho > 0 && s->Wo > 0, "%s: non-positive dimension", who);
  PB_CHECK_ARG(s->Cin * s->kt * s->kh * s->kw <= MAX_CT, "%s: Cin*taps = %d exceeds %d", who,
               s->Cin * s->kt * s->kh * s->kw, MAX_CT);
  PB_CHECK_ARG((s->To - 1) * s->st - s->pt < s->Ti && (s->Ho - 1) * s->sh - s->ph < s->Hi &&
               (s->Wo - 1) * s->sw - s->pw < s->Wi, "%s: output extent inconsistent", who);
  PB_CHECK_ARG(x_dtype >= 0 && x_dtype <= 2, "%s: bad x_dtype %d", who, x_dtype);
  PB_CHECK_ARG(dtype == PB_F32 || dtype == PB_BF16, "%s: bad dtype", who);
  return 0;
}

}  // namespace

#define PB_STEM_DISPATCH(KERNEL, grid)                                                                   \
  do {                                                                                                   \
    if (dtype == PB_BF16) {                                                                              \
      if (x_dtype == 0) hipLaunchKernelGGL((KERNEL<float, bf16_t>), grid, dim3(256), 0, st, a);          \
      else if (x_dtype == 1) hipLaunchKernelGGL((KERNEL<bf16_t, bf16_t>), grid, dim3(256), 0, st, a);    \
      else hipLaunchKernelGGL((KERNEL<unsigned char, bf16_t>), grid, dim3(256), 0, st, a);               \
    } else {                                                                                             \
      if (x_dtype == 0) hipLaunchKernelGGL((KERNEL<float, float>), grid, dim3(256), 0, st, a);           \
      else if (x_dtype == 1) hipLaunchKernelGGL((KERNEL<bf16_t, float>), grid, dim3(256), 0, st, a);     \
      else hipLaunchKernelGGL((KERNEL<unsigned char, float>), grid, dim3(256), 0, st, a);                \
    }                                                                                                    \
  } while (0)

extern "C" int pb_stem_conv3d_fwd(const void* x, int x_dtype, float x_scale, const float* w, const float* bias,
                                  void* y, const pb_stem_shape* s, pb_dtype dtype, float* stats, int device,
                                  void* stream) {
  PB_CHECK_ARG(x && w && y, "pb_stem_conv3d_fwd: null pointer");
  if (int rc = check_stem(s, x_dtype, dtype, "pb_stem_conv3d_fwd")) return rc;
  PB_ENTER(device);
  hipStream_t st = (hipStream_t)stream;
  StemArgs a = {};
  a.x = x; a.x_scale = x_scale; a.w = w; a.bias = bias; a.y = y; a.stats = stats; a.s = *s;
  a.npix = (int64_t)s->N * s->To * s->Ho * s->Wo;
  dim3 grid((unsigned)pb_cdiv(a.npix, 256), (unsigned)pb_cdiv(s->Cout, OCB));
  PB_STEM_DISPATCH(stem_fwd_kernel, grid);
  PB_LAUNCH_CHECK();
  return 0;
}

extern "C" int pb_stem_conv3d_bwd_weight(const void* dy, const void* x, int x_dtype, float x_scale, float* dw,
                                         float* dbias, const pb_stem_shape* s, pb_dtype dtype, int device,
                                         void* stream) {
  PB_CHECK_ARG(dy && x && dw, "pb_stem_conv3d_bwd_weight: null pointer");
  if (int rc = check_stem(s, x_dtype, dtype, "pb_stem_conv3d_bwd_weight")) return rc;
  PB_ENTER(device);
  hipStream_t st = (hipStream_t)stream;
  StemArgs a = {};
  a.x = x; a.x_scale = x_scale; a.y = const_cast<void*>(dy); a.dw = dw; a.dbias = dbias; a.s = *s;
  a.npix = (int64_t)s->N * s->To * s->Ho * s->Wo;
  int64_t gx = pb_cdiv(a.npix, 2);
  if (gx > 2048) gx = 2048;
  dim3 grid((unsigned)gx, (unsigned)pb_cdiv(s->Cout, OCB));
  PB_STEM_DISPATCH(stem_wgrad_kernel, grid);
  PB_LAUNCH_CHECK();
  return 0;
}
