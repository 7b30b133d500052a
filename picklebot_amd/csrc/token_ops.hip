// Token-side helpers of the MobileViT block (mobilevit.py:121-143): the unfold/fold patch
// rearrangements (pure row permutations with zero padding / cropping), LayerNorm without affine,
// and strided column copies (channel concat / split).
#include "pb_common.h"

namespace {

// ---- unfold / fold -------------------------------------------------------------------------------
// tokens [B][P = pt*ph*pw][n = t*h*w][D]  <->  voxels [B][T][H][W][D] with T <= t*pt etc. (the
// reference zero-pads up to multiples of the patch and crops after folding).  One workgroup row-lane
// per token: source and destination rows are both contiguous D-vectors.
//   to_tokens != 0: tok[b,p,i,:] = vox[b, tt*pt+a, hh*ph+c, ww*pw+e, :] (0 outside)      "unfold"
//   to_tokens == 0: vox[b,T,H,W,:] = tok[...]                                            "fold + crop"
struct UfArgs {
  const void* src;
  void* dst;
  int B, D, T, H, W, pt, ph, pw, t, h, w;
  int to_tokens;
  int CV, CVB, P;
  int64_t rows;
};

template <typename T_>
__global__ __launch_bounds__(256) void unfold_kernel(UfArgs a) {
  constexpr int VEC = ElemTraits<T_>::VEC;
  const int tid = threadIdx.x;
  const int cvl = tid % a.CVB, p = tid / a.CVB;
  if (p >= a.P) return;
  const int npatch = a.t * a.h * a.w, PP = a.pt * a.ph * a.pw;
  const T_* src = (const T_*)a.src;
  T_* dst = (T_*)a.dst;
  for (int64_t r = (int64_t)blockIdx.x * a.P + p; r < a.rows; r += (int64_t)gridDim.x * a.P) {
    int64_t tok_row, vox_row;
    bool inside = true;
    if (a.to_tokens) {            // r enumerates token rows
      int64_t q = r;
      const int i = (int)(q % npatch); q /= npatch;
      const int pp = (int)(q % PP);
      const int b = (int)(q / PP);
      const int ww = i % a.w, hh = (i / a.w) % a.h, tt = i / (a.w * a.h);
      const int e = pp % a.pw, c = (pp / a.pw) % a.ph, aa = pp / (a.pw * a.ph);
      const int T0 = tt * a.pt + aa, H0 = hh * a.ph + c, W0 = ww * a.pw + e;
      inside = T0 < a.T && H0 < a.H && W0 < a.W;
      tok_row = r;
      vox_row = (((int64_t)b * a.T + T0) * a.H + H0) * a.W + W0;
    } else {                      // r enumerates voxel rows (always inside the padded grid)
      int64_t q = r;
      const int W0 = (int)(q % a.W); q /= a.W;
      const int H0 = (int)(q % a.H); q /= a.H;
      const int T0 = (int)(q % a.T);
      const int b = (int)(q / a.T);
      const int tt = T0 / a.pt, aa = T0 - tt * a.pt, hh = H0 / a.ph, c = H0 - hh * a.ph, ww = W0 / a.pw, e = W0 - ww * a.pw;
      const int pp = (aa * a.ph + c) * a.pw + e, i = (tt * a.h + hh) * a.w + ww;
      vox_row = r;
      tok_row = ((int64_t)b * PP + pp) * npatch + i;
    }
    const int64_t srow = a.to_tokens ? vox_row : tok_row, drow = a.to_tokens ? tok_row : vox_row;
    for (int cv = cvl; cv < a.CV; cv += a.CVB) {
      Vec16<T_> v = vec_zero<T_>();
      if (inside) v = vec_load<T_>(src + srow * a.D + cv * VEC);
      vec_store<T_>(dst + drow * a.D + cv * VEC, v);
    }
  }
}

// ---- LayerNorm (no affine) -----------------------------------------------------------------------
// one wavefront per row, 4 elements per lane (D <= 256), shuffle reductions
//   forward : y = (x - mean) * rstd, saves mean/rstd
//   backward: dx = rstd * (dy - mean(dy) - xhat * mean(dy * xhat))
template <typename T_, bool BWD>
__global__ __launch_bounds__(256) void layernorm_kernel(const T_* x, const T_* dy, T_* out, float* stat, int64_t rows, int D,
                                                        float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int c = lane * 4;
  const bool ok = c < D;                 // D % 4 == 0
  float v[4] = {0.f, 0.f, 0.f, 0.f}, g[4] = {0.f, 0.f, 0.f, 0.f};
  if (ok) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = to_f32(x[row * D + c + i]);
    if constexpr (BWD) {
#pragma unroll
      for (int i = 0; i < 4; ++i) g[i] = to_f32(dy[row * D + c + i]);
    }
  }
  float mean, rstd;
  if constexpr (!BWD) {
    const float s = wave_sum(v[0] + v[1] + v[2] + v[3]);
    mean = s / D;
    float q = 0.f;
    if (ok) {
#pragma unroll
      for (int i = 0; i < 4; ++i) q += (v[i] - mean) * (v[i] - mean);
    }
    q = wave_sum(q);
    rstd = rsqrtf(q / D + eps);
    if (lane == 0) {
      stat[row * 2] = mean;
      stat[row * 2 + 1] = rstd;
    }
    if (ok) {
#pragma unroll
      for (int i = 0; i < 4; ++i) out[row * D + c + i] = from_f32<T_>((v[i] - mean) * rstd);
    }
  } else {
    mean = stat[row * 2];
    rstd = stat[row * 2 + 1];
    float sg = 0.f, sgx = 0.f;
    if (ok) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float xh = (v[i] - mean) * rstd;
        sg += g[i];
        sgx += g[i] * xh;
      }
    }
    sg = wave_sum(sg) / D;
    sgx = wave_sum(sgx) / D;
    if (ok) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float xh = (v[i] - mean) * rstd;
        out[row * D + c + i] = from_f32<T_>(rstd * (g[i] - sg - xh * sgx));
      }
    }
  }
}

// ---- the WIDE token stream (round 4) ------------------------------------------------------------
// Transformer.forward's `x = attn(x) + x; x = ff(x) + x` (mobilevit.py:100-104) with the running sum x kept in fp32
// while every GEMM operand stays bf16: LayerNorm reads the fp32 stream and writes the bf16 GEMM input; the residual
// add reads the fp32 stream and the bf16 block output and writes the fp32 stream (and, when the next consumer is a GEMM,
// its bf16 rounding) in ONE pass.  Backward runs the same two kernels on the gradient stream.
//   forward : xn = bf16((x - mean) * rstd), saves mean/rstd
//   backward: out32 = res32 + rstd * (dy - mean(dy) - xhat * mean(dy * xhat));  out_bf16 = bf16(out32) (optional)
template <bool BWD>
__global__ __launch_bounds__(256) void layernorm_wide_kernel(const float* x, const bf16_t* dy, const float* res, bf16_t* out_bf,
                                                             float* out32, float* stat, int64_t rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int c = lane * 4;
  const bool ok = c < D;                 // D % 4 == 0
  float v[4] = {0.f, 0.f, 0.f, 0.f}, g[4] = {0.f, 0.f, 0.f, 0.f}, r[4] = {0.f, 0.f, 0.f, 0.f};
  if (ok) {
    const float4 xv = *(const float4*)(x + row * D + c);
    v[0] = xv.x; v[1] = xv.y; v[2] = xv.z; v[3] = xv.w;
    if constexpr (BWD) {
#pragma unroll
      for (int i = 0; i < 4; ++i) g[i] = to_f32(dy[row * D + c + i]);
      const float4 rv = *(const float4*)(res + row * D + c);
      r[0] = rv.x; r[1] = rv.y; r[2] = rv.z; r[3] = rv.w;
    }
  }
  float mean, rstd;
  if constexpr (!BWD) {
    const float s = wave_sum(v[0] + v[1] + v[2] + v[3]);
    mean = s / D;
    float q = 0.f;
    if (ok) {
#pragma unroll
      for (int i = 0; i < 4; ++i) q += (v[i] - mean) * (v[i] - mean);
    }
    q = wave_sum(q);
    rstd = rsqrtf(q / D + eps);
    if (lane == 0) {
      stat[row * 2] = mean;
      stat[row * 2 + 1] = rstd;
    }
    if (ok) {
#pragma unroll
      for (int i = 0; i < 4; ++i) out_bf[row * D + c + i] = from_f32<bf16_t>((v[i] - mean) * rstd);
    }
  } else {
    mean = stat[row * 2];
    rstd = stat[row * 2 + 1];
    float sg = 0.f, sgx = 0.f;
    if (ok) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float xh = (v[i] - mean) * rstd;
        sg += g[i];
        sgx += g[i] * xh;
      }
    }
    sg = wave_sum(sg) / D;
    sgx = wave_sum(sgx) / D;
    if (ok) {
      float o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float xh = (v[i] - mean) * rstd;
        o[i] = r[i] + rstd * (g[i] - sg - xh * sgx);
      }
      *(float4*)(out32 + row * D + c) = make_float4(o[0], o[1], o[2], o[3]);
      if (out_bf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) out_bf[row * D + c + i] = from_f32<bf16_t>(o[i]);
      }
    }
  }
}

// out32 = a + b (a fp32 or bf16, b bf16), out_bf16 = bf16(out32); either output may be absent.  8 elements per thread.
template <typename TA>
__global__ __launch_bounds__(256) void stream_add_kernel(const TA* a, const bf16_t* b, float* out32, bf16_t* out_bf, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    float v[8];
    if constexpr (sizeof(TA) == 4) {
      const float4 a0 = *(const float4*)(a + i * 8), a1 = *(const float4*)(a + i * 8 + 4);
      v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
    } else {
      const Vec16<bf16_t> av = vec_load<bf16_t>((const bf16_t*)a + i * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = av.get(e);
    }
    const Vec16<bf16_t> bv = vec_load<bf16_t>(b + i * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += bv.get(e);
    if (out32) {
      *(float4*)(out32 + i * 8) = make_float4(v[0], v[1], v[2], v[3]);
      *(float4*)(out32 + i * 8 + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
    if (out_bf) {
      Vec16<bf16_t> o = vec_zero<bf16_t>();
#pragma unroll
      for (int e = 0; e < 8; ++e) o.set(e, v[e]);
      vec_store<bf16_t>(out_bf + i * 8, o);
    }
  }
}

// ---- strided column copy: dst[m*ldd + doff + c] = src[m*lds + soff + c], c < C ------------------
template <typename T_>
__global__ __launch_bounds__(256) void copy_cols_kernel(const T_* src, int64_t lds_, int soff, T_* dst, int64_t ldd, int doff,
                                                        int64_t M, int C, int accumulate) {
  constexpr int VEC = ElemTraits<T_>::VEC;
  const int CV = C / VEC;
  const int64_t total = M * CV;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / CV;
    const int cv = (int)(i - m * CV);
    Vec16<T_> v = vec_load<T_>(src + m * lds_ + soff + cv * VEC);
    if (accumulate) {
      Vec16<T_> o = vec_load<T_>(dst + m * ldd + doff + cv * VEC);
#pragma unroll
      for (int e = 0; e < VEC; ++e) v.set(e, v.get(e) + o.get(e));
    }
    vec_store<T_>(dst + m * ldd + doff + cv * VEC, v);
  }
}

}  // namespace

extern "C" int pb_unfold_fold(const void* src, void* dst, int B, int D, int T, int H, int W, int pt, int ph, int pw,
                              int to_tokens, pb_dtype dtype, int device, void* stream) {
  PB_CHECK_ARG(src && dst, "pb_unfold_fold: null pointer");
  PB_CHECK_ARG(B > 0 && D > 0 && T > 0 && H > 0 && W > 0 && pt > 0 && ph > 0 && pw > 0, "pb_unfold_fold: bad sizes");
  PB_CHECK_ARG(dtype == PB_F32 || dtype == PB_BF16, "pb_unfold_fold: bad dtype");
  const int vec = dtype == PB_BF16 ? 8 : 4;
  PB_CHECK_ARG(D % vec == 0, "pb_unfold_fold: D=%d must be a multiple of %d", D, vec);
  PB_ENTER(device);
  UfArgs a;
  a.src = src; a.dst = dst; a.B = B; a.D = D; a.T = T; a.H = H; a.W = W; a.pt = pt; a.ph = ph; a.pw = pw;
  a.t = (T + pt - 1) / pt; a.h = (H + ph - 1) / ph; a.w = (W + pw - 1) / pw;
  a.to_tokens = to_tokens;
  a.CV = D / vec;
  a.CVB = a.CV < 32 ? a.CV : 32;
  a.P = 256 / a.CVB;
  a.rows = to_tokens ? (int64_t)B * pt * ph * pw * a.t * a.h * a.w : (int64_t)B * T * H * W;
  int64_t gx = pb_cdiv(a.rows, a.P);
  if (gx > 4096) gx = 4096;
  if (dtype == PB_BF16) hipLaunchKernelGGL(unfold_kernel<bf16_t>, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(unfold_kernel<float>, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, a);
  PB_LAUNCH_CHECK();
  return 0;
}

extern "C" int pb_layernorm(const void* x, const void* dy, void* out, float* stat, int64_t rows, int D, float eps,
                            int backward, pb_dtype dtype, int device, void* stream) {
  PB_CHECK_ARG(x && out && stat && (!backward || dy), "pb_layernorm: null pointer");
  PB_CHECK_ARG(rows >= 0 && D > 0 && D <= 256 && D % 4 == 0, "pb_layernorm: D=%d must be a multiple of 4, <= 256", D);
  PB_CHECK_ARG(dtype == PB_F32 || dtype == PB_BF16, "pb_layernorm: bad dtype");
  if (rows == 0) return 0;
  PB_ENTER(device);
  dim3 grid((unsigned)pb_cdiv(rows, 4));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == PB_BF16) {
    if (backward) hipLaunchKernelGGL((layernorm_kernel<bf16_t, true>), grid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)out, stat, rows, D, eps);
    else hipLaunchKernelGGL((layernorm_kernel<bf16_t, false>), grid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)out, stat, rows, D, eps);
  } else {
    if (backward) hipLaunchKernelGGL((layernorm_kernel<float, true>), grid, dim3(256), 0, st, (const float*)x, (const float*)dy, (float*)out, stat, rows, D, eps);
    else hipLaunchKernelGGL((layernorm_kernel<float, false>), grid, dim3(256), 0, st, (const float*)x, (const float*)dy, (float*)out, stat, rows, D, eps);
  }
  PB_LAUNCH_CHECK();
  return 0;
}

extern "C" int pb_copy_cols(const void* src, int64_t src_ld, int src_off, void* dst, int64_t dst_ld, int dst_off, int64_t M,
                            int C, int accumulate, pb_dtype dtype, int device, void* stream) {
  PB_CHECK_ARG(src && dst && M >= 0 && C > 0, "pb_copy_cols: bad args");
  PB_CHECK_ARG(dtype == PB_F32 || dtype == PB_BF16, "pb_copy_cols: bad dtype");
  const int vec = dtype == PB_BF16 ? 8 : 4;
  PB_CHECK_ARG(C % vec == 0 && src_off % vec == 0 && dst_off % vec == 0 && src_ld % vec == 0 && dst_ld % vec == 0,
               "pb_copy_cols: sizes/offsets must be multiples of %d elements", vec);
  if (M == 0) return 0;
  PB_ENTER(device);
  int64_t gx = pb_cdiv(M * (C / vec), 256);
  if (gx > 8192) gx = 8192;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == PB_BF16) hipLaunchKernelGGL(copy_cols_kernel<bf16_t>, dim3((unsigned)gx), dim3(256), 0, st, (const bf16_t*)src, src_ld, src_off, (bf16_t*)dst, dst_ld, dst_off, M, C, accumulate);
  else hipLaunchKernelGGL(copy_cols_kernel<float>, dim3((unsigned)gx), dim3(256), 0, st, (const float*)src, src_ld, src_off, (float*)dst, dst_ld, dst_off, M, C, accumulate);
  PB_LAUNCH_CHECK();
  return 0;
}

extern "C" int pb_layernorm_wide(const float* x, const void* dy_bf16, const float* res32, void* out_bf16, float* out32,
                                 float* stat, int64_t rows, int D, float eps, int backward, int device, void* stream) {
  PB_CHECK_ARG(x && stat, "pb_layernorm_wide: null pointer");
  PB_CHECK_ARG(backward ? (dy_bf16 && res32 && out32) : (out_bf16 != nullptr), "pb_layernorm_wide: missing operand");
  PB_CHECK_ARG(rows >= 0 && D > 0 && D <= 256 && D % 4 == 0, "pb_layernorm_wide: D=%d must be a multiple of 4, <= 256", D);
  if (rows == 0) return 0;
  PB_ENTER(device);
  dim3 grid((unsigned)pb_cdiv(rows, 4));
  hipStream_t st = (hipStream_t)stream;
  if (backward) hipLaunchKernelGGL(layernorm_wide_kernel<true>, grid, dim3(256), 0, st, x, (const bf16_t*)dy_bf16, res32, (bf16_t*)out_bf16, out32, stat, rows, D, eps);
  else hipLaunchKernelGGL(layernorm_wide_kernel<false>, grid, dim3(256), 0, st, x, (const bf16_t*)dy_bf16, res32, (bf16_t*)out_bf16, out32, stat, rows, D, eps);
  PB_LAUNCH_CHECK();
  return 0;
}

extern "C" int pb_stream_add(const void* a, pb_dtype a_dtype, const void* b_bf16, float* out32, void* out_bf16, int64_t n,
                             int device, void* stream) {
  PB_CHECK_ARG(a && b_bf16 && (out32 || out_bf16), "pb_stream_add: null pointer");
  PB_CHECK_ARG(a_dtype == PB_F32 || a_dtype == PB_BF16, "pb_stream_add: bad dtype");
  PB_CHECK_ARG(n >= 0 && n % 8 == 0, "pb_stream_add: n must be a multiple of 8");
  if (n == 0) return 0;
  PB_ENTER(device);
  int64_t gx = pb_cdiv(n / 8, 256);
  if (gx > 8192) gx = 8192;
  hipStream_t st = (hipStream_t)stream;
  if (a_dtype == PB_F32) hipLaunchKernelGGL(stream_add_kernel<float>, dim3((unsigned)gx), dim3(256), 0, st, (const float*)a, (const bf16_t*)b_bf16, out32, (bf16_t*)out_bf16, n / 8);
  else hipLaunchKernelGGL(stream_add_kernel<bf16_t>, dim3((unsigned)gx), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b_bf16, out32, (bf16_t*)out_bf16, n / 8);
  PB_LAUNCH_CHECK();
  return 0;
}
