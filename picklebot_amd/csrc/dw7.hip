// Depthwise (1,k,k) "same" convolution on 7x7 frames (the last MobileNet stage: 960 channels, 7x7x(11..19)
// frames per clip).  A whole frame of a channel pair lives in one thread's registers: 49 independent 4-byte
// loads in flight per lane (25 KB per wave, ~200 KB per CU - the strip/tile kernels have ~12 KB per
// workgroup and are latency-bound here at 0.8-1.2 TB/s), lanes run along the channel pairs so every load of
// a wave is one contiguous 256-byte run of the NDHWC pixel, all tap/border logic is resolved at compile time
// (fully unrolled), no LDS, packed fp32 FMAs.
//   dw7_fwd_kernel  : y = corr(x, w) (+ SE pool); with flip/prologue it is also backward-data
//   dw7_wgrad_kernel: dw += sum over frames of dy (x) x windows, accumulators persistent over the walk
#include "dw_internal.h"

namespace {

constexpr int FS = 7;            // frame side
constexpr int NPX = FS * FS;

struct D7Args {
  const void* src;     // [N, Tsrc, 7, 7, C]
  const void* dstc;    // wgrad: dy [N, Tdst, 7, 7, C]
  void* dst;           // fwd: [N, Tdst, 7, 7, C]
  const float* w;      // [C][K*K]
  float* dw;
  const float* g_scale;
  const float* g_shift;
  float* pool;
  int N, C, Tsrc, Tdst, t_mul, t_off, flip;
  int frames_per_wg;   // wgrad: consecutive dst frames walked by one workgroup
};

template <typename T, int K, bool FLIP>
__global__ __launch_bounds__(256) void dw7_fwd_kernel(D7Args a) {
  typedef typename Pair<T>::raw_t raw_t;
  constexpr int P = K / 2;
  const int cp = blockIdx.y * 256 + threadIdx.x;           // channel pair
  const int c0 = cp * 2;
  if (c0 >= a.C) return;
  const int z = blockIdx.x;
  const int n = z / a.Tdst, td = z - n * a.Tdst;
  const int ts = td * a.t_mul + a.t_off;
  raw_t* dst = (raw_t*)a.dst + ((int64_t)z * NPX * a.C + c0) / 2;
  const int64_t pstride = a.C / 2;                          // pixel stride in raw_t units
  // squeeze-excite partial sums [Tdst][N][C]: one slot per output frame, plain stores (exactly one writer each)
  float* pslot = a.pool ? a.pool + ((int64_t)td * a.N + n) * a.C + c0 : nullptr;
  if (ts < 0 || ts >= a.Tsrc) {                             // the reference's zero frames
    const raw_t zero = Pair<T>::down(f32x2{0.f, 0.f});
#pragma unroll
    for (int p = 0; p < NPX; ++p) dst[p * pstride] = zero;
    if (pslot) *reinterpret_cast<f32x2*>(pslot) = f32x2{0.f, 0.f};
    return;
  }
  const raw_t* src = (const raw_t*)a.src + (((int64_t)n * a.Tsrc + ts) * NPX * a.C + c0) / 2;
  raw_t xr[NPX];
#pragma unroll
  for (int p = 0; p < NPX; ++p) xr[p] = src[p * pstride];
  // taps of the two channels: w[c0][0..K*K), w[c0+1][0..K*K) are 2*K*K consecutive floats
  float wt[2 * K * K];
#pragma unroll
  for (int i = 0; i < 2 * K * K; ++i) wt[i] = a.w[(int64_t)c0 * (K * K) + i];
  f32x2 gs = {1.f, 1.f}, gh = {0.f, 0.f};
  if (a.g_scale) gs = *reinterpret_cast<const f32x2*>(a.g_scale + (int64_t)n * a.C + c0);
  if (a.g_shift) gh = *reinterpret_cast<const f32x2*>(a.g_shift + (int64_t)n * a.C + c0);
  f32x2 xf[NPX];
#pragma unroll
  for (int p = 0; p < NPX; ++p) xf[p] = Pair<T>::up(xr[p]) * gs + gh;
  f32x2 psum = {0.f, 0.f};
#pragma unroll
  for (int h = 0; h < FS; ++h) {
#pragma unroll
    for (int w = 0; w < FS; ++w) {
      f32x2 acc = {0.f, 0.f};
#pragma unroll
      for (int kh = 0; kh < K; ++kh) {
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
          const int hs = h + kh - P, ws = w + kw - P;
          if (hs >= 0 && hs < FS && ws >= 0 && ws < FS) {
            constexpr int dummy = 0; (void)dummy;
            const int tap = FLIP ? (K - 1 - kh) * K + (K - 1 - kw) : kh * K + kw;
            acc += xf[hs * FS + ws] * f32x2{wt[tap], wt[K * K + tap]};
          }
        }
      }
      const raw_t o = Pair<T>::down(acc);
      dst[(h * FS + w) * pstride] = o;
      if (a.pool) psum += Pair<T>::up(o);
    }
  }
  if (pslot) *reinterpret_cast<f32x2*>(pslot) = psum;
}

template <typename T, int K>
__global__ __launch_bounds__(256) void dw7_wgrad_kernel(D7Args a) {
  typedef typename Pair<T>::raw_t raw_t;
  constexpr int P = K / 2;
  const int cp = blockIdx.y * 256 + threadIdx.x;
  const bool active = cp * 2 < a.C;
  const int c0 = active ? cp * 2 : 0;                       // inactive lanes shadow channel 0 and flush nothing
  const int64_t pstride = a.C / 2;
  f32x2 acc[K * K];
#pragma unroll
  for (int i = 0; i < K * K; ++i) acc[i] = f32x2{0.f, 0.f};
  const int nframes = a.N * a.Tdst;
  const int z0 = blockIdx.x * a.frames_per_wg, z1 = min(nframes, z0 + a.frames_per_wg);
  for (int z = z0; z < z1; ++z) {
    const int n = z / a.Tdst, td = z - n * a.Tdst;
    const int ts = td * a.t_mul + a.t_off;
    if (ts < 0 || ts >= a.Tsrc) continue;                   // uniform: zero frames do not depend on x
    const raw_t* xs = (const raw_t*)a.src + (((int64_t)n * a.Tsrc + ts) * NPX * a.C + c0) / 2;
    const raw_t* gy = (const raw_t*)a.dstc + ((int64_t)z * NPX * a.C + c0) / 2;
    raw_t xr[NPX], gr[NPX];
#pragma unroll
    for (int p = 0; p < NPX; ++p) xr[p] = xs[p * pstride];
#pragma unroll
    for (int p = 0; p < NPX; ++p) gr[p] = gy[p * pstride];
    f32x2 gs = {1.f, 1.f}, gh = {0.f, 0.f};
    if (a.g_scale) gs = *reinterpret_cast<const f32x2*>(a.g_scale + (int64_t)n * a.C + c0);
    if (a.g_shift) gh = *reinterpret_cast<const f32x2*>(a.g_shift + (int64_t)n * a.C + c0);
    f32x2 xf[NPX];
#pragma unroll
    for (int p = 0; p < NPX; ++p) xf[p] = Pair<T>::up(xr[p]);
#pragma unroll
    for (int h = 0; h < FS; ++h) {
#pragma unroll
      for (int w = 0; w < FS; ++w) {
        const f32x2 g = Pair<T>::up(gr[h * FS + w]) * gs + gh;
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
#pragma unroll
          for (int kw = 0; kw < K; ++kw) {
            const int hs = h + kh - P, ws = w + kw - P;
            if (hs >= 0 && hs < FS && ws >= 0 && ws < FS) acc[kh * K + kw] += g * xf[hs * FS + ws];
          }
        }
      }
    }
  }
  // the workgroup's 512 channels x K*K taps are one contiguous block of its partial gradient (a.dw = workspace
  // [gridDim.x][C][K*K]): through LDS so that the stores are consecutive floats; no atomics, the slots are added up
  // in a fixed order afterwards
  __shared__ float red[512 * K * K];
#pragma unroll
  for (int i = 0; i < K * K; ++i) {
    if (!active) acc[i] = f32x2{0.f, 0.f};
    red[(threadIdx.x * 2) * (K * K) + i] = acc[i][0];
    red[(threadIdx.x * 2 + 1) * (K * K) + i] = acc[i][1];
  }
  __syncthreads();
  const int cbase = blockIdx.y * 512;
  const int nval = min(512, a.C - cbase) * (K * K);
  float* part = a.dw + (int64_t)blockIdx.x * a.C * (K * K) + (int64_t)cbase * (K * K);
  for (int i = threadIdx.x; i < nval; i += 256) part[i] = red[i];
}

bool d7_ok(const pb_dwconv_shape* s) {
  static const bool enabled = !(getenv("PB_DW7") && atoi(getenv("PB_DW7")) == 0);
  return enabled && s->kt == 1 && s->kh == s->kw && (s->kh == 3 || s->kh == 5) && s->sh == 1 && s->sw == 1 &&
         s->ph == s->kh / 2 && s->pw == s->kw / 2 && s->Hi == FS && s->Wi == FS && s->Ho == FS && s->Wo == FS &&
         s->C % 2 == 0 && (int64_t)s->N * s->To <= 0x7fffffff;
}

}  // namespace

bool pb_dw7_eligible(const pb_dwconv_shape* s) { return d7_ok(s); }

int pb_dw7_fwd(const void* src, const float* w, void* dst, const pb_dwconv_shape* s, pb_dtype dtype, int flip,
               const float* g_scale, const float* g_shift, float* pool, hipStream_t st) {
  if (!d7_ok(s)) return -1;
  D7Args a = {};
  a.src = src; a.w = w; a.dst = dst; a.g_scale = g_scale; a.g_shift = g_shift; a.pool = pool; a.flip = flip;
  a.N = s->N; a.C = s->C; a.Tsrc = s->Ti; a.Tdst = s->To; a.t_mul = s->st; a.t_off = -s->pt;
  const dim3 grid((unsigned)(s->N * s->To), (unsigned)pb_cdiv(s->C / 2, 256));
#define PB_D7(TT, KK)                                                                                      \
  do {                                                                                                     \
    if (flip) hipLaunchKernelGGL((dw7_fwd_kernel<TT, KK, true>), grid, dim3(256), 0, st, a);               \
    else hipLaunchKernelGGL((dw7_fwd_kernel<TT, KK, false>), grid, dim3(256), 0, st, a);                   \
  } while (0)
  if (dtype == PB_BF16) {
    if (s->kh == 3) PB_D7(bf16_t, 3); else PB_D7(bf16_t, 5);
  } else {
    if (s->kh == 3) PB_D7(float, 3); else PB_D7(float, 5);
  }
#undef PB_D7
  if (hipGetLastError() != hipSuccess) { pb_set_error("dw7 forward launch failed"); return 1; }
  return 0;
}

namespace {
// ~384 workgroups in all (98 loads in flight per lane make up for the low occupancy); every workgroup writes one
// partial gradient of its channels
int d7_wgrad_wgs(const pb_dwconv_shape* s, int* frames_per_wg) {
  const int chunks = (int)pb_cdiv(s->C / 2, 256);
  const int nframes = s->N * s->To;
  int wgs = (int)pb_cdiv(384, chunks);
  if (wgs > nframes) wgs = nframes;
  const int fpw = (int)pb_cdiv(nframes, wgs);
  if (frames_per_wg) *frames_per_wg = fpw;
  return (int)pb_cdiv(nframes, fpw);
}
}  // namespace

int64_t pb_dw7_wgrad_slots(const pb_dwconv_shape* s) { return d7_ok(s) ? d7_wgrad_wgs(s, nullptr) : 0; }

int pb_dw7_wgrad(const void* dy, const void* x, float* part, const pb_dwconv_shape* s, pb_dtype dtype,
                 const float* g_scale, const float* g_shift, hipStream_t st, int64_t* slots) {
  if (!d7_ok(s)) return -1;
  D7Args a = {};
  a.src = x; a.dstc = dy; a.dw = part; a.g_scale = g_scale; a.g_shift = g_shift;
  a.N = s->N; a.C = s->C; a.Tsrc = s->Ti; a.Tdst = s->To; a.t_mul = s->st; a.t_off = -s->pt;
  const int chunks = (int)pb_cdiv(s->C / 2, 256);
  const int wgs = d7_wgrad_wgs(s, &a.frames_per_wg);
  *slots = wgs;
  const dim3 grid((unsigned)wgs, (unsigned)chunks);
  if (dtype == PB_BF16) {
    if (s->kh == 3) hipLaunchKernelGGL((dw7_wgrad_kernel<bf16_t, 3>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((dw7_wgrad_kernel<bf16_t, 5>), grid, dim3(256), 0, st, a);
  } else {
    if (s->kh == 3) hipLaunchKernelGGL((dw7_wgrad_kernel<float, 3>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((dw7_wgrad_kernel<float, 5>), grid, dim3(256), 0, st, a);
  }
  if (hipGetLastError() != hipSuccess) { pb_set_error("dw7 weight-gradient launch failed"); return 1; }
  return 0;
}
