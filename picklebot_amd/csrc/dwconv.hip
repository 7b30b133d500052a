// Depthwise 3-D convolution over NDHWC activations (HBM-bound stencil; no MFMA on purpose).
//
// Fast path ("strip" kernels) for Bottleneck3D's (1,k,k) filters, k in {3,5}, spatial stride
// s in {1,2} (mobilenet.py:67-75).  A thread owns CH channels (16 B or 8 B per access) of one
// output column and TH=4 vertically adjacent outputs, keeps the k*k*CH filter taps of its channel chunk in
// and walks the (TH-1)*s+k input rows once: each row it loads feeds up to k/s outputs from
// LDS.  Lanes run along (channel-vector, then x), so every access of a wave is one
// contiguous span of the NDHWC row.  The temporal axis is only an index map (kernel extent 1):
// frame t_in = t_out*t_mul + t_off; output frames that map outside the clip are the reference's
// zero frames (stride/padding applied to T, SURVEY "temporal growth quirk") and are written as
// zeros without reading anything.
//   forward            : y = conv(x)                      [+ fused squeeze-excite pooling]
//   backward-data, s=1 : the same kernel on dy with the filter flipped
//   backward-weight    : same walk, taps accumulated in registers across a persistent loop
// Generic path: any (kt,kh,kw)/stride/padding (MoViNet blocks, CausalConv3d, s=2 backward-data).
#include "dw_internal.h"

namespace {

template <typename T, int CH>
__device__ __forceinline__ void load_ch(const T* p, float (&o)[CH]) {
  if constexpr (sizeof(T) * CH == 16) {
    Vec16<T> v = vec_load<T>(p);
#pragma unroll
    for (int i = 0; i < CH; ++i) o[i] = v.get(i);
  } else if constexpr (sizeof(T) * CH == 8) {   // 4 x bf16
    bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
    for (int i = 0; i < CH; ++i) o[i] = (float)v[i];
  } else {
#pragma unroll
    for (int i = 0; i < CH; ++i) o[i] = to_f32(p[i]);
  }
}
template <typename T, int CH>
__device__ __forceinline__ void store_ch(T* p, const float (&o)[CH]) {
  if constexpr (sizeof(T) * CH == 16) {
    Vec16<T> v;
#pragma unroll
    for (int i = 0; i < CH; ++i) v.set(i, o[i]);
    vec_store<T>(p, v);
  } else if constexpr (sizeof(T) * CH == 8) {
    bf16x4 v;
#pragma unroll
    for (int i = 0; i < CH; ++i) v[i] = (bf16_t)o[i];
    *reinterpret_cast<bf16x4*>(p) = v;
  } else {
#pragma unroll
    for (int i = 0; i < CH; ++i) p[i] = from_f32<T>(o[i]);
  }
}

// Sum of per-thread channel vectors over the row lanes p of the (cvl, p) thread layout, in a fixed order (no atomics);
// the CVB*CH channel sums of the workgroup are stored to out[cbase + ...].  red: CH rows of 256 floats.
template <int CH>
__device__ __forceinline__ void block_chan_sum(float* red, const float (&v)[CH], bool lane_ok, int CVB, int P, int cbase,
                                               int C, float* out) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < CH; ++i) red[i * 256 + tid] = lane_ok ? v[i] : 0.f;
  __syncthreads();
  for (int o = tid; o < CVB * CH; o += 256) {
    const int cvl = o / CH, i = o - cvl * CH;
    const float* src = red + i * 256 + cvl;
    float s0 = 0.f, s1 = 0.f;
    int p = 0;
    for (; p + 2 <= P; p += 2) {
      s0 += src[p * CVB];
      s1 += src[(p + 1) * CVB];
    }
    if (p < P) s0 += src[p * CVB];
    if (cbase + o < C) out[cbase + o] = s0 + s1;
  }
}

struct StripArgs {
  const void* src;     // [N, Tsrc, Hs, Ws, C]
  const float* w;      // [C][K*K]
  void* dst;           // [N, Tdst, Hd, Wd, C]
  const void* x;       // backward-weight: the forward input
  float* dw;           // backward-weight: partial gradients [gridDim.x][C][K*K] (workspace)
  int N, C, CV, CVB, P;
  int Tsrc, Hs, Ws;
  int Tdst, Hd, Wd;
  int pad;             // spatial padding seen from dst: hs = hd*S - pad + kh
  int t_mul, t_off;    // tsrc = tdst*t_mul + t_off
  int flip;
  const float* g_scale;   // optional [N][C]: src value = v*g_scale + g_shift (valid positions only)
  const float* g_shift;
  float* pool;            // optional squeeze-excite partial sums [Tdst*gridDim.x][N][C]: slot (frame, blockIdx.x)
  int HB;                 // ceil(Hd / TH)
  int64_t items;          // backward-weight: N*Tdst*ceil(HB*Wd/P)
  int xblocks;            // ceil(HB*Wd / P)
};

constexpr int TH = 4;
constexpr int GR = 3;          // input rows fetched per batch of loads

// CH elements of T as one register block (16 B / 8 B / 32 B load)
template <typename T, int CH> using RawVec = T __attribute__((ext_vector_type(CH)));

template <typename T, int CH>
__device__ __forceinline__ RawVec<T, CH> raw_load(const T* p) {
  return *reinterpret_cast<const RawVec<T, CH>*>(p);
}

// One batch of input rows [R0, R0+GR) of a strip: K clamped, unconditional loads per row, so the
// whole batch is in flight at once (no branch between a load and the next); validity is applied
// as a select when the values are consumed.
template <typename T, int K, int S, int CH>
struct RowBatch {
  RawVec<T, CH> v[GR][K];
  template <int R0>
  __device__ __forceinline__ void load(const T* fbase, int hs0, int ws0, int Hs, int Ws, int C) {
    constexpr int RH = (TH - 1) * S + K;
#pragma unroll
    for (int g = 0; g < GR; ++g) {
      if (R0 + g < RH) {
        const int hs = min(max(hs0 + R0 + g, 0), Hs - 1);
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
          const int ws = min(max(ws0 + kw, 0), Ws - 1);
          v[g][kw] = raw_load<T, CH>(fbase + ((int64_t)hs * Ws + ws) * C);
        }
      }
    }
  }
};

template <typename T, int K, int S, int CH>
__global__ __launch_bounds__(256) void dw_strip_fwd_kernel(StripArgs a) {
  __shared__ __attribute__((aligned(16))) float wl[K * K * 32 * CH];   // [tap][CVB*CH] filter taps of this channel chunk
  __shared__ float red[CH * 256];
  constexpr int RH = (TH - 1) * S + K;
  constexpr int NB = (RH + GR - 1) / GR;
  const int tid = threadIdx.x;
  const int cvl = tid % a.CVB, p = tid / a.CVB;
  const int cv = blockIdx.y * a.CVB + cvl;
  const bool lane_ok = p < a.P && cv < a.CV;
  const int z = blockIdx.z;
  const int n = z / a.Tdst, td = z - n * a.Tdst;
  const int ts = td * a.t_mul + a.t_off;
  const bool frame_ok = ts >= 0 && ts < a.Tsrc;
  const int c0 = lane_ok ? cv * CH : 0;
  const int CW = a.CVB * CH;                 // channels of this chunk
  const T* src = (const T*)a.src;
  T* dst = (T*)a.dst;
  if (frame_ok) {
    // the chunk's taps are contiguous in the [C][K*K] filter: coalesced read, transposed to LDS
    const int cbase = blockIdx.y * CW;
    for (int i = tid; i < CW * K * K; i += 256) {
      const int cl = i / (K * K), tap = i - cl * (K * K);
      const int kh = tap / K, kw = tap - kh * K;
      const int t2 = a.flip ? (K - 1 - kh) * K + (K - 1 - kw) : tap;
      wl[t2 * CW + cl] = (cbase + cl < a.C) ? a.w[(int64_t)(cbase + cl) * (K * K) + tap] : 0.f;
    }
  }
  __syncthreads();

  float gs[CH], gh[CH], psum[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) { gs[i] = 1.f; gh[i] = 0.f; psum[i] = 0.f; }
  if (a.g_scale) {
#pragma unroll
    for (int q = 0; q < CH / 4; ++q) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(a.g_scale + (int64_t)n * a.C + c0 + 4 * q);
      gs[4 * q] = t[0]; gs[4 * q + 1] = t[1]; gs[4 * q + 2] = t[2]; gs[4 * q + 3] = t[3];
    }
  }
  if (a.g_shift) {
#pragma unroll
    for (int q = 0; q < CH / 4; ++q) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(a.g_shift + (int64_t)n * a.C + c0 + 4 * q);
      gh[4 * q] = t[0]; gh[4 * q + 1] = t[1]; gh[4 * q + 2] = t[2]; gh[4 * q + 3] = t[3];
    }
  }
  const T* fbase = src + ((int64_t)n * a.Tsrc + (frame_ok ? ts : 0)) * a.Hs * a.Ws * a.C + c0;
  const int npix = a.HB * a.Wd;

  for (int xb = blockIdx.x; xb < a.xblocks; xb += gridDim.x) {
    const int pix = xb * a.P + p;
    const bool active = lane_ok && pix < npix;
    const int pixc = min(pix, npix - 1);                 // inactive lanes shadow a valid strip
    const int hb = pixc / a.Wd, wd = pixc - hb * a.Wd;
    const int hd0 = hb * TH;
    const int hs0 = hd0 * S - a.pad, ws0 = wd * S - a.pad;
    float acc[TH][CH];
#pragma unroll
    for (int j = 0; j < TH; ++j)
#pragma unroll
      for (int i = 0; i < CH; ++i) acc[j][i] = 0.f;

    if (frame_ok) {
      RowBatch<T, K, S, CH> bat[2];
      bat[0].template load<0>(fbase, hs0, ws0, a.Hs, a.Ws, a.C);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        // issue the next batch before consuming this one
        if (b + 1 < NB) {
          if (b == 0) bat[1].template load<GR>(fbase, hs0, ws0, a.Hs, a.Ws, a.C);
          if (b == 1) bat[0].template load<2 * GR>(fbase, hs0, ws0, a.Hs, a.Ws, a.C);
          if (b == 2) bat[1].template load<3 * GR>(fbase, hs0, ws0, a.Hs, a.Ws, a.C);
        }
#pragma unroll
        for (int g = 0; g < GR; ++g) {
          const int r = b * GR + g;
          if (r < RH) {
            const bool row_ok = (hs0 + r >= 0) && (hs0 + r < a.Hs);
            float in[K][CH];
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
              const bool ok = row_ok && (ws0 + kw >= 0) && (ws0 + kw < a.Ws);
#pragma unroll
              for (int i = 0; i < CH; ++i) in[kw][i] = ok ? (to_f32(bat[b & 1].v[g][kw][i]) * gs[i] + gh[i]) : 0.f;
            }
#pragma unroll
            for (int j = 0; j < TH; ++j) {
              const int kh = r - j * S;            // compile-time after unrolling
              if (kh >= 0 && kh < K) {
#pragma unroll
                for (int kw = 0; kw < K; ++kw) {
                  float wv[CH];
#pragma unroll
                  for (int q = 0; q < CH / 4; ++q) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(&wl[(kh * K + kw) * CW + cvl * CH + 4 * q]);
                    wv[4 * q] = t[0]; wv[4 * q + 1] = t[1]; wv[4 * q + 2] = t[2]; wv[4 * q + 3] = t[3];
                  }
#pragma unroll
                  for (int i = 0; i < CH; ++i) acc[j][i] += in[kw][i] * wv[i];
                }
              }
            }
          }
        }
      }
    }
    if (active) {
#pragma unroll
      for (int j = 0; j < TH; ++j) {
        const int hd = hd0 + j;
        if (hd < a.Hd) {
          T* o = dst + (((int64_t)z * a.Hd + hd) * a.Wd + wd) * a.C + c0;
          store_ch<T, CH>(o, acc[j]);
          if (a.pool) {
#pragma unroll
            for (int i = 0; i < CH; ++i) psum[i] += to_f32(from_f32<T>(acc[j][i]));
          }
        }
      }
    }
  }
  if (a.pool)      // every (frame, blockIdx.x) slot is written (zero frames contribute zeros)
    block_chan_sum<CH>(red, psum, lane_ok, a.CVB, a.P, blockIdx.y * CW, a.C,
                       a.pool + (((int64_t)td * gridDim.x + blockIdx.x) * a.N + n) * a.C);
}

// Weight gradient for the strip family: dst-shaped tensor = dy (gradient of the conv output),
// src-shaped tensor = x (conv input).  dw[c][kh][kw] += sum dy[hd][wd] * x[hd*S-pad+kh][wd*S-pad+kw].
template <typename T, int K, int S, int CH>
__global__ __launch_bounds__(256) void dw_strip_wgrad_kernel(StripArgs a) {
  __shared__ float red[K * K * 32 * CH];
  constexpr int RH = (TH - 1) * S + K;
  constexpr int NB = (RH + GR - 1) / GR;
  const int tid = threadIdx.x;
  const int cvl = tid % a.CVB, p = tid / a.CVB;
  const int cv = blockIdx.y * a.CVB + cvl;
  const bool lane_ok = p < a.P && cv < a.CV;
  const int c0 = lane_ok ? cv * CH : 0;
  const int CW = a.CVB * CH;
  const T* dy = (const T*)a.dst;
  const T* x = (const T*)a.x;
  for (int i = tid; i < K * K * CW; i += 256) red[i] = 0.f;
  __syncthreads();

  float acc[K][K][CH];
#pragma unroll
  for (int kh = 0; kh < K; ++kh)
#pragma unroll
    for (int kw = 0; kw < K; ++kw)
#pragma unroll
      for (int i = 0; i < CH; ++i) acc[kh][kw][i] = 0.f;
  const int npix = a.HB * a.Wd;

  for (int64_t it = blockIdx.x; it < a.items; it += gridDim.x) {
    const int z = (int)(it / a.xblocks);
    const int xb = (int)(it - (int64_t)z * a.xblocks);
    const int n = z / a.Tdst, td = z - n * a.Tdst;
    const int ts = td * a.t_mul + a.t_off;
    if (ts < 0 || ts >= a.Tsrc) continue;        // zero frame: no dependence on x (uniform branch)
    const int pix = xb * a.P + p;
    const bool active = lane_ok && pix < npix;
    const int pixc = min(pix, npix - 1);
    const int hb = pixc / a.Wd, wd = pixc - hb * a.Wd;
    const int hd0 = hb * TH;
    const int hs0 = hd0 * S - a.pad, ws0 = wd * S - a.pad;
    const T* fbase = x + ((int64_t)n * a.Tsrc + ts) * a.Hs * a.Ws * a.C + c0;
    RowBatch<T, K, S, CH> bat[2];
    bat[0].template load<0>(fbase, hs0, ws0, a.Hs, a.Ws, a.C);
    RawVec<T, CH> graw[TH];
#pragma unroll
    for (int j = 0; j < TH; ++j) {
      const int hd = min(hd0 + j, a.Hd - 1);
      graw[j] = raw_load<T, CH>(dy + (((int64_t)z * a.Hd + hd) * a.Wd + wd) * a.C + c0);
    }
    // squeeze-excite backward rides on the gradient load: g = dy*g_scale[n,c] + g_shift[n,c]
    // (one vector read per item - never per element, that would serialise 2*TH*CH L2 round trips)
    float gs[CH], gh[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) { gs[i] = 1.f; gh[i] = 0.f; }
    if (a.g_scale) {
#pragma unroll
      for (int q = 0; q < CH / 4; ++q) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(a.g_scale + (int64_t)n * a.C + c0 + 4 * q);
        gs[4 * q] = t[0]; gs[4 * q + 1] = t[1]; gs[4 * q + 2] = t[2]; gs[4 * q + 3] = t[3];
      }
    }
    if (a.g_shift) {
#pragma unroll
      for (int q = 0; q < CH / 4; ++q) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(a.g_shift + (int64_t)n * a.C + c0 + 4 * q);
        gh[4 * q] = t[0]; gh[4 * q + 1] = t[1]; gh[4 * q + 2] = t[2]; gh[4 * q + 3] = t[3];
      }
    }
    float g[TH][CH];
#pragma unroll
    for (int j = 0; j < TH; ++j) {
      const bool ok = active && (hd0 + j < a.Hd);
#pragma unroll
      for (int i = 0; i < CH; ++i) g[j][i] = ok ? (to_f32(graw[j][i]) * gs[i] + gh[i]) : 0.f;
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (b + 1 < NB) {
        if (b == 0) bat[1].template load<GR>(fbase, hs0, ws0, a.Hs, a.Ws, a.C);
        if (b == 1) bat[0].template load<2 * GR>(fbase, hs0, ws0, a.Hs, a.Ws, a.C);
        if (b == 2) bat[1].template load<3 * GR>(fbase, hs0, ws0, a.Hs, a.Ws, a.C);
      }
#pragma unroll
      for (int gi = 0; gi < GR; ++gi) {
        const int r = b * GR + gi;
        if (r < RH) {
          const bool row_ok = (hs0 + r >= 0) && (hs0 + r < a.Hs);
          float in[K][CH];
#pragma unroll
          for (int kw = 0; kw < K; ++kw) {
            const bool ok = row_ok && (ws0 + kw >= 0) && (ws0 + kw < a.Ws);
#pragma unroll
            for (int i = 0; i < CH; ++i) in[kw][i] = ok ? to_f32(bat[b & 1].v[gi][kw][i]) : 0.f;
          }
#pragma unroll
          for (int j = 0; j < TH; ++j) {
            const int kh = r - j * S;
            if (kh >= 0 && kh < K) {
#pragma unroll
              for (int kw = 0; kw < K; ++kw)
#pragma unroll
                for (int i = 0; i < CH; ++i) acc[kh][kw][i] += g[j][i] * in[kw][i];
            }
          }
        }
      }
    }
  }
  // threads that share a channel vector differ in the row lane p: they add their taps in P rounds (fixed order)
  for (int r = 0; r < a.P; ++r) {
    if (lane_ok && p == r) {
#pragma unroll
      for (int kh = 0; kh < K; ++kh)
#pragma unroll
        for (int kw = 0; kw < K; ++kw)
#pragma unroll
          for (int i = 0; i < CH; ++i) red[(kh * K + kw) * CW + cvl * CH + i] += acc[kh][kw][i];
    }
    __syncthreads();
  }
  // the chunk's [channel][tap] block is contiguous in this workgroup's partial gradient
  const int cbase = blockIdx.y * CW;
  float* part = a.dw + (int64_t)blockIdx.x * a.C * (K * K) + (int64_t)cbase * (K * K);
  for (int i = tid; i < K * K * CW; i += 256) {
    const int cl = i / (K * K), tap = i - cl * (K * K);
    if (cbase + cl < a.C) part[i] = red[tap * CW + cl];
  }
}

// Backward-data for spatial stride 2 (the four stride-2 bottlenecks): a thread owns CH channels of
// a 2x2 block of dx.  Output parity fixes which taps contribute (compile-time after unrolling),
// and the whole block reads one NRxNR patch of dy (2x2 for k=3, 3x3 for k=5): 1-2.25 loads per
// output instead of k*k masked ones.  dx frames that no output frame reads (t+pt not a multiple of
// the temporal stride) are written as zeros.
// KT > 1: (kt,3,3) kernels with temporal stride 1 (MoViNet's stride-2 blocks, movinet.py:98-137): the same 2x2 block sums
// the patches of the KT output frames td + pt - dt that reach it, each against its own temporal slice of the filter.
template <typename T, int K, int CH, int KT = 1>
__global__ __launch_bounds__(256) void dw_strip_bwd_s2_kernel(StripArgs a) {
  __shared__ __attribute__((aligned(16))) float wl[KT * K * K * 32 * CH];
  constexpr int P = K / 2;
  constexpr int LO = (P - (K - 1)) / 2;       // ceil of a non-positive half (C++ truncates toward zero): 0 for k=3, -1 for k=5
  constexpr int HI = (1 + P) / 2;
  constexpr int NR = HI - LO + 1;
  const int tid = threadIdx.x;
  const int cvl = tid % a.CVB, p = tid / a.CVB;
  const int cv = blockIdx.y * a.CVB + cvl;
  const bool lane_ok = p < a.P && cv < a.CV;
  const int c0 = lane_ok ? cv * CH : 0;
  const int CW = a.CVB * CH;
  const T* dy = (const T*)a.src;      // [N, Tsrc, Hs, Ws, C]  (gradient of the conv output)
  T* dx = (T*)a.dst;                  // [N, Tdst, Hd, Wd, C]  (gradient of the conv input)
  {
    const int cbase = blockIdx.y * CW;
    for (int i = tid; i < CW * KT * K * K; i += 256) {
      const int cl = i / (KT * K * K), tap = i - cl * (KT * K * K);
      wl[tap * CW + cl] = (cbase + cl < a.C) ? a.w[(int64_t)(cbase + cl) * (KT * K * K) + tap] : 0.f;
    }
  }
  __syncthreads();
  const int HA = (a.Hd + 1) / 2, WB = (a.Wd + 1) / 2;
  const int nblk = HA * WB;
  for (int64_t it = blockIdx.x; it < a.items; it += gridDim.x) {
    const int z = (int)(it / a.xblocks);
    const int xb = (int)(it - (int64_t)z * a.xblocks);
    const int n = z / a.Tdst, td = z - n * a.Tdst;
    const int blk = xb * a.P + p;
    const bool active = lane_ok && blk < nblk;
    const int blkc = min(blk, nblk - 1);
    const int ba = blkc / WB, bb = blkc - ba * WB;
    float gs[CH], gh[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) { gs[i] = 1.f; gh[i] = 0.f; }
    if (a.g_scale) {
#pragma unroll
      for (int q = 0; q < CH / 4; ++q) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(a.g_scale + (int64_t)n * a.C + c0 + 4 * q);
        gs[4 * q] = t[0]; gs[4 * q + 1] = t[1]; gs[4 * q + 2] = t[2]; gs[4 * q + 3] = t[3];
      }
    }
    if (a.g_shift) {
#pragma unroll
      for (int q = 0; q < CH / 4; ++q) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(a.g_shift + (int64_t)n * a.C + c0 + 4 * q);
        gh[4 * q] = t[0]; gh[4 * q + 1] = t[1]; gh[4 * q + 2] = t[2]; gh[4 * q + 3] = t[3];
      }
    }
    float out[2][2][CH];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int i = 0; i < CH; ++i) out[u][v][i] = 0.f;
#pragma unroll
    for (int dt = 0; dt < KT; ++dt) {
      // temporal map of the forward conv: t_in = t_out*t_mul - t_off + dt  =>  t_out = (t_in + t_off - dt)/t_mul
      // (KT > 1 is only dispatched with t_mul = 1)
      const int tn = td + a.t_off - dt;
      const bool frame_ok = tn >= 0 && (tn % a.t_mul) == 0 && (tn / a.t_mul) < a.Tsrc;
      if (!frame_ok) continue;
      const int ts = tn / a.t_mul;
      const T* fbase = dy + ((int64_t)n * a.Tsrc + ts) * a.Hs * a.Ws * a.C + c0;
      RawVec<T, CH> raw[NR][NR];
#pragma unroll
      for (int dr = 0; dr < NR; ++dr) {
        const int ho = min(max(ba + LO + dr, 0), a.Hs - 1);
#pragma unroll
        for (int dc = 0; dc < NR; ++dc) {
          const int wo = min(max(bb + LO + dc, 0), a.Ws - 1);
          raw[dr][dc] = raw_load<T, CH>(fbase + ((int64_t)ho * a.Ws + wo) * a.C);
        }
      }
      float pt[NR][NR][CH];
#pragma unroll
      for (int dr = 0; dr < NR; ++dr)
#pragma unroll
        for (int dc = 0; dc < NR; ++dc) {
          const bool ok = (ba + LO + dr >= 0) && (ba + LO + dr < a.Hs) && (bb + LO + dc >= 0) && (bb + LO + dc < a.Ws);
#pragma unroll
          for (int i = 0; i < CH; ++i) pt[dr][dc][i] = ok ? (to_f32(raw[dr][dc][i]) * gs[i] + gh[i]) : 0.f;
        }
#pragma unroll
      for (int ph = 0; ph < 2; ++ph)
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
          if (((ph + P + kh) & 1) != 0) continue;
          const int dr = (ph + P - kh) / 2 - LO;          // exact: numerator is even
#pragma unroll
          for (int pw = 0; pw < 2; ++pw)
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
              if (((pw + P + kw) & 1) != 0) continue;
              const int dc = (pw + P - kw) / 2 - LO;
              float wv[CH];
#pragma unroll
              for (int q = 0; q < CH / 4; ++q) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(&wl[((dt * K + kh) * K + kw) * CW + cvl * CH + 4 * q]);
                wv[4 * q] = t[0]; wv[4 * q + 1] = t[1]; wv[4 * q + 2] = t[2]; wv[4 * q + 3] = t[3];
              }
#pragma unroll
              for (int i = 0; i < CH; ++i) out[ph][pw][i] += pt[dr][dc][i] * wv[i];
            }
        }
    }
    if (active) {
#pragma unroll
      for (int ph = 0; ph < 2; ++ph)
#pragma unroll
        for (int pw = 0; pw < 2; ++pw) {
          const int hi = 2 * ba + ph, wi = 2 * bb + pw;
          if (hi < a.Hd && wi < a.Wd)
            store_ch<T, CH>(dx + (((int64_t)z * a.Hd + hi) * a.Wd + wi) * a.C + c0, out[ph][pw]);
        }
    }
  }
}

// ----------------------------------------------------------------------------- generic path
struct GenArgs {
  const void* src;
  const float* w;
  void* dst;
  const void* x;
  float* dw;
  pb_dwconv_shape s;
  const float* g_scale;
  const float* g_shift;
  float* pool;
  int CV;               // C / CH
  int64_t total;        // threads of work
};

// forward: one thread = CH channels of one output voxel
template <typename T, int CH>
__global__ __launch_bounds__(256) void dw_gen_fwd_kernel(GenArgs a) {
  const pb_dwconv_shape& s = a.s;
  const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (f >= a.total) return;
  const int cv = (int)(f % a.CV);
  int64_t pix = f / a.CV;
  const int wo = (int)(pix % s.Wo); pix /= s.Wo;
  const int ho = (int)(pix % s.Ho); pix /= s.Ho;
  const int to = (int)(pix % s.To);
  const int n = (int)(pix / s.To);
  const int c0 = cv * CH;
  const int taps = s.kt * s.kh * s.kw;
  const T* x = (const T*)a.src;
  float acc[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) acc[i] = 0.f;
  for (int at = 0; at < s.kt; ++at) {
    const int ti = to * s.st - s.pt + at;
    if (ti < 0 || ti >= s.Ti) continue;
    for (int ah = 0; ah < s.kh; ++ah) {
      const int hi = ho * s.sh - s.ph + ah;
      if (hi < 0 || hi >= s.Hi) continue;
      for (int aw = 0; aw < s.kw; ++aw) {
        const int wi = wo * s.sw - s.pw + aw;
        if (wi < 0 || wi >= s.Wi) continue;
        float v[CH];
        load_ch<T, CH>(x + ((((int64_t)n * s.Ti + ti) * s.Hi + hi) * s.Wi + wi) * s.C + c0, v);
        const int tap = (at * s.kh + ah) * s.kw + aw;
#pragma unroll
        for (int i = 0; i < CH; ++i) acc[i] += v[i] * a.w[(int64_t)(c0 + i) * taps + tap];
      }
    }
  }
  store_ch<T, CH>((T*)a.dst + ((((int64_t)n * s.To + to) * s.Ho + ho) * s.Wo + wo) * s.C + c0, acc);
}

// pooled variant: block = (channel-vector lanes) x (voxel lanes) over ONE clip (blockIdx.z = n), so the
// squeeze-excite sums reduce in LDS (fixed order) into the workgroup's slot: pool is [gridDim.x][N][C]
template <typename T, int CH>
__global__ __launch_bounds__(256) void dw_gen_fwd_pool_kernel(GenArgs a, int CVB, int P, int cchunks) {
  __shared__ float red[CH * 256];
  const pb_dwconv_shape& s = a.s;
  const int tid = threadIdx.x;
  const int cvl = tid % CVB, p = tid / CVB;
  const int cv = blockIdx.y * CVB + cvl;
  const bool lane_ok = p < P && cv < a.CV;
  const int c0 = lane_ok ? cv * CH : 0;
  const int n = blockIdx.z;
  const int taps = s.kt * s.kh * s.kw;
  const T* x = (const T*)a.src;
  float psum[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) psum[i] = 0.f;
  const int64_t nvox = (int64_t)s.To * s.Ho * s.Wo;
  if (lane_ok) {
    for (int64_t v0 = (int64_t)blockIdx.x * P + p; v0 < nvox; v0 += (int64_t)gridDim.x * P) {
      int64_t pix = v0;
      const int wo = (int)(pix % s.Wo); pix /= s.Wo;
      const int ho = (int)(pix % s.Ho);
      const int to = (int)(pix / s.Ho);
      float acc[CH];
#pragma unroll
      for (int i = 0; i < CH; ++i) acc[i] = 0.f;
      for (int at = 0; at < s.kt; ++at) {
        const int ti = to * s.st - s.pt + at;
        if (ti < 0 || ti >= s.Ti) continue;
        for (int ah = 0; ah < s.kh; ++ah) {
          const int hi = ho * s.sh - s.ph + ah;
          if (hi < 0 || hi >= s.Hi) continue;
          for (int aw = 0; aw < s.kw; ++aw) {
            const int wi = wo * s.sw - s.pw + aw;
            if (wi < 0 || wi >= s.Wi) continue;
            float v[CH];
            load_ch<T, CH>(x + ((((int64_t)n * s.Ti + ti) * s.Hi + hi) * s.Wi + wi) * s.C + c0, v);
            const int tap = (at * s.kh + ah) * s.kw + aw;
#pragma unroll
            for (int i = 0; i < CH; ++i) acc[i] += v[i] * a.w[(int64_t)(c0 + i) * taps + tap];
          }
        }
      }
      store_ch<T, CH>((T*)a.dst + ((int64_t)n * nvox + v0) * s.C + c0, acc);
#pragma unroll
      for (int i = 0; i < CH; ++i) psum[i] += to_f32(from_f32<T>(acc[i]));
    }
  }
  block_chan_sum<CH>(red, psum, lane_ok, CVB, P, blockIdx.y * CVB * CH, s.C,
                     a.pool + ((int64_t)blockIdx.x * gridDim.z + n) * s.C);
}

// backward-data: one thread = CH channels of one INPUT voxel
template <typename T, int CH>
__global__ __launch_bounds__(256) void dw_gen_bwd_data_kernel(GenArgs a) {
  const pb_dwconv_shape& s = a.s;
  const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (f >= a.total) return;
  const int cv = (int)(f % a.CV);
  int64_t pix = f / a.CV;
  const int wi = (int)(pix % s.Wi); pix /= s.Wi;
  const int hi = (int)(pix % s.Hi); pix /= s.Hi;
  const int ti = (int)(pix % s.Ti);
  const int n = (int)(pix / s.Ti);
  const int c0 = cv * CH;
  const int taps = s.kt * s.kh * s.kw;
  const T* dy = (const T*)a.src;
  float gs[CH], gh[CH], acc[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    gs[i] = a.g_scale ? a.g_scale[(int64_t)n * s.C + c0 + i] : 1.f;
    gh[i] = a.g_shift ? a.g_shift[(int64_t)n * s.C + c0 + i] : 0.f;
    acc[i] = 0.f;
  }
  for (int at = 0; at < s.kt; ++at) {
    const int tn = ti + s.pt - at;
    if (tn < 0 || tn % s.st) continue;
    const int to = tn / s.st;
    if (to >= s.To) continue;
    for (int ah = 0; ah < s.kh; ++ah) {
      const int hn = hi + s.ph - ah;
      if (hn < 0 || hn % s.sh) continue;
      const int ho = hn / s.sh;
      if (ho >= s.Ho) continue;
      for (int aw = 0; aw < s.kw; ++aw) {
        const int wn = wi + s.pw - aw;
        if (wn < 0 || wn % s.sw) continue;
        const int wo = wn / s.sw;
        if (wo >= s.Wo) continue;
        float v[CH];
        load_ch<T, CH>(dy + ((((int64_t)n * s.To + to) * s.Ho + ho) * s.Wo + wo) * s.C + c0, v);
        const int tap = (at * s.kh + ah) * s.kw + aw;
#pragma unroll
        for (int i = 0; i < CH; ++i) acc[i] += (v[i] * gs[i] + gh[i]) * a.w[(int64_t)(c0 + i) * taps + tap];
      }
    }
  }
  store_ch<T, CH>((T*)a.dst + ((((int64_t)n * s.Ti + ti) * s.Hi + hi) * s.Wi + wi) * s.C + c0, acc);
}

// backward-weight: blockIdx.y = tap; threads (channel-vector, voxel lane) grid-stride over output
// voxels, then a block reduction per channel.
template <typename T, int CH>
__global__ __launch_bounds__(256) void dw_gen_wgrad_kernel(GenArgs a, int CVB, int P, int cchunks) {
  __shared__ float red[CH * 256 + 32 * CH];
  const pb_dwconv_shape& s = a.s;
  const int tid = threadIdx.x;
  const int cvl = tid % CVB, p = tid / CVB;
  const int tap = blockIdx.y / cchunks, chunk = blockIdx.y - tap * cchunks;
  const int cv = chunk * CVB + cvl;
  const bool lane_ok = p < P && cv < a.CV;
  const int c0 = cv * CH;
  const int aw = tap % s.kw, ah = (tap / s.kw) % s.kh, at = tap / (s.kw * s.kh);
  const int taps = s.kt * s.kh * s.kw;
  const T* dy = (const T*)a.src;
  const T* x = (const T*)a.x;
  float acc[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) acc[i] = 0.f;
  const int64_t nvox = (int64_t)s.N * s.To * s.Ho * s.Wo;
  if (lane_ok) {
    for (int64_t v0 = (int64_t)blockIdx.x * P + p; v0 < nvox; v0 += (int64_t)gridDim.x * P) {
      int64_t pix = v0;
      const int wo = (int)(pix % s.Wo); pix /= s.Wo;
      const int ho = (int)(pix % s.Ho); pix /= s.Ho;
      const int to = (int)(pix % s.To);
      const int n = (int)(pix / s.To);
      const int ti = to * s.st - s.pt + at, hi = ho * s.sh - s.ph + ah, wi = wo * s.sw - s.pw + aw;
      if (ti < 0 || ti >= s.Ti || hi < 0 || hi >= s.Hi || wi < 0 || wi >= s.Wi) continue;
      float g[CH], xv[CH];
      load_ch<T, CH>(dy + v0 * s.C + c0, g);
      load_ch<T, CH>(x + ((((int64_t)n * s.Ti + ti) * s.Hi + hi) * s.Wi + wi) * s.C + c0, xv);
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const float gs = a.g_scale ? a.g_scale[(int64_t)n * s.C + c0 + i] : 1.f;
        const float gh = a.g_shift ? a.g_shift[(int64_t)n * s.C + c0 + i] : 0.f;
        acc[i] += (g[i] * gs + gh) * xv[i];
      }
    }
  }
  // a.dw: partial gradients [gridDim.x][C][taps]; this workgroup's (tap, channel chunk) entries of its slot
  block_chan_sum<CH>(red, acc, lane_ok, CVB, P, 0, CVB * CH, red + CH * 256);
  __syncthreads();
  if (tid < CVB * CH) {
    const int c = chunk * CVB * CH + tid;
    if (c < s.C) a.dw[((int64_t)blockIdx.x * s.C + c) * taps + tap] = red[CH * 256 + tid];
  }
}

// channel-vector lanes per workgroup: the largest divisor of CV that is <= 32 (all lanes busy);
// a prime CV > 32 falls back to 16 with the tail chunk masked.
int pick_cvb(int CV) {
  for (int d = CV < 32 ? CV : 32; d >= 1; --d)
    if (CV % d == 0) return d >= 4 || CV <= 32 ? d : 16;
  return 1;
}

// (kt,3,3), spatial stride 2, temporal stride 1, padding 1 in space: backward-data on the 2x2-block kernel with KT frames
bool bwd_is_kt_s2(const pb_dwconv_shape& s) {
  return (s.kt == 3 || s.kt == 5) && s.kh == 3 && s.kw == 3 && s.st == 1 && s.sh == 2 && s.sw == 2 && s.ph == 1 && s.pw == 1 &&
         s.C % 4 == 0;
}

bool strip_ok(const pb_dwconv_shape& s) {
  return s.kt == 1 && s.kh == s.kw && (s.kh == 3 || s.kh == 5) && s.sh == s.sw && (s.sh == 1 || s.sh == 2) &&
         s.ph == s.pw && s.C % 4 == 0;
}

// strip geometry for CH channels per thread; returns the forward grid's x extent
int64_t strip_fwd_geom(StripArgs& a, int CH) {
  a.CV = a.C / CH;
  a.CVB = pick_cvb(a.CV);
  a.P = 256 / a.CVB;
  a.HB = (int)pb_cdiv(a.Hd, TH);
  a.xblocks = (int)pb_cdiv((int64_t)a.HB * a.Wd, a.P);
  // a workgroup loads its filter chunk once and then walks several pixel blocks of its frame;
  // keep >= ~8 workgroups per CU in the whole launch
  const int64_t chunks = pb_cdiv(a.CV, a.CVB), frames = (int64_t)a.N * a.Tdst;
  int64_t gx = pb_cdiv(2048, chunks * frames);
  if (gx < pb_cdiv(a.xblocks, 8)) gx = pb_cdiv(a.xblocks, 8);
  if (gx > a.xblocks) gx = a.xblocks;
  return gx;
}
template <typename T, int K, int S, int CH>
void launch_strip_fwd(StripArgs a, hipStream_t st) {
  const int64_t gx = strip_fwd_geom(a, CH);
  const int64_t chunks = pb_cdiv(a.CV, a.CVB), frames = (int64_t)a.N * a.Tdst;
  dim3 grid((unsigned)gx, (unsigned)chunks, (unsigned)frames);
  hipLaunchKernelGGL((dw_strip_fwd_kernel<T, K, S, CH>), grid, dim3(256), 0, st, a);
}
// returns the weight-gradient grid's x extent = the number of partial gradients the launch writes
int64_t strip_wgrad_geom(StripArgs& a, int CH) {
  a.CV = a.C / CH;
  a.CVB = pick_cvb(a.CV);
  a.P = 256 / a.CVB;
  a.HB = (int)pb_cdiv(a.Hd, TH);
  a.xblocks = (int)pb_cdiv((int64_t)a.HB * a.Wd, a.P);
  a.items = (int64_t)a.N * a.Tdst * a.xblocks;
  const int chunks = (int)pb_cdiv(a.CV, a.CVB);
  // persistent grid of ~1024 workgroups, each writing one partial gradient of its channel chunk
  int64_t gx = pb_cdiv(1024, chunks);
  // every workgroup pays a fixed flush (k*k*CW LDS + the partial's stores): give it >= 8 items
  if (gx > pb_cdiv(a.items, 8)) gx = pb_cdiv(a.items, 8);
  if (gx < 1) gx = 1;
  return gx;
}
template <typename T, int K, int S, int CH>
void launch_strip_wgrad(StripArgs a, hipStream_t st) {
  const int64_t gx = strip_wgrad_geom(a, CH);
  dim3 grid((unsigned)gx, (unsigned)pb_cdiv(a.CV, a.CVB));
  hipLaunchKernelGGL((dw_strip_wgrad_kernel<T, K, S, CH>), grid, dim3(256), 0, st, a);
}

// channels per thread of the strip kernels (see dispatch_strip)
int strip_ch(bool bf16, int K, int C) { return (bf16 && K == 3 && C % 8 == 0) ? 8 : 4; }

// bf16: 8 channels/thread for k=3 (16 B), 4 for k=5 (taps live in registers: 25*4 = 100 VGPRs);
// f32: 4 channels/thread (16 B).
template <typename T, bool WGRAD>
int dispatch_strip(const StripArgs& a, int K, int S, hipStream_t st) {
  constexpr bool BF = sizeof(T) == 2;
#define PB_STRIP(KK, SS, CC)                                                          \
  do {                                                                                \
    if (WGRAD) launch_strip_wgrad<T, KK, SS, CC>(a, st);                              \
    else launch_strip_fwd<T, KK, SS, CC>(a, st);                                      \
    return 0;                                                                         \
  } while (0)
  if (K == 3 && S == 1) { if (BF && a.C % 8 == 0) PB_STRIP(3, 1, 8); else PB_STRIP(3, 1, 4); }
  if (K == 3 && S == 2) { if (BF && a.C % 8 == 0) PB_STRIP(3, 2, 8); else PB_STRIP(3, 2, 4); }
  if (K == 5 && S == 1) PB_STRIP(5, 1, 4);
  if (K == 5 && S == 2) PB_STRIP(5, 2, 4);
#undef PB_STRIP
  pb_set_error("dwconv strip: unsupported k=%d s=%d", K, S);
  return 1;
}

int check_shape(const pb_dwconv_shape* s, const char* who) {
  PB_CHECK_ARG(s, "%s: null shape", who);
  PB_CHECK_ARG((int64_t)s->N * s->To <= 65535 && (int64_t)s->N * s->Ti <= 65535, "%s: N*T exceeds the grid limit", who);
  PB_CHECK_ARG(s->N > 0 && s->C > 0 && s->Ti > 0 && s->Hi > 0 && s->Wi > 0 && s->To > 0 && s->Ho > 0 && s->Wo > 0,
               "%s: non-positive dimension", who);
  PB_CHECK_ARG(s->kt > 0 && s->kh > 0 && s->kw > 0 && s->st > 0 && s->sh > 0 && s->sw > 0 && s->pt >= 0 &&
               s->ph >= 0 && s->pw >= 0, "%s: bad kernel/stride/padding", who);
  // the last output's first tap must start inside the input padded by at most `front` at the back
  // (the reference's (1,k,k) filters with padding on T produce whole zero frames there)
  PB_CHECK_ARG((s->To - 1) * s->st - s->pt < s->Ti + s->pt && (s->Ho - 1) * s->sh - s->ph < s->Hi + s->ph &&
               (s->Wo - 1) * s->sw - s->pw < s->Wi + s->pw, "%s: output extent inconsistent with input/stride/padding", who);
  return 0;
}

// stride-1 backward-data as a forward-type correlation of dy with the flipped filter (padding k-1-p)
bool bwd_is_s1_corr(const pb_dwconv_shape& s) {
  return strip_ok(s) && s.sh == 1 && s.st == 1 && s.Ho == s.Hi + 2 * s.ph - s.kh + 1 && s.Wo == s.Wi + 2 * s.pw - s.kw + 1;
}
pb_dwconv_shape bwd_corr_shape(const pb_dwconv_shape& s) {
  pb_dwconv_shape c = s;
  c.Ti = s.To; c.Hi = s.Ho; c.Wi = s.Wo;
  c.To = s.Ti; c.Ho = s.Hi; c.Wo = s.Wi;
  c.ph = c.pw = s.kh - 1 - s.ph;
  c.st = 1; c.pt = -s.pt;            // ts = td + pt
  return c;
}

// (kt,3,3) kernels at stride 1: backward-data = correlation of dy with the filter flipped in all three dimensions
bool bwd_is_dw3_corr(const pb_dwconv_shape& s) {
  return pb_dw3_eligible(&s) && s.sh == 1 && s.sw == 1 && s.st == 1;
}
pb_dwconv_shape bwd_dw3_shape(const pb_dwconv_shape& s) {
  pb_dwconv_shape c = s;
  c.Ti = s.To; c.Hi = s.Ho; c.Wi = s.Wo;
  c.To = s.Ti; c.Ho = s.Hi; c.Wo = s.Wi;
  c.pt = s.kt - 1 - s.pt; c.ph = s.kh - 1 - s.ph; c.pw = s.kw - 1 - s.pw;
  return c;
}

}  // namespace

extern "C" const char* pb_dwconv3d_kernel_name(const pb_dwconv_shape* s, pb_dtype dtype, int which) {
  if (!s) return "";
  if (which == 0) {
    if (pb_dw3_eligible(s)) return "dw3_fwd_kernel";
    if (pb_dw7_eligible(s)) return "dw7_fwd_kernel";
    if (pb_dw_rows_eligible(s, false)) return pb_dw_ring_eligible(s, dtype) ? "dw_ring_fwd_kernel" : "dw_rows_fwd_kernel";
    if (pb_dw_tile_eligible(s, dtype, false)) return "dw_tile_fwd_kernel";
    return strip_ok(*s) ? "dw_strip_fwd_kernel" : "dw_gen_fwd_kernel";
  }
  if (which == 1) {
    if (bwd_is_dw3_corr(*s)) return "dw3_fwd_kernel";
    if (bwd_is_s1_corr(*s)) {
      const pb_dwconv_shape c = bwd_corr_shape(*s);
      if (pb_dw7_eligible(&c)) return "dw7_fwd_kernel";
      if (pb_dw_rows_eligible(&c, false)) return pb_dw_ring_eligible(&c, dtype) ? "dw_ring_fwd_kernel" : "dw_rows_fwd_kernel";
      return pb_dw_tile_eligible(&c, dtype, false) ? "dw_tile_fwd_kernel" : "dw_strip_fwd_kernel";
    }
    if ((strip_ok(*s) && s->sh == 2 && s->ph == s->kh / 2) || bwd_is_kt_s2(*s)) return "dw_strip_bwd_s2_kernel";
    return "dw_gen_bwd_data_kernel";
  }
  if (pb_dw7_eligible(s)) return "dw7_wgrad_kernel";
  if (pb_dw_rows_eligible(s, true)) return "dw_rows_wgrad_kernel";
  if (pb_dw_tile_eligible(s, dtype, true)) return "dw_tile_wgrad_kernel";
  return strip_ok(*s) ? "dw_strip_wgrad_kernel" : "dw_gen_wgrad_kernel";
}

namespace {
// generic pooled forward: grid geometry (also the number of pool slots)
int64_t gen_pool_geom(const pb_dwconv_shape* s, int* ch_o, int* cvb_o, int* P_o, int* cchunks_o) {
  const int ch = s->C % 4 == 0 ? 4 : 1;
  const int CV = s->C / ch;
  const int cvb = pick_cvb(CV), P = 256 / cvb, cchunks = (int)pb_cdiv(CV, cvb);
  int64_t gx = pb_cdiv((int64_t)s->To * s->Ho * s->Wo, (int64_t)P * 4);
  const int64_t cap = pb_cdiv(4096, (int64_t)cchunks * s->N);
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  if (ch_o) *ch_o = ch;
  if (cvb_o) *cvb_o = cvb;
  if (P_o) *P_o = P;
  if (cchunks_o) *cchunks_o = cchunks;
  return gx;
}
StripArgs strip_fwd_args(const void* x, const float* w, void* y, const pb_dwconv_shape* s, float* pool) {
  StripArgs a = {};
  a.src = x; a.w = w; a.dst = y; a.N = s->N; a.C = s->C;
  a.Tsrc = s->Ti; a.Hs = s->Hi; a.Ws = s->Wi; a.Tdst = s->To; a.Hd = s->Ho; a.Wd = s->Wo;
  a.pad = s->ph; a.t_mul = s->st; a.t_off = -s->pt; a.flip = 0; a.pool = pool;
  return a;
}
}  // namespace

// Number of squeeze-excite partial-sum slots pb_dwconv3d_fwd writes for this shape: follows the kernel dispatch.
extern "C" int pb_dwconv3d_pool_slots(const pb_dwconv_shape* s, pb_dtype dtype) {
  if (!s || s->N <= 0 || s->C <= 0) return 0;
  if (pb_dw3_eligible(s)) return pb_dw3_pool_slots(s);
  if (pb_dw7_eligible(s)) return s->To;
  if (pb_dw_rows_eligible(s, false)) return pb_dw_rows_pool_slots(s, dtype);
  if (strip_ok(*s)) {      // (the tile kernel has no fused pooling: pooled forwards skip it)
    StripArgs a = strip_fwd_args(nullptr, nullptr, nullptr, s, nullptr);
    return (int)(s->To * strip_fwd_geom(a, strip_ch(dtype == PB_BF16, s->kh, s->C)));
  }
  return (int)gen_pool_geom(s, nullptr, nullptr, nullptr, nullptr);
}

extern "C" int pb_dwconv3d_fwd(const void* x, const float* w, void* y, const pb_dwconv_shape* s, pb_dtype dtype,
                               float* pool, int pool_slots, int device, void* stream) {
  PB_CHECK_ARG(x && w && y, "pb_dwconv3d_fwd: null pointer");
  if (int rc = check_shape(s, "pb_dwconv3d_fwd")) return rc;
  PB_CHECK_ARG(dtype == PB_F32 || dtype == PB_BF16, "pb_dwconv3d_fwd: bad dtype");
  PB_CHECK_ARG(!pool || pool_slots == pb_dwconv3d_pool_slots(s, dtype), "pb_dwconv3d_fwd: pool_slots=%d, this shape writes %d",
               pool_slots, pb_dwconv3d_pool_slots(s, dtype));
  PB_ENTER(device);
  hipStream_t st = (hipStream_t)stream;
  {
    int rc = pb_dw3_fwd(x, w, y, s, dtype, 0, nullptr, nullptr, pool, st);
    if (rc >= 0) return rc;
    rc = pb_dw7_fwd(x, w, y, s, dtype, 0, nullptr, nullptr, pool, st);
    if (rc >= 0) return rc;
    rc = pb_dw_rows_fwd(x, w, y, s, dtype, 0, nullptr, nullptr, pool, st);
    if (rc >= 0) return rc;
    if (!pool) {
      rc = pb_dw_tile_fwd(x, w, y, s, dtype, 0, nullptr, nullptr, st);
      if (rc >= 0) return rc;
    }
  }
  if (strip_ok(*s)) {
    StripArgs a = strip_fwd_args(x, w, y, s, pool);
    int rc = dtype == PB_BF16 ? dispatch_strip<bf16_t, false>(a, s->kh, s->sh, st)
                              : dispatch_strip<float, false>(a, s->kh, s->sh, st);
    if (rc) return rc;
  } else {
    GenArgs a = {};
    a.src = x; a.w = w; a.dst = y; a.s = *s; a.pool = pool;
    const int ch = s->C % 4 == 0 ? 4 : 1;
    a.CV = s->C / ch;
    a.total = (int64_t)s->N * s->To * s->Ho * s->Wo * a.CV;
    if (pool) {
      int cvb, P, cchunks;
      const int64_t gx = gen_pool_geom(s, nullptr, &cvb, &P, &cchunks);
      dim3 pg((unsigned)gx, (unsigned)cchunks, (unsigned)s->N);
      if (dtype == PB_BF16) {
        if (ch == 4) hipLaunchKernelGGL((dw_gen_fwd_pool_kernel<bf16_t, 4>), pg, dim3(256), 0, st, a, cvb, P, cchunks);
        else hipLaunchKernelGGL((dw_gen_fwd_pool_kernel<bf16_t, 1>), pg, dim3(256), 0, st, a, cvb, P, cchunks);
      } else {
        if (ch == 4) hipLaunchKernelGGL((dw_gen_fwd_pool_kernel<float, 4>), pg, dim3(256), 0, st, a, cvb, P, cchunks);
        else hipLaunchKernelGGL((dw_gen_fwd_pool_kernel<float, 1>), pg, dim3(256), 0, st, a, cvb, P, cchunks);
      }
      PB_LAUNCH_CHECK();
      return 0;
    }
    dim3 grid((unsigned)pb_cdiv(a.total, 256));
    if (dtype == PB_BF16) {
      if (ch == 4) hipLaunchKernelGGL((dw_gen_fwd_kernel<bf16_t, 4>), grid, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((dw_gen_fwd_kernel<bf16_t, 1>), grid, dim3(256), 0, st, a);
    } else {
      if (ch == 4) hipLaunchKernelGGL((dw_gen_fwd_kernel<float, 4>), grid, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((dw_gen_fwd_kernel<float, 1>), grid, dim3(256), 0, st, a);
    }
  }
  PB_LAUNCH_CHECK();
  return 0;
}

// Streaming (stateful) causal depthwise convolution: the `nhist` frames that precede x in time come from the
// caller-owned history buffer `hist` [N][nhist][Hi][Wi][C] (same dtype), not from zero padding - the chunked evaluation
// of a CausalConv3d (movinet.py:7-39) never re-reads history from a concatenated tensor.  s->pt = the total front
// padding of the equivalent unchunked call (>= nhist: frames before the history are zeros, as the conv's own padding).
// Only the (kt,3,3) kernels implement it: returns 3 ("shape not taken") otherwise and the caller concatenates.
extern "C" int pb_dwconv3d_fwd_hist(const void* x, const void* hist, int nhist, const float* w, void* y,
                                    const pb_dwconv_shape* s, pb_dtype dtype, int device, void* stream) {
  PB_CHECK_ARG(x && hist && w && y, "pb_dwconv3d_fwd_hist: null pointer");
  if (int rc = check_shape(s, "pb_dwconv3d_fwd_hist")) return rc;
  PB_CHECK_ARG(dtype == PB_F32 || dtype == PB_BF16, "pb_dwconv3d_fwd_hist: bad dtype");
  PB_CHECK_ARG(nhist >= 1 && nhist <= s->pt, "pb_dwconv3d_fwd_hist: nhist=%d must be in [1, pt=%d]", nhist, s->pt);
  if (!pb_dw3_eligible(s)) { pb_set_error("pb_dwconv3d_fwd_hist: shape not taken by the (kt,3,3) kernels"); return 3; }
  PB_ENTER(device);
  const int rc = pb_dw3_fwd(x, w, y, s, dtype, 0, nullptr, nullptr, nullptr, (hipStream_t)stream, hist, nhist);
  if (rc) return rc < 0 ? 3 : rc;
  PB_LAUNCH_CHECK();
  return 0;
}

extern "C" int pb_dwconv3d_bwd_data(const void* dy, const float* w, void* dx, const pb_dwconv_shape* s,
                                    pb_dtype dtype, const float* g_scale, const float* g_shift, int device,
                                    void* stream) {
  PB_CHECK_ARG(dy && w && dx, "pb_dwconv3d_bwd_data: null pointer");
  if (int rc = check_shape(s, "pb_dwconv3d_bwd_data")) return rc;
  PB_CHECK_ARG(dtype == PB_F32 || dtype == PB_BF16, "pb_dwconv3d_bwd_data: bad dtype");
  PB_ENTER(device);
  hipStream_t st = (hipStream_t)stream;
  if (bwd_is_dw3_corr(*s)) {
    const pb_dwconv_shape c = bwd_dw3_shape(*s);
    int rc = pb_dw3_fwd(dy, w, dx, &c, dtype, 1, g_scale, g_shift, nullptr, st);
    if (rc >= 0) return rc;
  }
  // stride-1 spatial + temporal: correlation of dy with the flipped filter, padding k-1-p
  if (bwd_is_s1_corr(*s)) {
    const pb_dwconv_shape c = bwd_corr_shape(*s);
    int rc = pb_dw7_fwd(dy, w, dx, &c, dtype, 1, g_scale, g_shift, nullptr, st);
    if (rc >= 0) return rc;
    rc = pb_dw_rows_fwd(dy, w, dx, &c, dtype, 1, g_scale, g_shift, nullptr, st);
    if (rc >= 0) return rc;
    rc = pb_dw_tile_fwd(dy, w, dx, &c, dtype, 1, g_scale, g_shift, st);
    if (rc >= 0) return rc;
  }
  if (bwd_is_s1_corr(*s)) {
    StripArgs a = {};
    a.src = dy; a.w = w; a.dst = dx; a.N = s->N; a.C = s->C;
    a.Tsrc = s->To; a.Hs = s->Ho; a.Ws = s->Wo; a.Tdst = s->Ti; a.Hd = s->Hi; a.Wd = s->Wi;
    a.pad = s->kh - 1 - s->ph; a.t_mul = 1; a.t_off = s->pt; a.flip = 1;
    a.g_scale = g_scale; a.g_shift = g_shift;
    int rc = dtype == PB_BF16 ? dispatch_strip<bf16_t, false>(a, s->kh, 1, st) : dispatch_strip<float, false>(a, s->kh, 1, st);
    if (rc) return rc;
  } else if ((strip_ok(*s) && s->sh == 2 && s->ph == s->kh / 2) || bwd_is_kt_s2(*s)) {
    StripArgs a = {};
    a.src = dy; a.w = w; a.dst = dx; a.N = s->N; a.C = s->C;
    a.Tsrc = s->To; a.Hs = s->Ho; a.Ws = s->Wo; a.Tdst = s->Ti; a.Hd = s->Hi; a.Wd = s->Wi;
    a.t_mul = s->st; a.t_off = s->pt; a.g_scale = g_scale; a.g_shift = g_shift;
    const bool bf = dtype == PB_BF16;
    const int ch = (bf && s->kh == 3 && s->C % 8 == 0) ? 8 : 4;
    a.CV = a.C / ch;
    a.CVB = pick_cvb(a.CV);
    a.P = 256 / a.CVB;
    const int64_t nblk = (int64_t)((s->Hi + 1) / 2) * ((s->Wi + 1) / 2);
    a.xblocks = (int)pb_cdiv(nblk, a.P);
    a.items = (int64_t)s->N * s->Ti * a.xblocks;
    const int chunks = (int)pb_cdiv(a.CV, a.CVB);
    int64_t gx = pb_cdiv(2048, chunks);
    if (gx > a.items) gx = a.items;
    dim3 grid((unsigned)gx, (unsigned)chunks);
    if (s->kt == 3) {
      if (bf && ch == 8) hipLaunchKernelGGL((dw_strip_bwd_s2_kernel<bf16_t, 3, 8, 3>), grid, dim3(256), 0, st, a);
      else if (bf) hipLaunchKernelGGL((dw_strip_bwd_s2_kernel<bf16_t, 3, 4, 3>), grid, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((dw_strip_bwd_s2_kernel<float, 3, 4, 3>), grid, dim3(256), 0, st, a);
    } else if (s->kt == 5) {
      if (bf && ch == 8) hipLaunchKernelGGL((dw_strip_bwd_s2_kernel<bf16_t, 3, 8, 5>), grid, dim3(256), 0, st, a);
      else if (bf) hipLaunchKernelGGL((dw_strip_bwd_s2_kernel<bf16_t, 3, 4, 5>), grid, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((dw_strip_bwd_s2_kernel<float, 3, 4, 5>), grid, dim3(256), 0, st, a);
    } else if (bf) {
      if (s->kh == 3 && ch == 8) hipLaunchKernelGGL((dw_strip_bwd_s2_kernel<bf16_t, 3, 8>), grid, dim3(256), 0, st, a);
      else if (s->kh == 3) hipLaunchKernelGGL((dw_strip_bwd_s2_kernel<bf16_t, 3, 4>), grid, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((dw_strip_bwd_s2_kernel<bf16_t, 5, 4>), grid, dim3(256), 0, st, a);
    } else {
      if (s->kh == 3) hipLaunchKernelGGL((dw_strip_bwd_s2_kernel<float, 3, 4>), grid, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((dw_strip_bwd_s2_kernel<float, 5, 4>), grid, dim3(256), 0, st, a);
    }
  } else {
    GenArgs a = {};
    a.src = dy; a.w = w; a.dst = dx; a.s = *s; a.g_scale = g_scale; a.g_shift = g_shift;
    const int ch = s->C % 4 == 0 ? 4 : 1;
    a.CV = s->C / ch;
    a.total = (int64_t)s->N * s->Ti * s->Hi * s->Wi * a.CV;
    dim3 grid((unsigned)pb_cdiv(a.total, 256));
    if (dtype == PB_BF16) {
      if (ch == 4) hipLaunchKernelGGL((dw_gen_bwd_data_kernel<bf16_t, 4>), grid, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((dw_gen_bwd_data_kernel<bf16_t, 1>), grid, dim3(256), 0, st, a);
    } else {
      if (ch == 4) hipLaunchKernelGGL((dw_gen_bwd_data_kernel<float, 4>), grid, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((dw_gen_bwd_data_kernel<float, 1>), grid, dim3(256), 0, st, a);
    }
  }
  PB_LAUNCH_CHECK();
  return 0;
}

namespace {
StripArgs strip_wgrad_args(const void* dy, const void* x, float* part, const pb_dwconv_shape* s, const float* g_scale,
                           const float* g_shift) {
  StripArgs a = {};
  a.dst = const_cast<void*>(dy); a.x = x; a.dw = part; a.N = s->N; a.C = s->C;
  a.Tsrc = s->Ti; a.Hs = s->Hi; a.Ws = s->Wi; a.Tdst = s->To; a.Hd = s->Ho; a.Wd = s->Wo;
  a.pad = s->ph; a.t_mul = s->st; a.t_off = -s->pt;
  a.g_scale = g_scale; a.g_shift = g_shift;
  return a;
}
int64_t gen_wgrad_gx(const pb_dwconv_shape* s, int* ch_o, int* cvb_o, int* P_o, int* cchunks_o) {
  const int ch = s->C % 4 == 0 ? 4 : 1;
  const int CV = s->C / ch;
  const int cvb = pick_cvb(CV), P = 256 / cvb, cchunks = (int)pb_cdiv(CV, cvb);
  const int taps = s->kt * s->kh * s->kw;
  const int64_t nvox = (int64_t)s->N * s->To * s->Ho * s->Wo;
  int64_t gx = pb_cdiv(nvox, P);
  const int64_t cap = pb_cdiv(4096, (int64_t)taps * cchunks);
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  if (ch_o) *ch_o = ch;
  if (cvb_o) *cvb_o = cvb;
  if (P_o) *P_o = P;
  if (cchunks_o) *cchunks_o = cchunks;
  return gx;
}
// partial gradients [slots][C][taps] the weight-gradient dispatch writes for this shape
int64_t wgrad_slots(const pb_dwconv_shape* s, pb_dtype dtype) {
  if (int64_t n = pb_dw7_wgrad_slots(s)) return n;
  if (int64_t n = pb_dw_rows_wgrad_slots(s, dtype)) return n;
  if (int64_t n = pb_dw_tile_wgrad_slots(s, dtype)) return n;
  if (strip_ok(*s)) {
    StripArgs a = strip_wgrad_args(nullptr, nullptr, nullptr, s, nullptr, nullptr);
    return strip_wgrad_geom(a, strip_ch(dtype == PB_BF16, s->kh, s->C));
  }
  return gen_wgrad_gx(s, nullptr, nullptr, nullptr, nullptr);
}
}  // namespace

extern "C" int64_t pb_dwconv3d_bwd_weight_ws_bytes(const pb_dwconv_shape* s, pb_dtype dtype) {
  if (!s || s->N <= 0 || s->C <= 0 || s->kt <= 0 || s->kh <= 0 || s->kw <= 0) return 0;
  return wgrad_slots(s, dtype) * s->C * s->kt * s->kh * s->kw * 4;
}

extern "C" int pb_dwconv3d_bwd_weight(const void* dy, const void* x, float* dw, const pb_dwconv_shape* s,
                                      pb_dtype dtype, const float* g_scale, const float* g_shift, void* ws,
                                      int64_t ws_bytes, pb_reduce_job* deferred, int device, void* stream) {
  PB_CHECK_ARG(dy && x && dw && ws, "pb_dwconv3d_bwd_weight: null pointer");
  if (int rc = check_shape(s, "pb_dwconv3d_bwd_weight")) return rc;
  PB_CHECK_ARG(dtype == PB_F32 || dtype == PB_BF16, "pb_dwconv3d_bwd_weight: bad dtype");
  PB_CHECK_ARG(((uintptr_t)ws % 16) == 0 && ws_bytes >= pb_dwconv3d_bwd_weight_ws_bytes(s, dtype),
               "pb_dwconv3d_bwd_weight: workspace too small or unaligned (%lld bytes needed)",
               (long long)pb_dwconv3d_bwd_weight_ws_bytes(s, dtype));
  PB_ENTER(device);
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)ws;
  const int taps = s->kt * s->kh * s->kw;
  int64_t slots = -1;
  {
    int rc = pb_dw7_wgrad(dy, x, part, s, dtype, g_scale, g_shift, st, &slots);
    if (rc < 0) rc = pb_dw_rows_wgrad(dy, x, part, s, dtype, g_scale, g_shift, st, &slots);
    if (rc < 0) rc = pb_dw_tile_wgrad(dy, x, part, s, dtype, g_scale, g_shift, st, &slots);
    if (rc > 0) return rc;
    if (rc < 0) slots = -1;
  }
  if (slots < 0 && strip_ok(*s)) {
    StripArgs a = strip_wgrad_args(dy, x, part, s, g_scale, g_shift);
    StripArgs g = a;
    slots = strip_wgrad_geom(g, strip_ch(dtype == PB_BF16, s->kh, s->C));
    int rc = dtype == PB_BF16 ? dispatch_strip<bf16_t, true>(a, s->kh, s->sh, st) : dispatch_strip<float, true>(a, s->kh, s->sh, st);
    if (rc) return rc;
  } else if (slots < 0) {
    GenArgs a = {};
    a.src = dy; a.x = x; a.dw = part; a.s = *s; a.g_scale = g_scale; a.g_shift = g_shift;
    int ch, cvb, P, cchunks;
    const int64_t gx = gen_wgrad_gx(s, &ch, &cvb, &P, &cchunks);
    a.CV = s->C / ch;
    slots = gx;
    dim3 grid((unsigned)gx, (unsigned)(taps * cchunks));
    if (dtype == PB_BF16) {
      if (ch == 4) hipLaunchKernelGGL((dw_gen_wgrad_kernel<bf16_t, 4>), grid, dim3(256), 0, st, a, cvb, P, cchunks);
      else hipLaunchKernelGGL((dw_gen_wgrad_kernel<bf16_t, 1>), grid, dim3(256), 0, st, a, cvb, P, cchunks);
    } else {
      if (ch == 4) hipLaunchKernelGGL((dw_gen_wgrad_kernel<float, 4>), grid, dim3(256), 0, st, a, cvb, P, cchunks);
      else hipLaunchKernelGGL((dw_gen_wgrad_kernel<float, 1>), grid, dim3(256), 0, st, a, cvb, P, cchunks);
    }
  }
  PB_LAUNCH_CHECK();
  // dw += sum of the partial gradients, in slot order
  const int64_t n = (int64_t)s->C * taps;
  const pb_reduce_job job = {part, dw, n, n, (int)slots, 1};
  if (deferred) { *deferred = job; return 0; }
  if (pb_launch_partials_reduce(&job, 1, st)) { pb_set_error("pb_dwconv3d_bwd_weight: combine launch failed"); return 2; }
  return 0;
}

// Backward-data AND weight gradient of a (1,k,k) stride-1 "same" depthwise conv in one launch (bf16): dy is read once.
// pb_dwconv3d_bwd_fused_ws_bytes = 0: the shape is not taken - call pb_dwconv3d_bwd_data and pb_dwconv3d_bwd_weight.
extern "C" int64_t pb_dwconv3d_bwd_fused_ws_bytes(const pb_dwconv_shape* s, pb_dtype dtype) {
  if (!s || s->N <= 0 || s->C <= 0 || s->kt <= 0 || s->kh <= 0 || s->kw <= 0) return 0;
  return pb_dw_rows_bwd_fused_slots(s, dtype) * s->C * s->kt * s->kh * s->kw * 4;
}

extern "C" int pb_dwconv3d_bwd_fused(const void* dy, const void* x, const float* w, void* dx, float* dw, const pb_dwconv_shape* s,
                                     pb_dtype dtype, const float* g_scale, const float* g_shift, void* ws, int64_t ws_bytes,
                                     pb_reduce_job* deferred, int device, void* stream) {
  PB_CHECK_ARG(dy && x && w && dx && dw && ws, "pb_dwconv3d_bwd_fused: null pointer");
  if (int rc = check_shape(s, "pb_dwconv3d_bwd_fused")) return rc;
  const int64_t need = pb_dwconv3d_bwd_fused_ws_bytes(s, dtype);
  PB_CHECK_ARG(need > 0, "pb_dwconv3d_bwd_fused: shape / dtype not taken by the fused kernel (ws_bytes query returns 0)");
  PB_CHECK_ARG(((uintptr_t)ws % 16) == 0 && ws_bytes >= need, "pb_dwconv3d_bwd_fused: workspace too small or unaligned (%lld bytes needed)", (long long)need);
  PB_ENTER(device);
  hipStream_t st = (hipStream_t)stream;
  int64_t slots = 0;
  const int rc = pb_dw_rows_bwd_fused(dy, x, w, dx, (float*)ws, s, dtype, g_scale, g_shift, st, &slots);
  if (rc) { if (rc < 0) pb_set_error("pb_dwconv3d_bwd_fused: shape not taken"); return rc < 0 ? 3 : rc; }
  PB_LAUNCH_CHECK();
  const int64_t n = (int64_t)s->C * s->kt * s->kh * s->kw;
  const pb_reduce_job job = {(float*)ws, dw, n, n, (int)slots, 1};
  if (deferred) { *deferred = job; return 0; }
  if (pb_launch_partials_reduce(&job, 1, st)) { pb_set_error("pb_dwconv3d_bwd_fused: combine launch failed"); return 2; }
  return 0;
}
