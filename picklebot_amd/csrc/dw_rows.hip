// Depthwise (1,k,k) "same" convolution with lanes along CHANNEL PAIRS and a rolling window of input rows in
// registers.  A thread owns two channels of a WSEG-wide column segment and walks down a band of output rows:
// every input element it needs is loaded exactly once (4 bytes per lane; a wave's load is one contiguous 256-byte
// run of an NDHWC pixel), converted to fp32 once (with the squeeze-excite prologue and the zero padding) when
// its row enters the k-row window, and the next step's rows are already in flight while a row is computed.
// No LDS, no cross-lane traffic, all tap indices static (the row loop is unrolled by k so that window slots are
// compile-time).  The strip kernels (lanes along x, one output column per thread) re-load every input column k
// times through L1, the tile kernels pay an LDS round trip and per-tile bookkeeping; this form reaches the same
// 3-4 TB/s on the 14x14 / 28x28 stages that dw7.hip reaches on 7x7.
#include "dw_internal.h"
#include <stdlib.h>

namespace {

struct RowsArgs {
  const void* src;
  void* dst;
  const float* w;
  const float* g_scale;
  const float* g_shift;
  float* pool;         // squeeze-excite partial sums [Tdst*nbands*nseg][N][C]: one slot per (frame, band, segment)
  int N, C, Tsrc, Hs, Ws, Tdst, Hd, Wd, t_mul, t_off;
  int nseg, nbands, band_rows, CP;
};

template <typename T, int K, int S, int WSEG, bool FLIP, bool PRO>
__global__ __launch_bounds__(256) void dw_rows_fwd_kernel(RowsArgs a) {
  typedef typename Pair<T>::raw_t raw_t;
  constexpr int P = K / 2;
  constexpr int IW = (WSEG - 1) * S + K;
  // work units (frame-band, segment, channel pair) are flattened over the grid: a frame with few channel pairs
  // does not leave most of a 256-thread workgroup idle; lanes still run along the channel pairs
  const int64_t gu = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int per_frame = a.CP * a.nseg;
  const int64_t zb = gu / per_frame;
  if (zb >= (int64_t)a.N * a.Tdst * a.nbands) return;
  const int u = (int)(gu - zb * per_frame);
  const int sg = u / a.CP, cp = u - sg * a.CP;
  const int c0 = cp * 2;
  const int z = (int)(zb / a.nbands), band = (int)(zb - (int64_t)z * a.nbands);
  const int n = z / a.Tdst, td = z - n * a.Tdst;
  const int ts = td * a.t_mul + a.t_off;
  const int h0 = band * a.band_rows;
  const int nrows = min(a.Hd, h0 + a.band_rows) - h0;
  if (nrows <= 0) return;
  const int w0 = sg * WSEG;
  const int ps = a.C / 2;                                   // pixel stride in raw_t units
  raw_t* dst = (raw_t*)a.dst + (((int64_t)z * a.Hd + h0) * a.Wd * a.C + c0) / 2 + (int64_t)w0 * ps;
  // this unit's slot of the pooled partial sums (plain store: every slot has exactly one writer)
  float* pslot = a.pool ? a.pool + ((((int64_t)td * a.nbands + band) * a.nseg + sg) * a.N + n) * a.C + c0 : nullptr;
  if (ts < 0 || ts >= a.Tsrc) {                             // the reference's zero frames
    const raw_t zero = Pair<T>::down(f32x2{0.f, 0.f});
    for (int t = 0; t < nrows; ++t)
#pragma unroll
      for (int j = 0; j < WSEG; ++j)
        if (w0 + j < a.Wd) dst[((int64_t)t * a.Wd + j) * ps] = zero;
    if (pslot) *reinterpret_cast<f32x2*>(pslot) = f32x2{0.f, 0.f};
    return;
  }
  const raw_t* src = (const raw_t*)a.src + ((((int64_t)n * a.Tsrc + ts) * a.Hs) * a.Ws * a.C + c0) / 2;
  const int ws0 = w0 * S - P;
  unsigned cmask = 0;
  int coloff[IW];
#pragma unroll
  for (int j = 0; j < IW; ++j) {
    const int ws = ws0 + j;
    cmask |= (ws >= 0 && ws < a.Ws) ? (1u << j) : 0u;
    coloff[j] = min(max(ws, 0), a.Ws - 1) * ps;
  }
  float wt[2 * K * K];
#pragma unroll
  for (int i = 0; i < 2 * K * K; ++i) wt[i] = a.w[(int64_t)c0 * (K * K) + i];
  f32x2 gs = {1.f, 1.f}, gh = {0.f, 0.f};
  if (a.g_scale) gs = *reinterpret_cast<const f32x2*>(a.g_scale + (int64_t)n * a.C + c0);
  if (a.g_shift) gh = *reinterpret_cast<const f32x2*>(a.g_shift + (int64_t)n * a.C + c0);

  f32x2 win[K][IW];            // slot of input row i (relative to the band's first window row) = i mod K
  raw_t raw[S][IW];            // the next step's new rows, in flight
  const int hs_base = h0 * S - P;                           // input row of window row 0 (band_rows % K == 0)
  auto load_row = [&](int hs, raw_t (&r)[IW]) {
    const raw_t* base = src + (int64_t)min(max(hs, 0), a.Hs - 1) * a.Ws * ps;
#pragma unroll
    for (int j = 0; j < IW; ++j) r[j] = base[coloff[j]];
  };
  // the kernels are VALU-bound (SQ_ACTIVE_INST_VALU is the highest of the whole step): interior rows of interior
  // segments skip the padding selects, and the (g_scale, g_shift) prologue only exists in the PRO instantiation
  constexpr unsigned FULL = IW >= 32 ? 0xffffffffu : ((1u << IW) - 1u);
  auto insert_row = [&](int hs, const raw_t (&r)[IW], f32x2 (&slot)[IW]) {
    const bool rowok = hs >= 0 && hs < a.Hs;
    if (rowok && cmask == FULL) {
#pragma unroll
      for (int j = 0; j < IW; ++j) slot[j] = PRO ? Pair<T>::up(r[j]) * gs + gh : Pair<T>::up(r[j]);
    } else {
#pragma unroll
      for (int j = 0; j < IW; ++j) {
        const bool ok = rowok && ((cmask >> j) & 1u);
        const f32x2 v = PRO ? Pair<T>::up(r[j]) * gs + gh : Pair<T>::up(r[j]);
        slot[j] = ok ? v : f32x2{0.f, 0.f};
      }
    }
  };
  // first step: all K window rows
  {
    raw_t r0[K][IW];
#pragma unroll
    for (int kh = 0; kh < K; ++kh) load_row(hs_base + kh, r0[kh]);
#pragma unroll
    for (int kh = 0; kh < K; ++kh) insert_row(hs_base + kh, r0[kh], win[kh % K]);
  }
  if (nrows > 1) {
#pragma unroll
    for (int s = 0; s < S; ++s) load_row(hs_base + K + s, raw[s]);      // new rows of step 1: window rows K .. K+S-1
  }
  f32x2 psum = {0.f, 0.f};
  for (int tk = 0; tk < nrows; tk += K) {
#pragma unroll
    for (int r = 0; r < K; ++r) {
      const int t = tk + r;
      if (t < nrows) {
        // window rows of step t: t*S .. t*S+K-1; new ones (t > 0): t*S+K-S .. t*S+K-1
        if (t > 0) {
#pragma unroll
          for (int s = 0; s < S; ++s) insert_row(hs_base + t * S + K - S + s, raw[s], win[(r * S + K - S + s) % K]);
        }
        if (t + 1 < nrows) {
#pragma unroll
          for (int s = 0; s < S; ++s) load_row(hs_base + (t + 1) * S + K - S + s, raw[s]);
        }
        raw_t* orow = dst + (int64_t)t * a.Wd * ps;
        // taps outside, the WSEG outputs inside: consecutive FMAs belong to different accumulators (a chain of dependent
        // v_pk_fma_f32 costs an s_nop per link - as many wait states as FMAs in the ISA of the output-major order)
        f32x2 acc[WSEG];
#pragma unroll
        for (int j = 0; j < WSEG; ++j) acc[j] = f32x2{0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < K; ++kh)
#pragma unroll
          for (int kw = 0; kw < K; ++kw) {
            const int tap = FLIP ? (K - 1 - kh) * K + (K - 1 - kw) : kh * K + kw;
            const f32x2 w2 = f32x2{wt[tap], wt[K * K + tap]};
#pragma unroll
            for (int j = 0; j < WSEG; ++j) pk_fma_keep(acc[j], win[(r * S + kh) % K][j * S + kw], w2);
          }
#pragma unroll
        for (int j = 0; j < WSEG; ++j) {
          if (w0 + j < a.Wd) {
            const raw_t o = Pair<T>::down(acc[j]);
            orow[(int64_t)j * ps] = o;
            if (a.pool) psum += Pair<T>::up(o);
          }
        }
      }
    }
  }
  if (pslot) *reinterpret_cast<f32x2*>(pslot) = psum;
}

struct RowsWgArgs {
  const void* x;       // [N, Tsrc, Hs, Ws, C]
  const void* dy;      // [N, Tdst, Hd, Wd, C]
  float* dw;           // partial gradients [slots][C][K*K] (workspace): slot = workgroup (CP <= 256) or (frame group, band, segment)
  const float* g_scale;
  const float* g_shift;
  int N, C, Tsrc, Hs, Ws, Tdst, Hd, Wd, t_mul, t_off;
  int nseg, nbands, band_rows, CP, FG;     // FG frame groups: a thread walks the frames z = fg, fg + FG, ...
  int tap_stride, tap_off;                 // a channel's k*k taps sit at [c*tap_stride + tap_off, +k*k) of a partial: a (kt,k,k)
                                           // filter is kt launches of this kernel, one per temporal tap plane (tap_off = dt*k*k)
};

// Weight gradient with the same thread layout: the k*k taps of the thread's two channels accumulate in
// registers over all the frames of its group; at the end every wave transposes its block (64 channel pairs x
// 2 x k*k floats, contiguous in dw) through LDS so that the atomics hit consecutive floats.
template <typename T, int K, int S, int WSEG>
__global__ __launch_bounds__(256) void dw_rows_wgrad_kernel(RowsWgArgs a) {
  typedef typename Pair<T>::raw_t raw_t;
  constexpr int P = K / 2;
  constexpr int IW = (WSEG - 1) * S + K;
  constexpr int KK = K * K;
  __shared__ float red[4][64 * 2 * KK];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t gu = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int per_frame = a.CP * a.nseg;
  const int64_t units = (int64_t)a.FG * a.nbands * per_frame;
  const bool active = gu < units;
  const int64_t guc = active ? gu : 0;
  const int fb = (int)(guc / per_frame);                    // (frame group, band)
  const int u = (int)(guc - (int64_t)fb * per_frame);
  const int sg = u / a.CP, cp = u - sg * a.CP;
  const int c0 = cp * 2;
  const int fg = fb / a.nbands, band = fb - fg * a.nbands;
  const int h0 = band * a.band_rows;
  const int nrows = min(a.Hd, h0 + a.band_rows) - h0;
  const int w0 = sg * WSEG;
  const int ps = a.C / 2;
  const int ws0 = w0 * S - P;
  unsigned cmask = 0;
  int coloff[IW];
#pragma unroll
  for (int j = 0; j < IW; ++j) {
    const int ws = ws0 + j;
    cmask |= (ws >= 0 && ws < a.Ws) ? (1u << j) : 0u;
    coloff[j] = min(max(ws, 0), a.Ws - 1) * ps;
  }
  f32x2 acc[KK];
#pragma unroll
  for (int i = 0; i < KK; ++i) acc[i] = f32x2{0.f, 0.f};
  const int hs_base = h0 * S - P;
  const int nframes = a.N * a.Tdst;
  if (active && nrows > 0) {
    for (int z = fg; z < nframes; z += a.FG) {
      const int n = z / a.Tdst, td = z - n * a.Tdst;
      const int ts = td * a.t_mul + a.t_off;
      if (ts < 0 || ts >= a.Tsrc) continue;                 // zero frames do not depend on x
      const raw_t* xs = (const raw_t*)a.x + ((((int64_t)n * a.Tsrc + ts) * a.Hs) * a.Ws * a.C + c0) / 2;
      const raw_t* gy = (const raw_t*)a.dy + (((int64_t)z * a.Hd + h0) * a.Wd * a.C + c0) / 2 + (int64_t)w0 * ps;
      f32x2 gs = {1.f, 1.f}, gh = {0.f, 0.f};
      if (a.g_scale) gs = *reinterpret_cast<const f32x2*>(a.g_scale + (int64_t)n * a.C + c0);
      if (a.g_shift) gh = *reinterpret_cast<const f32x2*>(a.g_shift + (int64_t)n * a.C + c0);
      f32x2 win[K][IW];
      raw_t raw[S][IW], graw[WSEG];
      auto load_row = [&](int hs, raw_t (&r)[IW]) {
        const raw_t* base = xs + (int64_t)min(max(hs, 0), a.Hs - 1) * a.Ws * ps;
#pragma unroll
        for (int j = 0; j < IW; ++j) r[j] = base[coloff[j]];
      };
      auto insert_row = [&](int hs, const raw_t (&r)[IW], f32x2 (&slot)[IW]) {
        const bool rowok = hs >= 0 && hs < a.Hs;
#pragma unroll
        for (int j = 0; j < IW; ++j) slot[j] = (rowok && ((cmask >> j) & 1u)) ? Pair<T>::up(r[j]) : f32x2{0.f, 0.f};
      };
      auto load_g = [&](int t) {
        const raw_t* base = gy + (int64_t)t * a.Wd * ps;
#pragma unroll
        for (int j = 0; j < WSEG; ++j) graw[j] = base[(int64_t)min(j, a.Wd - 1 - w0) * ps];
      };
      {
        raw_t r0[K][IW];
#pragma unroll
        for (int kh = 0; kh < K; ++kh) load_row(hs_base + kh, r0[kh]);
        load_g(0);
#pragma unroll
        for (int kh = 0; kh < K; ++kh) insert_row(hs_base + kh, r0[kh], win[kh % K]);
      }
      if (nrows > 1) {
#pragma unroll
        for (int s = 0; s < S; ++s) load_row(hs_base + K + s, raw[s]);
      }
      for (int tk = 0; tk < nrows; tk += K) {
#pragma unroll
        for (int r = 0; r < K; ++r) {
          const int t = tk + r;
          if (t < nrows) {
            f32x2 g[WSEG];
#pragma unroll
            for (int j = 0; j < WSEG; ++j) g[j] = (w0 + j < a.Wd) ? Pair<T>::up(graw[j]) * gs + gh : f32x2{0.f, 0.f};
            if (t > 0) {
#pragma unroll
              for (int s = 0; s < S; ++s) insert_row(hs_base + t * S + K - S + s, raw[s], win[(r * S + K - S + s) % K]);
            }
            if (t + 1 < nrows) {
#pragma unroll
              for (int s = 0; s < S; ++s) load_row(hs_base + (t + 1) * S + K - S + s, raw[s]);
              load_g(t + 1);
            }
#pragma unroll
            for (int kh = 0; kh < K; ++kh)
#pragma unroll
              for (int kw = 0; kw < K; ++kw)
#pragma unroll
                for (int j = 0; j < WSEG; ++j) acc[kh * K + kw] += g[j] * win[(r * S + kh) % K][j * S + kw];
          }
        }
      }
    }
  }
  const int nall = 2 * a.CP * KK;
  if (a.CP <= 256) {
    // narrow layers: the whole [C][k*k] gradient fits the workgroup's LDS.  Lanes of different segments / bands / frame
    // groups share channels: they add their taps in rounds (round r = the threads r*CP .. (r+1)*CP-1, which own distinct
    // channel pairs), so the order of the additions is fixed; then ONE contiguous store of the workgroup's partial.
    float* ra = &red[0][0];
    for (int i = threadIdx.x; i < nall; i += 256) ra[i] = 0.f;
    __syncthreads();
    const int rounds = (256 + a.CP - 1) / a.CP;
    const int my_round = threadIdx.x / a.CP;
    for (int r = 0; r < rounds; ++r) {
      if (r == my_round && active && nrows > 0) {
#pragma unroll
        for (int i = 0; i < KK; ++i) {
          ra[c0 * KK + i] += acc[i][0];
          ra[(c0 + 1) * KK + i] += acc[i][1];
        }
      }
      __syncthreads();
    }
    float* part = a.dw + (int64_t)blockIdx.x * 2 * a.CP * a.tap_stride + a.tap_off;
    for (int i = threadIdx.x; i < nall; i += 256) {
      const int c = i / KK;
      part[(int64_t)c * a.tap_stride + (i - c * KK)] = ra[i];
    }
    return;
  }
  // wave-private transposition: lane l holds the 2*KK consecutive floats [c0*KK, (c0+2)*KK) of its unit's partial
  float* rw = red[wave];
#pragma unroll
  for (int i = 0; i < KK; ++i) {
    rw[lane * 2 * KK + i] = acc[i][0];
    rw[lane * 2 * KK + KK + i] = acc[i][1];
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  const int64_t gu0 = gu - lane;                            // unit of lane 0
#pragma unroll
  for (int k = 0; k < 2 * KK; ++k) {
    const int e = lane + 64 * k;                            // element of the wave's block
    const int l2 = e / (2 * KK), off = e - l2 * (2 * KK);   // source lane, offset inside its 2*KK floats
    const int64_t g2 = gu0 + l2;
    if (g2 < units) {
      const int fb2 = (int)(g2 / per_frame);
      const int u2 = (int)(g2 - (int64_t)fb2 * per_frame);
      const int sg2 = u2 / a.CP, cp2 = u2 - sg2 * a.CP;
      // slot (frame group, band, segment) has exactly one unit per channel pair: plain stores, every element written
      const int ch2 = off >= KK;                            // which of the pair's two channels
      a.dw[(((int64_t)fb2 * a.nseg + sg2) * 2 * a.CP + cp2 * 2 + ch2) * a.tap_stride + a.tap_off + (off - ch2 * KK)] = rw[e];
    }
  }
}

// ---- backward-data AND weight gradient in ONE walk over the output gradient (round 4, bf16, stride 1) ---------------
// A bottleneck's depthwise backward used to be two launches that both read dS = dy (backward-data: dy -> dx; weight gradient:
// dy, x -> dw).  With the window held over dy, ONE product serves both:
//     dx[ih][iw]  = sum_{a,b} g[ih-a+P][iw-b+P] * w[a][b]          g = dy * g_scale + g_shift (squeeze-excite prologue), 0 outside
//     dw[a][b]    = sum_{ih,iw} x[ih][iw] * g[ih-a+P][iw-b+P]
// i.e. the window element G = g[ih-a+P][iw-b+P] that is multiplied by w[a][b] into dx[ih][iw] is the one that is multiplied
// by x[ih][iw] into dw[a][b].  A thread owns two channels of a WSEG-wide column segment (as above), walks a band of rows
// with the k-row window over g, loads one row of x (the segment's own columns: every x element belongs to exactly one
// segment, so nothing is counted twice) and issues two packed FMAs per (tap, column).  dy is read once instead of twice,
// one launch and its ramp-up / drain disappears per block.  The weight-gradient epilogue (fixed-order partial slots) and
// the persistent frame-group walk are dw_rows_wgrad_kernel's.
struct RowsFusedArgs {
  const void* x;       // [N, T, H, W, C]  forward input (read: weight gradient)
  const void* dy;      // [N, To, H, W, C] output gradient, To = T + 2*pt
  void* dx;            // [N, T, H, W, C]
  const float* w;      // [C][K*K]
  float* dw;           // partial gradients [slots][C][K*K]
  const float* g_scale;
  const float* g_shift;
  int N, C, T, H, W, To, pt;
  int nseg, nbands, band_rows, CP, FG;
};

template <int K, int WSEG, bool PRO>
__global__ __launch_bounds__(256) void dw_rows_bwd_fused_kernel(RowsFusedArgs a) {
  typedef unsigned raw_t;
  constexpr int P = K / 2;
  constexpr int IW = WSEG + K - 1;
  constexpr int KK = K * K;
  __shared__ float red[4][64 * 2 * KK];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t gu = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int per_frame = a.CP * a.nseg;
  const int64_t units = (int64_t)a.FG * a.nbands * per_frame;
  const bool active = gu < units;
  const int64_t guc = active ? gu : 0;
  const int fb = (int)(guc / per_frame);
  const int u = (int)(guc - (int64_t)fb * per_frame);
  const int sg = u / a.CP, cp = u - sg * a.CP;
  const int c0 = cp * 2;
  const int fg = fb / a.nbands, band = fb - fg * a.nbands;
  const int h0 = band * a.band_rows;
  const int nrows = min(a.H, h0 + a.band_rows) - h0;
  const int w0 = sg * WSEG;
  const int ps = a.C / 2;
  unsigned cmask = 0;                                        // window columns inside the frame
  int coloff[IW];
#pragma unroll
  for (int j = 0; j < IW; ++j) {
    const int ws = w0 - P + j;
    cmask |= (ws >= 0 && ws < a.W) ? (1u << j) : 0u;
    coloff[j] = min(max(ws, 0), a.W - 1) * ps;
  }
  float wt[2 * KK];
#pragma unroll
  for (int i = 0; i < 2 * KK; ++i) wt[i] = active ? a.w[(int64_t)c0 * KK + i] : 0.f;
  f32x2 acc[KK];
#pragma unroll
  for (int i = 0; i < KK; ++i) acc[i] = f32x2{0.f, 0.f};
  const int hs_base = h0 - P;
  const int nframes = a.N * a.T;
  if (active && nrows > 0) {
    for (int z = fg; z < nframes; z += a.FG) {
      const int n = z / a.T, ts = z - n * a.T;
      const int td = ts + a.pt;                              // (temporal stride 1: every input frame meets one output frame)
      const raw_t* gy = (const raw_t*)a.dy + ((((int64_t)n * a.To + td) * a.H) * a.W * a.C + c0) / 2;
      const raw_t* xs = (const raw_t*)a.x + (((int64_t)z * a.H + h0) * a.W * a.C + c0) / 2 + (int64_t)w0 * ps;
      raw_t* dxp = (raw_t*)a.dx + (((int64_t)z * a.H + h0) * a.W * a.C + c0) / 2 + (int64_t)w0 * ps;
      f32x2 gs = {1.f, 1.f}, gh = {0.f, 0.f};
      if (PRO) {
        if (a.g_scale) gs = *reinterpret_cast<const f32x2*>(a.g_scale + (int64_t)n * a.C + c0);
        if (a.g_shift) gh = *reinterpret_cast<const f32x2*>(a.g_shift + (int64_t)n * a.C + c0);
      }
      f32x2 win[K][IW];
      raw_t raw[IW], xraw[WSEG];
      auto load_row = [&](int hs, raw_t (&r)[IW]) {
        const raw_t* base = gy + (int64_t)min(max(hs, 0), a.H - 1) * a.W * ps;
#pragma unroll
        for (int j = 0; j < IW; ++j) r[j] = base[coloff[j]];
      };
      auto insert_row = [&](int hs, const raw_t (&r)[IW], f32x2 (&slot)[IW]) {
        const bool rowok = hs >= 0 && hs < a.H;
#pragma unroll
        for (int j = 0; j < IW; ++j) {
          const f32x2 v = PRO ? Pair<bf16_t>::up(r[j]) * gs + gh : Pair<bf16_t>::up(r[j]);
          slot[j] = (rowok && ((cmask >> j) & 1u)) ? v : f32x2{0.f, 0.f};
        }
      };
      auto load_x = [&](int t) {
        const raw_t* base = xs + (int64_t)t * a.W * ps;
#pragma unroll
        for (int j = 0; j < WSEG; ++j) xraw[j] = base[(int64_t)min(j, a.W - 1 - w0) * ps];
      };
      {
        raw_t r0[K][IW];
#pragma unroll
        for (int kh = 0; kh < K; ++kh) load_row(hs_base + kh, r0[kh]);
        load_x(0);
#pragma unroll
        for (int kh = 0; kh < K; ++kh) insert_row(hs_base + kh, r0[kh], win[kh % K]);
      }
      if (nrows > 1) load_row(hs_base + K, raw);
      for (int tk = 0; tk < nrows; tk += K) {
#pragma unroll
        for (int r = 0; r < K; ++r) {
          const int t = tk + r;
          if (t < nrows) {
            f32x2 xv[WSEG];
#pragma unroll
            for (int j = 0; j < WSEG; ++j) xv[j] = (w0 + j < a.W) ? Pair<bf16_t>::up(xraw[j]) : f32x2{0.f, 0.f};
            if (t > 0) insert_row(hs_base + t + K - 1, raw, win[(r + K - 1) % K]);
            if (t + 1 < nrows) {
              load_row(hs_base + t + K, raw);
              load_x(t + 1);
            }
            f32x2 dxa[WSEG];
#pragma unroll
            for (int j = 0; j < WSEG; ++j) dxa[j] = f32x2{0.f, 0.f};
#pragma unroll
            for (int kh = 0; kh < K; ++kh)
#pragma unroll
              for (int kw = 0; kw < K; ++kw) {
                const int tap = (K - 1 - kh) * K + (K - 1 - kw);          // w[a][b] with a = K-1-kh, b = K-1-kw
                const f32x2 w2 = f32x2{wt[tap], wt[KK + tap]};
#pragma unroll
                for (int j = 0; j < WSEG; ++j) {
                  const f32x2 G = win[(r + kh) % K][j + kw];
                  pk_fma_keep(dxa[j], G, w2);
                  acc[tap] += xv[j] * G;
                }
              }
            raw_t* orow = dxp + (int64_t)t * a.W * ps;
#pragma unroll
            for (int j = 0; j < WSEG; ++j)
              if (w0 + j < a.W) orow[(int64_t)j * ps] = Pair<bf16_t>::down(dxa[j]);
          }
        }
      }
    }
  }
  const int nall = 2 * a.CP * KK;
  if (a.CP <= 256) {      // narrow layers: fixed-order reduction in LDS, one contiguous partial per workgroup (as dw_rows_wgrad_kernel)
    float* ra = &red[0][0];
    for (int i = threadIdx.x; i < nall; i += 256) ra[i] = 0.f;
    __syncthreads();
    const int rounds = (256 + a.CP - 1) / a.CP;
    const int my_round = threadIdx.x / a.CP;
    for (int r = 0; r < rounds; ++r) {
      if (r == my_round && active && nrows > 0) {
#pragma unroll
        for (int i = 0; i < KK; ++i) {
          ra[c0 * KK + i] += acc[i][0];
          ra[(c0 + 1) * KK + i] += acc[i][1];
        }
      }
      __syncthreads();
    }
    float* part = a.dw + (int64_t)blockIdx.x * 2 * a.CP * KK;
    for (int i = threadIdx.x; i < nall; i += 256) part[i] = ra[i];
    return;
  }
  float* rw = red[wave];
#pragma unroll
  for (int i = 0; i < KK; ++i) {
    rw[lane * 2 * KK + i] = acc[i][0];
    rw[lane * 2 * KK + KK + i] = acc[i][1];
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  const int64_t gu0 = gu - lane;
#pragma unroll
  for (int k = 0; k < 2 * KK; ++k) {
    const int e = lane + 64 * k;
    const int l2 = e / (2 * KK), off = e - l2 * (2 * KK);
    const int64_t g2 = gu0 + l2;
    if (g2 < units) {
      const int fb2 = (int)(g2 / per_frame);
      const int u2 = (int)(g2 - (int64_t)fb2 * per_frame);
      const int sg2 = u2 / a.CP, cp2 = u2 - sg2 * a.CP;
      const int ch2 = off >= KK;
      a.dw[(((int64_t)fb2 * a.nseg + sg2) * 2 * a.CP + cp2 * 2 + ch2) * KK + (off - ch2 * KK)] = rw[e];
    }
  }
}

constexpr int fused_wseg(int K) { return K == 3 ? 7 : 2; }      // (k5 at 4 columns: 256 VGPRs + 48 AGPRs, one wave per SIMD)

bool fused_ok(const pb_dwconv_shape* s, pb_dtype dtype) {
  static const bool enabled = !(getenv("PB_DW_FUSED_BWD") && atoi(getenv("PB_DW_FUSED_BWD")) == 0);
  return enabled && dtype == PB_BF16 && s->kt == 1 && s->kh == s->kw && (s->kh == 3 || s->kh == 5) && s->st == 1 && s->sh == 1 &&
         s->sw == 1 && s->ph == s->kh / 2 && s->pw == s->kw / 2 && s->C % 2 == 0 && s->C >= 16 && s->Ho == s->Hi && s->Wo == s->Wi &&
         s->To == s->Ti + 2 * s->pt && (int64_t)s->Wi * s->C < (1 << 24) && !(s->Hi == 7 && s->Wi == 7);
}

// Workgroups that are co-resident on the chip for this kernel (occupancy x CUs), cached per instantiation.
template <typename KernelT>
int fused_resident(KernelT k) {
  static int cached = 0;
  if (cached == 0) {
    int per_cu = 0, cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, 256, 0) != hipSuccess || per_cu < 1) per_cu = 2;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    cached = per_cu * cus;
  }
  return cached;
}

// Geometry (frame groups FG x row bands) by ROUNDS, round 4: the launch time of these kernels is a staircase in
// ceil(workgroups / resident workgroups) (profiles/r04_dw_staircase_*.txt: 768 workgroups on 512 slots cost two rounds), so
// the split is chosen to minimise  rounds x frames per thread x (rows per band + halo)  - e.g. 195 frame groups = exactly
// 512 workgroups with 3 frames per thread for the 14x14x672 layer instead of 256 groups = 672 workgroups = two rounds of 2.
int64_t fused_geom(RowsFusedArgs& a, int K, int64_t* slots, int resident = 512) {
  const int WSEG = fused_wseg(K);
  a.CP = a.C / 2;
  a.nseg = (int)pb_cdiv(a.W, WSEG);
  const int64_t per_frame = (int64_t)a.CP * a.nseg;
  const int64_t frames = (int64_t)a.N * a.T;
  static const bool by_rounds = !(getenv("PB_DW_FUSED_ROUNDS") && atoi(getenv("PB_DW_FUSED_ROUNDS")) == 0);
  static const int want = getenv("PB_DW_FUSED_THREADS") ? atoi(getenv("PB_DW_FUSED_THREADS")) : 128 * 1024;
  int best_fg = 1, best_nb = 1;
  if (by_rounds) {
    double best = 1e30;
    const int max_bands = (int)pb_cdiv(a.H, 2 * K);
    // time ~ rounds x (start-up + frames per thread x rows per band): for every (bands, rounds) the LARGEST number of frame
    // groups whose workgroups still fit that many rounds; start-up of a workgroup (weights, first rows, ramp) ~ 8 row steps
    constexpr double STARTUP = 8.0;
    for (int nb = 1; nb <= max_bands && nb <= 4; ++nb) {
      const int band_rows = (int)pb_cdiv(pb_cdiv(a.H, nb), K) * K;
      const int nbe = (int)pb_cdiv(a.H, band_rows);
      if (nb > 1 && nbe != nb) continue;
      for (int r = 1; r <= 4; ++r) {
        int64_t fg = ((int64_t)r * resident * 256) / (nbe * per_frame);        // blocks(fg) <= r * resident
        if (fg > frames) fg = frames;
        if (a.CP > 256 && fg * nbe * a.nseg > 512) fg = 512 / ((int64_t)nbe * a.nseg);      // partial-gradient slots (12 MB at C = 672)
        if (fg < 1) continue;
        const int64_t fpt = pb_cdiv(frames, fg);
        const int64_t blocks = pb_cdiv(fg * nbe * per_frame, 256);
        const int64_t rounds = pb_cdiv(blocks, resident);
        const double cost = (double)rounds * (STARTUP + (double)fpt * (band_rows + K - 1));
        if (cost < best - 1e-9) { best = cost; best_fg = (int)fg; best_nb = nbe; }
        if (fg == frames) break;
      }
    }
  } else {
    int nbands = 1;
    const int max_bands = (int)pb_cdiv(a.H, 2 * K);
    int64_t fgroups = pb_cdiv(want, per_frame);
    if (fgroups > frames) {
      nbands = (int)pb_cdiv(fgroups, frames);
      if (nbands > max_bands) nbands = max_bands;
      fgroups = frames;
    }
    if (a.CP > 256) {
      const int64_t writer_cap = pb_cdiv(512, (int64_t)a.nseg * nbands);
      if (fgroups > writer_cap) fgroups = writer_cap;
    }
    best_fg = (int)(fgroups < 1 ? 1 : fgroups);
    best_nb = nbands;
  }
  a.band_rows = (int)pb_cdiv(pb_cdiv(a.H, best_nb), K) * K;
  a.nbands = (int)pb_cdiv(a.H, a.band_rows);
  a.FG = best_fg;
  const int64_t blocks = pb_cdiv((int64_t)a.FG * a.nbands * per_frame, 256);
  *slots = a.CP <= 256 ? blocks : (int64_t)a.FG * a.nbands * a.nseg;
  return blocks;
}

RowsFusedArgs fused_args(const pb_dwconv_shape* s) {
  RowsFusedArgs a = {};
  a.N = s->N; a.C = s->C; a.T = s->Ti; a.H = s->Hi; a.W = s->Wi; a.To = s->To; a.pt = s->pt;
  return a;
}

// (K, S) -> segment width: the fp32 window K x ((WSEG-1)*S + K) x 2 must stay near 100 registers
constexpr int wseg_of(int K, int S) { return K == 3 ? (S == 1 ? 14 : 7) : (S == 1 ? 7 : 4); }
// half-width segments: about half the window registers (twice the waves per SIMD) for 13-50 % more halo loads
constexpr int wseg_narrow(int K, int S) { return K == 3 ? (S == 1 ? 7 : 4) : (S == 1 ? 4 : 2); }

bool rows_ok(const pb_dwconv_shape* s) {
  static const bool enabled = !(getenv("PB_DW_ROWS") && atoi(getenv("PB_DW_ROWS")) == 0);
  static const int min_c = getenv("PB_DW_ROWS_MINC") ? atoi(getenv("PB_DW_ROWS_MINC")) : 16;
  return enabled && s->kt == 1 && s->kh == s->kw && (s->kh == 3 || s->kh == 5) && s->sh == s->sw && (s->sh == 1 || s->sh == 2) &&
         s->ph == s->kh / 2 && s->pw == s->kw / 2 && s->C % 2 == 0 && s->C >= min_c &&
         s->Ho == (s->Hi + 2 * s->ph - s->kh) / s->sh + 1 && s->Wo == (s->Wi + 2 * s->pw - s->kw) / s->sw + 1 &&
         (int64_t)s->Wi * s->C < (1 << 24);
}

// forward geometry (shared by the launcher and the pool-slot query): segment width, segments, row bands
bool rows_fwd_narrow(int K, int S, int C) {
  // PB_DW_ROWS_NARROW: bit 0 = k3 s1, bit 1 = k3 s2, bit 2 = k5 s1, bit 3 = k5 s2.  Measured on the training step:
  // k3 s1 gains 8-18 % from 64 channels up (the 16-channel 112x112 stage loses 17 %), the others do not move.
  static const int narrow = getenv("PB_DW_ROWS_NARROW") ? atoi(getenv("PB_DW_ROWS_NARROW")) : 1;
  const int bit = (K == 5 ? 2 : 0) + (S == 2 ? 1 : 0);
  return ((narrow >> bit) & 1) && C >= 64;
}
void rows_fwd_geom(RowsArgs& a, int K, int WSEG) {
  a.CP = a.C / 2;
  a.nseg = (int)pb_cdiv(a.Wd, WSEG);
  const int64_t per_frame = (int64_t)a.CP * a.nseg;
  const int64_t frames = (int64_t)a.N * a.Tdst;
  // enough threads for ~8 waves per SIMD... bands of rows (multiples of K) when the frames alone are too few
  static const int want = getenv("PB_DW_ROWS_THREADS") ? atoi(getenv("PB_DW_ROWS_THREADS")) : 160 * 1024;
  int nbands = (int)pb_cdiv(want, per_frame * frames);
  const int max_bands = (int)pb_cdiv(a.Hd, 2 * K);
  if (nbands > max_bands) nbands = max_bands;
  if (nbands < 1) nbands = 1;
  a.band_rows = (int)pb_cdiv(pb_cdiv(a.Hd, nbands), K) * K;
  a.nbands = (int)pb_cdiv(a.Hd, a.band_rows);
}

template <typename T, int K, int S, bool NARROW = false>
int launch_rows_fwd(RowsArgs a, int flip, hipStream_t st) {
  constexpr int WSEG = NARROW ? wseg_narrow(K, S) : wseg_of(K, S);
  if constexpr (!NARROW) {
    if (rows_fwd_narrow(K, S, a.C)) return launch_rows_fwd<T, K, S, true>(a, flip, st);
  }
  rows_fwd_geom(a, K, WSEG);
  const int64_t per_frame = (int64_t)a.CP * a.nseg;
  const int64_t frames = (int64_t)a.N * a.Tdst;
  const dim3 grid((unsigned)pb_cdiv(frames * a.nbands * per_frame, 256));
  const bool pro = a.g_scale || a.g_shift;
  if (flip) {
    if (pro) hipLaunchKernelGGL((dw_rows_fwd_kernel<T, K, S, WSEG, true, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((dw_rows_fwd_kernel<T, K, S, WSEG, true, false>), grid, dim3(256), 0, st, a);
  } else {
    if (pro) hipLaunchKernelGGL((dw_rows_fwd_kernel<T, K, S, WSEG, false, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((dw_rows_fwd_kernel<T, K, S, WSEG, false, false>), grid, dim3(256), 0, st, a);
  }
  if (hipGetLastError() != hipSuccess) { pb_set_error("dw_rows forward launch failed"); return 1; }
  return 0;
}

bool rows_wgrad_narrow(int K, int S, int C) {
  // bits as in the forward.  Measured: only the 16-channel 112x112 stage gains (95 -> 69 us), wider layers lose
  static const int narrow = getenv("PB_DW_ROWS_WG_NARROW") ? atoi(getenv("PB_DW_ROWS_WG_NARROW")) : 1;
  const int bit = (K == 5 ? 2 : 0) + (S == 2 ? 1 : 0);
  return ((narrow >> bit) & 1) && C < 64;
}
// weight-gradient geometry (shared by the launcher and the workspace query); returns the number of workgroups,
// *slots = partial gradients [C][K*K] the launch writes
int64_t rows_wgrad_geom(RowsWgArgs& a, int K, int WSEG, int64_t* slots) {
  a.CP = a.C / 2;
  a.nseg = (int)pb_cdiv(a.Wd, WSEG);
  const int64_t per_frame = (int64_t)a.CP * a.nseg;
  const int64_t frames = (int64_t)a.N * a.Tdst;
  // ~64K threads (the kernel's registers admit one wave per SIMD); row bands before frame groups; wide layers
  // (one partial gradient per (frame group, band, segment)) at most 128 partials
  int nbands = 1;
  const int max_bands = (int)pb_cdiv(a.Hd, 2 * K);
  int64_t fgroups = pb_cdiv(64 * 1024, per_frame);
  if (fgroups > frames) {
    nbands = (int)pb_cdiv(fgroups, frames);
    if (nbands > max_bands) nbands = max_bands;
    fgroups = frames;
  }
  if (a.CP > 256) {                 // (narrow layers reduce in LDS first: one partial per workgroup)
    const int64_t writer_cap = pb_cdiv(128, (int64_t)a.nseg * nbands);
    if (fgroups > writer_cap) fgroups = writer_cap;
  }
  if (fgroups < 1) fgroups = 1;
  a.band_rows = (int)pb_cdiv(pb_cdiv(a.Hd, nbands), K) * K;
  a.nbands = (int)pb_cdiv(a.Hd, a.band_rows);
  a.FG = (int)fgroups;
  const int64_t blocks = pb_cdiv(fgroups * a.nbands * per_frame, 256);
  *slots = a.CP <= 256 ? blocks : fgroups * a.nbands * a.nseg;
  return blocks;
}

// fp32 always takes the half-width segments: the full-width fp32 instantiations need 358-368 registers, and hipcc
// (ROCm 7.2) then splits live ranges into AGPRs with the save placed in a control-flow join block AHEAD of the exec
// restore - lanes that skipped the branch never save their value and later read back whatever the previous kernel left
// in that AGPR (observed on MI355X: the frame counter of dw_rows_wgrad_kernel<float,3,1,14>, i.e. waves that never
// leave the frame loop, depending on which kernels ran before).  tools/lint_isa.py scans the ISA for that pattern.
template <typename T, int K, int S, bool NARROW = (sizeof(T) == 4)>
int launch_rows_wgrad(RowsWgArgs a, hipStream_t st, int64_t* slots) {
  constexpr int WSEG = NARROW ? wseg_narrow(K, S) : wseg_of(K, S);
  if constexpr (!NARROW) {
    if (rows_wgrad_narrow(K, S, a.C)) return launch_rows_wgrad<T, K, S, true>(a, st, slots);
  }
  const int64_t blocks = rows_wgrad_geom(a, K, WSEG, slots);
  hipLaunchKernelGGL((dw_rows_wgrad_kernel<T, K, S, WSEG>), dim3((unsigned)blocks), dim3(256), 0, st, a);
  if (hipGetLastError() != hipSuccess) { pb_set_error("dw_rows weight-gradient launch failed"); return 1; }
  return 0;
}

}  // namespace

// (kt,k,k) filters with temporal stride 1 (MoViNet's depthwise convs, movinet.py:53): the weight gradient of tap plane dt
// is the (1,k,k) weight gradient of dy against x shifted by dt - pt frames - kt launches of the kernel above (the generic
// kernel these layers used ran at 0.09 TB/s: a third of a MoViNet-A2 training step)
bool rows_wgrad_planes_ok(const pb_dwconv_shape* s) {
  if (s->kt <= 1 || s->kt > 7 || s->st != 1) return false;
  pb_dwconv_shape p = *s;
  p.kt = 1;
  return rows_ok(&p);
}

bool pb_dw_rows_eligible(const pb_dwconv_shape* s, bool wgrad) {
  if (wgrad && rows_wgrad_planes_ok(s)) return true;
  // weight gradient, measured per layer on MI355X (MobileNetLarge3D, batch 32): this kernel wins on the wide layers
  // (C >= 480: 1.3-1.7x over the tile kernel) and on frames of 28x28 outputs or more (1.4-2.1x); the tile kernel
  // keeps the narrow 14x14 layers, the strip kernel the 64-channel 112-wide input
  static const bool all = getenv("PB_DW_ROWS_WGRAD_ALL") && atoi(getenv("PB_DW_ROWS_WGRAD_ALL")) != 0;
  if (!rows_ok(s)) return false;
  if (!wgrad || all) return true;
  return s->C >= 480 || (s->Ho * s->Wo >= 28 * 28 && (s->Wi < 112 || s->C < 64));
}

// number of partial gradients [C][k*k] pb_dw_rows_wgrad writes for this shape (0: shape not taken)
int64_t pb_dw_rows_wgrad_slots(const pb_dwconv_shape* s, pb_dtype dtype) {
  if (!pb_dw_rows_eligible(s, true)) return 0;
  RowsWgArgs a = {};
  a.N = s->N; a.C = s->C; a.Tdst = s->To; a.Hd = s->Ho; a.Wd = s->Wo;
  const int K = s->kh, S = s->sh;
  int64_t slots = 0;
  const bool narrow = dtype == PB_F32 || rows_wgrad_narrow(K, S, a.C);      // as launch_rows_wgrad
  rows_wgrad_geom(a, K, narrow ? wseg_narrow(K, S) : wseg_of(K, S), &slots);
  return slots;
}

// part: workspace of pb_dw_rows_wgrad_slots(s) partial gradients; *slots = how many were written
int pb_dw_rows_wgrad(const void* dy, const void* x, float* part, const pb_dwconv_shape* s, pb_dtype dtype,
                     const float* g_scale, const float* g_shift, hipStream_t st, int64_t* slots) {
  if (!pb_dw_rows_eligible(s, true)) return -1;
  RowsWgArgs a = {};
  a.x = x; a.dy = dy; a.dw = part; a.g_scale = g_scale; a.g_shift = g_shift;
  a.N = s->N; a.C = s->C; a.Tsrc = s->Ti; a.Hs = s->Hi; a.Ws = s->Wi; a.Tdst = s->To; a.Hd = s->Ho; a.Wd = s->Wo;
  a.t_mul = s->st;
  const int K = s->kh, S = s->sh;
  a.tap_stride = s->kt * K * K;
  // one launch per temporal tap plane: output frame t pairs with input frame t*st + dt - pt (frames outside the clip are zero)
  for (int dt = 0; dt < s->kt; ++dt) {
    a.t_off = dt - s->pt;
    a.tap_off = dt * K * K;
    int rc;
#define PB_ROWSW(TT)                                                                  \
  (K == 3 && S == 1 ? launch_rows_wgrad<TT, 3, 1>(a, st, slots)                        \
   : K == 3 && S == 2 ? launch_rows_wgrad<TT, 3, 2>(a, st, slots)                      \
   : K == 5 && S == 1 ? launch_rows_wgrad<TT, 5, 1>(a, st, slots) : launch_rows_wgrad<TT, 5, 2>(a, st, slots))
    if (dtype == PB_BF16) rc = PB_ROWSW(bf16_t);
    else rc = PB_ROWSW(float);
#undef PB_ROWSW
    if (rc) return rc;
  }
  return 0;
}

int pb_dw_rows_pool_slots(const pb_dwconv_shape* s, pb_dtype dtype) {
  if (const int n = pb_dw_ring_pool_slots(s, dtype)) return n;
  RowsArgs a = {};
  a.N = s->N; a.C = s->C; a.Tdst = s->To; a.Hd = s->Ho; a.Wd = s->Wo;
  const int K = s->kh, S = s->sh;
  rows_fwd_geom(a, K, rows_fwd_narrow(K, S, a.C) ? wseg_narrow(K, S) : wseg_of(K, S));
  return a.Tdst * a.nbands * a.nseg;
}

int pb_dw_rows_fwd(const void* src, const float* w, void* dst, const pb_dwconv_shape* s, pb_dtype dtype, int flip,
                   const float* g_scale, const float* g_shift, float* pool, hipStream_t st) {
  if (!rows_ok(s)) return -1;
  {
    const int rc = pb_dw_ring_fwd(src, w, dst, s, dtype, flip, g_scale, g_shift, pool, st);      // bf16: the LDS-ring form
    if (rc >= 0) return rc;
  }
  PB_CHECK_ARG((int64_t)s->N * s->To * s->Ho * s->Wo * s->C < (1LL << 40), "dw_rows: tensor too large");
  RowsArgs a = {};
  a.src = src; a.dst = dst; a.w = w; a.g_scale = g_scale; a.g_shift = g_shift; a.pool = pool;
  a.N = s->N; a.C = s->C; a.Tsrc = s->Ti; a.Hs = s->Hi; a.Ws = s->Wi; a.Tdst = s->To; a.Hd = s->Ho; a.Wd = s->Wo;
  a.t_mul = s->st; a.t_off = -s->pt;
  const int K = s->kh, S = s->sh;
#define PB_ROWS(TT)                                                              \
  do {                                                                           \
    if (K == 3 && S == 1) return launch_rows_fwd<TT, 3, 1>(a, flip, st);         \
    if (K == 3 && S == 2) return launch_rows_fwd<TT, 3, 2>(a, flip, st);         \
    if (K == 5 && S == 1) return launch_rows_fwd<TT, 5, 1>(a, flip, st);         \
    return launch_rows_fwd<TT, 5, 2>(a, flip, st);                               \
  } while (0)
  if (dtype == PB_BF16) PB_ROWS(bf16_t);
  PB_ROWS(float);
#undef PB_ROWS
}

// (one resident count per kernel size - the PRO variants differ by a few registers, the geometry must not depend on them)
static int fused_resident_of(int K) {
  return K == 3 ? fused_resident(dw_rows_bwd_fused_kernel<3, fused_wseg(3), true>) : fused_resident(dw_rows_bwd_fused_kernel<5, fused_wseg(5), true>);
}

// backward-data + weight gradient in one launch (bf16, (1,k,k), stride 1): 0 slots = shape not taken
int64_t pb_dw_rows_bwd_fused_slots(const pb_dwconv_shape* s, pb_dtype dtype) {
  if (!fused_ok(s, dtype)) return 0;
  RowsFusedArgs a = fused_args(s);
  int64_t slots = 0;
  fused_geom(a, s->kh, &slots, fused_resident_of(s->kh));
  return slots;
}

int pb_dw_rows_bwd_fused(const void* dy, const void* x, const float* w, void* dx, float* part, const pb_dwconv_shape* s,
                         pb_dtype dtype, const float* g_scale, const float* g_shift, hipStream_t st, int64_t* slots) {
  if (!fused_ok(s, dtype)) return -1;
  RowsFusedArgs a = fused_args(s);
  a.x = x; a.dy = dy; a.dx = dx; a.w = w; a.dw = part; a.g_scale = g_scale; a.g_shift = g_shift;
  const int64_t blocks = fused_geom(a, s->kh, slots, fused_resident_of(s->kh));
  const bool pro = g_scale || g_shift;
  const dim3 grid((unsigned)blocks);
  if (s->kh == 3) {
    if (pro) hipLaunchKernelGGL((dw_rows_bwd_fused_kernel<3, fused_wseg(3), true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((dw_rows_bwd_fused_kernel<3, fused_wseg(3), false>), grid, dim3(256), 0, st, a);
  } else {
    if (pro) hipLaunchKernelGGL((dw_rows_bwd_fused_kernel<5, fused_wseg(5), true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((dw_rows_bwd_fused_kernel<5, fused_wseg(5), false>), grid, dim3(256), 0, st, a);
  }
  if (hipGetLastError() != hipSuccess) { pb_set_error("dw_rows fused backward launch failed"); return 1; }
  return 0;
}
