// LDS-tiled depthwise (1,k,k) kernels for small frames (W <= ~32: the 28x28, 14x14 and 7x7 stages of
// MobileNetLarge3D/Small3D, mobilenet.py:67-75), where the register-strip kernels of dwconv.hip are
// latency-bound (few pixels per frame, k*k*CH accumulators per thread, redundant L1 loads).
//
// A workgroup owns a chunk of TCW channels and walks tiles = (clip, frame, band of R output rows)
// persistently.  Per tile every needed input element is fetched from HBM exactly once (16 B per lane,
// issued for the *next* tile before the current one is computed), converted to fp32 and committed to
// LDS together with the zero padding, so the compute phase is branch-free: threads read 16-byte
// channel quads from LDS and run packed fp32 FMAs (v_pk_fma_f32).
//   forward / backward-data (stride 1, flipped taps): thread = (channel quad, 4-wide output strip)
//   backward-weight: thread = (channel quad, filter row kh, share of the tile's strips); the k taps of
//     that row accumulate in registers over the whole persistent walk, then LDS + contiguous atomics.
#include "dw_internal.h"
#include <stdlib.h>

namespace {

constexpr int WT = 4;            // outputs per strip

struct TileArgs {
  const void* src;     // [N, Tsrc, Hs, Ws, C] input of the correlation
  const void* dstc;    // backward-weight: dy [N, Tdst, Hd, Wd, C]
  void* dst;           // forward: output [N, Tdst, Hd, Wd, C]
  const float* w;      // [C][K*K]
  float* dw;           // backward-weight: partial gradients [gridDim.x][C][K*K] (workspace)
  const float* g_scale;   // optional [N][C] prologue on src (fwd) / dy (wgrad): v*g_scale + g_shift
  const float* g_shift;
  int N, C, Tsrc, Hs, Ws, Tdst, Hd, Wd, pad, t_mul, t_off, flip;
  int R, row_tiles, Wdp, Wp, rows_x, nseg;
  int F, ftiles;       // frames per tile (tiny frames), ceil(Tdst / F)
  int64_t tiles;       // N * ftiles * row_tiles
};

template <typename T> __device__ __forceinline__ void to_f32v(const Vec16<T>& v, float* o) {
#pragma unroll
  for (int i = 0; i < ElemTraits<T>::VEC; ++i) o[i] = v.get(i);
}

// 4 channels of T from 4 floats
template <typename T> __device__ __forceinline__ void store_quad(T* p, const f32x4& v);
template <> __device__ __forceinline__ void store_quad<float>(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }
template <> __device__ __forceinline__ void store_quad<bf16_t>(bf16_t* p, const f32x4& v) {
  bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
  *reinterpret_cast<bf16x4*>(p) = o;
}

struct TileCoord {
  int n, td0, r0;
  int ft, rt;         // frame-group and row-band indices (for the division-free step to the next tile)
  bool any_ok;        // some frame of the tile reads src
};
__device__ __forceinline__ bool frame_reads(const TileArgs& a, int td) {
  const int ts = td * a.t_mul + a.t_off;
  return td < a.Tdst && ts >= 0 && ts < a.Tsrc;
}
__device__ __forceinline__ void tile_finish(const TileArgs& a, TileCoord& t) {
  t.td0 = t.ft * a.F;
  t.r0 = t.rt * a.R;
  t.any_ok = false;
  for (int f = 0; f < a.F; ++f) t.any_ok |= frame_reads(a, t.td0 + f);
}
__device__ __forceinline__ TileCoord tile_coord(const TileArgs& a, int it) {
  TileCoord t;
  const int zz = it / a.row_tiles;
  t.rt = it - zz * a.row_tiles;
  t.n = zz / a.ftiles;
  t.ft = zz - t.n * a.ftiles;
  tile_finish(a, t);
  return t;
}
// tile index + 1 (tiles are ordered clip-major, then frame group, then row band)
__device__ __forceinline__ void tile_next(const TileArgs& a, TileCoord& t) {
  if (++t.rt == a.row_tiles) {
    t.rt = 0;
    if (++t.ft == a.ftiles) {
      t.ft = 0;
      ++t.n;
    }
  }
  tile_finish(a, t);
}

// ----------------------------------------------------------------------------- staging
// Register image of one tile of an NDHWC tensor: NCH 16-byte chunks per thread.  Chunk i of thread tid
// sits at LDS float offset (tid + i*256) * VEC of a [frame][row][col][32 ch] block.  The position ->
// (frame slot, row, column) decode is the same for every tile, so it is done once (init); per tile only
// the frame/row validity and the address are evaluated.
constexpr int MAXF = 4;
template <typename T, int NCH>
struct Stager {
  Vec16<T> v[NCH];
  int rf[NCH];         // row | frame slot << 16
  int off[NCH];        // element offset from the tile's base pointer: slot*frame_step + row*row_stride + column*C +
                       // channel; -1: never valid (padding column / beyond the block)
  unsigned ok;
  // frame_step: elements between the frames of consecutive slots (t_mul * H * W * C for the src tensor)
  __device__ __forceinline__ void init(int tid, int W, int rows_per_frame, int F, int colshift, int Wsrc, int C, int c,
                                       bool cok, int frame_step, int row_stride) {
    constexpr int VEC = ElemTraits<T>::VEC, LPP = 32 / VEC;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int pix = (tid + i * 256) / LPP;
      const int vrow = pix / W, wcol = pix - vrow * W;
      const int f = vrow / rows_per_frame, r = vrow - f * rows_per_frame;
      const int ws = wcol - colshift;
      rf[i] = r | (f << 16);
      off[i] = (cok && f < F && ws >= 0 && ws < Wsrc) ? f * frame_step + r * row_stride + ws * C + c : -1;
    }
    ok = 0;
  }
  // base: address of (frame of slot 0, row `row_base`, column 0, channel 0) - may lie outside the tensor for
  // halo rows, those are never dereferenced; fmask: bit f set when slot f reads at all
  __device__ __forceinline__ void fetch(const T* base, int Hsrc, int row_base, unsigned fmask) {
    ok = 0;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int f = rf[i] >> 16, h = row_base + (rf[i] & 0xffff);
      const bool o = off[i] >= 0 && ((fmask >> f) & 1u) && (unsigned)h < (unsigned)Hsrc;
      v[i] = o ? vec_load<T>(base + off[i]) : vec_zero<T>();
      ok |= o ? (1u << i) : 0u;
    }
  }
  // fp32 commit; with `affine`: v*gs+gh on valid positions, exact zeros elsewhere
  __device__ __forceinline__ void commit(float* lds, int tid, int limit, const float* gs, const float* gh, bool affine) const {
    constexpr int VEC = ElemTraits<T>::VEC;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int pos = tid + i * 256;
      if (pos < limit) {
        float f[VEC];
        to_f32v<T>(v[i], f);
        if (affine) {
          const bool o = (ok >> i) & 1u;
#pragma unroll
          for (int e = 0; e < VEC; ++e) f[e] = o ? f[e] * gs[e] + gh[e] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < VEC / 4; ++q)
          *reinterpret_cast<f32x4*>(lds + pos * VEC + 4 * q) = f32x4{f[4 * q], f[4 * q + 1], f[4 * q + 2], f[4 * q + 3]};
      }
    }
  }
};

// bit f set when frame slot f of the tile exists and reads the source tensor
__device__ __forceinline__ unsigned frame_mask(const TileArgs& a, const TileCoord& t) {
  unsigned m = 0;
#pragma unroll
  for (int f = 0; f < MAXF; ++f)
    if (f < a.F && frame_reads(a, t.td0 + f)) m |= 1u << f;
  return m;
}

template <typename T>
__device__ __forceinline__ void load_gvec(const float* base, const TileArgs& a, int n, int c, bool cok, float fill, float* o) {
  constexpr int VEC = ElemTraits<T>::VEC;
#pragma unroll
  for (int e = 0; e < VEC; ++e) o[e] = fill;
  if (base && cok) {
#pragma unroll
    for (int q = 0; q < VEC / 4; ++q) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(base + (int64_t)n * a.C + c + 4 * q);
      o[4 * q] = t[0]; o[4 * q + 1] = t[1]; o[4 * q + 2] = t[2]; o[4 * q + 3] = t[3];
    }
  }
}

// Walk of a thread over the tile's strips (frame slot f, row r, segment ws), `step` strips at a time,
// without divisions in the loop (the decode of the start and of the step happens once).
struct StripWalk {
  int f, r, ws, df, dr, dws;
  __device__ __forceinline__ void init(int start, int step, int nseg, int R) {
    ws = start % nseg; const int t = start / nseg; r = t % R; f = t / R;
    dws = step % nseg; const int u = step / nseg; dr = u % R; df = u / R;
  }
  __device__ __forceinline__ void advance(int nseg, int R) {
    ws += dws;
    if (ws >= nseg) { ws -= nseg; ++r; }
    r += dr;
    if (r >= R) { r -= R; ++f; }
    f += df;
  }
};

// ----------------------------------------------------------------------------- forward
template <typename T, int K, int S>
__global__ __launch_bounds__(256) void dw_tile_fwd_kernel(TileArgs a) {
  constexpr int TCW = 32;
  constexpr int VEC = ElemTraits<T>::VEC, LPP = TCW / VEC;
  constexpr int XCH = 5 * (8 / VEC);
  constexpr int CQ = TCW / 4;                 // channel quads per chunk
  constexpr int SLOTS = 256 / CQ;
  constexpr int WIN = (WT - 1) * S + K;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xS = smem;                                   // [F][rows_x][Wp][TCW]
  float* wl = xS + a.F * a.rows_x * a.Wp * TCW;       // [K*K][TCW]
  const int tid = threadIdx.x;
  const int cbase = blockIdx.y * TCW;
  const int lq = tid % LPP, c_st = cbase + lq * VEC;
  const bool cok_st = c_st < a.C;
  const int cq = tid % CQ, slot = tid / CQ;
  const int c_cp = cbase + cq * 4;
  const bool cok_cp = c_cp < a.C;
  T* dst = (T*)a.dst;
  const T* src = (const T*)a.src;

  for (int i = tid; i < TCW * K * K; i += 256) {
    const int cl = i / (K * K), tap = i - cl * (K * K);
    const int kh = tap / K, kw = tap - kh * K;
    const int t2 = a.flip ? (K - 1 - kh) * K + (K - 1 - kw) : tap;
    wl[t2 * TCW + cl] = (cbase + cl < a.C) ? a.w[(int64_t)(cbase + cl) * (K * K) + tap] : 0.f;
  }
  const bool affine = a.g_scale || a.g_shift;
  const int xlimit = a.F * a.rows_x * a.Wp * LPP;
  const int64_t src_fs = (int64_t)a.Hs * a.Ws * a.C;
  const int src_rs = a.Ws * a.C;
  const int ntiles = (int)a.tiles;

  Stager<T, XCH> xr;
  xr.init(tid, a.Wp, a.rows_x, a.F, a.pad, a.Ws, a.C, c_st, cok_st, (int)(a.t_mul * src_fs), src_rs);
  float gs[VEC], gh[VEC];
  auto fetch = [&](const TileCoord& t) {
    const int hs_base = t.r0 * S - a.pad;
    const T* base = src + ((int64_t)t.n * a.Tsrc + (t.td0 * a.t_mul + a.t_off)) * src_fs + (int64_t)hs_base * src_rs;
    xr.fetch(base, a.Hs, hs_base, frame_mask(a, t));
    if (affine) {
      load_gvec<T>(a.g_scale, a, t.n, c_st, cok_st, 1.f, gs);
      load_gvec<T>(a.g_shift, a, t.n, c_st, cok_st, 0.f, gh);
    }
  };
  StripWalk w0;
  w0.init(slot, SLOTS, a.nseg, a.R);
  // each workgroup walks a contiguous range of tiles (stays within a clip: few pool flushes)
  const int per_wg = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  int it = blockIdx.x * per_wg;
  const int it_end = min(ntiles, it + per_wg);
  TileCoord tc;
  if (it < it_end) {
    tc = tile_coord(a, it);
    fetch(tc);
  }
  while (it < it_end) {
    const TileCoord cur = tc;
    // frame slots that read nothing (the reference's zero frames) are staged as zeros: their outputs
    // come out as zeros through the same arithmetic
    xr.commit(xS, tid, xlimit, gs, gh, affine);
    lds_barrier();
    const int nx = it + 1;
    if (nx < it_end) {
      tile_next(a, tc);
      fetch(tc);
    }
    const int rows_here = min(a.R, a.Hd - cur.r0);
    const int frames_here = min(a.F, a.Tdst - cur.td0);
    StripWalk sw = w0;
    for (; sw.f < frames_here; sw.advance(a.nseg, a.R)) {
      if (sw.r >= rows_here) continue;
      const int f = sw.f, r = sw.r, wd0 = sw.ws * WT;
      f32x4 acc[WT];
#pragma unroll
      for (int i = 0; i < WT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kh = 0; kh < K; ++kh) {
        const float* xp = xS + (((f * a.rows_x) + r * S + kh) * a.Wp + wd0 * S) * TCW + cq * 4;
        f32x4 xv[WIN];
#pragma unroll
        for (int i = 0; i < WIN; ++i) xv[i] = *reinterpret_cast<const f32x4*>(xp + i * TCW);
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(wl + (kh * K + kw) * TCW + cq * 4);
#pragma unroll
          for (int i = 0; i < WT; ++i) acc[i] += xv[i * S + kw] * wv;
        }
      }
      if (cok_cp) {
        const int64_t z = (int64_t)cur.n * a.Tdst + cur.td0 + f;
        T* o = dst + ((z * a.Hd + cur.r0 + r) * a.Wd + wd0) * a.C + c_cp;
#pragma unroll
        for (int i = 0; i < WT; ++i) {
          if (wd0 + i < a.Wd) store_quad<T>(o + (int64_t)i * a.C, acc[i]);
        }
      }
    }
    lds_barrier();
    it = nx;
  }
}

// ----------------------------------------------------------------------------- backward-weight
template <typename T, int K, int S>
__global__ __launch_bounds__(256) void dw_tile_wgrad_kernel(TileArgs a) {
  constexpr int TCW = 32;
  constexpr int VEC = ElemTraits<T>::VEC, LPP = TCW / VEC;
  constexpr int XCH = 5 * (8 / VEC);
  constexpr int GCH = 2 * (8 / VEC);
  constexpr int CQ = TCW / 4;
  constexpr int COMBOS = 256 / CQ;
  constexpr int J = COMBOS / K;
  constexpr int WIN = (WT - 1) * S + K;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xS = smem;                                   // [F][rows_x][Wp][TCW]
  float* gS = xS + a.F * a.rows_x * a.Wp * TCW;       // [F][R][Wdp][TCW]
  float* red = gS + a.F * a.R * a.Wdp * TCW;          // [K*K][TCW]
  const int tid = threadIdx.x;
  const int cbase = blockIdx.y * TCW;
  const int lq = tid % LPP, c_st = cbase + lq * VEC;
  const bool cok_st = c_st < a.C;
  const int cq = tid % CQ, combo = tid / CQ;
  const int kh = combo % K, j = combo / K;
  const T* dy = (const T*)a.dstc;
  const T* x = (const T*)a.src;
  for (int i = tid; i < K * K * TCW; i += 256) red[i] = 0.f;
  const bool affine = a.g_scale || a.g_shift;
  const int xlimit = a.F * a.rows_x * a.Wp * LPP, glimit = a.F * a.R * a.Wdp * LPP;
  const int64_t x_fs = (int64_t)a.Hs * a.Ws * a.C, g_fs = (int64_t)a.Hd * a.Wd * a.C;
  const int x_rs = a.Ws * a.C, g_rs = a.Wd * a.C;
  const int ntiles = (int)a.tiles;

  f32x4 acc[K];
#pragma unroll
  for (int kw = 0; kw < K; ++kw) acc[kw] = f32x4{0.f, 0.f, 0.f, 0.f};

  Stager<T, XCH> xr;
  Stager<T, GCH> gr;
  xr.init(tid, a.Wp, a.rows_x, a.F, a.pad, a.Ws, a.C, c_st, cok_st, (int)(a.t_mul * x_fs), x_rs);
  gr.init(tid, a.Wdp, a.R, a.F, 0, a.Wd, a.C, c_st, cok_st, (int)g_fs, g_rs);
  float gs[VEC], gh[VEC];
  auto fetch = [&](const TileCoord& t) {
    const unsigned fm = frame_mask(a, t);
    const int hs_base = t.r0 * S - a.pad;
    xr.fetch(x + ((int64_t)t.n * a.Tsrc + (t.td0 * a.t_mul + a.t_off)) * x_fs + (int64_t)hs_base * x_rs, a.Hs, hs_base, fm);
    gr.fetch(dy + ((int64_t)t.n * a.Tdst + t.td0) * g_fs + (int64_t)t.r0 * g_rs, a.Hd, t.r0, fm);
    if (affine) {
      load_gvec<T>(a.g_scale, a, t.n, c_st, cok_st, 1.f, gs);
      load_gvec<T>(a.g_shift, a, t.n, c_st, cok_st, 0.f, gh);
    }
  };
  // next tile with a frame that reads x at all (zero frames contribute nothing)
  // each workgroup walks a contiguous range of tiles (division-free stepping)
  const int per_wg = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int it_begin = min(ntiles, (int)blockIdx.x * per_wg), it_end = min(ntiles, it_begin + per_wg);
  auto skip_invalid = [&](int i, TileCoord& t) {
    while (i < it_end && !t.any_ok) {
      ++i;
      if (i < it_end) tile_next(a, t);
    }
    return i;
  };
  StripWalk w0;
  w0.init(j, J, a.nseg, a.R);
  TileCoord tc = tile_coord(a, min(it_begin, ntiles - 1));
  int it = skip_invalid(it_begin, tc);
  if (it < it_end) fetch(tc);
  while (it < it_end) {
    const TileCoord cur = tc;
    xr.commit(xS, tid, xlimit, nullptr, nullptr, false);
    gr.commit(gS, tid, glimit, gs, gh, affine);
    lds_barrier();
    int nx = it + 1;
    if (nx < it_end) {
      tile_next(a, tc);
      nx = skip_invalid(nx, tc);
    }
    if (nx < it_end) fetch(tc);
    if (j < J) {
      const int rows_here = min(a.R, a.Hd - cur.r0);
      StripWalk sw = w0;                              // frame slots beyond the clip are staged as zeros
      for (; sw.f < a.F; sw.advance(a.nseg, a.R)) {
        if (sw.r >= rows_here) continue;
        const float* gp = gS + ((sw.f * a.R + sw.r) * a.Wdp + sw.ws * WT) * TCW + cq * 4;
        const float* xp = xS + ((sw.f * a.rows_x + sw.r * S + kh) * a.Wp + sw.ws * WT * S) * TCW + cq * 4;
        f32x4 gv[WT], xv[WIN];
#pragma unroll
        for (int i = 0; i < WT; ++i) gv[i] = *reinterpret_cast<const f32x4*>(gp + i * TCW);
#pragma unroll
        for (int i = 0; i < WIN; ++i) xv[i] = *reinterpret_cast<const f32x4*>(xp + i * TCW);
#pragma unroll
        for (int i = 0; i < WT; ++i)
#pragma unroll
          for (int kw = 0; kw < K; ++kw) acc[kw] += gv[i] * xv[i * S + kw];
      }
    }
    lds_barrier();
    it = nx;
  }
  // threads that share (channel quad, kh) differ in j: they add their taps in J rounds (fixed order, no atomics)
  __syncthreads();
  for (int r = 0; r < J; ++r) {
    if (j == r) {
#pragma unroll
      for (int kw = 0; kw < K; ++kw)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[(kh * K + kw) * TCW + cq * 4 + e] += acc[kw][e];
    }
    __syncthreads();
  }
  // the chunk's [channel][tap] block is contiguous in this workgroup's partial gradient
  float* part = a.dw + (int64_t)blockIdx.x * a.C * (K * K) + (int64_t)cbase * (K * K);
  for (int i = tid; i < K * K * TCW; i += 256) {
    const int cl = i / (K * K), tap = i - cl * (K * K);
    if (cbase + cl < a.C) part[i] = red[tap * TCW + cl];
  }
}

// ----------------------------------------------------------------------------- host side
struct TilePlan {
  int R, row_tiles, Wdp, Wp, rows_x, nseg, F, ftiles;
  size_t lds_floats;
};

constexpr int XPIX = 320;        // staged src pixels per tile  (XCH = 5 chunks per thread)
constexpr int GPIX = 128;        // staged dy pixels per tile   (GCH = 2)

// Rows per band (balanced, no ragged last band) and, when a whole frame fits, frames per tile, so that
// the staged tile fits the per-thread register image and the strips fill the workgroup's slots.
bool plan_tile(int K, int S, int Tdst, int Hd, int Wd, bool wgrad, TilePlan& p) {
  p.Wdp = (int)pb_cdiv(Wd, WT) * WT;
  p.nseg = p.Wdp / WT;
  p.Wp = (p.Wdp - 1) * S + K;
  int rmax = 0;
  for (int r = 1; r <= Hd; ++r) {
    if (((r - 1) * S + K) * p.Wp > XPIX) break;
    if (wgrad && r * p.Wdp > GPIX) break;
    rmax = r;
  }
  if (rmax == 0) return false;
  const int bands = (int)pb_cdiv(Hd, rmax);
  p.R = (int)pb_cdiv(Hd, bands);
  p.row_tiles = (int)pb_cdiv(Hd, p.R);
  p.rows_x = (p.R - 1) * S + K;
  // halo rows are re-read (from L2) once per band: not worth it when a band is thinner than the halo
  if (p.row_tiles > 1 && 2 * p.R * S < K) return false;
  p.F = 1;
  if (p.row_tiles == 1) {
    while (p.F < 4 && p.F < Tdst && (p.F + 1) * p.rows_x * p.Wp <= XPIX && (!wgrad || (p.F + 1) * p.R * p.Wdp <= GPIX) &&
           p.F * p.R * p.nseg < 32)
      ++p.F;
  }
  p.ftiles = (int)pb_cdiv(Tdst, p.F);
  p.lds_floats = (size_t)p.F * p.rows_x * p.Wp * 32 + (wgrad ? (size_t)p.F * p.R * p.Wdp * 32 : 0) + (size_t)K * K * 32 +
                 0;
  return true;
}

// Occupancy queries cost tens of microseconds on the host: remember the answer per (kernel, LDS size, device).  Every
// dw_tile instantiation has the same function type, so the kernel's address is part of the key; the grid is a function
// of (kernel, shape, device) only - never of the call history (the slots a launch writes must be reproducible).
int persistent_gx(const void* kernel, size_t lds_bytes, int chunks, int64_t tiles, int device) {
  struct Memo { const void* kernel; size_t lds; int device, per_cu; };
  static thread_local Memo memo[16];
  static thread_local int nmemo = 0, cursor = 0;
  static thread_local int cus[16] = {0};
  int per_cu = 0;
  for (int i = 0; i < nmemo; ++i)
    if (memo[i].kernel == kernel && memo[i].lds == lds_bytes && memo[i].device == device) per_cu = memo[i].per_cu;
  if (per_cu == 0) {
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, lds_bytes) != hipSuccess || per_cu < 1) per_cu = 1;
    memo[cursor] = Memo{kernel, lds_bytes, device, per_cu};
    cursor = (cursor + 1) % 16;
    if (nmemo < 16) ++nmemo;
  }
  int& ncu = cus[device & 15];
  if (ncu == 0 && hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) ncu = 256;
  int64_t gx = (int64_t)per_cu * ncu / chunks;
  gx -= gx % 8;                  // chunks of one tile index land on the same XCD (ids 8k apart)
  if (gx < 8) gx = 8;
  if (gx > tiles) gx = tiles;
  return (int)gx;
}

// launch == false: only compute the grid (= the number of partial-gradient slots of the weight gradient)
template <typename T, int K, int S, bool WGRAD>
int launch_tile(const TileArgs& a, const TilePlan& p, hipStream_t st, bool launch, int64_t* slots) {
  const size_t lds = p.lds_floats * sizeof(float);
  const int chunks = (int)pb_cdiv(a.C, 32);
  int device = 0;
  (void)hipGetDevice(&device);
  if constexpr (WGRAD) {
    auto kern = dw_tile_wgrad_kernel<T, K, S>;
    const int gx = persistent_gx((const void*)kern, lds, chunks, a.tiles, device);
    if (slots) *slots = gx;
    if (launch) hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)chunks), dim3(256), lds, st, a);
  } else {
    auto kern = dw_tile_fwd_kernel<T, K, S>;
    const int gx = persistent_gx((const void*)kern, lds, chunks, a.tiles, device);
    if (launch) hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)chunks), dim3(256), lds, st, a);
  }
  if (launch && hipGetLastError() != hipSuccess) {
    pb_set_error("dw_tile launch failed (lds %zu bytes)", lds);
    return 1;
  }
  return 0;
}

template <typename T, bool WGRAD>
int dispatch_tile(const TileArgs& a, const TilePlan& p, int K, int S, hipStream_t st, bool launch = true, int64_t* slots = nullptr) {
  if (K == 3 && S == 1) return launch_tile<T, 3, 1, WGRAD>(a, p, st, launch, slots);
  if (K == 3 && S == 2) return launch_tile<T, 3, 2, WGRAD>(a, p, st, launch, slots);
  if (K == 5 && S == 1) return launch_tile<T, 5, 1, WGRAD>(a, p, st, launch, slots);
  if (K == 5 && S == 2) return launch_tile<T, 5, 2, WGRAD>(a, p, st, launch, slots);
  return -1;
}

bool tile_family_ok(const pb_dwconv_shape* s) {
  return s->kt == 1 && s->kh == s->kw && (s->kh == 3 || s->kh == 5) && s->sh == s->sw && (s->sh == 1 || s->sh == 2) &&
         s->ph == s->pw;
}

void fill_plan(TileArgs& a, const TilePlan& p) {
  a.R = p.R; a.row_tiles = p.row_tiles; a.Wdp = p.Wdp; a.Wp = p.Wp; a.rows_x = p.rows_x; a.nseg = p.nseg;
  a.F = p.F; a.ftiles = p.ftiles;
  a.tiles = (int64_t)a.N * p.ftiles * p.row_tiles;
}

}  // namespace

bool pb_dw_tile_eligible(const pb_dwconv_shape* s, pb_dtype dtype, bool wgrad) {
  static const bool enabled = !(getenv("PB_DW_TILE") && atoi(getenv("PB_DW_TILE")) == 0);
  const int vec = dtype == PB_BF16 ? 8 : 4;
  if (!enabled || !tile_family_ok(s) || s->C % vec != 0) return false;
  // measured on MI355X (tools/kbench.py dw): the forward-type tile kernel wins for k=3 (14x14: 1.2-1.4x over
  // the strip kernel) and loses for k=5 (25 taps x 4 outputs per strip: 180 VGPRs, two waves per SIMD);
  // the weight-gradient tile kernel wins wherever its plan fits (1.3-1.9x)
  static const bool fwd_k5 = getenv("PB_DW_TILE_FWD_K5") && atoi(getenv("PB_DW_TILE_FWD_K5")) != 0;
  if (!wgrad && s->kh == 5 && !fwd_k5) return false;
  TilePlan p;
  return plan_tile(s->kh, s->sh, s->To, s->Ho, s->Wo, wgrad, p);
}

namespace {
int tile_wgrad_run(const void* dy, const void* x, float* part, const pb_dwconv_shape* s, pb_dtype dtype,
                   const float* g_scale, const float* g_shift, hipStream_t st, bool launch, int64_t* slots) {
  if (!pb_dw_tile_eligible(s, dtype, true)) return -1;
  TilePlan p;
  plan_tile(s->kh, s->sh, s->To, s->Ho, s->Wo, true, p);
  TileArgs a = {};
  a.src = x; a.dstc = dy; a.dw = part; a.g_scale = g_scale; a.g_shift = g_shift;
  a.N = s->N; a.C = s->C; a.Tsrc = s->Ti; a.Hs = s->Hi; a.Ws = s->Wi; a.Tdst = s->To; a.Hd = s->Ho; a.Wd = s->Wo;
  a.pad = s->ph; a.t_mul = s->st; a.t_off = -s->pt;
  fill_plan(a, p);
  return dtype == PB_BF16 ? dispatch_tile<bf16_t, true>(a, p, s->kh, s->sh, st, launch, slots)
                          : dispatch_tile<float, true>(a, p, s->kh, s->sh, st, launch, slots);
}
}  // namespace

// number of partial gradients [C][k*k] the launch writes (0: shape not taken)
int64_t pb_dw_tile_wgrad_slots(const pb_dwconv_shape* s, pb_dtype dtype) {
  int64_t slots = 0;
  return tile_wgrad_run(nullptr, nullptr, nullptr, s, dtype, nullptr, nullptr, nullptr, false, &slots) == 0 ? slots : 0;
}

int pb_dw_tile_wgrad(const void* dy, const void* x, float* part, const pb_dwconv_shape* s, pb_dtype dtype,
                     const float* g_scale, const float* g_shift, hipStream_t st, int64_t* slots) {
  return tile_wgrad_run(dy, x, part, s, dtype, g_scale, g_shift, st, true, slots);
}

// Forward-type correlation dst = corr(src, w).  `s` describes it directly: Ti/Hi/Wi = src extent,
// To/Ho/Wo = dst extent, kh/sh, ph = spatial padding seen from dst, frame map ts = td*s->st - s->pt.
int pb_dw_tile_fwd(const void* src, const float* w, void* dst, const pb_dwconv_shape* s, pb_dtype dtype, int flip,
                   const float* g_scale, const float* g_shift, hipStream_t st) {
  if (!pb_dw_tile_eligible(s, dtype, false)) return -1;
  TilePlan p;
  plan_tile(s->kh, s->sh, s->To, s->Ho, s->Wo, false, p);
  TileArgs a = {};
  a.src = src; a.w = w; a.dst = dst; a.g_scale = g_scale; a.g_shift = g_shift; a.flip = flip;
  a.N = s->N; a.C = s->C; a.Tsrc = s->Ti; a.Hs = s->Hi; a.Ws = s->Wi; a.Tdst = s->To; a.Hd = s->Ho; a.Wd = s->Wo;
  a.pad = s->ph; a.t_mul = s->st; a.t_off = -s->pt;
  fill_plan(a, p);
  return dtype == PB_BF16 ? dispatch_tile<bf16_t, false>(a, p, s->kh, s->sh, st)
                          : dispatch_tile<float, false>(a, p, s->kh, s->sh, st);
}
