// Depthwise (kt,3,3) convolution, kt in {3,5}, spatial stride 1 or 2, temporal stride 1 (MoViNet blocks:
// movinet.py:53 `nn.Conv3d(e, e, kernel_size, stride, padding, groups=e)`, CausalConv3d movinet.py:8-41).
// Same thread layout as dw_rows.hip - lanes along CHANNEL PAIRS, a thread owns two channels of a WSEG-wide
// segment of ONE output row - but the walk is along time and it is INPUT-stationary: every input frame's
// 3 x IW patch is loaded once (the next frame's is in flight meanwhile), converted to fp32 once, and
// multiplied into a ring of kt output accumulators (the outputs t = f + pt - dt it contributes to); the oldest
// output leaves the ring after every frame.  All ring/tap indices are static (frame loop unrolled by kt).
// No LDS, no cross-lane traffic.  The generic kernel this replaces re-read every input element kt*9 times
// through L1 with per-tap bounds arithmetic (0.13 TB/s).
#include "dw_internal.h"
#include <stdlib.h>

namespace {

struct Dw3Args {
  const void* src;
  void* dst;
  const float* w;          // [C][KT*9]
  const float* g_scale;
  const float* g_shift;
  float* pool;
  const void* hist;        // optional [N][nhist][Hs][Ws][C]: the nhist frames that precede src in time (stream buffer of a
  int nhist;               // CausalConv3d, movinet.py:18-21,36): input frames -nhist .. -1 are read from it, not as zeros
  int N, C, Ts, Hs, Ws, Td, Hd, Wd, pt, ph, pw;
  int nseg, CP, nchunk, chunk;     // chunk = output frames per thread
};

template <typename T, int KT, int S, int WSEG, bool FLIP>
__global__ __launch_bounds__(256) void dw3_fwd_kernel(Dw3Args a) {
  typedef typename Pair<T>::raw_t raw_t;
  constexpr int IW = (WSEG - 1) * S + 3;
  const int64_t gu = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int per_row = a.CP * a.nseg;
  const int64_t q = gu / per_row;                               // (n, chunk, output row)
  if (q >= (int64_t)a.N * a.nchunk * a.Hd) return;
  const int u = (int)(gu - q * per_row);
  const int sg = u / a.CP, cp = u - sg * a.CP;
  const int c0 = cp * 2;
  const int ho = (int)(q % a.Hd);
  const int nc = (int)(q / a.Hd);
  const int n = nc / a.nchunk, ck = nc - n * a.nchunk;
  const int t0 = ck * a.chunk;
  const int nT = min(a.Td, t0 + a.chunk) - t0;
  if (nT <= 0) return;                                          // (never: nchunk = ceil(Td / chunk))
  const int w0 = sg * WSEG;
  const int ps = a.C / 2;                                       // pixel stride in raw_t units
  // column / row offsets of the 3 x IW input patch (clamped; validity in the masks)
  unsigned cmask = 0, rmask = 0;
  int coloff[IW];
  int64_t rowoff[3];
#pragma unroll
  for (int j = 0; j < IW; ++j) {
    const int ws = w0 * S - a.pw + j;
    cmask |= (ws >= 0 && ws < a.Ws) ? (1u << j) : 0u;
    coloff[j] = min(max(ws, 0), a.Ws - 1) * ps;
  }
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int hs = ho * S - a.ph + kh;
    rmask |= (hs >= 0 && hs < a.Hs) ? (1u << kh) : 0u;
    rowoff[kh] = (int64_t)min(max(hs, 0), a.Hs - 1) * a.Ws * ps;
  }
  f32x2 wv[KT * 9];
#pragma unroll
  for (int i = 0; i < KT * 9; ++i) {
    const int tap = FLIP ? KT * 9 - 1 - i : i;
    wv[i] = f32x2{a.w[(int64_t)c0 * (KT * 9) + tap], a.w[(int64_t)(c0 + 1) * (KT * 9) + tap]};
  }
  f32x2 gs = {1.f, 1.f}, gh = {0.f, 0.f};
  if (a.g_scale) gs = *reinterpret_cast<const f32x2*>(a.g_scale + (int64_t)n * a.C + c0);
  if (a.g_shift) gh = *reinterpret_cast<const f32x2*>(a.g_shift + (int64_t)n * a.C + c0);
  const raw_t* src = (const raw_t*)a.src + ((int64_t)n * a.Ts * a.Hs * a.Ws * a.C + c0) / 2;
  // history frames sit right "before" frame 0: hist_base + f*fstride is frame f of the history for f in [-nhist, 0)
  const raw_t* hist_base = a.hist ? (const raw_t*)a.hist + ((int64_t)(n + 1) * a.nhist * a.Hs * a.Ws * a.C + c0) / 2 : nullptr;
  raw_t* dst = (raw_t*)a.dst + ((((int64_t)n * a.Td + t0) * a.Hd + ho) * a.Wd * a.C + c0) / 2 + (int64_t)w0 * ps;
  const int64_t fstride_s = (int64_t)a.Hs * a.Ws * ps, fstride_d = (int64_t)a.Hd * a.Wd * ps;

  f32x2 acc[KT][WSEG];
#pragma unroll
  for (int d = 0; d < KT; ++d)
#pragma unroll
    for (int j = 0; j < WSEG; ++j) acc[d][j] = f32x2{0.f, 0.f};
  // two frames in flight: raw = the frame of this step, raw2 = the next one (KT = 3, stride 1 has the registers for it)
  constexpr bool DEEP = KT == 3 && S == 1;
  raw_t raw[3][IW], raw2[DEEP ? 3 : 1][DEEP ? IW : 1];
  auto load_into = [&](int f, auto& dst) {
    const raw_t* base = (f < 0 && f >= -a.nhist) ? hist_base + (int64_t)f * fstride_s        // (nhist = 0 without a history)
                                                 : src + (int64_t)min(max(f, 0), a.Ts - 1) * fstride_s;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int j = 0; j < IW; ++j) dst[kh][j] = base[rowoff[kh] + coloff[j]];
  };
  auto load_frame = [&](int f) { load_into(f, raw); };
  const int f0 = t0 - a.pt;                                     // input frame of step 0
  const int nI = nT + KT - 1;
  load_frame(f0);
  if constexpr (DEEP) load_into(f0 + 1, raw2);
  f32x2 psum = {0.f, 0.f};
  for (int ik = 0; ik < nI; ik += KT) {
#pragma unroll
    for (int r = 0; r < KT; ++r) {
      const int i = ik + r;
      if (i < nI) {
        const int f = f0 + i;
        const bool fok = f >= -a.nhist && f < a.Ts;
        f32x2 xin[3][IW];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int j = 0; j < IW; ++j) {
            const bool ok = fok && ((rmask >> kh) & 1u) && ((cmask >> j) & 1u);
            xin[kh][j] = ok ? Pair<T>::up(raw[kh][j]) * gs + gh : f32x2{0.f, 0.f};
          }
        if constexpr (DEEP) {
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int j = 0; j < IW; ++j) raw[kh][j] = raw2[kh][j];
          if (i + 2 < nI) load_into(f + 2, raw2);
        } else {
          if (i + 1 < nI) load_frame(f + 1);
        }
        if (fok) {
#pragma unroll
          for (int dt = 0; dt < KT; ++dt) {
            const int lo = i - dt;                              // local output frame this tap plane feeds
            if (lo >= 0 && lo < nT) {
#pragma unroll
              for (int j = 0; j < WSEG; ++j) {
                f32x2 s = acc[(r - dt + KT) % KT][j];
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                  for (int kw = 0; kw < 3; ++kw) s += xin[kh][j * S + kw] * wv[dt * 9 + kh * 3 + kw];
                acc[(r - dt + KT) % KT][j] = s;
              }
            }
          }
        }
        const int lo = i - (KT - 1);                            // this output has now seen all its kt frames
        if (lo >= 0) {
          raw_t* orow = dst + (int64_t)lo * fstride_d;
#pragma unroll
          for (int j = 0; j < WSEG; ++j) {
            if (w0 + j < a.Wd) {
              const raw_t o = Pair<T>::down(acc[(r + 1) % KT][j]);
              orow[(int64_t)j * ps] = o;
              if (a.pool) psum += Pair<T>::up(o);
            }
            acc[(r + 1) % KT][j] = f32x2{0.f, 0.f};
          }
        }
      }
    }
  }
  // squeeze-excite partial sums [nchunk*Hd*nseg][N][C]: one slot per (frame chunk, output row, segment), plain stores
  if (a.pool)
    *reinterpret_cast<f32x2*>(a.pool + (((((int64_t)ck * a.Hd + ho) * a.nseg + sg) * a.N + n) * a.C + c0)) = psum;
}

bool dw3_ok(const pb_dwconv_shape* s) {
  static const bool enabled = !(getenv("PB_DW3") && atoi(getenv("PB_DW3")) == 0);
  return enabled && (s->kt == 3 || s->kt == 5) && s->kh == 3 && s->kw == 3 && s->st == 1 && s->sh == s->sw &&
         (s->sh == 1 || s->sh == 2) && s->C % 2 == 0 && s->C >= 16 && (int64_t)s->Hi * s->Wi * s->C < (1LL << 30);
}

constexpr int dw3_wseg(int KT, int S) { return (S == 2 && KT == 5) ? 3 : 4; }
void dw3_geom(Dw3Args& a, int KT, int WSEG) {
  a.CP = a.C / 2;
  a.nseg = (int)pb_cdiv(a.Wd, WSEG);
  const int64_t per_chunk = (int64_t)a.N * a.Hd * a.nseg * a.CP;
  // ~128K threads (the register budget admits 2 waves per SIMD); a chunk re-reads kt-1 input frames
  int nchunk = (int)pb_cdiv(128 * 1024, per_chunk);
  const int max_chunk = (int)pb_cdiv(a.Td, 2 * KT);
  if (nchunk > max_chunk) nchunk = max_chunk;
  if (nchunk < 1) nchunk = 1;
  a.chunk = (int)pb_cdiv(a.Td, nchunk);
  a.nchunk = (int)pb_cdiv(a.Td, a.chunk);
}

template <typename T, int KT, int S>
int launch_dw3(Dw3Args a, int flip, hipStream_t st) {
  constexpr int WSEG = dw3_wseg(KT, S);
  dw3_geom(a, KT, WSEG);
  const int64_t per_chunk = (int64_t)a.N * a.Hd * a.nseg * a.CP;
  const dim3 grid((unsigned)pb_cdiv(per_chunk * a.nchunk, 256));
  if (flip) hipLaunchKernelGGL((dw3_fwd_kernel<T, KT, S, WSEG, true>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((dw3_fwd_kernel<T, KT, S, WSEG, false>), grid, dim3(256), 0, st, a);
  if (hipGetLastError() != hipSuccess) { pb_set_error("dw3 launch failed"); return 1; }
  return 0;
}

}  // namespace

bool pb_dw3_eligible(const pb_dwconv_shape* s) { return dw3_ok(s); }

int pb_dw3_pool_slots(const pb_dwconv_shape* s) {
  Dw3Args a = {};
  a.N = s->N; a.C = s->C; a.Td = s->To; a.Hd = s->Ho; a.Wd = s->Wo;
  dw3_geom(a, s->kt, dw3_wseg(s->kt, s->sh));
  return a.nchunk * a.Hd * a.nseg;
}

int pb_dw3_fwd(const void* src, const float* w, void* dst, const pb_dwconv_shape* s, pb_dtype dtype, int flip,
               const float* g_scale, const float* g_shift, float* pool, hipStream_t st, const void* hist, int nhist) {
  if (!dw3_ok(s)) return -1;
  PB_CHECK_ARG((int64_t)s->N * s->To * s->Ho * s->Wo * s->C < (1LL << 40), "dw3: tensor too large");
  Dw3Args a = {};
  a.src = src; a.dst = dst; a.w = w; a.g_scale = g_scale; a.g_shift = g_shift; a.pool = pool;
  a.hist = hist; a.nhist = hist ? nhist : 0;
  a.N = s->N; a.C = s->C; a.Ts = s->Ti; a.Hs = s->Hi; a.Ws = s->Wi; a.Td = s->To; a.Hd = s->Ho; a.Wd = s->Wo;
  a.pt = s->pt; a.ph = s->ph; a.pw = s->pw;
  const int KT = s->kt, S = s->sh;
#define PB_D3(TT)                                                     \
  do {                                                                \
    if (KT == 3 && S == 1) return launch_dw3<TT, 3, 1>(a, flip, st);  \
    if (KT == 3 && S == 2) return launch_dw3<TT, 3, 2>(a, flip, st);  \
    if (KT == 5 && S == 1) return launch_dw3<TT, 5, 1>(a, flip, st);  \
    return launch_dw3<TT, 5, 2>(a, flip, st);                         \
  } while (0)
  if (dtype == PB_BF16) PB_D3(bf16_t);
  PB_D3(float);
#undef PB_D3
}
