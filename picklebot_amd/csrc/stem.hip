// Dense 3-D convolution with a tiny input-channel count (the RGB stems).  The clip is read in
// place through element strides (NDHWC view of the dataloader tensor, or NCDHW), optionally
// straight from uint8 with the /255 folded in; the output is NDHWC in the compute dtype.
// K = Cin*kt*kh*kw <= 127 and Cout = 16: far too small for MFMA tiles to matter - the kernel is
// bound by reading the clip and writing the 16-channel map, so it is a plain VALU stencil with
// the (wave-uniform) filter broadcast from LDS.
#include "pb_common.h"
#include <stdlib.h>

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
  int ablate;      // (experiments only: PB_STEM_ABLATE)
};

template <typename XT> __device__ __forceinline__ float ldx(const XT* p) { return (float)*p; }
template <> __device__ __forceinline__ float ldx<bf16_t>(const bf16_t* p) { return (float)*p; }

template <typename XT, typename T>
__global__ __launch_bounds__(256) void stem_fwd_kernel(StemArgs a) {
  __shared__ __attribute__((aligned(16))) float wl[MAX_CT * OCB];
  __shared__ float red[2 * OCB * 256];       // statistics: one column per thread, summed in thread order
  const pb_stem_shape& s = a.s;
  const int tid = threadIdx.x;
  const int taps = s.kt * s.kh * s.kw, CT = s.Cin * taps;
  const int oc0 = blockIdx.y * OCB;
  for (int i = tid; i < CT * OCB; i += 256) {
    const int ct = i / OCB, oc = i - ct * OCB;
    wl[i] = (oc0 + oc < s.Cout) ? a.w[(int64_t)(oc0 + oc) * CT + ct] : 0.f;
  }
  __syncthreads();
  float ssum[OCB], ssq[OCB];
#pragma unroll
  for (int oc = 0; oc < OCB; ++oc) ssum[oc] = ssq[oc] = 0.f;
  const int64_t nblk = (a.npix + 255) / 256;
  for (int64_t pb = blockIdx.x; pb < nblk; pb += gridDim.x) {
  const int64_t pix = pb * 256 + tid;
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
#pragma unroll
    for (int oc = 0; oc < OCB; ++oc) {
      const float v = to_f32(from_f32<T>(acc[oc]));
      ssum[oc] += v;
      ssq[oc] += v * v;
    }
  }
  }
  if (a.stats) {
#pragma unroll
    for (int oc = 0; oc < OCB; ++oc) {
      red[oc * 256 + tid] = ssum[oc];
      red[(OCB + oc) * 256 + tid] = ssq[oc];
    }
    __syncthreads();
    if (tid < 2 * OCB) {
      const float* src = red + tid * 256;
      float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
      for (int i = 0; i < 256; i += 4) { t0 += src[i]; t1 += src[i + 1]; t2 += src[i + 2]; t3 += src[i + 3]; }
      const int oc = tid & (OCB - 1);
      if (oc0 + oc < s.Cout)      // this workgroup's own slot (gridDim.x <= PB_STAT_SLOTS)
        a.stats[(size_t)blockIdx.x * 2 * s.Cout + (tid >= OCB ? s.Cout : 0) + oc0 + oc] = (t0 + t1) + (t2 + t3);
    }
  }
}

// Input window of SW_PIX consecutive output pixels of one output row, as [at][ah][wl][ci] in LDS
// (zero outside the clip): row by row, lanes along the (pixel, channel) run - contiguous in memory for
// the NDHWC clip, still correct for any strides.
constexpr int FW_R = 9;      // max kt*kh window rows
constexpr int FW_I = 4;      // max 64-lane sweeps per row (row length <= 256 elements)

template <typename XT, typename T>
__device__ __forceinline__ void fill_window(T* win, const XT* x, float x_scale, const pb_stem_shape& s, int n, int to,
                                            int ho, int wo0, int WL, int lane) {
  // two passes: every (clamped, unconditional) global load is issued first, only then converted and
  // written to LDS - a load->use pair inside the sweep would cost one HBM round trip per iteration
  const int ti0 = to * s.st - s.pt, hi0 = ho * s.sh - s.ph, wi0 = wo0 * s.sw - s.pw;
  const int rowlen = WL * s.Cin, nrows = s.kt * s.kh;
  const XT* xn = x + (int64_t)n * s.xs_n;
  XT raw[FW_R][FW_I];
  bool okm[FW_R][FW_I];
#pragma unroll
  for (int rr = 0; rr < FW_R; ++rr) {
    const int rc = min(rr, nrows - 1);
    const int at = rc / s.kh, ah = rc - at * s.kh;
    const int ti = ti0 + at, hi = hi0 + ah;
    const bool row_ok = rr < nrows && ti >= 0 && ti < s.Ti && hi >= 0 && hi < s.Hi;
    const XT* base = xn + (int64_t)min(max(ti, 0), s.Ti - 1) * s.xs_t + (int64_t)min(max(hi, 0), s.Hi - 1) * s.xs_h;
#pragma unroll
    for (int it = 0; it < FW_I; ++it) {
      const int e = min(lane + 64 * it, rowlen - 1);
      const int wl = e / s.Cin, ci = e - wl * s.Cin;
      const int wi = wi0 + wl;
      okm[rr][it] = row_ok && wi >= 0 && wi < s.Wi;
      raw[rr][it] = base[(int64_t)min(max(wi, 0), s.Wi - 1) * s.xs_w + (int64_t)ci * s.xs_c];
    }
  }
#pragma unroll
  for (int rr = 0; rr < FW_R; ++rr)
#pragma unroll
    for (int it = 0; it < FW_I; ++it) {
      const int e = lane + 64 * it;
      if (rr < nrows && e < rowlen) win[rr * rowlen + e] = from_f32<T>(okm[rr][it] ? (float)raw[rr][it] * x_scale : 0.f);
    }
}

constexpr int SW_PIX = 32;          // pixels per wave step
constexpr int SW_MAXWIN = 2048;     // LDS window elements per wave (kt*kh*(31*sw+kw)*Cin)
constexpr int SW_NB = 8;            // column blocks of 16 (Cin*taps + 1 <= 128)
constexpr int SW_KS = 4;            // 32-wide k steps of the forward GEMM (Cin*taps <= 128)

// Forward as an MFMA GEMM over the same LDS window: out[pixel][oc] = sum_col patch[pixel][col] * W[oc][col].
// A = im2col gathers from the window (16 pixels x 32 columns per MFMA), B = the filter (kept in
// registers for the whole walk), 32 pixels per wave step.  The [32][16] result goes through a
// wave-private LDS tile so that the store is one contiguous 16 B/lane run; BatchNorm statistics are
// accumulated in registers over the walk.
template <typename XT, typename T>
__global__ __launch_bounds__(256) void stem_fwd_mfma_kernel(StemArgs a) {
  __shared__ __attribute__((aligned(16))) T win_s[4][SW_MAXWIN];
  __shared__ __attribute__((aligned(16))) T out_s[4][SW_PIX * OCB];
  __shared__ float red[4][2 * OCB];
  const pb_stem_shape& s = a.s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int taps = s.kt * s.kh * s.kw, CT = s.Cin * taps;
  const int nks = (CT + 31) / 32;
  const int oc0 = blockIdx.y * OCB;
  const int WL = (SW_PIX - 1) * s.sw + s.kw;
  const int wblocks = (s.Wo + SW_PIX - 1) / SW_PIX;
  const int64_t nitems = (int64_t)s.N * s.To * s.Ho * wblocks;
  const XT* x = (const XT*)a.x;
  T* y = (T*)a.y;
  T* win = win_s[wave];
  T* outt = out_s[wave];
  const int row = lane & 15, g = lane >> 4;          // A: pixel row / k group; B: oc / k group; D: oc column, row group
  // filter fragments: lane (oc = row, g) holds W[oc][32*ks + 8g + j]; window offsets of this lane's 8 columns
  float wf[SW_KS][8];
  int coff[SW_KS][8];
#pragma unroll
  for (int ks = 0; ks < SW_KS; ++ks)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = ks * 32 + g * 8 + j;
      const bool cv = c < CT && ks < nks;
      wf[ks][j] = (cv && oc0 + row < s.Cout) ? a.w[(int64_t)(oc0 + row) * CT + c] : 0.f;
      const int ci = cv ? c / taps : 0;
      const int tp = cv ? c - ci * taps : 0;
      const int aw = tp % s.kw, ah = (tp / s.kw) % s.kh, at = tp / (s.kw * s.kh);
      coff[ks][j] = cv ? ((at * s.kh + ah) * WL + aw) * s.Cin + ci : -1;
    }
  const float bias = (a.bias && oc0 + row < s.Cout) ? a.bias[oc0 + row] : 0.f;
  float ssum = 0.f, ssq = 0.f;

  for (int64_t it = (int64_t)blockIdx.x * 4 + wave; it < nitems; it += (int64_t)gridDim.x * 4) {
    int64_t r = it;
    const int wb = (int)(r % wblocks); r /= wblocks;
    const int ho = (int)(r % s.Ho); r /= s.Ho;
    const int to = (int)(r % s.To);
    const int n = (int)(r / s.To);
    const int wo0 = wb * SW_PIX;
    const int npx = min(SW_PIX, s.Wo - wo0);
    fill_window<XT, T>(win, x, a.x_scale, s, n, to, ho, wo0, WL, lane);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    f32x4 acc[2];
    acc[0] = acc[1] = f32x4{bias, bias, bias, bias};
#pragma unroll
    for (int ks = 0; ks < SW_KS; ++ks) {
      if (ks < nks) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int pbase = (mt * 16 + row) * s.sw * s.Cin;
          if constexpr (sizeof(T) == 2) {
            bf16x8 fa, fb;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              fa[j] = coff[ks][j] >= 0 ? win[coff[ks][j] + pbase] : (bf16_t)0.f;
              fb[j] = (bf16_t)wf[ks][j];
            }
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[mt], 0, 0, 0);
          } else {
            // exact-f32 16x16x4: lane (row, g) supplies k = 4g + s per step; 8 steps cover this lane group's
            // 32 columns in the order (j = 2s' + ...): use columns 8g'+j with g' = k group of the step
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              // step j: lane group g contributes column 32*ks + 8g + j  (any consistent A/B pairing is valid)
              const float fa = coff[ks][j] >= 0 ? win[coff[ks][j] + pbase] : 0.f;
              acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, wf[ks][j], acc[mt], 0, 0, 0);
            }
          }
        }
      }
    }
    // D map: column = oc = lane & 15, rows = pixels (lane>>4)*4 + r
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int p = mt * 16 + g * 4 + rr;
        const T tv = from_f32<T>(acc[mt][rr]);
        outt[p * OCB + row] = tv;
        if (a.stats && p < npx && oc0 + row < s.Cout) {
          const float fv = to_f32(tv);
          ssum += fv;
          ssq += fv * fv;
        }
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    {
      const int64_t pix0 = (((int64_t)n * s.To + to) * s.Ho + ho) * s.Wo + wo0;
      constexpr int VEC = ElemTraits<T>::VEC;
      if (oc0 + OCB <= s.Cout && s.Cout == OCB) {
        // the 32x16 tile is one contiguous run of the output
        for (int e = lane; e < SW_PIX * OCB / VEC; e += 64) {
          const int p = e * VEC / OCB;
          if (p < npx) vec_store<T>(y + pix0 * OCB + (int64_t)e * VEC, vec_load<T>(outt + e * VEC));
        }
      } else {
        for (int e = lane; e < SW_PIX * OCB; e += 64) {
          const int p = e / OCB, oc = e - p * OCB;
          if (p < npx && oc0 + oc < s.Cout) y[(pix0 + p) * s.Cout + oc0 + oc] = outt[e];
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
  if (a.stats) {
    ssum += __shfl_xor(ssum, 16, 64); ssum += __shfl_xor(ssum, 32, 64);
    ssq += __shfl_xor(ssq, 16, 64); ssq += __shfl_xor(ssq, 32, 64);
    if (lane < 16) {
      red[wave][lane] = ssum;
      red[wave][OCB + lane] = ssq;
    }
    __syncthreads();
    if (tid < OCB && oc0 + tid < s.Cout) {
      float* st = a.stats + (size_t)blockIdx.x * 2 * s.Cout;        // this workgroup's own slot (gridDim.x <= PB_STAT_SLOTS)
      st[oc0 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
      st[s.Cout + oc0 + tid] = (red[0][OCB + tid] + red[1][OCB + tid]) + (red[2][OCB + tid] + red[3][OCB + tid]);
    }
  }
}

// Weight gradient as a small MFMA GEMM: dW[oc][col] = sum_pix dy[pix][oc] * patch[pix][col],
// col = (ci, tap) (+ one virtual column of ones for the bias).  Each wave walks blocks of 32
// consecutive output pixels of one output row: dy's [32][16] tile is one contiguous 1 KB read,
// the input window those pixels see (kt x kh rows of 31*sw+kw pixels) is staged once in LDS, and
// the MFMA "k" index is the pixel: A = dy^T (transposing LDS read for bf16), B = im2col gathers
// from the LDS window.  Accumulators persist over the wave's whole walk.
template <typename XT, typename T>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(StemArgs a) {
  __shared__ __attribute__((aligned(16))) T win_s[4][SW_MAXWIN];
  __shared__ __attribute__((aligned(16))) T dy_s[4][SW_PIX * OCB];
  __shared__ float red[OCB * SW_NB * 16];
  const pb_stem_shape& s = a.s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int taps = s.kt * s.kh * s.kw, CT = s.Cin * taps;
  const int nb = (CT + 1 + 15) / 16;
  const int oc0 = blockIdx.y * OCB;
  const int WL = (SW_PIX - 1) * s.sw + s.kw;             // window width in input pixels
  const int win_elems = s.kt * s.kh * WL * s.Cin;
  const int wblocks = (s.Wo + SW_PIX - 1) / SW_PIX;
  const int64_t nitems = (int64_t)s.N * s.To * s.Ho * wblocks;
  const T* dy = (const T*)a.y;
  const XT* x = (const XT*)a.x;
  T* win = win_s[wave];
  T* dyt = dy_s[wave];

  // this lane's im2col column for every column block: offset of (ci, at, ah, aw) inside the window
  const int col = lane & 15, g = lane >> 4;
  int coff[SW_NB];
  bool cvalid[SW_NB], cbias[SW_NB];
#pragma unroll
  for (int b = 0; b < SW_NB; ++b) {
    const int c = b * 16 + col;
    cvalid[b] = c < CT;
    cbias[b] = c == CT;
    const int ci = cvalid[b] ? c / taps : 0;
    const int tp = cvalid[b] ? c - ci * taps : 0;
    const int aw = tp % s.kw, ah = (tp / s.kw) % s.kh, at = tp / (s.kw * s.kh);
    coff[b] = ((at * s.kh + ah) * WL + aw) * s.Cin + ci;
  }
  f32x4 acc[SW_NB];
#pragma unroll
  for (int b = 0; b < SW_NB; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int64_t it = (int64_t)blockIdx.x * 4 + wave; it < nitems; it += (int64_t)gridDim.x * 4) {
    int64_t r = it;
    const int wb = (int)(r % wblocks); r /= wblocks;
    const int ho = (int)(r % s.Ho); r /= s.Ho;
    const int to = (int)(r % s.To);
    const int n = (int)(r / s.To);
    const int wo0 = wb * SW_PIX;
    const int npx = min(SW_PIX, s.Wo - wo0);
    // dy tile [32][16] (rows past the row end are zero: they then contribute nothing)
    {
      const int64_t pix0 = (((int64_t)n * s.To + to) * s.Ho + ho) * s.Wo + wo0;
      constexpr int VEC = ElemTraits<T>::VEC;
      if (s.Cout == OCB) {
        // the [32][16] tile is one contiguous run: 16 B per lane
        for (int e = lane; e < SW_PIX * OCB / VEC; e += 64) {
          const int p = e * VEC / OCB;
          vec_store<T>(dyt + e * VEC, p < npx ? vec_load<T>(dy + pix0 * OCB + (int64_t)e * VEC) : vec_zero<T>());
        }
      } else {
        for (int e = lane; e < SW_PIX * OCB; e += 64) {
          const int p = e / OCB, oc = e - p * OCB;
          dyt[e] = (p < npx && oc0 + oc < s.Cout) ? dy[(pix0 + p) * s.Cout + oc0 + oc] : from_f32<T>(0.f);
        }
      }
    }
    fill_window<XT, T>(win, x, a.x_scale, s, n, to, ho, wo0, WL, lane);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if constexpr (sizeof(T) == 2) {
      // A = dy^T [oc][pixel]: lane (oc = lane&15, g) needs pixels 8g..8g+7 -> two transposing reads
      const int q = (lane & 15) >> 2, p4 = lane & 3;
      typedef __attribute__((address_space(3))) bf16x4* lds_p;
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(dyt + (8 * g + q) * OCB + 4 * p4));
      const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(dyt + (8 * g + 4 + q) * OCB + 4 * p4));
      const bf16x8 fa = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
      for (int b = 0; b < SW_NB; ++b) {
        if (b < nb) {
          bf16x8 fb;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int p = 8 * g + j;
            fb[j] = cvalid[b] ? win[coff[b] + p * s.sw * s.Cin] : (cbias[b] ? (bf16_t)1.f : (bf16_t)0.f);
          }
          acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[b], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int ms = 0; ms < 8; ++ms) {
        const int p = ms * 4 + g;
        const float fa = dyt[p * OCB + col];
#pragma unroll
        for (int b = 0; b < SW_NB; ++b) {
          if (b < nb) {
            const float fb = cvalid[b] ? win[coff[b] + p * s.sw * s.Cin] : (cbias[b] ? 1.f : 0.f);
            acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[b], 0, 0, 0);
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
  // cross-wave sum in wave order (C/D map: row = oc = (lane>>4)*4 + r, col = lane & 15; every lane owns its elements
  // of the wave's tile), then plain stores into this workgroup's slot of the partial-gradient workspace
  for (int i = tid; i < OCB * SW_NB * 16; i += 256) red[i] = 0.f;
  __syncthreads();
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int b = 0; b < SW_NB; ++b)
        if (b < nb)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[((lane >> 4) * 4 + r) * (SW_NB * 16) + b * 16 + (lane & 15)] += acc[b][r];
    }
    __syncthreads();
  }
  float* part = a.dw + (size_t)blockIdx.x * ((size_t)s.Cout * (CT + 1));     // [Cout][CT] then [Cout] (bias)
  for (int i = tid; i < OCB * (CT + 1); i += 256) {
    const int oc = i / (CT + 1), c = i - oc * (CT + 1);
    if (oc0 + oc >= s.Cout) continue;
    const float v = red[oc * (SW_NB * 16) + c];
    if (c < CT) part[(int64_t)(oc0 + oc) * CT + c] = v;
    else part[(int64_t)s.Cout * CT + oc0 + oc] = v;
  }
}

// ----------------------------------------------------------------------------- 3-channel / stride-2 / kw=3 fast path
// The reference stem (Conv3d(3, 16, 3, stride 2, padding 1) on an NDHWC bf16 clip, train.py:106).  An output pixel
// wo reads input pixels 2wo-1 .. 2wo+1: per window row that is 9 contiguous bf16.  The generic kernels above gather
// them element by element (8 ds_read_u16 + packing per MFMA fragment, 36 two-byte global loads per lane and item).
// Here a window row is staged as 16-byte SLOTS, slot p = the 6 elements of input pixels (2p-1, 2p) + [one, zero]:
//   * fill: one (unaligned) 12-byte global load + one ds_write_b128 per slot;
//   * the 9 taps of output pixel p are slot p's elements 0..5 and slot p+1's elements 0..2, so the im2col matrix
//     [pixel][8] of a (row, shift) pair IS a slot row: forward A fragments are one ds_read_b128 each, weight-
//     gradient B fragments two transposing reads, in a K order of 18 groups of 8 (9 window rows x {shift 0, 1});
//   * junk positions (elements 6..7 of shift 0 = the constant [1, 0]; elements 3..7 of shift 1 = the next pixel's
//     data) meet zero weights in the forward and are never stored by the weight gradient - except element 6 of
//     window row 0 (the ones), which carries the bias.
constexpr int S3_SP = 34;             // slots per window row (32 outputs + shifted neighbour + 1)
constexpr int S3_MAXROWS = 9;

struct S3Geom {
  int nrows, ti0, hi0, wi0;
};

// the window rows of output pixels [wo0, wo0+32) of output row (n, to, ho): S3_NT slots per lane, loaded into
// registers (s3_load: issued one item ahead of their use) and committed to LDS later (s3_store)
constexpr int S3_NT = (S3_MAXROWS * S3_SP + 63) / 64;       // 5 slot tasks per lane
struct S3Regs {
  u32x4 v[S3_NT];
};
// a lane's S3_NT slot tasks: window row, (frame, line) of that row, slot position, element offset of its 12 bytes from the
// window's first pixel (of the LAST window row for the tasks past the image: a valid address, masked) - constants of the launch
struct S3Tasks {
  int row[S3_NT], at[S3_NT], ah[S3_NT], sp[S3_NT], off[S3_NT];
  __device__ __forceinline__ void init(const pb_stem_shape& s, int lane) {
    const int nrows = s.kt * s.kh;
#pragma unroll
    for (int k = 0; k < S3_NT; ++k) {
      const int t = lane + 64 * k;
      row[k] = t / S3_SP;
      sp[k] = t - row[k] * S3_SP;
      const int rc = min(row[k], nrows - 1);
      at[k] = rc / s.kh;
      ah[k] = rc - at[k] * s.kh;
      off[k] = (int)((int64_t)at[k] * s.xs_t + (int64_t)ah[k] * s.xs_h) + 6 * sp[k];      // (< 2^31 elements: a few frames)
    }
  }
};
__device__ __forceinline__ void s3_load(S3Regs& rg, const bf16_t* x, const pb_stem_shape& s, const S3Tasks& tk, int n, int to, int ho,
                                        int wo0) {
  typedef __attribute__((ext_vector_type(3))) unsigned u32x3;
  const int nrows = s.kt * s.kh;
  const int ti0 = to * s.st - s.pt, hi0 = ho * s.sh - s.ph, wi0 = wo0 * 2 - s.pw;
  const bf16_t* xn = x + (int64_t)n * s.xs_n;
  // (wave-uniform) every window row inside the clip - 7 of 8 frames.  ALL of the lane's loads are issued unconditionally from
  // one base address plus the lane's constant offset, the pixels outside the row are masked afterwards: the general path below
  // branches per slot (whole / half / no slot) and its loads, each waited for inside its branch, went out one round trip after
  // the other - 3200 of the weight gradient's 5800 cycles per item.  A slot that hangs over the row's end reads the
  // neighbouring line's bytes: never over the frame's first or last line (excluded here), so always inside the clip.
  const bool over_first = wi0 < 0 && hi0 == 0, over_last = wi0 + 2 * (S3_SP - 1) + 1 >= s.Wi && hi0 + s.kh == s.Hi;
  if (ti0 >= 0 && ti0 + s.kt <= s.Ti && hi0 >= 0 && hi0 + s.kh <= s.Hi && !over_first && !over_last && s.xs_h >= (int64_t)s.Wi * 3) {
    const bf16_t* base = xn + (int64_t)ti0 * s.xs_t + (int64_t)hi0 * s.xs_h + (int64_t)wi0 * 3;
    u32x3 q[S3_NT];
#pragma unroll
    for (int k = 0; k < S3_NT; ++k) q[k] = *reinterpret_cast<const u32x3*>(base + tk.off[k]);     // 12 bytes at a 2-byte-aligned address
#pragma unroll
    for (int k = 0; k < S3_NT; ++k) {
      const int wa = wi0 + 2 * tk.sp[k];
      const bool rok = tk.row[k] < nrows;
      const bool oka = rok && (unsigned)wa < (unsigned)s.Wi, okb = rok && (unsigned)(wa + 1) < (unsigned)s.Wi;
      // elements 0..2 = pixel wa, 3..5 = pixel wa + 1
      rg.v[k] = u32x4{oka ? q[k][0] : 0u, (oka ? q[k][1] & 0x0000ffffu : 0u) | (okb ? q[k][1] & 0xffff0000u : 0u), okb ? q[k][2] : 0u,
                      tk.row[k] == 0 ? 0x3f80u : 0u};
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < S3_NT; ++k) {
    const int r = tk.row[k], sp = tk.sp[k];
    const int ti = ti0 + tk.at[k], hi = hi0 + tk.ah[k];
    const bool rowok = r < nrows && ti >= 0 && ti < s.Ti && hi >= 0 && hi < s.Hi;
    const int wa = wi0 + 2 * sp, wb = wa + 1;
    const bool oka = rowok && wa >= 0 && wa < s.Wi, okb = rowok && wb >= 0 && wb < s.Wi;
    const bf16_t* p = xn + (int64_t)ti * s.xs_t + (int64_t)hi * s.xs_h + (int64_t)wa * 3;
    u32x4 o = {0u, 0u, 0u, r == 0 ? 0x3f80u : 0u};        // element 6 = 1.0 on window row 0 (bias), element 7 = 0
    if (oka && okb) {
      const u32x3 q = *reinterpret_cast<const u32x3*>(p);  // 12 bytes at a 2-byte-aligned address
      o[0] = q[0]; o[1] = q[1]; o[2] = q[2];
    } else if (oka || okb) {
      const unsigned short* ps = reinterpret_cast<const unsigned short*>(p);
      unsigned e[6] = {0u, 0u, 0u, 0u, 0u, 0u};
      if (oka) { e[0] = ps[0]; e[1] = ps[1]; e[2] = ps[2]; }
      if (okb) { e[3] = ps[3]; e[4] = ps[4]; e[5] = ps[5]; }
      o[0] = e[0] | (e[1] << 16); o[1] = e[2] | (e[3] << 16); o[2] = e[4] | (e[5] << 16);
    }
    rg.v[k] = o;
  }
}
__device__ __forceinline__ void s3_store(const S3Regs& rg, bf16_t* win, int nrows, int lane) {
#pragma unroll
  for (int k = 0; k < S3_NT; ++k) {
    const int t = lane + 64 * k;
    if (t < nrows * S3_SP) *reinterpret_cast<u32x4*>(win + t * 8) = rg.v[k];
  }
}
struct S3Item {
  int n, to, ho, wo0, npx;
};
__device__ __forceinline__ S3Item s3_item(const pb_stem_shape& s, int64_t it, int wblocks) {
  S3Item q;
  int64_t rr = it;
  const int wb = (int)(rr % wblocks); rr /= wblocks;
  q.ho = (int)(rr % s.Ho); rr /= s.Ho;
  q.to = (int)(rr % s.To);
  q.n = (int)(rr / s.To);
  q.wo0 = wb * SW_PIX;
  q.npx = min(SW_PIX, s.Wo - q.wo0);
  return q;
}

// The items of a wave are it0, it0 + step, it0 + 2 step, ...: s3_item's three 64-bit divisions per item (and per look-ahead)
// were ~800 of the ~1700 instructions a wave issued per item - the stem kernels were bound by index arithmetic, not by
// memory.  The walker splits `step` into its (wb, ho, to, n) digits once and advances with carries.
struct S3Walk {
  int wb, ho, to, n;           // current item
  int dwb, dho, dto, dn;       // digits of the step
  int wblocks, Ho, To, Wo;
  __device__ __forceinline__ void init(const pb_stem_shape& s, int64_t it, int64_t step, int wblocks_) {
    wblocks = wblocks_; Ho = s.Ho; To = s.To; Wo = s.Wo;
    int64_t rr = it;
    wb = (int)(rr % wblocks); rr /= wblocks;
    ho = (int)(rr % Ho); rr /= Ho;
    to = (int)(rr % To);
    n = (int)(rr / To);
    rr = step;
    dwb = (int)(rr % wblocks); rr /= wblocks;
    dho = (int)(rr % Ho); rr /= Ho;
    dto = (int)(rr % To);
    dn = (int)(rr / To);
  }
  __device__ __forceinline__ void next() {
    wb += dwb;
    int c = wb >= wblocks;
    wb -= c ? wblocks : 0;
    ho += dho + c;
    c = ho >= Ho;
    ho -= c ? Ho : 0;
    to += dto + c;
    c = to >= To;
    to -= c ? To : 0;
    n += dn + c;
  }
  __device__ __forceinline__ S3Item item() const {
    S3Item q;
    q.n = n; q.to = to; q.ho = ho; q.wo0 = wb * SW_PIX; q.npx = min(SW_PIX, Wo - q.wo0);
    return q;
  }
};

// 16 waves per workgroup: every workgroup ends with 1312 float atomics onto the same addresses, and those serialise
// chip-wide (measured +147 us per 1024 extra workgroups) - same wave count per CU from a quarter of the workgroups
constexpr int S3W_NW = 16;
__global__ __launch_bounds__(64 * S3W_NW) void stem3_wgrad_kernel(StemArgs a) {
  typedef bf16_t T;
  __shared__ __attribute__((aligned(16))) T win_s[S3W_NW][S3_MAXROWS * S3_SP * 8];
  __shared__ __attribute__((aligned(16))) T dy_s[S3W_NW][SW_PIX * OCB];
  __shared__ float red[OCB * S3_MAXROWS * 16];
  const pb_stem_shape& s = a.s;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nrows = s.kt * s.kh, taps = nrows * 3, CT = 3 * taps;
  const int oc0 = blockIdx.y * OCB;
  const int wblocks = (s.Wo + SW_PIX - 1) / SW_PIX;
  const int64_t nitems = (int64_t)s.N * s.To * s.Ho * wblocks;
  const T* dy = (const T*)a.y;
  const T* x = (const T*)a.x;
  T* win = win_s[wave];
  T* dyt = dy_s[wave];
  f32x4 acc[S3_MAXROWS];
#pragma unroll
  for (int r = 0; r < S3_MAXROWS; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int g = lane >> 4, q = (lane & 15) >> 2, p4 = lane & 3;
  // B fragment source of this lane inside a window row: slot (8g + q [+4]) + shift, element quad
  const int boff = ((8 * g + q) + (p4 >> 1)) * 8 + (p4 & 1) * 4;

  // software pipeline over the wave's items: the next item's window slots and dy tile are fetched into registers
  // before this item's MFMAs, so one HBM round trip per item is no longer on the critical path
  S3Regs rg;
  Vec16<T> dyv;
  const int64_t step = (int64_t)gridDim.x * S3W_NW;
  S3Tasks tk;
  tk.init(s, lane);
  S3Walk walk;                                             // the next item to fetch
  {
    const int64_t first = (int64_t)blockIdx.x * S3W_NW + wave;
    walk.init(s, first < nitems ? first : 0, step, wblocks);
  }
  auto fetch = [&](int64_t) {
    const S3Item q = walk.item();
    walk.next();
    s3_load(rg, x, s, tk, q.n, q.to, q.ho, q.wo0);
    const int64_t pix0 = (((int64_t)q.n * s.To + q.to) * s.Ho + q.ho) * s.Wo + q.wo0;
    const int p = lane * 8 / OCB;                          // 64 lanes x 16 B = the whole [32][16] tile
    dyv = p < q.npx ? vec_load<T>(dy + pix0 * OCB + (int64_t)lane * 8) : vec_zero<T>();
  };
  int64_t it = (int64_t)blockIdx.x * S3W_NW + wave;
  if (it < nitems) fetch(it);
  long long tph[4] = {0, 0, 0, 0};
  long long tlast = __builtin_amdgcn_s_memtime();
  int nit = 0;
#ifdef PB_STEM_PROBE
#define S3W_TICK(i) do { if (a.ablate == 3) { const long long tn = __builtin_amdgcn_s_memtime(); tph[i] += tn - tlast; tlast = tn; } } while (0)
#define S3W_ABLATE a.ablate
#else
#define S3W_TICK(i) do { } while (0)
#define S3W_ABLATE 0
#endif
  for (; it < nitems; it += step) {
    ++nit;
    if (S3W_ABLATE == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    S3W_TICK(0);
    s3_store(rg, win, nrows, lane);
    vec_store<T>(dyt + lane * 8, dyv);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    S3W_TICK(1);
    if (it + step < nitems) fetch(it + step);
    S3W_TICK(2);
    typedef __attribute__((address_space(3))) bf16x4* lds_p;
    const bf16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(dyt + (8 * g + q) * OCB + 4 * p4));
    const bf16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(dyt + (8 * g + 4 + q) * OCB + 4 * p4));
    const bf16x8 fa = bf16x8{alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
#pragma unroll
    for (int r = 0; r < S3_MAXROWS; ++r) {
      if (r < nrows) {
        const bf16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(win + r * S3_SP * 8 + boff));
        const bf16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(win + r * S3_SP * 8 + boff + 4 * 8));
        const bf16x8 fb = bf16x8{blo[0], blo[1], blo[2], blo[3], bhi[0], bhi[1], bhi[2], bhi[3]};
        acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[r], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    S3W_TICK(3);
  }
  if (S3W_ABLATE == 3 && blockIdx.x == 7 && tid == 64)
    printf("stem3 wgrad phases (cycles/item, wave 1 of block 7, %d items): load wait %lld, lds stores %lld, fetch issue %lld, mfma %lld\n",
           nit, tph[0] / nit, tph[1] / nit, tph[2] / nit, tph[3] / nit);
#undef S3W_TICK
#undef S3W_ABLATE
  // D[oc][n] per window row: n < 6 -> tap (aw = n/3, ci = n%3); 8 <= n < 11 -> (aw = 2, ci = n-8); row 0, n = 6 -> bias.
  // Cross-wave sum in wave order (no atomics), then plain stores into this workgroup's slot of the workspace.
  for (int i = tid; i < OCB * S3_MAXROWS * 16; i += 64 * S3W_NW) red[i] = 0.f;
  __syncthreads();
  for (int w = 0; w < S3W_NW; ++w) {
    if (wave == w) {
#pragma unroll
      for (int r = 0; r < S3_MAXROWS; ++r)
        if (r < nrows)
#pragma unroll
          for (int i = 0; i < 4; ++i) red[((lane >> 4) * 4 + i) * (S3_MAXROWS * 16) + r * 16 + (lane & 15)] += acc[r][i];
    }
    __syncthreads();
  }
  float* part = a.dw + (size_t)blockIdx.x * ((size_t)s.Cout * (CT + 1));     // [Cout][CT] then [Cout] (bias)
  for (int i = tid; i < OCB * (CT + 1); i += 64 * S3W_NW) {
    const int oc = i / (CT + 1), c = i - oc * (CT + 1);
    if (oc0 + oc >= s.Cout) continue;
    if (c < CT) {
      const int ci = c / taps, tp = c - ci * taps;
      const int r = tp / 3, aw = tp - r * 3;
      const int nn = aw < 2 ? aw * 3 + ci : 8 + ci;
      part[(int64_t)(oc0 + oc) * CT + c] = red[oc * (S3_MAXROWS * 16) + r * 16 + nn];
    } else {
      part[(int64_t)s.Cout * CT + oc0 + oc] = red[oc * (S3_MAXROWS * 16) + 6];
    }
  }
}

// Forward on the same slot image: K runs over 18 groups of 8 (window row r, shift) = 144 (+16 zero) columns, an A
// fragment (16 pixels x 8 columns of one group) is ONE ds_read_b128 per lane, the filter (with the bias on the ones
// element) sits in registers for the whole walk.
__global__ __launch_bounds__(256) void stem3_fwd_kernel(StemArgs a) {
  typedef bf16_t T;
  __shared__ __attribute__((aligned(16))) T win_s[4][S3_MAXROWS * S3_SP * 8];
  __shared__ __attribute__((aligned(16))) T out_s[4][SW_PIX * OCB];
  __shared__ float red[4][2 * OCB];
  const pb_stem_shape& s = a.s;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nrows = s.kt * s.kh, taps = nrows * 3, CT = 3 * taps;
  const int wblocks = (s.Wo + SW_PIX - 1) / SW_PIX;
  const int64_t nitems = (int64_t)s.N * s.To * s.Ho * wblocks;
  const T* x = (const T*)a.x;
  T* y = (T*)a.y;
  T* win = win_s[wave];
  T* outt = out_s[wave];
  const int row = lane & 15, g = lane >> 4;
  constexpr int NKS = 5;                                   // 5 x 32 = 160 >= 18 groups x 8
  bf16x8 wf[NKS], wl[NKS];                                 // filter as a bf16 hi + lo pair: the first layer's rounding
  int aoff[NKS];                                           // error reaches every later one (logits: 2.6e-2 -> 1e-2 off)
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const int grp = 4 * ks + g, r = grp >> 1, shift = grp & 1;
    const bool gv = r < nrows;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = 0.f;
      if (gv) {
        if (shift == 0 && j < 6) v = a.w[(int64_t)row * CT + (j % 3) * taps + r * 3 + j / 3];
        else if (shift == 0 && j == 6 && r == 0 && a.bias) v = a.bias[row];
        else if (shift == 1 && j < 3) v = a.w[(int64_t)row * CT + j * taps + r * 3 + 2];
      }
      wf[ks][j] = (bf16_t)v;
      wl[ks][j] = (bf16_t)(v - (float)wf[ks][j]);
    }
    aoff[ks] = ((gv ? r : 0) * S3_SP + shift) * 8;
  }
  float ssum = 0.f, ssq = 0.f;

  // (fetching the next item's window before this item's MFMAs was measured slower here: 206 vs 152 us)
  const int64_t step = (int64_t)gridDim.x * 4;
  S3Tasks tk;
  tk.init(s, lane);
  S3Walk walk;
  walk.init(s, (int64_t)blockIdx.x * 4 + wave < nitems ? (int64_t)blockIdx.x * 4 + wave : 0, step, wblocks);
  for (int64_t it = (int64_t)blockIdx.x * 4 + wave; it < nitems; it += step) {
    const S3Item cur = walk.item();
    walk.next();
    const int n = cur.n, to = cur.to, ho = cur.ho, wo0 = cur.wo0, npx = cur.npx;
    {
      S3Regs rg;
      s3_load(rg, x, s, tk, n, to, ho, wo0);
      s3_store(rg, win, nrows, lane);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    f32x4 acc[2], accl[2];                                 // (separate chains for the lo parts: 4 independent MFMA streams)
    acc[0] = acc[1] = accl[0] = accl[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const bf16x8 fa = *reinterpret_cast<const bf16x8*>(win + aoff[ks] + (mt * 16 + row) * 8);
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, wf[ks], acc[mt], 0, 0, 0);
        accl[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, wl[ks], accl[mt], 0, 0, 0);
      }
    }
    acc[0] += accl[0];
    acc[1] += accl[1];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int p = mt * 16 + g * 4 + i;
        const T tv = from_f32<T>(acc[mt][i]);
        outt[p * OCB + row] = tv;
        if (a.stats && p < npx) {
          const float fv = to_f32(tv);
          ssum += fv;
          ssq += fv * fv;
        }
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    {
      const int64_t pix0 = (((int64_t)n * s.To + to) * s.Ho + ho) * s.Wo + wo0;
      for (int e = lane; e < SW_PIX * OCB / 8; e += 64) {
        const int p = e * 8 / OCB;
        if (p < npx) vec_store<T>(y + pix0 * OCB + (int64_t)e * 8, vec_load<T>(outt + e * 8));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
  if (a.stats) {
    ssum += __shfl_xor(ssum, 16, 64); ssum += __shfl_xor(ssum, 32, 64);
    ssq += __shfl_xor(ssq, 16, 64); ssq += __shfl_xor(ssq, 32, 64);
    if (lane < 16) {
      red[wave][lane] = ssum;
      red[wave][OCB + lane] = ssq;
    }
    __syncthreads();
    if (tid < OCB) {
      float* st = a.stats + (size_t)blockIdx.x * 2 * s.Cout;        // this workgroup's own slot (gridDim.x <= PB_STAT_SLOTS)
      st[tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
      st[s.Cout + tid] = (red[0][OCB + tid] + red[1][OCB + tid]) + (red[2][OCB + tid] + red[3][OCB + tid]);
    }
  }
}

// ---- the same forward with the window rows arriving by LDS-DMA (round 4) -----------------------------------------
// stem3_fwd_kernel fetches an item's 9 window rows as 5 unaligned 12-byte loads per lane, waits for them, computes, stores -
// one item at a time per wave, nothing in flight while it computes: 190 us for 257 MB (0.17 of the HBM peak, two rounds
// overdue).  Here the RAW rows (the 416 bytes of 16-byte-aligned chunks that cover input pixels 2wo0-1 .. 2wo0+64) go HBM ->
// LDS by `global_load_lds_dwordx4` into a ring of S3D_NRAW item buffers per wave, S3D_NRAW-1 items AHEAD of the one being
// computed and without registers; the slot image (slot p = the 6 elements of pixels (2p-1, 2p) + [one, zero]) is then built
// from LDS: 4 aligned ds_read_b32 + 3 v_alignbit per slot (the 12 bytes start 2 bytes into a dword).  MFMA part unchanged.
// vmcnt: 4 DMAs and 1 store per item, counted exactly (switch over immediates).
constexpr int S3D_NRAW = 2;
constexpr int S3D_RAWCH = 26;                 // 16-byte chunks per raw row: bytes [12 wo0 - 16, 12 wo0 + 400)
constexpr int S3D_RAWROW = S3D_RAWCH * 16;
constexpr int S3D_RAWBUF = 4096;              // 9 rows x 416 B = 3744 B, rounded to the 4 DMA instructions' 4 KiB
constexpr int S3D_LDS = 4 * S3D_NRAW * S3D_RAWBUF + 4 * S3_MAXROWS * S3_SP * 16;

__device__ __forceinline__ void s3d_glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

#ifdef PB_STEM_PROBE     // (hipcc -DPB_STEM_PROBE + PB_STEM_ABLATE=1 DMA only / 2 compute only / 3 cycles per phase: how the numbers above were taken)
#define S3D_TICK(i) do { if (a.ablate == 3) { const long long tn = __builtin_amdgcn_s_memtime(); tph[i] += tn - tlast; tlast = tn; } } while (0)
#define S3D_ABLATE a.ablate
#else
#define S3D_TICK(i) do { } while (0)
#define S3D_ABLATE 0
#endif
__global__ __launch_bounds__(256) void stem3_fwd_dma_kernel(StemArgs a) {
  typedef bf16_t T;
  extern __shared__ __attribute__((aligned(1024))) unsigned char s3d_smem[];
  __shared__ float red[4][2 * OCB];
  const pb_stem_shape& s = a.s;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* raw_w = s3d_smem + wave * (S3D_NRAW * S3D_RAWBUF);
  T* win = (T*)(s3d_smem + 4 * S3D_NRAW * S3D_RAWBUF) + wave * (S3_MAXROWS * S3_SP * 8);
  typedef __attribute__((address_space(3))) unsigned char* lds_b;
  const unsigned raw_lds = (unsigned)(uintptr_t)(lds_b)raw_w;
  const int nrows = s.kt * s.kh, taps = nrows * 3, CT = 3 * taps;
  const int wblocks = (s.Wo + SW_PIX - 1) / SW_PIX;
  const int64_t nitems = (int64_t)s.N * s.To * s.Ho * wblocks;
  const T* x = (const T*)a.x;
  T* y = (T*)a.y;
  const int row = lane & 15, g = lane >> 4;
  constexpr int NKS = 5;
  bf16x8 wf[NKS], wl[NKS];
  int aoff[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const int grp = 4 * ks + g, r = grp >> 1, shift = grp & 1;
    const bool gv = r < nrows;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = 0.f;
      if (gv) {
        if (shift == 0 && j < 6) v = a.w[(int64_t)row * CT + (j % 3) * taps + r * 3 + j / 3];
        else if (shift == 0 && j == 6 && r == 0 && a.bias) v = a.bias[row];
        else if (shift == 1 && j < 3) v = a.w[(int64_t)row * CT + j * taps + r * 3 + 2];
      }
      wf[ks][j] = (bf16_t)v;
      wl[ks][j] = (bf16_t)(v - (float)wf[ks][j]);
    }
    aoff[ks] = ((gv ? r : 0) * S3_SP + shift) * 8;
  }
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+v"(wf[ks]), "+v"(wl[ks]));     // (loads consumed: vmcnt is counted by hand below)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // statistics: this lane's four output channels 4g .. 4g+3 (the MFMA result is [channel][pixel])
  float cs[4] = {0.f, 0.f, 0.f, 0.f}, cq[4] = {0.f, 0.f, 0.f, 0.f};

  // A wave issues one instruction every ~5 cycles and only two waves share a SIMD here (74 KB of LDS per workgroup): the
  // kernel is bound by its own instruction count.  Measured per item (s_memtime): 1760 cycles building the slot image, 970
  // for the MFMAs and the output, 760 issuing four DMAs, 570 on the item arithmetic - so: the item walk is incremental
  // (S3Walk), everything that depends on the lane alone is computed once, interior items (all window rows and pixels inside
  // the clip: 7 of 8 frames, the two middle blocks of a row) skip the masks and the address clamps, the output goes from
  // the accumulators to memory (no LDS transpose: the MFMA operands are swapped so that a lane holds 4 channels of a pixel).
  // (An XCD-aware item order - XCD x sweeping whole clips so that the halo rows / frames are L2 hits - measured SLOWER: 159 ->
  // 174 us; the spread order stays.)
  const int64_t step = (int64_t)gridDim.x * 4;
  const int64_t it0 = (int64_t)blockIdx.x * 4 + wave;
  const int nloc = it0 < nitems ? (int)((nitems - it0 + step - 1) / step) : 0;        // this wave's items
  S3Walk wcur, wahead;                              // the item being computed / the next one to fetch (both wave-uniform)
  wcur.init(s, it0 < nitems ? it0 : 0, step, wblocks);
  wahead = wcur;
  const int rowbytes = s.Wi * 6;
  // DMA: instruction i moves raw chunks i*64 + lane (row = idx / 26, chunk = idx % 26; the 22 spare lanes repeat the last chunk);
  // the window row's (frame, line) and the chunk's byte offset are per-lane constants of the launch
  int dat[4], dah[4], dcb[4], doff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = min(i * 64 + lane, nrows * S3D_RAWCH - 1);
    const int dr = idx / S3D_RAWCH;
    dat[i] = dr / s.kh;
    dah[i] = dr - dat[i] * s.kh;
    dcb[i] = 16 * (idx - dr * S3D_RAWCH) - 16;
    doff[i] = (int)(((int64_t)dat[i] * s.xs_t + (int64_t)dah[i] * s.xs_h) * 2);      // (< 2^31: stem3_dma_ok)
  }
  int slot = 0;                                      // ring slot of the next fetch
  auto issue = [&]() {
    const S3Item q = wahead.item();
    wahead.next();
    const unsigned dst = raw_lds + (unsigned)slot * S3D_RAWBUF;
    slot = slot + 1 == S3D_NRAW ? 0 : slot + 1;
    const unsigned char* xn = (const unsigned char*)(x + (int64_t)q.n * s.xs_n);
    const int ti0 = q.to * s.st - s.pt, hi0 = q.ho * s.sh - s.ph;
    const int b0 = q.wo0 * 12;                                                       // (raw byte 0 = b0 - 16 in row bytes)
    if (ti0 >= 0 && ti0 + s.kt <= s.Ti && hi0 >= 0 && hi0 + s.kh <= s.Hi) {          // (wave-uniform) every window row exists
      const unsigned char* base = xn + ((int64_t)ti0 * s.xs_t + (int64_t)hi0 * s.xs_h) * 2;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int cb = min(max(b0 + dcb[i], 0), rowbytes - 16);                     // chunks outside the row: any valid address -
        s3d_glds16(base + (doff[i] + cb), dst + i * 1024);                           // masked when the slots are built
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ti = min(max(ti0 + dat[i], 0), s.Ti - 1), hi = min(max(hi0 + dah[i], 0), s.Hi - 1);   // rows outside: any valid
        const int cb = min(max(b0 + dcb[i], 0), rowbytes - 16);
        s3d_glds16(xn + ((int64_t)ti * s.xs_t + (int64_t)hi * s.xs_h) * 2 + cb, dst + i * 1024);
      }
    }
  };
  if (S3D_ABLATE != 2)
    for (int k = 0; k < min(nloc, S3D_NRAW); ++k) issue();
  // slot build: task kk of a lane is slot t = lane + 64 kk of the image - its window row, frame / line and position, its 12
  // source bytes inside a raw buffer and whether it exists are per-lane constants too
  int brow[S3_NT], bat[S3_NT], bah[S3_NT], bsp[S3_NT], bsrc[S3_NT];
  bool bok[S3_NT];
#pragma unroll
  for (int kk = 0; kk < S3_NT; ++kk) {
    const int t = lane + 64 * kk;
    brow[kk] = t / S3_SP;
    bsp[kk] = t - brow[kk] * S3_SP;
    bat[kk] = brow[kk] / s.kh;
    bah[kk] = brow[kk] - bat[kk] * s.kh;
    bok[kk] = brow[kk] < nrows && bsp[kk] < S3_SP - 1;
    bsrc[kk] = min(brow[kk], S3_MAXROWS - 1) * S3D_RAWROW + 8 + 12 * bsp[kk];          // the slot's 12 bytes start at +2 (all inside the 4 KiB buffer)
  }
  int cslot = 0;                                     // ring slot of the item being computed

  long long tph[6] = {0, 0, 0, 0, 0, 0};
  long long tlast = __builtin_amdgcn_s_memtime();

  for (int k = 0; k < nloc; ++k) {
    const S3Item cur = wcur.item();
    wcur.next();
    const int n = cur.n, to = cur.to, ho = cur.ho, wo0 = cur.wo0, npx = cur.npx;
    S3D_TICK(5);
    // behind item k's DMAs: the DMAs of the items issued after it and AT LEAST one store per iteration since it was issued (the
    // second store of an item is skipped when its 16 pixels are past the row's end: the allowance may only round down)
    {
      const int behind = 4 * min(S3D_NRAW - 1, nloc - 1 - k) + min(k, S3D_NRAW);
      switch (behind) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
      }
    }
    S3D_TICK(0);
    if (S3D_ABLATE == 1) {
      if (k + S3D_NRAW < nloc) issue();
      continue;
    }
    // ---- raw rows -> slot image: all of a lane's reads in flight together, then the shifts, then the writes
    {
      const unsigned char* raw = raw_w + cslot * S3D_RAWBUF;
      cslot = cslot + 1 == S3D_NRAW ? 0 : cslot + 1;
      const int ti0 = to * s.st - s.pt, hi0 = ho * s.sh - s.ph, wi0 = wo0 * 2 - s.pw;
      unsigned d[S3_NT][4];
#pragma unroll
      for (int kk = 0; kk < S3_NT; ++kk) {
        const unsigned* rp = reinterpret_cast<const unsigned*>(raw + bsrc[kk]);
        d[kk][0] = rp[0]; d[kk][1] = rp[1]; d[kk][2] = rp[2]; d[kk][3] = rp[3];
      }
      unsigned o[S3_NT][3];
#pragma unroll
      for (int kk = 0; kk < S3_NT; ++kk) {
        // elements 0..2 = pixel wa = wi0 + 2 sp, 3..5 = pixel wa + 1
        o[kk][0] = __builtin_amdgcn_alignbit(d[kk][1], d[kk][0], 16);
        o[kk][1] = __builtin_amdgcn_alignbit(d[kk][2], d[kk][1], 16);
        o[kk][2] = __builtin_amdgcn_alignbit(d[kk][3], d[kk][2], 16);
      }
      const bool inside = ti0 >= 0 && ti0 + s.kt <= s.Ti && hi0 >= 0 && hi0 + s.kh <= s.Hi && wi0 >= 0 &&
                          wi0 + 2 * (S3_SP - 2) + 1 < s.Wi;                           // (wave-uniform)
      if (!inside) {
#pragma unroll
        for (int kk = 0; kk < S3_NT; ++kk) {
          const int ti = ti0 + bat[kk], hi = hi0 + bah[kk];
          const bool rowok = ti >= 0 && ti < s.Ti && hi >= 0 && hi < s.Hi;
          const int wa = wi0 + 2 * bsp[kk], wb = wa + 1;
          const bool oka = rowok && wa >= 0 && wa < s.Wi, okb = rowok && wb >= 0 && wb < s.Wi;
          if (!oka) { o[kk][0] = 0u; o[kk][1] &= 0xffff0000u; }
          if (!okb) { o[kk][1] &= 0x0000ffffu; o[kk][2] = 0u; }
        }
      }
#pragma unroll
      for (int kk = 0; kk < S3_NT; ++kk)
        if (bok[kk]) *reinterpret_cast<u32x4*>(win + (lane + 64 * kk) * 8) = u32x4{o[kk][0], o[kk][1], o[kk][2], brow[kk] == 0 ? 0x3f80u : 0u};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    S3D_TICK(1);
    if (k + S3D_NRAW < nloc && S3D_ABLATE != 2) issue();         // this item's raw buffer is free again
    S3D_TICK(2);
    f32x4 acc[2], accl[2];
    acc[0] = acc[1] = accl[0] = accl[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        // (weights as the A operand: D[channel 4g + i][pixel mt*16 + row] - a lane ends up with 4 channels of one pixel)
        const bf16x8 fa = *reinterpret_cast<const bf16x8*>(win + aoff[ks] + (mt * 16 + row) * 8);
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks], fa, acc[mt], 0, 0, 0);
        accl[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ks], fa, accl[mt], 0, 0, 0);
      }
    }
    acc[0] += accl[0];
    acc[1] += accl[1];
    S3D_TICK(3);
    {
      // 8 bytes per lane: pixel p's channels 4g .. 4g+3; a store instruction covers 16 pixels = 512 contiguous bytes.  Lane 0 of
      // the first store is never off: at least ONE store instruction per item in the vmcnt count
      const int64_t pix0 = (((int64_t)n * s.To + to) * s.Ho + ho) * s.Wo + wo0;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int p = mt * 16 + row;
        const bf16x4 ov = __builtin_convertvector(acc[mt], bf16x4);
        if (p < npx) {
          *reinterpret_cast<bf16x4*>(y + (pix0 + p) * OCB + 4 * g) = ov;
          if (a.stats) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float fv = (float)ov[i];
              cs[i] += fv;
              cq[i] += fv * fv;
            }
          }
        }
      }
    }
    S3D_TICK(4);
  }
  if (S3D_ABLATE == 3 && blockIdx.x == 7 && tid == 64)
    printf("stem3 fwd phases (cycles/item, wave 1 of block 7, %d items): dma wait %lld, slot build %lld, issue %lld, mfma %lld, store+stats %lld, loop head %lld\n",
           nloc, tph[0] / nloc, tph[1] / nloc, tph[2] / nloc, tph[3] / nloc, tph[4] / nloc, tph[5] / nloc);
  if (a.stats) {
    // the 16 lanes of a group hold the same 4 channels for different pixels
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        cs[i] += __shfl_xor(cs[i], o, 64);
        cq[i] += __shfl_xor(cq[i], o, 64);
      }
    if (row == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        red[wave][4 * g + i] = cs[i];
        red[wave][OCB + 4 * g + i] = cq[i];
      }
    }
    __syncthreads();
    if (tid < OCB) {
      float* st = a.stats + (size_t)blockIdx.x * 2 * s.Cout;
      st[tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
      st[s.Cout + tid] = (red[0][OCB + tid] + red[1][OCB + tid]) + (red[2][OCB + tid] + red[3][OCB + tid]);
    }
  }
}
#undef S3D_TICK
#undef S3D_ABLATE

// the DMA forward additionally needs 16-byte-aligned rows (pointer, strides) of whole chunks and the reference stem's 3x3 window rows
bool stem3_dma_ok(const pb_stem_shape* s, const void* x) {
  static const bool enabled = !(getenv("PB_STEM3_DMA") && atoi(getenv("PB_STEM3_DMA")) == 0);
  return enabled && ((uintptr_t)x % 16) == 0 && s->xs_h % 8 == 0 && s->xs_t % 8 == 0 && s->xs_n % 8 == 0 && s->Wi % 8 == 0 &&
         s->Wi >= 72 && s->kt * s->kh == S3_MAXROWS && s->Wo % 4 == 0 &&
         (int64_t)s->xs_n * 2 < (1LL << 40);
}

bool stem3_ok(const pb_stem_shape* s, int x_dtype, pb_dtype dtype) {
  static const bool enabled = !(getenv("PB_STEM3") && atoi(getenv("PB_STEM3")) == 0);
  return enabled && x_dtype == 1 && dtype == PB_BF16 && s->Cin == 3 && s->kw == 3 && s->sw == 2 && s->pw == 1 &&
         s->xs_c == 1 && s->xs_w == 3 && s->kt * s->kh <= S3_MAXROWS && s->Cout == OCB;
}

int check_stem(const pb_stem_shape* s, int x_dtype, pb_dtype dtype, const char* who) {
  PB_CHECK_ARG(s, "%s: null shape", who);
  PB_CHECK_ARG(s->N > 0 && s->Cin > 0 && s->Cout > 0 && s->Ti > 0 && s->Hi > 0 && s->Wi > 0 && s->To > 0 &&
               s->Ho > 0 && s->Wo > 0, "%s: non-positive dimension", who);
  PB_CHECK_ARG(s->Cin * s->kt * s->kh * s->kw <= MAX_CT, "%s: Cin*taps = %d exceeds %d", who,
               s->Cin * s->kt * s->kh * s->kw, MAX_CT);
  PB_CHECK_ARG((s->To - 1) * s->st - s->pt < s->Ti + s->pt && (s->Ho - 1) * s->sh - s->ph < s->Hi + s->ph &&
               (s->Wo - 1) * s->sw - s->pw < s->Wi + s->pw, "%s: output extent inconsistent", who);
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
                                  void* y, const pb_stem_shape* s, pb_dtype dtype, float* stats, int stat_slots,
                                  int device, void* stream) {
  PB_CHECK_ARG(x && w && y, "pb_stem_conv3d_fwd: null pointer");
  PB_CHECK_ARG(!stats || stat_slots >= 1, "pb_stem_conv3d_fwd: stat_slots must be >= 1");
  if (int rc = check_stem(s, x_dtype, dtype, "pb_stem_conv3d_fwd")) return rc;
  PB_ENTER(device);
  hipStream_t st = (hipStream_t)stream;
  StemArgs a = {};
  a.x = x; a.x_scale = x_scale; a.w = w; a.bias = bias; a.y = y; a.stats = stats; a.s = *s;
  a.npix = (int64_t)s->N * s->To * s->Ho * s->Wo;
  const int wl = 31 * s->sw + s->kw;
  if (s->kt * s->kh * wl * s->Cin <= SW_MAXWIN && s->Cin * s->kt * s->kh * s->kw <= 32 * SW_KS &&
      s->kt * s->kh <= FW_R && wl * s->Cin <= 64 * FW_I) {
    const int64_t items = (int64_t)s->N * s->To * s->Ho * pb_cdiv(s->Wo, SW_PIX);
    static const int fwd_wgs = getenv("PB_STEM_FWD_WGS") ? atoi(getenv("PB_STEM_FWD_WGS")) : 0;
    const bool dma = stem3_ok(s, x_dtype, dtype) && x_scale == 1.f && stem3_dma_ok(s, x);
    // the DMA kernel: 3 resident workgroups per CU (52 KB of LDS each), each walking its share of the items - every
    // workgroup pays ~8k cycles before its first item (the filter, the item walk's divisions): 2048 workgroups in three rounds
    // measured 128 us, 768 in one round 122
    const int64_t cap = fwd_wgs > 0 ? fwd_wgs : (dma ? 768 : 2048);
    int64_t gx = pb_cdiv(items, 4);
    if (gx > cap) gx = cap;
    if (stats && gx > stat_slots) gx = stat_slots;             // one statistics slot per workgroup
    dim3 grid((unsigned)gx, (unsigned)pb_cdiv(s->Cout, OCB));
    if (dma) {
      static bool attr = false;
      if (!attr) {
        PB_HIP(hipFuncSetAttribute((const void*)stem3_fwd_dma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S3D_LDS));
        attr = true;
      }
      static const int ablate = getenv("PB_STEM_ABLATE") ? atoi(getenv("PB_STEM_ABLATE")) : 0;
      a.ablate = ablate;
      hipLaunchKernelGGL(stem3_fwd_dma_kernel, grid, dim3(256), S3D_LDS, st, a);
    } else if (stem3_ok(s, x_dtype, dtype) && x_scale == 1.f) hipLaunchKernelGGL(stem3_fwd_kernel, grid, dim3(256), 0, st, a);
    else PB_STEM_DISPATCH(stem_fwd_mfma_kernel, grid);
  } else {
    int64_t gx = pb_cdiv(a.npix, 256);
    if (gx > 2048) gx = 2048;
    if (stats && gx > stat_slots) gx = stat_slots;
    dim3 grid((unsigned)gx, (unsigned)pb_cdiv(s->Cout, OCB));
    PB_STEM_DISPATCH(stem_fwd_kernel, grid);
  }
  PB_LAUNCH_CHECK();
  return 0;
}

namespace {
// workgroups along x of the weight-gradient launch (= partial-gradient slots in the workspace)
int64_t stem_wgrad_gx(const pb_stem_shape* s, int x_dtype, float x_scale, pb_dtype dtype) {
  const int64_t items = (int64_t)s->N * s->To * s->Ho * pb_cdiv(s->Wo, SW_PIX);
  if (stem3_ok(s, x_dtype, dtype) && x_scale == 1.f) {
    static const int wg_cap = getenv("PB_STEM_WGRAD_WGS") ? atoi(getenv("PB_STEM_WGRAD_WGS")) : 256;
    int64_t g3 = pb_cdiv(items, S3W_NW);
    return g3 > wg_cap ? wg_cap : g3;
  }
  int64_t gx = pb_cdiv(items, 4);
  return gx > 512 ? 512 : gx;
}
}  // namespace

extern "C" int64_t pb_stem_conv3d_bwd_weight_ws_bytes(const pb_stem_shape* s, int x_dtype, float x_scale, pb_dtype dtype) {
  if (!s || s->Cout <= 0 || s->Cin <= 0) return 0;
  const int64_t CT = (int64_t)s->Cin * s->kt * s->kh * s->kw;
  return stem_wgrad_gx(s, x_dtype, x_scale, dtype) * s->Cout * (CT + 1) * 4;
}

extern "C" int pb_stem_conv3d_bwd_weight(const void* dy, const void* x, int x_dtype, float x_scale, float* dw,
                                         float* dbias, const pb_stem_shape* s, pb_dtype dtype, void* ws,
                                         int64_t ws_bytes, int device, void* stream) {
  PB_CHECK_ARG(dy && x && dw && ws, "pb_stem_conv3d_bwd_weight: null pointer");
  if (int rc = check_stem(s, x_dtype, dtype, "pb_stem_conv3d_bwd_weight")) return rc;
  PB_CHECK_ARG(ws_bytes >= pb_stem_conv3d_bwd_weight_ws_bytes(s, x_dtype, x_scale, dtype) && ((uintptr_t)ws % 16) == 0,
               "pb_stem_conv3d_bwd_weight: workspace too small or unaligned");
  PB_ENTER(device);
  hipStream_t st = (hipStream_t)stream;
  StemArgs a = {};
  a.x = x; a.x_scale = x_scale; a.y = const_cast<void*>(dy); a.dw = (float*)ws; a.dbias = nullptr; a.s = *s;
  a.npix = (int64_t)s->N * s->To * s->Ho * s->Wo;
  const int wl = 31 * s->sw + s->kw;
  PB_CHECK_ARG(s->kt * s->kh * wl * s->Cin <= SW_MAXWIN && s->kt * s->kh <= FW_R && wl * s->Cin <= 64 * FW_I,
               "pb_stem_conv3d_bwd_weight: window (%d rows x %d) exceeds the staged maximum", s->kt * s->kh, wl * s->Cin);
  const int64_t gx = stem_wgrad_gx(s, x_dtype, x_scale, dtype);
  dim3 grid((unsigned)gx, (unsigned)pb_cdiv(s->Cout, OCB));
  static const int ablate_w = getenv("PB_STEM_ABLATE") ? atoi(getenv("PB_STEM_ABLATE")) : 0;
  a.ablate = ablate_w;
  if (stem3_ok(s, x_dtype, dtype) && x_scale == 1.f) hipLaunchKernelGGL(stem3_wgrad_kernel, grid, dim3(64 * S3W_NW), 0, st, a);
  else PB_STEM_DISPATCH(stem_wgrad_kernel, grid);
  PB_LAUNCH_CHECK();
  // dw += sum of the workgroups' partial gradients, in slot order
  const int64_t CT = (int64_t)s->Cin * s->kt * s->kh * s->kw;
  const int64_t stride = (int64_t)s->Cout * (CT + 1);
  pb_reduce_job jobs[2] = {{(const float*)ws, dw, (int64_t)s->Cout * CT, stride, (int)gx, 1},
                           {(const float*)ws + (int64_t)s->Cout * CT, dbias, s->Cout, stride, (int)gx, 1}};
  if (pb_launch_partials_reduce(jobs, dbias ? 2 : 1, st)) { pb_set_error("pb_stem_conv3d_bwd_weight: combine launch failed"); return 2; }
  return 0;
}
