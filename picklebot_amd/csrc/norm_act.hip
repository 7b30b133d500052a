// Per-channel statistics, BatchNorm finalisation, fused affine+activation(+dropout/SE row scale)
// forward and backward, row-group reductions.  All are single-pass streaming kernels over the
// [M rows][C channels] NDHWC matrix: lanes run along (channel-vector, then row) so a wave's access
// is one contiguous span; per-channel sums are reduced thread -> LDS -> one atomic per workgroup
// and channel.
#include "pb_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

struct RowMap {       // thread -> (channel vector, row lane)
  int CV, CVB, P, chunks;
};

template <int CH>
RowMap make_map(int C) {
  RowMap m;
  m.CV = C / CH;
  int d = m.CV < 32 ? m.CV : 32;
  for (; d >= 1; --d)
    if (m.CV % d == 0) break;
  m.CVB = (d >= 4 || m.CV <= 32) ? d : 16;
  m.P = 256 / m.CVB;
  m.chunks = (int)pb_cdiv(m.CV, m.CVB);
  return m;
}

template <typename T, int CH>
__device__ __forceinline__ void ld(const T* p, float (&o)[CH]) {
  if constexpr (CH == ElemTraits<T>::VEC) {
    Vec16<T> v = vec_load<T>(p);
#pragma unroll
    for (int i = 0; i < CH; ++i) o[i] = v.get(i);
  } else {
#pragma unroll
    for (int i = 0; i < CH; ++i) o[i] = to_f32(p[i]);
  }
}
// A row chunk kept as loaded (16 bytes) until it is consumed: converting at load time would hold CH floats per
// in-flight row in registers (64 VGPRs for 4 rows x 2 tensors of bf16)
template <typename T, int CH, bool VECTOR = (CH == ElemTraits<T>::VEC)>
struct RawRow {
  float f[CH];
  __device__ __forceinline__ void load(const T* p) {
#pragma unroll
    for (int i = 0; i < CH; ++i) f[i] = to_f32(p[i]);
  }
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < CH; ++i) f[i] = 0.f;
  }
  __device__ __forceinline__ float get(int i) const { return f[i]; }
};
template <typename T, int CH>
struct RawRow<T, CH, true> {
  Vec16<T> v;
  __device__ __forceinline__ void load(const T* p) { v = vec_load<T>(p); }
  __device__ __forceinline__ void zero() { v = vec_zero<T>(); }
  __device__ __forceinline__ float get(int i) const { return v.get(i); }
};

template <typename T, int CH>
__device__ __forceinline__ void stv(T* p, const float (&o)[CH]) {
  if constexpr (CH == ElemTraits<T>::VEC) {
    Vec16<T> v;
#pragma unroll
    for (int i = 0; i < CH; ++i) v.set(i, o[i]);
    vec_store<T>(p, v);
  } else {
#pragma unroll
    for (int i = 0; i < CH; ++i) p[i] = from_f32<T>(o[i]);
  }
}

// Block-level reduction of NRED per-thread channel vectors over the row lanes p of the (cvl, p) thread layout, in a
// fixed order (no atomics: results are bit-reproducible), then ONE plain store per channel into this workgroup's
// own slot of the output.  red holds NRED*CH rows of 256 floats.
template <int CH, int NRED>
__device__ __forceinline__ void block_chan_reduce(float* red, const float (&v)[NRED][CH], bool lane_ok, int CVB, int P,
                                                  int cbase, int C, float* const (&out)[NRED]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int r = 0; r < NRED; ++r)
#pragma unroll
    for (int i = 0; i < CH; ++i) red[(r * CH + i) * 256 + tid] = lane_ok ? v[r][i] : 0.f;
  __syncthreads();
  const int CW = CVB * CH;
  for (int o = tid; o < NRED * CW; o += 256) {
    const int r = o / CW, cl = o - r * CW;
    const int cvl = cl / CH, i = cl - cvl * CH;
    const float* src = red + (r * CH + i) * 256 + cvl;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int p = 0;
    for (; p + 4 <= P; p += 4) {
      s0 += src[p * CVB];
      s1 += src[(p + 1) * CVB];
      s2 += src[(p + 2) * CVB];
      s3 += src[(p + 3) * CVB];
    }
    for (; p < P; ++p) s0 += src[p * CVB];
    const int c = cbase + cl;
    if (c < C) out[r][c] = (s0 + s1) + (s2 + s3);
  }
}

struct EwArgs {
  const void* a;        // primary input (x / dy / g)
  const void* b;        // secondary input (z)
  void* out;
  int64_t M;
  int C;
  const float* scale;
  const float* shift;
  int act;
  float slope;
  const float* row_scale;
  const float* row_shift;
  int64_t rows_per_group;
  int64_t dy_rows_per_group;
  float dy_mul;
  const float* mean;
  const float* rstd;
  const float* cA;
  const float* cB;
  const float* cC;
  float* sums;          // [2][C]
  RowMap map;
  // Frame map (fm_full > 0): the kernel walks M rows of a FULL tensor [clips][fm_full rows]; an operand marked compact holds
  // only the fm_data data rows of every clip (full rows [fm_lead, fm_lead + fm_data) of the clip) and reads as 0 elsewhere;
  // a compact output is written at the data rows only.  (The frames a (1,k,k) depthwise conv pads in time: zero before
  // BatchNorm, but counted by its statistics and present in its output.)
  int64_t fm_full, fm_data, fm_lead;
  int a_compact, b_compact, out_compact;
};


// MODE 0: chan_stats   sums += (sum a, sum a^2)
// MODE 1: affine_act   out = act(a*scale+shift)*row_scale ; optional sums of out
// MODE 2: act_bwd      g = dy*row_scale*act'(z*scale+shift) ; out = g (optional) ; sums += (sum g, sum g*xhat)
// MODE 3: affine2      out = cA*a + cB*b + cC
template <typename T, int CH, int MODE>
__global__ __launch_bounds__(256) void ew_kernel(EwArgs a) {
  __shared__ float red[(MODE == 3 || MODE == 4) ? 1 : 2 * CH * 256];
  const int tid = threadIdx.x;
  const RowMap& mp = a.map;
  const int cvl = tid % mp.CVB, p = tid / mp.CVB;
  const int cv = blockIdx.y * mp.CVB + cvl;
  const bool lane_ok = p < mp.P && cv < mp.CV;
  const int c0 = cv * CH;
  const int CW = mp.CVB * CH;
  const bool want_sums = a.sums != nullptr;
  float acc[2][CH];
  float sc[CH], sh[CH], mu[CH], rs[CH], kA[CH], kB[CH], kC[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    acc[0][i] = acc[1][i] = 0.f;
    const bool ok = lane_ok;
    sc[i] = (a.scale && ok) ? a.scale[c0 + i] : 1.f;
    sh[i] = (a.shift && ok) ? a.shift[c0 + i] : 0.f;
    mu[i] = (a.mean && ok) ? a.mean[c0 + i] : 0.f;
    rs[i] = (a.rstd && ok) ? a.rstd[c0 + i] : 1.f;
    kA[i] = (a.cA && ok) ? a.cA[c0 + i] : 1.f;
    kB[i] = (a.cB && ok) ? a.cB[c0 + i] : 0.f;
    kC[i] = (a.cC && ok) ? a.cC[c0 + i] : 0.f;
  }
  const T* A = (const T*)a.a;
  const T* B = (const T*)a.b;
  T* O = (T*)a.out;
  if (lane_ok) {
    // each workgroup owns a contiguous row range; UNR rows per thread are loaded before any is
    // consumed (UNR x 1-2 x 16 B in flight per lane)
    constexpr int UNR = 4;
    const int64_t rows_wg = (a.M + gridDim.x - 1) / gridDim.x;
    const int64_t mlo = (int64_t)blockIdx.x * rows_wg, mhi = min(a.M, mlo + rows_wg);
    const bool has_row = a.row_scale || a.row_shift;
    // per-clip row scale / shift (dropout mask, SE terms): a thread's rows only ever move forward, so the vectors
    // are re-read when a row crosses into the next clip, not per row (they used to cost 64 registers and 4 integer
    // divisions per iteration: 155-169 VGPRs = 2-3 waves per SIMD on these streaming kernels)
    float rsc[CH], rsh[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) { rsc[i] = 1.f; rsh[i] = 0.f; }
    int64_t g_end = -1;                                     // first row of the next clip (-1: nothing loaded yet)
    if constexpr (MODE == 1 || MODE == 2 || MODE == 4) {
      // the per-clip vectors of this thread's FIRST row are fetched here, beside the per-channel vectors and ahead of the row
      // loads - not as a third dependent memory round trip after the rows have arrived (small tensors: one or two rows per
      // thread, the launch is nothing but its round trips)
      if (has_row && mlo + p < mhi) {
        const int64_t grp = (int64_t)((uint32_t)(mlo + p) / (uint32_t)a.rows_per_group);
        g_end = (grp + 1) * a.rows_per_group;
        if (a.row_scale) {
          const float* rp = a.row_scale + grp * a.C + c0;
#pragma unroll
          for (int i = 0; i < CH; ++i) rsc[i] = rp[i];
        }
        if (a.row_shift) {
          const float* rq = a.row_shift + grp * a.C + c0;
#pragma unroll
          for (int i = 0; i < CH; ++i) rsh[i] = rq[i];
        }
      }
    }
    // the activation is a compile-time constant inside the row loop (one uniform dispatch per kernel instead of a
    // switch per element in the unrolled body)
    auto run = [&](auto act_tag) {
    constexpr int ACT_C = decltype(act_tag)::value;
    const int act_ = ACT_C >= 0 ? ACT_C : a.act;
    for (int64_t mb = mlo + p; mb < mhi; mb += (int64_t)UNR * mp.P) {
      RawRow<T, CH> ra[UNR], rb[UNR];
      bool ok[UNR], dat[UNR];
      int64_t offo[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) dat[u] = true;
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int64_t m = mb + (int64_t)u * mp.P;
        ok[u] = m < mhi;
        const int64_t mc = ok[u] ? m : mlo + p;          // shadow a valid row, discard later
        const int64_t off = mc * a.C + c0;
        int64_t offc = off;                               // the row's offset in a compact operand (frame map)
        if (a.fm_full > 0) {
          const uint32_t n_ = (uint32_t)mc / (uint32_t)a.fm_full;
          const int64_t j = mc - (int64_t)n_ * a.fm_full - a.fm_lead;
          dat[u] = j >= 0 && j < a.fm_data;
          offc = ((int64_t)n_ * a.fm_data + (dat[u] ? j : 0)) * a.C + c0;      // (a padded row shadows a data row, value dropped)
        }
        if constexpr (MODE == 2) {
          if (a.dy_rows_per_group > 0) ra[u].load(A + (int64_t)((uint32_t)mc / (uint32_t)a.dy_rows_per_group) * a.C + c0);
          else ra[u].load(A + (a.a_compact ? offc : off));
        } else {
          ra[u].load(A + (a.a_compact ? offc : off));
        }
        if constexpr (MODE == 2 || MODE == 3 || MODE == 4) {
          if (B) rb[u].load(B + (a.b_compact ? offc : off));
          else rb[u].zero();
        }
        offo[u] = a.out_compact ? offc : off;
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        if (!ok[u]) continue;
        const int64_t m_u = mb + (int64_t)u * mp.P;
        const int64_t off = m_u * a.C + c0;
        if constexpr (MODE == 1 || MODE == 2 || MODE == 4) {
          if (has_row && (g_end < 0 || m_u >= g_end)) {
            const int64_t grp = (int64_t)((uint32_t)m_u / (uint32_t)a.rows_per_group);
            g_end = (grp + 1) * a.rows_per_group;
            if (a.row_scale) {
              const float* rp = a.row_scale + grp * a.C + c0;
#pragma unroll
              for (int i = 0; i < CH; ++i) rsc[i] = rp[i];
            }
            if (a.row_shift) {
              const float* rq = a.row_shift + grp * a.C + c0;
#pragma unroll
              for (int i = 0; i < CH; ++i) rsh[i] = rq[i];
            }
          }
        }
        float vo[CH];
        const bool a_zero = a.a_compact && !dat[u], b_zero = a.b_compact && !dat[u];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const float va_ = a_zero ? 0.f : ra[u].get(i);
          float vb_ = 0.f;
          if constexpr (MODE == 2 || MODE == 3 || MODE == 4) vb_ = b_zero ? 0.f : rb[u].get(i);
          (void)vb_;
          if constexpr (MODE == 0) {
            acc[0][i] += va_;
            acc[1][i] += va_ * va_;
          } else if constexpr (MODE == 1) {
            const float y = act_fwd((va_ - mu[i]) * sc[i] + sh[i], act_, a.slope) * rsc[i] + rsh[i];
            vo[i] = y;
            const float yr = to_f32(from_f32<T>(y));
            acc[0][i] += yr;
            acc[1][i] += yr * yr;
          } else if constexpr (MODE == 2) {
            const float pre = (vb_ - mu[i]) * sc[i] + sh[i];
            const float g = va_ * a.dy_mul * rsc[i] * act_bwd(pre, act_, a.slope);
            vo[i] = g;
            const float gr = O ? to_f32(from_f32<T>(g)) : g;
            acc[0][i] += gr;
            acc[1][i] += gr * (vb_ - mu[i]) * rs[i];
          } else {
            if constexpr (MODE == 3) {
              vo[i] = kA[i] * va_ + kB[i] * (vb_ - mu[i]) + kC[i];
            } else {
              // g = dy * mask * act'((z - mean)*scale + shift) recomputed here (BatchNorm backward then never
              // materialises g)
              const float zc = vb_ - mu[i];
              const float gg = va_ * rsc[i] * act_bwd(zc * sc[i] + sh[i], act_, a.slope);
              vo[i] = kA[i] * gg + kB[i] * zc + kC[i];
            }
          }
        }
        if constexpr (MODE != 0) {
          if (O && (!a.out_compact || dat[u])) stv<T, CH>(O + offo[u], vo);
        }
      }
    }
    };
    if constexpr (MODE == 1 || MODE == 2 || MODE == 4) {
      switch (a.act) {
        case PB_ACT_NONE: run(std::integral_constant<int, (int)PB_ACT_NONE>{}); break;
        case PB_ACT_HSWISH: run(std::integral_constant<int, (int)PB_ACT_HSWISH>{}); break;
        case PB_ACT_RELU: run(std::integral_constant<int, (int)PB_ACT_RELU>{}); break;
        default: run(std::integral_constant<int, -1>{}); break;
      }
    } else {
      run(std::integral_constant<int, -1>{});
    }
  }
  if constexpr (MODE != 3 && MODE != 4) {
    if (want_sums) {
      float* rep = a.sums + (size_t)blockIdx.x * 2 * a.C;      // slot = row-range index (gridDim.x <= PB_STAT_SLOTS)
      float* const outs[2] = {rep, rep + a.C};
      block_chan_reduce<CH, 2>(red, acc, lane_ok, mp.CVB, mp.P, blockIdx.y * CW, a.C, outs);
    }
  }
}

template <typename T, int MODE>
int launch_ew(EwArgs a, hipStream_t st) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const bool vec = (a.C % VEC) == 0;
  a.map = vec ? make_map<VEC>(a.C) : make_map<1>(a.C);
  // >= 4 rows per thread; reducing launches stay <= ~512 workgroups per channel chunk so that, with the
  // every workgroup owns one statistics slot (PB_STAT_SLOTS of them)
  PB_CHECK_ARG(a.M < 0x7fffffffLL, "elementwise kernels: M must fit 31 bits");
  // as many workgroups as there are row groups, up to the cap: small tensors (the 7x7 / 14x14 stages) were
  // running 1 workgroup per CU with 8 dependent rows per thread - latency-bound at ~1 TB/s
  static const int sum_rows = getenv("PB_EW_SUM_ROWS") ? atoi(getenv("PB_EW_SUM_ROWS")) : 16;
  int64_t gx = pb_cdiv(a.M, (int64_t)a.map.P * (a.sums ? sum_rows : 8));   // 8 rows per thread on large tensors (measured best)
  if (gx * a.map.chunks < 1024) {
    gx = pb_cdiv(a.M, (int64_t)a.map.P);                      // small tensor: spread it over ~1024 workgroups
    if (gx * a.map.chunks > 1024) gx = pb_cdiv(1024, a.map.chunks);
  }
  const int64_t cap = a.sums ? PB_STAT_SLOTS : pb_cdiv(8192, a.map.chunks);
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  dim3 grid((unsigned)gx, (unsigned)a.map.chunks);
  if (vec) hipLaunchKernelGGL((ew_kernel<T, VEC, MODE>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((ew_kernel<T, 1, MODE>), grid, dim3(256), 0, st, a);
  return 0;
}

template <int MODE>
int launch_ew_dt(const EwArgs& a, pb_dtype dtype, hipStream_t st) {
  return dtype == PB_BF16 ? launch_ew<bf16_t, MODE>(a, st) : launch_ew<float, MODE>(a, st);
}

// ------------------------------------------------------------------ row-group reduce (pool / SE dot)
struct RgArgs {
  const void* a;
  const void* b;
  float* out;          // partial sums [gridDim.x][G][C] (gridDim.x == 1: the result itself)
  int64_t R;
  int C;
  RowMap map;
};

template <typename T, int CH>
__global__ __launch_bounds__(256) void rowgroup_kernel(RgArgs a) {
  __shared__ float red[CH * 256];
  const int tid = threadIdx.x;
  const RowMap& mp = a.map;
  const int cvl = tid % mp.CVB, p = tid / mp.CVB;
  const int cv = blockIdx.y * mp.CVB + cvl;
  const bool lane_ok = p < mp.P && cv < mp.CV;
  const int c0 = cv * CH, CW = mp.CVB * CH;
  const int64_t g = blockIdx.z;
  float acc[1][CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) acc[0][i] = 0.f;
  const T* A = (const T*)a.a + g * a.R * a.C;
  const T* B = a.b ? (const T*)a.b + g * a.R * a.C : nullptr;
  if (lane_ok) {
    const int64_t stride = (int64_t)gridDim.x * mp.P;
    int64_t r = (int64_t)blockIdx.x * mp.P + p;
    if (B) {
      // four rows (8 independent 16-byte loads) in flight per thread: the kernel is latency-bound otherwise
      for (; r < a.R; r += 4 * stride) {
        float va[4][CH], vb[4][CH];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int64_t ru = min(r + u * stride, a.R - 1);          // (rows past the end re-read the last row, weight 0)
          ld<T, CH>(A + ru * a.C + c0, va[u]);
          ld<T, CH>(B + ru * a.C + c0, vb[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float wgt = r + u * stride < a.R ? 1.f : 0.f;
#pragma unroll
          for (int i = 0; i < CH; ++i) acc[0][i] += va[u][i] * vb[u][i] * wgt;
        }
      }
    } else {
      for (; r < a.R; r += stride) {
        float va[CH];
        ld<T, CH>(A + r * a.C + c0, va);
#pragma unroll
        for (int i = 0; i < CH; ++i) acc[0][i] += va[i];
      }
    }
  }
  float* const outs[1] = {a.out + ((int64_t)blockIdx.x * gridDim.z + g) * a.C};
  block_chan_reduce<CH, 1>(red, acc, lane_ok, mp.CVB, mp.P, blockIdx.y * CW, a.C, outs);
}

// Sum of the slots of one [slot][2][C] statistics buffer for the 32 channels of this workgroup (1024 threads):
// thread (cl, sg) adds the slots sg, sg+32, ... with four independent chains (the loads of a chain step are in flight
// together: 51 -> ~5 us per launch), the 32 groups are combined through LDS in index order - a fixed association.
constexpr int FIN_T = 1024, FIN_G = FIN_T / 32;
__device__ __forceinline__ void slot_sums(const float* stats, int slots, int C, int c0, double (*red)[2][32], double& s1, double& s2) {
  const int cl = threadIdx.x & 31, sg = threadIdx.x >> 5;
  const int c = c0 + cl;
  double a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
  if (c < C) {
    int r = sg;
    for (; r + 3 * FIN_G < slots; r += 4 * FIN_G) {
      float x1[4], x2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        x1[u] = stats[(size_t)(r + u * FIN_G) * 2 * C + c];
        x2[u] = stats[(size_t)(r + u * FIN_G) * 2 * C + C + c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { a1[u] += x1[u]; a2[u] += x2[u]; }
    }
    for (; r < slots; r += FIN_G) {
      a1[0] += stats[(size_t)r * 2 * C + c];
      a2[0] += stats[(size_t)r * 2 * C + C + c];
    }
  }
  red[sg][0][cl] = (a1[0] + a1[1]) + (a1[2] + a1[3]);
  red[sg][1][cl] = (a2[0] + a2[1]) + (a2[2] + a2[3]);
  __syncthreads();
  s1 = s2 = 0;
  if (sg == 0) {
#pragma unroll
    for (int g = 0; g < FIN_G; ++g) {
      s1 += red[g][0][cl];
      s2 += red[g][1][cl];
    }
  }
}

__global__ __launch_bounds__(FIN_T) void bn_finalize_kernel(const float* stats, int slots, double count, const float* gamma, const float* beta,
                                                          float eps, float momentum, float* running_mean, float* running_var,
                                                          int64_t* num_batches_tracked, int use_running, float* scale,
                                                          float* shift, float* mean_o, float* rstd_o, int C) {
  __shared__ double red[FIN_G][2][32];
  const int c = blockIdx.x * 32 + (threadIdx.x & 31);
  double s1 = 0, s2 = 0;
  if (!use_running) slot_sums(stats, slots, C, blockIdx.x * 32, red, s1, s2);
  if (threadIdx.x >= 32 || c >= C) return;
  double mean, var;
  if (use_running) {
    mean = running_mean[c];
    var = running_var[c];
  } else {
    mean = s1 / count;
    var = s2 / count - mean * mean;
    if (var < 0) var = 0;
    if (running_mean) running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
    if (running_var) {
      const double unb = count > 1 ? var * count / (count - 1) : var;
      running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
    }
    if (num_batches_tracked && c == 0) *num_batches_tracked += 1;        // nn.BatchNorm's step counter (train mode)
  }
  const double rstd = 1.0 / sqrt(var + (double)eps);
  const double g = gamma ? gamma[c] : 1.0, b = beta ? beta[c] : 0.0;
  scale[c] = (float)(g * rstd);
  shift[c] = (float)(b - mean * g * rstd);
  if (mean_o) mean_o[c] = (float)mean;
  if (rstd_o) rstd_o[c] = (float)rstd;
}

// Inference fold of MANY BatchNorm modules in one launch: desc[j] = {gamma, beta, running_mean, running_var, out, C, eps bits};
// out[0..C) = scale = gamma*rstd, out[C..2C) = shift = beta - mean*scale, out[2C..3C) = mean, out[3C..4C) = rstd.
__global__ __launch_bounds__(256) void bn_fold_batch_kernel(const int64_t* desc) {
  const int64_t* d = desc + (size_t)blockIdx.x * 8;
  const float* gamma = reinterpret_cast<const float*>(d[0]);
  const float* beta = reinterpret_cast<const float*>(d[1]);
  const float* rm = reinterpret_cast<const float*>(d[2]);
  const float* rv = reinterpret_cast<const float*>(d[3]);
  float* out = reinterpret_cast<float*>(d[4]);
  const int C = (int)d[5];
  double eps;
  { int64_t bits = d[6]; __builtin_memcpy(&eps, &bits, 8); }
  for (int c = threadIdx.x; c < C; c += 256) {
    const double mean = rm[c], var = rv[c];
    const double rstd = 1.0 / sqrt(var + eps);
    const double g = gamma ? gamma[c] : 1.0, b = beta ? beta[c] : 0.0;
    out[c] = (float)(g * rstd);
    out[C + c] = (float)(b - mean * g * rstd);
    out[2 * C + c] = (float)mean;
    out[3 * C + c] = (float)rstd;
  }
}

__global__ __launch_bounds__(FIN_T) void bn_bwd_finalize_kernel(const float* sums, int slots, double count, const float* gamma, const float* mean,
                                                              const float* rstd, int training, float* cA, float* cB, float* cC,
                                                              float* dgamma, float* dbeta, int C) {
  __shared__ double red[FIN_G][2][32];
  const int c = blockIdx.x * 32 + (threadIdx.x & 31);
  double sg = 0, sgx = 0;
  slot_sums(sums, slots, C, blockIdx.x * 32, red, sg, sgx);
  if (threadIdx.x >= 32 || c >= C) return;
  const double g = gamma ? gamma[c] : 1.0, r = rstd[c];
  if (training) {
    // dz = A*g + B*(z - mean) + C: the mean is subtracted from z in the consumer, never folded into C
    // (a channel with |mean| >> std would lose the difference to cancellation)
    cA[c] = (float)(g * r);
    cB[c] = (float)(-g * r * r * (sgx / count));
    cC[c] = (float)(-g * r * (sg / count));
  } else {
    cA[c] = (float)(g * r);
    cB[c] = 0.f;
    cC[c] = 0.f;
  }
  if (dgamma) dgamma[c] = (float)sgx;
  if (dbeta) dbeta[c] = (float)sg;
}

__global__ void transpose_f32_kernel(const float* in, float* out, int R, int C) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)R * C) return;
  const int r = (int)(i / C), c = (int)(i - (int64_t)r * C);
  out[(int64_t)c * R + r] = in[i];
}

int check_mc(int64_t M, int C, pb_dtype dtype, const char* who) {
  PB_CHECK_ARG(M >= 0 && C > 0, "%s: bad sizes M=%lld C=%d", who, (long long)M, C);
  PB_CHECK_ARG(dtype == PB_F32 || dtype == PB_BF16, "%s: bad dtype %d", who, (int)dtype);
  return 0;
}

}  // namespace

extern "C" int pb_chan_stats(const void* x, int64_t M, int C, pb_dtype dtype, float* stats, int device, void* stream) {
  PB_CHECK_ARG(x && stats, "pb_chan_stats: null pointer");
  if (int rc = check_mc(M, C, dtype, "pb_chan_stats")) return rc;
  if (M == 0) return 0;
  PB_ENTER(device);
  EwArgs a = {};
  a.a = x; a.M = M; a.C = C; a.sums = stats; a.rows_per_group = 1; a.dy_mul = 1.f;
  launch_ew_dt<0>(a, dtype, (hipStream_t)stream);
  PB_LAUNCH_CHECK();
  return 0;
}

extern "C" int pb_bn_finalize(const float* stats, int slots, double count, const float* gamma, const float* beta, float eps,
                              float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                              int use_running, float* scale, float* shift, float* mean, float* rstd, int C, int device,
                              void* stream) {
  PB_CHECK_ARG(scale && shift && C > 0, "pb_bn_finalize: null output / bad C");
  PB_CHECK_ARG(use_running ? (running_mean && running_var) : (stats && count > 0 && slots >= 1),
               "pb_bn_finalize: missing statistics (use_running=%d)", use_running);
  PB_ENTER(device);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)pb_cdiv(C, 32)), dim3(FIN_T), 0, (hipStream_t)stream, stats,
                     slots, count, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked, use_running, scale,
                     shift, mean, rstd, C);
  PB_LAUNCH_CHECK();
  return 0;
}

extern "C" int pb_bn_fold_batch(const int64_t* desc, int count, int device, void* stream) {
  PB_CHECK_ARG(desc && count > 0 && count <= 65535, "pb_bn_fold_batch: bad arguments");
  PB_ENTER(device);
  hipLaunchKernelGGL(bn_fold_batch_kernel, dim3((unsigned)count), dim3(256), 0, (hipStream_t)stream, desc);
  PB_LAUNCH_CHECK();
  return 0;
}

static int check_frame_map(int64_t M, int64_t fm_full, int64_t fm_data, int64_t fm_lead, const char* who) {
  if (fm_full <= 0) return 0;
  PB_CHECK_ARG(M % fm_full == 0 && fm_data > 0 && fm_lead >= 0 && fm_lead + fm_data <= fm_full && M < 0x7fffffffLL,
               "%s: bad frame map (M %lld, rows per clip %lld, data rows %lld at %lld)", who, (long long)M, (long long)fm_full,
               (long long)fm_data, (long long)fm_lead);
  return 0;
}

extern "C" int pb_affine_act_fwd(const void* x, void* y, int64_t M, int C, pb_dtype dtype, const float* scale,
                                 const float* shift, const float* mean, pb_act act, float slope,
                                 const float* row_scale, const float* row_shift, int64_t rows_per_group, float* stats,
                                 int64_t fm_full, int64_t fm_data, int64_t fm_lead, int device, void* stream) {
  PB_CHECK_ARG(x && y, "pb_affine_act_fwd: null pointer");
  if (int rc = check_frame_map(M, fm_full, fm_data, fm_lead, "pb_affine_act_fwd")) return rc;
  if (int rc = check_mc(M, C, dtype, "pb_affine_act_fwd")) return rc;
  PB_CHECK_ARG(!(row_scale || row_shift) || rows_per_group > 0, "pb_affine_act_fwd: row_scale/shift need rows_per_group > 0");
  if (M == 0) return 0;
  PB_ENTER(device);
  EwArgs a = {};
  a.a = x; a.out = y; a.M = M; a.C = C; a.scale = scale; a.shift = shift; a.mean = mean; a.act = (int)act; a.slope = slope;
  a.row_scale = row_scale; a.row_shift = row_shift; a.rows_per_group = rows_per_group > 0 ? rows_per_group : 1;
  a.sums = stats; a.dy_mul = 1.f;
  a.fm_full = fm_full; a.fm_data = fm_data; a.fm_lead = fm_lead; a.a_compact = fm_full > 0;    // x compact -> y full
  launch_ew_dt<1>(a, dtype, (hipStream_t)stream);
  PB_LAUNCH_CHECK();
  return 0;
}

extern "C" int pb_act_bwd_reduce(const void* dy, const void* z, void* g_out, int64_t M, int C, pb_dtype dtype,
                                 const float* scale, const float* shift, pb_act act, float slope,
                                 const float* row_scale, int64_t rows_per_group, int64_t dy_rows_per_group,
                                 float dy_mul, const float* mean, const float* rstd, float* sums, int64_t fm_full,
                                 int64_t fm_data, int64_t fm_lead, int device, void* stream) {
  PB_CHECK_ARG(dy && z, "pb_act_bwd_reduce: null pointer");
  if (int rc = check_frame_map(M, fm_full, fm_data, fm_lead, "pb_act_bwd_reduce")) return rc;
  if (int rc = check_mc(M, C, dtype, "pb_act_bwd_reduce")) return rc;
  PB_CHECK_ARG(!row_scale || rows_per_group > 0, "pb_act_bwd_reduce: row_scale needs rows_per_group > 0");
  if (M == 0) return 0;
  PB_ENTER(device);
  EwArgs a = {};
  a.a = dy; a.b = z; a.out = g_out; a.M = M; a.C = C; a.scale = scale; a.shift = shift; a.act = (int)act;
  a.slope = slope; a.row_scale = row_scale; a.rows_per_group = rows_per_group > 0 ? rows_per_group : 1;
  a.dy_rows_per_group = dy_rows_per_group; a.dy_mul = dy_mul; a.mean = mean; a.rstd = rstd; a.sums = sums;
  a.fm_full = fm_full; a.fm_data = fm_data; a.fm_lead = fm_lead;      // dy full, z compact -> g compact, sums over all rows
  a.b_compact = a.out_compact = fm_full > 0;
  launch_ew_dt<2>(a, dtype, (hipStream_t)stream);
  PB_LAUNCH_CHECK();
  return 0;
}

extern "C" int pb_bn_bwd_finalize(const float* sums, int slots, double count, const float* gamma, const float* mean,
                                  const float* rstd, int training, float* coefA, float* coefB, float* coefC,
                                  float* dgamma, float* dbeta, int C, int device, void* stream) {
  PB_CHECK_ARG(sums && slots >= 1 && mean && rstd && coefA && coefB && coefC && C > 0 && count > 0, "pb_bn_bwd_finalize: bad args");
  PB_ENTER(device);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)pb_cdiv(C, 32)), dim3(FIN_T), 0, (hipStream_t)stream,
                     sums, slots, count, gamma, mean, rstd, training, coefA, coefB, coefC, dgamma, dbeta, C);
  PB_LAUNCH_CHECK();
  return 0;
}

extern "C" int pb_affine2(const void* g, const void* z, void* out, int64_t M, int C, pb_dtype dtype, const float* A,
                          const float* B, const float* Cc, const float* mean, const float* row_scale,
                          int64_t rows_per_group, const float* act_scale, const float* act_shift, pb_act act,
                          float slope, int device, void* stream) {
  PB_CHECK_ARG(act == PB_ACT_NONE || z, "pb_affine2: an activation needs z");
  PB_CHECK_ARG(g && out && A, "pb_affine2: null pointer");
  PB_CHECK_ARG(z || !B, "pb_affine2: B given without z");
  PB_CHECK_ARG(!row_scale || rows_per_group > 0, "pb_affine2: row_scale needs rows_per_group > 0");
  if (int rc = check_mc(M, C, dtype, "pb_affine2")) return rc;
  if (M == 0) return 0;
  PB_ENTER(device);
  EwArgs a = {};
  a.a = g; a.b = z; a.out = out; a.M = M; a.C = C; a.cA = A; a.cB = B; a.cC = Cc; a.mean = mean;
  a.row_scale = row_scale; a.rows_per_group = rows_per_group > 0 ? rows_per_group : 1;
  a.scale = act_scale; a.shift = act_shift; a.act = (int)act; a.slope = slope;
  a.dy_mul = 1.f;
  if (row_scale || act != PB_ACT_NONE) launch_ew_dt<4>(a, dtype, (hipStream_t)stream);
  else launch_ew_dt<3>(a, dtype, (hipStream_t)stream);
  PB_LAUNCH_CHECK();
  return 0;
}

namespace {
// row-group reduction plan: gx workgroups along the rows of a group, each writing one partial row
int64_t rowgroup_gx(int64_t G, int64_t rows_per_group, int C, pb_dtype dtype, RowMap* map_out) {
  const int vec = dtype == PB_BF16 ? 8 : 4;
  const bool v = (C % vec) == 0;
  const RowMap map = v ? (vec == 8 ? make_map<8>(C) : make_map<4>(C)) : make_map<1>(C);
  int64_t gx = pb_cdiv(rows_per_group, map.P);
  const int64_t cap = pb_cdiv(2048, (int64_t)map.chunks * G);
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  if (map_out) *map_out = map;
  return gx;
}
}  // namespace

extern "C" int pb_rowgroup_slots(int64_t G, int64_t rows_per_group, int C, pb_dtype dtype) {
  if (G <= 0 || rows_per_group <= 0 || C <= 0) return 0;
  return (int)rowgroup_gx(G, rows_per_group, C, dtype, nullptr);
}

extern "C" int pb_rowgroup_partials(const void* a_, const void* b_, float* part, int64_t G, int64_t rows_per_group, int C,
                                    pb_dtype dtype, int device, void* stream) {
  PB_CHECK_ARG(a_ && part, "pb_rowgroup_partials: null pointer");
  PB_CHECK_ARG(G > 0 && G <= 65535 && rows_per_group > 0 && C > 0, "pb_rowgroup_partials: bad sizes");
  PB_CHECK_ARG(dtype == PB_F32 || dtype == PB_BF16, "pb_rowgroup_partials: bad dtype");
  PB_ENTER(device);
  RgArgs a;
  a.a = a_; a.b = b_; a.R = rows_per_group; a.C = C; a.out = part;
  const int vec = dtype == PB_BF16 ? 8 : 4;
  const bool v = (C % vec) == 0;
  const int64_t gx = rowgroup_gx(G, rows_per_group, C, dtype, &a.map);
  dim3 grid((unsigned)gx, (unsigned)a.map.chunks, (unsigned)G);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == PB_BF16) {
    if (v) hipLaunchKernelGGL((rowgroup_kernel<bf16_t, 8>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((rowgroup_kernel<bf16_t, 1>), grid, dim3(256), 0, st, a);
  } else {
    if (v) hipLaunchKernelGGL((rowgroup_kernel<float, 4>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((rowgroup_kernel<float, 1>), grid, dim3(256), 0, st, a);
  }
  PB_LAUNCH_CHECK();
  return 0;
}

extern "C" int64_t pb_rowgroup_reduce_ws_bytes(int64_t G, int64_t rows_per_group, int C, pb_dtype dtype) {
  if (G <= 0 || rows_per_group <= 0 || C <= 0) return 0;
  const int64_t gx = rowgroup_gx(G, rows_per_group, C, dtype, nullptr);
  return gx > 1 ? gx * G * C * 4 : 0;
}

extern "C" int pb_rowgroup_reduce(const void* a_, const void* b_, float* out, int64_t G, int64_t rows_per_group,
                                  int C, pb_dtype dtype, void* ws, int64_t ws_bytes, int device, void* stream) {
  PB_CHECK_ARG(a_ && out, "pb_rowgroup_reduce: null pointer");
  PB_CHECK_ARG(G > 0 && G <= 65535 && rows_per_group > 0 && C > 0, "pb_rowgroup_reduce: bad sizes");
  PB_CHECK_ARG(dtype == PB_F32 || dtype == PB_BF16, "pb_rowgroup_reduce: bad dtype");
  PB_ENTER(device);
  RgArgs a;
  a.a = a_; a.b = b_; a.R = rows_per_group; a.C = C;
  const int vec = dtype == PB_BF16 ? 8 : 4;
  const bool v = (C % vec) == 0;
  const int64_t gx = rowgroup_gx(G, rows_per_group, C, dtype, &a.map);
  PB_CHECK_ARG(gx == 1 || (ws && ws_bytes >= gx * G * C * 4), "pb_rowgroup_reduce: workspace of %lld bytes needed",
               (long long)(gx * G * C * 4));
  a.out = gx == 1 ? out : (float*)ws;
  dim3 grid((unsigned)gx, (unsigned)a.map.chunks, (unsigned)G);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == PB_BF16) {
    if (v) hipLaunchKernelGGL((rowgroup_kernel<bf16_t, 8>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((rowgroup_kernel<bf16_t, 1>), grid, dim3(256), 0, st, a);
  } else {
    if (v) hipLaunchKernelGGL((rowgroup_kernel<float, 4>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((rowgroup_kernel<float, 1>), grid, dim3(256), 0, st, a);
  }
  PB_LAUNCH_CHECK();
  if (gx > 1) {
    pb_reduce_job job = {(const float*)ws, out, G * C, G * C, (int)gx, 0};
    if (pb_launch_partials_reduce(&job, 1, st)) { pb_set_error("pb_rowgroup_reduce: combine launch failed"); return 2; }
  }
  return 0;
}

extern "C" int pb_transpose_f32(const float* in, float* out, int R, int C, int device, void* stream) {
  PB_CHECK_ARG(in && out && R > 0 && C > 0, "pb_transpose_f32: bad args");
  PB_ENTER(device);
  hipLaunchKernelGGL(transpose_f32_kernel, dim3((unsigned)pb_cdiv((int64_t)R * C, 256)), dim3(256), 0,
                     (hipStream_t)stream, in, out, R, C);
  PB_LAUNCH_CHECK();
  return 0;
}
