// Pointwise (1x1x1) convolution = row-major GEMM over NDHWC activations.
//
//   pw_gemm_kernel   Y[M,N] = P(X)[M,K] * Wl[N,K]^T   forward and backward-data
//   pw_wgrad_kernel  dW[N,K] += dY[M,N]^T * P(X)[M,K] weight gradient
//
// Both are HBM-bound on this workload (K,N <= 960, arithmetic intensity 16-45 flop/B against a
// ridge near 300): the design goal is one pass over X and one over Y with full-line accesses;
// MFMA (16x16x32 bf16 / exact-f32 16x16x4) only has to keep out of the way.
//   * a workgroup owns 128 rows x BN columns, BN in {32,64,96,128,192} chosen per layer so that
//     N <= 192 is a single column tile (X is then read exactly once);
//   * column tiles of one row panel sit on the same XCD (block id % 8) so that X re-reads for
//     N > 192 hit that XCD's L2;
//   * X rows are contiguous [M,K], so a 128-row panel chunk is loaded with 16 B/lane, transformed
//     in registers (BN-apply/activation/SE scale prologue) and staged in LDS; weights (fp32
//     masters, L2 resident) are converted while staging;
//   * the accumulators go back through LDS so that Y is written as whole 16 B/lane row segments.
#include "pb_common.h"
#include <stdlib.h>

namespace {

constexpr int BM = 128;            // rows per workgroup (4 waves x 32)
constexpr int LDB = 144;           // LDS row stride in bytes: 128 B payload + 16 B pad

struct PwArgs {
  const void* X;
  const float* W;
  void* Y;
  int64_t M;
  int K, N, ldw, w_transposed;
  const float* in_scale;
  const float* in_shift;
  int in_act;
  float in_slope;
  const float* row_scale;
  int64_t rows_per_group;
  const float* out_scale;
  const float* out_shift;
  int out_act;
  float out_slope;
  float* stats;
  int n_tiles;
  int64_t m_tiles;
  // implicit-GEMM dense convolution (CONV variants): K = Cin per tap, the K loop runs over
  // (tap, channel stage) and X rows are the tap-shifted input voxels (zero outside the clip);
  // W is [N][taps][Cin] (ldw = taps*Cin)
  int cTi, cHi, cWi, cTo, cHo, cWo, ckt, ckh, ckw, cst, csh, csw, cpt, cph, cpw;
};

// Prologue on the X operand, split so that a prefetch only ISSUES loads: `RowScale` carries the
// chunk's per-(clip, channel) squeeze-excite factors fetched next to the chunk; the arithmetic
// (affine, activation, scale) runs when the chunk is committed to LDS, a stage later.
template <typename T> struct RowScale {
  float v[ElemTraits<T>::VEC];
  __device__ __forceinline__ void ones() {
#pragma unroll
    for (int i = 0; i < ElemTraits<T>::VEC; ++i) v[i] = 1.f;
  }
};

template <typename T>
__device__ __forceinline__ RowScale<T> fetch_rowscale(const PwArgs& a, int64_t group, int k) {
  constexpr int VEC = ElemTraits<T>::VEC;
  RowScale<T> r;
#pragma unroll
  for (int i = 0; i < VEC; ++i) r.v[i] = 1.f;
  if (a.row_scale) {
    const float* rs = a.row_scale + group * a.K + k;
    if ((a.K & 3) == 0 && k + VEC <= a.K) {
#pragma unroll
      for (int q = 0; q < VEC / 4; ++q) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(rs + 4 * q);
        r.v[4 * q] = t[0]; r.v[4 * q + 1] = t[1]; r.v[4 * q + 2] = t[2]; r.v[4 * q + 3] = t[3];
      }
    } else {
#pragma unroll
      for (int i = 0; i < VEC; ++i)
        if (k + i < a.K) r.v[i] = rs[i];
    }
  }
  return r;
}

template <typename T>
__device__ __forceinline__ Vec16<T> apply_prologue(Vec16<T> v, const RowScale<T>& rs, const PwArgs& a, int k) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const bool affine = a.in_scale || a.in_shift || a.in_act != PB_ACT_NONE;
  if (!affine) {            // squeeze-excite scale only (the training path): branch-free
#pragma unroll
    for (int i = 0; i < VEC; ++i) v.set(i, v.get(i) * rs.v[i]);
    return v;
  }
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    float x = v.get(i);
    if (k + i < a.K) {
      if (a.in_scale) x *= a.in_scale[k + i];
      if (a.in_shift) x += a.in_shift[k + i];
      x = act_fwd(x, a.in_act, a.in_slope);
    }
    v.set(i, x * rs.v[i]);
  }
  return v;
}

// clip index of row m (64-bit division kept out of the per-chunk path: rows of a 128-row panel span at
// most two clips when rows_per_group >= 128)
__device__ __forceinline__ int64_t row_group(int64_t m, int64_t m0, int64_t g0, int64_t g0_end, int64_t rpg) {
  return rpg >= 128 ? (m >= g0_end ? g0 + 1 : g0) : m / rpg;
}

// Load VEC consecutive elements X[m][k..k+VEC) (zero beyond K); vector access when aligned.
template <typename T>
__device__ __forceinline__ Vec16<T> load_row_chunk(const T* X, int64_t m, int k, int K, bool vec_ok) {
  constexpr int VEC = ElemTraits<T>::VEC;
  if (vec_ok) return vec_load<T>(X + m * K + k);
  Vec16<T> v = vec_zero<T>();
#pragma unroll
  for (int i = 0; i < VEC; ++i)
    if (k + i < K) v.set(i, to_f32(X[m * K + k + i]));
  return v;
}

// Persistent workgroups walk (row panel, column tile) pairs; the global loads of the NEXT K stage
// (or of the next tile's first stage) are issued into registers before the MFMAs of the current
// stage, so HBM/L2 latency hides behind compute and the epilogue.  When K fits one stage and there
// is a single column tile, the converted weight tile stays resident in LDS for the whole launch.
// ALIGNED: K % VEC == 0, N % VEC == 0, ldw % 4 == 0 -> every access is a 16-byte vector and the
// scalar fallbacks are compiled out of the hot path.  PRO: a prologue (affine / activation / row
// scale) is applied to X while staging.
template <typename T, int NT, bool TRANS, bool ALIGNED, bool PRO, bool CONV = false>
__global__ __launch_bounds__(256) void pw_gemm_kernel(PwArgs a) {
  constexpr int VEC = ElemTraits<T>::VEC;
  constexpr int KS = 128 / (int)sizeof(T);     // K elements staged per step (64 bf16 / 32 f32)
  constexpr int MK = 128 / (int)sizeof(T) / 2; // K consumed per fragment read (32 bf16 / 16 f32)
  constexpr int BN = NT * 16;
  constexpr int EPI_LD = BN + VEC;
  constexpr int EPI_BYTES = 4 * 16 * EPI_LD * (int)sizeof(T);
  constexpr int A_BYTES = BM * LDB > EPI_BYTES ? BM * LDB : EPI_BYTES;   // A tile, reused by the epilogue
  constexpr int W_BYTES = BN * LDB;
  constexpr int ACH = 4;                               // BM * 8 chunks / 256 threads
  constexpr int WCH = (BN * 8 + 255) / 256;            // weight chunks per thread (row-major W)
  constexpr int NQ = BN / 4;
  constexpr int WTQ = (KS * NQ + 255) / 256;           // float4 loads per thread (transposed W)
  constexpr int RS_MAXK = 1536;                          // squeeze-excite scale rows cached per tile: [2 clips][K]
  constexpr int RS_BYTES = PRO ? 2 * RS_MAXK * 4 : 0;
  __shared__ __attribute__((aligned(16))) unsigned char smem[A_BYTES + W_BYTES + 4 * BN * 2 * 4 + RS_BYTES];
  unsigned char* As = smem;
  unsigned char* Ws = smem + A_BYTES;
  float* sred = (float*)(smem + A_BYTES + W_BYTES);   // [4 waves][BN][2]
  float* rsS = (float*)(smem + A_BYTES + W_BYTES + 4 * BN * 2 * 4);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T* X = (const T*)a.X;
  T* Y = (T*)a.Y;
  const bool xvec = ALIGNED || (a.K % VEC) == 0;
  const bool wvec = ALIGNED || (a.ldw % 4) == 0;
  const bool yvec = ALIGNED || (a.N % VEC) == 0;
  const int nk = (a.K + KS - 1) / KS;                       // channel stages (per tap when CONV)
  const int taps = CONV ? a.ckt * a.ckh * a.ckw : 1;
  const int nst = nk * taps;                                // stages per tile
  const bool w_cached = !CONV && nk == 1 && a.n_tiles == 1;
  // all tile arithmetic in 32 bits (the host checks the tile count), single-column-tile fast path
  const int n_tiles = a.n_tiles, m_tiles = (int)a.m_tiles;
  const int total = n_tiles == 1 ? m_tiles : ((m_tiles + 7) / 8) * 8 * n_tiles;

  // tile id -> (row panel, column tile): ids with equal t % 8 run on blocks with equal blockIdx % 8
  // (gridDim is a multiple of 8), i.e. on one XCD: all column tiles of a row panel share its L2.
  auto tile_mt = [&](int t) { return n_tiles == 1 ? t : ((t >> 3) / n_tiles) * 8 + (t & 7); };
  auto tile_nt = [&](int t) { return n_tiles == 1 ? 0 : (t >> 3) % n_tiles; };
  auto next_valid = [&](int t) {
    while (t < total && tile_mt(t) >= m_tiles) t += gridDim.x;
    return t;
  };

  Vec16<T> ra[ACH];
  RowScale<T> rsa[PRO ? ACH : 1];
  // row scales come from an LDS cache of the tile's (at most two) clips when they fit
  const bool rs_lds = PRO && a.row_scale && a.K <= RS_MAXK && (a.K & 3) == 0 && a.rows_per_group >= BM;
  int64_t cur_g0_end = 0;
  Vec16<T> rw[TRANS ? 1 : WCH];
  f32x4 rt[TRANS ? WTQ : 1];

  // a stage holds kpad = MK or 2*MK (= KS) columns, i.e. 4 or 8 16-byte chunks per row: chunk ->
  // (row, col) is a shift and a mask
  auto stage_dims = [&](int k0, int& kpad, int& csh) {
    const int kc = min(KS, a.K - k0);
    kpad = kc > MK ? KS : MK;
    csh = kc > MK ? 3 : 2;
  };
  auto fetch_a = [&](int t, int ks_flat) {
    const int tap = CONV ? ks_flat / nk : 0;
    const int ks = CONV ? ks_flat - tap * nk : ks_flat;
    const int k0 = ks * KS;
    int kpad, csh;
    stage_dims(k0, kpad, csh);
    const int64_t m0 = (int64_t)tile_mt(t) * BM;
    int dt = 0, dh = 0, dw_ = 0;
    if constexpr (CONV) {
      dw_ = tap % a.ckw;
      dh = (tap / a.ckw) % a.ckh;
      dt = tap / (a.ckw * a.ckh);
    }
    int64_t g0 = 0, g0_end = 0;
    if constexpr (PRO) {
      if (a.row_scale) {
        g0 = m0 / a.rows_per_group;
        g0_end = (g0 + 1) * a.rows_per_group;
      }
    }
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
      const int f = tid + i * 256;
      ra[i] = vec_zero<T>();
      if constexpr (PRO) rsa[i].ones();
      if (f < (BM << csh)) {
        const int r = f >> csh, c = f & ((1 << csh) - 1);
        const int64_t m = m0 + r;
        const int k = k0 + c * VEC;
        if constexpr (CONV) {
          if (m < a.M && k < a.K) {
            // output voxel -> tap-shifted input voxel
            int64_t q = m;
            const int wo = (int)(q % a.cWo); q /= a.cWo;
            const int ho = (int)(q % a.cHo); q /= a.cHo;
            const int to = (int)(q % a.cTo);
            const int64_t n = q / a.cTo;
            const int ti = to * a.cst - a.cpt + dt, hi = ho * a.csh - a.cph + dh, wi = wo * a.csw - a.cpw + dw_;
            if (ti >= 0 && ti < a.cTi && hi >= 0 && hi < a.cHi && wi >= 0 && wi < a.cWi)
              ra[i] = vec_load<T>(X + (((n * a.cTi + ti) * a.cHi + hi) * a.cWi + wi) * a.K + k);
          }
        } else if (m < a.M && k < a.K) {
          if constexpr (ALIGNED) ra[i] = vec_load<T>(X + m * a.K + k);
          else ra[i] = load_row_chunk<T>(X, m, k, a.K, xvec);
          if constexpr (PRO) {
            if (!rs_lds) rsa[i] = fetch_rowscale<T>(a, row_group(m, m0, g0, g0_end, a.rows_per_group), k);
          }
        }
      }
    }
  };
  auto commit_a = [&](int t, int ks_flat) {
    const int ks = CONV ? ks_flat % nk : ks_flat;
    int kpad, csh;
    stage_dims(ks * KS, kpad, csh);
    const int64_t m0 = (int64_t)tile_mt(t) * BM;
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
      const int f = tid + i * 256;
      if (f < (BM << csh)) {
        const int r = f >> csh, c = f & ((1 << csh) - 1);
        if constexpr (PRO) {
          const int k = ks * KS + c * VEC;
          if (rs_lds) {
            if (k < a.K) {
              const float* src = rsS + ((m0 + r) >= cur_g0_end ? RS_MAXK : 0) + k;
#pragma unroll
              for (int q = 0; q < VEC / 4; ++q) {
                const f32x4 tq = *reinterpret_cast<const f32x4*>(src + 4 * q);
                rsa[i].v[4 * q] = tq[0]; rsa[i].v[4 * q + 1] = tq[1]; rsa[i].v[4 * q + 2] = tq[2]; rsa[i].v[4 * q + 3] = tq[3];
              }
            } else {
              rsa[i].ones();
            }
          }
          ra[i] = apply_prologue<T>(ra[i], rsa[i], a, k);
        }
        vec_store<T>((T*)(As + r * LDB) + c * VEC, ra[i]);
      }
    }
  };
  auto fetch_w = [&](int t, int ks_flat) {
    const int tap = CONV ? ks_flat / nk : 0;
    const int ks = CONV ? ks_flat - tap * nk : ks_flat;
    const int k0 = ks * KS, n0 = tile_nt(t) * BN;
    const int wcol0 = CONV ? tap * a.K : 0;                 // column offset of this tap in W
    int kpad, csh;
    stage_dims(k0, kpad, csh);
    if constexpr (!TRANS) {
#pragma unroll
      for (int i = 0; i < WCH; ++i) {
        const int f = tid + i * 256;
        rw[i] = vec_zero<T>();
        if (f < (BN << csh)) {
          const int r = f >> csh, c = f & ((1 << csh) - 1);
          const int n = n0 + r, k = k0 + c * VEC;
          if (n < a.N && k < a.K) {
            const float* src = a.W + (int64_t)n * a.ldw + wcol0 + k;
            if (ALIGNED || (wvec && k + VEC <= a.K)) {
#pragma unroll
              for (int q = 0; q < VEC / 4; ++q) {
                const f32x4 tq = *reinterpret_cast<const f32x4*>(src + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) rw[i].set(4 * q + e, tq[e]);
              }
            } else if constexpr (!ALIGNED) {
#pragma unroll
              for (int e = 0; e < VEC; ++e)
                if (k + e < a.K) rw[i].set(e, src[e]);
            }
          }
        }
      }
    } else {
      // Wl[n][k] = w[k*ldw + n]: 4 consecutive n per load (coalesced along n)
#pragma unroll
      for (int i = 0; i < WTQ; ++i) {
        const int f = tid + i * 256;
        rt[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (f < kpad * NQ) {
          const int kl = f / NQ, nq = f - kl * NQ;
          const int n = n0 + nq * 4, k = k0 + kl;
          if (k < a.K && n < a.N) {
            const float* src = a.W + (int64_t)k * a.ldw + n;
            if (ALIGNED || (wvec && n + 4 <= a.N)) {
              rt[i] = *reinterpret_cast<const f32x4*>(src);
            } else if constexpr (!ALIGNED) {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (n + e < a.N) rt[i][e] = src[e];
            }
          }
        }
      }
    }
  };
  auto commit_w = [&](int ks_flat) {
    const int ks = CONV ? ks_flat % nk : ks_flat;
    int kpad, csh;
    stage_dims(ks * KS, kpad, csh);
    if constexpr (!TRANS) {
#pragma unroll
      for (int i = 0; i < WCH; ++i) {
        const int f = tid + i * 256;
        if (f < (BN << csh)) {
          const int r = f >> csh, c = f & ((1 << csh) - 1);
          vec_store<T>((T*)(Ws + r * LDB) + c * VEC, rw[i]);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < WTQ; ++i) {
        const int f = tid + i * 256;
        if (f < kpad * NQ) {
          const int kl = f / NQ, nq = f - kl * NQ;
#pragma unroll
          for (int e = 0; e < 4; ++e) ((T*)(Ws + (nq * 4 + e) * LDB))[kl] = from_f32<T>(rt[i][e]);
        }
      }
    }
  };

  f32x4 acc[2][NT];
  float csum[NT], csq[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    csum[j] = csq[j] = 0.f;
    acc[0][j] = acc[1][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int frow = lane & 15, fk = lane >> 4;
  const int ccol = lane & 15, rgrp = lane >> 4;
  const bool plain = !a.out_scale && !a.out_shift && a.out_act == PB_ACT_NONE;
  float osc[NT], osh[NT];
  int osc_n0 = -1;
#pragma unroll
  for (int j = 0; j < NT; ++j) { osc[j] = 1.f; osh[j] = 0.f; }

  auto flush_stats = [&](int n0) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      float s = csum[j], q = csq[j];
      s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
      q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
      if (lane < 16) {
        sred[(wave * BN + j * 16 + lane) * 2 + 0] = s;
        sred[(wave * BN + j * 16 + lane) * 2 + 1] = q;
      }
      csum[j] = csq[j] = 0.f;
    }
    __syncthreads();
    float* st = a.stats + (size_t)blockIdx.x * 2 * a.N;      // this workgroup's own slot (grid <= PB_STAT_SLOTS): plain read-modify-write
    for (int c = tid; c < BN; c += 256) {
      const int n = n0 + c;
      if (n < a.N) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          s += sred[(w * BN + c) * 2 + 0];
          q += sred[(w * BN + c) * 2 + 1];
        }
        st[n] += s;
        st[a.N + n] += q;
      }
    }
    __syncthreads();
  };

  int t = next_valid((int)blockIdx.x);
  if (t >= total) return;
  if (w_cached) {
    fetch_w(t, 0);
    commit_w(0);
  }
  int ks = 0;
  fetch_a(t, 0);
  if (!w_cached) fetch_w(t, 0);
  while (t < total) {
    if constexpr (PRO) {
      if (rs_lds && ks == 0) {
        // this tile's clip rows of the scale matrix -> LDS (rows of a 128-row tile span <= 2 clips)
        const int64_t m0 = (int64_t)tile_mt(t) * BM;
        const int64_t g0 = m0 / a.rows_per_group;
        const int64_t ng = (a.M + a.rows_per_group - 1) / a.rows_per_group;
        cur_g0_end = (g0 + 1) * a.rows_per_group;
        for (int i = tid; i < 2 * (a.K >> 2); i += 256) {
          const int which = i >= (a.K >> 2), q = i - which * (a.K >> 2);
          const int64_t g = min(g0 + which, ng - 1);
          *reinterpret_cast<f32x4*>(rsS + which * RS_MAXK + 4 * q) =
              *reinterpret_cast<const f32x4*>(a.row_scale + g * a.K + 4 * q);
        }
        lds_barrier();
      }
    }
    commit_a(t, ks);
    if (!w_cached) commit_w(ks);
    lds_barrier();
    // issue the next stage's (or next tile's) global loads; they complete under the MFMAs below
    int tn = t;
    int ksn = ks + 1;
    if (ksn == nst) {
      ksn = 0;
      tn = next_valid(t + gridDim.x);
    }
    if (tn < total) {
      fetch_a(tn, ksn);
      if (!w_cached) fetch_w(tn, ksn);
    }
    int kpad, csh;
    stage_dims((CONV ? ks % nk : ks) * KS, kpad, csh);
    for (int kk = 0; kk < kpad; kk += MK) {
      if constexpr (sizeof(T) == 2) {
        const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(As + (wave * 32 + frow) * LDB + (kk + 8 * fk) * 2);
        const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(As + (wave * 32 + 16 + frow) * LDB + (kk + 8 * fk) * 2);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const bf16x8 bw = *reinterpret_cast<const bf16x8*>(Ws + (j * 16 + frow) * LDB + (kk + 8 * fk) * 2);
          acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bw, acc[0][j], 0, 0, 0);
          acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bw, acc[1][j], 0, 0, 0);
        }
      } else {
        // exact-f32 MFMA 16x16x4: lane (row, kgroup g) holds k = kk + 4g + s for step s
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(As + (wave * 32 + frow) * LDB + (kk + 4 * fk) * 4);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(As + (wave * 32 + 16 + frow) * LDB + (kk + 4 * fk) * 4);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const f32x4 bw = *reinterpret_cast<const f32x4*>(Ws + (j * 16 + frow) * LDB + (kk + 4 * fk) * 4);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], bw[s], acc[0][j], 0, 0, 0);
            acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], bw[s], acc[1][j], 0, 0, 0);
          }
        }
      }
    }
    lds_barrier();
    if (ks == nst - 1) {
      // ---- epilogue: affine/bias + activation, per-column statistics, LDS-transposed store
      const int64_t m0 = (int64_t)tile_mt(t) * BM;
      const int n0 = tile_nt(t) * BN;
      T* epi = (T*)smem + wave * 16 * EPI_LD;
      if (!plain) {
        if (n0 != osc_n0) {
        // per-column epilogue parameters, fetched once per column tile (never inside the element loops:
        // a load there costs a vmcnt(0) drain of the prefetch and of the previous stores)
        osc_n0 = n0;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int n = n0 + j * 16 + ccol;
          osc[j] = (a.out_scale && n < a.N) ? a.out_scale[n] : 1.f;
          osh[j] = (a.out_shift && n < a.N) ? a.out_shift[n] : 0.f;
        }
        }
        // affine + activation on the whole accumulator tile, activation switch OUTSIDE the element
        // loops (one uniform branch per tile instead of one ladder per element)
#define PB_EPI_LOOP(EXPR)                                             \
  _Pragma("unroll") for (int i = 0; i < 2; ++i)                       \
  _Pragma("unroll") for (int j = 0; j < NT; ++j)                      \
  _Pragma("unroll") for (int r = 0; r < 4; ++r) {                     \
    const float x = acc[i][j][r] * osc[j] + osh[j];                   \
    acc[i][j][r] = (EXPR);                                            \
  }
        switch (a.out_act) {
          case PB_ACT_RELU: PB_EPI_LOOP(x > 0.f ? x : 0.f) break;
          case PB_ACT_LEAKY_RELU: PB_EPI_LOOP(x > 0.f ? x : x * a.out_slope) break;
          case PB_ACT_HSWISH: PB_EPI_LOOP(x * fminf(fmaxf(x + 3.f, 0.f), 6.f) * (1.f / 6.f)) break;
          case PB_ACT_HSIGMOID: PB_EPI_LOOP(fminf(fmaxf(x * (1.f / 6.f) + 0.5f, 0.f), 1.f)) break;
          case PB_ACT_SILU: PB_EPI_LOOP(x / (1.f + __expf(-x))) break;
          default: PB_EPI_LOOP(x) break;
        }
#undef PB_EPI_LOOP
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int n = n0 + j * 16 + ccol;
          const bool nok = n < a.N;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int rl = rgrp * 4 + r;
            const int64_t m = m0 + wave * 32 + i * 16 + rl;
            const T tv = from_f32<T>(acc[i][j][r]);
            epi[rl * EPI_LD + j * 16 + ccol] = tv;
            if (a.stats && nok && m < a.M) {
              const float fv = to_f32(tv);
              csum[j] += fv;
              csq[j] += fv * fv;
            }
          }
          acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        lds_barrier();
        constexpr int CPR = BN / VEC;
        for (int f = lane; f < 16 * CPR; f += 64) {
          const int r = f / CPR, c = f - r * CPR;
          const int64_t m = m0 + wave * 32 + i * 16 + r;
          const int n = n0 + c * VEC;
          if (m < a.M && n < a.N) {
            if (ALIGNED || (yvec && n + VEC <= a.N)) {
              vec_store<T>(Y + m * a.N + n, vec_load<T>(epi + r * EPI_LD + c * VEC));
            } else if constexpr (!ALIGNED) {
              for (int e = 0; e < VEC; ++e)
                if (n + e < a.N) Y[m * a.N + n + e] = epi[r * EPI_LD + c * VEC + e];
            }
          }
        }
        lds_barrier();
      }
      // with several column tiles a workgroup's next tile may be another column range: flush now
      if (a.stats && a.n_tiles > 1) flush_stats(n0);
    }
    t = tn;
    ks = ksn;
  }
  if (a.stats && a.n_tiles == 1) flush_stats(0);
}

// ----------------------------------------------------------------------------- weight gradient
struct WgArgs {
  const void* dY;
  const void* X;
  float* dW;
  int64_t M;
  int K, N;
  const float* in_scale;
  const float* in_shift;
  int in_act;
  float in_slope;
  const float* row_scale;
  int64_t rows_per_group;
  int64_t rows_per_wg;
  int n_tiles, k_tiles;
  int ranges_per_group;   // > 0: row ranges are clip-aligned and the squeeze-excite scale is applied to the accumulators
  int split;           // pw_wgrad128: number of row ranges (grid.x is padded to 8 * tiles * ceil(split/8))
  // CONV: blockIdx.z = filter tap, X rows are the tap-shifted input voxels, dW is [N][taps][K]
  int cTi, cHi, cWi, cTo, cHo, cWo, ckt, ckh, ckw, cst, csh, csw, cpt, cph, cpw;
  float* P;            // pw_wgrad128_dma: per-range partial tiles [split][N][K] (plain stores) instead of atomics into dW
  const void* zero;    // CONV on the DMA kernel: 16 zero bytes in device memory (padding voxels)
};

// Each wave streams 32-row blocks of dY[:, n-tile] and X[:, k-tile] through its private LDS
// slab (row-major, exactly as loaded) and feeds MFMA with the reduction index m on the "k" axis:
// bf16 fragments come from ds_read_b64_tr_b16 (the hardware transposing read), f32 from plain
// 4-byte reads.  No workgroup barrier in the loop (waves are independent); the next block's global
// loads are issued before the current block's MFMAs.  At the end the four waves' tiles are summed
// in LDS (wave order) and stored as this row range's partial tile of the workspace a.P[range][N][K].
// DIAG (K = N = 16, four rows per 128-byte line: dY and X are read as [M/4][64]): only the four diagonal 16 x 16 blocks of
// the 64 x 64 product are computed; they are stored as four partial tiles [range][4][16][16] and the reduction adds them up.
template <typename T, int TNW, int TKW, bool CONV = false, bool DIAG = false>
__global__ __launch_bounds__(256) void pw_wgrad_kernel(WgArgs a) {
  constexpr int VEC = ElemTraits<T>::VEC;
  constexpr int TN = TNW * 16, TK = TKW * 16;
  constexpr int LDN = (TN + VEC) * (int)sizeof(T), LDK = (TK + VEC) * (int)sizeof(T);   // bytes
  constexpr int SLAB = 32 * (LDN + LDK);
  constexpr int CN = TN / VEC, CK = TK / VEC;
  constexpr int RN = 32 * CN / 64, RK = 32 * CK / 64;      // chunks per lane
  static_assert(4 * SLAB >= TN * TK * 4, "reduction tile must fit the staging slabs");
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * SLAB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nt = blockIdx.x % a.n_tiles, kt = blockIdx.x / a.n_tiles;
  const int n0 = nt * TN, k0 = kt * TK;
  const int64_t mbeg = (int64_t)blockIdx.y * a.rows_per_wg;
  const int64_t mend = min(a.M, mbeg + a.rows_per_wg);
  unsigned char* Ds = smem + wave * SLAB;       // dY slab [32][TN]
  unsigned char* Xs = Ds + 32 * LDN;            // X slab  [32][TK]
  const T* dY = (const T*)a.dY;
  const T* X = (const T*)a.X;
  const bool nvec = (a.N % VEC) == 0, kvec = (a.K % VEC) == 0;

  // reuse the forward prologue (same field names)
  PwArgs pa;
  pa.K = a.K; pa.in_scale = a.in_scale; pa.in_shift = a.in_shift; pa.in_act = a.in_act;
  pa.in_slope = a.in_slope; pa.row_scale = a.row_scale; pa.rows_per_group = a.rows_per_group;
  pa.M = a.M;

  f32x4 acc[TNW][TKW];
#pragma unroll
  for (int i = 0; i < TNW; ++i)
#pragma unroll
    for (int j = 0; j < TKW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  Vec16<T> rd[RN], rx[RK];
  RowScale<T> rsx[RK];
  const bool pro = a.in_scale || a.in_shift || a.in_act != PB_ACT_NONE || a.row_scale;
  auto fetch = [&](int64_t mb) {
    int64_t g0 = 0, g0_end = 0;
    if (a.row_scale) {
      g0 = mb / a.rows_per_group;
      g0_end = (g0 + 1) * a.rows_per_group;
    }
#pragma unroll
    for (int i = 0; i < RN; ++i) {
      const int f = lane + i * 64;
      const int r = f / CN, c = f - r * CN;
      const int64_t m = mb + r;
      const int n = n0 + c * VEC;
      rd[i] = (m < mend && n < a.N) ? load_row_chunk<T>(dY, m, n, a.N, nvec) : vec_zero<T>();
    }
#pragma unroll
    for (int i = 0; i < RK; ++i) {
      const int f = lane + i * 64;
      const int r = f / CK, c = f - r * CK;
      const int64_t m = mb + r;
      const int k = k0 + c * VEC;
      bool ok = m < mend && k < a.K;
      int64_t mx = m;
      if constexpr (CONV) {
        if (ok) {
          const int tap = blockIdx.z;
          const int dw_ = tap % a.ckw, dh = (tap / a.ckw) % a.ckh, dt = tap / (a.ckw * a.ckh);
          int64_t q = m;
          const int wo = (int)(q % a.cWo); q /= a.cWo;
          const int ho = (int)(q % a.cHo); q /= a.cHo;
          const int to = (int)(q % a.cTo);
          const int64_t n = q / a.cTo;
          const int ti = to * a.cst - a.cpt + dt, hi = ho * a.csh - a.cph + dh, wi = wo * a.csw - a.cpw + dw_;
          ok = ti >= 0 && ti < a.cTi && hi >= 0 && hi < a.cHi && wi >= 0 && wi < a.cWi;
          mx = ((n * a.cTi + ti) * a.cHi + hi) * a.cWi + wi;
        }
      }
      rx[i] = ok ? load_row_chunk<T>(X, mx, k, a.K, kvec) : vec_zero<T>();
      rsx[i].ones();
      if (pro && ok) rsx[i] = fetch_rowscale<T>(pa, row_group(m, mb, g0, g0_end, a.rows_per_group), k);
    }
  };

  int64_t mb = mbeg + wave * 32;
  if (mb < mend) fetch(mb);
  for (; mb < mend; mb += 128) {
#pragma unroll
    for (int i = 0; i < RN; ++i) {
      const int f = lane + i * 64;
      const int r = f / CN, c = f - r * CN;
      vec_store<T>((T*)(Ds + r * LDN) + c * VEC, rd[i]);
    }
#pragma unroll
    for (int i = 0; i < RK; ++i) {
      const int f = lane + i * 64;
      const int r = f / CK, c = f - r * CK;
      if (pro) rx[i] = apply_prologue<T>(rx[i], rsx[i], pa, k0 + c * VEC);
      vec_store<T>((T*)(Xs + r * LDK) + c * VEC, rx[i]);
    }
    if (mb + 128 < mend) fetch(mb + 128);         // in flight during the MFMAs below
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if constexpr (sizeof(T) == 2) {
      // 16-lane group g reads the 4x16 blocks at rows 8g..8g+3 and 8g+4..8g+7; lane 4q+p of the
      // group supplies the address of row q, columns 4p..4p+3; lane i receives column i.
      const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
      typedef __attribute__((address_space(3))) bf16x4* lds_p;
      bf16x8 fa[TNW], fb[TKW];
#pragma unroll
      for (int i = 0; i < TNW; ++i) {
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(Ds + (8 * g + q) * LDN + (i * 16 + 4 * p) * 2));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(Ds + (8 * g + 4 + q) * LDN + (i * 16 + 4 * p) * 2));
        fa[i] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int j = 0; j < TKW; ++j) {
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(Xs + (8 * g + q) * LDK + (j * 16 + 4 * p) * 2));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(Xs + (8 * g + 4 + q) * LDK + (j * 16 + 4 * p) * 2));
        fb[j] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int i = 0; i < TNW; ++i)
#pragma unroll
        for (int j = 0; j < TKW; ++j)
          if (!DIAG || i == j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    } else {
      const int col = lane & 15, g = lane >> 4;
#pragma unroll
      for (int ms = 0; ms < 8; ++ms) {
        float fa[TNW], fb[TKW];
#pragma unroll
        for (int i = 0; i < TNW; ++i) fa[i] = *reinterpret_cast<const float*>(Ds + (ms * 4 + g) * LDN + (i * 16 + col) * 4);
#pragma unroll
        for (int j = 0; j < TKW; ++j) fb[j] = *reinterpret_cast<const float*>(Xs + (ms * 4 + g) * LDK + (j * 16 + col) * 4);
#pragma unroll
        for (int i = 0; i < TNW; ++i)
#pragma unroll
          for (int j = 0; j < TKW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
  // cross-wave sum in LDS (C/D map: row (= n) = (lane>>4)*4 + r, col (= k) = lane & 15), then one
  // contiguous atomic per element and workgroup
  float* R = (float*)smem;    // [TN][TK]
  __syncthreads();
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int i = 0; i < TNW; ++i)
#pragma unroll
        for (int j = 0; j < TKW; ++j) {
          if (DIAG && i != j) continue;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int idx = (i * 16 + (lane >> 4) * 4 + r) * TK + j * 16 + (lane & 15);
            if (w == 0) R[idx] = acc[i][j][r];
            else R[idx] += acc[i][j][r];
          }
        }
    }
    __syncthreads();
  }
  if constexpr (DIAG) {
    static_assert(!DIAG || (TN == 64 && TK == 64), "diagonal form: one 64 x 64 tile");
    for (int e = tid; e < 4 * 256; e += 256) {
      const int b = e >> 8, nn = (e >> 4) & 15, kk = e & 15;
      a.P[(int64_t)blockIdx.y * 1024 + e] = R[(b * 16 + nn) * TK + b * 16 + kk];
    }
    return;
  }
  const int64_t ldw = CONV ? (int64_t)a.ckt * a.ckh * a.ckw * a.K : a.K;
  const int64_t woff = CONV ? (int64_t)blockIdx.z * a.K : 0;
  // this row range's partial tile (plain stores; the ranges are added up in a fixed order afterwards)
  for (int e = tid; e < TN * TK; e += 256) {
    const int n = n0 + e / TK, k = k0 + e % TK;
    if (n < a.N && k < a.K) a.P[((int64_t)blockIdx.y * a.N + n) * ldw + woff + k] = R[e];
  }
}

// 128 x 128 output tile per workgroup (bf16): the four waves own the four 64 x 64 quadrants and share
// every staged 32-row block of dY[:, 128] and X[:, 128] (double-buffered LDS, one barrier per block), so
// each operand is re-read N/128 resp. K/128 times instead of N/64, K/64 - the 64 x 64 wave-private
// version measured 2.3x its algorithmic bytes in HBM traffic (profiles/r01_pmc_traffic.json).
template <bool CONV>
__global__ __launch_bounds__(256) void pw_wgrad128_kernel(WgArgs a) {
  typedef bf16_t T;
  constexpr int VEC = 8, TT = 128;
  constexpr int LD = (TT + VEC) * 2;                 // bytes per LDS row (272)
  constexpr int SLAB = 32 * LD;                      // one operand, one buffer
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * SLAB];    // [buf][dY | X]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave & 1, wk = wave >> 1;
  // XCD-aware order: workgroup ids are dealt round-robin to the 8 XCDs, so the tiles that re-read the
  // same row range get ids 8 apart - same XCD (same L2), dispatched back to back.
  const int tiles = a.n_tiles * a.k_tiles;
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int tile = jj % tiles, srange = (jj / tiles) * 8 + xcd;
  if (srange >= a.split) return;
  const int nt = tile % a.n_tiles, kt = tile / a.n_tiles;
  const int n0 = nt * TT, k0 = kt * TT;
  // Squeeze-excite input scale (X[m][k] * s[clip(m)][k]): when every row range lies inside one clip the
  // factor leaves the reduction - dW[n][k] += s[clip][k] * sum_m dY[m][n] X[m][k] - and is applied once
  // to the accumulators; nothing rides on the streaming loop.
  int64_t mbeg, mend;
  int clip = 0;
  if (a.ranges_per_group > 0) {
    clip = srange / a.ranges_per_group;
    const int sub = srange - clip * a.ranges_per_group;
    mbeg = (int64_t)clip * a.rows_per_group + (int64_t)sub * a.rows_per_wg;
    mend = min((int64_t)(clip + 1) * a.rows_per_group, mbeg + a.rows_per_wg);
  } else {
    mbeg = (int64_t)srange * a.rows_per_wg;
    mend = min(a.M, mbeg + a.rows_per_wg);
  }
  const T* dY = (const T*)a.dY;
  const T* X = (const T*)a.X;
  PwArgs pa;
  pa.K = a.K; pa.in_scale = a.in_scale; pa.in_shift = a.in_shift; pa.in_act = a.in_act;
  pa.in_slope = a.in_slope; pa.row_scale = a.row_scale; pa.rows_per_group = a.rows_per_group; pa.M = a.M;
  const bool pro = a.ranges_per_group == 0 && (a.in_scale || a.in_shift || a.in_act != PB_ACT_NONE || a.row_scale);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // 32 rows x 16 chunks per operand = 512 chunks: 2 per thread
  Vec16<T> rd[2], rx[2];
  RowScale<T> rsx[2];
  auto fetch = [&](int64_t mb) {
    int64_t g0 = 0, g0_end = 0;
    if (a.row_scale) {
      g0 = mb / a.rows_per_group;
      g0_end = (g0 + 1) * a.rows_per_group;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int f = tid + i * 256;
      const int r = f >> 4, c = f & 15;
      const int64_t m = mb + r;
      const int n = n0 + c * VEC, k = k0 + c * VEC;
      rd[i] = (m < mend && n < a.N) ? vec_load<T>(dY + m * a.N + n) : vec_zero<T>();
      bool ok = m < mend && k < a.K;
      int64_t mx = m;
      if constexpr (CONV) {
        if (ok) {
          const int tap = blockIdx.z;
          const int dw_ = tap % a.ckw, dh = (tap / a.ckw) % a.ckh, dt = tap / (a.ckw * a.ckh);
          int64_t q = m;
          const int wo = (int)(q % a.cWo); q /= a.cWo;
          const int ho = (int)(q % a.cHo); q /= a.cHo;
          const int to = (int)(q % a.cTo);
          const int64_t nn = q / a.cTo;
          const int ti = to * a.cst - a.cpt + dt, hi = ho * a.csh - a.cph + dh, wi = wo * a.csw - a.cpw + dw_;
          ok = ti >= 0 && ti < a.cTi && hi >= 0 && hi < a.cHi && wi >= 0 && wi < a.cWi;
          mx = ((nn * a.cTi + ti) * a.cHi + hi) * a.cWi + wi;
        }
      }
      rx[i] = ok ? vec_load<T>(X + mx * a.K + k) : vec_zero<T>();
      rsx[i].ones();
      if (pro && ok) rsx[i] = fetch_rowscale<T>(pa, row_group(m, mb, g0, g0_end, a.rows_per_group), k);
    }
  };
  int buf = 0;
  if (mbeg < mend) fetch(mbeg);
  for (int64_t mb = mbeg; mb < mend; mb += 32) {
    unsigned char* Ds = smem + buf * 2 * SLAB;
    unsigned char* Xs = Ds + SLAB;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int f = tid + i * 256;
      const int r = f >> 4, c = f & 15;
      vec_store<T>((T*)(Ds + r * LD) + c * VEC, rd[i]);
      if (pro) rx[i] = apply_prologue<T>(rx[i], rsx[i], pa, k0 + c * VEC);
      vec_store<T>((T*)(Xs + r * LD) + c * VEC, rx[i]);
    }
    lds_barrier();
    if (mb + 32 < mend) fetch(mb + 32);              // in flight during the MFMAs
    {
      const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
      typedef __attribute__((address_space(3))) bf16x4* lds_p;
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(Ds + (8 * g + q) * LD + (wn * 64 + i * 16 + 4 * p) * 2));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(Ds + (8 * g + 4 + q) * LD + (wn * 64 + i * 16 + 4 * p) * 2));
        fa[i] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(Xs + (8 * g + q) * LD + (wk * 64 + j * 16 + 4 * p) * 2));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(Xs + (8 * g + 4 + q) * LD + (wk * 64 + j * 16 + 4 * p) * 2));
        fb[j] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    buf ^= 1;
  }
  // quadrants are disjoint: each wave stores its own 64 x 64 block of the range's partial tile (row = n = (lane>>4)*4 + r,
  // col = k = lane & 15)
  const int64_t ldw = CONV ? (int64_t)a.ckt * a.ckh * a.ckw * a.K : a.K;
  const int64_t woff = CONV ? (int64_t)blockIdx.z * a.K : 0;
  float sk[4] = {1.f, 1.f, 1.f, 1.f};
  if (a.ranges_per_group > 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + wk * 64 + j * 16 + (lane & 15);
      if (k < a.K) sk[j] = a.row_scale[(int64_t)clip * a.K + k];
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * 64 + i * 16 + (lane >> 4) * 4 + r;
        const int k = k0 + wk * 64 + j * 16 + (lane & 15);
        if (n < a.N && k < a.K) a.P[((int64_t)srange * a.N + n) * ldw + woff + k] = acc[i][j][r] * sk[j];
      }
}

// Skinny GEMM for M <= 64 rows (the classifier head on pooled features: 32 x 960 x 1280, 32 x 1280 x 13 and
// their backward-data): the tiled kernel would run it on 1-10 workgroups.  A workgroup owns 4 output
// columns and a block of 32 rows; its 4 waves split K, lanes stride over k, every lane keeps the 32 x 4
// partial sums of its k subset in registers (weights read exactly once from HBM, X rows from L2, 32
// independent loads in flight per step); one butterfly reduction per output at the end.  No MFMA: 40 MFLOP.
template <typename T>
__global__ __launch_bounds__(256) void pw_skinny_kernel(PwArgs a) {
  constexpr int RB = 32;
  __shared__ float part[4][RB][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * 4;
  const int m0 = blockIdx.y * RB;
  const int rows = min(RB, (int)a.M - m0);
  const T* X = (const T*)a.X + (int64_t)m0 * a.K;
  T* Y = (T*)a.Y;
  const int kq = ((a.K + 255) >> 8) << 6;            // K range of one wave, a multiple of 64
  const int kbeg = wave * kq, kend = min(a.K, kbeg + kq);
  float d[RB][4];
#pragma unroll
  for (int m = 0; m < RB; ++m)
#pragma unroll
    for (int c = 0; c < 4; ++c) d[m][c] = 0.f;
  const bool wvec = a.w_transposed && n0 + 3 < a.N && (a.ldw & 3) == 0;
  for (int k = kbeg + lane; k < kend; k += 64) {
    float w4[4];
    if (wvec) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(a.W + (int64_t)k * a.ldw + n0);
      w4[0] = t[0]; w4[1] = t[1]; w4[2] = t[2]; w4[3] = t[3];
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        w4[c] = n0 + c < a.N ? (a.w_transposed ? a.W[(int64_t)k * a.ldw + n0 + c] : a.W[(int64_t)(n0 + c) * a.ldw + k]) : 0.f;
    }
    float xv[RB];
#pragma unroll
    for (int m = 0; m < RB; ++m) xv[m] = m < rows ? to_f32(X[(int64_t)m * a.K + k]) : 0.f;
#pragma unroll
    for (int m = 0; m < RB; ++m)
#pragma unroll
      for (int c = 0; c < 4; ++c) d[m][c] += xv[m] * w4[c];
  }
  // lane m ends up with row m's four sums of this wave's K range
  float mine[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int m = 0; m < RB; ++m)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float v = wave_sum(d[m][c]);
      if (lane == m) mine[c] = v;
    }
  if (lane < RB) {
#pragma unroll
    for (int c = 0; c < 4; ++c) part[wave][lane][c] = mine[c];
  }
  __syncthreads();
  if (wave == 0 && lane < rows) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (n0 + c < a.N) {
        float v = (part[0][lane][c] + part[1][lane][c]) + (part[2][lane][c] + part[3][lane][c]);
        const float sc = a.out_scale ? a.out_scale[n0 + c] : 1.f;
        const float sh = a.out_shift ? a.out_shift[n0 + c] : 0.f;
        v = act_fwd(v * sc + sh, a.out_act, a.out_slope);
        Y[(int64_t)(m0 + lane) * a.N + n0 + c] = from_f32<T>(v);
      }
    }
  }
}

// The same for bf16 on the matrix cores: a workgroup owns 16 output columns, its four waves split K in 32-wide chunks
// (two 16x16x32 MFMAs per chunk for the 32 rows, twice: the fp32 weights enter as a bf16 hi + lo pair so that the head
// keeps the accuracy of fp32 weights), partial tiles meet in LDS.  The scalar kernel above spends its time
// in 32 two-byte loads per k and 768 cross-lane shuffles per wave (50-75 us for the 32 x 960 x 1280 head layers).
__global__ __launch_bounds__(256) void pw_skinny_mfma_kernel(PwArgs a) {
  __shared__ float part[4][32][17];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, kg = lane >> 4;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 32;
  const int rows = min(32, (int)a.M - m0);
  const bf16_t* X = (const bf16_t*)a.X + (int64_t)m0 * a.K;
  const int n = n0 + r16;
  const bool nok = n < a.N;
  const bool xvec = (a.K & 7) == 0 && a.K >= 8;
  const bool wvec = !a.w_transposed && (a.ldw & 3) == 0 && (a.K & 7) == 0 && a.K >= 8;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  // One K chunk of a wave = 32 columns.  Every operand is fetched from a CLAMPED (always valid) address and zeroed by a select:
  // no branch sits between the loads, so the compiler issues the loads of two chunks back to back (the loop is unrolled by 2)
  // instead of one round trip per chunk - the 32 x 1280 x 960 backward-data layer of the head (weights read transposed, eight
  // scalar loads per chunk) took 32 us for 4.9 MB that way.
  auto chunk = [&](int k0) {
    const int k = k0 + kg * 8;
    bf16x8 a0, a1, b, bl;                              // W = b + bl: the fp32 weights keep ~16 mantissa bits
    const bool kin = k < a.K, kfull = k + 8 <= a.K;
    if (xvec) {
      const int kc = min(k, a.K - 8);
      const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(X + (int64_t)min(r16, rows - 1) * a.K + kc);
      const bf16x8 v1 = *reinterpret_cast<const bf16x8*>(X + (int64_t)min(r16 + 16, rows - 1) * a.K + kc);
      a0 = (kin && r16 < rows) ? v0 : z;               // (K % 8 == 0: a chunk that starts inside K is whole)
      a1 = (kin && r16 + 16 < rows) ? v1 : z;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int kk = min(k + j, a.K - 1);
        const bf16_t x0 = X[(int64_t)min(r16, rows - 1) * a.K + kk], x1 = X[(int64_t)min(r16 + 16, rows - 1) * a.K + kk];
        a0[j] = (k + j < a.K && r16 < rows) ? x0 : (bf16_t)0.f;
        a1[j] = (k + j < a.K && r16 + 16 < rows) ? x1 : (bf16_t)0.f;
      }
    }
    const int nc = min(n, a.N - 1);
    float wv[8];
    if (wvec) {
      const int kc = min(k, a.K - 8);                  // ((ldw & 3) == 0 and K >= 8 are part of wvec)
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(a.W + (int64_t)nc * a.ldw + kc);
      const f32x4 w1 = *reinterpret_cast<const f32x4*>(a.W + (int64_t)nc * a.ldw + kc + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { wv[j] = w0[j]; wv[4 + j] = w1[j]; }
#pragma unroll
      for (int j = 0; j < 8; ++j) wv[j] = (nok && kfull) ? wv[j] : 0.f;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int kk = min(k + j, a.K - 1);
        const float w = a.w_transposed ? a.W[(int64_t)kk * a.ldw + nc] : a.W[(int64_t)nc * a.ldw + kk];
        wv[j] = (nok && k + j < a.K) ? w : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      b[j] = (bf16_t)wv[j];
      bl[j] = (bf16_t)(wv[j] - (float)b[j]);
    }
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bl, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bl, acc1, 0, 0, 0);
  };
  int k0 = wave * 32;
  for (; k0 + 128 < a.K; k0 += 256) {
    chunk(k0);
    chunk(k0 + 128);
  }
  if (k0 < a.K) chunk(k0);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    part[wave][kg * 4 + r][r16] = acc0[r];
    part[wave][16 + kg * 4 + r][r16] = acc1[r];
  }
  __syncthreads();
  bf16_t* Y = (bf16_t*)a.Y;
  for (int i = threadIdx.x; i < 32 * 16; i += 256) {
    const int m = i >> 4, c = i & 15;
    if (m < rows && n0 + c < a.N) {
      float v = (part[0][m][c] + part[1][m][c]) + (part[2][m][c] + part[3][m][c]);
      const float sc = a.out_scale ? a.out_scale[n0 + c] : 1.f;
      const float sh = a.out_shift ? a.out_shift[n0 + c] : 0.f;
      v = act_fwd(v * sc + sh, a.out_act, a.out_slope);
      Y[(int64_t)(m0 + m) * a.N + n0 + c] = from_f32<bf16_t>(v);
    }
  }
}

// LDS-DMA variant of the 128 x 128 weight-gradient tile (plain streams: no affine prologue, no conv):
// the 32-row blocks of dY[:, 128] and X[:, 128] go HBM -> LDS by global_load_lds_dwordx4 into a ring of
// NS stage buffers, NS-1 blocks ahead of the MFMAs, with no registers and no commit pass in between.
// The register-staged kernel above keeps at most one block in flight per workgroup and only while its
// MFMAs run - measured 1.4 TB/s on cold operands; the ring keeps (NS-1) x 16 KB in flight per workgroup
// all the time.  LDS rows are 256 B, unpadded (a DMA instruction writes 1 KiB = 4 rows linearly); bank
// conflicts of the transposing reads are avoided by an XOR swizzle of the 16-byte chunk index applied to
// the *source* address: LDS chunk c of row r holds global chunk c ^ (swz(r) << 1).
// One LDS-DMA instruction: 16 bytes per lane from each lane's own global address to LDS at
// (wave-uniform) lds_dst + lane * 16 (layout confirmed on MI355X by tools/probes/glds_probe.hip).
// Inline asm on purpose: hipcc tracks __builtin_amdgcn_global_load_lds as an LDS write and puts
// s_waitcnt vmcnt(0) in front of the next ds_read, which drains the whole ring every step (measured:
// +25 % on the weight-gradient kernel).  Hidden from the compiler, completion is ours to order: a counted
// vmcnt in the issuing wave, then a barrier, before any ds_read of the destination.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned char* lds_dst) {
  typedef __attribute__((address_space(3))) unsigned char* lds_b;
  const unsigned dst = (unsigned)(uintptr_t)(lds_b)lds_dst;
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
__device__ __forceinline__ int wg_swz(int r) { return ((r & 3) | (((r >> 3) & 1) << 2)) << 1; }

template <int NS, bool CONV = false>
__global__ __launch_bounds__(256) void pw_wgrad128_dma_kernel(WgArgs a) {
  typedef bf16_t T;
  constexpr int TT = 128;
  constexpr int SLAB = 32 * TT * 2;                  // 8 KB: one operand, one stage
  extern __shared__ __attribute__((aligned(1024))) unsigned char dsmem[];     // [NS][dY | X]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave & 1, wk = wave >> 1;
  const int tiles = a.n_tiles * a.k_tiles;
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int tile = jj % tiles, srange = (jj / tiles) * 8 + xcd;
  if (srange >= a.split) return;
  const int nt = tile % a.n_tiles, kt = tile / a.n_tiles;
  const int n0 = nt * TT, k0 = kt * TT;
  int64_t mbeg, mend;
  int clip = 0;
  if (a.ranges_per_group > 0) {
    clip = srange / a.ranges_per_group;
    const int sub = srange - clip * a.ranges_per_group;
    mbeg = (int64_t)clip * a.rows_per_group + (int64_t)sub * a.rows_per_wg;
    mend = min((int64_t)(clip + 1) * a.rows_per_group, mbeg + a.rows_per_wg);
  } else {
    mbeg = (int64_t)srange * a.rows_per_wg;
    mend = min(a.M, mbeg + a.rows_per_wg);
  }
  if (mbeg >= mend) return;                          // uniform
  const T* dY = (const T*)a.dY;
  const T* X = (const T*)a.X;

  // this thread's two chunks per operand and stage: f = i*256 + tid -> row f>>4, LDS chunk f&15
  // Chunks past the edge of dY / X are not loaded at all: their LDS bytes stay stale and only feed
  // outputs that are never stored (every DMA instruction keeps lanes with a valid chunk, so the per-wave
  // instruction count that the vmcnt waits rely on does not change).
  int rr[2], ncol[2], kcol[2];
  bool nok[2], kok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int f = i * 256 + tid;
    rr[i] = f >> 4;
    const int cs = (f & 15) ^ wg_swz(rr[i]);
    ncol[i] = n0 + cs * 8;
    kcol[i] = k0 + cs * 8;
    nok[i] = ncol[i] < a.N;
    kok[i] = kcol[i] < a.K;
  }
  const int wave_off = __builtin_amdgcn_readfirstlane(wave * 1024);
  // CONV (blockIdx.z = filter tap): the X row of output voxel m is the tap-shifted input voxel, or the zero page
  const int tap = CONV ? (int)blockIdx.z : 0;
  const int tap_w = CONV ? tap % a.ckw : 0, tap_h = CONV ? (tap / a.ckw) % a.ckh : 0, tap_t = CONV ? tap / (a.ckw * a.ckh) : 0;
  auto issue = [&](int64_t mb, int buf) {
    unsigned char* base = dsmem + buf * (2 * SLAB) + wave_off;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int64_t m = min(mb + rr[i], mend - 1);           // rows past the range re-read the last row; zeroed below
      if (nok[i]) glds16(dY + m * a.N + ncol[i], base + i * 4096);
      if constexpr (CONV) {
        if (kok[i]) {
          const unsigned mu = (unsigned)m;
          const unsigned wo = mu % (unsigned)a.cWo, r1 = mu / (unsigned)a.cWo;
          const unsigned ho = r1 % (unsigned)a.cHo, r2 = r1 / (unsigned)a.cHo;
          const unsigned to = r2 % (unsigned)a.cTo, n = r2 / (unsigned)a.cTo;
          const int ti = (int)to * a.cst - a.cpt + tap_t, hi = (int)ho * a.csh - a.cph + tap_h, wi = (int)wo * a.csw - a.cpw + tap_w;
          const T* src = (const T*)a.zero;
          if ((unsigned)ti < (unsigned)a.cTi && (unsigned)hi < (unsigned)a.cHi && (unsigned)wi < (unsigned)a.cWi)
            src = X + ((((int64_t)n * a.cTi + ti) * a.cHi + hi) * a.cWi + wi) * a.K + kcol[i];
          glds16(src, base + SLAB + i * 4096);
        }
      } else {
        if (kok[i]) glds16(X + m * a.K + kcol[i], base + SLAB + i * 4096);
      }
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nsteps = (int)((mend - mbeg + 31) >> 5);
#pragma unroll
  for (int s0 = 0; s0 < NS - 1; ++s0)
    if (s0 < nsteps) issue(mbeg + 32 * s0, s0);

  // transposing-read lane constants (see pw_wgrad128_kernel for the fragment layout)
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const int sw = wg_swz(8 * g + q);                  // = swizzle of rows 8g+q and 8g+4+q alike
  const int row_off = (8 * g + q) * 256 + (pp & 1) * 8;
  int a_off[4], b_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a_off[i] = row_off + (((wn * 8 + i * 2 + (pp >> 1)) ^ sw) << 4);
    b_off[i] = row_off + (((wk * 8 + i * 2 + (pp >> 1)) ^ sw) << 4) + SLAB;
  }

  int buf = 0;
  for (int s = 0; s < nsteps; ++s) {
    // retire this wave's DMAs of stage s; later stages stay in flight
    const int later = min(nsteps - 1 - s, NS - 2);
    if (later >= 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (later == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (later == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int64_t mb = mbeg + 32 * (int64_t)s;
    unsigned char* Ds = dsmem + buf * (2 * SLAB);
    if (mb + 32 > mend) {                            // ragged last block: rows past the range contribute zero
#pragma unroll
      for (int i = 0; i < 2; ++i)
        if (mb + rr[i] >= mend) {
          *reinterpret_cast<u32x4*>(Ds + (i * 256 + tid) * 16) = u32x4{0u, 0u, 0u, 0u};
          *reinterpret_cast<u32x4*>(Ds + SLAB + (i * 256 + tid) * 16) = u32x4{0u, 0u, 0u, 0u};
        }
    }
    lds_barrier();       // all waves' stage-s data has landed; all waves are done reading stage s-1
    if (s + NS - 1 < nsteps) {
      int nb = buf + NS - 1;
      if (nb >= NS) nb -= NS;
      issue(mb + 32 * (NS - 1), nb);
    }
    {
      typedef __attribute__((address_space(3))) bf16x4* lds_p;
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(Ds + a_off[i]));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(Ds + a_off[i] + 4 * 256));
        fa[i] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(Ds + b_off[j]));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_p)(Ds + b_off[j] + 4 * 256));
        fb[j] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (++buf == NS) buf = 0;
  }
  float sk[4] = {1.f, 1.f, 1.f, 1.f};
  if (a.ranges_per_group > 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + wk * 64 + j * 16 + (lane & 15);
      if (k < a.K) sk[j] = a.row_scale[(int64_t)clip * a.K + k];
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * 64 + i * 16 + (lane >> 4) * 4 + r;
        const int k = k0 + wk * 64 + j * 16 + (lane & 15);
        if (n < a.N && k < a.K) {
          // every range stores its partial tile (64-byte runs, plain stores) and the ranges are added up afterwards in a
          // fixed order: no atomics (a CU retires float atomics at ~1 per clock - 16 K of them per workgroup cost
          // 25-30 us whatever the layer, measured 39 -> 11 us without them at M = 8192 - and their order is not
          // reproducible)
          const int64_t ldw = CONV ? (int64_t)a.ckt * a.ckh * a.ckw * a.K : a.K;      // dW is [N][taps][K] for CONV
          const int64_t col = (CONV ? (int64_t)tap * a.K : 0) + k;
          a.P[((int64_t)srange * a.N + n) * ldw + col] = acc[i][j][r] * sk[j];
        }
      }
}

int pick_nt(int N, bool f32) {
  // cost = padded MFMA width + 16 per column tile (each extra tile re-reads the X panel through L2)
  const int cand[5] = {2, 4, 6, 8, 12};
  int best = 2;
  int64_t best_cost = INT64_MAX;
  for (int c : cand) {
    if (f32 && c > 8) continue;          // f32 epilogue staging of a 192-wide tile exceeds 64 KB LDS
    const int bn = c * 16;
    const int64_t tiles = pb_cdiv(N, bn);
    const int64_t cost = tiles * bn + 16 * tiles;
    if (cost < best_cost) { best = c; best_cost = cost; }
  }
  return best;
}

// Persistent grid = the workgroups that are actually co-resident (registers AND LDS): a grid
// sized from LDS alone left a second, nearly empty round of workgroups (2x the time).
template <typename T, int NT, bool TRANS, bool ALIGNED, bool PRO>
void launch_one(const PwArgs& a, int64_t total, int device, hipStream_t st) {
  // Persistent grid = the workgroups that are actually co-resident (registers AND LDS): a grid
  // sized from LDS alone left a second, nearly empty round of workgroups (2x the time).
  static int cached[16] = {0};
  const int d = device & 15;
  if (cached[d] == 0) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pw_gemm_kernel<T, NT, TRANS, ALIGNED, PRO>, 256, 0) != hipSuccess || per_cu < 1)
      per_cu = 1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1) cus = 256;
    cached[d] = per_cu * cus;
  }
  int64_t blocks = cached[d];
  if (blocks > total) blocks = total;
  blocks = pb_cdiv(blocks, 8) * 8;
  if (a.stats && blocks > PB_STAT_SLOTS) blocks = PB_STAT_SLOTS;   // one statistics slot per workgroup
  hipLaunchKernelGGL((pw_gemm_kernel<T, NT, TRANS, ALIGNED, PRO>), dim3((unsigned)blocks), dim3(256), 0, st, a);
}

template <typename T, int NT>
void launch_nt(const PwArgs& a, int64_t total, bool pro, int device, hipStream_t st) {
  if (a.w_transposed) {
    if (pro) launch_one<T, NT, true, true, true>(a, total, device, st);
    else launch_one<T, NT, true, true, false>(a, total, device, st);
  } else {
    if (pro) launch_one<T, NT, false, true, true>(a, total, device, st);
    else launch_one<T, NT, false, true, false>(a, total, device, st);
  }
}

template <typename T>
int launch_gemm(const PwArgs& a0, int device, hipStream_t st) {
  PwArgs a = a0;
  constexpr int VEC = ElemTraits<T>::VEC;
  const bool aligned = (a.K % VEC) == 0 && (a.N % VEC) == 0 && (a.ldw % 4) == 0;
  const bool pro = a.in_scale || a.in_shift || a.in_act != PB_ACT_NONE || a.row_scale;
  // the generic (unaligned) path exists for the tiny classifier-sized problems: one tile width only
  // (fp32: 4 column tiles - the 8-tile unaligned fp32 instantiation needs 328 registers and trips the AGPR-save-ahead-of-
  // the-exec-restore code generation that tools/lint_isa.py looks for)
  constexpr int UNT = sizeof(T) == 4 ? 4 : 8;
  const int nt = aligned ? pick_nt(a.N, sizeof(T) == 4) : UNT;
  a.n_tiles = (int)pb_cdiv(a.N, nt * 16);
  a.m_tiles = pb_cdiv(a.M, BM);
  const int64_t total = a.n_tiles == 1 ? a.m_tiles : pb_cdiv(a.m_tiles, 8) * 8 * a.n_tiles;
  if (total > 0x3fffffffLL) { pb_set_error("pb_pwconv_fwd: too many tiles"); return 1; }
  if (!aligned) {
    if (a.w_transposed) launch_one<T, UNT, true, false, true>(a, total, device, st);
    else launch_one<T, UNT, false, false, true>(a, total, device, st);
    return 0;
  }
  switch (nt) {
    case 2: launch_nt<T, 2>(a, total, pro, device, st); break;
    case 4: launch_nt<T, 4>(a, total, pro, device, st); break;
    case 6: launch_nt<T, 6>(a, total, pro, device, st); break;
    case 8: launch_nt<T, 8>(a, total, pro, device, st); break;
    default:
      if constexpr (sizeof(T) == 2) launch_nt<T, 12>(a, total, pro, device, st);
      else launch_nt<T, 8>(a, total, pro, device, st);
      break;
  }
  return 0;
}

}  // namespace

extern "C" int pb_pwconv_fwd(const void* x, const float* w, void* y, int64_t M, int K, int N, int ldw,
                             int w_transposed, pb_dtype dtype, const float* in_scale, const float* in_shift,
                             pb_act in_act, float in_slope, const float* row_scale, int64_t rows_per_group,
                             const float* out_scale, const float* out_shift, pb_act out_act, float out_slope,
                             float* stats, int device, void* stream) {
  PB_CHECK_ARG(x && w && y, "pb_pwconv_fwd: null pointer");
  PB_CHECK_ARG(M >= 0 && K > 0 && N > 0, "pb_pwconv_fwd: bad sizes M=%lld K=%d N=%d", (long long)M, K, N);
  PB_CHECK_ARG(ldw >= (w_transposed ? N : K), "pb_pwconv_fwd: ldw=%d too small", ldw);
  PB_CHECK_ARG(dtype == PB_F32 || dtype == PB_BF16, "pb_pwconv_fwd: bad dtype %d", (int)dtype);
  PB_CHECK_ARG(!row_scale || rows_per_group > 0, "pb_pwconv_fwd: row_scale needs rows_per_group > 0");
  PB_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)w % 16) == 0,
               "pb_pwconv_fwd: pointers must be 16-byte aligned");
  if (M == 0) return 0;
  PB_ENTER(device);
  PwArgs a;
  a.X = x; a.W = w; a.Y = y; a.M = M; a.K = K; a.N = N; a.ldw = ldw; a.w_transposed = w_transposed;
  a.in_scale = in_scale; a.in_shift = in_shift; a.in_act = (int)in_act; a.in_slope = in_slope;
  a.row_scale = row_scale; a.rows_per_group = rows_per_group > 0 ? rows_per_group : 1;
  a.out_scale = out_scale; a.out_shift = out_shift; a.out_act = (int)out_act; a.out_slope = out_slope;
  a.stats = stats; a.n_tiles = 0; a.m_tiles = 0;
  static const bool skinny_on = !(getenv("PB_SKINNY") && atoi(getenv("PB_SKINNY")) == 0);
  if (skinny_on && M <= 64 && !in_scale && !in_shift && in_act == PB_ACT_NONE && !row_scale && !stats) {
    const dim3 grid((unsigned)pb_cdiv(N, 4), (unsigned)pb_cdiv(M, 32));
    static const bool skinny_mfma = !(getenv("PB_SKINNY_MFMA") && atoi(getenv("PB_SKINNY_MFMA")) == 0);
    if (dtype == PB_BF16 && skinny_mfma)
      hipLaunchKernelGGL(pw_skinny_mfma_kernel, dim3((unsigned)pb_cdiv(N, 16), grid.y), dim3(256), 0, (hipStream_t)stream, a);
    else if (dtype == PB_BF16) hipLaunchKernelGGL(pw_skinny_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(pw_skinny_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, a);
    PB_LAUNCH_CHECK();
    return 0;
  }
  int rc = dtype == PB_BF16 ? launch_gemm<bf16_t>(a, device, (hipStream_t)stream) : launch_gemm<float>(a, device, (hipStream_t)stream);
  if (rc) return rc;
  PB_LAUNCH_CHECK();
  return 0;
}

namespace {

// launch plan of the 128x128-tile weight-gradient kernels (shared by the launcher and the workspace-size query)
struct WgPlan {
  bool dma;
  int ns, n_tiles, k_tiles, ranges_per_group;
  int64_t split, rows_per_wg;
};

WgPlan wgrad128_plan(int64_t M, int K, int N, bool has_row_scale, bool affine, int64_t rows_per_group, bool with_ws, int device) {
  WgPlan p;
  p.n_tiles = (int)pb_cdiv(N, 128);
  p.k_tiles = (int)pb_cdiv(K, 128);
  const int64_t tiles128 = (int64_t)p.n_tiles * p.k_tiles;
  // One round of co-resident workgroups (a second, nearly empty round would double the time).
  // Atomic epilogue: every workgroup is one atomic writer per element of its dW tile and same-address float atomics
  // serialise (~25 ns each), so layers with few tiles get fewer, deeper workgroups: 1 per CU with a 6-stage ring
  // (80 KB in flight) instead of 3 per CU with 3 stages.  Workspace epilogue: every workgroup writes and the
  // reduction re-reads a 64 KB partial tile, so the grid is sized by PB_WGRAD_WS_WGS workgroups per CU.
  static int cus = 0;
  if (cus == 0 && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) cus = 256;
  static const int force_ns = getenv("PB_WGRAD_NS") ? atoi(getenv("PB_WGRAD_NS")) : 0;
  static const int small_tiles = getenv("PB_WGRAD_SMALL_TILES") ? atoi(getenv("PB_WGRAD_SMALL_TILES")) : 4;
  static const int small_ns = getenv("PB_WGRAD_SMALL_NS") ? atoi(getenv("PB_WGRAD_SMALL_NS")) : 6;
  static const int ws_ns = getenv("PB_WGRAD_WS_NS") ? atoi(getenv("PB_WGRAD_WS_NS")) : 0;
  int ns = force_ns ? force_ns : (tiles128 <= small_tiles ? small_ns : 3);
  if (with_ws && ws_ns) ns = ws_ns;
  if (ns != 3 && ns != 4 && ns != 6) ns = 3;
  p.ns = ns;
  const int per_cu = ns == 3 ? 3 : (ns == 4 ? 2 : 1);        // 48 / 64 / 96 KB of LDS per workgroup
  const int64_t target_wgs = (int64_t)per_cu * cus;
  int64_t split = target_wgs / tiles128;
  static const int ws_min_rows = getenv("PB_WGRAD_WS_MINROWS") ? atoi(getenv("PB_WGRAD_WS_MINROWS")) : 512;
  const int64_t max_split128 = pb_cdiv(M, with_ws ? ws_min_rows : 256);
  if (split > max_split128) split = max_split128;
  if (split < 1) split = 1;
  p.rows_per_wg = pb_cdiv(pb_cdiv(M, split), 32) * 32;
  split = pb_cdiv(M, p.rows_per_wg);
  p.ranges_per_group = 0;
  const bool se_only = has_row_scale && !affine;
  if (se_only && rows_per_group >= 64 && M % rows_per_group == 0) {
    // clip-aligned ranges (the squeeze-excite scale then applies to the accumulators): at most the
    // same number of workgroups, none straddling two clips
    const int64_t clips = M / rows_per_group;
    int64_t rpc = split / clips;
    if (rpc < 1) rpc = 1;
    p.rows_per_wg = pb_cdiv(pb_cdiv(rows_per_group, rpc), 32) * 32;
    rpc = pb_cdiv(rows_per_group, p.rows_per_wg);
    p.ranges_per_group = (int)rpc;
    split = clips * rpc;
  }
  p.split = split;
  const bool in_loop_prologue = p.ranges_per_group == 0 && (has_row_scale || affine);
  static const bool dma_on = !(getenv("PB_WGRAD_DMA") && atoi(getenv("PB_WGRAD_DMA")) == 0);
  p.dma = !in_loop_prologue && dma_on;
  return p;
}

bool wgrad_is_big(int K, int N, pb_dtype dtype) {
  return dtype == PB_BF16 && (N > 64 || K > 64) && (N % 8) == 0 && (K % 8) == 0;
}

// K = N = 16 without a prologue: the four-rows-per-line form (pw_wgrad_kernel<DIAG>)
bool wgrad_is_fold4(int64_t M, int K, int N, pb_dtype dtype, bool affine, bool row_scale) {
  static const int on = getenv("PB_WGRAD_FOLD4") ? atoi(getenv("PB_WGRAD_FOLD4")) : 1;
  return on && dtype == PB_BF16 && K == 16 && N == 16 && M % 4 == 0 && M >= 4096 && !affine && !row_scale;
}

}  // namespace

namespace {
// launch plan of the 64x64 (bf16) / 32x32 (fp32) tile kernel for small N, K
struct WgSmallPlan {
  int tile, n_tiles, k_tiles;
  int64_t msplit, rows_per_wg;
};
WgSmallPlan wgrad_small_plan(int64_t M, int K, int N, pb_dtype dtype) {
  WgSmallPlan p;
  p.tile = dtype == PB_BF16 ? 64 : 32;
  p.n_tiles = (int)pb_cdiv(N, p.tile);
  p.k_tiles = (int)pb_cdiv(K, p.tile);
  const int64_t tiles = (int64_t)p.n_tiles * p.k_tiles;
  // ~512 workgroups (4 per CU is what the 36 KB of LDS admits), each writing one partial tile
  static const int small_wgs = getenv("PB_WGRAD_SMALL_WGS") ? atoi(getenv("PB_WGRAD_SMALL_WGS")) : 512;
  int64_t msplit = pb_cdiv(small_wgs, tiles);
  const int64_t max_split = pb_cdiv(M, 128);
  if (msplit > max_split) msplit = max_split;
  if (msplit > 65535) msplit = 65535;
  if (msplit < 1) msplit = 1;
  p.rows_per_wg = pb_cdiv(pb_cdiv(M, msplit), 128) * 128;
  p.msplit = pb_cdiv(M, p.rows_per_wg);
  return p;
}
}  // namespace

extern "C" int64_t pb_pwconv_bwd_weight_ws_bytes(int64_t M, int K, int N, pb_dtype dtype, int has_affine, int has_row_scale,
                                                 int64_t rows_per_group, int device) {
  if (M <= 0 || K <= 0 || N <= 0) return 0;
  if (wgrad_is_big(K, N, dtype)) {
    const WgPlan p = wgrad128_plan(M, K, N, has_row_scale != 0, has_affine != 0, rows_per_group > 0 ? rows_per_group : 1, true, device);
    return p.split * (int64_t)N * K * 4;
  }
  if (wgrad_is_fold4(M, K, N, dtype, has_affine != 0, has_row_scale != 0)) return wgrad_small_plan(M / 4, 64, 64, dtype).msplit * (int64_t)4096;
  return wgrad_small_plan(M, K, N, dtype).msplit * (int64_t)N * K * 4;
}

extern "C" int pb_pwconv_bwd_weight_ws(const void* dy, const void* x, float* dw, int64_t M, int K, int N,
                                       pb_dtype dtype, const float* in_scale, const float* in_shift, pb_act in_act,
                                       float in_slope, const float* row_scale, int64_t rows_per_group, void* ws,
                                       int64_t ws_bytes, pb_reduce_job* deferred, int device, void* stream) {
  PB_CHECK_ARG(dy && x && dw, "pb_pwconv_bwd_weight: null pointer");
  PB_CHECK_ARG(M >= 0 && K > 0 && N > 0, "pb_pwconv_bwd_weight: bad sizes");
  PB_CHECK_ARG(dtype == PB_F32 || dtype == PB_BF16, "pb_pwconv_bwd_weight: bad dtype %d", (int)dtype);
  PB_CHECK_ARG(!row_scale || rows_per_group > 0, "pb_pwconv_bwd_weight: row_scale needs rows_per_group > 0");
  PB_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0, "pb_pwconv_bwd_weight: unaligned pointer");
  if (deferred) *deferred = pb_reduce_job{nullptr, dw, 0, 0, 0, 1};
  if (M == 0) return 0;
  const bool affine = in_scale || in_shift || in_act != PB_ACT_NONE;
  const int64_t need = pb_pwconv_bwd_weight_ws_bytes(M, K, N, dtype, affine, row_scale != nullptr, rows_per_group, device);
  PB_CHECK_ARG(ws && ((uintptr_t)ws % 16) == 0 && ws_bytes >= need, "pb_pwconv_bwd_weight: workspace of %lld bytes needed (16-byte aligned)",
               (long long)need);
  PB_ENTER(device);
  WgArgs a;
  a.P = (float*)ws;
  a.dY = dy; a.X = x; a.dW = dw; a.M = M; a.K = K; a.N = N;
  a.in_scale = in_scale; a.in_shift = in_shift; a.in_act = (int)in_act; a.in_slope = in_slope;
  a.row_scale = row_scale; a.rows_per_group = rows_per_group > 0 ? rows_per_group : 1;
  int64_t nsplit;
  if (wgrad_is_fold4(M, K, N, dtype, affine, row_scale != nullptr)) {
    a.M = M / 4; a.K = 64; a.N = 64;
    const WgSmallPlan p = wgrad_small_plan(a.M, 64, 64, dtype);
    a.n_tiles = 1; a.k_tiles = 1; a.rows_per_wg = p.rows_per_wg;
    a.ranges_per_group = 0; a.split = (int)p.msplit;
    nsplit = 4 * p.msplit;                                  // (four diagonal blocks per row range, each a partial [16][16])
    hipLaunchKernelGGL((pw_wgrad_kernel<bf16_t, 4, 4, false, true>), dim3(1, (unsigned)p.msplit), dim3(256), 0, (hipStream_t)stream, a);
  } else if (wgrad_is_big(K, N, dtype)) {
    const WgPlan p = wgrad128_plan(M, K, N, row_scale != nullptr, affine, a.rows_per_group, true, device);
    a.n_tiles = p.n_tiles; a.k_tiles = p.k_tiles;
    const int64_t tiles128 = (int64_t)a.n_tiles * a.k_tiles;
    a.rows_per_wg = p.rows_per_wg;
    a.ranges_per_group = p.ranges_per_group;
    a.split = (int)p.split;
    nsplit = p.split;
    const int ns = p.ns;
    const dim3 grid((unsigned)(tiles128 * 8 * pb_cdiv(p.split, 8)));
    if (p.dma) {
      static bool attr_done[16] = {false};
      if (!attr_done[device & 15]) {
        PB_HIP(hipFuncSetAttribute((const void*)pw_wgrad128_dma_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 6 * 16384));
        PB_HIP(hipFuncSetAttribute((const void*)pw_wgrad128_dma_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 16384));
        attr_done[device & 15] = true;
      }
      if (ns == 6) hipLaunchKernelGGL((pw_wgrad128_dma_kernel<6>), grid, dim3(256), 6 * 16384, (hipStream_t)stream, a);
      else if (ns == 4) hipLaunchKernelGGL((pw_wgrad128_dma_kernel<4>), grid, dim3(256), 4 * 16384, (hipStream_t)stream, a);
      else hipLaunchKernelGGL((pw_wgrad128_dma_kernel<3>), grid, dim3(256), 3 * 16384, (hipStream_t)stream, a);
    } else {
      hipLaunchKernelGGL((pw_wgrad128_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, a);
    }
  } else {
    const WgSmallPlan p = wgrad_small_plan(M, K, N, dtype);
    a.n_tiles = p.n_tiles; a.k_tiles = p.k_tiles; a.rows_per_wg = p.rows_per_wg;
    a.ranges_per_group = 0; a.split = (int)p.msplit;
    nsplit = p.msplit;
    dim3 grid((unsigned)(p.n_tiles * p.k_tiles), (unsigned)p.msplit), block(256);
    if (dtype == PB_BF16)
      hipLaunchKernelGGL((pw_wgrad_kernel<bf16_t, 4, 4>), grid, block, 0, (hipStream_t)stream, a);
    else
      hipLaunchKernelGGL((pw_wgrad_kernel<float, 2, 2>), grid, block, 0, (hipStream_t)stream, a);
  }
  PB_LAUNCH_CHECK();
  // dw += sum of the row ranges' partial tiles, in range order
  const pb_reduce_job job = {(const float*)ws, dw, (int64_t)N * K, (int64_t)N * K, (int)nsplit, 1};
  if (deferred) { *deferred = job; return 0; }
  if (pb_launch_partials_reduce(&job, 1, (hipStream_t)stream)) { pb_set_error("pb_pwconv_bwd_weight: combine launch failed"); return 2; }
  return 0;
}


// ----------------------------------------------------------------------------- dense conv (implicit GEMM)
namespace {

// mode 0: w[Cout][Cin][taps] -> wp[Cout][taps][Cin]                       (forward operand)
// mode 1: w[Cout][Cin][taps] -> wp[Cin][taps][Cout], taps reversed        (backward-data operand)
// mode 2: wp[Cout][taps][Cin] -> w[Cout][Cin][taps]                       (weight gradient back to the reference layout)
__global__ void conv_weight_prep_kernel(const float* in, float* out, int Cout, int Cin, int taps, int mode) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)Cout * Cin * taps;
  if (i >= total) return;
  if (mode == 0) {            // i indexes out [co][tap][ci]
    const int ci = (int)(i % Cin), tap = (int)((i / Cin) % taps), co = (int)(i / ((int64_t)Cin * taps));
    out[i] = in[((int64_t)co * Cin + ci) * taps + tap];
  } else if (mode == 1) {     // i indexes out [ci][tap][co]
    const int co = (int)(i % Cout), tap = (int)((i / Cout) % taps), ci = (int)(i / ((int64_t)Cout * taps));
    out[i] = in[((int64_t)co * Cin + ci) * taps + (taps - 1 - tap)];
  } else {                    // i indexes out [co][ci][tap]
    const int tap = (int)(i % taps), ci = (int)((i / taps) % Cin), co = (int)(i / ((int64_t)Cin * taps));
    out[i] = in[((int64_t)co * taps + tap) * Cin + ci];
  }
}

template <typename T, int NT>
void launch_conv_nt(const PwArgs& a, int64_t total, int device, hipStream_t st) {
  static int cached[16] = {0};
  const int d = device & 15;
  if (cached[d] == 0) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pw_gemm_kernel<T, NT, false, true, false, true>, 256, 0) != hipSuccess || per_cu < 1)
      per_cu = 1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1) cus = 256;
    cached[d] = per_cu * cus;
  }
  int64_t blocks = cached[d];
  if (blocks > total) blocks = total;
  blocks = pb_cdiv(blocks, 8) * 8;
  if (a.stats && blocks > PB_STAT_SLOTS) blocks = PB_STAT_SLOTS;   // one statistics slot per workgroup
  hipLaunchKernelGGL((pw_gemm_kernel<T, NT, false, true, false, true>), dim3((unsigned)blocks), dim3(256), 0, st, a);
}

template <typename T>
int launch_conv(PwArgs a, int device, hipStream_t st) {
  const int nt = pick_nt(a.N, sizeof(T) == 4);
  a.n_tiles = (int)pb_cdiv(a.N, nt * 16);
  a.m_tiles = pb_cdiv(a.M, BM);
  const int64_t total = a.n_tiles == 1 ? a.m_tiles : pb_cdiv(a.m_tiles, 8) * 8 * a.n_tiles;
  if (total > 0x3fffffffLL) { pb_set_error("pb_conv3d_fwd: too many tiles"); return 1; }
  switch (nt) {
    case 2: launch_conv_nt<T, 2>(a, total, device, st); break;
    case 4: launch_conv_nt<T, 4>(a, total, device, st); break;
    case 6: launch_conv_nt<T, 6>(a, total, device, st); break;
    case 8: launch_conv_nt<T, 8>(a, total, device, st); break;
    default:
      if constexpr (sizeof(T) == 2) launch_conv_nt<T, 12>(a, total, device, st);
      else launch_conv_nt<T, 8>(a, total, device, st);
      break;
  }
  return 0;
}

int check_conv(const pb_conv_shape* s, pb_dtype dtype, const char* who) {
  PB_CHECK_ARG(s, "%s: null shape", who);
  PB_CHECK_ARG(s->N > 0 && s->Cin > 0 && s->Cout > 0 && s->Ti > 0 && s->Hi > 0 && s->Wi > 0 && s->To > 0 && s->Ho > 0 &&
               s->Wo > 0 && s->kt > 0 && s->kh > 0 && s->kw > 0 && s->st > 0 && s->sh > 0 && s->sw > 0,
               "%s: non-positive dimension", who);
  PB_CHECK_ARG(dtype == PB_F32 || dtype == PB_BF16, "%s: bad dtype", who);
  const int vec = dtype == PB_BF16 ? 8 : 4;
  PB_CHECK_ARG(s->Cin % vec == 0 && s->Cout % vec == 0, "%s: channel counts must be multiples of %d (Cin=%d Cout=%d)", who,
               vec, s->Cin, s->Cout);
  PB_CHECK_ARG((s->To - 1) * s->st - s->pt < s->Ti + s->pt + s->kt && (s->Ho - 1) * s->sh - s->ph < s->Hi + s->ph + s->kh &&
               (s->Wo - 1) * s->sw - s->pw < s->Wi + s->pw + s->kw, "%s: output extent inconsistent", who);
  return 0;
}

void fill_conv(PwArgs& a, const pb_conv_shape* s) {
  a.cTi = s->Ti; a.cHi = s->Hi; a.cWi = s->Wi; a.cTo = s->To; a.cHo = s->Ho; a.cWo = s->Wo;
  a.ckt = s->kt; a.ckh = s->kh; a.ckw = s->kw; a.cst = s->st; a.csh = s->sh; a.csw = s->sw;
  a.cpt = s->pt; a.cph = s->ph; a.cpw = s->pw;
}

}  // namespace

extern "C" int pb_conv_weight_prep(const float* in, float* out, int Cout, int Cin, int taps, int mode, int device,
                                   void* stream) {
  PB_CHECK_ARG(in && out && Cout > 0 && Cin > 0 && taps > 0 && mode >= 0 && mode <= 2, "pb_conv_weight_prep: bad args");
  PB_ENTER(device);
  const int64_t total = (int64_t)Cout * Cin * taps;
  hipLaunchKernelGGL(conv_weight_prep_kernel, dim3((unsigned)pb_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, in,
                     out, Cout, Cin, taps, mode);
  PB_LAUNCH_CHECK();
  return 0;
}

extern "C" int pb_conv3d_fwd(const void* x, const float* wp, void* y, const pb_conv_shape* s, pb_dtype dtype,
                             const float* out_scale, const float* out_shift, pb_act out_act, float out_slope,
                             float* stats, int device, void* stream) {
  PB_CHECK_ARG(x && wp && y, "pb_conv3d_fwd: null pointer");
  if (int rc = check_conv(s, dtype, "pb_conv3d_fwd")) return rc;
  PB_ENTER(device);
  PwArgs a = {};
  a.X = x; a.W = wp; a.Y = y;
  a.M = (int64_t)s->N * s->To * s->Ho * s->Wo;
  a.K = s->Cin; a.N = s->Cout; a.ldw = s->kt * s->kh * s->kw * s->Cin; a.w_transposed = 0;
  a.rows_per_group = 1;
  a.out_scale = out_scale; a.out_shift = out_shift; a.out_act = (int)out_act; a.out_slope = out_slope; a.stats = stats;
  fill_conv(a, s);
  int rc = dtype == PB_BF16 ? launch_conv<bf16_t>(a, device, (hipStream_t)stream) : launch_conv<float>(a, device, (hipStream_t)stream);
  if (rc) return rc;
  PB_LAUNCH_CHECK();
  return 0;
}

// Dense-conv weight gradient: grid z = filter taps, per-range partial tiles into the workspace, one fixed-order
// reduction.  bf16 with channel counts that are multiples of 8: the LDS-DMA kernel (any stride; zero16 = any 16 zero
// bytes of device memory for the padding voxels); everything else: the register-staged kernels.
namespace {
struct ConvWgPlan {
  bool dma, big;          // big: 128x128 tiles (bf16), else the 32x32 fp32 kernel
  int n_tiles, k_tiles;
  int64_t split, rows_per_wg;
};
ConvWgPlan conv_wgrad_plan(const pb_conv_shape* s, pb_dtype dtype) {
  ConvWgPlan p = {};
  const int taps = s->kt * s->kh * s->kw;
  const int64_t M = (int64_t)s->N * s->To * s->Ho * s->Wo;
  static const bool dma_on = !(getenv("PB_CONV_WGRAD_DMA") && atoi(getenv("PB_CONV_WGRAD_DMA")) == 0);
  p.big = dtype == PB_BF16;
  p.dma = dma_on && p.big && s->Cin % 8 == 0 && s->Cout % 8 == 0 && M < (1LL << 31);
  if (p.big) {
    p.n_tiles = (int)pb_cdiv(s->Cout, 128);
    p.k_tiles = (int)pb_cdiv(s->Cin, 128);
    const int64_t t128 = (int64_t)p.n_tiles * p.k_tiles;
    // ~1536 (DMA kernel) / 2048 workgroups over (128x128 tiles) x taps x row ranges
    static const int dma_wgs = getenv("PB_CONV_WGRAD_DMA_WGS") ? atoi(getenv("PB_CONV_WGRAD_DMA_WGS")) : 1536;
    static const int reg_wgs = getenv("PB_CONV_WGRAD_WGS") ? atoi(getenv("PB_CONV_WGRAD_WGS")) : 2048;
    int64_t split = p.dma ? dma_wgs / (t128 * taps) : pb_cdiv(reg_wgs, t128 * taps);
    if (split > pb_cdiv(M, 256)) split = pb_cdiv(M, 256);
    if (split < 1) split = 1;
    p.rows_per_wg = pb_cdiv(pb_cdiv(M, split), 32) * 32;
    p.split = pb_cdiv(M, p.rows_per_wg);
  } else {
    p.n_tiles = (int)pb_cdiv(s->Cout, 32);
    p.k_tiles = (int)pb_cdiv(s->Cin, 32);
    const int64_t tiles = (int64_t)p.n_tiles * p.k_tiles;
    int64_t msplit = pb_cdiv(2048, tiles * taps);
    const int64_t max_split = pb_cdiv(M, 128);
    if (msplit > max_split) msplit = max_split;
    if (msplit > 65535) msplit = 65535;
    if (msplit < 1) msplit = 1;
    p.rows_per_wg = pb_cdiv(pb_cdiv(M, msplit), 128) * 128;
    p.split = pb_cdiv(M, p.rows_per_wg);
  }
  return p;
}
}  // namespace

extern "C" int64_t pb_conv3d_bwd_weight_ws_bytes(const pb_conv_shape* s, pb_dtype dtype, int device) {
  (void)device;
  if (!s || s->N <= 0 || s->Cin <= 0 || s->Cout <= 0 || s->To <= 0 || s->Ho <= 0 || s->Wo <= 0) return 0;
  return conv_wgrad_plan(s, dtype).split * (int64_t)s->Cout * (s->kt * s->kh * s->kw) * s->Cin * 4;
}

extern "C" int pb_conv3d_bwd_weight_ws(const void* dy, const void* x, float* dwp, const pb_conv_shape* s, pb_dtype dtype,
                                       void* ws, int64_t ws_bytes, const void* zero16, int device, void* stream) {
  PB_CHECK_ARG(dy && x && dwp && ws && zero16, "pb_conv3d_bwd_weight_ws: null pointer");
  if (int rc = check_conv(s, dtype, "pb_conv3d_bwd_weight_ws")) return rc;
  PB_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0 && ((uintptr_t)zero16 % 16) == 0 && ((uintptr_t)ws % 16) == 0,
               "pb_conv3d_bwd_weight_ws: unaligned pointer");
  PB_CHECK_ARG(ws_bytes >= pb_conv3d_bwd_weight_ws_bytes(s, dtype, device), "pb_conv3d_bwd_weight_ws: workspace of %lld bytes needed",
               (long long)pb_conv3d_bwd_weight_ws_bytes(s, dtype, device));
  PB_ENTER(device);
  WgArgs a = {};
  a.dY = dy; a.X = x; a.dW = dwp; a.P = (float*)ws; a.zero = zero16;
  a.M = (int64_t)s->N * s->To * s->Ho * s->Wo;
  a.K = s->Cin; a.N = s->Cout; a.rows_per_group = 1;
  a.cTi = s->Ti; a.cHi = s->Hi; a.cWi = s->Wi; a.cTo = s->To; a.cHo = s->Ho; a.cWo = s->Wo;
  a.ckt = s->kt; a.ckh = s->kh; a.ckw = s->kw; a.cst = s->st; a.csh = s->sh; a.csw = s->sw;
  a.cpt = s->pt; a.cph = s->ph; a.cpw = s->pw;
  const int taps = s->kt * s->kh * s->kw;
  PB_CHECK_ARG(taps <= 65535, "pb_conv3d_bwd_weight_ws: too many taps");
  const ConvWgPlan p = conv_wgrad_plan(s, dtype);
  a.n_tiles = p.n_tiles; a.k_tiles = p.k_tiles;
  a.rows_per_wg = p.rows_per_wg;
  a.split = (int)p.split;
  a.ranges_per_group = 0;
  const int64_t tiles = (int64_t)p.n_tiles * p.k_tiles;
  if (p.dma)
    hipLaunchKernelGGL((pw_wgrad128_dma_kernel<3, true>), dim3((unsigned)(tiles * 8 * pb_cdiv(p.split, 8)), 1, (unsigned)taps), dim3(256),
                       3 * 16384, (hipStream_t)stream, a);
  else if (p.big)
    hipLaunchKernelGGL((pw_wgrad128_kernel<true>), dim3((unsigned)(tiles * 8 * pb_cdiv(p.split, 8)), 1, (unsigned)taps), dim3(256), 0,
                       (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((pw_wgrad_kernel<float, 2, 2, true>), dim3((unsigned)tiles, (unsigned)p.split, (unsigned)taps), dim3(256), 0,
                       (hipStream_t)stream, a);
  PB_LAUNCH_CHECK();
  const int64_t nk = (int64_t)s->Cout * taps * s->Cin;
  const pb_reduce_job job = {(const float*)ws, dwp, nk, nk, (int)p.split, 1};
  if (pb_launch_partials_reduce(&job, 1, (hipStream_t)stream)) { pb_set_error("pb_conv3d_bwd_weight_ws: combine launch failed"); return 2; }
  return 0;
}
