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
};

template <typename T>
__device__ __forceinline__ Vec16<T> prologue(Vec16<T> v, const PwArgs& a, int64_t m, int k) {
  constexpr int VEC = ElemTraits<T>::VEC;
  if (a.in_scale || a.in_shift || a.in_act != PB_ACT_NONE || a.row_scale) {
    const float* rs = a.row_scale ? a.row_scale + (m / a.rows_per_group) * a.K : nullptr;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float x = v.get(i);
      if (k + i < a.K) {
        if (a.in_scale) x *= a.in_scale[k + i];
        if (a.in_shift) x += a.in_shift[k + i];
        x = act_fwd(x, a.in_act, a.in_slope);
        if (rs) x *= rs[k + i];
      }
      v.set(i, x);
    }
  }
  return v;
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

template <typename T, int NT>
__global__ __launch_bounds__(256) void pw_gemm_kernel(PwArgs a) {
  constexpr int VEC = ElemTraits<T>::VEC;
  constexpr int KS = 128 / (int)sizeof(T);     // K elements staged per step (64 bf16 / 32 f32)
  constexpr int MK = 128 / (int)sizeof(T) / 2; // K consumed per fragment read (32 bf16 / 16 f32)
  constexpr int BN = NT * 16;
  constexpr int EPI_LD = BN + VEC;
  constexpr int TILE_BYTES = (BM + BN) * LDB;
  constexpr int EPI_BYTES = 4 * 16 * EPI_LD * (int)sizeof(T);
  constexpr int MAIN_BYTES = TILE_BYTES > EPI_BYTES ? TILE_BYTES : EPI_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[MAIN_BYTES + 4 * BN * 2 * 4];

  // block id -> (row panel, column tile); blocks b and b+8 share an XCD, so all column tiles of
  // a row panel are given ids with equal b % 8 (speed only; placement is never relied upon).
  const int64_t b = blockIdx.x;
  const int xcd = (int)(b & 7);
  const int64_t jj = b >> 3;
  const int nt = (int)(jj % a.n_tiles);
  const int64_t mt = (jj / a.n_tiles) * 8 + xcd;
  if (mt >= a.m_tiles) return;
  const int64_t m0 = mt * BM;
  const int n0 = nt * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T* X = (const T*)a.X;
  T* Y = (T*)a.Y;
  unsigned char* As = smem;
  unsigned char* Ws = smem + BM * LDB;
  const bool xvec = (a.K % VEC) == 0;
  const bool wvec = (a.ldw % 4) == 0;

  f32x4 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < a.K; k0 += KS) {
    const int kc = min(KS, a.K - k0);
    const int kpad = (kc + MK - 1) / MK * MK;
    const int cprp = kpad / VEC;
    // ---- stage the activation panel chunk (prologue fused)
    for (int f = tid; f < BM * cprp; f += 256) {
      const int r = f / cprp, c = f - r * cprp;
      const int64_t m = m0 + r;
      const int k = k0 + c * VEC;
      Vec16<T> v = vec_zero<T>();
      if (m < a.M && k < a.K) v = prologue<T>(load_row_chunk<T>(X, m, k, a.K, xvec), a, m, k);
      vec_store<T>((T*)(As + r * LDB) + c * VEC, v);
    }
    // ---- stage the weight tile, fp32 -> T
    if (!a.w_transposed) {
      for (int f = tid; f < BN * cprp; f += 256) {
        const int r = f / cprp, c = f - r * cprp;
        const int n = n0 + r, k = k0 + c * VEC;
        Vec16<T> v = vec_zero<T>();
        if (n < a.N) {
          const float* src = a.W + (int64_t)n * a.ldw + k;
          if (wvec && k + VEC <= a.K) {
#pragma unroll
            for (int q = 0; q < VEC / 4; ++q) {
              f32x4 t = *reinterpret_cast<const f32x4*>(src + 4 * q);
#pragma unroll
              for (int i = 0; i < 4; ++i) v.set(4 * q + i, t[i]);
            }
          } else {
#pragma unroll
            for (int i = 0; i < VEC; ++i)
              if (k + i < a.K) v.set(i, src[i]);
          }
        }
        vec_store<T>((T*)(Ws + r * LDB) + c * VEC, v);
      }
    } else {
      // Wl[n][k] = w[k*ldw + n]: read 4 consecutive n per thread (coalesced along n), scatter to LDS
      constexpr int NQ = BN / 4;
      for (int f = tid; f < kpad * NQ; f += 256) {
        const int kl = f / NQ, nq = f - kl * NQ;
        const int n = n0 + nq * 4, k = k0 + kl;
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        if (k < a.K) {
          const float* src = a.W + (int64_t)k * a.ldw + n;
          if (wvec && n + 4 <= a.N) {
            f32x4 q = *reinterpret_cast<const f32x4*>(src);
            t[0] = q[0]; t[1] = q[1]; t[2] = q[2]; t[3] = q[3];
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (n + i < a.N) t[i] = src[i];
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) ((T*)(Ws + (nq * 4 + i) * LDB))[kl] = from_f32<T>(t[i]);
      }
    }
    __syncthreads();
    // ---- MFMA over the staged chunk
    const int frow = lane & 15, fk = lane >> 4;
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
    __syncthreads();
  }

  // ---- epilogue: affine/bias + activation, optional per-column statistics, LDS-transposed store
  T* epi = (T*)smem + wave * 16 * EPI_LD;
  float* sred = (float*)(smem + MAIN_BYTES);   // [4 waves][BN][2]
  float csum[NT], csq[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) csum[j] = csq[j] = 0.f;
  const int ccol = lane & 15, rgrp = lane >> 4;
  const bool yvec = (a.N % VEC) == 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + j * 16 + ccol;
      const bool nok = n < a.N;
      const float osc = (a.out_scale && nok) ? a.out_scale[n] : 1.f;
      const float osh = (a.out_shift && nok) ? a.out_shift[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rl = rgrp * 4 + r;
        const int64_t m = m0 + wave * 32 + i * 16 + rl;
        const float v = act_fwd(acc[i][j][r] * osc + osh, a.out_act, a.out_slope);
        const T tv = from_f32<T>(v);
        epi[rl * EPI_LD + j * 16 + ccol] = tv;
        if (a.stats && nok && m < a.M) {
          const float fv = to_f32(tv);
          csum[j] += fv;
          csq[j] += fv * fv;
        }
      }
    }
    __syncthreads();
    constexpr int CPR = BN / VEC;
    for (int f = lane; f < 16 * CPR; f += 64) {
      const int r = f / CPR, c = f - r * CPR;
      const int64_t m = m0 + wave * 32 + i * 16 + r;
      const int n = n0 + c * VEC;
      if (m < a.M && n < a.N) {
        if (yvec && n + VEC <= a.N) {
          vec_store<T>(Y + m * a.N + n, vec_load<T>(epi + r * EPI_LD + c * VEC));
        } else {
          for (int e = 0; e < VEC; ++e)
            if (n + e < a.N) Y[m * a.N + n + e] = epi[r * EPI_LD + c * VEC + e];
        }
      }
    }
    __syncthreads();
  }
  if (a.stats) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      float s = csum[j], q = csq[j];
      s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
      q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
      if (lane < 16) {
        sred[(wave * BN + j * 16 + lane) * 2 + 0] = s;
        sred[(wave * BN + j * 16 + lane) * 2 + 1] = q;
      }
    }
    __syncthreads();
    for (int c = tid; c < BN; c += 256) {
      const int n = n0 + c;
      if (n < a.N) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          s += sred[(w * BN + c) * 2 + 0];
          q += sred[(w * BN + c) * 2 + 1];
        }
        atomicAdd(a.stats + n, s);
        atomicAdd(a.stats + a.N + n, q);
      }
    }
  }
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
};

// Each wave streams 32-row blocks of dY[:, n-tile] and X[:, k-tile] through its private LDS
// slab (row-major, exactly as loaded) and feeds MFMA with the reduction index m on the "k" axis:
// bf16 fragments come from ds_read_b64_tr_b16 (the hardware transposing read), f32 from plain
// 4-byte reads.  No workgroup barrier in the loop: waves are independent until the final atomics.
template <typename T, int TNW, int TKW>
__global__ __launch_bounds__(256) void pw_wgrad_kernel(WgArgs a) {
  constexpr int VEC = ElemTraits<T>::VEC;
  constexpr int TN = TNW * 16, TK = TKW * 16;
  constexpr int LDN = (TN + VEC) * (int)sizeof(T), LDK = (TK + VEC) * (int)sizeof(T);   // bytes
  constexpr int SLAB = 32 * (LDN + LDK);
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

  f32x4 acc[TNW][TKW];
#pragma unroll
  for (int i = 0; i < TNW; ++i)
#pragma unroll
    for (int j = 0; j < TKW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int64_t mb = mbeg + wave * 32; mb < mend; mb += 128) {
    constexpr int CN = TN / VEC, CK = TK / VEC;
    for (int f = lane; f < 32 * CN; f += 64) {
      const int r = f / CN, c = f - r * CN;
      const int64_t m = mb + r;
      const int n = n0 + c * VEC;
      Vec16<T> v = vec_zero<T>();
      if (m < mend && n < a.N) v = load_row_chunk<T>(dY, m, n, a.N, nvec);
      vec_store<T>((T*)(Ds + r * LDN) + c * VEC, v);
    }
    for (int f = lane; f < 32 * CK; f += 64) {
      const int r = f / CK, c = f - r * CK;
      const int64_t m = mb + r;
      const int k = k0 + c * VEC;
      Vec16<T> v = vec_zero<T>();
      if (m < mend && k < a.K) v = prologue<T>(load_row_chunk<T>(X, m, k, a.K, kvec), pa, m, k);
      vec_store<T>((T*)(Xs + r * LDK) + c * VEC, v);
    }
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
        for (int j = 0; j < TKW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
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
  // C/D map: row (= n) = (lane>>4)*4 + r, col (= k) = lane & 15
#pragma unroll
  for (int i = 0; i < TNW; ++i)
#pragma unroll
    for (int j = 0; j < TKW; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + i * 16 + (lane >> 4) * 4 + r;
        const int k = k0 + j * 16 + (lane & 15);
        if (n < a.N && k < a.K) atomicAdd(a.dW + (int64_t)n * a.K + k, acc[i][j][r]);
      }
}

int pick_nt(int N) {
  // cost = padded MFMA width + 16 per column tile (each extra tile re-reads the X panel through L2)
  const int cand[5] = {2, 4, 6, 8, 12};
  int best = 2;
  int64_t best_cost = INT64_MAX;
  for (int c : cand) {
    const int bn = c * 16;
    const int64_t tiles = pb_cdiv(N, bn);
    const int64_t cost = tiles * bn + 16 * tiles;
    if (cost < best_cost) { best = c; best_cost = cost; }
  }
  return best;
}

template <typename T>
int launch_gemm(const PwArgs& a0, hipStream_t st) {
  PwArgs a = a0;
  const int nt = pick_nt(a.N);
  a.n_tiles = (int)pb_cdiv(a.N, nt * 16);
  a.m_tiles = pb_cdiv(a.M, BM);
  const int64_t blocks = pb_cdiv(a.m_tiles, 8) * 8 * a.n_tiles;
  if (blocks > 0x7fffffffLL) { pb_set_error("pb_pwconv_fwd: grid too large"); return 1; }
  dim3 grid((unsigned)blocks), block(256);
  switch (nt) {
    case 2: hipLaunchKernelGGL((pw_gemm_kernel<T, 2>), grid, block, 0, st, a); break;
    case 4: hipLaunchKernelGGL((pw_gemm_kernel<T, 4>), grid, block, 0, st, a); break;
    case 6: hipLaunchKernelGGL((pw_gemm_kernel<T, 6>), grid, block, 0, st, a); break;
    case 8: hipLaunchKernelGGL((pw_gemm_kernel<T, 8>), grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL((pw_gemm_kernel<T, 12>), grid, block, 0, st, a); break;
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
  int rc = dtype == PB_BF16 ? launch_gemm<bf16_t>(a, (hipStream_t)stream) : launch_gemm<float>(a, (hipStream_t)stream);
  if (rc) return rc;
  PB_LAUNCH_CHECK();
  return 0;
}

extern "C" int pb_pwconv_bwd_weight(const void* dy, const void* x, float* dw, int64_t M, int K, int N,
                                    pb_dtype dtype, const float* in_scale, const float* in_shift, pb_act in_act,
                                    float in_slope, const float* row_scale, int64_t rows_per_group, int device,
                                    void* stream) {
  PB_CHECK_ARG(dy && x && dw, "pb_pwconv_bwd_weight: null pointer");
  PB_CHECK_ARG(M >= 0 && K > 0 && N > 0, "pb_pwconv_bwd_weight: bad sizes");
  PB_CHECK_ARG(dtype == PB_F32 || dtype == PB_BF16, "pb_pwconv_bwd_weight: bad dtype %d", (int)dtype);
  PB_CHECK_ARG(!row_scale || rows_per_group > 0, "pb_pwconv_bwd_weight: row_scale needs rows_per_group > 0");
  PB_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0, "pb_pwconv_bwd_weight: unaligned pointer");
  if (M == 0) return 0;
  PB_ENTER(device);
  WgArgs a;
  a.dY = dy; a.X = x; a.dW = dw; a.M = M; a.K = K; a.N = N;
  a.in_scale = in_scale; a.in_shift = in_shift; a.in_act = (int)in_act; a.in_slope = in_slope;
  a.row_scale = row_scale; a.rows_per_group = rows_per_group > 0 ? rows_per_group : 1;
  const int tile = dtype == PB_BF16 ? 64 : 32;
  a.n_tiles = (int)pb_cdiv(N, tile);
  a.k_tiles = (int)pb_cdiv(K, tile);
  const int64_t tiles = (int64_t)a.n_tiles * a.k_tiles;
  int64_t msplit = pb_cdiv(2048, tiles);
  const int64_t max_split = pb_cdiv(M, 128);
  if (msplit > max_split) msplit = max_split;
  if (msplit > 65535) msplit = 65535;
  a.rows_per_wg = pb_cdiv(pb_cdiv(M, msplit), 128) * 128;
  msplit = pb_cdiv(M, a.rows_per_wg);
  dim3 grid((unsigned)tiles, (unsigned)msplit), block(256);
  if (dtype == PB_BF16)
    hipLaunchKernelGGL((pw_wgrad_kernel<bf16_t, 4, 4>), grid, block, 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((pw_wgrad_kernel<float, 2, 2>), grid, block, 0, (hipStream_t)stream, a);
  PB_LAUNCH_CHECK();
  return 0;
}
