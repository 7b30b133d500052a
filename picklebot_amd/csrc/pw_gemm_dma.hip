// Pointwise GEMM with both operands streamed by LDS-DMA:  Y[M,N] = X[M,K] * Wb[g(m)][N,K]^T  (bf16).
//
// pw_gemm.hip's kernel stages X through registers one K stage ahead; with K >= 480 (the projection and
// the backward-data of the expansion convolutions of the last MobileNet stages) that is a chain of 8-15
// dependent HBM round trips per tile - measured 0.35-1.0 TB/s.  Here every byte of a stage, the X block
// AND the weight slice, goes HBM/L2 -> LDS by global_load_lds_dwordx4 into a ring of NS stage buffers,
// NS-1 stages ahead of the MFMAs, also across tile boundaries (the ring does not drain for the
// epilogue).  No registers, no commit pass, one barrier per stage.
//
// The weights arrive pre-converted to bf16 (pb_pw_weight_prep: once per optimiser step and orientation,
// it also replaces the per-call transpose of backward-data) and, for the squeeze-excite projection,
// pre-scaled per clip: Y_n = (X_n * s_n) W^T = X_n (W * s_n)^T, so the SE scale never touches the
// activation stream.  Row panels are then clip-aligned (a panel never straddles two clips).
//
// LDS image of a stage: rows of 64 bf16 = 128 B, unpadded (one DMA instruction writes 1 KiB = 8 rows
// linearly); ds_read_b128 bank conflicts are avoided by XOR-swizzling the 16-byte chunk index with
// (row >> 1) & 7 on the SOURCE address.  Chunks past K are not loaded on the X side (stale but finite:
// the whole ring is zeroed once at kernel start and afterwards only ever holds real data) and are zeroed
// on the weight side, so they contribute exactly 0.
#include "pb_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int BM = 128;

struct DmaArgs {
  const bf16_t* X;
  const bf16_t* WB;          // [groups][N][K]
  bf16_t* Y;
  int64_t M;
  int K, N;
  int64_t w_group_stride;    // N*K, or 0 when there is one weight matrix
  int64_t rows_per_group;    // > 0: row panels are clip-aligned, panel p -> (clip, sub)
  int panels_per_group;
  const float* out_scale;
  const float* out_shift;
  int out_act;
  float out_slope;
  float* stats;
  int n_tiles;
  int m_tiles;
  int step_g, step_nt;       // (grid / 8) divided by n_tiles: quotient and remainder (the kernel's tile cursor step)
  // implicit-GEMM convolution (CONV instantiations): X is [N,Ti,Hi,Wi,Cin], row m = output voxel, K = taps*Cin with
  // k = tap*Cin + c; a 16-byte chunk (8 channels) of a row comes from the voxel shifted by the tap, or from `zero`
  int Ti, Hi, Wi, To, Ho, Wo, pt, ph, pw, kh, kw, taps, Cin;
  unsigned cin_magic;        // ceil(2^32 / Cin): k / Cin == umulhi(k, cin_magic) for k < 65536
  const bf16_t* zero;        // >= 16 bytes of zeros in device memory
  const bf16_t* R;           // residual [M][N] added to the output in the store pass (y = P(x) W^T + r), or null
};

__device__ __forceinline__ void glds16(const void* gsrc, unsigned char* lds_dst) {
  typedef __attribute__((address_space(3))) unsigned char* lds_b;
  const unsigned dst = (unsigned)(uintptr_t)(lds_b)lds_dst;
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

template <int N> __device__ __forceinline__ void wait_vmcnt();
#define PB_VMCNT(n) template <> __device__ __forceinline__ void wait_vmcnt<n>() { asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); }
PB_VMCNT(0) PB_VMCNT(5) PB_VMCNT(6) PB_VMCNT(7) PB_VMCNT(8) PB_VMCNT(10) PB_VMCNT(12) PB_VMCNT(14) PB_VMCNT(16) PB_VMCNT(20)
PB_VMCNT(15) PB_VMCNT(18) PB_VMCNT(21) PB_VMCNT(24) PB_VMCNT(30)
#undef PB_VMCNT

template <int NT, int NS, bool CONV>
__global__ __launch_bounds__(256) void pw_gemm_dma_kernel(DmaArgs a) {
  typedef bf16_t T;
  constexpr int VEC = 8, KS = 64, MK = 32;
  constexpr int BN = NT * 16;
  constexpr int WCH = (NT + 1) / 2;                  // weight chunks (DMA instructions) per thread and stage (rows rounded up to 32)
  constexpr int A_STAGE = BM * 128;                  // bytes
  constexpr int W_STAGE = WCH * 4096;
  constexpr int STAGE = A_STAGE + W_STAGE;
  constexpr int GPS = 4 + WCH;                       // DMA instructions per thread and stage
  constexpr int EPI_LD = BN + VEC;
  extern __shared__ __attribute__((aligned(1024))) unsigned char dsmem[];
  unsigned char* ring = dsmem;                                   // [NS][A | W]
  T* epi_base = (T*)(dsmem + NS * STAGE);                        // [4 waves][16][EPI_LD]
  float* sred = (float*)(dsmem + NS * STAGE);                    // [4][BN][2]: aliases the staging tiles (flush_stats only)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nk = (a.K + KS - 1) / KS;
  const int n_tiles = a.n_tiles, m_tiles = a.m_tiles;
  const int total = n_tiles == 1 ? m_tiles : ((m_tiles + 7) / 8) * 8 * n_tiles;
  // Tile t = ((g * n_tiles + nt) * 8 + r8): row panel g*8 + r8, column tile nt.  Workgroups b, b+8, b+16, ... (one XCD)
  // hold the column tiles of the same row panels, so the X panel is fetched from HBM once and re-read through that XCD's
  // L2.  The grid is a multiple of 8, so r8 is fixed per workgroup and (g, nt) advance by a constant pair per step - no
  // division in the loop; when the grid is also a multiple of 8*n_tiles (the host arranges that when it can), nt is
  // fixed too: one weight slice and one statistics flush per workgroup.
  const int r8 = (int)(blockIdx.x & 7);
  struct Cur { int t, g, nt; };
  auto cur_valid = [&](const Cur& c) { return c.g * 8 + r8 < m_tiles; };
  auto cur_step = [&](Cur& c) {
    do {
      c.t += (int)gridDim.x;
      c.nt += a.step_nt;
      c.g += a.step_g;
      if (c.nt >= n_tiles) { c.nt -= n_tiles; ++c.g; }
    } while (c.t < total && !cur_valid(c));
  };
  // row panel -> first row, end row (exclusive) and weight group
  auto panel = [&](int mt, int64_t& m0, int64_t& mend, int& grp) {
    if (a.rows_per_group > 0) {
      grp = mt / a.panels_per_group;
      const int sub = mt - grp * a.panels_per_group;
      m0 = (int64_t)grp * a.rows_per_group + (int64_t)sub * BM;
      mend = min((int64_t)(grp + 1) * a.rows_per_group, m0 + BM);
    } else {
      grp = 0;
      m0 = (int64_t)mt * BM;
      mend = min(a.M, m0 + BM);
    }
  };

  // ring starts out as zeros: from then on every byte in it is real (finite) data
  for (int i = tid; i < NS * STAGE / 16; i += 256) *reinterpret_cast<u32x4*>(ring + i * 16) = u32x4{0u, 0u, 0u, 0u};
  __shared__ int taptab[CONV ? 64 : 1];                  // tap -> dt | dh << 8 | dw << 16
  if (CONV && tid < a.taps) {
    const int dw_ = tid % a.kw, r_ = tid / a.kw;
    taptab[tid] = (r_ / a.kh) | ((r_ % a.kh) << 8) | (dw_ << 16);
  }

  // this thread's chunks of a stage: A chunk f = i*256 + tid -> row f>>3, LDS chunk f&7 holds source chunk
  // (f&7) ^ swz(row); same for the weight slice
  int a_row[4], a_cs[4], w_row[WCH], w_cs[WCH];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f = i * 256 + tid;
    a_row[i] = f >> 3;
    a_cs[i] = ((f & 7) ^ ((a_row[i] >> 1) & 7)) * VEC;
  }
#pragma unroll
  for (int i = 0; i < WCH; ++i) {
    const int f = i * 256 + tid;
    w_row[i] = f >> 3;
    w_cs[i] = ((f & 7) ^ ((w_row[i] >> 1) & 7)) * VEC;
  }
  const int wave_off = __builtin_amdgcn_readfirstlane(wave * 1024);
  // the tile being REQUESTED (the issue cursor): this thread's source addresses at k = 0, set once per tile
  const T* a_src[CONV ? 1 : 4];
  const T* w_src[WCH];
  int rq_t[CONV ? 4 : 1], rq_h[CONV ? 4 : 1], rq_w[CONV ? 4 : 1], rq_pix[CONV ? 4 : 1];
  auto set_rows = [&](const Cur& c) {
    int64_t m0, mend;
    int grp;
    panel(c.g * 8 + r8, m0, mend, grp);
    const T* wg = a.WB + (int64_t)grp * a.w_group_stride;
    const int n0 = c.nt * BN;
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
      const int n = min(n0 + w_row[i], a.N - 1);             // columns past N: any valid row (never stored)
      w_src[i] = wg + (int64_t)n * a.K + w_cs[i];
    }
    if constexpr (CONV) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = (int)min(m0 + a_row[i], a.M - 1);
        const int wo = m % a.Wo, r1 = m / a.Wo;
        const int ho = r1 % a.Ho, r2 = r1 / a.Ho;
        const int to = r2 % a.To, n = r2 / a.To;
        rq_t[i] = to - a.pt; rq_h[i] = ho - a.ph; rq_w[i] = wo - a.pw;
        rq_pix[i] = ((n * a.Ti + rq_t[i]) * a.Hi + rq_h[i]) * a.Wi + rq_w[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t m = min(m0 + a_row[i], a.M - 1);       // rows past the panel: any valid row (never stored)
        a_src[i] = a.X + m * a.K + a_cs[i];
      }
    }
  };

  auto issue = [&](int ks, int buf) {
    const int k0 = ks * KS;
    unsigned char* A = ring + buf * STAGE + wave_off;
    unsigned char* W = ring + buf * STAGE + A_STAGE + wave_off;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + a_cs[i];
      if constexpr (CONV) {
        const T* src = a.zero;                               // padding voxels and the K tail
        if (k < a.K) {
          const int tap = (int)__umulhi((unsigned)k, a.cin_magic);
          const int c = k - tap * a.Cin;
          const int tp = taptab[tap];
          const int dt = tp & 255, dh = (tp >> 8) & 255, dw = tp >> 16;
          if ((unsigned)(rq_t[i] + dt) < (unsigned)a.Ti && (unsigned)(rq_h[i] + dh) < (unsigned)a.Hi &&
              (unsigned)(rq_w[i] + dw) < (unsigned)a.Wi)
            src = a.X + (int64_t)(rq_pix[i] + (dt * a.Hi + dh) * a.Wi + dw) * a.Cin + c;
        }
        glds16(src, A + i * 4096);
      } else {
        if (k < a.K) glds16(a_src[i] + k0, A + i * 4096);
      }
    }
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
      if (k0 + w_cs[i] < a.K) glds16(w_src[i] + k0, W + i * 4096);
      else *reinterpret_cast<u32x4*>(ring + buf * STAGE + A_STAGE + (i * 256 + tid) * 16) = u32x4{0u, 0u, 0u, 0u};
    }
  };

  // Accumulators: the weight slice is the MFMA's A operand and the activation rows its B operand, so a lane holds
  // FOUR CONSECUTIVE OUTPUT COLUMNS n = j*16 + rgrp*4 + (0..3) of one row m = i*16 + ccol: the epilogue converts them in
  // pairs and stages 8 bytes per lane and block (instead of four scattered 2-byte writes).
  f32x4 acc[2][NT];
  f32x4 csum[NT], csq[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) csum[j] = csq[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fk = lane >> 4;
  const int ccol = lane & 15, rgrp = lane >> 4;
  const bool plain = !a.out_scale && !a.out_shift && a.out_act == PB_ACT_NONE;
  // fragment read offsets inside a stage (bytes): rows wave*32 + frow (+16) of A, rows j*16 + frow of W
  const int a_rd = (wave * 32 + frow) * 128;
  const int a_sw = ((wave * 32 + frow) >> 1) & 7;            // rows r and r+16 share the swizzle
  const int w_sw = (frow >> 1) & 7;                          // rows j*16 + frow: (j*8 + frow/2) & 7 = frow/2

  auto flush_stats = [&](int n0) {
    lds_barrier();                                           // the staging tiles (aliased by sred) are idle
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = csum[j][r], q = csq[j][r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          s += __shfl_xor(s, o, 64);
          q += __shfl_xor(q, o, 64);
        }
        if (ccol == 0) {
          sred[(wave * BN + j * 16 + rgrp * 4 + r) * 2 + 0] = s;
          sred[(wave * BN + j * 16 + rgrp * 4 + r) * 2 + 1] = q;
        }
      }
      csum[j] = csq[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    lds_barrier();
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
    lds_barrier();
  };

  Cur ct;
  ct.t = (int)blockIdx.x;
  {
    const int u = ct.t >> 3;
    ct.g = u / n_tiles;
    ct.nt = u - ct.g * n_tiles;
  }
  if (ct.t < total && !cur_valid(ct)) cur_step(ct);
  if (ct.t >= total) return;                         // uniform
  lds_barrier();                                     // ring zero-fill visible before the first DMA lands on it
  // issue cursor: (ci, ksi) is the next stage to request; it runs NS-1 stages ahead of the stage being multiplied
  Cur ci = ct;
  int ksi = 0, ahead = 0;
  auto advance_cursor = [&]() {
    if (++ksi == nk) {
      ksi = 0;
      cur_step(ci);
      if (ci.t < total) set_rows(ci);
    }
  };
  set_rows(ci);
#pragma unroll
  for (int s0 = 0; s0 < NS - 1; ++s0) {
    if (ci.t < total) {
      issue(ksi, s0);
      advance_cursor();
      ++ahead;
    }
  }
  int buf = 0;
  // one K stage: wait for the oldest requested stage, request the next one, multiply
  auto stage = [&](auto first_tag) {
    constexpr bool FIRST = decltype(first_tag)::value;
    // the oldest of `ahead` requested stages must have landed: the younger ones stay in flight.  (Stores count in vmcnt
    // on gfx9 and may retire out of order with loads: right after an epilogue only vmcnt(0) is safe, which is what a
    // 2-deep ring asks for anyway; deeper rings drain the stores at the end of the epilogue.)
    if (ahead >= 3) wait_vmcnt<2 * GPS>();
    else if (ahead == 2) wait_vmcnt<GPS>();
    else wait_vmcnt<0>();
    lds_barrier();       // everybody's part of the stage has landed; everybody is done reading the previous stage
    if (ci.t < total) {
      int nb = buf + NS - 1;
      if (nb >= NS) nb -= NS;
      issue(ksi, nb);
      advance_cursor();
    } else {
      --ahead;
    }
    const unsigned char* As = ring + buf * STAGE;
    const unsigned char* Ws = As + A_STAGE;
#pragma unroll
    for (int kk = 0; kk < KS; kk += MK) {
      const int ca = (((kk >> 3) + fk) ^ a_sw) << 4, cw = (((kk >> 3) + fk) ^ w_sw) << 4;
      const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(As + a_rd + ca);
      const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(As + a_rd + 16 * 128 + ca);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const bf16x8 bw = *reinterpret_cast<const bf16x8*>(Ws + (j * 16 + frow) * 128 + cw);
        if (FIRST && kk == 0) {
          acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw, a0, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw, a1, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        } else {
          acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw, a0, acc[0][j], 0, 0, 0);
          acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw, a1, acc[1][j], 0, 0, 0);
        }
      }
    }
    if (++buf == NS) buf = 0;
  };

  // ---- epilogue of one tile: affine/bias + activation, per-column statistics, LDS-transposed store (the ring keeps
  // filling underneath: the staging tiles are not part of it)
  auto epilogue = [&](auto plain_tag, auto stats_tag, int64_t m0, int64_t mend, int n0) {
    constexpr bool PLAIN = decltype(plain_tag)::value, STATS = decltype(stats_tag)::value;
    T* epi = epi_base + wave * 16 * EPI_LD;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool m_ok = m0 + wave * 32 + i * 16 + ccol < mend;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        f32x4 v = acc[i][j];
        if constexpr (!PLAIN) {
          const int n = n0 + j * 16 + rgrp * 4;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int nc = min(n + r, a.N - 1);              // (columns past N are never stored)
            float x = v[r];
            if (a.out_scale) x *= a.out_scale[nc];
            if (a.out_shift) x += a.out_shift[nc];
            v[r] = act_fwd(x, a.out_act, a.out_slope);
          }
        }
        if constexpr (STATS) {
          if (m_ok) {                                        // (fp32 values, before the bf16 rounding of the store)
            csum[j] += v;
            csq[j] += v * v;
          }
        }
        const bf16x4 o = __builtin_convertvector(v, bf16x4);
        *reinterpret_cast<bf16x4*>(epi + ccol * EPI_LD + j * 16 + rgrp * 4) = o;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // staging tiles are wave-private: no workgroup barrier
      __builtin_amdgcn_wave_barrier();
      {
        const int row = lane >> 2;
        const int64_t m = m0 + wave * 32 + i * 16 + row;
        T* yrow = a.Y + m * a.N + n0;
        const T* rrow = a.R ? a.R + m * a.N + n0 : nullptr;
#pragma unroll
        for (int c = (lane & 3); c < 2 * NT; c += 4) {
          if (m < mend && n0 + c * VEC < a.N) {
            Vec16<T> v = vec_load<T>(epi + row * EPI_LD + c * VEC);
            if (rrow) {                                      // residual connection (MobileViT: x + attn(x), x + ff(x))
              const Vec16<T> rv = vec_load<T>(rrow + c * VEC);
#pragma unroll
              for (int e = 0; e < VEC; ++e) v.set(e, v.get(e) + rv.get(e));
            }
            vec_store<T>(yrow + c * VEC, v);
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
  };

  while (ct.t < total) {
    stage(std::true_type{});
    for (int ks = 1; ks < nk; ++ks) stage(std::false_type{});
    int64_t m0, mend;
    int grp;
    panel(ct.g * 8 + r8, m0, mend, grp);
    const int n0 = ct.nt * BN;
    if (plain) {
      if (a.stats) epilogue(std::true_type{}, std::true_type{}, m0, mend, n0);
      else epilogue(std::true_type{}, std::false_type{}, m0, mend, n0);
    } else {
      if (a.stats) epilogue(std::false_type{}, std::true_type{}, m0, mend, n0);
      else epilogue(std::false_type{}, std::false_type{}, m0, mend, n0);
    }
    const int nt_done = ct.nt;
    cur_step(ct);
    // several column tiles: the sums go out when the workgroup moves to another column tile (never, when the grid is a
    // multiple of 8*n_tiles: one flush per workgroup instead of one per tile)
    if (a.stats && n_tiles > 1 && (ct.t >= total || ct.nt != nt_done)) flush_stats(n0);
  }
  if (a.stats && n_tiles == 1) flush_stats(0);
}

// weight preparation: out[g][r][c] (bf16) = w[...] (* scale[g][k]); one thread per 8 output elements
struct PrepArgs {
  const float* w;        // [N][K] fp32
  bf16_t* out;
  const float* scale;    // [G][K] or null
  int N, K, G, transpose;   // transpose: out is [K][N] (element (k, n) = w[n][k])
};

__global__ __launch_bounds__(256) void pw_weight_prep_kernel(PrepArgs a) {
  const int R = a.transpose ? a.K : a.N, C = a.transpose ? a.N : a.K;
  const int64_t per_g = (int64_t)R * C;
  const int64_t total8 = (int64_t)a.G * per_g / 8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total8; i += (int64_t)gridDim.x * 256) {
    const int64_t e = i * 8;
    const int g = (int)(e / per_g);
    const int64_t rem = e - (int64_t)g * per_g;
    const int r = (int)(rem / C), c0 = (int)(rem - (int64_t)r * C);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = c0 + j;
      float v;
      if (a.transpose) v = a.w[(int64_t)c * a.K + r];           // out[k = r][n = c]
      else v = a.w[(int64_t)r * a.K + c] * (a.scale ? a.scale[(int64_t)g * a.K + c] : 1.f);
      o[j] = (bf16_t)v;
    }
    *reinterpret_cast<bf16x8*>(a.out + e) = o;
  }
}

// the same conversion for a whole model in one launch: desc[t] = {w, out, N << 32 | K, mode} (device array)
//   mode 0: out[N][K] = w;  1: out[K][N] = w^T;  2 / 3: blockdiag(B, B, B, B) [4R][4C] of the mode-0 / mode-1 matrix B [R][C]
//   (the 4-rows-per-line form of the K = 16 layers: [M,16] x W^T == [M/4,64] x blockdiag(W,W,W,W)^T)
__global__ __launch_bounds__(256) void pw_weight_prep_batch_kernel(const int64_t* desc) {
  const int64_t* d = desc + 4 * blockIdx.y;
  const float* w = (const float*)d[0];
  bf16_t* out = (bf16_t*)d[1];
  const int N = (int)(d[2] >> 32), K = (int)(d[2] & 0xffffffffLL), mode = (int)d[3];
  const bool transpose = mode & 1, fold = mode >= 2;
  const int R = transpose ? K : N, C = transpose ? N : K;           // the base matrix B
  const int RO = fold ? 4 * R : R, CO = fold ? 4 * C : C;
  const int64_t total8 = (int64_t)RO * CO / 8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total8; i += (int64_t)gridDim.x * 256) {
    const int64_t e = i * 8;
    const int ro = (int)(e / CO), co0 = (int)(e - (int64_t)ro * CO);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int co = co0 + j;
      const int r = fold ? ro % R : ro, c = fold ? co % C : co;
      float v = transpose ? w[(int64_t)c * K + r] : w[(int64_t)r * K + c];
      if (fold && ro / R != co / C) v = 0.f;
      o[j] = (bf16_t)v;
    }
    *reinterpret_cast<bf16x8*>(out + e) = o;
  }
}

// column-tile width: padded MFMA width + 16 per tile (every extra tile re-reads the X panel through L2).  Only
// tiles of <= 112 columns leave LDS for two workgroups per CU with a 2-deep ring, whose epilogues overlap each other's
// streaming: measured on the whole step (tools/bench_ab.sh) that beats the 128/192-wide tiles with a deeper ring.
int pick_nt_dma(int N, int K) {
  (void)K;
  static const int max_nt = getenv("PB_GEMM_DMA_MAXNT") ? atoi(getenv("PB_GEMM_DMA_MAXNT")) : 7;
  int best = 2;
  int64_t best_cost = INT64_MAX;
  for (int c = 2; c <= 7; ++c) {
    if (c > max_nt) continue;
    const int bn = c * 16;
    const int64_t tiles = pb_cdiv(N, bn);
    const int64_t cost = tiles * bn + 16 * tiles;
    if (cost < best_cost) { best = c; best_cost = cost; }
  }
  return best;
}

template <int NT, int NS, bool CONV = false>
int launch_dma(const DmaArgs& a_in, int64_t total, int device, hipStream_t st) {
  constexpr int BN = NT * 16;
  constexpr size_t lds = (size_t)NS * (BM * 128 + ((NT + 1) / 2) * 4096) + 4 * 16 * (BN + 8) * 2;
  static int resident_dev[16] = {0};                 // per device: the attribute is per (function, device)
  int& resident = resident_dev[device & 15];
  if (resident == 0) {
    if (hipFuncSetAttribute((const void*)pw_gemm_dma_kernel<NT, NS, CONV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      pb_set_error("pw_gemm_dma: cannot reserve %d bytes of LDS", (int)lds);
      return 1;
    }
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pw_gemm_dma_kernel<NT, NS, CONV>, 256, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1) cus = 256;
    resident = per_cu * cus;
  }
  DmaArgs a = a_in;
  int64_t blocks = resident;
  if (a.stats && blocks > PB_STAT_SLOTS) blocks = PB_STAT_SLOTS;   // one statistics slot per workgroup
  if (blocks > total) blocks = total;
  // a multiple of 8 (the kernel's tile cursor) and, when the chip is not left short of workgroups by it, of 8*n_tiles:
  // every workgroup then keeps ONE column tile (one weight slice, one statistics flush)
  const int64_t unit = 8 * (int64_t)a.n_tiles;
  if (a.n_tiles > 1 && blocks >= 2 * unit) blocks = blocks / unit * unit;
  else blocks = pb_cdiv(blocks, 8) * 8;
  if (a.stats && blocks > PB_STAT_SLOTS) blocks = PB_STAT_SLOTS;   // (a multiple of 8)
  const int64_t gq = blocks / 8;
  a.step_g = (int)(gq / a.n_tiles);
  a.step_nt = (int)(gq % a.n_tiles);
  hipLaunchKernelGGL((pw_gemm_dma_kernel<NT, NS, CONV>), dim3((unsigned)blocks), dim3(256), lds, st, a);
  return 0;
}

}  // namespace

extern "C" int pb_pw_weight_prep(const float* w, int N, int K, int transpose, const float* row_scale, int G,
                                 void* out_bf16, int device, void* stream) {
  PB_CHECK_ARG(w && out_bf16, "pb_pw_weight_prep: null pointer");
  PB_CHECK_ARG(N > 0 && K > 0 && G >= 1, "pb_pw_weight_prep: bad sizes");
  PB_CHECK_ARG(((int64_t)N * K) % 8 == 0 && (transpose ? N : K) % 8 == 0, "pb_pw_weight_prep: inner dimension must be a multiple of 8");
  PB_CHECK_ARG(!(transpose && (row_scale || G != 1)), "pb_pw_weight_prep: no scaling with transpose");
  PB_CHECK_ARG(((uintptr_t)out_bf16 % 16) == 0, "pb_pw_weight_prep: unaligned output");
  PB_ENTER(device);
  PrepArgs a;
  a.w = w; a.out = (bf16_t*)out_bf16; a.scale = row_scale; a.N = N; a.K = K; a.G = G; a.transpose = transpose;
  const int64_t total8 = (int64_t)G * N * K / 8;
  int64_t blocks = pb_cdiv(total8, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(pw_weight_prep_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  PB_LAUNCH_CHECK();
  return 0;
}

extern "C" int pb_pw_weight_prep_batch(const int64_t* desc, int count, int device, void* stream) {
  PB_CHECK_ARG(desc && count > 0 && count <= 65535, "pb_pw_weight_prep_batch: bad arguments");
  PB_ENTER(device);
  hipLaunchKernelGGL(pw_weight_prep_batch_kernel, dim3(16, (unsigned)count), dim3(256), 0, (hipStream_t)stream, desc);
  PB_LAUNCH_CHECK();
  return 0;
}

__attribute__((visibility("hidden"))) int pb_pw_stream_fwd(const void* x, const void* wb, void* y, int64_t M, int K, int N,
                                                           const float* out_scale, const float* out_shift, pb_act out_act,
                                                           float out_slope, float* stats, const void* residual, int device,
                                                           hipStream_t st, int stat_fold = 1);

__attribute__((visibility("hidden"))) int pb_pw_stream_wide_fwd(const void* x, const void* wb, void* y, int64_t M, int K, int N,
                                                                int groups, int64_t rows_per_group, const float* out_scale,
                                                                const float* out_shift, pb_act out_act, float out_slope,
                                                                float* stats, const void* residual, int device, hipStream_t st);

extern "C" int pb_pwconv_fwd_bf16w(const void* x, const void* wb, void* y, int64_t M, int K, int N, int groups,
                                   int64_t rows_per_group, const float* out_scale, const float* out_shift,
                                   pb_act out_act, float out_slope, float* stats, const void* residual, int device,
                                   void* stream) {
  PB_CHECK_ARG(x && wb && y, "pb_pwconv_fwd_bf16w: null pointer");
  PB_CHECK_ARG(M > 0 && K > 0 && N > 0 && K % 8 == 0 && N % 8 == 0, "pb_pwconv_fwd_bf16w: K and N must be multiples of 8 (M=%lld K=%d N=%d)",
               (long long)M, K, N);
  PB_CHECK_ARG(groups >= 1 && (groups == 1 || (rows_per_group > 0 && rows_per_group * groups == M)),
               "pb_pwconv_fwd_bf16w: groups * rows_per_group must equal M");
  PB_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)wb % 16) == 0, "pb_pwconv_fwd_bf16w: unaligned pointer");
  PB_ENTER(device);
  PB_CHECK_ARG(!residual || (((uintptr_t)residual % 16) == 0 && !stats), "pb_pwconv_fwd_bf16w: residual must be 16-byte aligned and excludes stats");
  if (groups == 1) {
    // short reductions: the weight-stationary streaming kernel (pw_stream.hip)
    const int rc = pb_pw_stream_fwd(x, wb, y, M, K, N, out_scale, out_shift, out_act, out_slope, stats, residual, device, (hipStream_t)stream);
    if (rc > 0) return rc;
    if (rc == 0) {
      PB_LAUNCH_CHECK();
      return 0;
    }
  }
  {
    // long reductions (also with per-clip weights): the columns-over-waves form of the streaming kernel (pw_stream_wide.hip)
    const int rc = pb_pw_stream_wide_fwd(x, wb, y, M, K, N, groups, rows_per_group, out_scale, out_shift, out_act, out_slope, stats,
                                         residual, device, (hipStream_t)stream);
    if (rc > 0) return rc;
    if (rc == 0) {
      PB_LAUNCH_CHECK();
      return 0;
    }
  }
  DmaArgs a = {};
  a.X = (const bf16_t*)x; a.WB = (const bf16_t*)wb; a.Y = (bf16_t*)y; a.M = M; a.K = K; a.N = N;
  a.R = (const bf16_t*)residual;
  a.out_scale = out_scale; a.out_shift = out_shift; a.out_act = (int)out_act; a.out_slope = out_slope; a.stats = stats;
  if (groups > 1) {
    a.w_group_stride = (int64_t)N * K;
    a.rows_per_group = rows_per_group;
    a.panels_per_group = (int)pb_cdiv(rows_per_group, BM);
    a.m_tiles = a.panels_per_group * groups;
  } else {
    a.w_group_stride = 0; a.rows_per_group = 0; a.panels_per_group = 0;
    a.m_tiles = (int)pb_cdiv(M, BM);
  }
  const int nt = pick_nt_dma(N, K);
  a.n_tiles = (int)pb_cdiv(N, nt * 16);
  const int64_t total = a.n_tiles == 1 ? a.m_tiles : pb_cdiv(a.m_tiles, 8) * 8 * a.n_tiles;
  PB_CHECK_ARG(total <= 0x3fffffffLL, "pb_pwconv_fwd_bf16w: too many tiles");
  hipStream_t st = (hipStream_t)stream;
  int rc = 0;
  // a 2-stage ring leaves room for two workgroups per CU, whose epilogues overlap each other's streaming.  Long reductions
  // (K >= 400: 8-15 stages per tile, each waited for with one stage in flight) run ONE workgroup per CU with a 4-stage ring -
  // three stages in flight: 32-34 -> 27 us on the 960 -> 160 layers of the 7x7 stage, 34.6 -> 30.5 on 480 -> 80; shorter
  // reductions measured slower that way (PB_GEMM_DMA_NS / PB_GEMM_DMA_NS_MINK)
  static const int deep_ns = getenv("PB_GEMM_DMA_NS") ? atoi(getenv("PB_GEMM_DMA_NS")) : 4;
  static const int deep_mink = getenv("PB_GEMM_DMA_NS_MINK") ? atoi(getenv("PB_GEMM_DMA_NS_MINK")) : 400;
  if (deep_ns == 4 && K >= deep_mink) {
    switch (nt) {
      case 2: rc = launch_dma<2, 4>(a, total, device, st); break;
      case 3: rc = launch_dma<3, 4>(a, total, device, st); break;
      case 4: rc = launch_dma<4, 4>(a, total, device, st); break;
      case 5: rc = launch_dma<5, 4>(a, total, device, st); break;
      case 6: rc = launch_dma<6, 4>(a, total, device, st); break;
      default: rc = launch_dma<7, 4>(a, total, device, st); break;
    }
  } else {
    switch (nt) {
      case 2: rc = launch_dma<2, 2>(a, total, device, st); break;
      case 3: rc = launch_dma<3, 2>(a, total, device, st); break;
      case 4: rc = launch_dma<4, 2>(a, total, device, st); break;
      case 5: rc = launch_dma<5, 2>(a, total, device, st); break;
      case 6: rc = launch_dma<6, 2>(a, total, device, st); break;
      default: rc = launch_dma<7, 2>(a, total, device, st); break;
    }
  }
  if (rc) return rc;
  PB_LAUNCH_CHECK();
  return 0;
}

__attribute__((visibility("hidden"))) int pb_conv3d_halo_fwd(const void* x, const void* wb, void* y, const pb_conv_shape* s,
                                                             const void* zero16, const float* out_scale, const float* out_shift,
                                                             pb_act out_act, float out_slope, float* stats, int device, hipStream_t st);

extern "C" int pb_pwconv_fwd_fold4(const void* x, const void* wb4, void* y, int64_t M, int N, pb_act out_act, float out_slope,
                                   float* stats, int device, void* stream) {
  PB_CHECK_ARG(x && wb4 && y, "pb_pwconv_fwd_fold4: null pointer");
  PB_CHECK_ARG(M > 0 && M % 4 == 0 && N > 0 && N % 8 == 0 && 4 * N <= 96, "pb_pwconv_fwd_fold4: M %% 4 == 0, N %% 8 == 0, N <= 24 (M=%lld N=%d)",
               (long long)M, N);
  PB_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)wb4 % 16) == 0, "pb_pwconv_fwd_fold4: unaligned pointer");
  PB_ENTER(device);
  const int rc = pb_pw_stream_fwd(x, wb4, y, M / 4, 64, 4 * N, nullptr, nullptr, out_act, out_slope, stats, nullptr, device, (hipStream_t)stream,
                                  stats ? 4 : 1);
  if (rc > 0) return rc;
  if (rc < 0) {
    pb_set_error("pb_pwconv_fwd_fold4: shape not taken by the streaming kernel (M=%lld N=%d)", (long long)M, N);
    return 1;
  }
  PB_LAUNCH_CHECK();
  return 0;
}

// Dense Conv3d (groups = 1, stride 1) as implicit GEMM on the same kernel: M = output voxels, K = taps*Cin.
extern "C" int pb_conv3d_fwd_bf16w(const void* x, const void* wb, void* y, const pb_conv_shape* s, const void* zero16,
                                   const float* out_scale, const float* out_shift, pb_act out_act, float out_slope,
                                   float* stats, int device, void* stream) {
  PB_CHECK_ARG(x && wb && y && s && zero16, "pb_conv3d_fwd_bf16w: null pointer");
  PB_CHECK_ARG(s->st == 1 && s->sh == 1 && s->sw == 1, "pb_conv3d_fwd_bf16w: stride 1 only");
  const int taps = s->kt * s->kh * s->kw;
  PB_CHECK_ARG(s->N > 0 && s->Cin > 0 && s->Cout > 0 && s->Cin % 8 == 0 && s->Cout % 8 == 0,
               "pb_conv3d_fwd_bf16w: channel counts must be multiples of 8 (Cin=%d Cout=%d)", s->Cin, s->Cout);
  PB_CHECK_ARG(taps >= 1 && taps <= 64 && s->kt < 256 && s->kh < 256 && s->kw < 256 && (int64_t)taps * s->Cin < 65536,
               "pb_conv3d_fwd_bf16w: kernel too large (taps=%d Cin=%d)", taps, s->Cin);
  PB_CHECK_ARG(s->To > 0 && s->Ho > 0 && s->Wo > 0 && s->Ti > 0 && s->Hi > 0 && s->Wi > 0, "pb_conv3d_fwd_bf16w: empty tensor");
  const int64_t M = (int64_t)s->N * s->To * s->Ho * s->Wo;
  PB_CHECK_ARG(M < (1LL << 31) && (int64_t)s->N * s->Ti * s->Hi * s->Wi < (1LL << 30), "pb_conv3d_fwd_bf16w: tensor too large");
  PB_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)wb % 16) == 0 && ((uintptr_t)zero16 % 16) == 0,
               "pb_conv3d_fwd_bf16w: unaligned pointer");
  PB_ENTER(device);
  {
    // 3x3x3: the halo-resident kernel (conv_halo.hip) reads the input once per 32-channel chunk instead of once per tap
    const int rc = pb_conv3d_halo_fwd(x, wb, y, s, zero16, out_scale, out_shift, out_act, out_slope, stats, device, (hipStream_t)stream);
    if (rc >= 0) return rc;
  }
  DmaArgs a = {};
  a.X = (const bf16_t*)x; a.WB = (const bf16_t*)wb; a.Y = (bf16_t*)y; a.M = M; a.K = taps * s->Cin; a.N = s->Cout;
  a.out_scale = out_scale; a.out_shift = out_shift; a.out_act = (int)out_act; a.out_slope = out_slope; a.stats = stats;
  a.m_tiles = (int)pb_cdiv(M, BM);
  a.Ti = s->Ti; a.Hi = s->Hi; a.Wi = s->Wi; a.To = s->To; a.Ho = s->Ho; a.Wo = s->Wo;
  a.pt = s->pt; a.ph = s->ph; a.pw = s->pw; a.kh = s->kh; a.kw = s->kw; a.taps = taps; a.Cin = s->Cin;
  a.cin_magic = (unsigned)(((1ULL << 32) + (unsigned)s->Cin - 1) / (unsigned)s->Cin);
  a.zero = (const bf16_t*)zero16;
  // few row panels (small frames): narrower column tiles give the chip more workgroups
  int nt = pick_nt_dma(a.N, a.K);
  if (nt & 1) nt = nt + 1 > 6 ? 6 : nt + 1;          // (the convolution instantiations: 2, 4, 6)
  static const int min_tiles = getenv("PB_CONV_MIN_TILES") ? atoi(getenv("PB_CONV_MIN_TILES")) : 1200;
  while (nt > 2 && (int64_t)a.m_tiles * pb_cdiv(a.N, nt * 16) < min_tiles) nt -= 2;
  a.n_tiles = (int)pb_cdiv(a.N, nt * 16);
  const int64_t total = a.n_tiles == 1 ? a.m_tiles : pb_cdiv(a.m_tiles, 8) * 8 * a.n_tiles;
  hipStream_t st = (hipStream_t)stream;
  int rc = 0;
  static const int ns = getenv("PB_CONV_DMA_NS") ? atoi(getenv("PB_CONV_DMA_NS")) : 2;
  if (ns == 2) {
    if (nt == 2) rc = launch_dma<2, 2, true>(a, total, device, st);
    else if (nt == 4) rc = launch_dma<4, 2, true>(a, total, device, st);
    else rc = launch_dma<6, 2, true>(a, total, device, st);
  } else {
    if (nt == 2) rc = launch_dma<2, 4, true>(a, total, device, st);
    else if (nt == 4) rc = launch_dma<4, 4, true>(a, total, device, st);
    else rc = launch_dma<6, 3, true>(a, total, device, st);
  }
  if (rc) return rc;
  PB_LAUNCH_CHECK();
  return 0;
}
