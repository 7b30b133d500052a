// Pointwise GEMM with a long reduction (the projection layers and the backward-data of the expansions, K > 160):
//     Y[M,N] = X[M,K] * Wb[g(m)][N,K]^T   (bf16, fp32 accumulate)
//
// pw_gemm_dma.hip moves both operands by LDS-DMA; measured on MI355X that path carries about 12-16 B/clk per CU, a
// quarter of the ordinary vector-load path, and with X (152 MB on the 672 -> 112 layer) and the re-fetched weight slices
// (another 130 MB) both on it the kernel sat at 2.0-2.3 TB/s whatever the ring depth.  A wave only ever multiplies its OWN
// 32 rows of X: here X takes the ordinary vector-load path, 8 lanes per 128-byte row segment (fully coalesced: loading
// the MFMA operand layout directly - 16 bytes from each of 16 rows per quarter wave - measured no faster than the DMA),
// KD K-stages ahead in a register ring - 4 KiB per stage and wave, 96 KiB in flight per CU - and is turned into operand
// layout through a wave-private, XOR-swizzled LDS tile (no barrier: the rows are the wave's own).  Only the weight slice of
// a stage (BN x 64, shared by the 8 waves) goes through an LDS ring by LDS-DMA.  The weights
// are the MFMA's A operand, so a lane holds four consecutive output columns of a row: pairs of column blocks exchange
// halves (v_permlane16_swap) and every lane stores 16 contiguous bytes - no staging tile.
//
// Tiles: 256 rows (8 waves x 32) x BN columns; workgroups walk them persistently, the loads run D-1 stages ahead across
// tile boundaries.  With per-clip weights (squeeze-excite scale folded in, pb_pw_weight_prep) row panels are clip-aligned.
#include "pb_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int KTHREADS = 512;
constexpr int KD = 4;              // K stages in flight per wave (register ring of X fragments, LDS ring of weight slices)

struct KloopArgs {
  const bf16_t* X;
  const bf16_t* WB;          // [groups][N][K]
  bf16_t* Y;
  const bf16_t* R;           // residual [M][N] or null
  int64_t M;
  int K, N;
  int64_t w_group_stride;    // N*K, or 0 when there is one weight matrix
  int64_t rows_per_group;    // > 0: row panels are clip-aligned
  int panels_per_group;
  const float* out_scale;
  const float* out_shift;
  int out_act;
  float out_slope;
  float* stats;
  int n_tiles, m_tiles;
  int step_g, step_nt;       // (grid / 8) divided by n_tiles: quotient and remainder (the tile cursor's step)
  int dbg;                   // diagnosis only (PB_GEMM_DMA_DBG): 1 = no X loads, 2 = no W DMA, 4 = no stores, 8 = no MFMA
};

__device__ __forceinline__ void glds16k(const void* gsrc, unsigned char* lds_dst) {
  typedef __attribute__((address_space(3))) unsigned char* lds_b;
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_b)lds_dst);    // (wave-uniform by construction)
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

__device__ __forceinline__ int wave_of(unsigned tid) { return __builtin_amdgcn_readfirstlane((int)(tid >> 6)); }

template <int N> __device__ __forceinline__ void wait_vmk() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int ACT> __device__ __forceinline__ float act_k(float x, float slope) {
  if constexpr (ACT == PB_ACT_RELU) return x > 0.f ? x : 0.f;
  else if constexpr (ACT == PB_ACT_LEAKY_RELU) return x > 0.f ? x : x * slope;
  else if constexpr (ACT == PB_ACT_HSWISH) return x * fminf(fmaxf(x + 3.f, 0.f), 6.f) * (1.f / 6.f);
  else if constexpr (ACT == PB_ACT_HSIGMOID) return fminf(fmaxf(x * (1.f / 6.f) + 0.5f, 0.f), 1.f);
  else if constexpr (ACT == PB_ACT_SILU) return x / (1.f + __expf(-x));
  else return x;
}

template <int NT, int RB, bool STATS, bool HYB>
__global__ __launch_bounds__(KTHREADS) void pw_kloop_kernel(KloopArgs a) {
  typedef bf16_t T;
  constexpr int KS = 64;
  constexpr int KBM = 8 * 16 * RB;                             // rows per tile: 8 waves x RB blocks of 16
  constexpr int BN = NT * 16;
  constexpr int NP = NT / 2;
  constexpr int WCH = (BN * 8 + KTHREADS - 1) / KTHREADS;      // weight DMA instructions per thread and stage
  constexpr int W_STAGE = WCH * KTHREADS * 16;                 // bytes (rows rounded up)
  constexpr int VPS = 2 * RB + WCH;                                 // vector-memory instructions per thread and stage
  constexpr int XT = 16 * RB * 128;                            // bytes of a wave's X tile of one stage
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  // [KD][W_STAGE] weight ring | [2][BN] scale, shift | 8 KiB dump for filler DMAs | X tiles | [8][BN][2] sums
  // X tiles: [8 waves][2][XT], or with HYB (second half of the rows by LDS-DMA, a second path into the CU next to the
  // vector loads' miss queue) [8 waves][XT/2] for the loaded half and [8 waves][KD][XT/2] for the DMA half
  constexpr int XBYTES = HYB ? 8 * (XT / 2) * (1 + KD) : 8 * 2 * XT;
  constexpr int NLD = HYB ? RB : 2 * RB;                       // vector-load instructions per stage (8 rows each)
  float* osc = (float*)(smem + KD * W_STAGE);
  unsigned char* xs = smem + KD * W_STAGE + 2 * BN * 4 + 8192 + wave_of(threadIdx.x) * (HYB ? XT / 2 : 2 * XT);
  unsigned char* xd = smem + KD * W_STAGE + 2 * BN * 4 + 8192 + 8 * (XT / 2) + wave_of(threadIdx.x) * KD * (XT / 2);   // (HYB)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fk = lane >> 4;
  const int nk = (a.K + KS - 1) / KS;
  const int n_tiles = a.n_tiles, m_tiles = a.m_tiles;
  const int total = n_tiles == 1 ? m_tiles : ((m_tiles + 7) / 8) * 8 * n_tiles;

  // ---- tile cursor (see pw_gemm_dma.hip): t = ((g * n_tiles + nt) * 8 + r8)
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
  auto panel = [&](int mt, int64_t& m0, int64_t& mend, int& grp) {
    if (a.rows_per_group > 0) {
      grp = mt / a.panels_per_group;
      const int sub = mt - grp * a.panels_per_group;
      m0 = (int64_t)grp * a.rows_per_group + (int64_t)sub * KBM;
      mend = min((int64_t)(grp + 1) * a.rows_per_group, m0 + KBM);
    } else {
      grp = 0;
      m0 = (int64_t)mt * KBM;
      mend = min(a.M, m0 + KBM);
    }
  };

  // ---- weight DMA: chunk f = i*512 + tid -> row f>>3 of the slice, LDS chunk f&7 holds source chunk (f&7) ^ swz(row)
  int w_row[WCH], w_cs[WCH];
#pragma unroll
  for (int i = 0; i < WCH; ++i) {
    const int f = i * KTHREADS + tid;
    w_row[i] = f >> 3;
    w_cs[i] = ((f & 7) ^ ((w_row[i] >> 1) & 7)) * 8;
  }
  const int w_sw = (frow >> 1) & 7;                  // rows j*16 + frow: (j*8 + frow/2) & 7 = frow/2
  // X loads: instruction q covers rows q*8 .. q*8+7 of the wave's tile, lane -> (row q*8 + lane/8, chunk lane%8)
  const int lrow = lane >> 3, lch = lane & 7;
  int xw_off[2 * RB];                                // where this lane's 16 bytes of instruction q go in the LDS tile
#pragma unroll
  for (int q = 0; q < 2 * RB; ++q) {
    const int row = q * 8 + lrow;
    xw_off[q] = row * 128 + ((lch ^ ((row >> 1) & 7)) << 4);
  }
  const int xr_off = frow * 128;                     // fragment reads: row i*16 + frow, chunk (h*4 + fk) ^ swz
  int xd_ch[2 * RB];                                 // (HYB) the source chunk that belongs at this lane's linear DMA position
#pragma unroll
  for (int q = 0; q < 2 * RB; ++q) xd_ch[q] = (lch ^ (((q * 8 + lrow) >> 1) & 7)) * 8;

  // issue-side tile state
  const T* x_src[2 * RB];                            // this lane's row of each load instruction, at k = lch*8
  const T* w_src[WCH];
  auto set_rows = [&](const Cur& c) __attribute__((always_inline)) {
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
#pragma unroll
    for (int q = 0; q < 2 * RB; ++q) {
      const int64_t m = min(m0 + wave * (16 * RB) + q * 8 + lrow, mend - 1);    // rows past the panel: a valid row (never stored)
      x_src[q] = a.X + m * a.K;
    }
  };

  bf16x8 xa[KD][NLD];                              // [slot][row block][k half]
  auto issue = [&](auto slot_tag, int ks) __attribute__((always_inline)) {
    constexpr int SLOT = decltype(slot_tag)::value;
    const int k0 = ks * KS;
    // (the last stage may reach past K: the weights there are zero, the row data only has to be valid memory)
    const int ka = min(k0 + lch * 8, a.K - 8);
    if (!(a.dbg & 1))
#pragma unroll
    for (int q = 0; q < NLD; ++q) xa[SLOT][q] = *reinterpret_cast<const bf16x8*>(x_src[q] + ka);
    if constexpr (HYB) {
      if (!(a.dbg & 1))
#pragma unroll
      for (int q = RB; q < 2 * RB; ++q)
        glds16k(x_src[q] + min(k0 + xd_ch[q], a.K - 8), xd + SLOT * (XT / 2) + (q - RB) * 1024);
    }
    unsigned char* W = smem + SLOT * W_STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
      if (a.dbg & 2) continue;
      if (k0 + w_cs[i] < a.K) glds16k(w_src[i] + k0, W + i * (KTHREADS * 16));
      else {
        *reinterpret_cast<u32x4*>(smem + SLOT * W_STAGE + (i * KTHREADS + tid) * 16) = u32x4{0u, 0u, 0u, 0u};
        glds16k(a.X + lane * 8, smem + KD * W_STAGE + 2 * BN * 4 + wave * 1024);     // (keeps vmcnt uniform; lands in a dump)
      }
    }
  };

  f32x4 acc[RB][NT];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 csum[STATS ? NT : 1], csq[STATS ? NT : 1];
#pragma unroll
  for (int j = 0; j < (STATS ? NT : 1); ++j) csum[j] = csq[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool plain = !a.out_scale && !a.out_shift && a.out_act == PB_ACT_NONE;
  const int pcol = (fk & 1) * 16 + (fk >> 1) * 8;
  int osc_n0 = -1;

  auto flush_stats = [&](int n0) __attribute__((always_inline)) {
    if constexpr (STATS) {
      float* sred = (float*)(smem + KD * W_STAGE + 2 * BN * 4 + 8192 + XBYTES);      // [8][BN][2]
#pragma unroll
      for (int j = 0; j < NT; ++j) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float s = csum[j][r], qv = csq[j][r];
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) {
            s += __shfl_xor(s, o, 64);
            qv += __shfl_xor(qv, o, 64);
          }
          if (frow == 0) {
            sred[(wave * BN + j * 16 + fk * 4 + r) * 2 + 0] = s;
            sred[(wave * BN + j * 16 + fk * 4 + r) * 2 + 1] = qv;
          }
        }
        csum[j] = csq[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      lds_barrier();
      float* st = a.stats + (size_t)blockIdx.x * 2 * a.N;
      for (int c = tid; c < BN; c += KTHREADS) {
        const int n = n0 + c;
        if (n < a.N) {
          float s = 0.f, qv = 0.f;
#pragma unroll
          for (int w = 0; w < 8; ++w) {
            s += sred[(w * BN + c) * 2 + 0];
            qv += sred[(w * BN + c) * 2 + 1];
          }
          st[n] += s;
          st[a.N + n] += qv;
        }
      }
      lds_barrier();
    }
  };

  auto epilogue = [&](auto plain_tag, int64_t m0, int64_t mend, int n0) __attribute__((always_inline)) {
    constexpr bool PLAIN = decltype(plain_tag)::value;
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int64_t m = m0 + wave * (16 * RB) + i * 16 + frow;
      const bool m_ok = m < mend;
      if constexpr (!PLAIN) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const f32x4 sc = *reinterpret_cast<const f32x4*>(osc + j * 16 + fk * 4);
          const f32x4 sh = *reinterpret_cast<const f32x4*>(osc + BN + j * 16 + fk * 4);
          acc[i][j] = acc[i][j] * sc + sh;
        }
#define PB_ACT_CASE(A)                                                         \
  case A:                                                                      \
    _Pragma("unroll") for (int j = 0; j < NT; ++j)                             \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) acc[i][j][r] = act_k<A>(acc[i][j][r], a.out_slope); \
    break;
        switch (a.out_act) {
          PB_ACT_CASE(PB_ACT_RELU) PB_ACT_CASE(PB_ACT_LEAKY_RELU) PB_ACT_CASE(PB_ACT_HSWISH) PB_ACT_CASE(PB_ACT_HSIGMOID)
          PB_ACT_CASE(PB_ACT_SILU)
          default: break;
        }
#undef PB_ACT_CASE
      }
      u32x2 pk[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if constexpr (STATS) {
          if (m_ok) {                                 // (fp32 values, before the bf16 rounding of the store)
            csum[j] += acc[i][j];
            csq[j] += acc[i][j] * acc[i][j];
          }
        }
        const bf16x4 o = __builtin_convertvector(acc[i][j], bf16x4);
        pk[j] = __builtin_bit_cast(u32x2, o);
      }
      T* yrow = a.Y + m * a.N + n0;
      const T* rrow = a.R ? a.R + m * a.N + n0 : nullptr;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        // rows of 16 lanes, (D, S) = (block 2p, block 2p+1);  swap: D <- [D0 S0 D2 S2], S <- [D1 S1 D3 S3]
        const auto lo = __builtin_amdgcn_permlane16_swap(pk[2 * p][0], pk[2 * p + 1][0], false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap(pk[2 * p][1], pk[2 * p + 1][1], false, false);
        u32x4 o = u32x4{lo[0], hi[0], lo[1], hi[1]};
        const int c = p * 32 + pcol;
        if (m_ok && n0 + c < a.N && !(a.dbg & 4)) {
          if (rrow) {
            const bf16x8 rv = *reinterpret_cast<const bf16x8*>(rrow + c);
            bf16x8 ov = __builtin_bit_cast(bf16x8, o);
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = (bf16_t)((float)ov[e] + (float)rv[e]);
            o = __builtin_bit_cast(u32x4, ov);
          }
          *reinterpret_cast<u32x4*>(yrow + c) = o;
        }
      }
      if (NT & 1) {
        const int c = (NT - 1) * 16 + fk * 4;
        u32x2 o = pk[NT - 1];
        if (m_ok && n0 + c < a.N) {
          if (rrow) {
            const bf16x4 rv = *reinterpret_cast<const bf16x4*>(rrow + c);
            bf16x4 ov = __builtin_bit_cast(bf16x4, o);
#pragma unroll
            for (int e = 0; e < 4; ++e) ov[e] = (bf16_t)((float)ov[e] + (float)rv[e]);
            o = __builtin_bit_cast(u32x2, ov);
          }
          *reinterpret_cast<u32x2*>(yrow + c) = o;
        }
      }
    }
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
  Cur ci = ct;
  int ksi = 0, ks = 0;
  set_rows(ci);
  auto advance_issue = [&]() __attribute__((always_inline)) {
    if (++ksi == nk) {
      ksi = 0;
      cur_step(ci);
      if (ci.t < total) set_rows(ci);
    }
  };
  // one stage of the pipeline, register/LDS slot SLOT: the stage multiplied now was requested KD-1 stages ago
  auto step = [&](auto slot_tag) __attribute__((always_inline)) -> bool {
    constexpr int SLOT = decltype(slot_tag)::value;
    constexpr int NEXT = (SLOT + KD - 1) % KD;
    // KD-2 younger stages stay in flight (stores of an epilogue in between only make the wait longer: vmcnt retires in
    // order).  Then everybody's part of the weight slice is in, and everybody is done reading the slice refilled next.
    wait_vmk<(KD - 2) * VPS>();
    // (LDS operations of a wave execute in order: with HYB the one loaded half-tile is rewritten behind the reads of the
    // previous stage; otherwise two tiles alternate)
    unsigned char* xt = HYB ? xs : xs + (SLOT & 1) * XT;
#pragma unroll
    for (int q = 0; q < NLD; ++q) *reinterpret_cast<bf16x8*>(xt + xw_off[q]) = xa[SLOT][q];
    const unsigned char* xrow[RB];                   // this lane's row of each 16-row block
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int row = i * 16 + frow;
      if constexpr (HYB) xrow[i] = row < 8 * RB ? xt + row * 128 : xd + SLOT * (XT / 2) + (row - 8 * RB) * 128;
      else xrow[i] = xt + row * 128;
    }
    lds_barrier();
    if (ci.t < total) {
      issue(std::integral_constant<int, NEXT>{}, ksi);
      advance_issue();
    } else {
      // nothing left to request: dummies keep the in-flight count uniform
#pragma unroll
      for (int i = 0; i < VPS; ++i) glds16k(a.X + lane * 8, smem + KD * W_STAGE + 2 * BN * 4 + wave * 1024);
    }
    const unsigned char* Ws = smem + SLOT * W_STAGE;
    if (!(a.dbg & 8))
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int cw = (((h * 4) + fk) ^ w_sw) << 4;
      bf16x8 xf[RB];
#pragma unroll
      for (int i = 0; i < RB; ++i) xf[i] = *reinterpret_cast<const bf16x8*>(xrow[i] + cw);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const bf16x8 bw = *reinterpret_cast<const bf16x8*>(Ws + (j * 16 + frow) * 128 + cw);
#pragma unroll
        for (int i = 0; i < RB; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw, xf[i], acc[i][j], 0, 0, 0);
      }
    }
    if (++ks == nk) {
      ks = 0;
      int64_t m0, mend;
      int grp;
      panel(ct.g * 8 + r8, m0, mend, grp);
      const int n0 = ct.nt * BN;
      if (!plain && n0 != osc_n0) {                   // (the staging of scale / shift needs the workgroup together)
        osc_n0 = n0;
        lds_barrier();
        for (int c = tid; c < BN; c += KTHREADS) {
          const int n = min(n0 + c, a.N - 1);
          osc[c] = a.out_scale ? a.out_scale[n] : 1.f;
          osc[BN + c] = a.out_shift ? a.out_shift[n] : 0.f;
        }
        lds_barrier();
      }
      if (plain) epilogue(std::true_type{}, m0, mend, n0);
      else epilogue(std::false_type{}, m0, mend, n0);
#pragma unroll
      for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int nt_done = ct.nt;
      cur_step(ct);
      if (STATS && n_tiles > 1 && (ct.t >= total || ct.nt != nt_done)) flush_stats(n0);
      if (ct.t >= total) return false;
    }
    return true;
  };

  // prologue: KD-1 stages requested
  {
    bool more = true;
    if (more && ci.t < total) { issue(std::integral_constant<int, 0>{}, ksi); advance_issue(); } else more = false;
    if (more && ci.t < total) { issue(std::integral_constant<int, 1>{}, ksi); advance_issue(); } else more = false;
    if (more && ci.t < total) { issue(std::integral_constant<int, 2>{}, ksi); advance_issue(); } else more = false;
  }
  static_assert(KD == 4, "the prologue and the unrolled ring below are written for four slots");
  for (;;) {
    if (!step(std::integral_constant<int, 0>{})) break;
    if (!step(std::integral_constant<int, 1>{})) break;
    if (!step(std::integral_constant<int, 2>{})) break;
    if (!step(std::integral_constant<int, 3>{})) break;
  }
  wait_vmk<0>();                                     // no DMA may land in LDS after the workgroup has given it back
  if (STATS && n_tiles == 1) flush_stats(0);
}

template <int NT, int RB, bool STATS, bool HYB>
int launch_kloop(KloopArgs a, int64_t total, int device, hipStream_t st) {
  constexpr int BN = NT * 16;
  constexpr int WCH = (BN * 8 + KTHREADS - 1) / KTHREADS;
  constexpr int XT = 16 * RB * 128;
  constexpr size_t lds = (size_t)KD * WCH * KTHREADS * 16 + 2 * BN * 4 + 8192 + (HYB ? 8 * (XT / 2) * (1 + KD) : 8 * 2 * XT) + (STATS ? 64 * BN : 0);
  static int cus_dev[16] = {0};
  int& cus = cus_dev[device & 15];
  if (cus == 0) {
    if (hipFuncSetAttribute((const void*)pw_kloop_kernel<NT, RB, STATS, HYB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      pb_set_error("pw_kloop: cannot reserve %d bytes of LDS", (int)lds);
      return 1;
    }
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1) cus = 256;
  }
  int64_t blocks = cus;                              // 8 waves of up to 256 registers: one workgroup per CU
  if (blocks > total) blocks = total;
  const int64_t unit = 8 * (int64_t)a.n_tiles;
  if (a.n_tiles > 1 && blocks >= 2 * unit) blocks = blocks / unit * unit;
  else blocks = pb_cdiv(blocks, 8) * 8;
  if (a.stats && blocks > PB_STAT_SLOTS) blocks = PB_STAT_SLOTS;
  const int64_t gq = blocks / 8;
  a.step_g = (int)(gq / a.n_tiles);
  a.step_nt = (int)(gq % a.n_tiles);
  static const int dbg = getenv("PB_GEMM_DMA_DBG") ? atoi(getenv("PB_GEMM_DMA_DBG")) : 0;
  a.dbg = dbg;
  hipLaunchKernelGGL((pw_kloop_kernel<NT, RB, STATS, HYB>), dim3((unsigned)blocks), dim3(KTHREADS), lds, st, a);
  return 0;
}

// rows per wave: two blocks of 16 share every weight fragment read; with statistics (8 more registers per column block)
// the wide tiles keep one block
constexpr int rb_of(int nt, bool stats) { return stats ? (nt <= 4 ? 2 : 1) : (nt <= 7 ? 2 : 1); }

template <bool STATS>
int dispatch_kloop(KloopArgs a, int nt, int device, hipStream_t st) {
  const int rb = rb_of(nt, STATS);
  const int kbm = 8 * 16 * rb;
  if (a.rows_per_group > 0) {
    a.panels_per_group = (int)pb_cdiv(a.rows_per_group, kbm);
    a.m_tiles = a.panels_per_group * (int)(a.M / a.rows_per_group);
  } else {
    a.m_tiles = (int)pb_cdiv(a.M, kbm);
  }
  const int64_t total = a.n_tiles == 1 ? a.m_tiles : pb_cdiv(a.m_tiles, 8) * 8 * a.n_tiles;
  if (total > 0x3fffffffLL) return -1;
  static const int hyb = getenv("PB_KLOOP_HYB") ? atoi(getenv("PB_KLOOP_HYB")) : 0;
#define PB_KL(NT_) (hyb ? launch_kloop<NT_, rb_of(NT_, STATS), STATS, true>(a, total, device, st) : launch_kloop<NT_, rb_of(NT_, STATS), STATS, false>(a, total, device, st))
  switch (nt) {
    case 2: return PB_KL(2);
    case 3: return PB_KL(3);
    case 4: return PB_KL(4);
    case 5: return PB_KL(5);
    case 6: return PB_KL(6);
    case 7: if constexpr (!STATS) return PB_KL(7); else return -1;
    case 8: if constexpr (!STATS) return PB_KL(8); else return -1;
    case 10: if constexpr (!STATS) return launch_kloop<10, 1, STATS, false>(a, total, device, st); else return -1;
  }
#undef PB_KL
  return -1;
}

}  // namespace

// -> 0 launched, > 0 error, -1 shape not taken (the caller falls back to the LDS-DMA ring kernel)
__attribute__((visibility("hidden"))) int pb_pw_kloop_fwd(const void* x, const void* wb, void* y, int64_t M, int K, int N,
                                                          int groups, int64_t rows_per_group, const float* out_scale,
                                                          const float* out_shift, pb_act out_act, float out_slope, float* stats,
                                                          const void* residual, int device, hipStream_t st) {
  static const int on = getenv("PB_GEMM_KLOOP") ? atoi(getenv("PB_GEMM_KLOOP")) : 1;
  if (!on || K % 8 || N % 8 || K < 64 || M < 512) return -1;
  // column tile: padded MFMA width + 16 per tile (every further tile re-reads the X panel through L2)
  const int cand[8] = {2, 3, 4, 5, 6, 7, 8, 10};
  int best = 0;
  int64_t best_cost = INT64_MAX;
  for (int c : cand) {
    if (stats && c > 6) continue;                      // (the statistics take 8 registers per column block)
    const int64_t tiles = pb_cdiv(N, c * 16);
    const int64_t cost = tiles * (c * 16 + 16);
    if (cost < best_cost) { best = c; best_cost = cost; }
  }
  KloopArgs a = {};
  a.X = (const bf16_t*)x; a.WB = (const bf16_t*)wb; a.Y = (bf16_t*)y; a.R = (const bf16_t*)residual;
  a.M = M; a.K = K; a.N = N;
  a.out_scale = out_scale; a.out_shift = out_shift; a.out_act = (int)out_act; a.out_slope = out_slope; a.stats = stats;
  if (groups > 1) {
    a.w_group_stride = (int64_t)N * K;
    a.rows_per_group = rows_per_group;
  }
  a.n_tiles = (int)pb_cdiv(N, best * 16);
  return stats ? dispatch_kloop<true>(a, best, device, st) : dispatch_kloop<false>(a, best, device, st);
}
