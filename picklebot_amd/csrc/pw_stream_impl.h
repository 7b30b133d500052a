// Weight-stationary streaming GEMM for the pointwise layers with a short reduction (K <= 160):
//     Y[M,N] = X[M,K] * Wb[N,K]^T   (bf16, fp32 accumulate)
//
// pw_gemm_dma.hip walks (row panel, column tile, K stage) tiles; with K <= 160 a tile is two or three stages and the
// per-tile work around the MFMAs (tile bookkeeping, DMA addressing of two operands, the staged epilogue) is what the
// waves spend their issue slots on - a wave issues at most one instruction every four cycles, whatever its kind:
// measured with all memory traffic switched off, the expansion layers still took 75 % of their time.
//
// Here every wave keeps ONE weight slice (BN x K) in registers for the whole launch as the MFMA's A operand, and the only
// stream is X: a stage of the LDS ring is SBM complete rows = one contiguous block of memory, fetched by LDS-DMA NS-1
// stages ahead.  The 8 waves of a workgroup are CG column groups x RG row groups: wave (cg, rg) multiplies rows
// rg*16*RB .. +16*RB of the stage (RB blocks of 16) by column tile ntw*CG + cg, so one DMA of the rows serves CG column
// tiles (the LDS-DMA path moves ~16 B/clk per CU: re-fetching the rows once per column tile is what bounds a one-tile-
// per-workgroup layout on the 6x expansions).  Per row block: K/32 LDS reads, NT*K/32 MFMAs, convert, exchange half of
// the packed results with the neighbouring 16-lane row (v_permlane16_swap) so that every lane owns 16 contiguous bytes of
// an output row, store - no staging tile, no second barrier.  Workgroups of the same rows run on the same XCD
// (blockIdx mod 8): X comes from HBM once, further column groups re-read it through that XCD's L2.
//
// LDS image of a stage: row r at r*K*2 bytes; for an even number of 16-byte chunks per row the chunks of row r are
// rotated by r mod 16 (on the DMA's source side), which makes the 16-row fragment reads conflict-free; an odd chunk count
// already is.  k-steps past K read a (valid, finite) chunk of the same row against zero weights.
#pragma once
#include "pb_common.h"
#include <stdlib.h>
#include <type_traits>

// -DPB_STREAM_PROBE: cycles per phase of one wave (block 9, wave 1) printed at the end of every launch (experiments only)
#ifdef PB_STREAM_PROBE
#define PWS_PROBE_DECL long long pws_t[5] = {0, 0, 0, 0, 0}; long long pws_last = __builtin_amdgcn_s_memtime();
#define PWS_TICK(i) do { const long long tn = __builtin_amdgcn_s_memtime(); pws_t[i] += tn - pws_last; pws_last = tn; } while (0)
#else
#define PWS_PROBE_DECL
#define PWS_TICK(i) do { } while (0)
#endif

namespace {

constexpr int STHREADS = 512;

struct StreamArgs {
  const bf16_t* X;
  const bf16_t* WB;          // [N][K]
  bf16_t* Y;
  const bf16_t* R;           // residual [M][N] or null
  int64_t M;
  int K, N;
  const float* out_scale;
  const float* out_shift;
  int out_act;
  float out_slope;
  float* stats;
  int stat_fold;             // > 1: columns n and n + N/stat_fold (+ ...) are replicas of ONE channel (the 4-rows-per-line form of
                             // K = 16 layers): their sums fold into stats [slots][2][N/stat_fold]; all columns in one workgroup
  int n_tiles, m_tiles;      // column tiles of BN, row panels of SBM
  int wgc;                   // workgroup columns = ceil(n_tiles / CG)
  int q;                     // grid = 8 * wgc * q: a workgroup's row panels are (g0 + i*q)*8 + (blockIdx & 7)
};

__device__ __forceinline__ void glds16s(const void* gsrc, unsigned char* lds_dst) {
  typedef __attribute__((address_space(3))) unsigned char* lds_b;
  const unsigned dst = (unsigned)(uintptr_t)(lds_b)lds_dst;
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int ACT> __device__ __forceinline__ float act_t(float x, float slope) {
  if constexpr (ACT == PB_ACT_RELU) return x > 0.f ? x : 0.f;
  else if constexpr (ACT == PB_ACT_LEAKY_RELU) return x > 0.f ? x : x * slope;
  else if constexpr (ACT == PB_ACT_HSWISH) return x * fminf(fmaxf(x + 3.f, 0.f), 6.f) * (1.f / 6.f);
  else if constexpr (ACT == PB_ACT_HSIGMOID) return fminf(fmaxf(x * (1.f / 6.f) + 0.5f, 0.f), 1.f);
  else if constexpr (ACT == PB_ACT_SILU) return x / (1.f + __expf(-x));
  else return x;
}

constexpr int ring_depth(int stage_max_bytes) {
  int ns = (144 * 1024) / stage_max_bytes;
  return ns > 8 ? 8 : ns;
}

template <int NT, int KS, int CG, int RB, bool STATS>
__global__ __launch_bounds__(STHREADS) void pw_stream_kernel(StreamArgs a) {
  typedef bf16_t T;
  constexpr int RG = 8 / CG;
  constexpr int SBM = 16 * RB * RG;                  // rows per stage
  constexpr int DPS = KS * SBM / 128;                // DMA instructions per thread and stage (8 KiB per round of 512 threads)
  constexpr int NS = ring_depth(SBM * KS * 64);
  constexpr int BN = NT * 16;
  constexpr int NP = NT / 2;                         // column-block pairs stored as 16 bytes per lane
  static_assert(NS >= 2 && DPS >= 1, "ring too shallow");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fk = lane >> 4;        // MFMA operand lanes: row, 8-element k chunk
  const int rg = wave % RG, cg = wave / RG;
  const int CH = a.K >> 3;                           // 16-byte chunks per row
  const int stage_bytes = SBM * CH * 16;
  unsigned char* dump = smem + NS * stage_bytes;     // 8 KiB: where the surplus DMA instructions of a stage land
  float* osc = (float*)(dump + 8192);                // [CG][2][BN] scale / shift (non-plain epilogues only)
  const bool rot = (CH & 1) == 0;

  PWS_PROBE_DECL
  // ---- this workgroup's tiles
  const int r8 = (int)(blockIdx.x & 7), u = (int)(blockIdx.x >> 3);
  const int g0 = u / a.wgc, ntw = u - g0 * a.wgc;
  const int nt = ntw * CG + cg;                      // this wave's column tile
  const bool active = nt < a.n_tiles;                // (wave-uniform; idle waves still move data and keep the barriers)
  const int n0 = nt * BN;
  const int G_r = a.m_tiles > r8 ? (a.m_tiles - r8 + 7) >> 3 : 0;          // row-panel groups that exist for this r8
  const int n_my = g0 < G_r ? (G_r - g0 + a.q - 1) / a.q : 0;
  if (n_my == 0) return;

  // ---- DMA: instruction q of a stage moves chunks q*512 + tid; constant per thread, the source advances by a fixed
  // stride from one of this workgroup's panels to the next
  const T* src[DPS];
  int drow[DPS];
  bool dvalid[DPS];
  const int64_t panel_stride = (int64_t)a.q * 8 * SBM * a.K;               // elements
  {
    const int64_t m0 = (int64_t)(g0 * 8 + r8) * SBM;
#pragma unroll
    for (int q = 0; q < DPS; ++q) {
      const int f = q * STHREADS + tid;
      dvalid[q] = (q * STHREADS + wave * 64) < SBM * CH;                   // wave-uniform (SBM*CH is a multiple of 64)
      const int row = min(f / CH, SBM - 1), c = f - (f / CH) * CH;
      int s = c;
      if (rot) {
        s = c + (row & 15) % CH;
        if (s >= CH) s -= CH;
      }
      drow[q] = row;
      src[q] = a.X + (m0 + row) * a.K + s * 8;
    }
  }
  // ---- fragment reads: row frow of a 16-row block, chunk ks*4 + fk (wrapped into the row past K)
  int a_off[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int s = (ks * 4 + fk) % CH;
    const int pos = rot ? (s + CH - frow % CH) % CH : s;
    a_off[ks] = (frow * CH + pos) * 16 + rg * RB * 256 * CH;
  }

  auto issue = [&](int i, int buf) {                 // panel i of this workgroup (or nothing: dummies keep vmcnt uniform)
    const bool real = i < n_my;
    unsigned char* dst = smem + buf * stage_bytes + wave * 1024;
    const int64_t m0 = (int64_t)((g0 + i * a.q) * 8 + r8) * SBM;
    const bool last = real && m0 + SBM > a.M;
#pragma unroll
    for (int q = 0; q < DPS; ++q) {
      if (real && dvalid[q]) {
        const T* p = src[q];
        if (last) {                                   // rows past M: re-read the last row (never stored)
          const int rmax = (int)(a.M - 1 - m0);
          if (drow[q] > rmax) p -= (int64_t)(drow[q] - rmax) * a.K;
        }
        glds16s(p, dst + q * (STHREADS * 16));
      } else {
        glds16s(a.X + lane * 8, dump + wave * 1024);
      }
      src[q] += panel_stride;
    }
  };

  // The first stage's DMA goes out BEFORE the weight slice is fetched: the two round trips used to follow each other (the
  // compiler waits for the weight loads with vmcnt(0) ahead of everything that follows), ~2500 of a workgroup's ~8000 prologue
  // cycles on layers that give a workgroup only 2-4 panels (PB_STREAM_PROBE).  The deeper stages follow once the weights are in.
  issue(0, 0);
  // ---- the weight slice, in registers for the whole launch (rows past N repeat row N-1: never stored)
  bf16x8 wf[NT][KS];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = min(n0 + j * 16 + frow, a.N - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int k = ks * 32 + fk * 8;
      bf16x8 z;
#pragma unroll
      for (int e = 0; e < 8; ++e) z[e] = (bf16_t)0.f;
      wf[j][ks] = (k < a.K && active) ? *reinterpret_cast<const bf16x8*>(a.WB + (int64_t)n * a.K + k) : z;
    }
  }

#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(wf[j][ks]));      // (consumed HERE: a wait for them at their first use inside the
  // panel loop is an s_waitcnt vmcnt(0) in that loop, behind the prefetched stages and the stores - measured 3-23 % per panel)
  // ---- epilogue constants
  const bool plain = !a.out_scale && !a.out_shift && a.out_act == PB_ACT_NONE;
  if (!plain) {
    for (int c = tid; c < CG * BN; c += STHREADS) {
      const int g = c / BN, cc = c - g * BN;
      const int n = min((ntw * CG + g) * BN + cc, a.N - 1);
      osc[g * 2 * BN + cc] = a.out_scale ? a.out_scale[n] : 1.f;
      osc[g * 2 * BN + BN + cc] = a.out_shift ? a.out_shift[n] : 0.f;
    }
  }
  const float* my_osc = osc + cg * 2 * BN;
  // after the exchange a lane owns 8 consecutive columns: block 2p + (fk & 1) of pair p, half fk >> 1
  const int pcol = (fk & 1) * 16 + (fk >> 1) * 8;
  // store instructions per stage that touch at least one valid column: a lower bound on what is in flight behind a DMA
  int n_st = 0;
#pragma unroll
  for (int p = 0; p < NP; ++p) n_st += n0 + p * 32 < a.N;
  if (NT & 1) n_st += n0 + (NT - 1) * 16 < a.N;
  n_st = (a.R || !active) ? 0 : n_st * RB;   // (residual loads are waited for with vmcnt(0) by the compiler)
  int allow = (NS - 2) * DPS + (NS - 2) * n_st;
  allow = allow > 63 ? 63 : allow;

  f32x4 csum[STATS ? NT : 1], csq[STATS ? NT : 1];   // (8*NT registers: the widest tiles exist without statistics only)
#pragma unroll
  for (int j = 0; j < (STATS ? NT : 1); ++j) csum[j] = csq[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // one 16-row block: multiply, convert, exchange, store
  auto block = [&](auto plain_tag, const unsigned char* Ab, int64_t m, bool full, T* yrow, const T* rrow) {
    constexpr bool PLAIN = decltype(plain_tag)::value;
    bf16x8 af[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) af[ks] = *reinterpret_cast<const bf16x8*>(Ab + a_off[ks]);
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][0], af[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
    for (int ks = 1; ks < KS; ++ks)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][ks], af[ks], acc[j], 0, 0, 0);
    // lane (frow, fk) holds columns n0 + j*16 + fk*4 + (0..3) of row m
    const bool m_ok = m < a.M;
    if constexpr (!PLAIN) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(my_osc + j * 16 + fk * 4);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(my_osc + BN + j * 16 + fk * 4);
        acc[j] = acc[j] * sc + sh;
      }
#define PB_ACT_CASE(A)                                                         \
  case A:                                                                      \
    _Pragma("unroll") for (int j = 0; j < NT; ++j)                             \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) acc[j][r] = act_t<A>(acc[j][r], a.out_slope); \
    break;
      switch (a.out_act) {
        PB_ACT_CASE(PB_ACT_RELU) PB_ACT_CASE(PB_ACT_LEAKY_RELU) PB_ACT_CASE(PB_ACT_HSWISH) PB_ACT_CASE(PB_ACT_HSIGMOID)
        PB_ACT_CASE(PB_ACT_SILU)
        default: break;
      }
#undef PB_ACT_CASE
    }
    if constexpr (STATS) {
      // (fp32 values, before the bf16 rounding of the store.)  Only the last block of the matrix can hold rows past M: the
      // test for it is wave-uniform - per-lane selects of 8*NT sums were two thirds of this kernel's statistics cost
      if (full) {
        asm volatile("");                             // (keeps the branch: hipcc otherwise if-converts both arms back into selects)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          csum[j] += acc[j];
          csq[j] += acc[j] * acc[j];
        }
      } else if (m_ok) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          csum[j] += acc[j];
          csq[j] += acc[j] * acc[j];
        }
      }
    }
    u32x2 pk[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const bf16x4 o = __builtin_convertvector(acc[j], bf16x4);
      pk[j] = __builtin_bit_cast(u32x2, o);
    }
    const bool st_ok = m_ok;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      // rows of 16 lanes, (D, S) = (block 2p, block 2p+1);  swap: D <- [D0 S0 D2 S2], S <- [D1 S1 D3 S3]
      const auto lo = __builtin_amdgcn_permlane16_swap(pk[2 * p][0], pk[2 * p + 1][0], false, false);
      const auto hi = __builtin_amdgcn_permlane16_swap(pk[2 * p][1], pk[2 * p + 1][1], false, false);
      u32x4 o = u32x4{lo[0], hi[0], lo[1], hi[1]};
      const int c = p * 32 + pcol;
      if (st_ok && n0 + c < a.N) {
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
      if (st_ok && n0 + c < a.N) {
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
  };

  int it = 1, ibuf = 1;                              // (stage 0 was issued ahead of the weight loads)
#pragma unroll 1
  for (; it < NS - 1; ++it) {
    issue(it, ibuf);
    if (++ibuf == NS) ibuf = 0;
  }
  PWS_TICK(0);
  int cbuf = 0;
  const int64_t row_step = (int64_t)16 * a.N;        // elements between row blocks
#pragma unroll 1
  for (int ct = 0; ct < n_my; ++ct) {
    // panel ct must have landed; the NS-2 younger stages and, in order behind and between them, the stores of the last
    // panels stay in flight.  vmcnt retires in order for global_* instructions, so allowing fewer operations than were
    // issued behind the panel only waits longer (and what it waits for first are the oldest stores).
    if (NS > 2 && ct >= NS - 1) {
      // (the count is an immediate: a few steps cover it - rounding the allowance down is always safe)
      if (allow >= 63) wait_vm<63>();
      else if (allow >= 48) wait_vm<48>();
      else if (allow >= 32) wait_vm<32>();
      else if (allow >= 24) wait_vm<24>();
      else if (allow >= 16) wait_vm<16>();
      else if (allow >= 12) wait_vm<12>();
      else if (allow >= 8) wait_vm<8>();
      else wait_vm<((NS - 2) * DPS < 8 ? (NS - 2) * DPS : 8)>();
    } else if (NS > 2) {
      wait_vm<((NS - 2) * DPS > 63 ? 63 : (NS - 2) * DPS)>();   // (the first NS-1 panels: no stores between their stages yet)
    } else {
      wait_vm<0>();
    }
    lds_barrier();                                   // everybody's chunks are in; everybody is done with the stage refilled next
    PWS_TICK(1);
    issue(it, ibuf);
    PWS_TICK(2);
    ++it;
    if (++ibuf == NS) ibuf = 0;
    const unsigned char* As = smem + cbuf * stage_bytes;
    if (++cbuf == NS) cbuf = 0;
    if (active) {
      const int64_t mb = (int64_t)((g0 + ct * a.q) * 8 + r8) * SBM + rg * (16 * RB);      // (wave-uniform)
      const int64_t m = mb + frow;
      T* yrow = a.Y + m * a.N + n0;
      const T* rrow = a.R ? a.R + m * a.N + n0 : nullptr;
      // (rolled: an unrolled loop hoists every block's fragment loads and spills the weight slice)
      if (plain) {
#pragma unroll 1
        for (int rb = 0; rb < RB; ++rb) {
          block(std::true_type{}, As + rb * 256 * CH, m + rb * 16, mb + rb * 16 + 16 <= a.M, yrow, rrow);
          yrow += row_step;
          if (rrow) rrow += row_step;
        }
      } else {
#pragma unroll 1
        for (int rb = 0; rb < RB; ++rb) {
          block(std::false_type{}, As + rb * 256 * CH, m + rb * 16, mb + rb * 16 + 16 <= a.M, yrow, rrow);
          yrow += row_step;
          if (rrow) rrow += row_step;
        }
      }
    }
    PWS_TICK(3);
  }
  PWS_TICK(4);
#ifdef PB_STREAM_PROBE
  if (blockIdx.x == 9 && tid == 64)
    printf("pw_stream<NT %d KS %d CG %d RB %d> M %lld K %d N %d grid %d: %d panels; cycles: prologue %lld, per panel: wait+barrier %lld, issue %lld, compute %lld; tail %lld\n",
           NT, KS, CG, RB, (long long)a.M, a.K, a.N, (int)gridDim.x, n_my, pws_t[0], pws_t[1] / n_my, pws_t[2] / n_my, pws_t[3] / n_my, pws_t[4]);
#endif
  wait_vm<0>();                                      // no DMA may land in LDS after the workgroup has given it back
  if constexpr (STATS) {
    // column sums: over the 16 rows of a block (lanes with equal fk), then over the RG row groups through LDS
    lds_barrier();
    float* sred = (float*)smem;                      // [8 waves][BN][2]
    if (active) {
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
      }
    }
    lds_barrier();
    if (a.stat_fold > 1) {
      const int Nr = a.N / a.stat_fold;
      float* st = a.stats + (size_t)blockIdx.x * 2 * Nr;
      for (int c = tid; c < Nr; c += STHREADS) {
        float s = 0.f, qv = 0.f;
        for (int f = 0; f < a.stat_fold; ++f) {
          const int n = c + f * Nr, g = n / BN, cc = n - g * BN;            // (wgc == 1: column tile = column group)
#pragma unroll
          for (int w = 0; w < RG; ++w) {
            s += sred[((g * RG + w) * BN + cc) * 2 + 0];
            qv += sred[((g * RG + w) * BN + cc) * 2 + 1];
          }
        }
        st[c] += s;
        st[Nr + c] += qv;
      }
      return;
    }
    float* st = a.stats + (size_t)blockIdx.x * 2 * a.N;      // this workgroup's own slot (grid <= PB_STAT_SLOTS)
    for (int c = tid; c < CG * BN; c += STHREADS) {
      const int g = c / BN, cc = c - g * BN;
      const int n = (ntw * CG + g) * BN + cc;
      if ((ntw * CG + g) < a.n_tiles && n < a.N) {
        float s = 0.f, qv = 0.f;
#pragma unroll
        for (int w = 0; w < RG; ++w) {               // waves g*RG .. g*RG + RG-1 hold column group g
          s += sred[((g * RG + w) * BN + cc) * 2 + 0];
          qv += sred[((g * RG + w) * BN + cc) * 2 + 1];
        }
        st[n] += s;
        st[a.N + n] += qv;
      }
    }
  }
}

template <int NT, int KS, int CG, int RB, bool STATS>
int launch_stream(StreamArgs a, int device, hipStream_t st) {
  constexpr int SBM = 16 * RB * (8 / CG);
  constexpr int NS = ring_depth(SBM * KS * 64);
  const size_t lds = (size_t)NS * SBM * (a.K / 8) * 16 + 8192 + (size_t)CG * 2 * NT * 16 * 4;
  static int cus_dev[16] = {0};
  int& cus = cus_dev[device & 15];
  if (cus == 0) {
    if (hipFuncSetAttribute((const void*)pw_stream_kernel<NT, KS, CG, RB, STATS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      pb_set_error("pw_stream: cannot reserve the LDS ring");
      return 1;
    }
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1) cus = 256;
  }
  a.m_tiles = (int)pb_cdiv(a.M, SBM);
  a.wgc = (int)pb_cdiv(a.n_tiles, CG);
  const int unit = 8 * a.wgc;
  int q = cus / unit;
  const int groups = (a.m_tiles + 7) / 8;
  if (q > groups) q = groups;
  if (q < 1) q = 1;
  a.q = q;
  hipLaunchKernelGGL((pw_stream_kernel<NT, KS, CG, RB, STATS>), dim3((unsigned)(unit * q)), dim3(STHREADS), lds, st, a);
  return 0;
}

// column-tile widths per K-step count: the register file (256 per wave at two waves per SIMD) holds the weight slice
// (NT*KS*4 VGPRs) next to the accumulators (4*NT), the row fragments and, with statistics, 8*NT sums
constexpr int kMaxNT[2][6] = {{0, 7, 7, 7, 6, 5}, {0, 7, 6, 6, 5, 4}};

template <int KS, int CG, int RB, bool STATS>
int dispatch_nt(const StreamArgs& a, int nt, int device, hipStream_t st) {
  constexpr int MAXNT = kMaxNT[STATS][KS];
  switch (nt) {
    case 1: if constexpr (CG == 1) return launch_stream<1, KS, CG, RB, STATS>(a, device, st); else return -1;
    case 2: if constexpr (CG == 1) return launch_stream<2, KS, CG, RB, STATS>(a, device, st); else return -1;
    case 3: if constexpr (CG == 1) return launch_stream<3, KS, CG, RB, STATS>(a, device, st); else return -1;
    case 4: return launch_stream<4, KS, CG, RB, STATS>(a, device, st);
    case 5: if constexpr (MAXNT >= 5) return launch_stream<5, KS, CG, RB, STATS>(a, device, st); else return -1;
    case 6: if constexpr (MAXNT >= 6) return launch_stream<6, KS, CG, RB, STATS>(a, device, st); else return -1;
    case 7: if constexpr (MAXNT >= 7) return launch_stream<7, KS, CG, RB, STATS>(a, device, st); else return -1;
  }
  return -1;
}

template <int KS, bool STATS>
int dispatch_cg(const StreamArgs& a, int nt, int cg, int device, hipStream_t st) {
  switch (cg) {
    case 1:
      if constexpr (KS <= 2) return dispatch_nt<KS, 1, 2, STATS>(a, nt, device, st);      // 256-row stages while they are small
      else return dispatch_nt<KS, 1, 1, STATS>(a, nt, device, st);
    case 2: return dispatch_nt<KS, 2, 2, STATS>(a, nt, device, st);
    case 4: return dispatch_nt<KS, 4, 4, STATS>(a, nt, device, st);
    default: return dispatch_nt<KS, 8, 8, STATS>(a, nt, device, st);
  }
}

template <bool STATS>
int dispatch_ks(const StreamArgs& a, int ks, int nt, int cg, int device, hipStream_t st) {
  switch (ks) {
    case 1: return dispatch_cg<1, STATS>(a, nt, cg, device, st);
    case 2: return dispatch_cg<2, STATS>(a, nt, cg, device, st);
    case 3: return dispatch_cg<3, STATS>(a, nt, cg, device, st);
    case 4: return dispatch_cg<4, STATS>(a, nt, cg, device, st);
    default: return dispatch_cg<5, STATS>(a, nt, cg, device, st);
  }
}

}  // namespace
