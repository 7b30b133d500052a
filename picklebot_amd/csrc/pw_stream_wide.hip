// Weight-stationary streaming GEMM for a LONG reduction and few output columns (the projection layers and the
// backward-data of the expansions: K up to 1024, N up to a few hundred):
//     Y[M,N] = X[M,K] * Wb[g(m)][N,K]^T   (bf16, fp32 accumulate)
//
// pw_stream_impl.h keeps a BN x K weight slice in every wave's registers, which only fits for K <= 160.  Here the
// COLUMNS are spread over the waves instead: wave w of a workgroup keeps 16 output columns x all of K (K/32 fragments, up
// to 128 registers) and multiplies every row of a stage by them.  A stage of the LDS ring is 16*RB COMPLETE rows = one
// contiguous block of X (43 KB at K = 672), fetched by LDS-DMA NS-1 stages ahead, read once from HBM - the column-slice
// walk of pw_gemm_dma.hip (128 bytes of each of 128 rows per stage, plus a weight slice per stage on the same DMA path)
// sat at 2.0-2.7 TB/s on these layers.  All 8 waves read the whole stage from LDS (K/32 ds_read_b128 per 16 rows each:
// ~16 B/clk of X per CU, above the CU's share of HBM), one MFMA per read, two accumulator chains per row block.
// Output: a lane holds 4 consecutive columns of a row - 8-byte stores (the output is the small tensor here).
//
// Per-clip weights (the squeeze-excite scale folded into the projection weights): a workgroup belongs to one clip and
// walks that clip's row panels, so its weight fragments stay put.
#include "pw_stream_impl.h"

namespace {

struct WideArgs {
  StreamArgs s;              // X, WB, Y, R, M, K, N, epilogue, stats
  int64_t w_group_stride;    // N*K, or 0 when there is one weight matrix
  int64_t rows_per_group;    // rows of a clip (M when there is one weight matrix)
  int groups, parts, wgc;    // grid = groups * parts * wgc: workgroup (g, p, ntw) walks panels p, p+parts, ... of clip g
};

template <int KSMAX, int RB, bool STATS>
__global__ __launch_bounds__(STHREADS) void pw_stream_wide_kernel(WideArgs wa) {
  typedef bf16_t T;
  const StreamArgs& a = wa.s;
  constexpr int SBM = 16 * RB;                       // rows per stage
  constexpr int DPS = KSMAX * SBM / 128;             // DMA instructions per thread and stage (8 KiB per round of 512 threads)
  constexpr int NS = ring_depth(SBM * KSMAX * 64);
  static_assert(NS >= 2 && DPS >= 1 && KSMAX % 8 == 0, "bad configuration");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fk = lane >> 4;        // MFMA operand lanes: row, 8-element k chunk
  const int CH = a.K >> 3;                           // 16-byte chunks per row
  const int nks = (a.K + 31) >> 5;
  const int stage_bytes = SBM * CH * 16;
  unsigned char* dump = smem + NS * stage_bytes;     // 8 KiB: where the surplus DMA instructions of a stage land
  const bool rot = (CH & 1) == 0;

  // ---- this workgroup: clip g, part p, column group ntw; this wave: 16 columns
  const int ntw = (int)(blockIdx.x % wa.wgc);
  const int gp = (int)(blockIdx.x / wa.wgc);
  const int g = gp / wa.parts, p = gp - g * wa.parts;
  const int n0 = (ntw * 8 + wave) * 16;
  const bool active = n0 < a.N;                      // (wave-uniform; idle waves still move data and keep the barriers)
  const int64_t mbeg = (int64_t)g * wa.rows_per_group, mend = mbeg + wa.rows_per_group;
  const int npanels = (int)((wa.rows_per_group + SBM - 1) / SBM);
  const int n_my = p < npanels ? (npanels - p + wa.parts - 1) / wa.parts : 0;
  if (n_my == 0) return;

  // ---- the weight columns, in registers for the whole launch (columns past N repeat column N-1: never stored)
  bf16x8 wf[KSMAX];
  {
    const T* wg = a.WB + (int64_t)g * wa.w_group_stride + (int64_t)min(n0 + frow, a.N - 1) * a.K;
#pragma unroll
    for (int ks = 0; ks < KSMAX; ++ks) {
      const int k = ks * 32 + fk * 8;
      bf16x8 z;
#pragma unroll
      for (int e = 0; e < 8; ++e) z[e] = (bf16_t)0.f;
      wf[ks] = (k < a.K && active) ? *reinterpret_cast<const bf16x8*>(wg + k) : z;
    }
  }

  // ---- DMA: instruction q of a stage moves chunks q*512 + tid; constant per thread, the source advances by a fixed stride
  const T* src[DPS];
  int drow[DPS];
  bool dvalid[DPS], lvalid[DPS];
  const int64_t panel_stride = (int64_t)wa.parts * SBM * a.K;              // elements
#pragma unroll
  for (int q = 0; q < DPS; ++q) {
    const int f = q * STHREADS + tid;
    dvalid[q] = (q * STHREADS + wave * 64) < SBM * CH;                     // wave-uniform: some lane of the wave has a chunk
    lvalid[q] = f < SBM * CH;                                              // (SBM*CH need not be a multiple of 64)
    const int row = min(f / CH, SBM - 1), c = f - (f / CH) * CH;
    int s = c;
    if (rot) {
      s = c + (row & 15) % CH;
      if (s >= CH) s -= CH;
    }
    drow[q] = row;
    src[q] = a.X + (mbeg + (int64_t)p * SBM + row) * a.K + s * 8;
  }
  // fragment reads: row frow of a 16-row block; chunk ks*4 + fk sits at position (chunk - rotation) mod CH of its row
  const int pos0 = rot ? (fk + CH - frow % CH) % CH : fk % CH;
  const int row_base = frow * CH * 16;

  auto issue = [&](int i, int buf) {                 // panel i of this workgroup (or nothing: dummies keep vmcnt uniform)
    const bool real = i < n_my;
    unsigned char* dst = smem + buf * stage_bytes + wave * 1024;
    const int64_t m0 = mbeg + (int64_t)(p + i * wa.parts) * SBM;
    const bool last = real && m0 + SBM > mend;
#pragma unroll
    for (int q = 0; q < DPS; ++q) {
      if (real && dvalid[q]) {
        const T* sp = src[q];
        if (last) {                                   // rows past the clip: re-read its last row (never stored)
          const int rmax = (int)(mend - 1 - m0);
          if (drow[q] > rmax) sp -= (int64_t)(drow[q] - rmax) * a.K;
        }
        if (lvalid[q]) glds16s(sp, dst + q * (STHREADS * 16));
      } else {
        glds16s(a.X + lane * 8, dump + wave * 1024);
      }
      src[q] += panel_stride;
    }
  };

  // ---- epilogue constants: this lane's 4 columns n0 + fk*4 .. +3
  const bool plain = !a.out_scale && !a.out_shift && a.out_act == PB_ACT_NONE;
  f32x4 osc = {1.f, 1.f, 1.f, 1.f}, osh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int n = min(n0 + fk * 4 + r, a.N - 1);
    if (a.out_scale) osc[r] = a.out_scale[n];
    if (a.out_shift) osh[r] = a.out_shift[n];
  }
  // Everything this wave loads into registers is CONSUMED here, ahead of the DMA prologue: a register whose load the compiler
  // still has to wait for at its first use inside the panel loop costs an s_waitcnt vmcnt(0) in that loop - behind the
  // prefetched stages and the stores - and the ring stops being a ring (pw_stream_impl.h: 3-23 % per panel, PB_STREAM_PROBE)
#pragma unroll
  for (int ks = 0; ks < KSMAX; ++ks) asm volatile("" : "+v"(wf[ks]));
  asm volatile("" : "+v"(osc), "+v"(osh));
  const bool col_ok = n0 + fk * 4 < a.N;             // N % 8 == 0: a lane's 4 columns are valid together
  const int n_st = (a.R || !active) ? 0 : RB;        // (residual loads are waited for with vmcnt(0) by the compiler)
  int allow = (NS - 2) * DPS + (NS - 2) * n_st;
  allow = allow > 63 ? 63 : allow;
  f32x4 csum = {0.f, 0.f, 0.f, 0.f}, csq = {0.f, 0.f, 0.f, 0.f};

  // one 16-row block: multiply by this wave's 16 columns over all of K, convert, store
  auto block = [&](const unsigned char* Ab, int64_t m, T* yrow, const T* rrow) __attribute__((always_inline)) {
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    int pos = pos0;
#pragma unroll
    for (int kc = 0; kc < KSMAX; kc += 8) {
      if (kc < nks) {                                 // (uniform)
        bf16x8 af[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          af[i] = *reinterpret_cast<const bf16x8*>(Ab + row_base + pos * 16);     // (past K: a valid chunk of the row, weights 0)
          pos += 4;
          if (pos >= CH) pos -= CH;
        }
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kc + i], af[i], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kc + i + 1], af[i + 1], acc1, 0, 0, 0);
        }
      }
    }
    f32x4 acc = acc0 + acc1;
    const bool m_ok = m < mend;
    if (!plain) {
      acc = acc * osc + osh;
#define PB_ACT_CASE(A)                                                                          \
  case A:                                                                                       \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) acc[r] = act_t<A>(acc[r], a.out_slope);       \
    break;
      switch (a.out_act) {
        PB_ACT_CASE(PB_ACT_RELU) PB_ACT_CASE(PB_ACT_LEAKY_RELU) PB_ACT_CASE(PB_ACT_HSWISH) PB_ACT_CASE(PB_ACT_HSIGMOID)
        PB_ACT_CASE(PB_ACT_SILU)
        default: break;
      }
#undef PB_ACT_CASE
    }
    if constexpr (STATS) {
      if (m_ok) {                                     // (fp32 values, before the bf16 rounding of the store)
        csum += acc;
        csq += acc * acc;
      }
    }
    bf16x4 o = __builtin_convertvector(acc, bf16x4);
    if (m_ok && col_ok) {
      if (rrow) {
        const bf16x4 rv = *reinterpret_cast<const bf16x4*>(rrow + fk * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16_t)((float)o[e] + (float)rv[e]);
      }
      *reinterpret_cast<bf16x4*>(yrow + fk * 4) = o;
    }
  };

  int it = 0, ibuf = 0;
#pragma unroll 1
  for (; it < NS - 1; ++it) {
    issue(it, ibuf);
    if (++ibuf == NS) ibuf = 0;
  }
  int cbuf = 0;
  const int64_t row_step = (int64_t)16 * a.N;
#pragma unroll 1
  for (int ct = 0; ct < n_my; ++ct) {
    // panel ct must have landed; the NS-2 younger stages and, in order behind and between them, the stores of the last
    // panels stay in flight (vmcnt retires in order: allowing fewer operations than were issued only waits longer)
    if (NS > 2 && ct >= NS - 1) {
      if (allow >= 24) wait_vm<24>();
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
    issue(it, ibuf);
    ++it;
    if (++ibuf == NS) ibuf = 0;
    const unsigned char* As = smem + cbuf * stage_bytes;
    if (++cbuf == NS) cbuf = 0;
    if (active) {
      const int64_t m = mbeg + (int64_t)(p + ct * wa.parts) * SBM + frow;
      T* yrow = a.Y + m * a.N + n0;
      const T* rrow = a.R ? a.R + m * a.N + n0 : nullptr;
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
        block(As + rb * 256 * CH, m + rb * 16, yrow + rb * row_step, rrow ? rrow + rb * row_step : nullptr);
    }
  }
  wait_vm<0>();                                      // no DMA may land in LDS after the workgroup has given it back
  if constexpr (STATS) {
    // this wave alone holds its 16 columns: sum over the 16 rows of a block (lanes with equal fk) and add to the slot
    if (active) {
      float* st = a.stats + (size_t)blockIdx.x * 2 * a.N;     // this workgroup's own slot (grid <= PB_STAT_SLOTS)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = csum[r], qv = csq[r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          s += __shfl_xor(s, o, 64);
          qv += __shfl_xor(qv, o, 64);
        }
        const int n = n0 + fk * 4 + r;
        if (frow == 0 && n < a.N) {
          st[n] += s;
          st[a.N + n] += qv;
        }
      }
    }
  }
}

template <int KSMAX, int RB, bool STATS>
int launch_wide(WideArgs wa, int device, hipStream_t st) {
  constexpr int SBM = 16 * RB;
  constexpr int NS = ring_depth(SBM * KSMAX * 64);
  const size_t lds = (size_t)NS * SBM * (wa.s.K / 8) * 16 + 8192;
  static int cus_dev[16] = {0};
  int& cus = cus_dev[device & 15];
  if (cus == 0) {
    if (hipFuncSetAttribute((const void*)pw_stream_wide_kernel<KSMAX, RB, STATS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      pb_set_error("pw_stream_wide: cannot reserve the LDS ring");
      return 1;
    }
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1) cus = 256;
  }
  const int64_t npanels = pb_cdiv(wa.rows_per_group, SBM);
  int64_t parts = cus / ((int64_t)wa.groups * wa.wgc);
  if (parts < 1) parts = 1;
  if (parts > npanels) parts = npanels;
  wa.parts = (int)parts;
  const int64_t grid = (int64_t)wa.groups * parts * wa.wgc;
  if (wa.s.stats && grid > PB_STAT_SLOTS) return -1;
  hipLaunchKernelGGL((pw_stream_wide_kernel<KSMAX, RB, STATS>), dim3((unsigned)grid), dim3(STHREADS), lds, st, wa);
  return 0;
}

template <bool STATS>
int dispatch_wide(const WideArgs& wa, int device, hipStream_t st) {
  const int nks = (wa.s.K + 31) / 32;
  if (nks <= 8) return launch_wide<8, 2, STATS>(wa, device, st);
  if (nks <= 16) return launch_wide<16, 2, STATS>(wa, device, st);
  if (nks <= 24) return launch_wide<24, 2, STATS>(wa, device, st);
  return launch_wide<32, 1, STATS>(wa, device, st);
}

bool wide_takes(int64_t M, int K, int N, int groups, int64_t rows_per_group) {
  static const int on = getenv("PB_GEMM_WIDE") ? atoi(getenv("PB_GEMM_WIDE")) : 1;
  if (!on || K % 8 || N % 8 || K < 64 || K > 768 || M < 2048) return false;
  // measured in the MobileNetLarge3D step (MI355X) against the ring kernel: 65 -> 58 us on 672 -> 112 (M = 112896), 57 -> 48
  // and 51 -> 45 us on the 100352-row 672 / 480 -> 112 layers; but slower where few of the 8 waves have columns (120 -> 40:
  // 33 -> 55 us) or X is walked twice with 16-row stages (960 -> 160: 35 -> 44 us) - those stay on the ring kernel
  const int64_t blocks = pb_cdiv(N, 16), wgc = pb_cdiv(blocks, 8);
  if (wgc > 2 || blocks * 4 < wgc * 8 * 3) return false;        // >= 3/4 of the waves busy; at most one re-read of X (through L2)
  if (groups > 1 && (rows_per_group < 64 || (int64_t)groups * wgc > 256)) return false;
  return true;
}

}  // namespace

__attribute__((visibility("hidden"))) int pb_pw_stream_wide_takes(int64_t M, int K, int N, int groups, int64_t rows_per_group) {
  return wide_takes(M, K, N, groups, rows_per_group) ? 1 : 0;
}

// -> 0 launched, > 0 error, -1 shape not taken (the caller falls back to the ring kernel)
__attribute__((visibility("hidden"))) int pb_pw_stream_wide_fwd(const void* x, const void* wb, void* y, int64_t M, int K, int N,
                                                                int groups, int64_t rows_per_group, const float* out_scale,
                                                                const float* out_shift, pb_act out_act, float out_slope,
                                                                float* stats, const void* residual, int device, hipStream_t st) {
  if (!wide_takes(M, K, N, groups, rows_per_group)) return -1;
  WideArgs wa = {};
  StreamArgs& a = wa.s;
  a.X = (const bf16_t*)x; a.WB = (const bf16_t*)wb; a.Y = (bf16_t*)y; a.R = (const bf16_t*)residual;
  a.M = M; a.K = K; a.N = N;
  a.out_scale = out_scale; a.out_shift = out_shift; a.out_act = (int)out_act; a.out_slope = out_slope; a.stats = stats;
  wa.groups = groups > 1 ? groups : 1;
  wa.rows_per_group = groups > 1 ? rows_per_group : M;
  wa.w_group_stride = groups > 1 ? (int64_t)N * K : 0;
  wa.wgc = (int)pb_cdiv(pb_cdiv(N, 16), 8);
  return stats ? dispatch_wide<true>(wa, device, st) : dispatch_wide<false>(wa, device, st);
}
