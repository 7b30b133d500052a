// Depthwise (1,k,k) "same" convolution, bf16, with the dw_rows thread layout (lanes along channel pairs, a rolling k-row
// fp32 window in registers, every input element loaded and converted once) - but the input rows no longer travel
// through registers: each WAVE owns a private ring of R row slots in LDS and fills it with `global_load_lds_dwordx4`
// R rows AHEAD of the row it is computing (round 4).
//
// Why: dw_rows keeps ONE row step of loads in flight per wave (the rows wait in VGPRs, and 140-233 registers leave 2-3
// waves per SIMD): ~30-40 KB in flight per CU, a chain of one memory round trip per output row - 3.0-3.8 TB/s in the
// step although the arithmetic sustains 3.9-4.8 TB/s from cache (profiles/r03_dw_standalone.txt).  The LDS-DMA takes no
// registers and no issue slots beyond the instruction itself, so the bytes in flight are set by the ring (R x 3-4 KB per
// wave, 100-130 KB per CU), not by the register file.
//
// Wave = one work unit: (frame, row band, segment group, channel slice).  Its 64 lanes are NSG column segments x 4*CW
// channel pairs (CW <= 16 chunks of 16 B = 8 channels, NSG*CW <= 16, both chosen per layer on the host).  One ring slot
// holds the IW = (WSEG-1)*S+K input pixels of every segment of the wave for its channel slice, in the order
// [pixel j][segment][chunk] - i.e. pixel j of ALL lanes is one contiguous run of dwords (conflict-free ds_read_b32), and
// a DMA instruction (64 lanes x 16 B) copies 64 consecutive chunks of that order, each lane from its own address.
// Everything is wave-private: no barrier, only the wave's own counted `s_waitcnt vmcnt`.
//
// vmcnt counts loads, DMAs and stores together in issue order, so every global memory operation inside the row loop is
// inline asm with a STATIC count per step (NDMA DMAs per row, WSEG stores per output row):
//   * the last, ragged segment of a row is right-aligned (it starts at Wd-WSEG and recomputes a few columns of its
//     neighbour with identical values), segments and channel chunks beyond the tensor are clamped onto the last valid
//     one (duplicates that compute and store the same values) - so every lane of every wave stores exactly WSEG outputs
//     per row and the control flow is wave-uniform; only the squeeze-excite pool sums mask the duplicates;
//   * a row r is waited for with vmcnt(N), N <= the number of operations issued after its DMAs (the rows issued behind
//     it and the stores of the steps since; `wait_row` below), computed per step (wave-uniform) and rounded DOWN onto a
//     ladder of immediates (a smaller N only waits longer).
#include "dw_internal.h"
#include <stdlib.h>

namespace {

struct RingArgs {
  const void* src;
  void* dst;
  const float* w;
  const float* g_scale;
  const float* g_shift;
  float* pool;          // [Tdst*nbands*nseg][N][C], one slot per (frame, band, segment); plain stores
  int N, C, Tsrc, Hs, Ws, Tdst, Hd, Wd, t_mul, t_off;
  int nseg, nbands, band_rows;
  int chunks, CW, NSLICE, NSG, nsgq;
  int R;                // ring depth in rows (>= K, a multiple of S)
  int64_t nwaves;
};

__device__ __forceinline__ void ring_dma(unsigned off, uint64_t base, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(off), "s"(base), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void ring_store(unsigned off, unsigned val, uint64_t base) {
  asm volatile("global_store_dword %0, %1, %2" ::"v"(off), "v"(val), "s"(base) : "memory");
}
template <int N> __device__ __forceinline__ void ring_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ uint64_t uniform64(uint64_t v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}

template <int K, int S, int WSEG, bool FLIP, bool PRO>
__global__ __launch_bounds__(256) void dw_ring_fwd_kernel(RingArgs a) {
  constexpr int P = K / 2;
  constexpr int IW = (WSEG - 1) * S + K;
  constexpr int NDMA = (IW + 3) / 4;                  // NSG*CW <= 16 chunks per pixel -> at most IW*16 chunks per row
  constexpr int SLOT = NDMA * 1024;
  const int R = a.R;
  extern __shared__ __attribute__((aligned(16))) unsigned char ring_lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
  if (wid >= a.nwaves) return;
  // wave-uniform decode: slice fastest (the four waves of a workgroup read neighbouring channels of the same pixels)
  int64_t q = wid;
  const int slice = (int)(q % a.NSLICE); q /= a.NSLICE;
  const int sgq = (int)(q % a.nsgq); q /= a.nsgq;
  const int band = (int)(q % a.nbands);
  const int z = (int)(q / a.nbands);
  const int n = z / a.Tdst, td = z - n * a.Tdst;
  const int ts = td * a.t_mul + a.t_off;
  const int h0 = band * a.band_rows;
  const int nrows = min(a.Hd, h0 + a.band_rows) - h0;
  // ---- this lane's (segment, channel pair)
  const int lanes_used = a.NSG * 4 * a.CW;
  const bool lane_ok = lane < lanes_used;
  const int le = lane_ok ? lane : lanes_used - 1;
  const int sgi = le / (4 * a.CW);
  int p = le - sgi * 4 * a.CW;
  const int pmax = 4 * min(a.CW, a.chunks - slice * a.CW) - 1;
  const int sg_raw = sgq * a.NSG + sgi;
  const bool own = lane_ok && p <= pmax && sg_raw < a.nseg;       // not a duplicate: counts for the pool sums
  p = min(p, pmax);
  const int sg = min(sg_raw, a.nseg - 1);
  const int w0 = min(sg * WSEG, a.Wd - WSEG);                      // (the last segment is right-aligned)
  const int jfirst = sg * WSEG - w0;                               // columns below it belong to the neighbour
  const int c0 = (slice * a.CW * 4 + p) * 2;
  const unsigned lds_wave = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)ring_lds + wave * (R * SLOT);
  const uint64_t dst_frame = uniform64((uint64_t)(uintptr_t)a.dst + (((int64_t)z * a.Hd + h0) * a.Wd * a.C) * 2);
  const unsigned ooff0 = (unsigned)((w0 * a.C + c0) * 2);
  const unsigned pixb = (unsigned)(a.C * 2);
  float* pslot = a.pool ? a.pool + ((((int64_t)td * a.nbands + band) * a.nseg + sg) * a.N + n) * a.C + c0 : nullptr;
  if (ts < 0 || ts >= a.Tsrc) {                                    // the reference's zero frames (wave-uniform)
    for (int t = 0; t < nrows; ++t) {
      const uint64_t orow = dst_frame + (uint64_t)t * a.Wd * pixb;
#pragma unroll
      for (int j = 0; j < WSEG; ++j) ring_store(ooff0 + j * pixb, 0u, orow);
    }
    if (pslot && own) *reinterpret_cast<f32x2*>(pslot) = f32x2{0.f, 0.f};
    return;
  }
  // ---- DMA side: lane -> (pixel j, segment, chunk) of each of the NDMA instructions of a row
  unsigned doff[NDMA];
  {
    const int tot = IW * a.NSG * a.CW;
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      int idx = i * 64 + lane;
      if (idx >= tot) idx %= tot;                                  // spare lanes re-fetch a valid chunk (lands behind the row)
      const int c = idx % a.CW;
      const int qq = idx / a.CW;
      const int sd = qq % a.NSG, j = qq / a.NSG;
      const int sgd = min(sgq * a.NSG + sd, a.nseg - 1);
      const int w0d = min(sgd * WSEG, a.Wd - WSEG);
      const int ws = min(max(w0d * S - P + j, 0), a.Ws - 1);
      const int ch = min(slice * a.CW + c, a.chunks - 1);
      doff[i] = (unsigned)((ws * a.C + ch * 8) * 2);
    }
  }
  const uint64_t src_frame = uniform64((uint64_t)(uintptr_t)a.src + (((int64_t)n * a.Tsrc + ts) * a.Hs) * a.Ws * (int64_t)a.C * 2);
  const int hs_base = h0 * S - P;
  const int NR = (nrows - 1) * S + K;                              // window rows of this band
  const unsigned rowb = (unsigned)(a.Ws * a.C * 2);
  auto issue_row = [&](int r) {
    const int hs = min(max(hs_base + r, 0), a.Hs - 1);
    const uint64_t base = src_frame + (uint64_t)hs * rowb;
    const unsigned slot = lds_wave + (unsigned)(r % R) * SLOT;
#pragma unroll
    for (int i = 0; i < NDMA; ++i) ring_dma(doff[i], base, slot + i * 1024);
  };
  // ---- compute side
  unsigned cmask = 0;
#pragma unroll
  for (int j = 0; j < IW; ++j) {
    const int ws = w0 * S - P + j;
    cmask |= (ws >= 0 && ws < a.Ws) ? (1u << j) : 0u;
  }
  float wt[2 * K * K];
#pragma unroll
  for (int i = 0; i < 2 * K * K; ++i) wt[i] = a.w[(int64_t)c0 * (K * K) + i];
  f32x2 gs = {1.f, 1.f}, gh = {0.f, 0.f};
  if (PRO) {
    if (a.g_scale) gs = *reinterpret_cast<const f32x2*>(a.g_scale + (int64_t)n * a.C + c0);
    if (a.g_shift) gh = *reinterpret_cast<const f32x2*>(a.g_shift + (int64_t)n * a.C + c0);
  }
  // the compiler waits for a load at its first use: use every loaded register HERE, or it would put vmcnt(0) in front of
  // the first FMA of every row step; from here on vmcnt is counted by hand
#pragma unroll
  for (int i = 0; i < 2 * K * K; ++i) asm volatile("" : "+v"(wt[i]));
  asm volatile("" : "+v"(gs), "+v"(gh));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const int pre = min(R, NR);
  for (int r = 0; r < pre; ++r) issue_row(r);

  const unsigned rd_lane = (unsigned)(sgi * a.CW * 16 + p * 4);    // this lane's dword inside a pixel's run
  const unsigned pix_stride = (unsigned)(a.NSG * a.CW * 16);
  typedef __attribute__((address_space(3))) const unsigned* lds_cu32;
  constexpr unsigned FULL = IW >= 32 ? 0xffffffffu : ((1u << IW) - 1u);
  f32x2 win[K][IW];
  auto read_row = [&](int r, f32x2 (&slot)[IW]) {                  // ring slot -> window slot (converted, padded)
    const unsigned base = lds_wave + (unsigned)(r % R) * SLOT + rd_lane;
    unsigned raw[IW];
#pragma unroll
    for (int j = 0; j < IW; ++j) raw[j] = *(lds_cu32)(uintptr_t)(base + j * pix_stride);
    const int hs = hs_base + r;
    const bool rowok = hs >= 0 && hs < a.Hs;
    if (rowok && cmask == FULL) {
#pragma unroll
      for (int j = 0; j < IW; ++j) slot[j] = PRO ? Pair<bf16_t>::up(raw[j]) * gs + gh : Pair<bf16_t>::up(raw[j]);
    } else {
#pragma unroll
      for (int j = 0; j < IW; ++j) {
        const bool ok = rowok && ((cmask >> j) & 1u);
        const f32x2 v = PRO ? Pair<bf16_t>::up(raw[j]) * gs + gh : Pair<bf16_t>::up(raw[j]);
        slot[j] = ok ? v : f32x2{0.f, 0.f};
      }
    }
  };
  // wait until row r (the LAST new row of its step) has landed: all but the N youngest operations complete, N <= what was
  // issued behind row r's DMAs.  A step is: wait - read the new rows - refill their slots (rows +R) - compute - WSEG
  // stores.  So behind row r sit the rows up to r-S+R (refills of the earlier steps; step 0: what the prologue issued)
  // and the stores of every step since the one that issued row r (prologue rows: of every step so far).  N is rounded
  // DOWN onto a ladder of immediates (all wave-uniform: scalar compares).
  auto step_of = [&](int r) { return r < K ? 0 : (r - K) / S + 1; };
  auto wait_row = [&](int r) {
    const int t = step_of(r);
    const int rows_behind = t == 0 ? pre - K : min(NR - 1 - r, R - S);
    const int behind = rows_behind * NDMA + (t - (r < pre ? 0 : step_of(r - R))) * WSEG;
    if (behind >= 63) ring_wait<63>();
    else if (behind >= 52) ring_wait<52>();
    else if (behind >= 42) ring_wait<42>();
    else if (behind >= 33) ring_wait<33>();
    else if (behind >= 25) ring_wait<25>();
    else if (behind >= 18) ring_wait<18>();
    else if (behind >= 12) ring_wait<12>();
    else if (behind >= 7) ring_wait<7>();
    else if (behind >= 3) ring_wait<3>();
    else ring_wait<0>();
  };
  auto refill = [&](int r) {                                       // row r has been read: its slot takes row r + R
    if (r + R < NR) issue_row(r + R);
  };
  // first step: all K window rows
  wait_row(K - 1);
#pragma unroll
  for (int kh = 0; kh < K; ++kh) read_row(kh, win[kh % K]);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int kh = 0; kh < K; ++kh) refill(kh);
  f32x2 psum = {0.f, 0.f};
  for (int tk = 0; tk < nrows; tk += K) {
#pragma unroll
    for (int rr = 0; rr < K; ++rr) {
      const int t = tk + rr;
      if (t < nrows) {
        if (t > 0) {
          const int r0 = t * S + K - S;                            // new window rows r0 .. r0+S-1
          wait_row(r0 + S - 1);
#pragma unroll
          for (int s = 0; s < S; ++s) read_row(r0 + s, win[(rr * S + K - S + s) % K]);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int s = 0; s < S; ++s) refill(r0 + s);
        }
        f32x2 acc[WSEG];
#pragma unroll
        for (int j = 0; j < WSEG; ++j) acc[j] = f32x2{0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < K; ++kh)
#pragma unroll
          for (int kw = 0; kw < K; ++kw) {
            const int tap = FLIP ? (K - 1 - kh) * K + (K - 1 - kw) : kh * K + kw;
            const f32x2 w2 = f32x2{wt[tap], wt[K * K + tap]};
#pragma unroll
            for (int j = 0; j < WSEG; ++j) pk_fma_keep(acc[j], win[(rr * S + kh) % K][j * S + kw], w2);
          }
        const uint64_t orow = dst_frame + (uint64_t)t * a.Wd * pixb;
#pragma unroll
        for (int j = 0; j < WSEG; ++j) {
          const unsigned o = Pair<bf16_t>::down(acc[j]);
          ring_store(ooff0 + j * pixb, o, orow);
          if (a.pool && j >= jfirst) psum += Pair<bf16_t>::up(o);
        }
      }
    }
  }
  if (pslot && own) *reinterpret_cast<f32x2*>(pslot) = psum;
}

// (K, S) -> segment width (as dw_rows) and ring depth: R >= K rows, R a multiple of S, R * NDMA KiB per wave so that two
// to three workgroups fit a CU's 160 KB
constexpr int ring_wseg(int K, int S, bool narrow) {
  return narrow ? (K == 3 ? (S == 1 ? 7 : 4) : (S == 1 ? 4 : 2)) : (K == 3 ? (S == 1 ? 14 : 7) : (S == 1 ? 7 : 4));
}
int ring_depth_rows(int K, int S, int WSEG) {
  // PB_DW_RING_KB: LDS per wave (default 16 KB: two workgroups per CU; 12: three)
  static const int kb = getenv("PB_DW_RING_KB") ? atoi(getenv("PB_DW_RING_KB")) : 16;
  const int IW = (WSEG - 1) * S + K, ndma = (IW + 3) / 4;
  int r = kb / ndma;
  if (r < K) r = K;
  if (r % S) r += S - r % S;
  return r;
}

bool ring_enabled() {
  static const bool on = !(getenv("PB_DW_RING") && atoi(getenv("PB_DW_RING")) == 0);
  return on;
}
bool ring_narrow(int K, int S, int C) {
  // PB_DW_RING_NARROW: bit 0 = k3 s1, bit 1 = k3 s2, bit 2 = k5 s1, bit 3 = k5 s2 take the half-width segments.
  // Default (measured in the step, profiles/r04_dw_ring_sweep.txt): only the first-stage layers (C <= 32, 112-wide frames)
  static const int narrow = getenv("PB_DW_RING_NARROW") ? atoi(getenv("PB_DW_RING_NARROW")) : -1;
  if (narrow < 0) return C <= 32 && K == 3 && S == 1;
  const int bit = (K == 5 ? 2 : 0) + (S == 2 ? 1 : 0);
  return (narrow >> bit) & 1;
}
// Which layers take the ring (measured per layer on the MobileNetLarge3D step, batch 32, against dw_rows): the stride-2
// layers (97.7 -> 81 us, 41.7 -> 31.6, 45.0 -> 41) and the 16-channel 112x112 stage (67.3 -> 54, 55.5 -> 48) gain; the
// 14x14 / 28x28 / 56x56 stride-1 layers do not (they are not bound by the load round trip: equal or 5-20 % slower with
// the ring's coarser lane packing), so they stay on the register-window kernel.  PB_DW_RING_ALL=1: every eligible layer.
bool ring_pays(int S, int C) {
  static const bool all = getenv("PB_DW_RING_ALL") && atoi(getenv("PB_DW_RING_ALL")) != 0;
  return all || S == 2 || C <= 32;
}

// channel slices and segments per wave: the (CW, NSG) with the most useful lanes, ties to the wider slice
void ring_lanes(RingArgs& a) {
  a.chunks = a.C / 8;
  double best = -1.0;
  for (int cw = 1; cw <= 16 && cw <= a.chunks; ++cw) {
    const int nslice = (int)pb_cdiv(a.chunks, cw);
    int nsg = 16 / cw;
    if (nsg > a.nseg) nsg = a.nseg;
    const int nsgq = (int)pb_cdiv(a.nseg, nsg);
    // useful lane-work / lanes spent
    const double util = ((double)a.chunks * a.nseg) / ((double)nslice * cw * nsgq * nsg) * (nsg * 4.0 * cw / 64.0);
    if (util >= best - 1e-9) {
      best = util; a.CW = cw; a.NSLICE = nslice; a.NSG = nsg; a.nsgq = nsgq;
    }
  }
}

void ring_geom(RingArgs& a, int K, int WSEG) {
  a.nseg = (int)pb_cdiv(a.Wd, WSEG);
  ring_lanes(a);
  const int64_t frames = (int64_t)a.N * a.Tdst;
  const int64_t per_band = (int64_t)a.NSLICE * a.nsgq;
  // ~10 waves per CU; row bands (multiples of K) only when the frames alone give fewer waves than that
  static const int want_env = getenv("PB_DW_RING_WAVES") ? atoi(getenv("PB_DW_RING_WAVES")) : 0;
  const int want = want_env ? want_env : (a.C <= 32 ? 5120 : 2560);       // (narrow first-stage kernels: 4 waves per SIMD fit)
  int nbands = (int)pb_cdiv(want, per_band * frames);
  const int max_bands = (int)pb_cdiv(a.Hd, 2 * K);
  if (nbands > max_bands) nbands = max_bands;
  if (nbands < 1) nbands = 1;
  a.band_rows = (int)pb_cdiv(pb_cdiv(a.Hd, nbands), K) * K;
  a.nbands = (int)pb_cdiv(a.Hd, a.band_rows);
  a.nwaves = frames * a.nbands * per_band;
}

bool ring_ok(const pb_dwconv_shape* s, pb_dtype dtype) {
  if (!ring_enabled() || dtype != PB_BF16 || !ring_pays(s->sh, s->C)) return false;
  if (!(s->kt == 1 && s->kh == s->kw && (s->kh == 3 || s->kh == 5) && s->sh == s->sw && (s->sh == 1 || s->sh == 2) &&
        s->ph == s->kh / 2 && s->pw == s->kw / 2 && s->C % 8 == 0 && s->C >= 16 &&
        s->Ho == (s->Hi + 2 * s->ph - s->kh) / s->sh + 1 && s->Wo == (s->Wi + 2 * s->pw - s->kw) / s->sw + 1))
    return false;
  const int WSEG = ring_wseg(s->kh, s->sh, ring_narrow(s->kh, s->sh, s->C));
  if (s->Wo < WSEG) return false;                                  // (right-aligned last segment needs one full segment)
  // 32-bit byte offsets inside a frame / a tensor
  if ((int64_t)s->N * s->Ti * s->Hi * s->Wi * s->C * 2 >= (1LL << 31) || (int64_t)s->N * s->To * s->Ho * s->Wo * s->C * 2 >= (1LL << 31)) return false;
  return true;
}

template <int K, int S, bool NARROW>
int launch_ring(RingArgs a, int flip, hipStream_t st) {
  constexpr int WSEG = ring_wseg(K, S, NARROW);
  constexpr int IW = (WSEG - 1) * S + K, NDMA = (IW + 3) / 4;
  const int R = a.R = ring_depth_rows(K, S, WSEG);
  ring_geom(a, K, WSEG);
  const dim3 grid((unsigned)pb_cdiv(a.nwaves, 4));
  const size_t lds = (size_t)4 * R * NDMA * 1024;
  const bool pro = a.g_scale || a.g_shift;
#define PB_RING_GO(FL, PR)                                                                                       \
  do {                                                                                                           \
    auto kern = dw_ring_fwd_kernel<K, S, WSEG, FL, PR>;                                                        \
    static size_t attr_set = 0;                                                                                  \
    if (attr_set < lds && lds > 48 * 1024) {                                                                          \
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { \
        pb_set_error("dw_ring: cannot raise the dynamic LDS limit"); return 1; }                                 \
      attr_set = lds;                                                                                            \
    }                                                                                                            \
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);                                                       \
  } while (0)
  if (flip) { if (pro) PB_RING_GO(true, true); else PB_RING_GO(true, false); }
  else { if (pro) PB_RING_GO(false, true); else PB_RING_GO(false, false); }
#undef PB_RING_GO
  if (hipGetLastError() != hipSuccess) { pb_set_error("dw_ring forward launch failed"); return 1; }
  return 0;
}

RingArgs ring_args(const pb_dwconv_shape* s) {
  RingArgs a = {};
  a.N = s->N; a.C = s->C; a.Tsrc = s->Ti; a.Hs = s->Hi; a.Ws = s->Wi; a.Tdst = s->To; a.Hd = s->Ho; a.Wd = s->Wo;
  a.t_mul = s->st; a.t_off = -s->pt;
  return a;
}

}  // namespace

bool pb_dw_ring_eligible(const pb_dwconv_shape* s, pb_dtype dtype) { return ring_ok(s, dtype); }

int pb_dw_ring_pool_slots(const pb_dwconv_shape* s, pb_dtype dtype) {
  if (!ring_ok(s, dtype)) return 0;
  RingArgs a = ring_args(s);
  ring_geom(a, s->kh, ring_wseg(s->kh, s->sh, ring_narrow(s->kh, s->sh, s->C)));
  return a.Tdst * a.nbands * a.nseg;
}

int pb_dw_ring_fwd(const void* src, const float* w, void* dst, const pb_dwconv_shape* s, pb_dtype dtype, int flip,
                   const float* g_scale, const float* g_shift, float* pool, hipStream_t st) {
  if (!ring_ok(s, dtype)) return -1;
  RingArgs a = ring_args(s);
  a.src = src; a.dst = dst; a.w = w; a.g_scale = g_scale; a.g_shift = g_shift; a.pool = pool;
  const int K = s->kh, S = s->sh;
  const bool nar = ring_narrow(K, S, s->C);
#define PB_RING(KK, SS) (nar ? launch_ring<KK, SS, true>(a, flip, st) : launch_ring<KK, SS, false>(a, flip, st))
  if (K == 3 && S == 1) return PB_RING(3, 1);
  if (K == 3 && S == 2) return PB_RING(3, 2);
  if (K == 5 && S == 1) return PB_RING(5, 1);
  return PB_RING(5, 2);
#undef PB_RING
}
