// Dense 3x3x3 convolution (stride 1) with the input HALO resident in LDS: MobileViT's conv_nxn_bn (mobilevit.py:19-24,
// :112, :119) forward and backward-data.
//
// The implicit GEMM on the pointwise LDS-DMA kernel (pw_gemm_dma.hip, CONV) fetches every 128-voxel x 64-channel A
// block once per tap, 27 times through L2: 56 flop per byte of L2 traffic, which is what bounds it (140-170 TFLOP/s).
// Here a workgroup owns a 2 x 8 x 8 block of output voxels (128 GEMM rows).  For a 32-channel chunk it DMAs the
// 4 x 10 x 10 halo of input voxels into LDS ONCE (26 KB, padding voxels and the channel tail come from a zero page)
// and runs all 27 taps from it: the A fragment of output row r and tap (at, ah, aw) is one ds_read_b128 at the
// halo voxel shifted by the tap - 8.3x fewer A bytes.  Only the filter slices [BN x 32] stream, through a ring of
// NSW stages kept NSW-1 taps ahead (global_load_lds_dwordx4, counted vmcnt, one barrier per tap).
//
// LDS rows are 64 B (32 bf16); the 16-byte chunk index is XOR-swizzled with (row >> 2) & 3 on the source address so
// that the 16 rows an MFMA fragment read touches (8 + 8 halo voxels, or 16 consecutive filter rows) spread over all
// banks.  Epilogue (bias/affine/activation, BatchNorm statistics, LDS-transposed 16-byte stores) as in the GEMM kernel;
// its staging tile reuses the halo buffer.
#include "pb_common.h"
#include <stdlib.h>

namespace {

constexpr int HB_T = 2, HB_H = 8, HB_W = 8;                       // output block
constexpr int HT = HB_T + 2, HH = HB_H + 2, HW = HB_W + 2;        // halo
constexpr int HVOX = HT * HH * HW;                                // 400 voxels
constexpr int H_INSTR = (HVOX * 4 + 255) / 256;                   // 7 DMA instructions per thread
constexpr int HALO_BYTES = H_INSTR * 4096;                        // 28 KB (padded to whole instructions)
constexpr int KC = 32;                                            // channels per chunk
constexpr int NSW = 5;                                            // filter-slice ring depth

struct HaloArgs {
  const bf16_t* X;       // [N][Ti][Hi][Wi][Cin]
  const bf16_t* WB;      // [Cout][27][Cin]
  bf16_t* Y;             // [N][To][Ho][Wo][Cout]
  const bf16_t* zero;    // >= 16 zero bytes
  int N, Cin, Cout, Ti, Hi, Wi, To, Ho, Wo, pt, ph, pw;
  int tb, hb, wb;        // blocks per dimension
  int n_tiles;
  int64_t m_tiles;
  const float* out_scale;
  const float* out_shift;
  int out_act;
  float out_slope;
  float* stats;
};

__device__ __forceinline__ void glds16h(const void* gsrc, unsigned char* lds_dst) {
  typedef __attribute__((address_space(3))) unsigned char* lds_b;
  const unsigned dst = (unsigned)(uintptr_t)(lds_b)lds_dst;
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

template <int N> __device__ __forceinline__ void wait_vm();
#define PB_VM(n) template <> __device__ __forceinline__ void wait_vm<n>() { asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); }
PB_VM(0) PB_VM(1) PB_VM(2) PB_VM(3) PB_VM(4) PB_VM(6) PB_VM(8)
#undef PB_VM

template <int NT>
__global__ __launch_bounds__(256) void conv3d_halo_kernel(HaloArgs a) {
  typedef bf16_t T;
  constexpr int BN = NT * 16;
  constexpr int W_INSTR = (BN * 4 + 255) / 256;                  // DMA instructions per thread and filter slice
  constexpr int W_STAGE = W_INSTR * 4096;
  constexpr int EPI_LD = BN + 8;
  extern __shared__ __attribute__((aligned(1024))) unsigned char dsmem[];
  unsigned char* halo = dsmem;                                   // [416+ voxels][32 ch], also the epilogue staging
  unsigned char* wring = dsmem + HALO_BYTES;                     // [NSW][BN rows][32 ch]
  float* sred = (float*)(dsmem + HALO_BYTES + NSW * W_STAGE);    // [4][BN][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 15, fk = lane >> 4;
  const int ccol = lane & 15, rgrp = lane >> 4;
  const int wave_off = __builtin_amdgcn_readfirstlane(wave * 1024);
  const int nck = (a.Cin + KC - 1) / KC;
  const int S = nck * 27;
  const int64_t ldw = (int64_t)27 * a.Cin;
  const bool plain = !a.out_scale && !a.out_shift && a.out_act == PB_ACT_NONE;

  // halo slots of this thread: slot s = q*256 + tid -> voxel s >> 2 (ht, hh, hw), LDS chunk s & 3 holds source chunk
  // (s & 3) ^ ((voxel >> 2) & 3)
  int h_vt[H_INSTR], h_vh[H_INSTR], h_vw[H_INSTR], h_c[H_INSTR];
#pragma unroll
  for (int q = 0; q < H_INSTR; ++q) {
    const int s = q * 256 + tid, hv = s >> 2;
    h_vt[q] = hv < HVOX ? hv / (HH * HW) : -100000;              // padding slots: never in bounds -> zero page
    h_vh[q] = (hv / HW) % HH;
    h_vw[q] = hv % HW;
    h_c[q] = ((s & 3) ^ ((hv >> 2) & 3)) * 8;
  }
  // filter slots: slot s -> row s >> 2, source chunk (s & 3) ^ ((row >> 2) & 3)
  int w_n[W_INSTR], w_c[W_INSTR];
#pragma unroll
  for (int q = 0; q < W_INSTR; ++q) {
    const int s = q * 256 + tid, n = s >> 2;
    w_n[q] = n < BN ? n : -1;
    w_c[q] = ((s & 3) ^ ((n >> 2) & 3)) * 8;
  }
  // A fragment rows of this lane: GEMM rows wave*32 + frow (+16) -> halo voxel of tap (0,0,0)
  int hv0[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = wave * 32 + i * 16 + frow;
    hv0[i] = ((r >> 6) * HH + ((r >> 3) & 7)) * HW + (r & 7);
  }

  f32x4 acc[2][NT];
  float csum[NT], csq[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    csum[j] = csq[j] = 0.f;
    acc[0][j] = acc[1][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const int64_t total = a.m_tiles * a.n_tiles;
  for (int64_t tile = blockIdx.x; tile < total; tile += gridDim.x) {
    const int nt = (int)(tile % a.n_tiles);
    int64_t mt = tile / a.n_tiles;
    const int bw = (int)(mt % a.wb); mt /= a.wb;
    const int bh = (int)(mt % a.hb); mt /= a.hb;
    const int bt = (int)(mt % a.tb);
    const int n = (int)(mt / a.tb);
    const int t0 = bt * HB_T, h0 = bh * HB_H, w0 = bw * HB_W, n0 = nt * BN;
    const int ti0 = t0 - a.pt, hi0 = h0 - a.ph, wi0 = w0 - a.pw;
    const T* xn = a.X + (int64_t)n * a.Ti * a.Hi * a.Wi * a.Cin;

    auto issue_halo = [&](int ck) {
      const int c0 = ck * KC;
#pragma unroll
      for (int q = 0; q < H_INSTR; ++q) {
        const int ti = ti0 + h_vt[q], hi = hi0 + h_vh[q], wi = wi0 + h_vw[q], c = c0 + h_c[q];
        const T* src = a.zero;
        if ((unsigned)ti < (unsigned)a.Ti && (unsigned)hi < (unsigned)a.Hi && (unsigned)wi < (unsigned)a.Wi && c < a.Cin)
          src = xn + (((int64_t)ti * a.Hi + hi) * a.Wi + wi) * a.Cin + c;
        glds16h(src, halo + q * 4096 + wave_off);
      }
    };
    auto issue_w = [&](int s) {                                   // filter slice of step s = ck*27 + tap
      const int ck = s / 27, tap = s - ck * 27;
      unsigned char* dst = wring + (s % NSW) * W_STAGE + wave_off;
#pragma unroll
      for (int q = 0; q < W_INSTR; ++q) {
        const int nn = n0 + w_n[q], c = ck * KC + w_c[q];
        const T* src = a.zero;
        if (w_n[q] >= 0 && nn < a.Cout && c < a.Cin) src = a.WB + (int64_t)nn * ldw + (int64_t)tap * a.Cin + c;
        glds16h(src, dst + q * 4096);
      }
    };

    lds_barrier();                       // the previous tile's epilogue is done with the halo buffer
    issue_halo(0);
#pragma unroll
    for (int s0 = 0; s0 < NSW - 1; ++s0)
      if (s0 < S) issue_w(s0);
    int tap = 0, ck = 0;
    for (int s = 0; s < S; ++s) {
      if (tap == 0) {
        if (s > 0) {
          lds_barrier();                 // everybody is done with the previous chunk's halo
          issue_halo(ck);
        }
        wait_vm<0>();                    // halo and the slices in flight
      } else {
        // slice s is the oldest of min(NSW-1, S-s) requested slices: leave the younger ones in flight
        const int younger = min(NSW - 2, S - 1 - s);
        if (younger >= 3) wait_vm<3 * W_INSTR>();
        else if (younger == 2) wait_vm<2 * W_INSTR>();
        else if (younger == 1) wait_vm<W_INSTR>();
        else wait_vm<0>();
      }
      lds_barrier();                     // slice s (and the halo) landed for everybody; slice s-1's buffer is free
      if (s + NSW - 1 < S) issue_w(s + NSW - 1);
      const int at = tap / 9, ah = (tap / 3) % 3, aw = tap % 3;
      const int toff = (at * HH + ah) * HW + aw;
      const unsigned char* Ws = wring + (s % NSW) * W_STAGE;
      const int hva = hv0[0] + toff, hvb = hv0[1] + toff;
      const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(halo + hva * 64 + ((fk ^ ((hva >> 2) & 3)) << 4));
      const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(halo + hvb * 64 + ((fk ^ ((hvb >> 2) & 3)) << 4));
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int wr = j * 16 + frow;
        const bf16x8 bw_ = *reinterpret_cast<const bf16x8*>(Ws + wr * 64 + ((fk ^ ((wr >> 2) & 3)) << 4));
        acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bw_, acc[0][j], 0, 0, 0);
        acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bw_, acc[1][j], 0, 0, 0);
      }
      if (++tap == 27) { tap = 0; ++ck; }
    }
    lds_barrier();                       // all MFMA reads of the halo are done: it becomes the epilogue staging tile

    // ---- epilogue
    if (!plain) {
      float osc[NT], osh[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int nn = n0 + j * 16 + ccol;
        osc[j] = (a.out_scale && nn < a.Cout) ? a.out_scale[nn] : 1.f;
        osh[j] = (a.out_shift && nn < a.Cout) ? a.out_shift[nn] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] = act_fwd(acc[i][j][r] * osc[j] + osh[j], a.out_act, a.out_slope);
    }
    T* epi = (T*)halo + wave * 16 * EPI_LD;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const bool nok = n0 + j * 16 + ccol < a.Cout;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int rl = rgrp * 4 + r, rr = wave * 32 + i * 16 + rl;
          const bool rok = t0 + (rr >> 6) < a.To && h0 + ((rr >> 3) & 7) < a.Ho && w0 + (rr & 7) < a.Wo;
          const T tv = from_f32<T>(acc[i][j][r]);
          epi[rl * EPI_LD + j * 16 + ccol] = tv;
          if (a.stats && nok && rok) {
            const float fv = to_f32(tv);
            csum[j] += fv;
            csq[j] += fv * fv;
          }
        }
        acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      constexpr int CPR = BN / 8;
      for (int f = lane; f < 16 * CPR; f += 64) {
        const int rl = f / CPR, c = f - rl * CPR;
        const int rr = wave * 32 + i * 16 + rl;
        const int to = t0 + (rr >> 6), ho = h0 + ((rr >> 3) & 7), wo = w0 + (rr & 7);
        const int nn = n0 + c * 8;
        if (to < a.To && ho < a.Ho && wo < a.Wo && nn < a.Cout)
          vec_store<T>(a.Y + ((((int64_t)n * a.To + to) * a.Ho + ho) * a.Wo + wo) * a.Cout + nn, vec_load<T>(epi + rl * EPI_LD + c * 8));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
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
        csum[j] = csq[j] = 0.f;
      }
      lds_barrier();
      float* st = a.stats + (size_t)blockIdx.x * 2 * a.Cout;     // this workgroup's own slot: plain read-modify-write
      for (int c = tid; c < BN; c += 256) {
        const int nn = n0 + c;
        if (nn < a.Cout) {
          float s = 0.f, q = 0.f;
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            s += sred[(w * BN + c) * 2 + 0];
            q += sred[(w * BN + c) * 2 + 1];
          }
          st[nn] += s;
          st[a.Cout + nn] += q;
        }
      }
    }
  }
}

template <int NT>
int launch_halo(const HaloArgs& a, int device, hipStream_t st) {
  constexpr int BN = NT * 16;
  constexpr int W_STAGE = ((BN * 4 + 255) / 256) * 4096;
  constexpr size_t lds = (size_t)HALO_BYTES + NSW * W_STAGE + 4 * BN * 2 * 4;
  static int resident_dev[16] = {0};                             // per device: the attribute is per (function, device)
  int& resident = resident_dev[device & 15];
  if (resident == 0) {
    if (hipFuncSetAttribute((const void*)conv3d_halo_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      pb_set_error("conv3d halo: cannot reserve %d bytes of LDS", (int)lds);
      return 1;
    }
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv3d_halo_kernel<NT>, 256, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1) cus = 256;
    resident = per_cu * cus;
  }
  int64_t blocks = a.m_tiles * a.n_tiles;
  if (blocks > resident) blocks = resident;
  if (a.stats && blocks > PB_STAT_SLOTS) blocks = PB_STAT_SLOTS;    // one statistics slot per workgroup
  hipLaunchKernelGGL((conv3d_halo_kernel<NT>), dim3((unsigned)blocks), dim3(256), lds, st, a);
  return 0;
}

}  // namespace

// -1: shape not taken (the caller falls back to the implicit GEMM)
__attribute__((visibility("hidden"))) int pb_conv3d_halo_fwd(const void* x, const void* wb, void* y, const pb_conv_shape* s, const void* zero16,
                       const float* out_scale, const float* out_shift, pb_act out_act, float out_slope, float* stats,
                       int device, hipStream_t st) {
  static const bool on = !(getenv("PB_CONV_HALO") && atoi(getenv("PB_CONV_HALO")) == 0);
  if (!on || s->kt != 3 || s->kh != 3 || s->kw != 3 || s->st != 1 || s->sh != 1 || s->sw != 1) return -1;
  if (s->Cin % 8 || s->Cout % 8) return -1;
  if ((int64_t)s->N * s->Ti * s->Hi * s->Wi * s->Cin >= (1LL << 40)) return -1;
  HaloArgs a = {};
  a.X = (const bf16_t*)x; a.WB = (const bf16_t*)wb; a.Y = (bf16_t*)y; a.zero = (const bf16_t*)zero16;
  a.N = s->N; a.Cin = s->Cin; a.Cout = s->Cout; a.Ti = s->Ti; a.Hi = s->Hi; a.Wi = s->Wi;
  a.To = s->To; a.Ho = s->Ho; a.Wo = s->Wo; a.pt = s->pt; a.ph = s->ph; a.pw = s->pw;
  a.tb = (int)pb_cdiv(s->To, HB_T); a.hb = (int)pb_cdiv(s->Ho, HB_H); a.wb = (int)pb_cdiv(s->Wo, HB_W);
  a.m_tiles = (int64_t)s->N * a.tb * a.hb * a.wb;
  a.out_scale = out_scale; a.out_shift = out_shift; a.out_act = (int)out_act; a.out_slope = out_slope; a.stats = stats;
  // 64-column tiles: 51 KB of LDS = three workgroups per CU, which measured faster than the 96/128-wide tiles (two per
  // CU) on every MobileViT layer even where 64 pads Cout (96 -> 128)
  static const int force_nt = getenv("PB_CONV_HALO_NT") ? atoi(getenv("PB_CONV_HALO_NT")) : 0;
  int nt = 4;
  // few voxel blocks (small frames): 32-column tiles double the number of workgroups
  static const int64_t narrow_below = getenv("PB_CONV_HALO_NARROW") ? atoi(getenv("PB_CONV_HALO_NARROW")) : 256;
  if (a.m_tiles * pb_cdiv(s->Cout, 64) < narrow_below) nt = 2;
  if (force_nt == 2 || force_nt == 4 || force_nt == 6 || force_nt == 8) nt = force_nt;
  a.n_tiles = (int)pb_cdiv(s->Cout, nt * 16);
  int lrc;
  if (nt == 2) lrc = launch_halo<2>(a, device, st);
  else if (nt == 4) lrc = launch_halo<4>(a, device, st);
  else if (nt == 6) lrc = launch_halo<6>(a, device, st);
  else lrc = launch_halo<8>(a, device, st);
  if (lrc) return lrc;
  if (hipGetLastError() != hipSuccess) { pb_set_error("conv3d halo launch failed"); return 1; }
  return 0;
}
