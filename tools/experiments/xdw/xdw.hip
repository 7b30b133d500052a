// Fused expansion (1x1x1) conv -> depthwise (1,k,k) conv of a Bottleneck3D (mobilenet.py:84-87: pointwise_conv1 then
// depthwise_conv with NOTHING in between): the expanded tensor e = pw1(x) never exists in HBM.
//
//   forward      d = dw(pw1(x))                      reads x [M,Cin] + writes d [Mo,E]   (unfused: + write e, + read e)
//   weight grad  dWdw += dS (*) pw1(x)               reads x + dS                        (unfused: reads e + dS)
//
// e is the largest tensor of a bottleneck (E = 3..6 x Cin channels at the input resolution); in a training step it was
// written once and read twice.  Here a workgroup owns a spatial tile of ONE frame (the depthwise kernel is (1,k,k): an
// output frame sees exactly one input frame) and walks the expanded channels in chunks of EC <= 64:
//
//   A  x halo tile [HP pixels][Cin] -> LDS (zero outside the frame: pw1 has no bias, so e is exactly 0 there = the
//      depthwise conv's zero padding), once per tile
//   per chunk:
//   B  W1 chunk [EC][Cin] (bf16 copy) -> LDS;  e^T = W1c . x^T on v_mfma_f32_16x16x32_bf16 with the WEIGHTS as the A
//      operand: a lane ends up with 4 consecutive channels of one pixel = one 8-byte LDS write into e[pixel][channel]
//      (bf16, rounded exactly as the unfused GEMM stores it); waves split the pixel blocks, every B fragment read from LDS
//      feeds all (<= 4) channel blocks of the chunk
//   C  depthwise stencil out of the LDS tile: lanes along channel pairs, a thread owns a WSEG-wide segment of output
//      columns and walks down a band of rows with a rolling k-row window in registers (the dw_rows.hip walk, fed from LDS
//      instead of HBM; no padding logic - the tile already holds the zeros); d is stored, the squeeze-excite pool partial
//      sums (mobilenet.py:15) and, in the weight-gradient kernel, the k*k tap sums are reduced in LDS in a fixed order.
//
// Stride 2 reads only the input frames / rows that an output needs (the unfused path expands all of them).
// Nothing here uses atomics; every cross-workgroup sum goes through slots (include/picklebot_amd.h).
#include "dw_internal.h"
#include <stdlib.h>

namespace {

struct XdwArgs {
  const bf16_t* x;       // [N, Ti, Hi, Wi, Cin]
  const bf16_t* w1;      // [E, Cin] bf16 copy of pointwise_conv1.weight
  const float* wdw;      // [E, K*K]
  bf16_t* d;             // forward: [N, To, Ho, Wo, E] (written)
  const bf16_t* dy;      // weight gradient: dS [N, To, Ho, Wo, E]
  const float* g_scale;  // weight gradient prologue g = dS * g_scale[n, c] + g_shift[n, c] (either may be null)
  const float* g_shift;
  float* part;           // forward: pool partials [slots][N][E] (or null); weight gradient: [slots][E][K*K]
  int N, Cin, E, Ti, Hi, Wi, To, Ho, Wo;
  int t_mul, t_off;      // input frame of output frame to: to * t_mul + t_off
  int TH, TW;            // output tile
  int tiles_y, tiles_x;
  int IHt, IWt, HP;      // halo tile (input pixels), HP = IHt * IWt
  int KS;                // k-steps of 32 (Cin padded to 32 * KS)
  int EC, nchunk;        // channels per chunk (16, 32 or 64), chunks per workgroup range
  int esplit;            // workgroups along E; workgroup y handles chunks [y * nchunk, ...)
  int xpitch;            // LDS row pitch in bytes of the x tile
  int nseg, nbands, band_rows;
  int x_off, e_off, a_off, w_off, r_off;   // LDS offsets in bytes; e / a / r are double buffers of e_size / a_size / r_size bytes
  int e_size, a_size, a_wave, w_size, r_size; // a_wave: bytes of one producer wave's region of a W1 tile; w: the whole range's depthwise weights
  int dbg;               // PB_XDW_DBG bits (timing experiments only): 1 = consumers idle, 2 = producers idle, 4 = no x tile load
};

constexpr int XDW_MAXKS = 5;    // Cin <= 160
constexpr int XDW_MAXCB = 4;    // EC <= 64
constexpr int XDW_EPITCH = 136; // bytes per pixel of the e tile: 64 channels bf16 + 8 (a constant: columns become immediate offsets)
constexpr int XDW_SLACK = 16;   // zeroed pixels behind an e tile (ragged last segments read into them)


typedef __attribute__((ext_vector_type(8))) __bf16 frag_t;

// 16 bytes per lane, global -> LDS without registers (lane l lands at lds_dst + 16 l; lds_dst is wave-uniform).  The compiler does
// not see these loads: the issuing wave orders them by hand (xdw_wait_dma before it reads what it fetched).
__device__ __forceinline__ void xdw_glds16(const void* gsrc, unsigned char* lds_dst) {
  typedef __attribute__((address_space(3))) unsigned char* lds_b;
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_b)lds_dst);   // (wave-uniform by construction)
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
__device__ __forceinline__ void xdw_wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// e[pixel][channel] for ALL pixel blocks and this producer wave's NCB channel blocks (starting at block cb0), KS k-steps:
// straight-line MFMA code (no per-MFMA branches: every conditional MFMA used to sit in its own basic block behind an
// s_waitcnt vmcnt(0)).  The producer waves split the CHANNEL blocks, so every W1 fragment is fetched from L2 once per
// workgroup (splitting the pixel blocks had every wave fetch the whole chunk: twice the activation bytes in weight traffic).
template <int NCB, int KS>
__device__ __forceinline__ void xdw_produce_blocks(const unsigned char* aw, int apitch, const unsigned char* xs, unsigned char* es,
                                                   int xpitch, int Cin, int ec, int npb, int cb0, int lane) {
  const bool ktail_ok = (KS - 1) * 32 + (lane >> 4) * 8 < Cin;      // only the last k-step can reach past Cin
  frag_t w[NCB][KS];                                                // this wave's W1 fragments (its rows of the LDS tile)
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      w[cb][ks] = *reinterpret_cast<const frag_t*>(aw + (size_t)(cb * 16 + (lane & 15)) * apitch + ks * 64 + (lane >> 4) * 16);
    if (!ktail_ok) w[cb][KS - 1] = frag_t{0, 0, 0, 0, 0, 0, 0, 0};
  }
  for (int pb = 0; pb < npb; ++pb) {
    const unsigned char* xrow = xs + (size_t)(pb * 16 + (lane & 15)) * xpitch + (lane >> 4) * 16;
    frag_t xb[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xb[ks] = *reinterpret_cast<const frag_t*>(xrow + ks * 64);
    if (!ktail_ok) xb[KS - 1] = frag_t{0, 0, 0, 0, 0, 0, 0, 0};
    f32x4 acc[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[cb][ks], xb[ks], acc[cb], 0, 0, 0);
    unsigned char* erow = es + (size_t)(pb * 16 + (lane & 15)) * XDW_EPITCH + (lane >> 4) * 8;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      if ((cb0 + cb) * 16 + (lane >> 4) * 4 < ec) {
        bf16x4 o = {(bf16_t)acc[cb][0], (bf16_t)acc[cb][1], (bf16_t)acc[cb][2], (bf16_t)acc[cb][3]};
        *reinterpret_cast<bf16x4*>(erow + (cb0 + cb) * 32) = o;
      }
    }
  }
}

// The workgroup is 8 waves with two roles (wave-uniform branches):
//   waves 4-7 PRODUCE the e tile of chunk c+1 (W1 fragments straight from global/L2 into registers, prefetched a chunk
//             ahead; x fragments from the LDS tile; MFMA; 8-byte LDS writes) and stage that chunk's depthwise weights,
//   waves 0-3 CONSUME the e tile of chunk c (depthwise walk, stores / tap sums),
// on double-buffered LDS tiles with ONE workgroup barrier per chunk: the MFMA pipe, the VALU pipe and the HBM stores of
// a CU run concurrently out of one workgroup, and no global-load latency sits on the consumers' path.
// NC consumer (stencil) waves 0 .. NC-1, NP producer (GEMM) waves NC .. NC+NP-1.  (3 x 3 taps: 12 + 4 waves at <= 128 registers -
// three stencil waves per SIMD issue VALU back to back, one alone issues every other slot; 5 x 5 taps need the registers: 6 + 2.)
template <int K, int S, int WSEG, bool WGRAD, int NC, int NP>
__global__ __launch_bounds__((NC + NP) * 64) void xdw_kernel(XdwArgs a) {
  constexpr int XDW_NC = NC, XDW_NP = NP, XDW_CT = NC * 64, NT = (NC + NP) * 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int P = K / 2;
  constexpr int IW = (WSEG - 1) * S + K;
  constexpr int KK = K * K;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool producer = wave >= XDW_NC;
  const int rt = producer ? tid - XDW_CT : tid;       // thread index inside the role
  // ---- which tile
  int b = blockIdx.x;
  const int tx = b % a.tiles_x; b /= a.tiles_x;
  const int ty = b % a.tiles_y; b /= a.tiles_y;
  const int z = b;                                   // n * To + to
  const int n = z / a.To, to = z - n * a.To;
  const int ti = to * a.t_mul + a.t_off;
  const int oy0 = ty * a.TH, ox0 = tx * a.TW;
  const int th = min(a.TH, a.Ho - oy0), tw = min(a.TW, a.Wo - ox0);
  const int c_begin = blockIdx.y * a.nchunk * a.EC;
  const int c_end = min(a.E, c_begin + a.nchunk * a.EC);
  const int slot = (to * a.tiles_y + ty) * a.tiles_x + tx;
  float* const wg_part = WGRAD ? a.part + ((int64_t)(z * a.tiles_y + ty) * a.tiles_x + tx) * a.E * KK : nullptr;
  if (ti < 0 || ti >= a.Ti) {                        // the reference's zero frames (temporal padding of a (1,k,k) kernel)
    if (!WGRAD) {
      const int cw = (c_end - c_begin) / 2;          // dwords per pixel
      for (int i = tid; i < th * tw * cw; i += NT) {
        const int c2 = i % cw, p = i / cw;
        const int px = p % tw, py = p / tw;
        reinterpret_cast<unsigned*>(a.d + (((int64_t)z * a.Ho + oy0 + py) * a.Wo + ox0 + px) * a.E + c_begin)[c2] = 0u;
      }
      if (a.part)
        for (int c = c_begin + tid; c < c_end; c += 512) a.part[((int64_t)slot * a.N + n) * a.E + c] = 0.f;
    } else {
      for (int i = tid; i < (c_end - c_begin) * KK; i += NT) wg_part[(int64_t)c_begin * KK + i] = 0.f;
    }
    return;
  }
  unsigned char* xs = smem + a.x_off;
  const int iy0 = oy0 * S - P, ix0 = ox0 * S - P;
  const int c8 = a.Cin / 8;                          // 16-byte chunks per x row
  const int npb = (a.HP + 15) / 16;                  // pixel blocks
  // ---- x halo tile -> LDS (zero outside the frame), all 8 waves
  {
    const bf16_t* xf = a.x + ((int64_t)n * a.Ti + ti) * a.Hi * a.Wi * a.Cin;
    const int total = (a.dbg & 4) ? 0 : npb * 16 * c8;
    for (int i = tid; i < total; i += NT) {
      const int ch = i % c8, p = i / c8;
      const int iy = p / a.IWt, ix = p - iy * a.IWt;
      const int gy = iy0 + iy, gx = ix0 + ix;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (p < a.HP && gy >= 0 && gy < a.Hi && gx >= 0 && gx < a.Wi)
        v = *reinterpret_cast<const u32x4*>(xf + ((int64_t)gy * a.Wi + gx) * a.Cin + ch * 8);
      *reinterpret_cast<u32x4*>(xs + (size_t)p * a.xpitch + ch * 16) = v;
    }
    if (!WGRAD) {                                      // the depthwise weights of the whole channel range
      float* wl = reinterpret_cast<float*>(smem + a.w_off);
      for (int i = tid; i < (c_end - c_begin) * KK; i += NT) wl[i] = a.wdw[(int64_t)c_begin * KK + i];
    }
    for (int i = tid; i < 2 * XDW_SLACK * XDW_EPITCH / 4; i += NT) {     // the slack behind both e tiles
      const int bsel = i / (XDW_SLACK * XDW_EPITCH / 4), j = i - bsel * (XDW_SLACK * XDW_EPITCH / 4);
      reinterpret_cast<unsigned*>(smem + a.e_off + (size_t)bsel * a.e_size + (size_t)npb * 16 * XDW_EPITCH)[j] = 0u;
    }
  }
  const int nch = (c_end - c_begin + a.EC - 1) / a.EC;
  const int cpc = a.EC / 2;                          // channel pairs per chunk: 8, 16 or 32

  // ================================================================= producer side
  // channel blocks of this producer wave inside a chunk: [cbw0, cbw0 + ncbw)
  const int ncb_chunk = a.EC / 16, ncbw = (ncb_chunk + XDW_NP - 1) / XDW_NP;
  const int cbw0 = (wave - XDW_NC) * ncbw;
  const int apitch = a.Cin * 2;
  // W1 rows of this wave for chunk ci -> its region of W1 tile (ci & 1), by LDS-DMA (no registers, no compiler-visible loads:
  // the compiler's s_waitcnt vmcnt(0) in front of anything that touches a loaded register used to serialise the prefetch)
  auto dma_w1 = [&](int ci) {
    if (ci >= nch || cbw0 * 16 >= a.EC) return;
    const int c0 = c_begin + ci * a.EC;
    unsigned char* dst = smem + a.a_off + (size_t)(ci & 1) * a.a_size + (size_t)(wave - XDW_NC) * a.a_wave;
    const int pieces = ncbw * 16 * c8;
    for (int q0 = 0; q0 < pieces; q0 += 64) {
      const int q = min(q0 + lane, pieces - 1);       // (lanes past the end re-fetch the last piece into the padding)
      const int r = q / c8, pc = q - r * c8;
      const int row = min(c0 + cbw0 * 16 + r, a.E - 1);
      xdw_glds16(a.w1 + (int64_t)row * a.Cin + pc * 8, dst + (size_t)q0 * 16);
    }
  };
  auto produce = [&](int ci) {
    if (a.dbg & 2) return;
    const int c0 = c_begin + ci * a.EC;
    const int ec = min(a.EC, c_end - c0);
    unsigned char* es = smem + a.e_off + (size_t)(ci & 1) * a.e_size;
    const unsigned char* aw = smem + a.a_off + (size_t)(ci & 1) * a.a_size + (size_t)(wave - XDW_NC) * a.a_wave;
    if (cbw0 * 16 >= ec) return;                     // (narrow last chunk / EC = 16: nothing for this wave)
#define XDW_P(NCB_, KS_) xdw_produce_blocks<NCB_, KS_>(aw, apitch, xs, es, a.xpitch, a.Cin, ec, npb, cbw0, lane)
#define XDW_PK(NCB_)                                                                                              \
  switch (a.KS) { case 1: XDW_P(NCB_, 1); break; case 2: XDW_P(NCB_, 2); break; case 3: XDW_P(NCB_, 3); break;   \
                  case 4: XDW_P(NCB_, 4); break; default: XDW_P(NCB_, 5); break; }
    if (XDW_NP <= 2 && ncbw >= 2 && (cbw0 + 1) * 16 < ec) { XDW_PK(2) } else { XDW_PK(1) }
#undef XDW_PK
#undef XDW_P
  };

  // ================================================================= consumer side
  // stencil unit of a consumer thread: lanes along channel pairs, then segments, then bands
  const int unit = rt / cpc, cp = rt - unit * cpc;
  const int sg = unit % a.nseg, band = unit / a.nseg;
  const int h0 = band * a.band_rows;
  const int nrows = (band < a.nbands) ? min(th, h0 + a.band_rows) - h0 : 0;
  const int w0 = sg * WSEG;
  const bool unit_on = nrows > 0 && w0 < tw;
  auto consume = [&](int ci) {
    const int c0 = c_begin + ci * a.EC;
    const int ec = min(a.EC, c_end - c0);
    const unsigned char* es = smem + a.e_off + (size_t)(ci & 1) * a.e_size;
    float* red = reinterpret_cast<float*>(smem + a.r_off + (size_t)(ci & 1) * a.r_size);
    const bool on = unit_on && 2 * cp < ec && !(a.dbg & 1);
    const int ch0 = c0 + 2 * cp;
    f32x2 psum = {0.f, 0.f};
    f32x2 wacc[WGRAD ? KK : 1];
    if (WGRAD) {
#pragma unroll
      for (int i = 0; i < KK; ++i) wacc[i] = f32x2{0.f, 0.f};
    }
    if (on) {
      float wt[WGRAD ? 1 : 2 * KK];
      f32x2 gs = {1.f, 1.f}, gh = {0.f, 0.f};
      if (!WGRAD) {
        const float* wb = reinterpret_cast<const float*>(smem + a.w_off) + (size_t)(ci * a.EC + 2 * cp) * KK;
#pragma unroll
        for (int i = 0; i < 2 * KK; ++i) wt[i] = wb[i];
      } else {
        if (a.g_scale) gs = *reinterpret_cast<const f32x2*>(a.g_scale + (int64_t)n * a.E + ch0);
        if (a.g_shift) gh = *reinterpret_cast<const f32x2*>(a.g_shift + (int64_t)n * a.E + ch0);
      }
      // LDS addressing: ONE address register per window row; the column j is an immediate offset (the e-tile pitch is a
      // compile-time constant).  A ragged last segment reads past its row into the next one / the zeroed slack behind the
      // tile: finite values that only feed discarded outputs.
      const unsigned char* et = es + (size_t)cp * 4 + (size_t)(w0 * S) * XDW_EPITCH;
      const int rowpitch = a.IWt * XDW_EPITCH;
      f32x2 win[K][IW];
      unsigned raw[S][IW];
      auto load_row = [&](int hr, unsigned (&r)[IW]) {                     // halo row hr
        const unsigned char* base = et + hr * rowpitch;
#pragma unroll
        for (int j = 0; j < IW; ++j) r[j] = *reinterpret_cast<const unsigned*>(base + j * XDW_EPITCH);
      };
      auto insert_row = [&](const unsigned (&r)[IW], f32x2 (&slot_)[IW]) {
#pragma unroll
        for (int j = 0; j < IW; ++j) slot_[j] = Pair<bf16_t>::up(r[j]);
      };
      const int hr_base = h0 * S;
      {
        unsigned r0[K][IW];
#pragma unroll
        for (int kh = 0; kh < K; ++kh) load_row(hr_base + kh, r0[kh]);
#pragma unroll
        for (int kh = 0; kh < K; ++kh) insert_row(r0[kh], win[kh % K]);
      }
      if (nrows > 1) {
#pragma unroll
        for (int s = 0; s < S; ++s) load_row(hr_base + K + s, raw[s]);
      }
      const int64_t orow0 = (((int64_t)z * a.Ho + oy0 + h0) * a.Wo + ox0 + w0) * a.E + ch0;
      unsigned graw[WGRAD ? WSEG : 1];
      auto load_g = [&](int t) {
        const bf16_t* base = a.dy + orow0 + (int64_t)t * a.Wo * a.E;
#pragma unroll
        for (int j = 0; j < WSEG; ++j) graw[j] = *reinterpret_cast<const unsigned*>(base + (int64_t)min(j, tw - 1 - w0) * a.E);
      };
      if (WGRAD) load_g(0);
      for (int tk = 0; tk < nrows; tk += K) {
#pragma unroll
        for (int r = 0; r < K; ++r) {
          const int t = tk + r;
          if (t < nrows) {
            f32x2 g[WGRAD ? WSEG : 1];
            if (WGRAD) {
#pragma unroll
              for (int j = 0; j < WSEG; ++j) g[j] = (w0 + j < tw) ? Pair<bf16_t>::up(graw[j]) * gs + gh : f32x2{0.f, 0.f};
            }
            if (t > 0) {
#pragma unroll
              for (int s = 0; s < S; ++s) insert_row(raw[s], win[(r * S + K - S + s) % K]);
            }
            if (t + 1 < nrows) {
#pragma unroll
              for (int s = 0; s < S; ++s) load_row(hr_base + (t + 1) * S + K - S + s, raw[s]);
              if (WGRAD) load_g(t + 1);
            }
            if (!WGRAD) {
              f32x2 acc[WSEG];
#pragma unroll
              for (int j = 0; j < WSEG; ++j) acc[j] = f32x2{0.f, 0.f};
#pragma unroll
              for (int kh = 0; kh < K; ++kh)
#pragma unroll
                for (int kw = 0; kw < K; ++kw) {
                  const f32x2 w2 = f32x2{wt[kh * K + kw], wt[KK + kh * K + kw]};
#pragma unroll
                  for (int j = 0; j < WSEG; ++j) pk_fma_keep(acc[j], win[(r * S + kh) % K][j * S + kw], w2);
                }
              bf16_t* orow = a.d + orow0 + (int64_t)t * a.Wo * a.E;
#pragma unroll
              for (int j = 0; j < WSEG; ++j) {
                if (w0 + j < tw) {
                  const unsigned o = Pair<bf16_t>::down(acc[j]);
                  *reinterpret_cast<unsigned*>(orow + (int64_t)j * a.E) = o;
                  psum += Pair<bf16_t>::up(o);
                }
              }
            } else {
#pragma unroll
              for (int kh = 0; kh < K; ++kh)
#pragma unroll
                for (int kw = 0; kw < K; ++kw)
#pragma unroll
                  for (int j = 0; j < WSEG; ++j) wacc[kh * K + kw] += g[j] * win[(r * S + kh) % K][j * S + kw];
            }
          }
        }
      }
    }
    // sums over the units that share a channel pair.  Inside a wave: a butterfly over the unit bits of the lane index (the
    // same value on every lane, the same association every run); across the four waves: LDS partials added in wave order
    // after the chunk's barrier (finish()).
    if (!WGRAD) {
      if (a.part) {
        f32x2 v = on ? psum : f32x2{0.f, 0.f};
        for (int off = cpc; off < 64; off <<= 1) {
          v[0] += __shfl_xor(v[0], off, 64);
          v[1] += __shfl_xor(v[1], off, 64);
        }
        if (lane < cpc) *reinterpret_cast<f32x2*>(red + ((size_t)wave * cpc + cp) * 2) = v;   // (consumer waves are 0 .. NC-1)
      }
    } else {
#pragma unroll
      for (int i = 0; i < KK; ++i) {
        f32x2 v = on ? wacc[i] : f32x2{0.f, 0.f};
        for (int off = cpc; off < 64; off <<= 1) {
          v[0] += __shfl_xor(v[0], off, 64);
          v[1] += __shfl_xor(v[1], off, 64);
        }
        if (lane < cpc) {
          red[((size_t)wave * cpc + cp) * 2 * KK + i] = v[0];
          red[((size_t)wave * cpc + cp) * 2 * KK + KK + i] = v[1];
        }
      }
    }
  };
  // after the barrier that ends chunk ci: add the four consumer waves' partials (wave order) and store
  auto finish = [&](int ci) {
    const int c0 = c_begin + ci * a.EC;
    const int ec = min(a.EC, c_end - c0);
    const float* red = reinterpret_cast<const float*>(smem + a.r_off + (size_t)(ci & 1) * a.r_size);
    const int per = WGRAD ? 2 * KK : 2;                 // floats per channel pair
    // consumer lanes cover cpc pairs per wave slot only when 64 % cpc == 0 (cpc in {8,16,32}); a wave's slot holds the pairs
    // of ITS lanes: with cpc <= 32 every wave sees every pair
    if (!WGRAD && !a.part) return;
    for (int i = rt; i < ec / 2 * per; i += XDW_CT) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < XDW_NC; ++w) s += red[(size_t)w * cpc * per + i];
      if (WGRAD) {
        wg_part[(int64_t)c0 * KK + i] = s;
      } else {
        a.part[((int64_t)slot * a.N + n) * a.E + c0 + i] = s;
      }
    }
  };

  // ================================================================= the pipeline
  // (the two roles are separate top-level branches with their own loops - the same number of barriers on both sides - so
  //  that the producers' 160 registers of W1 fragments are not live in the consumers' code and vice versa)
  lds_barrier();                                     // x tile (and slack) in place
  if (producer) {
    // step s: W1 of chunk s+2 starts flying (DMA), chunk s+1's e tile is computed from the W1 tile fetched a step ago, then the
    // wave waits for its own DMA - a whole MFMA phase after issuing it - and joins the barrier
    dma_w1(0);
    dma_w1(1);
    xdw_wait_dma();
    produce(0);
    lds_barrier();
    for (int ci = 0; ci < nch; ++ci) {
      dma_w1(ci + 2);                                  // into tile (ci & 1): chunk ci's, last read in the previous step
      if (ci + 1 < nch) produce(ci + 1);
      xdw_wait_dma();
      lds_barrier();
    }
  } else {
    lds_barrier();
    for (int ci = 0; ci < nch; ++ci) {
      consume(ci);
      lds_barrier();
      finish(ci);
    }
  }
}

// wave layout per variant (see xdw_kernel): the 3 x 3 forward runs 12 + 4 waves, everything else 6 + 2
constexpr int xdw_nc(int K, bool wgrad) { return (K == 3 && !wgrad) ? 12 : 6; }
constexpr int xdw_np(int K, bool wgrad) { return (K == 3 && !wgrad) ? 4 : 2; }

struct XdwPlan {
  XdwArgs a;
  int wseg;
  dim3 grid;
  size_t lds;
  int64_t slots;        // pool slots per clip (forward) / partial-gradient slots (weight gradient)
};

// segment widths: the fp32 window K x ((WSEG-1)*S + K) x 2 plus the in-flight rows must fit 256 registers (two workgroups
// per CU), like the narrow forms of dw_rows.hip
constexpr int wseg_of(int K, int S) { return K == 3 ? (S == 1 ? 7 : 4) : (S == 1 ? 4 : 2); }

// Tile / chunk geometry for a shape; false when the fused path does not take it.
bool xdw_plan(const pb_dwconv_shape* s, int Cin, bool wgrad, XdwPlan* pl) {
  if (s->kt != 1 || s->kh != s->kw || (s->kh != 3 && s->kh != 5)) return false;
  if (s->sh != s->sw || (s->sh != 1 && s->sh != 2)) return false;
  if (s->ph != s->kh / 2 || s->pw != s->kw / 2) return false;
  if (Cin % 8 || Cin < 8 || Cin > 32 * XDW_MAXKS || s->C % 8 || s->C < 16) return false;
  const int K = s->kh, S = s->sh, P = K / 2;
  if (s->Ho != (s->Hi + 2 * P - K) / S + 1 || s->Wo != (s->Wi + 2 * P - K) / S + 1) return false;
  if (s->To != (s->Ti + 2 * s->pt - 1) / s->st + 1) return false;
  XdwArgs& a = pl->a;
  a = XdwArgs{};
  a.N = s->N; a.Cin = Cin; a.E = s->C; a.Ti = s->Ti; a.Hi = s->Hi; a.Wi = s->Wi; a.To = s->To; a.Ho = s->Ho; a.Wo = s->Wo;
  a.t_mul = s->st; a.t_off = -s->pt;
  a.KS = (Cin + 31) / 32;
  const int wseg = wseg_of(K, S);
  pl->wseg = wseg;
  const int KK = K * K;
  // chunk width 16 / 32 / 64 channels (the consumer lanes run along cpc = EC/2 channel pairs, a divisor of the wave)
  if (a.E <= 16) a.EC = 16;
  else if (a.E <= 32) a.EC = 32;
  else {
    const double e64 = (double)a.E / (((a.E + 63) / 64) * 64), e32 = (double)a.E / (((a.E + 31) / 32) * 32);
    a.EC = e32 > e64 * 1.15 ? 32 : 64;
  }
  if (const char* ev = getenv("PB_XDW_EC")) { const int v = atoi(ev); if (v == 16 || v == 32 || v == 64) a.EC = v; }
  const int total_chunks = (a.E + a.EC - 1) / a.EC;
  a.xpitch = Cin * 2 + 16;
  const int cpc = a.EC / 2;
  const int XDW_NC = xdw_nc(K, wgrad), XDW_NP = xdw_np(K, wgrad), XDW_CT = XDW_NC * 64;
  const int units_max = XDW_CT / cpc;
  const int ncbw = (a.EC / 16 + XDW_NP - 1) / XDW_NP;
  a.a_wave = (ncbw * 16 * (Cin / 8) + 63) / 64 * 64 * 16;
  a.a_size = XDW_NP * a.a_wave;
  a.r_size = XDW_NC * cpc * (wgrad ? 2 * KK : 2) * 4;
  // workgroups along E are decided first (the depthwise weights of a workgroup's whole channel range sit in LDS): enough
  // workgroups for a few rounds of the chip even with whole-frame tiles
  int esplit = 1;
  {
    const int64_t frames = (int64_t)s->N * s->To;
    while (frames * esplit < 1024 && esplit * 2 <= total_chunks) esplit *= 2;
    if (const char* ev = getenv("PB_XDW_ESPLIT")) { const int v = atoi(ev); if (v >= 1 && v <= total_chunks) esplit = v; }
  }
  a.nchunk = (total_chunks + esplit - 1) / esplit;
  a.esplit = (total_chunks + a.nchunk - 1) / a.nchunk;
  a.w_size = wgrad ? 0 : (a.nchunk * a.EC * KK * 4 + 15) / 16 * 16;
  // tile: every (rows, columns) candidate that fits the LDS (one 8-wave workgroup per CU owns all of it) is priced - halo
  // recompute (MFMA work, e-tile writes, x re-reads), rolling-window refills of the stencil walk, idle or ragged stencil
  // lanes - and the cheapest is taken.  PB_XDW_TILE="rows,cols" overrides (A/B).
  int best_th = 0, best_tw = 0;
  double best_cost = 1e30;
  int force_th = 0, force_tw = 0;
  if (const char* ev = getenv("PB_XDW_TILE")) sscanf(ev, "%d,%d", &force_th, &force_tw);
  const size_t lds_cap = 158 * 1024;
  for (int tw = wseg; tw < s->Wo + wseg; tw += wseg) {
    const int twc = tw < s->Wo ? tw : s->Wo;
    const int nseg = (twc + wseg - 1) / wseg;
    if (nseg > units_max) break;
    for (int th = 1; th <= s->Ho; ++th) {
      const int iht = (th - 1) * S + K, iwt = (twc - 1) * S + K;
      const int hp = iht * iwt;
      const int npb = (hp + 15) / 16;
      const size_t lds = (size_t)(npb * 16) * a.xpitch + 2 * (size_t)(npb * 16 + XDW_SLACK) * XDW_EPITCH + 2 * a.a_size + a.w_size + 2 * a.r_size + 64;
      if (lds > lds_cap) break;
      if (force_th && (th != force_th || twc != (force_tw < s->Wo ? force_tw : s->Wo))) continue;
      const int ty_n = (s->Ho + th - 1) / th, tx_n = (s->Wo + twc - 1) / twc;
      int nbands = units_max / nseg;
      if (nbands > th) nbands = th;
      const int band_rows = (th + nbands - 1) / nbands;
      nbands = (th + band_rows - 1) / band_rows;
      const double halo = (double)hp * ty_n * tx_n / ((double)s->Ho * S * s->Wo * S);
      const double refill = (double)(band_rows * S + K - S) / (band_rows * S);
      const double busy = (double)(nseg * nbands * cpc) / (double)XDW_CT;     // share of the consumer threads with a unit
      const double ragged = (double)(ty_n * nbands * band_rows) * (tx_n * nseg * wseg) / ((double)s->Ho * s->Wo);
      const double produce = (0.10 * a.KS + 0.15) * halo * (64.0 / a.EC);    // relative to one stencil pass
      const double stencil = (0.75 + 0.25 * refill) * ragged / busy;
      const double xread = 0.3 * halo * Cin / a.E;                           // HBM: x tile re-reads against the d tile
      const double cost = (produce > stencil ? produce : stencil) + xread + 0.02 * ty_n * tx_n;
      if (cost < best_cost) { best_cost = cost; best_th = th; best_tw = twc; }
    }
  }
  if (!best_th) return false;
  a.TH = best_th; a.TW = best_tw;
  a.tiles_y = (s->Ho + a.TH - 1) / a.TH; a.tiles_x = (s->Wo + a.TW - 1) / a.TW;
  a.IHt = (a.TH - 1) * S + K; a.IWt = (a.TW - 1) * S + K; a.HP = a.IHt * a.IWt;
  a.nseg = (a.TW + wseg - 1) / wseg;
  a.nbands = units_max / a.nseg;
  if (a.nbands > a.TH) a.nbands = a.TH;
  a.band_rows = (a.TH + a.nbands - 1) / a.nbands;
  a.nbands = (a.TH + a.band_rows - 1) / a.band_rows;
  const int npb = (a.HP + 15) / 16;
  a.e_size = (npb * 16 + XDW_SLACK) * XDW_EPITCH;
  a.x_off = 0;
  a.e_off = npb * 16 * a.xpitch;
  a.a_off = a.e_off + 2 * a.e_size;
  a.w_off = a.a_off + 2 * a.a_size;
  a.r_off = a.w_off + a.w_size;
  pl->lds = (size_t)a.r_off + 2 * a.r_size;
  if (pl->lds > 160 * 1024) return false;
  const int64_t tiles = (int64_t)s->N * s->To * a.tiles_y * a.tiles_x;
  if (tiles > 0x7fffffff) return false;
  if (const char* ev = getenv("PB_XDW_DBG")) a.dbg = atoi(ev);
  pl->grid = dim3((unsigned)tiles, (unsigned)a.esplit);
  pl->slots = wgrad ? tiles : (int64_t)s->To * a.tiles_y * a.tiles_x;
  return true;
}

template <bool WGRAD>
int xdw_launch(const XdwPlan& pl, int K, int S, hipStream_t st) {
#define XDW_CASE(KK_, SS_, WW_)                                                                                       \
  if (K == KK_ && S == SS_) {                                                                                         \
    constexpr int NC_ = xdw_nc(KK_, WGRAD), NP_ = xdw_np(KK_, WGRAD);                                                 \
    auto kern = xdw_kernel<KK_, SS_, WW_, WGRAD, NC_, NP_>;                                                           \
    if (pl.lds > 64 * 1024)                                                                                           \
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds) != hipSuccess) return 2; \
    hipLaunchKernelGGL(kern, pl.grid, dim3((NC_ + NP_) * 64), pl.lds, st, pl.a);                                      \
    return 0;                                                                                                         \
  }
  XDW_CASE(3, 1, 7)
  XDW_CASE(3, 2, 4)
  XDW_CASE(5, 1, 4)
  XDW_CASE(5, 2, 2)
#undef XDW_CASE
  return 1;
}

}  // namespace

extern "C" int pb_xdw_supported(const pb_dwconv_shape* s, int Cin) {
  XdwPlan pl;
  return s && xdw_plan(s, Cin, false, &pl) ? 1 : 0;
}

extern "C" int pb_xdw_plan(const pb_dwconv_shape* s, int Cin, int wgrad, int* out12) {
  XdwPlan pl;
  if (!s || !out12 || !xdw_plan(s, Cin, wgrad != 0, &pl)) return 0;
  const XdwArgs& a = pl.a;
  const int v[12] = {a.TH, a.TW, a.EC, a.nchunk, a.esplit, (int)pl.lds, (int)pl.grid.x, a.nseg, a.nbands, a.band_rows, a.HP, a.KS};
  for (int i = 0; i < 12; ++i) out12[i] = v[i];
  return 1;
}

extern "C" int pb_xdw_pool_slots(const pb_dwconv_shape* s, int Cin) {
  XdwPlan pl;
  if (!s || !xdw_plan(s, Cin, false, &pl)) return 0;
  return (int)pl.slots;
}

extern "C" int pb_xdw_fwd(const void* x, const void* w1_bf16, const float* wdw, void* d, const pb_dwconv_shape* s, int Cin,
                          float* pool_part, int pool_slots, int device, void* stream) {
  PB_CHECK_ARG(x && w1_bf16 && wdw && d && s, "pb_xdw_fwd: null pointer");
  XdwPlan pl;
  PB_CHECK_ARG(xdw_plan(s, Cin, false, &pl), "pb_xdw_fwd: shape not supported by the fused expand+depthwise kernel");
  PB_CHECK_ARG(!pool_part || pool_slots == (int)pl.slots, "pb_xdw_fwd: pool_part needs %lld slots, got %d", (long long)pl.slots, pool_slots);
  PB_ENTER(device);
  pl.a.x = (const bf16_t*)x; pl.a.w1 = (const bf16_t*)w1_bf16; pl.a.wdw = wdw; pl.a.d = (bf16_t*)d; pl.a.part = pool_part;
  const int rc = xdw_launch<false>(pl, s->kh, s->sh, (hipStream_t)stream);
  PB_CHECK_ARG(rc == 0, "pb_xdw_fwd: launch setup failed (%d)", rc);
  PB_LAUNCH_CHECK();
  return 0;
}

extern "C" int64_t pb_xdw_bwd_weight_ws_bytes(const pb_dwconv_shape* s, int Cin) {
  XdwPlan pl;
  if (!s || !xdw_plan(s, Cin, true, &pl)) return 0;
  return pl.slots * (int64_t)s->C * s->kh * s->kw * 4;
}

extern "C" int pb_xdw_bwd_weight(const void* x, const void* w1_bf16, const void* dy, float* dwdw, const pb_dwconv_shape* s, int Cin,
                                 const float* g_scale, const float* g_shift, void* ws, int64_t ws_bytes, pb_reduce_job* defer,
                                 int device, void* stream) {
  PB_CHECK_ARG(x && w1_bf16 && dy && dwdw && s && ws, "pb_xdw_bwd_weight: null pointer");
  XdwPlan pl;
  PB_CHECK_ARG(xdw_plan(s, Cin, true, &pl), "pb_xdw_bwd_weight: shape not supported by the fused expand+depthwise kernel");
  const int64_t n = (int64_t)s->C * s->kh * s->kw;
  PB_CHECK_ARG(ws_bytes >= pl.slots * n * 4, "pb_xdw_bwd_weight: workspace of %lld bytes needed", (long long)(pl.slots * n * 4));
  PB_ENTER(device);
  pl.a.x = (const bf16_t*)x; pl.a.w1 = (const bf16_t*)w1_bf16; pl.a.dy = (const bf16_t*)dy; pl.a.part = (float*)ws;
  pl.a.g_scale = g_scale; pl.a.g_shift = g_shift;
  const int rc = xdw_launch<true>(pl, s->kh, s->sh, (hipStream_t)stream);
  PB_CHECK_ARG(rc == 0, "pb_xdw_bwd_weight: launch setup failed (%d)", rc);
  PB_LAUNCH_CHECK();
  pb_reduce_job job = {(const float*)ws, dwdw, n, n, (int)pl.slots, 1};
  if (defer) { *defer = job; return 0; }
  if (pb_launch_partials_reduce(&job, 1, (hipStream_t)stream)) { pb_set_error("pb_xdw_bwd_weight: reduce launch failed"); return 2; }
  return 0;
}
