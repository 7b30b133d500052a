// MobileViT patch attention (mobilevit.py:68-86 with the head merge restored):
// softmax(q k^T / sqrt(d)) v per (batch, intra-patch position, head); h = 8 heads of d = 16 fixed by
// MobileViTBlock (mobilevit.py:116), n = 32 ... 784 tokens, no mask, zero-padded tokens attended to.
//
// One workgroup per (group = (b,p), head).  d = 16 is one MFMA k-step, so the design goal is that the
// n x n probabilities never leave registers:
//   * scores are produced TRANSPOSED (S^T = K Q^T: key on the accumulator row, query on the lane), so a
//     16x16 accumulator tile is already the B operand of the next product that contracts over keys
//     (O^T = V^T P^T, dQ^T = K^T dS^T) - "accumulator tile as the next MFMA's operand";
//   * products that contract over queries (dV^T = dO^T P, dK^T = Q^T dS) use the un-transposed S = Q K^T
//     in a second kernel where waves own key tiles, so no gradient needs atomics;
//   * softmax statistics are lane-local (registers of one query column) + two shuffles;
//   * fp32 forward is two passes over the keys (online max/sum, then normalised P V), bf16 forward one pass;
//     backward recomputes P from the saved log-sum-exp.
// bf16: v_mfma_f32_16x16x32_bf16.  The d = 16 contractions fill only k-slots 0..15 of that instruction; the
// other half repeats the streamed operand against the LOW half of the pre-scaled register operand (the multiplier
// scale * log2 e with 16 mantissa bits, no multiply on the VALU), and the accumulator input carries the softmax's
// per-query constants (forward: minus the running maximum and the padding mask; backward: minus log-sum-exp and minus
// delta), so the subtractions, masks and clamps leave the VALU too - which is what bounds these kernels.  The key/query contractions use all 32
// slots: slot (g,j) <-> index 4g+j (j<4) / 16+4g+(j-4) of a 32-wide pair of tiles.
// fp32 (parity path): exact-f32 v_mfma_f32_16x16x4_f32, slot g of step r <-> index 4g+r of a tile.
#include "pb_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int HD = 16;        // head dim
constexpr int NH = 8;         // heads
constexpr int QKV_LD = 3 * NH * HD;   // 384
constexpr int OUT_LD = NH * HD;       // 128

struct AttnArgs {
  const void* qkv;     // [R][384]  q | k | v, each [h][d]
  void* out;           // fwd: [R][128]
  float* lse;          // [R][8]
  const void* dout;    // bwd: [R][128]
  const float* delta;  // bwd: [R][8]  rowsum(dO * O)
  void* dqkv;          // bwd: [R][384]
  int n, npad;         // tokens per group, rounded up to 32
  float scale;
  int split;           // workgroups per (group, head): each takes every split-th round of query (key) tiles
};

template <typename T> struct Frag;
template <> struct Frag<bf16_t> { bf16x8 v; };
template <> struct Frag<float> { float v[4]; };

// D = A[16 x d] * B[d x 16] over the d = 16 head dim.
//   bf16: lane (i = lane&15, g = lane>>4) of A holds A[i][8g..8g+7] for g < 2, zeros for g >= 2 (same for B^T)
//   fp32: lane holds A[i][4s+g] for s = 0..3
template <typename T>
__device__ __forceinline__ f32x4 mma_d16(const Frag<T>& a, const Frag<T>& b) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if constexpr (sizeof(T) == 2) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc, 0, 0, 0);
  } else {
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[s], b.v[s], acc, 0, 0, 0);
  }
  return acc;
}

// operand fragment of a [rows][16] row-major matrix (row pointer = start of this lane's row, or null -> zeros)
template <typename T>
__device__ __forceinline__ Frag<T> load_rowfrag(const T* row, int g) {
  Frag<T> f;
  if constexpr (sizeof(T) == 2) {
    bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    f.v = (row && g < 2) ? *reinterpret_cast<const bf16x8*>(row + 8 * g) : z;
  } else {
#pragma unroll
    for (int s = 0; s < 4; ++s) f.v[s] = row ? row[4 * s + g] : 0.f;
  }
  return f;
}

// cooperative staging of one [n][16] head slice of qkv/dout into LDS: row-major `rm` ([npad][16]) and,
// for bf16, transposed `tr` ([16][ldt]); rows >= n are zero
template <typename T>
__device__ __forceinline__ void stage_head(const T* src, int64_t ld, int n, int npad, T* rm, T* tr, int ldt) {
  constexpr int VEC = ElemTraits<T>::VEC;
  constexpr int CPR = HD / VEC;       // 16-byte chunks per row: 2 (bf16) / 4 (fp32)
  for (int i = threadIdx.x; i < npad * CPR; i += blockDim.x) {
    const int r = i / CPR, c = i - r * CPR;
    Vec16<T> v = vec_zero<T>();
    if (r < n) v = vec_load<T>(src + (int64_t)r * ld + c * VEC);
    if (rm) vec_store<T>(rm + r * HD + c * VEC, v);
    if (tr) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) tr[(c * VEC + e) * ldt + r] = from_f32<T>(v.get(e));
    }
  }
}

// ------------------------------------------------------------------------------------------ forward
template <typename T>
__global__ __launch_bounds__(512) void attn_fwd_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr bool BF = sizeof(T) == 2;
  const int n = a.n, npad = a.npad, ldt = npad + 8;
  T* Ks = (T*)smem;                                   // [npad][16]
  T* Vs = Ks + npad * HD;                             // bf16: V^T [16][ldt]; fp32: V [npad][16]
  const int gh = blockIdx.x / a.split, sp = blockIdx.x - gh * a.split, nwv = blockDim.x >> 6;
  const int grp = gh / NH, h = gh - grp * NH;
  const T* base = (const T*)a.qkv + (int64_t)grp * n * QKV_LD + h * HD;
  stage_head<T>(base + NH * HD, QKV_LD, n, npad, Ks, nullptr, 0);
  if constexpr (BF) stage_head<T>(base + 2 * NH * HD, QKV_LD, n, npad, nullptr, Vs, ldt);
  else stage_head<T>(base + 2 * NH * HD, QKV_LD, n, npad, Vs, nullptr, 0);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  const float NEG = -3.0e38f;
  // scores are kept in the log2 domain (scale * log2(e) folded into one multiplier, v_exp_f32 is exp2), and only the
  // last key tile, the one that can hold keys >= n, pays for the mask
  const float sl = a.scale * 1.44269504088896340736f;
  const int nfull = n >> 4;                                  // key tiles without padding keys
  for (int qt = sp * nwv + wave; qt * 16 < n; qt += nwv * a.split) {
    const int q = qt * 16 + li;
    const bool qok = q < n;
    const Frag<T> qf = load_rowfrag<T>(qok ? base + (int64_t)q * QKV_LD : nullptr, g);
    // pass 1: running max / sum over this lane's key rows (4g..4g+3 of every tile), merged over g afterwards
    float m = NEG, l = 0.f;
    auto pass1_tile = [&](int kt, auto masked) {
      const Frag<T> kf = load_rowfrag<T>(Ks + (kt * 16 + li) * HD, g);
      const f32x4 s = mma_d16<T>(kf, qf);
      float sv[4], mx = m;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sv[r] = (!decltype(masked)::value || kt * 16 + 4 * g + r < n) ? s[r] * sl : NEG;
        mx = fmaxf(mx, sv[r]);
      }
      float acc = l * __builtin_amdgcn_exp2f(m - mx);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc += __builtin_amdgcn_exp2f(sv[r] - mx);
      l = acc;
      m = mx;
    };
    for (int kt = 0; kt < nfull; ++kt) pass1_tile(kt, std::false_type{});
    for (int kt = nfull; kt * 16 < npad; ++kt) pass1_tile(kt, std::true_type{});
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
      const float m2 = __shfl_xor(m, off, 64), l2 = __shfl_xor(l, off, 64);
      const float mn = fmaxf(m, m2);
      l = l * __builtin_amdgcn_exp2f(m - mn) + l2 * __builtin_amdgcn_exp2f(m2 - mn);
      m = mn;
    }
    const float inv_l = 1.f / l;
    // pass 2: O^T[d][q] = sum_key V^T[d][key] * P^T[key][q]
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if constexpr (BF) {
      auto pass2_pair = [&](int kp, auto masked) {
        bf16x8 pb;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int kt = 2 * kp + half;
          const Frag<T> kf = load_rowfrag<T>(Ks + (kt * 16 + li) * HD, g);
          const f32x4 s = mma_d16<T>(kf, qf);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float p = __builtin_amdgcn_exp2f(s[r] * sl - m) * inv_l;
            if (decltype(masked)::value && !(kt * 16 + 4 * g + r < n)) p = 0.f;
            pb[4 * half + r] = (bf16_t)p;
          }
        }
        // A = V^T: lane (d = li, g) holds keys 32kp + 4g + j (j<4) and 32kp + 16 + 4g + (j-4)
        const bf16x4 v0 = *reinterpret_cast<const bf16x4*>(Vs + li * ldt + kp * 32 + 4 * g);
        const bf16x4 v1 = *reinterpret_cast<const bf16x4*>(Vs + li * ldt + kp * 32 + 16 + 4 * g);
        const bf16x8 va = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, pb, o, 0, 0, 0);
      };
      const int pfull = n >> 5;                              // key-tile pairs without padding keys
      for (int kp = 0; kp < pfull; ++kp) pass2_pair(kp, std::false_type{});
      for (int kp = pfull; kp * 32 < npad; ++kp) pass2_pair(kp, std::true_type{});
    } else {
      for (int kt = 0; kt * 16 < npad; ++kt) {
        const Frag<T> kf = load_rowfrag<T>(Ks + (kt * 16 + li) * HD, g);
        const f32x4 s = mma_d16<T>(kf, qf);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kt * 16 + 4 * g + r;
          const float p = (key < n) ? __builtin_amdgcn_exp2f(s[r] * sl - m) * inv_l : 0.f;
          o = __builtin_amdgcn_mfma_f32_16x16x4f32(Vs[key * HD + li], p, o, 0, 0, 0);   // A[d = li][slot g] = V[key][d]
        }
      }
    }
    if (qok) {
      // O^T tile: column = q (this lane), rows d = 4g + r
      T* op = (T*)a.out + ((int64_t)grp * n + q) * OUT_LD + h * HD + 4 * g;
#pragma unroll
      for (int r = 0; r < 4; ++r) op[r] = from_f32<T>(o[r]);
      if (g == 0) a.lse[((int64_t)grp * n + q) * NH + h] = m * 0.69314718055994530942f + __logf(l);
    }
  }
}

// ------------------------------------------------------------------------------------------ forward, one pass
// bf16 activations, ONE pass over the keys with an online softmax (the two-pass kernel above computed every score and
// every exp twice: 153 us at n = 768, batch 8 - 3-5 % of the dense MFMA peak).  Per wave: a tile of 16 queries; per
// step a PAIR of 16-key tiles: two S^T = K Q^T MFMAs, and the eight probabilities a lane holds ARE its slots of the B
// operand of O^T += V^T P^T (slot (g, r) <-> key 4g + r of the first tile, (g, 4 + r) <-> the second).  P never leaves
// registers.
//
// The loop is VALU-issue bound (per step and lane: 8 scores against 3 MFMAs), so the work per score is what counts,
// and the MFMA has two things to give away.  (1) d = 16 fills only k-slots 0..15 of the 16x16x32 contraction: slots
// 16..31 repeat the key values (the lanes of those slots re-read the row's two chunks) against the LOW half of the
// pre-scaled query, q * scale * log2(e) = hi + lo in bf16 - the scores arrive in the log2 domain with 16 mantissa
// bits of the multiplier instead of 8, and no multiply is left on the VALU.  (2) The accumulator input C: with
// C = -m[q] (the running maximum of the lane's query column, fp32) the MFMA returns x - m directly, and in the last
// step C also carries -1e30 for the padding keys - no subtraction and no mask selects.  The maximum is tracked lazily:
// a lane only asks whether any of its eight new scores is positive, i.e. above the reference (4 v_max3 + a compare);
// when some lane of the wave says yes - after the first few key tiles it rarely does - the column maximum is gathered
// (two shuffles), the accumulators and this step's scores are rescaled and C rewritten.  Per step: 56 -> 26 VALU
// instructions; 115 -> 84 us at n = 768, batch 8.
__device__ __forceinline__ void split_bf16(float x, bf16_t& hi, bf16_t& lo) {
  hi = (bf16_t)x;
  lo = (bf16_t)(x - (float)hi);
}

// B operand [hi | lo] of x * mult for a row of 16 bf16 values: lane group g holds elements 8 (g & 1) .. + 7 - the high
// halves for g < 2, the low halves for g >= 2 (null row -> zeros)
__device__ __forceinline__ bf16x8 load_scaled_hilo(const bf16_t* row, int g, float mult) {
  bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
  if (row) v = *reinterpret_cast<const bf16x8*>(row + 8 * (g & 1));
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    bf16_t hi, lo;
    split_bf16((float)v[e] * mult, hi, lo);
    v[e] = g < 2 ? hi : lo;
  }
  return v;
}

// 16 bytes global -> LDS without registers (lane i lands at lds_dst + 16 i); ordered by the issuing wave's vmcnt
__device__ __forceinline__ void attn_glds16(const void* gsrc, unsigned char* lds_dst) {
  typedef __attribute__((address_space(3))) unsigned char* lds_b;
  const unsigned dst = (unsigned)(uintptr_t)(lds_b)lds_dst;
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

// one [n][16] bf16 head slice (rows `ld` elements apart) -> LDS row-major [npad][16] at `dst`, by LDS-DMA; rows >= n repeat
// row n-1 (callers that need zeros there overwrite them after the wait).  All waves take part; the caller waits
// (s_waitcnt vmcnt(0)) and synchronises.
__device__ __forceinline__ void stage_slice_dma(const bf16_t* src, int64_t ld, int n, int npad, unsigned char* dst, int wave_u, int nwv, int lane) {
  const int nchunks = 2 * npad;                                 // 16-byte chunks (a multiple of 64)
  for (int c0 = wave_u * 64; c0 < nchunks; c0 += nwv * 64) {
    const int i = c0 + lane;
    const int r = min(i >> 1, n - 1), c = i & 1;
    attn_glds16(src + (int64_t)r * ld + c * 8, dst + (size_t)c0 * 16);
  }
}

// Round 4: K AND V are staged ROW-MAJOR ([npad][16], the layout they have in memory) by LDS-DMA - no registers, no
// transposing 2-byte LDS writes, every chunk of the slice in flight at once (staging was a third of the kernel:
// profiles/r04_attention_ablation.txt).  The V^T operand of O^T += V^T P^T comes from the hardware transposing read:
// within a 16-lane group, lane 4q+p supplies the address of row q (key 4g+q of the tile), columns 4p..4p+3, and lane i
// receives column i = V[4g..4g+3][d = i] - exactly the four k-slots (g, 0..3) the lane needs.  Rows >= n repeat row n-1
// (their scores carry -1e30: p = 0 against finite values).
__global__ __launch_bounds__(1024) void attn_fwd_onepass_bf16_kernel(AttnArgs a) {
  typedef bf16_t T;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int n = a.n, npad = a.npad;
  const int gh = blockIdx.x / a.split, sp = blockIdx.x - gh * a.split, nwv = blockDim.x >> 6;
  const int grp = gh / NH, h = gh - grp * NH;
  const T* base = (const T*)a.qkv + (int64_t)grp * n * QKV_LD + h * HD;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  // LDS: K [npad][16] | V [npad][16]
  T* Ks = (T*)smem;
  T* Vs = Ks + npad * HD;
  {
    const int nchunks = 2 * npad;                               // 16-byte chunks of one slice
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    for (int c0 = wave_u * 64; c0 < 2 * nchunks; c0 += nwv * 64) {    // K chunks, then V chunks (nchunks is a multiple of 64)
      const bool isv = c0 >= nchunks;
      const int i = c0 - (isv ? nchunks : 0) + lane;
      const int r = min(i >> 1, n - 1), c = i & 1;
      attn_glds16(base + (isv ? 2 : 1) * NH * HD + (int64_t)r * QKV_LD + c * 8, smem + (size_t)c0 * 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  const int npairs = npad >> 5;
  const float sl0 = a.scale * 1.44269504088896340736f;
  const T* krow = Ks + li * HD + 8 * (g & 1);                  // this lane's chunk of key row li of a tile
  typedef __attribute__((address_space(3))) bf16x4* lds_v4;
  const T* vtr = Vs + (4 * g + (li >> 2)) * HD + 4 * (li & 3);   // transposing read: row 4g + q, columns 4p.. of a tile
  // query tile t belongs to workgroup t % split, and within it to wave (t / split) % waves: every workgroup gets the same
  // number of tiles (+-1), and the waves with one tile more are spread evenly over the four SIMDs
  for (int qt = sp + a.split * wave; qt * 16 < n; qt += nwv * a.split) {
    const int q = qt * 16 + li;
    const bool qok = q < n;
    const bf16x8 qv = load_scaled_hilo(qok ? base + (int64_t)q * QKV_LD : nullptr, g, sl0);
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    f32x4 cm = {0.f, 0.f, 0.f, 0.f};                           // -m, the C operand of the score products
    float m = 0.f, l = 0.f;                                    // l: this lane's share (its 8 keys per pair) of the row sum
    auto step = [&](int kp, auto first, auto last) {
      const bf16x8 ka = *reinterpret_cast<const bf16x8*>(krow + kp * 32 * HD);
      const bf16x8 kb = *reinterpret_cast<const bf16x8*>(krow + kp * 32 * HD + 16 * HD);
      const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(vtr + kp * 32 * HD));
      const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(vtr + kp * 32 * HD + 16 * HD));
      f32x4 ca = cm, cb = cm;
      if constexpr (decltype(last)::value) {                   // the pair that can hold keys >= n
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          ca[r] = kp * 32 + 4 * g + r < n ? cm[r] : -1.0e30f;
          cb[r] = kp * 32 + 16 + 4 * g + r < n ? cm[r] : -1.0e30f;
        }
      }
      f32x4 sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, qv, ca, 0, 0, 0);             // x - m  (S^T[key][q])
      f32x4 sb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kb, qv, cb, 0, 0, 0);
      // (the maximum of the lane's eight scores stays a plain fmaxf chain: hand-written v_max3 instructions in inline asm read
      // the MFMA results before the hazard wait states the compiler inserts for its own instructions - NaNs, round 4)
      float mx = fmaxf(fmaxf(sa[0], sa[1]), sa[2]);
      mx = fmaxf(fmaxf(mx, sa[3]), sb[0]);
      mx = fmaxf(fmaxf(mx, sb[1]), sb[2]);
      mx = fmaxf(mx, sb[3]);
      if (decltype(first)::value || __any(mx > 0.f)) {
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));                // the four lane groups of a query column
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float delta = decltype(first)::value ? mx : fmaxf(mx, 0.f);   // the first reference may be negative
        if constexpr (!decltype(first)::value) {
          const float alpha = __builtin_amdgcn_exp2f(-delta);
          l *= alpha;
          o *= alpha;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { sa[r] -= delta; sb[r] -= delta; }
        m += delta;
        cm[0] = cm[1] = cm[2] = cm[3] = -m;
      }
      float p[8];
      bf16x8 pb;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        p[r] = __builtin_amdgcn_exp2f(sa[r]);
        p[4 + r] = __builtin_amdgcn_exp2f(sb[r]);
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) pb[r] = (bf16_t)p[r];
      l += ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
      const bf16x8 va = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, pb, o, 0, 0, 0);
    };
    // (Round 4 also tried NOT tracking the maximum when a Cauchy-Schwarz bound |q||k|max shows no score can overflow: 100 ->
    // 80 us at n = 768 - but the largest probability then no longer sits at exactly 1.0, every weight carries a bf16
    // rounding error, and MobileViT-XXS's eval-mode bf16 gradients moved from 0.90 to 1.29 of the reference's own deviation
    // (median; worst tensor 1.85 -> 2.99 of the 3.0 bar).  Parity first: the exact running maximum stays.)
    if (npairs == 1) {
      step(0, std::true_type{}, std::true_type{});
    } else {
      step(0, std::true_type{}, std::false_type{});
      for (int kp = 1; kp < npairs - 1; ++kp) step(kp, std::false_type{}, std::false_type{});
      step(npairs - 1, std::false_type{}, std::true_type{});
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.f / l;
    if (qok) {
      // O^T tile: column = q (this lane), rows d = 4g + r: 8 bytes per lane
      const bf16x4 ov = {(bf16_t)(o[0] * inv), (bf16_t)(o[1] * inv), (bf16_t)(o[2] * inv), (bf16_t)(o[3] * inv)};
      *reinterpret_cast<bf16x4*>((T*)a.out + ((int64_t)grp * n + q) * OUT_LD + h * HD + 4 * g) = ov;
      if (g == 0) a.lse[((int64_t)grp * n + q) * NH + h] = (m + __log2f(l)) * 0.69314718055994530942f;
    }
  }
}

// FP8 (BASELINE config 5, "CDNA4 fp8 MFMA attn"): the same one-pass scheme with the operands of both products in OCP
// e4m3 - K and V quantised per (group, head) with one scale each (amax over the head slice, found while staging), Q
// per query tile, P as p * 256 - fed to v_mfma_f32_16x16x32_fp8_fp8; softmax statistics stay fp32, the running maximum
// is kept the plain way (max, subtract and mask on the VALU).
typedef long fp8x8_t;

__device__ __forceinline__ fp8x8_t pack_fp8x8(const float (&v)[8]) {
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
  return (fp8x8_t)(((unsigned long)(unsigned)hi << 32) | (unsigned)lo);
}

__device__ __forceinline__ float block_max(float v, float* red) {       // max over the workgroup (order-independent)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  float m = red[0];
  for (int w = 1; w < nw; ++w) m = fmaxf(m, red[w]);
  __syncthreads();
  return m;
}

__global__ __launch_bounds__(1024) void attn_fwd_onepass_fp8_kernel(AttnArgs a) {
  typedef bf16_t T;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int n = a.n, npad = a.npad;
  const int gh = blockIdx.x / a.split, sp = blockIdx.x - gh * a.split, nwv = blockDim.x >> 6;
  const int grp = gh / NH, h = gh - grp * NH;
  const T* base = (const T*)a.qkv + (int64_t)grp * n * QKV_LD + h * HD;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  // LDS: K [npad][16] (16 B rows) | V^T [16][npad + 16] | 16 floats
  const int ldt = npad + 16;
  unsigned char* K8 = smem;
  unsigned char* V8 = smem + (size_t)npad * HD;
  float inv_sk = 1.f, inv_sv = 1.f;
  {
    // K and V head slices: registers -> amax -> one scale each -> e4m3 in LDS (rows >= n are zeros)
    constexpr int MAXC = 4;                                    // 16-byte chunks per thread and tensor: npad * 2 <= 4 * 512
    float* red = (float*)(smem + (size_t)npad * HD + (size_t)HD * ldt);
    bf16x8 kc[MAXC], vc[MAXC];
    float ak = 0.f, av = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int i = threadIdx.x + c * (int)blockDim.x;
      const int r = i >> 1, hf = i & 1;
      const bool ok = i < npad * 2 && r < n;
      bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      kc[c] = ok ? *reinterpret_cast<const bf16x8*>(base + NH * HD + (int64_t)r * QKV_LD + hf * 8) : z;
      vc[c] = ok ? *reinterpret_cast<const bf16x8*>(base + 2 * NH * HD + (int64_t)r * QKV_LD + hf * 8) : z;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        ak = fmaxf(ak, fabsf((float)kc[c][e]));
        av = fmaxf(av, fabsf((float)vc[c][e]));
      }
    }
    ak = block_max(ak, red);
    av = block_max(av, red);
    const float sk = ak > 0.f ? 448.f / ak : 1.f, sv = av > 0.f ? 448.f / av : 1.f;
    inv_sk = 1.f / sk; inv_sv = 1.f / sv;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int i = threadIdx.x + c * (int)blockDim.x;
      if (i < npad * 2) {
        const int r = i >> 1, hf = i & 1;
        float kf[8], vf[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { kf[e] = (float)kc[c][e] * sk; vf[e] = (float)vc[c][e] * sv; }
        *reinterpret_cast<fp8x8_t*>(K8 + (size_t)r * HD + hf * 8) = pack_fp8x8(kf);
        const fp8x8_t vq = pack_fp8x8(vf);
#pragma unroll
        for (int e = 0; e < 8; ++e) V8[(size_t)(hf * 8 + e) * ldt + r] = (unsigned char)((unsigned long)vq >> (8 * e));
      }
    }
    __syncthreads();
  }
  const float NEG = -3.0e38f;
  const int npairs = npad >> 5, pfull = n >> 5;               // key-tile pairs; pairs without padding keys
  for (int qt = sp + a.split * wave; qt * 16 < n; qt += nwv * a.split) {
    const int q = qt * 16 + li;
    const bool qok = q < n;
    bf16x8 qv = {0, 0, 0, 0, 0, 0, 0, 0};
    if (qok && g < 2) qv = *reinterpret_cast<const bf16x8*>(base + (int64_t)q * QKV_LD + 8 * g);
    // per query tile: amax -> scale -> e4m3; scale * log2(e) and the de-quantisation go into one score multiplier
    float aq = 0.f, qf[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) aq = fmaxf(aq, fabsf((float)qv[e]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) aq = fmaxf(aq, __shfl_xor(aq, o, 64));
    const float sq = aq > 0.f ? 448.f / aq : 1.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) qf[e] = (float)qv[e] * sq;
    const fp8x8_t q8 = pack_fp8x8(qf);
    const float sl = a.scale * 1.44269504088896340736f * inv_sk / sq;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    float m = NEG, l = 0.f;                                    // l: this lane's share (its 8 keys per pair) of the row sum
    auto step = [&](int kp, auto masked) {
      f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = sa;
      const fp8x8_t ka = g < 2 ? *reinterpret_cast<const fp8x8_t*>(K8 + (size_t)(kp * 32 + li) * HD + 8 * g) : 0;
      const fp8x8_t kb = g < 2 ? *reinterpret_cast<const fp8x8_t*>(K8 + (size_t)(kp * 32 + 16 + li) * HD + 8 * g) : 0;
      sa = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(ka, q8, sa, 0, 0, 0);
      sb = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(kb, q8, sb, 0, 0, 0);
      float sv8[8], mx = NEG;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sv8[r] = (!decltype(masked)::value || kp * 32 + 4 * g + r < n) ? sa[r] * sl : NEG;
        sv8[4 + r] = (!decltype(masked)::value || kp * 32 + 16 + 4 * g + r < n) ? sb[r] * sl : NEG;
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) mx = fmaxf(mx, sv8[r]);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));                  // the four lane groups of a query column
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      // after the first few key tiles the running maximum rarely moves: the rescale (one more exp and five multiplies
      // per lane) runs only when some lane of the wave raised its maximum
      if (__any(mx > m)) {
        const float mn = fmaxf(m, mx);
        const float alpha = __builtin_amdgcn_exp2f(m - mn);
        m = mn;
        l *= alpha;
        o *= alpha;
      }
      float p8[8], ps = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float p = __builtin_amdgcn_exp2f(sv8[r] - m);
        ps += p;
        p8[r] = p * 256.f;
      }
      l += ps;
      const fp8x8_t pb = pack_fp8x8(p8);
      const unsigned v0 = *reinterpret_cast<const unsigned*>(V8 + (size_t)li * ldt + kp * 32 + 4 * g);
      const unsigned v1 = *reinterpret_cast<const unsigned*>(V8 + (size_t)li * ldt + kp * 32 + 16 + 4 * g);
      const fp8x8_t va = (fp8x8_t)(((unsigned long)v1 << 32) | v0);
      o = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(va, pb, o, 0, 0, 0);
    };
    for (int kp = 0; kp < pfull; ++kp) step(kp, std::false_type{});
    for (int kp = pfull; kp < npairs; ++kp) step(kp, std::true_type{});
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = inv_sv * (1.f / 256.f) / l;
    if (qok) {
      const bf16x4 ov = {(bf16_t)(o[0] * inv), (bf16_t)(o[1] * inv), (bf16_t)(o[2] * inv), (bf16_t)(o[3] * inv)};
      *reinterpret_cast<bf16x4*>((T*)a.out + ((int64_t)grp * n + q) * OUT_LD + h * HD + 4 * g) = ov;
      if (g == 0) a.lse[((int64_t)grp * n + q) * NH + h] = (m + __log2f(l)) * 0.69314718055994530942f;
    }
  }
}

// delta[r][h] = sum_d dO[r][h][d] * O[r][h][d]
template <typename T>
__global__ __launch_bounds__(256) void attn_delta_kernel(const T* dout, const T* out, float* delta, int64_t R) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= R * NH) return;
  const T* a = dout + i * HD;
  const T* b = out + i * HD;
  float s = 0.f;
#pragma unroll
  for (int d = 0; d < HD; ++d) s += to_f32(a[d]) * to_f32(b[d]);
  delta[i] = s;
}

// ------------------------------------------------------------------------------------------ backward, fp32
// (the parity path: exact-f32 MFMAs, masks and subtractions spelled out; the bf16 kernels follow)
// dQ: waves own query tiles; S^T orientation (key on the accumulator row): dQ^T[d][q] = sum_key K^T[d][key] dS^T[key][q]
__global__ __launch_bounds__(512) void attn_bwd_dq_kernel(AttnArgs a) {
  typedef float T;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int n = a.n, npad = a.npad;
  T* Ks = (T*)smem;                 // [npad][16]
  T* Vs = Ks + npad * HD;           // [npad][16]
  const int gh = blockIdx.x / a.split, sp = blockIdx.x - gh * a.split, nwv = blockDim.x >> 6;
  const int grp = gh / NH, h = gh - grp * NH;
  const T* base = (const T*)a.qkv + (int64_t)grp * n * QKV_LD + h * HD;
  stage_head<T>(base + NH * HD, QKV_LD, n, npad, Ks, nullptr, 0);
  stage_head<T>(base + 2 * NH * HD, QKV_LD, n, npad, Vs, nullptr, 0);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  for (int qt = sp * nwv + wave; qt * 16 < n; qt += nwv * a.split) {
    const int q = qt * 16 + li;
    const bool qok = q < n;
    const int64_t row = (int64_t)grp * n + q;
    const Frag<T> qf = load_rowfrag<T>(qok ? base + (int64_t)q * QKV_LD : nullptr, g);
    const Frag<T> dof = load_rowfrag<T>(qok ? (const T*)a.dout + row * OUT_LD + h * HD : nullptr, g);
    const float L = qok ? a.lse[row * NH + h] : 0.f;
    const float dl = qok ? a.delta[row * NH + h] : 0.f;
    f32x4 dq = {0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt * 16 < npad; ++kt) {
      const Frag<T> kf = load_rowfrag<T>(Ks + (kt * 16 + li) * HD, g);
      const Frag<T> vf = load_rowfrag<T>(Vs + (kt * 16 + li) * HD, g);
      const f32x4 s = mma_d16<T>(kf, qf);        // S^T[key][q]
      const f32x4 dp = mma_d16<T>(vf, dof);      // dP^T[key][q] = sum_d V[key][d] dO[q][d]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 16 + 4 * g + r;
        const bool ok = qok && key < n;
        const float p = ok ? __expf(s[r] * a.scale - L) : 0.f;
        const float ds = p * (dp[r] - dl) * a.scale;
        dq = __builtin_amdgcn_mfma_f32_16x16x4f32(Ks[key * HD + li], ds, dq, 0, 0, 0);
      }
    }
    if (qok) {
      T* op = (T*)a.dqkv + row * QKV_LD + h * HD + 4 * g;
#pragma unroll
      for (int r = 0; r < 4; ++r) op[r] = dq[r];
    }
  }
}

// dK, dV: waves own key tiles; S orientation (query on the accumulator row, key on the lane):
//   dV^T[d][key] = sum_q dO^T[d][q] P[q][key],   dK^T[d][key] = sum_q Q^T[d][q] dS[q][key]
__global__ __launch_bounds__(1024) void attn_bwd_dkv_kernel(AttnArgs a) {
  typedef float T;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int n = a.n, npad = a.npad;
  T* Qs = (T*)smem;                      // [npad][16]
  T* Os = Qs + npad * HD;                // dO [npad][16]
  float* Ls = Os + npad * HD;            // lse [npad], delta [npad]
  float* Ds = Ls + npad;
  const int gh = blockIdx.x / a.split, sp = blockIdx.x - gh * a.split, nwv = blockDim.x >> 6;
  const int grp = gh / NH, h = gh - grp * NH;
  const T* base = (const T*)a.qkv + (int64_t)grp * n * QKV_LD + h * HD;
  const T* dob = (const T*)a.dout + (int64_t)grp * n * OUT_LD + h * HD;
  stage_head<T>(base, QKV_LD, n, npad, Qs, nullptr, 0);
  stage_head<T>(dob, OUT_LD, n, npad, Os, nullptr, 0);
  for (int i = threadIdx.x; i < npad; i += blockDim.x) {
    Ls[i] = i < n ? a.lse[((int64_t)grp * n + i) * NH + h] : 0.f;
    Ds[i] = i < n ? a.delta[((int64_t)grp * n + i) * NH + h] : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  for (int kt = sp * nwv + wave; kt * 16 < n; kt += nwv * a.split) {
    const int key = kt * 16 + li;
    const bool kok = key < n;
    const Frag<T> kf = load_rowfrag<T>(kok ? base + NH * HD + (int64_t)key * QKV_LD : nullptr, g);
    const Frag<T> vf = load_rowfrag<T>(kok ? base + 2 * NH * HD + (int64_t)key * QKV_LD : nullptr, g);
    f32x4 dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
    for (int qt = 0; qt * 16 < npad; ++qt) {
      const Frag<T> qf = load_rowfrag<T>(Qs + (qt * 16 + li) * HD, g);
      const Frag<T> dof = load_rowfrag<T>(Os + (qt * 16 + li) * HD, g);
      const f32x4 s = mma_d16<T>(qf, kf);        // S[q][key]: rows q = 4g + r, column = key
      const f32x4 dp = mma_d16<T>(dof, vf);      // dP[q][key]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = qt * 16 + 4 * g + r;
        const bool ok = kok && q < n;
        const float p = ok ? __expf(s[r] * a.scale - Ls[q]) : 0.f;
        const float ds = p * (dp[r] - Ds[q]) * a.scale;
        dv = __builtin_amdgcn_mfma_f32_16x16x4f32(Os[q * HD + li], p, dv, 0, 0, 0);
        dk = __builtin_amdgcn_mfma_f32_16x16x4f32(Qs[q * HD + li], ds, dk, 0, 0, 0);
      }
    }
    if (kok) {
      T* kp_ = (T*)a.dqkv + ((int64_t)grp * n + key) * QKV_LD + NH * HD + h * HD + 4 * g;
      T* vp_ = kp_ + NH * HD;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        kp_[r] = dk[r];
        vp_[r] = dv[r];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ backward, bf16
// The same two gifts of the MFMA as in the forward.  The scores are rebuilt with the forward's 16-bit multiplier
// (whichever operand of S = Q K^T sits in registers carries x * scale * log2(e) as [hi | lo]; the streamed operand's
// lanes of k-slots 16..31 re-read the row), and the per-query constants enter through the accumulator input:
//   S' = Q K^T - L2        (C = -lse * log2 e)         P  = exp2(S')
//   dP' = dO V^T - delta   (C = -delta)                dS = P * dP' * scale, `* scale` applied to the 16x16 result
// One exp and one multiply per score are left on the VALU (six operations before: scale, subtract, clamp, exp,
// subtract, two multiplies).  Both kernels were VALU-issue bound: 343 -> 214 us for the pair at n = 768, batch 8.

// dQ: waves own query tiles; S^T orientation (key on the accumulator row, the lane's own query on the column, so C is
// one value per lane).  A operands (K, V rows) stream from LDS, B operands (q' = [hi | lo], dO = [dO | 0]) are registers.
__global__ __launch_bounds__(512) void attn_bwd_dq_bf16_kernel(AttnArgs a) {
  typedef bf16_t T;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int n = a.n, npad = a.npad;
  T* Ks = (T*)smem;                 // [npad][16]
  T* Vs = Ks + npad * HD;           // [npad][16]   (round 4: no K^T copy - the transposing LDS read delivers it)
  const int gh = blockIdx.x / a.split, sp = blockIdx.x - gh * a.split, nwv = blockDim.x >> 6;
  const int grp = gh / NH, h = gh - grp * NH;
  const T* base = (const T*)a.qkv + (int64_t)grp * n * QKV_LD + h * HD;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  {
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    stage_slice_dma(base + NH * HD, QKV_LD, n, npad, (unsigned char*)Ks, wave_u, nwv, lane);
    stage_slice_dma(base + 2 * NH * HD, QKV_LD, n, npad, (unsigned char*)Vs, wave_u, nwv, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  const int li = lane & 15, g = lane >> 4;
  typedef __attribute__((address_space(3))) bf16x4* lds_v4;
  const T* ktr = Ks + (4 * g + (li >> 2)) * HD + 4 * (li & 3);   // transposing read: K^T[d = lane][keys 4g .. 4g+3] of a tile
  const T* krow = Ks + li * HD + 8 * (g & 1);
  const T* vrow = Vs + li * HD + 8 * (g & 1);
  const int npairs = npad >> 5, pfull = n >> 5;
  const float sl0 = a.scale * 1.44269504088896340736f;
  for (int qt = sp + a.split * wave; qt * 16 < n; qt += nwv * a.split) {
    const int q = qt * 16 + li;
    const bool qok = q < n;
    const int64_t row = (int64_t)grp * n + q;
    const bf16x8 qv = load_scaled_hilo(qok ? base + (int64_t)q * QKV_LD : nullptr, g, sl0);
    bf16x8 dov = {0, 0, 0, 0, 0, 0, 0, 0};
    if (qok && g < 2) dov = *reinterpret_cast<const bf16x8*>((const T*)a.dout + row * OUT_LD + h * HD + 8 * g);
    const float nl2 = qok ? -a.lse[row * NH + h] * 1.44269504088896340736f : 0.f;
    const float ndl = qok ? -a.delta[row * NH + h] : 0.f;
    const f32x4 cl = {nl2, nl2, nl2, nl2}, cd = {ndl, ndl, ndl, ndl};
    f32x4 dq = {0.f, 0.f, 0.f, 0.f};
    auto step = [&](int kp, auto masked) {
      const bf16x8 ka = *reinterpret_cast<const bf16x8*>(krow + kp * 32 * HD);
      const bf16x8 kb = *reinterpret_cast<const bf16x8*>(krow + kp * 32 * HD + 16 * HD);
      const bf16x8 va = *reinterpret_cast<const bf16x8*>(vrow + kp * 32 * HD);
      const bf16x8 vb = *reinterpret_cast<const bf16x8*>(vrow + kp * 32 * HD + 16 * HD);
      const f32x4 sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, qv, cl, 0, 0, 0);       // S'^T[key][q]
      const f32x4 sb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kb, qv, cl, 0, 0, 0);
      const f32x4 da = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, dov, cd, 0, 0, 0);      // dP'^T[key][q]
      const f32x4 db = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vb, dov, cd, 0, 0, 0);
      bf16x8 dsb;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float xa = __builtin_amdgcn_exp2f(sa[r]) * da[r], xb = __builtin_amdgcn_exp2f(sb[r]) * db[r];
        if constexpr (decltype(masked)::value) {
          // a padding key has K = V = 0, so S' = -L2: its weight is dropped here rather than trusted to stay finite
          xa = kp * 32 + 4 * g + r < n ? xa : 0.f;
          xb = kp * 32 + 16 + 4 * g + r < n ? xb : 0.f;
        }
        dsb[r] = (bf16_t)xa;
        dsb[4 + r] = (bf16_t)xb;
      }
      const bf16x4 k0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(ktr + kp * 32 * HD));
      const bf16x4 k1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(ktr + kp * 32 * HD + 16 * HD));
      const bf16x8 kt8 = {k0[0], k0[1], k0[2], k0[3], k1[0], k1[1], k1[2], k1[3]};
      dq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt8, dsb, dq, 0, 0, 0);
    };
    for (int kp = 0; kp < pfull; ++kp) step(kp, std::false_type{});
    for (int kp = pfull; kp < npairs; ++kp) step(kp, std::true_type{});
    if (qok) {
      const bf16x4 ov = {(bf16_t)(dq[0] * a.scale), (bf16_t)(dq[1] * a.scale), (bf16_t)(dq[2] * a.scale), (bf16_t)(dq[3] * a.scale)};
      *reinterpret_cast<bf16x4*>((T*)a.dqkv + row * QKV_LD + h * HD + 4 * g) = ov;
    }
  }
}

// dK, dV: waves own key tiles; S orientation (query on the accumulator row: C is four values per lane and step, read
// from LDS, where -L2 and -delta are staged once).  A operands (Q, dO rows) stream from LDS, B operands
// (k' = [hi | lo], V = [V | 0]) are registers.
__global__ __launch_bounds__(1024) void attn_bwd_dkv_bf16_kernel(AttnArgs a) {
  typedef bf16_t T;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int n = a.n, npad = a.npad;
  T* Qs = (T*)smem;                      // [npad][16]
  T* Os = Qs + npad * HD;                // dO [npad][16]   (round 4: no Q^T / dO^T copies - transposing LDS reads; 105 -> 55 KB
  float* Ls = (float*)(Os + npad * HD);  //  at n = 768: two workgroups per CU instead of one)    -lse * log2 e [npad], -delta [npad]
  float* Ds = Ls + npad;
  const int gh = blockIdx.x / a.split, sp = blockIdx.x - gh * a.split, nwv = blockDim.x >> 6;
  const int grp = gh / NH, h = gh - grp * NH;
  const T* base = (const T*)a.qkv + (int64_t)grp * n * QKV_LD + h * HD;
  const T* dob = (const T*)a.dout + (int64_t)grp * n * OUT_LD + h * HD;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  {
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    stage_slice_dma(base, QKV_LD, n, npad, (unsigned char*)Qs, wave_u, nwv, lane);
    stage_slice_dma(dob, OUT_LD, n, npad, (unsigned char*)Os, wave_u, nwv, lane);
  }
  for (int i = threadIdx.x; i < npad; i += blockDim.x) {
    Ls[i] = i < n ? -a.lse[((int64_t)grp * n + i) * NH + h] * 1.44269504088896340736f : 0.f;
    Ds[i] = i < n ? -a.delta[((int64_t)grp * n + i) * NH + h] : 0.f;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // the padding queries must be ZERO rows (Q = dO = 0 with L2 = delta = 0: P = 1, dS = 0 against zero dO^T / Q^T columns);
  // the DMA repeated row n-1 there
  for (int i = threadIdx.x; i < (npad - n) * 4; i += blockDim.x) {
    const int r = n + (i >> 2), c = i & 3;
    bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    *reinterpret_cast<bf16x8*>((c < 2 ? Qs : Os) + r * HD + (c & 1) * 8) = z;
  }
  __syncthreads();
  const int li = lane & 15, g = lane >> 4;
  typedef __attribute__((address_space(3))) bf16x4* lds_v4;
  const T* qtr = Qs + (4 * g + (li >> 2)) * HD + 4 * (li & 3);   // transposing reads: Q^T / dO^T[d = lane][queries 4g .. 4g+3]
  const T* otr = Os + (4 * g + (li >> 2)) * HD + 4 * (li & 3);
  const T* qrow = Qs + li * HD + 8 * (g & 1);
  const T* orow = Os + li * HD + 8 * (g & 1);
  const int npairs = npad >> 5;
  const float sl0 = a.scale * 1.44269504088896340736f;
  for (int kt = sp + a.split * wave; kt * 16 < n; kt += nwv * a.split) {
    const int key = kt * 16 + li;
    const bool kok = key < n;
    const bf16x8 kf = load_scaled_hilo(kok ? base + NH * HD + (int64_t)key * QKV_LD : nullptr, g, sl0);
    bf16x8 vf = {0, 0, 0, 0, 0, 0, 0, 0};
    if (kok && g < 2) vf = *reinterpret_cast<const bf16x8*>(base + 2 * NH * HD + (int64_t)key * QKV_LD + 8 * g);
    f32x4 dk = {0.f, 0.f, 0.f, 0.f}, dv = dk;
    // no masks: a padding query has Q = dO = 0 and L2 = delta = 0 (P = 1, dS = 0) and meets zero dO^T / Q^T columns; a
    // padding key's column is never stored
    for (int qp = 0; qp < npairs; ++qp) {
      const bf16x8 qa = *reinterpret_cast<const bf16x8*>(qrow + qp * 32 * HD);
      const bf16x8 qb = *reinterpret_cast<const bf16x8*>(qrow + qp * 32 * HD + 16 * HD);
      const bf16x8 oa = *reinterpret_cast<const bf16x8*>(orow + qp * 32 * HD);
      const bf16x8 ob = *reinterpret_cast<const bf16x8*>(orow + qp * 32 * HD + 16 * HD);
      const f32x4 la = *reinterpret_cast<const f32x4*>(Ls + qp * 32 + 4 * g), lb = *reinterpret_cast<const f32x4*>(Ls + qp * 32 + 16 + 4 * g);
      const f32x4 ea = *reinterpret_cast<const f32x4*>(Ds + qp * 32 + 4 * g), eb = *reinterpret_cast<const f32x4*>(Ds + qp * 32 + 16 + 4 * g);
      const f32x4 sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf, la, 0, 0, 0);       // S'[q][key]: rows q = 4g + r
      const f32x4 sb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qb, kf, lb, 0, 0, 0);
      const f32x4 da = __builtin_amdgcn_mfma_f32_16x16x32_bf16(oa, vf, ea, 0, 0, 0);       // dP'[q][key]
      const f32x4 db = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ob, vf, eb, 0, 0, 0);
      bf16x8 pb, dsb;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p0 = __builtin_amdgcn_exp2f(sa[r]), p1 = __builtin_amdgcn_exp2f(sb[r]);
        pb[r] = (bf16_t)p0;
        pb[4 + r] = (bf16_t)p1;
        dsb[r] = (bf16_t)(p0 * da[r]);
        dsb[4 + r] = (bf16_t)(p1 * db[r]);
      }
      const bf16x4 o0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(otr + qp * 32 * HD));
      const bf16x4 o1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(otr + qp * 32 * HD + 16 * HD));
      const bf16x8 ot8 = {o0[0], o0[1], o0[2], o0[3], o1[0], o1[1], o1[2], o1[3]};
      dv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ot8, pb, dv, 0, 0, 0);
      const bf16x4 q0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(qtr + qp * 32 * HD));
      const bf16x4 q1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4)(qtr + qp * 32 * HD + 16 * HD));
      const bf16x8 qt8 = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
      dk = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt8, dsb, dk, 0, 0, 0);
    }
    if (kok) {
      T* kp_ = (T*)a.dqkv + ((int64_t)grp * n + key) * QKV_LD + NH * HD + h * HD + 4 * g;
      const bf16x4 kv = {(bf16_t)(dk[0] * a.scale), (bf16_t)(dk[1] * a.scale), (bf16_t)(dk[2] * a.scale), (bf16_t)(dk[3] * a.scale)};
      const bf16x4 vv = {(bf16_t)dv[0], (bf16_t)dv[1], (bf16_t)dv[2], (bf16_t)dv[3]};
      *reinterpret_cast<bf16x4*>(kp_) = kv;
      *reinterpret_cast<bf16x4*>(kp_ + NH * HD) = vv;
    }
  }
}

// hipFuncSetAttribute costs tens of microseconds of HOST time: once per kernel and size, not per launch (an eager loop of
// attention launches was host-bound at ~90 us per launch, whatever the kernel did - tools/attn_bench.py, round 4)
template <typename KernelT>
int set_lds(KernelT k, size_t bytes) {
  if (bytes > 160 * 1024) {
    pb_set_error("attention: %zu bytes of LDS needed (n too large)", bytes);
    return 1;
  }
  static size_t have[8] = {0, 0, 0, 0, 0, 0, 0, 0};        // per device: the limit this kernel already has
  int dev = 0;
  PB_HIP(hipGetDevice(&dev));
  if (bytes > 48 * 1024 && (dev < 0 || dev >= 8 || have[dev] < bytes)) {
    PB_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    if (dev >= 0 && dev < 8) have[dev] = bytes;
  }
  return 0;
}

// 8 waves per workgroup (the staged K/V/Q/dO slices, up to 105 KB at n = 768, allow only 1-3 workgroups per CU) and
// enough workgroups per (group, head) to put ~1024 on the chip
constexpr int ATT_THREADS = 512, ATT_THREADS_DKV = 1024;
int attn_split(int64_t G, int n, int threads = ATT_THREADS) {
  static const int forced = getenv("PB_ATTN_SPLIT") ? atoi(getenv("PB_ATTN_SPLIT")) : 0;
  if (forced > 0) return forced;
  const int64_t tiles = pb_cdiv(n, 16), rounds = pb_cdiv(tiles, threads / 64);
  int64_t sp = pb_cdiv(1024, G * NH);
  if (G * NH >= 512) sp = 1;        // (every workgroup stages its head's K and V: once, when 512 heads already give 2 workgroups per CU)
  if (sp > rounds) sp = rounds;
  return sp < 1 ? 1 : (int)sp;
}

int onepass_threads(int n) {
  static const int forced = getenv("PB_ATTN_THREADS") ? atoi(getenv("PB_ATTN_THREADS")) : 0;
  if (forced == 256 || forced == 512 || forced == 1024) return forced;
  return n >= 512 ? 1024 : 512;
}

int check_attn(int64_t G, int n, pb_dtype dtype, const char* who) {
  PB_CHECK_ARG(G > 0 && n > 0 && G * NH * 64 < 0x7fffffffLL, "%s: bad sizes G=%lld n=%d", who, (long long)G, n);
  PB_CHECK_ARG(dtype == PB_F32 || dtype == PB_BF16, "%s: bad dtype", who);
  return 0;
}

}  // namespace

extern "C" int pb_attention_fwd(const void* qkv, void* out, float* lse, int64_t G, int n, float scale, pb_dtype dtype,
                                int device, void* stream) {
  PB_CHECK_ARG(qkv && out && lse, "pb_attention_fwd: null pointer");
  if (int rc = check_attn(G, n, dtype, "pb_attention_fwd")) return rc;
  PB_ENTER(device);
  AttnArgs a = {};
  a.qkv = qkv; a.out = out; a.lse = lse; a.n = n; a.npad = (n + 31) / 32 * 32; a.scale = scale;
  a.split = attn_split(G, n);
  const size_t es = dtype == PB_BF16 ? 2 : 4;
  const size_t lds = (size_t)a.npad * HD * es + (dtype == PB_BF16 ? (size_t)HD * (a.npad + 8) * es : (size_t)a.npad * HD * es);
  const size_t lds_onepass = (size_t)2 * a.npad * HD * 2;        // K and V row-major
  dim3 grid((unsigned)(G * NH * a.split));
  static const bool onepass = !(getenv("PB_ATTN_ONEPASS") && atoi(getenv("PB_ATTN_ONEPASS")) == 0);
  if (dtype == PB_BF16 && onepass) {
    // long sequences: 16 waves per workgroup, one workgroup per (group, head) - the K/V slices are staged once and every
    // round of workgroups fills the chip (512-thread workgroups at 3 per CU left a third of the second round empty)
    const int threads = onepass_threads(n);
    a.split = attn_split(G, n, threads);
    if (int rc = set_lds(attn_fwd_onepass_bf16_kernel, lds_onepass)) return rc;
    hipLaunchKernelGGL(attn_fwd_onepass_bf16_kernel, dim3((unsigned)(G * NH * a.split)), dim3(threads), lds_onepass, (hipStream_t)stream, a);
  } else if (dtype == PB_BF16) {
    if (int rc = set_lds(attn_fwd_kernel<bf16_t>, lds)) return rc;
    hipLaunchKernelGGL(attn_fwd_kernel<bf16_t>, grid, dim3(ATT_THREADS), lds, (hipStream_t)stream, a);
  } else {
    if (int rc = set_lds(attn_fwd_kernel<float>, lds)) return rc;
    hipLaunchKernelGGL(attn_fwd_kernel<float>, grid, dim3(ATT_THREADS), lds, (hipStream_t)stream, a);
  }
  PB_LAUNCH_CHECK();
  return 0;
}

// bf16 in / out, fp8 (OCP e4m3) operands in both MFMA products, fp32 softmax statistics: the forward of BASELINE's
// config 5 ("MobileViT-XXS fp8 forward ... CDNA4 fp8 MFMA attn").  Inference only (no backward form).
extern "C" int pb_attention_fwd_fp8(const void* qkv, void* out, float* lse, int64_t G, int n, float scale, int device,
                                    void* stream) {
  PB_CHECK_ARG(qkv && out && lse, "pb_attention_fwd_fp8: null pointer");
  if (int rc = check_attn(G, n, PB_BF16, "pb_attention_fwd_fp8")) return rc;
  PB_CHECK_ARG(n <= 1024, "pb_attention_fwd_fp8: n=%d exceeds the staged maximum of 1024 tokens", n);
  PB_ENTER(device);
  AttnArgs a = {};
  a.qkv = qkv; a.out = out; a.lse = lse; a.n = n; a.npad = (n + 31) / 32 * 32; a.scale = scale;
  const int threads = onepass_threads(n);
  a.split = attn_split(G, n, threads);
  const size_t lds = (size_t)a.npad * HD + (size_t)HD * (a.npad + 16) + 64;
  if (int rc = set_lds(attn_fwd_onepass_fp8_kernel, lds)) return rc;
  hipLaunchKernelGGL(attn_fwd_onepass_fp8_kernel, dim3((unsigned)(G * NH * a.split)), dim3(threads), lds, (hipStream_t)stream, a);
  PB_LAUNCH_CHECK();
  return 0;
}

extern "C" int pb_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* delta,
                                void* dqkv, int64_t G, int n, float scale, pb_dtype dtype, int device, void* stream) {
  PB_CHECK_ARG(qkv && out && dout && lse && delta && dqkv, "pb_attention_bwd: null pointer");
  if (int rc = check_attn(G, n, dtype, "pb_attention_bwd")) return rc;
  PB_ENTER(device);
  hipStream_t st = (hipStream_t)stream;
  AttnArgs a = {};
  a.qkv = qkv; a.dout = dout; a.lse = const_cast<float*>(lse); a.delta = delta; a.dqkv = dqkv; a.n = n; a.npad = (n + 31) / 32 * 32; a.scale = scale;
  a.split = attn_split(G, n);
  const int64_t R = G * n;
  const bool bf = dtype == PB_BF16;
  const size_t es = bf ? 2 : 4;
  const size_t rm = (size_t)a.npad * HD * es;                                              // one row-major slice
  dim3 grid((unsigned)(G * NH * a.split));
  AttnArgs akv = a;
  akv.split = attn_split(G, n, ATT_THREADS_DKV);
  dim3 grid_kv((unsigned)(G * NH * akv.split));
  if (bf) {
    hipLaunchKernelGGL(attn_delta_kernel<bf16_t>, dim3((unsigned)pb_cdiv(R * NH, 256)), dim3(256), 0, st, (const bf16_t*)dout, (const bf16_t*)out, delta, R);
    const size_t tr = (size_t)HD * (a.npad + 8) * es;                                      // one transposed slice
    const size_t lds_dkv = 2 * rm + 2 * (size_t)a.npad * 4, lds_dq = 2 * rm;     // (row-major slices only: transposing reads)
    if (int rc = set_lds(attn_bwd_dkv_bf16_kernel, lds_dkv)) return rc;
    hipLaunchKernelGGL(attn_bwd_dkv_bf16_kernel, grid_kv, dim3(ATT_THREADS_DKV), lds_dkv, st, akv);
    if (int rc = set_lds(attn_bwd_dq_bf16_kernel, lds_dq)) return rc;
    hipLaunchKernelGGL(attn_bwd_dq_bf16_kernel, grid, dim3(ATT_THREADS), lds_dq, st, a);
  } else {
    hipLaunchKernelGGL(attn_delta_kernel<float>, dim3((unsigned)pb_cdiv(R * NH, 256)), dim3(256), 0, st, (const float*)dout, (const float*)out, delta, R);
    const size_t lds_dq = 2 * rm, lds_dkv = 2 * rm + 2 * (size_t)a.npad * 4;
    if (int rc = set_lds(attn_bwd_dkv_kernel, lds_dkv)) return rc;
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, grid_kv, dim3(ATT_THREADS_DKV), lds_dkv, st, akv);
    if (int rc = set_lds(attn_bwd_dq_kernel, lds_dq)) return rc;
    hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, dim3(ATT_THREADS), lds_dq, st, a);
  }
  PB_LAUNCH_CHECK();
  return 0;
}
