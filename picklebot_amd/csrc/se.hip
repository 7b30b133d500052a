// Squeeze-excite gate (SEBlock3D, mobilenet.py:11-26): the two 1x1x1 convolutions on the pooled
// [N][C] vector are tiny GEMVs - one workgroup per clip, wavefront-shuffle dot products.
#include "pb_common.h"
#include <stdlib.h>

namespace {

constexpr int SE_MAX_C = 1 << 20;

// One wavefront per output element, lanes over the reduction index (coalesced weight rows,
// shuffle reduction): G*Cm + G*C wavefronts instead of G workgroups walking the weights serially.
//   MODE 0: hidden[g][j] = relu(b1[j] + W1[j,:] . (pool[g,:]*inv_count))          (out = Cm, red = C)
//   MODE 1: scale[g][c]  = hardsigmoid(b2[c] + W2[c,:] . hidden[g,:])              (out = C,  red = Cm)
template <int MODE>
__global__ __launch_bounds__(256) void se_gemv_kernel(const float* vec, float vscale, const float* w, const float* bias,
                                                      float* out, int G, int nout, int nred) {
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (wid >= (int64_t)G * nout) return;
  const int g = (int)(wid / nout), o = (int)(wid - (int64_t)g * nout);
  const float* wr = w + (int64_t)o * nred;
  const float* v = vec + (int64_t)g * nred;
  float s = 0.f;
  for (int c = lane; c < nred; c += 64) s += wr[c] * v[c];
  s = wave_sum(s) * vscale;
  if (lane == 0) {
    s += bias ? bias[o] : 0.f;
    out[wid] = MODE == 0 ? (s > 0.f ? s : 0.f) : fminf(fmaxf(s * (1.f / 6.f) + 0.5f, 0.f), 1.f);
  }
}

// Backward vectors.  The two matrix-transpose-vector products read the weight matrix along its
// contiguous dimension (lanes over outputs), split the reduction over the 4 waves of the workgroup
// with 4 independent accumulators each (16 loads in flight per lane), and combine in LDS:
//   out[g][o] = oscale * omask(g,o) * sum_r W[r][o] * v[g][r]
//   dh      (O = Cm, R = C ): W = W2 [C][Cm], v = dpre2,  omask = hidden[g][o] > 0
//   g_shift (O = C,  R = Cm): W = W1 [Cm][C], v = dh,     oscale = inv_count
constexpr int SE_RS = 16;     // reduction slices (waves) per workgroup: 60-240 dependent steps per thread with 4 were 12 us
__global__ __launch_bounds__(64 * SE_RS) void se_gemvT_kernel(const float* W, const float* v, int64_t vstride, float* out,
                                                              int64_t ostride, const float* omask_src, float oscale, int O,
                                                              int R) {
  __shared__ float part[SE_RS][64];
  const int ol = threadIdx.x & 63, rs = threadIdx.x >> 6;
  const int oblocks = (O + 63) / 64;
  const int g = blockIdx.x / oblocks, ob = blockIdx.x - g * oblocks;
  const int o = ob * 64 + ol;
  const float* vg = v + (int64_t)g * vstride;
  const int rq = (R + SE_RS - 1) / SE_RS, r0 = rs * rq, r1 = min(R, r0 + rq);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (o < O) {
    int r = r0;
    for (; r + 4 <= r1; r += 4) {
      s0 += W[(int64_t)r * O + o] * vg[r];
      s1 += W[(int64_t)(r + 1) * O + o] * vg[r + 1];
      s2 += W[(int64_t)(r + 2) * O + o] * vg[r + 2];
      s3 += W[(int64_t)(r + 3) * O + o] * vg[r + 3];
    }
    for (; r < r1; ++r) s0 += W[(int64_t)r * O + o] * vg[r];
  }
  part[rs][ol] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rs == 0 && o < O) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < SE_RS; ++i) s += part[i][ol];
    if (omask_src && !(omask_src[(int64_t)g * O + o] > 0.f)) s = 0.f;
    out[(int64_t)g * ostride + o] = s * oscale;
  }
}

// dpre2[g][c] = dscale * hardsigmoid'(.)  -> work[g][0..C)
__global__ __launch_bounds__(256) void se_dpre2_kernel(const float* dscale, const float* scale, float* work, int G, int C,
                                                       int Cm) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)G * C) return;
  const int g = (int)(idx / C), c = (int)(idx - (int64_t)g * C);
  const float sc = scale[idx];
  work[(int64_t)g * (C + Cm) + c] = (sc > 0.f && sc < 1.f) ? dscale[idx] * (1.f / 6.f) : 0.f;
}

// weight gradients: loops over the G clips, one thread per weight element (deterministic)
// (Co = outputs of the second FC: C for a squeeze-excite gate, the class count for the pooled head; act1 = 1: `hidden` holds
// the PRE-activation of a hardswish, act1 = 0: the ReLU's output)
__device__ __forceinline__ float se_hswish(float x) { return x * fminf(fmaxf(x + 3.f, 0.f), 6.f) * (1.f / 6.f); }
__global__ __launch_bounds__(256) void se_fc_bwd_w_kernel(const float* work, const float* pool, float inv_count,
                                                          const float* hidden, float* dw1, float* db1, float* dw2,
                                                          float* db2, int G, int C, int Cm, int Co, int act1) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t nW2 = (int64_t)Co * Cm, nW1 = (int64_t)C * Cm;
  const int stride = Co + Cm;
  if (idx < nW2) {                      // dW2[c][j] = sum_g dpre2[g][c] * h[g][j]
    const int c = (int)(idx / Cm), j = (int)(idx - (int64_t)c * Cm);
    float s = 0.f;
#pragma unroll 8
    for (int g = 0; g < G; ++g) {
      const float h = hidden[(int64_t)g * Cm + j];
      s += work[(int64_t)g * stride + c] * (act1 ? se_hswish(h) : h);
    }
    dw2[idx] += s;
  } else if (idx < nW2 + nW1) {         // dW1[j][c] = sum_g dh[g][j] * mean[g][c]
    const int64_t i = idx - nW2;
    const int j = (int)(i / C), c = (int)(i - (int64_t)j * C);
    float s = 0.f;
#pragma unroll 8
    for (int g = 0; g < G; ++g) s += work[(int64_t)g * stride + Co + j] * pool[(int64_t)g * C + c] * inv_count;
    dw1[i] += s;
  } else if (idx < nW2 + nW1 + Co) {    // db2
    const int c = (int)(idx - nW2 - nW1);
    float s = 0.f;
#pragma unroll 8
    for (int g = 0; g < G; ++g) s += work[(int64_t)g * stride + c];
    if (db2) db2[c] += s;
  } else if (idx < nW2 + nW1 + Co + Cm) {   // db1
    const int j = (int)(idx - nW2 - nW1 - Co);
    float s = 0.f;
#pragma unroll 8
    for (int g = 0; g < G; ++g) s += work[(int64_t)g * stride + Co + j];
    if (db1) db1[j] += s;
  }
}


// ---------------------------------------------------------------------------------------------- fused forms
// Two launches forward (was 4: slot reduction, two GEMVs, weight scaling) and three backward (was 6), every one with
// hundreds of workgroups: a single workgroup per clip was tried first and is latency-bound (116 us for C = 960).
// Every sum is added in a fixed order.
struct SeFwdArgs {
  const float* part;      // pool partial sums [slots][G][C] (pb_dwconv3d_fwd / pb_rowgroup_partials)
  int slots;
  float inv_count;
  const float* w1; const float* b1; const float* w2; const float* b2;
  float* pool;            // out [G][C]: sum over the slots (kept for backward)
  float* scale;           // out [G][C]
  float* hidden;          // out [G][Cm]
  const float* wproj;     // optional fp32 [N][C]: the projection conv's weights ...
  bf16_t* wbs;            // ... scaled per clip for the LDS-DMA GEMM: out bf16 [G][N][C] = wproj[n][c] * scale[g][c]
  int G, C, Cm, N;
  int assoc;              // slot-sum association (slot_sum_to_lds)
  int Co;                 // outputs of the second FC (= C for a squeeze-excite gate; the class count for the pooled head)
  int act1, act2;         // 0 / 0: ReLU, hardsigmoid (squeeze-excite); 1: hardswish with `hidden` = its PRE-activation; act2 = 1: none
  bf16_t* scale_bf16;     // optional bf16 copy of `scale` (the head's logits)
};

// sum over the slots of part[.][g][c] for c = tid, tid + nthreads, ... -> dst (LDS).  Eight loads in flight per thread:
// with 4 the loop was a chain of ~50 memory round trips (20 us per launch).
// `assoc` selects one of two equally valid associations of the fp32 sum (PB_SE_ASSOC, read per call by the launchers - a test
// knob: the parity suite runs the train-mode fixtures under both, so no fixture can depend on one summation order).
__device__ __forceinline__ void slot_sum_to_lds(const float* part, int slots, int G, int g, int C, float* dst, int nthreads, int assoc = 0) {
  const int n8 = slots & ~7;
  for (int c = threadIdx.x; c < C; c += nthreads) {
    const float* p = part + (int64_t)g * C + c;
    const int64_t st = (int64_t)G * C;
    // Eight chains (chain k adds the slots k, k+8, ... of the full groups of eight, in order); assoc 0: the tail of fewer than
    // eight slots goes to chain 0, then ((0+1)+(2+3))+((4+5)+(6+7)); assoc 1: the tail goes to chain 7 and the chains are
    // added left to right.  Sixteen loads are in flight per round and the tail's loads are issued together (a tail loop of
    // single loads was one memory round trip per slot).
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int sl = 0; sl < n8; sl += 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = sl + u < n8 ? p[(sl + u) * st] : 0.f;
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[u & 7] += v[u];           // (adding 0.f to a finite sum leaves its bits alone)
    }
    float t[7];
#pragma unroll
    for (int u = 0; u < 7; ++u) t[u] = n8 + u < slots ? p[(n8 + u) * st] : 0.f;
    if (assoc == 0) {
#pragma unroll
      for (int u = 0; u < 7; ++u) acc[0] += t[u];
      dst[c] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    } else {
#pragma unroll
      for (int u = 0; u < 7; ++u) acc[7] += t[u];
      dst[c] = ((((((acc[0] + acc[1]) + acc[2]) + acc[3]) + acc[4]) + acc[5]) + acc[6]) + acc[7];
    }
  }
}

constexpr int SE_HT = 1024;   // threads of se_hidden_kernel
constexpr int SE_JB = 64;     // hidden outputs per workgroup (16 waves x 4, computed together)

// hidden[g][j] = relu(b1[j] + inv_count * W1[j,:] . pool[g,:]), pool = slot sum.  grid (ceil(Cm/64), G), 1024 threads.
__global__ __launch_bounds__(SE_HT) void se_hidden_kernel(SeFwdArgs a) {
  extern __shared__ float sm[];                 // pool[C]
  const int g = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C = a.C;
  slot_sum_to_lds(a.part, a.slots, a.G, g, C, sm, SE_HT, a.assoc);
  __syncthreads();
  if (blockIdx.x == 0)
    for (int c = threadIdx.x; c < C; c += SE_HT) a.pool[(int64_t)g * C + c] = sm[c];
  const int j0 = blockIdx.x * SE_JB + wave * 4;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  // four column steps per round: 16 weight loads in flight per lane (one step per round was ~11 dependent round trips)
  for (int cb = 0; cb < C; cb += 256) {
    float wv[4][4], pv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = cb + u * 64 + lane;
      pv[u] = c < C ? sm[c] : 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) wv[u][q] = (c < C && j0 + q < a.Cm) ? a.w1[(int64_t)(j0 + q) * C + c] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int q = 0; q < 4; ++q) s[q] += wv[u][q] * pv[u];
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float v = wave_sum(s[q]) * a.inv_count + ((a.b1 && j0 + q < a.Cm) ? a.b1[j0 + q] : 0.f);
    if (lane == 0 && j0 + q < a.Cm) a.hidden[(int64_t)g * a.Cm + j0 + q] = a.act1 ? v : (v > 0.f ? v : 0.f);
  }
}

// scale[g][c] = hardsigmoid(b2[c] + W2[c,:] . hidden[g,:]) for a block of 64 channels, then the block's columns of the
// per-clip scaled bf16 projection weights.  grid (ceil(C/64), G), 256 threads.
__global__ __launch_bounds__(256) void se_scale_kernel(SeFwdArgs a) {
  extern __shared__ float sm[];                 // hidden[Cm] | scale block[64]
  const int g = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C = a.C, Cm = a.Cm, Co = a.Co;
  float* hid = sm;
  float* scl = sm + Cm;
  for (int j = tid; j < Cm; j += 256) {
    const float h = a.hidden[(int64_t)g * Cm + j];
    hid[j] = a.act1 ? se_hswish(h) : h;
  }
  __syncthreads();
  const int cb = blockIdx.x * 64;
  {
    // the wave's 16 channels together: 16 weight loads in flight per lane and step of 64 hidden units
    const int c0 = cb + wave * 16;
    float sacc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) sacc[q] = 0.f;
    for (int j = lane; j < Cm; j += 64) {
      const float h = hid[j];
      float wv[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) wv[q] = c0 + q < Co ? a.w2[(int64_t)(c0 + q) * Cm + j] : 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) sacc[q] += wv[q] * h;
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      float v = wave_sum(sacc[q]) + ((a.b2 && c0 + q < Co) ? a.b2[c0 + q] : 0.f);
      if (a.act2 == 0) v = fminf(fmaxf(v * (1.f / 6.f) + 0.5f, 0.f), 1.f);
      if (lane == 0) {
        scl[wave * 16 + q] = v;
        if (c0 + q < Co) {
          a.scale[(int64_t)g * Co + c0 + q] = v;
          if (a.scale_bf16) a.scale_bf16[(int64_t)g * Co + c0 + q] = (bf16_t)v;
        }
      }
    }
  }
  if (!a.wbs) return;
  __syncthreads();
  // this clip's scaled bf16 copy of the projection weights, columns cb .. cb+63 ((X * s) W^T = X (W * s)^T)
  const int chunks = min(64, C - cb) / 8;       // C % 8 == 0
  bf16_t* out = a.wbs + (int64_t)g * a.N * C;
  for (int i0 = tid; i0 < a.N * chunks; i0 += 4 * 256) {
    f32x4 w0[4], w1v[4];
    int64_t e[4];
    int chv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {                // all eight 16-byte loads of a round in flight
      const int i = min(i0 + u * 256, a.N * chunks - 1);
      const int n = i / chunks;
      chv[u] = i - n * chunks;
      e[u] = (int64_t)n * C + cb + chv[u] * 8;
      w0[u] = *reinterpret_cast<const f32x4*>(a.wproj + e[u]);
      w1v[u] = *reinterpret_cast<const f32x4*>(a.wproj + e[u] + 4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (i0 + u * 256 < a.N * chunks) {
        bf16x8 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          o[q] = (bf16_t)(w0[u][q] * scl[chv[u] * 8 + q]);
          o[4 + q] = (bf16_t)(w1v[u][q] * scl[chv[u] * 8 + 4 + q]);
        }
        *reinterpret_cast<bf16x8*>(out + e[u]) = o;
      }
    }
  }
}

struct SeBwdArgs {
  const float* dpart;     // dscale partial sums [slots][G][C]
  int slots;
  float inv_count;
  const float* scale; const float* hidden; const float* w1; const float* w2;
  float* g_shift;         // out [G][C]
  float* work;            // out [G][C + Cm]: dpre2 | dh (read by the weight-gradient kernel)
  int G, C, Cm;
  int assoc;
  int Co, act1, act2;     // as in SeFwdArgs (dpart / scale / the first Co entries of a `work` row have Co columns)
  bf16_t* g_shift_bf16;   // optional bf16 copy of g_shift (the head's pooled-input gradient row)
};

constexpr int SEB_T = 1024, SEB_W = SEB_T / 64;

// dpre2[g][c] = hardsigmoid'(.) * (slot sum of dscale); dh[g][j] = [hidden > 0] * sum_c W2[c][j] * dpre2[c] for a block of
// 64 outputs j: lanes over j (W2 rows are contiguous in j), the 16 waves split c.  grid (ceil(Cm/64), G), 1024 threads.
__global__ __launch_bounds__(SEB_T) void se_bwd_dh_kernel(SeBwdArgs a) {
  extern __shared__ float sm[];                 // dpre2[C] | partials [16][64]
  const int g = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C = a.Co, Cm = a.Cm;                // (this kernel only sees the second FC: its Co outputs)
  float* dpre2 = sm;
  float* red = sm + C;
  slot_sum_to_lds(a.dpart, a.slots, a.G, g, C, dpre2, SEB_T, a.assoc);
  __syncthreads();
  for (int c = tid; c < C; c += SEB_T) {
    float d = dpre2[c];
    if (a.act2 == 0) {
      const float sc = a.scale[(int64_t)g * C + c];
      d = (sc > 0.f && sc < 1.f) ? d * (1.f / 6.f) : 0.f;
    }
    dpre2[c] = d;
    if (blockIdx.x == 0) a.work[(int64_t)g * (C + Cm) + c] = d;
  }
  __syncthreads();
  const int j = blockIdx.x * 64 + lane;
  const int cq = (C + SEB_W - 1) / SEB_W, c0 = wave * cq, c1 = min(C, c0 + cq);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (j < Cm) {
    // 16 weight loads in flight per round (predicated tail), four interleaved chains
    for (int c = c0; c < c1; c += 16) {
      float wv[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) wv[u] = c + u < c1 ? a.w2[(int64_t)(c + u) * Cm + j] : 0.f;
#pragma unroll
      for (int u = 0; u < 16; u += 4) {
        s0 += wv[u] * (c + u < c1 ? dpre2[c + u] : 0.f);
        s1 += wv[u + 1] * (c + u + 1 < c1 ? dpre2[c + u + 1] : 0.f);
        s2 += wv[u + 2] * (c + u + 2 < c1 ? dpre2[c + u + 2] : 0.f);
        s3 += wv[u + 3] * (c + u + 3 < c1 ? dpre2[c + u + 3] : 0.f);
      }
    }
  }
  red[wave * 64 + lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (wave == 0 && j < Cm) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < SEB_W; ++w) s += red[w * 64 + lane];
    const float h = a.hidden[(int64_t)g * Cm + j];
    if (a.act1) s *= h <= -3.f ? 0.f : (h >= 3.f ? 1.f : (2.f * h + 3.f) * (1.f / 6.f));      // hardswish'(pre-activation)
    else if (!(h > 0.f)) s = 0.f;
    a.work[(int64_t)g * (C + Cm) + C + j] = s;
  }
}

// g_shift[g][c] = inv_count * sum_j W1[j][c] * dh[g][j] for a block of 64 channels: lanes over c, the 16 waves split j.
__global__ __launch_bounds__(SEB_T) void se_bwd_gshift_kernel(SeBwdArgs a) {
  extern __shared__ float sm[];                 // dh[Cm] | partials [16][64]
  const int g = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C = a.C, Cm = a.Cm;
  float* dh = sm;
  float* red = sm + Cm;
  for (int j = tid; j < Cm; j += SEB_T) dh[j] = a.work[(int64_t)g * (a.Co + Cm) + a.Co + j];
  __syncthreads();
  const int c = blockIdx.x * 64 + lane;
  const int jq = (Cm + SEB_W - 1) / SEB_W, j0 = wave * jq, j1 = min(Cm, j0 + jq);
  float s0 = 0.f, s1 = 0.f;
  if (c < C) {
    for (int j = j0; j < j1; j += 8) {            // 8 weight loads in flight per round (predicated tail)
      float wv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) wv[u] = j + u < j1 ? a.w1[(int64_t)(j + u) * C + c] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; u += 2) {
        s0 += wv[u] * (j + u < j1 ? dh[j + u] : 0.f);
        s1 += wv[u + 1] * (j + u + 1 < j1 ? dh[j + u + 1] : 0.f);
      }
    }
  }
  red[wave * 64 + lane] = s0 + s1;
  __syncthreads();
  if (wave == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < SEB_W; ++w) t += red[w * 64 + lane];
    a.g_shift[(int64_t)g * C + c] = t * a.inv_count;
    if (a.g_shift_bf16) a.g_shift_bf16[(int64_t)g * C + c] = (bf16_t)(t * a.inv_count);
  }
}

}  // namespace

// PB_SE_ASSOC (test knob, read on every call): which association slot_sum_to_lds uses
static int se_assoc() {
  const char* e = getenv("PB_SE_ASSOC");
  return e && e[0] == '1' ? 1 : 0;
}

extern "C" int pb_se_fc_fwd(const float* pool, float inv_count, const float* w1, const float* b1, const float* w2,
                            const float* b2, float* scale, float* hidden, int G, int C, int Cm, int device,
                            void* stream) {
  PB_CHECK_ARG(pool && w1 && w2 && scale && hidden, "pb_se_fc_fwd: null pointer");
  PB_CHECK_ARG(G > 0 && C > 0 && Cm > 0 && C <= SE_MAX_C && Cm <= SE_MAX_C / 4, "pb_se_fc_fwd: bad sizes C=%d Cm=%d", C, Cm);
  PB_ENTER(device);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(se_gemv_kernel<0>, dim3((unsigned)pb_cdiv((int64_t)G * Cm, 4)), dim3(256), 0, st, pool, inv_count,
                     w1, b1, hidden, G, Cm, C);
  hipLaunchKernelGGL(se_gemv_kernel<1>, dim3((unsigned)pb_cdiv((int64_t)G * C, 4)), dim3(256), 0, st,
                     (const float*)hidden, 1.f, w2, b2, scale, G, C, Cm);
  PB_LAUNCH_CHECK();
  return 0;
}

extern "C" int pb_se_fc_bwd(const float* dscale, const float* pool, float inv_count, const float* scale,
                            const float* hidden, const float* w1, const float* w2, float* g_shift, float* dw1,
                            float* db1, float* dw2, float* db2, float* work, int G, int C, int Cm, int device,
                            void* stream) {
  PB_CHECK_ARG(dscale && pool && scale && hidden && w1 && w2 && g_shift && dw1 && dw2 && work, "pb_se_fc_bwd: null pointer");
  PB_CHECK_ARG(G > 0 && C > 0 && Cm > 0 && C <= SE_MAX_C && Cm <= SE_MAX_C / 4, "pb_se_fc_bwd: bad sizes C=%d Cm=%d", C, Cm);
  PB_ENTER(device);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(se_dpre2_kernel, dim3((unsigned)pb_cdiv((int64_t)G * C, 256)), dim3(256), 0, st, dscale, scale, work,
                     G, C, Cm);
  hipLaunchKernelGGL(se_gemvT_kernel, dim3((unsigned)(G * pb_cdiv(Cm, 64))), dim3(64 * SE_RS), 0, st, w2, (const float*)work,
                     (int64_t)(C + Cm), work + C, (int64_t)(C + Cm), hidden, 1.f, Cm, C);
  hipLaunchKernelGGL(se_gemvT_kernel, dim3((unsigned)(G * pb_cdiv(C, 64))), dim3(64 * SE_RS), 0, st, w1,
                     (const float*)(work + C), (int64_t)(C + Cm), g_shift, (int64_t)C, (const float*)nullptr, inv_count,
                     C, Cm);
  const int64_t total = 2LL * C * Cm + C + Cm;
  hipLaunchKernelGGL(se_fc_bwd_w_kernel, dim3((unsigned)pb_cdiv(total, 256)), dim3(256), 0, st, work, pool,
                     inv_count, hidden, dw1, db1, dw2, db2, G, C, Cm, C, 0);
  PB_LAUNCH_CHECK();
  return 0;
}

// Fused squeeze-excite gate: pool partial sums -> pool, hidden, scale (and, optionally, the per-clip scaled bf16
// projection weights the LDS-DMA GEMM reads) in ONE launch.  pool_part: fp32 [slots][G][C].
// Fused squeeze-excite gate: pool partial sums -> pool, hidden, scale (and, optionally, the per-clip scaled bf16
// projection weights the LDS-DMA GEMM reads) in two launches.  pool_part: fp32 [slots][G][C].
extern "C" int pb_se_fused_fwd(const float* pool_part, int slots, float inv_count, const float* w1, const float* b1,
                               const float* w2, const float* b2, float* pool, float* scale, float* hidden,
                               const float* wproj, void* wbs_bf16, int N, int G, int C, int Cm, int device, void* stream) {
  PB_CHECK_ARG(pool_part && w1 && w2 && pool && scale && hidden && slots >= 1, "pb_se_fused_fwd: null pointer / bad slots");
  PB_CHECK_ARG(G > 0 && G <= 65535 && C > 0 && Cm > 0 && C <= 8192 && Cm <= 8192, "pb_se_fused_fwd: bad sizes G=%d C=%d Cm=%d", G, C, Cm);
  PB_CHECK_ARG((wproj == nullptr) == (wbs_bf16 == nullptr), "pb_se_fused_fwd: wproj and wbs go together");
  PB_CHECK_ARG(!wproj || (N > 0 && C % 8 == 0 && ((uintptr_t)wproj % 16) == 0 && ((uintptr_t)wbs_bf16 % 16) == 0),
               "pb_se_fused_fwd: scaled weights need C %% 8 == 0 and 16-byte aligned pointers");
  PB_ENTER(device);
  hipStream_t st = (hipStream_t)stream;
  SeFwdArgs a = {pool_part, slots, inv_count, w1, b1, w2, b2, pool, scale, hidden, wproj, (bf16_t*)wbs_bf16, G, C, Cm, N, se_assoc(), C, 0, 0, nullptr};
  hipLaunchKernelGGL(se_hidden_kernel, dim3((unsigned)pb_cdiv(Cm, SE_JB), (unsigned)G), dim3(SE_HT), (size_t)C * 4, st, a);
  hipLaunchKernelGGL(se_scale_kernel, dim3((unsigned)pb_cdiv(C, 64), (unsigned)G), dim3(256), ((size_t)Cm + 64) * 4, st, a);
  PB_LAUNCH_CHECK();
  return 0;
}

// Fused backward: dscale partial sums [slots][G][C] -> dh, g_shift, then the weight gradients (3 launches).
extern "C" int pb_se_fused_bwd(const float* dscale_part, int slots, const float* pool, float inv_count, const float* scale,
                               const float* hidden, const float* w1, const float* w2, float* g_shift, float* dw1,
                               float* db1, float* dw2, float* db2, float* work, int G, int C, int Cm, int device,
                               void* stream) {
  PB_CHECK_ARG(dscale_part && pool && scale && hidden && w1 && w2 && g_shift && work && slots >= 1 && (!dw1 == !dw2),
               "pb_se_fused_bwd: null pointer / bad slots");
  PB_CHECK_ARG(G > 0 && G <= 65535 && C > 0 && Cm > 0 && C <= 8192 && Cm <= 8192, "pb_se_fused_bwd: bad sizes G=%d C=%d Cm=%d", G, C, Cm);
  PB_ENTER(device);
  hipStream_t st = (hipStream_t)stream;
  SeBwdArgs a = {dscale_part, slots, inv_count, scale, hidden, w1, w2, g_shift, work, G, C, Cm, se_assoc(), C, 0, 0, nullptr};
  hipLaunchKernelGGL(se_bwd_dh_kernel, dim3((unsigned)pb_cdiv(Cm, 64), (unsigned)G), dim3(SEB_T), ((size_t)C + SEB_T) * 4, st, a);
  hipLaunchKernelGGL(se_bwd_gshift_kernel, dim3((unsigned)pb_cdiv(C, 64), (unsigned)G), dim3(SEB_T), ((size_t)Cm + SEB_T) * 4, st, a);
  if (dw1) {      // (dw1 == dw2 == NULL: the weight gradients are left to pb_se_fc_bwd_w - off the data-gradient chain)
    const int64_t total = 2LL * C * Cm + C + Cm;
    hipLaunchKernelGGL(se_fc_bwd_w_kernel, dim3((unsigned)pb_cdiv(total, 256)), dim3(256), 0, st, (const float*)work, pool,
                       inv_count, hidden, dw1, db1, dw2, db2, G, C, Cm, C, 0);
  }
  PB_LAUNCH_CHECK();
  return 0;
}

// The weight / bias gradients of the two squeeze-excite FCs from the `work` vectors pb_se_fused_bwd left ([G][C + Cm]: dz2, dh):
// nothing on the data-gradient chain reads them, so a caller can launch this later / on another stream.
extern "C" int pb_se_fc_bwd_w(const float* work, const float* pool, float inv_count, const float* hidden, float* dw1, float* db1,
                              float* dw2, float* db2, int G, int C, int Cm, int device, void* stream) {
  PB_CHECK_ARG(work && pool && hidden && dw1 && dw2, "pb_se_fc_bwd_w: null pointer");
  PB_CHECK_ARG(G > 0 && G <= 65535 && C > 0 && Cm > 0 && C <= 8192 && Cm <= 8192, "pb_se_fc_bwd_w: bad sizes G=%d C=%d Cm=%d", G, C, Cm);
  PB_ENTER(device);
  const int64_t total = 2LL * C * Cm + C + Cm;
  hipLaunchKernelGGL(se_fc_bwd_w_kernel, dim3((unsigned)pb_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, work, pool,
                     inv_count, hidden, dw1, db1, dw2, db2, G, C, Cm, C, 0);
  PB_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------- the pooled classifier head
// AdaptiveAvgPool3d(1) -> Conv3d(C, Cm, 1) -> Hardswish / ReLU -> Conv3d(Cm, Co, 1) (mobilenet.py classifier) is the squeeze-
// excite gate's machine with other activations and Co != C: the same kernels (fp32 weights, fixed-order sums), 3 launches
// forward and 2 + 1 backward where the generic path ran 8 and ~15 launches of ~5 us on [32 x 1280] tensors.
extern "C" int pb_pool_head_fwd(const float* pool_part, int slots, float inv_count, const float* w1, const float* b1, const float* w2,
                                const float* b2, pb_act act1, float* pool, float* logits, void* logits_bf16, float* pre1, int G, int C,
                                int Cm, int Co, int device, void* stream) {
  PB_CHECK_ARG(pool_part && w1 && w2 && pool && logits && pre1 && slots >= 1, "pb_pool_head_fwd: null pointer / bad slots");
  PB_CHECK_ARG(act1 == PB_ACT_HSWISH || act1 == PB_ACT_RELU, "pb_pool_head_fwd: the hidden activation must be hardswish or ReLU");
  PB_CHECK_ARG(G > 0 && G <= 65535 && C > 0 && Cm > 0 && Co > 0 && C <= 8192 && Cm <= 8192 && Co <= 8192, "pb_pool_head_fwd: bad sizes G=%d C=%d Cm=%d Co=%d", G, C, Cm, Co);
  PB_ENTER(device);
  hipStream_t st = (hipStream_t)stream;
  SeFwdArgs a = {pool_part, slots, inv_count, w1, b1, w2, b2, pool, logits, pre1, nullptr, nullptr, G, C, Cm, 0, se_assoc(), Co,
                 act1 == PB_ACT_HSWISH ? 1 : 0, 1, (bf16_t*)logits_bf16};
  hipLaunchKernelGGL(se_hidden_kernel, dim3((unsigned)pb_cdiv(Cm, SE_JB), (unsigned)G), dim3(SE_HT), (size_t)C * 4, st, a);
  hipLaunchKernelGGL(se_scale_kernel, dim3((unsigned)pb_cdiv(Co, 64), (unsigned)G), dim3(256), ((size_t)Cm + 64) * 4, st, a);
  PB_LAUNCH_CHECK();
  return 0;
}

// dlogits fp32 [G][Co] -> dpool_row [G][C] (= the gradient of EVERY pooled row: dlogits through both FCs, times inv_count) and
// `work` [G][Co + Cm] (dlogits | dpre1) for pb_pool_head_bwd_w.
extern "C" int pb_pool_head_bwd(const float* dlogits, float inv_count, const float* pre1, const float* w1, const float* w2, pb_act act1,
                                float* dpool_row, void* dpool_row_bf16, float* work, int G, int C, int Cm, int Co, int device,
                                void* stream) {
  PB_CHECK_ARG(dlogits && pre1 && w1 && w2 && dpool_row && work, "pb_pool_head_bwd: null pointer");
  PB_CHECK_ARG(act1 == PB_ACT_HSWISH || act1 == PB_ACT_RELU, "pb_pool_head_bwd: the hidden activation must be hardswish or ReLU");
  PB_CHECK_ARG(G > 0 && G <= 65535 && C > 0 && Cm > 0 && Co > 0 && C <= 8192 && Cm <= 8192 && Co <= 8192, "pb_pool_head_bwd: bad sizes G=%d C=%d Cm=%d Co=%d", G, C, Cm, Co);
  PB_ENTER(device);
  hipStream_t st = (hipStream_t)stream;
  SeBwdArgs a = {dlogits, 1, inv_count, nullptr, pre1, w1, w2, dpool_row, work, G, C, Cm, se_assoc(), Co, act1 == PB_ACT_HSWISH ? 1 : 0, 1,
                 (bf16_t*)dpool_row_bf16};
  hipLaunchKernelGGL(se_bwd_dh_kernel, dim3((unsigned)pb_cdiv(Cm, 64), (unsigned)G), dim3(SEB_T), ((size_t)Co + SEB_T) * 4, st, a);
  hipLaunchKernelGGL(se_bwd_gshift_kernel, dim3((unsigned)pb_cdiv(C, 64), (unsigned)G), dim3(SEB_T), ((size_t)Cm + SEB_T) * 4, st, a);
  PB_LAUNCH_CHECK();
  return 0;
}

// dw1 [Cm][C], db1 [Cm], dw2 [Co][Cm], db2 [Co] += ... from `work` (off the data-gradient chain, like pb_se_fc_bwd_w)
extern "C" int pb_pool_head_bwd_w(const float* work, const float* pool, float inv_count, const float* pre1, pb_act act1, float* dw1,
                                  float* db1, float* dw2, float* db2, int G, int C, int Cm, int Co, int device, void* stream) {
  PB_CHECK_ARG(work && pool && pre1 && dw1 && dw2, "pb_pool_head_bwd_w: null pointer");
  PB_CHECK_ARG(G > 0 && G <= 65535 && C > 0 && Cm > 0 && Co > 0 && C <= 8192 && Cm <= 8192 && Co <= 8192, "pb_pool_head_bwd_w: bad sizes");
  PB_ENTER(device);
  const int64_t total = (int64_t)Co * Cm + (int64_t)C * Cm + Co + Cm;
  hipLaunchKernelGGL(se_fc_bwd_w_kernel, dim3((unsigned)pb_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, work, pool,
                     inv_count, pre1, dw1, db1, dw2, db2, G, C, Cm, Co, act1 == PB_ACT_HSWISH ? 1 : 0);
  PB_LAUNCH_CHECK();
  return 0;
}
