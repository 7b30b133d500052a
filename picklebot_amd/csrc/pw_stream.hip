// Entry point and tile choice of the weight-stationary streaming GEMM (kernel: pw_stream_impl.h); the instantiations with
// BatchNorm statistics are compiled in pw_stream_stats.hip.
#include "pw_stream_impl.h"

__attribute__((visibility("hidden"))) int pb_pw_stream_stats_launch(const void* args, int ks, int nt, int cg, int device, hipStream_t st);

namespace {

// (column tile, column groups per workgroup) for a shape, or false when the streaming kernel does not take it
bool stream_choice(int64_t M, int K, int N, bool stats, int* nt_out, int* cg_out, bool one_column = false) {
  static const int on = getenv("PB_GEMM_STREAM") ? atoi(getenv("PB_GEMM_STREAM")) : 1;
  static const int force_cg = getenv("PB_GEMM_STREAM_CG") ? atoi(getenv("PB_GEMM_STREAM_CG")) : 0;
  if (!on || K > 160 || K % 8 || N % 8 || M < 1024) return false;
  const int ks = (K + 31) / 32;
  // the MFMA work is padded to tiles*BN columns and to wgc*CG tiles; the rows are fetched by LDS-DMA once per workgroup
  // column; and a grid of 8*wgc*q workgroups should cover the 256 CUs
  int best_nt = 0, best_cg = 0;
  double best_cost = 1e30;
  const int64_t groups = pb_cdiv(pb_cdiv(M, 128), 8);
  for (int cg = 1; cg <= 8; cg *= 2) {
    if (force_cg && cg != force_cg) continue;
    for (int c = (cg == 1 ? 1 : 4); c <= kMaxNT[stats][ks]; ++c) {
      const int64_t tiles = pb_cdiv(N, c * 16);
      const int64_t wgc = pb_cdiv(tiles, cg);
      if (8 * wgc > 256 || (one_column && (wgc != 1 || tiles > cg))) continue;
      int64_t q = 256 / (8 * wgc);
      if (q > groups) q = groups;
      const double occupancy = (double)(8 * wgc * q) / 256.0;
      const double mfma = (double)(wgc * cg * c * 16) / N;                  // padded MFMA columns per real column
      const double dma = (double)wgc * K / (K + N);                           // row re-fetches relative to the traffic
      const double fixed = cg == 1 ? 0.5 : 0.0;                               // per-stage overhead of the 16-row-per-wave layout
      const double cost = (0.6 * mfma + 0.8 * dma + fixed + 1.0) / occupancy;
      if (cost < best_cost) { best_cost = cost; best_nt = c; best_cg = cg; }
    }
  }
  if (!best_nt) return false;
  *nt_out = best_nt;
  *cg_out = best_cg;
  return true;
}

}  // namespace

__attribute__((visibility("hidden"))) int pb_pw_stream_wide_takes(int64_t M, int K, int N, int groups, int64_t rows_per_group);

extern "C" int pb_pwconv_fwd_bf16w_path(int64_t M, int K, int N, int groups, int64_t rows_per_group, int has_stats) {
  int nt, cg;
  if (groups == 1 && stream_choice(M, K, N, has_stats != 0, &nt, &cg)) return 1;
  return pb_pw_stream_wide_takes(M, K, N, groups, rows_per_group) ? 2 : 0;
}

// -> 0 launched, > 0 error, -1 shape not taken (the caller falls back to the ring kernel)
__attribute__((visibility("hidden"))) int pb_pw_stream_fwd(const void* x, const void* wb, void* y, int64_t M, int K, int N,
                                                           const float* out_scale, const float* out_shift, pb_act out_act,
                                                           float out_slope, float* stats, const void* residual, int device,
                                                           hipStream_t st, int stat_fold) {
  int best_nt, best_cg;
  if (!stream_choice(M, K, N, stats != nullptr, &best_nt, &best_cg, stat_fold > 1)) return -1;
  const int ks = (K + 31) / 32;
  StreamArgs a = {};
  a.X = (const bf16_t*)x; a.WB = (const bf16_t*)wb; a.Y = (bf16_t*)y; a.R = (const bf16_t*)residual;
  a.M = M; a.K = K; a.N = N;
  a.out_scale = out_scale; a.out_shift = out_shift; a.out_act = (int)out_act; a.out_slope = out_slope; a.stats = stats;
  a.stat_fold = stat_fold;
  a.n_tiles = (int)pb_cdiv(N, best_nt * 16);
  return stats ? pb_pw_stream_stats_launch(&a, ks, best_nt, best_cg, device, st) : dispatch_ks<false>(a, ks, best_nt, best_cg, device, st);
}
