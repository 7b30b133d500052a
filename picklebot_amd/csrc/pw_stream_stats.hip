// The weight-stationary streaming GEMM with per-column statistics in its epilogue (pw_stream_impl.h), in its own
// translation unit so that the two halves of the instantiation matrix compile side by side.
#include "pw_stream_impl.h"

__attribute__((visibility("hidden"))) int pb_pw_stream_stats_launch(const void* args, int ks, int nt, int cg, int device, hipStream_t st) {
  return dispatch_ks<true>(*(const StreamArgs*)args, ks, nt, cg, device, st);
}
