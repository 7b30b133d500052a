// Frame gather / scatter for temporally strided bottlenecks (mobilenet.py:67-75: the depthwise conv's scalar stride also
// strides T, with a (1,k,k) kernel - an output frame reads exactly ONE input frame, so with stride 2 half of the input
// frames are never read).  The reference expands all frames through pointwise_conv1; here the used frames are gathered
// first and the expansion conv, the depthwise conv and their backward passes run on half the rows; the input gradient is
// scattered back with exact zeros in the unused frames (what the reference computes for them).
#include "pb_common.h"

namespace {

// gather : dst[n, j]  = src[n, t0 + j * tstep]                                   (j < Tdst)
// scatter: dst[n, t]  = src[n, (t - t0) / tstep] when that is a frame of src, else the fill row (zeros when fill is null)
//          repeated over the frame (t < Tdst); fill = one pixel row of cvec 16-byte vectors
template <bool SCATTER>
__global__ __launch_bounds__(256) void frames_kernel(const u32x4* src, u32x4* dst, int Tsrc, int Tdst, int64_t fv, int t0, int tstep,
                                                     const u32x4* fill, int cvec) {
  const int z = blockIdx.y;                          // n * Tdst + t
  const int n = z / Tdst, t = z - n * Tdst;
  u32x4* out = dst + (int64_t)z * fv;
  const u32x4* in = nullptr;
  if (!SCATTER) {
    in = src + ((int64_t)n * Tsrc + t0 + (int64_t)t * tstep) * fv;
  } else {
    const int r = t - t0;
    if (r >= 0 && r % tstep == 0 && r / tstep < Tsrc) in = src + ((int64_t)n * Tsrc + r / tstep) * fv;
  }
  if (in || !fill) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < fv; i += (int64_t)gridDim.x * 256)
      out[i] = in ? in[i] : u32x4{0u, 0u, 0u, 0u};
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < fv; i += (int64_t)gridDim.x * 256) out[i] = fill[i % cvec];
  }
}

}  // namespace

extern "C" int pb_frames_gather(const void* src, void* dst, int N, int Tsrc, int Tdst, int64_t frame_bytes, int t0, int tstep,
                                int scatter, const void* fill, int fill_bytes, int device, void* stream) {
  PB_CHECK_ARG(src && dst && N > 0 && Tsrc > 0 && Tdst > 0 && frame_bytes > 0 && frame_bytes % 16 == 0 && tstep >= 1 && t0 >= 0,
               "pb_frames_gather: bad arguments");
  PB_CHECK_ARG(scatter || t0 + (int64_t)(Tdst - 1) * tstep < Tsrc, "pb_frames_gather: gathered frames reach past the source");
  PB_CHECK_ARG((int64_t)N * Tdst <= 65535, "pb_frames_gather: too many frames");
  PB_CHECK_ARG(!fill || (scatter && fill_bytes > 0 && fill_bytes % 16 == 0 && frame_bytes % fill_bytes == 0),
               "pb_frames_gather: the fill row must be a multiple of 16 bytes that divides the frame (scatter only)");
  const int cvec = fill ? fill_bytes / 16 : 1;
  PB_ENTER(device);
  const int64_t fv = frame_bytes / 16;
  int64_t gx = pb_cdiv(fv, 256 * 4);
  if (gx > 256) gx = 256;
  if (gx < 1) gx = 1;
  dim3 grid((unsigned)gx, (unsigned)(N * Tdst));
  if (scatter) hipLaunchKernelGGL(frames_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (const u32x4*)src, (u32x4*)dst, Tsrc, Tdst, fv, t0, tstep, (const u32x4*)fill, cvec);
  else hipLaunchKernelGGL(frames_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const u32x4*)src, (u32x4*)dst, Tsrc, Tdst, fv, t0, tstep, nullptr, 1);
  PB_LAUNCH_CHECK();
  return 0;
}
