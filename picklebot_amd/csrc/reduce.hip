// Fixed-order reduction of per-workgroup partial results ("slots").
//
// Nothing on the gradient / statistics path accumulates with floating-point atomics: the order in which
// workgroups retire is not reproducible, and on these BatchNorm-heavy nets a last-bit change in a sum can flip
// a hardswish / ReLU kink a few layers later.  Every reducing kernel therefore stores the partial result of
// each workgroup (or work unit) into a slot of its own with plain stores, and a consumer adds the slots up in
// an order that depends only on the shapes: either the kernel that needs the total anyway (pb_bn_finalize,
// pb_bn_bwd_finalize, pb_se_fc_fwd, pb_se_fc_bwd) or the kernel below (weight gradients).
#include "pb_common.h"

namespace {

constexpr int MAX_JOBS = 32;     // (1.4 KB of kernel arguments; a backward pass batches the jobs of several blocks into one launch)

struct ReduceJobs {
  const float* part[MAX_JOBS];
  float* out[MAX_JOBS];
  int64_t n[MAX_JOBS];
  int64_t stride[MAX_JOBS];
  int split[MAX_JOBS];
  int accumulate[MAX_JOBS];
  int blk0[MAX_JOBS + 1];       // first workgroup of each job
  int count;
};

// out[e] (+)= sum_{s < split} part[s*stride + e].  A workgroup = 32 columns x 8 slot groups; group g adds the slots
// g, g+8, g+16, ... with four independent accumulators (eight loads in flight), the eight groups are combined through LDS in index order:
// the association is a function of `split` alone.  Columns are float4 when n, stride and the pointers allow.
__global__ __launch_bounds__(256) void partials_reduce_kernel(ReduceJobs J) {
  __shared__ f32x4 red[8][32];
  int j = 0;
  for (int i = 1; i < J.count; ++i)
    if ((int)blockIdx.x >= J.blk0[i]) j = i;
  const float* P = J.part[j];
  float* out = J.out[j];
  const int64_t n = J.n[j], stride = J.stride[j];
  const int split = J.split[j];
  const bool vec = ((n | stride) & 3) == 0 && (((uintptr_t)P | (uintptr_t)out) & 15) == 0;
  const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int64_t cols = vec ? n / 4 : n;
  const int64_t e = (int64_t)(blockIdx.x - J.blk0[j]) * 32 + col;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  if (e < cols) {
    // eight slots' loads in flight per round, the tail predicated (a tail loop of single loads was one memory round trip
    // per slot); chain u & 3 takes the u-th slot of a round: the association is still a function of `split` alone
    if (vec) {
      const f32x4* p = reinterpret_cast<const f32x4*>(P) + e;
      const int64_t st4 = stride / 4;
      for (int i = grp; i < split; i += 64) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = i + 8 * u < split ? p[(int64_t)(i + 8 * u) * st4] : f32x4{0.f, 0.f, 0.f, 0.f};
        s0 += v[0]; s1 += v[1]; s2 += v[2]; s3 += v[3];
        s0 += v[4]; s1 += v[5]; s2 += v[6]; s3 += v[7];
      }
    } else {
      const float* p = P + e;
      for (int i = grp; i < split; i += 64) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = i + 8 * u < split ? p[(int64_t)(i + 8 * u) * stride] : 0.f;
        s0[0] += v[0]; s1[0] += v[1]; s2[0] += v[2]; s3[0] += v[3];
        s0[0] += v[4]; s1[0] += v[5]; s2[0] += v[6]; s3[0] += v[7];
      }
    }
  }
  red[grp][col] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (grp == 0 && e < cols) {
    f32x4 t = red[0][col];
#pragma unroll
    for (int g = 1; g < 8; ++g) t += red[g][col];
    if (vec) {
      f32x4* o = reinterpret_cast<f32x4*>(out) + e;
      *o = J.accumulate[j] ? *o + t : t;
    } else {
      out[e] = J.accumulate[j] ? out[e] + t[0] : t[0];
    }
  }
}

}  // namespace

int pb_launch_partials_reduce(const pb_reduce_job* jobs, int count, hipStream_t st) {
  for (int i = 0; i < count;) {        // the running index is carried across batches: skipped (empty) jobs must not shift them
    ReduceJobs J = {};
    int nb = 0, k = 0;
    for (; i < count && k < MAX_JOBS; ++i) {
      const pb_reduce_job& q = jobs[i];
      if (q.n <= 0 || q.split <= 0) continue;
      J.part[k] = q.part; J.out[k] = q.out; J.n[k] = q.n; J.stride[k] = q.stride; J.split[k] = q.split;
      J.accumulate[k] = q.accumulate;
      J.blk0[k] = nb;
      const bool vec = ((q.n | q.stride) & 3) == 0 && (((uintptr_t)q.part | (uintptr_t)q.out) & 15) == 0;
      nb += (int)pb_cdiv(vec ? q.n / 4 : q.n, 32);
      ++k;
    }
    if (k == 0) continue;
    J.blk0[k] = nb;
    J.count = k;
    hipLaunchKernelGGL(partials_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, st, J);
  }
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

extern "C" int pb_partials_reduce(const pb_reduce_job* jobs, int count, int device, void* stream) {
  PB_CHECK_ARG(jobs && count > 0 && count <= 64, "pb_partials_reduce: bad job list");
  for (int i = 0; i < count; ++i) {
    PB_CHECK_ARG(jobs[i].part && jobs[i].out && jobs[i].n >= 0 && jobs[i].split >= 0 && jobs[i].stride >= jobs[i].n,
                 "pb_partials_reduce: bad job %d", i);
    PB_CHECK_ARG(jobs[i].n < (1LL << 36), "pb_partials_reduce: job %d too large", i);
  }
  PB_ENTER(device);
  if (pb_launch_partials_reduce(jobs, count, (hipStream_t)stream)) {
    pb_set_error("pb_partials_reduce: launch failed");
    return 2;
  }
  return 0;
}
