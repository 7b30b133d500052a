// Shared device/host helpers for the gfx950 kernels.  Written for CDNA4 only: 64-wide
// wavefronts, MFMA, 160 KB LDS; there is no other backend.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/picklebot_amd.h"

// ----------------------------------------------------------------------------- host side
void pb_set_error(const char* fmt, ...);

#define PB_CHECK_ARG(cond, ...)                    \
  do {                                             \
    if (!(cond)) {                                 \
      pb_set_error(__VA_ARGS__);                   \
      return 1;                                    \
    }                                              \
  } while (0)

#define PB_HIP(call)                                                                   \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess) {                                                            \
      pb_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return 2;                                                                        \
    }                                                                                  \
  } while (0)

// Every entry point selects the device explicitly: backward runs on autograd worker threads
// whose "current device" is not ours to assume.
#define PB_ENTER(device) PB_HIP(hipSetDevice(device))
#define PB_LAUNCH_CHECK() PB_HIP(hipGetLastError())

static inline int64_t pb_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// fixed-order reduction of partial results (reduce.hip); 0 on success
__attribute__((visibility("hidden"))) int pb_launch_partials_reduce(const pb_reduce_job* jobs, int count, hipStream_t st);

// ----------------------------------------------------------------------------- device side
typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define PB_WAVE 64

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
  static constexpr int VEC = 4;          // elements per 16-byte access
  static constexpr pb_dtype DT = PB_F32;
};
template <> struct ElemTraits<bf16_t> {
  static constexpr int VEC = 8;
  static constexpr pb_dtype DT = PB_BF16;
};

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
// plain cast -> v_cvt_pk_bf16_f32 (RNE, NaN stays NaN on gfx950)
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// A 16-byte register block of VEC elements of T, with fp32 unpack/pack.
template <typename T> struct Vec16;
template <> struct Vec16<float> {
  f32x4 v;
  __device__ __forceinline__ float get(int i) const { return v[i]; }
  __device__ __forceinline__ void set(int i, float x) { v[i] = x; }
};
template <> struct Vec16<bf16_t> {
  bf16x8 v;
  __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
  __device__ __forceinline__ void set(int i, float x) { v[i] = (bf16_t)x; }
};

template <typename T> __device__ __forceinline__ Vec16<T> vec_zero() {
  Vec16<T> r;
#pragma unroll
  for (int i = 0; i < ElemTraits<T>::VEC; ++i) r.set(i, 0.f);
  return r;
}
template <typename T> __device__ __forceinline__ Vec16<T> vec_load(const T* p) {
  Vec16<T> r;
  r.v = *reinterpret_cast<const decltype(r.v)*>(p);
  return r;
}
template <typename T> __device__ __forceinline__ void vec_store(T* p, const Vec16<T>& r) {
  *reinterpret_cast<decltype(r.v)*>(p) = r.v;
}

__device__ __forceinline__ float act_fwd(float x, int act, float slope) {
  switch (act) {
    case PB_ACT_RELU: return x > 0.f ? x : 0.f;
    case PB_ACT_LEAKY_RELU: return x > 0.f ? x : x * slope;
    case PB_ACT_HSWISH: return x * fminf(fmaxf(x + 3.f, 0.f), 6.f) * (1.f / 6.f);
    case PB_ACT_HSIGMOID: return fminf(fmaxf(x * (1.f / 6.f) + 0.5f, 0.f), 1.f);
    case PB_ACT_SILU: return x / (1.f + __expf(-x));
    default: return x;
  }
}
// d act(x) / dx, with torch's conventions at the kinks (hardswish_backward: x<-3 -> 0,
// x<=3 -> x/3+.5, else 1; hardsigmoid_backward: -3<x<3 -> 1/6; relu/leaky: x>0).
__device__ __forceinline__ float act_bwd(float x, int act, float slope) {
  switch (act) {
    case PB_ACT_RELU: return x > 0.f ? 1.f : 0.f;
    case PB_ACT_LEAKY_RELU: return x > 0.f ? 1.f : slope;
    case PB_ACT_HSWISH: return x < -3.f ? 0.f : (x <= 3.f ? x * (1.f / 3.f) + 0.5f : 1.f);
    case PB_ACT_HSIGMOID: return (x > -3.f && x < 3.f) ? (1.f / 6.f) : 0.f;
    case PB_ACT_SILU: {
      float s = 1.f / (1.f + __expf(-x));
      return s * (1.f + x * (1.f - s));
    }
    default: return 1.f;
  }
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt(0): it would
// wait for every outstanding global store (and prefetched load) at each barrier - in the GEMM that
// serialises the epilogue stores and kills the register prefetch.  Here only the wave's LDS
// operations are waited for before the barrier; the "memory" clobber keeps the compiler from moving
// LDS accesses across it.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
