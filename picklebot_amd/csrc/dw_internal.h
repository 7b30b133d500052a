// Internal (not exported) entry points shared between the depthwise translation units.
#pragma once
#include "pb_common.h"

// two adjacent channels of T as one register: bf16 pairs are a dword, unpacked with a shift / a mask
template <typename T> struct Pair;
template <> struct Pair<bf16_t> {
  typedef unsigned raw_t;
  static __device__ __forceinline__ f32x2 up(unsigned r) {
    return f32x2{__builtin_bit_cast(float, r << 16), __builtin_bit_cast(float, r & 0xffff0000u)};
  }
  static __device__ __forceinline__ unsigned down(f32x2 v) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 o = {(bf16_t)v[0], (bf16_t)v[1]};
    return __builtin_bit_cast(unsigned, o);
  }
};
template <> struct Pair<float> {
  typedef f32x2 raw_t;
  static __device__ __forceinline__ f32x2 up(f32x2 r) { return r; }
  static __device__ __forceinline__ f32x2 down(f32x2 v) { return v; }
};

// acc += x * w on a pair of channels, as ONE instruction whose place in the instruction stream is kept: hipcc's scheduler
// lines the FMAs of an accumulator up one behind the other, and every dependent v_pk_fma_f32 then costs an s_nop (as
// many wait states as FMAs in the depthwise kernels' ISA).  Issued round-robin over several accumulators - the order the
// kernels write them in - no two neighbours depend on each other.
__device__ __forceinline__ void pk_fma_keep(f32x2& acc, const f32x2& x, const f32x2& w) {
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(w));
}

#define PB_HIDDEN __attribute__((visibility("hidden")))

// Conventions: launchers return 0 on launch, 1 on error, and -1 when the shape is not eligible (the caller falls
// through to the next kernel family).  `pool` = squeeze-excite partial sums [slots][N][C], every slot written with
// plain stores.  Weight-gradient launchers write `*slots` partial gradients [C][taps] into `part` (a workspace of
// ..._wgrad_slots(s) of them; 0 = shape not taken) - nothing accumulates with atomics.

// LDS-tiled kernels for small frames (dw_tile.hip); the forward has no fused pooling.
PB_HIDDEN int64_t pb_dw_tile_wgrad_slots(const pb_dwconv_shape* s, pb_dtype dtype);
PB_HIDDEN int pb_dw_tile_wgrad(const void* dy, const void* x, float* part, const pb_dwconv_shape* s, pb_dtype dtype,
                               const float* g_scale, const float* g_shift, hipStream_t st, int64_t* slots);
PB_HIDDEN int pb_dw_tile_fwd(const void* src, const float* w, void* dst, const pb_dwconv_shape* s, pb_dtype dtype, int flip,
                             const float* g_scale, const float* g_shift, hipStream_t st);
// would the tile kernels take this correlation (same test the launchers apply)?
PB_HIDDEN bool pb_dw_tile_eligible(const pb_dwconv_shape* s, pb_dtype dtype, bool wgrad);

// whole-frame-in-registers kernels for 7x7 frames (dw7.hip); pool slots = s->To
PB_HIDDEN bool pb_dw7_eligible(const pb_dwconv_shape* s);
PB_HIDDEN int pb_dw7_fwd(const void* src, const float* w, void* dst, const pb_dwconv_shape* s, pb_dtype dtype, int flip,
                         const float* g_scale, const float* g_shift, float* pool, hipStream_t st);
PB_HIDDEN int64_t pb_dw7_wgrad_slots(const pb_dwconv_shape* s);
PB_HIDDEN int pb_dw7_wgrad(const void* dy, const void* x, float* part, const pb_dwconv_shape* s, pb_dtype dtype,
                           const float* g_scale, const float* g_shift, hipStream_t st, int64_t* slots);

// rolling-window kernels with lanes along channel pairs (dw_rows.hip)
PB_HIDDEN bool pb_dw_rows_eligible(const pb_dwconv_shape* s, bool wgrad);
PB_HIDDEN int pb_dw_rows_pool_slots(const pb_dwconv_shape* s, pb_dtype dtype);
PB_HIDDEN int pb_dw_rows_fwd(const void* src, const float* w, void* dst, const pb_dwconv_shape* s, pb_dtype dtype, int flip,
                             const float* g_scale, const float* g_shift, float* pool, hipStream_t st);
PB_HIDDEN int64_t pb_dw_rows_wgrad_slots(const pb_dwconv_shape* s, pb_dtype dtype);
PB_HIDDEN int pb_dw_rows_wgrad(const void* dy, const void* x, float* part, const pb_dwconv_shape* s, pb_dtype dtype,
                               const float* g_scale, const float* g_shift, hipStream_t st, int64_t* slots);

// backward-data + weight gradient in ONE walk over dy (bf16, (1,k,k), stride 1); ..._slots = 0: shape not taken
PB_HIDDEN int64_t pb_dw_rows_bwd_fused_slots(const pb_dwconv_shape* s, pb_dtype dtype);
PB_HIDDEN int pb_dw_rows_bwd_fused(const void* dy, const void* x, const float* w, void* dx, float* part, const pb_dwconv_shape* s,
                                   pb_dtype dtype, const float* g_scale, const float* g_shift, hipStream_t st, int64_t* slots);

// the same walk fed from a wave-private LDS ring filled by LDS-DMA, bf16 (dw_ring.hip); pb_dw_rows_fwd / _pool_slots try it first
PB_HIDDEN bool pb_dw_ring_eligible(const pb_dwconv_shape* s, pb_dtype dtype);
PB_HIDDEN int pb_dw_ring_pool_slots(const pb_dwconv_shape* s, pb_dtype dtype);
PB_HIDDEN int pb_dw_ring_fwd(const void* src, const float* w, void* dst, const pb_dwconv_shape* s, pb_dtype dtype, int flip,
                             const float* g_scale, const float* g_shift, float* pool, hipStream_t st);

// (kt,3,3) kernels walking along time with a ring of kt output accumulators (dw3.hip)
PB_HIDDEN bool pb_dw3_eligible(const pb_dwconv_shape* s);
PB_HIDDEN int pb_dw3_pool_slots(const pb_dwconv_shape* s);
PB_HIDDEN int pb_dw3_fwd(const void* src, const float* w, void* dst, const pb_dwconv_shape* s, pb_dtype dtype, int flip,
                         const float* g_scale, const float* g_shift, float* pool, hipStream_t st,
                         const void* hist = nullptr, int nhist = 0);
