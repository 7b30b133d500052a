// Internal (not exported) entry points shared between the depthwise translation units.
#pragma once
#include "pb_common.h"

// LDS-tiled kernels for small frames (dw_tile.hip).  Return 0 on launch, 1 on error, and -1 when the
// shape is not eligible (caller falls through to the strip / generic kernels).
__attribute__((visibility("hidden"))) int pb_dw_tile_wgrad(const void* dy, const void* x, float* dw,
                                                           const pb_dwconv_shape* s, pb_dtype dtype,
                                                           const float* g_scale, const float* g_shift,
                                                           hipStream_t st);
__attribute__((visibility("hidden"))) int pb_dw_tile_fwd(const void* src, const float* w, void* dst,
                                                         const pb_dwconv_shape* s, pb_dtype dtype, int flip,
                                                         const float* g_scale, const float* g_shift, float* pool,
                                                         hipStream_t st);

// would the tile kernels take this correlation (same test the launchers apply)?
__attribute__((visibility("hidden"))) bool pb_dw_tile_eligible(const pb_dwconv_shape* s, pb_dtype dtype, bool wgrad);

// whole-frame-in-registers kernels for 7x7 frames (dw7.hip); same return convention
__attribute__((visibility("hidden"))) bool pb_dw7_eligible(const pb_dwconv_shape* s);
__attribute__((visibility("hidden"))) int pb_dw7_fwd(const void* src, const float* w, void* dst, const pb_dwconv_shape* s,
                                                     pb_dtype dtype, int flip, const float* g_scale, const float* g_shift,
                                                     float* pool, hipStream_t st);
__attribute__((visibility("hidden"))) int pb_dw7_wgrad(const void* dy, const void* x, float* dw, const pb_dwconv_shape* s,
                                                       pb_dtype dtype, const float* g_scale, const float* g_shift,
                                                       hipStream_t st);
