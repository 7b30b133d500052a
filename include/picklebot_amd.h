/*
 * picklebot_amd - MI355X (gfx950) kernels for Picklebot's video-model forward/backward hot path.
 *
 * C ABI (plain pointers + sizes, no torch types).  Every entry point:
 *   - takes raw DEVICE pointers, explicit sizes, a dtype enum for activations, the device
 *     ordinal and the hipStream_t (as void*) to launch on;
 *   - returns 0 on success, non-zero on failure (see pb_last_error(), thread-local);
 *   - never allocates, never synchronises the device, never throws; re-entrant (backward is
 *     driven from PyTorch's autograd thread, SURVEY.md 8b "Threading").
 *
 * Activations are NDHWC ("channels_last_3d"): a tensor of logical shape (N,C,T,H,W) is the
 * row-major matrix [M = N*T*H*W rows][C channels].  That is the physical layout the
 * reference's training loop already produces (train.py:106).  Parameters stay fp32 in the
 * reference's own shapes (state_dict compatible); kernels convert on load.
 *
 * What each entry replaces in the reference (which has no native code: every op below is a
 * stock torch op there) is cited per function as /root/reference file:line.
 */
#ifndef PICKLEBOT_AMD_H
#define PICKLEBOT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { PB_F32 = 0, PB_BF16 = 1 } pb_dtype;

/* Activations used on the path: mobilenet.py:18,20,143,147-175 (ReLU, Hardsigmoid, Hardswish,
 * LeakyReLU), mobilevit.py:16,23,34 (SiLU). */
typedef enum {
  PB_ACT_NONE = 0, PB_ACT_RELU = 1, PB_ACT_LEAKY_RELU = 2, PB_ACT_HSWISH = 3,
  PB_ACT_HSIGMOID = 4, PB_ACT_SILU = 5
} pb_act;

#define PB_ABI_VERSION 21

/* Determinism.  Nothing in this library accumulates with floating-point atomics: two runs of the same binary on the
 * same inputs give bit-identical outputs and gradients.  Every reduction that spans workgroups stores one partial
 * result per workgroup (or work unit) in a "slot" of its own, and a consumer adds the slots in an order that depends
 * only on the shapes.
 *
 * Per-channel statistics buffers (`stats`, `sums` below) are `slots` slots of [2][C] fp32, layout [slot][2][C],
 * ZERO-FILLED by the caller; slots = PB_STAT_SLOTS unless the entry takes a slot count (the stem, whose launch wants
 * more workgroups).  Workgroup b of a producing launch adds its partial sums into slot b with plain read-modify-writes
 * (the launchers keep their grids <= slots); pb_bn_finalize / pb_bn_bwd_finalize add the slots up in a fixed order
 * (in double). */
#define PB_STAT_SLOTS 512
int pb_abi_version(void);
/* Last error message of the calling thread ("" if none). */
const char* pb_last_error(void);

/* out[e] (+)= sum_{s < split} part[s*stride + e] for e < n, slots added in a fixed order (a function of `split`
 * alone).  The weight-gradient entries below run this themselves unless the caller asks for the job back
 * (`deferred` argument) in order to batch the reductions of several gradients into ONE launch. */
typedef struct {
  const float* part;
  float* out;
  int64_t n;
  int64_t stride;
  int split;
  int accumulate;      /* 0: out = sum, != 0: out += sum */
} pb_reduce_job;
int pb_partials_reduce(const pb_reduce_job* jobs, int count, int device, void* stream);

/* ---------------------------------------------------------------- pointwise 1x1x1 convolution
 * Replaces nn.Conv3d(Cin,Cout,1) forward / backward: mobilenet.py:64,79,141(head),179,187,189;
 * movinet.py:47,64,140; mobilevit.py:14; and every nn.Linear on tokens (mobilevit.py:32-36,58-61).
 *
 * GEMM  Y[M,N] = P(X)[M,K] * Wl[N,K]^T  with fused prologue/epilogue:
 *   P(x)[m,k] = act_in(x*in_scale[k] + in_shift[k]) * row_scale[m / rows_per_group, k]
 *   y[m,n]    = act_out(acc*out_scale[n] + out_shift[n])           (bias = out_shift)
 *   stats[slot][0..N) += sum_m y, stats[slot][N..2N) += sum_m y^2  (of the stored, rounded y; slot = workgroup)
 * Wl is fp32: element (n,k) at w[n*ldw + k], or at w[k*ldw + n] when w_transposed != 0 (this is
 * how backward-data reuses the entry: dX = dY * W is the same GEMM with W read transposed).
 * Any of in_scale/in_shift/row_scale/out_scale/out_shift/stats may be NULL.
 */
int pb_pwconv_fwd(const void* x, const float* w, void* y, int64_t M, int K, int N, int ldw,
                  int w_transposed, pb_dtype dtype,
                  const float* in_scale, const float* in_shift, pb_act in_act, float in_slope,
                  const float* row_scale, int64_t rows_per_group,
                  const float* out_scale, const float* out_shift, pb_act out_act, float out_slope,
                  float* stats, int device, void* stream);

/* bf16 weight preparation for pb_pwconv_fwd_bf16w (once per optimiser step and orientation):
 *   transpose == 0: out[g][n][k] = bf16(w[n][k] * row_scale[g][k])   (row_scale NULL, G = 1: plain convert;
 *                   with row_scale: the squeeze-excite scale of clip g folded into the weights, so that
 *                   (X_g * s_g) W^T = X_g (W * s_g)^T and the activation stream is never touched)
 *   transpose != 0: out[k][n] = bf16(w[n][k])   (backward-data: dX = dY * W is the forward GEMM on W^T)
 * w is fp32 [N][K] contiguous; the inner dimension of `out` must be a multiple of 8. */
int pb_pw_weight_prep(const float* w, int N, int K, int transpose, const float* row_scale, int G, void* out_bf16,
                      int device, void* stream);

/* The same conversion for many weight matrices in ONE launch (a model's forward prepares all of its pointwise
 * weights, both orientations, at once).  desc: device array of `count` records of four int64:
 * {fp32 source pointer, bf16 destination pointer, (N << 32) | K, mode}; mode 0: [N][K], 1: transposed [K][N],
 * 2 / 3: blockdiag(B,B,B,B) of the mode-0 / mode-1 matrix B (the 4-rows-per-line form of K = 16 layers); no scaling. */
int pb_pw_weight_prep_batch(const int64_t* desc, int count, int device, void* stream);

/* Y[M,N] = X[M,K] * Wb[g(m)][N,K]^T, bf16, both operands streamed by LDS-DMA rings (pw_gemm_dma.hip); same
 * epilogue as pb_pwconv_fwd (out_scale/out_shift/act, BN statistics).  Wb from pb_pw_weight_prep:
 * [groups][N][K]; groups > 1: rows [g*rows_per_group, (g+1)*rows_per_group) use Wb[g].  K, N multiples of 8.
 * residual (optional, bf16 [M][N], not together with stats) is added in the store pass: y = epilogue(x Wb^T) + residual
 * (the `x + attn(x)` / `x + ff(x)` of mobilevit.py:102-103 and their backward `dx = dy W + dy_skip`). */
int pb_pwconv_fwd_bf16w(const void* x, const void* wb, void* y, int64_t M, int K, int N, int groups,
                        int64_t rows_per_group, const float* out_scale, const float* out_shift, pb_act out_act,
                        float out_slope, float* stats, const void* residual, int device, void* stream);
/* K = 16 layers, four rows per 128-byte line: x [M][16] is the same memory as [M/4][64] and y [M][N] as [M/4][4N] =
 * X' * blockdiag(B,B,B,B)^T - a full-width K = 64 product on a quarter of the rows (the kernel is HBM-bound; the MFMA work
 * on the zero blocks is free).  wb4: the block-diagonal bf16 matrix [4N][64] (pb_pw_weight_prep_many mode 2 / 3).
 * stats (may be NULL): [slots][2][N] as for pb_pwconv_fwd - the four column replicas of a channel are added up inside
 * the workgroup before its slot is written.  M % 4 == 0, N % 8 == 0, N <= 24. */
int pb_pwconv_fwd_fold4(const void* x, const void* wb4, void* y, int64_t M, int N, pb_act out_act, float out_slope,
                        float* stats, int device, void* stream);
/* Which kernel pb_pwconv_fwd_bf16w runs for a shape: 1 = the weight-stationary streaming kernel (pw_stream_kernel: one
 * weight matrix, K <= 160), 2 = its long-reduction form (pw_stream_wide_kernel: the output columns spread over the waves),
 * 0 = the LDS-DMA ring kernel (pw_gemm_dma_kernel).  For profilers' labels; no device work. */
int pb_pwconv_fwd_bf16w_path(int64_t M, int K, int N, int groups, int64_t rows_per_group, int has_stats);

/* Weight gradient  dW[N,K] += sum_m dY[m,n] * P(X)[m,k]   (same prologue P as above).
 * Replaces the weight half of Conv3d/Linear backward for the sites listed above.
 * The row ranges of the launch store partial dW tiles in the caller-provided workspace `ws` (size from
 * pb_pwconv_bwd_weight_ws_bytes for the same arguments; 16-byte aligned) and a fixed-order reduction adds them into
 * dw - no atomics.  deferred != NULL: the reduction is NOT launched; *deferred receives the job for
 * pb_partials_reduce (ws must stay alive until that runs). */
int64_t pb_pwconv_bwd_weight_ws_bytes(int64_t M, int K, int N, pb_dtype dtype, int has_affine, int has_row_scale,
                                      int64_t rows_per_group, int device);
int pb_pwconv_bwd_weight_ws(const void* dy, const void* x, float* dw, int64_t M, int K, int N, pb_dtype dtype,
                            const float* in_scale, const float* in_shift, pb_act in_act, float in_slope,
                            const float* row_scale, int64_t rows_per_group, void* ws, int64_t ws_bytes,
                            pb_reduce_job* deferred, int device, void* stream);

/* ---------------------------------------------------------------- dense 3-D convolution (implicit GEMM)
 * Replaces nn.Conv3d(Cin,Cout,k,stride,padding,bias=False) with groups=1: mobilevit.py:19-24 (the
 * 3x3x3 convs of MobileViTBlock: conv1 C->C, conv4 2C->C; mobilevit.py:112,119) and a dense
 * movinet.py:31 CausalConv3d.  x [N,Ti,Hi,Wi,Cin] -> y [N,To,Ho,Wo,Cout], same K loop as
 * pb_pwconv_fwd run over (tap, channel stage) with the X rows shifted per tap.
 * The kernels take the filter as fp32 [Cout][taps][Cin] (pb_conv_weight_prep mode 0); backward-data
 * (stride 1) is the same entry on dy with the mode-1 filter ([Cin][taps reversed][Cout]) and padding
 * k-1-p; the weight gradient is produced in [Cout][taps][Cin] and returned to the reference layout
 * with mode 2.  Channel counts must be multiples of the 16-byte vector (8 bf16 / 4 fp32).
 * Epilogue (out_scale/out_shift/out_act/stats) as in pb_pwconv_fwd. */
typedef struct {
  int N, Cin, Cout;
  int Ti, Hi, Wi;
  int To, Ho, Wo;
  int kt, kh, kw;
  int st, sh, sw;
  int pt, ph, pw;      /* front paddings */
} pb_conv_shape;

int pb_conv_weight_prep(const float* in, float* out, int Cout, int Cin, int taps, int mode, int device, void* stream);
int pb_conv3d_fwd(const void* x, const float* wp, void* y, const pb_conv_shape* s, pb_dtype dtype,
                  const float* out_scale, const float* out_shift, pb_act out_act, float out_slope, float* stats,
                  int device, void* stream);
/* bf16, stride 1, Cin % 8 == 0, Cout % 8 == 0: both operands by LDS-DMA (the implicit-GEMM form of pb_pwconv_fwd_bf16w:
 * the 16-byte chunks of an im2col row are fetched from the tap-shifted voxel, or from `zero16` - any 16 zero bytes in
 * device memory - for padding).  wb: bf16 [Cout][taps][Cin] (pb_conv_weight_prep mode 0 or 1, then pb_pw_weight_prep). */
int pb_conv3d_fwd_bf16w(const void* x, const void* wb, void* y, const pb_conv_shape* s, const void* zero16,
                        const float* out_scale, const float* out_shift, pb_act out_act, float out_slope, float* stats,
                        int device, void* stream);
/* Weight gradient dwp[Cout][taps][Cin] += ...: partial tiles per row range in `ws` (size from
 * pb_conv3d_bwd_weight_ws_bytes) + one fixed-order reduction into dwp; zero16 = 16 zero bytes of device memory for
 * the padding voxels.  bf16 stride-1 shapes run on the LDS-DMA weight-gradient kernel, the rest on the
 * register-staged one. */
int64_t pb_conv3d_bwd_weight_ws_bytes(const pb_conv_shape* s, pb_dtype dtype, int device);
int pb_conv3d_bwd_weight_ws(const void* dy, const void* x, float* dwp, const pb_conv_shape* s, pb_dtype dtype, void* ws,
                            int64_t ws_bytes, const void* zero16, int device, void* stream);

/* ---------------------------------------------------------------- depthwise 3-D convolution
 * Replaces nn.Conv3d(C,C,(kt,kh,kw),groups=C,bias=False) forward/backward:
 * mobilenet.py:67-75 (kernel (1,k,k), scalar stride AND padding on T,H,W: T_out grows),
 * movinet.py:52-61 ((kt,k,k), stride (1,s,s), symmetric padding), movinet.py:31 (CausalConv3d:
 * pad_t_front = kt-1, pad_t_back = 0).
 * x [N,Ti,Hi,Wi,C] -> y [N,To,Ho,Wo,C]; w fp32 in the reference's layout [C][kt*kh*kw].
 * Temporal padding is asymmetric-capable: frame index t_in = t_out*st - pt_front + a.
 * pool_part (optional): squeeze for SEBlock3D (mobilenet.py:16) as per-work-unit partial sums over (To,Ho,Wo) of y,
 * fp32 [pool_slots][N][C] with pool_slots = pb_dwconv3d_pool_slots(s, dtype); every slot is written (plain stores,
 * no zero-fill needed); pb_se_fc_fwd (or pb_partials_reduce) adds the slots up.
 */
typedef struct {
  int N, C;
  int Ti, Hi, Wi;
  int To, Ho, Wo;
  int kt, kh, kw;
  int st, sh, sw;
  int pt, ph, pw;      /* front paddings (symmetric convs: also the back padding) */
} pb_dwconv_shape;

int pb_dwconv3d_pool_slots(const pb_dwconv_shape* s, pb_dtype dtype);
int pb_dwconv3d_fwd(const void* x, const float* w, void* y, const pb_dwconv_shape* s, pb_dtype dtype,
                    float* pool_part, int pool_slots, int device, void* stream);

/* Streaming causal depthwise convolution (CausalConv3d with a stream buffer, movinet.py:7-39): the nhist frames that
 * precede x in time are read from the caller-owned history `hist` [N][nhist][Hi][Wi][C] instead of being zero padding;
 * s->pt = total front padding of the equivalent unchunked call (frames before the history are zeros).  The caller
 * advances the history (last nhist frames of hist ++ x) between chunks.  Returns 3 when the shape is not one the
 * (kt,3,3) kernels take (the caller then concatenates history and chunk and calls pb_dwconv3d_fwd). */
int pb_dwconv3d_fwd_hist(const void* x, const void* hist, int nhist, const float* w, void* y, const pb_dwconv_shape* s,
                         pb_dtype dtype, int device, void* stream);

/* Backward.  The incoming gradient may carry the fused squeeze-excite backward:
 *   g[n,t,h,w,c] = dy[n,t,h,w,c] * g_scale[n,c] + g_shift[n,c]   (either may be NULL)
 * dx = conv_transpose(g, w);  dw[C][kt*kh*kw] += sum g * x: per-workgroup partial gradients go to `ws`
 * (pb_dwconv3d_bwd_weight_ws_bytes, 16-byte aligned) and are added in a fixed order; `deferred` as in
 * pb_pwconv_bwd_weight_ws. */
int pb_dwconv3d_bwd_data(const void* dy, const float* w, void* dx, const pb_dwconv_shape* s, pb_dtype dtype,
                         const float* g_scale, const float* g_shift, int device, void* stream);
int64_t pb_dwconv3d_bwd_weight_ws_bytes(const pb_dwconv_shape* s, pb_dtype dtype);
int pb_dwconv3d_bwd_weight(const void* dy, const void* x, float* dw, const pb_dwconv_shape* s, pb_dtype dtype,
                           const float* g_scale, const float* g_shift, void* ws, int64_t ws_bytes,
                           pb_reduce_job* deferred, int device, void* stream);
/* Both gradients of Bottleneck3D's depthwise conv (mobilenet.py:67-75) in ONE launch: dx as pb_dwconv3d_bwd_data, dw as
 * pb_dwconv3d_bwd_weight; dy is read once.  bf16, (1,k,k) with k in {3,5}, stride 1, "same" spatial padding.
 * pb_dwconv3d_bwd_fused_ws_bytes returns 0 when the shape / dtype is not taken (then call the two entry points above). */
int64_t pb_dwconv3d_bwd_fused_ws_bytes(const pb_dwconv_shape* s, pb_dtype dtype);
int pb_dwconv3d_bwd_fused(const void* dy, const void* x, const float* w, void* dx, float* dw, const pb_dwconv_shape* s,
                          pb_dtype dtype, const float* g_scale, const float* g_shift, void* ws, int64_t ws_bytes,
                          pb_reduce_job* deferred, int device, void* stream);

/* ---------------------------------------------------------------- frame gather / scatter (temporally strided bottlenecks)
 * Bottleneck3D's depthwise conv has a (1,k,k) kernel and strides T as well (mobilenet.py:67-75): with stride 2 only every
 * other input frame is ever read.  gather: dst[N][Tdst][frame] = src[N][Tsrc][frame] frames t0, t0+tstep, ...; scatter
 * (scatter != 0): dst[n][t] = src[n][(t - t0) / tstep] where that is a frame of src, exact zeros elsewhere (the input
 * gradient of the unread frames) - or, with `fill` (one pixel row of fill_bytes, a multiple of 16 that divides the frame), that
 * row repeated: the frames a (1,k,k) depthwise conv pads in time are exact zeros before BatchNorm and the per-channel
 * constant act(shift) after a folded inference BatchNorm.  frame_bytes % 16 == 0. */
int pb_frames_gather(const void* src, void* dst, int N, int Tsrc, int Tdst, int64_t frame_bytes, int t0, int tstep,
                     int scatter, const void* fill, int fill_bytes, int device, void* stream);

/* Name of the HIP kernel the dispatcher picks for this shape (profiling labels: rocprofv3 kernel names
 * <-> bench.py families).  which: 0 = forward, 1 = backward-data, 2 = backward-weight.  Static string. */
const char* pb_dwconv3d_kernel_name(const pb_dwconv_shape* s, pb_dtype dtype, int which);

/* ---------------------------------------------------------------- dense conv with tiny Cin (stems)
 * Replaces nn.Conv3d(3,Cout,k,stride,pad) on the raw clip: mobilenet.py:141,221 (3x3x3 s2 p1 +bias),
 * movinet.py:92 ((1,3,3) s(1,2,2) p(0,1,1)), mobilevit.py:165 (3x3x3 s2 p1, no bias).
 * The input is addressed through element strides, so both the NDHWC view train.py:106 produces and
 * a plain contiguous NCDHW tensor are read in place.  x_dtype: PB_F32 / PB_BF16 or 2 = uint8 (the
 * raw dataloader clip, scaled by x_scale = 1/255 on load: SURVEY.md 8f rank 2).
 * y [N,To,Ho,Wo,Cout] in `dtype`; Cout <= 32; stats [stat_slots][2][Cout] as in pb_pwconv_fwd (stat_slots >= 1). */
typedef struct {
  int N, Cin, Cout;
  int Ti, Hi, Wi;
  int To, Ho, Wo;
  int kt, kh, kw;
  int st, sh, sw;
  int pt, ph, pw;
  int64_t xs_n, xs_c, xs_t, xs_h, xs_w;   /* input element strides */
} pb_stem_shape;

int pb_stem_conv3d_fwd(const void* x, int x_dtype, float x_scale, const float* w, const float* bias, void* y,
                       const pb_stem_shape* s, pb_dtype dtype, float* stats, int stat_slots, int device, void* stream);
/* dw[Cout][Cin][kt][kh][kw] += ..., dbias[Cout] += sum dy (dbias may be NULL): per-workgroup partials in `ws`
 * (pb_stem_conv3d_bwd_weight_ws_bytes) added in a fixed order. */
int64_t pb_stem_conv3d_bwd_weight_ws_bytes(const pb_stem_shape* s, int x_dtype, float x_scale, pb_dtype dtype);
int pb_stem_conv3d_bwd_weight(const void* dy, const void* x, int x_dtype, float x_scale, float* dw, float* dbias,
                              const pb_stem_shape* s, pb_dtype dtype, void* ws, int64_t ws_bytes, int device,
                              void* stream);

/* ---------------------------------------------------------------- per-channel statistics / norm / activation
 * Replace nn.BatchNorm3d (+ the activation and Dropout3d after it): mobilenet.py:80-82,90-92,142-143;
 * movinet.py:65-66,75-76,93-94,141-143; mobilevit.py:15-16,22-23.
 */
/* stats[0..C) += sum_m x, stats[C..2C) += sum_m x^2 */
int pb_chan_stats(const void* x, int64_t M, int C, pb_dtype dtype, float* stats, int device, void* stream);

/* Training: mean/var from stats (count = M), running stats updated in place (unbiased var,
 * `momentum`), or evaluation (use_running != 0): from running_mean / running_var.
 * Outputs scale = gamma*rstd, shift = beta - mean*scale, plus mean and rstd for backward.
 * gamma/beta NULL = affine-free BatchNorm (mobilevit.py:15).  num_batches_tracked (optional, device int64) is
 * incremented in training, like nn.BatchNorm does. */
int pb_bn_finalize(const float* stats, int slots, double count, const float* gamma, const float* beta, float eps,
                   float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                   int use_running, float* scale, float* shift, float* mean, float* rstd, int C, int device,
                   void* stream);

/* Inference fold of `count` BatchNorm modules in ONE launch (one per top-level eval forward; nothing is cached across
 * forwards, so statistics / affine parameters updated by a replayed HIP graph are always seen).  desc: device array of
 * count x 8 int64 = {gamma*, beta* (NULL = affine-free), running_mean*, running_var*, out* (4*C floats: scale, shift,
 * mean, rstd), C, eps as the bit pattern of a double, 0}.  Replaces nn.BatchNorm3d in eval mode (mobilenet.py:80,
 * movinet.py:65, mobilevit.py:15). */
int pb_bn_fold_batch(const int64_t* desc, int count, int device, void* stream);

/* y = act((x - mean[c])*scale[c] + shift[c]) * row_scale[g, c] + row_shift[g, c], g = m / rows_per_group;
 * optional stats of y.  For BatchNorm pass scale = gamma*rstd, shift = beta, mean = the batch mean: the
 * mean is subtracted first (a folded shift = beta - mean*scale cancels badly when |mean| >> std). */
int pb_affine_act_fwd(const void* x, void* y, int64_t M, int C, pb_dtype dtype, const float* scale,
                      const float* shift, const float* mean, pb_act act, float slope, const float* row_scale,
                      const float* row_shift, int64_t rows_per_group, float* stats, int64_t fm_full, int64_t fm_data,
                      int64_t fm_lead, int device, void* stream);
/* Frame map (fm_full > 0; pb_affine_act_fwd and pb_act_bwd_reduce): M counts the rows of the FULL tensor [clips][fm_full];
 * the pre-BatchNorm tensor (x resp. z) is COMPACT - it holds only the fm_data data rows of every clip, full rows
 * [fm_lead, fm_lead + fm_data) - and reads as exact zeros on the other rows (the frames a (1,k,k) depthwise conv pads in
 * time, mobilenet.py:67-75); pb_affine_act_fwd writes the full y, pb_act_bwd_reduce reads the full dy, sums over all M rows
 * and writes g compact. */

/* g = dy * row_scale * act'((z-mean)*scale+shift); sums[0..C) += sum g, sums[C..2C) += sum g*(z-mean)*rstd.
 * dy_rows_per_group > 0: dy is [M/dy_rows_per_group][C] broadcast over each row group (the
 * backward of a global average pool, with dy pre-divided by the caller or via dy_mul).
 * g_out may be NULL (reduction only). */
int pb_act_bwd_reduce(const void* dy, const void* z, void* g_out, int64_t M, int C, pb_dtype dtype,
                      const float* scale, const float* shift, pb_act act, float slope,
                      const float* row_scale, int64_t rows_per_group, int64_t dy_rows_per_group, float dy_mul,
                      const float* mean, const float* rstd, float* sums, int64_t fm_full, int64_t fm_data, int64_t fm_lead, int device, void* stream);

/* BatchNorm backward coefficients: dz = A*g + B*(z - mean) + Cc;  dgamma = sum g*xhat, dbeta = sum g. */
int pb_bn_bwd_finalize(const float* sums, int slots, double count, const float* gamma, const float* mean, const float* rstd,
                       int training, float* coefA, float* coefB, float* coefC, float* dgamma, float* dbeta,
                       int C, int device, void* stream);

/* out = A[c]*geff + B[c]*(z - mean[c]) + Cc[c],
 *   geff = g * row_scale[m/rows_per_group, c] * act'((z - mean[c])*act_scale[c] + act_shift[c])
 * (B, Cc, mean, row_scale, act_scale, act_shift may be NULL; act = PB_ACT_NONE: geff = g*row_scale).
 * With the activation passed here, BatchNorm+activation backward reads the incoming gradient directly: the
 * masked, activation-differentiated gradient is never materialised (pb_act_bwd_reduce with g_out = NULL
 * supplies the two sums). */
int pb_affine2(const void* g, const void* z, void* out, int64_t M, int C, pb_dtype dtype, const float* A,
               const float* B, const float* Cc, const float* mean, const float* row_scale, int64_t rows_per_group,
               const float* act_scale, const float* act_shift, pb_act act, float slope, int device, void* stream);

/* ---------------------------------------------------------------- squeeze-excite / pooling
 * SEBlock3D (mobilenet.py:11-26) and the classifier's AdaptiveAvgPool3d (mobilenet.py:186,252;
 * movinet.py:147; mobilevit.py:191). */
/* out[g][c] = sum over the group's rows of a (b == NULL) or of a*b.  ws: pb_rowgroup_reduce_ws_bytes bytes of
 * scratch for the per-workgroup partial rows (may be 0 for small inputs), added up in a fixed order. */
int64_t pb_rowgroup_reduce_ws_bytes(int64_t G, int64_t rows_per_group, int C, pb_dtype dtype);
int pb_rowgroup_reduce(const void* a, const void* b, float* out, int64_t G, int64_t rows_per_group, int C,
                       pb_dtype dtype, void* ws, int64_t ws_bytes, int device, void* stream);

/* scale[g][c] = hardsigmoid(W2 relu(W1 (pool[g]/count) + b1) + b2); saves hidden (post-relu) [G][Cm].
 * W1 [Cm][C], W2 [C][Cm] fp32 (the reference's Conv3d 1x1x1 weights flattened). */
int pb_se_fc_fwd(const float* pool, float inv_count, const float* w1, const float* b1, const float* w2,
                 const float* b2, float* scale, float* hidden, int G, int C, int Cm, int device, void* stream);

/* Backward of the above.  dscale[g][c] = sum_rows dOut*x.  Produces g_shift[g][c] = dmean/count (the
 * term the depthwise backward adds to every element) and accumulates dW1,db1,dW2,db2.
 * work: fp32 scratch of G*(C+Cm) floats. */
int pb_se_fc_bwd(const float* dscale, const float* pool, float inv_count, const float* scale, const float* hidden,
                 const float* w1, const float* w2, float* g_shift, float* dw1, float* db1, float* dw2, float* db2,
                 float* work, int G, int C, int Cm, int device, void* stream);

/* The same two steps fused, reading the squeeze partial sums directly (one launch forward, two backward instead of
 * four + six): pool_part / dscale_part are fp32 [slots][G][C] (pb_dwconv3d_fwd's pool_part; pb_rowgroup_partials);
 * pool [G][C] is written forward (the slot sum) and read backward.  Optional wproj (fp32 [N][C]) + wbs (bf16 [G][N][C]):
 * the projection conv's weights scaled per clip, wbs[g][n][c] = wproj[n][c] * scale[g][c], which pb_pwconv_fwd_bf16w
 * takes as per-group weights (C % 8 == 0). */
int pb_se_fused_fwd(const float* pool_part, int slots, float inv_count, const float* w1, const float* b1, const float* w2,
                    const float* b2, float* pool, float* scale, float* hidden, const float* wproj, void* wbs_bf16, int N,
                    int G, int C, int Cm, int device, void* stream);
int pb_se_fused_bwd(const float* dscale_part, int slots, const float* pool, float inv_count, const float* scale,
                    const float* hidden, const float* w1, const float* w2, float* g_shift, float* dw1, float* db1,
                    float* dw2, float* db2, float* work, int G, int C, int Cm, int device, void* stream);
/* dw1 == dw2 == NULL in pb_se_fused_bwd: only the per-clip vectors (g_shift, and `work` = dz2 | dh) are produced; the FC weight /
 * bias gradients follow from `work` with pb_se_fc_bwd_w - later, or on another stream (nothing on the data-gradient chain reads them). */
int pb_se_fc_bwd_w(const float* work, const float* pool, float inv_count, const float* hidden, float* dw1, float* db1, float* dw2,
                   float* db2, int G, int C, int Cm, int device, void* stream);

/* The pooled classifier head - AdaptiveAvgPool3d(1) -> Conv3d(C, Cm, 1) -> Hardswish (or ReLU) -> Conv3d(Cm, Co, 1)
 * (mobilenet.py:143-146 / 227-230 `classifier`) - on the squeeze-excite gate's kernels (fp32 weights, fixed-order sums): three
 * launches forward (pb_rowgroup_partials + the two here), two on the data-gradient chain backward, the weight gradients apart.
 *   forward:  pool_part fp32 [slots][G][C] -> pool [G][C] (sums), pre1 [G][Cm] (pre-activation of the hidden layer),
 *             logits [G][Co] fp32 and, optionally, a bf16 copy (what the reference's autocast returns).
 *   backward: dlogits fp32 [G][Co] -> dpool_row [G][C] (optionally also as bf16): the gradient of EVERY row of clip g that
 *             was pooled (1/count included); work [G][Co + Cm] = dlogits | dpre1 for pb_pool_head_bwd_w.
 *   weights:  dw1 [Cm][C], db1 [Cm], dw2 [Co][Cm], db2 [Co] += (fixed order over the clips). */
int pb_pool_head_fwd(const float* pool_part, int slots, float inv_count, const float* w1, const float* b1, const float* w2, const float* b2,
                     pb_act act1, float* pool, float* logits, void* logits_bf16, float* pre1, int G, int C, int Cm, int Co, int device,
                     void* stream);
int pb_pool_head_bwd(const float* dlogits, float inv_count, const float* pre1, const float* w1, const float* w2, pb_act act1,
                     float* dpool_row, void* dpool_row_bf16, float* work, int G, int C, int Cm, int Co, int device, void* stream);
int pb_pool_head_bwd_w(const float* work, const float* pool, float inv_count, const float* pre1, pb_act act1, float* dw1, float* db1,
                       float* dw2, float* db2, int G, int C, int Cm, int Co, int device, void* stream);
/* Row-group partial sums only (the first half of pb_rowgroup_reduce): part fp32 [slots][G][C] with
 * slots = pb_rowgroup_slots(...) for a consumer that adds the slots itself (pb_se_fused_bwd / pb_se_fused_fwd). */
int pb_rowgroup_slots(int64_t G, int64_t rows_per_group, int C, pb_dtype dtype);
int pb_rowgroup_partials(const void* a, const void* b, float* part, int64_t G, int64_t rows_per_group, int C,
                         pb_dtype dtype, int device, void* stream);

/* ---------------------------------------------------------------- MobileViT token-side helpers
 * pb_unfold_fold: einops 'b d (t pt)(h ph)(w pw) -> b (pt ph pw)(t h w) d' and its inverse incl. the
 *   zero padding to patch multiples / the crop (mobilevit.py:129-137).  Voxels [B,T,H,W,D] (unpadded),
 *   tokens [B, pt*ph*pw, t*h*w, D] with t = ceil(T/pt)...  to_tokens != 0: unfold; == 0: fold + crop.
 *   (Each is the other's backward.)
 * pb_layernorm: nn.LayerNorm(D, elementwise_affine=False) (mobilevit.py:53,69); stat [rows][2] fp32
 *   = (mean, rstd) written by forward, read by backward (backward != 0: out = dx from x, dy, stat).
 * pb_copy_cols: dst[m*dst_ld + dst_off + c] (+)= src[m*src_ld + src_off + c]: torch.cat on channels
 *   (mobilevit.py:141) and its backward split, head slices. */
int pb_unfold_fold(const void* src, void* dst, int B, int D, int T, int H, int W, int pt, int ph, int pw, int to_tokens,
                   pb_dtype dtype, int device, void* stream);
int pb_layernorm(const void* x, const void* dy, void* out, float* stat, int64_t rows, int D, float eps, int backward,
                 pb_dtype dtype, int device, void* stream);
int pb_copy_cols(const void* src, int64_t src_ld, int src_off, void* dst, int64_t dst_ld, int dst_off, int64_t M, int C,
                 int accumulate, pb_dtype dtype, int device, void* stream);
/* The WIDE token stream (round 4): Transformer.forward's running sum `x = attn(x) + x; x = ff(x) + x`
 * (mobilevit.py:100-104) kept in fp32 between blocks while every GEMM operand stays bf16.
 * pb_layernorm_wide  forward : out_bf16 = bf16(LayerNorm(x fp32)), stat [rows][2] = mean, rstd
 *                    backward: out32 = res32 + dLayerNorm(x, dy_bf16, stat); out_bf16 (optional) = bf16(out32)
 * pb_stream_add      out32 (optional) = a + b, out_bf16 (optional) = bf16(a + b); a fp32 or bf16, b bf16; n % 8 == 0 */
int pb_layernorm_wide(const float* x, const void* dy_bf16, const float* res32, void* out_bf16, float* out32, float* stat,
                      int64_t rows, int D, float eps, int backward, int device, void* stream);
int pb_stream_add(const void* a, pb_dtype a_dtype, const void* b_bf16, float* out32, void* out_bf16, int64_t n, int device,
                  void* stream);

/* ---------------------------------------------------------------- MobileViT patch attention
 * Replaces F.scaled_dot_product_attention(q,k,v) on (b, p, h=8, n, d=16) + the rearranges around it
 * (mobilevit.py:71-74 with the head merge of :82 restored): qkv [R = G*n][384] is the to_qkv output
 * (q | k | v, each [h][d]); G = b*p groups of n tokens; out [R][128] heads merged; lse [R][8] fp32
 * (log-sum-exp of the scaled scores, saved for backward).  Backward recomputes the probabilities:
 * delta [R][8] fp32 is scratch (rowsum(dO*O)), dqkv [R][384] is fully written. */
int pb_attention_fwd(const void* qkv, void* out, float* lse, int64_t G, int n, float scale, pb_dtype dtype, int device,
                     void* stream);
/* The same forward with fp8 (OCP e4m3) MFMA operands - K and V quantised per (group, head) with one scale each, Q per
 * 16-query tile, the probabilities as p*256 - and fp32 softmax statistics; qkv / out bf16.  BASELINE config 5; the
 * reference has no fp8 path, so beyond the bf16 goldens its parity is unpinned (tests state the measured tolerance).
 * n <= 1024.  Inference only. */
int pb_attention_fwd_fp8(const void* qkv, void* out, float* lse, int64_t G, int n, float scale, int device, void* stream);
int pb_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv,
                     int64_t G, int n, float scale, pb_dtype dtype, int device, void* stream);

/* ---------------------------------------------------------------- MobileViT-V2 separable attention on 3-D patch tokens
 * The part of LinearSelfAttention.forward between qkv_proj and out_proj (mobilevitv2.py:37-52), on the token rows of the
 * 3-D unfold: G = batch * pixels-in-patch groups of n patches.  qkv [G*n][1+2D] keeps qkv_proj's column order
 * (q | k | v, :37); scores fp32 [G*n] = softmax over the n patches (:41); ctx fp32 [G][D] = sum_i scores*k (:46-48);
 * out [G*n][D] = relu(v)*ctx (:52).  Backward: dctx fp32 [G][D] and dscore fp32 [G*n] are scratch, dqkv is fully written.
 * D <= 256.  The reference module is 2-D and unregistered: same arithmetic, 3-D patches. */
int pb_linattn_fwd(const void* qkv, float* scores, float* ctx, void* out, int G, int n, int D, pb_dtype dtype, int device,
                   void* stream);
int pb_linattn_bwd(const void* qkv, const float* scores, const float* ctx, const void* dout, float* dctx, float* dscore,
                   void* dqkv, int G, int n, int D, pb_dtype dtype, int device, void* stream);

/* ---------------------------------------------------------------- misc
 * fp32 [R][C] -> [C][R] (depthwise weight [C][taps] -> [taps][C] staging). */
int pb_transpose_f32(const float* in, float* out, int R, int C, int device, void* stream);

/* ---------------------------------------------------------------- optimizer step
 * Fused AdamW (torch.optim.AdamW's update rule; the reference's bitsandbytes AdamW8bit, train.py:208-212, with fp32
 * state) over one flat fp32 bucket: p, g, m, v [n], 16-byte aligned.  `step`: device float holding the step number of
 * THIS update (the caller increments it first; device memory so that a captured HIP graph advances).  inv_scale /
 * found_inf (both optional device floats): gradients are multiplied by *inv_scale, and the whole update is skipped
 * when *found_inf != 0 - torch.amp.GradScaler's unscale + skip (train.py:283-289).  pb_grad_check sets *found_inf = 1
 * when any gradient is inf/nan (the caller zeroes it).  lr_wd (optional device float[2] = {lr, weight_decay}) overrides
 * the by-value pair: a step replayed from a HIP graph then follows the scheduler (train.py:215 CosineAnnealingLR). */
int pb_adamw_flat(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, const float* step, const float* inv_scale, const float* found_inf,
                  const float* lr_wd, int device, void* stream);
int pb_grad_check(const float* g, int64_t n, float* found_inf, int device, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PICKLEBOT_AMD_H */
