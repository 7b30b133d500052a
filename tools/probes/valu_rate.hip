// Issue cost of the VALU instructions the attention loop is made of, on gfx950: cycles per wave-instruction at 1, 2, 4
// and 8 resident waves per SIMD (is v_exp_f32 8 or 16 cycles?).  Each wave runs UNROLL independent
// instances of the instruction per loop iteration (no dependency between them), ITERS iterations, and stamps
// s_memtime (shader clock) around the loop; the host prints (max end - min start over the waves of one SIMD's share)
// / (instructions issued per SIMD) = cycles per instruction per SIMD, and the same per wave.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/valu_rate.hip -o tools/probes/valu_rate && tools/probes/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>

constexpr int UNROLL = 16;
constexpr int ITERS = 4096;

#define BODY(ASM)                                                                       \
  float r[UNROLL];                                                                      \
  for (int i = 0; i < UNROLL; ++i) r[i] = seed + i * 0.001f;                            \
  uint64_t t0 = __builtin_amdgcn_s_memtime();                                           \
  for (int it = 0; it < ITERS; ++it) {                                                  \
    _Pragma("unroll") for (int i = 0; i < UNROLL; ++i) asm volatile(ASM : "+v"(r[i]));  \
  }                                                                                     \
  uint64_t t1 = __builtin_amdgcn_s_memtime();                                           \
  float s = 0.f;                                                                        \
  for (int i = 0; i < UNROLL; ++i) s += r[i];                                           \
  if (threadIdx.x % 64 == 0) { out[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2] = t0;              \
                               out[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2 + 1] = t1; }       \
  if (s == 12345.678f) sink[0] = s;

__global__ void k_exp(uint64_t* out, float* sink, float seed) { BODY("v_exp_f32 %0, %0") }
__global__ void k_fma(uint64_t* out, float* sink, float seed) { BODY("v_fma_f32 %0, %0, %0, %0") }
__global__ void k_max3(uint64_t* out, float* sink, float seed) { BODY("v_max3_f32 %0, %0, %0, %0") }
__global__ void k_add(uint64_t* out, float* sink, float seed) { BODY("v_add_f32 %0, %0, %0") }
__global__ void k_rcp(uint64_t* out, float* sink, float seed) { BODY("v_rcp_f32 %0, %0") }
__global__ void k_cvt(uint64_t* out, float* sink, float seed) { BODY("v_cvt_pk_bf16_f32 %0, %0, %0") }
__global__ void k_ldexp(uint64_t* out, float* sink, float seed) { BODY("v_ldexp_f32 %0, %0, 1") }
__global__ void k_pkfma(uint64_t* out, float* sink, float seed) {
  typedef __attribute__((ext_vector_type(2))) float f2;
  f2 r[UNROLL];
  for (int i = 0; i < UNROLL; ++i) r[i] = f2{seed + i * 0.001f, seed};
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(r[i]));
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < UNROLL; ++i) s += r[i][0] + r[i][1];
  if (threadIdx.x % 64 == 0) { out[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2] = t0;
                               out[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 2 + 1] = t1; }
  if (s == 12345.678f) sink[0] = s;
}
// half the exps of a lane replaced by the polynomial form: magic-number split, degree-3 polynomial on packed FMAs, exponent insert
__global__ void k_mix(uint64_t* out, float* sink, float seed) { BODY("v_exp_f32 %0, %0\n\tv_add_f32 %0, %0, %0") }

typedef void (*kern_t)(uint64_t*, float*, float);

int main() {
  uint64_t* out; float* sink;
  hipMalloc(&out, 1 << 20); hipMalloc(&sink, 64);
  struct { const char* name; kern_t k; int per; } ks[] = {
      {"v_exp_f32", k_exp, 1}, {"v_rcp_f32", k_rcp, 1}, {"v_fma_f32", k_fma, 1}, {"v_add_f32", k_add, 1}, {"v_max3_f32", k_max3, 1},
      {"v_cvt_pk_bf16_f32", k_cvt, 1}, {"v_ldexp_f32", k_ldexp, 1}, {"v_pk_fma_f32", k_pkfma, 1}, {"v_exp_f32 + v_add_f32 (pair)", k_mix, 2}};
  printf("gfx950 VALU issue cost: cycles per wave-instruction, one workgroup (= one CU), W waves per SIMD, %d independent chains per wave\n", UNROLL);
  printf("%-30s %10s %10s %10s %10s   (per SIMD: elapsed shader cycles / instructions issued on that SIMD)\n", "instruction", "1 wave", "2 waves", "4 waves", "8 waves");
  for (auto& e : ks) {
    printf("%-30s", e.name);
    for (int wps : {1, 2, 4, 8}) {
      const int waves = 4 * wps;                       // one workgroup on one CU: waves spread over the 4 SIMDs
      if (waves * 64 > 1024) {                         // 8 per SIMD: two workgroups of 16 waves on (hopefully) one CU - use 2 blocks, report per block
      }
      const int blocks = wps == 8 ? 2 : 1, thr = wps == 8 ? 1024 : waves * 64;
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(thr), 0, 0, out, sink, 1.0f);       // warm-up
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(thr), 0, 0, out, sink, 1.0f);
      hipDeviceSynchronize();
      std::vector<uint64_t> h(blocks * (thr / 64) * 2);
      hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
      // per block: max end - min start; instructions per SIMD = (waves per block / 4) * UNROLL * ITERS * per
      double worst = 0;
      for (int b = 0; b < blocks; ++b) {
        uint64_t lo = ~0ull, hi = 0;
        for (int w = 0; w < thr / 64; ++w) { lo = std::min(lo, h[(b * (thr / 64) + w) * 2]); hi = std::max(hi, h[(b * (thr / 64) + w) * 2 + 1]); }
        const double inst = (double)(thr / 64 / 4) * UNROLL * ITERS * e.per;
        worst = std::max(worst, (double)(hi - lo) / inst);
      }
      printf(" %10.2f", worst);
    }
    printf("\n");
  }
  printf("(8 waves per SIMD: two 1024-thread workgroups; if the dispatcher puts them on different CUs the column repeats the 4-wave figure)\n");
  return 0;
}
