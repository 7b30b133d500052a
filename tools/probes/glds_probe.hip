// Probe: where do the bytes of global_load_lds_dwordx4 land in LDS?  (build: hipcc --offload-arch=gfx950)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((address_space(3))) unsigned char* lds_b;
typedef const __attribute__((address_space(1))) unsigned char* glb_b;
__global__ void probe(const unsigned* src, unsigned* out, int base_off) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char sm[];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) ((unsigned*)sm)[i] = 0xdeadbeef;
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // lane l of wave w loads the 16 bytes {tag(w,l,0..3)}
  __builtin_amdgcn_global_load_lds((glb_b)(src + (wave * 64 + lane) * 4), (lds_b)(sm + base_off + wave * 1024), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) out[i] = ((unsigned*)sm)[i];
}
int main() {
  std::vector<unsigned> h(1024);
  for (int t = 0; t < 256; ++t) for (int j = 0; j < 4; ++j) h[t * 4 + j] = (t << 8) | j;   // tag = thread<<8 | dword
  unsigned *d, *o; hipMalloc(&d, 4096); hipMalloc(&o, 16384);
  hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
  for (int base = 0; base <= 4096; base += 4096) {
    probe<<<1, 256, 16384>>>(d, o, base);
    std::vector<unsigned> r(4096); hipMemcpy(r.data(), o, 16384, hipMemcpyDeviceToHost);
    printf("base_off %d: first 24 dwords at base:", base);
    for (int i = 0; i < 24; ++i) printf(" %06x", r[base / 4 + i]);
    printf("\n");
    int linear = 1; for (int i = 0; i < 1024; ++i) if (r[base / 4 + i] != h[i]) { linear = 0; printf("  first mismatch at dword %d: got %06x want %06x\n", i, r[base / 4 + i], h[i]); break; }
    printf("  linear (lane*16) layout: %s\n", linear ? "YES" : "NO");
    int untouched = 0; for (int i = 0; i < 4096; ++i) untouched += r[i] == 0xdeadbeef; printf("  untouched dwords: %d of 4096\n", untouched);
  }
  return 0;
}
