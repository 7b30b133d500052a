// Probe: do 4-byte and 12-byte global loads work from 2-byte-aligned addresses on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(3))) unsigned u32x3;
__global__ void probe(const unsigned short* src, unsigned* out1, unsigned* out3) {
  const int t = threadIdx.x;
  const unsigned short* p = src + 1 + 6 * t;                 // byte offset 2 + 12 t: 2 mod 4
  out1[t] = *reinterpret_cast<const unsigned*>(p);
  const u32x3 v = *reinterpret_cast<const u32x3*>(p);
  out3[3 * t] = v[0]; out3[3 * t + 1] = v[1]; out3[3 * t + 2] = v[2];
}
int main() {
  std::vector<unsigned short> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = (unsigned short)i;
  unsigned short* d; unsigned *o1, *o3;
  hipMalloc(&d, 2048); hipMalloc(&o1, 256); hipMalloc(&o3, 768);
  hipMemcpy(d, h.data(), 2048, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(d, o1, o3);
  std::vector<unsigned> r1(64), r3(192);
  hipError_t e = hipMemcpy(r1.data(), o1, 256, hipMemcpyDeviceToHost);
  hipMemcpy(r3.data(), o3, 768, hipMemcpyDeviceToHost);
  printf("status %s\n", hipGetErrorString(e));
  int ok1 = 1, ok3 = 1;
  for (int t = 0; t < 64; ++t) {
    const unsigned e0 = (1 + 6 * t) | ((2 + 6 * t) << 16);
    if (r1[t] != e0) ok1 = 0;
    for (int j = 0; j < 3; ++j) {
      const unsigned ej = (1 + 6 * t + 2 * j) | ((2 + 6 * t + 2 * j) << 16);
      if (r3[3 * t + j] != ej) ok3 = 0;
    }
  }
  printf("unaligned dword load: %s, unaligned dwordx3 load: %s\n", ok1 ? "OK" : "WRONG", ok3 ? "OK" : "WRONG");
  return 0;
}
