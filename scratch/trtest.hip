#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
// LDS image: [pixel row r (0..15)][channel c (0..31)], pitch 36 elements; value = r*32 + c (exact in bf16 up to 256; use r<8)
__global__ void k(float* out) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[16 * 36];
  for (int i = threadIdx.x; i < 16 * 36; i += 64) { int r = i / 36, c = i % 36; lds[i] = (__bf16)(float)(c < 32 ? (r * 32 + c) : 999); }
  __syncthreads();
  int lane = threadIdx.x;
  int q = (lane & 15) >> 2, p = lane & 3, cb = (lane >> 4) & 1, h = lane >> 5;
  __bf16* ptr = &lds[(4 * h + q) * 36 + 16 * cb + 4 * p];
  bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)ptr);
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (float)v[e];
}
int main() {
  float* d; hipMalloc(&d, 256 * 4); k<<<1, 64>>>(d); float h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int e = 0; e < 4; ++e) printf(" (r%d,c%2d)", (int)h[l*4+e] / 32, (int)h[l*4+e] % 32); printf("\n"); }
  return 0;
}
