// Hardware check of global_load_lds_dwordx4 semantics on gfx950: LDS address = M0 + lane * 16, exec-masked lanes write nothing.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void dma16(const void* gptr, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gptr), "s"(__builtin_amdgcn_readfirstlane(lds_base)) : "memory", "m0");
}
__global__ void k(const uint4* __restrict__ g, uint4* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char raw[];   // 150 KB: the DMA target sits above 64 KB (M0 needs > 16 bits)
    uint4* sm = reinterpret_cast<uint4*>(raw + 100 * 1024);
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    sm[t] = make_uint4(0xdeadbeefu, 0, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)raw + 100 * 1024;
    if ((lane % 5) != 4) dma16(g + w * 64 + (lane * 7) % 64, base + w * 1024);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    out[t] = sm[t];
}
int main() {
    std::vector<uint4> h(256);
    for (int i = 0; i < 256; ++i) h[i] = make_uint4(i, i * 3, i * 5, i * 7);
    uint4 *d, *o; hipMalloc(&d, 4096); hipMalloc(&o, 4096);
    hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 150 * 1024, 0, d, o);
    std::vector<uint4> r(256);
    if (hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost) != hipSuccess) { printf("FAIL memcpy\n"); return 1; }
    int bad = 0;
    for (int t = 0; t < 256; ++t) {
        int lane = t & 63, w = t >> 6;
        unsigned exp = (lane % 5) != 4 ? (unsigned)(w * 64 + (lane * 7) % 64) : 0xdeadbeefu;
        if (r[t].x != exp || ((lane % 5) != 4 && r[t].w != exp * 7)) { if (bad < 8) printf("t=%d got %u exp %u\n", t, r[t].x, exp); ++bad; }
    }
    printf("dmatest: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
    return bad != 0;
}
