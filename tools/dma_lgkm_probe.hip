// Development aid: does an outstanding global_load_lds (LDS-DMA) hold up `s_waitcnt lgkmcnt(0)`?  One wave issues NDMA
// LDS-DMA instructions from a cold 64 MB buffer, then a ds_read of an unrelated LDS word, and stamps s_memtime after
// (a) s_waitcnt lgkmcnt(0) and (b) s_waitcnt vmcnt(0).  If (a) ~ LDS latency and (b) ~ memory latency the DMA is on the VM
// counter only; if (a) ~ (b) the compiler's lgkmcnt(0) waits serialise LDS reads behind every DMA in flight.
// Build: hipcc -O3 --offload-arch=gfx950 tools/dma_lgkm_probe.hip -o tools/dma_lgkm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(64) void probe(const uint4* __restrict__ src, long stride16, int ndma, unsigned long long* out, int* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[32 * 1024];
    __shared__ int other[64];
    other[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)lds;
    const uint4* p = src + (long)blockIdx.x * 16384 + threadIdx.x;          // 256 KB of the 64 MB buffer per block
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ndma; ++i) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(p + (long)i * 1024), "s"(base + (unsigned)i * 1024u) : "memory", "m0");
    }
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(uintptr_t)(const __attribute__((address_space(3))) int*)&other[threadIdx.x]) : "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = t2 - t0; }
    sink[blockIdx.x * 64 + threadIdx.x] = v + lds[threadIdx.x * 16];
}

int main() {
    const long n16 = 64L * 1024 * 1024 / 16;
    uint4* src; hipMalloc(&src, n16 * 16); hipMemset(src, 1, n16 * 16);
    unsigned long long* out; hipMalloc(&out, 256 * 2 * 8);
    int* sink; hipMalloc(&sink, 256 * 64 * 4);
    for (int ndma : {1, 4, 16}) {
        for (int blocks : {1, 256}) {
            // cold-ish source: a different 64-lane x 16-byte gather per DMA, 64 KB apart
            hipLaunchKernelGGL(probe, dim3(blocks), dim3(64), 0, 0, src, 0L, ndma, out, sink);
            hipDeviceSynchronize();
            unsigned long long h[512];
            hipMemcpy(h, out, blocks * 16, hipMemcpyDeviceToHost);
            double a = 0, b = 0;
            for (int i = 0; i < blocks; ++i) { a += h[2 * i]; b += h[2 * i + 1]; }
            printf("ndma %2d blocks %3d: after lgkmcnt(0) %7.0f cycles, after vmcnt(0) %7.0f cycles (s_memtime ticks, mean over blocks)\n", ndma, blocks, a / blocks, b / blocks);
        }
    }
    return 0;
}
