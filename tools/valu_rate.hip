// Development aid: issue cost of the integer ops a counter hash is built from (one wave per SIMD, dependent chains of 4 lanes
// of ILP).  Build: hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o tools/valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(uint32_t* out, uint32_t c, int iters) {
    uint32_t x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (OP == 0) { x0 = x0 * c; x1 = x1 * c; x2 = x2 * c; x3 = x3 * c; }
            if (OP == 1) {
                x0 = __umul24(x0, c); x1 = __umul24(x1, c);
                x2 = __umul24(x2, c); x3 = __umul24(x3, c);
            }
            if (OP == 2) { x0 = x0 + c; x1 = x1 ^ c; x2 = x2 + c; x3 = x3 ^ c; }
            if (OP == 3) { x0 ^= x0 >> 15; x1 ^= x1 >> 15; x2 ^= x2 >> 15; x3 ^= x3 >> 15; }
            if (OP == 4) { x0 = __umulhi(x0, c); x1 = __umulhi(x1, c); x2 = __umulhi(x2, c); x3 = __umulhi(x3, c); }
            if (OP == 5) { x0 = __umul24(x0, c) + x1; x1 = __umul24(x1, c) + x2; x2 = __umul24(x2, c) + x3; x3 = __umul24(x3, c) + x0; }
            asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3;
}

template <int OP> void run(const char* name, uint32_t* d) {
    const int iters = 4096, blocks = 256 * 4;      // one 256-thread block per SIMD set -> 4 waves per CU ... x4 for occupancy 4/SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    rate_kernel<OP><<<blocks, 256>>>(d, 0x9E3779B1u, 16);
    hipEventRecord(e0);
    rate_kernel<OP><<<blocks, 256>>>(d, 0x9E3779B1u, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * 4 /*waves*/ * iters * 16 * 4;          // wave-instructions (OP 3 and 5: two per statement)
    printf("%-28s %8.3f ms  %.2f ns per wave-instruction-slot per SIMD (x1024 SIMDs)\n", name, ms, ms * 1e6 / (ops / 1024.0));
}

int main() {
    uint32_t* d; hipMalloc(&d, 256 * 4 * 256 * 4);
    run<2>("v_add/v_xor", d);
    run<0>("v_mul_lo_u32", d);
    run<1>("v_mul_u32_u24", d);
    run<4>("v_mul_hi_u32", d);
    run<3>("x ^= x >> 15 (2 ops)", d);
    run<5>("v_mad_u32_u24", d);
    return 0;
}
