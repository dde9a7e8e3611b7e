// Asynchronous global -> LDS copies (gfx950 global_load_lds_dwordx4) for the tile rings of gemm_dma.hip and
// conv_wgrad_dma.hip.  Semantics verified on hardware (scratch/dmatest.hip): a wave instruction deposits lane i's 16 bytes
// at LDS byte address M0 + 16 i (any M0 up to the 160 KB of a CU), inactive lanes write nothing, completion is tracked by
// vmcnt in issue order.
#pragma once
#include <hip/hip_runtime.h>

#pragma clang diagnostic ignored "-Winline-asm"

// One wave instruction: lane i fetches 16 bytes from its own global address into LDS byte address lds_wave_base + 16 i.
__device__ __forceinline__ void dma16(const void* gptr, unsigned lds_wave_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :: "v"(gptr), "s"(__builtin_amdgcn_readfirstlane(lds_wave_base)) : "memory", "m0");
}
// Wait until at most PENDING of this wave's DMA instructions are still in flight (they complete in order), make this
// wave's LDS writes visible, then the workgroup barrier.  The compiler does not track the asm DMA, hence the explicit
// count; the "memory" clobber keeps LDS accesses from moving across.
template <int PENDING> __device__ __forceinline__ void dma_wait_barrier() {
    static_assert(PENDING >= 0 && PENDING < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(PENDING) : "memory");
}
__device__ __forceinline__ void dma_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ unsigned lds_address(const void* p) { return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)p; }
