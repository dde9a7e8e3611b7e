// Arguments of the 3x3 weight-gradient kernels (conv.hip: generic staged kernel; conv_wgrad_dma.hip: bf16 LDS-DMA kernel).
#pragma once
#include <hip/hip_runtime.h>

struct WgradArgs {
    const void* x; const void* dy; float* dw; float* db;
    const float* mean; const float* rstd;
    int B, Hr, Wr, CIN, Ho, Wo, COUT, sh, sw, tiles_w, tiles_h;
#ifdef OMR_WGRAD_DEBUG
    int dbg = 0;
#endif
};

// bf16 weight gradient with asynchronous global->LDS staging.  Returns OMR_ERR_UNSUPPORTED for shapes it does not
// cover (the caller then uses the generic kernel).
int omr_wgrad_dma_bf16(const WgradArgs& a, hipStream_t s);
