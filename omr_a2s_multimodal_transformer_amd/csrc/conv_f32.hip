// fp32 (parity mode) instantiations of the conv3x3 MFMA kernel (see conv3x3_mfma.h)
#include "conv3x3_mfma.h"
int omr_conv3x3_dispatch_f32(const omr_conv::ConvArgs& a, hipStream_t s) { return omr_conv::dispatch_conv<float>(a, s); }
