// bf16 instantiations of the conv3x3 MFMA kernel (see conv3x3_mfma.h)
#include "conv3x3_mfma.h"
int omr_conv3x3_dispatch_bf16(const omr_conv::ConvArgs& a, hipStream_t s) { return omr_conv::dispatch_conv<bf16>(a, s); }
