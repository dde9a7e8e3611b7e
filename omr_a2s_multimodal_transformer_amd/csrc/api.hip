// ABI version stamp of libomr_hip.so (bumped whenever include/omr_hip.h changes incompatibly).
#include "omr_hip.h"
extern "C" int omr_abi_version(void) { return 2; }
