// MFMA forward of the kernel convolution for the shapes the model uses (placeholder until built).
#include "kgnn_launch.h"

namespace mkgnn {

bool mfma_forward_supported(int d, int F, int E, int L) { return false; }

hipError_t launch_forward_mfma(int d, const FwdArgs& a, hipStream_t st) { return hipErrorNotSupported; }

}  // namespace mkgnn
