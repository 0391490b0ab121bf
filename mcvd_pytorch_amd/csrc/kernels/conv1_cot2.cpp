// instantiation unit: 1x1 conv, 64-channel cout tile (split out for parallel compilation)
#include "conv_mfma.h"
namespace mcvd {
int conv1_cot2(const ConvArgs& a, int shape, hipStream_t s) { return conv_mfma_dispatch_shape<1, 32, 2>(a, shape, s); }
}  // namespace mcvd
