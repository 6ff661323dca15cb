// dn_conv_gemm kernels instantiated for BF16 operands (see gemm_kernels.h).
#include "gemm_kernels.h"

namespace dn {
int gemm_dispatch_bf16(const DnGemmParams& p, hipStream_t s) { return dispatch_epi<BF16>(p, s); }
}  // namespace dn
