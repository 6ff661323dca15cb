// dn_conv_gemm kernels instantiated for DN_BF16X3 split operands (three bf16 MFMAs per product; see gemm_kernels.h).
#include "gemm_kernels.h"

namespace dn {
int gemm_dispatch_x3(const DnGemmParams& p, hipStream_t s) { return dispatch_epi<BF16X3>(p, s); }
}  // namespace dn
