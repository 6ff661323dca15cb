// dn_conv_gemm kernels instantiated for IEEE-half operands (DN_F16: v_mfma_f32_16x16x32_f16; see gemm_kernels.h).
#include "gemm_kernels.h"

namespace dn {
int gemm_dispatch_f16(const DnGemmParams& p, hipStream_t s) { return dispatch_epi<F16>(p, s); }
}  // namespace dn
