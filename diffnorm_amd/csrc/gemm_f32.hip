// dn_conv_gemm kernels instantiated for F32 operands (see gemm_kernels.h).
#include "gemm_kernels.h"

namespace dn {
int gemm_dispatch_f32(const DnGemmParams& p, hipStream_t s) { return dispatch_epi<F32>(p, s); }
}  // namespace dn
