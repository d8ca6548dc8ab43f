// srbdqp_mfma.hpp -- kernel variant v1 (fp64 MFMA contraction + tiled Cholesky inverse).  Placeholder until built.
#pragma once
#include "srbdqp_common.hpp"
namespace srbdqp {
constexpr bool kMfmaReady = false;
template <int N> struct MfmaTraits { static constexpr bool supported = false; static constexpr size_t lds_bytes = 0; static constexpr const char* name = "mfma_unavailable"; };
template <int N> __global__ void srbdqp_mfma_kernel(KArgs a) {}
}  // namespace srbdqp
