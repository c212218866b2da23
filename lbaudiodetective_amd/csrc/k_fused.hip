// k_fused.hip -- fused per-frame fingerprint kernel (placeholder until the specialisation lands)
#include "internal.hpp"

namespace lbad {

bool fused_supported(const Plan&) { return false; }

hipError_t launch_fused(const Plan&, const float*, uint64_t, uint64_t, uint32_t, uint32_t*, hipStream_t) {
    return hipErrorNotSupported;
}

}  // namespace lbad
