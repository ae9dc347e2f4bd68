// wnv_ring.hip -- pipelined, weight-stationary ring kernel (placeholder until the first GPU round trip
// of the generic kernel is green; see DESIGN.md section 5 for the design it will hold).
#include "wnv_ring.h"

bool wnv_ring_supported(const wnv_config&, int) { return false; }
const char* wnv_ring_why_not(const wnv_config&, int) { return "ring kernel not built in this revision"; }
wnv_status wnv_ring_generate(WnvRingState**, int, const wnv_config&, const TensorStore&, const WnvGenArgs&,
                             hipStream_t, std::string& err) {
    err = "ring kernel not built in this revision";
    return WNV_ERR_UNSUPPORTED;
}
void wnv_ring_destroy(WnvRingState*) {}
