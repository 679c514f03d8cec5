// scatter_binned.hip -- binned LDS-tile scatter path (placeholder: not yet enabled).
#include "engine.hpp"

namespace pcrhip {

bool binned_point_supported(const pcr_hip_engine*, uint32_t) { return false; }
int binned_point(pcr_hip_engine*, uint32_t, const PlanesDev&, const double*, const double*, const float*, uint64_t) {
    return fail(PCR_HIP_NOT_IMPLEMENTED, "binned point path not built");
}
bool binned_glyph_supported(const pcr_hip_engine*, const GlyphDev&, uint32_t) { return false; }
int binned_glyph(pcr_hip_engine*, const GlyphDev&, uint32_t, const PlanesDev&, const double*, const double*,
                 const float*, uint64_t) {
    return fail(PCR_HIP_NOT_IMPLEMENTED, "binned glyph path not built");
}

}  // namespace pcrhip
