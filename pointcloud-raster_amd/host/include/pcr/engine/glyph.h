// pcr/engine/glyph.h -- how a point is painted onto the raster (drop-in for the reference's
// include/pcr/engine/glyph.h).
#pragma once

#include <cstdint>
#include <string>

namespace pcr {

enum class GlyphType : uint8_t { Point, Line, Gaussian };

struct GlyphSpec {
    GlyphType type = GlyphType::Point;

    // Line: Bresenham segment of length 2*half_length (world units) along `direction` (radians).
    // A channel name that is empty, absent from the cloud or not Float32 selects the default.
    std::string direction_channel;
    float default_direction = 0.0f;
    std::string half_length_channel;
    float default_half_length = 1.0f;

    // Gaussian: sigmas in world units, rotation in radians.
    std::string sigma_x_channel;
    float default_sigma_x = 1.0f;
    std::string sigma_y_channel;
    float default_sigma_y = 1.0f;
    std::string rotation_channel;
    float default_rotation = 0.0f;

    float max_radius_cells = 32.0f;     // footprint clamp, cells
    bool normalize_weights = false;     // accepted, not applied (as in the reference)
};

}  // namespace pcr
