// pcr/io/tile_state_io.h -- the reference's tile-state checkpoint format (`.pcrt`), drop-in for
// include/pcr/io/tile_state_io.h:10-55.
//
//   header, 36 bytes, packed, little-endian:
//     u32 magic "PCRT" | u32 version = 1 | i32 tile_row | i32 tile_col | i32 cols | i32 rows |
//     i32 state_floats | u8 reduction (ReductionType) | u8 reserved[7] = 0
//   body: float[state_floats * cols * rows], band-sequential, tile-local row-major.
//
// Files written here are readable by the reference and vice versa (tests/test_tile_state.py checks
// both directions byte for byte against the reference's own src/io/tile_state_io.cpp).
#pragma once

#include "pcr/core/types.h"

#include <string>

namespace pcr {

Status write_tile_state(const std::string& path, TileIndex tile, int cols, int rows, int state_floats,
                        ReductionType type, const float* state);
Status read_tile_state(const std::string& path, TileIndex& tile, int& cols, int& rows, int& state_floats,
                       ReductionType& type, float* state);
Status read_tile_state_header(const std::string& path, TileIndex& tile, int& cols, int& rows,
                              int& state_floats, ReductionType& type);
/// "<dir>/tile_RRRR_CCCC.pcrt" (zero-padded to 4 digits)
std::string tile_state_filename(const std::string& dir, TileIndex tile);

}  // namespace pcr
