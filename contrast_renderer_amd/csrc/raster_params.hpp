// csrc/raster_params.hpp — launch parameters of the tile rasterizer, shared by raster.hip and api.hip.
#pragma once
#include <stdint.h>

namespace crh {

struct RasterParams {
    uint32_t width, height, tiles_x, tiles_y, n_tiles;
    uint32_t winding_mask;
    uint32_t load_existing; // 0: the frame was cleared (LoadOp::Clear), 1: composite over the resolved image already there
    const float* transforms; // [n_shapes][16] column-major
    const float* colors;     // [n_shapes][4] straight alpha
    uint32_t* shape_rect;    // [n_shapes] packed tile rectangle (lo), 0xFFFFFFFF = empty
    uint32_t* shape_rect_hi;
    uint32_t* tile_count;    // [n_tiles]
    uint32_t* tile_offset;   // [n_tiles + 1]
    uint32_t* tile_cursor;   // [n_tiles]
    uint32_t* tile_list;     // [pair_capacity]
    uint32_t pair_capacity;
    uint32_t* overflow;      // [2]: {flag, required pairs}
    uint8_t* rgba8;          // [height][width][4]
};

} // namespace crh
