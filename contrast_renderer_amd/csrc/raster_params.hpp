// csrc/raster_params.hpp — launch parameters of the tile rasterizer, shared by raster.hip and api.hip.
#pragma once
#include <hip/hip_runtime.h>
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
    uint32_t* tile_count;    // [n_tiles], immediately followed by tile_cursor (one memset clears both)
    uint32_t* tile_offset;   // [n_tiles + 1]
    uint32_t* tile_cursor;   // [n_tiles]
    uint32_t* shape_ncand;      // [n_shapes] candidate triangles per Shape
    uint32_t* shape_prim_begin; // [n_shapes + 1]
    uint32_t* scan_scratch;     // block sums of the two scans
    struct PrimRec* prim_rec;   // [prim capacity] 128-byte set-up triangles
    ushort4* prim_box;          // [prim capacity] inclusive pixel box x0 x1 y0 y1, x0 == 0xFFFF = nothing to draw
    uint32_t* tile_list;     // [pair_capacity]
    uint32_t pair_capacity;
    uint32_t* overflow;      // [2]: {flag, required pairs}
    uint8_t* rgba8;          // [height][width][4]
    uint32_t debug;          // CRH_RASTER_DEBUG ablation bits (tools only): 1 stop after the sort, 2 skip coverage, 4 count work into overflow[2..]
};

} // namespace crh
