// csrc/raster_params.hpp — launch parameters of the band rasterizer, shared by raster.hip and api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace crh {

struct PrimRec;

struct RasterParams {
    uint32_t width, height, tiles_x, tiles_y, n_tiles, n_bands; // 16x16 tiles, 4 bands (16x4 pixels) per tile: bin = tile * 4 + band
    uint32_t winding_mask;
    uint32_t load_existing; // 0: the frame was cleared (LoadOp::Clear), 1: composite over the resolved image already there
    const float* transforms; // [n_shapes][16] column-major
    const float* colors;     // [n_shapes][4] straight alpha
    uint32_t* tile_count;    // [n_tiles], immediately followed by tile_cursor (one memset clears both)
    uint32_t* tile_cursor;   // [n_tiles]
    uint32_t* tile_offset;   // [n_tiles + 1]
    uint32_t* tile_list;     // [pair_capacity] prim id << 8 | full-band mask << 4 | band mask
    uint32_t pair_capacity;
    uint32_t* overflow;      // [0] pair capacity exceeded, [1] required pairs, [2..] tools counters
    const uint32_t* shape_ncand;      // [n_shapes] candidate triangles per Shape (written by the hull kernel)
    uint32_t* shape_prim_begin;       // [n_shapes + 1]
    uint32_t* scan_scratch;           // block sums of the scans
    PrimRec* prim_rec;                // [prim capacity] 128-byte set-up triangles
    uint32_t prim_capacity;
    uint8_t* rgba8;                   // [height][width][4]
    uint32_t debug;                   // CRH_RASTER_DEBUG (tools only)
};

} // namespace crh
