// csrc/raster_params.hpp — launch parameters of the tile rasterizer, shared by raster.hip and api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef CRH_XCD_BLOCK_LOG2
#define CRH_XCD_BLOCK_LOG2 3 // the raster kernels' XCD blocks are 8x8 tiles (the host mirrors the mapping: api.hip order_tiles_heavy_first)
#endif

namespace crh {

struct PrimRec;
struct PrimProj;

// One Shape::render(Stencil) optionally followed by one cover operation of the same Shape and instance (the host merges adjacent
// draws of a recorded pass); the unit the setup and binning kernels are launched over.
struct DrawItem {
    uint32_t shape, instance;
    uint32_t ops;  // bit 0: stencil phase present; bits 4-6: cover operation + 1 (crh_render_op), 0 = none
    uint32_t refs; // bits 0-7 clip depth of the stencil phase, 8-15 clip depth of the cover phase, 16-23 alpha layer
};

constexpr uint32_t kTessStatusWord = 126; // ... the status word of the optimistic tessellation a pass drew (api.hip: crh_scene::optimistic), copied in by the host side
constexpr uint32_t kExtraTurnsWord = 76; // of RasterParams::overflow: behind the 8 flag words and the 64 cursors of the pair sub-streams
struct RasterParams {
    uint32_t width, height, tiles_x, tiles_y, n_tiles; // 16x16-pixel tiles
    uint32_t winding_mask;
    uint32_t clip_mask_count; // (1 << clip_nesting_counter_bits) - 1: clip depths are compared as plain counters
    const DrawItem* items;    // [n_items], or nullptr: item i = Shape i, instance i, Stencil + Color at clip depth 0
    uint32_t n_items;
    uint32_t load_existing; // 0: the frame was cleared (LoadOp::Clear), 1: composite over the resolved image already there
    const float* transforms; // [n_instances][16] column-major
    const float* colors;     // [n_instances][4] straight alpha
    uint32_t* tile_count;    // [n_tiles], in one allocation [tile_cursor | tile_count | overflow]: one memset clears what a pass needs
    uint32_t* tile_cursor;   // [n_tiles]
    uint32_t* tile_offset;   // [n_tiles + 1]
    uint32_t* tile_list;     // [pair_capacity] prim ids, grouped by tile
    uint32_t pair_capacity;
    uint32_t* overflow;      // [0] pair capacity exceeded, [1] required pairs, [2] a tile list is longer than LDS can sort, [3] longest tile list, [5] a region of the pair stream is full,
                             // [6] items queued for k_bin_edges, [7] an edge with a non-finite endpoint: the edge pass cannot draw this frame
    uint32_t sort_capacity;  // primitives per tile the raster kernel's LDS sort buffer holds (a power of two)
    const uint32_t* shape_ncand;      // [n_items] candidate triangles per item
    uint32_t* shape_prim_begin;       // [n_items + 1] contiguous primitive ids per item, ascending in draw order
    uint32_t* scan_scratch;           // block sums of the scans
    PrimRec* prim_rec;                // [prim capacity] 128-byte set-up triangles
    uint32_t prim_capacity;
    uint8_t* rgba8;                   // [height][width][4] unorm8, or [height][width][4] binary16 when format == CRH_FORMAT_RGBA16F
    uint32_t format;                  // CRH_FORMAT_* of the target
    // ---- the general pass (perspective instances, depth test): selects the OPS variant of the raster kernel
    uint32_t general;                 // 1: some instance is not plain (clip.w != 1 or clip.z not a constant in [0, 1]) or depth is tested / written
    PrimProj* prim_proj;              // [prim capacity] 1/w and z/w planes of the primitives of projective instances, or nullptr
    float* depth;                     // [height][width][samples] depth attachment, or nullptr: no depth test
    uint32_t depth_pass_mask;         // bit 0: fragment < stored passes, bit 1: ==, bit 2: >, bit 3: Always (crh_compare as a mask)
    uint32_t depth_write;             // Configuration::depth_write_enabled
    uint32_t cull_mode;               // crh_cull of the colour cover
    uint32_t debug;                   // CRH_RASTER_DEBUG (tools only)
    uint32_t occlude;                 // every colour of the pass has 0 <= alpha <= 1 and |rgb| <= 1e30: a tile's list may be started late (k_raster_edges)
    // ---- the edge pass (raster_edges.hip): the plain Stencil + Color pass binned in ONE traversal
    uint8_t* slots;                   // [slot_capacity][32 B] primitive heap: set-up triangles (4 slots), boundary edges and per-item cover slots (1 slot)
    uint32_t slot_capacity;
    const uint32_t* slot_begin;       // [n_items + 1] first slot of every item (a multiple of 4); slot numbers ascend in draw order and are the sort keys
    uint32_t* pair_tile;              // [pair_capacity] (tile, position in the tile's list, key) in the order the binning waves produced them ...
    uint32_t* pair_pos;
    uint32_t* pair_key;               // ... scattered into tile_list by k_scatter once the tile offsets are known
    uint32_t* pair_cursor;            // pairs written so far (one atomic per flushed block of a wave)
    uint32_t hint_tris, hint_edges;   // about how many stroke / curve triangles and boundary edges the pass' items have in total (0: unknown) — sizes the batches of k_bin_flat
    // Direct tile lists (edge pass, frames after the first verified one): the lists keep the places of the previous frame — tile_base[t] =
    // exclusive prefix of (c + c / 2 + 64), c = the longest count within three tiles, of the pass before (k_tile_caps) — and the binning kernels store every key where it belongs, straight from
    // their stages: no pair stream, no scan of the counts, no scatter kernel. A tile that outgrows its place sets overflow[0]; the host
    // draws the frame again the exact way and re-bases the lists.
    uint32_t direct;
    uint32_t skip_queue;              // the verified pass of this frame queued nothing for k_bin_edges<S, true>: its launch is left out (overflow[6] != 0 then means: draw again)
    const uint32_t* tile_base;        // [n_tiles + 1]
    uint32_t long_lists;              // the frame's tile lists hold many entries on average: k_raster_edges looks for its late start across chunks (host: crh_frame::mean_list)
    uint32_t* bin_queue;              // [n_items] items k_bin_flat hands on to k_bin_edges (their number: overflow[6])
    const uint32_t* tile_order;       // [workgroups of the raster grid] the tile each workgroup of the edge pass' raster kernels draws (0xFFFFFFFF: none), or nullptr: the kernels' own XCD-aware order
    // k_bin_flat's batches by what they cost (round 4): a verified pass writes, per item, what the item takes of a batch's tables
    // (item_cost[2 i] = tile cells of its rectangle, [2 i + 1] = triangles | edges << 9 | 1 << 29 if its hull strip folds | 1 << 31 if it is not binned there at all); the
    // host cuts the items into runs that fill ONE batch each (flat_batches, raster_edges.hip) and later passes start one workgroup per run.
    uint32_t* item_cost;              // [2 n_items] or nullptr
    const uint32_t* bin_batches;      // [2 n_bin_batches] first item of every run and the one behind its last, the long runs first, or nullptr: equal numbers of items
    uint32_t n_bin_batches;           // (overflow[kExtraTurnsWord]: turns beyond the first that the runs' workgroups needed — stale costs)
    uint32_t rows;                    // the edge pass' lists are drawn by k_raster_rows (winding numbers accumulated in LDS, lanes over (entry, sample row)): the host measured it to be the faster kernel for this Scene (msaa 1, no strokes)
    uint32_t order_places;            // with tile_order: the places of the order that hold a tile (a multiple of 8; the grid of the raster kernels), 0: all of them
    uint32_t slab_ty0, slab_ty1;      // the tile rows [slab_ty0, slab_ty1) this pass draws (crh_frame_set_tile_rows: the tile split of the multi-GPU path — every rank bins everything and draws its slab); the other tiles are left alone
    const uint8_t* item_elsewhere;    // [n_items] 1: the item misses this pass' slab (k_slab_items), or nullptr
    const float* shape_bounds;        // [n_shapes][4] min x, min y, max x, max y of everything a Shape draws, in its own coordinates (k_shape_bounds), or nullptr — a pass with a slab:
                                      // the binning kernels leave out an item whose box on the frame misses the slab's rows BEFORE they set its primitives up
    // ---- pass state that outlives a pass (renderer.rs:148-158, 257-266, 932-985: the stencil attachment and the alpha layers are caller-owned and
    //      persist from one Shape::render call to the next, whatever Shape it belongs to). A frame whose recorded pass ends with state left over
    //      (an open Clip, a Stencil without its cover, a saved alpha context) keeps it in HBM from then on until it is cleared: the OPS raster
    //      kernel starts every tile from these planes and leaves them behind (api.hip crh_frame::carry).
    uint8_t* state_stencil;           // [height][width][samples] the stencil byte (clip nesting counter << winding bits | winding counter), or nullptr: registers only
    float* state_alpha;               // [alpha layers][height][width][samples] the saved alphas (the R8 targets of renderer.rs:892-927, f32 like the colours)
    float* state_color;               // [height][width][samples][4] the colour attachment per SAMPLE, f32 (the resolved image keeps one value per pixel)
    uint32_t state_load;              // the planes hold what earlier passes left (0: first pass with state — the planes were initialised, nothing to read but the colours of a frame that was not cleared)
    uint32_t state_layers;            // alpha layers in state_alpha (Configuration::alpha_layer_count, <= 4)
    uint32_t winding_bits;            // Configuration::winding_counter_bits (the shift of the clip nesting counter inside the stencil byte)
    uint32_t fill_cells;              // no tile list of this frame has shown 16 384 entries: k_raster_fill's packed counters (fill + 65536 * hull per sample) are exact; 0: k_raster_edges
};

} // namespace crh
