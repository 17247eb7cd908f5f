// csrc/api.hip — the C ABI of include/contrast_hip.h: host-side orchestration only (validation, HBM buffers, launches,
// parity taps). Every arithmetic step of the hot path runs in the HIP kernels of tessellate.hip / raster.hip; there is no
// CPU fallback here and nothing under oracle/ is referenced.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "raster_params.hpp"
#include "scene.hpp"

namespace crh {
typedef void (*MarkFn)(void*, const char*, uint64_t);
void launch_tessellate(const SceneDev& s, hipStream_t stream, MarkFn mark, void* ctx, const uint64_t bytes[4], bool has_stroke, bool need_totals);
void launch_emit(const SceneDev& s, hipStream_t stream, MarkFn mark, void* ctx, const uint64_t bytes[4], bool has_stroke, bool big_shapes, const uint32_t* hull_queued);
void launch_build_elements(const UploadBuild& u, hipStream_t stream);
void launch_prim_ranges(const SceneDev& s, uint32_t* shape_ncand, uint32_t* shape_prim_begin, uint32_t* scratch, hipStream_t stream);
void launch_bin(const SceneDev& s, const RasterParams& r, uint32_t samples, hipStream_t stream, MarkFn mark, void* ctx, hipEvent_t after_setup);
void launch_fill(const SceneDev& s, const RasterParams& r, uint32_t samples, hipStream_t stream, MarkFn mark, void* ctx, hipEvent_t after_fill);
void launch_raster(const SceneDev& s, const RasterParams& r, uint32_t samples, hipStream_t stream, MarkFn mark, void* ctx, uint64_t raster_bytes, bool has_stroke);
void launch_item_ranges(const SceneDev& s, const RasterParams& r, uint32_t* item_ncand, uint32_t* item_prim_begin, uint32_t* scratch, hipStream_t stream);
void launch_composite(const uint8_t* const* layers_dev, uint32_t n_layers, uint64_t n_pixels, uint8_t* dst, hipStream_t stream);
void launch_state_colors_from_image(const RasterParams& r, uint32_t samples, hipStream_t stream);
// raster_edges.hip: the plain Stencil + Color pass as boundary edges + backdrop, binned in one traversal
void launch_slot_ranges(const SceneDev& s, const RasterParams& r, uint32_t n_items, uint32_t* item_nslots, uint32_t* slot_begin, uint32_t* scratch, hipStream_t stream);
void launch_bin_edges(const SceneDev& s, const RasterParams& r, uint32_t samples, hipStream_t stream, MarkFn mark, void* ctx, hipEvent_t after_bin);
void launch_scatter(const RasterParams& r, hipStream_t stream, MarkFn mark, void* ctx);
void launch_shape_bounds(const SceneDev& s, float* bounds, hipStream_t stream);
void launch_slab_items(const RasterParams& r, uint8_t* elsewhere, hipStream_t stream);
void launch_plain_ranges(const SceneDev& s, uint32_t* shape_ncand, uint32_t* shape_prim_begin, uint32_t* shape_nslots, uint32_t* shape_slot_begin, uint32_t* scratch0, uint32_t* scratch1, hipStream_t stream);
bool bin_itemwise(const RasterParams& r);
void flat_batches(const uint32_t* cost, uint32_t n_items, std::vector<uint32_t>& runs);
void flat_batch_limits(uint32_t n_items, uint32_t limits[4]);
void launch_tile_bases(const uint32_t* tile_count, uint32_t* caps, uint32_t* tile_base, uint32_t* scratch, uint32_t n_tiles, uint32_t tiles_x, uint32_t radius, hipStream_t stream);
void launch_raster_edges(const SceneDev& s, const RasterParams& r, uint32_t samples, hipStream_t stream, MarkFn mark, void* ctx, uint64_t raster_bytes, bool has_stroke);
void launch_fmath(int fn, const float* a, const float* b, float* out, uint64_t n, hipStream_t stream);
} // namespace crh

using namespace crh;

namespace {
thread_local std::string g_error;

bool hip_ok(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    g_error = std::string(what) + ": " + hipGetErrorString(e);
    return false;
}
#define HIP_TRY(expr)                                  \
    do {                                               \
        if (!hip_ok((expr), #expr)) return CRH_ERR_HIP; \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap && p) return hipSuccess;
        if (p) {
            hipError_t e = hipFree(p);
            if (e != hipSuccess) return e;
            p = nullptr;
            cap = 0;
        }
        const size_t want = bytes < 256 ? 256 : bytes;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T>
    T* as() const {
        return static_cast<T*>(p);
    }
};

// Host -> device copies that must not stall the host: pageable memory makes hipMemcpyAsync wait for the stream, so the bytes are first
// copied into one of two pinned staging buffers (the previous use of that buffer — two uploads ago — is waited for, normally long done)
// and sent from there.
struct PinnedUpload {
    void* host[2] = {nullptr, nullptr};
    size_t cap[2] = {0, 0};
    hipEvent_t sent[2] = {nullptr, nullptr};
    int next = 0;
    hipError_t copy(void* dst, const void* src, size_t bytes, hipStream_t stream) {
        if (bytes == 0) return hipSuccess;
        const int k = next;
        next ^= 1;
        hipError_t e;
        if (!sent[k] && (e = hipEventCreateWithFlags(&sent[k], hipEventDisableTiming)) != hipSuccess) return e;
        if (cap[k] && (e = hipEventSynchronize(sent[k])) != hipSuccess) return e;
        if (cap[k] < bytes) {
            if (host[k]) (void)hipHostFree(host[k]);
            host[k] = nullptr;
            cap[k] = 0;
            if ((e = hipHostMalloc(&host[k], bytes, hipHostMallocDefault)) != hipSuccess) return e;
            cap[k] = bytes;
        }
        memcpy(host[k], src, bytes);
        if ((e = hipMemcpyAsync(dst, host[k], bytes, hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
        return hipEventRecord(sent[k], stream);
    }
    void release() {
        for (int k = 0; k < 2; ++k) {
            if (sent[k]) {
                (void)hipEventSynchronize(sent[k]);
                (void)hipEventDestroy(sent[k]);
            }
            if (host[k]) (void)hipHostFree(host[k]);
            host[k] = nullptr;
            sent[k] = nullptr;
            cap[k] = 0;
        }
    }
};

// Pinned staging memory a caller fills in place (crh_scene_upload builds the element stream straight into it): two arenas in turn, each guarded
// by the event behind the copies that read it, so the call that filled it need not wait for them.
struct PinnedArena {
    void* host[2] = {nullptr, nullptr};
    size_t cap[2] = {0, 0};
    hipEvent_t sent[2] = {nullptr, nullptr};
    int next = 0, cur = 0;
    hipError_t begin(size_t bytes, uint8_t** out) {
        const int k = next;
        next ^= 1;
        cur = k;
        hipError_t e;
        if (!sent[k] && (e = hipEventCreateWithFlags(&sent[k], hipEventDisableTiming)) != hipSuccess) return e;
        if (cap[k] && (e = hipEventSynchronize(sent[k])) != hipSuccess) return e; // (the copies out of it two uploads ago: long done)
        if (cap[k] < bytes) {
            if (host[k]) (void)hipHostFree(host[k]);
            host[k] = nullptr, cap[k] = 0;
            const size_t want = bytes + bytes / 4 + 4096;
            if ((e = hipHostMalloc(&host[k], want, hipHostMallocDefault)) != hipSuccess) return e;
            cap[k] = want;
        }
        *out = static_cast<uint8_t*>(host[k]);
        return hipSuccess;
    }
    hipError_t done(hipStream_t stream) { return hipEventRecord(sent[cur], stream); }
    void release() {
        for (int k = 0; k < 2; ++k) {
            if (sent[k]) {
                (void)hipEventSynchronize(sent[k]);
                (void)hipEventDestroy(sent[k]);
            }
            if (host[k]) (void)hipHostFree(host[k]);
            host[k] = nullptr, sent[k] = nullptr, cap[k] = 0;
        }
    }
};

struct Mark {
    hipEvent_t event;
    std::string name;
    uint64_t bytes;
    int lane; // 0 = raster, 1 = tessellation, 2 = binning stream: durations are taken between consecutive marks of one lane
};
} // namespace

struct crh_renderer {
    int n_cus = 256; // compute units of the device (sizes the resident raster grid)
    crh_config config;
    int device;
    hipStream_t stream;       // raster kernel, copies
    hipStream_t bin_stream;   // primitive setup + tile binning: frame N + 1's overlap frame N's raster kernel (double-buffered records / lists)
    hipStream_t tess_stream;  // tessellation: frame N + 1's (small, latency bound) kernels overlap frame N's binning and raster
    hipStream_t upload_stream; // instance data of the next frame, copied while the current one is being binned
    hipStream_t aux_stream;   // crh_composite_over: the multi-GPU slab composite of frame N while frame N + 1 is being rendered
    void* composite_table = nullptr; // device array of layer pointers (aux stream)
    uint32_t composite_table_capacity = 0;
    bool pipeline = true;     // CRH_NO_PIPELINE=1 runs everything on `stream`
    // CRH_RASTER_EXCLUSIVE (measurement only): the raster kernel has the GPU to itself — the binning of frame i + 1 and the tessellation of
    // frame i + 2 both start when the raster kernel of frame i has finished, and overlap each other. Slower (0.53 against 0.45 ms per step on
    // the benchmark scene): the lanes do gain from running beside the raster kernel, however little each of them gets there.
    bool raster_exclusive = false;
    hipEvent_t raster_events[2] = {nullptr, nullptr}; // the raster_done events of the latest and of the previous render
    std::vector<crh_scene*> scenes; // live scenes and frames: orphaned (renderer = nullptr) when the renderer goes first, so that their own
                                    // destruction — host bindings finalise in any order — never touches a freed renderer
    std::vector<crh_frame*> frames; // live frames: a frame whose overflow check is still pending is settled before the scene it shows changes
    uint64_t render_serial = 0; // render calls so far
    unsigned timing = 0; // lanes whose kernels are bracketed by events: bit 0 raster, 1 tessellation, 2 binning (crh_renderer_enable_timing)
    std::vector<hipEvent_t> event_pool;
    std::vector<Mark> marks;
    size_t events_used = 0;

    hipEvent_t next_event() {
        if (events_used == event_pool.size()) {
            hipEvent_t e;
            (void)hipEventCreate(&e);
            event_pool.push_back(e);
        }
        return event_pool[events_used++];
    }
    // marks accumulate over calls (so a benchmark can time K steps without a sync per step); kernel_times() drains them
    hipStream_t tessellation_stream() const { return pipeline ? tess_stream : stream; }
    hipStream_t binning_stream() const { return pipeline ? bin_stream : stream; }
    hipStream_t lane_stream(int lane) const { return lane == 1 ? tessellation_stream() : (lane == 2 ? binning_stream() : stream); }
    void begin_marks(int lane = 0) {
        if (!(timing & (1u << lane))) return;
        hipEvent_t e = next_event();
        (void)hipEventRecord(e, lane_stream(lane));
        marks.push_back({e, "", 0, lane});
    }
    static void mark_cb_bin(void* ctx, const char* name, uint64_t bytes) {
        crh_renderer* r = static_cast<crh_renderer*>(ctx);
        hipEvent_t e = r->next_event();
        (void)hipEventRecord(e, r->binning_stream());
        r->marks.push_back({e, name, bytes, 2});
    }
    MarkFn mark_fn_bin() const { return (timing & 4u) ? &crh_renderer::mark_cb_bin : nullptr; }
    static void mark_cb(void* ctx, const char* name, uint64_t bytes) {
        crh_renderer* r = static_cast<crh_renderer*>(ctx);
        hipEvent_t e = r->next_event();
        (void)hipEventRecord(e, r->stream);
        r->marks.push_back({e, name, bytes, 0});
    }
    static void mark_cb_tess(void* ctx, const char* name, uint64_t bytes) {
        crh_renderer* r = static_cast<crh_renderer*>(ctx);
        hipEvent_t e = r->next_event();
        (void)hipEventRecord(e, r->tessellation_stream());
        r->marks.push_back({e, name, bytes, 1});
    }
    MarkFn mark_fn() const { return (timing & 1u) ? &crh_renderer::mark_cb : nullptr; }
    MarkFn mark_fn_tess() const { return (timing & 2u) ? &crh_renderer::mark_cb_tess : nullptr; }
    hipError_t sync() { // every stream
        const hipError_t e = hipStreamSynchronize(tess_stream);
        const hipError_t b = hipStreamSynchronize(bin_stream);
        const hipError_t a = hipStreamSynchronize(aux_stream);
        (void)hipStreamSynchronize(upload_stream);
        const hipError_t f = hipStreamSynchronize(stream);
        return e != hipSuccess ? e : (b != hipSuccess ? b : (a != hipSuccess ? a : f));
    }
};

// One of two sets of instance data (transforms + colours). Its copy runs on the upload stream, behind the k_prim_setup that last read
// the set (read_done) and ahead of the one that reads it next (ready), so it overlaps the binning of the frame in between.
struct InstanceSlot {
    hipEvent_t read_done = nullptr, ready = nullptr;
    hipEvent_t read_event = nullptr; // what the next copy into this set waits for: read_done, or (borrowed) the Scene's event behind the binning kernel that read it
    bool was_read = false, was_written = false;
    hipError_t init() {
        if (read_done) return hipSuccess;
        hipError_t e = hipEventCreateWithFlags(&read_done, hipEventDisableTiming);
        return e != hipSuccess ? e : hipEventCreateWithFlags(&ready, hipEventDisableTiming);
    }
    void release() {
        if (read_done) (void)hipEventDestroy(read_done);
        if (ready) (void)hipEventDestroy(ready);
        read_done = ready = read_event = nullptr;
        was_read = was_written = false;
    }
};

// how many frames the binning lane may run ahead of the raster lane: that many sets of tile lists per frame and of primitive records
// per scene (S10k@4096^2: 13 MB + 60 MB per set). Measured: 3 is no faster than 2 — the step is bound by the GPU's total throughput, not by
// the coupling of the lanes.
#ifndef CRH_PIPELINE_DEPTH
#define CRH_PIPELINE_DEPTH 2
#endif
constexpr int kPipelineDepth = CRH_PIPELINE_DEPTH;

struct crh_frame {
    crh_renderer* renderer; // nullptr once the renderer has been destroyed (only crh_frame_destroy is valid then)
    int device = 0;
    uint32_t width, height, tiles_x, tiles_y, n_tiles;
    uint32_t format = CRH_FORMAT_RGBA8; // CRH_FORMAT_RGBA16F: a layer of the multi-GPU exchange kept at half precision (8 bytes per pixel)
    size_t pixel_bytes() const { return format == CRH_FORMAT_RGBA16F ? 8u : 4u; }
    size_t image_bytes() const { return (size_t)width * height * pixel_bytes(); }
    DevBuf rgba8;
    // The exchange (csrc/comm.hip) reads a layer and writes the result frame on a stream of its own and does not wait for either on the
    // host: ext_read = its last read of this frame's pixels, ext_write = its last write. Whatever touches the pixels next is ordered
    // behind them on its stream (order_after_external).
    hipEvent_t ext_read = nullptr, ext_write = nullptr;
    bool ext_read_set = false, ext_write_set = false;
    // Two sets of binning buffers, used alternately: frame N + 1 is binned while frame N's raster kernel still reads the other set.
    struct BinSet {
        DevBuf tile_count_cursor, tile_offset, tile_list, scan_scratch; // tile_count_cursor: [tile_cursor | tile_count | overflow words] — one memset clears what a pass needs
        void* overflow_p = nullptr; // (inside tile_count_cursor)
        uint64_t raster_serial = 0; // the render call that recorded raster_done
        DevBuf pair_tile, pair_pos, pair_key;       // the edge pass: (tile, key) pairs as the binning waves produced them (same capacity as tile_list)
        DevBuf bin_queue;                           // the edge pass: items handed from k_bin_flat to k_bin_edges
        DevBuf item_elsewhere;                      // a pass with a slab: one byte per item, 1 = the item misses the slab (k_slab_items)
        hipEvent_t bin_done = nullptr;    // recorded on the binning stream after the fill pass
        hipEvent_t raster_done = nullptr; // recorded on the raster stream after the raster kernel that read this set
        // The pass' flag words (overflow[0 .. kExtraTurnsWord]) copied to pinned host memory on the side stream as soon as its raster kernel is
        // through: a caller that consumes every frame (crh_frame_synchronize before the target is drawn into again — any animation) finds them
        // there instead of paying two device-to-host copies and two stream synchronisations per frame
        uint32_t* flags_host = nullptr;
        hipEvent_t flags_ready = nullptr; // (fires after raster_done)
        bool used = false;
    } sets[kPipelineDepth];
    int next_set = 0, last_set = 0;
    // a recorded pass (crh_scene_render_draws): merged draw items, their instance data and primitive ranges
    DevBuf items, item_transforms, item_colors, item_ncand, item_prim_begin, item_scan_scratch;
    DevBuf item_nslots, item_slot_begin; // the edge pass: slots of the primitive heap per item
    uint32_t items_total_slots = 0;
    DevBuf item_transforms_b, item_colors_b; // second set of instance data: a re-submitted pass writes the set its predecessor did not read
    int item_inst_cur = 0, item_inst_last = 0, item_projective_of[2] = {0, 0}, item_tame_of[2] = {0, 0};
    PinnedUpload item_upload_t, item_upload_c;
    InstanceSlot item_slot[2];
    uint32_t n_items = 0;
    int last_instances = 0;        // which of the scene's two instance buffers the last plain pass read
    // the recorded pass the frame holds: when the next crh_scene_render_draws records the same items over the same geometry, their upload,
    // the per-item primitive ranges and the read-back of the primitive total are skipped (only the instance data is refreshed)
    std::vector<DrawItem> items_host;
    crh_scene* items_scene = nullptr;
    uint64_t items_generation = 0;
    uint32_t items_total = 0;
    bool items_ranges_valid = false;
    bool items_need_ops = false;   // the pass uses more than Stencil / Color at clip depth 0 (otherwise the plain raster kernel serves it)
    DevBuf depth;                  // [height][width][samples] f32, when the configuration tests or writes depth
    // Pass state that outlives a pass (RasterParams::state_*; renderer.rs:148-158, 257-266, 932-985: the reference's stencil attachment and alpha
    // layers are caller-owned textures, so `shape_a.render(Clip)` clips whatever Shapes are rendered afterwards until `shape_a.render(UnClip)`).
    // A recorded pass that ends with state left over (pass_leaves_state) makes the frame `carry`: from then on, until crh_frame_clear (the stencil's
    // LoadOp::Clear, main.rs:217-230), every pass into it is drawn by the OPS kernel, which starts its tiles from these planes and stores them back.
    DevBuf state_stencil, state_alpha, state_color;
    bool carry = false;            // the frame keeps its pass state in HBM
    bool carry_valid = false;      // ... and the planes hold what the earlier passes left (false: to be initialised in front of the next pass)
    bool carry_recolor = false;    // the exchange wrote the frame's pixels behind the planes' back: the sample colours start from the image again
    bool cleared = true;
    bool pairs_known = false;
    // Direct tile lists (RasterParams::direct): once a pass of a Scene into this frame has been verified, the places of its lists are kept
    // — with half as much again and sixteen entries of headroom per tile — and the following passes of the same Scene store their keys
    // straight into them (no pair stream, scan or scatter). A tile that outgrows its place is noticed like any overflow (settle_frame):
    // the pass is drawn again the exact way and the places are taken anew; after three such misses the frame stays on the exact way.
    DevBuf tile_base, tile_caps;
    // Round 5: the places FOLLOW the scene. Every pass with its lists in place leaves, behind its binning kernels and beside its raster kernel, the
    // places of the NEXT pass — the prefix of (the longest count within three tiles + half + 64) — in the other of two buffers: no host involvement, and a list is only
    // outgrown when it grows by half from one pass into this frame to the next (a zoom of 1 % per frame outgrew the places of one verified
    // pass within a dozen frames; after three redraws the frame fell back to the pair stream, 0.05 ms per frame slower).
    DevBuf tile_base_b;
    int base_cur = 0;            // which of tile_base / tile_base_b the next pass reads
    // A frame whose passes see the instances change is `moving`: its places are then sized by the longest list within kMovingListRadius tiles
    // (a camera shifts the content by whole tiles between two passes into one target); a frame of resident instances keeps tight places —
    // lists side by side, measured 1 % faster on the steady figure (0.3125 against 0.3155 ms per step).
    uint32_t slab_ty0 = 0, slab_ty1 = 0xFFFFFFFFu; // crh_frame_set_tile_rows: the tile rows the raster kernels draw (all of them by default)
    bool moving = false;
    uint64_t last_pass_instances = ~0ull;
    uint64_t last_pass_geometry = 0; // crh_scene::generation of the latest plain pass
    uint64_t places_instances = 0; // crh_scene::instances_version of the pass whose counts the current places come from (unchanged instances: unchanged counts, no new places)
    uint32_t direct_clean = 0;   // passes with lists in place since the last one that outgrew them (thirty-two forgive the misses)
    bool direct_ready = false;
    crh_scene* direct_scene = nullptr;
    uint64_t direct_generation = 0;
    uint64_t direct_entries = 0;
    uint32_t direct_misses = 0;
    crh_scene* seen_scene = nullptr; // the Scene (and its geometry) of the frame's latest passes, and how many in a row
    uint64_t seen_generation = 0;
    uint32_t seen_passes = 0;
    bool counts_describe_pixels = false; // the last pass drew into a cleared frame and nothing else has written the pixels since: a tile without entries is transparent (the exchange's occupancy bitmap then comes from the tile counts)
    bool last_edges = false;  // the formulation of the frame's last plain pass: the other one has other tile lists (their sizes are learned again)
    bool last_direct = false; // the pass pending verification was a direct one
    bool queue_seen = true;       // the verified pass handed items from k_bin_flat on to k_bin_edges (until known otherwise: the queue kernel is launched)
    bool last_skipped_queue = false;
    bool last_used_batches = false; // the pass pending verification took k_bin_flat's runs by cost (stale_batches)
    uint32_t sort_capacity = 128;  // primitives per tile the raster kernel sorts in LDS (a power of two, >= 128: the LDS path pads to 128); grown from the longest tile list a verified pass reports, before its raster kernel runs. (1 024 until round 4: at msaa 4 — four wavefronts, 16 KB — that alone kept the workgroups with their colours in LDS at three per CU.)
    uint32_t longest_list = 0;     // the longest tile list any pass into this frame has reported (overflow[3]); gates k_raster_rows
    uint32_t opaque_covers = 0;    // ... and how many (item, tile) covers of it were opaque over the whole tile (without them there is nothing to start late behind)
    uint32_t mean_list = 0;        // entries per tile of the last verified EDGE pass (the pairs the pass needed / tiles): long lists get k_raster_edges' LONG variant
    size_t pair_capacity_bytes = 1024 * 4; // size of a set's tile list (both sets grow to it)
    // last render, for the transparent re-run after a bin-capacity overflow
    crh_scene* last_scene = nullptr;
    bool check_pending = false;
    bool last_tess_optimistic = false; // the pass pending verification drew an optimistic tessellation: its flag words hold that run's status (kTessStatusWord)
    // The edge pass met a boundary edge with a non-finite endpoint on this frame (finite vertices times a finite matrix can overflow): an
    // unclosed chain has no backdrops, so passes of that Scene into this frame are drawn by the triangle pass, which skips exactly the
    // strip triangles with a non-finite determinant — as the reference's rasterizer does.
    DevBuf tile_order;             // [workgroups of the edge pass' raster grid] the tile each one draws (order_tiles_heavy_first), or not ready: the kernels' own order
    bool tile_order_ready = false;
    uint32_t tile_order_places = 0; // the leading places of the order that hold tiles (a frame with a slab: 1 / world of them)
    // k_bin_flat's batches by cost (RasterParams::item_cost / bin_batches): built at a verified edge pass, used by the later passes of the same
    // geometry with the same number of items (any partition of the items bins the same lists: stale costs only cost time)
    DevBuf item_cost, bin_batches;
    uint32_t n_bin_batches = 0, batches_items = 0, batches_age = 0; // (age: passes drawn with these runs)
    crh_scene* batches_scene = nullptr;
    uint64_t batches_generation = 0;
    crh_scene* triangle_pass_for = nullptr;
    uint64_t triangle_pass_generation = 0; // ... of that Scene's geometry: a re-upload (or a new Scene at the same address) starts on the edge pass again
};

constexpr int kTessBufs = 33; // buffers a tessellation run writes (crh_scene::tess_bufs)
struct crh_scene {
    crh_renderer* renderer; // nullptr once the renderer has been destroyed (only crh_scene_destroy is valid then)
    int device = 0;
    SceneDev d;
    uint32_t n_segments = 0;
    bool has_stroke = false, big_shapes = false;
    uint64_t structure_hash = 0; // of shape_path_begin and path_segment_begin of the uploaded batch (crh_scene_upload: same_structure)
    bool capacity_known = false;
    uint64_t instances_version = 0; // counts crh_scene_set_instances calls
    uint64_t input_bytes = 0, emitted_bytes = 0;
    uint32_t totals_host[NCH] = {};
    uint32_t cap_host[NCH] = {};   // records every stream of the current set holds (ensure_outputs: the totals + a sixteenth + 64) — what d.capacity says
    // New paths of the SAME STRUCTURE (as many Shapes, paths, elements: an animation of control points) uploaded into a Scene whose streams
    // are sized: the capacities are kept and the tessellation runs without the host waiting for its totals (`optimistic`; until something
    // synchronises, totals_host holds the capacities — upper bounds). A run that does not fit raises the overflow code and emits nothing;
    // the frame drawn from it finds the code among its flag words (kTessStatusWord) when it is settled, the host sizes the streams from
    // the totals and the frame is drawn again — the way an outgrown tile list is handled.
    bool optimistic = false;
    bool counts_valid = false;     // elem_cnt / run_base belong to the uploaded paths (the one-pass path counts once per upload)
    uint64_t lineage = 0;          // the generation of the first upload of this structure: what a frame's list places and batch runs are kept against
    // How many Shapes the hull kernels beyond k_hull_small find queued (a property of the geometry): fetched once behind the first
    // tessellation of these paths, asynchronously; later runs do not launch a kernel whose queue is empty (50 000 glyphs: two launches per
    // frame that executed no vector instruction and waited 51 us each for wave slots on the tessellation lane)
    uint32_t* hull_queued_host = nullptr; // pinned, [4]
    hipEvent_t hull_queued_ready = nullptr;
    int hull_queued_state = 0; // 0 unknown, 1 the copy is on its way, 2 known
    std::vector<uint32_t> shape_dyn_begin_host;
    // inputs
    DevBuf shape_bounds;             // [n_shapes][4] every Shape's box in its own coordinates (k_shape_bounds): passes with a slab (the tile split) leave out the items that miss it
    uint64_t bounds_generation = 0;  // ... of which upload (the same paths tessellate to the same hulls: once per upload, by the first pass with a slab)
    DevBuf geometry; // the uploaded element stream: ONE arena (the layout of crh_scene_upload's pinned staging arena), one copy
    DevBuf elem_cnt, run_base; // the one-pass tessellation: runs of Shapes, every element's packed counts, the runs' bases (scene.hpp) — per upload
    // scan state
    DevBuf elem_scan, wg_total, wg_base, group_base, totals, shape_base, hull_count, hull_large, hull_sort, hull_chain, status, path_scan;
    // outputs
    DevBuf line_v, joint_v, solid_v, iq_v, ic_v, rq_v, rc_v, hull_cand, hull_v, line_i, joint_i, solid_i, solid_flag, line_pair_cut, line_pair_mode, line_inc;
    // instances + binning
    DevBuf transforms, colors, shape_ncand, shape_prim_begin, prim_scan_scratch;
    DevBuf shape_nslots, shape_slot_begin; // the edge pass: slots of the primitive heap per Shape (transform independent, like shape_prim_begin)
    DevBuf prim_rec[kPipelineDepth];    // set-up triangles, one buffer per frame in flight like the frame's binning buffers
    DevBuf prim_proj[kPipelineDepth];                // 1/w and z/w planes of the primitives of projective instances (allocated on first use)
    bool instances_projective = false;  // of the current instance buffer
    // Instance data is double-buffered: crh_scene_set_instances writes the buffer the latest frame did NOT use, so that frame — whose
    // deferred tile-list check may still ask for it to be drawn again — needs no wait; only a frame two updates old is settled first.
    DevBuf transforms_b, colors_b;
    PinnedUpload upload_t, upload_c;
    PinnedArena geometry_stage;          // crh_scene_upload's element stream on its way to the device
    hipEvent_t geometry_ready = nullptr; // behind those copies (on the renderer's stream); the next tessellation waits for it once
    bool geometry_pending = false, tessellated_once_before_upload = false;
    UploadBuild pending_build = {}; // what the first tessellation behind an upload builds the element stream from (launch_build_elements)
    InstanceSlot slot[2];
    int instances_cur = 0;
    uint64_t generation = 0;            // bumped by every upload: what a frame's cached recorded pass was built against
    bool instances_projective_of[2] = {false, false};
    bool instances_tame_of[2] = {false, false}; // all_colors_tame of that instance buffer
    hipEvent_t rec_raster_done[kPipelineDepth] = {};
    bool last_render_one_event = false; // the last render that READ the current tessellation set recorded ONE event (vertices_free) for everything behind its binning:
                                        // the next tessellation into this set need not wait for ranges_free as well. Per set: flip_tess_set swaps it with the shadow's
    uint64_t rec_raster_serial[kPipelineDepth] = {}; // the render call that recorded it (crh_renderer::render_serial)
    bool rec_used[kPipelineDepth] = {};
    bool rec_used_ever[kPipelineDepth] = {}; // rec_raster_done[k] has been recorded (crh_scene_upload orders its copy behind them: rec_used is reset by every upload)
    int next_rec = 0;
    bool instances_set = false;
    // frame pipelining: tessellation runs on its own stream; these events order it against the raster stream
    hipEvent_t tess_done = nullptr;     // recorded on the tessellation stream after the last tessellation kernel
    hipEvent_t vertices_free = nullptr; // recorded on the raster stream after k_prim_setup (last reader of the vertex streams and hulls)
    hipEvent_t ranges_free = nullptr;   // recorded on the raster stream after the fill pass (last reader of the primitive ranges)
    bool rendered_once = false;
    // Everything a tessellation run writes exists twice once a Scene is tessellated again (animated paths, the benchmark step): the run
    // for frame i + 1 fills the set that frame i does NOT read, so it overlaps frame i's binning instead of waiting for it (with one set
    // the chain tessellate -> bin of consecutive frames was serial and, not the raster kernel, set the frame rate). The members above are
    // the current set; flip_tess_set() swaps them with this one and re-binds the pointers of `d`.
    struct TessShadow {
        DevBuf buf[kTessBufs];
        hipEvent_t tess_done = nullptr, vertices_free = nullptr, ranges_free = nullptr;
        bool allocated = false, capacity_known = false, rendered_once = false, last_render_one_event = false;
        uint32_t totals_host[NCH] = {}, cap_host[NCH] = {};
        uint64_t emitted_bytes = 0;
    } shadow;
    bool tessellated_once = false;
    // Which formulation draws this Scene's plain passes — boundary edges + backdrops (raster_edges.hip) or the reference's strip triangles
    // (raster.hip) — is decided by MEASUREMENT: both give the same pixels, and which is faster depends on the content (long strips
    // across many tiles favour the edges, tens of thousands of glyph-sized Shapes the triangles: the edge pass pays per item and per
    // (item, tile)). Frames 0-5 after an upload run the edge pass, frames 6-11 the triangle pass. The first frame of each is the verified
    // one (it may run twice), the first three size the buffers of both pipeline sets (allocations: the edge pass' lists move once more
    // when they are put in place); the three behind them run as every later frame will — pipelined as far as the caller lets them —
    // and are timed as a GROUP: from the end of the raster kernel before them to the end of their last one (two events per pass on the
    // raster stream). The thirteenth frame keeps the faster pass. (Rounds 2 - 3 timed single
    // synchronised frames kernel by kernel: on glyph scenes, where the passes are within a few percent of each other alone, that put the
    // Scene on the pass that is 14 % slower in the pipelined run.) A pass that had to be drawn again (an overflow) starts the trial over.
    // CRH_EDGE_PASS=1 / CRH_TRIANGLE_PASS=1 pin the choice.
    struct PassTrial {
        hipEvent_t e[2] = {}; // on the raster stream: before the first timed frame is enqueued, behind the raster kernel of the last one
        bool started = false, recorded = false;
    } pass_trial[3];
    int pass_choice = 0;       // 0 undecided, 1 edges (per-sample raster kernel), 2 triangles, 3 edges with the row-span raster kernel: for targets of the size class pass_class
    uint32_t pass_frames = 0;  // plain frames of the trial under way
    uint32_t pass_class = 0;   // size class of the target the trial / choice belongs to (log4 of its area)
    uint8_t pass_known[16] = {}; // choices already measured, by size class: a Scene drawn into a large frame and a thumbnail in turn measures twice, not for ever
    uint32_t pass_geometry = 0; // what the choices were measured on: (log2 of the Shape count, log2 of the segment count) — an upload of geometry of the same class keeps them
    float pass_ms[3] = {0.0f, 0.0f, 0.0f};
    void tess_bufs(DevBuf* (&out)[kTessBufs]) {
        DevBuf* all[kTessBufs] = {&elem_scan, &wg_total, &wg_base, &group_base, &totals, &shape_base, &hull_count, &hull_large, &hull_sort, &hull_chain, &status, &line_v, &joint_v,
                                  &solid_v, &iq_v, &ic_v, &rq_v, &rc_v, &hull_cand, &hull_v, &line_i, &joint_i, &solid_i, &solid_flag, &line_pair_cut, &line_pair_mode,
                                  &line_inc, &shape_ncand, &shape_prim_begin, &prim_scan_scratch, &shape_nslots, &shape_slot_begin, &path_scan};
        for (int i = 0; i < kTessBufs; ++i) out[i] = all[i];
    }
    // host copies for the parity taps
    std::vector<uint32_t> shape_base_host, hull_count_host;
    bool layout_valid = false;

    void release_all() {
        DevBuf* all[] = {&geometry, &shape_bounds, &elem_cnt, &run_base, &path_scan,
                         &elem_scan, &wg_total, &wg_base, &group_base, &totals, &shape_base, &hull_count, &hull_large, &hull_sort, &hull_chain, &status, &line_v, &joint_v,
                         &solid_v, &iq_v, &ic_v, &rq_v, &rc_v, &hull_cand, &hull_v, &line_i, &joint_i, &solid_i, &solid_flag, &line_pair_cut,
                         &line_pair_mode, &line_inc, &transforms, &colors, &transforms_b, &colors_b, &shape_ncand, &shape_prim_begin, &prim_scan_scratch, &shape_nslots, &shape_slot_begin};
        for (DevBuf* b : all) b->release();
        for (DevBuf& b : shadow.buf) b.release();
        shadow.allocated = false;
        for (DevBuf& b : prim_rec) b.release();
        for (DevBuf& b : prim_proj) b.release();
        upload_t.release();
        upload_c.release();
        geometry_stage.release();
        for (InstanceSlot& k : slot) k.release();
        if (hull_queued_host) (void)hipHostFree(hull_queued_host), hull_queued_host = nullptr;
        if (hull_queued_ready) (void)hipEventDestroy(hull_queued_ready), hull_queued_ready = nullptr;
        hull_queued_state = 0;
    }
};

namespace {
const int kSegmentFloats[5] = {2, 4, 6, 5, 10};

// renderer.rs:29-60
crh_status convert_options(const crh_dynamic_stroke_options& o, crh_dynamic_stroke_descriptor& out) {
    std::memset(&out, 0, sizeof(out));
    // the Rust enums make these states unrepresentable; out-of-range values would spill into the neighbouring nibbles of `caps`
    if (o.join > CRH_JOIN_ROUND || o.dashed > 1u) return CRH_ERR_INVALID_ARGUMENT;
    if (o.dashed) {
        if (o.pattern_len > CRH_MAX_DASH_INTERVALS) return CRH_ERR_TOO_MANY_DASH_INTERVALS;
        if (o.pattern_len == 0) return CRH_ERR_INVALID_ARGUMENT;
        out.count_dashed_join = ((o.pattern_len - 1u) << 3) | 4u | o.join;
        out.phase = o.phase;
        for (uint32_t i = 0; i < o.pattern_len; ++i) {
            if (o.pattern[i].dash_start > CRH_CAP_BUTT || o.pattern[i].dash_end > CRH_CAP_BUTT) return CRH_ERR_INVALID_ARGUMENT;
            if (!std::isfinite(o.pattern[i].gap_start) || !std::isfinite(o.pattern[i].gap_end)) return CRH_ERR_NON_FINITE;
            out.gap_start[i] = o.pattern[i].gap_start;
            out.gap_end[i] = o.pattern[i].gap_end;
            out.caps |= o.pattern[i].dash_start << (((i + o.pattern_len - 1u) % o.pattern_len) * 8u);
            out.caps |= o.pattern[i].dash_end << (i * 8u + 4u);
        }
    } else {
        if (o.start > CRH_CAP_BUTT || o.end > CRH_CAP_BUTT) return CRH_ERR_INVALID_ARGUMENT;
        out.caps = o.start | (o.end << 4);
        out.count_dashed_join = o.join;
        out.phase = 0.0f;
    }
    return CRH_OK;
}

template <typename T>
crh_status upload_vector(DevBuf& buf, const std::vector<T>& v, hipStream_t stream) {
    HIP_TRY(buf.ensure(v.size() * sizeof(T)));
    if (!v.empty()) HIP_TRY(hipMemcpyAsync(buf.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, stream));
    return CRH_OK;
}

crh_status decode_status(crh_scene* scene, uint32_t word) {
    (void)scene;
    if (word == 0xFFFFFFFFu) return CRH_OK;
    const uint32_t code = word & 0xFFu;
    if (code >= 0x80u) return CRH_ERR_UNSUPPORTED;
    return static_cast<crh_status>(code);
}

// the pointers of sc->d into the (current) set of tessellation buffers
void bind_tess_pointers(crh_scene* sc) {
    SceneDev& d = sc->d;
    d.elem_scan = sc->elem_scan.as<ElemScan>();
    d.wg_total = sc->wg_total.as<uint32_t>();
    d.wg_base = sc->wg_base.as<uint32_t>();
    d.group_base = sc->group_base.as<uint32_t>();
    d.totals = sc->totals.as<uint32_t>();
    d.shape_base = sc->shape_base.as<uint32_t>();
    d.path_scan = sc->path_scan.as<uint32_t>();
    d.hull_count = sc->hull_count.as<uint32_t>();
    d.hull_large_count = sc->hull_large.as<uint32_t>();
    d.hull_large_list = sc->hull_large.as<uint32_t>() + 4;
    d.status = sc->status.as<uint32_t>();
    d.line_v = sc->line_v.as<Vertex2f1i>();
    d.joint_v = sc->joint_v.as<Vertex3f1i>();
    d.solid_v = sc->solid_v.as<Vertex0>();
    d.iq_v = sc->iq_v.as<Vertex2f>();
    d.ic_v = sc->ic_v.as<Vertex3f>();
    d.rq_v = sc->rq_v.as<Vertex3f>();
    d.rc_v = sc->rc_v.as<Vertex4f>();
    d.hull_cand = sc->hull_cand.as<Vertex0>();
    d.hull_v = sc->hull_v.as<Vertex0>();
    d.hull_sort = sc->hull_sort.as<float2>();
    d.hull_chain = sc->hull_chain.as<float2>();
    d.line_i = sc->line_i.as<uint16_t>();
    d.joint_i = sc->joint_i.as<uint16_t>();
    d.solid_i = sc->solid_i.as<uint16_t>();
    d.solid_flag = sc->solid_flag.as<uint8_t>();
    d.line_pair_cut = sc->line_pair_cut.as<uint8_t>();
    d.line_pair_mode = sc->line_pair_mode.as<uint8_t>();
    d.line_inc = sc->line_inc.as<float>();
}

// Before a repeated tessellation run: continue on the other set of buffers (allocated like the current one the first time).
crh_status flip_tess_set(crh_scene* sc) {
    crh_scene::TessShadow& o = sc->shadow;
    DevBuf* cur[kTessBufs];
    sc->tess_bufs(cur);
    if (!o.allocated) {
        for (int i = 0; i < kTessBufs; ++i)
            if (cur[i]->cap) HIP_TRY(o.buf[i].ensure(cur[i]->cap));
        for (hipEvent_t* e : {&o.tess_done, &o.vertices_free, &o.ranges_free})
            if (!*e) HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
        // the same input gives the same totals: the sizes of the current set are right for this one (an overflow is caught as ever)
        o.capacity_known = sc->capacity_known;
        std::memcpy(o.totals_host, sc->totals_host, sizeof(o.totals_host));
        std::memcpy(o.cap_host, sc->cap_host, sizeof(o.cap_host));
        // (rows nobody has written yet must not send a pass anywhere: an optimistic run that does not fit leaves them as they are)
        if (o.buf[5].p) HIP_TRY(hipMemsetAsync(o.buf[5].p, 0, o.buf[5].cap, sc->renderer->tessellation_stream()));
        if (o.buf[6].p) HIP_TRY(hipMemsetAsync(o.buf[6].p, 0, o.buf[6].cap, sc->renderer->tessellation_stream()));
        o.emitted_bytes = sc->emitted_bytes;
        o.rendered_once = false;
        o.allocated = true;
    }
    for (int i = 0; i < kTessBufs; ++i) std::swap(*cur[i], o.buf[i]);
    std::swap(sc->tess_done, o.tess_done);
    std::swap(sc->vertices_free, o.vertices_free);
    std::swap(sc->ranges_free, o.ranges_free);
    std::swap(sc->capacity_known, o.capacity_known);
    std::swap(sc->rendered_once, o.rendered_once);
    std::swap(sc->last_render_one_event, o.last_render_one_event);
    std::swap(sc->emitted_bytes, o.emitted_bytes);
    for (int c = 0; c < NCH; ++c) std::swap(sc->totals_host[c], o.totals_host[c]), std::swap(sc->cap_host[c], o.cap_host[c]);
    for (int c = 0; c < NCH; ++c) sc->d.capacity[c] = sc->capacity_known ? sc->cap_host[c] : 0u;
    bind_tess_pointers(sc);
    return CRH_OK;
}

crh_status ensure_outputs(crh_scene* sc) {
    // (a sixteenth + 64 records beyond the totals: new paths of the same structure — crh_scene_upload into this Scene — then fit without a wait)
    uint32_t want[NCH];
    for (int c = 0; c < NCH; ++c) want[c] = (uint32_t)std::min<uint64_t>(0xFFFFFFF0ull, (uint64_t)sc->totals_host[c] + sc->totals_host[c] / 16u + 64u);
    want[CH_LINE_V] &= ~1u; // (pairs)
    const uint32_t* t = want;
    SceneDev& d = sc->d;
    const size_t pairs = t[CH_LINE_V] / 2 + 1;
    HIP_TRY(sc->line_v.ensure((size_t)t[CH_LINE_V] * 20));
    HIP_TRY(sc->line_i.ensure(((size_t)t[CH_LINE_V] + t[CH_LINE_CUT]) * 2));
    HIP_TRY(sc->line_inc.ensure(pairs * 4));
    HIP_TRY(sc->line_pair_cut.ensure(pairs));
    HIP_TRY(sc->line_pair_mode.ensure(pairs));
    HIP_TRY(sc->joint_v.ensure((size_t)t[CH_JOINT] * 5 * 24));
    HIP_TRY(sc->joint_i.ensure((size_t)t[CH_JOINT] * 6 * 2));
    HIP_TRY(sc->solid_v.ensure((size_t)t[CH_SOLID_V] * 8));
    HIP_TRY(sc->solid_i.ensure(((size_t)t[CH_SOLID_V] + t[CH_SOLID_END]) * 2));
    HIP_TRY(sc->solid_flag.ensure((size_t)t[CH_SOLID_V] + 2));
    HIP_TRY(sc->iq_v.ensure((size_t)t[CH_IQ] * 3 * 16));
    HIP_TRY(sc->ic_v.ensure((size_t)t[CH_IC_V] * 20));
    HIP_TRY(sc->rq_v.ensure((size_t)t[CH_RQ] * 3 * 20));
    HIP_TRY(sc->rc_v.ensure((size_t)t[CH_RC_V] * 24));
    HIP_TRY(sc->hull_cand.ensure((size_t)t[CH_HULL] * 8));
    HIP_TRY(sc->hull_v.ensure((size_t)t[CH_HULL] * 8));
    if (sc->big_shapes || sc->has_stroke) { // only Shapes beyond the LDS hull kernels (> 2048 candidates) use these — a stroked Shape reaches that with few
                                            // segments (every emitted line vertex is a hull candidate, stroke.rs:125); sized so that any Shape may
        HIP_TRY(sc->hull_sort.ensure((size_t)t[CH_HULL] * 16 + 16));
        HIP_TRY(sc->hull_chain.ensure((size_t)t[CH_HULL] * 16 + 16));
    }
    for (int c = 0; c < NCH; ++c) d.capacity[c] = sc->cap_host[c] = t[c]; // (what THIS sizing guarantees; the buffers themselves only grow)
    bind_tess_pointers(sc);
    t = sc->totals_host;
    sc->emitted_bytes = (uint64_t)t[CH_LINE_V] * 20 + ((uint64_t)t[CH_LINE_V] + t[CH_LINE_CUT]) * 2 + (uint64_t)t[CH_JOINT] * (5 * 24 + 6 * 2) +
                        (uint64_t)t[CH_SOLID_V] * 8 + ((uint64_t)t[CH_SOLID_V] + t[CH_SOLID_END]) * 2 + (uint64_t)t[CH_IQ] * 48 + (uint64_t)t[CH_IC_V] * 20 +
                        (uint64_t)t[CH_RQ] * 60 + (uint64_t)t[CH_RC_V] * 24;
    return CRH_OK;
}

crh_status run_tessellation(crh_scene* sc, bool again) {
    crh_renderer* r = sc->renderer;
    HIP_TRY(hipSetDevice(r->device));
    if (sc->tessellated_once && !again) { // not the re-run after a capacity overflow: that one repeats on the set it overflowed
        const crh_status st = flip_tess_set(sc);
        if (st != CRH_OK) return st;
    }
    sc->tessellated_once = true;
    if (again) sc->bounds_generation = 0; // (boxes taken from the hulls of a run that did not fit are nobody's)
    if (again) sc->hull_queued_state = 0; // (the run that overflowed queued nothing: its hull kernels returned at once; the caller has synchronised)
    SceneDev& d = sc->d;
    const hipStream_t ts = r->tessellation_stream();
    // the previous frame's k_prim_setup must have consumed the vertex streams this run overwrites; its binning and raster may still run
    if (sc->rendered_once) HIP_TRY(hipStreamWaitEvent(ts, sc->vertices_free, 0));
    if (sc->geometry_pending) { // the batch of the last crh_scene_upload is still on its way (an asynchronous copy out of pinned staging memory)
        HIP_TRY(hipStreamWaitEvent(ts, sc->geometry_ready, 0));
        // ... and its element stream is built from it HERE, in front of the kernels that read it, on their stream. (On the upload stream the kernel had a
        // compute queue of its own to wait in: at normal priority new paths every frame took 0.8 ms per step instead of 0.44, at high priority the mere
        // existence of the stream slowed every OTHER renderer of the process down by half — profiles/r06_experiments.txt.)
        launch_build_elements(sc->pending_build, ts);
        sc->geometry_pending = false;
    }
    if (r->raster_exclusive && r->raster_events[1]) HIP_TRY(hipStreamWaitEvent(ts, r->raster_events[1], 0));
    r->begin_marks(1);
    HIP_TRY(hipMemsetAsync(d.status, 0xFF, 4, ts));
    if (d.n_elems == 0) { // nothing to tessellate: every offset is zero
        HIP_TRY(hipMemsetAsync(d.totals, 0, NCH * 4, ts));
        if (d.n_shapes) HIP_TRY(hipMemsetAsync(d.shape_base, 0, (size_t)d.n_shapes * kShapeRow * 4, ts));
        if (d.n_shapes) HIP_TRY(hipMemsetAsync(d.hull_count, 0, (size_t)d.n_shapes * 4, ts));
    }
    const uint64_t bytes[4] = {sc->input_bytes, 0, sc->input_bytes + sc->emitted_bytes, 0};
    // (the one-pass path counts once per upload — elem_cnt, run_base —, the two-pass path every time)
    if (!sc->capacity_known)
        for (int c = 0; c < NCH; ++c) d.capacity[c] = 0u; // (not known: the counting kernels must not take the streams' old sizes for a verdict — k_shape_rows)
    // CRH_TESS_COUNT_EVERY_RUN (read per run: bench.py's `recount` side block switches it inside one process): the counts and bases are not kept —
    // k_tess_count_runs and k_scan_runs run in front of every k_tess_runs, as they do for new paths
    launch_tessellate(d, ts, r->mark_fn_tess(), r, bytes, sc->has_stroke, !sc->capacity_known || !sc->counts_valid || getenv("CRH_TESS_COUNT_EVERY_RUN") != nullptr);
    sc->counts_valid = true;
    if (!sc->capacity_known) { // first run: the output sizes are data dependent, fetch the totals once and allocate
        HIP_TRY(hipMemcpyAsync(sc->totals_host, d.totals, sizeof(uint32_t) * NCH, hipMemcpyDeviceToHost, ts));
        // (the tessellation stream alone: it has waited for the consumers of this set's streams — vertices_free, above — before the count kernel, and
        // the frames before this one go on binning and rasterizing on their streams while the host waits here. Until round 4 this was a wait for
        // every stream: new paths every frame, bench.py --reupload, ran one frame at a time)
        HIP_TRY(hipStreamSynchronize(ts));
        crh_status st = ensure_outputs(sc);
        if (st != CRH_OK) return st;
        sc->capacity_known = true;
        sc->optimistic = false;
    }
    if (sc->has_stroke) HIP_TRY(hipMemsetAsync(d.line_pair_cut, 0, sc->line_pair_cut.cap, ts));
    const uint64_t bytes2[4] = {sc->input_bytes, 0, sc->input_bytes + sc->emitted_bytes, (uint64_t)sc->totals_host[CH_HULL] * 8};
    if (sc->hull_queued_state == 1 && hipEventQuery(sc->hull_queued_ready) == hipSuccess) sc->hull_queued_state = 2;
    launch_emit(d, ts, r->mark_fn_tess(), r, bytes2, sc->has_stroke, sc->big_shapes, sc->hull_queued_state == 2 ? sc->hull_queued_host : nullptr);
    if (sc->hull_queued_state == 0 && (sc->has_stroke || sc->big_shapes)) {
        if (!sc->hull_queued_host) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&sc->hull_queued_host), 16, hipHostMallocDefault));
        if (!sc->hull_queued_ready) HIP_TRY(hipEventCreateWithFlags(&sc->hull_queued_ready, hipEventDisableTiming));
        HIP_TRY(hipMemcpyAsync(sc->hull_queued_host, d.hull_large_count, 16, hipMemcpyDeviceToHost, ts));
        HIP_TRY(hipEventRecord(sc->hull_queued_ready, ts));
        sc->hull_queued_state = 1;
    }
    // contiguous primitive ids per Shape, in draw order (transform independent, so it belongs to the tessellation); the previous
    // frame's tile walks read the old ranges until its fill pass is through
    if (sc->rendered_once && !sc->last_render_one_event) HIP_TRY(hipStreamWaitEvent(ts, sc->ranges_free, 0)); // (one event stood for both: waited for above)
    launch_plain_ranges(d, sc->shape_ncand.as<uint32_t>(), sc->shape_prim_begin.as<uint32_t>(), sc->shape_nslots.as<uint32_t>(), sc->shape_slot_begin.as<uint32_t>(),
                        sc->prim_scan_scratch.as<uint32_t>(), sc->prim_scan_scratch.as<uint32_t>() + ((size_t)(d.n_shapes + 511) / 512 + 2), ts);
    if (r->timing & 2u) crh_renderer::mark_cb_tess(r, "tess_prim_ranges", 0);
    HIP_TRY(hipEventRecord(sc->tess_done, ts));
    HIP_TRY(hipGetLastError());
    sc->layout_valid = false;
    return CRH_OK;
}

// after a sync: did the optimistic (no read-back) run overflow its buffers? then size them from the fresh totals and re-run.
crh_status settle_tessellation(crh_scene* sc, uint32_t* status_word) {
    crh_renderer* r = sc->renderer;
    for (int attempt = 0; attempt < 2; ++attempt) {
        uint32_t word = 0;
        // on the tessellation stream: ordered after the kernels that raise the status (the raster stream is not)
        HIP_TRY(hipMemcpyAsync(&word, sc->d.status, 4, hipMemcpyDeviceToHost, r->tessellation_stream()));
        HIP_TRY(r->sync());
        if (word != 0xFFFFFFFFu && (word & 0xFFu) >= 0x80u && attempt == 0) {
            sc->capacity_known = false, sc->optimistic = false;
            crh_status st = run_tessellation(sc, true);
            if (st != CRH_OK) return st;
            continue;
        }
        if (sc->optimistic && sc->capacity_known) { // the run fitted: the host may know its totals now (until here totals_host held the capacities)
            HIP_TRY(hipMemcpy(sc->totals_host, sc->d.totals, sizeof(uint32_t) * NCH, hipMemcpyDeviceToHost));
            sc->optimistic = false;
        }
        *status_word = word;
        return CRH_OK;
    }
    return CRH_ERR_UNSUPPORTED;
}

crh_status fetch_layout(crh_scene* sc) {
    if (sc->layout_valid) return CRH_OK;
    crh_renderer* r = sc->renderer;
    uint32_t word;
    crh_status st = settle_tessellation(sc, &word);
    if (st != CRH_OK) return st;
    sc->shape_base_host.resize((size_t)sc->d.n_shapes * kShapeRow);
    sc->hull_count_host.resize(sc->d.n_shapes);
    if (sc->d.n_shapes) HIP_TRY(hipMemcpyAsync(sc->shape_base_host.data(), sc->d.shape_base, sc->shape_base_host.size() * 4, hipMemcpyDeviceToHost, r->stream));
    if (sc->d.n_shapes == 0) sc->shape_base_host.clear();
    if (sc->d.n_shapes) HIP_TRY(hipMemcpyAsync(sc->hull_count_host.data(), sc->d.hull_count, sc->hull_count_host.size() * 4, hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(hipMemcpyAsync(sc->totals_host, sc->d.totals, sizeof(uint32_t) * NCH, hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(r->sync());
    sc->layout_valid = true;
    return CRH_OK;
}

void shape_layout(const crh_scene* sc, uint32_t s, uint64_t vo[8], uint64_t io[3]) {
    const uint32_t* a = &sc->shape_base_host[(size_t)s * kShapeRow]; // begin[NCH], end[NCH]
    const uint32_t* b = a + NCH;
    auto n = [&](int c) { return (uint64_t)(b[c] - a[c]); };
    uint64_t v = 0;
    v += n(CH_LINE_V) * 20;
    vo[0] = v;
    v += n(CH_JOINT) * 5 * 24;
    vo[1] = v;
    v += n(CH_SOLID_V) * 8;
    vo[2] = v;
    v += n(CH_IQ) * 3 * 16;
    vo[3] = v;
    v += n(CH_IC_V) * 20;
    vo[4] = v;
    v += n(CH_RQ) * 3 * 20;
    vo[5] = v;
    v += n(CH_RC_V) * 24;
    vo[6] = v;
    v += (uint64_t)sc->hull_count_host[s] * 8;
    vo[7] = v;
    uint64_t i = 0;
    i += (n(CH_LINE_V) + n(CH_LINE_CUT)) * 2;
    io[0] = i;
    i += n(CH_JOINT) * 6 * 2;
    io[1] = i;
    i += (n(CH_SOLID_V) + n(CH_SOLID_END)) * 2;
    io[2] = i;
}

struct HostCopy {
    std::vector<uint8_t> line_v, joint_v, solid_v, iq_v, ic_v, rq_v, rc_v, hull_v, line_i, joint_i, solid_i;
};
crh_status fetch_outputs(crh_scene* sc, HostCopy& h) {
    const uint32_t* t = sc->totals_host;
    hipStream_t st = sc->renderer->stream;
    auto get = [&](std::vector<uint8_t>& dst, const void* src, size_t bytes) -> crh_status {
        dst.resize(bytes);
        if (bytes) HIP_TRY(hipMemcpyAsync(dst.data(), src, bytes, hipMemcpyDeviceToHost, st));
        return CRH_OK;
    };
    crh_status rc;
    if ((rc = get(h.line_v, sc->d.line_v, (size_t)t[CH_LINE_V] * 20)) != CRH_OK) return rc;
    if ((rc = get(h.joint_v, sc->d.joint_v, (size_t)t[CH_JOINT] * 120)) != CRH_OK) return rc;
    if ((rc = get(h.solid_v, sc->d.solid_v, (size_t)t[CH_SOLID_V] * 8)) != CRH_OK) return rc;
    if ((rc = get(h.iq_v, sc->d.iq_v, (size_t)t[CH_IQ] * 48)) != CRH_OK) return rc;
    if ((rc = get(h.ic_v, sc->d.ic_v, (size_t)t[CH_IC_V] * 20)) != CRH_OK) return rc;
    if ((rc = get(h.rq_v, sc->d.rq_v, (size_t)t[CH_RQ] * 60)) != CRH_OK) return rc;
    if ((rc = get(h.rc_v, sc->d.rc_v, (size_t)t[CH_RC_V] * 24)) != CRH_OK) return rc;
    if ((rc = get(h.hull_v, sc->d.hull_v, (size_t)t[CH_HULL] * 8)) != CRH_OK) return rc;
    if ((rc = get(h.line_i, sc->d.line_i, ((size_t)t[CH_LINE_V] + t[CH_LINE_CUT]) * 2)) != CRH_OK) return rc;
    if ((rc = get(h.joint_i, sc->d.joint_i, (size_t)t[CH_JOINT] * 12)) != CRH_OK) return rc;
    if ((rc = get(h.solid_i, sc->d.solid_i, ((size_t)t[CH_SOLID_V] + t[CH_SOLID_END]) * 2)) != CRH_OK) return rc;
    HIP_TRY(hipStreamSynchronize(st));
    return CRH_OK;
}

// the byte image of renderer.rs:198-209 for shape s, assembled from the scene-wide streams
void assemble_shape(const crh_scene* sc, const HostCopy& h, uint32_t s, uint8_t* vb, uint8_t* ib) {
    const uint32_t* a = &sc->shape_base_host[(size_t)s * kShapeRow]; // begin[NCH], end[NCH]
    const uint32_t* b = a + NCH;
    auto put = [](uint8_t*& dst, const std::vector<uint8_t>& src, size_t begin, size_t bytes) {
        if (dst && bytes) {
            std::memcpy(dst, src.data() + begin, bytes);
            dst += bytes;
        }
    };
    put(vb, h.line_v, (size_t)a[CH_LINE_V] * 20, (size_t)(b[CH_LINE_V] - a[CH_LINE_V]) * 20);
    put(vb, h.joint_v, (size_t)a[CH_JOINT] * 120, (size_t)(b[CH_JOINT] - a[CH_JOINT]) * 120);
    put(vb, h.solid_v, (size_t)a[CH_SOLID_V] * 8, (size_t)(b[CH_SOLID_V] - a[CH_SOLID_V]) * 8);
    put(vb, h.iq_v, (size_t)a[CH_IQ] * 48, (size_t)(b[CH_IQ] - a[CH_IQ]) * 48);
    put(vb, h.ic_v, (size_t)a[CH_IC_V] * 20, (size_t)(b[CH_IC_V] - a[CH_IC_V]) * 20);
    put(vb, h.rq_v, (size_t)a[CH_RQ] * 60, (size_t)(b[CH_RQ] - a[CH_RQ]) * 60);
    put(vb, h.rc_v, (size_t)a[CH_RC_V] * 24, (size_t)(b[CH_RC_V] - a[CH_RC_V]) * 24);
    put(vb, h.hull_v, (size_t)a[CH_HULL] * 8, (size_t)sc->hull_count_host[s] * 8);
    put(ib, h.line_i, ((size_t)a[CH_LINE_V] + a[CH_LINE_CUT]) * 2, ((size_t)(b[CH_LINE_V] - a[CH_LINE_V]) + (b[CH_LINE_CUT] - a[CH_LINE_CUT])) * 2);
    put(ib, h.joint_i, (size_t)a[CH_JOINT] * 12, (size_t)(b[CH_JOINT] - a[CH_JOINT]) * 12);
    put(ib, h.solid_i, ((size_t)a[CH_SOLID_V] + a[CH_SOLID_END]) * 2, ((size_t)(b[CH_SOLID_V] - a[CH_SOLID_V]) + (b[CH_SOLID_END] - a[CH_SOLID_END])) * 2);
}

// Tile-list capacity after an overflow: ov[1] = the pairs the pass needs, ov[5] = a region of the edge pass' pair stream filled up although
// the total fits (the stream is cut into 64 regions that fill unevenly): half as much again, so that the regions have headroom.
size_t grown_pair_bytes(const crh_frame* f, const uint32_t ov[8]) {
    size_t pairs = (size_t)ov[1] + (ov[1] >> 1) + 65536;
    if (ov[5] != 0) pairs = std::max(pairs, f->pair_capacity_bytes / 4 + f->pair_capacity_bytes / 8);
    return pairs * 4;
}
// The pass of this plain frame — 1: boundary edges, per-sample raster kernel; 2: strip triangles; 3: boundary edges, row-span raster kernel
// (k_raster_rows: msaa 1, no strokes) — and, through `timed`, which trial (candidate 0 edges, 1 triangles, 2 rows) its events belong to, or -1.
constexpr uint32_t kMovingListRadius = 3u;
int choose_pass(crh_scene* sc, const crh_frame* f, int* timed) {
    *timed = -1;
    const bool rows_eligible = sc->renderer->config.msaa_sample_count == 1 && !sc->has_stroke && getenv("CRH_NO_ROWS") == nullptr;
    if (f->triangle_pass_for == sc && f->triangle_pass_generation == sc->generation) return 2;
    if (getenv("CRH_TRIANGLE_PASS")) return 2;
    if (getenv("CRH_ROWS")) return rows_eligible ? 3 : 1;
    if (getenv("CRH_EDGE_PASS")) return 1;
    if (sc->d.n_shapes < 256u) return 1; // a handful of Shapes (the reference's one-Shape-per-call use): launch overhead either way, not worth synchronising frames
    uint32_t cls = 0; // size class of the target: the faster formulation depends on how many tiles a Shape spans
    for (uint64_t area = (uint64_t)f->width * f->height; area > 3u && cls < 15u; area >>= 2) ++cls;
    if (cls != sc->pass_class) { // a target of another size class: its own choice, measured once
        sc->pass_class = cls;
        sc->pass_choice = sc->pass_known[cls], sc->pass_frames = 0;
        for (crh_scene::PassTrial& t : sc->pass_trial) t.started = t.recorded = false;
    }
    if (sc->pass_choice == 3 && !rows_eligible) return 1;
    if (sc->pass_choice) return sc->pass_choice;
    const uint32_t candidates = rows_eligible ? 3u : 2u;
    static const int code_of[3] = {1, 2, 3};
    const uint32_t n = sc->pass_frames++;
    if (n < 6u * candidates) { // six frames per candidate; *timed: 2 * candidate (the group's start marker goes in front of this frame) or 2 * candidate + 1 (its end marker behind it)
        const uint32_t cand = n / 6u, k = n % 6u;
        if (k == 3u) *timed = (int)(2u * cand);
        if (k == 5u) *timed = (int)(2u * cand + 1u);
        return code_of[cand];
    }
    bool all_recorded = true;
    for (uint32_t k = 0; k < candidates; ++k) all_recorded = all_recorded && sc->pass_trial[k].recorded;
    if (all_recorded) {
        // The host runs frames ahead of the GPU: left to a query, a pipelined caller would have submitted its whole animation on the losing
        // pass before the verdict arrived. One wait, on the first frame behind the trial.
        bool done = true;
        for (uint32_t k = 0; k < candidates; ++k) done = done && hipEventSynchronize(sc->pass_trial[k].e[1]) == hipSuccess;
        if (done) {
            uint32_t best = 0;
            for (uint32_t k = 0; k < candidates; ++k) {
                float ms = 0.0f;
                (void)hipEventElapsedTime(&ms, sc->pass_trial[k].e[0], sc->pass_trial[k].e[1]);
                sc->pass_ms[k] = ms / 3.0f;
                if (sc->pass_ms[k] < sc->pass_ms[best]) best = k;
            }
            sc->pass_choice = code_of[best];
            sc->pass_known[sc->pass_class] = (uint8_t)sc->pass_choice;
            if (getenv("CRH_PASS_VERBOSE"))
                std::fprintf(stderr, "[contrast-hip] pass trial: edges %.3f ms, triangles %.3f ms, edges as row spans %.3f ms per frame -> %s\n", sc->pass_ms[0], sc->pass_ms[1],
                             candidates == 3u ? sc->pass_ms[2] : 0.0f, best == 0u ? "edges" : (best == 1u ? "triangles" : "row spans"));
            return sc->pass_choice;
        }
    } else { // a marker was skipped (could not happen in sequence): start over
        sc->pass_frames = 0;
        return 1;
    }
    return code_of[candidates - 1u]; // still waiting for the trial's events: stay on the pass of the latest frames
}

// The order in which the edge pass' raster kernels take a frame's tiles. One wavefront draws one tile, tile times spread over an order of
// magnitude (entries per tile), and a long tile that starts late ends after everything else: the tail. The kernels' own order deals 8 x 8-tile
// blocks to the eight XCDs (workgroup b runs on XCD b mod 8; neighbouring tiles share records in one L2). This keeps every tile on the XCD its
// block belongs to and lets every XCD take its HEAVY tiles — more than `factor` x the mean number of entries in the verified pass — first,
// the others in block order behind them. Any permutation draws the same image; the counts of one verified pass order all later passes.
crh_status order_tiles_heavy_first(crh_frame* f, const uint32_t* tile_count_dev, hipStream_t stream) {
    const char* e = getenv("CRH_HEAVY_FIRST"); // the factor; 0 switches the re-ordering off (A/B runs)
    // the default: every XCD's tiles by falling count, in buckets of 2^bucket_shift entries — tiles of one bucket keep their block order (the sort is
    // stable), i.e. some of the L2 locality of neighbouring tiles ("b<shift>" pins the bucket; a threshold factor keeps block order below it)
    const bool sort_all = !e || std::strcmp(e, "sort") == 0 || e[0] == 'b';
    const uint32_t bucket_shift = (e && e[0] == 'b') ? (uint32_t)atoi(e + 1) : 0u;
    const double factor = sort_all ? 1.0 : atof(e);
    f->tile_order_ready = false;
    if (!(factor > 0.0)) return CRH_OK;
    const uint32_t n_tiles = f->n_tiles, tiles_x = f->tiles_x, tiles_y = f->tiles_y;
    std::vector<uint32_t> count(n_tiles);
    HIP_TRY(hipMemcpyAsync(count.data(), tile_count_dev, (size_t)n_tiles * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    uint64_t sum = 0;
    for (uint32_t c : count) sum += c;
    const double threshold = factor * (double)sum / (double)std::max(1u, n_tiles);
    constexpr uint32_t kB = CRH_XCD_BLOCK_LOG2, kBlock = 1u << kB;
    const uint32_t blocks_x = (tiles_x + kBlock - 1u) >> kB, blocks = blocks_x * ((tiles_y + kBlock - 1u) >> kB);
    const uint32_t grid = ((blocks + 7u) / 8u) * kBlock * kBlock * 8u, turns = grid / 8u;
    std::vector<uint32_t> order(grid), rest;
    uint32_t used_turns = 0;
    rest.reserve(turns);
    for (uint32_t x = 0; x < 8u; ++x) { // XCD x: its workgroups in launch order
        uint32_t at = 0;
        rest.clear();
        for (uint32_t turn = 0; turn < turns; ++turn) {
            const uint32_t block = (turn >> (2u * kB)) * 8u + x;
            const uint32_t tx = (block % blocks_x) * kBlock + (turn & (kBlock - 1u)), ty = (block / blocks_x) * kBlock + ((turn >> kB) & (kBlock - 1u));
            if (tx >= tiles_x || ty >= tiles_y || ty < f->slab_ty0 || ty >= f->slab_ty1) continue; // (beyond the frame or outside the frame's slab: no workgroup's business)
            const uint32_t tile = ty * tiles_x + tx;
            if ((double)count[tile] > threshold) order[(at++) * 8u + x] = tile;
            else rest.push_back(tile);
        }
        if (sort_all) { // (experiment: every XCD's tiles by falling count, block order lost)
            std::vector<uint32_t> all;
            for (uint32_t k = 0; k < at; ++k) all.push_back(order[k * 8u + x]);
            all.insert(all.end(), rest.begin(), rest.end());
            std::stable_sort(all.begin(), all.end(), [&](uint32_t a, uint32_t b) { return (count[a] >> bucket_shift) > (count[b] >> bucket_shift); });
            at = 0;
            rest.clear();
            for (uint32_t tile : all) order[(at++) * 8u + x] = tile;
        }
        for (uint32_t tile : rest) order[(at++) * 8u + x] = tile;
        used_turns = std::max(used_turns, at);
        for (; at < turns; ++at) order[at * 8u + x] = 0xFFFFFFFFu; // (workgroups beyond the frame: nothing to draw)
    }
    HIP_TRY(f->tile_order.ensure((size_t)grid * 4));
    HIP_TRY(hipMemcpyAsync(f->tile_order.p, order.data(), (size_t)grid * 4, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    f->tile_order_ready = true;
    f->tile_order_places = std::max(1u, used_turns) * 8u; // (the raster kernels' grid: the places behind hold no tile)
    return CRH_OK;
}

// A frame whose tiles hold many entries on average has most of them in lists of several 64-entry chunks: the raster kernel then looks for
// its late start across the chunks (k_raster_edges<.., LONG>; 100 000 paths @ 8192^2: 67 entries per tile, raster 2.34 -> 1.33 ms). For
// short lists the plain variant is the faster one (10 000 paths @ 4096^2: 27 per tile, 0.234 against 0.255 ms), and so it is where nothing
// covers a whole tile (50 000 glyphs @ 2048^2: long lists, no such cover: 0.56 against 0.63 ms).
uint32_t long_lists(const crh_frame* f) {
    const char* e = getenv("CRH_LONG_LISTS"); // A/B runs and tests: 0 never, 1 always (read per pass: they switch it inside one process)
    const int pinned = e ? atoi(e) : -1;
    const uint32_t n_tiles = ((f->width + 15u) / 16u) * ((f->height + 15u) / 16u);
    return pinned >= 0 ? (uint32_t)(pinned != 0) : (uint32_t)(f->mean_list >= 40u && f->opaque_covers >= n_tiles / 2u);
}

// the raster kernel sorts a tile's list in LDS: size that buffer (a power of two) from the longest list seen; true when it had to grow
bool grow_sort_capacity(crh_frame* f, uint32_t longest_list) {
    f->longest_list = std::max(f->longest_list, longest_list);
    if (longest_list <= f->sort_capacity) return false;
    const uint32_t limit = 32768u / (4u * (f->renderer->config.msaa_sample_count == 4 ? 4u : 1u)); // 32 KiB of dynamic LDS per workgroup (kSortBytesMax)
    uint32_t capacity = f->sort_capacity;
    while (capacity < longest_list && capacity < limit) capacity <<= 1;
    const bool grew = capacity != f->sort_capacity;
    f->sort_capacity = capacity; // lists beyond the limit are reported by the kernel as CRH_ERR_UNSUPPORTED
    return grew;
}

// oracle/raster.hpp is_plain_instance, for every instance: the plain pass needs no 1/w and z/w planes
// Colours the late start of a tile's list (k_raster_edges) may rely on: with 0 <= alpha <= 1 and bounded components, whatever stack of
// blends precedes an opaque cover stays finite, so "source + anything x (1 - 1)" is the source — also for the colours the shortcut skipped.
bool all_colors_tame(const float* c, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        const float* v = c + 4 * i;
        if (!(v[3] >= 0.0f && v[3] <= 1.0f) || !(std::fabs(v[0]) <= 1e30f && std::fabs(v[1]) <= 1e30f && std::fabs(v[2]) <= 1e30f)) return false;
    }
    return true;
}
bool all_instances_plain(const float* t, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        const float* m = t + 16 * i;
        if (!(m[3] == 0.0f && m[7] == 0.0f && m[15] == 1.0f && m[2] == 0.0f && m[6] == 0.0f && m[14] >= 0.0f && m[14] <= 1.0f)) return false;
    }
    return true;
}
uint32_t depth_pass_mask(uint32_t compare) { // bit 0: fragment < stored passes, 1: ==, 2: >, 3: always
    switch (compare) {
        case CRH_COMPARE_NEVER: return 0u;
        case CRH_COMPARE_LESS: return 1u;
        case CRH_COMPARE_EQUAL: return 2u;
        case CRH_COMPARE_LESS_EQUAL: return 3u;
        case CRH_COMPARE_GREATER: return 4u;
        case CRH_COMPARE_NOT_EQUAL: return 5u;
        case CRH_COMPARE_GREATER_EQUAL: return 6u;
        default: return 8u;
    }
}

crh_status settle_frame(crh_frame* f);
// Host wait for what an exchange (csrc/comm.hip) enqueued on its own stream against this frame's pixels: the tail of crh_frame_exchange —
// all-to-all, composite, gather, unpack — is in flight when that call returns, and asynchronous HIP / RCCL failures of it surface here.
crh_status wait_for_external(crh_frame* f) {
    if (f->ext_read_set) HIP_TRY(hipEventSynchronize(f->ext_read));
    if (f->ext_write_set) HIP_TRY(hipEventSynchronize(f->ext_write));
    return CRH_OK;
}
crh_status settle_frame_cheaply(crh_frame* f);
// whatever touches the frame's pixels on `stream` next runs behind the exchange's last read and write of them (csrc/comm.hip)
hipError_t order_after_external(crh_frame* f, hipStream_t stream) {
    hipError_t e = hipSuccess;
    if (f->ext_read_set) e = hipStreamWaitEvent(stream, f->ext_read, 0);
    if (e == hipSuccess && f->ext_write_set) e = hipStreamWaitEvent(stream, f->ext_write, 0);
    return e;
}
// The check whether a frame's optimistic tile-list capacity sufficed is deferred to the next call that synchronises; its remedy renders
// the frame again FROM THE SCENE, so the scene must still hold what the frame shows: everything that changes a scene (new geometry,
// instance data, stroke descriptors, destruction) first settles the frames that were rendered from it.
crh_status settle_frames_of(crh_scene* sc, bool forget) {
    crh_status result = CRH_OK;
    for (crh_frame* f : sc->renderer->frames) {
        if (f->last_scene != sc) continue;
        if (f->check_pending) {
            // (the frame's own flags, behind its own raster kernel — not a wait for every stream: with two Scenes in turn the frame in flight
            // belongs to the OTHER one, and waiting for it made every upload a full stop; only a frame that ran out of capacity pays for that)
            const crh_status st = settle_frame_cheaply(f);
            if (st != CRH_OK && result == CRH_OK) result = st;
        }
        if (forget) f->last_scene = nullptr;
    }
    return result;
}

// `again`: the remedy of settle_frame — the frame's last pass once more, from the instance buffer it used
crh_status render_impl(crh_scene* sc, crh_frame* f, bool again = false) {
    crh_renderer* r = sc->renderer;
    if (!f->cleared && f->check_pending) { // rendering over existing content: that content has to be final
        const crh_status st = settle_frame(f);
        if (st != CRH_OK) return st;
    }
    const bool recorded = f->n_items != 0; // crh_scene_render_draws stored a pass in the frame
    if (!recorded && !sc->instances_set) return CRH_ERR_INVALID_ARGUMENT;
    if (!sc->capacity_known) return CRH_ERR_INVALID_ARGUMENT; // tessellate first
    if (f->carry && sc->optimistic) { // a pass into a frame that keeps its state cannot be drawn a second time: the tessellation it draws has to be known to fit
        uint32_t word = 0;
        const crh_status st = settle_tessellation(sc, &word);
        if (st != CRH_OK) return st;
    }
    HIP_TRY(hipSetDevice(r->device));
    // Three lanes: tessellation -> binning (setup, tile walks) -> raster. This frame is binned into the frame's other set of tile
    // buffers and the scene's other record buffer while the previous frame's raster kernel may still be reading its own.
    const hipStream_t bin = r->binning_stream();
    crh_frame::BinSet& set = f->sets[f->next_set];
    const int rec = sc->next_rec;
    HIP_TRY(hipStreamWaitEvent(bin, sc->tess_done, 0)); // the tessellation this frame draws (a no-op when it finished long ago)
    if (f->ext_read_set) HIP_TRY(hipStreamWaitEvent(bin, f->ext_read, 0)); // (the exchange reads the tile counts of the frame's passes: csrc/comm.hip)
    if (r->raster_exclusive && r->raster_events[0]) HIP_TRY(hipStreamWaitEvent(bin, r->raster_events[0], 0));
    // (each wait on another stream's event is a packet the binning lane — the critical one — spends 5 - 10 us on, satisfied or not: the
    // frame's set and the Scene's record buffer were as a rule last used by the same raster kernel, and then one wait says it all)
    if (set.used) HIP_TRY(hipStreamWaitEvent(bin, set.raster_done, 0));
    if (set.used) HIP_TRY(hipStreamWaitEvent(bin, set.flags_ready, 0)); // (the copy of the flag words this pass is about to clear)
    if (sc->rec_used[rec] && !(set.used && set.raster_serial == sc->rec_raster_serial[rec])) HIP_TRY(hipStreamWaitEvent(bin, sc->rec_raster_done[rec], 0));
    // A pass drawn from an optimistic tessellation (crh_scene::optimistic) takes the tessellation's status word along among its flag words — here, behind
    // tess_done and in front of the binning kernels (whose end releases this set of streams to the tessellation after next, which clears the word)
    f->last_tess_optimistic = sc->optimistic;
    if (sc->optimistic && !std::getenv("CRH_NO_STATUS_COPY")) HIP_TRY(hipMemcpyAsync(static_cast<uint32_t*>(set.overflow_p) + kTessStatusWord, sc->d.status, 4, hipMemcpyDeviceToDevice, bin));
    RasterParams p;
    p.width = f->width;
    p.height = f->height;
    p.tiles_x = f->tiles_x;
    p.tiles_y = f->tiles_y;
    p.n_tiles = f->n_tiles;
    p.slab_ty0 = f->slab_ty0, p.slab_ty1 = std::min(f->slab_ty1, f->tiles_y);
    p.shape_bounds = nullptr, p.item_elsewhere = nullptr;
    // a pass with a slab (the tile split): the Shapes' boxes let the binning kernels leave out the items without a row in it before they set them up
    static const bool no_slab_cull = getenv("CRH_NO_SLAB_CULL") != nullptr; // A/B runs
    if ((p.slab_ty0 > 0u || p.slab_ty1 < f->tiles_y) && !no_slab_cull && sc->d.n_shapes != 0u) {
        HIP_TRY(sc->shape_bounds.ensure((size_t)sc->d.n_shapes * 16));
        if (sc->bounds_generation != sc->generation) {
            launch_shape_bounds(sc->d, sc->shape_bounds.as<float>(), bin);
            sc->bounds_generation = sc->generation;
        }
        p.shape_bounds = sc->shape_bounds.as<float>();
    }
    p.winding_mask = (1u << r->config.winding_counter_bits) - 1u;
    p.clip_mask_count = (1u << r->config.clip_nesting_counter_bits) - 1u;
    p.items = recorded ? f->items.as<DrawItem>() : nullptr;
    p.n_items = recorded ? f->n_items : sc->d.n_shapes;
    p.load_existing = f->cleared ? 0u : 1u;
    const int inst = again ? f->last_instances : sc->instances_cur;
    if (!recorded) f->last_instances = inst;
    const int item_inst = again ? f->item_inst_last : f->item_inst_cur;
    if (recorded) f->item_inst_last = item_inst;
    p.transforms = recorded ? (item_inst ? f->item_transforms_b : f->item_transforms).as<float>() : (inst ? sc->transforms_b : sc->transforms).as<float>();
    p.colors = recorded ? (item_inst ? f->item_colors_b : f->item_colors).as<float>() : (inst ? sc->colors_b : sc->colors).as<float>();
    p.tile_cursor = set.tile_count_cursor.as<uint32_t>();
    p.tile_count = set.tile_count_cursor.as<uint32_t>() + f->n_tiles;
    p.tile_offset = set.tile_offset.as<uint32_t>();
    p.shape_ncand = sc->shape_ncand.as<uint32_t>();
    p.shape_prim_begin = sc->shape_prim_begin.as<uint32_t>();
    HIP_TRY(set.scan_scratch.ensure(((size_t)(f->n_tiles + 511) / 512 + 2) * 4)); // (launch_tile_bases takes a sum per 512 tiles, the other scans one per 1 024)
    if (recorded) {
        // primitive ranges per draw item; a Shape may be drawn many times, so the record capacity comes from the scan's total
        HIP_TRY(f->item_ncand.ensure((size_t)f->n_items * 4 + 4));
        HIP_TRY(f->item_prim_begin.ensure(((size_t)f->n_items + 1) * 4));
        HIP_TRY(f->item_scan_scratch.ensure(((size_t)(f->n_items + 1023) / 1024 + 2) * 4));
        HIP_TRY(f->item_nslots.ensure((size_t)f->n_items * 4 + 4));
        HIP_TRY(f->item_slot_begin.ensure(((size_t)f->n_items + 1) * 4));
        uint32_t total = f->items_total, total_slots = f->items_total_slots;
        if (!f->items_ranges_valid) {
            launch_item_ranges(sc->d, p, f->item_ncand.as<uint32_t>(), f->item_prim_begin.as<uint32_t>(), f->item_scan_scratch.as<uint32_t>(), bin);
            HIP_TRY(hipMemcpyAsync(&total, f->item_prim_begin.as<uint32_t>() + f->n_items, 4, hipMemcpyDeviceToHost, bin));
            launch_slot_ranges(sc->d, p, f->n_items, f->item_nslots.as<uint32_t>(), f->item_slot_begin.as<uint32_t>(), f->item_scan_scratch.as<uint32_t>(), bin);
            HIP_TRY(hipMemcpyAsync(&total_slots, f->item_slot_begin.as<uint32_t>() + f->n_items, 4, hipMemcpyDeviceToHost, bin));
            HIP_TRY(r->sync());
            f->items_total = total;
            f->items_total_slots = total_slots;
            f->items_ranges_valid = true;
        }
        if (total >= 0xFFFFFFF0u || total_slots >= 0xFFFFFFF0u) return CRH_ERR_UNSUPPORTED;
        HIP_TRY(sc->prim_rec[rec].ensure(std::max(((size_t)total + 64) * 128, ((size_t)total_slots + 64) * 32)));
        p.prim_capacity = total + 64u;
        p.slot_capacity = total_slots + 64u;
        p.slot_begin = f->item_slot_begin.as<uint32_t>();
        p.shape_ncand = f->item_ncand.as<uint32_t>();
        p.shape_prim_begin = f->item_prim_begin.as<uint32_t>();
    } else {
        // every candidate triangle gets a record slot: an upper bound follows from the tessellation totals
        const uint32_t* t = sc->totals_host;
        const size_t prim_capacity = (size_t)t[CH_LINE_V] + 3u * (size_t)t[CH_JOINT] + t[CH_SOLID_V] + t[CH_IQ] + t[CH_IC_V] / 3u + t[CH_RQ] + t[CH_RC_V] / 3u + t[CH_HULL] + 64;
        // the edge pass: four slots per stroke / curve triangle, one per polygon vertex and hull vertex, the per-Shape cover slots
        const size_t slot_capacity = 4u * ((size_t)t[CH_LINE_V] + 3u * (size_t)t[CH_JOINT] + t[CH_IQ] + t[CH_IC_V] / 3u + t[CH_RQ] + t[CH_RC_V] / 3u) + t[CH_SOLID_V] + 4u * (size_t)t[CH_HULL] +
                                     40u * (size_t)sc->d.n_shapes + 64; // item_slots(): the hull region is sized for cover triangles (hull vertices <= candidates)
        if (prim_capacity >= 0xFFFFFFF0u || slot_capacity >= 0xFFFFFFF0u) return CRH_ERR_UNSUPPORTED; // 0xFFFFFFFF pads the tile sort
        HIP_TRY(sc->prim_rec[rec].ensure(std::max(prim_capacity * 128, slot_capacity * 32)));
        p.prim_capacity = (uint32_t)prim_capacity;
        p.slot_capacity = (uint32_t)slot_capacity;
        p.slot_begin = sc->shape_slot_begin.as<uint32_t>();
    }
    {   // totals of the scene's tessellation (hull vertices are at most the hull candidates; on the benchmark scene about half of them)
        const uint32_t* t = sc->totals_host;
        const uint64_t tris = (uint64_t)t[CH_LINE_V] + 3u * (uint64_t)t[CH_JOINT] + t[CH_IQ] + t[CH_IC_V] / 3u + t[CH_RQ] + t[CH_RC_V] / 3u, edges = (uint64_t)t[CH_SOLID_V] + t[CH_HULL] / 2u;
        const uint64_t per = std::max<uint64_t>(1, sc->d.n_shapes);
        p.hint_tris = (uint32_t)std::min<uint64_t>(0xFFFFFFFFu, tris * p.n_items / per), p.hint_edges = (uint32_t)std::min<uint64_t>(0xFFFFFFFFu, edges * p.n_items / per);
    }
    p.scan_scratch = set.scan_scratch.as<uint32_t>();
    p.prim_rec = static_cast<PrimRec*>(sc->prim_rec[rec].p);
    InstanceSlot& slot = recorded ? f->item_slot[item_inst] : sc->slot[inst];
    if (slot.was_written) HIP_TRY(hipStreamWaitEvent(bin, slot.ready, 0)); // the copy of this set on the upload stream
    const bool projective = recorded ? f->item_projective_of[item_inst] != 0 : sc->instances_projective_of[inst];
    const bool tame_colors = recorded ? f->item_tame_of[item_inst] != 0 : sc->instances_tame_of[inst];
    p.prim_proj = nullptr;
    if (projective) {
        HIP_TRY(sc->prim_proj[rec].ensure((size_t)p.prim_capacity * 32));
        p.prim_proj = static_cast<PrimProj*>(sc->prim_proj[rec].p);
    }
    p.depth = f->depth.as<float>(); // nullptr without a depth attachment
    p.depth_pass_mask = depth_pass_mask(r->config.depth_compare);
    p.depth_write = r->config.depth_write_enabled;
    p.cull_mode = r->config.cull_mode;
    // The general pass keeps the reference's triangle strips (raster.hip): clip nesting / alpha contexts, perspective, depth, and face
    // culling (a cull decision is per strip triangle). Everything else is the edge pass (raster_edges.hip).
    p.general = (projective || p.depth || r->config.cull_mode != CRH_CULL_NONE || (recorded && f->items_need_ops) || f->carry) ? 1u : 0u;
    p.state_stencil = nullptr, p.state_alpha = nullptr, p.state_color = nullptr, p.state_load = 0u, p.state_layers = 0u;
    p.winding_bits = r->config.winding_counter_bits;
    if (f->carry) { // the frame keeps clip counters, winding counters, saved alphas and sample colours from pass to pass
        const size_t n_samples = (size_t)f->width * f->height * r->config.msaa_sample_count;
        HIP_TRY(f->state_stencil.ensure(n_samples));
        HIP_TRY(f->state_alpha.ensure(std::max<size_t>(1, r->config.alpha_layer_count) * n_samples * 4));
        HIP_TRY(f->state_color.ensure(n_samples * 16));
        p.state_stencil = f->state_stencil.as<uint8_t>(), p.state_alpha = f->state_alpha.as<float>(), p.state_color = f->state_color.as<float>();
        p.state_layers = std::min<uint32_t>(r->config.alpha_layer_count, 4u);
        p.state_load = f->carry_valid ? 1u : 0u;
    }
    int timed = -1;
    const int pass = p.general == 0u ? choose_pass(sc, f, &timed) : 2;
    const bool edges = pass != 2;
    // k_raster_rows keeps fill + 65536 * hull in one cell: exact below 2^15 entries per tile. Lists in place grow by half at most between two
    // verified passes, so a frame that has ever shown a list of 16 384 entries keeps the per-sample kernel.
    p.rows = (pass == 3 && f->longest_list < 16384u) ? 1u : 0u;
    p.fill_cells = f->longest_list < 16384u ? 1u : 0u; // (k_raster_fill keeps the same cell)
    if (edges != f->last_edges) f->pairs_known = false;
    f->last_edges = edges;
    crh_scene::PassTrial* trial = (p.general == 0u && timed >= 0) ? &sc->pass_trial[timed / 2] : nullptr;
    if (trial) {
        for (hipEvent_t& e : trial->e)
            if (!e) HIP_TRY(hipEventCreate(&e));
        if (timed % 2 == 0) { // (fires when the raster kernel of the frame before is through)
            HIP_TRY(hipEventRecord(trial->e[0], r->stream));
            trial->started = true, trial->recorded = false;
        }
    }
    p.slots = static_cast<uint8_t*>(sc->prim_rec[rec].p);
    p.overflow = static_cast<uint32_t*>(set.overflow_p);
    p.pair_cursor = static_cast<uint32_t*>(set.overflow_p) + 8; // 64 sub-stream cursors
    p.sort_capacity = f->sort_capacity;
    p.long_lists = long_lists(f);
    p.rgba8 = f->rgba8.as<uint8_t>();
    p.format = f->format;
    p.debug = getenv("CRH_RASTER_DEBUG") ? (uint32_t)atoi(getenv("CRH_RASTER_DEBUG")) : 0u;
    p.occlude = tame_colors ? 1u : 0u;
    r->begin_marks(2);
    // Rendering over existing content is not repeatable (the target is read and overwritten), so the optimistic tile-list capacity with a
    // transparent re-run after the fact is only used for cleared frames; otherwise the pair count is checked before the raster kernel runs.
    // ... and a pass that writes depth is not repeatable either: the first attempt's depth writes would be tested against by the redraw
    // ... nor is a pass that starts from the state its predecessors left with the frame and stores its own
    if (!f->cleared || (f->depth.p && r->config.depth_write_enabled) || f->carry) f->pairs_known = false;
    const bool no_direct = getenv("CRH_NO_DIRECT_LISTS") != nullptr; // A/B runs and tests (read per pass: a test switches it inside one process)
    // Geometry that has stayed for two passes without its lists in place (a Scene drawn into this frame after another one, paths uploaded
    // again into the Scene): one verified pass more, which puts them in place. (Not at once: a caller that uploads new paths for every
    // frame would pay a read-back per frame for places it never uses.)
    if (!recorded) {
        if (f->seen_scene == sc && f->last_pass_instances != ~0ull && f->last_pass_instances != sc->instances_version) f->moving = true;
        f->last_pass_instances = sc->instances_version;
        if (f->seen_scene == sc && f->last_pass_geometry != 0 && f->last_pass_geometry != sc->generation) f->moving = true; // (new paths of the same structure: the lists change like those of moving instances)
        f->last_pass_geometry = sc->generation;
    }
    if (f->seen_scene == sc && f->seen_generation == sc->lineage) f->seen_passes += 1u;
    else f->seen_scene = sc, f->seen_generation = sc->lineage, f->seen_passes = 0u;
    if (edges && !recorded && f->pairs_known && f->seen_passes == 2u && !no_direct && f->direct_misses < 3u &&
        !(f->direct_ready && f->direct_scene == sc && f->direct_generation == sc->lineage))
        f->pairs_known = false;
    const bool direct = edges && !recorded && f->pairs_known && f->direct_ready && f->direct_scene == sc && f->direct_generation == sc->lineage && f->direct_misses < 3u && !no_direct;
    p.direct = direct ? 1u : 0u;
    const bool skip_queue = edges && !recorded && f->pairs_known && !f->queue_seen && f->direct_scene == sc && f->direct_generation == sc->lineage;
    p.skip_queue = skip_queue ? 1u : 0u;
    p.tile_base = (f->base_cur ? f->tile_base_b : f->tile_base).as<uint32_t>();
    p.tile_order = f->tile_order_ready ? f->tile_order.as<uint32_t>() : nullptr;
    p.order_places = f->tile_order_ready ? f->tile_order_places : 0u;
    const bool no_batches = getenv("CRH_NO_BIN_BATCHES") != nullptr; // A/B runs and tests (read per pass: they switch it inside one process)
    const bool batches = edges && !no_batches && f->n_bin_batches != 0u && f->batches_scene == sc && f->batches_generation == sc->lineage && f->batches_items == p.n_items;
    p.bin_batches = batches ? f->bin_batches.as<uint32_t>() : nullptr, p.n_bin_batches = batches ? f->n_bin_batches : 0u;
    p.item_cost = nullptr;
    static const bool bin_dump = getenv("CRH_BIN_DUMP") != nullptr; // tools/bin_phases.py (a library built with -DCRH_ABLATE): a record per workgroup behind the costs
    if (edges && (!f->pairs_known || bin_dump) && !no_batches) { // (a verified pass: it says what every item takes of a batch)
        HIP_TRY(f->item_cost.ensure((size_t)p.n_items * ((bin_dump || (p.debug & 65536u)) ? 40 : 8) + 48)); // (debug bit 16, -DCRH_ABLATE builds: every workgroup leaves a record behind the costs)
        p.item_cost = f->item_cost.as<uint32_t>();
    }
    if (direct) f->pair_capacity_bytes = std::max<size_t>(f->pair_capacity_bytes, (size_t)f->direct_entries * 6); // (half as much again: the places move with the scene; a pass whose places do not fit says so, overflow[0])
    HIP_TRY(set.tile_list.ensure(f->pair_capacity_bytes));
    for (int attempt = 0; attempt < 6; ++attempt) { // (a region of the edge pass' pair stream may fill before the total does: each retry adds headroom)
        p.tile_list = set.tile_list.as<uint32_t>();
        p.pair_capacity = (uint32_t)(set.tile_list.cap / 4);
        if (edges) {
            HIP_TRY(set.pair_tile.ensure(set.tile_list.cap));
            HIP_TRY(set.pair_key.ensure(set.tile_list.cap));
            HIP_TRY(set.pair_pos.ensure(set.tile_list.cap));
            p.pair_pos = set.pair_pos.as<uint32_t>();
            p.pair_tile = set.pair_tile.as<uint32_t>();
            p.pair_key = set.pair_key.as<uint32_t>();
            HIP_TRY(set.bin_queue.ensure((size_t)p.n_items * 4 + 4));
            p.bin_queue = set.bin_queue.as<uint32_t>();
            if (p.shape_bounds) { // a pass with a slab: which items miss it (transforms, items and slab are all set by now)
                HIP_TRY(set.item_elsewhere.ensure((size_t)p.n_items + 4));
                launch_slab_items(p, set.item_elsewhere.as<uint8_t>(), bin);
                p.item_elsewhere = set.item_elsewhere.as<uint8_t>();
            }
            launch_bin_edges(sc->d, p, r->config.msaa_sample_count, bin, r->mark_fn_bin(), r, sc->vertices_free);
        } else
        launch_bin(sc->d, p, r->config.msaa_sample_count, bin, r->mark_fn_bin(), r, sc->vertices_free);
        if (f->pairs_known) break;
        uint32_t ov[8];
        HIP_TRY(hipMemcpyAsync(ov, p.overflow, 32, hipMemcpyDeviceToHost, bin));
        HIP_TRY(r->sync());
        grow_sort_capacity(f, ov[3]);
        p.sort_capacity = f->sort_capacity;
        if (f->longest_list >= 16384u) p.rows = 0u, p.fill_cells = 0u;
        if (edges) f->mean_list = ov[1] / std::max(1u, p.n_tiles), f->opaque_covers = ov[4]; // (the triangle pass of the same Scene has other entries, and no such variant)
        p.long_lists = long_lists(f);
        if (getenv("CRH_PASS_VERBOSE")) std::fprintf(stderr, "[contrast-hip] %u entries in %u tiles (longest list %u, %u opaque whole-tile covers): %s raster variant\n", ov[1], p.n_tiles, ov[3], ov[4], p.long_lists ? "long-list" : "plain");
        if (edges && ov[7] != 0) { // an unclosed boundary chain: this pass and the following ones of this Scene into this frame as strip triangles
            if (getenv("CRH_PASS_VERBOSE")) std::fprintf(stderr, "[contrast-hip] a boundary edge with a non-finite end point: this Scene goes to the triangle pass\n");
            f->triangle_pass_for = sc, f->triangle_pass_generation = sc->generation;
            return render_impl(sc, f, again);
        }
        if (ov[0] == 0 && ov[5] == 0) {
            f->pairs_known = true; // from now on this frame's passes run without the read-back (checked after the fact, settle_frame)
            if (edges) { // ... and the edge pass keeps the places of this frame's lists (the counts are final: the stream was waited for)
                uint32_t total = 0;
                HIP_TRY(f->tile_caps.ensure((size_t)p.n_tiles * 4 + 4));
                HIP_TRY(f->tile_base.ensure((size_t)p.n_tiles * 4 + 4));
                HIP_TRY(f->tile_base_b.ensure((size_t)p.n_tiles * 4 + 4));
                uint32_t* const places = (f->base_cur ? f->tile_base_b : f->tile_base).as<uint32_t>(); // (the buffer the next pass reads)
                launch_tile_bases(p.tile_count, f->tile_caps.as<uint32_t>(), places, p.scan_scratch, p.n_tiles, p.tiles_x, f->moving ? kMovingListRadius : 0u, bin);
                HIP_TRY(hipMemcpyAsync(&total, places + p.n_tiles, 4, hipMemcpyDeviceToHost, bin));
                HIP_TRY(hipStreamSynchronize(bin));
                f->direct_entries = total, f->direct_ready = true, f->direct_scene = sc, f->direct_generation = sc->lineage;
                f->places_instances = recorded ? ~0ull : sc->instances_version;
                f->queue_seen = ov[6] != 0;
                f->n_bin_batches = 0;
                if (p.item_cost && !bin_itemwise(p) && p.n_items != 0u) { // k_bin_flat wrote every item's cost: the later passes' batches
                    std::vector<uint32_t> cost((size_t)p.n_items * 2), starts;
                    HIP_TRY(hipMemcpyAsync(cost.data(), p.item_cost, cost.size() * 4, hipMemcpyDeviceToHost, bin));
                    HIP_TRY(hipStreamSynchronize(bin));
                    flat_batches(cost.data(), p.n_items, starts);
                    HIP_TRY(f->bin_batches.ensure(starts.size() * 4));
                    HIP_TRY(hipMemcpyAsync(f->bin_batches.p, starts.data(), starts.size() * 4, hipMemcpyHostToDevice, bin));
                    HIP_TRY(hipStreamSynchronize(bin));
                    f->n_bin_batches = (uint32_t)(starts.size() / 2), f->batches_items = p.n_items, f->batches_scene = sc, f->batches_generation = sc->lineage, f->batches_age = 0;
                    if (getenv("CRH_PASS_VERBOSE")) std::fprintf(stderr, "[contrast-hip] %u items in %u batches of k_bin_flat\n", p.n_items, f->n_bin_batches);
                }
            }
            { // (both passes: the counts of this verified pass order the frame's later ones — the triangle pass' lists go with its strip triangles)
                const crh_status ordered = order_tiles_heavy_first(f, p.tile_count, bin);
                if (ordered != CRH_OK) return ordered;
            }
            break;
        }
        if (attempt == 5) { // (six doublings of the pair stream were not enough: not a capacity problem)
            g_error = "the tile lists of this pass do not fit after six attempts";
            return CRH_ERR_UNSUPPORTED;
        }
        f->pair_capacity_bytes = grown_pair_bytes(f, ov);
        HIP_TRY(set.tile_list.ensure(f->pair_capacity_bytes));
        r->begin_marks(2);
    }
    // With the lists in place nothing runs on this lane behind the binning kernels: "the vertex streams are free", "the instance data has been
    // read", "the slot ranges are free" and "binned" are one point in time, and the event launch_bin_edges recorded there (the Scene's
    // vertices_free) stands for all four — three packets less on the critical lane.
    const bool one_event = direct;
    sc->last_render_one_event = one_event;
    const hipEvent_t vertices_free_now = sc->vertices_free; // (the handle: flip_tess_set swaps the Scene's two)
    if (one_event) {
        if (slot.read_done) slot.read_event = vertices_free_now, slot.was_read = true;
    } else if (slot.read_done) { // k_prim_setup, the only reader of the instance data, is behind us on this stream
        HIP_TRY(hipEventRecord(slot.read_done, bin));
        slot.read_event = slot.read_done;
        slot.was_read = true;
    }
    if (edges) {
        if (!one_event) HIP_TRY(hipEventRecord(sc->ranges_free, bin)); // k_bin_edges, the only reader of the slot ranges, is behind us
        launch_scatter(p, bin, r->mark_fn_bin(), r);
        if (direct && f->places_instances != sc->instances_version) { // the places of the next pass into this frame, from this pass' counts: behind the event the raster kernel waits for, beside that kernel (resident instances: the same counts, the same places)
            f->places_instances = sc->instances_version;
            uint32_t* const next_places = (f->base_cur ? f->tile_base : f->tile_base_b).as<uint32_t>();
            //  The buffer written here is the one the raster kernel of the PREVIOUS pass into this frame may still be reading its
            // places from (r.tile_base[tile] at workgroup start) when the target is not consumed between two passes: the binning stream has
            // only waited for the pass before that one (the other BinSet). Nothing else on this stream runs before that kernel is through
            // anyway — the next pass' binning waits for the same event when it takes that pass' BinSet.
            const crh_frame::BinSet& previous = f->sets[f->last_set];
            // (asked on the host first: a wait packet costs the binning lane 5 - 10 us whether it is satisfied or not, and with two targets in turn — an
            // animation — the pass in question finished a step ago)
            if (previous.used && &previous != &set && hipEventQuery(previous.raster_done) != hipSuccess) HIP_TRY(hipStreamWaitEvent(bin, previous.raster_done, 0));
            launch_tile_bases(p.tile_count, f->tile_caps.as<uint32_t>(), next_places, p.scan_scratch, p.n_tiles, p.tiles_x, f->moving ? kMovingListRadius : 0u, bin);
            f->base_cur ^= 1;
        }
    } else {
        launch_fill(sc->d, p, r->config.msaa_sample_count, bin, r->mark_fn_bin(), r, sc->ranges_free);
    }
    if (!one_event) HIP_TRY(hipEventRecord(set.bin_done, bin));
    // ---- raster lane
    HIP_TRY(hipStreamWaitEvent(r->stream, one_event ? vertices_free_now : set.bin_done, 0));
    HIP_TRY(order_after_external(f, r->stream));
    r->begin_marks(0);
    // algorithmic bytes of the raster step: every emitted byte read once + 64 B transform + 16 B colour per shape,
    // the framebuffer written once
    const uint64_t raster_bytes = sc->emitted_bytes + (uint64_t)p.n_items * 80 + (uint64_t)f->image_bytes();
    if (f->carry && (!f->carry_valid || f->carry_recolor)) { // on the raster stream: behind every earlier pass into the frame, in front of this one
        const size_t n_samples = (size_t)f->width * f->height * r->config.msaa_sample_count;
        if (!f->carry_valid) {
            HIP_TRY(hipMemsetAsync(f->state_stencil.p, 0, n_samples, r->stream));
            HIP_TRY(hipMemsetAsync(f->state_alpha.p, 0, std::max<size_t>(1, r->config.alpha_layer_count) * n_samples * 4, r->stream));
        }
        if (p.load_existing) launch_state_colors_from_image(p, r->config.msaa_sample_count, r->stream);
        else HIP_TRY(hipMemsetAsync(f->state_color.p, 0, n_samples * 16, r->stream));
        f->carry_valid = true, f->carry_recolor = false;
        p.state_load = 1u; // (the planes now hold the frame's state: zero counters, the image's colours)
    }
    if (edges)
        launch_raster_edges(sc->d, p, r->config.msaa_sample_count, r->stream, r->mark_fn(), r, raster_bytes, sc->has_stroke);
    else
        launch_raster(sc->d, p, r->config.msaa_sample_count, r->stream, r->mark_fn(), r, raster_bytes, sc->has_stroke);
    HIP_TRY(hipEventRecord(set.raster_done, r->stream));
    HIP_TRY(hipEventRecord(sc->rec_raster_done[rec], r->stream));
    HIP_TRY(hipStreamWaitEvent(r->aux_stream, set.raster_done, 0));
    HIP_TRY(hipMemcpyAsync(set.flags_host, set.overflow_p, sizeof(uint32_t) * (kTessStatusWord + 1u), hipMemcpyDeviceToHost, r->aux_stream));
    HIP_TRY(hipEventRecord(set.flags_ready, r->aux_stream));
    set.raster_serial = sc->rec_raster_serial[rec] = ++r->render_serial;
    r->raster_events[1] = r->raster_events[0], r->raster_events[0] = sc->rec_raster_done[rec];
    if (trial && timed % 2 == 1 && trial->started) {
        HIP_TRY(hipEventRecord(trial->e[1], r->stream));
        trial->recorded = true;
    }
    set.used = true;
    sc->rec_used[rec] = true, sc->rec_used_ever[rec] = true;
    sc->rendered_once = true;
    HIP_TRY(hipGetLastError());
    f->last_set = f->next_set;
    if (r->pipeline) { // alternate the buffers; without pipelining everything is ordered on one stream anyway
        f->next_set = (f->next_set + 1) % kPipelineDepth;
        sc->next_rec = (sc->next_rec + 1) % kPipelineDepth;
    }
    f->cleared = false;
    f->last_scene = sc;
    f->check_pending = true;
    f->last_direct = direct;
    f->last_skipped_queue = skip_queue;
    f->last_used_batches = batches;
    if (batches) f->batches_age += 1u;
    f->counts_describe_pixels = p.load_existing == 0u;
    return CRH_OK;
}

// after a sync: if the optimistic bin capacity was too small, grow it and render again (the frame content is recomputed from scratch)
// The runs of k_bin_flat are cut by what the items cost in the frame's verified pass; when the instances move (a zoom), a run no longer fits
// one batch and its workgroup takes several turns — correct, but the balance is gone. A quarter of the workgroups in that state: the
// runs are dropped and the next pass is a verified one, which measures again. (Called where the frame's flags are read anyway.)
void stale_batches_known(crh_frame* f, uint32_t extra) {
    if (!f->last_used_batches || f->n_bin_batches == 0u) return;
    if ((uint64_t)extra * 4u > f->n_bin_batches && f->batches_age >= 16u) { // (not more often than every sixteen passes: a verified pass drains the pipeline)
        if (getenv("CRH_PASS_VERBOSE")) std::fprintf(stderr, "[contrast-hip] %u of %u runs of k_bin_flat needed more than one turn: the costs are measured again\n", extra, f->n_bin_batches);
        f->n_bin_batches = 0, f->pairs_known = false;
    }
}
crh_status stale_batches(crh_frame* f, hipStream_t stream) {
    if (!f->last_used_batches || f->n_bin_batches == 0u) return CRH_OK;
    uint32_t extra = 0;
    HIP_TRY(hipMemcpyAsync(&extra, static_cast<const uint32_t*>(f->sets[f->last_set].overflow_p) + kExtraTurnsWord, 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if ((uint64_t)extra * 4u > f->n_bin_batches && f->batches_age >= 16u) { // (not more often than every sixteen passes: a verified pass drains the pipeline)
        if (getenv("CRH_PASS_VERBOSE")) std::fprintf(stderr, "[contrast-hip] %u of %u runs of k_bin_flat needed more than one turn: the costs are measured again\n", extra, f->n_bin_batches);
        f->n_bin_batches = 0, f->pairs_known = false;
    }
    return CRH_OK;
}
bool tess_status_overflowed(uint32_t word) { return word != 0xFFFFFFFFu && (word & 0xFFu) >= 0x80u; }
crh_status settle_frame(crh_frame* f) {
    if (!f->check_pending) return CRH_OK;
    crh_renderer* r = f->renderer;
    uint32_t ov[8];
    HIP_TRY(r->sync());
    HIP_TRY(hipMemcpyAsync(ov, f->sets[f->last_set].overflow_p, 32, hipMemcpyDeviceToHost, r->stream));
    uint32_t tess_word = 0xFFFFFFFFu;
    if (f->last_tess_optimistic) HIP_TRY(hipMemcpyAsync(&tess_word, static_cast<const uint32_t*>(f->sets[f->last_set].overflow_p) + kTessStatusWord, 4, hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(r->sync());
    const bool tess_overflow = tess_status_overflowed(tess_word) && f->last_scene != nullptr;
    if (tess_overflow) { // the optimistic tessellation this pass drew did not fit its streams: size them from the totals, tessellate, and draw again
        if (getenv("CRH_PASS_VERBOSE")) std::fprintf(stderr, "[contrast-hip] the tessellation of re-uploaded paths outgrew the streams of the paths before them: sized again, the pass is drawn again\n");
        crh_scene* const sc = f->last_scene;
        sc->capacity_known = false, sc->optimistic = false;
        const crh_status st = run_tessellation(sc, true);
        if (st != CRH_OK) return st;
    }
    {
        const crh_status stale = stale_batches(f, r->stream);
        if (stale != CRH_OK) return stale;
    }
    f->check_pending = false;
    if (ov[2] != 0) return CRH_ERR_UNSUPPORTED; // a tile list longer than the LDS sort can hold (documented limit)
    const bool sort_overflow = grow_sort_capacity(f, ov[3]);
    const bool unclosed = ov[7] != 0 && f->last_scene && !(f->triangle_pass_for == f->last_scene && f->triangle_pass_generation == f->last_scene->generation); // (the edge pass drew it: see crh_frame::triangle_pass_for)
    if (unclosed) {
        if (getenv("CRH_PASS_VERBOSE")) std::fprintf(stderr, "[contrast-hip] a boundary edge with a non-finite end point (found after the pass): this Scene goes to the triangle pass\n");
        f->triangle_pass_for = f->last_scene, f->triangle_pass_generation = f->last_scene->generation;
    }
    if (ov[0] != 0 && f->last_direct) { // a tile outgrew the place the earlier frame left it: the exact way again, with the read-back, and new places
        if (getenv("CRH_PASS_VERBOSE")) std::fprintf(stderr, "[contrast-hip] direct tile lists: a list outgrew its place, the pass is drawn again\n");
        f->direct_ready = false, f->pairs_known = false, f->direct_misses += 1u, f->direct_clean = 0u;
    }
    const bool queue_missed = ov[6] != 0 && f->last_skipped_queue; // items were queued for a kernel that was not launched: again, with it
    if (queue_missed) f->queue_seen = true, f->pairs_known = false;
    if (ov[0] != 0 || ov[5] != 0 || sort_overflow || unclosed || queue_missed || tess_overflow) {
        if (tess_overflow) f->pairs_known = false; // (what the pass learned, it learned from stale rows)
        if (ov[0] != 0 || ov[5] != 0) f->pair_capacity_bytes = std::max(f->pair_capacity_bytes, grown_pair_bytes(f, ov)); // learned either way
        // ... and the remedy is drawn the VERIFIED way (round 6): with the lists not in place (CRH_NO_DIRECT_LISTS, a frame on the exact way) an optimistic second
        // attempt can fill one of the pair stream's 64 regions again — the whole pass draws nothing again, and nobody looked a second time: the frame stayed
        // transparent (tests/test_gpu_parity.py::test_binning_batches_by_cost… under that pin: instances that zoom by 3 x between two passes, 69 614 -> 158 801 entries)
        if (ov[0] != 0 || ov[5] != 0) f->pairs_known = false;
        // crh_frame_clear after the pass: what it drew is discarded anyway, and the caller's clear must stay in force for the next pass
        if (f->last_scene && !f->cleared) {
            f->cleared = true; // the pass is drawn again from scratch (only cleared frames take the optimistic path, see render_impl)
            if (f->last_scene->pass_choice == 0) { // a frame of the trial that drew nothing would win every race: the trial starts over
                f->last_scene->pass_frames = 0;
                for (crh_scene::PassTrial& t : f->last_scene->pass_trial) t.started = t.recorded = false;
            }
            crh_status st = render_impl(f->last_scene, f, true);
            if (st != CRH_OK) return st;
            HIP_TRY(r->sync());
            f->check_pending = false;
        }
    }
    return CRH_OK;
}
// Waits for the last render into this frame only, reads the frame's own flags on the side stream, and pays for the full settle (which
// waits for everything in flight and draws the frame again) only when the frame really ran out of tile-list or sort capacity.
crh_status settle_frame_cheaply(crh_frame* f) {
    crh_renderer* r = f->renderer;
    const crh_frame::BinSet& set = f->sets[f->last_set];
    if (set.used) HIP_TRY(hipEventSynchronize(set.flags_ready)); // (behind raster_done: the pixels are written, the pass' flag words are on the host)
    if (!f->check_pending) return CRH_OK;
    const uint32_t* ov = set.flags_host;
    const uint32_t limit = 32768u / (4u * (r->config.msaa_sample_count == 4 ? 4u : 1u));
    const bool sort_too_small = ov[3] > f->sort_capacity && f->sort_capacity < limit;
    if (ov[0] != 0 || ov[5] != 0 || ov[2] != 0 || ov[7] != 0 || sort_too_small || (ov[6] != 0 && f->last_skipped_queue) || (f->last_tess_optimistic && tess_status_overflowed(ov[kTessStatusWord])))
        return settle_frame(f);
    f->longest_list = std::max(f->longest_list, ov[3]);
    if (f->last_direct && ++f->direct_clean >= 32u) f->direct_misses = 0u, f->direct_clean = 0u;
    stale_batches_known(f, ov[kExtraTurnsWord]);
    f->check_pending = false;
    return CRH_OK;
}
} // namespace

namespace crh {
void set_last_error(const std::string& text) { g_error = text; } // used by csrc/text.cpp
} // namespace crh

extern "C" {

const char* crh_last_error(void) { return g_error.c_str(); }

const char* crh_version(void) { return "contrast_hip 0.1 (gfx950)"; }

crh_status crh_renderer_create(const crh_config* config, int device_ordinal, crh_renderer** out) {
    if (!config || !out) return CRH_ERR_INVALID_ARGUMENT;
    // renderer.rs:433-435
    if (config->winding_counter_bits == 0 || config->clip_nesting_counter_bits + config->winding_counter_bits > 8) return CRH_ERR_NUMBER_OF_STENCIL_BITS_IS_UNSUPPORTED;
    if (!(config->msaa_sample_count == 1 || config->msaa_sample_count == 4)) return CRH_ERR_UNSUPPORTED;
    if (config->cull_mode > CRH_CULL_BACK || config->depth_compare > CRH_COMPARE_GREATER_EQUAL || config->depth_write_enabled > 1) return CRH_ERR_INVALID_ARGUMENT;
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (count <= 0) {
        g_error = "no HIP device: contrast_hip has no CPU fallback";
        return CRH_ERR_HIP;
    }
    if (device_ordinal < 0 || device_ordinal >= count) return CRH_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(device_ordinal));
    crh_renderer* r = new crh_renderer;
    r->config = *config;
    r->device = device_ordinal;
    r->pipeline = getenv("CRH_NO_PIPELINE") == nullptr;
    // CRH_CU_SPLIT=n (experiment): the tessellation and binning lanes get n of the device's compute units, the raster
    // lane the others (the mask's bits are dealt round-robin to the XCDs by the driver, so both sets spread evenly over the eight dies).
    int front_cus = 0;
    if (const char* e = getenv("CRH_CU_SPLIT")) front_cus = atoi(e);
    hipDeviceProp_t prop;
    if (!hip_ok(hipGetDeviceProperties(&prop, device_ordinal), "hipGetDeviceProperties")) {
        delete r;
        return CRH_ERR_HIP;
    }
    const int n_cus = prop.multiProcessorCount;
    r->n_cus = n_cus;
    // Stream priorities of the tessellation, binning and raster lanes (-1 high, 0 normal, 1 low; CRH_LANE_PRIORITY="t b r" for A/B runs). The
    // tessellation of frame i + 1 runs in the gap between two raster kernels beside the binning of frame i, which the next raster kernel waits
    // for: at LOW priority its workgroups take the slots the binning workgroups leave (round 5, with the one-pass kernel — all of a frame's
    // tessellation in one grid —: S10k 0.334 -> 0.314 ms per step, glyphs 0.721 -> 0.711; with the four small kernels of the two-pass path
    // priorities had measured within noise). Starting it behind that binning instead — beside the raster kernel — leaves it without wave
    // slots until that grid drains, at any priority: 0.398.
    int lane_priority[3] = {1, 0, 0};
    if (const char* e = getenv("CRH_LANE_PRIORITY")) { // (three values out of -1, 0, 1, or the variable is ignored)
        int p[3] = {0, 0, 0};
        if (sscanf(e, "%d %d %d", &p[0], &p[1], &p[2]) == 3 && p[0] >= -1 && p[0] <= 1 && p[1] >= -1 && p[1] <= 1 && p[2] >= -1 && p[2] <= 1) lane_priority[0] = p[0], lane_priority[1] = p[1], lane_priority[2] = p[2];
        else std::fprintf(stderr, "[contrast-hip] CRH_LANE_PRIORITY=\"%s\" is not three values out of -1, 0, 1: ignored\n", e);
    }
    int prio_low = 0, prio_high = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
    // A stream is created with a CU mask OR with a priority (HIP offers no call for both): a requested CU partition (CRH_CU_SPLIT, an experiment) comes first
    // for the lanes it concerns — their priorities are then not applied (the tessellation lane's default low priority used to switch its mask off silently).
    const bool partition = r->pipeline && front_cus > 0 && front_cus < n_cus;
    auto make_stream = [&](hipStream_t* st, int lane, int priority = 0) -> bool { // lane 0: unrestricted, 1: front lanes, 2: raster lane
        if (priority != 0 && !(partition && lane != 0)) return hip_ok(hipStreamCreateWithPriority(st, hipStreamNonBlocking, priority < 0 ? prio_high : prio_low), "hipStreamCreateWithPriority");
        if (!partition || lane == 0) return hip_ok(hipStreamCreateWithFlags(st, hipStreamNonBlocking), "hipStreamCreate");
        std::vector<uint32_t> mask((size_t)(n_cus + 31) / 32, 0u);
        for (int c = 0; c < n_cus; ++c)
            if ((c < front_cus) == (lane == 1)) mask[(size_t)c / 32] |= 1u << (c % 32);
        return hip_ok(hipExtStreamCreateWithCUMask(st, (uint32_t)mask.size(), mask.data()), "hipExtStreamCreateWithCUMask");
    };
    if (!make_stream(&r->stream, 2, lane_priority[2]) || !make_stream(&r->tess_stream, 1, lane_priority[0]) || !make_stream(&r->bin_stream, 1, lane_priority[1]) || !make_stream(&r->aux_stream, 0) ||
        !make_stream(&r->upload_stream, 0)) {
        delete r;
        return CRH_ERR_HIP;
    }
    r->raster_exclusive = getenv("CRH_RASTER_EXCLUSIVE") != nullptr;

    *out = r;
    return CRH_OK;
}
void crh_renderer_destroy(crh_renderer* r) {
    if (!r) return;
    (void)hipSetDevice(r->device);
    (void)r->sync();
    for (crh_frame* f : r->frames) {
        f->renderer = nullptr;
        f->last_scene = nullptr;
        f->check_pending = false;
    }
    for (crh_scene* sc : r->scenes) sc->renderer = nullptr;
    for (hipEvent_t e : r->event_pool) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(r->stream);
    (void)hipStreamDestroy(r->tess_stream);
    (void)hipStreamDestroy(r->bin_stream);
    (void)hipStreamDestroy(r->aux_stream);
    (void)hipStreamDestroy(r->upload_stream);
    if (r->composite_table) (void)hipFree(r->composite_table);
    delete r;
}
crh_status crh_renderer_get_config(const crh_renderer* r, crh_config* out) {
    if (!r || !out) return CRH_ERR_INVALID_ARGUMENT;
    *out = r->config;
    return CRH_OK;
}
crh_status crh_convert_dynamic_stroke_options(const crh_dynamic_stroke_options* o, crh_dynamic_stroke_descriptor* out) {
    if (!o || !out) return CRH_ERR_INVALID_ARGUMENT;
    return convert_options(*o, *out);
}

crh_status crh_scene_upload(crh_renderer* r, const crh_path_batch* b, crh_scene* existing, crh_scene** out) {
    static const bool phase_timing = std::getenv("CRH_UPLOAD_TIMING") != nullptr; // development: host microseconds of the call's phases on stderr
    const auto t_begin = std::chrono::steady_clock::now();
    auto phase = [&](const char* name) {
        if (phase_timing) std::fprintf(stderr, "[upload] %8.1f us  %s\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count(), name);
    };
    if (!r || !b || !out) return CRH_ERR_INVALID_ARGUMENT;
    uint64_t structure_hash = 0xCBF29CE484222325ull;
    // ---- structure of the batch: a C ABI cannot trust its index arrays (the Rust types make these states unrepresentable)
    {
        // (... and, on the way, a hash of the two index arrays: paths "of the structure the Scene holds" are paths with the same Shapes of the same paths of
        // as many segments each, not just the same totals — ADVICE r05: kept places, batches and capacities were reused on scene-wide counts alone)
        auto prefix_ok = [&](const uint32_t* a, uint32_t n, uint32_t total) {
            if (!a || a[0] != 0u || a[n] != total) return false;
            uint64_t h = structure_hash;
            for (uint32_t i = 0; i < n; ++i) {
                if (a[i] > a[i + 1]) return false;
                h = (h ^ a[i + 1]) * 0x100000001B3ull;
            }
            structure_hash = h;
            return true;
        };
        if (!prefix_ok(b->shape_path_begin, b->n_shapes, b->n_paths) || !prefix_ok(b->path_segment_begin, b->n_paths, b->n_segments)) return CRH_ERR_INVALID_ARGUMENT;
        if (b->shape_dynamic_begin ? !prefix_ok(b->shape_dynamic_begin, b->n_shapes, b->n_dynamic_stroke_options) : b->n_dynamic_stroke_options != 0u) return CRH_ERR_INVALID_ARGUMENT;
        if ((b->n_paths && (!b->path_start || !b->path_stroke_options)) || (b->n_segments && !b->segment_types) || (b->n_control_floats && !b->control_data) ||
            (b->n_stroke_options && !b->stroke_options) || (b->n_dynamic_stroke_options && !b->dynamic_stroke_options))
            return CRH_ERR_INVALID_ARGUMENT;
    }
    // every segment type is valid, and the types stand for as many control floats as the batch says it holds; on the way, where every segment's record
    // begins in control_data (the device builds the element stream from it: launch_build_elements)
    static thread_local std::vector<uint32_t> seg_prefix, seg_path; // (... and whose path every segment is: one coalesced load on the device instead of a search)
    {
        if ((uint64_t)b->n_control_floats >= 0xFFFFFFF0ull) return CRH_ERR_UNSUPPORTED; // (32-bit pool offsets, as below)
        seg_prefix.resize((size_t)b->n_segments + 1u);
        seg_path.resize((size_t)b->n_segments + 1u);
        uint64_t floats = 0;
        uint32_t wrong = 0;
        for (uint32_t p = 0; p < b->n_paths; ++p)
            for (uint32_t g = b->path_segment_begin[p], g1 = b->path_segment_begin[p + 1]; g < g1; ++g) {
                const uint8_t t = b->segment_types[g];
                wrong |= (uint32_t)(t > 4);
                seg_prefix[g] = (uint32_t)floats;
                seg_path[g] = p;
                floats += (uint64_t)kSegmentFloats[t > 4 ? 0 : t];
            }
        seg_prefix[b->n_segments] = (uint32_t)floats;
        if (wrong != 0u || floats != b->n_control_floats) return CRH_ERR_INVALID_ARGUMENT;
    }
    // ---- validation: what the reference rejects with Err(..) before any arithmetic (renderer.rs:188-191, :210-215)
    std::vector<crh_dynamic_stroke_descriptor> descriptors(b->n_dynamic_stroke_options);
    for (uint32_t i = 0; i < b->n_dynamic_stroke_options; ++i) {
        const crh_status st = convert_options(b->dynamic_stroke_options[i], descriptors[i]);
        if (st != CRH_OK) return st;
    }
    bool has_stroke = false;
    for (uint32_t s = 0; s < b->n_shapes; ++s) {
        const uint32_t n_dyn = b->shape_dynamic_begin ? b->shape_dynamic_begin[s + 1] - b->shape_dynamic_begin[s] : 0u;
        for (uint32_t p = b->shape_path_begin[s]; p < b->shape_path_begin[s + 1]; ++p) {
            const int32_t so = b->path_stroke_options[p];
            if (so < 0) continue;
            if ((uint32_t)so >= b->n_stroke_options) return CRH_ERR_INVALID_ARGUMENT;
            has_stroke = true;
            const crh_stroke_options& o = b->stroke_options[so];
            if (o.dynamic_stroke_options_group >= n_dyn) return CRH_ERR_DYNAMIC_STROKE_OPTIONS_INDEX_OUT_OF_BOUNDS;
            if (!std::isfinite(o.width) || !std::isfinite(o.offset) || !std::isfinite(o.miter_clip) || !std::isfinite(o.angle_step)) return CRH_ERR_NON_FINITE;
            // CurveApproximation (path.rs:153-167) is an enum; its payload sizes the vertex streams (2 x parameters per curve segment, counted
            // in 32 bits on the device): a zero / negative angle step or a huge step count would make the counting pass wrap
            if (o.curve_approximation > CRH_CURVE_UNIFORM_TANGENT_ANGLE || o.closed > 1u) return CRH_ERR_INVALID_ARGUMENT;
            if (o.curve_approximation == CRH_CURVE_UNIFORM_TANGENT_ANGLE ? !(o.angle_step >= 1.0e-4f) : o.steps > (1u << 20)) return CRH_ERR_INVALID_ARGUMENT;
        }
    }
    if ((uint64_t)b->n_segments + 2ull * b->n_paths >= 0xFFFFFFF0ull || (uint64_t)b->n_control_floats + 2ull * b->n_paths >= 0xFFFFFFF0ull) return CRH_ERR_UNSUPPORTED; // 32-bit element / pool offsets
    const uint32_t n_elems = b->n_segments + 2u * b->n_paths;
    HIP_TRY(hipSetDevice(r->device));
    if (existing) {
        phase("validated");
        const crh_status st = settle_frames_of(existing, false);
        if (st != CRH_OK) return st;
        phase("frames settled");
    }
    crh_scene* sc = existing ? existing : new crh_scene;
    // ---- the one-pass tessellation's runs ... (below) and: are these paths of the structure the Scene already holds?
    // ---- element stream: MOVE, segments..., END per path; pool = start point + records, -0 canonicalised (safe_float.rs:44-52) — built straight
    //      into pinned staging memory (one arena for all thirteen arrays; the copies to the device are asynchronous, the tessellation waits
    //      for them through an event). Nothing of the Scene is touched before the geometry is known to be finite.
    const size_t n_pool = (size_t)b->n_control_floats + 2u * (size_t)b->n_paths;
    struct Part { size_t at, bytes; };
    // The one-pass tessellation's runs: consecutive Shapes with at most kTessBlock elements between them, one workgroup each; a Shape with more
    // elements than that sends the Scene down the two-pass path (n_runs = 0).
    std::vector<uint32_t> runs;
    uint32_t run_block = (uint32_t)kTessBlock;
    {
        const bool two_pass_only = std::getenv("CRH_TESS_TWO_PASS") != nullptr; // (A/B runs and the tests of that path; read per upload)
        auto cut = [&](uint32_t block) {
            runs.clear();
            bool fits = !two_pass_only && b->n_shapes != 0u;
            uint32_t in_run = 0;
            if (fits) runs.push_back(0u);
            for (uint32_t s = 0; s < b->n_shapes && fits; ++s) {
                const uint32_t p0 = b->shape_path_begin[s], p1 = b->shape_path_begin[s + 1];
                const uint64_t elems = (uint64_t)(b->path_segment_begin[p1] - b->path_segment_begin[p0]) + 2ull * (p1 - p0);
                if (elems > (uint64_t)block) fits = false;
                else if (in_run + elems > (uint64_t)block) runs.push_back(s), in_run = (uint32_t)elems;
                else in_run += (uint32_t)elems;
            }
            if (fits) runs.push_back(b->n_shapes);
            else runs.clear();
            return fits;
        };
        // Workgroups of 128 lanes for scenes of MANY small Shapes: beyond four rounds of resident workgroups (4 096 runs) the two-wave
        // workgroups find their slots in the gap between two raster kernels sooner (50 000 glyphs: 6 300 runs, pipelined step 0.70 -> 0.64 ms;
        // the metric's 266 runs and config 4's 2 660 are no faster that way: tools/r05b_variants.sh). CRH_TESS_RUN_BLOCK pins it.
        const char* pinned = std::getenv("CRH_TESS_RUN_BLOCK");
        if (cut(run_block) && kTessBlock == 256 && (pinned ? std::atoi(pinned) == 128 : runs.size() > 4096u)) {
            const std::vector<uint32_t> wide = runs;
            if (cut(128u)) run_block = 128u;
            else runs = wide;
        }
    }
    // The Scene's arena on the device: what the tessellation reads (built there: the element stream and the pool, launch_build_elements), scratch of that
    // build, then — one contiguous range, staged on the host and copied in one piece — the arrays the host fills and the batch as the caller handed it over.
    enum { P_TYPE, P_OFF, P_PREV, P_PATH, P_POOL, P_PATH_BEGIN, P_PATH_SHAPE, P_SHAPE_BEGIN,                                                              // built on the device
           P_PATH_STROKE, P_DYN_BEGIN, P_OPTIONS, P_DESCRIPTORS, P_RUNS, R_CONTROL, R_START, R_TYPES, R_PATH_SEG, R_SHAPE_PATH, R_SEG_PREFIX, R_SEG_PATH, N_PARTS, P_FIRST_STAGED = P_PATH_STROKE }; // staged by the host
    Part part[N_PARTS];
    size_t arena_bytes = 0;
    {
        const size_t sizes[N_PARTS] = {(size_t)n_elems, (size_t)n_elems * 4, (size_t)n_elems * 4, (size_t)n_elems * 4, n_pool * 4, ((size_t)b->n_paths + 1) * 4, (size_t)b->n_paths * 4, ((size_t)b->n_shapes + 1) * 4,
                                       (size_t)b->n_paths * 4, ((size_t)b->n_shapes + 1) * 4, (size_t)b->n_stroke_options * sizeof(crh_stroke_options),
                                       (size_t)b->n_dynamic_stroke_options * sizeof(crh_dynamic_stroke_descriptor), runs.size() * 4,
                                       (size_t)b->n_control_floats * 4, (size_t)b->n_paths * 8, (size_t)b->n_segments, ((size_t)b->n_paths + 1) * 4, ((size_t)b->n_shapes + 1) * 4, ((size_t)b->n_segments + 1) * 4, (size_t)b->n_segments * 4};
        for (int k = 0; k < N_PARTS; ++k) {
            part[k] = Part{arena_bytes, sizes[k]};
            arena_bytes += (sizes[k] + 255u) & ~(size_t)255u;
        }
    }
    const size_t staged_begin = part[P_FIRST_STAGED].at, staged_bytes = arena_bytes - staged_begin;
    uint8_t* staged = nullptr; // the host's copy of the range [staged_begin, arena_bytes) of the arena
    phase("runs");
    if (!hip_ok(sc->geometry_stage.begin(staged_bytes + 256u, &staged), "hipHostMalloc")) {
        if (!existing) delete sc;
        return CRH_ERR_HIP;
    }
    uint8_t* const arena = staged - staged_begin; // (so that arena + part[k].at addresses the staged parts; the device-built parts have no host copy)
    {
        // SafeFloat::from (safe_float.rs:44-52) over every coordinate of the batch, on the way into the staging memory: finite, and -0 -> +0 (x + 0 is x
        // for every x but -0). One pass over contiguous floats — round 5 went path by path (a memcpy each) and then over the pool a second time.
        auto canonical = [](float* dst, const float* src, size_t n) {
            uint32_t bad = 0;
            for (size_t i = 0; i < n; ++i) {
                uint32_t u;
                std::memcpy(&u, &src[i], 4);
                bad |= (uint32_t)((u & 0x7F800000u) == 0x7F800000u);
                dst[i] = src[i] + 0.0f;
            }
            return bad;
        };
        uint32_t bad = canonical(reinterpret_cast<float*>(arena + part[R_CONTROL].at), b->control_data, b->n_control_floats);
        bad |= canonical(reinterpret_cast<float*>(arena + part[R_START].at), b->path_start, 2u * (size_t)b->n_paths);
        if (bad != 0u) {
            if (!existing) {
                sc->geometry_stage.release();
                delete sc;
            }
            return CRH_ERR_NON_FINITE; // the reference panics in SafeFloat::from (safe_float.rs:46,114)
        }
        if (b->n_segments) std::memcpy(arena + part[R_TYPES].at, b->segment_types, b->n_segments);
        std::memcpy(arena + part[R_PATH_SEG].at, b->path_segment_begin, ((size_t)b->n_paths + 1) * 4);
        std::memcpy(arena + part[R_SHAPE_PATH].at, b->shape_path_begin, ((size_t)b->n_shapes + 1) * 4);
        std::memcpy(arena + part[R_SEG_PREFIX].at, seg_prefix.data(), ((size_t)b->n_segments + 1) * 4);
        if (b->n_segments) std::memcpy(arena + part[R_SEG_PATH].at, seg_path.data(), (size_t)b->n_segments * 4);
        if (b->n_paths) std::memcpy(arena + part[P_PATH_STROKE].at, b->path_stroke_options, (size_t)b->n_paths * 4);
        uint32_t* const dyn_begin = reinterpret_cast<uint32_t*>(arena + part[P_DYN_BEGIN].at);
        if (b->shape_dynamic_begin) std::memcpy(dyn_begin, b->shape_dynamic_begin, ((size_t)b->n_shapes + 1) * 4);
        else std::memset(dyn_begin, 0, ((size_t)b->n_shapes + 1) * 4);
        sc->shape_dyn_begin_host.assign(dyn_begin, dyn_begin + b->n_shapes + 1);
        if (b->n_stroke_options) std::memcpy(arena + part[P_OPTIONS].at, b->stroke_options, part[P_OPTIONS].bytes);
        if (!descriptors.empty()) std::memcpy(arena + part[P_DESCRIPTORS].at, descriptors.data(), part[P_DESCRIPTORS].bytes);
        if (!runs.empty()) std::memcpy(arena + part[P_RUNS].at, runs.data(), part[P_RUNS].bytes);
    }
    phase("element stream built");
    sc->renderer = r;
    sc->device = r->device;
    static std::atomic<uint64_t> next_generation{1}; // unique across scenes (and threads): a new Scene at a recycled address is not mistaken for the old one
    sc->generation = next_generation.fetch_add(1);
    if (!sc->tess_done) {
        if (!hip_ok(hipEventCreateWithFlags(&sc->tess_done, hipEventDisableTiming), "hipEventCreate") ||
            !hip_ok(hipEventCreateWithFlags(&sc->vertices_free, hipEventDisableTiming), "hipEventCreate") ||
            !hip_ok(hipEventCreateWithFlags(&sc->ranges_free, hipEventDisableTiming), "hipEventCreate")) {
            if (!existing) delete sc;
            return CRH_ERR_HIP;
        }
        for (hipEvent_t& e : sc->rec_raster_done)
            if (!hip_ok(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate")) {
                if (!existing) delete sc;
                return CRH_ERR_HIP;
            }
    }
    // New paths of the structure the Scene holds (as many Shapes, paths, elements, the same tessellation path) and sized streams: the capacities
    // stay, no wait for the totals (crh_scene::optimistic). CRH_NO_OPTIMISTIC_UPLOAD: A/B runs and the tests of the other way.
    const uint32_t new_runs = (runs.empty() || n_elems == 0u) ? 0u : (uint32_t)runs.size() - 1u;
    const bool same_structure = existing && sc->capacity_known && sc->d.n_elems == n_elems && sc->d.n_paths == b->n_paths && sc->d.n_shapes == b->n_shapes && sc->has_stroke == has_stroke &&
                                sc->d.n_runs == new_runs && sc->d.run_block == run_block && n_elems != 0u && sc->structure_hash == structure_hash && std::getenv("CRH_NO_OPTIMISTIC_UPLOAD") == nullptr;
    sc->structure_hash = structure_hash;
    sc->rendered_once = false;
    sc->last_render_one_event = false;
    if (sc->shadow.allocated && !same_structure) { // sized for the previous contents
        if (!hip_ok(r->sync(), "sync")) return CRH_ERR_HIP;
        for (DevBuf& buf : sc->shadow.buf) buf.release();
        sc->shadow.allocated = false;
    }
    sc->tessellated_once_before_upload = sc->tessellated_once;
    sc->tessellated_once = false;
    {   // New geometry: measure again which formulation draws it faster — unless it is geometry of the same kind (as many Shapes and segments,
        // to a factor of two): a caller that uploads new paths every frame (the reference's animated-path use) would otherwise never leave
        // the trial, which draws four frames on alternating passes and waits for the GPU on the fifth.
        auto log2_of = [](uint32_t v) { uint32_t n = 0; while (v > 1u) v >>= 1, ++n; return n; };
        const uint32_t geometry = (log2_of(b->n_shapes) << 8) | log2_of(b->n_segments);
        if (geometry != sc->pass_geometry || !existing) {
            sc->pass_choice = 0, sc->pass_frames = 0, sc->pass_class = 0;
            std::memset(sc->pass_known, 0, sizeof(sc->pass_known));
            for (crh_scene::PassTrial& t : sc->pass_trial) t.started = t.recorded = false;
        } else if (sc->pass_choice == 0) { // (a trial under way is started over on the new geometry)
            sc->pass_frames = 0;
            for (crh_scene::PassTrial& t : sc->pass_trial) t.started = t.recorded = false;
        }
        sc->pass_geometry = geometry;
    }
    for (bool& used : sc->rec_used) used = false;
    sc->n_segments = b->n_segments;
    sc->has_stroke = has_stroke;
    sc->capacity_known = same_structure, sc->optimistic = same_structure, sc->counts_valid = false;
    if (same_structure) {
        std::memcpy(sc->totals_host, sc->cap_host, sizeof(sc->totals_host)); // (upper bounds, for what the host sizes from the totals)
        if (sc->shadow.allocated) std::memcpy(sc->shadow.totals_host, sc->shadow.cap_host, sizeof(sc->shadow.totals_host));
    } else {
        sc->lineage = sc->generation;
    }
    if (std::getenv("CRH_NO_LINEAGE")) sc->lineage = sc->generation; // (A/B runs: a frame's list places and batch runs do not outlive an upload)
    if (sc->hull_queued_state == 1) (void)hipEventSynchronize(sc->hull_queued_ready); // (a copy of the old paths' counts still on its way)
    sc->hull_queued_state = 0;
    sc->instances_set = false;
    sc->layout_valid = false;
    // control bytes + 1 type byte per segment, 8 B start per path, 32 B options per stroked path
    uint64_t stroked = 0;
    for (uint32_t p = 0; p < b->n_paths; ++p) stroked += b->path_stroke_options[p] >= 0;
    sc->input_bytes = (uint64_t)b->n_control_floats * 4 + b->n_segments + 8ull * b->n_paths + 32ull * stroked;
    // a Shape whose hull candidates may exceed the small-LDS hull kernel
    sc->big_shapes = false;
    for (uint32_t s = 0; s < b->n_shapes; ++s) {
        const uint32_t p0 = b->shape_path_begin[s], p1 = b->shape_path_begin[s + 1];
        const uint64_t segs = b->path_segment_begin[p1] - b->path_segment_begin[p0];
        if (3 * segs + (p1 - p0) > 64) sc->big_shapes = true; // upper bound of the hull candidates (k_hull_small handles <= 64)
    }
    SceneDev& d = sc->d;
    std::memset(&d, 0, sizeof(d));
    d.n_elems = n_elems;
    d.n_paths = b->n_paths;
    d.n_shapes = b->n_shapes;
    d.n_wg = (n_elems + kTessBlock - 1) / kTessBlock;
    d.n_runs = new_runs;
    d.run_block = run_block;
    for (int c = 0; c < NCH; ++c) d.capacity[c] = same_structure ? sc->cap_host[c] : 0u;
    // ONE asynchronous copy of the whole arena into a device arena of the same layout (round 5; until then fourteen copies into fourteen buffers on the
    // raster stream, in line behind the raster kernel of the frame in front). It runs on the upload stream, behind the last readers of the Scene's old
    // paths only: its tessellation and the raster kernels of the passes that drew it (their binning comes first on the way) — the frames in flight
    // from OTHER Scenes are not waited for, so new paths travel while the frame in front is drawn. No host wait.
    const hipStream_t st = r->upload_stream;
    crh_status rc;
    {
        if (!sc->geometry_ready && !hip_ok(hipEventCreateWithFlags(&sc->geometry_ready, hipEventDisableTiming), "hipEventCreate")) {
            rc = CRH_ERR_HIP;
            goto fail;
        }
        bool ok = true;
        if (existing && sc->tess_done && sc->tessellated_once_before_upload) ok = hip_ok(hipStreamWaitEvent(st, sc->tess_done, 0), "hipStreamWaitEvent"); // a tessellation of the old paths may still run
        if (existing && sc->shadow.tess_done && sc->tessellated_once_before_upload) ok = ok && hip_ok(hipStreamWaitEvent(st, sc->shadow.tess_done, 0), "hipStreamWaitEvent");
        for (int k = 0; k < kPipelineDepth && ok; ++k)
            if (existing && sc->rec_used_ever[k]) ok = hip_ok(hipStreamWaitEvent(st, sc->rec_raster_done[k], 0), "hipStreamWaitEvent");
        // (the element offsets exist twice on the device when a path is stroked: k_stroke_records rewrites elem_off, elem_off0 stays)
        const size_t second_off = has_stroke ? ((part[P_OFF].bytes + 255u) & ~(size_t)255u) : 0u;
        ok = ok && hip_ok(sc->geometry.ensure(arena_bytes + second_off + 256u), "hipMalloc");
        if (ok && staged_bytes) ok = hip_ok(hipMemcpyAsync(static_cast<uint8_t*>(sc->geometry.p) + staged_begin, staged, staged_bytes, hipMemcpyHostToDevice, st), "hipMemcpyAsync");
        if (ok) { // ... and the element stream and the pool are built from it on the device
            uint8_t* const g = static_cast<uint8_t*>(sc->geometry.p);
            UploadBuild u = {};
            u.control = reinterpret_cast<const float*>(g + part[R_CONTROL].at), u.start = reinterpret_cast<const float*>(g + part[R_START].at), u.types = g + part[R_TYPES].at;
            u.path_seg = reinterpret_cast<const uint32_t*>(g + part[R_PATH_SEG].at), u.shape_path = reinterpret_cast<const uint32_t*>(g + part[R_SHAPE_PATH].at);
            u.n_segments = b->n_segments, u.n_paths = b->n_paths, u.n_shapes = b->n_shapes, u.n_elems = n_elems;
            u.seg_prefix = reinterpret_cast<const uint32_t*>(g + part[R_SEG_PREFIX].at), u.seg_path = reinterpret_cast<const uint32_t*>(g + part[R_SEG_PATH].at);
            u.elem_type = g + part[P_TYPE].at, u.elem_off = reinterpret_cast<uint32_t*>(g + part[P_OFF].at), u.elem_off_again = second_off ? reinterpret_cast<uint32_t*>(g + arena_bytes) : nullptr;
            u.elem_prev_off = reinterpret_cast<uint32_t*>(g + part[P_PREV].at), u.elem_path = reinterpret_cast<uint32_t*>(g + part[P_PATH].at), u.pool = reinterpret_cast<float*>(g + part[P_POOL].at);
            u.path_elem_begin = reinterpret_cast<uint32_t*>(g + part[P_PATH_BEGIN].at), u.path_shape = reinterpret_cast<uint32_t*>(g + part[P_PATH_SHAPE].at);
            u.shape_elem_begin = reinterpret_cast<uint32_t*>(g + part[P_SHAPE_BEGIN].at);
            sc->pending_build = u; // (launched by the tessellation that waits for geometry_ready: run_tessellation)
        }
        // the status word: cleared IN FRONT of geometry_ready — the tessellation stream waits for that event only, then clears the word itself and
        // lets its kernels write error codes; a memset enqueued behind the event would be unordered against those writes
        if (ok) ok = hip_ok(sc->status.ensure(4), "hipMalloc") && hip_ok(hipMemsetAsync(sc->status.p, 0xFF, 4, st), "hipMemset");
        if (ok) ok = hip_ok(sc->geometry_stage.done(st), "hipEventRecord") && hip_ok(hipEventRecord(sc->geometry_ready, st), "hipEventRecord");
        if (!ok) {
            rc = CRH_ERR_HIP;
            goto fail;
        }
        sc->geometry_pending = true;
    }
    // (the scan state of the two-pass path — 40 B per element and the rows' totals — exists only for Scenes that take it)
    if ((d.n_runs == 0u && (!hip_ok(sc->elem_scan.ensure((size_t)n_elems * sizeof(ElemScan)), "hipMalloc elem_scan") ||
        !hip_ok(sc->wg_total.ensure((size_t)d.n_wg * NCH * 4), "hipMalloc") || !hip_ok(sc->wg_base.ensure((size_t)d.n_wg * NCH * 4), "hipMalloc") || !hip_ok(sc->group_base.ensure(((size_t)d.n_wg / 64 + 2) * NCH * 4), "hipMalloc"))) ||
        (d.n_runs != 0u && (!hip_ok(sc->elem_cnt.ensure((size_t)n_elems * 4), "hipMalloc") || !hip_ok(sc->run_base.ensure(((size_t)d.n_runs + 1) * NCH * 4), "hipMalloc"))) ||
        (d.n_runs != 0u && has_stroke && !hip_ok(sc->path_scan.ensure((size_t)b->n_paths * 12), "hipMalloc")) ||
        !hip_ok(sc->totals.ensure(NCH * 4), "hipMalloc") || !hip_ok(sc->shape_base.ensure((size_t)b->n_shapes * kShapeRow * 4), "hipMalloc") ||
        !hip_ok(sc->hull_count.ensure((size_t)b->n_shapes * 4), "hipMalloc") || !hip_ok(sc->hull_large.ensure((3 * (size_t)b->n_shapes + 4) * 4), "hipMalloc") || !hip_ok(sc->status.ensure(4), "hipMalloc") ||
        !hip_ok(sc->transforms.ensure((size_t)b->n_shapes * 64), "hipMalloc") || !hip_ok(sc->colors.ensure((size_t)b->n_shapes * 16), "hipMalloc") ||
        !hip_ok(sc->shape_ncand.ensure((size_t)b->n_shapes * 4 + 4), "hipMalloc") || !hip_ok(sc->shape_prim_begin.ensure(((size_t)b->n_shapes + 1) * 4), "hipMalloc") ||
        !hip_ok(sc->shape_nslots.ensure((size_t)b->n_shapes * 4 + 4), "hipMalloc") || !hip_ok(sc->shape_slot_begin.ensure(((size_t)b->n_shapes + 1) * 4), "hipMalloc") ||
        !hip_ok(sc->prim_scan_scratch.ensure(((size_t)(b->n_shapes + 511) / 512 + 2) * 8), "hipMalloc")) { // (two scans side by side, a sum per 512 Shapes each)
        rc = CRH_ERR_HIP;
        goto fail;
    }
    {
        uint8_t* const g = static_cast<uint8_t*>(sc->geometry.p);
        auto at = [&](int k) { return g + part[k].at; };
        d.elem_type = at(P_TYPE);
        d.elem_off0 = reinterpret_cast<const uint32_t*>(at(P_OFF));
        d.elem_off = has_stroke ? reinterpret_cast<uint32_t*>(g + arena_bytes) : reinterpret_cast<uint32_t*>(at(P_OFF)); // (only k_stroke_records writes it)
        d.elem_prev_off = reinterpret_cast<uint32_t*>(at(P_PREV));
        d.elem_path = reinterpret_cast<const uint32_t*>(at(P_PATH));
        d.pool = reinterpret_cast<const float*>(at(P_POOL));
        d.path_elem_begin = reinterpret_cast<const uint32_t*>(at(P_PATH_BEGIN));
        d.path_shape = reinterpret_cast<const uint32_t*>(at(P_PATH_SHAPE));
        d.path_stroke = reinterpret_cast<const int32_t*>(at(P_PATH_STROKE));
        d.shape_elem_begin = reinterpret_cast<const uint32_t*>(at(P_SHAPE_BEGIN));
        d.shape_dyn_begin = reinterpret_cast<const uint32_t*>(at(P_DYN_BEGIN));
        d.stroke_options = reinterpret_cast<const crh_stroke_options*>(at(P_OPTIONS));
        d.descriptors = reinterpret_cast<crh_dynamic_stroke_descriptor*>(at(P_DESCRIPTORS));
        d.tess_run = reinterpret_cast<const uint32_t*>(at(P_RUNS));
    }
    d.elem_cnt = sc->elem_cnt.as<uint32_t>(), d.run_base = sc->run_base.as<uint32_t>();
    bind_tess_pointers(sc);
    if (!existing) r->scenes.push_back(sc);
    phase("enqueued");
    *out = sc;
    return CRH_OK;
fail:
    if (!existing) {
        sc->release_all();
        delete sc;
    }
    return rc;
}

crh_status crh_scene_tessellate(crh_scene* sc) {
    if (!sc) return CRH_ERR_INVALID_ARGUMENT;
    return run_tessellation(sc, false);
}
crh_status crh_scene_status(crh_scene* sc) {
    if (!sc) return CRH_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(sc->renderer->device));
    uint32_t word = 0xFFFFFFFFu;
    crh_status st = settle_tessellation(sc, &word);
    if (st != CRH_OK) return st;
    return decode_status(sc, word);
}
void crh_scene_destroy(crh_scene* sc) {
    if (!sc) return;
    (void)hipSetDevice(sc->device);
    if (sc->renderer) {
        (void)settle_frames_of(sc, true);
        (void)sc->renderer->sync();
        for (hipEvent_t& seen : sc->renderer->raster_events) // (the events below are about to be destroyed)
            for (hipEvent_t mine : sc->rec_raster_done)
                if (seen == mine) seen = nullptr;
        std::vector<crh_scene*>& live = sc->renderer->scenes;
        for (size_t i = 0; i < live.size(); ++i)
            if (live[i] == sc) {
                live.erase(live.begin() + (long)i);
                break;
            }
    }
    sc->release_all();
    if (sc->geometry_ready) (void)hipEventSynchronize(sc->geometry_ready);
    for (hipEvent_t e : {sc->tess_done, sc->vertices_free, sc->ranges_free, sc->shadow.tess_done, sc->shadow.vertices_free, sc->shadow.ranges_free, sc->geometry_ready})
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : sc->rec_raster_done)
        if (e) (void)hipEventDestroy(e);
    for (crh_scene::PassTrial& t : sc->pass_trial)
        for (hipEvent_t e : t.e)
            if (e) (void)hipEventDestroy(e);
    delete sc;
}
crh_status crh_shape_from_paths(crh_renderer* r, const crh_path_batch* one_shape, crh_scene* existing, crh_scene** out) {
    if (!one_shape || one_shape->n_shapes != 1) return CRH_ERR_INVALID_ARGUMENT;
    crh_status st = crh_scene_upload(r, one_shape, existing, out);
    if (st != CRH_OK) return st;
    st = crh_scene_tessellate(*out);
    if (st != CRH_OK) return st;
    return crh_scene_status(*out); // from_paths is synchronous in the reference and returns its Err here
}

crh_status crh_scene_shape_layout(crh_scene* sc, uint32_t shape, uint64_t vo[8], uint64_t io[3]) {
    if (!sc || shape >= sc->d.n_shapes) return CRH_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(sc->renderer->device));
    crh_status st = fetch_layout(sc);
    if (st != CRH_OK) return st;
    shape_layout(sc, shape, vo, io);
    return CRH_OK;
}
crh_status crh_scene_shape_download(crh_scene* sc, uint32_t shape, void* vb, void* ib) {
    if (!sc || shape >= sc->d.n_shapes) return CRH_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(sc->renderer->device));
    crh_status st = fetch_layout(sc);
    if (st != CRH_OK) return st;
    HostCopy h;
    if ((st = fetch_outputs(sc, h)) != CRH_OK) return st;
    assemble_shape(sc, h, shape, static_cast<uint8_t*>(vb), static_cast<uint8_t*>(ib));
    return CRH_OK;
}
crh_status crh_scene_layout_all(crh_scene* sc, uint64_t* layout, uint64_t* total_v, uint64_t* total_i) {
    if (!sc) return CRH_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(sc->renderer->device));
    crh_status st = fetch_layout(sc);
    if (st != CRH_OK) return st;
    uint64_t tv = 0, ti = 0;
    for (uint32_t s = 0; s < sc->d.n_shapes; ++s) {
        uint64_t vo[8], io[3];
        shape_layout(sc, s, vo, io);
        if (layout) {
            for (int k = 0; k < 8; ++k) layout[(size_t)s * 11 + k] = vo[k];
            for (int k = 0; k < 3; ++k) layout[(size_t)s * 11 + 8 + k] = io[k];
        }
        tv += vo[7];
        ti += io[2];
    }
    if (total_v) *total_v = tv;
    if (total_i) *total_i = ti;
    return CRH_OK;
}
crh_status crh_scene_download_all(crh_scene* sc, void* vb, void* ib) {
    if (!sc) return CRH_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(sc->renderer->device));
    crh_status st = fetch_layout(sc);
    if (st != CRH_OK) return st;
    HostCopy h;
    if ((st = fetch_outputs(sc, h)) != CRH_OK) return st;
    uint8_t* v = static_cast<uint8_t*>(vb);
    uint8_t* i = static_cast<uint8_t*>(ib);
    for (uint32_t s = 0; s < sc->d.n_shapes; ++s) {
        uint64_t vo[8], io[3];
        shape_layout(sc, s, vo, io);
        assemble_shape(sc, h, s, v, i);
        if (v) v += vo[7];
        if (i) i += io[2];
    }
    return CRH_OK;
}
crh_status crh_scene_traffic(crh_scene* sc, uint64_t* bytes_read, uint64_t* bytes_written) {
    if (!sc) return CRH_ERR_INVALID_ARGUMENT;
    if (bytes_read) *bytes_read = sc->input_bytes;
    if (bytes_written) *bytes_written = sc->emitted_bytes;
    return CRH_OK;
}
crh_status crh_scene_set_dynamic_stroke_options(crh_scene* sc, uint32_t shape, uint32_t group, const crh_dynamic_stroke_options* o) {
    if (!sc || !o || shape >= sc->d.n_shapes) return CRH_ERR_INVALID_ARGUMENT;
    const uint32_t begin = sc->shape_dyn_begin_host[shape], end = sc->shape_dyn_begin_host[shape + 1];
    if (group >= end - begin) return CRH_ERR_DYNAMIC_STROKE_OPTIONS_INDEX_OUT_OF_BOUNDS; // renderer.rs:366-368
    crh_dynamic_stroke_descriptor d;
    const crh_status st = convert_options(*o, d);
    if (st != CRH_OK) return st;
    HIP_TRY(hipSetDevice(sc->renderer->device));
    {
        const crh_status pending = settle_frames_of(sc, false);
        if (pending != CRH_OK) return pending;
    }
    if (sc->geometry_ready) HIP_TRY(hipEventSynchronize(sc->geometry_ready)); // (the arena of the last upload — the descriptors are part of it — may still be on its way)
    HIP_TRY(hipMemcpyAsync(sc->d.descriptors + begin + group, &d, sizeof(d), hipMemcpyHostToDevice, sc->renderer->stream));
    HIP_TRY(sc->renderer->sync());
    return CRH_OK;
}

crh_status crh_frame_create(crh_renderer* r, uint32_t width, uint32_t height, crh_frame** out) { return crh_frame_create_format(r, width, height, CRH_FORMAT_RGBA8, out); }
crh_status crh_frame_format(const crh_frame* f, uint32_t* format) {
    if (!f || !format) return CRH_ERR_INVALID_ARGUMENT;
    *format = f->format;
    return CRH_OK;
}
crh_status crh_frame_create_format(crh_renderer* r, uint32_t width, uint32_t height, uint32_t format, crh_frame** out) {
    // pixel boxes are 16-bit (0xFFFF = nothing to draw), so a frame is at most 65 535 pixels wide and high
    if (!r || !out || width == 0 || height == 0 || width > 65535u || height > 65535u || format > CRH_FORMAT_RGBA8_ATTACHMENT) return CRH_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(r->device));
    crh_frame* f = new crh_frame;
    f->renderer = r;
    f->format = format;
    f->device = r->device;
    f->width = width;
    f->height = height;
    f->tiles_x = (width + 15) / 16;
    f->tiles_y = (height + 15) / 16;
    f->n_tiles = f->tiles_x * f->tiles_y;
    bool ok = hip_ok(f->rgba8.ensure(f->image_bytes()), "hipMalloc frame") && hip_ok(hipEventCreateWithFlags(&f->ext_read, hipEventDisableTiming), "hipEventCreate") &&
              hip_ok(hipEventCreateWithFlags(&f->ext_write, hipEventDisableTiming), "hipEventCreate");
    for (crh_frame::BinSet& set : f->sets)
        ok = ok && hip_ok(set.tile_count_cursor.ensure((size_t)f->n_tiles * 8 + 512), "hipMalloc") && hip_ok(set.tile_offset.ensure((size_t)(f->n_tiles + 1) * 4), "hipMalloc") &&
             hip_ok(set.tile_list.ensure(f->pair_capacity_bytes), "hipMalloc") &&
             hip_ok(hipEventCreateWithFlags(&set.bin_done, hipEventDisableTiming), "hipEventCreate") &&
             hip_ok(hipEventCreateWithFlags(&set.raster_done, hipEventDisableTiming), "hipEventCreate") &&
             hip_ok(hipEventCreateWithFlags(&set.flags_ready, hipEventDisableTiming), "hipEventCreate") &&
             hip_ok(hipHostMalloc(reinterpret_cast<void**>(&set.flags_host), sizeof(uint32_t) * 128, hipHostMallocDefault), "hipHostMalloc");
    if (!ok) {
        crh_frame_destroy(f);
        return CRH_ERR_HIP;
    }
    HIP_TRY(hipMemsetAsync(f->rgba8.p, 0, f->image_bytes(), r->stream));
    if (r->config.depth_compare != CRH_COMPARE_ALWAYS || r->config.depth_write_enabled) { // the depth attachment, cleared to 1.0 (main.rs:223-226)
        const size_t n = (size_t)width * height * r->config.msaa_sample_count;
        HIP_TRY(f->depth.ensure(n * 4));
        HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(f->depth.p), 0x3f800000, n, r->stream));
    }
    for (crh_frame::BinSet& set : f->sets) {
        set.overflow_p = static_cast<uint8_t*>(set.tile_count_cursor.p) + (size_t)f->n_tiles * 8;
        HIP_TRY(hipMemsetAsync(set.overflow_p, 0, 512, r->stream));
    }
    HIP_TRY(r->sync());
    r->frames.push_back(f);
    *out = f;
    return CRH_OK;
}
void crh_frame_destroy(crh_frame* f) {
    if (!f) return;
    (void)hipSetDevice(f->device);
    if (f->ext_read_set) (void)hipEventSynchronize(f->ext_read);
    if (f->ext_write_set) (void)hipEventSynchronize(f->ext_write);
    if (f->renderer) {
        (void)f->renderer->sync();
        for (size_t i = 0; i < f->renderer->frames.size(); ++i)
            if (f->renderer->frames[i] == f) {
                f->renderer->frames.erase(f->renderer->frames.begin() + (long)i);
                break;
            }
    }
    DevBuf* all[] = {&f->tile_order, &f->item_cost, &f->bin_batches, &f->tile_base, &f->tile_base_b, &f->tile_caps, &f->rgba8, &f->depth, &f->items, &f->item_transforms, &f->item_colors, &f->item_transforms_b, &f->item_colors_b, &f->item_ncand, &f->item_prim_begin, &f->item_scan_scratch,
                     &f->item_nslots, &f->item_slot_begin, &f->state_stencil, &f->state_alpha, &f->state_color};
    for (DevBuf* b : all) b->release();
    f->item_upload_t.release();
    f->item_upload_c.release();
    for (InstanceSlot& k : f->item_slot) k.release();
    for (crh_frame::BinSet& set : f->sets) {
        DevBuf* bins[] = {&set.tile_count_cursor, &set.tile_offset, &set.tile_list, &set.scan_scratch, &set.pair_tile, &set.pair_pos, &set.pair_key, &set.bin_queue, &set.item_elsewhere};
        for (DevBuf* b : bins) b->release();
        if (set.bin_done) (void)hipEventDestroy(set.bin_done);
        if (set.raster_done) (void)hipEventDestroy(set.raster_done);
        if (set.flags_ready) (void)hipEventDestroy(set.flags_ready);
        if (set.flags_host) (void)hipHostFree(set.flags_host), set.flags_host = nullptr;
    }
    if (f->ext_read) { // an exchange may still be reading or writing the pixels on its own stream
        if (f->ext_read_set) (void)hipEventSynchronize(f->ext_read);
        (void)hipEventDestroy(f->ext_read);
    }
    if (f->ext_write) {
        if (f->ext_write_set) (void)hipEventSynchronize(f->ext_write);
        (void)hipEventDestroy(f->ext_write);
    }
    delete f;
}
crh_status crh_frame_clear(crh_frame* f) {
    if (!f) return CRH_ERR_INVALID_ARGUMENT;
    f->cleared = true; // LoadOp::Clear: the next render does not read the target, every tile is written
    f->carry = f->carry_valid = f->carry_recolor = false; // ... and the stencil attachment and the alpha layers start from zero (main.rs:217-230)
    return f->depth.p ? crh_frame_clear_depth(f, 1.0f) : CRH_OK;
}
crh_status crh_frame_keep_pass_state(crh_frame* f) {
    if (!f) return CRH_ERR_INVALID_ARGUMENT;
    f->carry = true; // (the planes are allocated and initialised — from the image, if the frame shows one — in front of the next pass: render_impl)
    return CRH_OK;
}
crh_status crh_frame_clear_depth(crh_frame* f, float value) {
    if (!f || !f->depth.p || !std::isfinite(value)) return CRH_ERR_INVALID_ARGUMENT;
    crh_renderer* r = f->renderer;
    HIP_TRY(hipSetDevice(r->device));
    uint32_t bits;
    memcpy(&bits, &value, 4);
    // on the raster stream: ordered after the raster kernel of the previous frame, before the next one
    HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(f->depth.p), (int)bits, (size_t)f->width * f->height * r->config.msaa_sample_count, r->stream));
    return CRH_OK;
}
crh_status crh_frame_upload_depth(crh_frame* f, const float* depth) {
    if (!f || !f->depth.p || !depth) return CRH_ERR_INVALID_ARGUMENT;
    crh_renderer* r = f->renderer;
    HIP_TRY(hipSetDevice(r->device));
    const uint32_t samples = r->config.msaa_sample_count;
    const size_t n = (size_t)f->width * f->height;
    std::vector<float> expanded(n * samples);
    for (size_t i = 0; i < n; ++i) {
        if (!std::isfinite(depth[i])) return CRH_ERR_NON_FINITE;
        for (uint32_t k = 0; k < samples; ++k) expanded[i * samples + k] = depth[i];
    }
    HIP_TRY(hipMemcpyAsync(f->depth.p, expanded.data(), expanded.size() * 4, hipMemcpyHostToDevice, r->stream));
    HIP_TRY(r->sync());
    return CRH_OK;
}
crh_status crh_frame_download_depth(crh_frame* f, float* depth_samples) {
    if (!f || !f->depth.p || !depth_samples) return CRH_ERR_INVALID_ARGUMENT;
    crh_renderer* r = f->renderer;
    HIP_TRY(hipSetDevice(r->device));
    crh_status st = settle_frame(f);
    if (st != CRH_OK) return st;
    HIP_TRY(hipMemcpyAsync(depth_samples, f->depth.p, (size_t)f->width * f->height * r->config.msaa_sample_count * 4, hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(r->sync());
    return CRH_OK;
}

crh_status crh_scene_set_instances(crh_scene* sc, const float* transforms, const float* colors) {
    if (!sc || !transforms || !colors) return CRH_ERR_INVALID_ARGUMENT;
    crh_renderer* r = sc->renderer;
    HIP_TRY(hipSetDevice(r->device));
    for (size_t i = 0; i < (size_t)sc->d.n_shapes * 16; ++i)
        if (!std::isfinite(transforms[i])) return CRH_ERR_NON_FINITE;
    for (size_t i = 0; i < (size_t)sc->d.n_shapes * 4; ++i)
        if (!std::isfinite(colors[i])) return CRH_ERR_NON_FINITE; // Color = SafeFloat<f32, 4> (renderer.rs:16)
    const int next = sc->instances_set ? sc->instances_cur ^ 1 : sc->instances_cur;
    for (crh_frame* f : r->frames) // a frame that would be drawn again from the buffer about to be overwritten (two updates old: long done)
        if (f->last_scene == sc && f->check_pending && f->n_items == 0 && f->last_instances == next) {
            const crh_status st = settle_frame_cheaply(f);
            if (st != CRH_OK) return st;
        }
    DevBuf& tb = next ? sc->transforms_b : sc->transforms;
    DevBuf& cb = next ? sc->colors_b : sc->colors;
    if (sc->d.n_shapes) {
        HIP_TRY(tb.ensure((size_t)sc->d.n_shapes * 64));
        HIP_TRY(cb.ensure((size_t)sc->d.n_shapes * 16));
        // On the BINNING stream, in order behind the binning of the frame before — the copies run while that frame's raster kernel does — and in
        // front of the binning that reads them; no host synchronisation (the host arrays are copied to pinned staging memory before the calls
        // return). Round 5: on a stream of their own (rounds 2 - 4) the copies were as early, but the binning stream's wait for the copy
        // engine's event cost a consumed animation 0.05 ms per frame (tools/r05_host_calls.py: 0.357 against 0.308 ms per step of 10 000 paths).
        InstanceSlot& slot = sc->slot[next];
        HIP_TRY(slot.init());
        const hipStream_t up = r->pipeline ? r->bin_stream : r->stream;
        if (slot.was_read) HIP_TRY(hipStreamWaitEvent(up, slot.read_event, 0)); // the setup that read this set two updates ago
        HIP_TRY(sc->upload_t.copy(tb.p, transforms, (size_t)sc->d.n_shapes * 64, up));
        HIP_TRY(sc->upload_c.copy(cb.p, colors, (size_t)sc->d.n_shapes * 16, up));
        HIP_TRY(hipEventRecord(slot.ready, up));
        slot.was_written = true;
    }
    sc->instances_cur = next;
    sc->instances_version += 1u;
    sc->instances_projective_of[next] = !all_instances_plain(transforms, sc->d.n_shapes);
    sc->instances_tame_of[next] = all_colors_tame(colors, sc->d.n_shapes);
    sc->instances_projective = sc->instances_projective_of[next];
    sc->instances_set = true;
    return CRH_OK;
}
crh_status crh_scene_render_resident(crh_scene* sc, crh_frame* f) {
    if (!sc || !f || f->renderer != sc->renderer) return CRH_ERR_INVALID_ARGUMENT;
    if (f->check_pending && !f->cleared) { // what the frame shows has to be final before the pass that produced it is forgotten
        HIP_TRY(hipSetDevice(sc->renderer->device));
        const crh_status st = settle_frame(f);
        if (st != CRH_OK) return st;
    }
    f->n_items = 0; // the plain pass: Stencil + Color of every Shape
    return render_impl(sc, f);
}
namespace {
// Does a recorded pass leave state in the stencil attachment or the alpha layers when it ends — state the reference would hand to the next
// Shape::render call into the same attachments (renderer.rs:148-158, 257-266)? Decided on the host from the draws alone, conservatively:
//   Stencil leaves winding counters until a Color or Clip cover of the same Shape and instance resets them (renderer.rs:747-752 pass / fail -> Zero;
//   :703-708 Replace(ref), whose winding bits are zero) — the alpha-context covers write no stencil (write_mask 0, renderer.rs:761-766);
//   Clip leaves a nesting level until the UnClip of the same Shape and instance (renderer.rs:722-727); SaveAlphaContext leaves a layer until
//   its RestoreAlphaContext. Anything unmatched, or matched out of order, counts as left over.
bool pass_leaves_state(const crh_draw* draws, uint32_t n_draws) {
    std::vector<std::pair<uint32_t, uint32_t>> stencils, clips;
    uint32_t open_layers = 0;
    for (uint32_t i = 0; i < n_draws; ++i) {
        const crh_draw& d = draws[i];
        const std::pair<uint32_t, uint32_t> who(d.shape, d.instance);
        switch (d.op) {
            case CRH_OP_STENCIL:
                if (std::find(stencils.begin(), stencils.end(), who) == stencils.end()) stencils.push_back(who);
                break;
            case CRH_OP_COLOR:
                stencils.erase(std::remove(stencils.begin(), stencils.end(), who), stencils.end());
                break;
            case CRH_OP_CLIP:
                stencils.erase(std::remove(stencils.begin(), stencils.end(), who), stencils.end());
                clips.push_back(who);
                break;
            case CRH_OP_UNCLIP:
                if (clips.empty() || clips.back() != who) return true; // closes a level this pass did not open (or not in this order)
                clips.pop_back();
                break;
            case CRH_OP_SAVE_ALPHA_CONTEXT: open_layers |= 1u << (d.alpha_layer & 31u); break;
            case CRH_OP_RESTORE_ALPHA_CONTEXT:
                if (!(open_layers >> (d.alpha_layer & 31u) & 1u)) return true; // restores a context an earlier pass saved
                open_layers &= ~(1u << (d.alpha_layer & 31u));
                break;
            default: break; // ScaleAlphaContext reads and writes the colour attachment only
        }
        if (d.clip_depth > clips.size()) return true; // drawn at a clip level an earlier pass opened: the frame holds state
    }
    return !stencils.empty() || !clips.empty() || open_layers != 0u;
}
} // namespace
// tests only (host code, no device): does a recorded pass leave state with the frame?
extern "C" int crh_debug_pass_leaves_state(const crh_draw* draws, uint32_t n_draws) { return (draws || n_draws == 0u) ? (pass_leaves_state(draws, n_draws) ? 1 : 0) : -1; }
crh_status crh_scene_render_draws(crh_scene* sc, crh_frame* f, const float* transforms, const float* colors, uint32_t n_instances, const crh_draw* draws,
                                  uint32_t n_draws) {
    if (!sc || !f || f->renderer != sc->renderer || (n_instances && (!transforms || !colors)) || (n_draws && !draws)) return CRH_ERR_INVALID_ARGUMENT;
    crh_renderer* r = sc->renderer;
    HIP_TRY(hipSetDevice(r->device));
    if (r->config.alpha_layer_count > 4) return CRH_ERR_UNSUPPORTED;
    for (size_t i = 0; i < (size_t)n_instances * 16; ++i)
        if (!std::isfinite(transforms[i])) return CRH_ERR_NON_FINITE;
    for (size_t i = 0; i < (size_t)n_instances * 4; ++i)
        if (!std::isfinite(colors[i])) return CRH_ERR_NON_FINITE;
    // validation in recording order, as the reference's calls would fail (renderer.rs:933-935, :947-949, :980-982)
    std::vector<DrawItem> items;
    bool need_ops = false;
    for (uint32_t i = 0; i < n_draws; ++i) {
        const crh_draw& d = draws[i];
        need_ops = need_ops || !(d.op == CRH_OP_STENCIL || d.op == CRH_OP_COLOR) || d.clip_depth != 0u;
        if (d.shape >= sc->d.n_shapes || d.instance >= n_instances || d.op > CRH_OP_RESTORE_ALPHA_CONTEXT) return CRH_ERR_INVALID_ARGUMENT;
        if (d.clip_depth >= (1u << r->config.clip_nesting_counter_bits)) return CRH_ERR_CLIP_STACK_OVERFLOW;
        if (d.op >= CRH_OP_SAVE_ALPHA_CONTEXT && d.alpha_layer >= r->config.alpha_layer_count) return CRH_ERR_TOO_MANY_NESTED_OPACITY_GROUPS;
        if (d.op == CRH_OP_STENCIL) {
            items.push_back(DrawItem{d.shape, d.instance, 1u, d.clip_depth});
        } else {
            const uint32_t cover = ((uint32_t)d.op + 1u) << 4, refs = (d.clip_depth << 8) | ((d.alpha_layer & 15u) << 16);
            // Stencil immediately followed by a cover operation of the same Shape and instance is one item (one walk of its tiles)
            if (!items.empty() && items.back().ops == 1u && items.back().shape == d.shape && items.back().instance == d.instance) {
                items.back().ops |= cover;
                items.back().refs |= refs;
            } else {
                items.push_back(DrawItem{d.shape, d.instance, cover, refs});
            }
        }
    }
    if (items.empty()) { // an empty pass still resolves a cleared frame
        f->n_items = 0;
        if (f->cleared) {
            HIP_TRY(order_after_external(f, r->stream));
            HIP_TRY(hipMemsetAsync(f->rgba8.p, 0, f->image_bytes(), r->stream));
        }
        f->cleared = false;
        return CRH_OK;
    }
    f->items_need_ops = need_ops;
    if (pass_leaves_state(draws, n_draws)) f->carry = true; // from here until crh_frame_clear the frame keeps clip / winding counters, saved alphas and sample colours in HBM
    const bool same_pass = f->items_ranges_valid && f->items_scene == sc && f->items_generation == sc->generation && f->n_items == items.size() &&
                           f->items_host.size() == items.size() && memcmp(f->items_host.data(), items.data(), items.size() * sizeof(DrawItem)) == 0 &&
                           f->item_transforms.cap >= (size_t)n_instances * 64 && f->item_colors.cap >= (size_t)n_instances * 16;
    if (same_pass) {
        // the same items over the same geometry (an animation: only the instance data moves): the new instance data goes to the set the
        // previous submission did not read — its deferred remedy may still want that one — on the binning stream, where k_prim_setup,
        // the only reader, runs: ordered behind the setup that read this set two submissions ago, no host synchronisation
        const int next = f->item_inst_cur ^ 1;
        DevBuf& tb = next ? f->item_transforms_b : f->item_transforms;
        DevBuf& cb = next ? f->item_colors_b : f->item_colors;
        HIP_TRY(tb.ensure((size_t)n_instances * 64 + 64));
        HIP_TRY(cb.ensure((size_t)n_instances * 16 + 16));
        InstanceSlot& slot = f->item_slot[next];
        HIP_TRY(slot.init());
        const hipStream_t up = r->pipeline ? r->bin_stream : r->stream;
        if (slot.was_read) HIP_TRY(hipStreamWaitEvent(up, slot.read_event, 0));
        HIP_TRY(f->item_upload_t.copy(tb.p, transforms, (size_t)n_instances * 64, up));
        HIP_TRY(f->item_upload_c.copy(cb.p, colors, (size_t)n_instances * 16, up));
        HIP_TRY(hipEventRecord(slot.ready, up));
        slot.was_written = true;
        f->item_inst_cur = next;
        f->item_projective_of[next] = all_instances_plain(transforms, n_instances) ? 0 : 1;
        f->item_tame_of[next] = all_colors_tame(colors, n_instances) ? 1 : 0;
        return render_impl(sc, f);
    }
    if (f->check_pending) { // the frame's recorded pass (its remedy for an overflowed tile list) is about to be replaced
        const crh_status st = settle_frame(f);
        if (st != CRH_OK) return st;
    }
    HIP_TRY(r->sync()); // the buffers below may still be read by a frame in flight
    f->item_inst_cur = 0;
    HIP_TRY(f->items.ensure(items.size() * sizeof(DrawItem)));
    HIP_TRY(f->item_transforms.ensure((size_t)n_instances * 64 + 64));
    HIP_TRY(f->item_colors.ensure((size_t)n_instances * 16 + 16));
    HIP_TRY(hipMemcpyAsync(f->items.p, items.data(), items.size() * sizeof(DrawItem), hipMemcpyHostToDevice, r->stream));
    HIP_TRY(hipMemcpyAsync(f->item_transforms.p, transforms, (size_t)n_instances * 64, hipMemcpyHostToDevice, r->stream));
    HIP_TRY(hipMemcpyAsync(f->item_colors.p, colors, (size_t)n_instances * 16, hipMemcpyHostToDevice, r->stream));
    HIP_TRY(r->sync()); // `items` and the caller's arrays may go away
    f->n_items = (uint32_t)items.size();
    f->items_host = std::move(items);
    f->items_scene = sc;
    f->items_generation = sc->generation;
    f->items_ranges_valid = false; // computed by render_impl together with the primitive total
    f->item_projective_of[0] = all_instances_plain(transforms, n_instances) ? 0 : 1;
    f->item_tame_of[0] = all_colors_tame(colors, n_instances) ? 1 : 0;
    f->pairs_known = false; // a different pass: re-learn the tile list size
    return render_impl(sc, f);
}
crh_status crh_scene_render(crh_scene* sc, crh_frame* f, const float* transforms, const float* colors) {
    crh_status st = crh_scene_set_instances(sc, transforms, colors);
    if (st != CRH_OK) return st;
    return crh_scene_render_resident(sc, f);
}
namespace {
crh_status download_pixels(crh_frame* f, void* out, uint32_t format) {
    if (!f || !out || (f->format == CRH_FORMAT_RGBA16F) != (format == CRH_FORMAT_RGBA16F)) return CRH_ERR_INVALID_ARGUMENT; // (both RGBA8 formats store RGBA8)
    crh_renderer* r = f->renderer;
    HIP_TRY(hipSetDevice(r->device));
    crh_status st = settle_frame(f);
    if (st != CRH_OK) return st;
    if (f->cleared) { // LoadOp::Clear without a pass since: transparent
        memset(out, 0, f->image_bytes());
        return CRH_OK;
    }
    HIP_TRY(order_after_external(f, r->stream));
    HIP_TRY(hipMemcpyAsync(out, f->rgba8.p, f->image_bytes(), hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(r->sync());
    return CRH_OK;
}
} // namespace
crh_status crh_frame_download(crh_frame* f, void* rgba8) { return download_pixels(f, rgba8, CRH_FORMAT_RGBA8); }
crh_status crh_frame_download_f16(crh_frame* f, void* rgba16f) { return download_pixels(f, rgba16f, CRH_FORMAT_RGBA16F); }
extern "C" crh_status crh_debug_frame_counters(crh_frame* f, uint32_t out[8]) { // tools only (not in the public header)
    HIP_TRY(hipSetDevice(f->renderer->device));
    HIP_TRY(f->renderer->sync());
    HIP_TRY(hipMemcpy(out, f->sets[f->last_set].overflow_p, 32, hipMemcpyDeviceToHost));
    return CRH_OK;
}
extern "C" crh_status crh_debug_frame_counters16(crh_frame* f, uint32_t out[16]) { // tools only: + the per-class entry counts of an ablation build
    HIP_TRY(hipSetDevice(f->renderer->device));
    HIP_TRY(f->renderer->sync());
    HIP_TRY(hipMemcpy(out, f->sets[f->last_set].overflow_p, 64, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(static_cast<uint8_t*>(f->sets[f->last_set].overflow_p) + 32, 0, 32));
    return CRH_OK;
}
extern "C" crh_status crh_debug_frame_words(crh_frame* f, uint32_t out[128]) { // tools only: the whole 512-byte flag / counter block of the last set
    HIP_TRY(hipSetDevice(f->renderer->device));
    HIP_TRY(f->renderer->sync());
    HIP_TRY(hipMemcpy(out, f->sets[f->last_set].overflow_p, 512, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(static_cast<uint8_t*>(f->sets[f->last_set].overflow_p) + 320, 0, 192));
    return CRH_OK;
}
extern "C" crh_status crh_debug_frame_bin_dump(crh_frame* f, uint32_t* out, uint32_t n_words) { // tools only (CRH_BIN_DUMP): the items' costs and k_bin_flat's workgroup records
    HIP_TRY(hipSetDevice(f->renderer->device));
    HIP_TRY(f->renderer->sync());
    if ((size_t)n_words * 4 > f->item_cost.cap) return CRH_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipMemcpy(out, f->item_cost.p, (size_t)n_words * 4, hipMemcpyDeviceToHost));
    return CRH_OK;
}
// tests only (host code, no device): the runs flat_batches cuts `n_items` items of the given costs into — runs[2 k], runs[2 k + 1] — and
// the limits of a batch (items, triangles, edges, tile cells) in limits[4]; returns the number of runs, or -1 if `cap_runs` is too small
extern "C" int crh_debug_flat_batches(const uint32_t* cost, uint32_t n_items, uint32_t* runs, uint32_t cap_runs, uint32_t limits[4]) {
    std::vector<uint32_t> out;
    flat_batches(cost, n_items, out);
    flat_batch_limits(n_items, limits);
    if (out.size() / 2 > cap_runs) return -1;
    std::copy(out.begin(), out.end(), runs);
    return (int)(out.size() / 2);
}
crh_status crh_frame_device_pointer(crh_frame* f, void** out) {
    if (!f || !out) return CRH_ERR_INVALID_ARGUMENT;
    crh_status st = settle_frame(f);
    if (st != CRH_OK) return st;
    HIP_TRY(hipSetDevice(f->renderer->device));
    HIP_TRY(order_after_external(f, f->renderer->stream));
    if (f->cleared) // LoadOp::Clear without a pass since: the buffer still holds the previous pass' pixels, the frame is transparent
        HIP_TRY(hipMemsetAsync(f->rgba8.p, 0, f->image_bytes(), f->renderer->stream));
    HIP_TRY(hipStreamSynchronize(f->renderer->stream));
    *out = f->rgba8.p;
    return CRH_OK;
}
// ---- internal accessors for csrc/comm.hip (the multi-GPU exchange); not part of the public header
crh_status crh_internal_frame_geometry(crh_frame* f, uint32_t* width, uint32_t* height, uint32_t* format, int* device) { // no waiting, cannot fail for a live frame
    if (!f || !f->renderer || !width || !height || !format || !device) return CRH_ERR_INVALID_ARGUMENT;
    *width = f->width, *height = f->height, *format = f->format, *device = f->renderer->device;
    return CRH_OK;
}
// the pixel rows the frame's passes draw (crh_frame_set_tile_rows; the whole frame by default)
crh_status crh_internal_frame_slab(crh_frame* f, uint32_t* row_begin, uint32_t* row_end) {
    if (!f || !row_begin || !row_end) return CRH_ERR_INVALID_ARGUMENT;
    *row_begin = std::min(f->height, f->slab_ty0 * 16u);
    *row_end = f->slab_ty1 == 0xFFFFFFFFu ? f->height : std::min(f->height, f->slab_ty1 * 16u);
    return CRH_OK;
}
crh_status crh_internal_frame_info(crh_frame* f, void** rgba8, uint32_t* width, uint32_t* height, int* device) {
    if (!f || !f->renderer || !rgba8 || !width || !height || !device) return CRH_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(f->renderer->device));
    const crh_status st = settle_frame_cheaply(f); // waits for the last pass into THIS frame only: the next step may already be in flight
    if (st != CRH_OK) return st;
    if (f->cleared) { // LoadOp::Clear without a pass since: transparent
        HIP_TRY(order_after_external(f, f->renderer->aux_stream));
        HIP_TRY(hipMemsetAsync(f->rgba8.p, 0, f->image_bytes(), f->renderer->aux_stream));
        HIP_TRY(hipStreamSynchronize(f->renderer->aux_stream));
    }
    *rgba8 = f->rgba8.p, *width = f->width, *height = f->height, *device = f->renderer->device;
    return CRH_OK;
}
// The exchange read (written = 0) or wrote (written = 1) the frame's pixels with work enqueued on `stream`: recorded as an event, so that
// the exchange need not wait on the host and the frame's next pass / download is ordered behind it.
crh_status crh_internal_frame_touched(crh_frame* f, void* stream, int written) {
    if (!f) return CRH_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipEventRecord(written ? f->ext_write : f->ext_read, static_cast<hipStream_t>(stream)));
    if (written) {
        f->counts_describe_pixels = false;
        f->ext_write_set = true;
        f->carry_recolor = f->carry;
        f->cleared = false; // it now shows an image, nothing of its own is pending
        f->check_pending = false;
        f->last_scene = nullptr;
    } else {
        f->ext_read_set = true;
    }
    return CRH_OK;
}
// The entries per tile of the frame's last pass, on the device, when a tile without entries is a transparent tile (NULL otherwise: the
// exchange then looks at the pixels). Valid after crh_internal_frame_info (which settles the pass).
crh_status crh_internal_frame_tile_counts(crh_frame* f, const uint32_t** counts, uint32_t* n_tiles, uint32_t* first_tile, uint32_t* end_tile) {
    if (!f || !counts || !n_tiles || !first_tile || !end_tile) return CRH_ERR_INVALID_ARGUMENT;
    const bool usable = f->counts_describe_pixels && !f->cleared && !f->check_pending && f->sets[f->last_set].used;
    *counts = usable ? f->sets[f->last_set].tile_count_cursor.as<uint32_t>() + f->n_tiles : nullptr;
    *n_tiles = f->n_tiles;
    // (a frame with a slab: the tiles outside it have entries — every rank bins everything — but were not drawn: transparent)
    *first_tile = std::min(f->slab_ty0, f->tiles_y) * f->tiles_x, *end_tile = std::min(f->slab_ty1, f->tiles_y) * f->tiles_x;
    return CRH_OK;
}
// The tile split of the multi-GPU path ("shard by tile"): the passes into this frame draw the tile rows that cover the pixel rows
// [row_begin, row_end) only — crh_comm_slab_rows gives a rank's —, everything else stays transparent (the buffer is cleared here, once; the
// raster kernels leave the other tiles alone from then on). Binning and tessellation are not split: every rank has every list. Exchanged
// with crh_frame_exchange like any layer: the all-to-all then carries nothing (a rank's layer is empty outside its own slab), the
// composite of a slab is its owner's pixels, and the gathered image is bit-equal to a single GPU's — no compositing of rounded layers.
crh_status crh_frame_set_tile_rows(crh_frame* f, uint32_t row_begin, uint32_t row_end) {
    if (!f || !f->renderer || row_begin > row_end || row_end > f->height || ((row_begin % 16u) != 0u && row_begin != f->height) || ((row_end % 16u) != 0u && row_end != f->height)) return CRH_ERR_INVALID_ARGUMENT; // (an empty slab, e.g. (height, height), is a rank without tile rows)
    crh_renderer* r = f->renderer;
    HIP_TRY(hipSetDevice(r->device));
    const crh_status st = settle_frame(f); // (what the frame shows is final, nothing of it will be drawn again)
    if (st != CRH_OK) return st;
    {
        const crh_status ext = wait_for_external(f);
        if (ext != CRH_OK) return ext;
    }
    HIP_TRY(hipMemsetAsync(f->rgba8.p, 0, f->image_bytes(), r->stream));
    f->slab_ty0 = (row_begin + 15u) / 16u, f->slab_ty1 = (row_begin == 0u && row_end == f->height) ? 0xFFFFFFFFu : (row_end + 15u) / 16u;
    f->cleared = true, f->counts_describe_pixels = false, f->check_pending = false, f->last_scene = nullptr;
    f->pairs_known = false, f->tile_order_ready = false; // (the next pass is a verified one: it orders the slab's tiles)
    return CRH_OK;
}
int crh_internal_renderer_device(crh_renderer* r) { return r ? r->device : -1; }

crh_status crh_composite_over(crh_renderer* r, const void* const* layers_dev, uint32_t n_layers, uint64_t n_pixels, void* dst_dev) {
    if (!r || !layers_dev || !dst_dev || n_layers == 0) return CRH_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(r->device));
    // on its own stream and synchronised alone: the next frame may already be in flight on the render streams
    if (n_layers > r->composite_table_capacity) {
        HIP_TRY(hipStreamSynchronize(r->aux_stream));
        if (r->composite_table) (void)hipFree(r->composite_table);
        r->composite_table = nullptr;
        r->composite_table_capacity = 0;
        HIP_TRY(hipMalloc(&r->composite_table, sizeof(void*) * n_layers));
        r->composite_table_capacity = n_layers;
    }
    HIP_TRY(hipMemcpyAsync(r->composite_table, layers_dev, sizeof(void*) * n_layers, hipMemcpyHostToDevice, r->aux_stream));
    launch_composite(static_cast<const uint8_t* const*>(r->composite_table), n_layers, n_pixels, static_cast<uint8_t*>(dst_dev), r->aux_stream);
    HIP_TRY(hipStreamSynchronize(r->aux_stream));
    return CRH_OK;
}
crh_status crh_frame_synchronize(crh_frame* f) {
    if (!f || !f->renderer) return CRH_ERR_INVALID_ARGUMENT;
    crh_renderer* r = f->renderer;
    HIP_TRY(hipSetDevice(r->device));
    const crh_status st = settle_frame_cheaply(f);
    if (st != CRH_OK) return st;
    return wait_for_external(f); // an exchange may still be unpacking into (or packing out of) the pixels on its communicator's stream
}

crh_status crh_selftest_fmath(crh_renderer* r, int fn, const float* a, const float* b, float* out, uint64_t n) {
    if (!r || !a || !b || !out) return CRH_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(r->device));
    DevBuf da, db, dout;
    crh_status rc = CRH_OK;
    if (!hip_ok(da.ensure(n * 4), "hipMalloc") || !hip_ok(db.ensure(n * 4), "hipMalloc") || !hip_ok(dout.ensure(n * 4), "hipMalloc")) rc = CRH_ERR_HIP;
    if (rc == CRH_OK && n) {
        if (!hip_ok(hipMemcpyAsync(da.p, a, n * 4, hipMemcpyHostToDevice, r->stream), "memcpy") ||
            !hip_ok(hipMemcpyAsync(db.p, b, n * 4, hipMemcpyHostToDevice, r->stream), "memcpy"))
            rc = CRH_ERR_HIP;
        if (rc == CRH_OK) {
            launch_fmath(fn, da.as<float>(), db.as<float>(), dout.as<float>(), n, r->stream);
            if (!hip_ok(hipMemcpyAsync(out, dout.p, n * 4, hipMemcpyDeviceToHost, r->stream), "memcpy") || !hip_ok(r->sync(), "sync")) rc = CRH_ERR_HIP;
        }
    }
    da.release();
    db.release();
    dout.release();
    return rc;
}

crh_status crh_renderer_synchronize(crh_renderer* r) {
    if (!r) return CRH_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(r->sync());
    for (crh_frame* f : r->frames) { // ... and the exchanges still working on its frames (their streams belong to the communicators)
        const crh_status st = wait_for_external(f);
        if (st != CRH_OK) return st;
    }
    return CRH_OK;
}
void* crh_renderer_stream(crh_renderer* r) { return r ? (void*)r->stream : nullptr; }
crh_status crh_renderer_enable_timing(crh_renderer* r, int enabled) {
    if (!r) return CRH_ERR_INVALID_ARGUMENT;
    r->timing = enabled == 1 ? 7u : (enabled == 2 ? 1u : 0u); // 1: every lane, 2: the raster lane only (two events per step instead of a dozen)
    r->marks.clear();
    r->events_used = 0;
    return CRH_OK;
}
crh_status crh_renderer_kernel_times(crh_renderer* r, crh_kernel_time* out, uint32_t capacity, uint32_t* count) {
    if (!r || !count) return CRH_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(r->sync());
    uint32_t n = 0;
    size_t previous_of_lane[3] = {SIZE_MAX, SIZE_MAX, SIZE_MAX};
    for (size_t i = 0; i < r->marks.size(); ++i) {
        const size_t before = previous_of_lane[r->marks[i].lane];
        previous_of_lane[r->marks[i].lane] = i;
        if (r->marks[i].name.empty() || before == SIZE_MAX) continue; // a begin mark: the gap before it is host time, not a kernel
        if (n < capacity && out) {
            float ms = 0.0f;
            HIP_TRY(hipEventElapsedTime(&ms, r->marks[before].event, r->marks[i].event));
            std::snprintf(out[n].name, sizeof(out[n].name), "%s", r->marks[i].name.c_str());
            out[n].ms = ms;
            out[n].algorithmic_bytes = r->marks[i].bytes;
        }
        ++n;
    }
    *count = n;
    if (out && getenv("CRH_TIMELINE") && !r->marks.empty()) { // development: when every mark was reached, relative to the first one (the lanes overlap)
        const size_t first = r->marks.size() > 120 ? r->marks.size() - 120 : 0;
        for (size_t i = first; i < r->marks.size(); ++i) {
            float ms = 0.0f;
            (void)hipEventElapsedTime(&ms, r->marks[first].event, r->marks[i].event);
            std::fprintf(stderr, "[timeline] %9.1f us  lane %d  %s\n", ms * 1000.0f, r->marks[i].lane, r->marks[i].name.empty() ? "(begin)" : r->marks[i].name.c_str());
        }
    }
    if (out) { // drained
        r->marks.clear();
        r->events_used = 0;
    }
    return CRH_OK;
}
}
