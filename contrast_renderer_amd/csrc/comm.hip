// csrc/comm.hip — the exchange step of the path-sharded renderer behind the C ABI (the reference has no multi-GPU
// code, component 18 of SURVEY.md §2 is absent upstream).
//
// One process per GPU. Rank g renders the contiguous Shape range crh_comm_shard() gives it into a private full-size premultiplied
// layer (RGBA8, or RGBA16F: CRH_FORMAT_RGBA16F frames); this file turns the `world` layers into one RGBA8 image on rank 0:
//   1. occupancy   a bitmap of the layer's 16x16 tiles that hold anything, and the non-empty tiles packed in tile order (1 or 2 KiB each);
//   2. all-gather  of [4 header words | bitmap] (n_tiles / 8 bytes per rank) — afterwards every rank can compute every transfer size on
//                  the host, and knows whether every rank's layer was readable (header: magic, width, height | format << 24, status);
//   3. all-to-all  the frame is cut into `world` slabs of tile rows; rank r receives the non-empty tiles of slab r of every layer: one
//                  grouped ncclSend / ncclRecv per peer, so a GPU drives all its xGMI links at once with 1/world of what it drew (xGMI is
//                  point to point: a ring reduction would be per-link bound and world - 1 steps deep, and "over" does not commute);
//   4. composite   rank r blends its slab in rank order — dst = src + dst * (1 - src.a), lower rank underneath — reading only the tiles
//                  that exist, quantises ONCE to RGBA8 and packs the non-empty result tiles;
//   5. gather      those go to rank 0 (again only non-empty tiles), which unpacks them into the result frame.
// Empty tiles never travel: for the benchmark scene a rank's layer of 1/8 of the Shapes is mostly empty.
// Host synchronisation: ONE wait per exchange, for the gathered bitmaps in pinned memory (the transfer sizes ncclSend / ncclRecv take are
// host arguments). Reads of the layer and writes of the result frame are handed to the frames as events (crh_internal_frame_touched), so
// the call returns with the tail of the exchange still in flight on the communicator's stream.
// Transports: RCCL (the librccl the process has already mapped, else librccl.so; resolved on first use, so single-GPU users need no
// RCCL) and an in-process loopback over several communicators of ONE device (crh_comm_create_local / crh_comm_local_exchange), which
// runs the same kernels and host logic with device-to-device copies and is what the single-GPU tests and bench.py --loopback drive.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <link.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/contrast_hip.h"

namespace crh {
void set_last_error(const std::string& text);
}
// internal accessors of api.hip (not part of the public header)
extern "C" crh_status crh_internal_frame_geometry(crh_frame* f, uint32_t* width, uint32_t* height, uint32_t* format, int* device);
extern "C" crh_status crh_internal_frame_info(crh_frame* f, void** rgba8, uint32_t* width, uint32_t* height, int* device);
extern "C" crh_status crh_internal_frame_touched(crh_frame* f, void* stream, int written);
extern "C" crh_status crh_internal_frame_slab(crh_frame* f, uint32_t* row_begin, uint32_t* row_end);
extern "C" crh_status crh_internal_frame_tile_counts(crh_frame* f, const uint32_t** counts, uint32_t* n_tiles, uint32_t* first_tile, uint32_t* end_tile);
extern "C" int crh_internal_renderer_device(crh_renderer* r);

namespace {
using crh::set_last_error;

struct Rccl { // the entry points used, resolved with dlsym
    void* lib = nullptr;
    std::string path;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr; // optional: crh_comm_info
    ncclResult_t (*GetVersion)(int*) = nullptr;
};
// A process that imported torch has torch's bundled librccl mapped already; a second copy opened by bare name would be a second RCCL
// runtime in one process (two sets of proxy threads and IPC state). So: the path of a librccl that is already mapped, if there is one.
int find_loaded_rccl(struct dl_phdr_info* info, size_t, void* out) {
    if (info->dlpi_name && std::strstr(info->dlpi_name, "librccl.so")) {
        *static_cast<std::string*>(out) = info->dlpi_name;
        return 1;
    }
    return 0;
}
Rccl* rccl() {
    static Rccl api;
    static std::once_flag once;
    std::call_once(once, [] {
        std::string loaded;
        dl_iterate_phdr(find_loaded_rccl, &loaded);
        std::vector<std::string> names;
        if (!loaded.empty()) names.push_back(loaded);
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) names.push_back(name);
        for (const std::string& name : names) {
            api.lib = dlopen(name.c_str(), RTLD_NOW | RTLD_GLOBAL);
            if (api.lib) {
                api.path = name;
                break;
            }
        }
        if (api.lib) {
#define CRH_SYM(field, symbol) api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.lib, symbol))
            CRH_SYM(GetUniqueId, "ncclGetUniqueId");
            CRH_SYM(CommInitRank, "ncclCommInitRank");
            CRH_SYM(CommDestroy, "ncclCommDestroy");
            CRH_SYM(GroupStart, "ncclGroupStart");
            CRH_SYM(GroupEnd, "ncclGroupEnd");
            CRH_SYM(Send, "ncclSend");
            CRH_SYM(Recv, "ncclRecv");
            CRH_SYM(AllGather, "ncclAllGather");
            CRH_SYM(GetErrorString, "ncclGetErrorString");
            CRH_SYM(CommCount, "ncclCommCount");
            CRH_SYM(GetVersion, "ncclGetVersion");
#undef CRH_SYM
            if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.GroupStart || !api.GroupEnd || !api.Send || !api.Recv || !api.AllGather) {
                dlclose(api.lib);
                api.lib = nullptr;
            }
        }
    });
    return api.lib ? &api : nullptr;
}

bool hip_ok(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    set_last_error(std::string(what) + ": " + hipGetErrorString(e));
    return false;
}
#define HIP_TRY(expr)                                  \
    do {                                               \
        if (!hip_ok((expr), #expr)) return CRH_ERR_HIP; \
    } while (0)
bool nccl_ok(ncclResult_t rc, const char* what) {
    if (rc == ncclSuccess) return true;
    Rccl* api = rccl();
    set_last_error(std::string(what) + ": " + ((api && api->GetErrorString) ? api->GetErrorString(rc) : "RCCL error"));
    return false;
}
#define NCCL_TRY(expr)                                    \
    do {                                                  \
        if (!nccl_ok((expr), #expr)) return CRH_ERR_HIP;  \
    } while (0)

struct Buf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap && p) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr, cap = 0;
        const size_t want = bytes < 256 ? 256 : bytes;
        const hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr, cap = 0;
    }
    template <typename T>
    T* as() const {
        return static_cast<T*>(p);
    }
};
struct PinnedBuf { // host memory the device copies into / out of without staging
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap && p) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr, cap = 0;
        const size_t want = bytes < 256 ? 256 : bytes;
        const hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr, cap = 0;
    }
    template <typename T>
    T* as() const {
        return static_cast<T*>(p);
    }
};

constexpr uint32_t kTilePixels = 256; // 16 x 16
constexpr size_t kResultTileBytes = 1024; // a composited (RGBA8) tile
constexpr uint32_t kHeaderWords = 4;  // in front of every rank's bitmap: magic, width, height | format << 24, status of the rank's layer
constexpr uint32_t kMagic = 0x43524831u; // "CRH1"

// ---------------------------------------------------------------------------------------------- kernels
// A layer's pixel: P = uint32_t (RGBA8 unorm) or uint2 (four binary16).
__device__ __forceinline__ bool pixel_nonzero(uint32_t v) { return v != 0u; }
__device__ __forceinline__ bool pixel_nonzero(uint2 v) { return ((v.x | v.y) & 0x7FFF7FFFu) != 0u; } // (-0.0 is empty too)
__device__ __forceinline__ void pixel_rgba(uint32_t p, float out[4]) {
    out[0] = (float)(p & 255u) * (1.0f / 255.0f), out[1] = (float)((p >> 8) & 255u) * (1.0f / 255.0f);
    out[2] = (float)((p >> 16) & 255u) * (1.0f / 255.0f), out[3] = (float)(p >> 24) * (1.0f / 255.0f);
}
__device__ __forceinline__ void pixel_rgba(uint2 p, float out[4]) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 a = __builtin_bit_cast(h2, p.x), b = __builtin_bit_cast(h2, p.y);
    out[0] = (float)a[0], out[1] = (float)a[1], out[2] = (float)b[0], out[3] = (float)b[1];
}
template <typename P>
__device__ __forceinline__ P pixel_zero();
template <>
__device__ __forceinline__ uint32_t pixel_zero<uint32_t>() { return 0u; }
template <>
__device__ __forceinline__ uint2 pixel_zero<uint2>() { return make_uint2(0u, 0u); }

// ---- a wavefront per 16x16 tile: lane l holds the four pixels (4 (l & 3) .. + 3, l >> 2) of the tile — 16 bytes of an RGBA8 layer, 32
// of an RGBA16F one —, so a tile row is one 64- / 128-byte segment of the frame and a packed tile (row-major, 1 or 2 KiB) one contiguous
// run of the pack buffer: every access of the exchange kernels is a full-width vector load or store.
template <typename P>
struct alignas(4 * sizeof(P)) Quad {
    P p[4];
};
__device__ __forceinline__ uint32_t wave_in_block() { return (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
template <typename P>
__device__ __forceinline__ Quad<P> load_quad_frame(const P* pixels, uint32_t width, uint32_t height, uint32_t tx, uint32_t ty) {
    const uint32_t lane = threadIdx.x & 63u, x = tx * 16u + (lane & 3u) * 4u, y = ty * 16u + (lane >> 2);
    Quad<P> q;
    for (int i = 0; i < 4; ++i) q.p[i] = pixel_zero<P>();
    if (y >= height) return q;
    const P* row = pixels + (size_t)y * width;
    if ((width & 3u) == 0u) { // rows start 16-byte aligned and a quad is inside the frame or outside it
        if (x < width) q = *reinterpret_cast<const Quad<P>*>(row + x);
    } else {
        for (uint32_t i = 0; i < 4u; ++i)
            if (x + i < width) q.p[i] = row[x + i];
    }
    return q;
}
template <typename P>
__device__ __forceinline__ void store_quad_frame(P* pixels, uint32_t width, uint32_t height, uint32_t tx, uint32_t ty, const Quad<P>& q) {
    const uint32_t lane = threadIdx.x & 63u, x = tx * 16u + (lane & 3u) * 4u, y = ty * 16u + (lane >> 2);
    if (y >= height) return;
    P* row = pixels + (size_t)y * width;
    if ((width & 3u) == 0u) {
        if (x < width) *reinterpret_cast<Quad<P>*>(row + x) = q;
    } else {
        for (uint32_t i = 0; i < 4u; ++i)
            if (x + i < width) row[x + i] = q.p[i];
    }
}
template <typename P>
__device__ __forceinline__ Quad<P> load_quad_packed(const P* pack, size_t slot) { return reinterpret_cast<const Quad<P>*>(pack + slot * kTilePixels)[threadIdx.x & 63u]; }
template <typename P>
__device__ __forceinline__ void store_quad_packed(P* pack, size_t slot, const Quad<P>& q) { reinterpret_cast<Quad<P>*>(pack + slot * kTilePixels)[threadIdx.x & 63u] = q; }

// bit t of `bitmap` <=> tile t of the layer holds a non-zero pixel. One workgroup per bitmap WORD (32 consecutive tiles, eight per
// wavefront with all eight loads in flight): the word is stored whole — no atomics, no clearing pass.
template <typename P>
__global__ __launch_bounds__(256) void k_tile_occupancy(const P* pixels, uint32_t width, uint32_t height, uint32_t tiles_x, uint32_t n_tiles, uint32_t* bitmap) {
    __shared__ uint32_t part[4];
    const uint32_t wave = wave_in_block(), first = blockIdx.x * 32u + wave * 8u;
    Quad<P> q[8];
#pragma unroll
    for (uint32_t i = 0; i < 8u; ++i) {
        const uint32_t tile = first + i;
        if (tile < n_tiles) q[i] = load_quad_frame(pixels, width, height, tile % tiles_x, tile / tiles_x);
        else
            for (int k = 0; k < 4; ++k) q[i].p[k] = pixel_zero<P>();
    }
    uint32_t bits = 0;
#pragma unroll
    for (uint32_t i = 0; i < 8u; ++i) {
        const bool any = pixel_nonzero(q[i].p[0]) | pixel_nonzero(q[i].p[1]) | pixel_nonzero(q[i].p[2]) | pixel_nonzero(q[i].p[3]);
        if (__ballot(any) != 0ull) bits |= 1u << i;
    }
    if ((threadIdx.x & 63u) == 0u) part[wave] = bits << (wave * 8u);
    __syncthreads();
    if (threadIdx.x == 0u) bitmap[blockIdx.x] = part[0] | part[1] | part[2] | part[3];
}
// ... or, where the layer's last pass says how many entries every tile had (and nothing else has touched the pixels): a tile without
// entries was written transparent by the raster kernel — the bitmap without reading the layer (a quarter of the packing at 8192^2).
// A tile with entries that came out transparent all the same is sent as what it is.
// [first_tile, end_tile): the tiles the layer's passes draw (a frame with a slab of tile rows, crh_frame_set_tile_rows: the others have entries but no pixels)
__global__ __launch_bounds__(256) void k_bitmap_from_counts(const uint32_t* counts, uint32_t n_tiles, uint32_t n_words, uint32_t* bitmap, uint32_t first_tile, uint32_t end_tile) {
    const uint32_t tile = blockIdx.x * 256u + threadIdx.x;
    const unsigned long long any = __ballot(tile < n_tiles && tile >= first_tile && tile < end_tile && counts[tile] != 0u);
    const uint32_t lane = threadIdx.x & 63u, word = tile >> 5;
    if ((lane & 31u) == 0u && word < n_words) bitmap[word] = (uint32_t)(any >> (lane & 32u));
}
// prefix[w] = number of set bits in words [0, w) of a bitmap (n_words + 1 entries). Workgroup k < world: bitmap k of `bitmaps`
// (`bitmap_stride` words apart) -> prefixes + k (n_words + 1); workgroup `world` (launched when or_bitmap is given): the union of all
// bitmaps -> or_bitmap, and its prefix -> or_prefix. Per thread a run of words, a shuffle scan per wavefront, sixteen totals through LDS.
__global__ __launch_bounds__(1024) void k_bit_prefix(const uint32_t* bitmaps, uint32_t bitmap_stride, uint32_t n_words, uint32_t world, uint32_t* prefixes, uint32_t* or_bitmap,
                                                     uint32_t* or_prefix) {
    __shared__ uint32_t wave_total[16];
    const bool united = blockIdx.x == world;
    const uint32_t* bitmap = bitmaps + (size_t)blockIdx.x * bitmap_stride;
    uint32_t* prefix = united ? or_prefix : prefixes + (size_t)blockIdx.x * (n_words + 1u);
    const uint32_t per = (n_words + 1023u) / 1024u, begin = threadIdx.x * per, end = min(n_words, begin + per);
    auto word = [&](uint32_t w) -> uint32_t {
        if (!united) return bitmap[w];
        uint32_t v = 0;
        for (uint32_t k = 0; k < world; ++k) v |= bitmaps[(size_t)k * bitmap_stride + w];
        return v;
    };
    uint32_t sum = 0;
    for (uint32_t w = begin; w < end; ++w) {
        const uint32_t v = word(w);
        if (united) or_bitmap[w] = v;
        sum += (uint32_t)__popc(v);
    }
    uint32_t scan = sum; // inclusive over the wavefront
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t d = 1; d < 64u; d <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)scan, d, 64);
        if (lane >= d) scan += v;
    }
    if (lane == 63u) wave_total[threadIdx.x >> 6] = scan;
    __syncthreads();
    uint32_t run = scan - sum;
    for (uint32_t k = 0; k < (threadIdx.x >> 6); ++k) run += wave_total[k];
    for (uint32_t w = begin; w < end; ++w) {
        prefix[w] = run;
        run += (uint32_t)__popc(united ? or_bitmap[w] : bitmap[w]); // (the thread's own stores above)
    }
    if (threadIdx.x == 1023u) prefix[n_words] = run;
}
__device__ __forceinline__ bool tile_bit(const uint32_t* bitmap, uint32_t tile) { return (bitmap[tile >> 5] >> (tile & 31u)) & 1u; }
__device__ __forceinline__ uint32_t tile_rank(const uint32_t* bitmap, const uint32_t* prefix, uint32_t tile) { // set bits below `tile`
    return prefix[tile >> 5] + (uint32_t)__popc(bitmap[tile >> 5] & ((1u << (tile & 31u)) - 1u));
}
// the non-empty tiles of the layer, packed in tile order (row-major inside the tile); a wavefront per tile
template <typename P>
__global__ __launch_bounds__(256) void k_pack_tiles(const P* pixels, uint32_t width, uint32_t height, uint32_t tiles_x, uint32_t n_tiles, const uint32_t* bitmap,
                                                    const uint32_t* prefix, P* pack) {
    const uint32_t tile = blockIdx.x * 4u + wave_in_block();
    if (tile >= n_tiles || !tile_bit(bitmap, tile)) return;
    store_quad_packed(pack, tile_rank(bitmap, prefix, tile), load_quad_frame(pixels, width, height, tile % tiles_x, tile / tiles_x));
}
// Rank r's slab = tiles [slab_begin, slab_end): ordered premultiplied "over" of the `world` layers (layer k = recv[k], the non-empty tiles
// of this slab of rank k's layer in tile order), written as the non-empty RGBA8 tiles of the result in tile order. f32 accumulation, one
// RGBA8 quantisation — the arithmetic of k_composite (raster.hip); a tile a layer does not have is a transparent layer (exact).
// A wavefront per tile; the tiles of up to eight layers are loaded before the first is blended.
struct CompositeJob {
    const uint32_t* bitmaps;  // [world] bitmaps, `bitmap_stride` words apart
    const uint32_t* prefixes; // [world][n_words + 1]
    const void* const* recv;  // [world]
    const uint32_t* or_bitmap;
    const uint32_t* or_prefix;
    uint32_t world, n_words, bitmap_stride, slab_begin, slab_end;
    uint32_t* out; // packed result tiles of the slab
};
template <typename P>
__global__ __launch_bounds__(256) void k_composite_tiles(CompositeJob j) {
    const uint32_t tile = j.slab_begin + blockIdx.x * 4u + wave_in_block();
    if (tile >= j.slab_end || !tile_bit(j.or_bitmap, tile)) return;
    float acc[4][4];
    for (int i = 0; i < 4; ++i)
        for (int c = 0; c < 4; ++c) acc[i][c] = 0.0f;
    for (uint32_t k0 = 0; k0 < j.world; k0 += 8u) {
        Quad<P> q[8];
        bool have[8];
#pragma unroll
        for (uint32_t i = 0; i < 8u; ++i) {
            const uint32_t k = k0 + i;
            have[i] = false;
            if (k < j.world) {
                const uint32_t* bitmap = j.bitmaps + (size_t)k * j.bitmap_stride;
                if (tile_bit(bitmap, tile)) {
                    const uint32_t* prefix = j.prefixes + (size_t)k * (j.n_words + 1u);
                    have[i] = true;
                    q[i] = load_quad_packed(static_cast<const P*>(j.recv[k]), (size_t)(tile_rank(bitmap, prefix, tile) - tile_rank(bitmap, prefix, j.slab_begin)));
                }
            }
        }
#pragma unroll
        for (uint32_t i = 0; i < 8u; ++i) {
            if (!have[i]) continue;
            for (int px = 0; px < 4; ++px) {
                float sr[4];
                pixel_rgba(q[i].p[px], sr);
                const float keep = 1.0f - sr[3];
                for (int c = 0; c < 4; ++c) acc[px][c] = sr[c] + acc[px][c] * keep;
            }
        }
    }
    Quad<uint32_t> out;
    for (int px = 0; px < 4; ++px) {
        uint32_t packed = 0;
        for (int c = 0; c < 4; ++c) {
            const float x = acc[px][c] < 0.0f ? 0.0f : (acc[px][c] > 1.0f ? 1.0f : acc[px][c]);
            packed |= (uint32_t)(int)(x * 255.0f + 0.5f) << (8 * c);
        }
        out.p[px] = packed;
    }
    store_quad_packed(j.out, (size_t)(tile_rank(j.or_bitmap, j.or_prefix, tile) - tile_rank(j.or_bitmap, j.or_prefix, j.slab_begin)), out);
}
// rank 0: the gathered tiles -> the result frame (tiles nobody drew are cleared)
__global__ __launch_bounds__(256) void k_unpack_tiles(uint32_t* rgba8, uint32_t width, uint32_t height, uint32_t tiles_x, uint32_t n_tiles, const uint32_t* bitmap,
                                                      const uint32_t* prefix, const uint32_t* pack) {
    const uint32_t tile = blockIdx.x * 4u + wave_in_block();
    if (tile >= n_tiles) return;
    Quad<uint32_t> q;
    for (int i = 0; i < 4; ++i) q.p[i] = 0u;
    if (tile_bit(bitmap, tile)) q = load_quad_packed(pack, tile_rank(bitmap, prefix, tile));
    store_quad_frame(rgba8, width, height, tile % tiles_x, tile / tiles_x, q);
}
} // namespace

struct crh_comm {
    crh_renderer* renderer = nullptr;
    int device = 0;
    uint32_t rank = 0, world = 1;
    ncclComm_t nccl = nullptr; // RCCL transport
    std::vector<crh_comm*>* local_group = nullptr; // loopback transport: the communicators of the group, by rank (owned by rank 0's)
    hipStream_t stream = nullptr;
    // geometry of the last exchange, and the geometry all ranks were found to agree on (checked once per change, with one extra wait)
    uint32_t width = 0, height = 0, format = 0, tiles_x = 0, tiles_y = 0, n_tiles = 0, n_words = 0;
    uint32_t agreed_width = 0, agreed_height = 0, agreed_format = 0, agreed_words = 0;
    bool agreed = false;
    uint32_t gather_width = 0, gather_height = 0; // crh_frame_gather_slabs: the geometry all ranks were found to share (checked once per change)
    bool gather_agreed = false;
    size_t tile_bytes() const { return format == CRH_FORMAT_RGBA16F ? 2048u : 1024u; }
    uint32_t stride() const { return kHeaderWords + n_words; } // words per rank in bitmaps_all
    Buf bitmap, prefix, pack;        // this rank's layer: [header | bitmap], prefix sums of the bitmap, the packed non-empty tiles
    Buf bitmaps_all, prefixes_all;   // every rank's [header | bitmap] (all-gather) and the prefix sums of the bitmaps
    Buf headers_all;                 // every rank's header alone: the fixed-size collective in front of every exchange
    Buf or_bitmap, or_prefix;        // union: the tiles of the composited image
    Buf recv, recv_table;            // received slab tiles, [world] pointers into `recv`
    Buf slab_out, gathered;          // this rank's composited slab (packed); rank 0: all slabs (packed, tile order)
    PinnedBuf host_headers;          // headers_all on the host (only read when this rank's geometry is not the agreed one)
    PinnedBuf host_all, host_table, host_header;  // bitmaps_all on the host; the pointer table on its way to recv_table; this rank's header on its way to `bitmap`
    std::vector<uint32_t> or_bits;   // union of the bitmaps (host)
    hipEvent_t bitmaps_on_host = nullptr; // THE host wait of an exchange
    hipEvent_t phase_begin[CRH_COMM_PHASES] = {}, phase_end[CRH_COMM_PHASES] = {}; // timing marks around the phases of the last exchange
    hipEvent_t packed = nullptr, composited = nullptr; // loopback: what the other communicators' streams wait for
    bool timed = false;
    // statistics of the last exchange (crh_comm_last_traffic / _peer_bytes)
    uint64_t bytes_sent = 0, bytes_dense = 0;
    std::vector<uint64_t> peer_bytes;
    const uint32_t* host_bitmap(uint32_t k) const { return host_all.as<uint32_t>() + (size_t)k * stride() + kHeaderWords; }
    const uint32_t* host_header_of(uint32_t k) const { return host_all.as<uint32_t>() + (size_t)k * stride(); }
};

namespace {
void shard(uint32_t n, uint32_t rank, uint32_t world, uint32_t* begin, uint32_t* end) {
    const uint32_t base = n / world, extra = n % world;
    *begin = rank * base + (rank < extra ? rank : extra);
    *end = *begin + base + (rank < extra ? 1u : 0u);
}
// tiles [begin, end) of rank r's slab: whole tile rows, so a slab is a contiguous range of tile indices
void slab_tiles(const crh_comm* c, uint32_t r, uint32_t* begin, uint32_t* end) {
    uint32_t r0, r1;
    shard(c->tiles_y, r, c->world, &r0, &r1);
    *begin = r0 * c->tiles_x, *end = r1 * c->tiles_x;
}
uint32_t host_rank(const uint32_t* bitmap, uint32_t tile) { // set bits below `tile`
    uint32_t n = 0;
    for (uint32_t w = 0; w < (tile >> 5); ++w) n += (uint32_t)__builtin_popcount(bitmap[w]);
    if (tile & 31u) n += (uint32_t)__builtin_popcount(bitmap[tile >> 5] & ((1u << (tile & 31u)) - 1u));
    return n;
}
uint32_t host_count(const uint32_t* bitmap, uint32_t begin, uint32_t end) { return host_rank(bitmap, end) - host_rank(bitmap, begin); }

void set_geometry(crh_comm* c, uint32_t w, uint32_t h, uint32_t format) {
    c->width = w, c->height = h, c->format = format;
    c->tiles_x = (w + 15u) / 16u, c->tiles_y = (h + 15u) / 16u, c->n_tiles = c->tiles_x * c->tiles_y;
    c->n_words = (c->n_tiles + 31u) / 32u;
}
crh_status ensure_buffers(crh_comm* c) {
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(c->bitmap.ensure((size_t)c->stride() * 4));
    HIP_TRY(c->prefix.ensure((size_t)(c->n_words + 1) * 4));
    HIP_TRY(c->pack.ensure((size_t)c->n_tiles * c->tile_bytes()));
    HIP_TRY(c->bitmaps_all.ensure((size_t)c->world * c->stride() * 4));
    HIP_TRY(c->prefixes_all.ensure((size_t)c->world * (c->n_words + 1) * 4));
    HIP_TRY(c->or_bitmap.ensure((size_t)c->n_words * 4));
    HIP_TRY(c->or_prefix.ensure((size_t)(c->n_words + 1) * 4));
    HIP_TRY(c->host_all.ensure((size_t)c->world * c->stride() * 4));
    HIP_TRY(c->host_table.ensure(sizeof(void*) * c->world));
    HIP_TRY(c->host_header.ensure(kHeaderWords * 4));
    HIP_TRY(c->headers_all.ensure((size_t)c->world * kHeaderWords * 4));
    HIP_TRY(c->host_headers.ensure((size_t)c->world * kHeaderWords * 4));
    HIP_TRY(c->recv_table.ensure(sizeof(void*) * c->world));
    return CRH_OK;
}
enum Phase { kPack = 0, kPlan, kAllToAll, kComposite, kGather, kUnpack };
void begin_phase(crh_comm* c, Phase k) { (void)hipEventRecord(c->phase_begin[k], c->stream); }
void end_phase(crh_comm* c, Phase k) { (void)hipEventRecord(c->phase_end[k], c->stream); }

// phase 1: [header | occupancy bitmap], its prefix sums and the packed tiles of this rank's layer (all on the communicator's stream).
// A layer that cannot be read (its pass failed) leaves an empty bitmap and its status in the header: the rank still takes part in the
// collectives, and every rank learns of it from the gathered headers.
crh_status phase_pack(crh_comm* c, crh_frame* layer) {
    uint32_t w = 0, h = 0, format = 0;
    int device = 0;
    crh_status st = crh_internal_frame_geometry(layer, &w, &h, &format, &device);
    if (st != CRH_OK) return st;
    if (device != c->device) return CRH_ERR_INVALID_ARGUMENT;
    set_geometry(c, w, h, format);
    if ((st = ensure_buffers(c)) != CRH_OK) return st;
    void* pixels = nullptr;
    const crh_status layer_status = crh_internal_frame_info(layer, &pixels, &w, &h, &device); // settles the frame: its pixels are final and visible
    HIP_TRY(hipSetDevice(c->device));
    begin_phase(c, kPack);
    uint32_t* header = c->host_header.as<uint32_t>(); // (the previous exchange's copy of it was waited for with its bitmaps)
    header[0] = kMagic, header[1] = w, header[2] = h | (format << 24), header[3] = (uint32_t)layer_status;
    HIP_TRY(hipMemcpyAsync(c->bitmap.p, header, kHeaderWords * 4, hipMemcpyHostToDevice, c->stream));
    uint32_t* bitmap = c->bitmap.as<uint32_t>() + kHeaderWords;
    const uint32_t* counts = nullptr;
    uint32_t counted_tiles = 0, first_tile = 0, end_tile = 0;
    if (layer_status == CRH_OK && getenv("CRH_EXCHANGE_SCAN_PIXELS") == nullptr) (void)crh_internal_frame_tile_counts(layer, &counts, &counted_tiles, &first_tile, &end_tile);
    if (layer_status == CRH_OK && counts && counted_tiles == c->n_tiles) {
        hipLaunchKernelGGL(k_bitmap_from_counts, dim3((c->n_words * 32u + 255u) / 256u), dim3(256), 0, c->stream, counts, c->n_tiles, c->n_words, bitmap, first_tile, end_tile);
    } else if (layer_status == CRH_OK) {
        if (format == CRH_FORMAT_RGBA16F)
            hipLaunchKernelGGL(k_tile_occupancy<uint2>, dim3(c->n_words), dim3(256), 0, c->stream, static_cast<const uint2*>(pixels), w, h, c->tiles_x, c->n_tiles, bitmap);
        else
            hipLaunchKernelGGL(k_tile_occupancy<uint32_t>, dim3(c->n_words), dim3(256), 0, c->stream, static_cast<const uint32_t*>(pixels), w, h, c->tiles_x, c->n_tiles, bitmap);
    } else {
        HIP_TRY(hipMemsetAsync(bitmap, 0, (size_t)c->n_words * 4, c->stream));
    }
    hipLaunchKernelGGL(k_bit_prefix, dim3(1), dim3(1024), 0, c->stream, bitmap, c->n_words, c->n_words, 1u, c->prefix.as<uint32_t>(), static_cast<uint32_t*>(nullptr),
                       static_cast<uint32_t*>(nullptr));
    if (layer_status == CRH_OK) {
        const dim3 grid((c->n_tiles + 3u) / 4u);
        if (format == CRH_FORMAT_RGBA16F)
            hipLaunchKernelGGL(k_pack_tiles<uint2>, grid, dim3(256), 0, c->stream, static_cast<const uint2*>(pixels), w, h, c->tiles_x, c->n_tiles, bitmap, c->prefix.as<uint32_t>(),
                               c->pack.as<uint2>());
        else
            hipLaunchKernelGGL(k_pack_tiles<uint32_t>, grid, dim3(256), 0, c->stream, static_cast<const uint32_t*>(pixels), w, h, c->tiles_x, c->n_tiles, bitmap,
                               c->prefix.as<uint32_t>(), c->pack.as<uint32_t>());
        st = crh_internal_frame_touched(layer, c->stream, 0); // the layer's next pass is ordered behind the packing
        if (st != CRH_OK) return st;
    }
    HIP_TRY(hipGetLastError());
    end_phase(c, kPack);
    return CRH_OK;
}
// phase 2 (the [header | bitmap] records of all ranks are in bitmaps_all, or on their way there on the stream): copy to the host, prefix
// sums and union on the device meanwhile, then THE host wait; the gathered headers decide whether the exchange goes ahead.
crh_status phase_plan(crh_comm* c) {
    HIP_TRY(hipMemcpyAsync(c->host_all.p, c->bitmaps_all.p, (size_t)c->world * c->stride() * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipEventRecord(c->bitmaps_on_host, c->stream));
    const uint32_t* first = c->bitmaps_all.as<uint32_t>() + kHeaderWords;
    hipLaunchKernelGGL(k_bit_prefix, dim3(c->world + 1u), dim3(1024), 0, c->stream, first, c->stride(), c->n_words, c->world, c->prefixes_all.as<uint32_t>(),
                       c->or_bitmap.as<uint32_t>(), c->or_prefix.as<uint32_t>()); // every rank's prefix sums, the union and its prefix sums: one launch
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventSynchronize(c->bitmaps_on_host)); // every transfer size follows from the bitmaps
    crh_status peer = CRH_OK;
    for (uint32_t k = 0; k < c->world; ++k) {
        const uint32_t* hd = c->host_header_of(k);
        if (hd[0] != kMagic || hd[1] != c->width || hd[2] != (c->height | (c->format << 24))) {
            set_last_error("crh_frame_exchange: rank " + std::to_string(k) + " exchanges a layer of another size or format");
            c->agreed = false; // the next exchange starts from the headers alone
            return CRH_ERR_INVALID_ARGUMENT; // (every rank sees the same headers and returns here together)
        }
        if (hd[3] != CRH_OK && peer == CRH_OK) {
            peer = (crh_status)hd[3];
            if (k != c->rank) set_last_error("crh_frame_exchange: the layer of rank " + std::to_string(k) + " could not be read (status " + std::to_string(hd[3]) + ")");
        }
    }
    if (peer != CRH_OK) return peer;
    c->or_bits.assign(c->n_words, 0u);
    for (uint32_t k = 0; k < c->world; ++k) {
        const uint32_t* b = c->host_bitmap(k);
        for (uint32_t w = 0; w < c->n_words; ++w) c->or_bits[w] |= b[w];
    }
    uint32_t s0, s1;
    slab_tiles(c, c->rank, &s0, &s1);
    size_t total = 0;
    std::vector<size_t> offset(c->world);
    for (uint32_t k = 0; k < c->world; ++k) {
        offset[k] = total;
        total += (size_t)host_count(c->host_bitmap(k), s0, s1) * c->tile_bytes();
    }
    HIP_TRY(c->recv.ensure(total + c->tile_bytes()));
    const void** table = c->host_table.as<const void*>(); // pinned: the copy below needs no wait (the previous exchange's use of it is behind the wait above)
    for (uint32_t k = 0; k < c->world; ++k) table[k] = static_cast<uint8_t*>(c->recv.p) + offset[k];
    HIP_TRY(hipMemcpyAsync(c->recv_table.p, table, sizeof(void*) * c->world, hipMemcpyHostToDevice, c->stream));
    end_phase(c, kPlan);
    return CRH_OK;
}
// where rank k's tiles of slab r start inside rank k's pack buffer / how many bytes they are
void segment(const crh_comm* c, uint32_t k, uint32_t r, size_t* offset, size_t* bytes) {
    uint32_t s0, s1;
    slab_tiles(c, r, &s0, &s1);
    const uint32_t* bitmap = c->host_bitmap(k);
    *offset = (size_t)host_rank(bitmap, s0) * c->tile_bytes();
    *bytes = (size_t)host_count(bitmap, s0, s1) * c->tile_bytes();
}
uint8_t* recv_slot(const crh_comm* c, uint32_t k) { return static_cast<uint8_t*>(const_cast<void*>(c->host_table.as<const void*>()[k])); } // where layer k's tiles of my slab are received
// phase 4: composite my slab (after the slab tiles of every layer are in `recv`)
crh_status phase_composite(crh_comm* c) {
    begin_phase(c, kComposite);
    uint32_t s0, s1;
    slab_tiles(c, c->rank, &s0, &s1);
    const size_t out_tiles = host_count(c->or_bits.data(), s0, s1);
    HIP_TRY(c->slab_out.ensure(out_tiles * kResultTileBytes + kResultTileBytes));
    if (s1 > s0) {
        CompositeJob j;
        j.bitmaps = c->bitmaps_all.as<uint32_t>() + kHeaderWords, j.prefixes = c->prefixes_all.as<uint32_t>();
        j.recv = static_cast<const void* const*>(c->recv_table.p);
        j.or_bitmap = c->or_bitmap.as<uint32_t>(), j.or_prefix = c->or_prefix.as<uint32_t>();
        j.world = c->world, j.n_words = c->n_words, j.bitmap_stride = c->stride(), j.slab_begin = s0, j.slab_end = s1;
        j.out = c->slab_out.as<uint32_t>();
        if (c->format == CRH_FORMAT_RGBA16F)
            hipLaunchKernelGGL(k_composite_tiles<uint2>, dim3((s1 - s0 + 3u) / 4u), dim3(256), 0, c->stream, j);
        else
            hipLaunchKernelGGL(k_composite_tiles<uint32_t>, dim3((s1 - s0 + 3u) / 4u), dim3(256), 0, c->stream, j);
    }
    HIP_TRY(hipGetLastError());
    end_phase(c, kComposite);
    return CRH_OK;
}
// phase 6 (rank 0, after every slab's tiles are in `gathered`): unpack into the result frame; no host wait — the frame is told
crh_status phase_unpack(crh_comm* c, crh_frame* result) {
    uint32_t w = 0, h = 0, format = 0;
    int device = 0;
    crh_status st = crh_internal_frame_geometry(result, &w, &h, &format, &device);
    if (st != CRH_OK) return st;
    if (w != c->width || h != c->height || device != c->device || format == CRH_FORMAT_RGBA16F) return CRH_ERR_INVALID_ARGUMENT; // (either RGBA8 storage format: the result is the gathered bytes)
    void* pixels = nullptr;
    if ((st = crh_internal_frame_info(result, &pixels, &w, &h, &device)) != CRH_OK) return st; // what the frame showed so far is settled (and discarded)
    HIP_TRY(hipSetDevice(c->device));
    begin_phase(c, kUnpack);
    hipLaunchKernelGGL(k_unpack_tiles, dim3((c->n_tiles + 3u) / 4u), dim3(256), 0, c->stream, static_cast<uint32_t*>(pixels), w, h, c->tiles_x, c->n_tiles,
                       c->or_bitmap.as<uint32_t>(), c->or_prefix.as<uint32_t>(), c->gathered.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    end_phase(c, kUnpack);
    return crh_internal_frame_touched(result, c->stream, 1);
}
void account(crh_comm* c) { // what this rank put on the wire vs. what dense slabs would have cost
    c->bytes_sent = 0;
    c->peer_bytes.assign(c->world, 0);
    for (uint32_t p = 0; p < c->world; ++p) {
        if (p == c->rank) continue;
        size_t off, bytes;
        segment(c, c->rank, p, &off, &bytes);
        c->peer_bytes[p] = bytes;
        c->bytes_sent += bytes;
    }
    uint32_t s0, s1;
    slab_tiles(c, c->rank, &s0, &s1);
    if (c->rank != 0) c->bytes_sent += (size_t)host_count(c->or_bits.data(), s0, s1) * kResultTileBytes;
    c->bytes_dense = (uint64_t)(c->n_tiles - (s1 - s0)) * c->tile_bytes() + (c->rank != 0 ? (uint64_t)(s1 - s0) * kResultTileBytes : 0ull);
}
crh_status create_common(crh_renderer* r, uint32_t rank, uint32_t world, crh_comm** out) {
    if (!r || !out || world == 0 || rank >= world) return CRH_ERR_INVALID_ARGUMENT;
    crh_comm* c = new crh_comm;
    c->renderer = r;
    c->device = crh_internal_renderer_device(r);
    c->rank = rank, c->world = world;
    bool ok = hip_ok(hipSetDevice(c->device), "hipSetDevice") && hip_ok(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking), "hipStreamCreate") &&
              hip_ok(hipEventCreateWithFlags(&c->bitmaps_on_host, hipEventDisableTiming), "hipEventCreate") &&
              hip_ok(hipEventCreateWithFlags(&c->packed, hipEventDisableTiming), "hipEventCreate") &&
              hip_ok(hipEventCreateWithFlags(&c->composited, hipEventDisableTiming), "hipEventCreate");
    for (hipEvent_t& e : c->phase_begin) ok = ok && hip_ok(hipEventCreate(&e), "hipEventCreate");
    for (hipEvent_t& e : c->phase_end) ok = ok && hip_ok(hipEventCreate(&e), "hipEventCreate");
    if (!ok) {
        crh_comm_destroy(c);
        return CRH_ERR_HIP;
    }
    *out = c;
    return CRH_OK;
}
} // namespace

extern "C" {

crh_status crh_comm_shard(uint32_t n_items, uint32_t rank, uint32_t world, uint32_t* begin, uint32_t* end) {
    if (!begin || !end || world == 0 || rank >= world) return CRH_ERR_INVALID_ARGUMENT;
    shard(n_items, rank, world, begin, end);
    return CRH_OK;
}
crh_status crh_comm_slab_rows(uint32_t height, uint32_t rank, uint32_t world, uint32_t* row_begin, uint32_t* row_end) {
    if (!row_begin || !row_end || world == 0 || rank >= world) return CRH_ERR_INVALID_ARGUMENT;
    uint32_t t0, t1;
    shard((height + 15u) / 16u, rank, world, &t0, &t1);
    *row_begin = t0 * 16u < height ? t0 * 16u : height;
    *row_end = t1 * 16u < height ? t1 * 16u : height;
    return CRH_OK;
}
crh_status crh_comm_unique_id(void* id128) {
    if (!id128) return CRH_ERR_INVALID_ARGUMENT;
    Rccl* api = rccl();
    if (!api) {
        set_last_error("librccl.so could not be opened");
        return CRH_ERR_UNSUPPORTED;
    }
    ncclUniqueId id;
    NCCL_TRY(api->GetUniqueId(&id));
    static_assert(sizeof(id) == CRH_COMM_ID_BYTES, "ncclUniqueId");
    std::memcpy(id128, &id, sizeof(id));
    return CRH_OK;
}
crh_status crh_comm_create(crh_renderer* r, uint32_t rank, uint32_t world, const void* id128, crh_comm** out) {
    if (!id128) return CRH_ERR_INVALID_ARGUMENT;
    Rccl* api = rccl();
    if (!api) {
        set_last_error("librccl.so could not be opened");
        return CRH_ERR_UNSUPPORTED;
    }
    crh_comm* c = nullptr;
    const crh_status st = create_common(r, rank, world, &c);
    if (st != CRH_OK) return st;
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    if (!nccl_ok(api->CommInitRank(&c->nccl, (int)world, id, (int)rank), "ncclCommInitRank")) {
        c->nccl = nullptr;
        crh_comm_destroy(c);
        return CRH_ERR_HIP;
    }
    *out = c;
    return CRH_OK;
}
crh_status crh_comm_create_local(crh_renderer* r, uint32_t rank, uint32_t world, crh_comm* rank0, crh_comm** out) {
    if (world == 0 || rank >= world) return CRH_ERR_INVALID_ARGUMENT;
    if ((rank == 0) != (rank0 == nullptr)) return CRH_ERR_INVALID_ARGUMENT; // rank 0 founds the group, the others join it
    if (rank0 && (!rank0->local_group || rank0->world != world || (*rank0->local_group)[rank] != nullptr)) return CRH_ERR_INVALID_ARGUMENT;
    crh_comm* c = nullptr;
    const crh_status st = create_common(r, rank, world, &c);
    if (st != CRH_OK) return st;
    c->local_group = rank0 ? rank0->local_group : new std::vector<crh_comm*>(world, nullptr);
    (*c->local_group)[rank] = c;
    *out = c;
    return CRH_OK;
}
void crh_comm_destroy(crh_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->nccl && rccl()) (void)rccl()->CommDestroy(c->nccl);
    if (c->local_group) {
        (*c->local_group)[c->rank] = nullptr;
        bool empty = true;
        for (crh_comm* m : *c->local_group) empty = empty && m == nullptr;
        if (empty) delete c->local_group;
    }
    for (Buf* b : {&c->bitmap, &c->prefix, &c->pack, &c->bitmaps_all, &c->prefixes_all, &c->or_bitmap, &c->or_prefix, &c->recv, &c->recv_table, &c->slab_out, &c->gathered, &c->headers_all}) b->release();
    for (PinnedBuf* b : {&c->host_all, &c->host_table, &c->host_header, &c->host_headers}) b->release();
    for (hipEvent_t e : {c->bitmaps_on_host, c->packed, c->composited})
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->phase_begin)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->phase_end)
        if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}
crh_status crh_comm_last_traffic(const crh_comm* c, uint64_t* bytes_sent, uint64_t* bytes_dense) {
    if (!c) return CRH_ERR_INVALID_ARGUMENT;
    if (bytes_sent) *bytes_sent = c->bytes_sent;
    if (bytes_dense) *bytes_dense = c->bytes_dense;
    return CRH_OK;
}
crh_status crh_comm_last_peer_bytes(const crh_comm* c, uint64_t* per_peer) {
    if (!c || !per_peer) return CRH_ERR_INVALID_ARGUMENT;
    for (uint32_t p = 0; p < c->world; ++p) per_peer[p] = p < c->peer_bytes.size() ? c->peer_bytes[p] : 0;
    return CRH_OK;
}
crh_status crh_comm_info(const crh_comm* c, uint32_t* nranks, int32_t* rccl_version) {
    if (!c) return CRH_ERR_INVALID_ARGUMENT;
    if (nranks) *nranks = c->world;
    if (rccl_version) *rccl_version = 0;
    if (!c->nccl) return CRH_OK; // a loopback communicator
    Rccl* api = rccl();
    if (!api) return CRH_ERR_UNSUPPORTED;
    if (nranks && api->CommCount) {
        int n = 0;
        NCCL_TRY(api->CommCount(c->nccl, &n));
        *nranks = (uint32_t)n;
    }
    if (rccl_version && api->GetVersion) {
        int v = 0;
        NCCL_TRY(api->GetVersion(&v));
        *rccl_version = v;
    }
    return CRH_OK;
}
crh_status crh_comm_last_timing(crh_comm* c, float ms[CRH_COMM_PHASES]) {
    if (!c || !ms) return CRH_ERR_INVALID_ARGUMENT;
    for (int k = 0; k < CRH_COMM_PHASES; ++k) ms[k] = 0.0f;
    if (!c->timed) return CRH_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const int last = c->rank == 0 ? CRH_COMM_PHASES : CRH_COMM_PHASES - 1; // only rank 0 unpacks
    for (int k = 0; k < last; ++k) HIP_TRY(hipEventElapsedTime(&ms[k], c->phase_begin[k], c->phase_end[k]));
    return CRH_OK;
}

// The collective: every rank calls it with its layer; rank 0 also passes the frame that receives the image (the others pass NULL).
crh_status crh_frame_exchange(crh_comm* c, crh_frame* layer, crh_frame* result) {
    if (!c || !layer || !c->nccl || (c->rank == 0) != (result != nullptr)) return CRH_ERR_INVALID_ARGUMENT;
    Rccl* api = rccl();
    c->timed = false;
    crh_status st = phase_pack(c, layer);
    if (st != CRH_OK) return st; // (only argument errors end here: a layer that cannot be read still takes part)
    // Sizes of the collectives follow from the frame geometry, which therefore has to be the same on every rank BEFORE a count is derived
    // from it. Two rules keep the ranks' collectives matched whatever one of them does to its layer:
    //   (1) every exchange opens with the all-gather of the 16-byte headers — a fixed size, nothing to disagree about;
    //   (2) a rank that holds an agreement issues the bitmap all-gather AT THE AGREED SIZE — also a rank whose own layer has since
    //       changed size or format (its payload is then never read): its peers, who cannot know yet, issue exactly that collective.
    // A rank whose geometry is the agreed one reads the headers together with the bitmaps, behind the ONE host wait of the exchange
    // (phase_plan), and learns of a peer's change there; a rank without an agreement, or whose own geometry has changed, waits for the
    // headers alone first: all equal to its own -> that is the new agreement and the bitmaps follow at the new size (one extra wait: the
    // first exchange, a resize of every rank's target); otherwise every rank returns CRH_ERR_INVALID_ARGUMENT — the changed rank here,
    // its peers in phase_plan — and the next exchange starts from the headers alone.
    begin_phase(c, kPlan);
    NCCL_TRY(api->AllGather(c->bitmap.p, c->headers_all.p, kHeaderWords * 4, ncclUint8, c->nccl, c->stream));
    const bool mine_agreed = c->agreed && c->agreed_width == c->width && c->agreed_height == c->height && c->agreed_format == c->format;
    if (c->agreed) // (the buffers only ever grow: they hold the agreed size whichever way this rank's geometry went)
        NCCL_TRY(api->AllGather(c->bitmap.p, c->bitmaps_all.p, (size_t)(kHeaderWords + c->agreed_words) * 4, ncclUint8, c->nccl, c->stream));
    if (!mine_agreed) {
        HIP_TRY(hipMemcpyAsync(c->host_headers.p, c->headers_all.p, (size_t)c->world * kHeaderWords * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipEventRecord(c->bitmaps_on_host, c->stream));
        HIP_TRY(hipEventSynchronize(c->bitmaps_on_host));
        for (uint32_t k = 0; k < c->world; ++k) {
            const uint32_t* hd = c->host_headers.as<uint32_t>() + (size_t)k * kHeaderWords;
            if (hd[0] != kMagic || hd[1] != c->width || hd[2] != (c->height | (c->format << 24))) {
                set_last_error("crh_frame_exchange: rank " + std::to_string(k) + " exchanges a layer of another size or format");
                c->agreed = false;
                return CRH_ERR_INVALID_ARGUMENT; // on every rank
            }
        }
        c->agreed = true, c->agreed_width = c->width, c->agreed_height = c->height, c->agreed_format = c->format, c->agreed_words = c->n_words;
        NCCL_TRY(api->AllGather(c->bitmap.p, c->bitmaps_all.p, (size_t)c->stride() * 4, ncclUint8, c->nccl, c->stream));
    }
    if ((st = phase_plan(c)) != CRH_OK) return st; // (a peer's failure is seen by all ranks here: they return together, nothing is in flight)
    // all-to-all of the slab tiles: one group, so that all links are driven at once. An error inside the group still closes it.
    begin_phase(c, kAllToAll);
    bool ok = nccl_ok(api->GroupStart(), "ncclGroupStart");
    if (!ok) return CRH_ERR_HIP;
    for (uint32_t p = 0; p < c->world && ok; ++p) {
        size_t off, bytes;
        segment(c, c->rank, p, &off, &bytes); // my tiles of slab p
        if (p == c->rank) {
            if (bytes) ok = hip_ok(hipMemcpyAsync(recv_slot(c, c->rank), static_cast<uint8_t*>(c->pack.p) + off, bytes, hipMemcpyDeviceToDevice, c->stream), "hipMemcpyAsync");
            continue;
        }
        if (bytes) ok = nccl_ok(api->Send(static_cast<uint8_t*>(c->pack.p) + off, bytes, ncclUint8, (int)p, c->nccl, c->stream), "ncclSend");
        segment(c, p, c->rank, &off, &bytes); // rank p's tiles of my slab
        if (bytes && ok) ok = nccl_ok(api->Recv(recv_slot(c, p), bytes, ncclUint8, (int)p, c->nccl, c->stream), "ncclRecv");
    }
    ok = nccl_ok(api->GroupEnd(), "ncclGroupEnd") && ok;
    if (!ok) return CRH_ERR_HIP;
    end_phase(c, kAllToAll);
    if ((st = phase_composite(c)) != CRH_OK) return st;
    // gather of the composited slabs' non-empty tiles on rank 0 (tile order = slab order)
    if (c->rank == 0) HIP_TRY(c->gathered.ensure((size_t)host_rank(c->or_bits.data(), c->n_tiles) * kResultTileBytes + kResultTileBytes));
    begin_phase(c, kGather);
    ok = nccl_ok(api->GroupStart(), "ncclGroupStart");
    if (!ok) return CRH_ERR_HIP;
    for (uint32_t p = 0; p < c->world && ok; ++p) {
        uint32_t s0, s1;
        slab_tiles(c, p, &s0, &s1);
        const size_t off = (size_t)host_rank(c->or_bits.data(), s0) * kResultTileBytes, bytes = (size_t)host_count(c->or_bits.data(), s0, s1) * kResultTileBytes;
        if (!bytes) continue;
        if (c->rank == 0 && p == 0)
            ok = hip_ok(hipMemcpyAsync(static_cast<uint8_t*>(c->gathered.p) + off, c->slab_out.p, bytes, hipMemcpyDeviceToDevice, c->stream), "hipMemcpyAsync");
        else if (c->rank == 0)
            ok = nccl_ok(api->Recv(static_cast<uint8_t*>(c->gathered.p) + off, bytes, ncclUint8, (int)p, c->nccl, c->stream), "ncclRecv");
        else if (p == c->rank)
            ok = nccl_ok(api->Send(c->slab_out.p, bytes, ncclUint8, 0, c->nccl, c->stream), "ncclSend");
    }
    ok = nccl_ok(api->GroupEnd(), "ncclGroupEnd") && ok;
    if (!ok) return CRH_ERR_HIP;
    end_phase(c, kGather);
    account(c);
    c->timed = true;
    if (c->rank == 0) return phase_unpack(c, result);
    return CRH_OK; // no wait: the layer's next pass is ordered behind its packing, the buffers' next use is on this stream
}

// The tile split's own exchange (round 5): rank g's layer holds its slab of tile rows and nothing else (crh_frame_set_tile_rows), every slab has
// the same place in every rank's frame, so the slabs go STRAIGHT from the layers' pixel rows into the result frame's — one grouped
// ncclSend / ncclRecv per rank, no occupancy bitmaps, no packing, no plan on the host, no composite, no unpacking (crh_frame_exchange of
// such layers works too, and spends 0.2 ms of its 0.25 on phases that have nothing to do). RGBA8 storage on both sides.
namespace {
crh_status slab_bytes(const crh_comm* c, uint32_t width, uint32_t height, uint32_t rank, size_t* offset, size_t* bytes) {
    uint32_t r0 = 0, r1 = 0;
    const crh_status st = crh_comm_slab_rows(height, rank, c->world, &r0, &r1);
    *offset = (size_t)r0 * width * 4u, *bytes = (size_t)(r1 - r0) * width * 4u;
    return st;
}
void mark_all_phases(crh_comm* c) { // (crh_comm_last_timing reads every phase: the ones this exchange does not have are empty)
    for (int k = 0; k < CRH_COMM_PHASES; ++k)
        if (k != kGather) begin_phase(c, (Phase)k), end_phase(c, (Phase)k);
}
} // namespace
crh_status crh_frame_gather_slabs(crh_comm* c, crh_frame* layer, crh_frame* result) {
    if (!c || !c->nccl) return CRH_ERR_INVALID_ARGUMENT; // (no communicator: nobody is waiting for this rank)
    Rccl* api = rccl();
    c->timed = false;
    // Everything that can be wrong on THIS rank — its arguments, its layer, rank 0's result frame, the layer's slab — becomes a status word that travels
    // with the header, so that every rank leaves the call together instead of one returning early from in front of a collective the others are already
    // in (rank 0 used to return before the all-gather on a result frame of another size, and behind it when the result could not be read).
    uint32_t w = 0, h = 0, format = 0, rw = 0, rh = 0, rformat = 0;
    int device = c->device;
    crh_status mine = (!layer || (c->rank == 0) != (result != nullptr)) ? CRH_ERR_INVALID_ARGUMENT : CRH_OK;
    if (mine == CRH_OK) mine = crh_internal_frame_geometry(layer, &w, &h, &format, &device);
    if (mine == CRH_OK && (device != c->device || format == CRH_FORMAT_RGBA16F)) mine = CRH_ERR_INVALID_ARGUMENT;
    if (mine == CRH_OK && result) {
        mine = crh_internal_frame_geometry(result, &rw, &rh, &rformat, &device);
        if (mine == CRH_OK && (rw != w || rh != h || rformat == CRH_FORMAT_RGBA16F || device != c->device)) mine = CRH_ERR_INVALID_ARGUMENT;
    }
    HIP_TRY(hipSetDevice(c->device));
    void *pixels = nullptr, *out = nullptr;
    if (mine == CRH_OK) mine = crh_internal_frame_info(layer, &pixels, &w, &h, &device); // settles the layer: its pixels are final
    if (mine == CRH_OK && result) mine = crh_internal_frame_info(result, &out, &rw, &rh, &device);
    if (mine == CRH_OK) { // the layer draws exactly this rank's slab of rows (crh_frame_set_tile_rows(crh_comm_slab_rows(...))): anything else would gather transparent or partial rows
        uint32_t row0 = 0, row1 = 0, slab0 = 0, slab1 = 0;
        mine = crh_internal_frame_slab(layer, &slab0, &slab1);
        if (mine == CRH_OK) mine = crh_comm_slab_rows(h, c->rank, c->world, &row0, &row1);
        if (mine == CRH_OK && row0 != row1 && (slab0 != row0 || slab1 != row1)) {
            set_last_error("crh_frame_gather_slabs: the layer of rank " + std::to_string(c->rank) + " draws rows " + std::to_string(slab0) + " .. " + std::to_string(slab1) + ", its slab is " + std::to_string(row0) +
                           " .. " + std::to_string(row1));
            mine = CRH_ERR_INVALID_ARGUMENT;
        }
    }
    // the ranks' transfers only match when their frames have one size, and a failed rank must fail every rank: 16 bytes all-gathered and read on the host
    HIP_TRY(c->host_header.ensure(kHeaderWords * 4));
    HIP_TRY(c->bitmap.ensure(kHeaderWords * 4));
    HIP_TRY(c->headers_all.ensure((size_t)c->world * kHeaderWords * 4));
    HIP_TRY(c->host_headers.ensure((size_t)c->world * kHeaderWords * 4));
    uint32_t* header = c->host_header.as<uint32_t>();
    header[0] = kMagic, header[1] = w, header[2] = h, header[3] = (uint32_t)mine;
    HIP_TRY(hipMemcpyAsync(c->bitmap.p, header, kHeaderWords * 4, hipMemcpyHostToDevice, c->stream));
    mark_all_phases(c);
    NCCL_TRY(api->AllGather(c->bitmap.p, c->headers_all.p, kHeaderWords * 4, ncclUint8, c->nccl, c->stream));
    HIP_TRY(hipMemcpyAsync(c->host_headers.p, c->headers_all.p, (size_t)c->world * kHeaderWords * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipEventRecord(c->bitmaps_on_host, c->stream));
    HIP_TRY(hipEventSynchronize(c->bitmaps_on_host));
    if (mine != CRH_OK) return mine; // (this rank's own reason; the others read it below)
    for (uint32_t k = 0; k < c->world; ++k) {
        const uint32_t* hd = c->host_headers.as<uint32_t>() + (size_t)k * kHeaderWords;
        if (hd[0] == kMagic && hd[3] != (uint32_t)CRH_OK) {
            set_last_error("crh_frame_gather_slabs: rank " + std::to_string(k) + " cannot take part (status " + std::to_string(hd[3]) + ")");
            return (crh_status)hd[3]; // on every rank
        }
        if (hd[0] != kMagic || hd[1] != w || hd[2] != h) {
            set_last_error("crh_frame_gather_slabs: rank " + std::to_string(k) + " gathers a layer of another size");
            return CRH_ERR_INVALID_ARGUMENT; // on every rank
        }
    }
    c->gather_agreed = true, c->gather_width = w, c->gather_height = h;
    crh_status st = CRH_OK;
    begin_phase(c, kGather);
    bool ok = nccl_ok(api->GroupStart(), "ncclGroupStart");
    if (!ok) return CRH_ERR_HIP;
    c->bytes_sent = 0, c->bytes_dense = 0;
    c->peer_bytes.assign(c->world, 0);
    for (uint32_t p = 0; p < c->world && ok; ++p) {
        size_t off = 0, bytes = 0;
        (void)slab_bytes(c, w, h, p, &off, &bytes);
        if (!bytes) continue;
        if (c->rank == 0 && p == 0)
            ok = hip_ok(hipMemcpyAsync(static_cast<uint8_t*>(out) + off, static_cast<const uint8_t*>(pixels) + off, bytes, hipMemcpyDeviceToDevice, c->stream), "hipMemcpyAsync");
        else if (c->rank == 0)
            ok = nccl_ok(api->Recv(static_cast<uint8_t*>(out) + off, bytes, ncclUint8, (int)p, c->nccl, c->stream), "ncclRecv");
        else if (p == c->rank) {
            ok = nccl_ok(api->Send(static_cast<const uint8_t*>(pixels) + off, bytes, ncclUint8, 0, c->nccl, c->stream), "ncclSend");
            c->bytes_sent = c->bytes_dense = bytes;
        }
    }
    ok = nccl_ok(api->GroupEnd(), "ncclGroupEnd") && ok;
    if (!ok) return CRH_ERR_HIP;
    end_phase(c, kGather);
    c->timed = true;
    if ((st = crh_internal_frame_touched(layer, c->stream, 0)) != CRH_OK) return st; // the layer's next pass is ordered behind the transfer
    if (result) return crh_internal_frame_touched(result, c->stream, 1);
    return CRH_OK;
}
// ... over a loopback group (one device, one thread): device-to-device copies in place of the transfers
crh_status crh_comm_local_gather_slabs(crh_comm* rank0, crh_frame* const* layers, crh_frame* result) {
    if (!rank0 || !rank0->local_group || rank0->rank != 0 || !layers || !result) return CRH_ERR_INVALID_ARGUMENT;
    std::vector<crh_comm*>& g = *rank0->local_group;
    const uint32_t world = rank0->world;
    uint32_t rw = 0, rh = 0, rformat = 0;
    int device = 0;
    crh_status st = crh_internal_frame_geometry(result, &rw, &rh, &rformat, &device);
    if (st != CRH_OK) return st;
    if (rformat == CRH_FORMAT_RGBA16F || device != rank0->device) return CRH_ERR_INVALID_ARGUMENT;
    std::vector<void*> pixels(world, nullptr);
    for (uint32_t k = 0; k < world; ++k) {
        if (!g[k] || !layers[k]) return CRH_ERR_INVALID_ARGUMENT;
        uint32_t w = 0, h = 0, format = 0;
        if ((st = crh_internal_frame_geometry(layers[k], &w, &h, &format, &device)) != CRH_OK) return st;
        if (w != rw || h != rh || format == CRH_FORMAT_RGBA16F) {
            set_last_error("crh_comm_local_gather_slabs: rank " + std::to_string(k) + " gathers a layer of another size or format");
            return CRH_ERR_INVALID_ARGUMENT;
        }
        if ((st = crh_internal_frame_info(layers[k], &pixels[k], &w, &h, &device)) != CRH_OK) return st;
    }
    void* out = nullptr;
    if ((st = crh_internal_frame_info(result, &out, &rw, &rh, &device)) != CRH_OK) return st;
    HIP_TRY(hipSetDevice(rank0->device));
    for (uint32_t k = 0; k < world; ++k) { // every rank "sends" on its own stream; rank 0's stream is where the result is complete
        crh_comm* c = g[k];
        c->timed = false;
        mark_all_phases(c);
        begin_phase(c, kGather);
        size_t off = 0, bytes = 0;
        (void)slab_bytes(c, rw, rh, k, &off, &bytes);
        if (bytes) HIP_TRY(hipMemcpyAsync(static_cast<uint8_t*>(out) + off, static_cast<const uint8_t*>(pixels[k]) + off, bytes, hipMemcpyDeviceToDevice, c->stream));
        end_phase(c, kGather);
        c->bytes_sent = c->bytes_dense = k ? bytes : 0;
        c->peer_bytes.assign(world, 0);
        c->timed = true;
        HIP_TRY(hipEventRecord(c->composited, c->stream));
        if ((st = crh_internal_frame_touched(layers[k], c->stream, 0)) != CRH_OK) return st;
    }
    for (uint32_t k = 1; k < world; ++k) HIP_TRY(hipStreamWaitEvent(rank0->stream, g[k]->composited, 0));
    return crh_internal_frame_touched(result, rank0->stream, 1);
}

// The same exchange over a loopback group (all communicators on one device, one thread): layers[k] = rank k's layer. Streams wait for
// each other through events where a rank reads what another one produced.
crh_status crh_comm_local_exchange(crh_comm* rank0, crh_frame* const* layers, crh_frame* result) {
    if (!rank0 || !rank0->local_group || rank0->rank != 0 || !layers || !result) return CRH_ERR_INVALID_ARGUMENT;
    std::vector<crh_comm*>& g = *rank0->local_group;
    const uint32_t world = rank0->world;
    for (uint32_t k = 0; k < world; ++k)
        if (!g[k] || !layers[k]) return CRH_ERR_INVALID_ARGUMENT;
    crh_status st;
    // CRH_LOOPBACK_SERIAL=1 (measurement only): every rank's part of a phase runs with the GPU to itself, so that crh_comm_last_timing
    // gives what a rank's kernels cost on a GPU of its own instead of what they cost while the other ranks' run beside them
    static const bool serial = getenv("CRH_LOOPBACK_SERIAL") != nullptr;
#define CRH_SERIAL_POINT(k_) \
    if (serial) HIP_TRY(hipStreamSynchronize(g[k_]->stream))
    for (uint32_t k = 0; k < world; ++k) {
        g[k]->timed = false;
        if (serial && k) HIP_TRY(hipStreamSynchronize(g[k - 1]->stream));
        if ((st = phase_pack(g[k], layers[k])) != CRH_OK) return st;
        HIP_TRY(hipEventRecord(g[k]->packed, g[k]->stream));
    }
    CRH_SERIAL_POINT(world - 1);
    for (uint32_t k = 0; k < world; ++k)
        if (g[k]->n_tiles != g[0]->n_tiles || g[k]->width != g[0]->width || g[k]->height != g[0]->height || g[k]->format != g[0]->format) {
            set_last_error("crh_comm_local_exchange: rank " + std::to_string(k) + " exchanges a layer of another size or format");
            return CRH_ERR_INVALID_ARGUMENT;
        }
    for (uint32_t k = 0; k < world; ++k) { // "all-gather" (every rank's [header | bitmap] was enqueued above), then rank k's plan
        begin_phase(g[k], kPlan);
        for (uint32_t q = 0; q < world; ++q) {
            if (q != k) HIP_TRY(hipStreamWaitEvent(g[k]->stream, g[q]->packed, 0));
            HIP_TRY(hipMemcpyAsync(g[k]->bitmaps_all.as<uint32_t>() + (size_t)q * g[k]->stride(), g[q]->bitmap.p, (size_t)g[k]->stride() * 4, hipMemcpyDeviceToDevice, g[k]->stream));
        }
        if ((st = phase_plan(g[k])) != CRH_OK) return st;
        CRH_SERIAL_POINT(k);
    }
    for (uint32_t k = 0; k < world; ++k) { // "all-to-all": rank k pulls its slab's tiles out of every rank's pack buffer (packed long ago: the plan waited)
        begin_phase(g[k], kAllToAll);
        for (uint32_t q = 0; q < world; ++q) {
            size_t off, bytes;
            segment(g[k], q, k, &off, &bytes);
            if (bytes) HIP_TRY(hipMemcpyAsync(recv_slot(g[k], q), static_cast<uint8_t*>(g[q]->pack.p) + off, bytes, hipMemcpyDeviceToDevice, g[k]->stream));
        }
        end_phase(g[k], kAllToAll);
        CRH_SERIAL_POINT(k);
    }
    for (uint32_t k = 0; k < world; ++k) {
        if ((st = phase_composite(g[k])) != CRH_OK) return st;
        HIP_TRY(hipEventRecord(g[k]->composited, g[k]->stream));
        CRH_SERIAL_POINT(k);
    }
    crh_comm* c = g[0];
    HIP_TRY(c->gathered.ensure((size_t)host_rank(c->or_bits.data(), c->n_tiles) * kResultTileBytes + kResultTileBytes));
    for (uint32_t k = 0; k < world; ++k) begin_phase(g[k], kGather);
    for (uint32_t p = 0; p < world; ++p) { // "gather"
        uint32_t s0, s1;
        slab_tiles(c, p, &s0, &s1);
        const size_t off = (size_t)host_rank(c->or_bits.data(), s0) * kResultTileBytes, bytes = (size_t)host_count(c->or_bits.data(), s0, s1) * kResultTileBytes;
        if (p != 0) HIP_TRY(hipStreamWaitEvent(c->stream, g[p]->composited, 0));
        if (bytes) HIP_TRY(hipMemcpyAsync(static_cast<uint8_t*>(c->gathered.p) + off, g[p]->slab_out.p, bytes, hipMemcpyDeviceToDevice, c->stream));
    }
    for (uint32_t k = 0; k < world; ++k) {
        end_phase(g[k], kGather);
        account(g[k]);
        g[k]->timed = true;
    }
    // the other ranks' pack buffers are read by rank k's stream: the next exchange's packing on rank q must come behind those reads
    for (uint32_t k = 0; k < world; ++k) HIP_TRY(hipEventRecord(g[k]->composited, g[k]->stream));
    for (uint32_t q = 0; q < world; ++q)
        for (uint32_t k = 0; k < world; ++k)
            if (k != q) HIP_TRY(hipStreamWaitEvent(g[q]->stream, g[k]->composited, 0));
    return phase_unpack(c, result);
#undef CRH_SERIAL_POINT
}
}
