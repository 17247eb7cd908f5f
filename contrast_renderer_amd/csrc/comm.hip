// csrc/comm.hip — the exchange step of the path-sharded renderer behind the C ABI (the reference has no multi-GPU
// code, component 18 of SURVEY.md §2 is absent upstream).
//
// One process per GPU. Rank g renders the contiguous Shape range crh_comm_shard() gives it into a private full-size premultiplied RGBA8
// layer; this file turns the `world` layers into one image on rank 0:
//   1. occupancy   a bitmap of the layer's 16x16 tiles that hold anything, and the non-empty tiles packed in tile order (1 KiB each);
//   2. all-gather  of the bitmaps (n_tiles / 8 bytes per rank) — afterwards every rank can compute every transfer size on the host;
//   3. all-to-all  the frame is cut into `world` slabs of tile rows; rank r receives the non-empty tiles of slab r of every layer: one
//                  grouped ncclSend / ncclRecv per peer, so a GPU drives all its xGMI links at once with 1/world of what it drew (xGMI is
//                  point to point: a ring reduction would be per-link bound and world - 1 steps deep, and "over" does not commute);
//   4. composite   rank r blends its slab in rank order — dst = src + dst * (1 - src.a), lower rank underneath — reading only the tiles
//                  that exist, and packs the non-empty result tiles;
//   5. gather      those go to rank 0 (again only non-empty tiles), which unpacks them into the result frame.
// Empty tiles never travel: for the benchmark scene a rank's layer of 1/8 of the Shapes is mostly empty.
// Transports: RCCL (librccl.so is opened on first use, so single-GPU users need no RCCL) and an in-process loopback over several
// communicators of ONE device (crh_comm_create_local / crh_comm_local_exchange), which runs the same kernels and host logic and is what
// the single-GPU tests drive.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>
#include <vector>

#include "../../include/contrast_hip.h"

namespace crh {
void set_last_error(const std::string& text);
}
// internal accessors of api.hip (not part of the public header)
extern "C" crh_status crh_internal_frame_info(crh_frame* f, void** rgba8, uint32_t* width, uint32_t* height, int* device);
extern "C" crh_status crh_internal_frame_written(crh_frame* f);
extern "C" int crh_internal_renderer_device(crh_renderer* r);

namespace {
using crh::set_last_error;

struct Rccl { // the entry points used, resolved with dlsym
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl* rccl() {
    static Rccl api;
    static bool tried = false;
    if (!tried) {
        tried = true;
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.lib) break;
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
#undef CRH_SYM
            if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.GroupStart || !api.GroupEnd || !api.Send || !api.Recv || !api.AllGather) {
                dlclose(api.lib);
                api.lib = nullptr;
            }
        }
    }
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
#define NCCL_TRY(expr)                                                                                                        \
    do {                                                                                                                      \
        const ncclResult_t rc_ = (expr);                                                                                      \
        if (rc_ != ncclSuccess) {                                                                                             \
            set_last_error(std::string(#expr) + ": " + (rccl()->GetErrorString ? rccl()->GetErrorString(rc_) : "RCCL error")); \
            return CRH_ERR_HIP;                                                                                               \
        }                                                                                                                     \
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

constexpr uint32_t kTilePixels = 256; // 16 x 16
constexpr size_t kTileBytes = 1024;

// ---------------------------------------------------------------------------------------------- kernels
// bit t of `bitmap` <=> tile t of the layer holds a non-zero pixel (one workgroup per tile, one pixel per lane)
__global__ __launch_bounds__(256) void k_tile_occupancy(const uint32_t* rgba8, uint32_t width, uint32_t height, uint32_t tiles_x, uint32_t* bitmap) {
    const uint32_t tile = blockIdx.x, tx = tile % tiles_x, ty = tile / tiles_x;
    const uint32_t x = tx * 16u + (threadIdx.x & 15u), y = ty * 16u + (threadIdx.x >> 4);
    const uint32_t v = (x < width && y < height) ? rgba8[(size_t)y * width + x] : 0u;
    if (__syncthreads_or(v != 0u) && threadIdx.x == 0) atomicOr(&bitmap[tile >> 5], 1u << (tile & 31u));
}
// prefix[w] = number of set bits in words [0, w) of one bitmap; one workgroup per bitmap (blockIdx.x), prefix has n_words + 1 entries
__global__ __launch_bounds__(1024) void k_bit_prefix(const uint32_t* bitmaps, uint32_t n_words, uint32_t* prefixes) {
    __shared__ uint32_t partial[1024];
    const uint32_t* bitmap = bitmaps + (size_t)blockIdx.x * n_words;
    uint32_t* prefix = prefixes + (size_t)blockIdx.x * (n_words + 1u);
    const uint32_t per = (n_words + 1023u) / 1024u, begin = threadIdx.x * per, end = min(n_words, begin + per);
    uint32_t sum = 0;
    for (uint32_t w = begin; w < end; ++w) sum += (uint32_t)__popc(bitmap[w]);
    partial[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) { // Hillis-Steele inclusive scan of the 1024 partial sums
        const uint32_t v = threadIdx.x >= d ? partial[threadIdx.x - d] : 0u;
        __syncthreads();
        partial[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = partial[threadIdx.x] - sum;
    for (uint32_t w = begin; w < end; ++w) {
        prefix[w] = run;
        run += (uint32_t)__popc(bitmap[w]);
    }
    if (threadIdx.x == 1023u) prefix[n_words] = partial[1023];
}
__device__ __forceinline__ bool tile_bit(const uint32_t* bitmap, uint32_t tile) { return (bitmap[tile >> 5] >> (tile & 31u)) & 1u; }
__device__ __forceinline__ uint32_t tile_rank(const uint32_t* bitmap, const uint32_t* prefix, uint32_t tile) { // set bits below `tile`
    return prefix[tile >> 5] + (uint32_t)__popc(bitmap[tile >> 5] & ((1u << (tile & 31u)) - 1u));
}
// the non-empty tiles of the layer, packed in tile order: 256 pixels (row-major inside the tile) per tile
__global__ __launch_bounds__(256) void k_pack_tiles(const uint32_t* rgba8, uint32_t width, uint32_t height, uint32_t tiles_x, const uint32_t* bitmap,
                                                    const uint32_t* prefix, uint32_t* pack) {
    const uint32_t tile = blockIdx.x;
    if (!tile_bit(bitmap, tile)) return;
    const uint32_t tx = tile % tiles_x, ty = tile / tiles_x;
    const uint32_t x = tx * 16u + (threadIdx.x & 15u), y = ty * 16u + (threadIdx.x >> 4);
    pack[(size_t)tile_rank(bitmap, prefix, tile) * kTilePixels + threadIdx.x] = (x < width && y < height) ? rgba8[(size_t)y * width + x] : 0u;
}
// Rank r's slab = tiles [slab_begin, slab_end): ordered premultiplied "over" of the `world` layers (layer k = recv[k], the non-empty tiles
// of this slab of rank k's layer in tile order), written as the non-empty tiles of the result in tile order. f32 accumulation, one
// RGBA8 quantisation — the arithmetic of k_composite (raster.hip); a tile a layer does not have is a transparent layer (exact).
struct CompositeJob {
    const uint32_t* bitmaps;  // [world][n_words]
    const uint32_t* prefixes; // [world][n_words + 1]
    const uint32_t* const* recv; // [world]
    const uint32_t* or_bitmap;
    const uint32_t* or_prefix;
    uint32_t world, n_words, slab_begin, slab_end;
    uint32_t* out; // packed result tiles of the slab
};
__global__ __launch_bounds__(256) void k_composite_tiles(CompositeJob j) {
    const uint32_t tile = j.slab_begin + blockIdx.x;
    if (!tile_bit(j.or_bitmap, tile)) return;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (uint32_t k = 0; k < j.world; ++k) {
        const uint32_t* bitmap = j.bitmaps + (size_t)k * j.n_words;
        if (!tile_bit(bitmap, tile)) continue;
        const uint32_t* prefix = j.prefixes + (size_t)k * (j.n_words + 1u);
        const uint32_t slot = tile_rank(bitmap, prefix, tile) - tile_rank(bitmap, prefix, j.slab_begin);
        const uint32_t p = j.recv[k][(size_t)slot * kTilePixels + threadIdx.x];
        const float sr[4] = {(float)(p & 255u) * (1.0f / 255.0f), (float)((p >> 8) & 255u) * (1.0f / 255.0f), (float)((p >> 16) & 255u) * (1.0f / 255.0f),
                             (float)(p >> 24) * (1.0f / 255.0f)};
        const float keep = 1.0f - sr[3];
        for (int c = 0; c < 4; ++c) acc[c] = sr[c] + acc[c] * keep;
    }
    uint32_t packed = 0;
    for (int c = 0; c < 4; ++c) {
        const float x = acc[c] < 0.0f ? 0.0f : (acc[c] > 1.0f ? 1.0f : acc[c]);
        packed |= (uint32_t)(int)(x * 255.0f + 0.5f) << (8 * c);
    }
    const uint32_t slot = tile_rank(j.or_bitmap, j.or_prefix, tile) - tile_rank(j.or_bitmap, j.or_prefix, j.slab_begin);
    j.out[(size_t)slot * kTilePixels + threadIdx.x] = packed;
}
__global__ __launch_bounds__(256) void k_or_bitmaps(const uint32_t* bitmaps, uint32_t world, uint32_t n_words, uint32_t* out) {
    const uint32_t w = blockIdx.x * 256u + threadIdx.x;
    if (w >= n_words) return;
    uint32_t v = 0;
    for (uint32_t k = 0; k < world; ++k) v |= bitmaps[(size_t)k * n_words + w];
    out[w] = v;
}
// rank 0: the gathered tiles -> the result frame (tiles nobody drew are cleared)
__global__ __launch_bounds__(256) void k_unpack_tiles(uint32_t* rgba8, uint32_t width, uint32_t height, uint32_t tiles_x, const uint32_t* bitmap, const uint32_t* prefix,
                                                      const uint32_t* pack) {
    const uint32_t tile = blockIdx.x, tx = tile % tiles_x, ty = tile / tiles_x;
    const uint32_t x = tx * 16u + (threadIdx.x & 15u), y = ty * 16u + (threadIdx.x >> 4);
    if (x >= width || y >= height) return;
    rgba8[(size_t)y * width + x] = tile_bit(bitmap, tile) ? pack[(size_t)tile_rank(bitmap, prefix, tile) * kTilePixels + threadIdx.x] : 0u;
}
} // namespace

struct crh_comm {
    crh_renderer* renderer = nullptr;
    int device = 0;
    uint32_t rank = 0, world = 1;
    ncclComm_t nccl = nullptr; // RCCL transport
    std::vector<crh_comm*>* local_group = nullptr; // loopback transport: the communicators of the group, by rank (owned by rank 0's)
    hipStream_t stream = nullptr;
    // geometry of the last exchange
    uint32_t width = 0, height = 0, tiles_x = 0, tiles_y = 0, n_tiles = 0, n_words = 0;
    Buf bitmap, prefix, pack;        // this rank's layer
    Buf bitmaps_all, prefixes_all;   // every rank's bitmap (all-gather) and their prefix sums
    Buf or_bitmap, or_prefix;        // union: the tiles of the composited image
    Buf recv, recv_table;            // received slab tiles, [world] pointers into `recv`
    Buf slab_out, gathered;          // this rank's composited slab (packed); rank 0: all slabs (packed, tile order)
    std::vector<uint32_t> host_bitmaps; // [world][n_words]
    // statistics of the last exchange (crh_comm_last_traffic)
    uint64_t bytes_sent = 0, bytes_dense = 0;
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

// phase 1: occupancy bitmap, its prefix sums and the packed tiles of this rank's layer (all on the communicator's stream)
crh_status phase_pack(crh_comm* c, crh_frame* layer) {
    void* pixels = nullptr;
    uint32_t w = 0, h = 0;
    int device = 0;
    crh_status st = crh_internal_frame_info(layer, &pixels, &w, &h, &device); // settles the frame: its pixels are final and visible
    if (st != CRH_OK) return st;
    if (device != c->device) return CRH_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(c->device));
    c->width = w, c->height = h;
    c->tiles_x = (w + 15u) / 16u, c->tiles_y = (h + 15u) / 16u, c->n_tiles = c->tiles_x * c->tiles_y;
    c->n_words = (c->n_tiles + 31u) / 32u;
    HIP_TRY(c->bitmap.ensure((size_t)c->n_words * 4));
    HIP_TRY(c->prefix.ensure((size_t)(c->n_words + 1) * 4));
    HIP_TRY(c->pack.ensure((size_t)c->n_tiles * kTileBytes));
    HIP_TRY(c->bitmaps_all.ensure((size_t)c->world * c->n_words * 4));
    HIP_TRY(c->prefixes_all.ensure((size_t)c->world * (c->n_words + 1) * 4));
    HIP_TRY(c->or_bitmap.ensure((size_t)c->n_words * 4));
    HIP_TRY(c->or_prefix.ensure((size_t)(c->n_words + 1) * 4));
    HIP_TRY(hipMemsetAsync(c->bitmap.p, 0, (size_t)c->n_words * 4, c->stream));
    hipLaunchKernelGGL(k_tile_occupancy, dim3(c->n_tiles), dim3(256), 0, c->stream, static_cast<const uint32_t*>(pixels), w, h, c->tiles_x, c->bitmap.as<uint32_t>());
    hipLaunchKernelGGL(k_bit_prefix, dim3(1), dim3(1024), 0, c->stream, c->bitmap.as<uint32_t>(), c->n_words, c->prefix.as<uint32_t>());
    hipLaunchKernelGGL(k_pack_tiles, dim3(c->n_tiles), dim3(256), 0, c->stream, static_cast<const uint32_t*>(pixels), w, h, c->tiles_x, c->bitmap.as<uint32_t>(),
                       c->prefix.as<uint32_t>(), c->pack.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    return CRH_OK;
}
// phase 2 (after the bitmaps of all ranks are in bitmaps_all): prefix sums, union, host copy of the bitmaps, receive buffers
crh_status phase_plan(crh_comm* c) {
    hipLaunchKernelGGL(k_bit_prefix, dim3(c->world), dim3(1024), 0, c->stream, c->bitmaps_all.as<uint32_t>(), c->n_words, c->prefixes_all.as<uint32_t>());
    hipLaunchKernelGGL(k_or_bitmaps, dim3((c->n_words + 255u) / 256u), dim3(256), 0, c->stream, c->bitmaps_all.as<uint32_t>(), c->world, c->n_words, c->or_bitmap.as<uint32_t>());
    hipLaunchKernelGGL(k_bit_prefix, dim3(1), dim3(1024), 0, c->stream, c->or_bitmap.as<uint32_t>(), c->n_words, c->or_prefix.as<uint32_t>());
    c->host_bitmaps.resize((size_t)c->world * c->n_words);
    HIP_TRY(hipMemcpyAsync(c->host_bitmaps.data(), c->bitmaps_all.p, c->host_bitmaps.size() * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream)); // the one host synchronisation of the exchange: every transfer size follows from the bitmaps
    uint32_t s0, s1;
    slab_tiles(c, c->rank, &s0, &s1);
    size_t total = 0;
    std::vector<const uint32_t*> table(c->world);
    std::vector<size_t> offset(c->world);
    for (uint32_t k = 0; k < c->world; ++k) {
        offset[k] = total;
        total += (size_t)host_count(&c->host_bitmaps[(size_t)k * c->n_words], s0, s1) * kTileBytes;
    }
    HIP_TRY(c->recv.ensure(total + kTileBytes));
    HIP_TRY(c->recv_table.ensure(sizeof(void*) * c->world));
    for (uint32_t k = 0; k < c->world; ++k) table[k] = reinterpret_cast<const uint32_t*>(static_cast<uint8_t*>(c->recv.p) + offset[k]);
    HIP_TRY(hipMemcpyAsync(c->recv_table.p, table.data(), sizeof(void*) * c->world, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream)); // `table` goes out of scope
    return CRH_OK;
}
// where rank k's tiles of slab r start inside rank k's pack buffer / how many bytes they are
void segment(const crh_comm* c, uint32_t k, uint32_t r, size_t* offset, size_t* bytes) {
    uint32_t s0, s1;
    slab_tiles(c, r, &s0, &s1);
    const uint32_t* bitmap = &c->host_bitmaps[(size_t)k * c->n_words];
    *offset = (size_t)host_rank(bitmap, s0) * kTileBytes;
    *bytes = (size_t)host_count(bitmap, s0, s1) * kTileBytes;
}
uint8_t* recv_slot(const crh_comm* c, uint32_t k) { // where layer k's tiles of my slab are received
    size_t offset = 0;
    uint32_t s0, s1;
    slab_tiles(c, c->rank, &s0, &s1);
    for (uint32_t q = 0; q < k; ++q) offset += (size_t)host_count(&c->host_bitmaps[(size_t)q * c->n_words], s0, s1) * kTileBytes;
    return static_cast<uint8_t*>(c->recv.p) + offset;
}
// phase 4: composite my slab (after the slab tiles of every layer are in `recv`)
crh_status phase_composite(crh_comm* c) {
    uint32_t s0, s1;
    slab_tiles(c, c->rank, &s0, &s1);
    std::vector<uint32_t> or_bits(c->n_words, 0u);
    for (uint32_t k = 0; k < c->world; ++k)
        for (uint32_t w = 0; w < c->n_words; ++w) or_bits[w] |= c->host_bitmaps[(size_t)k * c->n_words + w];
    const size_t out_tiles = host_count(or_bits.data(), s0, s1);
    HIP_TRY(c->slab_out.ensure(out_tiles * kTileBytes + kTileBytes));
    if (s1 > s0) {
        CompositeJob j;
        j.bitmaps = c->bitmaps_all.as<uint32_t>(), j.prefixes = c->prefixes_all.as<uint32_t>();
        j.recv = static_cast<const uint32_t* const*>(c->recv_table.p);
        j.or_bitmap = c->or_bitmap.as<uint32_t>(), j.or_prefix = c->or_prefix.as<uint32_t>();
        j.world = c->world, j.n_words = c->n_words, j.slab_begin = s0, j.slab_end = s1;
        j.out = c->slab_out.as<uint32_t>();
        hipLaunchKernelGGL(k_composite_tiles, dim3(s1 - s0), dim3(256), 0, c->stream, j);
    }
    HIP_TRY(hipGetLastError());
    return CRH_OK;
}
// phase 6 (rank 0, after every slab's tiles are in `gathered`): unpack into the result frame
crh_status phase_unpack(crh_comm* c, crh_frame* result) {
    void* pixels = nullptr;
    uint32_t w = 0, h = 0;
    int device = 0;
    crh_status st = crh_internal_frame_info(result, &pixels, &w, &h, &device);
    if (st != CRH_OK) return st;
    if (w != c->width || h != c->height || device != c->device) return CRH_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(k_unpack_tiles, dim3(c->n_tiles), dim3(256), 0, c->stream, static_cast<uint32_t*>(pixels), w, h, c->tiles_x, c->or_bitmap.as<uint32_t>(),
                       c->or_prefix.as<uint32_t>(), c->gathered.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    return crh_internal_frame_written(result);
}
void account(crh_comm* c) { // what this rank put on the wire vs. what dense slabs would have cost
    c->bytes_sent = 0;
    for (uint32_t p = 0; p < c->world; ++p) {
        if (p == c->rank) continue;
        size_t off, bytes;
        segment(c, c->rank, p, &off, &bytes);
        c->bytes_sent += bytes;
    }
    uint32_t s0, s1;
    slab_tiles(c, c->rank, &s0, &s1);
    std::vector<uint32_t> or_bits(c->n_words, 0u);
    for (uint32_t k = 0; k < c->world; ++k)
        for (uint32_t w = 0; w < c->n_words; ++w) or_bits[w] |= c->host_bitmaps[(size_t)k * c->n_words + w];
    if (c->rank != 0) c->bytes_sent += (size_t)host_count(or_bits.data(), s0, s1) * kTileBytes;
    c->bytes_dense = (uint64_t)(c->n_tiles - (s1 - s0)) * kTileBytes + (c->rank != 0 ? (uint64_t)(s1 - s0) * kTileBytes : 0ull);
}
crh_status create_common(crh_renderer* r, uint32_t rank, uint32_t world, crh_comm** out) {
    if (!r || !out || world == 0 || rank >= world) return CRH_ERR_INVALID_ARGUMENT;
    crh_comm* c = new crh_comm;
    c->renderer = r;
    c->device = crh_internal_renderer_device(r);
    c->rank = rank, c->world = world;
    if (!hip_ok(hipSetDevice(c->device), "hipSetDevice") || !hip_ok(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking), "hipStreamCreate")) {
        delete c;
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
    const ncclResult_t rc = api->CommInitRank(&c->nccl, (int)world, id, (int)rank);
    if (rc != ncclSuccess) {
        set_last_error(std::string("ncclCommInitRank: ") + (api->GetErrorString ? api->GetErrorString(rc) : "RCCL error"));
        (void)hipStreamDestroy(c->stream);
        delete c;
        return CRH_ERR_HIP;
    }
    *out = c;
    return CRH_OK;
}
crh_status crh_comm_create_local(crh_renderer* r, uint32_t rank, uint32_t world, crh_comm* rank0, crh_comm** out) {
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
    (void)hipStreamSynchronize(c->stream);
    if (c->nccl && rccl()) (void)rccl()->CommDestroy(c->nccl);
    if (c->local_group) {
        (*c->local_group)[c->rank] = nullptr;
        bool empty = true;
        for (crh_comm* m : *c->local_group) empty = empty && m == nullptr;
        if (empty) delete c->local_group;
    }
    for (Buf* b : {&c->bitmap, &c->prefix, &c->pack, &c->bitmaps_all, &c->prefixes_all, &c->or_bitmap, &c->or_prefix, &c->recv, &c->recv_table, &c->slab_out, &c->gathered}) b->release();
    (void)hipStreamDestroy(c->stream);
    delete c;
}
crh_status crh_comm_last_traffic(const crh_comm* c, uint64_t* bytes_sent, uint64_t* bytes_dense) {
    if (!c) return CRH_ERR_INVALID_ARGUMENT;
    if (bytes_sent) *bytes_sent = c->bytes_sent;
    if (bytes_dense) *bytes_dense = c->bytes_dense;
    return CRH_OK;
}

// The collective: every rank calls it with its layer; rank 0 also passes the frame that receives the image (the others pass NULL).
crh_status crh_frame_exchange(crh_comm* c, crh_frame* layer, crh_frame* result) {
    if (!c || !layer || !c->nccl || (c->rank == 0) != (result != nullptr)) return CRH_ERR_INVALID_ARGUMENT;
    Rccl* api = rccl();
    crh_status st = phase_pack(c, layer);
    if (st != CRH_OK) return st;
    NCCL_TRY(api->AllGather(c->bitmap.p, c->bitmaps_all.p, (size_t)c->n_words * 4, ncclUint8, c->nccl, c->stream));
    if ((st = phase_plan(c)) != CRH_OK) return st;
    // all-to-all of the slab tiles: one group, so that all links are driven at once
    NCCL_TRY(api->GroupStart());
    for (uint32_t p = 0; p < c->world; ++p) {
        size_t off, bytes;
        segment(c, c->rank, p, &off, &bytes); // my tiles of slab p
        if (p == c->rank) {
            if (bytes) HIP_TRY(hipMemcpyAsync(recv_slot(c, c->rank), static_cast<uint8_t*>(c->pack.p) + off, bytes, hipMemcpyDeviceToDevice, c->stream));
            continue;
        }
        if (bytes) NCCL_TRY(api->Send(static_cast<uint8_t*>(c->pack.p) + off, bytes, ncclUint8, (int)p, c->nccl, c->stream));
        segment(c, p, c->rank, &off, &bytes); // rank p's tiles of my slab
        if (bytes) NCCL_TRY(api->Recv(recv_slot(c, p), bytes, ncclUint8, (int)p, c->nccl, c->stream));
    }
    NCCL_TRY(api->GroupEnd());
    if ((st = phase_composite(c)) != CRH_OK) return st;
    // gather of the composited slabs' non-empty tiles on rank 0 (tile order = slab order)
    std::vector<uint32_t> or_bits(c->n_words, 0u);
    for (uint32_t k = 0; k < c->world; ++k)
        for (uint32_t w = 0; w < c->n_words; ++w) or_bits[w] |= c->host_bitmaps[(size_t)k * c->n_words + w];
    if (c->rank == 0) HIP_TRY(c->gathered.ensure((size_t)host_rank(or_bits.data(), c->n_tiles) * kTileBytes + kTileBytes));
    NCCL_TRY(api->GroupStart());
    for (uint32_t p = 0; p < c->world; ++p) {
        uint32_t s0, s1;
        slab_tiles(c, p, &s0, &s1);
        const size_t off = (size_t)host_rank(or_bits.data(), s0) * kTileBytes, bytes = (size_t)host_count(or_bits.data(), s0, s1) * kTileBytes;
        if (!bytes) continue;
        if (c->rank == 0 && p == 0)
            HIP_TRY(hipMemcpyAsync(static_cast<uint8_t*>(c->gathered.p) + off, c->slab_out.p, bytes, hipMemcpyDeviceToDevice, c->stream));
        else if (c->rank == 0)
            NCCL_TRY(api->Recv(static_cast<uint8_t*>(c->gathered.p) + off, bytes, ncclUint8, (int)p, c->nccl, c->stream));
        else if (p == c->rank)
            NCCL_TRY(api->Send(c->slab_out.p, bytes, ncclUint8, 0, c->nccl, c->stream));
    }
    NCCL_TRY(api->GroupEnd());
    account(c);
    if (c->rank == 0) return phase_unpack(c, result);
    HIP_TRY(hipStreamSynchronize(c->stream)); // the layer and the buffers may be reused
    return CRH_OK;
}

// The same exchange over a loopback group (all communicators on one device, one thread): layers[k] = rank k's layer.
crh_status crh_comm_local_exchange(crh_comm* rank0, crh_frame* const* layers, crh_frame* result) {
    if (!rank0 || !rank0->local_group || rank0->rank != 0 || !layers || !result) return CRH_ERR_INVALID_ARGUMENT;
    std::vector<crh_comm*>& g = *rank0->local_group;
    const uint32_t world = rank0->world;
    for (uint32_t k = 0; k < world; ++k)
        if (!g[k] || !layers[k]) return CRH_ERR_INVALID_ARGUMENT;
    crh_status st;
    for (uint32_t k = 0; k < world; ++k)
        if ((st = phase_pack(g[k], layers[k])) != CRH_OK) return st;
    for (uint32_t k = 0; k < world; ++k) HIP_TRY(hipStreamSynchronize(g[k]->stream));
    for (uint32_t k = 0; k < world; ++k) { // "all-gather"
        if (g[k]->n_tiles != g[0]->n_tiles || g[k]->width != g[0]->width) return CRH_ERR_INVALID_ARGUMENT;
        for (uint32_t q = 0; q < world; ++q)
            HIP_TRY(hipMemcpyAsync(g[k]->bitmaps_all.as<uint32_t>() + (size_t)q * g[k]->n_words, g[q]->bitmap.p, (size_t)g[k]->n_words * 4, hipMemcpyDeviceToDevice, g[k]->stream));
    }
    for (uint32_t k = 0; k < world; ++k)
        if ((st = phase_plan(g[k])) != CRH_OK) return st;
    for (uint32_t k = 0; k < world; ++k) // "all-to-all": rank k pulls its slab's tiles out of every rank's pack buffer
        for (uint32_t q = 0; q < world; ++q) {
            size_t off, bytes;
            segment(g[k], q, k, &off, &bytes);
            if (bytes) HIP_TRY(hipMemcpyAsync(recv_slot(g[k], q), static_cast<uint8_t*>(g[q]->pack.p) + off, bytes, hipMemcpyDeviceToDevice, g[k]->stream));
        }
    for (uint32_t k = 0; k < world; ++k)
        if ((st = phase_composite(g[k])) != CRH_OK) return st;
    for (uint32_t k = 0; k < world; ++k) HIP_TRY(hipStreamSynchronize(g[k]->stream));
    crh_comm* c = g[0];
    std::vector<uint32_t> or_bits(c->n_words, 0u);
    for (uint32_t k = 0; k < world; ++k)
        for (uint32_t w = 0; w < c->n_words; ++w) or_bits[w] |= c->host_bitmaps[(size_t)k * c->n_words + w];
    HIP_TRY(c->gathered.ensure((size_t)host_rank(or_bits.data(), c->n_tiles) * kTileBytes + kTileBytes));
    for (uint32_t p = 0; p < world; ++p) { // "gather"
        uint32_t s0, s1;
        slab_tiles(c, p, &s0, &s1);
        const size_t off = (size_t)host_rank(or_bits.data(), s0) * kTileBytes, bytes = (size_t)host_count(or_bits.data(), s0, s1) * kTileBytes;
        if (bytes) HIP_TRY(hipMemcpyAsync(static_cast<uint8_t*>(c->gathered.p) + off, g[p]->slab_out.p, bytes, hipMemcpyDeviceToDevice, c->stream));
    }
    for (uint32_t k = 0; k < world; ++k) account(g[k]);
    return phase_unpack(c, result);
}
}
