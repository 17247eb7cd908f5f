// tools/valu_rate.hip — issue-rate microbenchmark for the VALU classes the raster kernels are made of (the
// "4 cycles per wave64 VALU instruction" of bench.py was never measured on the box).
//
// For every instruction class a wavefront runs LOOPS x 64 instructions over eight independent accumulators (inline asm, so the compiler
// neither fuses nor removes them) and stamps s_memtime (= shader cycles, MI355X_MICROARCH.md) before and after. With w wavefronts
// resident per SIMD all running the same stream, a wavefront's duration is w x instructions x (cycles per wave-instruction at the issue
// port); with w = 1 it is instructions x max(issue, dependent latency / 8 chains). Occupancy is forced through LDS: one workgroup of
// 4w wavefronts per CU (w <= 4), or two of 2w (w = 6, 8).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate > profiles/r03_valu_rate.json
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kLoops = 512; // x 64 instructions

enum Class { FMA_F32 = 0, PK_FMA_F32, CMP_CNDMASK, CMP_SGPR, MOV_B32, ADD_U32, ADD_F32, MUL_F32, AND_B32, CNDMASK_ONLY, RASTER_MIX, CMP_VCC, CNDMASK_VCC, SIGN_TRICK, MUL_LO_U32, CVT_I32_F32, RCP_F32, READLANE, S_ADD_U32, S_AND_B64, S_BITREPLICATE, S_BFE_U32, MIX_FMA_SADD, CMP_SAND_CNDMASK, N_CLASSES };
static const char* kNames[N_CLASSES] = {"v_fma_f32",       "v_pk_fma_f32",  "v_cmp_ge_i32+v_cndmask_b32 (pair, per instruction)", "v_cmp_ge_i32 -> sgpr pair", "v_mov_b32", "v_add_u32", "v_add_f32",
                                        "v_mul_f32",       "v_and_b32",     "v_cndmask_b32 (sgpr mask)",
                                        "raster mix: 2 v_pk_fma_f32 + 4 v_cmp_ge_i32 + 4 v_cndmask_b32 + 4 v_add_u32 (per instruction)",
                                        "v_cmp_ge_i32 -> vcc (e32)", "v_cndmask_b32 (vcc, e32)", "v_sub_u32 + v_or_b32 + v_lshrrev_b32 (accept bit without a compare, per instruction)",
                                        "v_mul_lo_u32", "v_cvt_i32_f32", "v_rcp_f32", "v_readlane_b32",
                                        "s_add_u32", "s_and_b64", "s_bitreplicate_b64_b32", "s_bfe_u32", "v_fma_f32 + s_add_u32 alternating (per instruction)",
                                        "v_cmp_ge_i32 -> sgpr pair, s_and_b64 on it, v_cndmask_b32 with it (the raster kernel's accept chain, per instruction)"};
static const int kInstrPerBody[N_CLASSES] = {64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 56, 64, 64, 56, 64, 64, 64, 64, 64, 64, 64, 64, 64, 56};

template <int C>
__global__ __launch_bounds__(1024) void k_rate(unsigned long long* out, float seed, unsigned lds_words) {
    extern __shared__ unsigned lds[];
    if (lds_words == 0xFFFFFFFFu) lds[threadIdx.x] = 1u; // keeps the dynamic LDS allocation alive
    float a[8], b = seed, c = seed * 0.5f;
    f32x2 p[8], pb = {seed, seed}, pc = {c, c};
    int ia[8], ib = (int)seed + threadIdx.x;
    unsigned long long m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned sl[8] = {0, 0, 0, 0, 1, 2, 3, 4};
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed + (float)i, p[i] = f32x2{seed + (float)i, seed - (float)i}, ia[i] = (int)threadIdx.x + i;
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kLoops; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (C == FMA_F32) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                if (C == PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(pb), "v"(pc));
                if (C == CMP_CNDMASK) {
                    if (i & 1)
                        asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(ia[i]) : "v"(ib), "s"(m[(i >> 1) & 3]));
                    else
                        asm volatile("v_cmp_ge_i32 %0, %1, %2" : "=s"(m[(i >> 1) & 3]) : "v"(ia[i]), "v"(ib));
                }
                if (C == CMP_SGPR) asm volatile("v_cmp_ge_i32 %0, %1, %2" : "=s"(m[i & 3]) : "v"(ia[i]), "v"(ib));
                if (C == MOV_B32) asm volatile("v_mov_b32 %0, %1" : "=v"(ia[i]) : "v"(ib));
                if (C == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(ia[i]) : "v"(ib));
                if (C == ADD_F32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (C == MUL_F32) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (C == AND_B32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(ia[i]) : "v"(ib));
                if (C == CNDMASK_ONLY) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(ia[i]) : "v"(ib), "s"(m[i & 3]));
                if (C == CMP_VCC) asm volatile("v_cmp_ge_i32 vcc, %0, %1" : : "v"(ia[i]), "v"(ib) : "vcc");
                if (C == CNDMASK_VCC) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ia[i]) : "v"(ib) : "vcc");
                if (C == SIGN_TRICK && i < 7) { // three instructions per chain step, 21 per row of seven chains
                    if (r % 3 == 0) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(ia[i]) : "v"(ib));
                    if (r % 3 == 1) asm volatile("v_or_b32 %0, %0, %1" : "+v"(ia[i]) : "v"(ib));
                    if (r % 3 == 2) asm volatile("v_lshrrev_b32 %0, 31, %0" : "+v"(ia[i]));
                }
                if (C == MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(ia[i]) : "v"(ib));
                if (C == CVT_I32_F32) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(ia[i]) : "v"(a[i]));
                if (C == RCP_F32) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                if (C == READLANE) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sl[i & 3]) : "v"(ia[i]));
                if (C == S_ADD_U32) asm volatile("s_add_u32 %0, %0, %1" : "+s"(sl[i]) : "s"(lds_words) : "scc");
                if (C == S_AND_B64) asm volatile("s_and_b64 %0, %0, %1" : "+s"(m[i]) : "s"(t0) : "scc");
                if (C == S_BITREPLICATE) asm volatile("s_bitreplicate_b64_b32 %0, %1" : "=s"(m[i]) : "s"(sl[i]));
                if (C == S_BFE_U32) asm volatile("s_bfe_u32 %0, %0, 0x100001" : "+s"(sl[i]) : : "scc");
                if (C == MIX_FMA_SADD) {
                    if (i & 1) asm volatile("s_add_u32 %0, %0, %1" : "+s"(sl[i]) : "s"(lds_words) : "scc");
                    else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
                }
                if (C == CMP_SAND_CNDMASK && i < 7) { // three instructions per chain step, through an SGPR pair
                    if (r % 3 == 0) asm volatile("v_cmp_ge_i32 %0, %1, %2" : "=s"(m[i]) : "v"(ia[i]), "v"(ib));
                    if (r % 3 == 1) asm volatile("s_and_b64 %0, %0, %1" : "+s"(m[i]) : "s"(t0) : "scc");
                    if (r % 3 == 2) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(ia[i]) : "v"(ib), "s"(m[i]));
                }
            }
            if (C == RASTER_MIX && r < 4) { // what one edge entry costs two sample pairs of a lane in k_raster_edges, in its proportions
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[2 * r]) : "v"(pb), "v"(pc));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[2 * r + 1]) : "v"(pb), "v"(pc));
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    asm volatile("v_cmp_ge_i32 %0, %1, %2" : "=s"(m[q]) : "v"(ia[q]), "v"(ib));
                    asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(ia[4 + q]) : "v"(ib), "s"(m[q]));
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(ia[q]) : "v"(ia[4 + q]));
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
    int si = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + p[i][0] + p[i][1], si += ia[i];
    if (s == 12345.678f && si == 42 && (m[0] ^ m[1] ^ m[2] ^ m[3] ^ m[4] ^ m[5] ^ m[6] ^ m[7]) == 7ull && (sl[0] ^ sl[1] ^ sl[2] ^ sl[3] ^ sl[4] ^ sl[5] ^ sl[6] ^ sl[7]) == 9u) out[0] = 0; // keeps the results alive
    if ((threadIdx.x & 63u) == 0u) out[1 + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int C>
static void launch(dim3 grid, dim3 block, size_t lds, unsigned long long* out) {
    hipLaunchKernelGGL(k_rate<C>, grid, block, lds, 0, out, 1.0f, (unsigned)(lds / 4));
}
typedef void (*Launcher)(dim3, dim3, size_t, unsigned long long*);
static Launcher kLaunch[N_CLASSES] = {launch<FMA_F32>, launch<PK_FMA_F32>, launch<CMP_CNDMASK>, launch<CMP_SGPR>, launch<MOV_B32>, launch<ADD_U32>,
                                      launch<ADD_F32>, launch<MUL_F32>,    launch<AND_B32>,     launch<CNDMASK_ONLY>, launch<RASTER_MIX>, launch<CMP_VCC>, launch<CNDMASK_VCC>,
                                      launch<SIGN_TRICK>, launch<MUL_LO_U32>, launch<CVT_I32_F32>, launch<RCP_F32>, launch<READLANE>,
                                      launch<S_ADD_U32>, launch<S_AND_B64>, launch<S_BITREPLICATE>, launch<S_BFE_U32>, launch<MIX_FMA_SADD>, launch<CMP_SAND_CNDMASK>};

#define CHECK(e)                                                                         \
    do {                                                                                 \
        hipError_t err_ = (e);                                                           \
        if (err_ != hipSuccess) {                                                        \
            std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(err_));               \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned long long* out = nullptr;
    const size_t slots = 1 + (size_t)cus * 2 * 16;
    CHECK(hipMalloc(&out, slots * 8));
    unsigned long long* out_big = nullptr;
    CHECK(hipMalloc(&out_big, (1 + (size_t)cus * 8 * 4) * 8));
    // dynamic LDS beyond 64 KiB needs the attribute
#define ALLOW(C) CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rate<C>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024))
    ALLOW(FMA_F32); ALLOW(PK_FMA_F32); ALLOW(CMP_CNDMASK); ALLOW(CMP_SGPR); ALLOW(MOV_B32); ALLOW(ADD_U32); ALLOW(ADD_F32); ALLOW(MUL_F32); ALLOW(AND_B32); ALLOW(CNDMASK_ONLY); ALLOW(RASTER_MIX); ALLOW(CMP_VCC); ALLOW(CNDMASK_VCC); ALLOW(SIGN_TRICK); ALLOW(MUL_LO_U32); ALLOW(CVT_I32_F32); ALLOW(RCP_F32); ALLOW(READLANE); ALLOW(S_ADD_U32); ALLOW(S_AND_B64); ALLOW(S_BITREPLICATE); ALLOW(S_BFE_U32); ALLOW(MIX_FMA_SADD); ALLOW(CMP_SAND_CNDMASK);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int ws[] = {1, 2, 3, 4, 6, 8};
    std::printf("{\n  \"device\": \"%s\", \"gcn_arch\": \"%s\", \"compute_units\": %d, \"clock_rate_khz\": %d,\n", prop.name, prop.gcnArchName, cus, prop.clockRate);
    std::printf("  \"method\": \"per wavefront: s_memtime around %d x body independent-chain inline-asm instructions; w wavefronts per SIMD forced through LDS; cycles per wave-instruction = median duration / (instructions x w)\",\n", kLoops);
    std::printf("  \"classes\": {\n");
    for (int c = 0; c < N_CLASSES; ++c) {
        std::printf("    \"%s\": {", kNames[c]);
        for (size_t wi = 0; wi < sizeof(ws) / sizeof(ws[0]); ++wi) {
            const int w = ws[wi];
            const bool two = w > 4;                       // two workgroups per CU
            const int waves_per_block = two ? 2 * w : 4 * w; // spread over the CU's four SIMDs
            const size_t lds = two ? 70 * 1024 : 100 * 1024;
            const dim3 grid(two ? 2 * cus : cus), block(64 * waves_per_block);
            const size_t n_waves = (size_t)grid.x * waves_per_block;
            CHECK(hipMemset(out, 0, slots * 8));
            kLaunch[c](grid, block, lds, out); // warm-up
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0, 0));
            kLaunch[c](grid, block, lds, out);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipDeviceSynchronize());
            CHECK(hipGetLastError());
            float ms = 0.0f;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> d(n_waves);
            CHECK(hipMemcpy(d.data(), out + 1, n_waves * 8, hipMemcpyDeviceToHost));
            std::sort(d.begin(), d.end());
            const double median = (double)d[d.size() / 2], instr = (double)kLoops * kInstrPerBody[c];
            std::printf("%s\"w%d\": {\"cycles_per_wave_instruction\": %.3f, \"median_wave_cycles\": %.0f, \"kernel_ms\": %.4f}", wi ? ", " : "", w, median / (instr * w), median, ms);
        }
        { // the issue rate of the whole chip: 8 wavefronts per SIMD (no LDS games: 8 x 256 CUs workgroups of 1024 threads... as many as fit), wall clock
            const dim3 grid(8 * cus), block(256);
            kLaunch[c](grid, block, 0, out_big);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0, 0));
            kLaunch[c](grid, block, 0, out_big);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipDeviceSynchronize());
            float ms = 0.0f;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double total = (double)grid.x * 4.0 * kLoops * kInstrPerBody[c], simds = 4.0 * cus;
            std::printf(", \"chip\": {\"kernel_ms\": %.4f, \"wave_instructions\": %.0f, \"cycles_per_wave_instruction_at_%d_MHz\": %.3f}", ms, total, prop.clockRate / 1000, ms * 1e-3 * prop.clockRate * 1e3 * simds / total);
        }
        std::printf("}%s\n", c + 1 < N_CLASSES ? "," : "");
    }
    std::printf("  }\n}\n");
    return 0;
}
