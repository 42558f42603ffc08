// atomics_probe.hip -- developer probe (not part of the library): how fast are the film's scattered f32 atomics when every
// tile is owned by one XCD and the atomics are issued at workgroup scope (executed in that XCD's L2) instead of agent scope
// (executed at the memory side, 32 B of fabric traffic each)?  And do they still add up?
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/atomics_probe tools/atomics_probe.hip && /tmp/atomics_probe
//
// The access pattern is C2's: 1024 tiles of 32 x 32 pixels x 64 bins x {acc, weight}, 4096 chunks of 64 samples per tile,
// 10 exposures per sample to random bins of one random pixel of the tile, two atomics per exposure.
//   mode 0  agent scope, chunk = wave + k * waves (every wave of the GPU in the same tile at the same time): today's kernel
//   mode 1  agent scope, tiles owned by XCDs (tile % 8 == XCC_ID), chunks handed out by a per-XCD cursor
//   mode 2  workgroup scope, same ownership as mode 1
// Values are small integers, so the sums are exact in fp32 and the three films must be equal bit for bit.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

constexpr uint32_t TILES = 1024, CHUNKS = 4096, PIXELS = 1024, BINS = 64, EXPOSURES = 10;

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)); } // HW_REG_XCC_ID[3:0]

template <int MODE>
__device__ __forceinline__ void expose(float* film, uint32_t tile, uint32_t chunk, uint32_t lane) {
    const uint32_t h = hash32(tile * 0x9e3779b9u ^ (chunk * 64 + lane) * 0x85ebca6bu);
    const uint32_t pixel = h & (PIXELS - 1);
    float* base = film + ((size_t)tile * PIXELS + pixel) * BINS * 2;
    uint32_t g = h;
    for (uint32_t k = 0; k < EXPOSURES; ++k) {
        g = hash32(g + k);
        float* cell = base + (g & (BINS - 1)) * 2;
        const float value = (float)((g >> 8) & 7u);
        if (MODE == 2) {
            __hip_atomic_fetch_add(cell, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(cell + 1, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            __hip_atomic_fetch_add(cell, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(cell + 1, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void film_kernel(float* film, uint32_t* cursors, uint32_t spin) {
    const uint32_t lane = threadIdx.x & 63u;
    float sink = 0.0f;
    if (MODE == 0) {
        const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6), waves = gridDim.x * 4;
        for (uint32_t c = wave; c < TILES * CHUNKS; c += waves) {
            for (uint32_t s = 0; s < spin; ++s) sink = sink * 1.0001f + 1.0f; // stands for the path's arithmetic
            expose<MODE>(film, c / CHUNKS, c % CHUNKS, lane);
        }
    } else {
        const uint32_t xcc = xcc_id() & 7u;
        for (;;) {
            uint32_t n = 0;
            if (lane == 0) n = atomicAdd(&cursors[xcc * 32], 1u);
            n = __builtin_amdgcn_readfirstlane(n);
            const uint32_t tile = xcc + 8 * (n / CHUNKS);
            if (tile >= TILES) break;
            for (uint32_t s = 0; s < spin; ++s) sink = sink * 1.0001f + 1.0f;
            expose<MODE>(film, tile, n % CHUNKS, lane);
        }
    }
    if (sink == 123.456f) film[0] = sink;
}

__global__ void xcc_kernel(uint32_t* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}

int main() {
    const size_t cells = (size_t)TILES * PIXELS * BINS * 2;
    float* film[3];
    uint32_t* cursors;
    CHECK(hipMalloc(&cursors, 8 * 32 * 4));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount * 4;
    printf("device %s, %d CUs, grid %d x 256\n", prop.name, prop.multiProcessorCount, grid);

    {
        uint32_t* ids;
        CHECK(hipMalloc(&ids, grid * 4));
        xcc_kernel<<<grid, 64>>>(ids);
        std::vector<uint32_t> h(grid);
        CHECK(hipMemcpy(h.data(), ids, grid * 4, hipMemcpyDeviceToHost));
        int round_robin = 0;
        uint32_t seen = 0;
        for (int i = 0; i < grid; ++i) {
            round_robin += (h[i] & 15u) == (uint32_t)(i % 8);
            seen |= 1u << (h[i] & 15u);
        }
        printf("XCC_ID: raw of blocks 0..15:");
        for (int i = 0; i < 16; ++i) printf(" %u", h[i]);
        printf("\nXCC_ID == blockIdx %% 8 for %d of %d workgroups; ids seen mask 0x%x\n", round_robin, grid, seen);
    }

    for (uint32_t spin : {0u, 2000u}) {
        for (int mode = 0; mode < 3; ++mode) {
            if (spin == 0) CHECK(hipMalloc(&film[mode], cells * 4));
            float best = 1e30f;
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipMemset(film[mode], 0, cells * 4));
                CHECK(hipMemset(cursors, 0, 8 * 32 * 4));
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(a));
                if (mode == 0) film_kernel<0><<<grid, 256>>>(film[mode], cursors, spin);
                if (mode == 1) film_kernel<1><<<grid, 256>>>(film[mode], cursors, spin);
                if (mode == 2) film_kernel<2><<<grid, 256>>>(film[mode], cursors, spin);
                CHECK(hipEventRecord(b));
                CHECK(hipEventSynchronize(b));
                float ms;
                CHECK(hipEventElapsedTime(&ms, a, b));
                best = ms < best ? ms : best;
            }
            const double atomics = 2.0 * EXPOSURES * 64.0 * CHUNKS * TILES;
            printf("spin %4u mode %d: %8.2f ms  %6.1f G atomics/s\n", spin, mode, best, atomics / best / 1e6);
        }
    }
    // equality of the three films
    std::vector<float> h0(cells), h1(cells);
    CHECK(hipMemcpy(h0.data(), film[0], cells * 4, hipMemcpyDeviceToHost));
    for (int mode = 1; mode < 3; ++mode) {
        CHECK(hipMemcpy(h1.data(), film[mode], cells * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        double w0 = 0, w1 = 0;
        for (size_t i = 0; i < cells; ++i) {
            bad += h0[i] != h1[i];
            if (i & 1) w0 += h0[i], w1 += h1[i];
        }
        printf("film of mode %d vs mode 0: %zu cells differ; total weight %.0f vs %.0f (expected %.0f)\n", mode, bad, w1, w0,
               (double)EXPOSURES * 64.0 * CHUNKS * TILES);
    }
    return 0;
}
