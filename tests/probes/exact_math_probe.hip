// TEST INFRASTRUCTURE (tests/test_gpu_exact_math.py). Runs pyrite_amd/csrc/exact_math.h's functions against the compiler's
// correctly rounded ones on the GPU -- the square root and the reciprocal over EVERY float bit pattern -- and prints one JSON line.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "../../pyrite_amd/csrc/exact_math.h"

__global__ void sqrt_probe(unsigned long long* mismatches, uint32_t* lowest, uint32_t* highest) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long mine = 0;
    uint32_t lo = 0xFFFFFFFFu, hi = 0;
    for (uint64_t b = tid; b < (1ull << 32); b += stride) {
        const float x = __uint_as_float((uint32_t)b);
        const float a = sqrtf(x), c = pyr::sqrt32(x);
        if (__float_as_uint(a) != __float_as_uint(c) && !(a != a && c != c)) {
            mine++;
            const uint32_t magnitude = (uint32_t)b & 0x7FFFFFFFu;
            lo = magnitude < lo ? magnitude : lo, hi = magnitude > hi ? magnitude : hi;
        }
    }
    if (mine) {
        atomicAdd(mismatches, mine);
        atomicMin(lowest, lo);
        atomicMax(highest, hi);
    }
}

// rcp32 against 1.0f / x over every bit pattern: mismatches are counted inside [2^-126, 2^126) (must be none) and outside.
__global__ void rcp_probe(unsigned long long* inside, unsigned long long* outside) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long in = 0, out = 0;
    for (uint64_t b = tid; b < (1ull << 32); b += stride) {
        const float x = __uint_as_float((uint32_t)b);
        const float a = 1.0f / x, c = pyr::rcp32(x);
        if (__float_as_uint(a) != __float_as_uint(c) && !(a != a && c != c)) {
            const uint32_t exponent = ((uint32_t)b >> 23) & 0xFFu;
            if (exponent >= 1u && exponent <= 252u) in++;
            else out++;
        }
    }
    if (in) atomicAdd(inside, in);
    if (out) atomicAdd(outside, out);
}

int main() {
    unsigned long long* mismatches = nullptr;
    uint32_t *lowest = nullptr, *highest = nullptr;
    if (hipMalloc(&mismatches, 8) != hipSuccess || hipMalloc(&lowest, 4) != hipSuccess || hipMalloc(&highest, 4) != hipSuccess) return 2;
    if (hipMemset(mismatches, 0, 8) != hipSuccess || hipMemset(lowest, 0xFF, 4) != hipSuccess || hipMemset(highest, 0, 4) != hipSuccess) return 2;
    sqrt_probe<<<4096, 256>>>(mismatches, lowest, highest);
    unsigned long long m = 0;
    uint32_t lo = 0, hi = 0;
    if (hipMemcpy(&m, mismatches, 8, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(&lo, lowest, 4, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(&hi, highest, 4, hipMemcpyDeviceToHost) != hipSuccess)
        return 2;
    float flo, fhi;
    __builtin_memcpy(&flo, &lo, 4);
    __builtin_memcpy(&fhi, &hi, 4);
    unsigned long long *inside = nullptr, *outside = nullptr, rin = 0, rout = 0;
    if (hipMalloc(&inside, 8) != hipSuccess || hipMalloc(&outside, 8) != hipSuccess || hipMemset(inside, 0, 8) != hipSuccess || hipMemset(outside, 0, 8) != hipSuccess) return 2;
    rcp_probe<<<4096, 256>>>(inside, outside);
    if (hipMemcpy(&rin, inside, 8, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(&rout, outside, 8, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    std::printf("{\"sqrt_inputs\": 4294967296, \"sqrt_mismatches\": %llu, \"sqrt_mismatch_lowest_abs\": %.9g, \"sqrt_mismatch_highest_abs\": %.9g, "
                "\"rcp_inputs\": 4294967296, \"rcp_mismatches_in_range\": %llu, \"rcp_mismatches_outside\": %llu}\n",
                m, m ? flo : 0.0f, m ? fhi : 0.0f, rin, rout);
    return 0;
}
