// Developer tool: is the short correctly-rounded square root (v_sqrt_f32 + the two neighbour residuals, no denormal scaling, no
// class fix-up) equal to sqrtf for EVERY float? Exhaustive over all 2^32 bit patterns; prints the ranges where it is not.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cmath>
__device__ __forceinline__ float sqrt_short(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
    const float rm = __builtin_fmaf(-sm, s, x);
    const float rp = __builtin_fmaf(-sp, s, x);
    float r = (0.0f >= rm) ? sm : s;
    r = (0.0f < rp) ? sp : r;
    return r;
}
__global__ void probe(unsigned long long* mismatches, uint32_t* lowest, uint32_t* highest, uint32_t* first_examples) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b = tid; b < (1ull << 32); b += stride) {
        const float x = __uint_as_float((uint32_t)b);
        const float a = sqrtf(x), c = sqrt_short(x);
        const bool same = (__float_as_uint(a) == __float_as_uint(c)) || (a != a && c != c);
        if (!same) {
            unsigned long long k = atomicAdd(mismatches, 1ull);
            atomicMin(lowest, (uint32_t)b & 0x7FFFFFFFu);
            atomicMax(highest, (uint32_t)b & 0x7FFFFFFFu);
            if (k < 8) first_examples[k] = (uint32_t)b;
        }
    }
}
int main() {
    unsigned long long* m; uint32_t *lo, *hi, *ex;
    hipMalloc(&m, 8); hipMalloc(&lo, 4); hipMalloc(&hi, 4); hipMalloc(&ex, 32);
    hipMemset(m, 0, 8); hipMemset(lo, 0xFF, 4); hipMemset(hi, 0, 4); hipMemset(ex, 0, 32);
    probe<<<4096, 256>>>(m, lo, hi, ex);
    unsigned long long hm; uint32_t hlo, hhi, hex[8];
    hipMemcpy(&hm, m, 8, hipMemcpyDeviceToHost); hipMemcpy(&hlo, lo, 4, hipMemcpyDeviceToHost); hipMemcpy(&hhi, hi, 4, hipMemcpyDeviceToHost); hipMemcpy(hex, ex, 32, hipMemcpyDeviceToHost);
    printf("mismatches %llu of 2^32; |x| bit patterns from %#x (%g) to %#x (%g)\n", hm, hlo, *(float*)&hlo, hhi, *(float*)&hhi);
    for (int i = 0; i < 8 && i < (int)hm; ++i) printf("  example %#x = %g\n", hex[i], *(float*)&hex[i]);
    return 0;
}
