// Developer tool: is a short reciprocal (v_rcp_f32 + Newton steps on fma, no v_div_scale / v_div_fmas / v_div_fixup) equal to the
// IEEE 1.0f / x for EVERY float? Exhaustive over all 2^32 bit patterns, three candidates; prints where each differs.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
__device__ __forceinline__ float rcp_a(float x) { // one Newton step
    const float r = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float rcp_b(float x) { // two Newton steps
    float r = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float rcp_c(float x) { // one Newton step, then the residual of the rounded result decides between it and a neighbour
    float r = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    e = __builtin_fmaf(-x, r, 1.0f); // exact residual 1 - x r (one rounding)
    return __builtin_fmaf(e, r, r);
}
template <int WHICH>
__global__ void probe(unsigned long long* mismatches, uint32_t* lowest, uint32_t* highest, uint32_t* first_examples) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b = tid; b < (1ull << 32); b += stride) {
        const float x = __uint_as_float((uint32_t)b);
        const float a = 1.0f / x, c = WHICH == 0 ? rcp_a(x) : (WHICH == 1 ? rcp_b(x) : rcp_c(x));
        const bool same = (__float_as_uint(a) == __float_as_uint(c)) || (a != a && c != c);
        if (!same) {
            unsigned long long k = atomicAdd(mismatches, 1ull);
            atomicMin(lowest, (uint32_t)b & 0x7FFFFFFFu);
            atomicMax(highest, (uint32_t)b & 0x7FFFFFFFu);
            if (k < 8) first_examples[k] = (uint32_t)b;
        }
    }
}
// mismatches with |x| inside [2^lo_e, 2^hi_e): the clean exponent range
template <int WHICH>
__global__ void by_exponent(unsigned long long* per_exponent) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b = tid; b < (1ull << 32); b += stride) {
        const float x = __uint_as_float((uint32_t)b);
        const float a = 1.0f / x, c = WHICH == 0 ? rcp_a(x) : (WHICH == 1 ? rcp_b(x) : rcp_c(x));
        const bool same = (__float_as_uint(a) == __float_as_uint(c)) || (a != a && c != c);
        if (!same) atomicAdd(&per_exponent[((uint32_t)b >> 23) & 0xFFu], 1ull);
    }
}
template <int WHICH>
void run(const char* name) {
    unsigned long long* m; uint32_t *lo, *hi, *ex; unsigned long long* pe;
    (void)hipMalloc(&m, 8); (void)hipMalloc(&lo, 4); (void)hipMalloc(&hi, 4); (void)hipMalloc(&ex, 32); (void)hipMalloc(&pe, 256 * 8);
    (void)hipMemset(m, 0, 8); (void)hipMemset(lo, 0xFF, 4); (void)hipMemset(hi, 0, 4); (void)hipMemset(ex, 0, 32); (void)hipMemset(pe, 0, 256 * 8);
    probe<WHICH><<<4096, 256>>>(m, lo, hi, ex);
    by_exponent<WHICH><<<4096, 256>>>(pe);
    unsigned long long hm, hpe[256]; uint32_t hlo, hhi, hex[8];
    (void)hipMemcpy(&hm, m, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&hlo, lo, 4, hipMemcpyDeviceToHost); (void)hipMemcpy(&hhi, hi, 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hex, ex, 32, hipMemcpyDeviceToHost); (void)hipMemcpy(hpe, pe, sizeof(hpe), hipMemcpyDeviceToHost);
    printf("%s: mismatches %llu of 2^32", name, hm);
    if (hm) printf("; |x| bit patterns from %#x (%g) to %#x (%g)", hlo, *(float*)&hlo, hhi, *(float*)&hhi);
    printf("\n");
    int first_clean = -1, last_clean = -1;
    for (int e = 0; e < 256; ++e) {
        if (hpe[e] == 0) { if (first_clean < 0) first_clean = e; last_clean = e; }
    }
    printf("   biased exponents with mismatches:");
    for (int e = 0; e < 256; ++e) if (hpe[e]) printf(" %d(%llu)", e, hpe[e]);
    printf("\n");
    for (int i = 0; i < 8 && i < (int)hm; ++i) printf("   example %#x = %g\n", hex[i], *(float*)&hex[i]);
}
int main() {
    run<0>("one Newton step (3 instructions)");
    run<1>("two Newton steps (5 instructions)");
    return 0;
}
