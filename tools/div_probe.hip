// Developer tool: where does the short division (v_rcp_f32, one Newton step on the reciprocal, the quotient and its two residual
// corrections -- the compiler's own sequence without v_div_scale / v_div_fixup) differ from a / b? Random mantissas over a grid
// of exponent pairs; prints mismatch counts per (exponent of a, exponent of b) region.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
__device__ __forceinline__ float div_short(float a, float b) {
    float r = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q = a * r;
    const float e2 = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(e2, r, q);
    const float e3 = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(e3, r, q);
}
__device__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// grid: ea, eb in [1, 254]; for each pair `per` random mantissa/sign pairs
__global__ void probe(unsigned long long* bad /* [256*256] */, uint32_t per, uint32_t seed) {
    const uint32_t ea = blockIdx.x + 1, eb = blockIdx.y + 1;
    unsigned long long local = 0;
    for (uint32_t i = threadIdx.x; i < per; i += blockDim.x) {
        const uint32_t h1 = hash(seed ^ (ea << 24) ^ (eb << 16) ^ i), h2 = hash(h1 ^ 0x9e3779b9u);
        uint32_t ma = h1 & 0x7FFFFFu, mb = h2 & 0x7FFFFFu;
        if ((i & 15u) == 0) ma = 0;            // powers of two
        if ((i & 15u) == 1) mb = 0;
        if ((i & 15u) == 2) mb = 0x7FFFFFu;    // all ones (the hard case of the reciprocal)
        if ((i & 15u) == 3) ma = 0x7FFFFFu;
        const float a = __uint_as_float((h1 & 0x80000000u) | (ea << 23) | ma), b = __uint_as_float((h2 & 0x80000000u) | (eb << 23) | mb);
        const float x = a / b, y = div_short(a, b);
        if (__float_as_uint(x) != __float_as_uint(y) && !(x != x && y != y)) local++;
    }
    if (local) atomicAdd(&bad[ea * 256 + eb], local);
}
int main() {
    unsigned long long* bad;
    (void)hipMalloc(&bad, 256 * 256 * 8);
    (void)hipMemset(bad, 0, 256 * 256 * 8);
    const uint32_t per = 1u << 16;
    probe<<<dim3(254, 254), 256>>>(bad, per, 12345u);
    static unsigned long long h[256 * 256];
    (void)hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long total = 0, cells = 0;
    int min_diff = 1000, max_diff = -1000, lo_ea = 999, hi_ea = 0, lo_eb = 999, hi_eb = 0;
    // the region with NO mismatch: report mismatching cells by exponent difference and by operand exponents
    unsigned long long by_diff[512] = {0};
    for (int ea = 1; ea < 255; ++ea)
        for (int eb = 1; eb < 255; ++eb)
            if (h[ea * 256 + eb]) {
                total += h[ea * 256 + eb]; cells++;
                by_diff[ea - eb + 256] += h[ea * 256 + eb];
            }
    printf("pairs tested %llu, mismatches %llu in %llu exponent cells\n", (unsigned long long)per * 254 * 254, total, cells);
    // clean sub-domain: which (ea, eb) boxes are entirely clean?
    for (int lim = 1; lim < 127; ++lim) {  // |exponent - 127| <= K for both and |ea - eb| <= K
        const int K = 127 - lim;
        unsigned long long in = 0;
        for (int ea = 127 - K; ea <= 127 + K; ++ea)
            for (int eb = 127 - K; eb <= 127 + K; ++eb)
                if (ea >= 1 && ea <= 254 && eb >= 1 && eb <= 254) in += h[ea * 256 + eb];
        if (in == 0) { printf("clean for operand exponents within 2^+-%d of 1 (no condition on the quotient's)\n", K); break; }
    }
    for (int eb = 1; eb < 255; ++eb) { // per denominator exponent: the numerator exponents with mismatches
        int lo = 999, hi = -1, clean_lo = 999, clean_hi = -1;
        for (int ea = 1; ea < 255; ++ea) {
            if (h[ea * 256 + eb]) { lo = ea < lo ? ea : lo; hi = ea > hi ? ea : hi; }
            else { clean_lo = ea < clean_lo ? ea : clean_lo; clean_hi = ea > clean_hi ? ea : clean_hi; }
        }
        printf("eb %d (2^%d): clean ea in [%d, %d] = quotient exponent 2^[%d, %d]%s\n", eb, eb - 127, clean_lo, clean_hi, clean_lo - eb, clean_hi - eb, hi >= 0 ? "" : " (all clean)");
    }
    (void)lo_ea; (void)hi_ea; (void)lo_eb; (void)hi_eb; (void)min_diff; (void)max_diff;
    return 0;
}
