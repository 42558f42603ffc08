// kernels.hip -- CDNA4 (gfx950) kernels for Pyrite's camera-to-light integrator.
//
// What runs here is the whole of render_tile's per-sample loop (pyrite/src/renderer/simple.rs:78-140) and everything it
// calls: sample generation, tracer::trace (tracer.rs:208-345), trace_direct (:347-442), the spectral fold `contribute`
// (renderer/algorithm.rs:14-100, applied online bounce by bounce) and Film::expose (film.rs:89-95).
//
// Execution models, all running the same per-path code and so producing the same film:
//   render_kernel      one persistent launch; a wave walks a strided sequence of 64-iteration chunks, lane = iteration, and
//                      the 64 paths advance bounce by bounce (scenes that fit in LDS: C1, C2)
//   render_kernel_sm   every lane runs its path as a state machine (Walker) and refills itself; the wave votes on which
//                      phase code runs next; traversal is resumable (big scenes: C3-C5, textured scenes)
//   wf_logic_kernel +  the same state machine split at the ray, path state in a pool in HBM (optional)
//   wf_trav_kernel
//   intersect_kernel   World::intersect for ray batches: persistent waves with a segmented work feed
//   develop_kernel     film -> 8-bit sRGB
// Per lane: path state in VGPRs, the S - 1 spectral companions (wavelength / brightness / reflectance) and the traversal
// stack in LDS laid out [entry][lane] (bank-conflict free), BVH nodes fetched as 4 x dwordx4 (64 B, both children's boxes,
// tested with packed fp32), leaf primitives as 3 x dwordx4, one primitive per step. Film exposure is two no-return
// global_atomic_add_f32 per exposure.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <string>
#include <type_traits>

#include "device_scene.h"
#include "exact_math.h"

// This file is compiled three times and in parallel by pyrite_amd/build.py: -DPYR_TU=0 holds every kernel and launcher but the
// interpreter builds of render_kernel_sm, -DPYR_TU=1 holds only those (the heaviest kernels to compile: the interpreter in line)
// behind pick_interp_kernel(), -DPYR_TU=2 their PRODUCT forms behind pick_product_kernel(). Without the macro (-1: tools that compile
// kernels.hip by themselves, and the -DPYR_PHASE_PROFILE builds, whose device-side counters must live in one translation unit)
// everything is in one piece.
#ifndef PYR_TU
#define PYR_TU -1
#endif
#define PYR_TU_MAIN (PYR_TU == 0 || PYR_TU == -1)
#define PYR_TU_INTERP (PYR_TU == 1 || PYR_TU == -1)
#define PYR_TU_PRODUCT (PYR_TU == 2 || PYR_TU == -1)

namespace pyr {

#if PYR_TU_MAIN
namespace {
thread_local std::string g_kernel_error;
}
const char* kernels_last_error() { return g_kernel_error.c_str(); }
#endif

#define DEV __device__ __forceinline__
// The lanes for which p holds, as the mask the comparison produced (HIP's __ballot() goes through an integer and back: a
// v_cndmask and a v_cmp per call).
DEV unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

constexpr int BLOCK = 256;
#ifndef PYR_SM_WAVES
#define PYR_SM_WAVES 4 // waves per SIMD (= workgroups per CU) the stage-scheduled kernel is built for
#endif
constexpr float DIST_EPSILON = 0.0001f; // math.rs:4
constexpr float PI_F = 3.14159265358979323846f;
#define PYR_INF __builtin_huge_valf()

// ------------------------------------------------------------------------------------------------ vectors
struct f3 {
    float x, y, z;
};
DEV f3 mk(float x, float y, float z) { return f3{x, y, z}; }
DEV f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV f3 operator-(f3 a) { return mk(-a.x, -a.y, -a.z); }
DEV f3 operator*(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
DEV float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEV f3 cross(f3 a, f3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
DEV float magnitude(f3 a) { return sqrt32(dot(a, a)); }
DEV f3 normalize_to(f3 a, float m) { return a * (m / magnitude(a)); } // cgmath: v * (m / |v|)
DEV f3 normalize(f3 a) { return a * rcp32(magnitude(a)); } // normalize_to(a, 1.0f): 1.0f / |a| is a reciprocal (exact_math.h)
DEV f3 ld3(const float* p) { return mk(p[0], p[1], p[2]); }

// Transcendentals. The reference calls the platform libm through Rust's f32::sin / cos / acos; libms differ from one
// another by an ulp. Here sin, cos and acos are the single-precision Cephes kernels (Moshier; public domain): Cody-Waite
// reduction by pi/4 in three steps and degree-7/8 minimax polynomials, written as plain f32 operations in a fixed order,
// the same text as in oracle/oracle.cpp. With -ffp-contract=off both sides round identically, so directions agree bit for
// bit, and it is ~5x fewer instructions than the OCML routines (which cost 20 % of the C2 render). Accuracy: < 2 ulp on
// [-2 pi, 2 pi] for sin/cos and on [-1, 1] for acos (tests/test_oracle_kat.py). exp (blackbody only) goes through f64.
DEV float sin32(float xx) {
    float x = fabsf(xx);
    float sign = xx < 0.0f ? -1.0f : 1.0f;
    int j = (int)(1.27323954473516f * x); // 4/pi
    float y = (float)j;
    if (j & 1) {
        j += 1;
        y += 1.0f;
    }
    j &= 7;
    if (j > 3) {
        sign = -sign;
        j -= 4;
    }
    x = ((x - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
    float z = x * x;
    float r;
    if (j == 1 || j == 2)
        r = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
    else
        r = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * x + x;
    return sign * r;
}
DEV float cos32(float xx) {
    float x = fabsf(xx);
    float sign = 1.0f;
    int j = (int)(1.27323954473516f * x);
    float y = (float)j;
    if (j & 1) {
        j += 1;
        y += 1.0f;
    }
    j &= 7;
    if (j > 3) {
        sign = -sign;
        j -= 4;
    }
    if (j > 1) sign = -sign;
    x = ((x - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
    float z = x * x;
    float r;
    if (j == 1 || j == 2)
        r = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * x + x;
    else
        r = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
    return sign * r;
}
DEV float asin32_core(float a) { // asin(a) for 0 <= a <= 1
    if (a < 1.0e-4f) return a;
    float x, z;
    const bool big = a > 0.5f;
    if (big) {
        z = 0.5f * (1.0f - a);
        x = sqrt32(z);
    } else {
        x = a;
        z = x * x;
    }
    float r = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z + 1.6666752422e-1f) * z * x + x;
    if (big) {
        r = r + r;
        r = 1.5707963267948966f - r;
    }
    return r;
}
DEV float acos32(float x) {
    if (x < -0.5f) return 3.14159265358979323846f - 2.0f * asin32_core(sqrt32(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * asin32_core(sqrt32(0.5f * (1.0f - x)));
    float a = asin32_core(fabsf(x));
    return 1.5707963267948966f - (x < 0.0f ? -a : a);
}
DEV float exp32(float x) { return (float)exp((double)x); }

// ------------------------------------------------------------------------------------------------ RNG
// xorshift128 (rand_xorshift 0.3.0) per (seed, tile, iteration); distributions of rand 0.8.5. Same stream layout as
// oracle/oracle.cpp (DESIGN.md "RNG").
struct Rng {
    uint32_t x, y, z, w;
};
DEV uint32_t rng_u32(Rng& r) {
    uint32_t t = r.x ^ (r.x << 11);
    r.x = r.y;
    r.y = r.z;
    r.z = r.w;
    r.w = r.w ^ (r.w >> 19) ^ (t ^ (t >> 8));
    return r.w;
}
DEV uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
DEV Rng rng_seed(uint64_t seed, uint32_t tile, uint64_t iteration) {
    uint64_t counter = ((uint64_t)tile << 40) ^ iteration;
    uint64_t a = splitmix64(seed ^ splitmix64(counter));
    uint64_t b = splitmix64(a);
    Rng r{(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
    if ((r.x | r.y | r.z | r.w) == 0) r.w = 1;
    return r;
}
DEV float rng_f32(Rng& r) { return (float)(rng_u32(r) >> 8) * (1.0f / 16777216.0f); }
// UniformFloat::sample_single. The multiply-add is written with round-to-nearest intrinsics so that it is never fused:
// the sampled wavelength decides the film bin and must equal the oracle's bit for bit.
DEV float rng_range_f32(Rng& r, float low, float high) {
    float scale = __fsub_rn(high, low);
    for (;;) {
        float value0_1 = __uint_as_float((rng_u32(r) >> 9) | 0x3F800000u) - 1.0f;
        float res = __fadd_rn(__fmul_rn(value0_1, scale), low);
        if (res < high) return res;
        scale = __uint_as_float(__float_as_uint(scale) - 1u);
    }
}
// UniformInt<usize>::sample_single: 64-bit draw, widening multiply by `n` (< 2^32), rejection zone.
// (rand's zone rejects up to half the draws and a wave stays in these loops until its unluckiest lane is through: ~5 turns per
// call, a third of a SHADE visit. Written four draws per turn with the state's rotation spelled out -- no register moves per
// draw -- the loops were 2.5 % SLOWER on C3 (520 against 531-535 Msamples/s): four exits per turn cost more than the moves.)
DEV uint32_t rng_range_usize(Rng& r, uint32_t n) {
    uint64_t range = n;
    uint64_t zone = (range << __builtin_clzll(range)) - 1;
    for (;;) {
        uint64_t lo32 = rng_u32(r);
        uint64_t hi32 = rng_u32(r);
        uint64_t p0 = lo32 * range;           // < 2^64
        uint64_t p1 = hi32 * range;           // contributes p1 << 32
        uint64_t lo = p0 + (p1 << 32);        // low 64 bits of v * range
        uint64_t carry = lo < p0 ? 1 : 0;
        uint64_t hi = (p1 >> 32) + carry;     // high 64 bits
        if (lo <= zone) return (uint32_t)hi;
    }
}
// SliceRandom::choose -> gen_range(0..n as u32).
DEV uint32_t rng_choose(Rng& r, uint32_t n) {
    uint32_t zone = (n << __builtin_clz(n)) - 1;
    for (;;) {
        uint32_t v = rng_u32(r);
        uint64_t m = (uint64_t)v * n;
        if ((uint32_t)m <= zone) return (uint32_t)(m >> 32);
    }
}

// ------------------------------------------------------------------------------------------------ counters
struct Counters {
    uint32_t samples, extension_rays, shadow_rays, box_tests, triangle_tests, sphere_tests, plane_tests, shaded_hits, exposures;
};
template <bool COUNT>
DEV void flush_counters(const Counters& c, unsigned long long* out) {
    if constexpr (COUNT) {
        const uint32_t* v = reinterpret_cast<const uint32_t*>(&c);
        for (int k = 0; k < 9; ++k) {
            // wave reduction, then one atomic per wave
            unsigned long long s = v[k];
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
            if ((threadIdx.x & 63) == 0) atomicAdd(&out[k], s);
        }
    }
}

// ------------------------------------------------------------------------------------------------ spectra & the VM
// Spectrum::get, project/spectra.rs:30-58 (+ Interpolated::get, math.rs:22-72).
DEV float spectrum_get(const DevScene& S, uint32_t id, float w) {
    const PyrSpectrum sp = S.spectra[id];
    const float* data = S.spectrum_data + sp.offset;
    if (sp.format == PYR_SPECTRUM_ARRAY) {
        if (sp.count == 0) return 0.0f;
        if (w <= sp.min) return data[0];
        if (w >= sp.max) return data[sp.count - 1];
        float normalized = (w - sp.min) / (sp.max - sp.min);
        float float_index = normalized * ((float)sp.count - 1.0f);
        float min_float_index = truncf(float_index);
        uint32_t i0 = (uint32_t)min_float_index;
        float mix = float_index - min_float_index;
        return data[i0] * (1.0f - mix) + data[i0 + 1] * mix;
    }
    uint32_t count = sp.count;
    if (count == 0) return 0.0f;
    uint32_t mn = 0, mx = count - 1;
    if (data[2 * mn] >= w) return 0.0f;
    if (data[2 * mx] <= w) return 0.0f;
    while (mx > mn + 1) {
        uint32_t check = (mx + mn) / 2;
        float cx = data[2 * check];
        if (cx == w) return data[2 * check + 1];
        if (cx > w)
            mx = check;
        else
            mn = check;
    }
    float min_x = data[2 * mn], min_y = data[2 * mn + 1];
    float max_x = data[2 * mx], max_y = data[2 * mx + 1];
    if (w < min_x || w > max_x) return 0.0f;
    return min_y + (max_y - min_y) * ((w - min_x) / (max_x - min_x));
}

// math.rs:75-96 / :167-175
DEV float schlick(float n1, float n2, f3 normal, f3 incident) {
    float cos_psi = -dot(normal, incident);
    float r0 = (n1 - n2) / (n1 + n2);
    if (n1 > n2) {
        float n = n1 / n2;
        float sin_t2 = n * n * (1.0f - cos_psi * cos_psi);
        if (sin_t2 > 1.0f) return 1.0f;
        cos_psi = sqrt32(1.0f - sin_t2);
    }
    float inv_cos = 1.0f - cos_psi;
    return r0 * r0 + (1.0f - r0 * r0) * inv_cos * inv_cos * inv_cos * inv_cos * inv_cos;
}
DEV float fresnel(float ior, float env_ior, f3 normal, f3 incident) {
    if (dot(incident, normal) < 0.0f) return schlick(env_ior, ior, normal, incident);
    return schlick(ior, env_ior, -normal, incident);
}
// math.rs:177-182
DEV float blackbody(float wavelength, float temperature) {
    float wl = wavelength * 1.0e-9f;
    float a2 = wl * wl;
    float a4 = a2 * a2;
    float power_term = 3.74183e-16f * (1.0f / (wl * a4));
    return power_term / (exp32(1.4388e-2f / (wl * temperature)) - 1.0f);
}

struct VmInput { // RenderContext / ProbabilityInput / NormalInput
    float wavelength;
    f3 normal, incident;
    float tx = 0.0f, ty = 0.0f; // texture coordinates
};

// Texture::get_color (texture.rs:87-150) + bicubic_interpolate / cubic_interpolate (:297-334): 4 x 4 texels around the
// position with wrap-around, rows counted from the top of the image, one channel at a time (LinSrgba's operators are
// component-wise). Same operation order as oracle.cpp's texture_get_color.
DEV float cubic_interpolate(float v1, float v2, float v3, float v4, float pos) {
    float a = (v4 - v3) - (v1 - v2);
    float b = (v1 - v2) - a;
    float c = v3 - v1;
    float d = v2;
    return d + (c + (b + a * pos) * pos) * pos;
}
DEV void texture_get(const DevScene& S, uint32_t id, float px, float py, float out[4]) {
    const DevTexture t = S.textures[id];
    const float* data = S.texture_data + t.offset;
    const int w = (int)t.width, h = (int)t.height;
    float x = px * (float)t.width - 0.5f;
    float x_floor = floorf(x);
    float y = (1.0f - py) * (float)t.height - 0.5f;
    float y_floor = floorf(y);
    // `as isize` saturates and maps NaN to 0; rem_euclid keeps the result in [0, n)
    auto wrap = [](float f, int n) -> int {
        long long i = (f != f) ? 0ll : (f >= 9.2233720368547758e18f ? 0x7fffffffffffffffll : (f <= -9.2233720368547758e18f ? (long long)0x8000000000000000ull : (long long)f));
        long long r = i % (long long)n;
        return (int)(r < 0 ? r + n : r);
    };
    int xs[4], ys[4];
    xs[1] = wrap(x_floor, w);
    xs[0] = xs[1] == 0 ? w - 1 : xs[1] - 1;
    xs[2] = xs[1] == w - 1 ? 0 : xs[1] + 1;
    xs[3] = xs[2] == w - 1 ? 0 : xs[2] + 1;
    ys[1] = wrap(y_floor, h);
    ys[0] = ys[1] == 0 ? h - 1 : ys[1] - 1;
    ys[2] = ys[1] == h - 1 ? 0 : ys[1] + 1;
    ys[3] = ys[2] == h - 1 ? 0 : ys[2] + 1;
    const float fx = x - x_floor, fy = y - y_floor;
    if (t.channels == 4 && (t.offset & 3ull) == 0ull) {
        // a colour texture whose texels are 16-byte aligned: the sixteen texels as sixteen dwordx4 loads instead of sixty-four dword
        // loads, row by row; per channel the same cubic_interpolate calls on the same values as the loop below
        const float4* texels = reinterpret_cast<const float4*>(data);
        float rows[4][4];
        for (int r = 0; r < 4; ++r) {
            const float4* row = texels + (size_t)ys[r] * t.width;
            const float4 a = row[xs[0]], b = row[xs[1]], c = row[xs[2]], d = row[xs[3]];
            rows[r][0] = cubic_interpolate(a.x, b.x, c.x, d.x, fx);
            rows[r][1] = cubic_interpolate(a.y, b.y, c.y, d.y, fx);
            rows[r][2] = cubic_interpolate(a.z, b.z, c.z, d.z, fx);
            rows[r][3] = cubic_interpolate(a.w, b.w, c.w, d.w, fx);
        }
        for (int ch = 0; ch < 4; ++ch) out[ch] = cubic_interpolate(rows[0][ch], rows[1][ch], rows[2][ch], rows[3][ch], fy);
        return;
    }
    for (uint32_t ch = 0; ch < t.channels; ++ch) {
        float rows[4];
        for (int r = 0; r < 4; ++r) {
            const float* row = data + (size_t)ys[r] * t.width * t.channels + ch;
            rows[r] = cubic_interpolate(row[(size_t)xs[0] * t.channels], row[(size_t)xs[1] * t.channels], row[(size_t)xs[2] * t.channels],
                                        row[(size_t)xs[3] * t.channels], fx);
        }
        out[ch] = cubic_interpolate(rows[0], rows[1], rows[2], rows[3], fy);
    }
}

// The register interpreter (program/execution_context.rs:69-283). Programs are pure functions of their input, so the
// reference's memoised re-run (execute only wavelength-dependent instructions) and a full run give the same value.
// The register files and one instruction of the interpreter, as a body the stage scheduler's `contribute_now` runs in line:
// the files persist between runs, so a program can be run in full for the hero wavelength and then, for every companion
// wavelength, only the instructions that depend on the wavelength (PyrInstr::deps) -- the reference's memoised re-run
// (program/memoized.rs:22-39, execution_context.rs:311-342): a texture is sampled once per hit, not once per wavelength.
struct Vm {
    float num[PYR_MAX_NUMBER_REGISTERS];
    float vec[PYR_MAX_VECTOR_REGISTERS][4];
    float rgb[PYR_MAX_RGB_REGISTERS][4];
    DEV void step(const DevScene& S, const PyrInstr& ins, const VmInput& in);
    DEV float number(const DevProgram& p) const { return p.output_kind == PYR_OUTPUT_NUMBER ? num[p.output_reg & (PYR_MAX_NUMBER_REGISTERS - 1)] : vec[p.output_reg & (PYR_MAX_VECTOR_REGISTERS - 1)][0]; }
};

// Vm::step: one instruction of the register interpreter (program/execution_context.rs:69-283).
DEV void Vm::step(const DevScene& S, const PyrInstr& ins, const VmInput& in) {
    auto value = [&](const PyrOperand& o) -> float {
        if (o.kind == PYR_OPERAND_CONSTANT) return __uint_as_float(o.bits);
        if (o.kind == PYR_OPERAND_INPUT) return in.wavelength;
        return num[o.bits & (PYR_MAX_NUMBER_REGISTERS - 1)];
    };
    auto vinput = [&](uint32_t which, float out[4]) {
        f3 v = which == PYR_INPUT_NORMAL ? in.normal : (which == PYR_INPUT_INCIDENT ? in.incident : mk(in.tx, in.ty, 0.0f));
        out[0] = v.x;
        out[1] = v.y;
        out[2] = v.z;
        out[3] = 0.0f;
    };
    auto binop = [](uint32_t op, float l, float r) -> float {
        switch (op) {
        case PYR_BIN_ADD: return l + r;
        case PYR_BIN_SUB: return l - r;
        case PYR_BIN_MUL: return l * r;
        default: return l / r;
        }
    };
    const uint32_t out = ins.output;
    switch (ins.op) {
    case PYR_OP_NUMBER: num[out & (PYR_MAX_NUMBER_REGISTERS - 1)] = __uint_as_float(ins.x.bits); break;
    case PYR_OP_VECTOR: {
        float x = value(ins.x), y = value(ins.y), z = value(ins.z), w = value(ins.w);
        float* v = vec[out & (PYR_MAX_VECTOR_REGISTERS - 1)];
        v[0] = x, v[1] = y, v[2] = z, v[3] = w;
        break;
    }
    case PYR_OP_RGB: {
        float r = value(ins.x), g = value(ins.y), b = value(ins.z);
        float* v = rgb[out & (PYR_MAX_VECTOR_REGISTERS - 1)];
        v[0] = r, v[1] = g, v[2] = b, v[3] = 1.0f;
        break;
    }
    case PYR_OP_SPECTRUM: num[out & (PYR_MAX_NUMBER_REGISTERS - 1)] = spectrum_get(S, ins.a, value(ins.x)); break;
    case PYR_OP_COLOR_TEXTURE: { // execution_context.rs:114-126
        float pos[4];
        vinput(ins.b, pos);
        float c[4];
        texture_get(S, ins.a, pos[0], pos[1], c);
        float* v = rgb[out & (PYR_MAX_VECTOR_REGISTERS - 1)];
        for (int j = 0; j < 4; ++j) v[j] = c[j];
        break;
    }
    case PYR_OP_MONO_TEXTURE: { // :127-139
        float pos[4];
        vinput(ins.b, pos);
        float c[4];
        texture_get(S, ins.a, pos[0], pos[1], c);
        num[out & (PYR_MAX_NUMBER_REGISTERS - 1)] = c[0];
        break;
    }
    case PYR_OP_RGB_SPECTRUM: {
        float wl = value(ins.x);
        const float* c = rgb[ins.a & (PYR_MAX_VECTOR_REGISTERS - 1)];
        float resp[3] = {0, 0, 0};
        uint32_t count = S.rgb_count;
        if (count > 0) {
            const float* d = S.rgb_basis;
            if (wl <= S.rgb_min) {
                for (int j = 0; j < 3; ++j) resp[j] = d[j];
            } else if (wl >= S.rgb_max) {
                for (int j = 0; j < 3; ++j) resp[j] = d[3 * (count - 1) + j];
            } else {
                float normalized = (wl - S.rgb_min) / (S.rgb_max - S.rgb_min);
                float fi = normalized * ((float)count - 1.0f);
                float fmin_ = truncf(fi);
                uint32_t i0 = (uint32_t)fmin_;
                float mix = fi - fmin_;
                for (int j = 0; j < 3; ++j) resp[j] = d[3 * i0 + j] * (1.0f - mix) + d[3 * (i0 + 1) + j] * mix;
            }
        }
        num[out & (PYR_MAX_NUMBER_REGISTERS - 1)] = c[0] * resp[0] + c[1] * resp[1] + c[2] * resp[2];
        break;
    }
    case PYR_OP_FRESNEL: {
        float ior = value(ins.x), env = value(ins.y);
        float nn[4], ii[4];
        vinput(ins.a, nn);
        vinput(ins.b, ii);
        num[out & (PYR_MAX_NUMBER_REGISTERS - 1)] = fresnel(ior, env, mk(nn[0], nn[1], nn[2]), mk(ii[0], ii[1], ii[2]));
        break;
    }
    case PYR_OP_BLACKBODY: {
        float wl = value(ins.x), temp = value(ins.y);
        num[out & (PYR_MAX_NUMBER_REGISTERS - 1)] = blackbody(wl, temp);
        break;
    }
    case PYR_OP_RGB_TO_VECTOR: {
        const float* c = rgb[ins.a & (PYR_MAX_VECTOR_REGISTERS - 1)];
        float* v = vec[out & (PYR_MAX_VECTOR_REGISTERS - 1)];
        for (int j = 0; j < 4; ++j) v[j] = (c[j] * 2.0f) - 1.0f;
        break;
    }
    case PYR_OP_MIX: {
        float amount = fmaxf(fminf(value(ins.x), 1.0f), 0.0f);
        if (ins.value_type == PYR_VT_NUMBER) {
            float l = num[ins.a & (PYR_MAX_NUMBER_REGISTERS - 1)], r = num[ins.b & (PYR_MAX_NUMBER_REGISTERS - 1)];
            num[out & (PYR_MAX_NUMBER_REGISTERS - 1)] = l * (1.0f - amount) + r * amount;
        } else {
            float* l = ins.value_type == PYR_VT_VECTOR ? vec[ins.a & (PYR_MAX_VECTOR_REGISTERS - 1)] : rgb[ins.a & (PYR_MAX_VECTOR_REGISTERS - 1)];
            float* r = ins.value_type == PYR_VT_VECTOR ? vec[ins.b & (PYR_MAX_VECTOR_REGISTERS - 1)] : rgb[ins.b & (PYR_MAX_VECTOR_REGISTERS - 1)];
            float* o = ins.value_type == PYR_VT_VECTOR ? vec[out & (PYR_MAX_VECTOR_REGISTERS - 1)] : rgb[out & (PYR_MAX_VECTOR_REGISTERS - 1)];
            float t[4];
            for (int j = 0; j < 4; ++j) t[j] = l[j] + (r[j] - l[j]) * amount;
            for (int j = 0; j < 4; ++j) o[j] = t[j];
        }
        break;
    }
    case PYR_OP_BINARY: {
        if (ins.value_type == PYR_VT_NUMBER) {
            num[out & (PYR_MAX_NUMBER_REGISTERS - 1)] = binop(ins.operator_, num[ins.a & (PYR_MAX_NUMBER_REGISTERS - 1)], num[ins.b & (PYR_MAX_NUMBER_REGISTERS - 1)]);
        } else {
            float* l = ins.value_type == PYR_VT_VECTOR ? vec[ins.a & (PYR_MAX_VECTOR_REGISTERS - 1)] : rgb[ins.a & (PYR_MAX_VECTOR_REGISTERS - 1)];
            float* r = ins.value_type == PYR_VT_VECTOR ? vec[ins.b & (PYR_MAX_VECTOR_REGISTERS - 1)] : rgb[ins.b & (PYR_MAX_VECTOR_REGISTERS - 1)];
            float* o = ins.value_type == PYR_VT_VECTOR ? vec[out & (PYR_MAX_VECTOR_REGISTERS - 1)] : rgb[out & (PYR_MAX_VECTOR_REGISTERS - 1)];
            float t[4];
            for (int j = 0; j < 4; ++j) t[j] = binop(ins.operator_, l[j], r[j]);
            for (int j = 0; j < 4; ++j) o[j] = t[j];
        }
        break;
    }
    case PYR_OP_CLAMP: {
        float v = value(ins.x), mn = value(ins.y), mx = value(ins.z);
        num[out & (PYR_MAX_NUMBER_REGISTERS - 1)] = fmaxf(fminf(v, mx), mn);
        break;
    }
    default: break;
    }
}

// A whole program, out of line. `vector_out` (normal maps): receives the four components of a Vector program's output register.
__device__ __noinline__ float run_interpreter(const DevScene& S, const DevProgram& p, const VmInput& in, float* vector_out = nullptr) {
    Vm vm;
    for (uint32_t k = 0; k < p.num_instrs; ++k) vm.step(S, S.instrs[p.first_instr + k], in);
    if (p.output_kind == PYR_OUTPUT_NUMBER) {
        const float n = vm.num[p.output_reg & (PYR_MAX_NUMBER_REGISTERS - 1)];
        if (vector_out) vector_out[0] = vector_out[1] = vector_out[2] = vector_out[3] = n;
        return n;
    }
    if (vector_out)
        for (int j = 0; j < 4; ++j) vector_out[j] = vm.vec[p.output_reg & (PYR_MAX_VECTOR_REGISTERS - 1)][j];
    return vm.vec[p.output_reg & (PYR_MAX_VECTOR_REGISTERS - 1)][0];
}

// ExecutionContext::run (execution_context.rs:29-56) with the three shapes every Cornell-family program has short-cut.
// INTERP = false builds the kernel without the interpreter (and its register files in scratch) for scenes whose programs
// are all constants or one of the fast shapes -- every BASELINE configuration; the host picks the variant per scene.
template <bool INTERP>
DEV float run_program(const DevScene& S, uint32_t id, const VmInput& in) {
    const DevProgram p = S.programs[id];
    if (p.kind == PYR_PROGRAM_CONSTANT) return p.constant;
    switch (p.fast) {
    case FAST_SPECTRUM: return spectrum_get(S, p.fast_spectrum, in.wavelength);
    case FAST_SPECTRUM_MUL: return spectrum_get(S, p.fast_spectrum, in.wavelength) * p.fast_scale;
    case FAST_MUL_SPECTRUM: return p.fast_scale * spectrum_get(S, p.fast_spectrum, in.wavelength);
    default:
        if constexpr (INTERP) return run_interpreter(S, p, in);
        return 0.0f;
    }
}

// A colour program is evaluated for the hero wavelength and then for each companion: `prepare_program` pulls the program
// record and (for the fast shapes) the spectrum record into registers once, `eval_prepared` is then the arithmetic of
// Spectrum::get alone -- the same operations as spectrum_get / run_program, so the values are identical.
struct Prepared {
    uint32_t mode; // 0 constant, FAST_* for the fast shapes, 0xFF interpreter
    float c;       // the constant / the scale
    PyrSpectrum sp;
    const float* data;
    uint32_t id;
};
template <bool INTERP>
DEV Prepared prepare_program(const DevScene& S, uint32_t id) {
    Prepared q{};
    q.id = id;
    const DevProgram* p = S.programs + id;
    if (p->kind == PYR_PROGRAM_CONSTANT) {
        q.mode = 0;
        q.c = p->constant;
        return q;
    }
    const uint32_t fast = p->fast;
    if (fast == FAST_NONE) {
        q.mode = 0xFFu;
        return q;
    }
    q.mode = fast;
    q.c = p->fast_scale;
    q.sp = S.spectra[p->fast_spectrum];
    q.data = S.spectrum_data + q.sp.offset;
    return q;
}
DEV float spectrum_eval(const PyrSpectrum& sp, const float* data, float w) { // == spectrum_get with the record in registers
    if (sp.format == PYR_SPECTRUM_ARRAY) {
        if (sp.count == 0) return 0.0f;
        if (w <= sp.min) return data[0];
        if (w >= sp.max) return data[sp.count - 1];
        float normalized = (w - sp.min) / (sp.max - sp.min);
        float float_index = normalized * ((float)sp.count - 1.0f);
        float min_float_index = truncf(float_index);
        uint32_t i0 = (uint32_t)min_float_index;
        float mix = float_index - min_float_index;
        return data[i0] * (1.0f - mix) + data[i0 + 1] * mix;
    }
    uint32_t count = sp.count;
    if (count == 0) return 0.0f;
    uint32_t mn = 0, mx = count - 1;
    if (data[2 * mn] >= w) return 0.0f;
    if (data[2 * mx] <= w) return 0.0f;
    while (mx > mn + 1) {
        uint32_t check = (mx + mn) / 2;
        float cx = data[2 * check];
        if (cx == w) return data[2 * check + 1];
        if (cx > w)
            mx = check;
        else
            mn = check;
    }
    float min_x = data[2 * mn], min_y = data[2 * mn + 1];
    float max_x = data[2 * mx], max_y = data[2 * mx + 1];
    if (w < min_x || w > max_x) return 0.0f;
    return min_y + (max_y - min_y) * ((w - min_x) / (max_x - min_x));
}
template <bool INTERP>
DEV float eval_prepared(const DevScene& S, const Prepared& q, const VmInput& in) {
    switch (q.mode) {
    case 0: return q.c;
    case FAST_SPECTRUM: return spectrum_eval(q.sp, q.data, in.wavelength);
    case FAST_SPECTRUM_MUL: return spectrum_eval(q.sp, q.data, in.wavelength) * q.c;
    case FAST_MUL_SPECTRUM: return q.c * spectrum_eval(q.sp, q.data, in.wavelength);
    default:
        if constexpr (INTERP) return run_interpreter(S, S.programs[q.id], in);
        return 0.0f;
    }
}

// ------------------------------------------------------------------------------------------------ sampling helpers (math.rs)
DEV f3 ortho(f3 v) { // math.rs:98-114
    f3 unit;
    if (fabsf(v.x) < DIST_EPSILON)
        unit = mk(1, 0, 0);
    else if (fabsf(v.y) < DIST_EPSILON)
        unit = mk(0, 1, 0);
    else if (fabsf(v.z) < DIST_EPSILON)
        unit = mk(0, 0, 1);
    else
        unit = mk(-v.y, v.x, 0.0f);
    return cross(v, unit);
}
DEV f3 sample_cone(Rng& rng, f3 direction, float cos_half) { // math.rs:125-137
    f3 o1 = normalize(ortho(direction));
    f3 o2 = normalize(cross(direction, o1));
    float r1 = PI_F * 2.0f * rng_f32(rng);
    float r2 = cos_half + (1.0f - cos_half) * rng_f32(rng);
    float oneminus = sqrt32(1.0f - r2 * r2);
    return o1 * cos32(r1) * oneminus + o2 * sin32(r1) * oneminus + direction * r2;
}
DEV float solid_angle(float cos_half) { return cos_half >= 1.0f ? 0.0f : 2.0f * PI_F * (1.0f - cos_half); } // :139-145
DEV f3 sample_sphere(Rng& rng) { // math.rs:147-153
    float u = rng_f32(rng);
    float v = rng_f32(rng);
    float theta = 2.0f * PI_F * u;
    float phi = acos32(2.0f * v - 1.0f);
    float sp = sin32(phi);
    return mk(sp * cos32(theta), sp * sin32(theta), cos32(phi));
}
DEV f3 sample_hemisphere(Rng& rng, f3 direction) { // math.rs:155-164
    f3 s = sample_sphere(rng);
    f3 x = normalize_to(ortho(direction), s.x);
    f3 y = normalize_to(cross(x, direction), s.y);
    f3 z = normalize_to(direction, fabsf(s.z));
    return x + y + z;
}

// ------------------------------------------------------------------------------------------------ primitives
// shapes/mod.rs:75-119, same operation order as the reference.
DEV bool triangle_test(f3 v1, f3 e1, f3 e2, f3 o, f3 d, float& dist, float& u, float& v) {
    // The reference returns at each failed test; here every lane runs straight through and the verdicts are combined at the
    // end -- the same operations in the same order on the lanes that pass, garbage that nobody reads on the others. A wave
    // executes the whole routine for its slowest lane anyway, and each early exit cost a saved exec mask, a branch and the
    // mask bookkeeping to merge the outcomes (the nested exits also pushed the kernel's scalar registers into spills).
    f3 p = cross(d, e2);
    float det = dot(e1, p);
    const bool det_ok = !(det > -DIST_EPSILON && det < DIST_EPSILON);
    float inv_det = 1.0f / det;
    f3 t = o - v1;
    u = dot(t, p) * inv_det;
    const bool u_ok = !(u < 0.0f || u > 1.0f);
    f3 q = cross(t, e1);
    v = dot(d, q) * inv_det;
    const bool v_ok = !(v < 0.0f || u + v > 1.0f);
    dist = dot(e2, q) * inv_det;
    return det_ok & u_ok & v_ok & (dist > DIST_EPSILON);
}
// shapes/mod.rs:57-74 via collision::Sphere (oracle.cpp sphere_intersect).
DEV bool sphere_test(f3 center, float radius, f3 o, f3 d, float& dist, f3& point) {
    f3 l = center - o;
    float tca = dot(l, d);
    if (tca < 0.0f) return false;
    float d2 = dot(l, l) - tca * tca;
    if (d2 > radius * radius) return false;
    float thc = sqrt32(radius * radius - d2);
    point = o + d * (tca - thc);
    dist = magnitude(point - o);
    return true;
}
// shapes/mod.rs:441-452 (oracle.cpp plane_intersect).
DEV bool plane_test(f3 origin, f3 normal, f3 o, f3 d, float& dist, f3& point) {
    float t = (dot(origin, normal) - dot(o, normal)) / dot(d, normal);
    if (!(t >= 0.0f)) return false;
    point = o + d * t;
    dist = magnitude(point - o);
    return true;
}

struct Hit {
    float t;
    uint32_t shape; // PYR_HIT_NONE or (kind << 30) | index
    float u, v;
};

// math.rs:184-207: entry distance of the ray into a box, or -1 when it misses.
DEV float slab(f3 lo, f3 hi, f3 o, f3 inv) {
    float t1 = (lo.x - o.x) * inv.x, t2 = (hi.x - o.x) * inv.x;
    float tmin = fminf(t1, t2), tmax = fmaxf(t1, t2);
    t1 = (lo.y - o.y) * inv.y, t2 = (hi.y - o.y) * inv.y;
    tmin = fmaxf(tmin, fminf(t1, t2)), tmax = fminf(tmax, fmaxf(t1, t2));
    t1 = (lo.z - o.z) * inv.z, t2 = (hi.z - o.z) * inv.z;
    tmin = fmaxf(tmin, fminf(t1, t2)), tmax = fminf(tmax, fmaxf(t1, t2));
    return (tmax >= tmin && tmax >= 0.0f) ? fmaxf(tmin, 0.0f) : -1.0f;
}

// The reference never calls a shape's intersection routine unless the ray passes the shape's own bounding box
// (spatial/bvh.rs:201-230: one item per leaf; math.rs:184-207 for the box) and enters it before the closest hit so far. For
// triangles that changes nothing but ties at the box surface; collision's sphere routine, however, assumes a unit direction
// (`tca = l . d`, `d2 = l . l - tca^2`) and reports hits for rays that pass the sphere at a distance when the direction is
// longer than 1 -- a directional lamp whose `direction` is not normalised produces such rays (lamp.rs:24-35) -- and the
// reference is only saved from them by that box. Leaves here hold up to four primitives under one box, so spheres get the
// reference's own test, in its own arithmetic (IEEE 1 / d, (bound - o) * inv), before the sphere routine runs.
DEV bool sphere_box_guard(f3 center, float radius, f3 o, f3 d, float closest, bool any_hit) {
    const f3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const float entry = slab(mk(center.x - radius, center.y - radius, center.z - radius), mk(center.x + radius, center.y + radius, center.z + radius), o, inv);
    return entry >= 0.0f && (any_hit || entry < closest);
}

// Both children of a node at once. The node stores the bounds interleaved (bvh.h Node64: lo.x lo.y | lo.z hi.x | hi.y hi.z,
// each as a (child 0, child 1) pair), so the twelve plane distances are six v_pk_fma_f32: t = bound * inv - o * inv. That
// form rounds differently from math.rs:184-207's (bound - o) * inv; the boxes are this library's own and are padded at build
// time so that the test stays conservative (bvh.cpp).
// 1 / direction for the box tests only: v_rcp_f32 (1 ulp) instead of an IEEE division (~12 instructions each). The boxes are
// padded by 16 ulps of the scene extent, which covers it; primitive tests never see this value. rcp(+-0) = +-inf as 1 / 0.
DEV f3 box_reciprocal(f3 d) { return mk(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z)); }

typedef float f2v __attribute__((ext_vector_type(2)));
// Planes are picked by the sign of the ray's direction: `node` points at a Node64, sx / sy / sz are 0 for a
// positive direction component and 24 for a negative one (the byte distance from lo_* to hi_* in the node), so the six 8-byte
// loads fetch, per axis, the (child 0, child 1) pair of the plane the ray meets first and of the one it meets last; entry =
// max of the near distances, exit = min of the far ones -- 4 min / max instead of 12. t = bound * inv - o * inv is monotonic in
// `bound`, so the near distance IS min(t_lo, t_hi), value for value, unless 0 * inf or inf - inf makes one of the two NaN (a
// direction component of exactly zero): a min / max over both planes (rounds 1-2) then used the other one for both entry and exit
// and rejected boxes the ray lies inside of on that axis (an axis-parallel ray through the Cornell box missed everything); here the NaN
// drops out of max / min and the axis does not constrain -- what math.rs:184-207's (bound - o) * inv gives for such a ray.
typedef __attribute__((address_space(1))) const f2v global_f2v;
template <bool GLOBAL>
DEV void slab_pair_signed(const float4* nodes, uint32_t node, uint32_t sx, uint32_t sy, uint32_t sz, f3 o, f3 inv, float& e0, float& e1, int& c0, int& c1) {
    // one (uniform) base and a 32-bit byte offset per lane: an SGPR-based global_load, or a plain LDS address
    const char* base = reinterpret_cast<const char*>(nodes);
    auto pair = [&](uint32_t byte_offset) -> f2v {
        if constexpr (GLOBAL) return *(global_f2v*)(base + byte_offset);
        return *reinterpret_cast<const f2v*>(base + byte_offset);
    };
    const uint32_t at = node << 6, ax = at + sx, ay = at + sy, az = at + sz;
    const f2v nx = pair(ax), fx = pair(ax ^ 24u), ny = pair(ay + 8u), fy = pair((ay ^ 24u) + 8u), nz = pair(az + 16u), fz = pair((az ^ 24u) + 16u);
    const f2v ch = pair(at + 48u);
    const f2v ix = {inv.x, inv.x}, iy = {inv.y, inv.y}, iz = {inv.z, inv.z};
    const float ox = -(o.x * inv.x), oy = -(o.y * inv.y), oz = -(o.z * inv.z);
    const f2v nox = {ox, ox}, noy = {oy, oy}, noz = {oz, oz};
    const f2v tnx = __builtin_elementwise_fma(nx, ix, nox), tfx = __builtin_elementwise_fma(fx, ix, nox);
    const f2v tny = __builtin_elementwise_fma(ny, iy, noy), tfy = __builtin_elementwise_fma(fy, iy, noy);
    const f2v tnz = __builtin_elementwise_fma(nz, iz, noz), tfz = __builtin_elementwise_fma(fz, iz, noz);
    const float tmin0 = fmaxf(fmaxf(tnx.x, tny.x), tnz.x), tmax0 = fminf(fminf(tfx.x, tfy.x), tfz.x);
    const float tmin1 = fmaxf(fmaxf(tnx.y, tny.y), tnz.y), tmax1 = fminf(fminf(tfx.y, tfy.y), tfz.y);
    const float entry0 = fmaxf(tmin0, 0.0f), entry1 = fmaxf(tmin1, 0.0f);
    e0 = tmax0 >= entry0 ? entry0 : -1.0f; // tmax >= max(tmin, 0) is (tmax >= tmin) & (tmax >= 0); a NaN exit fails it
    e1 = tmax1 >= entry1 ? entry1 : -1.0f;
    c0 = __float_as_int(ch.x), c1 = __float_as_int(ch.y);
}
DEV uint32_t binary_sign_offset(float component) { return (__float_as_uint(component) >> 31) * 24u; }

// World::intersect (world.rs:273-299). SHADOW = false: closest hit with DIST_EPSILON < d < closest.
// SHADOW = true: answers trace_direct's visibility question (tracer.rs:381-389) -- "is there a hit with
// d > eps and d*d < limit" -- and returns as soon as one is found (equivalent to testing the closest hit, because
// d -> d*d is monotonic). `limit` = lamp_distance^2 - eps, or +inf when the lamp has no distance.
template <bool COUNT, bool SHADOW>
DEV bool traverse(const DevScene& S, const float4* nodes, const float4* prims, f3 o, f3 d, float limit, Hit& hit, int* stack, Counters& cnt) {
    float closest = PYR_INF;
    hit.shape = PYR_HIT_NONE;
    hit.t = PYR_INF;
    hit.u = hit.v = 0.0f;
    for (uint32_t i = 0; i < S.num_planes; ++i) {
        const float* pl = S.planes + 8 * i;
        float dist;
        f3 point;
        if (COUNT) cnt.plane_tests++;
        if (plane_test(ld3(pl), ld3(pl + 3), o, d, dist, point)) {
            if (SHADOW) {
                if (dist > DIST_EPSILON && dist * dist < limit) return true;
            } else if (dist > DIST_EPSILON && dist < closest) {
                closest = dist;
                hit.t = dist;
                hit.shape = ((uint32_t)PYR_SHAPE_PLANE << 30) | i;
            }
        }
    }
    const f3 inv = box_reciprocal(d);
    const uint32_t sign_x = binary_sign_offset(d.x), sign_y = binary_sign_offset(d.y), sign_z = binary_sign_offset(d.z);
    const float limit_cull = limit * S.shadow_margin + 1.0e-3f; // +inf stays +inf
    int sp = 0;
    int node = 0;
    // Every turn of the loop is one inner-node visit or ONE primitive of a leaf (the leaf code in `node` shrinks), so lanes at
    // inner nodes do not wait for the fullest leaf of the wave (C2: 585 -> 717 Msamples/s). Letting the wave run only the
    // kind of step most lanes wait for (the minority keeps its place) was slower on C2: 636.
    for (;;) {
        if (node >= 0) {
            if (COUNT) cnt.box_tests += 2;
            float e0, e1;
            int c0, c1;
            slab_pair_signed<false>(nodes, (uint32_t)node, sign_x, sign_y, sign_z, o, inv, e0, e1, c0, c1);
            bool h0, h1;
            if (SHADOW) {
                // A box can be skipped only if nothing in it can block. Its computed entry distance and a primitive's computed
                // hit distance are independent roundings of (at best) the same number -- for a zero-thickness box they differ
                // by ulps either way -- so the cut-off keeps a margin over the blocking limit (DevScene::shadow_margin: 0.1 %, 1 % where spheres are) instead of comparing against
                // it exactly (found on C3: at scene scale 50, d^2 ~ 2500 has an ulp of 2.4e-4 > DIST_EPSILON and an exact
                // comparison skipped lamp triangles the reference tests).
                h0 = e0 >= 0.0f && e0 * e0 < limit_cull;
                h1 = e1 >= 0.0f && e1 * e1 < limit_cull;
            } else {
                h0 = e0 >= 0.0f && e0 < closest; // bvh.rs:213: skip when distance >= max_distance
                h1 = e1 >= 0.0f && e1 < closest;
            }
            if (h0 && h1) {
                bool swap = e1 < e0;
                node = swap ? c1 : c0;
                stack[sp * BLOCK] = swap ? c0 : c1;
                sp++;
            } else if (h0) {
                node = c0;
            } else if (h1) {
                node = c1;
            } else {
                if (sp == 0) break;
                sp--;
                node = stack[sp * BLOCK];
            }
            continue;
        }
        const uint32_t code = (uint32_t)(-1 - node);
        const uint32_t first = code >> 3, count = code & 7u;
        if (count != 0) {
            const float4 a = prims[3 * first + 0], b = prims[3 * first + 1];
            const uint32_t shape = __float_as_uint(a.w);
            float dist, u = 0.0f, v = 0.0f;
            bool ok;
            if ((shape >> 30) == PYR_SHAPE_TRIANGLE) {
                const float4 c = prims[3 * first + 2];
                if (COUNT) cnt.triangle_tests++;
                ok = triangle_test(mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), mk(c.x, c.y, c.z), o, d, dist, u, v);
            } else {
                f3 point;
                if (COUNT) cnt.sphere_tests++;
                ok = sphere_box_guard(mk(a.x, a.y, a.z), b.x, o, d, closest, SHADOW) && sphere_test(mk(a.x, a.y, a.z), b.x, o, d, dist, point);
            }
            if (ok) {
                if (SHADOW) {
                    if (dist > DIST_EPSILON && dist * dist < limit) return true;
                } else if (dist > DIST_EPSILON && dist < closest) {
                    closest = dist;
                    hit.t = dist;
                    hit.shape = shape;
                    hit.u = u;
                    hit.v = v;
                }
            }
            if (count > 1) {
                node = -1 - (int)(((first + 1) << 3) | (count - 1));
                continue;
            }
        }
        if (sp == 0) break;
        sp--;
        node = stack[sp * BLOCK];
    }
    return hit.shape != PYR_HIT_NONE;
}

// ------------------------------------------------------------------------------------------------ camera / film mapping
struct TileArea {
    float from_x, from_y, size_x, size_y;
};
// Camera::to_view_area, cameras.rs:57-68 -- unfused so tile rectangles equal the oracle's bit for bit.
DEV TileArea to_view_area(uint32_t x, uint32_t y, uint32_t w, uint32_t h, uint32_t width, uint32_t height) {
    float iw = (float)width, ih = (float)height;
    float half_max = __fmul_rn(fmaxf(iw, ih), 0.5f);
    TileArea a;
    a.from_x = __fdiv_rn(__fadd_rn((float)x, -__fmul_rn(iw, 0.5f)), half_max);
    a.from_y = __fdiv_rn(__fadd_rn((float)y, -__fmul_rn(ih, 0.5f)), half_max);
    a.size_x = __fdiv_rn((float)w, half_max);
    a.size_y = __fdiv_rn((float)h, half_max);
    return a;
}
DEV f3 transform_point(const float* m, f3 p) { // cgmath Matrix4 * (p, 1), then / w
    float x = m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12];
    float y = m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13];
    float z = m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14];
    float w = m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15];
    float inv = 1.0f / w;
    return mk(x * inv, y * inv, z * inv);
}
DEV f3 transform_vector(const float* m, f3 v) {
    return mk(m[0] * v.x + m[4] * v.y + m[8] * v.z, m[1] * v.x + m[5] * v.y + m[9] * v.z, m[2] * v.x + m[6] * v.y + m[10] * v.z);
}
// Rust `as usize`: saturating, NaN -> 0 (values here are far below 2^32).
DEV uint32_t f32_as_index(float f) { return f > 0.0f ? (f >= 4294967040.0f ? 0xFFFFFFFFu : (uint32_t)f) : 0u; }

// ------------------------------------------------------------------------------------------------ the integrator
// LDS: [3 * S][BLOCK] floats (wavelength, brightness, reflectance of the S-1 companions; slot S-1 is scratch during
// sample generation) followed by [stack_depth][BLOCK] ints.
struct Spectral {
    float* base;
    uint32_t s;
    DEV float& wl(uint32_t k) { return base[(0 * s + k) * BLOCK]; }
    DEV float& bright(uint32_t k) { return base[(1 * s + k) * BLOCK]; }
    DEV float& refl(uint32_t k) { return base[(2 * s + k) * BLOCK]; }
};

// Surface normal at a hit (SurfacePoint::get_surface_data, shapes/mod.rs:484-494) and the material id.
DEV void surface_at(const DevScene& S, const Hit& hit, f3 o, f3 d, f3& position, f3& normal, uint32_t& material) {
    const uint32_t kind = hit.shape >> 30, index = hit.shape & 0x3FFFFFFFu;
    if (kind == PYR_SHAPE_TRIANGLE) {
        const float4* sh = reinterpret_cast<const float4*>(S.tri_shade) + 3 * (size_t)index;
        const float4 a = sh[0], b = sh[1], c = sh[2];
        float w = 1.0f - (hit.u + hit.v);
        normal = normalize(mk(a.x, a.y, a.z) * w + mk(b.x, b.y, b.z) * hit.u + mk(c.x, c.y, c.z) * hit.v); // Normal::on_triangle :550-558
        material = __float_as_uint(a.w);
        position = o + d * hit.t;
    } else if (kind == PYR_SHAPE_SPHERE) {
        const float4 sp = reinterpret_cast<const float4*>(S.spheres)[index];
        float dist;
        sphere_test(mk(sp.x, sp.y, sp.z), sp.w, o, d, dist, position); // the intersection point collision returned
        normal = normalize(position - mk(sp.x, sp.y, sp.z));
        material = S.sphere_material[index];
    } else {
        const float* pl = S.planes + 8 * index;
        float dist;
        plane_test(ld3(pl), ld3(pl + 3), o, d, dist, position);
        normal = ld3(pl + 3);
        material = S.plane_material[index];
    }
}

// ---- texture space (interpreter builds): Normal {vector, from_space} (shapes/mod.rs:531-584) ------------------------
// [3P] cgmath 0.17 quaternion arithmetic, the same operation order as oracle.cpp's Quat helpers.
struct Quat {
    float s, x, y, z;
};
DEV Quat quat_scale(Quat q, float f) { return Quat{q.s * f, q.x * f, q.y * f, q.z * f}; }
DEV Quat quat_add(Quat a, Quat b) { return Quat{a.s + b.s, a.x + b.x, a.y + b.y, a.z + b.z}; }
DEV Quat quat_normalize(Quat q) {
    float m = sqrt32(q.s * q.s + q.x * q.x + q.y * q.y + q.z * q.z);
    return quat_scale(q, 1.0f / m);
}
DEV Quat quat_conjugate(Quat q) { return Quat{q.s, -q.x, -q.y, -q.z}; }
DEV f3 quat_rotate(Quat q, f3 vec) {
    f3 v = mk(q.x, q.y, q.z);
    f3 tmp = cross(v, vec) + vec * q.s;
    return cross(v, tmp) * 2.0f + vec;
}
DEV Quat quat_from_cols(f3 c0, f3 c1, f3 c2) {
    const float m00 = c0.x, m01 = c0.y, m02 = c0.z, m10 = c1.x, m11 = c1.y, m12 = c1.z, m20 = c2.x, m21 = c2.y, m22 = c2.z;
    float trace = m00 + m11 + m22;
    if (trace >= 0.0f) {
        float s = sqrt32(1.0f + trace);
        float w = 0.5f * s;
        s = 0.5f / s;
        return Quat{w, (m12 - m21) * s, (m20 - m02) * s, (m01 - m10) * s};
    } else if (m00 > m11 && m00 > m22) {
        float s = sqrt32((m00 - m11 - m22) + 1.0f);
        float x = 0.5f * s;
        s = 0.5f / s;
        return Quat{(m12 - m21) * s, x, (m10 + m01) * s, (m02 + m20) * s};
    } else if (m11 > m22) {
        float s = sqrt32((m11 - m00 - m22) + 1.0f);
        float y = 0.5f * s;
        s = 0.5f / s;
        return Quat{(m20 - m02) * s, (m10 + m01) * s, y, (m21 + m12) * s};
    } else {
        float s = sqrt32((m22 - m00 - m11) + 1.0f);
        float z = 0.5f * s;
        s = 0.5f / s;
        return Quat{(m01 - m10) * s, (m02 + m20) * s, (m21 + m12) * s, z};
    }
}
DEV float atan2_32(float y, float x) { return (float)atan2((double)y, (double)x); }

// Texture coordinates of a point on a sphere (get_sphere_surface_data, shapes/mod.rs:346-372).
DEV void sphere_texture(f3 normal, float scale_x, float scale_y, float& latitude, float& longitude, float& tx, float& ty) {
    latitude = acos32(normal.y);
    longitude = atan2_32(normal.x, normal.z);
    tx = (longitude * (1.0f / PI_F) * 0.5f) / scale_x;
    ty = (1.0f - (latitude * (1.0f / PI_F))) / scale_y;
}

// SurfacePoint::get_surface_data with texture coordinates (shapes/mod.rs:346-385, :454-469, :484-494) followed by
// Material::apply_normal_map (materials/mod.rs:68-80, tracer.rs:227-232): the shading normal, the material, the texture
// coordinates. The tangent frame is only built for materials that have a normal map.
DEV void surface_textured(const DevScene& S, const Hit& hit, f3 o, f3 d, f3& position, f3& normal, uint32_t& material, float& tx, float& ty) {
    const uint32_t kind = hit.shape >> 30, index = hit.shape & 0x3FFFFFFFu;
    Quat frame{1.0f, 0.0f, 0.0f, 0.0f};
    if (kind == PYR_SHAPE_TRIANGLE) {
        const float4* sh = reinterpret_cast<const float4*>(S.tri_shade) + 3 * (size_t)index;
        const float4 a = sh[0], b = sh[1], c = sh[2];
        const float w = 1.0f - (hit.u + hit.v);
        normal = normalize(mk(a.x, a.y, a.z) * w + mk(b.x, b.y, b.z) * hit.u + mk(c.x, c.y, c.z) * hit.v);
        material = __float_as_uint(a.w);
        position = o + d * hit.t;
        const float4* tt = reinterpret_cast<const float4*>(S.tri_tex) + 5 * (size_t)index;
        const float4 uv12 = tt[0], uv3 = tt[1];
        tx = uv12.x * w + uv12.z * hit.u + uv3.x * hit.v;
        ty = uv12.y * w + uv12.w * hit.u + uv3.y * hit.v;
        if (S.materials[material].normal_map_program >= 0) {
            const float4 f1 = tt[2], f2 = tt[3], f3_ = tt[4];
            frame = quat_normalize(quat_add(quat_add(quat_scale(Quat{f1.x, f1.y, f1.z, f1.w}, w), quat_scale(Quat{f2.x, f2.y, f2.z, f2.w}, hit.u)),
                                            quat_scale(Quat{f3_.x, f3_.y, f3_.z, f3_.w}, hit.v)));
        }
    } else if (kind == PYR_SHAPE_SPHERE) {
        const float4 sp = reinterpret_cast<const float4*>(S.spheres)[index];
        float dist;
        sphere_test(mk(sp.x, sp.y, sp.z), sp.w, o, d, dist, position);
        normal = normalize(position - mk(sp.x, sp.y, sp.z));
        material = S.sphere_material[index];
        float latitude, longitude;
        sphere_texture(normal, S.sphere_tex_scale[2 * index], S.sphere_tex_scale[2 * index + 1], latitude, longitude, tx, ty);
        if (S.materials[material].normal_map_program >= 0) {
            // Matrix3::from_angle_y(longitude) * Matrix3::from_angle_x(latitude - PI / 2) [3P], see oracle.cpp surface_data
            const float sy = sin32(longitude), cy = cos32(longitude);
            const float ax = latitude - PI_F * 0.5f;
            const float sx = sin32(ax), cx = cos32(ax);
            const f3 bc[3] = {mk(1.0f, 0.0f, 0.0f), mk(0.0f, cx, sx), mk(0.0f, -sx, cx)};
            f3 col[3];
            for (int j = 0; j < 3; ++j)
                col[j] = mk(cy * bc[j].x + 0.0f * bc[j].y + sy * bc[j].z, 0.0f * bc[j].x + 1.0f * bc[j].y + 0.0f * bc[j].z,
                            -sy * bc[j].x + 0.0f * bc[j].y + cy * bc[j].z);
            frame = quat_from_cols(col[0], col[1], col[2]);
        }
    } else {
        const float* pl = S.planes + 8 * index;
        float dist;
        plane_test(ld3(pl), ld3(pl + 3), o, d, dist, position);
        normal = ld3(pl + 3);
        material = S.plane_material[index];
        const float* f = S.plane_frames + 4 * index;
        frame = Quat{f[0], f[1], f[2], f[3]};
        const f3 normal_space = quat_rotate(quat_conjugate(frame), position); // Normal::into_space
        tx = normal_space.x / pl[6];
        ty = normal_space.y / pl[7];
    }
    const int normal_map = S.materials[material].normal_map_program;
    if (normal_map >= 0) {
        const DevProgram prog = S.programs[normal_map];
        float v[4] = {prog.constant, prog.constant, prog.constant, prog.constant};
        if (prog.kind != PYR_PROGRAM_CONSTANT) { // in line: an out-of-line call here spills the walker around itself
            const VmInput in{0.0f, normal, d, tx, ty};
            Vm vm;
            for (uint32_t k = 0; k < prog.num_instrs; ++k) vm.step(S, S.instrs[prog.first_instr + k], in);
            if (prog.output_kind == PYR_OUTPUT_NUMBER)
                v[0] = v[1] = v[2] = v[3] = vm.num[prog.output_reg & (PYR_MAX_NUMBER_REGISTERS - 1)];
            else
                for (int j = 0; j < 4; ++j) v[j] = vm.vec[prog.output_reg & (PYR_MAX_VECTOR_REGISTERS - 1)][j];
        }
        normal = normalize(quat_rotate(frame, mk(v[0], v[1], v[2])));
    }
}

// materials/refractive.rs:47-91
DEV void refract(float ior, float env_ior, f3 in_direction, f3 normal, Rng& rng, f3& out, float& prob) {
    f3 nl = dot(normal, in_direction) < 0.0f ? normal : -normal;
    f3 reflected = in_direction - (normal * 2.0f * dot(normal, in_direction));
    bool into = dot(normal, nl) > 0.0f;
    float nnt = into ? env_ior / ior : ior / env_ior;
    float ddn = dot(in_direction, nl);
    float cos2t = 1.0f - nnt * nnt * (1.0f - ddn * ddn);
    if (cos2t < 0.0f) {
        out = reflected;
        prob = 1.0f;
        return;
    }
    float s = (into ? 1.0f : -1.0f) * (ddn * nnt + sqrt32(cos2t));
    f3 tdir = normalize(in_direction * nnt - normal * s);
    float a = ior - env_ior, b = ior + env_ior;
    float r0 = a * a / (b * b);
    float c = 1.0f - (into ? -ddn : dot(tdir, normal));
    float re = r0 + (1.0f - r0) * c * c * c * c * c;
    float tr = 1.0f - re;
    float p = 0.25f + 0.5f * re;
    if (rng_f32(rng) < p) {
        out = reflected;
        prob = re / p;
    } else {
        out = tdir;
        prob = tr / (1.0f - p);
    }
}

struct LampSample { // lamp.rs:116-130
    f3 direction;
    float sq_distance; // < 0: None
    bool physical;
    f3 normal;
    uint32_t material, color;
    float weight;
    float tx, ty; // Surface::Physical::texture (TEX builds only)
};

// Lamp::sample (lamp.rs:23-82) with Shape::sample_towards / sample_point / solid_angle_towards (shapes/mod.rs:166-271).
// TEX: also the sampled point's texture coordinates (lamp.rs:64, :75).
template <bool TEX = false>
DEV LampSample lamp_sample(const DevLamp& lamp, Rng& rng, f3 target) {
    LampSample ls;
    ls.tx = ls.ty = 0.0f;
    ls.physical = false;
    ls.material = 0;
    ls.color = lamp.color_program;
    ls.normal = mk(0, 0, 0);
    if (lamp.kind == PYR_LAMP_DIRECTIONAL) {
        f3 direction = ld3(lamp.v);
        ls.direction = lamp.width > 0.0f ? sample_cone(rng, direction, lamp.width) : direction;
        ls.sq_distance = -1.0f;
        ls.weight = 1.0f;
    } else if (lamp.kind == PYR_LAMP_POINT) {
        f3 v = ld3(lamp.v) - target;
        float distance = dot(v, v);
        ls.direction = normalize(v);
        ls.sq_distance = distance;
        ls.weight = 4.0f * PI_F / distance;
    } else if (lamp.shape_kind == PYR_SHAPE_SPHERE) {
        const f3 center = ld3(lamp.v);
        const float full_radius = lamp.width;
        float radius = fmaxf(full_radius - DIST_EPSILON, 0.0f);
        f3 dir = center - target;
        float dist2 = dot(dir, dir);
        f3 position;
        float distance;
        if (dist2 > radius * radius) {
            float cos_theta_max = sqrt32(fmaxf(1.0f - (radius * radius) / dist2, 0.0f));
            f3 ray_dir = sample_cone(rng, normalize(dir), cos_theta_max);
            if (!sphere_test(center, full_radius, target, ray_dir, distance, position)) {
                distance = 0.0f; // "cheat", shapes/mod.rs:229-236
                position = target;
            }
        } else {
            position = center + sample_sphere(rng) * full_radius;
            distance = magnitude(position - target);
        }
        f3 v = position - target;
        ls.sq_distance = distance * distance;
        ls.direction = normalize(v);
        ls.normal = normalize(position - center);
        float d2 = dot(center - target, center - target);
        if (d2 > full_radius * full_radius) {
            ls.weight = solid_angle(sqrt32(fmaxf(1.0f - (full_radius * full_radius) / d2, 0.0f)));
        } else {
            float cos_in = fabsf(dot(ls.normal, -ls.direction));
            ls.weight = cos_in * lamp.area / ls.sq_distance;
        }
        ls.physical = true;
        ls.material = lamp.material;
        if constexpr (TEX) {
            float latitude, longitude;
            sphere_texture(ls.normal, lamp.t1[0], lamp.t1[1], latitude, longitude, ls.tx, ls.ty);
        }
    } else {
        float u = rng_f32(rng);
        float v = rng_f32(rng);
        f3 p1 = ld3(lamp.p1);
        f3 a = ld3(lamp.p2) - p1, b = ld3(lamp.p3) - p1;
        if (u + v > 1.0f) {
            u = 1.0f - u;
            v = 1.0f - v;
        }
        f3 position = p1 + a * u + b * v;
        f3 delta = position - target;
        float distance = magnitude(delta);
        ls.sq_distance = distance * distance;
        ls.direction = normalize(delta);
        float w = 1.0f - (u + v);
        ls.normal = normalize(ld3(lamp.n1) * w + ld3(lamp.n2) * u + ld3(lamp.n3) * v);
        float cos_in = fabsf(dot(ls.normal, -ls.direction));
        ls.weight = cos_in * lamp.area / ls.sq_distance;
        ls.physical = true;
        ls.material = lamp.material;
        if constexpr (TEX) {
            ls.tx = lamp.t1[0] * w + lamp.t2[0] * u + lamp.t3[0] * v;
            ls.ty = lamp.t1[1] * w + lamp.t2[1] * u + lamp.t3[1] * v;
        }
    }
    return ls;
}

// Pixel a view-plane position exposes to (AspectRatio::to_pixel, film.rs:233-246 + Film::get_pixel :51-54), as the index of
// the pixel inside the launch's film buffer, or PIXEL_NONE when the position maps outside the image / the buffer. It depends
// on the sample's view-plane position alone, so it is worked out once when the sample starts. PYR_FILM_ROWS: the buffer holds
// whole pixel rows; PYR_FILM_TILE_BLOCKS: one (tile_size + 2)^2 block per tile of the launch, the tile's pixels with a ring
// of one pixel around them (pyrite_gpu.h).
constexpr uint32_t PIXEL_NONE = 0xFFFFFFFFu;
DEV uint32_t film_pixel(const RenderLaunch& L, uint32_t tile, float px, float py) {
    const uint32_t width = L.film.width, height = L.film.height;
    uint32_t x, y;
    if (width >= height) {
        float size = (float)width, ratio = __fdiv_rn((float)height, (float)width);
        if (!(fabsf(py) <= ratio)) return PIXEL_NONE;
        x = f32_as_index(__fmul_rn(__fmul_rn(size, __fadd_rn(px, 1.0f)), 0.5f));
        y = f32_as_index(__fmul_rn(__fmul_rn(size, __fadd_rn(py, ratio)), 0.5f));
    } else {
        float size = (float)height, ratio = __fdiv_rn((float)width, (float)height);
        if (!(fabsf(px) <= ratio)) return PIXEL_NONE;
        x = f32_as_index(__fmul_rn(__fmul_rn(size, __fadd_rn(px, ratio)), 0.5f));
        y = f32_as_index(__fmul_rn(__fmul_rn(size, __fadd_rn(py, 1.0f)), 0.5f));
    }
    if (x >= width || y >= height) return PIXEL_NONE;
    if (L.film_layout == PYR_FILM_TILE_BLOCKS) {
        const uint32_t ty = tile / L.tiles_x, tx = tile - ty * L.tiles_x;
        const uint32_t side = L.tile_size + 2u;
        const uint32_t bx = x + 1u - tx * L.tile_size, by = y + 1u - ty * L.tile_size; // wraps to a huge value left of / above the ring
        if (bx >= side || by >= side) return PIXEL_NONE;
        const uint32_t block = (tile - L.tile_begin) / L.tile_stride;
        return (block * side + by) * side + bx;
    }
    if (y < L.film_row_begin || y >= L.film_row_begin + L.film_row_count) return PIXEL_NONE;
    return x + (y - L.film_row_begin) * width;
}

// Film::expose (film.rs:89-95) into a known pixel: wavelength_to_grain (:85-87) + Grain::increment (:145-162) as two
// no-return float atomics (the reference's 5-try CAS may drop samples under contention; atomics never do).
template <bool COUNT>
DEV void expose_grain(const RenderLaunch& L, uint32_t pixel, float wavelength, float brightness, Counters& cnt) {
    if (pixel == PIXEL_NONE) return;
    uint32_t grain = f32_as_index(__fmul_rn(__fsub_rn(wavelength, L.film.wl_start), L.grains_per_wavelength));
    grain = grain < L.film.bins - 1 ? grain : L.film.bins - 1;
    float* g = reinterpret_cast<float*>(L.film_out + ((size_t)pixel * L.film.bins + grain));
    atomicAdd(g, brightness); // value * weight with weight == 1 (simple.rs:95-98)
    atomicAdd(g + 1, 1.0f);
    if (COUNT) cnt.exposures++;
}

// Register-resident state of one path (the hero wavelength; the S-1 companions live in LDS, see Spectral).
struct Path {
    Rng rng;
    uint32_t pixel; // film_pixel of the sample's view-plane position
    f3 o, d;
    float wl, bright, refl;
    uint32_t bounce, events; // bounces made; light_sample_events (tracer.rs:219)
    bool use_additional, sample_light;
};

// Scene pointers a workgroup traverses: HBM/L2, or the LDS copy when the scene is small (LDS_SCENE).
struct SceneView {
    const float4* nodes;
    const float4* prims;
    bool wide = false; // nodes are Node128 (four children); only the resumable traversal walks those
    const float4* pairs = nullptr; // wide tree only: its leaves index triangle pairs (DevPrimPair), not `prims`
};
DEV SceneView wide_or_binary_view(const DevScene& S); // the view the traversal kernels walk (defined below own_scalar)
// A uniform pointer as a scalar value of its own. The kernel arguments arrive by s_load_dwordx8 / x16 and the register
// allocator treats each such load's result as one tuple: when scalar registers run short (they do: ~100 uniform launch and
// scene fields are live across the stage loop) it spills and restores the tuple whole, and the traversal step -- which only
// wants the node and primitive pointers -- was restoring sixteen registers with sixteen v_readlane per step. The move below
// is opaque to the compiler, so its result is a two-register value that stays put or spills alone.
template <typename T>
DEV const T* own_scalar(const T* p) {
    unsigned long long in = (unsigned long long)p, out;
    asm volatile("s_mov_b64 %0, %1" : "=s"(out) : "s"(in));
    return reinterpret_cast<const T*>(out);
}

DEV SceneView wide_or_binary_view(const DevScene& S) {
    SceneView v{reinterpret_cast<const float4*>(S.nodes), reinterpret_cast<const float4*>(S.prims), false, nullptr};
    if (S.wide_nodes != nullptr) {
        v.nodes = reinterpret_cast<const float4*>(S.wide_nodes);
        v.wide = true;
    }
    return v;
}

template <bool LDS_SCENE>
DEV SceneView stage_scene(const DevScene& S, float* lds, uint32_t lds_floats_before, bool resumable = false) {
    SceneView v{reinterpret_cast<const float4*>(S.nodes), reinterpret_cast<const float4*>(S.prims)};
    if (!LDS_SCENE && resumable && S.wide_nodes != nullptr) {
        v.nodes = reinterpret_cast<const float4*>(S.wide_nodes);
        v.wide = true;
    }
    if (!LDS_SCENE && resumable && S.wide_nodes != nullptr && S.pair_prims != nullptr) {
        v.nodes = reinterpret_cast<const float4*>(S.wide_pair_nodes);
        v.pairs = reinterpret_cast<const float4*>(S.pair_prims);
    }
    if constexpr (!LDS_SCENE) {
        v.nodes = own_scalar(v.nodes);
        v.prims = own_scalar(v.prims);
        v.pairs = own_scalar(v.pairs);
    }
    if constexpr (LDS_SCENE) {
        float4* staged = reinterpret_cast<float4*>(lds + lds_floats_before);
        const uint32_t node_vecs = S.num_nodes * 4, prim_vecs = S.num_prims * 3;
        for (uint32_t i = threadIdx.x; i < node_vecs; i += BLOCK) staged[i] = v.nodes[i];
        for (uint32_t i = threadIdx.x; i < prim_vecs; i += BLOCK) staged[node_vecs + i] = v.prims[i];
        __syncthreads();
        v.nodes = staged;
        v.prims = staged + node_vecs;
    }
    return v;
}

// Spectrum tables (PyrSpectrum records + sample data) are small and read with a different index by every lane and
// wavelength: on C3 the BVH traffic evicts them from L1 and every lookup becomes two dependent L2 round trips (measured:
// the 10-wavelength work cost 139 of 250 ms). When they fit they are copied into LDS once per workgroup; the scene copy
// handed to the device functions then points at the LDS copy (generic pointers: the loads become flat loads that resolve
// to LDS).
// TABLES is a compile-time statement of where the tables live (0: HBM, as the kernel arguments point; 1: staged into LDS;
// -1: decided at run time from S.lds_table_floats). With a run-time choice the pointers are generic and every table access
// is a flat_load that must drain both vmcnt and lgkmcnt (MI355X_MICROARCH.md: flat completes out of order); with a
// compile-time one they are global_load / s_load or ds_read.
// COPY = false only rebuilds the record that points at the staged copies (a dozen scalar operations): the stage-scheduled
// kernel does that at the head of every phase from the kernel-argument segment (scene_from_kernarg) rather than keep the
// record's ~40 uniform words alive -- and spilled to VGPR lanes -- across the whole stage loop.
template <int TABLES, bool COPY = true>
DEV DevScene stage_tables(const DevScene& S, float* lds, uint32_t lds_floats_before) {
    DevScene local = S;
    if (TABLES == 1 || (TABLES == -1 && S.lds_table_floats != 0)) {
        float* dst = lds + lds_floats_before;
        auto stage = [&](const void* src, uint32_t floats) {
            const float* from = reinterpret_cast<const float*>(src);
            if (COPY)
                for (uint32_t i = threadIdx.x; i < floats; i += BLOCK) dst[i] = from[i];
            float* at = dst;
            dst += floats;
            return at;
        };
        local.spectra = reinterpret_cast<const PyrSpectrum*>(stage(S.spectra, S.num_spectra * (uint32_t)(sizeof(PyrSpectrum) / sizeof(float))));
        local.spectrum_data = stage(S.spectrum_data, S.num_spectrum_floats);
#ifndef PYR_LDS_SMALL_TABLES
#define PYR_LDS_SMALL_TABLES 1
#endif
#if PYR_LDS_SMALL_TABLES
        // The records a bounce walks through one after the other -- material -> component -> program, and the lamp of a
        // next-event estimation -- are a few hundred bytes per scene, but every step of that chain was an L2 round trip (the
        // BVH traffic keeps evicting them from L1) in phases that run at a third of the wave's width: staged with the spectra.
        local.materials = reinterpret_cast<const PyrMaterial*>(stage(S.materials, S.num_materials * (uint32_t)(sizeof(PyrMaterial) / sizeof(float))));
        local.components = reinterpret_cast<const PyrComponent*>(stage(S.components, S.num_components * (uint32_t)(sizeof(PyrComponent) / sizeof(float))));
        local.programs = reinterpret_cast<const DevProgram*>(stage(S.programs, S.num_programs * (uint32_t)(sizeof(DevProgram) / sizeof(float))));
        local.lamps = reinterpret_cast<const DevLamp*>(stage(S.lamps, S.num_lamps * (uint32_t)(sizeof(DevLamp) / sizeof(float))));
#endif
        if (COPY) __syncthreads();
    }
    return local;
}

// Start of render_tile's loop body (simple.rs:78-107) for iteration `iteration` of raster tile `tile`.
template <bool COMPANION_STATE = true>
DEV void start_sample(const RenderLaunch& L, uint32_t tile, uint64_t iteration, const TileArea& area, Path& p, Spectral& spec) {
    const uint32_t SS = L.spectrum_samples;
    p.rng = rng_seed(L.seed, tile, iteration);
    // Tile::sample_point, renderer/algorithm.rs:113-119 (unfused: decides the pixel)
    const float px = __fadd_rn(area.from_x, __fmul_rn(area.size_x, rng_f32(p.rng)));
    const float py = __fadd_rn(area.from_y, __fmul_rn(area.size_y, rng_f32(p.rng)));
    p.pixel = film_pixel(L, tile, px, py);
    // Camera::ray_towards, cameras.rs:70-97
    {
        float focus_x = px / L.camera.view_plane * L.camera.focus_distance;
        float focus_y = py / L.camera.view_plane * L.camera.focus_distance;
        f3 target = mk(focus_x, -focus_y, -L.camera.focus_distance);
        f3 origin = mk(0, 0, 0), direction = target;
        if (L.camera.aperture > 0.0f) {
            float sqrt_r = sqrt32(L.camera.aperture * rng_f32(p.rng));
            float psi = PI_F * 2.0f * rng_f32(p.rng);
            origin = mk(sqrt_r * cos32(psi), sqrt_r * sin32(psi), 0.0f);
            direction = target - origin;
        }
        p.o = transform_point(L.camera.cam_to_world, origin);
        p.d = transform_vector(L.camera.cam_to_world, normalize(direction));
    }
    // Film::sample_many_wavelengths (film.rs:68-83) + hero pick by swap_remove (simple.rs:105-107)
    {
        float step_size = L.film.wl_width / (float)SS;
        float from = L.film.wl_start;
        for (uint32_t k = 0; k < SS; ++k) {
            float to = __fadd_rn(from, step_size);
            spec.wl(k) = rng_range_f32(p.rng, from, to);
            from = to;
        }
        uint32_t hero = rng_range_usize(p.rng, SS);
        p.wl = spec.wl(hero);
        spec.wl(hero) = spec.wl(SS - 1);
        if constexpr (COMPANION_STATE)
            for (uint32_t k = 0; k + 1 < SS; ++k) {
                spec.bright(k) = 0.0f;
                spec.refl(k) = 1.0f;
            }
    }
    p.bright = 0.0f;
    p.refl = 1.0f;
    p.use_additional = true;
    p.sample_light = true; // tracer.rs:218-219
    p.events = 0;
    p.bounce = 0;
}

// Maps a chunk number of the launch to (tile, first iteration, tile rectangle): chunk c belongs to the launch's
// (c / chunks_per_tile)-th tile. Returns false when the lane's iteration lies beyond the tile's iteration count (the last
// chunk of a tile, and the empty chunk numbers of a tile cut by the image border).
DEV bool locate_chunk(const RenderLaunch& L, uint32_t chunk, uint32_t lane, uint32_t& tile, uint64_t& iteration, TileArea& area) {
    const uint32_t k = chunk / L.chunks_per_tile, within = chunk - k * L.chunks_per_tile;
    tile = L.tile_begin + k * L.tile_stride;
    const uint32_t ty = tile / L.tiles_x, tx = tile - ty * L.tiles_x;
    const uint32_t sx = tx * L.tile_size, sy = ty * L.tile_size;
    const uint32_t w = min(L.film.width - sx, L.tile_size), h = min(L.film.height - sy, L.tile_size);
    const uint64_t iterations = (uint64_t)w * h * L.pixel_samples;
    iteration = (uint64_t)within * 64u + lane;
    if (iteration >= iterations) return false;
    area = to_view_area(sx, sy, w, h, L.film.width, L.film.height);
    return true;
}

// Developer build only (-DPYR_PHASE_PROFILE, tools/phase_profile.py): lap timers of the synchronous walk's sections.
struct SyncProf {
    unsigned long long section[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last = 0;
};
#ifdef PYR_PHASE_PROFILE
__device__ unsigned long long g_phase_prof[32]; // [0..15] as tools/phase_profile.py reads them; [16..31] render_kernel_px seat census
#define SLAP(sp, i)                                 \
    {                                               \
        const unsigned long long now_ = clock64();  \
        (sp).section[i] += now_ - (sp).last;        \
        (sp).last = now_;                           \
    }
#else
#define SLAP(sp, i)
#endif

// Wave priority per section of the synchronous walk (s_setprio; see PYR_PRIO_* of the stage scheduler for why): traversal on
// top, the next-event estimation's own arithmetic below it, shading below that, exposure and sample start at the bottom.
// C2 803 -> 827 Msamples/s. All four equal = no instruction emitted.
#ifndef PYR_SYNC_PRIO_T
#define PYR_SYNC_PRIO_T 3
#endif
#ifndef PYR_SYNC_PRIO_N
#define PYR_SYNC_PRIO_N 2
#endif
#ifndef PYR_SYNC_PRIO_S
#define PYR_SYNC_PRIO_S 1
#endif
#ifndef PYR_SYNC_PRIO_E
#define PYR_SYNC_PRIO_E 0
#endif
#if PYR_SYNC_PRIO_T != PYR_SYNC_PRIO_N || PYR_SYNC_PRIO_N != PYR_SYNC_PRIO_S || PYR_SYNC_PRIO_S != PYR_SYNC_PRIO_E
#define SYNC_PRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define SYNC_PRIO(x)
#endif
// One iteration of tracer::trace's loop (tracer.rs:221-344) with `contribute` (renderer/algorithm.rs:14-100) applied online.
// Returns true when the path has ended (emission, miss). Does not touch p.bounce.
template <bool COUNT, bool INTERP>
DEV bool bounce_step(const DevScene& S, const RenderLaunch& L, const SceneView& view, Path& p, Spectral& spec, int* stack, Counters& cnt, SyncProf& sp) {
    const uint32_t n_add = L.spectrum_samples - 1;
    Hit hit;
    if (COUNT) cnt.extension_rays++;
    const f3 ray_o = p.o, ray_d = p.d;
    SYNC_PRIO(PYR_SYNC_PRIO_T);
    const bool found = traverse<COUNT, false>(S, view.nodes, view.prims, ray_o, ray_d, 0.0f, hit, stack, cnt);
    SYNC_PRIO(PYR_SYNC_PRIO_S);
    SLAP(sp, 1);
    if (!found) {
        // miss: first matching directional lamp (trace_directional, tracer.rs:444-459) or the sky; dispersed = false
        uint32_t color = S.sky_program;
        if (p.sample_light) {
            for (uint32_t i = 0; i < S.num_lamps; ++i) {
                const DevLamp& l = S.lamps[i];
                if (l.kind == PYR_LAMP_DIRECTIONAL && dot(ld3(l.v), ray_d) >= l.width) {
                    color = l.color_program;
                    break;
                }
            }
        }
        const Prepared q_prog = prepare_program<INTERP>(S, color);
        VmInput in{p.wl, -ray_d, ray_d};
        p.bright += eval_prepared<INTERP>(S, q_prog, in) * 1.0f * p.refl;
        if (p.use_additional)
            for (uint32_t k = 0; k < n_add; ++k) {
                in.wavelength = spec.wl(k);
                spec.bright(k) += eval_prepared<INTERP>(S, q_prog, in) * 1.0f * spec.refl(k);
            }
        return true;
    }
    if (COUNT) cnt.shaded_hits++;
    f3 position, normal;
    uint32_t material_id;
    surface_at(S, hit, ray_o, ray_d, position, normal, material_id);
    const PyrMaterial material = S.materials[material_id];
    const uint32_t pick = rng_choose(p.rng, material.num_components); // choose_component, materials/mod.rs:48-54
    const PyrComponent comp = S.components[material.first_component + pick];
    // get_probability, materials/mod.rs:238-248
    float component_probability = comp.selection_compensation;
    bool normal_dispersed = false;
    if (comp.probability_program >= 0) {
        VmInput pin{p.wl, normal, ray_d};
        component_probability = run_program<INTERP>(S, (uint32_t)comp.probability_program, pin) * comp.selection_compensation;
        normal_dispersed = S.programs[comp.probability_program].reads_wavelength != 0;
    }

    if (comp.bsdf == PYR_BSDF_EMISSIVE) { // Scattering::Emitted, tracer.rs:303-318
        if (p.sample_light) {
            p.use_additional = !normal_dispersed && p.use_additional;
            const Prepared q_prog = prepare_program<INTERP>(S, comp.color_program);
            VmInput in{p.wl, normal, ray_d};
            p.bright += eval_prepared<INTERP>(S, q_prog, in) * component_probability * p.refl;
            if (p.use_additional)
                for (uint32_t k = 0; k < n_add; ++k) {
                    in.wavelength = spec.wl(k);
                    spec.bright(k) += eval_prepared<INTERP>(S, q_prog, in) * component_probability * spec.refl(k);
                }
        }
        return true;
    }

    // SurfaceBsdfType::scatter, materials/mod.rs:344-359
    f3 out_direction;
    float scatter_probability = 1.0f;
    bool dispersed = false, has_brdf = false;
    if (comp.bsdf == PYR_BSDF_DIFFUSE) { // diffuse.rs:8-25
        f3 n = dot(ray_d, normal) < 0.0f ? normal : -normal;
        out_direction = sample_hemisphere(p.rng, n);
        has_brdf = true;
    } else if (comp.bsdf == PYR_BSDF_MIRROR) { // mirror.rs:5-21
        f3 n = dot(ray_d, normal) < 0.0f ? normal : -normal;
        float perp = dot(ray_d, n) * 2.0f;
        out_direction = ray_d - n * perp;
    } else { // refractive.rs:6-37
        dispersed = comp.dispersion != 0.0f || comp.env_dispersion != 0.0f;
        float ior = comp.ior, env_ior = comp.env_ior;
        if (dispersed) {
            float wl = p.wl * 0.001f;
            ior = comp.ior + comp.dispersion / (wl * wl);
            env_ior = comp.env_ior + comp.env_dispersion / (wl * wl);
        }
        refract(ior, env_ior, ray_d, normal, p.rng, out_direction, scatter_probability);
    }

    // contribute, non-emission bounce, first half (algorithm.rs:48-63): reflectance *= color * probability
    const float bounce_probability = scatter_probability * component_probability; // tracer.rs:296
    p.use_additional = !(dispersed || normal_dispersed) && p.use_additional;       // simple.rs:122-123, tracer.rs:290
    {
        const Prepared q_prog = prepare_program<INTERP>(S, comp.color_program);
        VmInput in{p.wl, normal, ray_d};
        p.refl *= eval_prepared<INTERP>(S, q_prog, in) * bounce_probability;
        if (p.use_additional)
            for (uint32_t k = 0; k < n_add; ++k) {
                in.wavelength = spec.wl(k);
                spec.refl(k) *= eval_prepared<INTERP>(S, q_prog, in) * bounce_probability;
            }
    }

    SLAP(sp, 2);
    SYNC_PRIO(PYR_SYNC_PRIO_N);
    // next-event estimation gate, tracer.rs:257-280
    if (p.events < 2) {
        p.sample_light = !has_brdf || L.light_samples == 0;
        if (has_brdf) {
            p.events += 1;
            if (S.num_lamps > 0) { // trace_direct, tracer.rs:347-442
                const uint32_t lamp_index = rng_range_usize(p.rng, S.num_lamps); // pick_lamp, world.rs:301-305
                const DevLamp& lamp = S.lamps[lamp_index];
                const float lamp_probability = 1.0f / (float)S.num_lamps;
                const f3 nff = dot(ray_d, normal) < 0.0f ? normal : -normal;
                const float probability = 1.0f / ((float)L.light_samples * 2.0f * PI_F * lamp_probability);
                // Companion wavelengths of the unblocked light samples. Without the interpreter a colour program is a function
                // of the wavelength alone (a constant or spectrum [* c]), so consecutive samples that show the same program
                // need its value at each companion wavelength only once: their probabilities are parked (up to four) and the
                // companions are brought up to date in one pass -- each brightness still receives the same addends in the
                // same order as in `contribute` (algorithm.rs:65-90), so the result is bit-identical, at a quarter of the
                // spectrum look-ups.
                float parked0 = 0.0f, parked1 = 0.0f, parked2 = 0.0f, parked3 = 0.0f;
                uint32_t parked = 0, parked_color = 0;
                auto flush_parked = [&]() {
                    if (parked == 0) return;
                    const Prepared q_prog = prepare_program<INTERP>(S, parked_color);
                    VmInput in{0.0f, mk(0, 0, 0), mk(0, 0, 0)};
                    for (uint32_t k = 0; k < n_add; ++k) {
                        in.wavelength = spec.wl(k);
                        const float color = eval_prepared<INTERP>(S, q_prog, in);
                        const float refl = spec.refl(k);
                        float b = spec.bright(k);
                        b += color * parked0 * refl;
                        if (parked > 1) b += color * parked1 * refl;
                        if (parked > 2) b += color * parked2 * refl;
                        if (parked > 3) b += color * parked3 * refl;
                        spec.bright(k) = b;
                    }
                    parked = 0;
                };
                for (uint32_t ls_i = 0; ls_i < L.light_samples; ++ls_i) {
                    const LampSample ls = lamp_sample(lamp, p.rng, position);
                    const float cos_out = fmaxf(dot(nff, ls.direction), 0.0f);
                    if (!(cos_out > 0.0f)) continue;
                    if (COUNT) cnt.shadow_rays++;
                    const float limit = ls.sq_distance >= 0.0f ? ls.sq_distance - DIST_EPSILON : PYR_INF;
                    Hit shadow_hit;
                    SLAP(sp, 3);
                    SYNC_PRIO(PYR_SYNC_PRIO_T);
                    const bool is_blocked = traverse<COUNT, true>(S, view.nodes, view.prims, position, ls.direction, limit, shadow_hit, stack, cnt);
                    SYNC_PRIO(PYR_SYNC_PRIO_N);
                    SLAP(sp, 4);
                    if (is_blocked) continue;
                    uint32_t l_color = ls.color;
                    float material_probability = 1.0f;
                    bool l_dispersed = false;
                    f3 target_normal = -ls.direction;
                    if (ls.physical) {
                        const PyrMaterial lm = S.materials[ls.material];
                        const uint32_t e_pick = rng_choose(p.rng, lm.num_emissive); // choose_emissive, materials/mod.rs:56-62
                        const PyrComponent ec = S.components[lm.first_emissive + e_pick];
                        material_probability = ec.selection_compensation;
                        if (ec.probability_program >= 0) {
                            VmInput pin{p.wl, ls.normal, ls.direction};
                            material_probability = run_program<INTERP>(S, (uint32_t)ec.probability_program, pin) * ec.selection_compensation;
                            l_dispersed = S.programs[ec.probability_program].reads_wavelength != 0;
                        }
                        l_color = ec.color_program;
                        target_normal = ls.normal;
                    }
                    const float scale = ls.weight * probability * (2.0f * fabsf(dot(ls.direction, nff))); // lambertian, diffuse.rs:27-29
                    const float l_probability = scale * material_probability;
                    // contribute, direct light (algorithm.rs:65-90)
                    const Prepared q_prog = prepare_program<INTERP>(S, l_color);
                    VmInput in{p.wl, target_normal, ls.direction};
                    p.bright += eval_prepared<INTERP>(S, q_prog, in) * l_probability * p.refl;
                    if (p.use_additional && !l_dispersed) {
                        if constexpr (INTERP) {
                            for (uint32_t k = 0; k < n_add; ++k) {
                                in.wavelength = spec.wl(k);
                                spec.bright(k) += eval_prepared<INTERP>(S, q_prog, in) * l_probability * spec.refl(k);
                            }
                        } else {
                            if (parked != 0 && (parked == 4 || parked_color != l_color)) flush_parked();
                            parked_color = l_color;
                            if (parked == 0) parked0 = l_probability;
                            if (parked == 1) parked1 = l_probability;
                            if (parked == 2) parked2 = l_probability;
                            if (parked == 3) parked3 = l_probability;
                            parked++;
                        }
                    }
                }
                flush_parked();
                SLAP(sp, 5);
            }
        }
    } else {
        p.sample_light = true;
    }

    // contribute, second half (algorithm.rs:92-98): reflectance *= brdf (2|n.out| for diffuse, tracer.rs:175-183)
    if (has_brdf) {
        const float brdf = 2.0f * fabsf(dot(out_direction, normal));
        p.refl *= brdf;
        if (p.use_additional)
            for (uint32_t k = 0; k < n_add; ++k) spec.refl(k) *= brdf;
    }
    p.o = position;
    p.d = out_direction;
    SLAP(sp, 6);
    return false;
}

// simple.rs:133-139: expose the hero always, the companions unless a bounce dispersed.
template <bool COUNT>
DEV void finish_path(const RenderLaunch& L, const Path& p, Spectral& spec, Counters& cnt) {
    expose_grain<COUNT>(L, p.pixel, p.wl, p.bright, cnt);
    if (p.use_additional)
        for (uint32_t k = 0; k + 1 < L.spectrum_samples; ++k) expose_grain<COUNT>(L, p.pixel, spec.wl(k), spec.bright(k), cnt);
}

// Bounce-synchronous integrator: a wave takes a chunk of 64 samples and walks the 64 paths bounce by bounce. Next-event
// estimation happens in the first two diffuse events of a path (tracer.rs:257), i.e. at the same time for all 64 lanes
// of a diffuse scene, which is why this plain walk beats per-lane refill on C2:
//   * refilling a lane as soon as its path ends (one bounce per loop turn)                         0.60x
//   * parking survivors of the first two bounces in a ballot-compacted HBM queue for a tail kernel  0.93x
// (MI355X, C2, 64 spp; both were built and measured, see DESIGN.md "Scheduling experiments").
// The launch record as the kernel-argument segment holds it, behind a pointer the compiler cannot see through. The stage
// loop keeps ~100 uniform scene / launch values alive; there are 104 scalar registers, so the allocator parks the rest in
// lanes of a VGPR and fetches them back with v_readlane -- vector instructions, sixteen in a row where a phase wants the
// camera -- in a kernel that is bound by vector issue. Read through this reference at the head of a phase the fields are
// s_load'ed from the (scalar-cached) argument segment where they are used and die with the phase.
#ifndef PYR_RELOAD_LAUNCH
#define PYR_RELOAD_LAUNCH 1
#endif
typedef __attribute__((address_space(4))) const RenderLaunch* kernarg_launch_ptr;
constexpr size_t kLaunchKernargOffset = (sizeof(DevScene) + alignof(RenderLaunch) - 1) / alignof(RenderLaunch) * alignof(RenderLaunch);
typedef __attribute__((address_space(4))) const DevScene* kernarg_scene_ptr;
DEV const DevScene& scene_from_kernarg(const DevScene& by_value) { // the scene record: the first kernel argument
    if (!PYR_RELOAD_LAUNCH) return by_value;
    unsigned long long at = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(at));
    return *(const DevScene*)(kernarg_scene_ptr)at;
}
DEV const RenderLaunch& launch_from_kernarg(const RenderLaunch& by_value) {
    if (!PYR_RELOAD_LAUNCH) return by_value;
    unsigned long long at = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr() + kLaunchKernargOffset;
    asm volatile("" : "+s"(at));
    return *(const RenderLaunch*)(kernarg_launch_ptr)at;
}

template <bool COUNT, bool INTERP, bool LDS_SCENE, bool LDS_TABLES>
__global__ __launch_bounds__(BLOCK, 4) void render_kernel(DevScene S0, RenderLaunch L) {
    extern __shared__ float lds[];
    const uint32_t SS = L.spectrum_samples;
    Spectral spec{lds + threadIdx.x, SS};
    int* stack = reinterpret_cast<int*>(lds + 3 * SS * BLOCK) + threadIdx.x;
    Counters cnt{};
    const uint32_t lds_base_floats = (3 * SS + S0.stack_depth) * BLOCK;
    const SceneView view = stage_scene<LDS_SCENE>(S0, lds, lds_base_floats);
    const DevScene S = stage_tables<LDS_TABLES ? 1 : 0>(S0, lds, lds_base_floats + (LDS_SCENE ? (S0.num_nodes * 16 + S0.num_prims * 12) : 0));

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t waves_per_block = BLOCK / 64;
    const uint32_t wave = blockIdx.x * waves_per_block + (threadIdx.x >> 6);
    const uint32_t total_waves = gridDim.x * waves_per_block;

    SyncProf sp;
#ifdef PYR_PHASE_PROFILE
    sp.last = clock64();
#endif
    for (uint32_t chunk = L.chunk_begin + wave; chunk < L.chunk_end; chunk += total_waves) {
        uint32_t tile;
        uint64_t iteration;
        TileArea area;
        if (!locate_chunk(L, chunk, lane, tile, iteration, area)) continue;
        Path p{};
        start_sample(launch_from_kernarg(L), tile, iteration, area, p, spec);
        if (COUNT) cnt.samples++;
        SLAP(sp, 0);
        while (p.bounce < L.bounces) {
            const bool ended = bounce_step<COUNT, INTERP>(S, launch_from_kernarg(L), view, p, spec, stack, cnt, sp);
            p.bounce++;
            if (ended) break;
        }
        SLAP(sp, 6);
        SYNC_PRIO(PYR_SYNC_PRIO_E);
        finish_path<COUNT>(launch_from_kernarg(L), p, spec, cnt);
        SLAP(sp, 7);
    }
#ifdef PYR_PHASE_PROFILE
    if (lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&g_phase_prof[i], sp.section[i]);
#endif
    flush_counters<COUNT>(cnt, L.counters);
}

// =================================================================================================
// Stage-scheduled integrator (render_kernel_sm)
//
// The bounce-synchronous walk above makes every lane wait for the slowest lane of each phase: on C3 (819 k triangles)
// the traversal of a ray that enters the mesh takes ~5x the steps of one that hits a wall and lane occupancy drops to 14 %
// (SQ_THREAD_CYCLES_VALU / 64 SQ_ACTIVE_INST_VALU); on C2 it is 33 %. Here every lane runs its path as a state machine
// whose states are the phases of render_tile / trace / trace_direct:
//     NEW -> TRAV(ext) -> SHADE -> { NEE -> TRAV(shadow) -> NEE ... } -> TRAV(ext) ... -> EXPOSE -> NEW
// and the WAVE decides, from ballots, which phase code to run next: a phase runs when at least sm_phase_lanes lanes wait for
// it, or when it is the most wanted one. Traversal is resumable (node index, stack pointer and closest hit stay in
// registers, the stack in LDS) and advances sm_trav_steps node/leaf steps per turn, so a lane that finishes a short ray goes
// on to shading while its neighbours keep walking the tree, and a lane whose path ends refills itself with the next
// sample. Results are those of the synchronous walk bit for bit: per-path order of operations and RNG draws is unchanged.
// =================================================================================================
// Developer build only (-DPYR_PHASE_PROFILE, tools/phase_profile.py): per-phase wave cycles / active lanes / turns.
#ifdef PYR_PHASE_PROFILE
#define PROF_DECL unsigned long long prof_c[4] = {0, 0, 0, 0}, prof_l[4] = {0, 0, 0, 0}, prof_n[4] = {0, 0, 0, 0}
#define PROF_BEGIN(ph, cond) const unsigned long long prof_t0_##ph = clock64(); prof_l[ph] += __popcll(ballot64(cond)); prof_n[ph]++
#define PROF_END(ph) prof_c[ph] += clock64() - prof_t0_##ph
#define PROF_LANES(ph, cond) prof_l[ph] += __popcll(ballot64(cond)); prof_n[ph]++
#define PROF_NOW() clock64()
#define PROF_EXTRA(idx, cycles) if ((threadIdx.x & 63u) == 0) atomicAdd(&g_phase_prof[idx], (unsigned long long)(cycles))
#define PROF_FLUSH()                                                                                  \
    if ((threadIdx.x & 63u) == 0)                                                                     \
        for (int i = 0; i < 4; ++i) {                                                                 \
            atomicAdd(&g_phase_prof[i], prof_c[i]);                                                   \
            atomicAdd(&g_phase_prof[4 + i], prof_l[i]);                                               \
            atomicAdd(&g_phase_prof[8 + i], prof_n[i]);                                               \
        }
#else
#define PROF_DECL
#define PROF_BEGIN(ph, cond)
#define PROF_END(ph)
#define PROF_LANES(ph, cond)
#define PROF_NOW() 0ull
#define PROF_EXTRA(idx, cycles)
#define PROF_FLUSH()
#endif

enum Stage : uint32_t { ST_NEW = 0, ST_TRAV = 1, ST_SHADE = 2, ST_NEE = 3, ST_EXPOSE = 4, ST_DONE = 5 };

struct Trav { // resumable World::intersect
    f3 o, d, inv;
    // Byte offsets (0 or 48) of the planes the ray meets FIRST on each axis inside a Node128: lo_* for a positive direction
    // component, hi_* (48 bytes further) for a negative one; the far plane is at the offset ^ 48. Set with `inv` where a ray is
    // about to be stepped (trav_ray_signs); the straight-line step loads near and far planes directly (wide_node_children).
    uint32_t nx = 0, ny = 0, nz = 0;
    // `closest` is the distance boxes are cut off at: the closest hit so far for an extension ray; for a shadow ray the square
    // root of the blocking limit with its margin (traverse<>'s limit_cull, DevScene::shadow_margin) -- one comparison serves both kinds of ray, and
    // the half-ulp of the root is far inside that margin. A shadow ray never reads it as a hit distance.
    float limit, closest;
    int node, sp;
    uint32_t shape;
    float u, v;
    bool shadow, blocked;
};

DEV void trav_ray_signs(Trav& t) {
    t.nx = (__float_as_uint(t.d.x) >> 31) * 48u;
    t.ny = (__float_as_uint(t.d.y) >> 31) * 48u;
    t.nz = (__float_as_uint(t.d.z) >> 31) * 48u;
}
// Puts a query whose ray, limit and plane results are set at the root of the tree.
// v_sqrt_f32 alone: the ulp it may be off by is far inside the margin (0.1 % at least). +inf stays +inf; a negative limit gives NaN: nothing passes
DEV float shadow_cutoff(float limit, float margin) { return __builtin_amdgcn_sqrtf(limit * margin + 1.0e-3f); }
// (t.inv is NOT set here: the kernels compute it where a ray is about to be stepped -- the stage scheduler at every entry of
// its traversal phase -- so that the three registers are free while the other phases run.)
DEV void trav_restart(Trav& t, float shadow_margin) {
    if (t.shadow) t.closest = shadow_cutoff(t.limit, shadow_margin);
    t.blocked = false;
    t.node = 0;
    t.sp = 0;
}

// Planes first (world.rs:277-285), then the tree from the root. Returns true when the query is already decided.
template <bool COUNT>
DEV bool trav_begin(const DevScene& S, Trav& t, f3 o, f3 d, bool shadow, float limit, Counters& cnt) {
    t.o = o;
    t.d = d;
    t.shadow = shadow;
    t.blocked = false;
    t.limit = shadow ? limit : -PYR_INF; // an extension ray is blocked by nothing: `dist * dist < t.limit` never holds, and the leaf tests need no ray kind
    t.closest = PYR_INF;
    t.shape = PYR_HIT_NONE;
    t.u = t.v = 0.0f;
    for (uint32_t i = 0; i < S.num_planes; ++i) {
        const float* pl = S.planes + 8 * i;
        float dist;
        f3 point;
        if (COUNT) cnt.plane_tests++;
        if (plane_test(ld3(pl), ld3(pl + 3), o, d, dist, point)) {
            if (shadow) {
                if (dist > DIST_EPSILON && dist * dist < limit) {
                    t.blocked = true;
                    return true;
                }
            } else if (dist > DIST_EPSILON && dist < t.closest) {
                t.closest = dist;
                t.shape = ((uint32_t)PYR_SHAPE_PLANE << 30) | i;
            }
        }
    }
    trav_restart(t, S.shadow_margin);
    return false;
}

// Traversal stack of the resumable walk: the first `lds_entries` levels live in LDS ([level][lane], conflict free), deeper
// levels in the lane's scratch. A SAH tree over 819 k triangles is ~30 levels deep but a ray rarely holds more than a
// dozen pending subtrees, so a short LDS part keeps the LDS footprint (and with it the waves per CU) independent of the
// tree's worst-case depth while the deep end is touched by a few rays only.
typedef __attribute__((address_space(3))) int lds_int;
struct TravStack {
    lds_int* lds; // + threadIdx.x; an LDS pointer by type: a generic one makes every push and pop a flat_ access that drains both counters
    int lds_entries;
    int* deep; // the lane's scratch part: kMaxStackDepth entries, or one where the whole stack is known to be in LDS (a scene staged in LDS)
    DEV void push(int sp, int value) {
        if (sp < lds_entries)
            lds[sp * BLOCK] = value;
        else
            deep[sp - lds_entries] = value;
    }
    DEV int pop(int sp) const { return sp < lds_entries ? lds[sp * BLOCK] : deep[sp - lds_entries]; }
};

// One node visit or one leaf. Returns true when the traversal has finished. Same tests, same order as traverse<>.
// One visit of a four-child node (bvh.h Node128). The twelve plane distances of two children at a time are v_pk_fma_f32;
// the children that are hit are ordered by entry distance with a five-comparator network, the nearest is entered and the
// others are pushed far to near, so they pop nearest first. Returns true when the traversal has finished.
// The four box tests of a node and the order of its children: c[] = the children that are hit, nearest first (entry distance
// e[]), the others INT32_MIN at the end.
// SIGNED: the first three vectors are the planes the ray meets first on each axis and the other three the ones it meets last
// (the caller picked them by the sign of the direction, trav_ray_signs), so entry = max of the near distances and exit = min of
// the far ones: 12 min / max per node instead of 36. t = bound * inv - o * inv is monotonic in `bound`, so the near distance IS
// min(t_lo, t_hi) of the unsigned form, value for value -- except where 0 * inf or inf - inf makes one of the pair NaN (a
// direction component of exactly zero): the unsigned form then takes the other one for both entry and exit and can reject a
// box the ray is inside of on that axis; here a NaN plane distance drops out of max3 / min3 and the axis does not constrain.
template <bool COUNT, bool SIGNED = false>
DEV void wide_node_children(const float4 lx, const float4 ly, const float4 lz, const float4 hx, const float4 hy, const float4 hz, const float4 ch, const Trav& t,
                            Counters& cnt, float (&e)[4], int (&c)[4]) {
    const f2v ix = {t.inv.x, t.inv.x}, iy = {t.inv.y, t.inv.y}, iz = {t.inv.z, t.inv.z};
    const float ox = -(t.o.x * t.inv.x), oy = -(t.o.y * t.inv.y), oz = -(t.o.z * t.inv.z);
    const f2v nox = {ox, ox}, noy = {oy, oy}, noz = {oz, oz};
    c[0] = __float_as_int(ch.x), c[1] = __float_as_int(ch.y), c[2] = __float_as_int(ch.z), c[3] = __float_as_int(ch.w);
    {
        const f2v alx = __builtin_elementwise_fma((f2v){lx.x, lx.y}, ix, nox), ahx = __builtin_elementwise_fma((f2v){hx.x, hx.y}, ix, nox);
        const f2v aly = __builtin_elementwise_fma((f2v){ly.x, ly.y}, iy, noy), ahy = __builtin_elementwise_fma((f2v){hy.x, hy.y}, iy, noy);
        const f2v alz = __builtin_elementwise_fma((f2v){lz.x, lz.y}, iz, noz), ahz = __builtin_elementwise_fma((f2v){hz.x, hz.y}, iz, noz);
        const f2v blx = __builtin_elementwise_fma((f2v){lx.z, lx.w}, ix, nox), bhx = __builtin_elementwise_fma((f2v){hx.z, hx.w}, ix, nox);
        const f2v bly = __builtin_elementwise_fma((f2v){ly.z, ly.w}, iy, noy), bhy = __builtin_elementwise_fma((f2v){hy.z, hy.w}, iy, noy);
        const f2v blz = __builtin_elementwise_fma((f2v){lz.z, lz.w}, iz, noz), bhz = __builtin_elementwise_fma((f2v){hz.z, hz.w}, iz, noz);
        const float tl[4][3] = {{alx.x, aly.x, alz.x}, {alx.y, aly.y, alz.y}, {blx.x, bly.x, blz.x}, {blx.y, bly.y, blz.y}};
        const float th[4][3] = {{ahx.x, ahy.x, ahz.x}, {ahx.y, ahy.y, ahz.y}, {bhx.x, bhy.x, bhz.x}, {bhx.y, bhy.y, bhz.y}};
        // no branches in here: the backend knows an fma result is never a signalling NaN only inside one basic block, and quiets
        // every min / max operand again (a v_max x, x, x each) on the far side of one
        for (int k = 0; k < 4; ++k) {
            const float tmin = SIGNED ? fmaxf(fmaxf(tl[k][0], tl[k][1]), tl[k][2])
                                      : fmaxf(fmaxf(fminf(tl[k][0], th[k][0]), fminf(tl[k][1], th[k][1])), fminf(tl[k][2], th[k][2]));
            const float tmax = SIGNED ? fminf(fminf(th[k][0], th[k][1]), th[k][2])
                                      : fminf(fminf(fmaxf(tl[k][0], th[k][0]), fmaxf(tl[k][1], th[k][1])), fmaxf(tl[k][2], th[k][2]));
            const float entry = fmaxf(tmin, 0.0f);
            if (COUNT) cnt.box_tests += c[k] != INT32_MIN ? 1u : 0u;
            // tmax >= max(tmin, 0) is (tmax >= tmin) & (tmax >= 0); an unused slot's box is NaN (bvh.cpp) and fails it
            const bool hit = (tmax >= entry) & (entry < t.closest);
            e[k] = hit ? entry : PYR_INF; // misses sort last
            c[k] = hit ? c[k] : INT32_MIN;
        }
    }
    // order (e, c) ascending: network (0,1) (2,3) (0,2) (1,3) (1,2)
    auto order = [&](int a, int b) {
        const bool sw = e[b] < e[a];
        const float ea = sw ? e[b] : e[a], eb = sw ? e[a] : e[b];
        const int ca = sw ? c[b] : c[a], cb = sw ? c[a] : c[b];
        e[a] = ea, e[b] = eb, c[a] = ca, c[b] = cb;
    };
    order(0, 1);
    order(2, 3);
    order(0, 2);
    order(1, 3);
    order(1, 2);
}
// GLOBAL says the scene pointers are known to point into device memory (everything but a scene staged in LDS): the loads
// are then global_load, not flat_load (own_scalar's result is an integer to the compiler, so the address space is stated here).
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const f4v global_f4v;
template <bool GLOBAL>
struct ScenePtr {
    const float4* p;
    DEV float4 operator[](size_t i) const {
        if constexpr (GLOBAL) {
            const f4v v = ((global_f4v*)p)[i];
            return make_float4(v.x, v.y, v.z, v.w);
        } else {
            return p[i];
        }
    }
};
// The seven vectors of a Node128 a visit needs, planes picked by the ray's signs (trav_ray_signs): near x / y / z, far x / y / z,
// children. A uniform base plus a 32-bit byte offset per lane (node * 128 + 0 or 48): global_load with an SGPR base.
struct WidePlanes {
    float4 nx, ny, nz, fx, fy, fz, ch;
};
DEV WidePlanes load_wide_planes(const float4* wide_nodes, const Trav& t) {
    const char* nodes = reinterpret_cast<const char*>(wide_nodes); // the wide tree is never staged in LDS
    auto plane = [&](uint32_t byte_offset) {
        const f4v v = *(global_f4v*)(nodes + byte_offset);
        return make_float4(v.x, v.y, v.z, v.w);
    };
    const uint32_t base = (uint32_t)t.node << 7;
    const uint32_t ax = base + t.nx, ay = base + t.ny, az = base + t.nz;
    return WidePlanes{plane(ax), plane(ay + 16u), plane(az + 32u), plane(ax ^ 48u), plane((ay ^ 48u) + 16u), plane((az ^ 48u) + 32u), plane(base + 96u)};
}
// c[] / e[] of the node a lane stands at. (Round 4 measured a quantized 64-byte node here -- four loads per visit instead of
// seven for +33 vector instructions: C3 -7 %, intersect_kernel +3 %, profiles/r04_c3_ta_tcp_counters.txt -- and deleted it.)
template <bool COUNT>
DEV void wide_children(const float4* wide_nodes, const Trav& t, Counters& cnt, float (&e)[4], int (&c)[4]) {
    const WidePlanes pl = load_wide_planes(wide_nodes, t); // near / far planes picked by the ray's signs (trav_ray_signs)
    wide_node_children<COUNT, true>(pl.nx, pl.ny, pl.nz, pl.fx, pl.fy, pl.fz, pl.ch, t, cnt, e, c);
}
template <bool COUNT, bool GLOBAL = true>
DEV bool trav_step_wide(const SceneView& view, Trav& t, TravStack& stack, Counters& cnt) {
    float e[4];
    int c[4];
    wide_children<COUNT>(view.nodes, t, cnt, e, c);
    if (c[0] == INT32_MIN) { // nothing hit
        if (t.sp == 0) return true;
        t.sp--;
        t.node = stack.pop(t.sp);
        return false;
    }
    if (c[3] != INT32_MIN) stack.push(t.sp++, c[3]);
    if (c[2] != INT32_MIN) stack.push(t.sp++, c[2]);
    if (c[1] != INT32_MIN) stack.push(t.sp++, c[1]);
    t.node = c[0];
    return false;
}

// One primitive of a leaf, its record already loaded (a, b, c = the three vectors of a DevPrim): the tests and the
// bookkeeping of trav_step's leaf part. `first` / `count` are the leaf code's fields. Returns true when the traversal has finished.
// The test and the verdict; true when the primitive blocks a shadow ray.
template <bool COUNT>
DEV bool leaf_prim_test(const float4 a, const float4 b, const float4 c, Trav& t, Counters& cnt) {
    const uint32_t shape = __float_as_uint(a.w);
    float dist = 0.0f, u = 0.0f, v = 0.0f;
    bool ok;
    if ((shape >> 30) == PYR_SHAPE_TRIANGLE) {
        if (COUNT) cnt.triangle_tests++;
        ok = triangle_test(mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), mk(c.x, c.y, c.z), t.o, t.d, dist, u, v);
    } else {
        f3 point;
        if (COUNT) cnt.sphere_tests++;
        ok = sphere_box_guard(mk(a.x, a.y, a.z), b.x, t.o, t.d, t.closest, t.shadow) && sphere_test(mk(a.x, a.y, a.z), b.x, t.o, t.d, dist, point);
    }
    // the verdict as selects (world.rs:290 for an extension ray, tracer.rs:381-389 for a shadow ray), then where to go next
    ok = ok & (dist > DIST_EPSILON);
    // An extension ray's limit is -inf (trav_begin): nothing blocks it. A shadow ray keeps its cut-off (trav_restart) whatever it
    // meets beyond the limit: where spheres are, a blocker's own box may lie BEYOND the lamp's hit (DevScene::shadow_margin), and the
    // reference -- a closest-hit walk from closest = inf -- counts such a blocker (fuzz scene 11941, tools/fuzz_trace.py).
    const bool blocks = ok & (dist * dist < t.limit);
    const bool closer = ok & (dist < t.closest) & !t.shadow;
    t.blocked = t.blocked | blocks;
    t.closest = closer ? dist : t.closest;
    t.shape = closer ? shape : t.shape;
    t.u = closer ? u : t.u;
    t.v = closer ? v : t.v;
    return blocks;
}
template <bool COUNT>
DEV bool leaf_prim_visit(const float4 a, const float4 b, const float4 c, uint32_t first, uint32_t count, Trav& t, TravStack& stack, Counters& cnt) {
    if (leaf_prim_test<COUNT>(a, b, c, t, cnt)) return true;
    if (count > 1) {
        t.node = -1 - (int)(((first + 1) << 3) | (count - 1));
        return false;
    }
    if (t.sp == 0) return true;
    t.sp--;
    t.node = stack.pop(t.sp);
    return false;
}

// An inner-node visit of a lane whose t.node >= 0. Returns true when the traversal has finished.
template <bool COUNT, bool GLOBAL = true>
DEV bool trav_node_step(const SceneView& view, Trav& t, TravStack& stack, Counters& cnt) {
    if (view.wide) return trav_step_wide<COUNT, GLOBAL>(view, t, stack, cnt);
    if (COUNT) cnt.box_tests += 2;
    float e0, e1;
    int c0, c1;
    slab_pair_signed<GLOBAL>(view.nodes, (uint32_t)t.node, t.nx >> 1, t.ny >> 1, t.nz >> 1, t.o, t.inv, e0, e1, c0, c1); // 48 -> 24: the binary node's stride
    const bool h0 = (e0 >= 0.0f) & (e0 < t.closest), h1 = (e1 >= 0.0f) & (e1 < t.closest);
    if (h0 && h1) {
        const bool swap = e1 < e0;
        t.node = swap ? c1 : c0;
        stack.push(t.sp, swap ? c0 : c1);
        t.sp++;
    } else if (h0) {
        t.node = c0;
    } else if (h1) {
        t.node = c1;
    } else {
        if (t.sp == 0) return true;
        t.sp--;
        t.node = stack.pop(t.sp);
    }
    return false;
}
// One primitive of the leaf a lane stands in (t.node < 0). A leaf is walked one primitive per step (the code in t.node shrinks:
// first + 1, count - 1), in leaf order. Looping over the whole leaf here made every wave pay for its fullest leaf (4
// primitives) at each step while most lanes were at inner nodes: 17 % VALU lane occupancy in the traversal kernel.
// Two triangles of a leaf in one step, both Moeller-Trumbore tests in packed f32 (DevPrimPair): the operations of
// triangle_test, component pairs side by side, and the verdicts applied in leaf order -- triangle A, then triangle B against
// the closest distance A left -- so the outcome is the one two single steps give. A ray makes 3.3 primitive steps on C3
// against 4.5 node visits; with pairs it makes ~1.9, the wave's vote goes to the node kind more often and finds more lanes
// there, and a pair costs ~1.3 single tests.
template <bool COUNT>
DEV bool leaf_pair_test(const float4 q0, const float4 q1, const float4 q2, const float4 q3, const float4 q4, uint32_t count, Trav& t, Counters& cnt) {
    const f2v v1x = {q0.x, q0.y}, v1y = {q0.z, q0.w}, v1z = {q1.x, q1.y};
    const f2v e1x = {q2.x, q2.y}, e1y = {q2.z, q2.w}, e1z = {q3.x, q3.y};
    const f2v e2x = {q3.z, q3.w}, e2y = {q4.x, q4.y}, e2z = {q4.z, q4.w};
    const uint32_t shape_a = __float_as_uint(q1.z), shape_b = __float_as_uint(q1.w);
    if (COUNT) cnt.triangle_tests += count >= 2u ? 2u : (count != 0u ? 1u : 0u); // an empty leaf's record holds no triangle
    const f2v dx = {t.d.x, t.d.x}, dy = {t.d.y, t.d.y}, dz = {t.d.z, t.d.z};
    const f2v ox = {t.o.x, t.o.x}, oy = {t.o.y, t.o.y}, oz = {t.o.z, t.o.z};
    // triangle_test, shapes/mod.rs:75-119, for both triangles: p = d x e2, det = e1 . p, t = o - v1, u = (t . p) / det,
    // q = t x e1, v = (d . q) / det, dist = (e2 . q) / det -- the same operations in the same order per component
    const f2v px = dy * e2z - dz * e2y, py = dz * e2x - dx * e2z, pz = dx * e2y - dy * e2x;
    const f2v det = e1x * px + e1y * py + e1z * pz;
    const f2v inv_det = {1.0f / det.x, 1.0f / det.y};
    const f2v tx = ox - v1x, ty = oy - v1y, tz = oz - v1z;
    const f2v u = (tx * px + ty * py + tz * pz) * inv_det;
    const f2v qx = ty * e1z - tz * e1y, qy = tz * e1x - tx * e1z, qz = tx * e1y - ty * e1x;
    const f2v v = (dx * qx + dy * qy + dz * qz) * inv_det;
    const f2v dist = (e2x * qx + e2y * qy + e2z * qz) * inv_det;
    auto passes = [](float det_, float u_, float v_, float dist_) {
        return !(det_ > -DIST_EPSILON && det_ < DIST_EPSILON) & !(u_ < 0.0f || u_ > 1.0f) & !(v_ < 0.0f || u_ + v_ > 1.0f) & (dist_ > DIST_EPSILON);
    };
    const bool ok_a = passes(det.x, u.x, v.x, dist.x), ok_b = passes(det.y, u.y, v.y, dist.y) & (count >= 2u);
    const bool blocks = (ok_a & (dist.x * dist.x < t.limit)) | (ok_b & (dist.y * dist.y < t.limit)); // see leaf_prim_test: no ray kind
    const bool closer_a = ok_a & (dist.x < t.closest);
    const float after_a = closer_a ? dist.x : t.closest;
    const bool closer_b = ok_b & (dist.y < after_a);
    t.blocked = t.blocked | blocks;
    t.closest = closer_b ? dist.y : after_a;
    t.shape = closer_b ? shape_b : (closer_a ? shape_a : t.shape);
    t.u = closer_b ? u.y : (closer_a ? u.x : t.u);
    t.v = closer_b ? v.y : (closer_a ? v.x : t.v);
    return blocks;
}
template <bool COUNT>
DEV bool leaf_pair_visit(const float4 q0, const float4 q1, const float4 q2, const float4 q3, const float4 q4, uint32_t first, uint32_t count, Trav& t,
                         TravStack& stack, Counters& cnt) {
    if (leaf_pair_test<COUNT>(q0, q1, q2, q3, q4, count, t, cnt)) return true;
    if (count > 2u) {
        t.node = -1 - (int)(((first + 1u) << 3) | (count - 2u));
        return false;
    }
    if (t.sp == 0) return true;
    t.sp--;
    t.node = stack.pop(t.sp);
    return false;
}

template <bool COUNT, bool GLOBAL = true>
DEV bool trav_leaf_step(const SceneView& view, Trav& t, TravStack& stack, Counters& cnt) {
    const uint32_t code = (uint32_t)(-1 - t.node);
    const uint32_t first = code >> 3, count = code & 7u;
    if (count != 0 && view.wide && view.pairs != nullptr) {
        const ScenePtr<true> pr{view.pairs + 5 * (size_t)first};
        return leaf_pair_visit<COUNT>(pr[0], pr[1], pr[2], pr[3], pr[4], first, count, t, stack, cnt);
    }
    if (count != 0) {
        const ScenePtr<GLOBAL> pr{view.prims + 3 * (size_t)first};
        return leaf_prim_visit<COUNT>(pr[0], pr[1], pr[2], first, count, t, stack, cnt);
    }
    if (t.sp == 0) return true;
    t.sp--;
    t.node = stack.pop(t.sp);
    return false;
}
// The step of trav_step_voted (majority kind only) for the tree the big scenes walk -- four-child nodes whose leaves index
// triangle pairs -- as straight-line code. What the generic step spends around the tests is control flow: three conditional
// pushes, each an LDS-or-scratch choice (two saved exec masks and their branches per push), the same again around the pop, and
// the exits of the leaf. Here, as long as no lane of the wave is within three levels of the end of its LDS stack (a wave-uniform
// test; otherwise the generic step runs, same results):
//   - the top of the stack is read when the step begins, next to the node's loads, whether or not it will be popped;
//   - the children that are hit are a prefix of the sorted four, so child k goes to level sp + n - k, and a child that is NOT
//     hit is stored too -- at level sp + n, above the new top, where nothing lives;
//   - node, stack pointer and the verdict are selects.
// Same tests, same order of visits, same stack contents below the top as the generic step.
template <bool COUNT>
DEV bool trav_step_lean(const SceneView& view, Trav& t, TravStack& stack, Counters& cnt, bool active);
// One step for the lanes of a wave that have a ray in flight (`active`), with a vote: a step is an inner-node visit or a
// primitive test, two different pieces of code, and a wave whose lanes want both runs both at partial occupancy. Only the
// kind most lanes wait for runs this turn; the minority keeps its place (rays are independent; a later turn serves them).
// Must be called by every lane of the wave. Measured on C3 (intersect Mrays/s | stage-scheduled render Msamples/s): both kinds
// every turn 5774 | 210; minority too when it has >= 16 / 32 lanes 5899 | 225, 5882 | 228; majority only 6087 | 236.
// Wave priority per phase of the stage scheduler (s_setprio, 0-3: a SIMD issues from the ready wave with the highest priority).
// With equal priorities the arbiter interleaves a wave that walks the tree with one that replays tapes instruction by
// instruction and both chains stretch; with the traversal on top, the replay (full width, no dependent fetches, the longest
// phase) at the bottom and SHADE / NEE between them, every phase runs close to its own speed whenever it is ready and the
// lower ones fill its waits: C3 533 -> 579 Msamples/s, C5 464 -> 507 (128 / 256 spp). Any split between traversal and the rest
// gives +6 % -- in EITHER direction --, four distinct levels +8.5 %; the same priority per WAVE instead of per phase gives nothing.
#ifndef PYR_PRIO_E
#define PYR_PRIO_E 0
#endif
#ifndef PYR_PRIO_S
#define PYR_PRIO_S 1
#endif
#ifndef PYR_PRIO_N
#define PYR_PRIO_N 2
#endif
#ifndef PYR_PRIO_T
#define PYR_PRIO_T 3
#endif
#define PYR_PRIO_ANY (PYR_PRIO_E != PYR_PRIO_S || PYR_PRIO_S != PYR_PRIO_N || PYR_PRIO_N != PYR_PRIO_T)
#if PYR_PRIO_ANY
#define PHASE_PRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define PHASE_PRIO(x)
#endif
template <bool COUNT, bool GLOBAL = true>
DEV bool trav_step_voted(const SceneView& view, Trav& t, TravStack& stack, Counters& cnt, bool active) {
    const bool at_node = t.node >= 0;
    const unsigned long long nodes = ballot64(active && at_node), leaves = ballot64(active && !at_node);
    // the choice is wave-uniform, so it is a scalar branch to ONE of the two bodies, not two masked regions
    if (__popcll(nodes) >= __popcll(leaves)) return (active && at_node) ? trav_node_step<COUNT, GLOBAL>(view, t, stack, cnt) : false;
    return (active && !at_node) ? trav_leaf_step<COUNT, GLOBAL>(view, t, stack, cnt) : false;
}

template <bool COUNT>
DEV bool trav_step_lean(const SceneView& view, Trav& t, TravStack& stack, Counters& cnt, bool active) {
    const bool at_node = t.node >= 0;
    // ballots of single comparisons, combined as scalars: the ballot of a conjunction is lowered through a v_cndmask 0 / 1 and a
    // v_cmp_ne (two vector instructions per ballot, three ballots per step)
    const unsigned long long with_ray = ballot64(active), at_nodes = ballot64(at_node);
    const unsigned long long nodes = with_ray & at_nodes, leaves = with_ray & ~at_nodes;
    // the counts as 32-bit scalars the compiler cannot see through: left alone it compares the two 64-bit population counts,
    // for which the scalar unit has no instruction -- two v_mov and a v_cmp_lt_u64 per step
    int n_nodes = __builtin_popcountll(nodes), n_leaves = __builtin_popcountll(leaves);
    asm volatile("" : "+s"(n_nodes), "+s"(n_leaves));
    if ((with_ray & ballot64(t.sp + 3 > stack.lds_entries)) != 0ull) { // somebody's stack is about to leave LDS: the generic step
        if (n_nodes >= n_leaves) return (active && at_node) ? trav_node_step<COUNT, true>(view, t, stack, cnt) : false;
        return (active && !at_node) ? trav_leaf_step<COUNT, true>(view, t, stack, cnt) : false;
    }
    bool done = false;
    const int below = max(t.sp - 1, 0);
    if (n_nodes >= n_leaves) {
        if (active && at_node) {
            const int top = stack.lds[below * BLOCK];
            float e[4];
            int c[4];
            wide_children<COUNT>(view.nodes, t, cnt, e, c);
            const bool none = c[0] == INT32_MIN, hit1 = c[1] != INT32_MIN, hit2 = c[2] != INT32_MIN, hit3 = c[3] != INT32_MIN;
            const int n = (hit1 ? 1 : 0) + (hit2 ? 1 : 0) + (hit3 ? 1 : 0); // children to push: c[1 .. n], far to near
            const int above = t.sp + n;
            stack.lds[(hit3 ? t.sp : above) * BLOCK] = c[3]; // hit3 means n == 3: c[3] is the farthest, at the bottom
            stack.lds[(hit2 ? above - 2 : above) * BLOCK] = c[2];
            stack.lds[(hit1 ? above - 1 : above) * BLOCK] = c[1];
            done = none & (t.sp == 0);
            t.node = none ? top : c[0];
            t.sp = none ? below : above;
        }
    } else if (active && !at_node) {
        const int top = stack.lds[below * BLOCK];
        const uint32_t code = (uint32_t)(-1 - t.node);
        const uint32_t first = code >> 3, count = code & 7u;
        const ScenePtr<true> pr{view.pairs + 5 * (size_t)first};
        const bool blocks = leaf_pair_test<COUNT>(pr[0], pr[1], pr[2], pr[3], pr[4], count, t, cnt);
        const bool more = count > 2u;
        done = blocks | (!more & (t.sp == 0));
        t.node = more ? -1 - (int)(((first + 1u) << 3) | (count - 2u)) : top;
        t.sp = more ? t.sp : below;
    }
    return done;
}

// Per-path state of the resumable integrator and the code of its phases (the stage-scheduled kernel keeps it in registers).
// Spectral tape (TAPE builds: the stage-scheduled kernel on scenes without interpreter programs, i.e. every BASELINE
// config; since round 4 also on scenes WITH interpreter programs whose colour programs have a tape form, see tape_pending). Without the interpreter a colour program is a function of the wavelength alone, and nothing a path decides
// depends on its brightness or reflectance (the reference has no Russian roulette; only the hero wavelength's value enters
// the geometry, through dispersion). So `contribute` (renderer/algorithm.rs:14-100) need not run while the path is walked,
// in phases that execute at a third of the wave's width: each path only records what contribute would be applied to --
//     MUL   program, s   reflectance *= program(wl) * s        a bounce (algorithm.rs:48-63)
//     ADD   program, s   brightness  += program(wl) * s * reflectance   emission, sky, an unblocked light sample (:30-46, :65-90)
//     SCALE s            reflectance *= s                      the BRDF factor (:92-98)
// -- 8 bytes per record, appended to the lane's column of a tape in HBM, and when paths end the wave replays the tapes of all
// its finished lanes with one (path, wavelength) pair per lane, at full width: the same f32 operations in the same order for
// every wavelength, so the film is bit-identical. The S wavelengths stay in LDS; the per-lane brightness / reflectance arrays
// ([2 (S - 1)][256] floats of LDS) are gone, which lets the traversal stack live in LDS down to 16 levels.
// Developer builds that time the parts of this scheme by leaving one out (the FILM IS WRONG under each of them; DESIGN.md 3.2
// quotes the numbers): -DPYR_TAPE_NOSTORE (no records written), -DPYR_TAPE_NOREPLAY (no replay), -DPYR_REPLAY_NOEVAL (record
// by record path: programs evaluate to 1), -DPYR_REPLAY_NOEXPOSE (no film atomics).
constexpr uint32_t kTapeEagerSlots = 8; // LDS rows for the values of the programs that read a spectrum (replay_tapes)
constexpr uint32_t TAPE_MUL = 0u, TAPE_ADD = 1u, TAPE_SCALE = 2u, TAPE_HERO_ONLY = 1u << 29, TAPE_PROGRAM_MASK = (1u << 28) - 1u;
// Records of the eager replay (the scene's spectrum-reading programs fit the LDS value rows): every record is "m = value[slot] * s;
// then reflectance *= m, or brightness += m * reflectance" -- bit 31 says which, bits 8-11 the slot (as an index into the value
// rows: slot * BLOCK). A constant program's value was folded into s when the record was written, and the BRDF factor is a
// plain factor: both name the slot that holds 1.0 (x * 1.0 is x, bit for bit), so the replay has one straight-line form.
constexpr uint32_t TAPE_EAGER_ADD = 1u << 31, TAPE_EAGER_SLOT_SHIFT = 8, TAPE_EAGER_SLOT_MASK = 0xFu << TAPE_EAGER_SLOT_SHIFT;
constexpr uint32_t kTapeOneSlot = kTapeEagerSlots - 1; // the value row that holds 1.0
static_assert(kTapeEagerSlots == kTapeValueRows && kTapeOneSlot == kTapeOneRow, "device_scene.h tape_row() and friends");
// Eager records of a HIT_RGB contribution (device_scene.h TapeForm; scenes with S.rgb_records): bits 12-13 of the word say what the
// record does with m = value[slot] * s -- 1: t = m (first coefficient times the red basis), 2: t = t + m (green, blue),
// 3: m = t * s, then applied like any record (the contribution's factor). 0: an ordinary record.
// A PRODUCT contribution (kernels built with PRODUCT: scenes with such a program) uses the same accumulator: 1 with the wavelength side's
// slot and the first hit factor, 4: t = t * m (m = 1.0 * the next hit factor) for each further one, then 3.
constexpr uint32_t TAPE_RGB_SHIFT = 12, TAPE_RGB_FIRST = 1u << TAPE_RGB_SHIFT, TAPE_RGB_NEXT = 2u << TAPE_RGB_SHIFT, TAPE_RGB_APPLY = 3u << TAPE_RGB_SHIFT,
                   TAPE_RGB_TIMES = 4u << TAPE_RGB_SHIFT;
static_assert(BLOCK == 1u << TAPE_EAGER_SLOT_SHIFT, "an eager record's slot field is an index into rows of BLOCK floats");

// Where record `op` of tape column `column` lives: [op][column], a row of one record index is contiguous (512 B per wave).
// (Round 4 tried [op / 4][column][op % 4] -- a path's four consecutive records in one 32-byte sector, read back as two dwordx4:
// the 8-byte stores do NOT merge in L2 before they are written back, the neighbouring lanes' records no longer share a sector
// either, and the fabric saw 85 GB instead of 68 GB per 32-spp frame: C3 -3.5 %, profiles/r04_write_traffic_split.txt.)
DEV size_t tape_index(uint32_t op, uint32_t lanes, uint32_t column) { return (size_t)op * lanes + column; }
template <bool COUNT, bool INTERP, bool TAPE = false, bool PRODUCT = false> // PRODUCT: the scene has TAPE_FORM_PRODUCT colour programs (device_scene.h)
struct Walker {
    uint32_t stage = ST_NEW;
    uint32_t n_ops = 0; // TAPE: records on this path's tape
    uint32_t tape_column = 0; // TAPE: this path's column of the tape (the lane of the persistent grid, or the pool slot)
    const uint32_t* tape_prepared = nullptr; // TAPE, eager replay: the kernel's LDS table of prepared programs, else nullptr
    DEV void tape_push(const RenderLaunch& L, uint32_t kind, uint32_t program, float s, bool hero_only = false) {
#ifndef PYR_TAPE_NOSTORE
        if (n_ops < L.tape_max_ops)
#else
        if (n_ops > 1000000u)
#endif
        {
            uint32_t word = (kind << 30) | (hero_only ? TAPE_HERO_ONLY : 0u) | (program & TAPE_PROGRAM_MASK);
            if (tape_prepared != nullptr) {
                // eager replay: the record names the LDS slot of the program's value; a constant program is folded into the
                // factor -- c * s is the very product `contribute` forms (program value times probability)
                uint32_t slot = kTapeOneSlot;
                if (kind != TAPE_SCALE) {
                    const uint32_t* e = tape_prepared + 8 * program;
                    if (e[0] == 0u)
                        s = __uint_as_float(e[1]) * s;
                    else
                        slot = e[7];
                }
                word = (kind == TAPE_ADD ? TAPE_EAGER_ADD : 0u) | (hero_only ? TAPE_HERO_ONLY : 0u) | (slot << TAPE_EAGER_SLOT_SHIFT);
            }
            L.tape[tape_index(n_ops, L.tape_lanes, tape_column)] = (unsigned long long)word | ((unsigned long long)__float_as_uint(s) << 32);
        }
#ifndef PYR_TAPE_NOSTORE
        else {
            // more records than tape_ops_bound() allows for: the bound was derived by hand from trace / trace_direct, so a change
            // there that outgrows it must not pass as a slightly wrong film -- the host turns this word into PYR_ERR_DEVICE
            *L.tape_overflow = 1u;
        }
#endif
        n_ops++;
    }
    // One eager record as it stands (HIT_VALUE / HIT_RGB contributions: Walker::tape_pending).
    DEV void tape_push_raw(const RenderLaunch& L, uint32_t word, float s) {
        if (n_ops < L.tape_max_ops)
            L.tape[tape_index(n_ops, L.tape_lanes, tape_column)] = (unsigned long long)word | ((unsigned long long)__float_as_uint(s) << 32);
        else
            *L.tape_overflow = 1u;
        n_ops++;
    }
    uint32_t rgb_slot = 0; // eager replay of a scene with HIT_RGB programs: the value slot of the red basis (green and blue follow)
    uint32_t chunk = 0; // next chunk of this lane's sample sequence (chunk_begin + wave, + total_waves, ...)
    Path p{};
    Trav t{};
    // context of the bounce in flight (between SHADE and the end of its next-event estimation)
    f3 b_position = mk(0, 0, 0), b_normal = mk(0, 0, 0), b_out = mk(0, 0, 0);
    bool b_flip = false; // the face-forwarded normal of trace_direct (tracer.rs:359-363) is -b_normal
    DEV f3 b_nff() const { return b_flip ? -b_normal : b_normal; }
    bool b_has_brdf = false;
    uint32_t nee_lamp = 0, nee_i = 0;
    // 1 / (samples * 2 pi * p_lamp), tracer.rs:365: a function of the launch alone, formed where it is used
    DEV static float nee_probability(const DevScene& S, const RenderLaunch& L) {
        const float lamp_probability = 1.0f / (float)S.num_lamps;
        return 1.0f / ((float)L.light_samples * 2.0f * PI_F * lamp_probability);
    }
    // the light sample whose shadow ray is in flight
    bool ls_pending = false, ls_physical = false;
    uint32_t ls_material = 0, ls_color = 0;
    f3 ls_normal = mk(0, 0, 0);
    float ls_scale = 0.0f;
    float ls_tx = 0.0f, ls_ty = 0.0f; // its texture coordinates (interpreter builds)

    // Interpreter builds (INTERP, no tape): what `contribute` (renderer/algorithm.rs:14-100) is to be applied to by the phase
    // that has just run -- at most one program evaluation (a bounce's colour, an emission / sky / light-sample addition) and
    // the bounce's closing BRDF factor -- is noted here and applied by contribute_pending() right behind the phases: ONE
    // in-line copy of the interpreter instead of eleven out-of-line calls (each of which spilled the walker around itself:
    // 558 VGPR spills, 2.1 KB of scratch per lane), and one memoised run per hit instead of one full run per wavelength.
    // Order is the reference's: the program's contribution first, then the BRDF factor; nothing else touches brightness or
    // reflectance in between (a lane enters SHADE and then NEE in one turn at most, and NEE's first visit adds nothing).
    enum : uint32_t { CONTRIB_NONE = 0, CONTRIB_MUL = 1, CONTRIB_ADD = 2 };
    uint32_t c_kind = CONTRIB_NONE, c_color = 0;
    int c_probability = -1;  // a probability program to evaluate at the hero wavelength: cp = value * c_cp; else cp = c_cp
    float c_cp = 1.0f, c_outer = 1.0f, c_scale = 1.0f;
    bool c_use_outer = false, c_companions = false, c_has_scale = false;
    f3 c_normal = mk(0, 0, 0), c_incident = mk(0, 0, 0);
    float c_tx = 0.0f, c_ty = 0.0f;
    DEV void contribution(uint32_t kind, uint32_t color, int probability, float cp, bool use_outer, float outer, bool companions, f3 normal, f3 incident, float tx,
                          float ty) {
        c_kind = kind, c_color = color, c_probability = probability, c_cp = cp, c_use_outer = use_outer, c_outer = outer, c_companions = companions;
        c_normal = normal, c_incident = incident, c_tx = tx, c_ty = ty;
    }
    // What the phases noted lives for one turn of the stage loop only: applied (or recorded) right behind the phases, then CLEARED --
    // to constants, on every turn -- so that the compiler sees none of these sixteen words alive across the loop's back edge and the
    // traversal phase does not carry them in registers (the compiler cannot tell that they are only read while c_kind says so).
    DEV void contribution_clear() {
        c_kind = CONTRIB_NONE, c_color = 0u, c_probability = -1;
        c_cp = 1.0f, c_outer = 1.0f, c_scale = 1.0f;
        c_use_outer = false, c_companions = false, c_has_scale = false;
        c_normal = mk(0, 0, 0), c_incident = mk(0, 0, 0);
        c_tx = 0.0f, c_ty = 0.0f;
    }
    DEV void contribute_pending(const DevScene& S, const RenderLaunch& L, Spectral& spec) {
        if constexpr (INTERP && !TAPE) {
            const uint32_t n_add = L.spectrum_samples - 1;
            if (c_kind != CONTRIB_NONE) {
                VmInput in{p.wl, c_normal, c_incident, c_tx, c_ty};
                Vm vm;
                float cp = c_cp, factor = 0.0f;
                // job 0: the probability program (hero wavelength only: ProbabilityInput), job 1: the colour program for the hero
                // and, memoised, for the companions
                for (uint32_t job = c_probability >= 0 ? 0u : 1u; job < 2u; ++job) {
                    const uint32_t id = job == 0u ? (uint32_t)c_probability : c_color;
                    const DevProgram prog = S.programs[id];
                    const bool interpreted = prog.kind != PYR_PROGRAM_CONSTANT && prog.fast == FAST_NONE;
                    const Prepared q = prepare_program<false>(S, id);
                    const uint32_t passes = (job == 1u && c_companions) ? 1u + n_add : 1u;
                    if (job == 1u) factor = c_use_outer ? c_outer * cp : cp;
                    for (uint32_t pass = 0; pass < passes; ++pass) {
                        in.wavelength = pass == 0u ? p.wl : spec.wl(pass - 1u);
                        float v;
                        if (interpreted) {
                            for (uint32_t k = 0; k < prog.num_instrs; ++k) {
                                const PyrInstr& ins = S.instrs[prog.first_instr + k];
                                if (pass == 0u || (ins.deps & PYR_DEP_WAVELENGTH) != 0u) vm.step(S, ins, in);
                            }
                            v = vm.number(prog);
                        } else {
                            v = eval_prepared<false>(S, q, in);
                        }
                        if (job == 0u) {
                            cp = v * c_cp;
                        } else if (pass == 0u) {
                            if (c_kind == CONTRIB_MUL) p.refl *= v * factor;
                            else p.bright += v * factor * p.refl;
                        } else {
                            if (c_kind == CONTRIB_MUL) spec.refl(pass - 1u) *= v * factor;
                            else spec.bright(pass - 1u) += v * factor * spec.refl(pass - 1u);
                        }
                    }
                }
                c_kind = CONTRIB_NONE;
            }
            if (c_has_scale) {
                p.refl *= c_scale;
                if (p.use_additional)
                    for (uint32_t k = 0; k < n_add; ++k) spec.refl(k) *= c_scale;
                c_has_scale = false;
            }
        }
    }

    // Interpreter builds WITH a tape (scenes whose colour programs all have a tape form, device_scene.h TapeForm; round 4): what the
    // phase noted is not applied wavelength by wavelength here, at the width of a SHADE / NEE phase -- that was 60 % of such a
    // scene's wave cycles (profiles/r04_phase_profile_interpreter.txt) -- it is RECORDED: the interpreter runs what depends on the
    // hit alone, once (the probability program at the hero wavelength, ProbabilityInput; a HIT_VALUE colour program whole; the rgb
    // register of a HIT_RGB one), and the replay (replay_tapes) does the per-wavelength part for one (path, wavelength) pair per
    // lane. The products are the ones contribute forms: value * (outer * (probability * compensation)), then the BRDF factor.
    DEV void tape_pending(const DevScene& S, const RenderLaunch& L) {
        if constexpr (INTERP && TAPE) {
            if (c_kind != CONTRIB_NONE) {
                VmInput in{p.wl, c_normal, c_incident, c_tx, c_ty};
                Vm vm;
                float cp = c_cp, hit_value = 1.0f;
                const DevProgram colour = S.programs[c_color];
                // (PRODUCT is a build of its own: the three conditions it adds here, compiled into every interpreter kernel, cost the reference's
                // example scenes 5-11 % -- these kernels sit on the edge of their register budget)
                const bool product = PRODUCT && colour.tape_form == TAPE_FORM_PRODUCT; // its hit side is a program of its own (api.cpp split_product)
                const bool run_colour = colour.tape_form == TAPE_FORM_HIT_VALUE || colour.tape_form == TAPE_FORM_HIT_RGB || product; // the interpreter runs its hit part; everything else is looked up by the replay
                // job 0: the probability program in full (it is evaluated for the hero wavelength only); job 1: the colour program's
                // instructions that do not depend on the wavelength -- all of a HIT_VALUE program, all but the closing one of a HIT_RGB
                for (uint32_t job = c_probability >= 0 ? 0u : 1u; job < 2u; ++job) {
                    if (job == 1u && !run_colour) break;
                    uint32_t id = job == 0u ? (uint32_t)c_probability : c_color;
                    if constexpr (PRODUCT) id = (job == 1u && product) ? (colour.tape_rgb_reg & 255u) : id; // DevProgram::tape_rgb_reg of a PRODUCT program
                    const DevProgram prog = S.programs[id];
                    const bool interpreted = prog.kind != PYR_PROGRAM_CONSTANT && prog.fast == FAST_NONE;
                    float v;
                    if (interpreted) {
                        const uint32_t count = (job == 1u && prog.tape_form == TAPE_FORM_HIT_RGB) ? prog.num_instrs - 1u : prog.num_instrs;
                        for (uint32_t k = 0; k < count; ++k) vm.step(S, S.instrs[prog.first_instr + k], in); // in line: an out-of-line copy shared with the normal map (tried: scratch 400 -> 1000 B, textures 759 -> 526)
                        v = vm.number(prog);
                    } else {
                        const Prepared q = prepare_program<false>(S, id);
                        v = eval_prepared<false>(S, q, in);
                    }
                    if (job == 0u)
                        cp = v * c_cp;
                    else
                        hit_value = v;
                }
                const float factor = c_use_outer ? c_outer * cp : cp;
                const uint32_t flags = (c_kind == CONTRIB_ADD ? TAPE_EAGER_ADD : 0u) | (c_companions ? 0u : TAPE_HERO_ONLY);
                if (!run_colour) {
                    tape_push(L, c_kind == CONTRIB_ADD ? TAPE_ADD : TAPE_MUL, c_color, factor, !c_companions);
                } else if (colour.tape_form == TAPE_FORM_HIT_VALUE) {
                    tape_push_raw(L, flags | (kTapeOneSlot << TAPE_EAGER_SLOT_SHIFT), hit_value * factor); // value * factor is the product contribute forms
                } else if (product) { // t = value[slot of the wavelength side] * h1, t = t * h2 ...: the program's products in its order; then t * factor
                    const uint32_t packed = colour.tape_rgb_reg;
                    const uint32_t slot = tape_prepared != nullptr ? tape_prepared[8 * ((packed >> 8) & 255u) + 7] : kTapeOneSlot; // (a hit-tape scene always replays eagerly: api.cpp)
                    const uint32_t hero = c_companions ? 0u : TAPE_HERO_ONLY, factors = (packed >> 16) & 15u;
                    tape_push_raw(L, hero | TAPE_RGB_FIRST | (slot << TAPE_EAGER_SLOT_SHIFT), vm.num[(packed >> 20) & 15u]);
                    for (uint32_t k = 1; k < factors; ++k)
                        tape_push_raw(L, hero | TAPE_RGB_TIMES | (kTapeOneSlot << TAPE_EAGER_SLOT_SHIFT), vm.num[(packed >> (20u + 4u * k)) & 15u]);
                    tape_push_raw(L, flags | TAPE_RGB_APPLY | (kTapeOneSlot << TAPE_EAGER_SLOT_SHIFT), factor);
                } else { // HIT_RGB: c0 * basis_r + c1 * basis_g + c2 * basis_b (execution_context.rs:140-152), then times the factor
                    const float* c = vm.rgb[colour.tape_rgb_reg & (PYR_MAX_VECTOR_REGISTERS - 1)];
                    const uint32_t hero = c_companions ? 0u : TAPE_HERO_ONLY;
                    tape_push_raw(L, hero | TAPE_RGB_FIRST | ((rgb_slot + 0u) << TAPE_EAGER_SLOT_SHIFT), c[0]);
                    tape_push_raw(L, hero | TAPE_RGB_NEXT | ((rgb_slot + 1u) << TAPE_EAGER_SLOT_SHIFT), c[1]);
                    tape_push_raw(L, hero | TAPE_RGB_NEXT | ((rgb_slot + 2u) << TAPE_EAGER_SLOT_SHIFT), c[2]);
                    tape_push_raw(L, flags | TAPE_RGB_APPLY | (kTapeOneSlot << TAPE_EAGER_SLOT_SHIFT), factor);
                }
                c_kind = CONTRIB_NONE;
            }
            if (c_has_scale) {
                tape_push(L, TAPE_SCALE, 0u, c_scale);
                c_has_scale = false;
            }
        }
    }

    // tracer.rs:288-301 tail of a bounce + loop head :221: reflectance *= brdf, next ray, bounce count
    DEV void finish_bounce(const DevScene& S, const RenderLaunch& L, Spectral& spec, Counters& cnt) {
        const uint32_t n_add = L.spectrum_samples - 1;
        if (b_has_brdf) {
            const float brdf = 2.0f * fabsf(dot(b_out, b_normal));
            if constexpr (TAPE && !INTERP) {
                tape_push(L, TAPE_SCALE, 0u, brdf);
            } else if constexpr (INTERP) {
                c_scale = brdf, c_has_scale = true; // behind this turn's contribution (contribute_pending)
            } else {
                p.refl *= brdf;
                if (p.use_additional) {
                    for (uint32_t k = 0; k < n_add; ++k) spec.refl(k) *= brdf;
                }
            }
        }
        p.o = b_position;
        p.d = b_out;
        p.bounce++;
        if (p.bounce >= L.bounces) {
            stage = ST_EXPOSE;
        } else {
            if (COUNT) cnt.extension_rays++;
            stage = trav_begin<COUNT>(S, t, p.o, p.d, false, 0.0f, cnt) ? ST_SHADE : ST_TRAV;
        }
    }

    // EXPOSE / NEW: finish the path (simple.rs:133-139) and start the next sample of this lane's sequence (simple.rs:78-107)
    DEV void expose_and_restart(const DevScene& S, const RenderLaunch& L, Spectral& spec, Counters& cnt, uint32_t lane, uint32_t total_waves) {
        if (stage == ST_EXPOSE) {
            if constexpr (!TAPE) finish_path<COUNT>(L, p, spec, cnt); // TAPE: the wave has replayed this lane's tape (replay_tapes)
            stage = ST_NEW;
        }
        if (stage == ST_NEW) {
            stage = ST_DONE;
            while (chunk < L.chunk_end) {
                uint32_t tile;
                uint64_t iteration;
                TileArea area;
                const bool ok = locate_chunk(L, chunk, lane, tile, iteration, area);
                chunk += total_waves;
                if (ok) {
                    start_sample<!TAPE>(L, tile, iteration, area, p, spec);
                    n_ops = 0;
                    if (COUNT) cnt.samples++;
                    if (L.bounces == 0) {
                        stage = ST_EXPOSE;
                    } else {
                        if (COUNT) cnt.extension_rays++;
                        stage = trav_begin<COUNT>(S, t, p.o, p.d, false, 0.0f, cnt) ? ST_SHADE : ST_TRAV;
                    }
                    break;
                }
            }
        }
    }

    // SHADE: the hit/miss handling of tracer::trace (tracer.rs:222-341) up to the start of next-event estimation
    DEV void shade(const DevScene& S, const RenderLaunch& L, Spectral& spec, Counters& cnt) {
        if (stage != ST_SHADE) return;
        const uint32_t n_add = L.spectrum_samples - 1;
        const f3 ray_o = t.o, ray_d = t.d;
        if (t.shape == PYR_HIT_NONE) {
            uint32_t color = S.sky_program;
            if (p.sample_light) {
                for (uint32_t i = 0; i < S.num_lamps; ++i) {
                    const DevLamp& l = S.lamps[i];
                    if (l.kind == PYR_LAMP_DIRECTIONAL && dot(ld3(l.v), ray_d) >= l.width) {
                        color = l.color_program;
                        break;
                    }
                }
            }
            if constexpr (TAPE && !INTERP) {
                tape_push(L, TAPE_ADD, color, 1.0f);
            } else if constexpr (INTERP) {
                contribution(CONTRIB_ADD, color, -1, 1.0f, false, 1.0f, p.use_additional, -ray_d, ray_d, 0.0f, 0.0f);
            } else {
                const Prepared q_prog = prepare_program<INTERP>(S, color);
                VmInput in{p.wl, -ray_d, ray_d};
                p.bright += eval_prepared<INTERP>(S, q_prog, in) * 1.0f * p.refl;
                if (p.use_additional)
                    for (uint32_t k = 0; k < n_add; ++k) {
                        in.wavelength = spec.wl(k);
                        spec.bright(k) += eval_prepared<INTERP>(S, q_prog, in) * 1.0f * spec.refl(k);
                    }
            }
            stage = ST_EXPOSE;
            return;
        }
        if (COUNT) cnt.shaded_hits++;
        Hit hit{t.closest, t.shape, t.u, t.v};
        f3 position, normal;
        uint32_t material_id;
        float tx = 0.0f, ty = 0.0f;
        if constexpr (INTERP)
            surface_textured(S, hit, ray_o, ray_d, position, normal, material_id, tx, ty);
        else
            surface_at(S, hit, ray_o, ray_d, position, normal, material_id);
        const PyrMaterial material = S.materials[material_id];
        const uint32_t pick = rng_choose(p.rng, material.num_components);
        const PyrComponent comp = S.components[material.first_component + pick];
        float component_probability = comp.selection_compensation;
        bool normal_dispersed = false;
        int deferred_probability = -1; // interpreter builds evaluate it with the colour program (contribute_pending)
        if (comp.probability_program >= 0) {
            if constexpr (INTERP) {
                deferred_probability = comp.probability_program;
            } else {
                VmInput pin{p.wl, normal, ray_d, tx, ty};
                component_probability = run_program<INTERP>(S, (uint32_t)comp.probability_program, pin) * comp.selection_compensation;
            }
            normal_dispersed = S.programs[comp.probability_program].reads_wavelength != 0;
        }
        if (comp.bsdf == PYR_BSDF_EMISSIVE) {
            if (p.sample_light) {
                p.use_additional = !normal_dispersed && p.use_additional;
                if constexpr (TAPE && !INTERP) {
                    tape_push(L, TAPE_ADD, comp.color_program, component_probability);
                } else if constexpr (INTERP) {
                    contribution(CONTRIB_ADD, comp.color_program, deferred_probability, comp.selection_compensation, false, 1.0f, p.use_additional, normal, ray_d, tx, ty);
                } else {
                    const Prepared q_prog = prepare_program<INTERP>(S, comp.color_program);
                    VmInput in{p.wl, normal, ray_d, tx, ty};
                    p.bright += eval_prepared<INTERP>(S, q_prog, in) * component_probability * p.refl;
                    if (p.use_additional)
                        for (uint32_t k = 0; k < n_add; ++k) {
                            in.wavelength = spec.wl(k);
                            spec.bright(k) += eval_prepared<INTERP>(S, q_prog, in) * component_probability * spec.refl(k);
                        }
                }
            }
            stage = ST_EXPOSE;
            return;
        }
        f3 out_direction;
        float scatter_probability = 1.0f;
        bool dispersed = false, has_brdf = false;
        if (comp.bsdf == PYR_BSDF_DIFFUSE) {
            f3 n = dot(ray_d, normal) < 0.0f ? normal : -normal;
            out_direction = sample_hemisphere(p.rng, n);
            has_brdf = true;
        } else if (comp.bsdf == PYR_BSDF_MIRROR) {
            f3 n = dot(ray_d, normal) < 0.0f ? normal : -normal;
            float perp = dot(ray_d, n) * 2.0f;
            out_direction = ray_d - n * perp;
        } else {
            dispersed = comp.dispersion != 0.0f || comp.env_dispersion != 0.0f;
            float ior = comp.ior, env_ior = comp.env_ior;
            if (dispersed) {
                float wl = p.wl * 0.001f;
                ior = comp.ior + comp.dispersion / (wl * wl);
                env_ior = comp.env_ior + comp.env_dispersion / (wl * wl);
            }
            refract(ior, env_ior, ray_d, normal, p.rng, out_direction, scatter_probability);
        }
        const float bounce_probability = scatter_probability * component_probability;
        p.use_additional = !(dispersed || normal_dispersed) && p.use_additional;
        if constexpr (TAPE && !INTERP) {
            tape_push(L, TAPE_MUL, comp.color_program, bounce_probability);
        } else if constexpr (INTERP) {
            // bounce_probability = scatter_probability * (probability program's value * compensation), formed where the value is
            contribution(CONTRIB_MUL, comp.color_program, deferred_probability, comp.selection_compensation, true, scatter_probability, p.use_additional, normal, ray_d, tx,
                         ty);
        } else {
            const Prepared q_prog = prepare_program<INTERP>(S, comp.color_program);
            VmInput in{p.wl, normal, ray_d, tx, ty};
            p.refl *= eval_prepared<INTERP>(S, q_prog, in) * bounce_probability;
            if (p.use_additional)
                for (uint32_t k = 0; k < n_add; ++k) {
                    in.wavelength = spec.wl(k);
                    spec.refl(k) *= eval_prepared<INTERP>(S, q_prog, in) * bounce_probability;
                }
        }
        b_position = position;
        b_normal = normal;
        b_out = out_direction;
        b_has_brdf = has_brdf;
        bool nee = false;
        if (p.events < 2) { // tracer.rs:257-280
            p.sample_light = !has_brdf || L.light_samples == 0;
            if (has_brdf) {
                p.events += 1;
                if (S.num_lamps > 0) {
                    nee_lamp = rng_range_usize(p.rng, S.num_lamps); // pick_lamp, world.rs:301-305
                    b_flip = !(dot(ray_d, normal) < 0.0f);
                    nee_i = 0;
                    ls_pending = false;
                    nee = true;
                }
            }
        } else {
            p.sample_light = true;
        }
        if (nee)
            stage = ST_NEE;
        else
            finish_bounce(S, L, spec, cnt);
    }

    // NEE: trace_direct (tracer.rs:347-442), one light sample per visit: account for the shadow ray that came back, then
    // draw the next sample that needs one
    DEV void next_event(const DevScene& S, const RenderLaunch& L, Spectral& spec, Counters& cnt) {
        if (stage != ST_NEE) return;
        const uint32_t n_add = L.spectrum_samples - 1;
        if (ls_pending) {
            ls_pending = false;
            if (!t.blocked) {
                uint32_t l_color = ls_color;
                float material_probability = 1.0f;
                bool l_dispersed = false;
                f3 target_normal = -t.d;
                int deferred_probability = -1; // interpreter builds evaluate it with the colour program (contribute_pending)
                if (ls_physical) {
                    const PyrMaterial lm = S.materials[ls_material];
                    const uint32_t e_pick = rng_choose(p.rng, lm.num_emissive);
                    const PyrComponent ec = S.components[lm.first_emissive + e_pick];
                    material_probability = ec.selection_compensation;
                    if (ec.probability_program >= 0) {
                        if constexpr (INTERP) {
                            deferred_probability = ec.probability_program;
                        } else {
                            VmInput pin{p.wl, ls_normal, t.d, ls_tx, ls_ty};
                            material_probability = run_program<INTERP>(S, (uint32_t)ec.probability_program, pin) * ec.selection_compensation;
                        }
                        l_dispersed = S.programs[ec.probability_program].reads_wavelength != 0;
                    }
                    l_color = ec.color_program;
                    target_normal = ls_normal;
                }
                const float l_probability = ls_scale * material_probability;
                if constexpr (TAPE && !INTERP) {
                    tape_push(L, TAPE_ADD, l_color, l_probability, l_dispersed);
                } else if constexpr (INTERP) {
                    // l_probability = ls_scale * (probability program's value * compensation); the material's inputs are the light's
                    contribution(CONTRIB_ADD, l_color, deferred_probability, material_probability, true, ls_scale, p.use_additional && !l_dispersed, target_normal, t.d,
                                 ls_physical ? ls_tx : 0.0f, ls_physical ? ls_ty : 0.0f);
                } else {
                    const Prepared q_prog = prepare_program<INTERP>(S, l_color);
                    VmInput in{p.wl, target_normal, t.d, ls_physical ? ls_tx : 0.0f, ls_physical ? ls_ty : 0.0f};
                    p.bright += eval_prepared<INTERP>(S, q_prog, in) * l_probability * p.refl;
                    if (p.use_additional && !l_dispersed)
                        for (uint32_t k = 0; k < n_add; ++k) {
                            in.wavelength = spec.wl(k);
                            spec.bright(k) += eval_prepared<INTERP>(S, q_prog, in) * l_probability * spec.refl(k);
                        }
                }
            }
        }
        const DevLamp& lamp = S.lamps[nee_lamp];
        while (nee_i < L.light_samples) {
            const LampSample ls = lamp_sample<INTERP>(lamp, p.rng, b_position);
            nee_i++;
            const float cos_out = fmaxf(dot(b_nff(), ls.direction), 0.0f);
            if (!(cos_out > 0.0f)) continue;
            if (COUNT) cnt.shadow_rays++;
            const float limit = ls.sq_distance >= 0.0f ? ls.sq_distance - DIST_EPSILON : PYR_INF;
            ls_scale = ls.weight * nee_probability(S, L) * (2.0f * fabsf(dot(ls.direction, b_nff())));
            ls_physical = ls.physical;
            ls_material = ls.material;
            ls_color = ls.color;
            ls_normal = ls.normal;
            ls_tx = ls.tx, ls_ty = ls.ty;
            ls_pending = true;
            // a shadow ray decided by a plane alone comes straight back to this phase
            stage = trav_begin<COUNT>(S, t, b_position, ls.direction, true, limit, cnt) ? ST_NEE : ST_TRAV;
            break;
        }
        if (stage == ST_NEE && !ls_pending) finish_bounce(S, L, spec, cnt);
    }
};

// A LAMBDA program (device_scene.h TapeForm) at one wavelength: the number-only subset of Vm::step -- the same expressions in the
// same order (program/execution_context.rs:81-283) -- on a register file of its own. Runs once per replay item and slot, in uniform
// control flow (every lane of the wave interprets the same program).
DEV float lambda_eval(const DevScene& S, const DevProgram& p, float wavelength) {
    float num[PYR_MAX_NUMBER_REGISTERS];
    auto value = [&](const PyrOperand& o) -> float {
        if (o.kind == PYR_OPERAND_CONSTANT) return __uint_as_float(o.bits);
        if (o.kind == PYR_OPERAND_INPUT) return wavelength;
        return num[o.bits & (PYR_MAX_NUMBER_REGISTERS - 1)];
    };
    for (uint32_t k = 0; k < p.num_instrs; ++k) {
        const PyrInstr& ins = S.instrs[p.first_instr + k];
        float r = 0.0f;
        switch (ins.op) {
        case PYR_OP_NUMBER: r = __uint_as_float(ins.x.bits); break;
        case PYR_OP_SPECTRUM: r = spectrum_get(S, ins.a, value(ins.x)); break;
        case PYR_OP_BLACKBODY: {
            const float wl = value(ins.x), temp = value(ins.y);
            r = blackbody(wl, temp);
            break;
        }
        case PYR_OP_MIX: {
            const float amount = fmaxf(fminf(value(ins.x), 1.0f), 0.0f);
            const float l = num[ins.a & (PYR_MAX_NUMBER_REGISTERS - 1)], rr = num[ins.b & (PYR_MAX_NUMBER_REGISTERS - 1)];
            r = l * (1.0f - amount) + rr * amount;
            break;
        }
        case PYR_OP_BINARY: {
            const float l = num[ins.a & (PYR_MAX_NUMBER_REGISTERS - 1)], rr = num[ins.b & (PYR_MAX_NUMBER_REGISTERS - 1)];
            r = ins.operator_ == PYR_BIN_ADD ? l + rr : (ins.operator_ == PYR_BIN_SUB ? l - rr : (ins.operator_ == PYR_BIN_MUL ? l * rr : l / rr));
            break;
        }
        default: { // PYR_OP_CLAMP (api.cpp admits no other opcode into a LAMBDA program)
            const float v = value(ins.x), mn = value(ins.y), mx = value(ins.z);
            r = fmaxf(fminf(v, mx), mn);
            break;
        }
        }
        num[ins.output & (PYR_MAX_NUMBER_REGISTERS - 1)] = r;
    }
    return num[p.output_reg & (PYR_MAX_NUMBER_REGISTERS - 1)];
}

// Replays the tapes of the wave's finished lanes (`exposing`) and exposes the film: contribute + Film::expose for every
// (path, wavelength) pair, one pair per lane. Item i of the wave is wavelength i % S of the (i / S)-th finished lane; index
// S - 1 stands for the hero (kept in the lane's registers), 0 .. S - 2 for the companions in LDS. A path that dispersed
// exposes its hero only (simple.rs:133-139); a light sample whose material reads the wavelength is added for the hero only
// (algorithm.rs:78). Consecutive records of one program (the light samples of one estimation) share one look-up, as in the
// synchronous walk. Must be called by every lane of the wave.
template <bool COUNT, bool RGB = false, bool TIMES = false> // RGB: the scene may hold HIT_RGB records (interpreter builds that record a tape); TIMES: and PRODUCT ones
DEV void replay_tapes(const DevScene& S, const RenderLaunch& L, bool exposing, uint32_t n_ops, uint32_t tape_column, const Path& p, const float* wl_rows,
                      uint32_t wl_column, uint32_t* wave_list, const uint32_t* prepared_lds, float* spectral_values, uint32_t n_spectral, bool eager, Counters& cnt) {
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long mask = ballot64(exposing);
    const uint32_t n = (uint32_t)__popcll(mask);
    if (n == 0) return;
#ifdef PYR_TAPE_NOREPLAY
    if (n_ops < 1000000u) return;
#endif
    // The finished lanes are listed with the paths that keep their companions first, the dispersed ones behind them: a
    // dispersed path exposes its hero wavelength only (simple.rs:133-139), so it is ONE item, not S of which S - 1 idle --
    // on C5 a third of the paths disperse and a turn's items drop from 10 n to ~7 n, often a whole pass of 64 less.
    // Within each group the lanes are listed by tape length, long tapes first (three classes: above 2/3 of the longest, above
    // 1/3, the rest): a pass of 64 items walks as many rows as ITS longest tape has, and a path that left the scene after two
    // bounces has a fifth of the records of one that made all eight -- listed in lane order every pass held a long one.
    uint32_t max_ops = exposing ? (n_ops < L.tape_max_ops ? n_ops : L.tape_max_ops) : 0u;
    const uint32_t my_ops = max_ops;
    for (int off = 32; off > 0; off >>= 1) max_ops = max(max_ops, (uint32_t)__shfl_xor((int)max_ops, off));
    const bool keeps = exposing && p.use_additional, lost = exposing && !p.use_additional;
    const bool longest = 3u * my_ops > 2u * max_ops, shortest = 3u * my_ops <= max_ops;
    const unsigned long long below = (1ull << lane) - 1ull;
    const unsigned long long k0 = ballot64(keeps && longest), k1 = ballot64(keeps && !longest && !shortest), k2 = ballot64(keeps && shortest);
    const unsigned long long l0 = ballot64(lost && longest), l1 = ballot64(lost && !longest && !shortest), l2 = ballot64(lost && shortest);
    const uint32_t n_full = (uint32_t)__builtin_popcountll(k0 | k1 | k2);
    if (exposing) {
        const unsigned long long mine = keeps ? (longest ? k0 : (shortest ? k2 : k1)) : (longest ? l0 : (shortest ? l2 : l1));
        uint32_t before = keeps ? 0u : n_full;
        before += (uint32_t)__builtin_popcountll(keeps ? ((longest ? 0ull : k0) | (shortest ? k1 : 0ull)) : ((longest ? 0ull : l0) | (shortest ? l1 : 0ull)));
        wave_list[before + (uint32_t)__builtin_popcountll(mine & below)] = lane;
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t SS = L.spectrum_samples, full_items = n_full * SS, items = full_items + (n - n_full);
    // The wave reads the tape row by row: a row (one record index of all 64 lanes) is 512 contiguous bytes, so every finished
    // lane loads its own column's record -- one coalesced load per row, eight rows in flight -- and an item takes the record of
    // the lane it replays with a cross-lane read. (Reading record after record of one column from the item's lane was a chain
    // of dependent HBM round trips: it took a quarter of the render.) Lanes that are not being replayed load nothing.
    // Rows past the end of a tape read as the identity record of the eager form -- "reflectance *= value[one] * 1.0", and
    // x * 1.0 is x bit for bit -- so the straight-line replay needs no "is this row still on my tape" test per record.
    const unsigned long long past_the_end =
        eager ? ((unsigned long long)(kTapeOneSlot << TAPE_EAGER_SLOT_SHIFT) | ((unsigned long long)__float_as_uint(1.0f) << 32)) : 0ull;
#ifndef PYR_REPLAY_ROWS
#define PYR_REPLAY_ROWS 8
#endif
    constexpr uint32_t ROWS = PYR_REPLAY_ROWS;
    for (uint32_t base = 0; base < items; base += 64) {
        const uint32_t i = base + lane;
        const bool active = i < items;
        const bool full = i < full_items; // an item of a path that kept its companions; else the hero of a dispersed path
        const uint32_t rank = !active ? 0u : (full ? i / SS : n_full + (i - full_items)), k = !active ? 0u : (full ? i - rank * SS : SS - 1u);
        const uint32_t src = wave_list[rank];
        const uint32_t ops = (uint32_t)__shfl((int)n_ops, (int)src);
        // the rows this pass needs: the longest tape among ITS items
        uint32_t pass_ops = active ? (ops < L.tape_max_ops ? ops : L.tape_max_ops) : 0u;
        for (int off = 32; off > 0; off >>= 1) pass_ops = max(pass_ops, (uint32_t)__shfl_xor((int)pass_ops, off));
        const uint32_t pixel = (uint32_t)__shfl((int)p.pixel, (int)src);
        const float hero_wl = __shfl(p.wl, (int)src);
        const bool hero = k == SS - 1;
        const bool run = active;
        // the companions' wavelengths live in the LDS column of the path's home (the lane itself in render_kernel_sm; wherever the
        // path's sample sequence belongs in render_kernel_px, whose paths move between lanes)
        const uint32_t src_column = (uint32_t)__shfl((int)wl_column, (int)src);
        const float wl = hero ? hero_wl : wl_rows[k * BLOCK + src_column];
        float refl = 1.0f, bright = 0.0f, value = 0.0f;
        uint32_t value_of = 0xFFFFFFFFu;
        // A scene has few programs that read a spectrum (C3: three wall colours and the lamp). When they fit the LDS rows
        // reserved for them, each is looked up once per item, here, in uniform control flow -- the records below then only pick
        // the value (and apply the program's constant factor, the same multiplication run_program makes).
        [[maybe_unused]] float rgb_sum = 0.0f; // HIT_RGB contributions: c0 * basis_r + c1 * basis_g + c2 * basis_b in the making
        if (eager) {
            // Spectrum::get (project/spectra.rs:32-55) of every slot at this item's wavelength, the operations of spectrum_eval.
            // Array spectra over the same grid (min, max, count: C3's three wall colours) share the index arithmetic -- one
            // division instead of three; the look-ups themselves are straight-line (the clamped ends are selects).
            const uint32_t* slot_program = prepared_lds + 8 * L.tape_programs_lds;
            uint32_t grid_min = 0, grid_max = 0, grid_count = 0; // the grid i0 / mix / below / above belong to (count 0: none yet)
            uint32_t i0 = 0, i1 = 0;
            float mix = 0.0f;
            bool below = false, above = false;
            for (uint32_t slot = 0; slot < n_spectral; ++slot) {
                const uint32_t* e = prepared_lds + 8 * slot_program[slot];
                const uint32_t mode = e[0], format = e[2], count = e[6];
                const float c = __uint_as_float(e[1]);
                const float* data = S.spectrum_data + e[5];
                float v;
#ifdef PYR_REPLAY_NOEAGER_EVAL // timing ablation (the film is wrong): what the per-item look-ups of the spectrum-reading programs cost
                v = wl * 1.0e-3f;
#else
                if (RGB && mode == 0xFFu) { // (uniform) a LAMBDA program: interpreted once per item (hit-tape scenes only)
                    v = lambda_eval(S, S.programs[slot_program[slot]], wl);
                } else if (format == PYR_SPECTRUM_ARRAY && count != 0u) {
                    if (e[3] != grid_min || e[4] != grid_max || count != grid_count) { // (uniform) another grid than the previous slot's
                        grid_min = e[3], grid_max = e[4], grid_count = count;
                        const float lo = __uint_as_float(grid_min), hi = __uint_as_float(grid_max);
                        below = wl <= lo, above = wl >= hi;
                        const float normalized = (wl - lo) / (hi - lo);
                        const float float_index = normalized * ((float)count - 1.0f);
                        const float min_float_index = truncf(float_index);
                        i0 = (below | above) ? 0u : (uint32_t)min_float_index;
                        i1 = (below | above) ? 0u : i0 + 1u; // a clamped wavelength reads nothing past a one-point spectrum
                        mix = float_index - min_float_index;
                    }
                    const float inside = data[i0] * (1.0f - mix) + data[i1] * mix;
                    v = below ? data[0] : (above ? data[count - 1] : inside);
                } else {
                    PyrSpectrum sp;
                    sp.format = format, sp.min = __uint_as_float(e[3]), sp.max = __uint_as_float(e[4]), sp.offset = e[5], sp.count = count;
                    v = spectrum_eval(sp, data, wl);
                }
#endif
                spectral_values[tape_row(slot) * BLOCK] = (mode == FAST_SPECTRUM || mode == 0xFFu) ? v : v * c; // FAST_SPECTRUM_MUL and FAST_MUL_SPECTRUM: v * c is c * v
            }
            if (RGB && S.rgb_records != 0) { // (uniform) the RGB basis at this item's wavelength: RgbSpectrumValue's look-up, execution_context.rs:140-152
                float resp[3] = {0.0f, 0.0f, 0.0f};
                const uint32_t count = S.rgb_count;
                if (count > 0) {
                    const float* d = S.rgb_basis;
                    if (wl <= S.rgb_min) {
                        for (int j = 0; j < 3; ++j) resp[j] = d[j];
                    } else if (wl >= S.rgb_max) {
                        for (int j = 0; j < 3; ++j) resp[j] = d[3 * (count - 1) + j];
                    } else {
                        const float normalized = (wl - S.rgb_min) / (S.rgb_max - S.rgb_min);
                        const float fi = normalized * ((float)count - 1.0f);
                        const float fmin_ = truncf(fi);
                        const uint32_t b0 = (uint32_t)fmin_;
                        const float bmix = fi - fmin_;
                        for (int j = 0; j < 3; ++j) resp[j] = d[3 * b0 + j] * (1.0f - bmix) + d[3 * (b0 + 1) + j] * bmix;
                    }
                }
                for (uint32_t j = 0; j < 3; ++j) spectral_values[(tape_rgb_row(n_spectral) + j) * BLOCK] = resp[j];
            }
        }
        for (uint32_t r0 = 0; r0 < pass_ops; r0 += ROWS) {
            unsigned long long rows[ROWS];
#pragma unroll
            for (uint32_t j = 0; j < ROWS; ++j) rows[j] = r0 + j < my_ops ? L.tape[tape_index(r0 + j, L.tape_lanes, tape_column)] : past_the_end;
            // all cross-lane reads of the batch first (one wait), then, in the eager form, all value reads (one more)
            uint32_t words[ROWS];
            float factors[ROWS];
#pragma unroll
            for (uint32_t j = 0; j < ROWS; ++j) {
                words[j] = (uint32_t)__shfl((int)(uint32_t)rows[j], (int)src);
                factors[j] = __uint_as_float((uint32_t)__shfl((int)(uint32_t)(rows[j] >> 32), (int)src));
            }
            if (eager) { // straight-line: m = value[slot] * s, then one of the two updates, chosen by selects
                float values[ROWS];
#pragma unroll
                for (uint32_t j = 0; j < ROWS; ++j) values[j] = spectral_values[words[j] & TAPE_EAGER_SLOT_MASK]; // slot << 8 is slot * BLOCK
                if (RGB && (TIMES ? S.micro_records : S.rgb_records) != 0) { // (uniform) HIT_RGB contributions: three coefficient records build the sum, the fourth applies it; PRODUCT: one to three and one
                    const uint32_t not_mine = hero ? 0u : TAPE_HERO_ONLY;
#pragma unroll
                    for (uint32_t j = 0; j < ROWS; ++j) {
                        const float m = values[j] * factors[j];
                        const uint32_t op = (words[j] >> TAPE_RGB_SHIFT) & (TIMES ? 7u : 3u);
                        if constexpr (TIMES)
                            rgb_sum = op == 1u ? m : (op == 2u ? rgb_sum + m : (op == 4u ? rgb_sum * m : rgb_sum));
                        else
                            rgb_sum = op == 1u ? m : (op == 2u ? rgb_sum + m : rgb_sum);
                        const float mm = op == 3u ? rgb_sum * factors[j] : m;
                        const bool apply = ((words[j] & not_mine) == 0u) & ((op == 0u) | (op == 3u));
                        const bool adds = (int)words[j] < 0;
                        const float multiplied = refl * mm, added = bright + multiplied;
                        bright = (apply & adds) ? added : bright;
                        refl = (apply & !adds) ? multiplied : refl;
                    }
                    continue;
                }
                if (S.hero_only_records == 0) { // (uniform) no record of this scene is for the hero alone: every record applies
#pragma unroll
                    for (uint32_t j = 0; j < ROWS; ++j) {
                        const float m = values[j] * factors[j];
                        const bool adds = (int)words[j] < 0;
                        const float multiplied = refl * m, added = bright + multiplied; // m * refl is refl * m
                        bright = adds ? added : bright;
                        refl = adds ? refl : multiplied;
                    }
                    continue;
                }
                const uint32_t not_for_me = hero ? 0u : TAPE_HERO_ONLY; // a hero-only record is skipped by the companions
#pragma unroll
                for (uint32_t j = 0; j < ROWS; ++j) {
                    const float m = values[j] * factors[j];
                    const bool apply = (words[j] & not_for_me) == 0u;
                    const bool adds = (int)words[j] < 0;
                    const float multiplied = refl * m, added = bright + multiplied;
                    bright = (apply & adds) ? added : bright;
                    refl = (apply & !adds) ? multiplied : refl;
                }
                continue;
            }
#pragma unroll
            for (uint32_t j = 0; j < ROWS; ++j) {
                const uint32_t word = words[j];
                const float s = factors[j];
                if (!run || r0 + j >= ops) continue;
                const uint32_t kind = word >> 30;
                if (kind == TAPE_SCALE) {
                    refl *= s;
                    continue;
                }
                if ((word & TAPE_HERO_ONLY) && !hero) continue;
                const uint32_t program = word & TAPE_PROGRAM_MASK;
                if (program != value_of) {
#ifdef PYR_REPLAY_NOEVAL
                    value = 1.0f;
#else
                    // the prepared form of a program: from the LDS table when the kernel staged one (a replay item changes
                    // program with nearly every record; fetching the program record from HBM each time was 16 % of the render)
                    Prepared q_prog;
                    if (program < L.tape_programs_lds) {
                        const uint32_t* e = prepared_lds + 8 * program;
                        q_prog.mode = e[0];
                        q_prog.c = __uint_as_float(e[1]);
                        q_prog.sp.format = e[2];
                        q_prog.sp.min = __uint_as_float(e[3]);
                        q_prog.sp.max = __uint_as_float(e[4]);
                        q_prog.sp.offset = e[5];
                        q_prog.sp.count = e[6];
                        q_prog.data = S.spectrum_data + e[5];
                        q_prog.id = program;
                    } else {
                        q_prog = prepare_program<false>(S, program);
                    }
                    VmInput in{wl, mk(0, 0, 0), mk(0, 0, 0)};
                    value = eval_prepared<false>(S, q_prog, in);
#endif
                    value_of = program;
                }
                if (kind == TAPE_MUL)
                    refl *= value * s;
                else
                    bright += value * s * refl;
            }
        }
#ifdef PYR_REPLAY_NOEXPOSE
        if (run && bright == 123.456f) expose_grain<COUNT>(L, pixel, wl, bright, cnt);
#else
        if (run) expose_grain<COUNT>(L, pixel, wl, bright, cnt);
#endif
    }
    __builtin_amdgcn_wave_barrier();
}

// The replay's LDS table of prepared programs (8 words each: mode, constant / scale, the spectrum record, and the slot of the
// program's value among the programs that read a spectrum), followed by the slot -> program list. Returns the number of
// spectrum-reading programs when they fit the kTapeEagerSlots value rows (the replay then looks each up once per item), else
// 0 (looked up record by record). Called by every thread of the workgroup; ends with a barrier.
DEV uint32_t prepare_tape_tables(const DevScene& S0, const DevScene& S, const RenderLaunch& L, uint32_t* prepared_lds, bool& eager) {
    // Programs that evaluate alike -- the same shape, factor and spectrum (a scene compiles one colour program per use: C3's
    // three white walls are three programs over one spectrum) -- share a value slot: the replay looks a slot up once per item.
    // (a LAMBDA program -- device_scene.h TapeForm: a number-only function of the wavelength, hit-tape scenes -- takes a slot of its own)
    auto reads_spectrum = [&](uint32_t i) {
        return S0.programs[i].kind != PYR_PROGRAM_CONSTANT && (S0.programs[i].fast != FAST_NONE || (S0.hit_tape != 0 && S0.programs[i].tape_form == TAPE_FORM_LAMBDA));
    };
    auto alike = [&](uint32_t i, uint32_t j) {
        const DevProgram &a = S0.programs[i], &b = S0.programs[j];
        return a.fast != FAST_NONE && a.fast == b.fast && __float_as_uint(a.fast_scale) == __float_as_uint(b.fast_scale) && a.fast_spectrum == b.fast_spectrum;
    };
    auto first_alike = [&](uint32_t i) { // the first spectrum-reading program that evaluates like program i
        for (uint32_t j = 0; j < i; ++j)
            if (reads_spectrum(j) && alike(i, j)) return j;
        return i;
    };
    auto slot_of = [&](uint32_t representative) { // its rank among the representatives
        uint32_t slot = 0;
        for (uint32_t j = 0; j < representative; ++j) slot += (reads_spectrum(j) && first_alike(j) == j) ? 1u : 0u;
        return slot;
    };
    uint32_t n_spectral = 0;
    for (uint32_t i = 0; i < L.tape_programs_lds; ++i) n_spectral += (reads_spectrum(i) && first_alike(i) == i) ? 1u : 0u;
    for (uint32_t i = threadIdx.x; i < L.tape_programs_lds; i += BLOCK) {
        const Prepared q = prepare_program<false>(S, i);
        const bool spectral = reads_spectrum(i);
        const uint32_t representative = spectral ? first_alike(i) : i;
        const uint32_t slot = spectral ? slot_of(representative) : 0u;
        uint32_t* e = prepared_lds + 8 * i;
        e[0] = q.mode, e[1] = __float_as_uint(q.c), e[2] = q.sp.format, e[3] = __float_as_uint(q.sp.min), e[4] = __float_as_uint(q.sp.max);
        e[5] = q.sp.offset, e[6] = q.sp.count, e[7] = tape_row(slot); // what a record carries: the value ROW
        if (spectral && representative == i && slot < kTapeMaxValueRows - 1u) prepared_lds[8 * L.tape_programs_lds + slot] = i;
    }
    __syncthreads();
    // The value rows (the last one holds 1.0; a scene with HIT_RGB programs keeps the RGB basis in three of them, behind the
    // programs' slots): too many spectrum-reading programs for them and the replay looks values up record by record. A scene
    // without any such program has nothing to look up eagerly -- unless it records hit-tape forms, which only the eager replay knows
    // (api.cpp makes sure they fit).
    eager = L.tape_programs_lds != 0 && tape_rows_needed(n_spectral, S0.rgb_records != 0) <= S0.tape_value_rows && (n_spectral != 0 || S0.hit_tape != 0);
    return eager ? n_spectral : 0u;
}

// Interpreter builds keep the program interpreter in line (an out-of-line copy spills the walker around every call: rounds 3 and 4)
// and are built for THREE waves per SIMD (168 VGPRs). Round 3 needed two (252 VGPRs: the pending contribution's sixteen words were
// carried across the stage loop; contribution_clear() ends that, 256 -> 208 VGPRs unconstrained); measured in round 4 at 2 / 3 / 4
// waves, hit tape: spheres 962 / 995 / 830, lamps 1005 / 1111 / 1063, textures 894 / 929 / 871 Msamples/s; the C3 mesh with a
// fresnel-coated rgb() material (tools/bench_interp_mesh.py), where the tree walk waits for memory: 246 / 320 / 308.
#ifndef PYR_SM_WAVES_INTERP
#define PYR_SM_WAVES_INTERP 3
#endif
constexpr int sm_waves(bool interp, bool /*lds_scene*/, bool /*hit_tape*/) { return interp ? PYR_SM_WAVES_INTERP : PYR_SM_WAVES; }
template <bool COUNT, bool INTERP, bool LDS_SCENE, bool LDS_TABLES, bool HIT_TAPE = false, bool PRODUCT = false>
__global__ __launch_bounds__(BLOCK, sm_waves(INTERP, LDS_SCENE, HIT_TAPE)) void render_kernel_sm(DevScene S0, RenderLaunch L) {
    extern __shared__ float lds[];
    static_assert(INTERP || !HIT_TAPE, "HIT_TAPE is a form of the interpreter build");
    static_assert(HIT_TAPE || !PRODUCT, "PRODUCT is a form of the hit tape (device_scene.h TapeForm)");
    constexpr bool TAPE = !INTERP || HIT_TAPE; // see "Spectral tape"; interpreter builds: scenes whose colour programs all have a tape form
    const uint32_t SS = L.spectrum_samples;
    Spectral spec{lds + threadIdx.x, SS};
    // LDS rows of 256 floats: TAPE: S wavelengths + one row of per-wave lane lists; else wavelengths / brightness / reflectance
    const uint32_t spectral_rows = TAPE ? SS + 1 + S0.tape_value_rows : 3 * SS;
    TravStack stack;
    int deep_levels[LDS_SCENE ? 1 : kMaxStackDepth]; // a scene that lives in LDS has its whole stack there (launch_render)
    stack.deep = deep_levels;
    stack.lds = (lds_int*)(reinterpret_cast<int*>(lds + spectral_rows * BLOCK) + threadIdx.x);
    stack.lds_entries = (int)L.stack_lds;
    Counters cnt{};
    const uint32_t lds_base_floats = (spectral_rows + L.stack_lds) * BLOCK;
    const SceneView view = stage_scene<LDS_SCENE>(S0, lds, lds_base_floats, true);
    const DevScene S = stage_tables<LDS_TABLES ? 1 : 0>(S0, lds, lds_base_floats + (LDS_SCENE ? (S0.num_nodes * 16 + S0.num_prims * 12) : 0));

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t waves_per_block = BLOCK / 64;
    const uint32_t total_waves = gridDim.x * waves_per_block;
    const int phase_lanes = (int)L.sm_phase_lanes, trav_steps = (int)L.sm_trav_steps;
    const int expose_lanes = TAPE ? (int)L.sm_expose_lanes : phase_lanes; // the replay works at full width whatever the count, but has a fixed cost per turn
    Walker<COUNT, INTERP, TAPE, PRODUCT> w;
    w.chunk = L.chunk_begin + blockIdx.x * waves_per_block + (threadIdx.x >> 6);
    w.tape_prepared = nullptr;
    uint32_t* wave_list = reinterpret_cast<uint32_t*>(lds + SS * BLOCK) + (threadIdx.x & ~63u);
    // prepared programs for the replay: 8 words each, behind everything else in LDS
    uint32_t* prepared_lds = reinterpret_cast<uint32_t*>(lds + lds_base_floats + (LDS_SCENE ? (S0.num_nodes * 16 + S0.num_prims * 12) : 0) + (LDS_TABLES ? S0.lds_table_floats : 0));
    // programs that read a spectrum get a slot (their rank among such programs); the slot -> program list follows the table
    uint32_t n_spectral = 0;
    bool eager = false;
    float* spectral_values = lds + (SS + 1) * BLOCK + threadIdx.x;
    if constexpr (TAPE) {
        // eager records of constant programs and BRDF factors (tape_push). TAPE builds only: without the tape these rows are not
        // reserved, and with few wavelengths the store landed in the staged scene (found by the fuzz campaign of round 3: two
        // wavelengths, a tree four levels deep -- row SS + 8 was the first KB of the LDS copy of the nodes)
        spectral_values[kTapeOneSlot * BLOCK] = 1.0f;
        n_spectral = prepare_tape_tables(S0, S, L, prepared_lds, eager);
        if (eager) w.tape_prepared = prepared_lds;
        w.rgb_slot = tape_rgb_row(n_spectral);
        w.tape_column = blockIdx.x * BLOCK + threadIdx.x;
        if (HIT_TAPE && !eager) *L.tape_overflow = 1u; // api.cpp only marks a scene hit_tape when its value slots fit: never taken
    }

    // the scene record as the phases see it (table pointers at the staged copies), rebuilt where a phase starts
    auto scene_view = [&](const RenderLaunch& Lp) {
        if (!PYR_RELOAD_LAUNCH || LDS_SCENE || INTERP) return S; // interpreter builds hand the record to run_interpreter by address: one copy in scratch, made once
        const uint32_t rows = (TAPE ? Lp.spectrum_samples + 1 + S0.tape_value_rows : 3 * Lp.spectrum_samples) + Lp.stack_lds;
        return stage_tables<LDS_TABLES ? 1 : 0, false>(scene_from_kernarg(S0), lds, rows * BLOCK);
    };
    PROF_DECL;
    for (;;) {
        // every decision looks at the lanes as they are now: a lane that has just been shaded and starts its next-event
        // estimation is counted for the NEE phase of this same turn, one that has just drawn a shadow ray for the traversal
        int nT = __popcll(ballot64(w.stage == ST_TRAV));
        int nS = __popcll(ballot64(w.stage == ST_SHADE));
        int nN = __popcll(ballot64(w.stage == ST_NEE));
        int nE = __popcll(ballot64(w.stage == ST_EXPOSE || w.stage == ST_NEW));
        if (max(max(nT, nS), max(nN, nE)) == 0) break; // every lane is DONE

        if (nE >= expose_lanes || nE == max(max(nT, nS), max(nN, nE))) {
            PHASE_PRIO(PYR_PRIO_E);
            PROF_BEGIN(0, w.stage == ST_EXPOSE || w.stage == ST_NEW);
            const RenderLaunch& Lp = launch_from_kernarg(L);
            const DevScene Sp = scene_view(Lp);
            // (without hit-tape forms the replay is eager exactly when it has value slots: one uniform less to keep across the loop)
            if constexpr (TAPE)
                replay_tapes<COUNT, HIT_TAPE, PRODUCT>(Sp, Lp, w.stage == ST_EXPOSE, w.n_ops, w.tape_column, w.p, lds, threadIdx.x, wave_list, prepared_lds, spectral_values, n_spectral,
                                              HIT_TAPE ? eager : n_spectral != 0, cnt);
            w.expose_and_restart(Sp, Lp, spec, cnt, lane, total_waves);
            PROF_END(0);
            nT = __popcll(ballot64(w.stage == ST_TRAV));
            nS = __popcll(ballot64(w.stage == ST_SHADE));
            nE = 0;
        }
        if (nS >= phase_lanes || nS == max(max(nT, nS), max(nN, nE))) {
            PHASE_PRIO(PYR_PRIO_S);
            PROF_BEGIN(1, w.stage == ST_SHADE);
            const RenderLaunch& Lp = launch_from_kernarg(L);
            w.shade(scene_view(Lp), Lp, spec, cnt);
            PROF_END(1);
            nT = __popcll(ballot64(w.stage == ST_TRAV));
            nN = __popcll(ballot64(w.stage == ST_NEE));
            nE = __popcll(ballot64(w.stage == ST_EXPOSE || w.stage == ST_NEW));
            nS = 0;
        }
        if (nN >= phase_lanes || nN == max(max(nT, nS), max(nN, nE))) {
            PHASE_PRIO(PYR_PRIO_N);
            PROF_BEGIN(2, w.stage == ST_NEE);
            const RenderLaunch& Lp = launch_from_kernarg(L);
            w.next_event(scene_view(Lp), Lp, spec, cnt);
            PROF_END(2);
            nT = __popcll(ballot64(w.stage == ST_TRAV));
            nN = __popcll(ballot64(w.stage == ST_NEE));
            nE = __popcll(ballot64(w.stage == ST_EXPOSE || w.stage == ST_NEW));
        }
        if constexpr (INTERP) { // what SHADE / NEE of this turn noted for `contribute`: one in-line interpreter (Walker::contribute_pending)
            const unsigned long long pending = ballot64(w.c_kind != 0u || w.c_has_scale);
            if (pending != 0ull) {
                [[maybe_unused]] const unsigned long long t_c0 = PROF_NOW();
                PROF_EXTRA(13, __popcll(pending));
                PROF_EXTRA(14, 1);
                const RenderLaunch& Lp = launch_from_kernarg(L);
                if constexpr (HIT_TAPE)
                    w.tape_pending(scene_view(Lp), Lp);
                else
                    w.contribute_pending(scene_view(Lp), Lp, spec);
                PROF_EXTRA(12, PROF_NOW() - t_c0);
            }
            w.contribution_clear();
        }
        // ---- TRAV: sm_trav_steps node / leaf steps of every lane with a ray in flight
        if (nT >= phase_lanes || nT == max(max(nT, nS), max(nN, nE))) {
#ifdef PYR_PHASE_PROFILE
            const unsigned long long prof_t0_3 = clock64();
#endif
            PHASE_PRIO(PYR_PRIO_T);
            w.t.inv = box_reciprocal(w.t.d); // 1 / direction for the box tests, live in this phase only
            trav_ray_signs(w.t);
            if (!LDS_SCENE && view.wide && view.pairs != nullptr) {
                for (int step = 0; step < trav_steps; ++step) {
                    PROF_LANES(3, w.stage == ST_TRAV);
                    if (trav_step_lean<COUNT>(view, w.t, stack, cnt, w.stage == ST_TRAV)) w.stage = w.t.shadow ? ST_NEE : ST_SHADE;
                }
            } else {
                for (int step = 0; step < trav_steps; ++step) {
                    PROF_LANES(3, w.stage == ST_TRAV);
                    if (trav_step_voted<COUNT, !LDS_SCENE>(view, w.t, stack, cnt, w.stage == ST_TRAV)) w.stage = w.t.shadow ? ST_NEE : ST_SHADE;
                }
            }
            PROF_END(3);
        }
    }
    PROF_FLUSH();
    flush_counters<COUNT>(cnt, L.counters);
}

// (Round 4 built a path-exchange scheduler here -- render_kernel_px, commit f7c5438: two waves of a workgroup walk the tree, two run
// the logic phases, a path's 35 words move between them through LDS slot arrays at the start and end of every ray; films and
// counters equal to the stage scheduler's. 0.4-0.5x: throughput followed the queues' LDS capacity, 128 / 225 / 269 Msamples/s
// at 16 / 32 / 64 slots against 592, and the copies cost more than half of what full lanes save. profiles/r04_phase_profile_px.txt
// holds the seat census and the arithmetic. Deleted with the other schedulers that lost.)

// ------------------------------------------------------------------------------------------------ work feed
// Hands the items [0, n) of a batch to persistent waves. One cursor word serves only ~88 M atomics/s (they serialise in
// one L2 channel), which capped a 4 M-item batch at 0.47 ms even with 128-item reservations, so the batch is cut into
// kFeedSegments contiguous segments with a cursor each (on its own cache line); a workgroup starts in segment
// blockIdx % 8 -- workgroups are dealt round-robin to the 8 XCDs, so an XCD mostly walks its own segment through its own
// L2 -- and moves on to the next segment when it finds one drained. All fields are wave-uniform.
struct WorkFeed {
    uint32_t next = 0, end = 0; // this wave's reserved slice
    uint32_t segment = 0, visited = 0;
    bool drained = false;
};
DEV uint32_t feed_segment_begin(uint32_t n, uint32_t segment) { return (uint32_t)(((uint64_t)n * segment / kFeedSegments) & ~63ull); }
DEV uint32_t feed_segment_end(uint32_t n, uint32_t segment) { return segment + 1 == kFeedSegments ? n : feed_segment_begin(n, segment + 1); }
// Makes sure the wave holds a non-empty slice, or marks the feed drained.
DEV void feed_reserve(WorkFeed& f, uint32_t* cursors, uint32_t n, uint32_t reserve, uint32_t lane) {
    while (f.next == f.end && !f.drained) {
        const uint32_t begin = feed_segment_begin(n, f.segment), end = feed_segment_end(n, f.segment);
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&cursors[f.segment * kFeedCursorStride], reserve);
        base = __shfl(base, 0, 64);
        if (base < end - begin) {
            f.next = begin + base;
            f.end = begin + min(base + reserve, end - begin);
        } else if (++f.visited == kFeedSegments) {
            f.drained = true;
        } else {
            f.segment = (f.segment + 1) % kFeedSegments;
        }
    }
}

// ------------------------------------------------------------------------------------------------ intersect kernel
// World::intersect for a batch of rays: persistent waves with dynamic ray fetch. A lane that finishes its ray does not
// wait for the slowest ray of the wave: when kRefillLanes lanes are idle the wave takes that many rays from the batch with
// ONE atomic (ballot + prefix count hand each idle lane its ray) and goes on stepping. On C3 the ray lengths are heavy
// tailed (most rays see 3-6 nodes, rays that graze the mesh 50-100); a one-ray-per-lane kernel ran at 7 % lane occupancy.
// Triangle-only scenes with a wide tree are walked as the render kernels walk them: the pair tree, straight-line steps
// (trav_step_lean). PYR_INTERSECT_PAIRS=0 keeps the 48-byte records and the generic step (A/B).
#ifndef PYR_INTERSECT_PAIRS
#define PYR_INTERSECT_PAIRS 1
#endif
// 93 VGPRs: five waves per SIMD (8.35 -> 9.75 Grays/s with the pair tree and the lean step at four waves, 10.3 at five; six
// waves spill: 6.1)
#ifndef PYR_INTERSECT_WAVES
#define PYR_INTERSECT_WAVES 5
#endif
template <bool COUNT>
__global__ __launch_bounds__(BLOCK, PYR_INTERSECT_WAVES) void intersect_kernel(DevScene S, IntersectLaunch L) {
    extern __shared__ int lds_stack[];
    TravStack stack;
    int deep_levels[kMaxStackDepth];
    stack.deep = deep_levels;
    stack.lds = (lds_int*)(lds_stack + threadIdx.x);
    stack.lds_entries = (int)L.stack_lds;
    Counters cnt{};
    SceneView view = wide_or_binary_view(S);
    const bool lean = PYR_INTERSECT_PAIRS && S.wide_nodes != nullptr && S.pair_prims != nullptr;
    if (lean) {
        view.nodes = reinterpret_cast<const float4*>(S.wide_pair_nodes);
        view.pairs = reinterpret_cast<const float4*>(S.pair_prims);
    }
    const uint32_t lane = threadIdx.x & 63u;
#ifndef PYR_INTERSECT_STEPS
#define PYR_INTERSECT_STEPS 4
#endif

#ifndef PYR_INTERSECT_REFILL
#define PYR_INTERSECT_REFILL 16
#endif
    constexpr int kRefillLanes = PYR_INTERSECT_REFILL, kSteps = PYR_INTERSECT_STEPS;
    bool busy = false;
    uint32_t ray = 0;
    Trav t{};
    WorkFeed feed;
    feed.segment = blockIdx.x % kFeedSegments;
    for (;;) {
        const unsigned long long idle_mask = ballot64(!busy);
        const int idle = __popcll(idle_mask);
        if (!feed.drained && (idle >= kRefillLanes || idle == 64)) {
            feed_reserve(feed, L.next, L.n, L.reserve, lane);
            if (!feed.drained) {
                const uint32_t available = feed.end - feed.next;
                const uint32_t rank = (uint32_t)__popcll(idle_mask & ((1ull << lane) - 1ull));
                if (!busy && rank < available) {
                    ray = feed.next + rank;
                    const float* r = L.rays + 6 * (size_t)ray;
                    trav_begin<COUNT>(S, t, ld3(r), ld3(r + 3), false, 0.0f, cnt);
                    t.inv = box_reciprocal(t.d);
                    trav_ray_signs(t);
                    busy = true;
                }
                feed.next += min((uint32_t)idle, available);
            }
        }
        if (ballot64(busy) == 0) {
            if (feed.drained) break;
            continue;
        }
        for (int step = 0; step < kSteps; ++step) {
            if (lean ? trav_step_lean<COUNT>(view, t, stack, cnt, busy) : trav_step_voted<COUNT>(view, t, stack, cnt, busy)) {
                PyrHit out;
                out.distance = t.shape != PYR_HIT_NONE ? t.closest : PYR_INF;
                out.shape = t.shape;
                out.u = t.u;
                out.v = t.v;
                L.hits[ray] = out;
                busy = false;
            }
        }
    }
    flush_counters<COUNT>(cnt, L.counters);
}

#if PYR_TU_MAIN
// ------------------------------------------------------------------------------------------------ film development
// main.rs:315-327: every pixel spectrum -> spectrum_to_xyz (main.rs:352-418, trapezoid rule against the CIE observer
// tables) -> linear sRGB -> sRGB u8. One thread per pixel; the film is read once (bins * 8 B per pixel), HBM-bound.
// The last pixel is skipped as in DevelopedPixels::next (film.rs:299).
__global__ __launch_bounds__(BLOCK) void develop_kernel(DevelopLaunch D) {
    const size_t pixels = (size_t)D.film.width * D.film.height;
    const uint32_t bins = D.film.bins;
    const float min = D.film.wl_start, max = D.film.wl_start + D.film.wl_width;
    for (size_t px = (size_t)blockIdx.x * BLOCK + threadIdx.x; px < pixels; px += (size_t)gridDim.x * BLOCK) {
        uint8_t out[3] = {0, 0, 0};
        if ((px + 1) * bins < pixels * bins) {
            const PyrGrain* g = D.grains + px * bins;
            auto xyz_get = [&](int channel, float w) {
                const float* d = D.xyz_table;
                const uint32_t n = D.xyz_count;
                if (w <= D.xyz_min) return d[channel];
                if (w >= D.xyz_max) return d[3 * (n - 1) + channel];
                float normalized = (w - D.xyz_min) / (D.xyz_max - D.xyz_min);
                float fi = normalized * ((float)n - 1.0f);
                float fmin_ = truncf(fi);
                uint32_t i0 = (uint32_t)fmin_;
                float mix = fi - fmin_;
                return d[3 * i0 + channel] * (1.0f - mix) + d[3 * (i0 + 1) + channel] * mix;
            };
            auto sample = [&](float w, uint32_t i) {
                float intensity;
                if (w < min || w > max) {
                    intensity = 0.0f;
                } else {
                    float normalized = (w - min) / (max - min);
                    float float_index = normalized * (float)bins;
                    uint32_t index = (uint32_t)fminf(floorf(float_index), (float)(bins - 1));
                    const PyrGrain gr = g[index];
                    intensity = gr.weight > 0.0f ? gr.acc / gr.weight : 0.0f; // Grain::develop, film.rs:132-143
                }
                if (D.filter) intensity = intensity * D.filter[i];
                if (D.white_div) intensity = (intensity / D.white_div[i]) * D.white_mul[i];
                return intensity;
            };
            float sum[3] = {0, 0, 0}, weight = 0.0f;
            float wl_min = min;
            uint32_t i = 0;
            float spectrum_min = sample(wl_min, i);
            float start[3] = {xyz_get(0, wl_min), xyz_get(1, wl_min), xyz_get(2, wl_min)};
            while (wl_min < max) {
                float wl_max = wl_min + D.step_size;
                i += 1;
                float spectrum_max = sample(wl_max, i < D.sample_count ? i : D.sample_count - 1);
                float end[3] = {xyz_get(0, wl_max), xyz_get(1, wl_max), xyz_get(2, wl_max)};
                float w = wl_max - wl_min;
                for (int c = 0; c < 3; ++c) sum[c] += (start[c] * spectrum_min + end[c] * spectrum_max) * 0.5f * w;
                weight += w;
                wl_min = wl_max;
                spectrum_min = spectrum_max;
                for (int c = 0; c < 3; ++c) start[c] = end[c];
            }
            float xyz[3];
            for (int c = 0; c < 3; ++c) xyz[c] = (weight == 0.0f ? sum[c] : sum[c] / weight) * D.xyz_scale;
            const float rgb[3] = {3.2404542f * xyz[0] + -1.5371385f * xyz[1] + -0.4985314f * xyz[2],
                                  -0.9692660f * xyz[0] + 1.8760108f * xyz[1] + 0.0415560f * xyz[2],
                                  0.0556434f * xyz[0] + -0.2040259f * xyz[1] + 1.0572252f * xyz[2]};
            for (int c = 0; c < 3; ++c) {
                float v = fminf(fmaxf(rgb[c], 0.0f), 1.0f);
                float e = v <= 0.0031308f ? 12.92f * v : 1.055f * (float)pow((double)v, 1.0 / 2.4) - 0.055f;
                e = fminf(fmaxf(e, 0.0f), 1.0f);
                out[c] = (uint8_t)(e * 255.0f + 0.5f);
            }
        }
        D.rgb_out[3 * px + 0] = out[0];
        D.rgb_out[3 * px + 1] = out[1];
        D.rgb_out[3 * px + 2] = out[2];
    }
}

int launch_develop(const DevelopLaunch& launch, void* stream) {
    const size_t pixels = (size_t)launch.film.width * launch.film.height;
    if (pixels == 0) return PYR_OK;
    uint32_t grid = (uint32_t)std::min<size_t>((pixels + BLOCK - 1) / BLOCK, 256 * 16);
    hipLaunchKernelGGL(develop_kernel, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, launch);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        g_kernel_error = std::string("develop kernel launch: ") + hipGetErrorString(err);
        return PYR_ERR_DEVICE;
    }
    return PYR_OK;
}

// ------------------------------------------------------------------------------------------------ film blocks -> film
// Rank 0's side of the multi-GPU gather: the blocks a rank rendered (PYR_FILM_TILE_BLOCKS: the tile's pixels plus a ring of
// one pixel) are added into the whole-image film. The pixels of a tile belong to one block of the set only, so they are
// plain read-modify-writes, one grain (8 bytes) per thread, contiguous along a pixel row in both buffers: HBM-bound, two
// reads and one write of 8 B per grain. Ring pixels lie inside neighbouring tiles -- which may be in the same set -- so
// they go second, as atomics, and only where something was exposed (about one sample in 1e6 lands there).
__global__ __launch_bounds__(BLOCK) void assemble_interior_kernel(AssembleLaunch A) {
    const uint32_t ts = A.tile_size, side = ts + 2u, bins = A.film.bins;
    const uint64_t row_grains = (uint64_t)ts * bins, tile_grains = row_grains * ts, total = tile_grains * A.tile_count;
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < total; i += (uint64_t)gridDim.x * BLOCK) {
        const uint32_t k = (uint32_t)(i / tile_grains);
        const uint64_t r = i - (uint64_t)k * tile_grains;
        const uint32_t row = (uint32_t)(r / row_grains), in_row = (uint32_t)(r - (uint64_t)row * row_grains);
        const uint32_t col = in_row / bins, bin = in_row - col * bins;
        const uint32_t tile = A.tile_begin + k * A.tile_stride;
        const uint32_t ty = tile / A.tiles_x, tx = tile - ty * A.tiles_x;
        const uint32_t x = tx * ts + col, y = ty * ts + row;
        if (x >= A.film.width || y >= A.film.height) continue; // a tile cut by the image border
        const PyrGrain g = A.blocks[(((size_t)k * side + row + 1u) * side + col + 1u) * bins + bin];
        PyrGrain* out = A.film_out + ((size_t)x + (size_t)y * A.film.width) * bins + bin;
        PyrGrain f = *out;
        f.acc += g.acc;
        f.weight += g.weight;
        *out = f;
    }
}
__global__ __launch_bounds__(BLOCK) void assemble_ring_kernel(AssembleLaunch A) {
    const uint32_t ts = A.tile_size, side = ts + 2u, bins = A.film.bins;
    const uint32_t ring = 4u * (ts + 1u); // pixels of the ring
    const uint64_t tile_grains = (uint64_t)ring * bins, total = tile_grains * A.tile_count;
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < total; i += (uint64_t)gridDim.x * BLOCK) {
        const uint32_t k = (uint32_t)(i / tile_grains);
        const uint32_t r = (uint32_t)(i - (uint64_t)k * tile_grains);
        const uint32_t q = r / bins, bin = r - q * bins;
        // ring pixel q: top row (side pixels), bottom row (side), then the left and right columns without their corners
        uint32_t bx, by;
        if (q < side)
            bx = q, by = 0u;
        else if (q < 2u * side)
            bx = q - side, by = side - 1u;
        else if (q < 2u * side + ts)
            bx = 0u, by = q - 2u * side + 1u;
        else
            bx = side - 1u, by = q - 2u * side - ts + 1u;
        const PyrGrain g = A.blocks[(((size_t)k * side + by) * side + bx) * bins + bin];
        if (g.acc == 0.0f && g.weight == 0.0f) continue;
        const uint32_t tile = A.tile_begin + k * A.tile_stride;
        const uint32_t ty = tile / A.tiles_x, tx = tile - ty * A.tiles_x;
        const uint32_t x = tx * ts + bx - 1u, y = ty * ts + by - 1u; // wraps for the ring left of / above the image
        if (x >= A.film.width || y >= A.film.height) continue;
        float* out = reinterpret_cast<float*>(A.film_out + ((size_t)x + (size_t)y * A.film.width) * bins + bin);
        atomicAdd(out, g.acc);
        atomicAdd(out + 1, g.weight);
    }
}

int launch_assemble(const AssembleLaunch& launch, void* stream) {
    if (launch.tile_count == 0) return PYR_OK;
    const uint64_t interior = (uint64_t)launch.tile_size * launch.tile_size * launch.film.bins * launch.tile_count;
    const uint64_t ring = (uint64_t)4 * (launch.tile_size + 1) * launch.film.bins * launch.tile_count;
    const uint32_t grid_i = (uint32_t)std::min<uint64_t>((interior + BLOCK - 1) / BLOCK, 256 * 32);
    const uint32_t grid_r = (uint32_t)std::min<uint64_t>((ring + BLOCK - 1) / BLOCK, 256 * 32);
    hipLaunchKernelGGL(assemble_interior_kernel, dim3(grid_i), dim3(BLOCK), 0, (hipStream_t)stream, launch);
    hipLaunchKernelGGL(assemble_ring_kernel, dim3(grid_r), dim3(BLOCK), 0, (hipStream_t)stream, launch);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        g_kernel_error = std::string("assemble kernel launch: ") + hipGetErrorString(err);
        return PYR_ERR_DEVICE;
    }
    return PYR_OK;
}

// ------------------------------------------------------------------------------------------------ launchers
constexpr size_t kLdsSceneBytes = 8 * 1024; // nodes + primitives staged in LDS when they fit (C1, C2: < 3 KB)
static bool scene_fits_lds(const DevScene& scene) { return (size_t)scene.num_nodes * 64 + (size_t)scene.num_prims * 48 <= kLdsSceneBytes; }
// Levels of the traversal stack kept in LDS. The synchronous walk keeps the whole stack there (its scenes are shallow). The
// resumable walk spills deeper levels to scratch (TravStack) and keeps as many levels in LDS as still let `workgroups`
// workgroups share a CU's 160 KB next to `other_bytes` of LDS each -- on C3 the render runs at 103 / 135 / 160 Msamples/s
// with 2 / 3 / 4 workgroups per CU and does not care whether 4 or 12 levels are in LDS. PYRITE_LDS_STACK overrides.
constexpr uint32_t kShortStackMax = 16;
static uint32_t short_stack_levels(const DevScene& scene, size_t other_bytes, uint32_t workgroups) {
    const char* e = std::getenv("PYRITE_LDS_STACK");
    uint32_t levels;
    if (e && *e) {
        levels = (uint32_t)std::strtoul(e, nullptr, 10);
    } else {
        const size_t budget = (160 * 1024) / std::max(workgroups, 1u);
        levels = budget > other_bytes ? (uint32_t)((budget - other_bytes) / (BLOCK * sizeof(int))) : 0u;
        levels = std::min(levels, kShortStackMax);
    }
    return std::max(1u, std::min(levels, scene.wide_nodes ? scene.wide_stack_depth : scene.stack_depth));
}
// Records a path can append: one MUL and one SCALE per bounce, light_samples ADDs in each of the two next-event estimations
// (tracer.rs:257), one closing ADD (emission or sky).
// A contribution whose colour program is HIT_RGB is four records (three coefficients and the factor).
uint32_t tape_ops_bound(const DevScene& scene, const RenderLaunch& launch) {
    return (launch.bounces + 2u * launch.light_samples + 1u) * (scene.micro_records ? 4u : 1u) + launch.bounces; // HIT_RGB: four records a contribution; PRODUCT: two to four
}
uint32_t tape_lanes_bound(int num_cus) { return (uint32_t)num_cus * 8u * BLOCK; } // launch_render never starts more than 8 blocks per CU
constexpr uint32_t kTapeProgramsLds = 128; // prepared programs kept in LDS for the replay (4 KB); scenes with more use the HBM records
static uint32_t tape_programs_in_lds(const DevScene& scene) { return scene.num_programs <= kTapeProgramsLds ? scene.num_programs : 0u; }
// Interpreter scenes record a tape when their colour programs allow it (DevScene::hit_tape) and there are wavelengths to share a
// hit's work among: with one or two per sample the online form wins (diamonds.lua, one wavelength, 256 bounces: 538 against 486).
bool uses_hit_tape(const DevScene& scene, const RenderLaunch& launch) {
    static const char* const least = std::getenv("PYRITE_HIT_TAPE_WAVELENGTHS"); // development: the fewest wavelengths per sample a hit tape is recorded for
    return scene.needs_interpreter != 0 && scene.hit_tape != 0 && launch.spectrum_samples >= (least && *least ? (uint32_t)std::strtoul(least, nullptr, 10) : 4u);
}
static bool uses_tape(const DevScene& scene, const RenderLaunch& launch) {
    return launch.scheduler == 1 && (scene.needs_interpreter == 0 || uses_hit_tape(scene, launch));
}
static size_t render_lds_bytes(const DevScene& scene, const RenderLaunch& launch) {
    const size_t spectral_rows = uses_tape(scene, launch) ? launch.spectrum_samples + 1 + scene.tape_value_rows : 3 * launch.spectrum_samples;
    size_t bytes = (spectral_rows + launch.stack_lds) * BLOCK * sizeof(float);
    if (scene_fits_lds(scene)) bytes += (size_t)scene.num_nodes * 64 + (size_t)scene.num_prims * 48;
    bytes += (size_t)scene.lds_table_floats * sizeof(float);
    if (uses_tape(scene, launch)) bytes += ((size_t)tape_programs_in_lds(scene) * 8 + kTapeMaxValueRows) * sizeof(uint32_t);
    return bytes;
}

#endif // PYR_TU_MAIN

using RenderKernel = void (*)(DevScene, RenderLaunch);
// A scene staged in LDS never has its tables staged too (api.cpp: lds_table_floats is only set for scenes that do not live in
// LDS), so that combination is never instantiated.
#if PYR_TU_PRODUCT
// The hit-tape interpreter builds for scenes with TAPE_FORM_PRODUCT colour programs: builds of their own (Walker::tape_pending says why)
RenderKernel pick_product_kernel(bool with_counters, bool lds_scene, bool lds_tables) {
    auto pick = [&](auto counters) -> RenderKernel {
        constexpr bool C = decltype(counters)::value;
        if (lds_scene) return render_kernel_sm<C, true, true, false, true, true>;
        return lds_tables ? render_kernel_sm<C, true, false, true, true, true> : render_kernel_sm<C, true, false, false, true, true>;
    };
    return with_counters ? pick(std::true_type{}) : pick(std::false_type{});
}
#else
RenderKernel pick_product_kernel(bool with_counters, bool lds_scene, bool lds_tables);
#endif
#if PYR_TU_INTERP
// The interpreter builds of the stage scheduler (the synchronous walk is built without the interpreter: a scene with interpreter
// programs always runs on the stage scheduler, which keeps the interpreter in line). HIT_TAPE: see device_scene.h TapeForm.
RenderKernel pick_interp_kernel(bool with_counters, bool lds_scene, bool lds_tables, bool hit_tape, bool product) {
    auto pick = [&](auto counters) -> RenderKernel {
        constexpr bool C = decltype(counters)::value;
        if (hit_tape && product) return pick_product_kernel(C, lds_scene, lds_tables);
        if (lds_scene) return hit_tape ? render_kernel_sm<C, true, true, false, true> : render_kernel_sm<C, true, true, false>;
        if (hit_tape) return lds_tables ? render_kernel_sm<C, true, false, true, true> : render_kernel_sm<C, true, false, false, true>;
        return lds_tables ? render_kernel_sm<C, true, false, true> : render_kernel_sm<C, true, false, false>;
    };
    return with_counters ? pick(std::true_type{}) : pick(std::false_type{});
}
#else
RenderKernel pick_interp_kernel(bool with_counters, bool lds_scene, bool lds_tables, bool hit_tape, bool product);
#endif

#if PYR_TU_MAIN
static RenderKernel pick_kernel(bool sm, bool with_counters, bool interp, bool lds_scene, bool lds_tables, bool hit_tape, bool product) {
#ifdef PYR_DEV_ONLY_SM // developer builds for reading the ISA (tools/asm_sm.sh): only the kernel the BASELINE meshes run is instantiated
    return render_kernel_sm<false, false, false, true>;
#endif
#ifdef PYR_DEV_ONLY_SM_INTERP // ... or only the interpreter build the reference's textures example runs
    return render_kernel_sm<false, true, true, false, true>;
#endif
    if (interp) return pick_interp_kernel(with_counters, lds_scene, lds_tables, hit_tape, product);
    auto pick = [&](auto counters) -> RenderKernel {
        constexpr bool C = decltype(counters)::value;
        if (lds_scene) return sm ? render_kernel_sm<C, false, true, false> : render_kernel<C, false, true, false>;
        if (sm) return lds_tables ? render_kernel_sm<C, false, false, true> : render_kernel_sm<C, false, false, false>;
        return lds_tables ? render_kernel<C, false, false, true> : render_kernel<C, false, false, false>;
    };
    return with_counters ? pick(std::true_type{}) : pick(std::false_type{});
}

bool scene_is_lds_resident(const DevScene& scene) { return scene_fits_lds(scene); }

int launch_render(const DevScene& scene, const RenderLaunch& launch_in, bool with_counters, void* stream, int num_cus) {
    if (scene.stack_depth > kMaxStackDepth) {
        g_kernel_error = "BVH deeper than kMaxStackDepth";
        return PYR_ERR_UNSUPPORTED;
    }
    RenderLaunch launch = launch_in;
    launch.stack_lds = 0;
    launch.tape_programs_lds = uses_tape(scene, launch) ? tape_programs_in_lds(scene) : 0u;
    // the stage-scheduled kernels are built for 4 waves per SIMD (__launch_bounds__(BLOCK, 4))
    launch.tape_programs_lds = uses_tape(scene, launch) ? tape_programs_in_lds(scene) : 0u;
    launch.stack_lds = launch.scheduler == 1 ? short_stack_levels(scene, render_lds_bytes(scene, launch), (uint32_t)sm_waves(scene.needs_interpreter != 0, scene_fits_lds(scene), uses_hit_tape(scene, launch)))
                                             : scene.stack_depth;
    // a scene staged in LDS is a few dozen nodes: its whole stack is kept in LDS (the kernels built for such scenes have no
    // scratch part: TravStack::deep is one entry), whatever the budget or PYRITE_LDS_STACK say; the 160 KB check below applies
    if (scene_fits_lds(scene)) launch.stack_lds = std::max(launch.stack_lds, scene.stack_depth);
    const size_t lds = render_lds_bytes(scene, launch);
    if (lds > 160 * 1024) {
        g_kernel_error = "spectrum_samples + BVH depth need more than 160 KB of LDS per workgroup";
        return PYR_ERR_UNSUPPORTED;
    }
    const uint32_t chunks = launch.chunk_end - launch.chunk_begin;
    if (chunks == 0) return PYR_OK;
    RenderKernel kernel = pick_kernel(launch.scheduler == 1, with_counters, scene.needs_interpreter != 0, scene_fits_lds(scene), scene.lds_table_floats != 0,
                                      launch.scheduler == 1 && uses_hit_tape(scene, launch), scene.product_records != 0);
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess) {
        g_kernel_error = std::string("hipFuncSetAttribute: ") + hipGetErrorString(err);
        return PYR_ERR_DEVICE;
    }
    // Residency of a 256-thread block (one wave per SIMD): waves per SIMD allowed by the 512-entry register file
    // (8-register granules, MI355X_MICROARCH.md "Register files") and by the 160 KB of LDS. The grid is persistent but needs
    // no co-residency (no inter-block hand-off), so an over-estimate only queues blocks.
    hipFuncAttributes attr{};
    int blocks_per_cu = 4;
    if (hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(kernel)) == hipSuccess && attr.numRegs > 0) {
        int regs = ((attr.numRegs + 7) / 8) * 8;
        blocks_per_cu = std::min(8, 512 / regs);
    }
    blocks_per_cu = std::max(1, std::min<int>(blocks_per_cu, (int)((160 * 1024) / std::max<size_t>(lds, 1))));
    uint32_t grid = (uint32_t)num_cus * (uint32_t)blocks_per_cu;
    const uint32_t path_waves = BLOCK / 64; // waves of a workgroup that take chunks
    uint32_t blocks_needed = (chunks + path_waves - 1) / path_waves;
    if (grid > blocks_needed) grid = blocks_needed;
    if (uses_tape(scene, launch) && (launch.tape == nullptr || (size_t)grid * BLOCK > launch.tape_lanes || launch.tape_max_ops < tape_ops_bound(scene, launch))) {
        g_kernel_error = "the spectral tape is missing or too small for this launch";
        return PYR_ERR_INVALID_ARGUMENT;
    }
    if (const char* cut = std::getenv("PYRITE_TEST_TAPE_OPS")) // test switch: pretend the bound were smaller, to see the overflow word work
        if (uses_tape(scene, launch) && *cut) launch.tape_max_ops = std::min<uint32_t>(launch.tape_max_ops, (uint32_t)std::strtoul(cut, nullptr, 10));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(BLOCK), lds, (hipStream_t)stream, scene, launch);
    err = hipGetLastError();
    if (err != hipSuccess) {
        g_kernel_error = std::string("render kernel launch: ") + hipGetErrorString(err);
        return PYR_ERR_DEVICE;
    }
    return PYR_OK;
}

int launch_intersect(const DevScene& scene, const IntersectLaunch& launch, bool with_counters, void* stream) {
    if (launch.n == 0) return PYR_OK;
    if (scene.stack_depth > kMaxStackDepth) {
        g_kernel_error = "BVH deeper than kMaxStackDepth";
        return PYR_ERR_UNSUPPORTED;
    }
    const uint32_t stack_lds = short_stack_levels(scene, 0, 8);
    const size_t lds = (size_t)stack_lds * BLOCK * sizeof(int);
    auto kernel = with_counters ? intersect_kernel<true> : intersect_kernel<false>;
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess) {
        g_kernel_error = std::string("hipFuncSetAttribute: ") + hipGetErrorString(err);
        return PYR_ERR_DEVICE;
    }
    // persistent grid: as many workgroups as the registers and the LDS stack let a CU hold (no co-residency is required)
    int blocks_per_cu = std::max(1, std::min<int>(8, (int)((160 * 1024) / std::max<size_t>(lds, 1))));
    uint32_t grid = (uint32_t)launch.num_cus * (uint32_t)blocks_per_cu;
    uint32_t needed = (launch.n + BLOCK - 1) / BLOCK;
    if (grid > needed) grid = needed;
    IntersectLaunch sized = launch;
    sized.stack_lds = stack_lds;
    // reservation per atomic: about a quarter of a wave's share of the batch, a multiple of 64, at most 2048
    const uint32_t waves = grid * (BLOCK / 64);
    sized.reserve = std::max<uint32_t>(64, std::min<uint32_t>(2048, (launch.n / std::max<uint32_t>(waves * 4, 1)) & ~63u));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(BLOCK), lds, (hipStream_t)stream, scene, sized);
    err = hipGetLastError();
    if (err != hipSuccess) {
        g_kernel_error = std::string("intersect kernel launch: ") + hipGetErrorString(err);
        return PYR_ERR_DEVICE;
    }
    return PYR_OK;
}

#endif // PYR_TU_MAIN

} // namespace pyr

#if defined(PYR_PHASE_PROFILE) && PYR_TU_MAIN
extern "C" int pyr_debug_phase_profile32(unsigned long long* out32, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(pyr::g_phase_prof), 32 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long zero[32] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(pyr::g_phase_prof), zero, sizeof(zero)) != hipSuccess) return -1;
    }
    return 0;
}
extern "C" int pyr_debug_phase_profile(unsigned long long* out16, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(pyr::g_phase_prof), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long zero[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(pyr::g_phase_prof), zero, sizeof(zero)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
