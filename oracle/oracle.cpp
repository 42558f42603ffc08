// oracle.cpp -- CPU restatement of Pyrite's camera-to-light renderer. TEST INFRASTRUCTURE ONLY.
//
// PARITY UNPINNED (see oracle.h): no reference test, golden vector or runnable reference exists for this
// path. Every function cites the reference lines it follows; paths are relative to
// /root/reference/pyrite/src/ unless they start with "collision", "cgmath", "rand", "palette"
// (third-party crates pinned in /root/reference/Cargo.lock and NOT present in the tree: their published
// algorithms are restated from memory and marked [3P]).
//
// Build: g++ -O2 -ffp-contract=off (Rust never fuses multiply-add; neither may this file).

#include "oracle.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_error;
int fail(int code, const std::string& message) {
    g_error = message;
    return code;
}

constexpr float DIST_EPSILON = 0.0001f; // math.rs:4
constexpr float PI = 3.14159265358979323846f; // std::f32::consts::PI
constexpr float INF = std::numeric_limits<float>::infinity();

// Transcendentals. Rust's f32::sin / cos / acos call the platform libm (sinf, ...), whose results are within an ulp of,
// but not always equal to, the correctly rounded value and differ between libms. So that the oracle and the HIP kernels
// agree bit for bit, both evaluate sin, cos and acos with the same single-precision Cephes kernels (S. Moshier's published
// sinf.c / asinf.c algorithms: Cody-Waite reduction by pi/4, minimax polynomials), as plain f32 operations in a fixed
// order; measured error < 2 ulp (tests/test_oracle_kat.py). exp and atan2 (blackbody; sphere uv) go through f64.
inline float sin32(float xx) {
    float x = std::fabs(xx);
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
inline float cos32(float xx) {
    float x = std::fabs(xx);
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
inline float asin32_core(float a) { // asin(a) for 0 <= a <= 1
    if (a < 1.0e-4f) return a;
    float x, z;
    const bool big = a > 0.5f;
    if (big) {
        z = 0.5f * (1.0f - a);
        x = std::sqrt(z);
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
inline float acos32(float x) {
    if (x < -0.5f) return 3.14159265358979323846f - 2.0f * asin32_core(std::sqrt(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * asin32_core(std::sqrt(0.5f * (1.0f - x)));
    float a = asin32_core(std::fabs(x));
    return 1.5707963267948966f - (x < 0.0f ? -a : a);
}
inline float exp32(float x) { return (float)std::exp((double)x); }
inline float atan2_32(float y, float x) { return (float)std::atan2((double)y, (double)x); }

// Rust f32::min / f32::max: IEEE minNum / maxNum (a NaN operand is ignored).
inline float rmin(float a, float b) { return std::fmin(a, b); }
inline float rmax(float a, float b) { return std::fmax(a, b); }

// ------------------------------------------------------------------------------------------------
// [3P] cgmath 0.17.0 Vector3 / Point3 arithmetic. dot = (x*x' + y*y') + z*z' (Array::sum), cross,
// magnitude = sqrt(dot), normalize_to(m) = v * (m / |v|), normalize = normalize_to(1).
// ------------------------------------------------------------------------------------------------
struct V3 {
    float x, y, z;
};
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline float magnitude2(V3 a) { return dot(a, a); }
inline float magnitude(V3 a) { return std::sqrt(magnitude2(a)); }
inline V3 normalize_to(V3 a, float m) { return a * (m / magnitude(a)); }
inline V3 normalize(V3 a) { return normalize_to(a, 1.0f); }
inline float axis(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

// [3P] cgmath 0.17 Quaternion {s, v}: the operations Normal uses (shapes/mod.rs:531-584), restated from the published
// source: From<Matrix3> (the trace / largest-diagonal branches of quaternion.rs), component-wise + and * f32, normalize via
// InnerSpace::normalize_to, Mul<Vector3> = (v x (v x vec + vec * s)) * 2 + vec, conjugate.
struct Quat {
    float s, x, y, z;
};
inline Quat quat_scale(Quat q, float f) { return Quat{q.s * f, q.x * f, q.y * f, q.z * f}; }
inline Quat quat_add(Quat a, Quat b) { return Quat{a.s + b.s, a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Quat quat_normalize(Quat q) {
    float m = std::sqrt(q.s * q.s + q.x * q.x + q.y * q.y + q.z * q.z); // dot: s*s + v.dot(v) [3P]
    return quat_scale(q, 1.0f / m);
}
inline Quat quat_conjugate(Quat q) { return Quat{q.s, -q.x, -q.y, -q.z}; }
inline V3 quat_rotate(Quat q, V3 vec) {
    V3 v = v3(q.x, q.y, q.z);
    V3 tmp = cross(v, vec) + vec * q.s;
    return cross(v, tmp) * 2.0f + vec;
}
// Matrix3::from_cols(c0, c1, c2).into(): m[c][r] is column c, row r.
inline Quat quat_from_cols(V3 c0, V3 c1, V3 c2) {
    const float m00 = c0.x, m01 = c0.y, m02 = c0.z, m10 = c1.x, m11 = c1.y, m12 = c1.z, m20 = c2.x, m21 = c2.y, m22 = c2.z;
    float trace = m00 + m11 + m22;
    if (trace >= 0.0f) {
        float s = std::sqrt(1.0f + trace);
        float w = 0.5f * s;
        s = 0.5f / s;
        return Quat{w, (m12 - m21) * s, (m20 - m02) * s, (m01 - m10) * s};
    } else if (m00 > m11 && m00 > m22) {
        float s = std::sqrt((m00 - m11 - m22) + 1.0f);
        float x = 0.5f * s;
        s = 0.5f / s;
        return Quat{(m12 - m21) * s, x, (m10 + m01) * s, (m02 + m20) * s};
    } else if (m11 > m22) {
        float s = std::sqrt((m11 - m00 - m22) + 1.0f);
        float y = 0.5f * s;
        s = 0.5f / s;
        return Quat{(m20 - m02) * s, (m10 + m01) * s, y, (m21 + m12) * s};
    } else {
        float s = std::sqrt((m22 - m00 - m11) + 1.0f);
        float z = 0.5f * s;
        s = 0.5f / s;
        return Quat{(m01 - m10) * s, (m02 + m20) * s, (m21 + m12) * s, z};
    }
}

struct Ray {
    V3 origin, direction;
};

// [3P] collision 0.20.1 Aabb3: new = component-wise min/max, grow, union, dim = max-min,
// center = min + dim/2, surface_area = 2*((dx*dy) + (dx*dz) + (dy*dz)).
struct Aabb {
    V3 min, max;
};
inline Aabb aabb_new(V3 a, V3 b) {
    return Aabb{v3(std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z)),
                v3(std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z))};
}
inline Aabb aabb_grow(Aabb a, V3 p) {
    return Aabb{v3(std::min(a.min.x, p.x), std::min(a.min.y, p.y), std::min(a.min.z, p.z)),
                v3(std::max(a.max.x, p.x), std::max(a.max.y, p.y), std::max(a.max.z, p.z))};
}
inline Aabb aabb_union(Aabb a, Aabb b) {
    return Aabb{v3(std::min(a.min.x, b.min.x), std::min(a.min.y, b.min.y), std::min(a.min.z, b.min.z)),
                v3(std::max(a.max.x, b.max.x), std::max(a.max.y, b.max.y), std::max(a.max.z, b.max.z))};
}
inline V3 aabb_dim(Aabb a) { return a.max - a.min; }
inline V3 aabb_center(Aabb a) { return a.min + aabb_dim(a) / 2.0f; }
inline float aabb_surface_area(Aabb a) {
    V3 d = aabb_dim(a);
    return 2.0f * ((d.x * d.y) + (d.x * d.z) + (d.y * d.z));
}

// ------------------------------------------------------------------------------------------------
// RNG. [3P] rand_xorshift 0.3.0 XorShiftRng::next_u32 (xorshift128, shifts 11/19/8); next_u64 = low word
// first (rand_core impls::next_u64_via_u32). The reference seeds one generator per TILE from thread_rng()
// (renderer/simple.rs:26-28,42) and is not reproducible; here every (tile, iteration) gets its own generator
// seeded by SplitMix64 of (seed, tile, iteration) -- the distributions and the per-sample draw order are the
// reference's, the stream layout is this build's (DESIGN.md "RNG").
// ------------------------------------------------------------------------------------------------
struct Rng {
    uint32_t x, y, z, w;
    uint32_t next_u32() {
        uint32_t t = x ^ (x << 11);
        x = y;
        y = z;
        z = w;
        w = w ^ (w >> 19) ^ (t ^ (t >> 8));
        return w;
    }
    uint64_t next_u64() {
        uint64_t lo = next_u32();
        uint64_t hi = next_u32();
        return (hi << 32) | lo;
    }
};

inline uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

inline Rng rng_seed(uint64_t seed, uint32_t tile, uint64_t iteration) {
    uint64_t counter = ((uint64_t)tile << 40) ^ iteration;
    uint64_t a = splitmix64(seed ^ splitmix64(counter));
    uint64_t b = splitmix64(a);
    Rng r{(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
    if ((r.x | r.y | r.z | r.w) == 0) r.w = 1; // xorshift must not start at zero
    return r;
}

// [3P] rand 0.8.5 Standard for f32: 24 random bits, [0,1).
inline float gen_f32(Rng& r) { return (float)(r.next_u32() >> 8) * (1.0f / 16777216.0f); }

// [3P] rand 0.8.5 UniformFloat<f32>::sample_single: 23 random bits -> [1,2) -> [0,1); res = v*scale + low;
// retry with scale decreased by one ulp when res rounds up to `high`.
inline float gen_range_f32(Rng& r, float low, float high) {
    float scale = high - low;
    for (;;) {
        uint32_t bits = (r.next_u32() >> 9) | 0x3F800000u;
        float value1_2;
        std::memcpy(&value1_2, &bits, 4);
        float value0_1 = value1_2 - 1.0f;
        float res = value0_1 * scale + low;
        if (res < high) return res;
        uint32_t sb;
        std::memcpy(&sb, &scale, 4);
        sb -= 1;
        std::memcpy(&scale, &sb, 4);
    }
}

// [3P] rand 0.8.5 UniformInt<usize>::sample_single (64-bit widening multiply with rejection zone).
inline uint32_t gen_range_usize(Rng& r, uint32_t n) {
    uint64_t range = n;
    uint64_t zone = (range << __builtin_clzll(range)) - 1;
    for (;;) {
        uint64_t v = r.next_u64();
        unsigned __int128 m = (unsigned __int128)v * range;
        uint64_t hi = (uint64_t)(m >> 64), lo = (uint64_t)m;
        if (lo <= zone) return (uint32_t)hi;
    }
}

// [3P] rand 0.8.5 SliceRandom::choose -> gen_index -> gen_range(0..len as u32): 32-bit widening multiply.
inline uint32_t choose_index(Rng& r, uint32_t n) {
    uint32_t range = n;
    uint32_t zone = (range << __builtin_clz(range)) - 1;
    for (;;) {
        uint32_t v = r.next_u32();
        uint64_t m = (uint64_t)v * range;
        uint32_t hi = (uint32_t)(m >> 32), lo = (uint32_t)m;
        if (lo <= zone) return hi;
    }
}

// ------------------------------------------------------------------------------------------------
// math.rs
// ------------------------------------------------------------------------------------------------

// math.rs:184-207 aabb_intersection_distance.
inline bool aabb_intersection_distance(const Aabb& aabb, const Ray& ray, float& out) {
    V3 inv = v3(1.0f / ray.direction.x, 1.0f / ray.direction.y, 1.0f / ray.direction.z);
    float t1 = (aabb.min.x - ray.origin.x) * inv.x;
    float t2 = (aabb.max.x - ray.origin.x) * inv.x;
    float tmin = rmin(t1, t2);
    float tmax = rmax(t1, t2);
    for (int i = 1; i < 3; ++i) {
        t1 = (axis(aabb.min, i) - axis(ray.origin, i)) * axis(inv, i);
        t2 = (axis(aabb.max, i) - axis(ray.origin, i)) * axis(inv, i);
        tmin = rmax(tmin, rmin(t1, t2));
        tmax = rmin(tmax, rmax(t1, t2));
    }
    if (tmax >= tmin && tmax >= 0.0f) {
        out = rmax(tmin, 0.0f);
        return true;
    }
    return false;
}

// math.rs:75-96 schlick.
inline float schlick(float n1, float n2, V3 normal, V3 incident) {
    float cos_psi = -dot(normal, incident);
    float r0 = (n1 - n2) / (n1 + n2);
    if (n1 > n2) {
        float n = n1 / n2;
        float sin_t2 = n * n * (1.0f - cos_psi * cos_psi);
        if (sin_t2 > 1.0f) return 1.0f;
        cos_psi = std::sqrt(1.0f - sin_t2);
    }
    float inv_cos = 1.0f - cos_psi;
    return r0 * r0 + (1.0f - r0 * r0) * inv_cos * inv_cos * inv_cos * inv_cos * inv_cos;
}

// math.rs:167-175 fresnel.
inline float fresnel(float ior, float env_ior, V3 normal, V3 incident) {
    if (dot(incident, normal) < 0.0f) return schlick(env_ior, ior, normal, incident);
    return schlick(ior, env_ior, -normal, incident);
}

// math.rs:98-114 ortho.
inline V3 ortho(V3 v) {
    V3 unit;
    if (std::fabs(v.x) < DIST_EPSILON)
        unit = v3(1, 0, 0);
    else if (std::fabs(v.y) < DIST_EPSILON)
        unit = v3(0, 1, 0);
    else if (std::fabs(v.z) < DIST_EPSILON)
        unit = v3(0, 0, 1);
    else
        unit = v3(-v.y, v.x, 0.0f);
    return cross(v, unit);
}

// math.rs:125-137 sample_cone.
inline V3 sample_cone(Rng& rng, V3 direction, float cos_half) {
    V3 o1 = normalize(ortho(direction));
    V3 o2 = normalize(cross(direction, o1));
    float r1 = PI * 2.0f * gen_f32(rng);
    float r2 = cos_half + (1.0f - cos_half) * gen_f32(rng);
    float oneminus = std::sqrt(1.0f - r2 * r2);
    return o1 * cos32(r1) * oneminus + o2 * sin32(r1) * oneminus + direction * r2;
}

// math.rs:139-145 solid_angle.
inline float solid_angle(float cos_half) {
    if (cos_half >= 1.0f) return 0.0f;
    return 2.0f * PI * (1.0f - cos_half);
}

// math.rs:147-153 sample_sphere.
inline V3 sample_sphere(Rng& rng) {
    float u = gen_f32(rng);
    float v = gen_f32(rng);
    float theta = 2.0f * PI * u;
    float phi = acos32(2.0f * v - 1.0f);
    return v3(sin32(phi) * cos32(theta), sin32(phi) * sin32(theta), cos32(phi));
}

// math.rs:155-164 sample_hemisphere.
inline V3 sample_hemisphere(Rng& rng, V3 direction) {
    V3 s = sample_sphere(rng);
    V3 x = normalize_to(ortho(direction), s.x);
    V3 y = normalize_to(cross(x, direction), s.y);
    V3 z = normalize_to(direction, std::fabs(s.z));
    return x + y + z;
}

// math.rs:177-182 blackbody; powi(-5) as LLVM expands it: a * (a^2)^2, then reciprocal.
inline float blackbody(float wavelength, float temperature) {
    float wl = wavelength * 1.0e-9f;
    float a2 = wl * wl;
    float a4 = a2 * a2;
    float powi = 1.0f / (wl * a4);
    float power_term = 3.74183e-16f * powi;
    return power_term / (exp32(1.4388e-2f / (wl * temperature)) - 1.0f);
}

// math.rs:22-72 Interpolated::get over (x,y) pairs.
inline float interpolated_get(const float* pts, uint32_t count, float input) {
    if (count == 0) return 0.0f;
    uint32_t min = 0, max = count - 1;
    if (pts[2 * min] >= input) return 0.0f;
    if (pts[2 * max] <= input) return 0.0f;
    while (max > min + 1) {
        uint32_t check = (max + min) / 2;
        float cx = pts[2 * check], cy = pts[2 * check + 1];
        if (cx == input) return cy;
        if (cx > input)
            max = check;
        else
            min = check;
    }
    float min_x = pts[2 * min], min_y = pts[2 * min + 1];
    float max_x = pts[2 * max], max_y = pts[2 * max + 1];
    if (input < min_x) return 0.0f;
    if (input > max_x) return 0.0f;
    return min_y + (max_y - min_y) * ((input - min_x) / (max_x - min_x));
}

// project/spectra.rs:30-58 Spectrum::get.
inline float spectrum_get(uint32_t format, float min, float max, const float* data, uint32_t count, float w) {
    if (format == PYR_SPECTRUM_ARRAY) {
        if (count == 0) return 0.0f;
        if (w <= min) return data[0];
        if (w >= max) return data[count - 1];
        float normalized = (w - min) / (max - min);
        float float_index = normalized * ((float)count - 1.0f);
        float min_float_index = std::trunc(float_index);
        uint32_t min_index = (uint32_t)min_float_index;
        uint32_t max_index = min_index + 1;
        float mix = float_index - min_float_index;
        return data[min_index] * (1.0f - mix) + data[max_index] * mix;
    }
    return interpolated_get(data, count, w);
}

// ------------------------------------------------------------------------------------------------
// Scene
// ------------------------------------------------------------------------------------------------
struct Triangle { // Shape::Triangle, shapes/mod.rs:39-46
    V3 p1, p2, p3;
    V3 n1, n2, n3;
    float t1[2], t2[2], t3[2];
    V3 edge1, edge2;
    uint32_t material;
    Quat f1, f2, f3; // Vertex.normal.from_space
};
struct Sphere { // Shape::Sphere, shapes/mod.rs:33-38
    V3 position;
    float radius;
    float tex_scale[2];
    uint32_t material;
};
struct Plane { // shapes::Plane, shapes/mod.rs:434-439 ([3P] collision::Plane {n, d})
    V3 origin, normal;
    float tex_scale[2];
    uint32_t material;
    Quat frame; // normal.from_space
};
struct Texture { // texture.rs:18-22; channels = 4 (LinSrgba) or 1 (LinLuma)
    uint32_t width, height, channels;
    const float* data;
};

struct ShapeRef { // &Shape
    uint32_t kind; // PyrShapeKind
    uint32_t index;
};

struct FlatNode { // FlatBvhNode, spatial/bvh.rs:289-306
    Aabb bounding_box;
    uint32_t subtree_size; // 0 for a leaf
    ShapeRef item;         // leaf only
};

struct Intersection { // shapes/mod.rs:472-482 with ShapeSurfacePoint inlined
    float distance;
    V3 position;
    ShapeRef shape;
    float u, v;
};

struct SurfaceData { // shapes/mod.rs:526-529 (Normal = vector + from_space, :531-535)
    V3 normal;
    Quat from_space;
    float texture[2];
};

} // namespace

struct OracleScene {
    std::vector<Triangle> triangles;
    std::vector<Sphere> spheres;
    std::vector<Plane> planes;
    std::vector<PyrLamp> lamps;
    std::vector<PyrMaterial> materials;
    std::vector<PyrComponent> components;
    std::vector<PyrProgram> programs;
    std::vector<PyrInstr> instrs;
    std::vector<PyrSpectrum> spectra;
    std::vector<float> spectrum_data;
    std::vector<float> rgb_basis;
    float rgb_basis_min = 0, rgb_basis_max = 0;
    uint32_t sky_program = 0;
    std::vector<FlatNode> nodes; // Bvh::nodes
    std::vector<float> texture_data;
    std::vector<Texture> textures; // Resources.textures
};

namespace {

struct alignas(128) Counters { // one per worker thread, bumped at every box test: a cache line pair of its own, or 256 threads fight over lines
    uint64_t samples = 0, extension_rays = 0, shadow_rays = 0, box_tests = 0, triangle_tests = 0, sphere_tests = 0,
             plane_tests = 0, shaded_hits = 0, exposures = 0;
    void add_to(PyrCounters* c) const {
        c->samples += samples;
        c->extension_rays += extension_rays;
        c->shadow_rays += shadow_rays;
        c->box_tests += box_tests;
        c->triangle_tests += triangle_tests;
        c->sphere_tests += sphere_tests;
        c->plane_tests += plane_tests;
        c->shaded_hits += shaded_hits;
        c->exposures += exposures;
    }
};

// ------------------------------------------------------------------------------------------------
// shapes/mod.rs
// ------------------------------------------------------------------------------------------------

// shapes/mod.rs:75-119 Moeller-Trumbore, two-sided, absolute epsilons.
inline bool triangle_intersect(const Triangle& tri, const Ray& ray, float& dist, float& u, float& v) {
    V3 e1 = tri.edge1, e2 = tri.edge2;
    V3 p = cross(ray.direction, e2);
    float det = dot(e1, p);
    if (det > -DIST_EPSILON && det < DIST_EPSILON) return false;
    float inv_det = 1.0f / det;
    V3 t = ray.origin - tri.p1;
    u = dot(t, p) * inv_det;
    if (u < 0.0f || u > 1.0f) return false;
    V3 q = cross(t, e1);
    v = dot(ray.direction, q) * inv_det;
    if (v < 0.0f || u + v > 1.0f) return false;
    dist = dot(e2, q) * inv_det;
    return dist > DIST_EPSILON;
}

// shapes/mod.rs:57-74 -> [3P] collision 0.20.1 `Continuous<Ray3> for Sphere`:
// l = c-o; tca = l.d; tca<0 -> None; d2 = l.l - tca^2; d2>r^2 -> None; thc = sqrt(r^2-d2); P = o + d*(tca-thc).
// pyrite then takes distance = |P - o|.
inline bool sphere_intersect(V3 center, float radius, const Ray& ray, float& dist, V3& point) {
    V3 l = center - ray.origin;
    float tca = dot(l, ray.direction);
    if (tca < 0.0f) return false;
    float d2 = dot(l, l) - tca * tca;
    if (d2 > radius * radius) return false;
    float thc = std::sqrt(radius * radius - d2);
    point = ray.origin + ray.direction * (tca - thc);
    dist = magnitude(point - ray.origin);
    return true;
}

// shapes/mod.rs:441-452 -> [3P] collision::Plane::intersection. The crate's sign convention for `d` could not be
// checked offline; this restates the geometrically intended plane through `origin`:
// t = ((origin - o).n) / (d.n); t < 0 -> None; P = o + d*t; distance = |P - o|. Not used by configs C1-C5.
inline bool plane_intersect(const Plane& pl, const Ray& ray, float& dist, V3& point) {
    float t = (dot(pl.origin, pl.normal) - dot(ray.origin, pl.normal)) / dot(ray.direction, pl.normal);
    if (!(t >= 0.0f)) return false; // NaN (parallel ray inside the plane) misses
    point = ray.origin + ray.direction * t;
    dist = magnitude(point - ray.origin);
    return true;
}

// Bounded::aabb, shapes/mod.rs:408-432.
inline Aabb shape_aabb(const OracleScene& s, ShapeRef r) {
    if (r.kind == PYR_SHAPE_SPHERE) {
        const Sphere& sp = s.spheres[r.index];
        V3 rr = v3(sp.radius, sp.radius, sp.radius);
        return aabb_new(sp.position - rr, sp.position + rr);
    }
    const Triangle& t = s.triangles[r.index];
    return aabb_grow(aabb_new(t.p1, t.p2), t.p3);
}

// Shape::ray_intersect, shapes/mod.rs:55-156.
inline bool shape_intersect(const OracleScene& s, ShapeRef r, const Ray& ray, Intersection& out, Counters& c) {
    out.shape = r;
    out.u = out.v = 0.0f;
    if (r.kind == PYR_SHAPE_SPHERE) {
        c.sphere_tests++;
        const Sphere& sp = s.spheres[r.index];
        return sphere_intersect(sp.position, sp.radius, ray, out.distance, out.position);
    }
    c.triangle_tests++;
    const Triangle& t = s.triangles[r.index];
    if (!triangle_intersect(t, ray, out.distance, out.u, out.v)) return false;
    out.position = ray.origin + ray.direction * out.distance;
    return true;
}

// SurfacePoint::get_surface_data, shapes/mod.rs:484-494 (+ :346-385, :454-469, :550-558).
inline SurfaceData surface_data(const OracleScene& s, const Intersection& hit) {
    SurfaceData sd;
    if (hit.shape.kind == PYR_SHAPE_SPHERE) {
        const Sphere& sp = s.spheres[hit.shape.index];
        V3 normal = normalize(hit.position - sp.position);
        float latitude = acos32(normal.y);
        float longitude = atan2_32(normal.x, normal.z);
        // Matrix3::from_angle_y(longitude) * Matrix3::from_angle_x(latitude - PI/2) [3P cgmath: columns (c,0,-s),(0,1,0),(s,0,c)
        // and (1,0,0),(0,c,s),(0,-s,c); the product's column j is A * B.col(j), each entry a row(i).dot(col) summed x, y, z]
        float sy = sin32(longitude), cy = cos32(longitude);
        float ax = latitude - PI * 0.5f;
        float sx = sin32(ax), cx = cos32(ax);
        auto rowdot = [](float a0, float a1, float a2, V3 b) { return a0 * b.x + a1 * b.y + a2 * b.z; };
        // A rows: (cy, 0, sy), (0, 1, 0), (-sy, 0, cy); B columns: (1,0,0), (0,cx,sx), (0,-sx,cx)
        V3 bc[3] = {v3(1.0f, 0.0f, 0.0f), v3(0.0f, cx, sx), v3(0.0f, -sx, cx)};
        V3 col[3];
        for (int j = 0; j < 3; ++j) col[j] = v3(rowdot(cy, 0.0f, sy, bc[j]), rowdot(0.0f, 1.0f, 0.0f, bc[j]), rowdot(-sy, 0.0f, cy, bc[j]));
        sd.normal = normal;
        sd.from_space = quat_from_cols(col[0], col[1], col[2]);
        sd.texture[0] = (longitude * (1.0f / PI) * 0.5f) / sp.tex_scale[0];
        sd.texture[1] = (1.0f - (latitude * (1.0f / PI))) / sp.tex_scale[1];
    } else if (hit.shape.kind == PYR_SHAPE_TRIANGLE) {
        const Triangle& t = s.triangles[hit.shape.index];
        float u = hit.u, v = hit.v;
        float w = 1.0f - (u + v);
        sd.normal = normalize(t.n1 * w + t.n2 * u + t.n3 * v);
        sd.from_space = quat_normalize(quat_add(quat_add(quat_scale(t.f1, w), quat_scale(t.f2, u)), quat_scale(t.f3, v))); // :552
        sd.texture[0] = t.t1[0] * w + t.t2[0] * u + t.t3[0] * v;
        sd.texture[1] = t.t1[1] * w + t.t2[1] * u + t.t3[1] * v;
    } else {
        const Plane& p = s.planes[hit.shape.index];
        sd.normal = p.normal;
        sd.from_space = p.frame;
        V3 normal_space = quat_rotate(quat_conjugate(p.frame), hit.position); // Normal::into_space, :568-570
        sd.texture[0] = normal_space.x / p.tex_scale[0];
        sd.texture[1] = normal_space.y / p.tex_scale[1];
    }
    return sd;
}

inline uint32_t shape_material(const OracleScene& s, ShapeRef r) {
    if (r.kind == PYR_SHAPE_SPHERE) return s.spheres[r.index].material;
    if (r.kind == PYR_SHAPE_TRIANGLE) return s.triangles[r.index].material;
    return s.planes[r.index].material;
}

// ------------------------------------------------------------------------------------------------
// spatial/bvh.rs
// ------------------------------------------------------------------------------------------------
struct Hull { // bvh.rs:318-370
    Aabb aabbs, centroids;
};
inline Hull hull_new(Aabb a) { return Hull{a, aabb_new(aabb_center(a), aabb_center(a))}; }
inline Hull hull_expand(const Hull& h, const Aabb& a) { return Hull{aabb_union(h.aabbs, a), aabb_grow(h.centroids, aabb_center(a))}; }
inline Hull hull_join(const Hull& h, const Hull& o) { return Hull{aabb_union(h.aabbs, o.aabbs), aabb_union(h.centroids, o.centroids)}; }
inline void hull_largest_axis(const Hull& h, float& width, int& ax) { // bvh.rs:355-369
    V3 d = aabb_dim(h.centroids);
    if (d.y > d.x) {
        width = d.y;
        ax = 1;
    } else {
        width = d.x;
        ax = 0;
    }
    if (d.z > width) {
        width = d.z;
        ax = 2;
    }
}

struct TreeNode { // BvhNode, bvh.rs:237-248
    Aabb bounding_box;
    uint32_t subtree_size;
    int32_t first = -1, second = -1; // indices into the pool; leaf when first < 0
    ShapeRef item{0, 0};
};

// Bvh::new, bvh.rs:13-155, including flatten (:250-275).
void build_bvh(OracleScene& s) {
    std::vector<ShapeRef> items;
    for (uint32_t i = 0; i < s.spheres.size(); ++i) items.push_back(ShapeRef{PYR_SHAPE_SPHERE, i});
    for (uint32_t i = 0; i < s.triangles.size(); ++i) items.push_back(ShapeRef{PYR_SHAPE_TRIANGLE, i});
    s.nodes.clear();
    if (items.empty()) return;

    struct Entry {
        bool join;
        Aabb bounding_box;       // join
        std::vector<ShapeRef> items; // split
        Hull hull;
    };
    std::vector<TreeNode> pool;
    std::vector<int32_t> nodes; // the `nodes` stack of bvh.rs:19
    std::vector<Entry> stack;

    Hull hull = hull_new(shape_aabb(s, items[0]));
    for (const ShapeRef& it : items) hull = hull_expand(hull, shape_aabb(s, it));
    stack.push_back(Entry{false, Aabb{}, std::move(items), hull});

    while (!stack.empty()) {
        Entry entry = std::move(stack.back());
        stack.pop_back();
        if (entry.join) {
            int32_t first = nodes.back();
            nodes.pop_back();
            int32_t second = nodes.back();
            nodes.pop_back();
            TreeNode n;
            n.bounding_box = entry.bounding_box;
            n.subtree_size = pool[first].subtree_size + pool[second].subtree_size + 2;
            n.first = first;
            n.second = second;
            pool.push_back(n);
            nodes.push_back((int32_t)pool.size() - 1);
            continue;
        }
        if (entry.items.size() == 1) {
            TreeNode n;
            n.bounding_box = entry.hull.aabbs;
            n.subtree_size = 0;
            n.item = entry.items[0];
            pool.push_back(n);
            nodes.push_back((int32_t)pool.size() - 1);
            continue;
        }
        float width;
        int ax;
        hull_largest_axis(entry.hull, width, ax);
        std::vector<ShapeRef> first_items, second_items;
        Hull first_hull, second_hull;
        if (width < DIST_EPSILON) {
            size_t half = entry.items.size() / 2;
            first_items.assign(entry.items.begin(), entry.items.begin() + half);
            second_items.assign(entry.items.begin() + half, entry.items.end());
            first_hull = hull_new(shape_aabb(s, first_items[0]));
            for (const ShapeRef& it : first_items) first_hull = hull_expand(first_hull, shape_aabb(s, it));
            second_hull = hull_new(shape_aabb(s, second_items[0]));
            for (const ShapeRef& it : second_items) second_hull = hull_expand(second_hull, shape_aabb(s, it));
        } else {
            constexpr int BUCKETS = 6;
            std::vector<ShapeRef> bucket_items[BUCKETS];
            Hull bucket_hull[BUCKETS];
            bool bucket_used[BUCKETS] = {false, false, false, false, false, false};
            float min_bound = axis(entry.hull.centroids.min, ax);
            for (const ShapeRef& it : entry.items) {
                Aabb bb = shape_aabb(s, it);
                float position = axis(aabb_center(bb), ax);
                float float_index = (float)BUCKETS * (position - min_bound) / width;
                // `as usize` saturates: negative / NaN -> 0
                int index = float_index > 0.0f ? (int)std::min(float_index, 1.0e9f) : 0;
                index = std::min(index, BUCKETS - 1);
                if (bucket_used[index]) {
                    bucket_items[index].push_back(it);
                    bucket_hull[index] = hull_expand(bucket_hull[index], bb);
                } else {
                    bucket_used[index] = true;
                    bucket_items[index].push_back(it);
                    bucket_hull[index] = hull_new(bb);
                }
            }
            auto stats = [&](int from, int to, size_t& count, float& area) { // get_bucket_stats, bvh.rs:167-183
                count = 0;
                bool any = false;
                Aabb acc{};
                for (int b = from; b < to; ++b) {
                    if (!bucket_used[b]) continue;
                    acc = any ? aabb_union(acc, bucket_hull[b].aabbs) : bucket_hull[b].aabbs;
                    any = true;
                    count += bucket_items[b].size();
                }
                area = any ? aabb_surface_area(acc) : 0.0f;
            };
            float min_cost = INF;
            int min_cost_split = 0;
            float hull_area = aabb_surface_area(entry.hull.aabbs);
            for (int index = 1; index < BUCKETS; ++index) {
                size_t c1, c2;
                float a1, a2;
                stats(0, index, c1, a1);
                stats(index, BUCKETS, c2, a2);
                float cost = (a1 * (float)c1 + a2 * (float)c2) / hull_area;
                if (cost < min_cost) {
                    min_cost_split = index;
                    min_cost = cost;
                }
            }
            auto merge = [&](int from, int to, std::vector<ShapeRef>& out_items, Hull& out_hull) { // merge_buckets :185-199
                bool any = false;
                for (int b = from; b < to; ++b) {
                    if (!bucket_used[b]) continue;
                    out_hull = any ? hull_join(bucket_hull[b], out_hull) : bucket_hull[b];
                    any = true;
                    out_items.insert(out_items.end(), bucket_items[b].begin(), bucket_items[b].end());
                }
            };
            merge(0, min_cost_split, first_items, first_hull);
            merge(min_cost_split, BUCKETS, second_items, second_hull);
        }
        stack.push_back(Entry{true, entry.hull.aabbs, {}, Hull{}});
        stack.push_back(Entry{false, Aabb{}, std::move(second_items), second_hull});
        stack.push_back(Entry{false, Aabb{}, std::move(first_items), first_hull});
    }

    // BvhNode::flatten, bvh.rs:250-275: pre-order, `first` before `second`.
    std::vector<int32_t> fstack{nodes.back()};
    while (!fstack.empty()) {
        int32_t id = fstack.back();
        fstack.pop_back();
        const TreeNode& n = pool[id];
        FlatNode f;
        f.bounding_box = n.bounding_box;
        f.subtree_size = n.subtree_size;
        f.item = n.item;
        if (n.first >= 0) {
            fstack.push_back(n.second);
            fstack.push_back(n.first);
        }
        s.nodes.push_back(f);
    }
}

// World::intersect, world.rs:273-299, with Intersections::next (bvh.rs:207-229) inlined.
bool world_intersect(const OracleScene& s, const Ray& ray, Intersection& result, Counters& c) {
    bool found = false;
    float closest_distance = INF;
    for (uint32_t i = 0; i < s.planes.size(); ++i) {
        Intersection it;
        it.shape = ShapeRef{PYR_SHAPE_PLANE, i};
        it.u = it.v = 0.0f;
        c.plane_tests++;
        if (plane_intersect(s.planes[i], ray, it.distance, it.position)) {
            if (it.distance > DIST_EPSILON && it.distance < closest_distance) {
                closest_distance = it.distance;
                result = it;
                found = true;
            }
        }
    }
    size_t i = 0, n = s.nodes.size();
    while (i < n) {
        const FlatNode& node = s.nodes[i];
        i += 1;
        c.box_tests++;
        float distance;
        if (aabb_intersection_distance(node.bounding_box, ray, distance)) {
            if (distance >= closest_distance) {
                i += node.subtree_size;
                continue;
            }
            if (node.subtree_size == 0) {
                Intersection it;
                if (shape_intersect(s, node.item, ray, it, c)) {
                    if (it.distance > DIST_EPSILON && it.distance < closest_distance) {
                        closest_distance = it.distance;
                        result = it;
                        found = true;
                    }
                }
            }
        } else {
            i += node.subtree_size;
        }
    }
    return found;
}

// ------------------------------------------------------------------------------------------------
// program/execution_context.rs -- the register VM
// ------------------------------------------------------------------------------------------------
struct V4 {
    float x, y, z, w;
};

struct ProgramInput { // RenderContext (tracer.rs:72-77) / ProbabilityInput (materials/mod.rs:251-257)
    float wavelength;
    V3 normal, incident;
    float texture[2];
    bool wavelength_used = false; // ProbabilityInput::wavelength_used
};

// texture.rs:297-334 (cubic_interpolate, bicubic_interpolate) on one channel; LinSrgba's +, -, * are component-wise.
inline float cubic_interpolate(float v1, float v2, float v3_, float v4, float pos) {
    float a = (v4 - v3_) - (v1 - v2);
    float b = (v1 - v2) - a;
    float c = v3_ - v1;
    float d = v2;
    return d + (c + (b + a * pos) * pos) * pos;
}
// isize::rem_euclid for a positive modulus.
inline int64_t rem_euclid(int64_t a, int64_t m) {
    int64_t r = a % m;
    return r < 0 ? r + m : r;
}
// Texture::get_color, texture.rs:87-150: 4 x 4 texels with wrap-around, rows from the top (y flipped), bicubic.
inline void texture_get_color(const Texture& t, float px, float py, float* out) {
    float width_f = (float)t.width, height_f = (float)t.height;
    float x = px * width_f - 0.5f;
    float x_floor = std::floor(x);
    int64_t w = t.width, h = t.height;
    int64_t xs[4], ys[4];
    // Rust `as isize` saturates and maps NaN to 0
    auto to_isize = [](float f) -> int64_t {
        if (f != f) return 0;
        if (f >= 9.2233720368547758e18f) return INT64_MAX;
        if (f <= -9.2233720368547758e18f) return INT64_MIN;
        return (int64_t)f;
    };
    xs[1] = rem_euclid(to_isize(x_floor), w);
    xs[0] = xs[1] == 0 ? w - 1 : xs[1] - 1;
    xs[2] = xs[1] == w - 1 ? 0 : xs[1] + 1;
    xs[3] = xs[2] == w - 1 ? 0 : xs[2] + 1;
    float y = (1.0f - py) * height_f - 0.5f;
    float y_floor = std::floor(y);
    ys[1] = rem_euclid(to_isize(y_floor), h);
    ys[0] = ys[1] == 0 ? h - 1 : ys[1] - 1;
    ys[2] = ys[1] == h - 1 ? 0 : ys[1] + 1;
    ys[3] = ys[2] == h - 1 ? 0 : ys[2] + 1;
    float fx = x - x_floor, fy = y - y_floor;
    for (uint32_t ch = 0; ch < t.channels; ++ch) {
        float rows[4];
        for (int r = 0; r < 4; ++r) {
            float v[4];
            for (int k = 0; k < 4; ++k) v[k] = t.data[((size_t)xs[k] + (size_t)ys[r] * t.width) * t.channels + ch];
            rows[r] = cubic_interpolate(v[0], v[1], v[2], v[3], fx);
        }
        out[ch] = cubic_interpolate(rows[0], rows[1], rows[2], rows[3], fy);
    }
}

struct Exe { // ExecutionContext, execution_context.rs:15-18 (+ Registers, registers.rs)
    const OracleScene* scene;
    std::vector<float> number;
    std::vector<V4> vector;
    std::vector<V4> rgb; // LinSrgba
    explicit Exe(const OracleScene* s) : scene(s) {}

    void reserve(const PyrProgram& p) { // registers.rs:44-51
        if (number.size() < p.num_numbers) number.resize(p.num_numbers, 0.0f);
        if (vector.size() < p.num_vectors) vector.resize(p.num_vectors, V4{0, 0, 0, 0});
        if (rgb.size() < p.num_rgbs) rgb.resize(p.num_rgbs, V4{0, 0, 0, 0});
    }

    float number_value(const PyrOperand& o, ProgramInput& in) { // get_number_value :286-296
        if (o.kind == PYR_OPERAND_CONSTANT) {
            float f;
            std::memcpy(&f, &o.bits, 4);
            return f;
        }
        if (o.kind == PYR_OPERAND_INPUT) {
            in.wavelength_used = true; // materials/mod.rs:263-268 (only observed for probability programs)
            return in.wavelength;
        }
        return number[o.bits];
    }
    V4 vector_value(uint32_t input, const ProgramInput& in) { // get_vector_value :298-302; Vector::from(Vector3) extends with 0
        if (input == PYR_INPUT_NORMAL) return V4{in.normal.x, in.normal.y, in.normal.z, 0.0f};
        if (input == PYR_INPUT_INCIDENT) return V4{in.incident.x, in.incident.y, in.incident.z, 0.0f};
        return V4{in.texture[0], in.texture[1], 0.0f, 0.0f};
    }

    static float binop(uint32_t op, float l, float r) {
        switch (op) {
        case PYR_BIN_ADD: return l + r;
        case PYR_BIN_SUB: return l - r;
        case PYR_BIN_MUL: return l * r;
        default: return l / r;
        }
    }

    // run_instructions, execution_context.rs:69-283. `changes` is the Inputs mask.
    void run_instructions(const PyrProgram& p, ProgramInput& in, uint32_t changes) {
        const OracleScene& s = *scene;
        for (uint32_t k = 0; k < p.num_instrs; ++k) {
            const PyrInstr& ins = s.instrs[p.first_instr + k];
            if (ins.deps != 0 && (ins.deps & changes) == 0) continue; // :76-78
            switch (ins.op) {
            case PYR_OP_NUMBER: {
                float f;
                std::memcpy(&f, &ins.x.bits, 4);
                number[ins.output] = f;
                break;
            }
            case PYR_OP_VECTOR: {
                float x = number_value(ins.x, in), y = number_value(ins.y, in), z = number_value(ins.z, in), w = number_value(ins.w, in);
                vector[ins.output] = V4{x, y, z, w};
                break;
            }
            case PYR_OP_RGB: {
                float r = number_value(ins.x, in), g = number_value(ins.y, in), b = number_value(ins.z, in);
                rgb[ins.output] = V4{r, g, b, 1.0f};
                break;
            }
            case PYR_OP_SPECTRUM: {
                float wl = number_value(ins.x, in);
                const PyrSpectrum& sp = s.spectra[ins.a];
                number[ins.output] = spectrum_get(sp.format, sp.min, sp.max, s.spectrum_data.data() + sp.offset, sp.count, wl);
                break;
            }
            case PYR_OP_RGB_SPECTRUM: { // :140-152; RGB basis is Spectrum::Array<LinSrgb> (build.rs:18-59)
                float wl = number_value(ins.x, in);
                V4 c = rgb[ins.a];
                uint32_t count = (uint32_t)(s.rgb_basis.size() / 3);
                float resp[3] = {0, 0, 0};
                if (count > 0) {
                    const float* d = s.rgb_basis.data();
                    float mn = s.rgb_basis_min, mx = s.rgb_basis_max;
                    if (wl <= mn) {
                        for (int j = 0; j < 3; ++j) resp[j] = d[j];
                    } else if (wl >= mx) {
                        for (int j = 0; j < 3; ++j) resp[j] = d[3 * (count - 1) + j];
                    } else {
                        float normalized = (wl - mn) / (mx - mn);
                        float fi = normalized * ((float)count - 1.0f);
                        float fmin = std::trunc(fi);
                        uint32_t i0 = (uint32_t)fmin, i1 = i0 + 1;
                        float mix = fi - fmin;
                        for (int j = 0; j < 3; ++j) resp[j] = d[3 * i0 + j] * (1.0f - mix) + d[3 * i1 + j] * mix;
                    }
                }
                float rr = c.x * resp[0], gg = c.y * resp[1], bb = c.z * resp[2];
                number[ins.output] = rr + gg + bb;
                break;
            }
            case PYR_OP_FRESNEL: {
                float ior = number_value(ins.x, in);
                float env = number_value(ins.y, in);
                V4 nn = vector_value(ins.a, in), ii = vector_value(ins.b, in);
                number[ins.output] = fresnel(ior, env, v3(nn.x, nn.y, nn.z), v3(ii.x, ii.y, ii.z));
                break;
            }
            case PYR_OP_BLACKBODY: {
                float wl = number_value(ins.x, in);
                float temp = number_value(ins.y, in);
                number[ins.output] = blackbody(wl, temp);
                break;
            }
            case PYR_OP_RGB_TO_VECTOR: {
                V4 c = rgb[ins.a];
                vector[ins.output] = V4{(c.x * 2.0f) - 1.0f, (c.y * 2.0f) - 1.0f, (c.z * 2.0f) - 1.0f, (c.w * 2.0f) - 1.0f};
                break;
            }
            case PYR_OP_MIX: {
                float amount = number_value(ins.x, in);
                amount = rmax(rmin(amount, 1.0f), 0.0f);
                if (ins.value_type == PYR_VT_NUMBER) {
                    float l = number[ins.a], r = number[ins.b];
                    number[ins.output] = l * (1.0f - amount) + r * amount;
                } else {
                    // [3P] cgmath lerp / palette Mix: self + (other - self) * amount, component-wise (alpha included)
                    std::vector<V4>& file = ins.value_type == PYR_VT_VECTOR ? vector : rgb;
                    V4 l = file[ins.a], r = file[ins.b];
                    file[ins.output] = V4{l.x + (r.x - l.x) * amount, l.y + (r.y - l.y) * amount, l.z + (r.z - l.z) * amount,
                                          l.w + (r.w - l.w) * amount};
                }
                break;
            }
            case PYR_OP_BINARY: {
                if (ins.value_type == PYR_VT_NUMBER) {
                    number[ins.output] = binop(ins.operator_, number[ins.a], number[ins.b]);
                } else {
                    std::vector<V4>& file = ins.value_type == PYR_VT_VECTOR ? vector : rgb;
                    V4 l = file[ins.a], r = file[ins.b];
                    file[ins.output] = V4{binop(ins.operator_, l.x, r.x), binop(ins.operator_, l.y, r.y),
                                          binop(ins.operator_, l.z, r.z), binop(ins.operator_, l.w, r.w)};
                }
                break;
            }
            case PYR_OP_CLAMP: {
                float value = number_value(ins.x, in), mn = number_value(ins.y, in), mx = number_value(ins.z, in);
                number[ins.output] = rmax(rmin(value, mx), mn);
                break;
            }
            case PYR_OP_COLOR_TEXTURE: { // :114-126
                V4 pos = vector_value(ins.b, in);
                float c[4];
                texture_get_color(s.textures[ins.a], pos.x, pos.y, c);
                rgb[ins.output] = V4{c[0], c[1], c[2], c[3]};
                break;
            }
            case PYR_OP_MONO_TEXTURE: { // :127-139
                V4 pos = vector_value(ins.b, in);
                float c[1];
                texture_get_color(s.textures[ins.a], pos.x, pos.y, c);
                number[ins.output] = c[0];
                break;
            }
            default: break;
            }
        }
    }

    float output(const PyrProgram& p) {
        if (p.output_kind == PYR_OUTPUT_NUMBER) return number[p.output_reg];
        return vector[p.output_reg].x; // f32 programs never read a vector register (compiler.rs:561-563)
    }

    // ExecutionContext::run for a Vector program (normal maps): a constant program is a number broadcast (compiler.rs).
    V4 run_vector(uint32_t program, ProgramInput& in) {
        const PyrProgram& p = scene->programs[program];
        if (p.kind == PYR_PROGRAM_CONSTANT) return V4{p.constant, p.constant, p.constant, p.constant};
        reserve(p);
        run_instructions(p, in, 0xFFu);
        if (p.output_kind == PYR_OUTPUT_VECTOR) return vector[p.output_reg];
        float n = number[p.output_reg];
        return V4{n, n, n, n};
    }

    // ExecutionContext::run, execution_context.rs:29-56.
    float run(uint32_t program, ProgramInput& in) {
        const PyrProgram& p = scene->programs[program];
        if (p.kind == PYR_PROGRAM_CONSTANT) return p.constant;
        reserve(p);
        run_instructions(p, in, 0xFFu);
        return output(p);
    }
};

// MemoizedProgram (program/memoized.rs:9-40) + MemoizedContext::run (execution_context.rs:311-342).
struct Memoized {
    Exe& exe;
    const PyrProgram& program;
    ProgramInput input;
    uint32_t changes = 0xFFu;
    bool fresh = true;
    Memoized(Exe& e, uint32_t program_id, const ProgramInput& initial) : exe(e), program(e.scene->programs[program_id]), input(initial) {}
    void set_wavelength(float w) { // RenderContextUpdater::set_wavelength, tracer.rs:117-122
        input.wavelength = w;
        changes |= PYR_DEP_WAVELENGTH;
    }
    float run() {
        float result;
        if (program.kind == PYR_PROGRAM_CONSTANT) {
            result = program.constant;
        } else {
            uint32_t ch = changes;
            if (fresh) {
                exe.reserve(program);
                fresh = false;
                ch = 0xFFu;
            }
            exe.run_instructions(program, input, ch);
            result = exe.output(program);
        }
        changes = 0;
        return result;
    }
};

// ------------------------------------------------------------------------------------------------
// materials/
// ------------------------------------------------------------------------------------------------
struct Scattering { // materials/mod.rs:361-369; brdf: Some(lambertian) only for diffuse
    bool emitted;
    V3 out_direction;
    float probability;
    bool dispersed;
    bool has_brdf;
};

// materials/diffuse.rs:27-29 lambertian(_ray_in, ray_out, normal).
inline float lambertian(V3 /*ray_in*/, V3 ray_out, V3 normal) { return 2.0f * std::fabs(dot(normal, ray_out)); }

// materials/refractive.rs:47-91 refract.
inline void refract(float ior, float env_ior, V3 in_direction, V3 normal, Rng& rng, V3& out, float& prob) {
    V3 nl = dot(normal, in_direction) < 0.0f ? normal : -normal;
    V3 reflected = in_direction - (normal * 2.0f * dot(normal, in_direction));
    bool into = dot(normal, nl) > 0.0f;
    float nnt = into ? env_ior / ior : ior / env_ior;
    float ddn = dot(in_direction, nl);
    float cos2t = 1.0f - nnt * nnt * (1.0f - ddn * ddn);
    if (cos2t < 0.0f) {
        out = reflected;
        prob = 1.0f;
        return;
    }
    float s = (into ? 1.0f : -1.0f) * (ddn * nnt + std::sqrt(cos2t));
    V3 tdir = normalize(in_direction * nnt - normal * s);
    float a = ior - env_ior;
    float b = ior + env_ior;
    float r0 = a * a / (b * b);
    float c = 1.0f - (into ? -ddn : dot(tdir, normal));
    float re = r0 + (1.0f - r0) * c * c * c * c * c;
    float tr = 1.0f - re;
    float p = 0.25f + 0.5f * re;
    float rp = re / p;
    float tp = tr / (1.0f - p);
    if (gen_f32(rng) < p) {
        out = reflected;
        prob = rp;
    } else {
        out = tdir;
        prob = tp;
    }
}

// SurfaceBsdfType::scatter, materials/mod.rs:344-359 + diffuse.rs:8-25, mirror.rs:5-21, refractive.rs:6-37.
inline Scattering scatter(const PyrComponent& comp, V3 in_direction, V3 normal, float wavelength, Rng& rng) {
    Scattering sc{};
    switch (comp.bsdf) {
    case PYR_BSDF_EMISSIVE: sc.emitted = true; return sc;
    case PYR_BSDF_DIFFUSE: {
        V3 n = dot(in_direction, normal) < 0.0f ? normal : -normal;
        sc.out_direction = sample_hemisphere(rng, n);
        sc.probability = 1.0f;
        sc.dispersed = false;
        sc.has_brdf = true;
        return sc;
    }
    case PYR_BSDF_MIRROR: {
        V3 n = dot(in_direction, normal) < 0.0f ? normal : -normal;
        float perp = dot(in_direction, n) * 2.0f;
        n = n * perp;
        sc.out_direction = in_direction - n;
        sc.probability = 1.0f;
        sc.dispersed = false;
        sc.has_brdf = false;
        return sc;
    }
    default: {
        bool dispersed = comp.dispersion != 0.0f || comp.env_dispersion != 0.0f;
        if (dispersed) {
            float wl = wavelength * 0.001f;
            float ior = comp.ior + comp.dispersion / (wl * wl);
            float env_ior = comp.env_ior + comp.env_dispersion / (wl * wl);
            refract(ior, env_ior, in_direction, normal, rng, sc.out_direction, sc.probability);
        } else {
            refract(comp.ior, comp.env_ior, in_direction, normal, rng, sc.out_direction, sc.probability);
        }
        sc.dispersed = dispersed;
        sc.has_brdf = false;
        return sc;
    }
    }
}

// MaterialComponent::get_probability, materials/mod.rs:238-248.
inline float get_probability(const PyrComponent& comp, Exe& exe, ProgramInput& input) {
    if (comp.probability_program >= 0) return exe.run((uint32_t)comp.probability_program, input) * comp.selection_compensation;
    return comp.selection_compensation;
}

// ------------------------------------------------------------------------------------------------
// lamp.rs / shapes sampling helpers
// ------------------------------------------------------------------------------------------------
struct LampSample { // lamp.rs:116-130
    V3 direction;
    bool has_sq_distance;
    float sq_distance;
    bool physical;
    V3 normal;         // Physical
    float texture[2];  // Physical
    uint32_t material; // Physical
    uint32_t color;    // Color
    float weight;
};

// Shape::surface_area, shapes/mod.rs:273-288.
inline float surface_area(const OracleScene& s, ShapeRef r) {
    if (r.kind == PYR_SHAPE_SPHERE) {
        float radius = s.spheres[r.index].radius;
        return radius * radius * 4.0f * PI;
    }
    const Triangle& t = s.triangles[r.index];
    V3 a = t.p2 - t.p1, b = t.p3 - t.p1;
    return 0.5f * magnitude(cross(a, b));
}

// Shape::sample_point, shapes/mod.rs:166-207.
inline Intersection sample_point(const OracleScene& s, ShapeRef r, Rng& rng) {
    Intersection it{};
    it.shape = r;
    if (r.kind == PYR_SHAPE_SPHERE) {
        const Sphere& sp = s.spheres[r.index];
        V3 sphere_point = sample_sphere(rng);
        it.position = sp.position + sphere_point * sp.radius;
    } else {
        const Triangle& t = s.triangles[r.index];
        float u = gen_f32(rng);
        float v = gen_f32(rng);
        V3 a = t.p2 - t.p1, b = t.p3 - t.p1;
        if (u + v > 1.0f) {
            u = 1.0f - u;
            v = 1.0f - v;
        }
        it.position = t.p1 + a * u + b * v;
        it.u = u;
        it.v = v;
    }
    return it;
}

// Shape::sample_towards, shapes/mod.rs:209-251.
inline Intersection sample_towards(const OracleScene& s, ShapeRef r, Rng& rng, V3 target, Counters& c) {
    if (r.kind == PYR_SHAPE_SPHERE) {
        const Sphere& sp = s.spheres[r.index];
        float radius = rmax(sp.radius - DIST_EPSILON, 0.0f);
        V3 dir = sp.position - target;
        float dist2 = magnitude2(dir);
        if (dist2 > radius * radius) {
            float cos_theta_max = std::sqrt(rmax(1.0f - (radius * radius) / dist2, 0.0f));
            V3 ray_dir = sample_cone(rng, normalize(dir), cos_theta_max);
            Intersection it;
            if (shape_intersect(s, r, Ray{target, ray_dir}, it, c)) return it;
            it.distance = 0.0f; // "cheat", :229-236
            it.position = target;
            it.shape = r;
            it.u = it.v = 0.0f;
            return it;
        }
    }
    Intersection it = sample_point(s, r, rng);
    it.distance = magnitude(it.position - target);
    return it;
}

// Shape::solid_angle_towards, shapes/mod.rs:253-271.
inline bool solid_angle_towards(const OracleScene& s, ShapeRef r, V3 target, float& out) {
    if (r.kind != PYR_SHAPE_SPHERE) return false;
    const Sphere& sp = s.spheres[r.index];
    float dist2 = magnitude2(sp.position - target);
    if (dist2 > sp.radius * sp.radius) {
        float cos_theta_max = std::sqrt(rmax(1.0f - (sp.radius * sp.radius) / dist2, 0.0f));
        out = solid_angle(cos_theta_max);
        return true;
    }
    return false;
}

// Lamp::sample, lamp.rs:23-82.
inline LampSample lamp_sample(const OracleScene& s, const PyrLamp& lamp, Rng& rng, V3 target, Counters& c) {
    LampSample ls{};
    if (lamp.kind == PYR_LAMP_DIRECTIONAL) {
        V3 direction = v3(lamp.v[0], lamp.v[1], lamp.v[2]);
        ls.direction = lamp.width > 0.0f ? sample_cone(rng, direction, lamp.width) : direction;
        ls.has_sq_distance = false;
        ls.physical = false;
        ls.color = lamp.color_program;
        ls.weight = 1.0f;
    } else if (lamp.kind == PYR_LAMP_POINT) {
        V3 v = v3(lamp.v[0], lamp.v[1], lamp.v[2]) - target;
        float distance = magnitude2(v);
        ls.direction = normalize(v);
        ls.has_sq_distance = true;
        ls.sq_distance = distance;
        ls.physical = false;
        ls.color = lamp.color_program;
        ls.weight = 4.0f * PI / distance;
    } else {
        ShapeRef shape{lamp.shape_kind, lamp.shape_index};
        Intersection it = sample_towards(s, shape, rng, target, c);
        V3 v = it.position - target;
        float sq_distance = it.distance * it.distance;
        V3 direction = normalize(v);
        SurfaceData sd = surface_data(s, it);
        float weight;
        if (!solid_angle_towards(s, shape, target, weight)) {
            float cos_in = std::fabs(dot(sd.normal, -direction));
            weight = cos_in * surface_area(s, shape) / sq_distance;
        }
        ls.direction = direction;
        ls.has_sq_distance = true;
        ls.sq_distance = sq_distance;
        ls.physical = true;
        ls.normal = sd.normal;
        ls.texture[0] = sd.texture[0];
        ls.texture[1] = sd.texture[1];
        ls.material = shape_material(s, shape);
        ls.weight = weight;
    }
    return ls;
}

// ------------------------------------------------------------------------------------------------
// tracer.rs
// ------------------------------------------------------------------------------------------------
struct DirectLight { // tracer.rs:193-200
    bool dispersed;
    uint32_t color;
    V3 incident, normal;
    float texture[2];
    float probability;
};

enum BounceKind { BOUNCE_DIFFUSE, BOUNCE_SPECULAR, BOUNCE_EMISSION }; // BounceType, tracer.rs:169-173

struct Bounce { // tracer.rs:157-167
    BounceKind ty;
    V3 out; // Diffuse(brdf, out)
    bool dispersed;
    uint32_t color;
    V3 incident, position, normal;
    float texture[2];
    float probability;
    // Vec<DirectLight> in the reference (tracer.rs:166). The light samples of all bounces of a path sit in one array the
    // tile's loop reuses from sample to sample (no allocation per bounce); a bounce names its slice.
    uint32_t light_first = 0, light_count = 0;
};

static thread_local bool g_debug_shadow_rays = false; // developer aid (ORACLE_DEBUG_PIXEL): set around the traced sample

// trace_direct, tracer.rs:347-442. The reference panics in pick_lamp's gen_range(0..0) when the world has no
// lamps; this restatement returns no direct light (and draws nothing) in that case.
void trace_direct(const OracleScene& s, Rng& rng, uint32_t samples, float wavelength, V3 ray_in, V3 position, V3 normal, Exe& exe,
                  std::vector<DirectLight>& out, Counters& c) {
    if (s.lamps.empty()) return;
    // World::pick_lamp, world.rs:301-305
    uint32_t lamp_index = gen_range_usize(rng, (uint32_t)s.lamps.size());
    const PyrLamp& lamp = s.lamps[lamp_index];
    float lamp_probability = 1.0f / (float)s.lamps.size();

    if (!(dot(ray_in, normal) < 0.0f)) normal = -normal;
    float probability = 1.0f / ((float)samples * 2.0f * PI * lamp_probability);

    for (uint32_t k = 0; k < samples; ++k) {
        LampSample ls = lamp_sample(s, lamp, rng, position, c);
        Ray ray_out{position, ls.direction};
        float cos_out = rmax(dot(normal, ray_out.direction), 0.0f);
        if (cos_out > 0.0f) {
            Intersection hit;
            c.shadow_rays++;
            bool has_hit = world_intersect(s, ray_out, hit, c);
            float hit_dist = has_hit ? hit.distance * hit.distance : 0.0f;
            bool blocked;
            if (has_hit && ls.has_sq_distance && hit_dist >= ls.sq_distance - DIST_EPSILON)
                blocked = false;
            else if (!has_hit)
                blocked = false;
            else
                blocked = true;
            if (g_debug_shadow_rays)
                std::fprintf(stderr, "[shadow] origin (%.9g %.9g %.9g) direction (%.9g %.9g %.9g) hit %d distance %.9g limit %.9g blocked %d\n", position.x, position.y, position.z,
                             ls.direction.x, ls.direction.y, ls.direction.z, (int)has_hit, has_hit ? hit.distance : 0.0f, ls.has_sq_distance ? ls.sq_distance - DIST_EPSILON : -1.0f, (int)blocked);
            if (!blocked) {
                DirectLight dl{};
                float material_probability;
                if (ls.physical) {
                    const PyrMaterial& m = s.materials[ls.material];
                    // Material::choose_emissive, materials/mod.rs:56-62
                    uint32_t pick = choose_index(rng, m.num_emissive);
                    const PyrComponent& comp = s.components[m.first_emissive + pick];
                    ProgramInput input{wavelength, ls.normal, ray_out.direction, {ls.texture[0], ls.texture[1]}};
                    material_probability = get_probability(comp, exe, input);
                    dl.color = comp.color_program;
                    dl.dispersed = input.wavelength_used;
                    dl.normal = ls.normal;
                    dl.texture[0] = ls.texture[0];
                    dl.texture[1] = ls.texture[1];
                } else {
                    dl.color = ls.color;
                    material_probability = 1.0f;
                    dl.dispersed = false;
                    dl.normal = -ray_out.direction;
                    dl.texture[0] = dl.texture[1] = 0.0f;
                }
                float scale = ls.weight * probability * lambertian(ray_in, normal, ray_out.direction);
                dl.incident = ray_out.direction;
                dl.probability = scale * material_probability;
                out.push_back(dl);
            }
        }
    }
}

// trace_directional, tracer.rs:444-459.
inline bool trace_directional(const OracleScene& s, V3 ray, uint32_t& color) {
    for (const PyrLamp& l : s.lamps) {
        if (l.kind == PYR_LAMP_DIRECTIONAL) {
            if (dot(v3(l.v[0], l.v[1], l.v[2]), ray) >= l.width) {
                color = l.color_program;
                return true;
            }
        }
    }
    return false;
}

// trace, tracer.rs:208-345.
void trace(const OracleScene& s, std::vector<Bounce>& path, std::vector<DirectLight>& lights, Rng& rng, Ray ray, float wavelength, uint32_t bounces,
           uint32_t light_samples, Exe& exe, Counters& c) {
    bool sample_light = true;
    uint32_t light_sample_events = 0;
    for (uint32_t b = 0; b < bounces; ++b) {
        Intersection hit;
        c.extension_rays++;
        if (world_intersect(s, ray, hit, c)) {
            c.shaded_hits++;
            const PyrMaterial& material = s.materials[shape_material(s, hit.shape)];
            SurfaceData sd = surface_data(s, hit);
            V3 normal = sd.normal;
            if (material.normal_map_program >= 0) { // Material::apply_normal_map, materials/mod.rs:68-80 (tracer.rs:227-232)
                ProgramInput normal_input{wavelength, sd.normal, ray.direction, {sd.texture[0], sd.texture[1]}};
                V4 n = exe.run_vector((uint32_t)material.normal_map_program, normal_input);
                normal = normalize(quat_rotate(sd.from_space, v3(n.x, n.y, n.z)));
            }
            V3 position = hit.position;
            // Material::choose_component, materials/mod.rs:48-54
            uint32_t pick = choose_index(rng, material.num_components);
            const PyrComponent& component = s.components[material.first_component + pick];
            ProgramInput probability_input{wavelength, normal, ray.direction, {sd.texture[0], sd.texture[1]}};
            float component_probability = get_probability(component, exe, probability_input);
            bool normal_dispersed = probability_input.wavelength_used;
            Scattering sc = scatter(component, ray.direction, normal, wavelength, rng);
            if (!sc.emitted) {
                Bounce bounce;
                if (light_sample_events < 2) {
                    sample_light = !sc.has_brdf || light_samples == 0;
                    if (sc.has_brdf) {
                        light_sample_events += 1;
                        bounce.light_first = (uint32_t)lights.size();
                        trace_direct(s, rng, light_samples, wavelength, ray.direction, position, normal, exe, lights, c);
                        bounce.light_count = (uint32_t)lights.size() - bounce.light_first;
                    }
                } else {
                    sample_light = true;
                }
                bounce.ty = sc.has_brdf ? BOUNCE_DIFFUSE : BOUNCE_SPECULAR;
                bounce.out = sc.out_direction;
                bounce.dispersed = sc.dispersed || normal_dispersed;
                bounce.color = component.color_program;
                bounce.incident = ray.direction;
                bounce.position = position;
                bounce.normal = normal;
                bounce.texture[0] = sd.texture[0];
                bounce.texture[1] = sd.texture[1];
                bounce.probability = sc.probability * component_probability;
                ray = Ray{position, sc.out_direction};
                path.push_back(std::move(bounce));
            } else {
                if (sample_light) {
                    Bounce bounce;
                    bounce.ty = BOUNCE_EMISSION;
                    bounce.dispersed = normal_dispersed;
                    bounce.color = component.color_program;
                    bounce.incident = ray.direction;
                    bounce.position = position;
                    bounce.normal = normal;
                    bounce.texture[0] = sd.texture[0];
                    bounce.texture[1] = sd.texture[1];
                    bounce.probability = component_probability;
                    path.push_back(std::move(bounce));
                }
                break;
            }
        } else {
            uint32_t color = s.sky_program;
            if (sample_light) {
                uint32_t directional;
                if (trace_directional(s, ray.direction, directional)) color = directional;
            }
            Bounce bounce;
            bounce.ty = BOUNCE_EMISSION;
            bounce.dispersed = false;
            bounce.color = color;
            bounce.incident = ray.direction;
            bounce.position = ray.direction * INF;
            bounce.normal = -ray.direction;
            bounce.texture[0] = bounce.texture[1] = 0.0f;
            bounce.probability = 1.0f;
            path.push_back(std::move(bounce));
            break;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// renderer/algorithm.rs contribute (:14-100)
// ------------------------------------------------------------------------------------------------
struct SpectralSample { // (Sample {brightness, wavelength, weight}, reflectance), simple.rs:91-103
    float brightness, wavelength, weight, reflectance;
};

void contribute(const Bounce& bounce, const DirectLight* lights, SpectralSample& main_sample, SpectralSample* additional, size_t n_additional, Exe& exe) {
    if (bounce.ty == BOUNCE_EMISSION) {
        ProgramInput initial{main_sample.wavelength, bounce.normal, bounce.incident, {bounce.texture[0], bounce.texture[1]}};
        Memoized m(exe, bounce.color, initial);
        main_sample.brightness += m.run() * bounce.probability * main_sample.reflectance;
        for (size_t i = 0; i < n_additional; ++i) {
            m.set_wavelength(additional[i].wavelength);
            additional[i].brightness += m.run() * bounce.probability * additional[i].reflectance;
        }
    } else {
        {
            ProgramInput initial{main_sample.wavelength, bounce.normal, bounce.incident, {bounce.texture[0], bounce.texture[1]}};
            Memoized m(exe, bounce.color, initial);
            main_sample.reflectance *= m.run() * bounce.probability;
            for (size_t i = 0; i < n_additional; ++i) {
                m.set_wavelength(additional[i].wavelength);
                additional[i].reflectance *= m.run() * bounce.probability;
            }
        }
        for (uint32_t li = 0; li < bounce.light_count; ++li) {
            const DirectLight& direct = lights[bounce.light_first + li];
            ProgramInput initial{main_sample.wavelength, direct.normal, direct.incident, {direct.texture[0], direct.texture[1]}};
            Memoized m(exe, direct.color, initial);
            main_sample.brightness += m.run() * direct.probability * main_sample.reflectance;
            if (!direct.dispersed) {
                for (size_t i = 0; i < n_additional; ++i) {
                    m.set_wavelength(additional[i].wavelength);
                    additional[i].brightness += m.run() * direct.probability * additional[i].reflectance;
                }
            }
        }
        // BounceType::brdf, tracer.rs:175-183
        float brdf = bounce.ty == BOUNCE_DIFFUSE ? lambertian(bounce.incident, bounce.normal, bounce.out) : 1.0f;
        main_sample.reflectance *= brdf;
        for (size_t i = 0; i < n_additional; ++i) additional[i].reflectance *= brdf;
    }
}

// ------------------------------------------------------------------------------------------------
// cameras.rs / film.rs / tiles
// ------------------------------------------------------------------------------------------------
struct Area {
    float from_x, from_y, size_x, size_y;
};

// Camera::to_view_area, cameras.rs:57-68.
inline Area to_view_area(uint32_t x, uint32_t y, uint32_t w, uint32_t h, uint32_t width, uint32_t height) {
    float iw = (float)width, ih = (float)height;
    float cx = (float)x, cy = (float)y;
    float sx = (float)w, sy = (float)h;
    float max_dimension = rmax(iw, ih);
    Area a;
    a.from_x = (cx + (-iw * 0.5f)) / (max_dimension * 0.5f);
    a.from_y = (cy + (-ih * 0.5f)) / (max_dimension * 0.5f);
    a.size_x = sx / (max_dimension * 0.5f);
    a.size_y = sy / (max_dimension * 0.5f);
    return a;
}

// [3P] cgmath Matrix4 * Vector4 = c0*x + c1*y + c2*z + c3*w; transform_point divides by w (from_homogeneous).
inline V3 transform_point(const float m[16], V3 p) {
    float x = m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12] * 1.0f;
    float y = m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13] * 1.0f;
    float z = m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14] * 1.0f;
    float w = m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15] * 1.0f;
    float inv = 1.0f / w;
    return v3(x * inv, y * inv, z * inv);
}
inline V3 transform_vector(const float m[16], V3 v) {
    return v3(m[0] * v.x + m[4] * v.y + m[8] * v.z + m[12] * 0.0f, m[1] * v.x + m[5] * v.y + m[9] * v.z + m[13] * 0.0f,
              m[2] * v.x + m[6] * v.y + m[10] * v.z + m[14] * 0.0f);
}

// Camera::ray_towards, cameras.rs:70-97.
inline Ray ray_towards(const PyrCamera& cam, float tx, float ty, Rng& rng) {
    float focus_x = tx / cam.view_plane * cam.focus_distance;
    float focus_y = ty / cam.view_plane * cam.focus_distance;
    V3 target = v3(focus_x, -focus_y, -cam.focus_distance);
    V3 origin, direction;
    if (cam.aperture > 0.0f) {
        float sqrt_r = std::sqrt(cam.aperture * gen_f32(rng));
        float psi = PI * 2.0f * gen_f32(rng);
        float lens_x = sqrt_r * cos32(psi);
        float lens_y = sqrt_r * sin32(psi);
        origin = v3(lens_x, lens_y, 0.0f);
        direction = target - origin;
    } else {
        origin = v3(0, 0, 0);
        direction = target;
    }
    V3 d = normalize(direction);
    return Ray{transform_point(cam.cam_to_world, origin), transform_vector(cam.cam_to_world, d)};
}

// Rust `f32 as usize`: saturating, NaN -> 0.
inline uint64_t f32_as_usize(float f) {
    if (!(f > 0.0f)) return 0;
    if (f >= 18446744073709551616.0f) return UINT64_MAX;
    return (uint64_t)f;
}

// AspectRatio::to_pixel, film.rs:203-246.
inline bool to_pixel(uint32_t width, uint32_t height, float x, float y, uint64_t& px, uint64_t& py) {
    if (width >= height) {
        float size = (float)width, ratio = (float)height / (float)width;
        if (!(std::fabs(y) <= ratio)) return false;
        px = f32_as_usize(size * (x + 1.0f) * 0.5f);
        py = f32_as_usize(size * (y + ratio) * 0.5f);
    } else {
        float size = (float)height, ratio = (float)width / (float)height;
        if (!(std::fabs(x) <= ratio)) return false;
        px = f32_as_usize(size * (x + ratio) * 0.5f);
        py = f32_as_usize(size * (y + 1.0f) * 0.5f);
    }
    return true;
}

// Film::wavelength_to_grain, film.rs:85-87. The reference would panic on an index == bins; clamp instead.
inline uint32_t wavelength_to_grain(float wavelength, float start, float width, uint32_t bins) {
    float grains_per_wavelength = (float)bins / width; // film.rs:38
    uint64_t g = f32_as_usize((wavelength - start) * grains_per_wavelength);
    return (uint32_t)std::min<uint64_t>(g, bins - 1);
}

struct FilmView {
    PyrGrain* grains;
    PyrFilmDesc desc;
    uint32_t row_begin, row_count;
    // PYR_FILM_TILE_BLOCKS (pyrite_gpu.h): one (tile_size + 2)^2 block per rendered tile, the tile's pixels with a ring of one
    // pixel; block k belongs to tile tile_begin + k * tile_stride. Not a reference concept: the multi-GPU plan's buffer.
    uint32_t layout = PYR_FILM_ROWS, tile_size = 0, tiles_x = 0, tile_begin = 0, tile_stride = 1;
};

// Film::expose (film.rs:89-95) + Grain::increment (film.rs:145-162). The reference gives up after five failed
// compare-exchange attempts and drops the sample; the oracle retries until it succeeds (deterministic totals).
inline void film_expose(const FilmView& film, uint32_t tile, float px, float py, float wavelength, float brightness, float weight, Counters& c) {
    uint32_t grain = wavelength_to_grain(wavelength, film.desc.wl_start, film.desc.wl_width, film.desc.bins);
    uint64_t x, y;
    if (!to_pixel(film.desc.width, film.desc.height, px, py, x, y)) return;
    if (x >= film.desc.width || y >= film.desc.height) return; // Film::get_pixel, film.rs:51-54
    size_t index;
    if (film.layout == PYR_FILM_TILE_BLOCKS) {
        const uint64_t side = (uint64_t)film.tile_size + 2, tx = tile % film.tiles_x, ty = tile / film.tiles_x;
        const int64_t bx = (int64_t)x + 1 - (int64_t)(tx * film.tile_size), by = (int64_t)y + 1 - (int64_t)(ty * film.tile_size);
        if (bx < 0 || by < 0 || bx >= (int64_t)side || by >= (int64_t)side) return;
        const uint64_t block = (tile - film.tile_begin) / film.tile_stride;
        index = (size_t)(((block * side + (uint64_t)by) * side + (uint64_t)bx) * film.desc.bins + grain);
    } else {
        if (y < film.row_begin || y >= (uint64_t)film.row_begin + film.row_count) return;
        index = ((size_t)x + (size_t)(y - film.row_begin) * film.desc.width) * film.desc.bins + grain;
    }
    static_assert(sizeof(PyrGrain) == 8, "grain must be 8 bytes");
    auto* cell = reinterpret_cast<std::atomic<uint64_t>*>(&film.grains[index]);
    uint64_t current = cell->load(std::memory_order_relaxed);
    for (;;) {
        PyrGrain g;
        std::memcpy(&g, &current, 8);
        g.acc = g.acc + brightness * weight; // Grain::expose: increment(value*weight, weight), film.rs:128-130
        g.weight = g.weight + weight;
        uint64_t next;
        std::memcpy(&next, &g, 8);
        if (cell->compare_exchange_weak(current, next, std::memory_order_relaxed)) break;
    }
    c.exposures++;
}

struct Tile { // renderer/algorithm.rs:102-106
    Area area;
    uint32_t width, height;
    uint32_t raster_index;
};

// make_tiles, renderer/algorithm.rs:152-188. `tiles.sort()` is a stable sort by |center|^2 (Ord via partial_cmp, :131-147).
std::vector<Tile> make_tiles(uint32_t film_width, uint32_t film_height, uint32_t tile_size) {
    uint32_t tiles_x = film_width / tile_size;
    if (tiles_x * tile_size < film_width) tiles_x += 1;
    uint32_t tiles_y = film_height / tile_size;
    if (tiles_y * tile_size < film_height) tiles_y += 1;
    std::vector<Tile> tiles;
    for (uint32_t y = 0; y < tiles_y; ++y)
        for (uint32_t x = 0; x < tiles_x; ++x) {
            uint32_t sx = x * tile_size, sy = y * tile_size;
            uint32_t w = std::min(film_width - sx, tile_size), h = std::min(film_height - sy, tile_size);
            tiles.push_back(Tile{to_view_area(sx, sy, w, h, film_width, film_height), w, h, y * tiles_x + x});
        }
    auto key = [](const Tile& t) {
        float cx = t.area.from_x + t.area.size_x / 2.0f, cy = t.area.from_y + t.area.size_y / 2.0f; // Area::center, film.rs:262-267
        return cx * cx + cy * cy;
    };
    std::stable_sort(tiles.begin(), tiles.end(), [&](const Tile& a, const Tile& b) { return key(a) < key(b); });
    return tiles;
}

// render_tile, renderer/simple.rs:58-141.
void render_tile(const OracleScene& s, const Tile& tile, const FilmView& film, const PyrCamera& camera, const PyrRenderParams& p, Counters& c) {
    std::vector<SpectralSample> additional_samples;
    additional_samples.reserve(p.spectrum_samples);
    std::vector<Bounce> path;
    path.reserve(p.bounces);
    std::vector<DirectLight> lights;
    lights.reserve(2 * (size_t)p.light_samples);
    Exe exe(&s);

    static const char* const dbg = std::getenv("ORACLE_DEBUG_PIXEL"); // read once, not per sample
    uint64_t iterations = (uint64_t)tile.width * tile.height * (uint64_t)p.pixel_samples;
    for (uint64_t i = 0; i < iterations; ++i) {
        Rng rng = rng_seed(p.seed, tile.raster_index, i);
        additional_samples.clear();
        path.clear();
        lights.clear();
        c.samples++;

        // Tile::sample_point, algorithm.rs:113-119
        float ox = tile.area.size_x * gen_f32(rng);
        float oy = tile.area.size_y * gen_f32(rng);
        float px = tile.area.from_x + ox, py = tile.area.from_y + oy;

        Ray ray = ray_towards(camera, px, py, rng);

        // Film::sample_many_wavelengths, film.rs:68-83
        {
            float step_size = film.desc.wl_width / (float)p.spectrum_samples;
            float from = film.desc.wl_start;
            for (uint32_t k = 0; k < p.spectrum_samples; ++k) {
                float to = from + step_size;
                float wavelength = gen_range_f32(rng, from, to);
                from = to;
                additional_samples.push_back(SpectralSample{0.0f, wavelength, 1.0f, 1.0f});
            }
        }
        // swap_remove(gen_range(0..len)), simple.rs:105-106
        uint32_t hero = gen_range_usize(rng, (uint32_t)additional_samples.size());
        SpectralSample main_sample = additional_samples[hero];
        additional_samples[hero] = additional_samples.back();
        additional_samples.pop_back();
        float wavelength = main_sample.wavelength;

        const Ray camera_ray = ray;
        if (dbg) { // the sample's shadow rays are printed as they are traced (trace_direct), blocked ones too
            unsigned dx = 0, dy = 0;
            uint64_t qx, qy;
            g_debug_shadow_rays = std::strcmp(dbg, "all") == 0 || (std::sscanf(dbg, "%u,%u", &dx, &dy) == 2 && to_pixel(film.desc.width, film.desc.height, px, py, qx, qy) && qx == dx && qy == dy);
        }
        trace(s, path, lights, rng, ray, wavelength, p.bounces, p.light_samples, exe, c);
        g_debug_shadow_rays = false;

        bool use_additional = true;
        for (const Bounce& bounce : path) {
            use_additional = !bounce.dispersed && use_additional;
            contribute(bounce, lights.data(), main_sample, additional_samples.data(), use_additional ? additional_samples.size() : 0, exe);
        }

        if (dbg) { // developer aid: ORACLE_DEBUG_PIXEL=x,y prints the samples of one pixel
            unsigned dx = 0, dy = 0;
            uint64_t qx, qy;
            if (std::strcmp(dbg, "all") == 0 || (std::sscanf(dbg, "%u,%u", &dx, &dy) == 2 && to_pixel(film.desc.width, film.desc.height, px, py, qx, qy) && qx == dx && qy == dy)) { // ORACLE_DEBUG_PIXEL=all: every sample (small images)
                std::fprintf(stderr, "[oracle] tile %u iteration %llu hero %.9g nm brightness %.9g use_additional %d bounces %zu\n", tile.raster_index,
                             (unsigned long long)i, main_sample.wavelength, main_sample.brightness, (int)use_additional, path.size());
                std::fprintf(stderr, "    camera ray origin (%.9g %.9g %.9g) direction (%.9g %.9g %.9g)\n", camera_ray.origin.x, camera_ray.origin.y, camera_ray.origin.z,
                             camera_ray.direction.x, camera_ray.direction.y, camera_ray.direction.z);
                for (const Bounce& b : path) {
                    std::fprintf(stderr, "    bounce type %d color %u prob %.9g dispersed %d pos (%.9g %.9g %.9g) normal (%.9g %.9g %.9g) out (%.9g %.9g %.9g) incident (%.9g %.9g %.9g) lights %zu:", (int)b.ty,
                                 b.color, b.probability, (int)b.dispersed, b.position.x, b.position.y, b.position.z, b.normal.x, b.normal.y, b.normal.z, b.out.x, b.out.y,
                                 b.out.z, b.incident.x, b.incident.y, b.incident.z, (size_t)b.light_count);
                    for (uint32_t li = 0; li < b.light_count; ++li) {
                        const DirectLight& dl = lights[b.light_first + li];
                        std::fprintf(stderr, " [color %u prob %.9g dir (%.9g %.9g %.9g)]", dl.color, dl.probability, dl.incident.x, dl.incident.y, dl.incident.z);
                    }
                    std::fprintf(stderr, "\n");
                }
            }
        }
        film_expose(film, tile.raster_index, px, py, main_sample.wavelength, main_sample.brightness, main_sample.weight, c);
        if (use_additional)
            for (const SpectralSample& sm : additional_samples) film_expose(film, tile.raster_index, px, py, sm.wavelength, sm.brightness, sm.weight, c);
    }
}

int validate_desc(const PyrSceneDesc* d) {
    if (!d) return fail(PYR_ERR_INVALID_ARGUMENT, "null scene description");
    if (d->sky_program >= d->num_programs) return fail(PYR_ERR_INVALID_ARGUMENT, "sky program out of range");
    for (uint32_t i = 0; i < d->num_instrs; ++i) {
        const PyrInstr& ins = d->instrs[i];
        if (ins.op == PYR_OP_COLOR_TEXTURE || ins.op == PYR_OP_MONO_TEXTURE) {
            if (ins.a >= d->num_textures) return fail(PYR_ERR_INVALID_ARGUMENT, "texture id out of range");
            if (d->textures[ins.a].format != (ins.op == PYR_OP_COLOR_TEXTURE ? PYR_TEXTURE_COLOR : PYR_TEXTURE_MONO))
                return fail(PYR_ERR_INVALID_ARGUMENT, "texture format does not match the opcode");
        }
    }
    for (uint32_t i = 0; i < d->num_textures; ++i) {
        const PyrTexture& t = d->textures[i];
        uint64_t floats = (uint64_t)t.width * t.height * (t.format == PYR_TEXTURE_COLOR ? 4u : 1u);
        if (t.width == 0 || t.height == 0 || t.format > PYR_TEXTURE_MONO || t.offset + floats > d->num_texture_floats)
            return fail(PYR_ERR_INVALID_ARGUMENT, "texture data out of range");
    }
    for (uint32_t i = 0; i < d->num_materials; ++i) {
        const PyrMaterial& m = d->materials[i];
        if (m.normal_map_program >= 0 && (uint32_t)m.normal_map_program >= d->num_programs)
            return fail(PYR_ERR_INVALID_ARGUMENT, "normal map program out of range");
        if (m.num_components == 0) return fail(PYR_ERR_INVALID_ARGUMENT, "material without components");
        if (m.first_component + m.num_components > d->num_components || m.first_emissive + m.num_emissive > d->num_components)
            return fail(PYR_ERR_INVALID_ARGUMENT, "material component range out of bounds");
    }
    for (uint32_t i = 0; i < d->num_components; ++i) {
        const PyrComponent& c = d->components[i];
        if (c.color_program >= d->num_programs || (c.probability_program >= 0 && (uint32_t)c.probability_program >= d->num_programs))
            return fail(PYR_ERR_INVALID_ARGUMENT, "component program out of range");
    }
    for (uint32_t i = 0; i < d->num_programs; ++i) {
        const PyrProgram& p = d->programs[i];
        if (p.kind == PYR_PROGRAM_INSTRUCTIONS && p.first_instr + p.num_instrs > d->num_instrs)
            return fail(PYR_ERR_INVALID_ARGUMENT, "program instruction range out of bounds");
    }
    for (uint32_t i = 0; i < d->num_lamps; ++i) {
        const PyrLamp& l = d->lamps[i];
        if (l.kind == PYR_LAMP_SHAPE) {
            uint32_t limit = l.shape_kind == PYR_SHAPE_SPHERE ? d->num_spheres : (l.shape_kind == PYR_SHAPE_TRIANGLE ? d->num_triangles : 0);
            if (l.shape_index >= limit) return fail(PYR_ERR_INVALID_ARGUMENT, "lamp shape out of range");
        } else if (l.color_program >= d->num_programs) {
            return fail(PYR_ERR_INVALID_ARGUMENT, "lamp program out of range");
        }
    }
    return PYR_OK;
}

} // namespace

// =================================================================================================
// C entry points
// =================================================================================================
extern "C" {

const char* oracle_last_error(void) { return g_error.c_str(); }

int oracle_scene_create(const PyrSceneDesc* d, OracleScene** out) {
    if (!out) return fail(PYR_ERR_INVALID_ARGUMENT, "null out pointer");
    int rc = validate_desc(d);
    if (rc != PYR_OK) return rc;
    std::unique_ptr<OracleScene> s(new OracleScene());
    for (uint32_t i = 0; i < d->num_triangles; ++i) {
        const float* p = d->tri_positions + 9 * (size_t)i;
        const float* n = d->tri_normals + 9 * (size_t)i;
        Triangle t{};
        t.p1 = v3(p[0], p[1], p[2]);
        t.p2 = v3(p[3], p[4], p[5]);
        t.p3 = v3(p[6], p[7], p[8]);
        t.n1 = v3(n[0], n[1], n[2]);
        t.n2 = v3(n[3], n[4], n[5]);
        t.n3 = v3(n[6], n[7], n[8]);
        if (d->tri_uvs) {
            const float* uv = d->tri_uvs + 6 * (size_t)i;
            t.t1[0] = uv[0];
            t.t1[1] = uv[1];
            t.t2[0] = uv[2];
            t.t2[1] = uv[3];
            t.t3[0] = uv[4];
            t.t3[1] = uv[5];
        }
        t.edge1 = t.p2 - t.p1; // make_triangle / Shape::transform, world.rs:330-331, shapes/mod.rs:338-339
        t.edge2 = t.p3 - t.p1;
        t.material = d->tri_material[i];
        if (t.material >= d->num_materials) return fail(PYR_ERR_INVALID_ARGUMENT, "triangle material out of range");
        t.f1 = t.f2 = t.f3 = Quat{1.0f, 0.0f, 0.0f, 0.0f};
        if (d->tri_frames) {
            const float* f = d->tri_frames + 12 * (size_t)i;
            t.f1 = Quat{f[0], f[1], f[2], f[3]};
            t.f2 = Quat{f[4], f[5], f[6], f[7]};
            t.f3 = Quat{f[8], f[9], f[10], f[11]};
        }
        s->triangles.push_back(t);
    }
    for (uint32_t i = 0; i < d->num_spheres; ++i) {
        const float* p = d->spheres + 4 * (size_t)i;
        Sphere sp{};
        sp.position = v3(p[0], p[1], p[2]);
        sp.radius = p[3];
        sp.tex_scale[0] = d->sphere_tex_scale ? d->sphere_tex_scale[2 * i] : 1.0f;
        sp.tex_scale[1] = d->sphere_tex_scale ? d->sphere_tex_scale[2 * i + 1] : 1.0f;
        sp.material = d->sphere_material[i];
        if (sp.material >= d->num_materials) return fail(PYR_ERR_INVALID_ARGUMENT, "sphere material out of range");
        s->spheres.push_back(sp);
    }
    for (uint32_t i = 0; i < d->num_planes; ++i) {
        const float* p = d->planes + 8 * (size_t)i;
        Plane pl{};
        pl.origin = v3(p[0], p[1], p[2]);
        pl.normal = v3(p[3], p[4], p[5]);
        pl.tex_scale[0] = p[6];
        pl.tex_scale[1] = p[7];
        pl.material = d->plane_material[i];
        if (pl.material >= d->num_materials) return fail(PYR_ERR_INVALID_ARGUMENT, "plane material out of range");
        if (d->plane_frames) {
            const float* f = d->plane_frames + 4 * (size_t)i;
            pl.frame = Quat{f[0], f[1], f[2], f[3]};
        } else { // world.rs:88-100: basis(normal) -> Matrix3::from_cols(binormal, tangent, normal).into()
            V3 z = normalize(ortho(pl.normal));
            V3 y = normalize(cross(z, pl.normal));
            pl.frame = quat_from_cols(y, z, pl.normal);
        }
        s->planes.push_back(pl);
    }
    s->lamps.assign(d->lamps, d->lamps + d->num_lamps);
    s->materials.assign(d->materials, d->materials + d->num_materials);
    s->components.assign(d->components, d->components + d->num_components);
    s->programs.assign(d->programs, d->programs + d->num_programs);
    s->instrs.assign(d->instrs, d->instrs + d->num_instrs);
    s->spectra.assign(d->spectra, d->spectra + d->num_spectra);
    s->spectrum_data.assign(d->spectrum_data, d->spectrum_data + d->num_spectrum_floats);
    if (d->rgb_basis) s->rgb_basis.assign(d->rgb_basis, d->rgb_basis + 3 * (size_t)d->rgb_basis_count);
    s->rgb_basis_min = d->rgb_basis_min;
    s->rgb_basis_max = d->rgb_basis_max;
    s->sky_program = d->sky_program;
    if (d->num_textures) {
        s->texture_data.assign(d->texture_data, d->texture_data + d->num_texture_floats);
        for (uint32_t i = 0; i < d->num_textures; ++i) {
            const PyrTexture& t = d->textures[i];
            s->textures.push_back(Texture{t.width, t.height, t.format == PYR_TEXTURE_COLOR ? 4u : 1u, s->texture_data.data() + t.offset});
        }
    }
    build_bvh(*s);
    *out = s.release();
    return PYR_OK;
}

void oracle_scene_destroy(OracleScene* scene) { delete scene; }

int oracle_render_simple(OracleScene* scene, const PyrCamera* camera, const PyrFilmDesc* film, const PyrRenderParams* params,
                         PyrGrain* film_inout, int threads, PyrCounters* counters) {
    if (!scene || !camera || !film || !params || !film_inout) return fail(PYR_ERR_INVALID_ARGUMENT, "null argument");
    if (params->spectrum_samples == 0 || params->tile_size == 0 || film->bins == 0 || film->width == 0 || film->height == 0)
        return fail(PYR_ERR_INVALID_ARGUMENT, "zero-sized parameter");
    std::vector<Tile> tiles = make_tiles(film->width, film->height, params->tile_size);
    uint32_t tile_begin = params->tile_begin, tile_end = params->tile_end ? params->tile_end : (uint32_t)tiles.size();
    const uint32_t tile_stride = std::max(1u, params->tile_stride);
    std::vector<Tile> selected;
    for (const Tile& t : tiles)
        if (t.raster_index >= tile_begin && t.raster_index < tile_end && (t.raster_index - tile_begin) % tile_stride == 0) selected.push_back(t);
    FilmView view{film_inout, *film, params->film_row_begin, params->film_row_count ? params->film_row_count : film->height};
    if ((uint64_t)view.row_begin + view.row_count > film->height) return fail(PYR_ERR_INVALID_ARGUMENT, "film window exceeds the image");
    if (params->film_layout > PYR_FILM_TILE_BLOCKS) return fail(PYR_ERR_INVALID_ARGUMENT, "unknown film layout");
    if (params->film_layout == PYR_FILM_TILE_BLOCKS) {
        if (params->film_row_begin || params->film_row_count) return fail(PYR_ERR_INVALID_ARGUMENT, "a film of tile blocks has no row window");
        view.layout = PYR_FILM_TILE_BLOCKS;
        view.tile_size = params->tile_size;
        view.tiles_x = (film->width + params->tile_size - 1) / params->tile_size;
        view.tile_begin = tile_begin;
        view.tile_stride = tile_stride;
    }

    int n_threads = std::max(1, threads);
    std::atomic<size_t> next{0};
    std::vector<Counters> per_thread(n_threads);
    auto worker = [&](int id) { // TaskRunner::run_tasks, renderer/mod.rs:125-189: idle workers pull the next tile
        for (;;) {
            size_t k = next.fetch_add(1);
            if (k >= selected.size()) break;
            render_tile(*scene, selected[k], view, *camera, *params, per_thread[id]);
        }
    };
    if (n_threads == 1) {
        worker(0);
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < n_threads; ++t) pool.emplace_back(worker, t);
        for (auto& th : pool) th.join();
    }
    if (counters) {
        std::memset(counters, 0, sizeof(*counters));
        for (const Counters& c : per_thread) c.add_to(counters);
    }
    return PYR_OK;
}

int oracle_intersect(OracleScene* scene, const float* rays, uint32_t n, PyrHit* hits, PyrCounters* counters) {
    if (!scene || (!rays && n) || (!hits && n)) return fail(PYR_ERR_INVALID_ARGUMENT, "null argument");
    Counters c;
    for (uint32_t i = 0; i < n; ++i) {
        const float* r = rays + 6 * (size_t)i;
        Ray ray{v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5])};
        Intersection it;
        if (world_intersect(*scene, ray, it, c)) {
            hits[i].distance = it.distance;
            hits[i].shape = (it.shape.kind << 30) | it.shape.index;
            hits[i].u = it.u;
            hits[i].v = it.v;
        } else {
            hits[i].distance = INF;
            hits[i].shape = PYR_HIT_NONE;
            hits[i].u = hits[i].v = 0.0f;
        }
    }
    if (counters) {
        std::memset(counters, 0, sizeof(*counters));
        c.add_to(counters);
    }
    return PYR_OK;
}

uint32_t oracle_bvh_num_nodes(OracleScene* scene) { return scene ? (uint32_t)scene->nodes.size() : 0; }

int oracle_bvh_node(OracleScene* scene, uint32_t index, float* aabb6, uint32_t* subtree_size, uint32_t* item) {
    if (!scene || index >= scene->nodes.size()) return fail(PYR_ERR_INVALID_ARGUMENT, "node index out of range");
    const FlatNode& n = scene->nodes[index];
    aabb6[0] = n.bounding_box.min.x;
    aabb6[1] = n.bounding_box.min.y;
    aabb6[2] = n.bounding_box.min.z;
    aabb6[3] = n.bounding_box.max.x;
    aabb6[4] = n.bounding_box.max.y;
    aabb6[5] = n.bounding_box.max.z;
    *subtree_size = n.subtree_size;
    *item = n.subtree_size == 0 ? ((n.item.kind << 30) | n.item.index) : PYR_HIT_NONE;
    return PYR_OK;
}

// ---- KAT entry points ----
static Rng load(const uint32_t st[4]) { return Rng{st[0], st[1], st[2], st[3]}; }
static void store(const Rng& r, uint32_t st[4]) {
    st[0] = r.x;
    st[1] = r.y;
    st[2] = r.z;
    st[3] = r.w;
}
static V3 lv(const float* p) { return v3(p[0], p[1], p[2]); }
static void sv(V3 v, float* p) {
    p[0] = v.x;
    p[1] = v.y;
    p[2] = v.z;
}

void oracle_rng_seed(uint64_t seed, uint32_t tile, uint64_t iteration, uint32_t state[4]) { store(rng_seed(seed, tile, iteration), state); }
uint32_t oracle_rng_next_u32(uint32_t state[4]) {
    Rng r = load(state);
    uint32_t v = r.next_u32();
    store(r, state);
    return v;
}
float oracle_rng_gen_f32(uint32_t state[4]) {
    Rng r = load(state);
    float v = gen_f32(r);
    store(r, state);
    return v;
}
float oracle_rng_gen_range_f32(uint32_t state[4], float low, float high) {
    Rng r = load(state);
    float v = gen_range_f32(r, low, high);
    store(r, state);
    return v;
}
uint32_t oracle_rng_gen_range_usize(uint32_t state[4], uint32_t n) {
    Rng r = load(state);
    uint32_t v = gen_range_usize(r, n);
    store(r, state);
    return v;
}
uint32_t oracle_rng_choose_index(uint32_t state[4], uint32_t n) {
    Rng r = load(state);
    uint32_t v = choose_index(r, n);
    store(r, state);
    return v;
}

int oracle_aabb_intersection_distance(const float a[6], const float r[6], float* distance) {
    Aabb bb{lv(a), lv(a + 3)};
    Ray ray{lv(r), lv(r + 3)};
    float d = 0.0f;
    bool hit = aabb_intersection_distance(bb, ray, d);
    *distance = d;
    return hit ? 1 : 0;
}
float oracle_schlick(float n1, float n2, const float normal[3], const float incident[3]) { return schlick(n1, n2, lv(normal), lv(incident)); }
float oracle_fresnel(float ior, float env_ior, const float normal[3], const float incident[3]) { return fresnel(ior, env_ior, lv(normal), lv(incident)); }
void oracle_ortho(const float v[3], float out[3]) { sv(ortho(lv(v)), out); }
void oracle_sample_sphere(uint32_t state[4], float out[3]) {
    Rng r = load(state);
    sv(sample_sphere(r), out);
    store(r, state);
}
void oracle_sample_hemisphere(uint32_t state[4], const float dir[3], float out[3]) {
    Rng r = load(state);
    sv(sample_hemisphere(r, lv(dir)), out);
    store(r, state);
}
void oracle_sample_cone(uint32_t state[4], const float dir[3], float cos_half, float out[3]) {
    Rng r = load(state);
    sv(sample_cone(r, lv(dir), cos_half), out);
    store(r, state);
}
float oracle_solid_angle(float cos_half) { return solid_angle(cos_half); }
float oracle_sin32(float x) { return sin32(x); }
float oracle_cos32(float x) { return cos32(x); }
float oracle_acos32(float x) { return acos32(x); }
float oracle_blackbody(float wavelength, float temperature) { return blackbody(wavelength, temperature); }

int oracle_triangle_intersect(const float v1[3], const float v2[3], const float v3_[3], const float r[6], float* dist, float* u, float* v) {
    Triangle t{};
    t.p1 = lv(v1);
    t.p2 = lv(v2);
    t.p3 = lv(v3_);
    t.edge1 = t.p2 - t.p1;
    t.edge2 = t.p3 - t.p1;
    Ray ray{lv(r), lv(r + 3)};
    float d = 0, uu = 0, vv = 0;
    bool hit = triangle_intersect(t, ray, d, uu, vv);
    *dist = d;
    *u = uu;
    *v = vv;
    return hit ? 1 : 0;
}
int oracle_sphere_intersect(const float centre[3], float radius, const float r[6], float* dist, float point[3]) {
    Ray ray{lv(r), lv(r + 3)};
    float d = 0;
    V3 p = v3(0, 0, 0);
    bool hit = sphere_intersect(lv(centre), radius, ray, d, p);
    *dist = d;
    sv(p, point);
    return hit ? 1 : 0;
}
float oracle_spectrum_get(uint32_t format, float min, float max, const float* data, uint32_t count, float wavelength) {
    return spectrum_get(format, min, max, data, count, wavelength);
}
float oracle_refract(uint32_t state[4], float ior, float env_ior, const float in_dir[3], const float normal[3], float out_dir[3]) {
    Rng r = load(state);
    V3 out;
    float p;
    refract(ior, env_ior, lv(in_dir), lv(normal), r, out, p);
    store(r, state);
    sv(out, out_dir);
    return p;
}
void oracle_sample_wavelengths(uint32_t state[4], float start, float width, uint32_t s, float* out) {
    Rng r = load(state);
    std::vector<float> w;
    float step_size = width / (float)s;
    float from = start;
    for (uint32_t k = 0; k < s; ++k) {
        float to = from + step_size;
        w.push_back(gen_range_f32(r, from, to));
        from = to;
    }
    uint32_t hero = gen_range_usize(r, s);
    float main = w[hero];
    w[hero] = w.back();
    w.pop_back();
    out[0] = main;
    for (uint32_t k = 0; k + 1 < s; ++k) out[k + 1] = w[k];
    store(r, state);
}
uint32_t oracle_wavelength_to_grain(float wavelength, float start, float width, uint32_t bins) { return wavelength_to_grain(wavelength, start, width, bins); }
int oracle_to_pixel(uint32_t width, uint32_t height, float x, float y, uint32_t* px, uint32_t* py) {
    uint64_t a = 0, b = 0;
    if (!to_pixel(width, height, x, y, a, b)) return 0;
    if (a >= width || b >= height) return 0;
    *px = (uint32_t)a;
    *py = (uint32_t)b;
    return 1;
}
void oracle_to_view_area(uint32_t x, uint32_t y, uint32_t w, uint32_t h, uint32_t image_w, uint32_t image_h, float out[4]) {
    Area a = to_view_area(x, y, w, h, image_w, image_h);
    out[0] = a.from_x;
    out[1] = a.from_y;
    out[2] = a.size_x;
    out[3] = a.size_y;
}
void oracle_ray_towards(const PyrCamera* camera, uint32_t state[4], float x, float y, float ray6[6]) {
    Rng r = load(state);
    Ray ray = ray_towards(*camera, x, y, r);
    store(r, state);
    sv(ray.origin, ray6);
    sv(ray.direction, ray6 + 3);
}
uint32_t oracle_tile_order(uint32_t width, uint32_t height, uint32_t tile_size, uint32_t* order, uint32_t capacity) {
    std::vector<Tile> tiles = make_tiles(width, height, tile_size);
    for (uint32_t i = 0; i < tiles.size() && i < capacity; ++i) order[i] = tiles[i].raster_index;
    return (uint32_t)tiles.size();
}
float oracle_run_program(OracleScene* scene, uint32_t program, float wavelength, const float normal[3], const float incident[3],
                         const float texture[2], int* wavelength_used) {
    Exe exe(scene);
    ProgramInput in{wavelength, lv(normal), lv(incident), {texture[0], texture[1]}};
    float v = exe.run(program, in);
    if (wavelength_used) *wavelength_used = in.wavelength_used ? 1 : 0;
    return v;
}

void oracle_texture_get(uint32_t channels, uint32_t width, uint32_t height, const float* texels, float x, float y, float* out) {
    Texture t{width, height, channels, texels};
    texture_get_color(t, x, y, out);
}
void oracle_quat_from_cols(const float c0[3], const float c1[3], const float c2[3], float out[4]) {
    Quat q = quat_from_cols(lv(c0), lv(c1), lv(c2));
    out[0] = q.s, out[1] = q.x, out[2] = q.y, out[3] = q.z;
}
void oracle_quat_rotate(const float q[4], const float v[3], float out[3]) { sv(quat_rotate(Quat{q[0], q[1], q[2], q[3]}, lv(v)), out); }
int oracle_surface_data(OracleScene* scene, const float ray6[6], float wavelength, float normal[3], float texture[2], float frame[4],
                        float shading_normal[3]) {
    Ray ray{v3(ray6[0], ray6[1], ray6[2]), v3(ray6[3], ray6[4], ray6[5])};
    Intersection hit;
    Counters c;
    if (!world_intersect(*scene, ray, hit, c)) return 0;
    SurfaceData sd = surface_data(*scene, hit);
    sv(sd.normal, normal);
    texture[0] = sd.texture[0], texture[1] = sd.texture[1];
    frame[0] = sd.from_space.s, frame[1] = sd.from_space.x, frame[2] = sd.from_space.y, frame[3] = sd.from_space.z;
    V3 n = sd.normal;
    const PyrMaterial& material = scene->materials[shape_material(*scene, hit.shape)];
    if (material.normal_map_program >= 0) {
        Exe exe(scene);
        ProgramInput in{wavelength, sd.normal, ray.direction, {sd.texture[0], sd.texture[1]}};
        V4 m = exe.run_vector((uint32_t)material.normal_map_program, in);
        n = normalize(quat_rotate(sd.from_space, v3(m.x, m.y, m.z)));
    }
    sv(n, shading_normal);
    return 1;
}

// ---- film development: main.rs:315-327 (final pass), spectrum_to_xyz :352-369, spectrum_to_tristimulus :371-418,
// film::Spectrum::get film.rs:321-337, DevelopedPixels film.rs:282-313, Grain::develop film.rs:132-143.
// [3P] palette 0.7.2: Xyz(D65) -> linear sRGB with the sRGB/D65 matrix, clamp to [0,1] (FromColor), sRGB transfer function,
// u8 = round(255 * v). The crate derives its matrix from the primaries at run time and converts to u8 through a lookup
// table; both may differ from the textbook constants used here in the last bit (unverifiable offline).
int oracle_film_develop(const PyrFilmDesc* film, const PyrGrain* grains, const PyrDevelopParams* p, uint8_t* rgb_out) {
    if (!film || !grains || !p || !rgb_out || !p->xyz_table) return fail(PYR_ERR_INVALID_ARGUMENT, "null argument");
    const uint32_t bins = film->bins;
    const size_t pixels = (size_t)film->width * film->height;
    const float min = film->wl_start, max = film->wl_start + film->wl_width;
    auto xyz_get = [&](int channel, float w) { // Spectrum::Array::get on the interleaved table
        const float* d = p->xyz_table;
        const uint32_t n = p->xyz_count;
        if (w <= p->xyz_min) return d[channel];
        if (w >= p->xyz_max) return d[3 * (n - 1) + channel];
        float normalized = (w - p->xyz_min) / (p->xyz_max - p->xyz_min);
        float fi = normalized * ((float)n - 1.0f);
        float fmin_ = std::trunc(fi);
        uint32_t i0 = (uint32_t)fmin_;
        float mix = fi - fmin_;
        return d[3 * i0 + channel] * (1.0f - mix) + d[3 * (i0 + 1) + channel] * mix;
    };
    auto encode = [](float v) -> uint8_t { // sRGB transfer function, pow through f64
        v = rmin(rmax(v, 0.0f), 1.0f);
        float e = v <= 0.0031308f ? 12.92f * v : 1.055f * (float)std::pow((double)v, 1.0 / 2.4) - 0.055f;
        e = rmin(rmax(e, 0.0f), 1.0f);
        return (uint8_t)(e * 255.0f + 0.5f);
    };
    std::memset(rgb_out, 0, pixels * 3);
    for (size_t px = 0; px < pixels; ++px) {
        if ((px + 1) * bins >= pixels * bins) break; // DevelopedPixels::next stops when `end < len` fails: the last pixel is skipped
        const PyrGrain* g = grains + px * bins;
        auto sample = [&](float w, uint32_t i) {
            float intensity;
            if (w < min || w > max) {
                intensity = 0.0f;
            } else {
                float normalized = (w - min) / (max - min);
                float float_index = normalized * (float)bins;
                uint32_t index = (uint32_t)std::min<float>(std::floor(float_index), (float)(bins - 1));
                intensity = g[index].weight > 0.0f ? g[index].acc / g[index].weight : 0.0f;
            }
            if (p->filter) intensity = intensity * p->filter[i];
            if (p->white_div) intensity = (intensity / p->white_div[i]) * p->white_mul[i];
            return intensity;
        };
        float sum[3] = {0, 0, 0}, weight = 0.0f;
        float wl_min = min;
        uint32_t i = 0;
        float spectrum_min = sample(wl_min, i);
        float start[3] = {xyz_get(0, wl_min), xyz_get(1, wl_min), xyz_get(2, wl_min)};
        while (wl_min < max) {
            float wl_max = wl_min + p->step_size;
            i += 1;
            float spectrum_max = sample(wl_max, i < p->sample_count ? i : p->sample_count - 1);
            float end[3] = {xyz_get(0, wl_max), xyz_get(1, wl_max), xyz_get(2, wl_max)};
            float w = wl_max - wl_min;
            for (int c = 0; c < 3; ++c) sum[c] += (start[c] * spectrum_min + end[c] * spectrum_max) * 0.5f * w;
            weight += w;
            wl_min = wl_max;
            spectrum_min = spectrum_max;
            for (int c = 0; c < 3; ++c) start[c] = end[c];
        }
        float xyz[3];
        for (int c = 0; c < 3; ++c) xyz[c] = (weight == 0.0f ? sum[c] : sum[c] / weight) * p->xyz_scale;
        float r = 3.2404542f * xyz[0] + -1.5371385f * xyz[1] + -0.4985314f * xyz[2];
        float gg = -0.9692660f * xyz[0] + 1.8760108f * xyz[1] + 0.0415560f * xyz[2];
        float b = 0.0556434f * xyz[0] + -0.2040259f * xyz[1] + 1.0572252f * xyz[2];
        rgb_out[3 * px + 0] = encode(r);
        rgb_out[3 * px + 1] = encode(gg);
        rgb_out[3 * px + 2] = encode(b);
    }
    return PYR_OK;
}

} // extern "C"
