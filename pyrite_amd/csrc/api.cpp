// api.cpp -- host side of the C ABI declared in include/pyrite_gpu.h: scene validation and packing, BVH build,
// upload, launch orchestration. No radiance is ever computed on the host; without a gfx950 device every render /
// intersect entry point fails with PYR_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "api_internal.h"
#include "bvh.h"
#include "device_scene.h"

using namespace pyr;

namespace {

thread_local std::string g_error;
int fail(int code, const std::string& message) {
    g_error = message;
    return code;
}
int hip_fail(hipError_t e, const char* what) { return fail(PYR_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e)); }

#define HIP_TRY(expr)                                  \
    do {                                               \
        hipError_t e_ = (expr);                        \
        if (e_ != hipSuccess) return hip_fail(e_, #expr); \
    } while (0)

struct DeviceBuffer {
    void* ptr = nullptr;
    size_t bytes = 0;
    ~DeviceBuffer() { release(); }
    void release() {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        bytes = 0;
    }
    int upload(const void* src, size_t n) {
        bytes = n;
        if (n == 0) n = 16; // keep pointers non-null
        HIP_TRY(hipMalloc(&ptr, n));
        if (bytes) HIP_TRY(hipMemcpy(ptr, src, bytes, hipMemcpyHostToDevice));
        return PYR_OK;
    }
    int alloc(size_t n) {
        bytes = n;
        HIP_TRY(hipMalloc(&ptr, n ? n : 16));
        return PYR_OK;
    }
};

int validate(const PyrSceneDesc* d) {
    if (!d) return fail(PYR_ERR_INVALID_ARGUMENT, "null scene description");
    if (d->num_programs == 0 || d->sky_program >= d->num_programs) return fail(PYR_ERR_INVALID_ARGUMENT, "sky program out of range");
    if ((d->num_triangles && (!d->tri_positions || !d->tri_normals || !d->tri_material)) || (d->num_spheres && (!d->spheres || !d->sphere_material)) ||
        (d->num_planes && (!d->planes || !d->plane_material)))
        return fail(PYR_ERR_INVALID_ARGUMENT, "null geometry array");
    // a leaf code packs (first_primitive << 3 | count) into 31 bits (bvh.h): the primitives together must stay below 2^28
    if ((uint64_t)d->num_triangles + d->num_spheres >= (1ull << 28)) return fail(PYR_ERR_INVALID_ARGUMENT, "too many primitives");
    if ((d->num_programs && !d->programs) || (d->num_instrs && !d->instrs) || (d->num_materials && !d->materials) || (d->num_components && !d->components) ||
        (d->num_lamps && !d->lamps) || (d->num_spectra && !d->spectra) || (d->num_spectrum_floats && !d->spectrum_data))
        return fail(PYR_ERR_INVALID_ARGUMENT, "null table with a non-zero count");
    for (uint32_t i = 0; i < d->num_instrs; ++i) {
        const PyrInstr& ins = d->instrs[i];
        if (ins.op == PYR_OP_COLOR_TEXTURE || ins.op == PYR_OP_MONO_TEXTURE) {
            if (ins.a >= d->num_textures || !d->textures) return fail(PYR_ERR_INVALID_ARGUMENT, "texture id out of range");
            if (d->textures[ins.a].format != (ins.op == PYR_OP_COLOR_TEXTURE ? PYR_TEXTURE_COLOR : PYR_TEXTURE_MONO))
                return fail(PYR_ERR_INVALID_ARGUMENT, "texture format does not match the opcode");
        }
        if (ins.op > PYR_OP_CLAMP) return fail(PYR_ERR_INVALID_ARGUMENT, "unknown opcode");
        if (ins.op == PYR_OP_SPECTRUM && ins.a >= d->num_spectra) return fail(PYR_ERR_INVALID_ARGUMENT, "spectrum id out of range");
        if (ins.op == PYR_OP_RGB_SPECTRUM && !d->rgb_basis) return fail(PYR_ERR_INVALID_ARGUMENT, "RgbSpectrumValue needs rgb_basis");
    }
    for (uint32_t i = 0; i < d->num_spectra; ++i) {
        const PyrSpectrum& s = d->spectra[i];
        uint64_t floats = s.format == PYR_SPECTRUM_CURVE ? 2ull * s.count : s.count;
        if (s.offset + floats > d->num_spectrum_floats) return fail(PYR_ERR_INVALID_ARGUMENT, "spectrum data out of range");
        // Spectrum::get interpolates between samples i and i + 1 for min < w < max (project/spectra.rs:44-54): with one sample
        // the reference indexes out of bounds and panics; here the description is refused
        if (s.format == PYR_SPECTRUM_ARRAY && s.count == 1 && s.min < s.max) return fail(PYR_ERR_INVALID_ARGUMENT, "array spectrum with one sample over a non-empty span");
        if (s.format > PYR_SPECTRUM_CURVE) return fail(PYR_ERR_INVALID_ARGUMENT, "unknown spectrum format");
    }
    for (uint32_t i = 0; i < d->num_textures; ++i) {
        const PyrTexture& t = d->textures[i];
        uint64_t floats = (uint64_t)t.width * t.height * (t.format == PYR_TEXTURE_COLOR ? 4u : 1u);
        if (t.width == 0 || t.height == 0 || t.format > PYR_TEXTURE_MONO || !d->texture_data || t.offset + floats > d->num_texture_floats)
            return fail(PYR_ERR_INVALID_ARGUMENT, "texture data out of range");
    }
    for (uint32_t i = 0; i < d->num_programs; ++i) {
        const PyrProgram& p = d->programs[i];
        if (p.kind == PYR_PROGRAM_INSTRUCTIONS) {
            if ((uint64_t)p.first_instr + p.num_instrs > d->num_instrs) return fail(PYR_ERR_INVALID_ARGUMENT, "program instruction range out of bounds");
            if (p.num_numbers > PYR_MAX_NUMBER_REGISTERS || p.num_vectors > PYR_MAX_VECTOR_REGISTERS || p.num_rgbs > PYR_MAX_RGB_REGISTERS)
                return fail(PYR_ERR_UNSUPPORTED, "program needs more registers than the GPU VM provides");
        }
    }
    for (uint32_t i = 0; i < d->num_materials; ++i) {
        const PyrMaterial& m = d->materials[i];
        if (m.normal_map_program >= 0 && (uint32_t)m.normal_map_program >= d->num_programs)
            return fail(PYR_ERR_INVALID_ARGUMENT, "normal map program out of range");
        if (m.num_components == 0) return fail(PYR_ERR_INVALID_ARGUMENT, "material without components");
        if ((uint64_t)m.first_component + m.num_components > d->num_components || (uint64_t)m.first_emissive + m.num_emissive > d->num_components)
            return fail(PYR_ERR_INVALID_ARGUMENT, "material component range out of bounds");
    }
    for (uint32_t i = 0; i < d->num_components; ++i) {
        const PyrComponent& c = d->components[i];
        if (c.bsdf > PYR_BSDF_REFRACTIVE) return fail(PYR_ERR_INVALID_ARGUMENT, "unknown bsdf");
        if (c.color_program >= d->num_programs || (c.probability_program >= 0 && (uint32_t)c.probability_program >= d->num_programs))
            return fail(PYR_ERR_INVALID_ARGUMENT, "component program out of range");
    }
    for (uint32_t i = 0; i < d->num_triangles; ++i)
        if (d->tri_material[i] >= d->num_materials) return fail(PYR_ERR_INVALID_ARGUMENT, "triangle material out of range");
    for (uint32_t i = 0; i < d->num_spheres; ++i)
        if (d->sphere_material[i] >= d->num_materials) return fail(PYR_ERR_INVALID_ARGUMENT, "sphere material out of range");
    for (uint32_t i = 0; i < d->num_planes; ++i)
        if (d->plane_material[i] >= d->num_materials) return fail(PYR_ERR_INVALID_ARGUMENT, "plane material out of range");
    for (uint32_t i = 0; i < d->num_lamps; ++i) {
        const PyrLamp& l = d->lamps[i];
        if (l.kind == PYR_LAMP_SHAPE) {
            uint32_t limit = l.shape_kind == PYR_SHAPE_SPHERE ? d->num_spheres : (l.shape_kind == PYR_SHAPE_TRIANGLE ? d->num_triangles : 0);
            if (l.shape_index >= limit) return fail(PYR_ERR_INVALID_ARGUMENT, "lamp shape out of range");
            uint32_t mat = l.shape_kind == PYR_SHAPE_SPHERE ? d->sphere_material[l.shape_index] : d->tri_material[l.shape_index];
            if (d->materials[mat].num_emissive == 0) return fail(PYR_ERR_INVALID_ARGUMENT, "lamp shape has no emissive component");
        } else if (l.kind > PYR_LAMP_SHAPE || l.color_program >= d->num_programs) {
            return fail(PYR_ERR_INVALID_ARGUMENT, "lamp program out of range");
        }
    }
    return PYR_OK;
}

float bits_to_float(uint32_t b) {
    float f;
    std::memcpy(&f, &b, 4);
    return f;
}
float shape_bits(uint32_t shape) { return bits_to_float(shape); }

bool operand_is_wavelength(const PyrOperand& o) { return o.kind == PYR_OPERAND_INPUT && o.bits == PYR_INPUT_WAVELENGTH; }

// Which operand slots an opcode evaluates (execution_context.rs:81-281): decides ProbabilityInput::wavelength_used.
bool instr_reads_wavelength(const PyrInstr& ins) {
    switch (ins.op) {
    case PYR_OP_VECTOR: return operand_is_wavelength(ins.x) || operand_is_wavelength(ins.y) || operand_is_wavelength(ins.z) || operand_is_wavelength(ins.w);
    case PYR_OP_RGB:
    case PYR_OP_CLAMP: return operand_is_wavelength(ins.x) || operand_is_wavelength(ins.y) || operand_is_wavelength(ins.z);
    case PYR_OP_SPECTRUM:
    case PYR_OP_RGB_SPECTRUM:
    case PYR_OP_MIX: return operand_is_wavelength(ins.x);
    case PYR_OP_FRESNEL:
    case PYR_OP_BLACKBODY: return operand_is_wavelength(ins.x) || operand_is_wavelength(ins.y);
    default: return false;
    }
}

DevProgram pack_program(const PyrInstr* instrs, const PyrProgram& p) {
    DevProgram o{};
    o.kind = p.kind;
    o.constant = p.constant;
    o.first_instr = p.first_instr;
    o.num_instrs = p.num_instrs;
    o.output_kind = p.output_kind;
    o.output_reg = p.output_reg;
    o.fast = FAST_NONE;
    o.tape_form = TAPE_FORM_DIRECT;
    if (p.kind != PYR_PROGRAM_INSTRUCTIONS) return o;
    const PyrInstr* I = instrs + p.first_instr;
    for (uint32_t k = 0; k < p.num_instrs; ++k)
        if (instr_reads_wavelength(I[k])) o.reads_wavelength = 1;
    // the tape form of a program the interpreter has to run (device_scene.h TapeForm); a fast shape found below is DIRECT again
    {
        auto depends_on_wavelength = [&](const PyrInstr& ins) { return (ins.deps & PYR_DEP_WAVELENGTH) != 0u || instr_reads_wavelength(ins); };
        uint32_t dependent = 0;
        for (uint32_t k = 0; k < p.num_instrs; ++k) dependent += depends_on_wavelength(I[k]) ? 1u : 0u;
        o.tape_form = TAPE_FORM_NONE;
        if (p.output_kind == PYR_OUTPUT_NUMBER && p.num_instrs != 0) {
            const PyrInstr& last = I[p.num_instrs - 1];
            // a function of the wavelength alone, numbers only: the subset kernels.hip lambda_eval interprets
            bool lambda = true;
            for (uint32_t k = 0; k < p.num_instrs; ++k) {
                const PyrInstr& ins = I[k];
                const bool number_op = ins.op == PYR_OP_NUMBER || ins.op == PYR_OP_SPECTRUM || ins.op == PYR_OP_BLACKBODY || ins.op == PYR_OP_CLAMP ||
                                       ((ins.op == PYR_OP_BINARY || ins.op == PYR_OP_MIX) && ins.value_type == PYR_VT_NUMBER);
                if (!number_op || (ins.deps & (PYR_DEP_NORMAL | PYR_DEP_INCIDENT | PYR_DEP_TEXTURE)) != 0u) lambda = false;
            }
            if (dependent == 0)
                o.tape_form = TAPE_FORM_HIT_VALUE;
            else if (lambda)
                o.tape_form = TAPE_FORM_LAMBDA;
            else if (dependent == 1 && last.op == PYR_OP_RGB_SPECTRUM && operand_is_wavelength(last.x) && last.output == p.output_reg) {
                o.tape_form = TAPE_FORM_HIT_RGB;
                o.tape_rgb_reg = last.a;
            }
        }
    }
    auto is_spectrum = [&](const PyrInstr& ins) { return ins.op == PYR_OP_SPECTRUM && operand_is_wavelength(ins.x); };
    auto is_mul = [&](const PyrInstr& ins) { return ins.op == PYR_OP_BINARY && ins.value_type == PYR_VT_NUMBER && ins.operator_ == PYR_BIN_MUL; };
    if (p.output_kind != PYR_OUTPUT_NUMBER) return o;
    if (p.num_instrs == 1 && is_spectrum(I[0]) && p.output_reg == I[0].output) {
        o.fast = FAST_SPECTRUM;
        o.fast_spectrum = I[0].a;
    } else if (p.num_instrs == 3 && is_mul(I[2]) && p.output_reg == I[2].output) {
        // [Spectrum -> r, Number c -> q, r * q] or [Number c -> q, Spectrum -> r, q * r] (compiler.rs convert_operands order)
        if (is_spectrum(I[0]) && I[1].op == PYR_OP_NUMBER && I[2].a == I[0].output && I[2].b == I[1].output && I[0].output != I[1].output) {
            o.fast = FAST_SPECTRUM_MUL;
            o.fast_spectrum = I[0].a;
            o.fast_scale = bits_to_float(I[1].x.bits);
        } else if (I[0].op == PYR_OP_NUMBER && is_spectrum(I[1]) && I[2].a == I[0].output && I[2].b == I[1].output && I[0].output != I[1].output) {
            o.fast = FAST_MUL_SPECTRUM;
            o.fast_spectrum = I[1].a;
            o.fast_scale = bits_to_float(I[0].x.bits);
        }
    }
    if (o.fast != FAST_NONE) o.tape_form = TAPE_FORM_DIRECT;
    return o;
}

bool instr_depends_on_wavelength(const PyrInstr& ins) { return (ins.deps & PYR_DEP_WAVELENGTH) != 0u || instr_reads_wavelength(ins); }

// TAPE_FORM_PRODUCT (device_scene.h): a number program without a tape form of its own whose value is a chain of products
// ((lambda * h1) * h2) ... -- ONE factor made by number-only instructions that depend on the wavelength and on nothing else, every other
// factor made by instructions that do not depend on the wavelength -- is split into those two instruction lists, appended to `instrs`
// as two programs (the hit side first; registers keep their numbers). `chain` receives the hit side's registers in the order the
// products are formed, innermost first (at most three). Returns false when the program does not factor that way.
bool split_product(std::vector<PyrInstr>& instrs, const PyrProgram& p, PyrProgram& hit, PyrProgram& lambda, std::vector<uint32_t>& chain) {
    if (p.kind != PYR_PROGRAM_INSTRUCTIONS || p.output_kind != PYR_OUTPUT_NUMBER || p.num_instrs < 3) return false;
    const size_t first = p.first_instr, end = first + p.num_instrs;
    auto writes_number = [](const PyrInstr& ins) {
        return ins.op == PYR_OP_NUMBER || ins.op == PYR_OP_SPECTRUM || ins.op == PYR_OP_BLACKBODY || ins.op == PYR_OP_CLAMP || ins.op == PYR_OP_FRESNEL ||
               ins.op == PYR_OP_MONO_TEXTURE || ins.op == PYR_OP_RGB_SPECTRUM || ((ins.op == PYR_OP_BINARY || ins.op == PYR_OP_MIX) && ins.value_type == PYR_VT_NUMBER);
    };
    auto is_number_mul = [](const PyrInstr& ins) { return ins.op == PYR_OP_BINARY && ins.value_type == PYR_VT_NUMBER && ins.operator_ == PYR_BIN_MUL; };
    auto hit_deps = [](const PyrInstr& ins) { return (ins.deps & (PYR_DEP_NORMAL | PYR_DEP_INCIDENT | PYR_DEP_TEXTURE)) != 0u; };
    auto writer_before = [&](uint32_t reg, size_t before) { // the instruction whose result a read of number register `reg` at `before` sees
        for (size_t k = before; k-- > first;)
            if (writes_number(instrs[k]) && instrs[k].output == reg) return (long)k;
        return -1L;
    };
    auto written_once_more = [&](uint32_t reg, size_t after) { // ... and nobody writes it again (the split programs read registers at their ends)
        for (size_t k = after + 1; k < end; ++k)
            if (writes_number(instrs[k]) && instrs[k].output == reg) return true;
        return false;
    };
    // from the closing product down to the factor that depends on the wavelength alone
    std::vector<bool> on_chain(p.num_instrs, false);
    size_t cur = end - 1;
    if (!is_number_mul(instrs[cur]) || instrs[cur].output != p.output_reg) return false;
    std::vector<uint32_t> outer_first;
    long lambda_writer = -1;
    for (;;) {
        const PyrInstr mul = instrs[cur];
        if (mul.a == mul.b) return false;
        const long wa = writer_before(mul.a, cur), wb = writer_before(mul.b, cur);
        if (wa < 0 || wb < 0) return false;
        const bool la = instr_depends_on_wavelength(instrs[(size_t)wa]), lb = instr_depends_on_wavelength(instrs[(size_t)wb]);
        if (la == lb) return false;
        const long lw = la ? wa : wb, hw = la ? wb : wa;
        const uint32_t hit_reg = la ? mul.b : mul.a;
        if (written_once_more(hit_reg, (size_t)hw) || outer_first.size() == 3) return false;
        outer_first.push_back(hit_reg);
        on_chain[cur - first] = true;
        if (hit_deps(instrs[(size_t)lw])) { // the wavelength side is itself a product with something of the hit in it: one level down
            if (!is_number_mul(instrs[(size_t)lw])) return false;
            cur = (size_t)lw;
            continue;
        }
        lambda_writer = lw;
        break;
    }
    if (written_once_more(instrs[(size_t)lambda_writer].output, (size_t)lambda_writer)) return false;
    std::vector<PyrInstr> hit_list, lambda_list;
    for (size_t k = first; k < end; ++k) {
        const PyrInstr& ins = instrs[k];
        if (on_chain[k - first]) continue;
        if (instr_depends_on_wavelength(ins)) {
            const bool number_op = ins.op == PYR_OP_SPECTRUM || ins.op == PYR_OP_BLACKBODY || ins.op == PYR_OP_CLAMP || ((ins.op == PYR_OP_BINARY || ins.op == PYR_OP_MIX) && ins.value_type == PYR_VT_NUMBER);
            if (!number_op || hit_deps(ins) || k > (size_t)lambda_writer) return false; // a second factor that reads the wavelength, or one that reads the hit too
            lambda_list.push_back(ins);
        } else {
            hit_list.push_back(ins);
            if (ins.op == PYR_OP_NUMBER && k < (size_t)lambda_writer) lambda_list.push_back(ins); // a constant either side may read
        }
    }
    if (hit_list.empty() || lambda_list.empty()) return false;
    // the wavelength side must be closed: every number register it reads was written by one of its own instructions (a constant that
    // is not a NumberValue -- 2 * 3 left unfolded -- stands on the hit side only, and the program keeps the online form)
    {
        bool written[PYR_MAX_NUMBER_REGISTERS] = {};
        bool closed = true;
        auto reads = [&](const PyrOperand& o) {
            if (o.kind == PYR_OPERAND_REGISTER && !(o.bits < PYR_MAX_NUMBER_REGISTERS && written[o.bits])) closed = false;
        };
        auto reads_register = [&](uint32_t r) {
            if (!(r < PYR_MAX_NUMBER_REGISTERS && written[r])) closed = false;
        };
        for (const PyrInstr& ins : lambda_list) {
            switch (ins.op) {
            case PYR_OP_SPECTRUM: reads(ins.x); break;
            case PYR_OP_BLACKBODY: reads(ins.x), reads(ins.y); break;
            case PYR_OP_CLAMP: reads(ins.x), reads(ins.y), reads(ins.z); break;
            case PYR_OP_BINARY: reads_register(ins.a), reads_register(ins.b); break;
            case PYR_OP_MIX: reads(ins.x), reads_register(ins.a), reads_register(ins.b); break;
            default: break; // NumberValue
            }
            if (ins.output < PYR_MAX_NUMBER_REGISTERS) written[ins.output] = true;
        }
        if (!closed) return false;
    }
    chain.assign(outer_first.rbegin(), outer_first.rend());
    for (uint32_t reg : chain)
        if (reg >= PYR_MAX_NUMBER_REGISTERS) return false;
    hit = lambda = p;
    hit.first_instr = (uint32_t)instrs.size();
    hit.num_instrs = (uint32_t)hit_list.size();
    hit.output_reg = chain[0];
    instrs.insert(instrs.end(), hit_list.begin(), hit_list.end());
    lambda.first_instr = (uint32_t)instrs.size();
    lambda.num_instrs = (uint32_t)lambda_list.size();
    lambda.output_reg = instrs[(size_t)lambda_writer].output;
    instrs.insert(instrs.end(), lambda_list.begin(), lambda_list.end());
    return true;
}

} // namespace

struct PyrScene {
    int device = 0;
    int num_cus = 0;
    DevScene dev{};
    PyrBvhInfo info{};
    DeviceBuffer wide_nodes, wide_pair_nodes, pair_prims, nodes, prims, tri_shade, spheres, sphere_material, planes, plane_material, lamps, materials, components, programs, instrs, spectra,
        spectrum_data, rgb_basis, counters, tri_tex, sphere_tex_scale, plane_frames, textures, texture_data;
    PyrCounters last_counters{};
    bool have_counters = false;
    uint32_t* tail_count = nullptr; // device, kFeedBytes: the work-feed cursors of the intersect kernel
    DeviceBuffer tape; // spectral tape of the stage-scheduled kernel (grown on demand, kept between renders)
    DeviceBuffer tape_overflow; // one word the kernels set when a path outgrew the tape (checked after blocking renders and by pyr_scene_counters)
    ~PyrScene() {
        if (tail_count) (void)hipFree(tail_count);
    }
};

namespace pyr {
int api_fail(int code, const std::string& message) { return fail(code, message); }
int scene_device(const PyrScene* scene) { return scene->device; }
} // namespace pyr

namespace {

// world.rs:88-100 for a caller that did not pass plane_frames: basis(normal) (math.rs:98-123) and
// Matrix3::from_cols(binormal, tangent, normal).into() -- the f32 operations of oracle.cpp's ortho / normalize / cross /
// quat_from_cols in the same order (this file is built with -ffp-contract=off).
void plane_frame_from_normal(const float n[3], float q[4]) {
    auto cross = [](const float a[3], const float b[3], float o[3]) {
        o[0] = a[1] * b[2] - a[2] * b[1];
        o[1] = a[2] * b[0] - a[0] * b[2];
        o[2] = a[0] * b[1] - a[1] * b[0];
    };
    auto normalize = [](float v[3]) {
        float m = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        float k = 1.0f / m;
        v[0] *= k, v[1] *= k, v[2] *= k;
    };
    float unit[3] = {-n[1], n[0], 0.0f};
    if (std::fabs(n[0]) < 1.0e-4f)
        unit[0] = 1.0f, unit[1] = 0.0f, unit[2] = 0.0f;
    else if (std::fabs(n[1]) < 1.0e-4f)
        unit[0] = 0.0f, unit[1] = 1.0f, unit[2] = 0.0f;
    else if (std::fabs(n[2]) < 1.0e-4f)
        unit[0] = 0.0f, unit[1] = 0.0f, unit[2] = 1.0f;
    float z[3], y[3];
    cross(n, unit, z);
    normalize(z);
    cross(z, n, y);
    normalize(y);
    const float m00 = y[0], m01 = y[1], m02 = y[2], m10 = z[0], m11 = z[1], m12 = z[2], m20 = n[0], m21 = n[1], m22 = n[2];
    const float trace = m00 + m11 + m22;
    if (trace >= 0.0f) {
        float s = std::sqrt(1.0f + trace), w = 0.5f * s;
        s = 0.5f / s;
        q[0] = w, q[1] = (m12 - m21) * s, q[2] = (m20 - m02) * s, q[3] = (m01 - m10) * s;
    } else if (m00 > m11 && m00 > m22) {
        float s = std::sqrt((m00 - m11 - m22) + 1.0f), x = 0.5f * s;
        s = 0.5f / s;
        q[0] = (m12 - m21) * s, q[1] = x, q[2] = (m10 + m01) * s, q[3] = (m02 + m20) * s;
    } else if (m11 > m22) {
        float s = std::sqrt((m11 - m00 - m22) + 1.0f), yy = 0.5f * s;
        s = 0.5f / s;
        q[0] = (m20 - m02) * s, q[1] = (m10 + m01) * s, q[2] = yy, q[3] = (m21 + m12) * s;
    } else {
        float s = std::sqrt((m22 - m00 - m11) + 1.0f), zz = 0.5f * s;
        s = 0.5f / s;
        q[0] = (m01 - m10) * s, q[1] = (m02 + m20) * s, q[2] = (m21 + m12) * s, q[3] = zz;
    }
}

int pack_and_upload(const PyrSceneDesc* d, PyrScene* s) {
    // ---- primitives + BVH
    std::vector<PrimBounds> bounds;
    bounds.reserve((size_t)d->num_spheres + d->num_triangles);
    for (uint32_t i = 0; i < d->num_spheres; ++i) { // Bounded::aabb, shapes/mod.rs:411-416
        const float* p = d->spheres + 4 * (size_t)i;
        PrimBounds b;
        for (int a = 0; a < 3; ++a) {
            b.lo[a] = p[a] - p[3];
            b.hi[a] = p[a] + p[3];
        }
        b.shape = ((uint32_t)PYR_SHAPE_SPHERE << 30) | i;
        bounds.push_back(b);
    }
    for (uint32_t i = 0; i < d->num_triangles; ++i) { // shapes/mod.rs:417-428
        const float* p = d->tri_positions + 9 * (size_t)i;
        PrimBounds b;
        for (int a = 0; a < 3; ++a) {
            b.lo[a] = std::min(p[a], std::min(p[3 + a], p[6 + a]));
            b.hi[a] = std::max(p[a], std::max(p[3 + a], p[6 + a]));
        }
        b.shape = ((uint32_t)PYR_SHAPE_TRIANGLE << 30) | i;
        bounds.push_back(b);
    }
    // exact_math.h: `normalize` multiplies by rcp32(sqrt32(|v|^2)), which is the correctly rounded IEEE result (what the reference
    // computes) only while lengths and squared lengths stay normal f32 numbers. Coordinates are the user's units, so a scene whose
    // extent leaves the verified range is refused instead of rendered differently from the reference (pyrite_gpu.h "Coordinates").
    constexpr float kMaxCoordinate = 1.0e15f;
    for (const PrimBounds& b : bounds)
        for (int a = 0; a < 3; ++a)
            if (!(std::fabs(b.lo[a]) <= kMaxCoordinate && std::fabs(b.hi[a]) <= kMaxCoordinate))
                return fail(PYR_ERR_UNSUPPORTED, "a primitive lies beyond 1e15 units from the origin (or is not finite): outside the range the kernels' arithmetic is verified for");
    // Leaves are tested in pairs only by the four-child pair tree: a triangle-only scene too big to live in LDS (its
    // primitives alone outgrow the 8 KB the LDS-resident walk allows) with neither tree switched off.
    const char* wide_switch = std::getenv("PYRITE_WIDE_BVH");
    const char* pair_switch = std::getenv("PYRITE_PAIR_PRIMS");
    const bool pair_tree_expected = d->num_spheres == 0 && bounds.size() * 48 > 8 * 1024 && !(wide_switch && wide_switch[0] == '0') && !(pair_switch && pair_switch[0] == '0');
    BuiltBvh bvh = build_bvh(bounds, pair_tree_expected);
    if (bvh.max_depth > 96) return fail(PYR_ERR_UNSUPPORTED, "BVH deeper than the LDS traversal stack allows");

    std::vector<DevPrim> prims(bvh.prim_order.size());
    for (size_t k = 0; k < prims.size(); ++k) {
        uint32_t shape = bvh.prim_order[k], index = shape & 0x3FFFFFFFu;
        DevPrim& o = prims[k];
        std::memset(&o, 0, sizeof(o));
        if ((shape >> 30) == PYR_SHAPE_TRIANGLE) {
            const float* p = d->tri_positions + 9 * (size_t)index;
            for (int a = 0; a < 3; ++a) {
                o.a[a] = p[a];
                o.b[a] = p[3 + a] - p[a]; // edge1 = v2 - v1, edge2 = v3 - v1 (world.rs:330-331, shapes/mod.rs:338-339)
                o.c[a] = p[6 + a] - p[a];
            }
        } else {
            const float* p = d->spheres + 4 * (size_t)index;
            for (int a = 0; a < 3; ++a) o.a[a] = p[a];
            o.b[0] = p[3];
        }
        o.a[3] = shape_bits(shape);
    }
    std::vector<DevTriShade> shade(d->num_triangles);
    for (uint32_t i = 0; i < d->num_triangles; ++i) {
        const float* n = d->tri_normals + 9 * (size_t)i;
        DevTriShade& o = shade[i];
        std::memset(&o, 0, sizeof(o));
        for (int a = 0; a < 3; ++a) {
            o.n1[a] = n[a];
            o.n2[a] = n[3 + a];
            o.n3[a] = n[6 + a];
        }
        o.n1[3] = bits_to_float(d->tri_material[i]);
    }
    std::vector<DevLamp> lamps(d->num_lamps);
    for (uint32_t i = 0; i < d->num_lamps; ++i) {
        const PyrLamp& l = d->lamps[i];
        DevLamp& o = lamps[i];
        std::memset(&o, 0, sizeof(o));
        o.kind = l.kind;
        o.shape_kind = l.shape_kind;
        o.shape_index = l.shape_index;
        o.color_program = l.color_program;
        for (int a = 0; a < 3; ++a) o.v[a] = l.v[a];
        o.width = l.width;
        if (l.kind == PYR_LAMP_SHAPE && l.shape_kind == PYR_SHAPE_SPHERE) {
            const float* p = d->spheres + 4 * (size_t)l.shape_index;
            for (int a = 0; a < 3; ++a) o.v[a] = p[a];
            o.width = p[3];
            o.area = p[3] * p[3] * 4.0f * 3.14159265358979323846f; // Shape::surface_area, shapes/mod.rs:275
            o.material = d->sphere_material[l.shape_index];
            o.t1[0] = d->sphere_tex_scale ? d->sphere_tex_scale[2 * (size_t)l.shape_index] : 1.0f;
            o.t1[1] = d->sphere_tex_scale ? d->sphere_tex_scale[2 * (size_t)l.shape_index + 1] : 1.0f;
        } else if (l.kind == PYR_LAMP_SHAPE) {
            const float* p = d->tri_positions + 9 * (size_t)l.shape_index;
            const float* n = d->tri_normals + 9 * (size_t)l.shape_index;
            for (int a = 0; a < 3; ++a) {
                o.p1[a] = p[a];
                o.p2[a] = p[3 + a];
                o.p3[a] = p[6 + a];
                o.n1[a] = n[a];
                o.n2[a] = n[3 + a];
                o.n3[a] = n[6 + a];
            }
            // 0.5 * |a x b| (shapes/mod.rs:276-285), evaluated in f32 without fusing
            volatile float ax = p[3] - p[0], ay = p[4] - p[1], az = p[5] - p[2];
            volatile float bx = p[6] - p[0], by = p[7] - p[1], bz = p[8] - p[2];
            volatile float m1 = ay * bz, m2 = az * by, m3 = az * bx, m4 = ax * bz, m5 = ax * by, m6 = ay * bx;
            volatile float cx = m1 - m2, cy = m3 - m4, cz = m5 - m6;
            volatile float xx = cx * cx, yy = cy * cy, zz = cz * cz;
            volatile float s1 = xx + yy;
            volatile float s2 = s1 + zz;
            o.area = 0.5f * std::sqrt(s2);
            o.material = d->tri_material[l.shape_index];
            if (d->tri_uvs) {
                const float* uv = d->tri_uvs + 6 * (size_t)l.shape_index;
                o.t1[0] = uv[0], o.t1[1] = uv[1], o.t2[0] = uv[2], o.t2[1] = uv[3], o.t3[0] = uv[4], o.t3[1] = uv[5];
            }
        }
    }
    std::vector<DevProgram> programs(d->num_programs);
    for (uint32_t i = 0; i < d->num_programs; ++i) programs[i] = pack_program(d->instrs, d->programs[i]);
    // programs without a tape form that factor into a hit side and a wavelength side get both as programs of their own, behind the
    // caller's (TAPE_FORM_PRODUCT); the instruction array grows by their instructions
    std::vector<PyrInstr> instrs(d->instrs, d->instrs + d->num_instrs);
    for (uint32_t i = 0; i < d->num_programs; ++i) {
        if (programs[i].kind != PYR_PROGRAM_INSTRUCTIONS || programs[i].tape_form != TAPE_FORM_NONE || programs.size() + 2 > 128) continue; // (a hit tape takes at most 128 programs)
        const size_t instrs_before = instrs.size();
        PyrProgram hit, lambda;
        std::vector<uint32_t> chain;
        if (!split_product(instrs, d->programs[i], hit, lambda, chain)) continue;
        const DevProgram dev_hit = pack_program(instrs.data(), hit), dev_lambda = pack_program(instrs.data(), lambda);
        if (dev_hit.tape_form != TAPE_FORM_HIT_VALUE || !(dev_lambda.tape_form == TAPE_FORM_LAMBDA || dev_lambda.fast != FAST_NONE)) {
            instrs.resize(instrs_before);
            continue;
        }
        programs[i].tape_form = TAPE_FORM_PRODUCT;
        uint32_t packed = (uint32_t)programs.size() | (((uint32_t)programs.size() + 1u) << 8) | ((uint32_t)chain.size() << 16); // device_scene.h DevProgram::tape_rgb_reg
        for (size_t c = 0; c < chain.size(); ++c) packed |= chain[c] << (20u + 4u * (uint32_t)c); // PYR_MAX_NUMBER_REGISTERS == 16
        programs[i].tape_rgb_reg = packed;
        programs.push_back(dev_hit);
        programs.push_back(dev_lambda);
    }

    bool needs_interpreter = false, uses_textures = false;
    for (const DevProgram& pr : programs)
        if (pr.kind == PYR_PROGRAM_INSTRUCTIONS && pr.fast == FAST_NONE) needs_interpreter = true;
    for (uint32_t i = 0; i < d->num_instrs; ++i)
        if (d->instrs[i].op == PYR_OP_COLOR_TEXTURE || d->instrs[i].op == PYR_OP_MONO_TEXTURE) uses_textures = true;
    for (uint32_t i = 0; i < d->num_materials; ++i)
        if (d->materials[i].normal_map_program >= 0) uses_textures = needs_interpreter = true;

    // texture space: only the interpreter builds of the kernels read it
    std::vector<DevTriTex> tri_tex;
    std::vector<float> sphere_scale, plane_frames;
    std::vector<DevTexture> textures(d->num_textures);
    if (needs_interpreter) {
        tri_tex.resize(d->num_triangles);
        for (uint32_t i = 0; i < d->num_triangles; ++i) {
            DevTriTex& o = tri_tex[i];
            std::memset(&o, 0, sizeof(o));
            if (d->tri_uvs) {
                const float* uv = d->tri_uvs + 6 * (size_t)i;
                for (int a = 0; a < 4; ++a) o.uv12[a] = uv[a];
                o.uv3[0] = uv[4], o.uv3[1] = uv[5];
            }
            o.f1[0] = o.f2[0] = o.f3[0] = 1.0f; // identity
            if (d->tri_frames) {
                const float* f = d->tri_frames + 12 * (size_t)i;
                for (int a = 0; a < 4; ++a) o.f1[a] = f[a], o.f2[a] = f[4 + a], o.f3[a] = f[8 + a];
            }
        }
        sphere_scale.assign(2 * (size_t)d->num_spheres, 1.0f);
        if (d->sphere_tex_scale) sphere_scale.assign(d->sphere_tex_scale, d->sphere_tex_scale + 2 * (size_t)d->num_spheres);
        plane_frames.resize(4 * (size_t)d->num_planes);
        for (uint32_t i = 0; i < d->num_planes; ++i) {
            if (d->plane_frames) {
                for (int a = 0; a < 4; ++a) plane_frames[4 * (size_t)i + a] = d->plane_frames[4 * (size_t)i + a];
            } else {
                plane_frame_from_normal(d->planes + 8 * (size_t)i + 3, &plane_frames[4 * (size_t)i]);
            }
        }
    }
    for (uint32_t i = 0; i < d->num_textures; ++i)
        textures[i] = DevTexture{d->textures[i].format == PYR_TEXTURE_COLOR ? 4u : 1u, d->textures[i].width, d->textures[i].height, 0u, d->textures[i].offset};

    int rc;
    if ((rc = s->tri_tex.upload(tri_tex.data(), tri_tex.size() * sizeof(DevTriTex)))) return rc;
    if ((rc = s->sphere_tex_scale.upload(sphere_scale.data(), sphere_scale.size() * 4))) return rc;
    if ((rc = s->plane_frames.upload(plane_frames.data(), plane_frames.size() * 4))) return rc;
    if ((rc = s->textures.upload(textures.data(), textures.size() * sizeof(DevTexture)))) return rc;
    if ((rc = s->texture_data.upload(d->texture_data, d->num_textures ? (size_t)d->num_texture_floats * 4 : 0))) return rc;
    // the traversal addresses a node by a 32-bit byte offset from the tree's base (one SGPR pair + one VGPR per load): 4 GB of 64-byte
    // binary nodes, 4 GB of 128-byte wide nodes -- about 200 M triangles, beyond which the call says so instead of wrapping around
    if (bvh.nodes.size() >= (1ull << 26)) return fail(PYR_ERR_UNSUPPORTED, "scene too large: the acceleration structure has 2^26 nodes or more (4 GB)");
    if ((rc = s->nodes.upload(bvh.nodes.data(), bvh.nodes.size() * sizeof(Node64)))) return rc;
    // scenes that do not live in LDS also get the 4-wide tree for the resumable traversal (latency bound there);
    // PYRITE_WIDE_BVH=0 keeps the binary tree (A/B)
    WideBvh wide;
    const char* wide_env = std::getenv("PYRITE_WIDE_BVH");
    const bool want_wide = (size_t)bvh.nodes.size() * 64 + prims.size() * 48 > 8 * 1024 && !(wide_env && wide_env[0] == '0');
    if (want_wide) {
        wide = collapse_to_wide(bvh);
        if (wide.stack_need > kMaxStackDepth || wide.nodes.size() >= (1ull << 25)) wide = WideBvh{}; // too deep, or past 4 GB: the binary tree
    }
    // Triangle pairs for the wide tree's leaves (device_scene.h DevPrimPair): every leaf gets ceil(n / 2) records of its own and
    // its code in the wide nodes is rewritten to count in records. PYRITE_PAIR_PRIMS=0 keeps the one-primitive records (A/B).
    std::vector<DevPrimPair> pairs;
    std::vector<Node128> pair_nodes;
    const char* pair_env = std::getenv("PYRITE_PAIR_PRIMS");
    if (!wide.nodes.empty() && d->num_spheres == 0 && !(pair_env && pair_env[0] == '0')) {
        pair_nodes = wide.nodes;
        for (Node128& node : pair_nodes)
            for (int k = 0; k < 4; ++k) {
                const int32_t code = node.child[k];
                if (code >= 0 || code == kEmptyChild) continue;
                const uint32_t first = (uint32_t)(-1 - code) >> 3, count = (uint32_t)(-1 - code) & 7u;
                node.child[k] = encode_leaf((uint32_t)pairs.size(), count);
                if (count == 0) { // an empty leaf still names a record (trav_step_lean loads it before it looks at the count): one no triangle passes
                    DevPrimPair none;
                    std::memset(&none, 0, sizeof(none));
                    none.q[1][2] = none.q[1][3] = shape_bits(PYR_HIT_NONE);
                    pairs.push_back(none);
                }
                for (uint32_t j = 0; j < count; j += 2) {
                    DevPrimPair pr;
                    std::memset(&pr, 0, sizeof(pr));
                    for (uint32_t h = 0; h < 2; ++h) {
                        if (j + h >= count) {
                            pr.q[1][2 + h] = shape_bits(PYR_HIT_NONE);
                            continue;
                        }
                        const DevPrim& t = prims[first + j + h];
                        pr.q[0][0 + h] = t.a[0], pr.q[0][2 + h] = t.a[1], pr.q[1][0 + h] = t.a[2], pr.q[1][2 + h] = t.a[3];
                        pr.q[2][0 + h] = t.b[0], pr.q[2][2 + h] = t.b[1], pr.q[3][0 + h] = t.b[2];
                        pr.q[3][2 + h] = t.c[0], pr.q[4][0 + h] = t.c[1], pr.q[4][2 + h] = t.c[2];
                    }
                    pairs.push_back(pr);
                }
            }
    }
    if ((rc = s->pair_prims.upload(pairs.data(), pairs.size() * sizeof(DevPrimPair)))) return rc;
    if ((rc = s->wide_pair_nodes.upload(pair_nodes.data(), pair_nodes.size() * sizeof(Node128)))) return rc;
    if ((rc = s->wide_nodes.upload(wide.nodes.data(), wide.nodes.size() * sizeof(Node128)))) return rc;
    if ((rc = s->prims.upload(prims.data(), prims.size() * sizeof(DevPrim)))) return rc;
    if ((rc = s->tri_shade.upload(shade.data(), shade.size() * sizeof(DevTriShade)))) return rc;
    if ((rc = s->spheres.upload(d->spheres, (size_t)d->num_spheres * 16))) return rc;
    if ((rc = s->sphere_material.upload(d->sphere_material, (size_t)d->num_spheres * 4))) return rc;
    if ((rc = s->planes.upload(d->planes, (size_t)d->num_planes * 32))) return rc;
    if ((rc = s->plane_material.upload(d->plane_material, (size_t)d->num_planes * 4))) return rc;
    if ((rc = s->lamps.upload(lamps.data(), lamps.size() * sizeof(DevLamp)))) return rc;
    if ((rc = s->materials.upload(d->materials, (size_t)d->num_materials * sizeof(PyrMaterial)))) return rc;
    if ((rc = s->components.upload(d->components, (size_t)d->num_components * sizeof(PyrComponent)))) return rc;
    if ((rc = s->programs.upload(programs.data(), programs.size() * sizeof(DevProgram)))) return rc;
    if ((rc = s->instrs.upload(instrs.data(), instrs.size() * sizeof(PyrInstr)))) return rc;
    if ((rc = s->spectra.upload(d->spectra, (size_t)d->num_spectra * sizeof(PyrSpectrum)))) return rc;
    if ((rc = s->spectrum_data.upload(d->spectrum_data, (size_t)d->num_spectrum_floats * 4))) return rc;
    if ((rc = s->rgb_basis.upload(d->rgb_basis, d->rgb_basis ? (size_t)d->rgb_basis_count * 12 : 0))) return rc;
    if ((rc = s->counters.alloc(sizeof(PyrCounters)))) return rc;

    DevScene& v = s->dev;
    v.nodes = (const float*)s->nodes.ptr;
    v.wide_nodes = wide.nodes.empty() ? nullptr : (const float*)s->wide_nodes.ptr;
    v.wide_stack_depth = std::max(1u, wide.stack_need);
    v.pair_prims = pairs.empty() ? nullptr : (const float*)s->pair_prims.ptr;
    v.wide_pair_nodes = pairs.empty() ? nullptr : (const float*)s->wide_pair_nodes.ptr;
    v.prims = (const float*)s->prims.ptr;
    v.tri_shade = (const float*)s->tri_shade.ptr;
    v.spheres = (const float*)s->spheres.ptr;
    v.sphere_material = (const uint32_t*)s->sphere_material.ptr;
    v.planes = (const float*)s->planes.ptr;
    v.plane_material = (const uint32_t*)s->plane_material.ptr;
    v.lamps = (const DevLamp*)s->lamps.ptr;
    v.materials = (const PyrMaterial*)s->materials.ptr;
    v.components = (const PyrComponent*)s->components.ptr;
    v.programs = (const DevProgram*)s->programs.ptr;
    v.instrs = (const PyrInstr*)s->instrs.ptr;
    v.spectra = (const PyrSpectrum*)s->spectra.ptr;
    v.spectrum_data = (const float*)s->spectrum_data.ptr;
    v.rgb_basis = (const float*)s->rgb_basis.ptr;
    v.num_planes = d->num_planes;
    v.num_lamps = d->num_lamps;
    v.rgb_count = d->rgb_basis ? d->rgb_basis_count : 0;
    v.rgb_min = d->rgb_basis_min;
    v.rgb_max = d->rgb_basis_max;
    v.sky_program = d->sky_program;
    v.stack_depth = std::max(1u, bvh.max_depth);
    v.num_nodes = (uint32_t)bvh.nodes.size();
    v.num_prims = (uint32_t)prims.size();
    v.num_spectra = d->num_spectra;
    v.num_programs = (uint32_t)programs.size();
    v.num_spectrum_floats = d->num_spectrum_floats;
    v.num_materials = d->num_materials;
    v.num_components = d->num_components;
    {
        // the small tables the kernels stage into LDS (kernels.hip stage_tables): spectra + the material / component / program /
        // lamp records. <= 16 KB, and only for scenes too big to live in LDS themselves: a small scene leaves L1 to the tables
        // (C2: staging the spectra costs a workgroup per CU and is 0.9x), a big one evicts them all the time (C3: 1.33x)
        const uint64_t floats = (uint64_t)d->num_spectra * (sizeof(PyrSpectrum) / 4) + d->num_spectrum_floats + (uint64_t)d->num_materials * (sizeof(PyrMaterial) / 4) +
                                (uint64_t)d->num_components * (sizeof(PyrComponent) / 4) + (uint64_t)programs.size() * (sizeof(DevProgram) / 4) +
                                (uint64_t)d->num_lamps * (sizeof(DevLamp) / 4);
        const bool big_scene = (size_t)bvh.nodes.size() * 64 + prims.size() * 48 > 8 * 1024;
        v.lds_table_floats = (floats <= 4096 && big_scene) ? (uint32_t)floats : 0;
    }
    v.needs_interpreter = needs_interpreter ? 1u : 0u;
    v.uses_textures = uses_textures ? 1u : 0u;
    // Round 4: the spectral tape for scenes WITH interpreter programs (device_scene.h TapeForm). Every colour program -- of a
    // component, of a lamp, the sky -- must have a tape form; the value slots the replay keeps in LDS (kTapeEagerSlots = 8: one
    // holds 1.0, three the RGB basis when a HIT_RGB program exists) must hold every spectrum-reading fast program (counted here
    // without the sharing the kernel finds, so never fewer), and the prepared programs must fit their LDS table (128).
    v.hit_tape = v.rgb_records = v.micro_records = v.product_records = 0;
    // value slots of the replay: one per LAMBDA program and one per DISTINCT fast shape -- programs of the same shape, factor and
    // spectrum share a slot (kernels.hip prepare_tape_tables `alike`: C3's three white walls are three programs over one spectrum)
    uint32_t fast_programs = 0;
    {
        for (size_t i = 0; i < programs.size(); ++i) {
            const DevProgram& pr = programs[i];
            if (pr.kind != PYR_PROGRAM_INSTRUCTIONS) continue;
            if (pr.tape_form == TAPE_FORM_LAMBDA) {
                fast_programs += 1u;
            } else if (pr.fast != FAST_NONE) {
                bool seen = false;
                for (size_t j = 0; j < i && !seen; ++j) {
                    const DevProgram& q = programs[j];
                    seen = q.kind == PYR_PROGRAM_INSTRUCTIONS && q.fast == pr.fast && q.fast_spectrum == pr.fast_spectrum && std::memcmp(&q.fast_scale, &pr.fast_scale, sizeof(float)) == 0;
                }
                fast_programs += seen ? 0u : 1u;
            }
        }
    }
    if (needs_interpreter) {
        bool ok = programs.size() <= 128;
        auto colour = [&](uint32_t id) {
            if (id >= programs.size()) return;
            const DevProgram& pr = programs[id];
            if (pr.kind != PYR_PROGRAM_INSTRUCTIONS) return;
            if (pr.tape_form == TAPE_FORM_NONE) ok = false;
            if (pr.tape_form == TAPE_FORM_HIT_RGB) v.rgb_records = 1u;
            if (pr.tape_form == TAPE_FORM_HIT_RGB || pr.tape_form == TAPE_FORM_PRODUCT) v.micro_records = 1u;
            if (pr.tape_form == TAPE_FORM_PRODUCT) v.product_records = 1u;
        };
        for (uint32_t i = 0; i < d->num_components; ++i) colour(d->components[i].color_program);
        for (const DevLamp& l : lamps) colour(l.color_program);
        colour(d->sky_program);
        if (tape_rows_needed(fast_programs, v.rgb_records != 0) > kTapeMaxValueRows) ok = false; // (counted here without LAMBDA's hit-tape condition: never fewer than the kernel finds)
        const char* off = std::getenv("PYRITE_HIT_TAPE"); // A/B and tests: PYRITE_HIT_TAPE=0 keeps the online form (read at scene creation)
        if (off && off[0] == '0') ok = false;
        v.hit_tape = ok ? 1u : 0u;
        if (!ok) v.rgb_records = v.micro_records = v.product_records = 0u;
    }
    // the eager replay's value rows (device_scene.h): eight, or what the scene's programs need; past sixteen the replay looks values up record by record
    v.tape_value_rows = tape_rows_needed(fast_programs, v.rgb_records != 0) <= kTapeMaxValueRows ? tape_rows_needed(fast_programs, v.rgb_records != 0) : kTapeValueRows;
    v.shadow_margin = d->num_spheres != 0 ? 1.01f : 1.001f; // device_scene.h
    v.hero_only_records = 0;
    for (uint32_t i = 0; i < d->num_materials; ++i)
        for (uint32_t k = 0; k < d->materials[i].num_emissive; ++k) {
            const int probability = d->components[d->materials[i].first_emissive + k].probability_program;
            if (probability >= 0 && programs[(size_t)probability].reads_wavelength) v.hero_only_records = 1u;
        }
    v.tri_tex = (const float*)s->tri_tex.ptr;
    v.sphere_tex_scale = (const float*)s->sphere_tex_scale.ptr;
    v.plane_frames = (const float*)s->plane_frames.ptr;
    v.textures = (const DevTexture*)s->textures.ptr;
    v.texture_data = (const float*)s->texture_data.ptr;

    s->info.num_nodes = (uint32_t)bvh.nodes.size();
    s->info.num_leaves = bvh.num_leaves;
    s->info.max_depth = bvh.max_depth;
    s->info.num_primitives = (uint32_t)prims.size();
    s->info.node_bytes = bvh.nodes.size() * sizeof(Node64);
    s->info.primitive_bytes = prims.size() * sizeof(DevPrim);
    s->info.num_wide_nodes = (uint32_t)wide.nodes.size();
    s->info.num_pair_records = (uint32_t)pairs.size();
    s->info.wide_node_bytes = wide.nodes.size() * sizeof(Node128);
    s->info.pair_record_bytes = pairs.size() * sizeof(DevPrimPair);
    return PYR_OK;
}

struct TilePlan {
    uint32_t tiles_x = 0, tiles_y = 0;
    uint32_t tile_begin = 0, tile_stride = 1, tile_count = 0;
    uint32_t chunks_per_tile = 0;
    uint32_t chunk_begin = 0, chunk_end = 0; // chunk numbers of the call's tiles (device_scene.h RenderLaunch)
};

int plan_tiles(const PyrFilmDesc* film, const PyrRenderParams* p, TilePlan& plan) {
    // make_tiles, renderer/algorithm.rs:158-166
    const uint32_t ts = p->tile_size;
    plan.tiles_x = (film->width + ts - 1) / ts;
    plan.tiles_y = (film->height + ts - 1) / ts;
    const uint64_t total = (uint64_t)plan.tiles_x * plan.tiles_y;
    const uint64_t begin = p->tile_begin, end = p->tile_end ? p->tile_end : total;
    if (total >= 0xFFFFFFFFull || begin > end || end > total) return fail(PYR_ERR_INVALID_ARGUMENT, "tile range out of bounds");
    plan.tile_begin = (uint32_t)begin;
    plan.tile_stride = std::max(1u, p->tile_stride);
    plan.tile_count = (uint32_t)((end - begin + plan.tile_stride - 1) / plan.tile_stride);
    const uint64_t per_tile = ((uint64_t)ts * ts * p->pixel_samples + 63) / 64; // iterations of a full tile: simple.rs:73
    if (per_tile * plan.tile_count >= 0xFFFFFFFFull) return fail(PYR_ERR_UNSUPPORTED, "too many samples for one call: more than 2^32 chunks");
    plan.chunks_per_tile = (uint32_t)std::max<uint64_t>(per_tile, 1); // pixel_samples == 0: no chunk at all
    plan.chunk_begin = 0;
    plan.chunk_end = (uint32_t)(per_tile * plan.tile_count);
    return PYR_OK;
}

uint64_t film_buffer_grains(const PyrFilmDesc* film, const PyrRenderParams* p, const TilePlan& plan) {
    if (p->film_layout == PYR_FILM_TILE_BLOCKS) return (uint64_t)plan.tile_count * (p->tile_size + 2ull) * (p->tile_size + 2ull) * film->bins;
    const uint32_t rows = p->film_row_count ? p->film_row_count : film->height;
    return (uint64_t)rows * film->width * film->bins;
}

int check_render_args(PyrScene* scene, const PyrCamera* camera, const PyrFilmDesc* film, const PyrRenderParams* p, const void* film_ptr) {
    if (!scene || !camera || !film || !p || !film_ptr) return fail(PYR_ERR_INVALID_ARGUMENT, "null argument");
    if (film->width == 0 || film->height == 0 || film->bins == 0 || p->tile_size == 0 || p->spectrum_samples == 0)
        return fail(PYR_ERR_INVALID_ARGUMENT, "zero-sized parameter");
    if (!(film->wl_width > 0.0f)) return fail(PYR_ERR_INVALID_ARGUMENT, "empty wavelength span");
    uint32_t rows = p->film_row_count ? p->film_row_count : film->height;
    if ((uint64_t)p->film_row_begin + rows > film->height) return fail(PYR_ERR_INVALID_ARGUMENT, "film window exceeds the image");
    if (p->spectrum_samples > 64) return fail(PYR_ERR_UNSUPPORTED, "spectrum_samples > 64");
    // a pixel is a 32-bit index into the call's film buffer (0xFFFFFFFF stands for "none"): 65,535 x 65,535 still fits
    if ((uint64_t)film->width * film->height >= 0xFFFFFFFFull) return fail(PYR_ERR_UNSUPPORTED, "image too large: 2^32 pixels or more");
    if (p->film_layout == PYR_FILM_TILE_BLOCKS) {
        const uint64_t tiles = (uint64_t)((film->width + p->tile_size - 1) / p->tile_size) * ((film->height + p->tile_size - 1) / p->tile_size);
        if (tiles * (p->tile_size + 2ull) * (p->tile_size + 2ull) >= 0xFFFFFFFFull) return fail(PYR_ERR_UNSUPPORTED, "image too large for a film of tile blocks");
    }
    if (p->film_layout > PYR_FILM_TILE_BLOCKS) return fail(PYR_ERR_INVALID_ARGUMENT, "unknown film layout");
    if (p->film_layout == PYR_FILM_TILE_BLOCKS && (p->film_row_begin || p->film_row_count))
        return fail(PYR_ERR_INVALID_ARGUMENT, "a film of tile blocks has no row window");
    return PYR_OK;
}

// Scheduler choice (kernels.hip): the bounce-synchronous walk wins when the scene lives in LDS and traversal is cheap
// (C2: 573 vs 300 Msamples/s); the stage scheduler wins when traversal lengths are heavy tailed (C3: 110 vs 91).
// PYRITE_SCHEDULER=sync|sm overrides; PYRITE_SM_LANES / PYRITE_SM_STEPS tune the stage scheduler.

// The spectral tape ([tape_max_ops][tape_lanes] 8-byte records) and the overflow word, kept on the scene between renders.
int reserve_tape(PyrScene* scene, RenderLaunch& L, hipStream_t stream) {
    const size_t bytes = (size_t)L.tape_lanes * L.tape_max_ops * sizeof(unsigned long long);
    if (bytes > scene->tape.bytes) {
        HIP_TRY(hipStreamSynchronize(stream)); // an earlier render of this scene may still be reading the old tape
        scene->tape.release();
        int rc = scene->tape.alloc(bytes);
        if (rc != PYR_OK) return rc;
    }
    L.tape = (unsigned long long*)scene->tape.ptr;
    if (!scene->tape_overflow.ptr) {
        int rc = scene->tape_overflow.alloc(sizeof(uint32_t));
        if (rc != PYR_OK) return rc;
        // cleared on the stream the render is enqueued on: a null-stream memset is not ordered against a kernel on a
        // hipStreamNonBlocking stream (the multi-device entries use such streams)
        HIP_TRY(hipMemsetAsync(scene->tape_overflow.ptr, 0, sizeof(uint32_t), stream));
    }
    L.tape_overflow = (uint32_t*)scene->tape_overflow.ptr;
    return PYR_OK;
}

// Which schedule a render of this scene runs, and with which phase thresholds (defaults and the development switches).
void choose_schedule(const PyrScene* scene, RenderLaunch& L) {
    const char* e = std::getenv("PYRITE_SCHEDULER");
    if (e && std::string(e) == "sm")
        L.scheduler = 1;
    else if (e && std::string(e) == "sync")
        L.scheduler = 0;
    else // the synchronous walk for scenes that live in LDS -- unless they run interpreter programs: the stage scheduler keeps the
         // interpreter in line and memoised (spheres example 572 -> 737, lamps 549 -> 724 Msamples/s against the synchronous walk)
        L.scheduler = scene_is_lds_resident(scene->dev) && scene->dev.needs_interpreter == 0 ? 0u : 1u;
    // the program interpreter (and with it texture coordinates and normal maps) lives in the resumable integrator (Walker), in
    // line; the synchronous walk is built without it: PYRITE_SCHEDULER=sync on such a scene runs the stage scheduler
    if (L.scheduler == 0 && scene->dev.needs_interpreter != 0) L.scheduler = 1;
    const char* lanes = std::getenv("PYRITE_SM_LANES");
    const char* steps = std::getenv("PYRITE_SM_STEPS");
    // lanes that make a phase run: 16 on the BASELINE meshes (swept in rounds 2 and 3); 32 where the phases are heavy and the rays
    // short -- scenes that run the program interpreter (round 4, every example scene of the reference: textures 755 -> 859, spheres
    // 887 -> 969, lamps 726 -> 812, diamonds 543 -> 564 Msamples/s; flat from 28 to 48)
    L.sm_phase_lanes = lanes && *lanes ? (uint32_t)std::strtoul(lanes, nullptr, 10) : (scene->dev.needs_interpreter != 0 ? 32u : 16u);
    L.sm_trav_steps = steps && *steps ? (uint32_t)std::strtoul(steps, nullptr, 10) : 8u;
    const char* expose = std::getenv("PYRITE_SM_EXPOSE_LANES");
    L.sm_expose_lanes = expose && *expose ? (uint32_t)std::strtoul(expose, nullptr, 10) : L.sm_phase_lanes;
}

int render_batches(PyrScene* scene, RenderLaunch L, bool count, hipStream_t stream) {
    if (L.chunk_end == L.chunk_begin) return PYR_OK;
    choose_schedule(scene, L);
    if (L.scheduler != 0 && (scene->dev.needs_interpreter == 0 || uses_hit_tape(scene->dev, L))) {
        L.tape_lanes = tape_lanes_bound(scene->num_cus);
        L.tape_max_ops = tape_ops_bound(scene->dev, L);
        int rc = reserve_tape(scene, L, stream);
        if (rc != PYR_OK) return rc;
    }
    int rc = launch_render(scene->dev, L, count, stream, scene->num_cus);
    if (rc != PYR_OK) return fail(rc, kernels_last_error());
    return PYR_OK;
}

// After a render has been waited for: did a path outgrow the spectral tape (kernels.hip tape_push)? The film is wrong then.
int check_tape_overflow(PyrScene* scene) {
    if (!scene->tape_overflow.ptr) return PYR_OK;
    uint32_t word = 0;
    HIP_TRY(hipMemcpy(&word, scene->tape_overflow.ptr, sizeof(word), hipMemcpyDeviceToHost));
    if (word == 0) return PYR_OK;
    HIP_TRY(hipMemset(scene->tape_overflow.ptr, 0, sizeof(uint32_t)));
    if (word == 2) return fail(PYR_ERR_DEVICE, "a wave gave up waiting on its workgroup's LDS queues (spin limit): the film of that render is invalid");
    return fail(PYR_ERR_DEVICE, "a path appended more records than the spectral tape's bound allows: the film of that render is invalid");
}

} // namespace

namespace pyr {
// For the translation units that enqueue renders without waiting for them (multi.cpp): the word itself, and its check.
uint32_t* scene_overflow_word(PyrScene* scene) { return scene ? (uint32_t*)scene->tape_overflow.ptr : nullptr; }
int scene_check_overflow(PyrScene* scene) { return check_tape_overflow(scene); }
} // namespace pyr

namespace {

RenderLaunch make_launch(const PyrCamera* camera, const PyrFilmDesc* film, const PyrRenderParams* p, const TilePlan& plan) {
    RenderLaunch L{};
    L.camera = *camera;
    L.film = *film;
    L.bounces = p->bounces;
    L.light_samples = p->light_samples;
    L.spectrum_samples = p->spectrum_samples;
    L.tile_size = p->tile_size;
    L.pixel_samples = p->pixel_samples;
    L.tiles_x = plan.tiles_x;
    L.tiles_y = plan.tiles_y;
    L.tile_begin = plan.tile_begin;
    L.tile_stride = plan.tile_stride;
    L.tile_count = plan.tile_count;
    L.chunks_per_tile = plan.chunks_per_tile;
    L.chunk_begin = plan.chunk_begin;
    L.chunk_end = plan.chunk_end;
    L.film_layout = p->film_layout;
    L.film_row_begin = p->film_row_begin;
    L.film_row_count = p->film_row_count ? p->film_row_count : film->height;
    L.seed = p->seed;
    L.grains_per_wavelength = (float)film->bins / film->wl_width; // film.rs:38
    return L;
}

} // namespace

extern "C" {

int pyr_abi_version(void) { return PYR_ABI_VERSION; }

int pyr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* pyr_last_error(void) { return g_error.c_str(); }

int pyr_scene_create(const PyrSceneDesc* desc, int device, PyrScene** out_scene) {
    if (!out_scene) return fail(PYR_ERR_INVALID_ARGUMENT, "null out pointer");
    *out_scene = nullptr;
    int rc = validate(desc);
    if (rc != PYR_OK) return rc;
    int n = pyr_device_count();
    if (n <= 0) return fail(PYR_ERR_DEVICE, "no HIP device is visible; pyrite_gpu has no CPU path");
    if (device < 0 || device >= n) return fail(PYR_ERR_INVALID_ARGUMENT, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    std::unique_ptr<PyrScene> s(new PyrScene());
    s->device = device;
    s->num_cus = prop.multiProcessorCount;
    rc = pack_and_upload(desc, s.get());
    if (rc != PYR_OK) return rc;
    *out_scene = s.release();
    return PYR_OK;
}

void pyr_scene_destroy(PyrScene* scene) {
    if (!scene) return;
    (void)hipSetDevice(scene->device);
    delete scene;
}

int pyr_render_simple_device(PyrScene* scene, const PyrCamera* camera, const PyrFilmDesc* film, const PyrRenderParams* params,
                             PyrGrain* film_device, void* hip_stream) {
    int rc = check_render_args(scene, camera, film, params, film_device);
    if (rc != PYR_OK) return rc;
    HIP_TRY(hipSetDevice(scene->device));
    TilePlan plan;
    if ((rc = plan_tiles(film, params, plan)) != PYR_OK) return rc;
    if (plan.chunk_end == plan.chunk_begin) return PYR_OK;
    hipStream_t stream = (hipStream_t)hip_stream;
    RenderLaunch L = make_launch(camera, film, params, plan);
    L.film_out = film_device;
    const bool count = (params->flags & PYR_FLAG_COUNTERS) != 0;
    if (count) {
        HIP_TRY(hipMemsetAsync(scene->counters.ptr, 0, sizeof(PyrCounters), stream));
        L.counters = (unsigned long long*)scene->counters.ptr;
        scene->have_counters = true;
    }
    return render_batches(scene, L, count, stream);
}

int pyr_render_simple(PyrScene* scene, const PyrCamera* camera, const PyrFilmDesc* film, const PyrRenderParams* params, PyrGrain* film_inout,
                      PyrProgressFn on_status, void* user) {
    int rc = check_render_args(scene, camera, film, params, film_inout);
    if (rc != PYR_OK) return rc;
    HIP_TRY(hipSetDevice(scene->device));
    TilePlan plan;
    if ((rc = plan_tiles(film, params, plan)) != PYR_OK) return rc;
    const char* message = "Rendering"; // simple.rs:30
    if (on_status) on_status(user, 0, message);
    const size_t bytes = (size_t)film_buffer_grains(film, params, plan) * sizeof(PyrGrain);
    DeviceBuffer film_dev;
    if ((rc = film_dev.upload(film_inout, bytes)) != PYR_OK) return rc;
    RenderLaunch L = make_launch(camera, film, params, plan);
    L.film_out = (PyrGrain*)film_dev.ptr;
    const bool count = (params->flags & PYR_FLAG_COUNTERS) != 0;
    if (count) {
        HIP_TRY(hipMemset(scene->counters.ptr, 0, sizeof(PyrCounters)));
        L.counters = (unsigned long long*)scene->counters.ptr;
        scene->have_counters = true;
    }
    // With a progress callback the chunk range is cut into slices so the caller hears back between launches
    // (the reference reports after every finished tile, simple.rs:49-55); results do not depend on the slicing.
    const uint32_t total_chunks = plan.chunk_end - plan.chunk_begin;
    const uint32_t slices = on_status ? std::min<uint32_t>(20, std::max<uint32_t>(1, total_chunks / 4096)) : 1;
    for (uint32_t sidx = 0; sidx < slices; ++sidx) {
        RenderLaunch part = L;
        part.chunk_begin = plan.chunk_begin + (uint32_t)((uint64_t)total_chunks * sidx / slices);
        part.chunk_end = plan.chunk_begin + (uint32_t)((uint64_t)total_chunks * (sidx + 1) / slices);
        rc = render_batches(scene, part, count, nullptr);
        if (rc != PYR_OK) return rc;
        HIP_TRY(hipDeviceSynchronize());
        if (on_status) on_status(user, (uint8_t)((sidx + 1) * 100 / slices), message);
    }
    if ((rc = check_tape_overflow(scene)) != PYR_OK) return rc;
    HIP_TRY(hipMemcpy(film_inout, film_dev.ptr, bytes, hipMemcpyDeviceToHost));
    return PYR_OK;
}

uint64_t pyr_film_blocks_grains(const PyrFilmDesc* film, const PyrRenderParams* params) {
    if (!film || !params || film->width == 0 || film->height == 0 || film->bins == 0 || params->tile_size == 0) {
        fail(PYR_ERR_INVALID_ARGUMENT, "zero-sized parameter");
        return 0;
    }
    TilePlan plan;
    PyrRenderParams p = *params;
    if (p.pixel_samples == 0) p.pixel_samples = 1; // the size does not depend on it
    if (plan_tiles(film, &p, plan) != PYR_OK) return 0;
    p.film_layout = PYR_FILM_TILE_BLOCKS;
    return film_buffer_grains(film, &p, plan);
}

int pyr_film_blocks_assemble_device(const PyrFilmDesc* film, const PyrRenderParams* params, const PyrGrain* blocks_device, PyrGrain* film_device, int device,
                                    void* hip_stream) {
    if (!film || !params || !blocks_device || !film_device) return fail(PYR_ERR_INVALID_ARGUMENT, "null argument");
    if (film->width == 0 || film->height == 0 || film->bins == 0 || params->tile_size == 0) return fail(PYR_ERR_INVALID_ARGUMENT, "zero-sized parameter");
    if (pyr_device_count() <= device || device < 0) return fail(PYR_ERR_DEVICE, "no such HIP device; pyrite_gpu has no CPU path");
    TilePlan plan;
    PyrRenderParams p = *params;
    if (p.pixel_samples == 0) p.pixel_samples = 1;
    int rc = plan_tiles(film, &p, plan);
    if (rc != PYR_OK) return rc;
    HIP_TRY(hipSetDevice(device));
    AssembleLaunch A{};
    A.film = *film;
    A.tile_size = params->tile_size;
    A.tiles_x = plan.tiles_x;
    A.tile_begin = plan.tile_begin;
    A.tile_stride = plan.tile_stride;
    A.tile_count = plan.tile_count;
    A.blocks = blocks_device;
    A.film_out = film_device;
    rc = launch_assemble(A, hip_stream);
    if (rc != PYR_OK) return fail(rc, kernels_last_error());
    return PYR_OK;
}

int pyr_scene_counters(PyrScene* scene, PyrCounters* out) {
    if (!scene || !out) return fail(PYR_ERR_INVALID_ARGUMENT, "null argument");
    if (!scene->have_counters) return fail(PYR_ERR_INVALID_ARGUMENT, "no render with PYR_FLAG_COUNTERS has run on this scene");
    HIP_TRY(hipSetDevice(scene->device));
    HIP_TRY(hipDeviceSynchronize());
    int rc = check_tape_overflow(scene);
    if (rc != PYR_OK) return rc;
    HIP_TRY(hipMemcpy(out, scene->counters.ptr, sizeof(PyrCounters), hipMemcpyDeviceToHost));
    return PYR_OK;
}

int pyr_scene_intersect_device(PyrScene* scene, const float* rays_device, uint32_t n, PyrHit* hits_device, void* hip_stream) {
    if (!scene || (n && (!rays_device || !hits_device))) return fail(PYR_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(scene->device));
    if (!scene->tail_count) HIP_TRY(hipMalloc((void**)&scene->tail_count, kFeedBytes));
    HIP_TRY(hipMemsetAsync(scene->tail_count, 0, kFeedBytes, (hipStream_t)hip_stream));
    IntersectLaunch L{rays_device, hits_device, n, nullptr, scene->tail_count, (uint32_t)scene->num_cus};
    int rc = launch_intersect(scene->dev, L, false, hip_stream);
    if (rc != PYR_OK) return fail(rc, kernels_last_error());
    return PYR_OK;
}

int pyr_scene_intersect(PyrScene* scene, const float* rays, uint32_t n, PyrHit* hits, float* elapsed_ms, PyrCounters* counters) {
    if (!scene || (n && (!rays || !hits))) return fail(PYR_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(scene->device));
    if (elapsed_ms) *elapsed_ms = 0.0f;
    if (counters) std::memset(counters, 0, sizeof(*counters));
    if (n == 0) return PYR_OK;
    DeviceBuffer rays_dev, hits_dev;
    int rc;
    if ((rc = rays_dev.upload(rays, (size_t)n * 24)) != PYR_OK) return rc;
    if ((rc = hits_dev.alloc((size_t)n * sizeof(PyrHit))) != PYR_OK) return rc;
    if (!scene->tail_count) HIP_TRY(hipMalloc((void**)&scene->tail_count, kFeedBytes));
    IntersectLaunch L{(const float*)rays_dev.ptr, (PyrHit*)hits_dev.ptr, n, nullptr, scene->tail_count, (uint32_t)scene->num_cus};
    if (counters) {
        HIP_TRY(hipMemset(scene->tail_count, 0, kFeedBytes));
        HIP_TRY(hipMemset(scene->counters.ptr, 0, sizeof(PyrCounters)));
        L.counters = (unsigned long long*)scene->counters.ptr;
        rc = launch_intersect(scene->dev, L, true, nullptr);
        if (rc != PYR_OK) return fail(rc, kernels_last_error());
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(counters, scene->counters.ptr, sizeof(PyrCounters), hipMemcpyDeviceToHost));
        L.counters = nullptr;
    }
    hipEvent_t start, stop;
    HIP_TRY(hipEventCreate(&start));
    HIP_TRY(hipEventCreate(&stop));
    HIP_TRY(hipMemset(scene->tail_count, 0, kFeedBytes));
    HIP_TRY(hipEventRecord(start, nullptr));
    rc = launch_intersect(scene->dev, L, false, nullptr);
    if (rc != PYR_OK) return fail(rc, kernels_last_error());
    HIP_TRY(hipEventRecord(stop, nullptr));
    HIP_TRY(hipEventSynchronize(stop));
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, start, stop));
    (void)hipEventDestroy(start);
    (void)hipEventDestroy(stop);
    if (elapsed_ms) *elapsed_ms = ms;
    HIP_TRY(hipMemcpy(hits, hits_dev.ptr, (size_t)n * sizeof(PyrHit), hipMemcpyDeviceToHost));
    return PYR_OK;
}

static int develop_common(const PyrFilmDesc* film, const PyrGrain* grains_device, const PyrDevelopParams* p, uint8_t* rgb_device, hipStream_t stream,
                          bool blocking) {
    if (!(p->step_size > 0.0f) || p->sample_count == 0 || p->xyz_count < 2) return fail(PYR_ERR_INVALID_ARGUMENT, "bad development parameters");
    if ((p->white_div == nullptr) != (p->white_mul == nullptr)) return fail(PYR_ERR_INVALID_ARGUMENT, "white_div and white_mul go together");
    // the small per-wavelength tables are copied for the call (stream-ordered allocation, freed after the kernel)
    const size_t n = p->sample_count;
    float* tables = nullptr;
    const size_t floats = 3 * n + 3 * (size_t)p->xyz_count;
    HIP_TRY(hipMallocAsync((void**)&tables, floats * sizeof(float), stream));
    std::vector<float> host(floats, 0.0f);
    if (p->filter) std::memcpy(host.data(), p->filter, n * 4);
    if (p->white_div) {
        std::memcpy(host.data() + n, p->white_div, n * 4);
        std::memcpy(host.data() + 2 * n, p->white_mul, n * 4);
    }
    std::memcpy(host.data() + 3 * n, p->xyz_table, 3 * (size_t)p->xyz_count * 4);
    {
        const hipError_t copied = hipMemcpy(tables, host.data(), floats * sizeof(float), hipMemcpyHostToDevice); // synchronous: `host` dies with this frame
        if (copied != hipSuccess) {
            (void)hipFreeAsync(tables, stream);
            return hip_fail(copied, "hipMemcpy(development tables)");
        }
    }
    DevelopLaunch D{};
    D.film = *film;
    D.grains = grains_device;
    D.step_size = p->step_size;
    D.xyz_scale = p->xyz_scale;
    D.sample_count = p->sample_count;
    D.filter = p->filter ? tables : nullptr;
    D.white_div = p->white_div ? tables + n : nullptr;
    D.white_mul = p->white_div ? tables + 2 * n : nullptr;
    D.xyz_table = tables + 3 * n;
    D.xyz_count = p->xyz_count;
    D.xyz_min = p->xyz_min;
    D.xyz_max = p->xyz_max;
    D.rgb_out = rgb_device;
    int rc = launch_develop(D, stream);
    hipError_t e = hipFreeAsync(tables, stream);
    if (rc != PYR_OK) return fail(rc, kernels_last_error());
    if (e != hipSuccess) return hip_fail(e, "hipFreeAsync");
    if (blocking) HIP_TRY(hipStreamSynchronize(stream));
    return PYR_OK;
}

int pyr_film_develop_device(const PyrFilmDesc* film, const PyrGrain* grains_device, const PyrDevelopParams* params, uint8_t* rgb_device, int device,
                            void* hip_stream) {
    if (!film || !grains_device || !params || !rgb_device || !params->xyz_table) return fail(PYR_ERR_INVALID_ARGUMENT, "null argument");
    if (pyr_device_count() <= device || device < 0) return fail(PYR_ERR_DEVICE, "no such HIP device; pyrite_gpu has no CPU path");
    HIP_TRY(hipSetDevice(device));
    return develop_common(film, grains_device, params, rgb_device, (hipStream_t)hip_stream, false);
}

int pyr_film_develop(const PyrFilmDesc* film, const PyrGrain* grains, const PyrDevelopParams* params, uint8_t* rgb_out, int device) {
    if (!film || !grains || !params || !rgb_out || !params->xyz_table) return fail(PYR_ERR_INVALID_ARGUMENT, "null argument");
    if (pyr_device_count() <= device || device < 0) return fail(PYR_ERR_DEVICE, "no such HIP device; pyrite_gpu has no CPU path");
    HIP_TRY(hipSetDevice(device));
    const size_t pixels = (size_t)film->width * film->height;
    DeviceBuffer film_dev, rgb_dev;
    int rc;
    if ((rc = film_dev.upload(grains, pixels * film->bins * sizeof(PyrGrain))) != PYR_OK) return rc;
    if ((rc = rgb_dev.alloc(pixels * 3)) != PYR_OK) return rc;
    if ((rc = develop_common(film, (const PyrGrain*)film_dev.ptr, params, (uint8_t*)rgb_dev.ptr, nullptr, true)) != PYR_OK) return rc;
    HIP_TRY(hipMemcpy(rgb_out, rgb_dev.ptr, pixels * 3, hipMemcpyDeviceToHost));
    return PYR_OK;
}

int pyr_scene_path_info(PyrScene* scene, const PyrRenderParams* params, PyrPathInfo* out) {
    if (!scene || !params || !out) return fail(PYR_ERR_INVALID_ARGUMENT, "null argument");
    RenderLaunch L{};
    L.spectrum_samples = params->spectrum_samples;
    choose_schedule(scene, L);
    *out = PyrPathInfo{};
    out->stage_scheduler = L.scheduler;
    out->interpreter = scene->dev.needs_interpreter;
    out->scene_in_lds = scene_is_lds_resident(scene->dev) ? 1u : 0u;
    out->tape = L.scheduler == 0 ? 0u : scene->dev.needs_interpreter == 0 ? 1u : uses_hit_tape(scene->dev, L) ? 2u : 0u;
    out->phase_lanes = L.scheduler != 0 ? L.sm_phase_lanes : 0u;
    return PYR_OK;
}

int pyr_scene_bvh_info(PyrScene* scene, PyrBvhInfo* out) {
    if (!scene || !out) return fail(PYR_ERR_INVALID_ARGUMENT, "null argument");
    *out = scene->info;
    return PYR_OK;
}

} // extern "C"
