// pyrite_host.cpp -- the C++ host layer of include/pyrite_host.hpp: project tree -> flat scene -> C ABI.
//
// Built with -ffp-contract=off: every f32 expression below rounds like the reference's Rust (and like the Python front-end
// pyrite_amd/compiler.py, against which tests/test_host_cpp.py compares the flattened scenes byte for byte).
#include "pyrite_host.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <sstream>
#include <unordered_map>

namespace pyrite {

namespace {
#include "builtin_tables.inc"

float bits_to_float(uint32_t b) {
    float f;
    std::memcpy(&f, &b, 4);
    return f;
}
uint32_t float_bits(float f) {
    uint32_t b;
    std::memcpy(&b, &f, 4);
    return b;
}
std::vector<float> table(const uint32_t* bits, size_t n) {
    std::vector<float> out(n);
    for (size_t i = 0; i < n; ++i) out[i] = bits_to_float(bits[i]);
    return out;
}

std::shared_ptr<ExprNode> make_node(ExprKind kind, std::vector<Expression> args = {}) {
    auto n = std::make_shared<ExprNode>();
    n->kind = kind;
    n->args = std::move(args);
    return n;
}

void check_status(int status) {
    if (status != PYR_OK) throw GpuError(status, std::string("libpyrite_gpu: ") + pyr_last_error());
}
} // namespace

// ================================================================================================ expressions
Expression::Expression(double number) {
    auto n = std::make_shared<ExprNode>();
    n->kind = ExprKind::Number;
    n->number = number;
    node_ = n;
}
bool Expression::is_number() const { return node_->kind == ExprKind::Number; }
double Expression::number() const { return node_->number; }
Expression Expression::mix(const Expression& other, const Expression& amount) const { return pyrite::mix(*this, other, amount); }

static Expression binary(BinaryOp op, const Expression& a, const Expression& b) { // lib.lua:2-11
    auto n = make_node(ExprKind::Binary, {a, b});
    n->op = op;
    return Expression(n);
}
Expression operator+(const Expression& a, const Expression& b) { return binary(BinaryOp::Add, a, b); }
Expression operator-(const Expression& a, const Expression& b) { return binary(BinaryOp::Sub, a, b); }
Expression operator*(const Expression& a, const Expression& b) { return binary(BinaryOp::Mul, a, b); }
Expression operator/(const Expression& a, const Expression& b) { return binary(BinaryOp::Div, a, b); }
Expression mix(const Expression& lhs, const Expression& rhs, const Expression& amount) { return Expression(make_node(ExprKind::Mix, {lhs, rhs, amount})); }
Expression clamp(const Expression& value, const Expression& min, const Expression& max) { return Expression(make_node(ExprKind::Clamp, {value, min, max})); }
Expression fresnel(const Expression& ior, const Expression& env_ior) { return Expression(make_node(ExprKind::Fresnel, {ior, env_ior})); }
Expression vector(const Expression& x, const Expression& y, const Expression& z, const Expression& w) { return Expression(make_node(ExprKind::Vector, {x, y, z, w})); }
Expression blackbody(const Expression& temperature) { return Expression(make_node(ExprKind::Blackbody, {temperature})); }
Expression rgb(const Expression& red, const Expression& green, const Expression& blue) { return Expression(make_node(ExprKind::Rgb, {red, green, blue})); }
Expression spectrum_array(float min, float max, std::vector<float> points) {
    auto n = make_node(ExprKind::Spectrum);
    n->format = SpectrumFormat::Array;
    n->min = min;
    n->max = max;
    n->points = std::move(points);
    return Expression(n);
}
Expression spectrum_curve(std::vector<std::pair<float, float>> points) {
    auto n = make_node(ExprKind::Spectrum);
    n->format = SpectrumFormat::Curve;
    for (auto& p : points) {
        n->points.push_back(p.first);
        n->points.push_back(p.second);
    }
    return Expression(n);
}
static Expression texture_node(ExprKind kind, uint32_t width, uint32_t height, std::vector<float> texels, uint32_t channels) {
    if ((size_t)width * height * channels != texels.size() || texels.empty()) throw ProjectError("texture: texel count does not match width x height");
    auto n = make_node(kind);
    n->tex_width = width;
    n->tex_height = height;
    n->texels = std::move(texels);
    return Expression(n);
}
Expression color_texture(uint32_t width, uint32_t height, std::vector<float> rgba) { return texture_node(ExprKind::ColorTexture, width, height, std::move(rgba), 4); }
Expression mono_texture(uint32_t width, uint32_t height, std::vector<float> luma) { return texture_node(ExprKind::MonoTexture, width, height, std::move(luma), 1); }
namespace light_source {
static Expression builtin(SpectrumFormat f) {
    auto n = make_node(ExprKind::Spectrum);
    n->format = f;
    return Expression(n);
}
// one node each: SpectrumId::from_lua hands out one id per Lua table (project/spectra.rs:116-145)
Expression d65() {
    static const Expression e = builtin(SpectrumFormat::BuiltinD65);
    return e;
}
Expression a() {
    static const Expression e = builtin(SpectrumFormat::BuiltinA);
    return e;
}
} // namespace light_source

// ================================================================================================ materials
static SurfaceMaterial leaf(MaterialKind kind, const Expression& color) {
    auto n = std::make_shared<MaterialNode>();
    n->kind = kind;
    n->color = color;
    return SurfaceMaterial(n);
}
namespace material {
SurfaceMaterial diffuse(const Expression& color) { return leaf(MaterialKind::Diffuse, color); }
SurfaceMaterial emissive(const Expression& color) { return leaf(MaterialKind::Emissive, color); }
SurfaceMaterial mirror(const Expression& color) { return leaf(MaterialKind::Mirror, color); }
SurfaceMaterial refractive(const Expression& color, const Expression& ior, std::optional<Expression> dispersion, std::optional<Expression> env_ior,
                           std::optional<Expression> env_dispersion) {
    auto n = std::make_shared<MaterialNode>();
    n->kind = MaterialKind::Refractive;
    n->color = color;
    n->ior = ior;
    n->dispersion = std::move(dispersion);
    n->env_ior = std::move(env_ior);
    n->env_dispersion = std::move(env_dispersion);
    return SurfaceMaterial(n);
}
} // namespace material
SurfaceMaterial operator+(const SurfaceMaterial& a, const SurfaceMaterial& b) {
    auto n = std::make_shared<MaterialNode>();
    n->kind = MaterialKind::Add;
    n->lhs = a;
    n->rhs = b;
    return SurfaceMaterial(n);
}
SurfaceMaterial mix(const SurfaceMaterial& lhs, const SurfaceMaterial& rhs, const Expression& amount) {
    auto n = std::make_shared<MaterialNode>();
    n->kind = MaterialKind::Mix;
    n->lhs = lhs;
    n->rhs = rhs;
    n->amount = amount;
    return SurfaceMaterial(n);
}
SurfaceMaterial SurfaceMaterial::mix(const SurfaceMaterial& other, const Expression& amount) const { return pyrite::mix(*this, other, amount); }

// ================================================================================================ constant evaluation
// project/expressions.rs:75-258 (EvalContext): Evaluate<f32> :270-296, Evaluate<Vector> :326-353.
namespace {
struct V3 {
    float x, y, z;
};
struct V4 {
    float x, y, z, w;
};

float eval_number(const Expression& e) {
    const ExprNode& n = e.node();
    switch (n.kind) {
    case ExprKind::Number: return (float)n.number;
    case ExprKind::Binary: {
        const float l = eval_number(n.args[0]), r = eval_number(n.args[1]);
        switch (n.op) {
        case BinaryOp::Add: return l + r;
        case BinaryOp::Sub: return l - r;
        case BinaryOp::Mul: return l * r;
        default: return l / r;
        }
    }
    case ExprKind::Mix: {
        const float amount = std::min(std::max(eval_number(n.args[2]), 0.0f), 1.0f);
        return eval_number(n.args[0]) * (1.0f - amount) + eval_number(n.args[1]) * amount;
    }
    case ExprKind::Clamp: return std::max(std::min(eval_number(n.args[0]), eval_number(n.args[2])), eval_number(n.args[1]));
    case ExprKind::Vector: throw ProjectError("expected a number, but found a vector");
    case ExprKind::Rgb: throw ProjectError("expected a number, but found an RGB color");
    default: throw ProjectError("cannot evaluate this expression as a constant");
    }
}

V4 eval_vector(const Expression& e) {
    const ExprNode& n = e.node();
    switch (n.kind) {
    case ExprKind::Number: {
        const float f = (float)n.number;
        return V4{f, f, f, f};
    }
    case ExprKind::Vector: return V4{eval_number(n.args[0]), eval_number(n.args[1]), eval_number(n.args[2]), eval_number(n.args[3])};
    case ExprKind::Binary: {
        const V4 l = eval_vector(n.args[0]), r = eval_vector(n.args[1]);
        switch (n.op) {
        case BinaryOp::Add: return V4{l.x + r.x, l.y + r.y, l.z + r.z, l.w + r.w};
        case BinaryOp::Sub: return V4{l.x - r.x, l.y - r.y, l.z - r.z, l.w - r.w};
        case BinaryOp::Mul: return V4{l.x * r.x, l.y * r.y, l.z * r.z, l.w * r.w};
        default: return V4{l.x / r.x, l.y / r.y, l.z / r.z, l.w / r.w};
        }
    }
    case ExprKind::Mix: {
        const float amount = std::min(std::max(eval_number(n.args[2]), 0.0f), 1.0f);
        const V4 l = eval_vector(n.args[0]), r = eval_vector(n.args[1]);
        return V4{l.x + (r.x - l.x) * amount, l.y + (r.y - l.y) * amount, l.z + (r.z - l.z) * amount, l.w + (r.w - l.w) * amount};
    }
    case ExprKind::Rgb: throw ProjectError("expected a vector, but found an RGB color");
    default: throw ProjectError("cannot evaluate this expression as a constant");
    }
}
V3 xyz(const V4& v) { return V3{v.x, v.y, v.z}; }

// cgmath 0.17 [3P]: the same operation order as the oracle and pyrite_amd/compiler.py
V3 sub(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
V3 add(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
V3 scale(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
V3 normalize(V3 v) { // v * (1 / |v|), |v| = sqrt((x*x + y*y) + z*z)
    const float mag = std::sqrt((v.x * v.x + v.y * v.y) + v.z * v.z);
    return scale(v, 1.0f / mag);
}

struct Mat4 {
    float m[16];
};
Mat4 identity() { return Mat4{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}}; }
// Transform::LookAt (project/mod.rs:250-266): Matrix4::look_at(from, to, up).invert() -- the inverse of that rigid view
// matrix is [s u -f | from], written out directly.
Mat4 eval_transform(const LookAt& t) {
    const V3 from = xyz(eval_vector(t.from)), to = xyz(eval_vector(t.to));
    const V3 up = t.up ? xyz(eval_vector(*t.up)) : V3{0, 1, 0};
    const V3 f = normalize(sub(to, from));
    const V3 s = normalize(cross(f, up));
    const V3 u = cross(s, f);
    Mat4 m{};
    m.m[0] = s.x, m.m[1] = s.y, m.m[2] = s.z;
    m.m[4] = u.x, m.m[5] = u.y, m.m[6] = u.z;
    m.m[8] = -f.x, m.m[9] = -f.y, m.m[10] = -f.z;
    m.m[12] = from.x, m.m[13] = from.y, m.m[14] = from.z;
    m.m[15] = 1.0f;
    return m;
}
V3 transform_point(const Mat4& mm, V3 p) {
    const float* m = mm.m;
    const float x = m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12];
    const float y = m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13];
    const float z = m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14];
    const float w = m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15];
    const float inv = 1.0f / w;
    return V3{x * inv, y * inv, z * inv};
}
V3 transform_vector(const Mat4& mm, V3 v) {
    const float* m = mm.m;
    return V3{m[0] * v.x + m[4] * v.y + m[8] * v.z, m[1] * v.x + m[5] * v.y + m[9] * v.z, m[2] * v.x + m[6] * v.y + m[10] * v.z};
}
struct Quat {
    float s, x, y, z;
};
Quat quat_from_cols(V3 c0, V3 c1, V3 c2) { // Quaternion::from(Matrix3::from_cols(c0, c1, c2))
    const float m00 = c0.x, m01 = c0.y, m02 = c0.z, m10 = c1.x, m11 = c1.y, m12 = c1.z, m20 = c2.x, m21 = c2.y, m22 = c2.z;
    const float trace = m00 + m11 + m22;
    if (trace >= 0.0f) {
        float s = std::sqrt(1.0f + trace);
        const float w = 0.5f * s;
        s = 0.5f / s;
        return Quat{w, (m12 - m21) * s, (m20 - m02) * s, (m01 - m10) * s};
    }
    if (m00 > m11 && m00 > m22) {
        float s = std::sqrt((m00 - m11 - m22) + 1.0f);
        const float x = 0.5f * s;
        s = 0.5f / s;
        return Quat{(m12 - m21) * s, x, (m10 + m01) * s, (m02 + m20) * s};
    }
    if (m11 > m22) {
        float s = std::sqrt((m11 - m00 - m22) + 1.0f);
        const float y = 0.5f * s;
        s = 0.5f / s;
        return Quat{(m20 - m02) * s, (m10 + m01) * s, y, (m21 + m12) * s};
    }
    float s = std::sqrt((m22 - m00 - m11) + 1.0f);
    const float z = 0.5f * s;
    s = 0.5f / s;
    return Quat{(m01 - m10) * s, (m02 + m20) * s, (m21 + m12) * s, z};
}
V3 quat_rotate(Quat q, V3 vec) {
    const V3 v{q.x, q.y, q.z};
    const V3 tmp = add(cross(v, vec), scale(vec, q.s));
    return add(scale(cross(v, tmp), 2.0f), vec);
}
V3 ortho(V3 v) { // math.rs:98-114
    const float eps = 1.0e-4f;
    V3 unit;
    if (std::fabs(v.x) < eps)
        unit = V3{1, 0, 0};
    else if (std::fabs(v.y) < eps)
        unit = V3{0, 1, 0};
    else if (std::fabs(v.z) < eps)
        unit = V3{0, 0, 1};
    else
        unit = V3{-v.y, v.x, 0};
    return cross(v, unit);
}
void normal_transform(const Mat4& xform, V3 vec, Quat frame, V3& n_out, Quat& frame_out) { // Normal::transform, shapes/mod.rs:572-583
    n_out = normalize(transform_vector(xform, vec));
    const V3 x = normalize(transform_vector(xform, quat_rotate(frame, V3{1, 0, 0})));
    const V3 y = normalize(transform_vector(xform, quat_rotate(frame, V3{0, 1, 0})));
    frame_out = quat_from_cols(x, y, n_out);
}

// expression helpers used by material flattening (expressions.rs:20-63); Expression::Number is f64
Expression insert_sub(const Expression& l, const Expression& r) {
    if (l.is_number() && r.is_number()) return Expression(l.number() - r.number());
    return l - r;
}
Expression insert_mul(const Expression& l, const Expression& r) {
    if (l.is_number() && r.is_number()) return Expression(l.number() * r.number());
    return l * r;
}
Expression insert_clamp(const Expression& v, const Expression& mn, const Expression& mx) {
    if (v.is_number() && mn.is_number() && mx.is_number()) return Expression(std::max(std::min(v.number(), mx.number()), mn.number()));
    return clamp(v, mn, mx);
}
} // namespace

// ================================================================================================ the flat scene
struct FlatScene::Impl {
    std::vector<float> tri_positions, tri_normals, tri_uvs, tri_frames;
    std::vector<uint32_t> tri_material;
    std::vector<float> spheres, sphere_tex_scale;
    std::vector<uint32_t> sphere_material;
    std::vector<float> planes, plane_frames;
    std::vector<uint32_t> plane_material;
    std::vector<PyrLamp> lamps;
    std::vector<PyrMaterial> materials;
    std::vector<PyrComponent> components;
    std::vector<PyrProgram> programs;
    std::vector<PyrInstr> instrs;
    std::vector<PyrSpectrum> spectra;
    std::vector<float> spectrum_data;
    std::vector<PyrTexture> textures;
    std::vector<float> texture_data;
    std::vector<float> rgb_basis;
    std::unordered_map<const ExprNode*, uint32_t> spectrum_ids, texture_ids;
    std::vector<Expression> keep_alive; // ids stay unique while the expression lives
    bool uses_rgb_basis = false, uses_normal_maps = false;
    uint32_t sky_program = 0;
    std::string base_dir = ".";
    PyrSceneDesc desc{};

    uint32_t spectrum_id(const Expression& e);
    uint32_t texture_id(const Expression& e);
    void add_mesh(size_t index, const Mesh& mesh, FlatScene& self);
    void add_mesh_triangle(const MeshData& mesh, const std::vector<MeshData::Index>& poly, uint32_t material, float scale_factor, const Mat4& xform,
                           FlatScene& self);
};

FlatScene::FlatScene() : impl_(new Impl) {}
FlatScene::~FlatScene() = default;
size_t FlatScene::num_triangles() const { return impl_->tri_material.size(); }
size_t FlatScene::num_spheres() const { return impl_->sphere_material.size(); }
size_t FlatScene::num_planes() const { return impl_->plane_material.size(); }

// SpectrumId::from_lua (project/spectra.rs:116-145): one id per Lua table
uint32_t FlatScene::Impl::spectrum_id(const Expression& e) {
    auto it = spectrum_ids.find(e.id());
    if (it != spectrum_ids.end()) return it->second;
    const ExprNode& n = e.node();
    PyrSpectrum rec{};
    std::vector<float> data;
    switch (n.format) {
    case SpectrumFormat::BuiltinD65:
    case SpectrumFormat::BuiltinA:
        rec.format = PYR_SPECTRUM_ARRAY;
        rec.min = k_light_min;
        rec.max = k_light_max;
        data = n.format == SpectrumFormat::BuiltinD65 ? table(k_d65_bits, k_d65_rows) : table(k_a_bits, k_a_rows);
        rec.count = (uint32_t)data.size();
        break;
    case SpectrumFormat::Array:
        rec.format = PYR_SPECTRUM_ARRAY;
        rec.min = n.min;
        rec.max = n.max;
        data = n.points;
        rec.count = (uint32_t)data.size();
        break;
    case SpectrumFormat::Curve:
        rec.format = PYR_SPECTRUM_CURVE;
        data = n.points;
        rec.count = (uint32_t)(data.size() / 2);
        break;
    }
    rec.offset = (uint32_t)spectrum_data.size();
    const uint32_t id = (uint32_t)spectra.size();
    spectra.push_back(rec);
    spectrum_data.insert(spectrum_data.end(), data.begin(), data.end());
    spectrum_ids[e.id()] = id;
    keep_alive.push_back(e);
    return id;
}

// TextureLoader::load_color / load_mono (project/textures.rs:56-118): one id per texture and kind
uint32_t FlatScene::Impl::texture_id(const Expression& e) {
    auto it = texture_ids.find(e.id());
    if (it != texture_ids.end()) return it->second;
    const ExprNode& n = e.node();
    PyrTexture rec{};
    rec.format = n.kind == ExprKind::MonoTexture ? PYR_TEXTURE_MONO : PYR_TEXTURE_COLOR;
    rec.width = n.tex_width;
    rec.height = n.tex_height;
    rec.offset = texture_data.size();
    const uint32_t id = (uint32_t)textures.size();
    textures.push_back(rec);
    texture_data.insert(texture_data.end(), n.texels.begin(), n.texels.end());
    texture_ids[e.id()] = id;
    keep_alive.push_back(e);
    return id;
}

// ---- ProgramCompiler::compile (program/compiler.rs:48-586) ----------------------------------------------------------------
namespace {
enum RegKind { RN = 0, RV = 1, RC = 2 }; // number / vector / rgb register files
struct Pending {
    Expression child;
};
struct Operand {
    uint32_t kind, bits;
};
struct Got { // try_get_register's result: a literal number or a finished register
    bool is_number;
    float number;
    RegKind kind;
    uint32_t reg, deps;
};
struct Status {
    bool done = false;
    RegKind kind = RN;
    uint32_t reg = 0, deps = 0;
};
} // namespace

uint32_t FlatScene::compile(const Expression& expression, bool allow_wavelength, bool vector_output) {
    Impl& S = *impl_;
    if (expression.is_number()) { // compiler.rs:62-69
        PyrProgram p{};
        p.kind = PYR_PROGRAM_CONSTANT;
        p.constant = (float)expression.number();
        p.output_kind = vector_output ? PYR_OUTPUT_VECTOR : PYR_OUTPUT_NUMBER;
        S.programs.push_back(p);
        return (uint32_t)S.programs.size() - 1;
    }
    std::unordered_map<const ExprNode*, Status> status;
    std::vector<Expression> pending{expression};
    status[expression.id()];
    std::vector<PyrInstr> instructions;
    uint32_t counts[3] = {0, 0, 0};
    const Operand zero{PYR_OPERAND_CONSTANT, 0};

    auto next_reg = [&](RegKind k) { return counts[k]++; };
    auto emit = [&](uint32_t op, uint32_t output, uint32_t deps, Operand x, Operand y, Operand z, Operand w, uint32_t a = 0, uint32_t b = 0, uint32_t value_type = 0,
                    uint32_t oper = 0) {
        PyrInstr r{};
        r.op = op, r.value_type = value_type, r.operator_ = oper, r.deps = deps, r.output = output, r.a = a, r.b = b;
        r.x = PyrOperand{x.kind, x.bits}, r.y = PyrOperand{y.kind, y.bits}, r.z = PyrOperand{z.kind, z.bits}, r.w = PyrOperand{w.kind, w.bits};
        instructions.push_back(r);
    };
    auto number_input = [&](uint32_t& deps) -> Operand { // get_number_input, compiler.rs:970-975
        if (!allow_wavelength) throw ProjectError("the wavelength is not available during normal mapping");
        deps = PYR_DEP_WAVELENGTH;
        return Operand{PYR_OPERAND_INPUT, PYR_INPUT_WAVELENGTH};
    };
    auto try_get_register = [&](const Expression& e) -> Got { // compiler.rs:609-634
        if (e.is_number()) return Got{true, (float)e.number(), RN, 0, 0};
        Status& st = status[e.id()];
        if (st.done) return Got{false, 0.0f, st.kind, st.reg, st.deps};
        throw Pending{e};
    };
    auto const_operand = [](float v) { return Operand{PYR_OPERAND_CONSTANT, float_bits(v)}; };
    auto try_get_number_value = [&](const Expression& e, uint32_t& deps) -> Operand { // compiler.rs:636-680
        const Got got = try_get_register(e);
        if (got.is_number) {
            deps = 0;
            return const_operand(got.number);
        }
        if (got.kind == RN) {
            deps = got.deps;
            return Operand{PYR_OPERAND_REGISTER, got.reg};
        }
        if (got.kind == RV) throw ProjectError("cannot use a vector as a number");
        uint32_t wl_deps = 0;
        const Operand wl = number_input(wl_deps);
        const uint32_t out = next_reg(RN);
        S.uses_rgb_basis = true;
        emit(PYR_OP_RGB_SPECTRUM, out, got.deps | wl_deps, wl, zero, zero, zero, got.reg);
        deps = got.deps | wl_deps;
        return Operand{PYR_OPERAND_REGISTER, out};
    };
    auto number_constant_to = [&](RegKind kind, float number) { // compiler.rs:991-1008, :1047-1064
        const uint32_t out = next_reg(kind);
        const Operand c = const_operand(number);
        if (kind == RV)
            emit(PYR_OP_VECTOR, out, 0, c, c, c, c);
        else
            emit(PYR_OP_RGB, out, 0, c, c, c, zero);
        return out;
    };
    auto number_register_to = [&](RegKind kind, uint32_t reg, uint32_t deps) { // compiler.rs:1010-1028, :1066-1083
        const uint32_t out = next_reg(kind);
        const Operand r{PYR_OPERAND_REGISTER, reg};
        if (kind == RV)
            emit(PYR_OP_VECTOR, out, deps, r, r, r, r);
        else
            emit(PYR_OP_RGB, out, deps, r, r, r, zero);
        return out;
    };
    auto rgb_register_to_vector = [&](uint32_t reg, uint32_t deps) { // compiler.rs:1030-1045
        const uint32_t out = next_reg(RV);
        emit(PYR_OP_RGB_TO_VECTOR, out, deps, zero, zero, zero, zero, reg);
        return out;
    };
    struct Side {
        uint32_t reg, deps;
    };
    auto convert_operands = [&](const Got& lhs, const Got& rhs, Side& l, Side& r) -> RegKind { // compiler.rs:682-968
        if (lhs.is_number && rhs.is_number) {
            const uint32_t lo = next_reg(RN), ro = next_reg(RN);
            emit(PYR_OP_NUMBER, lo, 0, const_operand(lhs.number), zero, zero, zero);
            emit(PYR_OP_NUMBER, ro, 0, const_operand(rhs.number), zero, zero, zero);
            l = Side{lo, 0}, r = Side{ro, 0};
            return RN;
        }
        if (lhs.is_number) {
            if (rhs.kind == RN) {
                const uint32_t lo = next_reg(RN);
                emit(PYR_OP_NUMBER, lo, 0, const_operand(lhs.number), zero, zero, zero);
                l = Side{lo, 0}, r = Side{rhs.reg, rhs.deps};
                return RN;
            }
            l = Side{number_constant_to(rhs.kind, lhs.number), 0}, r = Side{rhs.reg, rhs.deps};
            return rhs.kind;
        }
        if (rhs.is_number) {
            if (lhs.kind == RN) {
                const uint32_t ro = next_reg(RN);
                emit(PYR_OP_NUMBER, ro, 0, const_operand(rhs.number), zero, zero, zero);
                l = Side{lhs.reg, lhs.deps}, r = Side{ro, 0};
                return RN;
            }
            l = Side{lhs.reg, lhs.deps}, r = Side{number_constant_to(lhs.kind, rhs.number), 0};
            return lhs.kind;
        }
        if (lhs.kind == rhs.kind) {
            l = Side{lhs.reg, lhs.deps}, r = Side{rhs.reg, rhs.deps};
            return lhs.kind;
        }
        if (lhs.kind == RN) { // number with vector / rgb: widen the number
            l = Side{number_register_to(rhs.kind, lhs.reg, lhs.deps), lhs.deps}, r = Side{rhs.reg, rhs.deps};
            return rhs.kind;
        }
        if (rhs.kind == RN) {
            l = Side{lhs.reg, lhs.deps}, r = Side{number_register_to(lhs.kind, rhs.reg, rhs.deps), rhs.deps};
            return lhs.kind;
        }
        if (lhs.kind == RV) { // vector with rgb: rgb -> vector
            l = Side{lhs.reg, lhs.deps}, r = Side{rgb_register_to_vector(rhs.reg, rhs.deps), rhs.deps};
            return RV;
        }
        l = Side{rgb_register_to_vector(lhs.reg, lhs.deps), lhs.deps}, r = Side{rhs.reg, rhs.deps};
        return RV;
    };
    auto done = [&](const Expression& e, RegKind kind, uint32_t reg, uint32_t deps) {
        Status& st = status[e.id()];
        st.done = true, st.kind = kind, st.reg = reg, st.deps = deps;
    };
    const uint32_t VT[3] = {PYR_VT_NUMBER, PYR_VT_VECTOR, PYR_VT_RGB};

    while (!pending.empty()) {
        const Expression e = pending.back();
        pending.pop_back();
        if (status[e.id()].done) continue;
        try {
            const ExprNode& n = e.node();
            switch (n.kind) {
            case ExprKind::Vector: {
                uint32_t xd, yd, zd, wd;
                const Operand x = try_get_number_value(n.args[0], xd);
                const Operand y = try_get_number_value(n.args[1], yd);
                const Operand z = try_get_number_value(n.args[2], zd);
                const Operand w = try_get_number_value(n.args[3], wd);
                const uint32_t out = next_reg(RV), deps = xd | yd | zd | wd;
                emit(PYR_OP_VECTOR, out, deps, x, y, z, w);
                done(e, RV, out, deps);
                break;
            }
            case ExprKind::Rgb: {
                uint32_t rd, gd, bd;
                const Operand r = try_get_number_value(n.args[0], rd);
                const Operand g = try_get_number_value(n.args[1], gd);
                const Operand b = try_get_number_value(n.args[2], bd);
                const uint32_t out = next_reg(RC), deps = rd | gd | bd;
                emit(PYR_OP_RGB, out, deps, r, g, b, zero);
                done(e, RC, out, deps);
                break;
            }
            case ExprKind::Fresnel: {
                uint32_t iord, envd;
                const Operand ior = try_get_number_value(n.args[0], iord);
                const Operand env = try_get_number_value(n.args[1], envd);
                const uint32_t out = next_reg(RN), deps = PYR_DEP_NORMAL | PYR_DEP_INCIDENT | iord | envd;
                emit(PYR_OP_FRESNEL, out, deps, ior, env, zero, zero, PYR_INPUT_NORMAL, PYR_INPUT_INCIDENT);
                done(e, RN, out, deps);
                break;
            }
            case ExprKind::Blackbody: {
                uint32_t wld, td;
                const Operand wl = number_input(wld);
                const Operand temp = try_get_number_value(n.args[0], td);
                const uint32_t out = next_reg(RN), deps = wld | td;
                emit(PYR_OP_BLACKBODY, out, deps, wl, temp, zero, zero);
                done(e, RN, out, deps);
                break;
            }
            case ExprKind::Spectrum: {
                uint32_t deps;
                const Operand wl = number_input(deps);
                const uint32_t out = next_reg(RN);
                emit(PYR_OP_SPECTRUM, out, deps, wl, zero, zero, zero, S.spectrum_id(e));
                done(e, RN, out, deps);
                break;
            }
            case ExprKind::ColorTexture:
            case ExprKind::MonoTexture: { // compiler.rs:282-323
                const RegKind kind = n.kind == ExprKind::ColorTexture ? RC : RN;
                const uint32_t out = next_reg(kind);
                emit(kind == RC ? PYR_OP_COLOR_TEXTURE : PYR_OP_MONO_TEXTURE, out, PYR_DEP_TEXTURE, zero, zero, zero, zero, S.texture_id(e), PYR_INPUT_TEXTURE);
                done(e, kind, out, PYR_DEP_TEXTURE);
                break;
            }
            case ExprKind::Mix: {
                uint32_t ad;
                const Operand amount = try_get_number_value(n.args[2], ad);
                const Got lhs = try_get_register(n.args[0]);
                const Got rhs = try_get_register(n.args[1]);
                Side l, r;
                const RegKind kind = convert_operands(lhs, rhs, l, r);
                const uint32_t deps = ad | l.deps | r.deps;
                const uint32_t out = next_reg(kind);
                done(e, kind, out, deps);
                emit(PYR_OP_MIX, out, deps, amount, zero, zero, zero, l.reg, r.reg, VT[kind]);
                break;
            }
            case ExprKind::Binary: {
                const Got lhs = try_get_register(n.args[0]);
                const Got rhs = try_get_register(n.args[1]);
                Side l, r;
                const RegKind kind = convert_operands(lhs, rhs, l, r);
                const uint32_t deps = l.deps | r.deps;
                const uint32_t out = next_reg(kind);
                done(e, kind, out, deps);
                const uint32_t op = n.op == BinaryOp::Add ? PYR_BIN_ADD : n.op == BinaryOp::Sub ? PYR_BIN_SUB : n.op == BinaryOp::Mul ? PYR_BIN_MUL : PYR_BIN_DIV;
                emit(PYR_OP_BINARY, out, deps, zero, zero, zero, zero, l.reg, r.reg, VT[kind], op);
                break;
            }
            case ExprKind::Clamp: {
                uint32_t vd, mnd, mxd;
                const Operand v = try_get_number_value(n.args[0], vd);
                const Operand mn = try_get_number_value(n.args[1], mnd);
                const Operand mx = try_get_number_value(n.args[2], mxd);
                const uint32_t out = next_reg(RN), deps = vd | mnd | mxd;
                emit(PYR_OP_CLAMP, out, deps, v, mn, mx, zero);
                done(e, RN, out, deps);
                break;
            }
            default: throw ProjectError("not an expression");
            }
        } catch (const Pending& p) { // unwrap_or_push!, compiler.rs:25-36
            pending.push_back(e);
            pending.push_back(p.child);
        }
    }

    const Status st = status[expression.id()];
    if (!st.done) throw ProjectError("the expression was not compiled to completion");
    uint32_t reg = st.reg;
    uint32_t output_kind;
    if (!vector_output) { // compiler.rs:528-563
        if (st.kind == RV) throw ProjectError("cannot use a vector as a number");
        if (st.kind == RC) {
            uint32_t wld;
            const Operand wl = number_input(wld);
            const uint32_t out = next_reg(RN);
            S.uses_rgb_basis = true;
            emit(PYR_OP_RGB_SPECTRUM, out, st.deps | wld, wl, zero, zero, zero, reg);
            reg = out;
        }
        output_kind = PYR_OUTPUT_NUMBER;
    } else {
        if (st.kind == RN)
            reg = number_register_to(RV, reg, st.deps);
        else if (st.kind == RC)
            reg = rgb_register_to_vector(reg, st.deps);
        output_kind = PYR_OUTPUT_VECTOR;
    }
    if (counts[RN] > PYR_MAX_NUMBER_REGISTERS || counts[RV] > PYR_MAX_VECTOR_REGISTERS || counts[RC] > PYR_MAX_RGB_REGISTERS)
        throw ProjectError("program needs more registers than the GPU VM provides");
    PyrProgram p{};
    p.kind = PYR_PROGRAM_INSTRUCTIONS;
    p.first_instr = (uint32_t)S.instrs.size();
    p.num_instrs = (uint32_t)instructions.size();
    p.output_kind = output_kind;
    p.output_reg = reg;
    p.num_numbers = counts[RN], p.num_vectors = counts[RV], p.num_rgbs = counts[RC];
    S.instrs.insert(S.instrs.end(), instructions.begin(), instructions.end());
    S.programs.push_back(p);
    return (uint32_t)S.programs.size() - 1;
}

// ---- SurfaceMaterial::from_project (materials/mod.rs:90-227) + Material::from_project (:33-46) ------------------------------
std::pair<uint32_t, bool> FlatScene::add_material(const Material& mat) {
    Impl& S = *impl_;
    int32_t normal_map_program = -1;
    if (mat.normal_map) { // a Vector program over NormalInput (no wavelength), materials/mod.rs:41-44
        normal_map_program = (int32_t)compile(*mat.normal_map, false, true);
        S.uses_normal_maps = true;
    }
    struct Item {
        SurfaceMaterial node;
        std::optional<Expression> probability;
    };
    std::vector<Item> stack{Item{mat.surface, std::nullopt}};
    std::vector<PyrComponent> components, emissive;
    while (!stack.empty()) {
        const Item item = stack.back();
        stack.pop_back();
        const MaterialNode* node = item.node.get();
        if (node == nullptr) throw ProjectError("missing material");
        switch (node->kind) {
        case MaterialKind::Emissive:
        case MaterialKind::Diffuse:
        case MaterialKind::Mirror:
        case MaterialKind::Refractive: {
            PyrComponent c{};
            c.probability_program = item.probability ? (int32_t)compile(*item.probability) : -1;
            c.color_program = compile(node->color);
            if (node->kind == MaterialKind::Refractive) {
                c.bsdf = PYR_BSDF_REFRACTIVE;
                c.ior = eval_number(node->ior);
                c.env_ior = node->env_ior ? eval_number(*node->env_ior) : 1.0f;
                c.dispersion = node->dispersion ? eval_number(*node->dispersion) : 0.0f;
                c.env_dispersion = node->env_dispersion ? eval_number(*node->env_dispersion) : 0.0f;
            } else {
                c.bsdf = node->kind == MaterialKind::Emissive ? PYR_BSDF_EMISSIVE : node->kind == MaterialKind::Diffuse ? PYR_BSDF_DIFFUSE : PYR_BSDF_MIRROR;
            }
            components.push_back(c);
            if (node->kind == MaterialKind::Emissive) emissive.push_back(c);
            break;
        }
        case MaterialKind::Mix: {
            const Expression amount = insert_clamp(node->amount, 0.0, 1.0);
            const Expression lhs_probability = item.probability ? insert_mul(*item.probability, amount) : amount;
            stack.push_back(Item{node->lhs, lhs_probability});
            stack.push_back(Item{node->rhs, insert_sub(1.0, lhs_probability)});
            break;
        }
        case MaterialKind::Add:
            stack.push_back(Item{node->lhs, item.probability});
            stack.push_back(Item{node->rhs, item.probability});
            break;
        }
    }
    for (auto& c : components) c.selection_compensation = (float)components.size();
    for (auto& c : emissive) c.selection_compensation = (float)emissive.size();
    PyrMaterial m{};
    m.first_component = (uint32_t)S.components.size();
    m.num_components = (uint32_t)components.size();
    S.components.insert(S.components.end(), components.begin(), components.end());
    m.first_emissive = (uint32_t)S.components.size();
    m.num_emissive = (uint32_t)emissive.size();
    S.components.insert(S.components.end(), emissive.begin(), emissive.end());
    m.normal_map_program = normal_map_program;
    S.materials.push_back(m);
    return {(uint32_t)S.materials.size() - 1, !emissive.empty()};
}

void FlatScene::add_triangle(const float positions[9], const float normals[9], const float uvs[6], uint32_t material, const float frames[12]) {
    Impl& S = *impl_;
    S.tri_positions.insert(S.tri_positions.end(), positions, positions + 9);
    S.tri_normals.insert(S.tri_normals.end(), normals, normals + 9);
    if (uvs != nullptr)
        S.tri_uvs.insert(S.tri_uvs.end(), uvs, uvs + 6);
    else
        S.tri_uvs.insert(S.tri_uvs.end(), 6, 0.0f);
    static const float identity_frames[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
    const float* f = frames != nullptr ? frames : identity_frames;
    S.tri_frames.insert(S.tri_frames.end(), f, f + 12);
    S.tri_material.push_back(material);
}

// make_triangle (world.rs:308-374) + Shape::scale / transform (shapes/mod.rs:290-344)
void FlatScene::Impl::add_mesh_triangle(const MeshData& mesh, const std::vector<MeshData::Index>& poly, uint32_t material, float scale_factor, const Mat4& xform,
                                        FlatScene& self) {
    V3 v[3], n[3];
    float uv[3][2];
    bool all_normals = true;
    for (int k = 0; k < 3; ++k) {
        const MeshData::Index ix = poly[k];
        if (ix.position < 0 || (size_t)ix.position * 3 + 2 >= mesh.position.size()) throw ProjectError("mesh: vertex index out of range");
        v[k] = V3{mesh.position[3 * ix.position], mesh.position[3 * ix.position + 1], mesh.position[3 * ix.position + 2]};
        all_normals = all_normals && ix.normal >= 0;
        if (ix.texture >= 0) {
            if ((size_t)ix.texture * 2 + 1 >= mesh.texture.size()) throw ProjectError("mesh: texture index out of range");
            uv[k][0] = mesh.texture[2 * ix.texture], uv[k][1] = mesh.texture[2 * ix.texture + 1];
        } else {
            uv[k][0] = uv[k][1] = 0.0f;
        }
    }
    if (all_normals) {
        for (int k = 0; k < 3; ++k) {
            if ((size_t)poly[k].normal * 3 + 2 >= mesh.normal.size()) throw ProjectError("mesh: normal index out of range");
            n[k] = V3{mesh.normal[3 * poly[k].normal], mesh.normal[3 * poly[k].normal + 1], mesh.normal[3 * poly[k].normal + 2]};
        }
    } else {
        const V3 flat = normalize(cross(sub(v[1], v[0]), sub(v[2], v[0])));
        n[0] = n[1] = n[2] = flat;
    }
    // tangent space from the uv deltas, world.rs:337-346 (before scale / transform)
    const V3 dp1 = sub(v[1], v[0]), dp2 = sub(v[2], v[0]);
    const float dt1x = uv[1][0] - uv[0][0], dt1y = uv[1][1] - uv[0][1], dt2x = uv[2][0] - uv[0][0], dt2y = uv[2][1] - uv[0][1];
    const float r = 1.0f / (dt1x * dt2y - dt1y * dt2x);
    const V3 tangent = scale(sub(scale(dp1, dt2y), scale(dp2, dt1y)), r);
    const V3 bitangent = scale(sub(scale(dp2, dt1x), scale(dp1, dt2x)), r);
    float positions[9], normals[9], uvs[6], frames[12];
    for (int k = 0; k < 3; ++k) {
        const Quat frame = quat_from_cols(tangent, bitangent, n[k]);
        V3 nt;
        Quat ft;
        normal_transform(xform, n[k], frame, nt, ft);
        const V3 p = transform_point(xform, scale(v[k], scale_factor));
        positions[3 * k] = p.x, positions[3 * k + 1] = p.y, positions[3 * k + 2] = p.z;
        normals[3 * k] = nt.x, normals[3 * k + 1] = nt.y, normals[3 * k + 2] = nt.z;
        uvs[2 * k] = uv[k][0], uvs[2 * k + 1] = uv[k][1];
        frames[4 * k] = ft.s, frames[4 * k + 1] = ft.x, frames[4 * k + 2] = ft.y, frames[4 * k + 3] = ft.z;
    }
    self.add_triangle(positions, normals, uvs, material, frames);
}

void FlatScene::Impl::add_mesh(size_t index, const Mesh& mesh, FlatScene& self) { // world.rs:184-236
    std::shared_ptr<const MeshData> data = mesh.data;
    if (!data) {
        const bool absolute = !mesh.file.empty() && mesh.file[0] == '/';
        data = std::make_shared<MeshData>(load_obj(absolute ? mesh.file : base_dir + "/" + mesh.file));
    }
    std::map<std::string, Material> materials = mesh.materials;
    for (const MeshData::Object& o : data->objects) {
        auto it = materials.find(o.name);
        if (it == materials.end()) throw ProjectError("objects[" + std::to_string(index) + "]: missing material for '" + o.name + "'");
        const auto added = self.add_material(it->second);
        materials.erase(it);
        const Mat4 xform = mesh.transform ? eval_transform(*mesh.transform) : identity();
        const float scale_factor = mesh.scale ? eval_number(*mesh.scale) : 1.0f;
        for (const auto& poly : o.polys) {
            if (poly.size() != 3) continue; // only `[x, y, z]` polys are taken, world.rs:218-232
            add_mesh_triangle(*data, poly, added.first, scale_factor, xform, self);
            if (added.second) {
                PyrLamp l{};
                l.kind = PYR_LAMP_SHAPE, l.shape_kind = PYR_SHAPE_TRIANGLE, l.shape_index = (uint32_t)tri_material.size() - 1;
                lamps.push_back(l);
            }
        }
    }
}

// ---- World::from_project (world.rs:39-271) ----------------------------------------------------------------------------------
void FlatScene::add_world(const WorldProject& world, const std::string& base_dir) {
    Impl& S = *impl_;
    S.base_dir = base_dir;
    S.sky_program = compile(world.sky ? *world.sky : Expression(0.0));
    for (size_t i = 0; i < world.objects.size(); ++i) {
        const WorldObject& obj = world.objects[i];
        switch (obj.kind) {
        case WorldObject::Kind::Sphere: {
            const auto added = add_material(obj.sphere.material);
            const V4 position = eval_vector(obj.sphere.position);
            const float radius = eval_number(obj.sphere.radius);
            const V4 ts = obj.sphere.texture_scale ? eval_vector(*obj.sphere.texture_scale) : V4{1, 1, 1, 1};
            S.spheres.insert(S.spheres.end(), {position.x, position.y, position.z, radius});
            S.sphere_tex_scale.insert(S.sphere_tex_scale.end(), {ts.x, ts.y});
            S.sphere_material.push_back(added.first);
            if (added.second) {
                PyrLamp l{};
                l.kind = PYR_LAMP_SHAPE, l.shape_kind = PYR_SHAPE_SPHERE, l.shape_index = (uint32_t)S.sphere_material.size() - 1;
                S.lamps.push_back(l);
            }
            break;
        }
        case WorldObject::Kind::Plane: {
            const auto added = add_material(obj.plane.material);
            const V3 normal = normalize(xyz(eval_vector(obj.plane.normal)));
            const V4 origin = eval_vector(obj.plane.origin);
            const V4 ts = obj.plane.texture_scale ? eval_vector(*obj.plane.texture_scale) : V4{1, 1, 1, 1};
            S.planes.insert(S.planes.end(), {origin.x, origin.y, origin.z, normal.x, normal.y, normal.z, ts.x, ts.y});
            const V3 z = normalize(ortho(normal)); // math::utils::basis, math.rs:119-123
            const V3 y = normalize(cross(z, normal));
            const Quat q = quat_from_cols(y, z, normal); // world.rs:95-99
            S.plane_frames.insert(S.plane_frames.end(), {q.s, q.x, q.y, q.z});
            S.plane_material.push_back(added.first);
            break;
        }
        case WorldObject::Kind::Mesh: S.add_mesh(i, obj.mesh, *this); break;
        case WorldObject::Kind::DirectionalLight: {
            PyrLamp l{};
            l.kind = PYR_LAMP_DIRECTIONAL;
            const V4 d = eval_vector(obj.directional.direction);
            l.v[0] = d.x, l.v[1] = d.y, l.v[2] = d.z;
            l.width = eval_number(obj.directional.width);
            l.color_program = compile(obj.directional.color);
            S.lamps.push_back(l);
            break;
        }
        case WorldObject::Kind::PointLight: {
            PyrLamp l{};
            l.kind = PYR_LAMP_POINT;
            const V4 p = eval_vector(obj.point.position);
            l.v[0] = p.x, l.v[1] = p.y, l.v[2] = p.z;
            l.color_program = compile(obj.point.color);
            S.lamps.push_back(l);
            break;
        }
        }
    }
}

const PyrSceneDesc& FlatScene::desc() {
    Impl& S = *impl_;
    PyrSceneDesc& d = S.desc;
    d = PyrSceneDesc{};
    d.num_triangles = (uint32_t)S.tri_material.size();
    d.tri_positions = S.tri_positions.data();
    d.tri_normals = S.tri_normals.data();
    d.tri_uvs = S.tri_uvs.data();
    d.tri_material = S.tri_material.data();
    d.num_spheres = (uint32_t)S.sphere_material.size();
    d.spheres = S.spheres.data();
    d.sphere_tex_scale = S.sphere_tex_scale.data();
    d.sphere_material = S.sphere_material.data();
    d.num_planes = (uint32_t)S.plane_material.size();
    d.planes = S.planes.data();
    d.plane_material = S.plane_material.data();
    d.num_lamps = (uint32_t)S.lamps.size(), d.lamps = S.lamps.data();
    d.num_materials = (uint32_t)S.materials.size(), d.materials = S.materials.data();
    d.num_components = (uint32_t)S.components.size(), d.components = S.components.data();
    d.num_programs = (uint32_t)S.programs.size(), d.programs = S.programs.data();
    d.num_instrs = (uint32_t)S.instrs.size(), d.instrs = S.instrs.data();
    d.num_spectra = (uint32_t)S.spectra.size(), d.spectra = S.spectra.data();
    d.num_spectrum_floats = (uint32_t)S.spectrum_data.size(), d.spectrum_data = S.spectrum_data.data();
    if (S.uses_rgb_basis) { // crate::rgb::response::RGB, build.rs:18-59
        if (S.rgb_basis.empty()) S.rgb_basis = table(k_rgb_basis_bits, (size_t)k_rgb_basis_rows * 3);
        d.rgb_basis = S.rgb_basis.data();
        d.rgb_basis_count = k_rgb_basis_rows;
        d.rgb_basis_min = k_rgb_min, d.rgb_basis_max = k_rgb_max;
    }
    d.sky_program = S.sky_program;
    if (S.uses_normal_maps && d.num_triangles) d.tri_frames = S.tri_frames.data();
    if (d.num_planes) d.plane_frames = S.plane_frames.data();
    if (!S.textures.empty()) {
        d.num_textures = (uint32_t)S.textures.size(), d.textures = S.textures.data();
        d.num_texture_floats = S.texture_data.size(), d.texture_data = S.texture_data.data();
    }
    return d;
}

// ================================================================================================ OBJ ingest
// The `obj` crate's data model (0.10.2): objects -> polys of IndexTuple(v, vt?, vn?); negative indices count from the end.
MeshData load_obj(const std::string& path) {
    std::ifstream f(path);
    if (!f) throw ProjectError("could not open " + path);
    MeshData mesh;
    MeshData::Object* current = nullptr;
    std::string line;
    auto index = [](const std::string& token, size_t count) -> int32_t {
        if (token.empty()) return -1;
        const long k = std::stol(token);
        return (int32_t)(k > 0 ? k - 1 : (long)count + k);
    };
    while (std::getline(f, line)) {
        std::istringstream in(line);
        std::string tag;
        if (!(in >> tag) || tag[0] == '#') continue;
        if (tag == "v") {
            float x = 0, y = 0, z = 0;
            in >> x >> y >> z;
            mesh.position.insert(mesh.position.end(), {x, y, z});
        } else if (tag == "vt") {
            float u = 0, v = 0;
            in >> u;
            if (!(in >> v)) v = 0.0f;
            mesh.texture.insert(mesh.texture.end(), {u, v});
        } else if (tag == "vn") {
            float x = 0, y = 0, z = 0;
            in >> x >> y >> z;
            mesh.normal.insert(mesh.normal.end(), {x, y, z});
        } else if (tag == "o") {
            std::string name;
            in >> name;
            mesh.objects.push_back(MeshData::Object{name, {}});
            current = &mesh.objects.back();
        } else if (tag == "f") {
            std::vector<MeshData::Index> poly;
            std::string tok;
            while (in >> tok) {
                std::string fields[3];
                size_t k = 0, start = 0;
                for (size_t i = 0; i <= tok.size() && k < 3; ++i)
                    if (i == tok.size() || tok[i] == '/') {
                        fields[k++] = tok.substr(start, i - start);
                        start = i + 1;
                    }
                poly.push_back(MeshData::Index{index(fields[0], mesh.position.size() / 3), index(fields[1], mesh.texture.size() / 2), index(fields[2], mesh.normal.size() / 3)});
            }
            if (current == nullptr) {
                mesh.objects.push_back(MeshData::Object{"default", {}});
                current = &mesh.objects.back();
            }
            current->polys.push_back(std::move(poly));
        }
    }
    return mesh;
}

// ================================================================================================ World / Camera / Renderer / Film
std::unique_ptr<World> World::from_project(const WorldProject& world, const std::string& base_dir) {
    std::unique_ptr<World> w(new World());
    w->flat_.add_world(world, base_dir);
    return w;
}
World::~World() {
    for (auto& kv : scenes_) pyr_scene_destroy(kv.second);
}
PyrScene* World::scene(int device, int copy) {
    auto it = scenes_.find({device, copy});
    if (it != scenes_.end()) return it->second;
    PyrScene* handle = nullptr;
    check_status(pyr_scene_create(&flat_.desc(), device, &handle));
    scenes_[{device, copy}] = handle;
    return handle;
}

std::vector<PyrHit> World::intersect(const std::vector<float>& rays, int device, PyrCounters* counters) {
    if (rays.size() % 6 != 0) throw ProjectError("intersect: rays must hold six floats each (origin, direction)");
    std::vector<PyrHit> hits(rays.size() / 6);
    float ms = 0.0f;
    check_status(pyr_scene_intersect(scene(device), rays.data(), (uint32_t)hits.size(), hits.data(), &ms, counters));
    return hits;
}

Camera Camera::from_project(const CameraProject& cam) { // cameras.rs:30-55
    Camera out;
    const float fov = eval_number(cam.fov);
    const float half = (fov * 0.5f) * (float)(M_PI / 180.0); // cgmath Deg -> Rad
    out.c.view_plane = std::cos(half) / std::sin(half);
    const Mat4 m = eval_transform(cam.transform);
    std::memcpy(out.c.cam_to_world, m.m, sizeof(m.m));
    out.c.focus_distance = cam.focus_distance ? eval_number(*cam.focus_distance) : 1.0f;
    out.c.aperture = cam.aperture ? eval_number(*cam.aperture) : 0.0f;
    return out;
}

Renderer Renderer::from_project(const RendererProject& r) { // renderer/mod.rs:31-75 (defaults :63-75)
    Renderer out;
    out.pixel_samples = r.pixel_samples;
    out.bounces = r.bounces.value_or(8);
    out.light_samples = r.light_samples.value_or(4);
    out.spectrum_samples = r.spectrum_samples.value_or(10);
    out.spectrum_bins = r.spectrum_resolution.value_or(64);
    out.tile_size = r.tile_size.value_or(32);
    return out;
}

namespace {
struct Trampoline {
    const std::function<void(Progress)>* fn;
};
void on_status_trampoline(void* user, uint8_t percent, const char* message) {
    const Trampoline* t = static_cast<const Trampoline*>(user);
    (*t->fn)(Progress{percent, message});
}
} // namespace

void Renderer::render(Film& film, const Camera& camera, World& world, const std::function<void(Progress)>& on_status, int device, PyrCounters* counters) const {
    PyrRenderParams p{};
    p.bounces = bounces, p.pixel_samples = pixel_samples, p.light_samples = light_samples, p.spectrum_samples = spectrum_samples, p.tile_size = tile_size;
    p.flags = counters != nullptr ? PYR_FLAG_COUNTERS : 0u;
    p.seed = seed;
    const PyrFilmDesc desc = film.desc();
    Trampoline t{&on_status};
    check_status(pyr_render_simple(world.scene(device), &camera.c, &desc, &p, film.grains.data(), on_status ? on_status_trampoline : nullptr, on_status ? &t : nullptr));
    if (counters != nullptr) check_status(pyr_scene_counters(world.scene(device), counters));
}

void Renderer::render(Film& film, const Camera& camera, World& world, const std::vector<int>& devices, const std::function<void(Progress)>& on_status) const {
    if (devices.empty()) throw ProjectError("render: no device given");
    PyrRenderParams p{};
    p.bounces = bounces, p.pixel_samples = pixel_samples, p.light_samples = light_samples, p.spectrum_samples = spectrum_samples, p.tile_size = tile_size;
    p.seed = seed;
    std::vector<PyrScene*> scenes;
    std::map<int, int> seen;
    for (int device : devices) scenes.push_back(world.scene(device, seen[device]++));
    const PyrFilmDesc desc = film.desc();
    Trampoline t{&on_status};
    check_status(pyr_render_simple_multi(scenes.data(), (uint32_t)scenes.size(), &camera.c, &desc, &p, film.grains.data(), on_status ? on_status_trampoline : nullptr,
                                         on_status ? &t : nullptr));
}

Film::Film(uint32_t width_, uint32_t height_, uint32_t grains_per_pixel, float wavelength_start_, float wavelength_end)
    : width(width_), height(height_), bins(grains_per_pixel), wavelength_start(wavelength_start_), wavelength_width(wavelength_end - wavelength_start_),
      grains((size_t)width_ * height_ * grains_per_pixel, PyrGrain{0.0f, 0.0f}) {}
PyrFilmDesc Film::desc() const { return PyrFilmDesc{width, height, bins, wavelength_start, wavelength_width}; }
double Film::total_weight() const {
    double s = 0.0;
    for (const PyrGrain& g : grains) s += g.weight;
    return s;
}

// ---- image.filter / image.white (main.rs:190-238, :470-518): f32 evaluation of a wavelength-only expression -------------------
namespace {
float array_get(const float* data, size_t n, float mn, float mx, float w) { // Spectrum::Array::get, project/spectra.rs:32-55
    if (n == 0) return 0.0f;
    if (w <= mn) return data[0];
    if (w >= mx) return data[n - 1];
    const float normalized = (w - mn) / (mx - mn);
    const float fi = normalized * ((float)n - 1.0f);
    const float fl = std::trunc(fi);
    const size_t i0 = (size_t)fl;
    const float mixf = fi - fl;
    return data[i0] * (1.0f - mixf) + data[i0 + 1] * mixf;
}
float curve_get(const std::vector<float>& pts, float w) { // Interpolated::get, math.rs:22-72: zero at and outside the end points
    const size_t count = pts.size() / 2;
    if (count == 0 || pts[0] >= w || pts[2 * (count - 1)] <= w) return 0.0f;
    size_t lo = 0, hi = count - 1;
    while (hi > lo + 1) {
        const size_t mid = (lo + hi) / 2;
        if (pts[2 * mid] == w) return pts[2 * mid + 1];
        if (pts[2 * mid] > w)
            hi = mid;
        else
            lo = mid;
    }
    const float x0 = pts[2 * lo], y0 = pts[2 * lo + 1], x1 = pts[2 * hi], y1 = pts[2 * hi + 1];
    return y0 + (y1 - y0) * ((w - x0) / (x1 - x0));
}
float d65_at(float w) {
    static const std::vector<float> t = table(k_d65_bits, k_d65_rows);
    return array_get(t.data(), t.size(), k_light_min, k_light_max, w);
}
} // namespace

float evaluate_at(const Expression& e, float w) {
    const ExprNode& n = e.node();
    switch (n.kind) {
    case ExprKind::Number: return (float)n.number;
    case ExprKind::Spectrum:
        switch (n.format) {
        case SpectrumFormat::BuiltinD65: return d65_at(w);
        case SpectrumFormat::BuiltinA: {
            static const std::vector<float> t = table(k_a_bits, k_a_rows);
            return array_get(t.data(), t.size(), k_light_min, k_light_max, w);
        }
        case SpectrumFormat::Array: return array_get(n.points.data(), n.points.size(), n.min, n.max, w);
        default: return curve_get(n.points, w);
        }
    case ExprKind::Blackbody: { // math.rs:177-182
        const float temperature = evaluate_at(n.args[0], w);
        const float wl = w * 1.0e-9f;
        const float a2 = wl * wl, a4 = a2 * a2;
        const float power = 3.74183e-16f * (1.0f / (wl * a4));
        return power / ((float)std::exp((double)(1.4388e-2f / (wl * temperature))) - 1.0f);
    }
    case ExprKind::Binary: {
        const float l = evaluate_at(n.args[0], w), r = evaluate_at(n.args[1], w);
        switch (n.op) {
        case BinaryOp::Add: return l + r;
        case BinaryOp::Sub: return l - r;
        case BinaryOp::Mul: return l * r;
        default: return l / r;
        }
    }
    case ExprKind::Mix: {
        const float amount = std::min(std::max(evaluate_at(n.args[2], w), 0.0f), 1.0f);
        return evaluate_at(n.args[0], w) * (1.0f - amount) + evaluate_at(n.args[1], w) * amount;
    }
    case ExprKind::Clamp: return std::max(std::min(evaluate_at(n.args[0], w), evaluate_at(n.args[2], w)), evaluate_at(n.args[1], w));
    case ExprKind::Fresnel: throw ProjectError("the surface normal cannot be used while sampling a constant spectrum");
    default: throw ProjectError("cannot sample this expression as a spectrum");
    }
}

std::vector<uint8_t> Film::develop(const std::optional<Expression>& filter, const std::optional<Expression>& white, float step_size, int device) const {
    // wl_i of spectrum_to_tristimulus (main.rs:393-411): start at the span's minimum, add step_size in f32 while below the maximum
    const float lo = wavelength_start, hi = wavelength_start + wavelength_width;
    std::vector<float> wl{lo};
    while (wl.back() < hi) wl.push_back(wl.back() + step_size);
    static const std::vector<float> xyz_table = table(k_xyz_bits, (size_t)k_xyz_rows * 3);
    PyrDevelopParams p{};
    p.step_size = step_size, p.xyz_scale = 3.444f, p.sample_count = (uint32_t)wl.size();
    p.xyz_table = xyz_table.data(), p.xyz_count = k_xyz_rows, p.xyz_min = k_xyz_min, p.xyz_max = k_xyz_max;
    std::vector<float> filter_values, white_div, white_mul;
    if (filter) { // main.rs:197-202
        for (float w : wl) filter_values.push_back(evaluate_at(*filter, w));
        p.filter = filter_values.data();
    }
    if (white) { // main.rs:204-222
        float mx = 0.0f, d65_mx = 0.0f;
        for (float w = lo; w < hi; w = w + 1.0f) {
            mx = std::max(mx, evaluate_at(*white, w));
            d65_mx = std::max(d65_mx, d65_at(w));
        }
        for (float w : wl) {
            white_div.push_back(std::max(evaluate_at(*white, w) / mx, 0.000001f));
            white_mul.push_back(d65_at(w) / d65_mx);
        }
        p.white_div = white_div.data(), p.white_mul = white_mul.data();
    }
    std::vector<uint8_t> out((size_t)width * height * 3, 0);
    const PyrFilmDesc d = desc();
    check_status(pyr_film_develop(&d, grains.data(), &p, out.data(), device));
    return out;
}

// Minimal PNG writer (8-bit RGB, stored deflate blocks) -- the image::save of main.rs:327.
void save_png(const std::string& path, const std::vector<uint8_t>& rgb, uint32_t width, uint32_t height) {
    if (rgb.size() != (size_t)width * height * 3) throw ProjectError("save_png: buffer size does not match the image size");
    static uint32_t crc_table[256];
    static bool crc_ready = false;
    if (!crc_ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            crc_table[i] = c;
        }
        crc_ready = true;
    }
    auto be32 = [](std::vector<uint8_t>& v, uint32_t x) {
        for (int s = 24; s >= 0; s -= 8) v.push_back((uint8_t)(x >> s));
    };
    std::vector<uint8_t> raw;
    raw.reserve((size_t)height * (width * 3 + 1));
    for (uint32_t y = 0; y < height; ++y) {
        raw.push_back(0);
        raw.insert(raw.end(), rgb.begin() + (size_t)y * width * 3, rgb.begin() + (size_t)(y + 1) * width * 3);
    }
    std::vector<uint8_t> z{0x78, 0x01};
    uint32_t a = 1, b = 0;
    for (uint8_t c : raw) {
        a = (a + c) % 65521u;
        b = (b + a) % 65521u;
    }
    for (size_t pos = 0; pos < raw.size() || pos == 0;) {
        const size_t n = std::min<size_t>(65535, raw.size() - pos);
        const bool last = pos + n >= raw.size();
        z.push_back(last ? 1 : 0);
        z.push_back((uint8_t)(n & 0xff)), z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xff)), z.push_back((uint8_t)((~n >> 8) & 0xff));
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        pos += n;
        if (last) break;
    }
    be32(z, (b << 16) | a);
    std::vector<uint8_t> file{0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    auto chunk = [&](const char* tag, const std::vector<uint8_t>& data) {
        be32(file, (uint32_t)data.size());
        std::vector<uint8_t> body(tag, tag + 4);
        body.insert(body.end(), data.begin(), data.end());
        uint32_t c = 0xFFFFFFFFu;
        for (uint8_t x : body) c = crc_table[(c ^ x) & 0xff] ^ (c >> 8);
        file.insert(file.end(), body.begin(), body.end());
        be32(file, c ^ 0xFFFFFFFFu);
    };
    std::vector<uint8_t> ihdr;
    be32(ihdr, width), be32(ihdr, height);
    ihdr.insert(ihdr.end(), {8, 2, 0, 0, 0});
    chunk("IHDR", ihdr);
    chunk("IDAT", z);
    chunk("IEND", {});
    std::ofstream f(path, std::ios::binary);
    if (!f) throw ProjectError("could not write " + path);
    f.write(reinterpret_cast<const char*>(file.data()), (std::streamsize)file.size());
}

} // namespace pyrite

// ================================================================================================ test hooks (C ABI)
extern "C" int pyrh_test_png(const char* path, const uint8_t* rgb, uint32_t width, uint32_t height) {
    try {
        pyrite::save_png(path, std::vector<uint8_t>(rgb, rgb + (size_t)width * height * 3), width, height);
        return 0;
    } catch (const std::exception&) {
        return 1;
    }
}

extern "C" int64_t pyrh_test_load_texture(const char* path, int linear, int mono, float* out, uint64_t capacity, uint32_t* width, uint32_t* height) {
    try {
        const std::vector<float> texels = pyrite::load_texture_file(path, linear != 0, mono != 0, *width, *height);
        if (texels.size() <= capacity) std::memcpy(out, texels.data(), texels.size() * sizeof(float));
        return (int64_t)texels.size();
    } catch (const std::exception&) {
        return -1;
    }
}

// canonical scene bytes
extern "C" uint64_t pyrh_serialize_desc(const PyrSceneDesc* d, uint8_t* out, uint64_t capacity) {
    uint64_t size = 0;
    auto put = [&](const void* data, uint64_t bytes) {
        if (out != nullptr && size + bytes <= capacity && bytes != 0) std::memcpy(out + size, data, bytes);
        size += bytes;
    };
    auto section = [&](const char* tag, const void* data, uint64_t count, uint64_t elem) {
        put(tag, 4);
        const uint64_t n = data != nullptr ? count : 0;
        put(&n, 8);
        put(data, n * elem);
    };
    section("TPOS", d->tri_positions, (uint64_t)d->num_triangles * 9, 4);
    section("TNRM", d->tri_normals, (uint64_t)d->num_triangles * 9, 4);
    section("TUVS", d->tri_uvs, (uint64_t)d->num_triangles * 6, 4);
    section("TMAT", d->tri_material, d->num_triangles, 4);
    section("SPHR", d->spheres, (uint64_t)d->num_spheres * 4, 4);
    section("STEX", d->sphere_tex_scale, (uint64_t)d->num_spheres * 2, 4);
    section("SMAT", d->sphere_material, d->num_spheres, 4);
    section("PLAN", d->planes, (uint64_t)d->num_planes * 8, 4);
    section("PMAT", d->plane_material, d->num_planes, 4);
    section("LAMP", d->lamps, d->num_lamps, sizeof(PyrLamp));
    section("MATS", d->materials, d->num_materials, sizeof(PyrMaterial));
    section("COMP", d->components, d->num_components, sizeof(PyrComponent));
    section("PROG", d->programs, d->num_programs, sizeof(PyrProgram));
    section("INST", d->instrs, d->num_instrs, sizeof(PyrInstr));
    section("SPEC", d->spectra, d->num_spectra, sizeof(PyrSpectrum));
    section("SDAT", d->spectrum_data, d->num_spectrum_floats, 4);
    section("RGBB", d->rgb_basis, (uint64_t)d->rgb_basis_count * 3, 4);
    put(&d->rgb_basis_min, 4), put(&d->rgb_basis_max, 4), put(&d->sky_program, 4);
    section("TEXR", d->textures, d->num_textures, sizeof(PyrTexture));
    section("TEXD", d->texture_data, d->num_texture_floats, 4);
    section("TFRM", d->tri_frames, (uint64_t)d->num_triangles * 12, 4);
    section("PFRM", d->plane_frames, (uint64_t)d->num_planes * 4, 4);
    return size;
}
