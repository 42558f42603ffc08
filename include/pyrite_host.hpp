// pyrite_host.hpp -- C++ host side above the C ABI of pyrite_gpu.h.
//
// The reference is compiled code (Rust) whose toolchain is absent from this image, so the host layer that mirrors its
// operator surface is C++ (libpyrite_host.so, pyrite_amd/csrc/host/pyrite_host.cpp). It offers, with the reference's names,
// argument meaning and defaults:
//
//   * the project tree the Lua prelude builds               pyrite/src/project/lib.lua:1-309, project/mod.rs:103-252
//       expressions (numbers, vector, rgb, spectrum, blackbody, fresnel, texture, + - * /, mix, clamp, light_source.d65 / a)
//       materials   (emissive, diffuse, mirror, refractive, mix, a + b)
//       objects     (sphere, plane, mesh, directional light, point light), look_at transform, perspective camera
//   * ProgramCompiler::compile                                pyrite/src/program/compiler.rs:48-586 (+ operand coercion :682-968)
//   * SurfaceMaterial::from_project                            pyrite/src/materials/mod.rs:90-227
//   * World::from_project (+ make_triangle, OBJ ingest)        pyrite/src/world.rs:39-271, :308-374
//   * Camera::from_project, Renderer::from_project             pyrite/src/cameras.rs:30-55, renderer/mod.rs:31-75
//   * the seam Renderer::render(film, camera, world)            pyrite/src/renderer/mod.rs:77-111  -> pyr_render_simple
//   * Film (film.rs:9-114) and its development to 8-bit sRGB    pyrite/src/main.rs:190-238, :315-418 -> pyr_film_develop
//
// Nothing here computes radiance: rendering and development are calls into libpyrite_gpu.so, which fails with
// PYR_ERR_DEVICE when no MI355X is present. Errors are reported as pyrite::ProjectError (what the reference reports as a
// project error) or pyrite::GpuError (a non-zero PyrStatus).
#ifndef PYRITE_HOST_HPP
#define PYRITE_HOST_HPP

#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "pyrite_gpu.h"

namespace pyrite {

struct ProjectError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct GpuError : std::runtime_error {
    int status;
    GpuError(int status_, const std::string& what) : std::runtime_error(what), status(status_) {}
};

// ------------------------------------------------------------------------------------------------ expressions
// ComplexExpression / Expression (project/expressions.rs:163-201). A plain number is Expression::Number (f64).
struct ExprNode;
class Expression {
  public:
    Expression(double number = 0.0); // NOLINT: numbers convert implicitly, as in the Lua prelude
    Expression(int number) : Expression((double)number) {}
    explicit Expression(std::shared_ptr<const ExprNode> node) : node_(std::move(node)) {}
    const ExprNode& node() const { return *node_; }
    const ExprNode* id() const { return node_.get(); } // identity: one spectrum / register per Lua table
    bool is_number() const;
    double number() const;
    Expression mix(const Expression& other, const Expression& amount) const;

  private:
    std::shared_ptr<const ExprNode> node_;
};

enum class ExprKind { Number, Vector, Rgb, Spectrum, Fresnel, Blackbody, Binary, Mix, Clamp, ColorTexture, MonoTexture };
enum class BinaryOp { Add, Sub, Mul, Div };
enum class SpectrumFormat { Array, Curve, BuiltinD65, BuiltinA };

struct ExprNode {
    ExprKind kind = ExprKind::Number;
    double number = 0.0;
    BinaryOp op = BinaryOp::Add;
    std::vector<Expression> args; // Vector x y z w | Rgb r g b | Fresnel ior env_ior | Blackbody T | Binary l r | Mix l r amount | Clamp value min max
    // Spectrum (project/spectra.rs:13-24)
    SpectrumFormat format = SpectrumFormat::Array;
    float min = 0.0f, max = 0.0f;
    std::vector<float> points; // Array: values; Curve: (wavelength, value) pairs
    // Textures: linear f32 texels [height][width][channels] (4 for colour, 1 for mono), top row first -- what
    // Texture::from_path leaves in memory (texture.rs:25-85). Decoding image files is the front-end's job.
    uint32_t tex_width = 0, tex_height = 0;
    std::vector<float> texels;
};

Expression operator+(const Expression& a, const Expression& b);
Expression operator-(const Expression& a, const Expression& b);
Expression operator*(const Expression& a, const Expression& b);
Expression operator/(const Expression& a, const Expression& b);
Expression mix(const Expression& lhs, const Expression& rhs, const Expression& amount);             // lib.lua:104-118
Expression clamp(const Expression& value, const Expression& min, const Expression& max);            // expressions.rs:47-63
Expression fresnel(const Expression& ior, const Expression& env_ior = 1.0);                         // lib.lua:120-125
Expression vector(const Expression& x = 0.0, const Expression& y = 0.0, const Expression& z = 0.0, const Expression& w = 0.0); // :128-150
Expression blackbody(const Expression& temperature);                                                // lib.lua:152-157
Expression rgb(const Expression& red = 0.0, const Expression& green = 0.0, const Expression& blue = 0.0); // lib.lua:166-176
Expression spectrum_array(float min, float max, std::vector<float> points);                         // lib.lua:159-164, format = "array"
Expression spectrum_curve(std::vector<std::pair<float, float>> points);                             // format = "curve"
Expression color_texture(uint32_t width, uint32_t height, std::vector<float> rgba_linear);           // lib.lua:178-195
Expression mono_texture(uint32_t width, uint32_t height, std::vector<float> luma_linear);
namespace light_source { // lib.lua:254-258
Expression d65();
Expression a();
} // namespace light_source

// ------------------------------------------------------------------------------------------------ materials
// SurfaceMaterial nodes (project/materials.rs:5-35).
enum class MaterialKind { Emissive, Diffuse, Mirror, Refractive, Mix, Add };
struct MaterialNode;
class SurfaceMaterial {
  public:
    SurfaceMaterial() = default;
    explicit SurfaceMaterial(std::shared_ptr<const MaterialNode> node) : node_(std::move(node)) {}
    const MaterialNode* get() const { return node_.get(); }
    SurfaceMaterial mix(const SurfaceMaterial& other, const Expression& amount) const;

  private:
    std::shared_ptr<const MaterialNode> node_;
};
struct MaterialNode {
    MaterialKind kind = MaterialKind::Diffuse;
    Expression color;
    Expression ior = 1.0;
    std::optional<Expression> dispersion, env_ior, env_dispersion;
    SurfaceMaterial lhs, rhs;
    Expression amount;
};
namespace material { // lib.lua:231-252
SurfaceMaterial diffuse(const Expression& color);
SurfaceMaterial emissive(const Expression& color);
SurfaceMaterial mirror(const Expression& color);
SurfaceMaterial refractive(const Expression& color, const Expression& ior, std::optional<Expression> dispersion = std::nullopt,
                           std::optional<Expression> env_ior = std::nullopt, std::optional<Expression> env_dispersion = std::nullopt);
} // namespace material
SurfaceMaterial operator+(const SurfaceMaterial& a, const SurfaceMaterial& b); // lib.lua:88-90
SurfaceMaterial mix(const SurfaceMaterial& lhs, const SurfaceMaterial& rhs, const Expression& amount);

struct Material { // project::Material: {surface, normal_map}
    SurfaceMaterial surface;
    std::optional<Expression> normal_map;
    Material() = default;
    Material(SurfaceMaterial s) : surface(std::move(s)) {} // NOLINT
    Material(SurfaceMaterial s, Expression n) : surface(std::move(s)), normal_map(std::move(n)) {}
};

// ------------------------------------------------------------------------------------------------ transforms, camera, objects
struct LookAt { // Transform::LookAt, project/mod.rs:243-266
    Expression from = vector(), to = vector();
    std::optional<Expression> up; // default (0, 1, 0)
};
namespace transform {
inline LookAt look_at(Expression from, Expression to, std::optional<Expression> up = std::nullopt) { return LookAt{std::move(from), std::move(to), std::move(up)}; }
} // namespace transform

// The `obj` crate's data model (0.10.2): objects -> polygons of (position, texture?, normal?) index tuples.
struct MeshData {
    struct Index {
        int32_t position = -1, texture = -1, normal = -1; // -1 = absent
    };
    struct Object {
        std::string name;
        std::vector<std::vector<Index>> polys;
    };
    std::vector<float> position; // [n][3]
    std::vector<float> texture;  // [n][2]
    std::vector<float> normal;   // [n][3]
    std::vector<Object> objects;
};
MeshData load_obj(const std::string& path);

struct Sphere {
    Expression position, radius;
    Material material;
    std::optional<Expression> texture_scale;
};
struct Plane {
    Expression origin, normal;
    Material material;
    std::optional<Expression> texture_scale;
};
struct Mesh {
    std::string file;                          // OBJ path, relative to the project directory ...
    std::shared_ptr<const MeshData> data;      // ... or geometry already in memory
    std::map<std::string, Material> materials; // by OBJ object name (world.rs:199-208)
    std::optional<Expression> scale;
    std::optional<LookAt> transform;
};
struct DirectionalLight {
    Expression direction, width, color;
};
struct PointLight {
    Expression position, color;
};
struct WorldObject { // project::WorldObject, project/mod.rs:169-203
    enum class Kind { Sphere, Plane, Mesh, DirectionalLight, PointLight } kind;
    Sphere sphere;
    Plane plane;
    Mesh mesh;
    DirectionalLight directional;
    PointLight point;
    WorldObject(Sphere s) : kind(Kind::Sphere), sphere(std::move(s)) {}                     // NOLINT
    WorldObject(Plane p) : kind(Kind::Plane), plane(std::move(p)) {}                         // NOLINT
    WorldObject(Mesh m) : kind(Kind::Mesh), mesh(std::move(m)) {}                            // NOLINT
    WorldObject(DirectionalLight l) : kind(Kind::DirectionalLight), directional(std::move(l)) {} // NOLINT
    WorldObject(PointLight l) : kind(Kind::PointLight), point(std::move(l)) {}               // NOLINT
};
struct WorldProject { // project::World, project/mod.rs:163-167
    std::optional<Expression> sky;
    std::vector<WorldObject> objects;
};
struct CameraProject { // project::Camera::Perspective, project/mod.rs:120-129
    LookAt transform;
    Expression fov = 45.0;
    std::optional<Expression> focus_distance, aperture;
};
struct RendererProject { // project::Renderer::Simple + RendererShared, project/mod.rs:131-161
    uint32_t pixel_samples = 1;
    std::optional<uint32_t> bounces, light_samples, spectrum_samples, spectrum_resolution, tile_size;
};
struct ImageProject { // project::Image, project/mod.rs:111-118
    uint32_t width = 0, height = 0;
    std::optional<Expression> filter, white;
};
struct Project {
    ImageProject image;
    CameraProject camera;
    RendererProject renderer;
    WorldProject world;
};

// ------------------------------------------------------------------------------------------------ the frozen scene
// What World::from_project + Resources hold, flattened into the arrays PyrSceneDesc points at.
class FlatScene {
  public:
    FlatScene();
    ~FlatScene();
    FlatScene(const FlatScene&) = delete;
    FlatScene& operator=(const FlatScene&) = delete;

    // ProgramCompiler::compile. Returns the program's index.
    uint32_t compile(const Expression& expression, bool allow_wavelength = true, bool vector_output = false);
    // Material::from_project: (material index, has emissive components)
    std::pair<uint32_t, bool> add_material(const Material& material);
    void add_world(const WorldProject& world, const std::string& base_dir = ".");
    void add_triangle(const float positions[9], const float normals[9], const float uvs[6], uint32_t material, const float frames[12] = nullptr);

    const PyrSceneDesc& desc(); // borrows this object's arrays
    size_t num_triangles() const;
    size_t num_spheres() const;
    size_t num_planes() const;

  private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

class World { // world.rs:31-36
  public:
    static std::unique_ptr<World> from_project(const WorldProject& world, const std::string& base_dir = ".");
    ~World();
    PyrScene* scene(int device = 0, int copy = 0); // created on first use (BVH build + upload); `copy` > 0: a further scene on the same device
    // World::intersect (world.rs:273-299) for a batch of rays, [n][6] = origin, direction: closest hits, on the GPU
    std::vector<PyrHit> intersect(const std::vector<float>& rays, int device = 0, PyrCounters* counters = nullptr);
    FlatScene& flat() { return flat_; }
    size_t num_objects() { return flat_.num_triangles() + flat_.num_spheres() + flat_.num_planes(); } // world.rs:251-254

  private:
    World() = default;
    FlatScene flat_;
    std::map<std::pair<int, int>, PyrScene*> scenes_;
};

struct Camera { // cameras.rs:20-27
    PyrCamera c{};
    static Camera from_project(const CameraProject& camera);
};

class Film { // film.rs:9-18
  public:
    Film(uint32_t width, uint32_t height, uint32_t grains_per_pixel = 64, float wavelength_start = 380.0f, float wavelength_end = 780.0f);
    PyrFilmDesc desc() const;
    uint32_t width, height, bins;
    float wavelength_start, wavelength_width;
    std::vector<PyrGrain> grains; // (x + y * width) * bins + bin, film.rs:56
    double total_weight() const;
    // main.rs:315-327: develop into 8-bit sRGB [height][width][3] on the GPU, with the image's filter / white programs.
    std::vector<uint8_t> develop(const std::optional<Expression>& filter = std::nullopt, const std::optional<Expression>& white = std::nullopt,
                                 float step_size = 2.0f, int device = 0) const;
};
void save_png(const std::string& path, const std::vector<uint8_t>& rgb, uint32_t width, uint32_t height);

struct Progress { // renderer/mod.rs:229-232
    uint8_t progress;
    const char* message;
};

class Renderer { // renderer/mod.rs:18-28, Algorithm::Simple
  public:
    uint32_t bounces = 8, pixel_samples = 1, light_samples = 4, spectrum_samples = 10, spectrum_bins = 64, tile_size = 32;
    float spectrum_span[2] = {380.0f, 780.0f};
    uint64_t seed = 1; // no reference counterpart: the reference seeds from OS entropy (simple.rs:26-28)
    static Renderer from_project(const RendererProject& renderer);
    Film new_film(uint32_t width, uint32_t height) const { return Film(width, height, spectrum_bins, spectrum_span[0], spectrum_span[1]); } // main.rs:190-195
    // The seam: blocking; adds into `film`; `on_status` runs on the calling thread. Returns the kernel counters when asked to.
    void render(Film& film, const Camera& camera, World& world, const std::function<void(Progress)>& on_status = nullptr, int device = 0,
                PyrCounters* counters = nullptr) const;
    // The same seam over several GPUs of this process (pyr_render_simple_multi): the scene is replicated on every listed
    // device, the image's tiles are dealt round-robin, one launch per device, one RCCL gather of the film blocks to
    // devices[0]. What the reference does with its worker threads (renderer/mod.rs:125-189) a host does with its GPUs.
    // A device listed twice is the one-GPU test rig (pyrite_gpu.h).
    void render(Film& film, const Camera& camera, World& world, const std::vector<int>& devices, const std::function<void(Progress)>& on_status = nullptr) const;
};

// ------------------------------------------------------------------------------------------------ project files
// Reads a project file (*.lua): the declarative subset of Lua project files are written in, evaluated against the prelude
// of pyrite/src/project/lib.lua, then typed_nodes' FromLua step (pyrite_amd/csrc/host/lua_project.cpp; main.rs:111-134).
// Image files are decoded by `textures`, which turns (absolute path, linear?, mono?) into linear f32 texels
// ([height][width][4] or [height][width]) -- what Texture::from_path does (texture.rs:25-85). One texture per (file, kind).
using TextureLoader = std::function<std::vector<float>(const std::string& path, bool linear, bool mono, uint32_t& width, uint32_t& height)>;
struct LoadedProject {
    Project project;
    std::string base_dir; // mesh and texture paths are relative to the project file's directory (project/mod.rs:73-76)
};
// The built-in decoder: PNG and baseline JPEG files -> linear texels (pyrite_amd/csrc/host/images.cpp). It is what
// load_project / evaluate_project use when no loader is given.
std::vector<float> load_texture_file(const std::string& path, bool linear, bool mono, uint32_t& width, uint32_t& height);
LoadedProject load_project(const std::string& path, const TextureLoader& textures = nullptr);
LoadedProject evaluate_project(const std::string& text, const std::string& name, const std::string& base_dir, const TextureLoader& textures = nullptr);

// Value of a wavelength-only expression with the VM's f32 arithmetic (image.filter / image.white, main.rs:470-518).
float evaluate_at(const Expression& expression, float wavelength);

} // namespace pyrite

// Canonical byte image of a PyrSceneDesc (every array with its length, in declaration order): equal scenes give equal
// bytes. Test infrastructure for comparing front-ends; returns the size needed, writes at most `capacity` bytes.
extern "C" uint64_t pyrh_serialize_desc(const PyrSceneDesc* desc, uint8_t* out, uint64_t capacity);
// pyrite::save_png through a C entry point (tests). Returns 0 on success.
extern "C" int64_t pyrh_test_load_texture(const char* path, int linear, int mono, float* out, uint64_t capacity, uint32_t* width, uint32_t* height);
extern "C" int pyrh_test_png(const char* path, const uint8_t* rgb, uint32_t width, uint32_t height);

#endif // PYRITE_HOST_HPP
