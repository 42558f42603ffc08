// pyrite_host_tool -- the BASELINE configurations and the example scenes written against the C++ host surface
// (include/pyrite_host.hpp), the way pyrite_amd/scenes.py writes them against the Python surface.
//
//   pyrite_host_tool dump   <scene> <data_dir> <out.bin>                      flatten only (no GPU): canonical scene bytes + camera + renderer
//   pyrite_host_tool dump-project   <project.lua> <texel dir | -> <out.bin>    the same for a project file (lua_project.cpp)
//   pyrite_host_tool render-project <project.lua> <texel dir | -> <seed> <out.png> [film.bin]   what `pyrite project.lua` does (main.rs:46-330)
//   pyrite_host_tool intersect <scene> <data_dir> <rays.f32> <hits.bin>       World::intersect for a ray batch ([n][6] f32 -> PyrHit[n])
//   pyrite_host_tool render <scene> <data_dir> <w> <h> <spp> <seed> <film.bin> [out.png]
//                                                                             Renderer::render on device 0; film as raw {acc, weight} f32
// scenes: c1 c2 spheres diamonds lamps textures      data_dir: pyrite_amd/data (cornell_spectra.json, cornell_box.obj, diamonds.obj)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

#include "pyrite_host.hpp"

using namespace pyrite;

// {"name": {"format": "array", "min": a, "max": b, "points": [...]}, ...} -- the one shape cornell_spectra.json has
static Expression read_spectrum(const std::string& json, const std::string& name) {
    size_t at = json.find("\"" + name + "\"");
    if (at == std::string::npos) throw ProjectError("spectrum " + name + " not found");
    auto number_after = [&](const char* key) {
        size_t k = json.find(key, at);
        return (float)std::strtod(json.c_str() + k + std::strlen(key), nullptr);
    };
    const float mn = number_after("\"min\":"), mx = number_after("\"max\":");
    size_t k = json.find("\"points\":", at);
    k = json.find('[', k) + 1;
    std::vector<float> points;
    for (;;) {
        char* end = nullptr;
        const double v = std::strtod(json.c_str() + k, &end);
        points.push_back((float)v);
        k = (size_t)(end - json.c_str());
        while (json[k] == ' ' || json[k] == ',') ++k;
        if (json[k] == ']') break;
    }
    return spectrum_array(mn, mx, std::move(points));
}

struct Cornell {
    Material light, white, green, red;
    Expression lamp;
};
static Cornell cornell_materials(const std::string& data_dir) { // pyrite/test/cornell/cornell.lua:4-7, :41-51
    std::ifstream f(data_dir + "/cornell_spectra.json");
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string json = ss.str();
    Cornell c;
    c.lamp = read_spectrum(json, "lamp");
    c.light = Material(material::emissive(c.lamp * 3) + material::diffuse(0.78));
    c.white = Material(material::diffuse(read_spectrum(json, "white")));
    c.green = Material(material::diffuse(read_spectrum(json, "green")));
    c.red = Material(material::diffuse(read_spectrum(json, "red")));
    return c;
}
static CameraProject cornell_camera() { // cornell.lua:28-35
    CameraProject cam;
    cam.fov = 37.7;
    cam.transform = transform::look_at(vector(-2.78, -8.0, 2.73), vector(-2.78, 0, 2.73), vector(0, 0, 1));
    return cam;
}

static Project scene_c1(const std::string& data_dir) { // scenes.py c1_spheres
    const Cornell m = cornell_materials(data_dir);
    const double R = 100.0, x0 = -5.56, x1 = 0.0, y1 = 5.592, z0 = 0.0, z1 = 5.488;
    const double cx = (x0 + x1) / 2, cy = y1 / 2, cz = (z0 + z1) / 2;
    Project p;
    p.camera = cornell_camera();
    p.world.objects = {
        Sphere{vector(x0 - R, cy, cz), R, m.red, {}},   Sphere{vector(x1 + R, cy, cz), R, m.green, {}}, Sphere{vector(cx, y1 + R, cz), R, m.white, {}},
        Sphere{vector(cx, cy, z0 - R), R, m.white, {}}, Sphere{vector(cx, cy, z1 + R), R, m.white, {}}, Sphere{vector(-3.7, 3.3, 0.9), 0.9, m.white, {}},
        Sphere{vector(-1.6, 1.7, 0.8), 0.8, m.white, {}},
        Sphere{vector(-2.78, 2.795, 4.9), 0.5, Material(material::emissive(cornell_materials(data_dir).lamp * 3)), {}},
    };
    return p;
}

static Project scene_c2(const std::string& data_dir) { // scenes.py c2_cornell: test/cornell/box.obj, materials per cornell.lua:41-51
    const Cornell m = cornell_materials(data_dir);
    Mesh mesh;
    mesh.file = data_dir + "/cornell_box.obj";
    mesh.materials = {{"light", m.light}, {"left", m.red},     {"right", m.green}, {"tall", m.white},
                      {"short", m.white}, {"back", m.white},   {"ceiling", m.white}, {"floor", m.white}};
    Project p;
    p.camera = cornell_camera();
    p.world.objects = {mesh};
    return p;
}

static Project scene_spheres() { // pyrite/test/spheres/spheres.lua:1-69
    const Expression green = spectrum_curve({{400, 0}, {450, 0.3f}, {500, 0}, {550, 1}, {600, 0}});
    const Expression red = spectrum_curve({{580, 0}, {600, 1}, {610, 1}, {650, 0}});
    Project p;
    p.camera.fov = 53;
    p.camera.transform = transform::look_at(vector(0, 1, 0), vector(0, 1, 1));
    p.renderer.spectrum_samples = 10, p.renderer.tile_size = 32, p.renderer.light_samples = 4;
    p.world.objects = {
        Sphere{vector(0, -50, 10), 50.0, Material(material::diffuse(1)), {}},
        Sphere{vector(0, 1.5, 10), 1.5, Material(material::emissive(light_source::d65() * 3)), {}},
        Sphere{vector(-3, 1.4, 10), 1.5, Material(mix(material::mirror(1), material::diffuse(green), fresnel(1.5))), {}},
        Sphere{vector(3, 1.4, 10), 1.5, Material(material::diffuse(red)), {}},
    };
    return p;
}

static Project scene_diamonds(const std::string& data_dir) { // pyrite/test/diamonds/diamonds.lua:1-60
    Mesh mesh;
    mesh.file = data_dir + "/diamonds.obj";
    mesh.materials = {
        {"diamonds", Material(material::refractive(1, 2.37782, Expression(0.01371)))},
        {"light_left", Material(material::emissive(light_source::d65()))},
        {"light_right", Material(material::emissive(light_source::d65() * 2))},
        {"bottom", Material(material::mirror(mix(Expression(0), Expression(0.2), fresnel(1.1))))},
    };
    Project p;
    p.renderer.spectrum_samples = 1, p.renderer.tile_size = 32, p.renderer.bounces = 32;
    p.camera.fov = 12.5;
    p.camera.focus_distance = 11.08, p.camera.aperture = 0.02;
    p.camera.transform = transform::look_at(vector(-6.55068, -8.55076, 4.0), vector(0.1, 0, 0.1), vector(0, 0, 1));
    p.world.objects = {mesh};
    return p;
}

static Project scene_lamps() { // scenes.py lamps_example: the lamp kinds and opcodes no other scene reaches
    Project p;
    p.renderer.light_samples = 2, p.renderer.bounces = 6, p.renderer.tile_size = 16;
    p.camera.fov = 45;
    p.camera.transform = transform::look_at(vector(0, -7, 2.5), vector(0, 0, 0.8), vector(0, 0, 1));
    p.world.sky = light_source::d65() * 0.2;
    p.world.objects = {
        Plane{vector(0, 0, 0), vector(0, 0, 1), Material(material::diffuse(0.5)), {}},
        Sphere{vector(-1.2, 0, 1), 1.0, Material(material::refractive(1, 1.5)), {}},
        Sphere{vector(1.2, 0.5, 0.7), 0.7, Material(material::diffuse(blackbody(3000) * 2e-13)), {}},
        Sphere{vector(0.2, -1.6, 0.5), 0.5, Material(material::diffuse(rgb(0.8, 0.3, 0.1))), {}},
        PointLight{vector(3, -3, 5), light_source::d65() * 40},
        DirectionalLight{vector(-0.3, 0.2, 0.933), 0.98, light_source::a() * 2},
    };
    return p;
}

// scenes.py textures_example: colour / mono textures and normal maps on all three shape kinds, a uv-mapped mesh held in memory
// with scale + transform, a textured lamp. The 16 x 16 texel arrays (already linear f32, what Texture::from_path leaves in
// memory) are read from <data_dir>/<name>.f32, written there by the test from the same generated images.
static std::vector<float> read_floats(const std::string& path, size_t count) {
    std::ifstream f(path, std::ios::binary);
    std::vector<float> v(count);
    if (!f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)(count * 4))) throw ProjectError("could not read " + path);
    return v;
}
static Project scene_textures(const std::string& dir) {
    const uint32_t n = 16;
    const Expression checker = color_texture(n, n, read_floats(dir + "/checker.f32", n * n * 4));
    const Expression nmap = color_texture(n, n, read_floats(dir + "/nmap_linear.f32", n * n * 4));
    const Expression rgba = color_texture(n, n, read_floats(dir + "/rgba.f32", n * n * 4));
    const Expression mono_linear = mono_texture(n, n, read_floats(dir + "/mono_linear.f32", n * n));
    const Expression mono_srgb = mono_texture(n, n, read_floats(dir + "/mono_srgb.f32", n * n));
    auto quad = std::make_shared<MeshData>();
    quad->position = {-1.5f, 1.0f, 0.2f, 0.5f, 1.0f, 0.2f, 0.5f, 2.6f, 1.4f, -1.5f, 2.6f, 1.4f};
    quad->texture = {0, 0, 2, 0, 2, 1.5f, 0, 1.5f};
    quad->normal = {0, -0.6f, 0.8f, 0.1f, -0.6f, 0.8f, 0, -0.55f, 0.83f, -0.1f, -0.6f, 0.8f};
    quad->objects = {MeshData::Object{"quad", {{{0, 0, 0}, {1, 1, 1}, {2, 2, 2}}, {{0, 0, 0}, {2, 2, 2}, {3, 3, 3}}}}};
    Mesh mesh;
    mesh.data = quad;
    mesh.transform = transform::look_at(vector(0.3, 0.2, 0), vector(0.3, 0.2, -1), vector(0.1, 1, 0));
    mesh.scale = 1.1;
    mesh.materials = {{"quad", Material(material::diffuse(checker), nmap)}};
    Project p;
    p.renderer.light_samples = 2, p.renderer.bounces = 5, p.renderer.tile_size = 16, p.renderer.spectrum_samples = 6;
    p.camera.fov = 50;
    p.camera.transform = transform::look_at(vector(0, -6, 2.6), vector(0, 0, 0.8), vector(0, 0, 1));
    p.world.sky = light_source::d65() * 0.15;
    p.world.objects = {
        Plane{vector(0, 0, 0), vector(0, 0, 1), Material(material::diffuse(checker * 0.9), nmap), vector(1.5, 2.5)},
        Sphere{vector(-1.6, -0.4, 0.9), 0.9, Material(material::diffuse(rgba), nmap), vector(0.25, 0.5)},
        Sphere{vector(1.5, 0.2, 0.7), 0.7, Material(mix(material::mirror(1), material::diffuse(rgb(0.9, 0.8, 0.3)), mono_linear)), {}},
        mesh,
        Sphere{vector(0.2, -1.4, 2.6), 0.35, Material(material::emissive(light_source::d65() * mono_srgb * 25)), vector(0.5, 0.5)},
        PointLight{vector(3, -3, 4), light_source::a() * 6},
    };
    return p;
}

static Project make_scene(const std::string& name, const std::string& data_dir) {
    if (name == "textures") return scene_textures(data_dir);
    if (name == "c1") return scene_c1(data_dir);
    if (name == "c2") return scene_c2(data_dir);
    if (name == "spheres") return scene_spheres();
    if (name == "diamonds") return scene_diamonds(data_dir);
    if (name == "lamps") return scene_lamps();
    throw ProjectError("unknown scene " + name);
}

// Texels for project files: <dir>/<file name>.<linear|srgb>.<mono|color>.f32 = u32 width, u32 height, f32 texels (already
// linear) -- written by whoever decodes the images (tests: pyrite_amd/images.py).
static TextureLoader texel_files(const std::string& dir) {
    return [dir](const std::string& path, bool linear, bool mono, uint32_t& width, uint32_t& height) {
        const size_t slash = path.find_last_of('/');
        const std::string file = dir + "/" + path.substr(slash == std::string::npos ? 0 : slash + 1) + (linear ? ".linear" : ".srgb") + (mono ? ".mono" : ".color") + ".f32";
        std::ifstream f(file, std::ios::binary);
        uint32_t wh[2] = {0, 0};
        if (!f.read(reinterpret_cast<char*>(wh), 8)) throw ProjectError("could not load " + path + " (no " + file + ")");
        width = wh[0], height = wh[1];
        std::vector<float> texels((size_t)width * height * (mono ? 1 : 4));
        if (!f.read(reinterpret_cast<char*>(texels.data()), (std::streamsize)(texels.size() * 4))) throw ProjectError("short texel file " + file);
        return texels;
    };
}

static void write_dump(const char* path, FlatScene& flat, const Project& project) {
    const PyrSceneDesc& d = flat.desc();
    std::vector<uint8_t> bytes(pyrh_serialize_desc(&d, nullptr, 0));
    pyrh_serialize_desc(&d, bytes.data(), bytes.size());
    const Camera cam = Camera::from_project(project.camera);
    const Renderer r = Renderer::from_project(project.renderer);
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char*>(bytes.data()), (std::streamsize)bytes.size());
    f.write(reinterpret_cast<const char*>(&cam.c), sizeof(cam.c));
    const uint32_t params[8] = {r.bounces, r.pixel_samples, r.light_samples, r.spectrum_samples, r.spectrum_bins, r.tile_size, project.image.width, project.image.height};
    f.write(reinterpret_cast<const char*>(params), sizeof(params));
    std::printf("%zu scene bytes, %zu triangles, %zu spheres, %zu planes\n", bytes.size(), flat.num_triangles(), flat.num_spheres(), flat.num_planes());
}

int main(int argc, char** argv) {
    try {
        if (argc >= 5 && std::string(argv[1]) == "dump-project") { // dump-project <project.lua> <texel dir | -> <out.bin>
            const LoadedProject loaded = load_project(argv[2], std::string(argv[3]) == "-" ? TextureLoader() : texel_files(argv[3]));
            FlatScene flat;
            flat.add_world(loaded.project.world, loaded.base_dir);
            write_dump(argv[4], flat, loaded.project);
            return 0;
        }
        if (argc >= 6 && std::string(argv[1]) == "render-project") { // render-project <project.lua> <texel dir | -> <seed> <out.png> [film.bin]
            const LoadedProject loaded = load_project(argv[2], std::string(argv[3]) == "-" ? TextureLoader() : texel_files(argv[3]));
            const Project& project = loaded.project;
            std::unique_ptr<World> world = World::from_project(project.world, loaded.base_dir);
            const Camera cam = Camera::from_project(project.camera);
            Renderer r = Renderer::from_project(project.renderer);
            r.seed = std::strtoull(argv[4], nullptr, 10);
            Film film = r.new_film(project.image.width, project.image.height);
            std::printf("The scene contains %zu objects.\n", world->num_objects()); // world.rs:251-254
            r.render(film, cam, *world);
            std::printf("Saving final result...\n"); // main.rs:313
            save_png(argv[5], film.develop(project.image.filter, project.image.white), film.width, film.height);
            if (argc >= 7) {
                std::ofstream f(argv[6], std::ios::binary);
                f.write(reinterpret_cast<const char*>(film.grains.data()), (std::streamsize)(film.grains.size() * sizeof(PyrGrain)));
            }
            return 0;
        }
        if (argc >= 5 && std::string(argv[1]) == "dump") {
            Project project = make_scene(argv[2], argv[3]);
            FlatScene flat;
            flat.add_world(project.world, argv[3]);
            std::printf("%s: ", argv[2]);
            write_dump(argv[4], flat, project);
            return 0;
        }
        if (argc >= 6 && std::string(argv[1]) == "intersect") { // intersect <scene> <data_dir> <rays.f32> <hits.bin>
            Project project = make_scene(argv[2], argv[3]);
            std::unique_ptr<World> world = World::from_project(project.world, argv[3]);
            std::ifstream in(argv[4], std::ios::binary);
            std::vector<char> bytes((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
            std::vector<float> rays(bytes.size() / 4);
            std::memcpy(rays.data(), bytes.data(), rays.size() * 4);
            const std::vector<PyrHit> hits = world->intersect(rays);
            std::ofstream out(argv[5], std::ios::binary);
            out.write(reinterpret_cast<const char*>(hits.data()), (std::streamsize)(hits.size() * sizeof(PyrHit)));
            std::printf("%zu rays\n", hits.size());
            return 0;
        }
        if (argc >= 9 && std::string(argv[1]) == "render") {
            Project project = make_scene(argv[2], argv[3]);
            project.image.width = (uint32_t)std::atoi(argv[4]), project.image.height = (uint32_t)std::atoi(argv[5]);
            project.renderer.pixel_samples = (uint32_t)std::atoi(argv[6]);
            std::unique_ptr<World> world = World::from_project(project.world, argv[3]);
            const Camera cam = Camera::from_project(project.camera);
            Renderer r = Renderer::from_project(project.renderer);
            r.seed = std::strtoull(argv[7], nullptr, 10);
            Film film = r.new_film(project.image.width, project.image.height);
            std::printf("The scene contains %zu objects.\n", world->num_objects()); // world.rs:251-254
            int last = -1;
            auto report = [&](Progress p) {
                if (p.progress != last) std::printf("%s... %3d %%\n", p.message, (int)p.progress);
                last = p.progress;
            };
            if (const char* list = std::getenv("PYRITE_DEVICES")) { // e.g. 0,1,2,3 -- Renderer::render over several GPUs
                std::vector<int> devices;
                for (const char* c = list; *c;) {
                    devices.push_back((int)std::strtol(c, const_cast<char**>(&c), 10));
                    if (*c == ',') ++c;
                }
                r.render(film, cam, *world, devices, report);
            } else {
                r.render(film, cam, *world, report);
            }
            std::ofstream f(argv[8], std::ios::binary);
            f.write(reinterpret_cast<const char*>(film.grains.data()), (std::streamsize)(film.grains.size() * sizeof(PyrGrain)));
            std::printf("film weight %.0f\n", film.total_weight());
            if (argc >= 10) {
                const std::vector<uint8_t> rgb = film.develop(project.image.filter, project.image.white);
                save_png(argv[9], rgb, film.width, film.height);
                std::printf("wrote %s\n", argv[9]);
            }
            return 0;
        }
        std::fprintf(stderr, "usage: pyrite_host_tool dump <scene> <data_dir> <out.bin> | render <scene> <data_dir> <w> <h> <spp> <seed> <film.bin> [out.png]\n");
        return 2;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
