// Test driver (tests/test_host_cpp.py::test_host_cpp_and_oracle_under_sanitizers): the host-side C++ that parses untrusted input
// -- the project-file reader (lua_project.cpp), the OBJ / image ingest and the program compiler (pyrite_host.cpp, images.cpp,
// jpeg.c), the tree builder (bvh.cpp) -- and the CPU oracle (oracle.cpp), all compiled with -fsanitize=address,undefined into
// ONE executable, with no GPU involved:
//     host_asan <file> ...
// Every file is read as a project: it must either load, flatten (FlatScene), build its trees (build_bvh + collapse_to_wide on
// the flattened primitives) and render 16 x 8 x 2 spp through the oracle -- or be refused with an exception. A crash, an
// out-of-bounds access, an overflow or a leak ends the process with the sanitizer's report. Prints "<loaded> loaded, <refused> refused".
#include <cmath>
#include <cstdio>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

#include "../include/pyrite_host.hpp"
#include "../oracle/oracle.h"
#include "../pyrite_amd/csrc/bvh.h"

using namespace pyrite;

static void build_trees(const PyrSceneDesc& d) {
    std::vector<pyr::PrimBounds> bounds;
    for (uint32_t i = 0; i < d.num_spheres; ++i) {
        const float* p = d.spheres + 4 * (size_t)i;
        pyr::PrimBounds b;
        for (int a = 0; a < 3; ++a) b.lo[a] = p[a] - p[3], b.hi[a] = p[a] + p[3];
        b.shape = i;
        bounds.push_back(b);
    }
    for (uint32_t i = 0; i < d.num_triangles; ++i) {
        const float* p = d.tri_positions + 9 * (size_t)i;
        pyr::PrimBounds b;
        for (int a = 0; a < 3; ++a) b.lo[a] = std::fmin(p[a], std::fmin(p[3 + a], p[6 + a])), b.hi[a] = std::fmax(p[a], std::fmax(p[3 + a], p[6 + a]));
        b.shape = (1u << 30) | i;
        bounds.push_back(b);
    }
    for (int pairs = 0; pairs < 2; ++pairs) {
        const pyr::BuiltBvh bvh = pyr::build_bvh(bounds, pairs != 0);
        const pyr::WideBvh wide = pyr::collapse_to_wide(bvh);
        if (bvh.prim_order.size() != bounds.size() || wide.nodes.empty()) std::abort();
    }
    // hostile bounds the flattener never makes: NaN, infinite and inverted boxes, all primitives in one point
    std::vector<pyr::PrimBounds> odd = bounds;
    for (size_t i = 0; i < odd.size(); ++i) {
        if (i % 5 == 0) odd[i].lo[0] = odd[i].hi[0] = NAN;
        if (i % 7 == 0) odd[i].hi[1] = INFINITY;
        if (i % 11 == 0) std::swap(odd[i].lo[2], odd[i].hi[2]);
    }
    (void)pyr::collapse_to_wide(pyr::build_bvh(odd, true));
    std::vector<pyr::PrimBounds> point(257);
    for (auto& b : point) std::memset(&b, 0, sizeof(b));
    (void)pyr::collapse_to_wide(pyr::build_bvh(point, false));
    (void)pyr::collapse_to_wide(pyr::build_bvh({}, false));
}

int main(int argc, char** argv) {
    unsigned loaded = 0, refused = 0;
    for (int i = 1; i < argc; ++i) {
        try {
            const LoadedProject project = load_project(argv[i]);
            FlatScene flat;
            flat.add_world(project.project.world, project.base_dir);
            const PyrSceneDesc& desc = flat.desc();
            build_trees(desc);
            OracleScene* scene = nullptr;
            if (oracle_scene_create(&desc, &scene) != 0) throw std::runtime_error(std::string("oracle: ") + oracle_last_error());
            const Camera cam = Camera::from_project(project.project.camera);
            Renderer r = Renderer::from_project(project.project.renderer);
            Film film = r.new_film(16, 8);
            PyrRenderParams params{};
            params.bounces = r.bounces < 4 ? r.bounces : 4, params.pixel_samples = 2, params.light_samples = r.light_samples, params.spectrum_samples = r.spectrum_samples;
            params.tile_size = 8, params.seed = 1;
            const PyrFilmDesc fd = film.desc();
            PyrCounters counters{};
            const int rc = oracle_render_simple(scene, &cam.c, &fd, &params, film.grains.data(), 2, &counters);
            oracle_scene_destroy(scene);
            if (rc != 0) throw std::runtime_error(std::string("oracle: ") + oracle_last_error());
            if (counters.samples != 16u * 8u * 2u) std::abort();
            ++loaded;
        } catch (const std::exception& e) {
            if (!e.what()[0]) return 3; // a refusal without a message
            std::fprintf(stderr, "refused %s: %.200s\n", argv[i], e.what());
            ++refused;
        }
    }
    std::printf("%u loaded, %u refused\n", loaded, refused);
    return 0;
}
