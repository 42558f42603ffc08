// device_scene.h -- layout of the frozen scene in HBM, shared by the host packer (api.cpp) and the kernels.
#pragma once
#include <cstdint>

#include "../../include/pyrite_gpu.h"

namespace pyr {

// One primitive in BVH leaf order, 48 bytes = three float4 (one dwordx4 load each):
//   triangle: a = (v1.xyz, shape), b = (edge1.xyz, 0), c = (edge2.xyz, 0)   [edges as shapes/mod.rs:44-45]
//   sphere:   a = (centre.xyz, shape), b = (radius, 0, 0, 0), c = 0
// `shape` = (PyrShapeKind << 30) | index, stored as float bits.
struct DevPrim {
    float a[4], b[4], c[4];
};

// Two triangles of one leaf of the WIDE tree, component by component, 80 bytes = five float4:
//   (v1.x[2], v1.y[2]) (v1.z[2], shape[2]) (e1.x[2], e1.y[2]) (e1.z[2], e2.x[2]) (e2.y[2], e2.z[2])
// so that a (triangle A, triangle B) pair of every component sits in an aligned register pair and both Moeller-Trumbore
// tests run in packed-f32 instructions. A leaf of n triangles owns ceil(n / 2) consecutive records; the odd one out is paired
// with an all-zero triangle (det = 0: never hit) whose shape is PYR_HIT_NONE. Only built when every primitive in the tree is
// a triangle; the wide tree's leaf codes then count in these records: -1 - (first_pair << 3 | triangles left).
struct DevPrimPair {
    float q[5][4];
};

// Shading record of a triangle, indexed by ORIGINAL triangle index, 48 bytes:
//   (n1.xyz, material), (n2.xyz, 0), (n3.xyz, 0).
struct DevTriShade {
    float n1[4], n2[4], n3[4];
};

// Texture-space record of a triangle (only uploaded for scenes that run the program interpreter), 80 bytes:
//   uv = (t1.xy, t2.xy), (t3.xy, 0, 0); frames = Vertex.normal.from_space of v1, v2, v3 as (s, x, y, z).
struct DevTriTex {
    float uv12[4], uv3[4], f1[4], f2[4], f3[4];
};

struct DevTexture {
    uint32_t channels; // 4 (LinSrgba) or 1 (LinLuma)
    uint32_t width, height;
    uint32_t reserved;
    unsigned long long offset; // floats into DevScene::texture_data
};

// Lamp with everything Lamp::sample (lamp.rs:23-82) touches pre-gathered.
struct DevLamp {
    uint32_t kind, shape_kind, shape_index, color_program;
    float v[3]; // direction / position / sphere centre
    float width; // directional: cos half angle; sphere: radius
    float p1[3], p2[3], p3[3]; // triangle lamp vertices
    float n1[3], n2[3], n3[3]; // triangle lamp vertex normals
    float area;                // Shape::surface_area
    uint32_t material;
    float t1[2], t2[2], t3[2]; // triangle lamp: vertex texture coordinates; sphere lamp: t1 = texture_scale
};

enum FastProgram : uint32_t {
    FAST_NONE = 0,         // run the interpreter
    FAST_SPECTRUM = 1,     // [SpectrumValue(wavelength)]
    FAST_SPECTRUM_MUL = 2, // [SpectrumValue(wavelength), NumberValue(c), Binary Mul]  == spectrum * c
    FAST_MUL_SPECTRUM = 3  // [NumberValue(c), SpectrumValue(wavelength), Binary Mul]  == c * spectrum
};

// How a colour program gets onto the spectral tape (kernels.hip "Spectral tape"; round 4: scenes WITH interpreter programs):
//   DIRECT     a constant or a fast shape: the record names it, the replay looks its value up per wavelength;
//   HIT_VALUE  no instruction depends on the wavelength (diamonds.lua's `mix(0, 0.2, fresnel(1.1))`): the interpreter runs it ONCE
//              per hit, where the path is shaded, and the value is folded into the record's factor;
//   HIT_RGB    everything but a closing RgbSpectrumValue(wavelength, rgb) is independent of the wavelength (a colour texture, an
//              rgb() expression): the interpreter evaluates the rgb register once per hit, the record carries its three components
//              and the replay forms rgb . RGB_basis(wavelength) per wavelength, the very sum execution_context.rs:140-152 forms;
//   LAMBDA     a function of the wavelength alone made of numbers only (`blackbody(4000) * 3`, a product or mix of spectra): like a
//              fast shape it gets a value slot -- the replay evaluates it once per item with a small number-only interpreter
//              (kernels.hip lambda_eval) -- and the record names it;
//   PRODUCT    the program's value is a chain of products ((l * h1) * h2) ... with ONE factor l that depends on the wavelength alone in
//              LAMBDA's sense and up to three factors that depend on the hit alone (a mono texture times a spectrum times a number: the
//              textured lamps of the fuzz scenes and of the generated textures scene, a tinted mask): api.cpp splits it into two programs
//              of its own instructions -- the hit side (HIT_VALUE) and the wavelength side (DIRECT or LAMBDA, with a value slot) -- the
//              interpreter runs the hit side once per hit, and records carry h1 to the slot's value, h2 ... to that product and the
//              contribution's factor to the result: the products the interpreter and `contribute` form, in their order;
//   NONE       needs an interpreter run per hit AND wavelength (a mix of spectra by a fresnel term): a
//              scene with such a COLOUR program keeps the online form of round 3 (Walker::contribute_pending).
enum TapeForm : uint32_t { TAPE_FORM_DIRECT = 0, TAPE_FORM_HIT_VALUE = 1, TAPE_FORM_HIT_RGB = 2, TAPE_FORM_NONE = 3, TAPE_FORM_LAMBDA = 4, TAPE_FORM_PRODUCT = 5 };

struct DevProgram {
    uint32_t kind; // PyrProgramKind
    float constant;
    uint32_t first_instr, num_instrs;
    uint32_t output_kind, output_reg;
    uint32_t fast;            // FastProgram
    uint32_t fast_spectrum;   // spectrum id of the fast forms
    float fast_scale;         // c of the fast forms
    uint32_t reads_wavelength; // some executed operand is Input(Wavelength): ProbabilityInput::wavelength_used
    uint32_t tape_form;        // TapeForm
    // HIT_RGB: the rgb register the closing RgbSpectrumValue reads. PRODUCT (one word, so that the record stays twelve: three more were 8 % of
    // the example scenes' throughput): bits 0-7 / 8-15 the two programs api.cpp made of this one's instructions, the hit side and the
    // wavelength side; bits 16-19 the number of hit-side factors (1-3), bits 20-23, 24-27, 28-31 their number registers, innermost product first
    uint32_t tape_rgb_reg;
};

struct DevScene {
    const float* nodes;  // Node64[], 16 floats each
    const float* wide_nodes; // Node128[], 32 floats each, or nullptr: the tree the resumable traversal walks on big scenes
    uint32_t wide_stack_depth; // stack entries that tree can need
    const float* pair_prims;      // DevPrimPair[], or nullptr
    const float* wide_pair_nodes; // the wide tree once more with leaf codes that count in DevPrimPair records (the stage-scheduled render walks this copy; the ray-batch kernels keep the one-primitive records: they are bound by bytes through L1, and a pair is 80 B where a lone triangle is 48)
    const float* prims;  // DevPrim[], 12 floats each
    const float* tri_shade; // DevTriShade[]
    const float* spheres;   // [n][4] centre, radius (original order)
    const uint32_t* sphere_material;
    const float* planes;    // [n][8]
    const uint32_t* plane_material;
    const DevLamp* lamps;
    const PyrMaterial* materials;
    const PyrComponent* components;
    const DevProgram* programs;
    const PyrInstr* instrs;
    const PyrSpectrum* spectra;
    const float* spectrum_data;
    const float* rgb_basis;
    uint32_t num_planes, num_lamps;
    uint32_t rgb_count;
    float rgb_min, rgb_max;
    uint32_t sky_program;
    uint32_t stack_depth; // LDS stack entries per lane = BVH max depth
    uint32_t needs_interpreter; // some program is neither a constant nor a fast shape
    uint32_t num_nodes, num_prims;
    uint32_t num_spectra, num_spectrum_floats;
    uint32_t num_programs;
    uint32_t num_materials, num_components;
    uint32_t lds_table_floats; // > 0: spectra, materials, components, programs and lamps are staged into LDS (this many floats in all)
    // texture space (interpreter builds only)
    const float* tri_tex;          // DevTriTex[] by original triangle index, or nullptr
    const float* sphere_tex_scale; // [n][2]
    const float* plane_frames;     // [n][4] quaternion (s, x, y, z)
    const DevTexture* textures;
    const float* texture_data;
    uint32_t uses_textures; // some program holds a texture opcode or some material a normal map
    // some emissive component's probability program reads the wavelength: a light sample of such a material is added for the hero
    // wavelength only (algorithm.rs:78), i.e. the spectral tape can hold hero-only records. No BASELINE scene has one.
    uint32_t hero_only_records;
    // needs_interpreter != 0 and every colour program (components, lamps, sky) has a tape form: the stage scheduler records a tape
    // for this scene too and the interpreter runs once per hit instead of once per hit and wavelength
    uint32_t hit_tape;
    uint32_t rgb_records; // some colour program is HIT_RGB: its contributions are four records (three coefficients + the factor)
    uint32_t product_records; // some colour program is PRODUCT: the scene runs the PRODUCT builds of the hit-tape kernels
    uint32_t micro_records; // some colour program is HIT_RGB or PRODUCT: the replay reads bits 12-14 of a record (kernels.hip TAPE_RGB_*)
    // Boxes a shadow ray may skip lie beyond limit * shadow_margin (+ 1e-3), limit = the blocking limit (a squared distance): 1.001
    // covers the ulps between a box's entry distance and a triangle's hit distance; scenes with spheres take 1.01 -- collision's
    // sphere routine loses every digit of l.l - tca^2 for a ray that passes at a thousand radii or more and then reports hits up to a
    // radius NEARER than the sphere's own box, which the reference, walking with closest = inf, still counts as blockers.
    float shadow_margin;
    uint32_t tape_value_rows; // LDS value rows of the eager replay (above): 8, or as many as the scene's spectrum-reading programs need, up to 16
};

// Value rows of the eager replay (kernels.hip replay_tapes): rows of BLOCK floats in LDS, one per spectrum-reading program -- its value at the
// replay item's wavelength, looked up once per item -- and one that holds 1.0 (row 7: what the records of constants and plain factors name).
// Eight rows serve every BASELINE scene (four programs); a scene with more such programs than the seven rows in front of the 1.0 gets more
// rows behind it, up to the sixteen a record's 4-bit slot field can name (DevScene::tape_value_rows): slot s lives in row s, or s + 1 from
// the seventh on; the three rows of the RGB basis stay together. (C3 with six more spectra, past the rows: 573 -> 508 Msamples/s looking
// values up record by record -- tools/bench_many_spectra.py.)
constexpr uint32_t kTapeValueRows = 8, kTapeOneRow = kTapeValueRows - 1, kTapeMaxValueRows = 16;
constexpr uint32_t tape_row(uint32_t slot) { return slot + (slot >= kTapeOneRow ? 1u : 0u); }
constexpr uint32_t tape_rgb_row(uint32_t n_spectral) { return n_spectral + 3u <= kTapeOneRow ? n_spectral : (n_spectral >= kTapeOneRow ? n_spectral + 1u : kTapeValueRows); }
constexpr uint32_t tape_rows_needed(uint32_t n_spectral, bool rgb) {
    return (rgb ? tape_rgb_row(n_spectral) + 3u : (n_spectral == 0u ? 0u : tape_row(n_spectral - 1u) + 1u)) > kTapeValueRows
               ? (rgb ? tape_rgb_row(n_spectral) + 3u : tape_row(n_spectral - 1u) + 1u)
               : kTapeValueRows;
}

constexpr uint32_t kMaxStackDepth = 64; // >= kMaxBvhDepth (bvh.h) and >= the wide tree's stack need (else the binary tree is walked)

// Everything one render launch needs besides the scene.
struct RenderLaunch {
    PyrCamera camera;
    PyrFilmDesc film;
    uint32_t bounces, light_samples, spectrum_samples, tile_size, pixel_samples;
    uint32_t tiles_x, tiles_y;
    // The launch renders the tiles tile_begin + k * tile_stride (k = 0 .. tile_count - 1) of the raster grid. A chunk is 64
    // consecutive iterations of one tile; every tile owns chunks_per_tile chunk numbers (the count a full tile_size^2 tile
    // needs: tiles cut by the image border leave their last ones empty), so chunk c belongs to the launch's tile
    // c / chunks_per_tile. [chunk_begin, chunk_end) is the range this launch renders (progress slices cut it).
    uint32_t tile_begin, tile_stride, tile_count;
    uint32_t chunks_per_tile;
    uint32_t chunk_begin, chunk_end;
    uint32_t film_layout; // PYR_FILM_ROWS: film_out holds pixel rows [film_row_begin, + film_row_count); PYR_FILM_TILE_BLOCKS: one ringed block per tile
    uint32_t film_row_begin, film_row_count;
    uint64_t seed;
    float grains_per_wavelength; // bins / wl_width (film.rs:38)
    PyrGrain* film_out;
    unsigned long long* counters; // 9 words (PyrCounters order) or nullptr
    uint32_t scheduler; // 0 = bounce-synchronous walk (render_kernel), 1 = stage-scheduled state machine (render_kernel_sm)
    uint32_t sm_phase_lanes, sm_trav_steps; // stage scheduler: lanes that make a phase run; traversal steps per turn
    uint32_t sm_expose_lanes;               // finished lanes that make the tape replay run (TAPE builds)
    uint32_t stack_lds; // traversal stack levels kept in LDS (set by launch_render; deeper levels spill to scratch in the sm kernel)
    // Spectral tape of the stage-scheduled kernel (kernels.hip "Spectral tape"): [tape_max_ops][tape_lanes] 8-byte records,
    // one column per lane of the persistent grid.
    unsigned long long* tape;
    uint32_t tape_lanes, tape_max_ops;
    uint32_t* tape_overflow; // device word, set when a path wanted to append more than tape_max_ops records
    uint32_t tape_programs_lds; // programs whose prepared form the kernel keeps in LDS for the replay (set by launch_render; 0 = none)
};

// Work feed of the persistent traversal kernels: kFeedSegments cursor words, kFeedCursorStride words apart (kernels.hip WorkFeed).
constexpr uint32_t kFeedSegments = 8, kFeedCursorStride = 64;
constexpr size_t kFeedBytes = (size_t)kFeedSegments * kFeedCursorStride * sizeof(uint32_t);

struct IntersectLaunch {
    const float* rays;
    PyrHit* hits;
    uint32_t n;
    unsigned long long* counters;
    uint32_t* next;  // device, kFeedBytes, zero at launch: the work-feed cursors of the batch
    uint32_t num_cus;
    uint32_t reserve; // rays a wave reserves per atomic (set by launch_intersect)
    uint32_t stack_lds; // traversal stack levels kept in LDS (set by launch_intersect)
};

struct DevelopLaunch {
    PyrFilmDesc film;
    const PyrGrain* grains; // device
    float step_size, xyz_scale;
    uint32_t sample_count;
    const float* filter;    // device or nullptr
    const float* white_div; // device or nullptr
    const float* white_mul;
    const float* xyz_table; // device
    uint32_t xyz_count;
    float xyz_min, xyz_max;
    uint8_t* rgb_out; // device
};
int launch_develop(const DevelopLaunch& launch, void* stream);

// Adds the PYR_FILM_TILE_BLOCKS buffer of the tiles tile_begin + k * tile_stride (k < tile_count) into a whole-image film.
struct AssembleLaunch {
    PyrFilmDesc film;
    uint32_t tile_size, tiles_x;
    uint32_t tile_begin, tile_stride, tile_count;
    const PyrGrain* blocks; // device
    PyrGrain* film_out;     // device, whole image
};
int launch_assemble(const AssembleLaunch& launch, void* stream);

// launchers (kernels.hip)
int launch_render(const DevScene& scene, const RenderLaunch& launch, bool with_counters, void* stream, int num_cus);
uint32_t tape_ops_bound(const DevScene& scene, const RenderLaunch& launch); // records per path the stage-scheduled kernel may append to its spectral tape
bool uses_hit_tape(const DevScene& scene, const RenderLaunch& launch); // an interpreter scene that records a tape in this launch
uint32_t tape_lanes_bound(int num_cus);              // lanes (tape columns) of the largest grid launch_render starts
int launch_intersect(const DevScene& scene, const IntersectLaunch& launch, bool with_counters, void* stream);
const char* kernels_last_error();
bool scene_is_lds_resident(const DevScene& scene);

} // namespace pyr
