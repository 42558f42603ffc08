/*
 * pyrite_gpu.h -- C ABI of the MI355X-native replacement for Pyrite's camera-to-light
 * ("simple") renderer hot path.
 *
 * The reference (Ogeon/pyrite) has no FFI; the seam this ABI drops in behind is
 *
 *     Renderer::render(&self, film, task_runner, on_status, camera, world, resources)
 *         pyrite/src/renderer/mod.rs:77-111   (match Algorithm::Simple => simple::render, :87-89)
 *
 * i.e. everything the reference does from `simple::render` (pyrite/src/renderer/simple.rs:17-56)
 * downwards. A Rust caller binds these symbols with an `extern "C"` block
 * and calls them from a new `Algorithm::Gpu` arm.
 *
 * Conventions
 *   - plain C, plain pointers and sizes; no C++ / torch / HIP types in any signature
 *     (`hip_stream` is an opaque `void*` holding a hipStream_t, NULL = default stream);
 *   - the caller owns every input array and the film buffer; the library copies what it needs in
 *     pyr_scene_create and owns the returned PyrScene;
 *   - every function returns PYR_OK (0) or a negative PyrStatus and never aborts the process
 *     (the reference panics with panic=abort: Cargo.toml:6,10); pyr_last_error() returns a
 *     thread-local message for the last failure;
 *   - all floating point data is IEEE binary32, matrices are column-major (cgmath::Matrix4);
 *   - progress callbacks are only ever invoked on the calling thread
 *     (FnMut, not Send: pyrite/src/renderer/mod.rs:181-183).
 */
#ifndef PYRITE_GPU_H
#define PYRITE_GPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PYR_ABI_VERSION 4

typedef enum PyrStatus {
    PYR_OK = 0,
    PYR_ERR_INVALID_ARGUMENT = -1,
    PYR_ERR_UNSUPPORTED = -2, /* texture / normal-map opcodes, ray-marched shapes: SURVEY.md section 8 "out" rows */
    PYR_ERR_DEVICE = -3,      /* a HIP call failed or no gfx950 device is present */
    PYR_ERR_OUT_OF_MEMORY = -4
} PyrStatus;

/* ---------------------------------------------------------------- film ------------------------ */

/* == GrainData {accumulator, weight}: pyrite/src/film.rs:165-169 (field order acc, weight). */
typedef struct PyrGrain {
    float acc;
    float weight;
} PyrGrain;

/* == Film {width, height, grains_per_pixel, wavelength_start, wavelength_width}: film.rs:9-18.
 * Grain index = (x + y*width)*bins + bin  (film.rs:56). */
typedef struct PyrFilmDesc {
    uint32_t width;
    uint32_t height;
    uint32_t bins;
    float wl_start; /* 380 by default: renderer/mod.rs:16 */
    float wl_width; /* 400 by default */
} PyrFilmDesc;

/* ---------------------------------------------------------------- renderer parameters --------- */

#define PYR_FLAG_COUNTERS 1u /* run the instrumented kernel build and fill PyrCounters */

/* == Renderer {bounces, pixel_samples, light_samples, spectrum_samples, tile_size}
 * (renderer/mod.rs:18-28; defaults :63-75: 8 / - / 4 / 10 / 32). `threads` has no meaning here.
 *
 * The reference seeds one entropy RNG per tile (simple.rs:26-28) and is not reproducible; here
 * every sample (tile, iteration) owns a xorshift128 stream seeded from (seed, tile, iteration), see
 * DESIGN.md "RNG". `seed` selects the run.
 *
 * Sharding: a call renders the raster-order tile range [tile_begin, tile_end) (tile index
 * ty*tiles_x + tx over the grid of make_tiles, renderer/algorithm.rs:152-188); tile_end == 0 means
 * "all tiles"; with tile_stride > 1 only the tiles tile_begin, tile_begin + tile_stride, ... below tile_end are
 * rendered (the multi-GPU plan deals tiles round-robin: rank r of n renders tile_begin = r, tile_stride = n).
 * Tiles are independent units in the reference too: each has its own RNG and writes its own pixels
 * (renderer/simple.rs:36-55).
 *
 * film_layout says what the film buffer handed to the call holds:
 *   PYR_FILM_ROWS         pixel rows [film_row_begin, film_row_begin + film_row_count) of the image in the film.rs:56 layout
 *                         (film_row_count == 0 means the whole image);
 *   PYR_FILM_TILE_BLOCKS  one block per rendered tile, in the order the tiles are rendered: block k belongs to tile
 *                         tile_begin + k*stride and holds (tile_size + 2)^2 pixels x bins grains -- the tile's
 *                         tile_size x tile_size pixel square with a one-pixel ring around it; image pixel (x, y) sits
 *                         at block-local (x - tile_x0 + 1, y - tile_y0 + 1), row-major, bins grains per pixel. The ring is
 *                         there because Film::expose recomputes the pixel from the view-plane position (film.rs:233-246) and
 *                         rounding can move a sample drawn on a tile edge into the neighbouring pixel (~1e-6 per sample).
 *                         pyr_film_blocks_assemble[_device] adds such blocks into a whole-image film.
 * Exposures that map outside the buffer are dropped exactly like exposures outside the image (film.rs:51-54,92). */
#define PYR_FILM_ROWS 0u
#define PYR_FILM_TILE_BLOCKS 1u
typedef struct PyrRenderParams {
    uint32_t bounces;
    uint32_t pixel_samples;
    uint32_t light_samples;
    uint32_t spectrum_samples;
    uint32_t tile_size;
    uint32_t flags;
    uint64_t seed;
    uint32_t tile_begin;
    uint32_t tile_end;
    uint32_t film_row_begin;
    uint32_t film_row_count;
    uint32_t tile_stride; /* 0 and 1 both mean every tile of the range */
    uint32_t film_layout; /* PYR_FILM_ROWS | PYR_FILM_TILE_BLOCKS */
} PyrRenderParams;

/* == Camera::Perspective {transform, view_plane, focus_distance, aperture}: cameras.rs:20-27.
 * cam_to_world = look_at(from,to,up)^-1 (project/mod.rs:257-266), column-major.
 * view_plane = cos(fov/2)/sin(fov/2), fov in degrees (cameras.rs:43-45). */
typedef struct PyrCamera {
    float cam_to_world[16];
    float view_plane;
    float focus_distance;
    float aperture;
} PyrCamera;

/* ---------------------------------------------------------------- programs (the expression VM) */

/* Flattened form of Instruction / InstructionType (program/instruction.rs:12-119). */
typedef enum PyrOp {
    PYR_OP_NUMBER = 0,        /* NumberValue     {number=x.constant, output}                 :20-23 */
    PYR_OP_VECTOR = 1,        /* VectorValue     {x,y,z,w, output}                           :24-30 */
    PYR_OP_RGB = 2,           /* RgbValue        {red=x, green=y, blue=z, output}            :31-36 */
    PYR_OP_SPECTRUM = 3,      /* SpectrumValue   {wavelength=x, spectrum=a, output}          :37-41 */
    PYR_OP_COLOR_TEXTURE = 4, /* ColorTextureValue {texture_coordinates=b (vector input), texture=a, output (rgb)} :42-46 */
    PYR_OP_MONO_TEXTURE = 5,  /* MonoTextureValue  {texture_coordinates=b (vector input), texture=a, output (number)} :47-51 */
    PYR_OP_RGB_SPECTRUM = 6,  /* RgbSpectrumValue{wavelength=x, source=a (rgb reg), output}  :52-56 */
    PYR_OP_FRESNEL = 7,       /* Fresnel {ior=x, env_ior=y, normal=a, incident=b (vector inputs), output} :57-63 */
    PYR_OP_BLACKBODY = 8,     /* Blackbody {wavelength=x, temperature=y, output}             :64-68 */
    PYR_OP_RGB_TO_VECTOR = 9, /* Convert::RgbToVector {source=a, output}                     :69-71,104-110 */
    PYR_OP_BINARY = 10,       /* Binary {value_type, operator, lhs=a, rhs=b, output}         :72-78 */
    PYR_OP_MIX = 11,          /* Mix {value_type, lhs=a, rhs=b, amount=x, output}            :79-85 */
    PYR_OP_CLAMP = 12         /* Clamp {value=x, min=y, max=z, output}                       :86-91 */
} PyrOp;

typedef enum PyrValueType { PYR_VT_NUMBER = 0, PYR_VT_VECTOR = 1, PYR_VT_RGB = 2 } PyrValueType; /* :112-118 */
typedef enum PyrBinaryOperator { PYR_BIN_ADD = 0, PYR_BIN_SUB = 1, PYR_BIN_MUL = 2, PYR_BIN_DIV = 3 } PyrBinaryOperator;

/* NumberValue<N> (instruction.rs:94-99). */
typedef enum PyrOperandKind { PYR_OPERAND_CONSTANT = 0, PYR_OPERAND_INPUT = 1, PYR_OPERAND_REGISTER = 2 } PyrOperandKind;
typedef enum PyrNumberInput { PYR_INPUT_WAVELENGTH = 0 } PyrNumberInput;                 /* program/mod.rs:117-120 */
typedef enum PyrVectorInput { PYR_INPUT_NORMAL = 0, PYR_INPUT_INCIDENT = 1, PYR_INPUT_TEXTURE = 2 } PyrVectorInput; /* :130-135 */

/* Inputs bitflags (program/mod.rs:150-159). */
#define PYR_DEP_WAVELENGTH 0x01u
#define PYR_DEP_NORMAL 0x10u
#define PYR_DEP_INCIDENT 0x20u
#define PYR_DEP_TEXTURE 0x40u

typedef struct PyrOperand {
    uint32_t kind; /* PyrOperandKind */
    uint32_t bits; /* CONSTANT: the f32 bit pattern; INPUT: PyrNumberInput; REGISTER: number register */
} PyrOperand;

typedef struct PyrInstr {
    uint32_t op;         /* PyrOp */
    uint32_t value_type; /* PyrValueType, BINARY / MIX only */
    uint32_t operator_;  /* PyrBinaryOperator, BINARY only */
    uint32_t deps;       /* Instruction::dependencies (instruction.rs:15) */
    uint32_t output;     /* register index; the register file follows from op / value_type */
    uint32_t a;
    uint32_t b;
    uint32_t reserved;
    PyrOperand x, y, z, w;
} PyrInstr; /* 64 bytes */

#define PYR_MAX_NUMBER_REGISTERS 16
#define PYR_MAX_VECTOR_REGISTERS 8
#define PYR_MAX_RGB_REGISTERS 8

/* ProgramType (program/mod.rs:61-73): Constant short-circuits, Instructions reads one output register. */
typedef enum PyrProgramKind { PYR_PROGRAM_CONSTANT = 0, PYR_PROGRAM_INSTRUCTIONS = 1 } PyrProgramKind;
typedef enum PyrProgramOutput { PYR_OUTPUT_NUMBER = 0, PYR_OUTPUT_VECTOR = 1 } PyrProgramOutput; /* mod.rs:103-106 */

typedef struct PyrProgram {
    uint32_t kind;        /* PyrProgramKind */
    float constant;       /* value of a Constant program */
    uint32_t first_instr; /* into PyrSceneDesc::instrs */
    uint32_t num_instrs;
    uint32_t output_kind; /* PyrProgramOutput */
    uint32_t output_reg;
    uint32_t num_numbers; /* register counts: program/mod.rs:67-71 */
    uint32_t num_vectors;
    uint32_t num_rgbs;
} PyrProgram;

/* Spectrum<f32> (project/spectra.rs:13-24). ARRAY: `count` samples evenly spaced over [min,max],
 * clamped to the end values outside (spectra.rs:32-55). CURVE: `count` (x,y) pairs (2*count floats),
 * zero at and outside the end points (math.rs:22-72). Data lives at spectrum_data[offset...]. */
typedef enum PyrSpectrumFormat { PYR_SPECTRUM_ARRAY = 0, PYR_SPECTRUM_CURVE = 1 } PyrSpectrumFormat;
typedef struct PyrSpectrum {
    uint32_t format;
    float min;
    float max;
    uint32_t offset;
    uint32_t count;
} PyrSpectrum;

/* ---------------------------------------------------------------- materials ------------------- */

/* SurfaceBsdfType (materials/mod.rs:336-342). */
typedef enum PyrBsdf { PYR_BSDF_EMISSIVE = 0, PYR_BSDF_DIFFUSE = 1, PYR_BSDF_MIRROR = 2, PYR_BSDF_REFRACTIVE = 3 } PyrBsdf;

/* MaterialComponent (materials/mod.rs:230-235) + refractive::Properties (refractive.rs:39-45). */
typedef struct PyrComponent {
    uint32_t bsdf;                /* PyrBsdf */
    uint32_t color_program;       /* SurfaceBsdf::color */
    int32_t probability_program;  /* -1 = None */
    float selection_compensation; /* = number of entries in the list this component belongs to (mod.rs:213-221) */
    float ior, env_ior, dispersion, env_dispersion;
} PyrComponent;

/* Material {surface{components, emissive}, normal_map} (materials/mod.rs:27-31, :83-87). The emissive
 * list holds its own copies (with their own selection_compensation) of the emissive components. */
typedef struct PyrMaterial {
    uint32_t first_component;
    uint32_t num_components;
    uint32_t first_emissive;
    uint32_t num_emissive;
    int32_t normal_map_program; /* -1 = None; else a program with vector output run on NormalInput (materials/mod.rs:68-80) */
} PyrMaterial;

/* Texture<LinSrgba> / Texture<LinLuma> (texture.rs:18-22): linearised texels, row-major from the top row of the image
 * (texture.rs:163: data[x + y * width]). Color textures hold (red, green, blue, alpha), mono textures one luma value. */
typedef enum PyrTextureFormat { PYR_TEXTURE_COLOR = 0, PYR_TEXTURE_MONO = 1 } PyrTextureFormat;
typedef struct PyrTexture {
    uint32_t format; /* PyrTextureFormat */
    uint32_t width, height;
    uint32_t reserved;
    uint64_t offset; /* first texel's float in PyrSceneDesc.texture_data */
} PyrTexture;

/* Lamp (lamp.rs:12-20). */
typedef enum PyrLampKind { PYR_LAMP_DIRECTIONAL = 0, PYR_LAMP_POINT = 1, PYR_LAMP_SHAPE = 2 } PyrLampKind;
typedef enum PyrShapeKind { PYR_SHAPE_SPHERE = 0, PYR_SHAPE_TRIANGLE = 1, PYR_SHAPE_PLANE = 2 } PyrShapeKind;
typedef struct PyrLamp {
    uint32_t kind;          /* PyrLampKind */
    uint32_t shape_kind;    /* SHAPE: PyrShapeKind (sphere or triangle) */
    uint32_t shape_index;   /* SHAPE: index into the sphere / triangle arrays */
    uint32_t color_program; /* DIRECTIONAL / POINT */
    float v[3];             /* DIRECTIONAL: direction (as given, not normalised); POINT: position */
    float width;            /* DIRECTIONAL: cosine of the half angle (lamp.rs:30-34, tracer.rs:452) */
} PyrLamp;

/* ---------------------------------------------------------------- the scene ------------------- */

/* World {sky, lights, planes, finite_objects} (world.rs:31-36) + Resources.spectra (program/mod.rs:144-148),
 * after World::from_project (world.rs:39-271) has applied mesh scale and transform. */
typedef struct PyrSceneDesc {
    /* Shape::Triangle (shapes/mod.rs:39-46): v1,v2,v3 positions, unit vertex normals, uvs. */
    uint32_t num_triangles;
    const float* tri_positions;   /* [num_triangles][3][3] */
    const float* tri_normals;     /* [num_triangles][3][3] */
    const float* tri_uvs;         /* [num_triangles][3][2] or NULL (all zero) */
    const uint32_t* tri_material; /* [num_triangles] */

    /* Shape::Sphere (shapes/mod.rs:33-38). */
    uint32_t num_spheres;
    const float* spheres;           /* [num_spheres][4] = centre xyz, radius */
    const float* sphere_tex_scale;  /* [num_spheres][2] or NULL (1,1) */
    const uint32_t* sphere_material;

    /* shapes::Plane (shapes/mod.rs:434-439): point on the plane, unit normal, texture scale. */
    uint32_t num_planes;
    const float* planes; /* [num_planes][8] = origin xyz, normal xyz, texture_scale xy */
    const uint32_t* plane_material;

    uint32_t num_lamps;
    const PyrLamp* lamps; /* order == World::lights, pick_lamp indexes it (world.rs:301-305) */

    uint32_t num_materials;
    const PyrMaterial* materials;
    uint32_t num_components;
    const PyrComponent* components;

    uint32_t num_programs;
    const PyrProgram* programs;
    uint32_t num_instrs;
    const PyrInstr* instrs;

    uint32_t num_spectra;
    const PyrSpectrum* spectra;
    uint32_t num_spectrum_floats;
    const float* spectrum_data;

    /* crate::rgb::response::RGB (build.rs:18-59): `rgb_basis_count` rows of (r,g,b), an ARRAY spectrum over
     * [rgb_basis_min, rgb_basis_max]. NULL unless a program holds PYR_OP_RGB_SPECTRUM. */
    const float* rgb_basis;
    uint32_t rgb_basis_count;
    float rgb_basis_min;
    float rgb_basis_max;

    uint32_t sky_program; /* World::sky */

    /* Resources.textures (project/textures.rs:13-16); PYR_OP_COLOR_TEXTURE / PYR_OP_MONO_TEXTURE index this one list. */
    uint32_t num_textures;
    const PyrTexture* textures;
    uint64_t num_texture_floats;
    const float* texture_data;

    /* Normal::from_space (shapes/mod.rs:531-535), the tangent-space rotation as a quaternion (s, x, y, z), after
     * make_triangle (world.rs:308-374) and Shape::transform (shapes/mod.rs:322-344). Needed for triangles whose material
     * has a normal map (NULL: identity) and for every plane (texture coordinates come from it, shapes/mod.rs:454-468;
     * NULL: derived from the plane normal as world.rs:88-100 does). Sphere frames are computed at the hit. */
    const float* tri_frames;   /* [num_triangles][3][4] or NULL */
    const float* plane_frames; /* [num_planes][4] or NULL */
} PyrSceneDesc;

typedef struct PyrScene PyrScene;

/* == Progress {progress: u8, message} (renderer/mod.rs:229-232). */
typedef void (*PyrProgressFn)(void* user, uint8_t percent, const char* message);

/* Work counters of one render (flags & PYR_FLAG_COUNTERS) -- the units SURVEY.md section 8(d) prices. */
typedef struct PyrCounters {
    uint64_t samples;         /* iterations of the simple.rs:78 loop */
    uint64_t extension_rays;  /* World::intersect calls from tracer.rs:222 */
    uint64_t shadow_rays;     /* World::intersect calls from tracer.rs:381 */
    uint64_t box_tests;       /* AABBs slab-tested (32 B each) */
    uint64_t triangle_tests;  /* Moeller-Trumbore tests (36 B each) */
    uint64_t sphere_tests;    /* 16 B each */
    uint64_t plane_tests;     /* 16 B each */
    uint64_t shaded_hits;     /* surface-data fetches (52 B each) */
    uint64_t exposures;       /* Film::expose calls that landed in the window (16 B each) */
} PyrCounters;

/* One closest-hit result of pyr_scene_intersect: == Intersection {distance, surface_point} (shapes/mod.rs:472-482). */
#define PYR_HIT_NONE 0xFFFFFFFFu
typedef struct PyrHit {
    float distance;
    uint32_t shape; /* PYR_HIT_NONE, or (PyrShapeKind << 30) | index */
    float u, v;     /* triangle barycentrics (ShapeSurfacePoint::Triangle {u, v}), else 0 */
} PyrHit;

/* ---------------------------------------------------------------- entry points ---------------- */

int pyr_abi_version(void);

/* Number of gfx950 devices visible to the process (0 if none; never fails). */
int pyr_device_count(void);

/* Thread-local description of the last error returned on this thread ("" if none). */
const char* pyr_last_error(void);

/* Replaces the part of World::from_project that builds the acceleration structure
 * (Bvh::new, world.rs:262 / spatial/bvh.rs:13-155) and freezes the scene: copies the description, builds the
 * BVH on the host, uploads everything to `device`.
 *   Sizes: fewer than 2^28 triangles + spheres; the kernels address a node by a 32-bit byte offset, so an acceleration
 * structure of 4 GB (2^26 binary nodes, about 200 M triangles) or more is refused with PYR_ERR_UNSUPPORTED rather than
 * wrapped around. A render call takes fewer than 2^32 pixels (and, as tile blocks, fewer than 2^32 block pixels), at most
 * 64 wavelengths per sample and fewer than 2^32 chunks of 64 samples; larger calls are refused the same way. */
int pyr_scene_create(const PyrSceneDesc* desc, int device, PyrScene** out_scene);
void pyr_scene_destroy(PyrScene* scene);

/* Replaces Renderer::render / simple::render (renderer/mod.rs:77-111, renderer/simple.rs:17-56) for
 * Algorithm::Simple. Blocking. Adds the exposures of this call into `film_inout`, a HOST buffer of
 * (film_row_count or height) * width * bins grains in the film.rs:56 layout. `on_status` may be NULL. */
int pyr_render_simple(PyrScene* scene, const PyrCamera* camera, const PyrFilmDesc* film, const PyrRenderParams* params,
                      PyrGrain* film_inout, PyrProgressFn on_status, void* user);

/* Same, but `film_device` is DEVICE memory on the scene's device and the work is enqueued on `hip_stream`
 * (a hipStream_t, NULL = default stream) without synchronising: the caller synchronises the stream.
 * A PyrScene owns device-side working memory (counters, the spectral tape) that serves one render at a
 * time: renders of ONE scene must be issued on one stream (or otherwise ordered); different scenes are independent. */
int pyr_render_simple_device(PyrScene* scene, const PyrCamera* camera, const PyrFilmDesc* film,
                             const PyrRenderParams* params, PyrGrain* film_device, void* hip_stream);

/* Counters of the last render on this scene that ran with PYR_FLAG_COUNTERS (synchronises the device). */
int pyr_scene_counters(PyrScene* scene, PyrCounters* out);

/* World::intersect (world.rs:273-299) for a batch of rays: rays = [n][6] (origin xyz, direction xyz), HOST memory;
 * hits = [n], HOST memory. `elapsed_ms`, if not NULL, receives the kernel's duration measured with HIP events;
 * `counters`, if not NULL, receives box/triangle/sphere/plane test counts for the batch (instrumented build). */
int pyr_scene_intersect(PyrScene* scene, const float* rays, uint32_t n, PyrHit* hits, float* elapsed_ms,
                        PyrCounters* counters);

/* Same for rays and hits already resident on the device, enqueued on `hip_stream` without synchronising. */
int pyr_scene_intersect_device(PyrScene* scene, const float* rays_device, uint32_t n, PyrHit* hits_device,
                               void* hip_stream);

/* Introspection of the acceleration structure the library built (node count, bytes, depth). */
typedef struct PyrBvhInfo {
    uint32_t num_nodes;   /* 64-byte two-child nodes */
    uint32_t num_leaves;
    uint32_t max_depth;
    uint32_t num_primitives;
    uint64_t node_bytes;
    uint64_t primitive_bytes;
    /* ABI 4: the tree the resumable traversal walks on scenes that do not live in LDS (0 when the scene has none): 128-byte
     * four-child nodes; for triangle-only scenes a second copy of them whose leaves index 80-byte records of two triangles */
    uint32_t num_wide_nodes;
    uint32_t num_pair_records;
    uint64_t wide_node_bytes;
    uint64_t pair_record_bytes;
} PyrBvhInfo;
int pyr_scene_bvh_info(PyrScene* scene, PyrBvhInfo* out);

/* Introspection of the kernel a render of `scene` with `params` would run (nothing is launched; only spectrum_samples is read
 * today). Results never depend on it -- every schedule is the same per-sample arithmetic as tracer.rs:208-345 -- but throughput
 * does, and a maintainer wants to see why a scene is slow: */
typedef struct PyrPathInfo {
    uint32_t stage_scheduler; /* 0: the bounce-synchronous walk (scenes that live in LDS and run no interpreter programs); 1: the
                                 stage-scheduled state machine */
    uint32_t interpreter;     /* 1: some program of the scene is not a constant / spectrum / one of the fast shapes: the kernel
                                 carries the program interpreter (and texture / normal-map sampling) */
    uint32_t scene_in_lds;    /* 1: nodes and primitives are staged in LDS */
    uint32_t tape;            /* 0: every wavelength's throughput is kept online; 1: spectral tape (no interpreter programs);
                                 2: hit tape -- interpreter programs run once per hit, the per-wavelength part is replayed at
                                 full width; needs spectrum_samples >= 4 and a tape form for every colour program (a product
                                 of a mono texture and a spectrum has none) */
    uint32_t phase_lanes;     /* lanes of a wave that must want a phase before it runs (stage scheduler; 0 otherwise) */
    uint32_t reserved[3];
} PyrPathInfo;
int pyr_scene_path_info(PyrScene* scene, const PyrRenderParams* params, PyrPathInfo* out);

/* ---------------------------------------------------------------- multi-GPU -----------
 * The reference is one process with shared memory; its unit of parallel work is the tile (renderer/simple.rs:36-55: every
 * tile has its own RNG and exposes its own pixels, renderer/mod.rs:125-189 hands tiles to worker threads). Here the scene
 * is replicated on every GPU, the tiles of the image are dealt round-robin to the ranks (rank r of n renders tiles
 * r, r + n, ...: every rank sees every part of the image, which balances the cost without measuring it), every rank
 * renders its tiles in ONE launch into a private PYR_FILM_TILE_BLOCKS buffer with no data-path collective, and ONE gather
 * -- a group of ncclSend / ncclRecv over xGMI (RCCL) -- brings the blocks to rank 0, which adds them into the film.
 * With the per-(tile, iteration) RNG the n-GPU film equals the 1-GPU film up to the order of the float additions. */

/* Grains a PYR_FILM_TILE_BLOCKS buffer needs for the tiles `params` selects (tile_begin / tile_end / tile_stride /
 * tile_size); 0 when the arguments are invalid. */
uint64_t pyr_film_blocks_grains(const PyrFilmDesc* film, const PyrRenderParams* params);

/* Adds the blocks a render with these `params` (film_layout = PYR_FILM_TILE_BLOCKS) produced into `film_device`, a
 * whole-image film in the film.rs:56 layout; both buffers on `device`, enqueued on `hip_stream`. Ring pixels outside the
 * image do not exist and are skipped (nothing was exposed there). */
int pyr_film_blocks_assemble_device(const PyrFilmDesc* film, const PyrRenderParams* params, const PyrGrain* blocks_device,
                                    PyrGrain* film_device, int device, void* hip_stream);

/* One process per GPU (torch.distributed, MPI, ...): a communicator over RCCL. Rank 0 obtains an id with
 * pyr_comm_unique_id (ncclGetUniqueId; 128 bytes), the host brings it to the other ranks by whatever means it has, and
 * every rank calls pyr_comm_create (ncclCommInitRank) with its device. librccl is loaded when the first of these is
 * called; single-GPU use of the library never touches it. */
#define PYR_COMM_ID_BYTES 128
typedef struct PyrComm PyrComm;
int pyr_comm_unique_id(uint8_t id_out[PYR_COMM_ID_BYTES]);
/* A world of one rank needs no RCCL communicator and gets none -- unless the environment says PYRITE_FORCE_RCCL=1: then
 * ncclCommInitRank(nranks = 1) really runs and the rank's blocks travel through a grouped self ncclSend / ncclRecv, which
 * exercises the gather's code on a single GPU. pyr_comm_uses_rccl tells which kind a communicator is (1 / 0). */
int pyr_comm_create(const uint8_t id[PYR_COMM_ID_BYTES], int rank, int num_ranks, int device, PyrComm** out_comm);
int pyr_comm_uses_rccl(const PyrComm* comm);
void pyr_comm_destroy(PyrComm* comm);

/* This rank's part of a sharded render, enqueued on `hip_stream`: of the tiles `params` selects (normally all:
 * tile_begin = tile_end = 0, tile_stride <= 1) rank r renders every num_ranks-th starting at the r-th, sends its blocks
 * to rank 0 (grouped ncclSend / ncclRecv: the one gather), and rank 0 adds everybody's blocks into `film_device_rank0` (a
 * whole-image film on rank 0's device; ignored on the other ranks, may be NULL there). `scene` must live on the
 * communicator's device. Working buffers are kept on the communicator between calls.
 *   Failures up to the gather never leave a peer blocked (the reference's workers report to one collecting thread,
 * renderer/mod.rs:181-183): before anything is sent the ranks agree -- one one-word ncclAllReduce, waited for on the host -- that every rank got
 * through its argument checks and buffer growth; if one did not, EVERY rank returns an error and nothing is rendered. What
 * fails later (a launch, or the kernels flagging their own film invalid) travels in a trailer grain behind each rank's
 * blocks, so every rank still enters the gather; pyr_comm_status() reports it on rank 0 (every rank's trailer) and on the
 * sender (its own) once `hip_stream` has been waited for: PYR_OK, or PYR_ERR_DEVICE naming the rank -- the film is invalid
 * then. An error inside the collective calls themselves aborts THIS rank's communicator (ncclCommAbort); every later call on
 * it fails. pyr_render_simple_multi then aborts the sibling ranks' communicators too, at once, so none of its threads stays
 * in the gather; ranks in OTHER processes cannot be reached from here -- their wait on `hip_stream` ends when RCCL notices
 * the lost peer, so a multi-process caller should bound that wait.
 *   Coordinates: the kernels' normalize / square root are the correctly rounded IEEE results (the reference's) for lengths whose
 * squares are normal f32 numbers; pyr_scene_create refuses (PYR_ERR_UNSUPPORTED) a scene with a primitive beyond 1e15 units
 * from the origin, and features below ~1e-15 units are outside the verified range (the reference itself ignores anything
 * under DIST_EPSILON = 1e-4, math.rs:4). */
int pyr_comm_status(PyrComm* comm);
int pyr_render_simple_sharded(PyrComm* comm, PyrScene* scene, const PyrCamera* camera, const PyrFilmDesc* film,
                              const PyrRenderParams* params, PyrGrain* film_device_rank0, void* hip_stream);

/* One process driving several GPUs (what a Rust host does with its thread pool): `scenes[i]` is the scene created on
 * the i-th device to use. Blocking; one host thread per device; the same plan and the same gather as above
 * (ncclCommInitAll). Adds the exposures into `film_inout`, a HOST film of the whole image. If the same device appears
 * more than once (a test rig: several logical ranks on one GPU, which RCCL refuses) the blocks travel by
 * hipMemcpyPeerAsync instead. `on_status` (may be NULL) is called on the calling thread only. */
int pyr_render_simple_multi(PyrScene* const* scenes, uint32_t num_devices, const PyrCamera* camera, const PyrFilmDesc* film,
                            const PyrRenderParams* params, PyrGrain* film_inout, PyrProgressFn on_status, void* user);

/* ---------------------------------------------------------------- film development ("next" row f1) -------------
 * The step after the hot path: main.rs:315-327 turns every developed pixel spectrum into an 8-bit sRGB pixel through
 * spectrum_to_xyz (main.rs:352-369: trapezoid rule over the film's wavelength span in `step_size` steps against the
 * CIE 1931 observer tables, divided by the span, times 3.444) and palette's Xyz -> linear sRGB -> sRGB encoding.
 * The optional `filter` and `white` programs of the project's image settings (main.rs:197-238) act on the sampled
 * intensity as  ((intensity * filter[i]) / white_div[i]) * white_mul[i]  at the i-th sampling wavelength
 * wl_i = wl_start + i*step_size (i = 0 .. sample_count-1); the host evaluates those programs once per wavelength and
 * passes the three arrays (NULL = stage absent).
 * As in the reference, the LAST pixel of the film is never developed (film.rs:299 `end < len`) and stays black. */
typedef struct PyrDevelopParams {
    float step_size;         /* 2.0 for the final image, 30.0 for previews (main.rs:270, :311) */
    float xyz_scale;         /* 3.444 (main.rs:368) */
    uint32_t sample_count;   /* number of sampling wavelengths = trapezoid steps + 1 */
    const float* filter;     /* [sample_count] or NULL */
    const float* white_div;  /* [sample_count] or NULL: max(white(wl)/white_max, 1e-6) */
    const float* white_mul;  /* [sample_count] or NULL: D65(wl)/D65_max */
    const float* xyz_table;  /* [xyz_count][3]: crate::xyz::response::{X,Y,Z} (build.rs:68-121), ARRAY spectra over [xyz_min, xyz_max] */
    uint32_t xyz_count;
    float xyz_min, xyz_max;
} PyrDevelopParams;

/* film: HOST grains of the whole image (height*width*bins); rgb_out: HOST, height*width*3 bytes, row-major RGB. Blocking. */
int pyr_film_develop(const PyrFilmDesc* film, const PyrGrain* grains, const PyrDevelopParams* params, uint8_t* rgb_out, int device);
/* Same with the film and the output resident on `device`; the PyrDevelopParams arrays stay HOST pointers (copied by the
 * call); enqueued on `hip_stream`. */
int pyr_film_develop_device(const PyrFilmDesc* film, const PyrGrain* grains_device, const PyrDevelopParams* params, uint8_t* rgb_device,
                            int device, void* hip_stream);

#ifdef __cplusplus
}
#endif

#endif /* PYRITE_GPU_H */
