/*
 * oracle.h -- C entry points of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY. This library is a scalar CPU restatement of the reference's
 * camera-to-light renderer (Ogeon/pyrite, pyrite/src/renderer/simple.rs and everything below it).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product
 * (pyrite_amd/ and libpyrite_gpu.so) never does.
 *
 * PARITY UNPINNED at the numeric level: the reference has no tests, golden vectors or fixtures for this path
 * (SURVEY.md section 4 / 8(c)), cannot be built here (Rust, no toolchain, no network), and seeds its RNG
 * from OS entropy, so no numeric output of the reference exists to pin this restatement against. It is written
 * line by line from the cited sources; third-party arithmetic (collision, cgmath, rand, palette) is
 * restated from the published algorithms of the pinned versions in Cargo.lock and labelled as such.
 * WEAK PIN: the three example images the reference rendered with this renderer (pyrite/test/{spheres,diamonds,textures}/
 * hq_example.png, reduced to 8 x 8 block means in tests/golden/reference_example_images.npz) are reproduced:
 * textures -- correlation 0.997, median luminance ratio 0.99, colour-checker patches within a few percent per channel;
 * diamonds -- 0.96x mean luminance, correlation 0.999 at the project's own 200 spp x 256 bounces; spheres (an older
 * image) -- floor luminance 0.90x. tests/test_reference_images.py.
 *
 * It consumes the same plain-data scene description as the product (include/pyrite_gpu.h) -- the data
 * format is shared, no code is.
 */
#ifndef PYRITE_ORACLE_H
#define PYRITE_ORACLE_H

#include "../include/pyrite_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OracleScene OracleScene;

const char* oracle_last_error(void);

/* World::from_project's Bvh::new (spatial/bvh.rs:13-155) + scene freeze. */
int oracle_scene_create(const PyrSceneDesc* desc, OracleScene** out);
void oracle_scene_destroy(OracleScene* scene);

/* simple::render (renderer/simple.rs:17-56) with `threads` workers pulling tiles in make_tiles order
 * (renderer/mod.rs:125-189). Adds into `film_inout` (window semantics of PyrRenderParams). `counters` may be NULL. */
int oracle_render_simple(OracleScene* scene, const PyrCamera* camera, const PyrFilmDesc* film,
                         const PyrRenderParams* params, PyrGrain* film_inout, int threads, PyrCounters* counters);

/* World::intersect (world.rs:273-299) per ray; rays = [n][6]. `counters` may be NULL. */
int oracle_intersect(OracleScene* scene, const float* rays, uint32_t n, PyrHit* hits, PyrCounters* counters);

/* Flattened reference BVH for inspection: node i = {min xyz, max xyz, subtree_size (0 = leaf), item}. */
uint32_t oracle_bvh_num_nodes(OracleScene* scene);
int oracle_bvh_node(OracleScene* scene, uint32_t index, float* aabb6, uint32_t* subtree_size, uint32_t* item);

/* ---- per-function known-answer entry points (each restates the cited lines) ---- */

/* RNG: xorshift128 (rand_xorshift 0.3.0) seeded per (seed, tile, iteration); rand 0.8.5 distributions. */
void oracle_rng_seed(uint64_t seed, uint32_t tile, uint64_t iteration, uint32_t state[4]);
uint32_t oracle_rng_next_u32(uint32_t state[4]);
float oracle_rng_gen_f32(uint32_t state[4]);
float oracle_rng_gen_range_f32(uint32_t state[4], float low, float high);
uint32_t oracle_rng_gen_range_usize(uint32_t state[4], uint32_t n);
uint32_t oracle_rng_choose_index(uint32_t state[4], uint32_t n);

/* math.rs */
int oracle_aabb_intersection_distance(const float aabb6[6], const float ray6[6], float* distance); /* :184-207 */
float oracle_schlick(float n1, float n2, const float normal[3], const float incident[3]);          /* :75-96 */
float oracle_fresnel(float ior, float env_ior, const float normal[3], const float incident[3]);    /* :167-175 */
void oracle_ortho(const float v[3], float out[3]);                                                 /* :98-114 */
void oracle_sample_sphere(uint32_t state[4], float out[3]);                                        /* :147-153 */
void oracle_sample_hemisphere(uint32_t state[4], const float dir[3], float out[3]);                /* :155-164 */
void oracle_sample_cone(uint32_t state[4], const float dir[3], float cos_half, float out[3]);      /* :125-137 */
float oracle_solid_angle(float cos_half);                                                          /* :139-145 */
float oracle_sin32(float x); /* the f32 sin / cos / acos kernels both sides use instead of libm (oracle.cpp "Transcendentals") */
float oracle_cos32(float x);
float oracle_acos32(float x);
float oracle_blackbody(float wavelength, float temperature);                                       /* :177-182 */

/* shapes/mod.rs */
int oracle_triangle_intersect(const float v1[3], const float v2[3], const float v3[3], const float ray6[6],
                              float* dist, float* u, float* v);                                    /* :75-119 */
int oracle_sphere_intersect(const float centre[3], float radius, const float ray6[6], float* dist, float point[3]); /* :57-74 */

/* project/spectra.rs:30-58 + math.rs:22-72 */
float oracle_spectrum_get(uint32_t format, float min, float max, const float* data, uint32_t count, float wavelength);

/* materials/refractive.rs:47-91; returns the branch probability weight, writes out_dir. */
float oracle_refract(uint32_t state[4], float ior, float env_ior, const float in_dir[3], const float normal[3], float out_dir[3]);

/* film.rs:68-83 (+ simple.rs:105-107 hero pick): writes S wavelengths after swap_remove, hero first. */
void oracle_sample_wavelengths(uint32_t state[4], float start, float width, uint32_t s, float* hero_then_companions);
/* film.rs:85-87 and :233-246; returns 0 if rejected. */
uint32_t oracle_wavelength_to_grain(float wavelength, float start, float width, uint32_t bins);
int oracle_to_pixel(uint32_t width, uint32_t height, float x, float y, uint32_t* px, uint32_t* py);
/* cameras.rs:57-68: view-plane rectangle of a pixel rectangle: out = from.x, from.y, size.x, size.y */
void oracle_to_view_area(uint32_t x, uint32_t y, uint32_t w, uint32_t h, uint32_t image_w, uint32_t image_h, float out[4]);
/* cameras.rs:70-97 */
void oracle_ray_towards(const PyrCamera* camera, uint32_t state[4], float x, float y, float ray6[6]);
/* make_tiles order (renderer/algorithm.rs:152-188): writes raster tile indices in render order; returns count. */
uint32_t oracle_tile_order(uint32_t width, uint32_t height, uint32_t tile_size, uint32_t* order, uint32_t capacity);

/* program/execution_context.rs:29-56: run program `program` of the scene with the given inputs. */
float oracle_run_program(OracleScene* scene, uint32_t program, float wavelength, const float normal[3],
                         const float incident[3], const float texture[2], int* wavelength_used);

/* texture.rs:87-150 + :297-334: bicubic lookup with wrap-around in a [height][width][channels] texture ("next" row f4). */
void oracle_texture_get(uint32_t channels, uint32_t width, uint32_t height, const float* texels, float x, float y, float* out);
/* [3P] cgmath: Quaternion::from(Matrix3::from_cols(c0, c1, c2)) as (s, x, y, z); Quaternion * Vector3. */
void oracle_quat_from_cols(const float c0[3], const float c1[3], const float c2[3], float out[4]);
void oracle_quat_rotate(const float q[4], const float v[3], float out[3]);
/* SurfacePoint::get_surface_data (shapes/mod.rs:484-494) + Material::apply_normal_map (materials/mod.rs:68-80) at the first
 * hit of a ray; returns 0 on a miss. */
int oracle_surface_data(OracleScene* scene, const float ray6[6], float wavelength, float normal[3], float texture[2], float frame[4],
                        float shading_normal[3]);

/* main.rs:315-327 + :352-418 + film.rs:282-337: develop a whole film into 8-bit sRGB ("next" row f1). */
int oracle_film_develop(const PyrFilmDesc* film, const PyrGrain* grains, const PyrDevelopParams* params, uint8_t* rgb_out);

#ifdef __cplusplus
}
#endif
#endif
