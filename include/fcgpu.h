/*
 * fcgpu.h -- C-ABI boundary of the MI355X-native Forward-MIS / Forward-BSDF path tracer.
 *
 * This is the drop-in boundary for ONE hot path of full-confession/ray-tracing: the per-sample
 * bounce loop of `fc::forward_mis_integrator::run_once`
 * (reference: src/integrators/forward_mis_integrator.hpp:13-210) and its degenerate subset
 * `fc::forward_bsdf_integrator::run_once` (src/integrators/forward_bsdf_integrator.hpp:15-99),
 * together with the renderer loop that calls it W*H*spp times
 * (src/renderer/renderer.hpp:154-179) and the film read-out (src/renderer/renderer.hpp:115-141).
 *
 * A per-sample virtual call cannot feed a GPU, so the boundary sits one level up: the host
 * describes the scene with plain-old-data arrays (the same parameters the reference's
 * `example_scenes.hpp` passes to its constructors), the library flattens it for the device,
 * renders whole sample passes, and hands back the film as `double rgb_sum[W*H*3]` + a sample
 * count, exactly what `renderer::export_image` divides.
 *
 * Conventions: every entry point returns 0 on success or a negative FCGPU_E_* code; no C++
 * exception crosses the ABI; `fcgpu_last_error` returns a human-readable message for the last
 * failure on that context (or for a failed `fcgpu_create` when passed NULL).
 * A context is single-owner: one host thread drives one context (= one GPU, one HIP stream).
 */
#ifndef FCGPU_H
#define FCGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ error codes */
#define FCGPU_OK 0
#define FCGPU_E_INVALID (-1)   /* bad argument / inconsistent scene description */
#define FCGPU_E_HIP (-2)       /* a HIP runtime call failed (message has the HIP error string) */
#define FCGPU_E_NOSCENE (-3)   /* render/trace before fcgpu_upload_scene */
#define FCGPU_E_LIMIT (-4)     /* a documented capacity was exceeded (media count, BVH depth ...) */
#define FCGPU_E_NOMEM (-5)     /* device or host allocation failed */
#define FCGPU_E_UNSUPPORTED (-6)

/* ------------------------------------------------------------------ scene description
 * All angles in radians, all colours linear RGB, all transforms as in
 * src/core/transform.hpp:14-17 (T * Ry * Rx * Rz [* S]).                                  */

/* surfaces: src/surfaces/{sphere,plane,mesh}_surface.hpp */
#define FCGPU_SURFACE_SPHERE 0
#define FCGPU_SURFACE_PLANE 1
#define FCGPU_SURFACE_MESH 2

typedef struct fcgpu_surface {
    int32_t kind;
    int32_t has_normals;      /* mesh: per-vertex normals present in the shared `normals` pool */
    int32_t has_uvs;          /* mesh: per-vertex uvs present in the shared `uvs` pool          */
    int32_t reserved;
    double position[3];
    double rotation[3];       /* euler x,y,z */
    double scale[3];          /* mesh only (prs_transform); must be 1,1,1 for sphere / plane    */
    double radius;            /* sphere: sphere_surface.hpp:11                                   */
    double size[2];           /* plane: full extent along local x and z, plane_surface.hpp:10    */
    uint32_t vertex_offset;   /* mesh: first vertex in the shared pools                          */
    uint32_t vertex_count;
    uint32_t index_offset;    /* mesh: first index in the shared `indices` pool                  */
    uint32_t index_count;     /* multiple of 3; indices are relative to vertex_offset            */
} fcgpu_surface;

/* materials: the classes under src/materials/; every parameter is a constant (const_texture_2d_*) or, through the
 * tex_* handles below, an image / checker texture; tex_normal puts the lobes inside normal_mapping<T>. */
#define FCGPU_MATERIAL_DIFFUSE 0       /* diffuse_material.hpp:18-38       color_a = reflectance               */
#define FCGPU_MATERIAL_MIRROR 1        /* mirror_material.hpp:20-56        color_a = reflectance, roughness     */
#define FCGPU_MATERIAL_GLASS 2         /* glass_material.hpp:19-41         color_a = reflectance, color_b = transmittance, roughness */
#define FCGPU_MATERIAL_PLASTIC 3       /* plastic_material.hpp:24-52       color_a = diffuse, color_b = specular, roughness, ior     */
#define FCGPU_MATERIAL_STANDARD 4      /* standard_material.hpp:29-107     color_a = base colour, roughness[0], metalness, ior       */
#define FCGPU_MATERIAL_TRANSMISSION 5  /* transmission_material.hpp:19-40  color_b = transmittance, roughness   */

typedef struct fcgpu_material {
    int32_t kind;
    int32_t reserved;
    double color_a[3];
    double color_b[3];
    double roughness[2];
    double ior;
    double metalness;
    /* image textures (image_texture_2d_rgb / image_texture_2d_r, image_texture.hpp) in place of the
     * constants above: 0 = constant, k > 0 = textures[k - 1], evaluated at the hit's uv.
     * rgb parameters: color_a, color_b (every material). Single-channel parameters (.r of the image):
     * ior (PLASTIC, STANDARD), metalness and roughness (STANDARD: its roughness is one texture_2d_r).
     * The two-channel roughness of the other materials is a texture_2d_rg, for which the reference
     * has no image class: constants only. tex_normal: the normal map of DIFFUSE, MIRROR and STANDARD
     * (their lobes then sit inside normal_mapping<T>, bsdfs/normal_mapping.hpp); an rgb texture holding
     * a tangent-space normal as n * 0.5 + 0.5 with blue = up (diffuse_material.hpp:20-27). A handle on a
     * parameter the material class does not have gives FCGPU_E_INVALID. */
    int32_t tex_color_a;
    int32_t tex_color_b;
    int32_t tex_roughness;
    int32_t tex_metalness;
    int32_t tex_ior;
    int32_t tex_normal;
} fcgpu_material;

/* media: src/core/medium.hpp:40-54 (uniform_medium). The implicit vacuum (priority -1, ior 1)
 * of src/core/integrator.hpp:115 is not listed.                                              */
typedef struct fcgpu_medium {
    int32_t priority;
    int32_t reserved;
    double ior;
    double color[3];
    double density;
} fcgpu_medium;

/* entity: src/core/acceleration_structure.hpp:12-18. Entity order defines primitive ids:
 * primitive id = (sum of primitive counts of earlier entities) + primitive index in its surface,
 * i.e. the index in the list entity_scene hands to the acceleration structure (scene.hpp:49-55). */
typedef struct fcgpu_entity {
    int32_t surface;          /* index into surfaces  */
    int32_t material;         /* index into materials */
    int32_t medium;           /* index into media, -1 = none */
    int32_t emissive;         /* 1 = const_diffuse_area_light bound to this surface */
    double emission_color[3]; /* const_diffuse_area_light.hpp:11: colour, strength  */
    double emission_strength;
} fcgpu_entity;

/* camera: src/renderer/cameras/perspective_camera.hpp:19-27 + resolution of the render target */
typedef struct fcgpu_camera {
    double position[3];
    double rotation[3];
    double fov;               /* vertical field of view */
    double lens_radius;       /* 0 = pinhole */
    double focus_distance;
    int32_t width;
    int32_t height;
} fcgpu_camera;

#define FCGPU_ENV_NONE 0
#define FCGPU_ENV_CONST 1     /* const_infinity_area_light.hpp: env_color                      */
#define FCGPU_ENV_TEXTURE 2   /* texture_infinity_area_light.hpp:15-131: env_texture           */

/* images: what assets::get_image returns (src/core/assets.cpp:162-234 and the pixel classes under src/images); pixels are
 * the raw bytes of the .asset file, row-major, top row first; copied at upload. */
#define FCGPU_IMAGE_R8 0      /* 1 byte:   (v/255, 0, 0)                  r8_image.hpp      */
#define FCGPU_IMAGE_RGB8 1    /* 3 bytes:  v/255                          rgb8_image.hpp    */
#define FCGPU_IMAGE_SRGB8 2   /* 3 bytes:  srgb_to_rgb, core/color.hpp:13-26  srgb8_image.hpp */
#define FCGPU_IMAGE_RGB32 3   /* 3 floats: as stored                      rgb32_image.hpp   */

typedef struct fcgpu_image {
    int32_t format;
    int32_t width;
    int32_t height;
    int32_t reserved;
    const void* pixels;
} fcgpu_image;

#define FCGPU_FILTER_BOX 0        /* image_texture.hpp:88-97   */
#define FCGPU_FILTER_BILINEAR 1   /* image_texture.hpp:99-124 (clamp addressing) */

/* Textures of material parameters (src/textures/): image_texture_2d_rgb{image, filter, .} / image_texture_2d_r{image,
 * filter, .} (image_texture.hpp) or checker_texture_2d_rgb{color_a, color_b, scale} (checker_texture.hpp:6-30: a where
 * floor(u * scale) + floor(v * scale) is even, b elsewhere; rgb parameters only) */
#define FCGPU_TEXTURE_IMAGE 0
#define FCGPU_TEXTURE_CHECKER 1
typedef struct fcgpu_texture {
    int32_t image;      /* IMAGE: index into images */
    int32_t filter;     /* IMAGE: FCGPU_FILTER_* */
    int32_t kind;       /* FCGPU_TEXTURE_* */
    int32_t reserved;
    double color_a[3];  /* CHECKER */
    double color_b[3];
    double scale;
} fcgpu_texture;

/* texture_infinity_area_light{pr_transform{{}, rotation}, image_texture_2d_rgb{image, filter,
 * integral_sample_count}, strength, radiance_distribution_resolution} (example_scenes.hpp:339-341) */
typedef struct fcgpu_env_texture {
    int32_t image;                    /* index into images */
    int32_t filter;
    int32_t integral_sample_count;    /* image_texture_2d_rgb::integrate, image_texture.hpp:35-78 */
    int32_t distribution_width;       /* radiance_distribution_resolution (texture_infinity_area_light.hpp:18-38) */
    int32_t distribution_height;
    int32_t reserved;
    double strength;
    double rotation[3];               /* euler x,y,z of the light's pr_transform */
} fcgpu_env_texture;

typedef struct fcgpu_scene_desc {
    const fcgpu_surface* surfaces;
    const fcgpu_material* materials;
    const fcgpu_medium* media;
    const fcgpu_entity* entities;
    const float* positions;   /* 3 floats per vertex, object space (src/core/mesh.hpp:16) */
    const float* normals;     /* 3 floats per vertex or NULL */
    const float* uvs;         /* 2 floats per vertex or NULL */
    const uint32_t* indices;
    uint32_t surface_count;
    uint32_t material_count;
    uint32_t medium_count;
    uint32_t entity_count;
    uint32_t vertex_total;
    uint32_t index_total;
    int32_t env_kind;
    int32_t reserved;
    double env_color[3];
    fcgpu_camera camera;
    const fcgpu_image* images;        /* NULL when image_count == 0 */
    uint32_t image_count;
    uint32_t reserved2;
    fcgpu_env_texture env_texture;    /* read when env_kind == FCGPU_ENV_TEXTURE */
    const fcgpu_texture* textures;    /* NULL when texture_count == 0 */
    uint32_t texture_count;
    uint32_t reserved3;
} fcgpu_scene_desc;

/* ------------------------------------------------------------------ options / records */
#define FCGPU_PRECISION_F32 0   /* fast path: fp32 arithmetic, fp64 film                          */
#define FCGPU_PRECISION_F64 1   /* reference precision: fp64 everywhere, as the reference computes */

#define FCGPU_INTEGRATOR_BSDF 0 /* forward_bsdf_integrator */
#define FCGPU_INTEGRATOR_MIS 1  /* forward_mis_integrator  */

/* Who builds the binary hierarchy that is then collapsed to the 4-wide device layout. The hits do not depend on it
 * (exact t ties aside); the number of node steps per ray does.
 * HOST_SAH (default): the reference's top-down SAH build (bvh_acceleration_structure.hpp:228-365) with a wider split
 *   search: 32 buckets on each of the three axes instead of 12 on the longest one.
 * DEVICE_LBVH: Morton-order linear BVH built on the GPU, for very large meshes where the host recursion
 *   dominates scene upload; lower tree quality.
 * HOST_REFERENCE: the reference's build with its own split search, i.e. its tree and primitive order. */
#define FCGPU_BVH_HOST_SAH 0
#define FCGPU_BVH_DEVICE_LBVH 1
#define FCGPU_BVH_HOST_REFERENCE 2
#define FCGPU_BVH_WIDE8 0x100       /* fcgpu_check_scene only: OR into its bvh_builder to flatten 8-wide (fcgpu_options.bvh_width = 8) */

/* entity_scene::raycast / visibility lift a ray's origin `epsilon` off its surface along the geometric normal
 * (scene.hpp:85-93, 118-131, 146-152). NORMAL does exactly that. ALONG_RAY advances the origin along the ray until
 * it is `epsilon` above the surface (capped at 100 epsilon at grazing angles, the rest along the normal): the ray stays
 * on the reference's line however large epsilon has to be, which is what fp32 needs (its epsilon is 1e-4).
 * DEFAULT: NORMAL in F64 (the reference's arithmetic), ALONG_RAY in F32. */
#define FCGPU_OFFSET_DEFAULT 0
#define FCGPU_OFFSET_NORMAL 1
#define FCGPU_OFFSET_ALONG_RAY 2

typedef struct fcgpu_options {
    int32_t precision;          /* FCGPU_PRECISION_* */
    int32_t sort_materials;     /* must be 1: material-sorted shade queues (one kernel instantiation per lobe type);
                                   anything else is refused with FCGPU_E_UNSUPPORTED */
    double epsilon;             /* ray-origin offset (scene.hpp:182 uses 1e-6); <=0 selects the default:
                                   1e-6 in f64, 1e-4 in f32 (1e-6 is below fp32 resolution at |p|~16) */
    uint32_t max_paths_in_flight; /* 0 = default (2^27); a pass is split into batches of at most this. When the device
                                   cannot hold the queues of a batch, the batch size is halved (down to 2^16) before
                                   FCGPU_E_NOMEM is returned */
    uint32_t bvh_builder;       /* FCGPU_BVH_* */
    uint32_t origin_offset;     /* FCGPU_OFFSET_*: how a ray's origin leaves the surface it starts on */
    uint32_t bvh_width;         /* children per node of the flattened hierarchy: 0 or 4 = the 4-wide 64-byte node (default),
                                   8 = the 8-wide 80-byte node with octant-ordered child slots (measured slower on
                                   MI355X: DESIGN.md section 5; hierarchies deeper than 15 levels are refused) */
} fcgpu_options;

typedef struct fcgpu_tile {
    int32_t x0, y0, w, h;       /* pixel rectangle, y = 0 is the top row (perspective_camera.hpp:53) */
} fcgpu_tile;

typedef struct fcgpu_ray {      /* src/core/math.hpp ray3 + t_max */
    double origin[3];
    double direction[3];
    double t_max;
} fcgpu_ray;

typedef struct fcgpu_hit {
    double t;                   /* parametric distance, +inf on miss */
    double b0, b1;              /* triangle barycentrics of vertex 0 and 1 (0 for sphere / plane) */
    uint32_t prim;              /* primitive id (see fcgpu_entity), 0xFFFFFFFF on miss */
    uint32_t reserved;
} fcgpu_hit;

typedef struct fcgpu_counters {
    uint64_t samples;           /* run_once equivalents rendered since the last clear             */
    uint64_t closest_rays;      /* closest-hit casts (incl. medium-boundary re-casts)             */
    uint64_t shadow_rays;       /* any-hit casts                                                  */
    uint64_t shadow_occluded;
    uint64_t nan_samples;       /* samples zeroed by the NaN/Inf guard (perspective_camera.hpp:126-131) */
    uint64_t stack_overflows;   /* traversal stack overflows (must stay 0)                        */
    uint64_t bvh_nodes;         /* wide nodes uploaded                                            */
    uint64_t primitives;
    uint64_t kernel_launches;
    uint64_t reserved[7];       /* diagnostics: [0] medium-stack overflows, [1] worst-case traversal stack of the BVH,
                                   [2] binary BVH nodes, [3] node steps, [4] primitive tests, [5] traversal-stack
                                   pushes spilled to global memory (deep BVHs only), [6] microseconds the last
                                   fcgpu_upload_scene spent flattening the scene incl. the BVH build */
    uint64_t medium_overflows;  /* pushes onto a path's priority-medium stack beyond the reference's capacity (the vacuum
                                   + 9 nested media, integrator.hpp:24, where the reference writes past its array). The
                                   push is dropped, so the image is not valid: must stay 0 like stack_overflows
                                   (also in reserved[0])                                                          */
    uint64_t medium_deep_pushes;/* diagnostics: pushes that landed in entries 6..9 of that stack                  */
    uint64_t reduce_bytes_copied;/* bytes the copies of fcgpu_reduce_film moved into this context (as root) so far  */
} fcgpu_counters;

/* per-stage device time, measured with HIP events on the context's stream around each launch
 * group while profiling is enabled (bench.py's roofline leg). No counterpart in the reference
 * (it has no profiling). */
#define FCGPU_STAGE_GENERATE 0
#define FCGPU_STAGE_EXTEND 1
#define FCGPU_STAGE_SHADE 2
#define FCGPU_STAGE_SHADOW 3
#define FCGPU_STAGE_ACCUMULATE 4
#define FCGPU_STAGE_SORT 5      /* material sort of the hit queue (k_sort), between extend and shade */
#define FCGPU_STAGE_COUNT 8
typedef struct fcgpu_stage_times {
    double ms[FCGPU_STAGE_COUNT];        /* summed device milliseconds per stage */
    uint64_t launches[FCGPU_STAGE_COUNT];/* timed launch groups per stage */
} fcgpu_stage_times;

typedef struct fcgpu_ctx fcgpu_ctx;

/* ------------------------------------------------------------------ entry points */

/* GPUs this process can use (fcgpu_create takes 0 .. count - 1): what std::thread::hardware_concurrency() is to the
 * reference's worker_count (example_scenes.hpp:259). */
int fcgpu_device_count(int* count);

/* Lifetime. Replaces construction/destruction of fc::renderer (renderer.hpp:23-45). */
int fcgpu_create(int device, fcgpu_ctx** out);
int fcgpu_destroy(fcgpu_ctx* ctx);
const char* fcgpu_last_error(const fcgpu_ctx* ctx);

int fcgpu_set_options(fcgpu_ctx* ctx, const fcgpu_options* opt);

/* Replaces fc::entity_scene's constructor (scene.hpp:35-74) and the BVH build + flattening of
 * bvh_acceleration_structure (bvh_acceleration_structure.hpp:10-24,228-365). Also (re)creates
 * and clears the film at camera.width x camera.height (render_target.hpp:11-13).            */
int fcgpu_upload_scene(fcgpu_ctx* ctx, const fcgpu_scene_desc* scene);

/* Replaces sampler_source::set_sample seeding (sampler.hpp:33, random_sampler.hpp:26-31):
 * the counter-based stream is keyed by (seed, pixel x, pixel y, sample index, dimension).   */
int fcgpu_set_sampler(fcgpu_ctx* ctx, uint64_t seed);

/* Replaces stratified_sampler{sample_count, seed} (stratified_sampler.hpp:12-19, 94-150), the sampler
 * every example scene uses (example_scenes.hpp:90,176,252,305,351): per pixel and dimension the
 * floor(sqrt(sample_count))^2 samples are jittered points of a square grid in shuffled order
 * (fcgpu_sampler.h, FCGPU_SAMPLER_STRATIFIED). Statistical, not bitwise, equivalence with the
 * reference's pcg32-driven tables. Sample indices beyond the grid start a new grid.
 * FCGPU_E_INVALID when sample_count is 0 or above 2^24.                                         */
int fcgpu_set_sampler_stratified(fcgpu_ctx* ctx, uint64_t seed, uint32_t sample_count);

/* Replaces the pixel x sample loop of renderer::worker_thread (renderer.hpp:154-179) calling
 * integrator::run_once (integrator.hpp:18) for every pixel of `tile` and every sample index in
 * [sample_begin, sample_begin + sample_count). Asynchronous; ordered on the context's stream. */
int fcgpu_render_pass(fcgpu_ctx* ctx, const fcgpu_tile* tile, uint32_t sample_begin,
                      uint32_t sample_count, int integrator, int max_path_length,
                      int visible_infinity_area_light);

/* Tile sharding over N GPUs (BASELINE config 5; the reference's analogue is its N workers pulling pixels from one
 * counter, renderer.hpp:65-81, 154-179): the film's rows are cut into bands of `band_rows` rows, i.e. full-width tiles,
 * and this call renders the bands b with b % band_stride == band_first for every sample index of the range, as ONE
 * pass (large batches) however many bands that is. With band_first = rank and band_stride = world size every pixel
 * has exactly one owner, so the summed films (fcgpu_reduce_film, or an RCCL reduce over fcgpu_film_device_ptr) are
 * bit-identical to a single GPU's. Otherwise like fcgpu_render_pass. */
int fcgpu_render_bands(fcgpu_ctx* ctx, uint32_t band_rows, uint32_t band_first, uint32_t band_stride,
                       uint32_t sample_begin, uint32_t sample_count, int integrator, int max_path_length,
                       int visible_infinity_area_light);

/* Parity hooks. Replace acceleration_structure::raycast_surface_point / raycast
 * (acceleration_structure.hpp:38-39; bvh_acceleration_structure.hpp:31-91, 93-138).
 * Host pointers; synchronous.                                                                */
int fcgpu_trace_closest(fcgpu_ctx* ctx, uint64_t n, const fcgpu_ray* rays, fcgpu_hit* hits);
int fcgpu_trace_any(fcgpu_ctx* ctx, uint64_t n, const fcgpu_ray* rays, uint8_t* occluded);

/* Same kernels on rays already resident in device memory (bench: no PCIe in the timed region).
 * `rays_dev` / `hits_dev` / `occluded_dev` are device pointers; asynchronous.                */
int fcgpu_trace_closest_device(fcgpu_ctx* ctx, uint64_t n, const fcgpu_ray* rays_dev,
                               fcgpu_hit* hits_dev);
int fcgpu_trace_any_device(fcgpu_ctx* ctx, uint64_t n, const fcgpu_ray* rays_dev,
                           uint8_t* occluded_dev);

/* Parity hook for the BSDF layer: what `material::evaluate(p) -> bsdf` followed by bsdf::sample_bxdf(u_lobe),
 * bsdf::evaluate(wo, wi), bsdf::pdf_wi(wo, wi) and bsdf::sample_wi(wo, u1, u2) return (core/bsdf.hpp:46-100, 131-141 over
 * core/bxdf.hpp:57-127 and the lobe classes of src/bsdfs/) at a surface point with the given frame, computed ON THE
 * DEVICE by the functions the shade kernels call, in the context's precision. `material`: constants only (no
 * texture handles). ng / tangent / ns / bitangent: the surface point's geometric normal and shading frame
 * (surface_point.hpp:48-62). Host pointers; synchronous. Needs no uploaded scene. */
typedef struct fcgpu_bsdf_query {
    double wo[3], wi[3];
    double u_lobe;              /* sample_bxdf draw */
    double u1[2], u2[2];        /* sample_wi draws */
} fcgpu_bsdf_query;
typedef struct fcgpu_bsdf_result {
    double f[3];                /* evaluate */
    double pdf;                 /* pdf_wi */
    double s_wi[3], s_value[3], s_pdf;  /* sample_wi outputs, zero when it failed */
    int32_t lobe, is_delta, s_ok, reserved;
} fcgpu_bsdf_result;
int fcgpu_bsdf_table(fcgpu_ctx* ctx, const fcgpu_material* material, const double* ng, const double* tangent,
                     const double* ns, const double* bitangent, double eta_a, double eta_b, uint64_t n,
                     const fcgpu_bsdf_query* queries, fcgpu_bsdf_result* results);

/* Parity hook: the discrete decisions of every path of the NEXT render passes, to attribute differences between two
 * renders of the same samples (fp32 against fp64: tools/divergence.py). No counterpart in the reference. While enabled a
 * pass must fit one batch; path p of a pass = (sample index within the pass) * pixels + (pixel index within the tile).
 * hit_words[(h - 1) * paths + p]: what path p hit at its h-th vertex: lobe index (2 bits) | primitive kind (2) | index of
 * the triangle / sphere / plane record (28), FCGPU_DECISION_MISS when the ray left the scene, FCGPU_DECISION_DEAD when the
 * path ended without a hit record, 0xFFFFFFFF when it had ended before. flag_words[...]: FCGPU_DECISION_* bits of
 * what was decided there. fcgpu_read_decision_log returns the sizes (either output pointer may be NULL). */
#define FCGPU_DECISION_MISS 0xFFFFFFFEu
#define FCGPU_DECISION_DEAD 0xFFFFFFFDu
#define FCGPU_DECISION_SHADED 1u        /* the hit was shaded                                             */
#define FCGPU_DECISION_SAMPLED 2u       /* bsdf->sample_wi succeeded: the path goes on                    */
#define FCGPU_DECISION_TRANSMITTED 4u   /* the sampled direction crosses the surface                      */
#define FCGPU_DECISION_SHADOW_CAST 8u   /* the light strategy cast a visibility ray (bsdf->evaluate != 0) */
#define FCGPU_DECISION_OCCLUDED 16u     /* ... and it was occluded                                        */
#define FCGPU_DECISION_LOBE_SHIFT 8     /* bits 8..11: lobe type shaded (7 = last vertex)                 */
int fcgpu_set_decision_log(fcgpu_ctx* ctx, int enabled);
int fcgpu_read_decision_log(fcgpu_ctx* ctx, uint32_t* bounces, uint64_t* paths, uint32_t* hit_words, uint32_t* flag_words);

/* The no-allocation variant of fcgpu_trace_closest_device. It has NO counterpart in the reference; it exists for callers
 * that cannot call hipMalloc: code under HIP graph capture, tight serving loops, harnesses that forbid allocation. Same
 * kernel, same results (bit for bit); the kernel's scratch -- its statistics words and, for hierarchies deeper than the
 * LDS stack, the per-lane spill columns -- lives in a static device array of 8 MiB + 256 B per GPU that the library holds
 * until the process ends, and the launch uses a fixed grid of 256 blocks. Calls no hipMalloc, hipFree or other
 * allocating routine; asynchronous on the context's stream like the _device entry point.
 * Limits (FCGPU_E_LIMIT beyond them): n <= FCGPU_NOALLOC_MAX_RAYS; the scene's worst-case traversal stack
 * (fcgpu_counters.reserved[1]) <= FCGPU_NOALLOC_MAX_STACK.
 * Every context on one GPU shares that GPU's array: calls that overlap on one GPU (two contexts, two threads) race on
 * it. Run them one after the other (stream order on one context is enough). */
#define FCGPU_NOALLOC_MAX_RAYS (1ull << 30)
#define FCGPU_NOALLOC_MAX_STACK 62
int fcgpu_trace_closest_noalloc(fcgpu_ctx* ctx, uint64_t n, const fcgpu_ray* rays_dev, fcgpu_hit* hits_dev);

/* Film. Replaces render_target (render_target.hpp:15-23, 30-38) and the sum/count that
 * renderer::export_image divides (renderer.hpp:118-135): rgb_sum[(y*W + x)*3 + c] / sample_count
 * is the pixel value the reference writes. Synchronous.                                      */
int fcgpu_clear_film(fcgpu_ctx* ctx);
int fcgpu_read_film(fcgpu_ctx* ctx, double* rgb_sum, uint64_t* sample_count);
/* Checkpoint / resume of a render between sample passes. The reference has no such thing (it writes its image once, at
 * the end: renderer.hpp:115-141), but its film is resumable by construction -- running sums and a count
 * (render_target.hpp:15-23, 45-47) -- and so is this one: fcgpu_save_film copies out the W*H*3 UNSCALED sums exactly as
 * the device holds them (fcgpu_read_film's rgb_sum is these times W*H, a rounding that does not invert) and the sample
 * count; fcgpu_restore_film puts both back into a context that holds a scene of the same resolution. A render that is
 * saved after samples [0, k), restored into a fresh context and continued with [k, n) ends with the film of an
 * uninterrupted render of [0, n), bit for bit (samples are added per pixel in index order either way). Synchronous.  */
int fcgpu_save_film(fcgpu_ctx* ctx, double* raw_sum, uint64_t* sample_count);
int fcgpu_restore_film(fcgpu_ctx* ctx, const double* raw_sum, uint64_t sample_count);
/* Device view of the film (W*H*3 doubles, unscaled radiance sums: multiply by W*H to get the
 * reference's rgb_sum) and of the per-context sample counter, for in-place multi-GPU reduction
 * (the analogue of the per-worker film sum, renderer.hpp:129-133).                           */
int fcgpu_film_device_ptr(fcgpu_ctx* ctx, void** rgb_dev, uint64_t* bytes);
int fcgpu_add_sample_count(fcgpu_ctx* ctx, int64_t delta);
/* Replaces the per-worker film sum of renderer::export_image (renderer.hpp:125-135) for one process that drives
 * several GPUs (one context per device): adds the films and sample counts of all `count` contexts into
 * ctxs[root], then clears the other films and zeroes their sample counts, so that rendering more and reducing again
 * accumulates. A contributor that has rendered only its bands (fcgpu_render_bands, one pattern since its last clear)
 * sends only those rows, and all such contributors copy at once, each on its own stream, into disjoint rows of one
 * staging film the root keeps (W*H*24 bytes on the root's device); a contributor that rendered anything else sends the
 * whole frame. Copies between devices are direct peer accesses (hipDeviceEnablePeerAccess: xGMI) where the devices
 * allow it, hipMemcpyPeerAsync otherwise; no host round trip of the library's own. fcgpu_counters.reduce_bytes_copied
 * of the root counts what was moved. All contexts must hold scenes of the same resolution. Synchronous; the caller must
 * not be rendering on any of the contexts from another thread. With disjoint pixel ownership the adds are additions
 * of zero: the result is bit-identical to one GPU's film. Errors are reported on ctxs[root]. */
int fcgpu_reduce_film(fcgpu_ctx* const* ctxs, int count, int root);

int fcgpu_get_counters(fcgpu_ctx* ctx, fcgpu_counters* out);
int fcgpu_reset_counters(fcgpu_ctx* ctx);
int fcgpu_sync(fcgpu_ctx* ctx);
/* Stage timing (see fcgpu_stage_times). fcgpu_get_stage_times synchronises, then returns and
 * clears the sums accumulated since the previous call.                                       */
int fcgpu_set_profiling(fcgpu_ctx* ctx, int enabled);
int fcgpu_get_stage_times(fcgpu_ctx* ctx, fcgpu_stage_times* out);
/* HIP stream the context launches on (hipStream_t as void*), so callers can bracket launches
 * with their own events.                                                                     */
int fcgpu_stream(fcgpu_ctx* ctx, void** stream);

/* Writes the image exactly as renderer::export_image does (renderer.hpp:115-141): row-major,
 * top row first, float32 RGB, no header, file `<name>.raw`. Synchronous.                     */
int fcgpu_export_image(fcgpu_ctx* ctx, const char* name);

/* Validates a scene description and builds its hierarchy on the host without touching a device: what the
 * entity_scene constructor and bvh_acceleration_structure's build (bvh_acceleration_structure.hpp:10-24) do for the
 * reference, as far as it can fail or be measured before rendering. `bvh_builder`: FCGPU_BVH_HOST_SAH or
 * FCGPU_BVH_HOST_REFERENCE, optionally | FCGPU_BVH_WIDE8 (then wide_nodes counts 8-wide nodes and max_stack the group
 * entries their traversal needs). On failure the message is fcgpu_last_error(NULL) of the calling thread.        */
typedef struct fcgpu_scene_stats {
    uint64_t primitives, triangles;
    uint64_t binary_nodes, wide_nodes;
    uint64_t max_stack;         /* worst-case traversal stack of the wide hierarchy */
    uint64_t lobe_mask;         /* bit k: lobe type k occurs */
    uint64_t reserved[2];
} fcgpu_scene_stats;
int fcgpu_check_scene(const fcgpu_scene_desc* scene, uint32_t bvh_builder, fcgpu_scene_stats* out);

/* Library/ABI version: major*1000 + minor. */
int fcgpu_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FCGPU_H */
