/*
 * raytracer_amd.h — C-ABI of the MI355X (gfx950) backend for the per-pixel tracing loop of
 * matteoilardi/Raytracer.
 *
 * The reference has no plugin / FFI layer.  The seam this library sits behind is the single call
 *
 *     tracer.fire_all_rays([&renderer](const Ray &r) { return (*renderer)(r); }, show_progress);
 *                                                           (reference src/raytracer.cpp:151-153)
 *
 * on an ImageTracer built at src/raytracer.cpp:140-144 with a Renderer from build_renderer
 * (src/raytracer.cpp:147,333-356).  A per-ray std::function cannot be GPU-backed, so the backend
 * replaces the whole fire_all_rays call and keeps its post-condition: HdrImage::pixels filled,
 * W*H Color (3 x f32), row-major, row 0 = top (include/colors.hpp:448-464).
 *
 * Everything here is plain C: pointers, sizes and PODs.  No torch / HIP types in the signatures
 * (a hipStream_t travels as void*).  Every entry point returns rt_status; rt_last_error() gives the
 * thread-local message of the last failing call.  No exception crosses this boundary.
 *
 * Matrices: a "hom matrix" is 12 floats = the 3x3 linear part row-major followed by the translation
 * vector, exactly HomMatrix::linear_part / HomMatrix::translation_vec (include/geometry.hpp:332-333).
 * The flattener copies Transformation::hom_matrix ("fwd") and Transformation::inverse_hom_matrix
 * ("inv") verbatim (include/geometry.hpp:437-438) — the library never re-inverts.
 */
#ifndef RAYTRACER_AMD_H
#define RAYTRACER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__) || defined(__clang__)
#define RT_API __attribute__((visibility("default")))
#else
#define RT_API
#endif

typedef enum rt_status {
  RT_OK = 0,
  RT_ERR_INVALID_ARGUMENT = 1, /* bad sizes / indices / enum values / limits exceeded           */
  RT_ERR_HIP = 2,              /* a HIP runtime call failed (message carries hipGetErrorString) */
  RT_ERR_NO_DEVICE = 3,        /* no gfx950 device visible                                      */
  RT_ERR_OUT_OF_MEMORY = 4
} rt_status;

/* Sphere / Plane: include/shapes.hpp:193,309 */
enum { RT_SHAPE_SPHERE = 0, RT_SHAPE_PLANE = 1 };
/* CSGObject::Operation: include/shapes.hpp:380 (same order) */
enum { RT_CSG_UNION = 0, RT_CSG_INTERSECTION = 1, RT_CSG_DIFFERENCE = 2, RT_CSG_FUSION = 3 };
/* DiffusiveBRDF / SpecularBRDF: include/materials.hpp:180,219 */
enum { RT_BRDF_DIFFUSE = 0, RT_BRDF_SPECULAR = 1 };
/* UniformPigment / CheckeredPigment / ImagePigment: include/materials.hpp:50,72,102 */
enum { RT_PIGMENT_UNIFORM = 0, RT_PIGMENT_CHECKERED = 1, RT_PIGMENT_IMAGE = 2 };
/* OrthogonalCamera / PerspectiveCamera: include/cameras.hpp:124,140 */
enum { RT_CAMERA_ORTHOGONAL = 0, RT_CAMERA_PERSPECTIVE = 1 };
/* ModeCfg::Mode: src/raytracer.cpp:48 (--mode onoff|flat|pointlight|path) */
enum { RT_MODE_ONOFF = 0, RT_MODE_FLAT = 1, RT_MODE_POINTLIGHT = 2, RT_MODE_PATH = 3 };
/* How path mode consumes random numbers (see rt_render_cfg::rng_mode) */
enum { RT_RNG_PER_NODE_STREAMS = 0, RT_RNG_SERIAL_REFERENCE = 1 };

/* rt_render_cfg::flags */
#define RT_FLAG_EXACT_ARITHMETIC 1u
#define RT_FLAG_OBJECT_SPACE_SPHERES 2u
#define RT_FLAG_ACCEL_BVH 4u
#define RT_FLAG_VALU_ONLY 8u
#define RT_FLAG_SCAN_ALL_TILES 16u

/* Limits of the device CSG evaluator (fixed-capacity, no heap). */
#define RT_MAX_CSG_DEPTH 8    /* nesting depth of CSG nodes below one top-level object */
#define RT_MAX_CSG_LEAVES 32  /* Sphere/Plane leaves below one top-level CSG object    */

/*
 * Object reference used by `objects`, `csg_left`, `csg_right`:
 *   ref >= 0 : index into the shape arrays
 *   ref <  0 : CSG node index = ~ref  (i.e. -ref-1)
 */
typedef int32_t rt_ref;

/*
 * Flattened World (include/shapes.hpp:520-527) + the Materials its Shapes point to
 * (include/shapes.hpp:140, include/scenefiles.hpp:553).  Struct-of-arrays; the caller owns every
 * array, rt_scene_create copies them to the device.
 */
typedef struct rt_scene_desc {
  /* Shapes (top-level and CSG leaves alike) */
  uint32_t n_shapes;
  const uint8_t *shape_kind;      /* [n_shapes] RT_SHAPE_*                                   */
  const uint32_t *shape_material; /* [n_shapes] index into the material arrays               */
  const float *shape_fwd;         /* [n_shapes][12] Object::transformation.hom_matrix         */
  const float *shape_inv;         /* [n_shapes][12] Object::transformation.inverse_hom_matrix */

  /* CSG nodes (include/shapes.hpp:374-391) */
  uint32_t n_csg_nodes;
  const uint8_t *csg_op;    /* [n_csg_nodes] RT_CSG_*      */
  const rt_ref *csg_left;   /* [n_csg_nodes] object1       */
  const rt_ref *csg_right;  /* [n_csg_nodes] object2       */
  const float *csg_fwd;     /* [n_csg_nodes][12]           */
  const float *csg_inv;     /* [n_csg_nodes][12]           */

  /* World::objects in insertion order (ties in closest-hit go to the lowest index,
   * include/shapes.hpp:556) */
  uint32_t n_objects;
  const rt_ref *objects; /* [n_objects] */

  /* Materials (include/materials.hpp:269-286) */
  uint32_t n_materials;
  const uint8_t *material_brdf;              /* [n_materials] RT_BRDF_*           */
  const uint32_t *material_brdf_pigment;     /* [n_materials] index into pigments */
  const uint32_t *material_emission_pigment; /* [n_materials] index into pigments */

  /* Pigments (include/materials.hpp:39-138) */
  uint32_t n_pigments;
  const uint8_t *pigment_kind;        /* [n_pigments] RT_PIGMENT_*                            */
  const float *pigment_color1;        /* [n_pigments][3] uniform colour / checkered colour 1  */
  const float *pigment_color2;        /* [n_pigments][3] checkered colour 2                   */
  const int32_t *pigment_n_intervals; /* [n_pigments] checkered only                          */
  const uint32_t *pigment_tex_offset; /* [n_pigments] image only: first texel in `texels`     */
  const int32_t *pigment_tex_width;   /* [n_pigments] image only                              */
  const int32_t *pigment_tex_height;  /* [n_pigments] image only                              */

  /* Texture pool for ImagePigment: HdrImage::pixels row-major, 3 floats per texel */
  uint32_t n_texels;
  const float *texels; /* [n_texels][3] */

  /* World::light_sources (include/shapes.hpp:498-513) */
  uint32_t n_lights;
  const float *light_point;  /* [n_lights][3] */
  const float *light_color;  /* [n_lights][3] */
  const float *light_radius; /* [n_lights] emission_radius */
} rt_scene_desc;

/* Camera (include/cameras.hpp:73-160) */
typedef struct rt_camera {
  uint32_t kind;         /* RT_CAMERA_*                                                        */
  float fwd[12];         /* Camera::transformation.hom_matrix                                  */
  float asp_ratio;       /* used iff has_asp_ratio != 0                                        */
  uint32_t has_asp_ratio;/* 0: asp = float(width)/float(height) (include/cameras.hpp:184-186)  */
  float distance;        /* PerspectiveCamera::distance                                        */
} rt_camera;

/*
 * Render configuration = ImgSize + ModeCfg + PathCfg (src/raytracer.cpp:35-51) + the constants
 * build_renderer passes (src/raytracer.cpp:337,348,352) + the row sharding of the image.
 */
typedef struct rt_render_cfg {
  int32_t width, height;   /* full image size                                                   */
  int32_t spp_edge;        /* --antialiasing: samples per pixel edge                            */
  uint32_t mode;           /* RT_MODE_*                                                         */
  int32_t n_rays;          /* --n_rays                                                          */
  int32_t roulette;        /* --roulette                                                        */
  int32_t max_depth;       /* --max-depth                                                       */
  uint64_t seq_number;     /* --seq-number: path PCG stream = PCG(42, seq_number)               */
  float background[3];     /* Renderer::background_color                                        */
  float ambient[3];        /* PointLightTracer::ambient_color                                   */
  /*
   * The reference's PointLightTracer mirror loop has no bounce cap and can spin forever
   * (include/renderers.hpp:129-152).  A lane must never spin: after this many specular bounces
   * the sample returns the background colour (what a miss returns) and is counted in
   * rt_stats.capped_samples.
   */
  int32_t max_mirror_bounces;
  /*
   * RT_RNG_PER_NODE_STREAMS: every RAY owns a PCG stream — a camera ray's from (seq_number, global sample index), a
   * scattered ray's from its parent's stream and its index among the children; image independent of GPU count,
   * grid and kernel shape.  Statistical parity.
   * RT_RNG_SERIAL_REFERENCE is not available on the device (the reference consumes one stream
   * serially over the whole image); the value exists so that the CPU oracle can be asked for it.
   */
  uint32_t rng_mode;
  /*
   * Rows this call renders: local row lr in [0, n_local_rows) is image row
   *     (lr / band_rows) * band_stride + band_offset + lr % band_rows .
   * Whole image: band_rows = band_stride = height, band_offset = 0, n_local_rows = height.
   * Rank r of R with 16-row cyclic bands: band_rows = 16, band_stride = 16*R, band_offset = 16*r.
   */
  int32_t band_rows, band_stride, band_offset, n_local_rows;
  /*
   * RT_FLAG_EXACT_ARITHMETIC: path mode evaluates the closest-hit scan with the reference's unfused
   * fp32 operation order (as the deterministic modes always do) instead of the fused-multiply-add
   * form.  Slower; for parity work.  The deterministic modes ignore the flag (always exact).
   * RT_FLAG_OBJECT_SPACE_SPHERES: path mode on large worlds tests every sphere in its own frame
   * (inverse matrix, as the reference does) even when all spheres are translation * uniform scaling
   * and could be tested in world space from (centre, r^2).
   * RT_FLAG_ACCEL_BVH: path mode finds the closest top-level sphere through a bounding-volume
   * hierarchy built at rt_scene_create instead of testing every sphere (the reference has no such
   * structure: World::ray_intersection is a linear scan).  Same hits, same tie rule, far fewer tests on
   * worlds with thousands of spheres; planes and CSG objects are still tested one by one.
   * RT_FLAG_VALU_ONLY: path mode on large worlds evaluates the sphere filter with vector-ALU instructions only
   * (k_render_path_wide) instead of on the matrix cores (k_render_path_tile, the default: the same two inner products per
   * ray and sphere, binary32 operands split exactly into three bf16 parts, binary32 accumulation — same accuracy).  For
   * measurement: the configuration without any MFMA instruction.
   * RT_FLAG_SCAN_ALL_TILES: path mode on worlds of 1024 register spheres or more multiplies EVERY queued ray with every tile
   * of 32 spheres (k_render_path_tile), as smaller worlds always do.  By default such worlds keep their spheres as spatial
   * clusters of 32 with a bounding sphere each and multiply a tile only with the rays whose line meets its bound
   * (k_render_path_tile_cull): the same products for every pair that can give a hit, hence the same frame bit for bit, in
   * less than half the time at 4096 spheres.  For measurement: the kernel whose work does not depend on where the spheres are.
   */
  uint32_t flags;
} rt_render_cfg;

typedef struct rt_stats {
  uint64_t rays;           /* closest-hit / any-hit / shadow queries (World::ray_intersection &c.) */
  uint64_t samples;        /* pixel samples traced                                                */
  uint64_t capped_samples; /* pointlight samples that hit max_mirror_bounces                       */
  float kernel_ms;         /* trace (+ resolve) kernel time from hipEvents; 0 if !timing            */
  /* which kernel traced the frame and how it was launched (measurement reports name it; the CPU oracle leaves "") */
  uint32_t grid, block;    /* workgroups, lanes per workgroup                                      */
  char kernel[48];         /* e.g. "k_render_path_wide<ISO,K=8,W=1>", NUL-terminated               */
} rt_stats;

/* World::ray_intersection result for rt_trace_rays: HitRecord (include/shapes.hpp:65-96) as POD. */
typedef struct rt_hit {
  int32_t object;   /* index into World::objects, -1 = miss                */
  int32_t shape;    /* shape (leaf) index that was hit                     */
  float t;
  float point[3];   /* HitRecord::world_point                              */
  float normal[3];  /* HitRecord::normal (NOT normalised, as the reference) */
  float uv[2];      /* HitRecord::surface_point                            */
} rt_hit;

/* One ray for rt_trace_rays: Ray (include/cameras.hpp:34-41). */
typedef struct rt_ray {
  float origin[3];
  float direction[3];
  float tmin, tmax;
} rt_ray;

typedef struct rt_scene rt_scene; /* opaque: device copy of a flattened World */

/* Library / device ------------------------------------------------------------------------------ */

/* Message of the last failing call on this thread ("" if none). Never NULL. */
RT_API const char *rt_last_error(void);

/* Number of visible HIP devices (0 if none / runtime unavailable). */
RT_API int rt_device_count(void);

/* Scene -------------------------------------------------------------------------------------------
 * Replaces: the World the reference hands to build_renderer (src/raytracer.cpp:147) — flattened
 * once, reused for any number of renders (animation frames change only the camera).
 * Validates every index / enum / CSG limit; copies the arrays to `device`.  The description must be a forest, as the
 * reference's World is (every Object has one owner): a shape or CSG node referenced twice — by objects[] and / or by
 * CSG children — is refused with RT_ERR_INVALID_ARGUMENT; list the shape twice instead. */
RT_API rt_status rt_scene_create(const rt_scene_desc *desc, int device, rt_scene **out);
RT_API void rt_scene_destroy(rt_scene *scene);

/* Render ------------------------------------------------------------------------------------------
 * Replaces: ImageTracer::fire_all_rays (include/cameras.hpp:221-251) driving
 * OnOffTracer / FlatTracer / PointLightTracer / PathTracer::operator()
 * (include/renderers.hpp:62,88,121,208).
 *
 * rgb_out_device: device pointer on the scene's device, n_local_rows*width*3 floats, row-major,
 *                 local row 0 first (HdrImage::pixels layout for the rows this call owns).
 * stream:         hipStream_t as void* (NULL = default stream).  The call enqueues and returns;
 *                 it synchronises only if `stats` is non-NULL (to read the counters back).
 * One render in flight per rt_scene: the handle owns the call's scratch (tally counters, sample buffer, node stacks,
 * timing events), so calls on one handle must be ordered on one stream (or synchronised between); for concurrent
 * renders of one World create one handle per stream.  The calling thread's current device is left as it was.
 */
RT_API rt_status rt_render(rt_scene *scene, const rt_camera *camera, const rt_render_cfg *cfg,
                    float *rgb_out_device, void *stream, rt_stats *stats);

/* rt_render for callers that cannot allocate device memory — code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc.  It has NO counterpart in the reference (which has no device at all).  Same
 * arguments, same result bit for bit as rt_render, but the call's scratch (tally counters, the sample buffer of
 * spp_edge > 1, the node stacks of path mode) is carved out of one file-scope device array of
 * RT_NOALLOC_SCRATCH_BYTES that every GPU the library runs on holds until the process ends; the call makes no
 * hipMalloc / hipFree.  Limits: 1024 + width * n_local_rows * spp_edge^2 * 12 bytes (spp_edge > 1 only) + in path mode
 * at least one workgroup's node stacks per compute unit must fit: 64 bytes (one 16-word node record) x block x
 * max(max_depth, 1) per workgroup — 64 x block when n_rays == 1 — with block = 256 lanes for small worlds and
 * 64 ... 512 for large ones (rt_stats.block); otherwise RT_ERR_INVALID_ARGUMENT with the sizes in rt_last_error() — render fewer rows per
 * call (band_* fields) or use rt_render.  All calls on one GPU share that GPU's array (whatever the scene handle):
 * calls that overlap on one GPU race; order them on one stream or synchronise between them. */
#define RT_NOALLOC_SCRATCH_BYTES (128u * 1024u * 1024u)
RT_API rt_status rt_render_noalloc(rt_scene *scene, const rt_camera *camera, const rt_render_cfg *cfg,
                                   float *rgb_out_device, void *stream, rt_stats *stats);

/* Same, into host memory (allocates a device framebuffer, renders, copies back, synchronises).
 * This is what the C++ adapter for the reference's ImageTracer calls. */
RT_API rt_status rt_render_host(rt_scene *scene, const rt_camera *camera, const rt_render_cfg *cfg,
                         float *rgb_out_host, rt_stats *stats);

/* The whole fire_all_rays call into host memory, with its progress callback: ImageTracer::fire_all_rays(func, callback)
 * (include/cameras.hpp:221-251) reports (col + 1) / W after every column (cameras.hpp:249); here `progress` is called
 * at most n_slices times with increasing values in (0, 1], the last one 1 — the deterministic modes render n_slices
 * horizontal slices queued back to back on one stream and report rows_done / height as each finishes, path mode is one
 * launch whose sample queue is watched from a second stream (fraction of the frame's samples handed out) — and ONE
 * device-to-host copy delivers HdrImage::pixels at the end.  The framebuffer in HBM
 * belongs to the scene handle and is reused from frame to frame (no allocation per frame once it is large enough).
 * cfg must describe the whole image.  progress may be NULL (then the frame is one launch). */
typedef void (*rt_progress_fn)(float fraction_done, void *user);
RT_API rt_status rt_render_frame_host(rt_scene *scene, const rt_camera *camera, const rt_render_cfg *cfg, float *rgb_out_host,
                                      int n_slices, rt_progress_fn progress, void *user, rt_stats *stats);

/* One frame on several GPUs of one node ------------------------------------------------------------
 * Replaces the same call, tracer.fire_all_rays(...) at src/raytracer.cpp:151-153, when more than one GPU is visible
 * (the reference is single-threaded; its README only suggests GNU parallel over animation frames).  The image rows
 * are dealt round-robin to the ranks (row r, r + n, r + 2n, ... to rank r: neighbouring rows cost alike), every rank
 * renders its rows on its own device and stream from its own replica of the World, the bands are gathered on the first
 * rank's device with RCCL point-to-point calls (ncclSend / ncclRecv in one group, every peer over its own xGMI link)
 * and put in place there.  No exchange happens while rendering.  Because a sample's random streams depend only on its
 * global index and sample values are integer sums, the frame is bit-identical to rt_render's for any number of ranks.
 *
 * devices: rank r renders on devices[r]; n_devices == 0 means every visible device.  A device may be listed several
 * times ("virtual ranks" sharing one GPU, for tests and rehearsals); RCCL cannot hold one device twice in a
 * communicator, so then — or with RT_MULTI_PEER_COPIES — the bands travel by hipMemcpyPeerAsync instead.
 * RT_MULTI_RCCL_SELF: rank 0's own band goes through ncclSend / ncclRecv to itself as well (so that the RCCL path can
 * be exercised on a single GPU). */
#define RT_MULTI_PEER_COPIES 1u
#define RT_MULTI_RCCL_SELF 2u
#define RT_MULTI_MAX_TIMED_RANKS 16
typedef struct rt_multi rt_multi; /* opaque */
typedef struct rt_multi_timing {
  int32_t n_ranks;
  float slowest_render_ms;                      /* the frame's kernel time: max over ranks                               */
  float gather_ms;                              /* transfer + placement: HIP events on the first device's stream, which waits  */
                                                /* for every rank's render before the gather starts                         */
  float render_ms[RT_MULTI_MAX_TIMED_RANKS];    /* per rank, HIP events on its stream                                     */
  uint64_t rays[RT_MULTI_MAX_TIMED_RANKS];      /* per rank                                                               */
} rt_multi_timing;
RT_API rt_status rt_multi_create(const rt_scene_desc *desc, const int *devices, int n_devices, uint32_t flags, rt_multi **out);
RT_API void rt_multi_destroy(rt_multi *multi);
RT_API int rt_multi_rank_count(const rt_multi *multi);
RT_API int rt_multi_uses_rccl(const rt_multi *multi); /* 1: the gather goes through RCCL, 0: peer copies */
/* cfg describes the whole image (band fields as for a whole-image rt_render).  rgb_out_device0: device pointer on
 * devices[0], height * width * 3 floats.  Synchronises before returning.  stats (summed over ranks; kernel_ms = the
 * slowest rank) and timing may be NULL. */
RT_API rt_status rt_multi_render(rt_multi *multi, const rt_camera *camera, const rt_render_cfg *cfg, float *rgb_out_device0,
                                 rt_stats *stats, rt_multi_timing *timing);
/* The same into host memory (HdrImage::pixels): one device-to-host copy from the first rank's device. */
RT_API rt_status rt_multi_render_host(rt_multi *multi, const rt_camera *camera, const rt_render_cfg *cfg, float *rgb_out_host,
                                      rt_stats *stats, rt_multi_timing *timing);
/* ... with ImageTracer::fire_all_rays' progress reports (include/cameras.hpp:249, once per image column there): `progress`
 * is called with n_reports rising fractions in (0, 1] while the ranks render — the share of the frame's pixel samples their
 * sample queues have handed out, read by small copies on a second stream per device — the last one (1.0) when the frame is in
 * rgb_out_host.  progress may be NULL. */
RT_API rt_status rt_multi_render_frame_host(rt_multi *multi, const rt_camera *camera, const rt_render_cfg *cfg, float *rgb_out_host,
                                            int n_reports, rt_progress_fn progress, void *user, rt_stats *stats, rt_multi_timing *timing);

/* Batch closest-hit --------------------------------------------------------------------------------
 * Replaces: World::ray_intersection (include/shapes.hpp:547-561) for n rays at once; one ray per
 * lane.  rays / hits are HOST pointers (debug / parity entry point, not a hot path). */
RT_API rt_status rt_trace_rays(rt_scene *scene, const rt_ray *rays, size_t n, rt_hit *hits);

/* Batch shadow test: World::offset_if_visible (include/shapes.hpp:584-603).
 * viewer/surface/normal: [n][3] host floats; visible[n] receives 1/0. */
RT_API rt_status rt_offset_if_visible(rt_scene *scene, const float *viewer, const float *surface,
                               const float *normal, size_t n, uint8_t *visible);

/* Antialiasing jitter of sample k: the 2k-th and (2k+1)-th PCG(42,54).random_float()
 * (include/cameras.hpp:232-233, include/random.hpp:52-70), by LCG jump-ahead on the device.
 * Host pointers; parity entry point for the exact-jitter claim. */
RT_API rt_status rt_aa_jitter(int device, const uint64_t *sample_index, size_t n, float *u_out, float *v_out);

/* Batch BRDF::scatter_ray directions: DiffusiveBRDF (include/materials.hpp:199-211, consumes two
 * draws of the given PCG via PCG::random_phong(1), include/random.hpp:75-84) or SpecularBRDF
 * (include/materials.hpp:252-261, consumes none).  pcg: [n][2] = {state, inc} in/out; in_dir,
 * normal, dir_out: [n][3].  flags: RT_FLAG_EXACT_ARITHMETIC = the reference's formulas (libm acos / sin / cos,
 * IEEE divisions), what path mode runs under that flag; 0 = the forms path mode runs by default (reciprocal
 * square root, cos theta = sqrt(x), sin theta = sqrt(1 - x), hardware sin / cos of phi).  Host pointers; parity
 * entry point. */
RT_API rt_status rt_scatter_rays(int device, uint32_t brdf_kind, uint32_t flags, uint64_t *pcg, const float *in_dir,
                                 const float *normal, size_t n, float *dir_out);

/* ---------------------------------------------------------------------------------------------
 * Tone-mapping tail: what main() does with the rendered HdrImage before it writes the PNG
 * (src/raytracer.cpp:176-186) — HdrImage::normalize_image, clamp_image and the per-component
 * transform of write_ldr_image (include/colors.hpp:469-516), on the framebuffer while it is still
 * in HBM, so that a PNG-only run copies 3 bytes per pixel to the host instead of 12.
 * ------------------------------------------------------------------------------------------ */
/* Accumulate the log-luminosities in the reference's pixel order in binary32 (one wave, same roundings
 * as HdrImage::average_luminosity's serial loop); default: exact sum in 64-bit fixed point. */
#define RT_TONEMAP_SERIAL_SUM 1u

typedef struct rt_tonemap_cfg {
  float alpha;                 /* --alpha: normalize_image's factor */
  float gamma;                 /* --gamma */
  uint32_t has_avg_luminosity; /* 1: use avg_luminosity (dark mode, src/raytracer.cpp:180) */
  float avg_luminosity;
  float delta;                 /* DEFAULT_DELTA_LOG = 1e-10f (include/colors.hpp:41) */
  uint32_t flags;              /* RT_TONEMAP_* */
} rt_tonemap_cfg;

/* HdrImage::average_luminosity (include/colors.hpp:469-476) of n_pixels rgb pixels in device memory. */
RT_API rt_status rt_average_luminosity(int device, const float *rgb_device, size_t n_pixels, float delta,
                                       uint32_t flags, void *stream, float *avg_out);

/* normalize_image(alpha[, avg]) + clamp_image + pow(x, 1/gamma) * 255 rounded to 8 bits
 * (include/colors.hpp:481-516).  ldr_out_device: 3 * n_pixels bytes, rgb, row 0 = top (the buffer
 * write_ldr_image hands to stbi_write_png).  hdr_out_device: optional, the normalised and clamped
 * floats (may alias rgb_device).  avg_used_out: optional, the average luminosity that was used.
 * Float images must be 16-byte aligned and the 8-bit image 4-byte aligned (whole hipMalloc'ed buffers are). */
RT_API rt_status rt_tonemap(int device, const float *rgb_device, size_t n_pixels, const rt_tonemap_cfg *cfg,
                            uint8_t *ldr_out_device, float *hdr_out_device, void *stream, float *avg_used_out);

/* The same with host buffers (copies in, copies the 8-bit image out). */
RT_API rt_status rt_tonemap_host(int device, const float *rgb_host, size_t n_pixels, const rt_tonemap_cfg *cfg,
                                 uint8_t *ldr_out_host, float *avg_used_out);

#ifdef __cplusplus
}
#endif
#endif /* RAYTRACER_AMD_H */
