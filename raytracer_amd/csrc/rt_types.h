// rt_types.h — device-side layout of a flattened World (see include/raytracer_amd.h for the ABI).
//
// HBM layout, chosen for how the tracing loop reads it:
//  * Top-level spheres / planes ("scan lists"): one 64-byte record per object, read WAVE-UNIFORMLY
//    (every lane of a wave tests the same object at the same time) through scalar loads, in
//    World::objects order.  Record = inverse hom matrix (12 f32) + object index + shape index.
//  * Per-shape tables (fwd / inv / kind / material), pigments, materials, lights: gathered per lane,
//    once per ray, for the closest hit only.
//  * CSG: per top-level CSG object a left-to-right leaf list; per leaf the chain of CSG nodes from
//    the root down to its parent; per (node, side) an RPN "point inside sibling?" program.
#pragma once
#include <stdint.h>

namespace rtd {

struct ScanRec {       // 64 B, 64-B aligned: one top-level Sphere or Plane
  float inv[12];       // Object::transformation.inverse_hom_matrix
  int32_t object;      // index in World::objects
  int32_t shape;       // index in the shape tables
  int32_t pad0, pad1;
};

struct CsgTop {        // one top-level CSGObject
  int32_t object;      // index in World::objects
  int32_t leaf_begin;  // range in csg_leaf[]
  int32_t leaf_end;
  int32_t tt;          // 1: the tree has at most CSG_TT_MAX_LEAVES leaves and csg_leaf_tt[] holds their truth tables;
                       // 2: ... and the tables say the tree is the intersection of all its leaves; 3: their union (every hit valid)
};
// Truth-table form of the validity rule (FMA path kernels, trees of up to 6 leaves): a boundary hit on leaf l is kept iff
// every ancestor's rule accepts it, and each rule looks only at "is the point inside the sibling subtree", a boolean
// function of the leaves' inside tests — so validity is ONE boolean function of the inside mask (bit i: the point is
// inside leaf i of the tree), tabulated by the host: bit m of csg_leaf_tt[l] = "kept when the inside mask is m".
constexpr int CSG_TT_MAX_LEAVES = 6;

struct CsgLeaf {       // one Sphere/Plane leaf below a top-level CSG object (left-to-right order)
  int32_t shape;
  int32_t path_begin;  // range in csg_path[]: root ... parent of the leaf
  int32_t path_len;
  int32_t pad;
};

struct CsgStep {       // one CSG node on a root->leaf chain
  int32_t node;        // index in the node tables
  int32_t side;        // 0: the chain continues into object1, 1: into object2
};

struct CsgProgEntry {  // RPN program "is point inside this subtree?"
  int32_t is_op;       // 0: leaf, 1: boolean op of the two previous results
  int32_t arg;         // leaf: shape index; op: RT_CSG_*
  int32_t path_begin;  // leaf: range in csg_prog_path[] (CSG nodes from the subtree root to the leaf's parent)
  int32_t path_len;
};

// Bounding-volume hierarchy over the top-level spheres (optional mode, RT_FLAG_ACCEL_BVH): a binary tree in HEAP order
// (root = slot 1, children of n = 2n, 2n + 1, slot 0 unused), so that traversal needs no stack — see rt_bvh.h.
// One node = 16 bytes: the box quantised to 16 bits per bound on the lattice bvh_origin + q * bvh_scale (rounded
// outwards), and the leaf range.
#ifndef RT_BVH_LEAF_MAX
#define RT_BVH_LEAF_MAX 4
#endif
constexpr int BVH_LEAF_MAX = RT_BVH_LEAF_MAX; // spheres per leaf
struct BvhNode {
  uint32_t x, y, z;    // per axis: lower bound | upper bound << 16
  uint32_t leaf;       // inner node: 0; leaf: number of spheres << 24 | first position in bvh_recs[] / bvh_iso[]
};

struct ShadeRec {      // 32 B per shape: what PathTracer needs from the hit shape's material in ONE load round trip
  float brdf_color[3];   // colour of the BRDF pigment when it is uniform
  uint32_t flags;        // bit 0: SpecularBRDF, bit 1: BRDF pigment is uniform, bit 2: emission pigment is uniform
  float emit_color[3];
  uint32_t material;
};

#ifndef RT_TILE_BF16
#define RT_TILE_BF16 1 // the tile kernel's sphere filter on bf16 MFMAs (operands split 3-way); 0: on binary32 MFMAs (rounds 2-3a)
#endif
constexpr int RT_TILE_WORDS = RT_TILE_BF16 ? 512 : 192;
// From how many sphere tiles on a world's register spheres are ALSO kept as spatial clusters with bounding spheres, and the
// tile kernel multiplies a tile only with the rays whose line meets its bound (rt_tile_cull.h)
// Measured (Mrays/s, plain | culled; scripts/ab_cull.sh): 16 tiles (N=512) 14 176 | 12 267, 32 (N=1024) 8 022 | 9 943, 64 (N=2048) 4 365 | 7 160,
// 128 (N=4096) 2 404 | 5 279.
#ifndef RT_CULL_MIN_TILES
#define RT_CULL_MIN_TILES 32
#endif
constexpr int RT_CULL_W = 4; // waves per workgroup of that kernel: cluster j belongs to wave j % RT_CULL_W
#ifndef RT_CULL_TILES
#define RT_CULL_TILES 1
#endif
constexpr int RT_CULL_C = RT_CULL_TILES; // tiles per cluster (a power of two): cluster j = tiles C j ... C j + C - 1.  Measured at N=4096: 1 -> 5165,
                                         // 2 -> 4702, 4 -> 4039 Mrays/s: tighter bounds beat fewer lists

struct Scene {
  // scan lists
  const ScanRec *spheres; int32_t n_spheres;
  // wide path kernel (rt_wide.h): the spheres held in registers, regrouped in ascending object order as wide_rows rows
  // of 64 (the last one may be partly empty) — records, and world centre + r^2 (valid iff spheres_all_iso); and the
  // ones scanned per lane instead (big spheres, a short leftover), ascending too
  const ScanRec *wide_recs; const float *wide_iso; int32_t n_wide, wide_rows;
  const ScanRec *wide_lane; int32_t n_wide_lane;
  // the same register spheres as MFMA operands (rt_tile.h), RT_TILE_WORDS 32-bit words per tile of 32:
  //  RT_TILE_BF16 (default): 2 K-chunks x 64 lanes x 4 words of bf16 pairs — the 3-way bf16 splits of cx cy cz k laid out in the
  //    32 K slots of two v_mfma_f32_32x32x16_bf16 (rt_tile.h says which slot holds what); a place past n_wide: centre 0, k = 1e30
  //  binary32 build: three rows of 64 floats — lane l takes cx | cy, cz | 1, cz | k of sphere l % 32 for l < 32 | l >= 32;
  //    a place past n_wide holds 0 0 0 and k = +inf (never hit)
  const float *wide_tiles; int32_t n_wide_tiles;
  // rt_tile_cull.h (worlds of RT_CULL_MIN_TILES tiles or more, bf16 build; else n_cull_tiles = 0): the same spheres regrouped into
  // spatial clusters of 32 RT_CULL_C — tiles as above in cluster order; the clusters' bounding spheres as tiles of the same form,
  // one tile per group of 32 RT_CULL_W clusters and wave: row i of tile g RT_CULL_W + w bounds cluster 32 RT_CULL_W g + RT_CULL_W i + w;
  // cluster position -> position in wide_recs (-1: an empty place) and the inverse
  const float *cull_tiles; const float *cull_bounds; const int32_t *cull_rank; const int32_t *cull_pos; int32_t n_cull_tiles;
  int32_t spheres_all_iso;        // every top-level sphere is translation * uniform scaling
  // every transformation of the world (shapes, CSG leaves with their ancestors folded in) is translation * axis-aligned scaling:
  // all off-diagonal entries of shape_fwd / shape_inv / shape_w2l / shape_l2w are zero.  True for every sample scene of the
  // reference and for the synthetic worlds; the small-world FMA kernel then multiplies by the three diagonal entries only
  // (6 instructions per transformed ray instead of 21: the same bits, the zero terms add nothing)
  int32_t all_diag;
  // RT_FLAG_ACCEL_BVH (rt_bvh.h): heap of n_bvh_nodes slots (0: no hierarchy); the spheres inside it regrouped by
  // leaf (records, and centre + r^2 when spheres_all_iso); the spheres too big for it, scanned linearly first
  const BvhNode *bvh; int32_t n_bvh_nodes;
  float bvh_origin[3], bvh_scale[3], bvh_inv_scale[3];
  const ScanRec *bvh_recs;
  const float *bvh_iso;
  const ScanRec *bvh_big; int32_t n_bvh_big, n_bvh_small;
  const ScanRec *planes;  int32_t n_planes;
  const CsgTop *csg_tops; int32_t n_csg_tops;
  int32_t n_objects;
  // shape tables
  const float *shape_fwd;         // [n_shapes][12]
  const float *shape_inv;         // [n_shapes][12]
  const uint8_t *shape_kind;
  const uint32_t *shape_material;
  const ShadeRec *shape_shade;
  // per shape: {index in World::objects (-1: a CSG leaf), position in wide_recs (-1: none), position in bvh_recs (-1: none),
  // bit 0: top-level sphere} — what a ray needs to know about the sphere it LEAVES (rt_device.h self_sphere_exact)
  const int32_t *shape_origin;    // [n_shapes][4]
  // world <-> shape-local transforms with the CSG ancestors' transforms folded in (== shape_inv / shape_fwd for a
  // top-level shape): what the FMA path kernels use instead of walking the chain of nodes
  const float *shape_w2l;         // [n_shapes][12]
  const float *shape_l2w;         // [n_shapes][12]
  // CSG tables
  const CsgLeaf *csg_leaf;
  const unsigned long long *csg_leaf_tt; // [n leaves] truth tables (see CsgTop::tt)
  const CsgStep *csg_path;
  const float *csg_fwd;           // [n_nodes][12]
  const float *csg_inv;
  const uint8_t *csg_op;
  const int32_t *csg_prog_range;  // [n_nodes][2 sides][2] begin,end in csg_prog[]
  const CsgProgEntry *csg_prog;
  const int32_t *csg_prog_path;
  // materials / pigments / lights
  const uint8_t *material_brdf;
  const uint32_t *material_brdf_pigment;
  const uint32_t *material_emission_pigment;
  const uint8_t *material_needs_uv; // 1 if either pigment of the material reads (u,v)
  const uint8_t *pigment_kind;
  const float *pigment_color1;
  const float *pigment_color2;
  const int32_t *pigment_n_intervals;
  const uint32_t *pigment_tex_offset;
  const int32_t *pigment_tex_width;
  const int32_t *pigment_tex_height;
  const float *texels;
  const float *light_point;
  const float *light_color;
  const float *light_radius;
  int32_t n_lights;
};

struct Camera {
  float fwd[12];
  float asp;
  float distance;
  int32_t kind;
};

struct RenderParams {
  Camera cam;
  int32_t width, height, spp_edge, mode;
  int32_t n_rays, roulette, max_depth, max_mirror_bounces;
  uint64_t seq_number;
  float background[3];
  float ambient[3];
  int32_t band_rows, band_stride, band_offset, n_local_rows;
  uint64_t n_samples;          // n_local_rows * width * spp_edge^2
  uint64_t sample_stride;      // path mode: coprime to n_samples, about 0.618 n_samples (see rt_path.h)
  float *samples_out;          // [n_samples][3] (== framebuffer when spp_edge == 1)
  unsigned long long *counters; // [0] next sample; rays and capped samples: spread over slots from COUNTER_BASE (rt_path.h)
  // path mode: one node stack per WAVE in HBM, 64-byte records: node i of wave v at pool[(v * pool_cap + i) * 16 ...]  (rt_wavepath.h)
  float *pool;
  int32_t pool_cap;
  // path mode: pixel samples a wave reserves from the call's queue per atomic, at least (it always reserves what its idle lanes
  // ask for).  What a wave has reserved it must finish alone, so the last reservation of the slowest wave is the launch's tail:
  // set by the host from the launch's size (rt_api.hip)
  int32_t sample_chunk;
  // path mode: fresh pixel samples a wave takes at most per iteration.  A wave tops up only when its whole stack holds fewer
  // children than it has lanes, so the samples it takes then are finished together as one batch before the next are taken;
  // the batch a wave is in when the queue runs dry is the launch's tail (rt_api.hip says how the host sets it)
  int32_t fresh_max;
  // RT_FLAG_ACCEL_BVH: heap slots and leaf spheres staged in LDS; whether leaves use the (centre, r^2) form (rt_bvh.h)
  int32_t bvh_lds_nodes, bvh_lds_spheres, bvh_iso;
  // sizes of the lookup tables (k_render_path stages them in LDS when they are all small, rt_kernels.h)
  int32_t tab_shapes, tab_materials, tab_pigments, tab_csg_leaves, tab_csg_path, tab_csg_nodes, tab_csg_prog;
};

} // namespace rtd
