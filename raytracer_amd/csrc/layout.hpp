// layout.hpp — the scene as it lies in HBM (and, for small scenes, in LDS).
//
// The tree is binary (a surface-area build by default; the reference's median split on
// request) and stored "children in the parent": an inner node holds the boxes of BOTH
// children, so one 64-byte fetch (f32) decides where a ray goes next and leaves are never
// fetched as nodes.  Primitives are stored in leaf order, 32 bytes each (f32), spheres before
// cubes inside a leaf so that the leaf loop has no per-primitive type branch.  A few primitives
// that dwarf the rest (ground slab, sun) form the first leaf, which every ray tests before it
// enters the tree.  Materials are indexed by object id and fetched once per ray, at shading time.
#pragma once
#include <cstdint>

#include "../../include/rthip.h"

namespace rthip {

constexpr int kTile = 8;              // tiles are 8x8 pixels = one wave of primary rays
constexpr int kTilePixels = 64;
constexpr int kStackMax = 48;         // per-lane traversal stack entries (tree depth limit)
constexpr int kAccumChannels = 3;     // r, g, b planes of 64 words per tile
constexpr double kAccumScale = 68719476736.0;  // 2^36 fixed-point units per 1.0
constexpr int kMatShift = 28;         // Prim::id = object id | material type << 28 | kind << 30
constexpr int kKindShift = 30;
constexpr int kMaxObjects = 1 << 25;  // leaf code holds the first primitive in 25 bits

// child >= 0: index of an inner node.  child < 0: leaf, ~child = first_prim << 6 | n_spheres << 3 | n_cubes.
constexpr int32_t leaf_code(int32_t first, int32_t ns, int32_t nc) { return ~((first << 6) | (ns << 3) | nc); }

// Plane order inside one child box of a Node (and of an f32 cube): lo.xyz, hi.xyz.  Kept behind
// names because it was not always so: a paired order (lo.x, lo.y) (hi.x, hi.y) (lo.z, hi.z) fed
// two-wide packed FMAs, which measured slower than one scalar FMA per plane.
constexpr int kBoxLoX = 0, kBoxLoY = 1, kBoxLoZ = 2, kBoxHiX = 3, kBoxHiY = 4, kBoxHiZ = 5;

template <typename Real>
struct alignas(sizeof(Real) * 16) Node {
	Real box[12];      // per child (left, then right): f64 six planes in the order of kBox* above;
	                   // f32 centre.xyz in the lo slots and half extent.xyz (widened by 2^-20,
	                   // negative for an empty box) in the hi slots
	int32_t child[2];  // left, right
};
static_assert(sizeof(Node<float>) == 64, "f32 node is one 64-byte line");

// Half-size node for f32 trees that live in global memory, where a node visit costs one L1
// access per 16 bytes per lane: the twelve planes as signed 16-bit steps of a scene-wide grid
// (QFrame: plane = centre + q * step per axis), rounded outward by at least one step, so every
// box contains the box it stands for and the closest hit is unchanged.  Two 16-byte loads.
struct alignas(32) QNode {
	int16_t q[12];     // per child: low planes x, y, z (signed steps), then widths x, y, z (unsigned steps)
	int32_t child[2];
};
static_assert(sizeof(QNode) == 32, "quantised node is 32 bytes");
struct QFrame {
	float centre[3], step[3];
	int32_t start;     // where traversal begins: an inner node of the quantised array, or a leaf code
	int32_t big_leaf;  // leaf of oversized primitives tested before the tree, 0 = none
};
constexpr int kQMax = 32767;
static_assert(sizeof(Node<double>) == 128, "f64 node is 128 bytes");

// Sphere: v = centre.xyz, radius; the slots v[4] and v[5] carry two INTEGERS in their low 32 bits
//         (not numbers: read them with prim_gate_first / prim_gate_count): the index of the
//         sphere's first gate box and the number of its gate boxes (see Gate), 0 and 0 for none.
// Cube, f64: v = centre.xyz, half_extent.xyz (the reference's own representation).
// Cube, f32: v = the six face planes lo.xyz, hi.xyz (the order of kBox*), computed
//            in f64 as centre -/+ half_extent and rounded once, so that huge cubes (ground slabs
//            of half extent 1e5) keep exact faces.
template <typename Real>
struct alignas(sizeof(Real) * 8) Prim {
	Real v[6];
	int32_t id;    // object id | material type << kMatShift | kind << kKindShift
	int32_t rank;  // tie rank: among equal t the larger rank wins
};
static_assert(sizeof(Prim<float>) == 32, "f32 primitive is 32 bytes");
// The integer slots of a sphere (little endian: the low word of the slot in either precision).
#if defined(__HIPCC__) || defined(__HIP__)
#define RT_LAYOUT_FN __host__ __device__ inline
#else
#define RT_LAYOUT_FN inline
#endif
template <typename Real> RT_LAYOUT_FN uint32_t prim_gate_first(const Prim<Real>& p) { uint32_t w; __builtin_memcpy(&w, &p.v[4], 4); return w; }
template <typename Real> RT_LAYOUT_FN uint32_t prim_gate_count(const Prim<Real>& p) { uint32_t w; __builtin_memcpy(&w, &p.v[5], 4); return w; }
template <typename Real> RT_LAYOUT_FN void prim_set_gates(Prim<Real>& p, uint32_t first, uint32_t count) {
	p.v[4] = Real(0); p.v[5] = Real(0);
	__builtin_memcpy(&p.v[4], &first, 4);
	__builtin_memcpy(&p.v[5], &count, 4);
}
static_assert(sizeof(Prim<double>) == 64, "f64 primitive is 64 bytes");

// Gate box.  BVH::new (src/bvh.rs:53-56,68) unions the boxes objects cache, and a sphere of
// negative radius caches an INVERTED box (src/shapes/sphere.rs:13-19) that adds nothing to a
// union.  Where a split of the reference's tree separates such an inner sphere from the outer
// shell that would have covered it, the reference's boxes above it are smaller than its true
// bounds, and the reference reaches it only for rays that pass its per-axis test
// (src/math.rs:356-371) on those boxes.  Such a sphere carries those boxes as gates and counts
// as hit only if the ray passes the reference's test on each of them (the integer slots of a
// sphere's Prim say which).  For every other object that test is implied by hitting the
// object, and nothing is stored.
template <typename Real>
struct Gate {
	Real box[6];  // lo.xyz hi.xyz as the reference computes them
};

template <typename Real>
struct alignas(sizeof(Real) * 4) Mat {
	Real albedo[3];
	Real param;  // Lambertian: emission ; Metal: fuzz ; Dielectric: ir
};

template <typename Real>
struct CameraT {  // src/camera.rs:4-12
	Real origin[3], pixel00[3], delta_x[3], delta_y[3], disk_x[3], disk_y[3];
	Real defocus_angle;
};

}  // namespace rthip
