// host.hpp — host-side half of librthip: seeded scene factories, Camera::new, and the
// reference-shaped BVH build flattened into the arrays the kernels read.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rthip.h"
#include "layout.hpp"

namespace rthip {

// ---- scene factories -------------------------------------------------------------------
// Returns false when `name` is unknown.
bool generate_scene(const std::string& name, uint32_t seed, std::vector<rthip_object>* out, std::string* err);

// ---- camera (src/camera.rs:25-69) --------------------------------------------------------
struct CameraFields {
	double origin[3], pixel00[3], delta_x[3], delta_y[3], disk_x[3], disk_y[3];
	double defocus_angle;
	int width, height;
};
CameraFields camera_new(const rthip_camera_desc& d);
void camera_rotation(double yaw_deg, double pitch_deg, double out[3]);

// ---- BVH (src/bvh.rs:17-81) --------------------------------------------------------------
struct FlatScene {
	// Both precisions of every array, ready to copy to the device.
	std::vector<Node<float>> nodes32;
	std::vector<Prim<float>> prims32;
	std::vector<Mat<float>> mats32;
	std::vector<Gate<float>> gates32;
	std::vector<Node<double>> nodes64;
	std::vector<Prim<double>> prims64;
	std::vector<Mat<double>> mats64;
	std::vector<Gate<double>> gates64;
	std::vector<QNode> qnodes;  // f32, large scenes only (empty otherwise): quantised copy of nodes32
	QFrame qframe{};
	int32_t big_leaf = 0;  // leaf of oversized primitives that every ray tests before the tree (0 = none) ...
	int32_t start = 0;     // ... and where traversal starts then: node 0's right child (an inner node or a leaf code)
	int depth = 0;      // device tree: levels of inner nodes on the longest root-to-leaf path
	int ref_depth = 0;  // the same for the reference's tree
	int n_leaves = 0;
	std::vector<int32_t> leaf_ids;    // reference tree: object ids, leaves left to right, reference order inside a leaf
	std::vector<int32_t> leaf_sizes;
	// Build option tree_gpu: the host has laid out the primitives (the leaf of oversized ones first, in
	// leaf order; the rest in object order), reserved the nodes and written node 0's left child if
	// there is such a leaf; the tree over primitives [first, first + count) is built on the device
	// (build_gpu.hpp), which also fills in depth, start and the quantised nodes.
	struct GpuPlan {
		bool on = false;
		int32_t first = 0, count = 0, leaf_size = 1, n_leaves = 0, node_base = 0;
		int32_t n_nodes = 0;  // node_base + n_leaves - 1 (plan_device_scene keeps only node 0 on the host)
		double clo[3] = {0, 0, 0}, cinv[3] = {0, 0, 0};  // box of the primitive centres: low corner, 1 / extent
		bool quantise = false;
	} gpu;
};
// Knobs of the host build.  A context reads them from the RTHIP_* environment ONCE, when it is
// created (rthip_set_option changes them afterwards); the host-only entry points, which have no
// context, read the environment per call.
struct BuildOptions {
	bool tree_ref = false;     // RTHIP_TREE=ref: the device tree keeps the reference's shape (default: surface-area build)
	bool tree_gpu = false;     // RTHIP_TREE=gpu: the device tree is a linear BVH built on the GPU (scenes of at least two leaves; contexts only)
	int ref_gpu = -1;          // RTHIP_REF_GPU: tie ranks and gates (what hangs on the reference's tree) and the primitive records made on the device,
	                           // for uploads with tree_gpu and for refits: 1 = always, 0 = never (the host's pass), -1 = from 2048 objects on
	                           // (below, the host is done before the device's launches are queued: 0.4 against 0.8 ms for 560 objects)
	int gpu_leaf = 1;          // RTHIP_GPU_LEAF: primitives per leaf of that build, 1..7 (1 traces fastest: 1.4 against 0.5 Gpaths/s with 4 on a million objects)
	int sah_leaf = 4;          // RTHIP_SAH_LEAF: largest leaf of the surface-area build, 1..7
	int sah_kind_cap = 0;      // RTHIP_SAH_KIND_CAP: at most this many spheres and this many cubes per leaf, 0 = no cap
	double sah_cprim = -1.0;   // RTHIP_SAH_CPRIM: cost of a primitive test against a node visit; < 0 = chosen per scene
	int sah_sweep = -1;        // RTHIP_SAH_SWEEP: exact sweep up to this many primitives per range; < 0 = chosen per scene
	int big_leaf_min = 32;     // RTHIP_BIG_LEAF_MIN: scenes larger than this may get the leaf of oversized primitives
	bool no_big_leaf = false;  // RTHIP_NO_BIG_LEAF
	bool no_qnodes = false;    // RTHIP_NO_QNODES
	bool threads = true;       // RTHIP_BUILD_THREADS=1 switches the threaded build off
	bool timing = false;       // RTHIP_BUILD_TIMING: lap times on stderr
	bool tie_reverse = false;  // RTHIP_TIE_REVERSE: objects with equal sort keys in BVH::new end up in reversed instead of
	                           // kept order (sort_unstable_by, src/bvh.rs:49, leaves their order open)
	bool ranks_elsewhere = false;  // not a knob: the caller computes tie ranks and gates on the device (refpass_gpu.hip) and rewrites the primitive
	                               // records there, so the pass over the reference's tree is left out (ranks 0, no gates, no leaf order in the result)
	static BuildOptions from_env();
};
// Returns false (with a message) for an empty scene or non-finite input.
bool build_flat_scene(const rthip_object* objs, int64_t n, FlatScene* out, std::string* err, const BuildOptions& opt);
// The same objects at new places, for a device tree that keeps its shape: `slot_ids` holds the id
// words (Prim::id) of that tree's primitives in their order on the device.  Fills in primitives (with
// the tie ranks and gates of the reference's tree over the NEW places), gates and materials in that
// order and no nodes.  Returns false when the objects are not the ones the tree was built over: another
// count of objects that can be hit, or another kind of primitive in some place.
bool refit_flat_scene(const rthip_object* objs, int64_t n, const int32_t* slot_ids, int64_t n_slots, FlatScene* out, std::string* err, const BuildOptions& opt);
// Build option tree_gpu with the reference's ranks and gates computed on the device as well: checks
// the objects and lays the scene out as build_flat_scene does for the device build (gpu plan, nodes
// reserved, node 0, leaf of oversized primitives) WITHOUT primitives, materials, ranks or gates;
// slot_ids receives the object id of every primitive place.  out->gpu.on stays false for scenes of
// fewer than two leaves (build those on the host).
bool plan_device_scene(const rthip_object* objs, int64_t n, FlatScene* out, std::vector<int32_t>* slot_ids, std::string* err, const BuildOptions& opt);
// The checks every build starts with (known kinds and materials, finite fields), and the number of
// objects that can be hit at all (every sphere; cubes without a negative half extent).
bool check_scene_objects(const rthip_object* objs, int64_t n, std::string* err, int64_t* hittable = nullptr);
// Checks the invariant every culling decision rests on: each f32 child box (centre and half
// extent), and each quantised one, contains the f32 bounds of every primitive below it; every
// primitive that can be hit is in exactly one leaf; the tree is no deeper than reported.
// Returns the number of violations (0 = sound) and describes the first one.
int64_t validate_flat_scene(const FlatScene& s, int64_t n_objects, std::string* first);
// FNV-1a over everything the device would receive: equal scenes, equal builds, equal value.
uint64_t checksum_flat_scene(const FlatScene& s);

}  // namespace rthip
