// build_gpu.hpp — the device-side tree build (build_gpu.hip) as the host code sees it.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "layout.hpp"

namespace rthip {

struct GpuBuildIO {
	char* blob32;  // f32 scene blob on the device: nodes at 0, then primitives, gates, materials, quantised nodes
	char* blob64;  // f64 scene blob
	uint32_t prims_off32, mats_off32, qnodes_off32, prims_off64, mats_off64;
	int32_t first, count;  // the primitives to build over: [first, first + count) of both primitive arrays
	int32_t leaf_size;     // primitives per leaf, 1..7
	int32_t n_leaves;      // ceil(count / leaf_size), at least 2
	int32_t node_base;     // 0, or 1 when node 0 pairs a leaf of oversized primitives (left) with this tree (right)
	int32_t n_nodes;       // node_base + n_leaves - 1
	double clo[3], cinv[3];  // box of the primitive centres: low corner and 1 / extent (0 for a flat axis)
	bool quantise;         // also write the quantised copy of the f32 nodes
};
struct GpuBuildResult {
	int32_t levels;   // inner nodes on the longest path from the hierarchy's root to a leaf
	bool quantised;   // the quantised nodes were written and `frame` describes their grid
	QFrame frame;
};
// Runs on `stream` and waits for it.  Allocates and frees its scratch memory (about 200 bytes per primitive).
hipError_t gpu_build_tree(const GpuBuildIO& io, hipStream_t stream, GpuBuildResult* res);

// Refit: the tree keeps its shape, its boxes are fitted again to primitives that have moved
// (build_gpu.hip).  Works on any device tree (surface-area, reference-shaped or built on the GPU).
struct GpuRefitIO {
	char* blob32;
	char* blob64;
	uint32_t prims_off32, prims_off64, qnodes_off32;
	int32_t n_nodes;   // inner nodes; node 0 is the root
	int32_t levels;    // inner nodes on the longest path from node 0 to a leaf
	bool big_leaf;     // node 0 pairs the leaf of oversized primitives (left) with the tree proper (right)
	bool quantise;     // also rewrite the quantised copy of the f32 nodes
};
struct GpuRefitResult {
	int32_t unreached;  // nodes no path from node 0 leads to within `levels` steps (0 for a sound tree; nothing is fitted otherwise)
	bool quantised;
	QFrame frame;       // centre and step of the new grid (start and big_leaf are the caller's)
};
// Runs on `stream` and waits for it.  Scratch: 100 bytes per node, allocated and freed.
hipError_t gpu_refit_tree(const GpuRefitIO& io, hipStream_t stream, GpuRefitResult* res);
// The id words (Prim::id) of the primitives in their order on the device, into host memory.
hipError_t gpu_slot_ids(const char* blob32, uint32_t prims_off32, int32_t n_prims, int32_t* host_ids, hipStream_t stream);

// ---- what hangs on the reference's tree, computed on the device (refpass_gpu.hip) ---------------
struct GpuRefPass {  // device arrays; the holder calls release()
	rthip_object* objs = nullptr;  // the objects as the caller gave them
	int32_t n = 0;
	int32_t* rank = nullptr;                               // per object id: tie rank
	uint32_t *gate_first = nullptr, *gate_count = nullptr;  // per object id: its gate boxes in the arrays below, 0 and 0 for none
	Gate<double>* gates64 = nullptr;
	Gate<float>* gates32 = nullptr;
	int64_t n_gates = 0;
	void release();
};
// Inner-node levels on the longest path of the reference's tree over n objects, and its number of
// leaves: both follow from n alone (src/bvh.rs:52,62: leaves of fewer than six, splits at len / 2).
int ref_tree_levels(int64_t n);
int64_t ref_tree_leaves(int64_t n);
// The checks of host.cpp's check_scene_objects for objects in device memory: first_bad = the lowest
// index of an invalid object or -1; hittable = objects that can be hit at all.
hipError_t gpu_check_objects(const rthip_object* dev_objs, int64_t n, int64_t* first_bad, int64_t* hittable, hipStream_t stream);
// Copies the objects (from host or device memory) and computes tie ranks and gates on the device.  The objects must have
// been checked (finite fields, known kinds).  Runs on `stream` and waits for it; about 250 bytes of scratch per object.
hipError_t gpu_reference_pass(const rthip_object* objs, int64_t n, hipStream_t stream, GpuRefPass* out);
// The primitive and the material of object (dev_ids[s * id_stride] & id mask) into slot s of the four
// arrays; the ids may be the id words of those very primitives (stride 8 from &p32[0].id).
hipError_t gpu_emit_prims(const GpuRefPass& ref, const int32_t* dev_ids, int32_t id_stride, int32_t count, Prim<float>* p32, Prim<double>* p64, Mat<float>* m32, Mat<double>* m64, hipStream_t stream);
// Refit: counts the slots whose id word (dev_slot_ids[s * id_stride], a Prim::id) names an object that is now of another kind or can no longer be hit.
hipError_t gpu_same_objects(const GpuRefPass& ref, const int32_t* dev_slot_ids, int32_t id_stride, int32_t count, int32_t* mismatches, hipStream_t stream);

}  // namespace rthip
