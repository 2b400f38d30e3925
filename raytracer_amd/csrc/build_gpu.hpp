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

}  // namespace rthip
