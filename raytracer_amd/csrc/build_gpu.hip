// build_gpu.hip — the device tree built ON the GPU (build option "tree_gpu").
//
// The reference builds its tree on the host by recursive median splits (src/bvh.rs:17-81).  The
// closest hit does not depend on the tree that finds it (tie ranks and gates carry what the
// reference's own tree decides, see host.cpp), so the tree the kernels walk may come from any
// builder.  This one is a linear BVH: Morton codes of the primitive centres, one radix sort,
// leaves of `leaf_size` consecutive primitives, the hierarchy over the leaves' codes found by
// every inner node on its own (Karras 2012), and the boxes bottom up, one launch per level.  It is for scenes that change from frame to frame: a million
// primitives take 1.3 ms of kernels, where the host's surface-area build takes a quarter of a
// second; the tree is coarser and traces slower than that one.
//
// Input, already in the scene blobs (both precisions): the primitives [first, first + count) in
// any order and their materials in the same order; node_base + n_leaves - 1 zeroed nodes.
// Output: the primitives and materials permuted into leaf order (spheres before cubes inside a
// leaf), nodes [node_base, node_base + n_leaves - 1) with inner node 0 of the hierarchy at
// node_base, and, when node_base is 1, the right child of node 0 (whose left child, the leaf of
// oversized primitives, the host has written).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <vector>

#include "build_gpu.hpp"

namespace rthip {
namespace {

constexpr int kBlock = 256;
constexpr int kLoSlot[3] = {kBoxLoX, kBoxLoY, kBoxLoZ}, kHiSlot[3] = {kBoxHiX, kBoxHiY, kBoxHiZ};

struct Box6 { double lo[3], hi[3]; };

__device__ __forceinline__ unsigned long long spread21(unsigned long long v) {  // bit i -> bit 3 i
	v &= 0x1fffffull;
	v = (v | v << 32) & 0x1f00000000ffffull;
	v = (v | v << 16) & 0x1f0000ff0000ffull;
	v = (v | v << 8) & 0x100f00f00f00f00full;
	v = (v | v << 4) & 0x10c30c30c30c30c3ull;
	v = (v | v << 2) & 0x1249249249249249ull;
	return v;
}

// 63-bit Morton code of every primitive's centre inside the box of all centres.
__global__ void codes_kernel(GpuBuildIO io, unsigned long long* keys, uint32_t* vals) {
	const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= (uint32_t)io.count) return;
	const Prim<double>* prims = reinterpret_cast<const Prim<double>*>(io.blob64 + io.prims_off64);
	const Prim<double>& p = prims[io.first + i];
	unsigned long long code = 0;
	for (int k = 0; k < 3; k++) {
		double u = (p.v[k] - io.clo[k]) * io.cinv[k];  // the centre: slots 0..2 of spheres and of f64 cubes
		u = fmin(fmax(u, 0.0), 1.0);
		code |= spread21((unsigned long long)(u * 2097151.0)) << (2 - k);
	}
	keys[i] = code;
	vals[i] = i;
}

// Bounds of a primitive as each precision stores it (host.cpp bounds64 / bounds32).
__device__ __forceinline__ void prim_bounds(const Prim<double>& p64, const Prim<float>& p32, Box6* b64, Box6* b32) {
	const bool sphere = ((p64.id >> kKindShift) & 1) == RTHIP_SPHERE;
	for (int k = 0; k < 3; k++) {
		const double e = fabs(sphere ? p64.v[3] : p64.v[3 + k]);
		b64->lo[k] = p64.v[k] - e; b64->hi[k] = p64.v[k] + e;
		if (sphere) {
			const double c = (double)p32.v[k], r = fabs((double)p32.v[3]);
			b32->lo[k] = c - r; b32->hi[k] = c + r;
		} else {
			const double lo = (double)p32.v[kLoSlot[k]], hi = (double)p32.v[kHiSlot[k]];
			b32->lo[k] = fmin(lo, hi); b32->hi[k] = fmax(lo, hi);
		}
	}
}
__device__ __forceinline__ void box_join(Box6* a, const Box6& b) {
	for (int k = 0; k < 3; k++) { a->lo[k] = fmin(a->lo[k], b.lo[k]); a->hi[k] = fmax(a->hi[k], b.hi[k]); }
}

// One thread per leaf: moves its primitives and materials into place (spheres first), writes the
// leaf's code, key and boxes.
__global__ void leaves_kernel(GpuBuildIO io, const unsigned long long* keys, const uint32_t* vals,
                              const Prim<float>* in_p32, const Prim<double>* in_p64, const Mat<float>* in_m32, const Mat<double>* in_m64,
                              int32_t* leaf_codes, unsigned long long* leaf_keys, Box6* leaf_b64, Box6* leaf_b32) {
	const uint32_t leaf = blockIdx.x * kBlock + threadIdx.x;
	if (leaf >= (uint32_t)io.n_leaves) return;
	Prim<float>* p32 = reinterpret_cast<Prim<float>*>(io.blob32 + io.prims_off32);
	Prim<double>* p64 = reinterpret_cast<Prim<double>*>(io.blob64 + io.prims_off64);
	Mat<float>* m32 = reinterpret_cast<Mat<float>*>(io.blob32 + io.mats_off32);
	Mat<double>* m64 = reinterpret_cast<Mat<double>*>(io.blob64 + io.mats_off64);
	const uint32_t j0 = leaf * (uint32_t)io.leaf_size, j1 = min((uint32_t)io.count, j0 + (uint32_t)io.leaf_size);
	Box6 b64, b32;
	for (int k = 0; k < 3; k++) { b64.lo[k] = b32.lo[k] = DBL_MAX; b64.hi[k] = b32.hi[k] = -DBL_MAX; }
	uint32_t pos = (uint32_t)io.first + j0;
	int ns = 0, nc = 0;
	for (int pass = 0; pass < 2; pass++) {
		for (uint32_t j = j0; j < j1; j++) {
			const uint32_t src = vals[j];  // index into the unsorted copies
			const Prim<double> q64 = in_p64[src];
			const bool sphere = ((q64.id >> kKindShift) & 1) == RTHIP_SPHERE;
			if (sphere != (pass == 0)) continue;
			const Prim<float> q32 = in_p32[src];
			p64[pos] = q64; p32[pos] = q32;
			m64[pos] = in_m64[src]; m32[pos] = in_m32[src];
			Box6 a64, a32;
			prim_bounds(q64, q32, &a64, &a32);
			box_join(&b64, a64); box_join(&b32, a32);
			pos++;
			if (sphere) ns++; else nc++;
		}
	}
	leaf_codes[leaf] = leaf_code((int32_t)((uint32_t)io.first + j0), ns, nc);
	leaf_keys[leaf] = keys[j0];
	leaf_b64[leaf] = b64; leaf_b32[leaf] = b32;
}

// Length of the common prefix of the keys of leaves i and j, the leaf index breaking ties.
__device__ __forceinline__ int prefix(const unsigned long long* keys, int n, int i, int j) {
	if (j < 0 || j >= n) return -1;
	const unsigned long long a = keys[i], b = keys[j];
	if (a == b) return 64 + __clz((unsigned)(i ^ j));
	return __clzll((long long)(a ^ b));
}

// One thread per inner node: its range of leaves and where that range splits (Karras 2012).
// parent_*: index of the parent inner node, bit 31 set for "I am its right child"; the root has -1.
// child_ref[2 i + side]: the child as the fitting pass wants it, a leaf index or ~(inner node index).
__global__ void hierarchy_kernel(GpuBuildIO io, const unsigned long long* keys, const int32_t* leaf_codes, int32_t* parent_inner, int32_t* parent_leaf, int32_t* child_ref) {
	const int i = (int)(blockIdx.x * kBlock + threadIdx.x);
	const int n = io.n_leaves;
	if (i >= n - 1) return;
	const int d = prefix(keys, n, i, i + 1) - prefix(keys, n, i, i - 1) >= 0 ? 1 : -1;
	const int dmin = prefix(keys, n, i, i - d);
	int lmax = 2;
	while (prefix(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
	int l = 0;
	for (int t = lmax / 2; t >= 1; t /= 2)
		if (prefix(keys, n, i, i + (l + t) * d) > dmin) l += t;
	const int j = i + l * d;
	const int dnode = prefix(keys, n, i, j);
	int s = 0, t = l;
	do {
		t = (t + 1) / 2;
		if (prefix(keys, n, i, i + (s + t) * d) > dnode) s += t;
	} while (t > 1);
	const int gamma = i + s * d + min(d, 0);
	const int lo = min(i, j), hi = max(i, j);
	Node<float>* n32 = reinterpret_cast<Node<float>*>(io.blob32);
	Node<double>* n64 = reinterpret_cast<Node<double>*>(io.blob64);
	int32_t child[2];
	if (lo == gamma) { child[0] = leaf_codes[gamma]; parent_leaf[gamma] = i; child_ref[2 * i] = gamma; }
	else { child[0] = io.node_base + gamma; parent_inner[gamma] = i; child_ref[2 * i] = ~gamma; }
	if (hi == gamma + 1) { child[1] = leaf_codes[gamma + 1]; parent_leaf[gamma + 1] = i | (int32_t)0x80000000; child_ref[2 * i + 1] = gamma + 1; }
	else { child[1] = io.node_base + gamma + 1; parent_inner[gamma + 1] = i | (int32_t)0x80000000; child_ref[2 * i + 1] = ~(gamma + 1); }
	for (int c = 0; c < 2; c++) { n32[io.node_base + i].child[c] = child[c]; n64[io.node_base + i].child[c] = child[c]; }
	if (i == 0) parent_inner[0] = -1;
}

// A child's box into its parent's node, in the forms the kernels read (host.cpp write_node): f64
// as planes; f32 as centre and half extent, the half extent taken from the rounded centre, widened
// by 2^-20 and rounded up.
__device__ __forceinline__ float float_at_least(double x) {  // smallest float >= x, for x > 0
	float f = (float)x;
	if ((double)f < x) f = __uint_as_float(__float_as_uint(f) + 1u);
	return f;
}
__device__ __forceinline__ void store_child_box(Node<float>* n32, Node<double>* n64, int side, const Box6& b64, const Box6& b32) {
	for (int k = 0; k < 3; k++) {
		n64->box[6 * side + kLoSlot[k]] = b64.lo[k];
		n64->box[6 * side + kHiSlot[k]] = b64.hi[k];
		const double lo = b32.lo[k], hi = b32.hi[k];
		if (lo <= hi) {
			const float centre = (float)(0.5 * lo + 0.5 * hi);
			const double half = fmax(hi - (double)centre, (double)centre - lo);
			n32->box[6 * side + kLoSlot[k]] = centre;
			n32->box[6 * side + kHiSlot[k]] = float_at_least(half * (1.0 + 0x1p-20) + 1e-37);
		} else {
			n32->box[6 * side + kLoSlot[k]] = 0.0f;
			n32->box[6 * side + kHiSlot[k]] = -1e30f;
		}
	}
}
// Boxes, bottom up, one launch per level of the hierarchy (deepest first): a launch boundary is all the
// ordering the pass needs.  (One kernel in which the second child to arrive at a node carries on
// upwards needs a fence and reads past the caches at every step, and took 10.7 ms for a million
// leaves where these launches take under one.)
// depth_kernel: inner nodes above each inner node; the deepest count + 1 is the stack a ray can need.
__global__ void depth_kernel(GpuBuildIO io, const int32_t* parent_inner, int32_t* depth_inner, int32_t* deepest) {
	const int i = (int)(blockIdx.x * kBlock + threadIdx.x);
	if (i >= io.n_leaves - 1) return;
	int d = 0;
	for (int32_t p = parent_inner[i]; p != -1; p = parent_inner[p & 0x7fffffff]) d++;
	depth_inner[i] = d;
	atomicMax(deepest, d + 1);
}
// fit_level_kernel: every inner node of depth `level` takes its children's boxes (a leaf's own, or the
// one an inner child computed in the launch before), writes them into its node and keeps their union.
__global__ void fit_level_kernel(GpuBuildIO io, int level, const int32_t* depth_inner, const int32_t* child_ref, const Box6* leaf_b64, const Box6* leaf_b32,
                                 Box6* node_b64, Box6* node_b32) {
	const int i = (int)(blockIdx.x * kBlock + threadIdx.x);
	if (i >= io.n_leaves - 1 || depth_inner[i] != level) return;
	Node<float>* n32 = reinterpret_cast<Node<float>*>(io.blob32);
	Node<double>* n64 = reinterpret_cast<Node<double>*>(io.blob64);
	Box6 u64, u32;
	for (int side = 0; side < 2; side++) {
		const int32_t ref = child_ref[2 * i + side];
		const Box6 b64 = ref >= 0 ? leaf_b64[ref] : node_b64[~ref];
		const Box6 b32 = ref >= 0 ? leaf_b32[ref] : node_b32[~ref];
		store_child_box(&n32[io.node_base + i], &n64[io.node_base + i], side, b64, b32);
		if (side == 0) { u64 = b64; u32 = b32; } else { box_join(&u64, b64); box_join(&u32, b32); }
	}
	node_b64[i] = u64; node_b32[i] = u32;
	if (i == 0 && io.node_base == 1) {
		// the root of the hierarchy: with a leaf of oversized primitives beside it, node 0 holds
		// that leaf on the left and this tree on the right
		store_child_box(&n32[0], &n64[0], 1, u64, u32);
		n32[0].child[1] = n64[0].child[1] = io.node_base;
	}
}

// Quantised copy of the f32 nodes (QNode, layout.hpp; host.cpp does the same for its own trees):
// planes in steps of the frame's grid, rounded outward by at least a step.
__global__ void quantise_kernel(GpuBuildIO io, QFrame f, int32_t first_node, int* overflow) {
	const int i = first_node + (int)(blockIdx.x * kBlock + threadIdx.x);
	if (i >= io.n_nodes) return;
	const Node<float>& n = reinterpret_cast<const Node<float>*>(io.blob32)[i];
	QNode q;
	for (int c = 0; c < 2; c++) {
		for (int k = 0; k < 3; k++) {
			const float centre = n.box[6 * c + kLoSlot[k]], half = n.box[6 * c + kHiSlot[k]];
			if (half < 0.0f) { q.q[6 * c + kLoSlot[k]] = kQMax; q.q[6 * c + kHiSlot[k]] = 0; continue; }  // empty box
			const double l = (double)centre - (double)half, h = (double)centre + (double)half;
			double ql = floor((l - (double)f.centre[k]) / (double)f.step[k]) - 1.0;
			double qh = ceil((h - (double)f.centre[k]) / (double)f.step[k]) + 1.0;
			if (ql < -kQMax || qh > kQMax) *overflow = 1;
			ql = fmax(ql, (double)-kQMax); qh = fmin(qh, (double)kQMax);
			q.q[6 * c + kLoSlot[k]] = (int16_t)ql;
			q.q[6 * c + kHiSlot[k]] = (int16_t)(uint16_t)(qh - ql);
		}
		q.child[c] = n.child[c];
	}
	reinterpret_cast<QNode*>(io.blob32 + io.qnodes_off32)[i] = q;
}

// ---- refit ------------------------------------------------------------------------------
// Level of every inner node below node 0, one launch per level from the top: a node of level L
// labels its inner children L + 1.
__global__ void refit_label_kernel(int32_t n_nodes, const Node<float>* n32, int level, int32_t* node_level) {
	const int i = (int)(blockIdx.x * kBlock + threadIdx.x);
	if (i >= n_nodes || node_level[i] != level) return;
	for (int c = 0; c < 2; c++) {
		const int32_t child = n32[i].child[c];
		if (child >= 0 && child < n_nodes) node_level[child] = level + 1;
	}
}
__global__ void refit_unreached_kernel(int32_t n_nodes, const int32_t* node_level, int32_t* unreached) {
	const int i = (int)(blockIdx.x * kBlock + threadIdx.x);
	if (i < n_nodes && node_level[i] < 0) atomicAdd(unreached, 1);
}
// Every node of level `level` (deepest level first): the box of each child, from the leaf's
// primitives as they now are or from what the inner child kept in the launch before, into the
// node in both precisions; their union is kept for the parent.
__global__ void refit_fit_kernel(GpuRefitIO io, int level, const int32_t* node_level, Box6* node_b64, Box6* node_b32) {
	const int i = (int)(blockIdx.x * kBlock + threadIdx.x);
	if (i >= io.n_nodes || node_level[i] != level) return;
	Node<float>* n32 = reinterpret_cast<Node<float>*>(io.blob32);
	Node<double>* n64 = reinterpret_cast<Node<double>*>(io.blob64);
	const Prim<float>* p32 = reinterpret_cast<const Prim<float>*>(io.blob32 + io.prims_off32);
	const Prim<double>* p64 = reinterpret_cast<const Prim<double>*>(io.blob64 + io.prims_off64);
	Box6 u64, u32;
	for (int side = 0; side < 2; side++) {
		const int32_t child = n32[i].child[side];
		Box6 b64, b32;
		if (child >= 0) {
			b64 = node_b64[child]; b32 = node_b32[child];
		} else {
			const uint32_t code = ~(uint32_t)child;
			const uint32_t first = code >> 6, count = ((code >> 3) & 7u) + (code & 7u);
			for (int k = 0; k < 3; k++) { b64.lo[k] = b32.lo[k] = INFINITY; b64.hi[k] = b32.hi[k] = -INFINITY; }  // an empty leaf keeps this
			for (uint32_t j = first; j < first + count; j++) {
				Box6 a64, a32;
				prim_bounds(p64[j], p32[j], &a64, &a32);
				box_join(&b64, a64); box_join(&b32, a32);
			}
		}
		store_child_box(&n32[i], &n64[i], side, b64, b32);
		if (side == 0) { u64 = b64; u32 = b32; } else { box_join(&u64, b64); box_join(&u32, b32); }
	}
	node_b64[i] = u64; node_b32[i] = u32;
}
__global__ void slot_ids_kernel(const Prim<float>* p32, int32_t n, int32_t* ids) {
	const int i = (int)(blockIdx.x * kBlock + threadIdx.x);
	if (i < n) ids[i] = p32[i].id;
}

struct Scratch {  // device memory of one build, freed when it goes out of scope
	std::vector<void*> ptrs;
	~Scratch() { for (void* p : ptrs) (void)hipFree(p); }
	template <typename T> hipError_t get(T** out, size_t n) {
		void* p = nullptr;
		hipError_t e = hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T));
		if (e != hipSuccess) return e;
		ptrs.push_back(p);
		*out = static_cast<T*>(p);
		return hipSuccess;
	}
};

#define GB_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return e_; } while (0)

}  // namespace

// The grid of the quantised nodes: it spans the tree's own bounds as node 0 holds them (the leaf of
// oversized primitives, node 0's left child when there is one, stays outside it).
static void frame_from_root(const Node<float>& node0, bool big_leaf, QFrame* f) {
	for (int k = 0; k < 3; k++) {
		double lo = DBL_MAX, hi = -DBL_MAX;
		for (int c = big_leaf ? 1 : 0; c < 2; c++) {
			const float centre = node0.box[6 * c + kLoSlot[k]], half = node0.box[6 * c + kHiSlot[k]];
			if (half < 0.0f) continue;
			lo = std::fmin(lo, (double)centre - (double)half); hi = std::fmax(hi, (double)centre + (double)half);
		}
		f->centre[k] = (float)(0.5 * (lo + hi));
		const double half = std::fmax(hi - (double)f->centre[k], (double)f->centre[k] - lo);
		f->step[k] = (float)std::fmax(half / (kQMax - 64), 1e-30);  // headroom for the outward rounding
	}
}

hipError_t gpu_build_tree(const GpuBuildIO& io_in, hipStream_t stream, GpuBuildResult* res) {
	GpuBuildIO io = io_in;
	if (io.count < 2 || io.leaf_size < 1 || io.leaf_size > 7 || io.n_leaves != (io.count + io.leaf_size - 1) / io.leaf_size || io.n_leaves < 2 ||
	    io.n_nodes != io.node_base + io.n_leaves - 1 || (io.node_base != 0 && io.node_base != 1))
		return hipErrorInvalidValue;
	Scratch mem;
	const size_t n = (size_t)io.count, nl = (size_t)io.n_leaves;
	unsigned long long *keys_a, *keys_b, *leaf_keys;
	uint32_t *vals_a, *vals_b;
	Prim<float>* in_p32; Prim<double>* in_p64; Mat<float>* in_m32; Mat<double>* in_m64;
	int32_t *leaf_codes, *parent_inner, *parent_leaf, *deepest, *child_ref, *depth_inner;
	Box6 *leaf_b64, *leaf_b32, *node_b64, *node_b32;
	int* overflow;
	GB_TRY(mem.get(&keys_a, n)); GB_TRY(mem.get(&keys_b, n)); GB_TRY(mem.get(&vals_a, n)); GB_TRY(mem.get(&vals_b, n));
	GB_TRY(mem.get(&in_p32, n)); GB_TRY(mem.get(&in_p64, n)); GB_TRY(mem.get(&in_m32, n)); GB_TRY(mem.get(&in_m64, n));
	GB_TRY(mem.get(&leaf_codes, nl)); GB_TRY(mem.get(&leaf_keys, nl)); GB_TRY(mem.get(&leaf_b64, nl)); GB_TRY(mem.get(&leaf_b32, nl));
	GB_TRY(mem.get(&parent_inner, nl)); GB_TRY(mem.get(&parent_leaf, nl)); GB_TRY(mem.get(&node_b64, nl)); GB_TRY(mem.get(&node_b32, nl));
	GB_TRY(mem.get(&child_ref, 2 * nl)); GB_TRY(mem.get(&depth_inner, nl));
	GB_TRY(mem.get(&deepest, 1)); GB_TRY(mem.get(&overflow, 1));
	GB_TRY(hipMemsetAsync(deepest, 0, sizeof(int32_t), stream));
	GB_TRY(hipMemsetAsync(overflow, 0, sizeof(int), stream));
	// the unsorted primitives and materials, aside: the scene blobs receive them in leaf order
	GB_TRY(hipMemcpyAsync(in_p32, io.blob32 + io.prims_off32 + (size_t)io.first * sizeof(Prim<float>), n * sizeof(Prim<float>), hipMemcpyDeviceToDevice, stream));
	GB_TRY(hipMemcpyAsync(in_p64, io.blob64 + io.prims_off64 + (size_t)io.first * sizeof(Prim<double>), n * sizeof(Prim<double>), hipMemcpyDeviceToDevice, stream));
	GB_TRY(hipMemcpyAsync(in_m32, io.blob32 + io.mats_off32 + (size_t)io.first * sizeof(Mat<float>), n * sizeof(Mat<float>), hipMemcpyDeviceToDevice, stream));
	GB_TRY(hipMemcpyAsync(in_m64, io.blob64 + io.mats_off64 + (size_t)io.first * sizeof(Mat<double>), n * sizeof(Mat<double>), hipMemcpyDeviceToDevice, stream));
	const unsigned grid_p = (unsigned)((n + kBlock - 1) / kBlock), grid_l = (unsigned)((nl + kBlock - 1) / kBlock);
	hipLaunchKernelGGL(codes_kernel, dim3(grid_p), dim3(kBlock), 0, stream, io, keys_a, vals_a);
	GB_TRY(hipGetLastError());
	size_t temp_bytes = 0;
	GB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, keys_a, keys_b, vals_a, vals_b, (int)n, 0, 63, stream));
	char* temp;
	GB_TRY(mem.get(&temp, temp_bytes));
	GB_TRY(hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys_a, keys_b, vals_a, vals_b, (int)n, 0, 63, stream));
	hipLaunchKernelGGL(leaves_kernel, dim3(grid_l), dim3(kBlock), 0, stream, io, keys_b, vals_b, in_p32, in_p64, in_m32, in_m64, leaf_codes, leaf_keys, leaf_b64, leaf_b32);
	GB_TRY(hipGetLastError());
	hipLaunchKernelGGL(hierarchy_kernel, dim3(grid_l), dim3(kBlock), 0, stream, io, leaf_keys, leaf_codes, parent_inner, parent_leaf, child_ref);
	GB_TRY(hipGetLastError());
	hipLaunchKernelGGL(depth_kernel, dim3(grid_l), dim3(kBlock), 0, stream, io, parent_inner, depth_inner, deepest);
	GB_TRY(hipGetLastError());
	int32_t levels = 0;
	GB_TRY(hipMemcpyAsync(&levels, deepest, sizeof(levels), hipMemcpyDeviceToHost, stream));
	GB_TRY(hipStreamSynchronize(stream));
	res->levels = levels;
	res->quantised = false;
	if (levels > kStackMax) return hipSuccess;  // deeper than a traversal stack: the caller turns the scene down
	for (int level = levels - 1; level >= 0; level--) {
		hipLaunchKernelGGL(fit_level_kernel, dim3(grid_l), dim3(kBlock), 0, stream, io, level, depth_inner, child_ref, leaf_b64, leaf_b32, node_b64, node_b32);
		GB_TRY(hipGetLastError());
	}
	Node<float> node0;
	GB_TRY(hipMemcpyAsync(&node0, io.blob32, sizeof(node0), hipMemcpyDeviceToHost, stream));
	GB_TRY(hipStreamSynchronize(stream));
	if (io.quantise) {
		QFrame& f = res->frame;
		frame_from_root(node0, io.node_base == 1, &f);
		f.start = io.node_base;
		f.big_leaf = 0;  // the caller knows the leaf's code
		const int first_node = io.node_base;
		hipLaunchKernelGGL(quantise_kernel, dim3((unsigned)((io.n_nodes - first_node + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, io, f, first_node, overflow);
		GB_TRY(hipGetLastError());
		int over = 0;
		GB_TRY(hipMemcpyAsync(&over, overflow, sizeof(over), hipMemcpyDeviceToHost, stream));
		GB_TRY(hipStreamSynchronize(stream));
		res->quantised = over == 0;
	}
	return hipSuccess;
}

hipError_t gpu_refit_tree(const GpuRefitIO& io, hipStream_t stream, GpuRefitResult* res) {
	if (io.n_nodes < 1 || io.levels < 1 || io.levels > kStackMax) return hipErrorInvalidValue;
	Scratch mem;
	const size_t nn = (size_t)io.n_nodes;
	int32_t *node_level, *flags;
	Box6 *node_b64, *node_b32;
	GB_TRY(mem.get(&node_level, nn)); GB_TRY(mem.get(&node_b64, nn)); GB_TRY(mem.get(&node_b32, nn)); GB_TRY(mem.get(&flags, 2));
	GB_TRY(hipMemsetAsync(node_level, 0xff, nn * sizeof(int32_t), stream));
	GB_TRY(hipMemsetAsync(node_level, 0, sizeof(int32_t), stream));  // node 0 is level 0
	GB_TRY(hipMemsetAsync(flags, 0, 2 * sizeof(int32_t), stream));
	const unsigned grid = (unsigned)((nn + kBlock - 1) / kBlock);
	const Node<float>* n32 = reinterpret_cast<const Node<float>*>(io.blob32);
	for (int level = 0; level + 1 < io.levels; level++) {
		hipLaunchKernelGGL(refit_label_kernel, dim3(grid), dim3(kBlock), 0, stream, io.n_nodes, n32, level, node_level);
		GB_TRY(hipGetLastError());
	}
	hipLaunchKernelGGL(refit_unreached_kernel, dim3(grid), dim3(kBlock), 0, stream, io.n_nodes, node_level, flags);
	GB_TRY(hipGetLastError());
	int32_t unreached = 0;
	GB_TRY(hipMemcpyAsync(&unreached, flags, sizeof(unreached), hipMemcpyDeviceToHost, stream));
	GB_TRY(hipStreamSynchronize(stream));
	res->unreached = unreached;
	res->quantised = false;
	if (unreached) return hipSuccess;
	for (int level = io.levels - 1; level >= 0; level--) {
		hipLaunchKernelGGL(refit_fit_kernel, dim3(grid), dim3(kBlock), 0, stream, io, level, node_level, node_b64, node_b32);
		GB_TRY(hipGetLastError());
	}
	if (io.quantise) {
		Node<float> node0;
		GB_TRY(hipMemcpyAsync(&node0, io.blob32, sizeof(node0), hipMemcpyDeviceToHost, stream));
		GB_TRY(hipStreamSynchronize(stream));
		frame_from_root(node0, io.big_leaf, &res->frame);
		GpuBuildIO q{};
		q.blob32 = io.blob32; q.qnodes_off32 = io.qnodes_off32; q.n_nodes = io.n_nodes;
		const int first_node = io.big_leaf ? 1 : 0;
		if (io.n_nodes > first_node) {
			hipLaunchKernelGGL(quantise_kernel, dim3((unsigned)((io.n_nodes - first_node + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, q, res->frame, first_node, flags + 1);
			GB_TRY(hipGetLastError());
		}
		int over = 0;
		GB_TRY(hipMemcpyAsync(&over, flags + 1, sizeof(over), hipMemcpyDeviceToHost, stream));
		GB_TRY(hipStreamSynchronize(stream));
		res->quantised = over == 0;
	} else {
		GB_TRY(hipStreamSynchronize(stream));
	}
	return hipSuccess;
}

hipError_t gpu_slot_ids(const char* blob32, uint32_t prims_off32, int32_t n_prims, int32_t* host_ids, hipStream_t stream) {
	if (n_prims < 1) return hipSuccess;  // a scene in which nothing can be hit
	Scratch mem;
	int32_t* ids;
	GB_TRY(mem.get(&ids, (size_t)n_prims));
	hipLaunchKernelGGL(slot_ids_kernel, dim3((unsigned)((n_prims + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream,
	                   reinterpret_cast<const Prim<float>*>(blob32 + prims_off32), n_prims, ids);
	GB_TRY(hipGetLastError());
	GB_TRY(hipMemcpyAsync(host_ids, ids, (size_t)n_prims * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
	return hipStreamSynchronize(stream);
}

}  // namespace rthip
