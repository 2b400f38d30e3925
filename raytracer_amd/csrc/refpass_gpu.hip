// refpass_gpu.hip — what hangs on the REFERENCE's tree, computed on the GPU.
//
// BVH::new (src/bvh.rs:17-81) sorts the objects of every node by min + max of their cached boxes on
// the node's widest axis and splits at len / 2; leaves hold fewer than six.  Nothing of that tree is
// traversed on the GPU, but three things follow from it and have to be exactly the reference's: the
// tie rank of every object (leaves rank left to right, objects inside a leaf in reverse:
// src/scene.rs:105, src/bvh.rs:98-104), and the gate boxes of spheres with a negative radius (the
// boxes above them that do not contain their true bounds, root first: src/math.rs:356-371 decides
// on those whether the reference reaches them).  host.cpp computes them by building that tree; this
// file computes the same numbers level by level without a tree: the ranges of a level are known from
// the object count alone (every node splits in the middle), so a level is one segmented reduction
// (boxes, widest axis), one stable segmented sort (two radix sorts: by key, then by the range's
// start) and a step of every position to its half.  A million objects take 19 levels.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <functional>
#include <map>
#include <vector>

#include "build_gpu.hpp"

namespace rthip {
namespace {

constexpr int kBlock = 256;
constexpr int kLoSlot[3] = {kBoxLoX, kBoxLoY, kBoxLoZ}, kHiSlot[3] = {kBoxHiX, kBoxHiY, kBoxHiZ};
constexpr int kHeadLoopMax = 192;  // ranges up to this long are reduced by the lane at their start

// Doubles as unsigned integers of the same order (for radix sorts and integer atomic min / max).
__device__ __forceinline__ unsigned long long ordered(double x) {
	const unsigned long long b = (unsigned long long)__double_as_longlong(x);
	return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double unordered(unsigned long long u) {
	const unsigned long long b = (u >> 63) ? (u & 0x7fffffffffffffffull) : ~u;
	return __longlong_as_double((long long)b);
}

// Per object: the box the reference caches (src/shapes/sphere.rs:19, cube.rs:17: centre -/+ the size
// AS GIVEN, inverted for negative sizes), one array per axis and side.
__global__ void ref_boxes_kernel(const rthip_object* objs, int32_t n, double* rlo, double* rhi, int32_t* order) {
	const int i = (int)(blockIdx.x * kBlock + threadIdx.x);
	if (i >= n) return;
	const rthip_object& o = objs[i];
	for (int k = 0; k < 3; k++) {
		const double e = o.kind == RTHIP_SPHERE ? o.extent[0] : o.extent[k];
		rlo[(size_t)k * n + i] = o.center[k] - e;
		rhi[(size_t)k * n + i] = o.center[k] + e;
	}
	order[i] = i;
}

// Spheres of a negative radius, numbered in object order (the only objects a box above them may
// fail to contain: they alone can have gates).
__global__ void neg_flags_kernel(const rthip_object* objs, int32_t n, uint32_t* flags) {
	const int i = (int)(blockIdx.x * kBlock + threadIdx.x);
	if (i < n) flags[i] = (objs[i].kind == RTHIP_SPHERE && objs[i].extent[0] < 0.0) ? 1u : 0u;
}
__global__ void neg_index_kernel(int32_t n, const uint32_t* flags, const uint32_t* at, int32_t* neg_index) {
	const int i = (int)(blockIdx.x * kBlock + threadIdx.x);
	if (i < n) neg_index[i] = flags[i] ? (int32_t)at[i] : -1;
}

// The state of a position: the range [b, e) of the node it belongs to at the current level, and
// whether that node was a leaf of an earlier level (frozen: nothing moves there any more).
struct Ranges { uint32_t *b, *e; uint8_t* frozen; };

__global__ void ranges_init_kernel(Ranges r, int32_t n) {
	const int i = (int)(blockIdx.x * kBlock + threadIdx.x);
	if (i >= n) return;
	r.b[i] = 0; r.e[i] = (uint32_t)n; r.frozen[i] = 0;
}

// Box of every live range, long ranges: a wave that lies inside one range reduces in registers and
// adds one atomic per plane; a wave across a boundary lets every lane do its own.
__global__ void seg_box_init_kernel(Ranges r, int32_t n, unsigned long long* segbox) {
	const int i = (int)(blockIdx.x * kBlock + threadIdx.x);
	if (i >= n || r.frozen[i] || r.b[i] != (uint32_t)i) return;
	for (int k = 0; k < 3; k++) { segbox[(size_t)k * n + i] = ~0ull; segbox[(size_t)(3 + k) * n + i] = 0ull; }
}
__global__ void seg_box_long_kernel(Ranges r, int32_t n, const int32_t* order, const double* rlo, const double* rhi, unsigned long long* segbox) {
	const int i = (int)(blockIdx.x * kBlock + threadIdx.x);
	const bool live = i < n && !r.frozen[i];
	const uint32_t b = live ? r.b[i] : 0xffffffffu;
	const uint32_t b0 = (uint32_t)__shfl((int)b, 0);
	const bool uniform = __all(b == b0) && b0 != 0xffffffffu;
	unsigned long long v[6];
	if (live) {
		const int32_t id = order[i];
		for (int k = 0; k < 3; k++) { v[k] = ordered(rlo[(size_t)k * n + id]); v[3 + k] = ordered(rhi[(size_t)k * n + id]); }
	}
	if (uniform) {
		for (int off = 32; off >= 1; off >>= 1)
			for (int k = 0; k < 3; k++) {
				const unsigned long long a = __shfl_xor(v[k], off), c = __shfl_xor(v[3 + k], off);
				v[k] = min(v[k], a); v[3 + k] = max(v[3 + k], c);
			}
		if ((threadIdx.x & 63) == 0)
			for (int k = 0; k < 3; k++) { atomicMin(&segbox[(size_t)k * n + b], v[k]); atomicMax(&segbox[(size_t)(3 + k) * n + b], v[3 + k]); }
	} else if (live) {
		for (int k = 0; k < 3; k++) { atomicMin(&segbox[(size_t)k * n + b], v[k]); atomicMax(&segbox[(size_t)(3 + k) * n + b], v[3 + k]); }
	}
}
// ... short ranges: the lane at the start of a range walks it.
__global__ void seg_box_short_kernel(Ranges r, int32_t n, const int32_t* order, const double* rlo, const double* rhi, unsigned long long* segbox) {
	const int i = (int)(blockIdx.x * kBlock + threadIdx.x);
	if (i >= n || r.frozen[i] || r.b[i] != (uint32_t)i) return;
	double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
	const uint32_t e = r.e[i];
	for (uint32_t j = (uint32_t)i; j < e; j++) {
		const int32_t id = order[j];
		for (int k = 0; k < 3; k++) { lo[k] = fmin(lo[k], rlo[(size_t)k * n + id]); hi[k] = fmax(hi[k], rhi[(size_t)k * n + id]); }
	}
	for (int k = 0; k < 3; k++) { segbox[(size_t)k * n + i] = ordered(lo[k]); segbox[(size_t)(3 + k) * n + i] = ordered(hi[k]); }
}

// Sort key of every position at this level (src/bvh.rs:18-49: min + max on the axis of largest
// extent, the first such axis on ties), the range's start packed with the object id as the value;
// and the gate test of spheres with a negative radius against this level's box.
__global__ void keys_kernel(Ranges r, int32_t n, int level, const int32_t* order, const double* rlo, const double* rhi, const unsigned long long* segbox,
                            const rthip_object* objs, const int32_t* neg_index, int32_t max_levels,
                            unsigned long long* keys, unsigned long long* vals, unsigned long long* gate_mask, double* gate_tmp) {
	const int i = (int)(blockIdx.x * kBlock + threadIdx.x);
	if (i >= n) return;
	const int32_t id = order[i];
	const uint32_t b = r.b[i];
	vals[i] = ((unsigned long long)b << 32) | (uint32_t)id;
	if (r.frozen[i]) { keys[i] = 0ull; return; }
	double lo[3], hi[3];
	for (int k = 0; k < 3; k++) { lo[k] = unordered(segbox[(size_t)k * n + b]); hi[k] = unordered(segbox[(size_t)(3 + k) * n + b]); }
	int axis = 0;
	double best = 0.0;
	for (int a = 0; a < 3; a++) {
		const double range = hi[a] - lo[a];
		if (a == 0 || range > best) { axis = a; best = range; }
	}
	const double key = (rlo[(size_t)axis * n + id] + rhi[(size_t)axis * n + id]) + 0.0;  // + 0.0: -0 and +0 are one key
	keys[i] = ordered(key);
	const int32_t g = neg_index[id];
	if (g >= 0) {
		const rthip_object& o = objs[id];
		const double rad = fabs(o.extent[0]);
		bool inside = true;
		for (int k = 0; k < 3; k++) inside = inside && lo[k] <= o.center[k] - rad && o.center[k] + rad <= hi[k];
		if (!inside) {
			gate_mask[g] |= 1ull << level;  // one position holds this object: no other thread writes the word
			double* t = gate_tmp + ((size_t)g * max_levels + level) * 6;
			for (int k = 0; k < 3; k++) { t[k] = lo[k]; t[3 + k] = hi[k]; }
		}
	}
}
__global__ void unpack_kernel(int32_t n, const unsigned long long* vals, uint32_t* seg, int32_t* ids) {
	const int i = (int)(blockIdx.x * kBlock + threadIdx.x);
	if (i >= n) return;
	seg[i] = (uint32_t)(vals[i] >> 32);
	ids[i] = (int32_t)(uint32_t)vals[i];
}
// Every position steps into its half of the node it was in; the positions of a leaf freeze.
__global__ void advance_kernel(Ranges r, int32_t n) {
	const int i = (int)(blockIdx.x * kBlock + threadIdx.x);
	if (i >= n || r.frozen[i]) return;
	const uint32_t b = r.b[i], e = r.e[i], len = e - b;
	if (len < 6) { r.frozen[i] = 1; return; }
	const uint32_t mid = b + len / 2;  // src/bvh.rs:62-79: the upper half is the right child
	if ((uint32_t)i < mid) r.e[i] = mid; else r.b[i] = mid;
}

// The leaves now stand left to right in `order`.  Tie ranks; gate counts in leaf order.
__global__ void ranks_kernel(Ranges r, int32_t n, const int32_t* order, const int32_t* neg_index, const unsigned long long* gate_mask,
                             int32_t* rank, uint32_t* counts) {
	const int i = (int)(blockIdx.x * kBlock + threadIdx.x);
	if (i >= n) return;
	const int32_t id = order[i];
	const uint32_t b = r.b[i], e = r.e[i];
	rank[id] = (int32_t)(b + (e - 1 - (uint32_t)i));
	const int32_t g = neg_index[id];
	counts[i] = g >= 0 ? (uint32_t)__popcll(gate_mask[g]) : 0u;
}
__global__ void gates_kernel(int32_t n, const int32_t* order, const int32_t* neg_index, const unsigned long long* gate_mask, const double* gate_tmp,
                             int32_t max_levels, const uint32_t* counts, const uint32_t* firsts, uint32_t* gate_first, uint32_t* gate_count,
                             Gate<double>* g64, Gate<float>* g32) {
	const int i = (int)(blockIdx.x * kBlock + threadIdx.x);
	if (i >= n) return;
	const int32_t id = order[i];
	const uint32_t count = counts[i];
	gate_first[id] = count ? firsts[i] : 0u;
	gate_count[id] = count;
	if (!count) return;
	const int32_t g = neg_index[id];
	unsigned long long mask = gate_mask[g];
	uint32_t at = firsts[i];
	while (mask) {
		const int level = __ffsll((long long)mask) - 1;
		mask &= mask - 1;
		const double* t = gate_tmp + ((size_t)g * max_levels + level) * 6;
		for (int k = 0; k < 6; k++) { g64[at].box[k] = t[k]; g32[at].box[k] = (float)t[k]; }
		at++;
	}
}

// The primitive and the material of object ids[s] into slot s, in both precisions (host.cpp emit_prim).
__global__ void emit_kernel(const rthip_object* objs, const int32_t* ids, int32_t id_stride, int32_t count, const int32_t* rank, const uint32_t* gate_first, const uint32_t* gate_count,
                            Prim<float>* p32, Prim<double>* p64, Mat<float>* m32, Mat<double>* m64) {
	const int s = (int)(blockIdx.x * kBlock + threadIdx.x);
	if (s >= count) return;
	const int32_t id = ids[(size_t)s * id_stride] & ((1 << kMatShift) - 1);  // (ids may be the id words of the very primitives rewritten here)
	const rthip_object& o = objs[id];
	Prim<double> a;
	Prim<float> f;
	for (int q = 0; q < 6; q++) { a.v[q] = 0.0; f.v[q] = 0.0f; }
	for (int q = 0; q < 3; q++) { a.v[q] = o.center[q]; f.v[q] = (float)o.center[q]; }
	if (o.kind == RTHIP_SPHERE) {
		a.v[3] = o.extent[0];
		f.v[3] = (float)o.extent[0];
		prim_set_gates(a, gate_first[id], gate_count[id]);
		prim_set_gates(f, gate_first[id], gate_count[id]);
	} else {
		for (int q = 0; q < 3; q++) {
			a.v[3 + q] = o.extent[q];
			f.v[kLoSlot[q]] = (float)(o.center[q] - o.extent[q]);
			f.v[kHiSlot[q]] = (float)(o.center[q] + o.extent[q]);
		}
	}
	a.id = f.id = id | (o.material << kMatShift) | (o.kind << kKindShift);
	a.rank = f.rank = rank[id];
	p64[s] = a; p32[s] = f;
	Mat<double> ma;
	Mat<float> mf;
	for (int k = 0; k < 3; k++) { ma.albedo[k] = o.albedo[k]; mf.albedo[k] = (float)o.albedo[k]; }
	ma.param = o.param; mf.param = (float)o.param;
	m64[s] = ma; m32[s] = mf;
}
// Refit: are the objects the ones whose primitives the tree holds?  (kind as before, still possible to hit)
__global__ void same_objects_kernel(const rthip_object* objs, int32_t n, const int32_t* slot_ids, int32_t id_stride, int32_t count, int32_t* bad) {
	const int s = (int)(blockIdx.x * kBlock + threadIdx.x);
	if (s >= count) return;
	const int32_t old = slot_ids[(size_t)s * id_stride], id = old & ((1 << kMatShift) - 1);
	bool ok = id >= 0 && id < n;
	if (ok) {
		const rthip_object& o = objs[id];
		ok = ((old >> kKindShift) & 1) == o.kind && (o.kind == RTHIP_SPHERE || (o.extent[0] >= 0.0 && o.extent[1] >= 0.0 && o.extent[2] >= 0.0));
	}
	if (!ok) atomicAdd(bad, 1);
}

// The checks every build starts with (host.cpp check_objects), for objects that live on the device:
// the lowest index of an object with an unknown kind or material or a non-finite field, and the
// number of objects that can be hit at all.
__global__ void check_objects_kernel(const rthip_object* objs, int32_t n, int32_t* first_bad, unsigned long long* hittable) {
	const int i = (int)(blockIdx.x * kBlock + threadIdx.x);
	bool can_hit = false;
	if (i < n) {
		const rthip_object& o = objs[i];
		bool ok = (o.kind == RTHIP_SPHERE || o.kind == RTHIP_CUBE) && o.material >= 0 && o.material <= 2 && isfinite(o.param);
		for (int k = 0; k < 3; k++) ok = ok && isfinite(o.center[k]) && isfinite(o.albedo[k]) && (o.kind == RTHIP_SPHERE ? isfinite(o.extent[0]) : isfinite(o.extent[k]));
		if (!ok) atomicMin(first_bad, i);
		can_hit = ok && (o.kind == RTHIP_SPHERE || (o.extent[0] >= 0.0 && o.extent[1] >= 0.0 && o.extent[2] >= 0.0));
	}
	const unsigned long long votes = __ballot(can_hit);
	if ((threadIdx.x & 63) == 0 && votes) atomicAdd(hittable, (unsigned long long)__popcll(votes));
}

#define RP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return e_; } while (0)

template <typename T> hipError_t dev_alloc(std::vector<void*>* owned, T** out, size_t n) {
	void* p = nullptr;
	hipError_t e = hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T));
	if (e != hipSuccess) return e;
	owned->push_back(p);
	*out = static_cast<T*>(p);
	return hipSuccess;
}
struct Owned {
	std::vector<void*> ptrs;
	~Owned() { for (void* p : ptrs) (void)hipFree(p); }
};

}  // namespace

void GpuRefPass::release() {
	for (void* p : {(void*)objs, (void*)rank, (void*)gate_first, (void*)gate_count, (void*)gates64, (void*)gates32})
		if (p) (void)hipFree(p);
	*this = GpuRefPass();
}

int ref_tree_levels(int64_t n) {
	int levels = 0;
	for (int64_t len = n; len >= 6; len -= len / 2) levels++;  // the upper half is the longer one
	return levels;
}
int64_t ref_tree_leaves(int64_t n) {
	// a node of fewer than six objects is a leaf, any other splits into len / 2 and the rest: two
	// lengths per level, so the memo stays small
	std::map<int64_t, int64_t> memo;
	std::function<int64_t(int64_t)> leaves = [&](int64_t len) -> int64_t {
		if (len < 6) return 1;
		auto it = memo.find(len);
		if (it != memo.end()) return it->second;
		const int64_t v = leaves(len / 2) + leaves(len - len / 2);
		memo[len] = v;
		return v;
	};
	return leaves(n);
}

hipError_t gpu_reference_pass(const rthip_object* host_objs, int64_t n64, hipStream_t stream, GpuRefPass* out) {
	*out = GpuRefPass();
	if (n64 < 1 || n64 > kMaxObjects) return hipErrorInvalidValue;
	const int32_t n = (int32_t)n64;
	const size_t nn = (size_t)n;
	const int levels = ref_tree_levels(n64);  // inner levels; nodes (leaves included) stand on levels 0 .. levels
	if (levels + 1 > 62) return hipErrorInvalidValue;
	const int32_t max_levels = levels + 1;

	Owned keep;   // what the caller receives (moved out at the end)
	Owned mem;    // scratch
	rthip_object* objs; int32_t* rank; uint32_t *gate_first, *gate_count;
	RP_TRY(dev_alloc(&keep.ptrs, &objs, nn)); RP_TRY(dev_alloc(&keep.ptrs, &rank, nn));
	RP_TRY(dev_alloc(&keep.ptrs, &gate_first, nn)); RP_TRY(dev_alloc(&keep.ptrs, &gate_count, nn));
	double *rlo, *rhi, *gate_tmp;
	int32_t *order_a, *order_b, *d_neg;
	uint32_t *seg_a, *seg_b, *counts, *firsts;
	unsigned long long *segbox, *keys_a, *keys_b, *vals_a, *vals_b, *gate_mask;
	Ranges r;
	const unsigned grid = (unsigned)((nn + kBlock - 1) / kBlock);
	RP_TRY(dev_alloc(&mem.ptrs, &rlo, 3 * nn)); RP_TRY(dev_alloc(&mem.ptrs, &rhi, 3 * nn));
	RP_TRY(dev_alloc(&mem.ptrs, &order_a, nn)); RP_TRY(dev_alloc(&mem.ptrs, &order_b, nn)); RP_TRY(dev_alloc(&mem.ptrs, &d_neg, nn));
	RP_TRY(dev_alloc(&mem.ptrs, &seg_a, nn)); RP_TRY(dev_alloc(&mem.ptrs, &seg_b, nn));
	RP_TRY(dev_alloc(&mem.ptrs, &counts, nn)); RP_TRY(dev_alloc(&mem.ptrs, &firsts, nn));
	RP_TRY(dev_alloc(&mem.ptrs, &segbox, 6 * nn));
	RP_TRY(dev_alloc(&mem.ptrs, &keys_a, nn)); RP_TRY(dev_alloc(&mem.ptrs, &keys_b, nn));
	RP_TRY(dev_alloc(&mem.ptrs, &vals_a, nn)); RP_TRY(dev_alloc(&mem.ptrs, &vals_b, nn));
	RP_TRY(dev_alloc(&mem.ptrs, &r.b, nn)); RP_TRY(dev_alloc(&mem.ptrs, &r.e, nn)); RP_TRY(dev_alloc(&mem.ptrs, &r.frozen, nn));
	int seg_bits = 1;
	while ((1ll << seg_bits) < n64) seg_bits++;
	size_t temp_a = 0, temp_b = 0, temp_c = 0;
	RP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_a, keys_a, keys_b, vals_a, vals_b, n, 0, 64, stream));
	RP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_b, seg_a, seg_b, order_a, order_b, n, 0, seg_bits, stream));
	RP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, temp_c, counts, firsts, n, stream));
	const size_t temp_bytes = std::max(temp_a, std::max(temp_b, temp_c));
	char* temp;
	RP_TRY(dev_alloc(&mem.ptrs, &temp, temp_bytes));

	RP_TRY(hipMemcpyAsync(objs, host_objs, nn * sizeof(rthip_object), hipMemcpyDefault, stream));  // from the host, or from the caller's device memory
	// the spheres that can have gates, numbered (counts and firsts serve as scratch here)
	hipLaunchKernelGGL(neg_flags_kernel, dim3(grid), dim3(kBlock), 0, stream, objs, n, counts);
	RP_TRY(hipGetLastError());
	{
		size_t bytes = temp_bytes;
		RP_TRY(hipcub::DeviceScan::ExclusiveSum(temp, bytes, counts, firsts, n, stream));
	}
	hipLaunchKernelGGL(neg_index_kernel, dim3(grid), dim3(kBlock), 0, stream, n, counts, firsts, d_neg);
	RP_TRY(hipGetLastError());
	uint32_t neg_last_at = 0, neg_last_flag = 0;
	RP_TRY(hipMemcpyAsync(&neg_last_at, firsts + (nn - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
	RP_TRY(hipMemcpyAsync(&neg_last_flag, counts + (nn - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
	RP_TRY(hipStreamSynchronize(stream));
	const int32_t n_neg = (int32_t)(neg_last_at + neg_last_flag);
	RP_TRY(dev_alloc(&mem.ptrs, &gate_mask, (size_t)std::max(n_neg, 1)));
	RP_TRY(dev_alloc(&mem.ptrs, &gate_tmp, (size_t)std::max(n_neg, 1) * (size_t)max_levels * 6));
	RP_TRY(hipMemsetAsync(gate_mask, 0, (size_t)std::max(n_neg, 1) * sizeof(unsigned long long), stream));
	hipLaunchKernelGGL(ref_boxes_kernel, dim3(grid), dim3(kBlock), 0, stream, objs, n, rlo, rhi, order_a);
	hipLaunchKernelGGL(ranges_init_kernel, dim3(grid), dim3(kBlock), 0, stream, r, n);
	RP_TRY(hipGetLastError());
	int64_t len = n64;  // the shorter of the (at most two) node lengths of the level
	for (int level = 0; level <= levels; level++) {
		if (len > kHeadLoopMax) {
			hipLaunchKernelGGL(seg_box_init_kernel, dim3(grid), dim3(kBlock), 0, stream, r, n, segbox);
			hipLaunchKernelGGL(seg_box_long_kernel, dim3(grid), dim3(kBlock), 0, stream, r, n, order_a, rlo, rhi, segbox);
		} else {
			hipLaunchKernelGGL(seg_box_short_kernel, dim3(grid), dim3(kBlock), 0, stream, r, n, order_a, rlo, rhi, segbox);
		}
		hipLaunchKernelGGL(keys_kernel, dim3(grid), dim3(kBlock), 0, stream, r, n, level, order_a, rlo, rhi, segbox, objs, d_neg, max_levels,
		                   keys_a, vals_a, gate_mask, gate_tmp);
		RP_TRY(hipGetLastError());
		size_t bytes = temp_bytes;
		RP_TRY(hipcub::DeviceRadixSort::SortPairs(temp, bytes, keys_a, keys_b, vals_a, vals_b, n, 0, 64, stream));
		hipLaunchKernelGGL(unpack_kernel, dim3(grid), dim3(kBlock), 0, stream, n, vals_b, seg_a, order_b);
		RP_TRY(hipGetLastError());
		bytes = temp_bytes;
		RP_TRY(hipcub::DeviceRadixSort::SortPairs(temp, bytes, seg_a, seg_b, order_b, order_a, n, 0, seg_bits, stream));
		if (level < levels) {
			hipLaunchKernelGGL(advance_kernel, dim3(grid), dim3(kBlock), 0, stream, r, n);
			RP_TRY(hipGetLastError());
		}
		len /= 2;
	}
	hipLaunchKernelGGL(ranks_kernel, dim3(grid), dim3(kBlock), 0, stream, r, n, order_a, d_neg, gate_mask, rank, counts);
	RP_TRY(hipGetLastError());
	size_t bytes = temp_bytes;
	RP_TRY(hipcub::DeviceScan::ExclusiveSum(temp, bytes, counts, firsts, n, stream));
	uint32_t last_first = 0, last_count = 0;
	RP_TRY(hipMemcpyAsync(&last_first, firsts + (nn - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
	RP_TRY(hipMemcpyAsync(&last_count, counts + (nn - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
	RP_TRY(hipStreamSynchronize(stream));
	const int64_t n_gates = (int64_t)last_first + last_count;
	Gate<double>* g64; Gate<float>* g32;
	RP_TRY(dev_alloc(&keep.ptrs, &g64, (size_t)n_gates)); RP_TRY(dev_alloc(&keep.ptrs, &g32, (size_t)n_gates));
	hipLaunchKernelGGL(gates_kernel, dim3(grid), dim3(kBlock), 0, stream, n, order_a, d_neg, gate_mask, gate_tmp, max_levels, counts, firsts,
	                   gate_first, gate_count, g64, g32);
	RP_TRY(hipGetLastError());
	RP_TRY(hipStreamSynchronize(stream));
	out->objs = objs; out->rank = rank; out->gate_first = gate_first; out->gate_count = gate_count;
	out->gates64 = g64; out->gates32 = g32; out->n_gates = n_gates; out->n = n;
	keep.ptrs.clear();
	return hipSuccess;
}

hipError_t gpu_check_objects(const rthip_object* dev_objs, int64_t n, int64_t* first_bad, int64_t* hittable, hipStream_t stream) {
	if (n < 1 || n > kMaxObjects) return hipErrorInvalidValue;
	Owned mem;
	unsigned long long* words;  // [0] hittable, [1] first bad index (as int32 in its low half)
	RP_TRY(dev_alloc(&mem.ptrs, &words, 2));
	const unsigned long long init[2] = {0ull, 0x7fffffffull};
	RP_TRY(hipMemcpyAsync(words, init, sizeof(init), hipMemcpyHostToDevice, stream));
	hipLaunchKernelGGL(check_objects_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, dev_objs, (int32_t)n,
	                   reinterpret_cast<int32_t*>(words + 1), words);
	RP_TRY(hipGetLastError());
	unsigned long long out[2];
	RP_TRY(hipMemcpyAsync(out, words, sizeof(out), hipMemcpyDeviceToHost, stream));
	RP_TRY(hipStreamSynchronize(stream));
	const int32_t bad = (int32_t)(uint32_t)out[1];
	*first_bad = bad == 0x7fffffff ? -1 : bad;
	*hittable = (int64_t)out[0];
	return hipSuccess;
}

hipError_t gpu_emit_prims(const GpuRefPass& ref, const int32_t* dev_ids, int32_t id_stride, int32_t count, Prim<float>* p32, Prim<double>* p64, Mat<float>* m32, Mat<double>* m64, hipStream_t stream) {
	if (count < 1) return hipSuccess;
	hipLaunchKernelGGL(emit_kernel, dim3((unsigned)((count + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, ref.objs, dev_ids, id_stride, count, ref.rank, ref.gate_first, ref.gate_count,
	                   p32, p64, m32, m64);
	return hipGetLastError();
}

hipError_t gpu_same_objects(const GpuRefPass& ref, const int32_t* dev_slot_ids, int32_t id_stride, int32_t count, int32_t* mismatches, hipStream_t stream) {
	*mismatches = 0;
	if (count < 1) return hipSuccess;  // a scene in which nothing can be hit
	Owned mem;
	int32_t* bad;
	RP_TRY(dev_alloc(&mem.ptrs, &bad, 1));
	RP_TRY(hipMemsetAsync(bad, 0, sizeof(int32_t), stream));
	hipLaunchKernelGGL(same_objects_kernel, dim3((unsigned)((count + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, ref.objs, ref.n, dev_slot_ids, id_stride, count, bad);
	RP_TRY(hipGetLastError());
	RP_TRY(hipMemcpyAsync(mismatches, bad, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
	return hipStreamSynchronize(stream);
}

}  // namespace rthip
