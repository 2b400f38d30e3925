// wavefront.hpp — the wavefront variant of the path tracer: the same per-path arithmetic as the
// megakernel (kernels.hpp), split by stage, with path state in HBM between stages.
//
//   wf_generate   camera rays of one batch of paths          Camera::get_ray, src/camera.rs:83-92
//   wf_extend     closest hit of every live ray              BVH::trace, src/bvh.rs:84-142
//   wf_shade      scatter / emission / sky, accumulation     src/renderer.rs:14-25, src/materials.rs:17-85
//
// One bounce = extend + shade.  Live paths are kept DENSE: shade writes the survivors of
// bounce b to the other half of a ping-pong state buffer at positions handed out by a
// ballot/prefix over the workgroup (one atomic per 1024 paths), so every stage reads and writes its state
// coalesced and a wave always starts a stage with 64 live rays, however divergent the
// paths' lengths are (depth 50 leaves long thin tails that this regrouping keeps packed).
// extend is itself persistent: a lane whose ray is done pulls the next ray of the wave's
// pool, so a wave is not held up by its slowest ray.  No host round trip inside a bounce:
// queue sizes live in a device counter block that the next stage reads.
//
// State per path (f32: 64 bytes): q0 = o.xyz, d.x | q1 = d.yz, T.xy | q2 = T.z, L.xyz |
// q3 = rng state, rng increment, accumulation index, depth left.  Hit: t, primitive index.
#pragma once
#include "kernels.hpp"

namespace rthip {

template <typename Real> struct alignas(sizeof(Real) * 4) Quad { Real x, y, z, w; };

template <typename Real> struct WfState {
	Quad<Real>*q0, *q1, *q2;
	uint4* q3;
	uint32_t* slot;  // sample slot of the radiance probe, or null
};

template <typename Real> struct WfArgs {
	const char* scene;
	uint32_t prims_off, mats_off, gates_off, scene_bytes, lds_bytes;
	int32_t stack_entries;
	CameraT<Real> cam;
	int32_t width, height, tiles_x, tiles_y;
	uint32_t spp_begin, spp_count;
	int32_t max_depth;
	uint32_t seed;
	int32_t sky_mode;
	int32_t part_index, part_count;
	uint32_t chunk, n_chunks;
	unsigned long long path_begin;  // first path of this batch in the launch's enumeration
	uint32_t path_count;            // paths of this batch (before dropping the invalid ones)
	WfState<Real> in, out;          // state read / written by this stage
	Real* hit_t;
	int32_t* hit_prim;
	uint32_t* n_in;    // live rays entering this stage
	uint32_t* n_out;   // survivors appended by this stage (generate, shade)
	uint32_t* fetch;   // extend: work counter, zero at launch
	unsigned long long* accum;
	Real* samples_out;
	unsigned long long* stats;  // [0] paths [1] rays
};

constexpr int kWfStreamThreads = 1024;  // generate and shade: one queue atomic per 1024 paths

// Appends one element per active thread of the workgroup to the queue behind `counter` and
// returns the thread's position.  Ballot + mbcnt rank inside a wave, a scan over the waves'
// counts in LDS, ONE atomic per workgroup: a single queue word sustains only ~90 atomics per
// microsecond on this chip, which a per-wave append of 8 M paths would exceed tenfold.
// Must be called by every thread of the workgroup; `scratch` holds kThreads / 64 words.
template <int kThreads>
__device__ __forceinline__ uint32_t block_append(uint32_t* counter, bool active, uint32_t* scratch) {
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	unsigned long long m = __ballot(active);
	if (lane == 0) scratch[wave] = (uint32_t)__popcll(m);
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t total = 0;
		for (int w = 0; w < kThreads / 64; w++) total += scratch[w];
		uint32_t base = total ? atomicAdd(counter, total) : 0u;
		for (int w = 0; w < kThreads / 64; w++) { uint32_t c = scratch[w]; scratch[w] = base; base += c; }
	}
	__syncthreads();
	uint32_t pos = scratch[wave] + lane_rank(m);
	__syncthreads();
	return pos;
}

template <typename Real>
__device__ __forceinline__ void wf_store(const WfState<Real>& s, uint32_t i, const RayT<Real>& r, V3<Real> T, V3<Real> L, const Rng& rng,
                                         uint32_t accum_index, int32_t depth, uint32_t slot) {
	s.q0[i] = Quad<Real>{r.o.x, r.o.y, r.o.z, r.d.x};
	s.q1[i] = Quad<Real>{r.d.y, r.d.z, T.x, T.y};
	s.q2[i] = Quad<Real>{T.z, L.x, L.y, L.z};
	s.q3[i] = make_uint4(rng.state, rng.inc, accum_index, (uint32_t)depth);
	if (s.slot) s.slot[i] = slot;
}

// ---- generate ----------------------------------------------------------------------------
template <typename Real>
__global__ __launch_bounds__(kWfStreamThreads) void wf_generate(WfArgs<Real> A) {
	__shared__ uint32_t scratch[kWfStreamThreads / 64];
	const uint32_t stride = gridDim.x * blockDim.x;
	uint32_t n_paths = 0;
	// every thread of a workgroup runs the same number of iterations (block_append is a workgroup operation)
	const uint32_t rounds = (A.path_count + stride - 1) / stride;
	for (uint32_t k = 0; k < rounds; k++) {
		uint32_t p = k * stride + blockIdx.x * blockDim.x + threadIdx.x;
		bool valid = p < A.path_count;
		PathId id{};
		if (valid) {
			unsigned long long g = A.path_begin + p;
			uint32_t per_item = kTilePixels * A.chunk;
			uint32_t item = (uint32_t)(g / per_item), j = (uint32_t)(g - (unsigned long long)item * per_item);
			valid = decode_path(item, j, A.n_chunks, A.chunk, A.spp_count, A.part_index, A.part_count, A.tiles_x, A.width, A.height, &id);
		}
		if (valid) n_paths++;
		bool emit = valid && A.max_depth > 0;  // renderer.rs:10-12: depth <= 0 is black, nothing to trace
		uint32_t pos = block_append<kWfStreamThreads>(A.n_out, emit, scratch);
		if (emit) {
			Rng rng;
			rng.init(A.seed, id.py * (uint32_t)A.width + id.px, A.spp_begin + id.s_local);
			RayT<Real> r = camera_ray<Real>(A.cam, (Real)id.px, (Real)id.py, rng, true);
			wf_store<Real>(A.out, pos, r, mk<Real>(1, 1, 1), mk<Real>(0, 0, 0), rng, id.accum_index, A.max_depth,
			               (id.s_local * (uint32_t)A.height + id.py) * (uint32_t)A.width + id.px);
		}
	}
	unsigned long long s = n_paths;
	for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
	if ((threadIdx.x & 63u) == 0 && s) atomicAdd(A.stats, s);
}

// ---- extend ------------------------------------------------------------------------------
template <typename Real, int kThreads, bool kLds>
__global__ __launch_bounds__(kThreads) void wf_extend(WfArgs<Real> A) {
	extern __shared__ __align__(64) char smem[];
	const uint32_t n = *A.n_in;
	constexpr uint32_t kPool = 256;  // rays a wave takes from the queue at a time
	constexpr uint32_t kRefill = 24; // idle lanes that trigger a refill
	// a late bounce has few rays left: workgroups beyond what they can occupy leave before
	// paying for the LDS copy of the scene
	if ((unsigned long long)blockIdx.x * kThreads >= n) return;

	const uint32_t lane = threadIdx.x & 63u;
	using Stack = StackLds<kThreads>;
	const Stack stack{reinterpret_cast<int*>(smem) + threadIdx.x};
	const uint32_t stack_bytes = (uint32_t)A.stack_entries * kThreads * 4u;
	const Node<Real>* nodes;
	const Prim<Real>* prims;
	const Gate<Real>* gates;
	if constexpr (kLds) {
		char* dst = smem + stack_bytes;
		const uint4* src = reinterpret_cast<const uint4*>(A.scene);
		for (uint32_t i = threadIdx.x; i < A.lds_bytes / 16u; i += kThreads) reinterpret_cast<uint4*>(dst)[i] = src[i];
		__syncthreads();
		nodes = reinterpret_cast<const Node<Real>*>(dst);
		prims = reinterpret_cast<const Prim<Real>*>(dst + A.prims_off);
		gates = reinterpret_cast<const Gate<Real>*>(dst + A.gates_off);
	} else {
		nodes = reinterpret_cast<const Node<Real>*>(A.scene);
		prims = reinterpret_cast<const Prim<Real>*>(A.scene + A.prims_off);
		gates = reinterpret_cast<const Gate<Real>*>(A.scene + A.gates_off);
	}
	bool busy = false;
	uint32_t index = 0;
	TravRay<Real> ray;
	Trav<Real> tv;
	Tally tally{0, 0, 0};
	Regions rg{};
	tv.cur = kDone; tv.sp = 0; tv.best_t = 0; tv.best_prim = -1;
	ray.o = ray.d = ray.inv = ray.noi = ray.qscale = mk<Real>(0, 0, 0); ray.eps = ray.rcp_a = 0;
	uint32_t pool_pos = 0, pool_end = 0;
	bool dry = false;

	for (;;) {
		// idle lanes take the next rays of the wave's pool -- in batches: a refill waits for its
		// rays to arrive from HBM and stalls the whole wave, so it is worth it only when enough
		// lanes are idle (or none is busy)
		for (;;) {
			unsigned long long need = __ballot(!busy);
			if (need == 0ull) break;
			if ((uint32_t)__popcll(need) < kRefill && need != ~0ull && !dry) { if (__ballot(busy) != 0ull) break; }
			if (pool_pos == pool_end) {
				if (dry) break;
				uint32_t b = 0;
				if (lane == 0) b = atomicAdd(A.fetch, kPool);
				b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
				if (b >= n) { dry = true; break; }
				pool_pos = b; pool_end = b + kPool < n ? b + kPool : n;
			}
			uint32_t want = (uint32_t)__popcll(need), avail = pool_end - pool_pos;
			uint32_t take = want < avail ? want : avail;
			if (!busy) {
				uint32_t rk = lane_rank(need);
				if (rk < take) {
					index = pool_pos + rk;
					Quad<Real> a = A.in.q0[index], b = A.in.q1[index];
					ray = prepare<Real>(RayT<Real>{mk<Real>(a.x, a.y, a.z), mk<Real>(a.w, b.x, b.y)});
					trav_begin<Real>(tv, stack.bottom());
					busy = true;
				}
			}
			pool_pos += take;
		}
		if (__ballot(busy) == 0ull) break;
		if (busy) {
			trav_inner<Real, Stack, false>(tv, ray, nodes, stack, tally, rg);
			if (tv.cur != kDone) trav_leaf<Real, Stack, false>(tv, ray, prims, gates, stack, tally, rg);
			if (tv.cur == kDone) {
				A.hit_t[index] = tv.best_t;
				A.hit_prim[index] = tv.best_prim;
				busy = false;
			}
		}
	}
}

// ---- shade -------------------------------------------------------------------------------
template <typename Real>
__global__ __launch_bounds__(kWfStreamThreads) void wf_shade(WfArgs<Real> A) {
	__shared__ uint32_t scratch[kWfStreamThreads / 64];
	const Prim<Real>* prims = reinterpret_cast<const Prim<Real>*>(A.scene + A.prims_off);
	const Mat<Real>* mats = reinterpret_cast<const Mat<Real>*>(A.scene + A.mats_off);
	const uint32_t n = *A.n_in;
	const uint32_t stride = gridDim.x * blockDim.x;
	const uint32_t rounds = (n + stride - 1) / stride;
	uint32_t n_rays = 0;
	for (uint32_t k = 0; k < rounds; k++) {
		uint32_t i = k * stride + blockIdx.x * blockDim.x + threadIdx.x;
		bool live = i < n, go_on = false;
		RayT<Real> next{};
		V3<Real> T{}, L{};
		Rng rng{0, 1};
		uint32_t accum_index = 0, slot = 0;
		int32_t depth = 0;
		if (live) {
			Quad<Real> a = A.in.q0[i], b = A.in.q1[i], c = A.in.q2[i];
			uint4 m = A.in.q3[i];
			TravRay<Real> ray;
			ray.o = mk<Real>(a.x, a.y, a.z);
			ray.d = mk<Real>(a.w, b.x, b.y);
			T = mk<Real>(b.z, b.w, c.x);
			L = mk<Real>(c.y, c.z, c.w);
			rng.state = m.x; rng.inc = m.y; accum_index = m.z; depth = (int32_t)m.w;
			if (A.in.slot) slot = A.in.slot[i];
			n_rays++;
			go_on = shade_ray<Real>(ray, A.hit_t[i], A.hit_prim[i], prims, mats, A.sky_mode, rng, T, L, depth, &next);
			if (!go_on) {
				if (A.samples_out) {
					Real* so = A.samples_out + (size_t)slot * 3;
					so[0] = L.x; so[1] = L.y; so[2] = L.z;
				}
				accumulate_sample<Real>(A.accum, accum_index, L);
			}
		}
		// regroup the survivors: dense positions in the other half of the state buffer
		uint32_t pos = block_append<kWfStreamThreads>(A.n_out, go_on, scratch);
		if (go_on) wf_store<Real>(A.out, pos, next, T, L, rng, accum_index, depth, slot);
	}
	unsigned long long s = n_rays;
	for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
	if ((threadIdx.x & 63u) == 0 && s) atomicAdd(A.stats + 1, s);
}

}  // namespace rthip
