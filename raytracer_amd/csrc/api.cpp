// api.cpp — the C ABI of librthip (include/rthip.h): context, uploads, launches, readback.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <exception>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/rthip.h"
#include "build_gpu.hpp"
#include "comm.hpp"
#include "host.hpp"
#include "launch.hpp"

using namespace rthip;

namespace {
thread_local std::string g_create_error;

struct SceneBlob {  // one precision of the scene in HBM: [nodes | prims | mats]
	char* dev = nullptr;
	uint32_t prims_off = 0, mats_off = 0, gates_off = 0, bytes = 0;
	uint32_t lds_bytes = 0;  // nodes + prims + gates: what an LDS-resident launch copies
	uint32_t lds_bytes_mats = 0;  // the same plus the materials, which follow them in the blob
	uint32_t qnodes_off = 0; // quantised nodes behind the materials (f32 blob of a large scene), 0 = none
	QFrame qframe{};
};
inline uint32_t align_up(uint64_t v, uint32_t a) { return (uint32_t)((v + a - 1) / a * a); }

// The entry points that work without a context build on the host alone.
BuildOptions host_only_options() {
	BuildOptions o = BuildOptions::from_env();
	o.tree_gpu = false;
	return o;
}

int env_int(const char* name, int dflt) {
	const char* v = std::getenv(name);
	return (v && *v) ? std::atoi(v) : dflt;
}

// Launch-time knobs.  Read from the RTHIP_* environment ONCE, when the context is created (a
// library under a threaded host must not call getenv per launch: it races with setenv, and the
// kernel's behaviour would follow the ambient environment); rthip_set_option changes them later.
struct Tuning {
	int scene_lds = -1;       // RTHIP_SCENE_LDS: -1 = by size, 0 = never, 1 = whenever it fits 100 KB
	int qnodes = 1;           // RTHIP_QNODES: traverse the quantised nodes of large f32 scenes
	int mats_lds = 1;         // RTHIP_MATS_LDS: keep the materials in LDS too when that costs no resident wave
	int stack_mem = 0;        // RTHIP_STACK_MEM: scenes in LDS keep the lanes' traversal stacks in global memory (f32 megakernel)
	int xchg = 0;             // RTHIP_XCHG: exchange queues (rays regrouped by shading arm), f32 megakernel with the scene in LDS
	int xchg_cap = 128;       // RTHIP_XCHG_CAP: records per queue (rounded down to a power of two, 64 at least)
	int xchg_pop_min = 16;    // RTHIP_XCHG_POP_MIN
	int xchg_push_max = 64;   // RTHIP_XCHG_PUSH_MAX
	int overlap = 1;          // RTHIP_OVERLAP: short launches on the context's own stream take turns on three streams, so that the tail of one runs under the start of the next
	int packet = 1;           // RTHIP_PACKET: the camera rays of a batch walk the tree together, as one packet per wave (f32 megakernel, scenes in global memory; 0 = each on its own)
	int pixel_major = -1;     // RTHIP_PIXEL_MAJOR: packets of camera rays cover all samples of a few neighbouring pixels (1) or one sample of a tile's 64 pixels (0); -1 = the former from 65 536 primitives on
	int root_leaf = 1;        // RTHIP_ROOT_LEAF: scenes with one leaf and no tree skip the traversal loop (0 = walk the root like any tree)
	int acc_lds = -1;         // RTHIP_ACC_LDS: per-wave accumulation tiles in LDS: -1 = scenes with nothing below the root, 0 = never, 1 = wherever they fit
	int threads = 0;          // RTHIP_THREADS: force the workgroup size (256, 512, 1024), 0 = most resident waves
	int blocks_per_cu = 0;    // RTHIP_BLOCKS_PER_CU: cap on resident workgroups per CU, 0 = none
	int grid_reserve = -1;    // RTHIP_GRID_RESERVE: workgroup slots of the chip a launch leaves empty (for kernels of other streams); -1 = none, or 1/16 of them with a multi-rank communicator
	int chunk = 0;            // RTHIP_CHUNK: samples of a tile per work item, 0 = by launch size
	int trav_threshold = -1;  // RTHIP_TRAV_THRESHOLD: -1 = 14 (scene in LDS) / 36 (global memory)
	int inner_exit = -1;      // RTHIP_INNER_EXIT: sixteenths, -1 = 4 (LDS) / 8 (global memory)
	int regen_min = 12;       // RTHIP_REGEN_MIN (12 since camera rays are made in batches: handing them out is cheap)
	int block_order = 32;     // RTHIP_BLOCK_ORDER
	int wf_batch = 64 << 20;  // RTHIP_WF_BATCH: paths per batch of the wavefront variant
	int count = 0;            // RTHIP_COUNT: instrumented launches (test tallies, region counters)
	std::string region_dump;  // RTHIP_REGION_DUMP: file that instrumented runs append region utilisation to
	static Tuning from_env() {
		Tuning t;
		t.scene_lds = env_int("RTHIP_SCENE_LDS", t.scene_lds);
		t.qnodes = env_int("RTHIP_QNODES", t.qnodes);
		t.mats_lds = env_int("RTHIP_MATS_LDS", t.mats_lds);
		t.acc_lds = env_int("RTHIP_ACC_LDS", t.acc_lds);
		t.root_leaf = env_int("RTHIP_ROOT_LEAF", t.root_leaf);
		t.packet = env_int("RTHIP_PACKET", t.packet);
		t.overlap = env_int("RTHIP_OVERLAP", t.overlap);
		t.pixel_major = env_int("RTHIP_PIXEL_MAJOR", t.pixel_major);
		t.stack_mem = env_int("RTHIP_STACK_MEM", t.stack_mem);
		t.xchg = env_int("RTHIP_XCHG", t.xchg);
		t.xchg_cap = env_int("RTHIP_XCHG_CAP", t.xchg_cap);
		t.xchg_pop_min = env_int("RTHIP_XCHG_POP_MIN", t.xchg_pop_min);
		t.xchg_push_max = env_int("RTHIP_XCHG_PUSH_MAX", t.xchg_push_max);
		t.threads = env_int("RTHIP_THREADS", t.threads);
		t.blocks_per_cu = env_int("RTHIP_BLOCKS_PER_CU", t.blocks_per_cu);
		t.grid_reserve = env_int("RTHIP_GRID_RESERVE", t.grid_reserve);
		t.chunk = env_int("RTHIP_CHUNK", t.chunk);
		t.trav_threshold = env_int("RTHIP_TRAV_THRESHOLD", t.trav_threshold);
		t.inner_exit = env_int("RTHIP_INNER_EXIT", t.inner_exit);
		t.regen_min = env_int("RTHIP_REGEN_MIN", t.regen_min);
		t.block_order = env_int("RTHIP_BLOCK_ORDER", t.block_order);
		t.wf_batch = env_int("RTHIP_WF_BATCH", t.wf_batch);
		t.count = env_int("RTHIP_COUNT", t.count);
		if (const char* v = std::getenv("RTHIP_REGION_DUMP")) t.region_dump = v;
		return t;
	}
};
}  // namespace

struct rthip_ctx {
	int device = 0;
	int num_cu = 0;
	hipStream_t own_stream = nullptr, stream = nullptr;
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	// Short launches (a viewer's 1-spp frames) end in a tail of a few long paths in nearly empty waves.  Samples
	// are integer adds into the accumulation buffer, so consecutive launches need no order among themselves:
	// on the context's own stream they take turns on that stream and two side streams, each with a control
	// block and events of its own, and the tail of one runs under the start of the next.  Every other call
	// first orders the own stream behind the side streams (join_lanes).
	struct Lane {
		hipStream_t stream = nullptr;
		char* control = nullptr;
		hipEvent_t ev0 = nullptr, ev1 = nullptr, done = nullptr;
		bool busy = false;
	};
	static constexpr int kSideLanes = 2;
	Lane side[kSideLanes];
	hipEvent_t fork = nullptr;          // recorded on the own stream before a launch goes to a side stream
	int lane_turn = 0;                  // whose turn the next short launch is: 0 = the own stream, k = side[k - 1]
	hipEvent_t last_ev0 = nullptr, last_ev1 = nullptr;   // the timing events of the last launch
	hipEvent_t noalloc_done = nullptr;  // recorded behind every launch of rthip_render_accumulate_noalloc
	std::string err;

	bool has_scene = false, has_camera = false;
	SceneBlob blob32, blob64;
	int tree_depth = 0, n_nodes = 0, n_leaves = 0, n_prims = 0, n_gates = 0;
	int64_t n_objects = 0;
	int32_t big_leaf = 0, start = 0;  // leaf of oversized primitives tested at ray start, and the tree's entry after it
	int32_t root_leaf = 0;            // the one leaf of a scene with nothing else below the root, 0 = the scene has a tree

	CameraFields cam{};
	int tiles_x = 0, tiles_y = 0;

	unsigned long long* own_accum = nullptr;
	uint64_t own_words = 0;
	unsigned long long* accum = nullptr;  // active buffer (own or caller's)
	uint64_t accum_words = 0;

	char* stack_mem = nullptr;  // traversal stacks of launches that keep them in global memory: tree depth x resident threads words
	size_t stack_mem_bytes = 0;
	char* d_control = nullptr;  // 1024 bytes: statistics words from +8, work-queue counters from +512 (one per 64-byte line)
	char* static_control = nullptr;  // the same block inside this device's static scratch array (rthip_render_accumulate_noalloc)
	char* last_control = nullptr;    // the block the last launch used: where its statistics are
	Tuning tune;
	BuildOptions build_opt;
	// resident workgroups per CU by (f64, wavefront, threads, scene in LDS, quantised, dynamic LDS bytes): asked of the runtime once
	std::map<std::tuple<bool, bool, int, int, bool, size_t>, int> occupancy;
	// wavefront variant: ping-pong path state, hit records and the per-bounce counter block
	char* wf_mem = nullptr;
	size_t wf_bytes = 0;
	void* readback = nullptr;  // device staging of the read-back calls
	size_t readback_bytes = 0;
	uint32_t* wf_counters = nullptr;  // kWfMaxDepth + 1 queue sizes, then kWfMaxDepth work counters
	rthip_stats stats{};
	bool launch_pending = false;
	bool last_in_lds = false;
	int last_threads = 0, last_grid = 0, last_lds_bytes = 0, last_waves_per_cu = 0;  // shape of the last megakernel launch
	ncclComm_t comm = nullptr;  // this context's rank in a communicator of GPUs (rthip_comm_init)
	int comm_rank = -1, comm_world = 0;
	// Reduces run on a stream of their own, so that the next frame's tracing (into another buffer)
	// overlaps them.  Each one in flight is remembered by the buffer it works on: whatever this
	// library does next to that buffer first makes the render stream wait for it.
	hipStream_t comm_stream = nullptr;
	struct PendingMerge { void* buf = nullptr; hipEvent_t ready = nullptr, done = nullptr; bool active = false; };
	PendingMerge merges[4];
};

namespace {

int32_t fail(rthip_ctx* c, int32_t code, const std::string& msg) noexcept {
	try {
		if (c) c->err = msg; else g_create_error = msg;
	} catch (...) {  // no memory for the message: the status still goes out
	}
	return code;
}
// Every extern "C" body runs inside this: nothing a C++ library routine throws (bad_alloc from a
// vector, a thread that cannot start) may unwind across the C ABI.
int32_t join_lanes(rthip_ctx* c);
template <typename F>
int32_t guarded(rthip_ctx* c, F&& body, bool join = true) noexcept {
	try {
		if (c && join) {  // what the context spread over its side streams comes first (see rthip_ctx::Lane)
			const int32_t rc = join_lanes(c);
			if (rc != RTHIP_OK) return rc;
		}
		return body();
	} catch (const std::bad_alloc&) {
		return fail(c, RTHIP_ERR_LIMIT, "out of host memory");
	} catch (const std::exception& e) {
		return fail(c, RTHIP_ERR_INTERNAL, e.what());
	} catch (...) {
		return fail(c, RTHIP_ERR_INTERNAL, "unknown exception");
	}
}
#define HIP_TRY(ctx, expr)                                                                                     \
	do {                                                                                                       \
		hipError_t e_ = (expr);                                                                                \
		if (e_ != hipSuccess) return fail((ctx), RTHIP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
	} while (0)

template <typename Real> void fill_camera(const CameraFields& f, CameraT<Real>* c) {
	for (int k = 0; k < 3; k++) {
		c->origin[k] = (Real)f.origin[k]; c->pixel00[k] = (Real)f.pixel00[k];
		c->delta_x[k] = (Real)f.delta_x[k]; c->delta_y[k] = (Real)f.delta_y[k];
		c->disk_x[k] = (Real)f.disk_x[k]; c->disk_y[k] = (Real)f.disk_y[k];
	}
	c->defocus_angle = (Real)f.defocus_angle;
}

template <typename Real>
int32_t upload_blob(rthip_ctx* c, const std::vector<Node<Real>>& nodes, const std::vector<Prim<Real>>& prims,
                    const std::vector<Mat<Real>>& mats, const std::vector<Gate<Real>>& gates, SceneBlob* b,
                    const std::vector<QNode>* qnodes = nullptr, const QFrame* qframe = nullptr, bool records_follow = false) {
	uint64_t nb = nodes.size() * sizeof(Node<Real>), pb = prims.size() * sizeof(Prim<Real>), mb = mats.size() * sizeof(Mat<Real>);
	uint64_t gb = gates.size() * sizeof(Gate<Real>);
	uint64_t prims_off = align_up(nb, 64), gates_off = align_up(prims_off + pb, 64), mats_off = align_up(gates_off + gb, 64);
	uint64_t qb = (qnodes && !qnodes->empty()) ? qnodes->size() * sizeof(QNode) : 0;
	uint64_t q_off = align_up(mats_off + mb, 64);
	uint64_t total = align_up(q_off + qb, 64);
	if (total > 0xFFFFFFF0ull) return fail(c, RTHIP_ERR_LIMIT, "scene larger than 4 GiB");
	if (b->dev) { HIP_TRY(c, hipFree(b->dev)); b->dev = nullptr; }
	b->qnodes_off = qb ? (uint32_t)q_off : 0u;
	if (qb) b->qframe = *qframe;
	HIP_TRY(c, hipMalloc(&b->dev, (size_t)total));
	// each array straight to its place (a staging copy of a million-object scene is a quarter of a gigabyte);
	// the padding between the arrays is zeroed with the blob
	HIP_TRY(c, hipMemset(b->dev, 0, (size_t)total));
	if (nb) HIP_TRY(c, hipMemcpy(b->dev, nodes.data(), (size_t)nb, hipMemcpyHostToDevice));
	if (!records_follow) {  // (else the device writes primitives, materials and gates itself: only their room is needed)
		if (pb) HIP_TRY(c, hipMemcpy(b->dev + prims_off, prims.data(), (size_t)pb, hipMemcpyHostToDevice));
		if (mb) HIP_TRY(c, hipMemcpy(b->dev + mats_off, mats.data(), (size_t)mb, hipMemcpyHostToDevice));
		if (gb) HIP_TRY(c, hipMemcpy(b->dev + gates_off, gates.data(), (size_t)gb, hipMemcpyHostToDevice));
	}
	if (qb) HIP_TRY(c, hipMemcpy(b->dev + q_off, qnodes->data(), (size_t)qb, hipMemcpyHostToDevice));
	b->prims_off = (uint32_t)prims_off; b->mats_off = (uint32_t)mats_off; b->gates_off = (uint32_t)gates_off; b->bytes = (uint32_t)total;
	b->lds_bytes = (uint32_t)mats_off;
	b->lds_bytes_mats = align_up(mats_off + mb, 64);
	return RTHIP_OK;
}


// Refit: everything behind the nodes of a blob (primitives, gates, materials) replaced, the nodes
// kept.  In place when the new arrays lie where the old ones did (the number of gate boxes can
// change when objects move); otherwise into a new allocation that the nodes are copied into.
template <typename Real>
int32_t replace_blob_tail(rthip_ctx* c, size_t n_nodes, const std::vector<Prim<Real>>& prims, const std::vector<Mat<Real>>& mats,
                          const std::vector<Gate<Real>>& gates, SceneBlob* b) {
	const uint64_t nb = n_nodes * sizeof(Node<Real>), pb = prims.size() * sizeof(Prim<Real>), mb = mats.size() * sizeof(Mat<Real>);
	const uint64_t gb = gates.size() * sizeof(Gate<Real>);
	const uint64_t prims_off = align_up(nb, 64), gates_off = align_up(prims_off + pb, 64), mats_off = align_up(gates_off + gb, 64);
	const uint64_t qb = b->qnodes_off ? n_nodes * sizeof(QNode) : 0;
	const uint64_t q_off = align_up(mats_off + mb, 64);
	const uint64_t total = align_up(q_off + qb, 64);
	if (total > 0xFFFFFFF0ull) return fail(c, RTHIP_ERR_LIMIT, "scene larger than 4 GiB");
	if (prims_off != b->prims_off) return fail(c, RTHIP_ERR_INTERNAL, "refit: the primitives moved");
	std::vector<char> host((size_t)(q_off - prims_off), 0);
	if (pb) std::memcpy(host.data(), prims.data(), (size_t)pb);  // (a scene in which nothing can be hit has no records: no null pointers into memcpy)
	if (gb) std::memcpy(host.data() + (gates_off - prims_off), gates.data(), (size_t)gb);
	if (mb) std::memcpy(host.data() + (mats_off - prims_off), mats.data(), (size_t)mb);
	const bool same_place = gates_off == b->gates_off && mats_off == b->mats_off && total == b->bytes && (!qb || q_off == b->qnodes_off);
	if (!same_place) {
		char* fresh = nullptr;
		HIP_TRY(c, hipMalloc(&fresh, (size_t)total));
		hipError_t e = hipMemcpy(fresh, b->dev, (size_t)nb, hipMemcpyDeviceToDevice);
		if (e == hipSuccess && qb) e = hipMemset(fresh + q_off, 0, (size_t)qb);
		if (e != hipSuccess) { (void)hipFree(fresh); HIP_TRY(c, e); }
		HIP_TRY(c, hipFree(b->dev));
		b->dev = fresh;
		b->gates_off = (uint32_t)gates_off; b->mats_off = (uint32_t)mats_off; b->bytes = (uint32_t)total;
		if (qb) b->qnodes_off = (uint32_t)q_off;
		b->lds_bytes = (uint32_t)mats_off;
		b->lds_bytes_mats = align_up(mats_off + mb, 64);
	}
	if (!host.empty()) HIP_TRY(c, hipMemcpy(b->dev + prims_off, host.data(), host.size(), hipMemcpyHostToDevice));
	return RTHIP_OK;
}

// A blob of the given shape, allocated and zeroed, nothing in it yet (the device makes the contents).
template <typename Real>
int32_t alloc_blob(rthip_ctx* c, size_t n_nodes, size_t n_prims, size_t n_gates, bool with_qnodes, SceneBlob* b) {
	const uint64_t nb = n_nodes * sizeof(Node<Real>), pb = n_prims * sizeof(Prim<Real>), mb = n_prims * sizeof(Mat<Real>), gb = n_gates * sizeof(Gate<Real>);
	const uint64_t prims_off = align_up(nb, 64), gates_off = align_up(prims_off + pb, 64), mats_off = align_up(gates_off + gb, 64);
	const uint64_t qb = with_qnodes ? n_nodes * sizeof(QNode) : 0;
	const uint64_t q_off = align_up(mats_off + mb, 64);
	const uint64_t total = align_up(q_off + qb, 64);
	if (total > 0xFFFFFFF0ull) return fail(c, RTHIP_ERR_LIMIT, "scene larger than 4 GiB");
	if (b->dev) { HIP_TRY(c, hipFree(b->dev)); b->dev = nullptr; }
	HIP_TRY(c, hipMalloc(&b->dev, (size_t)total));
	HIP_TRY(c, hipMemset(b->dev, 0, (size_t)total));
	b->prims_off = (uint32_t)prims_off; b->mats_off = (uint32_t)mats_off; b->gates_off = (uint32_t)gates_off; b->bytes = (uint32_t)total;
	b->qnodes_off = qb ? (uint32_t)q_off : 0u;
	b->lds_bytes = (uint32_t)mats_off;
	b->lds_bytes_mats = align_up(mats_off + mb, 64);
	return RTHIP_OK;
}
// Refit on the device: room for another number of gate boxes.  Nodes and primitives are copied
// across (the primitives still name the objects in their places); gates and materials are rewritten anyway.
template <typename Real>
int32_t regate_blob(rthip_ctx* c, size_t n_nodes, size_t n_prims, size_t n_gates, SceneBlob* b) {
	const uint64_t pb = n_prims * sizeof(Prim<Real>), mb = n_prims * sizeof(Mat<Real>), gb = n_gates * sizeof(Gate<Real>);
	const uint64_t gates_off = b->gates_off, mats_off = align_up(gates_off + gb, 64);
	if (mats_off == b->mats_off) return RTHIP_OK;
	const uint64_t qb = b->qnodes_off ? n_nodes * sizeof(QNode) : 0;
	const uint64_t q_off = align_up(mats_off + mb, 64);
	const uint64_t total = align_up(q_off + qb, 64);
	if (total > 0xFFFFFFF0ull) return fail(c, RTHIP_ERR_LIMIT, "scene larger than 4 GiB");
	char* fresh = nullptr;
	HIP_TRY(c, hipMalloc(&fresh, (size_t)total));
	hipError_t e = hipMemset(fresh, 0, (size_t)total);
	if (e == hipSuccess) e = hipMemcpy(fresh, b->dev, (size_t)(b->prims_off + pb), hipMemcpyDeviceToDevice);
	if (e != hipSuccess) { (void)hipFree(fresh); HIP_TRY(c, e); }
	HIP_TRY(c, hipFree(b->dev));
	b->dev = fresh;
	b->mats_off = (uint32_t)mats_off; b->bytes = (uint32_t)total;
	if (qb) b->qnodes_off = (uint32_t)q_off;
	b->lds_bytes = (uint32_t)mats_off;
	b->lds_bytes_mats = align_up(mats_off + mb, 64);
	return RTHIP_OK;
}
// Callers may hand over objects that live in device memory (a simulation on the GPU, a torch tensor).
bool in_device_memory(const void* p) {
	hipPointerAttribute_t attr{};
	if (hipPointerGetAttributes(&attr, p) != hipSuccess) { (void)hipGetLastError(); return false; }  // plain host memory the runtime knows nothing of
	return attr.type == hipMemoryTypeDevice;
}
// Whether ranks, gates and primitive records of a scene of n objects are made on the device (see BuildOptions::ref_gpu).
bool reference_pass_on_device(const BuildOptions& o, int64_t n) {
	if (o.tie_reverse) return false;  // the other tie order is the host pass's knob
	return o.ref_gpu > 0 || (o.ref_gpu < 0 && n >= 2048);
}
struct RefPassHolder {  // frees the device arrays of a reference pass on every way out
	GpuRefPass ref;
	~RefPassHolder() { ref.release(); }
};

// Scenes up to this size are copied into LDS by every workgroup of the render kernel.
constexpr uint32_t kLdsSceneLimit = 64 * 1024;
constexpr int kWfMaxDepth = 4096;
constexpr size_t kControlBytes = 1024;  // control block of a launch: statistics from +8, work-queue counters from +512

struct Plan {
	LaunchCfg cfg;
	uint32_t chunk, n_chunks, n_items;
	int32_t threshold;
	int waves_per_cu;             // resident waves per CU the launch is planned for
	bool short_launch;            // fewer paths than 512 per resident wave (a viewer's 1-spp frame)
	bool overlap_ok;              // small enough for its tail to be worth hiding under the next launch
	uint32_t xchg_off, xchg_cap;  // exchange queues: where they start in the workgroup's LDS, records per queue
};

int32_t validate(rthip_ctx* c, const rthip_render_params* p) {
	if (!c) return RTHIP_ERR_INVALID;
	if (!p) return fail(c, RTHIP_ERR_INVALID, "params is NULL");
	if (!c->has_scene || !c->has_camera) return fail(c, RTHIP_ERR_STATE, "upload a scene and set a camera first");
	if (p->precision != RTHIP_F32 && p->precision != RTHIP_F64) return fail(c, RTHIP_ERR_INVALID, "precision must be RTHIP_F32 or RTHIP_F64");
	if (p->sky_mode != RTHIP_SKY_REFERENCE && p->sky_mode != RTHIP_SKY_WALLPAPER) return fail(c, RTHIP_ERR_INVALID, "unknown sky_mode");
	if (p->variant != RTHIP_MEGAKERNEL && p->variant != RTHIP_WAVEFRONT) return fail(c, RTHIP_ERR_INVALID, "unknown variant");
	if (p->variant == RTHIP_WAVEFRONT && p->max_depth > kWfMaxDepth) return fail(c, RTHIP_ERR_LIMIT, "the wavefront variant supports max_depth <= 4096");
	if (p->part_count < 1 || p->part_index < 0 || p->part_index >= p->part_count) return fail(c, RTHIP_ERR_INVALID, "bad tile partition");
	if (p->reserved[0] || p->reserved[1] || p->reserved[2]) return fail(c, RTHIP_ERR_INVALID, "reserved fields must be zero");
	if ((uint64_t)p->spp_begin + p->spp_count > 0xFFFFFFFFull) return fail(c, RTHIP_ERR_INVALID, "sample range overflows");
	return RTHIP_OK;
}

// Resident workgroups per CU, asked of the runtime once per kernel variant and LDS size.
int32_t blocks_per_cu(rthip_ctx* c, bool f64, bool wavefront, int threads, int lds_mode, bool quantised, size_t lds, int* out, bool acc_lds = false, bool stack_mem = false, bool xchg = false,
                      bool root_only = false) {
	const auto key = std::make_tuple(f64, wavefront, threads, lds_mode + (acc_lds ? 8 : 0) + (stack_mem ? 16 : 0) + (xchg ? 32 : 0) + (root_only ? 64 : 0), quantised, lds);
	auto it = c->occupancy.find(key);
	if (it != c->occupancy.end()) { *out = it->second; return RTHIP_OK; }
	int b = 0;
	if (wavefront) HIP_TRY(c, f64 ? wf_extend_blocks_per_cu_f64(threads, lds_mode != 0, lds, &b) : wf_extend_blocks_per_cu_f32(threads, lds_mode != 0, lds, &b));
	else HIP_TRY(c, f64 ? render_blocks_per_cu_f64(threads, lds_mode, lds, &b) : render_blocks_per_cu_f32(threads, lds_mode, lds, &b, quantised, acc_lds, stack_mem, xchg, root_only));
	c->occupancy[key] = b;
	*out = b;
	return RTHIP_OK;
}

int32_t make_plan(rthip_ctx* c, const rthip_render_params* p, bool count, Plan* plan, bool wavefront = false, bool frugal = false) {
	const bool f64 = p->precision == RTHIP_F64;
	const SceneBlob& blob = f64 ? c->blob64 : c->blob32;
	const Tuning& tune = c->tune;
	bool in_lds = tune.scene_lds < 0 ? blob.lds_bytes <= kLdsSceneLimit : (tune.scene_lds != 0 && blob.lds_bytes <= 100 * 1024);
	plan->cfg.device = c->device;
	plan->cfg.scene_in_lds = in_lds;
	plan->cfg.count = count;
	plan->cfg.quantised = !in_lds && !f64 && !wavefront && blob.qnodes_off != 0 && tune.qnodes != 0;
	// Workgroup size and LDS contents: the kernel is latency bound, so pick what keeps the most
	// waves resident.  Every workgroup holds its own LDS copy of the scene, so larger workgroups
	// amortise it; among equals, the one that has room for the materials too wins (tried first).
	const int forced = tune.threads;
	uint64_t n_tiles = (uint64_t)c->tiles_x * c->tiles_y;
	uint64_t my_tiles = n_tiles > (uint64_t)p->part_index ? (n_tiles - p->part_index + p->part_count - 1) / p->part_count : 0;
	// Short launches (a viewer's 1-spp frame at 1080p: ~340 paths per resident wave) are all start-up
	// and tail: small workgroups start sooner and finish closer together, which is worth more there than
	// resident waves (0.38 ms against 0.46 ms for that frame).
	plan->short_launch = false;
	plan->overlap_ok = false;
	const bool short_launch = in_lds && !wavefront && my_tiles * kTilePixels * (uint64_t)p->spp_count < 512ull * 24 * (uint64_t)c->num_cu;
	plan->short_launch = short_launch;
	// (launches up to four times that size still gain from overlapping their tails with the next launch's start:
	//  2- and 4-spp 1080p frames -23 % and -13 %, a 1-spp 4K frame -16 %; 12-spp frames lose 5 %)
	plan->overlap_ok = in_lds && !wavefront && my_tiles * kTilePixels * (uint64_t)p->spp_count < 2048ull * 24 * (uint64_t)c->num_cu;
	// Stacks in global memory: f32 megakernel with the scene in LDS (what the freed LDS is for is decided below).
	const bool stack_mem = tune.stack_mem > 0 && (in_lds || plan->cfg.quantised) && !f64 && !wavefront && !frugal;  // (the entry point that allocates nothing keeps its stacks in LDS)
	plan->cfg.stack_mem = stack_mem;
	// Exchange queues: the same launches, long ones only (the queues pay once they have filled, and a
	// launch's last rays leave them one wave at a time).
	uint32_t xcap = 64;
	while (xcap * 2 <= (uint32_t)std::max(64, std::min(4096, tune.xchg_cap))) xcap *= 2;
	const bool xchg = tune.xchg > 0 && in_lds && !f64 && !wavefront && !short_launch;
	plan->cfg.xchg = xchg;
	plan->xchg_cap = xchg ? xcap : 0;
	plan->xchg_off = 0;
	int best_waves = -1, bpc = 0;
	for (int lds_mode : {2, 1, 0}) {
		if ((lds_mode != 0) != in_lds) continue;
		if (lds_mode == 2 && (wavefront || tune.mats_lds == 0 || blob.lds_bytes_mats > kLdsSceneLimit)) continue;
		for (int threads : {256, 512, 768, 1024}) {
			if (forced && threads != forced) continue;
			if (short_launch && !forced && threads != 256) continue;
			if (threads == 1024 && (wavefront || forced != 1024 || (lds_mode != 0 && stack_mem) || lds_mode == 2)) continue;  // 32 waves per CU at 64 VGPRs measured no faster than 24 at 80: on request only
			if (threads == 768 && lds_mode != 2) continue;  // two workgroups of twelve waves: only built for the variant that needs the room
			size_t lds = (stack_mem ? 0 : (size_t)c->tree_depth * threads * 4) + (lds_mode == 2 ? blob.lds_bytes_mats : lds_mode == 1 ? blob.lds_bytes : 0);
			const size_t xoff = align_up(lds, 64);
			if (xchg) lds = xoff + xchg_bytes(xcap);
			int b = 0;
			int32_t rc = blocks_per_cu(c, f64, wavefront, threads, lds_mode, plan->cfg.quantised, lds, &b, false, stack_mem, xchg);
			if (rc != RTHIP_OK) return rc;
			if (b * threads / 64 > best_waves) {
				best_waves = b * threads / 64; bpc = b;
				plan->cfg.threads = threads; plan->cfg.dyn_lds = lds; plan->cfg.lds_mode = lds_mode;
				plan->xchg_off = (uint32_t)xoff;
			}
		}
	}
	if (bpc < 1) return fail(c, RTHIP_ERR_LIMIT, "render kernel does not fit on a CU");
	// Scenes with nothing below the root (a few objects under an open sky: paths of 1.3 rays, two
	// node visits each) are bound by the three accumulation atomics per path, not by arithmetic:
	// there the waves sum their samples in LDS tiles and write a tile once per work item
	// (kAccLds in kernels.hpp).  Recognised by the tree alone, so that a launch does not depend on
	// the launches before it; only where the tiles cost no resident wave, and not for launches of a
	// few samples, whose items are one wave's worth of paths anyway.
	plan->cfg.acc_lds = false;
	plan->cfg.root_only = false;
	plan->cfg.acc_tile_off = 0;
	const bool one_leaf = c->root_leaf != 0 && tune.root_leaf != 0;   // the launch tests that leaf at ray start and walks nothing: there is a kernel without the walk
	const bool acc_wanted = tune.acc_lds > 0 || (tune.acc_lds < 0 && c->n_nodes <= 1);
	if (acc_wanted && !f64 && !wavefront && !count && !stack_mem && !xchg && plan->cfg.lds_mode == 2 && p->spp_count >= 8) {
		for (int threads : {256, 512}) {
			if (forced && threads != forced) continue;
			const size_t base = (size_t)c->tree_depth * threads * 4 + blob.lds_bytes_mats;
			const size_t lds = base + (size_t)(threads / 64) * kTilePixels * kAccumChannels * 8;
			int b = 0;
			int32_t rc = blocks_per_cu(c, f64, wavefront, threads, 2, false, lds, &b, true, false, false, one_leaf);
			if (rc != RTHIP_OK) return rc;
			if (b * threads / 64 >= best_waves && (!plan->cfg.acc_lds || b * threads / 64 > best_waves)) {
				best_waves = b * threads / 64; bpc = b;
				plan->cfg.threads = threads; plan->cfg.dyn_lds = lds; plan->cfg.acc_lds = true; plan->cfg.acc_tile_off = (uint32_t)base;
				plan->cfg.root_only = one_leaf;
			}
		}
	}
	const int cap = tune.blocks_per_cu;
	if (cap > 0) bpc = std::min(bpc, cap);
	plan->waves_per_cu = bpc * plan->cfg.threads / 64;
	// Samples of a tile per work item.  Small items balance short launches (the tail is one item per
	// wave); long launches want fewer, larger items (millions of items contend on the queue counters
	// and scatter a tile's accumulation lines).  Measured optimum: 4 up to ~128 spp, 8-16 beyond.
	uint32_t auto_chunk = 4;
	while (auto_chunk < 16 && auto_chunk * 64 <= p->spp_count) auto_chunk *= 2;
	// ... and items of equal size: a launch's samples rarely divide by that figure, and the items of the
	// last, short chunk of every tile cost nearly what full ones do (25 spp as 6 x 4 + 1: 7 % slower than
	// 5 x 5; 13 spp as 3 x 4 + 1 against 5 + 5 + 3: 7 %).  Among the sizes from that figure up to just
	// under its double, the one that leaves the fullest last chunk, the larger of equals.
	if (p->spp_count > auto_chunk) {
		uint32_t best = auto_chunk;
		double best_fill = 0.0;
		for (uint32_t cand = auto_chunk; cand < 2 * auto_chunk; cand++) {
			const uint32_t n = (p->spp_count + cand - 1) / cand;
			const double fill = (double)p->spp_count / ((double)n * cand);
			if (fill >= best_fill) { best_fill = fill; best = cand; }
		}
		auto_chunk = best;
	}
	// Very short paths (a few objects under an open sky: 1.3 rays per path, two node visits each)
	// spend a third of their time on accumulation atomics, and those run faster when a tile's
	// samples follow one another: 32 samples per item gave 52 instead of 38 Gpaths/s on the
	// three-object scene at any spp.  Recognised by the tree alone (nothing below the root), so
	// that a launch does not depend on the launches before it.
	if (c->n_nodes <= 1) auto_chunk = 32;
	// With LDS tiles an item costs one flush and one drain of the wave: as many samples per item as
	// divide the launch evenly, up to 64.
	if (plan->cfg.acc_lds) auto_chunk = (p->spp_count + (p->spp_count + 63) / 64 - 1) / ((p->spp_count + 63) / 64);
	uint32_t chunk = (uint32_t)std::max(1, tune.chunk > 0 ? tune.chunk : (int)auto_chunk);
	chunk = std::min<uint32_t>(chunk, std::max<uint32_t>(1u, p->spp_count));
	uint32_t n_chunks = (p->spp_count + chunk - 1) / chunk;
	uint64_t n_items = my_tiles * n_chunks;
	if (n_items > 0xFFFFFF00ull) return fail(c, RTHIP_ERR_LIMIT, "too many work items in one call: lower spp_count");
	plan->chunk = chunk; plan->n_chunks = n_chunks; plan->n_items = (uint32_t)n_items;
	uint64_t waves_per_block = (uint64_t)plan->cfg.threads / 64;
	// Short launches (a viewer's 1-spp frame: ~340 paths per resident wave) are all start-up and
	// tail; two thirds of the resident workgroups finish them 10 % sooner.
	const uint64_t my_paths = n_items * kTilePixels * chunk;
	if (cap <= 0 && bpc >= 3 && my_paths < 512ull * (uint64_t)bpc * c->num_cu * waves_per_block) bpc = bpc * 2 / 3;
	uint64_t want_blocks = (n_items + waves_per_block - 1) / waves_per_block;
	// A launch that fills every workgroup slot of the chip keeps the kernels of other streams out until
	// its tail (measured: a 50 MB add on a second stream finished 1.99 ms after the start of a 1.9 ms
	// render, and 0.35 ms after it with 32 slots of 12 waves left empty).  The reduce that merges the
	// GPUs' buffers is such a kernel and is meant to run under the next frame's tracing, so a context
	// with a communicator of more than one rank leaves 1/16 of the chip's slots empty.
	const uint64_t slots = (uint64_t)bpc * c->num_cu;
	const uint64_t reserve = tune.grid_reserve >= 0 ? (uint64_t)tune.grid_reserve : (c->comm && c->comm_world > 1 ? slots / 16 : 0);
	plan->cfg.grid = (int)std::max<uint64_t>(1, std::min<uint64_t>(slots > reserve ? slots - reserve : 1, want_blocks));
	// Leave the traversal loop for shading / regeneration once this few lanes are still in it.
	// Measured optimum: 12-16 with the scene in LDS, about 36 when every node is a trip to L1 or L2.
	plan->threshold = tune.trav_threshold >= 0 ? tune.trav_threshold : (in_lds ? 14 : 36);
	return RTHIP_OK;
}

// Orders the context's own stream behind whatever its side streams still run (a stream wait, no host wait).
int32_t join_lanes(rthip_ctx* c) {
	for (auto& l : c->side) {
		if (!l.busy) continue;
		HIP_TRY(c, hipSetDevice(c->device));
		HIP_TRY(c, hipStreamWaitEvent(c->own_stream, l.done, 0));
		l.busy = false;
	}
	c->lane_turn = 0;
	return RTHIP_OK;
}
// A side lane's stream, control block and events, made when first used.
int32_t ensure_lane(rthip_ctx* c, rthip_ctx::Lane* l) {
	if (!c->fork) HIP_TRY(c, hipEventCreateWithFlags(&c->fork, hipEventDisableTiming));
	if (!l->stream) HIP_TRY(c, hipStreamCreateWithFlags(&l->stream, hipStreamNonBlocking));
	if (!l->ev0) HIP_TRY(c, hipEventCreate(&l->ev0));
	if (!l->ev1) HIP_TRY(c, hipEventCreate(&l->ev1));
	if (!l->done) HIP_TRY(c, hipEventCreateWithFlags(&l->done, hipEventDisableTiming));
	if (!l->control) HIP_TRY(c, hipMalloc(&l->control, kControlBytes));
	return RTHIP_OK;
}

// `frugal` (rthip_render_accumulate_noalloc): nothing may be allocated, and the stream may be under
// graph capture, where events cannot be queried afterwards: `timed` = false then leaves out the
// timing events and the statistics of the launch.
template <typename Real>
int32_t enqueue_render(rthip_ctx* c, const rthip_render_params* p, bool count, unsigned long long* accum, Real* samples_out, char* control = nullptr,
                       bool frugal = false, bool timed = true, bool may_overlap = false) {
	may_overlap = may_overlap && !control && !frugal && timed && !samples_out && c->stream == c->own_stream;
	if (!control) control = c->d_control;
	if (samples_out && std::is_same<Real, float>::value) count = true;   // only the instrumented f32 kernels write per-sample radiance
	Plan plan;
	int32_t rc = make_plan(c, p, count, &plan, false, frugal);
	if (rc != RTHIP_OK) return rc;
	const SceneBlob& blob = std::is_same<Real, double>::value ? c->blob64 : c->blob32;
	RenderArgs<Real> a;
	std::memset(&a, 0, sizeof(a));
	a.scene = blob.dev; a.prims_off = blob.prims_off; a.mats_off = blob.mats_off; a.gates_off = blob.gates_off; a.scene_bytes = blob.bytes;
	a.lds_bytes = plan.cfg.lds_mode == 2 ? blob.lds_bytes_mats : blob.lds_bytes;
	a.acc_tile_off = plan.cfg.acc_tile_off;
	a.qnodes_off = blob.qnodes_off; a.qframe = blob.qframe;
	a.big_leaf = c->big_leaf; a.start = c->start;
	if (c->root_leaf != 0 && c->tune.root_leaf != 0) { a.big_leaf = c->root_leaf; a.start = kDone; }  // no tree: every ray tests the one leaf and is done
	a.stack_entries = c->tree_depth;
	fill_camera<Real>(c->cam, &a.cam);
	a.width = c->cam.width; a.height = c->cam.height; a.tiles_x = c->tiles_x; a.tiles_y = c->tiles_y;
	a.spp_begin = p->spp_begin; a.spp_count = p->spp_count; a.max_depth = p->max_depth; a.seed = p->seed;
	a.sky_mode = p->sky_mode; a.part_index = p->part_index; a.part_count = p->part_count;
	a.n_items = plan.n_items; a.chunk = plan.chunk; a.n_chunks = plan.n_chunks; a.trav_threshold = plan.threshold;
	// sixteenths of the lanes that entered a round; beyond 16 no lane would ever take a step
	a.inner_exit = std::max(0, std::min(16, c->tune.inner_exit >= 0 ? c->tune.inner_exit : (plan.cfg.scene_in_lds ? 4 : 8)));
	a.regen_min = std::max(1, std::min(64, c->tune.regen_min));
	a.block_order = (uint32_t)std::max(0, std::min(1024, c->tune.block_order));
	a.accum = accum; a.samples_out = samples_out;
	a.queue = reinterpret_cast<uint32_t*>(control + 512);
	a.stats = reinterpret_cast<unsigned long long*>(control + 8);
	a.xchg_off = plan.xchg_off; a.xchg_cap = plan.xchg_cap;
	a.packet = (c->tune.packet != 0 && c->tree_depth <= 64) ? 1 : 0;   // the wave's stack holds an entry per lane
	// Narrow packets pay where the walk is long (large scenes); the samples of a pixel then finish at about the
	// same time and their accumulation adds meet at one address, which costs more than it gains on small scenes.
	a.pixel_major = c->tune.pixel_major >= 0 ? (c->tune.pixel_major != 0) : (c->n_prims >= 65536);
	a.xchg_pop_min = std::max(1, std::min(64, c->tune.xchg_pop_min));
	a.xchg_push_max = std::max(0, std::min(64, c->tune.xchg_push_max));
	if (plan.cfg.stack_mem) {
		const size_t need = (size_t)std::max(1, c->tree_depth) * (size_t)plan.cfg.grid * (size_t)plan.cfg.threads * 4;
		if (need > 0xFFFFFFFFull) return fail(c, RTHIP_ERR_LIMIT, "traversal stacks in global memory exceed 4 GiB");
		if (c->stack_mem_bytes < need) {
			HIP_TRY(c, hipStreamSynchronize(c->stream));  // an earlier launch may still be using the old array
			if (c->stack_mem) { HIP_TRY(c, hipFree(c->stack_mem)); c->stack_mem = nullptr; c->stack_mem_bytes = 0; }
			HIP_TRY(c, hipMalloc(&c->stack_mem, need));
			c->stack_mem_bytes = need;
		}
		a.stack_mem = c->stack_mem;
	}
	// Which stream the launch goes to: the context's, or (short launches on its own stream) the next in turn.
	hipStream_t stream = c->stream;
	hipEvent_t ev0 = c->ev0, ev1 = c->ev1;
	rthip_ctx::Lane* lane = nullptr;
	if (may_overlap && plan.overlap_ok && !plan.cfg.stack_mem) {
		if (c->lane_turn > 0) {
			lane = &c->side[c->lane_turn - 1];
			if ((rc = ensure_lane(c, lane)) != RTHIP_OK) return rc;
			// behind what the context has queued so far (buffer clears, uploads), not behind its launches in flight
			HIP_TRY(c, hipEventRecord(c->fork, c->own_stream));
			HIP_TRY(c, hipStreamWaitEvent(lane->stream, c->fork, 0));
			stream = lane->stream; ev0 = lane->ev0; ev1 = lane->ev1; control = lane->control;
			a.queue = reinterpret_cast<uint32_t*>(control + 512);
			a.stats = reinterpret_cast<unsigned long long*>(control + 8);
		}
		c->lane_turn = (c->lane_turn + 1) % (rthip_ctx::kSideLanes + 1);
	} else if ((rc = join_lanes(c)) != RTHIP_OK) {   // a launch of full size, or one that is not the context's to spread: one after the other
		return rc;
	}
	HIP_TRY(c, hipMemsetAsync(control, 0, kControlBytes, stream));
	if (timed) HIP_TRY(c, hipEventRecord(ev0, stream));
	if (plan.n_items > 0 && p->spp_count > 0) HIP_TRY(c, launch_render(a, plan.cfg, stream));
	if (!timed) return RTHIP_OK;  // a captured launch: its statistics stay in the control block, nobody reads them
	HIP_TRY(c, hipEventRecord(ev1, stream));
	if (lane) { HIP_TRY(c, hipEventRecord(lane->done, stream)); lane->busy = true; }
	c->last_ev0 = ev0; c->last_ev1 = ev1;
	c->launch_pending = true;
	c->last_control = control;
	c->last_in_lds = plan.cfg.scene_in_lds;
	c->last_threads = plan.cfg.threads; c->last_grid = plan.cfg.grid; c->last_lds_bytes = (int)plan.cfg.dyn_lds; c->last_waves_per_cu = plan.waves_per_cu;
	return RTHIP_OK;
}

// Wavefront variant: batches of paths go through generate, then extend + shade per bounce.
template <typename Real>
int32_t enqueue_wavefront(rthip_ctx* c, const rthip_render_params* p, unsigned long long* accum, Real* samples_out) {
	Plan plan;
	int32_t rc = make_plan(c, p, false, &plan, true);
	if (rc != RTHIP_OK) return rc;
	const SceneBlob& blob = std::is_same<Real, double>::value ? c->blob64 : c->blob32;
	const uint64_t per_item = (uint64_t)kTilePixels * plan.chunk;
	const uint64_t total = (uint64_t)plan.n_items * per_item;
	uint64_t batch = (uint64_t)std::max(1, c->tune.wf_batch);
	batch = std::min<uint64_t>(std::max<uint64_t>(batch, per_item), std::max<uint64_t>(total, 1));
	if (batch > 0xFFFFFF00ull) batch = 0xFFFFFF00ull;
	// carve the state memory: two states (3 quads + meta + slot) and the hit records
	const size_t quad = sizeof(Quad<Real>);
	auto up = [](size_t v) { return (v + 255) / 256 * 256; };
	const size_t state_bytes = 3 * up(batch * quad) + up(batch * 16) + up(batch * 4);
	const size_t need = 2 * state_bytes + up(batch * sizeof(Real)) + up(batch * 4);
	if (c->wf_bytes < need) {
		if (c->wf_mem) { HIP_TRY(c, hipFree(c->wf_mem)); c->wf_mem = nullptr; c->wf_bytes = 0; }
		HIP_TRY(c, hipMalloc(&c->wf_mem, need));
		c->wf_bytes = need;
	}
	if (!c->wf_counters) HIP_TRY(c, hipMalloc(&c->wf_counters, (2 * kWfMaxDepth + 2) * sizeof(uint32_t)));
	WfState<Real> st[2];
	char* m = c->wf_mem;
	for (int k = 0; k < 2; k++) {
		st[k].q0 = reinterpret_cast<Quad<Real>*>(m); m += up(batch * quad);
		st[k].q1 = reinterpret_cast<Quad<Real>*>(m); m += up(batch * quad);
		st[k].q2 = reinterpret_cast<Quad<Real>*>(m); m += up(batch * quad);
		st[k].q3 = reinterpret_cast<uint4*>(m); m += up(batch * 16);
		st[k].slot = samples_out ? reinterpret_cast<uint32_t*>(m) : nullptr; m += up(batch * 4);
	}
	WfArgs<Real> a;
	std::memset(&a, 0, sizeof(a));
	a.hit_t = reinterpret_cast<Real*>(m); m += up(batch * sizeof(Real));
	a.hit_prim = reinterpret_cast<int32_t*>(m);
	a.scene = blob.dev; a.prims_off = blob.prims_off; a.mats_off = blob.mats_off; a.gates_off = blob.gates_off;
	a.scene_bytes = blob.bytes; a.lds_bytes = blob.lds_bytes; a.stack_entries = c->tree_depth;
	fill_camera<Real>(c->cam, &a.cam);
	a.width = c->cam.width; a.height = c->cam.height; a.tiles_x = c->tiles_x; a.tiles_y = c->tiles_y;
	a.spp_begin = p->spp_begin; a.spp_count = p->spp_count; a.max_depth = p->max_depth; a.seed = p->seed; a.sky_mode = p->sky_mode;
	a.part_index = p->part_index; a.part_count = p->part_count; a.chunk = plan.chunk; a.n_chunks = plan.n_chunks;
	a.accum = accum; a.samples_out = samples_out;
	a.stats = reinterpret_cast<unsigned long long*>(c->d_control + 8);
	const int stream_grid = c->num_cu * 2;  // workgroups of 1024 threads
	const int depth = std::max(0, p->max_depth);
	uint32_t* n = c->wf_counters;
	uint32_t* fetch = c->wf_counters + kWfMaxDepth + 1;
	HIP_TRY(c, hipMemsetAsync(c->d_control, 0, kControlBytes, c->stream));
	HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
	for (uint64_t begin = 0; begin < total && p->spp_count > 0; begin += batch) {
		HIP_TRY(c, hipMemsetAsync(c->wf_counters, 0, (2 * kWfMaxDepth + 2) * sizeof(uint32_t), c->stream));
		a.path_begin = begin;
		a.path_count = (uint32_t)std::min<uint64_t>(batch, total - begin);
		a.out = st[0]; a.n_out = n;
		HIP_TRY(c, launch_wf_generate(a, stream_grid, c->stream));
		for (int b = 0; b < depth; b++) {
			a.in = st[b & 1]; a.out = st[(b + 1) & 1];
			a.n_in = n + b; a.n_out = n + b + 1; a.fetch = fetch + b;
			HIP_TRY(c, launch_wf_extend(a, plan.cfg, c->stream));
			HIP_TRY(c, launch_wf_shade(a, stream_grid, c->stream));
			// Stages whose queue is empty return at once (a few microseconds of dispatch each), so paths of
			// up to 64 bounces are queued without a look at the device; beyond that, one look every 64
			// bounces stops the launches of a batch whose paths have all ended.
			if ((b & 63) == 63 && b + 1 < depth) {
				uint32_t left = 0;
				HIP_TRY(c, hipMemcpyAsync(&left, n + b + 1, sizeof(left), hipMemcpyDeviceToHost, c->stream));
				HIP_TRY(c, hipStreamSynchronize(c->stream));
				if (left == 0) break;
			}
		}
	}
	HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
	c->last_ev0 = c->ev0; c->last_ev1 = c->ev1;
	c->launch_pending = true;
	c->last_control = c->d_control;
	c->last_in_lds = plan.cfg.scene_in_lds;
	c->last_threads = c->last_grid = c->last_lds_bytes = c->last_waves_per_cu = 0;
	return RTHIP_OK;
}

int32_t collect_stats(rthip_ctx* c) {
	if (!c->launch_pending) return RTHIP_OK;
	int32_t rc_join = join_lanes(c);
	if (rc_join != RTHIP_OK) return rc_join;
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	if (c->stream != c->own_stream) HIP_TRY(c, hipStreamSynchronize(c->own_stream));   // (side streams were joined into the own stream)
	unsigned long long s[5 + 2 * kRegCount + kPhCount];
	HIP_TRY(c, hipMemcpy(s, (c->last_control ? c->last_control : c->d_control) + 8, sizeof(s), hipMemcpyDeviceToHost));
	if (!c->tune.region_dump.empty() && c->tune.count) {  // instrumented runs: lane utilisation per code region
		static const char* names[kRegCount] = {"inner", "sphere", "cube", "leaf", "shade", "sky", "lambert", "metal", "glass", "start", "trav_round", "xchg_push", "xchg_pop", "xchg_full", "xchg_spin", "packet"};
		if (FILE* f = std::fopen(c->tune.region_dump.c_str(), "a")) {
			for (int k = 0; k < kRegCount; k++)
				std::fprintf(f, "%s lane_iters %llu wave_iters %llu util %.3f\n", names[k], s[5 + k], s[5 + kRegCount + k],
				             s[5 + kRegCount + k] ? (double)s[5 + k] / (64.0 * (double)s[5 + kRegCount + k]) : 0.0);
			// wave clock ticks (s_memtime) per phase of the main loop, summed over the waves, and their share of the waves' lifetime
			static const char* phases[kPhCount] = {"shade", "regen", "fresh", "inner", "leaf", "xchg", "packet", "total"};
			for (int k = 0; k < kPhCount; k++)
				std::fprintf(f, "phase %s ticks %llu share %.3f\n", phases[k], s[5 + 2 * kRegCount + k],
				             s[5 + 2 * kRegCount + kPhTotal] ? (double)s[5 + 2 * kRegCount + k] / (double)s[5 + 2 * kRegCount + kPhTotal] : 0.0);
			std::fclose(f);
		}
	}
	float ms = 0.f;
	HIP_TRY(c, hipEventElapsedTime(&ms, c->last_ev0 ? c->last_ev0 : c->ev0, c->last_ev1 ? c->last_ev1 : c->ev1));
	c->stats.paths = s[0]; c->stats.rays = s[1]; c->stats.box_tests = s[2]; c->stats.sphere_tests = s[3]; c->stats.cube_tests = s[4];
	c->stats.kernel_ms = ms;
	c->stats.scene_in_lds = c->last_in_lds ? 1 : 0;
	c->stats.launch_threads = c->last_threads; c->stats.launch_grid = c->last_grid;
	c->stats.launch_lds_bytes = c->last_lds_bytes; c->stats.waves_per_cu = c->last_waves_per_cu;
	c->launch_pending = false;
	return RTHIP_OK;
}

// Orders the render stream behind the reduces in flight on `buf` (all of them when buf is null).
int32_t wait_merges(rthip_ctx* c, const void* buf) {
	for (auto& m : c->merges) {
		if (!m.active || (buf && m.buf != buf)) continue;
		HIP_TRY(c, hipStreamWaitEvent(c->stream, m.done, 0));
		m.active = false;
	}
	return RTHIP_OK;
}

int32_t ensure_accum(rthip_ctx* c) {
	uint64_t words = (uint64_t)c->tiles_x * c->tiles_y * kTilePixels * kAccumChannels;
	if (c->accum && c->accum != c->own_accum) {
		if (c->accum_words != words) return fail(c, RTHIP_ERR_STATE, "caller-owned accumulation buffer does not match the image size");
		return wait_merges(c, c->accum);
	}
	if (c->own_words != words) {
		if (c->own_accum) { HIP_TRY(c, hipFree(c->own_accum)); c->own_accum = nullptr; c->own_words = 0; }
		HIP_TRY(c, hipMalloc(&c->own_accum, words * 8));
		c->own_words = words;
		HIP_TRY(c, hipMemsetAsync(c->own_accum, 0, words * 8, c->stream));
	}
	c->accum = c->own_accum; c->accum_words = words;
	return wait_merges(c, c->accum);
}

}  // namespace

namespace {
// Who used a GPU's static control block last: the launches of all contexts on one GPU share it, and
// two of them in flight at once on different streams would count in each other's work queues.
struct ScratchUse {
	std::mutex mu;
	rthip_ctx* owner = nullptr;
	hipStream_t stream = nullptr;
	hipEvent_t done = nullptr;  // the owner's event, recorded behind its launch
};
ScratchUse g_scratch_use[64];
// A context that goes away (its stream has been waited for) no longer stands for a launch in flight.
void forget_scratch_use(rthip_ctx* c) {
	if (c->device < 0 || c->device >= 64) return;
	ScratchUse& use = g_scratch_use[c->device];
	std::lock_guard<std::mutex> lock(use.mu);
	if (use.owner == c) { use.owner = nullptr; use.stream = nullptr; use.done = nullptr; }
}
}  // namespace

extern "C" {

int32_t rthip_create(int32_t device, rthip_ctx** out) {
	return guarded(nullptr, [&]() -> int32_t {
		if (!out) return fail(nullptr, RTHIP_ERR_INVALID, "out is NULL");
		*out = nullptr;
		int n = 0;
		hipError_t e = hipGetDeviceCount(&n);
		if (e != hipSuccess || n < 1) return fail(nullptr, RTHIP_ERR_HIP, std::string("no HIP device: ") + hipGetErrorString(e));
		if (device < 0 || device >= n) return fail(nullptr, RTHIP_ERR_INVALID, "device ordinal out of range");
		rthip_ctx* c = new rthip_ctx();
		c->device = device;
		auto bail = [&](const char* what, hipError_t err) {
			std::string msg = std::string(what) + ": " + hipGetErrorString(err);
			rthip_destroy(c);
			return fail(nullptr, RTHIP_ERR_HIP, msg);
		};
		try {
			c->tune = Tuning::from_env();
			c->build_opt = BuildOptions::from_env();
		} catch (...) {
			rthip_destroy(c);
			throw;
		}
		if ((e = hipSetDevice(device)) != hipSuccess) return bail("hipSetDevice", e);
		hipDeviceProp_t prop;
		if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return bail("hipGetDeviceProperties", e);
		c->num_cu = prop.multiProcessorCount;
		if ((e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
		c->stream = c->own_stream;
		// The side streams of short launches (rthip_ctx::Lane) are made right behind the own stream: a process has few
		// hardware queues (four by default) that its streams are dealt out to in turn, and streams that share one
		// do not overlap; made together, the three get three different queues.
		for (auto& l : c->side)
			if ((e = hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
		if ((e = hipEventCreate(&c->ev0)) != hipSuccess) return bail("hipEventCreate", e);
		if ((e = hipEventCreate(&c->ev1)) != hipSuccess) return bail("hipEventCreate", e);
		if ((e = hipEventCreateWithFlags(&c->noalloc_done, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
		if ((e = hipMalloc(&c->d_control, kControlBytes)) != hipSuccess) return bail("hipMalloc", e);
		// this device's copy of the static scratch array (the allocation-free entry point launches on it)
		void* scratch = nullptr;
		if ((e = static_scratch_for_device(device, &scratch)) != hipSuccess) return bail("hipGetSymbolAddress(static scratch)", e);
		c->static_control = static_cast<char*>(scratch);
		*out = c;
		return RTHIP_OK;
	});
}

void rthip_destroy(rthip_ctx* c) {
	if (!c) return;
	try {
		(void)hipSetDevice(c->device);
		if (c->stream) (void)hipStreamSynchronize(c->stream);
		forget_scratch_use(c);
		if (c->own_stream && c->own_stream != c->stream) (void)hipStreamSynchronize(c->own_stream);
		for (auto& l : c->side) {
			if (l.stream) { (void)hipStreamSynchronize(l.stream); (void)hipStreamDestroy(l.stream); }
			if (l.control) (void)hipFree(l.control);
			if (l.ev0) (void)hipEventDestroy(l.ev0);
			if (l.ev1) (void)hipEventDestroy(l.ev1);
			if (l.done) (void)hipEventDestroy(l.done);
		}
		if (c->fork) (void)hipEventDestroy(c->fork);
		if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
		if (c->comm) {
			if (const RcclApi* rccl = RcclApi::get(nullptr)) (void)rccl->CommDestroy(c->comm);
		}
		for (auto& m : c->merges) {
			if (m.ready) (void)hipEventDestroy(m.ready);
			if (m.done) (void)hipEventDestroy(m.done);
		}
		if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
		if (c->blob32.dev) (void)hipFree(c->blob32.dev);
		if (c->blob64.dev) (void)hipFree(c->blob64.dev);
		if (c->own_accum) (void)hipFree(c->own_accum);
		if (c->d_control) (void)hipFree(c->d_control);
		if (c->stack_mem) (void)hipFree(c->stack_mem);
		if (c->wf_mem) (void)hipFree(c->wf_mem);
		if (c->readback) (void)hipFree(c->readback);
		if (c->wf_counters) (void)hipFree(c->wf_counters);
		if (c->ev0) (void)hipEventDestroy(c->ev0);
		if (c->ev1) (void)hipEventDestroy(c->ev1);
		if (c->noalloc_done) (void)hipEventDestroy(c->noalloc_done);
		if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
		delete c;
	} catch (...) {
	}
}

const char* rthip_last_error(const rthip_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int32_t rthip_set_stream(rthip_ctx* c, void* hip_stream) {
	if (!c) return RTHIP_ERR_INVALID;
	return guarded(c, [&]() -> int32_t {
		HIP_TRY(c, hipSetDevice(c->device));
		// Everything queued on the old stream (renders, but also the zero-fills of rthip_camera_set and
		// rthip_reset_accumulation) must be done before work on the new stream may touch the same
		// buffers: the two streams are not ordered against each other.
		int32_t rc = collect_stats(c);
		if (rc != RTHIP_OK) return rc;
		HIP_TRY(c, hipStreamSynchronize(c->stream));
		forget_scratch_use(c);
		c->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
		return RTHIP_OK;
	});
}

int32_t rthip_set_option(rthip_ctx* c, const char* name, double value) {
	if (!c) return RTHIP_ERR_INVALID;
	return guarded(c, [&]() -> int32_t {
		if (!name) return fail(c, RTHIP_ERR_INVALID, "option name is NULL");
		// (int)value of a NaN, an infinity or anything beyond the int range is undefined
		if (!(value == value) || !(value >= -2147483648.0 && value <= 2147483647.0)) return fail(c, RTHIP_ERR_INVALID, "option value is not a number in the int range");
		const std::string k(name);
		const int iv = (int)value;
		Tuning& t = c->tune;
		BuildOptions& b = c->build_opt;
		if (k == "scene_lds") t.scene_lds = iv;
		else if (k == "qnodes") t.qnodes = iv;
		else if (k == "mats_lds") t.mats_lds = iv;
		else if (k == "acc_lds") t.acc_lds = iv;
		else if (k == "root_leaf") t.root_leaf = iv;
		else if (k == "packet") t.packet = iv;
		else if (k == "overlap") t.overlap = iv;
		else if (k == "pixel_major") t.pixel_major = iv;
		else if (k == "stack_mem") t.stack_mem = iv;
		else if (k == "xchg") t.xchg = iv;
		else if (k == "xchg_cap") t.xchg_cap = iv;
		else if (k == "xchg_pop_min") t.xchg_pop_min = iv;
		else if (k == "xchg_push_max") t.xchg_push_max = iv;
		else if (k == "threads") t.threads = iv;
		else if (k == "blocks_per_cu") t.blocks_per_cu = iv;
		else if (k == "grid_reserve") t.grid_reserve = iv;
		else if (k == "chunk") t.chunk = iv;
		else if (k == "trav_threshold") t.trav_threshold = iv;
		else if (k == "inner_exit") t.inner_exit = iv;
		else if (k == "regen_min") t.regen_min = iv;
		else if (k == "block_order") t.block_order = iv;
		else if (k == "wf_batch") t.wf_batch = iv;
		else if (k == "count") t.count = iv;
		else if (k == "tree_ref") b.tree_ref = iv != 0;
		else if (k == "tree_gpu") b.tree_gpu = iv != 0;
		else if (k == "ref_gpu") b.ref_gpu = iv;
		else if (k == "gpu_leaf") b.gpu_leaf = iv;
		else if (k == "sah_leaf") b.sah_leaf = iv;
		else if (k == "sah_kind_cap") b.sah_kind_cap = iv;
		else if (k == "sah_cprim") b.sah_cprim = value;
		else if (k == "sah_sweep") b.sah_sweep = iv;
		else if (k == "big_leaf_min") b.big_leaf_min = iv;
		else if (k == "no_big_leaf") b.no_big_leaf = iv != 0;
		else if (k == "no_qnodes") b.no_qnodes = iv != 0;
		else if (k == "build_threads") b.threads = iv != 1;
		else if (k == "tie_reverse") b.tie_reverse = iv != 0;
		else return fail(c, RTHIP_ERR_INVALID, "unknown option '" + k + "'");
		return RTHIP_OK;
	});
}

int32_t rthip_scene_upload(rthip_ctx* c, const rthip_object* objects, int64_t n) {
	if (!c) return RTHIP_ERR_INVALID;
	return guarded(c, [&]() -> int32_t {
		if (n < 0 || (n > 0 && !objects)) return fail(c, RTHIP_ERR_INVALID, "objects is NULL or n < 0");
		if (n == 0) return fail(c, RTHIP_ERR_EMPTY, "empty scene: BVH::new returns None");
		HIP_TRY(c, hipSetDevice(c->device));
		HIP_TRY(c, hipStreamSynchronize(c->stream));
		std::vector<rthip_object> fetched;  // objects handed over in device memory: a build starts on the host
		if (in_device_memory(objects)) {
			if (n > kMaxObjects) return fail(c, RTHIP_ERR_INVALID, "too many objects");
			fetched.resize((size_t)n);
			HIP_TRY(c, hipMemcpy(fetched.data(), objects, (size_t)n * sizeof(rthip_object), hipMemcpyDeviceToHost));
			objects = fetched.data();
		}
		FlatScene flat;
		std::string err;
		int32_t rc = RTHIP_OK;
		int64_t n_prims = 0, n_gates = 0;
		bool on_device = false;
		if (c->build_opt.tree_gpu && reference_pass_on_device(c->build_opt, n)) {
			// Everything that hangs on the objects is made on the device: the host checks them, picks the
			// oversized ones and says which object goes where (host.cpp); tie ranks and gates, primitive and
			// material records (refpass_gpu.hip) and the tree (build_gpu.hip) are the GPU's.
			auto t0 = std::chrono::steady_clock::now();
			const auto lap = [&](const char* what) {
				if (!c->build_opt.timing) return;
				const auto now = std::chrono::steady_clock::now();
				std::fprintf(stderr, "[rthip build] %-28s %.3f s\n", what, std::chrono::duration<double>(now - t0).count());
				t0 = now;
			};
			std::vector<int32_t> slot_ids;
			if (!plan_device_scene(objects, n, &flat, &slot_ids, &err, c->build_opt)) return fail(c, RTHIP_ERR_INVALID, err);
			lap("checks and layout (host)");
			if (flat.gpu.on) {
				RefPassHolder h;
				HIP_TRY(c, gpu_reference_pass(objects, n, c->stream, &h.ref));
				lap("copy, ranks, gates (GPU)");
				c->has_scene = false;
				n_prims = (int64_t)slot_ids.size(); n_gates = h.ref.n_gates;
				rc = alloc_blob<float>(c, (size_t)flat.gpu.n_nodes, (size_t)n_prims, (size_t)n_gates, flat.gpu.quantise, &c->blob32);
				if (rc != RTHIP_OK) return rc;
				rc = alloc_blob<double>(c, (size_t)flat.gpu.n_nodes, (size_t)n_prims, (size_t)n_gates, false, &c->blob64);
				if (rc != RTHIP_OK) return rc;
				if (flat.gpu.node_base == 1) {  // node 0: the leaf of oversized primitives on the left, the tree to come on the right
					HIP_TRY(c, hipMemcpy(c->blob32.dev, flat.nodes32.data(), sizeof(Node<float>), hipMemcpyHostToDevice));
					HIP_TRY(c, hipMemcpy(c->blob64.dev, flat.nodes64.data(), sizeof(Node<double>), hipMemcpyHostToDevice));
				}
				if (n_gates) {
					HIP_TRY(c, hipMemcpy(c->blob32.dev + c->blob32.gates_off, h.ref.gates32, (size_t)n_gates * sizeof(Gate<float>), hipMemcpyDeviceToDevice));
					HIP_TRY(c, hipMemcpy(c->blob64.dev + c->blob64.gates_off, h.ref.gates64, (size_t)n_gates * sizeof(Gate<double>), hipMemcpyDeviceToDevice));
				}
				int32_t* d_ids = nullptr;
				HIP_TRY(c, hipMalloc(&d_ids, slot_ids.size() * sizeof(int32_t)));
				hipError_t e = hipMemcpy(d_ids, slot_ids.data(), slot_ids.size() * sizeof(int32_t), hipMemcpyHostToDevice);
				if (e == hipSuccess)
					e = gpu_emit_prims(h.ref, d_ids, 1, (int32_t)n_prims, reinterpret_cast<Prim<float>*>(c->blob32.dev + c->blob32.prims_off),
					                   reinterpret_cast<Prim<double>*>(c->blob64.dev + c->blob64.prims_off), reinterpret_cast<Mat<float>*>(c->blob32.dev + c->blob32.mats_off),
					                   reinterpret_cast<Mat<double>*>(c->blob64.dev + c->blob64.mats_off), c->stream);
				if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
				(void)hipFree(d_ids);
				HIP_TRY(c, e);
				flat.n_leaves = (int)ref_tree_leaves(n);
				flat.ref_depth = std::max(1, ref_tree_levels(n));
				on_device = true;
				lap("records (GPU)");
			}
		}
		if (!on_device) {
			// The device tree from the host (surface-area build, or the reference's shape).  For large scenes
			// the GPU computes tie ranks and gates first (17 ms for a million objects, where the host's pass over
			// the reference's tree takes 0.27 s) and rewrites the uploaded primitive records with them.
			RefPassHolder h;
			BuildOptions opt = c->build_opt;
			const bool ranks_on_device = !opt.tree_gpu && !opt.tree_ref && reference_pass_on_device(opt, n);
			if (ranks_on_device) {
				if (!check_scene_objects(objects, n, &err)) return fail(c, RTHIP_ERR_INVALID, err);
				const auto t0 = std::chrono::steady_clock::now();
				HIP_TRY(c, gpu_reference_pass(objects, n, c->stream, &h.ref));
				if (opt.timing)
					std::fprintf(stderr, "[rthip build] %-28s %.3f s\n", "copy, ranks, gates (GPU)", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
				opt.ranks_elsewhere = true;
			}
			if (!build_flat_scene(objects, n, &flat, &err, opt)) return fail(c, flat.depth > kStackMax ? RTHIP_ERR_LIMIT : RTHIP_ERR_INVALID, err);
			if (ranks_on_device) {  // room for the gates in the layout
				flat.gates32.assign((size_t)h.ref.n_gates, Gate<float>{});
				flat.gates64.assign((size_t)h.ref.n_gates, Gate<double>{});
				flat.n_leaves = (int)ref_tree_leaves(n);
				flat.ref_depth = std::max(1, ref_tree_levels(n));
			}
			c->has_scene = false;
			rc = upload_blob<float>(c, flat.nodes32, flat.prims32, flat.mats32, flat.gates32, &c->blob32, &flat.qnodes, &flat.qframe, ranks_on_device);
			if (rc != RTHIP_OK) return rc;
			rc = upload_blob<double>(c, flat.nodes64, flat.prims64, flat.mats64, flat.gates64, &c->blob64, nullptr, nullptr, ranks_on_device);
			if (rc != RTHIP_OK) return rc;
			n_prims = (int64_t)flat.prims32.size(); n_gates = (int64_t)flat.gates32.size();
			if (ranks_on_device) {
				if (n_gates) {
					HIP_TRY(c, hipMemcpy(c->blob32.dev + c->blob32.gates_off, h.ref.gates32, (size_t)n_gates * sizeof(Gate<float>), hipMemcpyDeviceToDevice));
					HIP_TRY(c, hipMemcpy(c->blob64.dev + c->blob64.gates_off, h.ref.gates64, (size_t)n_gates * sizeof(Gate<double>), hipMemcpyDeviceToDevice));
				}
				// which object the host's tree holds in which place: 4 bytes per primitive go up instead of its 144 bytes of records
				std::vector<int32_t> slot_ids((size_t)n_prims);
				for (size_t i = 0; i < slot_ids.size(); i++) slot_ids[i] = flat.prims32[i].id;
				int32_t* d_ids = nullptr;
				HIP_TRY(c, hipMalloc(&d_ids, std::max<size_t>(slot_ids.size(), 1) * sizeof(int32_t)));
				hipError_t e = hipMemcpy(d_ids, slot_ids.data(), slot_ids.size() * sizeof(int32_t), hipMemcpyHostToDevice);
				if (e == hipSuccess)
					e = gpu_emit_prims(h.ref, d_ids, 1, (int32_t)n_prims, reinterpret_cast<Prim<float>*>(c->blob32.dev + c->blob32.prims_off),
					                   reinterpret_cast<Prim<double>*>(c->blob64.dev + c->blob64.prims_off), reinterpret_cast<Mat<float>*>(c->blob32.dev + c->blob32.mats_off),
					                   reinterpret_cast<Mat<double>*>(c->blob64.dev + c->blob64.mats_off), c->stream);
				if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
				(void)hipFree(d_ids);
				HIP_TRY(c, e);
			}
		}
		if (flat.gpu.on) {
			// the tree over the laid-out primitives is built on the device (build_gpu.hip)
			const FlatScene::GpuPlan& g = flat.gpu;
			GpuBuildIO io{};
			io.blob32 = c->blob32.dev; io.blob64 = c->blob64.dev;
			io.prims_off32 = c->blob32.prims_off; io.mats_off32 = c->blob32.mats_off; io.qnodes_off32 = c->blob32.qnodes_off;
			io.prims_off64 = c->blob64.prims_off; io.mats_off64 = c->blob64.mats_off;
			io.first = g.first; io.count = g.count; io.leaf_size = g.leaf_size; io.n_leaves = g.n_leaves; io.node_base = g.node_base;
			io.n_nodes = g.n_nodes;
			for (int k = 0; k < 3; k++) { io.clo[k] = g.clo[k]; io.cinv[k] = g.cinv[k]; }
			io.quantise = g.quantise && c->blob32.qnodes_off != 0;
			GpuBuildResult res{};
			const auto t0 = std::chrono::steady_clock::now();
			HIP_TRY(c, gpu_build_tree(io, c->stream, &res));
			if (c->build_opt.timing)
				std::fprintf(stderr, "[rthip build] %-28s %.3f s\n", "device tree (on the GPU)", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
			flat.depth = std::max(1, res.levels + g.node_base);
			if (flat.depth > kStackMax) return fail(c, RTHIP_ERR_LIMIT, "tree deeper than the traversal stack (device build: " + std::to_string(flat.depth) + " levels)");
			if (res.quantised) { c->blob32.qframe = res.frame; c->blob32.qframe.big_leaf = flat.big_leaf; }
			else c->blob32.qnodes_off = 0;
		}
		c->big_leaf = flat.big_leaf; c->start = flat.start;
		// A scene whose root holds one leaf and nothing else (a handful of objects): the megakernel tests
		// that leaf at ray start, like a leaf of oversized primitives, and has no tree to walk after it.
		c->root_leaf = 0;
		if (!flat.gpu.on && flat.big_leaf == 0 && flat.nodes32.size() == 1 && flat.nodes32[0].child[0] < 0 && flat.nodes32[0].child[0] != leaf_code(0, 0, 0) &&
		    flat.nodes32[0].child[1] == leaf_code(0, 0, 0))
			c->root_leaf = flat.nodes32[0].child[0];
		c->tree_depth = flat.depth; c->n_nodes = flat.gpu.on ? flat.gpu.n_nodes : (int)flat.nodes32.size(); c->n_leaves = flat.n_leaves; c->n_prims = (int)n_prims;
		c->n_gates = (int)n_gates;
		c->n_objects = n;
		c->stats.n_nodes = c->n_nodes; c->stats.n_leaves = c->n_leaves; c->stats.n_prims = c->n_prims; c->stats.tree_depth = c->tree_depth;
		c->stats.n_gates = c->n_gates;
		c->has_scene = true;
		return RTHIP_OK;
	});
}

int32_t rthip_scene_refit(rthip_ctx* c, const rthip_object* objects, int64_t n) {
	if (!c) return RTHIP_ERR_INVALID;
	return guarded(c, [&]() -> int32_t {
		if (!c->has_scene) return fail(c, RTHIP_ERR_STATE, "upload a scene first");
		if (n < 0 || (n > 0 && !objects)) return fail(c, RTHIP_ERR_INVALID, "objects is NULL or n < 0");
		if (n != c->n_objects) return fail(c, RTHIP_ERR_INVALID, "refit: the scene on the device has " + std::to_string(c->n_objects) + " objects, not " + std::to_string(n) + " (upload the scene instead)");
		HIP_TRY(c, hipSetDevice(c->device));
		HIP_TRY(c, hipStreamSynchronize(c->stream));
		std::string err;
		int32_t rc = RTHIP_OK;
		int64_t n_gates = 0;
		const auto t_start = std::chrono::steady_clock::now();
		const bool device_objects = in_device_memory(objects);
		std::vector<rthip_object> fetched;
		if (device_objects && !reference_pass_on_device(c->build_opt, n)) {  // the host's pass wants them on the host
			fetched.resize((size_t)n);
			HIP_TRY(c, hipMemcpy(fetched.data(), objects, (size_t)n * sizeof(rthip_object), hipMemcpyDeviceToHost));
			objects = fetched.data();
		}
		if (reference_pass_on_device(c->build_opt, n)) {
			// ranks, gates and the primitive records on the device; the id words of the primitives the
			// tree holds say which object goes where
			int64_t hittable = 0;
			if (device_objects) {  // nothing of such a refit touches the host
				int64_t first_bad = -1;
				HIP_TRY(c, gpu_check_objects(objects, n, &first_bad, &hittable, c->stream));
				if (first_bad >= 0) return fail(c, RTHIP_ERR_INVALID, "object " + std::to_string(first_bad) + " has an invalid kind, material or a non-finite field");
			} else if (!check_scene_objects(objects, n, &err, &hittable)) {
				return fail(c, RTHIP_ERR_INVALID, err);
			}
			if (hittable != c->n_prims)
				return fail(c, RTHIP_ERR_INVALID, "refit: " + std::to_string(hittable) + " objects can be hit, the tree holds " + std::to_string(c->n_prims) + " (upload the scene instead)");
			RefPassHolder h;
			HIP_TRY(c, gpu_reference_pass(objects, n, c->stream, &h.ref));
			const int32_t* id_words = reinterpret_cast<const int32_t*>(c->blob32.dev + c->blob32.prims_off + offsetof(Prim<float>, id));
			constexpr int32_t kIdStride = (int32_t)(sizeof(Prim<float>) / sizeof(int32_t));
			int32_t other = 0;
			HIP_TRY(c, gpu_same_objects(h.ref, id_words, kIdStride, c->n_prims, &other, c->stream));
			if (other) return fail(c, RTHIP_ERR_INVALID, "refit: " + std::to_string(other) + " objects are not the kind of primitive the tree holds in their place, or can no longer be hit (upload the scene instead)");
			// From here on the device holds a mixture of old and new until the last step has run.
			c->has_scene = false;
			n_gates = h.ref.n_gates;
			rc = regate_blob<float>(c, (size_t)c->n_nodes, (size_t)c->n_prims, (size_t)n_gates, &c->blob32);
			if (rc != RTHIP_OK) return rc;
			rc = regate_blob<double>(c, (size_t)c->n_nodes, (size_t)c->n_prims, (size_t)n_gates, &c->blob64);
			if (rc != RTHIP_OK) return rc;
			if (n_gates) {
				HIP_TRY(c, hipMemcpy(c->blob32.dev + c->blob32.gates_off, h.ref.gates32, (size_t)n_gates * sizeof(Gate<float>), hipMemcpyDeviceToDevice));
				HIP_TRY(c, hipMemcpy(c->blob64.dev + c->blob64.gates_off, h.ref.gates64, (size_t)n_gates * sizeof(Gate<double>), hipMemcpyDeviceToDevice));
			}
			id_words = reinterpret_cast<const int32_t*>(c->blob32.dev + c->blob32.prims_off + offsetof(Prim<float>, id));
			HIP_TRY(c, gpu_emit_prims(h.ref, id_words, kIdStride, c->n_prims, reinterpret_cast<Prim<float>*>(c->blob32.dev + c->blob32.prims_off),
			                          reinterpret_cast<Prim<double>*>(c->blob64.dev + c->blob64.prims_off), reinterpret_cast<Mat<float>*>(c->blob32.dev + c->blob32.mats_off),
			                          reinterpret_cast<Mat<double>*>(c->blob64.dev + c->blob64.mats_off), c->stream));
			HIP_TRY(c, hipStreamSynchronize(c->stream));
		} else {
			// which object sits in which place of the device tree's primitive array
			std::vector<int32_t> slot_ids((size_t)c->n_prims);
			HIP_TRY(c, gpu_slot_ids(c->blob32.dev, c->blob32.prims_off, c->n_prims, slot_ids.data(), c->stream));
			FlatScene flat;
			if (!refit_flat_scene(objects, n, slot_ids.data(), (int64_t)slot_ids.size(), &flat, &err, c->build_opt)) return fail(c, RTHIP_ERR_INVALID, err);
			// From here on the device holds a mixture of old and new until the last step has run.
			c->has_scene = false;
			rc = replace_blob_tail<float>(c, (size_t)c->n_nodes, flat.prims32, flat.mats32, flat.gates32, &c->blob32);
			if (rc != RTHIP_OK) return rc;
			rc = replace_blob_tail<double>(c, (size_t)c->n_nodes, flat.prims64, flat.mats64, flat.gates64, &c->blob64);
			if (rc != RTHIP_OK) return rc;
			n_gates = (int64_t)flat.gates32.size();
		}
		if (c->build_opt.timing)
			std::fprintf(stderr, "[rthip build] %-28s %.3f s\n", "refit: ranks, gates, records", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count());
		GpuRefitIO io{};
		io.blob32 = c->blob32.dev; io.blob64 = c->blob64.dev;
		io.prims_off32 = c->blob32.prims_off; io.prims_off64 = c->blob64.prims_off; io.qnodes_off32 = c->blob32.qnodes_off;
		io.n_nodes = c->n_nodes; io.levels = c->tree_depth;
		io.big_leaf = c->big_leaf != 0;
		io.quantise = c->blob32.qnodes_off != 0;
		GpuRefitResult res{};
		const auto t0 = std::chrono::steady_clock::now();
		HIP_TRY(c, gpu_refit_tree(io, c->stream, &res));
		if (c->build_opt.timing)
			std::fprintf(stderr, "[rthip build] %-28s %.3f s\n", "refit (on the GPU)", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
		if (res.unreached) return fail(c, RTHIP_ERR_INTERNAL, "refit: " + std::to_string(res.unreached) + " nodes of the device tree are not below its root");
		if (io.quantise) {
			if (res.quantised) { for (int k = 0; k < 3; k++) { c->blob32.qframe.centre[k] = res.frame.centre[k]; c->blob32.qframe.step[k] = res.frame.step[k]; } }
			else c->blob32.qnodes_off = 0;
		}
		c->n_gates = (int)n_gates;
		c->stats.n_gates = c->n_gates;
		c->has_scene = true;
		return RTHIP_OK;
	});
}

int32_t rthip_camera_set(rthip_ctx* c, const rthip_camera_desc* d) {
	if (!c) return RTHIP_ERR_INVALID;
	return guarded(c, [&]() -> int32_t {
		if (!d) return fail(c, RTHIP_ERR_INVALID, "camera is NULL");
		if (d->width < 1 || d->height < 1 || (int64_t)d->width * d->height > (1ll << 28)) return fail(c, RTHIP_ERR_INVALID, "image size out of range");
		if (!(d->focus_dist == d->focus_dist) || !(d->fov == d->fov)) return fail(c, RTHIP_ERR_INVALID, "camera has NaN fields");
		HIP_TRY(c, hipSetDevice(c->device));
		bool resized = !c->has_camera || d->width != c->cam.width || d->height != c->cam.height;
		c->cam = camera_new(*d);
		c->tiles_x = (d->width + kTile - 1) / kTile;
		c->tiles_y = (d->height + kTile - 1) / kTile;
		c->has_camera = true;
		if (resized) {
			HIP_TRY(c, hipStreamSynchronize(c->stream));
			if (c->accum && c->accum != c->own_accum) { c->accum = nullptr; c->accum_words = 0; }  // caller's buffer no longer fits
			int32_t rc = ensure_accum(c);
			if (rc != RTHIP_OK) return rc;
			HIP_TRY(c, hipMemsetAsync(c->accum, 0, c->accum_words * 8, c->stream));
		}
		return RTHIP_OK;
	});
}

int32_t rthip_camera_fields(const rthip_camera_desc* d, double out[19]) {
	return guarded(nullptr, [&]() -> int32_t {
		if (!d || !out) return fail(nullptr, RTHIP_ERR_INVALID, "camera or out is NULL");
		if (d->width < 1 || d->height < 1) return fail(nullptr, RTHIP_ERR_INVALID, "image size out of range");
		CameraFields f = camera_new(*d);
		const double* src[6] = {f.origin, f.pixel00, f.delta_x, f.delta_y, f.disk_x, f.disk_y};
		for (int i = 0; i < 6; i++) for (int k = 0; k < 3; k++) out[3 * i + k] = src[i][k];
		out[18] = f.defocus_angle;
		return RTHIP_OK;
	});
}

int32_t rthip_render_accumulate_async(rthip_ctx* c, const rthip_render_params* p) {
	if (!c) return RTHIP_ERR_INVALID;
	return guarded(c, [&]() -> int32_t {
		int32_t rc = validate(c, p);
		if (rc != RTHIP_OK) return rc;
		HIP_TRY(c, hipSetDevice(c->device));
		const bool may_overlap = c->tune.overlap != 0 && c->stream == c->own_stream && p->variant != RTHIP_WAVEFRONT && c->accum != nullptr &&
		                         c->accum_words == (uint64_t)c->tiles_x * c->tiles_y * kTilePixels * kAccumChannels;
		if (!may_overlap && (rc = join_lanes(c)) != RTHIP_OK) return rc;
		// No wait for the previous launch: the control block (work counters, statistics) is cleared and
		// used in stream order, so launches queue up back to back; statistics are those of the last one.
		if ((rc = ensure_accum(c)) != RTHIP_OK) return rc;
		const bool count = c->tune.count != 0;
		if (p->variant == RTHIP_WAVEFRONT) {
			if (p->precision == RTHIP_F64) return enqueue_wavefront<double>(c, p, c->accum, nullptr);
			return enqueue_wavefront<float>(c, p, c->accum, nullptr);
		}
		if (p->precision == RTHIP_F64) return enqueue_render<double>(c, p, count, c->accum, nullptr, nullptr, false, true, may_overlap);
		return enqueue_render<float>(c, p, count, c->accum, nullptr, nullptr, false, true, may_overlap);
	}, false);
}

int32_t rthip_render_accumulate_noalloc(rthip_ctx* c, const rthip_render_params* p) {
	if (!c) return RTHIP_ERR_INVALID;
	return guarded(c, [&]() -> int32_t {
		int32_t rc = validate(c, p);
		if (rc != RTHIP_OK) return rc;
		if (p->variant != RTHIP_MEGAKERNEL)
			return fail(c, RTHIP_ERR_LIMIT, "rthip_render_accumulate_noalloc: the wavefront variant keeps gigabytes of path state in allocated memory; "
			                                "only RTHIP_MEGAKERNEL runs without allocation");
		static_assert(kControlBytes <= kStaticScratchBytes, "the control block must fit the static scratch array");
		if (!c->static_control) return fail(c, RTHIP_ERR_STATE, "rthip_render_accumulate_noalloc: no static scratch on this device");
		const uint64_t words = (uint64_t)c->tiles_x * c->tiles_y * kTilePixels * kAccumChannels;
		if (!c->accum || c->accum_words != words)
			return fail(c, RTHIP_ERR_STATE, "rthip_render_accumulate_noalloc: no accumulation buffer of the image's size is set (it will not allocate one)");
		if (c->device < 0 || c->device >= 64) return fail(c, RTHIP_ERR_LIMIT, "rthip_render_accumulate_noalloc: device ordinal beyond 63");
		HIP_TRY(c, hipSetDevice(c->device));
		// a reduce still in flight on this buffer comes first (a stream wait: nothing is allocated)
		if ((rc = wait_merges(c, c->accum)) != RTHIP_OK) return rc;
		hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
		HIP_TRY(c, hipStreamIsCapturing(c->stream, &cap));
		const bool capturing = cap != hipStreamCaptureStatusNone;
		ScratchUse& use = g_scratch_use[c->device];
		std::lock_guard<std::mutex> lock(use.mu);
		// The block is free once the launch that used it last has finished, or when this launch is
		// queued behind it on the same stream.  Anything else would be two launches sharing one set of
		// work-queue counters: refused.  (A captured launch leaves no event behind: replays of a graph
		// are the caller's to keep apart from other users of this entry point.)
		if (use.owner && use.stream != c->stream) {
			const hipError_t q = hipEventQuery(use.done);
			if (q == hipErrorNotReady)
				return fail(c, RTHIP_ERR_STATE, "rthip_render_accumulate_noalloc: another launch on this GPU is still using the static control block "
				                                "(calls that overlap must share a stream, or use rthip_render_accumulate_async)");
			if (q != hipSuccess) HIP_TRY(c, q);
			use.owner = nullptr;
		}
		// the kernels take the scratch pointer as they take the allocated control block in rthip_render_accumulate_async
		rc = p->precision == RTHIP_F64 ? enqueue_render<double>(c, p, false, c->accum, nullptr, c->static_control, true, !capturing)
		                               : enqueue_render<float>(c, p, false, c->accum, nullptr, c->static_control, true, !capturing);
		if (rc != RTHIP_OK) return rc;
		if (!capturing) {
			HIP_TRY(c, hipEventRecord(c->noalloc_done, c->stream));
			use.owner = c; use.stream = c->stream; use.done = c->noalloc_done;
		}
		return RTHIP_OK;
	});
}

int32_t rthip_synchronize(rthip_ctx* c) {
	if (!c) return RTHIP_ERR_INVALID;
	return guarded(c, [&]() -> int32_t {
		HIP_TRY(c, hipSetDevice(c->device));
		int32_t rc = collect_stats(c);
		if (rc != RTHIP_OK) return rc;
		if ((rc = wait_merges(c, nullptr)) != RTHIP_OK) return rc;
		HIP_TRY(c, hipStreamSynchronize(c->stream));
		return RTHIP_OK;
	});
}

int32_t rthip_render_accumulate(rthip_ctx* c, const rthip_render_params* p) {
	int32_t rc = rthip_render_accumulate_async(c, p);
	if (rc != RTHIP_OK) return rc;
	return rthip_synchronize(c);
}

int32_t rthip_reset_accumulation(rthip_ctx* c) {
	if (!c) return RTHIP_ERR_INVALID;
	return guarded(c, [&]() -> int32_t {
		if (!c->has_camera) return fail(c, RTHIP_ERR_STATE, "set a camera first");
		HIP_TRY(c, hipSetDevice(c->device));
		int32_t rc = ensure_accum(c);
		if (rc != RTHIP_OK) return rc;
		HIP_TRY(c, hipMemsetAsync(c->accum, 0, c->accum_words * 8, c->stream));
		return RTHIP_OK;
	});
}

int32_t rthip_comm_unique_id(uint8_t id[128]) {
	return guarded(nullptr, [&]() -> int32_t {
		if (!id) return fail(nullptr, RTHIP_ERR_INVALID, "id is NULL");
		std::string why;
		const RcclApi* rccl = RcclApi::get(&why);
		if (!rccl) return fail(nullptr, RTHIP_ERR_STATE, why);
		static_assert(sizeof(ncclUniqueId) == 128, "RCCL's unique id is 128 bytes");
		ncclUniqueId u;
		ncclResult_t r = rccl->GetUniqueId(&u);
		if (r != ncclSuccess) return fail(nullptr, RTHIP_ERR_HIP, std::string("ncclGetUniqueId: ") + rccl->GetErrorString(r));
		std::memcpy(id, &u, 128);
		return RTHIP_OK;
	});
}

int32_t rthip_comm_init(rthip_ctx* c, const uint8_t id[128], int32_t rank, int32_t world) {
	if (!c) return RTHIP_ERR_INVALID;
	return guarded(c, [&]() -> int32_t {
		if (!id || world < 1 || rank < 0 || rank >= world) return fail(c, RTHIP_ERR_INVALID, "bad communicator id, rank or size");
		std::string why;
		const RcclApi* rccl = RcclApi::get(&why);
		if (!rccl) return fail(c, RTHIP_ERR_STATE, why);
		HIP_TRY(c, hipSetDevice(c->device));
		if (c->comm) {
			// reduces still queued belong to the old communicator: let them finish before it goes
			if (c->comm_stream) HIP_TRY(c, hipStreamSynchronize(c->comm_stream));
			for (auto& m : c->merges) m.active = false;
			(void)rccl->CommDestroy(c->comm);
			c->comm = nullptr; c->comm_rank = -1; c->comm_world = 0;
		}
		if (!c->comm_stream) HIP_TRY(c, hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
		for (auto& m : c->merges) {
			if (!m.ready) HIP_TRY(c, hipEventCreateWithFlags(&m.ready, hipEventDisableTiming));
			if (!m.done) HIP_TRY(c, hipEventCreateWithFlags(&m.done, hipEventDisableTiming));
		}
		ncclUniqueId u;
		std::memcpy(&u, id, 128);
		ncclComm_t comm = nullptr;
		ncclResult_t r = rccl->CommInitRank(&comm, world, u, rank);
		if (r != ncclSuccess) return fail(c, RTHIP_ERR_HIP, std::string("ncclCommInitRank: ") + rccl->GetErrorString(r));
		c->comm = comm; c->comm_rank = rank; c->comm_world = world;  // only once everything it needs exists
		return RTHIP_OK;
	});
}

int32_t rthip_merge_reduce(rthip_ctx* c, int32_t root) {
	if (!c) return RTHIP_ERR_INVALID;
	return guarded(c, [&]() -> int32_t {
		if (!c->comm) return fail(c, RTHIP_ERR_STATE, "no communicator: call rthip_comm_init first");
		if (root < -1 || root >= c->comm_world) return fail(c, RTHIP_ERR_INVALID, "root is not a rank of the communicator");
		if (!c->has_camera) return fail(c, RTHIP_ERR_STATE, "set a camera first");
		const RcclApi* rccl = RcclApi::get(nullptr);
		HIP_TRY(c, hipSetDevice(c->device));
		int32_t rc = ensure_accum(c);  // also orders this reduce behind an earlier one on the same buffer
		if (rc != RTHIP_OK) return rc;
		// a slot for this buffer: a free one, else the first (its reduce is waited for before the slot is reused)
		rthip_ctx::PendingMerge* slot = nullptr;
		for (auto& m : c->merges) if (!m.active) { slot = &m; break; }
		if (!slot) { slot = &c->merges[0]; HIP_TRY(c, hipStreamWaitEvent(c->stream, slot->done, 0)); slot->active = false; }
		// In place, behind the renders that filled the buffer, on the communication stream: sums of
		// 64-bit integers are exact and independent of the order RCCL adds them in.
		HIP_TRY(c, hipEventRecord(slot->ready, c->stream));
		HIP_TRY(c, hipStreamWaitEvent(c->comm_stream, slot->ready, 0));
		ncclResult_t r = root < 0 ? rccl->AllReduce(c->accum, c->accum, (size_t)c->accum_words, ncclUint64, ncclSum, c->comm, c->comm_stream)
		                          : rccl->Reduce(c->accum, c->accum, (size_t)c->accum_words, ncclUint64, ncclSum, root, c->comm, c->comm_stream);
		if (r != ncclSuccess) return fail(c, RTHIP_ERR_HIP, std::string(root < 0 ? "ncclAllReduce: " : "ncclReduce: ") + rccl->GetErrorString(r));
		HIP_TRY(c, hipEventRecord(slot->done, c->comm_stream));
		slot->buf = c->accum; slot->active = true;
		return RTHIP_OK;
	});
}

int32_t rthip_merge_wait(rthip_ctx* c, const void* device_ptr) {
	if (!c) return RTHIP_ERR_INVALID;
	return guarded(c, [&]() -> int32_t {
		HIP_TRY(c, hipSetDevice(c->device));
		return wait_merges(c, device_ptr);
	});
}

int32_t rthip_merge_local(rthip_ctx* const* ctxs, int32_t n, int32_t root) {
	if (!ctxs || n < 1 || root < 0 || root >= n || !ctxs[root]) return fail(nullptr, RTHIP_ERR_INVALID, "bad context list or root");
	rthip_ctx* dst = ctxs[root];
	void* staging = nullptr;
	int32_t rc = guarded(dst, [&]() -> int32_t {
		if (!dst->has_camera) return fail(dst, RTHIP_ERR_STATE, "set a camera first");
		HIP_TRY(dst, hipSetDevice(dst->device));
		int32_t rc = ensure_accum(dst);
		if (rc != RTHIP_OK) return rc;
		for (int32_t k = 0; k < n; k++) {
			rthip_ctx* src = ctxs[k];
			if (k == root) continue;
			if (!src || !src->has_camera) return fail(dst, RTHIP_ERR_STATE, "a context of the list has no camera");
			HIP_TRY(dst, hipSetDevice(src->device));
			if ((rc = join_lanes(src)) != RTHIP_OK) return fail(dst, rc, src->err);   // short launches it spread over its side streams come first
			if ((rc = ensure_accum(src)) != RTHIP_OK) return fail(dst, rc, src->err);
			if (src->accum_words != dst->accum_words) return fail(dst, RTHIP_ERR_INVALID, "the contexts render frames of different sizes");
			if (src->accum == dst->accum) return fail(dst, RTHIP_ERR_INVALID, "two contexts of the list share one accumulation buffer");
			HIP_TRY(dst, hipStreamSynchronize(src->stream));  // its renders are done before another stream reads them
			HIP_TRY(dst, hipSetDevice(dst->device));
			const unsigned long long* from = src->accum;
			if (src->device != dst->device) {  // over xGMI into a staging buffer on the root's GPU
				if (!staging) HIP_TRY(dst, hipMalloc(&staging, dst->accum_words * 8));
				HIP_TRY(dst, hipMemcpyPeerAsync(staging, dst->device, src->accum, src->device, dst->accum_words * 8, dst->stream));
				from = static_cast<const unsigned long long*>(staging);
			}
			HIP_TRY(dst, launch_add_words(dst->accum, from, dst->accum_words, dst->stream));
		}
		HIP_TRY(dst, hipStreamSynchronize(dst->stream));
		return RTHIP_OK;
	});
	if (staging) { (void)hipSetDevice(dst->device); (void)hipStreamSynchronize(dst->stream); (void)hipFree(staging); }
	return rc;
}

int32_t rthip_accum_buffer(rthip_ctx* c, void** device_ptr, uint64_t* n_words) {
	if (!c || !device_ptr || !n_words) return RTHIP_ERR_INVALID;
	return guarded(c, [&]() -> int32_t {
		if (!c->has_camera) return fail(c, RTHIP_ERR_STATE, "set a camera first");
		HIP_TRY(c, hipSetDevice(c->device));
		int32_t rc = ensure_accum(c);
		if (rc != RTHIP_OK) return rc;
		*device_ptr = c->accum; *n_words = c->accum_words;
		return RTHIP_OK;
	});
}

int32_t rthip_set_accum_buffer(rthip_ctx* c, void* device_ptr, uint64_t n_words) {
	if (!c) return RTHIP_ERR_INVALID;
	return guarded(c, [&]() -> int32_t {
		if (!c->has_camera) return fail(c, RTHIP_ERR_STATE, "set a camera first");
		HIP_TRY(c, hipSetDevice(c->device));
		// No wait here: launches already queued carry the pointer they were given, and later calls
		// are ordered behind them on the stream.
		if (!device_ptr) { c->accum = nullptr; c->accum_words = 0; return ensure_accum(c); }
		uint64_t words = (uint64_t)c->tiles_x * c->tiles_y * kTilePixels * kAccumChannels;
		if (n_words != words) return fail(c, RTHIP_ERR_INVALID, "accumulation buffer must hold tiles_x*tiles_y*192 words");
		if (reinterpret_cast<uintptr_t>(device_ptr) % 8) return fail(c, RTHIP_ERR_INVALID, "accumulation buffer must be 8-byte aligned");
		c->accum = static_cast<unsigned long long*>(device_ptr); c->accum_words = words;
		return RTHIP_OK;
	});
}

static int32_t accum_copy(rthip_ctx* c, void* host, uint64_t n_words, bool to_host) {
	if (!c || !host) return RTHIP_ERR_INVALID;
	return guarded(c, [&]() -> int32_t {
		if (!c->has_camera) return fail(c, RTHIP_ERR_STATE, "set a camera first");
		HIP_TRY(c, hipSetDevice(c->device));
		int32_t rc = ensure_accum(c);
		if (rc != RTHIP_OK) return rc;
		if (n_words != c->accum_words) return fail(c, RTHIP_ERR_INVALID, "accumulation buffer holds tiles_x*tiles_y*192 words");
		if (to_host) HIP_TRY(c, hipMemcpyAsync(host, c->accum, n_words * 8, hipMemcpyDeviceToHost, c->stream));
		else HIP_TRY(c, hipMemcpyAsync(c->accum, host, n_words * 8, hipMemcpyHostToDevice, c->stream));
		HIP_TRY(c, hipStreamSynchronize(c->stream));
		return RTHIP_OK;
	});
}
int32_t rthip_accum_download(rthip_ctx* c, uint64_t* words, uint64_t n_words) { return accum_copy(c, words, n_words, true); }
int32_t rthip_accum_upload(rthip_ctx* c, const uint64_t* words, uint64_t n_words) { return accum_copy(c, const_cast<uint64_t*>(words), n_words, false); }

// Device staging for the read-back calls: kept between calls (a viewer reads a frame every few
// milliseconds), grown on demand, freed with the context.
static int32_t ensure_readback(rthip_ctx* c, size_t bytes) {
	if (c->readback_bytes >= bytes) return RTHIP_OK;
	if (c->readback) { HIP_TRY(c, hipFree(c->readback)); c->readback = nullptr; c->readback_bytes = 0; }
	HIP_TRY(c, hipMalloc(&c->readback, bytes));
	c->readback_bytes = bytes;
	return RTHIP_OK;
}

int32_t rthip_read_accum(rthip_ctx* c, double* rgb) {
	if (!c || !rgb) return RTHIP_ERR_INVALID;
	return guarded(c, [&]() -> int32_t {
		if (!c->has_camera) return fail(c, RTHIP_ERR_STATE, "set a camera first");
		HIP_TRY(c, hipSetDevice(c->device));
		int32_t rc = ensure_accum(c);
		if (rc != RTHIP_OK) return rc;
		size_t n = (size_t)c->cam.width * c->cam.height * 3;
		rc = ensure_readback(c, n * sizeof(double));
		if (rc != RTHIP_OK) return rc;
		double* d = static_cast<double*>(c->readback);
		HIP_TRY(c, launch_unpack(c->accum, d, c->cam.width, c->cam.height, c->tiles_x, c->stream));
		HIP_TRY(c, hipMemcpyAsync(rgb, d, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(c, hipStreamSynchronize(c->stream));
		return RTHIP_OK;
	});
}

static int32_t read_quantized(rthip_ctx* c, uint32_t spp_total, void* out, int xrgb) {
	if (!c || !out) return RTHIP_ERR_INVALID;
	return guarded(c, [&]() -> int32_t {
		if (!c->has_camera) return fail(c, RTHIP_ERR_STATE, "set a camera first");
		if (spp_total == 0) return fail(c, RTHIP_ERR_INVALID, "spp_total must be positive");
		HIP_TRY(c, hipSetDevice(c->device));
		int32_t rc = ensure_accum(c);
		if (rc != RTHIP_OK) return rc;
		size_t bytes = (size_t)c->cam.width * c->cam.height * (xrgb ? 4 : 3);
		rc = ensure_readback(c, bytes);
		if (rc != RTHIP_OK) return rc;
		HIP_TRY(c, launch_finalize(c->accum, c->readback, c->cam.width, c->cam.height, c->tiles_x, spp_total, xrgb, c->stream));
		HIP_TRY(c, hipMemcpyAsync(out, c->readback, bytes, hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(c, hipStreamSynchronize(c->stream));
		return RTHIP_OK;
	});
}
int32_t rthip_read_rgb8(rthip_ctx* c, uint32_t spp_total, uint8_t* rgb) { return read_quantized(c, spp_total, rgb, 0); }
int32_t rthip_read_xrgb32(rthip_ctx* c, uint32_t spp_total, uint32_t* xrgb) { return read_quantized(c, spp_total, xrgb, 1); }

int32_t rthip_trace_primary(rthip_ctx* c, int32_t precision, int32_t jitter, uint32_t seed, uint32_t sample, int32_t* id, double* t,
                            double* normal, uint8_t* front_face) {
	if (!c) return RTHIP_ERR_INVALID;
	return guarded(c, [&]() -> int32_t {
		if (!id || !t || !normal || !front_face) return fail(c, RTHIP_ERR_INVALID, "output pointer is NULL");
		if (!c->has_scene || !c->has_camera) return fail(c, RTHIP_ERR_STATE, "upload a scene and set a camera first");
		if (precision != RTHIP_F32 && precision != RTHIP_F64 && precision != RTHIP_F32_RAW) return fail(c, RTHIP_ERR_INVALID, "precision must be RTHIP_F32, RTHIP_F64 or RTHIP_F32_RAW");
		HIP_TRY(c, hipSetDevice(c->device));
		size_t n = (size_t)c->cam.width * c->cam.height;
		char* d = nullptr;
		size_t off_t = align_up(n * 4, 64), off_n = off_t + align_up(n * 8, 64), off_f = off_n + align_up(n * 24, 64), total = off_f + align_up(n, 64);
		HIP_TRY(c, hipMalloc(&d, total));
		size_t lds = (size_t)c->tree_depth * kProbeBlock * 4;
		hipError_t e;
		if (precision == RTHIP_F64) {
			ProbeArgs<double> a;
			std::memset(&a, 0, sizeof(a));
			a.scene = c->blob64.dev; a.prims_off = c->blob64.prims_off; a.mats_off = c->blob64.mats_off; a.gates_off = c->blob64.gates_off;
			a.scene_bytes = c->blob64.bytes;
			a.stack_entries = c->tree_depth; fill_camera<double>(c->cam, &a.cam);
			a.width = c->cam.width; a.height = c->cam.height; a.seed = seed; a.sample = sample; a.jitter = jitter;
			a.id = reinterpret_cast<int32_t*>(d); a.t = reinterpret_cast<double*>(d + off_t); a.normal = reinterpret_cast<double*>(d + off_n);
			a.front = reinterpret_cast<uint8_t*>(d + off_f);
			e = launch_probe(a, lds, c->stream);
		} else {
			ProbeArgs<float> a;
			std::memset(&a, 0, sizeof(a));
			a.scene = c->blob32.dev; a.prims_off = c->blob32.prims_off; a.mats_off = c->blob32.mats_off; a.gates_off = c->blob32.gates_off;
			a.scene_bytes = c->blob32.bytes;
			a.stack_entries = c->tree_depth; fill_camera<float>(c->cam, &a.cam);
			if (precision == RTHIP_F32) {  // refine the winner in f64; RTHIP_F32_RAW reports the f32 record as it is
				fill_camera<double>(c->cam, &a.cam64);
				a.scene64 = c->blob64.dev; a.prims_off64 = c->blob64.prims_off;
			}
			a.width = c->cam.width; a.height = c->cam.height; a.seed = seed; a.sample = sample; a.jitter = jitter;
			a.id = reinterpret_cast<int32_t*>(d); a.t = reinterpret_cast<double*>(d + off_t); a.normal = reinterpret_cast<double*>(d + off_n);
			a.front = reinterpret_cast<uint8_t*>(d + off_f);
			e = launch_probe(a, lds, c->stream);
		}
		if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
		if (e == hipSuccess) e = hipMemcpy(id, d, n * 4, hipMemcpyDeviceToHost);
		if (e == hipSuccess) e = hipMemcpy(t, d + off_t, n * 8, hipMemcpyDeviceToHost);
		if (e == hipSuccess) e = hipMemcpy(normal, d + off_n, n * 24, hipMemcpyDeviceToHost);
		if (e == hipSuccess) e = hipMemcpy(front_face, d + off_f, n, hipMemcpyDeviceToHost);
		(void)hipFree(d);
		HIP_TRY(c, e);
		return RTHIP_OK;
	});
}

int32_t rthip_render_samples(rthip_ctx* c, const rthip_render_params* p, double* samples) {
	if (!c) return RTHIP_ERR_INVALID;
	char* d_samples = nullptr;
	unsigned long long* d_scratch = nullptr;
	int32_t rc = guarded(c, [&]() -> int32_t {
		int32_t rc = validate(c, p);
		if (rc != RTHIP_OK) return rc;
		if (!samples) return fail(c, RTHIP_ERR_INVALID, "samples is NULL");
		if (p->part_count != 1) return fail(c, RTHIP_ERR_INVALID, "the sample probe renders the whole image");
		HIP_TRY(c, hipSetDevice(c->device));
		if ((rc = collect_stats(c)) != RTHIP_OK) return rc;
		size_t n = (size_t)c->cam.width * c->cam.height * 3 * p->spp_count;
		if (n > (1ull << 31)) return fail(c, RTHIP_ERR_LIMIT, "sample probe is for small images");
		uint64_t words = (uint64_t)c->tiles_x * c->tiles_y * kTilePixels * kAccumChannels;
		size_t elem = p->precision == RTHIP_F64 ? 8 : 4;
		HIP_TRY(c, hipMalloc(&d_samples, std::max<size_t>(n * elem, 8)));
		HIP_TRY(c, hipMalloc(&d_scratch, words * 8));
		HIP_TRY(c, hipMemsetAsync(d_samples, 0, n * elem, c->stream));
		HIP_TRY(c, hipMemsetAsync(d_scratch, 0, words * 8, c->stream));
		const bool count = c->tune.count != 0;
		if (p->variant == RTHIP_WAVEFRONT) {
			if (p->precision == RTHIP_F64) rc = enqueue_wavefront<double>(c, p, d_scratch, reinterpret_cast<double*>(d_samples));
			else rc = enqueue_wavefront<float>(c, p, d_scratch, reinterpret_cast<float*>(d_samples));
		} else if (p->precision == RTHIP_F64) rc = enqueue_render<double>(c, p, count, d_scratch, reinterpret_cast<double*>(d_samples));
		else rc = enqueue_render<float>(c, p, count, d_scratch, reinterpret_cast<float*>(d_samples));
		if (rc == RTHIP_OK) rc = collect_stats(c);
		if (rc != RTHIP_OK) return rc;
		if (p->precision == RTHIP_F64) {
			HIP_TRY(c, hipMemcpy(samples, d_samples, n * 8, hipMemcpyDeviceToHost));
		} else {
			std::vector<float> tmp(n);
			HIP_TRY(c, hipMemcpy(tmp.data(), d_samples, n * 4, hipMemcpyDeviceToHost));
			for (size_t i = 0; i < n; i++) samples[i] = (double)tmp[i];
		}
		return RTHIP_OK;
	});
	// whatever happened above (an exception included), the kernels are done before their buffers go
	if (d_samples || d_scratch) (void)hipStreamSynchronize(c->stream);
	if (d_samples) (void)hipFree(d_samples);
	if (d_scratch) (void)hipFree(d_scratch);
	return rc;
}

int32_t rthip_get_stats(rthip_ctx* c, rthip_stats* out) {
	if (!c || !out) return RTHIP_ERR_INVALID;
	return guarded(c, [&]() -> int32_t {
		HIP_TRY(c, hipSetDevice(c->device));
		int32_t rc = collect_stats(c);
		if (rc != RTHIP_OK) return rc;
		*out = c->stats;
		return RTHIP_OK;
	});
}

int32_t rthip_bvh_leaf_order(const rthip_object* objects, int64_t n, int32_t* ids, int32_t* leaf_sizes, int64_t* n_leaves,
                             int32_t* depth) {
	return guarded(nullptr, [&]() -> int32_t {
		if (n < 0 || (n > 0 && !objects)) return fail(nullptr, RTHIP_ERR_INVALID, "objects is NULL or n < 0");
		if (n == 0) return fail(nullptr, RTHIP_ERR_EMPTY, "empty scene: BVH::new returns None");
		FlatScene flat;
		std::string err;
		if (!build_flat_scene(objects, n, &flat, &err, host_only_options())) return fail(nullptr, flat.depth > kStackMax ? RTHIP_ERR_LIMIT : RTHIP_ERR_INVALID, err);
		if (ids) std::copy(flat.leaf_ids.begin(), flat.leaf_ids.end(), ids);
		if (leaf_sizes) std::copy(flat.leaf_sizes.begin(), flat.leaf_sizes.end(), leaf_sizes);
		if (n_leaves) *n_leaves = (int64_t)flat.leaf_sizes.size();
		if (depth) *depth = flat.ref_depth;
		return RTHIP_OK;
	});
}

int32_t rthip_bvh_validate(const rthip_object* objects, int64_t n, int64_t* violations, int32_t* n_nodes, int32_t* depth, uint64_t* checksum) {
	return guarded(nullptr, [&]() -> int32_t {
		if (n < 0 || (n > 0 && !objects)) return fail(nullptr, RTHIP_ERR_INVALID, "objects is NULL or n < 0");
		if (n == 0) return fail(nullptr, RTHIP_ERR_EMPTY, "empty scene: BVH::new returns None");
		FlatScene flat;
		std::string err;
		if (!build_flat_scene(objects, n, &flat, &err, host_only_options())) return fail(nullptr, flat.depth > kStackMax ? RTHIP_ERR_LIMIT : RTHIP_ERR_INVALID, err);
		std::string first;
		const int64_t bad = validate_flat_scene(flat, n, &first);
		if (violations) *violations = bad;
		if (n_nodes) *n_nodes = (int32_t)flat.nodes32.size();
		if (depth) *depth = flat.depth;
		if (checksum) *checksum = checksum_flat_scene(flat);
		if (bad) return fail(nullptr, RTHIP_ERR_STATE, first);
		return RTHIP_OK;
	});
}

int32_t rthip_scene_validate(rthip_ctx* c, int64_t* violations, int32_t* n_nodes, int32_t* depth) {
	if (!c) return RTHIP_ERR_INVALID;
	return guarded(c, [&]() -> int32_t {
		if (!c->has_scene) return fail(c, RTHIP_ERR_STATE, "upload a scene first");
		HIP_TRY(c, hipSetDevice(c->device));
		HIP_TRY(c, hipStreamSynchronize(c->stream));
		FlatScene s;
		s.nodes32.resize((size_t)c->n_nodes); s.nodes64.resize((size_t)c->n_nodes);
		s.prims32.resize((size_t)c->n_prims); s.prims64.resize((size_t)c->n_prims);
		HIP_TRY(c, hipMemcpy(s.nodes32.data(), c->blob32.dev, s.nodes32.size() * sizeof(Node<float>), hipMemcpyDeviceToHost));
		HIP_TRY(c, hipMemcpy(s.nodes64.data(), c->blob64.dev, s.nodes64.size() * sizeof(Node<double>), hipMemcpyDeviceToHost));
		HIP_TRY(c, hipMemcpy(s.prims32.data(), c->blob32.dev + c->blob32.prims_off, s.prims32.size() * sizeof(Prim<float>), hipMemcpyDeviceToHost));
		HIP_TRY(c, hipMemcpy(s.prims64.data(), c->blob64.dev + c->blob64.prims_off, s.prims64.size() * sizeof(Prim<double>), hipMemcpyDeviceToHost));
		if (c->blob32.qnodes_off) {
			s.qnodes.resize((size_t)c->n_nodes);
			HIP_TRY(c, hipMemcpy(s.qnodes.data(), c->blob32.dev + c->blob32.qnodes_off, s.qnodes.size() * sizeof(QNode), hipMemcpyDeviceToHost));
			s.qframe = c->blob32.qframe;
		}
		s.big_leaf = c->big_leaf; s.start = c->start; s.depth = c->tree_depth;
		std::string first;
		const int64_t bad = validate_flat_scene(s, c->n_objects, &first);
		if (violations) *violations = bad;
		if (n_nodes) *n_nodes = c->n_nodes;
		if (depth) *depth = c->tree_depth;
		if (bad) return fail(c, RTHIP_ERR_STATE, first);
		return RTHIP_OK;
	});
}

int32_t rthip_bvh_gate_count(const rthip_object* objects, int64_t n, int64_t* n_gates, int64_t* n_gated_spheres) {
	return guarded(nullptr, [&]() -> int32_t {
		if (n < 0 || (n > 0 && !objects)) return fail(nullptr, RTHIP_ERR_INVALID, "objects is NULL or n < 0");
		if (n == 0) return fail(nullptr, RTHIP_ERR_EMPTY, "empty scene: BVH::new returns None");
		FlatScene flat;
		std::string err;
		if (!build_flat_scene(objects, n, &flat, &err, host_only_options())) return fail(nullptr, flat.depth > kStackMax ? RTHIP_ERR_LIMIT : RTHIP_ERR_INVALID, err);
		int64_t gated = 0;
		for (const Prim<float>& p : flat.prims32)
			if (((p.id >> kKindShift) & 1) == RTHIP_SPHERE && prim_gate_count(p) != 0) gated++;
		if (n_gates) *n_gates = (int64_t)flat.gates32.size();
		if (n_gated_spheres) *n_gated_spheres = gated;
		return RTHIP_OK;
	});
}

int32_t rthip_scene_generate(const char* name, uint32_t seed, rthip_object* out, int64_t capacity, int64_t* n) {
	return guarded(nullptr, [&]() -> int32_t {
		if (!name || !n) return fail(nullptr, RTHIP_ERR_INVALID, "name or n is NULL");
		std::vector<rthip_object> objs;
		std::string err;
		if (!generate_scene(name, seed, &objs, &err)) return fail(nullptr, RTHIP_ERR_INVALID, err);
		*n = (int64_t)objs.size();
		if (out && capacity > 0) std::memcpy(out, objs.data(), sizeof(rthip_object) * (size_t)std::min<int64_t>(capacity, *n));
		return RTHIP_OK;
	});
}

void rthip_camera_rotation(double yaw_deg, double pitch_deg, double out[3]) { camera_rotation(yaw_deg, pitch_deg, out); }

}  // extern "C"
