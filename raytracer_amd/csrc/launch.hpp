// launch.hpp — host-callable launchers, one set per arithmetic type (each defined in its own
// translation unit so that f64 can be built with -ffp-contract=off and f32 without).
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.hpp"
#include "wavefront.hpp"

namespace rthip {

struct LaunchCfg {
	int device;          // HIP device the launch goes to
	int grid;            // workgroups
	int threads;         // threads per workgroup: 256, 512, 768 or 1024
	size_t dyn_lds;      // bytes of dynamic LDS
	bool scene_in_lds;
	int lds_mode;        // render kernel: 0 = scene in global memory, 1 = nodes / primitives / gates in LDS, 2 = materials too
	bool count;          // instrumented (box / sphere / cube test counters)
	bool quantised;      // f32, scene in global memory: traverse the 32-byte quantised nodes
	uint32_t acc_tile_off;  // acc_lds: where the tiles start in the workgroup's dynamic LDS
	bool acc_lds;        // f32, lds_mode 2: samples go to a per-wave tile in LDS, flushed once per work item
	bool root_only;      // with acc_lds: the scene is one leaf, the kernel without traversal code
	bool stack_mem;      // f32, scene in LDS: the lanes' traversal stacks in global memory instead of LDS
	bool xchg;           // f32, scene in LDS: rays waiting to be shaded are regrouped by shading arm through queues in LDS
};

hipError_t launch_render(const RenderArgs<float>& a, const LaunchCfg& cfg, hipStream_t s);
hipError_t launch_render(const RenderArgs<double>& a, const LaunchCfg& cfg, hipStream_t s);
hipError_t launch_probe(const ProbeArgs<float>& a, size_t dyn_lds, hipStream_t s);
hipError_t launch_probe(const ProbeArgs<double>& a, size_t dyn_lds, hipStream_t s);
// Resident workgroups per CU for the render kernel at this much dynamic LDS.
hipError_t render_blocks_per_cu_f32(int threads, int lds_mode, size_t dyn_lds, int* blocks, bool quantised = false, bool acc_lds = false, bool stack_mem = false, bool xchg = false,
                                    bool root_only = false);
hipError_t render_blocks_per_cu_f64(int threads, int lds_mode, size_t dyn_lds, int* blocks);
// Accumulation buffer (tile-major fixed point) -> row-major [H][W][3] doubles (sums).
hipError_t launch_unpack(const unsigned long long* accum, double* rgb, int width, int height, int tiles_x, hipStream_t s);
// raytracer_headless/src/main.rs:6-12,55: (accum / spp * 255.0) as u8.  xrgb = 0: RGB8 bytes
// [H][W][3]; xrgb = 1: one 0x00RRGGBB word per pixel (raytracer_runtime/src/main.rs:15-17).
hipError_t launch_finalize(const unsigned long long* accum, void* out, int width, int height, int tiles_x, unsigned spp, int xrgb, hipStream_t s);


// dst[i] += src[i] over n 64-bit words (the merge of two accumulation buffers on one GPU).
hipError_t launch_add_words(unsigned long long* dst, const unsigned long long* src, uint64_t n, hipStream_t s);

// A small array in static device memory (file scope of kernels_f32.hip), for the entry point that
// must not allocate: it holds the control block of a launch (statistics and work-queue
// counters), which every other entry point keeps in memory allocated at rthip_create.  Each GPU
// has its own copy at its own address: `device` must be the current device; the address is
// looked up once per device and kept.
constexpr size_t kStaticScratchBytes = 4096;
hipError_t static_scratch_for_device(int device, void** ptr);

// Wavefront variant: one launcher per stage.  `grid` = workgroups of kWfStreamThreads for the
// streaming stages; extend is configured like the megakernel (threads, dynamic LDS, scene in LDS).
hipError_t launch_wf_generate(const WfArgs<float>& a, int grid, hipStream_t s);
hipError_t launch_wf_generate(const WfArgs<double>& a, int grid, hipStream_t s);
hipError_t launch_wf_extend(const WfArgs<float>& a, const LaunchCfg& cfg, hipStream_t s);
hipError_t launch_wf_extend(const WfArgs<double>& a, const LaunchCfg& cfg, hipStream_t s);
hipError_t launch_wf_shade(const WfArgs<float>& a, int grid, hipStream_t s);
hipError_t launch_wf_shade(const WfArgs<double>& a, int grid, hipStream_t s);
hipError_t wf_extend_blocks_per_cu_f32(int threads, bool scene_in_lds, size_t dyn_lds, int* blocks);
hipError_t wf_extend_blocks_per_cu_f64(int threads, bool scene_in_lds, size_t dyn_lds, int* blocks);

}  // namespace rthip
