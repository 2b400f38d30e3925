// kernels_f32.hip — production instantiation: f32 VALU arithmetic, fp contraction allowed.
#include <algorithm>
#include <mutex>

#include "launch.hpp"
#define RT_REAL float
#include "kernels_impl.inc"

namespace rthip {

__device__ __forceinline__ size_t accum_offset(uint32_t x, uint32_t y, int tiles_x) {
	uint32_t tile = (y / kTile) * (uint32_t)tiles_x + (x / kTile);
	return (size_t)tile * (kTilePixels * kAccumChannels) + (y % kTile) * kTile + (x % kTile);  // + channel * kTilePixels
}
__global__ void unpack_accum_kernel(const unsigned long long* accum, double* rgb, int width, int height, int tiles_x) {
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= (uint32_t)width * (uint32_t)height) return;
	size_t src = accum_offset(i % (uint32_t)width, i / (uint32_t)width, tiles_x);
	for (int c = 0; c < 3; c++) rgb[(size_t)i * 3 + c] = (double)accum[src + c * kTilePixels] * (1.0 / kAccumScale);
}
// `(v * 255.0) as u8` of Rust: truncation toward zero, saturating, NaN -> 0.
__device__ __forceinline__ uint32_t quantize(unsigned long long fixed, double spp) {
	double v = ((double)fixed * (1.0 / kAccumScale)) / spp * 255.0;
	if (!(v > 0.0)) return 0u;
	if (v >= 255.0) return 255u;
	return (uint32_t)v;
}
__global__ void finalize_kernel(const unsigned long long* accum, void* out, int width, int height, int tiles_x, unsigned spp, int xrgb) {
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= (uint32_t)width * (uint32_t)height) return;
	size_t src = accum_offset(i % (uint32_t)width, i / (uint32_t)width, tiles_x);
	uint32_t r = quantize(accum[src], (double)spp), g = quantize(accum[src + kTilePixels], (double)spp), b = quantize(accum[src + 2 * kTilePixels], (double)spp);
	if (xrgb) {
		static_cast<uint32_t*>(out)[i] = (r << 16) | (g << 8) | b;
	} else {
		uint8_t* o = static_cast<uint8_t*>(out) + (size_t)i * 3;
		o[0] = (uint8_t)r; o[1] = (uint8_t)g; o[2] = (uint8_t)b;
	}
}
hipError_t launch_finalize(const unsigned long long* accum, void* out, int width, int height, int tiles_x, unsigned spp, int xrgb, hipStream_t s) {
	unsigned n = (unsigned)width * (unsigned)height;
	hipLaunchKernelGGL(finalize_kernel, dim3((n + 255) / 256), dim3(256), 0, s, accum, out, width, height, tiles_x, spp, xrgb);
	return hipGetLastError();
}

hipError_t wf_extend_blocks_per_cu_f32(int threads, bool scene_in_lds, size_t dyn_lds, int* blocks) { return wf_extend_blocks_per_cu_impl(threads, scene_in_lds, dyn_lds, blocks); }
hipError_t render_blocks_per_cu_f32(int threads, int lds_mode, size_t dyn_lds, int* blocks, bool quantised, bool acc_lds, bool stack_mem, bool xchg, bool root_only) {
	return blocks_per_cu_impl(threads, lds_mode, dyn_lds, blocks, quantised, acc_lds, stack_mem, xchg, root_only);
}

// Streams both buffers once: 24 bytes of HBM traffic per word, two words per thread and trip.
__global__ __launch_bounds__(256) void add_words_kernel(unsigned long long* __restrict__ dst, const unsigned long long* __restrict__ src, uint64_t n) {
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	const uint64_t pairs = n / 2;
	ulonglong2* d2 = reinterpret_cast<ulonglong2*>(dst);
	const ulonglong2* s2 = reinterpret_cast<const ulonglong2*>(src);
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += stride) {
		ulonglong2 a = d2[i];
		const ulonglong2 b = s2[i];
		a.x += b.x; a.y += b.y;
		d2[i] = a;
	}
	if (blockIdx.x == 0 && threadIdx.x == 0 && (n & 1)) dst[n - 1] += src[n - 1];
}
hipError_t launch_add_words(unsigned long long* dst, const unsigned long long* src, uint64_t n, hipStream_t s) {
	if (n == 0) return hipSuccess;
	const bool aligned = (reinterpret_cast<uintptr_t>(dst) % 16 == 0) && (reinterpret_cast<uintptr_t>(src) % 16 == 0);
	if (!aligned) return hipErrorInvalidValue;  // accumulation buffers come from hipMalloc or a tensor allocator: 256-byte aligned
	const unsigned grid = (unsigned)std::min<uint64_t>((n / 2 + 255) / 256 + 1, 256u * 8u);
	hipLaunchKernelGGL(add_words_kernel, dim3(grid), dim3(256), 0, s, dst, src, n);
	return hipGetLastError();
}

// Static scratch (see launch.hpp).  In host code the array's name is only a handle for
// hipGetSymbolAddress; kernels and memsets get the per-device address looked up here.
static __device__ __attribute__((used)) __align__(256) unsigned char scratch[kStaticScratchBytes];
hipError_t static_scratch_for_device(int device, void** ptr) {
	constexpr int kMaxDevices = 64;
	static std::mutex mu;
	static void* table[kMaxDevices] = {};
	if (!ptr || device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
	std::lock_guard<std::mutex> lock(mu);
	if (!table[device]) {
		int current = -1;
		hipError_t e = hipGetDevice(&current);
		if (e != hipSuccess) return e;
		if (current != device) return hipErrorInvalidDevice;  // the lookup answers for the current device
		void* p = nullptr;
		e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
		if (e != hipSuccess) return e;
		table[device] = p;
	}
	*ptr = table[device];
	return hipSuccess;
}

hipError_t launch_unpack(const unsigned long long* accum, double* rgb, int width, int height, int tiles_x, hipStream_t s) {
	unsigned n = (unsigned)width * (unsigned)height;
	hipLaunchKernelGGL(unpack_accum_kernel, dim3((n + 255) / 256), dim3(256), 0, s, accum, rgb, width, height, tiles_x);
	return hipGetLastError();
}
}  // namespace rthip
