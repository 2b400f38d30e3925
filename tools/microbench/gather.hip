// How fast can a CU gather 64-byte records at random?  (a) every lane loads its own record with
// four 16-byte loads; (b) the four lanes of a quad load ONE record per instruction, 16 bytes
// each (one cache line per quad), four instructions for the quad's four records; (c) every lane
// loads a 32-byte record with two loads.  Prints records per second and per clock per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int kIters = 256;
__device__ __forceinline__ uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }
__global__ __launch_bounds__(256) void own64(const uint4* __restrict__ recs, uint32_t mask, uint32_t* out) {
	uint32_t s = blockIdx.x * 256u + threadIdx.x + 1u, acc = 0;
	for (int i = 0; i < kIters; i++) {
		const uint4* p = recs + (size_t)(lcg(s) & mask) * 4;
		uint4 a = p[0], b = p[1], c = p[2], d = p[3];
		acc ^= a.x ^ b.y ^ c.z ^ d.w;
	}
	out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void quad64(const uint4* __restrict__ recs, uint32_t mask, uint32_t* out) {
	uint32_t s = blockIdx.x * 256u + threadIdx.x + 1u, acc = 0;
	const uint32_t part = threadIdx.x & 3u;
	for (int i = 0; i < kIters; i++) {
		const int mine = (int)(lcg(s) & mask);
		const int i0 = __builtin_amdgcn_update_dpp(0, mine, 0x00, 0xf, 0xf, false);
		const int i1 = __builtin_amdgcn_update_dpp(0, mine, 0x55, 0xf, 0xf, false);
		const int i2 = __builtin_amdgcn_update_dpp(0, mine, 0xAA, 0xf, 0xf, false);
		const int i3 = __builtin_amdgcn_update_dpp(0, mine, 0xFF, 0xf, 0xf, false);
		uint4 a = recs[(size_t)i0 * 4 + part], b = recs[(size_t)i1 * 4 + part], c = recs[(size_t)i2 * 4 + part], d = recs[(size_t)i3 * 4 + part];
		acc ^= a.x ^ b.y ^ c.z ^ d.w;
	}
	out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void own32(const uint4* __restrict__ recs, uint32_t mask, uint32_t* out) {
	uint32_t s = blockIdx.x * 256u + threadIdx.x + 1u, acc = 0;
	for (int i = 0; i < kIters; i++) {
		const uint4* p = recs + (size_t)(lcg(s) & (mask * 2u + 1u)) * 2;
		uint4 a = p[0], b = p[1];
		acc ^= a.x ^ b.y;
	}
	out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <typename K> static void run(const char* name, K k, const uint4* recs, uint32_t mask, uint32_t* out, int grid, int cus, double clk) {
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, recs, mask, out);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, recs, mask, out);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
	double recs_n = (double)grid * 256 * kIters;
	printf("  %-34s %7.3f ms  %7.1f G records/s  %.2f records per clock per CU\n", name, ms, recs_n / ms / 1e6, recs_n / (ms * 1e-3) / clk / cus);
}
int main() {
	hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
	const int cus = p.multiProcessorCount, grid = cus * 6 * 4;  // 24 waves per CU
	const double clk = p.clockRate * 1e3;
	uint32_t* out; hipMalloc(&out, (size_t)grid * 256 * 4);
	for (uint32_t log2n : {10u, 16u, 19u, 22u}) {  // 64 KB, 4 MB, 32 MB, 256 MB of records
		const uint32_t n = 1u << log2n;
		uint4* recs; hipMalloc(&recs, (size_t)n * 64); hipMemset(recs, 1, (size_t)n * 64);
		printf("%u records of 64 bytes (%.1f MB):\n", n, n * 64 / 1e6);
		run("own record, 4 x 16 B per lane", own64, recs, n - 1, out, grid, cus, clk);
		run("one record per quad per load", quad64, recs, n - 1, out, grid, cus, clk);
		run("own 32-byte record, 2 x 16 B", own32, recs, n - 1, out, grid, cus, clk);
		hipFree(recs);
	}
	return 0;
}
