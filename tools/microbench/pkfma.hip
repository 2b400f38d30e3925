// VALU issue rate of v_fma_f32 versus v_pk_fma_f32 on one GPU: eight independent chains per
// lane so that latency does not limit either.  Prints GFLOP/s and cycles per wave instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int kIters = 4096;
__global__ __launch_bounds__(256) void scalar_fma(float* out, float a, float b) {
	float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
	for (int i = 0; i < kIters; i++) {
		x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
		x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ __launch_bounds__(256) void packed_fma(float* out, float a, float b) {
	f2 va = {a, a}, vb = {b, b};
	float t = threadIdx.x;
	f2 x0 = {t, t + 1}, x1 = {t + 2, t + 3}, x2 = {t + 4, t + 5}, x3 = {t + 6, t + 7}, x4 = {t + 8, t + 9}, x5 = {t + 10, t + 11}, x6 = {t + 12, t + 13}, x7 = {t + 14, t + 15};
	for (int i = 0; i < kIters; i++) {
		x0 = __builtin_elementwise_fma(x0, va, vb); x1 = __builtin_elementwise_fma(x1, va, vb); x2 = __builtin_elementwise_fma(x2, va, vb); x3 = __builtin_elementwise_fma(x3, va, vb);
		x4 = __builtin_elementwise_fma(x4, va, vb); x5 = __builtin_elementwise_fma(x5, va, vb); x6 = __builtin_elementwise_fma(x6, va, vb); x7 = __builtin_elementwise_fma(x7, va, vb);
	}
	f2 s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
	out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
__global__ __launch_bounds__(256) void scalar_minmax(float* out, float a, float b) {
	float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
	for (int i = 0; i < kIters; i++) {
		x0 = __builtin_fmaxf(__builtin_fminf(x0, a), b); x1 = __builtin_fmaxf(__builtin_fminf(x1, a), b); x2 = __builtin_fmaxf(__builtin_fminf(x2, a), b); x3 = __builtin_fmaxf(__builtin_fminf(x3, a), b);
		x4 = __builtin_fmaxf(__builtin_fminf(x4, a), b); x5 = __builtin_fmaxf(__builtin_fminf(x5, a), b); x6 = __builtin_fmaxf(__builtin_fminf(x6, a), b); x7 = __builtin_fmaxf(__builtin_fminf(x7, a), b);
		a += 1.0f;
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
template <typename K> static float time_ms(K k, float* out, int grid) {
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, 1.0001f, 0.5f);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, 1.0001f, 0.5f);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms = 0; hipEventElapsedTime(&ms, e0, e1);
	return ms / 5;
}
int main() {
	hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
	const int cus = p.multiProcessorCount, grid = cus * 8 * 4;  // 8 waves per SIMD
	float* out; hipMalloc(&out, (size_t)grid * 256 * sizeof(float));
	const double waves = (double)grid * 4, clk = p.clockRate * 1e3;
	float ms = time_ms(scalar_fma, out, grid);
	double instr = waves * kIters * 8;
	printf("CUs %d clock %.0f MHz\n", cus, clk / 1e6);
	printf("v_fma_f32:    %.3f ms  %.1f TFLOP/s  %.2f cycles per wave instruction per SIMD\n", ms, instr * 64 * 2 / ms / 1e9, ms * 1e-3 * clk * cus * 4 / instr);
	ms = time_ms(packed_fma, out, grid);
	printf("v_pk_fma_f32: %.3f ms  %.1f TFLOP/s  %.2f cycles per wave instruction per SIMD\n", ms, instr * 64 * 4 / ms / 1e9, ms * 1e-3 * clk * cus * 4 / instr);
	ms = time_ms(scalar_minmax, out, grid);
	printf("v_min/v_max:  %.3f ms  %.2f cycles per wave instruction per SIMD\n", ms, ms * 1e-3 * clk * cus * 4 / (instr * 2));
	hipFree(out);
	return 0;
}
