// Does a VALU instruction of a wave with few active lanes cost less?  The same FMA chains as
// pkfma.hip, executed under an exec mask given as a 64-bit pattern (build with
// -fno-slp-vectorize, or the chains pair up into packed FMAs).
// Measured on MI355X: no.  Inside one loop, blocks of eight FMAs under masks of 64, 32, 28 ... 8
// lanes, dense or spread over the four 16-lane rows, all cost 4.6 cycles per wave instruction:
// the SIMD skips nothing, so only the lanes that are filled count.  (A kernel whose WHOLE loop
// runs under a mask of 16 lanes or fewer measures 12 cycles per instruction, one of 32 or more
// lanes 4.3: a clock or power-state effect of a nearly idle chip, not a cost per instruction,
// as the mixed cases show.)
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int kIters = 4096;
__global__ __launch_bounds__(256) void masked_fma(float* out, float a, float b, unsigned long long mask) {
	float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
	if ((mask >> (threadIdx.x & 63u)) & 1ull) {
		for (int i = 0; i < kIters; i++) {
			x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
			x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
// Each iteration runs the eight FMAs once under mask_a and once under mask_b.
__global__ __launch_bounds__(256) void mixed_fma(float* out, float a, float b, unsigned long long mask_a, unsigned long long mask_b) {
	float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
	const bool in_a = (mask_a >> (threadIdx.x & 63u)) & 1ull, in_b = (mask_b >> (threadIdx.x & 63u)) & 1ull;
	for (int i = 0; i < kIters / 2; i++) {
		if (in_a) {
			x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
			x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
		}
		asm volatile("" ::: "memory");
		if (in_b) {
			x0 = __builtin_fmaf(x0, b, a); x1 = __builtin_fmaf(x1, b, a); x2 = __builtin_fmaf(x2, b, a); x3 = __builtin_fmaf(x3, b, a);
			x4 = __builtin_fmaf(x4, b, a); x5 = __builtin_fmaf(x5, b, a); x6 = __builtin_fmaf(x6, b, a); x7 = __builtin_fmaf(x7, b, a);
		}
		asm volatile("" ::: "memory");
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
int main() {
	hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
	const int cus = p.multiProcessorCount, grid = cus * 8 * 4;  // 8 waves per SIMD
	float* out; hipMalloc(&out, (size_t)grid * 256 * sizeof(float));
	const double waves = (double)grid * 4, clk = p.clockRate * 1e3, instr = waves * kIters * 8;
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	struct { const char* name; unsigned long long mask; } cases[] = {
		{"all 64 lanes", ~0ull}, {"low 32 lanes", 0xffffffffull}, {"high 32 lanes", 0xffffffff00000000ull}, {"low 16 lanes", 0xffffull},
		{"lanes 16-31", 0xffff0000ull}, {"every 4th lane", 0x1111111111111111ull}, {"every 2nd lane", 0x5555555555555555ull}, {"one lane", 1ull}, {"lane 37", 1ull << 37}, {"two lanes", 3ull}, {"lanes 0 and 40", 1ull | (1ull << 40)}, {"four lanes", 0xfull}, {"eight lanes", 0xffull}, {"all 64 again", ~0ull}};
	for (auto& c : cases) {
		hipLaunchKernelGGL(masked_fma, dim3(grid), dim3(256), 0, 0, out, 1.0001f, 0.5f, c.mask);
		hipDeviceSynchronize();
		hipEventRecord(e0);
		for (int r = 0; r < 5; r++) hipLaunchKernelGGL(masked_fma, dim3(grid), dim3(256), 0, 0, out, 1.0001f, 0.5f, c.mask);
		hipEventRecord(e1); hipEventSynchronize(e1);
		float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
		printf("%-16s %.3f ms  %.2f cycles per wave instruction per SIMD\n", c.name, ms, ms * 1e-3 * clk * cus * 4 / instr);
	}
	struct { const char* name; unsigned long long a, b; } mixes[] = {
		{"64 then 64", ~0ull, ~0ull}, {"64 then 8", ~0ull, 0xffull}, {"8 then 8", 0xffull, 0xffull}, {"32 then 8", 0xffffffffull, 0xffull},
		{"17 lanes x2", 0x1ffffull, 0x1ffffull}, {"20 lanes x2", 0xfffffull, 0xfffffull}, {"24 lanes x2", 0xffffffull, 0xffffffull}, {"28 lanes x2", 0xfffffffull, 0xfffffffull},
		{"16 spread over 4 rows", 0x000f000f000f000full, 0x000f000f000f000full}, {"low16 then high16", 0xffffull, 0xffff000000000000ull}};
	for (auto& c : mixes) {
		hipLaunchKernelGGL(mixed_fma, dim3(grid), dim3(256), 0, 0, out, 1.0001f, 0.5f, c.a, c.b);
		hipDeviceSynchronize();
		hipEventRecord(e0);
		for (int r = 0; r < 5; r++) hipLaunchKernelGGL(mixed_fma, dim3(grid), dim3(256), 0, 0, out, 1.0001f, 0.5f, c.a, c.b);
		hipEventRecord(e1); hipEventSynchronize(e1);
		float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
		printf("mixed %-22s %.3f ms  %.2f cycles per wave instruction per SIMD\n", c.name, ms, ms * 1e-3 * clk * cus * 4 / instr);
	}
	hipFree(out);
	return 0;
}
