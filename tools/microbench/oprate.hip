// Issue cost of single VALU instructions on one GPU: eight independent registers per lane, one
// instruction each per loop iteration, 8 waves per SIMD.  Prints cycles per wave instruction per
// SIMD (4 = one instruction per 16-lane pass of a 64-wide wave).
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int kIters = 2048;
#define OP8(INS)                                                                                               \
	asm volatile(INS(%0) "\n" INS(%1) "\n" INS(%2) "\n" INS(%3) "\n" INS(%4) "\n" INS(%5) "\n" INS(%6) "\n" INS(%7) \
	             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b), "s"(m));
#define KERNEL(NAME, INS)                                                                     \
	__global__ __launch_bounds__(256) void NAME(float* out, float a, float b) {                \
		float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
		unsigned long long m = 0x5555555555555555ull;                                           \
		for (int i = 0; i < kIters; i++) { OP8(INS) OP8(INS) }                                  \
		out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;     \
	}
#define I_FMA(r) "v_fma_f32 " #r ", " #r ", %8, %9"
#define I_MUL(r) "v_mul_f32 " #r ", " #r ", %8"
#define I_ADD(r) "v_add_f32 " #r ", " #r ", %8"
#define I_MIN(r) "v_min_f32 " #r ", " #r ", %8"
#define I_MAX3(r) "v_max3_f32 " #r ", " #r ", %8, %9"
#define I_MOV(r) "v_mov_b32 " #r ", %8"
#define I_CND(r) "v_cndmask_b32 " #r ", " #r ", %8, %10"
#define I_IADD(r) "v_add_u32 " #r ", " #r ", %8"
#define I_AND(r) "v_and_b32 " #r ", " #r ", %8"
#define I_LSHLADD(r) "v_lshl_add_u32 " #r ", " #r ", 2, %8"
#define I_CMP(r) "v_cmp_lt_f32 vcc, " #r ", %8"
#define I_MINI(r) "v_min_i32 " #r ", " #r ", %8"
#define I_RCP(r) "v_rcp_f32 " #r ", " #r
#define I_SQRT(r) "v_sqrt_f32 " #r ", " #r
#define I_CVT(r) "v_cvt_f32_i32 " #r ", " #r
#define I_PKFMA(r) "v_pk_mul_f32 %0, %0, %1"
KERNEL(k_fma, I_FMA) KERNEL(k_mul, I_MUL) KERNEL(k_add, I_ADD) KERNEL(k_min, I_MIN) KERNEL(k_max3, I_MAX3) KERNEL(k_mov, I_MOV)
KERNEL(k_cnd, I_CND) KERNEL(k_iadd, I_IADD) KERNEL(k_and, I_AND) KERNEL(k_lshladd, I_LSHLADD) KERNEL(k_cmp, I_CMP) KERNEL(k_mini, I_MINI)
KERNEL(k_rcp, I_RCP) KERNEL(k_sqrt, I_SQRT) KERNEL(k_cvt, I_CVT)
int main() {
	hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
	const int cus = p.multiProcessorCount, grid = cus * 8 * 4;
	float* out; (void)hipMalloc(&out, (size_t)grid * 256 * sizeof(float));
	const double waves = (double)grid * 4, clk = p.clockRate * 1e3, instr = waves * kIters * 16;
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	struct { const char* name; void (*k)(float*, float, float); } ks[] = {
		{"v_fma_f32", k_fma}, {"v_mul_f32", k_mul}, {"v_add_f32", k_add}, {"v_min_f32", k_min}, {"v_max3_f32", k_max3}, {"v_mov_b32", k_mov},
		{"v_cndmask_b32", k_cnd}, {"v_add_u32", k_iadd}, {"v_and_b32", k_and}, {"v_lshl_add_u32", k_lshladd}, {"v_cmp_lt_f32", k_cmp}, {"v_min_i32", k_mini},
		{"v_rcp_f32", k_rcp}, {"v_sqrt_f32", k_sqrt}, {"v_cvt_f32_i32", k_cvt}};
	for (auto& c : ks) {
		hipLaunchKernelGGL(c.k, dim3(grid), dim3(256), 0, 0, out, 1.0001f, 0.5f);
		(void)hipDeviceSynchronize();
		(void)hipEventRecord(e0);
		for (int r = 0; r < 5; r++) hipLaunchKernelGGL(c.k, dim3(grid), dim3(256), 0, 0, out, 1.0001f, 0.5f);
		(void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
		float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
		printf("%-16s %.3f ms  %.2f cycles per wave instruction per SIMD\n", c.name, ms, ms * 1e-3 * clk * cus * 4 / instr);
	}
	(void)hipFree(out);
	return 0;
}
