// Issue cost of single VALU instructions on one GPU: eight independent registers per lane, one
// instruction each per loop iteration, 8 waves per SIMD.  Prints cycles per wave instruction per
// SIMD (4 = one instruction per 16-lane pass of a 64-wide wave).
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int kIters = 2048;
#define OP8(INS)                                                                                               \
	asm volatile(INS(%0) "\n" INS(%1) "\n" INS(%2) "\n" INS(%3) "\n" INS(%4) "\n" INS(%5) "\n" INS(%6) "\n" INS(%7) \
	             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b), "s"(m) : "v20", "v21", "v22", "v23", "vcc", "s20", "s21");
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
#define I_MULLO(r) "v_mul_lo_u32 " #r ", " #r ", %8"
#define I_XOR(r) "v_xor_b32 " #r ", " #r ", %8"
#define I_LSHR(r) "v_lshrrev_b32 " #r ", 3, " #r
#define I_LSHRV(r) "v_lshrrev_b32 " #r ", %8, " #r
#define I_RSQ(r) "v_rsq_f32 " #r ", " #r
#define I_CVTU(r) "v_cvt_f32_u32 " #r ", " #r
#define I_FMAC(r) "v_fmac_f32 " #r ", %8, %9"
#define I_FMAS(r) "v_fma_f32 " #r ", " #r ", s2, %9"
#define I_FMAABS(r) "v_fma_f32 " #r ", -" #r ", |%8|, %9"
#define I_CNDVCC(r) "v_cndmask_b32 " #r ", " #r ", %8, vcc"
#define I_CNDVCC64(r) "v_cndmask_b32_e64 " #r ", " #r ", %8, vcc"
#define I_CNDS(r) "v_cndmask_b32_e64 " #r ", " #r ", %8, s[20:21]"
#define I_CMPS(r) "v_cmp_lt_f32_e64 s[20:21], " #r ", %8"
#define I_ADDC(r) "v_addc_co_u32 " #r ", vcc, " #r ", %8, vcc"
#define I_MAD24(r) "v_mad_u32_u24 " #r ", " #r ", %8, %9"
#define I_BFE(r) "v_bfe_u32 " #r ", " #r ", 3, 5"
#define I_MULK(r) "v_mul_f32 " #r ", 0x3f800020, " #r
#define I_MED3(r) "v_med3_f32 " #r ", " #r ", %8, %9"
#define I_SUB(r) "v_sub_f32 " #r ", " #r ", %8"
#define I_MAD64(r) "v_mad_u64_u32 v[20:21], vcc, " #r ", %8, v[22:23]"
#define I_MADU32(r) "v_mad_u32_u24 " #r ", " #r ", %8, %9"
#define I_BFI(r) "v_bfi_b32 " #r ", %8, " #r ", %9"
#define I_TRUNC(r) "v_trunc_f32 " #r ", " #r "\n"
#define I_PERM(r) "v_perm_b32 " #r ", " #r ", %8, %9"
KERNEL(k_fma, I_FMA) KERNEL(k_mul, I_MUL) KERNEL(k_add, I_ADD) KERNEL(k_min, I_MIN) KERNEL(k_max3, I_MAX3) KERNEL(k_mov, I_MOV)
KERNEL(k_cnd, I_CND) KERNEL(k_iadd, I_IADD) KERNEL(k_and, I_AND) KERNEL(k_lshladd, I_LSHLADD) KERNEL(k_cmp, I_CMP) KERNEL(k_mini, I_MINI)
KERNEL(k_rcp, I_RCP) KERNEL(k_sqrt, I_SQRT) KERNEL(k_cvt, I_CVT)
KERNEL(k_mullo, I_MULLO) KERNEL(k_xor, I_XOR) KERNEL(k_lshr, I_LSHR) KERNEL(k_lshrv, I_LSHRV) KERNEL(k_rsq, I_RSQ) KERNEL(k_cvtu, I_CVTU) KERNEL(k_fmac, I_FMAC)
KERNEL(k_fmas, I_FMAS) KERNEL(k_fmaabs, I_FMAABS) KERNEL(k_cndvcc, I_CNDVCC) KERNEL(k_cndvcc64, I_CNDVCC64) KERNEL(k_cnds, I_CNDS) KERNEL(k_cmps, I_CMPS) KERNEL(k_mad24, I_MAD24) KERNEL(k_bfe, I_BFE) KERNEL(k_mulk, I_MULK) KERNEL(k_med3, I_MED3) KERNEL(k_sub, I_SUB) KERNEL(k_mad64, I_MAD64) KERNEL(k_bfi, I_BFI) KERNEL(k_perm, I_PERM)
int main() {
	hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
	const int cus = p.multiProcessorCount, grid = cus * 8 * 4;
	float* out; (void)hipMalloc(&out, (size_t)grid * 256 * sizeof(float));
	const double waves = (double)grid * 4, clk = p.clockRate * 1e3, instr = waves * kIters * 16;
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	struct { const char* name; void (*k)(float*, float, float); } ks[] = {
		{"v_fma_f32", k_fma}, {"v_mul_f32", k_mul}, {"v_add_f32", k_add}, {"v_min_f32", k_min}, {"v_max3_f32", k_max3}, {"v_mov_b32", k_mov},
		{"v_cndmask_b32", k_cnd}, {"v_add_u32", k_iadd}, {"v_and_b32", k_and}, {"v_lshl_add_u32", k_lshladd}, {"v_cmp_lt_f32", k_cmp}, {"v_min_i32", k_mini},
		{"v_rcp_f32", k_rcp}, {"v_sqrt_f32", k_sqrt}, {"v_cvt_f32_i32", k_cvt},
		{"v_mul_lo_u32", k_mullo}, {"v_xor_b32", k_xor}, {"v_lshrrev_b32 imm", k_lshr}, {"v_lshrrev_b32 v", k_lshrv}, {"v_rsq_f32", k_rsq}, {"v_cvt_f32_u32", k_cvtu}, {"v_fmac_f32", k_fmac},
		{"v_fma_f32 sgpr", k_fmas}, {"v_fma_f32 -a,|b|", k_fmaabs}, {"v_cndmask vcc", k_cndvcc}, {"v_cndmask_e64 vcc", k_cndvcc64}, {"v_cndmask_e64 s[20:21]", k_cnds}, {"v_cmp_lt_f32_e64 s[20:21]", k_cmps}, {"v_mad_u32_u24", k_mad24}, {"v_bfe_u32", k_bfe}, {"v_mul_f32 literal", k_mulk}, {"v_med3_f32", k_med3}, {"v_sub_f32", k_sub}, {"v_mad_u64_u32", k_mad64}, {"v_bfi_b32", k_bfi}, {"v_perm_b32", k_perm}};
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
