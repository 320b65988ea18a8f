// What does a VALU instruction cost on gfx950? Throughput (8 waves per SIMD, independent chains) and one wave alone per SIMD
// (dependent chain = latency). cycles per wave-instruction per SIMD, from wall time at the measured shader clock.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; usage: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#define REP 64
#define STR2(x) #x
#define STR(x) STR2(x)
// OP: an asm string using %0..%3 (four independent v2f / float accumulators) and %4, %5 inputs
#define KERNEL(name, TYPE, CONSTR, BODY) KERNELC(name, TYPE, CONSTR, BODY, "memory")
#define KERNELC(name, TYPE, CONSTR, BODY, ...)                                                          \
__global__ void __launch_bounds__(256) name(int iters, float *sink)                              \
{                                                                                                 \
	TYPE a0 = (TYPE)threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, x = a0 * 0.5f, y = a0 * 0.25f; \
	for (int i = 0; i < iters; i++)                                                               \
	{                                                                                             \
		_Pragma("unroll") for (int r = 0; r < REP / 4; r++)                                       \
			asm volatile(BODY : "+" CONSTR(a0), "+" CONSTR(a1), "+" CONSTR(a2), "+" CONSTR(a3) : CONSTR(x), CONSTR(y) : __VA_ARGS__); \
	}                                                                                             \
	TYPE s = a0 + a1 + a2 + a3;                                                                   \
	if (*(float *)&s == 1.2345f) *sink = *(float *)&s;                                            \
}
typedef float v2f __attribute__((ext_vector_type(2)));
KERNEL(k_fma, float, "v", "v_fma_f32 %0, %4, %5, %0\nv_fma_f32 %1, %4, %5, %1\nv_fma_f32 %2, %4, %5, %2\nv_fma_f32 %3, %4, %5, %3")
KERNEL(k_mul, float, "v", "v_mul_f32 %0, %4, %0\nv_mul_f32 %1, %4, %1\nv_mul_f32 %2, %4, %2\nv_mul_f32 %3, %4, %3")
KERNEL(k_pkfma, v2f, "v", "v_pk_fma_f32 %0, %4, %5, %0\nv_pk_fma_f32 %1, %4, %5, %1\nv_pk_fma_f32 %2, %4, %5, %2\nv_pk_fma_f32 %3, %4, %5, %3")
KERNEL(k_pkmul, v2f, "v", "v_pk_mul_f32 %0, %4, %0\nv_pk_mul_f32 %1, %4, %1\nv_pk_mul_f32 %2, %4, %2\nv_pk_mul_f32 %3, %4, %3")
KERNEL(k_pkadd, v2f, "v", "v_pk_add_f32 %0, %4, %0\nv_pk_add_f32 %1, %4, %1\nv_pk_add_f32 %2, %4, %2\nv_pk_add_f32 %3, %4, %3")
KERNEL(k_exp, float, "v", "v_exp_f32 %0, %0\nv_exp_f32 %1, %1\nv_exp_f32 %2, %2\nv_exp_f32 %3, %3")
KERNEL(k_cndmask, float, "v", "v_cndmask_b32 %0, %4, %0, vcc\nv_cndmask_b32 %1, %4, %1, vcc\nv_cndmask_b32 %2, %4, %2, vcc\nv_cndmask_b32 %3, %4, %3, vcc")
KERNELC(k_cmp, float, "v", "v_cmp_lt_f32 vcc, %4, %0\nv_cmp_lt_f32 vcc, %4, %1\nv_cmp_lt_f32 vcc, %4, %2\nv_cmp_lt_f32 vcc, %4, %3", "vcc")
KERNELC(k_cmp_s, float, "v", "v_cmp_lt_f32 s[20:21], %4, %0\nv_cmp_lt_f32 s[22:23], %4, %1\nv_cmp_lt_f32 s[24:25], %4, %2\nv_cmp_lt_f32 s[26:27], %4, %3", "s20","s21","s22","s23","s24","s25","s26","s27")
KERNEL(k_min, float, "v", "v_min_f32 %0, %4, %0\nv_min_f32 %1, %4, %1\nv_min_f32 %2, %4, %2\nv_min_f32 %3, %4, %3")
KERNEL(k_mov, float, "v", "v_mov_b32 %0, %4\nv_mov_b32 %1, %4\nv_mov_b32 %2, %4\nv_mov_b32 %3, %4")
KERNEL(k_and, float, "v", "v_and_b32 %0, %4, %0\nv_and_b32 %1, %4, %1\nv_and_b32 %2, %4, %2\nv_and_b32 %3, %4, %3")
KERNEL(k_sub, float, "v", "v_sub_f32 %0, %4, %0\nv_sub_f32 %1, %4, %1\nv_sub_f32 %2, %4, %2\nv_sub_f32 %3, %4, %3")
KERNEL(k_fmac, float, "v", "v_fmac_f32 %0, %4, %5\nv_fmac_f32 %1, %4, %5\nv_fmac_f32 %2, %4, %5\nv_fmac_f32 %3, %4, %5")
KERNEL(k_dep_fma, float, "v", "v_fma_f32 %0, %4, %5, %0\nv_fma_f32 %0, %4, %5, %0\nv_fma_f32 %0, %4, %5, %0\nv_fma_f32 %0, %4, %5, %0")
KERNEL(k_dep_pkfma, v2f, "v", "v_pk_fma_f32 %0, %4, %5, %0\nv_pk_fma_f32 %0, %4, %5, %0\nv_pk_fma_f32 %0, %4, %5, %0\nv_pk_fma_f32 %0, %4, %5, %0")
KERNEL(k_dep_exp, float, "v", "v_exp_f32 %0, %0\nv_exp_f32 %0, %0\nv_exp_f32 %0, %0\nv_exp_f32 %0, %0")
KERNELC(k_salu, float, "v", "s_add_u32 s20, s20, 1\ns_add_u32 s21, s21, 1\ns_add_u32 s22, s22, 1\ns_add_u32 s23, s23, 1", "s20","s21","s22","s23","scc")
KERNELC(k_mix_valu_salu, float, "v", "v_fma_f32 %0, %4, %5, %0\ns_add_u32 s20, s20, 1\nv_fma_f32 %1, %4, %5, %1\ns_add_u32 s21, s21, 1", "s20","s21","scc")
KERNELC(k_dsread, float, "v", "ds_read_b128 v[40:43], %0\nds_read_b128 v[44:47], %0\nds_read_b128 v[48:51], %0\nds_read_b128 v[52:55], %0\ns_waitcnt lgkmcnt(0)", "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55")

// does a wave64 VALU instruction with one 32-lane half of EXEC empty take one pass instead of two?
#define KERNEL_HALF(name, BODY, EXECSET)                                                          \
__global__ void __launch_bounds__(256) name(int iters, float *sink)                              \
{                                                                                                 \
	float a0 = (float)threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, x = a0 * 0.5f, y = a0 * 0.25f; \
	asm volatile(EXECSET ::: "exec");                                                             \
	for (int i = 0; i < iters; i++)                                                               \
	{                                                                                             \
		_Pragma("unroll") for (int r = 0; r < REP / 4; r++)                                       \
			asm volatile(BODY : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y));          \
	}                                                                                             \
	asm volatile("s_mov_b64 exec, -1" ::: "exec");                                                \
	float s = a0 + a1 + a2 + a3;                                                                  \
	if (s == 1.2345f) *sink = s;                                                                  \
}
KERNEL_HALF(k_fma_lo32, "v_fma_f32 %0, %4, %5, %0\nv_fma_f32 %1, %4, %5, %1\nv_fma_f32 %2, %4, %5, %2\nv_fma_f32 %3, %4, %5, %3", "s_mov_b32 exec_hi, 0")
KERNEL_HALF(k_fma_hi32, "v_fma_f32 %0, %4, %5, %0\nv_fma_f32 %1, %4, %5, %1\nv_fma_f32 %2, %4, %5, %2\nv_fma_f32 %3, %4, %5, %3", "s_mov_b32 exec_lo, 0")
KERNEL_HALF(k_fma_even16, "v_fma_f32 %0, %4, %5, %0\nv_fma_f32 %1, %4, %5, %1\nv_fma_f32 %2, %4, %5, %2\nv_fma_f32 %3, %4, %5, %3", "s_mov_b32 exec_lo, 0xffff\ns_mov_b32 exec_hi, 0xffff")
KERNEL_HALF(k_exp_lo32, "v_exp_f32 %0, %0\nv_exp_f32 %1, %1\nv_exp_f32 %2, %2\nv_exp_f32 %3, %3", "s_mov_b32 exec_hi, 0")
KERNEL_HALF(k_cmp_lo32, "v_cmp_lt_f32 vcc, %4, %0\nv_cmp_lt_f32 vcc, %4, %1\nv_cmp_lt_f32 vcc, %4, %2\nv_cmp_lt_f32 vcc, %4, %3", "s_mov_b32 exec_hi, 0")

template <typename K> static void run(const char *name, K k, int waves_per_simd, int iters, float *sink, double ghz)
{
	const int blocks = 256 * waves_per_simd; // 256-thread workgroups = 4 waves = one per SIMD
	for (int rep = 0; rep < 3; rep++)
	{
		hipDeviceSynchronize();
		auto t0 = std::chrono::high_resolution_clock::now();
		hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, iters, sink);
		hipDeviceSynchronize();
		const double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count();
		if (rep == 2)
		{
			const double instr_per_simd = (double)iters * REP * waves_per_simd;
			printf("%-18s %d waves/SIMD: %9.1f us, %6.2f cycles per wave-instruction per SIMD (at %.2f GHz)\n", name, waves_per_simd, us, us * 1e-6 * ghz * 1e9 / instr_per_simd, ghz);
		}
	}
}
int main()
{
	float *sink; hipMalloc(&sink, 64);
	int khz = 0; hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
	const double ghz = khz * 1e-6;
	const int it = 20000;
#define R(k) run(#k, k, 8, it, sink, ghz); run(#k, k, 1, it * 4, sink, ghz);
	R(k_fma) R(k_mul) R(k_fmac) R(k_sub) R(k_min) R(k_mov) R(k_and) R(k_pkfma) R(k_pkmul) R(k_pkadd) R(k_exp) R(k_cndmask) R(k_cmp) R(k_cmp_s)
	R(k_fma_lo32) R(k_fma_hi32) R(k_fma_even16) R(k_exp_lo32) R(k_cmp_lo32)
	R(k_dep_fma) R(k_dep_pkfma) R(k_dep_exp) R(k_salu) R(k_mix_valu_salu) R(k_dsread)
	return 0;
}
