// Micro-benchmark: issue cost of single VALU instructions on gfx950, 16 independent chains per wave,
// 8 waves per SIMD, operands in vector registers unless the name says otherwise.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CASES(X) \
	X(0, "v_add_u32 v,v", "v_add_u32 %0, %1, %0", "v") \
	X(1, "v_add_u32 s,v", "v_add_u32 %0, %2, %0", "v") \
	X(2, "v_xor_b32 v,v", "v_xor_b32 %0, %1, %0", "v") \
	X(3, "v_sub_u32 v,v", "v_sub_u32 %0, %1, %0", "v") \
	X(4, "v_max_i32 v,v", "v_max_i32 %0, %1, %0", "v") \
	X(5, "v_lshrrev_b32 v,v", "v_lshrrev_b32 %0, %3, %0", "v") \
	X(6, "v_lshrrev_b32 s,v", "v_lshrrev_b32 %0, %2, %0", "v") \
	X(7, "v_lshrrev_b32 5,v", "v_lshrrev_b32 %0, 5, %0", "v") \
	X(8, "v_ashrrev_i32 5,v", "v_ashrrev_i32 %0, 5, %0", "v") \
	X(9, "v_dot2c_i32_i16 v,v", "v_dot2c_i32_i16 %0, %1, %3", "v") \
	X(10, "v_dot2c_i32_i16 s,v", "v_dot2c_i32_i16 %0, %2, %1", "v") \
	X(11, "v_dot2_i32_i16 v,v,v", "v_dot2_i32_i16 %0, %1, %3, %0", "v") \
	X(12, "v_mad_i32_i24 v,v,v", "v_mad_i32_i24 %0, %1, %3, %0", "v") \
	X(13, "v_mad_u32_u24 v,v,v", "v_mad_u32_u24 %0, %1, %3, %0", "v") \
	X(14, "v_mul_i32_i24 v,v", "v_mul_i32_i24 %0, %1, %0", "v") \
	X(15, "v_mul_lo_u32 v,v", "v_mul_lo_u32 %0, %1, %0", "v") \
	X(16, "v_sad_u32 v,v,v", "v_sad_u32 %0, %1, %3, %0", "v") \
	X(17, "v_add3_u32 v,v,v", "v_add3_u32 %0, %1, %3, %0", "v") \
	X(18, "v_lshl_add_u32 v,2,v", "v_lshl_add_u32 %0, %1, 2, %0", "v") \
	X(19, "v_add_lshl_u32 v,v,2", "v_add_lshl_u32 %0, %1, %0, 2", "v") \
	X(20, "v_and_or_b32 v,v,v", "v_and_or_b32 %0, %1, %3, %0", "v") \
	X(21, "v_perm_b32 v,v,v", "v_perm_b32 %0, %1, %0, %3", "v") \
	X(22, "v_alignbit_b32 v,v,v", "v_alignbit_b32 %0, %1, %0, %3", "v") \
	X(23, "v_bfe_u32 v,v,v", "v_bfe_u32 %0, %0, %1, %3", "v") \
	X(24, "v_bfe_i32 v,0,16", "v_bfe_i32 %0, %0, 0, 16", "v") \
	X(25, "v_cndmask_b32 v,v,vcc", "v_cndmask_b32 %0, %1, %0, vcc", "v") \
	X(26, "v_fma_f32 v,v,v", "v_fma_f32 %0, %1, %3, %0", "v") \
	X(27, "v_mul_f32 v,v", "v_mul_f32 %0, %1, %0", "v") \
	X(28, "v_add_f32 v,v", "v_add_f32 %0, %1, %0", "v") \
	X(29, "v_mov_b32 v", "v_mov_b32 %0, %1", "v") \
	X(30, "v_readfirstlane_b32", "v_readfirstlane_b32 %4, %0", "v") \
	X(31, "v_ffbh_u32 v", "v_ffbh_u32 %0, %0", "v") \
	X(32, "v_min_u32 v,v", "v_min_u32 %0, %1, %0", "v") \
	X(33, "v_mad_u64_u32", "v_mad_u64_u32 %5, vcc, %1, %3, %5", "v") \
	X(34, "v_lshlrev_b64 v", "v_lshlrev_b64 %5, 3, %5", "v") \
	X(35, "v_add_co_u32 v,v", "v_add_co_u32 %0, vcc, %1, %0", "v") \
	X(36, "v_pk_add_u16 v,v", "v_pk_add_u16 %0, %1, %0", "v") \
	X(37, "v_pk_mad_i16 v,v,v", "v_pk_mad_i16 %0, %1, %3, %0", "v") \
	X(38, "v_pk_mul_f32 v,v", "v_pk_mul_f32 %5, %5, %5", "v") \
	X(39, "v_pk_add_f32 v,v", "v_pk_add_f32 %5, %5, %5", "v") \
	X(40, "v_pk_fma_f32 v,v,v", "v_pk_fma_f32 %5, %5, %5, %5", "v") \
	X(41, "v_dot4_i32_i8 v,v,v", "v_dot4_i32_i8 %0, %1, %3, %0", "v") \
	X(42, "v_dot2_i32_i16 v,s,v", "v_dot2_i32_i16 %0, %1, %2, %0", "v") \
	X(43, "v_sad_u32 v,s,v", "v_sad_u32 %0, %1, %2, %0", "v")

template <int OP>
__global__ __launch_bounds__(256) void k(int32_t *out, int iters, int32_t q)
{
	int32_t a[16];
	uint64_t w[16];
#pragma unroll
	for (int i = 0; i < 16; i++) {
		a[i] = threadIdx.x + i;
		w[i] = threadIdx.x * 77 + i;
	}
	int32_t x = threadIdx.x * 3 + 1, y = (threadIdx.x & 7) + 1;
	int32_t sg = 0;
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int r = 0; r < 4; r++)
#pragma unroll
			for (int i = 0; i < 16; i++) {
#define X(N, NAME, ASM, C) if (OP == N) asm volatile(ASM : "+v"(a[i]) : "v"(x), "s"(q), "v"(y), "s"(sg), "v"(w[i]));
				CASES(X)
#undef X
			}
	}
	int32_t s = sg;
#pragma unroll
	for (int i = 0; i < 16; i++) s ^= a[i] ^ (int32_t)w[i] ^ (int32_t)(w[i] >> 32);
	out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
void run(const char *name)
{
	int32_t *out;
	hipMalloc(&out, 256 * 8 * 256 * 4);
	const int iters = 5000;
	const int wg = 8;
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(k<OP>, dim3(256 * wg), dim3(256), 0, 0, out, 100, 3);
	hipEventRecord(e0);
	hipLaunchKernelGGL(k<OP>, dim3(256 * wg), dim3(256), 0, 0, out, iters, 3);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	const double insts = (double)iters * 64;
	printf("%-26s %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, ms * 1e6 / (insts * wg) * 2.4);
	hipFree(out);
}

int main()
{
#define X(N, NAME, ASM, C) run<N>(NAME);
	CASES(X)
#undef X
	return 0;
}
