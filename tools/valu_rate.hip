// Micro-benchmark: issue rate of integer VALU instructions on gfx950 at 1, 2, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int OP>
__global__ __launch_bounds__(256) void k(int32_t *out, int iters, int32_t q)
{
	int32_t a[16];
#pragma unroll
	for (int i = 0; i < 16; i++) a[i] = threadIdx.x + i;
	int32_t x = threadIdx.x * 3 + 1;
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int i = 0; i < 16; i++) {
			if (OP == 0) a[i] += __mul24(q, x + i);            // v_mad_i32_i24 (+ add folded?)
			else if (OP == 1) a[i] = a[i] * q + x;              // v_mul_lo + add / mad_u64?
			else if (OP == 2) a[i] = (a[i] ^ x) + q;            // xor+add
			else if (OP == 4) asm volatile("v_dot2c_i32_i16 %0, %1, %2" : "+v"(a[i]) : "s"(q), "v"(x));           // one dot2 per step, 16 independent chains
			else if (OP == 5) asm volatile("v_sad_u32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "s"(q));            // one sad per step
			else if (OP == 6) asm volatile("v_lshrrev_b32 %0, %1, %0" : "+v"(a[i]) : "s"(q));                     // one shift per step
			else if (OP == 7) asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a[i]) : "s"(q), "v"(x));        // one mad24 per step
			else if (OP == 8) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(x));                          // one add per step
			else if (OP == 9) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "s"(q));       // VOP3P form
			else if (OP == 3) a[i] = __builtin_amdgcn_sdot2(__builtin_bit_cast(__attribute__((ext_vector_type(2))) short, a[i] ^ x), __builtin_bit_cast(__attribute__((ext_vector_type(2))) short, q), a[i], false);
		}
		x += q;
	}
	int32_t s = 0;
#pragma unroll
	for (int i = 0; i < 16; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
void run(const char *name, int per_iter)
{
	int32_t *out;
	hipMalloc(&out, 256 * 16 * 256 * 4);
	const int iters = 20000;
	for (int wg = 1; wg <= 8; wg *= 2) {
		hipEvent_t e0, e1;
		hipEventCreate(&e0); hipEventCreate(&e1);
		hipLaunchKernelGGL(k<OP>, dim3(256 * wg), dim3(256), 0, 0, out, 100, 3);
		hipEventRecord(e0);
		hipLaunchKernelGGL(k<OP>, dim3(256 * wg), dim3(256), 0, 0, out, iters, 3);
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		const double insts = (double)iters * per_iter;  // per wave
		printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name, wg, ms,
		       ms * 1e6 / (insts * wg), ms * 1e6 / (insts * wg) * 2.4);
	}
	hipFree(out);
}

int main()
{
	run<0>("mad_i32_i24 (+add)", 32);
	run<2>("xor+add", 32);
	run<1>("mul_lo + add", 32);
	run<3>("xor + sdot2", 32);
	run<4>("v_dot2c_i32_i16", 16);
	run<9>("v_dot2_i32_i16 (vop3p)", 16);
	run<5>("v_sad_u32", 16);
	run<6>("v_lshrrev_b32", 16);
	run<7>("v_mad_i32_i24", 16);
	run<8>("v_add_u32", 16);
	return 0;
}
