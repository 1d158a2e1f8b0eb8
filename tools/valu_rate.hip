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
	return 0;
}
