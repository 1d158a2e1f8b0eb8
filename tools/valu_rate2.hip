// Micro-benchmark: issue cost of 64-bit shifts, 64-bit multiply-adds and v_alignbit_b32 on gfx950 (for the decoder's
// bit window and wide predictor), at 1, 2, 4, 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int OP>
__global__ __launch_bounds__(256) void k(uint64_t *out, int iters, int32_t q)
{
	uint64_t a[16];
#pragma unroll
	for (int i = 0; i < 16; i++) a[i] = ((uint64_t)threadIdx.x << 20) + i * 77 + 1;
	int32_t x = threadIdx.x * 3 + 1;
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int i = 0; i < 16; i++) {
			if (OP == 0) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(a[i]) : "v"(x & 3));
			else if (OP == 1) a[i] += (int64_t)q * (int64_t)(x + i);
			else if (OP == 2) { uint32_t lo = (uint32_t)a[i], hi = (uint32_t)(a[i] >> 32), r; asm volatile("v_alignbit_b32 %0, %1, %2, %3" : "=v"(r) : "v"(hi), "v"(lo), "v"(x & 31)); a[i] = ((uint64_t)lo << 32) | r; }
			else if (OP == 3) { uint32_t lo = (uint32_t)a[i]; lo = (lo ^ (uint32_t)x) + (uint32_t)q; a[i] = (a[i] & 0xffffffff00000000ull) | lo; }
		}
		x += q;
	}
	uint64_t s = 0;
#pragma unroll
	for (int i = 0; i < 16; i++) s ^= a[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
void run(const char *name, int per_iter)
{
	uint64_t *out;
	hipMalloc(&out, 256 * 16 * 256 * 8);
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
		const double insts = (double)iters * per_iter;
		printf("%-22s waves/SIMD=%d  %.3f ms  -> %.2f cycles (2.4 GHz) per counted instruction per SIMD\n", name, wg, ms, ms * 1e6 / (insts * wg) * 2.4);
	}
	hipFree(out);
}

int main()
{
	run<3>("xor+add (2 counted)", 32);
	run<0>("v_lshlrev_b64", 16);
	run<1>("v_mad_i64_i32", 16);
	run<2>("v_alignbit_b32", 16);
	return 0;
}
