// Micro-benchmark: the shader clock a short, thinly spread kernel actually runs at, and what a dependent / an independent
// integer instruction costs a wave that has its SIMD to itself, in cycles of that clock.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/clock_probe tools/clock_probe.hip && tools/bin/clock_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>

// DEP: one chain of dependent adds; else sixteen independent chains
template <bool DEP>
__global__ __launch_bounds__(64) void k(uint64_t *stamps, int32_t *out, int iters, int32_t q)
{
	int32_t a[16];
#pragma unroll
	for (int i = 0; i < 16; i++)
		a[i] = threadIdx.x + i;
	const uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int i = 0; i < 16; i++) {
			if (DEP)
				asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[0]) : "v"(q));
			else
				asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "v"(q));
		}
	}
	const uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
	int32_t s = 0;
#pragma unroll
	for (int i = 0; i < 16; i++)
		s ^= a[i];
	out[blockIdx.x * 64 + threadIdx.x] = s;
	if (threadIdx.x == 0) {
		stamps[2 * blockIdx.x] = c1 - c0;
		stamps[2 * blockIdx.x + 1] = r1 - r0;
	}
}

template <bool DEP>
void run(const char *name, int waves, int iters)
{
	uint64_t *stamps;
	int32_t *out;
	hipMalloc(&stamps, sizeof(uint64_t) * 2 * waves);
	hipMalloc(&out, sizeof(int32_t) * 64 * waves);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	for (int rep = 0; rep < 3; rep++) {
		hipEventRecord(e0);
		hipLaunchKernelGGL(k<DEP>, dim3(waves), dim3(64), 0, 0, stamps, out, iters, 3);
		hipEventRecord(e1);
		hipEventSynchronize(e1);
	}
	float ms;
	hipEventElapsedTime(&ms, e0, e1);
	std::vector<uint64_t> h(2 * waves);
	hipMemcpy(h.data(), stamps, sizeof(uint64_t) * 2 * waves, hipMemcpyDeviceToHost);
	std::vector<double> mhz, cyc;
	for (int w = 0; w < waves; w++) {
		mhz.push_back((double)h[2 * w] / (double)h[2 * w + 1] * 100.0);
		cyc.push_back((double)h[2 * w] / ((double)iters * 16));
	}
	std::sort(mhz.begin(), mhz.end());
	std::sort(cyc.begin(), cyc.end());
	printf("%-12s waves %6d  kernel %.3f ms  clock (median over waves) %.0f MHz  cycles per instruction: median %.2f  (%.1f ns)\n", name, waves, ms,
	       mhz[waves / 2], cyc[waves / 2], cyc[waves / 2] / mhz[waves / 2] * 1e3);
	hipFree(stamps);
	hipFree(out);
}

int main()
{
	for (int waves : {64, 808, 2048, 8192}) {
		for (int iters : {2000, 20000}) {
			run<true>("dependent", waves, iters);
			run<false>("independent", waves, iters);
		}
	}
	return 0;
}
