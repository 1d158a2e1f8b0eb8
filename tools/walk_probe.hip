// Micro-benchmark: what one Rice symbol costs a wave that walks a stream the way k_dec_frames' plain block does
// (two stream words and a count of unread bits in registers, the words behind them from a ring in LDS), in variants:
//   0  the ring word for the slot that opens up is requested after the step (k_dec_frames as built)
//   1  no ring read at all (the words are made up): the floor of the arithmetic alone
//   2  the ring word is requested at the start of the step, before it is known whether it will be needed
//   3  two independent streams per lane, their steps interleaved (variant 0 twice)
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/walk_probe tools/walk_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>

#define RING 128

struct St {
	uint32_t cur, nxt, n2, n3, rem, wrel;
};

template <int V>
__device__ __forceinline__ uint32_t step(St &s, const uint32_t *ring, uint32_t kp1, uint32_t &tmax)
{
	uint32_t early = 0;
	if (V == 2)
		early = ring[(s.wrel + 2) & (RING - 1)];
	const uint32_t win = __builtin_amdgcn_alignbit(s.cur, s.nxt, s.rem);
	const uint32_t z = (uint32_t)__builtin_clz(win | 1u);   // (the probe's streams never hold 32 zero bits)
	const uint32_t total = z + kp1;
	tmax = total > tmax ? total : tmax;
	const int32_t left = (int32_t)s.rem - (int32_t)total;
	const bool cross = left < 0;
	s.cur = cross ? s.nxt : s.cur;
	s.nxt = cross ? s.n2 : s.nxt;
	s.n2 = cross ? s.n3 : s.n2;
	s.wrel += cross ? 1u : 0u;
	s.rem = (uint32_t)left & 31u;
	if (V == 0 || V == 3)
		s.n3 = ring[(s.wrel + 1) & (RING - 1)];
	else if (V == 1)
		s.n3 = s.n3 * 1664525u + 1013904223u;
	else
		s.n3 = cross ? early : s.n3;
	return z;
}

template <int V>
__global__ __launch_bounds__(64) void k(uint64_t *stamps, uint32_t *out, int blocks, uint32_t lanes, uint32_t kp1)
{
	__shared__ uint32_t ring[2][16][RING];
	const uint32_t tid = threadIdx.x;
	for (uint32_t e = tid; e < 2 * 16 * RING; e += 64) {
		uint32_t x = e * 2654435761u + blockIdx.x * 97u;
		x ^= x >> 13; x *= 0x5bd1e995u; x ^= x >> 15;
		(&ring[0][0][0])[e] = x | 0x01010101u;   // a one bit in every byte: no symbol is longer than 8 + k bits
	}
	__syncthreads();
	const uint32_t lane = tid < 16 ? tid : 0;
	const bool act = tid < lanes;
	St a, b;
	a.cur = ring[0][lane][0]; a.nxt = ring[0][lane][1]; a.n2 = ring[0][lane][2]; a.n3 = ring[0][lane][3]; a.rem = 17; a.wrel = 2;
	b.cur = ring[1][lane][0]; b.nxt = ring[1][lane][1]; b.n2 = ring[1][lane][2]; b.n3 = ring[1][lane][3]; b.rem = 5; b.wrel = 2;
	uint32_t tmax = 0, acc = 0;
	const uint64_t c0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < blocks; it++) {
#pragma unroll
		for (int s = 0; s < 16; s++) {
			acc += step<V>(a, ring[0][lane], kp1, tmax);
			if (V == 3)
				acc += step<V>(b, ring[1][lane], kp1, tmax);
		}
		if (__any(act && tmax > 40))   // never true: stands for the block's one check
			break;
	}
	const uint64_t c1 = __builtin_amdgcn_s_memtime();
	out[blockIdx.x * 64 + tid] = acc + a.cur + b.cur + tmax;
	if (tid == 0)
		stamps[blockIdx.x] = c1 - c0;
}

template <int V>
void run(const char *name, int waves, uint32_t lanes, uint32_t kp1 = 7u)
{
	uint64_t *stamps;
	uint32_t *out;
	hipMalloc(&stamps, sizeof(uint64_t) * waves);
	hipMalloc(&out, sizeof(uint32_t) * 64 * waves);
	const int blocks = 256;
	for (int rep = 0; rep < 3; rep++)
		hipLaunchKernelGGL(k<V>, dim3(waves), dim3(64), 0, 0, stamps, out, blocks, lanes, kp1);
	hipDeviceSynchronize();
	std::vector<uint64_t> h(waves);
	hipMemcpy(h.data(), stamps, sizeof(uint64_t) * waves, hipMemcpyDeviceToHost);
	std::sort(h.begin(), h.end());
	const double per = (double)h[waves / 2] / (blocks * 16.0) / (V == 3 ? 2.0 : 1.0);
	printf("%-44s waves %5d lanes %2u  k+1 %2u  cycles per symbol (median wave) %.1f\n", name, waves, lanes, kp1, per);
	hipFree(stamps);
	hipFree(out);
}

int main()
{
	for (int waves : {256, 808, 1616, 3232}) {
		run<0>("0 ring word requested after the step", waves, 8);
		run<1>("1 no ring read", waves, 8);
		run<2>("2 ring word requested before the step", waves, 8);
		run<3>("3 two streams per lane, interleaved", waves, 8);
	}
	// symbols of 24-bit streams are twice as long: a step into the next word every other symbol
	for (uint32_t kp1 : {7u, 11u, 15u, 19u})
		run<0>("0 ring word requested after the step", 352, 8, kp1);
	return 0;
}
