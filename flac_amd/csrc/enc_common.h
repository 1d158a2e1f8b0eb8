// Device helpers shared by the encoder kernels: PCM access, staging of a frame's signal through
// LDS, residual arithmetic, Rice parameter estimates.
#ifndef FLAC_AMD_ENC_COMMON_H
#define FLAC_AMD_ENC_COMMON_H

#include <hip/hip_runtime.h>

#include "flac_amd_internal.h"

#ifndef M_LN2
#define M_LN2 0.69314718055994530942
#endif

__device__ __forceinline__ uint32_t ilog2u(uint32_t v) { return 31u - (uint32_t)__clz((int)v); }
__device__ __forceinline__ uint32_t ilog2u64(uint64_t v) { return 63u - (uint32_t)__clzll((long long)v); }
__device__ __forceinline__ uint32_t abs32(int32_t x) { return x < 0 ? 0u - (uint32_t)x : (uint32_t)x; }
__device__ __forceinline__ uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t umax32(uint32_t a, uint32_t b) { return a > b ? a : b; }

__device__ __forceinline__ uint32_t frame_len(const FaBatch &B, uint32_t bs, uint32_t f)
{
	const uint64_t start = (uint64_t)f * bs;
	const uint64_t rem = B.wide_samples - start;
	return rem < bs ? (uint32_t)rem : bs;
}

__device__ __forceinline__ int32_t read_pcm(const void *pcm, uint32_t sample_bytes, uint64_t idx)
{
	if (sample_bytes == 2)
		return ((const int16_t *)pcm)[idx];
	if (sample_bytes == 4)
		return ((const int32_t *)pcm)[idx];
	const uint8_t *b = (const uint8_t *)pcm + idx * 3;
	return (int32_t)((uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)(int32_t)(int8_t)b[2] << 16));
}

// Signal s of a wide sample: an input channel, or mid = (L+R)>>1 / side = L-R
// (stream_encoder.c:1991-1992, arithmetic shift, not a division).  `layout` (fa_layout): 0 the caller's interleaved
// samples of sample_bytes each; 1 "pairs", the buffer is 16-bit stereo on a 4-byte boundary, so one 32-bit load brings
// both channels; a blocksize: the samples have been laid out again frame by frame as 32-bit planes,
// [frame][channel][blocksize] (k_planes), so that a workgroup that reads one channel reads whole cache lines.
__device__ __forceinline__ uint32_t fa_layout(const FaBatch &B) { return B.pairs ? 1u : B.planar; }

// signal s at one position of a frame's block of planes: fr points at channel 0's sample
__device__ __forceinline__ int32_t frame_plane_signal(const int32_t *fr, uint32_t channels, uint32_t bs, uint32_t s)
{
	if (s < channels)
		return fr[(size_t)s * bs];
	const int32_t l = fr[0], r = fr[bs];
	return s == channels ? ((l + r) >> 1) : (l - r);
}

__device__ __forceinline__ int32_t fetch_signal(const void *pcm, uint32_t channels, uint32_t sample_bytes, uint32_t pairs, uint64_t wide, uint32_t s)
{
	if (pairs > 1) {
		const uint32_t bs = pairs;
		const uint64_t f = wide / bs;
		return frame_plane_signal((const int32_t *)pcm + f * bs * channels + (wide - f * bs), channels, bs, s);
	}
	if (pairs) {
		const uint32_t w = ((const uint32_t *)pcm)[wide];
		const int32_t l = (int32_t)(int16_t)(w & 0xffff), r = (int32_t)w >> 16;
		return s == 0 ? l : s == 1 ? r : s == 2 ? ((l + r) >> 1) : (l - r);
	}
	if (s < channels)
		return read_pcm(pcm, sample_bytes, wide * channels + s);
	const int32_t l = read_pcm(pcm, sample_bytes, wide * channels);
	const int32_t r = read_pcm(pcm, sample_bytes, wide * channels + 1);
	return s == channels ? ((l + r) >> 1) : (l - r);
}

__device__ __forceinline__ uint32_t wave_or(uint32_t v)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1)
		v |= __shfl_xor(v, o, 64);
	return v;
}

__device__ __forceinline__ uint64_t wave_sum64(uint64_t v)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1)
		v += __shfl_xor(v, o, 64);
	return v;
}

// ---------------------------------------------------------------------------------------------
// Staging.  A workgroup covers FA_TILE samples per pass, thread t owning samples
// [16 t, 16 t + 16) of the tile plus HIST samples of history in front of them.  The tile (with
// ST_HP(HIST) samples of history, a multiple of 16) is read from global memory with lanes along
// consecutive samples and stored in LDS in rows of 16 samples padded to 17 words: thread t then
// reads row t + const at a fixed column, i.e. word 17 t + const, which spreads a wave over all banks.
// ---------------------------------------------------------------------------------------------
#define ST_HP(H) (((H) + 15) & ~15)
#define ST_ROWS(H) ((ST_HP(H) + FA_TILE) / 16)
#define ST_WORDS(H) (ST_ROWS(H) * 17)

// Stage samples [tile_base - ST_HP, tile_base + tile_len16) of signal s, shifted right by `wasted`;
// positions outside the frame read as 0.  tile_len16 = samples of the tile in the frame, rounded up
// to whole rows.
// How a signal is formed from the PCM buffer, resolved once per workgroup so that the loads of a
// whole block of samples can be issued back to back without branches in between.
struct SigAccess {
	uint32_t fmt;     // 0: 16-bit stereo pairs in one dword, 2 / 3 / 4: bytes per sample
	uint32_t kind;    // 0 a channel, 1 mid, 2 side
	uint32_t ca, cb;  // the channel(s) read
	uint32_t channels;
	uint32_t mul, offa, offb;   // sample `rel` of the frame, channel ca / cb: element rel * mul + offa / offb behind the frame's first
};

__device__ __forceinline__ SigAccess sig_access(uint32_t channels, uint32_t sample_bytes, uint32_t pairs, uint32_t s)
{
	SigAccess a;
	a.fmt = pairs == 1 ? 0 : (pairs ? 4 : sample_bytes);
	a.kind = s < channels ? 0 : (s == channels ? 1 : 2);
	a.ca = a.kind ? 0 : s;
	a.cb = a.kind ? 1 : s;
	a.channels = channels;
	// interleaved: sample-major; planes (pairs = the blocksize): channel-major inside the frame's block
	a.mul = pairs > 1 ? 1u : channels;
	a.offa = pairs > 1 ? a.ca * pairs : a.ca;
	a.offb = pairs > 1 ? a.cb * pairs : a.cb;
	return a;
}

__device__ __forceinline__ int32_t sig_combine(uint32_t kind, int32_t x, int32_t y) { return kind == 0 ? x : kind == 1 ? ((x + y) >> 1) : (x - y); }

// v[i] = signal value of wide sample frame_start + rel[i], for N samples; one branch on the format,
// none per sample.  The frame's base address is the same in every lane and the per-sample offsets
// are 32-bit, so each load needs one address register.
template <int N>
__device__ __forceinline__ void load_signal_block(int32_t (&v)[N], const void *pcm, const SigAccess &a, uint64_t frame_start, const uint32_t (&rel)[N])
{
	if (a.fmt == 0) {
		const uint32_t *base = (const uint32_t *)pcm + frame_start;
		uint32_t w[N];
#pragma unroll
		for (int i = 0; i < N; i++)
			w[i] = base[rel[i]];
#pragma unroll
		for (int i = 0; i < N; i++) {
			const int32_t l = (int32_t)(int16_t)(w[i] & 0xffff), r = (int32_t)w[i] >> 16;
			v[i] = a.kind == 0 ? (a.ca ? r : l) : a.kind == 1 ? ((l + r) >> 1) : (l - r);
		}
	} else if (a.fmt == 2) {
		const int16_t *base = (const int16_t *)pcm + frame_start * a.channels;
		int32_t x[N], y[N];
#pragma unroll
		for (int i = 0; i < N; i++) {
			x[i] = base[rel[i] * a.mul + a.offa];
			y[i] = base[rel[i] * a.mul + a.offb];
		}
#pragma unroll
		for (int i = 0; i < N; i++)
			v[i] = sig_combine(a.kind, x[i], y[i]);
	} else if (a.fmt == 4) {
		const int32_t *base = (const int32_t *)pcm + frame_start * a.channels;
		int32_t x[N], y[N];
#pragma unroll
		for (int i = 0; i < N; i++) {
			x[i] = base[rel[i] * a.mul + a.offa];
			y[i] = base[rel[i] * a.mul + a.offb];
		}
#pragma unroll
		for (int i = 0; i < N; i++)
			v[i] = sig_combine(a.kind, x[i], y[i]);
	} else {
		const uint8_t *base = (const uint8_t *)pcm + frame_start * a.channels * 3;
		uint32_t x0[N], x1[N], y0[N], y1[N];
		int32_t x2[N], y2[N];
#pragma unroll
		for (int i = 0; i < N; i++) {
			const uint8_t *px = base + (rel[i] * a.mul + a.offa) * 3;
			const uint8_t *py = base + (rel[i] * a.mul + a.offb) * 3;
			x0[i] = px[0]; x1[i] = px[1]; x2[i] = (int8_t)px[2];
			y0[i] = py[0]; y1[i] = py[1]; y2[i] = (int8_t)py[2];
		}
#pragma unroll
		for (int i = 0; i < N; i++)
			v[i] = sig_combine(a.kind, (int32_t)(x0[i] | (x1[i] << 8) | ((uint32_t)x2[i] << 16)), (int32_t)(y0[i] | (y1[i] << 8) | ((uint32_t)y2[i] << 16)));
	}
}

// The two halves of staging: all loads first (raw values at clamped, always valid positions), the
// LDS stores later, so that other work can be placed between them.
template <int HIST>
__device__ __forceinline__ void stage_load(int32_t (&v)[(ST_HP(HIST) + FA_TILE + FA_THREADS - 1) / FA_THREADS], const void *pcm, uint32_t channels,
                                           uint32_t sample_bytes, uint32_t pairs, uint64_t frame_start, uint32_t bsf, uint32_t s, uint32_t tile_base,
                                           uint32_t tid = threadIdx.x)
{
	constexpr uint32_t HP = ST_HP(HIST);
	constexpr uint32_t NIT = (HP + FA_TILE + FA_THREADS - 1) / FA_THREADS;
	const SigAccess a = sig_access(channels, sample_bytes, pairs, s);
	uint32_t rel[NIT];
#pragma unroll
	for (uint32_t i = 0; i < NIT; i++) {
		const uint32_t j = tid + i * FA_THREADS;
		int32_t idx = (int32_t)(tile_base + j) - (int32_t)HP;
		idx = idx < 0 ? 0 : ((uint32_t)idx >= bsf ? (int32_t)bsf - 1 : idx);
		rel[i] = (uint32_t)idx;
	}
	load_signal_block<(int)NIT>(v, pcm, a, frame_start, rel);
}

template <int HIST>
__device__ __forceinline__ void stage_store(int32_t *lds, const int32_t (&v)[(ST_HP(HIST) + FA_TILE + FA_THREADS - 1) / FA_THREADS], uint32_t bsf,
                                            uint32_t wasted, uint32_t tile_base, uint32_t tid = threadIdx.x)
{
	constexpr uint32_t HP = ST_HP(HIST);
	constexpr uint32_t NIT = (HP + FA_TILE + FA_THREADS - 1) / FA_THREADS;
	uint32_t tile_len = bsf - tile_base;
	if (tile_len > FA_TILE)
		tile_len = FA_TILE;
	const uint32_t n = HP + ((tile_len + 15) & ~15u);
#pragma unroll
	for (uint32_t i = 0; i < NIT; i++) {
		const uint32_t j = tid + i * FA_THREADS;
		const int32_t idx = (int32_t)(tile_base + j) - (int32_t)HP;
		const bool inside = idx >= 0 && (uint32_t)idx < bsf;
		if (j < n)
			lds[j + (j >> 4)] = inside ? (v[i] >> wasted) : 0;
	}
}

template <int HIST>
__device__ __forceinline__ void stage_tile(int32_t *lds, const void *pcm, uint32_t channels, uint32_t sample_bytes, uint32_t pairs,
                                           uint64_t frame_start, uint32_t bsf, uint32_t s, uint32_t wasted, uint32_t tile_base,
                                           uint32_t tid = threadIdx.x)
{
	int32_t v[(ST_HP(HIST) + FA_TILE + FA_THREADS - 1) / FA_THREADS];
	stage_load<HIST>(v, pcm, channels, sample_bytes, pairs, frame_start, bsf, s, tile_base, tid);
	stage_store<HIST>(lds, v, bsf, wasted, tile_base, tid);
}

// The thread index as a value the optimiser cannot relate to threadIdx.x: address arithmetic built
// on it is not shared with (and kept alive for) distant code that does the same sums.
__device__ __forceinline__ uint32_t opaque_tid()
{
	uint32_t t = threadIdx.x;
	asm volatile("" : "+v"(t));
	return t;
}

// d[k] = sample (16 t - HIST + k) of the staged tile, k = 0 .. HIST + 15
template <int HIST>
__device__ __forceinline__ void load_chunk_lds(int32_t (&d)[HIST + FA_CHUNK], const int32_t *lds)
{
	constexpr int HP = ST_HP(HIST);
	const int32_t *row = lds + (threadIdx.x + HP / 16) * 17;
#pragma unroll
	for (int k = 0; k < HIST + FA_CHUNK; k++) {
		const int rel = k - HIST;                    // -HIST .. 15
		const int r = rel >= 0 ? 0 : -((-rel + 15) / 16);  // floor(rel / 16)
		const int c = rel - 16 * r;
		d[k] = row[r * 17 + c];
	}
}

// format.c:548: largest partition order whose first partition still holds more than the warm-up
__device__ __forceinline__ uint32_t limit_part_order(uint32_t limit, uint32_t bs, uint32_t pred_order)
{
	uint32_t o = limit;
	while (o > 0 && (bs >> o) <= pred_order)
		o--;
	return o;
}

// stream_encoder.c:4047-4054: smallest k with n << k >= sum, clamped below the escape code
__device__ __forceinline__ uint32_t rice_k_from_sum(uint64_t sum, uint32_t n, uint32_t limit)
{
	uint32_t k = 0;
	if (sum > n) {
		const uint32_t a = ilog2u64(sum), b = ilog2u(n);
		k = a - b;
		if (((uint64_t)n << k) < sum)
			k++;
	}
	if (k >= limit)
		k = limit - 1;
	return k;
}

// stream_encoder.c:3918 count_rice_bits_in_partition_ (sum-based estimate; unsigned wraparound kept)
__device__ __forceinline__ uint32_t rice_bits_estimate(uint32_t k, uint32_t n, uint64_t sum)
{
	return 4u + (1u + k) * n + (k ? (uint32_t)(sum >> (k - 1)) : (uint32_t)(sum << 1)) - (n >> 1);
}

// the same two for sums known to fit 32 bits
__device__ __forceinline__ uint32_t rice_k_from_sum32(uint32_t sum, uint32_t n, uint32_t limit)
{
	uint32_t k = 0;
	if (sum > n) {
		k = ilog2u(sum) - ilog2u(n);
		if ((n << k) < sum)
			k++;
	}
	if (k >= limit)
		k = limit - 1;
	return k;
}

__device__ __forceinline__ uint32_t rice_bits_estimate32(uint32_t k, uint32_t n, uint32_t sum)
{
	return 4u + (1u + k) * n + (k ? (sum >> (k - 1)) : (sum << 1)) - (n >> 1);
}

// ---------------------------------------------------------------------------------------------
// residual arithmetic: d[HIST + i] is sample i of the thread's chunk; res[i] for its 16 samples
// ---------------------------------------------------------------------------------------------
// FLAC__fixed_compute_residual (fixed.c:352)
template <int HIST>
__device__ __forceinline__ void fixed_residual16(const int32_t (&d)[HIST + FA_CHUNK], uint32_t order, int32_t (&res)[FA_CHUNK])
{
#pragma unroll
	for (int i = 0; i < FA_CHUNK; i++) {
		const uint32_t a = (uint32_t)d[HIST + i], b = (uint32_t)d[HIST + i - 1], c = (uint32_t)d[HIST + i - 2], e = (uint32_t)d[HIST + i - 3],
		               g = (uint32_t)d[HIST + i - 4];
		uint32_t r;
		switch (order) {
		case 0: r = a; break;
		case 1: r = a - b; break;
		case 2: r = a - 2 * b + c; break;
		case 3: r = a - 3 * b + 3 * c - e; break;
		default: r = a - 4 * b + 6 * c - 4 * e + g; break;
		}
		res[i] = (int32_t)r;
	}
}

// The prediction of FLAC__lpc_compute_residual_from_qlp_coefficients (lpc.c:265, int32 sum that
// wraps) and _wide (lpc.c:531, int64 sum) for the thread's 16 samples.  MODE 0: 24x24-bit
// multiplies (operands known to fit), wrapping int32 sum.  MODE 1: int64 sum truncated to int32
// before the shift (== the wrapping int32 sum).  MODE 2: int64 sum, shifted, then truncated.
// `order` and q[] are the same for every lane: taps are added one at a time under a scalar branch,
// with the coefficient in a scalar register, so one copy of the code serves every order up to
// MAXORDER.
template <int HIST, int MAXORDER, int MODE, typename Q>
__device__ __forceinline__ void lpc_pred16_taps(const int32_t (&d)[HIST + FA_CHUNK], const Q &q, uint32_t order, int shift,
                                                int32_t (&pred)[FA_CHUNK])
{
	if (MODE == 0) {
		int32_t sum[FA_CHUNK];
#pragma unroll
		for (int i = 0; i < FA_CHUNK; i++)
			sum[i] = 0;
#pragma unroll
		for (int j = 0; j < MAXORDER; j++) {
			if ((uint32_t)j < order) {
				const int32_t qj = (int32_t)__builtin_amdgcn_readfirstlane((uint32_t)q[j]);
#pragma unroll
				for (int i = 0; i < FA_CHUNK; i++)
					sum[i] += __mul24(qj, d[HIST + i - 1 - j]);
			}
		}
#pragma unroll
		for (int i = 0; i < FA_CHUNK; i++)
			pred[i] = sum[i] >> shift;
	} else {
		int64_t sum[FA_CHUNK];
#pragma unroll
		for (int i = 0; i < FA_CHUNK; i++)
			sum[i] = 0;
#pragma unroll
		for (int j = 0; j < MAXORDER; j++) {
			if ((uint32_t)j < order) {
				const int64_t qj = (int64_t)(int32_t)__builtin_amdgcn_readfirstlane((uint32_t)q[j]);
#pragma unroll
				for (int i = 0; i < FA_CHUNK; i++)
					sum[i] += qj * (int64_t)d[HIST + i - 1 - j];
			}
		}
#pragma unroll
		for (int i = 0; i < FA_CHUNK; i++)
			pred[i] = (MODE == 1) ? ((int32_t)sum[i] >> shift) : (int32_t)(sum[i] >> shift);
	}
}

// The same prediction when samples and coefficients fit 16 bits: two taps per v_dot2_i32_i16.
// P[k] holds samples (16 t - HIST + k) and (.. + k + 1) of the tile in its low and high half;
// qp[jj] holds coefficient 2 jj + 1 in its low half and 2 jj in its high half (0 past the order).
typedef short fa_short2 __attribute__((ext_vector_type(2)));

template <int HIST, int MAXORDER, typename Q>
__device__ __forceinline__ void lpc_pred16_dot2(const int32_t (&P)[HIST + FA_CHUNK - 1], const Q &qp, uint32_t order, int shift,
                                                int32_t (&pred)[FA_CHUNK])
{
	int32_t sum[FA_CHUNK];
	if (order == 0) {
#pragma unroll
		for (int i = 0; i < FA_CHUNK; i++)
			pred[i] = 0;
		return;
	}
	{
		// the first pair of taps starts the sums (three-operand form with a literal zero: no
		// separate clearing of 16 registers per candidate)
		const int32_t qv = (int32_t)__builtin_amdgcn_readfirstlane((uint32_t)qp[0]);
#pragma unroll
		for (int i = 0; i < FA_CHUNK; i++)
			asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(sum[i]) : "v"(P[HIST + i - 2]), "s"(qv));
	}
#pragma unroll
	for (int jj = 1; jj < (MAXORDER + 1) / 2; jj++) {
		if ((uint32_t)(2 * jj) < order) {
			const int32_t qv = (int32_t)__builtin_amdgcn_readfirstlane((uint32_t)qp[jj]);
#pragma unroll
			for (int i = 0; i < FA_CHUNK; i++)
				sum[i] = __builtin_amdgcn_sdot2(__builtin_bit_cast(fa_short2, P[HIST + i - 2 - 2 * jj]), __builtin_bit_cast(fa_short2, qv), sum[i], false);
		}
	}
	if (shift) {
#pragma unroll
		for (int i = 0; i < FA_CHUNK; i++)
			pred[i] = sum[i] >> shift;
	} else {
#pragma unroll
		for (int i = 0; i < FA_CHUNK; i++)
			pred[i] = sum[i];
	}
}

// The prediction for samples wider than 16 bits, still two taps per instruction: a sample is split
// as s = sh * 2^12 + sl with sl its low 12 bits (unsigned) and sh = s >> 12 (signed), both of which
// fit a signed 16-bit lane, so that  sum q s = (sum q sh) * 2^12 + (sum q sl) = A * 2^12 + B  with A
// and B each one chain of v_dot2_i32_i16.  Ph[k] / Pl[k] hold sh / sl of samples (16 t - HIST + k)
// and (.. + k + 1) in their low and high halves; qp as for lpc_pred16_dot2.  The caller turns the
// candidate's arithmetic into three shift counts, pred = ((A << sa) + (B >> sb)) >> sc:
//   wrapping 32-bit sum (lpc.c:265)       sa = 12, sb = 0, sc = shift:  ((A << 12) + B) is the
//                                          sum modulo 2^32 whether or not A and B wrapped
//   64-bit sum (lpc.c:531), shift <= 12    sa = 12 - shift, sb = shift, sc = 0:  A 2^12 is a multiple
//                                          of 2^shift, so floor(sum / 2^shift) = A 2^(12-shift) +
//                                          floor(B / 2^shift), and truncation to 32 bits commutes
//                                          with the addition
//   64-bit sum, shift > 12                 sa = 0, sb = 12, sc = shift - 12:  the low 12 bits of B
//                                          cannot carry, so floor(sum / 2^shift) =
//                                          floor((A + floor(B / 2^12)) / 2^(shift-12))
// The last two need A and B (and A + (B >> 12)) to be exact in 32 bits, which the caller checks from
// the sum of the coefficients' magnitudes (fa_split_exact).
template <int HIST, int MAXORDER, typename Q>
__device__ __forceinline__ void lpc_pred16_split(const int32_t (&Ph)[HIST + FA_CHUNK - 1], const int32_t (&Pl)[HIST + FA_CHUNK - 1], const Q &qp,
                                                 uint32_t order, uint32_t sa, uint32_t sb, uint32_t sc, int32_t (&pred)[FA_CHUNK])
{
	int32_t A[FA_CHUNK], Bl[FA_CHUNK];
	if (order == 0) {
#pragma unroll
		for (int i = 0; i < FA_CHUNK; i++)
			pred[i] = 0;
		return;
	}
	{
		const int32_t qv = (int32_t)__builtin_amdgcn_readfirstlane((uint32_t)qp[0]);
#pragma unroll
		for (int i = 0; i < FA_CHUNK; i++) {
			asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(A[i]) : "v"(Ph[HIST + i - 2]), "s"(qv));
			asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(Bl[i]) : "v"(Pl[HIST + i - 2]), "s"(qv));
		}
	}
#pragma unroll
	for (int jj = 1; jj < (MAXORDER + 1) / 2; jj++) {
		if ((uint32_t)(2 * jj) < order) {
			const int32_t qv = (int32_t)__builtin_amdgcn_readfirstlane((uint32_t)qp[jj]);
#pragma unroll
			for (int i = 0; i < FA_CHUNK; i++) {
				A[i] = __builtin_amdgcn_sdot2(__builtin_bit_cast(fa_short2, Ph[HIST + i - 2 - 2 * jj]), __builtin_bit_cast(fa_short2, qv), A[i], false);
				Bl[i] = __builtin_amdgcn_sdot2(__builtin_bit_cast(fa_short2, Pl[HIST + i - 2 - 2 * jj]), __builtin_bit_cast(fa_short2, qv), Bl[i], false);
			}
		}
	}
#pragma unroll
	for (int i = 0; i < FA_CHUNK; i++)
		pred[i] = (int32_t)(((uint32_t)A[i] << sa) + (uint32_t)(Bl[i] >> sb)) >> sc;
}

// Whether lpc_pred16_split is exact for a candidate that needs the 64-bit sum: sumq = sum |q[j]|,
// sbps = bits per sample of the signal.  |B| <= sumq * 4095, |A| <= sumq * 2^max(sbps-13, 0), and
// A + (B >> 12) must fit as well.
__device__ __forceinline__ bool fa_split_exact(uint64_t sumq, uint32_t sbps)
{
	const uint32_t hb = sbps > 13 ? sbps - 13 : 0;
	return sumq <= 524415u && (sumq << hb) + (1u << 19) < (1ull << 31);
}

// The 64-bit prediction (lpc.c:531) spelled out tap by tap, for the candidates the two-taps paths
// cannot do exactly.  getd(k) returns sample (16 t - HIST + k) of the window, getq(j) coefficient j
// (the same in every lane).  wrap: the candidate's sum is the reference's wrapping 32-bit one
// (lpc.c:265), i.e. the 64-bit sum truncated before the shift.  Four samples at a time, so that it
// adds little to the register budget of the kernel around it: it is there for correctness
// (coefficient sets that large do not come out of a real signal), not speed.
template <int HIST, int MAXORDER, typename GetD, typename GetQ>
__device__ __forceinline__ void lpc_pred16_exact64(GetD getd, GetQ getq, uint32_t order, int shift, bool wrap, int32_t (&pred)[FA_CHUNK])
{
#pragma unroll
	for (int blk = 0; blk < FA_CHUNK / 4; blk++) {
		int64_t sum[4] = {0, 0, 0, 0};
#pragma unroll
		for (int j = 0; j < MAXORDER; j++) {
			if ((uint32_t)j < order) {
				const int64_t qj = (int64_t)getq(j);
#pragma unroll
				for (int i = 0; i < 4; i++)
					sum[i] += qj * (int64_t)getd(HIST + blk * 4 + i - 1 - j);
			}
		}
#pragma unroll
		for (int i = 0; i < 4; i++)
			pred[blk * 4 + i] = wrap ? ((int32_t)sum[i] >> shift) : (int32_t)(sum[i] >> shift);
	}
}

// The same as x[i] = 2^31 - residual, the form lpc_negres16_* produce at shift 0.
template <int HIST, int MAXORDER, typename GetD, typename GetQ>
__device__ __forceinline__ void lpc_negres16_exact64(GetD getd, GetQ getq, uint32_t order, int shift, bool wrap, uint32_t (&x)[FA_CHUNK])
{
#pragma unroll
	for (int blk = 0; blk < FA_CHUNK / 4; blk++) {
		int64_t sum[4] = {0, 0, 0, 0};
#pragma unroll
		for (int j = 0; j < MAXORDER; j++) {
			if ((uint32_t)j < order) {
				const int64_t qj = (int64_t)getq(j);
#pragma unroll
				for (int i = 0; i < 4; i++)
					sum[i] += qj * (int64_t)getd(HIST + blk * 4 + i - 1 - j);
			}
		}
#pragma unroll
		for (int i = 0; i < 4; i++) {
			const int32_t pred = wrap ? ((int32_t)sum[i] >> shift) : (int32_t)(sum[i] >> shift);
			x[blk * 4 + i] = 0x80000000u - ((uint32_t)getd(HIST + blk * 4 + i) - (uint32_t)pred);
		}
	}
}

// coefficient j out of words that hold two 16-bit coefficients each, the even one in the high half
// (the packing lpc_pred16_dot2 and lpc_pred16_split take)
template <typename Q>
__device__ __forceinline__ int32_t packed_coef(const Q &qp, int j)
{
	const int32_t qw = (int32_t)__builtin_amdgcn_readfirstlane((uint32_t)qp[j >> 1]);
	return (j & 1) ? (int32_t)(int16_t)(qw & 0xffff) : (qw >> 16);
}

// ---------------------------------------------------------------------------------------------
// The residual out of the multiply-add chain itself.  With c = 2^31 as the chain's starting value,
// the sample as one more tap with coefficient -2^shift and a LOGICAL shift at the end,
//     x = (2^31 + sum q[j] d[i-1-j] - d[i] 2^shift) >> shift
//       = 2^(31-shift) + floor(sum / 2^shift) - d[i]  =  B - residual,    B = 2^(31-shift),
// as long as |sum - d 2^shift| < 2^31 as an integer, which the caller guarantees per candidate from
// the sum of the coefficients' magnitudes (fa_fold_exact).  |residual| = |x - B| is then one
// v_sad_u32 against the scalar B: two instructions per sample besides the taps, where prediction,
// shift, subtraction and absolute value were four.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool fa_fold_exact(uint64_t sumq, uint32_t shift, uint32_t sbps)
{
	return shift <= 15 && ((sumq + (1ull << shift)) << (sbps - 1)) < (1ull << 31);
}

// acc + |a - b|, unsigned: written the way the compiler turns into v_sad_u32
__device__ __forceinline__ uint32_t sad_u32(uint32_t a, uint32_t b, uint32_t acc) { return acc + (a > b ? a - b : b - a); }

// Samples and coefficients fit 16 bits: two taps per v_dot2_i32_i16.  PW[k] holds samples
// (16 t - HIST + k - 1) and (.. + k) of the tile in its low and high half (PW[0]: zero and the first);
// with c[0] = -2^shift and c[m] = q[m-1], qf[jj] holds c[2 jj + 1] in its low half and c[2 jj] in its
// high half (0 past the order): order / 2 + 1 instructions per sample.  c31: 2^31 in a vector register.
template <int HIST, int MAXORDER, typename Q>
__device__ __forceinline__ void lpc_negres16_dot2(const int32_t (&PW)[HIST + FA_CHUNK], const Q &qf, uint32_t order, uint32_t shift, uint32_t c31,
                                                  uint32_t (&x)[FA_CHUNK])
{
	int32_t sum[FA_CHUNK];
	{
		const int32_t qv = (int32_t)__builtin_amdgcn_readfirstlane((uint32_t)qf[0]);
#pragma unroll
		for (int i = 0; i < FA_CHUNK; i++)
			asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(sum[i]) : "v"(PW[HIST + i]), "s"(qv), "v"(c31));
	}
#pragma unroll
	for (int jj = 1; jj < MAXORDER / 2 + 1; jj++) {
		if ((uint32_t)(2 * jj) <= order) {
			const int32_t qv = (int32_t)__builtin_amdgcn_readfirstlane((uint32_t)qf[jj]);
#pragma unroll
			for (int i = 0; i < FA_CHUNK; i++)
				sum[i] = __builtin_amdgcn_sdot2(__builtin_bit_cast(fa_short2, PW[HIST + i - 2 * jj]), __builtin_bit_cast(fa_short2, qv), sum[i], false);
		}
	}
	if (shift) {
#pragma unroll
		for (int i = 0; i < FA_CHUNK; i++)
			x[i] = (uint32_t)sum[i] >> shift;
	} else {
#pragma unroll
		for (int i = 0; i < FA_CHUNK; i++)
			x[i] = (uint32_t)sum[i];
	}
}

// The same with a tap per v_mad_i32_i24, for samples of up to 24 bits whose sums stay inside 32 bits.
template <int HIST, int MAXORDER, typename Q>
__device__ __forceinline__ void lpc_negres16_taps(const int32_t (&d)[HIST + FA_CHUNK], const Q &q, uint32_t order, uint32_t shift, uint32_t c31,
                                                  uint32_t (&x)[FA_CHUNK])
{
	uint32_t sum[FA_CHUNK];   // unsigned: the chain passes 2^31 on purpose
	const int32_t m2s = -(int32_t)(1u << shift);
#pragma unroll
	for (int i = 0; i < FA_CHUNK; i++)
		sum[i] = (uint32_t)__mul24(m2s, d[HIST + i]) + c31;
#pragma unroll
	for (int j = 0; j < MAXORDER; j++) {
		if ((uint32_t)j < order) {
			const int32_t qj = (int32_t)__builtin_amdgcn_readfirstlane((uint32_t)q[j]);
#pragma unroll
			for (int i = 0; i < FA_CHUNK; i++)
				sum[i] += (uint32_t)__mul24(qj, d[HIST + i - 1 - j]);
		}
	}
	if (shift) {
#pragma unroll
		for (int i = 0; i < FA_CHUNK; i++)
			x[i] = sum[i] >> shift;
	} else {
#pragma unroll
		for (int i = 0; i < FA_CHUNK; i++)
			x[i] = sum[i];
	}
}

// NARROW: the caller knows every candidate of the stream is MODE 0 (see fa_narrow_only)
template <int HIST, bool NARROW, typename Q>
__device__ __forceinline__ void lpc_pred16_any(const int32_t (&d)[HIST + FA_CHUNK], const Q &q, uint32_t order, int shift, int mode,
                                               int32_t (&pred)[FA_CHUNK])
{
	constexpr int MAXORDER = HIST < FA_MAX_ORDER ? HIST : FA_MAX_ORDER;
	if (NARROW || mode == 0)
		lpc_pred16_taps<HIST, MAXORDER, 0>(d, q, order, shift, pred);
	else if (mode == 1)
		lpc_pred16_taps<HIST, MAXORDER, 1>(d, q, order, shift, pred);
	else
		lpc_pred16_taps<HIST, MAXORDER, 2>(d, q, order, shift, pred);
}

template <int HIST, bool NARROW>
__device__ __forceinline__ void lpc_residual16_any(const int32_t (&d)[HIST + FA_CHUNK], const int32_t *q, uint32_t order, int shift, int mode,
                                                   int32_t (&res)[FA_CHUNK])
{
	int32_t pred[FA_CHUNK];
	lpc_pred16_any<HIST, NARROW>(d, q, order, shift, mode, pred);
#pragma unroll
	for (int i = 0; i < FA_CHUNK; i++)
		res[i] = (int32_t)((uint32_t)d[HIST + i] - (uint32_t)pred[i]);
}

// acc + |x - 2^31| for x = (value ^ 0x80000000): the absolute value of a signed `value`, added up
// (v_sad_u32 takes the unsigned absolute difference)
__device__ __forceinline__ uint32_t sad_biased(uint32_t x, uint32_t bias, uint32_t acc)
{
	uint32_t r;
	asm("v_sad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "s"(bias), "v"(acc));
	return r;
}

// which arithmetic the residual of an LPC candidate needs
__device__ __forceinline__ int lpc_mode(uint32_t wide, uint32_t sbps) { return wide ? 2 : (sbps <= 24 ? 0 : 1); }

// True when no LPC candidate of a stream with these settings can need 64-bit arithmetic or full
// 32-bit multiplies: subframe samples fit 24 bits and the reference's own rule
// bps + precision + ilog2(order) <= 32 (stream_encoder.c:3592) holds for the largest values possible.
static inline bool fa_narrow_only(const FaEncParams &P)
{
	const uint32_t max_sbps = P.bps + (P.do_ms ? 1u : 0u);
	if (P.max_lpc_order == 0)
		return max_sbps <= 24;   // (the fixed predictors' taps are 24-bit multiplies too: a side signal of 25 bits is not for them)
	const uint32_t max_prec = P.prec_search ? 15u : P.qlp_precision;
	uint32_t lg = 0;
	while ((2u << lg) <= P.max_lpc_order)
		lg++;
	return max_sbps <= 24 && max_sbps + max_prec + lg <= 32;
}

static inline int fa_hist_class(const FaEncParams &P)
{
	if (P.max_lpc_order == 0)
		return fa_narrow_only(P) ? 4 : 12;   // (the four-sample history class is compiled for narrow streams only)
	return P.max_lpc_order <= 12 ? 12 : 32;
}

#endif
