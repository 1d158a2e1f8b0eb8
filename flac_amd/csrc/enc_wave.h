// Pieces shared by the wave-per-subframe kernels of the encoder (enc_search_wave.hip, enc_pack_wave.hip):
// a wave covers one (frame, signal) of a stream with blocks of 4096 samples, lane l owning samples
// [64 l, 64 l + 64) of the frame with twelve samples of history in front of them, all in registers.
#ifndef FLAC_AMD_ENC_WAVE_H
#define FLAC_AMD_ENC_WAVE_H

#include "enc_common.h"

#define SW_N 64                  // samples per lane
#define SW_HIST 12               // history in front of them = largest predictor order
#define SW_W (SW_HIST + SW_N)
#define SW_ROW 68                // LDS words per staged row of 64 samples (four words of padding)
#define SW_NQF (SW_HIST / 2 + 1) // words of the folded two-taps form (see lpc_negres16_dot2)


// The lane's run of signal s of the frame, shifted right by `wasted`, positions outside the frame read
// as 0: d[k] = sample 64 l - SW_HIST + k.  A quarter of the frame at a time goes through LDS (stage:
// 17 * SW_ROW words), read with lanes along consecutive samples, stored as rows of 64, and sixteen lanes
// pick up a row each plus the tail of the row before it.  partial: the frame is shorter than 4096 samples.
__device__ __forceinline__ void sw_load_run(int32_t (&d)[SW_W], int32_t *stage, const void *pcm, const SigAccess &a, uint64_t frame_start, uint32_t bsf,
                                            bool partial, uint32_t wasted, uint32_t ln)
{
#pragma unroll
	for (int p = 0; p < 4; p++) {
		uint32_t rel[17];
		int32_t v[17];
		if (!partial) {
			// (only the very first position lies in front of the frame)
#pragma unroll
			for (int i = 0; i < 17; i++)
				rel[i] = (p == 0 && i == 0) ? ln : (uint32_t)(1024 * p - 64 + 64 * i) + ln;
			load_signal_block<17>(v, pcm, a, frame_start, rel);
#pragma unroll
			for (int i = 0; i < 17; i++)
				stage[i * SW_ROW + ln] = (p == 0 && i == 0) ? 0 : (v[i] >> wasted);
		} else {
#pragma unroll
			for (int i = 0; i < 17; i++) {
				const int32_t idx = 1024 * p - 64 + 64 * i + (int32_t)ln;
				rel[i] = idx < 0 ? 0u : umin32((uint32_t)idx, bsf - 1);
			}
			load_signal_block<17>(v, pcm, a, frame_start, rel);
#pragma unroll
			for (int i = 0; i < 17; i++) {
				const int32_t idx = 1024 * p - 64 + 64 * i + (int32_t)ln;
				stage[i * SW_ROW + ln] = (idx < 0 || (uint32_t)idx >= bsf) ? 0 : (v[i] >> wasted);
			}
		}
		__syncthreads();
		if ((ln >> 4) == (uint32_t)p) {
			const int32_t *row = stage + ((ln & 15u) + 1u) * SW_ROW;
#pragma unroll
			for (int k = 0; k < SW_HIST; k++)
				d[k] = row[-SW_ROW + 64 - SW_HIST + k];
#pragma unroll
			for (int k = 0; k < SW_N; k++)
				d[SW_HIST + k] = row[k];
		}
		__syncthreads();
	}
}

// x[i] = B - residual for samples OFF .. OFF + 15 of the lane's run, B = 2^(31 - shift), a tap per
// v_mad_i32_i24 (the arithmetic of lpc_negres16_taps in enc_common.h; q in scalar registers).  The
// taps nest: the first one the candidate's order does not reach ends the chain with one branch.
template <int OFF, int J>
__device__ __forceinline__ void sw_taps_groups(const int32_t (&d)[SW_W], const int32_t (&q)[SW_HIST], uint32_t order, uint32_t (&sum)[16])
{
	if constexpr (J < SW_HIST) {
		if ((uint32_t)J < order) {
#pragma unroll
			for (int i = 0; i < 16; i++)
				sum[i] += (uint32_t)__mul24(q[J], d[SW_HIST + OFF + i - 1 - J]);
			sw_taps_groups<OFF, J + 1>(d, q, order, sum);
		}
	}
}

template <int OFF>
__device__ __forceinline__ void sw_negres_taps(const int32_t (&d)[SW_W], const int32_t (&q)[SW_HIST], uint32_t order, uint32_t shift, uint32_t c31,
                                               uint32_t (&x)[16])
{
	uint32_t sum[16];
	const int32_t m2s = -(int32_t)(1u << shift);
#pragma unroll
	for (int i = 0; i < 16; i++)
		sum[i] = (uint32_t)__mul24(m2s, d[SW_HIST + OFF + i]) + c31;
	sw_taps_groups<OFF, 0>(d, q, order, sum);
	if (shift) {
#pragma unroll
		for (int i = 0; i < 16; i++)
			x[i] = sum[i] >> shift;
	} else {
#pragma unroll
		for (int i = 0; i < 16; i++)
			x[i] = sum[i];
	}
}

// inclusive prefix sum over the wave's lanes: four shifts inside each row of sixteen, then the last
// lane of row 0 (2) added to row 1 (3) and the last lane of row 1 to rows 2 and 3
__device__ __forceinline__ uint32_t sw_scan_incl(uint32_t v)
{
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);    // row_shr:1
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);    // row_shr:2
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);    // row_shr:4
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);    // row_shr:8
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15, rows 1 and 3
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31, rows 2 and 3
	return v;
}

#endif
