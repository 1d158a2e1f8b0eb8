// FLAC frame encoder kernels for gfx950 (MI355X), part 1: signal statistics, autocorrelation,
// LPC analysis, channel decision, frame offsets and compaction.
//
// One batch = F consecutive frames of one stream.  Work decomposition:
//   k_analyze   grid F*S  (S = input channels [+ mid + side])   wasted bits, fixed-predictor sums
//   k_autoc     grid ceil(F*S*A / tasks-per-block)              windowed autocorrelation, exact order
//   k_lpc       lane per (frame, signal)                         Levinson-Durbin, order guess, quantise
//   k_search    grid F*S             (enc_search.hip)            residual + Rice partition search
//   k_decide    lane per frame                                   channel assignment
//   k_pack      grid F               (enc_pack.hip)              bit packing into frame slots, CRCs
//   k_scan      one block                                        frame byte offsets
//   k_compact   grid F                                           slots -> contiguous stream bytes
//
// Arithmetic follows the reference (FLAC 1.2.1) operation for operation where rounding or
// wraparound is observable; each kernel says which reference function it stands in for.
// Build with -ffp-contract=off: the reference's float/double expressions are not fused.
#include <hip/hip_runtime.h>

#include "enc_common.h"
#include "enc_kernels.h"

// ---------------------------------------------------------------------------------------------
// k_planes: the part of FLAC__stream_encoder_process_interleaved that takes the interleaved samples apart
// (stream_encoder.c:2071-2097) for streams of three or more channels: 256 wide samples of a frame per
// workgroup, read along memory, turned in LDS, written as one run per channel into the frame's block
// [channel][blocksize] of 32-bit words.  Positions behind a short last block are never read by anyone.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_planes(FaEncParams P, FaBatch B, const void *pcm, int32_t *planes, uint32_t tiles)
{
	__shared__ int32_t t[FLACAMD_MAX_CHANNELS * 257];
	const uint32_t f = blockIdx.x / tiles, t0 = (blockIdx.x % tiles) * 256, tid = threadIdx.x;
	const uint32_t bsf = frame_len(B, P.blocksize, f), ch = P.channels;
	if (t0 >= bsf)
		return;
	const uint32_t n = umin32(256u, bsf - t0);
	const uint64_t first = ((uint64_t)f * P.blocksize + t0) * ch;
	for (uint32_t e = tid; e < n * ch; e += 256) {
		const uint32_t i = e / ch, c = e - i * ch;
		t[c * 257 + i] = read_pcm(pcm, B.sample_bytes, first + e);
	}
	__syncthreads();
	if (tid < n)
		for (uint32_t c = 0; c < ch; c++)
			planes[((size_t)f * ch + c) * P.blocksize + t0 + tid] = t[c * 257 + tid];
}

// What the statistics passes leave per (frame, signal), from the sums of the fixed predictors' absolute errors over the
// signal with its wasted bits gone (fixed.c:264-282: the order guess and the bits per sample each order would cost).
__device__ __forceinline__ FaSubStat make_stat(const FaEncParams &P, uint32_t bsf, uint32_t wasted, uint32_t extra_bit, int32_t first, uint32_t ne,
                                               const uint64_t (&sums)[5])
{
	FaSubStat st;
	st.wasted = wasted;
	st.bps = P.bps - wasted + extra_bit;
	st.first_sample = first >> wasted;
	st.is_constant = 0;
	st.fixed_guess = 0;
	for (int k = 0; k < 5; k++)
		st.rbps[k] = 0.0f;
	if (bsf >= 4) {
		uint64_t t[5];
		for (int k = 0; k < 5; k++)
			t[k] = P.wide_fixed ? sums[k] : (sums[k] & 0xffffffffull);  // uint32 accumulators wrap (fixed.c:234)
		auto mn = [](uint64_t a, uint64_t b) { return a < b ? a : b; };
		uint32_t order;
		if (t[0] < mn(mn(mn(t[1], t[2]), t[3]), t[4])) order = 0;
		else if (t[1] < mn(mn(t[2], t[3]), t[4])) order = 1;
		else if (t[2] < mn(t[3], t[4])) order = 2;
		else if (t[3] < t[4]) order = 3;
		else order = 4;
		st.fixed_guess = order;
		const double n = (double)(bsf - 4);
		for (int k = 0; k < 5; k++)
			st.rbps[k] = (float)(t[k] > 0 ? log(M_LN2 * (double)t[k] / n) / M_LN2 : 0.0);
		st.is_constant = (!P.disable_constant && st.rbps[1] == 0.0f && ne == 0) ? 1u : 0u;
	}
	return st;
}

// ---------------------------------------------------------------------------------------------
// k_planes_stats + k_stats_finish: k_planes and k_analyze in one pass over the samples, for streams of up to eight
// channels of up to 24 bits (with mid and side when there are two; without the planes when the stream is read as it
// lies).  A workgroup takes PS_TILE wide samples of a frame: reads them along memory, turns them
// in LDS, writes each channel's run into its plane, and a thread per (signal, sixteen samples) adds up what the
// statistics need of them: the OR of the samples, whether any differs from the channel's first, and the five error
// sums, taken over the samples AS THEY ARE.  The wasted bits are only known once the whole block has been seen, but
// every difference of samples that all end in w zero bits ends in w zero bits, so the sums over the shifted samples
// are the sums over the unshifted ones shifted by w: k_stats_finish, a lane per (frame, channel), does that and the
// rest.  The tiles of a block meet in 64-bit atomic adds (`acc`, cleared beforehand).
// ---------------------------------------------------------------------------------------------
// CPS: threads per signal (each takes sixteen samples), so a tile is 16 * CPS wide samples: 32 for up to eight
// signals, 64 / 128 / 256 for up to four / two / one - always 256 threads' worth.
#define PS_HP 16                                 // rows in front of the tile (the predictors need four)
#define PS_ROWW(TILE) ((((TILE) + PS_HP) / 16) * 17)   // words of a channel's row in LDS: sixteen samples to seventeen words
struct FaStatAcc {
	unsigned long long sums[5];
	uint32_t orv, ne;
	int32_t first;
	uint32_t pad;
};

template <int CPS>
__global__ __launch_bounds__(256) void k_planes_stats(FaEncParams P, FaBatch B, const void *pcm, int32_t *planes, FaStatAcc *acc, uint32_t tiles)
{
	constexpr uint32_t PS_TILE = 16 * CPS, ROWW = PS_ROWW(PS_TILE);
	constexpr uint32_t MAXCH = CPS == 32 ? FLACAMD_MAX_CHANNELS : CPS == 64 ? 4 : CPS == 128 ? 2 : 1;   // channels <= signals <= 256 / CPS
	__shared__ int32_t x[MAXCH * ROWW];
	const uint32_t f = blockIdx.x / tiles, t0 = (blockIdx.x % tiles) * PS_TILE, tid = threadIdx.x;
	const uint32_t bsf = frame_len(B, P.blocksize, f), ch = P.channels;
	if (t0 >= bsf)
		return;
	const uint32_t n = umin32((uint32_t)PS_TILE, bsf - t0);        // samples of the tile
	const uint32_t lead = t0 ? (uint32_t)PS_HP : 0u;               // rows in front of it that exist
	const uint64_t first_el = ((uint64_t)f * P.blocksize + t0 - lead) * ch;
	const uint32_t nel = (n + lead) * ch;
	constexpr uint32_t NIT = ((PS_TILE + PS_HP) * MAXCH + 255) / 256;
	// every load first; local row j = PS_HP + (sample - t0)
	int32_t v[NIT];
#pragma unroll
	for (uint32_t k = 0; k < NIT; k++) {
		const uint32_t e = tid + 256 * k;
		v[k] = e < nel ? read_pcm(pcm, B.sample_bytes, first_el + e) : 0;
	}
	// thread -> (signal, sixteen samples); a signal is a channel, or mid / side of the first two (stream_encoder.c:1991-1992)
	const uint32_t sg = tid / CPS, chunk = tid % CPS;
	const bool worker = sg < P.nsignals;
	const uint32_t kind = sg < ch ? 0u : (sg == ch ? 1u : 2u), ca = kind ? 0u : sg, cb = kind ? 1u : sg;
	// the signal's first sample of the block (for the all-samples-equal test)
	int32_t first = 0;
	if (worker) {
		const int32_t fa = read_pcm(pcm, B.sample_bytes, (uint64_t)f * P.blocksize * ch + ca);
		const int32_t fb = kind ? read_pcm(pcm, B.sample_bytes, (uint64_t)f * P.blocksize * ch + cb) : fa;
		first = sig_combine(kind, fa, fb);
	}
#pragma unroll
	for (uint32_t k = 0; k < NIT; k++) {
		const uint32_t e = tid + 256 * k;
		if (e < nel) {
			const uint32_t i = e / ch, c = e - i * ch, j = i + (PS_HP - lead);
			x[c * ROWW + j + (j >> 4)] = v[k];
		}
	}
	__syncthreads();
	// the planes
	for (uint32_t c = 0; planes && c < ch; c++) {
		int32_t *dst = planes + ((size_t)f * ch + c) * P.blocksize + t0;
		for (uint32_t i = tid; i < n; i += 256) {
			const uint32_t j = i + PS_HP;
			dst[i] = x[c * ROWW + j + (j >> 4)];
		}
	}
	// the statistics
	if (worker) {
		const uint32_t base = t0 + 16 * chunk;   // the thread's first sample, counted in the block
		uint32_t orv = 0, ne = 0;
		uint64_t te[5] = {0, 0, 0, 0, 0};
		if (base < bsf) {
			int32_t d[4 + 16];
			const int32_t *rowa = x + ca * ROWW + (chunk + PS_HP / 16) * 17, *rowb = x + cb * ROWW + (chunk + PS_HP / 16) * 17;
#pragma unroll
			for (int k = 0; k < 4; k++)
				d[k] = sig_combine(kind, rowa[-17 + 12 + k], rowb[-17 + 12 + k]);          // the last four samples of the sixteen before
#pragma unroll
			for (int k = 0; k < 16; k++)
				d[4 + k] = sig_combine(kind, rowa[k], rowb[k]);
			const int32_t fs = first;
			const uint32_t BIAS = 0x80000000u;
			uint32_t t32[5] = {0, 0, 0, 0, 0};
			uint32_t p0 = 0, p1 = 0, p2 = 0, p3 = 0;
#pragma unroll
			for (int k = 0; k < 4 + 16; k++) {
				const uint32_t b0 = (uint32_t)d[k] ^ BIAS;
				const uint32_t b1 = (b0 - p0) ^ BIAS, b2 = (b1 - p1) ^ BIAS, b3 = (b2 - p2) ^ BIAS;
				if (k >= 4) {
					const uint32_t idx = base + (uint32_t)(k - 4);
					if (idx < bsf) {
						orv |= (uint32_t)d[k];
						ne |= (uint32_t)(d[k] != fs);
					}
					if (idx >= 4 && idx < bsf) {
						t32[0] = sad_u32(b0, BIAS, t32[0]);
						t32[1] = sad_u32(b0, p0, t32[1]);
						t32[2] = sad_u32(b1, p1, t32[2]);
						t32[3] = sad_u32(b2, p2, t32[3]);
						t32[4] = sad_u32(b3, p3, t32[4]);
					}
					if ((k & 3) == 3) {
						// (four errors of up to 29 bits - a side signal of 25, differenced four times: no carry out of 32)
						te[0] += t32[0]; te[1] += t32[1]; te[2] += t32[2]; te[3] += t32[3]; te[4] += t32[4];
						t32[0] = t32[1] = t32[2] = t32[3] = t32[4] = 0;
					}
				}
				p0 = b0;
				p1 = b1;
				p2 = b2;
				p3 = b3;
			}
		}
		// the threads of a signal are half a wave, or whole waves
#pragma unroll
		for (int o = (CPS < 64 ? CPS : 64) / 2; o > 0; o >>= 1) {
			orv |= (uint32_t)__shfl_xor((int)orv, o, 64);
			ne |= (uint32_t)__shfl_xor((int)ne, o, 64);
#pragma unroll
			for (int k = 0; k < 5; k++)
				te[k] += __shfl_xor(te[k], o, 64);
		}
		if ((chunk & 63u) == 0) {
			FaStatAcc *a = acc + (size_t)f * P.nsignals + sg;
#pragma unroll
			for (int k = 0; k < 5; k++)
				if (te[k])
					atomicAdd(&a->sums[k], (unsigned long long)te[k]);
			if (orv)
				atomicOr(&a->orv, orv);
			if (ne)
				atomicOr(&a->ne, 1u);
			if (t0 == 0 && chunk == 0)
				a->first = first;
		}
	}
}

__global__ __launch_bounds__(256) void k_stats_finish(FaEncParams P, FaBatch B, const FaStatAcc *acc, FaSubStat *stats)
{
	const uint32_t task = blockIdx.x * 256 + threadIdx.x;
	if (task >= B.nframes * P.nsignals)
		return;
	const FaStatAcc a = acc[task];
	const uint32_t wasted = a.orv ? (uint32_t)__ffs((int)a.orv) - 1u : 0u;
	uint64_t t[5];
	for (int k = 0; k < 5; k++)
		t[k] = a.sums[k] >> wasted;
	const uint32_t f = task / P.nsignals, sg = task - f * P.nsignals;
	stats[task] = make_stat(P, frame_len(B, P.blocksize, f), wasted, (P.do_ms && sg == P.channels + 1) ? 1u : 0u, a.first, a.ne, t);
}

// ---------------------------------------------------------------------------------------------
// k_analyze: get_wasted_bits_ (stream_encoder.c:4108) + FLAC__fixed_compute_best_predictor[_wide]
// (fixed.c:224,282) + the all-samples-equal test of process_subframe_ (:3210-3227)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(FA_THREADS) void k_analyze(FaEncParams P, FaBatch B, const void *pcm, FaSubStat *stats)
{
	// Workgroups go to the chip's eight dies in turn, each with an L2 of its own: the workgroups of one frame's
	// signals, which read the same interleaved samples, are given indices that are equal modulo eight.
	const uint32_t f = (blockIdx.x / (8 * P.nsignals)) * 8 + (blockIdx.x & 7u), s = (blockIdx.x % (8 * P.nsignals)) >> 3;
	if (f >= B.nframes)
		return;
	const uint32_t task = f * P.nsignals + s;
	const uint32_t tid = threadIdx.x;
	const uint32_t bsf = frame_len(B, P.blocksize, f);
	const uint64_t frame_start = (uint64_t)f * P.blocksize;
	__shared__ uint32_t sh_or[4], sh_ne[4];
	__shared__ uint64_t sh_te[4][5];
	__shared__ int32_t stage[ST_WORDS(4)];

	// pass 1: OR of all samples (wasted bits) and whether they are all equal; the first tile stays
	// staged (unshifted) for pass 2.  Loads are issued a tile at a time, back to back.
	constexpr uint32_t HP = ST_HP(4);
	constexpr uint32_t NIT = (HP + FA_TILE + FA_THREADS - 1) / FA_THREADS;
	const SigAccess acc = sig_access(P.channels, B.sample_bytes, fa_layout(B), s);
	int32_t first;
	uint32_t orv = 0, ne = 0;
	{
		uint32_t wide[NIT + 1];
		int32_t v[NIT + 1];
#pragma unroll
		for (uint32_t i = 0; i < NIT; i++) {
			const uint32_t j = tid + i * FA_THREADS;
			int32_t idx = (int32_t)j - (int32_t)HP;
			idx = idx < 0 ? 0 : ((uint32_t)idx >= bsf ? (int32_t)bsf - 1 : idx);
			wide[i] = (uint32_t)idx;
		}
		wide[NIT] = 0;
		load_signal_block<(int)NIT + 1>(v, pcm, acc, frame_start, wide);
		first = v[NIT];
		const uint32_t n0 = HP + ((umin32(bsf, FA_TILE) + 15) & ~15u);
#pragma unroll
		for (uint32_t i = 0; i < NIT; i++) {
			const uint32_t j = tid + i * FA_THREADS;
			const bool inside = j >= HP && j - HP < bsf;
			const int32_t x = inside ? v[i] : 0;
			orv |= (uint32_t)x;
			ne |= (uint32_t)(inside && x != first);
			if (j < n0)
				stage[j + (j >> 4)] = x;
		}
	}
	for (uint32_t t0 = FA_TILE; t0 < bsf; t0 += FA_TILE) {
		int32_t v[NIT];
		stage_load<4>(v, pcm, P.channels, B.sample_bytes, fa_layout(B), frame_start, bsf, s, t0);
#pragma unroll
		for (uint32_t i = 0; i < NIT; i++) {
			const uint32_t j = tid + i * FA_THREADS;
			const bool inside = j >= HP && t0 + j - HP < bsf;
			orv |= inside ? (uint32_t)v[i] : 0u;
			ne |= (uint32_t)(inside && v[i] != first);
		}
	}
	orv = wave_or(orv);
	ne = wave_or(ne);
	if ((tid & 63) == 0) {
		sh_or[tid >> 6] = orv;
		sh_ne[tid >> 6] = ne;
	}
	__syncthreads();
	orv = sh_or[0] | sh_or[1] | sh_or[2] | sh_or[3];
	ne = sh_ne[0] | sh_ne[1] | sh_ne[2] | sh_ne[3];
	const uint32_t wasted = orv ? (uint32_t)__ffs((int)orv) - 1u : 0u;

	uint64_t te[5] = {0, 0, 0, 0, 0};
	if (bsf >= 4) {
		for (uint32_t tile = 0; tile * FA_TILE < bsf; tile++) {
			if (tile > 0) {
				__syncthreads();
				stage_tile<4>(stage, pcm, P.channels, B.sample_bytes, fa_layout(B), frame_start, bsf, s, 0, tile * FA_TILE);
				__syncthreads();
			}
			const uint32_t base = tile * FA_TILE + tid * FA_CHUNK;
			if (base >= bsf)
				continue;
			int32_t d[4 + FA_CHUNK];
			load_chunk_lds<4>(d, stage);
#pragma unroll
			for (int k = 0; k < 4 + FA_CHUNK; k++)
				d[k] >>= wasted;
			// the wide (64-bit) sums are only needed when a partial sum could pass 2^32
			uint32_t t32[5] = {0, 0, 0, 0, 0};
			if (P.bps <= 24) {
				// Samples of up to 24 bits (25 for a side signal): no difference of up to the fourth order leaves 32 bits,
				// so the errors are the absolute values of successive differences kept biased by 2^31 (a signed value
				// and its biased form compare alike as unsigned numbers): |u - v| of two biased values is the unsigned
				// absolute difference.  Four samples' worth (< 2^31) is added up in 32 bits before it goes into the
				// 64-bit totals.
				const uint32_t BIAS = 0x80000000u;
				uint32_t p0 = 0, p1 = 0, p2 = 0, p3 = 0;
#pragma unroll
				for (int k = 0; k < 4 + FA_CHUNK; k++) {
					const uint32_t b0 = (uint32_t)d[k] ^ BIAS;
					const uint32_t b1 = (b0 - p0) ^ BIAS, b2 = (b1 - p1) ^ BIAS, b3 = (b2 - p2) ^ BIAS;
					if (k >= 4) {
						const uint32_t idx = base + (uint32_t)(k - 4);
						if (idx >= 4 && idx < bsf) {
							t32[0] = sad_u32(b0, BIAS, t32[0]);
							t32[1] = sad_u32(b0, p0, t32[1]);
							t32[2] = sad_u32(b1, p1, t32[2]);
							t32[3] = sad_u32(b2, p2, t32[3]);
							t32[4] = sad_u32(b3, p3, t32[4]);
						}
						if (P.wide_fixed && (k & 3) == 3) {
							te[0] += t32[0]; te[1] += t32[1]; te[2] += t32[2]; te[3] += t32[3]; te[4] += t32[4];
							t32[0] = t32[1] = t32[2] = t32[3] = t32[4] = 0;
						}
					}
					p0 = b0;
					p1 = b1;
					p2 = b2;
					p3 = b3;
				}
			} else {
#pragma unroll
			for (int i = 0; i < FA_CHUNK; i++) {
				const uint32_t idx = base + i;
				if (idx >= 4 && idx < bsf) {
					const uint32_t a = (uint32_t)d[4 + i], b = (uint32_t)d[3 + i], c = (uint32_t)d[2 + i], e = (uint32_t)d[1 + i], g = (uint32_t)d[i];
					const uint32_t e0 = abs32((int32_t)a), e1 = abs32((int32_t)(a - b)), e2 = abs32((int32_t)(a - 2 * b + c));
					const uint32_t e3 = abs32((int32_t)(a - 3 * b + 3 * c - e)), e4 = abs32((int32_t)(a - 4 * b + 6 * c - 4 * e + g));
					if (P.wide_fixed) {
						te[0] += e0; te[1] += e1; te[2] += e2; te[3] += e3; te[4] += e4;
					} else {
						t32[0] += e0; t32[1] += e1; t32[2] += e2; t32[3] += e3; t32[4] += e4;
					}
				}
			}
			}
			if (!P.wide_fixed)
				for (int k = 0; k < 5; k++)
					te[k] += t32[k];  // wraps like the reference's uint32 accumulators after the final mask
		}
	}
#pragma unroll
	for (int k = 0; k < 5; k++)
		te[k] = wave_sum64(te[k]);
	if ((tid & 63) == 0)
		for (int k = 0; k < 5; k++)
			sh_te[tid >> 6][k] = te[k];
	__syncthreads();
	if (tid == 0) {
		uint64_t t[5];
		for (int k = 0; k < 5; k++)
			t[k] = sh_te[0][k] + sh_te[1][k] + sh_te[2][k] + sh_te[3][k];
		stats[task] = make_stat(P, bsf, wasted, (P.do_ms && s == P.channels + 1) ? 1u : 0u, first, ne, t);
	}
}

// ---------------------------------------------------------------------------------------------
// k_analyze_quad: the same statistics for 16-bit stereo with mid/side, all four signals of a frame
// in one workgroup.  One 32-bit word holds a wide sample; it is loaded and staged once, and left,
// right, mid and side are formed in registers.  The sums are kept modulo 2^32, which is all the
// reference's 32-bit accumulators hold (fixed.c:234); streams that need the 64-bit variant take
// k_analyze.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_sum32(uint32_t v)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1)
		v += (uint32_t)__shfl_xor((int)v, o, 64);
	return v;
}

__device__ __forceinline__ void quad_split(uint32_t w, int32_t (&x)[4])
{
	const int32_t l = (int32_t)(int16_t)(w & 0xffff), r = (int32_t)w >> 16;
	x[0] = l;
	x[1] = r;
	x[2] = (l + r) >> 1;
	x[3] = l - r;
}

// the sum over the wave, in lane 63: four shifts inside each row of sixteen lanes, then row 0 onto row 1 and
// rows 0-1 onto rows 2-3 (data-parallel primitives, no LDS crossbar)
__device__ __forceinline__ uint32_t wave_total32(uint32_t v)
{
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);    // row_shr:1
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);    // row_shr:2
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);    // row_shr:4
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);    // row_shr:8
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15, rows 1 and 3
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31, rows 2 and 3
	return v;
}

// acc + |a - b| for unsigned a, b
__device__ __forceinline__ uint32_t sad_acc(uint32_t a, uint32_t b, uint32_t acc)
{
	uint32_t r;
	asm("v_sad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(acc));
	return r;
}

// Wave priority: load issue and the reductions ahead of the other workgroups' difference sums (0.130 -> 0.128 ms).
#define QPRIO(n) __builtin_amdgcn_s_setprio(n)
__global__ __launch_bounds__(FA_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_analyze_quad(FaEncParams P, FaBatch B, const void *pcm, FaSubStat *stats)
{
	const uint32_t f = blockIdx.x, tid = threadIdx.x;
	const uint32_t bsf = frame_len(B, P.blocksize, f);
	const uint32_t *words = (const uint32_t *)pcm + (uint64_t)f * P.blocksize;
	__shared__ uint32_t sh_or[4][4], sh_ne[4][4];
	__shared__ uint32_t sh_te[4][4][5];
	__shared__ uint32_t stage[ST_WORDS(4)];
	constexpr uint32_t HP = ST_HP(4);
	constexpr uint32_t NIT = (HP + FA_TILE + FA_THREADS - 1) / FA_THREADS;

	QPRIO(3);
	int32_t first[4];
	quad_split(words[0], first);
	uint32_t orv[4] = {0, 0, 0, 0}, ne[4] = {0, 0, 0, 0};
	// pass 1: OR of all samples (wasted bits) and whether they are all equal; tile 0 stays staged
	for (uint32_t t0 = 0; t0 < bsf; t0 += FA_TILE) {
		uint32_t w[NIT];
#pragma unroll
		for (uint32_t i = 0; i < NIT; i++) {
			const uint32_t j = tid + i * FA_THREADS;
			int32_t idx = (int32_t)(t0 + j) - (int32_t)HP;
			idx = idx < 0 ? 0 : ((uint32_t)idx >= bsf ? (int32_t)bsf - 1 : idx);
			w[i] = words[idx];
		}
		const uint32_t n0 = HP + ((umin32(bsf, FA_TILE) + 15) & ~15u);
#pragma unroll
		for (uint32_t i = 0; i < NIT; i++) {
			const uint32_t j = tid + i * FA_THREADS;
			const bool inside = j >= HP && t0 + j - HP < bsf;
			const uint32_t ww = inside ? w[i] : (first[0] & 0xffff) | ((uint32_t)first[1] << 16);
			int32_t x[4];
			quad_split(ww, x);
#pragma unroll
			for (int s = 0; s < 4; s++) {
				orv[s] |= (uint32_t)x[s];
				ne[s] |= (uint32_t)(x[s] != first[s]);
			}
			if (t0 == 0 && j < n0)
				stage[j + (j >> 4)] = inside ? w[i] : 0u;
		}
	}
	// samples outside the frame were replaced by the first sample above; that one is inside anyway
#pragma unroll
	for (int s = 0; s < 4; s++) {
		orv[s] = wave_or(orv[s]);
		ne[s] = wave_or(ne[s]);
		if ((tid & 63) == 0) {
			sh_or[tid >> 6][s] = orv[s];
			sh_ne[tid >> 6][s] = ne[s];
		}
	}
	__syncthreads();
	uint32_t wasted[4];
#pragma unroll
	for (int s = 0; s < 4; s++) {
		orv[s] = sh_or[0][s] | sh_or[1][s] | sh_or[2][s] | sh_or[3][s];
		ne[s] = sh_ne[0][s] | sh_ne[1][s] | sh_ne[2][s] | sh_ne[3][s];
		wasted[s] = orv[s] ? (uint32_t)__ffs((int)orv[s]) - 1u : 0u;
	}

	uint32_t te[4][5];
#pragma unroll
	for (int s = 0; s < 4; s++)
#pragma unroll
		for (int k = 0; k < 5; k++)
			te[s][k] = 0;
	if (bsf >= 4) {
		for (uint32_t tile = 0; tile * FA_TILE < bsf; tile++) {
			if (tile > 0) {
				__syncthreads();
				uint32_t w[NIT];
#pragma unroll
				for (uint32_t i = 0; i < NIT; i++) {
					const uint32_t j = tid + i * FA_THREADS;
					int32_t idx = (int32_t)(tile * FA_TILE + j) - (int32_t)HP;
					idx = idx < 0 ? 0 : ((uint32_t)idx >= bsf ? (int32_t)bsf - 1 : idx);
					w[i] = words[idx];
				}
				const uint32_t n = HP + ((umin32(bsf - tile * FA_TILE, FA_TILE) + 15) & ~15u);
#pragma unroll
				for (uint32_t i = 0; i < NIT; i++) {
					const uint32_t j = tid + i * FA_THREADS;
					const uint32_t idx = tile * FA_TILE + j - HP;
					if (j < n)
						stage[j + (j >> 4)] = idx < bsf ? w[i] : 0u;
				}
				__syncthreads();
			}
			const uint32_t base = tile * FA_TILE + tid * FA_CHUNK;
			if (base >= bsf)
				continue;
			int32_t raw[4 + FA_CHUNK];
			load_chunk_lds<4>(raw, (const int32_t *)stage);
			QPRIO(0);
			// The errors of the five fixed predictors (fixed.c:240-262) are the absolute values of the signal's
			// successive differences.  Everything is kept biased by 2^31 (a signed value and its biased form
			// compare alike as unsigned numbers), so |u - v| of two biased values is one v_sad_u32 that also
			// adds to the running total: an order costs a subtraction, the bias and the v_sad_u32 per sample.
			const bool whole = base >= 4 && base + FA_CHUNK <= bsf;   // every sample of the chunk counts (fixed.c:238 starts at the fifth)
			constexpr uint32_t BIAS = 0x80000000u;
#pragma unroll
			for (int s = 0; s < 4; s++) {
				// one sample at a time, with the previous value of every order carried along
				uint32_t p0 = 0, p1 = 0, p2 = 0, p3 = 0;
#pragma unroll
				for (int k = 0; k < 4 + FA_CHUNK; k++) {
					int32_t x[4];
					quad_split((uint32_t)raw[k], x);
					const uint32_t b0 = (uint32_t)(x[s] >> wasted[s]) ^ BIAS;
					const uint32_t b1 = (b0 - p0) ^ BIAS, b2 = (b1 - p1) ^ BIAS, b3 = (b2 - p2) ^ BIAS;
					if (k >= 4) {
						// (a sample that does not count adds |v - v| = 0)
						const uint32_t idx = base + (uint32_t)(k - 4);
						const bool counts = whole || (idx >= 4 && idx < bsf);
						te[s][0] = sad_acc(b0, counts ? BIAS : b0, te[s][0]);
						te[s][1] = sad_acc(b0, counts ? p0 : b0, te[s][1]);
						te[s][2] = sad_acc(b1, counts ? p1 : b1, te[s][2]);
						te[s][3] = sad_acc(b2, counts ? p2 : b2, te[s][3]);
						te[s][4] = sad_acc(b3, counts ? p3 : b3, te[s][4]);
					}
					p0 = b0;
					p1 = b1;
					p2 = b2;
					p3 = b3;
				}
				// (the four signals one after the other: side by side they take twice the registers)
				asm volatile("" : "+v"(te[s][4]));
			}
		}
	}
	QPRIO(3);
#pragma unroll
	for (int s = 0; s < 4; s++)
#pragma unroll
		for (int k = 0; k < 5; k++) {
			const uint32_t v = wave_total32(te[s][k]);   // in lane 63
			if ((tid & 63) == 63)
				sh_te[tid >> 6][s][k] = v;
		}
	__syncthreads();
	if (tid < 4) {
		const uint32_t s = tid;
		const uint32_t ws = s == 0 ? wasted[0] : s == 1 ? wasted[1] : s == 2 ? wasted[2] : wasted[3];
		const uint32_t nes = s == 0 ? ne[0] : s == 1 ? ne[1] : s == 2 ? ne[2] : ne[3];
		const int32_t fs = s == 0 ? first[0] : s == 1 ? first[1] : s == 2 ? first[2] : first[3];
		FaSubStat st;
		st.wasted = ws;
		st.bps = P.bps - ws + (s == 3 ? 1u : 0u);
		st.first_sample = fs >> ws;
		st.is_constant = 0;
		st.fixed_guess = 0;
		for (int k = 0; k < 5; k++)
			st.rbps[k] = 0.0f;
		if (bsf >= 4) {
			uint64_t t[5];
			for (int k = 0; k < 5; k++)
				t[k] = (uint32_t)(sh_te[0][s][k] + sh_te[1][s][k] + sh_te[2][s][k] + sh_te[3][s][k]);
			auto mn = [](uint64_t a, uint64_t b) { return a < b ? a : b; };
			uint32_t order;
			if (t[0] < mn(mn(mn(t[1], t[2]), t[3]), t[4])) order = 0;
			else if (t[1] < mn(mn(t[2], t[3]), t[4])) order = 1;
			else if (t[2] < mn(t[3], t[4])) order = 2;
			else if (t[3] < t[4]) order = 3;
			else order = 4;
			st.fixed_guess = order;
			const double n = (double)(bsf - 4);
			for (int k = 0; k < 5; k++)
				st.rbps[k] = (float)(t[k] > 0 ? log(M_LN2 * (double)t[k] / n) / M_LN2 : 0.0);
			st.is_constant = (!P.disable_constant && st.rbps[1] == 0.0f && nes == 0) ? 1u : 0u;
		}
		stats[(size_t)f * 4 + s] = st;
	}
}

// ---------------------------------------------------------------------------------------------
// k_autoc: FLAC__lpc_window_data (lpc.c:56) + FLAC__lpc_compute_autocorrelation (lpc.c:63).
// The reference adds the products of one lag in sample order into a float accumulator; a lane
// here does the same for its lags, so the sums are the same floats.
//
// Tasks (frame, signal, apodization) are taken two at a time: the windowed samples of a task pair
// sit interleaved in LDS, {a[k], b[k]}, so that one packed multiply and one packed add advance the
// same lag of both tasks.  A lane owns ONE lag of its pair (the sums are serial chains, so the
// number of lanes is all the parallelism there is) and walks the samples two steps per iteration:
// one 16-byte read for x[i], x[i+1] and one for x[i+l], x[i+l+1].  The first two waves take the even
// lags, whose second read is aligned as it stands; the other two take the odd lags and carry the
// first of their two values over from the previous read.  Positions past the frame are zero, and
// adding the resulting +-0 products leaves an accumulator unchanged.
//
// Tiles of AC_T steps are double-buffered: the global loads for the next tile are in flight while
// the current one is computed.  16-bit stereo with mid/side and a single window (the common case)
// has a dedicated staging path in which one 32-bit load yields the sample of all four signals of a
// frame, i.e. of two task pairs; stereo with mid/side out of 32-bit words has the same path with one
// 64-bit load per wide sample, and the planes of k_planes one with a 32-bit load per task (the pair of a
// row are neighbouring planes).  Any other format is staged element by element.
// ---------------------------------------------------------------------------------------------
#define AC_T 128          // steps per tile
#define AC_MAXROWS 36     // task pairs per workgroup at most (even)
#define AC_FPW 5          // frames per wave in the dedicated staging paths for stereo: ceil(AC_MAXROWS / 2 / 4)
#define AC_RPW 9          // rows per wave in the one for planes: AC_MAXROWS / 4
#define AC_NK 3           // 64-lane strips per row: AC_T + 33 + 1 <= 192

typedef float fa_f2 __attribute__((ext_vector_type(2)));
typedef float fa_f4 __attribute__((ext_vector_type(4)));

// Wave priority: staging (global loads in flight, LDS stores, the barrier) runs ahead of the other workgroups'
// multiply-add loops on the same SIMD, which only wait for LDS anyway (measured 0.157 -> 0.144 ms).
#define APRIO(n) __builtin_amdgcn_s_setprio(n)
enum { AC_PLAIN = 0, AC_QUAD16 = 1, AC_QUAD32 = 2, AC_PLANES = 3 };
template <int MODE>
__global__ __launch_bounds__(FA_THREADS) void k_autoc(FaEncParams P, FaBatch B, const void *pcm, const FaSubStat *stats,
                                                       const float *windows, float *autoc, uint32_t rows_per_block, uint32_t G)
{
	extern __shared__ __attribute__((aligned(16))) float xs[];
	const uint32_t lag = P.max_lpc_order + 1;
	const uint32_t row_pairs = (AC_T + lag + 1) & ~1u;
	const uint32_t buf_floats = rows_per_block * row_pairs * 2;
	const uint32_t tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
	const uint32_t ntasks = B.nframes * P.nsignals * P.napod;
	const uint32_t nrows = (ntasks + 1) / 2;
	const uint32_t row0 = blockIdx.x * rows_per_block;
	// G lanes per row in each half of the workgroup: lags 2 g (threads 0..127) and 2 g + 1 (128..255)
	const uint32_t odd = tid >> 7, my_j = (tid & 127) / G, my_l = 2 * ((tid & 127) % G) + odd;
	const bool active = my_j < rows_per_block && row0 + my_j < nrows && my_l < lag;
	constexpr bool quad = MODE == AC_QUAD16 || MODE == AC_QUAD32;  // rows 2f, 2f+1 = (L,R), (M,S) of frame f
	const uint32_t ntiles = (P.blocksize + AC_T - 1) / AC_T;
	const uint32_t frames_pb = rows_per_block / 2, frame0 = row0 / 2;

	// dedicated path: per frame of this wave, its length and the four wasted-bit counts
	uint32_t q_bsf[AC_FPW], q_wasted[AC_FPW];
	if (quad) {
#pragma unroll
		for (uint32_t n = 0; n < AC_FPW; n++) {
			const uint32_t fi = wv + 4 * n, f = frame0 + fi;
			q_bsf[n] = 0;
			q_wasted[n] = 0;
			if (fi < frames_pb && f < B.nframes) {
				q_bsf[n] = frame_len(B, P.blocksize, f);
				const FaSubStat *st = stats + (size_t)f * 4;
				q_wasted[n] = st[0].wasted | (st[1].wasted << 8) | (st[2].wasted << 16) | (st[3].wasted << 24);
			}
		}
	}
	uint32_t q_w[AC_FPW][AC_NK], q_w2[MODE == AC_QUAD32 ? AC_FPW : 1][AC_NK];
	float q_win[AC_FPW][AC_NK];
	auto quad_load = [&](uint32_t tile) {
		const uint32_t c0 = tile * AC_T;
#pragma unroll
		for (uint32_t n = 0; n < AC_FPW; n++) {
			const uint32_t f = frame0 + wv + 4 * n;
			if (wv + 4 * n >= frames_pb)
				continue;
#pragma unroll
			for (uint32_t m = 0; m < AC_NK; m++) {
				const uint32_t idx = c0 + lane + 64 * m;
				const uint32_t ci = q_bsf[n] ? umin32(idx, q_bsf[n] - 1) : 0;
				if (MODE == AC_QUAD32) {
					const int2 lr = q_bsf[n] ? ((const int2 *)pcm)[(uint64_t)f * P.blocksize + ci] : int2{0, 0};
					q_w[n][m] = (uint32_t)lr.x;
					q_w2[n][m] = (uint32_t)lr.y;
				} else
					q_w[n][m] = q_bsf[n] ? ((const uint32_t *)pcm)[(uint64_t)f * P.blocksize + ci] : 0u;
				q_win[n][m] = windows[umin32(idx, P.blocksize - 1)];
			}
		}
	};
	auto quad_store = [&](uint32_t tile, float *dst) {
		const uint32_t c0 = tile * AC_T;
#pragma unroll
		for (uint32_t n = 0; n < AC_FPW; n++) {
			const uint32_t fi = wv + 4 * n;
			if (fi >= frames_pb)
				continue;
#pragma unroll
			for (uint32_t m = 0; m < AC_NK; m++) {
				const uint32_t k = lane + 64 * m;
				if (k >= row_pairs)
					continue;
				fa_f2 lr = {0.0f, 0.0f}, ms = {0.0f, 0.0f};
				if (c0 + k < q_bsf[n]) {
					const uint32_t w = q_w[n][m];
					const int32_t l = MODE == AC_QUAD32 ? (int32_t)w : (int32_t)(int16_t)(w & 0xffff);
					const int32_t r = MODE == AC_QUAD32 ? (int32_t)q_w2[MODE == AC_QUAD32 ? n : 0][m] : (int32_t)w >> 16;
					const float win = q_win[n][m];
					lr.x = __fmul_rn((float)(l >> (q_wasted[n] & 0xff)), win);
					lr.y = __fmul_rn((float)(r >> ((q_wasted[n] >> 8) & 0xff)), win);
					ms.x = __fmul_rn((float)(((l + r) >> 1) >> ((q_wasted[n] >> 16) & 0xff)), win);
					ms.y = __fmul_rn((float)((l - r) >> (q_wasted[n] >> 24)), win);
				}
				*(fa_f2 *)(dst + ((size_t)(2 * fi) * row_pairs + k) * 2) = lr;
				*(fa_f2 *)(dst + ((size_t)(2 * fi + 1) * row_pairs + k) * 2) = ms;
			}
		}
	};
	// planes: per row of this wave, the lengths of its two tasks' frames (0: no such task) and their wasted-bit counts
	uint32_t r_bsf[AC_RPW], r_wasted[AC_RPW];
	int32_t r_v[MODE == AC_PLANES ? AC_RPW : 1][AC_NK][2];
	float r_win[AC_NK];
	if (MODE == AC_PLANES) {
#pragma unroll
		for (uint32_t n = 0; n < AC_RPW; n++) {
			const uint32_t j = wv + 4 * n, ta = 2 * (row0 + j);
			r_bsf[n] = 0;
			r_wasted[n] = 0;
			if (j < rows_per_block && ta < ntasks) {
				r_bsf[n] = frame_len(B, P.blocksize, ta / P.nsignals);
				r_wasted[n] = stats[ta].wasted;
				if (ta + 1 < ntasks) {
					r_bsf[n] |= frame_len(B, P.blocksize, (ta + 1) / P.nsignals) << 16;
					r_wasted[n] |= stats[ta + 1].wasted << 8;
				}
			}
		}
	}
	auto rows_load = [&](uint32_t tile) {
		const uint32_t c0 = tile * AC_T;
#pragma unroll
		for (uint32_t m = 0; m < AC_NK; m++)
			r_win[m] = windows[umin32(c0 + lane + 64 * m, P.blocksize - 1)];
#pragma unroll
		for (uint32_t n = 0; n < AC_RPW; n++) {
			const uint32_t j = wv + 4 * n;
			if (j >= rows_per_block)
				continue;
			const int32_t *pa = (const int32_t *)pcm + (size_t)(2 * (row0 + j)) * P.blocksize;
			const uint32_t la = r_bsf[n] & 0xffff, lb = r_bsf[n] >> 16;
#pragma unroll
			for (uint32_t m = 0; m < AC_NK; m++) {
				const uint32_t idx = c0 + lane + 64 * m;
				r_v[MODE == AC_PLANES ? n : 0][m][0] = la ? pa[umin32(idx, la - 1)] : 0;
				r_v[MODE == AC_PLANES ? n : 0][m][1] = lb ? pa[P.blocksize + umin32(idx, lb - 1)] : 0;
			}
		}
	};
	auto rows_store = [&](uint32_t tile, float *dst) {
		const uint32_t c0 = tile * AC_T;
#pragma unroll
		for (uint32_t n = 0; n < AC_RPW; n++) {
			const uint32_t j = wv + 4 * n;
			if (j >= rows_per_block)
				continue;
			const uint32_t la = r_bsf[n] & 0xffff, lb = r_bsf[n] >> 16;
#pragma unroll
			for (uint32_t m = 0; m < AC_NK; m++) {
				const uint32_t k = lane + 64 * m;
				if (k >= row_pairs)
					continue;
				fa_f2 ab = {0.0f, 0.0f};
				if (c0 + k < la)
					ab.x = __fmul_rn((float)(r_v[MODE == AC_PLANES ? n : 0][m][0] >> (r_wasted[n] & 0xff)), r_win[m]);
				if (c0 + k < lb)
					ab.y = __fmul_rn((float)(r_v[MODE == AC_PLANES ? n : 0][m][1] >> (r_wasted[n] >> 8)), r_win[m]);
				*(fa_f2 *)(dst + ((size_t)j * row_pairs + k) * 2) = ab;
			}
		}
	};
	// general path: any format, element by element
	auto plain_stage = [&](uint32_t tile, float *dst) {
		const uint32_t c0 = tile * AC_T;
		for (uint32_t e = tid; e < buf_floats; e += FA_THREADS) {
			const uint32_t h = e & 1, pk = e >> 1;
			const uint32_t j = pk / row_pairs, k = pk % row_pairs;
			const uint32_t task = 2 * (row0 + j) + h;
			float v = 0.0f;
			if (task < ntasks) {
				const uint32_t sub = task / P.napod, a = task % P.napod;
				const uint32_t f = sub / P.nsignals, s = sub % P.nsignals;
				const uint32_t bsf = frame_len(B, P.blocksize, f);
				const uint32_t idx = c0 + k;
				if (idx < bsf) {
					// (planes: the frame's block and the position inside it are at hand, no division needed to find them)
					const int32_t raw = B.planar ? frame_plane_signal((const int32_t *)pcm + (size_t)f * P.blocksize * P.channels + idx, P.channels, P.blocksize, s)
					                             : fetch_signal(pcm, P.channels, B.sample_bytes, fa_layout(B), (uint64_t)f * P.blocksize + idx, s);
					const int32_t smp = raw >> stats[sub].wasted;
					v = __fmul_rn((float)smp, windows[(size_t)a * P.blocksize + idx]);
				}
			}
			dst[e] = v;
		}
	};

	APRIO(3);
	if (quad) {
		quad_load(0);
		quad_store(0, xs);
	} else if (MODE == AC_PLANES) {
		rows_load(0);
		rows_store(0, xs);
	} else
		plain_stage(0, xs);
	__syncthreads();

	fa_f2 acc = {0.0f, 0.0f};
	for (uint32_t t = 0; t < ntiles; t++) {
		const float *cur_buf = xs + (t & 1) * buf_floats;
		float *next_buf = xs + ((t + 1) & 1) * buf_floats;
		const bool more = t + 1 < ntiles;
		if (more && quad)
			quad_load(t + 1);
		if (more && MODE == AC_PLANES)
			rows_load(t + 1);
		APRIO(0);
		if (active) {
			const float *row = cur_buf + (size_t)my_j * row_pairs * 2;
			if (!odd) {
#pragma unroll 4
				for (uint32_t i = 0; i < AC_T; i += 2) {
					const fa_f4 a = *(const fa_f4 *)(row + 2 * i);             // pairs i, i+1
					const fa_f4 b = *(const fa_f4 *)(row + 2 * (i + my_l));    // pairs i+l, i+l+1
					const fa_f2 a0 = {a.x, a.y}, a1 = {a.z, a.w}, b0 = {b.x, b.y}, b1 = {b.z, b.w};
					acc = acc + a0 * b0;
					acc = acc + a1 * b1;
				}
			} else {
				fa_f2 win = *(const fa_f2 *)(row + 2 * my_l);                      // pair l
#pragma unroll 4
				for (uint32_t i = 0; i < AC_T; i += 2) {
					const fa_f4 a = *(const fa_f4 *)(row + 2 * i);                 // pairs i, i+1
					const fa_f4 b = *(const fa_f4 *)(row + 2 * (i + my_l + 1));    // pairs i+l+1, i+l+2
					const fa_f2 a0 = {a.x, a.y}, a1 = {a.z, a.w}, b1 = {b.x, b.y};
					acc = acc + a0 * win;
					acc = acc + a1 * b1;
					win = fa_f2{b.z, b.w};
				}
			}
		}
		APRIO(3);
		if (more) {
			if (quad)
				quad_store(t + 1, next_buf);
			else if (MODE == AC_PLANES)
				rows_store(t + 1, next_buf);
			else
				plain_stage(t + 1, next_buf);
		}
		__syncthreads();
	}
	if (active) {
		const uint32_t task = 2 * (row0 + my_j);
		float *o = autoc + (size_t)task * (FA_MAX_ORDER + 1);
		o[my_l] = acc.x;
		if (task + 1 < ntasks)
			o[FA_MAX_ORDER + 1 + my_l] = acc.y;
	}
}

// ---------------------------------------------------------------------------------------------
// k_lpc: per (frame, signal) lane: FLAC__lpc_compute_lp_coefficients (lpc.c:112),
// FLAC__lpc_compute_best_order (lpc.c:1353), expected bits (lpc.c:1325-1351), and
// FLAC__lpc_quantize_coefficients (lpc.c:156) for every candidate process_subframe_ would try
// (stream_encoder.c:3292-3382).  Per-lane arrays live in LDS, indexed [element][lane].
// ---------------------------------------------------------------------------------------------
#define LPC_LANES 64

__device__ __forceinline__ double expected_bits_scaled(double lpc_error, double error_scale)
{
	if (lpc_error > 0.0) {
		const double bps = (double)0.5 * log(error_scale * lpc_error) / M_LN2;
		return bps >= 0.0 ? bps : 0.0;
	}
	if (lpc_error < 0.0)
		return 1e32;
	return 0.0;
}

// One Levinson-Durbin step i (0-based): updates lpc[0..i] and err.  lpc is the lane's LDS column.
__device__ __forceinline__ void levinson_step(uint32_t i, const float *ac, double *lpc, uint32_t lane, double &err)
{
	double r = -(double)ac[(i + 1) * LPC_LANES + lane];
	for (uint32_t j = 0; j < i; j++)
		r -= lpc[j * LPC_LANES + lane] * (double)ac[(i - j) * LPC_LANES + lane];
	r /= err;
	lpc[i * LPC_LANES + lane] = r;
	uint32_t j;
	for (j = 0; j < (i >> 1); j++) {
		const double tmp = lpc[j * LPC_LANES + lane];
		lpc[j * LPC_LANES + lane] = tmp + r * lpc[(i - 1 - j) * LPC_LANES + lane];
		lpc[(i - 1 - j) * LPC_LANES + lane] += r * tmp;
	}
	if (i & 1)
		lpc[j * LPC_LANES + lane] += lpc[j * LPC_LANES + lane] * r;
	err *= (1.0 - r * r);
}

__global__ __launch_bounds__(LPC_LANES) void k_lpc(FaEncParams P, FaBatch B, const FaSubStat *stats, const float *autoc,
                                                   FaLpcCand *cands, int32_t *coefs)
{
	__shared__ double lpc[FA_MAX_ORDER * LPC_LANES];
	__shared__ double errs[FA_MAX_ORDER * LPC_LANES];
	__shared__ float ac[(FA_MAX_ORDER + 1) * LPC_LANES];
	const uint32_t lane = threadIdx.x;
	const uint32_t sub = blockIdx.x * LPC_LANES + lane;
	const uint32_t nsub = B.nframes * P.nsignals;
	if (sub >= nsub)
		return;
	const uint32_t f = sub / P.nsignals;
	const uint32_t bsf = frame_len(B, P.blocksize, f);
	const FaSubStat st = stats[sub];
	FaLpcCand *my_cands = cands + (size_t)sub * P.ncand;
	int32_t *my_coefs = coefs + (size_t)sub * P.ncand * P.max_lpc_order;
	FaLpcCand none;
	none.valid = 0; none.order = 0; none.precision = 0; none.shift = 0; none.rice = 0; none.wide = 0; none.sumq32 = 0;
	for (uint32_t c = 0; c < P.ncand; c++)
		my_cands[c] = none;
	if (bsf < 4 || st.is_constant || P.max_lpc_order == 0)
		return;
	uint32_t max_lpc = P.max_lpc_order >= bsf ? bsf - 1 : P.max_lpc_order;
	if (max_lpc == 0)
		return;
	const uint32_t sbps = st.bps;

	for (uint32_t a = 0; a < P.napod; a++) {
		const float *src = autoc + (size_t)(sub * P.napod + a) * (FA_MAX_ORDER + 1);
		for (uint32_t l = 0; l <= max_lpc; l++)
			ac[l * LPC_LANES + lane] = src[l];
		if (ac[lane] == 0.0f)
			continue;  // stream_encoder.c:3303
		// pass 1: prediction errors of every order (lpc.c:122-153)
		double err = (double)ac[lane];
		for (uint32_t i = 0; i < max_lpc; i++) {
			levinson_step(i, ac, lpc, lane, err);
			errs[i * LPC_LANES + lane] = err;
			if (err == 0.0) {
				max_lpc = i + 1;
				break;
			}
		}
		uint32_t min_lpc;
		if (P.exhaustive)
			min_lpc = 1;
		else {
			// lpc.c:1353
			const uint32_t overhead = sbps + (P.prec_search ? 5u : P.qlp_precision);
			const double scale = 0.5 * M_LN2 * M_LN2 / (double)bsf;
			uint32_t best = 0;
			double best_bits = (double)(unsigned)(-1);
			for (uint32_t idx = 0, order = 1; idx < max_lpc; idx++, order++) {
				const double bits = expected_bits_scaled(errs[idx * LPC_LANES + lane], scale) * (double)(bsf - order) + (double)(order * overhead);
				if (bits < best_bits) {
					best = idx;
					best_bits = bits;
				}
			}
			min_lpc = max_lpc = best + 1;
		}
		if (max_lpc >= bsf)
			max_lpc = bsf - 1;
		// pass 2: the same recursion again; quantise at the orders that are tried
		err = (double)ac[lane];
		for (uint32_t i = 0; i < max_lpc; i++) {
			levinson_step(i, ac, lpc, lane, err);
			const uint32_t order = i + 1;
			if (order < min_lpc)
				continue;
			const double scale = 0.5 * M_LN2 * M_LN2 / (double)(bsf - order);
			const double lbps = expected_bits_scaled(errs[i * LPC_LANES + lane], scale);
			if (lbps >= (double)sbps)
				continue;  // stream_encoder.c:3326
			uint32_t rice = (lbps > 0.0) ? (uint32_t)(lbps + 0.5) : 0u;
			rice++;
			if (rice >= P.rice_limit)
				rice = P.rice_limit - 1;
			uint32_t pmin, pmax;
			if (P.prec_search) {
				pmin = 5;
				if (sbps <= 17) {
					pmax = umin32(32 - sbps - order, 15u);
					pmax = umax32(pmax, pmin);
				} else
					pmax = 15;
			} else
				pmin = pmax = P.qlp_precision;
			// cmax = max |coefficient| of this order (lpc.c:172-177); the coefficients are
			// the negated recursion state rounded to float (lpc.c:145)
			double cmax = 0.0;
			for (uint32_t j = 0; j <= i; j++) {
				const double dd = fabs((double)(float)(-lpc[j * LPC_LANES + lane]));
				if (dd > cmax)
					cmax = dd;
			}
			for (uint32_t prec = pmin; prec <= pmax; prec++) {
				const uint32_t slot = (a * P.max_lpc_order + i) * P.nprec + (P.prec_search ? prec - 5 : 0);
				uint32_t precision = prec;
				if (sbps <= 16)
					precision = umin32(precision, 32 - sbps - ilog2u(order));  // stream_encoder.c:3582-3586
				// lpc.c:156
				const uint32_t pm1 = precision - 1;
				int32_t qmax = 1 << pm1;
				const int32_t qmin = -qmax;
				qmax--;
				if (cmax <= 0.0)
					continue;
				int log2cmax;
				(void)frexp(cmax, &log2cmax);
				log2cmax--;
				int shift = (int)pm1 - log2cmax - 1;
				if (shift > 15)
					shift = 15;
				else if (shift < -16)
					continue;
				double qerr = 0.0;
				uint32_t sumq = 0;
				int32_t *q = my_coefs + (size_t)slot * P.max_lpc_order;
				for (uint32_t j = 0; j <= i; j++) {
					const float c = (float)(-lpc[j * LPC_LANES + lane]);
					if (shift >= 0)
						qerr += (double)(c * (float)(1 << shift));
					else
						qerr += (double)(c / (float)(1 << (-shift)));
					int32_t v = qerr >= 0.0 ? (int32_t)(qerr + 0.5) : (int32_t)(qerr - 0.5);
					if (v > qmax) v = qmax;
					else if (v < qmin) v = qmin;
					qerr -= v;
					q[j] = v;
					sumq += abs32(v);
				}
				FaLpcCand cd;
				cd.valid = 1;
				cd.order = (uint8_t)order;
				cd.precision = (uint8_t)precision;
				cd.shift = (int8_t)(shift < 0 ? 0 : shift);
				cd.rice = (uint8_t)rice;
				cd.wide = (sbps + precision + ilog2u(order) <= 32) ? 0 : 1;  // stream_encoder.c:3592
				cd.sumq32 = (uint16_t)((sumq + 31) >> 5);  // at most 32 * 2^14 / 32
				my_cands[slot] = cd;
			}
		}
	}
}
// ---------------------------------------------------------------------------------------------
// k_decide: channel assignment (stream_encoder.c:3047-3082) incl. the loose mid/side rule
// (:2952-2962): within a period every frame follows its leader frame's decision.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t assignment_argmin(const FaPlan *p /* L,R,M,S */)
{
	const uint32_t b0 = p[0].bits + p[1].bits, b1 = p[0].bits + p[3].bits, b2 = p[1].bits + p[3].bits, b3 = p[2].bits + p[3].bits;
	uint32_t ca = 0, mn = b0;
	if (b1 < mn) { mn = b1; ca = 1; }
	if (b2 < mn) { mn = b2; ca = 2; }
	if (b3 < mn) { mn = b3; ca = 3; }
	return ca;
}

__global__ void k_decide(FaEncParams P, FaBatch B, const FaPlan *plans, uint32_t *assign, uint32_t *loose_out, unsigned long long *tickets)
{
	const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
	if (f >= B.nframes)
		return;
	tickets[f] = 0;   // nothing known yet about where frame f goes (k_pack_frame, k_pack_sub)
	{
		// nor about the pieces of it (k_pack_sub: two words per channel, behind the tickets)
		uint2 *sub = (uint2 *)(tickets + B.nframes) + (size_t)f * P.channels;
		for (uint32_t c = 0; c < P.channels; c++)
			sub[c] = make_uint2(0, 0);
	}
	uint32_t ca = 0;
	if (P.do_ms) {
		if (P.loose_ms) {
			const uint32_t gnum = B.first_frame + f;
			const uint32_t phase = gnum % P.loose_frames;
			if (phase == 0)
				ca = assignment_argmin(plans + (size_t)f * P.nsignals);
			else {
				uint32_t lead;
				if (phase > f)
					lead = B.loose_carry;  // the leader was in an earlier batch
				else
					lead = assignment_argmin(plans + (size_t)(f - phase) * P.nsignals);
				ca = lead == 0 ? 0 : 3;
			}
		} else
			ca = assignment_argmin(plans + (size_t)f * P.nsignals);
	}
	assign[f] = ca;
	if (f == B.nframes - 1)
		*loose_out = ca;
}
// ---------------------------------------------------------------------------------------------
// k_scan: exclusive prefix sum of the frame sizes (single block)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_scan(const uint32_t *frame_bytes, uint32_t n, uint64_t *offsets)
{
	__shared__ uint64_t wsum[16];
	__shared__ uint64_t carry;
	const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	if (tid == 0)
		carry = 0;
	__syncthreads();
	for (uint32_t base = 0; base < n; base += 1024) {
		const uint32_t i = base + tid;
		const uint64_t v = i < n ? frame_bytes[i] : 0;
		uint64_t inc = v;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const uint64_t t = __shfl_up(inc, o, 64);
			if (lane >= (uint32_t)o)
				inc += t;
		}
		if (lane == 63)
			wsum[wv] = inc;
		__syncthreads();
		uint64_t off = carry;
		for (uint32_t w = 0; w < wv; w++)
			off += wsum[w];
		if (i < n)
			offsets[i] = off + inc - v;
		__syncthreads();
		if (tid == 1023)
			carry = off + inc;
		__syncthreads();
	}
	if (tid == 0)
		offsets[n] = carry;
}

// ---------------------------------------------------------------------------------------------
// k_compact: frame slots -> back-to-back frames
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(FA_THREADS) void k_compact(const uint32_t *slots, uint32_t slot_bytes, const uint32_t *frame_bytes,
                                                        const uint64_t *offsets, uint8_t *out, uint64_t out_capacity, uint32_t *errflag)
{
	const uint32_t f = blockIdx.x, tid = threadIdx.x;
	const uint32_t n = frame_bytes[f];
	const uint64_t off = offsets[f];
	if (off + n > out_capacity) {
		if (tid == 0)
			atomicOr(errflag, 2u);
		return;
	}
	const uint8_t *src = (const uint8_t *)slots + (size_t)f * slot_bytes;
	uint8_t *dst = out + off;
	// leading bytes up to 4-byte alignment of dst, then aligned words, then the tail
	const uint32_t head = umin32((uint32_t)((4 - (off & 3)) & 3), n);
	if (tid < head)
		dst[tid] = src[tid];
	const uint32_t nwords = (n - head) / 4;
	uint32_t *dw = (uint32_t *)(dst + head);
	for (uint32_t w = tid; w < nwords; w += FA_THREADS) {
		const uint8_t *p = src + head + 4 * (size_t)w;
		// src is 4-byte aligned only when head == 0: assemble from two aligned words
		const uint32_t sh = (uint32_t)((uintptr_t)p & 3);
		const uint32_t *pa = (const uint32_t *)(p - sh);
		uint32_t v = pa[0];
		if (sh)
			v = (v >> (8 * sh)) | (pa[1] << (32 - 8 * sh));
		dw[w] = v;
	}
	const uint32_t done = head + nwords * 4;
	if (tid < n - done)
		dst[done + tid] = src[done + tid];
}

// ---------------------------------------------------------------------------------------------
// k_widen: host int32 samples are already int32; helper for the decoder tests is elsewhere.
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
void fa_launch_planes(const FaEncParams &P, const FaBatch &B, const void *pcm, int32_t *planes, hipStream_t st)
{
	const uint32_t tiles = (P.blocksize + 255) / 256;
	hipLaunchKernelGGL(k_planes, dim3(B.nframes * tiles), dim3(256), 0, st, P, B, pcm, planes, tiles);
}

size_t fa_stat_acc_bytes(uint32_t nframes, uint32_t nsignals) { return sizeof(FaStatAcc) * (size_t)nframes * nsignals; }

// Whether k_planes_stats can make this stream's statistics (then fa_launch_analyze is not needed).
bool fa_planes_stats_fit(const FaEncParams &P)
{
	return P.bps <= 24 && P.channels <= 8 && (P.nsignals == P.channels || (P.channels == 2 && P.nsignals == 4));
}

// planes: null = the statistics alone (the kernels behind read the samples as they lie)
void fa_launch_planes_stats(const FaEncParams &P, const FaBatch &B, const void *pcm, int32_t *planes, void *acc, FaSubStat *stats, hipStream_t st)
{
	const uint32_t cps = P.nsignals > 4 ? 32u : P.nsignals > 2 ? 64u : P.nsignals > 1 ? 128u : 256u;
	const uint32_t tiles = (P.blocksize + 16 * cps - 1) / (16 * cps);
	const dim3 grid(B.nframes * tiles);
	(void)hipMemsetAsync(acc, 0, fa_stat_acc_bytes(B.nframes, P.nsignals), st);
	switch (cps) {
	case 32: hipLaunchKernelGGL(k_planes_stats<32>, grid, dim3(256), 0, st, P, B, pcm, planes, (FaStatAcc *)acc, tiles); break;
	case 64: hipLaunchKernelGGL(k_planes_stats<64>, grid, dim3(256), 0, st, P, B, pcm, planes, (FaStatAcc *)acc, tiles); break;
	case 128: hipLaunchKernelGGL(k_planes_stats<128>, grid, dim3(256), 0, st, P, B, pcm, planes, (FaStatAcc *)acc, tiles); break;
	default: hipLaunchKernelGGL(k_planes_stats<256>, grid, dim3(256), 0, st, P, B, pcm, planes, (FaStatAcc *)acc, tiles); break;
	}
	hipLaunchKernelGGL(k_stats_finish, dim3((B.nframes * P.nsignals + 255) / 256), dim3(256), 0, st, P, B, (const FaStatAcc *)acc, stats);
}

void fa_launch_analyze(const FaEncParams &P, const FaBatch &B, const void *pcm, FaSubStat *stats, hipStream_t st)
{
	if (B.pairs && P.nsignals == 4 && !P.wide_fixed)
		hipLaunchKernelGGL(k_analyze_quad, dim3(B.nframes), dim3(FA_THREADS), 0, st, P, B, pcm, stats);
	else
		hipLaunchKernelGGL(k_analyze, dim3((B.nframes + 7) / 8 * 8 * P.nsignals), dim3(FA_THREADS), 0, st, P, B, pcm, stats);
}

void fa_launch_autoc(const FaEncParams &P, const FaBatch &B, const void *pcm, const FaSubStat *stats, const float *windows, float *autoc,
                     hipStream_t st)
{
	const uint32_t lag = P.max_lpc_order + 1;
	const uint32_t G = (lag + 1) / 2;  // lanes per task pair in each half of a workgroup (even lags, odd lags)
	uint32_t rows = (FA_THREADS / 2) / G;
	if (rows > AC_MAXROWS)
		rows = AC_MAXROWS;
	// both tile buffers of a workgroup within 48 KiB, so that three workgroups share a compute unit
	const uint32_t row_bytes = ((AC_T + lag + 1) & ~1u) * 2 * sizeof(float);
	if (rows > 49152 / (2 * row_bytes))
		rows = 49152 / (2 * row_bytes);
	rows &= ~1u;
	const uint32_t ntasks = B.nframes * P.nsignals * P.napod;
	const uint32_t nrows = (ntasks + 1) / 2;
	const size_t lds = (size_t)2 * rows * row_bytes;
	const dim3 grid((nrows + rows - 1) / rows), block(FA_THREADS);
	// (a frame is at most 65 535 samples: the planes' path keeps two lengths in a word)
	const bool ms4 = P.nsignals == 4 && P.napod == 1;
	if (B.pairs && ms4)
		hipLaunchKernelGGL(k_autoc<AC_QUAD16>, grid, block, lds, st, P, B, pcm, stats, windows, autoc, rows, G);
	else if (!B.pairs && !B.planar && ms4 && P.channels == 2 && B.sample_bytes == 4 && ((uintptr_t)pcm & 7) == 0)
		hipLaunchKernelGGL(k_autoc<AC_QUAD32>, grid, block, lds, st, P, B, pcm, stats, windows, autoc, rows, G);
	else if (B.planar && P.napod == 1 && P.nsignals == P.channels && P.blocksize <= 65535)
		hipLaunchKernelGGL(k_autoc<AC_PLANES>, grid, block, lds, st, P, B, pcm, stats, windows, autoc, rows, G);
	else
		hipLaunchKernelGGL(k_autoc<AC_PLAIN>, grid, block, lds, st, P, B, pcm, stats, windows, autoc, rows, G);
}

void fa_launch_lpc(const FaEncParams &P, const FaBatch &B, const FaSubStat *stats, const float *autoc, FaLpcCand *cands, int32_t *coefs,
                   hipStream_t st)
{
	const uint32_t nsub = B.nframes * P.nsignals;
	hipLaunchKernelGGL(k_lpc, dim3((nsub + LPC_LANES - 1) / LPC_LANES), dim3(LPC_LANES), 0, st, P, B, stats, autoc, cands, coefs);
}

void fa_launch_decide(const FaEncParams &P, const FaBatch &B, const FaPlan *plans, uint32_t *assign, uint32_t *loose_out, unsigned long long *tickets,
                      hipStream_t st)
{
	hipLaunchKernelGGL(k_decide, dim3((B.nframes + 255) / 256), dim3(256), 0, st, P, B, plans, assign, loose_out, tickets);
}

void fa_launch_scan(const uint32_t *frame_bytes, uint32_t n, uint64_t *offsets, hipStream_t st)
{
	hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, st, frame_bytes, n, offsets);
}

void fa_launch_compact(const FaEncParams &P, const FaBatch &B, const uint32_t *slots, const uint32_t *frame_bytes, const uint64_t *offsets,
                       uint8_t *out, uint64_t out_capacity, uint32_t *errflag, hipStream_t st)
{
	hipLaunchKernelGGL(k_compact, dim3(B.nframes), dim3(FA_THREADS), 0, st, slots, P.slot_bytes, frame_bytes, offsets, out, out_capacity, errflag);
}
