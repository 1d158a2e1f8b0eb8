// FLAC frame encoder kernels for gfx950 (MI355X).
//
// One batch = F consecutive frames of one stream.  Work decomposition:
//   k_analyze   grid F*S  (S = input channels [+ mid + side])   wasted bits, fixed-predictor sums
//   k_autoc     grid ceil(F*S*A / tasks-per-block)              windowed autocorrelation, exact order
//   k_lpc       lane per (frame, signal)                         Levinson-Durbin, order guess, quantise
//   k_search    grid F*S                                         residual + Rice partition search
//   k_decide    lane per frame                                   channel assignment
//   k_pack      grid F                                           bit packing into frame slots, CRCs
//   k_scan      one block                                        frame byte offsets
//   k_compact   grid F                                           slots -> contiguous stream bytes
//
// Arithmetic follows the reference (FLAC 1.2.1) operation for operation where rounding or
// wraparound is observable; each kernel says which reference function it stands in for.
// Build with -ffp-contract=off: the reference's float/double expressions are not fused.
#include <hip/hip_runtime.h>

#include "flac_amd_internal.h"
#include "enc_kernels.h"

#ifndef M_LN2
#define M_LN2 0.69314718055994530942
#endif

// ---------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------
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
// (stream_encoder.c:1991-1992, arithmetic shift, not a division).
__device__ __forceinline__ int32_t fetch_signal(const void *pcm, uint32_t channels, uint32_t sample_bytes, uint64_t wide, uint32_t s)
{
	if (s < channels)
		return read_pcm(pcm, sample_bytes, wide * channels + s);
	const int32_t l = read_pcm(pcm, sample_bytes, wide * channels);
	const int32_t r = read_pcm(pcm, sample_bytes, wide * channels + 1);
	return s == channels ? ((l + r) >> 1) : (l - r);
}

// d[k] = sample (base - HIST + k) of the frame, shifted right by the wasted bits; 0 outside the frame.
template <int HIST>
__device__ __forceinline__ void load_chunk(int32_t (&d)[HIST + FA_CHUNK], const void *pcm, uint32_t channels, uint32_t sample_bytes,
                                           uint64_t frame_start, uint32_t bsf, uint32_t s, uint32_t wasted, int32_t base)
{
#pragma unroll
	for (int k = 0; k < HIST + FA_CHUNK; k++) {
		const int32_t idx = base - HIST + k;
		int32_t v = 0;
		if (idx >= 0 && (uint32_t)idx < bsf)
			v = fetch_signal(pcm, channels, sample_bytes, frame_start + (uint32_t)idx, s) >> wasted;
		d[k] = v;
	}
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

// ---------------------------------------------------------------------------------------------
// k_analyze: get_wasted_bits_ (stream_encoder.c:4108) + FLAC__fixed_compute_best_predictor[_wide]
// (fixed.c:224,282) + the all-samples-equal test of process_subframe_ (:3210-3227)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(FA_THREADS) void k_analyze(FaEncParams P, FaBatch B, const void *pcm, FaSubStat *stats)
{
	const uint32_t task = blockIdx.x;
	const uint32_t f = task / P.nsignals, s = task % P.nsignals;
	const uint32_t tid = threadIdx.x;
	const uint32_t bsf = frame_len(B, P.blocksize, f);
	const uint64_t frame_start = (uint64_t)f * P.blocksize;
	__shared__ uint32_t sh_or[4], sh_ne[4];
	__shared__ uint64_t sh_te[4][5];

	const int32_t first = fetch_signal(pcm, P.channels, B.sample_bytes, frame_start, s);
	uint32_t orv = 0, ne = 0;
	for (uint32_t i = tid; i < bsf; i += FA_THREADS) {
		const int32_t x = fetch_signal(pcm, P.channels, B.sample_bytes, frame_start + i, s);
		orv |= (uint32_t)x;
		ne |= (uint32_t)(x != first);
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
			int32_t d[4 + FA_CHUNK];
			const int32_t base = (int32_t)(tile * FA_TILE + tid * FA_CHUNK);
			if ((uint32_t)base >= bsf)
				break;
			load_chunk<4>(d, pcm, P.channels, B.sample_bytes, frame_start, bsf, s, wasted, base);
#pragma unroll
			for (int i = 0; i < FA_CHUNK; i++) {
				const uint32_t idx = (uint32_t)base + i;
				if (idx >= 4 && idx < bsf) {
					const uint32_t a = (uint32_t)d[4 + i], b = (uint32_t)d[3 + i], c = (uint32_t)d[2 + i], e = (uint32_t)d[1 + i], g = (uint32_t)d[i];
					te[0] += abs32((int32_t)a);
					te[1] += abs32((int32_t)(a - b));
					te[2] += abs32((int32_t)(a - 2 * b + c));
					te[3] += abs32((int32_t)(a - 3 * b + 3 * c - e));
					te[4] += abs32((int32_t)(a - 4 * b + 6 * c - 4 * e + g));
				}
			}
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
		FaSubStat st;
		st.wasted = wasted;
		st.bps = P.bps - wasted + ((P.do_ms && s == P.channels + 1) ? 1u : 0u);
		st.first_sample = first >> wasted;
		st.is_constant = 0;
		st.fixed_guess = 0;
		for (int k = 0; k < 5; k++)
			st.rbps[k] = 0.0f;
		if (bsf >= 4) {
			uint64_t t[5];
			for (int k = 0; k < 5; k++) {
				t[k] = sh_te[0][k] + sh_te[1][k] + sh_te[2][k] + sh_te[3][k];
				if (!P.wide_fixed)
					t[k] &= 0xffffffffull;  // uint32 accumulators wrap (fixed.c:234)
			}
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
		stats[task] = st;
	}
}

// ---------------------------------------------------------------------------------------------
// k_autoc: FLAC__lpc_window_data (lpc.c:56) + FLAC__lpc_compute_autocorrelation (lpc.c:63).
// The reference adds the products of one lag in sample order into a float accumulator; each
// (task, lag) lane here walks the samples in that same order, so the sums are the same floats.
// Samples are staged through LDS in tiles; positions past the frame are zero, and adding the
// resulting +-0 products leaves an accumulator unchanged.
// ---------------------------------------------------------------------------------------------
#define AC_TILE 256
#define AC_MAXTASKS 16
// Row stride AC_TILE + lag: AC_TILE is a multiple of 32, so lane (task j, lag l) = thread j*lag + l
// reads bank (thread + sample) mod 32 and a half-wave never collides.
#define AC_LDS_FLOATS (AC_MAXTASKS * (AC_TILE + FA_MAX_ORDER + 1))

__global__ __launch_bounds__(FA_THREADS) void k_autoc(FaEncParams P, FaBatch B, const void *pcm, const FaSubStat *stats,
                                                       const float *windows, float *autoc, uint32_t tasks_per_block)
{
	__shared__ float x[AC_LDS_FLOATS];
	const uint32_t lag = P.max_lpc_order + 1;
	const uint32_t row_len = AC_TILE + lag;
	const uint32_t tid = threadIdx.x;
	const uint32_t ntasks = B.nframes * P.nsignals * P.napod;
	const uint32_t task0 = blockIdx.x * tasks_per_block;
	const uint32_t my_j = tid / lag, my_l = tid % lag;
	const bool active = my_j < tasks_per_block && task0 + my_j < ntasks;
	float acc = 0.0f;

	for (uint32_t c0 = 0; c0 < P.blocksize; c0 += AC_TILE) {
		// stage tile [c0, c0 + AC_TILE + lag) of every task's windowed signal
		for (uint32_t e = tid; e < tasks_per_block * row_len; e += FA_THREADS) {
			const uint32_t j = e / row_len, k = e % row_len;
			const uint32_t task = task0 + j;
			float v = 0.0f;
			if (task < ntasks) {
				const uint32_t sub = task / P.napod, a = task % P.napod;
				const uint32_t f = sub / P.nsignals, s = sub % P.nsignals;
				const uint32_t bsf = frame_len(B, P.blocksize, f);
				const uint32_t idx = c0 + k;
				if (idx < bsf) {
					const int32_t smp = fetch_signal(pcm, P.channels, B.sample_bytes, (uint64_t)f * P.blocksize + idx, s) >> stats[sub].wasted;
					v = __fmul_rn((float)smp, windows[(size_t)a * P.blocksize + idx]);
				}
			}
			x[e] = v;
		}
		__syncthreads();
		if (active) {
			const float *row = x + my_j * row_len;
#pragma unroll 8
			for (uint32_t sidx = 0; sidx < AC_TILE; sidx++)
				acc = __fadd_rn(acc, __fmul_rn(row[sidx], row[sidx + my_l]));
		}
		__syncthreads();
	}
	if (active)
		autoc[(size_t)(task0 + my_j) * (FA_MAX_ORDER + 1) + my_l] = acc;
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
	none.valid = 0; none.order = 0; none.precision = 0; none.shift = 0; none.rice = 0; none.wide = 0; none.pad[0] = none.pad[1] = 0;
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
				}
				FaLpcCand cd;
				cd.valid = 1;
				cd.order = (uint8_t)order;
				cd.precision = (uint8_t)precision;
				cd.shift = (int8_t)(shift < 0 ? 0 : shift);
				cd.rice = (uint8_t)rice;
				cd.wide = (sbps + precision + ilog2u(order) <= 32) ? 0 : 1;  // stream_encoder.c:3592
				cd.pad[0] = cd.pad[1] = 0;
				my_cands[slot] = cd;
			}
		}
	}
}

// ---------------------------------------------------------------------------------------------
// residual arithmetic
// ---------------------------------------------------------------------------------------------
// d[HIST + i] is sample i of the thread's chunk; res[i] for the 16 chunk samples.
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

// FLAC__lpc_compute_residual_from_qlp_coefficients (lpc.c:265, int32 sum that wraps) and _wide
// (lpc.c:531, int64 sum).  MODE 0: 24x24-bit multiplies (operands known to fit), wrapping int32
// sum.  MODE 1: int64 sum truncated to int32 before the shift (== the wrapping int32 sum).
// MODE 2: int64 sum, shifted, then truncated.
template <int HIST, int ORDER, int MODE>
__device__ __forceinline__ void lpc_residual16(const int32_t (&d)[HIST + FA_CHUNK], const int32_t *__restrict__ q, int shift, int32_t (&res)[FA_CHUNK])
{
	int32_t qq[ORDER];
#pragma unroll
	for (int j = 0; j < ORDER; j++)
		qq[j] = q[j];
#pragma unroll
	for (int i = 0; i < FA_CHUNK; i++) {
		int32_t pred;
		if (MODE == 0) {
			int32_t sum = 0;
#pragma unroll
			for (int j = 0; j < ORDER; j++)
				sum += __mul24(qq[j], d[HIST + i - 1 - j]);
			pred = sum >> shift;
		} else {
			int64_t sum = 0;
#pragma unroll
			for (int j = 0; j < ORDER; j++)
				sum += (int64_t)qq[j] * (int64_t)d[HIST + i - 1 - j];
			pred = (MODE == 1) ? ((int32_t)sum >> shift) : (int32_t)(sum >> shift);
		}
		res[i] = (int32_t)((uint32_t)d[HIST + i] - (uint32_t)pred);
	}
}

template <int HIST, int ORDER>
__device__ __forceinline__ void lpc_residual16_mode(const int32_t (&d)[HIST + FA_CHUNK], const int32_t *q, int shift, int mode, int32_t (&res)[FA_CHUNK])
{
	if (mode == 0)
		lpc_residual16<HIST, ORDER, 0>(d, q, shift, res);
	else if (mode == 1)
		lpc_residual16<HIST, ORDER, 1>(d, q, shift, res);
	else
		lpc_residual16<HIST, ORDER, 2>(d, q, shift, res);
}

#define LPC_CASE(N) \
	case N: \
		if (N <= HIST) lpc_residual16_mode<HIST, (N <= HIST ? N : 1)>(d, q, shift, mode, res); \
		break;

template <int HIST>
__device__ __forceinline__ void lpc_residual16_any(const int32_t (&d)[HIST + FA_CHUNK], const int32_t *q, uint32_t order, int shift, int mode,
                                                   int32_t (&res)[FA_CHUNK])
{
	switch (order) {
		LPC_CASE(1) LPC_CASE(2) LPC_CASE(3) LPC_CASE(4) LPC_CASE(5) LPC_CASE(6) LPC_CASE(7) LPC_CASE(8)
		LPC_CASE(9) LPC_CASE(10) LPC_CASE(11) LPC_CASE(12) LPC_CASE(13) LPC_CASE(14) LPC_CASE(15) LPC_CASE(16)
		LPC_CASE(17) LPC_CASE(18) LPC_CASE(19) LPC_CASE(20) LPC_CASE(21) LPC_CASE(22) LPC_CASE(23) LPC_CASE(24)
		LPC_CASE(25) LPC_CASE(26) LPC_CASE(27) LPC_CASE(28) LPC_CASE(29) LPC_CASE(30) LPC_CASE(31) LPC_CASE(32)
	default: break;
	}
}

// which arithmetic the residual of an LPC candidate needs
__device__ __forceinline__ int lpc_mode(uint32_t wide, uint32_t sbps) { return wide ? 2 : (sbps <= 24 ? 0 : 1); }

// ---------------------------------------------------------------------------------------------
// k_search: process_subframe_ (stream_encoder.c:3164) for one (frame, signal): every candidate's
// residual is produced in registers, only |residual| partition sums go to LDS, and
// find_best_partition_order_ / set_partitioned_rice_ (stream_encoder.c:3666,3943) run on those.
// ---------------------------------------------------------------------------------------------
#define SR_SUM_BYTES 32768
#define SR_MAXGROUP 32

struct SrCand {
	uint32_t kind;      // FA_SUB_FIXED / FA_SUB_LPC
	uint32_t order;
	uint32_t seed;      // Rice parameter used at partition order 0
	uint32_t slot;      // LPC candidate slot
	uint32_t precision;
	int32_t  shift;
	uint32_t mode;
	uint32_t pmax, pmin, psize, use32;
	uint32_t hdr_bits;  // subframe bits besides the residual
	uint32_t best_po, best_rbits, total_bits;
};

template <int HIST>
__global__ __launch_bounds__(FA_THREADS) void k_search(FaEncParams P, FaBatch B, const void *pcm, const FaSubStat *stats,
                                                       const FaLpcCand *cands, const int32_t *coefs, FaPlan *plans)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	uint64_t *sums = (uint64_t *)smem;                                  // [group][2 * PF]
	uint32_t *pbits = (uint32_t *)(smem + SR_SUM_BYTES);                // [group][16]
	SrCand *gc = (SrCand *)(pbits + SR_MAXGROUP * 16);                  // [group]
	FaPlan *best = (FaPlan *)(gc + SR_MAXGROUP);
	uint32_t *ctl = (uint32_t *)(best + 1);                             // [0]=ngroup [1]=cursor [2]=best_changed_to [3]=rice2 flag

	const uint32_t task = blockIdx.x;
	const uint32_t f = task / P.nsignals, s = task % P.nsignals;
	const uint32_t tid = threadIdx.x;
	const uint32_t bsf = frame_len(B, P.blocksize, f);
	const uint64_t frame_start = (uint64_t)f * P.blocksize;
	const FaSubStat st = stats[task];
	const uint32_t sbps = st.bps, wasted = st.wasted;
	const uint32_t hdr = 8 + wasted;
	const bool fractional = bsf != P.blocksize;
	// process_subframes_ (stream_encoder.c:2929-2936)
	uint32_t frame_max_po = 0;
	if (!fractional) {
		uint32_t o = 0, b = bsf;
		while (!(b & 1)) { o++; b >>= 1; }
		frame_max_po = umin32(umin32(o, 15u), P.max_po);
	}
	const uint32_t frame_min_po = umin32(P.min_po, frame_max_po);
	const uint32_t PF = 1u << frame_max_po;
	const uint32_t group_cap = umin32(SR_MAXGROUP, SR_SUM_BYTES / (2 * PF * 8));
	const FaLpcCand *my_cands = cands + (size_t)task * P.ncand;
	const int32_t *my_coefs = coefs + (size_t)task * P.ncand * P.max_lpc_order;

	if (tid == 0) {
		best->type = FA_SUB_VERBATIM;
		best->order = 0;
		best->precision = 0;
		best->shift = 0;
		best->part_order = 0;
		best->rice2 = 0;
		best->wide = 0;
		best->pad = 0;
		best->constant = 0;
		// evaluate_verbatim_subframe_ (stream_encoder.c:3641)
		best->bits = (P.disable_verbatim && bsf >= 4) ? 0xffffffffu : hdr + bsf * sbps;
		if (bsf >= 4 && st.is_constant) {
			// evaluate_constant_subframe_ (stream_encoder.c:3470)
			const uint32_t cb = hdr + sbps;
			if (cb < best->bits) {
				best->type = FA_SUB_CONSTANT;
				best->bits = cb;
				best->constant = st.first_sample;
			}
		}
		ctl[1] = 0;
	}
	__syncthreads();

	const bool search = bsf >= 4 && !st.is_constant;
	const uint32_t total_cands = 5 + P.ncand;
	while (search) {
		// ---- thread 0 gathers the next group of candidates, in the reference's trial order ----
		if (tid == 0) {
			uint32_t n = 0, cur = ctl[1];
			while (cur < total_cands && n < group_cap) {
				SrCand c;
				bool take = false;
				if (cur < 5) {
					// fixed predictors (stream_encoder.c:3236-3287)
					const uint32_t order = cur;
					uint32_t lo, hi;
					if (P.exhaustive) { lo = 0; hi = 4; }
					else lo = hi = st.fixed_guess;
					if (hi >= bsf) hi = bsf - 1;
					const bool enabled = !P.disable_fixed || (P.max_lpc_order == 0 && best->bits == 0xffffffffu);
					if (enabled && order >= lo && order <= hi && !(st.rbps[order] >= (float)sbps)) {
						uint32_t rice = (st.rbps[order] > 0.0f) ? (uint32_t)((double)st.rbps[order] + 0.5) : 0u;
						rice++;
						if (rice >= P.rice_limit) rice = P.rice_limit - 1;
						c.kind = FA_SUB_FIXED;
						c.order = order;
						c.seed = rice;
						c.slot = 0;
						c.precision = 0;
						c.shift = 0;
						c.mode = 0;
						c.hdr_bits = hdr + order * sbps;
						take = true;
					}
				} else {
					const FaLpcCand lc = my_cands[cur - 5];
					if (lc.valid) {
						c.kind = FA_SUB_LPC;
						c.order = lc.order;
						c.seed = lc.rice;
						c.slot = cur - 5;
						c.precision = lc.precision;
						c.shift = lc.shift;
						c.mode = (uint32_t)lpc_mode(lc.wide, sbps);
						c.hdr_bits = hdr + 4 + 5 + lc.order * (lc.precision + sbps);
						take = true;
					}
				}
				if (take) {
					c.pmax = limit_part_order(frame_max_po, bsf, c.order);
					c.pmin = umin32(frame_min_po, c.pmax);
					c.psize = bsf >> c.pmax;
					c.use32 = (ilog2u(c.psize) + sbps < 32) ? 1u : 0u;  // stream_encoder.c:3803
					c.best_po = 0;
					c.best_rbits = 0;
					c.total_bits = 0;
					gc[n++] = c;
				}
				cur++;
			}
			ctl[0] = n;
			ctl[1] = cur;
		}
		__syncthreads();
		const uint32_t ng = ctl[0];
		if (ng == 0)
			break;
		for (uint32_t e = tid; e < ng * 2 * PF; e += FA_THREADS)
			sums[e] = 0;
		for (uint32_t e = tid; e < ng * 16; e += FA_THREADS)
			pbits[e] = 0;
		__syncthreads();

		// ---- residuals of every candidate of the group, tile by tile, all in registers ----
		for (uint32_t tile = 0; tile * FA_TILE < bsf; tile++) {
			const int32_t base = (int32_t)(tile * FA_TILE + tid * FA_CHUNK);
			if ((uint32_t)base >= bsf)
				continue;
			int32_t d[HIST + FA_CHUNK];
			load_chunk<HIST>(d, pcm, P.channels, B.sample_bytes, frame_start, bsf, s, wasted, base);
			for (uint32_t g = 0; g < ng; g++) {
				const uint32_t kind = __builtin_amdgcn_readfirstlane(gc[g].kind);
				const uint32_t order = __builtin_amdgcn_readfirstlane(gc[g].order);
				const uint32_t psize = __builtin_amdgcn_readfirstlane(gc[g].psize);
				int32_t res[FA_CHUNK];
				if (kind == FA_SUB_FIXED)
					fixed_residual16<HIST>(d, order, res);
				else {
					const uint32_t slot = __builtin_amdgcn_readfirstlane(gc[g].slot);
					const int shift = (int)__builtin_amdgcn_readfirstlane((uint32_t)gc[g].shift);
					const int mode = (int)__builtin_amdgcn_readfirstlane(gc[g].mode);
					lpc_residual16_any<HIST>(d, my_coefs + (size_t)slot * P.max_lpc_order, order, shift, mode, res);
				}
				uint64_t *gs = sums + (size_t)g * 2 * PF;
				if ((psize & (FA_CHUNK - 1)) == 0) {
					// the whole chunk lies in one partition
					uint64_t t = 0;
#pragma unroll
					for (int i = 0; i < FA_CHUNK; i++) {
						const uint32_t idx = (uint32_t)base + i;
						if (idx >= order && idx < bsf)
							t += abs32(res[i]);
					}
					atomicAdd((unsigned long long *)&gs[(uint32_t)base / psize], (unsigned long long)t);
				} else {
					uint32_t part = (uint32_t)base / psize;
					uint32_t next = (part + 1) * psize;
					uint64_t t = 0;
#pragma unroll
					for (int i = 0; i < FA_CHUNK; i++) {
						const uint32_t idx = (uint32_t)base + i;
						if (idx == next) {
							if (t) atomicAdd((unsigned long long *)&gs[part], (unsigned long long)t);
							t = 0;
							part++;
							next += psize;
						}
						if (idx >= order && idx < bsf)
							t += abs32(res[i]);
					}
					if (t) atomicAdd((unsigned long long *)&gs[part], (unsigned long long)t);
				}
			}
		}
		__syncthreads();

		// ---- precompute_partition_info_sums_ (stream_encoder.c:3774): truncate, then merge ----
		for (uint32_t e = tid; e < ng * PF; e += FA_THREADS) {
			const uint32_t g = e / PF, part = e % PF;
			if (part < (1u << gc[g].pmax) && gc[g].use32)
				sums[(size_t)g * 2 * PF + part] &= 0xffffffffull;
		}
		__syncthreads();
		for (int lvl = (int)frame_max_po - 1; lvl >= 0; lvl--) {
			for (uint32_t e = tid; e < ng * (1u << lvl); e += FA_THREADS) {
				const uint32_t g = e >> lvl, part = e & ((1u << lvl) - 1);
				const uint32_t pm = gc[g].pmax;
				if ((uint32_t)lvl < pm && (uint32_t)lvl >= gc[g].pmin) {
					uint64_t *gs = sums + (size_t)g * 2 * PF;
					const uint32_t off_hi = (2u << pm) - (2u << (lvl + 1));
					const uint32_t off_lo = (2u << pm) - (2u << lvl);
					gs[off_lo + part] = gs[off_hi + 2 * part] + gs[off_hi + 2 * part + 1];
				}
			}
			__syncthreads();
		}
		// ---- set_partitioned_rice_ (stream_encoder.c:3943) for every (candidate, order, partition) ----
		for (uint32_t e = tid; e < ng * 2 * PF; e += FA_THREADS) {
			const uint32_t g = e / (2 * PF), idx = e % (2 * PF);
			const SrCand &c = gc[g];
			const uint32_t span = 2u << c.pmax;
			if (idx + 1 >= span)
				continue;
			const uint32_t r = span - idx;
			const uint32_t po = ilog2u(r - 1);
			if (po < c.pmin)
				continue;
			const uint32_t part = (2u << po) - r;
			uint32_t n = bsf >> po;
			if (part == 0)
				n -= c.order;
			const uint64_t sm = sums[(size_t)g * 2 * PF + idx];
			const uint32_t k = (po == 0) ? c.seed : rice_k_from_sum(sm, n, P.rice_limit);
			atomicAdd(&pbits[g * 16 + po], rice_bits_estimate(k, n, sm));
		}
		__syncthreads();
		// ---- find_best_partition_order_ (stream_encoder.c:3700-3729): scan max -> min, strict < ----
		if (tid < ng) {
			SrCand &c = gc[tid];
			uint32_t bestb = 0, bestpo = 0;
			for (int po = (int)c.pmax; po >= (int)c.pmin; po--) {
				const uint32_t bits = 2 + 4 + pbits[tid * 16 + po];
				if (bestb == 0 || bits < bestb) {
					bestb = bits;
					bestpo = (uint32_t)po;
				}
			}
			c.best_po = bestpo;
			c.best_rbits = bestb;
			c.total_bits = c.hdr_bits + bestb;
		}
		__syncthreads();
		if (tid == 0) {
			uint32_t winner = 0xffffffffu;
			uint32_t bb = best->bits;
			for (uint32_t g = 0; g < ng; g++)
				if (gc[g].total_bits < bb) {
					bb = gc[g].total_bits;
					winner = g;
				}
			ctl[2] = winner;
			ctl[3] = 0;
		}
		__syncthreads();
		const uint32_t winner = ctl[2];
		if (winner != 0xffffffffu) {
			// keep the winner's parameters: its sums are overwritten by the next group
			const SrCand &c = gc[winner];
			const uint32_t po = c.best_po;
			const uint32_t off = (2u << c.pmax) - (2u << po);
			if (tid < (1u << po)) {
				uint32_t n = bsf >> po;
				if (tid == 0)
					n -= c.order;
				const uint64_t sm = sums[(size_t)winner * 2 * PF + off + tid];
				const uint32_t k = (po == 0) ? c.seed : rice_k_from_sum(sm, n, P.rice_limit);
				best->params[tid] = (uint8_t)k;
				if (k >= 15)
					atomicOr(&ctl[3], 1u);  // stream_encoder.c:3752-3757
			}
			if (tid < c.order && c.kind == FA_SUB_LPC)
				best->q[tid] = my_coefs[(size_t)c.slot * P.max_lpc_order + tid];
			__syncthreads();
			if (tid == 0) {
				best->bits = c.total_bits;
				best->type = (uint8_t)c.kind;
				best->order = (uint8_t)c.order;
				best->precision = (uint8_t)c.precision;
				best->shift = (int8_t)c.shift;
				best->part_order = (uint8_t)po;
				best->rice2 = (uint8_t)ctl[3];
				best->wide = (uint8_t)c.mode;
			}
		}
		__syncthreads();
	}
	if (tid == 0 && best->bits == 0xffffffffu) {
		// stream_encoder.c:3388-3391
		best->type = FA_SUB_VERBATIM;
		best->bits = hdr + bsf * sbps;
	}
	__syncthreads();
	// write the plan out
	{
		const uint32_t *src = (const uint32_t *)best;
		uint32_t *dst = (uint32_t *)(plans + task);
		for (uint32_t e = tid; e < sizeof(FaPlan) / 4; e += FA_THREADS)
			dst[e] = src[e];
	}
}

size_t fa_search_lds_bytes(void)
{
	return SR_SUM_BYTES + SR_MAXGROUP * 16 * sizeof(uint32_t) + SR_MAXGROUP * sizeof(SrCand) + sizeof(FaPlan) + 16 * sizeof(uint32_t);
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

__global__ void k_decide(FaEncParams P, FaBatch B, const FaPlan *plans, uint32_t *assign, uint32_t *loose_out)
{
	const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
	if (f >= B.nframes)
		return;
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
// k_pack
// ---------------------------------------------------------------------------------------------
#define PK_CAPW 8192  // window of the frame's bitstream held in LDS, in 32-bit words

__constant__ uint16_t c_crc16_table[256];
__constant__ uint16_t c_crc16_pow[32];  // x^(8 * 2^j) mod the CRC-16 polynomial
__constant__ uint8_t c_crc8_table[256];

void fa_upload_crc_tables(const uint16_t *t16, const uint16_t *pow16, const uint8_t *t8)
{
	(void)hipMemcpyToSymbol(HIP_SYMBOL(c_crc16_table), t16, 512);
	(void)hipMemcpyToSymbol(HIP_SYMBOL(c_crc16_pow), pow16, 64);
	(void)hipMemcpyToSymbol(HIP_SYMBOL(c_crc8_table), t8, 256);
}

// product of two polynomials modulo x^16 + x^15 + x^2 + 1
__device__ __forceinline__ uint32_t gf16_mul(uint32_t a, uint32_t b)
{
	uint32_t r = 0;
#pragma unroll
	for (int i = 0; i < 16; i++) {
		if (b & 1)
			r ^= a;
		b >>= 1;
		a <<= 1;
		if (a & 0x10000)
			a ^= 0x18005;
	}
	return r;
}

// x^(8 n) modulo the polynomial
__device__ __forceinline__ uint32_t gf16_xpow8(uint32_t n)
{
	uint32_t r = 1;
	for (int j = 0; n; j++, n >>= 1)
		if (n & 1)
			r = gf16_mul(r, c_crc16_pow[j]);
	return r;
}

struct PkShared {
	uint32_t W[PK_CAPW];
	uint32_t scan[4];
	uint32_t crcs[FA_THREADS];
	uint16_t crc_table[256];
	uint32_t base_word;
	uint32_t frame_crc;
	uint32_t error;
	uint8_t  params[FA_MAX_PARTS];
	int32_t  q[FA_MAX_ORDER];
};

// OR the low nbits of value, MSB first, at bit position bitpos of the frame; only words inside
// the LDS window are touched (zero bits never need writing: the window is pre-cleared).
__device__ __forceinline__ void emit_bits(PkShared &S, uint32_t bitpos, uint32_t nbits, uint32_t value)
{
	if (nbits == 0)
		return;
	if (nbits < 32)
		value &= (1u << nbits) - 1u;
	const uint32_t w0 = bitpos >> 5, sh = bitpos & 31;
	const uint64_t v = ((uint64_t)value << (64 - nbits)) >> sh;
	const uint32_t hi = (uint32_t)(v >> 32), lo = (uint32_t)v;
	const uint32_t r0 = w0 - S.base_word;  // wraps to a huge value for words already flushed
	if (hi && r0 < PK_CAPW)
		atomicOr(&S.W[r0], hi);
	if (lo && r0 + 1 < PK_CAPW)
		atomicOr(&S.W[r0 + 1], lo);
}

// Block-wide exclusive prefix sum of one uint32 per thread; returns the prefix, *total gets the sum.
__device__ __forceinline__ uint32_t block_exclusive_scan(PkShared &S, uint32_t v, uint32_t *total)
{
	const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	uint32_t inc = v;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const uint32_t t = __shfl_up(inc, o, 64);
		if (lane >= (uint32_t)o)
			inc += t;
	}
	__syncthreads();
	if (lane == 63)
		S.scan[wv] = inc;
	__syncthreads();
	uint32_t off = 0;
	for (uint32_t w = 0; w < wv; w++)
		off += S.scan[w];
	*total = S.scan[0] + S.scan[1] + S.scan[2] + S.scan[3];
	return off + inc - v;
}

// CRC-16 of nbytes bytes that start at window word 0 (FLAC__crc16, crc.c:134): each thread takes an
// equal run of bytes, the runs are right-aligned (leading virtual zeros do not change a CRC that
// starts at 0), and a tree folds them with multiplications by powers of x.
__device__ __forceinline__ uint32_t window_crc16(PkShared &S, uint32_t nbytes)
{
	const uint32_t tid = threadIdx.x;
	const uint32_t R = (nbytes + FA_THREADS - 1) / FA_THREADS;
	uint32_t crc = 0;
	if (R) {
		const int64_t start = (int64_t)nbytes - (int64_t)(FA_THREADS - tid) * R;
		for (uint32_t i = 0; i < R; i++) {
			const int64_t b = start + i;
			if (b >= 0) {
				const uint32_t byte = (S.W[b >> 2] >> (24 - 8 * (b & 3))) & 0xff;
				crc = ((crc << 8) & 0xffff) ^ S.crc_table[(crc >> 8) ^ byte];
			}
		}
	}
	S.crcs[tid] = crc;
	__syncthreads();
	uint32_t m = gf16_xpow8(R);
	for (uint32_t L = 0; L < 8; L++) {
		const uint32_t mask = (2u << L) - 1;
		if ((tid & mask) == mask)
			S.crcs[tid] = gf16_mul(S.crcs[tid - (1u << L)], m) ^ S.crcs[tid];
		m = gf16_mul(m, m);
		__syncthreads();
	}
	const uint32_t out = S.crcs[FA_THREADS - 1];
	__syncthreads();
	return out;
}

// Write window words [0, nwords) to the frame slot (big-endian bit order -> byte order), fold
// their CRC into the frame CRC, slide the window.  keep_partial: word nwords holds the
// unfinished tail and becomes word 0.
__device__ __forceinline__ void flush_window(PkShared &S, uint32_t *slot, uint32_t slot_words, uint32_t nwords, bool keep_partial, bool with_crc)
{
	const uint32_t tid = threadIdx.x;
	if (with_crc && nwords) {
		const uint32_t c = window_crc16(S, nwords * 4);
		if (tid == 0)
			S.frame_crc = gf16_mul(S.frame_crc, gf16_xpow8(nwords * 4)) ^ c;
	}
	const uint32_t bw = S.base_word;
	for (uint32_t w = tid; w < nwords; w += FA_THREADS) {
		if (bw + w < slot_words)
			slot[bw + w] = __builtin_bswap32(S.W[w]);
		else
			S.error = 1;
	}
	__syncthreads();
	const uint32_t carry = keep_partial ? S.W[nwords] : 0u;
	__syncthreads();
	for (uint32_t w = tid; w < PK_CAPW; w += FA_THREADS)
		S.W[w] = 0;
	__syncthreads();
	if (tid == 0) {
		S.W[0] = carry;
		S.base_word = bw + nwords;
	}
	__syncthreads();
}

// Emit everything the threads hold for bit range [*, end_bit) of the current tile, as many
// window passes as it takes.  `emit` is called once per pass and must be idempotent.
template <typename F>
__device__ __forceinline__ void drain(PkShared &S, uint32_t *slot, uint32_t slot_words, uint32_t end_bit, F emit)
{
	for (;;) {
		emit();
		__syncthreads();
		const uint32_t end_word = end_bit >> 5;  // first word that is not complete yet
		const uint32_t bw = S.base_word;
		if (end_word <= bw + PK_CAPW - 1) {
			// everything fits: flush the complete words, keep the partial one
			flush_window(S, slot, slot_words, end_word - bw, true, true);
			return;
		}
		flush_window(S, slot, slot_words, PK_CAPW, false, true);
	}
}

template <int HIST>
__global__ __launch_bounds__(FA_THREADS) void k_pack(FaEncParams P, FaBatch B, const void *pcm, const FaSubStat *stats, const FaPlan *plans,
                                                     const uint32_t *assign, uint32_t *slots, uint32_t *frame_bytes, uint32_t *errflag)
{
	__shared__ PkShared S;
	const uint32_t f = blockIdx.x, tid = threadIdx.x;
	const uint32_t bsf = frame_len(B, P.blocksize, f);
	const uint64_t frame_start = (uint64_t)f * P.blocksize;
	const uint32_t slot_words = P.slot_bytes / 4;
	uint32_t *slot = slots + (size_t)f * slot_words;
	const uint32_t ca = assign[f];

	for (uint32_t w = tid; w < PK_CAPW; w += FA_THREADS)
		S.W[w] = 0;
	S.crc_table[tid] = c_crc16_table[tid];
	if (tid == 0) {
		S.base_word = 0;
		S.frame_crc = 0;
		S.error = 0;
	}
	__syncthreads();

	// ---- frame header: FLAC__frame_add_header (stream_encoder_framing.c:219) ----
	uint32_t bitpos;
	{
		uint8_t h[16];
		uint32_t n = 0;
		h[n++] = 0xff;
		h[n++] = 0xf8;
		uint32_t u, bs_hint = 0, sr_hint = 0;
		switch (bsf) {
		case 192: u = 1; break;
		case 576: u = 2; break;
		case 1152: u = 3; break;
		case 2304: u = 4; break;
		case 4608: u = 5; break;
		case 256: u = 8; break;
		case 512: u = 9; break;
		case 1024: u = 10; break;
		case 2048: u = 11; break;
		case 4096: u = 12; break;
		case 8192: u = 13; break;
		case 16384: u = 14; break;
		case 32768: u = 15; break;
		default: bs_hint = u = bsf <= 0x100 ? 6 : 7; break;
		}
		uint32_t v;
		switch (P.sample_rate) {
		case 88200: v = 1; break;
		case 176400: v = 2; break;
		case 192000: v = 3; break;
		case 8000: v = 4; break;
		case 16000: v = 5; break;
		case 22050: v = 6; break;
		case 24000: v = 7; break;
		case 32000: v = 8; break;
		case 44100: v = 9; break;
		case 48000: v = 10; break;
		case 96000: v = 11; break;
		default:
			if (P.sample_rate <= 255000 && P.sample_rate % 1000 == 0) sr_hint = v = 12;
			else if (P.sample_rate % 10 == 0) sr_hint = v = 14;
			else if (P.sample_rate <= 0xffff) sr_hint = v = 13;
			else v = 0;
			break;
		}
		h[n++] = (uint8_t)((u << 4) | v);
		const uint32_t chcode = ca == 0 ? P.channels - 1 : 7 + ca;
		uint32_t bcode;
		switch (P.bps) {
		case 8: bcode = 1; break;
		case 12: bcode = 2; break;
		case 16: bcode = 4; break;
		case 20: bcode = 5; break;
		case 24: bcode = 6; break;
		default: bcode = 0; break;
		}
		h[n++] = (uint8_t)((chcode << 4) | (bcode << 1));
		// FLAC__bitwriter_write_utf8_uint32 (bitwriter.c:784)
		const uint32_t val = B.first_frame + f;
		if (val < 0x80)
			h[n++] = (uint8_t)val;
		else {
			const int cont = val < 0x800 ? 1 : val < 0x10000 ? 2 : val < 0x200000 ? 3 : val < 0x4000000 ? 4 : 5;
			const uint32_t lead = (0xff00u >> (cont + 1)) & 0xff;  // 0xC0, 0xE0, 0xF0, 0xF8, 0xFC
			h[n++] = (uint8_t)(lead | (val >> (6 * cont)));
			for (int i = cont - 1; i >= 0; i--)
				h[n++] = (uint8_t)(0x80 | ((val >> (6 * i)) & 0x3f));
		}
		if (bs_hint == 6)
			h[n++] = (uint8_t)(bsf - 1);
		else if (bs_hint == 7) {
			h[n++] = (uint8_t)((bsf - 1) >> 8);
			h[n++] = (uint8_t)(bsf - 1);
		}
		if (sr_hint == 12)
			h[n++] = (uint8_t)(P.sample_rate / 1000);
		else if (sr_hint == 13) {
			h[n++] = (uint8_t)(P.sample_rate >> 8);
			h[n++] = (uint8_t)P.sample_rate;
		} else if (sr_hint == 14) {
			h[n++] = (uint8_t)((P.sample_rate / 10) >> 8);
			h[n++] = (uint8_t)(P.sample_rate / 10);
		}
		uint8_t crc = 0;  // FLAC__crc8 (crc.c:124)
		for (uint32_t i = 0; i < n; i++)
			crc = c_crc8_table[crc ^ h[i]];
		h[n++] = crc;
		if (tid == 0)
			for (uint32_t i = 0; i < n; i++)
				emit_bits(S, 8 * i, 8, h[i]);
		bitpos = 8 * n;
	}
	__syncthreads();

	// ---- subframes in written order (stream_encoder.c:3091-3151) ----
	for (uint32_t c = 0; c < P.channels; c++) {
		uint32_t sig = c;
		if (ca == 1 && c == 1) sig = P.channels + 1;       // left, side
		else if (ca == 2 && c == 0) sig = P.channels + 1;  // side, right
		else if (ca == 3) sig = P.channels + c;            // mid, side
		const FaPlan *pl = plans + (size_t)f * P.nsignals + sig;
		const FaSubStat st = stats[(size_t)f * P.nsignals + sig];
		const uint32_t type = pl->type, order = pl->order, sbps = st.bps, wasted = st.wasted;
		const uint32_t precision = pl->precision, po = pl->part_order, rice2 = pl->rice2, mode = pl->wide;
		const int shift = pl->shift;
		S.params[tid] = pl->params[tid];
		if (tid < FA_MAX_ORDER)
			S.q[tid] = pl->q[tid];
		__syncthreads();

		// FLAC__subframe_add_* (stream_encoder_framing.c:366-479): type byte, wasted-bits unary
		uint32_t typebits;
		switch (type) {
		case FA_SUB_CONSTANT: typebits = 0x00; break;
		case FA_SUB_VERBATIM: typebits = 0x02; break;
		case FA_SUB_FIXED: typebits = 0x10 | (order << 1); break;
		default: typebits = 0x40 | ((order - 1) << 1); break;
		}
		uint32_t pos = bitpos;
		if (tid == 0) {
			emit_bits(S, pos, 8, typebits | (wasted ? 1u : 0u));
			if (wasted)
				emit_bits(S, pos + 8 + wasted - 1, 1, 1);
		}
		pos += 8 + wasted;
		if (type == FA_SUB_CONSTANT) {
			if (tid == 0)
				emit_bits(S, pos, sbps, (uint32_t)pl->constant);
			pos += sbps;
			__syncthreads();
			bitpos = pos;
			continue;
		}
		uint32_t plen = 0, psize = bsf, first_code_sample = 0, code_kind = 0;  // code_kind 0: verbatim fields, 1: Rice codes
		if (type != FA_SUB_VERBATIM) {
			// warm-up samples, LPC parameters, entropy coding method header
			if (tid < order) {
				const int32_t smp = fetch_signal(pcm, P.channels, B.sample_bytes, frame_start + tid, sig) >> wasted;
				emit_bits(S, pos + tid * sbps, sbps, (uint32_t)smp);
			}
			pos += order * sbps;
			if (type == FA_SUB_LPC) {
				if (tid == 0) {
					emit_bits(S, pos, 4, precision - 1);
					emit_bits(S, pos + 4, 5, (uint32_t)shift);
				}
				pos += 9;
				if (tid < order)
					emit_bits(S, pos + tid * precision, precision, (uint32_t)S.q[tid]);
				pos += order * precision;
			}
			if (tid == 0) {
				emit_bits(S, pos, 2, rice2 ? 1u : 0u);
				emit_bits(S, pos + 2, 4, po);
			}
			pos += 6;
			plen = rice2 ? 5 : 4;
			psize = bsf >> po;
			first_code_sample = order;
			code_kind = 1;
		}
		__syncthreads();

		// ---- per-sample codes, tile by tile ----
		uint32_t tile_start_bit = pos;
		for (uint32_t tile = 0; tile * FA_TILE < bsf; tile++) {
			const int32_t base = (int32_t)(tile * FA_TILE + tid * FA_CHUNK);
			int32_t res[FA_CHUNK];
			uint32_t klist[FA_CHUNK];
			uint32_t nbits = 0;
			const bool have = (uint32_t)base < bsf;
			if (have) {
				int32_t d[HIST + FA_CHUNK];
				load_chunk<HIST>(d, pcm, P.channels, B.sample_bytes, frame_start, bsf, sig, wasted, base);
				if (type == FA_SUB_VERBATIM) {
#pragma unroll
					for (int i = 0; i < FA_CHUNK; i++)
						res[i] = d[HIST + i];
				} else if (type == FA_SUB_FIXED)
					fixed_residual16<HIST>(d, order, res);
				else
					lpc_residual16_any<HIST>(d, S.q, order, shift, (int)mode, res);
				uint32_t part = (uint32_t)base / psize;
				uint32_t next = (part + 1) * psize;
#pragma unroll
				for (int i = 0; i < FA_CHUNK; i++) {
					const uint32_t idx = (uint32_t)base + i;
					if (idx == next) {
						part++;
						next += psize;
					}
					uint32_t k = 0;
					if (idx >= first_code_sample && idx < bsf) {
						if (code_kind == 0)
							nbits += sbps;
						else {
							k = S.params[part];
							const uint32_t uz = ((uint32_t)res[i] << 1) ^ (uint32_t)(res[i] >> 31);
							nbits += (uz >> k) + 1 + k;
							if (idx == first_code_sample || idx == part * psize)
								nbits += plen;
						}
					}
					klist[i] = k;
				}
			}
			uint32_t tile_bits;
			const uint32_t my_start = tile_start_bit + block_exclusive_scan(S, nbits, &tile_bits);
			const uint32_t tile_end_bit = tile_start_bit + tile_bits;
			drain(S, slot, slot_words, tile_end_bit, [&]() {
				if (!have)
					return;
				uint32_t p = my_start;
				uint32_t part = (uint32_t)base / psize;
				uint32_t next = (part + 1) * psize;
#pragma unroll
				for (int i = 0; i < FA_CHUNK; i++) {
					const uint32_t idx = (uint32_t)base + i;
					if (idx == next) {
						part++;
						next += psize;
					}
					if (idx >= first_code_sample && idx < bsf) {
						if (code_kind == 0) {
							emit_bits(S, p, sbps, (uint32_t)res[i]);
							p += sbps;
						} else {
							const uint32_t k = klist[i];
							if (idx == first_code_sample || idx == part * psize) {
								emit_bits(S, p, plen, k);
								p += plen;
							}
							// FLAC__bitwriter_write_rice_signed_block (bitwriter.c:544)
							const uint32_t uz = ((uint32_t)res[i] << 1) ^ (uint32_t)(res[i] >> 31);
							const uint32_t msbs = uz >> k;
							emit_bits(S, p + msbs, k + 1, (1u << k) | (uz & ((1u << k) - 1u)));
							p += msbs + 1 + k;
						}
					}
				}
			});
			tile_start_bit = tile_end_bit;
		}
		bitpos = tile_start_bit;
		__syncthreads();
	}

	// ---- zero padding, CRC-16 over everything so far (stream_encoder.c:2885-2900), final flush ----
	const uint32_t nbytes = (bitpos + 7) >> 3;
	{
		const uint32_t bw = S.base_word;
		const uint32_t tail_bytes = nbytes - bw * 4;
		const uint32_t c = window_crc16(S, tail_bytes);
		if (tid == 0) {
			const uint32_t crc = gf16_mul(S.frame_crc, gf16_xpow8(tail_bytes)) ^ c;
			emit_bits(S, nbytes * 8, 16, crc);
		}
		__syncthreads();
		const uint32_t total_words = (nbytes + 2 + 3) / 4;
		flush_window(S, slot, slot_words, total_words - bw, false, false);
	}
	if (tid == 0) {
		frame_bytes[f] = nbytes + 2;
		if (S.error)
			atomicOr(errflag, 1u);
	}
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
static inline int hist_class(const FaEncParams &P)
{
	if (P.max_lpc_order == 0)
		return 4;
	return P.max_lpc_order <= 12 ? 12 : 32;
}

void fa_launch_analyze(const FaEncParams &P, const FaBatch &B, const void *pcm, FaSubStat *stats, hipStream_t st)
{
	hipLaunchKernelGGL(k_analyze, dim3(B.nframes * P.nsignals), dim3(FA_THREADS), 0, st, P, B, pcm, stats);
}

void fa_launch_autoc(const FaEncParams &P, const FaBatch &B, const void *pcm, const FaSubStat *stats, const float *windows, float *autoc,
                     hipStream_t st)
{
	const uint32_t lag = P.max_lpc_order + 1;
	uint32_t tpb = FA_THREADS / lag;
	if (tpb > AC_MAXTASKS)
		tpb = AC_MAXTASKS;
	const uint32_t ntasks = B.nframes * P.nsignals * P.napod;
	hipLaunchKernelGGL(k_autoc, dim3((ntasks + tpb - 1) / tpb), dim3(FA_THREADS), 0, st, P, B, pcm, stats, windows, autoc, tpb);
}

void fa_launch_lpc(const FaEncParams &P, const FaBatch &B, const FaSubStat *stats, const float *autoc, FaLpcCand *cands, int32_t *coefs,
                   hipStream_t st)
{
	const uint32_t nsub = B.nframes * P.nsignals;
	hipLaunchKernelGGL(k_lpc, dim3((nsub + LPC_LANES - 1) / LPC_LANES), dim3(LPC_LANES), 0, st, P, B, stats, autoc, cands, coefs);
}

void fa_launch_search(const FaEncParams &P, const FaBatch &B, const void *pcm, const FaSubStat *stats, const FaLpcCand *cands,
                      const int32_t *coefs, FaPlan *plans, hipStream_t st)
{
	const dim3 grid(B.nframes * P.nsignals), block(FA_THREADS);
	const size_t lds = fa_search_lds_bytes();
	switch (hist_class(P)) {
	case 4: hipLaunchKernelGGL(k_search<4>, grid, block, lds, st, P, B, pcm, stats, cands, coefs, plans); break;
	case 12: hipLaunchKernelGGL(k_search<12>, grid, block, lds, st, P, B, pcm, stats, cands, coefs, plans); break;
	default: hipLaunchKernelGGL(k_search<32>, grid, block, lds, st, P, B, pcm, stats, cands, coefs, plans); break;
	}
}

void fa_launch_decide(const FaEncParams &P, const FaBatch &B, const FaPlan *plans, uint32_t *assign, uint32_t *loose_out, hipStream_t st)
{
	hipLaunchKernelGGL(k_decide, dim3((B.nframes + 255) / 256), dim3(256), 0, st, P, B, plans, assign, loose_out);
}

void fa_launch_pack(const FaEncParams &P, const FaBatch &B, const void *pcm, const FaSubStat *stats, const FaPlan *plans, const uint32_t *assign,
                    uint32_t *slots, uint32_t *frame_bytes, uint32_t *errflag, hipStream_t st)
{
	const dim3 grid(B.nframes), block(FA_THREADS);
	switch (hist_class(P)) {
	case 4: hipLaunchKernelGGL(k_pack<4>, grid, block, 0, st, P, B, pcm, stats, plans, assign, slots, frame_bytes, errflag); break;
	case 12: hipLaunchKernelGGL(k_pack<12>, grid, block, 0, st, P, B, pcm, stats, plans, assign, slots, frame_bytes, errflag); break;
	default: hipLaunchKernelGGL(k_pack<32>, grid, block, 0, st, P, B, pcm, stats, plans, assign, slots, frame_bytes, errflag); break;
	}
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
