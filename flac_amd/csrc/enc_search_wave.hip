// FLAC frame encoder kernels for gfx950 (MI355X): the model search with a wave per subframe.
#include <hip/hip_runtime.h>

#include "enc_kernels.h"
#include "enc_wave.h"

// ---------------------------------------------------------------------------------------------
// k_search_wave: process_subframe_ (stream_encoder.c:3164) for one (frame, signal) of a stream with
// blocks of 4096 samples, run by ONE wave that never waits for another.
//
// Lane l owns samples [64 l, 64 l + 64) of the frame and keeps them, with twelve samples of history,
// in registers for the whole search.  With at most 2^6 Rice partitions in a frame a lane's samples
// never straddle a partition, so the |residual| sum of a candidate's finest partitions
// (precompute_partition_info_sums_, stream_encoder.c:3774) is one register per lane: no atomics, no
// shared memory, and it is already in the layout the pricing step (set_partitioned_rice_,
// stream_encoder.c:3943) wants, lane l holding finest sum l.  The candidates come one after the other
// in the reference's trial order; their parameters and coefficients are the same in every lane and
// live in scalar registers, loaded with scalar loads.  Four such waves share a SIMD and fill each
// other's stalls; nothing in the kernel is a barrier between waves.
// ---------------------------------------------------------------------------------------------
// Sixteen samples' sums are finished (shifted, added up) before the next sixteen start: the optimiser,
// left alone, sinks every shift and add to the end of the candidate, keeps all 64 chains alive until
// then and runs out of registers.  An empty volatile statement that "uses" the running sum pins them.
#define SW_FENCE() asm volatile("" : "+v"(t))


// x[i] = B - residual for samples OFF .. OFF + 15 of the lane's run, two taps per v_dot2_i32_i16
// (the arithmetic of lpc_negres16_dot2 in enc_common.h; qf in scalar registers).  The tap groups
// nest: the first one the candidate's order does not reach ends the chain with one branch.
template <int OFF, int JJ>
__device__ __forceinline__ void sw_dot2_groups(const int32_t (&PW)[SW_W], const int32_t (&qf)[SW_NQF], uint32_t order, int32_t (&sum)[16])
{
	if constexpr (JJ < SW_NQF) {
		if ((uint32_t)(2 * JJ) <= order) {
#pragma unroll
			for (int i = 0; i < 16; i++)
				sum[i] = __builtin_amdgcn_sdot2(__builtin_bit_cast(fa_short2, PW[SW_HIST + OFF + i - 2 * JJ]), __builtin_bit_cast(fa_short2, qf[JJ]), sum[i], false);
			sw_dot2_groups<OFF, JJ + 1>(PW, qf, order, sum);
		}
	}
}

template <int OFF>
__device__ __forceinline__ void sw_negres_dot2(const int32_t (&PW)[SW_W], const int32_t (&qf)[SW_NQF], uint32_t order, uint32_t shift, uint32_t c31,
                                               uint32_t (&x)[16])
{
	int32_t sum[16];
#pragma unroll
	for (int i = 0; i < 16; i++)
		asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(sum[i]) : "v"(PW[SW_HIST + OFF + i]), "s"(qf[0]), "v"(c31));
	sw_dot2_groups<OFF, 1>(PW, qf, order, sum);
	if (shift) {
#pragma unroll
		for (int i = 0; i < 16; i++)
			x[i] = (uint32_t)sum[i] >> shift;
	} else {   // the fixed predictors
#pragma unroll
		for (int i = 0; i < 16; i++)
			x[i] = (uint32_t)sum[i];
	}
}

// The same for samples wider than 16 bits (up to 25): a sample is s = sh 2^12 + sl with sl its low 12 bits, both halves
// fit a 16-bit lane, so the folded sum  S = sum q[j] d[i-1-j] - d[i] 2^shift  is  A 2^12 + B  with A and B one chain
// of v_dot2_i32_i16 each (Ph / Pl: the halves as pairs, laid out like PW above).  What the reference does with S
// decides how A and B are put together (the candidate's fold, worked out once per candidate by sw_split_fold):
//   SWF_W  wrapping 32-bit sum (lpc.c:265)            x = ((A << 12) + B) >> shift        B = 2^(31-shift)
//   SWF_S  64-bit sum (lpc.c:531), shift <= 12        x = (A << (12-shift)) + (B >> shift) B = 2^31: A 2^12 is a multiple of
//                                                     2^shift, so floor(S / 2^shift) splits exactly
//   SWF_L  64-bit sum, shift > 12                     x = (A + (B >> 12)) >> (shift-12)    B = 2^(43-shift): the low 12 bits
//                                                     of B cannot carry, floor(S / 2^shift) = floor((A + floor(B / 2^12)) / 2^(shift-12))
// with x = B - residual as in the 16-bit form: the 2^31 rides in A's starting value a0 = 2^31 >> (A's left shift), the
// outer shift is logical, B's inner one arithmetic.  Exact while |S| (and B itself, where it is shifted) stay inside
// 32 bits, which sw_split_fold checks from the sum of the coefficients' magnitudes.
#define SWF_W 0
#define SWF_S 1
#define SWF_L 2

struct SwFold {
	uint32_t form, sa, sb, sc, a0;
	bool exact;
};

__device__ __forceinline__ SwFold sw_split_fold(uint64_t sumq, uint32_t shift, uint32_t sbps, bool wide64)
{
	SwFold f;
	const uint64_t lim = 1ull << 31;
	const uint64_t all = sumq + (1ull << shift);                  // the folded chain's coefficient magnitudes
	const bool b_fits = all * 4095u < lim;                        // |B| <= all * (2^12 - 1)
	if (!wide64) {
		f.form = SWF_W; f.sa = 12; f.sb = 0; f.sc = shift;
		f.exact = fa_fold_exact(sumq, shift, sbps);
	} else if (shift <= 12) {
		f.form = SWF_S; f.sa = 12 - shift; f.sb = shift; f.sc = 0;
		// |prediction - sample| <= (sumq / 2^shift + 1) 2^(sbps-1), rounded up
		f.exact = b_fits && (((sumq >> shift) + 2) << (sbps - 1)) < lim;
	} else {
		f.form = SWF_L; f.sa = 0; f.sb = 12; f.sc = shift - 12;
		// |A + (B >> 12)| <= all * 2^(sbps-1) / 2^12, rounded up
		f.exact = shift <= 15 && b_fits && ((all << (sbps > 13 ? sbps - 13 : 0)) + (1u << 19)) < lim;
	}
	f.a0 = 0x80000000u >> f.sa;
	return f;
}

template <int OFF, int JJ>
__device__ __forceinline__ void sw_split_groups(const int32_t (&Ph)[SW_W], const int32_t (&Pl)[SW_W], const int32_t (&qf)[SW_NQF], uint32_t order,
                                                int32_t (&A)[16], int32_t (&Bs)[16])
{
	if constexpr (JJ < SW_NQF) {
		if ((uint32_t)(2 * JJ) <= order) {
#pragma unroll
			for (int i = 0; i < 16; i++) {
				A[i] = __builtin_amdgcn_sdot2(__builtin_bit_cast(fa_short2, Ph[SW_HIST + OFF + i - 2 * JJ]), __builtin_bit_cast(fa_short2, qf[JJ]), A[i], false);
				Bs[i] = __builtin_amdgcn_sdot2(__builtin_bit_cast(fa_short2, Pl[SW_HIST + OFF + i - 2 * JJ]), __builtin_bit_cast(fa_short2, qf[JJ]), Bs[i], false);
			}
			sw_split_groups<OFF, JJ + 1>(Ph, Pl, qf, order, A, Bs);
		}
	}
}

template <int OFF>
__device__ __forceinline__ void sw_negres_split(const int32_t (&Ph)[SW_W], const int32_t (&Pl)[SW_W], const int32_t (&qf)[SW_NQF], uint32_t order,
                                                const SwFold &f, uint32_t a0v, uint32_t (&x)[16])
{
	int32_t A[16], Bs[16];
#pragma unroll
	for (int i = 0; i < 16; i++) {
		asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(A[i]) : "v"(Ph[SW_HIST + OFF + i]), "s"(qf[0]), "v"(a0v));
		asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(Bs[i]) : "v"(Pl[SW_HIST + OFF + i]), "s"(qf[0]));
	}
	sw_split_groups<OFF, 1>(Ph, Pl, qf, order, A, Bs);
	if (f.form == SWF_W) {
		if (f.sc) {
#pragma unroll
			for (int i = 0; i < 16; i++)
				x[i] = (((uint32_t)A[i] << 12) + (uint32_t)Bs[i]) >> f.sc;
		} else {   // the fixed predictors
#pragma unroll
			for (int i = 0; i < 16; i++)
				x[i] = ((uint32_t)A[i] << 12) + (uint32_t)Bs[i];
		}
	} else if (f.form == SWF_S) {
#pragma unroll
		for (int i = 0; i < 16; i++)
			x[i] = ((uint32_t)A[i] << f.sa) + (uint32_t)(Bs[i] >> f.sb);
	} else {
#pragma unroll
		for (int i = 0; i < 16; i++)
			x[i] = ((uint32_t)A[i] + (uint32_t)(Bs[i] >> 12)) >> f.sc;
	}
}

// the warm-up samples (the first `order` of the frame: lane 0, first block only) count as zero
template <int I>
__device__ __forceinline__ void sw_warm_up(uint32_t (&x)[16], uint32_t Bc, uint32_t order, bool lane0)
{
	if constexpr (I < SW_HIST) {
		if ((uint32_t)I < order) {
			x[I] = lane0 ? Bc : x[I];
			sw_warm_up<I + 1>(x, Bc, order, lane0);
		}
	}
}

// |residual| of sixteen samples added to the lane's sum.  partial (the same in every lane): the frame is
// the stream's short last one and only the first nv0 - OFF of these sixteen samples belong to it.
template <int OFF>
__device__ __forceinline__ uint32_t sw_add_abs(uint32_t (&x)[16], uint32_t Bc, uint32_t order, bool lane0, bool partial, uint32_t nv0, uint32_t t)
{
	if (OFF == 0)
		sw_warm_up<0>(x, Bc, order, lane0);
	if (partial) {
		// (an opaque count: its sixteen comparisons are made here, not ahead of the candidate loop and kept)
		uint32_t nv = nv0 > (uint32_t)OFF ? nv0 - (uint32_t)OFF : 0u;
		asm volatile("" : "+v"(nv));
#pragma unroll
		for (int i = 0; i < 16; i++)
			t = sad_u32((uint32_t)i < nv ? x[i] : Bc, Bc, t);
		return t;
	}
#pragma unroll
	for (int i = 0; i < 16; i++)
		t = sad_u32(x[i], Bc, t);
	return t;
}

// The lane's sum for a candidate whose folded chain could leave 32 bits (fa_fold_exact fails: the
// coefficients are too large for the shortcut, which does not happen with coefficients that come out
// of a real signal).  Spelled out tap by tap in 64 bits from the samples in memory; wrap: the
// candidate's sum is the reference's wrapping 32-bit one (lpc.c:265) rather than the 64-bit one
// (lpc.c:531).  There for correctness, not speed.
__device__ __forceinline__ uint32_t sw_lane_sum_exact(const void *pcm, uint32_t channels, uint32_t sample_bytes, uint32_t pairs, uint64_t frame_start,
                                                   uint32_t s, uint32_t wasted, uint32_t ln, const int32_t *q, uint32_t order, int shift, bool wrap, uint32_t bsf)
{
	uint32_t t = 0;
#pragma nounroll
	for (uint32_t i = 0; i < SW_N; i++) {
		const uint32_t idx = ln * SW_N + i;
		if (idx < order || idx >= bsf)
			continue;
		int64_t sum = 0;
#pragma nounroll
		for (uint32_t j = 0; j < order; j++)
			sum += (int64_t)q[j] * (int64_t)(fetch_signal(pcm, channels, sample_bytes, pairs, frame_start + idx - 1 - j, s) >> wasted);
		const int32_t pred = wrap ? ((int32_t)sum >> shift) : (int32_t)(sum >> shift);
		const int32_t d = fetch_signal(pcm, channels, sample_bytes, pairs, frame_start + idx, s) >> wasted;
		t += abs32((int32_t)((uint32_t)d - (uint32_t)pred));
	}
	return t;
}

// rice_k_from_sum32 / rice_bits_estimate32 (enc_common.h; stream_encoder.c:4047-4054, 3918) as straight-line
// code: selects instead of branches on per-lane conditions, 24-bit multiplies (n <= 4096, k < 32)
__device__ __forceinline__ uint32_t sw_rice_k32(uint32_t sum, uint32_t n, uint32_t limit)
{
	const uint32_t cs = umin32((uint32_t)__clz((int)sum), 32u), cn = (uint32_t)__clz((int)n);   // n >= 1
	uint32_t k = cn > cs ? cn - cs : 0u;       // ilog2(sum) - ilog2(n) when sum > n
	k += ((n << k) < sum) ? 1u : 0u;
	k = sum > n ? k : 0u;
	return umin32(k, limit - 1u);
}

__device__ __forceinline__ uint32_t sw_rice_bits32(uint32_t k, uint32_t n, uint32_t sum)
{
	const uint32_t tail = k ? (sum >> ((k - 1u) & 31u)) : (sum << 1);
	return 4u + __umul24(1u + k, n) + tail - (n >> 1);
}

#define SW_DPP_ADD(x, ctrl) ((x) + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), ctrl, 0xf, 0xf, true))
#define SW_DPP_PAIRS 0xB1   /* quad_perm [1,0,3,2] */
#define SW_DPP_QUADS 0x4E   /* quad_perm [2,3,0,1] */
#define SW_DPP_EIGHTS 0x141 /* row_half_mirror */
#define SW_DPP_ROWS 0x140   /* row_mirror */

// set_partitioned_rice_ (stream_encoder.c:3943) for every partition order of one candidate and the
// scan of find_best_partition_order_ (stream_encoder.c:3700-3729), from the finest sums (lane l < 2^pm
// holds sum l, truncated to 32 bits as the reference's accumulator is, stream_encoder.c:3803-3833).
// Lane c prices finest partition c and "coarser item c": the partition whose level is the highest zero
// bit of c below bit pm and whose index is the bits of c below that (every partition of every coarser
// level is the item of exactly one lane, and those of one level sit in adjacent lanes).  A coarser
// partition is a run of finest ones, so its sum is a difference of two prefix sums, fetched with two
// lane permutes.  The bits of a level are then sums over aligned lane groups: four cross-lane adds and
// a few scalar reads.  Out: the best order, its residual bits, and per lane the Rice parameters of its
// two items (finest | coarser << 8).
// What a lane's two items are does not depend on the candidate: SwLane holds it, worked out once per wave.
struct SwLane {
	uint32_t packed;             // first << 2 | last << 10 | poc << 18: the finest partitions [first, last] the coarser item covers, and its level
	bool mine, have_c, lane0, pc0;   // lane prices a finest partition / a coarser one; the items that start the frame (they hold the warm-up samples)
};

__device__ __forceinline__ SwLane sw_lane_items(uint32_t ln, uint32_t pm, uint32_t pmn)
{
	SwLane L;
	const uint32_t full = (1u << pm) - 1u;
	const uint32_t zeros = ~ln & full;
	const uint32_t poc = zeros ? ilog2u(zeros) : 0u;
	const uint32_t pc = ln & ((1u << poc) - 1u);
	const uint32_t span = (pm - poc) & 31u;
	const uint32_t first = (pc << span) & 63u, last = (first + (1u << span) - 1u) & 63u;
	L.packed = (first << 2) | (last << 10) | (poc << 18);
	L.mine = ln <= full;
	L.have_c = L.mine && zeros != 0 && poc >= pmn;
	L.lane0 = ln == 0;
	L.pc0 = pc == 0;
	return L;
}

__device__ __forceinline__ void sw_price(uint64_t fr64, uint32_t pm, uint32_t pmn, uint32_t order, uint32_t seed, uint32_t bsf, uint32_t rice_limit,
                                         uint32_t lane, const SwLane &L, uint32_t &bestb_out, uint32_t &bestpo_out, uint32_t &kpair_out)
{
	const bool mine = L.mine, have_c = L.have_c;
	const uint32_t poc = (L.packed >> 18) & 7u;
	const uint32_t nf = (bsf >> pm) - (L.lane0 ? order : 0u);
	const uint32_t nc = (bsf >> poc) - (L.pc0 ? order : 0u);
	// 32-bit arithmetic when every finest sum is below 2^26 (no sum of 64 of them reaches 2^32)
	const bool narrow = __all((fr64 >> 26) == 0);
	const uint32_t fr = (uint32_t)fr64;
	uint32_t bf, bc, kf, kc;
	if (narrow) {
		const uint32_t incl = sw_scan_incl(fr);
		const uint32_t excl = incl - fr;
		const uint32_t sc = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((L.packed >> 8) & 0xfcu), (int)incl) -
		                    (uint32_t)__builtin_amdgcn_ds_bpermute((int)(L.packed & 0xfcu), (int)excl);
		kf = (pm == 0) ? seed : sw_rice_k32(fr, nf, rice_limit);
		kc = (poc == 0) ? seed : sw_rice_k32(sc, nc, rice_limit);
		bf = sw_rice_bits32(kf, nf, fr);
		bc = sw_rice_bits32(kc, nc, sc);
	} else {
		// (sums that need 64 bits: pathological input; the exchange pattern of the workgroup form)
		uint32_t ln = lane;
		asm volatile("" : "+v"(ln));
		const uint32_t tz = ln ? (uint32_t)__builtin_ctz(ln) : 0u;
		const uint32_t pc = ln & ((1u << poc) - 1u);
		const uint32_t src = (2u * pc + 1u) << ((pm - 1u - poc) & 31u);   // the lane that holds the item's sum after the exchanges
		uint64_t v[7];
		v[0] = fr64;
#pragma unroll
		for (int j = 0; j < 6; j++)
			v[j + 1] = v[j] + (uint64_t)__shfl_xor((unsigned long long)v[j], 1 << j, 64);
		uint64_t up = v[1];
#pragma unroll
		for (int j = 1; j < 6; j++)
			up = tz == (uint32_t)j ? v[j + 1] : up;
		const uint64_t sc = (uint64_t)__shfl((unsigned long long)up, (int)src, 64);
		kf = (pm == 0) ? seed : rice_k_from_sum(v[0], nf, rice_limit);
		kc = (poc == 0) ? seed : rice_k_from_sum(sc, nc, rice_limit);
		bf = rice_bits_estimate(kf, nf, v[0]);
		bc = rice_bits_estimate(kc, nc, sc);
	}
	kpair_out = kf | (kc << 8);
	// bits per level: the finest one is the sum over the wave, level po that of the 2^po lanes from lane 2^pm - 2^(po+1) on
	uint32_t wf[5], wc[5];
	wf[0] = mine ? bf : 0u;
	wc[0] = have_c ? bc : 0u;
	wf[1] = SW_DPP_ADD(wf[0], SW_DPP_PAIRS);
	wc[1] = SW_DPP_ADD(wc[0], SW_DPP_PAIRS);
	wf[2] = SW_DPP_ADD(wf[1], SW_DPP_QUADS);
	wc[2] = SW_DPP_ADD(wc[1], SW_DPP_QUADS);
	wf[3] = SW_DPP_ADD(wf[2], SW_DPP_EIGHTS);
	wc[3] = SW_DPP_ADD(wc[2], SW_DPP_EIGHTS);
	wf[4] = SW_DPP_ADD(wf[3], SW_DPP_ROWS);
	wc[4] = SW_DPP_ADD(wc[3], SW_DPP_ROWS);
	// find_best_partition_order_: from the largest order down, a smaller one must be strictly better
	uint32_t bestb = 6u + __builtin_amdgcn_readlane(wf[4], 0) + __builtin_amdgcn_readlane(wf[4], 16) + __builtin_amdgcn_readlane(wf[4], 32) +
	                 __builtin_amdgcn_readlane(wf[4], 48);
	uint32_t bestpo = pm;
#pragma unroll
	for (int po = 5; po >= 0; po--) {
		if ((uint32_t)po < pm && (uint32_t)po >= pmn) {
			const uint32_t at = (1u << pm) - (2u << po);
			const uint32_t lb = po == 5 ? __builtin_amdgcn_readlane(wc[4], 0) + __builtin_amdgcn_readlane(wc[4], 16)
			                            : __builtin_amdgcn_readlane(wc[po], at);
			if (6u + lb < bestb) {
				bestb = 6u + lb;
				bestpo = (uint32_t)po;
			}
		}
	}
	bestb_out = bestb;
	bestpo_out = bestpo;
}

__device__ __forceinline__ int32_t sw_pack16(int32_t hi, int32_t lo) { return (int32_t)(((uint32_t)hi << 16) | ((uint32_t)lo & 0xffffu)); }

// One candidate predictor, the same in every lane.
struct SwCand {
	uint32_t kind, order, seed, precision, shift, mode, hdr_bits, slot;
	uint64_t sumq;   // an upper bound of the sum of the coefficients' magnitudes
	int32_t q[SW_HIST];
};

#define SW_TAPS 0
#define SW_DOT2 1
#define SW_SPLIT 2   // samples of up to 25 bits as two 16-bit halves (sw_negres_split): streams wider than 16 bits
#define SW_REDO 0xffu   // plan type left by the two-taps launch for the one-tap launch to pick up

// PAIRS: the samples are 16-bit stereo pairs, one 32-bit word per wide sample (the only layout compiled in then)
#ifndef SW_WAVES
#define SW_WAVES 4   // waves per SIMD: 128 registers each
#endif
template <int RES, bool PAIRS>
#ifndef SW_SPLIT_WAVES
#define SW_SPLIT_WAVES 2   // the split form: 229 registers (at three waves 168 and 324 bytes of scratch per lane: 0.75 -> 0.93 ms for eight channels of 24 bits)
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RES == SW_SPLIT ? SW_SPLIT_WAVES : SW_WAVES, RES == SW_SPLIT ? SW_SPLIT_WAVES : SW_WAVES)))
void k_search_wave(FaEncParams P, FaBatch B, const void *__restrict__ pcm, const FaSubStat *__restrict__ stats, const FaLpcCand *__restrict__ cands,
                   const int32_t *__restrict__ coefs, FaPlan *__restrict__ plans, unsigned long long sig_list, uint32_t nsel)
{
	__shared__ __attribute__((aligned(16))) int32_t stage[17 * SW_ROW];
	// Workgroups go to the chip's eight dies in turn, each with an L2 of its own: the waves of one frame's
	// signals, which read the same samples, are given indices that are equal modulo eight.
	const uint32_t f = (blockIdx.x / (8 * nsel)) * 8 + (blockIdx.x & 7u);
	if (f >= B.nframes)
		return;
	const uint32_t s = (uint32_t)(sig_list >> (4 * ((blockIdx.x % (8 * nsel)) >> 3))) & 15u;
	const uint32_t task = f * P.nsignals + s;
	const uint32_t ln = threadIdx.x;
	FaPlan *pl = plans + task;
	if (RES == SW_TAPS && pl->type != SW_REDO)
		return;   // the two-taps launch has done this one
	const uint32_t bsf = frame_len(B, SW_N * 64, f);
	const bool partial = bsf != SW_N * 64;   // the stream's short last frame
	const uint64_t frame_start = (uint64_t)f * (SW_N * 64);
	const uint32_t ncoef_all = P.ncand * P.max_lpc_order;
	const FaLpcCand *my_cands = cands + (size_t)task * P.ncand;
	const int32_t *my_coefs = coefs + (size_t)task * ncoef_all;

	const FaSubStat st = stats[task];
	const uint32_t sbps = st.bps, wasted = st.wasted;
	const uint32_t hdr = 8 + wasted;
	// process_subframes_ (stream_encoder.c:2929-2936): a full block has twelve factors of two, a short
	// one is coded with partition order 0; format.c:548 never lowers the limit here (64 > 12)
	// (a full block of a stream whose limit is below 6 has coarser finest partitions than a lane's run: the sums of
	// 2^(6 - pm) lanes are put together in consider())
	const uint32_t pm = partial ? 0u : umin32(P.max_po, 6u);
	const uint32_t pmn = umin32(P.min_po, pm);

	// evaluate_verbatim_subframe_ (stream_encoder.c:3641), evaluate_constant_subframe_ (stream_encoder.c:3470)
	uint32_t best_bits = (P.disable_verbatim && bsf >= 4) ? 0xffffffffu : hdr + bsf * sbps;
	uint32_t best_type = FA_SUB_VERBATIM, best_order = 0, best_prec = 0, best_po = 0, best_mode = 0, best_slot = 0, best_shift = 0, best_c = 0;
	int32_t best_const = 0;
	if (bsf >= 4 && st.is_constant) {
		const uint32_t cb = hdr + sbps;
		if (cb < best_bits) {
			best_type = FA_SUB_CONSTANT;
			best_bits = cb;
			best_const = st.first_sample;
		}
	}
	uint32_t best_k = 0;   // per lane: Rice parameters of the best candidate's two items

	if (bsf >= 4 && !st.is_constant) {
		// ---- the lane's run of the signal: a quarter of the frame at a time goes through LDS, read with
		// lanes along consecutive samples, stored as rows of 64, and sixteen lanes pick up a row each
		// plus the tail of the row before it ----
		int32_t d[SW_W];
		SigAccess acc = sig_access(P.channels, B.sample_bytes, fa_layout(B), s);
		if (PAIRS)
			acc.fmt = 0;
		sw_load_run(d, stage, pcm, acc, frame_start, bsf, partial, wasted, ln);
		// Which arithmetic the residuals take is decided by what the samples are, not by what they could
		// be: two taps per v_dot2_i32_i16 when every sample of the frame's signal fits 16 bits (always so
		// for streams of up to 16 bits, nearly always so for their side signal), a tap per v_mad_i32_i24
		// otherwise.  Both are exact, so the choice does not show in the result.  Each form is a launch of
		// its own (the registers of one do not weigh on the other): the two-taps launch comes first and
		// leaves the subframes it cannot take to the second, which looks at nothing else.
		// (w takes the place of d, last word first, so that the two never exist side by side: 152 registers otherwise)
		int32_t (&w)[SW_W] = d;
		if (RES == SW_DOT2) {
			if (P.bps + ((P.do_ms && s == P.channels + 1) ? 1u : 0u) > 16) {
				uint32_t mag = 0;
#pragma unroll
				for (int k = 0; k < SW_W; k++)
					mag |= (uint32_t)(d[k] ^ (d[k] >> 31));
				if (wave_or(mag) >= 0x8000u) {
					if (ln == 0)
						pl->type = SW_REDO;
					return;
				}
			}
			// w[k] holds samples k - 1 and k of the window in its low and high half
#pragma unroll
			for (int k = SW_W - 1; k >= 1; k--)
				w[k] = (int32_t)__builtin_amdgcn_perm((uint32_t)d[k], (uint32_t)d[k - 1], 0x05040100u);
			w[0] = (int32_t)((uint32_t)d[0] << 16);
		}
		// the split form: w becomes the high halves' pairs, wl the low halves' (12 bits, unsigned)
		int32_t wl[RES == SW_SPLIT ? SW_W : 1];
		if constexpr (RES == SW_SPLIT) {
#pragma unroll
			for (int k = SW_W - 1; k >= 1; k--) {
				wl[k] = (int32_t)((((uint32_t)d[k] & 0xfffu) << 16) | ((uint32_t)d[k - 1] & 0xfffu));
				w[k] = (int32_t)__builtin_amdgcn_perm((uint32_t)(d[k] >> 12), (uint32_t)(d[k - 1] >> 12), 0x05040100u);
			}
			wl[0] = (int32_t)(((uint32_t)d[0] & 0xfffu) << 16);
			w[0] = (int32_t)((uint32_t)(d[0] >> 12) << 16);
		}
		const uint32_t fold_bps = RES == SW_DOT2 ? umin32(sbps, 16u) : sbps;
		const uint32_t c31 = 0x80000000u;
		const bool lane0 = ln == 0;
		const uint32_t nv0 = bsf > ln * SW_N ? bsf - ln * SW_N : 0u;   // samples of the lane's run inside the frame
		const uint32_t total_cands = 5 + P.ncand;
		const uint32_t rice_limit = P.rice_limit, max_lpc_order = P.max_lpc_order;
		const SwLane items = sw_lane_items(ln, pm, pmn);

		// The fixed predictors that take part (stream_encoder.c:3236-3287), six bits each: taking part, and the
		// Rice parameter their residual starts from.
		uint32_t fx = 0;
		{
			uint32_t lo, hi;
			if (P.exhaustive) { lo = 0; hi = 4; }
			else lo = hi = st.fixed_guess;
			if (hi >= bsf) hi = bsf - 1;
			const bool enabled = !P.disable_fixed || (P.max_lpc_order == 0 && best_bits == 0xffffffffu);
#pragma unroll
			for (uint32_t o = 0; o < 5; o++) {
				const float rb = st.rbps[o];
				if (enabled && o >= lo && o <= hi && !(rb >= (float)sbps)) {
					uint32_t rice = (rb > 0.0f) ? (uint32_t)((double)rb + 0.5) : 0u;
					rice++;
					if (rice >= rice_limit) rice = rice_limit - 1;
					fx |= (32u | rice) << (6 * o);
				}
			}
			fx = __builtin_amdgcn_readfirstlane(fx);
		}

		// The words of LPC candidate c as they lie in memory, still in vector registers: with two waves per SIMD (the
		// split form) there is too little else going on to hide two dependent trips to memory per candidate, so that
		// form asks for candidate c + 1's words before it starts on candidate c.
		struct SwRaw {
			uint2 head;
			int32_t q[SW_HIST];
		};
		auto fetch = [&](uint32_t c, SwRaw &r) {
			if (c >= 5 && c < total_cands) {
				r.head = *(const uint2 *)(my_cands + (c - 5));
				const int32_t *cq = my_coefs + (c - 5) * max_lpc_order;
#pragma unroll
				for (int j = 0; j < SW_HIST; j++)
					r.q[j] = cq[j];   // the buffer ends in slack, so the twelve reads are always inside it
			}
		};
		// candidate c of the reference's trial order: fixed orders ascending, then the LPC candidates as
		// k_lpc listed them.  False: it does not take part.  pre: its words, fetched earlier (else they are read here).
		auto candidate = [&](uint32_t c, SwCand &k, const SwRaw *pre = nullptr) -> bool {
#pragma unroll
			for (int j = 0; j < SW_HIST; j++)
				k.q[j] = 0;
			if (c < 5) {
				// a fixed predictor of order o is the filter {1}, {2,-1}, {3,-3,1}, {4,-6,4,-1} with shift 0 (fixed.c:352-388)
				const uint32_t e = (fx >> (6 * c)) & 63u;
				if (!(e & 32u))
					return false;
				const uint32_t order = c;
				k.kind = FA_SUB_FIXED;
				k.order = order;
				k.seed = e & 31u;
				k.precision = 0;
				k.shift = 0;
				k.slot = 0;
				k.hdr_bits = hdr + order * sbps;
				k.sumq = (1u << order) - 1u;
				k.mode = (uint32_t)lpc_mode(0, sbps);
				k.q[0] = (int32_t)order;
				k.q[1] = order == 2 ? -1 : order == 3 ? -3 : order == 4 ? -6 : 0;
				k.q[2] = order == 3 ? 1 : order == 4 ? 4 : 0;
				k.q[3] = order == 4 ? -1 : 0;
				return true;
			}
			const uint32_t slot = c - 5;
			// (the slot is the same in every lane: its words and the coefficients below belong in scalar registers)
			const uint2 lcw = pre ? pre->head : *(const uint2 *)(my_cands + slot);
			const uint32_t w0 = __builtin_amdgcn_readfirstlane(lcw.x), w1 = __builtin_amdgcn_readfirstlane(lcw.y);
			if (!(w0 & 0xffu))   // FaLpcCand: valid, order, precision, shift | rice, wide, sumq32
				return false;
			const uint32_t order = (w0 >> 8) & 0xffu;
			k.kind = FA_SUB_LPC;
			k.order = order;
			k.precision = (w0 >> 16) & 0xffu;
			k.shift = (uint32_t)(int32_t)(int8_t)(w0 >> 24);
			k.seed = w1 & 0xffu;
			k.mode = (uint32_t)lpc_mode((w1 >> 8) & 0xffu, sbps);
			k.slot = slot;
			k.hdr_bits = hdr + 4 + 5 + order * (k.precision + sbps);
			k.sumq = (uint64_t)(w1 >> 16) << 5;
			const int32_t *cq = my_coefs + slot * max_lpc_order;
#pragma unroll
			for (int j = 0; j < SW_HIST; j++) {
				const int32_t qj = (int32_t)__builtin_amdgcn_readfirstlane((uint32_t)(pre ? pre->q[j] : cq[j]));   // the buffer ends in slack, so the twelve reads are always inside it
				k.q[j] = (uint32_t)j < order ? qj : 0;
			}
			return true;
		};
		// price candidate c from its lanes' sums and keep it if it beats what came before: in trial order a
		// later one must be strictly better (stream_encoder.c:3283,3372); `late` candidates are looked at after
		// the others and win ties against those that follow them in trial order
		auto consider = [&](uint32_t c, const SwCand &k, uint32_t t, bool late) {
			// the finest partitions of the frame cover 2^(6 - pm) lanes each; lane p takes the sum of partition p
			// (modulo 2^32, which is the reference's accumulator at these sizes, stream_encoder.c:3803)
			uint64_t fr = t;
			if (pm < 6) {
				if (RES == SW_SPLIT) {
					// wide samples: the sum of several lanes' runs can pass 2^32, where the reference switches to a 64-bit
					// accumulator by the partition's length and the sample width (stream_encoder.c:3803), not by the sum
#pragma unroll
					for (int j = 0; j < 6; j++)
						if ((uint32_t)j < 6 - pm)
							fr += (uint64_t)__shfl_xor((unsigned long long)fr, 1 << j, 64);
					fr = (uint64_t)__shfl((unsigned long long)fr, (int)((ln << (6 - pm)) & 63u), 64);
					if (ilog2u(bsf >> pm) + sbps < 32)
						fr &= 0xffffffffull;
				} else {
					uint32_t f32 = t;
#pragma unroll
					for (int j = 0; j < 6; j++)
						if ((uint32_t)j < 6 - pm)
							f32 += (uint32_t)__shfl_xor((int)f32, 1 << j, 64);
					fr = (uint32_t)__shfl((int)f32, (int)((ln << (6 - pm)) & 63u), 64);
				}
				if (ln >= (1u << pm))
					fr = 0;
			}
			uint32_t bestb, bestpo, kpair;
			sw_price(fr, pm, pmn, k.order, k.seed, bsf, rice_limit, ln, items, bestb, bestpo, kpair);
			const uint32_t total = k.hdr_bits + bestb;
			if (total < best_bits || (late && total == best_bits && c < best_c)) {
				best_bits = total;
				best_type = k.kind;
				best_order = k.order;
				best_prec = k.precision;
				best_shift = k.shift;
				best_po = bestpo;
				best_mode = k.mode;
				best_slot = k.slot;
				best_c = c;
				best_k = kpair;
			}
		};

		bool any_late = false;
		SwRaw ahead;
		if constexpr (RES == SW_SPLIT)
			fetch(5, ahead);
#pragma nounroll
		for (uint32_t c = 0; c < total_cands; c++) {
			SwCand k;
			bool takes_part;
			if constexpr (RES == SW_SPLIT) {
				const SwRaw mine = ahead;
				if (c >= 5)
					fetch(c + 1, ahead);
				takes_part = candidate(c, k, &mine);
			} else
				takes_part = candidate(c, k);
			if (!takes_part)
				continue;
			SwFold fold;
			if (RES == SW_SPLIT)
				fold = sw_split_fold(k.sumq, k.shift, sbps, k.mode == 2u);
			if (RES == SW_SPLIT ? !fold.exact : !fa_fold_exact(k.sumq, k.shift, fold_bps)) {
				any_late = true;   // the folded chain could leave 32 bits: done exactly, below
				continue;
			}
			const uint32_t order = k.order, shift = k.shift;
			const uint32_t Bc = c31 >> (RES == SW_SPLIT ? fold.sc : shift);
			uint32_t t = 0;
			uint32_t x[16];
			if constexpr (RES == SW_SPLIT) {
				int32_t qf[SW_NQF];
				qf[0] = sw_pack16(-(int32_t)(1u << shift), k.q[0]);
#pragma unroll
				for (int jj = 1; jj < SW_NQF; jj++)
					qf[jj] = sw_pack16(k.q[2 * jj - 1], 2 * jj < SW_HIST ? k.q[2 * jj] : 0);
				uint32_t a0v = fold.a0;
				asm volatile("" : "+v"(a0v));
				sw_negres_split<0>(w, wl, qf, order, fold, a0v, x);
				t = sw_add_abs<0>(x, Bc, order, lane0, partial, nv0, t);
				SW_FENCE();
				sw_negres_split<16>(w, wl, qf, order, fold, a0v, x);
				t = sw_add_abs<16>(x, Bc, order, lane0, partial, nv0, t);
				SW_FENCE();
				sw_negres_split<32>(w, wl, qf, order, fold, a0v, x);
				t = sw_add_abs<32>(x, Bc, order, lane0, partial, nv0, t);
				SW_FENCE();
				sw_negres_split<48>(w, wl, qf, order, fold, a0v, x);
				t = sw_add_abs<48>(x, Bc, order, lane0, partial, nv0, t);
			} else if constexpr (RES == SW_DOT2) {
				// c[0] = -2^shift, c[m] = q[m - 1], two to a word
				int32_t qf[SW_NQF];
				qf[0] = sw_pack16(-(int32_t)(1u << shift), k.q[0]);
#pragma unroll
				for (int jj = 1; jj < SW_NQF; jj++)
					qf[jj] = sw_pack16(k.q[2 * jj - 1], 2 * jj < SW_HIST ? k.q[2 * jj] : 0);
				sw_negres_dot2<0>(w, qf, order, shift, c31, x);
				t = sw_add_abs<0>(x, Bc, order, lane0, partial, nv0, t);
				SW_FENCE();
				sw_negres_dot2<16>(w, qf, order, shift, c31, x);
				t = sw_add_abs<16>(x, Bc, order, lane0, partial, nv0, t);
				SW_FENCE();
				sw_negres_dot2<32>(w, qf, order, shift, c31, x);
				t = sw_add_abs<32>(x, Bc, order, lane0, partial, nv0, t);
				SW_FENCE();
				sw_negres_dot2<48>(w, qf, order, shift, c31, x);
				t = sw_add_abs<48>(x, Bc, order, lane0, partial, nv0, t);
			} else {
				sw_negres_taps<0>(w, k.q, order, shift, c31, x);
				t = sw_add_abs<0>(x, Bc, order, lane0, partial, nv0, t);
				SW_FENCE();
				sw_negres_taps<16>(w, k.q, order, shift, c31, x);
				t = sw_add_abs<16>(x, Bc, order, lane0, partial, nv0, t);
				SW_FENCE();
				sw_negres_taps<32>(w, k.q, order, shift, c31, x);
				t = sw_add_abs<32>(x, Bc, order, lane0, partial, nv0, t);
				SW_FENCE();
				sw_negres_taps<48>(w, k.q, order, shift, c31, x);
				t = sw_add_abs<48>(x, Bc, order, lane0, partial, nv0, t);
			}
			consider(c, k, t, false);
		}
		if (any_late) {
#pragma nounroll
			for (uint32_t c = 0; c < total_cands; c++) {
				SwCand k;
				if (!candidate(c, k) || (RES == SW_SPLIT ? sw_split_fold(k.sumq, k.shift, sbps, k.mode == 2u).exact : fa_fold_exact(k.sumq, k.shift, fold_bps)))
					continue;
				const uint32_t t = sw_lane_sum_exact(pcm, P.channels, B.sample_bytes, fa_layout(B), frame_start, s, wasted, ln, k.q, k.order, (int)k.shift, k.mode != 2u, bsf);
				consider(c, k, t, true);
			}
		}
	}
	if (best_bits == 0xffffffffu) {
		// stream_encoder.c:3388-3391
		best_type = FA_SUB_VERBATIM;
		best_bits = hdr + bsf * sbps;
	}
	// ---- the plan ----
	uint32_t rice2 = 0;
	if (best_type == FA_SUB_FIXED || best_type == FA_SUB_LPC) {
		// the winner's parameters at its partition order: the finest items when that is the largest order, else
		// the coarser items of the lanes from 2^pm - 2^(po+1) on
		const uint32_t from = best_po == pm ? ln : ((1u << pm) - (2u << best_po) + ln) & 63u;
		const uint32_t kk = (uint32_t)__shfl((int)best_k, (int)from, 64);
		const uint32_t k = best_po == pm ? (kk & 0xffu) : (kk >> 8);
		const bool have = ln < (1u << best_po);
		if (have)
			pl->params[ln] = (uint8_t)k;
		rice2 = __ballot(have && k >= 15) ? 1u : 0u;   // stream_encoder.c:3752-3757
		if (best_type == FA_SUB_LPC && ln < best_order)
			pl->q[ln] = my_coefs[best_slot * P.max_lpc_order + ln];
	}
	if (ln == 0) {
		pl->bits = best_bits;
		pl->type = (uint8_t)best_type;
		pl->order = (uint8_t)best_order;
		pl->precision = (uint8_t)best_prec;
		pl->shift = (int8_t)best_shift;
		pl->part_order = (uint8_t)best_po;
		pl->rice2 = (uint8_t)rice2;
		pl->wide = (uint8_t)best_mode;
		pl->pad = 0;
		pl->constant = best_const;
	}
}

// Whether the wave form can take this stream's frames (the launcher still chooses the signals).
bool fa_search_wave_eligible(const FaEncParams &P)
{
	if (P.blocksize != SW_N * 64 || P.max_po > 6 || P.max_lpc_order > SW_HIST)
		return false;
	// the reference's 32-bit partition sums (stream_encoder.c:3803) at every partition order this block can use
	return 12 + P.bps + (P.do_ms ? 1u : 0u) < 32;
}

// Streams wider than that (20 to 24 bits, whose candidates mostly need the reference's 64-bit sums) take the same
// kernel in its split-sample form, two waves per SIMD; their partition sums are put together in 64 bits where the
// reference's accumulator is 64 bits wide.  Not with the search over coefficient precisions: with 137 candidates
// per subframe what a candidate costs besides its residuals (fetching it, pricing it: lane exchanges and scalar
// reads whose latency two waves per SIMD do not hide) decides, and the workgroup form, which prices two candidates
// per step at three waves per SIMD, is the faster one there (150 s of 24-bit / 96 kHz stereo at -8 -e -p: 4.53 ms
// against 5.64; eight channels of 24-bit / 192 kHz at -8, 17 candidates: 1.08 ms against 0.82).
bool fa_search_wave_split_eligible(const FaEncParams &P)
{
	if (P.blocksize != SW_N * 64 || P.max_po > 6 || P.max_lpc_order > SW_HIST || fa_search_wave_eligible(P) || P.prec_search)
		return false;
	return P.bps + (P.do_ms ? 1u : 0u) <= 25;
}

void fa_launch_search_wave_split(const FaEncParams &P, const FaBatch &B, const void *pcm, const FaSubStat *stats, const FaLpcCand *cands,
                                 const int32_t *coefs, FaPlan *plans, unsigned long long sig_list, uint32_t nsel, hipStream_t st)
{
	const dim3 grid((B.nframes + 7) / 8 * 8 * nsel);
	hipLaunchKernelGGL((k_search_wave<SW_SPLIT, false>), grid, dim3(64), 0, st, P, B, pcm, stats, cands, coefs, plans, sig_list, nsel);
}

// wide_list: those of the signals that can be wider than 16 bits (the rest never need the second launch)
void fa_launch_search_wave(const FaEncParams &P, const FaBatch &B, const void *pcm, const FaSubStat *stats, const FaLpcCand *cands,
                           const int32_t *coefs, FaPlan *plans, unsigned long long sig_list, uint32_t nsel, unsigned long long wide_list, uint32_t nwide,
                           hipStream_t st)
{
	const dim3 block(64);
	const dim3 grid((B.nframes + 7) / 8 * 8 * nsel), gridw((B.nframes + 7) / 8 * 8 * nwide);
	if (B.pairs) {
		hipLaunchKernelGGL((k_search_wave<SW_DOT2, true>), grid, block, 0, st, P, B, pcm, stats, cands, coefs, plans, sig_list, nsel);
		if (nwide)
			hipLaunchKernelGGL((k_search_wave<SW_TAPS, true>), gridw, block, 0, st, P, B, pcm, stats, cands, coefs, plans, wide_list, nwide);
	} else {
		hipLaunchKernelGGL((k_search_wave<SW_DOT2, false>), grid, block, 0, st, P, B, pcm, stats, cands, coefs, plans, sig_list, nsel);
		if (nwide)
			hipLaunchKernelGGL((k_search_wave<SW_TAPS, false>), gridw, block, 0, st, P, B, pcm, stats, cands, coefs, plans, wide_list, nwide);
	}
}
