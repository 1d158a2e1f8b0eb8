// The frame decoder proper (k_dec_frames and everything under it), as a header: each instance of the kernel
// is compiled in a translation unit of its own (dec_frames_*.hip), side by side, because one instance
// takes minutes to compile (every predictor-order class, three arithmetic variants, sixteen samples unrolled).
#ifndef FLAC_AMD_DEC_FRAMES_H
#define FLAC_AMD_DEC_FRAMES_H

#include <hip/hip_runtime.h>

#include <type_traits>

#include "dec_kernels.h"

#define DEC_LANES 64

__device__ __forceinline__ uint32_t d_ilog2(uint32_t v) { return 31u - (uint32_t)__clz((int)v); }
__device__ __forceinline__ uint32_t umin_u32(uint32_t a, uint32_t b) { return a < b ? a : b; }

// the fixed predictors as filter coefficients (fixed.c:395)
__constant__ static const int8_t c_fixed_q[5][4] = {{0, 0, 0, 0}, {1, 0, 0, 0}, {2, -1, 0, 0}, {3, -3, 1, 0}, {4, -6, 4, -1}};

// ---------------------------------------------------------------------------------------------
// bit reader (bitreader.c): MSB-first bits out of big-endian 32-bit words.
//
// A lane keeps two consecutive words of its frame in registers and counts the unread bits of the
// first (0..31); the next 32 bits of the stream are one v_alignbit_b32 of the pair, and moving on
// by up to 32 bits is a subtraction plus, when a word boundary is crossed, a step to the next word:
// no 64-bit shifts.  Words come from a per-lane ring in LDS that the whole wave refills together: 64
// lanes fetch 64 consecutive words of ONE lane's frame in a single coalesced load, between blocks of
// samples, long before that lane gets there.  A lane that outruns its ring (a very long unary run)
// reads global memory itself.
// ---------------------------------------------------------------------------------------------
#define DEC_MAXLW 16        // frames per wave at most, except for the largest batches (k_dec_frames<.., 32>)
#define RING_CHUNK 64       // words per refill (one per lane of the wave)
#define RING_WORDS_MIN 128  // words per lane in LDS: two chunks (the instance with 32 frames per wave), or four (RING_WORDS_BIG)
#define RING_WORDS_BIG 256
#define DEC_BLOCK 16        // samples per unrolled block
// samples decoded between refill / flush points where every lane has nothing but plain symbols ahead: as many as the
// ring is sure to hold while refills are in flight (see pump_service), four blocks with the deep ring, two without
#define DEC_RUN_BIG 64
#define DEC_RUN_MIN 32

struct BitRd {
	const uint32_t *wg;  // the frame's first word in global memory (word index wbase of the input)
	uint64_t wbase;      // index of that word in the input
	uint32_t cur;        // the word being read: its last `rem` bits are unread
	uint32_t nxt;        // the word after it, all unread (both already byte-swapped)
	uint32_t rem;        // 0..31
	uint32_t wrel;       // index, relative to wbase, of the word after nxt
	uint32_t nrel;       // words from wbase to the end of the input (saturated)
	uint32_t limit;      // words [0, limit) have been in the ring; the last rmask + 1 of them still are
	uint32_t rmask;      // words per lane in the ring, minus one
	uint32_t bend;       // first bit past the stream on the same scale (saturated)
	const uint32_t *ring;  // this lane's words of LDS
	bool err;
};

// Bits consumed, counted from the start of word wbase: cur and nxt are words wrel-2 and wrel-1 of the frame, of which
// rem + 32 bits are still unread.
__device__ __forceinline__ uint32_t br_pos(const BitRd &r) { return 32u * (r.wrel - 1u) - r.rem; }

// The refill protocol keeps limit - (words in the ring) <= wrel, so one compare tells whether the ring has
// the word; a lane that has outrun its ring reads global memory itself.
__device__ __forceinline__ uint32_t br_fetch(BitRd &r)
{
	uint32_t v;
	if (r.wrel < r.limit)
		v = r.ring[r.wrel & r.rmask];   // the ring holds the words already byte-swapped
	else
		v = __builtin_bswap32(r.wrel < r.nrel ? r.wg[r.wrel] : 0u);
	r.wrel++;
	return v;
}

// start reading at bit sbit (< 32 * RING_CHUNK - 96) of the frame's first chunk
__device__ __forceinline__ void br_start(BitRd &r, uint32_t sbit)
{
	r.err = false;
	const uint32_t i = sbit >> 5, s = sbit & 31;
	// a position on a word boundary is "nothing left of the word before"
	r.cur = s ? r.ring[i] : 0u;
	r.nxt = r.ring[s ? i + 1 : i];
	r.rem = (32u - s) & 31u;
	r.wrel = s ? i + 2 : i + 1;
}

// the next 32 bits: the last rem bits of cur, then the first 32 - rem of nxt (one v_alignbit_b32)
__device__ __forceinline__ uint32_t br_peek(const BitRd &r) { return __builtin_amdgcn_alignbit(r.cur, r.nxt, r.rem); }

// n <= 32
__device__ __forceinline__ void br_drop(BitRd &r, uint32_t n)
{
	const int32_t left = (int32_t)r.rem - (int32_t)n;
	if (left < 0) {
		r.cur = r.nxt;
		r.nxt = br_fetch(r);
	}
	r.rem = (uint32_t)left & 31u;
}

// n <= 32
__device__ __forceinline__ uint32_t br_get(BitRd &r, uint32_t n)
{
	if (n == 0)
		return 0;
	const uint32_t v = br_peek(r) >> (32 - n);
	br_drop(r, n);
	return v;
}

__device__ __forceinline__ int32_t br_get_signed(BitRd &r, uint32_t n)
{
	if (n == 0)
		return 0;
	const uint32_t v = br_get(r, n);
	return (int32_t)(v << (32 - n)) >> (32 - n);
}

// FLAC__bitreader_read_unary_unsigned (bitreader.c:666): zero bits before the next one bit
__device__ __forceinline__ uint32_t br_unary(BitRd &r)
{
	uint32_t n = 0;
	for (;;) {
		const uint32_t hi = br_peek(r);
		if (hi) {
			const uint32_t z = (uint32_t)__clz((int)hi);
			br_drop(r, z + 1);
			return n + z;
		}
		n += 32;
		br_drop(r, 32);
		if (br_pos(r) > r.bend) {
			r.err = true;
			return n;
		}
	}
}

// ---------------------------------------------------------------------------------------------
// k_dec_frames
// ---------------------------------------------------------------------------------------------
// MAXLW: the most frames a wave of this instance carries (the LDS it takes grows with it)
// (every lane's ring starts at a multiple of its size in LDS: the plain blocks step through it with one and-or per word)
template <int MAXLW, int RINGW>
struct __attribute__((aligned(RINGW * 4))) DecLds {
	static constexpr int maxlw = MAXLW;
	static constexpr int ring_words = RINGW;
	static constexpr int run = RINGW > RING_WORDS_MIN ? DEC_RUN_BIG : DEC_RUN_MIN;
	uint32_t ring[MAXLW][RINGW];             // input words per lane
	int32_t q[FA_MAX_ORDER][MAXLW];          // predictor coefficients per lane
	int32_t w[FA_MAX_ORDER][MAXLW];          // warm-up samples per lane
	// a run of decoded samples, [sample][lane]; the last word: for the lanes without a frame.  (The two-wave forms use a
	// row stride of MAXLW + 1 words: their way out reads a column per lane.)
	int32_t out[run * (MAXLW + 1) + 1];
};
typedef __attribute__((address_space(3))) const uint32_t LdsWord;

// What the two waves of the pipelined form (DEC_PIPE) pass each other: the entropy wave leaves up to a run of samples'
// worth in res[0] and res[1] in turn, how many in cnt[], and, once per subframe, what the restoring wave has to know
// about every lane's subframe.  A lane's words of a run are either values as read from the stream (residuals, warm-up
// and verbatim samples, a constant: kf = PIPE_RAW) or, from the plain blocks, the 32 bits that start with each Rice
// symbol, for the restoring wave to cut the symbol out of with all its lanes (kf = the Rice parameter).
#define PIPE_RAW 0x80u
template <int MAXLW, int RUN>
struct PipeLds {
	int32_t res[2][RUN * MAXLW];            // [sample][lane]
	int32_t dummy[DEC_BLOCK * (MAXLW + 1)]; // where the lanes without a frame write
	uint32_t kf[2][MAXLW];
	uint32_t cnt[2];
	// DEC_PAIR, per frame of the workgroup: 1 = its samples go straight to the output (at wide sample fobase), else to the planes
	uint32_t fkind[MAXLW], fassign[MAXLW], fbs[MAXLW];
	uint64_t fobase[MAXLW];
	uint32_t order[MAXLW];                  // predictor order; 0 where there is no predictor
	uint32_t wasted[MAXLW];
	int32_t shift[MAXLW];
	uint32_t flags[MAXLW];                  // bit 0: a predictor; bit 1: its sum needs 64 bits; bit 2: samples of up to 24 bits
};
struct NoPipe {};

// The wave-level input pump.  Up to four refills are in flight (one word per thread each); they are
// put into the rings at the next service point, a block of samples later.  (Four, because two cannot feed
// sixteen lanes of 24-bit frames, 7.5 words per lane and block: 2.94 -> 1.96 ms for 14 063 such frames.)
#define PUMP_SLOTS 4
struct Pump {
	uint32_t data[PUMP_SLOTS];
	int lane[PUMP_SLOTS];      // whose ring the word goes to, -1 = slot free
	uint32_t half[PUMP_SLOTS]; // which chunk of that ring
};

// what a lane knows about its current subframe once the subframe header is parsed
struct SubHdr {
	uint32_t kind;      // 0 constant, 1 verbatim, 2 fixed, 3 lpc
	uint32_t order, wasted, sbps, prec;
	uint32_t po, plen, pesc, psize;
	int shift;
	bool lane_wide;
	int32_t constant;
};

__device__ __forceinline__ uint32_t wave_value(uint32_t v, int lane_id) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane_id); }

// Service point (every lane of the wave takes part): land the refills in flight, then start refills for up to four
// lanes whose oldest chunk is read up.  (With two chunks per ring a lane can ask only once it is inside the newest
// one, and the lanes of a wave, which start together and read at similar rates, all ask at about the same time: the
// last ones served are then nearly dry.  With four chunks there are two chunks of slack.)
template <typename LDS>
__device__ __forceinline__ void pump_service(Pump &pm, BitRd &r, LDS &L, uint32_t tid, bool live, uint64_t nwords)
{
	constexpr int NSL = PUMP_SLOTS;
#pragma unroll
	for (int sl = 0; sl < NSL; sl++) {
		if (pm.lane[sl] >= 0) {
			L.ring[pm.lane[sl]][pm.half[sl] * RING_CHUNK + tid] = __builtin_bswap32(pm.data[sl]);
			if ((int)tid == pm.lane[sl])
				r.limit += RING_CHUNK;
			pm.lane[sl] = -1;
		}
	}
	// a lane wants the next chunk once its read position is inside the newest one
	const bool want = live && !r.err && r.wrel + (LDS::ring_words - RING_CHUNK) >= r.limit && r.limit < r.nrel;
	uint64_t m = __ballot(want);
#pragma unroll
	for (int sl = 0; sl < NSL; sl++) {
		if (m) {
			const int ln = __ffsll((unsigned long long)m) - 1;
			m &= m - 1;
			const uint32_t lim = wave_value(r.limit, ln);
			const uint64_t wb = ((uint64_t)wave_value((uint32_t)(r.wbase >> 32), ln) << 32) | wave_value((uint32_t)r.wbase, ln);
			const uint64_t at = wb + lim + tid;
			pm.data[sl] = at < nwords ? (r.wg - r.wbase)[at] : 0u;
			pm.lane[sl] = ln;
			pm.half[sl] = (lim / RING_CHUNK) & (LDS::ring_words / RING_CHUNK - 1);
		}
	}
}

// The samples of one subframe for every lane of the wave: verbatim words, or Rice-coded residuals
// (stream_decoder.c:2692,2715) turned into samples on the spot by the predictor recurrence
// (FLAC__fixed_restore_signal, fixed.c:395; FLAC__lpc_restore_signal[_wide], lpc.c:795,1061; a
// fixed predictor is the LPC filter {1}, {2,-1}, {3,-3,1}, {4,-6,4,-1} with shift 0 in wrapping
// 32-bit arithmetic).  M bounds the predictor orders present in the wave.
//
// The common case runs without control flow: every lane cuts a Rice symbol out of the next 32 bits
// (one count of leading zeros, a saturating add, one bit-field extract), predicts, shifts its history
// and, if the symbol ended in the next word, steps on with the ring word requested at the start of the
// sample.  Lanes for which that is not the right
// thing at this sample (warm-up and verbatim samples, a partition boundary, an escaped partition,
// a symbol longer than 32 bits, a ring that has fallen behind) throw the result away and take
// the one divergent region, which uses the general bit reader.  Because the prediction and the
// history shift run for all lanes on every sample, the history is renamed across the unrolled
// samples rather than moved.
//
// Samples, shifted back up by the wasted bits (stream_decoder.c:2511-2515), are collected a block
// at a time in LDS and written to plane[sample * LW + lane] by all threads together.
// SKIP: the subframe is only walked through, to find where it ends: symbols are measured, not turned into
// values, nothing is predicted or stored (M is 0).
// PX = PipeLds: the values go to the restoring wave as they are (M is 0), a run at a time, instead of to the planes.
template <int M, bool WIDE, bool MUL24, bool SKIP = false, typename LDS, typename PX = NoPipe>
__device__ __forceinline__ void decode_samples(BitRd &r, Pump &pm, const SubHdr &h, LDS &L, int32_t *plane, uint32_t LW, uint32_t tid, uint32_t bs_w,
                                               uint32_t bsf, bool live, uint64_t nwords, uint32_t run_len, PX *X = nullptr, uint8_t *pinfo = nullptr)
{
	constexpr bool PIPE = !std::is_same<PX, NoPipe>::value;
	uint32_t pidx = 0;   // partition parameters read so far (they are noted in pinfo[], if the caller wants them)
	uint32_t hand = 0;   // PIPE: which of the two exchange buffers is being filled
	// the coefficients stay zero through the warm-up samples, which are stored as they are, and
	// are switched on at the first residual; lanes without a predictor keep zeros
	int32_t q[M > 0 ? M : 1], hist[M > 0 ? M : 1];
	const uint32_t lane = tid < (uint32_t)LDS::maxlw ? tid : 0;
#pragma unroll
	for (int j = 0; j < M; j++) {
		q[j] = 0;
		hist[j] = 0;
	}
	uint32_t next_part = h.order;  // sample index where the next partition parameter is read
	uint32_t part_end = 0;         // end of the partition whose parameter was read last
	uint32_t k = 0, kp1 = 1, raw_bits = 0;
	bool escape = false;
	// samples below seg_end are plain Rice symbols with parameter k; fast_end is seg_end while
	// the ring is sure to hold every word the rest of the block can ask for, else 0
	uint32_t seg_end = 0, fast_end = 0;
	// samples below work_end need parsing; a constant subframe needs none
	uint32_t work_end = (live && !r.err && h.kind != 0) ? bsf : 0;
	// sample s of the run in hand goes to slot[s * sstride]; lanes that carry no frame all write to one spare word
	// (PIPE: the stride is the instance's, so that a block's stores differ by constants; lanes without a frame write
	// block after block over the same spare words)
	int32_t *slot;
	if constexpr (PIPE)
		slot = tid < LW ? &X->res[0][tid] : &X->dummy[0];
	else
		slot = tid < LW ? &L.out[tid] : &L.out[LDS::run * LDS::maxlw];
	const uint32_t sstride = PIPE ? (uint32_t)LDS::maxlw : tid < LW ? LW : 0u;
	const uint32_t bstride = tid < LW ? sstride : 0u;   // from block to block of a run
	const uint32_t up = PIPE ? 0u : h.wasted;   // (the restoring wave shifts the wasted bits back in)
	// the partition parameter(s) due at sample i (stream_decoder.c:2741-2767); the predictor is switched on with the first
	auto read_part = [&](uint32_t i) __attribute__((always_inline)) {
		if (M > 0 && i == h.order) {
#pragma unroll
			for (int j = 0; j < M; j++)
				q[j] = (uint32_t)j < h.order ? L.q[j][lane] : 0;
		}
		// a first partition as long as the warm-up holds no residual but still has its
		// parameter in the stream (stream_decoder.c:2734 lets it through, :2747-2767
		// reads it): its successor's parameter follows at once
		do {
			k = br_get(r, h.plen);
			escape = k >= h.pesc;
			if (escape) {
				raw_bits = br_get(r, 5);
				k = 0;
			}
			if (!SKIP && pinfo) {
				if (pidx < FA_INFO_PARAMS)
					pinfo[pidx] = (uint8_t)(escape ? 0x80u | raw_bits : k);
				pidx++;
			}
			part_end = h.po > 0 ? part_end + h.psize : bsf;
		} while (part_end <= i && !r.err && br_pos(r) <= r.bend);
		kp1 = k + 1;
		next_part = part_end;
		seg_end = escape ? 0 : next_part;
	};
	// one sample through the predictor: the value, which also becomes the newest history entry
	auto restore = [&](int32_t res) __attribute__((always_inline)) -> int32_t {
		int32_t v = res;
		if (M > 0) {
			int32_t pred;
			if (WIDE) {
				int64_t sum = 0;
#pragma unroll
				for (int j = 0; j < M; j++)
					sum += (int64_t)q[j] * (int64_t)hist[j];
				pred = h.lane_wide ? (int32_t)(sum >> h.shift) : ((int32_t)sum >> h.shift);
			} else {
				int32_t sum = 0;
				if (MUL24) {
					// two chains of 24-bit multiply-adds, one instruction per tap
					int32_t s0, s1 = 0;
					asm("v_mul_i32_i24 %0, %1, %2" : "=v"(s0) : "v"(q[0]), "v"(hist[0]));
					if (M > 1)
						asm("v_mul_i32_i24 %0, %1, %2" : "=v"(s1) : "v"(q[1]), "v"(hist[1]));
#pragma unroll
					for (int j = 2; j < M; j += 2) {
						asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(s0) : "v"(q[j]), "v"(hist[j]), "v"(s0));
						if (j + 1 < M)
							asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(s1) : "v"(q[j + 1]), "v"(hist[j + 1]), "v"(s1));
					}
					sum = s0 + s1;
				} else {
#pragma unroll
					for (int j = 0; j < M; j++)
						sum += q[j] * hist[j];
				}
				pred = sum >> h.shift;
			}
			v = (int32_t)((uint32_t)res + (uint32_t)pred);
#pragma unroll
			for (int j = M - 1; j > 0; j--)
				hist[j] = hist[j - 1];
			hist[0] = v;
		}
		return v;
	};
	for (uint32_t i0 = 0; i0 < bs_w;) {
		pump_service(pm, r, L, tid, live, nwords);
		// A partition that starts here (partitions of 16 k samples always start with a block) has its parameter read
		// now, so that the block itself is nothing but symbols.
		if (i0 < work_end && h.kind >= 2 && i0 == next_part) {
			read_part(i0);
			if (r.err) {
				work_end = 0;
				seg_end = 0;
			}
		}
		// ---- plain blocks ----
		// Most blocks are sixteen Rice symbols of one parameter for every lane that is still reading, each symbol inside
		// the next 32 bits, all words in the ring.  Such a block runs without a single branch or comparison against the
		// sample index: count of leading zeros, saturating add, the window stepped on by at most one word (the word after
		// next and the one after that are held in registers, the second requested a symbol before it can be needed), and
		// the largest symbol length seen is looked at once, afterwards.  Should a symbol have been longer than 32 bits (or
		// the stream have run out into the zeros behind it), nothing is kept and the samples are done again the careful
		// way.  Lanes with nothing to read (constant subframes, finished or broken frames, lanes without a frame) compute
		// garbage alongside and are put right at the end.  What surrounds a block - the input pump, these tests, handing
		// the samples on - costs as much as ten of its samples, so where every lane has a whole run (LDS::run) of plain symbols ahead
		// the blocks of the run go through back to back.
		const bool idle = i0 >= work_end;
		uint32_t nplain = 0;
		const uint32_t DEC_RUN = run_len;   // at most LDS::run
		{
			const uint32_t lim = seg_end < work_end ? seg_end : work_end;
			const bool run_ok = i0 + DEC_RUN <= lim && r.wrel + DEC_RUN + 2 < r.limit;
			const bool blk_ok = i0 + DEC_BLOCK <= lim && r.wrel + 2 * DEC_BLOCK < r.limit;
			if (i0 + DEC_RUN <= bs_w && __all(idle || run_ok))
				nplain = DEC_RUN;
			else if (__all(idle || blk_ok))
				nplain = DEC_BLOCK;
		}
		bool block_done = false;
		if (nplain) {
			uint32_t cur = r.cur, nxt = r.nxt, rem = r.rem, wrel = r.wrel;
			uint32_t n2 = r.ring[wrel & r.rmask], n3 = r.ring[(wrel + 1) & r.rmask];
			uint32_t tmax = 0;
			// where the word after n3 lies: a byte offset into the lane's ring that is let run on without wrapping (the
			// words taken are counted off it afterwards), wrapped and turned into an LDS address by one and-or
			constexpr uint32_t ring_bytes = LDS::ring_words * 4;
			const uint32_t ring_lds = (uint32_t)(uintptr_t)(LdsWord *)r.ring;
			const uint32_t at0 = (wrel + 1) * 4;
			uint32_t at = at0;
			// the history as it is now, in case the samples have to be done again: kept in the warm-up samples' place in
			// LDS, which nobody reads any more once a partition parameter has been read
			if (M > 0 && tid < LW) {
#pragma unroll
				for (int j = 0; j < M; j++)
					L.w[j][lane] = hist[j];
			}
#pragma nounroll
			for (uint32_t b0 = 0; b0 < nplain; b0 += DEC_BLOCK) {
				int32_t *const bslot = slot + b0 * bstride;
#pragma unroll
				for (uint32_t s = 0; s < DEC_BLOCK; s++) {
					const uint32_t win = __builtin_amdgcn_alignbit(cur, nxt, rem);
					uint32_t z, total;
					asm("v_ffbh_u32 %0, %1" : "=v"(z) : "v"(win));
					asm("v_add_u32_e64 %0, %1, %2 clamp" : "=v"(total) : "v"(z), "v"(kp1));
					tmax = total > tmax ? total : tmax;
					// the borrow of rem - total says whether the symbol ends in the next word; the selects take it as it comes
					// out of the subtraction (the compiler spends a comparison and two more instructions on the same)
					uint32_t left, step;
					uint64_t cross;
					asm("v_sub_co_u32_e64 %0, %1, %2, %3" : "=v"(left), "=s"(cross) : "v"(rem), "v"(total));
					asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(cur) : "v"(cur), "v"(nxt), "s"(cross));
					asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(nxt) : "v"(nxt), "v"(n2), "s"(cross));
					asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(n2) : "v"(n2), "v"(n3), "s"(cross));
					asm("v_cndmask_b32_e64 %0, 0, 4, %1" : "=v"(step) : "s"(cross));
					rem = left & 31u;
					at += step;
					n3 = *(LdsWord *)(uintptr_t)((at & (ring_bytes - 1)) | ring_lds);
					if (PIPE)
						bslot[s * sstride] = (int32_t)win;   // (the restoring wave cuts the symbol out)
					else if (!SKIP) {
						const uint32_t lsbs = __builtin_amdgcn_ubfe(win, 32u - total, k);
						const uint32_t uf = (z << k) | lsbs;
						const int32_t res = (int32_t)(uf >> 1) ^ __builtin_amdgcn_sbfe((int32_t)uf, 0, 1);
						bslot[s * sstride] = (int32_t)((uint32_t)restore(res) << up);
					}
				}
			}
			wrel += (at - at0) >> 2;
			if (!__any(!idle && tmax > 32)) {
				if (!idle) {
					r.cur = cur;
					r.nxt = nxt;
					r.rem = rem;
					r.wrel = wrel;
				}
				if constexpr (PIPE) {
					if (tid < LW)
						X->kf[hand][lane] = idle ? PIPE_RAW : k;
				}
				if (!SKIP && __any(idle && tid < LW)) {
					if (idle) {
						for (uint32_t s = 0; s < nplain; s++)
							slot[s * bstride] = (int32_t)((uint32_t)h.constant << up);
					}
				}
				block_done = true;
			} else {
				nplain = 0;
				if (M > 0 && tid < LW) {
#pragma unroll
					for (int j = 0; j < M; j++)
						hist[j] = L.w[j][lane];
				}
			}
		}
		uint32_t nout = nplain;
		if (!block_done) {
		nout = DEC_BLOCK;
		if constexpr (PIPE) {
			if (tid < LW)
				X->kf[hand][lane] = PIPE_RAW;
		}
		// (a stream that ends early feeds zeros, which no plain block gets through: the end is looked for here)
		if (br_pos(r) > r.bend) {
			r.err = true;
			work_end = 0;
			seg_end = 0;
		}
		// the fast path takes at most one word per sample
		fast_end = r.wrel + 2 * DEC_BLOCK < r.limit ? seg_end : 0;
		// SKIP: a sample is half a dozen instructions, fewer cycles than an LDS read takes, so the ring is read two
		// words ahead of the pair in use: n2 and n3 are the words after nxt, n3 requested a step before it can be
		// needed (garbage while the ring has not got that far, which is when fast_end is 0 and nothing uses them)
		uint32_t n2 = 0, n3 = 0;
		if (SKIP) {
			n2 = r.ring[r.wrel & r.rmask];
			n3 = r.ring[(r.wrel + 1) & r.rmask];
		}
		// unrolled in full: the history needs no moves at all, and the compiler schedules across samples
		// (unrolled by four: 1.70 ms for the benchmark's frames, in full: 1.63)
#pragma unroll
		for (uint32_t s = 0; s < DEC_BLOCK; s++) {
			const uint32_t i = i0 + s;
			// the word a step into the next word at the end of this sample would bring in
			const uint32_t ring_word = SKIP ? n2 : r.ring[r.wrel & r.rmask];
			// FLAC__bitreader_read_rice_signed_block (bitreader.c:775), symbol inside the next 32 bits: count of
			// leading zeros (all ones for "none of the 32 is set": the saturating add then fails the length test),
			// stop bit, k low bits
			const uint32_t win = br_peek(r);
			uint32_t z, total;
			asm("v_ffbh_u32 %0, %1" : "=v"(z) : "v"(win));
			asm("v_add_u32_e64 %0, %1, %2 clamp" : "=v"(total) : "v"(z), "v"(kp1));
			const bool fast = i < fast_end && total <= 32;
			const uint32_t lsbs = __builtin_amdgcn_ubfe(win, 32u - total, k);
			const uint32_t uf = (z << k) | lsbs;
			int32_t res = fast ? (int32_t)((uf >> 1) ^ (0u - (uf & 1))) : h.constant;
			const uint32_t used = fast ? total : 0;
			{
				// at most one word boundary is crossed (used <= 32)
				const int32_t left = (int32_t)r.rem - (int32_t)used;
				const bool cross = left < 0;
				r.cur = cross ? r.nxt : r.cur;
				r.nxt = cross ? ring_word : r.nxt;
				r.wrel += cross ? 1u : 0u;
				r.rem = (uint32_t)left & 31u;
				if (SKIP) {
					n2 = cross ? n3 : n2;
					n3 = r.ring[(r.wrel + 1) & r.rmask];
				}
			}
			if (!fast && i < work_end) {
				if (h.kind >= 2) {
					if (i < h.order)
						res = L.w[i][lane];
					else {
						if (i == next_part)
							read_part(i);
						if (!escape) {
							const uint32_t msbs = br_unary(r);
							const uint32_t low = br_get(r, k);
							const uint32_t u = (msbs << k) | low;
							res = (int32_t)(u >> 1) ^ -(int32_t)(u & 1);
						} else
							res = br_get_signed(r, raw_bits);
					}
				} else
					res = br_get_signed(r, h.sbps);
				if (r.err) {
					work_end = 0;
					seg_end = 0;
				}
				fast_end = r.wrel + 2 * DEC_BLOCK < r.limit ? seg_end : 0;
				if (SKIP) {
					n2 = r.ring[r.wrel & r.rmask];
					n3 = r.ring[(r.wrel + 1) & r.rmask];
				}
			}
			const int32_t v = restore(res);
			if (!SKIP)
				slot[s * sstride] = (int32_t)((uint32_t)v << up);
		}
		}
		if constexpr (PIPE) {
			// hand the run over: the restoring wave works on it while the next one is read into the other buffer
			if (tid == 0)
				X->cnt[hand] = i0 + nout < bs_w ? nout : bs_w - i0;
			__syncthreads();
			hand ^= 1u;
			slot = tid < LW ? &X->res[hand][tid] : &X->dummy[0];
		} else if (!SKIP) {
			// the samples go out with all 64 threads along memory order
			const uint32_t nblk = (i0 + nout < bs_w ? nout : bs_w - i0) * LW;
			for (uint32_t e = tid; e < nblk; e += DEC_LANES)
				plane[(size_t)i0 * LW + e] = L.out[e];
		}
		i0 += nout;
	}
}

// Where the restoring wave's samples go.
#define PAIR_FRAMES 8   // frames per workgroup of DEC_PAIR at most (half the columns)
struct PipeOut {
	int32_t *plane;     // DEC_PIPE: the channel's plane of the group; DEC_PAIR: channel 0's
	int32_t *plane1;    // DEC_PAIR: channel 1's plane
	uint32_t sb;        // DEC_PAIR: bytes per sample of the caller's interleaved PCM (2 or 4)
	// DEC_PAIR, per frame of the workgroup, the same in every lane (scalar registers): where its first stereo pair goes
	// (null: to the planes), its length, its channel assignment and the wasted bits of its two subframes
	uint8_t *dst[PAIR_FRAMES];
	uint32_t bs[PAIR_FRAMES], assign[PAIR_FRAMES], wasted0[PAIR_FRAMES], wasted1[PAIR_FRAMES];
};

// A run of n samples from L.out ([sample][column], row stride MAXLW + 1, as restored: the wasted bits are shifted back in
// here, stream_decoder.c:2511-2515) on its way out.  DEC_PIPE: to the channel's plane, along memory order.  DEC_PAIR: the
// columns are FPW frames' first channel and then the same frames' second channel, so both samples of a stereo frame are at
// hand: the decorrelation is undone (stream_decoder.c:2067-2102) and the pair goes straight to its place in the caller's
// interleaved PCM, a lane per sample, 256 contiguous bytes per frame and run - unless the frame's CRC-16 has not been seen
// to be right, in which case it goes to the planes like everything did before and waits for k_dec_output.
template <bool PAIR, typename LDS, typename PX>
__device__ __forceinline__ void pipe_flush(LDS &L, PX &X, const PipeOut &O, uint32_t LW, uint32_t tid, uint32_t i0, uint32_t n)
{
	constexpr uint32_t ostride = LDS::maxlw + 1;
	if constexpr (!PAIR) {
		const uint32_t col = tid & (LDS::maxlw - 1);
		static_assert(DEC_LANES % LDS::maxlw == 0, "a thread stays with one column");
		if (col < LW) {
			const uint32_t col_wasted = X.wasted[col];
			// thread t takes samples t / 16, t / 16 + 4, ... of its column; a run is 16, 32 or 64 samples but for a block's
			// last one, and goes out in chunks of those sizes with every load requested before the first store
			constexpr uint32_t per = DEC_LANES / LDS::maxlw;   // samples a step of the wave covers
			const uint32_t s0 = tid / LDS::maxlw;
			const int32_t *src = &L.out[s0 * ostride + col];
			int32_t *dst = O.plane + (size_t)(i0 + s0) * LW + col;
			const size_t dstep = (size_t)per * LW;
			auto chunk = [&](auto count) __attribute__((always_inline)) {
				constexpr int N = decltype(count)::value;
				int32_t v[N];
#pragma unroll
				for (int u = 0; u < N; u++)
					v[u] = src[u * per * ostride];
#pragma unroll
				for (int u = 0; u < N; u++)
					dst[u * dstep] = (int32_t)((uint32_t)v[u] << col_wasted);
				src += N * per * ostride;
				dst += N * dstep;
			};
			uint32_t base = 0;
			for (; base + 16 * per <= n; base += 16 * per)
				chunk(std::integral_constant<int, 16>());
			if (base + 8 * per <= n) {
				chunk(std::integral_constant<int, 8>());
				base += 8 * per;
			}
			if (base + 4 * per <= n) {
				chunk(std::integral_constant<int, 4>());
				base += 4 * per;
			}
			for (; base + s0 < n; base += per) {
				*dst = (int32_t)((uint32_t)*src << col_wasted);
				src += per * ostride;
				dst += dstep;
			}
		}
	} else {
		const uint32_t FPW = LW >> 1;
		const uint32_t i = i0 + tid;
#pragma unroll
		for (uint32_t k = 0; k < PAIR_FRAMES; k++) {
			if (k < FPW && tid < n && i < O.bs[k]) {
				int32_t a = (int32_t)((uint32_t)L.out[tid * ostride + k] << O.wasted0[k]);
				int32_t b = (int32_t)((uint32_t)L.out[tid * ostride + k + FPW] << O.wasted1[k]);
				if (O.dst[k]) {
					if (O.assign[k] == 1)
						b = a - b;
					else if (O.assign[k] == 2)
						a = a + b;
					else if (O.assign[k] == 3) {
						int32_t mid = (int32_t)((uint32_t)a << 1);
						mid |= (b & 1);
						a = (mid + b) >> 1;
						b = (mid - b) >> 1;
					}
					if (O.sb == 2)
						((uint32_t *)O.dst[k])[i] = ((uint32_t)a & 0xffffu) | ((uint32_t)b << 16);
					else
						((int2 *)O.dst[k])[i] = make_int2(a, b);
				} else {
					O.plane[(size_t)i * FPW + k] = a;
					O.plane1[(size_t)i * FPW + k] = b;
				}
			}
		}
	}
}

// N x 64 of the windows the reading wave handed over, cut into residuals in place: all N requested before the first is
// used (an LDS round trip costs a lone wave as much as twenty instructions).  A thread that looks after a raw column
// writes back what it read, so that no thread is ever masked out.
template <int N>
__device__ __forceinline__ void pipe_cut_chunk(int32_t *col, uint32_t k, bool mine)
{
	uint32_t win[N];
#pragma unroll
	for (int u = 0; u < N; u++)
		win[u] = (uint32_t)col[u * DEC_LANES];
#pragma unroll
	for (int u = 0; u < N; u++) {
		const uint32_t z = (uint32_t)__clz((int)win[u]);
		const uint32_t lsbs = __builtin_amdgcn_ubfe(win[u], 31u - z - k, k);
		const uint32_t uf = (z << k) | lsbs;
		const int32_t v = (int32_t)(uf >> 1) ^ -(int32_t)(uf & 1u);
		col[u * DEC_LANES] = mine ? v : (int32_t)win[u];
	}
}

// Rice symbols that were handed over as they lie in the stream: thread t looks after the words of lane t mod 16,
// four samples apart (bitreader.c:775: zeros up to the stop bit, then the parameter's worth of low bits; zigzag).
// A run is 16, 32 or 64 samples but for a block's last one: 256, 512 or 1024 words, taken in chunks of those sizes.
template <typename LDS, typename PX>
__device__ __forceinline__ void pipe_cut_symbols(PX &X, uint32_t hand, uint32_t n, uint32_t tid)
{
	constexpr uint32_t xstride = LDS::maxlw;
	static_assert(DEC_LANES % xstride == 0, "a thread stays with one lane's words");
	const uint32_t kf = X.kf[hand][tid & (xstride - 1)];
	if (!__all(kf & PIPE_RAW)) {
		const uint32_t k = kf & 31u;
		const bool mine = !(kf & PIPE_RAW);
		const uint32_t lim = n * xstride;
		int32_t *col = &X.res[hand][tid];
		uint32_t base = 0;
		for (; base + 16 * DEC_LANES <= lim; base += 16 * DEC_LANES)
			pipe_cut_chunk<16>(col + base, k, mine);
		if (base + 8 * DEC_LANES <= lim) {
			pipe_cut_chunk<8>(col + base, k, mine);
			base += 8 * DEC_LANES;
		}
		if (base + 4 * DEC_LANES <= lim) {
			pipe_cut_chunk<4>(col + base, k, mine);
			base += 4 * DEC_LANES;
		}
		// what is left of a block's last run: fewer than sixteen samples
		for (; base + tid < lim; base += DEC_LANES) {
			const uint32_t win = (uint32_t)col[base];
			const uint32_t z = (uint32_t)__clz((int)win);
			const uint32_t lsbs = __builtin_amdgcn_ubfe(win, 31u - z - k, k);
			const uint32_t uf = (z << k) | lsbs;
			if (mine)
				col[base] = (int32_t)(uf >> 1) ^ -(int32_t)(uf & 1u);
		}
	}
}

// The restoring wave where no lane's sum needs 64 bits: four threads to a subframe, thread t of the four holding taps
// T t .. T t + T - 1 and the samples they multiply.  A sample costs T multiply-adds, two adds across the four (after
// which each has the whole sum; 32-bit sums wrap the same in any order), shift, add, and one move that hands every
// thread's oldest sample to the next thread while the first keeps the new one: ten instructions at order 12 where one
// thread per subframe issues twenty, and a lone wave issues a dependent instruction only every nine cycles.
// The four sit in the four banks of a row of sixteen lanes (lane = 16 r + 4 t + i for subframe 4 r + i), so that
// "the next thread" is a shift by four lanes and "not the first" is a bank mask.  The samples stay as they are in LDS
// ([sample][lane], the instance's stride); the wasted bits are shifted back in on the way out.
template <int T, bool MUL24, bool PAIR, typename LDS, typename PX>
__device__ __forceinline__ void restore_run_quad(LDS &L, PX &X, const PipeOut &O, uint32_t LW, uint32_t tid, uint32_t bs_w, uint32_t need)
{
	static_assert(LDS::maxlw == 16 && DEC_LANES == 64, "four threads per subframe, sixteen subframes");
	constexpr uint32_t xstride = LDS::maxlw, ostride = LDS::maxlw + 1;
	const uint32_t f = ((tid >> 4) << 2) | (tid & 3u), t = (tid >> 2) & 3u;
	const bool have = f < LW;
	const uint32_t fl = have ? f : 0;
	const bool predictive = have && (X.flags[fl] & 1u);
	const uint32_t order = predictive ? X.order[fl] : 0;
	const int shift = have ? X.shift[fl] : 0;
	int32_t q[T], h[T];
#pragma unroll
	for (int u = 0; u < T; u++) {
		q[u] = 0;
		h[u] = 0;
	}
	auto step = [&](int32_t res) __attribute__((always_inline)) -> int32_t {
		int32_t part;
		if (MUL24) {
			asm("v_mul_i32_i24 %0, %1, %2" : "=v"(part) : "v"(q[T - 1]), "v"(h[T - 1]));
#pragma unroll
			for (int u = T - 2; u >= 0; u--)
				asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(part) : "v"(q[u]), "v"(h[u]), "v"(part));
		} else {
			part = q[T - 1] * h[T - 1];
#pragma unroll
			for (int u = T - 2; u >= 0; u--)
				part += q[u] * h[u];
		}
		part += __builtin_amdgcn_update_dpp(0, part, 0x124, 0xf, 0xf, true);   // row_ror:4
		part += __builtin_amdgcn_update_dpp(0, part, 0x128, 0xf, 0xf, true);   // row_ror:8
		const int32_t v = (int32_t)((uint32_t)res + (uint32_t)(part >> shift));
		// row_shr:4, written in banks 1 to 3 only: every thread's oldest sample to the next of the four, the first keeps the new one
		const int32_t in = __builtin_amdgcn_update_dpp(v, h[T - 1], 0x114, 0xf, 0xe, false);
#pragma unroll
		for (int u = T - 1; u > 0; u--)
			h[u] = h[u - 1];
		h[0] = in;
		return v;
	};
	uint32_t hand = 0;
	for (uint32_t i0 = 0; i0 < bs_w;) {
		__syncthreads();
		const uint32_t n = __builtin_amdgcn_readfirstlane(X.cnt[hand]);   // (the same in every lane: loops over it are scalar)
		pipe_cut_symbols<LDS>(X, hand, n, tid);
		const int32_t *src = have ? &X.res[hand][f] : &X.dummy[0];
#pragma nounroll
		for (uint32_t b0 = 0; b0 < n; b0 += DEC_BLOCK) {
			const int32_t *bsrc = src + (have ? b0 * xstride : 0u);
			int32_t *bslot = have ? &L.out[b0 * ostride + f] : &X.dummy[0];
			// (all of the block's values first: a load between the stores would be waited for sample by sample)
			int32_t rv[DEC_BLOCK];
#pragma unroll
			for (uint32_t s = 0; s < DEC_BLOCK; s++)
				rv[s] = bsrc[s * xstride];
			if (i0 + b0 <= need) {
				// a block in which some subframe's warm-up may end
#pragma unroll
				for (uint32_t s = 0; s < DEC_BLOCK; s++) {
					if (predictive && i0 + b0 + s == order) {
#pragma unroll
						for (int u = 0; u < T; u++)
							q[u] = T * t + u < order ? L.q[T * t + u][fl] : 0;
					}
					bslot[s * ostride] = step(rv[s]);
				}
			} else {
#pragma unroll
				for (uint32_t s = 0; s < DEC_BLOCK; s++)
					bslot[s * ostride] = step(rv[s]);
			}
		}
		pipe_flush<PAIR>(L, X, O, LW, tid, i0, n);
		i0 += n;
		hand ^= 1u;
	}
}

// The restoring wave of the pipelined form: FLAC__fixed_restore_signal (fixed.c:395) and FLAC__lpc_restore_signal[_wide]
// (lpc.c:795,1061) over the values the entropy wave hands over, the wasted bits shifted back in
// (stream_decoder.c:2511-2515), the samples collected a run at a time and written to the planes along memory order.
// A lane's predictor is switched on at the sample that follows its warm-up; only the first blocks of a subframe
// (up to the largest order in the wave) have to look for that sample.
template <int M, bool WIDE, bool MUL24, bool PAIR, typename LDS, typename PX>
__device__ __forceinline__ void restore_run(LDS &L, PX &X, const PipeOut &O, uint32_t LW, uint32_t tid, uint32_t bs_w, uint32_t order, bool predictive,
                                            int shift, bool lane_wide, uint32_t need)
{
	int32_t q[M > 0 ? M : 1], hist[M > 0 ? M : 1];
#pragma unroll
	for (int j = 0; j < M; j++) {
		q[j] = 0;
		hist[j] = 0;
	}
	const uint32_t lane = tid < (uint32_t)LDS::maxlw ? tid : 0;
	constexpr uint32_t xstride = LDS::maxlw, ostride = LDS::maxlw + 1;   // of the exchange buffers, of the samples on their way out
	auto step = [&](int32_t res) __attribute__((always_inline)) -> int32_t {
		int32_t v = res;
		if (M > 0) {
			int32_t pred;
			if (WIDE) {
				int64_t sum = 0;
#pragma unroll
				for (int j = 0; j < M; j++)
					sum += (int64_t)q[j] * (int64_t)hist[j];
				pred = lane_wide ? (int32_t)(sum >> shift) : ((int32_t)sum >> shift);
			} else {
				int32_t sum = 0;
				if (MUL24) {
					int32_t s0, s1 = 0;
					asm("v_mul_i32_i24 %0, %1, %2" : "=v"(s0) : "v"(q[0]), "v"(hist[0]));
					if (M > 1)
						asm("v_mul_i32_i24 %0, %1, %2" : "=v"(s1) : "v"(q[1]), "v"(hist[1]));
#pragma unroll
					for (int j = 2; j < M; j += 2) {
						asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(s0) : "v"(q[j]), "v"(hist[j]), "v"(s0));
						if (j + 1 < M)
							asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(s1) : "v"(q[j + 1]), "v"(hist[j + 1]), "v"(s1));
					}
					sum = s0 + s1;
				} else {
#pragma unroll
					for (int j = 0; j < M; j++)
						sum += q[j] * hist[j];
				}
				pred = sum >> shift;
			}
			v = (int32_t)((uint32_t)res + (uint32_t)pred);
#pragma unroll
			for (int j = M - 1; j > 0; j--)
				hist[j] = hist[j - 1];
			hist[0] = v;
		}
		return v;
	};
	uint32_t hand = 0;
	for (uint32_t i0 = 0; i0 < bs_w;) {
		__syncthreads();
		const uint32_t n = __builtin_amdgcn_readfirstlane(X.cnt[hand]);   // (the same in every lane: loops over it are scalar)
		pipe_cut_symbols<LDS>(X, hand, n, tid);
		const int32_t *src = tid < LW ? &X.res[hand][tid] : &X.dummy[0];
#pragma nounroll
		for (uint32_t b0 = 0; b0 < n; b0 += DEC_BLOCK) {
			const int32_t *bsrc = src + (tid < LW ? b0 * xstride : 0u);
			int32_t *bslot = tid < LW ? &L.out[b0 * ostride + tid] : &X.dummy[0];
			// (all of the block's values first: a load between the stores would be waited for sample by sample)
			int32_t rv[DEC_BLOCK];
#pragma unroll
			for (uint32_t s = 0; s < DEC_BLOCK; s++)
				rv[s] = bsrc[s * xstride];
			if (M > 0 && i0 + b0 <= need) {
				// a block in which some lane's warm-up may end
#pragma unroll
				for (uint32_t s = 0; s < DEC_BLOCK; s++) {
					if (predictive && i0 + b0 + s == order) {
#pragma unroll
						for (int j = 0; j < M; j++)
							q[j] = (uint32_t)j < order ? L.q[j][lane] : 0;
					}
					bslot[s * ostride] = step(rv[s]);
				}
			} else {
#pragma unroll
				for (uint32_t s = 0; s < DEC_BLOCK; s++)
					bslot[s * ostride] = step(rv[s]);
			}
		}
		pipe_flush<PAIR>(L, X, O, LW, tid, i0, n);
		i0 += n;
		hand ^= 1u;
	}
}

// One lane per subframe; a wave carries one channel of LW <= MAXLW frames in its first lanes, the other lanes
// only help moving data.  A wave's pace is set by its longest dependent chain, not by how many of its lanes
// decode, so a batch is spread over as many waves as the chip can run at once.
//
// The chain is a frame's samples one after the other, and a subframe can only be started where the one
// before it ends.  DEC_SUB decodes the subframe that starts at a known bit position (sub_start, or the end of
// the frame header for channel 0 of a chained pass); who knows the positions depends on the batch:
//   chained (chain = c): one launch per channel, in channel order.  Launch c decodes channel c of every frame
//     and leaves where channel c + 1 starts; nothing is walked twice.  For batches that fill the chip anyway.
//   split (chain = DEC_NO_CHAIN): DEC_SKIP first walks the subframes of every channel but the last without
//     decoding them (a symbol's length is all it takes: less than half the instructions) and leaves every
//     start; then one DEC_SUB launch runs all (frame, channel) pairs at once, so the longest chain is one
//     subframe.  For batches that leave the chip room.
// A frame found malformed before its last subframe gets its status where that is found and is left alone
// afterwards; the lane of the last channel finishes the frame (padding, end, status).
// (A kernel that loops over the channels with a lane per frame is the obvious third form and was the first one
// here; with two bodies per sample block the compiler needs more than 512 registers for it, and one launch per
// channel is that loop without its cost.)
#define DEC_NO_CHAIN 0xffffffffu
// DEC_PIPE is DEC_SUB with two waves to a workgroup: the first reads the symbols, the second turns them into samples and
// writes them out, a run behind.  Reading is the longer half, so the chain per subframe shrinks to that; the chip has
// to have room for twice the waves, which is what the split form is used for anyway.
// DEC_PAIR is DEC_PIPE for stereo with both channels of a frame in one workgroup: the columns are LW / 2 frames' first
// subframes and then the same frames' second ones, so the restoring wave has both samples of every stereo pair and writes
// final interleaved PCM itself (pipe_flush): no planes, no output kernel for those frames.
#define DEC_SKIP 1
#define DEC_SUB 2
#define DEC_PIPE 3
#define DEC_PAIR 4
#define DEC_NO_START 0xffffffffffffffffull

// DEEP: four chunks per ring instead of two (and room for runs of four blocks): half as much LDS again per wave, so only
// for launches whose waves all find room on the chip even so (the launcher decides)
template <int MODE, int MAXLW, bool DEEP>
// LW: columns (subframes) per wave.  ahead, pcm_out: DEC_PAIR only (what the CRC pass that ran ahead found; the caller's PCM).
__global__ __launch_bounds__((MODE == DEC_PIPE || MODE == DEC_PAIR) ? 2 * DEC_LANES : DEC_LANES, 2) void k_dec_frames(FaDecParams P, const uint32_t *flac_words, const FaStream *streams,
                                                          FaFrameCand *cands, const uint32_t *ncand_ptr, int32_t *tmp, uint32_t LW, uint64_t *sub_start,
                                                          uint32_t chain, const uint64_t *ahead, void *pcm_out)
{
	__shared__ DecLds<MAXLW, (DEEP ? RING_WORDS_BIG : RING_WORDS_MIN)> L;
	constexpr bool PAIR = MODE == DEC_PAIR;
	constexpr bool PIPE = MODE == DEC_PIPE || PAIR;
	constexpr bool SUBM = MODE == DEC_SUB || PIPE;   // one subframe per lane, from a known start
	using Xchg = PipeLds<MAXLW, (DEEP ? DEC_RUN_BIG : DEC_RUN_MIN)>;
	__shared__ typename std::conditional<PIPE, Xchg, NoPipe>::type X;
	const uint32_t run_len = umin_u32(P.run, (uint32_t)(DEEP ? DEC_RUN_BIG : DEC_RUN_MIN));
	const uint32_t tid = threadIdx.x & (DEC_LANES - 1);
	const uint32_t ncand = ncand_ptr[0];
	const bool chained = SUBM && !PAIR && chain != DEC_NO_CHAIN;
	// frames per workgroup (a power of two), which frame and which channel a column is
	const uint32_t FPW = PAIR ? LW >> 1 : LW;
	const uint32_t grp = (SUBM && !PAIR && !chained) ? blockIdx.x / P.channels : blockIdx.x;
	const uint32_t only_c = PAIR ? (tid < LW && tid >= FPW ? 1u : 0u) : SUBM ? (chained ? chain : blockIdx.x % P.channels) : 0;
	if (grp * FPW >= ncand)
		return;
	const uint32_t j = grp * FPW + (PAIR ? (tid & (FPW - 1)) : tid);
	const bool have_frame = tid < LW && j < ncand;
	bool live = have_frame;
	const uint32_t lane = tid < (uint32_t)MAXLW ? tid : 0;
	int32_t *frame_tmp = tmp + (size_t)grp * P.channels * P.max_blocksize * FPW;
	if constexpr (PIPE) {
		if (threadIdx.x >= DEC_LANES) {
			// ---- the restoring wave ----
			uint32_t bs_w = have_frame ? cands[j].blocksize : 0;
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) {
				const uint32_t t = __shfl_xor(bs_w, o, 64);
				bs_w = t > bs_w ? t : bs_w;
			}
			PipeOut O;
			O.plane = frame_tmp + (PAIR ? 0 : (size_t)only_c * P.max_blocksize * LW);
			O.plane1 = frame_tmp + (size_t)P.max_blocksize * FPW;
			O.sb = P.out_sample_bytes;
			if constexpr (PAIR) {
				// Where each frame's samples go.  Straight to the output if the CRC pass that ran ahead found the CRC-16 right
				// at the end it guessed (the frame is what it seems then, bar one chance in 65 536) and the frame has a place
				// there (what k_dec_output checks); else to the planes.
				if (tid < FPW) {
					const uint32_t jj = grp * FPW + tid;
					uint32_t kind = 0, assign = 0, bs = 0;
					uint64_t obase = 0;
					if (jj < ncand) {
						const FaFrameCand *cf = cands + jj;
						bs = cf->blocksize;
						assign = cf->assignment;
						const uint64_t smp = cf->sample;
						const FaStream st = streams[cf->stream];
						const uint64_t first = st.first_sample == ~0ull ? cands[0].sample : st.first_sample;
						const bool verified = ahead && (ahead[jj] >> 63);
						if (verified && smp >= first) {
							const uint64_t rel = smp - first;
							if (!(st.total_samples && rel + bs > st.total_samples) && st.sample_offset + rel + bs <= P.out_capacity_wide) {
								kind = 1;
								obase = st.sample_offset + rel;
							}
						}
					}
					X.fkind[tid] = kind;
					X.fassign[tid] = assign;
					X.fbs[tid] = bs;
					X.fobase[tid] = obase;
				}
			}
			__syncthreads();   // the subframe headers are in
			if constexpr (PAIR) {
#pragma unroll
				for (uint32_t k = 0; k < PAIR_FRAMES; k++) {
					const bool in = k < FPW;
					const uint32_t kind = in ? X.fkind[k] : 0;
					const uint64_t at = in ? X.fobase[k] : 0;
					const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)at), hi = __builtin_amdgcn_readfirstlane((uint32_t)(at >> 32));
					O.dst[k] = __builtin_amdgcn_readfirstlane(kind) == 1
						? (uint8_t *)pcm_out + (((uint64_t)hi << 32) | lo) * (2u * P.out_sample_bytes) : nullptr;
					O.bs[k] = __builtin_amdgcn_readfirstlane(in ? X.fbs[k] : 0);
					O.assign[k] = __builtin_amdgcn_readfirstlane(in ? X.fassign[k] : 0);
					O.wasted0[k] = __builtin_amdgcn_readfirstlane(in ? X.wasted[k] : 0);
					O.wasted1[k] = __builtin_amdgcn_readfirstlane(in ? X.wasted[k + FPW] : 0);
				}
			}
			const uint32_t flags = tid < LW ? X.flags[lane] : 0;
			const bool predictive = flags & 1u, lane_wide = flags & 2u;
			const uint32_t order = predictive ? X.order[lane] : 0;
			const int shift = tid < LW ? X.shift[lane] : 0;
			uint32_t need = order;
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) {
				const uint32_t t = __shfl_xor(need, o, 64);
				need = t > need ? t : need;
			}
			const bool any_wide = __any(predictive && lane_wide);
			const bool all24 = __all(!predictive || (flags & 4u));
#define RESTORE(M, WIDE) restore_run<M, WIDE, false, PAIR>(L, X, O, LW, tid, bs_w, order, predictive, shift, lane_wide, need)
#define RESTORE_QUAD(T) \
	do { \
		if (all24) restore_run_quad<T, true, PAIR>(L, X, O, LW, tid, bs_w, need); \
		else restore_run_quad<T, false, PAIR>(L, X, O, LW, tid, bs_w, need); \
	} while (0)
			if (need == 0) RESTORE(0, false);
			else if (!any_wide) {
				if (need <= 4) RESTORE_QUAD(1);
				else if (need <= 8) RESTORE_QUAD(2);
				else if (need <= 12) RESTORE_QUAD(3);
				else if (need <= 16) RESTORE_QUAD(4);
				else RESTORE_QUAD(8);
			}
			else if (need <= 4) RESTORE(4, true);
			else if (need <= 8) RESTORE(8, true);
			else if (need <= 12) RESTORE(12, true);
			else if (need <= 16) RESTORE(16, true);
			else RESTORE(32, true);
#undef RESTORE_QUAD
#undef RESTORE
			return;
		}
	}
	FaFrameCand cd;
	if (live)
		cd = cands[j];
	else {
		cd.pos = 0; cd.sample = 0; cd.hdr_bytes = 0; cd.blocksize = 0; cd.assignment = 0; cd.stream = 0; cd.status = 0; cd.end_byte = 0;
	}
	// where the lane starts reading, in bits from the start of the input
	uint64_t start_bit = (cd.pos + cd.hdr_bytes) * 8;
	if (SUBM && !(chained && only_c == 0)) {
		if (live) {
			start_bit = sub_start[(size_t)j * P.channels + only_c];
			if (start_bit >= P.flac_bytes * 8) {   // DEC_NO_START: the frame ended in an earlier subframe
				live = false;
				start_bit = 0;
			}
		} else
			start_bit = 0;
	}
	const uint64_t nwords = (P.flac_bytes + 3) / 4;
	BitRd r;
	r.wbase = SUBM ? start_bit >> 5 : cd.pos >> 2;
	r.wg = flac_words + r.wbase;
	r.nrel = nwords - r.wbase > 0xffffffffull ? 0xffffffffu : (uint32_t)(nwords - r.wbase);
	// positions are 32-bit, relative to the frame's first word; a stream that goes on for more
	// than that is treated as endless
	bool endless = false;
	{
		const uint64_t end_abs = live ? (streams[cd.stream].byte_offset + streams[cd.stream].byte_length) * 8 : 0;
		const uint64_t rel = end_abs > r.wbase * 32 ? end_abs - r.wbase * 32 : 0;
		endless = rel > 0xf0000000ull;
		r.bend = endless ? 0xf0000000u : (uint32_t)rel;
	}
	r.ring = L.ring[lane];
	r.rmask = (DEEP ? RING_WORDS_BIG : RING_WORDS_MIN) - 1;
	// the first two chunks of every frame, all requested before the first is waited for
	{
		uint32_t first[2 * MAXLW];
#pragma unroll
		for (uint32_t ln = 0; ln < (uint32_t)MAXLW; ln++) {
			const uint64_t wb = ((uint64_t)wave_value((uint32_t)(r.wbase >> 32), (int)ln) << 32) | wave_value((uint32_t)r.wbase, (int)ln);
#pragma unroll
			for (uint32_t c = 0; c < 2; c++) {
				const uint64_t at = wb + c * RING_CHUNK + tid;
				first[2 * ln + c] = (ln < LW && at < nwords) ? flac_words[at] : 0u;
			}
		}
#pragma unroll
		for (uint32_t ln = 0; ln < (uint32_t)MAXLW; ln++) {
			L.ring[ln][tid] = __builtin_bswap32(first[2 * ln]);
			L.ring[ln][RING_CHUNK + tid] = __builtin_bswap32(first[2 * ln + 1]);
		}
		r.limit = 2 * RING_CHUNK;
	}
	Pump pm;
	for (int sl = 0; sl < PUMP_SLOTS; sl++) {
		pm.lane[sl] = -1;
		pm.data[sl] = 0;
		pm.half[sl] = 0;
	}
	br_start(r, (uint32_t)(start_bit - r.wbase * 32));
	const uint32_t bsf = cd.blocksize;
	uint32_t bs_w = bsf;
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		const uint32_t t = __shfl_xor(bs_w, o, 64);
		bs_w = t > bs_w ? t : bs_w;
	}
	bool reserved_code = false;  // the parse stopped at a reserved subframe type or residual coding method

	// the walk passes every channel but the last; the other modes take one subframe per lane (in DEC_PAIR: of either channel)
	const uint32_t npass = SUBM ? 1u : P.channels - 1;
	for (uint32_t pass = 0; pass < npass; pass++) {
		const uint32_t c = SUBM ? only_c : pass;
		int32_t *plane = PAIR ? frame_tmp : frame_tmp + (size_t)c * P.max_blocksize * LW;
		if (MODE == DEC_SKIP && live)
			sub_start[(size_t)j * P.channels + c] = r.err ? DEC_NO_START : r.wbase * 32 + br_pos(r);
		pump_service(pm, r, L, tid, live, nwords);
		// ---- subframe header (stream_decoder.c:2450-2518) ----
		SubHdr h;
		h.kind = 0; h.order = 0; h.wasted = 0; h.po = 0; h.plen = 4; h.pesc = 15; h.psize = 0; h.shift = 0; h.lane_wide = false; h.constant = 0; h.prec = 0;
		h.sbps = P.bps;
		if ((cd.assignment == 1 && c == 1) || (cd.assignment == 2 && c == 0) || (cd.assignment == 3 && c == 1))
			h.sbps++;
		const bool ok_in = live && !r.err;
		if (ok_in) {
			uint32_t x = br_get(r, 8);
			const uint32_t has_wasted = x & 1;
			x &= 0xfe;
			if (has_wasted) {
				h.wasted = br_unary(r) + 1;
				if (h.wasted >= h.sbps)
					r.err = true;
				else
					h.sbps -= h.wasted;
			}
			// reserved type codes are "unparseable", a set padding bit is lost sync (stream_decoder.c:2476-2503)
			if (x & 0x80) r.err = true;
			else if (x == 0) h.kind = 0;
			else if (x == 2) h.kind = 1;
			else if (x < 16) r.err = reserved_code = true;
			else if (x <= 24) {
				h.kind = 2;
				h.order = (x >> 1) & 7;
				if (h.order > 4) r.err = true;
			} else if (x < 64) r.err = reserved_code = true;
			else {
				h.kind = 3;
				h.order = ((x >> 1) & 31) + 1;
			}
			if (h.order > bsf)
				r.err = true;
		}
		if (ok_in && !r.err) {
			if (h.kind == 0)
				h.constant = br_get_signed(r, h.sbps);
			if (h.kind >= 2) {
				for (uint32_t i = 0; i < h.order; i++)
					L.w[i][lane] = br_get_signed(r, h.sbps);
				if (h.kind == 3) {
					uint32_t prec = br_get(r, 4);
					if (prec == 15)
						r.err = true;  // stream_decoder.c:2622
					prec++;
					h.prec = prec;
					h.shift = br_get_signed(r, 5);
					if (h.shift < 0)
						h.shift = 0;  // the reference's shift by a negative count is undefined; it never writes one
					for (uint32_t i = 0; i < h.order; i++)
						L.q[i][lane] = br_get_signed(r, prec);
					h.lane_wide = !(h.sbps + prec + d_ilog2(h.order) <= 32);  // stream_decoder.c:2676
				} else {
					for (uint32_t i = 0; i < h.order; i++)
						L.q[i][lane] = c_fixed_q[h.order][i];
				}
				const uint32_t method = br_get(r, 2);
				if (method > 1)
					r.err = reserved_code = true;  // stream_decoder.c:2575,2654
				if (method == 1) {
					h.plen = 5;
					h.pesc = 31;
				}
				h.po = br_get(r, 4);
				// stream_decoder.c:2721-2739
				h.psize = h.po > 0 ? bsf >> h.po : bsf - h.order;
				if (h.po > 0 && (h.psize < h.order || (h.psize << h.po) != bsf))
					r.err = true;
			}
			if (br_pos(r) > r.bend)
				r.err = true;
		}
		// ---- samples, dispatched on what the wave as a whole needs ----
		const bool predictive = h.kind >= 2 && live && !r.err;
		uint32_t need = predictive ? h.order : 0;
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) {
			const uint32_t t = __shfl_xor(need, o, 64);
			need = t > need ? t : need;
		}
		const bool any_wide = __any(predictive && h.lane_wide);
		const bool all24 = __all(!predictive || h.sbps <= 24);
		const bool run = live && !r.err;
		// what the caller may want to know about the subframe besides its samples
		uint8_t *pinfo = nullptr;
		if (SUBM && P.subinfo && run) {
			FaSubInfo *info = P.subinfo + (size_t)j * P.channels + c;
			info->type = (uint8_t)h.kind;
			info->order = (uint8_t)h.order;
			info->wasted = (uint8_t)h.wasted;
			info->precision = (uint8_t)h.prec;
			info->shift = (int8_t)h.shift;
			info->po = (uint8_t)h.po;
			info->rice2 = h.plen == 5 ? 1 : 0;
			info->constant = h.constant;
			if (h.kind >= 2) {
				for (uint32_t i = 0; i < h.order; i++) {
					info->warm[i] = L.w[i][lane];
					info->q[i] = h.kind == 3 ? L.q[i][lane] : 0;
				}
			}
			pinfo = info->params;
		}
		if constexpr (PIPE) {
			if (tid < LW) {
				X.order[lane] = h.order;
				X.wasted[lane] = h.wasted;
				X.shift[lane] = h.shift;
				X.flags[lane] = (predictive ? 1u : 0u) | (h.lane_wide ? 2u : 0u) | (h.sbps <= 24 ? 4u : 0u);
			}
			__syncthreads();
			decode_samples<0, false, false, false>(r, pm, h, L, plane, LW, tid, bs_w, bsf, run, nwords, run_len, &X, pinfo);
			continue;
		}
#define DECODE(M) \
	do { \
		if (any_wide) decode_samples<M, true, false>(r, pm, h, L, plane, LW, tid, bs_w, bsf, run, nwords, run_len, (NoPipe *)nullptr, pinfo); \
		else if (all24) decode_samples<M, false, true>(r, pm, h, L, plane, LW, tid, bs_w, bsf, run, nwords, run_len, (NoPipe *)nullptr, pinfo); \
		else decode_samples<M, false, false>(r, pm, h, L, plane, LW, tid, bs_w, bsf, run, nwords, run_len, (NoPipe *)nullptr, pinfo); \
	} while (0)
		if (MODE == DEC_SKIP) {
			decode_samples<0, false, false, true>(r, pm, h, L, plane, LW, tid, bs_w, bsf, run, nwords, run_len);
			continue;
		}
		if (need == 0) DECODE(0);
		else if (need <= 4) DECODE(4);
		else if (need <= 8) DECODE(8);
		else if (need <= 12) DECODE(12);
		else if (need <= 16) DECODE(16);
		else DECODE(32);
#undef DECODE
	}
	if (MODE == DEC_SKIP) {
		if (live) {
			// where the last subframe starts; a frame that went wrong before that is finished here
			if (br_pos(r) > r.bend)
				r.err = true;
			sub_start[(size_t)j * P.channels + P.channels - 1] = r.err ? DEC_NO_START : r.wbase * 32 + br_pos(r);
			if (r.err) {
				const uint32_t bpos = br_pos(r);
				const bool past = !endless && ((bpos > r.bend) || ((uint64_t)(bpos >> 3) + 2) * 8 > r.bend);
				cands[j].end_byte = r.wbase * 4 + (bpos >> 3) + 2;
				cands[j].status = past ? 4u : 2u;
				cands[j].reserved = (!past && reserved_code) ? 1u : 0u;
			}
		}
		return;
	}
	if constexpr (PAIR) {
		// (the first channel's lanes have nothing to finish: a subframe the walk got through cannot fail here)
		if (only_c + 1 != P.channels)
			live = false;
	} else if (SUBM && only_c + 1 != P.channels) {
		// split: a subframe the walk got through cannot fail here (the same bits, the same checks).  Chained: this
		// pass is the only one that looks at the subframe, so it says where the next one starts, or that the frame ends here
		if (chained && have_frame && !live)
			sub_start[(size_t)j * P.channels + only_c + 1] = DEC_NO_START;
		if (chained && live) {
			if (br_pos(r) > r.bend)
				r.err = true;
			sub_start[(size_t)j * P.channels + only_c + 1] = r.err ? DEC_NO_START : r.wbase * 32 + br_pos(r);
			if (r.err) {
				const uint32_t bpos = br_pos(r);
				const bool past = !endless && ((bpos > r.bend) || ((uint64_t)(bpos >> 3) + 2) * 8 > r.bend);
				cands[j].end_byte = r.wbase * 4 + (bpos >> 3) + 2;
				cands[j].status = past ? 4u : 2u;
				cands[j].reserved = (!past && reserved_code) ? 1u : 0u;
			}
		}
		return;
	}
	if (live) {
		if (br_pos(r) > r.bend)
			r.err = true;
		// read_zero_padding_ (stream_decoder.c:2773): the pad bits must be zero
		if (!r.err && (br_pos(r) & 7)) {
			const uint32_t n = 8 - (br_pos(r) & 7);
			if (br_get(r, n) != 0)
				r.err = true;
		}
		const uint32_t bpos = br_pos(r);
		const uint64_t end_byte = r.wbase * 4 + (bpos >> 3) + 2;
		const bool past = !endless && ((bpos > r.bend) || ((uint64_t)(bpos >> 3) + 2) * 8 > r.bend);
		if (past)
			r.err = true;
		cands[j].end_byte = end_byte;
		// a parse that ran off the end of the input is a cut-off frame, not a malformed one
		cands[j].status = past ? 4u : r.err ? 2u : 1u;
		uint32_t note = (!past && r.err && reserved_code) ? 1u : 0u;
		if constexpr (PAIR) {
			// bit 1: the frame's samples are in the output already (k_dec_output leaves it alone)
			if (X.fkind[tid & (FPW - 1)] == 1)
				note |= 2u;
		}
		cands[j].reserved = note;
	}
}


// what every dec_frames_*.hip defines: the launch of its instance
#define DEC_FRAMES_LAUNCHER(name, MODE, MAXLW, DEEP) \
	void name(const FaDecParams &P, const uint32_t *words, const FaStream *streams, FaFrameCand *cands, const uint32_t *ncand, int32_t *tmp, uint32_t lw, \
	          uint64_t *sub_start, uint32_t nwaves, uint32_t chain, const uint64_t *ahead, void *pcm_out, hipStream_t st) \
	{ \
		hipLaunchKernelGGL((k_dec_frames<MODE, MAXLW, DEEP>), dim3(nwaves), dim3((MODE == DEC_PIPE || MODE == DEC_PAIR) ? 2 * DEC_LANES : DEC_LANES), 0, st, P, words, streams, \
		                   cands, ncand, tmp, lw, sub_start, chain, ahead, pcm_out); \
	}

#endif
