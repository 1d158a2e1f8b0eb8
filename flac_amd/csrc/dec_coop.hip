// FLAC frame decoder kernels for gfx950 (MI355X), cooperative form: for batches of few frames.
//
// k_dec_frames (dec_kernels.hip) gives every frame one lane, which fills the chip only when there
// are tens of thousands of frames; with fewer, its running time is that of ONE lane walking a whole
// frame's bits (a lone wave issues a dependent instruction every ~8 cycles).  Here a whole wave
// reads one frame:
//
//   k_dec_entropy   one wave per frame.  Subframe headers are read by the wave as a whole; warm-up
//                   samples, predictor coefficients, verbatim samples and escaped (raw) residuals
//                   are fixed-width fields, so lane l reads field l: 64 per step.  Rice-coded
//                   residuals (FLAC__bitreader_read_rice_signed_block, bitreader.c:775) are read 64
//                   BIT POSITIONS per step: lane l looks at the 32 bits from position bp + l, the
//                   wave's ballot of "my first bit is set" is the map of one-bits of the window,
//                   and a scalar walk over that 64-bit map finds the stop bit of code after code
//                   (first one-bit at or after the code's start; the next code starts k + 1 bits
//                   further) - a handful of scalar instructions per code, no memory access.  The
//                   lanes that sit on stop bits then cut out their code's low bits, learn the length
//                   of their zero run from the stop in front of them, and drop the residual into a
//                   staging ring that leaves for memory 64 values at a time.  Zero runs longer than
//                   the window carry over to the next step.
//   k_dec_restore   one lane per subframe (FLAC__fixed_restore_signal fixed.c:395,
//                   FLAC__lpc_restore_signal[_wide] lpc.c:795,1061): the predictor recurrence is serial
//                   per subframe, but now there are (frames x channels) of them side by side, fed with
//                   residual blocks through LDS and writing the same scratch planes k_dec_output reads.
#include <hip/hip_runtime.h>

#include "dec_kernels.h"

#define CO_RING 1024  // input words held in LDS
#define CO_FILL 512   // words brought in per refill
#define CO_STAGE 256  // staging ring for decoded values

__constant__ int8_t c_co_fixed_q[5][4] = {{0, 0, 0, 0}, {1, 0, 0, 0}, {2, -1, 0, 0}, {3, -3, 1, 0}, {4, -6, 4, -1}};

__device__ __forceinline__ int32_t co_sext(uint32_t v, uint32_t bits) { return bits ? (int32_t)(v << (32 - bits)) >> (32 - bits) : 0; }
__device__ __forceinline__ uint32_t co_uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t co_ilog2(uint32_t v) { return 31u - (uint32_t)__clz((int)v); }

__global__ __launch_bounds__(64) void k_dec_entropy(FaDecParams P, const uint32_t *flac_words, const FaStream *streams, FaFrameCand *cands,
                                                    const uint32_t *ncand_ptr, int32_t *vals, FaSubDesc *subs)
{
	__shared__ uint32_t ring[CO_RING];
	__shared__ int32_t stage[CO_STAGE];
	const uint32_t j = blockIdx.x;
	if (j >= ncand_ptr[0])
		return;
	const uint32_t lane = threadIdx.x;
	// the frame's row of the table, the same in every lane: kept in scalar registers
	FaFrameCand cd;
	{
		const FaFrameCand ld = cands[j];
		cd.pos = ((uint64_t)co_uniform((uint32_t)(ld.pos >> 32)) << 32) | co_uniform((uint32_t)ld.pos);
		cd.hdr_bytes = co_uniform(ld.hdr_bytes);
		cd.blocksize = co_uniform(ld.blocksize);
		cd.assignment = co_uniform(ld.assignment);
		cd.stream = co_uniform(ld.stream);
	}
	const uint64_t nwords = (P.flac_bytes + 3) / 4;
	const uint64_t wbase = cd.pos >> 2;
	const uint32_t *wg = flac_words + wbase;
	const uint64_t navail = nwords - wbase;
	// positions are 32-bit bit counts from the frame's first word; a stream that goes on for more
	// than that is treated as endless
	bool endless;
	uint32_t bend;
	{
		const uint64_t end_abs = (streams[cd.stream].byte_offset + streams[cd.stream].byte_length) * 8;
		const uint64_t rel = end_abs > wbase * 32 ? end_abs - wbase * 32 : 0;
		endless = rel > 0xf0000000ull;
		bend = endless ? 0xf0000000u : (uint32_t)rel;
	}
	auto load_chunk = [&](uint32_t first_word) -> uint32_t {
		const uint64_t idx = (uint64_t)first_word + lane;
		return idx < navail ? wg[idx] : 0u;
	};
	// words [ring_hi - CO_RING, ring_hi) of the frame are in the ring
	uint32_t ring_hi = 0;
	auto bring = [&]() {
		uint32_t w[CO_FILL / 64];
#pragma unroll
		for (uint32_t r = 0; r < CO_FILL / 64; r++)
			w[r] = load_chunk(ring_hi + 64 * r);
#pragma unroll
		for (uint32_t r = 0; r < CO_FILL / 64; r++)
			ring[(ring_hi + 64 * r + lane) & (CO_RING - 1)] = __builtin_bswap32(w[r]);
		ring_hi += CO_FILL;
		__syncthreads();
	};
	bring();

	uint32_t bp = ((uint32_t)cd.pos & 3) * 8 + cd.hdr_bytes * 8;  // next unread bit (the same in every lane)
	bool err = false, reserved_code = false;
	const uint32_t bsf = cd.blocksize;

	// any read below may look up to 64 fields of 32 bits ahead; a refill is one burst of loads that is waited
	// for on the spot (there is one per 16 Kibit of stream, and nothing else to do meanwhile in this wave)
	auto refill = [&]() {
		while ((bp >> 5) + 72 > ring_hi)
			bring();
	};
	// the 32 bits that start at bit position pos
	auto peek = [&](uint32_t pos) -> uint32_t {
		const uint32_t wi = pos >> 5, o = pos & 31;
		const uint32_t a = ring[wi & (CO_RING - 1)], b = ring[(wi + 1) & (CO_RING - 1)];
		return (uint32_t)((((uint64_t)a << 32) | b) >> (32 - o));
	};
	auto getbits = [&](uint32_t n) -> uint32_t {  // n <= 32, the same in every lane
		if (n == 0)
			return 0;
		const uint32_t v = co_uniform(peek(bp) >> (32 - n));
		bp += n;
		return v;
	};
	auto getunary = [&]() -> uint32_t {  // FLAC__bitreader_read_unary_unsigned (bitreader.c:666)
		uint32_t n = 0;
		for (;;) {
			refill();
			const uint32_t w = co_uniform(peek(bp));
			if (w) {
				const uint32_t z = (uint32_t)__clz((int)w);
				bp += z + 1;
				return n + z;
			}
			n += 32;
			bp += 32;
			if (bp > bend) {
				err = true;
				return n;
			}
		}
	};

	for (uint32_t c = 0; c < P.channels; c++) {
		int32_t *out = vals + ((size_t)j * P.channels + c) * P.max_blocksize;
		uint32_t flushed = 0;
		// values reach memory 64 at a time, in sample order, from the staging ring
		auto flush_to = [&](uint32_t upto, bool final) {
			while (flushed + 64 <= upto || (final && flushed < upto)) {
				__syncthreads();
				const uint32_t idx = flushed + lane;
				if (idx < upto)
					out[idx] = stage[idx & (CO_STAGE - 1)];
				flushed += 64;
				__syncthreads();
			}
		};
		// n fields of `bits` bits each, lane l reading field l of every run of 64
		auto read_fields = [&](uint32_t first_i, uint32_t n, uint32_t bits) {
			for (uint32_t done = 0; done < n; done += 64) {
				refill();
				const uint32_t m = n - done < 64 ? n - done : 64;
				if (lane < m)
					stage[(first_i + done + lane) & (CO_STAGE - 1)] = bits ? co_sext(peek(bp + lane * bits) >> (32 - bits), bits) : 0;
				bp += m * bits;
				flush_to(first_i + done + m, false);
				if (bp > bend) {
					err = true;
					break;
				}
			}
		};

		// ---- subframe header (stream_decoder.c:2450-2518) ----
		uint32_t kind = 0, order = 0, wasted = 0, wide = 0;
		int32_t shift = 0, constant = 0;
		uint32_t sbps = P.bps;
		if ((cd.assignment == 1 && c == 1) || (cd.assignment == 2 && c == 0) || (cd.assignment == 3 && c == 1))
			sbps++;
		if (sbps > 32)
			err = true;
		if (!err) {
			refill();
			uint32_t x = getbits(8);
			const uint32_t has_wasted = x & 1;
			x &= 0xfe;
			if (has_wasted) {
				wasted = getunary() + 1;
				if (wasted >= sbps)
					err = true;
				else
					sbps -= wasted;
			}
			// reserved type codes are "unparseable", a set padding bit is lost sync (stream_decoder.c:2476-2503)
			if (x & 0x80) err = true;
			else if (x == 0) kind = 0;
			else if (x == 2) kind = 1;
			else if (x < 16) err = reserved_code = true;
			else if (x <= 24) {
				kind = 2;
				order = (x >> 1) & 7;
				if (order > 4) err = true;
			} else if (x < 64) err = reserved_code = true;
			else {
				kind = 3;
				order = ((x >> 1) & 31) + 1;
			}
			if (order > bsf)
				err = true;
		}
		int32_t qv = 0;
		if (!err) {
			refill();
			if (kind == 0)
				constant = co_sext(getbits(sbps), sbps);
			else if (kind == 1)
				read_fields(0, bsf, sbps);
			else {
				// warm-up samples, one per lane
				if (lane < order)
					stage[lane] = co_sext(peek(bp + lane * sbps) >> (32 - sbps), sbps);
				bp += order * sbps;
				refill();
				if (kind == 3) {
					uint32_t prec = getbits(4);
					if (prec == 15)
						err = true;  // stream_decoder.c:2622
					prec++;
					shift = co_sext(getbits(5), 5);
					if (shift < 0)
						shift = 0;  // the reference's shift by a negative count is undefined; it never writes one
					if (lane < order)
						qv = co_sext(peek(bp + lane * prec) >> (32 - prec), prec);
					bp += order * prec;
					wide = !(sbps + prec + co_ilog2(order) <= 32);  // stream_decoder.c:2676
					refill();
				} else if (lane < order)
					qv = c_co_fixed_q[order][lane & 3];
				const uint32_t method = getbits(2);
				if (method > 1)
					err = reserved_code = true;  // stream_decoder.c:2575,2654
				const uint32_t plen = method == 1 ? 5 : 4, pesc = method == 1 ? 31 : 15;
				const uint32_t po = getbits(4);
				// stream_decoder.c:2721-2739
				const uint32_t psize = po > 0 ? bsf >> po : bsf - order;
				if (po > 0 && (psize < order || (psize << po) != bsf))
					err = true;
				uint32_t i = order;
				for (uint32_t part = 0; part < (1u << po) && !err; part++) {
					const uint32_t n = po == 0 ? bsf - order : (part == 0 ? psize - order : psize);
					refill();
					const uint32_t k = getbits(plen);
					if (k >= pesc) {
						// escaped partition: n raw fields (stream_decoder.c:2758-2767)
						const uint32_t raw = getbits(5);
						read_fields(i, n, raw);
						i += n;
					} else {
						const uint32_t k1 = k + 1;
						uint32_t rem = n;
						uint32_t zc = 0;  // zeros of the code in progress that lie in earlier windows
						while (rem && !err) {
							refill();
							const uint32_t w = peek(bp + lane);
							const uint64_t ones = __ballot((int32_t)w < 0);
							// per lane: if a code started at my bit, where would its stop bit be, and where the next code's start
							const uint64_t ahead = ones >> lane;
							const uint32_t zl = ahead ? (uint32_t)__builtin_ctzll(ahead) : 64u;
							const uint32_t nxt = ahead ? lane + zl + k1 : 127u;  // 127: the stop bit is beyond the window
							// the walk from code start to code start: one lane read per code
							uint64_t starts = 0;
							uint32_t cur = 0;
							do {
								starts |= 1ull << cur;
								cur = (uint32_t)__builtin_amdgcn_readlane((int)nxt, (int)cur);
							} while (cur < 64);
							bool open = cur == 127u;
							uint32_t cnt = (uint32_t)__popcll(starts);
							uint32_t last_start = 63u - (uint32_t)__clzll((long long)starts);
							if (open) {
								starts &= ~(1ull << last_start);  // the last code has no stop bit yet
								cnt--;
							}
							if (cnt > rem || (open && cnt == rem)) {
								// the partition ends inside the window (what follows its last code is the next
								// partition's parameter, not a code): walk again, counting
								starts = 0;
								cur = 0;
								for (uint32_t n2 = 0; n2 < rem; n2++) {
									starts |= 1ull << cur;
									cur = (uint32_t)__builtin_amdgcn_readlane((int)nxt, (int)cur);
								}
								cnt = rem;
								open = false;
							}
							// the lanes on code starts make the values; the low bits come from the lane on the stop bit
							const uint32_t wstop = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((lane + zl) & 63u) << 2), (int)w);
							if ((starts >> lane) & 1ull) {
								const uint32_t z = zl + (lane == 0 ? zc : 0u);
								const uint32_t lsbs = k ? (wstop << 1) >> (32 - k) : 0u;
								const uint32_t uf = (z << k) | lsbs;
								const uint64_t below = starts & ((1ull << lane) - 1ull);
								stage[(i + (uint32_t)__popcll(below)) & (CO_STAGE - 1)] = (int32_t)(uf >> 1) ^ -(int32_t)(uf & 1);
							}
							if (open) {
								zc = (last_start == 0 ? zc : 0u) + (64 - last_start);
								bp += 64;
							} else {
								zc = 0;
								bp += cur;
							}
							i += cnt;
							rem -= cnt;
							flush_to(i, false);
							if (bp > bend)
								err = true;
						}
					}
					if (bp > bend)
						err = true;
				}
			}
		}
		flush_to(kind == 1 ? bsf : (kind >= 2 ? bsf : 0), true);
		// what the restore pass needs to know about this subframe
		FaSubDesc *sd = subs + (size_t)j * P.channels + c;
		if (lane < FA_MAX_ORDER)
			sd->q[lane] = (kind >= 2 && lane < order) ? qv : 0;
		if (lane == 0) {
			sd->kind = kind;
			sd->order = kind >= 2 ? order : 0;
			sd->wasted = wasted;
			sd->sbps = sbps;
			sd->shift = shift;
			sd->wide = wide;
			sd->constant = constant;
			sd->pad = 0;
		}
		if (bp > bend)
			err = true;
	}

	// read_zero_padding_ (stream_decoder.c:2773): the pad bits must be zero
	if (!err && (bp & 7)) {
		refill();
		if (getbits(8 - (bp & 7)) != 0)
			err = true;
	}
	if (lane == 0) {
		const uint64_t end_byte = wbase * 4 + (bp >> 3) + 2;
		const bool past = !endless && ((bp > bend) || ((uint64_t)(bp >> 3) + 2) * 8 > bend);
		cands[j].end_byte = end_byte;
		// a parse that ran off the end of the input is a cut-off frame, not a malformed one
		cands[j].status = past ? 4u : err ? 2u : 1u;
		cands[j].reserved = (!past && err && reserved_code) ? 1u : 0u;
	}
}

// ---------------------------------------------------------------------------------------------
// k_dec_restore: lane (c, fl) of wave g restores channel c of frame g * LW + fl.
// ARITH 0: 24-bit multiplies (every subframe of the wave holds at most 24 bits per sample and needs
// 32-bit sums only), 1: 32-bit multiplies, wrapping sums, 2: 64-bit sums where the lane's subframe
// asks for them (stream_decoder.c:2676-2686).
// ---------------------------------------------------------------------------------------------
#define RS_BLOCK 16

template <int M, int ARITH>
__device__ __forceinline__ void restore_lanes(const FaDecParams &P, const int32_t *vals, int32_t *planes, int32_t *lds_in, uint32_t lane, uint32_t LW, bool live,
                                              uint32_t sub_index, uint32_t c, uint32_t fl, const FaSubDesc *sd, uint32_t bs, uint32_t bs_w)
{
	int32_t q[M > 0 ? M : 1], hist[M > 0 ? M : 1];
	uint32_t order = 0, wasted = 0, kind = 1;
	int32_t shift = 0, constant = 0;
	bool wide = false;
#pragma unroll
	for (int jx = 0; jx < (M > 0 ? M : 1); jx++) {
		q[jx] = 0;
		hist[jx] = 0;
	}
	if (live) {
		kind = sd->kind;
		order = sd->order;
		wasted = sd->wasted;
		shift = sd->shift;
		wide = sd->wide != 0;
		constant = sd->constant;
#pragma unroll
		for (int jx = 0; jx < M; jx++)
			q[jx] = sd->q[jx];
	}
	const size_t my_base = (size_t)sub_index * P.max_blocksize;
	// block b of every lane's values comes in through LDS: four lanes fetch one subframe's 16 values
	auto fetch = [&](uint32_t i0, int32_t (&v)[4][4]) {
#pragma unroll
		for (int r = 0; r < 4; r++) {
			const uint32_t sub = (lane >> 2) + 16 * r, part = lane & 3;
			const uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(my_base >> 32), (int)sub, 64), lo = (uint32_t)__shfl((int)(uint32_t)my_base, (int)sub, 64);
			const int ok = __shfl((int)(live && i0 < bs), (int)sub, 64);
			const int32_t *src = vals + (((size_t)hi << 32) | lo) + i0 + part * 4;
#pragma unroll
			for (int e = 0; e < 4; e++)
				v[r][e] = ok ? src[e] : 0;
		}
	};
	auto stash = [&](const int32_t (&v)[4][4]) {
#pragma unroll
		for (int r = 0; r < 4; r++) {
			const uint32_t sub = (lane >> 2) + 16 * r, part = lane & 3;
#pragma unroll
			for (int e = 0; e < 4; e++)
				lds_in[sub * 17 + part * 4 + e] = v[r][e];
		}
	};
	int32_t nextv[4][4];
	fetch(0, nextv);
	for (uint32_t i0 = 0; i0 < bs_w; i0 += RS_BLOCK) {
		__syncthreads();
		stash(nextv);
		__syncthreads();
		if (i0 + RS_BLOCK < bs_w)
			fetch(i0 + RS_BLOCK, nextv);
		int32_t x[RS_BLOCK];
#pragma unroll
		for (int s = 0; s < RS_BLOCK; s++)
			x[s] = lds_in[lane * 17 + s];
#pragma unroll
		for (int s = 0; s < RS_BLOCK; s++) {
			const uint32_t i = i0 + s;
			int32_t v = kind == 0 ? constant : x[s];
			if (M > 0) {
				int32_t pred;
				if (ARITH == 2) {
					int64_t sum = 0;
#pragma unroll
					for (int jx = 0; jx < M; jx++)
						sum += (int64_t)q[jx] * (int64_t)hist[jx];
					pred = wide ? (int32_t)(sum >> shift) : ((int32_t)sum >> shift);
				} else if (ARITH == 0) {
					int32_t s0, s1 = 0;
					asm("v_mul_i32_i24 %0, %1, %2" : "=v"(s0) : "v"(q[0]), "v"(hist[0]));
					if (M > 1)
						asm("v_mul_i32_i24 %0, %1, %2" : "=v"(s1) : "v"(q[1]), "v"(hist[1]));
#pragma unroll
					for (int jx = 2; jx < M; jx += 2) {
						asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(s0) : "v"(q[jx]), "v"(hist[jx]), "v"(s0));
						if (jx + 1 < M)
							asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(s1) : "v"(q[jx + 1]), "v"(hist[jx + 1]), "v"(s1));
					}
					pred = (s0 + s1) >> shift;
				} else {
					int32_t sum = 0;
#pragma unroll
					for (int jx = 0; jx < M; jx++)
						sum += q[jx] * hist[jx];
					pred = sum >> shift;
				}
				// warm-up samples are stored as they are; the predictor comes in at sample `order`
				v = (int32_t)((uint32_t)v + (uint32_t)(i >= order ? pred : 0));
#pragma unroll
				for (int jx = M - 1; jx > 0; jx--)
					hist[jx] = hist[jx - 1];
				hist[0] = v;
			}
			if (live && i < bs)
				planes[((size_t)c * P.max_blocksize + i) * LW + fl] = (int32_t)((uint32_t)v << wasted);
		}
	}
}

__global__ __launch_bounds__(64) void k_dec_restore(FaDecParams P, const FaFrameCand *cands, const uint32_t *ncand_ptr, const int32_t *vals, const FaSubDesc *subs,
                                                    int32_t *tmp, uint32_t LW, uint32_t lw_log)
{
	__shared__ int32_t lds_in[64 * 17];
	const uint32_t ncand = ncand_ptr[0];
	const uint32_t g = blockIdx.x;
	if (g * LW >= ncand)
		return;
	const uint32_t lane = threadIdx.x;
	const uint32_t c = lane >> lw_log, fl = lane & (LW - 1);
	const uint32_t f = g * LW + fl;
	bool live = c < P.channels && f < ncand;
	uint32_t bs = 0;
	if (live) {
		live = cands[f].status == 1;
		bs = cands[f].blocksize;
	}
	if (!live)
		bs = 0;
	const uint32_t sub_index = live ? f * P.channels + c : 0;
	const FaSubDesc *sd = subs + sub_index;
	uint32_t need = 0, any_wide = 0, over24 = 0;
	if (live) {
		need = sd->order;
		any_wide = sd->wide;
		over24 = sd->sbps > 24 ? 1u : 0u;
	}
	uint32_t bs_w = bs;
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		const uint32_t a = (uint32_t)__shfl_xor((int)need, o, 64), b = (uint32_t)__shfl_xor((int)bs_w, o, 64);
		need = a > need ? a : need;
		bs_w = b > bs_w ? b : bs_w;
		any_wide |= (uint32_t)__shfl_xor((int)any_wide, o, 64);
		over24 |= (uint32_t)__shfl_xor((int)over24, o, 64);
	}
	int32_t *planes = tmp + (size_t)g * P.channels * P.max_blocksize * LW;
#define RESTORE(M) \
	do { \
		if (any_wide) restore_lanes<M, 2>(P, vals, planes, lds_in, lane, LW, live, sub_index, c, fl, sd, bs, bs_w); \
		else if (!over24) restore_lanes<M, 0>(P, vals, planes, lds_in, lane, LW, live, sub_index, c, fl, sd, bs, bs_w); \
		else restore_lanes<M, 1>(P, vals, planes, lds_in, lane, LW, live, sub_index, c, fl, sd, bs, bs_w); \
	} while (0)
	if (need == 0) restore_lanes<0, 1>(P, vals, planes, lds_in, lane, LW, live, sub_index, c, fl, sd, bs, bs_w);
	else if (need <= 4) RESTORE(4);
	else if (need <= 8) RESTORE(8);
	else if (need <= 12) RESTORE(12);
	else if (need <= 16) RESTORE(16);
	else RESTORE(32);
#undef RESTORE
}

// frames per group of the scratch planes in this form: as many as a wave has lanes for, a power of two
uint32_t fa_dec_coop_lw(uint32_t channels)
{
	uint32_t lw = 64;
	while (lw * channels > 64)
		lw >>= 1;
	return lw ? lw : 1;
}

void fa_dec_launch_coop(const FaDecParams &P, const uint8_t *flac, const FaStream *streams, FaFrameCand *cands, const uint32_t *ncand, int32_t *vals,
                        FaSubDesc *subs, int32_t *tmp, hipStream_t st)
{
	hipLaunchKernelGGL(k_dec_entropy, dim3(P.max_frames), dim3(64), 0, st, P, (const uint32_t *)flac, streams, cands, ncand, vals, subs);
	uint32_t lw_log = 0;
	while ((1u << lw_log) < P.lw)
		lw_log++;
	const uint32_t groups = (P.max_frames + P.lw - 1) / P.lw;
	hipLaunchKernelGGL(k_dec_restore, dim3(groups), dim3(64), 0, st, P, cands, ncand, vals, subs, tmp, P.lw, lw_log);
}
