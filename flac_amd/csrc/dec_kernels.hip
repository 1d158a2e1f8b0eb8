// FLAC frame decoder kernels for gfx950 (MI355X).
//
//   k_dec_sync<false/true>  frame_sync_ + read_frame_header_ (stream_decoder.c:1941,2141): every byte
//                           position is tested for a sync code with a parseable header and a good
//                           CRC-8; pass 1 counts per block, pass 2 writes the candidates in order
//   k_dec_frames            read_subframe_* / read_residual_partitioned_rice_ (stream_decoder.c:2450-
//                           2771) and the predictor restore (fixed.c:395, lpc.c:795,1061): one lane
//                           per frame, the 64 lanes of a wave walk sample index i together so the
//                           scratch planes [channel][i][lane] are read and written coalesced
//   k_dec_crc               FLAC__crc16 over each frame (stream_decoder.c:2061-2064), one block per frame
//   k_dec_output            channel un-decorrelation (stream_decoder.c:2067-2102), width conversion
//                           and transposition into interleaved PCM
//   k_dec_summary           frame / sample / bad-frame counts
//
// A frame's place in the output comes from the number in its own header, so frames need no
// knowledge of each other.  Candidates that fail the structure checks or the CRC-16 write nothing;
// the output is cleared beforehand, which is also what the reference hands out for a frame with a
// bad CRC (stream_decoder.c:2105-2113).
#include <hip/hip_runtime.h>

#include "dec_kernels.h"

#define DEC_LANES 64

__device__ __forceinline__ uint32_t d_ilog2(uint32_t v) { return 31u - (uint32_t)__clz((int)v); }

// ---------------------------------------------------------------------------------------------
// frame header
// ---------------------------------------------------------------------------------------------
__constant__ int8_t c_fixed_q[5][4] = {{0, 0, 0, 0}, {1, 0, 0, 0}, {2, -1, 0, 0}, {3, -3, 1, 0}, {4, -6, 4, -1}};
__constant__ uint8_t c_dec_crc8[256];
__constant__ uint16_t c_dec_crc16[256];
__constant__ uint16_t c_dec_pow16[32];

void fa_dec_upload_tables(const uint16_t *t16, const uint16_t *pow16, const uint8_t *t8)
{
	(void)hipMemcpyToSymbol(HIP_SYMBOL(c_dec_crc16), t16, 512);
	(void)hipMemcpyToSymbol(HIP_SYMBOL(c_dec_pow16), pow16, 64);
	(void)hipMemcpyToSymbol(HIP_SYMBOL(c_dec_crc8), t8, 256);
}

__device__ __forceinline__ int find_stream(const FaStream *streams, uint32_t n, uint64_t p)
{
	int lo = 0, hi = (int)n - 1, ans = -1;
	while (lo <= hi) {
		const int mid = (lo + hi) >> 1;
		if (streams[mid].byte_offset <= p) {
			ans = mid;
			lo = mid + 1;
		} else
			hi = mid - 1;
	}
	if (ans >= 0 && p >= streams[ans].byte_offset + streams[ans].byte_length)
		ans = -1;
	return ans;
}

// read_frame_header_ (stream_decoder.c:2141-2448).  Returns true when position p holds a sync code
// followed by a header that parses, matches the stream's channel count / sample width, and
// carries the right CRC-8.
__device__ bool parse_frame_header(const uint8_t *d, uint64_t p, const FaDecParams &P, const FaStream *streams, FaFrameCand &c)
{
	if (d[p] != 0xff)
		return false;
	const int sidx = find_stream(streams, P.nstreams, p);
	if (sidx < 0)
		return false;
	const uint64_t limit = streams[sidx].byte_offset + streams[sidx].byte_length;
	if (p + 6 > limit)
		return false;
	const uint32_t b1 = d[p + 1], b2 = d[p + 2], b3 = d[p + 3];
	if ((b1 & 0xfe) != 0xf8)  // 14-bit sync + reserved bit 0 (stream_decoder.c:2158)
		return false;
	const uint32_t variable = b1 & 1;
	const uint32_t bs_code = b2 >> 4, sr_code = b2 & 15;
	if (bs_code == 0 || sr_code == 15)
		return false;
	if (b3 & 1)
		return false;
	const uint32_t ch_code = b3 >> 4, bps_code = (b3 >> 1) & 7;
	uint32_t nch, assignment = 0;
	if (ch_code & 8) {
		if ((ch_code & 7) > 2)
			return false;
		nch = 2;
		assignment = (ch_code & 7) + 1;
	} else
		nch = ch_code + 1;
	if (nch != P.channels)
		return false;
	uint32_t bps;
	switch (bps_code) {
	case 0: bps = P.bps; break;
	case 1: bps = 8; break;
	case 2: bps = 12; break;
	case 4: bps = 16; break;
	case 5: bps = 20; break;
	case 6: bps = 24; break;
	default: return false;
	}
	if (bps != P.bps)
		return false;
	// the frame / sample number (bitreader.c:999,1054)
	uint32_t hp = 4;
	uint64_t v;
	int extra;
	const uint32_t x = d[p + hp++];
	if (!(x & 0x80)) { v = x; extra = 0; }
	else if ((x & 0xe0) == 0xc0) { v = x & 0x1f; extra = 1; }
	else if ((x & 0xf0) == 0xe0) { v = x & 0x0f; extra = 2; }
	else if ((x & 0xf8) == 0xf0) { v = x & 0x07; extra = 3; }
	else if ((x & 0xfc) == 0xf8) { v = x & 0x03; extra = 4; }
	else if ((x & 0xfe) == 0xfc) { v = x & 0x01; extra = 5; }
	else if (x == 0xfe && variable) { v = 0; extra = 6; }
	else return false;
	if (p + hp + extra + 1 > limit)
		return false;
	for (int i = 0; i < extra; i++) {
		const uint32_t b = d[p + hp++];
		if ((b & 0xc0) != 0x80)
			return false;
		v = (v << 6) | (b & 0x3f);
	}
	uint32_t blocksize;
	if (bs_code == 1) blocksize = 192;
	else if (bs_code <= 5) blocksize = 576u << (bs_code - 2);
	else if (bs_code == 6) {
		if (p + hp + 2 > limit) return false;
		blocksize = d[p + hp] + 1u;
		hp += 1;
	} else if (bs_code == 7) {
		if (p + hp + 3 > limit) return false;
		blocksize = (((uint32_t)d[p + hp] << 8) | d[p + hp + 1]) + 1u;
		hp += 2;
	} else blocksize = 256u << (bs_code - 8);
	if (sr_code == 12) hp += 1;
	else if (sr_code == 13 || sr_code == 14) hp += 2;
	if (p + hp + 1 > limit)
		return false;
	uint8_t crc = 0;
	for (uint32_t i = 0; i < hp; i++)
		crc = c_dec_crc8[crc ^ d[p + i]];
	if (crc != d[p + hp])
		return false;
	hp++;
	if (blocksize > P.max_blocksize)
		return false;
	c.pos = p;
	c.hdr_bytes = hp;
	c.blocksize = blocksize;
	c.assignment = assignment;
	c.stream = (uint32_t)sidx;
	// a fixed-blocksize stream numbers frames; the sample number follows from the stream's
	// blocksize (stream_decoder.c:2416-2427)
	c.sample = variable ? v : v * (uint64_t)P.number_scale;
	c.status = 0;
	c.end_byte = 0;
	return true;
}

#define SYNC_BYTES_PER_THREAD 16
#define SYNC_THREADS 256

template <bool EMIT>
__global__ __launch_bounds__(SYNC_THREADS) void k_dec_sync(FaDecParams P, const uint8_t *d, const FaStream *streams, uint32_t *block_counts,
                                                           const uint64_t *block_offsets, FaFrameCand *cands, uint32_t *ncand_out)
{
	__shared__ uint32_t wsum[4];
	const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const uint64_t p0 = ((uint64_t)blockIdx.x * SYNC_THREADS + tid) * SYNC_BYTES_PER_THREAD;
	FaFrameCand found[2];  // more than two real headers in 16 bytes cannot happen
	uint32_t n = 0;
	for (uint32_t i = 0; i < SYNC_BYTES_PER_THREAD; i++) {
		const uint64_t p = p0 + i;
		if (p >= P.flac_bytes)
			break;
		FaFrameCand c;
		if (parse_frame_header(d, p, P, streams, c)) {
			if (n < 2)
				found[n] = c;
			n++;
		}
	}
	if (n > 2)
		n = 2;
	uint32_t inc = n;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const uint32_t t = __shfl_up(inc, o, 64);
		if (lane >= (uint32_t)o)
			inc += t;
	}
	if (lane == 63)
		wsum[wv] = inc;
	__syncthreads();
	uint32_t off = 0;
	for (uint32_t w = 0; w < wv; w++)
		off += wsum[w];
	if (!EMIT) {
		if (tid == SYNC_THREADS - 1)
			block_counts[blockIdx.x] = off + inc;
	} else {
		const uint64_t base = block_offsets[blockIdx.x] + off + inc - n;
		for (uint32_t i = 0; i < n; i++)
			if (base + i < P.max_frames)
				cands[base + i] = found[i];
		if (blockIdx.x == gridDim.x - 1 && tid == SYNC_THREADS - 1) {
			const uint64_t total = block_offsets[blockIdx.x] + off + inc;
			ncand_out[0] = total > P.max_frames ? P.max_frames : (uint32_t)total;
			ncand_out[1] = total > P.max_frames ? 1u : 0u;  // overflow flag
		}
	}
}

// exclusive scan of the per-block counts (single block)
__global__ __launch_bounds__(1024) void k_dec_scan(const uint32_t *counts, uint32_t n, uint64_t *offsets)
{
	__shared__ uint64_t wsum[16];
	__shared__ uint64_t carry;
	const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	if (tid == 0)
		carry = 0;
	__syncthreads();
	for (uint32_t base = 0; base < n; base += 1024) {
		const uint32_t i = base + tid;
		const uint64_t v = i < n ? counts[i] : 0;
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
}

// ---------------------------------------------------------------------------------------------
// bit reader (bitreader.c): MSB-first bits out of big-endian 32-bit words
// ---------------------------------------------------------------------------------------------
struct BitRd {
	const uint32_t *w;
	uint64_t nwords;
	uint64_t pos;   // absolute bit position
	uint64_t end;   // first bit past the stream
	bool err;
};

__device__ __forceinline__ uint32_t br_word(const BitRd &r, uint64_t i) { return i < r.nwords ? __builtin_bswap32(r.w[i]) : 0u; }

__device__ __forceinline__ uint32_t br_peek32(const BitRd &r)
{
	const uint64_t i = r.pos >> 5;
	const uint32_t s = (uint32_t)r.pos & 31;
	const uint32_t w0 = br_word(r, i);
	if (s == 0)
		return w0;
	return (w0 << s) | (br_word(r, i + 1) >> (32 - s));
}

__device__ __forceinline__ uint32_t br_get(BitRd &r, uint32_t n)
{
	if (n == 0)
		return 0;
	const uint32_t v = br_peek32(r) >> (32 - n);
	r.pos += n;
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
		const uint32_t v = br_peek32(r);
		if (v) {
			const uint32_t z = (uint32_t)__clz((int)v);
			r.pos += z + 1;
			return n + z;
		}
		n += 32;
		r.pos += 32;
		if (r.pos > r.end) {
			r.err = true;
			return n;
		}
	}
}

// ---------------------------------------------------------------------------------------------
// k_dec_frames
// ---------------------------------------------------------------------------------------------
struct DecLds {
	int32_t q[FA_MAX_ORDER][DEC_LANES];
};

// Predictor restore for one channel plane: FLAC__fixed_restore_signal (fixed.c:395) and
// FLAC__lpc_restore_signal[_wide] (lpc.c:795,1061) as one recurrence; a fixed predictor is the
// LPC filter {1}, {2,-1}, {3,-3,1}, {4,-6,4,-1} with shift 0 in wrapping 32-bit arithmetic.
template <int M, bool WIDE, bool MUL24>
__device__ __forceinline__ void restore_plane(int32_t *plane, const DecLds &L, uint32_t lane, uint32_t bs_w, uint32_t bsf, uint32_t order,
                                              int shift, bool lane_wide, bool predictive, uint32_t wasted, bool live)
{
	int32_t q[M > 0 ? M : 1], h[M > 0 ? M : 1];
#pragma unroll
	for (int j = 0; j < M; j++) {
		q[j] = ((uint32_t)j < order && predictive) ? L.q[j][lane] : 0;
		h[j] = 0;
	}
	for (uint32_t i = 0; i < bs_w; i++) {
		if (!(live && i < bsf))
			continue;
		int32_t v = plane[(size_t)i * DEC_LANES + lane];
		if (M > 0 && predictive && i >= order) {
			int32_t pred;
			if (WIDE) {
				int64_t sum = 0;
#pragma unroll
				for (int j = 0; j < M; j++)
					sum += (int64_t)q[j] * (int64_t)h[j];
				pred = lane_wide ? (int32_t)(sum >> shift) : ((int32_t)sum >> shift);
			} else {
				int32_t sum = 0;
#pragma unroll
				for (int j = 0; j < M; j++)
					sum += MUL24 ? __mul24(q[j], h[j]) : q[j] * h[j];
				pred = sum >> shift;
			}
			v = (int32_t)((uint32_t)v + (uint32_t)pred);
		}
		if (M > 0) {
#pragma unroll
			for (int j = M - 1; j > 0; j--)
				h[j] = h[j - 1];
			h[0] = v;
		}
		// wasted bits are restored after prediction (stream_decoder.c:2511-2515)
		plane[(size_t)i * DEC_LANES + lane] = (int32_t)((uint32_t)v << wasted);
	}
}

__global__ __launch_bounds__(DEC_LANES) void k_dec_frames(FaDecParams P, const uint32_t *flac_words, const FaStream *streams,
                                                          FaFrameCand *cands, const uint32_t *ncand_ptr, int32_t *tmp)
{
	__shared__ DecLds L;
	const uint32_t lane = threadIdx.x;
	const uint32_t ncand = ncand_ptr[0];
	if (blockIdx.x * DEC_LANES >= ncand)
		return;
	const uint32_t j = blockIdx.x * DEC_LANES + lane;
	const bool live = j < ncand;
	FaFrameCand cd;
	if (live)
		cd = cands[j];
	else {
		cd.pos = 0; cd.sample = 0; cd.hdr_bytes = 0; cd.blocksize = 0; cd.assignment = 0; cd.stream = 0; cd.status = 0; cd.end_byte = 0;
	}
	BitRd r;
	r.w = flac_words;
	r.nwords = (P.flac_bytes + 3) / 4;
	r.pos = (cd.pos + cd.hdr_bytes) * 8;
	r.end = live ? (streams[cd.stream].byte_offset + streams[cd.stream].byte_length) * 8 : 0;
	r.err = false;
	const uint32_t bsf = cd.blocksize;
	uint32_t bs_w = bsf;
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		const uint32_t t = __shfl_xor(bs_w, o, 64);
		bs_w = t > bs_w ? t : bs_w;
	}
	int32_t *frame_tmp = tmp + (size_t)blockIdx.x * P.channels * P.max_blocksize * DEC_LANES;

	for (uint32_t c = 0; c < P.channels; c++) {
		int32_t *plane = frame_tmp + (size_t)c * P.max_blocksize * DEC_LANES;
		// ---- subframe header (stream_decoder.c:2450-2518) ----
		uint32_t sbps = P.bps;
		if ((cd.assignment == 1 && c == 1) || (cd.assignment == 2 && c == 0) || (cd.assignment == 3 && c == 1))
			sbps++;
		uint32_t kind = 0;  // 0 constant, 1 verbatim, 2 fixed, 3 lpc
		uint32_t order = 0, wasted = 0, po = 0, plen = 4, pesc = 15, psize = 0;
		int shift = 0;
		bool lane_wide = false;
		int32_t constant = 0;
		const bool ok_in = live && !r.err;
		if (ok_in) {
			uint32_t x = br_get(r, 8);
			const uint32_t has_wasted = x & 1;
			x &= 0xfe;
			if (has_wasted) {
				wasted = br_unary(r) + 1;
				if (wasted >= sbps)
					r.err = true;
				else
					sbps -= wasted;
			}
			if (x & 0x80) r.err = true;
			else if (x == 0) kind = 0;
			else if (x == 2) kind = 1;
			else if (x < 16) r.err = true;
			else if (x <= 24) {
				kind = 2;
				order = (x >> 1) & 7;
				if (order > 4) r.err = true;
			} else if (x < 64) r.err = true;
			else {
				kind = 3;
				order = ((x >> 1) & 31) + 1;
			}
			if (order > bsf)
				r.err = true;
		}
		if (ok_in && !r.err) {
			if (kind == 0)
				constant = br_get_signed(r, sbps);
			if (kind >= 2) {
				for (uint32_t i = 0; i < order; i++)
					plane[(size_t)i * DEC_LANES + lane] = br_get_signed(r, sbps);
				if (kind == 3) {
					uint32_t prec = br_get(r, 4);
					if (prec == 15)
						r.err = true;  // stream_decoder.c:2622
					prec++;
					shift = br_get_signed(r, 5);
					if (shift < 0)
						shift = 0;  // the reference's shift by a negative count is undefined; it never writes one
					for (uint32_t i = 0; i < order; i++)
						L.q[i][lane] = br_get_signed(r, prec);
					lane_wide = !(sbps + prec + d_ilog2(order) <= 32);  // stream_decoder.c:2676
				} else {
					for (uint32_t i = 0; i < order; i++)
						L.q[i][lane] = c_fixed_q[order][i];
				}
				const uint32_t method = br_get(r, 2);
				if (method > 1)
					r.err = true;
				if (method == 1) {
					plen = 5;
					pesc = 31;
				}
				po = br_get(r, 4);
				// stream_decoder.c:2721-2739
				psize = po > 0 ? bsf >> po : bsf - order;
				if (po > 0 && (psize < order || (psize << po) != bsf))
					r.err = true;
			}
		}
		// ---- samples: verbatim words or Rice-coded residual (stream_decoder.c:2692,2715) ----
		uint32_t next_part = order;   // sample index where the next partition parameter is read
		uint32_t k = 0, raw_bits = 0;
		bool escape = false;
		for (uint32_t i = 0; i < bs_w; i++) {
			if (!(live && !r.err && i < bsf))
				continue;
			int32_t v;
			if (kind == 0)
				v = constant;
			else if (kind == 1)
				v = br_get_signed(r, sbps);
			else {
				if (i < order)
					continue;  // warm-up already stored
				if (i == next_part) {
					k = br_get(r, plen);
					escape = k >= pesc;
					if (escape)
						raw_bits = br_get(r, 5);
					next_part = (po > 0) ? ((i / psize) + 1) * psize : bsf;
				}
				if (escape)
					v = br_get_signed(r, raw_bits);
				else {
					// FLAC__bitreader_read_rice_signed_block (bitreader.c:775)
					const uint32_t msbs = br_unary(r);
					const uint32_t lsbs = br_get(r, k);
					const uint32_t u = (msbs << k) | lsbs;
					v = (int32_t)(u >> 1) ^ -(int32_t)(u & 1);
				}
				if (r.pos > r.end)
					r.err = true;
			}
			plane[(size_t)i * DEC_LANES + lane] = v;
		}
		// ---- predictor restore, dispatched on what the wave as a whole needs ----
		const bool predictive = kind >= 2 && live && !r.err;
		uint32_t need = predictive ? order : 0;
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) {
			const uint32_t t = __shfl_xor(need, o, 64);
			need = t > need ? t : need;
		}
		const bool any_wide = __any(predictive && lane_wide);
		const bool all24 = __all(!predictive || sbps <= 24);
		const bool run = live && !r.err;
#define RESTORE(M) \
	do { \
		if (any_wide) restore_plane<M, true, false>(plane, L, lane, bs_w, bsf, order, shift, lane_wide, predictive, wasted, run); \
		else if (all24) restore_plane<M, false, true>(plane, L, lane, bs_w, bsf, order, shift, lane_wide, predictive, wasted, run); \
		else restore_plane<M, false, false>(plane, L, lane, bs_w, bsf, order, shift, lane_wide, predictive, wasted, run); \
	} while (0)
		if (need == 0) RESTORE(0);
		else if (need <= 4) RESTORE(4);
		else if (need <= 8) RESTORE(8);
		else if (need <= 12) RESTORE(12);
		else if (need <= 16) RESTORE(16);
		else RESTORE(32);
#undef RESTORE
	}
	if (live) {
		// read_zero_padding_ (stream_decoder.c:2773): the pad bits must be zero
		if (!r.err && (r.pos & 7)) {
			const uint32_t n = 8 - ((uint32_t)r.pos & 7);
			if (br_get(r, n) != 0)
				r.err = true;
		}
		const uint64_t end_byte = (r.pos >> 3) + 2;
		if (end_byte * 8 > r.end)
			r.err = true;
		cands[j].end_byte = end_byte;
		// a parse that ran off the end of the input is a cut-off frame, not a malformed one
		const bool cut = r.pos > r.end || end_byte * 8 > r.end;
		cands[j].status = cut ? 4u : r.err ? 2u : 1u;
	}
}

// ---------------------------------------------------------------------------------------------
// k_dec_crc: CRC-16 of every structurally good frame, one workgroup per frame
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t dgf16_mul(uint32_t a, uint32_t b)
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

__global__ __launch_bounds__(256) void k_dec_crc(const uint8_t *d, FaFrameCand *cands, const uint32_t *ncand_ptr)
{
	__shared__ uint32_t crcs[256];
	__shared__ uint16_t table[256];
	const uint32_t tid = threadIdx.x;
	table[tid] = c_dec_crc16[tid];
	for (uint32_t f = blockIdx.x; f < ncand_ptr[0]; f += gridDim.x) {
		__syncthreads();
		const FaFrameCand cd = cands[f];
		if (cd.status != 1)
			continue;
		const uint32_t nbytes = (uint32_t)(cd.end_byte - cd.pos) - 2;
		const uint8_t *src = d + cd.pos;
		const uint32_t R = (nbytes + 255) / 256;
		uint32_t crc = 0;
		const int64_t start = (int64_t)nbytes - (int64_t)(256 - tid) * R;
		for (uint32_t i = 0; i < R; i++) {
			const int64_t b = start + i;
			if (b >= 0)
				crc = ((crc << 8) & 0xffff) ^ table[(crc >> 8) ^ src[b]];
		}
		crcs[tid] = crc;
		__syncthreads();
		uint32_t m = 1;
		for (uint32_t jj = 0, n = R; n; jj++, n >>= 1)
			if (n & 1)
				m = dgf16_mul(m, c_dec_pow16[jj]);
		for (uint32_t Lv = 0; Lv < 8; Lv++) {
			const uint32_t mask = (2u << Lv) - 1;
			if ((tid & mask) == mask)
				crcs[tid] = dgf16_mul(crcs[tid - (1u << Lv)], m) ^ crcs[tid];
			m = dgf16_mul(m, m);
			__syncthreads();
		}
		if (tid == 0) {
			const uint32_t want = ((uint32_t)src[nbytes] << 8) | src[nbytes + 1];
			if (crcs[255] != want)
				cands[f].status = 3;
		}
	}
}

// ---------------------------------------------------------------------------------------------
// k_dec_output
// ---------------------------------------------------------------------------------------------
#define OUT_LDS_BYTES 32768

__global__ __launch_bounds__(256) void k_dec_output(FaDecParams P, const FaStream *streams, const FaFrameCand *cands, const uint32_t *ncand_ptr,
                                                    const int32_t *tmp, void *out, uint32_t tile_samples)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
	const uint32_t ncand = ncand_ptr[0];
	const uint32_t g = blockIdx.x;
	if (g * DEC_LANES >= ncand)
		return;
	const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const uint32_t i0 = blockIdx.y * tile_samples;
	const uint32_t ch = P.channels, sb = P.out_sample_bytes;
	const uint32_t j = g * DEC_LANES + lane;
	const bool live = j < ncand;
	FaFrameCand cd;
	cd.blocksize = 0;
	cd.status = 0;
	cd.assignment = 0;
	if (live)
		cd = cands[j];
	const int32_t *frame_tmp = tmp + (size_t)g * ch * P.max_blocksize * DEC_LANES;
	const uint32_t row = tile_samples * ch;  // elements per frame in the tile
	// phase 1: lane = frame; each wave takes every 4th sample of the tile
	for (uint32_t ii = wv; ii < tile_samples; ii += 4) {
		const uint32_t i = i0 + ii;
		if (!(live && cd.status == 1 && i < cd.blocksize))
			continue;
		if (ch == 2 && cd.assignment != 0) {
			int32_t a = frame_tmp[(size_t)i * DEC_LANES + lane];
			int32_t b = frame_tmp[((size_t)P.max_blocksize + i) * DEC_LANES + lane];
			// stream_decoder.c:2067-2102
			if (cd.assignment == 1)
				b = a - b;
			else if (cd.assignment == 2)
				a = a + b;
			else {
				int32_t mid = (int32_t)((uint32_t)a << 1);
				mid |= (b & 1);
				a = (mid + b) >> 1;
				b = (mid - b) >> 1;
			}
			if (sb == 2) {
				((int16_t *)lds)[lane * row + ii * 2] = (int16_t)a;
				((int16_t *)lds)[lane * row + ii * 2 + 1] = (int16_t)b;
			} else {
				((int32_t *)lds)[lane * row + ii * 2] = a;
				((int32_t *)lds)[lane * row + ii * 2 + 1] = b;
			}
		} else {
			for (uint32_t c = 0; c < ch; c++) {
				const int32_t v = frame_tmp[((size_t)c * P.max_blocksize + i) * DEC_LANES + lane];
				if (sb == 2)
					((int16_t *)lds)[lane * row + ii * ch + c] = (int16_t)v;
				else
					((int32_t *)lds)[lane * row + ii * ch + c] = v;
			}
		}
	}
	__syncthreads();
	// phase 2: each wave writes whole frames' runs, lanes along the run
	for (uint32_t fl = wv; fl < DEC_LANES; fl += 4) {
		const uint32_t jf = g * DEC_LANES + fl;
		if (jf >= ncand)
			break;
		const FaFrameCand cf = cands[jf];
		if (cf.status != 1 || i0 >= cf.blocksize)
			continue;
		const FaStream st = streams[cf.stream];
		// FLACAMD_FIRST_SAMPLE_AUTO (single stream): the first header found anchors the output
		const uint64_t first = st.first_sample == ~0ull ? cands[0].sample : st.first_sample;
		if (cf.sample < first)
			continue;
		const uint64_t rel = cf.sample - first + i0;
		uint32_t nvalid = cf.blocksize - i0;
		if (nvalid > tile_samples)
			nvalid = tile_samples;
		if (st.total_samples && rel + nvalid > st.total_samples)
			continue;
		if (st.sample_offset + rel + nvalid > P.out_capacity_wide)
			continue;
		const uint64_t obase = (st.sample_offset + rel) * ch;
		const uint32_t nel = nvalid * ch;
		for (uint32_t e = lane; e < nel; e += 64) {
			if (sb == 2)
				((int16_t *)out)[obase + e] = ((const int16_t *)lds)[fl * row + e];
			else
				((int32_t *)out)[obase + e] = ((const int32_t *)lds)[fl * row + e];
		}
	}
}

__global__ void k_dec_summary(const FaFrameCand *cands, const uint32_t *ncand_ptr, unsigned long long *result)
{
	// result[0] good frames, [1] wide samples in good frames, [2] bad candidates
	const uint32_t n = ncand_ptr[0];
	unsigned long long good = 0, wide = 0, bad = 0;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		if (cands[i].status == 1) {
			good++;
			wide += cands[i].blocksize;
		} else
			bad++;
	}
	if (good) atomicAdd(&result[0], good);
	if (wide) atomicAdd(&result[1], wide);
	if (bad) atomicAdd(&result[2], bad);
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
uint32_t fa_dec_sync_blocks(uint64_t flac_bytes)
{
	const uint64_t per = (uint64_t)SYNC_THREADS * SYNC_BYTES_PER_THREAD;
	return (uint32_t)((flac_bytes + per - 1) / per);
}

void fa_dec_launch_sync(const FaDecParams &P, const uint8_t *flac, const FaStream *streams, uint32_t *block_counts, uint64_t *block_offsets,
                        FaFrameCand *cands, uint32_t *ncand, hipStream_t st)
{
	const uint32_t nb = fa_dec_sync_blocks(P.flac_bytes);
	hipLaunchKernelGGL(k_dec_sync<false>, dim3(nb), dim3(SYNC_THREADS), 0, st, P, flac, streams, block_counts, block_offsets, cands, ncand);
	hipLaunchKernelGGL(k_dec_scan, dim3(1), dim3(1024), 0, st, block_counts, nb, block_offsets);
	hipLaunchKernelGGL(k_dec_sync<true>, dim3(nb), dim3(SYNC_THREADS), 0, st, P, flac, streams, block_counts, block_offsets, cands, ncand);
}

void fa_dec_launch_frames(const FaDecParams &P, const uint8_t *flac, const FaStream *streams, FaFrameCand *cands, const uint32_t *ncand,
                          int32_t *tmp, hipStream_t st)
{
	const uint32_t groups = (P.max_frames + DEC_LANES - 1) / DEC_LANES;
	hipLaunchKernelGGL(k_dec_frames, dim3(groups), dim3(DEC_LANES), 0, st, P, (const uint32_t *)flac, streams, cands, ncand, tmp);
}

void fa_dec_launch_crc(const FaDecParams &P, const uint8_t *flac, FaFrameCand *cands, const uint32_t *ncand, hipStream_t st)
{
	uint32_t blocks = P.max_frames < 8192 ? P.max_frames : 8192;
	if (blocks == 0)
		blocks = 1;
	hipLaunchKernelGGL(k_dec_crc, dim3(blocks), dim3(256), 0, st, flac, cands, ncand);
}

void fa_dec_launch_output(const FaDecParams &P, const FaStream *streams, const FaFrameCand *cands, const uint32_t *ncand, const int32_t *tmp,
                          void *out, unsigned long long *result, hipStream_t st)
{
	uint32_t tile = OUT_LDS_BYTES / (DEC_LANES * P.channels * P.out_sample_bytes);
	if (tile > 64)
		tile = 64;
	const uint32_t groups = (P.max_frames + DEC_LANES - 1) / DEC_LANES;
	const uint32_t tiles = (P.max_blocksize + tile - 1) / tile;
	const size_t lds = (size_t)DEC_LANES * tile * P.channels * P.out_sample_bytes;
	hipLaunchKernelGGL(k_dec_output, dim3(groups, tiles), dim3(256), lds, st, P, streams, cands, ncand, tmp, out, tile);
	hipLaunchKernelGGL(k_dec_summary, dim3(64), dim3(256), 0, st, cands, ncand, result);
}
