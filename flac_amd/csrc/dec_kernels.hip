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
__constant__ uint16_t c_dec_crc16x4[4][256];  // table k: CRC-16 of a byte followed by k zero bytes
// k_dec_crc works in segments of 64 runs of 8 << c words; lane l's run is followed by 63 - l more
// runs, so its CRC is multiplied by c_dec_runpow[c][l] = x^(8 * 4 * (8 << c) * (63 - l)); a whole
// segment further on multiplies by c_dec_segpow[c]
__constant__ uint16_t c_dec_runpow[4][64];
__constant__ uint16_t c_dec_segpow[4];

hipError_t fa_dec_upload_tables(const uint16_t *t16, const uint16_t *pow16, const uint8_t *t8)
{
	hipError_t rc = hipMemcpyToSymbol(HIP_SYMBOL(c_dec_crc16), t16, 512);
	if (rc == hipSuccess)
		rc = hipMemcpyToSymbol(HIP_SYMBOL(c_dec_pow16), pow16, 64);
	if (rc == hipSuccess)
		rc = hipMemcpyToSymbol(HIP_SYMBOL(c_dec_crc8), t8, 256);
	if (rc != hipSuccess)
		return rc;
	static uint16_t x4[4][256];
	for (int e = 0; e < 256; e++) {
		uint32_t t = t16[e];
		x4[0][e] = (uint16_t)t;
		for (int k = 1; k < 4; k++) {
			t = ((t << 8) & 0xffff) ^ t16[t >> 8];
			x4[k][e] = (uint16_t)t;
		}
	}
	rc = hipMemcpyToSymbol(HIP_SYMBOL(c_dec_crc16x4), x4, sizeof(x4));
	if (rc != hipSuccess)
		return rc;
	// powers of x modulo x^16 + x^15 + x^2 + 1
	auto mul = [](uint32_t a, uint32_t b) {
		uint32_t r = 0;
		for (int i = 0; i < 16; i++) {
			if (b & 1)
				r ^= a;
			b >>= 1;
			a <<= 1;
			if (a & 0x10000)
				a ^= 0x18005;
		}
		return r;
	};
	auto xpow8 = [&](uint64_t nbytes) {
		uint32_t r = 1;
		for (int j = 0; nbytes; j++, nbytes >>= 1)
			if (nbytes & 1)
				r = mul(r, pow16[j]);
		return r;
	};
	static uint16_t runpow[4][64], segpow[4];
	for (int c = 0; c < 4; c++) {
		const uint64_t run_bytes = 4ull * (8u << c);
		for (int l = 0; l < 64; l++)
			runpow[c][l] = (uint16_t)xpow8(run_bytes * (uint64_t)(63 - l));
		segpow[c] = (uint16_t)xpow8(run_bytes * 64);
	}
	rc = hipMemcpyToSymbol(HIP_SYMBOL(c_dec_runpow), runpow, sizeof(runpow));
	if (rc != hipSuccess)
		return rc;
	return hipMemcpyToSymbol(HIP_SYMBOL(c_dec_segpow), segpow, sizeof(segpow));
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

// The counting pass: every byte position is tested for a frame header; what comes out is the number of
// headers per 4 KiB block and, per wave, the mask of lanes (16-byte runs) that hold one.
__global__ __launch_bounds__(SYNC_THREADS) void k_dec_sync(FaDecParams P, const uint8_t *d, const FaStream *streams, uint32_t *block_counts)
{
	__shared__ uint32_t wsum[4];
	const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const uint64_t p0 = ((uint64_t)blockIdx.x * SYNC_THREADS + tid) * SYNC_BYTES_PER_THREAD;
	uint32_t n = 0;  // headers in this run (more than two real ones in 16 bytes cannot happen)
	// the lane masks lie behind the counts, four 64-bit words per block
	uint64_t *masks = (uint64_t *)(block_counts + ((gridDim.x + 1) & ~1u)) + (size_t)blockIdx.x * (SYNC_THREADS / 64);
	// the thread's 16 bytes and the one after them, as five words; only positions that hold the
	// two sync bytes are looked at further
	uint32_t w[SYNC_BYTES_PER_THREAD / 4 + 1];
	{
		const uint32_t *dw = (const uint32_t *)d;
		const uint64_t nwords = (P.flac_bytes + 3) / 4;
#pragma unroll
		for (uint32_t k = 0; k < SYNC_BYTES_PER_THREAD / 4 + 1; k++) {
			const uint64_t idx = p0 / 4 + k;
			w[k] = idx < nwords ? dw[idx] : 0u;
		}
	}
	uint32_t hits = 0;
	// most 16-byte runs hold no 0xff byte at all: one test per word (a byte of w is 0xff where the byte of ~w is zero)
	uint32_t any_ff = 0;
#pragma unroll
	for (uint32_t k = 0; k < SYNC_BYTES_PER_THREAD / 4; k++)
		any_ff |= (~w[k] - 0x01010101u) & w[k] & 0x80808080u;
	if (any_ff) {
#pragma unroll
		for (uint32_t i = 0; i < SYNC_BYTES_PER_THREAD; i++) {
			const uint32_t b0 = (w[i / 4] >> (8 * (i % 4))) & 0xff;
			const uint32_t b1 = (w[(i + 1) / 4] >> (8 * ((i + 1) % 4))) & 0xff;
			if (b0 == 0xff && (b1 & 0xfe) == 0xf8)
				hits |= 1u << i;
		}
	}
	while (hits) {
		const uint32_t i = (uint32_t)__ffs((int)hits) - 1u;
		hits &= hits - 1;
		const uint64_t p = p0 + i;
		if (p >= P.flac_bytes)
			break;
		FaFrameCand c;
		if (parse_frame_header(d, p, P, streams, c))
			n++;
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
	const uint64_t m = __ballot(n > 0);
	if (lane == 0)
		masks[wv] = m;
	if (tid == SYNC_THREADS - 1)
		block_counts[blockIdx.x] = off + inc;
}

// exclusive scan of the per-block counts (single block, eight counts per thread and round)
__global__ __launch_bounds__(1024) void k_dec_scan(const uint32_t *counts, uint32_t n, uint64_t *offsets)
{
	__shared__ uint64_t wsum[16];
	__shared__ uint64_t carry;
	const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	if (tid == 0)
		carry = 0;
	__syncthreads();
	for (uint32_t base = 0; base < n; base += 8 * 1024) {
		const uint32_t i0 = base + tid * 8;
		uint32_t v[8];
		uint64_t mine = 0;
#pragma unroll
		for (uint32_t k = 0; k < 8; k++) {
			v[k] = i0 + k < n ? counts[i0 + k] : 0u;
			mine += v[k];
		}
		uint64_t inc = mine;
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
		uint64_t run = off + inc - mine;
#pragma unroll
		for (uint32_t k = 0; k < 8; k++) {
			if (i0 + k < n)
				offsets[i0 + k] = run;
			run += v[k];
		}
		__syncthreads();
		if (tid == 1023)
			carry = off + inc;
		__syncthreads();
	}
}

// The emitting pass: one thread per 4 KiB block of the counting pass.  It walks the block's lane
// masks in byte order and parses the (few) flagged 16-byte runs again, which puts the headers into
// the table in the order of their positions.
__global__ __launch_bounds__(256) void k_dec_sync_emit(FaDecParams P, const uint8_t *d, const FaStream *streams, const uint32_t *block_counts,
                                                      const uint64_t *block_offsets, FaFrameCand *cands, uint32_t *ncand_out, uint32_t nb)
{
	const uint32_t b = blockIdx.x * 256 + threadIdx.x;
	if (b >= nb)
		return;
	const uint32_t count = block_counts[b];
	uint64_t at = block_offsets[b];
	if (b == nb - 1) {
		const uint64_t total = at + count;
		ncand_out[0] = total > P.max_frames ? P.max_frames : (uint32_t)total;
		ncand_out[1] = total > P.max_frames ? 1u : 0u;  // overflow flag
	}
	if (count == 0)
		return;
	const uint64_t *masks = (const uint64_t *)(block_counts + ((nb + 1) & ~1u)) + (size_t)b * (SYNC_THREADS / 64);
	for (uint32_t wv = 0; wv < SYNC_THREADS / 64; wv++) {
		uint64_t m = masks[wv];
		while (m) {
			const uint32_t lane = (uint32_t)__ffsll((unsigned long long)m) - 1u;
			m &= m - 1;
			const uint64_t p0 = ((uint64_t)b * SYNC_THREADS + wv * 64 + lane) * SYNC_BYTES_PER_THREAD;
			// the run and the byte after it in one go, as the counting pass reads them
			const uint64_t nwords = (P.flac_bytes + 3) / 4;
			uint32_t w[SYNC_BYTES_PER_THREAD / 4 + 1];
#pragma unroll
			for (uint32_t k = 0; k < SYNC_BYTES_PER_THREAD / 4 + 1; k++) {
				const uint64_t idx = p0 / 4 + k;
				w[k] = idx < nwords ? ((const uint32_t *)d)[idx] : 0u;
			}
			uint32_t hits = 0;
#pragma unroll
			for (uint32_t i = 0; i < SYNC_BYTES_PER_THREAD; i++) {
				const uint32_t b0 = (w[i / 4] >> (8 * (i % 4))) & 0xff;
				const uint32_t b1 = (w[(i + 1) / 4] >> (8 * ((i + 1) % 4))) & 0xff;
				if (b0 == 0xff && (b1 & 0xfe) == 0xf8)
					hits |= 1u << i;
			}
			uint32_t n = 0;
			while (hits && n < 2) {
				const uint32_t i = (uint32_t)__ffs((int)hits) - 1u;
				hits &= hits - 1;
				const uint64_t p = p0 + i;
				if (p >= P.flac_bytes)
					break;
				FaFrameCand c;
				if (parse_frame_header(d, p, P, streams, c)) {
					if (at < P.max_frames)
						cands[at] = c;
					at++;
					n++;
				}
			}
		}
	}
}

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
#define RING_WORDS 128      // words per lane in LDS: two chunks
#define DEC_BLOCK 16        // samples decoded between refill / flush points

struct BitRd {
	const uint32_t *wg;  // the frame's first word in global memory (word index wbase of the input)
	uint64_t wbase;      // index of that word in the input
	uint32_t cur;        // the word being read: its last `rem` bits are unread
	uint32_t nxt;        // the word after it, all unread (both already byte-swapped)
	uint32_t rem;        // 0..31
	uint32_t wrel;       // index, relative to wbase, of the word after nxt
	uint32_t nrel;       // words from wbase to the end of the input (saturated)
	uint32_t limit;      // words [0, limit) have been in the ring; the last RING_WORDS of them still are
	uint32_t bend;       // first bit past the stream on the same scale (saturated)
	const uint32_t *ring;  // this lane's RING_WORDS words of LDS
	bool err;
};

// Bits consumed, counted from the start of word wbase: cur and nxt are words wrel-2 and wrel-1 of the frame, of which
// rem + 32 bits are still unread.
__device__ __forceinline__ uint32_t br_pos(const BitRd &r) { return 32u * (r.wrel - 1u) - r.rem; }

// The refill protocol keeps limit - RING_WORDS <= wrel, so one compare tells whether the ring has
// the word; a lane that has outrun its ring reads global memory itself.
__device__ __forceinline__ uint32_t br_fetch(BitRd &r)
{
	uint32_t v;
	if (r.wrel < r.limit)
		v = r.ring[r.wrel & (RING_WORDS - 1)];   // the ring holds the words already byte-swapped
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
template <int MAXLW>
struct DecLds {
	static constexpr int maxlw = MAXLW;
	int32_t q[FA_MAX_ORDER][MAXLW];          // predictor coefficients per lane
	int32_t w[FA_MAX_ORDER][MAXLW];          // warm-up samples per lane
	uint32_t ring[MAXLW][RING_WORDS];        // input words per lane
	int32_t out[2 * DEC_BLOCK * MAXLW];      // a block of decoded samples, [sample][lane]; second half: spare
};

// The wave-level input pump.  Up to four refills are in flight (one word per thread each); they are
// put into the rings at the next service point, a block of samples later.  (Four, because two cannot feed
// sixteen lanes of 24-bit frames, 7.5 words per lane and block: 2.94 -> 1.96 ms for 14 063 such frames.)
#define PUMP_SLOTS 4
struct Pump {
	uint32_t data[PUMP_SLOTS];
	int lane[PUMP_SLOTS];      // whose ring the word goes to, -1 = slot free
	uint32_t half[PUMP_SLOTS]; // which half of that ring
};

// what a lane knows about its current subframe once the subframe header is parsed
struct SubHdr {
	uint32_t kind;      // 0 constant, 1 verbatim, 2 fixed, 3 lpc
	uint32_t order, wasted, sbps;
	uint32_t po, plen, pesc, psize;
	int shift;
	bool lane_wide;
	int32_t constant;
};

__device__ __forceinline__ uint32_t wave_value(uint32_t v, int lane_id) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane_id); }

// Service point (every lane of the wave takes part): land the refills in flight, then start
// refills for up to two lanes that have moved into the newest chunk of their ring.
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
	const bool want = live && !r.err && r.wrel + RING_CHUNK >= r.limit && r.limit < r.nrel;
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
			pm.half[sl] = (lim / RING_CHUNK) & 1;
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
template <int M, bool WIDE, bool MUL24, bool SKIP = false, typename LDS>
__device__ __forceinline__ void decode_samples(BitRd &r, Pump &pm, const SubHdr &h, LDS &L, int32_t *plane, uint32_t LW, uint32_t tid, uint32_t bs_w,
                                               uint32_t bsf, bool live, uint64_t nwords)
{
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
	// lanes that carry no frame write their block samples to a spare area
	int32_t *const slot = tid < LW ? &L.out[tid] : &L.out[DEC_BLOCK * LDS::maxlw];
	for (uint32_t i0 = 0; i0 < bs_w; i0 += DEC_BLOCK) {
		pump_service(pm, r, L, tid, live, nwords);
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
			n2 = r.ring[r.wrel & (RING_WORDS - 1)];
			n3 = r.ring[(r.wrel + 1) & (RING_WORDS - 1)];
		}
		// unrolled in full: the history needs no moves at all, and the compiler schedules across samples
		// (unrolled by four: 1.70 ms for the benchmark's frames, in full: 1.63)
#pragma unroll
		for (uint32_t s = 0; s < DEC_BLOCK; s++) {
			const uint32_t i = i0 + s;
			// the word a step into the next word at the end of this sample would bring in
			const uint32_t ring_word = SKIP ? n2 : r.ring[r.wrel & (RING_WORDS - 1)];
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
					n3 = r.ring[(r.wrel + 1) & (RING_WORDS - 1)];
				}
			}
			if (!fast && i < work_end) {
				if (h.kind >= 2) {
					if (i < h.order)
						res = L.w[i][lane];
					else {
						if (i == next_part) {
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
								part_end = h.po > 0 ? part_end + h.psize : bsf;
							} while (part_end <= i && !r.err && br_pos(r) <= r.bend);
							kp1 = k + 1;
							next_part = part_end;
							seg_end = escape ? 0 : next_part;
						}
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
					n2 = r.ring[r.wrel & (RING_WORDS - 1)];
					n3 = r.ring[(r.wrel + 1) & (RING_WORDS - 1)];
				}
			}
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
			if (!SKIP)
				slot[s * LW] = (int32_t)((uint32_t)v << h.wasted);
		}
		if (!SKIP) {
			// the block goes out with all 64 threads along memory order
			const uint32_t nblk = (i0 + DEC_BLOCK < bs_w ? DEC_BLOCK : bs_w - i0) * LW;
			for (uint32_t e = tid; e < nblk; e += DEC_LANES)
				plane[(size_t)i0 * LW + e] = L.out[e];
		}
	}
}

// One lane per frame; a wave carries LW <= 16 frames in its first lanes, the other lanes only help
// moving data.  A wave's pace is set by its longest dependent chain, not by how many of its lanes
// decode, so a batch is spread over as many waves as the chip can run at once.
//
// The chain is a frame's samples one after the other, and a subframe can only be started where the one
// before it ends.  For batches that leave the chip room, the chain is cut: DEC_SKIP walks the subframes
// of every channel but the last without decoding them (a symbol's length is all it takes: a quarter of
// the instructions) and leaves the bit position at which each subframe starts; DEC_SUB then runs one lane
// per (frame, channel), a wave taking one channel of LW frames, so the longest chain is one subframe.
// A frame the walk found malformed gets its status there and is left alone by DEC_SUB; the lane of the
// last channel finishes the frame (padding, end, status) as the single pass does.
#define DEC_WHOLE 0
#define DEC_SKIP 1
#define DEC_SUB 2
#define DEC_NO_START 0xffffffffffffffffull

template <int MODE, int MAXLW>
__global__ __launch_bounds__(DEC_LANES) void k_dec_frames(FaDecParams P, const uint32_t *flac_words, const FaStream *streams,
                                                          FaFrameCand *cands, const uint32_t *ncand_ptr, int32_t *tmp, uint32_t LW, uint64_t *sub_start)
{
	__shared__ DecLds<MAXLW> L;
	const uint32_t tid = threadIdx.x;
	const uint32_t ncand = ncand_ptr[0];
	const uint32_t grp = MODE == DEC_SUB ? blockIdx.x / P.channels : blockIdx.x;
	const uint32_t only_c = MODE == DEC_SUB ? blockIdx.x % P.channels : 0;
	if (grp * LW >= ncand)
		return;
	const uint32_t j = grp * LW + tid;
	bool live = tid < LW && j < ncand;
	const uint32_t lane = tid < (uint32_t)MAXLW ? tid : 0;
	FaFrameCand cd;
	if (live)
		cd = cands[j];
	else {
		cd.pos = 0; cd.sample = 0; cd.hdr_bytes = 0; cd.blocksize = 0; cd.assignment = 0; cd.stream = 0; cd.status = 0; cd.end_byte = 0;
	}
	// where the lane starts reading, in bits from the start of the input
	uint64_t start_bit = (cd.pos + cd.hdr_bytes) * 8;
	if (MODE == DEC_SUB) {
		if (live) {
			start_bit = sub_start[(size_t)j * P.channels + only_c];
			if (start_bit == DEC_NO_START) {
				live = false;
				start_bit = 0;
			}
		} else
			start_bit = 0;
	}
	const uint64_t nwords = (P.flac_bytes + 3) / 4;
	BitRd r;
	r.wbase = MODE == DEC_SUB ? start_bit >> 5 : cd.pos >> 2;
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
	int32_t *frame_tmp = tmp + (size_t)grp * P.channels * P.max_blocksize * LW;
	bool reserved_code = false;  // the parse stopped at a reserved subframe type or residual coding method

	const uint32_t c_first = MODE == DEC_SUB ? only_c : 0;
	const uint32_t c_end = MODE == DEC_SUB ? only_c + 1 : MODE == DEC_SKIP ? P.channels - 1 : P.channels;
	for (uint32_t c = c_first; c < c_end; c++) {
		int32_t *plane = frame_tmp + (size_t)c * P.max_blocksize * LW;
		if (MODE == DEC_SKIP && live)
			sub_start[(size_t)j * P.channels + c] = r.err ? DEC_NO_START : r.wbase * 32 + br_pos(r);
		pump_service(pm, r, L, tid, live, nwords);
		// ---- subframe header (stream_decoder.c:2450-2518) ----
		SubHdr h;
		h.kind = 0; h.order = 0; h.wasted = 0; h.po = 0; h.plen = 4; h.pesc = 15; h.psize = 0; h.shift = 0; h.lane_wide = false; h.constant = 0;
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
#define DECODE(M) \
	do { \
		if (any_wide) decode_samples<M, true, false>(r, pm, h, L, plane, LW, tid, bs_w, bsf, run, nwords); \
		else if (all24) decode_samples<M, false, true>(r, pm, h, L, plane, LW, tid, bs_w, bsf, run, nwords); \
		else decode_samples<M, false, false>(r, pm, h, L, plane, LW, tid, bs_w, bsf, run, nwords); \
	} while (0)
		if (MODE == DEC_SKIP) {
			decode_samples<0, false, false, true>(r, pm, h, L, plane, LW, tid, bs_w, bsf, run, nwords);
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
	if (MODE == DEC_SUB && only_c + 1 != P.channels)
		return;   // (a subframe the walk got through cannot fail here: the same bits, the same checks)
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
		cands[j].reserved = (!past && r.err && reserved_code) ? 1u : 0u;
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

#define CRC_SEG_WORDS 4096  // 64 lanes x 64 words: the largest segment

// One wave per frame.  The frame is cut into segments of 64 equal runs, the first segment padded
// with zeros IN FRONT (leading zeros do not change a CRC that starts at 0), so every run length and
// every distance to the end of a segment is one of a few fixed sizes whose powers of x come from
// tables.  A segment is brought into LDS with coalesced word loads; each lane takes its run through
// four byte tables at once (table k: the CRC of a byte followed by k zero bytes) and multiplies the
// result by x^(8 * bytes that follow in the segment); the 64 products are XORed together (the CRC is
// linear), and segments are chained by crc(A || B) = crc(A) * x^(8 |B|) + crc(B).
__global__ __launch_bounds__(64) void k_dec_crc(const uint8_t *d, FaFrameCand *cands, const uint32_t *ncand_ptr)
{
	__shared__ uint32_t seg[CRC_SEG_WORDS];
	__shared__ uint16_t table[4][256];
	const uint32_t lane = threadIdx.x;
	for (uint32_t e = lane; e < 4 * 256 / 2; e += 64)
		((uint32_t *)&table[0][0])[e] = ((const uint32_t *)&c_dec_crc16x4[0][0])[e];
	const uint32_t *dw = (const uint32_t *)d;
	for (uint32_t f = blockIdx.x; f < ncand_ptr[0]; f += gridDim.x) {
		const FaFrameCand cd = cands[f];
		if (cd.status != 1)
			continue;
		const uint32_t nbytes = (uint32_t)(cd.end_byte - cd.pos) - 2;
		// the smallest run size that covers the frame in one segment, else the largest
		uint32_t c = 0;
		while (c < 3 && 64u * 4u * (8u << c) < nbytes)
			c++;
		const uint32_t run = 8u << c, seg_bytes = 256u * run;
		const uint32_t nseg = (nbytes + seg_bytes - 1) / seg_bytes;
		const uint32_t len0 = nbytes - (nseg - 1) * seg_bytes;  // bytes in the first segment
		const uint32_t my_pow = c_dec_runpow[c][lane], seg_pow = c_dec_segpow[c];
		uint32_t total = 0;
		for (uint32_t sg = 0; sg < nseg; sg++) {
			const uint32_t len = sg == 0 ? len0 : seg_bytes;
			const uint32_t pad = seg_bytes - len;  // zero bytes in front
			// LDS word j holds input bytes [src0 + 4 j, +4), src0 = where the segment would start
			// if it were full
			const int64_t src0 = (int64_t)cd.pos + (sg == 0 ? 0 : (int64_t)len0 + (int64_t)(sg - 1) * seg_bytes) - (int64_t)pad;
			const uint32_t sh = (uint32_t)(src0 & 3) * 8;
			const int64_t w0 = src0 >> 2;
			__syncthreads();
			// eight words per lane at a time, all sixteen loads issued before the first is used
			for (uint32_t j0 = lane; j0 < 64 * run; j0 += 64 * 8) {
				uint32_t lo[8], hi[8];
#pragma unroll
				for (uint32_t u = 0; u < 8; u++) {
					const uint32_t j = j0 + 64 * u;
					const int64_t gi = w0 + j;
					const bool need = 4 * (j + 1) > pad;
					lo[u] = (need && gi >= 0) ? dw[gi] : 0u;
					hi[u] = (need && sh) ? dw[gi + 1] : 0u;
				}
#pragma unroll
				for (uint32_t u = 0; u < 8; u++) {
					const uint32_t j = j0 + 64 * u;
					uint32_t v = sh ? (lo[u] >> sh) | (hi[u] << (32 - sh)) : lo[u];
					if (4 * j < pad)
						v = 4 * (j + 1) > pad ? v & (0xffffffffu << (8 * (pad - 4 * j))) : 0u;
					seg[j] = v;
				}
			}
			__syncthreads();
			uint32_t crc = 0;
			if (4 * (lane + 1) * run > pad) {
				for (uint32_t i = 0; i < run; i++) {
					const uint32_t v = seg[lane * run + i];
					crc = table[3][(v & 0xff) ^ (crc >> 8)] ^ table[2][((v >> 8) & 0xff) ^ (crc & 0xff)] ^ table[1][(v >> 16) & 0xff] ^ table[0][v >> 24];
				}
			}
			uint32_t part = dgf16_mul(crc, my_pow);
#pragma unroll
			for (int o = 32; o > 0; o >>= 1)
				part ^= (uint32_t)__shfl_xor((int)part, o, 64);
			total = dgf16_mul(total, seg_pow) ^ part;
		}
		if (lane == 0) {
			const uint8_t *src = d + cd.pos;
			const uint32_t want = ((uint32_t)src[nbytes] << 8) | src[nbytes + 1];
			if (total != want)
				cands[f].status = 3;
		}
	}
}

// ---------------------------------------------------------------------------------------------
// k_dec_output
// ---------------------------------------------------------------------------------------------
#define OUT_LDS_BYTES 32768

// Scratch planes [group of LW frames][channel][sample][LW] -> interleaved PCM.  Phase 1 reads the
// planes with threads along (sample, frame), which is their memory order, undoes the stereo
// decorrelation and drops the values into LDS frame by frame; phase 2 writes each frame's run of
// the tile with lanes along the run.
__global__ __launch_bounds__(256) void k_dec_output(FaDecParams P, const FaStream *streams, const FaFrameCand *cands, const uint32_t *ncand_ptr,
                                                    const int32_t *tmp, void *out, uint32_t tile_samples, uint32_t LW, uint32_t lw_log)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
	__shared__ FaFrameCand sc[DEC_LANES];
	const uint32_t ncand = ncand_ptr[0];
	const uint32_t g = blockIdx.x;
	if (g * LW >= ncand)
		return;
	const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const uint32_t i0 = blockIdx.y * tile_samples;
	const uint32_t ch = P.channels, sb = P.out_sample_bytes;
	if (tid < LW) {
		FaFrameCand cd;
		cd.blocksize = 0;
		cd.status = 0;
		cd.assignment = 0;
		cd.sample = 0;
		cd.stream = 0;
		if (g * LW + tid < ncand)
			cd = cands[g * LW + tid];
		sc[tid] = cd;
	}
	__syncthreads();
	const int32_t *frame_tmp = tmp + (size_t)g * ch * P.max_blocksize * LW;
	const uint32_t row = tile_samples * ch;  // elements per frame in the tile
	if (ch == 2 && LW >= 4) {
		// stereo, the common case: four frames' values of one sample are 16 bytes of each plane, and a
		// frame's left/right pair goes to LDS as one word (int16 output) or two
		const int32_t *pa = frame_tmp, *pb = frame_tmp + (size_t)P.max_blocksize * LW;
		for (uint32_t idx = tid * 4; idx < tile_samples * LW; idx += 256 * 4) {
			const uint32_t fl0 = idx & (LW - 1), ii = idx >> lw_log;
			const uint32_t i = i0 + ii;
			if (i >= P.max_blocksize)
				continue;
			const int4 a4 = *(const int4 *)(pa + (size_t)i * LW + fl0);
			const int4 b4 = *(const int4 *)(pb + (size_t)i * LW + fl0);
			const int32_t av[4] = {a4.x, a4.y, a4.z, a4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
			for (int j = 0; j < 4; j++) {
				const uint32_t fl = fl0 + j;
				if (!(sc[fl].status == 1 && i < sc[fl].blocksize))
					continue;
				const uint32_t assignment = sc[fl].assignment;
				int32_t a = av[j], b = bv[j];
				// stream_decoder.c:2067-2102
				if (assignment == 1)
					b = a - b;
				else if (assignment == 2)
					a = a + b;
				else if (assignment == 3) {
					int32_t mid = (int32_t)((uint32_t)a << 1);
					mid |= (b & 1);
					a = (mid + b) >> 1;
					b = (mid - b) >> 1;
				}
				if (sb == 2)
					((uint32_t *)lds)[(fl * row + ii * 2) >> 1] = ((uint32_t)a & 0xffffu) | ((uint32_t)b << 16);
				else
					((int2 *)lds)[(fl * row + ii * 2) >> 1] = make_int2(a, b);
			}
		}
	} else
	for (uint32_t idx = tid; idx < tile_samples * LW; idx += 256) {
		const uint32_t fl = idx & (LW - 1), ii = idx >> lw_log;
		const uint32_t i = i0 + ii;
		if (!(sc[fl].status == 1 && i < sc[fl].blocksize))
			continue;
		const uint32_t assignment = sc[fl].assignment;
		if (ch == 2 && assignment != 0) {
			int32_t a = frame_tmp[(size_t)i * LW + fl];
			int32_t b = frame_tmp[((size_t)P.max_blocksize + i) * LW + fl];
			// stream_decoder.c:2067-2102
			if (assignment == 1)
				b = a - b;
			else if (assignment == 2)
				a = a + b;
			else {
				int32_t mid = (int32_t)((uint32_t)a << 1);
				mid |= (b & 1);
				a = (mid + b) >> 1;
				b = (mid - b) >> 1;
			}
			if (sb == 2) {
				((int16_t *)lds)[fl * row + ii * 2] = (int16_t)a;
				((int16_t *)lds)[fl * row + ii * 2 + 1] = (int16_t)b;
			} else {
				((int32_t *)lds)[fl * row + ii * 2] = a;
				((int32_t *)lds)[fl * row + ii * 2 + 1] = b;
			}
		} else {
			for (uint32_t c = 0; c < ch; c++) {
				const int32_t v = frame_tmp[((size_t)c * P.max_blocksize + i) * LW + fl];
				if (sb == 2)
					((int16_t *)lds)[fl * row + ii * ch + c] = (int16_t)v;
				else
					((int32_t *)lds)[fl * row + ii * ch + c] = v;
			}
		}
	}
	__syncthreads();
	for (uint32_t fl = wv; fl < LW; fl += 4) {
		const FaFrameCand &cf = sc[fl];
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
		// 16 bytes per lane where the run starts on a 16-byte boundary on both sides, single samples for the rest
		uint8_t *gdst = (uint8_t *)out + obase * sb;
		const uint8_t *lsrc = lds + (size_t)fl * row * sb;
		uint32_t e0 = 0;
		if ((((uintptr_t)gdst | (uintptr_t)((size_t)fl * row * sb)) & 15) == 0) {
			const uint32_t nvec = (nel * sb) >> 4;
			for (uint32_t v = lane; v < nvec; v += 64)
				((uint4 *)gdst)[v] = ((const uint4 *)lsrc)[v];
			e0 = (nvec << 4) / sb;
		}
		for (uint32_t e = e0 + lane; e < nel; e += 64) {
			if (sb == 2)
				((int16_t *)out)[obase + e] = ((const int16_t *)lds)[fl * row + e];
			else
				((int32_t *)out)[obase + e] = ((const int32_t *)lds)[fl * row + e];
		}
	}
}

__global__ void k_dec_summary(const FaFrameCand *cands, const uint32_t *ncand_ptr, unsigned long long *result)
{
	// result[0] good frames, [1] wide samples in good frames, [2] bad candidates outside good frames
	const uint32_t n = ncand_ptr[0];
	unsigned long long good = 0, wide = 0, bad = 0;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		if (cands[i].status == 1) {
			good++;
			wide += cands[i].blocksize;
		} else {
			// A sync pattern with a valid header CRC-8 inside the bytes of a good frame is not a frame: a
			// reader that walks the stream (frame_sync_, stream_decoder.c:1941) never looks there.  The
			// table is in byte order, so the nearest good frame in front decides.
			bool shadowed = false;
			for (uint32_t j = i; j-- > 0;) {
				if (cands[j].stream != cands[i].stream)
					break;
				if (cands[j].status == 1) {
					shadowed = cands[j].end_byte > cands[i].pos;
					break;
				}
			}
			if (!shadowed)
				bad++;
		}
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
	hipLaunchKernelGGL(k_dec_sync, dim3(nb), dim3(SYNC_THREADS), 0, st, P, flac, streams, block_counts);
	hipLaunchKernelGGL(k_dec_scan, dim3(1), dim3(1024), 0, st, block_counts, nb, block_offsets);
	hipLaunchKernelGGL(k_dec_sync_emit, dim3((nb + 255) / 256), dim3(256), 0, st, P, flac, streams, block_counts, block_offsets, cands, ncand, nb);
}

void fa_dec_launch_frames(const FaDecParams &P, const uint8_t *flac, const FaStream *streams, FaFrameCand *cands, const uint32_t *ncand,
                          int32_t *tmp, uint64_t *sub_start, hipStream_t st)
{
	const uint32_t groups = (P.max_frames + P.lw - 1) / P.lw;
	const uint32_t *words = (const uint32_t *)flac;
	if (sub_start && P.channels >= 2 && P.lw <= 16 && P.skip_lw <= 16) {
		hipLaunchKernelGGL((k_dec_frames<DEC_SKIP, 16>), dim3((P.max_frames + P.skip_lw - 1) / P.skip_lw), dim3(DEC_LANES), 0, st, P, words, streams, cands, ncand,
		                   tmp, P.skip_lw, sub_start);
		hipLaunchKernelGGL((k_dec_frames<DEC_SUB, 16>), dim3(groups * P.channels), dim3(DEC_LANES), 0, st, P, words, streams, cands, ncand, tmp, P.lw, sub_start);
		return;
	}
	// a lane per frame; the largest batches put 32 frames into a wave (twice the LDS, half the waves)
	if (P.lw <= 16)
		hipLaunchKernelGGL((k_dec_frames<DEC_WHOLE, 16>), dim3(groups), dim3(DEC_LANES), 0, st, P, words, streams, cands, ncand, tmp, P.lw, sub_start);
	else
		hipLaunchKernelGGL((k_dec_frames<DEC_WHOLE, 32>), dim3(groups), dim3(DEC_LANES), 0, st, P, words, streams, cands, ncand, tmp, P.lw, sub_start);
}

void fa_dec_launch_crc(const FaDecParams &P, const uint8_t *flac, FaFrameCand *cands, const uint32_t *ncand, hipStream_t st)
{
	uint32_t blocks = P.max_frames < 2048 ? P.max_frames : 2048;  // one wave each, eight per CU
	if (blocks == 0)
		blocks = 1;
	hipLaunchKernelGGL(k_dec_crc, dim3(blocks), dim3(64), 0, st, flac, cands, ncand);
}

void fa_dec_launch_output(const FaDecParams &P, const FaStream *streams, const FaFrameCand *cands, const uint32_t *ncand, const int32_t *tmp,
                          void *out, unsigned long long *result, hipStream_t st)
{
	uint32_t tile = OUT_LDS_BYTES / (P.lw * P.channels * P.out_sample_bytes);
	if (tile > 512)
		tile = 512;
	if (tile > P.max_blocksize)
		tile = P.max_blocksize;
	uint32_t lw_log = 0;
	while ((1u << lw_log) < P.lw)
		lw_log++;
	const uint32_t groups = (P.max_frames + P.lw - 1) / P.lw;
	const uint32_t tiles = (P.max_blocksize + tile - 1) / tile;
	const size_t lds = (size_t)P.lw * tile * P.channels * P.out_sample_bytes;
	hipLaunchKernelGGL(k_dec_output, dim3(groups, tiles), dim3(256), lds, st, P, streams, cands, ncand, tmp, out, tile, P.lw, lw_log);
	hipLaunchKernelGGL(k_dec_summary, dim3(64), dim3(256), 0, st, cands, ncand, result);
}
