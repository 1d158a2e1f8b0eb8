// FLAC frame decoder kernels for gfx950 (MI355X).
//
//   k_dec_sync, k_dec_sync_emit
//                           frame_sync_ + read_frame_header_ (stream_decoder.c:1941,2141): every byte
//                           position is tested for a sync code with a parseable header and a good
//                           CRC-8; the first counts per block, the second writes the candidates in order
//   k_dec_frames            read_subframe_* / read_residual_partitioned_rice_ (stream_decoder.c:2450-
//                           2771) and the predictor restore (fixed.c:395, lpc.c:795,1061): one lane
//                           per subframe, the lanes of a wave walk sample index i together so the
//                           scratch planes [channel][i][lane] are read and written coalesced (dec_frames.h)
//   k_dec_crc               FLAC__crc16 over each frame (stream_decoder.c:2061-2064), one wave per frame
//   k_dec_output            channel un-decorrelation (stream_decoder.c:2067-2102), width conversion
//                           and transposition into interleaved PCM; frame / sample / bad-frame counts
//
// A frame's place in the output comes from the number in its own header, so frames need no
// knowledge of each other.  Candidates that fail the structure checks or the CRC-16 write nothing;
// the output is cleared beforehand, which is also what the reference hands out for a frame with a
// bad CRC (stream_decoder.c:2105-2113).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "dec_kernels.h"
#include "dec_frames.h"



// ---------------------------------------------------------------------------------------------
// frame header
// ---------------------------------------------------------------------------------------------
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
// headers per 4 KiB block and, per wave, the mask of lanes (16-byte runs) that hold one.  A workgroup takes
// SYNC_SUB consecutive blocks, all their loads requested before the first is looked at (a workgroup per block
// lived for one load: 1.4 TB/s; four blocks: the 46 MB of the bench stream in 38 -> 20 us).
// The headers found are also kept as parsed, the first SYNC_KEEP of a block in the order the threads come by (`kept`,
// SYNC_KEEP rows per block): the emitting pass then has nothing to read again and parse, only to put them in order.
#ifndef SYNC_SUB
#define SYNC_SUB 4
#endif
#define SYNC_KEEP 8
__global__ __launch_bounds__(SYNC_THREADS) void k_dec_sync(FaDecParams P, const uint8_t *d, const FaStream *streams, uint32_t *block_counts, uint32_t nb,
                                                           FaFrameCand *kept)
{
	__shared__ uint32_t wsum[SYNC_SUB][4];
	__shared__ uint32_t nkept[SYNC_SUB];
	const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	if (tid < SYNC_SUB)
		nkept[tid] = 0;
	__syncthreads();
	// the lane masks lie behind the counts, four 64-bit words per block
	uint64_t *masks = (uint64_t *)(block_counts + ((nb + 1) & ~1u));
	// the thread's 16 bytes and the one after them, as five words; only positions that hold the
	// two sync bytes are looked at further
	uint32_t w[SYNC_SUB][SYNC_BYTES_PER_THREAD / 4 + 1];
	const uint32_t *dw = (const uint32_t *)d;
	const uint64_t nwords = (P.flac_bytes + 3) / 4;
	static_assert(SYNC_BYTES_PER_THREAD == 16, "one 16-byte load per thread");
#pragma unroll
	for (uint32_t sb = 0; sb < SYNC_SUB; sb++) {
		const uint64_t p0 = (((uint64_t)blockIdx.x * SYNC_SUB + sb) * SYNC_THREADS + tid) * SYNC_BYTES_PER_THREAD;
		if (p0 / 4 + 5 <= nwords && (((uintptr_t)d) & 15) == 0) {
			// (the common case: the run and the word behind it lie inside the input)
			const uint4 v = *(const uint4 *)(dw + p0 / 4);
			w[sb][0] = v.x;
			w[sb][1] = v.y;
			w[sb][2] = v.z;
			w[sb][3] = v.w;
			w[sb][4] = dw[p0 / 4 + 4];
		} else {
#pragma unroll
			for (uint32_t k = 0; k < SYNC_BYTES_PER_THREAD / 4 + 1; k++) {
				const uint64_t idx = p0 / 4 + k;
				w[sb][k] = idx < nwords ? dw[idx] : 0u;
			}
		}
	}
#pragma unroll
	for (uint32_t sb = 0; sb < SYNC_SUB; sb++) {
		const uint32_t blk = blockIdx.x * SYNC_SUB + sb;
		const uint64_t p0 = ((uint64_t)blk * SYNC_THREADS + tid) * SYNC_BYTES_PER_THREAD;
		uint32_t n = 0;  // headers in this run (more than two real ones in 16 bytes cannot happen)
		uint32_t hits = 0;
		// most 16-byte runs hold no 0xff byte at all: one test per word (a byte of w is 0xff where the byte of ~w is zero)
		uint32_t any_ff = 0;
#pragma unroll
		for (uint32_t k = 0; k < SYNC_BYTES_PER_THREAD / 4; k++)
			any_ff |= (~w[sb][k] - 0x01010101u) & w[sb][k] & 0x80808080u;
		if (any_ff) {
#pragma unroll
			for (uint32_t i = 0; i < SYNC_BYTES_PER_THREAD; i++) {
				const uint32_t b0 = (w[sb][i / 4] >> (8 * (i % 4))) & 0xff;
				const uint32_t b1 = (w[sb][(i + 1) / 4] >> (8 * ((i + 1) % 4))) & 0xff;
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
			if (parse_frame_header(d, p, P, streams, c)) {
				if (n < 2) {   // (what the emitting pass takes of a run)
					const uint32_t slot = atomicAdd(&nkept[sb], 1u);
					if (slot < SYNC_KEEP)
						kept[(size_t)blk * SYNC_KEEP + slot] = c;
				}
				n++;
			}
		}
		const uint64_t m1 = __ballot(n > 0), m2 = __ballot(n > 1);
		if (lane == 0) {
			if (blk < nb)
				masks[(size_t)blk * (SYNC_THREADS / 64) + wv] = m1;
			wsum[sb][wv] = (uint32_t)__popcll(m1) + (uint32_t)__popcll(m2);
		}
	}
	__syncthreads();
	if (tid < SYNC_SUB && blockIdx.x * SYNC_SUB + tid < nb)
		block_counts[blockIdx.x * SYNC_SUB + tid] = wsum[tid][0] + wsum[tid][1] + wsum[tid][2] + wsum[tid][3];
}

// The emitting pass: one thread per 4 KiB block of the counting pass.  It walks the block's lane
// masks in byte order and parses the (few) flagged 16-byte runs again, which puts the headers into
// the table in the order of their positions.  Where a block's headers go follows from the counts in
// front of it: every workgroup adds up the counts before its own 256 (a few thousand words out of L2:
// cheaper than a scan kernel of its own in between) and scans its own.
__global__ __launch_bounds__(256) void k_dec_sync_emit(FaDecParams P, const uint8_t *d, const FaStream *streams, const uint32_t *block_counts,
                                                      FaFrameCand *cands, uint32_t *ncand_out, uint32_t nb, unsigned long long *result,
                                                      const FaFrameCand *kept)
{
	__shared__ uint64_t wsum[4], wpre[4];
	const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const uint32_t b = blockIdx.x * 256 + tid;
	const uint32_t count = b < nb ? block_counts[b] : 0u;
	uint64_t at;
	{
		uint64_t pre = 0;
		for (uint32_t i = tid; i < blockIdx.x * 256; i += 256)
			pre += block_counts[i];
		uint64_t inc = count;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const uint64_t t = __shfl_up(inc, o, 64);
			if (lane >= (uint32_t)o)
				inc += t;
			pre += __shfl_xor(pre, o, 64);
		}
		if (lane == 63) {
			wsum[wv] = inc;
			wpre[wv] = pre;
		}
		__syncthreads();
		at = wpre[0] + wpre[1] + wpre[2] + wpre[3] + inc - count;
		for (uint32_t w = 0; w < wv; w++)
			at += wsum[w];
	}
	if (b >= nb)
		return;
	if (b == nb - 1) {
		const uint64_t total = at + count;
		ncand_out[0] = total > P.max_frames ? P.max_frames : (uint32_t)total;
		ncand_out[1] = total > P.max_frames ? 1u : 0u;  // overflow flag
		// the counters the output pass adds to start at zero
		result[0] = 0;
		result[1] = 0;
		result[2] = 0;
	}
	if (count == 0)
		return;
	if (count <= SYNC_KEEP) {
		// the counting pass kept them all: in order of position (a handful at most: selection by rank)
		uint64_t pos[SYNC_KEEP];
#pragma unroll
		for (uint32_t i = 0; i < SYNC_KEEP; i++)
			pos[i] = i < count ? kept[(size_t)b * SYNC_KEEP + i].pos : ~0ull;
#pragma unroll
		for (uint32_t i = 0; i < SYNC_KEEP; i++) {
			if (i >= count)
				break;
			uint32_t rank = 0;
#pragma unroll
			for (uint32_t j = 0; j < SYNC_KEEP; j++)
				rank += pos[j] < pos[i] ? 1u : 0u;
			if (at + rank < P.max_frames)
				cands[at + rank] = kept[(size_t)b * SYNC_KEEP + i];
		}
		return;
	}
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
//
// AHEAD: the pass runs beside the frame decoder, before anybody knows where a frame ends, on the guess that it ends
// where the next header of its stream was found (or with the stream); it leaves the guess and whether the CRC-16
// there matches.  The pass after the frame decoder takes the answer where the guess was right, which is every frame
// of a sound stream, and works out the rest: frames cut short or damaged, false headers inside a frame.
#define CRC_AHEAD_OK (1ull << 63)
template <bool AHEAD>
__global__ __launch_bounds__(64) void k_dec_crc(const uint8_t *d, FaFrameCand *cands, const uint32_t *ncand_ptr, const FaStream *streams,
                                                uint64_t *ahead, uint64_t max_frame_bytes)
{
	__shared__ uint32_t seg[CRC_SEG_WORDS];
	__shared__ uint16_t table[4][256];
	const uint32_t lane = threadIdx.x;
	bool have_table = false;   // (the pass after the frame decoder mostly has nothing to compute)
	const uint32_t *dw = (const uint32_t *)d;
	const uint32_t ncand = ncand_ptr[0];
	for (uint32_t f = blockIdx.x; f < ncand; f += gridDim.x) {
		FaFrameCand cd;
		if (AHEAD) {
			// (the frame decoder is writing the other fields of the row meanwhile)
			cd.pos = cands[f].pos;
			cd.stream = cands[f].stream;
			const FaStream sd = streams[cd.stream];
			cd.end_byte = (f + 1 < ncand && cands[f + 1].stream == cd.stream) ? cands[f + 1].pos : sd.byte_offset + sd.byte_length;
			if (cd.end_byte < cd.pos + 8 || cd.end_byte - cd.pos > max_frame_bytes) {
				if (lane == 0)
					ahead[f] = 0;
				continue;
			}
		} else {
			cd = cands[f];
			if (cd.status != 1)
				continue;
			if (ahead) {
				const uint64_t a = ahead[f];
				if ((a & ~CRC_AHEAD_OK) == cd.end_byte) {
					if (lane == 0 && !(a & CRC_AHEAD_OK))
						cands[f].status = 3;
					continue;
				}
			}
		}
		if (!have_table) {
			for (uint32_t e = lane; e < 4 * 256 / 2; e += 64)
				((uint32_t *)&table[0][0])[e] = ((const uint32_t *)&c_dec_crc16x4[0][0])[e];
			have_table = true;
		}
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
			if (AHEAD)
				ahead[f] = cd.end_byte | (total == want ? CRC_AHEAD_OK : 0ull);
			else if (total != want)
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
                                                    const int32_t *tmp, void *out, uint32_t tile_samples, uint32_t LW, uint32_t lw_log,
                                                    unsigned long long *result)
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
		// the group's first tile also counts the group's frames: result[0] good frames, [1] wide samples in good frames,
		// [2] bad candidates outside good frames
		if (blockIdx.y == 0) {
			const uint32_t i = g * LW + tid;
			const bool have = i < ncand, good = have && cd.status == 1;
			bool bad = false;
			if (have && !good) {
				// A sync pattern with a valid header CRC-8 inside the bytes of a good frame is not a frame: a
				// reader that walks the stream (frame_sync_, stream_decoder.c:1941) never looks there.  The
				// table is in byte order, so the nearest good frame in front decides.
				bool shadowed = false;
				for (uint32_t j = i; j-- > 0;) {
					if (cands[j].stream != cd.stream)
						break;
					if (cands[j].status == 1) {
						shadowed = cands[j].end_byte > cd.pos;
						break;
					}
				}
				bad = !shadowed;
			}
			// (LW <= 64: these lanes are one wave's)
			const unsigned long long ngood = __popcll(__ballot(good)), nbad = __popcll(__ballot(bad));
			unsigned long long wide = good ? cd.blocksize : 0;
#pragma unroll
			for (int o = 32; o > 0; o >>= 1)
				wide += __shfl_xor(wide, o, 64);
			if (tid == 0) {
				if (ngood) atomicAdd(&result[0], ngood);
				if (wide) atomicAdd(&result[1], wide);
				if (nbad) atomicAdd(&result[2], nbad);
			}
		}
	}
	__syncthreads();
	// (frames the frame decoder wrote itself, FaFrameCand.reserved bit 1, are none of this kernel's business: mostly all of them)
	{
		bool work = false;
		for (uint32_t fl = 0; fl < LW; fl++)
			work = work || (sc[fl].status == 1 && !(sc[fl].reserved & 2u));
		if (!work)
			return;
	}
	const int32_t *frame_tmp = tmp + (size_t)g * ch * P.max_blocksize * LW;
	const uint32_t row = tile_samples * ch;  // elements per frame in the tile
	const bool planar = P.planar != 0;
	if (ch == 2 && LW >= 4 && !planar) {
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
				if (!(sc[fl].status == 1 && !(sc[fl].reserved & 2u) && i < sc[fl].blocksize))
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
	} else if (ch > 2 && ch <= 8 && LW >= 4 && !planar) {
		// more channels, the same way: four frames' values of one sample are 16 bytes of every plane; a frame's values of the
		// sample leave for LDS together
		for (uint32_t idx = tid * 4; idx < tile_samples * LW; idx += 256 * 4) {
			const uint32_t fl0 = idx & (LW - 1), ii = idx >> lw_log;
			const uint32_t i = i0 + ii;
			if (i >= P.max_blocksize)
				continue;
			int32_t v[8][4];
#pragma unroll
			for (uint32_t c = 0; c < 8; c++) {
				int4 t = make_int4(0, 0, 0, 0);
				if (c < ch)
					t = *(const int4 *)(frame_tmp + ((size_t)c * P.max_blocksize + i) * LW + fl0);
				v[c][0] = t.x; v[c][1] = t.y; v[c][2] = t.z; v[c][3] = t.w;
			}
#pragma unroll
			for (int j = 0; j < 4; j++) {
				const uint32_t fl = fl0 + j;
				if (!(sc[fl].status == 1 && !(sc[fl].reserved & 2u) && i < sc[fl].blocksize))
					continue;
				const uint32_t at = fl * row + ii * ch;
				if (sb == 4 && (ch & 3u) == 0) {
					((int4 *)lds)[at >> 2] = make_int4(v[0][j], v[1][j], v[2][j], v[3][j]);
					if (ch == 8)
						((int4 *)lds)[(at >> 2) + 1] = make_int4(v[4][j], v[5][j], v[6][j], v[7][j]);
				} else {
#pragma unroll
					for (uint32_t c = 0; c < 8; c++)
						if (c < ch) {
							if (sb == 2)
								((int16_t *)lds)[at + c] = (int16_t)v[c][j];
							else
								((int32_t *)lds)[at + c] = v[c][j];
						}
				}
			}
		}
	} else
	for (uint32_t idx = tid; idx < tile_samples * LW; idx += 256) {
		const uint32_t fl = idx & (LW - 1), ii = idx >> lw_log;
		const uint32_t i = i0 + ii;
		if (!(sc[fl].status == 1 && !(sc[fl].reserved & 2u) && i < sc[fl].blocksize))
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
			const uint32_t pa = planar ? fl * row + ii : fl * row + ii * 2, pb = planar ? fl * row + tile_samples + ii : pa + 1;
			if (sb == 2) {
				((int16_t *)lds)[pa] = (int16_t)a;
				((int16_t *)lds)[pb] = (int16_t)b;
			} else {
				((int32_t *)lds)[pa] = a;
				((int32_t *)lds)[pb] = b;
			}
		} else {
			for (uint32_t c = 0; c < ch; c++) {
				const int32_t v = frame_tmp[((size_t)c * P.max_blocksize + i) * LW + fl];
				const uint32_t at = planar ? fl * row + c * tile_samples + ii : fl * row + ii * ch + c;
				if (sb == 2)
					((int16_t *)lds)[at] = (int16_t)v;
				else
					((int32_t *)lds)[at] = v;
			}
		}
	}
	__syncthreads();
	for (uint32_t fl = wv; fl < LW; fl += 4) {
		const FaFrameCand &cf = sc[fl];
		if (cf.status != 1 || (cf.reserved & 2u) || i0 >= cf.blocksize)
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
		if (planar) {
			// the frame's stretch of the output holds its channels one after the other: a run per channel
			// (so the whole frame has to have its place, not just this tile)
			const uint64_t frel = cf.sample - first;
			if ((st.total_samples && frel + cf.blocksize > st.total_samples) || st.sample_offset + frel + cf.blocksize > P.out_capacity_wide)
				continue;
			const uint64_t fbase = (st.sample_offset + frel) * ch;
			for (uint32_t c = 0; c < ch; c++) {
				const uint64_t ob = fbase + (uint64_t)c * cf.blocksize + i0;
				const uint32_t ls = fl * row + c * tile_samples;
				for (uint32_t e = lane; e < nvalid; e += 64) {
					if (sb == 2)
						((int16_t *)out)[ob + e] = ((const int16_t *)lds)[ls + e];
					else
						((int32_t *)out)[ob + e] = ((const int32_t *)lds)[ls + e];
				}
			}
			continue;
		}
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

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
uint32_t fa_dec_sync_blocks(uint64_t flac_bytes)
{
	const uint64_t per = (uint64_t)SYNC_THREADS * SYNC_BYTES_PER_THREAD;
	return (uint32_t)((flac_bytes + per - 1) / per);
}

void fa_dec_launch_sync(const FaDecParams &P, const uint8_t *flac, const FaStream *streams, uint32_t *block_counts, FaFrameCand *kept,
                        FaFrameCand *cands, uint32_t *ncand, unsigned long long *result, hipStream_t st)
{
	const uint32_t nb = fa_dec_sync_blocks(P.flac_bytes);
	hipLaunchKernelGGL(k_dec_sync, dim3((nb + SYNC_SUB - 1) / SYNC_SUB), dim3(SYNC_THREADS), 0, st, P, flac, streams, block_counts, nb, kept);
	hipLaunchKernelGGL(k_dec_sync_emit, dim3((nb + 255) / 256), dim3(256), 0, st, P, flac, streams, block_counts, cands, ncand, nb, result, kept);
}

// the instances of k_dec_frames, each in a translation unit of its own (dec_frames_*.hip)
#define DEC_FRAMES_DECL(name) \
	void name(const FaDecParams &P, const uint32_t *words, const FaStream *streams, FaFrameCand *cands, const uint32_t *ncand, int32_t *tmp, uint32_t lw, \
	          uint64_t *sub_start, uint32_t nwaves, uint32_t chain, const uint64_t *ahead, void *pcm_out, hipStream_t st)
DEC_FRAMES_DECL(fa_dec_frames_skip16);
DEC_FRAMES_DECL(fa_dec_frames_skip16d);
DEC_FRAMES_DECL(fa_dec_frames_sub16);
DEC_FRAMES_DECL(fa_dec_frames_sub16d);
DEC_FRAMES_DECL(fa_dec_frames_sub32);
DEC_FRAMES_DECL(fa_dec_frames_pipe16);
DEC_FRAMES_DECL(fa_dec_frames_pipe16d);
DEC_FRAMES_DECL(fa_dec_frames_pair16);
DEC_FRAMES_DECL(fa_dec_frames_pair16d);

// Whether the split form's subframe launch can take both channels of a stereo frame into one workgroup and write final
// PCM itself (DEC_PAIR): two channels, P.lw frames per group with 2 x P.lw <= 16 columns, every workgroup resident at once.
bool fa_dec_pair_fits(uint32_t channels, uint32_t max_frames, uint32_t frames_per_group)
{
	if (channels != 2 || frames_per_group < 1 || frames_per_group > 8)
		return false;
	return (max_frames + frames_per_group - 1) / frames_per_group <= 256 * 6;
}

// split: walk, then every (frame, channel) at once; else one launch per channel, in order (see dec_frames.h).
// pair (with split, two channels): P.lw is frames per group, a workgroup takes both channels of them; `ahead` is what the
// CRC pass that ran ahead left, `before_sub` (may be null) the event after which it is complete, `out` the caller's PCM.
void fa_dec_launch_frames(const FaDecParams &Pin, const uint8_t *flac, const FaStream *streams, FaFrameCand *cands, const uint32_t *ncand,
                          int32_t *tmp, uint64_t *sub_start, bool split, bool pair, const uint64_t *ahead, hipEvent_t before_sub, void *out,
                          hipStream_t st)
{
	FaDecParams P = Pin;
	const uint32_t groups = (P.max_frames + P.lw - 1) / P.lw;
	const uint32_t *words = (const uint32_t *)flac;
	// The deep-ring instances take 24 KB of LDS per wave, six waves per CU: used while every wave of the launch still
	// finds room at once on the 256 CUs; beyond that the shallow ones, whose pump (four refills of 64 words per service
	// point) feeds runs of two blocks of streams of up to 16 bits and single blocks of wider ones.
	uint32_t deep_waves = 256 * 6;
	uint32_t shallow_run = P.bps <= 16 ? 32u : 16u;
	// experiment switches: FLAC_AMD_DEC_RUN = run length of the shallow instances, FLAC_AMD_DEC_DEEP = most waves for the deep ones
	if (const char *env = getenv("FLAC_AMD_DEC_RUN"))
		shallow_run = (uint32_t)atoi(env);
	if (const char *env = getenv("FLAC_AMD_DEC_DEEP"))
		deep_waves = (uint32_t)atoi(env);
	if (split && P.channels >= 2 && P.lw <= 16 && P.skip_lw <= 16) {
		const uint32_t walk_waves = (P.max_frames + P.skip_lw - 1) / P.skip_lw, sub_waves = groups * P.channels;
		P.run = walk_waves <= deep_waves ? 64u : shallow_run;
		(walk_waves <= deep_waves ? fa_dec_frames_skip16d : fa_dec_frames_skip16)(P, words, streams, cands, ncand, tmp, P.skip_lw, sub_start, walk_waves, 0xffffffffu, nullptr, nullptr, st);
		if (pair && fa_dec_pair_fits(P.channels, P.max_frames, P.lw)) {
			if (before_sub)
				(void)hipStreamWaitEvent(st, before_sub, 0);
			const bool deep = groups <= 256 * 4;
			P.run = deep ? 64u : shallow_run;
			(deep ? fa_dec_frames_pair16d : fa_dec_frames_pair16)(P, words, streams, cands, ncand, tmp, 2 * P.lw, sub_start, groups, 0xffffffffu, ahead, out, st);
			return;
		}
		// Two waves per subframe group (one reads, one restores) while all of them find room on the chip at once: 35 KB
		// of LDS per workgroup with the deep ring, four to a CU; with the shallow ring the registers (147 per thread)
		// allow six.  Measured (frames stage, one wave -> two): 16-bit stereo 256 frames 0.63 -> 0.50 ms, 6 460 frames
		// 0.85 -> 0.65; eight channels of 24 bits, 2 813 frames (1 408 groups) 2.70 -> 2.38; beyond what fits at once
		// (24-bit stereo 14 063 frames, 1 758 groups) 1.46 -> 1.63, so those keep one wave.
		uint32_t pipe_groups = 256 * 6;
		if (const char *env = getenv("FLAC_AMD_DEC_PIPE"))
			pipe_groups = (uint32_t)atoi(env);
		if (sub_waves <= pipe_groups) {
			const bool deep = sub_waves <= 256 * 4;
			P.run = deep ? 64u : shallow_run;
			(deep ? fa_dec_frames_pipe16d : fa_dec_frames_pipe16)(P, words, streams, cands, ncand, tmp, P.lw, sub_start, sub_waves, 0xffffffffu, nullptr, nullptr, st);
			return;
		}
		P.run = sub_waves <= deep_waves ? 64u : shallow_run;
		(sub_waves <= deep_waves ? fa_dec_frames_sub16d : fa_dec_frames_sub16)(P, words, streams, cands, ncand, tmp, P.lw, sub_start, sub_waves, 0xffffffffu, nullptr, nullptr, st);
		return;
	}
	// the largest batches put 32 frames into a wave (twice the LDS, half the waves)
	for (uint32_t c = 0; c < P.channels; c++) {
		if (P.lw <= 16) {
			P.run = groups <= deep_waves ? 64u : shallow_run;
			(groups <= deep_waves ? fa_dec_frames_sub16d : fa_dec_frames_sub16)(P, words, streams, cands, ncand, tmp, P.lw, sub_start, groups, c, nullptr, nullptr, st);
		} else {
			P.run = 16;   // (32 lanes of a wave read 32 samples' worth of words faster than four refills per service point bring them)
			fa_dec_frames_sub32(P, words, streams, cands, ncand, tmp, P.lw, sub_start, groups, c, nullptr, nullptr, st);
		}
	}
}

// before_frames: the pass that guesses the frames' ends and fills `ahead` (max_frames words); else the pass after the
// frame decoder, which uses `ahead` if given
void fa_dec_launch_crc(const FaDecParams &P, const uint8_t *flac, FaFrameCand *cands, const uint32_t *ncand, const FaStream *streams, uint64_t *ahead,
                       bool before_frames, hipStream_t st)
{
	uint32_t blocks = P.max_frames < 2048 ? P.max_frames : 2048;  // one wave each, eight per CU
	if (blocks == 0)
		blocks = 1;
	// no frame is longer than its samples written out in full plus the headers
	const uint64_t max_frame_bytes = (uint64_t)P.max_blocksize * P.channels * 5 + 1024;
	if (before_frames)
		hipLaunchKernelGGL(k_dec_crc<true>, dim3(blocks), dim3(64), 0, st, flac, cands, ncand, streams, ahead, max_frame_bytes);
	else
		hipLaunchKernelGGL(k_dec_crc<false>, dim3(blocks), dim3(64), 0, st, flac, cands, ncand, streams, ahead, max_frame_bytes);
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
	hipLaunchKernelGGL(k_dec_output, dim3(groups, tiles), dim3(256), lds, st, P, streams, cands, ncand, tmp, out, tile, P.lw, lw_log, result);
}
