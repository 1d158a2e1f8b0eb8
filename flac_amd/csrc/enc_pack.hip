// FLAC frame encoder kernels for gfx950 (MI355X), part 3: bit packing of the chosen subframes.
#include <hip/hip_runtime.h>

#include <stdlib.h>
#include <string.h>

#include "enc_common.h"
#include "enc_kernels.h"
#include "enc_wave.h"

// ---------------------------------------------------------------------------------------------
// k_pack
// ---------------------------------------------------------------------------------------------
#define PK_CAPW 4096  // window of the frame's bitstream held in LDS, in 32-bit words

__constant__ uint16_t c_crc16_table[256];
__constant__ uint16_t c_crc16_pow[32];  // x^(8 * 2^j) mod the CRC-16 polynomial
// x^(8 n) for n = i and n = 256 i: two look-ups and one product give any power a frame needs (gf16_xpow8)
__constant__ uint16_t c_crc16_xp_lo[256];
__constant__ uint16_t c_crc16_xp_hi[1024];
__constant__ uint8_t c_crc8_table[256] __attribute__((aligned(16)));
__constant__ uint16_t c_crc16x4[4][256];    // table k: CRC-16 of a byte followed by k zero bytes
// window_crc16 works on 256 runs of 1 << c words; thread t's run is followed by 255 - t more
// runs, so its CRC is multiplied by c_crc16_runpow[c][t] = x^(8 * 4 * (1 << c) * (255 - t))
#define PK_RUN_CLASSES 6
__constant__ uint16_t c_crc16_runpow[PK_RUN_CLASSES][FA_THREADS];

hipError_t fa_upload_crc_tables(const uint16_t *t16, const uint16_t *pow16, const uint8_t *t8)
{
	hipError_t rc = hipMemcpyToSymbol(HIP_SYMBOL(c_crc16_table), t16, 512);
	if (rc == hipSuccess)
		rc = hipMemcpyToSymbol(HIP_SYMBOL(c_crc16_pow), pow16, 64);
	if (rc == hipSuccess)
		rc = hipMemcpyToSymbol(HIP_SYMBOL(c_crc8_table), t8, 256);
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
	rc = hipMemcpyToSymbol(HIP_SYMBOL(c_crc16x4), x4, sizeof(x4));
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
	static uint16_t xp_lo[256], xp_hi[1024];
	for (int i = 0; i < 256; i++)
		xp_lo[i] = (uint16_t)xpow8((uint64_t)i);
	for (int i = 0; i < 1024; i++)
		xp_hi[i] = (uint16_t)xpow8(256ull * (uint64_t)i);
	rc = hipMemcpyToSymbol(HIP_SYMBOL(c_crc16_xp_lo), xp_lo, sizeof(xp_lo));
	if (rc == hipSuccess)
		rc = hipMemcpyToSymbol(HIP_SYMBOL(c_crc16_xp_hi), xp_hi, sizeof(xp_hi));
	if (rc != hipSuccess)
		return rc;
	static uint16_t runpow[PK_RUN_CLASSES][FA_THREADS];
	for (int c = 0; c < PK_RUN_CLASSES; c++)
		for (int t = 0; t < FA_THREADS; t++)
			runpow[c][t] = (uint16_t)xpow8(4ull * (1u << c) * (uint64_t)(FA_THREADS - 1 - t));
	return hipMemcpyToSymbol(HIP_SYMBOL(c_crc16_runpow), runpow, sizeof(runpow));
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

// x^(8 n) modulo the polynomial.  (Bit by bit over n this was sixteen products in a row, each a hundred dependent
// instructions of one lane: 20 000 cycles in front of a frame's last bytes.)
__device__ __forceinline__ uint32_t gf16_xpow8(uint32_t n)
{
	uint32_t r = gf16_mul(c_crc16_xp_lo[n & 255u], c_crc16_xp_hi[(n >> 8) & 1023u]);
	n >>= 18;
	for (int j = 18; n; j++, n >>= 1)
		if (n & 1)
			r = gf16_mul(r, c_crc16_pow[j]);
	return r;
}

// The bytes of a frame header (at most 16) are collected in two 64-bit words, first byte on top.
__device__ __forceinline__ void hdr_put(uint64_t (&hw)[2], uint32_t &n, uint32_t byte)
{
	const uint64_t v = (uint64_t)(byte & 0xff) << (56 - 8 * (n & 7));
	if (n < 8)
		hw[0] |= v;
	else
		hw[1] |= v;
	n++;
}

__device__ __forceinline__ uint32_t hdr_get(const uint64_t (&hw)[2], uint32_t i) { return (uint32_t)((i < 8 ? hw[0] : hw[1]) >> (56 - 8 * (i & 7))) & 0xff; }

// FLAC__frame_add_header (stream_encoder_framing.c:219) up to, not including, its CRC-8; returns the
// number of bytes.  Every argument is the same in all threads.
__device__ __forceinline__ uint32_t frame_header_bytes(const FaEncParams &P, uint32_t bsf, uint32_t frame_number, uint32_t ca, uint64_t (&hw)[2])
{
	hw[0] = hw[1] = 0;
	uint32_t n = 0;
	auto put = [&](uint32_t byte) { hdr_put(hw, n, byte); };
	put(0xff);
	put(0xf8);
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
	put((uint8_t)((u << 4) | v));
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
	put((uint8_t)((chcode << 4) | (bcode << 1)));
	// FLAC__bitwriter_write_utf8_uint32 (bitwriter.c:784)
	const uint32_t val = frame_number;
	if (val < 0x80)
		put((uint8_t)val);
	else {
		const int cont = val < 0x800 ? 1 : val < 0x10000 ? 2 : val < 0x200000 ? 3 : val < 0x4000000 ? 4 : 5;
		const uint32_t lead = (0xff00u >> (cont + 1)) & 0xff;  // 0xC0, 0xE0, 0xF0, 0xF8, 0xFC
		put((uint8_t)(lead | (val >> (6 * cont))));
		for (int i = cont - 1; i >= 0; i--)
			put((uint8_t)(0x80 | ((val >> (6 * i)) & 0x3f)));
	}
	if (bs_hint == 6)
		put((uint8_t)(bsf - 1));
	else if (bs_hint == 7) {
		put((uint8_t)((bsf - 1) >> 8));
		put((uint8_t)(bsf - 1));
	}
	if (sr_hint == 12)
		put((uint8_t)(P.sample_rate / 1000));
	else if (sr_hint == 13) {
		put((uint8_t)(P.sample_rate >> 8));
		put((uint8_t)P.sample_rate);
	} else if (sr_hint == 14) {
		put((uint8_t)((P.sample_rate / 10) >> 8));
		put((uint8_t)(P.sample_rate / 10));
	}
	return n;
}

struct PkShared {
	uint32_t W[PK_CAPW] __attribute__((aligned(16)));
	uint32_t scan[4];
	uint32_t crcs[4];
	uint16_t crc_tables[4][256];
	uint32_t base_word;
	uint32_t frame_crc;
	uint32_t error;
	uint8_t  params[FA_MAX_PARTS];
	int32_t  q[FA_MAX_ORDER];
};

template <int HIST>
struct PkStage {
	int32_t v[ST_WORDS(HIST)];
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

// CRC-16 of nbytes (<= 4 PK_CAPW) bytes that start at window word 0 (FLAC__crc16, crc.c:134).  The
// bytes are taken as 256 runs of 1, 2, 4, 8, 16 or 32 words, padded with zeros IN FRONT (leading zeros do
// not change a CRC that starts at 0) so that the last run ends with the last byte.  Each thread puts
// its run through four byte tables at once and multiplies the result by x^(8 * bytes that follow
// it), a power that comes from a table because the run sizes are fixed; the CRC being linear, the
// XOR of the 256 products is the answer.  nbytes is the same in all threads.
__device__ __forceinline__ uint32_t window_crc16(const uint32_t *W, const uint16_t (*crc_tables)[256], uint32_t *crcs, uint32_t nbytes)
{
	const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const uint32_t nw = (nbytes + 3) / 4;       // words after padding in front
	const uint32_t pad = nw * 4 - nbytes;       // 0..3 zero bytes in front
	uint32_t c = 0;
	while (c + 1 < PK_RUN_CLASSES && ((uint32_t)FA_THREADS << c) < nw)
		c++;
	const uint32_t run = 1u << c;
	const uint32_t my_pow = c_crc16_runpow[c][tid];
	// padded word j = stream bytes [4 j - pad, 4 j - pad + 4); the window holds stream bytes
	// big-endian, four to a word
	const int32_t first = (int32_t)(tid * run) - (int32_t)(FA_THREADS * run - nw);
	uint32_t crc = 0;
	if (first + (int32_t)run > 0) {
		uint32_t prev = first > 0 ? W[first - 1] : 0u;
		for (uint32_t i = 0; i < run; i++) {
			const int32_t j = first + (int32_t)i;
			if (j < 0)
				continue;
			const uint32_t cur = W[j];
			const uint32_t v = pad ? (prev << (32 - 8 * pad)) | (cur >> (8 * pad)) : cur;
			prev = cur;
			crc = crc_tables[3][(v >> 24) ^ (crc >> 8)] ^ crc_tables[2][((v >> 16) & 0xff) ^ (crc & 0xff)] ^ crc_tables[1][(v >> 8) & 0xff] ^
			      crc_tables[0][v & 0xff];
		}
	}
	uint32_t part = gf16_mul(crc, my_pow);
#pragma unroll
	for (int o = 32; o > 0; o >>= 1)
		part ^= (uint32_t)__shfl_xor((int)part, o, 64);
	__syncthreads();
	if (lane == 0)
		crcs[wv] = part;
	__syncthreads();
	const uint32_t out = crcs[0] ^ crcs[1] ^ crcs[2] ^ crcs[3];
	__syncthreads();
	return out;
}

// Write window words [0, nwords) to the frame slot (big-endian bit order -> byte order), fold
// their CRC into the frame CRC, slide the window.  keep_partial: word nwords holds the
// unfinished tail and becomes word 0.
__device__ __forceinline__ void flush_window(PkShared &S, uint32_t *slot, uint32_t slot_words, uint32_t nwords, bool keep_partial, bool with_crc,
                                             bool final = false)
{
	const uint32_t tid = threadIdx.x;
	if (with_crc && nwords) {
		const uint32_t c = window_crc16(S.W, S.crc_tables, S.crcs, nwords * 4);
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
	if (final)
		return;  // the window is not used again
	const uint32_t carry = keep_partial ? S.W[nwords] : 0u;
	__syncthreads();
	for (uint32_t w = tid; w < nwords + 2 && w < PK_CAPW; w += FA_THREADS)
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
		// everything so far is inside the window, with a word to spare for the frame CRC: it goes out
		// when the window fills or the frame ends
		if (end_word + 2 <= bw + PK_CAPW)
			return;
		if (end_word < bw + PK_CAPW) {
			// all of this tile's bits are in, but the window is nearly full: make room
			flush_window(S, slot, slot_words, end_word - bw, true, true);
			return;
		}
		flush_window(S, slot, slot_words, PK_CAPW, false, true);
	}
}

// signal sg (left, right, mid, side) of a 16-bit stereo word; sg is the same in every lane
__device__ __forceinline__ int32_t quad_signal(uint32_t w, uint32_t sg)
{
	const int32_t l = (int32_t)(int16_t)(w & 0xffff), r = (int32_t)w >> 16;
	return sg == 0 ? l : sg == 1 ? r : sg == 2 ? ((l + r) >> 1) : (l - r);
}

// What a stereo frame with mid/side needs of its four plans, fetched at entry before it is known
// which two of them get written.
struct PkPlans {
	uint8_t params[4][FA_MAX_PARTS];
	int32_t q[4][FA_MAX_ORDER];
	uint4 head[4];   // bits | type, order, precision, shift | part_order, rice2, wide, - | constant
	uint2 stat[4];   // wasted, subframe bps
};

// QUAD: 16-bit stereo held as one 32-bit word per wide sample, mid/side evaluated.  Everything the
// frame needs from global memory (the words of its first tile, the four plans, the channel
// assignment) is requested at entry in one go; the tile is staged once, as raw words, and each
// written signal is formed from it in registers.
template <int HIST, bool NARROW, bool QUAD>
__global__ __launch_bounds__(FA_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_pack(FaEncParams P, FaBatch B, const void *pcm, const FaSubStat *stats, const FaPlan *plans,
                                                     const uint32_t *assign, uint32_t *slots, uint32_t *frame_bytes, uint32_t *errflag, unsigned long long *stamps,
                                                     uint32_t first)
{
// debugging aid: thread 0's shader clock at point k goes to stamps[block][k]
#define STAMP(k) do { if (stamps && threadIdx.x == 0) stamps[(size_t)f * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
	__shared__ PkShared S;
	__shared__ PkStage<HIST> T;
	__shared__ PkPlans PL;
	const uint32_t f = first + blockIdx.x, tid = threadIdx.x;   // the frames before `first` are k_pack_frame's
	const uint32_t bsf = frame_len(B, P.blocksize, f);
	const uint64_t frame_start = (uint64_t)f * P.blocksize;
	const uint32_t slot_words = P.slot_bytes / 4;
	uint32_t *slot = slots + (size_t)f * slot_words;
	constexpr uint32_t HP = ST_HP(HIST);
	constexpr uint32_t NIT = (HP + FA_TILE + FA_THREADS - 1) / FA_THREADS;
	const uint32_t *words = (const uint32_t *)pcm + frame_start;
	STAMP(0);

	uint32_t pre_w[QUAD ? NIT : 1];
	uint8_t pre_params[4] = {0, 0, 0, 0};
	int32_t pre_q[4] = {0, 0, 0, 0};
	uint4 pre_head = {0, 0, 0, 0};
	uint2 pre_stat = {0, 0};
	if (QUAD) {
#pragma unroll
		for (uint32_t i = 0; i < NIT; i++) {
			const uint32_t j = tid + i * FA_THREADS;
			int32_t idx = (int32_t)j - (int32_t)HP;
			idx = idx < 0 ? 0 : ((uint32_t)idx >= bsf ? (int32_t)bsf - 1 : idx);
			pre_w[i] = words[idx];
		}
		const FaPlan *fp = plans + (size_t)f * 4;
#pragma unroll
		for (int sg = 0; sg < 4; sg++) {
			pre_params[sg] = fp[sg].params[tid];
			if (tid < FA_MAX_ORDER)
				pre_q[sg] = fp[sg].q[tid];
		}
		if (tid < 4) {
			pre_head = *(const uint4 *)(fp + tid);
			pre_stat.x = stats[(size_t)f * 4 + tid].wasted;
			pre_stat.y = stats[(size_t)f * 4 + tid].bps;
		}
	}
	const uint32_t ca = assign[f];

	for (uint32_t w = tid; w < PK_CAPW / 4; w += FA_THREADS)
		((uint4 *)S.W)[w] = make_uint4(0, 0, 0, 0);
	for (uint32_t e = tid; e < 4 * 256 / 2; e += FA_THREADS)
		((uint32_t *)&S.crc_tables[0][0])[e] = ((const uint32_t *)&c_crc16x4[0][0])[e];
	if (tid == 0) {
		S.base_word = 0;
		S.frame_crc = 0;
		S.error = 0;
	}
	if (QUAD) {
#pragma unroll
		for (int sg = 0; sg < 4; sg++) {
			PL.params[sg][tid] = pre_params[sg];
			if (tid < FA_MAX_ORDER)
				PL.q[sg][tid] = pre_q[sg];
		}
		if (tid < 4) {
			PL.head[tid] = pre_head;
			PL.stat[tid] = pre_stat;
		}
		const uint32_t n0 = HP + ((umin32(bsf, FA_TILE) + 15) & ~15u);
#pragma unroll
		for (uint32_t i = 0; i < NIT; i++) {
			const uint32_t j = tid + i * FA_THREADS;
			const bool inside = j >= HP && j - HP < bsf;
			if (j < n0)
				T.v[j + (j >> 4)] = inside ? (int32_t)pre_w[i] : 0;
		}
	}
	__syncthreads();
	STAMP(1);

	// ---- frame header: FLAC__frame_add_header (stream_encoder_framing.c:219) ----
	uint32_t bitpos;
	{
		// everything here is the same in all threads
		uint64_t hw[2];
		uint32_t n = frame_header_bytes(P, bsf, B.first_frame + f, ca, hw);
		auto put = [&](uint32_t byte) { hdr_put(hw, n, byte); };
		auto get = [&](uint32_t i) { return hdr_get(hw, i); };
		uint8_t crc = 0;  // FLAC__crc8 (crc.c:124)
		for (uint32_t i = 0; i < n; i++)
			crc = c_crc8_table[crc ^ get(i)];
		put(crc);
		if (tid == 0)
			for (uint32_t i = 0; i < n; i += 4) {
				const uint32_t nb = n - i < 4 ? n - i : 4;  // bytes in this word
				const uint32_t word = (uint32_t)((i < 8 ? hw[0] : hw[1]) >> (i & 4 ? 0 : 32));
				emit_bits(S, 8 * i, 8 * nb, word >> (32 - 8 * nb));
			}
		bitpos = 8 * n;
	}
	__syncthreads();
	STAMP(2);

	// ---- subframes in written order (stream_encoder.c:3091-3151) ----
	for (uint32_t c = 0; c < P.channels; c++) {
		{
			// The fields in front of a subframe's codes are written once, not under drain(): they need room in the window now
			// (at most 8 + 32 + 32 * 33 + 9 + 32 * 15 + 6 bits); what does not fit a window would be dropped.
			const uint32_t bw = S.base_word;
			if ((bitpos >> 5) + 64 > bw + PK_CAPW)
				flush_window(S, slot, slot_words, (bitpos >> 5) - bw, true, true);
		}
		uint32_t sig = c;
		if (ca == 1 && c == 1) sig = P.channels + 1;       // left, side
		else if (ca == 2 && c == 0) sig = P.channels + 1;  // side, right
		else if (ca == 3) sig = P.channels + c;            // mid, side
		const FaPlan *pl = plans + (size_t)f * P.nsignals + sig;
		uint32_t type, order, sbps, wasted, precision, po, rice2, mode;
		int shift;
		int32_t constant;
		if (QUAD) {
			const uint4 hd = PL.head[sig];
			const uint2 sw = PL.stat[sig];
			type = hd.y & 0xff;
			order = (hd.y >> 8) & 0xff;
			precision = (hd.y >> 16) & 0xff;
			shift = (int)(int8_t)(hd.y >> 24);
			po = hd.z & 0xff;
			rice2 = (hd.z >> 8) & 0xff;
			mode = (hd.z >> 16) & 0xff;
			constant = (int32_t)hd.w;
			wasted = sw.x;
			sbps = sw.y;
		} else {
			const FaSubStat st = stats[(size_t)f * P.nsignals + sig];
			type = pl->type; order = pl->order; sbps = st.bps; wasted = st.wasted;
			precision = pl->precision; po = pl->part_order; rice2 = pl->rice2; mode = pl->wide;
			shift = pl->shift;
			constant = pl->constant;
			S.params[tid] = pl->params[tid];
			if (tid < FA_MAX_ORDER)
				S.q[tid] = pl->q[tid];
			__syncthreads();
		}
		const uint8_t *params = QUAD ? PL.params[sig] : S.params;
		const int32_t *qv = QUAD ? PL.q[sig] : S.q;

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
				emit_bits(S, pos, sbps, (uint32_t)constant);
			pos += sbps;
			__syncthreads();
			bitpos = pos;
			continue;
		}
		uint32_t plen = 0, psize = bsf, first_code_sample = 0, code_kind = 0;  // code_kind 0: verbatim fields, 1: Rice codes
		if (type != FA_SUB_VERBATIM) {
			// warm-up samples, LPC parameters, entropy coding method header
			if (tid < order) {
				int32_t smp;
				if (QUAD && bsf <= FA_TILE) {
					const uint32_t j = HP + tid;
					smp = quad_signal((uint32_t)T.v[j + (j >> 4)], sig) >> wasted;
				} else
					smp = fetch_signal(pcm, P.channels, B.sample_bytes, fa_layout(B), frame_start + tid, sig) >> wasted;
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
					emit_bits(S, pos + tid * precision, precision, (uint32_t)qv[tid]);
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
		if (!QUAD)
			__syncthreads();  // S.params / S.q of this subframe; with QUAD the plans are in LDS since entry
		STAMP(3 + 4 * (c & 1));

		// ---- per-sample codes, tile by tile ----
		uint32_t tile_start_bit = pos;
		for (uint32_t tile = 0; tile * FA_TILE < bsf; tile++) {
			if (!QUAD) {
				stage_tile<HIST>(T.v, pcm, P.channels, B.sample_bytes, fa_layout(B), frame_start, bsf, sig, wasted, tile * FA_TILE);
				__syncthreads();
			} else if (bsf > FA_TILE) {
				// more than one tile: each is staged when its turn comes, raw like the first
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
					const int32_t idx = (int32_t)(tile * FA_TILE + j) - (int32_t)HP;
					if (j < n)
						T.v[j + (j >> 4)] = (idx >= 0 && (uint32_t)idx < bsf) ? (int32_t)w[i] : 0;
				}
				__syncthreads();
			}
			const int32_t base = (int32_t)(tile * FA_TILE + tid * FA_CHUNK);
			int32_t res[FA_CHUNK];
			uint32_t nbits = 0;
			const bool have = (uint32_t)base < bsf;
			if (have) {
				int32_t d[HIST + FA_CHUNK];
				load_chunk_lds<HIST>(d, T.v);
				if (QUAD) {
					// the staged words become the written signal (mid = (L+R)>>1, side = L-R:
					// stream_encoder.c:1991-1992), minus its wasted bits
#pragma unroll
					for (int k = 0; k < HIST + FA_CHUNK; k++)
						d[k] = quad_signal((uint32_t)d[k], sig) >> wasted;
				}
				if (type == FA_SUB_VERBATIM) {
#pragma unroll
					for (int i = 0; i < FA_CHUNK; i++)
						res[i] = d[HIST + i];
				} else if (type == FA_SUB_FIXED)
					fixed_residual16<HIST>(d, order, res);
				else
					lpc_residual16_any<HIST, NARROW>(d, qv, order, shift, (int)mode, res);
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
						if (code_kind == 0)
							nbits += sbps;
						else {
							const uint32_t k = params[part];
							const uint32_t uz = ((uint32_t)res[i] << 1) ^ (uint32_t)(res[i] >> 31);
							nbits += (uz >> k) + 1 + k;
							if (idx == first_code_sample || idx == part * psize)
								nbits += plen;
						}
					}
				}
			}
			STAMP(4 + 4 * (c & 1));
			uint32_t tile_bits;
			const uint32_t my_start = tile_start_bit + block_exclusive_scan(S, nbits, &tile_bits);
			STAMP(5 + 4 * (c & 1));
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
							const uint32_t k = params[part];
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
			STAMP(6 + 4 * (c & 1));
			tile_start_bit = tile_end_bit;
		}
		bitpos = tile_start_bit;
		__syncthreads();
	}

	STAMP(11);
	// ---- zero padding, CRC-16 over everything so far (stream_encoder.c:2885-2900), final flush ----
	const uint32_t nbytes = (bitpos + 7) >> 3;
	{
		const uint32_t bw = S.base_word;
		const uint32_t tail_bytes = nbytes - bw * 4;
		const uint32_t c = window_crc16(S.W, S.crc_tables, S.crcs, tail_bytes);
		STAMP(12);
		if (tid == 0) {
			const uint32_t crc = gf16_mul(S.frame_crc, gf16_xpow8(tail_bytes)) ^ c;
			emit_bits(S, nbytes * 8, 16, crc);
		}
		__syncthreads();
		const uint32_t total_words = (nbytes + 2 + 3) / 4;
		flush_window(S, slot, slot_words, total_words - bw, false, false, true);
	}
	STAMP(13);
	if (tid == 0) {
		frame_bytes[f] = nbytes + 2;
		if (S.error)
			atomicOr(errflag, 1u);
	}
#undef STAMP
}

// ---------------------------------------------------------------------------------------------
// k_pack_frame: the same frame (process_frame_, stream_encoder.c:2861) for the common shape - blocks
// of 16-bit stereo with mid/side evaluated, at most FA_TILE samples, residuals that need 32-bit sums
// only, Rice partitions that are whole multiples of a thread's sixteen samples (a short last block has
// one partition, stream_encoder.c:2929) - written so that the frame is passed over once.
//
// The whole frame's bit stream fits one LDS window, which takes the place of the staged samples once
// every thread has its own in registers.  Both written subframes are worked out before anything is
// emitted: a thread keeps the sixteen values it will code for each of them (zigzagged residuals or
// verbatim samples), one block scan over the pair of lengths places every thread's run of both
// subframes, and the runs, the subframe headers (whose sizes follow from the plans alone), the warm-up
// samples, the coefficients and the frame header all go into the window side by side.  A field is
// positioned with one shift and one v_alignbit_b32 and lands with two ds_or_b32, no conditions: zero
// bits never need writing and the window was cleared.  CRC-16 and the copy to the frame's slot follow.
// ---------------------------------------------------------------------------------------------
#define PQ_CAPW 5184   // words of the window (a C2 slot is 5152 words)

template <int HIST>
struct PqShared {
	union {
		uint32_t W[PQ_CAPW];
		int32_t T[ST_WORDS(HIST)];
	} u __attribute__((aligned(16)));
	uint16_t crc_tables[4][256];
	uint32_t crcs[4];
	uint32_t scan[4][2];
	uint8_t crc8[256] __attribute__((aligned(4)));
	uint8_t params[4][FA_MAX_PARTS];
	int32_t q[4][FA_MAX_ORDER];
	uint4 head[4];   // FaPlan's first words: bits | type, order, precision, shift | part_order, rice2, wide, - | constant
	uint2 stat[4];   // wasted, subframe bps
	uint64_t place;  // bytes in front of the frame in the output
};

// One written subframe, the same in every thread.
struct PqSub {
	uint32_t sig, type, order, precision, shift, po, rice2, wasted, sbps;
	uint32_t hbits;   // bits in front of the first partition's parameter (of the first sample when verbatim)
	uint32_t plen, psize;
	uint32_t constant;
};

template <typename SH>
__device__ __forceinline__ PqSub pq_sub(const SH &S, uint32_t sig, uint32_t bsf)
{
	PqSub D;
	const uint4 hd = S.head[sig];
	const uint2 sw = S.stat[sig];
	const uint32_t h1 = __builtin_amdgcn_readfirstlane(hd.y), h2 = __builtin_amdgcn_readfirstlane(hd.z);
	D.sig = sig;
	D.type = h1 & 0xff;
	D.order = (h1 >> 8) & 0xff;
	D.precision = (h1 >> 16) & 0xff;
	D.shift = (uint32_t)(int32_t)(int8_t)(h1 >> 24);
	D.po = h2 & 0xff;
	D.rice2 = (h2 >> 8) & 0xff;
	D.constant = __builtin_amdgcn_readfirstlane(hd.w);
	D.wasted = __builtin_amdgcn_readfirstlane(sw.x);
	D.sbps = __builtin_amdgcn_readfirstlane(sw.y);
	D.plen = D.rice2 ? 5 : 4;
	D.psize = bsf >> D.po;
	// FLAC__subframe_add_* (stream_encoder_framing.c:366-479)
	D.hbits = 8 + D.wasted;
	if (D.type == FA_SUB_CONSTANT)
		D.hbits += D.sbps;
	else if (D.type == FA_SUB_FIXED)
		D.hbits += D.order * D.sbps + 6;
	else if (D.type == FA_SUB_LPC)
		D.hbits += D.order * (D.sbps + D.precision) + 9 + 6;
	return D;
}

// OR a field into the window at bit position q, most significant bit first; vh holds the field's bits
// left-aligned with zeros behind them.  The caller has made sure that q / 32 + 1 is inside the window.
__device__ __forceinline__ void pq_emit(uint32_t *W, uint32_t q, uint32_t vh)
{
	const uint32_t sh = q & 31u;
	uint32_t *p = W + (q >> 5);
	atomicOr(p, vh >> sh);
	atomicOr(p + 1, __builtin_amdgcn_alignbit(vh, 0u, sh));
}

__device__ __forceinline__ uint32_t pq_mask(uint32_t n) { return n >= 32 ? 0xffffffffu : (1u << n) - 1u; }

template <int HIST, bool NARROW>
__device__ __forceinline__ uint32_t pq_values_of(const int32_t (&d)[HIST + FA_CHUNK], const PqSub &D, const int32_t *qv, uint32_t base, uint32_t bsf, uint32_t k,
                                                 bool pstart, int mode, uint32_t (&uz)[FA_CHUNK]);

// What the thread codes for its sixteen samples of a subframe (uz: the zigzagged residual, or the sample
// itself when verbatim) from the staged stereo words, and the length of its run in bits.  k: the Rice
// parameter of the partition the samples lie in; pstart: they are its first, so the parameter precedes them.
template <int HIST>
__device__ __forceinline__ uint32_t pq_values(const int32_t (&raw)[HIST + FA_CHUNK], const PqSub &D, const int32_t *qv, uint32_t base, uint32_t bsf, uint32_t k,
                                              bool pstart, uint32_t (&uz)[FA_CHUNK])
{
	// mid = (L+R)>>1, side = L-R (stream_encoder.c:1991-1992), minus the wasted bits
	int32_t d[HIST + FA_CHUNK];
	switch (D.sig) {
	case 0:
#pragma unroll
		for (int j = 0; j < HIST + FA_CHUNK; j++)
			d[j] = (int32_t)(int16_t)((uint32_t)raw[j] & 0xffff);
		break;
	case 1:
#pragma unroll
		for (int j = 0; j < HIST + FA_CHUNK; j++)
			d[j] = raw[j] >> 16;
		break;
	case 2:
#pragma unroll
		for (int j = 0; j < HIST + FA_CHUNK; j++)
			d[j] = ((int32_t)(int16_t)((uint32_t)raw[j] & 0xffff) + (raw[j] >> 16)) >> 1;
		break;
	default:
#pragma unroll
		for (int j = 0; j < HIST + FA_CHUNK; j++)
			d[j] = (int32_t)(int16_t)((uint32_t)raw[j] & 0xffff) - (raw[j] >> 16);
		break;
	}
	if (D.wasted) {
#pragma unroll
		for (int j = 0; j < HIST + FA_CHUNK; j++)
			d[j] >>= D.wasted;
	}
	return pq_values_of<HIST, true>(d, D, qv, base, bsf, k, pstart, 0, uz);
}

// the same from the subframe's own samples (wasted bits gone); mode: the arithmetic an LPC residual needs (lpc_mode)
template <int HIST, bool NARROW>
__device__ __forceinline__ uint32_t pq_values_of(const int32_t (&d)[HIST + FA_CHUNK], const PqSub &D, const int32_t *qv, uint32_t base, uint32_t bsf, uint32_t k,
                                                 bool pstart, int mode, uint32_t (&uz)[FA_CHUNK])
{
	if (D.type == FA_SUB_VERBATIM) {
		const uint32_t mask = pq_mask(D.sbps);
#pragma unroll
		for (int i = 0; i < FA_CHUNK; i++)
			uz[i] = (uint32_t)d[HIST + i] & mask;
		return umin32(FA_CHUNK, bsf - base) * D.sbps;
	}
	int32_t res[FA_CHUNK];
	if (D.type == FA_SUB_FIXED)
		fixed_residual16<HIST>(d, D.order, res);
	else
		lpc_residual16_any<HIST, NARROW>(d, qv, D.order, (int)D.shift, mode, res);
	uint32_t nb = pstart ? D.plen : 0u;
	const uint32_t k1 = k + 1u;
	// the frame's first `order` samples are the warm-up, and a short last block ends inside a chunk:
	// sample idx is coded when order <= idx < bsf, one unsigned comparison
	const uint32_t from = base - D.order, span = bsf - D.order;
#pragma unroll
	for (int i = 0; i < FA_CHUNK; i++) {
		uz[i] = ((uint32_t)res[i] << 1) ^ (uint32_t)(res[i] >> 31);
		const bool coded = from + (uint32_t)i < span;
		nb += coded ? (uz[i] >> k) + k1 : 0u;
	}
	return nb;
}

// The thread's run of a subframe into the window from bit p on (FLAC__bitwriter_write_rice_signed_block,
// bitwriter.c:544; verbatim samples: stream_encoder_framing.c:466).
__device__ __forceinline__ void pq_codes(uint32_t *W, const PqSub &D, const uint32_t (&uz)[FA_CHUNK], uint32_t base, uint32_t bsf, uint32_t k, bool pstart,
                                         uint32_t p)
{
	if (D.type == FA_SUB_VERBATIM) {
		const uint32_t lsh = 32u - D.sbps;
#pragma unroll
		for (int i = 0; i < FA_CHUNK; i++) {
			const bool inside = base + (uint32_t)i < bsf;
			pq_emit(W, p, inside ? uz[i] << lsh : 0u);
			p += inside ? D.sbps : 0u;
		}
		return;
	}
	if (pstart) {
		pq_emit(W, p, k << (32u - D.plen));
		p += D.plen;
	}
	const uint32_t kmask = (1u << k) - 1u, stop = 1u << k, lsh = 31u - k, k1 = k + 1u;
	const uint32_t from = base - D.order, span = bsf - D.order;
#pragma unroll
	for (int i = 0; i < FA_CHUNK; i++) {
		const bool coded = from + (uint32_t)i < span;
		const uint32_t q = p + (uz[i] >> k);
		const uint32_t vh = ((uz[i] & kmask) | stop) << lsh;
		pq_emit(W, coded ? q : p, coded ? vh : 0u);
		p = coded ? q + k1 : p;
	}
}

// Where a frame goes in the output is the sum of the sizes of the frames before it, and a frame's size is known well
// before its bits are (after the scan over the code lengths).  Every workgroup publishes its size in its ticket at that
// point (PQ_SIZE), and when its bits are ready one wave looks back over the tickets in front of it, 64 at a time,
// adding sizes until it meets a ticket that already holds a finished sum (PQ_SUM: that frame's own end), and then
// publishes its own.  Workgroups start in order and none waits before it has published its size, so the wait ends.
// This is what lets the frame leave the LDS window straight for its place: no slots, no scan and copy kernels behind.
#define PQ_SIZE (1ull << 62)
#define PQ_SUM (2ull << 62)
#define PQ_VALUE ((1ull << 62) - 1)
// WAIT = false: one look at the 64 tickets in front, no waiting; false if that does not settle it (nothing is published then).
// PUBLISH = false: only asks where frame f starts (for a workgroup that is not the one that knows the frame's size).
template <bool WAIT, bool PUBLISH = true>
__device__ __forceinline__ bool pq_place(unsigned long long *tickets, uint32_t f, uint32_t size, uint32_t lane, uint64_t &place, uint32_t *dbg = nullptr)
{
	uint64_t before = 0;
	int64_t at = (int64_t)f - 1;
	while (at >= 0) {
#ifdef PS_STAMPS
		if (dbg) dbg[0]++;
#endif
		const int64_t i = at - (int64_t)lane;
		// (in front of the first frame: a finished sum of nothing)
		const uint64_t t = i >= 0 ? __hip_atomic_load(&tickets[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : PQ_SUM;
		const uint64_t is_sum = __ballot((t >> 62) == 2), is_none = __ballot((t >> 62) == 0);
		// the tickets that count: up to and including the nearest finished sum
		const uint32_t nearest = is_sum ? (uint32_t)__ffsll((unsigned long long)is_sum) - 1u : 63u;
		const uint64_t upto = nearest == 63u ? ~0ull : (2ull << nearest) - 1ull;
		if (is_none & upto) {
			if (!WAIT)
				return false;
#ifdef PS_STAMPS
			if (dbg) { dbg[1]++; dbg[2] = (uint32_t)__popcll(is_none & upto); }
#endif
			__builtin_amdgcn_s_sleep(2);
			continue;
		}
		if (!WAIT && !is_sum)
			return false;
		uint64_t v = lane <= nearest ? (t & PQ_VALUE) : 0;
#pragma unroll
		for (int o = 32; o > 0; o >>= 1)
			v += __shfl_xor(v, o, 64);
		before += v;
		if (is_sum)
			break;
		at -= 64;
	}
	if (PUBLISH && lane == 0)
		__hip_atomic_store(&tickets[f], PQ_SUM | (before + size), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	place = before;
	return true;
}

#ifndef PQ_WAVES
#define PQ_WAVES 4   // workgroups per CU (one wave of each per SIMD)
#endif
// tickets: not null = the frame goes to its place in `out` (pq_place; its place also into offsets[f]), else into its slot
template <int HIST>
__global__ __launch_bounds__(FA_THREADS) __attribute__((amdgpu_waves_per_eu(PQ_WAVES, PQ_WAVES)))
void k_pack_frame(FaEncParams P, FaBatch B, const void *__restrict__ pcm, const FaSubStat *__restrict__ stats, const FaPlan *__restrict__ plans,
                  const uint32_t *__restrict__ assign, uint32_t *__restrict__ slots, uint32_t *__restrict__ frame_bytes, uint32_t *errflag,
                  uint8_t *__restrict__ out, uint64_t out_capacity, uint64_t *__restrict__ offsets, unsigned long long *tickets)
{
	__shared__ PqShared<HIST> S;
	const uint32_t f = blockIdx.x, tid = threadIdx.x, lane = tid & 63u;
	const uint32_t wv = __builtin_amdgcn_readfirstlane(tid >> 6);
	const uint32_t bsf = frame_len(B, P.blocksize, f);   // the stream's last block may be short
	const uint64_t frame_start = (uint64_t)f * P.blocksize;
	const uint32_t slot_words = P.slot_bytes / 4;
	uint32_t *slot = slots + (size_t)f * slot_words;
	constexpr uint32_t HP = ST_HP(HIST);
	constexpr uint32_t NIT = (HP + FA_TILE + FA_THREADS - 1) / FA_THREADS;
	const uint32_t *words = (const uint32_t *)pcm + frame_start;

	// ---- everything the frame needs from global memory, requested in one go ----
	{
		uint32_t pre_w[NIT];
#pragma unroll
		for (uint32_t i = 0; i < NIT; i++) {
			const uint32_t j = tid + i * FA_THREADS;
			int32_t idx = (int32_t)j - (int32_t)HP;
			idx = idx < 0 ? 0 : ((uint32_t)idx >= bsf ? (int32_t)bsf - 1 : idx);
			pre_w[i] = words[idx];
		}
		const FaPlan *fp = plans + (size_t)f * 4;
		uint8_t pre_params[4];
		int32_t pre_q[4] = {0, 0, 0, 0};
		uint4 pre_head = {0, 0, 0, 0};
		uint2 pre_stat = {0, 0};
#pragma unroll
		for (int sg = 0; sg < 4; sg++) {
			pre_params[sg] = fp[sg].params[tid];
			if (tid < FA_MAX_ORDER)
				pre_q[sg] = fp[sg].q[tid];
		}
		if (tid < 4) {
			pre_head = *(const uint4 *)(fp + tid);
			pre_stat.x = stats[(size_t)f * 4 + tid].wasted;
			pre_stat.y = stats[(size_t)f * 4 + tid].bps;
		}
		for (uint32_t e = tid; e < 4 * 256 / 2; e += FA_THREADS)
			((uint32_t *)&S.crc_tables[0][0])[e] = ((const uint32_t *)&c_crc16x4[0][0])[e];
		if (tid < 64)
			((uint32_t *)S.crc8)[tid] = ((const uint32_t *)c_crc8_table)[tid];
#pragma unroll
		for (int sg = 0; sg < 4; sg++) {
			S.params[sg][tid] = pre_params[sg];
			if (tid < FA_MAX_ORDER)
				S.q[sg][tid] = pre_q[sg];
		}
		if (tid < 4) {
			S.head[tid] = pre_head;
			S.stat[tid] = pre_stat;
		}
		const uint32_t n0 = HP + ((bsf + 15) & ~15u);
#pragma unroll
		for (uint32_t i = 0; i < NIT; i++) {
			const uint32_t j = tid + i * FA_THREADS;
			if (j < n0)
				S.u.T[j + (j >> 4)] = (j >= HP && j - HP < bsf) ? (int32_t)pre_w[i] : 0;
		}
	}
	const uint32_t ca = __builtin_amdgcn_readfirstlane(assign[f]);
	__syncthreads();

	// ---- the two subframes in written order (stream_encoder.c:3091-3151) ----
	uint32_t sig0 = 0, sig1 = 1;
	if (ca == 1) sig1 = 3;                    // left, side
	else if (ca == 2) sig0 = 3;               // side, right
	else if (ca == 3) { sig0 = 2; sig1 = 3; } // mid, side
	const PqSub D0 = pq_sub(S, sig0, bsf), D1 = pq_sub(S, sig1, bsf);

	const uint32_t base = tid * FA_CHUNK;
	const bool have = base < bsf;
	uint32_t uz0[FA_CHUNK], uz1[FA_CHUNK];
	uint32_t nb0 = 0, nb1 = 0, k0 = 0, k1 = 0;
	bool ps0 = false, ps1 = false;
	uint32_t warm = 0;   // waves 0 and 1: warm-up sample `lane` of subframe 0 / 1
	{
		int32_t raw[HIST + FA_CHUNK];
		if (have)
			load_chunk_lds<HIST>(raw, S.u.T);
		if (wv < 2) {
			const PqSub &D = wv == 0 ? D0 : D1;
			if (lane < D.order) {
				const uint32_t j = HP + lane;
				warm = (uint32_t)(quad_signal((uint32_t)S.u.T[j + (j >> 4)], D.sig) >> D.wasted);
			}
		}
		if (have && D0.type != FA_SUB_CONSTANT) {
			if (D0.type != FA_SUB_VERBATIM) {
				const uint32_t part = base / D0.psize;
				k0 = S.params[sig0][part];
				ps0 = base == part * D0.psize;
			}
			nb0 = pq_values<HIST>(raw, D0, S.q[sig0], base, bsf, k0, ps0, uz0);
		}
		if (have && D1.type != FA_SUB_CONSTANT) {
			if (D1.type != FA_SUB_VERBATIM) {
				const uint32_t part = base / D1.psize;
				k1 = S.params[sig1][part];
				ps1 = base == part * D1.psize;
			}
			nb1 = pq_values<HIST>(raw, D1, S.q[sig1], base, bsf, k1, ps1, uz1);
		}
	}

	// ---- where every run starts: one block scan over the pair of lengths ----
	const uint32_t in0 = sw_scan_incl(nb0), in1 = sw_scan_incl(nb1);
	if (lane == 63) {
		S.scan[wv][0] = in0;
		S.scan[wv][1] = in1;
	}
	__syncthreads();   // (also: every thread has read what it needs of the staged samples)
	uint32_t off0 = 0, off1 = 0, total0 = 0, total1 = 0;
#pragma unroll
	for (uint32_t w = 0; w < 4; w++) {
		const uint32_t a = __builtin_amdgcn_readfirstlane(S.scan[w][0]), b = __builtin_amdgcn_readfirstlane(S.scan[w][1]);
		off0 += w < wv ? a : 0u;
		off1 += w < wv ? b : 0u;
		total0 += a;
		total1 += b;
	}
	// FLAC__frame_add_header (stream_encoder_framing.c:219); its CRC-8 is one more byte
	uint64_t hw[2];
	uint32_t hn = frame_header_bytes(P, bsf, B.first_frame + f, ca, hw);
	const uint32_t pos0 = 8 * (hn + 1);
	const uint32_t pos1 = pos0 + D0.hbits + total0;
	const uint32_t bitpos = pos1 + D1.hbits + total1;
	const uint32_t nbytes = (bitpos + 7) >> 3;           // zero padding to a whole byte (stream_encoder.c:2885)
	const uint32_t total_words = (nbytes + 2 + 3) / 4;   // with the CRC-16
	const uint32_t cap = slot_words < PQ_CAPW ? slot_words : PQ_CAPW;
	const bool ragged = (D0.po && (D0.psize % FA_CHUNK || D0.psize == 0)) || (D1.po && (D1.psize % FA_CHUNK || D1.psize == 0));
	if (total_words + 2 > cap || ragged) {
		// (cannot happen: the slot is sized by slot_bytes_for, a full block's partitions are whole chunks by
		// pack_frame_eligible and a short block has one partition, stream_encoder.c:2929; loud if it ever does)
		if (tid == 0) {
			atomicOr(errflag, 1u);
			frame_bytes[f] = 0;
		}
		if (tickets && wv == 0) {
			uint64_t before = 0;
			pq_place<true>(tickets, f, 0, lane, before);
			if (lane == 0) {
				offsets[f] = before;
				if (f == B.nframes - 1)
					offsets[B.nframes] = before;
			}
		}
		return;
	}
	// The place: tried without waiting as soon as the size is known (a wait at the end of the frame is time in which the
	// workgroup holds its share of the CU for nothing; a sum published early also spares the frames behind their wait),
	// again after the codes are out, and waited for only before the bits leave.
	bool placed = false;
	uint64_t place = 0;
	if (tickets && wv == 0) {
		if (lane == 0)
			__hip_atomic_store(&tickets[f], PQ_SIZE | (uint64_t)(nbytes + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		placed = pq_place<false>(tickets, f, nbytes + 2, lane, place);
	}
	uint32_t *W = S.u.W;
	for (uint32_t i = tid; i < (total_words + 2 + 3) / 4; i += FA_THREADS)
		((uint4 *)W)[i] = make_uint4(0, 0, 0, 0);
	__syncthreads();

	// ---- everything into the window ----
	if (have) {
		if (D0.type != FA_SUB_CONSTANT)
			pq_codes(W, D0, uz0, base, bsf, k0, ps0, pos0 + D0.hbits + off0 + (in0 - nb0));
		if (D1.type != FA_SUB_CONSTANT)
			pq_codes(W, D1, uz1, base, bsf, k1, ps1, pos1 + D1.hbits + off1 + (in1 - nb1));
	}
	{
		// subframe headers (stream_encoder_framing.c:366-479): waves 0 and 1 write the warm-up samples and
		// the small fields of subframe 0 and 1, waves 2 and 3 their coefficients
		const PqSub &D = (wv & 1u) == 0 ? D0 : D1;
		const uint32_t pos = (wv & 1u) == 0 ? pos0 : pos1;
		const bool predicted = D.type == FA_SUB_FIXED || D.type == FA_SUB_LPC;
		const uint32_t after_warm = pos + 8 + D.wasted + D.order * D.sbps;
		if (wv < 2) {
			if (predicted && lane < D.order)
				pq_emit(W, pos + 8 + D.wasted + lane * D.sbps, (warm & pq_mask(D.sbps)) << (32u - D.sbps));
			if (lane == 63) {
				uint32_t typebits;
				switch (D.type) {
				case FA_SUB_CONSTANT: typebits = 0x00; break;
				case FA_SUB_VERBATIM: typebits = 0x02; break;
				case FA_SUB_FIXED: typebits = 0x10 | (D.order << 1); break;
				default: typebits = 0x40 | ((D.order - 1) << 1); break;
				}
				pq_emit(W, pos, (typebits | (D.wasted ? 1u : 0u)) << 24);
				if (D.wasted)
					pq_emit(W, pos + 8 + D.wasted - 1, 0x80000000u);
				if (D.type == FA_SUB_CONSTANT)
					pq_emit(W, pos + 8 + D.wasted, (D.constant & pq_mask(D.sbps)) << (32u - D.sbps));
				if (D.type == FA_SUB_LPC)
					pq_emit(W, after_warm, (((D.precision - 1u) << 5) | (D.shift & 31u)) << 23);
				if (predicted)
					pq_emit(W, pos + D.hbits - 6, (((D.rice2 ? 1u : 0u) << 4) | D.po) << 26);
			}
		} else if (D.type == FA_SUB_LPC && lane < D.order)
			pq_emit(W, after_warm + 9 + lane * D.precision, ((uint32_t)S.q[D.sig][lane] & pq_mask(D.precision)) << (32u - D.precision));
	}
	if (tid == FA_THREADS - 1) {
		uint32_t crc = 0;  // FLAC__crc8 (crc.c:124)
		for (uint32_t i = 0; i < hn; i++)
			crc = S.crc8[crc ^ hdr_get(hw, i)];
		hdr_put(hw, hn, crc);
#pragma unroll
		for (uint32_t i = 0; i < 4; i++)
			if (4 * i < hn)
				pq_emit(W, 32 * i, (uint32_t)(hw[i >> 1] >> ((i & 1u) ? 0 : 32)));   // (the bytes behind the header read as zero)
	}
	if (tickets && wv == 0 && !placed)
		placed = pq_place<false>(tickets, f, nbytes + 2, lane, place);
	__syncthreads();

	// ---- CRC-16 over everything so far (stream_encoder.c:2890-2900), then out to the frame's slot ----
	const uint32_t crc16 = window_crc16(W, S.crc_tables, S.crcs, nbytes);
	if (tid == 0)
		pq_emit(W, nbytes * 8, crc16 << 16);
	if (tickets) {
		// ---- straight to the frame's place: bytes up to a 16-byte boundary of the output, whole vectors, the rest ----
		if (wv == 0) {
			if (!placed)
				pq_place<true>(tickets, f, nbytes + 2, lane, place);
			if (lane == 0) {
				S.place = place;
				offsets[f] = place;
				if (f == B.nframes - 1)
					offsets[B.nframes] = place + nbytes + 2;
				frame_bytes[f] = nbytes + 2;
			}
		}
		__syncthreads();
		const uint64_t before = S.place;
		const uint32_t total = nbytes + 2;
		if (before + total > out_capacity) {
			if (tid == 0)
				atomicOr(errflag, 2u);
			return;
		}
		uint8_t *dst = out + before;
		const uint32_t head = umin32((uint32_t)((16 - ((uintptr_t)dst & 15)) & 15), total);
		const uint32_t nvec = (total - head) / 16, tail0 = head + 16 * nvec;
		// byte k of the frame is byte k mod 4, counted from the top, of window word k / 4
		if (tid < head)
			dst[tid] = (uint8_t)(W[tid >> 2] >> (24 - 8 * (tid & 3)));
		if (tid >= FA_THREADS - 16 && tail0 + (tid - (FA_THREADS - 16)) < total) {
			const uint32_t k = tail0 + (tid - (FA_THREADS - 16));
			dst[k] = (uint8_t)(W[k >> 2] >> (24 - 8 * (k & 3)));
		}
		const uint32_t sh = head & 3, w0 = head >> 2;
		for (uint32_t i = tid; i < nvec; i += FA_THREADS) {
			uint32_t w[5];
#pragma unroll
			for (int j = 0; j < 5; j++)
				w[j] = W[w0 + 4 * i + j];
			uint4 v;
			// (a funnel shift by 32 - 8 sh bits; by nothing when the places agree modulo four)
			v.x = __builtin_bswap32(sh ? __builtin_amdgcn_alignbit(w[0], w[1], 32 - 8 * sh) : w[0]);
			v.y = __builtin_bswap32(sh ? __builtin_amdgcn_alignbit(w[1], w[2], 32 - 8 * sh) : w[1]);
			v.z = __builtin_bswap32(sh ? __builtin_amdgcn_alignbit(w[2], w[3], 32 - 8 * sh) : w[2]);
			v.w = __builtin_bswap32(sh ? __builtin_amdgcn_alignbit(w[3], w[4], 32 - 8 * sh) : w[3]);
			((uint4 *)(dst + head))[i] = v;
		}
		return;
	}
	__syncthreads();
	for (uint32_t i = tid; i < (total_words + 3) / 4; i += FA_THREADS) {
		uint4 v = ((const uint4 *)W)[i];
		v.x = __builtin_bswap32(v.x);
		v.y = __builtin_bswap32(v.y);
		v.z = __builtin_bswap32(v.z);
		v.w = __builtin_bswap32(v.w);
		((uint4 *)slot)[i] = v;
	}
	if (tid == 0)
		frame_bytes[f] = nbytes + 2;
}


// ---------------------------------------------------------------------------------------------
// k_pack_sub: the frame writer for every other stream whose blocks fit one pass (any channel count, any
// sample width, with or without mid/side): a workgroup per WRITTEN SUBFRAME, so that a frame of eight
// channels is eight workgroups side by side and not one that walks them in turn.  A workgroup's piece
// of the frame is its subframe, with the frame header in front for the first channel and the padding
// and CRC-16 behind for the last.  Pieces meet at any bit, so three things pass between the
// workgroups of a frame, each through a word of `subt` (two per piece, cleared by k_decide):
//   - the piece's length in bits, as soon as the scan over the code lengths has it; the pieces in
//     front are then added up to find where this one starts, and its bits go into the LDS window at
//     that position modulo 8, so that window bytes are output bytes;
//   - the byte in which the piece ends, if it ends inside one: the next piece ORs it into its first
//     byte and writes that byte, the piece itself writes whole bytes only;
//   - the CRC-16 of the piece's bytes (own bits only, which is what makes the sum work: the CRC is
//     linear).  The last channel's workgroup shifts every piece's CRC to the end of the frame
//     (multiplication by a power of x) and XORs them.
// The frame's place in the output comes from the tickets as in k_pack_frame: the last channel's
// workgroup, which knows the frame's size, publishes it and later the running sum; the others only
// ask.  Every wait is for a workgroup of an older frame or of the same frame, workgroups start in
// order, and nobody waits before its length is out, so the waits end.
// ---------------------------------------------------------------------------------------------
#define PS_VALID 0x80000000u

template <int HIST>
struct PsShared {
	union {
		uint32_t W[PQ_CAPW];
		int32_t T[ST_WORDS(HIST)];
	} u __attribute__((aligned(16)));
	uint16_t crc_tables[4][256];
	uint32_t crcs[4];
	uint32_t scan[4];
	uint8_t crc8[256] __attribute__((aligned(4)));
	uint8_t params[FA_MAX_PARTS];
	int32_t q[FA_MAX_ORDER];
	uint64_t place;
	uint32_t start_bit;   // of the piece, counted from the frame's first bit
};

__device__ __forceinline__ uint32_t ps_wait(const uint32_t *word)
{
	uint32_t v;
	while (!((v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & PS_VALID))
		__builtin_amdgcn_s_sleep(1);
	return v;
}

#ifndef PS_WAVES
#define PS_WAVES 6   // workgroups per CU (3 / 4 / 5 / 6: 0.67 / 0.55 / 0.49 / 0.45 ms for eight channels of 24 bits: the kernel waits, see below; with a smaller window seven or eight fit, and spill: 0.31 -> 0.36 / 0.45)
#endif
template <int HIST, bool NARROW>
__global__ __launch_bounds__(FA_THREADS) __attribute__((amdgpu_waves_per_eu(PS_WAVES, PS_WAVES)))
void k_pack_sub(FaEncParams P, FaBatch B, const void *__restrict__ pcm, const FaSubStat *__restrict__ stats, const FaPlan *__restrict__ plans,
                const uint32_t *__restrict__ assign, uint32_t *__restrict__ frame_bytes, uint32_t *errflag, uint8_t *__restrict__ out,
                uint64_t out_capacity, uint64_t *__restrict__ offsets, unsigned long long *tickets, uint32_t *subt, unsigned long long *stamps)
{
// debugging aid (builds with -DPS_STAMPS): thread 0's time at point k goes to stamps[block][k] (the 100 MHz wall clock, which all
// dies share, scaled to shader cycles)
#ifdef PS_STAMPS
#define PSTAMP(k) do { if (stamps && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 16 + (k)] = wall_clock64() * 24; } while (0)
#else
#define PSTAMP(k) do { } while (0)
#endif
	PSTAMP(0);
	__shared__ PsShared<HIST> S;
	const uint32_t ch = P.channels;
	const uint32_t f = blockIdx.x / ch, tid = threadIdx.x, lane = tid & 63u;
	// Workgroups go to the chip's eight dies in turn.  Were workgroup j of a frame always channel j, one die would get
	// every frame's last piece (which waits for all the others) and the rest would wait for that die; so the channels
	// of a frame are dealt starting one further on each time the deal comes round to the same dies.  (A workgroup may
	// now wait for one of the up to seven that follow it in its own frame; workgroups still start in order, so those
	// are never held back by anything but older frames.)
	const uint32_t low = ch & (0u - ch), period = 8u / (low < 8u ? low : 8u);
	// (with two channels the pieces in dispatch order are the faster: 0.40 against 0.42 ms for 24-bit stereo)
	const uint32_t c = ch <= 2 ? blockIdx.x - f * ch : (blockIdx.x - f * ch + f / period) % ch;
	const uint32_t wv = __builtin_amdgcn_readfirstlane(tid >> 6);
	const bool first_piece = c == 0, last_piece = c == ch - 1;
	const uint32_t bsf = frame_len(B, P.blocksize, f);
	const uint64_t frame_start = (uint64_t)f * P.blocksize;
	constexpr uint32_t HP = ST_HP(HIST);
	constexpr uint32_t NIT = (HP + FA_TILE + FA_THREADS - 1) / FA_THREADS;
	uint32_t *my_words = subt + 2 * ((size_t)f * ch + c);
	const uint32_t *frame_words = subt + 2 * (size_t)f * ch;

	// ---- which signal this subframe is (stream_encoder.c:3091-3151), and everything it needs from memory ----
	// (without mid/side nothing else waits for this load: the assignment is "independent" then, stream_encoder.c:3091)
	const uint32_t ca = P.do_ms ? __builtin_amdgcn_readfirstlane(assign[f]) : 0u;
	uint32_t sig = c;
	if (ca == 1 && c == 1) sig = ch + 1;       // left, side
	else if (ca == 2 && c == 0) sig = ch + 1;  // side, right
	else if (ca == 3) sig = ch + c;            // mid, side
	const FaPlan *pl = plans + (size_t)f * P.nsignals + sig;
	PqSub D;
	int mode;
	{
		int32_t v[NIT];
		stage_load<HIST>(v, pcm, ch, B.sample_bytes, fa_layout(B), frame_start, bsf, sig, 0);
		const uint8_t pre_param = pl->params[tid];
		const int32_t pre_q = tid < FA_MAX_ORDER ? pl->q[tid] : 0;
		const uint4 hd = *(const uint4 *)pl;
		const FaSubStat *st = stats + (size_t)f * P.nsignals + sig;
		const uint32_t h1 = __builtin_amdgcn_readfirstlane(hd.y), h2 = __builtin_amdgcn_readfirstlane(hd.z);
		D.sig = sig;
		D.type = h1 & 0xff;
		D.order = (h1 >> 8) & 0xff;
		D.precision = (h1 >> 16) & 0xff;
		D.shift = (uint32_t)(int32_t)(int8_t)(h1 >> 24);
		D.po = h2 & 0xff;
		D.rice2 = (h2 >> 8) & 0xff;
		mode = (int)((h2 >> 16) & 0xff);
		D.constant = __builtin_amdgcn_readfirstlane(hd.w);
		D.wasted = __builtin_amdgcn_readfirstlane(st->wasted);
		D.sbps = __builtin_amdgcn_readfirstlane(st->bps);
		D.plen = D.rice2 ? 5 : 4;
		D.psize = bsf >> D.po;
		// FLAC__subframe_add_* (stream_encoder_framing.c:366-479)
		D.hbits = 8 + D.wasted;
		if (D.type == FA_SUB_CONSTANT)
			D.hbits += D.sbps;
		else if (D.type == FA_SUB_FIXED)
			D.hbits += D.order * D.sbps + 6;
		else if (D.type == FA_SUB_LPC)
			D.hbits += D.order * (D.sbps + D.precision) + 9 + 6;
		for (uint32_t e = tid; e < 4 * 256 / 2; e += FA_THREADS)
			((uint32_t *)&S.crc_tables[0][0])[e] = ((const uint32_t *)&c_crc16x4[0][0])[e];
		if (tid < 64)
			((uint32_t *)S.crc8)[tid] = ((const uint32_t *)c_crc8_table)[tid];
		S.params[tid] = pre_param;
		if (tid < FA_MAX_ORDER)
			S.q[tid] = pre_q;
		PSTAMP(1);
		stage_store<HIST>(S.u.T, v, bsf, D.wasted, 0);
	}
	__syncthreads();
	PSTAMP(2);

	// ---- what every thread will code, and how long that is ----
	const uint32_t base = tid * FA_CHUNK;
	const bool ragged = D.type != FA_SUB_CONSTANT && D.type != FA_SUB_VERBATIM && D.po && (D.psize % FA_CHUNK || D.psize == 0);
	const bool have = base < bsf && !ragged;
	uint32_t uz[FA_CHUNK];
	uint32_t nb = 0, k = 0, warm = 0;
	bool ps = false;
	{
		int32_t d[HIST + FA_CHUNK];
		if (have)
			load_chunk_lds<HIST>(d, S.u.T);
		if (wv == 0 && lane < D.order) {
			const uint32_t j = HP + lane;
			warm = (uint32_t)S.u.T[j + (j >> 4)];
		}
		if (have && D.type != FA_SUB_CONSTANT) {
			if (D.type != FA_SUB_VERBATIM) {
				const uint32_t part = base / D.psize;
				k = S.params[part];
				ps = base == part * D.psize;
			}
			nb = pq_values_of<HIST, NARROW>(d, D, S.q, base, bsf, k, ps, mode, uz);
		}
	}
	const uint32_t in = sw_scan_incl(nb);
	PSTAMP(3);
	if (lane == 63)
		S.scan[wv] = in;
	__syncthreads();   // (also: every thread has read what it needs of the staged samples)
	PSTAMP(4);
	uint32_t off = 0, total = 0;
#pragma unroll
	for (uint32_t w = 0; w < 4; w++) {
		const uint32_t a = __builtin_amdgcn_readfirstlane(S.scan[w]);
		off += w < wv ? a : 0u;
		total += a;
	}
	// FLAC__frame_add_header (stream_encoder_framing.c:219); its CRC-8 is one more byte
	uint64_t hw[2] = {0, 0};
	uint32_t hn = 0;
	if (first_piece)
		hn = frame_header_bytes(P, bsf, B.first_frame + f, ca, hw);
	const uint32_t front = first_piece ? 8 * (hn + 1) : 0u;
	uint32_t piece_bits = front + D.hbits + total;
	// (cannot happen: pack_sub_eligible sized the window for the largest subframe the search can choose, a full block's
	// partitions are whole chunks and a short block has one, stream_encoder.c:2929; loud if it ever does - the piece is
	// then written as one empty byte so that the workgroups that wait for it get on)
	const bool bad = ragged || (piece_bits + 7 + 7 + 16 + 31) / 32 + 3 > PQ_CAPW;
	if (bad) {
		piece_bits = front + 8;
		if (tid == 0)
			atomicOr(errflag, 1u);
	}
	if (tid == 0)
		__hip_atomic_store(&my_words[0], PS_VALID | piece_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

	// ---- where the piece starts: the pieces in front of it in the frame ----
	// A workgroup's time is mostly round trips to other workgroups' words, so they are taken side by side: wave 0 talks to
	// the frame's other pieces, wave 1 to the tickets of the frames in front (and, in the last piece, adds up the frame as
	// well, rather than wait for wave 0 to tell it).
	uint32_t before_bits = 0, len_l = 0;   // (waves 0 and 1) lane l < c: the length of piece l and the bits in front of it
	bool placed = false;                   // (wave 1)
	uint64_t place = 0;
	uint32_t nbytes = 0;                   // last piece: the frame's bytes without its CRC-16
	if (wv == 0 || (wv == 1 && last_piece)) {
		len_l = lane < c ? (ps_wait(&frame_words[2 * lane]) & ~PS_VALID) : 0u;
		const uint32_t incl = sw_scan_incl(len_l);
		before_bits = incl - len_l;
		const uint32_t start = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
		if (wv == 0 && lane == 0)
			S.start_bit = start;
		if (last_piece) {
			nbytes = (start + piece_bits + 7) >> 3;   // zero padding to a whole byte (stream_encoder.c:2885)
			if (wv == 1 && lane == 0)
				__hip_atomic_store(&tickets[f], PQ_SIZE | (uint64_t)(nbytes + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
	if (wv == 1)
		placed = last_piece ? pq_place<false>(tickets, f, nbytes + 2, lane, place) : pq_place<false, false>(tickets, f, 0, lane, place);
	uint32_t *W = S.u.W;
	{
		const uint32_t nw = (piece_bits + 7 + 7 + 16 + 31) / 32 + 3;   // at most PQ_CAPW, see above
		for (uint32_t i = tid; i < (nw + 3) / 4 && i < PQ_CAPW / 4; i += FA_THREADS)
			((uint4 *)W)[i] = make_uint4(0, 0, 0, 0);
	}
	PSTAMP(5);
	__syncthreads();
	PSTAMP(6);
	const uint32_t start_bit = S.start_bit;
	const uint32_t r = start_bit & 7u, first_byte = start_bit >> 3;   // window bit 0 is bit 0 of frame byte first_byte
	const uint32_t pos = r + front;                                   // the subframe's first bit in the window

	// ---- everything into the window ----
	if (!bad) {
		if (have && D.type != FA_SUB_CONSTANT)
			pq_codes(W, D, uz, base, bsf, k, ps, pos + D.hbits + off + (in - nb));
		// the subframe header (stream_encoder_framing.c:366-479): wave 0 the warm-up samples and the small fields, wave 2 the coefficients
		const bool predicted = D.type == FA_SUB_FIXED || D.type == FA_SUB_LPC;
		const uint32_t after_warm = pos + 8 + D.wasted + D.order * D.sbps;
		if (wv == 0) {
			if (predicted && lane < D.order)
				pq_emit(W, pos + 8 + D.wasted + lane * D.sbps, (warm & pq_mask(D.sbps)) << (32u - D.sbps));
			if (lane == 63) {
				uint32_t typebits;
				switch (D.type) {
				case FA_SUB_CONSTANT: typebits = 0x00; break;
				case FA_SUB_VERBATIM: typebits = 0x02; break;
				case FA_SUB_FIXED: typebits = 0x10 | (D.order << 1); break;
				default: typebits = 0x40 | ((D.order - 1) << 1); break;
				}
				pq_emit(W, pos, (typebits | (D.wasted ? 1u : 0u)) << 24);
				if (D.wasted)
					pq_emit(W, pos + 8 + D.wasted - 1, 0x80000000u);
				if (D.type == FA_SUB_CONSTANT)
					pq_emit(W, pos + 8 + D.wasted, (D.constant & pq_mask(D.sbps)) << (32u - D.sbps));
				if (D.type == FA_SUB_LPC)
					pq_emit(W, after_warm, (((D.precision - 1u) << 5) | (D.shift & 31u)) << 23);
				if (predicted)
					pq_emit(W, pos + D.hbits - 6, (((D.rice2 ? 1u : 0u) << 4) | D.po) << 26);
			}
		} else if (wv == 2 && D.type == FA_SUB_LPC && lane < D.order)
			pq_emit(W, after_warm + 9 + lane * D.precision, ((uint32_t)S.q[lane] & pq_mask(D.precision)) << (32u - D.precision));
	}
	if (first_piece && tid == FA_THREADS - 1) {
		uint32_t crc = 0;  // FLAC__crc8 (crc.c:124)
		for (uint32_t i = 0; i < hn; i++)
			crc = S.crc8[crc ^ hdr_get(hw, i)];
		hdr_put(hw, hn, crc);
#pragma unroll
		for (uint32_t i = 0; i < 4; i++)
			if (4 * i < hn)
				pq_emit(W, 32 * i, (uint32_t)(hw[i >> 1] >> ((i & 1u) ? 0 : 32)));   // (the bytes behind the header read as zero)
	}
	if (wv == 1 && !placed)
		placed = last_piece ? pq_place<false>(tickets, f, nbytes + 2, lane, place) : pq_place<false, false>(tickets, f, 0, lane, place);
	PSTAMP(7);
	__syncthreads();
	PSTAMP(8);

	// ---- the CRC-16 of the piece's bytes; the last piece gathers them all (stream_encoder.c:2890-2900) ----
	const uint32_t end_bit = r + piece_bits;                 // in the window
	const uint32_t piece_bytes = (end_bit + 7) >> 3;
	const uint32_t own_crc = window_crc16(W, S.crc_tables, S.crcs, piece_bytes);
	PSTAMP(9);
	if (!last_piece) {
		if (tid == 0) {
			const uint32_t kb = piece_bytes - 1;
			const uint32_t tail = (end_bit & 7u) ? ((W[kb >> 2] >> (24 - 8 * (kb & 3))) & 0xffu) : 0u;
			__hip_atomic_store(&my_words[1], PS_VALID | (own_crc << 8) | tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
	if (wv == 0) {
		// what the pieces in front leave behind: lane l < c, piece l
		const uint32_t tw = (lane < c && (last_piece || lane == c - 1)) ? ps_wait(&frame_words[2 * lane + 1]) : 0u;
		if (last_piece) {
			// piece l covers the frame's bytes [before_bits >> 3, that + ((before_bits & 7) + len_l + 7) >> 3)
			const uint32_t after = nbytes - (before_bits >> 3) - (((before_bits & 7u) + len_l + 7) >> 3);
			uint32_t part = lane < c ? gf16_mul((tw >> 8) & 0xffffu, gf16_xpow8(after)) : 0u;
#pragma unroll
			for (int o = 32; o > 0; o >>= 1)
				part ^= (uint32_t)__shfl_xor((int)part, o, 64);
			if (lane == 0)
				pq_emit(W, piece_bytes * 8, (part ^ own_crc) << 16);
		}
		if (r && lane == c - 1)
			atomicOr(&W[0], (tw & 0xffu) << 24);
	} else if (wv == 1) {
		// ---- the frame's place ----
#ifdef PS_STAMPS
		if (stamps && lane == 0) {
			stamps[(size_t)blockIdx.x * 16 + 13] = wall_clock64() * 24;
			stamps[(size_t)blockIdx.x * 16 + 15] = placed ? 1 : 0;
		}
#endif
		uint32_t dbg[3] = {0, 0, 0};
		if (!placed) {
			if (last_piece)
				pq_place<true>(tickets, f, nbytes + 2, lane, place, dbg);
			else
				pq_place<true, false>(tickets, f, 0, lane, place, dbg);
		}
#ifdef PS_STAMPS
		if (stamps && lane == 0) {
			stamps[(size_t)blockIdx.x * 16 + 14] = wall_clock64() * 24;
			stamps[(size_t)blockIdx.x * 16 + 15] = (placed ? 1 : 0) | ((uint64_t)dbg[0] << 8) | ((uint64_t)dbg[1] << 24) | ((uint64_t)dbg[2] << 48);
		}
#endif
		if (lane == 0) {
			S.place = place;
			if (last_piece) {
				offsets[f] = place;
				if (f == B.nframes - 1)
					offsets[B.nframes] = place + nbytes + 2;
				frame_bytes[f] = nbytes + 2;
			}
		}
	}
	PSTAMP(10);
	__syncthreads();
	PSTAMP(11);

	// ---- out: the piece's whole bytes (the last piece ends on one, CRC-16 included) ----
	const uint32_t count = last_piece ? piece_bytes + 2 : end_bit >> 3;
	const uint64_t at = S.place + first_byte;
	if (at + count > out_capacity) {
		if (tid == 0)
			atomicOr(errflag, 2u);
		return;
	}
	uint8_t *dst = out + at;
	const uint32_t head = umin32((uint32_t)((16 - ((uintptr_t)dst & 15)) & 15), count);
	const uint32_t nvec = (count - head) / 16, tail0 = head + 16 * nvec;
	// byte k of the piece is byte k mod 4, counted from the top, of window word k / 4
	if (tid < head)
		dst[tid] = (uint8_t)(W[tid >> 2] >> (24 - 8 * (tid & 3)));
	if (tid >= FA_THREADS - 16 && tail0 + (tid - (FA_THREADS - 16)) < count) {
		const uint32_t kk = tail0 + (tid - (FA_THREADS - 16));
		dst[kk] = (uint8_t)(W[kk >> 2] >> (24 - 8 * (kk & 3)));
	}
	const uint32_t sh = head & 3, w0 = head >> 2;
	for (uint32_t i = tid; i < nvec; i += FA_THREADS) {
		uint32_t w[5];
#pragma unroll
		for (int j = 0; j < 5; j++)
			w[j] = W[w0 + 4 * i + j];
		uint4 v;
		// (a funnel shift by 32 - 8 sh bits; by nothing when the places agree modulo four)
		v.x = __builtin_bswap32(sh ? __builtin_amdgcn_alignbit(w[0], w[1], 32 - 8 * sh) : w[0]);
		v.y = __builtin_bswap32(sh ? __builtin_amdgcn_alignbit(w[1], w[2], 32 - 8 * sh) : w[1]);
		v.z = __builtin_bswap32(sh ? __builtin_amdgcn_alignbit(w[2], w[3], 32 - 8 * sh) : w[2]);
		v.w = __builtin_bswap32(sh ? __builtin_amdgcn_alignbit(w[3], w[4], 32 - 8 * sh) : w[3]);
		((uint4 *)(dst + head))[i] = v;
	}
	PSTAMP(12);
#undef PSTAMP
}

// Whether k_pack_sub can write this stream's frames.
static bool pack_sub_eligible(const FaEncParams &P, const FaBatch &B)
{
	if (P.blocksize > FA_TILE || P.blocksize % FA_CHUNK || P.channels > 32)
		return false;
	uint32_t reach = 0;
	while (reach < P.max_po && !((P.blocksize >> reach) & 1u))
		reach++;
	if ((P.blocksize >> reach) % FA_CHUNK)
		return false;
	// The largest subframe: a chosen subframe's estimate never exceeds the verbatim estimate (of a side signal, one bit wider),
	// and its written size exceeds its estimate by at most half a bit per sample plus one bit per partition; a frame header is
	// at most 16 bytes.  Behind it the window holds the padding, the CRC-16 and the words a field may reach into.
	const uint64_t bits = (uint64_t)P.blocksize * (P.bps + 3) + 4096 + 8 * 17 + 64;
	return (bits + 7 + 7 + 16 + 31) / 32 + 3 <= PQ_CAPW;
}

// Whether k_pack_frame can write this stream's frames.
static bool pack_frame_eligible(const FaEncParams &P, const FaBatch &B)
{
	if (!(B.pairs && P.nsignals == 4 && fa_narrow_only(P)))
		return false;
	if (P.blocksize > FA_TILE || P.blocksize % FA_CHUNK)
		return false;
	// the smallest partition a full block can be cut into (stream_encoder.c:2929-2936) is whole chunks
	uint32_t reach = 0;
	while (reach < P.max_po && !((P.blocksize >> reach) & 1u))
		reach++;
	if ((P.blocksize >> reach) % FA_CHUNK)
		return false;
	return P.slot_bytes / 4 <= PQ_CAPW;
}

bool fa_launch_pack(const FaEncParams &P, const FaBatch &B, const void *pcm, const FaSubStat *stats, const FaPlan *plans, const uint32_t *assign,
                    uint32_t *slots, uint32_t *frame_bytes, uint32_t *errflag, unsigned long long *stamps, uint8_t *out, uint64_t out_capacity,
                    uint64_t *offsets, unsigned long long *tickets, hipStream_t st)
{
	const dim3 block(FA_THREADS);
	const bool narrow = fa_narrow_only(P);
	const bool quad = B.pairs && P.nsignals == 4;
	const int hist = fa_hist_class(P);
	// Full frames of the common shape go to k_pack_frame (FLAC_AMD_PACK=generic keeps every frame on k_pack, for
	// comparisons); first: the first frame k_pack still has to write.
	uint32_t first = 0;
	{
		const char *mode = getenv("FLAC_AMD_PACK");
#ifdef PS_STAMPS
		const bool generic_only = mode && !strcmp(mode, "generic");
#else
		const bool generic_only = (mode && !strcmp(mode, "generic")) || stamps;
#endif
		const uint32_t nfull = B.nframes;   // (a short last block is one partition: it fits too)
		// FLAC_AMD_PACK=sub: k_pack_sub also where k_pack_frame would do (for comparisons)
		if (!generic_only && !(mode && !strcmp(mode, "sub")) && nfull && pack_frame_eligible(P, B)) {
			const dim3 grid(nfull);
			// FLAC_AMD_PACK=slots: k_pack_frame into the slots, scan and copy kernels behind it (for comparisons)
			unsigned long long *tk = (mode && !strcmp(mode, "slots")) ? nullptr : tickets;
			switch (hist) {
			case 4: hipLaunchKernelGGL(k_pack_frame<4>, grid, block, 0, st, P, B, pcm, stats, plans, assign, slots, frame_bytes, errflag, out, out_capacity, offsets, tk); break;
			case 12: hipLaunchKernelGGL(k_pack_frame<12>, grid, block, 0, st, P, B, pcm, stats, plans, assign, slots, frame_bytes, errflag, out, out_capacity, offsets, tk); break;
			default: hipLaunchKernelGGL(k_pack_frame<32>, grid, block, 0, st, P, B, pcm, stats, plans, assign, slots, frame_bytes, errflag, out, out_capacity, offsets, tk); break;
			}
			return tk != nullptr;
		}
	}
	{
		const char *mode = getenv("FLAC_AMD_PACK");
#ifdef PS_STAMPS
		const bool sub_ok = true;
#else
		const bool sub_ok = !stamps;
#endif
		if ((!mode || !strcmp(mode, "sub")) && sub_ok && tickets && pack_sub_eligible(P, B)) {
			const dim3 grid(B.nframes * P.channels);
			uint32_t *subt = (uint32_t *)(tickets + B.nframes);   // two words per (frame, channel), cleared by k_decide like the tickets
#define SUB_LAUNCH(H, N) \
	hipLaunchKernelGGL((k_pack_sub<H, N>), grid, block, 0, st, P, B, pcm, stats, plans, assign, frame_bytes, errflag, out, out_capacity, offsets, tickets, subt, stamps)
			switch (hist) {
			case 4: SUB_LAUNCH(4, true); break;
			case 12: if (narrow) SUB_LAUNCH(12, true); else SUB_LAUNCH(12, false); break;
			default: if (narrow) SUB_LAUNCH(32, true); else SUB_LAUNCH(32, false); break;
			}
#undef SUB_LAUNCH
			return true;
		}
	}
	if (first >= B.nframes)
		return false;
	const dim3 grid(B.nframes - first);
#define PACK_LAUNCH(H, N) \
	do { \
		if (quad) hipLaunchKernelGGL((k_pack<H, N, true>), grid, block, 0, st, P, B, pcm, stats, plans, assign, slots, frame_bytes, errflag, stamps, first); \
		else hipLaunchKernelGGL((k_pack<H, N, false>), grid, block, 0, st, P, B, pcm, stats, plans, assign, slots, frame_bytes, errflag, stamps, first); \
	} while (0)
	switch (hist) {
	case 4: PACK_LAUNCH(4, true); break;
	case 12: if (narrow) PACK_LAUNCH(12, true); else PACK_LAUNCH(12, false); break;
	default: if (narrow) PACK_LAUNCH(32, true); else PACK_LAUNCH(32, false); break;
	}
#undef PACK_LAUNCH
	return false;
}
