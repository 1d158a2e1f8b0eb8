// MD5 of the unencoded audio, for STREAMINFO (FLAC__MD5Init/Accumulate/Final, src/libFLAC/md5.c).
//
// Two digests exist for the same bytes.  RFC 1321 reads each 32-bit message word little-endian.
// The reference snapshot this library stands in for passes every message word through ntohl()
// on little-endian hosts (md5.c:24-32 with :181-183,191-193,239-249) while storing the two
// bit-length words as plain values (md5.c:252-253), so its digest differs; its encoder writes that
// digest and its decoder and `flac -t` check against it.  reference_word_order = 1 reproduces the
// snapshot (the default, so that streams are interchangeable with it); 0 is RFC 1321.
#include "md5.h"

#include <cstdlib>
#include <cstring>

static inline uint32_t rol(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }

static void transform(uint32_t h[4], const uint8_t *p, int be)
{
	static const uint32_t K[64] = {
		0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501,
		0x698098d8, 0x8b44f7af, 0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821,
		0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8,
		0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a,
		0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70,
		0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665,
		0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1,
		0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
	static const int S[4][4] = {{7, 12, 17, 22}, {5, 9, 14, 20}, {4, 11, 16, 23}, {6, 10, 15, 21}};
	uint32_t w[16];
	for (int i = 0; i < 16; i++) {
		uint32_t v;
		memcpy(&v, p + 4 * i, 4);  // little-endian host
		w[i] = be ? __builtin_bswap32(v) : v;
	}
	uint32_t a = h[0], b = h[1], c = h[2], d = h[3];
	for (int i = 0; i < 64; i++) {
		uint32_t f;
		int g;
		const int r = i >> 4;
		if (r == 0) { f = d ^ (b & (c ^ d)); g = i; }
		else if (r == 1) { f = c ^ (d & (b ^ c)); g = (5 * i + 1) & 15; }
		else if (r == 2) { f = b ^ c ^ d; g = (3 * i + 5) & 15; }
		else { f = c ^ (b | ~d); g = (7 * i) & 15; }
		const uint32_t t = d;
		d = c;
		c = b;
		b = b + rol(a + f + K[i] + w[g], S[r][i & 3]);
		a = t;
	}
	h[0] += a; h[1] += b; h[2] += c; h[3] += d;
}

void fa_md5_init(FaMd5 *c, int reference_word_order)
{
	c->h[0] = 0x67452301; c->h[1] = 0xefcdab89; c->h[2] = 0x98badcfe; c->h[3] = 0x10325476;
	c->nbytes = 0;
	c->fill = 0;
	c->reference_word_order = reference_word_order;
}

void fa_md5_update(FaMd5 *c, const uint8_t *p, size_t n)
{
	c->nbytes += n;
	if (c->fill) {
		size_t take = 64 - c->fill;
		if (take > n) take = n;
		memcpy(c->block + c->fill, p, take);
		c->fill += (uint32_t)take;
		p += take;
		n -= take;
		if (c->fill < 64)
			return;
		transform(c->h, c->block, c->reference_word_order);
		c->fill = 0;
	}
	while (n >= 64) {
		transform(c->h, p, c->reference_word_order);
		p += 64;
		n -= 64;
	}
	if (n) {
		memcpy(c->block, p, n);
		c->fill = (uint32_t)n;
	}
}

void fa_md5_final(FaMd5 *c, uint8_t digest[16])
{
	const uint64_t bits = c->nbytes * 8;
	uint8_t tail[72];
	size_t n = 0;
	tail[n++] = 0x80;
	while ((c->fill + n) % 64 != 56)
		tail[n++] = 0;
	uint8_t len[8];
	for (int i = 0; i < 8; i++)
		len[i] = (uint8_t)(bits >> (8 * i));
	if (c->reference_word_order) {
		// the length words are stored as values, after the message words were swapped
		for (int w = 0; w < 2; w++) {
			uint8_t t = len[4 * w]; len[4 * w] = len[4 * w + 3]; len[4 * w + 3] = t;
			t = len[4 * w + 1]; len[4 * w + 1] = len[4 * w + 2]; len[4 * w + 2] = t;
		}
	}
	memcpy(tail + n, len, 8);
	n += 8;
	fa_md5_update(c, tail, n);
	for (int i = 0; i < 4; i++)
		for (int j = 0; j < 4; j++)
			digest[4 * i + j] = (uint8_t)(c->h[i] >> (8 * j));
}

void fa_md5_accumulate_interleaved(FaMd5 *c, const int32_t *samples, size_t count, unsigned bps_bytes)
{
	uint8_t buf[4096];
	size_t fill = 0;
	for (size_t i = 0; i < count; i++) {
		const uint32_t v = (uint32_t)samples[i];
		for (unsigned b = 0; b < bps_bytes; b++)
			buf[fill++] = (uint8_t)(v >> (8 * b));
		if (fill + 4 > sizeof buf) {
			fa_md5_update(c, buf, fill);
			fill = 0;
		}
	}
	if (fill)
		fa_md5_update(c, buf, fill);
}

int fa_md5_default_mode(void)
{
	const char *e = getenv("FLAC_AMD_MD5");
	return (e && strcmp(e, "standard") == 0) ? 0 : 1;
}
