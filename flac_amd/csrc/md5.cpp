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

// One 64-byte block (RFC 1321 section 3.4), the 64 steps written out so that the message word and
// the rotation of each step are constants.
template <bool BE>
static void transform_block(uint32_t h[4], const uint8_t *p)
{
	uint32_t w[16];
	for (int i = 0; i < 16; i++) {
		uint32_t v;
		memcpy(&v, p + 4 * i, 4);  // little-endian host
		w[i] = BE ? __builtin_bswap32(v) : v;
	}
	uint32_t a = h[0], b = h[1], c = h[2], d = h[3];
#define F1(x, y, z) (z ^ (x & (y ^ z)))
#define F2(x, y, z) (y ^ (z & (x ^ y)))
#define F3(x, y, z) (x ^ y ^ z)
#define F4(x, y, z) (y ^ (x | ~z))
#define STEP(f, a, b, c, d, g, k, r) a = b + rol(a + f(b, c, d) + (k) + w[g], r)
	STEP(F1, a, b, c, d, 0, 0xd76aa478, 7);  STEP(F1, d, a, b, c, 1, 0xe8c7b756, 12); STEP(F1, c, d, a, b, 2, 0x242070db, 17); STEP(F1, b, c, d, a, 3, 0xc1bdceee, 22);
	STEP(F1, a, b, c, d, 4, 0xf57c0faf, 7);  STEP(F1, d, a, b, c, 5, 0x4787c62a, 12); STEP(F1, c, d, a, b, 6, 0xa8304613, 17); STEP(F1, b, c, d, a, 7, 0xfd469501, 22);
	STEP(F1, a, b, c, d, 8, 0x698098d8, 7);  STEP(F1, d, a, b, c, 9, 0x8b44f7af, 12); STEP(F1, c, d, a, b, 10, 0xffff5bb1, 17); STEP(F1, b, c, d, a, 11, 0x895cd7be, 22);
	STEP(F1, a, b, c, d, 12, 0x6b901122, 7); STEP(F1, d, a, b, c, 13, 0xfd987193, 12); STEP(F1, c, d, a, b, 14, 0xa679438e, 17); STEP(F1, b, c, d, a, 15, 0x49b40821, 22);
	STEP(F2, a, b, c, d, 1, 0xf61e2562, 5);  STEP(F2, d, a, b, c, 6, 0xc040b340, 9);  STEP(F2, c, d, a, b, 11, 0x265e5a51, 14); STEP(F2, b, c, d, a, 0, 0xe9b6c7aa, 20);
	STEP(F2, a, b, c, d, 5, 0xd62f105d, 5);  STEP(F2, d, a, b, c, 10, 0x02441453, 9); STEP(F2, c, d, a, b, 15, 0xd8a1e681, 14); STEP(F2, b, c, d, a, 4, 0xe7d3fbc8, 20);
	STEP(F2, a, b, c, d, 9, 0x21e1cde6, 5);  STEP(F2, d, a, b, c, 14, 0xc33707d6, 9); STEP(F2, c, d, a, b, 3, 0xf4d50d87, 14);  STEP(F2, b, c, d, a, 8, 0x455a14ed, 20);
	STEP(F2, a, b, c, d, 13, 0xa9e3e905, 5); STEP(F2, d, a, b, c, 2, 0xfcefa3f8, 9);  STEP(F2, c, d, a, b, 7, 0x676f02d9, 14);  STEP(F2, b, c, d, a, 12, 0x8d2a4c8a, 20);
	STEP(F3, a, b, c, d, 5, 0xfffa3942, 4);  STEP(F3, d, a, b, c, 8, 0x8771f681, 11); STEP(F3, c, d, a, b, 11, 0x6d9d6122, 16); STEP(F3, b, c, d, a, 14, 0xfde5380c, 23);
	STEP(F3, a, b, c, d, 1, 0xa4beea44, 4);  STEP(F3, d, a, b, c, 4, 0x4bdecfa9, 11); STEP(F3, c, d, a, b, 7, 0xf6bb4b60, 16);  STEP(F3, b, c, d, a, 10, 0xbebfbc70, 23);
	STEP(F3, a, b, c, d, 13, 0x289b7ec6, 4); STEP(F3, d, a, b, c, 0, 0xeaa127fa, 11); STEP(F3, c, d, a, b, 3, 0xd4ef3085, 16);  STEP(F3, b, c, d, a, 6, 0x04881d05, 23);
	STEP(F3, a, b, c, d, 9, 0xd9d4d039, 4);  STEP(F3, d, a, b, c, 12, 0xe6db99e5, 11); STEP(F3, c, d, a, b, 15, 0x1fa27cf8, 16); STEP(F3, b, c, d, a, 2, 0xc4ac5665, 23);
	STEP(F4, a, b, c, d, 0, 0xf4292244, 6);  STEP(F4, d, a, b, c, 7, 0x432aff97, 10); STEP(F4, c, d, a, b, 14, 0xab9423a7, 15); STEP(F4, b, c, d, a, 5, 0xfc93a039, 21);
	STEP(F4, a, b, c, d, 12, 0x655b59c3, 6); STEP(F4, d, a, b, c, 3, 0x8f0ccc92, 10); STEP(F4, c, d, a, b, 10, 0xffeff47d, 15); STEP(F4, b, c, d, a, 1, 0x85845dd1, 21);
	STEP(F4, a, b, c, d, 8, 0x6fa87e4f, 6);  STEP(F4, d, a, b, c, 15, 0xfe2ce6e0, 10); STEP(F4, c, d, a, b, 6, 0xa3014314, 15); STEP(F4, b, c, d, a, 13, 0x4e0811a1, 21);
	STEP(F4, a, b, c, d, 4, 0xf7537e82, 6);  STEP(F4, d, a, b, c, 11, 0xbd3af235, 10); STEP(F4, c, d, a, b, 2, 0x2ad7d2bb, 15); STEP(F4, b, c, d, a, 9, 0xeb86d391, 21);
#undef STEP
#undef F1
#undef F2
#undef F3
#undef F4
	h[0] += a; h[1] += b; h[2] += c; h[3] += d;
}

static void transform(uint32_t h[4], const uint8_t *p, int be)
{
	if (be)
		transform_block<true>(h, p);
	else
		transform_block<false>(h, p);
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
	uint8_t buf[4096 + 8];
	size_t fill = 0;
	for (size_t i = 0; i < count;) {
		// a buffer-full at a time, each sample's low bytes little-endian
		size_t n = (4096 - fill) / bps_bytes;
		if (n > count - i)
			n = count - i;
		uint8_t *o = buf + fill;
		if (bps_bytes == 2) {
			for (size_t k = 0; k < n; k++) {
				const uint16_t v = (uint16_t)samples[i + k];
				memcpy(o + 2 * k, &v, 2);
			}
		} else if (bps_bytes == 4)
			memcpy(o, samples + i, 4 * n);
		else if (bps_bytes == 3) {
			for (size_t k = 0; k < n; k++) {
				const uint32_t v = (uint32_t)samples[i + k];
				memcpy(o + 3 * k, &v, 4);  // the fourth byte is overwritten by the next sample (slack at the end)
			}
		} else {
			for (size_t k = 0; k < n; k++)
				o[k] = (uint8_t)samples[i + k];
		}
		fill += n * bps_bytes;
		i += n;
		if (fill + bps_bytes > 4096) {
			fa_md5_update(c, buf, fill);
			fill = 0;
		}
	}
	if (fill)
		fa_md5_update(c, buf, fill);
}

void fa_md5_accumulate_planes(FaMd5 *c, const int32_t *const *planes, unsigned channels, size_t wide, unsigned bps_bytes)
{
	if (channels == 1) {
		fa_md5_accumulate_interleaved(c, planes[0], wide, bps_bytes);
		return;
	}
	uint8_t buf[4096 + 8];
	const size_t per = 4096 / (bps_bytes * channels);   // wide samples per buffer-full
	for (size_t i = 0; i < wide;) {
		const size_t n = wide - i < per ? wide - i : per;
		for (unsigned ch = 0; ch < channels; ch++) {
			const int32_t *s = planes[ch] + i;
			uint8_t *o = buf + ch * bps_bytes;
			const size_t step = (size_t)bps_bytes * channels;
			if (bps_bytes == 2) {
				for (size_t k = 0; k < n; k++) {
					const uint16_t v = (uint16_t)s[k];
					memcpy(o + k * step, &v, 2);
				}
			} else if (bps_bytes == 4) {
				for (size_t k = 0; k < n; k++)
					memcpy(o + k * step, s + k, 4);
			} else if (bps_bytes == 3) {
				for (size_t k = 0; k < n; k++) {
					const uint32_t v = (uint32_t)s[k];
					o[k * step] = (uint8_t)v;
					o[k * step + 1] = (uint8_t)(v >> 8);
					o[k * step + 2] = (uint8_t)(v >> 16);
				}
			} else {
				for (size_t k = 0; k < n; k++)
					o[k * step] = (uint8_t)s[k];
			}
		}
		fa_md5_update(c, buf, n * bps_bytes * channels);
		i += n;
	}
}

void fa_md5_accumulate_container(FaMd5 *c, const void *samples, unsigned container_bytes, size_t count, unsigned bps_bytes)
{
	if (container_bytes == 4) {
		fa_md5_accumulate_interleaved(c, (const int32_t *)samples, count, bps_bytes);
		return;
	}
	const int16_t *s16 = (const int16_t *)samples;
	if (bps_bytes == 2) {
		// little-endian 16-bit samples are the digest's input as they stand
		fa_md5_update(c, (const uint8_t *)s16, count * 2);
		return;
	}
	uint8_t buf[4096];
	for (size_t i = 0; i < count;) {
		size_t n = count - i < sizeof buf ? count - i : sizeof buf;
		for (size_t k = 0; k < n; k++)
			buf[k] = (uint8_t)s16[i + k];
		fa_md5_update(c, buf, n);
		i += n;
	}
}

int fa_md5_default_mode(void)
{
	const char *e = getenv("FLAC_AMD_MD5");
	return (e && strcmp(e, "standard") == 0) ? 0 : 1;
}
