// PCM signature used for STREAMINFO (host side; inherently sequential per stream).
#ifndef FLAC_AMD_MD5_H
#define FLAC_AMD_MD5_H

#include <stddef.h>
#include <stdint.h>

struct FaMd5 {
	uint32_t h[4];
	uint64_t nbytes;
	uint8_t block[64];
	uint32_t fill;
	int reference_word_order;
};

// reference_word_order: see fa_md5_default_mode()
void fa_md5_init(FaMd5 *c, int reference_word_order);
void fa_md5_update(FaMd5 *c, const uint8_t *p, size_t n);
void fa_md5_final(FaMd5 *c, uint8_t digest[16]);
// Feed interleaved int32 samples the way FLAC__MD5Accumulate does (md5.c:271,392): little-endian,
// (bps+7)/8 bytes per sample.
void fa_md5_accumulate_interleaved(FaMd5 *c, const int32_t *samples, size_t count, unsigned bytes_per_sample);
// The same from one run of `wide` samples per channel (planes[c] + offset), interleaved on the fly.
void fa_md5_accumulate_planes(FaMd5 *c, const int32_t *const *planes, unsigned channels, size_t wide, unsigned bytes_per_sample);
// The same from samples held at a container width of 2 bytes (streams of up to 16 bits) or 4.
void fa_md5_accumulate_container(FaMd5 *c, const void *samples, unsigned container_bytes, size_t count, unsigned bytes_per_sample);
// 1 unless the environment variable FLAC_AMD_MD5 is "standard".
int fa_md5_default_mode(void);

#endif
