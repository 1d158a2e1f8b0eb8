/*
 * flac_oracle.h - TEST INFRASTRUCTURE ONLY.
 *
 * Scalar CPU restatement of the FLAC 1.2.1 frame encode / frame decode algorithm, written from
 * the behaviour of the reference sources (each function in flac_oracle.c cites the reference
 * file:line it follows).  It exists to check the HIP kernels and to serve as a timed CPU
 * baseline; it is never linked into or called from the product library.
 *
 * Pinning: tests/test_oracle_golden.py compares this code byte-for-byte with streams produced
 * by the real reference libFLAC (built by oracle/Makefile into oracle/_ref/) and stored under
 * tests/golden/.
 */
#ifndef FLAC_ORACLE_H
#define FLAC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FO_MAX_CHANNELS 8
#define FO_MAX_LPC_ORDER 32
#define FO_MAX_APODIZATIONS 32

enum { FO_APOD_TUKEY = 0, FO_APOD_HANN = 1, FO_APOD_RECTANGLE = 2, FO_APOD_HAMMING = 3,
       FO_APOD_BLACKMAN = 4, FO_APOD_WELCH = 5, FO_APOD_TRIANGLE = 6, FO_APOD_BARTLETT = 7 };

enum { FO_SUB_CONSTANT = 0, FO_SUB_VERBATIM = 1, FO_SUB_FIXED = 2, FO_SUB_LPC = 3 };

/* Encoder settings after the reference's init-time resolution (stream_encoder.c:649-759). */
typedef struct {
	uint32_t channels;
	uint32_t bits_per_sample;
	uint32_t sample_rate;
	uint32_t blocksize;
	uint32_t do_mid_side_stereo;
	uint32_t loose_mid_side_stereo;
	uint32_t max_lpc_order;
	uint32_t qlp_coeff_precision;
	uint32_t do_qlp_coeff_prec_search;
	uint32_t do_exhaustive_model_search;
	uint32_t min_residual_partition_order;
	uint32_t max_residual_partition_order;
	uint32_t disable_constant_subframes;
	uint32_t disable_fixed_subframes;
	uint32_t disable_verbatim_subframes;
	uint32_t num_apodizations;
	uint32_t apod_type[FO_MAX_APODIZATIONS];
	float    apod_param[FO_MAX_APODIZATIONS];
} fo_params;

/* What the model search chose for one subframe (for debugging kernel mismatches). */
typedef struct {
	uint32_t type;            /* FO_SUB_* */
	uint32_t wasted_bits;
	uint32_t subframe_bps;
	uint32_t order;
	uint32_t estimate_bits;
	uint32_t qlp_precision;
	int32_t  qlp_shift;
	int32_t  qlp_coeff[FO_MAX_LPC_ORDER];
	uint32_t partition_order;
	uint32_t is_rice2;
	uint32_t rice_param0;     /* parameter of partition 0 */
} fo_subframe_info;

typedef struct {
	uint32_t channel_assignment;               /* 0 indep, 1 left/side, 2 right/side, 3 mid/side */
	uint32_t frame_bytes;
	fo_subframe_info sub[FO_MAX_CHANNELS];      /* in written order */
} fo_frame_info;

/* Fill *p the way FLAC__stream_encoder_set_compression_level + init do for a preset.
 * blocksize 0 means "default for the preset".  Returns 0 on success, <0 for invalid settings. */
int fo_params_preset(fo_params *p, uint32_t level, uint32_t channels, uint32_t bits_per_sample,
                     uint32_t sample_rate, uint32_t blocksize);
/* Apply the init-time resolution rules to settings that were filled by hand. */
int fo_params_resolve(fo_params *p);

/* Upper bound on the size of one encoded frame, in bytes. */
size_t fo_max_frame_bytes(const fo_params *p);

/* Encode interleaved int32 PCM (n wide samples) to FLAC frames, blocked like the reference:
 * full blocks of p->blocksize, then a final short block.  Frames are written back to back into
 * out; frame_bytes[i] receives each frame's size; info (may be NULL) receives the decisions.
 * first_frame_number lets a shard start mid-stream.  Returns total bytes, or (size_t)-1. */
size_t fo_encode_frames(const fo_params *p, const int32_t *pcm_interleaved, uint64_t wide_samples,
                        uint32_t first_frame_number, uint8_t *out, size_t out_cap,
                        uint32_t *frame_bytes, uint32_t frame_cap, uint32_t *num_frames,
                        fo_frame_info *info);

/* "fLaC" + STREAMINFO + empty VORBIS_COMMENT, exactly what the reference's init_stream emits
 * when no metadata is supplied, with the finish()-time STREAMINFO fields filled in.
 * Returns the number of bytes written (42 + 8 + vendor length). */
size_t fo_write_stream_header(const fo_params *p, uint64_t total_samples, uint32_t min_framesize,
                              uint32_t max_framesize, const uint8_t md5[16], uint8_t *out, size_t out_cap);

/* Signature of the PCM as the reference computes it: interleaved, little-endian, (bps+7)/8 bytes
 * per sample, hashed with the reference snapshot's MD5 variant (see flac_oracle.c).
 * fo_md5_pcm_mode(..., 0, ...) gives RFC 1321 MD5 of the same bytes instead. */
void fo_md5_pcm(const int32_t *pcm_interleaved, uint64_t wide_samples, uint32_t channels,
                uint32_t bits_per_sample, uint8_t digest[16]);
void fo_md5_pcm_mode(const int32_t *pcm_interleaved, uint64_t wide_samples, uint32_t channels,
                     uint32_t bits_per_sample, int reference_word_order, uint8_t digest[16]);

/* Decoder side. */
typedef struct {
	uint32_t min_blocksize, max_blocksize, min_framesize, max_framesize;
	uint32_t sample_rate, channels, bits_per_sample;
	uint64_t total_samples;
	uint8_t  md5[16];
	size_t   audio_offset;   /* byte offset of the first frame */
} fo_streaminfo;

/* Parse "fLaC" + metadata blocks; returns 0 on success. */
int fo_parse_stream_header(const uint8_t *data, size_t len, fo_streaminfo *si);

/* Decode back-to-back frames starting at data[0].  Output is interleaved int32, max_wide_samples
 * capacity.  Returns the number of wide samples decoded, or <0:
 *   -1 lost sync / bad header, -2 frame CRC mismatch, -3 unparseable, -4 output too small.
 * frames_decoded (may be NULL) receives the frame count. */
int64_t fo_decode_frames(const uint8_t *data, size_t len, uint32_t channels, uint32_t bits_per_sample,
                         uint32_t sample_rate, int32_t *out_interleaved, uint64_t max_wide_samples,
                         uint32_t *frames_decoded);

/* Building blocks exported for unit tests. */
uint8_t  fo_crc8(const uint8_t *data, size_t len);
uint16_t fo_crc16(const uint8_t *data, size_t len);
void     fo_md5(const uint8_t *data, size_t len, uint8_t digest[16]);
size_t   fo_utf8_encode(uint64_t v, uint8_t *out);
void     fo_window(uint32_t type, float param, uint32_t L, float *window);
/* bit writer KAT helper: writes (value,bits) pairs MSB-first, returns bytes (zero padded). */
size_t   fo_bitpack(const uint64_t *values, const uint32_t *nbits, uint32_t n, uint8_t *out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
