/*
 * flac_amd_pcm.h - the PCM container stage either side of the frame engine: WAVE / RF64 / Wave64 /
 * AIFF / AIFF-C header parsing and writing on the host, and the conversion between a container's
 * sample bytes and the engine's native samples on the device, so that PCM crosses PCIe at the width
 * it has in the file instead of as 32-bit integers.
 *
 * Plain C types only.  Every function returns 0 on success or a negative FLACAMD_E_* code;
 * flacamd_last_error() gives a thread-local message.  Citations are to the reference tree.
 */
#ifndef FLAC_AMD_PCM_H
#define FLAC_AMD_PCM_H

#include <stddef.h>
#include <stdint.h>

#include "flac_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Containers, as the reference's command line tells them apart (src/flac/main.c:1720-1729). */
enum {
	FLACAMD_PCM_RAW = 0,
	FLACAMD_PCM_WAVE,
	FLACAMD_PCM_RF64,
	FLACAMD_PCM_WAVE64,
	FLACAMD_PCM_AIFF,
	FLACAMD_PCM_AIFF_C
};

/* What a container header says about its sample data; the fields of the reference's
 * EncoderSession.info / fmt.iff after get_sample_info_wave / _aiff (src/flac/encode.c:195,674). */
typedef struct {
	uint32_t container;            /* FLACAMD_PCM_* */
	uint32_t channels;
	uint32_t sample_rate;
	uint32_t bits_per_sample;      /* bits a sample occupies in the file: 8, 16, 24 (or 32, which the
	                                  reference's format_input refuses) */
	uint32_t shift;                /* low-order bits that carry no signal and must be zero; the stream
	                                  is encoded at bits_per_sample - shift bits */
	uint32_t is_big_endian;
	uint32_t is_unsigned;
	uint32_t channel_mask;
	uint32_t bytes_per_wide_sample;
	uint32_t reserved;
	uint64_t data_offset;          /* first byte of sample data, from the start of the file */
	uint64_t data_bytes;           /* bytes of sample data the header announces */
	uint64_t total_wide_samples;   /* data_bytes / bytes_per_wide_sample */
} flacamd_pcm_format;

/* Replaces get_sample_info_wave / get_sample_info_aiff (src/flac/encode.c:195-672, 674-900) plus the
 * container detection of src/flac/main.c:1715-1729, on the first `size` bytes of a file held in
 * memory.  channel_map_none mirrors --channel-map=none.  Fails with FLACAMD_E_INVALID on everything
 * the reference reports as an ERROR, and with FLACAMD_E_OVERFLOW when the header runs past `size`
 * bytes (give it more of the file). */
int flacamd_pcm_parse_header(const uint8_t *file, size_t size, int channel_map_none, flacamd_pcm_format *out);

/* Replaces write_iff_headers (src/flac/decode.c:536-783, without foreign metadata): the header of a
 * WAVE / RF64 / Wave64 / AIFF file that holds `wide_samples` samples of a stream with these
 * parameters.  *bytes gets the header's length; out may be NULL to ask for the length only. */
int flacamd_pcm_write_header(uint32_t container, uint32_t channels, uint32_t bits_per_sample, uint32_t sample_rate,
                             uint32_t channel_mask, uint64_t wide_samples, uint8_t *out, size_t capacity, size_t *bytes);

/* Replaces format_input (src/flac/encode.c:2333-2423) on the device: `wide_samples` interleaved
 * samples in the container's byte layout at src_dev (8-bit signed or unsigned, 16- or 24-bit of
 * either byte order) become signed native samples of dst_sample_bytes (2 or 4) bytes at dst_dev,
 * shifted right by fmt->shift: what flacamd_encode_frames takes.  *flag_dev (a device word the
 * caller zeroed) is set to 1 when a sample has non-zero bits among the `shift` low ones, which the
 * reference reports as an error (encode.c:2415).  Asynchronous on `stream`. */
int flacamd_pcm_unpack(const void *src_dev, const flacamd_pcm_format *fmt, uint64_t wide_samples, void *dst_dev,
                       uint32_t dst_sample_bytes, uint32_t *flag_dev, void *stream);

/* Replaces the sample formatting of the command-line decoder's write callback
 * (src/flac/decode.c:957-1255): interleaved native samples (2 or 4 bytes each, as
 * flacamd_decoder_frames writes them) become the bytes of a WAVE-family file (little-endian,
 * unsigned when 8-bit) or an AIFF file (big-endian, signed); a bit depth that is not a multiple of 8
 * is shifted up to the next one (decode.c:962,1094-1098).  dst_dev needs
 * wide_samples * channels * ((bits_per_sample + 7) / 8) bytes.  Asynchronous on `stream`. */
int flacamd_pcm_pack(const void *src_dev, uint32_t src_sample_bytes, uint32_t channels, uint32_t bits_per_sample,
                     uint32_t container, uint64_t wide_samples, void *dst_dev, void *stream);

/* ---- ReplayGain analysis (host code: the loudness filter is a serial recurrence over the stream) ----
 *
 * The reference's own names and meanings.  Core: include/share/replaygain_analysis.h:33-52
 * (src/share/replaygain_analysis/replaygain_analysis.c); one analysis at a time per process, as there.
 * Gains are in dB relative to ReplayGainReferenceLoudness; GAIN_NOT_ENOUGH_SAMPLES is -24601. */
extern float ReplayGainReferenceLoudness;
int InitGainAnalysis(long samplefreq);                 /* 1 = ok, 0 = unsupported sample rate */
int ResetSampleFrequency(long samplefreq);
int AnalyzeSamples(const float *left_samples, const float *right_samples, size_t num_samples, int num_channels);
float GetTitleGain(void);
float GetAlbumGain(void);
/* The wrappers the command line calls per decoded or encoded block
 * (include/share/grabbag/replaygain.h:44-52, src/share/grabbag/replaygain.c:150-275): integer samples of
 * `bps` bits, one array per channel; the peak is relative to full scale. */
int grabbag__replaygain_is_valid_sample_frequency(unsigned sample_frequency);
int grabbag__replaygain_init(unsigned sample_frequency);
int grabbag__replaygain_analyze(const int32_t *const input[], int is_stereo, unsigned bps, unsigned samples);
void grabbag__replaygain_get_album(float *gain, float *peak);
void grabbag__replaygain_get_title(float *gain, float *peak);

#ifdef __cplusplus
}
#endif

#endif
