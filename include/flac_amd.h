/*
 * flac_amd.h - C ABI of the MI355X FLAC frame engine (device-resident batch interface).
 *
 * This is the thin layer between host C code and the HIP kernels.  The libFLAC-compatible
 * stream API (flac_amd_stream.h) is written on top of it; callers that already hold PCM or FLAC
 * bytes in device memory (the benchmark, multi-GPU sharding) call it directly.
 *
 * Plain C types only.  Every function returns 0 on success or a negative FLACAMD_E_* code;
 * flacamd_last_error() gives a thread-local message.  Citations are to the reference tree.
 */
#ifndef FLAC_AMD_H
#define FLAC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLACAMD_MAX_CHANNELS 8
#define FLACAMD_MAX_LPC_ORDER 32
#define FLACAMD_MAX_APODIZATIONS 32
/* Largest Rice partition order the device path evaluates (the FLAC subset limit,
 * include/FLAC/format.h:119 FLAC__SUBSET_MAX_RICE_PARTITION_ORDER).  It bounds the order a block can
 * reach - the setting capped by how often the blocksize halves evenly (stream_encoder.c:2929-2936) -
 * so a setting of 15 with blocksize 1152 (order 7 at most) is accepted. */
#define FLACAMD_MAX_PARTITION_ORDER 8

enum {
	FLACAMD_OK = 0,
	FLACAMD_E_INVALID = -1,      /* bad argument or settings the reference's init would refuse */
	FLACAMD_E_UNSUPPORTED = -2,  /* legal for the reference, outside this engine's device limits */
	FLACAMD_E_DEVICE = -3,       /* HIP runtime error, or no HIP device */
	FLACAMD_E_NOMEM = -4,
	FLACAMD_E_OVERFLOW = -5,     /* output buffer too small */
	FLACAMD_E_STREAM = -6        /* undecodable input (decoder) */
};

/* Apodization codes; same order as FLAC__ApodizationFunction
 * (src/libFLAC/include/protected/stream_encoder.h:46-62). */
enum {
	FLACAMD_APOD_BARTLETT = 0, FLACAMD_APOD_BARTLETT_HANN, FLACAMD_APOD_BLACKMAN,
	FLACAMD_APOD_BLACKMAN_HARRIS_4TERM_92DB, FLACAMD_APOD_CONNES, FLACAMD_APOD_FLATTOP,
	FLACAMD_APOD_GAUSS, FLACAMD_APOD_HAMMING, FLACAMD_APOD_HANN, FLACAMD_APOD_KAISER_BESSEL,
	FLACAMD_APOD_NUTTALL, FLACAMD_APOD_RECTANGLE, FLACAMD_APOD_TRIANGLE, FLACAMD_APOD_TUKEY,
	FLACAMD_APOD_WELCH
};

/* Encoder settings, the fields of FLAC__StreamEncoderProtected that shape a frame
 * (src/libFLAC/include/protected/stream_encoder.h:78-108), after init-time resolution. */
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
	uint32_t apod_type[FLACAMD_MAX_APODIZATIONS];
	float    apod_param[FLACAMD_MAX_APODIZATIONS];
} flacamd_encode_params;

/* replaces FLAC__stream_encoder_set_compression_level (src/libFLAC/stream_encoder.c:1458) followed
 * by the defaulting rules of init_stream_internal_ (:663-759).  blocksize 0 = preset default. */
int flacamd_params_preset(flacamd_encode_params *p, uint32_t level, uint32_t channels,
                          uint32_t bits_per_sample, uint32_t sample_rate, uint32_t blocksize);
/* replaces the validation / defaulting block of init_stream_internal_ (stream_encoder.c:660-759)
 * for settings filled in by hand.  Returns FLACAMD_E_INVALID where init would fail. */
int flacamd_params_resolve(flacamd_encode_params *p);
/* replaces the window generators FLAC__window_* (src/libFLAC/window.c:49-223): fills window[L]. */
int flacamd_window(uint32_t apod_type, float apod_param, uint32_t L, float *window);

typedef struct flacamd_encoder flacamd_encoder;

/* replaces resize_buffers_ (stream_encoder.c:2245): allocates every device buffer for batches of
 * up to max_frames frames on HIP device `device`, uploads the window tables.  The device-buffer
 * entry points allocate nothing further; the *_host ones allocate their staging buffers on first
 * use. */
int flacamd_encoder_create(const flacamd_encode_params *p, int device, uint32_t max_frames,
                           flacamd_encoder **out);
void flacamd_encoder_destroy(flacamd_encoder *e);
/* bytes of output that max_frames frames can need at worst */
size_t flacamd_encoder_max_output_bytes(const flacamd_encoder *e, uint32_t nframes);
uint32_t flacamd_encoder_max_frames(const flacamd_encoder *e);

/* replaces process_frame_ (stream_encoder.c:2861) for a whole batch of consecutive frames.
 *   pcm_dev        interleaved PCM in device memory, `wide_samples` wide samples, each sample
 *                  sample_bytes wide (2: int16, 3: packed little-endian int24, 4: int32)
 *   wide_samples   full blocks, plus (only when is_last != 0) a final short block
 *   first_frame    frame number of the first frame of this batch (stream_encoder.c:2947)
 *   loose_state    in/out: channel assignment of the last loose-mid/side leader frame carried
 *                  between batches (stream_encoder.c:2954-2961,3159); may be NULL
 *   out_dev        device buffer, frames written back to back
 *   frame_bytes_dev / frame_offsets_dev   device arrays [nframes] / [nframes+1] (offsets may be NULL)
 *   stream         a hipStream_t, or NULL for the default stream
 * The call enqueues work on `stream` and returns; *nframes_out is known on return. */
int flacamd_encode_frames(flacamd_encoder *e, const void *pcm_dev, uint32_t sample_bytes,
                          uint64_t wide_samples, uint32_t first_frame, int is_last,
                          uint32_t *loose_state, void *out_dev, size_t out_capacity,
                          uint32_t *frame_bytes_dev, uint64_t *frame_offsets_dev,
                          uint32_t *nframes_out, void *stream);
/* Same, host buffers in and out (int32 interleaved PCM, as FLAC__stream_encoder_process_interleaved
 * receives it, stream_encoder.c:2020); synchronous.  frame_bytes has room for max_frames. */
int flacamd_encode_frames_host(flacamd_encoder *e, const int32_t *pcm, uint64_t wide_samples,
                               uint32_t first_frame, int is_last, uint32_t *loose_state,
                               uint8_t *out, size_t out_capacity, uint32_t *frame_bytes,
                               uint32_t *nframes_out, size_t *bytes_out);
/* No counterpart in the reference.  flacamd_encode_frames for callers that cannot allocate device
 * memory - code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc: it
 * takes the settings instead of a handle and keeps every intermediate buffer of the kernels
 * (statistics, autocorrelations, candidates, plans, frame slots, window tables) in one static
 * device array of FLACAMD_STATIC_SCRATCH_BYTES per GPU, which the GPU holds until the process ends.
 * It calls no hipMalloc / hipFree.  A batch whose buffers do not fit is refused with
 * FLACAMD_E_UNSUPPORTED (about 1 200 frames of 16-bit stereo at -8), as are more than 131 072
 * window samples (apodizations x blocksize).  Arguments as for flacamd_encode_frames, plus
 *   device       HIP device the buffers of this call live on
 *   status_dev   device word that receives the error bits flacamd_encoder_status would report
 *                (0 = none, bit 0 a frame outgrew its slot, bit 1 out_dev too small); may be NULL
 * Every call on a GPU shares that GPU's array: calls that overlap in time on one GPU (two streams,
 * two threads) race (the library serialises only its own bookkeeping, the caller serialises the
 * calls); calls on different GPUs do not.  Three things in it are not legal while a stream is being
 * captured into a HIP graph, so a caller that captures does them beforehand or avoids them: the first
 * call on a device uploads the CRC tables (hipMemcpyToSymbol) and any call that changes blocksize or
 * apodizations waits for the device and uploads window tables with a synchronous copy - make one
 * warm-up call with the same settings outside the capture; and a call with loose mid/side settings
 * (-1, -4) and loose_state != NULL copies that word back and waits for the stream - pass NULL and
 * cut batches at multiples of the loose mid/side period instead. */
#define FLACAMD_STATIC_SCRATCH_BYTES (32u << 20)
int flacamd_encode_frames_static(const flacamd_encode_params *p, int device, const void *pcm_dev,
                                 uint32_t sample_bytes, uint64_t wide_samples, uint32_t first_frame,
                                 int is_last, uint32_t *loose_state, void *out_dev, size_t out_capacity,
                                 uint32_t *frame_bytes_dev, uint64_t *frame_offsets_dev,
                                 uint32_t *status_dev, uint32_t *nframes_out, void *stream);
/* After an encode has completed: any device-side error flag (0 = none). */
int flacamd_encoder_status(flacamd_encoder *e);
/* Time, in ms, spent in the named stage during the last flacamd_encode_frames call that ran with
 * profiling enabled (flacamd_encoder_set_profiling): stages 0..FLACAMD_ENC_STAGES-1. */
#define FLACAMD_ENC_STAGES 8
int flacamd_encoder_set_profiling(flacamd_encoder *e, int on);
int flacamd_encoder_stage_ms(flacamd_encoder *e, float ms[FLACAMD_ENC_STAGES]);
const char *flacamd_encoder_stage_name(int stage);

/* ---- decoder ---------------------------------------------------------------------------- */
typedef struct flacamd_decoder flacamd_decoder;

/* One stream's description for batch decode: where its frames lie in the input buffer. */
typedef struct {
	uint64_t byte_offset;     /* first frame's offset in flac_dev */
	uint64_t byte_length;     /* bytes of frame data */
	uint64_t sample_offset;   /* where this stream's first wide sample goes in the output */
	uint64_t total_samples;   /* wide samples expected (0 = unknown: decode what is there) */
	uint64_t first_sample;    /* sample number, as coded in the frame headers, that lands at
	                             sample_offset (0 for a stream decoded from its beginning);
	                             FLACAMD_FIRST_SAMPLE_AUTO with a single stream: the number in
	                             the first frame header found */
} flacamd_stream_desc;
#define FLACAMD_FIRST_SAMPLE_AUTO (~(uint64_t)0)

/* channels / bits_per_sample / max_blocksize come from STREAMINFO (stream_decoder.c:1423
 * read_metadata_); all streams of one decoder share them. */
int flacamd_decoder_create(uint32_t channels, uint32_t bits_per_sample, uint32_t max_blocksize,
                           int device, uint64_t max_input_bytes, uint32_t max_frames,
                           flacamd_decoder **out);
void flacamd_decoder_destroy(flacamd_decoder *d);
/* replaces frame_sync_ + read_frame_ (stream_decoder.c:1941,1996) for a batch of streams whose
 * bytes are resident in device memory.  Output: interleaved PCM, out_sample_bytes wide (2 or 4),
 * at each stream's sample_offset.  frames_out / status per frame are kept on the device and
 * summarised by flacamd_decoder_result.  flac_dev must be 4-byte aligned; the buffer is read in
 * 32-bit words, so the word that holds its last byte is read whole (the extra bytes are ignored). */
int flacamd_decode_streams(flacamd_decoder *d, const uint8_t *flac_dev, uint64_t flac_bytes,
                           const flacamd_stream_desc *streams_host, uint32_t nstreams,
                           void *pcm_out_dev, uint32_t out_sample_bytes, uint64_t out_capacity_wide,
                           void *stream);
/* Same, host buffers in and out; synchronous; fills the three counters of flacamd_decoder_result. */
int flacamd_decode_streams_host(flacamd_decoder *d, const uint8_t *flac, uint64_t flac_bytes,
                                const flacamd_stream_desc *streams, uint32_t nstreams, void *pcm_out,
                                uint32_t out_sample_bytes, uint64_t out_capacity_wide,
                                uint32_t *frames, uint64_t *wide_samples, uint32_t *bad_frames);
/* Frame numbers in fixed-blocksize headers count frames; the sample number is that count times the
 * stream's blocksize (stream_decoder.c:2416-2439).  Defaults to max_blocksize; a caller that knows
 * better (STREAMINFO min_blocksize, or the first frame's blocksize) sets it here. */
int flacamd_decoder_set_frame_number_scale(flacamd_decoder *d, uint32_t samples_per_frame_number);
/* One row per frame header found by the last decode, in byte order: what read_frame_header_ and
 * read_frame_ (stream_decoder.c:2141,1996) learn about a frame.  status: 1 good, 2 malformed,
 * 3 CRC-16 mismatch, 4 cut off by the end of the input. */
typedef struct {
	uint64_t pos;        /* byte offset of the sync code in the input */
	uint64_t sample;     /* number of the frame's first sample */
	uint64_t end_byte;   /* first byte after the frame's CRC-16 */
	uint32_t hdr_bytes, blocksize;
	uint32_t assignment; /* FLAC__ChannelAssignment */
	uint32_t stream;
	uint32_t status;
	uint32_t reserved;   /* bit 0, with status 2: the parse stopped at a reserved subframe type or residual coding
	                        method (the reference's UNPARSEABLE_STREAM); bit 1 (internal): the frame decoder wrote the
	                        frame's samples to the output itself */
} flacamd_frame_info;
/* on: from the next decode on a frame's samples leave channel by channel - the frame's stretch of the output (blocksize x
 * channels samples, where its interleaved samples would lie) holds channel 0's blocksize samples, then channel 1's, and so
 * on - which is what the reference's write callback hands out (`const FLAC__int32 *const buffer[]`, stream_decoder.c:2913).
 * Default: interleaved. */
int flacamd_decoder_set_planar_output(flacamd_decoder *d, int on);
/* What read_subframe_ (stream_decoder.c:2450-2713) learns about one subframe besides its samples: what the
 * reference's write callback finds in FLAC__Frame.subframes[] (consumed by `flac --analyze`, src/flac/analyze.c:63-144),
 * except for the residual itself, which stays on the device. */
#define FLACAMD_SUBFRAME_PARAMS 256
typedef struct {
	uint8_t type;        /* 0 constant, 1 verbatim, 2 fixed, 3 LPC */
	uint8_t order;
	uint8_t wasted_bits;
	uint8_t precision;   /* LPC: bits per coefficient */
	int8_t shift;        /* LPC: quantization level */
	uint8_t partition_order;
	uint8_t rice2;       /* 1: five-bit Rice parameters (FLAC__ENTROPY_CODING_METHOD_PARTITIONED_RICE2) */
	uint8_t pad;
	int32_t constant;    /* type 0: the value (before the wasted bits are shifted back in) */
	uint32_t pad2[2];
	int32_t qlp_coeff[32];
	int32_t warmup[32];
	/* per partition, the first FLACAMD_SUBFRAME_PARAMS of them: the Rice parameter, or 0x80 | raw bits of an escaped partition */
	uint8_t params[FLACAMD_SUBFRAME_PARAMS];
} flacamd_subframe_info;
/* Off by default.  on: the decoder keeps one record per (frame, channel) on the device, filled by every decode from then on. */
int flacamd_decoder_enable_subframe_info(flacamd_decoder *d, int on);
/* Blocks until the decode finished, then copies the records of frames [first_frame, first_frame + nframes) of the table
 * of flacamd_decoder_frames: `channels` consecutive records per frame.  Records of frames whose status is not 1 or 3 are
 * undefined. */
int flacamd_decoder_subframe_info(flacamd_decoder *d, uint32_t first_frame, uint32_t nframes, flacamd_subframe_info *out);
/* Blocks until the decode finished, then copies up to `capacity` rows.  *truncated is set when
 * the input held more frame headers than the decoder's max_frames or than `capacity`. */
int flacamd_decoder_frames(flacamd_decoder *d, flacamd_frame_info *out, uint32_t capacity,
                           uint32_t *count, int *truncated);
/* Blocks until the decode finished; reports frames decoded, wide samples written and the number
 * of frames whose CRC-16 or structure was bad (those are zero-filled, stream_decoder.c:2105-2113).
 * A sync pattern that happens to carry a valid header CRC-8 inside the bytes of a good frame stays
 * in the table of flacamd_decoder_frames (status 2 or 3) but is not a frame and is not counted. */
int flacamd_decoder_result(flacamd_decoder *d, uint32_t *frames, uint64_t *wide_samples,
                           uint32_t *bad_frames);
#define FLACAMD_DEC_STAGES 4
int flacamd_decoder_set_profiling(flacamd_decoder *d, int on);
int flacamd_decoder_stage_ms(flacamd_decoder *d, float ms[FLACAMD_DEC_STAGES]);
const char *flacamd_decoder_stage_name(int stage);

const char *flacamd_last_error(void);
const char *flacamd_version(void);

#ifdef __cplusplus
}
#endif
#endif
