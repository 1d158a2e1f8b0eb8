/*
 * flac_amd_stream.h - the libFLAC 1.2.1 stream encoder / decoder C API as exported by
 * libflac_amd.so.
 *
 * A client compiled against the reference's own headers (include/FLAC/stream_encoder.h,
 * stream_decoder.h, format.h) links against these symbols unchanged: names, argument meaning,
 * return conventions, enum values and structure layouts are the reference's.  This header
 * restates only what the exported functions need, for clients that do not have the reference's
 * headers; each group cites the declaration it mirrors (paths relative to the reference tree).
 *
 * Frames are produced by the GPU in batches, so write callbacks arrive in bursts (inside
 * process*() when a batch fills, and inside finish()); order, arguments and bytes are the
 * reference's.  The reference makes no promise about callback timing (stream_encoder.h:523-533).
 */
#ifndef FLAC_AMD_STREAM_H
#define FLAC_AMD_STREAM_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* include/FLAC/ordinals.h */
typedef int FLAC__bool;
typedef uint8_t FLAC__byte;
typedef uint8_t FLAC__uint8;
typedef uint16_t FLAC__uint16;
typedef int32_t FLAC__int32;
typedef uint32_t FLAC__uint32;
typedef int64_t FLAC__int64;
typedef uint64_t FLAC__uint64;

/* include/FLAC/format.h:92-153 */
#define FLAC__MAX_METADATA_TYPE_CODE 126u
#define FLAC__MIN_BLOCK_SIZE 16u
#define FLAC__MAX_BLOCK_SIZE 65535u
#define FLAC__MAX_CHANNELS 8u
#define FLAC__MIN_BITS_PER_SAMPLE 4u
#define FLAC__MAX_BITS_PER_SAMPLE 32u
#define FLAC__REFERENCE_CODEC_MAX_BITS_PER_SAMPLE 24u
#define FLAC__MAX_SAMPLE_RATE 655350u
#define FLAC__MAX_LPC_ORDER 32u
#define FLAC__MIN_QLP_COEFF_PRECISION 5u
#define FLAC__MAX_QLP_COEFF_PRECISION 15u
#define FLAC__MAX_FIXED_ORDER 4u
#define FLAC__MAX_RICE_PARTITION_ORDER 15u
#define FLAC__SUBSET_MAX_RICE_PARTITION_ORDER 8u

/* ---- frame description handed to the decoder's write callback (include/FLAC/format.h:193-476) ---- */
typedef enum { FLAC__ENTROPY_CODING_METHOD_PARTITIONED_RICE = 0, FLAC__ENTROPY_CODING_METHOD_PARTITIONED_RICE2 = 1 } FLAC__EntropyCodingMethodType;
typedef struct { unsigned *parameters; unsigned *raw_bits; unsigned capacity_by_order; } FLAC__EntropyCodingMethod_PartitionedRiceContents;
typedef struct { unsigned order; const FLAC__EntropyCodingMethod_PartitionedRiceContents *contents; } FLAC__EntropyCodingMethod_PartitionedRice;
typedef struct { FLAC__EntropyCodingMethodType type; union { FLAC__EntropyCodingMethod_PartitionedRice partitioned_rice; } data; } FLAC__EntropyCodingMethod;
typedef enum { FLAC__SUBFRAME_TYPE_CONSTANT = 0, FLAC__SUBFRAME_TYPE_VERBATIM = 1, FLAC__SUBFRAME_TYPE_FIXED = 2, FLAC__SUBFRAME_TYPE_LPC = 3 } FLAC__SubframeType;
typedef struct { FLAC__int32 value; } FLAC__Subframe_Constant;
typedef struct { const FLAC__int32 *data; } FLAC__Subframe_Verbatim;
typedef struct { FLAC__EntropyCodingMethod entropy_coding_method; unsigned order; FLAC__int32 warmup[FLAC__MAX_FIXED_ORDER]; const FLAC__int32 *residual; } FLAC__Subframe_Fixed;
typedef struct {
	FLAC__EntropyCodingMethod entropy_coding_method;
	unsigned order, qlp_coeff_precision;
	int quantization_level;
	FLAC__int32 qlp_coeff[FLAC__MAX_LPC_ORDER];
	FLAC__int32 warmup[FLAC__MAX_LPC_ORDER];
	const FLAC__int32 *residual;
} FLAC__Subframe_LPC;
typedef struct {
	FLAC__SubframeType type;
	union { FLAC__Subframe_Constant constant; FLAC__Subframe_Fixed fixed; FLAC__Subframe_LPC lpc; FLAC__Subframe_Verbatim verbatim; } data;
	unsigned wasted_bits;
} FLAC__Subframe;
typedef enum { FLAC__CHANNEL_ASSIGNMENT_INDEPENDENT = 0, FLAC__CHANNEL_ASSIGNMENT_LEFT_SIDE = 1, FLAC__CHANNEL_ASSIGNMENT_RIGHT_SIDE = 2, FLAC__CHANNEL_ASSIGNMENT_MID_SIDE = 3 } FLAC__ChannelAssignment;
typedef enum { FLAC__FRAME_NUMBER_TYPE_FRAME_NUMBER, FLAC__FRAME_NUMBER_TYPE_SAMPLE_NUMBER } FLAC__FrameNumberType;
typedef struct {
	unsigned blocksize, sample_rate, channels;
	FLAC__ChannelAssignment channel_assignment;
	unsigned bits_per_sample;
	FLAC__FrameNumberType number_type;
	union { FLAC__uint32 frame_number; FLAC__uint64 sample_number; } number;
	FLAC__uint8 crc;
} FLAC__FrameHeader;
typedef struct { FLAC__uint16 crc; } FLAC__FrameFooter;
typedef struct { FLAC__FrameHeader header; FLAC__Subframe subframes[FLAC__MAX_CHANNELS]; FLAC__FrameFooter footer; } FLAC__Frame;

/* ---- metadata blocks (include/FLAC/format.h:478-846) ---- */
typedef enum {
	FLAC__METADATA_TYPE_STREAMINFO = 0, FLAC__METADATA_TYPE_PADDING = 1, FLAC__METADATA_TYPE_APPLICATION = 2,
	FLAC__METADATA_TYPE_SEEKTABLE = 3, FLAC__METADATA_TYPE_VORBIS_COMMENT = 4, FLAC__METADATA_TYPE_CUESHEET = 5,
	FLAC__METADATA_TYPE_PICTURE = 6, FLAC__METADATA_TYPE_UNDEFINED = 7
} FLAC__MetadataType;
typedef struct {
	unsigned min_blocksize, max_blocksize, min_framesize, max_framesize, sample_rate, channels, bits_per_sample;
	FLAC__uint64 total_samples;
	FLAC__byte md5sum[16];
} FLAC__StreamMetadata_StreamInfo;
#define FLAC__STREAM_METADATA_STREAMINFO_LENGTH (34u)
typedef struct { int dummy; } FLAC__StreamMetadata_Padding;
typedef struct { FLAC__byte id[4]; FLAC__byte *data; } FLAC__StreamMetadata_Application;
typedef struct { FLAC__uint64 sample_number, stream_offset; unsigned frame_samples; } FLAC__StreamMetadata_SeekPoint;
#define FLAC__STREAM_METADATA_SEEKPOINT_LENGTH (18u)
typedef struct { unsigned num_points; FLAC__StreamMetadata_SeekPoint *points; } FLAC__StreamMetadata_SeekTable;
typedef struct { FLAC__uint32 length; FLAC__byte *entry; } FLAC__StreamMetadata_VorbisComment_Entry;
typedef struct { FLAC__StreamMetadata_VorbisComment_Entry vendor_string; FLAC__uint32 num_comments; FLAC__StreamMetadata_VorbisComment_Entry *comments; } FLAC__StreamMetadata_VorbisComment;
typedef struct { FLAC__uint64 offset; FLAC__byte number; } FLAC__StreamMetadata_CueSheet_Index;
typedef struct {
	FLAC__uint64 offset;
	FLAC__byte number;
	char isrc[13];
	unsigned type : 1;
	unsigned pre_emphasis : 1;
	FLAC__byte num_indices;
	FLAC__StreamMetadata_CueSheet_Index *indices;
} FLAC__StreamMetadata_CueSheet_Track;
typedef struct { char media_catalog_number[129]; FLAC__uint64 lead_in; FLAC__bool is_cd; unsigned num_tracks; FLAC__StreamMetadata_CueSheet_Track *tracks; } FLAC__StreamMetadata_CueSheet;
typedef int FLAC__StreamMetadata_Picture_Type;
typedef struct {
	FLAC__StreamMetadata_Picture_Type type;
	char *mime_type;
	FLAC__byte *description;
	FLAC__uint32 width, height, depth, colors, data_length;
	FLAC__byte *data;
} FLAC__StreamMetadata_Picture;
typedef struct { FLAC__byte *data; } FLAC__StreamMetadata_Unknown;
typedef struct {
	FLAC__MetadataType type;
	FLAC__bool is_last;
	unsigned length;
	union {
		FLAC__StreamMetadata_StreamInfo stream_info;
		FLAC__StreamMetadata_Padding padding;
		FLAC__StreamMetadata_Application application;
		FLAC__StreamMetadata_SeekTable seek_table;
		FLAC__StreamMetadata_VorbisComment vorbis_comment;
		FLAC__StreamMetadata_CueSheet cue_sheet;
		FLAC__StreamMetadata_Picture picture;
		FLAC__StreamMetadata_Unknown unknown;
	} data;
} FLAC__StreamMetadata;

/* exported data (src/libFLAC/format.c:56-62,66-120) */
extern const char *FLAC__VERSION_STRING;
extern const char *FLAC__VENDOR_STRING;
extern const FLAC__byte FLAC__STREAM_SYNC_STRING[4];
extern const unsigned FLAC__STREAM_SYNC;
extern const unsigned FLAC__STREAM_SYNC_LEN;
extern const FLAC__uint64 FLAC__STREAM_METADATA_SEEKPOINT_PLACEHOLDER;

/* bit-field widths, masks and name tables (src/libFLAC/format.c:66-211) */
extern const unsigned FLAC__STREAM_METADATA_STREAMINFO_MIN_BLOCK_SIZE_LEN;
extern const unsigned FLAC__STREAM_METADATA_STREAMINFO_MAX_BLOCK_SIZE_LEN;
extern const unsigned FLAC__STREAM_METADATA_STREAMINFO_MIN_FRAME_SIZE_LEN;
extern const unsigned FLAC__STREAM_METADATA_STREAMINFO_MAX_FRAME_SIZE_LEN;
extern const unsigned FLAC__STREAM_METADATA_STREAMINFO_SAMPLE_RATE_LEN;
extern const unsigned FLAC__STREAM_METADATA_STREAMINFO_CHANNELS_LEN;
extern const unsigned FLAC__STREAM_METADATA_STREAMINFO_BITS_PER_SAMPLE_LEN;
extern const unsigned FLAC__STREAM_METADATA_STREAMINFO_TOTAL_SAMPLES_LEN;
extern const unsigned FLAC__STREAM_METADATA_STREAMINFO_MD5SUM_LEN;
extern const unsigned FLAC__STREAM_METADATA_APPLICATION_ID_LEN;
extern const unsigned FLAC__STREAM_METADATA_SEEKPOINT_SAMPLE_NUMBER_LEN;
extern const unsigned FLAC__STREAM_METADATA_SEEKPOINT_STREAM_OFFSET_LEN;
extern const unsigned FLAC__STREAM_METADATA_SEEKPOINT_FRAME_SAMPLES_LEN;
extern const unsigned FLAC__STREAM_METADATA_VORBIS_COMMENT_ENTRY_LENGTH_LEN;
extern const unsigned FLAC__STREAM_METADATA_VORBIS_COMMENT_NUM_COMMENTS_LEN;
extern const unsigned FLAC__STREAM_METADATA_IS_LAST_LEN;
extern const unsigned FLAC__STREAM_METADATA_TYPE_LEN;
extern const unsigned FLAC__STREAM_METADATA_LENGTH_LEN;
extern const unsigned FLAC__FRAME_HEADER_SYNC;
extern const unsigned FLAC__FRAME_HEADER_SYNC_LEN;
extern const unsigned FLAC__FRAME_HEADER_RESERVED_LEN;
extern const unsigned FLAC__FRAME_HEADER_BLOCKING_STRATEGY_LEN;
extern const unsigned FLAC__FRAME_HEADER_BLOCK_SIZE_LEN;
extern const unsigned FLAC__FRAME_HEADER_SAMPLE_RATE_LEN;
extern const unsigned FLAC__FRAME_HEADER_CHANNEL_ASSIGNMENT_LEN;
extern const unsigned FLAC__FRAME_HEADER_BITS_PER_SAMPLE_LEN;
extern const unsigned FLAC__FRAME_HEADER_ZERO_PAD_LEN;
extern const unsigned FLAC__FRAME_HEADER_CRC_LEN;
extern const unsigned FLAC__FRAME_FOOTER_CRC_LEN;
extern const unsigned FLAC__ENTROPY_CODING_METHOD_TYPE_LEN;
extern const unsigned FLAC__ENTROPY_CODING_METHOD_PARTITIONED_RICE_ORDER_LEN;
extern const unsigned FLAC__ENTROPY_CODING_METHOD_PARTITIONED_RICE_PARAMETER_LEN;
extern const unsigned FLAC__ENTROPY_CODING_METHOD_PARTITIONED_RICE2_PARAMETER_LEN;
extern const unsigned FLAC__ENTROPY_CODING_METHOD_PARTITIONED_RICE_RAW_LEN;
extern const unsigned FLAC__ENTROPY_CODING_METHOD_PARTITIONED_RICE_ESCAPE_PARAMETER;
extern const unsigned FLAC__ENTROPY_CODING_METHOD_PARTITIONED_RICE2_ESCAPE_PARAMETER;
extern const unsigned FLAC__SUBFRAME_LPC_QLP_COEFF_PRECISION_LEN;
extern const unsigned FLAC__SUBFRAME_LPC_QLP_SHIFT_LEN;
extern const unsigned FLAC__SUBFRAME_ZERO_PAD_LEN;
extern const unsigned FLAC__SUBFRAME_TYPE_LEN;
extern const unsigned FLAC__SUBFRAME_WASTED_BITS_FLAG_LEN;
extern const unsigned FLAC__SUBFRAME_TYPE_CONSTANT_BYTE_ALIGNED_MASK;
extern const unsigned FLAC__SUBFRAME_TYPE_VERBATIM_BYTE_ALIGNED_MASK;
extern const unsigned FLAC__SUBFRAME_TYPE_FIXED_BYTE_ALIGNED_MASK;
extern const unsigned FLAC__SUBFRAME_TYPE_LPC_BYTE_ALIGNED_MASK;
extern const char *const FLAC__EntropyCodingMethodTypeString[];
extern const char *const FLAC__SubframeTypeString[];
extern const char *const FLAC__ChannelAssignmentString[];
extern const char *const FLAC__FrameNumberTypeString[];
extern const char *const FLAC__MetadataTypeString[];

/* src/libFLAC/format.c:213-314 */
FLAC__bool FLAC__format_sample_rate_is_valid(unsigned sample_rate);
FLAC__bool FLAC__format_blocksize_is_subset(unsigned blocksize, unsigned sample_rate);
FLAC__bool FLAC__format_sample_rate_is_subset(unsigned sample_rate);
FLAC__bool FLAC__format_seektable_is_legal(const FLAC__StreamMetadata_SeekTable *seek_table);
unsigned FLAC__format_seektable_sort(FLAC__StreamMetadata_SeekTable *seek_table);

/* =============================================================================================
 * stream encoder (include/FLAC/stream_encoder.h)
 * =========================================================================================== */
/* stream_encoder.h:240-280 */
typedef enum {
	FLAC__STREAM_ENCODER_OK = 0, FLAC__STREAM_ENCODER_UNINITIALIZED, FLAC__STREAM_ENCODER_OGG_ERROR,
	FLAC__STREAM_ENCODER_VERIFY_DECODER_ERROR, FLAC__STREAM_ENCODER_VERIFY_MISMATCH_IN_AUDIO_DATA,
	FLAC__STREAM_ENCODER_CLIENT_ERROR, FLAC__STREAM_ENCODER_IO_ERROR, FLAC__STREAM_ENCODER_FRAMING_ERROR,
	FLAC__STREAM_ENCODER_MEMORY_ALLOCATION_ERROR
} FLAC__StreamEncoderState;
/* stream_encoder.h:292-350 */
typedef enum {
	FLAC__STREAM_ENCODER_INIT_STATUS_OK = 0, FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR,
	FLAC__STREAM_ENCODER_INIT_STATUS_UNSUPPORTED_CONTAINER, FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_CALLBACKS,
	FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_NUMBER_OF_CHANNELS, FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_BITS_PER_SAMPLE,
	FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_SAMPLE_RATE, FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_BLOCK_SIZE,
	FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_MAX_LPC_ORDER, FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_QLP_COEFF_PRECISION,
	FLAC__STREAM_ENCODER_INIT_STATUS_BLOCK_SIZE_TOO_SMALL_FOR_LPC_ORDER, FLAC__STREAM_ENCODER_INIT_STATUS_NOT_STREAMABLE,
	FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_METADATA, FLAC__STREAM_ENCODER_INIT_STATUS_ALREADY_INITIALIZED
} FLAC__StreamEncoderInitStatus;
/* stream_encoder.h:362-442 */
typedef enum { FLAC__STREAM_ENCODER_READ_STATUS_CONTINUE, FLAC__STREAM_ENCODER_READ_STATUS_END_OF_STREAM, FLAC__STREAM_ENCODER_READ_STATUS_ABORT, FLAC__STREAM_ENCODER_READ_STATUS_UNSUPPORTED } FLAC__StreamEncoderReadStatus;
typedef enum { FLAC__STREAM_ENCODER_WRITE_STATUS_OK = 0, FLAC__STREAM_ENCODER_WRITE_STATUS_FATAL_ERROR } FLAC__StreamEncoderWriteStatus;
typedef enum { FLAC__STREAM_ENCODER_SEEK_STATUS_OK, FLAC__STREAM_ENCODER_SEEK_STATUS_ERROR, FLAC__STREAM_ENCODER_SEEK_STATUS_UNSUPPORTED } FLAC__StreamEncoderSeekStatus;
typedef enum { FLAC__STREAM_ENCODER_TELL_STATUS_OK, FLAC__STREAM_ENCODER_TELL_STATUS_ERROR, FLAC__STREAM_ENCODER_TELL_STATUS_UNSUPPORTED } FLAC__StreamEncoderTellStatus;

extern const char *const FLAC__StreamEncoderStateString[];
extern const char *const FLAC__StreamEncoderInitStatusString[];
extern const char *const FLAC__treamEncoderReadStatusString[]; /* the reference's spelling (stream_encoder.c:480) */
extern const char *const FLAC__StreamEncoderWriteStatusString[];
extern const char *const FLAC__StreamEncoderSeekStatusString[];
extern const char *const FLAC__StreamEncoderTellStatusString[];

/* stream_encoder.h:464-467: opaque to clients */
struct FLAC__StreamEncoderProtected;
struct FLAC__StreamEncoderPrivate;
typedef struct { struct FLAC__StreamEncoderProtected *protected_; struct FLAC__StreamEncoderPrivate *private_; } FLAC__StreamEncoder;

/* stream_encoder.h:518,556,590,635,654,677 */
typedef FLAC__StreamEncoderReadStatus (*FLAC__StreamEncoderReadCallback)(const FLAC__StreamEncoder *encoder, FLAC__byte buffer[], size_t *bytes, void *client_data);
typedef FLAC__StreamEncoderWriteStatus (*FLAC__StreamEncoderWriteCallback)(const FLAC__StreamEncoder *encoder, const FLAC__byte buffer[], size_t bytes, unsigned samples, unsigned current_frame, void *client_data);
typedef FLAC__StreamEncoderSeekStatus (*FLAC__StreamEncoderSeekCallback)(const FLAC__StreamEncoder *encoder, FLAC__uint64 absolute_byte_offset, void *client_data);
typedef FLAC__StreamEncoderTellStatus (*FLAC__StreamEncoderTellCallback)(const FLAC__StreamEncoder *encoder, FLAC__uint64 *absolute_byte_offset, void *client_data);
typedef void (*FLAC__StreamEncoderMetadataCallback)(const FLAC__StreamEncoder *encoder, const FLAC__StreamMetadata *metadata, void *client_data);
typedef void (*FLAC__StreamEncoderProgressCallback)(const FLAC__StreamEncoder *encoder, FLAC__uint64 bytes_written, FLAC__uint64 samples_written, unsigned frames_written, unsigned total_frames_estimate, void *client_data);

/* stream_encoder.h:693-703 */
FLAC__StreamEncoder *FLAC__stream_encoder_new(void);
void FLAC__stream_encoder_delete(FLAC__StreamEncoder *encoder);
/* stream_encoder.h:727-1181 (setters: false unless the encoder is uninitialised) */
FLAC__bool FLAC__stream_encoder_set_ogg_serial_number(FLAC__StreamEncoder *encoder, long serial_number);
FLAC__bool FLAC__stream_encoder_set_verify(FLAC__StreamEncoder *encoder, FLAC__bool value);
FLAC__bool FLAC__stream_encoder_set_streamable_subset(FLAC__StreamEncoder *encoder, FLAC__bool value);
FLAC__bool FLAC__stream_encoder_set_channels(FLAC__StreamEncoder *encoder, unsigned value);
FLAC__bool FLAC__stream_encoder_set_bits_per_sample(FLAC__StreamEncoder *encoder, unsigned value);
FLAC__bool FLAC__stream_encoder_set_sample_rate(FLAC__StreamEncoder *encoder, unsigned value);
FLAC__bool FLAC__stream_encoder_set_compression_level(FLAC__StreamEncoder *encoder, unsigned value);
FLAC__bool FLAC__stream_encoder_set_blocksize(FLAC__StreamEncoder *encoder, unsigned value);
FLAC__bool FLAC__stream_encoder_set_do_mid_side_stereo(FLAC__StreamEncoder *encoder, FLAC__bool value);
FLAC__bool FLAC__stream_encoder_set_loose_mid_side_stereo(FLAC__StreamEncoder *encoder, FLAC__bool value);
FLAC__bool FLAC__stream_encoder_set_apodization(FLAC__StreamEncoder *encoder, const char *specification);
FLAC__bool FLAC__stream_encoder_set_max_lpc_order(FLAC__StreamEncoder *encoder, unsigned value);
FLAC__bool FLAC__stream_encoder_set_qlp_coeff_precision(FLAC__StreamEncoder *encoder, unsigned value);
FLAC__bool FLAC__stream_encoder_set_do_qlp_coeff_prec_search(FLAC__StreamEncoder *encoder, FLAC__bool value);
FLAC__bool FLAC__stream_encoder_set_do_escape_coding(FLAC__StreamEncoder *encoder, FLAC__bool value);
FLAC__bool FLAC__stream_encoder_set_do_exhaustive_model_search(FLAC__StreamEncoder *encoder, FLAC__bool value);
FLAC__bool FLAC__stream_encoder_set_min_residual_partition_order(FLAC__StreamEncoder *encoder, unsigned value);
FLAC__bool FLAC__stream_encoder_set_max_residual_partition_order(FLAC__StreamEncoder *encoder, unsigned value);
FLAC__bool FLAC__stream_encoder_set_rice_parameter_search_dist(FLAC__StreamEncoder *encoder, unsigned value);
FLAC__bool FLAC__stream_encoder_set_total_samples_estimate(FLAC__StreamEncoder *encoder, FLAC__uint64 value);
FLAC__bool FLAC__stream_encoder_set_metadata(FLAC__StreamEncoder *encoder, FLAC__StreamMetadata **metadata, unsigned num_blocks);
/* exported by the reference but not declared in its public header (stream_encoder.c:1414,1742-1773,1840) */
FLAC__bool FLAC__stream_encoder_set_do_md5(FLAC__StreamEncoder *encoder, FLAC__bool value);
FLAC__bool FLAC__stream_encoder_get_do_md5(const FLAC__StreamEncoder *encoder);
FLAC__bool FLAC__stream_encoder_disable_constant_subframes(FLAC__StreamEncoder *encoder, FLAC__bool value);
FLAC__bool FLAC__stream_encoder_disable_fixed_subframes(FLAC__StreamEncoder *encoder, FLAC__bool value);
FLAC__bool FLAC__stream_encoder_disable_verbatim_subframes(FLAC__StreamEncoder *encoder, FLAC__bool value);
/* stream_encoder.h:1191-1408 */
FLAC__StreamEncoderState FLAC__stream_encoder_get_state(const FLAC__StreamEncoder *encoder);
int FLAC__stream_encoder_get_verify_decoder_state(const FLAC__StreamEncoder *encoder);
const char *FLAC__stream_encoder_get_resolved_state_string(const FLAC__StreamEncoder *encoder);
void FLAC__stream_encoder_get_verify_decoder_error_stats(const FLAC__StreamEncoder *encoder, FLAC__uint64 *absolute_sample, unsigned *frame_number, unsigned *channel, unsigned *sample, FLAC__int32 *expected, FLAC__int32 *got);
FLAC__bool FLAC__stream_encoder_get_verify(const FLAC__StreamEncoder *encoder);
FLAC__bool FLAC__stream_encoder_get_streamable_subset(const FLAC__StreamEncoder *encoder);
unsigned FLAC__stream_encoder_get_channels(const FLAC__StreamEncoder *encoder);
unsigned FLAC__stream_encoder_get_bits_per_sample(const FLAC__StreamEncoder *encoder);
unsigned FLAC__stream_encoder_get_sample_rate(const FLAC__StreamEncoder *encoder);
unsigned FLAC__stream_encoder_get_blocksize(const FLAC__StreamEncoder *encoder);
FLAC__bool FLAC__stream_encoder_get_do_mid_side_stereo(const FLAC__StreamEncoder *encoder);
FLAC__bool FLAC__stream_encoder_get_loose_mid_side_stereo(const FLAC__StreamEncoder *encoder);
unsigned FLAC__stream_encoder_get_max_lpc_order(const FLAC__StreamEncoder *encoder);
unsigned FLAC__stream_encoder_get_qlp_coeff_precision(const FLAC__StreamEncoder *encoder);
FLAC__bool FLAC__stream_encoder_get_do_qlp_coeff_prec_search(const FLAC__StreamEncoder *encoder);
FLAC__bool FLAC__stream_encoder_get_do_escape_coding(const FLAC__StreamEncoder *encoder);
FLAC__bool FLAC__stream_encoder_get_do_exhaustive_model_search(const FLAC__StreamEncoder *encoder);
unsigned FLAC__stream_encoder_get_min_residual_partition_order(const FLAC__StreamEncoder *encoder);
unsigned FLAC__stream_encoder_get_max_residual_partition_order(const FLAC__StreamEncoder *encoder);
unsigned FLAC__stream_encoder_get_rice_parameter_search_dist(const FLAC__StreamEncoder *encoder);
FLAC__uint64 FLAC__stream_encoder_get_total_samples_estimate(const FLAC__StreamEncoder *encoder);
/* stream_encoder.h:1471-1673 */
FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_stream(FLAC__StreamEncoder *encoder, FLAC__StreamEncoderWriteCallback write_callback, FLAC__StreamEncoderSeekCallback seek_callback, FLAC__StreamEncoderTellCallback tell_callback, FLAC__StreamEncoderMetadataCallback metadata_callback, void *client_data);
FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_ogg_stream(FLAC__StreamEncoder *encoder, FLAC__StreamEncoderReadCallback read_callback, FLAC__StreamEncoderWriteCallback write_callback, FLAC__StreamEncoderSeekCallback seek_callback, FLAC__StreamEncoderTellCallback tell_callback, FLAC__StreamEncoderMetadataCallback metadata_callback, void *client_data);
FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_FILE(FLAC__StreamEncoder *encoder, FILE *file, FLAC__StreamEncoderProgressCallback progress_callback, void *client_data);
FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_ogg_FILE(FLAC__StreamEncoder *encoder, FILE *file, FLAC__StreamEncoderProgressCallback progress_callback, void *client_data);
FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_file(FLAC__StreamEncoder *encoder, const char *filename, FLAC__StreamEncoderProgressCallback progress_callback, void *client_data);
FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_ogg_file(FLAC__StreamEncoder *encoder, const char *filename, FLAC__StreamEncoderProgressCallback progress_callback, void *client_data);
/* stream_encoder.h:1701,1728,1760 */
FLAC__bool FLAC__stream_encoder_finish(FLAC__StreamEncoder *encoder);
FLAC__bool FLAC__stream_encoder_process(FLAC__StreamEncoder *encoder, const FLAC__int32 *const buffer[], unsigned samples);
FLAC__bool FLAC__stream_encoder_process_interleaved(FLAC__StreamEncoder *encoder, const FLAC__int32 buffer[], unsigned samples);

/* =============================================================================================
 * stream decoder (include/FLAC/stream_decoder.h)
 * =========================================================================================== */
/* stream_decoder.h:201-243 */
typedef enum {
	FLAC__STREAM_DECODER_SEARCH_FOR_METADATA = 0, FLAC__STREAM_DECODER_READ_METADATA, FLAC__STREAM_DECODER_SEARCH_FOR_FRAME_SYNC,
	FLAC__STREAM_DECODER_READ_FRAME, FLAC__STREAM_DECODER_END_OF_STREAM, FLAC__STREAM_DECODER_OGG_ERROR,
	FLAC__STREAM_DECODER_SEEK_ERROR, FLAC__STREAM_DECODER_ABORTED, FLAC__STREAM_DECODER_MEMORY_ALLOCATION_ERROR,
	FLAC__STREAM_DECODER_UNINITIALIZED
} FLAC__StreamDecoderState;
/* stream_decoder.h:255-281 */
typedef enum {
	FLAC__STREAM_DECODER_INIT_STATUS_OK = 0, FLAC__STREAM_DECODER_INIT_STATUS_UNSUPPORTED_CONTAINER,
	FLAC__STREAM_DECODER_INIT_STATUS_INVALID_CALLBACKS, FLAC__STREAM_DECODER_INIT_STATUS_MEMORY_ALLOCATION_ERROR,
	FLAC__STREAM_DECODER_INIT_STATUS_ERROR_OPENING_FILE, FLAC__STREAM_DECODER_INIT_STATUS_ALREADY_INITIALIZED
} FLAC__StreamDecoderInitStatus;
/* stream_decoder.h:293-440 */
typedef enum { FLAC__STREAM_DECODER_READ_STATUS_CONTINUE, FLAC__STREAM_DECODER_READ_STATUS_END_OF_STREAM, FLAC__STREAM_DECODER_READ_STATUS_ABORT } FLAC__StreamDecoderReadStatus;
typedef enum { FLAC__STREAM_DECODER_SEEK_STATUS_OK, FLAC__STREAM_DECODER_SEEK_STATUS_ERROR, FLAC__STREAM_DECODER_SEEK_STATUS_UNSUPPORTED } FLAC__StreamDecoderSeekStatus;
typedef enum { FLAC__STREAM_DECODER_TELL_STATUS_OK, FLAC__STREAM_DECODER_TELL_STATUS_ERROR, FLAC__STREAM_DECODER_TELL_STATUS_UNSUPPORTED } FLAC__StreamDecoderTellStatus;
typedef enum { FLAC__STREAM_DECODER_LENGTH_STATUS_OK, FLAC__STREAM_DECODER_LENGTH_STATUS_ERROR, FLAC__STREAM_DECODER_LENGTH_STATUS_UNSUPPORTED } FLAC__StreamDecoderLengthStatus;
typedef enum { FLAC__STREAM_DECODER_WRITE_STATUS_CONTINUE, FLAC__STREAM_DECODER_WRITE_STATUS_ABORT } FLAC__StreamDecoderWriteStatus;
typedef enum {
	FLAC__STREAM_DECODER_ERROR_STATUS_LOST_SYNC, FLAC__STREAM_DECODER_ERROR_STATUS_BAD_HEADER,
	FLAC__STREAM_DECODER_ERROR_STATUS_FRAME_CRC_MISMATCH, FLAC__STREAM_DECODER_ERROR_STATUS_UNPARSEABLE_STREAM
} FLAC__StreamDecoderErrorStatus;

extern const char *const FLAC__StreamDecoderStateString[];
extern const char *const FLAC__StreamDecoderInitStatusString[];
extern const char *const FLAC__StreamDecoderReadStatusString[];
extern const char *const FLAC__StreamDecoderSeekStatusString[];
extern const char *const FLAC__StreamDecoderTellStatusString[];
extern const char *const FLAC__StreamDecoderLengthStatusString[];
extern const char *const FLAC__StreamDecoderWriteStatusString[];
extern const char *const FLAC__StreamDecoderErrorStatusString[];

/* stream_decoder.h:462-465 */
struct FLAC__StreamDecoderProtected;
struct FLAC__StreamDecoderPrivate;
typedef struct { struct FLAC__StreamDecoderProtected *protected_; struct FLAC__StreamDecoderPrivate *private_; } FLAC__StreamDecoder;

/* stream_decoder.h:516-726 */
typedef FLAC__StreamDecoderReadStatus (*FLAC__StreamDecoderReadCallback)(const FLAC__StreamDecoder *decoder, FLAC__byte buffer[], size_t *bytes, void *client_data);
typedef FLAC__StreamDecoderSeekStatus (*FLAC__StreamDecoderSeekCallback)(const FLAC__StreamDecoder *decoder, FLAC__uint64 absolute_byte_offset, void *client_data);
typedef FLAC__StreamDecoderTellStatus (*FLAC__StreamDecoderTellCallback)(const FLAC__StreamDecoder *decoder, FLAC__uint64 *absolute_byte_offset, void *client_data);
typedef FLAC__StreamDecoderLengthStatus (*FLAC__StreamDecoderLengthCallback)(const FLAC__StreamDecoder *decoder, FLAC__uint64 *stream_length, void *client_data);
typedef FLAC__bool (*FLAC__StreamDecoderEofCallback)(const FLAC__StreamDecoder *decoder, void *client_data);
typedef FLAC__StreamDecoderWriteStatus (*FLAC__StreamDecoderWriteCallback)(const FLAC__StreamDecoder *decoder, const FLAC__Frame *frame, const FLAC__int32 *const buffer[], void *client_data);
typedef void (*FLAC__StreamDecoderMetadataCallback)(const FLAC__StreamDecoder *decoder, const FLAC__StreamMetadata *metadata, void *client_data);
typedef void (*FLAC__StreamDecoderErrorCallback)(const FLAC__StreamDecoder *decoder, FLAC__StreamDecoderErrorStatus status, void *client_data);

/* stream_decoder.h:742-750 */
FLAC__StreamDecoder *FLAC__stream_decoder_new(void);
void FLAC__stream_decoder_delete(FLAC__StreamDecoder *decoder);
/* stream_decoder.h:775-879 */
FLAC__bool FLAC__stream_decoder_set_ogg_serial_number(FLAC__StreamDecoder *decoder, long serial_number);
FLAC__bool FLAC__stream_decoder_set_md5_checking(FLAC__StreamDecoder *decoder, FLAC__bool value);
FLAC__bool FLAC__stream_decoder_set_metadata_respond(FLAC__StreamDecoder *decoder, FLAC__MetadataType type);
FLAC__bool FLAC__stream_decoder_set_metadata_respond_application(FLAC__StreamDecoder *decoder, const FLAC__byte id[4]);
FLAC__bool FLAC__stream_decoder_set_metadata_respond_all(FLAC__StreamDecoder *decoder);
FLAC__bool FLAC__stream_decoder_set_metadata_ignore(FLAC__StreamDecoder *decoder, FLAC__MetadataType type);
FLAC__bool FLAC__stream_decoder_set_metadata_ignore_application(FLAC__StreamDecoder *decoder, const FLAC__byte id[4]);
FLAC__bool FLAC__stream_decoder_set_metadata_ignore_all(FLAC__StreamDecoder *decoder);
/* stream_decoder.h:889-1006 */
FLAC__StreamDecoderState FLAC__stream_decoder_get_state(const FLAC__StreamDecoder *decoder);
const char *FLAC__stream_decoder_get_resolved_state_string(const FLAC__StreamDecoder *decoder);
FLAC__bool FLAC__stream_decoder_get_md5_checking(const FLAC__StreamDecoder *decoder);
FLAC__uint64 FLAC__stream_decoder_get_total_samples(const FLAC__StreamDecoder *decoder);
unsigned FLAC__stream_decoder_get_channels(const FLAC__StreamDecoder *decoder);
FLAC__ChannelAssignment FLAC__stream_decoder_get_channel_assignment(const FLAC__StreamDecoder *decoder);
unsigned FLAC__stream_decoder_get_bits_per_sample(const FLAC__StreamDecoder *decoder);
unsigned FLAC__stream_decoder_get_sample_rate(const FLAC__StreamDecoder *decoder);
unsigned FLAC__stream_decoder_get_blocksize(const FLAC__StreamDecoder *decoder);
FLAC__bool FLAC__stream_decoder_get_decode_position(const FLAC__StreamDecoder *decoder, FLAC__uint64 *position);
/* stream_decoder.h:1072-1341 */
FLAC__StreamDecoderInitStatus FLAC__stream_decoder_init_stream(FLAC__StreamDecoder *decoder, FLAC__StreamDecoderReadCallback read_callback, FLAC__StreamDecoderSeekCallback seek_callback, FLAC__StreamDecoderTellCallback tell_callback, FLAC__StreamDecoderLengthCallback length_callback, FLAC__StreamDecoderEofCallback eof_callback, FLAC__StreamDecoderWriteCallback write_callback, FLAC__StreamDecoderMetadataCallback metadata_callback, FLAC__StreamDecoderErrorCallback error_callback, void *client_data);
FLAC__StreamDecoderInitStatus FLAC__stream_decoder_init_ogg_stream(FLAC__StreamDecoder *decoder, FLAC__StreamDecoderReadCallback read_callback, FLAC__StreamDecoderSeekCallback seek_callback, FLAC__StreamDecoderTellCallback tell_callback, FLAC__StreamDecoderLengthCallback length_callback, FLAC__StreamDecoderEofCallback eof_callback, FLAC__StreamDecoderWriteCallback write_callback, FLAC__StreamDecoderMetadataCallback metadata_callback, FLAC__StreamDecoderErrorCallback error_callback, void *client_data);
FLAC__StreamDecoderInitStatus FLAC__stream_decoder_init_FILE(FLAC__StreamDecoder *decoder, FILE *file, FLAC__StreamDecoderWriteCallback write_callback, FLAC__StreamDecoderMetadataCallback metadata_callback, FLAC__StreamDecoderErrorCallback error_callback, void *client_data);
FLAC__StreamDecoderInitStatus FLAC__stream_decoder_init_ogg_FILE(FLAC__StreamDecoder *decoder, FILE *file, FLAC__StreamDecoderWriteCallback write_callback, FLAC__StreamDecoderMetadataCallback metadata_callback, FLAC__StreamDecoderErrorCallback error_callback, void *client_data);
FLAC__StreamDecoderInitStatus FLAC__stream_decoder_init_file(FLAC__StreamDecoder *decoder, const char *filename, FLAC__StreamDecoderWriteCallback write_callback, FLAC__StreamDecoderMetadataCallback metadata_callback, FLAC__StreamDecoderErrorCallback error_callback, void *client_data);
FLAC__StreamDecoderInitStatus FLAC__stream_decoder_init_ogg_file(FLAC__StreamDecoder *decoder, const char *filename, FLAC__StreamDecoderWriteCallback write_callback, FLAC__StreamDecoderMetadataCallback metadata_callback, FLAC__StreamDecoderErrorCallback error_callback, void *client_data);
/* stream_decoder.h:1369-1551 */
FLAC__bool FLAC__stream_decoder_finish(FLAC__StreamDecoder *decoder);
FLAC__bool FLAC__stream_decoder_flush(FLAC__StreamDecoder *decoder);
FLAC__bool FLAC__stream_decoder_reset(FLAC__StreamDecoder *decoder);
FLAC__bool FLAC__stream_decoder_process_single(FLAC__StreamDecoder *decoder);
FLAC__bool FLAC__stream_decoder_process_until_end_of_metadata(FLAC__StreamDecoder *decoder);
FLAC__bool FLAC__stream_decoder_process_until_end_of_stream(FLAC__StreamDecoder *decoder);
FLAC__bool FLAC__stream_decoder_skip_single_frame(FLAC__StreamDecoder *decoder);
FLAC__bool FLAC__stream_decoder_seek_absolute(FLAC__StreamDecoder *decoder, FLAC__uint64 sample);

/* =============================================================================================
 * metadata objects and level-0 readers (include/FLAC/metadata.h; src/libFLAC/metadata_object.c,
 * metadata_iterators.c:166-333, format.c:363-523).  The level-1 / level-2 file editors (simple
 * iterator, chain) are not part of this library.
 * =========================================================================================== */
/* format.h:96-120,187-209 */
extern const unsigned FLAC__STREAM_METADATA_CUESHEET_INDEX_OFFSET_LEN;
extern const unsigned FLAC__STREAM_METADATA_CUESHEET_INDEX_NUMBER_LEN;
extern const unsigned FLAC__STREAM_METADATA_CUESHEET_INDEX_RESERVED_LEN;
extern const unsigned FLAC__STREAM_METADATA_CUESHEET_TRACK_OFFSET_LEN;
extern const unsigned FLAC__STREAM_METADATA_CUESHEET_TRACK_NUMBER_LEN;
extern const unsigned FLAC__STREAM_METADATA_CUESHEET_TRACK_ISRC_LEN;
extern const unsigned FLAC__STREAM_METADATA_CUESHEET_TRACK_TYPE_LEN;
extern const unsigned FLAC__STREAM_METADATA_CUESHEET_TRACK_PRE_EMPHASIS_LEN;
extern const unsigned FLAC__STREAM_METADATA_CUESHEET_TRACK_RESERVED_LEN;
extern const unsigned FLAC__STREAM_METADATA_CUESHEET_TRACK_NUM_INDICES_LEN;
extern const unsigned FLAC__STREAM_METADATA_CUESHEET_MEDIA_CATALOG_NUMBER_LEN;
extern const unsigned FLAC__STREAM_METADATA_CUESHEET_LEAD_IN_LEN;
extern const unsigned FLAC__STREAM_METADATA_CUESHEET_IS_CD_LEN;
extern const unsigned FLAC__STREAM_METADATA_CUESHEET_RESERVED_LEN;
extern const unsigned FLAC__STREAM_METADATA_CUESHEET_NUM_TRACKS_LEN;
extern const unsigned FLAC__STREAM_METADATA_PICTURE_TYPE_LEN;
extern const unsigned FLAC__STREAM_METADATA_PICTURE_MIME_TYPE_LENGTH_LEN;
extern const unsigned FLAC__STREAM_METADATA_PICTURE_DESCRIPTION_LENGTH_LEN;
extern const unsigned FLAC__STREAM_METADATA_PICTURE_WIDTH_LEN;
extern const unsigned FLAC__STREAM_METADATA_PICTURE_HEIGHT_LEN;
extern const unsigned FLAC__STREAM_METADATA_PICTURE_DEPTH_LEN;
extern const unsigned FLAC__STREAM_METADATA_PICTURE_COLORS_LEN;
extern const unsigned FLAC__STREAM_METADATA_PICTURE_DATA_LENGTH_LEN;
extern const char *const FLAC__StreamMetadata_Picture_TypeString[];
/* format.h:917-1006 */
FLAC__bool FLAC__format_vorbiscomment_entry_name_is_legal(const char *name);
FLAC__bool FLAC__format_vorbiscomment_entry_value_is_legal(const FLAC__byte *value, unsigned length);
FLAC__bool FLAC__format_vorbiscomment_entry_is_legal(const FLAC__byte *entry, unsigned length);
FLAC__bool FLAC__format_cuesheet_is_legal(const FLAC__StreamMetadata_CueSheet *cue_sheet, FLAC__bool check_cd_da_subset, const char **violation);
FLAC__bool FLAC__format_picture_is_legal(const FLAC__StreamMetadata_Picture *picture, const char **violation);
/* metadata.h:158-236 */
FLAC__bool FLAC__metadata_get_streaminfo(const char *filename, FLAC__StreamMetadata *streaminfo);
FLAC__bool FLAC__metadata_get_tags(const char *filename, FLAC__StreamMetadata **tags);
FLAC__bool FLAC__metadata_get_cuesheet(const char *filename, FLAC__StreamMetadata **cuesheet);
FLAC__bool FLAC__metadata_get_picture(const char *filename, FLAC__StreamMetadata **picture, FLAC__StreamMetadata_Picture_Type type, const char *mime_type, const FLAC__byte *description, unsigned max_width, unsigned max_height, unsigned max_depth, unsigned max_colors);
/* metadata.h:1312-1375; private/metadata.h for the two *_delete_data */
FLAC__StreamMetadata *FLAC__metadata_object_new(FLAC__MetadataType type);
FLAC__StreamMetadata *FLAC__metadata_object_clone(const FLAC__StreamMetadata *object);
void FLAC__metadata_object_delete(FLAC__StreamMetadata *object);
void FLAC__metadata_object_delete_data(FLAC__StreamMetadata *object);
FLAC__bool FLAC__metadata_object_is_equal(const FLAC__StreamMetadata *block1, const FLAC__StreamMetadata *block2);
FLAC__bool FLAC__metadata_object_application_set_data(FLAC__StreamMetadata *object, FLAC__byte *data, unsigned length, FLAC__bool copy);
/* metadata.h:1392-1563 */
FLAC__bool FLAC__metadata_object_seektable_resize_points(FLAC__StreamMetadata *object, unsigned new_num_points);
void FLAC__metadata_object_seektable_set_point(FLAC__StreamMetadata *object, unsigned point_num, FLAC__StreamMetadata_SeekPoint point);
FLAC__bool FLAC__metadata_object_seektable_insert_point(FLAC__StreamMetadata *object, unsigned point_num, FLAC__StreamMetadata_SeekPoint point);
FLAC__bool FLAC__metadata_object_seektable_delete_point(FLAC__StreamMetadata *object, unsigned point_num);
FLAC__bool FLAC__metadata_object_seektable_is_legal(const FLAC__StreamMetadata *object);
FLAC__bool FLAC__metadata_object_seektable_template_append_placeholders(FLAC__StreamMetadata *object, unsigned num);
FLAC__bool FLAC__metadata_object_seektable_template_append_point(FLAC__StreamMetadata *object, FLAC__uint64 sample_number);
FLAC__bool FLAC__metadata_object_seektable_template_append_points(FLAC__StreamMetadata *object, FLAC__uint64 sample_numbers[], unsigned num);
FLAC__bool FLAC__metadata_object_seektable_template_append_spaced_points(FLAC__StreamMetadata *object, unsigned num, FLAC__uint64 total_samples);
FLAC__bool FLAC__metadata_object_seektable_template_append_spaced_points_by_samples(FLAC__StreamMetadata *object, unsigned samples, FLAC__uint64 total_samples);
FLAC__bool FLAC__metadata_object_seektable_template_sort(FLAC__StreamMetadata *object, FLAC__bool compact);
/* metadata.h:1588-1836 */
FLAC__bool FLAC__metadata_object_vorbiscomment_set_vendor_string(FLAC__StreamMetadata *object, FLAC__StreamMetadata_VorbisComment_Entry entry, FLAC__bool copy);
FLAC__bool FLAC__metadata_object_vorbiscomment_resize_comments(FLAC__StreamMetadata *object, unsigned new_num_comments);
FLAC__bool FLAC__metadata_object_vorbiscomment_set_comment(FLAC__StreamMetadata *object, unsigned comment_num, FLAC__StreamMetadata_VorbisComment_Entry entry, FLAC__bool copy);
FLAC__bool FLAC__metadata_object_vorbiscomment_insert_comment(FLAC__StreamMetadata *object, unsigned comment_num, FLAC__StreamMetadata_VorbisComment_Entry entry, FLAC__bool copy);
FLAC__bool FLAC__metadata_object_vorbiscomment_append_comment(FLAC__StreamMetadata *object, FLAC__StreamMetadata_VorbisComment_Entry entry, FLAC__bool copy);
FLAC__bool FLAC__metadata_object_vorbiscomment_replace_comment(FLAC__StreamMetadata *object, FLAC__StreamMetadata_VorbisComment_Entry entry, FLAC__bool all, FLAC__bool copy);
FLAC__bool FLAC__metadata_object_vorbiscomment_delete_comment(FLAC__StreamMetadata *object, unsigned comment_num);
FLAC__bool FLAC__metadata_object_vorbiscomment_entry_from_name_value_pair(FLAC__StreamMetadata_VorbisComment_Entry *entry, const char *field_name, const char *field_value);
FLAC__bool FLAC__metadata_object_vorbiscomment_entry_to_name_value_pair(const FLAC__StreamMetadata_VorbisComment_Entry entry, char **field_name, char **field_value);
FLAC__bool FLAC__metadata_object_vorbiscomment_entry_matches(const FLAC__StreamMetadata_VorbisComment_Entry entry, const char *field_name, unsigned field_name_length);
int FLAC__metadata_object_vorbiscomment_find_entry_from(const FLAC__StreamMetadata *object, unsigned offset, const char *field_name);
int FLAC__metadata_object_vorbiscomment_remove_entry_matching(FLAC__StreamMetadata *object, const char *field_name);
int FLAC__metadata_object_vorbiscomment_remove_entries_matching(FLAC__StreamMetadata *object, const char *field_name);
/* metadata.h:1845-2084 */
FLAC__StreamMetadata_CueSheet_Track *FLAC__metadata_object_cuesheet_track_new(void);
FLAC__StreamMetadata_CueSheet_Track *FLAC__metadata_object_cuesheet_track_clone(const FLAC__StreamMetadata_CueSheet_Track *object);
void FLAC__metadata_object_cuesheet_track_delete(FLAC__StreamMetadata_CueSheet_Track *object);
void FLAC__metadata_object_cuesheet_track_delete_data(FLAC__StreamMetadata_CueSheet_Track *object);
FLAC__bool FLAC__metadata_object_cuesheet_track_resize_indices(FLAC__StreamMetadata *object, unsigned track_num, unsigned new_num_indices);
FLAC__bool FLAC__metadata_object_cuesheet_track_insert_index(FLAC__StreamMetadata *object, unsigned track_num, unsigned index_num, FLAC__StreamMetadata_CueSheet_Index index);
FLAC__bool FLAC__metadata_object_cuesheet_track_insert_blank_index(FLAC__StreamMetadata *object, unsigned track_num, unsigned index_num);
FLAC__bool FLAC__metadata_object_cuesheet_track_delete_index(FLAC__StreamMetadata *object, unsigned track_num, unsigned index_num);
FLAC__bool FLAC__metadata_object_cuesheet_resize_tracks(FLAC__StreamMetadata *object, unsigned new_num_tracks);
FLAC__bool FLAC__metadata_object_cuesheet_set_track(FLAC__StreamMetadata *object, unsigned track_num, FLAC__StreamMetadata_CueSheet_Track *track, FLAC__bool copy);
FLAC__bool FLAC__metadata_object_cuesheet_insert_track(FLAC__StreamMetadata *object, unsigned track_num, FLAC__StreamMetadata_CueSheet_Track *track, FLAC__bool copy);
FLAC__bool FLAC__metadata_object_cuesheet_insert_blank_track(FLAC__StreamMetadata *object, unsigned track_num);
FLAC__bool FLAC__metadata_object_cuesheet_delete_track(FLAC__StreamMetadata *object, unsigned track_num);
FLAC__bool FLAC__metadata_object_cuesheet_is_legal(const FLAC__StreamMetadata *object, FLAC__bool check_cd_da_subset, const char **violation);
FLAC__uint32 FLAC__metadata_object_cuesheet_calculate_cddb_id(const FLAC__StreamMetadata *object);
/* metadata.h:2107-2173 */
FLAC__bool FLAC__metadata_object_picture_set_mime_type(FLAC__StreamMetadata *object, char *mime_type, FLAC__bool copy);
FLAC__bool FLAC__metadata_object_picture_set_description(FLAC__StreamMetadata *object, FLAC__byte *description, FLAC__bool copy);
FLAC__bool FLAC__metadata_object_picture_set_data(FLAC__StreamMetadata *object, FLAC__byte *data, FLAC__uint32 length, FLAC__bool copy);
FLAC__bool FLAC__metadata_object_picture_is_legal(const FLAC__StreamMetadata *object, const char **violation);

/* ---- metadata level 1 and level 2: editing the metadata of a FLAC file in place (host code) ---- */
#include <sys/types.h>
/* include/FLAC/callback.h:85-176 */
typedef void *FLAC__IOHandle;
typedef size_t (*FLAC__IOCallback_Read)(void *ptr, size_t size, size_t nmemb, FLAC__IOHandle handle);
typedef size_t (*FLAC__IOCallback_Write)(const void *ptr, size_t size, size_t nmemb, FLAC__IOHandle handle);
typedef int (*FLAC__IOCallback_Seek)(FLAC__IOHandle handle, FLAC__int64 offset, int whence);
typedef FLAC__int64 (*FLAC__IOCallback_Tell)(FLAC__IOHandle handle);
typedef int (*FLAC__IOCallback_Eof)(FLAC__IOHandle handle);
typedef int (*FLAC__IOCallback_Close)(FLAC__IOHandle handle);
typedef struct {
	FLAC__IOCallback_Read read;
	FLAC__IOCallback_Write write;
	FLAC__IOCallback_Seek seek;
	FLAC__IOCallback_Tell tell;
	FLAC__IOCallback_Eof eof;
	FLAC__IOCallback_Close close;
} FLAC__IOCallbacks;
/* metadata.h:299-354 */
struct FLAC__Metadata_SimpleIterator;
typedef struct FLAC__Metadata_SimpleIterator FLAC__Metadata_SimpleIterator;
typedef enum {
	FLAC__METADATA_SIMPLE_ITERATOR_STATUS_OK = 0,
	FLAC__METADATA_SIMPLE_ITERATOR_STATUS_ILLEGAL_INPUT,
	FLAC__METADATA_SIMPLE_ITERATOR_STATUS_ERROR_OPENING_FILE,
	FLAC__METADATA_SIMPLE_ITERATOR_STATUS_NOT_A_FLAC_FILE,
	FLAC__METADATA_SIMPLE_ITERATOR_STATUS_NOT_WRITABLE,
	FLAC__METADATA_SIMPLE_ITERATOR_STATUS_BAD_METADATA,
	FLAC__METADATA_SIMPLE_ITERATOR_STATUS_READ_ERROR,
	FLAC__METADATA_SIMPLE_ITERATOR_STATUS_SEEK_ERROR,
	FLAC__METADATA_SIMPLE_ITERATOR_STATUS_WRITE_ERROR,
	FLAC__METADATA_SIMPLE_ITERATOR_STATUS_RENAME_ERROR,
	FLAC__METADATA_SIMPLE_ITERATOR_STATUS_UNLINK_ERROR,
	FLAC__METADATA_SIMPLE_ITERATOR_STATUS_MEMORY_ALLOCATION_ERROR,
	FLAC__METADATA_SIMPLE_ITERATOR_STATUS_INTERNAL_ERROR
} FLAC__Metadata_SimpleIteratorStatus;
extern const char *const FLAC__Metadata_SimpleIteratorStatusString[];
/* metadata.h:364-648 */
FLAC__Metadata_SimpleIterator *FLAC__metadata_simple_iterator_new(void);
void FLAC__metadata_simple_iterator_delete(FLAC__Metadata_SimpleIterator *iterator);
FLAC__Metadata_SimpleIteratorStatus FLAC__metadata_simple_iterator_status(FLAC__Metadata_SimpleIterator *iterator);
FLAC__bool FLAC__metadata_simple_iterator_init(FLAC__Metadata_SimpleIterator *iterator, const char *filename, FLAC__bool read_only, FLAC__bool preserve_file_stats);
FLAC__bool FLAC__metadata_simple_iterator_is_writable(const FLAC__Metadata_SimpleIterator *iterator);
FLAC__bool FLAC__metadata_simple_iterator_next(FLAC__Metadata_SimpleIterator *iterator);
FLAC__bool FLAC__metadata_simple_iterator_prev(FLAC__Metadata_SimpleIterator *iterator);
FLAC__bool FLAC__metadata_simple_iterator_is_last(const FLAC__Metadata_SimpleIterator *iterator);
off_t FLAC__metadata_simple_iterator_get_block_offset(const FLAC__Metadata_SimpleIterator *iterator);
FLAC__MetadataType FLAC__metadata_simple_iterator_get_block_type(const FLAC__Metadata_SimpleIterator *iterator);
unsigned FLAC__metadata_simple_iterator_get_block_length(const FLAC__Metadata_SimpleIterator *iterator);
FLAC__bool FLAC__metadata_simple_iterator_get_application_id(FLAC__Metadata_SimpleIterator *iterator, FLAC__byte *id);
FLAC__StreamMetadata *FLAC__metadata_simple_iterator_get_block(FLAC__Metadata_SimpleIterator *iterator);
FLAC__bool FLAC__metadata_simple_iterator_set_block(FLAC__Metadata_SimpleIterator *iterator, FLAC__StreamMetadata *block, FLAC__bool use_padding);
FLAC__bool FLAC__metadata_simple_iterator_insert_block_after(FLAC__Metadata_SimpleIterator *iterator, FLAC__StreamMetadata *block, FLAC__bool use_padding);
FLAC__bool FLAC__metadata_simple_iterator_delete_block(FLAC__Metadata_SimpleIterator *iterator, FLAC__bool use_padding);
/* metadata.h:703-780 */
struct FLAC__Metadata_Chain;
typedef struct FLAC__Metadata_Chain FLAC__Metadata_Chain;
struct FLAC__Metadata_Iterator;
typedef struct FLAC__Metadata_Iterator FLAC__Metadata_Iterator;
typedef enum {
	FLAC__METADATA_CHAIN_STATUS_OK = 0,
	FLAC__METADATA_CHAIN_STATUS_ILLEGAL_INPUT,
	FLAC__METADATA_CHAIN_STATUS_ERROR_OPENING_FILE,
	FLAC__METADATA_CHAIN_STATUS_NOT_A_FLAC_FILE,
	FLAC__METADATA_CHAIN_STATUS_NOT_WRITABLE,
	FLAC__METADATA_CHAIN_STATUS_BAD_METADATA,
	FLAC__METADATA_CHAIN_STATUS_READ_ERROR,
	FLAC__METADATA_CHAIN_STATUS_SEEK_ERROR,
	FLAC__METADATA_CHAIN_STATUS_WRITE_ERROR,
	FLAC__METADATA_CHAIN_STATUS_RENAME_ERROR,
	FLAC__METADATA_CHAIN_STATUS_UNLINK_ERROR,
	FLAC__METADATA_CHAIN_STATUS_MEMORY_ALLOCATION_ERROR,
	FLAC__METADATA_CHAIN_STATUS_INTERNAL_ERROR,
	FLAC__METADATA_CHAIN_STATUS_INVALID_CALLBACKS,
	FLAC__METADATA_CHAIN_STATUS_READ_WRITE_MISMATCH,
	FLAC__METADATA_CHAIN_STATUS_WRONG_WRITE_CALL
} FLAC__Metadata_ChainStatus;
extern const char *const FLAC__Metadata_ChainStatusString[];
/* metadata.h:798-1242; the two *_read_ogg* calls fail with ERROR_OPENING_FILE / INTERNAL_ERROR like a reference built
 * without libogg */
FLAC__Metadata_Chain *FLAC__metadata_chain_new(void);
void FLAC__metadata_chain_delete(FLAC__Metadata_Chain *chain);
FLAC__Metadata_ChainStatus FLAC__metadata_chain_status(FLAC__Metadata_Chain *chain);
FLAC__bool FLAC__metadata_chain_read(FLAC__Metadata_Chain *chain, const char *filename);
FLAC__bool FLAC__metadata_chain_read_ogg(FLAC__Metadata_Chain *chain, const char *filename);
FLAC__bool FLAC__metadata_chain_read_with_callbacks(FLAC__Metadata_Chain *chain, FLAC__IOHandle handle, FLAC__IOCallbacks callbacks);
FLAC__bool FLAC__metadata_chain_read_ogg_with_callbacks(FLAC__Metadata_Chain *chain, FLAC__IOHandle handle, FLAC__IOCallbacks callbacks);
FLAC__bool FLAC__metadata_chain_check_if_tempfile_needed(FLAC__Metadata_Chain *chain, FLAC__bool use_padding);
FLAC__bool FLAC__metadata_chain_write(FLAC__Metadata_Chain *chain, FLAC__bool use_padding, FLAC__bool preserve_file_stats);
FLAC__bool FLAC__metadata_chain_write_with_callbacks(FLAC__Metadata_Chain *chain, FLAC__bool use_padding, FLAC__IOHandle handle, FLAC__IOCallbacks callbacks);
FLAC__bool FLAC__metadata_chain_write_with_callbacks_and_tempfile(FLAC__Metadata_Chain *chain, FLAC__bool use_padding, FLAC__IOHandle handle, FLAC__IOCallbacks callbacks, FLAC__IOHandle temp_handle, FLAC__IOCallbacks temp_callbacks);
void FLAC__metadata_chain_merge_padding(FLAC__Metadata_Chain *chain);
void FLAC__metadata_chain_sort_padding(FLAC__Metadata_Chain *chain);
FLAC__Metadata_Iterator *FLAC__metadata_iterator_new(void);
void FLAC__metadata_iterator_delete(FLAC__Metadata_Iterator *iterator);
void FLAC__metadata_iterator_init(FLAC__Metadata_Iterator *iterator, FLAC__Metadata_Chain *chain);
FLAC__bool FLAC__metadata_iterator_next(FLAC__Metadata_Iterator *iterator);
FLAC__bool FLAC__metadata_iterator_prev(FLAC__Metadata_Iterator *iterator);
FLAC__MetadataType FLAC__metadata_iterator_get_block_type(const FLAC__Metadata_Iterator *iterator);
FLAC__StreamMetadata *FLAC__metadata_iterator_get_block(FLAC__Metadata_Iterator *iterator);
FLAC__bool FLAC__metadata_iterator_set_block(FLAC__Metadata_Iterator *iterator, FLAC__StreamMetadata *block);
FLAC__bool FLAC__metadata_iterator_delete_block(FLAC__Metadata_Iterator *iterator, FLAC__bool replace_with_padding);
FLAC__bool FLAC__metadata_iterator_insert_block_before(FLAC__Metadata_Iterator *iterator, FLAC__StreamMetadata *block);
FLAC__bool FLAC__metadata_iterator_insert_block_after(FLAC__Metadata_Iterator *iterator, FLAC__StreamMetadata *block);

#ifdef __cplusplus
}
#endif
#endif
