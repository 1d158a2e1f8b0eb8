// libFLAC-compatible stream encoder on top of the device batch ABI.
//
// Mirrors src/libFLAC/stream_encoder.c: settings and their validation, the stream prologue, the
// one-sample look-ahead blocking of process / process_interleaved, per-frame write / tell / seek
// callback sequencing, seek-table fill-in and the finish()-time STREAMINFO rewrite.  Frames
// themselves come from flacamd_encode_frames_host in batches; callbacks run on the caller's thread.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "flac_amd.h"
#include "md5.h"

#include <thread>
#include "stream_common.h"

extern "C" {

extern const char *const FLAC__StreamEncoderStateString[] = {
	"FLAC__STREAM_ENCODER_OK", "FLAC__STREAM_ENCODER_UNINITIALIZED", "FLAC__STREAM_ENCODER_OGG_ERROR",
	"FLAC__STREAM_ENCODER_VERIFY_DECODER_ERROR", "FLAC__STREAM_ENCODER_VERIFY_MISMATCH_IN_AUDIO_DATA",
	"FLAC__STREAM_ENCODER_CLIENT_ERROR", "FLAC__STREAM_ENCODER_IO_ERROR", "FLAC__STREAM_ENCODER_FRAMING_ERROR",
	"FLAC__STREAM_ENCODER_MEMORY_ALLOCATION_ERROR"};
extern const char *const FLAC__StreamEncoderInitStatusString[] = {
	"FLAC__STREAM_ENCODER_INIT_STATUS_OK", "FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR",
	"FLAC__STREAM_ENCODER_INIT_STATUS_UNSUPPORTED_CONTAINER", "FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_CALLBACKS",
	"FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_NUMBER_OF_CHANNELS", "FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_BITS_PER_SAMPLE",
	"FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_SAMPLE_RATE", "FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_BLOCK_SIZE",
	"FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_MAX_LPC_ORDER", "FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_QLP_COEFF_PRECISION",
	"FLAC__STREAM_ENCODER_INIT_STATUS_BLOCK_SIZE_TOO_SMALL_FOR_LPC_ORDER", "FLAC__STREAM_ENCODER_INIT_STATUS_NOT_STREAMABLE",
	"FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_METADATA", "FLAC__STREAM_ENCODER_INIT_STATUS_ALREADY_INITIALIZED"};
extern const char *const FLAC__treamEncoderReadStatusString[] = {
	"FLAC__STREAM_ENCODER_READ_STATUS_CONTINUE", "FLAC__STREAM_ENCODER_READ_STATUS_END_OF_STREAM",
	"FLAC__STREAM_ENCODER_READ_STATUS_ABORT", "FLAC__STREAM_ENCODER_READ_STATUS_UNSUPPORTED"};
extern const char *const FLAC__StreamEncoderWriteStatusString[] = {"FLAC__STREAM_ENCODER_WRITE_STATUS_OK", "FLAC__STREAM_ENCODER_WRITE_STATUS_FATAL_ERROR"};
extern const char *const FLAC__StreamEncoderSeekStatusString[] = {
	"FLAC__STREAM_ENCODER_SEEK_STATUS_OK", "FLAC__STREAM_ENCODER_SEEK_STATUS_ERROR", "FLAC__STREAM_ENCODER_SEEK_STATUS_UNSUPPORTED"};
extern const char *const FLAC__StreamEncoderTellStatusString[] = {
	"FLAC__STREAM_ENCODER_TELL_STATUS_OK", "FLAC__STREAM_ENCODER_TELL_STATUS_ERROR", "FLAC__STREAM_ENCODER_TELL_STATUS_UNSUPPORTED"};

}  // extern "C"

struct ApodSpec {
	uint32_t type;
	float param;
};

// src/libFLAC/include/protected/stream_encoder.h:78-108
struct FLAC__StreamEncoderProtected {
	FLAC__StreamEncoderState state;
	FLAC__bool verify, streamable_subset, do_md5, do_mid_side_stereo, loose_mid_side_stereo;
	unsigned channels, bits_per_sample, sample_rate, blocksize;
	unsigned num_apodizations;
	ApodSpec apodizations[FLACAMD_MAX_APODIZATIONS];
	unsigned max_lpc_order, qlp_coeff_precision;
	FLAC__bool do_qlp_coeff_prec_search, do_exhaustive_model_search, do_escape_coding;
	unsigned min_residual_partition_order, max_residual_partition_order, rice_parameter_search_dist;
	FLAC__uint64 total_samples_estimate;
	FLAC__StreamMetadata **metadata;
	unsigned num_metadata_blocks;
	FLAC__uint64 streaminfo_offset, seektable_offset, audio_offset;
};

struct FLAC__StreamEncoderPrivate {
	FLAC__StreamEncoderWriteCallback write_callback;
	FLAC__StreamEncoderSeekCallback seek_callback;
	FLAC__StreamEncoderTellCallback tell_callback;
	FLAC__StreamEncoderMetadataCallback metadata_callback;
	FLAC__StreamEncoderProgressCallback progress_callback;
	void *client_data;
	FILE *file;
	FLAC__bool disable_constant_subframes, disable_fixed_subframes, disable_verbatim_subframes;
	FLAC__StreamMetadata streaminfo;
	FLAC__StreamMetadata_SeekTable *seek_table;
	unsigned first_seekpoint_to_check;
	FLAC__uint64 bytes_written, samples_written;
	unsigned frames_written, total_frames_estimate;
	unsigned current_frame_number;
	FaMd5 md5;
	// batching
	flacamd_encoder *dev;
	flacamd_decoder *verify_dec;
	unsigned batch_frames;
	unsigned configured_blocksize;
	std::vector<int32_t> *ring;    // interleaved samples not yet encoded
	size_t buffered;               // wide samples in ring
	std::vector<uint8_t> *out;
	std::vector<uint32_t> *frame_bytes;
	uint32_t loose_state;
	// verify error report (stream_encoder.h:1242)
	FLAC__uint64 ve_absolute_sample;
	unsigned ve_frame, ve_channel, ve_sample;
	FLAC__int32 ve_expected, ve_got;
	FLAC__bool is_being_deleted;
};

namespace {

// stream_encoder.c:120-141
const struct { FLAC__bool ms, loose; unsigned lpc; FLAC__bool exhaustive; unsigned maxpart; } kLevels[9] = {
	{0, 0, 0, 0, 3}, {1, 1, 0, 0, 3}, {1, 0, 0, 0, 3}, {0, 0, 6, 0, 4}, {1, 1, 8, 0, 4},
	{1, 0, 8, 0, 5}, {1, 0, 8, 0, 6}, {1, 0, 8, 1, 6}, {1, 0, 12, 1, 6}};

// set_defaults_ (stream_encoder.c:2108).  The reference ends this function with
// set_compression_level(5), but both of its call sites run while state != UNINITIALIZED
// (:557 before :589, :1365 before :1368), so that call is always refused: what a fresh or
// finished encoder really holds is what is written here.
void set_defaults(FLAC__StreamEncoder *e)
{
	FLAC__StreamEncoderProtected *p = e->protected_;
	p->verify = 0;
	p->streamable_subset = 1;
	p->do_md5 = 1;
	p->do_mid_side_stereo = 0;
	p->loose_mid_side_stereo = 0;
	p->channels = 2;
	p->bits_per_sample = 16;
	p->sample_rate = 44100;
	p->blocksize = 0;
	p->num_apodizations = 1;
	p->apodizations[0].type = FLACAMD_APOD_TUKEY;
	p->apodizations[0].param = 0.5f;
	p->max_lpc_order = 0;
	p->qlp_coeff_precision = 0;
	p->do_qlp_coeff_prec_search = 0;
	p->do_exhaustive_model_search = 0;
	p->do_escape_coding = 0;
	p->min_residual_partition_order = 0;
	p->max_residual_partition_order = 0;
	p->rice_parameter_search_dist = 0;
	p->total_samples_estimate = 0;
	p->metadata = 0;
	p->num_metadata_blocks = 0;
	FLAC__StreamEncoderPrivate *q = e->private_;
	q->seek_table = 0;
	q->disable_constant_subframes = q->disable_fixed_subframes = q->disable_verbatim_subframes = 0;
	q->write_callback = 0;
	q->seek_callback = 0;
	q->tell_callback = 0;
	q->metadata_callback = 0;
	q->progress_callback = 0;
	q->client_data = 0;
}

void free_run_state(FLAC__StreamEncoder *e)
{
	FLAC__StreamEncoderPrivate *q = e->private_;
	if (e->protected_->metadata) {
		free(e->protected_->metadata);
		e->protected_->metadata = 0;
		e->protected_->num_metadata_blocks = 0;
	}
	if (q->dev) {
		flacamd_encoder_destroy(q->dev);
		q->dev = 0;
	}
	if (q->verify_dec) {
		flacamd_decoder_destroy(q->verify_dec);
		q->verify_dec = 0;
	}
	q->ring->clear();
	q->ring->shrink_to_fit();
	q->buffered = 0;
	q->out->clear();
	q->out->shrink_to_fit();
	q->frame_bytes->clear();
}

#define UNINIT_OR_FALSE(e) \
	do { \
		if ((e)->protected_->state != FLAC__STREAM_ENCODER_UNINITIALIZED) \
			return 0; \
	} while (0)

// write_frame_ (stream_encoder.c:2426): one buffer through tell + seek-point bookkeeping + write
FLAC__StreamEncoderWriteStatus write_frame(FLAC__StreamEncoder *e, const FLAC__byte *buffer, size_t bytes, unsigned samples)
{
	FLAC__StreamEncoderProtected *p = e->protected_;
	FLAC__StreamEncoderPrivate *q = e->private_;
	FLAC__uint64 output_position = 0;
	if (q->tell_callback && q->tell_callback(e, &output_position, q->client_data) == FLAC__STREAM_ENCODER_TELL_STATUS_ERROR) {
		p->state = FLAC__STREAM_ENCODER_CLIENT_ERROR;
		return FLAC__STREAM_ENCODER_WRITE_STATUS_FATAL_ERROR;
	}
	if (samples == 0) {
		const unsigned type = buffer[0] & 0x7f;
		if (type == FLAC__METADATA_TYPE_STREAMINFO)
			p->streaminfo_offset = output_position;
		else if (type == FLAC__METADATA_TYPE_SEEKTABLE && p->seektable_offset == 0)
			p->seektable_offset = output_position;
	}
	if (q->seek_table && p->audio_offset > 0 && q->seek_table->num_points > 0) {
		const unsigned blocksize = p->blocksize;
		const FLAC__uint64 first = q->samples_written, last = first + blocksize - 1;
		for (unsigned i = q->first_seekpoint_to_check; i < q->seek_table->num_points; i++) {
			const FLAC__uint64 t = q->seek_table->points[i].sample_number;
			if (t > last)
				break;
			if (t >= first) {
				q->seek_table->points[i].sample_number = first;
				q->seek_table->points[i].stream_offset = output_position - p->audio_offset;
				q->seek_table->points[i].frame_samples = blocksize;
			}
			q->first_seekpoint_to_check++;
		}
	}
	const FLAC__StreamEncoderWriteStatus st = q->write_callback(e, buffer, bytes, samples, q->current_frame_number, q->client_data);
	if (st == FLAC__STREAM_ENCODER_WRITE_STATUS_OK) {
		q->bytes_written += bytes;
		q->samples_written += samples;
		if (q->frames_written < q->current_frame_number + 1)
			q->frames_written = q->current_frame_number + 1;
	} else
		p->state = FLAC__STREAM_ENCODER_CLIENT_ERROR;
	return st;
}

// write_bitbuffer_ (stream_encoder.c:2379) for a metadata buffer
bool write_metadata_buffer(FLAC__StreamEncoder *e, const std::vector<uint8_t> &buf)
{
	if (write_frame(e, buf.data(), buf.size(), 0) != FLAC__STREAM_ENCODER_WRITE_STATUS_OK) {
		e->protected_->state = FLAC__STREAM_ENCODER_CLIENT_ERROR;
		return false;
	}
	return true;
}

// Encode `wide` samples from the front of the ring (whole blocks, or the final short block) and
// hand the frames to the client in order: process_frame_ (stream_encoder.c:2861) + write_bitbuffer_.
bool encode_and_emit(FLAC__StreamEncoder *e, size_t wide, bool is_last)
{
	FLAC__StreamEncoderProtected *p = e->protected_;
	FLAC__StreamEncoderPrivate *q = e->private_;
	const unsigned bs = q->configured_blocksize, ch = p->channels;
	const int32_t *pcm = q->ring->data();
	const uint32_t nfr_max = (uint32_t)((wide + bs - 1) / bs);
	q->out->resize(flacamd_encoder_max_output_bytes(q->dev, nfr_max));
	q->frame_bytes->resize(nfr_max);
	uint32_t nf = 0;
	size_t total = 0;
	// The MD5 of these samples (FLAC__MD5Accumulate, process_frame_ stream_encoder.c:2870) is serial
	// host work; it runs on a helper thread while the GPU codes the same samples.  Every sample
	// passes here exactly once and in order, so the digest is the one the reference computes.
	std::thread md5_helper;
	if (p->do_md5)
		md5_helper = std::thread([q, pcm, wide, ch, p]() {
			fa_md5_accumulate_interleaved(&q->md5, pcm, wide * ch, (p->bits_per_sample + 7) / 8);
		});
	int rc = flacamd_encode_frames_host(q->dev, pcm, wide, q->current_frame_number, is_last ? 1 : 0, &q->loose_state, q->out->data(),
	                                    q->out->size(), q->frame_bytes->data(), &nf, &total);
	if (md5_helper.joinable())
		md5_helper.join();
	if (rc != FLACAMD_OK) {
		p->state = rc == FLACAMD_E_NOMEM ? FLAC__STREAM_ENCODER_MEMORY_ALLOCATION_ERROR : FLAC__STREAM_ENCODER_FRAMING_ERROR;
		return false;
	}
	if (p->verify) {
		// verify_write_callback_ (stream_encoder.c:4161): decode what was just produced, compare
		std::vector<int32_t> back(wide * ch);
		flacamd_stream_desc sd;
		sd.byte_offset = 0;
		sd.byte_length = total;
		sd.sample_offset = 0;
		sd.total_samples = wide;
		sd.first_sample = (FLAC__uint64)q->current_frame_number * bs;
		uint32_t dfr = 0, bad = 0;
		uint64_t dwide = 0;
		rc = flacamd_decode_streams_host(q->verify_dec, q->out->data(), total, &sd, 1, back.data(), 4, wide, &dfr, &dwide, &bad);
		if (rc != FLACAMD_OK || bad || dfr != nf) {
			p->state = FLAC__STREAM_ENCODER_VERIFY_DECODER_ERROR;
			return false;
		}
		if (memcmp(back.data(), pcm, sizeof(int32_t) * wide * ch) != 0) {
			for (size_t i = 0; i < wide * ch; i++)
				if (back[i] != pcm[i]) {
					const size_t w = i / ch;
					q->ve_absolute_sample = (FLAC__uint64)q->current_frame_number * bs + w;
					q->ve_frame = q->current_frame_number + (unsigned)(w / bs);
					q->ve_channel = (unsigned)(i % ch);
					q->ve_sample = (unsigned)(w % bs);
					q->ve_expected = pcm[i];
					q->ve_got = back[i];
					break;
				}
			p->state = FLAC__STREAM_ENCODER_VERIFY_MISMATCH_IN_AUDIO_DATA;
			return false;
		}
	}
	size_t off = 0, done = 0;
	for (uint32_t f = 0; f < nf; f++) {
		const unsigned n = (unsigned)((wide - done) < bs ? (wide - done) : bs);
		const size_t fb = (*q->frame_bytes)[f];
		p->blocksize = n;  // finish() sets the short block's size before its frame is written (:1319)
		if (write_frame(e, q->out->data() + off, fb, n) != FLAC__STREAM_ENCODER_WRITE_STATUS_OK) {
			p->state = FLAC__STREAM_ENCODER_CLIENT_ERROR;
			return false;
		}
		// write_bitbuffer_ :2418-2421, process_frame_ :2913-2915
		FLAC__StreamMetadata_StreamInfo &si = q->streaminfo.data.stream_info;
		if (fb < si.min_framesize) si.min_framesize = (unsigned)fb;
		if (fb > si.max_framesize) si.max_framesize = (unsigned)fb;
		q->current_frame_number++;
		si.total_samples += n;
		off += fb;
		done += n;
	}
	p->blocksize = bs;
	// drop the consumed samples
	const size_t keep = q->buffered - wide;
	if (keep)
		memmove(q->ring->data(), q->ring->data() + wide * ch, sizeof(int32_t) * keep * ch);
	q->buffered = keep;
	return true;
}

// The reference encodes a block once blocksize+1 samples are buffered (OVERREAD_, :504-515), so
// the last block is always left for finish().  Batched: encode every whole block that is followed
// by at least one more sample, once a full batch of them has accumulated (or when forced).
bool drain_ring(FLAC__StreamEncoder *e, bool force)
{
	FLAC__StreamEncoderPrivate *q = e->private_;
	const unsigned bs = q->configured_blocksize;
	for (;;) {
		if (q->buffered <= bs)
			return true;
		size_t ready = (q->buffered - 1) / bs;
		if (ready > q->batch_frames)
			ready = q->batch_frames;
		if (ready < q->batch_frames && !force)
			return true;
		if (!encode_and_emit(e, ready * bs, false))
			return false;
	}
}


// update_metadata_ (stream_encoder.c:2516): patch STREAMINFO and the seek table in place
void update_metadata(FLAC__StreamEncoder *e)
{
	FLAC__StreamEncoderProtected *p = e->protected_;
	FLAC__StreamEncoderPrivate *q = e->private_;
	const FLAC__StreamMetadata_StreamInfo &si = q->streaminfo.data.stream_info;
	FLAC__byte b[18];
	auto seek_to = [&](FLAC__uint64 off) {
		const FLAC__StreamEncoderSeekStatus s = q->seek_callback(e, off, q->client_data);
		if (s != FLAC__STREAM_ENCODER_SEEK_STATUS_OK) {
			if (s == FLAC__STREAM_ENCODER_SEEK_STATUS_ERROR)
				p->state = FLAC__STREAM_ENCODER_CLIENT_ERROR;
			return false;
		}
		return true;
	};
	auto put = [&](const FLAC__byte *buf, size_t n) {
		if (q->write_callback(e, buf, n, 0, 0, q->client_data) != FLAC__STREAM_ENCODER_WRITE_STATUS_OK) {
			p->state = FLAC__STREAM_ENCODER_CLIENT_ERROR;
			return false;
		}
		return true;
	};
	// MD5 at header(4) + 18
	if (!seek_to(p->streaminfo_offset + 4 + 18) || !put(si.md5sum, 16))
		return;
	// total samples: the byte that also holds the low 4 bits of bps-1, at header(4) + 13
	b[0] = (FLAC__byte)(((si.bits_per_sample - 1) << 4) | ((si.total_samples >> 32) & 0x0f));
	b[1] = (FLAC__byte)(si.total_samples >> 24);
	b[2] = (FLAC__byte)(si.total_samples >> 16);
	b[3] = (FLAC__byte)(si.total_samples >> 8);
	b[4] = (FLAC__byte)si.total_samples;
	if (!seek_to(p->streaminfo_offset + 4 + 13) || !put(b, 5))
		return;
	// min / max frame size at header(4) + 4
	b[0] = (FLAC__byte)(si.min_framesize >> 16);
	b[1] = (FLAC__byte)(si.min_framesize >> 8);
	b[2] = (FLAC__byte)si.min_framesize;
	b[3] = (FLAC__byte)(si.max_framesize >> 16);
	b[4] = (FLAC__byte)(si.max_framesize >> 8);
	b[5] = (FLAC__byte)si.max_framesize;
	if (!seek_to(p->streaminfo_offset + 4 + 4) || !put(b, 6))
		return;
	if (q->seek_table && q->seek_table->num_points > 0 && p->seektable_offset > 0) {
		FLAC__format_seektable_sort(q->seek_table);
		if (!seek_to(p->seektable_offset + 4))
			return;
		for (unsigned i = 0; i < q->seek_table->num_points; i++) {
			const FLAC__StreamMetadata_SeekPoint &sp = q->seek_table->points[i];
			for (int k = 0; k < 8; k++) {
				b[k] = (FLAC__byte)(sp.sample_number >> (56 - 8 * k));
				b[8 + k] = (FLAC__byte)(sp.stream_offset >> (56 - 8 * k));
			}
			b[16] = (FLAC__byte)(sp.frame_samples >> 8);
			b[17] = (FLAC__byte)sp.frame_samples;
			if (!put(b, 18))
				return;
		}
	}
}

// callbacks of the FILE flavours (stream_encoder.c:4261-4329)
FLAC__StreamEncoderSeekStatus file_seek(const FLAC__StreamEncoder *e, FLAC__uint64 off, void *)
{
	return fseeko(e->private_->file, (off_t)off, SEEK_SET) < 0 ? FLAC__STREAM_ENCODER_SEEK_STATUS_ERROR : FLAC__STREAM_ENCODER_SEEK_STATUS_OK;
}

FLAC__StreamEncoderTellStatus file_tell(const FLAC__StreamEncoder *e, FLAC__uint64 *off, void *)
{
	const off_t o = ftello(e->private_->file);
	if (o < 0)
		return FLAC__STREAM_ENCODER_TELL_STATUS_ERROR;
	*off = (FLAC__uint64)o;
	return FLAC__STREAM_ENCODER_TELL_STATUS_OK;
}

FLAC__StreamEncoderWriteStatus file_write(const FLAC__StreamEncoder *e, const FLAC__byte buffer[], size_t bytes, unsigned samples, unsigned, void *)
{
	FLAC__StreamEncoderPrivate *q = e->private_;
	if (fwrite(buffer, 1, bytes, q->file) != bytes)
		return FLAC__STREAM_ENCODER_WRITE_STATUS_FATAL_ERROR;
	if (q->progress_callback && samples > 0)
		q->progress_callback(e, q->bytes_written + bytes, q->samples_written + samples, q->frames_written + 1, q->total_frames_estimate, q->client_data);
	return FLAC__STREAM_ENCODER_WRITE_STATUS_OK;
}

// init_stream_internal_ (stream_encoder.c:633)
FLAC__StreamEncoderInitStatus init_stream_internal(FLAC__StreamEncoder *e, FLAC__StreamEncoderWriteCallback write_callback,
                                                   FLAC__StreamEncoderSeekCallback seek_callback, FLAC__StreamEncoderTellCallback tell_callback,
                                                   FLAC__StreamEncoderMetadataCallback metadata_callback, void *client_data, bool is_ogg)
{
	FLAC__StreamEncoderProtected *p = e->protected_;
	FLAC__StreamEncoderPrivate *q = e->private_;
	if (p->state != FLAC__STREAM_ENCODER_UNINITIALIZED)
		return FLAC__STREAM_ENCODER_INIT_STATUS_ALREADY_INITIALIZED;
	if (is_ogg)
		return FLAC__STREAM_ENCODER_INIT_STATUS_UNSUPPORTED_CONTAINER;  // as a reference build without libogg (:652-655)
	if (!write_callback || (seek_callback && !tell_callback))
		return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_CALLBACKS;
	if (p->channels == 0 || p->channels > FLAC__MAX_CHANNELS)
		return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_NUMBER_OF_CHANNELS;
	if (p->channels != 2) {
		p->do_mid_side_stereo = 0;
		p->loose_mid_side_stereo = 0;
	} else if (!p->do_mid_side_stereo)
		p->loose_mid_side_stereo = 0;
	if (p->bits_per_sample >= 32)
		p->do_mid_side_stereo = 0;
	if (p->bits_per_sample < FLAC__MIN_BITS_PER_SAMPLE || p->bits_per_sample > FLAC__REFERENCE_CODEC_MAX_BITS_PER_SAMPLE)
		return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_BITS_PER_SAMPLE;
	if (!FLAC__format_sample_rate_is_valid(p->sample_rate))
		return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_SAMPLE_RATE;
	if (p->blocksize == 0)
		p->blocksize = p->max_lpc_order == 0 ? 1152 : 4096;
	if (p->blocksize < FLAC__MIN_BLOCK_SIZE || p->blocksize > FLAC__MAX_BLOCK_SIZE)
		return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_BLOCK_SIZE;
	if (p->max_lpc_order > FLAC__MAX_LPC_ORDER)
		return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_MAX_LPC_ORDER;
	if (p->blocksize < p->max_lpc_order)
		return FLAC__STREAM_ENCODER_INIT_STATUS_BLOCK_SIZE_TOO_SMALL_FOR_LPC_ORDER;
	if (p->qlp_coeff_precision != 0 && (p->qlp_coeff_precision < FLAC__MIN_QLP_COEFF_PRECISION || p->qlp_coeff_precision > FLAC__MAX_QLP_COEFF_PRECISION))
		return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_QLP_COEFF_PRECISION;

	// the remaining defaulting (:695-759) is shared with the batch ABI
	flacamd_encode_params ep;
	memset(&ep, 0, sizeof ep);
	ep.channels = p->channels;
	ep.bits_per_sample = p->bits_per_sample;
	ep.sample_rate = p->sample_rate;
	ep.blocksize = p->blocksize;
	ep.do_mid_side_stereo = p->do_mid_side_stereo;
	ep.loose_mid_side_stereo = p->loose_mid_side_stereo;
	ep.max_lpc_order = p->max_lpc_order;
	ep.qlp_coeff_precision = p->qlp_coeff_precision;
	ep.do_qlp_coeff_prec_search = p->do_qlp_coeff_prec_search;
	ep.do_exhaustive_model_search = p->do_exhaustive_model_search;
	ep.min_residual_partition_order = p->min_residual_partition_order;
	ep.max_residual_partition_order = p->max_residual_partition_order;
	ep.disable_constant_subframes = q->disable_constant_subframes;
	ep.disable_fixed_subframes = q->disable_fixed_subframes;
	ep.disable_verbatim_subframes = q->disable_verbatim_subframes;
	ep.num_apodizations = p->num_apodizations;
	for (unsigned i = 0; i < p->num_apodizations; i++) {
		ep.apod_type[i] = p->apodizations[i].type;
		ep.apod_param[i] = p->apodizations[i].param;
	}
	if (flacamd_params_resolve(&ep) != FLACAMD_OK)
		return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR;
	p->qlp_coeff_precision = ep.qlp_coeff_precision;

	if (p->streamable_subset) {  // :730-754
		if (!FLAC__format_blocksize_is_subset(p->blocksize, p->sample_rate) || !FLAC__format_sample_rate_is_subset(p->sample_rate))
			return FLAC__STREAM_ENCODER_INIT_STATUS_NOT_STREAMABLE;
		const unsigned b = p->bits_per_sample;
		if (b != 8 && b != 12 && b != 16 && b != 20 && b != 24)
			return FLAC__STREAM_ENCODER_INIT_STATUS_NOT_STREAMABLE;
		if (p->max_residual_partition_order > FLAC__SUBSET_MAX_RICE_PARTITION_ORDER)
			return FLAC__STREAM_ENCODER_INIT_STATUS_NOT_STREAMABLE;
		if (p->sample_rate <= 48000 && (p->blocksize > 4608 || p->max_lpc_order > 12))
			return FLAC__STREAM_ENCODER_INIT_STATUS_NOT_STREAMABLE;
	}
	p->max_residual_partition_order = ep.max_residual_partition_order;
	p->min_residual_partition_order = ep.min_residual_partition_order;

	// metadata: first SEEKTABLE is the one kept up to date; validation (:776-838)
	if (p->metadata && p->num_metadata_blocks > 0)
		for (unsigned i = 0; i < p->num_metadata_blocks; i++)
			if (p->metadata[i] && p->metadata[i]->type == FLAC__METADATA_TYPE_SEEKTABLE) {
				q->seek_table = &p->metadata[i]->data.seek_table;
				break;
			}
	if (!p->metadata && p->num_metadata_blocks > 0)
		return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_METADATA;
	bool has_seektable = false, has_vc = false, pic1 = false, pic2 = false;
	for (unsigned i = 0; i < p->num_metadata_blocks; i++) {
		const FLAC__StreamMetadata *m = p->metadata[i];
		if (m->type == FLAC__METADATA_TYPE_STREAMINFO)
			return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_METADATA;
		if (m->type == FLAC__METADATA_TYPE_SEEKTABLE) {
			if (has_seektable || !FLAC__format_seektable_is_legal(&m->data.seek_table))
				return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_METADATA;
			has_seektable = true;
		} else if (m->type == FLAC__METADATA_TYPE_VORBIS_COMMENT) {
			if (has_vc)
				return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_METADATA;
			has_vc = true;
		} else if (m->type == FLAC__METADATA_TYPE_PICTURE) {
			if (m->data.picture.type == 1) {
				if (pic1)
					return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_METADATA;
				pic1 = true;
				if ((strcmp(m->data.picture.mime_type, "image/png") && strcmp(m->data.picture.mime_type, "-->")) || m->data.picture.width != 32 ||
				    m->data.picture.height != 32)
					return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_METADATA;
			} else if (m->data.picture.type == 2) {
				if (pic2)
					return FLAC__STREAM_ENCODER_INIT_STATUS_INVALID_METADATA;
				pic2 = true;
			}
		}
	}

	p->state = FLAC__STREAM_ENCODER_OK;  // from here on errors are fatal (:947)
	q->write_callback = write_callback;
	q->seek_callback = seek_callback;
	q->tell_callback = tell_callback;
	q->metadata_callback = metadata_callback;
	q->client_data = client_data;
	q->current_frame_number = 0;
	q->first_seekpoint_to_check = 0;
	q->samples_written = 0;
	q->loose_state = 0;
	q->buffered = 0;
	q->configured_blocksize = p->blocksize;
	p->streaminfo_offset = p->seektable_offset = p->audio_offset = 0;
	q->ve_absolute_sample = 0;
	q->ve_frame = q->ve_channel = q->ve_sample = 0;
	q->ve_expected = q->ve_got = 0;

	// resize_buffers_ (:2245): device buffers for one batch
	{
		const char *env = getenv("FLAC_AMD_BATCH_FRAMES");
		size_t bf = env ? (size_t)strtoul(env, 0, 10) : 0;
		if (bf == 0) {
			bf = ((size_t)32 << 20) / ((size_t)p->blocksize * p->channels);  // about 32 M samples in flight
			if (bf > 2048) bf = 2048;
		}
		if (bf < 1) bf = 1;
		q->batch_frames = (unsigned)bf;
	}
	int rc = flacamd_encoder_create(&ep, fa_stream_device(), q->batch_frames + 1, &q->dev);
	if (rc != FLACAMD_OK) {
		p->state = FLAC__STREAM_ENCODER_MEMORY_ALLOCATION_ERROR;
		return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR;
	}
	if (p->verify) {
		const uint64_t max_in = flacamd_encoder_max_output_bytes(q->dev, q->batch_frames + 1);
		rc = flacamd_decoder_create(p->channels, p->bits_per_sample, p->blocksize, fa_stream_device(), max_in, q->batch_frames + 8, &q->verify_dec);
		if (rc != FLACAMD_OK) {
			p->state = FLAC__STREAM_ENCODER_VERIFY_DECODER_ERROR;
			return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR;
		}
	}
	q->ring->reserve(((size_t)q->batch_frames + 2) * p->blocksize * p->channels);

	// stream prologue (:1029-1116): "fLaC", STREAMINFO, VORBIS_COMMENT if the client gave none, client blocks
	std::vector<uint8_t> buf;
	buf.assign(FLAC__STREAM_SYNC_STRING, FLAC__STREAM_SYNC_STRING + 4);
	if (!write_metadata_buffer(e, buf))
		return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR;
	q->streaminfo.type = FLAC__METADATA_TYPE_STREAMINFO;
	q->streaminfo.is_last = 0;
	q->streaminfo.length = FLAC__STREAM_METADATA_STREAMINFO_LENGTH;
	FLAC__StreamMetadata_StreamInfo &si = q->streaminfo.data.stream_info;
	si.min_blocksize = si.max_blocksize = p->blocksize;
	si.min_framesize = si.max_framesize = 0;
	si.sample_rate = p->sample_rate;
	si.channels = p->channels;
	si.bits_per_sample = p->bits_per_sample;
	si.total_samples = p->total_samples_estimate;
	memset(si.md5sum, 0, 16);
	if (p->do_md5)
		fa_md5_init(&q->md5, fa_md5_default_mode());
	buf.clear();
	fa_serialize_metadata(&q->streaminfo, buf);
	if (!write_metadata_buffer(e, buf))
		return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR;
	si.min_framesize = (1u << 24) - 1;
	si.total_samples = 0;
	if (!has_vc) {
		FLAC__StreamMetadata vc;
		memset(&vc, 0, sizeof vc);
		vc.type = FLAC__METADATA_TYPE_VORBIS_COMMENT;
		vc.is_last = p->num_metadata_blocks == 0;
		vc.length = 4 + 4;
		buf.clear();
		fa_serialize_metadata(&vc, buf);
		if (!write_metadata_buffer(e, buf))
			return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR;
	}
	for (unsigned i = 0; i < p->num_metadata_blocks; i++) {
		p->metadata[i]->is_last = (i == p->num_metadata_blocks - 1);
		buf.clear();
		fa_serialize_metadata(p->metadata[i], buf);
		if (!write_metadata_buffer(e, buf))
			return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR;
	}
	if (q->tell_callback && q->tell_callback(e, &p->audio_offset, q->client_data) == FLAC__STREAM_ENCODER_TELL_STATUS_ERROR) {
		p->state = FLAC__STREAM_ENCODER_CLIENT_ERROR;
		return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR;
	}
	return FLAC__STREAM_ENCODER_INIT_STATUS_OK;
}

FLAC__StreamEncoderInitStatus init_FILE_internal(FLAC__StreamEncoder *e, FILE *file, FLAC__StreamEncoderProgressCallback progress, void *client_data, bool is_ogg)
{
	FLAC__StreamEncoderProtected *p = e->protected_;
	FLAC__StreamEncoderPrivate *q = e->private_;
	if (p->state != FLAC__STREAM_ENCODER_UNINITIALIZED)
		return FLAC__STREAM_ENCODER_INIT_STATUS_ALREADY_INITIALIZED;
	if (!file) {
		p->state = FLAC__STREAM_ENCODER_IO_ERROR;
		return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR;
	}
	q->file = file;
	q->progress_callback = progress;
	q->bytes_written = q->samples_written = 0;
	q->frames_written = 0;
	const bool is_stdout = file == stdout;
	const FLAC__StreamEncoderInitStatus st =
		init_stream_internal(e, file_write, is_stdout ? 0 : file_seek, is_stdout ? 0 : file_tell, 0, client_data, is_ogg);
	if (st != FLAC__STREAM_ENCODER_INIT_STATUS_OK)
		return st;
	q->total_frames_estimate = (unsigned)((p->total_samples_estimate + p->blocksize - 1) / p->blocksize);
	return st;
}

FLAC__StreamEncoderInitStatus init_file_internal(FLAC__StreamEncoder *e, const char *filename, FLAC__StreamEncoderProgressCallback progress, void *client_data, bool is_ogg)
{
	if (e->protected_->state != FLAC__STREAM_ENCODER_UNINITIALIZED)
		return FLAC__STREAM_ENCODER_INIT_STATUS_ALREADY_INITIALIZED;
	FILE *f = filename ? fopen(filename, "w+b") : stdout;
	if (!f) {
		e->protected_->state = FLAC__STREAM_ENCODER_IO_ERROR;
		return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR;
	}
	return init_FILE_internal(e, f, progress, client_data, is_ogg);
}

}  // namespace

extern "C" {

// ---- construction (stream_encoder.c:523-632) ----
FLAC__StreamEncoder *FLAC__stream_encoder_new(void)
{
	FLAC__StreamEncoder *e = (FLAC__StreamEncoder *)calloc(1, sizeof *e);
	if (!e)
		return 0;
	e->protected_ = (FLAC__StreamEncoderProtected *)calloc(1, sizeof *e->protected_);
	e->private_ = (FLAC__StreamEncoderPrivate *)calloc(1, sizeof *e->private_);
	if (!e->protected_ || !e->private_) {
		free(e->protected_);
		free(e->private_);
		free(e);
		return 0;
	}
	e->private_->ring = new (std::nothrow) std::vector<int32_t>();
	e->private_->out = new (std::nothrow) std::vector<uint8_t>();
	e->private_->frame_bytes = new (std::nothrow) std::vector<uint32_t>();
	set_defaults(e);
	e->protected_->state = FLAC__STREAM_ENCODER_UNINITIALIZED;
	return e;
}

void FLAC__stream_encoder_delete(FLAC__StreamEncoder *e)
{
	if (!e)
		return;
	e->private_->is_being_deleted = 1;
	(void)FLAC__stream_encoder_finish(e);
	delete e->private_->ring;
	delete e->private_->out;
	delete e->private_->frame_bytes;
	free(e->private_);
	free(e->protected_);
	free(e);
}

// ---- init ----
FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_stream(FLAC__StreamEncoder *e, FLAC__StreamEncoderWriteCallback w, FLAC__StreamEncoderSeekCallback s,
                                                               FLAC__StreamEncoderTellCallback t, FLAC__StreamEncoderMetadataCallback m, void *cd)
{
	return init_stream_internal(e, w, s, t, m, cd, false);
}
FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_ogg_stream(FLAC__StreamEncoder *e, FLAC__StreamEncoderReadCallback, FLAC__StreamEncoderWriteCallback w,
                                                                   FLAC__StreamEncoderSeekCallback s, FLAC__StreamEncoderTellCallback t,
                                                                   FLAC__StreamEncoderMetadataCallback m, void *cd)
{
	return init_stream_internal(e, w, s, t, m, cd, true);
}
FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_FILE(FLAC__StreamEncoder *e, FILE *f, FLAC__StreamEncoderProgressCallback p, void *cd)
{
	return init_FILE_internal(e, f, p, cd, false);
}
FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_ogg_FILE(FLAC__StreamEncoder *e, FILE *f, FLAC__StreamEncoderProgressCallback p, void *cd)
{
	return init_FILE_internal(e, f, p, cd, true);
}
FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_file(FLAC__StreamEncoder *e, const char *fn, FLAC__StreamEncoderProgressCallback p, void *cd)
{
	return init_file_internal(e, fn, p, cd, false);
}
FLAC__StreamEncoderInitStatus FLAC__stream_encoder_init_ogg_file(FLAC__StreamEncoder *e, const char *fn, FLAC__StreamEncoderProgressCallback p, void *cd)
{
	return init_file_internal(e, fn, p, cd, true);
}

// ---- finish (stream_encoder.c:1305) ----
FLAC__bool FLAC__stream_encoder_finish(FLAC__StreamEncoder *e)
{
	FLAC__StreamEncoderProtected *p = e->protected_;
	FLAC__StreamEncoderPrivate *q = e->private_;
	bool error = false;
	if (p->state == FLAC__STREAM_ENCODER_UNINITIALIZED)
		return 1;
	if (p->state == FLAC__STREAM_ENCODER_OK && !q->is_being_deleted) {
		if (!drain_ring(e, true))
			error = true;
		else if (q->buffered != 0) {
			// the last block, short or not (:1317-1322)
			if (!encode_and_emit(e, q->buffered, true))
				error = true;
		}
	}
	if (p->do_md5 && q->dev)
		fa_md5_final(&q->md5, q->streaminfo.data.stream_info.md5sum);
	if (!q->is_being_deleted) {
		if (p->state == FLAC__STREAM_ENCODER_OK) {
			if (q->seek_callback) {
				update_metadata(e);
				if (p->state != FLAC__STREAM_ENCODER_OK)
					error = true;
			}
			if (q->metadata_callback)
				q->metadata_callback(e, &q->streaminfo, q->client_data);
		}
	}
	if (q->file) {
		if (q->file != stdout)
			fclose(q->file);
		q->file = 0;
	}
	free_run_state(e);
	set_defaults(e);
	if (!error)
		p->state = FLAC__STREAM_ENCODER_UNINITIALIZED;
	return !error;
}

// ---- process (stream_encoder.c:1968, 2020) ----
FLAC__bool FLAC__stream_encoder_process_interleaved(FLAC__StreamEncoder *e, const FLAC__int32 buffer[], unsigned samples)
{
	FLAC__StreamEncoderProtected *p = e->protected_;
	FLAC__StreamEncoderPrivate *q = e->private_;
	if (p->state != FLAC__STREAM_ENCODER_OK)
		return 0;
	const unsigned ch = p->channels;
	unsigned done = 0;
	while (done < samples) {
		// keep the ring bounded: top up to one batch plus the look-ahead, then drain
		const size_t room = (size_t)(q->batch_frames + 1) * q->configured_blocksize + 1 - q->buffered;
		const unsigned n = (unsigned)((samples - done) < room ? (samples - done) : room);
		q->ring->resize((q->buffered + n) * ch);
		memcpy(q->ring->data() + q->buffered * ch, buffer + (size_t)done * ch, sizeof(int32_t) * (size_t)n * ch);
		q->buffered += n;
		done += n;
		if (!drain_ring(e, false))
			return 0;
	}
	return 1;
}

FLAC__bool FLAC__stream_encoder_process(FLAC__StreamEncoder *e, const FLAC__int32 *const buffer[], unsigned samples)
{
	FLAC__StreamEncoderProtected *p = e->protected_;
	if (p->state != FLAC__STREAM_ENCODER_OK)
		return 0;
	const unsigned ch = p->channels;
	std::vector<int32_t> tmp;
	const unsigned step = 65536;
	for (unsigned done = 0; done < samples;) {
		const unsigned n = samples - done < step ? samples - done : step;
		tmp.resize((size_t)n * ch);
		for (unsigned c = 0; c < ch; c++)
			for (unsigned i = 0; i < n; i++)
				tmp[(size_t)i * ch + c] = buffer[c][done + i];
		if (!FLAC__stream_encoder_process_interleaved(e, tmp.data(), n))
			return 0;
		done += n;
	}
	return 1;
}

// ---- setters (stream_encoder.c:1373-1773) ----
FLAC__bool FLAC__stream_encoder_set_ogg_serial_number(FLAC__StreamEncoder *e, long) { UNINIT_OR_FALSE(e); return 0; /* no Ogg support, as :1384-1387 */ }
FLAC__bool FLAC__stream_encoder_set_verify(FLAC__StreamEncoder *e, FLAC__bool v) { UNINIT_OR_FALSE(e); e->protected_->verify = v; return 1; }
FLAC__bool FLAC__stream_encoder_set_streamable_subset(FLAC__StreamEncoder *e, FLAC__bool v) { UNINIT_OR_FALSE(e); e->protected_->streamable_subset = v; return 1; }
FLAC__bool FLAC__stream_encoder_set_do_md5(FLAC__StreamEncoder *e, FLAC__bool v) { UNINIT_OR_FALSE(e); e->protected_->do_md5 = v; return 1; }
FLAC__bool FLAC__stream_encoder_set_channels(FLAC__StreamEncoder *e, unsigned v) { UNINIT_OR_FALSE(e); e->protected_->channels = v; return 1; }
FLAC__bool FLAC__stream_encoder_set_bits_per_sample(FLAC__StreamEncoder *e, unsigned v) { UNINIT_OR_FALSE(e); e->protected_->bits_per_sample = v; return 1; }
FLAC__bool FLAC__stream_encoder_set_sample_rate(FLAC__StreamEncoder *e, unsigned v) { UNINIT_OR_FALSE(e); e->protected_->sample_rate = v; return 1; }
FLAC__bool FLAC__stream_encoder_set_blocksize(FLAC__StreamEncoder *e, unsigned v) { UNINIT_OR_FALSE(e); e->protected_->blocksize = v; return 1; }
FLAC__bool FLAC__stream_encoder_set_do_mid_side_stereo(FLAC__StreamEncoder *e, FLAC__bool v) { UNINIT_OR_FALSE(e); e->protected_->do_mid_side_stereo = v; return 1; }
FLAC__bool FLAC__stream_encoder_set_loose_mid_side_stereo(FLAC__StreamEncoder *e, FLAC__bool v) { UNINIT_OR_FALSE(e); e->protected_->loose_mid_side_stereo = v; return 1; }
FLAC__bool FLAC__stream_encoder_set_max_lpc_order(FLAC__StreamEncoder *e, unsigned v) { UNINIT_OR_FALSE(e); e->protected_->max_lpc_order = v; return 1; }
FLAC__bool FLAC__stream_encoder_set_qlp_coeff_precision(FLAC__StreamEncoder *e, unsigned v) { UNINIT_OR_FALSE(e); e->protected_->qlp_coeff_precision = v; return 1; }
FLAC__bool FLAC__stream_encoder_set_do_qlp_coeff_prec_search(FLAC__StreamEncoder *e, FLAC__bool v) { UNINIT_OR_FALSE(e); e->protected_->do_qlp_coeff_prec_search = v; return 1; }
FLAC__bool FLAC__stream_encoder_set_do_escape_coding(FLAC__StreamEncoder *e, FLAC__bool) { UNINIT_OR_FALSE(e); return 1; /* a no-op in the reference (:1630-1644) */ }
FLAC__bool FLAC__stream_encoder_set_do_exhaustive_model_search(FLAC__StreamEncoder *e, FLAC__bool v) { UNINIT_OR_FALSE(e); e->protected_->do_exhaustive_model_search = v; return 1; }
FLAC__bool FLAC__stream_encoder_set_min_residual_partition_order(FLAC__StreamEncoder *e, unsigned v) { UNINIT_OR_FALSE(e); e->protected_->min_residual_partition_order = v; return 1; }
FLAC__bool FLAC__stream_encoder_set_max_residual_partition_order(FLAC__StreamEncoder *e, unsigned v) { UNINIT_OR_FALSE(e); e->protected_->max_residual_partition_order = v; return 1; }
FLAC__bool FLAC__stream_encoder_set_rice_parameter_search_dist(FLAC__StreamEncoder *e, unsigned) { UNINIT_OR_FALSE(e); return 1; /* a no-op in the reference (:1679-1693) */ }
FLAC__bool FLAC__stream_encoder_set_total_samples_estimate(FLAC__StreamEncoder *e, FLAC__uint64 v) { UNINIT_OR_FALSE(e); e->protected_->total_samples_estimate = v; return 1; }
FLAC__bool FLAC__stream_encoder_disable_constant_subframes(FLAC__StreamEncoder *e, FLAC__bool v) { UNINIT_OR_FALSE(e); e->private_->disable_constant_subframes = v; return 1; }
FLAC__bool FLAC__stream_encoder_disable_fixed_subframes(FLAC__StreamEncoder *e, FLAC__bool v) { UNINIT_OR_FALSE(e); e->private_->disable_fixed_subframes = v; return 1; }
FLAC__bool FLAC__stream_encoder_disable_verbatim_subframes(FLAC__StreamEncoder *e, FLAC__bool v) { UNINIT_OR_FALSE(e); e->private_->disable_verbatim_subframes = v; return 1; }

FLAC__bool FLAC__stream_encoder_set_compression_level(FLAC__StreamEncoder *e, unsigned value)
{
	UNINIT_OR_FALSE(e);
	if (value > 8)
		value = 8;
	FLAC__StreamEncoderProtected *p = e->protected_;
	p->do_mid_side_stereo = kLevels[value].ms;
	p->loose_mid_side_stereo = kLevels[value].loose;
	p->num_apodizations = 1;
	p->apodizations[0].type = FLACAMD_APOD_TUKEY;
	p->apodizations[0].param = 0.5f;
	p->max_lpc_order = kLevels[value].lpc;
	p->qlp_coeff_precision = 0;
	p->do_qlp_coeff_prec_search = 0;
	p->do_exhaustive_model_search = kLevels[value].exhaustive;
	p->min_residual_partition_order = 0;
	p->max_residual_partition_order = kLevels[value].maxpart;
	return 1;
}

// stream_encoder.c:1526
FLAC__bool FLAC__stream_encoder_set_apodization(FLAC__StreamEncoder *e, const char *spec)
{
	UNINIT_OR_FALSE(e);
	FLAC__StreamEncoderProtected *p = e->protected_;
	static const struct { const char *name; uint32_t type; } names[] = {
		{"bartlett", FLACAMD_APOD_BARTLETT}, {"bartlett_hann", FLACAMD_APOD_BARTLETT_HANN}, {"blackman", FLACAMD_APOD_BLACKMAN},
		{"blackman_harris_4term_92db", FLACAMD_APOD_BLACKMAN_HARRIS_4TERM_92DB}, {"connes", FLACAMD_APOD_CONNES}, {"flattop", FLACAMD_APOD_FLATTOP},
		{"hamming", FLACAMD_APOD_HAMMING}, {"hann", FLACAMD_APOD_HANN}, {"kaiser_bessel", FLACAMD_APOD_KAISER_BESSEL}, {"nuttall", FLACAMD_APOD_NUTTALL},
		{"rectangle", FLACAMD_APOD_RECTANGLE}, {"triangle", FLACAMD_APOD_TRIANGLE}, {"welch", FLACAMD_APOD_WELCH}};
	p->num_apodizations = 0;
	for (;;) {
		const char *s = strchr(spec, ';');
		const size_t n = s ? (size_t)(s - spec) : strlen(spec);
		bool matched = false;
		for (size_t k = 0; k < sizeof names / sizeof names[0] && !matched; k++)
			if (n == strlen(names[k].name) && strncmp(names[k].name, spec, n) == 0) {
				p->apodizations[p->num_apodizations++].type = names[k].type;
				matched = true;
			}
		if (!matched && n > 7 && strncmp("gauss(", spec, 6) == 0) {
			const float stddev = (float)strtod(spec + 6, 0);
			if (stddev > 0.0 && stddev <= 0.5) {
				p->apodizations[p->num_apodizations].param = stddev;
				p->apodizations[p->num_apodizations++].type = FLACAMD_APOD_GAUSS;
			}
		} else if (!matched && n > 7 && strncmp("tukey(", spec, 6) == 0) {
			const float pp = (float)strtod(spec + 6, 0);
			if (pp >= 0.0 && pp <= 1.0) {
				p->apodizations[p->num_apodizations].param = pp;
				p->apodizations[p->num_apodizations++].type = FLACAMD_APOD_TUKEY;
			}
		}
		if (p->num_apodizations == 32 || !s)
			break;
		spec = s + 1;
	}
	if (p->num_apodizations == 0) {
		p->num_apodizations = 1;
		p->apodizations[0].type = FLACAMD_APOD_TUKEY;
		p->apodizations[0].param = 0.5f;
	}
	return 1;
}

// stream_encoder.c:1708
FLAC__bool FLAC__stream_encoder_set_metadata(FLAC__StreamEncoder *e, FLAC__StreamMetadata **metadata, unsigned num_blocks)
{
	UNINIT_OR_FALSE(e);
	FLAC__StreamEncoderProtected *p = e->protected_;
	if (!metadata)
		num_blocks = 0;
	if (num_blocks == 0)
		metadata = 0;
	if (p->metadata) {
		free(p->metadata);
		p->metadata = 0;
		p->num_metadata_blocks = 0;
	}
	if (num_blocks) {
		FLAC__StreamMetadata **m = (FLAC__StreamMetadata **)malloc(sizeof(m[0]) * num_blocks);
		if (!m)
			return 0;
		memcpy(m, metadata, sizeof(m[0]) * num_blocks);
		p->metadata = m;
		p->num_metadata_blocks = num_blocks;
	}
	return 1;
}

// ---- getters (stream_encoder.c:1775-1966) ----
FLAC__StreamEncoderState FLAC__stream_encoder_get_state(const FLAC__StreamEncoder *e) { return e->protected_->state; }
int FLAC__stream_encoder_get_verify_decoder_state(const FLAC__StreamEncoder *e) { return e->protected_->verify ? 2 /* SEARCH_FOR_FRAME_SYNC */ : 9 /* UNINITIALIZED */; }
const char *FLAC__stream_encoder_get_resolved_state_string(const FLAC__StreamEncoder *e) { return FLAC__StreamEncoderStateString[e->protected_->state]; }
void FLAC__stream_encoder_get_verify_decoder_error_stats(const FLAC__StreamEncoder *e, FLAC__uint64 *absolute_sample, unsigned *frame_number, unsigned *channel,
                                                         unsigned *sample, FLAC__int32 *expected, FLAC__int32 *got)
{
	const FLAC__StreamEncoderPrivate *q = e->private_;
	if (absolute_sample) *absolute_sample = q->ve_absolute_sample;
	if (frame_number) *frame_number = q->ve_frame;
	if (channel) *channel = q->ve_channel;
	if (sample) *sample = q->ve_sample;
	if (expected) *expected = q->ve_expected;
	if (got) *got = q->ve_got;
}
FLAC__bool FLAC__stream_encoder_get_verify(const FLAC__StreamEncoder *e) { return e->protected_->verify; }
FLAC__bool FLAC__stream_encoder_get_streamable_subset(const FLAC__StreamEncoder *e) { return e->protected_->streamable_subset; }
FLAC__bool FLAC__stream_encoder_get_do_md5(const FLAC__StreamEncoder *e) { return e->protected_->do_md5; }
unsigned FLAC__stream_encoder_get_channels(const FLAC__StreamEncoder *e) { return e->protected_->channels; }
unsigned FLAC__stream_encoder_get_bits_per_sample(const FLAC__StreamEncoder *e) { return e->protected_->bits_per_sample; }
unsigned FLAC__stream_encoder_get_sample_rate(const FLAC__StreamEncoder *e) { return e->protected_->sample_rate; }
unsigned FLAC__stream_encoder_get_blocksize(const FLAC__StreamEncoder *e) { return e->protected_->blocksize; }
FLAC__bool FLAC__stream_encoder_get_do_mid_side_stereo(const FLAC__StreamEncoder *e) { return e->protected_->do_mid_side_stereo; }
FLAC__bool FLAC__stream_encoder_get_loose_mid_side_stereo(const FLAC__StreamEncoder *e) { return e->protected_->loose_mid_side_stereo; }
unsigned FLAC__stream_encoder_get_max_lpc_order(const FLAC__StreamEncoder *e) { return e->protected_->max_lpc_order; }
unsigned FLAC__stream_encoder_get_qlp_coeff_precision(const FLAC__StreamEncoder *e) { return e->protected_->qlp_coeff_precision; }
FLAC__bool FLAC__stream_encoder_get_do_qlp_coeff_prec_search(const FLAC__StreamEncoder *e) { return e->protected_->do_qlp_coeff_prec_search; }
FLAC__bool FLAC__stream_encoder_get_do_escape_coding(const FLAC__StreamEncoder *e) { return e->protected_->do_escape_coding; }
FLAC__bool FLAC__stream_encoder_get_do_exhaustive_model_search(const FLAC__StreamEncoder *e) { return e->protected_->do_exhaustive_model_search; }
unsigned FLAC__stream_encoder_get_min_residual_partition_order(const FLAC__StreamEncoder *e) { return e->protected_->min_residual_partition_order; }
unsigned FLAC__stream_encoder_get_max_residual_partition_order(const FLAC__StreamEncoder *e) { return e->protected_->max_residual_partition_order; }
unsigned FLAC__stream_encoder_get_rice_parameter_search_dist(const FLAC__StreamEncoder *e) { return e->protected_->rice_parameter_search_dist; }
FLAC__uint64 FLAC__stream_encoder_get_total_samples_estimate(const FLAC__StreamEncoder *e) { return e->protected_->total_samples_estimate; }

}  // extern "C"
