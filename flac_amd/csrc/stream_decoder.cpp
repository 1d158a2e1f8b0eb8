// libFLAC-compatible stream decoder on top of the device batch ABI.
//
// Mirrors src/libFLAC/stream_decoder.c: the state machine of process_single /
// process_until_end_of_metadata / process_until_end_of_stream, metadata parsing and filtering,
// read / seek / tell / length / eof callback handling, MD5 checking, flush / reset and sample-exact
// seeking.  Frames are not decoded one by one: input is gathered into a large buffer, every frame in
// it is decoded by one flacamd_decode_streams call, and the write callback is then fed frame after
// frame in stream order from the decoded batch.  Callbacks run on the caller's thread.
//
// What the write callback's FLAC__Frame carries: the complete header and footer, and per subframe what
// read_subframe_ (stream_decoder.c:2450-2713) learns on the way to the samples - type, wasted bits, order,
// warm-up samples, coefficient precision, quantization level, coefficients, Rice method, partition order, and
// the first 256 partitions' parameters / raw bits - which the device kernels note for every subframe and the
// batch brings back beside the samples.  The residual itself stays on the device: the residual pointers
// (FLAC__Subframe_Fixed / _LPC.residual, FLAC__Subframe_Verbatim.data) point at zeros.
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <deque>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "flac_amd.h"
#include "md5.h"
#include "stream_common.h"

extern "C" {

extern const char *const FLAC__StreamDecoderStateString[] = {
	"FLAC__STREAM_DECODER_SEARCH_FOR_METADATA", "FLAC__STREAM_DECODER_READ_METADATA", "FLAC__STREAM_DECODER_SEARCH_FOR_FRAME_SYNC",
	"FLAC__STREAM_DECODER_READ_FRAME", "FLAC__STREAM_DECODER_END_OF_STREAM", "FLAC__STREAM_DECODER_OGG_ERROR",
	"FLAC__STREAM_DECODER_SEEK_ERROR", "FLAC__STREAM_DECODER_ABORTED", "FLAC__STREAM_DECODER_MEMORY_ALLOCATION_ERROR",
	"FLAC__STREAM_DECODER_UNINITIALIZED"};
extern const char *const FLAC__StreamDecoderInitStatusString[] = {
	"FLAC__STREAM_DECODER_INIT_STATUS_OK", "FLAC__STREAM_DECODER_INIT_STATUS_UNSUPPORTED_CONTAINER",
	"FLAC__STREAM_DECODER_INIT_STATUS_INVALID_CALLBACKS", "FLAC__STREAM_DECODER_INIT_STATUS_MEMORY_ALLOCATION_ERROR",
	"FLAC__STREAM_DECODER_INIT_STATUS_ERROR_OPENING_FILE", "FLAC__STREAM_DECODER_INIT_STATUS_ALREADY_INITIALIZED"};
extern const char *const FLAC__StreamDecoderReadStatusString[] = {
	"FLAC__STREAM_DECODER_READ_STATUS_CONTINUE", "FLAC__STREAM_DECODER_READ_STATUS_END_OF_STREAM", "FLAC__STREAM_DECODER_READ_STATUS_ABORT"};
extern const char *const FLAC__StreamDecoderSeekStatusString[] = {
	"FLAC__STREAM_DECODER_SEEK_STATUS_OK", "FLAC__STREAM_DECODER_SEEK_STATUS_ERROR", "FLAC__STREAM_DECODER_SEEK_STATUS_UNSUPPORTED"};
extern const char *const FLAC__StreamDecoderTellStatusString[] = {
	"FLAC__STREAM_DECODER_TELL_STATUS_OK", "FLAC__STREAM_DECODER_TELL_STATUS_ERROR", "FLAC__STREAM_DECODER_TELL_STATUS_UNSUPPORTED"};
extern const char *const FLAC__StreamDecoderLengthStatusString[] = {
	"FLAC__STREAM_DECODER_LENGTH_STATUS_OK", "FLAC__STREAM_DECODER_LENGTH_STATUS_ERROR", "FLAC__STREAM_DECODER_LENGTH_STATUS_UNSUPPORTED"};
extern const char *const FLAC__StreamDecoderWriteStatusString[] = {
	"FLAC__STREAM_DECODER_WRITE_STATUS_CONTINUE", "FLAC__STREAM_DECODER_WRITE_STATUS_ABORT"};
extern const char *const FLAC__StreamDecoderErrorStatusString[] = {
	"FLAC__STREAM_DECODER_ERROR_STATUS_LOST_SYNC", "FLAC__STREAM_DECODER_ERROR_STATUS_BAD_HEADER",
	"FLAC__STREAM_DECODER_ERROR_STATUS_FRAME_CRC_MISMATCH", "FLAC__STREAM_DECODER_ERROR_STATUS_UNPARSEABLE_STREAM"};

}  // extern "C"

// src/libFLAC/include/protected/stream_decoder.h:40-51
struct FLAC__StreamDecoderProtected {
	FLAC__StreamDecoderState state;
	unsigned channels;
	FLAC__ChannelAssignment channel_assignment;
	unsigned bits_per_sample, sample_rate, blocksize;
	FLAC__bool md5_checking;
};

// The MD5 of the decoded samples (FLAC__MD5Accumulate in write_audio_frame_to_client_, stream_decoder.c:2953) is serial
// work over the whole stream, about as long as everything else the decoder does on the host: it runs on a thread of its
// own, over the frames' samples where they lie in the decoded batch, in the order the frames are handed out.
struct Md5Lane {
	std::thread thread;
	std::mutex mu;
	std::condition_variable cv;
	// count samples at p, interleaved; or, channels > 0, `count` wide samples per channel, channel c's at p + c * count
	struct Job { const FLAC__int32 *p; size_t count; unsigned bytes_per_sample, channels; };
	std::deque<Job> jobs;
	uint64_t posted, finished;
	bool quit, started, failed;
	FaMd5 *md5;
	Md5Lane() : posted(0), finished(0), quit(false), started(false), failed(false), md5(0) {}
};

struct FLAC__StreamDecoderPrivate {
	FLAC__StreamDecoderReadCallback read_callback;
	FLAC__StreamDecoderSeekCallback seek_callback;
	FLAC__StreamDecoderTellCallback tell_callback;
	FLAC__StreamDecoderLengthCallback length_callback;
	FLAC__StreamDecoderEofCallback eof_callback;
	FLAC__StreamDecoderWriteCallback write_callback;
	FLAC__StreamDecoderMetadataCallback metadata_callback;
	FLAC__StreamDecoderErrorCallback error_callback;
	void *client_data;
	FILE *file;
	FLAC__uint64 samples_decoded;
	FLAC__bool has_stream_info, has_seek_table;
	FLAC__StreamMetadata stream_info, seek_table;
	FLAC__bool metadata_filter[128];
	std::vector<FLAC__byte> *filter_ids;  // 4 bytes per application id
	FLAC__Frame frame, last_frame;
	FLAC__bool cached;
	FLAC__byte header_warmup[2], lookahead;
	FLAC__bool do_md5_checking, internal_reset_hack, is_seeking;
	FaMd5 md5;
	Md5Lane *md5_lane;
	FLAC__byte computed_md5sum[16];
	FLAC__uint64 first_frame_offset, target_sample;
	unsigned unparseable_frame_count;
	unsigned fixed_block_size;  // samples per frame number, once known (stream_decoder.c:178)

	// ---- input gathered from the read callback ----
	FLAC__byte *in;          // malloc'ed until the device is in use, then page-locked
	bool in_pinned;
	size_t in_cap, in_rd, in_wr;
	bool input_eof;          // the client reported end of stream; what is buffered is all there is

	// ---- device side ----
	flacamd_decoder *dev;
	unsigned dev_channels, dev_bps, dev_blocksize;
	uint32_t max_frames;
	size_t batch_bytes;      // input bytes decoded per device call
	uint64_t out_cap_wide;
	FLAC__byte *d_in;
	FLAC__int32 *d_out, *h_out;
	FLAC__int32 *planes[FLAC__MAX_CHANNELS];
	const FLAC__int32 *plane_ptrs[FLAC__MAX_CHANNELS];

	// ---- the decoded batch being handed out ----
	std::vector<flacamd_frame_info> *table;
	std::vector<flacamd_subframe_info> *subinfo;   // `channels` records per row of the table
	// what FLAC__Frame.subframes[] points at: per channel the partition parameters and raw bits of the frame in hand
	FLAC__EntropyCodingMethod_PartitionedRiceContents rice_contents[FLAC__MAX_CHANNELS];
	std::vector<unsigned> *rice_words;             // 2 x FLACAMD_SUBFRAME_PARAMS per channel
	std::vector<FLAC__int32> *no_residual;         // max blocksize zeros
	size_t cursor;
	bool batch_active, batch_eof, batch_cut, batch_full, want_subinfo;
	bool planar;   // the device hands a frame's samples out channel by channel: the write callback's planes point into the batch
	size_t batch_base, batch_len;   // the batch covers in[batch_base, batch_base+batch_len)
	uint64_t walk_pos;              // bytes of the batch already accounted for
	bool gap_reported;
};

namespace {

// FLAC_AMD_DEC_TIMES=1: where a decoder's host time went, printed by finish() (a measurement aid)
struct DecTimes { double setup, gather, to_dev, decode, from_dev; } g_times;
inline double now_s() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

const FLAC__byte kId3[3] = {'I', 'D', '3'};

void set_defaults(FLAC__StreamDecoder *d)
{
	FLAC__StreamDecoderPrivate *q = d->private_;
	q->read_callback = 0;
	q->seek_callback = 0;
	q->tell_callback = 0;
	q->length_callback = 0;
	q->eof_callback = 0;
	q->write_callback = 0;
	q->metadata_callback = 0;
	q->error_callback = 0;
	q->client_data = 0;
	memset(q->metadata_filter, 0, sizeof q->metadata_filter);
	q->metadata_filter[FLAC__METADATA_TYPE_STREAMINFO] = 1;
	q->filter_ids->clear();
	d->protected_->md5_checking = 0;
}

void send_error(const FLAC__StreamDecoder *d, FLAC__StreamDecoderErrorStatus status)
{
	// send_error_to_client_ (stream_decoder.c:2965)
	if (!d->private_->is_seeking)
		d->private_->error_callback(d, status, d->private_->client_data);
	else if (status == FLAC__STREAM_DECODER_ERROR_STATUS_UNPARSEABLE_STREAM)
		d->private_->unparseable_frame_count++;
}

// read_callback_ (stream_decoder.c:2787): one trip to the client.  False when nothing more will
// come; the state then says why.
bool client_read(FLAC__StreamDecoder *d, FLAC__byte *dst, size_t *bytes)
{
	FLAC__StreamDecoderPrivate *q = d->private_;
	if (q->eof_callback && q->eof_callback(d, q->client_data)) {
		*bytes = 0;
		d->protected_->state = FLAC__STREAM_DECODER_END_OF_STREAM;
		return false;
	}
	if (*bytes == 0) {
		d->protected_->state = FLAC__STREAM_DECODER_ABORTED;
		return false;
	}
	if (q->is_seeking && q->unparseable_frame_count > 20) {
		d->protected_->state = FLAC__STREAM_DECODER_ABORTED;
		return false;
	}
	const FLAC__StreamDecoderReadStatus status = q->read_callback(d, dst, bytes, q->client_data);
	if (status == FLAC__STREAM_DECODER_READ_STATUS_ABORT) {
		d->protected_->state = FLAC__STREAM_DECODER_ABORTED;
		return false;
	}
	if (*bytes == 0 && (status == FLAC__STREAM_DECODER_READ_STATUS_END_OF_STREAM || (q->eof_callback && q->eof_callback(d, q->client_data)))) {
		d->protected_->state = FLAC__STREAM_DECODER_END_OF_STREAM;
		return false;
	}
	return true;
}

// ---------------------------------------------------------------------------------------------
// input buffer
// ---------------------------------------------------------------------------------------------
void release_input(FLAC__StreamDecoderPrivate *q)
{
	if (q->in) {
		if (q->in_pinned)
			(void)hipHostFree(q->in);
		else
			free(q->in);
	}
	q->in = 0;
	q->in_pinned = false;
	q->in_cap = q->in_rd = q->in_wr = 0;
}

bool reserve_input(FLAC__StreamDecoder *d, size_t cap)
{
	FLAC__StreamDecoderPrivate *q = d->private_;
	if (cap <= q->in_cap)
		return true;
	if (q->in_pinned)
		return false;  // the page-locked buffer has its final size
	FLAC__byte *nb = (FLAC__byte *)realloc(q->in, cap);
	if (!nb) {
		d->protected_->state = FLAC__STREAM_DECODER_MEMORY_ALLOCATION_ERROR;
		return false;
	}
	q->in = nb;
	q->in_cap = cap;
	return true;
}

// Drop consumed bytes, except the last `history` of them (find_metadata un-reads two).
void compact_input(FLAC__StreamDecoderPrivate *q, size_t history = 0)
{
	if (q->in_rd <= history)
		return;
	const size_t from = q->in_rd - history;
	memmove(q->in, q->in + from, q->in_wr - from);
	q->in_wr -= from;
	q->in_rd = history;
}

// Make at least n unread bytes available (metadata phase).  False: the state says why.
bool need(FLAC__StreamDecoder *d, size_t n)
{
	FLAC__StreamDecoderPrivate *q = d->private_;
	while (q->in_wr - q->in_rd < n) {
		if (q->input_eof) {
			d->protected_->state = FLAC__STREAM_DECODER_END_OF_STREAM;
			return false;
		}
		const size_t room = n > 8192 ? n : 8192;
		if (q->in_wr + room > q->in_cap) {
			compact_input(q, 2);
			if (q->in_wr + room > q->in_cap && !reserve_input(d, q->in_wr + room + 8192)) {
				d->protected_->state = FLAC__STREAM_DECODER_MEMORY_ALLOCATION_ERROR;
				return false;
			}
		}
		size_t bytes = q->in_cap - q->in_wr;
		if (bytes > 8192 && !q->in_pinned)
			bytes = 8192;  // the reference's reader asks for at most its 8 KiB buffer (bitreader.c:96)
		if (!client_read(d, q->in + q->in_wr, &bytes))
			return false;
		q->in_wr += bytes;
	}
	return true;
}

bool read_u8(FLAC__StreamDecoder *d, unsigned *x)
{
	if (!need(d, 1))
		return false;
	*x = d->private_->in[d->private_->in_rd++];
	return true;
}

bool read_be(FLAC__StreamDecoder *d, unsigned nbytes, FLAC__uint64 *x)
{
	if (!need(d, nbytes))
		return false;
	FLAC__StreamDecoderPrivate *q = d->private_;
	FLAC__uint64 v = 0;
	for (unsigned i = 0; i < nbytes; i++)
		v = (v << 8) | q->in[q->in_rd++];
	*x = v;
	return true;
}

bool read_be32(FLAC__StreamDecoder *d, unsigned nbytes, FLAC__uint32 *x)
{
	FLAC__uint64 v;
	if (!read_be(d, nbytes, &v))
		return false;
	*x = (FLAC__uint32)v;
	return true;
}

bool read_le32(FLAC__StreamDecoder *d, FLAC__uint32 *x)
{
	if (!need(d, 4))
		return false;
	FLAC__StreamDecoderPrivate *q = d->private_;
	const FLAC__byte *p = q->in + q->in_rd;
	*x = (FLAC__uint32)p[0] | ((FLAC__uint32)p[1] << 8) | ((FLAC__uint32)p[2] << 16) | ((FLAC__uint32)p[3] << 24);
	q->in_rd += 4;
	return true;
}

bool read_block(FLAC__StreamDecoder *d, FLAC__byte *dst, size_t n)
{
	FLAC__StreamDecoderPrivate *q = d->private_;
	while (n) {
		const size_t step = n < 65536 ? n : 65536;
		if (!need(d, step))
			return false;
		memcpy(dst, q->in + q->in_rd, step);
		q->in_rd += step;
		dst += step;
		n -= step;
	}
	return true;
}

bool skip_block(FLAC__StreamDecoder *d, size_t n)
{
	FLAC__StreamDecoderPrivate *q = d->private_;
	while (n) {
		const size_t step = n < 65536 ? n : 65536;
		if (!need(d, step))
			return false;
		q->in_rd += step;
		n -= step;
	}
	return true;
}

// ---------------------------------------------------------------------------------------------
// metadata
// ---------------------------------------------------------------------------------------------
bool has_id_filtered(FLAC__StreamDecoder *d, const FLAC__byte *id)
{
	const std::vector<FLAC__byte> &v = *d->private_->filter_ids;
	for (size_t i = 0; i + 4 <= v.size(); i += 4)
		if (0 == memcmp(&v[i], id, 4))
			return true;
	return false;
}

// skip_id3v2_tag_ (stream_decoder.c:1919)
bool skip_id3v2_tag(FLAC__StreamDecoder *d)
{
	FLAC__uint32 x;
	if (!read_be32(d, 3, &x))
		return false;
	unsigned skip = 0;
	for (int i = 0; i < 4; i++) {
		unsigned b;
		if (!read_u8(d, &b))
			return false;
		skip = (skip << 7) | (b & 0x7f);
	}
	return skip_block(d, skip);
}

// find_metadata_ (stream_decoder.c:1362)
bool find_metadata(FLAC__StreamDecoder *d)
{
	FLAC__StreamDecoderPrivate *q = d->private_;
	unsigned x, i, id;
	bool first = true;
	for (i = id = 0; i < 4;) {
		if (q->cached) {
			x = q->lookahead;
			q->cached = 0;
		} else if (!read_u8(d, &x))
			return false;
		if (x == FLAC__STREAM_SYNC_STRING[i]) {
			first = true;
			i++;
			id = 0;
			continue;
		}
		if (x == kId3[id]) {
			id++;
			i = 0;
			if (id == 3 && !skip_id3v2_tag(d))
				return false;
			if (id == 3)
				id = 0;  // the reference would index one past its 3-byte tag here
			continue;
		}
		id = 0;
		if (x == 0xff) {
			q->header_warmup[0] = (FLAC__byte)x;
			if (!read_u8(d, &x))
				return false;
			if (x == 0xff) {
				q->lookahead = (FLAC__byte)x;
				q->cached = 1;
			} else if (x >> 1 == 0x7c) {
				// a stream that begins with a frame: hand the two sync bytes back to the frame scan
				q->header_warmup[1] = (FLAC__byte)x;
				q->in_rd -= 2;
				d->protected_->state = FLAC__STREAM_DECODER_READ_FRAME;
				return true;
			}
		}
		i = 0;
		if (first) {
			send_error(d, FLAC__STREAM_DECODER_ERROR_STATUS_LOST_SYNC);
			first = false;
		}
	}
	d->protected_->state = FLAC__STREAM_DECODER_READ_METADATA;
	return true;
}

bool alloc_fail(FLAC__StreamDecoder *d)
{
	d->protected_->state = FLAC__STREAM_DECODER_MEMORY_ALLOCATION_ERROR;
	return false;
}

// read_metadata_streaminfo_ (stream_decoder.c:1598)
bool read_streaminfo(FLAC__StreamDecoder *d, FLAC__bool is_last, unsigned length)
{
	FLAC__StreamDecoderPrivate *q = d->private_;
	FLAC__byte b[34];
	if (!need(d, 34))
		return false;
	memcpy(b, q->in + q->in_rd, 34);
	q->in_rd += 34;
	FLAC__StreamMetadata &m = q->stream_info;
	m.type = FLAC__METADATA_TYPE_STREAMINFO;
	m.is_last = is_last;
	m.length = length;
	FLAC__StreamMetadata_StreamInfo &s = m.data.stream_info;
	s.min_blocksize = (b[0] << 8) | b[1];
	s.max_blocksize = (b[2] << 8) | b[3];
	s.min_framesize = (b[4] << 16) | (b[5] << 8) | b[6];
	s.max_framesize = (b[7] << 16) | (b[8] << 8) | b[9];
	s.sample_rate = (b[10] << 12) | (b[11] << 4) | (b[12] >> 4);
	s.channels = ((b[12] >> 1) & 7) + 1;
	s.bits_per_sample = (((b[12] & 1) << 4) | (b[13] >> 4)) + 1;
	s.total_samples = ((FLAC__uint64)(b[13] & 15) << 32) | ((FLAC__uint64)b[14] << 24) | (b[15] << 16) | (b[16] << 8) | b[17];
	memcpy(s.md5sum, b + 18, 16);
	return skip_block(d, length - 34);
}

// read_metadata_seektable_ (stream_decoder.c:1669)
bool read_seektable(FLAC__StreamDecoder *d, FLAC__bool is_last, unsigned length)
{
	FLAC__StreamDecoderPrivate *q = d->private_;
	FLAC__StreamMetadata &m = q->seek_table;
	m.type = FLAC__METADATA_TYPE_SEEKTABLE;
	m.is_last = is_last;
	m.length = length;
	FLAC__StreamMetadata_SeekTable &t = m.data.seek_table;
	t.num_points = length / FLAC__STREAM_METADATA_SEEKPOINT_LENGTH;
	void *np = realloc(t.points, (t.num_points ? t.num_points : 1) * sizeof(FLAC__StreamMetadata_SeekPoint));
	if (!np)
		return alloc_fail(d);
	t.points = (FLAC__StreamMetadata_SeekPoint *)np;
	for (unsigned i = 0; i < t.num_points; i++) {
		FLAC__uint64 a, b;
		FLAC__uint32 c;
		if (!read_be(d, 8, &a) || !read_be(d, 8, &b) || !read_be32(d, 2, &c))
			return false;
		t.points[i].sample_number = a;
		t.points[i].stream_offset = b;
		t.points[i].frame_samples = c;
	}
	length -= t.num_points * FLAC__STREAM_METADATA_SEEKPOINT_LENGTH;
	return length ? skip_block(d, length) : true;
}

// n bytes of the block, followed by `extra` zero bytes, into a buffer that grows as the bytes arrive: a length field of a
// damaged stream cannot reserve more memory than the stream backs up.  *out is set (and owned by the caller) even when
// the input ends early.
bool read_blob(FLAC__StreamDecoder *d, size_t n, size_t extra, FLAC__byte **out)
{
	const size_t kStep = (size_t)1 << 20;
	size_t cap = (n < kStep ? n : kStep) + extra, have = 0;
	if (0 == (*out = (FLAC__byte *)malloc(cap ? cap : 1)))
		return alloc_fail(d);
	while (have < n) {
		const size_t want = n - have < kStep ? n - have : kStep;
		if (have + want + extra > cap) {
			cap = have + want + extra > 2 * cap ? have + want + extra : 2 * cap;
			if (cap > n + extra)
				cap = n + extra;
			FLAC__byte *nb = (FLAC__byte *)realloc(*out, cap);
			if (!nb)
				return alloc_fail(d);
			*out = nb;
		}
		if (!read_block(d, *out + have, want))
			return false;
		have += want;
	}
	memset(*out + n, 0, extra);
	return true;
}

bool read_vc_entry(FLAC__StreamDecoder *d, FLAC__StreamMetadata_VorbisComment_Entry *e)
{
	if (!read_le32(d, &e->length))
		return false;
	if (e->length == 0) {
		e->entry = 0;
		return true;
	}
	return read_blob(d, e->length, 1, &e->entry);
}

// read_metadata_vorbiscomment_ (stream_decoder.c:1711)
bool read_vorbiscomment(FLAC__StreamDecoder *d, FLAC__StreamMetadata_VorbisComment *o)
{
	if (!read_vc_entry(d, &o->vendor_string))
		return false;
	if (!read_le32(d, &o->num_comments))
		return false;
	o->comments = 0;
	// the array grows with the comments that are really there (each takes four bytes of input at least), so a damaged
	// count cannot reserve 64 GiB; num_comments is what the array holds if reading stops early
	const FLAC__uint32 declared = o->num_comments;
	FLAC__uint32 cap = 0;
	o->num_comments = 0;
	for (FLAC__uint32 i = 0; i < declared; i++) {
		if (i == cap) {
			const FLAC__uint32 ncap = cap ? (cap > declared / 2 ? declared : 2 * cap) : (declared < 1024 ? declared : 1024);
			void *nb = realloc(o->comments, (size_t)ncap * sizeof *o->comments);
			if (!nb)
				return alloc_fail(d);
			o->comments = (FLAC__StreamMetadata_VorbisComment_Entry *)nb;
			memset(o->comments + cap, 0, (size_t)(ncap - cap) * sizeof *o->comments);
			cap = ncap;
		}
		o->num_comments = i + 1;
		if (!read_vc_entry(d, &o->comments[i]))
			return false;
	}
	return true;
}

// read_metadata_cuesheet_ (stream_decoder.c:1768)
bool read_cuesheet(FLAC__StreamDecoder *d, FLAC__StreamMetadata_CueSheet *o)
{
	memset(o, 0, sizeof *o);
	if (!read_block(d, (FLAC__byte *)o->media_catalog_number, 128))
		return false;
	if (!read_be(d, 8, &o->lead_in))
		return false;
	unsigned x;
	if (!read_u8(d, &x))
		return false;
	o->is_cd = (x & 0x80) ? 1 : 0;
	if (!skip_block(d, 258))
		return false;
	if (!read_u8(d, &x))
		return false;
	o->num_tracks = x;
	if (o->num_tracks) {
		o->tracks = (FLAC__StreamMetadata_CueSheet_Track *)calloc(o->num_tracks, sizeof *o->tracks);
		if (!o->tracks)
			return alloc_fail(d);
		for (unsigned i = 0; i < o->num_tracks; i++) {
			FLAC__StreamMetadata_CueSheet_Track *t = &o->tracks[i];
			if (!read_be(d, 8, &t->offset) || !read_u8(d, &x))
				return false;
			t->number = (FLAC__byte)x;
			if (!read_block(d, (FLAC__byte *)t->isrc, 12) || !read_u8(d, &x))
				return false;
			t->type = (x >> 7) & 1;
			t->pre_emphasis = (x >> 6) & 1;
			if (!skip_block(d, 13) || !read_u8(d, &x))
				return false;
			t->num_indices = (FLAC__byte)x;
			if (t->num_indices) {
				t->indices = (FLAC__StreamMetadata_CueSheet_Index *)calloc(t->num_indices, sizeof *t->indices);
				if (!t->indices)
					return alloc_fail(d);
				for (unsigned j = 0; j < t->num_indices; j++) {
					if (!read_be(d, 8, &t->indices[j].offset) || !read_u8(d, &x))
						return false;
					t->indices[j].number = (FLAC__byte)x;
					if (!skip_block(d, 3))
						return false;
				}
			}
		}
	}
	return true;
}

// read_metadata_picture_ (stream_decoder.c:1851)
bool read_picture(FLAC__StreamDecoder *d, FLAC__StreamMetadata_Picture *o)
{
	FLAC__uint32 x;
	if (!read_be32(d, 4, &x))
		return false;
	o->type = (FLAC__StreamMetadata_Picture_Type)x;
	if (!read_be32(d, 4, &x))
		return false;
	if (!read_blob(d, x, 1, (FLAC__byte **)&o->mime_type))
		return false;
	if (!read_be32(d, 4, &x))
		return false;
	if (!read_blob(d, x, 1, &o->description))
		return false;
	if (!read_be32(d, 4, &o->width) || !read_be32(d, 4, &o->height) || !read_be32(d, 4, &o->depth) || !read_be32(d, 4, &o->colors) ||
	    !read_be32(d, 4, &o->data_length))
		return false;
	return read_blob(d, o->data_length, 0, &o->data);
}

void free_block_contents(FLAC__StreamMetadata *b)
{
	switch (b->type) {
	case FLAC__METADATA_TYPE_PADDING:
		break;
	case FLAC__METADATA_TYPE_APPLICATION:
		free(b->data.application.data);
		break;
	case FLAC__METADATA_TYPE_VORBIS_COMMENT:
		free(b->data.vorbis_comment.vendor_string.entry);
		if (b->data.vorbis_comment.comments)
			for (FLAC__uint32 i = 0; i < b->data.vorbis_comment.num_comments; i++)
				free(b->data.vorbis_comment.comments[i].entry);
		free(b->data.vorbis_comment.comments);
		break;
	case FLAC__METADATA_TYPE_CUESHEET:
		if (b->data.cue_sheet.tracks)
			for (unsigned i = 0; i < b->data.cue_sheet.num_tracks; i++)
				free(b->data.cue_sheet.tracks[i].indices);
		free(b->data.cue_sheet.tracks);
		break;
	case FLAC__METADATA_TYPE_PICTURE:
		free(b->data.picture.mime_type);
		free(b->data.picture.description);
		free(b->data.picture.data);
		break;
	default:
		free(b->data.unknown.data);
		break;
	}
}

// read_metadata_ (stream_decoder.c:1423)
bool read_metadata(FLAC__StreamDecoder *d)
{
	FLAC__StreamDecoderPrivate *q = d->private_;
	FLAC__uint32 head;
	if (!read_be32(d, 4, &head))
		return false;
	const FLAC__bool is_last = (head >> 31) & 1;
	const unsigned type = (head >> 24) & 0x7f;
	unsigned length = head & 0xffffff;

	if (type == FLAC__METADATA_TYPE_STREAMINFO) {
		if (!read_streaminfo(d, is_last, length))
			return false;
		q->has_stream_info = 1;
		static const FLAC__byte zero[16] = {0};
		if (0 == memcmp(q->stream_info.data.stream_info.md5sum, zero, 16))
			q->do_md5_checking = 0;
		if (!q->is_seeking && q->metadata_filter[FLAC__METADATA_TYPE_STREAMINFO] && q->metadata_callback)
			q->metadata_callback(d, &q->stream_info, q->client_data);
	} else if (type == FLAC__METADATA_TYPE_SEEKTABLE) {
		if (!read_seektable(d, is_last, length))
			return false;
		q->has_seek_table = 1;
		if (!q->is_seeking && q->metadata_filter[FLAC__METADATA_TYPE_SEEKTABLE] && q->metadata_callback)
			q->metadata_callback(d, &q->seek_table, q->client_data);
	} else {
		bool skip_it = !q->metadata_filter[type];
		unsigned real_length = length;
		FLAC__StreamMetadata block;
		memset(&block, 0, sizeof block);
		block.is_last = is_last;
		block.type = (FLAC__MetadataType)type;
		block.length = length;
		if (type == FLAC__METADATA_TYPE_APPLICATION) {
			if (!read_block(d, block.data.application.id, 4))
				return false;
			if (real_length < 4)
				return alloc_fail(d);  // the reference's choice of state (stream_decoder.c:1472)
			real_length -= 4;
			if (!q->filter_ids->empty() && has_id_filtered(d, block.data.application.id))
				skip_it = !skip_it;
		}
		if (skip_it) {
			if (!skip_block(d, real_length))
				return false;
		} else {
			bool ok = true;
			switch (type) {
			case FLAC__METADATA_TYPE_PADDING:
				ok = skip_block(d, real_length);
				break;
			case FLAC__METADATA_TYPE_APPLICATION:
				if (real_length) {
					if (0 == (block.data.application.data = (FLAC__byte *)malloc(real_length)))
						ok = alloc_fail(d);
					else
						ok = read_block(d, block.data.application.data, real_length);
				}
				break;
			case FLAC__METADATA_TYPE_VORBIS_COMMENT:
				ok = read_vorbiscomment(d, &block.data.vorbis_comment);
				break;
			case FLAC__METADATA_TYPE_CUESHEET:
				ok = read_cuesheet(d, &block.data.cue_sheet);
				break;
			case FLAC__METADATA_TYPE_PICTURE:
				ok = read_picture(d, &block.data.picture);
				break;
			default:
				if (real_length) {
					if (0 == (block.data.unknown.data = (FLAC__byte *)malloc(real_length)))
						ok = alloc_fail(d);
					else
						ok = read_block(d, block.data.unknown.data, real_length);
				}
				break;
			}
			if (ok && !q->is_seeking && q->metadata_callback)
				q->metadata_callback(d, &block, q->client_data);
			free_block_contents(&block);
			if (!ok)
				return false;
		}
	}
	if (is_last) {
		if (!FLAC__stream_decoder_get_decode_position(d, &q->first_frame_offset))
			q->first_frame_offset = 0;
		d->protected_->state = FLAC__STREAM_DECODER_SEARCH_FOR_FRAME_SYNC;
	}
	return true;
}

// ---------------------------------------------------------------------------------------------
// frames: host-side header parse (what the write callback's FLAC__FrameHeader needs beyond the
// device's frame table), device set-up, batch decode, in-order hand-out
// ---------------------------------------------------------------------------------------------
struct HostHeader {
	unsigned blocksize, sample_rate, channels, bits_per_sample;
	FLAC__ChannelAssignment assignment;
	bool variable;
	FLAC__uint64 number;
	unsigned length;  // bytes, CRC-8 included
	FLAC__byte crc;
};

FLAC__byte crc8(const FLAC__byte *p, unsigned n)
{
	unsigned crc = 0;
	for (unsigned i = 0; i < n; i++) {
		crc ^= p[i];
		for (int b = 0; b < 8; b++)
			crc = (crc & 0x80) ? ((crc << 1) ^ 0x07) & 0xff : (crc << 1) & 0xff;
	}
	return (FLAC__byte)crc;
}

enum HeaderVerdict {
	HDR_VALID,        // parsed, CRC-8 good
	HDR_BAD,          // FLAC__STREAM_DECODER_ERROR_STATUS_BAD_HEADER
	HDR_UNPARSEABLE,  // CRC-8 good but uses reserved codes (or needs a STREAMINFO that is not there)
	HDR_SHORT         // the bytes end before the header does
};

// read_frame_header_ (stream_decoder.c:2141) over bytes already in memory, p[0..1] being a sync
// code.  si = STREAMINFO or 0.  *consumed = bytes the reference's reader has moved past when it
// gives its verdict (a byte it puts back into its one-byte look-ahead is not counted).
HeaderVerdict classify_header(const FLAC__byte *p, size_t avail, const FLAC__StreamMetadata_StreamInfo *si, HostHeader *h, unsigned *consumed)
{
	bool unparseable = (p[1] & 2) != 0;
	if (avail < 4) {
		for (size_t i = 2; i < avail; i++)
			if (p[i] == 0xff) {
				*consumed = (unsigned)i;
				return HDR_BAD;
			}
		return HDR_SHORT;
	}
	for (unsigned i = 2; i < 4; i++)
		if (p[i] == 0xff) {  // a sync code cannot start inside a header (:2186)
			*consumed = i;
			return HDR_BAD;
		}
	const unsigned bs_code = p[2] >> 4, sr_code = p[2] & 15;
	static const unsigned rates[12] = {0, 88200, 176400, 192000, 8000, 16000, 22050, 24000, 32000, 44100, 48000, 96000};
	h->blocksize = h->sample_rate = 0;
	if (bs_code == 0)
		unparseable = true;
	if (sr_code == 0) {
		if (si)
			h->sample_rate = si->sample_rate;
		else
			unparseable = true;
	} else if (sr_code < 12)
		h->sample_rate = rates[sr_code];
	else if (sr_code == 15) {
		*consumed = 4;
		return HDR_BAD;
	}
	const unsigned ch_code = p[3] >> 4;
	h->channels = 2;
	h->assignment = FLAC__CHANNEL_ASSIGNMENT_INDEPENDENT;
	if (ch_code & 8) {
		if ((ch_code & 7) > 2)
			unparseable = true;
		else
			h->assignment = (FLAC__ChannelAssignment)((ch_code & 7) + 1);
	} else
		h->channels = ch_code + 1;
	h->bits_per_sample = 0;
	switch ((p[3] >> 1) & 7) {
	case 0:
		if (si)
			h->bits_per_sample = si->bits_per_sample;
		else
			unparseable = true;
		break;
	case 1: h->bits_per_sample = 8; break;
	case 2: h->bits_per_sample = 12; break;
	case 4: h->bits_per_sample = 16; break;
	case 5: h->bits_per_sample = 20; break;
	case 6: h->bits_per_sample = 24; break;
	default: unparseable = true; break;
	}
	if (p[3] & 1)
		unparseable = true;
	// the coded number: a sample number in 36 bits or a frame number in 31 (bitreader.c:999,1054)
	h->variable = (p[1] & 1) || (si && si->min_blocksize != si->max_blocksize);
	unsigned hp = 4;
	if (hp >= avail)
		return HDR_SHORT;
	const unsigned x = p[hp++];
	FLAC__uint64 v;
	unsigned extra;
	if (!(x & 0x80)) { v = x; extra = 0; }
	else if ((x & 0xc0) && !(x & 0x20)) { v = x & 0x1f; extra = 1; }
	else if ((x & 0xe0) && !(x & 0x10)) { v = x & 0x0f; extra = 2; }
	else if ((x & 0xf0) && !(x & 0x08)) { v = x & 0x07; extra = 3; }
	else if ((x & 0xf8) && !(x & 0x04)) { v = x & 0x03; extra = 4; }
	else if ((x & 0xfc) && !(x & 0x02)) { v = x & 0x01; extra = 5; }
	else if (h->variable && (x & 0xfe) && !(x & 0x01)) { v = 0; extra = 6; }
	else {
		*consumed = hp - 1;  // put back
		return HDR_BAD;
	}
	for (unsigned i = 0; i < extra; i++) {
		if (hp >= avail)
			return HDR_SHORT;
		const unsigned b = p[hp++];
		if ((b & 0xc0) != 0x80) {
			*consumed = hp - 1;
			return HDR_BAD;
		}
		v = (v << 6) | (b & 0x3f);
	}
	h->number = v;
	if (bs_code == 1) h->blocksize = 192;
	else if (bs_code >= 2 && bs_code <= 5) h->blocksize = 576u << (bs_code - 2);
	else if (bs_code == 6) {
		if (hp + 1 > avail) return HDR_SHORT;
		h->blocksize = p[hp] + 1u;
		hp += 1;
	} else if (bs_code == 7) {
		if (hp + 2 > avail) return HDR_SHORT;
		h->blocksize = (((unsigned)p[hp] << 8) | p[hp + 1]) + 1u;
		hp += 2;
	} else if (bs_code >= 8) h->blocksize = 256u << (bs_code - 8);
	if (sr_code == 12) {
		if (hp + 1 > avail) return HDR_SHORT;
		h->sample_rate = p[hp] * 1000u;
		hp += 1;
	} else if (sr_code == 13 || sr_code == 14) {
		if (hp + 2 > avail) return HDR_SHORT;
		const unsigned r = ((unsigned)p[hp] << 8) | p[hp + 1];
		h->sample_rate = sr_code == 13 ? r : r * 10u;
		hp += 2;
	}
	if (hp + 1 > avail)
		return HDR_SHORT;
	*consumed = hp + 1;
	if (crc8(p, hp) != p[hp])
		return HDR_BAD;
	h->crc = p[hp];
	h->length = hp + 1;
	return unparseable ? HDR_UNPARSEABLE : HDR_VALID;
}

// frame_sync_ + read_frame_header_ (stream_decoder.c:1941,2141) over in[from,to): bytes in which
// the device found no frame.  Reports what the reference's byte-serial hunt reports there: one
// LOST_SYNC per hunt that has to skip bytes, BAD_HEADER / UNPARSEABLE_STREAM for sync codes whose
// header does not hold up.
void report_gap(FLAC__StreamDecoder *d, size_t from, size_t to)
{
	FLAC__StreamDecoderPrivate *q = d->private_;
	const FLAC__byte *b = q->in + q->batch_base;
	const FLAC__StreamMetadata_StreamInfo *si = q->has_stream_info ? &q->stream_info.data.stream_info : 0;
	bool first = !q->gap_reported;
	size_t pos = from;
	while (pos < to) {
		if (b[pos] == 0xff && pos + 1 < to && (b[pos + 1] >> 1) == 0x7c) {
			HostHeader h;
			unsigned consumed = 0;
			const HeaderVerdict v = classify_header(b + pos, to - pos, si, &h, &consumed);
			if (v == HDR_SHORT)
				break;
			// a header that parses yet was not decoded uses stream parameters the device decoder
			// was not set up for
			send_error(d, v == HDR_BAD ? FLAC__STREAM_DECODER_ERROR_STATUS_BAD_HEADER : FLAC__STREAM_DECODER_ERROR_STATUS_UNPARSEABLE_STREAM);
			pos += consumed;
			first = true;
			continue;
		}
		if (first) {
			send_error(d, FLAC__STREAM_DECODER_ERROR_STATUS_LOST_SYNC);
			first = false;
		}
		pos++;
	}
	q->gap_reported = !first;
}

void md5_worker(Md5Lane *m)
{
	for (;;) {
		Md5Lane::Job job;
		{
			std::unique_lock<std::mutex> lk(m->mu);
			m->cv.wait(lk, [&] { return m->quit || !m->jobs.empty(); });
			if (m->jobs.empty())
				return;
			job = m->jobs.front();
			m->jobs.pop_front();
		}
		if (job.channels) {
			const FLAC__int32 *pl[FLAC__MAX_CHANNELS];
			for (unsigned c = 0; c < job.channels; c++)
				pl[c] = job.p + (size_t)c * job.count;
			fa_md5_accumulate_planes(m->md5, pl, job.channels, job.count, job.bytes_per_sample);
		} else
			fa_md5_accumulate_interleaved(m->md5, job.p, job.count, job.bytes_per_sample);
		{
			std::lock_guard<std::mutex> lk(m->mu);
			m->finished++;
		}
		m->cv.notify_all();
	}
}

// the samples stay where they are until md5_drain has returned
void md5_post(FLAC__StreamDecoderPrivate *q, const FLAC__int32 *p, size_t count, unsigned bytes_per_sample, unsigned channels)
{
	Md5Lane *m = q->md5_lane;
	if (!m->started && !m->failed) {
		m->md5 = &q->md5;
		m->quit = false;
		try {
			m->thread = std::thread(md5_worker, m);
			m->started = true;
		} catch (...) {
			m->failed = true;   // no thread to be had: the digest is worked out here
		}
	}
	if (!m->started) {
		if (channels) {
			const FLAC__int32 *pl[FLAC__MAX_CHANNELS];
			for (unsigned c = 0; c < channels; c++)
				pl[c] = p + (size_t)c * count;
			fa_md5_accumulate_planes(&q->md5, pl, channels, count, bytes_per_sample);
		} else
			fa_md5_accumulate_interleaved(&q->md5, p, count, bytes_per_sample);
		return;
	}
	{
		std::lock_guard<std::mutex> lk(m->mu);
		m->jobs.push_back(Md5Lane::Job{p, count, bytes_per_sample, channels});
		m->posted++;
	}
	m->cv.notify_all();
}

// everything posted has gone into the digest
void md5_drain(FLAC__StreamDecoderPrivate *q)
{
	Md5Lane *m = q->md5_lane;
	if (!m || !m->started)
		return;
	std::unique_lock<std::mutex> lk(m->mu);
	m->cv.wait(lk, [&] { return m->finished == m->posted; });
}

void md5_stop(FLAC__StreamDecoderPrivate *q)
{
	Md5Lane *m = q->md5_lane;
	if (!m || !m->started)
		return;
	md5_drain(q);
	{
		std::lock_guard<std::mutex> lk(m->mu);
		m->quit = true;
	}
	m->cv.notify_all();
	m->thread.join();
	m->started = false;
}

void release_device(FLAC__StreamDecoderPrivate *q)
{
	md5_drain(q);   // (its jobs point into h_out)
	if (q->dev)
		flacamd_decoder_destroy(q->dev);
	q->dev = 0;
	if (q->d_in) (void)hipFree(q->d_in);
	if (q->d_out) (void)hipFree(q->d_out);
	if (q->h_out) (void)hipHostFree(q->h_out);
	q->d_in = 0;
	q->d_out = q->h_out = 0;
	for (unsigned c = 0; c < FLAC__MAX_CHANNELS; c++) {
		free(q->planes[c]);
		q->planes[c] = 0;
	}
	q->dev_channels = q->dev_bps = q->dev_blocksize = 0;
}

bool device_fail(FLAC__StreamDecoder *d, const char *what)
{
	fprintf(stderr, "flac_amd: %s: %s\n", what, flacamd_last_error());
	d->protected_->state = FLAC__STREAM_DECODER_MEMORY_ALLOCATION_ERROR;
	return false;
}

// Stream parameters for the device decoder: STREAMINFO when there is one, else the first frame
// header in the buffer (a stream decoded from its middle, stream_decoder.c:2231-2234,2431-2438).
bool ensure_device(FLAC__StreamDecoder *d)
{
	FLAC__StreamDecoderPrivate *q = d->private_;
	unsigned ch = 0, bps = 0, bs = 0, scale = 0;
	if (q->has_stream_info) {
		const FLAC__StreamMetadata_StreamInfo &s = q->stream_info.data.stream_info;
		ch = s.channels;
		bps = s.bits_per_sample;
		bs = s.max_blocksize;
		scale = s.min_blocksize == s.max_blocksize ? s.min_blocksize : 1;
	}
	if (ch == 0 || bps == 0 || bs == 0) {
		HostHeader h;
		bool found = false;
		unsigned consumed;
		for (size_t p = q->in_rd; p + 6 <= q->in_wr && !found; p++)
			if (q->in[p] == 0xff && (q->in[p + 1] >> 1) == 0x7c && classify_header(q->in + p, q->in_wr - p, 0, &h, &consumed) == HDR_VALID)
				found = true;
		if (!found)
			return true;  // nothing to decode yet; the caller deals with the bytes
		ch = h.channels;
		bps = h.bits_per_sample;
		bs = h.blocksize;
		scale = h.blocksize;
	}
	if (q->dev && q->dev_channels == ch && q->dev_bps == bps && q->dev_blocksize == bs)
		return true;
	release_device(q);
	size_t target = (size_t)32 << 20;  // samples (all channels) per device call
	if (const char *env = getenv("FLAC_AMD_DEC_BATCH_SAMPLES")) {
		const size_t v = (size_t)strtoull(env, 0, 10);
		if (v)
			target = v;
	}
	size_t frames = target / ((size_t)ch * bs);
	if (frames < 8) frames = 8;
	if (frames > 16384) frames = 16384;
	q->max_frames = (uint32_t)frames;
	const size_t worst_frame = (size_t)bs * ch * ((bps + 7) / 8 + 1) + 64 * ch + 32;
	q->batch_bytes = frames * ((size_t)bs * ch * ((bps + 7) / 8) + 16 * ch + 16);
	if (q->batch_bytes < 4 * worst_frame)
		q->batch_bytes = 4 * worst_frame;
	q->out_cap_wide = (uint64_t)frames * bs;
	if (flacamd_decoder_create(ch, bps, bs, fa_stream_device(), q->batch_bytes, q->max_frames, &q->dev) != FLACAMD_OK)
		return device_fail(d, "cannot create the device decoder");
	(void)flacamd_decoder_set_frame_number_scale(q->dev, scale);
	// FLAC_AMD_DEC_SUBFRAMES=0: FLAC__Frame.subframes[] stay zero (saves the records' way back: 528 bytes per subframe)
	{
		const char *env = getenv("FLAC_AMD_DEC_SUBFRAMES");
		if (!(env && !strcmp(env, "0")) && flacamd_decoder_enable_subframe_info(q->dev, 1) != FLACAMD_OK)
			return device_fail(d, "cannot create the device decoder");
		q->want_subinfo = !(env && !strcmp(env, "0"));
		// FLAC_AMD_DEC_PLANAR=0: interleaved samples from the device, taken apart per frame on the host
		const char *pl = getenv("FLAC_AMD_DEC_PLANAR");
		q->planar = !(pl && !strcmp(pl, "0"));
		if (q->planar && flacamd_decoder_set_planar_output(q->dev, 1) != FLACAMD_OK)
			return device_fail(d, "cannot create the device decoder");
		q->rice_words->assign(((size_t)2 << FLAC__MAX_RICE_PARTITION_ORDER) * FLAC__MAX_CHANNELS, 0u);
		q->no_residual->assign((size_t)bs + 1, 0);
	}
	const size_t out_bytes = (size_t)q->out_cap_wide * ch * sizeof(FLAC__int32);
	if (hipMalloc((void **)&q->d_in, q->batch_bytes + 16) != hipSuccess || hipMalloc((void **)&q->d_out, out_bytes) != hipSuccess ||
	    hipHostMalloc((void **)&q->h_out, out_bytes, hipHostMallocDefault) != hipSuccess) {
		release_device(q);
		fprintf(stderr, "flac_amd: cannot allocate the decoder's device buffers\n");
		d->protected_->state = FLAC__STREAM_DECODER_MEMORY_ALLOCATION_ERROR;
		return false;
	}
	for (unsigned c = 0; c < ch; c++) {
		q->planes[c] = (FLAC__int32 *)malloc(sizeof(FLAC__int32) * (bs ? bs : 1));
		if (!q->planes[c]) {
			release_device(q);
			return alloc_fail(d);
		}
		q->plane_ptrs[c] = q->planes[c];
	}
	q->dev_channels = ch;
	q->dev_bps = bps;
	q->dev_blocksize = bs;
	// move the gathered input into a page-locked buffer of the batch size
	FLAC__byte *nb = 0;
	if (hipHostMalloc((void **)&nb, q->batch_bytes + 16, hipHostMallocDefault) != hipSuccess) {
		release_device(q);
		fprintf(stderr, "flac_amd: cannot allocate the decoder's input buffer\n");
		d->protected_->state = FLAC__STREAM_DECODER_MEMORY_ALLOCATION_ERROR;
		return false;
	}
	size_t have = q->in_wr - q->in_rd;
	if (have > q->batch_bytes)
		have = q->batch_bytes;  // cannot happen: the metadata reader keeps far less
	memcpy(nb, q->in + q->in_rd, have);
	release_input(q);
	q->in = nb;
	q->in_pinned = true;
	q->in_cap = q->batch_bytes;
	q->in_rd = 0;
	q->in_wr = have;
	return true;
}

void end_batch(FLAC__StreamDecoderPrivate *q)
{
	q->batch_active = false;
	q->table->clear();
	q->cursor = 0;
}

// Gather input, then decode every frame in it.  True: a batch is active (possibly with no frames).
// False: no batch; the state says why (END_OF_STREAM with nothing left, ABORTED, an error).
bool start_batch(FLAC__StreamDecoder *d)
{
	FLAC__StreamDecoderPrivate *q = d->private_;
	const FLAC__StreamDecoderState entry_state = d->protected_->state;
	compact_input(q);
	const size_t start_fill = q->in_wr;
	const double t_enter = now_s();
	double t_setup = 0;
	// read until the buffer holds a batch (or, while seeking, a few frames' worth)
	for (;;) {
		size_t goal = q->in_pinned ? q->in_cap : (size_t)65536;
		if (q->is_seeking && q->in_pinned) {
			// a seek probe needs a frame or two, not a whole batch
			const size_t few = 3 * ((size_t)q->dev_blocksize * q->dev_channels * ((q->dev_bps + 7) / 8 + 1) + 64 * q->dev_channels + 32);
			if (start_fill + few < goal)
				goal = start_fill + few;
		}
		if (q->input_eof || q->in_wr >= goal)
			break;
		if (!reserve_input(d, goal))
			return false;
		size_t bytes = goal - q->in_wr;
		if (!client_read(d, q->in + q->in_wr, &bytes)) {
			if (d->protected_->state != FLAC__STREAM_DECODER_END_OF_STREAM)
				return false;
			q->input_eof = true;
			d->protected_->state = entry_state;
			break;
		}
		q->in_wr += bytes;
		if (!q->in_pinned && q->in_wr >= 16) {
			// first data of the frame phase: set the device up, which also swaps the buffer
			const double t0 = now_s();
			if (!ensure_device(d))
				return false;
			t_setup += now_s() - t0;
		}
	}
	if (q->in_wr == 0 && q->input_eof) {
		d->protected_->state = FLAC__STREAM_DECODER_END_OF_STREAM;
		return false;
	}
	if (!ensure_device(d))
		return false;
	q->table->clear();
	q->cursor = 0;
	q->batch_base = 0;
	q->batch_len = q->in_wr;
	q->batch_eof = q->input_eof;
	q->batch_cut = false;
	q->batch_full = q->in_pinned && q->in_wr >= q->in_cap;
	q->walk_pos = 0;
	q->batch_active = true;
	if (!q->dev || q->batch_len < 6)
		return true;  // no frame header anywhere in the buffer
	(void)hipSetDevice(fa_stream_device());
	const double t_gathered = now_s();
	g_times.setup += t_setup;
	g_times.gather += t_gathered - t_enter - t_setup;
	if (hipMemcpy(q->d_in, q->in, q->batch_len, hipMemcpyHostToDevice) != hipSuccess) {
		fprintf(stderr, "flac_amd: copy to the device failed\n");
		d->protected_->state = FLAC__STREAM_DECODER_ABORTED;
		return false;
	}
	const double t_up = now_s();
	g_times.to_dev += t_up - t_gathered;
	flacamd_stream_desc sd;
	sd.byte_offset = 0;
	sd.byte_length = q->batch_len;
	sd.sample_offset = 0;
	sd.total_samples = 0;
	sd.first_sample = FLACAMD_FIRST_SAMPLE_AUTO;
	if (flacamd_decode_streams(q->dev, q->d_in, q->batch_len, &sd, 1, q->d_out, 4, q->out_cap_wide, 0) != FLACAMD_OK)
		return device_fail(d, "decode failed");
	q->table->resize(q->max_frames);
	uint32_t n = 0;
	int cut = 0;
	if (flacamd_decoder_frames(q->dev, q->table->data(), q->max_frames, &n, &cut) != FLACAMD_OK)
		return device_fail(d, "decode failed");
	q->table->resize(n);
	q->batch_cut = cut != 0;
	q->subinfo->clear();
	if (q->want_subinfo && n) {
		q->subinfo->resize((size_t)n * q->dev_channels);
		if (flacamd_decoder_subframe_info(q->dev, 0, n, q->subinfo->data()) != FLACAMD_OK)
			return device_fail(d, "decode failed");
	}
	const double t_dec = now_s();
	g_times.decode += t_dec - t_up;
	// bring back the span of samples the good frames cover
	uint64_t used = 0;
	if (n) {
		const uint64_t s0 = (*q->table)[0].sample;
		for (uint32_t i = 0; i < n; i++) {
			const flacamd_frame_info &f = (*q->table)[i];
			if (f.status != 1 || f.sample < s0)
				continue;
			const uint64_t end = f.sample - s0 + f.blocksize;
			if (end <= q->out_cap_wide && end > used)
				used = end;
		}
	}
	md5_drain(q);   // (the digest thread may still be reading the batch before this one)
	if (used && hipMemcpy(q->h_out, q->d_out, (size_t)used * q->dev_channels * sizeof(FLAC__int32), hipMemcpyDeviceToHost) != hipSuccess) {
		fprintf(stderr, "flac_amd: copy from the device failed\n");
		d->protected_->state = FLAC__STREAM_DECODER_ABORTED;
		return false;
	}
	g_times.from_dev += now_s() - t_dec;
	return true;
}

// FLAC__Frame.subframes[] of the frame in hand from the records the device kept (what read_subframe_constant_ /
// _fixed_ / _lpc_ / _verbatim_, stream_decoder.c:2520-2713, leave in the frame)
void fill_subframes(FLAC__StreamDecoderPrivate *q, const flacamd_subframe_info *rec, unsigned channels)
{
	for (unsigned c = 0; c < channels; c++) {
		const flacamd_subframe_info &r = rec[c];
		FLAC__Subframe &sf = q->frame.subframes[c];
		sf.wasted_bits = r.wasted_bits;
		FLAC__EntropyCodingMethod *ecm = 0;
		switch (r.type) {
		case 0:
			sf.type = FLAC__SUBFRAME_TYPE_CONSTANT;
			sf.data.constant.value = r.constant;
			break;
		case 1:
			sf.type = FLAC__SUBFRAME_TYPE_VERBATIM;
			sf.data.verbatim.data = q->no_residual->data();
			break;
		case 2:
			sf.type = FLAC__SUBFRAME_TYPE_FIXED;
			sf.data.fixed.order = r.order;
			for (unsigned i = 0; i < r.order && i < FLAC__MAX_FIXED_ORDER; i++)
				sf.data.fixed.warmup[i] = r.warmup[i];
			sf.data.fixed.residual = q->no_residual->data();
			ecm = &sf.data.fixed.entropy_coding_method;
			break;
		default:
			sf.type = FLAC__SUBFRAME_TYPE_LPC;
			sf.data.lpc.order = r.order;
			sf.data.lpc.qlp_coeff_precision = r.precision;
			sf.data.lpc.quantization_level = r.shift;
			for (unsigned i = 0; i < r.order && i < FLAC__MAX_LPC_ORDER; i++) {
				sf.data.lpc.qlp_coeff[i] = r.qlp_coeff[i];
				sf.data.lpc.warmup[i] = r.warmup[i];
			}
			sf.data.lpc.residual = q->no_residual->data();
			ecm = &sf.data.lpc.entropy_coding_method;
			break;
		}
		if (ecm) {
			ecm->type = r.rice2 ? FLAC__ENTROPY_CODING_METHOD_PARTITIONED_RICE2 : FLAC__ENTROPY_CODING_METHOD_PARTITIONED_RICE;
			ecm->data.partitioned_rice.order = r.partition_order;
			// (room for the 2^15 partitions the format allows, since a reader trusts the order, not capacity_by_order)
			const size_t most = (size_t)1 << FLAC__MAX_RICE_PARTITION_ORDER;
			unsigned *par = q->rice_words->data() + (size_t)c * 2 * most, *raw = par + most;
			const unsigned esc = r.rice2 ? 31u : 15u;
			// (a reader looks at 2^order entries; beyond the 256th there are none here)
			const unsigned nparts = r.partition_order >= 8 ? (unsigned)FLACAMD_SUBFRAME_PARAMS : 1u << r.partition_order;
			for (unsigned i = 0; i < nparts; i++) {
				const bool escaped = (r.params[i] & 0x80u) != 0;
				par[i] = escaped ? esc : r.params[i];
				raw[i] = escaped ? (r.params[i] & 31u) : 0u;
			}
			if (r.partition_order > 8 && r.partition_order <= FLAC__MAX_RICE_PARTITION_ORDER) {
				// a frame cut finer than the records reach: its first 256 partitions are shown, the rest read as parameter 0
				memset(par + nparts, 0, sizeof(unsigned) * (((size_t)1 << r.partition_order) - nparts));
				memset(raw + nparts, 0, sizeof(unsigned) * (((size_t)1 << r.partition_order) - nparts));
			}
			FLAC__EntropyCodingMethod_PartitionedRiceContents &ct = q->rice_contents[c];
			ct.parameters = par;
			ct.raw_bits = raw;
			ct.capacity_by_order = FLAC__MAX_RICE_PARTITION_ORDER;
			ecm->data.partitioned_rice.contents = &ct;
		}
	}
}

// write_audio_frame_to_client_ (stream_decoder.c:2913)
FLAC__StreamDecoderWriteStatus write_audio_frame(FLAC__StreamDecoder *d, const FLAC__int32 *interleaved)
{
	FLAC__StreamDecoderPrivate *q = d->private_;
	const FLAC__Frame *frame = &q->frame;
	if (q->is_seeking) {
		const FLAC__uint64 this_frame_sample = frame->header.number.sample_number;
		const FLAC__uint64 next_frame_sample = this_frame_sample + frame->header.blocksize;
		const FLAC__uint64 target = q->target_sample;
		q->last_frame = *frame;
		if (this_frame_sample <= target && target < next_frame_sample) {
			const unsigned delta = (unsigned)(target - this_frame_sample);
			q->is_seeking = 0;
			if (delta > 0) {
				const FLAC__int32 *shifted[FLAC__MAX_CHANNELS];
				for (unsigned c = 0; c < frame->header.channels; c++)
					shifted[c] = q->plane_ptrs[c] + delta;
				q->last_frame.header.blocksize -= delta;
				q->last_frame.header.number.sample_number += delta;
				return q->write_callback(d, &q->last_frame, shifted, q->client_data);
			}
			return q->write_callback(d, frame, q->plane_ptrs, q->client_data);
		}
		return FLAC__STREAM_DECODER_WRITE_STATUS_CONTINUE;
	}
	if (!q->has_stream_info)
		q->do_md5_checking = 0;
	if (q->do_md5_checking) {
		if (q->planar)
			md5_post(q, interleaved, frame->header.blocksize, (frame->header.bits_per_sample + 7) / 8, frame->header.channels);
		else
			md5_post(q, interleaved, (size_t)frame->header.blocksize * frame->header.channels, (frame->header.bits_per_sample + 7) / 8, 0);
	}
	return q->write_callback(d, frame, q->plane_ptrs, q->client_data);
}

// Hand out the next frame of the stream: frame_sync_ + read_frame_ (stream_decoder.c:1941,1996)
// over the decoded batch.  Returns false on a fatal condition (the state says which).
bool next_frame(FLAC__StreamDecoder *d, bool *got_a_frame, bool do_full_decode)
{
	FLAC__StreamDecoderPrivate *q = d->private_;
	FLAC__StreamDecoderProtected *p = d->protected_;
	*got_a_frame = false;
	for (;;) {
		// stop at the stream's declared length, e.g. before a trailing ID3v1 tag (:1948)
		const FLAC__uint64 total = FLAC__stream_decoder_get_total_samples(d);
		if (total > 0 && q->samples_decoded >= total) {
			p->state = FLAC__STREAM_DECODER_END_OF_STREAM;
			return true;
		}
		if (!q->batch_active && !start_batch(d)) {
			// like frame_sync_ running dry: not an error of this call
			return p->state == FLAC__STREAM_DECODER_END_OF_STREAM || p->state == FLAC__STREAM_DECODER_ABORTED;
		}
		if (q->cursor >= q->table->size()) {
			// no more headers in this batch
			const bool final_batch = q->batch_eof && !q->batch_cut;
			size_t keep_from = (size_t)q->walk_pos;
			if (final_batch)
				keep_from = q->batch_len;
			else if (!q->batch_cut && q->batch_len > 16 && (!q->dev || q->batch_len - keep_from > q->in_cap / 2)) {
				// a long stretch without a single header is noise, not a frame waiting for its tail
				if (q->batch_len - 16 > keep_from)
					keep_from = q->batch_len - 16;
			}
			if (keep_from > q->walk_pos)
				report_gap(d, (size_t)q->walk_pos, keep_from);
			q->in_rd = q->batch_base + keep_from;
			end_batch(q);
			if (final_batch || (q->input_eof && q->in_rd >= q->in_wr)) {
				p->state = FLAC__STREAM_DECODER_END_OF_STREAM;
				return true;
			}
			continue;
		}
		const flacamd_frame_info f = (*q->table)[q->cursor];
		if (f.pos < q->walk_pos) {
			q->cursor++;  // a sync pattern inside bytes already accounted for
			continue;
		}
		if (f.pos > q->walk_pos) {
			report_gap(d, (size_t)q->walk_pos, (size_t)f.pos);
			q->walk_pos = f.pos;
			q->in_rd = q->batch_base + (size_t)f.pos;
		}
		if (f.status == 4 || f.status == 0) {
			// cut off by the end of the buffer
			if (q->batch_eof && !q->batch_cut) {
				// no tail will come: the reference runs out of input inside read_frame_
				q->in_rd = q->batch_base + q->batch_len;
				end_batch(q);
				p->state = FLAC__STREAM_DECODER_END_OF_STREAM;
				return false;
			}
			if (f.pos == 0 && q->batch_full) {
				// claims to be longer than any frame of this stream can be
				send_error(d, FLAC__STREAM_DECODER_ERROR_STATUS_LOST_SYNC);
				q->gap_reported = true;
				q->walk_pos = f.pos + f.hdr_bytes;
				q->cursor++;
				continue;
			}
			end_batch(q);  // read its tail, decode again from here
			continue;
		}
		const uint64_t s0 = (*q->table)[0].sample;
		if ((f.status == 1 || f.status == 3) && q->cursor > 0 && (f.sample < s0 || f.sample - s0 + f.blocksize > q->out_cap_wide)) {
			end_batch(q);  // its samples lie outside this batch's window: it opens the next batch
			continue;
		}
		if (f.status == 2) {
			// malformed subframe data behind a good header: a reserved subframe type or residual coding
			// method is "unparseable" (:2489,2500,2575,2654), anything else lost sync (:2476,2623,2728,2780)
			send_error(d, (f.reserved & 1u) ? FLAC__STREAM_DECODER_ERROR_STATUS_UNPARSEABLE_STREAM : FLAC__STREAM_DECODER_ERROR_STATUS_LOST_SYNC);
			// the reference then starts a fresh hunt where the parse stopped, which reports the bytes
			// it has to skip on its own account
			q->gap_reported = false;
			q->walk_pos = f.pos + f.hdr_bytes;
			q->in_rd = q->batch_base + (size_t)q->walk_pos;
			q->cursor++;
			continue;
		}
		// a whole frame, CRC-16 good (1) or bad (3)
		HostHeader h;
		unsigned consumed;
		const FLAC__byte *src = q->in + q->batch_base + f.pos;
		const FLAC__StreamMetadata_StreamInfo *si = q->has_stream_info ? &q->stream_info.data.stream_info : 0;
		if (classify_header(src, (size_t)(f.end_byte - f.pos), si, &h, &consumed) != HDR_VALID) {
			send_error(d, FLAC__STREAM_DECODER_ERROR_STATUS_UNPARSEABLE_STREAM);
			q->gap_reported = false;
			q->walk_pos = f.pos + f.hdr_bytes;
			q->in_rd = q->batch_base + (size_t)q->walk_pos;
			q->cursor++;
			continue;
		}
		q->gap_reported = false;
		p->state = FLAC__STREAM_DECODER_READ_FRAME;
		memset(&q->frame, 0, sizeof q->frame);
		FLAC__FrameHeader &fh = q->frame.header;
		fh.blocksize = f.blocksize;
		fh.sample_rate = h.sample_rate;
		fh.channels = h.channels;
		fh.channel_assignment = h.assignment;
		fh.bits_per_sample = h.bits_per_sample;
		fh.number_type = FLAC__FRAME_NUMBER_TYPE_SAMPLE_NUMBER;
		fh.crc = h.crc;
		// the frame number becomes a sample number by the stream's fixed blocksize (:2416-2439)
		unsigned next_fixed_block_size = 0;
		if (h.variable)
			fh.number.sample_number = h.number;
		else if (q->fixed_block_size)
			fh.number.sample_number = (FLAC__uint64)q->fixed_block_size * h.number;
		else if (si) {
			fh.number.sample_number = (FLAC__uint64)si->min_blocksize * h.number;
			next_fixed_block_size = si->max_blocksize;
		} else if (h.number == 0) {
			fh.number.sample_number = 0;
			next_fixed_block_size = h.blocksize;
		} else
			fh.number.sample_number = (FLAC__uint64)h.blocksize * h.number;
		q->frame.footer.crc = (FLAC__uint16)(((unsigned)src[f.end_byte - f.pos - 2] << 8) | src[f.end_byte - f.pos - 1]);
		if (q->subinfo->size() >= (q->cursor + 1) * (size_t)h.channels)
			fill_subframes(q, &(*q->subinfo)[q->cursor * (size_t)h.channels], h.channels);
		FLAC__int32 *inter = q->h_out + (size_t)(f.sample - s0) * h.channels;
		if (f.status == 3) {
			// :2105-2113
			send_error(d, FLAC__STREAM_DECODER_ERROR_STATUS_FRAME_CRC_MISMATCH);
			if (do_full_decode) {
				md5_drain(q);   // (a frame with the same number may have been posted from these very samples)
				memset(inter, 0, sizeof(FLAC__int32) * (size_t)f.blocksize * h.channels);
			}
		}
		if (do_full_decode && q->planar) {
			for (unsigned c = 0; c < h.channels; c++)
				q->plane_ptrs[c] = inter + (size_t)c * f.blocksize;
		} else if (do_full_decode) {
			const unsigned ch = h.channels, bs = f.blocksize;
			if (ch == 1)
				memcpy(q->planes[0], inter, sizeof(FLAC__int32) * bs);
			else if (ch == 2) {
				FLAC__int32 *a = q->planes[0], *b = q->planes[1];
				for (unsigned i = 0; i < bs; i++) {
					a[i] = inter[2 * i];
					b[i] = inter[2 * i + 1];
				}
			} else
				for (unsigned c = 0; c < ch; c++) {
					FLAC__int32 *dst = q->planes[c];
					for (unsigned i = 0; i < bs; i++)
						dst[i] = inter[(size_t)i * ch + c];
				}
		}
		q->walk_pos = f.end_byte;
		q->in_rd = q->batch_base + (size_t)f.end_byte;
		q->cursor++;
		*got_a_frame = true;
		if (next_fixed_block_size)
			q->fixed_block_size = next_fixed_block_size;
		// :2121-2129
		p->channels = fh.channels;
		p->channel_assignment = fh.channel_assignment;
		p->bits_per_sample = fh.bits_per_sample;
		p->sample_rate = fh.sample_rate;
		p->blocksize = fh.blocksize;
		q->samples_decoded = fh.number.sample_number + fh.blocksize;
		if (do_full_decode && write_audio_frame(d, inter) != FLAC__STREAM_DECODER_WRITE_STATUS_CONTINUE)
			return false;
		p->state = FLAC__STREAM_DECODER_SEARCH_FOR_FRAME_SYNC;
		return true;
	}
}

// Seeking (what seek_to_absolute_sample_, stream_decoder.c:2973, does for the reference), built on
// what this decoder has anyway: whatever stretch of the stream is read next is decoded as a window
// of whole frames whose sample numbers come back in the frame table, and a frame seen while
// is_seeking is set is inspected by write_audio_frame instead of delivered.  Like the reference the
// hunt keeps a bracket: [lo_pos, lo_sample] is a place where reading finds a frame that starts at or
// before the target, [hi_pos, hi_sample] one where it finds a frame beyond it (or the end of the
// stream); the seek table and STREAMINFO give the first bracket, every probe narrows it, and no
// probe is ever aimed outside it, so the hunt cannot run in circles whatever the local density.
//   1. aim - inside the bracket, by interpolation between its ends (a little in front of where the
//      target would be), every third time by bisection so that a stream whose density changes
//      abruptly (silence, then noise) still closes the bracket geometrically;
//   2. look - read from there; every frame that ends at or before the target is walked past and
//      raises the lower end, the frame that holds the target ends the hunt (its tail is delivered);
//      when the target is still many frames ahead, aim again instead of walking there;
//   3. overshoot - the first frame seen starts beyond the target: this place becomes the upper end;
//   4. once the bracket is a few dozen frames wide, start at its lower end and only walk.
bool seek_to_absolute_sample(FLAC__StreamDecoder *d, FLAC__uint64 stream_length, FLAC__uint64 target)
{
	FLAC__StreamDecoderPrivate *q = d->private_;
	FLAC__StreamDecoderProtected *p = d->protected_;
	const FLAC__StreamMetadata_StreamInfo &si = q->stream_info.data.stream_info;
	const FLAC__uint64 origin = q->first_frame_offset;
	const FLAC__uint64 total = FLAC__stream_decoder_get_total_samples(d);
	const unsigned channels = p->channels ? p->channels : si.channels;
	const unsigned bps = p->bits_per_sample ? p->bits_per_sample : si.bits_per_sample;
	const unsigned block = si.max_blocksize ? si.max_blocksize : 4096;
	auto fail = [&]() {
		p->state = FLAC__STREAM_DECODER_SEEK_ERROR;
		return false;
	};
	if (stream_length <= origin)
		return fail();

	double density;  // bytes of stream per sample, for the first aim and where the bracket has no upper sample number
	if (total > 0)
		density = (double)(stream_length - origin) / (double)total;
	else if (si.max_framesize > 0)
		density = 0.5 * ((double)si.min_framesize + (double)si.max_framesize) / (double)block;
	else
		density = (double)channels * (double)bps / 8.0;
	// the most bytes a frame can take: what STREAMINFO says, else every sample verbatim
	const double frame_bytes = (si.max_framesize > 0 ? (double)si.max_framesize : (double)channels * ((double)bps + 1.0) / 8.0 * (double)block + 32.0) + 64.0;

	// the first bracket: the whole stream, narrowed by the seek points either side of the target
	FLAC__uint64 lo_pos = origin, lo_sample = 0;
	FLAC__uint64 hi_pos = stream_length, hi_sample = total;  // hi_sample 0: not known
	if (q->has_seek_table) {
		const FLAC__StreamMetadata_SeekTable &st = q->seek_table.data.seek_table;
		for (unsigned i = 0; i < st.num_points; i++) {
			const FLAC__StreamMetadata_SeekPoint &pt = st.points[i];
			if (pt.sample_number == FLAC__STREAM_METADATA_SEEKPOINT_PLACEHOLDER || pt.frame_samples == 0)
				continue;
			if (total > 0 && pt.sample_number >= total)
				continue;
			const FLAC__uint64 at = origin + pt.stream_offset;
			if (at >= stream_length)
				continue;
			if (pt.sample_number <= target) {
				if (pt.sample_number >= lo_sample && at >= lo_pos) {
					lo_sample = pt.sample_number;
					lo_pos = at;
				}
			} else if (at < hi_pos && at > lo_pos) {
				hi_pos = at;
				hi_sample = pt.sample_number;
			}
		}
		if (hi_pos <= lo_pos) {  // a table that contradicts itself is not used
			lo_pos = origin;
			lo_sample = 0;
			hi_pos = stream_length;
			hi_sample = total;
		}
	}

	q->target_sample = target;
	const double near = 48.0 * frame_bytes;  // a bracket this narrow is walked through
	bool walk_only = false;
	for (unsigned attempt = 0; attempt < 96; attempt++) {
		// 1. aim
		double aim;
		if (walk_only || (double)(hi_pos - lo_pos) <= near) {
			// 4. (the lower end is always a place where a frame starts: the first frame, a seek point, or the byte
			// behind a frame that was walked past)
			aim = (double)lo_pos;
			walk_only = true;
		} else if (attempt % 3 == 2)
			aim = 0.5 * ((double)lo_pos + (double)hi_pos);
		else {
			double est;
			if (hi_sample > lo_sample && target < hi_sample)
				est = (double)(target - lo_sample) / (double)(hi_sample - lo_sample) * (double)(hi_pos - lo_pos);
			else
				est = (double)(target - lo_sample) * density;
			aim = (double)lo_pos + est - 1.5 * frame_bytes;
			if (aim >= (double)hi_pos - frame_bytes)
				aim = 0.5 * ((double)lo_pos + (double)hi_pos);
		}
		if (aim < (double)lo_pos)
			aim = (double)lo_pos;
		FLAC__uint64 pos = (FLAC__uint64)aim;
		if (pos >= stream_length)
			pos = stream_length - 1;
		if (q->seek_callback(d, pos, q->client_data) != FLAC__STREAM_DECODER_SEEK_STATUS_OK)
			return fail();
		if (!FLAC__stream_decoder_flush(d))
			return false;
		q->unparseable_frame_count = 0;
		// 2. look
		bool first_here = true;
		for (;;) {
			q->samples_decoded = 0;
			if (!FLAC__stream_decoder_process_single(d))
				return fail();
			if (!q->is_seeking)
				return true;  // write_audio_frame found the target in this frame and delivered its tail
			if (p->state == FLAC__STREAM_DECODER_END_OF_STREAM || q->samples_decoded == 0) {
				// ran off the end without meeting the target: nothing from here on holds it
				if (!first_here || pos <= origin)
					return fail();
				if (pos <= lo_pos) {
					// a lower end that came from a seek point that is not true: fall back on the stream's start
					lo_pos = origin;
					lo_sample = 0;
					walk_only = false;
				}
				hi_pos = pos;
				break;
			}
			const FLAC__uint64 first = q->last_frame.header.number.sample_number;
			const FLAC__uint64 end = first + q->last_frame.header.blocksize;
			if (target < first) {
				// 3. overshoot
				if (!first_here || pos <= origin)
					return fail();  // the stream's frames do not hold the target
				if (pos <= lo_pos) {
					lo_pos = origin;
					lo_sample = 0;
					walk_only = false;
				}
				hi_pos = pos;
				hi_sample = first;
				break;
			}
			first_here = false;
			// the target lies ahead (end <= target): the next frame starts here
			FLAC__uint64 here;
			if (!FLAC__stream_decoder_get_decode_position(d, &here)) {
				walk_only = true;  // no position to aim from: walk on from here
				continue;
			}
			if (here > lo_pos && here < hi_pos) {
				lo_pos = here;
				lo_sample = end;
			}
			if (walk_only)
				continue;
			double ahead;
			if (hi_sample > end && target < hi_sample && hi_pos > here)
				ahead = (double)(target - end) / (double)(hi_sample - end) * (double)(hi_pos - here);
			else
				ahead = (double)(target - end) * density;
			if (ahead > near && (double)(hi_pos - here) > near)
				break;  // aim again, inside the narrowed bracket
		}
		if (hi_pos <= lo_pos)
			return fail();
	}
	return fail();
}

// ---- FILE* callbacks (stream_decoder.c:3324-3387) ----
FLAC__StreamDecoderReadStatus file_read_cb(const FLAC__StreamDecoder *d, FLAC__byte buffer[], size_t *bytes, void *)
{
	if (*bytes > 0) {
		*bytes = fread(buffer, 1, *bytes, d->private_->file);
		if (ferror(d->private_->file))
			return FLAC__STREAM_DECODER_READ_STATUS_ABORT;
		if (*bytes == 0)
			return FLAC__STREAM_DECODER_READ_STATUS_END_OF_STREAM;
		return FLAC__STREAM_DECODER_READ_STATUS_CONTINUE;
	}
	return FLAC__STREAM_DECODER_READ_STATUS_ABORT;
}
FLAC__StreamDecoderSeekStatus file_seek_cb(const FLAC__StreamDecoder *d, FLAC__uint64 off, void *)
{
	if (d->private_->file == stdin)
		return FLAC__STREAM_DECODER_SEEK_STATUS_UNSUPPORTED;
	if (fseeko(d->private_->file, (off_t)off, SEEK_SET) < 0)
		return FLAC__STREAM_DECODER_SEEK_STATUS_ERROR;
	return FLAC__STREAM_DECODER_SEEK_STATUS_OK;
}
FLAC__StreamDecoderTellStatus file_tell_cb(const FLAC__StreamDecoder *d, FLAC__uint64 *off, void *)
{
	if (d->private_->file == stdin)
		return FLAC__STREAM_DECODER_TELL_STATUS_UNSUPPORTED;
	const off_t pos = ftello(d->private_->file);
	if (pos < 0)
		return FLAC__STREAM_DECODER_TELL_STATUS_ERROR;
	*off = (FLAC__uint64)pos;
	return FLAC__STREAM_DECODER_TELL_STATUS_OK;
}
FLAC__StreamDecoderLengthStatus file_length_cb(const FLAC__StreamDecoder *d, FLAC__uint64 *len, void *)
{
	if (d->private_->file == stdin)
		return FLAC__STREAM_DECODER_LENGTH_STATUS_UNSUPPORTED;
	const off_t cur = ftello(d->private_->file);
	if (cur < 0 || fseeko(d->private_->file, 0, SEEK_END) < 0)
		return FLAC__STREAM_DECODER_LENGTH_STATUS_ERROR;
	const off_t end = ftello(d->private_->file);
	if (fseeko(d->private_->file, cur, SEEK_SET) < 0 || end < 0)
		return FLAC__STREAM_DECODER_LENGTH_STATUS_ERROR;
	*len = (FLAC__uint64)end;
	return FLAC__STREAM_DECODER_LENGTH_STATUS_OK;
}
FLAC__bool file_eof_cb(const FLAC__StreamDecoder *d, void *) { return feof(d->private_->file) ? 1 : 0; }

// init_stream_internal_ (stream_decoder.c:370)
FLAC__StreamDecoderInitStatus init_stream_internal(FLAC__StreamDecoder *d, FLAC__StreamDecoderReadCallback read_cb, FLAC__StreamDecoderSeekCallback seek_cb,
                                                   FLAC__StreamDecoderTellCallback tell_cb, FLAC__StreamDecoderLengthCallback length_cb,
                                                   FLAC__StreamDecoderEofCallback eof_cb, FLAC__StreamDecoderWriteCallback write_cb,
                                                   FLAC__StreamDecoderMetadataCallback metadata_cb, FLAC__StreamDecoderErrorCallback error_cb,
                                                   void *client_data, bool is_ogg)
{
	if (d->protected_->state != FLAC__STREAM_DECODER_UNINITIALIZED)
		return FLAC__STREAM_DECODER_INIT_STATUS_ALREADY_INITIALIZED;
	if (is_ogg)
		return FLAC__STREAM_DECODER_INIT_STATUS_UNSUPPORTED_CONTAINER;  // built without Ogg, as FLAC__HAS_OGG=0
	if (!read_cb || !write_cb || !error_cb || (seek_cb && (!tell_cb || !length_cb || !eof_cb)))
		return FLAC__STREAM_DECODER_INIT_STATUS_INVALID_CALLBACKS;
	FLAC__StreamDecoderPrivate *q = d->private_;
	q->read_callback = read_cb;
	q->seek_callback = seek_cb;
	q->tell_callback = tell_cb;
	q->length_callback = length_cb;
	q->eof_callback = eof_cb;
	q->write_callback = write_cb;
	q->metadata_callback = metadata_cb;
	q->error_callback = error_cb;
	q->client_data = client_data;
	q->samples_decoded = 0;
	q->has_stream_info = 0;
	q->cached = 0;
	q->do_md5_checking = d->protected_->md5_checking;
	q->is_seeking = 0;
	q->internal_reset_hack = 1;
	if (!FLAC__stream_decoder_reset(d))
		return FLAC__STREAM_DECODER_INIT_STATUS_MEMORY_ALLOCATION_ERROR;
	return FLAC__STREAM_DECODER_INIT_STATUS_OK;
}

// init_FILE_internal_ / init_file_internal_ (stream_decoder.c:534,602).  On the two early
// refusals the reference stores the init status into the decoder's state field; kept.
FLAC__StreamDecoderInitStatus init_FILE_internal(FLAC__StreamDecoder *d, FILE *file, FLAC__StreamDecoderWriteCallback write_cb,
                                                 FLAC__StreamDecoderMetadataCallback metadata_cb, FLAC__StreamDecoderErrorCallback error_cb, void *client_data,
                                                 bool is_ogg)
{
	if (d->protected_->state != FLAC__STREAM_DECODER_UNINITIALIZED) {
		d->protected_->state = (FLAC__StreamDecoderState)FLAC__STREAM_DECODER_INIT_STATUS_ALREADY_INITIALIZED;
		return FLAC__STREAM_DECODER_INIT_STATUS_ALREADY_INITIALIZED;
	}
	if (!write_cb || !error_cb) {
		d->protected_->state = (FLAC__StreamDecoderState)FLAC__STREAM_DECODER_INIT_STATUS_INVALID_CALLBACKS;
		return FLAC__STREAM_DECODER_INIT_STATUS_INVALID_CALLBACKS;
	}
	if (d->private_->file && d->private_->file != stdin && d->private_->file != file)
		fclose(d->private_->file);
	d->private_->file = file;
	const bool is_stdin = file == stdin;
	return init_stream_internal(d, file_read_cb, is_stdin ? 0 : file_seek_cb, is_stdin ? 0 : file_tell_cb, is_stdin ? 0 : file_length_cb, file_eof_cb,
	                            write_cb, metadata_cb, error_cb, client_data, is_ogg);
}

FLAC__StreamDecoderInitStatus init_file_internal(FLAC__StreamDecoder *d, const char *filename, FLAC__StreamDecoderWriteCallback write_cb,
                                                 FLAC__StreamDecoderMetadataCallback metadata_cb, FLAC__StreamDecoderErrorCallback error_cb, void *client_data,
                                                 bool is_ogg)
{
	if (d->protected_->state != FLAC__STREAM_DECODER_UNINITIALIZED) {
		d->protected_->state = (FLAC__StreamDecoderState)FLAC__STREAM_DECODER_INIT_STATUS_ALREADY_INITIALIZED;
		return FLAC__STREAM_DECODER_INIT_STATUS_ALREADY_INITIALIZED;
	}
	if (!write_cb || !error_cb) {
		d->protected_->state = (FLAC__StreamDecoderState)FLAC__STREAM_DECODER_INIT_STATUS_INVALID_CALLBACKS;
		return FLAC__STREAM_DECODER_INIT_STATUS_INVALID_CALLBACKS;
	}
	FILE *file = filename ? fopen(filename, "rb") : stdin;
	if (!file)
		return FLAC__STREAM_DECODER_INIT_STATUS_ERROR_OPENING_FILE;
	return init_FILE_internal(d, file, write_cb, metadata_cb, error_cb, client_data, is_ogg);
}

}  // namespace

extern "C" {

FLAC__StreamDecoder *FLAC__stream_decoder_new(void)
{
	FLAC__StreamDecoder *d = (FLAC__StreamDecoder *)calloc(1, sizeof *d);
	if (!d)
		return 0;
	d->protected_ = (FLAC__StreamDecoderProtected *)calloc(1, sizeof *d->protected_);
	d->private_ = (FLAC__StreamDecoderPrivate *)calloc(1, sizeof *d->private_);
	if (d->private_) {
		d->private_->filter_ids = new (std::nothrow) std::vector<FLAC__byte>();
		d->private_->table = new (std::nothrow) std::vector<flacamd_frame_info>();
		d->private_->subinfo = new (std::nothrow) std::vector<flacamd_subframe_info>();
		d->private_->rice_words = new (std::nothrow) std::vector<unsigned>();
		d->private_->no_residual = new (std::nothrow) std::vector<FLAC__int32>();
		d->private_->md5_lane = new (std::nothrow) Md5Lane();
	}
	if (!d->protected_ || !d->private_ || !d->private_->filter_ids || !d->private_->table || !d->private_->subinfo || !d->private_->rice_words ||
	    !d->private_->no_residual || !d->private_->md5_lane) {
		if (d->private_) {
			delete d->private_->filter_ids;
			delete d->private_->table;
			delete d->private_->subinfo;
			delete d->private_->rice_words;
			delete d->private_->no_residual;
			delete d->private_->md5_lane;
		}
		free(d->private_);
		free(d->protected_);
		free(d);
		return 0;
	}
	set_defaults(d);
	d->protected_->state = FLAC__STREAM_DECODER_UNINITIALIZED;
	return d;
}

void FLAC__stream_decoder_delete(FLAC__StreamDecoder *d)
{
	(void)FLAC__stream_decoder_finish(d);
	FLAC__StreamDecoderPrivate *q = d->private_;
	release_device(q);
	release_input(q);
	if (q->file && q->file != stdin)
		fclose(q->file);  // left behind by an init that was refused after the file was opened
	free(q->seek_table.data.seek_table.points);
	delete q->filter_ids;
	delete q->table;
	delete q->subinfo;
	delete q->rice_words;
	delete q->no_residual;
	md5_stop(q);
	delete q->md5_lane;
	free(d->private_);
	free(d->protected_);
	free(d);
}

FLAC__StreamDecoderInitStatus FLAC__stream_decoder_init_stream(FLAC__StreamDecoder *d, FLAC__StreamDecoderReadCallback read_cb,
                                                               FLAC__StreamDecoderSeekCallback seek_cb, FLAC__StreamDecoderTellCallback tell_cb,
                                                               FLAC__StreamDecoderLengthCallback length_cb, FLAC__StreamDecoderEofCallback eof_cb,
                                                               FLAC__StreamDecoderWriteCallback write_cb, FLAC__StreamDecoderMetadataCallback metadata_cb,
                                                               FLAC__StreamDecoderErrorCallback error_cb, void *client_data)
{
	return init_stream_internal(d, read_cb, seek_cb, tell_cb, length_cb, eof_cb, write_cb, metadata_cb, error_cb, client_data, false);
}

FLAC__StreamDecoderInitStatus FLAC__stream_decoder_init_ogg_stream(FLAC__StreamDecoder *d, FLAC__StreamDecoderReadCallback read_cb,
                                                                   FLAC__StreamDecoderSeekCallback seek_cb, FLAC__StreamDecoderTellCallback tell_cb,
                                                                   FLAC__StreamDecoderLengthCallback length_cb, FLAC__StreamDecoderEofCallback eof_cb,
                                                                   FLAC__StreamDecoderWriteCallback write_cb, FLAC__StreamDecoderMetadataCallback metadata_cb,
                                                                   FLAC__StreamDecoderErrorCallback error_cb, void *client_data)
{
	return init_stream_internal(d, read_cb, seek_cb, tell_cb, length_cb, eof_cb, write_cb, metadata_cb, error_cb, client_data, true);
}

FLAC__StreamDecoderInitStatus FLAC__stream_decoder_init_FILE(FLAC__StreamDecoder *d, FILE *file, FLAC__StreamDecoderWriteCallback write_cb,
                                                             FLAC__StreamDecoderMetadataCallback metadata_cb, FLAC__StreamDecoderErrorCallback error_cb,
                                                             void *client_data)
{
	return init_FILE_internal(d, file, write_cb, metadata_cb, error_cb, client_data, false);
}

FLAC__StreamDecoderInitStatus FLAC__stream_decoder_init_ogg_FILE(FLAC__StreamDecoder *d, FILE *file, FLAC__StreamDecoderWriteCallback write_cb,
                                                                 FLAC__StreamDecoderMetadataCallback metadata_cb, FLAC__StreamDecoderErrorCallback error_cb,
                                                                 void *client_data)
{
	return init_FILE_internal(d, file, write_cb, metadata_cb, error_cb, client_data, true);
}

FLAC__StreamDecoderInitStatus FLAC__stream_decoder_init_file(FLAC__StreamDecoder *d, const char *filename, FLAC__StreamDecoderWriteCallback write_cb,
                                                             FLAC__StreamDecoderMetadataCallback metadata_cb, FLAC__StreamDecoderErrorCallback error_cb,
                                                             void *client_data)
{
	return init_file_internal(d, filename, write_cb, metadata_cb, error_cb, client_data, false);
}

FLAC__StreamDecoderInitStatus FLAC__stream_decoder_init_ogg_file(FLAC__StreamDecoder *d, const char *filename, FLAC__StreamDecoderWriteCallback write_cb,
                                                                 FLAC__StreamDecoderMetadataCallback metadata_cb, FLAC__StreamDecoderErrorCallback error_cb,
                                                                 void *client_data)
{
	return init_file_internal(d, filename, write_cb, metadata_cb, error_cb, client_data, true);
}

// stream_decoder.c:659
FLAC__bool FLAC__stream_decoder_finish(FLAC__StreamDecoder *d)
{
	FLAC__StreamDecoderPrivate *q = d->private_;
	if (d->protected_->state == FLAC__STREAM_DECODER_UNINITIALIZED)
		return 1;
	bool md5_failed = false;
	const double t_fin = now_s();
	md5_drain(q);
	if (getenv("FLAC_AMD_DEC_TIMES")) {
		fprintf(stderr, "flac_amd decoder times: device set-up %.1f ms, gathering input %.1f, to the device %.1f, decode %.1f, from the device %.1f, "
		        "waiting for the digest at the end %.1f\n", 1e3 * g_times.setup, 1e3 * g_times.gather, 1e3 * g_times.to_dev, 1e3 * g_times.decode,
		        1e3 * g_times.from_dev, 1e3 * (now_s() - t_fin));
		g_times = DecTimes();
	}
	fa_md5_final(&q->md5, q->computed_md5sum);
	if (q->has_seek_table && q->seek_table.data.seek_table.points) {
		free(q->seek_table.data.seek_table.points);
		q->seek_table.data.seek_table.points = 0;
		q->has_seek_table = 0;
	}
	end_batch(q);
	release_device(q);
	release_input(q);
	q->input_eof = false;
	if (q->file) {
		if (q->file != stdin)
			fclose(q->file);
		q->file = 0;
	}
	if (q->do_md5_checking && memcmp(q->stream_info.data.stream_info.md5sum, q->computed_md5sum, 16))
		md5_failed = true;
	q->is_seeking = 0;
	set_defaults(d);
	d->protected_->state = FLAC__STREAM_DECODER_UNINITIALIZED;
	return !md5_failed;
}

FLAC__bool FLAC__stream_decoder_set_ogg_serial_number(FLAC__StreamDecoder *d, long)
{
	(void)d;
	return 0;  // no Ogg layer (stream_decoder.c:736-739)
}

FLAC__bool FLAC__stream_decoder_set_md5_checking(FLAC__StreamDecoder *d, FLAC__bool value)
{
	if (d->protected_->state != FLAC__STREAM_DECODER_UNINITIALIZED)
		return 0;
	d->protected_->md5_checking = value;
	return 1;
}

FLAC__bool FLAC__stream_decoder_set_metadata_respond(FLAC__StreamDecoder *d, FLAC__MetadataType type)
{
	if ((unsigned)type > 126)
		return 0;
	if (d->protected_->state != FLAC__STREAM_DECODER_UNINITIALIZED)
		return 0;
	d->private_->metadata_filter[type] = 1;
	if (type == FLAC__METADATA_TYPE_APPLICATION)
		d->private_->filter_ids->clear();
	return 1;
}

FLAC__bool FLAC__stream_decoder_set_metadata_respond_application(FLAC__StreamDecoder *d, const FLAC__byte id[4])
{
	if (d->protected_->state != FLAC__STREAM_DECODER_UNINITIALIZED)
		return 0;
	if (d->private_->metadata_filter[FLAC__METADATA_TYPE_APPLICATION])
		return 1;
	d->private_->filter_ids->insert(d->private_->filter_ids->end(), id, id + 4);
	return 1;
}

FLAC__bool FLAC__stream_decoder_set_metadata_respond_all(FLAC__StreamDecoder *d)
{
	if (d->protected_->state != FLAC__STREAM_DECODER_UNINITIALIZED)
		return 0;
	for (unsigned i = 0; i < 128; i++)
		d->private_->metadata_filter[i] = 1;
	d->private_->filter_ids->clear();
	return 1;
}

FLAC__bool FLAC__stream_decoder_set_metadata_ignore(FLAC__StreamDecoder *d, FLAC__MetadataType type)
{
	if ((unsigned)type > 126)
		return 0;
	if (d->protected_->state != FLAC__STREAM_DECODER_UNINITIALIZED)
		return 0;
	d->private_->metadata_filter[type] = 0;
	if (type == FLAC__METADATA_TYPE_APPLICATION)
		d->private_->filter_ids->clear();
	return 1;
}

FLAC__bool FLAC__stream_decoder_set_metadata_ignore_application(FLAC__StreamDecoder *d, const FLAC__byte id[4])
{
	if (d->protected_->state != FLAC__STREAM_DECODER_UNINITIALIZED)
		return 0;
	if (!d->private_->metadata_filter[FLAC__METADATA_TYPE_APPLICATION])
		return 1;
	d->private_->filter_ids->insert(d->private_->filter_ids->end(), id, id + 4);
	return 1;
}

FLAC__bool FLAC__stream_decoder_set_metadata_ignore_all(FLAC__StreamDecoder *d)
{
	if (d->protected_->state != FLAC__STREAM_DECODER_UNINITIALIZED)
		return 0;
	memset(d->private_->metadata_filter, 0, sizeof d->private_->metadata_filter);
	d->private_->filter_ids->clear();
	return 1;
}

FLAC__StreamDecoderState FLAC__stream_decoder_get_state(const FLAC__StreamDecoder *d) { return d->protected_->state; }
const char *FLAC__stream_decoder_get_resolved_state_string(const FLAC__StreamDecoder *d) { return FLAC__StreamDecoderStateString[d->protected_->state]; }
FLAC__bool FLAC__stream_decoder_get_md5_checking(const FLAC__StreamDecoder *d) { return d->protected_->md5_checking; }
FLAC__uint64 FLAC__stream_decoder_get_total_samples(const FLAC__StreamDecoder *d)
{
	return d->private_->has_stream_info ? d->private_->stream_info.data.stream_info.total_samples : 0;
}
unsigned FLAC__stream_decoder_get_channels(const FLAC__StreamDecoder *d) { return d->protected_->channels; }
FLAC__ChannelAssignment FLAC__stream_decoder_get_channel_assignment(const FLAC__StreamDecoder *d) { return d->protected_->channel_assignment; }
unsigned FLAC__stream_decoder_get_bits_per_sample(const FLAC__StreamDecoder *d) { return d->protected_->bits_per_sample; }
unsigned FLAC__stream_decoder_get_sample_rate(const FLAC__StreamDecoder *d) { return d->protected_->sample_rate; }
unsigned FLAC__stream_decoder_get_blocksize(const FLAC__StreamDecoder *d) { return d->protected_->blocksize; }

// stream_decoder.c:929: the client's position minus what was read ahead of the decode point
FLAC__bool FLAC__stream_decoder_get_decode_position(const FLAC__StreamDecoder *d, FLAC__uint64 *position)
{
	const FLAC__StreamDecoderPrivate *q = d->private_;
	if (!q->tell_callback)
		return 0;
	if (q->tell_callback(d, position, q->client_data) != FLAC__STREAM_DECODER_TELL_STATUS_OK)
		return 0;
	*position -= q->in_wr - q->in_rd;
	return 1;
}

// stream_decoder.c:951
FLAC__bool FLAC__stream_decoder_flush(FLAC__StreamDecoder *d)
{
	FLAC__StreamDecoderPrivate *q = d->private_;
	q->samples_decoded = 0;
	q->do_md5_checking = 0;
	end_batch(q);
	q->in_rd = q->in_wr = 0;
	q->input_eof = false;
	q->cached = 0;
	q->gap_reported = false;
	d->protected_->state = FLAC__STREAM_DECODER_SEARCH_FOR_FRAME_SYNC;
	return 1;
}

// stream_decoder.c:974
FLAC__bool FLAC__stream_decoder_reset(FLAC__StreamDecoder *d)
{
	FLAC__StreamDecoderPrivate *q = d->private_;
	if (!FLAC__stream_decoder_flush(d))
		return 0;
	if (!q->internal_reset_hack) {
		if (q->file == stdin)
			return 0;
		if (q->seek_callback && q->seek_callback(d, 0, q->client_data) == FLAC__STREAM_DECODER_SEEK_STATUS_ERROR)
			return 0;
	} else
		q->internal_reset_hack = 0;
	d->protected_->state = FLAC__STREAM_DECODER_SEARCH_FOR_METADATA;
	q->has_stream_info = 0;
	if (q->has_seek_table && q->seek_table.data.seek_table.points) {
		free(q->seek_table.data.seek_table.points);
		q->seek_table.data.seek_table.points = 0;
		q->has_seek_table = 0;
	}
	q->do_md5_checking = d->protected_->md5_checking;
	md5_drain(q);
	fa_md5_init(&q->md5, fa_md5_default_mode());
	q->fixed_block_size = 0;
	q->first_frame_offset = 0;
	q->unparseable_frame_count = 0;
	return 1;
}

// stream_decoder.c:1034
FLAC__bool FLAC__stream_decoder_process_single(FLAC__StreamDecoder *d)
{
	bool got;
	for (;;) {
		switch (d->protected_->state) {
		case FLAC__STREAM_DECODER_SEARCH_FOR_METADATA:
			if (!find_metadata(d))
				return 0;
			break;
		case FLAC__STREAM_DECODER_READ_METADATA:
			return read_metadata(d) ? 1 : 0;
		case FLAC__STREAM_DECODER_SEARCH_FOR_FRAME_SYNC:
		case FLAC__STREAM_DECODER_READ_FRAME:
			if (!next_frame(d, &got, true))
				return 0;
			if (got)
				return 1;
			break;
		case FLAC__STREAM_DECODER_END_OF_STREAM:
		case FLAC__STREAM_DECODER_ABORTED:
			return 1;
		default:
			return 0;
		}
	}
}

// stream_decoder.c:1071
FLAC__bool FLAC__stream_decoder_process_until_end_of_metadata(FLAC__StreamDecoder *d)
{
	for (;;) {
		switch (d->protected_->state) {
		case FLAC__STREAM_DECODER_SEARCH_FOR_METADATA:
			if (!find_metadata(d))
				return 0;
			break;
		case FLAC__STREAM_DECODER_READ_METADATA:
			if (!read_metadata(d))
				return 0;
			break;
		case FLAC__STREAM_DECODER_SEARCH_FOR_FRAME_SYNC:
		case FLAC__STREAM_DECODER_READ_FRAME:
		case FLAC__STREAM_DECODER_END_OF_STREAM:
		case FLAC__STREAM_DECODER_ABORTED:
			return 1;
		default:
			return 0;
		}
	}
}

// stream_decoder.c:1098
FLAC__bool FLAC__stream_decoder_process_until_end_of_stream(FLAC__StreamDecoder *d)
{
	bool got;
	for (;;) {
		switch (d->protected_->state) {
		case FLAC__STREAM_DECODER_SEARCH_FOR_METADATA:
			if (!find_metadata(d))
				return 0;
			break;
		case FLAC__STREAM_DECODER_READ_METADATA:
			if (!read_metadata(d))
				return 0;
			break;
		case FLAC__STREAM_DECODER_SEARCH_FOR_FRAME_SYNC:
		case FLAC__STREAM_DECODER_READ_FRAME:
			if (!next_frame(d, &got, true))
				return 0;
			break;
		case FLAC__STREAM_DECODER_END_OF_STREAM:
		case FLAC__STREAM_DECODER_ABORTED:
			return 1;
		default:
			return 0;
		}
	}
}

// stream_decoder.c:1132
FLAC__bool FLAC__stream_decoder_skip_single_frame(FLAC__StreamDecoder *d)
{
	bool got;
	for (;;) {
		switch (d->protected_->state) {
		case FLAC__STREAM_DECODER_SEARCH_FOR_METADATA:
		case FLAC__STREAM_DECODER_READ_METADATA:
			return 0;
		case FLAC__STREAM_DECODER_SEARCH_FOR_FRAME_SYNC:
		case FLAC__STREAM_DECODER_READ_FRAME:
			if (!next_frame(d, &got, false))
				return 0;
			if (got)
				return 1;
			break;
		case FLAC__STREAM_DECODER_END_OF_STREAM:
		case FLAC__STREAM_DECODER_ABORTED:
			return 1;
		default:
			return 0;
		}
	}
}

// stream_decoder.c:1163
FLAC__bool FLAC__stream_decoder_seek_absolute(FLAC__StreamDecoder *d, FLAC__uint64 sample)
{
	FLAC__StreamDecoderPrivate *q = d->private_;
	const FLAC__StreamDecoderState s = d->protected_->state;
	if (s != FLAC__STREAM_DECODER_SEARCH_FOR_METADATA && s != FLAC__STREAM_DECODER_READ_METADATA && s != FLAC__STREAM_DECODER_SEARCH_FOR_FRAME_SYNC &&
	    s != FLAC__STREAM_DECODER_READ_FRAME && s != FLAC__STREAM_DECODER_END_OF_STREAM)
		return 0;
	if (!q->seek_callback)
		return 0;
	if (FLAC__stream_decoder_get_total_samples(d) > 0 && sample >= FLAC__stream_decoder_get_total_samples(d))
		return 0;
	q->is_seeking = 1;
	q->do_md5_checking = 0;
	FLAC__uint64 length;
	if (q->length_callback(d, &length, q->client_data) != FLAC__STREAM_DECODER_LENGTH_STATUS_OK) {
		q->is_seeking = 0;
		return 0;
	}
	if (s == FLAC__STREAM_DECODER_SEARCH_FOR_METADATA || s == FLAC__STREAM_DECODER_READ_METADATA) {
		if (!FLAC__stream_decoder_process_until_end_of_metadata(d)) {
			q->is_seeking = 0;
			return 0;
		}
		if (FLAC__stream_decoder_get_total_samples(d) > 0 && sample >= FLAC__stream_decoder_get_total_samples(d)) {
			q->is_seeking = 0;
			return 0;
		}
	}
	const bool ok = seek_to_absolute_sample(d, length, sample);
	q->is_seeking = 0;
	return ok ? 1 : 0;
}

}  // extern "C"
