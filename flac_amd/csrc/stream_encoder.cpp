// libFLAC-compatible stream encoder on top of the device batch ABI.
//
// Mirrors src/libFLAC/stream_encoder.c: settings and their validation, the stream prologue, the
// one-sample look-ahead blocking of process / process_interleaved, per-frame write / tell / seek
// callback sequencing, seek-table fill-in and the finish()-time STREAMINFO rewrite.  Frames
// themselves come from flacamd_encode_frames in batches that are pipelined: while the caller's
// samples are being staged for batch k+1, batch k is on the GPU and batch k-1's frames are coming
// back; the MD5 of the samples runs on a thread of its own.  Callbacks run on the caller's thread.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include <hip/hip_runtime.h>

#include <emmintrin.h>

#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

#include "flac_amd.h"
#include "md5.h"
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
	struct EncPipe *pipe;          // staging slots, streams, helper threads
	uint32_t loose_state;
	// verify error report (stream_encoder.h:1242)
	FLAC__uint64 ve_absolute_sample;
	unsigned ve_frame, ve_channel, ve_sample;
	FLAC__int32 ve_expected, ve_got;
	FLAC__bool is_being_deleted;
};

#define ENC_SLOTS 3

struct EncSlot {
	uint8_t *pin_in, *dev_in;
	uint8_t *pin_out, *dev_out;
	uint32_t *pin_fb, *dev_fb;
	uint64_t *pin_total, *dev_offs;
	hipEvent_t uploaded, computed, done;
	size_t fill;        // wide samples staged so far
	size_t wide;        // wide samples of the submitted batch
	uint32_t nframes;
	uint32_t first_frame;
	uint64_t md5_ticket;
	bool in_flight;
};

struct EncPipe {
	int device;
	unsigned cw;              // container width of the staging buffers, bytes per sample
	unsigned channels, blocksize, bytes_per_sample;
	size_t slot_samples;      // wide samples a slot can stage: (batch_frames + 1) blocks
	size_t in_bytes, out_cap, out_guess;
	EncSlot slot[ENC_SLOTS];
	unsigned filling;         // slot the caller's samples go to
	unsigned oldest;          // oldest slot in flight (frames go out in submission order)
	unsigned in_flight;
	hipStream_t s_in, s_comp, s_out;
	// MD5 of the staged samples, in order, on its own thread (FLAC__MD5Accumulate, stream_encoder.c:2870,
	// is serial work over the whole stream: here it overlaps everything else)
	std::thread md5_thread;
	std::mutex mu;
	std::condition_variable cv;
	struct Md5Job { const uint8_t *p; size_t samples; };
	std::deque<Md5Job> md5_jobs;
	uint64_t md5_posted, md5_finished;
	bool md5_quit, md5_on;
	bool loose_carry;         // batches do not start on loose mid/side decision boundaries
	FaMd5 *md5;
	// helpers that share out the staging of large process() calls
	std::vector<std::thread> copiers;
	struct CopyJob { const int32_t *src; uint8_t *dst; size_t n; };
	std::vector<CopyJob> copy_jobs;
	uint64_t copy_generation;
	unsigned copy_pending;
	bool copy_quit;
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

void pipe_destroy(FLAC__StreamEncoderPrivate *q);

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
	pipe_destroy(q);
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

// ---------------------------------------------------------------------------------------------
// The batch pipeline.  A slot is one batch on its way through: pinned host staging for the samples
// (at container width: int16 for streams of up to 16 bits, which halves what crosses PCIe and puts
// stereo on the kernels' one-word-per-wide-sample paths, else int32), the same on the device, the
// device and pinned host buffers for the frames.  Three slots rotate: one being filled by the
// caller, one on the GPU, one whose frames are coming back.  Three HIP streams keep the roles
// apart - uploads, kernels (which share the encoder handle's work buffers and therefore run in
// order), downloads - chained by events.  The download copies the frame sizes, the total, and a
// guessed share of the frame bytes at once; what a batch produced beyond the guess is fetched when
// the batch is collected.
// ---------------------------------------------------------------------------------------------
// int32 samples to the staging width.  Values keep their low 16 bits, as a cast does; streams of up
// to 16 bits hold nothing above them.
void narrow_samples(const int32_t *src, uint8_t *dst, size_t n, unsigned cw)
{
	if (cw == 4) {
		memcpy(dst, src, n * 4);
		return;
	}
	int16_t *d = (int16_t *)dst;
	size_t i = 0;
	for (; i + 8 <= n; i += 8) {
		__m128i a = _mm_loadu_si128((const __m128i *)(src + i)), b = _mm_loadu_si128((const __m128i *)(src + i + 4));
		a = _mm_srai_epi32(_mm_slli_epi32(a, 16), 16);
		b = _mm_srai_epi32(_mm_slli_epi32(b, 16), 16);
		_mm_storeu_si128((__m128i *)(d + i), _mm_packs_epi32(a, b));
	}
	for (; i < n; i++)
		d[i] = (int16_t)src[i];
}

void md5_worker(EncPipe *P)
{
	for (;;) {
		EncPipe::Md5Job job;
		{
			std::unique_lock<std::mutex> lk(P->mu);
			P->cv.wait(lk, [&] { return P->md5_quit || !P->md5_jobs.empty(); });
			if (P->md5_jobs.empty())
				return;
			job = P->md5_jobs.front();
			P->md5_jobs.pop_front();
		}
		fa_md5_accumulate_container(P->md5, job.p, P->cw, job.samples, P->bytes_per_sample);
		{
			std::lock_guard<std::mutex> lk(P->mu);
			P->md5_finished++;
		}
		P->cv.notify_all();
	}
}

void copy_worker(EncPipe *P, unsigned me)
{
	uint64_t seen = 0;
	for (;;) {
		EncPipe::CopyJob job;
		{
			std::unique_lock<std::mutex> lk(P->mu);
			P->cv.wait(lk, [&] { return P->copy_quit || P->copy_generation != seen; });
			if (P->copy_quit)
				return;
			seen = P->copy_generation;
			job = P->copy_jobs[me];
		}
		if (job.n)
			narrow_samples(job.src, job.dst, job.n, P->cw);
		{
			std::lock_guard<std::mutex> lk(P->mu);
			P->copy_pending--;
		}
		P->cv.notify_all();
	}
}

// Stage n interleaved samples (n counts single samples, not wide ones); large runs are shared out.
void stage_run(EncPipe *P, const int32_t *src, uint8_t *dst, size_t n)
{
	const size_t big = (size_t)1 << 19;
	if (n < big) {
		narrow_samples(src, dst, n, P->cw);
		return;
	}
	if (P->copiers.empty()) {
		unsigned want = 3;
		if (const char *env = getenv("FLAC_AMD_COPY_THREADS"))
			want = (unsigned)atoi(env);
		if (want > 15)
			want = 15;
		P->copy_jobs.resize(want);
		for (unsigned i = 0; i < want; i++)
			P->copiers.emplace_back(copy_worker, P, i);
		if (P->copiers.empty()) {
			narrow_samples(src, dst, n, P->cw);
			return;
		}
	}
	const size_t parts = P->copiers.size() + 1;
	const size_t per = ((n / parts) + 63) & ~(size_t)63;
	size_t at = 0;
	{
		std::lock_guard<std::mutex> lk(P->mu);
		for (size_t i = 0; i < P->copiers.size(); i++) {
			const size_t m = at + per <= n ? per : n - at;
			P->copy_jobs[i] = {src + at, dst + at * P->cw, m};
			at += m;
		}
		P->copy_pending = (unsigned)P->copiers.size();
		P->copy_generation++;
	}
	P->cv.notify_all();
	narrow_samples(src + at, dst + at * P->cw, n - at, P->cw);
	std::unique_lock<std::mutex> lk(P->mu);
	P->cv.wait(lk, [&] { return P->copy_pending == 0; });
}

void pipe_destroy(FLAC__StreamEncoderPrivate *q)
{
	EncPipe *P = q->pipe;
	if (!P)
		return;
	{
		std::lock_guard<std::mutex> lk(P->mu);
		P->md5_quit = true;
		P->copy_quit = true;
	}
	P->cv.notify_all();
	if (P->md5_thread.joinable())
		P->md5_thread.join();
	for (auto &t : P->copiers)
		t.join();
	(void)hipSetDevice(P->device);
	if (P->s_in) {
		(void)hipStreamSynchronize(P->s_in);
		(void)hipStreamSynchronize(P->s_comp);
		(void)hipStreamSynchronize(P->s_out);
	}
	for (EncSlot &S : P->slot) {
		(void)hipHostFree(S.pin_in);
		(void)hipHostFree(S.pin_out);
		(void)hipHostFree(S.pin_fb);
		(void)hipFree(S.dev_in);
		(void)hipFree(S.dev_out);
		(void)hipFree(S.dev_fb);
		(void)hipFree(S.dev_offs);
		if (S.uploaded) (void)hipEventDestroy(S.uploaded);
		if (S.computed) (void)hipEventDestroy(S.computed);
		if (S.done) (void)hipEventDestroy(S.done);
	}
	if (P->s_in) (void)hipStreamDestroy(P->s_in);
	if (P->s_comp) (void)hipStreamDestroy(P->s_comp);
	if (P->s_out) (void)hipStreamDestroy(P->s_out);
	delete P;
	q->pipe = 0;
}

bool pipe_create(FLAC__StreamEncoder *e)
{
	FLAC__StreamEncoderProtected *p = e->protected_;
	FLAC__StreamEncoderPrivate *q = e->private_;
	EncPipe *P = new (std::nothrow) EncPipe();
	if (!P)
		return false;
	q->pipe = P;
	P->device = fa_stream_device();
	P->cw = p->bits_per_sample <= 16 ? 2 : 4;
	P->channels = p->channels;
	P->blocksize = p->blocksize;
	P->bytes_per_sample = (p->bits_per_sample + 7) / 8;
	P->slot_samples = ((size_t)q->batch_frames + 1) * p->blocksize;
	P->in_bytes = P->slot_samples * p->channels * P->cw;
	P->out_cap = flacamd_encoder_max_output_bytes(q->dev, q->batch_frames + 1);
	// lossless audio rarely needs more than 3/4 of its PCM size; a batch that does is topped up at collection
	P->out_guess = (size_t)((double)P->slot_samples * p->channels * P->bytes_per_sample * 0.75) + 4096;
	if (P->out_guess > P->out_cap)
		P->out_guess = P->out_cap;
	P->filling = P->oldest = P->in_flight = 0;
	P->s_in = P->s_comp = P->s_out = 0;
	P->md5_posted = P->md5_finished = 0;
	P->md5_quit = false;
	P->md5_on = p->do_md5 != 0;
	P->md5 = &q->md5;
	{
		unsigned period = (unsigned)((double)p->sample_rate * 0.4 / (double)p->blocksize + 0.5);  // stream_encoder.c:871
		if (period == 0)
			period = 1;
		P->loose_carry = p->loose_mid_side_stereo && (q->batch_frames % period) != 0;
	}
	P->copy_generation = 0;
	P->copy_pending = 0;
	P->copy_quit = false;
	memset(P->slot, 0, sizeof P->slot);
	if (hipSetDevice(P->device) != hipSuccess)
		return false;
	bool ok = hipStreamCreateWithFlags(&P->s_in, hipStreamNonBlocking) == hipSuccess &&
	          hipStreamCreateWithFlags(&P->s_comp, hipStreamNonBlocking) == hipSuccess &&
	          hipStreamCreateWithFlags(&P->s_out, hipStreamNonBlocking) == hipSuccess;
	const size_t nfmax = (size_t)q->batch_frames + 1;
	for (EncSlot &S : P->slot) {
		ok = ok && hipHostMalloc((void **)&S.pin_in, P->in_bytes + 64, hipHostMallocDefault) == hipSuccess;
		ok = ok && hipHostMalloc((void **)&S.pin_out, P->out_cap + 64, hipHostMallocDefault) == hipSuccess;
		ok = ok && hipHostMalloc((void **)&S.pin_fb, sizeof(uint32_t) * nfmax + 16, hipHostMallocDefault) == hipSuccess;
		ok = ok && hipMalloc((void **)&S.dev_in, P->in_bytes + 64) == hipSuccess;
		ok = ok && hipMalloc((void **)&S.dev_out, P->out_cap + 64) == hipSuccess;
		ok = ok && hipMalloc((void **)&S.dev_fb, sizeof(uint32_t) * nfmax + 16) == hipSuccess;
		ok = ok && hipMalloc((void **)&S.dev_offs, sizeof(uint64_t) * (nfmax + 1)) == hipSuccess;
		ok = ok && hipEventCreateWithFlags(&S.uploaded, hipEventDisableTiming) == hipSuccess;
		ok = ok && hipEventCreateWithFlags(&S.computed, hipEventDisableTiming) == hipSuccess;
		ok = ok && hipEventCreateWithFlags(&S.done, hipEventDisableTiming) == hipSuccess;
		if (!ok)
			break;
		S.pin_total = (uint64_t *)(S.pin_fb + ((nfmax + 1) & ~(size_t)1));
	}
	if (!ok)
		return false;
	if (P->md5_on)
		P->md5_thread = std::thread(md5_worker, P);
	return true;
}

// Hand the frames of the oldest batch in flight to the client: process_frame_'s tail
// (stream_encoder.c:2885-2915) + write_bitbuffer_ (:2379), in frame order.
bool collect_oldest(FLAC__StreamEncoder *e)
{
	FLAC__StreamEncoderProtected *p = e->protected_;
	FLAC__StreamEncoderPrivate *q = e->private_;
	EncPipe *P = q->pipe;
	EncSlot &S = P->slot[P->oldest];
	const unsigned bs = q->configured_blocksize, ch = p->channels;
	auto retire = [&]() {
		S.in_flight = false;
		P->oldest = (P->oldest + 1) % ENC_SLOTS;
		P->in_flight--;
	};
	if (hipSetDevice(P->device) != hipSuccess || hipEventSynchronize(S.done) != hipSuccess || flacamd_encoder_status(q->dev) != FLACAMD_OK) {
		p->state = FLAC__STREAM_ENCODER_FRAMING_ERROR;
		retire();
		return false;
	}
	const size_t total = (size_t)*S.pin_total;
	if (total > P->out_cap) {
		p->state = FLAC__STREAM_ENCODER_FRAMING_ERROR;
		retire();
		return false;
	}
	if (total > P->out_guess && hipMemcpy(S.pin_out + P->out_guess, S.dev_out + P->out_guess, total - P->out_guess, hipMemcpyDeviceToHost) != hipSuccess) {
		p->state = FLAC__STREAM_ENCODER_FRAMING_ERROR;
		retire();
		return false;
	}
	const size_t wide = S.wide;
	const uint32_t nf = S.nframes;
	if (p->verify) {
		// verify_write_callback_ (stream_encoder.c:4161): decode what was just produced, compare with what went in
		// (both at the staging width, which holds every sample exactly)
		std::vector<uint8_t> back(wide * ch * P->cw);
		flacamd_stream_desc sd;
		sd.byte_offset = 0;
		sd.byte_length = total;
		sd.sample_offset = 0;
		sd.total_samples = wide;
		sd.first_sample = (FLAC__uint64)S.first_frame * bs;
		uint32_t dfr = 0, bad = 0;
		uint64_t dwide = 0;
		const int rc = flacamd_decode_streams_host(q->verify_dec, S.pin_out, total, &sd, 1, back.data(), P->cw, wide, &dfr, &dwide, &bad);
		if (rc != FLACAMD_OK || bad || dfr != nf) {
			p->state = FLAC__STREAM_ENCODER_VERIFY_DECODER_ERROR;
			retire();
			return false;
		}
		if (memcmp(back.data(), S.pin_in, back.size()) != 0) {
			for (size_t i = 0; i < wide * ch; i++) {
				const int32_t want = P->cw == 2 ? ((const int16_t *)S.pin_in)[i] : ((const int32_t *)S.pin_in)[i];
				const int32_t got = P->cw == 2 ? ((const int16_t *)back.data())[i] : ((const int32_t *)back.data())[i];
				if (want != got) {
					const size_t w = i / ch;
					q->ve_absolute_sample = (FLAC__uint64)S.first_frame * bs + w;
					q->ve_frame = S.first_frame + (unsigned)(w / bs);
					q->ve_channel = (unsigned)(i % ch);
					q->ve_sample = (unsigned)(w % bs);
					q->ve_expected = want;
					q->ve_got = got;
					break;
				}
			}
			p->state = FLAC__STREAM_ENCODER_VERIFY_MISMATCH_IN_AUDIO_DATA;
			retire();
			return false;
		}
	}
	size_t off = 0, done = 0;
	bool ok = true;
	for (uint32_t f = 0; f < nf; f++) {
		const unsigned n = (unsigned)((wide - done) < bs ? (wide - done) : bs);
		const size_t fb = S.pin_fb[f];
		p->blocksize = n;  // finish() sets the short block's size before its frame is written (:1319)
		if (off + fb > total || write_frame(e, S.pin_out + off, fb, n) != FLAC__STREAM_ENCODER_WRITE_STATUS_OK) {
			p->state = off + fb > total ? FLAC__STREAM_ENCODER_FRAMING_ERROR : FLAC__STREAM_ENCODER_CLIENT_ERROR;
			ok = false;
			break;
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
	retire();
	return ok;
}

// Send the first `wide` staged samples of the filling slot on their way (whole blocks, or with
// is_last the stream's final blocks) and move the samples staged beyond them to the next slot.
bool submit_filling(FLAC__StreamEncoder *e, size_t wide, bool is_last)
{
	FLAC__StreamEncoderProtected *p = e->protected_;
	FLAC__StreamEncoderPrivate *q = e->private_;
	EncPipe *P = q->pipe;
	EncSlot &S = P->slot[P->filling];
	const unsigned ch = p->channels;
	const size_t wbytes = (size_t)ch * P->cw;
	auto device_failed = [&]() {
		p->state = FLAC__STREAM_ENCODER_FRAMING_ERROR;
		return false;
	};
	if (hipSetDevice(P->device) != hipSuccess)
		return device_failed();
	S.wide = wide;
	S.first_frame = q->current_frame_number + [&] {
		// frames of the batches in flight have not reached current_frame_number yet
		uint32_t ahead = 0;
		for (unsigned k = 0, i = P->oldest; k < P->in_flight; k++, i = (i + 1) % ENC_SLOTS)
			ahead += P->slot[i].nframes;
		return ahead;
	}();
	if (hipMemcpyAsync(S.dev_in, S.pin_in, wide * wbytes, hipMemcpyHostToDevice, P->s_in) != hipSuccess || hipEventRecord(S.uploaded, P->s_in) != hipSuccess ||
	    hipStreamWaitEvent(P->s_comp, S.uploaded, 0) != hipSuccess)
		return device_failed();
	// The loose mid/side decision of a batch's last leader frame has to reach the next batch (through the host, which
	// makes the call wait for the kernels) unless every batch starts on a decision boundary
	// (stream_encoder.c:2952-2962), which the batch size is chosen for.
	uint32_t nf = 0;
	const int rc = flacamd_encode_frames(q->dev, S.dev_in, P->cw, wide, S.first_frame, is_last ? 1 : 0, P->loose_carry ? &q->loose_state : 0, S.dev_out,
	                                     P->out_cap, S.dev_fb, S.dev_offs, &nf, P->s_comp);
	if (rc != FLACAMD_OK) {
		p->state = rc == FLACAMD_E_NOMEM ? FLAC__STREAM_ENCODER_MEMORY_ALLOCATION_ERROR : FLAC__STREAM_ENCODER_FRAMING_ERROR;
		return false;
	}
	S.nframes = nf;
	if (hipEventRecord(S.computed, P->s_comp) != hipSuccess || hipStreamWaitEvent(P->s_out, S.computed, 0) != hipSuccess ||
	    hipMemcpyAsync(S.pin_fb, S.dev_fb, sizeof(uint32_t) * nf, hipMemcpyDeviceToHost, P->s_out) != hipSuccess ||
	    hipMemcpyAsync(S.pin_total, S.dev_offs + nf, sizeof(uint64_t), hipMemcpyDeviceToHost, P->s_out) != hipSuccess ||
	    hipMemcpyAsync(S.pin_out, S.dev_out, P->out_guess, hipMemcpyDeviceToHost, P->s_out) != hipSuccess || hipEventRecord(S.done, P->s_out) != hipSuccess)
		return device_failed();
	if (P->md5_on) {
		{
			std::lock_guard<std::mutex> lk(P->mu);
			P->md5_jobs.push_back({S.pin_in, wide * ch});
			S.md5_ticket = ++P->md5_posted;
		}
		P->cv.notify_all();
	}
	S.in_flight = true;
	P->in_flight++;
	// the next slot to fill: free it first if it is still on its way
	const unsigned next = (P->filling + 1) % ENC_SLOTS;
	if (P->in_flight == ENC_SLOTS) {
		if (!collect_oldest(e))
			return false;
	}
	EncSlot &N = P->slot[next];
	if (P->md5_on && N.md5_ticket) {
		std::unique_lock<std::mutex> lk(P->mu);
		P->cv.wait(lk, [&] { return P->md5_finished >= N.md5_ticket; });
	}
	const size_t left = S.fill - wide;
	if (left)
		memcpy(N.pin_in, S.pin_in + wide * wbytes, left * wbytes);
	N.fill = left;
	S.fill = 0;
	P->filling = next;
	// frames that are ready go out now rather than when their slot is needed
	while (P->in_flight && hipEventQuery(P->slot[P->oldest].done) == hipSuccess)
		if (!collect_oldest(e))
			return false;
	return true;
}

// The reference encodes a block once blocksize+1 samples are buffered (OVERREAD_, :504-515), so
// the last block is always left for finish().  Batched: a batch goes out when batch_frames whole
// blocks are staged and at least one more sample follows them.
bool submit_if_full(FLAC__StreamEncoder *e)
{
	FLAC__StreamEncoderPrivate *q = e->private_;
	EncPipe *P = q->pipe;
	const size_t batch = (size_t)q->batch_frames * q->configured_blocksize;
	if (P->slot[P->filling].fill > batch)
		return submit_filling(e, batch, false);
	return true;
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
			// about 2 M samples per batch, three batches in flight (measured: larger batches leave the caller's thread
			// waiting at the pipeline's ends, smaller ones pay per-batch launches); a stream known to be short gets
			// buffers to match (two batches of half its length)
			bf = ((size_t)2 << 20) / ((size_t)p->blocksize * p->channels);
			if (bf > 1024) bf = 1024;
			if (p->total_samples_estimate > 0) {
				const size_t half = (size_t)((p->total_samples_estimate / p->blocksize + 2) / 2 + 1);
				if (half < bf) bf = half;
			}
			if (bf < 8) bf = 8;
			if (p->loose_mid_side_stereo) {
				// batches that start where the loose mid/side decision is taken afresh need no state from their predecessor
				size_t period = (size_t)((double)p->sample_rate * 0.4 / (double)p->blocksize + 0.5);
				if (period == 0) period = 1;
				bf = (bf + period - 1) / period * period;
			}
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
	if (!pipe_create(e)) {
		p->state = FLAC__STREAM_ENCODER_MEMORY_ALLOCATION_ERROR;
		return FLAC__STREAM_ENCODER_INIT_STATUS_ENCODER_ERROR;
	}

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
	// frames arrive a batch at a time: a large stdio buffer turns thousands of small writes into a few
	// (only for a file this library opened itself; the buffer is the C library's own)
	if (filename)
		(void)setvbuf(f, 0, _IOFBF, (size_t)1 << 20);
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
	if (p->state == FLAC__STREAM_ENCODER_OK && !q->is_being_deleted && q->pipe) {
		// what is staged goes out as the stream's last blocks, the final one short or not (:1317-1322)
		EncPipe *P = q->pipe;
		if (P->slot[P->filling].fill != 0 && !submit_filling(e, P->slot[P->filling].fill, true))
			error = true;
		while (!error && P->in_flight)
			if (!collect_oldest(e))
				error = true;
	}
	if (p->do_md5 && q->dev && q->pipe) {
		// every sample has been posted by now; wait for the digest thread to have taken them all in
		EncPipe *P = q->pipe;
		{
			std::unique_lock<std::mutex> lk(P->mu);
			P->cv.wait(lk, [&] { return P->md5_finished >= P->md5_posted; });
		}
		fa_md5_final(&q->md5, q->streaminfo.data.stream_info.md5sum);
	}
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
	EncPipe *P = q->pipe;
	const unsigned ch = p->channels;
	const size_t wbytes = (size_t)ch * P->cw;
	unsigned done = 0;
	while (done < samples) {
		// the samples are copied (narrowed) straight into the filling slot's pinned staging buffer
		EncSlot &S = P->slot[P->filling];
		const size_t room = P->slot_samples - S.fill;
		const unsigned n = (unsigned)((samples - done) < room ? (samples - done) : room);
		stage_run(P, buffer + (size_t)done * ch, S.pin_in + S.fill * wbytes, (size_t)n * ch);
		S.fill += n;
		done += n;
		if (!submit_if_full(e))
			return 0;
	}
	return 1;
}

FLAC__bool FLAC__stream_encoder_process(FLAC__StreamEncoder *e, const FLAC__int32 *const buffer[], unsigned samples)
{
	FLAC__StreamEncoderProtected *p = e->protected_;
	FLAC__StreamEncoderPrivate *q = e->private_;
	if (p->state != FLAC__STREAM_ENCODER_OK)
		return 0;
	EncPipe *P = q->pipe;
	const unsigned ch = p->channels;
	const size_t wbytes = (size_t)ch * P->cw;
	unsigned done = 0;
	while (done < samples) {
		EncSlot &S = P->slot[P->filling];
		const size_t room = P->slot_samples - S.fill;
		const unsigned n = (unsigned)((samples - done) < room ? (samples - done) : room);
		uint8_t *dst = S.pin_in + S.fill * wbytes;
		// planar in, interleaved (and narrowed) into the staging buffer
		for (unsigned c = 0; c < ch; c++) {
			const FLAC__int32 *src = buffer[c] + done;
			if (P->cw == 2) {
				int16_t *d = (int16_t *)dst + c;
				for (unsigned i = 0; i < n; i++)
					d[(size_t)i * ch] = (int16_t)src[i];
			} else {
				int32_t *d = (int32_t *)dst + c;
				for (unsigned i = 0; i < n; i++)
					d[(size_t)i * ch] = src[i];
			}
		}
		S.fill += n;
		done += n;
		if (!submit_if_full(e))
			return 0;
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
