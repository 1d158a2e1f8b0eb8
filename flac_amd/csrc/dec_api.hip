// C ABI of the batch decoder: handle management and kernel sequencing.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <new>

#include "dec_kernels.h"
#include "flac_amd.h"
#include "flac_amd_internal.h"

#define HIP_TRY(expr) \
	do { \
		hipError_t e_ = (expr); \
		if (e_ != hipSuccess) \
			return fa_fail(FLACAMD_E_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); \
	} while (0)

struct flacamd_decoder {
	int device;
	uint32_t channels, bps, max_blocksize, max_frames, number_scale, lw;
	uint64_t max_input_bytes;
	uint32_t max_streams;
	FaStream *streams;
	uint32_t *block_counts;
	FaFrameCand *kept;              // the headers of each 4 KiB block as the counting pass parsed them
	FaSubInfo *subinfo;             // one record per (frame, channel) once flacamd_decoder_enable_subframe_info asked for them
	uint64_t *crc_ahead;            // per header: the CRC pass that runs beside the frame decoder leaves its guess and verdict here
	hipStream_t side;               // that pass and the clearing of the output run here, between ev_fork and ev_join
	hipEvent_t ev_fork, ev_join;
	FaFrameCand *cands;
	uint32_t *ncand;                // [0] count, [1] overflow flag
	int32_t *tmp;
	uint64_t *sub_start;            // bit position at which each (frame, channel)'s subframe starts
	uint32_t planar;                // flacamd_decoder_set_planar_output
	uint32_t pair;                  // split form, stereo: a workgroup takes both channels of lw frames and writes final PCM (DEC_PAIR)
	uint32_t split, skip_lw;        // the split form (walk, then every subframe at once) is used, not a launch per channel; frames per wave of its walk
	unsigned long long *result;     // good frames, wide samples, bad candidates
	int profiling, have_times;
	hipEvent_t ev[FLACAMD_DEC_STAGES + 1];
	hipStream_t last_stream;
	// staging for the host-buffer entry point
	uint8_t *stage_in;
	void *stage_out;
	size_t stage_out_bytes;
};

extern "C" void flacamd_decoder_destroy(flacamd_decoder *d)
{
	if (!d)
		return;
	(void)hipSetDevice(d->device);
	(void)hipFree(d->streams);
	(void)hipFree(d->block_counts);
	(void)hipFree(d->kept);
	(void)hipFree(d->crc_ahead);
	(void)hipFree(d->subinfo);
	if (d->side)
		(void)hipStreamDestroy(d->side);
	if (d->ev_fork)
		(void)hipEventDestroy(d->ev_fork);
	if (d->ev_join)
		(void)hipEventDestroy(d->ev_join);
	(void)hipFree(d->cands);
	(void)hipFree(d->ncand);
	(void)hipFree(d->tmp);
	(void)hipFree(d->sub_start);
	(void)hipFree(d->result);
	(void)hipFree(d->stage_in);
	(void)hipFree(d->stage_out);
	for (int i = 0; i <= FLACAMD_DEC_STAGES; i++)
		if (d->ev[i])
			(void)hipEventDestroy(d->ev[i]);
	delete d;
}

extern "C" int flacamd_decoder_create(uint32_t channels, uint32_t bits_per_sample, uint32_t max_blocksize, int device,
                                      uint64_t max_input_bytes, uint32_t max_frames, flacamd_decoder **out)
{
	if (!out)
		return fa_fail(FLACAMD_E_INVALID, "null out pointer");
	*out = nullptr;
	if (channels == 0 || channels > FLACAMD_MAX_CHANNELS)
		return fa_fail(FLACAMD_E_INVALID, "invalid number of channels %u", channels);
	if (bits_per_sample < 4 || bits_per_sample > 32)
		return fa_fail(FLACAMD_E_INVALID, "invalid bits per sample %u", bits_per_sample);
	if (max_blocksize < 1 || max_blocksize > 65535)
		return fa_fail(FLACAMD_E_INVALID, "invalid blocksize %u", max_blocksize);
	if (max_frames == 0 || max_input_bytes == 0)
		return fa_fail(FLACAMD_E_INVALID, "empty capacity");
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
		return fa_fail(FLACAMD_E_DEVICE, "no HIP device available: the FLAC engine needs an AMD GPU");
	if (device < 0 || device >= ndev)
		return fa_fail(FLACAMD_E_DEVICE, "HIP device %d out of range (%d present)", device, ndev);
	HIP_TRY(hipSetDevice(device));
	flacamd_decoder *d = new (std::nothrow) flacamd_decoder();
	if (!d)
		return fa_fail(FLACAMD_E_NOMEM, "out of host memory");
	memset(d, 0, sizeof *d);
	d->device = device;
	d->channels = channels;
	d->bps = bits_per_sample;
	d->max_blocksize = max_blocksize;
	d->number_scale = max_blocksize;
	d->max_frames = max_frames;
	d->max_input_bytes = max_input_bytes;
	d->max_streams = max_frames;  // a stream holds at least one frame
	const uint32_t nb = fa_dec_sync_blocks(max_input_bytes);
	const size_t groups = (max_frames + 63) / 64;
#define ALLOC(ptr, bytes) \
	do { \
		hipError_t e_ = hipMalloc((void **)&(ptr), (bytes) ? (bytes) : 16); \
		if (e_ != hipSuccess) { \
			flacamd_decoder_destroy(d); \
			return fa_fail(FLACAMD_E_NOMEM, "hipMalloc of %zu bytes failed: %s", (size_t)(bytes), hipGetErrorString(e_)); \
		} \
	} while (0)
	ALLOC(d->streams, sizeof(FaStream) * d->max_streams);
	// per 4 KiB of input: a count of frame headers, and behind the counts four 64-bit lane masks
	ALLOC(d->block_counts, sizeof(uint32_t) * ((size_t)nb + 2) + sizeof(uint64_t) * 4 * (size_t)nb);
	{
		// FLAC_AMD_DEC_SIDE=0: everything on the caller's stream, one kernel after the other
		const char *env = getenv("FLAC_AMD_DEC_SIDE");
		if (!(env && !strcmp(env, "0"))) {
			ALLOC(d->crc_ahead, sizeof(uint64_t) * max_frames);
			if (hipStreamCreateWithFlags(&d->side, hipStreamNonBlocking) != hipSuccess ||
			    hipEventCreateWithFlags(&d->ev_fork, hipEventDisableTiming) != hipSuccess ||
			    hipEventCreateWithFlags(&d->ev_join, hipEventDisableTiming) != hipSuccess) {
				flacamd_decoder_destroy(d);
				return fa_fail(FLACAMD_E_DEVICE, "creating the decoder's second stream failed");
			}
		}
	}
	ALLOC(d->kept, sizeof(FaFrameCand) * FA_SYNC_KEEP * (size_t)nb);
	ALLOC(d->cands, sizeof(FaFrameCand) * max_frames);
	ALLOC(d->ncand, 16);
	ALLOC(d->tmp, sizeof(int32_t) * groups * 64 * channels * max_blocksize);
	// frames per wave: few frames are spread thin over many waves (about two per SIMD of the chip).  While
	// the batch leaves the chip room, frames with two or more channels are decoded in the split form, a walk that
	// finds where every subframe starts and then a lane per subframe (FLAC_AMD_DEC_MODE=lane keeps a lane per frame,
	// =split forces the split form, for comparisons).  The walk costs about half of what decoding the same samples
	// costs, so a batch large enough to fill the chip with whole frames is better off without it.  Measured
	// (tools/dec_mode_probe.py, tools/bench_configs.py; frames stage, a lane per frame -> split): 16-bit stereo
	// 256 frames 1.31 -> 1.01 ms, 6 460 frames 1.49 -> 1.20, 16 384 frames 1.61 -> 1.39; 24-bit stereo 14 063 frames
	// 1.78 -> 1.46; eight channels of 24 bits 2 813 frames 5.9 -> 3.9; 67 500 stereo frames (135 000 subframes)
	// 3.70 (32 frames per wave) -> 4.39.  Eight frames per wave up to 4 096 subframes, sixteen beyond.
	{
		const char *mode = getenv("FLAC_AMD_DEC_MODE");
		const uint64_t units = (uint64_t)max_frames * channels;
		d->split = channels >= 2 && units <= 40000 && !(mode && !strcmp(mode, "lane"));
		if (mode && !strcmp(mode, "split") && channels >= 2)
			d->split = 1;
		uint32_t lw = 4;
		if (d->split)
			lw = units <= 4096 ? 8 : 16;   // (4 096 stereo frames, frames stage: 0.68 ms at eight, 0.59 at sixteen)
		else {
			while (lw < 16 && (max_frames + lw - 1) / lw > 2048)
				lw *= 2;
			// beyond what the chip holds at once at sixteen frames per wave, thirty-two: half the waves, each as long
			// (C5: frames stage 4.52 -> 3.70 ms)
			if ((max_frames + 15) / 16 > 3000)
				lw = 32;
		}
		ALLOC(d->sub_start, sizeof(uint64_t) * (size_t)max_frames * channels);
		if (const char *env = getenv("FLAC_AMD_DEC_LW")) {
			const uint32_t v = (uint32_t)atoi(env);
			if (v >= 1 && v <= (d->split ? 16u : 32u) && (v & (v - 1)) == 0)
				lw = v;
		}
		// Stereo in the split form, the CRC pass running ahead: both channels of a frame in one workgroup (half as many
		// frames per group, the same columns per wave), which then writes final PCM itself.  FLAC_AMD_DEC_PAIR=0: planes for all.
		{
			const char *env = getenv("FLAC_AMD_DEC_PAIR"), *side = getenv("FLAC_AMD_DEC_SIDE");
			const bool off = (env && !strcmp(env, "0")) || (side && !strcmp(side, "0"));
			// (the pair kernel is 25-60 us slower than the one-channel one whatever the batch - a wave's lanes are of two
			// channels, whose partitions start at different samples - and saves the output kernel's pass over the planes,
			// which grows with the batch: 4 096 frames 0.714 -> 0.734 ms, 6 460 frames 0.859 -> 0.814)
			uint32_t least = 5000;
			if (env && atoi(env) > 1)
				least = (uint32_t)atoi(env);   // FLAC_AMD_DEC_PAIR=n: from n frames up
			if (d->split && !off && lw >= 2 && max_frames >= least && fa_dec_pair_fits(channels, max_frames, lw / 2)) {
				d->pair = 1;
				lw /= 2;
			}
		}
		d->lw = lw;
		// the walk of the split form: a lane per frame again, thin enough for the input pump to keep every lane fed
		// (sixteen where the words per lane and block are few: 6 460 16-bit stereo frames 0.656 -> 0.643 ms, 16 384 frames
		// 1.15 -> 1.07; 14 063 24-bit frames 1.46 -> 1.67)
		// (four for wider samples while that still leaves every wave a SIMD of its own: 2 813 frames of eight 24-bit channels,
		// frames stage 2.53 -> 2.37 ms at four, 2.39 at two; 14 063 frames of 24-bit stereo 1.55 -> 1.82 at four)
		uint32_t slw = (bits_per_sample <= 16 && channels == 2 && max_frames >= 6000) ? 16 : (bits_per_sample > 16 && max_frames <= 4096) ? 4 : 8;
		if (const char *env = getenv("FLAC_AMD_DEC_SKIP_LW")) {
			const uint32_t v = (uint32_t)atoi(env);
			if (v >= 1 && v <= 16 && (v & (v - 1)) == 0)
				slw = v;
		}
		d->skip_lw = slw;
	}
	ALLOC(d->result, 32);
#undef ALLOC
	uint16_t t16[256], pow16[32];
	uint8_t t8[256];
	for (uint32_t i = 0; i < 256; i++) {
		uint32_t c = i << 8;
		for (int b = 0; b < 8; b++)
			c = (c & 0x8000) ? ((c << 1) ^ 0x8005) : (c << 1);
		t16[i] = (uint16_t)c;
		uint32_t e = i;
		for (int b = 0; b < 8; b++)
			e = (e & 0x80) ? ((e << 1) ^ 0x07) : (e << 1);
		t8[i] = (uint8_t)e;
	}
	auto mul = [](uint32_t a, uint32_t b) {
		uint32_t r = 0;
		for (int i = 0; i < 16; i++) {
			if (b & 1) r ^= a;
			b >>= 1;
			a <<= 1;
			if (a & 0x10000) a ^= 0x18005;
		}
		return r;
	};
	uint32_t p = 0x100;
	for (int j = 0; j < 32; j++) {
		pow16[j] = (uint16_t)p;
		p = mul(p, p);
	}
	if (hipError_t up = fa_dec_upload_tables(t16, pow16, t8); up != hipSuccess) {
		flacamd_decoder_destroy(d);
		return fa_fail(FLACAMD_E_DEVICE, "CRC table upload failed: %s", hipGetErrorString(up));
	}
	for (int i = 0; i <= FLACAMD_DEC_STAGES; i++)
		if (hipEventCreate(&d->ev[i]) != hipSuccess) {
			flacamd_decoder_destroy(d);
			return fa_fail(FLACAMD_E_DEVICE, "hipEventCreate failed");
		}
	if (hipError_t sy = hipDeviceSynchronize(); sy != hipSuccess) {
		flacamd_decoder_destroy(d);
		return fa_fail(FLACAMD_E_DEVICE, "hipDeviceSynchronize failed: %s", hipGetErrorString(sy));
	}
	*out = d;
	return FLACAMD_OK;
}

extern "C" int flacamd_decode_streams(flacamd_decoder *d, const uint8_t *flac_dev, uint64_t flac_bytes, const flacamd_stream_desc *streams_host,
                                      uint32_t nstreams, void *pcm_out_dev, uint32_t out_sample_bytes, uint64_t out_capacity_wide, void *stream)
{
	if (!d || !flac_dev || !streams_host || !pcm_out_dev)
		return fa_fail(FLACAMD_E_INVALID, "flacamd_decode_streams: null argument");
	if (out_sample_bytes != 2 && out_sample_bytes != 4)
		return fa_fail(FLACAMD_E_INVALID, "out_sample_bytes must be 2 or 4");
	if (out_sample_bytes * 8 < d->bps)
		return fa_fail(FLACAMD_E_INVALID, "%u-byte output samples cannot hold %u bits", out_sample_bytes, d->bps);
	if (flac_bytes > d->max_input_bytes)
		return fa_fail(FLACAMD_E_INVALID, "input of %llu bytes exceeds the decoder's capacity", (unsigned long long)flac_bytes);
	if (nstreams == 0 || nstreams > d->max_streams)
		return fa_fail(FLACAMD_E_INVALID, "bad stream count %u", nstreams);
	if ((uintptr_t)flac_dev & 3)
		return fa_fail(FLACAMD_E_INVALID, "the FLAC buffer must be 4-byte aligned");
	uint64_t prev_end = 0;
	for (uint32_t i = 0; i < nstreams; i++) {
		const flacamd_stream_desc &s = streams_host[i];
		if (s.byte_offset < prev_end || s.byte_offset + s.byte_length > flac_bytes)
			return fa_fail(FLACAMD_E_INVALID, "stream %u: byte range out of order or out of bounds", i);
		prev_end = s.byte_offset + s.byte_length;
		if (s.first_sample == FLACAMD_FIRST_SAMPLE_AUTO && nstreams != 1)
			return fa_fail(FLACAMD_E_INVALID, "FLACAMD_FIRST_SAMPLE_AUTO needs a single stream");
	}
	HIP_TRY(hipSetDevice(d->device));
	hipStream_t st = (hipStream_t)stream;
	if (flac_bytes == 0) {
		// nothing to look through: no frames, the output cleared like that of any other input
		HIP_TRY(hipMemsetAsync(d->ncand, 0, 16, st));
		HIP_TRY(hipMemsetAsync(d->result, 0, 32, st));
		HIP_TRY(hipMemsetAsync(pcm_out_dev, 0, (size_t)out_capacity_wide * d->channels * out_sample_bytes, st));
		d->last_stream = st;
		return FLACAMD_OK;
	}
	static_assert(sizeof(FaStream) == sizeof(flacamd_stream_desc), "stream descriptor layout");
	HIP_TRY(hipMemcpyAsync(d->streams, streams_host, sizeof(FaStream) * nstreams, hipMemcpyHostToDevice, st));
	FaDecParams P;
	P.channels = d->channels;
	P.bps = d->bps;
	P.max_blocksize = d->max_blocksize;
	P.out_sample_bytes = out_sample_bytes;
	P.nstreams = nstreams;
	P.max_frames = d->max_frames;
	P.number_scale = d->number_scale;
	P.lw = d->lw;
	P.skip_lw = d->skip_lw;
	P.run = 16;
	P.flac_bytes = flac_bytes;
	P.out_capacity_wide = out_capacity_wide;
	P.subinfo = d->subinfo;
	P.planar = d->planar;
	P.pad = 0;
	const bool prof = d->profiling != 0;
	int evi = 0;
#define STAGE_MARK() \
	do { \
		if (prof) \
			HIP_TRY(hipEventRecord(d->ev[evi++], st)); \
	} while (0)
	STAGE_MARK();
	// Frames that fail their checks leave zeros, like the reference's zeroed blocks.  The frame decoder is a few long
	// chains of dependent instructions and leaves most of the chip's bandwidth and issue slots idle: clearing the
	// output and the CRC-16 of the frames (on the guess that each ends where the next header was found) run beside it
	// on a second stream, forked and joined with events so that the caller's stream sees one ordered piece of
	// work.
	const size_t out_bytes = (size_t)out_capacity_wide * d->channels * out_sample_bytes;
	if (!d->side)
		HIP_TRY(hipMemsetAsync(pcm_out_dev, 0, out_bytes, st));
	fa_dec_launch_sync(P, flac_dev, d->streams, d->block_counts, d->kept, d->cands, d->ncand, d->result, st);
	if (d->side) {
		HIP_TRY(hipEventRecord(d->ev_fork, st));
		HIP_TRY(hipStreamWaitEvent(d->side, d->ev_fork, 0));
		HIP_TRY(hipMemsetAsync(pcm_out_dev, 0, out_bytes, d->side));
		fa_dec_launch_crc(P, flac_dev, d->cands, d->ncand, d->streams, d->crc_ahead, true, d->side);
		HIP_TRY(hipEventRecord(d->ev_join, d->side));
	}
	STAGE_MARK();
	// (the pair form needs the other stream's work - output cleared, CRC verdicts - before its second launch; otherwise
	// the join comes after the frame decoder)
	const bool pair = d->pair && d->side && !d->planar;
	fa_dec_launch_frames(P, flac_dev, d->streams, d->cands, d->ncand, d->tmp, d->sub_start, d->split != 0, pair, d->crc_ahead,
	                     pair ? d->ev_join : nullptr, pcm_out_dev, st);
	if (d->side && !pair)
		HIP_TRY(hipStreamWaitEvent(st, d->ev_join, 0));
	STAGE_MARK();
	fa_dec_launch_crc(P, flac_dev, d->cands, d->ncand, d->streams, d->crc_ahead, false, st);
	STAGE_MARK();
	fa_dec_launch_output(P, d->streams, d->cands, d->ncand, d->tmp, pcm_out_dev, d->result, st);
	STAGE_MARK();
#undef STAGE_MARK
	if (prof)
		d->have_times = 1;
	d->last_stream = st;
	HIP_TRY(hipGetLastError());
	return FLACAMD_OK;
}

static_assert(sizeof(FaSubInfo) == sizeof(flacamd_subframe_info), "subframe record layout");

extern "C" int flacamd_decoder_enable_subframe_info(flacamd_decoder *d, int on)
{
	if (!d)
		return fa_fail(FLACAMD_E_INVALID, "null decoder");
	HIP_TRY(hipSetDevice(d->device));
	if (d->last_stream || d->subinfo)
		HIP_TRY(hipDeviceSynchronize());   // (a decode in flight may be writing records)
	if (on && !d->subinfo) {
		const size_t bytes = sizeof(FaSubInfo) * (size_t)d->max_frames * d->channels;
		hipError_t e = hipMalloc((void **)&d->subinfo, bytes);
		if (e != hipSuccess) {
			d->subinfo = nullptr;
			return fa_fail(FLACAMD_E_NOMEM, "hipMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
		}
	} else if (!on && d->subinfo) {
		(void)hipFree(d->subinfo);
		d->subinfo = nullptr;
	}
	return FLACAMD_OK;
}

extern "C" int flacamd_decoder_set_planar_output(flacamd_decoder *d, int on)
{
	if (!d)
		return fa_fail(FLACAMD_E_INVALID, "null decoder");
	d->planar = on ? 1 : 0;
	return FLACAMD_OK;
}

extern "C" int flacamd_decoder_subframe_info(flacamd_decoder *d, uint32_t first_frame, uint32_t nframes, flacamd_subframe_info *out)
{
	if (!d || !out)
		return fa_fail(FLACAMD_E_INVALID, "null argument");
	if (!d->subinfo)
		return fa_fail(FLACAMD_E_INVALID, "subframe records were not asked for (flacamd_decoder_enable_subframe_info)");
	if ((uint64_t)first_frame + nframes > d->max_frames)
		return fa_fail(FLACAMD_E_INVALID, "frames [%u, %u) are beyond the decoder's capacity of %u", first_frame, first_frame + nframes, d->max_frames);
	HIP_TRY(hipSetDevice(d->device));
	HIP_TRY(hipStreamSynchronize(d->last_stream));
	HIP_TRY(hipMemcpy(out, d->subinfo + (size_t)first_frame * d->channels, sizeof(FaSubInfo) * (size_t)nframes * d->channels, hipMemcpyDeviceToHost));
	return FLACAMD_OK;
}

extern "C" int flacamd_decoder_result(flacamd_decoder *d, uint32_t *frames, uint64_t *wide_samples, uint32_t *bad_frames)
{
	if (!d)
		return fa_fail(FLACAMD_E_INVALID, "null decoder");
	HIP_TRY(hipSetDevice(d->device));
	HIP_TRY(hipStreamSynchronize(d->last_stream));
	unsigned long long r[3];
	uint32_t nc[2];
	HIP_TRY(hipMemcpy(r, d->result, sizeof r, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(nc, d->ncand, sizeof nc, hipMemcpyDeviceToHost));
	if (frames) *frames = (uint32_t)r[0];
	if (wide_samples) *wide_samples = r[1];
	if (bad_frames) *bad_frames = (uint32_t)r[2];
	if (nc[1])
		return fa_fail(FLACAMD_E_OVERFLOW, "more frames in the input than the decoder's capacity of %u", d->max_frames);
	return FLACAMD_OK;
}

extern "C" int flacamd_decoder_set_frame_number_scale(flacamd_decoder *d, uint32_t samples_per_frame_number)
{
	if (!d || samples_per_frame_number == 0)
		return fa_fail(FLACAMD_E_INVALID, "bad arguments");
	d->number_scale = samples_per_frame_number;
	return FLACAMD_OK;
}

extern "C" int flacamd_decoder_frames(flacamd_decoder *d, flacamd_frame_info *out, uint32_t capacity, uint32_t *count, int *truncated)
{
	if (!d || !count || (capacity && !out))
		return fa_fail(FLACAMD_E_INVALID, "bad arguments");
	static_assert(sizeof(flacamd_frame_info) == sizeof(FaFrameCand), "frame table layout");
	HIP_TRY(hipSetDevice(d->device));
	HIP_TRY(hipStreamSynchronize(d->last_stream));
	uint32_t nc[2];
	HIP_TRY(hipMemcpy(nc, d->ncand, sizeof nc, hipMemcpyDeviceToHost));
	uint32_t n = nc[0];
	int cut = nc[1] ? 1 : 0;
	if (n > capacity) {
		n = capacity;
		cut = 1;
	}
	if (n)
		HIP_TRY(hipMemcpy(out, d->cands, sizeof(FaFrameCand) * n, hipMemcpyDeviceToHost));
	*count = n;
	if (truncated)
		*truncated = cut;
	return FLACAMD_OK;
}

extern "C" int flacamd_decode_streams_host(flacamd_decoder *d, const uint8_t *flac, uint64_t flac_bytes, const flacamd_stream_desc *streams,
                                           uint32_t nstreams, void *pcm_out, uint32_t out_sample_bytes, uint64_t out_capacity_wide,
                                           uint32_t *frames, uint64_t *wide_samples, uint32_t *bad_frames)
{
	if (!d || !flac || !pcm_out)
		return fa_fail(FLACAMD_E_INVALID, "flacamd_decode_streams_host: null argument");
	if (flac_bytes > d->max_input_bytes)
		return fa_fail(FLACAMD_E_INVALID, "input of %llu bytes exceeds the decoder's capacity", (unsigned long long)flac_bytes);
	HIP_TRY(hipSetDevice(d->device));
	const size_t out_bytes = (size_t)out_capacity_wide * d->channels * out_sample_bytes;
	if (!d->stage_in)
		if (hipMalloc((void **)&d->stage_in, d->max_input_bytes + 8) != hipSuccess)
			return fa_fail(FLACAMD_E_NOMEM, "staging input: hipMalloc failed");
	if (d->stage_out_bytes < out_bytes) {
		(void)hipFree(d->stage_out);
		d->stage_out = nullptr;
		d->stage_out_bytes = 0;
		if (hipMalloc(&d->stage_out, out_bytes ? out_bytes : 16) != hipSuccess)
			return fa_fail(FLACAMD_E_NOMEM, "staging output: hipMalloc failed");
		d->stage_out_bytes = out_bytes;
	}
	HIP_TRY(hipMemcpy(d->stage_in, flac, flac_bytes, hipMemcpyHostToDevice));
	int rc = flacamd_decode_streams(d, d->stage_in, flac_bytes, streams, nstreams, d->stage_out, out_sample_bytes, out_capacity_wide, nullptr);
	if (rc)
		return rc;
	rc = flacamd_decoder_result(d, frames, wide_samples, bad_frames);
	if (rc)
		return rc;
	HIP_TRY(hipMemcpy(pcm_out, d->stage_out, out_bytes, hipMemcpyDeviceToHost));
	return FLACAMD_OK;
}

extern "C" int flacamd_decoder_set_profiling(flacamd_decoder *d, int on)
{
	if (!d)
		return fa_fail(FLACAMD_E_INVALID, "null decoder");
	d->profiling = on ? 1 : 0;
	d->have_times = 0;
	return FLACAMD_OK;
}

static const char *kDecStage[FLACAMD_DEC_STAGES] = {"sync", "frames", "crc", "output"};
extern "C" const char *flacamd_decoder_stage_name(int s) { return (s >= 0 && s < FLACAMD_DEC_STAGES) ? kDecStage[s] : ""; }

extern "C" int flacamd_decoder_stage_ms(flacamd_decoder *d, float ms[FLACAMD_DEC_STAGES])
{
	if (!d || !ms)
		return fa_fail(FLACAMD_E_INVALID, "bad arguments");
	if (!d->have_times)
		return fa_fail(FLACAMD_E_INVALID, "no profiled decode has run");
	HIP_TRY(hipSetDevice(d->device));
	HIP_TRY(hipEventSynchronize(d->ev[FLACAMD_DEC_STAGES]));
	for (int i = 0; i < FLACAMD_DEC_STAGES; i++)
		HIP_TRY(hipEventElapsedTime(&ms[i], d->ev[i], d->ev[i + 1]));
	return FLACAMD_OK;
}
