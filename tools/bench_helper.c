/* Callbacks in C for bench.py's end-to-end decode line, so that the stream decoder's per-frame write callback
 * (stream_decoder.h: FLAC__StreamDecoderWriteCallback) does not run through the Python interpreter.  The write
 * callback does what the cheapest real client does: it looks at every sample once (a running sum per channel), so the
 * decoded block has to be in host memory and readable.  client_data points at a struct fa_bench_sink.
 * Built by __graft_entry__.build() into tools/bin/libbench_helper.so; plain C, no dependency on any FLAC library. */
#include <stdint.h>

struct fa_bench_sink {
	uint64_t frames, wide_samples, errors;
	int64_t sum;
};

/* leading fields of FLAC__Frame (FLAC__FrameHeader, include/FLAC/format.h:446) */
struct fa_bench_frame_head {
	unsigned blocksize, sample_rate, channels;
};

int fa_bench_write_cb(const void *decoder, const void *frame, const int32_t *const buffer[], void *client_data)
{
	struct fa_bench_sink *s = (struct fa_bench_sink *)client_data;
	const struct fa_bench_frame_head *h = (const struct fa_bench_frame_head *)frame;
	int64_t acc = 0;
	(void)decoder;
	for (unsigned c = 0; c < h->channels; c++) {
		const int32_t *p = buffer[c];
		for (unsigned i = 0; i < h->blocksize; i++)
			acc += p[i];
	}
	s->sum += acc;
	s->frames++;
	s->wide_samples += h->blocksize;
	return 0; /* FLAC__STREAM_DECODER_WRITE_STATUS_CONTINUE */
}

void fa_bench_error_cb(const void *decoder, int status, void *client_data)
{
	(void)decoder;
	(void)status;
	((struct fa_bench_sink *)client_data)->errors++;
}
