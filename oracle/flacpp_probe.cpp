// Test infrastructure: drives the reference's libFLAC++ classes (compiled from the reference's own
// sources, linked against the product library) through one encode and one decode in memory.
// usage: flacpp_probe <wide samples>   prints "ok <bytes>" when the round trip is exact.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "FLAC++/decoder.h"
#include "FLAC++/encoder.h"

static std::vector<FLAC__byte> g_stream;
static size_t g_read_pos = 0;

class Enc : public FLAC::Encoder::Stream {
protected:
	::FLAC__StreamEncoderWriteStatus write_callback(const FLAC__byte buffer[], size_t bytes, unsigned, unsigned)
	{
		g_stream.insert(g_stream.end(), buffer, buffer + bytes);
		return ::FLAC__STREAM_ENCODER_WRITE_STATUS_OK;
	}
};

class Dec : public FLAC::Decoder::Stream {
public:
	std::vector<FLAC__int32> pcm;
	unsigned errors;
	Dec() : errors(0) {}

protected:
	::FLAC__StreamDecoderReadStatus read_callback(FLAC__byte buffer[], size_t *bytes)
	{
		size_t n = g_stream.size() - g_read_pos;
		if (n > *bytes)
			n = *bytes;
		if (n == 0) {
			*bytes = 0;
			return ::FLAC__STREAM_DECODER_READ_STATUS_END_OF_STREAM;
		}
		memcpy(buffer, &g_stream[g_read_pos], n);
		g_read_pos += n;
		*bytes = n;
		return ::FLAC__STREAM_DECODER_READ_STATUS_CONTINUE;
	}
	::FLAC__StreamDecoderWriteStatus write_callback(const ::FLAC__Frame *frame, const FLAC__int32 *const buffer[])
	{
		for (unsigned i = 0; i < frame->header.blocksize; i++)
			for (unsigned c = 0; c < frame->header.channels; c++)
				pcm.push_back(buffer[c][i]);
		return ::FLAC__STREAM_DECODER_WRITE_STATUS_CONTINUE;
	}
	void error_callback(::FLAC__StreamDecoderErrorStatus) { errors++; }
};

int main(int argc, char **argv)
{
	const unsigned wide = argc > 1 ? (unsigned)atoi(argv[1]) : 20000;
	std::vector<FLAC__int32> pcm(wide * 2);
	unsigned x = 12345;
	int a = 0, b = 0;
	for (unsigned i = 0; i < wide; i++) {
		x = x * 1664525u + 1013904223u;
		a += (int)((x >> 20) & 63) - 31;
		b = (a * 3) / 4 + (int)((x >> 8) & 15);
		pcm[2 * i] = a % 20000;
		pcm[2 * i + 1] = b % 20000;
	}
	Enc enc;
	if (!enc.set_channels(2) || !enc.set_bits_per_sample(16) || !enc.set_sample_rate(44100) || !enc.set_compression_level(8) || !enc.set_verify(true)) {
		fprintf(stderr, "a setter failed\n");
		return 1;
	}
	if (enc.init() != ::FLAC__STREAM_ENCODER_INIT_STATUS_OK) {
		fprintf(stderr, "encoder init failed\n");
		return 1;
	}
	if (!enc.process_interleaved(&pcm[0], wide) || !enc.finish()) {
		fprintf(stderr, "encode failed: %s\n", enc.get_state().as_cstring());
		return 1;
	}
	Dec dec;
	if (dec.init() != ::FLAC__STREAM_DECODER_INIT_STATUS_OK || !dec.process_until_end_of_stream() || !dec.finish()) {
		fprintf(stderr, "decode failed\n");
		return 1;
	}
	if (dec.errors || dec.pcm != pcm) {
		fprintf(stderr, "round trip differs (%u errors, %zu of %zu samples)\n", dec.errors, dec.pcm.size(), pcm.size());
		return 1;
	}
	printf("ok %zu\n", g_stream.size());
	return 0;
}
