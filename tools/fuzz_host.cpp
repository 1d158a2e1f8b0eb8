// Mutation fuzzer for the host-side parsers that read untrusted files: metadata level 1 and 2
// (metadata_chain.cpp, metadata_object.cpp, stream_common.cpp), level 0 and the metadata phase of the stream decoder
// (stream_decoder.cpp, with the device entry points stubbed out) and the PCM container headers (pcm_io.cpp).
// Built with the host compiler and -fsanitize=address,undefined (see the build line in tools/fuzz_host.sh); no GPU code.
//   fuzz_host <iterations> <seed> <work dir> <base .flac files...>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdarg>
#include <string>
#include <vector>

#include "flac_amd_stream.h"
#include "flac_amd_pcm.h"

extern "C" int fa_fail(int code, const char *, ...) { return code; }
// The stream decoder's metadata phase is host code; its frame phase needs the device, which this program does not
// have: the device entry points fail, which the decoder reports as an aborted decode.
extern "C" {
int flacamd_decoder_create(uint32_t, uint32_t, uint32_t, int, uint64_t, uint32_t, flacamd_decoder **) { return FLACAMD_E_UNSUPPORTED; }
void flacamd_decoder_destroy(flacamd_decoder *) {}
int flacamd_decode_streams(flacamd_decoder *, const uint8_t *, uint64_t, const flacamd_stream_desc *, uint32_t, void *, uint32_t, uint64_t, void *)
{
	return FLACAMD_E_UNSUPPORTED;
}
int flacamd_decoder_set_frame_number_scale(flacamd_decoder *, uint32_t) { return FLACAMD_E_UNSUPPORTED; }
int flacamd_decoder_frames(flacamd_decoder *, flacamd_frame_info *, uint32_t, uint32_t *, int *) { return FLACAMD_E_UNSUPPORTED; }
int flacamd_decoder_enable_subframe_info(flacamd_decoder *, int) { return FLACAMD_E_UNSUPPORTED; }
int flacamd_decoder_set_planar_output(flacamd_decoder *, int) { return FLACAMD_E_UNSUPPORTED; }
int flacamd_decoder_subframe_info(flacamd_decoder *, uint32_t, uint32_t, flacamd_subframe_info *) { return FLACAMD_E_UNSUPPORTED; }
const char *flacamd_last_error(void) { return "no device in the fuzz build"; }
int hipMalloc(void **, size_t) { return 2; }
int hipFree(void *) { return 0; }
int hipHostMalloc(void **, size_t, unsigned) { return 2; }
int hipHostFree(void *) { return 0; }
int hipMemcpy(void *, const void *, size_t, int) { return 1; }
int hipSetDevice(int) { return 0; }
}

static uint64_t rs;
static uint32_t rnd()
{
	rs ^= rs << 13;
	rs ^= rs >> 7;
	rs ^= rs << 17;
	return (uint32_t)(rs >> 16);
}

static std::vector<uint8_t> slurp(const char *path)
{
	std::vector<uint8_t> v;
	FILE *f = fopen(path, "rb");
	if (!f) return v;
	uint8_t buf[65536];
	size_t n;
	while ((n = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
	fclose(f);
	return v;
}

static void spill(const std::string &path, const std::vector<uint8_t> &v)
{
	FILE *f = fopen(path.c_str(), "wb");
	fwrite(v.data(), 1, v.size(), f);
	fclose(f);
}

static void mutate(std::vector<uint8_t> &v, size_t span)
{
	if (span > v.size()) span = v.size();
	const uint32_t n = 1 + rnd() % 6;
	for (uint32_t i = 0; i < n && span; i++) {
		const size_t at = rnd() % span;
		switch (rnd() % 6) {
		case 0: v[at] = (uint8_t)rnd(); break;
		case 1: v[at] ^= 1u << (rnd() % 8); break;
		case 2: v[at] = 0xff; break;
		case 3: v[at] = 0; break;
		case 4: if (at + 4 <= v.size()) { const uint32_t x = rnd() % 3 ? 0xffffffffu : rnd(); memcpy(&v[at], &x, 4); } break;
		default: if (rnd() % 4 == 0) v.resize(at + 1); if (span > v.size()) span = v.size(); break;
		}
	}
}

static void walk_object(const FLAC__StreamMetadata *b)
{
	FLAC__StreamMetadata *c = FLAC__metadata_object_clone(b);
	if (c) {
		(void)FLAC__metadata_object_is_equal(b, c);
		if (c->type == FLAC__METADATA_TYPE_SEEKTABLE) (void)FLAC__format_seektable_is_legal(&c->data.seek_table);
		if (c->type == FLAC__METADATA_TYPE_CUESHEET) { const char *why; (void)FLAC__format_cuesheet_is_legal(&c->data.cue_sheet, 1, &why); }
		if (c->type == FLAC__METADATA_TYPE_PICTURE) { const char *why; (void)FLAC__format_picture_is_legal(&c->data.picture, &why); }
		if (c->type == FLAC__METADATA_TYPE_VORBIS_COMMENT)
			for (unsigned i = 0; i < c->data.vorbis_comment.num_comments; i++)
				(void)FLAC__format_vorbiscomment_entry_is_legal(c->data.vorbis_comment.comments[i].entry, c->data.vorbis_comment.comments[i].length);
		FLAC__metadata_object_delete(c);
	}
}

static void level2(const std::string &path)
{
	FLAC__Metadata_Chain *ch = FLAC__metadata_chain_new();
	if (FLAC__metadata_chain_read(ch, path.c_str())) {
		FLAC__Metadata_Iterator *it = FLAC__metadata_iterator_new();
		FLAC__metadata_iterator_init(it, ch);
		unsigned n = 0;
		do {
			walk_object(FLAC__metadata_iterator_get_block(it));
			n++;
		} while (FLAC__metadata_iterator_next(it));
		// an edit: drop one block (not the first), add padding, write back either way
		while (FLAC__metadata_iterator_prev(it)) {}
		if (n > 1 && FLAC__metadata_iterator_next(it)) (void)FLAC__metadata_iterator_delete_block(it, rnd() & 1);
		FLAC__StreamMetadata *pad = FLAC__metadata_object_new(FLAC__METADATA_TYPE_PADDING);
		pad->length = rnd() % 300;
		if (!FLAC__metadata_iterator_insert_block_after(it, pad)) FLAC__metadata_object_delete(pad);
		FLAC__metadata_iterator_delete(it);
		if (rnd() & 1) FLAC__metadata_chain_sort_padding(ch);
		if (rnd() & 1) FLAC__metadata_chain_merge_padding(ch);
		(void)FLAC__metadata_chain_check_if_tempfile_needed(ch, rnd() & 1);
		(void)FLAC__metadata_chain_write(ch, rnd() & 1, 0);
	}
	(void)FLAC__metadata_chain_status(ch);
	FLAC__metadata_chain_delete(ch);
}

static void level1(const std::string &path)
{
	FLAC__Metadata_SimpleIterator *it = FLAC__metadata_simple_iterator_new();
	if (FLAC__metadata_simple_iterator_init(it, path.c_str(), rnd() % 4 == 0, 0)) {
		unsigned n = 0;
		do {
			FLAC__byte id[4];
			(void)FLAC__metadata_simple_iterator_get_block_offset(it);
			(void)FLAC__metadata_simple_iterator_get_block_length(it);
			(void)FLAC__metadata_simple_iterator_is_last(it);
			if (FLAC__metadata_simple_iterator_get_block_type(it) == FLAC__METADATA_TYPE_APPLICATION)
				(void)FLAC__metadata_simple_iterator_get_application_id(it, id);
			FLAC__StreamMetadata *b = FLAC__metadata_simple_iterator_get_block(it);
			if (b) {
				walk_object(b);
				if (n && rnd() % 4 == 0) {
					if (b->type == FLAC__METADATA_TYPE_PADDING) b->length = rnd() % 200;
					(void)FLAC__metadata_simple_iterator_set_block(it, b, rnd() & 1);
				}
				FLAC__metadata_object_delete(b);
			}
			n++;
		} while (FLAC__metadata_simple_iterator_next(it) && n < 64);
		while (FLAC__metadata_simple_iterator_prev(it)) {}
		if (FLAC__metadata_simple_iterator_is_writable(it)) {
			FLAC__StreamMetadata *pad = FLAC__metadata_object_new(FLAC__METADATA_TYPE_PADDING);
			pad->length = rnd() % 100;
			(void)FLAC__metadata_simple_iterator_insert_block_after(it, pad, rnd() & 1);
			FLAC__metadata_object_delete(pad);
			if (FLAC__metadata_simple_iterator_next(it)) (void)FLAC__metadata_simple_iterator_delete_block(it, rnd() & 1);
		}
	}
	(void)FLAC__metadata_simple_iterator_status(it);
	FLAC__metadata_simple_iterator_delete(it);
}

static FLAC__StreamDecoderWriteStatus on_write(const FLAC__StreamDecoder *, const FLAC__Frame *, const FLAC__int32 *const[], void *)
{
	return FLAC__STREAM_DECODER_WRITE_STATUS_CONTINUE;
}
static void on_meta(const FLAC__StreamDecoder *, const FLAC__StreamMetadata *m, void *) { walk_object(m); }
static void on_error(const FLAC__StreamDecoder *, FLAC__StreamDecoderErrorStatus, void *) {}

// level 0 and the stream decoder's own metadata reader (stream_decoder.c:1423-1917)
static void level0(const std::string &path)
{
	FLAC__StreamMetadata si, *m = 0;
	(void)FLAC__metadata_get_streaminfo(path.c_str(), &si);
	if (FLAC__metadata_get_tags(path.c_str(), &m)) FLAC__metadata_object_delete(m);
	if (FLAC__metadata_get_cuesheet(path.c_str(), &m)) FLAC__metadata_object_delete(m);
	if (FLAC__metadata_get_picture(path.c_str(), &m, (FLAC__StreamMetadata_Picture_Type)-1, 0, 0, ~0u, ~0u, ~0u, ~0u)) FLAC__metadata_object_delete(m);
	FLAC__StreamDecoder *d = FLAC__stream_decoder_new();
	FLAC__stream_decoder_set_metadata_respond_all(d);
	if (rnd() & 1) FLAC__stream_decoder_set_metadata_ignore_application(d, (const FLAC__byte *)"test");
	if (FLAC__stream_decoder_init_file(d, path.c_str(), on_write, on_meta, on_error, 0) == FLAC__STREAM_DECODER_INIT_STATUS_OK)
		(void)FLAC__stream_decoder_process_until_end_of_metadata(d);
	(void)FLAC__stream_decoder_get_state(d);
	(void)FLAC__stream_decoder_finish(d);
	FLAC__stream_decoder_delete(d);
}

static void pcm_headers()
{
	static const uint32_t kinds[4] = {FLACAMD_PCM_WAVE, FLACAMD_PCM_RF64, FLACAMD_PCM_WAVE64, FLACAMD_PCM_AIFF};
	uint8_t hdr[512];
	size_t n = 0;
	const uint32_t ch = 1 + rnd() % 8, bps = 8 * (1 + rnd() % 3);
	const uint32_t mask = ch > 2 ? 0 : 0;
	if (flacamd_pcm_write_header(kinds[rnd() % 4], ch > 2 ? 2 : ch, bps, 44100, mask, 1000 + rnd() % 100000, hdr, sizeof hdr, &n)) return;
	std::vector<uint8_t> v(hdr, hdr + n);
	v.resize(n + rnd() % 64, 0x55);
	mutate(v, v.size());
	flacamd_pcm_format f;
	(void)flacamd_pcm_parse_header(v.data(), v.size(), rnd() & 1, &f);
}

int main(int argc, char **argv)
{
	if (argc < 5) return 2;
	const long iters = atol(argv[1]);
	rs = 0x9e3779b97f4a7c15ull ^ (uint64_t)atol(argv[2]) * 0x100000001b3ull;
	const std::string dir = argv[3];
	std::vector<std::vector<uint8_t>> bases;
	for (int i = 4; i < argc; i++) bases.push_back(slurp(argv[i]));
	const std::string path = dir + "/fz_work.flac";
	for (long i = 0; i < iters; i++) {
		std::vector<uint8_t> v = bases[rnd() % bases.size()];
		mutate(v, v.size() > 1600 ? v.size() - 1600 : v.size());  // the metadata, not the stand-in audio
		spill(path, v);
		level2(path);
		spill(path, v);
		level1(path);
		spill(path, v);
		level0(path);
		pcm_headers();
		if (i % 2000 == 0) fprintf(stderr, "iteration %ld\n", i);
	}
	printf("done: %ld iterations\n", iters);
	return 0;
}
