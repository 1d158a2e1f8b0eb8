// PCM container headers on the host: WAVE / RF64 / Wave64 / AIFF / AIFF-C parsing and writing.
//
// The parser walks the chunks of a file held in memory the way the reference's command line walks
// them on a FILE (src/flac/encode.c:195-900) and refuses what it refuses; the writer produces the
// bytes of src/flac/decode.c:536-932.  No sample data is touched here.
#include "flac_amd_pcm.h"
#include "flac_amd_internal.h"

#include <cstring>

namespace {

struct Cursor {
	const uint8_t *p;
	size_t size, at;
	bool short_read;  // the header runs past the bytes we were given

	bool need(size_t n)
	{
		if (size - at < n) {
			short_read = true;
			return false;
		}
		return true;
	}
	bool u16(bool be, uint32_t *v)
	{
		if (!need(2)) return false;
		*v = be ? ((uint32_t)p[at] << 8 | p[at + 1]) : ((uint32_t)p[at + 1] << 8 | p[at]);
		at += 2;
		return true;
	}
	bool u32(bool be, uint32_t *v)
	{
		if (!need(4)) return false;
		*v = be ? ((uint32_t)p[at] << 24 | (uint32_t)p[at + 1] << 16 | (uint32_t)p[at + 2] << 8 | p[at + 3])
		        : ((uint32_t)p[at + 3] << 24 | (uint32_t)p[at + 2] << 16 | (uint32_t)p[at + 1] << 8 | p[at]);
		at += 4;
		return true;
	}
	bool u64le(uint64_t *v)
	{
		uint32_t lo, hi;
		if (!u32(false, &lo) || !u32(false, &hi)) return false;
		*v = (uint64_t)hi << 32 | lo;
		return true;
	}
	bool skip(uint64_t n)
	{
		if ((uint64_t)(size - at) < n) {
			short_read = true;
			return false;
		}
		at += (size_t)n;
		return true;
	}
};

const uint8_t kW64Tail[12] = {0xF3, 0xAC, 0xD3, 0x11, 0xD1, 0x8C, 0x00, 0xC0, 0x4F, 0x8E, 0xDB, 0x8A};

uint32_t popcount32(uint32_t m)
{
	uint32_t n = 0;
	for (; m; m &= m - 1) n++;
	return n;
}

int short_header() { return fa_fail(FLACAMD_E_OVERFLOW, "the container header runs past the bytes supplied"); }

// src/flac/encode.c:195-672
int parse_wave(Cursor &c, uint32_t container, int map_none, flacamd_pcm_format *o)
{
	const bool w64 = container == FLACAMD_PCM_WAVE64;
	bool got_fmt = false, got_data = false, got_ds64 = false;
	uint64_t ds64_data = 0;
	uint32_t bps = 0, shift = 0, channels = 0, rate = 0, mask = 0;
	o->is_unsigned = 0;
	o->is_big_endian = 0;
	// past "RIFFxxxxWAVE" / "RF64xxxxWAVE", or the riff GUID + size + wave GUID
	if (!c.skip(w64 ? 16 + 8 + 16 : 12)) return short_header();
	while (c.at < c.size && !got_data) {
		uint8_t id[16] = {0};
		const size_t idn = w64 ? 16 : 4;
		if (c.size - c.at < idn)
			return fa_fail(FLACAMD_E_INVALID, "incomplete chunk identifier");
		memcpy(id, c.p + c.at, idn);
		c.at += idn;
		if (container == FLACAMD_PCM_RF64 && !memcmp(id, "ds64", 4)) {
			uint32_t n, x;
			if (got_ds64) return fa_fail(FLACAMD_E_INVALID, "file has multiple 'ds64' chunks");
			if (got_fmt || got_data) return fa_fail(FLACAMD_E_INVALID, "'ds64' chunk appears after 'fmt ' or 'data' chunk");
			if (!c.u32(false, &n)) return short_header();
			if (n < 28) return fa_fail(FLACAMD_E_INVALID, "non-standard 'ds64' chunk has length = %u", n);
			n += n & 1;
			if (!c.u32(false, &x) || !c.u32(false, &x) || !c.u64le(&ds64_data)) return short_header();
			if (!c.skip(n - 16)) return short_header();
			got_ds64 = true;
		} else if (!memcmp(id, "fmt ", 4) && (!w64 || !memcmp(id + 4, kW64Tail, 12))) {
			uint32_t n, x, tag, block_align;
			if (got_fmt) return fa_fail(FLACAMD_E_INVALID, "file has multiple 'fmt ' chunks");
			if (!c.u32(false, &n)) return short_header();
			if (w64) {
				if (!c.u32(false, &x)) return short_header();
				if (x) return fa_fail(FLACAMD_E_INVALID, "freakishly large Wave64 'fmt ' chunk");
				if (n < 16 + 8) return fa_fail(FLACAMD_E_INVALID, "freakishly small Wave64 'fmt ' chunk");
				n -= 16 + 8;
			}
			if (n < 16) return fa_fail(FLACAMD_E_INVALID, "non-standard 'fmt ' chunk has length = %u", n);
			if (!w64) n += n & 1;
			else n = (n + 7) & ~7u;
			if (!c.u16(false, &tag)) return short_header();
			if (tag != 1 && tag != 65534) return fa_fail(FLACAMD_E_INVALID, "unsupported format type %u", tag);
			if (!c.u16(false, &channels) || !c.u32(false, &rate) || !c.u32(false, &x) || !c.u16(false, &block_align) || !c.u16(false, &bps))
				return short_header();
			(void)block_align;
			o->is_unsigned = bps <= 8;
			if (tag == 1) {
				if (bps != 8 && bps != 16 && bps != 24 && bps != 32)
					return fa_fail(FLACAMD_E_INVALID, "legacy WAVE file has format type %u but bits-per-sample=%u", tag, bps);
				shift = 0;
				if (channels > 2 && !map_none)
					return fa_fail(FLACAMD_E_INVALID, "WAVE has >2 channels but is not WAVE_FORMAT_EXTENSIBLE; cannot assign channels");
				n -= 16;
			} else {
				if (n < 40) return fa_fail(FLACAMD_E_INVALID, "invalid WAVEFORMATEXTENSIBLE chunk with size %u", n);
				if (!c.u16(false, &x)) return short_header();
				if (x < 22) return fa_fail(FLACAMD_E_INVALID, "invalid WAVEFORMATEXTENSIBLE chunk with cbSize %u", x);
				if (!c.u16(false, &x)) return short_header();
				if (x > bps)
					return fa_fail(FLACAMD_E_INVALID, "invalid WAVEFORMATEXTENSIBLE chunk with wValidBitsPerSample (%u) > wBitsPerSample (%u)", x, bps);
				shift = bps - x;
				if (!c.u32(false, &mask)) return short_header();
				if (mask == 0) {
					if (channels == 1) mask = 0x0001;
					else if (channels == 2) mask = 0x0003;
				}
				const bool known = mask == 0x0001 || mask == 0x0003 || mask == 0x0007 || mask == 0x0033 || mask == 0x0603 || mask == 0x0037 ||
				                   mask == 0x0607 || mask == 0x003f || mask == 0x060f;
				if (!map_none && !known)
					return fa_fail(FLACAMD_E_INVALID, "WAVEFORMATEXTENSIBLE chunk with unsupported channel mask=0x%04X", mask);
				if (!map_none) {
					if (popcount32(mask) < channels)
						return fa_fail(FLACAMD_E_INVALID, "channel mask 0x%04X has unassigned channels (#channels=%u)", mask, channels);
					if (popcount32(mask) > channels)
						return fa_fail(FLACAMD_E_INVALID, "channel mask 0x%04X has extra bits for non-existant channels (#channels=%u)", mask, channels);
				}
				if (!c.u16(false, &x)) return short_header();
				if (x != 1) return fa_fail(FLACAMD_E_INVALID, "unsupported WAVEFORMATEXTENSIBLE chunk with non-PCM format %u", x);
				n -= 26;
			}
			o->bytes_per_wide_sample = channels * (bps / 8);
			if (!c.skip(n)) return short_header();
			got_fmt = true;
		} else if (!memcmp(id, "data", 4) && (!w64 || !memcmp(id + 4, kW64Tail, 12))) {
			uint64_t n;
			if (!got_fmt) return fa_fail(FLACAMD_E_INVALID, "got 'data' chunk before 'fmt' chunk");
			if (!w64) {
				uint32_t x;
				if (!c.u32(false, &x)) return short_header();
				n = x;
			} else {
				if (!c.u64le(&n)) return short_header();
				if (n < 16 + 8) return fa_fail(FLACAMD_E_INVALID, "freakishly small Wave64 'data' chunk");
				n -= 16 + 8;
			}
			if (container == FLACAMD_PCM_RF64) {
				if (!got_ds64) return fa_fail(FLACAMD_E_INVALID, "RF64 file has no 'ds64' chunk before 'data' chunk");
				if (n == 0xffffffffull) n = ds64_data;
			}
			if (n == 0) return fa_fail(FLACAMD_E_INVALID, "'data' chunk has size of 0");
			o->data_bytes = n;
			o->data_offset = c.at;
			got_data = true;
		} else {
			uint64_t n;
			if (!w64) {
				uint32_t x;
				if (!c.u32(false, &x)) return short_header();
				n = x;
				n += n & 1;
			} else {
				if (!c.u64le(&n)) return short_header();
				n = (n + 7) & ~(uint64_t)7;
				if (n < 16 + 8) return fa_fail(FLACAMD_E_INVALID, "freakishly small Wave64 chunk");
				n -= 16 + 8;
			}
			if (!c.skip(n)) return short_header();
		}
	}
	if (!got_fmt) return fa_fail(FLACAMD_E_INVALID, "didn't find fmt chunk");
	if (!got_data) return c.short_read ? short_header() : fa_fail(FLACAMD_E_INVALID, "didn't find data chunk");
	o->sample_rate = rate;
	o->channels = channels;
	o->bits_per_sample = bps;
	o->shift = shift;
	o->channel_mask = mask;
	return 0;
}

// read_sane_extended (src/flac/encode.c:2784): an 80-bit extended float as a rounded integer
bool sane_extended(Cursor &c, uint32_t *val, bool *bad)
{
	*bad = false;
	if (!c.need(10)) return false;
	const uint8_t *b = c.p + c.at;
	c.at += 10;
	const int e = (int)(int16_t)(((uint16_t)b[0] << 8 | b[1]) - 0x3FFF);
	const int shift = 63 - e;
	if ((b[0] >> 7) == 1 || e < 0 || e > 63) {
		*bad = true;
		return false;
	}
	uint64_t p = 0;
	for (int i = 0; i < 8; i++)
		p |= (uint64_t)b[i + 2] << (56 - i * 8);
	*val = (uint32_t)((p >> shift) + ((shift > 0 ? p >> (shift - 1) : 0) & 1));
	return true;
}

// src/flac/encode.c:674-900
int parse_aiff(Cursor &c, uint32_t container, int map_none, flacamd_pcm_format *o)
{
	const bool aifc = container == FLACAMD_PCM_AIFF_C;
	bool got_comm = false, got_ssnd = false;
	uint32_t bps = 0, shift = 0, channels = 0, rate = 0, frames = 0;
	o->is_unsigned = 0;
	o->is_big_endian = 1;
	if (!c.skip(12)) return short_header();
	while (c.at < c.size && !got_ssnd) {
		if (c.size - c.at < 4) return fa_fail(FLACAMD_E_INVALID, "incomplete chunk identifier");
		const uint8_t *id = c.p + c.at;
		c.at += 4;
		if (!memcmp(id, "COMM", 4)) {
			uint32_t n, x;
			const uint32_t min_size = aifc ? 22 : 18;
			if (got_comm) return fa_fail(FLACAMD_E_INVALID, "file has multiple 'COMM' chunks");
			if (!c.u32(true, &n)) return short_header();
			if (n < min_size) return fa_fail(FLACAMD_E_INVALID, "non-standard %s 'COMM' chunk has length = %u", aifc ? "AIFF-C" : "AIFF", n);
			const uint64_t skip = (n - min_size) + (n & 1);
			if (!c.u16(true, &channels)) return short_header();
			if (channels > 2 && !map_none) return fa_fail(FLACAMD_E_INVALID, "unsupported number of channels %u for AIFF", channels);
			if (!c.u32(true, &frames) || !c.u16(true, &bps)) return short_header();
			shift = (bps % 8) ? 8 - (bps % 8) : 0;
			bps += shift;
			bool bad;
			if (!sane_extended(c, &rate, &bad)) return bad ? fa_fail(FLACAMD_E_INVALID, "invalid floating-point value") : short_header();
			if (aifc) {
				if (!c.u32(true, &x)) return short_header();
				if (x == 0x736F7774) o->is_big_endian = 0;  // "sowt"
				else if (x != 0x4E4F4E45)                   // "NONE"
					return fa_fail(FLACAMD_E_INVALID, "can't handle AIFF-C compression type 0x%08X", x);
			}
			if (!(map_none || channels == 1 || channels == 2 || channels == 3 || channels == 5))
				return fa_fail(FLACAMD_E_INVALID, "unsupported number of channels %u for AIFF", channels);
			o->bytes_per_wide_sample = channels * (bps / 8);
			if (!c.skip(skip)) return short_header();
			got_comm = true;
		} else if (!memcmp(id, "SSND", 4)) {
			uint32_t x, offset;
			uint64_t n;
			if (!got_comm) return fa_fail(FLACAMD_E_INVALID, "got 'SSND' chunk before 'COMM' chunk");
			if (!c.u32(true, &x)) return short_header();
			n = x;
			if (n <= 8) return fa_fail(FLACAMD_E_INVALID, "'SSND' chunk has size <= 8");
			n -= 8;
			if (!c.u32(true, &offset)) return short_header();
			n -= offset;
			if (!c.u32(true, &x)) return short_header();
			if (x) n -= (x - (n % x));
			// the offset bytes sit between the chunk header and the first sample
			o->data_offset = c.at + offset;
			o->data_bytes = n;
			got_ssnd = true;
		} else {
			uint32_t x;
			if (!c.u32(true, &x)) return short_header();
			if (!c.skip((uint64_t)x + (x & 1))) return short_header();
		}
	}
	if (!got_comm) return c.short_read ? short_header() : fa_fail(FLACAMD_E_INVALID, "didn't find COMM chunk");
	if (!got_ssnd && frames) return c.short_read ? short_header() : fa_fail(FLACAMD_E_INVALID, "didn't find SSND chunk");
	o->sample_rate = rate;
	o->channels = channels;
	o->bits_per_sample = bps;
	o->shift = shift;
	o->channel_mask = 0;
	return 0;
}

struct Out {
	uint8_t *p;
	size_t cap, n;
	void bytes(const void *s, size_t k)
	{
		if (p && n + k <= cap) memcpy(p + n, s, k);
		n += k;
	}
	void le16(uint32_t v) { const uint8_t b[2] = {(uint8_t)v, (uint8_t)(v >> 8)}; bytes(b, 2); }
	void le32(uint32_t v) { const uint8_t b[4] = {(uint8_t)v, (uint8_t)(v >> 8), (uint8_t)(v >> 16), (uint8_t)(v >> 24)}; bytes(b, 4); }
	void le64(uint64_t v) { le32((uint32_t)v); le32((uint32_t)(v >> 32)); }
	void be16(uint32_t v) { const uint8_t b[2] = {(uint8_t)(v >> 8), (uint8_t)v}; bytes(b, 2); }
	void be32(uint32_t v) { const uint8_t b[4] = {(uint8_t)(v >> 24), (uint8_t)(v >> 16), (uint8_t)(v >> 8), (uint8_t)v}; bytes(b, 4); }
};

}  // namespace

extern "C" int flacamd_pcm_parse_header(const uint8_t *file, size_t size, int channel_map_none, flacamd_pcm_format *out)
{
	if (!file || !out)
		return fa_fail(FLACAMD_E_INVALID, "flacamd_pcm_parse_header: null argument");
	memset(out, 0, sizeof *out);
	if (size < 12)
		return fa_fail(FLACAMD_E_INVALID, "no supported container header in the first %zu bytes", size);
	// src/flac/main.c:1715-1729
	uint32_t container;
	if (!memcmp(file, "ID3", 3))
		return fa_fail(FLACAMD_E_INVALID, "input has an ID3v2 tag");
	else if (!memcmp(file, "RIFF", 4) && !memcmp(file + 8, "WAVE", 4)) container = FLACAMD_PCM_WAVE;
	else if (!memcmp(file, "RF64", 4) && !memcmp(file + 8, "WAVE", 4)) container = FLACAMD_PCM_RF64;
	else if (!memcmp(file, "riff\x2E\x91\xCF\x11\xD6\xA5\x28\xDB", 12)) container = FLACAMD_PCM_WAVE64;
	else if (!memcmp(file, "FORM", 4) && !memcmp(file + 8, "AIFF", 4)) container = FLACAMD_PCM_AIFF;
	else if (!memcmp(file, "FORM", 4) && !memcmp(file + 8, "AIFC", 4)) container = FLACAMD_PCM_AIFF_C;
	else
		return fa_fail(FLACAMD_E_INVALID, "no supported container header: raw samples");
	out->container = container;
	Cursor c = {file, size, 0, false};
	const int rc = (container == FLACAMD_PCM_AIFF || container == FLACAMD_PCM_AIFF_C) ? parse_aiff(c, container, channel_map_none, out)
	                                                                                   : parse_wave(c, container, channel_map_none, out);
	if (rc)
		return rc;
	// what EncoderSession_init_encoder and format_input accept (src/flac/encode.c:2347-2410)
	if (out->channels == 0 || out->channels > FLACAMD_MAX_CHANNELS)
		return fa_fail(FLACAMD_E_INVALID, "unsupported number of channels %u", out->channels);
	if (out->bits_per_sample != 8 && out->bits_per_sample != 16 && out->bits_per_sample != 24)
		return fa_fail(FLACAMD_E_INVALID, "unsupported bits-per-sample %u", out->bits_per_sample);
	if (out->bits_per_sample - out->shift < 4)
		return fa_fail(FLACAMD_E_INVALID, "unsupported bits-per-sample %u", out->bits_per_sample - out->shift);
	out->total_wide_samples = out->bytes_per_wide_sample ? out->data_bytes / out->bytes_per_wide_sample : 0;
	return 0;
}

extern "C" int flacamd_pcm_write_header(uint32_t container, uint32_t channels, uint32_t bps, uint32_t sample_rate, uint32_t channel_mask,
                                        uint64_t samples, uint8_t *out, size_t capacity, size_t *bytes)
{
	if (!bytes)
		return fa_fail(FLACAMD_E_INVALID, "flacamd_pcm_write_header: null argument");
	const bool wave_family = container == FLACAMD_PCM_WAVE || container == FLACAMD_PCM_WAVE64 || container == FLACAMD_PCM_RF64;
	if (!wave_family && container != FLACAMD_PCM_AIFF && container != FLACAMD_PCM_AIFF_C)
		return fa_fail(FLACAMD_E_INVALID, "flacamd_pcm_write_header: container %u has no header", container);
	if (channels == 0 || bps == 0 || sample_rate == 0)
		return fa_fail(FLACAMD_E_INVALID, "flacamd_pcm_write_header: bad stream parameters");
	// decode.c:544-556
	const bool ext = wave_family && (channel_mask == 2 || channel_mask > 3 || bps % 8 || channels > 2);
	const uint64_t data_size = samples * channels * ((bps + 7) / 8);
	const uint64_t aligned = container == FLACAMD_PCM_WAVE64 ? (data_size + 7) & ~(uint64_t)7 : (data_size + 1) & ~(uint64_t)1;
	uint64_t iff_size;
	if (samples == 0) iff_size = 0;
	else if (container == FLACAMD_PCM_WAVE || container == FLACAMD_PCM_RF64) iff_size = 4 + (container == FLACAMD_PCM_RF64 ? 36 : 0) + 8 + (ext ? 40 : 16) + 8 + aligned;
	else if (container == FLACAMD_PCM_WAVE64) iff_size = 16 + 8 + 16 + 16 + 8 + (ext ? 40 : 16) + 16 + 8 + aligned;
	else iff_size = 46 + aligned;
	if (container != FLACAMD_PCM_WAVE64 && container != FLACAMD_PCM_RF64 && iff_size >= 0xFFFFFFF4ull)
		return fa_fail(FLACAMD_E_INVALID, "stream is too big to fit in a single file of this format");
	Out o = {out, out ? capacity : 0, 0};
	if (wave_family) {
		if (container == FLACAMD_PCM_WAVE) {
			o.bytes("RIFF", 4);
			o.le32((uint32_t)iff_size);
			o.bytes("WAVE", 4);
		} else if (container == FLACAMD_PCM_WAVE64) {
			o.bytes("\x72\x69\x66\x66\x2E\x91\xCF\x11\xD6\xA5\x28\xDB\x04\xC1\x00\x00", 16);
			o.le64(iff_size);
			o.bytes("\x77\x61\x76\x65\xF3\xAC\xD3\x11\xD1\x8C\x00\xC0\x4F\x8E\xDB\x8A", 16);
		} else {
			o.bytes("RF64", 4);
			o.le32(0xffffffffu);
			o.bytes("WAVE", 4);
			o.bytes("ds64", 4);
			o.le32(28);
			o.le64(iff_size);
			o.le64(data_size);
			o.le64(samples);
			o.le32(0);
		}
		if (container != FLACAMD_PCM_WAVE64) {
			o.bytes("fmt ", 4);
			o.le32(ext ? 40 : 16);
		} else {
			o.bytes("\x66\x6D\x74\x20\xF3\xAC\xD3\x11\xD1\x8C\x00\xC0\x4F\x8E\xDB\x8A", 16);
			o.le64(16 + 8 + (ext ? 40 : 16));
		}
		// write_riff_wave_fmt_chunk_body (decode.c:785)
		o.le16(ext ? 65534 : 1);
		o.le16(channels);
		o.le32(sample_rate);
		o.le32(sample_rate * channels * ((bps + 7) / 8));
		o.le16(channels * ((bps + 7) / 8));
		o.le16(((bps + 7) / 8) * 8);
		if (ext) {
			o.le16(22);
			o.le16(bps);
			o.le32(channel_mask);
			o.bytes("\x01\x00\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71", 16);
		}
		if (container != FLACAMD_PCM_WAVE64) {
			o.bytes("data", 4);
			o.le32(container == FLACAMD_PCM_RF64 ? 0xffffffffu : (uint32_t)data_size);
		} else {
			o.bytes("\x64\x61\x74\x61\xF3\xAC\xD3\x11\xD1\x8C\x00\xC0\x4F\x8E\xDB\x8A", 16);
			o.le64(16 + 8 + data_size);
		}
	} else {
		o.bytes("FORM", 4);
		o.be32((uint32_t)iff_size);
		o.bytes("AIFF", 4);
		// write_aiff_form_comm_chunk (decode.c:823), write_sane_extended (decode.c:903)
		o.bytes("COMM", 4);
		o.be32(18);
		o.be16(channels);
		o.be32((uint32_t)samples);
		o.be16(bps);
		uint32_t val = sample_rate, sh = 0;
		while ((val >> (31 - sh)) == 0) sh++;
		val <<= sh;
		o.be16(63 - (sh + 32) + 0x3FFF);
		o.be32(val);
		o.be32(0);
		o.bytes("SSND", 4);
		o.be32((uint32_t)data_size + 8);
		o.be32(0);
		o.be32(0);
	}
	*bytes = o.n;
	if (out && o.n > capacity)
		return fa_fail(FLACAMD_E_OVERFLOW, "header needs %zu bytes", o.n);
	return 0;
}
