// Pieces shared by the stream encoder and decoder layers: exported format constants and string
// tables (src/libFLAC/format.c:56-211), seek-table helpers (format.c:248-314), and metadata block
// serialisation (FLAC__add_metadata_block, src/libFLAC/stream_encoder_framing.c:50-217).
#include "stream_common.h"

#include <cstdlib>
#include <cstring>

extern "C" {

const char *FLAC__VERSION_STRING = "1.2.1";
const char *FLAC__VENDOR_STRING = "reference libFLAC 1.2.1 20070917";
const FLAC__byte FLAC__STREAM_SYNC_STRING[4] = {'f', 'L', 'a', 'C'};
extern const unsigned FLAC__STREAM_SYNC = 0x664C6143;
extern const unsigned FLAC__STREAM_SYNC_LEN = 32;
const FLAC__uint64 FLAC__STREAM_METADATA_SEEKPOINT_PLACEHOLDER = 0xffffffffffffffffull;

// bit-field widths (format.c:66-140)
extern const unsigned FLAC__STREAM_METADATA_STREAMINFO_MIN_BLOCK_SIZE_LEN = 16;
extern const unsigned FLAC__STREAM_METADATA_STREAMINFO_MAX_BLOCK_SIZE_LEN = 16;
extern const unsigned FLAC__STREAM_METADATA_STREAMINFO_MIN_FRAME_SIZE_LEN = 24;
extern const unsigned FLAC__STREAM_METADATA_STREAMINFO_MAX_FRAME_SIZE_LEN = 24;
extern const unsigned FLAC__STREAM_METADATA_STREAMINFO_SAMPLE_RATE_LEN = 20;
extern const unsigned FLAC__STREAM_METADATA_STREAMINFO_CHANNELS_LEN = 3;
extern const unsigned FLAC__STREAM_METADATA_STREAMINFO_BITS_PER_SAMPLE_LEN = 5;
extern const unsigned FLAC__STREAM_METADATA_STREAMINFO_TOTAL_SAMPLES_LEN = 36;
extern const unsigned FLAC__STREAM_METADATA_STREAMINFO_MD5SUM_LEN = 128;
extern const unsigned FLAC__STREAM_METADATA_APPLICATION_ID_LEN = 32;
extern const unsigned FLAC__STREAM_METADATA_SEEKPOINT_SAMPLE_NUMBER_LEN = 64;
extern const unsigned FLAC__STREAM_METADATA_SEEKPOINT_STREAM_OFFSET_LEN = 64;
extern const unsigned FLAC__STREAM_METADATA_SEEKPOINT_FRAME_SAMPLES_LEN = 16;
extern const unsigned FLAC__STREAM_METADATA_VORBIS_COMMENT_ENTRY_LENGTH_LEN = 32;
extern const unsigned FLAC__STREAM_METADATA_VORBIS_COMMENT_NUM_COMMENTS_LEN = 32;
extern const unsigned FLAC__STREAM_METADATA_IS_LAST_LEN = 1;
extern const unsigned FLAC__STREAM_METADATA_TYPE_LEN = 7;
extern const unsigned FLAC__STREAM_METADATA_LENGTH_LEN = 24;
extern const unsigned FLAC__FRAME_HEADER_SYNC = 0x3ffe;
extern const unsigned FLAC__FRAME_HEADER_SYNC_LEN = 14;
extern const unsigned FLAC__FRAME_HEADER_RESERVED_LEN = 1;
extern const unsigned FLAC__FRAME_HEADER_BLOCKING_STRATEGY_LEN = 1;
extern const unsigned FLAC__FRAME_HEADER_BLOCK_SIZE_LEN = 4;
extern const unsigned FLAC__FRAME_HEADER_SAMPLE_RATE_LEN = 4;
extern const unsigned FLAC__FRAME_HEADER_CHANNEL_ASSIGNMENT_LEN = 4;
extern const unsigned FLAC__FRAME_HEADER_BITS_PER_SAMPLE_LEN = 3;
extern const unsigned FLAC__FRAME_HEADER_ZERO_PAD_LEN = 1;
extern const unsigned FLAC__FRAME_HEADER_CRC_LEN = 8;
extern const unsigned FLAC__FRAME_FOOTER_CRC_LEN = 16;
extern const unsigned FLAC__ENTROPY_CODING_METHOD_TYPE_LEN = 2;
extern const unsigned FLAC__ENTROPY_CODING_METHOD_PARTITIONED_RICE_ORDER_LEN = 4;
extern const unsigned FLAC__ENTROPY_CODING_METHOD_PARTITIONED_RICE_PARAMETER_LEN = 4;
extern const unsigned FLAC__ENTROPY_CODING_METHOD_PARTITIONED_RICE2_PARAMETER_LEN = 5;
extern const unsigned FLAC__ENTROPY_CODING_METHOD_PARTITIONED_RICE_RAW_LEN = 5;
extern const unsigned FLAC__ENTROPY_CODING_METHOD_PARTITIONED_RICE_ESCAPE_PARAMETER = 15;
extern const unsigned FLAC__ENTROPY_CODING_METHOD_PARTITIONED_RICE2_ESCAPE_PARAMETER = 31;
extern const unsigned FLAC__SUBFRAME_LPC_QLP_COEFF_PRECISION_LEN = 4;
extern const unsigned FLAC__SUBFRAME_LPC_QLP_SHIFT_LEN = 5;
extern const unsigned FLAC__SUBFRAME_ZERO_PAD_LEN = 1;
extern const unsigned FLAC__SUBFRAME_TYPE_LEN = 6;
extern const unsigned FLAC__SUBFRAME_WASTED_BITS_FLAG_LEN = 1;
extern const unsigned FLAC__SUBFRAME_TYPE_CONSTANT_BYTE_ALIGNED_MASK = 0x00;
extern const unsigned FLAC__SUBFRAME_TYPE_VERBATIM_BYTE_ALIGNED_MASK = 0x02;
extern const unsigned FLAC__SUBFRAME_TYPE_FIXED_BYTE_ALIGNED_MASK = 0x10;
extern const unsigned FLAC__SUBFRAME_TYPE_LPC_BYTE_ALIGNED_MASK = 0x40;

extern const char *const FLAC__EntropyCodingMethodTypeString[] = {"PARTITIONED_RICE", "PARTITIONED_RICE2"};
extern const char *const FLAC__SubframeTypeString[] = {"CONSTANT", "VERBATIM", "FIXED", "LPC"};
extern const char *const FLAC__ChannelAssignmentString[] = {"INDEPENDENT", "LEFT_SIDE", "RIGHT_SIDE", "MID_SIDE"};
extern const char *const FLAC__FrameNumberTypeString[] = {"FRAME_NUMBER_TYPE_FRAME_NUMBER", "FRAME_NUMBER_TYPE_SAMPLE_NUMBER"};
extern const char *const FLAC__MetadataTypeString[] = {"STREAMINFO", "PADDING", "APPLICATION", "SEEKTABLE", "VORBIS_COMMENT", "CUESHEET", "PICTURE"};

// format.c:96-120: cue sheet and picture field widths, in bits
extern const unsigned FLAC__STREAM_METADATA_CUESHEET_INDEX_OFFSET_LEN = 64;
extern const unsigned FLAC__STREAM_METADATA_CUESHEET_INDEX_NUMBER_LEN = 8;
extern const unsigned FLAC__STREAM_METADATA_CUESHEET_INDEX_RESERVED_LEN = 3 * 8;
extern const unsigned FLAC__STREAM_METADATA_CUESHEET_TRACK_OFFSET_LEN = 64;
extern const unsigned FLAC__STREAM_METADATA_CUESHEET_TRACK_NUMBER_LEN = 8;
extern const unsigned FLAC__STREAM_METADATA_CUESHEET_TRACK_ISRC_LEN = 12 * 8;
extern const unsigned FLAC__STREAM_METADATA_CUESHEET_TRACK_TYPE_LEN = 1;
extern const unsigned FLAC__STREAM_METADATA_CUESHEET_TRACK_PRE_EMPHASIS_LEN = 1;
extern const unsigned FLAC__STREAM_METADATA_CUESHEET_TRACK_RESERVED_LEN = 6 + 13 * 8;
extern const unsigned FLAC__STREAM_METADATA_CUESHEET_TRACK_NUM_INDICES_LEN = 8;
extern const unsigned FLAC__STREAM_METADATA_CUESHEET_MEDIA_CATALOG_NUMBER_LEN = 128 * 8;
extern const unsigned FLAC__STREAM_METADATA_CUESHEET_LEAD_IN_LEN = 64;
extern const unsigned FLAC__STREAM_METADATA_CUESHEET_IS_CD_LEN = 1;
extern const unsigned FLAC__STREAM_METADATA_CUESHEET_RESERVED_LEN = 7 + 258 * 8;
extern const unsigned FLAC__STREAM_METADATA_CUESHEET_NUM_TRACKS_LEN = 8;
extern const unsigned FLAC__STREAM_METADATA_PICTURE_TYPE_LEN = 32;
extern const unsigned FLAC__STREAM_METADATA_PICTURE_MIME_TYPE_LENGTH_LEN = 32;
extern const unsigned FLAC__STREAM_METADATA_PICTURE_DESCRIPTION_LENGTH_LEN = 32;
extern const unsigned FLAC__STREAM_METADATA_PICTURE_WIDTH_LEN = 32;
extern const unsigned FLAC__STREAM_METADATA_PICTURE_HEIGHT_LEN = 32;
extern const unsigned FLAC__STREAM_METADATA_PICTURE_DEPTH_LEN = 32;
extern const unsigned FLAC__STREAM_METADATA_PICTURE_COLORS_LEN = 32;
extern const unsigned FLAC__STREAM_METADATA_PICTURE_DATA_LENGTH_LEN = 32;
// format.c:187-209: the picture types of the ID3v2 APIC frame
extern const char *const FLAC__StreamMetadata_Picture_TypeString[] = {
	"Other",
	"32x32 pixels 'file icon' (PNG only)",
	"Other file icon",
	"Cover (front)",
	"Cover (back)",
	"Leaflet page",
	"Media (e.g. label side of CD)",
	"Lead artist/lead performer/soloist",
	"Artist/performer",
	"Conductor",
	"Band/Orchestra",
	"Composer",
	"Lyricist/text writer",
	"Recording Location",
	"During recording",
	"During performance",
	"Movie/video screen capture",
	"A bright coloured fish",
	"Illustration",
	"Band/artist logotype",
	"Publisher/Studio logotype"};

FLAC__bool FLAC__format_sample_rate_is_valid(unsigned sample_rate) { return !(sample_rate == 0 || sample_rate > FLAC__MAX_SAMPLE_RATE); }

FLAC__bool FLAC__format_blocksize_is_subset(unsigned blocksize, unsigned sample_rate)
{
	if (blocksize > 16384)
		return 0;
	if (sample_rate <= 48000 && blocksize > 4608)
		return 0;
	return 1;
}

FLAC__bool FLAC__format_sample_rate_is_subset(unsigned sample_rate)
{
	if (!FLAC__format_sample_rate_is_valid(sample_rate))
		return 0;
	if (sample_rate >= (1u << 16) && !(sample_rate % 1000 == 0 || sample_rate % 10 == 0))
		return 0;
	return 1;
}

FLAC__bool FLAC__format_seektable_is_legal(const FLAC__StreamMetadata_SeekTable *t)
{
	FLAC__uint64 prev = 0;
	for (unsigned i = 0; i < t->num_points; i++) {
		if (i > 0 && t->points[i].sample_number != FLAC__STREAM_METADATA_SEEKPOINT_PLACEHOLDER && t->points[i].sample_number <= prev)
			return 0;
		prev = t->points[i].sample_number;
	}
	return 1;
}

static int seekpoint_cmp(const void *a, const void *b)
{
	const FLAC__StreamMetadata_SeekPoint *l = (const FLAC__StreamMetadata_SeekPoint *)a, *r = (const FLAC__StreamMetadata_SeekPoint *)b;
	return l->sample_number == r->sample_number ? 0 : l->sample_number < r->sample_number ? -1 : 1;
}

unsigned FLAC__format_seektable_sort(FLAC__StreamMetadata_SeekTable *t)
{
	qsort(t->points, t->num_points, sizeof(FLAC__StreamMetadata_SeekPoint), seekpoint_cmp);
	unsigned j = 0;
	bool first = true;
	for (unsigned i = 0; i < t->num_points; i++) {
		if (t->points[i].sample_number != FLAC__STREAM_METADATA_SEEKPOINT_PLACEHOLDER && !first &&
		    t->points[i].sample_number == t->points[j - 1].sample_number)
			continue;
		first = false;
		t->points[j++] = t->points[i];
	}
	for (unsigned i = j; i < t->num_points; i++) {
		t->points[i].sample_number = FLAC__STREAM_METADATA_SEEKPOINT_PLACEHOLDER;
		t->points[i].stream_offset = 0;
		t->points[i].frame_samples = 0;
	}
	return j;
}


// Length of the UTF-8 sequence at p (at most `avail` bytes are looked at), 0 if it is malformed,
// overlong, a surrogate or U+FFFE / U+FFFF (format.c:322).
static unsigned fa_utf8_len(const FLAC__byte *p, size_t avail)
{
	auto cont = [&](unsigned i) { return i < avail && (p[i] & 0xC0) == 0x80; };
	const unsigned b = p[0];
	if (b < 0x80)
		return 1;
	if ((b & 0xE0) == 0xC0 && cont(1))
		return (b & 0xFE) == 0xC0 ? 0 : 2;
	if ((b & 0xF0) == 0xE0 && cont(1) && cont(2)) {
		if (b == 0xE0 && (p[1] & 0xE0) == 0x80)
			return 0;
		if (b == 0xED && (p[1] & 0xE0) == 0xA0)
			return 0;
		if (b == 0xEF && p[1] == 0xBF && (p[2] & 0xFE) == 0xBE)
			return 0;
		return 3;
	}
	if ((b & 0xF8) == 0xF0 && cont(1) && cont(2) && cont(3))
		return (b == 0xF0 && (p[1] & 0xF0) == 0x80) ? 0 : 4;
	if ((b & 0xFC) == 0xF8 && cont(1) && cont(2) && cont(3) && cont(4))
		return (b == 0xF8 && (p[1] & 0xF8) == 0x80) ? 0 : 5;
	if ((b & 0xFE) == 0xFC && cont(1) && cont(2) && cont(3) && cont(4) && cont(5))
		return (b == 0xFC && (p[1] & 0xFC) == 0x80) ? 0 : 6;
	return 0;
}

static bool fa_utf8_run_ok(const FLAC__byte *p, size_t n)
{
	size_t i = 0;
	while (i < n) {
		const unsigned k = fa_utf8_len(p + i, n - i);
		if (k == 0)
			return false;
		i += k;
	}
	return true;
}

// format.c:363
FLAC__bool FLAC__format_vorbiscomment_entry_name_is_legal(const char *name)
{
	for (; *name; name++)
		if (*name < 0x20 || *name == 0x3d || *name > 0x7d)
			return 0;
	return 1;
}

// format.c:372; length (unsigned)-1 means NUL-terminated
FLAC__bool FLAC__format_vorbiscomment_entry_value_is_legal(const FLAC__byte *value, unsigned length)
{
	const size_t n = length == (unsigned)(-1) ? strlen((const char *)value) : length;
	return fa_utf8_run_ok(value, n) ? 1 : 0;
}

// format.c:396
FLAC__bool FLAC__format_vorbiscomment_entry_is_legal(const FLAC__byte *entry, unsigned length)
{
	unsigned i = 0;
	for (; i < length && entry[i] != '='; i++)
		if (entry[i] < 0x20 || entry[i] > 0x7D)
			return 0;
	if (i == length)
		return 0;
	i++;
	return fa_utf8_run_ok(entry + i, length - i) ? 1 : 0;
}

// format.c:422
FLAC__bool FLAC__format_cuesheet_is_legal(const FLAC__StreamMetadata_CueSheet *cs, FLAC__bool cdda, const char **violation)
{
	auto fail = [&](const char *why) {
		if (violation)
			*violation = why;
		return 0;
	};
	if (cdda) {
		if (cs->lead_in < 2 * 44100)
			return fail("CD-DA cue sheet must have a lead-in length of at least 2 seconds");
		if (cs->lead_in % 588 != 0)
			return fail("CD-DA cue sheet lead-in length must be evenly divisible by 588 samples");
	}
	if (cs->num_tracks == 0)
		return fail("cue sheet must have at least one track (the lead-out)");
	if (cdda && cs->tracks[cs->num_tracks - 1].number != 170)
		return fail("CD-DA cue sheet must have a lead-out track number 170 (0xAA)");
	for (unsigned i = 0; i < cs->num_tracks; i++) {
		const FLAC__StreamMetadata_CueSheet_Track &t = cs->tracks[i];
		const bool lead_out = i == cs->num_tracks - 1;
		if (t.number == 0)
			return fail("cue sheet may not have a track number 0");
		if (cdda && !((t.number >= 1 && t.number <= 99) || t.number == 170))
			return fail("CD-DA cue sheet track number must be 1-99 or 170");
		if (cdda && t.offset % 588 != 0)
			return fail(lead_out ? "CD-DA cue sheet lead-out offset must be evenly divisible by 588 samples"
			                     : "CD-DA cue sheet track offset must be evenly divisible by 588 samples");
		if (!lead_out) {
			if (t.num_indices == 0)
				return fail("cue sheet track must have at least one index point");
			if (t.indices[0].number > 1)
				return fail("cue sheet track's first index number must be 0 or 1");
		}
		for (unsigned j = 0; j < t.num_indices; j++) {
			if (cdda && t.indices[j].offset % 588 != 0)
				return fail("CD-DA cue sheet track index offset must be evenly divisible by 588 samples");
			if (j > 0 && t.indices[j].number != t.indices[j - 1].number + 1)
				return fail("cue sheet track index numbers must increase by 1");
		}
	}
	return 1;
}

// format.c:501
FLAC__bool FLAC__format_picture_is_legal(const FLAC__StreamMetadata_Picture *picture, const char **violation)
{
	for (const char *p = picture->mime_type; *p; p++)
		if (*p < 0x20 || *p > 0x7e) {
			if (violation)
				*violation = "MIME type string must contain only printable ASCII characters (0x20-0x7e)";
			return 0;
		}
	if (!fa_utf8_run_ok(picture->description, strlen((const char *)picture->description))) {
		if (violation)
			*violation = "description string must be valid UTF-8";
		return 0;
	}
	return 1;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// metadata block -> bytes
// ---------------------------------------------------------------------------------------------
namespace {
struct Bytes {
	std::vector<uint8_t> &v;
	void be(uint64_t x, int nbytes) { for (int i = nbytes - 1; i >= 0; i--) v.push_back((uint8_t)(x >> (8 * i))); }
	void le32(uint32_t x) { for (int i = 0; i < 4; i++) v.push_back((uint8_t)(x >> (8 * i))); }
	void raw(const void *p, size_t n) { const uint8_t *b = (const uint8_t *)p; v.insert(v.end(), b, b + n); }
	void zeros(size_t n) { v.insert(v.end(), n, 0); }
};
}  // namespace

int fa_stream_device(void)
{
	const char *env = getenv("FLAC_AMD_DEVICE");
	return env ? atoi(env) : 0;
}

void fa_serialize_metadata(const FLAC__StreamMetadata *m, std::vector<uint8_t> &out, bool own_vendor)
{
	Bytes w{out};
	const uint32_t vlen = own_vendor ? (uint32_t)strlen(FLAC__VENDOR_STRING) : m->data.vorbis_comment.vendor_string.length;
	const void *vstr = own_vendor ? (const void *)FLAC__VENDOR_STRING : (const void *)m->data.vorbis_comment.vendor_string.entry;
	uint32_t len = m->length;
	if (m->type == FLAC__METADATA_TYPE_VORBIS_COMMENT && own_vendor) {
		// the library's own vendor string replaces the caller's (framing.c:64-69)
		len -= m->data.vorbis_comment.vendor_string.length;
		len += vlen;
	}
	w.be(((m->is_last ? 1u : 0u) << 7) | ((unsigned)m->type & 0x7f), 1);
	w.be(len, 3);
	switch (m->type) {
	case FLAC__METADATA_TYPE_STREAMINFO: {
		const FLAC__StreamMetadata_StreamInfo &s = m->data.stream_info;
		w.be(s.min_blocksize, 2);
		w.be(s.max_blocksize, 2);
		w.be(s.min_framesize, 3);
		w.be(s.max_framesize, 3);
		// 20 bits rate, 3 bits channels-1, 5 bits bps-1, 36 bits total samples
		const uint64_t packed = ((uint64_t)s.sample_rate << 44) | ((uint64_t)(s.channels - 1) << 41) | ((uint64_t)(s.bits_per_sample - 1) << 36) |
		                        (s.total_samples & 0xfffffffffull);
		w.be(packed, 8);
		w.raw(s.md5sum, 16);
		break;
	}
	case FLAC__METADATA_TYPE_PADDING:
		w.zeros(m->length);
		break;
	case FLAC__METADATA_TYPE_APPLICATION:
		w.raw(m->data.application.id, 4);
		w.raw(m->data.application.data, m->length - 4);
		break;
	case FLAC__METADATA_TYPE_SEEKTABLE:
		for (unsigned i = 0; i < m->data.seek_table.num_points; i++) {
			w.be(m->data.seek_table.points[i].sample_number, 8);
			w.be(m->data.seek_table.points[i].stream_offset, 8);
			w.be(m->data.seek_table.points[i].frame_samples, 2);
		}
		break;
	case FLAC__METADATA_TYPE_VORBIS_COMMENT:
		w.le32(vlen);
		w.raw(vstr, vlen);
		w.le32(m->data.vorbis_comment.num_comments);
		for (unsigned i = 0; i < m->data.vorbis_comment.num_comments; i++) {
			w.le32(m->data.vorbis_comment.comments[i].length);
			w.raw(m->data.vorbis_comment.comments[i].entry, m->data.vorbis_comment.comments[i].length);
		}
		break;
	case FLAC__METADATA_TYPE_CUESHEET: {
		const FLAC__StreamMetadata_CueSheet &c = m->data.cue_sheet;
		w.raw(c.media_catalog_number, 128);
		w.be(c.lead_in, 8);
		w.be(c.is_cd ? 0x80 : 0, 1);  // 1 bit flag + 7 reserved bits
		w.zeros(258);
		w.be(c.num_tracks, 1);
		for (unsigned i = 0; i < c.num_tracks; i++) {
			const FLAC__StreamMetadata_CueSheet_Track &t = c.tracks[i];
			w.be(t.offset, 8);
			w.be(t.number, 1);
			w.raw(t.isrc, 12);
			w.be(((unsigned)t.type << 7) | ((unsigned)t.pre_emphasis << 6), 1);  // + 6 reserved bits
			w.zeros(13);
			w.be(t.num_indices, 1);
			for (unsigned j = 0; j < t.num_indices; j++) {
				w.be(t.indices[j].offset, 8);
				w.be(t.indices[j].number, 1);
				w.zeros(3);
			}
		}
		break;
	}
	case FLAC__METADATA_TYPE_PICTURE: {
		const FLAC__StreamMetadata_Picture &p = m->data.picture;
		w.be((uint32_t)p.type, 4);
		size_t n = strlen(p.mime_type);
		w.be(n, 4);
		w.raw(p.mime_type, n);
		n = strlen((const char *)p.description);
		w.be(n, 4);
		w.raw(p.description, n);
		w.be(p.width, 4);
		w.be(p.height, 4);
		w.be(p.depth, 4);
		w.be(p.colors, 4);
		w.be(p.data_length, 4);
		w.raw(p.data, p.data_length);
		break;
	}
	default:
		w.raw(m->data.unknown.data, m->length);
		break;
	}
}
