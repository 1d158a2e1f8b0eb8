// libFLAC's metadata object API (include/FLAC/metadata.h:1250-2175; src/libFLAC/metadata_object.c):
// constructors, deep copies, comparison and the per-type editing helpers for the blocks that sit
// in front of the frames.  Host code only.  Objects are plain malloc()ed C structures, because
// clients free pieces of them with free() and hand over buffers they malloc()ed themselves
// (copy == false).  `length` is kept equal to the block's serialised size after every edit.
//
// Level 0 (FLAC__metadata_get_*) is built on this library's own stream decoder, which parses
// metadata on the host without touching the GPU.
#include "stream_common.h"

#include <cstdlib>
#include <cstring>
#include <strings.h>

namespace {

typedef FLAC__StreamMetadata Block;
typedef FLAC__StreamMetadata_VorbisComment_Entry Entry;
typedef FLAC__StreamMetadata_CueSheet_Track Track;
typedef FLAC__StreamMetadata_CueSheet_Index Index;
typedef FLAC__StreamMetadata_SeekPoint SeekPoint;

const SeekPoint kPlaceholder = {0xffffffffffffffffull, 0, 0};

// malloc'ed copy of n bytes (0 when there is nothing to copy); ok tells the two apart from failure
void *dup_bytes(const void *from, size_t n, bool &ok)
{
	ok = true;
	if (n == 0 || from == 0)
		return 0;
	void *x = malloc(n);
	if (!x) {
		ok = false;
		return 0;
	}
	memcpy(x, from, n);
	return x;
}

// Resize a malloc'ed array of T from `have` to `want` elements; new elements are set to `fill`.
// An array of no elements is a null pointer.  On failure nothing has changed.
template <typename T>
bool resize_array(T *&arr, unsigned have, unsigned want, const T &fill)
{
	if (want == 0) {
		free(arr);
		arr = 0;
		return true;
	}
	if ((size_t)want > SIZE_MAX / sizeof(T))
		return false;
	T *x = (T *)realloc(arr, (size_t)want * sizeof(T));
	if (!x)
		return false;
	for (unsigned i = have; i < want; i++)
		memcpy(&x[i], &fill, sizeof(T));  // bytes, so that padding is as defined as the pattern's
	arr = x;
	return true;
}

// entry text is always kept NUL-terminated one past `length` (metadata_object.c:126)
bool dup_entry(Entry &to, const Entry &from)
{
	Entry e;
	e.length = from.length;
	e.entry = 0;
	if (from.entry) {
		e.entry = (FLAC__byte *)malloc((size_t)from.length + 1);
		if (!e.entry)
			return false;
		memcpy(e.entry, from.entry, from.length);
		e.entry[from.length] = 0;
	}
	to = e;
	return true;
}

bool dup_track(Track &to, const Track &from)
{
	Track t = from;
	if (from.indices) {
		bool ok;
		t.indices = (Index *)dup_bytes(from.indices, (size_t)from.num_indices * sizeof(Index), ok);
		if (!ok)
			return false;
	}
	to = t;
	return true;
}

void free_entries(Entry *arr, unsigned n)
{
	for (unsigned i = 0; i < n; i++)
		free(arr[i].entry);
	free(arr);
}

void free_tracks(Track *arr, unsigned n)
{
	for (unsigned i = 0; i < n; i++)
		free(arr[i].indices);
	free(arr);
}

// serialised sizes (metadata_object.c:162,190,311)
void relength_seektable(Block *b) { b->length = b->data.seek_table.num_points * FLAC__STREAM_METADATA_SEEKPOINT_LENGTH; }

void relength_vorbiscomment(Block *b)
{
	const FLAC__StreamMetadata_VorbisComment &vc = b->data.vorbis_comment;
	unsigned n = 4 + vc.vendor_string.length + 4;
	for (unsigned i = 0; i < vc.num_comments; i++)
		n += 4 + vc.comments[i].length;
	b->length = n;
}

void relength_cuesheet(Block *b)
{
	const FLAC__StreamMetadata_CueSheet &cs = b->data.cue_sheet;
	unsigned n = (128 * 8 + 64 + 1 + 7 + 258 * 8 + 8) / 8;          // catalog number, lead-in, flags, reserved, track count
	n += cs.num_tracks * ((64 + 8 + 12 * 8 + 1 + 1 + 6 + 13 * 8 + 8) / 8);  // per track
	for (unsigned i = 0; i < cs.num_tracks; i++)
		n += cs.tracks[i].num_indices * ((64 + 8 + 3 * 8) / 8);      // per index
	b->length = n;
}

// metadata_object.c:249: put src into *dest, by copy or by taking the caller's buffer over
bool store_entry(Block *b, Entry *dest, const Entry *src, bool copy)
{
	FLAC__byte *old = dest->entry;
	if (src->entry && src->length > 0) {
		if (copy) {
			if (!dup_entry(*dest, *src))
				return false;
		} else {
			// the buffer becomes ours: make room for the terminator
			FLAC__byte *x = (FLAC__byte *)realloc(src->entry, (size_t)src->length + 1);
			if (!x)
				return false;
			x[src->length] = 0;
			dest->length = src->length;
			dest->entry = x;
		}
	} else
		*dest = *src;
	free(old);
	relength_vorbiscomment(b);
	return true;
}

int find_comment(const Block *b, unsigned from, const char *name, unsigned name_len)
{
	const FLAC__StreamMetadata_VorbisComment &vc = b->data.vorbis_comment;
	for (unsigned i = from; i < vc.num_comments; i++)
		if (FLAC__metadata_object_vorbiscomment_entry_matches(vc.comments[i], name, name_len))
			return (int)i;
	return -1;
}

bool same_bytes(const void *a, const void *b, size_t n)
{
	if (a && b)
		return memcmp(a, b, n) == 0;
	return a == b;
}

bool same_cstring(const char *a, const char *b) { return a == b || (a && b && strcmp(a, b) == 0); }

}  // namespace

extern "C" {

// metadata_object.c:433
FLAC__StreamMetadata *FLAC__metadata_object_new(FLAC__MetadataType type)
{
	if ((unsigned)type > FLAC__MAX_METADATA_TYPE_CODE)
		return 0;
	Block *b = (Block *)calloc(1, sizeof(Block));
	if (!b)
		return 0;
	b->type = type;
	switch (type) {
	case FLAC__METADATA_TYPE_STREAMINFO:
		b->length = FLAC__STREAM_METADATA_STREAMINFO_LENGTH;
		break;
	case FLAC__METADATA_TYPE_APPLICATION:
		b->length = 4;
		break;
	case FLAC__METADATA_TYPE_VORBIS_COMMENT: {
		Entry vendor;
		vendor.length = (FLAC__uint32)strlen(FLAC__VENDOR_STRING);
		vendor.entry = (FLAC__byte *)FLAC__VENDOR_STRING;
		if (!dup_entry(b->data.vorbis_comment.vendor_string, vendor)) {
			free(b);
			return 0;
		}
		relength_vorbiscomment(b);
		break;
	}
	case FLAC__METADATA_TYPE_CUESHEET:
		relength_cuesheet(b);
		break;
	case FLAC__METADATA_TYPE_PICTURE:
		b->length = (32 * 8) / 8;  // type, two string lengths, width, height, depth, colours, data length
		b->data.picture.mime_type = strdup("");
		b->data.picture.description = (FLAC__byte *)strdup("");
		if (!b->data.picture.mime_type || !b->data.picture.description) {
			free(b->data.picture.mime_type);
			free(b->data.picture.description);
			free(b);
			return 0;
		}
		break;
	default:
		break;  // padding, seek table, unknown: empty
	}
	return b;
}

void FLAC__metadata_object_delete_data(FLAC__StreamMetadata *b)
{
	switch (b->type) {
	case FLAC__METADATA_TYPE_STREAMINFO:
	case FLAC__METADATA_TYPE_PADDING:
		break;
	case FLAC__METADATA_TYPE_APPLICATION:
		free(b->data.application.data);
		b->data.application.data = 0;
		break;
	case FLAC__METADATA_TYPE_SEEKTABLE:
		free(b->data.seek_table.points);
		b->data.seek_table.points = 0;
		break;
	case FLAC__METADATA_TYPE_VORBIS_COMMENT:
		free(b->data.vorbis_comment.vendor_string.entry);
		b->data.vorbis_comment.vendor_string.entry = 0;
		if (b->data.vorbis_comment.comments)
			free_entries(b->data.vorbis_comment.comments, b->data.vorbis_comment.num_comments);
		break;
	case FLAC__METADATA_TYPE_CUESHEET:
		if (b->data.cue_sheet.tracks)
			free_tracks(b->data.cue_sheet.tracks, b->data.cue_sheet.num_tracks);
		break;
	case FLAC__METADATA_TYPE_PICTURE:
		free(b->data.picture.mime_type);
		free(b->data.picture.description);
		free(b->data.picture.data);
		b->data.picture.mime_type = 0;
		b->data.picture.description = 0;
		b->data.picture.data = 0;
		break;
	default:
		free(b->data.unknown.data);
		b->data.unknown.data = 0;
		break;
	}
}

void FLAC__metadata_object_delete(FLAC__StreamMetadata *b)
{
	FLAC__metadata_object_delete_data(b);
	free(b);
}

// metadata_object.c:524
FLAC__StreamMetadata *FLAC__metadata_object_clone(const FLAC__StreamMetadata *from)
{
	Block *to = FLAC__metadata_object_new(from->type);
	if (!to)
		return 0;
	to->is_last = from->is_last;
	to->length = from->length;
	bool ok = true;
	switch (from->type) {
	case FLAC__METADATA_TYPE_STREAMINFO:
		to->data.stream_info = from->data.stream_info;
		break;
	case FLAC__METADATA_TYPE_PADDING:
		break;
	case FLAC__METADATA_TYPE_APPLICATION:
		if (from->length < 4)
			ok = false;
		else {
			memcpy(to->data.application.id, from->data.application.id, 4);
			to->data.application.data = (FLAC__byte *)dup_bytes(from->data.application.data, from->length - 4, ok);
		}
		break;
	case FLAC__METADATA_TYPE_SEEKTABLE:
		to->data.seek_table.num_points = from->data.seek_table.num_points;
		to->data.seek_table.points =
			(SeekPoint *)dup_bytes(from->data.seek_table.points, (size_t)from->data.seek_table.num_points * sizeof(SeekPoint), ok);
		break;
	case FLAC__METADATA_TYPE_VORBIS_COMMENT: {
		const FLAC__StreamMetadata_VorbisComment &src = from->data.vorbis_comment;
		FLAC__StreamMetadata_VorbisComment &dst = to->data.vorbis_comment;
		free(dst.vendor_string.entry);
		dst.vendor_string.entry = 0;
		ok = dup_entry(dst.vendor_string, src.vendor_string);
		if (ok && src.num_comments) {
			dst.comments = (Entry *)calloc(src.num_comments, sizeof(Entry));
			ok = dst.comments != 0;
			if (ok)
				dst.num_comments = src.num_comments;
			for (unsigned i = 0; ok && i < src.num_comments; i++)
				ok = dup_entry(dst.comments[i], src.comments[i]);
		}
		break;
	}
	case FLAC__METADATA_TYPE_CUESHEET: {
		const FLAC__StreamMetadata_CueSheet &src = from->data.cue_sheet;
		FLAC__StreamMetadata_CueSheet &dst = to->data.cue_sheet;
		dst = src;
		dst.tracks = 0;
		dst.num_tracks = 0;
		if (src.num_tracks) {
			dst.tracks = (Track *)calloc(src.num_tracks, sizeof(Track));
			ok = dst.tracks != 0;
			if (ok)
				dst.num_tracks = src.num_tracks;
			for (unsigned i = 0; ok && i < src.num_tracks; i++)
				ok = dup_track(dst.tracks[i], src.tracks[i]);
		}
		break;
	}
	case FLAC__METADATA_TYPE_PICTURE: {
		const FLAC__StreamMetadata_Picture &src = from->data.picture;
		FLAC__StreamMetadata_Picture &dst = to->data.picture;
		char *mime = strdup(src.mime_type);
		char *desc = strdup((const char *)src.description);
		FLAC__byte *data = (FLAC__byte *)dup_bytes(src.data, src.data_length, ok);
		if (!mime || !desc || !ok) {
			free(mime);
			free(desc);
			free(data);
			ok = false;
			break;
		}
		free(dst.mime_type);
		free(dst.description);
		dst = src;
		dst.mime_type = mime;
		dst.description = (FLAC__byte *)desc;
		dst.data = data;
		break;
	}
	default:
		to->data.unknown.data = (FLAC__byte *)dup_bytes(from->data.unknown.data, from->length, ok);
		break;
	}
	if (!ok) {
		FLAC__metadata_object_delete(to);
		return 0;
	}
	return to;
}

// metadata_object.c:869
FLAC__bool FLAC__metadata_object_is_equal(const FLAC__StreamMetadata *a, const FLAC__StreamMetadata *b)
{
	if (a->type != b->type || a->is_last != b->is_last || a->length != b->length)
		return 0;
	switch (a->type) {
	case FLAC__METADATA_TYPE_STREAMINFO: {
		const FLAC__StreamMetadata_StreamInfo &x = a->data.stream_info, &y = b->data.stream_info;
		return x.min_blocksize == y.min_blocksize && x.max_blocksize == y.max_blocksize && x.min_framesize == y.min_framesize &&
		       x.max_framesize == y.max_framesize && x.sample_rate == y.sample_rate && x.channels == y.channels &&
		       x.bits_per_sample == y.bits_per_sample && x.total_samples == y.total_samples && memcmp(x.md5sum, y.md5sum, 16) == 0;
	}
	case FLAC__METADATA_TYPE_PADDING:
		return 1;  // the padding bytes themselves are not compared
	case FLAC__METADATA_TYPE_APPLICATION:
		return memcmp(a->data.application.id, b->data.application.id, 4) == 0 &&
		       same_bytes(a->data.application.data, b->data.application.data, a->length - 4);
	case FLAC__METADATA_TYPE_SEEKTABLE: {
		const FLAC__StreamMetadata_SeekTable &x = a->data.seek_table, &y = b->data.seek_table;
		if (x.num_points != y.num_points)
			return 0;
		if (!x.points || !y.points)
			return x.points == y.points;
		for (unsigned i = 0; i < x.num_points; i++)
			if (x.points[i].sample_number != y.points[i].sample_number || x.points[i].stream_offset != y.points[i].stream_offset ||
			    x.points[i].frame_samples != y.points[i].frame_samples)
				return 0;
		return 1;
	}
	case FLAC__METADATA_TYPE_VORBIS_COMMENT: {
		const FLAC__StreamMetadata_VorbisComment &x = a->data.vorbis_comment, &y = b->data.vorbis_comment;
		if (x.vendor_string.length != y.vendor_string.length || !same_bytes(x.vendor_string.entry, y.vendor_string.entry, x.vendor_string.length))
			return 0;
		if (x.num_comments != y.num_comments)
			return 0;
		for (unsigned i = 0; i < x.num_comments; i++)
			if (!same_bytes(x.comments[i].entry, y.comments[i].entry, x.comments[i].length))
				return 0;
		return 1;
	}
	case FLAC__METADATA_TYPE_CUESHEET: {
		const FLAC__StreamMetadata_CueSheet &x = a->data.cue_sheet, &y = b->data.cue_sheet;
		if (strcmp(x.media_catalog_number, y.media_catalog_number) != 0 || x.lead_in != y.lead_in || x.is_cd != y.is_cd ||
		    x.num_tracks != y.num_tracks)
			return 0;
		if (!x.tracks || !y.tracks)
			return x.tracks == y.tracks;
		for (unsigned i = 0; i < x.num_tracks; i++) {
			const Track &s = x.tracks[i], &t = y.tracks[i];
			if (s.offset != t.offset || s.number != t.number || memcmp(s.isrc, t.isrc, sizeof(s.isrc)) != 0 || s.type != t.type ||
			    s.pre_emphasis != t.pre_emphasis || s.num_indices != t.num_indices)
				return 0;
			if (!s.indices || !t.indices) {
				if (s.indices != t.indices)
					return 0;
				continue;
			}
			for (unsigned j = 0; j < s.num_indices; j++)
				if (s.indices[j].offset != t.indices[j].offset || s.indices[j].number != t.indices[j].number)
					return 0;
		}
		return 1;
	}
	case FLAC__METADATA_TYPE_PICTURE: {
		const FLAC__StreamMetadata_Picture &x = a->data.picture, &y = b->data.picture;
		return x.type == y.type && same_cstring(x.mime_type, y.mime_type) &&
		       same_cstring((const char *)x.description, (const char *)y.description) && x.width == y.width && x.height == y.height &&
		       x.depth == y.depth && x.colors == y.colors && x.data_length == y.data_length &&
		       (x.data == y.data || (x.data && y.data && memcmp(x.data, y.data, x.data_length) == 0));
	}
	default:
		return same_bytes(a->data.unknown.data, b->data.unknown.data, a->length);
	}
}

// metadata_object.c:903
FLAC__bool FLAC__metadata_object_application_set_data(FLAC__StreamMetadata *b, FLAC__byte *data, unsigned length, FLAC__bool copy)
{
	FLAC__byte *mine = data;
	if (copy) {
		bool ok;
		mine = (FLAC__byte *)dup_bytes(data, length, ok);
		if (!ok)
			return 0;
	}
	free(b->data.application.data);
	b->data.application.data = mine;
	b->length = 4 + length;
	return 1;
}

// ---- seek table (metadata_object.c:929-1151) ----
FLAC__bool FLAC__metadata_object_seektable_resize_points(FLAC__StreamMetadata *b, unsigned n)
{
	FLAC__StreamMetadata_SeekTable &st = b->data.seek_table;
	if (!resize_array(st.points, st.num_points, n, kPlaceholder))
		return 0;
	st.num_points = n;
	relength_seektable(b);
	return 1;
}

void FLAC__metadata_object_seektable_set_point(FLAC__StreamMetadata *b, unsigned at, FLAC__StreamMetadata_SeekPoint point)
{
	b->data.seek_table.points[at] = point;
}

FLAC__bool FLAC__metadata_object_seektable_insert_point(FLAC__StreamMetadata *b, unsigned at, FLAC__StreamMetadata_SeekPoint point)
{
	FLAC__StreamMetadata_SeekTable &st = b->data.seek_table;
	if (!FLAC__metadata_object_seektable_resize_points(b, st.num_points + 1))
		return 0;
	memmove(st.points + at + 1, st.points + at, (size_t)(st.num_points - 1 - at) * sizeof(SeekPoint));
	st.points[at] = point;
	return 1;
}

FLAC__bool FLAC__metadata_object_seektable_delete_point(FLAC__StreamMetadata *b, unsigned at)
{
	FLAC__StreamMetadata_SeekTable &st = b->data.seek_table;
	memmove(st.points + at, st.points + at + 1, (size_t)(st.num_points - 1 - at) * sizeof(SeekPoint));
	return FLAC__metadata_object_seektable_resize_points(b, st.num_points - 1);
}

FLAC__bool FLAC__metadata_object_seektable_is_legal(const FLAC__StreamMetadata *b) { return FLAC__format_seektable_is_legal(&b->data.seek_table); }

FLAC__bool FLAC__metadata_object_seektable_template_append_placeholders(FLAC__StreamMetadata *b, unsigned num)
{
	return num == 0 ? 1 : FLAC__metadata_object_seektable_resize_points(b, b->data.seek_table.num_points + num);
}

// the template points carry only a target sample; offset and frame size are filled in by the encoder
static FLAC__bool append_targets(FLAC__StreamMetadata *b, unsigned num, FLAC__uint64 (*target)(unsigned, const void *), const void *ctx)
{
	FLAC__StreamMetadata_SeekTable &st = b->data.seek_table;
	const unsigned first = st.num_points;
	if (num == 0)
		return 1;
	if (!FLAC__metadata_object_seektable_resize_points(b, first + num))
		return 0;
	for (unsigned j = 0; j < num; j++) {
		st.points[first + j].sample_number = target(j, ctx);
		st.points[first + j].stream_offset = 0;
		st.points[first + j].frame_samples = 0;
	}
	return 1;
}

FLAC__bool FLAC__metadata_object_seektable_template_append_point(FLAC__StreamMetadata *b, FLAC__uint64 sample_number)
{
	return append_targets(b, 1, [](unsigned, const void *c) { return *(const FLAC__uint64 *)c; }, &sample_number);
}

FLAC__bool FLAC__metadata_object_seektable_template_append_points(FLAC__StreamMetadata *b, FLAC__uint64 sample_numbers[], unsigned num)
{
	return append_targets(b, num, [](unsigned j, const void *c) { return ((const FLAC__uint64 *)c)[j]; }, sample_numbers);
}

FLAC__bool FLAC__metadata_object_seektable_template_append_spaced_points(FLAC__StreamMetadata *b, unsigned num, FLAC__uint64 total_samples)
{
	if (num == 0 || total_samples == 0)
		return 1;
	const FLAC__uint64 ctx[2] = {total_samples, num};
	return append_targets(b, num, [](unsigned j, const void *c) { return ((const FLAC__uint64 *)c)[0] * (FLAC__uint64)j / ((const FLAC__uint64 *)c)[1]; }, ctx);
}

FLAC__bool FLAC__metadata_object_seektable_template_append_spaced_points_by_samples(FLAC__StreamMetadata *b, unsigned samples, FLAC__uint64 total_samples)
{
	if (samples == 0 || total_samples == 0)
		return 1;
	// one point at 0 and then every `samples`, none at total_samples itself
	FLAC__uint64 num = 1 + total_samples / samples;
	if (total_samples % samples == 0)
		num--;
	const FLAC__uint64 step = samples;
	return append_targets(b, (unsigned)num, [](unsigned j, const void *c) { return *(const FLAC__uint64 *)c * j; }, &step);
}

FLAC__bool FLAC__metadata_object_seektable_template_sort(FLAC__StreamMetadata *b, FLAC__bool compact)
{
	const unsigned unique = FLAC__format_seektable_sort(&b->data.seek_table);
	return !compact || FLAC__metadata_object_seektable_resize_points(b, unique);
}

// ---- Vorbis comment (metadata_object.c:1153-1434) ----
FLAC__bool FLAC__metadata_object_vorbiscomment_set_vendor_string(FLAC__StreamMetadata *b, FLAC__StreamMetadata_VorbisComment_Entry entry, FLAC__bool copy)
{
	if (!FLAC__format_vorbiscomment_entry_value_is_legal(entry.entry, entry.length))
		return 0;
	return store_entry(b, &b->data.vorbis_comment.vendor_string, &entry, copy);
}

FLAC__bool FLAC__metadata_object_vorbiscomment_resize_comments(FLAC__StreamMetadata *b, unsigned n)
{
	FLAC__StreamMetadata_VorbisComment &vc = b->data.vorbis_comment;
	for (unsigned i = n; i < vc.num_comments; i++) {
		free(vc.comments[i].entry);
		vc.comments[i].entry = 0;
	}
	Entry none;
	memset(&none, 0, sizeof(none));
	if (!resize_array(vc.comments, vc.num_comments, n, none))
		return 0;
	vc.num_comments = n;
	relength_vorbiscomment(b);
	return 1;
}

FLAC__bool FLAC__metadata_object_vorbiscomment_set_comment(FLAC__StreamMetadata *b, unsigned at, FLAC__StreamMetadata_VorbisComment_Entry entry, FLAC__bool copy)
{
	if (!FLAC__format_vorbiscomment_entry_is_legal(entry.entry, entry.length))
		return 0;
	return store_entry(b, &b->data.vorbis_comment.comments[at], &entry, copy);
}

FLAC__bool FLAC__metadata_object_vorbiscomment_insert_comment(FLAC__StreamMetadata *b, unsigned at, FLAC__StreamMetadata_VorbisComment_Entry entry, FLAC__bool copy)
{
	FLAC__StreamMetadata_VorbisComment &vc = b->data.vorbis_comment;
	if (!FLAC__format_vorbiscomment_entry_is_legal(entry.entry, entry.length))
		return 0;
	if (!FLAC__metadata_object_vorbiscomment_resize_comments(b, vc.num_comments + 1))
		return 0;
	memmove(vc.comments + at + 1, vc.comments + at, (size_t)(vc.num_comments - 1 - at) * sizeof(Entry));
	vc.comments[at].length = 0;
	vc.comments[at].entry = 0;
	return FLAC__metadata_object_vorbiscomment_set_comment(b, at, entry, copy);
}

FLAC__bool FLAC__metadata_object_vorbiscomment_append_comment(FLAC__StreamMetadata *b, FLAC__StreamMetadata_VorbisComment_Entry entry, FLAC__bool copy)
{
	return FLAC__metadata_object_vorbiscomment_insert_comment(b, b->data.vorbis_comment.num_comments, entry, copy);
}

FLAC__bool FLAC__metadata_object_vorbiscomment_delete_comment(FLAC__StreamMetadata *b, unsigned at)
{
	FLAC__StreamMetadata_VorbisComment &vc = b->data.vorbis_comment;
	free(vc.comments[at].entry);
	memmove(vc.comments + at, vc.comments + at + 1, (size_t)(vc.num_comments - 1 - at) * sizeof(Entry));
	vc.comments[vc.num_comments - 1].length = 0;
	vc.comments[vc.num_comments - 1].entry = 0;
	return FLAC__metadata_object_vorbiscomment_resize_comments(b, vc.num_comments - 1);
}

// the first comment of the entry's field name is replaced (or the entry appended); with `all`, later
// comments of that name are deleted
FLAC__bool FLAC__metadata_object_vorbiscomment_replace_comment(FLAC__StreamMetadata *b, FLAC__StreamMetadata_VorbisComment_Entry entry, FLAC__bool all, FLAC__bool copy)
{
	if (!FLAC__format_vorbiscomment_entry_is_legal(entry.entry, entry.length))
		return 0;
	const FLAC__byte *eq = (const FLAC__byte *)memchr(entry.entry, '=', entry.length);
	if (!eq)
		return 0;
	const unsigned name_len = (unsigned)(eq - entry.entry);
	int i = find_comment(b, 0, (const char *)entry.entry, name_len);
	if (i < 0)
		return FLAC__metadata_object_vorbiscomment_append_comment(b, entry, copy);
	unsigned at = (unsigned)i;
	if (!FLAC__metadata_object_vorbiscomment_set_comment(b, at, entry, copy))
		return 0;
	// the caller's buffer may have moved when it was taken over: use the stored entry from here on
	entry = b->data.vorbis_comment.comments[at];
	at++;
	while (all && at < b->data.vorbis_comment.num_comments && (i = find_comment(b, at, (const char *)entry.entry, name_len)) >= 0) {
		at = (unsigned)i;
		if (!FLAC__metadata_object_vorbiscomment_delete_comment(b, at))
			return 0;
	}
	return 1;
}

FLAC__bool FLAC__metadata_object_vorbiscomment_entry_from_name_value_pair(FLAC__StreamMetadata_VorbisComment_Entry *entry, const char *name, const char *value)
{
	if (!FLAC__format_vorbiscomment_entry_name_is_legal(name) || !FLAC__format_vorbiscomment_entry_value_is_legal((const FLAC__byte *)value, (unsigned)(-1)))
		return 0;
	const size_t nn = strlen(name), nv = strlen(value);
	FLAC__byte *x = (FLAC__byte *)malloc(nn + 1 + nv + 1);
	entry->length = (FLAC__uint32)(nn + 1 + nv);
	entry->entry = x;
	if (!x)
		return 0;
	memcpy(x, name, nn);
	x[nn] = '=';
	memcpy(x + nn + 1, value, nv);
	x[nn + 1 + nv] = 0;
	return 1;
}

FLAC__bool FLAC__metadata_object_vorbiscomment_entry_to_name_value_pair(const FLAC__StreamMetadata_VorbisComment_Entry entry, char **name, char **value)
{
	if (!FLAC__format_vorbiscomment_entry_is_legal(entry.entry, entry.length))
		return 0;
	const FLAC__byte *eq = (const FLAC__byte *)memchr(entry.entry, '=', entry.length);
	if (!eq)
		return 0;
	const size_t nn = (size_t)(eq - entry.entry), nv = entry.length - nn - 1;
	char *n = (char *)malloc(nn + 1);
	if (!n)
		return 0;
	char *v = (char *)malloc(nv + 1);
	*name = n;
	if (!v) {
		free(n);
		return 0;
	}
	*value = v;
	memcpy(n, entry.entry, nn);
	n[nn] = 0;
	memcpy(v, eq + 1, nv);
	v[nv] = 0;
	return 1;
}

// field names compare without regard to ASCII case
FLAC__bool FLAC__metadata_object_vorbiscomment_entry_matches(const FLAC__StreamMetadata_VorbisComment_Entry entry, const char *name, unsigned name_len)
{
	const FLAC__byte *eq = (const FLAC__byte *)memchr(entry.entry, '=', entry.length);
	return eq && (unsigned)(eq - entry.entry) == name_len && strncasecmp(name, (const char *)entry.entry, name_len) == 0;
}

int FLAC__metadata_object_vorbiscomment_find_entry_from(const FLAC__StreamMetadata *b, unsigned offset, const char *name)
{
	return find_comment(b, offset, name, (unsigned)strlen(name));
}

int FLAC__metadata_object_vorbiscomment_remove_entry_matching(FLAC__StreamMetadata *b, const char *name)
{
	const int i = find_comment(b, 0, name, (unsigned)strlen(name));
	if (i < 0)
		return 0;
	return FLAC__metadata_object_vorbiscomment_delete_comment(b, (unsigned)i) ? 1 : -1;
}

int FLAC__metadata_object_vorbiscomment_remove_entries_matching(FLAC__StreamMetadata *b, const char *name)
{
	const unsigned name_len = (unsigned)strlen(name);
	int removed = 0;
	// from the back, so that deleting does not disturb what is still to be visited
	for (int i = (int)b->data.vorbis_comment.num_comments - 1; i >= 0; i--)
		if (FLAC__metadata_object_vorbiscomment_entry_matches(b->data.vorbis_comment.comments[i], name, name_len)) {
			removed++;
			if (!FLAC__metadata_object_vorbiscomment_delete_comment(b, (unsigned)i))
				return -1;
		}
	return removed;
}

// ---- cue sheet (metadata_object.c:1436-1724) ----
FLAC__StreamMetadata_CueSheet_Track *FLAC__metadata_object_cuesheet_track_new(void) { return (Track *)calloc(1, sizeof(Track)); }

FLAC__StreamMetadata_CueSheet_Track *FLAC__metadata_object_cuesheet_track_clone(const FLAC__StreamMetadata_CueSheet_Track *from)
{
	Track *t = FLAC__metadata_object_cuesheet_track_new();
	if (t && !dup_track(*t, *from)) {
		free(t);
		return 0;
	}
	return t;
}

void FLAC__metadata_object_cuesheet_track_delete_data(FLAC__StreamMetadata_CueSheet_Track *t) { free(t->indices); }

void FLAC__metadata_object_cuesheet_track_delete(FLAC__StreamMetadata_CueSheet_Track *t)
{
	free(t->indices);
	free(t);
}

FLAC__bool FLAC__metadata_object_cuesheet_track_resize_indices(FLAC__StreamMetadata *b, unsigned track_num, unsigned n)
{
	Track &t = b->data.cue_sheet.tracks[track_num];
	Index none;
	memset(&none, 0, sizeof(none));
	if (!resize_array(t.indices, t.num_indices, n, none))
		return 0;
	t.num_indices = (FLAC__byte)n;
	relength_cuesheet(b);
	return 1;
}

FLAC__bool FLAC__metadata_object_cuesheet_track_insert_index(FLAC__StreamMetadata *b, unsigned track_num, unsigned at, FLAC__StreamMetadata_CueSheet_Index index)
{
	Track &t = b->data.cue_sheet.tracks[track_num];
	if (!FLAC__metadata_object_cuesheet_track_resize_indices(b, track_num, t.num_indices + 1u))
		return 0;
	memmove(t.indices + at + 1, t.indices + at, (size_t)(t.num_indices - 1u - at) * sizeof(Index));
	t.indices[at] = index;
	return 1;
}

FLAC__bool FLAC__metadata_object_cuesheet_track_insert_blank_index(FLAC__StreamMetadata *b, unsigned track_num, unsigned at)
{
	Index blank;
	memset(&blank, 0, sizeof(blank));
	return FLAC__metadata_object_cuesheet_track_insert_index(b, track_num, at, blank);
}

FLAC__bool FLAC__metadata_object_cuesheet_track_delete_index(FLAC__StreamMetadata *b, unsigned track_num, unsigned at)
{
	Track &t = b->data.cue_sheet.tracks[track_num];
	memmove(t.indices + at, t.indices + at + 1, (size_t)(t.num_indices - 1u - at) * sizeof(Index));
	FLAC__metadata_object_cuesheet_track_resize_indices(b, track_num, t.num_indices - 1u);
	return 1;
}

FLAC__bool FLAC__metadata_object_cuesheet_resize_tracks(FLAC__StreamMetadata *b, unsigned n)
{
	FLAC__StreamMetadata_CueSheet &cs = b->data.cue_sheet;
	for (unsigned i = n; i < cs.num_tracks; i++) {
		free(cs.tracks[i].indices);
		cs.tracks[i].indices = 0;
	}
	Track none;
	memset(&none, 0, sizeof(none));
	if (!resize_array(cs.tracks, cs.num_tracks, n, none))
		return 0;
	cs.num_tracks = n;
	relength_cuesheet(b);
	return 1;
}

FLAC__bool FLAC__metadata_object_cuesheet_set_track(FLAC__StreamMetadata *b, unsigned at, FLAC__StreamMetadata_CueSheet_Track *track, FLAC__bool copy)
{
	Track &dest = b->data.cue_sheet.tracks[at];
	Index *old = dest.indices;
	if (copy) {
		if (!dup_track(dest, *track))
			return 0;
	} else
		dest = *track;
	free(old);
	relength_cuesheet(b);
	return 1;
}

FLAC__bool FLAC__metadata_object_cuesheet_insert_track(FLAC__StreamMetadata *b, unsigned at, FLAC__StreamMetadata_CueSheet_Track *track, FLAC__bool copy)
{
	FLAC__StreamMetadata_CueSheet &cs = b->data.cue_sheet;
	if (!FLAC__metadata_object_cuesheet_resize_tracks(b, cs.num_tracks + 1))
		return 0;
	memmove(cs.tracks + at + 1, cs.tracks + at, (size_t)(cs.num_tracks - 1 - at) * sizeof(Track));
	cs.tracks[at].num_indices = 0;
	cs.tracks[at].indices = 0;
	return FLAC__metadata_object_cuesheet_set_track(b, at, track, copy);
}

FLAC__bool FLAC__metadata_object_cuesheet_insert_blank_track(FLAC__StreamMetadata *b, unsigned at)
{
	Track blank;
	memset(&blank, 0, sizeof(blank));
	return FLAC__metadata_object_cuesheet_insert_track(b, at, &blank, 0);
}

FLAC__bool FLAC__metadata_object_cuesheet_delete_track(FLAC__StreamMetadata *b, unsigned at)
{
	FLAC__StreamMetadata_CueSheet &cs = b->data.cue_sheet;
	free(cs.tracks[at].indices);
	memmove(cs.tracks + at, cs.tracks + at + 1, (size_t)(cs.num_tracks - 1 - at) * sizeof(Track));
	cs.tracks[cs.num_tracks - 1].num_indices = 0;
	cs.tracks[cs.num_tracks - 1].indices = 0;
	return FLAC__metadata_object_cuesheet_resize_tracks(b, cs.num_tracks - 1);
}

FLAC__bool FLAC__metadata_object_cuesheet_is_legal(const FLAC__StreamMetadata *b, FLAC__bool check_cd_da_subset, const char **violation)
{
	return FLAC__format_cuesheet_is_legal(&b->data.cue_sheet, check_cd_da_subset, violation);
}

// where a track's index 01 lies on the disc, in samples; 0 for the lead-out or when there is none
static FLAC__uint64 index01_position(const FLAC__StreamMetadata_CueSheet &cs, unsigned track)
{
	if (track + 1 >= cs.num_tracks)
		return 0;
	const Track &t = cs.tracks[track];
	// only the first two index points are looked at
	for (unsigned j = 0; j < 2 && j < t.num_indices; j++)
		if (t.indices[j].number == 1)
			return t.indices[j].offset + t.offset + cs.lead_in;
	return 0;
}

// the freedb disc id (metadata_object.c:1704): digit sums of the track start seconds, the disc
// length in seconds, and the track count
FLAC__uint32 FLAC__metadata_object_cuesheet_calculate_cddb_id(const FLAC__StreamMetadata *b)
{
	const FLAC__StreamMetadata_CueSheet &cs = b->data.cue_sheet;
	if (cs.num_tracks < 2)
		return 0;
	FLAC__uint32 sum = 0;
	for (unsigned i = 0; i + 1 < cs.num_tracks; i++)
		for (FLAC__uint32 x = (FLAC__uint32)(index01_position(cs, i) / 44100); x; x /= 10)
			sum += x % 10;
	const FLAC__uint32 length =
		(FLAC__uint32)((cs.tracks[cs.num_tracks - 1].offset + cs.lead_in) / 44100) - (FLAC__uint32)(index01_position(cs, 0) / 44100);
	return (sum % 0xFF) << 24 | length << 8 | (FLAC__uint32)(cs.num_tracks - 1);
}

// ---- picture (metadata_object.c:1726-1824) ----
static FLAC__bool set_picture_string(FLAC__StreamMetadata *b, char **field, char *text, FLAC__bool copy)
{
	char *old = *field;
	const size_t old_len = old ? strlen(old) : 0, new_len = strlen(text);
	char *mine = text;
	if (copy) {
		mine = strdup(text);
		if (!mine)
			return 0;
	}
	*field = mine;
	free(old);
	b->length = b->length - (unsigned)old_len + (unsigned)new_len;
	return 1;
}

FLAC__bool FLAC__metadata_object_picture_set_mime_type(FLAC__StreamMetadata *b, char *mime_type, FLAC__bool copy)
{
	return set_picture_string(b, &b->data.picture.mime_type, mime_type, copy);
}

FLAC__bool FLAC__metadata_object_picture_set_description(FLAC__StreamMetadata *b, FLAC__byte *description, FLAC__bool copy)
{
	return set_picture_string(b, (char **)&b->data.picture.description, (char *)description, copy);
}

FLAC__bool FLAC__metadata_object_picture_set_data(FLAC__StreamMetadata *b, FLAC__byte *data, FLAC__uint32 length, FLAC__bool copy)
{
	FLAC__byte *mine = data;
	if (copy) {
		bool ok;
		mine = (FLAC__byte *)dup_bytes(data, length, ok);
		if (!ok)
			return 0;
	}
	free(b->data.picture.data);
	b->data.picture.data = mine;
	b->length = b->length - b->data.picture.data_length + length;
	b->data.picture.data_length = length;
	return 1;
}

FLAC__bool FLAC__metadata_object_picture_is_legal(const FLAC__StreamMetadata *b, const char **violation)
{
	return FLAC__format_picture_is_legal(&b->data.picture, violation);
}

// ---- level 0: one block out of a file (metadata_iterators.c:166-333) ----
// The stream decoder is set to hand over only the wanted block type and is run to the end of the
// metadata; no frame is decoded, so no GPU is involved.
namespace {
struct PickState {
	bool failed;
	FLAC__StreamMetadata *found;
	// picture selection (FLAC__metadata_get_picture)
	bool pictures;
	FLAC__StreamMetadata_Picture_Type type;
	const char *mime_type;
	const FLAC__byte *description;
	unsigned max_width, max_height, max_depth, max_colors;
	FLAC__uint64 best_area, best_depth;
};

FLAC__StreamDecoderWriteStatus pick_write(const FLAC__StreamDecoder *, const FLAC__Frame *, const FLAC__int32 *const[], void *)
{
	return FLAC__STREAM_DECODER_WRITE_STATUS_CONTINUE;
}

void pick_metadata(const FLAC__StreamDecoder *, const FLAC__StreamMetadata *block, void *client)
{
	PickState *st = (PickState *)client;
	if (st->failed)
		return;
	if (!st->pictures) {
		// the first block of the type is the one
		if (!st->found && !(st->found = FLAC__metadata_object_clone(block)))
			st->failed = true;
		return;
	}
	// the largest picture (by area, then colour depth) among those that meet the constraints
	const FLAC__StreamMetadata_Picture &pic = block->data.picture;
	const FLAC__uint64 area = (FLAC__uint64)pic.width * pic.height;
	const bool fits = (st->type == (FLAC__StreamMetadata_Picture_Type)(-1) || st->type == pic.type) &&
	                  (!st->mime_type || strcmp(st->mime_type, pic.mime_type) == 0) &&
	                  (!st->description || strcmp((const char *)st->description, (const char *)pic.description) == 0) &&
	                  pic.width <= st->max_width && pic.height <= st->max_height && pic.depth <= st->max_depth && pic.colors <= st->max_colors;
	if (fits && (area > st->best_area || (area == st->best_area && pic.depth > st->best_depth))) {
		FLAC__StreamMetadata *copy = FLAC__metadata_object_clone(block);
		if (!copy) {
			st->failed = true;
			return;
		}
		if (st->found)
			FLAC__metadata_object_delete(st->found);
		st->found = copy;
		st->best_area = area;
		st->best_depth = pic.depth;
	}
}

void pick_error(const FLAC__StreamDecoder *, FLAC__StreamDecoderErrorStatus status, void *client)
{
	if (status != FLAC__STREAM_DECODER_ERROR_STATUS_LOST_SYNC)
		((PickState *)client)->failed = true;
}

FLAC__StreamMetadata *pick_block(const char *filename, FLAC__MetadataType type, PickState &st)
{
	st.failed = false;
	st.found = 0;
	FLAC__StreamDecoder *dec = FLAC__stream_decoder_new();
	if (!dec)
		return 0;
	FLAC__stream_decoder_set_md5_checking(dec, 0);
	FLAC__stream_decoder_set_metadata_ignore_all(dec);
	FLAC__stream_decoder_set_metadata_respond(dec, type);
	bool ok = FLAC__stream_decoder_init_file(dec, filename, pick_write, pick_metadata, pick_error, &st) == FLAC__STREAM_DECODER_INIT_STATUS_OK && !st.failed;
	if (ok)
		ok = FLAC__stream_decoder_process_until_end_of_metadata(dec) && !st.failed;
	(void)FLAC__stream_decoder_finish(dec);
	FLAC__stream_decoder_delete(dec);
	if (!ok && st.found) {
		FLAC__metadata_object_delete(st.found);
		st.found = 0;
	}
	return st.found;
}
}  // namespace

FLAC__bool FLAC__metadata_get_streaminfo(const char *filename, FLAC__StreamMetadata *streaminfo)
{
	PickState st;
	memset(&st, 0, sizeof(st));
	FLAC__StreamMetadata *b = pick_block(filename, FLAC__METADATA_TYPE_STREAMINFO, st);
	if (!b)
		return 0;
	*streaminfo = *b;  // STREAMINFO owns no memory
	FLAC__metadata_object_delete(b);
	return 1;
}

FLAC__bool FLAC__metadata_get_tags(const char *filename, FLAC__StreamMetadata **tags)
{
	PickState st;
	memset(&st, 0, sizeof(st));
	*tags = pick_block(filename, FLAC__METADATA_TYPE_VORBIS_COMMENT, st);
	return *tags != 0;
}

FLAC__bool FLAC__metadata_get_cuesheet(const char *filename, FLAC__StreamMetadata **cuesheet)
{
	PickState st;
	memset(&st, 0, sizeof(st));
	*cuesheet = pick_block(filename, FLAC__METADATA_TYPE_CUESHEET, st);
	return *cuesheet != 0;
}

FLAC__bool FLAC__metadata_get_picture(const char *filename, FLAC__StreamMetadata **picture, FLAC__StreamMetadata_Picture_Type type, const char *mime_type,
                                      const FLAC__byte *description, unsigned max_width, unsigned max_height, unsigned max_depth, unsigned max_colors)
{
	PickState st;
	memset(&st, 0, sizeof(st));
	st.pictures = true;
	st.type = type;
	st.mime_type = mime_type;
	st.description = description;
	st.max_width = max_width;
	st.max_height = max_height;
	st.max_depth = max_depth;
	st.max_colors = max_colors;
	*picture = pick_block(filename, FLAC__METADATA_TYPE_PICTURE, st);
	return *picture != 0;
}

}  // extern "C"
