// Metadata level 1 (simple iterator) and level 2 (chain + iterator): reading and editing the
// metadata blocks of a FLAC file on disk or behind I/O callbacks.  Host code, no device work.
//
// Behaviour follows src/libFLAC/metadata_iterators.c of the reference: what is rewritten in place and
// what goes through a temporary file ("<name>.metadata_edit", renamed over the original), how padding
// absorbs growth and shrinkage, which calls are refused with which status.  Blocks are written by the
// stream layer's serialiser with the object's own vendor string (metadata_iterators.c:2640).
#include <errno.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include <utime.h>

#include <vector>

#include "flac_amd_stream.h"
#include "stream_common.h"

extern "C" {
// metadata_iterators.c:361
const char *const FLAC__Metadata_SimpleIteratorStatusString[] = {
	"FLAC__METADATA_SIMPLE_ITERATOR_STATUS_OK", "FLAC__METADATA_SIMPLE_ITERATOR_STATUS_ILLEGAL_INPUT",
	"FLAC__METADATA_SIMPLE_ITERATOR_STATUS_ERROR_OPENING_FILE", "FLAC__METADATA_SIMPLE_ITERATOR_STATUS_NOT_A_FLAC_FILE",
	"FLAC__METADATA_SIMPLE_ITERATOR_STATUS_NOT_WRITABLE", "FLAC__METADATA_SIMPLE_ITERATOR_STATUS_BAD_METADATA",
	"FLAC__METADATA_SIMPLE_ITERATOR_STATUS_READ_ERROR", "FLAC__METADATA_SIMPLE_ITERATOR_STATUS_SEEK_ERROR",
	"FLAC__METADATA_SIMPLE_ITERATOR_STATUS_WRITE_ERROR", "FLAC__METADATA_SIMPLE_ITERATOR_STATUS_RENAME_ERROR",
	"FLAC__METADATA_SIMPLE_ITERATOR_STATUS_UNLINK_ERROR", "FLAC__METADATA_SIMPLE_ITERATOR_STATUS_MEMORY_ALLOCATION_ERROR",
	"FLAC__METADATA_SIMPLE_ITERATOR_STATUS_INTERNAL_ERROR"};
// metadata_iterators.c:924
const char *const FLAC__Metadata_ChainStatusString[] = {
	"FLAC__METADATA_CHAIN_STATUS_OK", "FLAC__METADATA_CHAIN_STATUS_ILLEGAL_INPUT", "FLAC__METADATA_CHAIN_STATUS_ERROR_OPENING_FILE",
	"FLAC__METADATA_CHAIN_STATUS_NOT_A_FLAC_FILE", "FLAC__METADATA_CHAIN_STATUS_NOT_WRITABLE", "FLAC__METADATA_CHAIN_STATUS_BAD_METADATA",
	"FLAC__METADATA_CHAIN_STATUS_READ_ERROR", "FLAC__METADATA_CHAIN_STATUS_SEEK_ERROR", "FLAC__METADATA_CHAIN_STATUS_WRITE_ERROR",
	"FLAC__METADATA_CHAIN_STATUS_RENAME_ERROR", "FLAC__METADATA_CHAIN_STATUS_UNLINK_ERROR",
	"FLAC__METADATA_CHAIN_STATUS_MEMORY_ALLOCATION_ERROR", "FLAC__METADATA_CHAIN_STATUS_INTERNAL_ERROR",
	"FLAC__METADATA_CHAIN_STATUS_INVALID_CALLBACKS", "FLAC__METADATA_CHAIN_STATUS_READ_WRITE_MISMATCH",
	"FLAC__METADATA_CHAIN_STATUS_WRONG_WRITE_CALL"};
}

namespace {

typedef FLAC__Metadata_SimpleIteratorStatus SStatus;
const unsigned kHeader = 4;  // FLAC__STREAM_METADATA_HEADER_LENGTH

// ---- FILE * behind the callback signatures (metadata_iterators.c:3331-3345) ----
int file_seek(FLAC__IOHandle h, FLAC__int64 offset, int whence) { return fseeko((FILE *)h, (off_t)offset, whence); }
FLAC__int64 file_tell(FLAC__IOHandle h) { return ftello((FILE *)h); }
size_t file_read(void *p, size_t size, size_t n, FLAC__IOHandle h) { return fread(p, size, n, (FILE *)h); }
size_t file_write(const void *p, size_t size, size_t n, FLAC__IOHandle h) { return fwrite(p, size, n, (FILE *)h); }
int file_eof(FLAC__IOHandle h) { return feof((FILE *)h); }

uint64_t be(const FLAC__byte *b, unsigned n)
{
	uint64_t v = 0;
	for (unsigned i = 0; i < n; i++)
		v = (v << 8) | b[i];
	return v;
}
uint32_t le32(const FLAC__byte *b) { return (uint32_t)b[0] | (uint32_t)b[1] << 8 | (uint32_t)b[2] << 16 | (uint32_t)b[3] << 24; }

// seek_to_first_metadata_block_cb_ (metadata_iterators.c:2986): 0 ok, 1 read error, 2 seek error, 3 not FLAC
unsigned seek_to_first_block(FLAC__IOHandle h, FLAC__IOCallback_Read rd, FLAC__IOCallback_Seek sk)
{
	FLAC__byte buf[4];
	errno = 0;
	size_t n = rd(buf, 1, 4, h);
	if (errno) return 1;
	if (n != 4) return 3;
	if (!memcmp(buf, "ID3", 3)) {
		unsigned tag = 0;
		if (sk(h, 2, SEEK_CUR) < 0) return 2;
		for (int i = 0; i < 4; i++) {
			if (rd(buf, 1, 1, h) < 1 || (buf[0] & 0x80)) return 1;
			tag = (tag << 7) | (buf[0] & 0x7f);
		}
		if (sk(h, tag, SEEK_CUR) < 0) return 2;
		errno = 0;
		n = rd(buf, 1, 4, h);
		if (errno) return 1;
		if (n != 4) return 3;
	}
	return memcmp(buf, "fLaC", 4) ? 3 : 0;
}

bool read_header(FLAC__IOHandle h, FLAC__IOCallback_Read rd, FLAC__bool *is_last, FLAC__MetadataType *type, unsigned *length)
{
	FLAC__byte raw[kHeader];
	if (rd(raw, 1, kHeader, h) != kHeader)
		return false;
	*is_last = (raw[0] & 0x80) ? 1 : 0;
	*type = (FLAC__MetadataType)(raw[0] & 0x7f);
	*length = (unsigned)be(raw + 1, 3);
	return true;
}

// `n` bytes (plus `extra` zero bytes at the end) into a fresh buffer that grows as the bytes arrive: a length field of a
// damaged file cannot make the process reserve more memory than the file backs up.  Same results as reading into one
// malloc(n): READ_ERROR when the file ends early.
SStatus read_blob(FLAC__IOHandle h, FLAC__IOCallback_Read rd, size_t n, size_t extra, FLAC__byte **out)
{
	const size_t kStep = (size_t)1 << 20;
	size_t cap = (n < kStep ? n : kStep) + extra, have = 0;
	FLAC__byte *p = (FLAC__byte *)malloc(cap ? cap : 1);
	if (!p) return FLAC__METADATA_SIMPLE_ITERATOR_STATUS_MEMORY_ALLOCATION_ERROR;
	while (have < n) {
		const size_t want = n - have < kStep ? n - have : kStep;
		if (have + want + extra > cap) {
			cap = have + want + extra > 2 * cap ? have + want + extra : 2 * cap;
			if (cap > n + extra) cap = n + extra;
			FLAC__byte *q = (FLAC__byte *)realloc(p, cap);
			if (!q) {
				free(p);
				return FLAC__METADATA_SIMPLE_ITERATOR_STATUS_MEMORY_ALLOCATION_ERROR;
			}
			p = q;
		}
		if (rd(p + have, 1, want, h) != want) {
			free(p);
			return FLAC__METADATA_SIMPLE_ITERATOR_STATUS_READ_ERROR;
		}
		have += want;
	}
	memset(p + n, 0, extra);
	*out = p;
	return FLAC__METADATA_SIMPLE_ITERATOR_STATUS_OK;
}

SStatus read_entry(FLAC__IOHandle h, FLAC__IOCallback_Read rd, FLAC__StreamMetadata_VorbisComment_Entry *e)
{
	FLAC__byte b[4];
	if (rd(b, 1, 4, h) != 4)
		return FLAC__METADATA_SIMPLE_ITERATOR_STATUS_READ_ERROR;
	e->length = le32(b);
	free(e->entry);
	e->entry = 0;
	if (e->length)
		return read_blob(h, rd, e->length, 1, &e->entry);
	return FLAC__METADATA_SIMPLE_ITERATOR_STATUS_OK;
}

// a length-prefixed string of a PICTURE block, NUL-terminated in memory (metadata_iterators.c:2391)
SStatus read_cstring(FLAC__IOHandle h, FLAC__IOCallback_Read rd, FLAC__byte **data, FLAC__uint32 *length)
{
	FLAC__byte b[4];
	if (rd(b, 1, 4, h) != 4)
		return FLAC__METADATA_SIMPLE_ITERATOR_STATUS_READ_ERROR;
	*length = (FLAC__uint32)be(b, 4);
	free(*data);
	*data = 0;
	return read_blob(h, rd, *length, 1, data);
}

// read_metadata_block_data_cb_ (metadata_iterators.c:2105-2490): the body of a block whose type and
// length are already set, into an object fresh from FLAC__metadata_object_new
SStatus read_body(FLAC__IOHandle h, FLAC__IOCallback_Read rd, FLAC__IOCallback_Seek sk, FLAC__StreamMetadata *m)
{
	const SStatus OK = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_OK, RD = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_READ_ERROR,
	              MEM = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_MEMORY_ALLOCATION_ERROR;
	FLAC__byte b[40];
	switch (m->type) {
	case FLAC__METADATA_TYPE_STREAMINFO: {
		FLAC__StreamMetadata_StreamInfo &s = m->data.stream_info;
		if (rd(b, 1, 34, h) != 34) return RD;
		s.min_blocksize = (unsigned)be(b, 2);
		s.max_blocksize = (unsigned)be(b + 2, 2);
		s.min_framesize = (unsigned)be(b + 4, 3);
		s.max_framesize = (unsigned)be(b + 7, 3);
		s.sample_rate = (unsigned)(be(b + 10, 3) >> 4);
		s.channels = ((b[12] >> 1) & 7) + 1;
		s.bits_per_sample = (((b[12] & 1) << 4) | (b[13] >> 4)) + 1;
		s.total_samples = ((uint64_t)(b[13] & 0x0f) << 32) | be(b + 14, 4);
		memcpy(s.md5sum, b + 18, 16);
		return OK;
	}
	case FLAC__METADATA_TYPE_PADDING:
		return sk(h, m->length, SEEK_CUR) ? FLAC__METADATA_SIMPLE_ITERATOR_STATUS_SEEK_ERROR : OK;
	case FLAC__METADATA_TYPE_APPLICATION: {
		if (rd(m->data.application.id, 1, 4, h) != 4 || m->length < 4) return RD;
		const unsigned n = m->length - 4;
		m->data.application.data = 0;
		if (n) {
			if (!(m->data.application.data = (FLAC__byte *)malloc(n))) return MEM;
			if (rd(m->data.application.data, 1, n, h) != n) return RD;
		}
		return OK;
	}
	case FLAC__METADATA_TYPE_SEEKTABLE: {
		FLAC__StreamMetadata_SeekTable &t = m->data.seek_table;
		t.num_points = m->length / 18;
		t.points = 0;
		if (t.num_points && !(t.points = (FLAC__StreamMetadata_SeekPoint *)malloc((size_t)t.num_points * sizeof *t.points))) return MEM;
		for (unsigned i = 0; i < t.num_points; i++) {
			if (rd(b, 1, 18, h) != 18) return RD;
			t.points[i].sample_number = be(b, 8);
			t.points[i].stream_offset = be(b + 8, 8);
			t.points[i].frame_samples = (unsigned)be(b + 16, 2);
		}
		return OK;
	}
	case FLAC__METADATA_TYPE_VORBIS_COMMENT: {
		FLAC__StreamMetadata_VorbisComment &v = m->data.vorbis_comment;
		SStatus st = read_entry(h, rd, &v.vendor_string);
		if (st != OK) return st;
		if (rd(b, 1, 4, h) != 4) return RD;
		v.num_comments = le32(b);
		v.comments = 0;
		// the array grows with the entries that are really there (each takes four bytes of file at least), so that a
		// damaged count cannot reserve 64 GiB; num_comments is what the array holds if reading stops early
		const unsigned declared = v.num_comments;
		unsigned cap = 0;
		v.num_comments = 0;
		for (unsigned i = 0; i < declared; i++) {
			if (i == cap) {
				const unsigned ncap = cap ? (cap > declared / 2 ? declared : 2 * cap) : (declared < 1024 ? declared : 1024);
				void *q = realloc(v.comments, (size_t)ncap * sizeof *v.comments);
				if (!q) return MEM;
				v.comments = (FLAC__StreamMetadata_VorbisComment_Entry *)q;
				memset(v.comments + cap, 0, (size_t)(ncap - cap) * sizeof *v.comments);
				cap = ncap;
			}
			v.num_comments = i + 1;
			if ((st = read_entry(h, rd, v.comments + i)) != OK) return st;
		}
		return OK;
	}
	case FLAC__METADATA_TYPE_CUESHEET: {
		FLAC__StreamMetadata_CueSheet &c = m->data.cue_sheet;
		if (rd(c.media_catalog_number, 1, 128, h) != 128) return RD;
		if (rd(b, 1, 8, h) != 8) return RD;
		c.lead_in = be(b, 8);
		FLAC__byte res[259];
		if (rd(res, 1, 259, h) != 259) return RD;
		c.is_cd = (res[0] & 0x80) ? 1 : 0;
		if (rd(b, 1, 1, h) != 1) return RD;
		c.num_tracks = b[0];
		c.tracks = 0;
		if (c.num_tracks && !(c.tracks = (FLAC__StreamMetadata_CueSheet_Track *)calloc(c.num_tracks, sizeof *c.tracks))) {
			c.num_tracks = 0;
			return MEM;
		}
		for (unsigned i = 0; i < c.num_tracks; i++) {
			FLAC__StreamMetadata_CueSheet_Track &t = c.tracks[i];
			if (rd(b, 1, 8 + 1, h) != 9) return RD;
			t.offset = be(b, 8);
			t.number = b[8];
			if (rd(t.isrc, 1, 12, h) != 12) return RD;
			if (rd(b, 1, 14, h) != 14) return RD;
			t.type = b[0] >> 7;
			t.pre_emphasis = (b[0] >> 6) & 1;
			if (rd(b, 1, 1, h) != 1) return RD;
			t.num_indices = b[0];
			t.indices = 0;
			if (t.num_indices && !(t.indices = (FLAC__StreamMetadata_CueSheet_Index *)calloc(t.num_indices, sizeof *t.indices))) {
				t.num_indices = 0;
				return MEM;
			}
			for (unsigned j = 0; j < t.num_indices; j++) {
				if (rd(b, 1, 12, h) != 12) return RD;
				t.indices[j].offset = be(b, 8);
				t.indices[j].number = b[8];
			}
		}
		return OK;
	}
	case FLAC__METADATA_TYPE_PICTURE: {
		FLAC__StreamMetadata_Picture &p = m->data.picture;
		FLAC__uint32 n;
		if (rd(b, 1, 4, h) != 4) return RD;
		p.type = (FLAC__StreamMetadata_Picture_Type)be(b, 4);
		SStatus st = read_cstring(h, rd, (FLAC__byte **)&p.mime_type, &n);
		if (st != OK) return st;
		if ((st = read_cstring(h, rd, &p.description, &n)) != OK) return st;
		if (rd(b, 1, 20, h) != 20) return RD;
		p.width = (FLAC__uint32)be(b, 4);
		p.height = (FLAC__uint32)be(b + 4, 4);
		p.depth = (FLAC__uint32)be(b + 8, 4);
		p.colors = (FLAC__uint32)be(b + 12, 4);
		p.data_length = (FLAC__uint32)be(b + 16, 4);
		free(p.data);
		p.data = 0;
		if (p.data_length)
			return read_blob(h, rd, p.data_length, 0, &p.data);
		return OK;
	}
	default:
		m->data.unknown.data = 0;
		if (m->length) {
			if (!(m->data.unknown.data = (FLAC__byte *)malloc(m->length))) return MEM;
			if (rd(m->data.unknown.data, 1, m->length, h) != m->length) return RD;
		}
		return OK;
	}
}

// header + body of one block through a write callback
bool write_block(FLAC__IOHandle h, FLAC__IOCallback_Write wr, const FLAC__StreamMetadata *m)
{
	std::vector<uint8_t> bytes;
	fa_serialize_metadata(m, bytes, false);
	return wr(bytes.data(), 1, bytes.size(), h) == bytes.size();
}

bool copy_n(FLAC__IOHandle in, FLAC__IOCallback_Read rd, FLAC__IOHandle out, FLAC__IOCallback_Write wr, off_t bytes, SStatus *st)
{
	FLAC__byte buf[8192];
	while (bytes > 0) {
		const size_t n = bytes < (off_t)sizeof buf ? (size_t)bytes : sizeof buf;
		if (rd(buf, 1, n, in) != n) {
			*st = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_READ_ERROR;
			return false;
		}
		if (wr(buf, 1, n, out) != n) {
			*st = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_WRITE_ERROR;
			return false;
		}
		bytes -= (off_t)n;
	}
	return true;
}

bool copy_rest(FLAC__IOHandle in, FLAC__IOCallback_Read rd, FLAC__IOCallback_Eof eof, FLAC__IOHandle out, FLAC__IOCallback_Write wr, SStatus *st)
{
	FLAC__byte buf[8192];
	while (!eof(in)) {
		const size_t n = rd(buf, 1, sizeof buf, in);
		if (n == 0 && !eof(in)) {
			*st = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_READ_ERROR;
			return false;
		}
		if (n > 0 && wr(buf, 1, n, out) != n) {
			*st = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_WRITE_ERROR;
			return false;
		}
	}
	return true;
}

// open_tempfile_ / transport_tempfile_ / cleanup_tempfile_ (metadata_iterators.c:3224-3305)
struct TempFile {
	FILE *f = 0;
	char *name = 0;
	bool open(const char *filename, SStatus *st)
	{
		static const char *suffix = ".metadata_edit";
		name = (char *)malloc(strlen(filename) + strlen(suffix) + 1);
		if (!name) {
			*st = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_MEMORY_ALLOCATION_ERROR;
			return false;
		}
		strcpy(name, filename);
		strcat(name, suffix);
		if (!(f = fopen(name, "w+b"))) {
			*st = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_ERROR_OPENING_FILE;
			return false;
		}
		return true;
	}
	void cleanup()
	{
		if (f) fclose(f);
		f = 0;
		if (name) {
			unlink(name);
			free(name);
			name = 0;
		}
	}
	bool transport(const char *filename, SStatus *st)
	{
		fclose(f);
		f = 0;
		if (rename(name, filename)) {
			cleanup();
			*st = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_RENAME_ERROR;
			return false;
		}
		cleanup();
		return true;
	}
};

void set_file_stats(const char *filename, const struct stat *stats)
{
	struct utimbuf t;
	t.actime = stats->st_atime;
	t.modtime = stats->st_mtime;
	(void)chmod(filename, stats->st_mode);
	(void)utime(filename, &t);
	(void)!chown(filename, stats->st_uid, (gid_t)-1);
	(void)!chown(filename, (uid_t)-1, stats->st_gid);
}

FLAC__Metadata_ChainStatus chain_status_of(SStatus s)
{
	// the two enumerations agree on their first thirteen values (metadata_iterators.c:3347)
	return (FLAC__Metadata_ChainStatus)s;
}

}  // namespace

// =============================================================================================
// level 1
// =============================================================================================
#define SIMPLE_ITERATOR_MAX_PUSH_DEPTH 1

struct FLAC__Metadata_SimpleIterator {
	FILE *file;
	char *filename;
	struct stat stats;
	FLAC__bool has_stats;
	FLAC__bool is_writable;
	SStatus status;
	off_t offset[SIMPLE_ITERATOR_MAX_PUSH_DEPTH + 1];
	off_t first_offset;
	unsigned depth;
	FLAC__bool is_last;
	FLAC__MetadataType type;
	unsigned length;
};

namespace {

typedef FLAC__Metadata_SimpleIterator SI;

bool si_read_header(SI *it)
{
	if (!read_header(it->file, file_read, &it->is_last, &it->type, &it->length)) {
		it->status = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_READ_ERROR;
		return false;
	}
	return true;
}

void si_free_guts(SI *it)
{
	if (it->file) {
		fclose(it->file);
		it->file = 0;
		if (it->has_stats)
			set_file_stats(it->filename, &it->stats);
	}
	free(it->filename);
	it->filename = 0;
}

// simple_iterator_prime_input_ (metadata_iterators.c:435)
bool si_prime(SI *it, bool read_only)
{
	if (read_only || !(it->file = fopen(it->filename, "r+b"))) {
		it->is_writable = 0;
		if (read_only || errno == EACCES) {
			if (!(it->file = fopen(it->filename, "rb"))) {
				it->status = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_ERROR_OPENING_FILE;
				return false;
			}
		} else {
			it->status = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_ERROR_OPENING_FILE;
			return false;
		}
	} else
		it->is_writable = 1;
	switch (seek_to_first_block(it->file, file_read, file_seek)) {
	case 0:
		it->depth = 0;
		it->first_offset = it->offset[0] = ftello(it->file);
		return si_read_header(it);
	case 1: it->status = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_READ_ERROR; return false;
	case 2: it->status = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_SEEK_ERROR; return false;
	default: it->status = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_NOT_A_FLAC_FILE; return false;
	}
}

void si_push(SI *it)
{
	it->offset[it->depth + 1] = it->offset[it->depth];
	it->depth++;
}

bool si_pop(SI *it)
{
	it->depth--;
	if (fseeko(it->file, it->offset[it->depth], SEEK_SET)) {
		it->status = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_SEEK_ERROR;
		return false;
	}
	return si_read_header(it);
}

bool si_write_block(SI *it, FILE *f, const FLAC__StreamMetadata *m)
{
	if (!write_block(f, file_write, m)) {
		it->status = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_WRITE_ERROR;
		return false;
	}
	return true;
}

// write_metadata_block_stationary_ (metadata_iterators.c:2846)
bool si_write_stationary(SI *it, const FLAC__StreamMetadata *block)
{
	if (fseeko(it->file, it->offset[it->depth], SEEK_SET)) {
		it->status = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_SEEK_ERROR;
		return false;
	}
	if (!si_write_block(it, it->file, block))
		return false;
	if (fseeko(it->file, it->offset[it->depth], SEEK_SET)) {
		it->status = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_SEEK_ERROR;
		return false;
	}
	return si_read_header(it);
}

// write_metadata_block_stationary_with_padding_ (metadata_iterators.c:2867)
bool si_write_stationary_padded(SI *it, FLAC__StreamMetadata *block, unsigned padding_length, FLAC__bool padding_is_last)
{
	if (fseeko(it->file, it->offset[it->depth], SEEK_SET)) {
		it->status = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_SEEK_ERROR;
		return false;
	}
	block->is_last = 0;
	if (!si_write_block(it, it->file, block))
		return false;
	FLAC__StreamMetadata *padding = FLAC__metadata_object_new(FLAC__METADATA_TYPE_PADDING);
	if (!padding) {
		it->status = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_MEMORY_ALLOCATION_ERROR;
		return false;
	}
	padding->is_last = padding_is_last;
	padding->length = padding_length;
	const bool ok = si_write_block(it, it->file, padding);
	FLAC__metadata_object_delete(padding);
	if (!ok)
		return false;
	if (fseeko(it->file, it->offset[it->depth], SEEK_SET)) {
		it->status = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_SEEK_ERROR;
		return false;
	}
	return si_read_header(it);
}

// rewrite_whole_file_ with its prefix / postfix copies (metadata_iterators.c:2910, 3041, 3061)
bool si_rewrite_file(SI *it, FLAC__StreamMetadata *block, bool append)
{
	int fixup = 0;  // 1: clear, -1: set the is_last flag at fixup_offset of the new file
	off_t fixup_offset = -1;
	if (it->is_last) {
		if (append) {
			fixup = 1;
			fixup_offset = it->offset[it->depth];
		} else if (!block) {
			si_push(it);
			if (!FLAC__metadata_simple_iterator_prev(it)) {
				(void)si_pop(it);
				return false;
			}
			fixup = -1;
			fixup_offset = it->offset[it->depth];
			if (!si_pop(it))
				return false;
		}
	}
	// prefix
	const off_t offset_end = append ? it->offset[it->depth] + (off_t)kHeader + (off_t)it->length : it->offset[it->depth];
	if (fseeko(it->file, 0, SEEK_SET)) {
		it->status = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_SEEK_ERROR;
		return false;
	}
	TempFile tmp;
	if (!tmp.open(it->filename, &it->status) || !copy_n(it->file, file_read, tmp.f, file_write, offset_end, &it->status)) {
		tmp.cleanup();
		return false;
	}
	if (block && !si_write_block(it, tmp.f, block)) {
		tmp.cleanup();
		return false;
	}
	// postfix
	const off_t save_offset = it->offset[it->depth];
	const bool backup = block == 0;
	if (fseeko(it->file, save_offset + (off_t)kHeader + (off_t)it->length, SEEK_SET)) {
		tmp.cleanup();
		it->status = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_SEEK_ERROR;
		return false;
	}
	if (!copy_rest(it->file, file_read, file_eof, tmp.f, file_write, &it->status)) {
		tmp.cleanup();
		return false;
	}
	if (fixup) {
		FLAC__byte x;
		if (fseeko(tmp.f, fixup_offset, SEEK_SET) || fread(&x, 1, 1, tmp.f) != 1) {
			tmp.cleanup();
			it->status = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_READ_ERROR;
			return false;
		}
		x = fixup > 0 ? (x & 0x7f) : (x | 0x80);
		if (fseeko(tmp.f, fixup_offset, SEEK_SET) || fwrite(&x, 1, 1, tmp.f) != 1) {
			tmp.cleanup();
			it->status = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_WRITE_ERROR;
			return false;
		}
	}
	fclose(it->file);
	it->file = 0;
	if (!tmp.transport(it->filename, &it->status))
		return false;
	if (it->has_stats)
		set_file_stats(it->filename, &it->stats);
	if (!si_prime(it, !it->is_writable))
		return false;
	if (backup) {
		while (it->offset[it->depth] + (off_t)kHeader + (off_t)it->length < save_offset)
			if (!FLAC__metadata_simple_iterator_next(it))
				return false;
	} else {
		it->offset[0] = save_offset;
		it->depth++;
		if (!si_pop(it))
			return false;
	}
	if (append)
		return FLAC__metadata_simple_iterator_next(it);
	return true;
}

}  // namespace

extern "C" {

FLAC__Metadata_SimpleIterator *FLAC__metadata_simple_iterator_new(void)
{
	SI *it = (SI *)calloc(1, sizeof(SI));
	if (it) {
		it->status = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_OK;
		it->first_offset = it->offset[0] = -1;
	}
	return it;
}

void FLAC__metadata_simple_iterator_delete(FLAC__Metadata_SimpleIterator *it)
{
	si_free_guts(it);
	free(it);
}

FLAC__Metadata_SimpleIteratorStatus FLAC__metadata_simple_iterator_status(FLAC__Metadata_SimpleIterator *it)
{
	const SStatus s = it->status;
	it->status = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_OK;
	return s;
}

FLAC__bool FLAC__metadata_simple_iterator_init(FLAC__Metadata_SimpleIterator *it, const char *filename, FLAC__bool read_only, FLAC__bool preserve_file_stats)
{
	si_free_guts(it);
	it->has_stats = 0;
	if (!read_only && preserve_file_stats)
		it->has_stats = stat(filename, &it->stats) == 0;
	if (!(it->filename = strdup(filename))) {
		it->status = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_MEMORY_ALLOCATION_ERROR;
		return 0;
	}
	return si_prime(it, read_only != 0);
}

FLAC__bool FLAC__metadata_simple_iterator_is_writable(const FLAC__Metadata_SimpleIterator *it) { return it->is_writable; }

FLAC__bool FLAC__metadata_simple_iterator_next(FLAC__Metadata_SimpleIterator *it)
{
	if (it->is_last)
		return 0;
	if (fseeko(it->file, it->length, SEEK_CUR)) {
		it->status = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_SEEK_ERROR;
		return 0;
	}
	it->offset[it->depth] = ftello(it->file);
	return si_read_header(it);
}

FLAC__bool FLAC__metadata_simple_iterator_prev(FLAC__Metadata_SimpleIterator *it)
{
	if (it->offset[it->depth] == it->first_offset)
		return 0;
	if (fseeko(it->file, it->first_offset, SEEK_SET)) {
		it->status = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_SEEK_ERROR;
		return 0;
	}
	off_t this_offset = it->first_offset;
	if (!si_read_header(it))
		return 0;
	// walk forward until the block that ends where the current one starts
	while (ftello(it->file) + (off_t)it->length < it->offset[it->depth]) {
		if (fseeko(it->file, it->length, SEEK_CUR)) {
			it->status = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_SEEK_ERROR;
			return 0;
		}
		this_offset = ftello(it->file);
		if (!si_read_header(it))
			return 0;
	}
	it->offset[it->depth] = this_offset;
	return 1;
}

FLAC__bool FLAC__metadata_simple_iterator_is_last(const FLAC__Metadata_SimpleIterator *it) { return it->is_last; }
off_t FLAC__metadata_simple_iterator_get_block_offset(const FLAC__Metadata_SimpleIterator *it) { return it->offset[it->depth]; }
FLAC__MetadataType FLAC__metadata_simple_iterator_get_block_type(const FLAC__Metadata_SimpleIterator *it) { return it->type; }
unsigned FLAC__metadata_simple_iterator_get_block_length(const FLAC__Metadata_SimpleIterator *it) { return it->length; }

FLAC__bool FLAC__metadata_simple_iterator_get_application_id(FLAC__Metadata_SimpleIterator *it, FLAC__byte *id)
{
	if (it->type != FLAC__METADATA_TYPE_APPLICATION) {
		it->status = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_ILLEGAL_INPUT;
		return 0;
	}
	if (fread(id, 1, 4, it->file) != 4) {
		it->status = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_READ_ERROR;
		return 0;
	}
	if (fseeko(it->file, -4, SEEK_CUR)) {
		it->status = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_SEEK_ERROR;
		return 0;
	}
	return 1;
}

FLAC__StreamMetadata *FLAC__metadata_simple_iterator_get_block(FLAC__Metadata_SimpleIterator *it)
{
	FLAC__StreamMetadata *block = FLAC__metadata_object_new(it->type);
	if (!block) {
		it->status = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_MEMORY_ALLOCATION_ERROR;
		return 0;
	}
	block->is_last = it->is_last;
	block->length = it->length;
	const SStatus st = read_body(it->file, file_read, file_seek, block);
	if (st != FLAC__METADATA_SIMPLE_ITERATOR_STATUS_OK) {
		it->status = st;
		FLAC__metadata_object_delete(block);
		return 0;
	}
	if (fseeko(it->file, it->offset[it->depth] + (off_t)kHeader, SEEK_SET)) {
		it->status = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_SEEK_ERROR;
		FLAC__metadata_object_delete(block);
		return 0;
	}
	return block;
}

// metadata_iterators.c:673
FLAC__bool FLAC__metadata_simple_iterator_set_block(FLAC__Metadata_SimpleIterator *it, FLAC__StreamMetadata *block, FLAC__bool use_padding)
{
	if (!it->is_writable) {
		it->status = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_NOT_WRITABLE;
		return 0;
	}
	if ((it->type == FLAC__METADATA_TYPE_STREAMINFO || block->type == FLAC__METADATA_TYPE_STREAMINFO) && it->type != block->type) {
		it->status = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_ILLEGAL_INPUT;
		return 0;
	}
	block->is_last = it->is_last;
	if (it->length == block->length)
		return si_write_stationary(it, block);
	if (it->length > block->length) {
		if (use_padding && it->length >= kHeader + block->length)
			return si_write_stationary_padded(it, block, it->length - kHeader - block->length, block->is_last);
		return si_rewrite_file(it, block, false);
	}
	// the block grew: a padding block right behind it may give the room
	unsigned padding_leftover = 0;
	FLAC__bool padding_is_last = 0;
	if (use_padding) {
		if (it->is_last)
			use_padding = 0;
		else {
			const unsigned extra = block->length - it->length;
			si_push(it);
			if (!FLAC__metadata_simple_iterator_next(it)) {
				(void)si_pop(it);
				return 0;
			}
			if (it->type != FLAC__METADATA_TYPE_PADDING)
				use_padding = 0;
			else if (kHeader + it->length == extra) {
				padding_leftover = 0;
				block->is_last = it->is_last;
			} else if (it->length < extra)
				use_padding = 0;
			else {
				padding_leftover = kHeader + it->length - extra;
				padding_is_last = it->is_last;
				block->is_last = 0;
			}
			if (!si_pop(it))
				return 0;
		}
	}
	if (use_padding) {
		if (padding_leftover == 0)
			return si_write_stationary(it, block);
		return si_write_stationary_padded(it, block, padding_leftover - kHeader, padding_is_last);
	}
	return si_rewrite_file(it, block, false);
}

// metadata_iterators.c:771
FLAC__bool FLAC__metadata_simple_iterator_insert_block_after(FLAC__Metadata_SimpleIterator *it, FLAC__StreamMetadata *block, FLAC__bool use_padding)
{
	unsigned padding_leftover = 0;
	FLAC__bool padding_is_last = 0;
	if (!it->is_writable)
		return 0;
	if (block->type == FLAC__METADATA_TYPE_STREAMINFO) {
		it->status = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_ILLEGAL_INPUT;
		return 0;
	}
	block->is_last = it->is_last;
	if (use_padding) {
		if (it->is_last)
			use_padding = 0;
		else {
			si_push(it);
			if (!FLAC__metadata_simple_iterator_next(it)) {
				(void)si_pop(it);
				return 0;
			}
			if (it->type != FLAC__METADATA_TYPE_PADDING)
				use_padding = 0;
			else if (it->length == block->length) {
				padding_leftover = 0;
				block->is_last = it->is_last;
			} else if (it->length < kHeader + block->length)
				use_padding = 0;
			else {
				padding_leftover = it->length - block->length;
				padding_is_last = it->is_last;
				block->is_last = 0;
			}
			if (!si_pop(it))
				return 0;
		}
	}
	if (use_padding) {
		// move to the padding block and overwrite it
		if (!FLAC__metadata_simple_iterator_next(it))
			return 0;
		if (padding_leftover == 0)
			return si_write_stationary(it, block);
		return si_write_stationary_padded(it, block, padding_leftover - kHeader, padding_is_last);
	}
	return si_rewrite_file(it, block, true);
}

// metadata_iterators.c:850
FLAC__bool FLAC__metadata_simple_iterator_delete_block(FLAC__Metadata_SimpleIterator *it, FLAC__bool use_padding)
{
	if (it->type == FLAC__METADATA_TYPE_STREAMINFO) {
		it->status = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_ILLEGAL_INPUT;
		return 0;
	}
	if (use_padding) {
		FLAC__StreamMetadata *padding = FLAC__metadata_object_new(FLAC__METADATA_TYPE_PADDING);
		if (!padding) {
			it->status = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_MEMORY_ALLOCATION_ERROR;
			return 0;
		}
		padding->length = it->length;
		const FLAC__bool ok = FLAC__metadata_simple_iterator_set_block(it, padding, 0);
		FLAC__metadata_object_delete(padding);
		if (!ok)
			return 0;
		return FLAC__metadata_simple_iterator_prev(it);
	}
	return si_rewrite_file(it, 0, false);
}

}  // extern "C"

// =============================================================================================
// level 2
// =============================================================================================
struct FaNode {
	FLAC__StreamMetadata *data;
	FaNode *prev, *next;
};

struct FLAC__Metadata_Chain {
	char *filename;  // NULL when the chain was read through callbacks
	FLAC__bool is_ogg;
	FaNode *head, *tail;
	unsigned nodes;
	FLAC__Metadata_ChainStatus status;
	off_t first_offset, last_offset;
	off_t initial_length;  // of the metadata as read: decides between rewriting in place and rewriting the file
};

struct FLAC__Metadata_Iterator {
	FLAC__Metadata_Chain *chain;
	FaNode *current;
};

namespace {

typedef FLAC__Metadata_Chain Chain;

void node_delete(FaNode *n)
{
	if (n->data)
		FLAC__metadata_object_delete(n->data);
	free(n);
}

void chain_init(Chain *c)
{
	c->filename = 0;
	c->is_ogg = 0;
	c->head = c->tail = 0;
	c->nodes = 0;
	c->status = FLAC__METADATA_CHAIN_STATUS_OK;
	c->initial_length = 0;
}

void chain_clear(Chain *c)
{
	for (FaNode *n = c->head; n;) {
		FaNode *next = n->next;
		node_delete(n);
		n = next;
	}
	free(c->filename);
	chain_init(c);
}

void chain_append(Chain *c, FaNode *n)
{
	n->next = n->prev = 0;
	n->data->is_last = 1;
	if (c->tail)
		c->tail->data->is_last = 0;
	if (!c->head)
		c->head = n;
	else {
		c->tail->next = n;
		n->prev = c->tail;
	}
	c->tail = n;
	c->nodes++;
}

void chain_remove(Chain *c, FaNode *n)
{
	if (n == c->head) c->head = n->next;
	else n->prev->next = n->next;
	if (n == c->tail) c->tail = n->prev;
	else n->next->prev = n->prev;
	if (c->tail)
		c->tail->data->is_last = 1;
	c->nodes--;
}

void chain_delete_node(Chain *c, FaNode *n)
{
	chain_remove(c, n);
	node_delete(n);
}

off_t chain_length(const Chain *c)
{
	off_t len = 0;
	for (const FaNode *n = c->head; n; n = n->next)
		len += kHeader + n->data->length;
	return len;
}

// chain_prepare_for_write_ (metadata_iterators.c:1119): let trailing padding absorb the change in size
off_t chain_prepare(Chain *c, bool use_padding)
{
	off_t cur = chain_length(c);
	if (!use_padding)
		return cur;
	if (cur < c->initial_length && c->tail->data->type == FLAC__METADATA_TYPE_PADDING) {
		const off_t delta = c->initial_length - cur;
		c->tail->data->length += (unsigned)delta;
		cur += delta;
	} else if (cur + (off_t)kHeader <= c->initial_length) {
		FLAC__StreamMetadata *padding = FLAC__metadata_object_new(FLAC__METADATA_TYPE_PADDING);
		FaNode *n = padding ? (FaNode *)calloc(1, sizeof(FaNode)) : 0;
		if (!n) {
			if (padding) FLAC__metadata_object_delete(padding);
			c->status = FLAC__METADATA_CHAIN_STATUS_MEMORY_ALLOCATION_ERROR;
			return 0;
		}
		padding->length = (unsigned)(c->initial_length - ((off_t)kHeader + cur));
		n->data = padding;
		chain_append(c, n);
		cur = chain_length(c);
	} else if (cur > c->initial_length) {
		const off_t delta = cur - c->initial_length;
		if (c->tail->data->type == FLAC__METADATA_TYPE_PADDING) {
			if ((off_t)c->tail->data->length + (off_t)kHeader == delta) {
				chain_delete_node(c, c->tail);
				cur = chain_length(c);
			} else if ((off_t)c->tail->data->length >= delta) {
				c->tail->data->length -= (unsigned)delta;
				cur -= delta;
			}
		}
	}
	return cur;
}

// chain_read_cb_ (metadata_iterators.c:1173)
bool chain_read_cb(Chain *c, FLAC__IOHandle h, FLAC__IOCallback_Read rd, FLAC__IOCallback_Seek sk, FLAC__IOCallback_Tell tell)
{
	switch (seek_to_first_block(h, rd, sk)) {
	case 0: break;
	case 1: c->status = FLAC__METADATA_CHAIN_STATUS_READ_ERROR; return false;
	case 2: c->status = FLAC__METADATA_CHAIN_STATUS_SEEK_ERROR; return false;
	default: c->status = FLAC__METADATA_CHAIN_STATUS_NOT_A_FLAC_FILE; return false;
	}
	FLAC__int64 pos = tell(h);
	if (pos < 0) {
		c->status = FLAC__METADATA_CHAIN_STATUS_READ_ERROR;
		return false;
	}
	c->first_offset = (off_t)pos;
	FLAC__bool is_last;
	do {
		FLAC__MetadataType type;
		unsigned length;
		FaNode *n = (FaNode *)calloc(1, sizeof(FaNode));
		if (!n) {
			c->status = FLAC__METADATA_CHAIN_STATUS_MEMORY_ALLOCATION_ERROR;
			return false;
		}
		if (!read_header(h, rd, &is_last, &type, &length)) {
			node_delete(n);
			c->status = FLAC__METADATA_CHAIN_STATUS_READ_ERROR;
			return false;
		}
		if (!(n->data = FLAC__metadata_object_new(type))) {
			node_delete(n);
			c->status = FLAC__METADATA_CHAIN_STATUS_MEMORY_ALLOCATION_ERROR;
			return false;
		}
		n->data->is_last = is_last;
		n->data->length = length;
		c->status = chain_status_of(read_body(h, rd, sk, n->data));
		if (c->status != FLAC__METADATA_CHAIN_STATUS_OK) {
			node_delete(n);
			return false;
		}
		chain_append(c, n);
	} while (!is_last);
	pos = tell(h);
	if (pos < 0) {
		c->status = FLAC__METADATA_CHAIN_STATUS_READ_ERROR;
		return false;
	}
	c->last_offset = (off_t)pos;
	c->initial_length = chain_length(c);
	return true;
}

// chain_rewrite_metadata_in_place_cb_ (metadata_iterators.c:1348)
bool chain_write_in_place(Chain *c, FLAC__IOHandle h, FLAC__IOCallback_Write wr, FLAC__IOCallback_Seek sk)
{
	if (sk(h, c->first_offset, SEEK_SET)) {
		c->status = FLAC__METADATA_CHAIN_STATUS_SEEK_ERROR;
		return false;
	}
	for (FaNode *n = c->head; n; n = n->next)
		if (!write_block(h, wr, n->data)) {
			c->status = FLAC__METADATA_CHAIN_STATUS_WRITE_ERROR;
			return false;
		}
	c->status = FLAC__METADATA_CHAIN_STATUS_OK;
	return true;
}

// chain_rewrite_file_cb_ (metadata_iterators.c:1466): prefix, new metadata, everything after the old metadata
bool chain_rewrite_cb(Chain *c, FLAC__IOHandle h, FLAC__IOCallback_Read rd, FLAC__IOCallback_Seek sk, FLAC__IOCallback_Eof eof, FLAC__IOHandle th,
                      FLAC__IOCallback_Write twr)
{
	SStatus st = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_OK;
	if (!copy_n(h, rd, th, twr, c->first_offset, &st)) {
		c->status = chain_status_of(st);
		return false;
	}
	for (FaNode *n = c->head; n; n = n->next)
		if (!write_block(th, twr, n->data)) {
			c->status = FLAC__METADATA_CHAIN_STATUS_WRITE_ERROR;
			return false;
		}
	if (sk(h, c->last_offset, SEEK_SET)) {
		c->status = FLAC__METADATA_CHAIN_STATUS_SEEK_ERROR;
		return false;
	}
	if (!copy_rest(h, rd, eof, th, twr, &st)) {
		c->status = chain_status_of(st);
		return false;
	}
	return true;
}

void chain_recompute_offsets(Chain *c, off_t current_length)
{
	c->initial_length = current_length;
	c->last_offset = c->first_offset;
	for (const FaNode *n = c->head; n; n = n->next)
		c->last_offset += kHeader + n->data->length;
}

bool chain_merge_adjacent_padding(Chain *c, FaNode *n)
{
	if (n->data->type == FLAC__METADATA_TYPE_PADDING && n->next && n->next->data->type == FLAC__METADATA_TYPE_PADDING) {
		n->data->length += kHeader + n->next->data->length;
		chain_delete_node(c, n->next);
		return true;
	}
	return false;
}

}  // namespace

extern "C" {

FLAC__Metadata_Chain *FLAC__metadata_chain_new(void)
{
	Chain *c = (Chain *)calloc(1, sizeof(Chain));
	if (c)
		chain_init(c);
	return c;
}

void FLAC__metadata_chain_delete(FLAC__Metadata_Chain *c)
{
	chain_clear(c);
	free(c);
}

FLAC__Metadata_ChainStatus FLAC__metadata_chain_status(FLAC__Metadata_Chain *c)
{
	const FLAC__Metadata_ChainStatus s = c->status;
	c->status = FLAC__METADATA_CHAIN_STATUS_OK;
	return s;
}

FLAC__bool FLAC__metadata_chain_read(FLAC__Metadata_Chain *c, const char *filename)
{
	chain_clear(c);
	if (!(c->filename = strdup(filename))) {
		c->status = FLAC__METADATA_CHAIN_STATUS_MEMORY_ALLOCATION_ERROR;
		return 0;
	}
	FILE *f = fopen(filename, "rb");
	if (!f) {
		c->status = FLAC__METADATA_CHAIN_STATUS_ERROR_OPENING_FILE;
		return 0;
	}
	const bool ok = chain_read_cb(c, f, file_read, file_seek, file_tell);
	fclose(f);
	return ok;
}

// Ogg FLAC is not part of this library (a reference built without libogg has no Ogg decoder either)
FLAC__bool FLAC__metadata_chain_read_ogg(FLAC__Metadata_Chain *c, const char *filename)
{
	(void)filename;
	chain_clear(c);
	c->status = FLAC__METADATA_CHAIN_STATUS_INTERNAL_ERROR;
	return 0;
}

FLAC__bool FLAC__metadata_chain_read_with_callbacks(FLAC__Metadata_Chain *c, FLAC__IOHandle handle, FLAC__IOCallbacks cb)
{
	chain_clear(c);
	if (!cb.read || !cb.seek || !cb.tell) {
		c->status = FLAC__METADATA_CHAIN_STATUS_INVALID_CALLBACKS;
		return 0;
	}
	if (cb.seek(handle, 0, SEEK_SET)) {
		c->status = FLAC__METADATA_CHAIN_STATUS_SEEK_ERROR;
		return 0;
	}
	return chain_read_cb(c, handle, cb.read, cb.seek, cb.tell);
}

FLAC__bool FLAC__metadata_chain_read_ogg_with_callbacks(FLAC__Metadata_Chain *c, FLAC__IOHandle handle, FLAC__IOCallbacks cb)
{
	(void)handle;
	(void)cb;
	chain_clear(c);
	c->status = FLAC__METADATA_CHAIN_STATUS_INTERNAL_ERROR;
	return 0;
}

// metadata_iterators.c:1622: the checks of chain_prepare without touching the chain
FLAC__bool FLAC__metadata_chain_check_if_tempfile_needed(FLAC__Metadata_Chain *c, FLAC__bool use_padding)
{
	const off_t cur = chain_length(c);
	if (use_padding) {
		if (cur < c->initial_length && c->tail->data->type == FLAC__METADATA_TYPE_PADDING)
			return 0;
		if (cur + (off_t)kHeader <= c->initial_length)
			return 0;
		if (cur > c->initial_length && c->tail->data->type == FLAC__METADATA_TYPE_PADDING) {
			const off_t delta = cur - c->initial_length;
			if ((off_t)c->tail->data->length + (off_t)kHeader == delta)
				return 0;
			if ((off_t)c->tail->data->length >= delta)
				return 0;
		}
	}
	return cur != c->initial_length;
}

FLAC__bool FLAC__metadata_chain_write(FLAC__Metadata_Chain *c, FLAC__bool use_padding, FLAC__bool preserve_file_stats)
{
	struct stat stats;
	bool have_stats = false;
	if (c->is_ogg) {
		c->status = FLAC__METADATA_CHAIN_STATUS_INTERNAL_ERROR;
		return 0;
	}
	if (!c->filename) {
		c->status = FLAC__METADATA_CHAIN_STATUS_READ_WRITE_MISMATCH;
		return 0;
	}
	const off_t cur = chain_prepare(c, use_padding != 0);
	if (cur == 0)
		return 0;
	if (preserve_file_stats)
		have_stats = stat(c->filename, &stats) == 0;
	if (cur == c->initial_length) {
		FILE *f = fopen(c->filename, "r+b");
		if (!f) {
			c->status = FLAC__METADATA_CHAIN_STATUS_ERROR_OPENING_FILE;
			return 0;
		}
		const bool ok = chain_write_in_place(c, f, file_write, file_seek);
		fclose(f);
		if (!ok)
			return 0;
	} else {
		// chain_rewrite_file_ (metadata_iterators.c:1397)
		FILE *f = fopen(c->filename, "rb");
		if (!f) {
			c->status = FLAC__METADATA_CHAIN_STATUS_ERROR_OPENING_FILE;
			return 0;
		}
		TempFile tmp;
		SStatus st = FLAC__METADATA_SIMPLE_ITERATOR_STATUS_OK;
		if (!tmp.open(c->filename, &st)) {
			c->status = chain_status_of(st);
			tmp.cleanup();
			fclose(f);
			return 0;
		}
		if (!chain_rewrite_cb(c, f, file_read, file_seek, file_eof, tmp.f, file_write)) {
			tmp.cleanup();
			fclose(f);
			return 0;
		}
		fclose(f);
		if (!tmp.transport(c->filename, &st)) {
			c->status = chain_status_of(st);
			return 0;
		}
		chain_recompute_offsets(c, cur);
	}
	if (preserve_file_stats && have_stats)
		set_file_stats(c->filename, &stats);
	return 1;
}

FLAC__bool FLAC__metadata_chain_write_with_callbacks(FLAC__Metadata_Chain *c, FLAC__bool use_padding, FLAC__IOHandle handle, FLAC__IOCallbacks cb)
{
	if (c->is_ogg) {
		c->status = FLAC__METADATA_CHAIN_STATUS_INTERNAL_ERROR;
		return 0;
	}
	if (c->filename) {
		c->status = FLAC__METADATA_CHAIN_STATUS_READ_WRITE_MISMATCH;
		return 0;
	}
	if (!cb.write || !cb.seek) {
		c->status = FLAC__METADATA_CHAIN_STATUS_INVALID_CALLBACKS;
		return 0;
	}
	if (FLAC__metadata_chain_check_if_tempfile_needed(c, use_padding)) {
		c->status = FLAC__METADATA_CHAIN_STATUS_WRONG_WRITE_CALL;
		return 0;
	}
	if (chain_prepare(c, use_padding != 0) == 0)
		return 0;
	return chain_write_in_place(c, handle, cb.write, cb.seek);
}

FLAC__bool FLAC__metadata_chain_write_with_callbacks_and_tempfile(FLAC__Metadata_Chain *c, FLAC__bool use_padding, FLAC__IOHandle handle,
                                                                  FLAC__IOCallbacks cb, FLAC__IOHandle temp_handle, FLAC__IOCallbacks temp_cb)
{
	if (c->is_ogg) {
		c->status = FLAC__METADATA_CHAIN_STATUS_INTERNAL_ERROR;
		return 0;
	}
	if (c->filename) {
		c->status = FLAC__METADATA_CHAIN_STATUS_READ_WRITE_MISMATCH;
		return 0;
	}
	if (!cb.read || !cb.seek || !cb.eof || !temp_cb.write) {
		c->status = FLAC__METADATA_CHAIN_STATUS_INVALID_CALLBACKS;
		return 0;
	}
	if (!FLAC__metadata_chain_check_if_tempfile_needed(c, use_padding)) {
		c->status = FLAC__METADATA_CHAIN_STATUS_WRONG_WRITE_CALL;
		return 0;
	}
	const off_t cur = chain_prepare(c, use_padding != 0);
	if (cur == 0)
		return 0;
	if (cb.seek(handle, 0, SEEK_SET)) {
		c->status = FLAC__METADATA_CHAIN_STATUS_SEEK_ERROR;
		return 0;
	}
	if (!chain_rewrite_cb(c, handle, cb.read, cb.seek, cb.eof, temp_handle, temp_cb.write))
		return 0;
	chain_recompute_offsets(c, cur);
	return 1;
}

void FLAC__metadata_chain_merge_padding(FLAC__Metadata_Chain *c)
{
	for (FaNode *n = c->head; n;)
		if (!chain_merge_adjacent_padding(c, n))
			n = n->next;
}

void FLAC__metadata_chain_sort_padding(FLAC__Metadata_Chain *c)
{
	// every padding block moves to the end, in order, then they are merged (metadata_iterators.c:1815)
	FaNode *n = c->head;
	for (unsigned i = 0, count = c->nodes; i < count; i++) {
		if (n->data->type == FLAC__METADATA_TYPE_PADDING) {
			FaNode *save = n->next;
			chain_remove(c, n);
			chain_append(c, n);
			n = save;
		} else
			n = n->next;
	}
	FLAC__metadata_chain_merge_padding(c);
}

FLAC__Metadata_Iterator *FLAC__metadata_iterator_new(void) { return (FLAC__Metadata_Iterator *)calloc(1, sizeof(FLAC__Metadata_Iterator)); }
void FLAC__metadata_iterator_delete(FLAC__Metadata_Iterator *it) { free(it); }

void FLAC__metadata_iterator_init(FLAC__Metadata_Iterator *it, FLAC__Metadata_Chain *c)
{
	it->chain = c;
	it->current = c->head;
}

FLAC__bool FLAC__metadata_iterator_next(FLAC__Metadata_Iterator *it)
{
	if (!it->current || !it->current->next)
		return 0;
	it->current = it->current->next;
	return 1;
}

FLAC__bool FLAC__metadata_iterator_prev(FLAC__Metadata_Iterator *it)
{
	if (!it->current || !it->current->prev)
		return 0;
	it->current = it->current->prev;
	return 1;
}

FLAC__MetadataType FLAC__metadata_iterator_get_block_type(const FLAC__Metadata_Iterator *it) { return it->current->data->type; }
FLAC__StreamMetadata *FLAC__metadata_iterator_get_block(FLAC__Metadata_Iterator *it) { return it->current->data; }

FLAC__bool FLAC__metadata_iterator_delete_block(FLAC__Metadata_Iterator *it, FLAC__bool replace_with_padding)
{
	if (!it->current->prev)
		return 0;  // the STREAMINFO block stays
	FaNode *save = it->current->prev;
	if (replace_with_padding) {
		FLAC__metadata_object_delete_data(it->current->data);
		it->current->data->type = FLAC__METADATA_TYPE_PADDING;
	} else
		chain_delete_node(it->chain, it->current);
	it->current = save;
	return 1;
}

FLAC__bool FLAC__metadata_iterator_insert_block_before(FLAC__Metadata_Iterator *it, FLAC__StreamMetadata *block)
{
	if (block->type == FLAC__METADATA_TYPE_STREAMINFO || !it->current->prev)
		return 0;
	FaNode *n = (FaNode *)calloc(1, sizeof(FaNode));
	if (!n)
		return 0;
	n->data = block;
	// iterator_insert_node_ (metadata_iterators.c:1046)
	block->is_last = 0;
	n->prev = it->current->prev;
	n->next = it->current;
	if (!n->prev) it->chain->head = n;
	else n->prev->next = n;
	it->current->prev = n;
	it->chain->nodes++;
	it->current = n;
	return 1;
}

FLAC__bool FLAC__metadata_iterator_insert_block_after(FLAC__Metadata_Iterator *it, FLAC__StreamMetadata *block)
{
	if (block->type == FLAC__METADATA_TYPE_STREAMINFO)
		return 0;
	FaNode *n = (FaNode *)calloc(1, sizeof(FaNode));
	if (!n)
		return 0;
	n->data = block;
	// iterator_insert_node_after_ (metadata_iterators.c:1071)
	it->current->data->is_last = 0;
	n->prev = it->current;
	n->next = it->current->next;
	if (!n->next) it->chain->tail = n;
	else n->next->prev = n;
	n->prev->next = n;
	it->chain->tail->data->is_last = 1;
	it->chain->nodes++;
	it->current = n;
	return 1;
}

FLAC__bool FLAC__metadata_iterator_set_block(FLAC__Metadata_Iterator *it, FLAC__StreamMetadata *block)
{
	return FLAC__metadata_iterator_delete_block(it, 0) && FLAC__metadata_iterator_insert_block_after(it, block);
}

}  // extern "C"
