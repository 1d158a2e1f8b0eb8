// Shared declarations of the libFLAC-compatible stream layer.
#ifndef FLAC_AMD_STREAM_COMMON_H
#define FLAC_AMD_STREAM_COMMON_H

#include <vector>

#include "flac_amd_stream.h"

// FLAC__add_metadata_block (stream_encoder_framing.c:50): block header + body, appended to out.
// own_vendor: a Vorbis comment goes out with this library's vendor string, as the encoder writes it;
// otherwise with the object's own (write_metadata_block_data_vorbis_comment_cb_, metadata_iterators.c:2640).
void fa_serialize_metadata(const FLAC__StreamMetadata *m, std::vector<uint8_t> &out, bool own_vendor = true);

// HIP device the stream layer runs on: environment variable FLAC_AMD_DEVICE, default 0.
int fa_stream_device(void);

#endif
