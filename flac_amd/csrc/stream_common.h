// Shared declarations of the libFLAC-compatible stream layer.
#ifndef FLAC_AMD_STREAM_COMMON_H
#define FLAC_AMD_STREAM_COMMON_H

#include <vector>

#include "flac_amd_stream.h"

// FLAC__add_metadata_block (stream_encoder_framing.c:50): block header + body, appended to out.
void fa_serialize_metadata(const FLAC__StreamMetadata *m, std::vector<uint8_t> &out);

// HIP device the stream layer runs on: environment variable FLAC_AMD_DEVICE, default 0.
int fa_stream_device(void);

#endif
