// Types and launchers of the decoder kernels (dec_kernels.hip).
#ifndef FLAC_AMD_DEC_KERNELS_H
#define FLAC_AMD_DEC_KERNELS_H

#include <hip/hip_runtime.h>

#include "flac_amd_internal.h"

struct FaSubInfo;

struct FaDecParams {
	uint32_t channels, bps, max_blocksize, out_sample_bytes;
	uint32_t nstreams, max_frames;
	uint32_t number_scale;   // samples per frame number in fixed-blocksize headers
	uint32_t lw;             // frames per wave in k_dec_frames and per group in the scratch planes (power of two <= 64)
	uint32_t skip_lw;        // frames per wave in the walk of the split form (which does not touch the planes)
	uint32_t run;            // samples between the frame decoder's service points where nothing but plain symbols lies ahead (16, 32, 64)
	uint64_t flac_bytes;
	uint64_t out_capacity_wide;
	FaSubInfo *subinfo;      // max_frames * channels records to fill, or null
	uint32_t planar;         // 1: a frame's samples leave as [channel][blocksize] in the frame's stretch of the output, not interleaved
	uint32_t pad;
};

// One record per (frame, channel) when the caller wants them: the layout of flacamd_subframe_info (include/flac_amd.h).
#define FA_INFO_PARAMS 256
struct FaSubInfo {
	uint8_t type, order, wasted, precision;
	int8_t shift;
	uint8_t po, rice2, pad;
	int32_t constant;
	uint32_t pad2[2];
	int32_t q[32];
	int32_t warm[32];
	uint8_t params[FA_INFO_PARAMS];
};

struct FaStream {
	uint64_t byte_offset, byte_length, sample_offset, total_samples, first_sample;
};

// A byte position that holds a sync code and a valid frame header.
struct FaFrameCand {
	uint64_t pos;        // byte offset of the sync code
	uint64_t sample;     // number of the frame's first sample, from the header
	uint64_t end_byte;   // first byte after the frame (after the CRC-16)
	uint32_t hdr_bytes;
	uint32_t blocksize;
	uint32_t assignment;
	uint32_t stream;
	uint32_t status;     // 0 not decoded, 1 good, 2 structure error, 3 CRC-16 mismatch, 4 cut off by the end of input
	uint32_t reserved;   // bit 0, with status 2: the parse stopped at a reserved code; bit 1: the frame's samples are in the output already
};

hipError_t fa_dec_upload_tables(const uint16_t *t16, const uint16_t *pow16, const uint8_t *t8);
uint32_t fa_dec_sync_blocks(uint64_t flac_bytes);
// (also leaves the count of headers in ncand[0], the overflow flag in ncand[1], and clears the three counters of `result`)
// kept: FA_SYNC_KEEP rows per 4 KiB of input (fa_dec_sync_blocks), scratch between the two passes
#define FA_SYNC_KEEP 8
void fa_dec_launch_sync(const FaDecParams &P, const uint8_t *flac, const FaStream *streams, uint32_t *block_counts, FaFrameCand *kept,
                        FaFrameCand *cands, uint32_t *ncand, unsigned long long *result, hipStream_t st);
// sub_start: max_frames * channels words, the bit position at which each (frame, channel)'s subframe starts.
// split: a walk that leaves every start, then a lane per subframe in one launch; else one launch per channel
// pair (with split, two channels, fa_dec_pair_fits): P.lw is frames per group and a workgroup takes both channels of them and writes
// final PCM to `out` for the frames whose CRC-16 the pass that ran ahead found right (`ahead`, complete after `before_sub`)
bool fa_dec_pair_fits(uint32_t channels, uint32_t max_frames, uint32_t frames_per_group);
void fa_dec_launch_frames(const FaDecParams &P, const uint8_t *flac, const FaStream *streams, FaFrameCand *cands, const uint32_t *ncand,
                          int32_t *tmp, uint64_t *sub_start, bool split, bool pair, const uint64_t *ahead, hipEvent_t before_sub, void *out,
                          hipStream_t st);
// before_frames: CRC-16 of every header's frame on the guess that it ends where the next header was found, into `ahead`
// (max_frames words; runs beside the frame decoder); else the pass after the frame decoder, which takes from `ahead`
// (may be null) what was guessed right
void fa_dec_launch_crc(const FaDecParams &P, const uint8_t *flac, FaFrameCand *cands, const uint32_t *ncand, const FaStream *streams, uint64_t *ahead,
                       bool before_frames, hipStream_t st);
void fa_dec_launch_output(const FaDecParams &P, const FaStream *streams, const FaFrameCand *cands, const uint32_t *ncand, const int32_t *tmp,
                          void *out, unsigned long long *result, hipStream_t st);

#endif
