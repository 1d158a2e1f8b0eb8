// Internal types shared by the host layer and the HIP kernels.
#ifndef FLAC_AMD_INTERNAL_H
#define FLAC_AMD_INTERNAL_H

#include <stddef.h>
#include <stdint.h>

#include "flac_amd.h"

#define FA_MAX_SIGNALS (FLACAMD_MAX_CHANNELS + 2)  // input channels + mid + side
#define FA_MAX_ORDER FLACAMD_MAX_LPC_ORDER
#define FA_MAX_PARTS (1 << FLACAMD_MAX_PARTITION_ORDER)
#define FA_THREADS 256
#define FA_CHUNK 16                       // consecutive samples owned by one thread
#define FA_TILE (FA_THREADS * FA_CHUNK)   // samples a workgroup covers per pass

enum { FA_SUB_CONSTANT = 0, FA_SUB_VERBATIM = 1, FA_SUB_FIXED = 2, FA_SUB_LPC = 3 };

// Resolved encoder settings as the kernels see them (passed by value).
struct FaEncParams {
	uint32_t channels, bps, sample_rate, blocksize;
	uint32_t nsignals;            // channels, +2 when mid/side is evaluated
	uint32_t do_ms, loose_ms, loose_frames;
	uint32_t max_lpc_order, qlp_precision, prec_search, exhaustive;
	uint32_t min_po, max_po;
	uint32_t disable_constant, disable_fixed, disable_verbatim;
	uint32_t napod;
	uint32_t rice_limit;          // 15, or 31 when the stream is wider than 16 bits
	uint32_t wide_fixed;          // 64-bit sums in the fixed-predictor statistics
	uint32_t nprec;               // precision candidates per (apodization, order): 1 or 11
	uint32_t ncand;               // LPC candidate slots per subframe = napod * max_lpc_order * nprec
	uint32_t slot_bytes;          // bytes reserved per frame in the frame slot buffer
};

// One batch of consecutive frames.
struct FaBatch {
	uint64_t wide_samples;        // wide samples in pcm
	uint32_t nframes;
	uint32_t first_frame;
	uint32_t sample_bytes;        // 2, 3 or 4
	uint32_t loose_carry;         // assignment of the leader frame preceding this batch
	uint32_t pairs;               // 16-bit stereo on a 4-byte boundary: one 32-bit load per wide sample
	uint32_t planar;              // 0, or the blocksize: the kernels read 32-bit planes [frame][channel][blocksize] made by k_planes
};

// Per (frame, signal) statistics from the analysis pass.
struct FaSubStat {
	uint32_t wasted;
	uint32_t bps;                 // subframe bits per sample after wasted-bit removal
	uint32_t fixed_guess;
	uint32_t is_constant;
	float    rbps[5];
	int32_t  first_sample;
};

// One quantised LPC candidate.
struct FaLpcCand {
	uint8_t valid;
	uint8_t order;
	uint8_t precision;
	int8_t  shift;
	uint8_t rice;                 // Rice parameter seed used at partition order 0
	uint8_t wide;                 // 64-bit accumulation in the residual
	uint16_t sumq32;              // sum of the coefficients' magnitudes, in units of 32, rounded up
};

// What the model search chose for one (frame, signal).
struct FaPlan {
	uint32_t bits;                // the reference's bit estimate for the subframe
	uint8_t  type, order, precision;
	int8_t   shift;
	uint8_t  part_order, rice2, wide, pad;
	int32_t  constant;
	int32_t  q[FA_MAX_ORDER];
	uint8_t  params[FA_MAX_PARTS];
};

#ifdef __cplusplus
extern "C" {
#endif
// records a thread-local message and returns code
int fa_fail(int code, const char *fmt, ...);
#ifdef __cplusplus
}
#endif

#endif
