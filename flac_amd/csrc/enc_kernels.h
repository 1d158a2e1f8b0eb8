// Launchers of the encoder kernels (enc_analyze.hip, enc_search.hip, enc_pack.hip).
#ifndef FLAC_AMD_ENC_KERNELS_H
#define FLAC_AMD_ENC_KERNELS_H

#include <hip/hip_runtime.h>

#include "flac_amd_internal.h"

hipError_t fa_upload_crc_tables(const uint16_t *t16, const uint16_t *pow16, const uint8_t *t8);
// interleaved samples of any width -> 32-bit planes [frame][channel][blocksize]
void fa_launch_planes(const FaEncParams &P, const FaBatch &B, const void *pcm, int32_t *planes, hipStream_t st);
// planes (or none: null) and statistics in one pass (streams that fa_planes_stats_fit): acc is scratch of fa_stat_acc_bytes
size_t fa_stat_acc_bytes(uint32_t nframes, uint32_t nsignals);
bool fa_planes_stats_fit(const FaEncParams &P);
void fa_launch_planes_stats(const FaEncParams &P, const FaBatch &B, const void *pcm, int32_t *planes, void *acc, FaSubStat *stats, hipStream_t st);
void fa_launch_analyze(const FaEncParams &P, const FaBatch &B, const void *pcm, FaSubStat *stats, hipStream_t st);
void fa_launch_autoc(const FaEncParams &P, const FaBatch &B, const void *pcm, const FaSubStat *stats, const float *windows, float *autoc,
                     hipStream_t st);
void fa_launch_lpc(const FaEncParams &P, const FaBatch &B, const FaSubStat *stats, const float *autoc, FaLpcCand *cands, int32_t *coefs,
                   hipStream_t st);
void fa_launch_search(const FaEncParams &P, const FaBatch &B, const void *pcm, const FaSubStat *stats, const FaLpcCand *cands,
                      const int32_t *coefs, FaPlan *plans, unsigned long long *stamps, hipStream_t st);
// the wave-per-subframe form of the search (enc_search_wave.hip) for the listed signals of every frame
bool fa_search_wave_eligible(const FaEncParams &P);
bool fa_search_wave_split_eligible(const FaEncParams &P);
void fa_launch_search_wave_split(const FaEncParams &P, const FaBatch &B, const void *pcm, const FaSubStat *stats, const FaLpcCand *cands,
                                 const int32_t *coefs, FaPlan *plans, unsigned long long sig_list, uint32_t nsel, hipStream_t st);
void fa_launch_search_wave(const FaEncParams &P, const FaBatch &B, const void *pcm, const FaSubStat *stats, const FaLpcCand *cands,
                           const int32_t *coefs, FaPlan *plans, unsigned long long sig_list, uint32_t nsel, unsigned long long wide_list, uint32_t nwide,
                           hipStream_t st);
// tickets: one word per frame, cleared here for the frame writer's placement (see fa_launch_pack)
void fa_launch_decide(const FaEncParams &P, const FaBatch &B, const FaPlan *plans, uint32_t *assign, uint32_t *loose_out, unsigned long long *tickets,
                      hipStream_t st);
// Returns true when the frames went straight to their places in `out` (sizes in frame_bytes, places in offsets[0 .. nframes]):
// the streams k_pack_frame writes.  Otherwise they are in their slots, for fa_launch_scan and fa_launch_compact.
bool fa_launch_pack(const FaEncParams &P, const FaBatch &B, const void *pcm, const FaSubStat *stats, const FaPlan *plans, const uint32_t *assign,
                    uint32_t *slots, uint32_t *frame_bytes, uint32_t *errflag, unsigned long long *stamps, uint8_t *out, uint64_t out_capacity,
                    uint64_t *offsets, unsigned long long *tickets, hipStream_t st);
void fa_launch_scan(const uint32_t *frame_bytes, uint32_t n, uint64_t *offsets, hipStream_t st);
void fa_launch_compact(const FaEncParams &P, const FaBatch &B, const uint32_t *slots, const uint32_t *frame_bytes, const uint64_t *offsets,
                       uint8_t *out, uint64_t out_capacity, uint32_t *errflag, hipStream_t st);

#endif
