// PCM container samples <-> native samples on the device: the byte shuffling the reference's
// command line does on the host either side of the codec (format_input, src/flac/encode.c:2333;
// the decoder's write callback, src/flac/decode.c:957), so that PCM crosses PCIe at the width it has
// in the file.  Both kernels are pure streaming work bound by HBM bandwidth: a thread converts
// four consecutive samples, reading and writing whole 32-bit words when the buffers allow it.
#include <hip/hip_runtime.h>

#include "flac_amd_internal.h"
#include "flac_amd_pcm.h"

#define PCM_THREADS 256
#define PCM_PER_THREAD 4

// bytes [at, at + n) of a buffer, n <= 12, as up to three little-endian words: word loads when the
// whole run is 4-byte aligned and inside the buffer, byte loads otherwise
template <int NBYTES>
__device__ __forceinline__ void load_run(const uint8_t *base, uint64_t at, uint64_t total, bool aligned, uint32_t (&w)[(NBYTES + 3) / 4])
{
	constexpr int NW = (NBYTES + 3) / 4;
	if (aligned && at + 4 * NW <= total) {
		const uint32_t *p = (const uint32_t *)(base + at);
#pragma unroll
		for (int i = 0; i < NW; i++)
			w[i] = p[i];
	} else {
#pragma unroll
		for (int i = 0; i < NW; i++)
			w[i] = 0;
#pragma unroll
		for (int i = 0; i < NBYTES; i++)
			if (at + i < total)
				w[i / 4] |= (uint32_t)base[at + i] << (8 * (i % 4));
	}
}

__device__ __forceinline__ uint32_t run_byte(const uint32_t *w, int i) { return (w[i / 4] >> (8 * (i % 4))) & 0xff; }

// format_input (encode.c:2333-2423): BYTES per sample in the file, big- or little-endian, signed or
// unsigned, `shift` dead low bits; OUT bytes per native sample
template <int BYTES, int OUT>
__global__ __launch_bounds__(PCM_THREADS) void k_pcm_unpack(const uint8_t *src, uint64_t nsamples, uint32_t big_endian, uint32_t is_unsigned,
                                                             uint32_t shift, uint8_t *dst, uint32_t *flag, uint32_t src_aligned)
{
	const uint64_t first = ((uint64_t)blockIdx.x * PCM_THREADS + threadIdx.x) * PCM_PER_THREAD;
	if (first >= nsamples)
		return;
	constexpr int NB = BYTES * PCM_PER_THREAD;
	uint32_t w[(NB + 3) / 4];
	load_run<NB>(src, first * BYTES, nsamples * BYTES, src_aligned != 0, w);
	int32_t v[PCM_PER_THREAD];
	uint32_t dead = 0;
#pragma unroll
	for (int s = 0; s < PCM_PER_THREAD; s++) {
		uint32_t x = 0;
#pragma unroll
		for (int b = 0; b < BYTES; b++) {
			const uint32_t byte = run_byte(w, s * BYTES + b);
			x |= byte << (8 * (big_endian ? BYTES - 1 - b : b));
		}
		int32_t sv;
		if (is_unsigned)
			sv = (int32_t)x - (int32_t)(1u << (8 * BYTES - 1));
		else
			sv = (int32_t)(x << (32 - 8 * BYTES)) >> (32 - 8 * BYTES);
		if (first + s < nsamples)
			dead |= (uint32_t)sv & ((1u << shift) - 1u);
		v[s] = sv >> shift;
	}
	if (dead)
		*flag = 1;  // every writer stores the same value
	if (first + PCM_PER_THREAD <= nsamples) {
		if (OUT == 2) {
			uint2 o;
			o.x = ((uint32_t)v[0] & 0xffff) | ((uint32_t)v[1] << 16);
			o.y = ((uint32_t)v[2] & 0xffff) | ((uint32_t)v[3] << 16);
			*(uint2 *)(dst + first * 2) = o;
		} else {
			*(int4 *)(dst + first * 4) = make_int4(v[0], v[1], v[2], v[3]);
		}
	} else {
		for (int s = 0; s < PCM_PER_THREAD && first + s < nsamples; s++) {
			if (OUT == 2)
				((int16_t *)dst)[first + s] = (int16_t)v[s];
			else
				((int32_t *)dst)[first + s] = v[s];
		}
	}
}

// the sample formatting of decode.c:1094-1236: IN bytes per native sample, BYTES per file sample
template <int IN, int BYTES>
__global__ __launch_bounds__(PCM_THREADS) void k_pcm_pack(const uint8_t *src, uint64_t nsamples, uint32_t big_endian, uint32_t is_unsigned, uint32_t shift,
                                                           uint8_t *dst, uint32_t dst_aligned)
{
	const uint64_t first = ((uint64_t)blockIdx.x * PCM_THREADS + threadIdx.x) * PCM_PER_THREAD;
	if (first >= nsamples)
		return;
	int32_t v[PCM_PER_THREAD];
	if (first + PCM_PER_THREAD <= nsamples) {
		if (IN == 2) {
			const uint2 i = *(const uint2 *)(src + first * 2);
			v[0] = (int32_t)(int16_t)(i.x & 0xffff);
			v[1] = (int32_t)i.x >> 16;
			v[2] = (int32_t)(int16_t)(i.y & 0xffff);
			v[3] = (int32_t)i.y >> 16;
		} else {
			const int4 i = *(const int4 *)(src + first * 4);
			v[0] = i.x; v[1] = i.y; v[2] = i.z; v[3] = i.w;
		}
	} else {
#pragma unroll
		for (int s = 0; s < PCM_PER_THREAD; s++) {
			v[s] = 0;
			if (first + s < nsamples)
				v[s] = IN == 2 ? (int32_t)((const int16_t *)src)[first + s] : ((const int32_t *)src)[first + s];
		}
	}
	constexpr int NB = BYTES * PCM_PER_THREAD;
	uint32_t w[(NB + 3) / 4];
#pragma unroll
	for (int i = 0; i < (NB + 3) / 4; i++)
		w[i] = 0;
#pragma unroll
	for (int s = 0; s < PCM_PER_THREAD; s++) {
		uint32_t x = (uint32_t)v[s] << shift;
		if (is_unsigned)
			x += 1u << (8 * BYTES - 1);
#pragma unroll
		for (int b = 0; b < BYTES; b++) {
			const uint32_t byte = (x >> (8 * (big_endian ? BYTES - 1 - b : b))) & 0xff;
			const int at = s * BYTES + b;
			w[at / 4] |= byte << (8 * (at % 4));
		}
	}
	const uint64_t at = first * BYTES, total = nsamples * BYTES;
	if (dst_aligned && first + PCM_PER_THREAD <= nsamples) {   // NB is a multiple of 4 and so is `at`
		uint32_t *p = (uint32_t *)(dst + at);
#pragma unroll
		for (int i = 0; i < NB / 4; i++)
			p[i] = w[i];
	} else {
#pragma unroll
		for (int i = 0; i < NB; i++)
			if (at + i < total)
				dst[at + i] = (uint8_t)run_byte(w, i);
	}
}

extern "C" int flacamd_pcm_unpack(const void *src_dev, const flacamd_pcm_format *fmt, uint64_t wide_samples, void *dst_dev, uint32_t dst_sample_bytes,
                                  uint32_t *flag_dev, void *stream)
{
	if (!src_dev || !fmt || !dst_dev || !flag_dev)
		return fa_fail(FLACAMD_E_INVALID, "flacamd_pcm_unpack: null argument");
	if (fmt->bits_per_sample != 8 && fmt->bits_per_sample != 16 && fmt->bits_per_sample != 24)
		return fa_fail(FLACAMD_E_INVALID, "unsupported bits-per-sample %u", fmt->bits_per_sample);
	if (fmt->shift >= fmt->bits_per_sample)
		return fa_fail(FLACAMD_E_INVALID, "shift %u leaves no bits of %u", fmt->shift, fmt->bits_per_sample);
	if (dst_sample_bytes != 2 && dst_sample_bytes != 4)
		return fa_fail(FLACAMD_E_INVALID, "native samples are 2 or 4 bytes wide");
	if (dst_sample_bytes * 8 < fmt->bits_per_sample - fmt->shift)
		return fa_fail(FLACAMD_E_INVALID, "%u-byte samples cannot hold %u bits", dst_sample_bytes, fmt->bits_per_sample - fmt->shift);
	if (fmt->channels == 0 || fmt->channels > FLACAMD_MAX_CHANNELS)
		return fa_fail(FLACAMD_E_INVALID, "invalid number of channels %u", fmt->channels);
	if (((uintptr_t)dst_dev & 15) != 0)
		return fa_fail(FLACAMD_E_INVALID, "flacamd_pcm_unpack: the destination must be 16-byte aligned");
	const uint64_t n = wide_samples * fmt->channels;
	if (n == 0)
		return 0;
	const uint64_t blocks = (n + PCM_THREADS * PCM_PER_THREAD - 1) / (PCM_THREADS * PCM_PER_THREAD);
	if (blocks > 0x7fffffffull)
		return fa_fail(FLACAMD_E_INVALID, "flacamd_pcm_unpack: too many samples for one call");
	const uint32_t aligned = ((uintptr_t)src_dev & 3) == 0;
	hipStream_t st = (hipStream_t)stream;
	const dim3 grid((uint32_t)blocks), block(PCM_THREADS);
	const uint8_t *s = (const uint8_t *)src_dev;
	uint8_t *d = (uint8_t *)dst_dev;
#define UNPACK(B, O) hipLaunchKernelGGL((k_pcm_unpack<B, O>), grid, block, 0, st, s, n, fmt->is_big_endian, fmt->is_unsigned, fmt->shift, d, flag_dev, aligned)
	switch (fmt->bits_per_sample / 8 * 10 + dst_sample_bytes) {
	case 12: UNPACK(1, 2); break;
	case 14: UNPACK(1, 4); break;
	case 22: UNPACK(2, 2); break;
	case 24: UNPACK(2, 4); break;
	case 32: UNPACK(3, 2); break;
	default: UNPACK(3, 4); break;
	}
#undef UNPACK
	if (hipGetLastError() != hipSuccess)
		return fa_fail(FLACAMD_E_DEVICE, "flacamd_pcm_unpack: kernel launch failed");
	return 0;
}

extern "C" int flacamd_pcm_pack(const void *src_dev, uint32_t src_sample_bytes, uint32_t channels, uint32_t bits_per_sample, uint32_t container,
                                uint64_t wide_samples, void *dst_dev, void *stream)
{
	if (!src_dev || !dst_dev)
		return fa_fail(FLACAMD_E_INVALID, "flacamd_pcm_pack: null argument");
	if (src_sample_bytes != 2 && src_sample_bytes != 4)
		return fa_fail(FLACAMD_E_INVALID, "native samples are 2 or 4 bytes wide");
	if (bits_per_sample < 4 || bits_per_sample > 24 || src_sample_bytes * 8 < bits_per_sample)
		return fa_fail(FLACAMD_E_INVALID, "unsupported bits-per-sample %u for %u-byte samples", bits_per_sample, src_sample_bytes);
	if (channels == 0 || channels > FLACAMD_MAX_CHANNELS)
		return fa_fail(FLACAMD_E_INVALID, "invalid number of channels %u", channels);
	if (((uintptr_t)src_dev & 15) != 0)
		return fa_fail(FLACAMD_E_INVALID, "flacamd_pcm_pack: the source must be 16-byte aligned");
	// decode.c:962-972
	const bool aiff = container == FLACAMD_PCM_AIFF || container == FLACAMD_PCM_AIFF_C;
	const bool wave = container == FLACAMD_PCM_WAVE || container == FLACAMD_PCM_WAVE64 || container == FLACAMD_PCM_RF64;
	if (!aiff && !wave)
		return fa_fail(FLACAMD_E_INVALID, "flacamd_pcm_pack: container %u", container);
	const uint32_t shift = (bits_per_sample % 8) ? 8 - (bits_per_sample % 8) : 0;
	const uint32_t big_endian = aiff ? 1 : 0;
	const uint32_t is_unsigned = wave && bits_per_sample <= 8 ? 1 : 0;
	const uint32_t bytes = (bits_per_sample + shift) / 8;
	const uint64_t n = wide_samples * channels;
	if (n == 0)
		return 0;
	const uint64_t blocks = (n + PCM_THREADS * PCM_PER_THREAD - 1) / (PCM_THREADS * PCM_PER_THREAD);
	if (blocks > 0x7fffffffull)
		return fa_fail(FLACAMD_E_INVALID, "flacamd_pcm_pack: too many samples for one call");
	const uint32_t aligned = ((uintptr_t)dst_dev & 3) == 0;
	hipStream_t st = (hipStream_t)stream;
	const dim3 grid((uint32_t)blocks), block(PCM_THREADS);
	const uint8_t *s = (const uint8_t *)src_dev;
	uint8_t *d = (uint8_t *)dst_dev;
#define PACK(I, B) hipLaunchKernelGGL((k_pcm_pack<I, B>), grid, block, 0, st, s, n, big_endian, is_unsigned, shift, d, aligned)
	switch (src_sample_bytes * 10 + bytes) {
	case 21: PACK(2, 1); break;
	case 22: PACK(2, 2); break;
	case 41: PACK(4, 1); break;
	case 42: PACK(4, 2); break;
	default: PACK(4, 3); break;
	}
#undef PACK
	if (hipGetLastError() != hipSuccess)
		return fa_fail(FLACAMD_E_DEVICE, "flacamd_pcm_pack: kernel launch failed");
	return 0;
}
