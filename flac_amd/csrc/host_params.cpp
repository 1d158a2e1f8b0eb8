// Host-side parameter resolution and apodization window tables.
//
// The windows are built on the host with the C library's cos()/exp() in the same arithmetic
// types the reference uses (src/libFLAC/window.c), so the float tables are the same bits the
// reference multiplies by; the device only ever sees the finished tables.
#include "flac_amd.h"
#include "flac_amd_internal.h"

#include <cmath>
#include <cstring>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace {

// src/libFLAC/stream_encoder.c:120-141 compression_levels_[]
struct Level { uint32_t ms, loose, lpc, exhaustive, maxpart; };
const Level kLevels[9] = {
	{0, 0, 0, 0, 3}, {1, 1, 0, 0, 3}, {1, 0, 0, 0, 3}, {0, 0, 6, 0, 4}, {1, 1, 8, 0, 4},
	{1, 0, 8, 0, 5}, {1, 0, 8, 0, 6}, {1, 0, 8, 1, 6}, {1, 0, 12, 1, 6},
};

void cosine_sum(float *w, int32_t L, double a0, double a1, double a2, double a3, double a4)
{
	// the a_k arrive already converted from the float literals the reference spells out
	const int32_t N = L - 1;
	for (int32_t n = 0; n < L; n++) {
		double v = a0 - a1 * cos(2.0f * M_PI * n / N);
		if (a2 != 0.0) v = v + a2 * cos(4.0f * M_PI * n / N);
		if (a3 != 0.0) v = v - a3 * cos(6.0f * M_PI * n / N);
		if (a4 != 0.0) v = v + a4 * cos(8.0f * M_PI * n / N);
		w[n] = (float)v;
	}
}

}  // namespace

extern "C" int flacamd_window(uint32_t type, float param, uint32_t Lu, float *w)
{
	if (!w || Lu == 0)
		return fa_fail(FLACAMD_E_INVALID, "flacamd_window: empty window");
	const int32_t L = (int32_t)Lu;
	const int32_t N = L - 1;
	int32_t n;
	switch (type) {
	case FLACAMD_APOD_BARTLETT:  // window.c:49
		if (L & 1) {
			for (n = 0; n <= N / 2; n++) w[n] = 2.0f * n / (float)N;
			for (; n <= N; n++) w[n] = 2.0f - 2.0f * n / (float)N;
		} else {
			for (n = 0; n <= L / 2 - 1; n++) w[n] = 2.0f * n / (float)N;
			for (; n <= N; n++) w[n] = 2.0f - 2.0f * (N - n) / (float)N;
		}
		break;
	case FLACAMD_APOD_BARTLETT_HANN:  // window.c:68
		for (n = 0; n < L; n++)
			w[n] = (float)(0.62f - 0.48f * fabs((float)n / (float)N + 0.5f) + 0.38f * cos(2.0f * M_PI * ((float)n / (float)N + 0.5f)));
		break;
	case FLACAMD_APOD_BLACKMAN:  // window.c:77
		cosine_sum(w, L, 0.42f, 0.5f, 0.08f, 0, 0);
		break;
	case FLACAMD_APOD_BLACKMAN_HARRIS_4TERM_92DB:  // window.c:87
		cosine_sum(w, L, 0.35875f, 0.48829f, 0.14128f, 0.01168f, 0);
		break;
	case FLACAMD_APOD_CONNES: {  // window.c:96
		const double N2 = (double)N / 2.;
		for (n = 0; n <= N; n++) {
			double k = ((double)n - N2) / N2;
			k = 1.0f - k * k;
			w[n] = (float)(k * k);
		}
		break;
	}
	case FLACAMD_APOD_FLATTOP:  // window.c:109
		cosine_sum(w, L, 1.0f, 1.93f, 1.29f, 0.388f, 0.0322f);
		break;
	case FLACAMD_APOD_GAUSS: {  // window.c:118
		const double N2 = (double)N / 2.;
		for (n = 0; n <= N; n++) {
			const double k = ((double)n - N2) / (param * N2);
			w[n] = (float)exp(-0.5f * k * k);
		}
		break;
	}
	case FLACAMD_APOD_HAMMING:  // window.c:130
		cosine_sum(w, L, 0.54f, 0.46f, 0, 0, 0);
		break;
	case FLACAMD_APOD_HANN:  // window.c:139
		cosine_sum(w, L, 0.5f, 0.5f, 0, 0, 0);
		break;
	case FLACAMD_APOD_KAISER_BESSEL:  // window.c:148
		cosine_sum(w, L, 0.402f, 0.498f, 0.098f, 0.001f, 0);
		break;
	case FLACAMD_APOD_NUTTALL:  // window.c:157
		cosine_sum(w, L, 0.3635819f, 0.4891775f, 0.1365995f, 0.0106411f, 0);
		break;
	case FLACAMD_APOD_RECTANGLE:  // window.c:166
		for (n = 0; n < L; n++) w[n] = 1.0f;
		break;
	case FLACAMD_APOD_TRIANGLE:  // window.c:174 (odd lengths keep the reference's loop bound L+1/2)
		if (L & 1) {
			for (n = 1; n <= L + 1 / 2; n++) w[n - 1] = 2.0f * n / ((float)L + 1.0f);
			for (; n <= L; n++) w[n - 1] = -(float)(2 * (L - n + 1)) / ((float)L + 1.0f);
		} else {
			for (n = 1; n <= L / 2; n++) w[n - 1] = 2.0f * n / (float)L;
			for (; n <= L; n++) w[n - 1] = ((float)(2 * (L - n)) + 1.0f) / (float)L;
		}
		break;
	case FLACAMD_APOD_TUKEY: {  // window.c:192
		const float p = param;
		if (p <= 0.0)
			return flacamd_window(FLACAMD_APOD_RECTANGLE, 0, Lu, w);
		if (p >= 1.0)
			return flacamd_window(FLACAMD_APOD_HANN, 0, Lu, w);
		const int32_t Np = (int32_t)(p / 2.0f * L) - 1;
		for (n = 0; n < L; n++) w[n] = 1.0f;
		if (Np > 0)
			for (n = 0; n <= Np; n++) {
				w[n] = (float)(0.5f - 0.5f * cos(M_PI * n / Np));
				w[L - Np - 1 + n] = (float)(0.5f - 0.5f * cos(M_PI * (n + Np) / Np));
			}
		break;
	}
	case FLACAMD_APOD_WELCH: {  // window.c:213
		const double N2 = (double)N / 2.;
		for (n = 0; n <= N; n++) {
			const double k = ((double)n - N2) / N2;
			w[n] = (float)(1.0f - k * k);
		}
		break;
	}
	default:
		return fa_fail(FLACAMD_E_INVALID, "flacamd_window: unknown apodization %u", type);
	}
	return FLACAMD_OK;
}

extern "C" int flacamd_params_resolve(flacamd_encode_params *p)
{
	if (!p)
		return fa_fail(FLACAMD_E_INVALID, "null params");
	// stream_encoder.c:660-668
	if (p->channels == 0 || p->channels > FLACAMD_MAX_CHANNELS)
		return fa_fail(FLACAMD_E_INVALID, "invalid number of channels %u", p->channels);
	if (p->channels != 2) {
		p->do_mid_side_stereo = 0;
		p->loose_mid_side_stereo = 0;
	} else if (!p->do_mid_side_stereo)
		p->loose_mid_side_stereo = 0;
	// :670-677
	if (p->bits_per_sample >= 32)
		p->do_mid_side_stereo = 0;
	if (p->bits_per_sample < 4 || p->bits_per_sample > 24)
		return fa_fail(FLACAMD_E_INVALID, "invalid bits per sample %u", p->bits_per_sample);
	if (p->sample_rate == 0 || p->sample_rate > 655350)
		return fa_fail(FLACAMD_E_INVALID, "invalid sample rate %u", p->sample_rate);
	// :679-693
	if (p->blocksize == 0)
		p->blocksize = p->max_lpc_order == 0 ? 1152 : 4096;
	if (p->blocksize < 16 || p->blocksize > 65535)
		return fa_fail(FLACAMD_E_INVALID, "invalid blocksize %u", p->blocksize);
	if (p->max_lpc_order > FLACAMD_MAX_LPC_ORDER)
		return fa_fail(FLACAMD_E_INVALID, "invalid max LPC order %u", p->max_lpc_order);
	if (p->blocksize < p->max_lpc_order)
		return fa_fail(FLACAMD_E_INVALID, "blocksize %u too small for LPC order %u", p->blocksize, p->max_lpc_order);
	// :695-728
	if (p->qlp_coeff_precision == 0) {
		const uint32_t bs = p->blocksize;
		if (p->bits_per_sample < 16) {
			const uint32_t g = 2 + p->bits_per_sample / 2;
			p->qlp_coeff_precision = g > 5 ? g : 5;
		} else if (p->bits_per_sample == 16)
			p->qlp_coeff_precision = bs <= 192 ? 7 : bs <= 384 ? 8 : bs <= 576 ? 9 : bs <= 1152 ? 10 : bs <= 2304 ? 11 : bs <= 4608 ? 12 : 13;
		else
			p->qlp_coeff_precision = bs <= 384 ? 13 : bs <= 1152 ? 14 : 15;
	} else if (p->qlp_coeff_precision < 5 || p->qlp_coeff_precision > 15)
		return fa_fail(FLACAMD_E_INVALID, "invalid qlp coefficient precision %u", p->qlp_coeff_precision);
	// :756-759
	if (p->max_residual_partition_order >= 16)
		p->max_residual_partition_order = 15;
	if (p->min_residual_partition_order >= p->max_residual_partition_order)
		p->min_residual_partition_order = p->max_residual_partition_order;
	if (p->num_apodizations > FLACAMD_MAX_APODIZATIONS)
		return fa_fail(FLACAMD_E_INVALID, "too many apodizations");
	if (p->num_apodizations == 0) {  // the setters never leave zero windows (:1585-1591)
		p->num_apodizations = 1;
		p->apod_type[0] = FLACAMD_APOD_TUKEY;
		p->apod_param[0] = 0.5f;
	}
	return FLACAMD_OK;
}

extern "C" int flacamd_params_preset(flacamd_encode_params *p, uint32_t level, uint32_t channels,
                                     uint32_t bits_per_sample, uint32_t sample_rate, uint32_t blocksize)
{
	if (!p)
		return fa_fail(FLACAMD_E_INVALID, "null params");
	memset(p, 0, sizeof *p);
	if (level > 8)
		level = 8;
	p->channels = channels;
	p->bits_per_sample = bits_per_sample;
	p->sample_rate = sample_rate;
	p->blocksize = blocksize;
	p->do_mid_side_stereo = kLevels[level].ms;
	p->loose_mid_side_stereo = kLevels[level].loose;
	p->max_lpc_order = kLevels[level].lpc;
	p->do_exhaustive_model_search = kLevels[level].exhaustive;
	p->max_residual_partition_order = kLevels[level].maxpart;
	p->num_apodizations = 1;
	p->apod_type[0] = FLACAMD_APOD_TUKEY;
	p->apod_param[0] = 0.5f;
	return flacamd_params_resolve(p);
}
