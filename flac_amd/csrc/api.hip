// C ABI of the frame engine: handle management, buffer sizing, kernel sequencing.
#include <hip/hip_runtime.h>

#include <emmintrin.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "enc_kernels.h"
#include "flac_amd.h"
#include "flac_amd_internal.h"

// ---------------------------------------------------------------------------------------------
// error reporting
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

extern "C" int fa_fail(int code, const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof g_err, fmt, ap);
	va_end(ap);
	return code;
}

extern "C" const char *flacamd_last_error(void) { return g_err; }
extern "C" const char *flacamd_version(void) { return "flac_amd 0.1 (gfx950)"; }

#define HIP_TRY(expr) \
	do { \
		hipError_t e_ = (expr); \
		if (e_ != hipSuccess) \
			return fa_fail(FLACAMD_E_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); \
	} while (0)

// ---------------------------------------------------------------------------------------------
// CRC tables (crc.c:40-110 holds the same tables as literals; generated here from the polynomials)
// ---------------------------------------------------------------------------------------------
static void make_crc_tables(uint16_t t16[256], uint16_t pow16[32], uint8_t t8[256])
{
	for (uint32_t i = 0; i < 256; i++) {
		uint32_t c = i << 8;
		for (int b = 0; b < 8; b++)
			c = (c & 0x8000) ? ((c << 1) ^ 0x8005) : (c << 1);
		t16[i] = (uint16_t)c;
		uint32_t d = i;
		for (int b = 0; b < 8; b++)
			d = (d & 0x80) ? ((d << 1) ^ 0x07) : (d << 1);
		t8[i] = (uint8_t)d;
	}
	// x^8 mod P, then repeated squaring
	auto mul = [](uint32_t a, uint32_t b) {
		uint32_t r = 0;
		for (int i = 0; i < 16; i++) {
			if (b & 1) r ^= a;
			b >>= 1;
			a <<= 1;
			if (a & 0x10000) a ^= 0x18005;
		}
		return r;
	};
	uint32_t p = 0x100;
	for (int j = 0; j < 32; j++) {
		pow16[j] = (uint16_t)p;
		p = mul(p, p);
	}
}

// ---------------------------------------------------------------------------------------------
// encoder handle
// ---------------------------------------------------------------------------------------------
struct flacamd_encoder {
	flacamd_encode_params params;
	FaEncParams P;
	int device;
	uint32_t max_frames;
	float *windows;
	FaSubStat *stats;
	float *autoc;
	FaLpcCand *cands;
	int32_t *coefs;
	FaPlan *plans;
	uint32_t *assign;
	uint32_t *slots;
	uint32_t *frame_bytes;
	uint64_t *offsets;
	uint32_t *errflag;    // [0] device error bits, [1] loose mid/side state out
	int32_t *planes;      // streams of three or more channels: the batch's samples as 32-bit planes per frame (k_planes)
	void *stat_acc;       // where the tiles of a block add up its statistics when the planes' pass makes them too
	// staging for the host-buffer entry point
	int32_t *stage_pcm;
	int16_t *stage_host16;  // pinned: samples narrowed to 16 bits before they cross PCIe
	uint8_t *stage_out;
	size_t stage_out_bytes;
	int profiling;
	hipEvent_t ev[FLACAMD_ENC_STAGES + 1];
	int have_times;
};

static size_t slot_bytes_for(const flacamd_encode_params &p)
{
	// A chosen subframe's estimate never exceeds the verbatim estimate, and its written size
	// exceeds its estimate by at most half a bit per sample plus one bit per partition.
	const size_t per_sub = ((size_t)p.blocksize * (p.bits_per_sample + 3) + 7) / 8 + 512;
	size_t s = 32 + per_sub * p.channels;
	return (s + 63) / 64 * 64 + 64;
}

// The kernels' view of resolved settings.
static void fill_kernel_params(const flacamd_encode_params &p, FaEncParams &P)
{
	P.channels = p.channels;
	P.bps = p.bits_per_sample;
	P.sample_rate = p.sample_rate;
	P.blocksize = p.blocksize;
	P.do_ms = p.do_mid_side_stereo ? 1 : 0;
	P.loose_ms = p.loose_mid_side_stereo ? 1 : 0;
	P.nsignals = p.channels + (P.do_ms ? 2 : 0);
	// stream_encoder.c:871
	P.loose_frames = (uint32_t)((double)p.sample_rate * 0.4 / (double)p.blocksize + 0.5);
	if (P.loose_frames == 0)
		P.loose_frames = 1;
	P.max_lpc_order = p.max_lpc_order;
	P.qlp_precision = p.qlp_coeff_precision;
	P.prec_search = p.do_qlp_coeff_prec_search ? 1 : 0;
	P.exhaustive = p.do_exhaustive_model_search ? 1 : 0;
	P.min_po = p.min_residual_partition_order;
	P.max_po = p.max_residual_partition_order;
	P.disable_constant = p.disable_constant_subframes ? 1 : 0;
	P.disable_fixed = p.disable_fixed_subframes ? 1 : 0;
	P.disable_verbatim = p.disable_verbatim_subframes ? 1 : 0;
	P.napod = p.num_apodizations;
	P.rice_limit = p.bits_per_sample > 16 ? 31 : 15;  // stream_encoder.c:3195
	uint32_t il = 0;
	for (uint32_t b = p.blocksize; b >>= 1;)
		il++;
	P.wide_fixed = (p.bits_per_sample + il + 1 > 30) ? 1 : 0;  // stream_encoder.c:887
	P.nprec = P.prec_search ? 11 : 1;
	P.ncand = P.napod * P.max_lpc_order * P.nprec;
	P.slot_bytes = (uint32_t)slot_bytes_for(p);
}

// what counts for the device limit is the order a full block can reach: the setting, capped by how
// often the blocksize halves evenly (stream_encoder.c:2929-2936)
static int check_partition_reach(const flacamd_encode_params &p)
{
	uint32_t reach = 0;
	while (reach < p.max_residual_partition_order && !((p.blocksize >> reach) & 1))
		reach++;
	if (reach > FLACAMD_MAX_PARTITION_ORDER)
		return fa_fail(FLACAMD_E_UNSUPPORTED, "Rice partition order %u (blocksize %u, max order setting %u) exceeds the device limit %d", reach,
		               p.blocksize, p.max_residual_partition_order, FLACAMD_MAX_PARTITION_ORDER);
	return FLACAMD_OK;
}

extern "C" size_t flacamd_encoder_max_output_bytes(const flacamd_encoder *e, uint32_t nframes)
{
	return e ? (size_t)e->P.slot_bytes * nframes : 0;
}

extern "C" uint32_t flacamd_encoder_max_frames(const flacamd_encoder *e) { return e ? e->max_frames : 0; }

extern "C" void flacamd_encoder_destroy(flacamd_encoder *e)
{
	if (!e)
		return;
	(void)hipSetDevice(e->device);
	(void)hipFree(e->windows);
	(void)hipFree(e->stats);
	(void)hipFree(e->autoc);
	(void)hipFree(e->cands);
	(void)hipFree(e->coefs);
	(void)hipFree(e->plans);
	(void)hipFree(e->assign);
	(void)hipFree(e->slots);
	(void)hipFree(e->planes);
	(void)hipFree(e->stat_acc);
	(void)hipFree(e->frame_bytes);
	(void)hipFree(e->offsets);
	(void)hipFree(e->errflag);
	(void)hipFree(e->stage_pcm);
	(void)hipHostFree(e->stage_host16);
	(void)hipFree(e->stage_out);
	for (int i = 0; i <= FLACAMD_ENC_STAGES; i++)
		if (e->ev[i])
			(void)hipEventDestroy(e->ev[i]);
	delete e;
}

extern "C" int flacamd_encoder_create(const flacamd_encode_params *pin, int device, uint32_t max_frames, flacamd_encoder **out)
{
	if (!pin || !out || max_frames == 0)
		return fa_fail(FLACAMD_E_INVALID, "flacamd_encoder_create: bad arguments");
	*out = nullptr;
	flacamd_encode_params p = *pin;
	int rc = flacamd_params_resolve(&p);
	if (rc)
		return rc;
	rc = check_partition_reach(p);
	if (rc)
		return rc;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
		return fa_fail(FLACAMD_E_DEVICE, "no HIP device available: the FLAC engine needs an AMD GPU");
	if (device < 0 || device >= ndev)
		return fa_fail(FLACAMD_E_DEVICE, "HIP device %d out of range (%d present)", device, ndev);
	HIP_TRY(hipSetDevice(device));

	flacamd_encoder *e = new (std::nothrow) flacamd_encoder();
	if (!e)
		return fa_fail(FLACAMD_E_NOMEM, "out of host memory");
	memset(e, 0, sizeof *e);
	e->params = p;
	e->device = device;
	e->max_frames = max_frames;
	FaEncParams &P = e->P;
	fill_kernel_params(p, P);

	const size_t F = max_frames, S = P.nsignals;
#define ALLOC(ptr, bytes) \
	do { \
		hipError_t e_ = hipMalloc((void **)&(ptr), (bytes) ? (bytes) : 16); \
		if (e_ != hipSuccess) { \
			flacamd_encoder_destroy(e); \
			return fa_fail(FLACAMD_E_NOMEM, "hipMalloc of %zu bytes failed: %s", (size_t)(bytes), hipGetErrorString(e_)); \
		} \
	} while (0)
	ALLOC(e->windows, sizeof(float) * P.napod * P.blocksize);
	ALLOC(e->stats, sizeof(FaSubStat) * F * S);
	ALLOC(e->autoc, sizeof(float) * F * S * P.napod * (FA_MAX_ORDER + 1));
	ALLOC(e->cands, sizeof(FaLpcCand) * F * S * P.ncand);
	ALLOC(e->coefs, sizeof(int32_t) * F * S * P.ncand * P.max_lpc_order + 64);   // slack: the wave search reads twelve coefficients per slot whatever the order
	ALLOC(e->plans, sizeof(FaPlan) * F * S);
	ALLOC(e->assign, sizeof(uint32_t) * (F + 2) + sizeof(uint64_t) * F * (1 + p.channels));   // and the frame writers' tickets (run_batch)
	ALLOC(e->slots, (size_t)P.slot_bytes * F + 64);
	// Three or more interleaved channels: a workgroup that takes ONE channel out of them uses 4 of every 4 x channels
	// bytes it fetches, and every channel's workgroup fetches every line.  Such streams are laid out again first, frame
	// by frame, as planes (one pass over the samples; FLAC_AMD_PLANES=0 switches it off, =1 on for two channels too).
	{
		const char *env = getenv("FLAC_AMD_PLANES");
		const bool on = env ? (atoi(env) != 0 && p.channels >= 2) : p.channels >= 3;
		if (on)
			ALLOC(e->planes, sizeof(int32_t) * F * p.channels * p.blocksize);
		if (fa_planes_stats_fit(P))
			ALLOC(e->stat_acc, fa_stat_acc_bytes((uint32_t)F, P.nsignals));
	}
	ALLOC(e->frame_bytes, sizeof(uint32_t) * F);
	ALLOC(e->offsets, sizeof(uint64_t) * (F + 1));
	ALLOC(e->errflag, 16);
#undef ALLOC
	(void)hipMemset(e->errflag, 0, 16);

	// window tables, built on the host
	if (P.max_lpc_order > 0) {
		float *w = (float *)malloc(sizeof(float) * P.blocksize);
		if (!w) {
			flacamd_encoder_destroy(e);
			return fa_fail(FLACAMD_E_NOMEM, "out of host memory");
		}
		for (uint32_t a = 0; a < P.napod; a++) {
			rc = flacamd_window(p.apod_type[a], p.apod_param[a], P.blocksize, w);
			if (rc == FLACAMD_OK && hipMemcpy(e->windows + (size_t)a * P.blocksize, w, sizeof(float) * P.blocksize, hipMemcpyHostToDevice) != hipSuccess)
				rc = fa_fail(FLACAMD_E_DEVICE, "window upload failed");
			if (rc) {
				free(w);
				flacamd_encoder_destroy(e);
				return rc;
			}
		}
		free(w);
	}
	uint16_t t16[256], pow16[32];
	uint8_t t8[256];
	make_crc_tables(t16, pow16, t8);
	if (hipError_t up = fa_upload_crc_tables(t16, pow16, t8); up != hipSuccess) {
		flacamd_encoder_destroy(e);
		return fa_fail(FLACAMD_E_DEVICE, "CRC table upload failed: %s", hipGetErrorString(up));
	}
	for (int i = 0; i <= FLACAMD_ENC_STAGES; i++)
		if (hipEventCreate(&e->ev[i]) != hipSuccess) {
			flacamd_encoder_destroy(e);
			return fa_fail(FLACAMD_E_DEVICE, "hipEventCreate failed");
		}
	if (hipError_t sy = hipDeviceSynchronize(); sy != hipSuccess) {
		flacamd_encoder_destroy(e);
		return fa_fail(FLACAMD_E_DEVICE, "hipDeviceSynchronize failed: %s", hipGetErrorString(sy));
	}
	*out = e;
	return FLACAMD_OK;
}

extern "C" int flacamd_encoder_set_profiling(flacamd_encoder *e, int on)
{
	if (!e)
		return fa_fail(FLACAMD_E_INVALID, "null encoder");
	e->profiling = on ? 1 : 0;
	e->have_times = 0;
	return FLACAMD_OK;
}

static const char *kEncStage[FLACAMD_ENC_STAGES] = {"analyze", "autoc", "lpc", "search", "decide", "pack", "scan", "compact"};
extern "C" const char *flacamd_encoder_stage_name(int s) { return (s >= 0 && s < FLACAMD_ENC_STAGES) ? kEncStage[s] : ""; }

extern "C" int flacamd_encoder_stage_ms(flacamd_encoder *e, float ms[FLACAMD_ENC_STAGES])
{
	if (!e || !ms)
		return fa_fail(FLACAMD_E_INVALID, "bad arguments");
	if (!e->have_times)
		return fa_fail(FLACAMD_E_INVALID, "no profiled encode has run");
	HIP_TRY(hipSetDevice(e->device));
	HIP_TRY(hipEventSynchronize(e->ev[FLACAMD_ENC_STAGES]));
	for (int i = 0; i < FLACAMD_ENC_STAGES; i++)
		HIP_TRY(hipEventElapsedTime(&ms[i], e->ev[i], e->ev[i + 1]));
	return FLACAMD_OK;
}

// The device buffers one batch works in, whoever owns them.
struct EncWork {
	int32_t *planes;      // null: the kernels read the caller's samples as they lie
	void *stat_acc;       // not null: the planes' pass also makes the statistics
	const float *windows;
	FaSubStat *stats;
	float *autoc;
	FaLpcCand *cands;
	int32_t *coefs;
	FaPlan *plans;
	uint32_t *assign;
	uint32_t *slots;
	uint32_t *errflag;    // [0] device error bits, [1] loose mid/side state out
};

// Argument checks shared by the batch entry points; fills B and *nf.
static int check_batch(const FaEncParams &P, const void *pcm_dev, const void *out_dev, uint32_t sample_bytes, uint64_t wide_samples,
                       uint32_t first_frame, int is_last, const uint32_t *loose_state, FaBatch &B, uint64_t *nf_out)
{
	if (!pcm_dev || !out_dev)
		return fa_fail(FLACAMD_E_INVALID, "encode: null argument");
	if (sample_bytes < 2 || sample_bytes > 4)
		return fa_fail(FLACAMD_E_INVALID, "sample_bytes must be 2, 3 or 4");
	if (sample_bytes * 8 < P.bps)
		return fa_fail(FLACAMD_E_INVALID, "%u-byte samples cannot hold %u bits", sample_bytes, P.bps);
	const uint64_t full = wide_samples / P.blocksize;
	const uint32_t rem = (uint32_t)(wide_samples % P.blocksize);
	if (rem && !is_last)
		return fa_fail(FLACAMD_E_INVALID, "only the last batch may end in a short block (%u samples left over)", rem);
	const uint64_t nf = full + (rem ? 1 : 0);
	*nf_out = nf;
	if ((uint64_t)first_frame + nf > 0x7fffffffull)
		return fa_fail(FLACAMD_E_INVALID, "frame number exceeds 31 bits");
	B.wide_samples = wide_samples;
	B.nframes = (uint32_t)nf;
	B.first_frame = first_frame;
	B.sample_bytes = sample_bytes;
	B.loose_carry = loose_state ? *loose_state : 0;
	B.pairs = (sample_bytes == 2 && P.channels == 2 && ((uintptr_t)pcm_dev & 3) == 0) ? 1u : 0u;
	B.planar = 0;
	return FLACAMD_OK;
}

// The kernel sequence of one batch.  ev: FLACAMD_ENC_STAGES + 1 events to record between the
// stages, or null.
static int run_batch(const FaEncParams &P, const FaBatch &Bin, const EncWork &W, const void *pcm_in, void *out_dev, size_t out_capacity,
                     uint32_t *fb, uint64_t *fo, hipStream_t st, hipEvent_t *ev)
{
	FaBatch B = Bin;
	const void *pcm_dev = pcm_in;
	int evi = 0;
#define STAGE_MARK() \
	do { \
		if (ev) \
			HIP_TRY(hipEventRecord(ev[evi++], st)); \
	} while (0)
	const char *stamps_env = getenv("FLAC_AMD_STAMPS");
	STAGE_MARK();
	bool have_stats = false;
	// FLAC_AMD_ANALYZE=split: the planes and the statistics in two passes, k_analyze for every stream but 16-bit stereo (for comparisons)
	const char *am = getenv("FLAC_AMD_ANALYZE");
	const bool one_pass = W.stat_acc && !(am && !strcmp(am, "split"));
	if (W.planes && !B.pairs) {
		if (one_pass) {
			fa_launch_planes_stats(P, B, pcm_in, W.planes, W.stat_acc, W.stats, st);
			have_stats = true;
		} else
			fa_launch_planes(P, B, pcm_in, W.planes, st);
		B.planar = P.blocksize;
		B.sample_bytes = 4;
		pcm_dev = W.planes;
	}
	else if (one_pass && !(B.pairs && P.nsignals == 4 && !P.wide_fixed)) {
		// (16-bit stereo with mid/side keeps k_analyze_quad: one word per wide sample, sums of 32 bits)
		fa_launch_planes_stats(P, B, pcm_in, nullptr, W.stat_acc, W.stats, st);
		have_stats = true;
	}
	if (!have_stats)
		fa_launch_analyze(P, B, pcm_dev, W.stats, st);
	STAGE_MARK();
	if (P.max_lpc_order > 0)
		fa_launch_autoc(P, B, pcm_dev, W.stats, W.windows, W.autoc, st);
	STAGE_MARK();
	if (P.max_lpc_order > 0)
		fa_launch_lpc(P, B, W.stats, W.autoc, W.cands, W.coefs, st);
	STAGE_MARK();
	// FLAC_AMD_STAMPS: per-workgroup cycle stamps of k_search land in the (by now consumed) autocorrelation buffer
	fa_launch_search(P, B, pcm_dev, W.stats, W.cands, W.coefs, W.plans, (stamps_env && strcmp(stamps_env, "pack")) ? (unsigned long long *)W.autoc : nullptr, st);
	STAGE_MARK();
	// (the frame writers' tickets lie behind the assignments: one 64-bit word per frame, then one per frame and channel)
	unsigned long long *tickets = (unsigned long long *)(W.assign + ((B.nframes + 1) & ~1u));
	fa_launch_decide(P, B, W.plans, W.assign, W.errflag + 1, tickets, st);
	STAGE_MARK();
	// FLAC_AMD_STAMPS=pack: the same for k_pack (the candidate buffer is consumed by then)
	const bool placed = fa_launch_pack(P, B, pcm_dev, W.stats, W.plans, W.assign, W.slots, fb, W.errflag,
	                                   (stamps_env && !strcmp(stamps_env, "pack")) ? (unsigned long long *)W.autoc : nullptr, (uint8_t *)out_dev,
	                                   out_capacity, fo, tickets, st);
	STAGE_MARK();
	if (!placed)
		fa_launch_scan(fb, B.nframes, fo, st);
	STAGE_MARK();
	if (!placed)
		fa_launch_compact(P, B, W.slots, fb, fo, (uint8_t *)out_dev, out_capacity, W.errflag, st);
	STAGE_MARK();
#undef STAGE_MARK
	HIP_TRY(hipGetLastError());
	return FLACAMD_OK;
}

extern "C" int flacamd_encode_frames(flacamd_encoder *e, const void *pcm_dev, uint32_t sample_bytes, uint64_t wide_samples,
                                     uint32_t first_frame, int is_last, uint32_t *loose_state, void *out_dev, size_t out_capacity,
                                     uint32_t *frame_bytes_dev, uint64_t *frame_offsets_dev, uint32_t *nframes_out, void *stream)
{
	if (!e)
		return fa_fail(FLACAMD_E_INVALID, "flacamd_encode_frames: null argument");
	const FaEncParams &P = e->P;
	FaBatch B;
	uint64_t nf = 0;
	int rc = check_batch(P, pcm_dev, out_dev, sample_bytes, wide_samples, first_frame, is_last, loose_state, B, &nf);
	if (rc)
		return rc;
	if (nframes_out)
		*nframes_out = (uint32_t)nf;
	if (nf == 0)
		return FLACAMD_OK;
	if (nf > e->max_frames)
		return fa_fail(FLACAMD_E_INVALID, "batch of %llu frames exceeds the encoder's capacity of %u", (unsigned long long)nf, e->max_frames);
	HIP_TRY(hipSetDevice(e->device));
	hipStream_t st = (hipStream_t)stream;
	EncWork W;
	W.planes = e->planes;
	W.stat_acc = e->stat_acc;
	W.windows = e->windows;
	W.stats = e->stats;
	W.autoc = e->autoc;
	W.cands = e->cands;
	W.coefs = e->coefs;
	W.plans = e->plans;
	W.assign = e->assign;
	W.slots = e->slots;
	W.errflag = e->errflag;
	rc = run_batch(P, B, W, pcm_dev, out_dev, out_capacity, frame_bytes_dev ? frame_bytes_dev : e->frame_bytes,
	               frame_offsets_dev ? frame_offsets_dev : e->offsets, st, e->profiling ? e->ev : nullptr);
	if (rc)
		return rc;
	if (e->profiling)
		e->have_times = 1;
	if (loose_state && P.loose_ms) {
		// the next batch needs this batch's last decision
		HIP_TRY(hipMemcpyAsync(loose_state, e->errflag + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
	}
	return FLACAMD_OK;
}

extern "C" int flacamd_encoder_status(flacamd_encoder *e)
{
	if (!e)
		return fa_fail(FLACAMD_E_INVALID, "null encoder");
	HIP_TRY(hipSetDevice(e->device));
	uint32_t flag = 0;
	HIP_TRY(hipMemcpy(&flag, e->errflag, sizeof flag, hipMemcpyDeviceToHost));
	if (flag) {
		(void)hipMemset(e->errflag, 0, sizeof flag);
		if (flag & 2)
			return fa_fail(FLACAMD_E_OVERFLOW, "output buffer too small for the encoded frames");
		return fa_fail(FLACAMD_E_OVERFLOW, "a frame outgrew its slot (device error bits %u)", flag);
	}
	return FLACAMD_OK;
}

extern "C" int flacamd_encode_frames_host(flacamd_encoder *e, const int32_t *pcm, uint64_t wide_samples, uint32_t first_frame, int is_last,
                                          uint32_t *loose_state, uint8_t *out, size_t out_capacity, uint32_t *frame_bytes,
                                          uint32_t *nframes_out, size_t *bytes_out)
{
	if (!e || (!pcm && wide_samples) || !out || !frame_bytes)
		return fa_fail(FLACAMD_E_INVALID, "flacamd_encode_frames_host: null argument");
	if (nframes_out)
		*nframes_out = 0;
	if (bytes_out)
		*bytes_out = 0;
	if (wide_samples == 0)
		return FLACAMD_OK;
	HIP_TRY(hipSetDevice(e->device));
	const FaEncParams &P = e->P;
	if (!e->stage_pcm) {
		const size_t in_bytes = sizeof(int32_t) * (size_t)e->max_frames * P.blocksize * P.channels;
		e->stage_out_bytes = (size_t)P.slot_bytes * e->max_frames;
		if (hipMalloc((void **)&e->stage_pcm, in_bytes) != hipSuccess || hipMalloc((void **)&e->stage_out, e->stage_out_bytes) != hipSuccess)
			return fa_fail(FLACAMD_E_NOMEM, "staging buffers: hipMalloc failed");
	}
	if (wide_samples > (uint64_t)e->max_frames * P.blocksize)
		return fa_fail(FLACAMD_E_INVALID, "batch larger than the encoder's capacity");
	// Streams of up to 16 bits cross PCIe as int16 (half the bytes), which also puts stereo on the
	// kernels' one-word-per-wide-sample paths; the results do not depend on the container width.
	uint32_t container = 4;
	if (P.bps <= 16) {
		const size_t n = (size_t)wide_samples * P.channels;
		if (!e->stage_host16 &&
		    hipHostMalloc((void **)&e->stage_host16, sizeof(int16_t) * (size_t)e->max_frames * P.blocksize * P.channels, hipHostMallocDefault) != hipSuccess)
			return fa_fail(FLACAMD_E_NOMEM, "pinned staging buffer: hipHostMalloc failed");
		// (values keep their low 16 bits, as a cast does; eight samples per step)
		size_t i = 0;
		for (; i + 8 <= n; i += 8) {
			__m128i a = _mm_loadu_si128((const __m128i *)(pcm + i)), b = _mm_loadu_si128((const __m128i *)(pcm + i + 4));
			a = _mm_srai_epi32(_mm_slli_epi32(a, 16), 16);
			b = _mm_srai_epi32(_mm_slli_epi32(b, 16), 16);
			_mm_storeu_si128((__m128i *)(e->stage_host16 + i), _mm_packs_epi32(a, b));
		}
		for (; i < n; i++)
			e->stage_host16[i] = (int16_t)pcm[i];
		HIP_TRY(hipMemcpyAsync(e->stage_pcm, e->stage_host16, sizeof(int16_t) * n, hipMemcpyHostToDevice, nullptr));
		container = 2;
	} else
		HIP_TRY(hipMemcpyAsync(e->stage_pcm, pcm, sizeof(int32_t) * wide_samples * P.channels, hipMemcpyHostToDevice, nullptr));
	uint32_t nf = 0;
	int rc = flacamd_encode_frames(e, e->stage_pcm, container, wide_samples, first_frame, is_last, loose_state, e->stage_out, e->stage_out_bytes,
	                               nullptr, nullptr, &nf, nullptr);
	if (rc)
		return rc;
	HIP_TRY(hipStreamSynchronize(nullptr));   // (this call's work is all on the null stream; other streams are not waited for)
	rc = flacamd_encoder_status(e);
	if (rc)
		return rc;
	uint64_t total = 0;
	HIP_TRY(hipMemcpy(frame_bytes, e->frame_bytes, sizeof(uint32_t) * nf, hipMemcpyDeviceToHost));
	HIP_TRY(hipMemcpy(&total, e->offsets + nf, sizeof total, hipMemcpyDeviceToHost));
	if (total > out_capacity)
		return fa_fail(FLACAMD_E_OVERFLOW, "output buffer too small: need %llu bytes", (unsigned long long)total);
	HIP_TRY(hipMemcpy(out, e->stage_out, total, hipMemcpyDeviceToHost));
	if (nframes_out)
		*nframes_out = nf;
	if (bytes_out)
		*bytes_out = total;
	return FLACAMD_OK;
}

// ---------------------------------------------------------------------------------------------
// The batch encode for callers that cannot allocate device memory (code under HIP graph capture,
// serving loops, harnesses that forbid hipMalloc): no handle, every intermediate buffer of the
// kernels carved out of one static device array per GPU.
// ---------------------------------------------------------------------------------------------
static __device__ __align__(256) unsigned char g_static_scratch[FLACAMD_STATIC_SCRATCH_BYTES];

#define STATIC_MAX_DEVICES 64
#define STATIC_MAX_WINDOW_FLOATS 131072
#include <mutex>
namespace {
struct StaticDevice {
	unsigned char *ptr;          // this device's copy of g_static_scratch, from hipGetSymbolAddress
	bool crc_ready;
	bool win_valid;              // the window tables at the start of the scratch match win_key
	flacamd_encode_params win_key;
};
StaticDevice g_static_dev[STATIC_MAX_DEVICES];
float g_static_win_host[STATIC_MAX_WINDOW_FLOATS];
std::mutex g_static_mutex;

struct StaticLayout {
	size_t windows, stats, autoc, cands, coefs, plans, assign, slots, frame_bytes, offsets, errflag, total;
};

StaticLayout static_layout(const FaEncParams &P, size_t F)
{
	StaticLayout L;
	size_t o = 0;
	auto take = [&](size_t bytes) {
		const size_t at = o;
		o = (o + bytes + 255) & ~(size_t)255;
		return at;
	};
	const size_t S = P.nsignals;
	// a fixed region whatever the call's settings: the tables are kept between calls, and a call with
	// fewer or no window samples must not lay its other buffers over the tail of a kept table
	L.windows = take(sizeof(float) * STATIC_MAX_WINDOW_FLOATS);
	L.stats = take(sizeof(FaSubStat) * F * S);
	L.autoc = take(sizeof(float) * F * S * P.napod * (FA_MAX_ORDER + 1));
	L.cands = take(sizeof(FaLpcCand) * F * S * P.ncand);
	L.coefs = take(sizeof(int32_t) * F * S * P.ncand * P.max_lpc_order + 64);
	L.plans = take(sizeof(FaPlan) * F * S);
	L.assign = take(sizeof(uint32_t) * (F + 2) + sizeof(uint64_t) * F * (1 + P.channels));
	L.slots = take((size_t)P.slot_bytes * F + 64);
	L.frame_bytes = take(sizeof(uint32_t) * F);
	L.offsets = take(sizeof(uint64_t) * (F + 1));
	L.errflag = take(16);
	L.total = o;
	return L;
}
}  // namespace

extern "C" int flacamd_encode_frames_static(const flacamd_encode_params *pin, int device, const void *pcm_dev, uint32_t sample_bytes,
                                            uint64_t wide_samples, uint32_t first_frame, int is_last, uint32_t *loose_state, void *out_dev,
                                            size_t out_capacity, uint32_t *frame_bytes_dev, uint64_t *frame_offsets_dev, uint32_t *status_dev,
                                            uint32_t *nframes_out, void *stream)
{
	if (!pin)
		return fa_fail(FLACAMD_E_INVALID, "flacamd_encode_frames_static: null settings");
	flacamd_encode_params p = *pin;
	int rc = flacamd_params_resolve(&p);
	if (rc)
		return rc;
	rc = check_partition_reach(p);
	if (rc)
		return rc;
	FaEncParams P;
	fill_kernel_params(p, P);
	FaBatch B;
	uint64_t nf = 0;
	rc = check_batch(P, pcm_dev, out_dev, sample_bytes, wide_samples, first_frame, is_last, loose_state, B, &nf);
	if (rc)
		return rc;
	if (nframes_out)
		*nframes_out = (uint32_t)nf;
	if (nf == 0)
		return FLACAMD_OK;
	const size_t win_floats = (size_t)P.napod * P.blocksize;
	if (P.max_lpc_order > 0 && win_floats > STATIC_MAX_WINDOW_FLOATS)
		return fa_fail(FLACAMD_E_UNSUPPORTED, "flacamd_encode_frames_static: %zu window samples (apodizations x blocksize) exceed the limit %d", win_floats,
		               STATIC_MAX_WINDOW_FLOATS);
	const StaticLayout L = static_layout(P, (size_t)nf);
	if (L.total > FLACAMD_STATIC_SCRATCH_BYTES)
		return fa_fail(FLACAMD_E_UNSUPPORTED,
		               "flacamd_encode_frames_static: a batch of %llu frames with these settings needs %zu bytes of scratch, the static buffer holds %u",
		               (unsigned long long)nf, L.total, (unsigned)FLACAMD_STATIC_SCRATCH_BYTES);
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
		return fa_fail(FLACAMD_E_DEVICE, "no HIP device available: the FLAC engine needs an AMD GPU");
	if (device < 0 || device >= ndev || device >= STATIC_MAX_DEVICES)
		return fa_fail(FLACAMD_E_DEVICE, "HIP device %d out of range (%d present)", device, ndev);
	HIP_TRY(hipSetDevice(device));
	hipStream_t st = (hipStream_t)stream;

	std::lock_guard<std::mutex> guard(g_static_mutex);
	StaticDevice &D = g_static_dev[device];
	if (!D.ptr) {
		// each GPU holds its own copy of the array at its own address; the array's name is only a
		// handle for this lookup, never a pointer host code may pass on
		void *sym = nullptr;
		HIP_TRY(hipGetSymbolAddress(&sym, HIP_SYMBOL(g_static_scratch)));
		if (!sym)
			return fa_fail(FLACAMD_E_DEVICE, "hipGetSymbolAddress returned no address for the static scratch buffer");
		D.ptr = (unsigned char *)sym;
	}
	unsigned char *base = D.ptr;
	if (!D.crc_ready) {
		uint16_t t16[256], pow16[32];
		uint8_t t8[256];
		make_crc_tables(t16, pow16, t8);
		HIP_TRY(fa_upload_crc_tables(t16, pow16, t8));
		D.crc_ready = true;
	}
	if (P.max_lpc_order > 0) {
		// the window tables stay in the scratch between calls; they are rebuilt (on the host: the
		// reference's cosines come from the C library) when blocksize or apodizations change
		bool same = D.win_valid && D.win_key.blocksize == p.blocksize && D.win_key.num_apodizations == p.num_apodizations;
		for (uint32_t a = 0; same && a < p.num_apodizations; a++)
			same = D.win_key.apod_type[a] == p.apod_type[a] && memcmp(&D.win_key.apod_param[a], &p.apod_param[a], sizeof(float)) == 0;
		if (!same) {
			D.win_valid = false;
			for (uint32_t a = 0; a < P.napod; a++) {
				rc = flacamd_window(p.apod_type[a], p.apod_param[a], P.blocksize, g_static_win_host + (size_t)a * P.blocksize);
				if (rc)
					return rc;
			}
			// an earlier call on another stream may still be reading the old tables
			HIP_TRY(hipDeviceSynchronize());
			HIP_TRY(hipMemcpy(base + L.windows, g_static_win_host, sizeof(float) * win_floats, hipMemcpyHostToDevice));
			D.win_key = p;
			D.win_valid = true;
		}
	}
	EncWork W;
	W.planes = nullptr;   // (planes of a whole batch would not fit the static buffer)
	W.stat_acc = nullptr;
	W.windows = (const float *)(base + L.windows);
	W.stats = (FaSubStat *)(base + L.stats);
	W.autoc = (float *)(base + L.autoc);
	W.cands = (FaLpcCand *)(base + L.cands);
	W.coefs = (int32_t *)(base + L.coefs);
	W.plans = (FaPlan *)(base + L.plans);
	W.assign = (uint32_t *)(base + L.assign);
	W.slots = (uint32_t *)(base + L.slots);
	W.errflag = (uint32_t *)(base + L.errflag);
	// the scratch keeps the last call's contents: the error bits must start at zero (every other
	// buffer is written before it is read)
	HIP_TRY(hipMemsetAsync(W.errflag, 0, 16, st));
	rc = run_batch(P, B, W, pcm_dev, out_dev, out_capacity, frame_bytes_dev ? frame_bytes_dev : (uint32_t *)(base + L.frame_bytes),
	               frame_offsets_dev ? frame_offsets_dev : (uint64_t *)(base + L.offsets), st, nullptr);
	if (rc)
		return rc;
	if (status_dev)
		HIP_TRY(hipMemcpyAsync(status_dev, W.errflag, sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
	if (loose_state && P.loose_ms) {
		HIP_TRY(hipMemcpyAsync(loose_state, W.errflag + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
	}
	return FLACAMD_OK;
}

// ---------------------------------------------------------------------------------------------
// test / debug access to intermediate device results (not part of the public header)
// ---------------------------------------------------------------------------------------------
extern "C" int flacamd_debug_read(flacamd_encoder *e, int what, void *dst, size_t bytes)
{
	if (!e || !dst)
		return fa_fail(FLACAMD_E_INVALID, "bad arguments");
	HIP_TRY(hipSetDevice(e->device));
	const void *src = nullptr;
	switch (what) {
	case 0: src = e->stats; break;
	case 1: src = e->autoc; break;
	case 2: src = e->cands; break;
	case 3: src = e->coefs; break;
	case 4: src = e->plans; break;
	case 5: src = e->assign; break;
	case 6: src = e->frame_bytes; break;
	default: return fa_fail(FLACAMD_E_INVALID, "unknown debug buffer %d", what);
	}
	HIP_TRY(hipDeviceSynchronize());
	HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
	return FLACAMD_OK;
}

extern "C" size_t flacamd_debug_sizeof(int what)
{
	switch (what) {
	case 0: return sizeof(FaSubStat);
	case 2: return sizeof(FaLpcCand);
	case 4: return sizeof(FaPlan);
	default: return 0;
	}
}
