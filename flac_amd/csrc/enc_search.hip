// FLAC frame encoder kernels for gfx950 (MI355X), part 2: the model search.
#include <hip/hip_runtime.h>

#include <stdlib.h>
#include <string.h>

#include "enc_common.h"
#include "enc_kernels.h"

// ---------------------------------------------------------------------------------------------
// k_search: process_subframe_ (stream_encoder.c:3164) for one (frame, signal).
//
// Every candidate predictor (the fixed ones are coefficient rows like the LPC ones) is run over
// the frame with the samples in registers; only the |residual| sums of the finest Rice partitions
// go to LDS.  find_best_partition_order_ / set_partitioned_rice_ (stream_encoder.c:3666,3943) then
// work from those sums: the sum of a coarser partition is the sum of the finest ones it covers.
//
// A workgroup's life is a chain of dependent steps, so everything it needs from global memory (the
// frame's samples, the candidate list, the coefficients) is requested at entry in one go.
// ---------------------------------------------------------------------------------------------
#define SR_MAXGROUP 32
#define SR_MAXSUM_BYTES 32768
#define SR_PRE_CANDS 256   // candidate slots fetched at entry
#define SR_PRE_COEFS 1024  // coefficients fetched at entry

struct SrCand {
	uint32_t kind;      // FA_SUB_FIXED / FA_SUB_LPC
	uint32_t order;
	uint32_t seed;      // Rice parameter used at partition order 0
	uint32_t slot;      // LPC candidate slot
	uint32_t precision;
	int32_t  shift;
	uint32_t mode;
	uint32_t pmax, pmin, psize, use32;
	uint32_t fast;      // every |residual| of this candidate is below 2^27
	uint32_t hdr_bits;  // subframe bits besides the residual
	uint32_t sumq;      // an upper bound of the sum of the coefficients' magnitudes (exact for the fixed predictors)
	uint32_t best_po, best_rbits, total_bits;
};

// Add |res[i]| of the chunk's samples that lie in [order, bsf) to the candidate's finest-partition
// sums (the lowest level of precompute_partition_info_sums_, stream_encoder.c:3774).
__device__ __forceinline__ void accumulate_chunk(const int32_t (&res)[FA_CHUNK], uint64_t *gs, uint32_t base, uint32_t order, uint32_t bsf,
                                                 uint32_t psize)
{
	uint32_t part = base / psize;
	uint32_t next = (part + 1) * psize;
	uint64_t t = 0;
#pragma unroll
	for (int i = 0; i < FA_CHUNK; i++) {
		const uint32_t idx = base + i;
		if (idx == next) {
			if (t) atomicAdd((unsigned long long *)&gs[part], (unsigned long long)t);
			t = 0;
			part++;
			next += psize;
		}
		if (idx >= order && idx < bsf)
			t += abs32(res[i]);
	}
	if (t) atomicAdd((unsigned long long *)&gs[part], (unsigned long long)t);
}

// The same from the biased samples db[i] = sample ^ 2^31 and the predictions: the biased residual
// (sample ^ 2^31) - prediction = residual ^ 2^31 feeds v_sad_u32 directly.  quick: the chunk lies
// inside the frame and inside one partition and its sum fits 32 bits; warm-up samples (index <
// order, the first thread or two) are taken back out afterwards.
__device__ __forceinline__ void accumulate_biased(const uint32_t (&db)[FA_CHUNK], const int32_t (&pred)[FA_CHUNK], uint64_t *gs, uint32_t base,
                                                  uint32_t order, uint32_t bsf, uint32_t psize, bool quick, uint32_t bias, uint32_t quick_part)
{
	if (quick) {
		uint32_t t32 = 0;
#pragma unroll
		for (int i = 0; i < FA_CHUNK; i++)
			t32 = sad_biased(db[i] - (uint32_t)pred[i], bias, t32);
		if (base < order) {
#pragma unroll
			for (int i = 0; i < FA_CHUNK; i++)
				if (base + i < order)
					t32 -= abs32((int32_t)((db[i] - (uint32_t)pred[i]) ^ bias));
		}
		atomicAdd((unsigned long long *)&gs[quick_part], (unsigned long long)t32);
	} else {
		int32_t res[FA_CHUNK];
#pragma unroll
		for (int i = 0; i < FA_CHUNK; i++)
			res[i] = (int32_t)((db[i] - (uint32_t)pred[i]) ^ bias);
		accumulate_chunk(res, gs, base, order, bsf, psize);
	}
}

// The same from x[i] = B - residual (lpc_negres16_*): |residual| = |x - B| is one v_sad_u32.  The
// rare paths take the chunk's position through an opaque copy, so that nothing derived from it is
// computed ahead of the candidate loop and kept in registers across it.
__device__ __forceinline__ void accumulate_negres(const uint32_t (&x)[FA_CHUNK], uint32_t B, uint64_t *gs, uint32_t base, uint32_t order,
                                                  uint32_t bsf, uint32_t psize, bool quick, uint32_t quick_part)
{
	if (quick) {
		uint32_t t32 = 0;
#pragma unroll
		for (int i = 0; i < FA_CHUNK; i++)
			t32 = sad_u32(x[i], B, t32);
		if (base < order) {
			uint32_t b2 = base;
			asm volatile("" : "+v"(b2));
#pragma unroll
			for (int i = 0; i < FA_CHUNK; i++)
				if (b2 + i < order)
					t32 -= abs32((int32_t)(B - x[i]));
		}
		atomicAdd((unsigned long long *)&gs[quick_part], (unsigned long long)t32);
	} else {
		uint32_t b2 = base;
		asm volatile("" : "+v"(b2));
		int32_t res[FA_CHUNK];
#pragma unroll
		for (int i = 0; i < FA_CHUNK; i++)
			res[i] = (int32_t)(B - x[i]);
		accumulate_chunk(res, gs, b2, order, bsf, psize);
	}
}

// LDS carve-up, computed the same way on the host (for the size) and in the kernel.
struct SrLayout {
	uint32_t sums, stage, pbits, gc, best, ctl, cq, cqp, gp, lcand, allq, total;
};

__host__ __device__ inline SrLayout sr_layout(uint32_t sum_bytes, uint32_t stage_bytes, uint32_t alias, uint32_t ncmax, uint32_t qstride,
                                               uint32_t npre_cands, uint32_t npre_coefs)
{
	SrLayout L;
	uint32_t o = 0;
	L.sums = o;
	if (alias) {
		L.stage = o;
		o += sum_bytes > stage_bytes ? sum_bytes : stage_bytes;
	} else {
		o += sum_bytes;
		L.stage = o;
		o += stage_bytes;
	}
	o = (o + 15) & ~15u;
	L.pbits = o; o += ncmax * 16 * 4;
	L.gc = o; o += ncmax * (uint32_t)sizeof(SrCand);
	L.best = o; o += (uint32_t)sizeof(FaPlan);
	L.ctl = o; o += 16 * 4;
	L.cq = o; o += ncmax * qstride * 4;
	o = (o + 15) & ~15u;
	L.cqp = o; o += ncmax * (qstride / 2 + 1) * 4;
	o = (o + 15) & ~15u;
	L.gp = o; o += ncmax * 8 * 4;
	L.lcand = o; o += npre_cands * (uint32_t)sizeof(FaLpcCand);
	L.allq = o; o += npre_coefs * 4;
	L.total = (o + 15) & ~15u;
	return L;
}

#ifndef SEARCH_WAVES
#define SEARCH_WAVES 4
#endif
// Wave priorities: the phases either side of the residual loop are chains of dependent loads, LDS round trips and
// barriers with few instructions; the residual loop is issue-heavy.  With four workgroups sharing each SIMD, letting the
// short phases issue ahead of the other workgroups' residual loops shortens every workgroup's critical path and costs the
// loops nothing they would not have waited for anyway (measured: 0.828 -> 0.783 ms; the reverse assignment gains nothing).
#ifndef SEARCH_PRIO_EDGE
#define SEARCH_PRIO_EDGE 3
#define SEARCH_PRIO_LOOP 0
#endif
#define SPRIO(n) __builtin_amdgcn_s_setprio(n)

// PAIRS: the PCM is 16-bit stereo read as one 32-bit word per wide sample (a compile-time fact
// here, so that the other sample formats' code does not weigh on register allocation)
// RES: which residual arithmetic is compiled in.  SR_RES_DOT2: every signal this launch covers fits
// 16 bits, two taps per v_dot2_i32_i16.  SR_RES_TAPS: samples of up to 24 bits and 32-bit sums only
// (NARROW streams), a tap per v_mad_i32_i24.  SR_RES_SPLIT: anything wider, samples split into two
// 16-bit halves, two chains of v_dot2_i32_i16.  The two dot2 forms carry a slow exact path for the
// candidates whose 64-bit sum they cannot guarantee.  A launch covers the signals listed in sig_list
// (four bits each, nsel of them) of every frame.
#define SR_RES_TAPS 0
#define SR_RES_DOT2 1
#define SR_RES_SPLIT 2
// Waves per SIMD = workgroups per CU, which sets the register budget (512 / waves).  The split form
// keeps both halves of the sample window in registers: 168 registers at order <= 12, 256 beyond.
#define SR_WAVES(H, R) ((R) == SR_RES_SPLIT ? ((H) > 12 ? 2 : 3) : SEARCH_WAVES)
template <int HIST, bool NARROW, bool PAIRS, int RES>
__global__ __launch_bounds__(FA_THREADS) __attribute__((amdgpu_waves_per_eu(SR_WAVES(HIST, RES), SR_WAVES(HIST, RES))))
void k_search(FaEncParams P, FaBatch B, const void *pcm, const FaSubStat *stats, const FaLpcCand *cands, const int32_t *coefs, FaPlan *plans,
              uint32_t sum_bytes, uint32_t alias, uint32_t ncmax, unsigned long long *stamps, unsigned long long sig_list, uint32_t nsel, uint32_t frame0)
{
// debugging aid: the first time thread 0 passes point k, the shader clock goes to stamps[block][k]
#define STAMP(k) do { if (stamps && threadIdx.x == 0 && first_pass) stamps[(size_t)task * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
	bool first_pass = true;
	constexpr int MAXORDER = HIST < FA_MAX_ORDER ? HIST : FA_MAX_ORDER;   // largest predictor order this kernel runs
	constexpr int NQP = (MAXORDER + 1) / 2;   // words of tap pairs (split form)
	constexpr int NQF = MAXORDER / 2 + 1;     // words of the folded form: the sample's own tap comes first
	constexpr int NQW = RES == SR_RES_DOT2 ? NQF : NQP;
	constexpr uint32_t NIT = (ST_HP(HIST) + FA_TILE + FA_THREADS - 1) / FA_THREADS;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const uint32_t ncoef_all = P.ncand * P.max_lpc_order;
	const uint32_t npre_cands = umin32(P.ncand, SR_PRE_CANDS), npre_coefs = umin32(ncoef_all, SR_PRE_COEFS);
	const SrLayout L = sr_layout(sum_bytes, ST_WORDS(HIST) * 4, alias, ncmax, MAXORDER, npre_cands, npre_coefs);
	uint64_t *sums = (uint64_t *)(smem + L.sums);       // [group][2 PF] partition sums: finest level first, then each coarser one
	int32_t *stage = (int32_t *)(smem + L.stage);       // the staged signal tile
	uint32_t *pbits = (uint32_t *)(smem + L.pbits);     // [group][16] residual bits per partition order
	SrCand *gc = (SrCand *)(smem + L.gc);               // [group]
	FaPlan *best = (FaPlan *)(smem + L.best);
	uint32_t *ctl = (uint32_t *)(smem + L.ctl);         // [0]=group size [1]=cursor [2]=winner [3]=rice2 flag
	int32_t *cq = (int32_t *)(smem + L.cq);             // [group][MAXORDER] quantised coefficients
	int32_t *cqp = (int32_t *)(smem + L.cqp);           // [group][NQW] the same, two 16-bit taps per word (dot2: folded form, see lpc_negres16_dot2)
	uint32_t *gp = (uint32_t *)(smem + L.gp);           // [group][4] order, shift, psize, fast | mode << 1 | slow << 3 | shift << 8 | (frame's - candidate's largest partition order) << 16
	uint32_t *gq = gp + ncmax * 4;                      // [group][4] pmax | pmin << 8 | use32 << 16, order, seed
	FaLpcCand *lcand = (FaLpcCand *)(smem + L.lcand);   // candidate slots fetched at entry
	int32_t *allq = (int32_t *)(smem + L.allq);         // coefficients fetched at entry

	// Workgroups go to the chip's eight dies in turn, each with an L2 of its own: the workgroups of one
	// frame's signals, which read the same samples, are given indices that are equal modulo eight.
	// (frames below frame0 are not this launch's)
	const uint32_t f = frame0 + (blockIdx.x / (8 * nsel)) * 8 + (blockIdx.x & 7u);
	if (f >= B.nframes)
		return;
	const uint32_t s = (uint32_t)(sig_list >> (4 * ((blockIdx.x % (8 * nsel)) >> 3))) & 15u;
	const uint32_t task = f * P.nsignals + s;
	const uint32_t tid = threadIdx.x;
	const uint32_t bsf = frame_len(B, P.blocksize, f);
	const uint64_t frame_start = (uint64_t)f * P.blocksize;
	const FaLpcCand *my_cands = cands + (size_t)task * P.ncand;
	const int32_t *my_coefs = coefs + (size_t)task * ncoef_all;

	STAMP(0);
	SPRIO(SEARCH_PRIO_EDGE);
	// ---- everything this workgroup reads from global memory, requested up front ----
	FaLpcCand pre_c;
	pre_c.valid = 0;
	if (tid < npre_cands)
		pre_c = my_cands[tid];
	int32_t pre_q[SR_PRE_COEFS / FA_THREADS];
#pragma unroll
	for (uint32_t k = 0; k < SR_PRE_COEFS / FA_THREADS; k++)
		pre_q[k] = (tid + k * FA_THREADS < npre_coefs) ? my_coefs[tid + k * FA_THREADS] : 0;
	int32_t tile_v[NIT];
	stage_load<HIST>(tile_v, pcm, P.channels, B.sample_bytes, PAIRS ? 1u : B.planar, frame_start, bsf, s, 0);
	const FaSubStat st = stats[task];
	const uint32_t sbps = st.bps, wasted = st.wasted;
	const uint32_t hdr = 8 + wasted;
	const bool fractional = bsf != P.blocksize;
	// process_subframes_ (stream_encoder.c:2929-2936)
	uint32_t frame_max_po = 0;
	if (!fractional) {
		uint32_t o = 0, b = bsf;
		while (!(b & 1)) { o++; b >>= 1; }
		frame_max_po = umin32(umin32(o, 15u), P.max_po);
	}
	const uint32_t frame_min_po = umin32(P.min_po, frame_max_po);
	const uint32_t PF = 1u << frame_max_po;
	const uint32_t group_cap = umin32(ncmax, sum_bytes / (2 * PF * 8));
	const uint32_t ntiles = (bsf + FA_TILE - 1) / FA_TILE;

	if (tid == 0) {
		best->type = FA_SUB_VERBATIM;
		best->order = 0;
		best->precision = 0;
		best->shift = 0;
		best->part_order = 0;
		best->rice2 = 0;
		best->wide = 0;
		best->pad = 0;
		best->constant = 0;
		// evaluate_verbatim_subframe_ (stream_encoder.c:3641)
		best->bits = (P.disable_verbatim && bsf >= 4) ? 0xffffffffu : hdr + bsf * sbps;
		if (bsf >= 4 && st.is_constant) {
			// evaluate_constant_subframe_ (stream_encoder.c:3470)
			const uint32_t cb = hdr + sbps;
			if (cb < best->bits) {
				best->type = FA_SUB_CONSTANT;
				best->bits = cb;
				best->constant = st.first_sample;
			}
		}
		ctl[1] = 0;
	}
	stage_store<HIST>(stage, tile_v, bsf, wasted, 0);
	if (tid < npre_cands)
		lcand[tid] = pre_c;
#pragma unroll
	for (uint32_t k = 0; k < SR_PRE_COEFS / FA_THREADS; k++)
		if (tid + k * FA_THREADS < npre_coefs)
			allq[tid + k * FA_THREADS] = pre_q[k];
	__syncthreads();
	STAMP(1);

	const bool search = bsf >= 4 && !st.is_constant;
	const uint32_t total_cands = 5 + P.ncand;
	bool staged = true;   // the stage area holds tile 0
	while (search) {
		// ---- wave 0 gathers the next group of candidates, in the reference's trial order: lane i
		// looks at candidate cursor + i, the takers are packed in order ----
		if (tid < 64) {
			uint32_t n = 0, cur = ctl[1];
			const uint32_t best_bits = best->bits;
			while (cur < total_cands && n < group_cap) {
				const uint32_t my = cur + tid;
				SrCand c;
				c.kind = FA_SUB_FIXED;
				c.order = 0;
				c.seed = 0;
				c.slot = 0;
				c.precision = 0;
				c.shift = 0;
				c.mode = (uint32_t)lpc_mode(0, sbps);
				c.hdr_bits = 0;
				c.sumq = 0;
				bool take = false;
				if (my < 5) {
					// fixed predictors (stream_encoder.c:3236-3287)
					const uint32_t order = my;
					uint32_t lo, hi;
					if (P.exhaustive) { lo = 0; hi = 4; }
					else lo = hi = st.fixed_guess;
					if (hi >= bsf) hi = bsf - 1;
					const bool enabled = !P.disable_fixed || (P.max_lpc_order == 0 && best_bits == 0xffffffffu);
					const float rb = order == 0 ? st.rbps[0] : order == 1 ? st.rbps[1] : order == 2 ? st.rbps[2] : order == 3 ? st.rbps[3] : st.rbps[4];
					if (enabled && order >= lo && order <= hi && !(rb >= (float)sbps)) {
						uint32_t rice = (rb > 0.0f) ? (uint32_t)((double)rb + 0.5) : 0u;
						rice++;
						if (rice >= P.rice_limit) rice = P.rice_limit - 1;
						c.order = order;
						c.seed = rice;
						c.hdr_bits = hdr + order * sbps;
						c.sumq = (1u << order) - 1u;   // 0, 1, 2+1, 3+3+1, 4+6+4+1
						take = true;
					}
				} else if (my < total_cands) {
					const uint32_t slot = my - 5;
					FaLpcCand lc;
					if (slot < npre_cands)
						lc = lcand[slot];
					else
						lc = my_cands[slot];
					if (lc.valid) {
						c.kind = FA_SUB_LPC;
						c.order = lc.order;
						c.seed = lc.rice;
						c.slot = slot;
						c.precision = lc.precision;
						c.shift = lc.shift;
						c.mode = (uint32_t)lpc_mode(lc.wide, sbps);
						c.hdr_bits = hdr + 4 + 5 + lc.order * (lc.precision + sbps);
						c.sumq = (uint32_t)lc.sumq32 << 5;
						take = true;
					}
				}
				c.pmax = limit_part_order(frame_max_po, bsf, c.order);
				c.pmin = umin32(frame_min_po, c.pmax);
				c.psize = bsf >> c.pmax;
				c.use32 = (ilog2u(c.psize) + sbps < 32) ? 1u : 0u;  // stream_encoder.c:3803
				c.best_po = 0;
				c.best_rbits = 0;
				c.total_bits = 0;
				c.fast = 0;
				const uint64_t takers = __ballot(take);
				const uint32_t before = (uint32_t)__popcll(takers & ((1ull << tid) - 1ull));
				const uint32_t room = group_cap - n;
				if (take && before < room)
					gc[n + before] = c;
				const uint32_t cnt = (uint32_t)__popcll(takers);
				if (cnt <= room) {
					n += cnt;
					cur += 64;
				} else {
					// the group is full: resume right after the last candidate that got in
					const uint64_t last = __ballot(take && before == room - 1);
					cur += (uint32_t)__ffsll((unsigned long long)last);
					n += room;
				}
			}
			if (tid == 0) {
				ctl[0] = n;
				ctl[1] = cur;
			}
		}
		__syncthreads();
		STAMP(2);
		const uint32_t ng = ctl[0];
		if (ng == 0)
			break;
		// ---- per-candidate tables: coefficients (plain and packed in pairs), loop parameters ----
		// coefficient j of group member g; a fixed predictor of order o is the filter {1}, {2,-1},
		// {3,-3,1}, {4,-6,4,-1} with shift 0 (fixed.c:352-388) and runs through the same code
		auto coef = [&](uint32_t g, uint32_t j) -> int32_t {
			const uint32_t o = gc[g].order;
			if (j >= o)
				return 0;
			if (gc[g].kind == FA_SUB_LPC) {
				const uint32_t at = gc[g].slot * P.max_lpc_order + j;
				return at < npre_coefs ? allq[at] : my_coefs[at];
			}
			return j == 0 ? (int32_t)o : j == 1 ? (o == 2 ? -1 : o == 3 ? -3 : -6) : j == 2 ? (o == 3 ? 1 : 4) : -1;
		};
		for (uint32_t e = tid; e < ng * MAXORDER; e += FA_THREADS)
			cq[e] = coef(e / MAXORDER, e % MAXORDER);
		for (uint32_t e = tid; e < ng * NQW; e += FA_THREADS) {
			const uint32_t g = e / NQW, jj = e % NQW;
			uint32_t q0, q1;
			if (RES == SR_RES_DOT2) {
				// c[0] = -2^shift, c[m] = q[m - 1]
				q0 = jj ? (uint32_t)coef(g, 2 * jj - 1) : 0u - (1u << (gc[g].shift & 15));
				q1 = (uint32_t)coef(g, 2 * jj);
			} else {
				q0 = (uint32_t)coef(g, 2 * jj);
				q1 = (uint32_t)coef(g, 2 * jj + 1);
			}
			cqp[e] = (int32_t)((q0 << 16) | (q1 & 0xffffu));
		}
		if (PF > 64)
			for (uint32_t e = tid; e < ng * 16; e += FA_THREADS)
				pbits[e] = 0;
		if (tid >= 64 && tid - 64 < ng) {
			const uint32_t g = tid - 64;
			SrCand &c = gc[g];
			// |prediction| <= (sum |q| * max |sample|) >> shift, |residual| <= |sample| + |prediction|
			const uint64_t sumq = c.sumq;
			const uint64_t bound = (1ull << (sbps - 1)) + ((sumq << (sbps - 1)) >> c.shift) + 1;
			// (a sum that wraps in 32 bits stays below 2^(31 - shift), which the bound covers too)
			const uint32_t fast = ((RES != SR_RES_TAPS || c.mode == 0) && bound < (1ull << 27)) ? 1u : 0u;
			c.fast = fast;
			uint32_t slow = 0, sh = (uint32_t)c.shift;
			if (RES != SR_RES_SPLIT) {
				// the folded chain is exact while sum - sample * 2^shift stays inside 32 bits
				slow = fa_fold_exact(sumq, (uint32_t)c.shift, sbps) ? 0u : 1u;
			} else {
				const uint32_t shf = (uint32_t)c.shift;
				if (c.mode != 2)
					sh = 12u | (shf << 16);
				else if (shf <= 12)
					sh = (12u - shf) | (shf << 8);
				else
					sh = (12u << 8) | ((shf - 12u) << 16);
				slow = (c.mode == 2 && !fa_split_exact(sumq, sbps)) ? 1u : 0u;
			}
			gp[4 * g] = c.order;
			gp[4 * g + 1] = sh;
			gp[4 * g + 2] = c.psize;
			gp[4 * g + 3] = fast | (c.mode << 1) | (slow << 3) | ((uint32_t)c.shift << 8) | ((frame_max_po - c.pmax) << 16);
			gq[4 * g] = c.pmax | (c.pmin << 8) | (c.use32 << 16);
			gq[4 * g + 1] = c.order;
			gq[4 * g + 2] = c.seed;
			gq[4 * g + 3] = 0;
		}
		// Clearing the sums before the residuals add into them.  With at most 64 finest partitions the
		// coarser levels are written, not added to, so only each candidate's finest sums need it.
		auto clear_sums = [&]() {
			if (PF >= 2 && PF <= 64) {
				const uint32_t half = PF / 2;   // 16-byte stores: two sums each
				for (uint32_t e = tid; e < ng * half; e += FA_THREADS)
					((uint4 *)sums)[(e / half) * PF + (e % half)] = make_uint4(0, 0, 0, 0);
			} else {
				for (uint32_t e = tid; e < ng * 2 * PF; e += FA_THREADS)
					sums[e] = 0;
			}
		};
		if (!alias)
			clear_sums();
		STAMP(3);

		// ---- residuals of every candidate of the group, tile by tile, all in registers ----
		for (uint32_t tile = 0; tile < ntiles; tile++) {
			if (!(tile == 0 && staged)) {
				__syncthreads();
				stage_tile<HIST>(stage, pcm, P.channels, B.sample_bytes, PAIRS ? 1u : B.planar, frame_start, bsf, s, wasted, tile * FA_TILE, opaque_tid());
			}
			__syncthreads();
			staged = false;
			const uint32_t base = tile * FA_TILE + tid * FA_CHUNK;
			const bool have = base < bsf;
			int32_t d[HIST + FA_CHUNK];
			if (have)
				load_chunk_lds<HIST>(d, stage);
			if (alias) {
				// the sums take over the stage area once every thread holds its samples
				__syncthreads();
				clear_sums();
				__syncthreads();
			}
			STAMP(4);
			if (!have)
				continue;
			SPRIO(SEARCH_PRIO_LOOP);
			const uint32_t bias = 0x80000000u;
			uint32_t db[FA_CHUNK];
#pragma unroll
			for (int i = 0; i < FA_CHUNK; i++)
				db[i] = (uint32_t)d[HIST + i] ^ bias;
			const bool chunk_inside = base + FA_CHUNK <= bsf;
			// the chunk's finest partition, recomputed only when a candidate's partition size differs
			// from the one before (the size is the same in every lane)
			uint32_t part_psize = 0, part_of_chunk = 0;
			// the next candidate's parameters and coefficients are fetched from LDS while the current
			// one is computed; at the loop top they move to scalar registers
			// the chunk's finest partition at the frame's largest partition order; a candidate whose own
			// largest order is lower shifts it down (2^order divides the frame length)
			const uint32_t part_finest = base / (bsf >> frame_max_po);
			if (RES == SR_RES_DOT2) {
				// samples and coefficients fit 16 bits: two taps per instruction, residual folded into the chain
				int32_t PW[HIST + FA_CHUNK];
				PW[0] = (int32_t)((uint32_t)d[0] << 16);
#pragma unroll
				for (int k = 1; k < HIST + FA_CHUNK; k++)
					PW[k] = (int32_t)__builtin_amdgcn_perm((uint32_t)d[k], (uint32_t)d[k - 1], 0x05040100u);
				const uint32_t c31 = bias;
				uint32_t nxp[4];
				int32_t nxq[NQF];
#pragma unroll
				for (int k = 0; k < 4; k++)
					nxp[k] = gp[k];
#pragma unroll
				for (int k = 0; k < NQF; k++)
					nxq[k] = cqp[k];
#pragma nounroll
				for (uint32_t g = 0; g < ng; g++) {
					const uint32_t order = __builtin_amdgcn_readfirstlane(nxp[0]);
					const uint32_t shift = __builtin_amdgcn_readfirstlane(nxp[1]);
					const uint32_t psize = __builtin_amdgcn_readfirstlane(nxp[2]);
					const uint32_t flags = __builtin_amdgcn_readfirstlane(nxp[3]);
					const uint32_t fast = flags & 1u;
					int32_t qs[NQF];
#pragma unroll
					for (int k = 0; k < NQF; k++)
						qs[k] = (int32_t)__builtin_amdgcn_readfirstlane((uint32_t)nxq[k]);
					if (g + 1 < ng) {
#pragma unroll
						for (int k = 0; k < 4; k++)
							nxp[k] = gp[4 * (g + 1) + k];
#pragma unroll
						for (int k = 0; k < NQF; k++)
							nxq[k] = cqp[(g + 1) * NQF + k];
					}
					uint32_t x[FA_CHUNK], B;
					if (flags & 8u) {
						// exact 64-bit taps; sample k of the window is the high half of PW[k], coefficient j is c[j + 1]
						B = bias;
						lpc_negres16_exact64<HIST, MAXORDER>([&](int k) -> int32_t { return PW[k] >> 16; },
						                                     [&](int j) -> int32_t { return packed_coef(qs, j + 1); }, order, (int)shift,
						                                     ((flags >> 1) & 3u) != 2u, x);
					} else {
						lpc_negres16_dot2<HIST, MAXORDER>(PW, qs, order, shift, c31, x);
						B = bias >> shift;
					}
					accumulate_negres(x, B, sums + (size_t)g * 2 * PF, base, order, bsf, psize, fast && chunk_inside && (psize & (FA_CHUNK - 1)) == 0,
					                  part_finest >> ((flags >> 16) & 15u));
				}
			} else if (RES == SR_RES_SPLIT) {
				// wider samples in two 16-bit halves: two chains of two taps per instruction
				int32_t Ph[HIST + FA_CHUNK - 1], Pl[HIST + FA_CHUNK - 1];
#pragma unroll
				for (int k = 0; k < HIST + FA_CHUNK - 1; k++) {
					Ph[k] = (int32_t)__builtin_amdgcn_perm((uint32_t)(d[k + 1] >> 12), (uint32_t)(d[k] >> 12), 0x05040100u);
					Pl[k] = (int32_t)(((uint32_t)d[k] & 0xfffu) | (((uint32_t)d[k + 1] & 0xfffu) << 16));
				}
				uint32_t nxp[4];
				int32_t nxq[NQP];
#pragma unroll
				for (int k = 0; k < 4; k++)
					nxp[k] = gp[k];
#pragma unroll
				for (int k = 0; k < NQP; k++)
					nxq[k] = cqp[k];
#pragma nounroll
				for (uint32_t g = 0; g < ng; g++) {
					const uint32_t order = __builtin_amdgcn_readfirstlane(nxp[0]);
					const uint32_t shc = __builtin_amdgcn_readfirstlane(nxp[1]);
					const uint32_t psize = __builtin_amdgcn_readfirstlane(nxp[2]);
					const uint32_t flags = __builtin_amdgcn_readfirstlane(nxp[3]);
					const uint32_t fast = flags & 1u;
					int32_t qs[NQP];
#pragma unroll
					for (int k = 0; k < NQP; k++)
						qs[k] = (int32_t)__builtin_amdgcn_readfirstlane((uint32_t)nxq[k]);
					if (g + 1 < ng) {
#pragma unroll
						for (int k = 0; k < 4; k++)
							nxp[k] = gp[4 * (g + 1) + k];
#pragma unroll
						for (int k = 0; k < NQP; k++)
							nxq[k] = cqp[(g + 1) * NQP + k];
					}
					int32_t pred[FA_CHUNK];
					if (flags & 8u) {
						// sample k of the window, put together again from its halves
						// (the taps of the chunk's samples reach window samples 0 .. HIST + 14 only)
						auto getd = [&](int k) -> int32_t {
							return (int32_t)(((uint32_t)(int32_t)(int16_t)((uint32_t)Ph[k] & 0xffff) << 12) | ((uint32_t)Pl[k] & 0xfffu));
						};
						lpc_pred16_exact64<HIST, MAXORDER>(getd, [&](int j) -> int32_t { return packed_coef(qs, j); }, order, (int)((flags >> 8) & 0xff), false, pred);
					} else
						lpc_pred16_split<HIST, MAXORDER>(Ph, Pl, qs, order, shc & 0xff, (shc >> 8) & 0xff, shc >> 16, pred);
					if (psize != part_psize) {
						part_psize = psize;
						part_of_chunk = base / psize;
					}
					accumulate_biased(db, pred, sums + (size_t)g * 2 * PF, base, order, bsf, psize,
					                  fast && chunk_inside && (psize & (FA_CHUNK - 1)) == 0, bias, part_of_chunk);
				}
			} else {
				static_assert(RES != SR_RES_TAPS || NARROW, "the tap-at-a-time form is compiled for 32-bit sums only");
				const uint32_t c31 = bias;
				uint32_t nxp[4];
				int32_t nxq[MAXORDER];
#pragma unroll
				for (int k = 0; k < 4; k++)
					nxp[k] = gp[k];
#pragma unroll
				for (int k = 0; k < MAXORDER; k++)
					nxq[k] = cq[k];
#pragma nounroll
				for (uint32_t g = 0; g < ng; g++) {
					const uint32_t order = __builtin_amdgcn_readfirstlane(nxp[0]);
					const uint32_t shift = __builtin_amdgcn_readfirstlane(nxp[1]);
					const uint32_t psize = __builtin_amdgcn_readfirstlane(nxp[2]);
					const uint32_t flags = __builtin_amdgcn_readfirstlane(nxp[3]);
					const uint32_t fast = flags & 1u;
					int32_t qs[MAXORDER];
#pragma unroll
					for (int k = 0; k < MAXORDER; k++)
						qs[k] = (int32_t)__builtin_amdgcn_readfirstlane((uint32_t)nxq[k]);
					if (g + 1 < ng) {
#pragma unroll
						for (int k = 0; k < 4; k++)
							nxp[k] = gp[4 * (g + 1) + k];
#pragma unroll
						for (int k = 0; k < MAXORDER; k++)
							nxq[k] = cq[(g + 1) * MAXORDER + k];
					}
					uint32_t x[FA_CHUNK], B;
					if (flags & 8u) {
						// the wrapping 32-bit sum as it stands
						int32_t pred[FA_CHUNK];
						lpc_pred16_taps<HIST, MAXORDER, 0>(d, qs, order, (int)shift, pred);
						B = bias;
#pragma unroll
						for (int i = 0; i < FA_CHUNK; i++)
							x[i] = B - ((uint32_t)d[HIST + i] - (uint32_t)pred[i]);
					} else {
						lpc_negres16_taps<HIST, MAXORDER>(d, qs, order, shift, c31, x);
						B = bias >> shift;
					}
					accumulate_negres(x, B, sums + (size_t)g * 2 * PF, base, order, bsf, psize, fast && chunk_inside && (psize & (FA_CHUNK - 1)) == 0,
					                  part_finest >> ((flags >> 16) & 15u));
				}
			}
		}
		STAMP(5);
		SPRIO(SEARCH_PRIO_EDGE);
		__syncthreads();
		STAMP(6);
		{
		// what follows works from a fresh copy of the thread index, so that nothing derived from it
		// earlier is kept alive (spilled) across the residual loop
		const uint32_t tid = opaque_tid();

		const uint32_t fpo = frame_max_po;
		if (PF <= 64) {
			// ---- precompute_partition_info_sums_ (stream_encoder.c:3774), set_partitioned_rice_
			// (stream_encoder.c:3943) and the scan of find_best_partition_order_ (stream_encoder.c:3700-3729)
			// with a wave per candidate and nothing but lane exchanges between the steps.  Lane l holds
			// finest sum l.  Group sums of 2, 4, 8 and 16 lanes come from four cross-lane adds (quad
			// permutes, then the mirror patterns: the values are already uniform inside the smaller group),
			// those of 32 and 64 from the four row sums read into scalars.  The lane whose index has t
			// trailing zeros then holds the sum of the partition t + 1 levels up that starts at it; one
			// permute moves these into level order (level max-1 in lanes 0.., then max-2, ..: the order they
			// have behind the finest sums in LDS), so that lane c prices finest partition c and coarser
			// item c, and the bits of a level are again sums over aligned lane groups. ----
			const uint32_t wv = tid >> 6, ln = tid & 63;
			const uint32_t tz = ln ? (uint32_t)__builtin_ctz(ln) : 0u;
#define DPP_ADD(x, ctrl) ((x) + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), ctrl, 0xf, 0xf, true))
#define DPP_PAIRS 0xB1   /* quad_perm [1,0,3,2] */
#define DPP_QUADS 0x4E   /* quad_perm [2,3,0,1] */
#define DPP_EIGHTS 0x141 /* row_half_mirror */
#define DPP_ROWS 0x140   /* row_mirror */
			for (uint32_t g0 = wv; g0 < ng; g0 += 2 * (FA_THREADS / 64)) {
				const uint32_t gb = g0 + FA_THREADS / 64;
				const bool two = gb < ng;
				const uint32_t gi[2] = {g0, two ? gb : g0};
				uint4 q[2];
				uint64_t f64[2];
				uint32_t hdrb[2];
#pragma unroll
				for (int c = 0; c < 2; c++) {
					q[c] = *(const uint4 *)(gq + 4 * gi[c]);
					hdrb[c] = gc[gi[c]].hdr_bits;
					f64[c] = ln < (1u << (q[c].x & 0xff)) ? sums[(size_t)gi[c] * 2 * PF + ln] : 0ull;
					if (q[c].x >> 16)
						f64[c] &= 0xffffffffull;   // the reference's 32-bit accumulator (stream_encoder.c:3803-3833)
				}
				// 32-bit arithmetic when every finest sum is below 2^26 (no sum of 64 of them reaches 2^32),
				// which is always so for streams of up to 16 bits
				const bool narrow = __all(((f64[0] | f64[1]) >> 26) == 0);
#pragma unroll
				for (int c = 0; c < 2; c++) {
					if (c == 1 && !two)
						break;
					const uint32_t g = gi[c];
					const uint32_t pm = __builtin_amdgcn_readfirstlane(q[c].x & 0xff), pmn = __builtin_amdgcn_readfirstlane((q[c].x >> 8) & 0xff);
					const uint32_t use32 = __builtin_amdgcn_readfirstlane(q[c].x >> 16);
					const uint32_t order = __builtin_amdgcn_readfirstlane(q[c].y), seed = __builtin_amdgcn_readfirstlane(q[c].z);
					uint64_t *gs = sums + (size_t)g * 2 * PF;
					const bool mine = ln < (1u << pm);
					// coarser item of this lane: with x = the zero bits of ln below bit pm, its level is the
					// highest one of them, its partition the bits of ln below that
					const uint32_t zeros = ~ln & ((1u << pm) - 1u);
					const uint32_t poc = zeros ? ilog2u(zeros) : 0u;
					const uint32_t pc = ln & ((1u << poc) - 1u);
					const bool have_c = mine && zeros != 0 && poc >= pmn;
					const uint32_t src = (2u * pc + 1u) << ((pm - 1u - poc) & 31u);   // the lane that holds its sum after the exchanges
					const uint32_t nf = (bsf >> pm) - (ln == 0 ? order : 0u);
					const uint32_t nc = (bsf >> poc) - (pc == 0 ? order : 0u);
					uint32_t bf, bc;
					if (narrow) {
						uint32_t v[7];
						v[0] = (uint32_t)f64[c];
						v[1] = DPP_ADD(v[0], DPP_PAIRS);
						v[2] = DPP_ADD(v[1], DPP_QUADS);
						v[3] = DPP_ADD(v[2], DPP_EIGHTS);
						v[4] = DPP_ADD(v[3], DPP_ROWS);
						const uint32_t r0 = __builtin_amdgcn_readlane(v[4], 0), r1 = __builtin_amdgcn_readlane(v[4], 16);
						const uint32_t r2 = __builtin_amdgcn_readlane(v[4], 32), r3 = __builtin_amdgcn_readlane(v[4], 48);
						v[5] = ln < 32 ? r0 + r1 : r2 + r3;
						v[6] = r0 + r1 + r2 + r3;
						uint32_t up = v[1];
#pragma unroll
						for (int j = 1; j < 6; j++)
							up = tz == (uint32_t)j ? v[j + 1] : up;
						const uint32_t sc = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)up);
						const uint32_t kf = (pm == 0) ? seed : rice_k_from_sum32(v[0], nf, P.rice_limit);
						const uint32_t kc = (poc == 0) ? seed : rice_k_from_sum32(sc, nc, P.rice_limit);
						bf = rice_bits_estimate32(kf, nf, v[0]);
						bc = rice_bits_estimate32(kc, nc, sc);
						if (mine && use32)
							gs[ln] = v[0];
						if (have_c)
							gs[(1u << pm) + ln] = sc;   // the winner's parameters are derived from these
					} else {
						uint64_t v[7];
						v[0] = f64[c];
#pragma unroll
						for (int j = 0; j < 6; j++)
							v[j + 1] = v[j] + (uint64_t)__shfl_xor((unsigned long long)v[j], 1 << j, 64);
						uint64_t up = v[1];
#pragma unroll
						for (int j = 1; j < 6; j++)
							up = tz == (uint32_t)j ? v[j + 1] : up;
						const uint64_t sc = (uint64_t)__shfl((unsigned long long)up, (int)src, 64);
						const uint32_t kf = (pm == 0) ? seed : rice_k_from_sum(v[0], nf, P.rice_limit);
						const uint32_t kc = (poc == 0) ? seed : rice_k_from_sum(sc, nc, P.rice_limit);
						bf = rice_bits_estimate(kf, nf, v[0]);
						bc = rice_bits_estimate(kc, nc, sc);
						if (mine && use32)
							gs[ln] = v[0];
						if (have_c)
							gs[(1u << pm) + ln] = sc;
					}
					// bits per level: the finest one is the sum over the wave, level po that of the 2^po lanes
					// from lane 2^pm - 2^(po+1) on
					uint32_t wf[5], wc[5];
					wf[0] = mine ? bf : 0u;
					wc[0] = have_c ? bc : 0u;
					wf[1] = DPP_ADD(wf[0], DPP_PAIRS);
					wc[1] = DPP_ADD(wc[0], DPP_PAIRS);
					wf[2] = DPP_ADD(wf[1], DPP_QUADS);
					wc[2] = DPP_ADD(wc[1], DPP_QUADS);
					wf[3] = DPP_ADD(wf[2], DPP_EIGHTS);
					wc[3] = DPP_ADD(wc[2], DPP_EIGHTS);
					wf[4] = DPP_ADD(wf[3], DPP_ROWS);
					wc[4] = DPP_ADD(wc[3], DPP_ROWS);
					// find_best_partition_order_: from the largest order down, a smaller one must be strictly better
					uint32_t bestb = 6u + __builtin_amdgcn_readlane(wf[4], 0) + __builtin_amdgcn_readlane(wf[4], 16) + __builtin_amdgcn_readlane(wf[4], 32) +
					                 __builtin_amdgcn_readlane(wf[4], 48);
					uint32_t bestpo = pm;
#pragma unroll
					for (int po = 5; po >= 0; po--) {
						if ((uint32_t)po < pm && (uint32_t)po >= pmn) {
							const uint32_t at = (1u << pm) - (2u << po);
							const uint32_t lb = po == 5 ? __builtin_amdgcn_readlane(wc[4], 0) + __builtin_amdgcn_readlane(wc[4], 16)
							                            : __builtin_amdgcn_readlane(wc[po], at);
							if (6u + lb < bestb) {
								bestb = 6u + lb;
								bestpo = (uint32_t)po;
							}
						}
					}
					if (ln == 0) {
						SrCand &cd = gc[g];
						cd.best_po = bestpo;
						cd.best_rbits = bestb;
						cd.total_bits = hdrb[c] + bestb;
					}
				}
			}
#undef DPP_ADD
		} else {
		// ---- precompute_partition_info_sums_ (stream_encoder.c:3774): every finest sum, truncated to
		// 32 bits where the reference's accumulator is that narrow (:3803-3833), is added into the
		// partition above it at each coarser level.  The adds are LDS atomics nobody waits for. ----
		for (uint32_t e = tid; e < (ng << fpo); e += FA_THREADS) {
			const uint32_t g = e >> fpo, part = e & (PF - 1);
			const uint4 q = *(const uint4 *)(gq + 4 * g);
			const uint32_t pm = q.x & 0xff, pmn = (q.x >> 8) & 0xff;
			uint64_t *gs = sums + (size_t)g * 2 * PF;
			if (part < (1u << pm)) {
				uint64_t v = gs[part];
				if (q.x >> 16) {
					v &= 0xffffffffull;
					gs[part] = v;
				}
				for (uint32_t lvl = pm; lvl-- > pmn;)
					atomicAdd((unsigned long long *)&gs[(2u << pm) - (2u << lvl) + (part >> (pm - lvl))], (unsigned long long)v);
			}
		}
		__syncthreads();
		// ---- set_partitioned_rice_ (stream_encoder.c:3943) for every (candidate, order, partition) ----
		for (uint32_t e = tid; e < (ng << (fpo + 1)); e += FA_THREADS) {
			const uint32_t g = e >> (fpo + 1), idx = e & (2 * PF - 1);
			const uint4 q = *(const uint4 *)(gq + 4 * g);
			const uint64_t sm = sums[(size_t)g * 2 * PF + idx];
			const uint32_t pm = q.x & 0xff, pmn = (q.x >> 8) & 0xff;
			const uint32_t span = 2u << pm;
			if (idx + 1 < span) {
				const uint32_t r = span - idx;
				const uint32_t po = ilog2u(r - 1);
				if (po >= pmn) {
					const uint32_t part = (2u << po) - r;
					uint32_t n = bsf >> po;
					if (part == 0)
						n -= q.y;
					const uint32_t k = (po == 0) ? q.z : rice_k_from_sum(sm, n, P.rice_limit);
					atomicAdd(&pbits[g * 16 + po], rice_bits_estimate(k, n, sm));
				}
			}
		}
		}
		__syncthreads();
		STAMP(7);
		// ---- find_best_partition_order_ (stream_encoder.c:3700-3729): scan max -> min, strict <;
		// then the first candidate with the fewest bits wins if it beats what came before ----
		if (tid < 64) {
			uint32_t total = 0xffffffffu;
			if (tid < ng) {
				SrCand &c = gc[tid];
				if (PF > 64) {
					uint32_t bestb = 0, bestpo = 0;
					for (int po = (int)c.pmax; po >= (int)c.pmin; po--) {
						const uint32_t bits = 2 + 4 + pbits[tid * 16 + po];
						if (bestb == 0 || bits < bestb) {
							bestb = bits;
							bestpo = (uint32_t)po;
						}
					}
					c.best_po = bestpo;
					c.best_rbits = bestb;
					c.total_bits = c.hdr_bits + bestb;
				}
				total = c.total_bits;
			}
			uint32_t mn = total;
#pragma unroll
			for (int o = 32; o > 0; o >>= 1)
				mn = umin32(mn, (uint32_t)__shfl_xor((int)mn, o, 64));
			const uint64_t at = __ballot(tid < ng && total == mn);
			if (tid == 0) {
				ctl[2] = (at && mn < best->bits) ? (uint32_t)__ffsll((unsigned long long)at) - 1u : 0xffffffffu;
				ctl[3] = 0;
			}
		}
		__syncthreads();
		const uint32_t winner = ctl[2];
		if (winner != 0xffffffffu) {
			// keep the winner's parameters: its sums are overwritten by the next group
			const SrCand &c = gc[winner];
			const uint32_t po = c.best_po;
			if (tid < (1u << po)) {
				const uint64_t sm = sums[(size_t)winner * 2 * PF + (2u << c.pmax) - (2u << po) + tid];
				uint32_t n = bsf >> po;
				if (tid == 0)
					n -= c.order;
				const uint32_t k = (po == 0) ? c.seed : rice_k_from_sum(sm, n, P.rice_limit);
				best->params[tid] = (uint8_t)k;
				if (k >= 15)
					atomicOr(&ctl[3], 1u);  // stream_encoder.c:3752-3757
			}
			if (tid < c.order && c.kind == FA_SUB_LPC)
				best->q[tid] = cq[winner * MAXORDER + tid];
			__syncthreads();
			if (tid == 0) {
				best->bits = c.total_bits;
				best->type = (uint8_t)c.kind;
				best->order = (uint8_t)c.order;
				best->precision = (uint8_t)c.precision;
				best->shift = (int8_t)c.shift;
				best->part_order = (uint8_t)po;
				best->rice2 = (uint8_t)ctl[3];
				best->wide = (uint8_t)c.mode;
			}
		}
		__syncthreads();
		}
		STAMP(8);
		first_pass = false;
	}
	if (tid == 0 && best->bits == 0xffffffffu) {
		// stream_encoder.c:3388-3391
		best->type = FA_SUB_VERBATIM;
		best->bits = hdr + bsf * sbps;
	}
	__syncthreads();
	// write the plan out
	{
		const uint32_t *src = (const uint32_t *)best;
		uint32_t *dst = (uint32_t *)(plans + task);
		for (uint32_t e = tid; e < sizeof(FaPlan) / 4; e += FA_THREADS)
			dst[e] = src[e];
	}
	first_pass = true;
	STAMP(9);
#undef STAMP
}

// LDS for the finest-partition sums: room for every candidate of a subframe in one group when
// that fits
static inline uint64_t umin64(uint64_t a, uint64_t b) { return a < b ? a : b; }

static uint32_t search_sum_bytes(const FaEncParams &P, uint32_t ncmax)
{
	uint32_t po = 0;
	while (po < P.max_po && !((P.blocksize >> po) & 1))
		po++;
	const uint32_t per_cand = (2u << po) * 8;
	uint32_t want = ncmax * per_cand;
	if (want > SR_MAXSUM_BYTES)
		want = SR_MAXSUM_BYTES;
	if (want < per_cand)
		want = per_cand;
	return (want + 15) & ~15u;
}

void fa_launch_search(const FaEncParams &P, const FaBatch &B, const void *pcm, const FaSubStat *stats, const FaLpcCand *cands,
                      const int32_t *coefs, FaPlan *plans, unsigned long long *stamps, hipStream_t st)
{
	const dim3 block(FA_THREADS);
	const int hist = fa_hist_class(P);
	uint32_t ncmax = 5 + P.ncand;
	if (ncmax > SR_MAXGROUP)
		ncmax = SR_MAXGROUP;
	const uint32_t sum_bytes = search_sum_bytes(P, ncmax);
	// with one tile per frame the samples are in registers before the sums are first touched, so
	// the two can share LDS
	const uint32_t alias = P.blocksize <= FA_TILE ? 1u : 0u;
	const uint32_t stage_bytes = (hist == 4 ? ST_WORDS(4) : hist == 12 ? ST_WORDS(12) : ST_WORDS(32)) * 4;
	const uint32_t qstride = hist == 4 ? 4 : hist == 12 ? 12 : 32;
	const uint32_t ncoef = P.ncand * P.max_lpc_order;
	const SrLayout L = sr_layout(sum_bytes, stage_bytes, alias, ncmax, qstride, P.ncand < SR_PRE_CANDS ? P.ncand : SR_PRE_CANDS,
	                             ncoef < SR_PRE_COEFS ? ncoef : SR_PRE_COEFS);
	const size_t lds = L.total;
	// signals that fit 16 bits go to the launch with the two-taps-per-instruction residual: all of
	// them below 16 bits per sample, all but the side signal (one bit wider) at 16
	unsigned long long list_dot2 = 0, list_taps = 0;
	uint32_t n_dot2 = 0, n_taps = 0;
	for (uint32_t sg = 0; sg < P.nsignals; sg++) {
		const uint32_t width = P.bps + ((P.do_ms && sg == P.channels + 1) ? 1u : 0u);
		if (hist >= 12 && width <= 16)
			list_dot2 |= (unsigned long long)sg << (4 * n_dot2++);
		else
			list_taps |= (unsigned long long)sg << (4 * n_taps++);
	}
	// Streams with blocks of 4096 samples go to the wave-per-subframe form where it applies (FLAC_AMD_SEARCH=generic
	// keeps every signal on the workgroup form, for comparisons): every signal when the stream needs 32-bit sums
	// only, the signals of up to 16 bits otherwise.  first[R]: the first frame the workgroup form still has to do.
	const bool narrow = fa_narrow_only(P);
	uint32_t first_dot2 = 0, first_taps = 0;
	{
		const char *mode = getenv("FLAC_AMD_SEARCH");
		const bool generic_only = (mode && !strcmp(mode, "generic")) || stamps;
		if (!generic_only && !B.pairs && fa_search_wave_split_eligible(P)) {
			// 20- to 24-bit streams: every signal to the split-sample form of the wave search
			unsigned long long all = 0;
			for (uint32_t sg = 0; sg < P.nsignals; sg++)
				all |= (unsigned long long)sg << (4 * sg);
			fa_launch_search_wave_split(P, B, pcm, stats, cands, coefs, plans, all, P.nsignals, st);
			first_dot2 = first_taps = B.nframes;
		} else if (!generic_only && fa_search_wave_eligible(P)) {
			unsigned long long wl = 0, wide = 0;
			uint32_t wn = 0, nwide = 0, left = 0;
			for (uint32_t sg = 0; sg < P.nsignals; sg++) {
				const uint32_t width = P.bps + ((P.do_ms && sg == P.channels + 1) ? 1u : 0u);
				if (width <= 16 || narrow) {
					wl |= (unsigned long long)sg << (4 * wn++);
					if (width > 16)
						wide |= (unsigned long long)sg << (4 * nwide++);
				} else
					left++;
			}
			// (the two lists of the workgroup form are split by the same widths when it has a two-taps form at all)
			if (wn && (left == 0 || hist >= 12)) {
				fa_launch_search_wave(P, B, pcm, stats, cands, coefs, plans, wl, wn, wide, nwide, st);
				if (left == 0)
					first_dot2 = first_taps = B.nframes;
				else
					first_dot2 = B.nframes;
			}
		}
	}
#define SEARCH_LAUNCH1(H, N, PR, R, LIST, NSEL, FIRST) \
	do { \
		if ((FIRST) < B.nframes) \
			hipLaunchKernelGGL((k_search<H, N, PR, R>), dim3((B.nframes - (FIRST) + 7) / 8 * 8 * (NSEL)), block, lds, st, P, B, pcm, stats, cands, coefs, plans, \
			                   sum_bytes, alias, ncmax, stamps, LIST, NSEL, FIRST); \
	} while (0)
	// the signals too wide for one 16-bit lane: a tap per instruction while 32-bit sums of 24-bit
	// samples are all the stream can need, the split form otherwise
#define SEARCH_LAUNCH(H, N) \
	do { \
		if (B.pairs) { \
			if (n_dot2) SEARCH_LAUNCH1(H, N, true, (H >= 12 ? SR_RES_DOT2 : SR_RES_TAPS), list_dot2, n_dot2, first_dot2); \
			if (n_taps) SEARCH_LAUNCH1(H, N, true, (N ? SR_RES_TAPS : SR_RES_SPLIT), list_taps, n_taps, first_taps); \
		} else { \
			if (n_dot2) SEARCH_LAUNCH1(H, N, false, (H >= 12 ? SR_RES_DOT2 : SR_RES_TAPS), list_dot2, n_dot2, first_dot2); \
			if (n_taps) SEARCH_LAUNCH1(H, N, false, (N ? SR_RES_TAPS : SR_RES_SPLIT), list_taps, n_taps, first_taps); \
		} \
	} while (0)
	switch (hist) {
	case 4: SEARCH_LAUNCH(4, true); break;
	case 12: if (narrow) SEARCH_LAUNCH(12, true); else SEARCH_LAUNCH(12, false); break;
	default: if (narrow) SEARCH_LAUNCH(32, true); else SEARCH_LAUNCH(32, false); break;
	}
#undef SEARCH_LAUNCH1
#undef SEARCH_LAUNCH
}
