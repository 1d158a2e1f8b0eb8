// FLAC frame encoder kernels for gfx950 (MI355X), part 2: the model search.
#include <hip/hip_runtime.h>

#include "enc_common.h"
#include "enc_kernels.h"

// ---------------------------------------------------------------------------------------------
// k_search: process_subframe_ (stream_encoder.c:3164) for one (frame, signal): every candidate's
// residual is produced in registers, only |residual| partition sums go to LDS, and
// find_best_partition_order_ / set_partitioned_rice_ (stream_encoder.c:3666,3943) run on those.
// ---------------------------------------------------------------------------------------------
#define SR_SUM_BYTES 32768
#define SR_MAXGROUP 32

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
	uint32_t best_po, best_rbits, total_bits;
};

// Add |res[i]| of the chunk's samples that lie in [order, bsf) to the candidate's partition sums
// (the lowest level of precompute_partition_info_sums_, stream_encoder.c:3774).  fast: every
// |res[i]| of this candidate is known to be below 2^27, so a chunk's sum fits 32 bits.
__device__ __forceinline__ void accumulate_chunk(const int32_t (&res)[FA_CHUNK], uint64_t *gs, uint32_t base, uint32_t order, uint32_t bsf,
                                                 uint32_t psize, uint32_t fast)
{
	if ((psize & (FA_CHUNK - 1)) == 0) {
		// the whole chunk lies in one partition
		uint64_t t;
		if (fast) {
			uint32_t t32 = 0;
			if (base >= order && base + FA_CHUNK <= bsf) {
#pragma unroll
				for (int i = 0; i < FA_CHUNK; i++)
					t32 += abs32(res[i]);
			} else {
#pragma unroll
				for (int i = 0; i < FA_CHUNK; i++) {
					const uint32_t idx = base + i;
					if (idx >= order && idx < bsf)
						t32 += abs32(res[i]);
				}
			}
			t = t32;
		} else {
			t = 0;
#pragma unroll
			for (int i = 0; i < FA_CHUNK; i++) {
				const uint32_t idx = base + i;
				if (idx >= order && idx < bsf)
					t += abs32(res[i]);
			}
		}
		atomicAdd((unsigned long long *)&gs[base / psize], (unsigned long long)t);
	} else {
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
}

template <int HIST, bool NARROW>
__global__ __launch_bounds__(FA_THREADS) void k_search(FaEncParams P, FaBatch B, const void *pcm, const FaSubStat *stats,
                                                       const FaLpcCand *cands, const int32_t *coefs, FaPlan *plans, uint32_t sum_bytes)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	uint64_t *sums = (uint64_t *)smem;                                  // [group][2 * PF]
	uint32_t *pbits = (uint32_t *)(smem + sum_bytes);                // [group][16]
	SrCand *gc = (SrCand *)(pbits + SR_MAXGROUP * 16);                  // [group]
	FaPlan *best = (FaPlan *)(gc + SR_MAXGROUP);
	uint32_t *ctl = (uint32_t *)(best + 1);                             // [0]=ngroup [1]=cursor [2]=best_changed_to [3]=rice2 flag
	int32_t *cq = (int32_t *)(ctl + 16);                                // [group][FA_MAX_ORDER] quantised coefficients
	int32_t *stage = cq + SR_MAXGROUP * FA_MAX_ORDER;                   // the staged signal tile

	const uint32_t task = blockIdx.x;
	const uint32_t f = task / P.nsignals, s = task % P.nsignals;
	const uint32_t tid = threadIdx.x;
	const uint32_t bsf = frame_len(B, P.blocksize, f);
	const uint64_t frame_start = (uint64_t)f * P.blocksize;
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
	const uint32_t group_cap = umin32(SR_MAXGROUP, sum_bytes / (2 * PF * 8));
	const FaLpcCand *my_cands = cands + (size_t)task * P.ncand;
	const int32_t *my_coefs = coefs + (size_t)task * P.ncand * P.max_lpc_order;

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
	__syncthreads();

	const bool search = bsf >= 4 && !st.is_constant;
	const uint32_t total_cands = 5 + P.ncand;
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
				c.mode = 0;
				c.hdr_bits = 0;
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
						take = true;
					}
				} else if (my < total_cands) {
					const FaLpcCand lc = my_cands[my - 5];
					if (lc.valid) {
						c.kind = FA_SUB_LPC;
						c.order = lc.order;
						c.seed = lc.rice;
						c.slot = my - 5;
						c.precision = lc.precision;
						c.shift = lc.shift;
						c.mode = (uint32_t)lpc_mode(lc.wide, sbps);
						c.hdr_bits = hdr + 4 + 5 + lc.order * (lc.precision + sbps);
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
				c.fast = (c.kind == FA_SUB_FIXED && sbps <= 23) ? 1u : 0u;  // |4th difference| < 2^(sbps+3)
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
		const uint32_t ng = ctl[0];
		if (ng == 0)
			break;
		for (uint32_t e = tid; e < ng * 2 * PF; e += FA_THREADS)
			sums[e] = 0;
		for (uint32_t e = tid; e < ng * 16; e += FA_THREADS)
			pbits[e] = 0;
		for (uint32_t e = tid; e < ng * FA_MAX_ORDER; e += FA_THREADS) {
			const uint32_t g = e / FA_MAX_ORDER, j = e % FA_MAX_ORDER;
			cq[e] = (gc[g].kind == FA_SUB_LPC && j < gc[g].order) ? my_coefs[(size_t)gc[g].slot * P.max_lpc_order + j] : 0;
		}
		__syncthreads();
		if (tid < ng) {
			SrCand &c = gc[tid];
			if (c.kind == FA_SUB_LPC) {
				// |prediction| <= (sum |q| * max |sample|) >> shift, |residual| <= |sample| + |prediction|
				uint64_t sumq = 0;
				for (uint32_t j = 0; j < c.order; j++)
					sumq += abs32(cq[tid * FA_MAX_ORDER + j]);
				const uint64_t bound = (1ull << (sbps - 1)) + ((sumq << (sbps - 1)) >> c.shift) + 1;
				c.fast = (c.mode == 0 && bound < (1ull << 27)) ? 1u : 0u;
			}
		}
		if (tid == 64) {
			uint32_t nf = 0;
			while (nf < ng && gc[nf].kind == FA_SUB_FIXED)
				nf++;
			ctl[4] = nf;
		}

		// ---- residuals of every candidate of the group, tile by tile, all in registers ----
		for (uint32_t tile = 0; tile * FA_TILE < bsf; tile++) {
			__syncthreads();
			stage_tile<HIST>(stage, pcm, P.channels, B.sample_bytes, B.pairs, frame_start, bsf, s, wasted, tile * FA_TILE);
			__syncthreads();
			const int32_t base = (int32_t)(tile * FA_TILE + tid * FA_CHUNK);
			if ((uint32_t)base >= bsf)
				continue;
			int32_t d[HIST + FA_CHUNK];
			load_chunk_lds<HIST>(d, stage);
			// fixed predictors (always the leading candidates of the first group): the order-o
			// residual is the o-th difference (fixed.c:352), taken in place one order at a time
			const uint32_t nfixed = __builtin_amdgcn_readfirstlane(ctl[4]);
			if (nfixed) {
				int32_t r[4 + FA_CHUNK];
#pragma unroll
				for (int k = 0; k < 4 + FA_CHUNK; k++)
					r[k] = d[HIST - 4 + k];
				uint32_t g = 0;
#pragma unroll
				for (uint32_t o = 0; o <= 4; o++) {
					if (o > 0) {
#pragma unroll
						for (int k = 4 + FA_CHUNK - 1; k >= (int)o; k--)
							r[k] = (int32_t)((uint32_t)r[k] - (uint32_t)r[k - 1]);
					}
					if (g < nfixed && __builtin_amdgcn_readfirstlane(gc[g].order) == o) {
						int32_t res[FA_CHUNK];
#pragma unroll
						for (int i = 0; i < FA_CHUNK; i++)
							res[i] = r[4 + i];
						accumulate_chunk(res, sums + (size_t)g * 2 * PF, (uint32_t)base, o, bsf, __builtin_amdgcn_readfirstlane(gc[g].psize),
						                 __builtin_amdgcn_readfirstlane(gc[g].fast));
						g++;
					}
				}
			}
#pragma nounroll
			for (uint32_t g = nfixed; g < ng; g++) {
				const uint32_t order = __builtin_amdgcn_readfirstlane(gc[g].order);
				const int shift = (int)__builtin_amdgcn_readfirstlane((uint32_t)gc[g].shift);
				const int mode = (int)__builtin_amdgcn_readfirstlane(gc[g].mode);
				int32_t res[FA_CHUNK];
#ifdef EXP_FIXED12
				lpc_residual16<HIST, 12, 0>(d, cq + g * FA_MAX_ORDER, shift, res);
#else
				lpc_residual16_any<HIST, NARROW>(d, cq + g * FA_MAX_ORDER, order, shift, mode, res);
#endif
				accumulate_chunk(res, sums + (size_t)g * 2 * PF, (uint32_t)base, order, bsf, __builtin_amdgcn_readfirstlane(gc[g].psize),
				                 __builtin_amdgcn_readfirstlane(gc[g].fast));
			}
		}
		__syncthreads();

		// ---- precompute_partition_info_sums_ (stream_encoder.c:3774): truncate, then merge ----
		for (uint32_t e = tid; e < ng * PF; e += FA_THREADS) {
			const uint32_t g = e / PF, part = e % PF;
			if (part < (1u << gc[g].pmax) && gc[g].use32)
				sums[(size_t)g * 2 * PF + part] &= 0xffffffffull;
		}
		__syncthreads();
		for (int lvl = (int)frame_max_po - 1; lvl >= 0; lvl--) {
			for (uint32_t e = tid; e < ng * (1u << lvl); e += FA_THREADS) {
				const uint32_t g = e >> lvl, part = e & ((1u << lvl) - 1);
				const uint32_t pm = gc[g].pmax;
				if ((uint32_t)lvl < pm && (uint32_t)lvl >= gc[g].pmin) {
					uint64_t *gs = sums + (size_t)g * 2 * PF;
					const uint32_t off_hi = (2u << pm) - (2u << (lvl + 1));
					const uint32_t off_lo = (2u << pm) - (2u << lvl);
					gs[off_lo + part] = gs[off_hi + 2 * part] + gs[off_hi + 2 * part + 1];
				}
			}
			__syncthreads();
		}
		// ---- set_partitioned_rice_ (stream_encoder.c:3943) for every (candidate, order, partition) ----
		for (uint32_t e = tid; e < ng * 2 * PF; e += FA_THREADS) {
			const uint32_t g = e / (2 * PF), idx = e % (2 * PF);
			const SrCand &c = gc[g];
			const uint32_t span = 2u << c.pmax;
			if (idx + 1 >= span)
				continue;
			const uint32_t r = span - idx;
			const uint32_t po = ilog2u(r - 1);
			if (po < c.pmin)
				continue;
			const uint32_t part = (2u << po) - r;
			uint32_t n = bsf >> po;
			if (part == 0)
				n -= c.order;
			const uint64_t sm = sums[(size_t)g * 2 * PF + idx];
			const uint32_t k = (po == 0) ? c.seed : rice_k_from_sum(sm, n, P.rice_limit);
			atomicAdd(&pbits[g * 16 + po], rice_bits_estimate(k, n, sm));
		}
		__syncthreads();
		// ---- find_best_partition_order_ (stream_encoder.c:3700-3729): scan max -> min, strict < ----
		if (tid < ng) {
			SrCand &c = gc[tid];
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
		__syncthreads();
		if (tid < 64) {
			// the first candidate with the fewest bits wins, if it beats what came before (strict <)
			uint32_t bits = tid < ng ? gc[tid].total_bits : 0xffffffffu;
			uint32_t mn = bits;
#pragma unroll
			for (int o = 32; o > 0; o >>= 1)
				mn = umin32(mn, (uint32_t)__shfl_xor((int)mn, o, 64));
			const uint64_t at = __ballot(tid < ng && bits == mn);
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
			const uint32_t off = (2u << c.pmax) - (2u << po);
			if (tid < (1u << po)) {
				uint32_t n = bsf >> po;
				if (tid == 0)
					n -= c.order;
				const uint64_t sm = sums[(size_t)winner * 2 * PF + off + tid];
				const uint32_t k = (po == 0) ? c.seed : rice_k_from_sum(sm, n, P.rice_limit);
				best->params[tid] = (uint8_t)k;
				if (k >= 15)
					atomicOr(&ctl[3], 1u);  // stream_encoder.c:3752-3757
			}
			if (tid < c.order && c.kind == FA_SUB_LPC)
				best->q[tid] = my_coefs[(size_t)c.slot * P.max_lpc_order + tid];
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
}

// LDS for the partition sums: room for every candidate of a subframe in one group when that fits
static uint32_t search_sum_bytes(const FaEncParams &P)
{
	uint32_t po = 0;
	while (po < P.max_po && !((P.blocksize >> po) & 1))
		po++;
	const uint32_t per_cand = (2u << po) * 8;
	uint32_t want = (5 + P.ncand) * per_cand;
	if (want > SR_MAXGROUP * per_cand)
		want = SR_MAXGROUP * per_cand;
	if (want > SR_SUM_BYTES)
		want = SR_SUM_BYTES;
	if (want < per_cand)
		want = per_cand;
	return (want + 15) & ~15u;
}

static size_t search_lds_bytes(int hist, uint32_t sum_bytes)
{
	const size_t stage_words = hist == 4 ? ST_WORDS(4) : hist == 12 ? ST_WORDS(12) : ST_WORDS(32);
	return sum_bytes + SR_MAXGROUP * 16 * sizeof(uint32_t) + SR_MAXGROUP * sizeof(SrCand) + sizeof(FaPlan) + 16 * sizeof(uint32_t) +
	       SR_MAXGROUP * FA_MAX_ORDER * sizeof(int32_t) + stage_words * sizeof(int32_t);
}

void fa_launch_search(const FaEncParams &P, const FaBatch &B, const void *pcm, const FaSubStat *stats, const FaLpcCand *cands,
                      const int32_t *coefs, FaPlan *plans, hipStream_t st)
{
	const dim3 grid(B.nframes * P.nsignals), block(FA_THREADS);
	const int hist = fa_hist_class(P);
	const uint32_t sum_bytes = search_sum_bytes(P);
	const size_t lds = search_lds_bytes(hist, sum_bytes);
#define SEARCH_LAUNCH(H, N) hipLaunchKernelGGL((k_search<H, N>), grid, block, lds, st, P, B, pcm, stats, cands, coefs, plans, sum_bytes)
	const bool narrow = fa_narrow_only(P);
#ifdef EXP_ONLY12
	SEARCH_LAUNCH(12, true);
#else
	switch (hist) {
	case 4: SEARCH_LAUNCH(4, true); break;
	case 12: if (narrow) SEARCH_LAUNCH(12, true); else SEARCH_LAUNCH(12, false); break;
	default: if (narrow) SEARCH_LAUNCH(32, true); else SEARCH_LAUNCH(32, false); break;
	}
#endif
#undef SEARCH_LAUNCH
}
