// pack.hip — the encoder's entropy stage on the GPU: bit-plane coder
// (encode.c:60-95), progressive plane schedule (encode.c:183-221), zero-run RLE
// (rle.h:56-103), adaptive VLI (vli.h:67-84), LSB-first bit packing
// (bits.h:58-78) and the root/header coder (encode.c:97-110,166-182).
//
// The reference is one serial loop: every coefficient visit may emit bits whose
// position and length depend on everything emitted before (a shared zero-run
// counter and a shared adaptive VLI order).  Here the stream is rebuilt from
// data-parallel passes over a batch of images:
//
//   k_hist    one wave per 1024-coefficient tile: cumulative magnitude
//             histogram cum[p] = #(|v| < 2^p) by ballot/popcount.  Every
//             (tile, plane) symbol count — zeros, ones, refinement bits — is a
//             difference of two entries, so the coefficients are read once here
//             for all planes; max plane count per channel by atomicMax.
//   k_plan    per image (serial, tiny): header + root image + plane counts
//             written straight into the stream, the segment schedule, VLI order
//             after the header.
//   k_entries_* an "entry" is (segment, tile).  Symbol counts per
//             entry from the histograms, exclusive scans -> first token slot
//             and refinement rank of every entry.
//   k_tokens  one wave per tile, all planes: classify 64 coefficients per row at
//             each plane that codes the tile; every newly significant one
//             becomes a token (zero run since the previous one in this tile,
//             sign).
//   k_carry_* batch-wide segmented scan of pending zero runs across entries,
//             segment ends (phantom terminators, rle.h:79-89) and the final
//             flush; patches the first token of each entry.
//   k_orders_fast  the VLI order recurrence o' = max(ilog2(v+2^o)-2,0) per 64-token
//             group from the chains started at order 0 and 31 (they almost
//             always meet); k_lut/k_chain*/k_orders redo flagged images exactly
//             as a scan over monotone maps on 32 states.
//   k_bitscan per image: exclusive scan of chunk bit totals -> bit offsets.
//   k_clear_stream  zeroes the words the stream will occupy.
//   k_emit    one lane per four tokens: their codes glued into one bit string,
//             a wave's strings merged in LDS, words merged with atomicOr.
//   k_refine  one wave per tile, all planes: refinement bits, compacted through LDS.
#include "dwtx_internal.h"

#include <stdlib.h>
#include <string.h>

namespace {

constexpr int TILE = 1024;
constexpr int ROWS = TILE / 64;
constexpr int NCUM = 32;              // cum[0..31]
constexpr int MAX_PLANES = 16;        // 8-bit sources stay far below (checked in k_plan)
constexpr int MAX_SEGS = 3 * 16 * MAX_PLANES;
constexpr int SUB = 64;               // tokens per lane
constexpr int CHUNK = SUB * 64;       // tokens per wave
constexpr int GROUP = 64;             // chunks per group

constexpr unsigned F_SIGN = 1, F_HAS_SIGN = 2, F_BREAK = 4, F_FLUSH = 8, F_VOID = 16;

struct PackGeom {
	int levels, C, W, H;
	long total;
	int pixels[DWTX_MAX_LEVELS + 1];
	int tile_first[DWTX_MAX_LEVELS + 1];   // tile_first[levels] = tiles per plane
};

struct ImgInfo {
	int planes[3];
	int pmax;
	int K;                 // segments
	int E;                 // entries
	unsigned T;            // tokens
	int order0;            // VLI order after header + root + plane counts
	unsigned hdr_bits;
	unsigned root_bits;
	unsigned meta_bits;
	unsigned pad0;
	unsigned long long total_bits;
	unsigned long long nbytes;
	int error;
	int pad;
};

// pending-run map of the carry scan (k_carry_*): s -> add + (keep ? s : 0)
struct RunMap {
	unsigned keep, add;
};

struct Work {
	// per plane
	unsigned short *cum;        // [nplanes][NT][32]
	int *planes_dev;            // [nplanes]
	// per image
	ImgInfo *info;              // [n]
	int *seg_desc;              // [n][MAX_SEGS]   c | l<<4 | (p+1)<<8
	int *seg_ebase;             // [n][MAX_SEGS+1]
	unsigned *seg_refs;         // [n][MAX_SEGS]
	unsigned long long *seg_rawoff; // [n][MAX_SEGS] bit offset of the segment's refinement block
	unsigned *brk_tok;          // [n][MAX_SEGS] token index of the segment's break slot
	int *segidx;                // [n][3][16][MAX_PLANES] -> k+1 of the segment coding (channel, level, plane)
	// per entry
	unsigned short *ent_ones, *ent_zeros, *ent_refs, *ent_tz;   // [n][ES]
	unsigned *ent_tokbase;      // [n][ES+1]
	unsigned *ent_refscum;      // [n][ES+1]
	// per token
	unsigned *tok_run;          // [n][TS]
	unsigned char *tok_flag;    // [n][TS]
	unsigned char *tok_ord;     // [n][TS] VLI order each token is coded with
	unsigned short *tok_off;    // [n][TS] bit offset inside its group of 64 tokens
	// per chunk
	unsigned char *sublut;      // [n][NCS*64][32]
	unsigned char *lut;         // [n][NCS][32]
	unsigned char *glut;        // [n][NGS][32]
	unsigned char *chunk_entry; // [n][NCS]
	unsigned char *group_entry; // [n][NGS]
	unsigned long long *chunk_bits;    // [n][NCS]
	unsigned long long *chunk_base;    // [n][NCS]
	unsigned long long *lane_bits;     // [n][NCS*64] per 64-token group: bit offset inside its wave's chunk
	RunMap *carry_agg;                 // [n][NCB] map of each block of 1024 entries
	unsigned *carry_in;                // [n][NCB] pending run entering the block
	unsigned *ent_blk;                 // [n][NCB][2] token slots / refinement bits of each block of 1024 entries, then their scan
	unsigned long long *stream_bits;   // [n] bits of the whole stream before any capacity clip (k_bitscan -> k_clear_stream)
	int *slow;                         // [n] set when the fast order pass could not resolve an image
	long ES, TS, NCS, NGS, NCB;
	int NT;
};

__device__ __forceinline__ int lane_id()
{
	return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0));
}

__device__ __forceinline__ int popc_below(unsigned long long m)
{
	return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
}

__device__ __forceinline__ int ilog2u(unsigned v) { return 31 - __builtin_clz(v); }

// vli.h:67-84 in closed form: order o, value v -> o* 
__device__ __forceinline__ int vli_top(int o, unsigned v) { return ilog2u(v + (1u << o)); }
__device__ __forceinline__ int vli_next(int top) { return top >= 2 ? top - 2 : 0; }

__device__ __forceinline__ void seg_unpack(int d, int &c, int &l, int &p)
{
	c = d & 15;
	l = (d >> 4) & 15;
	p = (d >> 8) - 1;
}

// ------------------------------------------------------------------ k_hist ---

__global__ __launch_bounds__(256) void k_hist(PackGeom g, const int *__restrict__ lin, Work w)
{
	const int lane = threadIdx.x & 63;
	const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
	const int plane = blockIdx.y;
	if (tile >= w.NT)
		return;
	int l = 0;
	while (l + 1 < g.levels && tile >= g.tile_first[l + 1])
		++l;
	const long ring0 = g.pixels[l], ring1 = g.pixels[l + 1];
	const long base = ring0 + (long)(tile - g.tile_first[l]) * TILE;
	const int *src = lin + plane * g.total;
	unsigned m[ROWS];
	unsigned mx = 0;
	int valid = 0;
#pragma unroll
	for (int r = 0; r < ROWS; ++r) {
		const long i = base + r * 64 + lane;
		const bool in = i < ring1;
		const int v = in ? src[i] : 0;
		m[r] = in ? (unsigned)(v < 0 ? -v : v) : 0xffffffffu;   // out-of-ring lanes count nowhere
		mx |= in ? m[r] : 0u;
		valid += __builtin_popcountll(ballot64(in));
	}
	for (int o = 32; o; o >>= 1)
		mx |= __shfl_xor(mx, o);
	// bits needed by the largest magnitude of the tile: cum[p] = valid for every p >= that
	const int top = mx ? ilog2u(mx) + 1 : 0;
	int mine = valid;
	for (int p = 0; p < top; ++p) {
		int c = 0;
#pragma unroll
		for (int r = 0; r < ROWS; ++r)
			c += __builtin_popcountll(ballot64(m[r] < (1u << p)));
		mine = lane == p ? c : mine;
	}
	if (lane < NCUM)
		w.cum[((long)plane * w.NT + tile) * NCUM + lane] = (unsigned short)mine;
	// planes = 1 + ilog2(max |v|) (encode.c:130), over the detail rings only (encode.c:165)
	// 100k waves hammering one word per plane would serialise in L2: the value only grows, so a
	// (possibly stale) plain read filters out all but the first few
	if (lane == 0 && top > __hip_atomic_load(w.planes_dev + plane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
		atomicMax(w.planes_dev + plane, top);
}

// ------------------------------------------------------------------ k_plan ---

struct HdrWriter {
	unsigned *w;
	long cap_words;
	unsigned long long acc;
	int n;
	long pos;
	int order;
	// What the reference's own counters would say if CAPACITY cut into this part of the stream: a refused
	// byte makes write_bits()/put_vli() give up the field they are writing (bits.h:58-78, vli.h:67-84, the
	// order then stays as it was) while encode_root() carries on with the next value (encode.c:97-110).
	// Only the statistics lines need this; the bytes are the prefix of the unlimited stream either way.
	long rc_cap, rc_len;
	int rc_n, rc_order;
	__device__ bool rc_bits(int nb)   // false: a byte was refused
	{
		if (nb <= 0)
			return true;
		rc_n += nb;
		while (rc_n >= 8) {
			if (rc_cap > 0 && rc_len >= rc_cap) {
				rc_n = 0;
				return false;
			}
			++rc_len;
			rc_n -= 8;
		}
		return true;
	}
	__device__ void rc_vli(unsigned v)
	{
		const int top = vli_top(rc_order, v);
		if (!rc_bits(top - rc_order) || !rc_bits(1) || !rc_bits(top))
			return;
		rc_order = vli_next(top);
	}
	__device__ unsigned rc_count() const { return (unsigned)(rc_len * 8 + rc_n); }
	__device__ void put(unsigned v, int nb)
	{
		rc_bits(nb);
		put_raw(v, nb);
	}
	__device__ void put_raw(unsigned v, int nb)
	{
		if (nb <= 0)
			return;
		acc |= (unsigned long long)(nb < 32 ? v & ((1u << nb) - 1u) : v) << n;
		n += nb;
		while (n >= 32) {
			if (pos < cap_words)
				w[pos] = (unsigned)acc;
			++pos;
			acc >>= 32;
			n -= 32;
		}
	}
	__device__ void vli(unsigned v)
	{
		rc_vli(v);
		const int top = vli_top(order, v);
		put_raw(0, top - order);
		put_raw(1, 1);
		put_raw(v + (1u << order) - (1u << top), top);
		order = vli_next(top);
	}
	__device__ unsigned bits() const { return (unsigned)(pos * 32 + n); }
};

__global__ void k_plan(PackGeom g, const int *__restrict__ lin, Work w, unsigned *out, long out_words, long capacity)
{
	if (threadIdx.x)
		return;
	const int img = blockIdx.x;
	ImgInfo &I = w.info[img];
	int planes[3] = { 0, 0, 0 };
	int pmax = 0;
	for (int c = 0; c < g.C; ++c) {
		planes[c] = w.planes_dev[img * g.C + c];
		pmax = planes[c] > pmax ? planes[c] : pmax;
		I.planes[c] = planes[c];
	}
	I.pmax = pmax;
	I.error = pmax > MAX_PLANES ? 1 : 0;

	HdrWriter hw;
	hw.w = out + img * out_words;
	hw.cap_words = out_words;
	hw.acc = 0;
	hw.n = 0;
	hw.pos = 0;
	hw.order = 0;
	hw.rc_cap = capacity;
	hw.rc_len = 0;
	hw.rc_n = 0;
	hw.rc_order = 0;
	// encode.c:169-172 header bytes
	hw.put('W', 8);
	hw.put(g.C == 3 ? '6' : '5', 8);
	hw.put((unsigned)(g.W - 1) & 0xffffu, 16);
	hw.put((unsigned)(g.H - 1) & 0xffffu, 16);
	I.meta_bits = hw.rc_count();   // encode.c:175-176
	// encode.c:97-110 root image per channel
	for (int c = 0; c < g.C; ++c) {
		const int *r = lin + (long)(img * g.C + c) * g.total;
		unsigned mx = 0;
		for (int i = 0; i < g.pixels[0]; ++i) {
			const int v = r[i];
			const unsigned a = (unsigned)(v < 0 ? -v : v);
			mx = a > mx ? a : mx;
		}
		const int cnt = mx ? ilog2u(mx) + 1 : 0;
		hw.vli((unsigned)cnt);
		if (cnt)
			for (int i = 0; i < g.pixels[0]; ++i) {
				const int v = r[i];
				hw.put((unsigned)(v < 0 ? -v : v), cnt);
				if (v)
					hw.put(v < 0, 1);
			}
	}
	I.root_bits = hw.rc_count() - I.meta_bits;   // encode.c:179-180
	for (int c = 0; c < g.C; ++c)   // encode.c:181-182
		hw.vli((unsigned)planes[c]);
	I.hdr_bits = hw.bits();
	I.order0 = hw.order;
	if (hw.n && hw.pos < hw.cap_words)
		hw.w[hw.pos] = (unsigned)hw.acc;

	// encode.c:183-221 schedule.  A flat image (planes all 0) still codes luma
	// level 0 at "plane -1" (SURVEY §5.9-2): all symbols are zero, which is what
	// plane 0 of an all-zero ring yields, so p is clamped to 0 there.
	int *sd = w.seg_desc + (long)img * MAX_SEGS;
	int *eb = w.seg_ebase + (long)img * (MAX_SEGS + 1);
	int *sx = w.segidx + (long)img * 48 * MAX_PLANES;
	for (int i = 0; i < 48 * MAX_PLANES; ++i)
		sx[i] = 0;
	int K = 0, E = 0;
	auto add = [&](int c, int l, int p) {
		if (K >= MAX_SEGS)
			return;
		sd[K] = c | (l << 4) | ((p < 0 ? 0 : p) + 1) << 8;
		sx[(c * 16 + l) * MAX_PLANES + (p < 0 ? 0 : p)] = K + 1;
		eb[K] = E;
		E += g.tile_first[l + 1] - g.tile_first[l];
		++K;
	};
	const int levels = g.levels;
	const int layers_max = 2 * (levels > pmax ? levels : pmax) - 1;
	if (pmax == planes[0])
		add(0, 0, planes[0] - 1);
	for (int layer = 0; layer < layers_max; ++layer) {
		for (int l = 0; l < levels && l <= layer + 1; ++l) {
			const int p = pmax - 1 - (layer + 1 - l);
			if (p >= 0 && p < planes[0])
				add(0, l, p);
		}
		for (int l = 0; l < levels && l <= layer; ++l) {
			const int p = pmax - 1 - (layer - l);
			for (int c = 1; c < g.C; ++c)
				if (p >= 0 && p < planes[c])
					add(c, l, p);
		}
	}
	eb[K] = E;
	I.K = K;
	I.E = E;
}

// --------------------------------------------------------------- k_entries ---

__device__ __forceinline__ int seg_of_entry(const int *eb, int K, int e)
{
	int lo = 0, hi = K - 1;   // largest k with eb[k] <= e
	while (lo < hi) {
		const int mid = (lo + hi + 1) >> 1;
		if (eb[mid] <= e)
			lo = mid;
		else
			hi = mid - 1;
	}
	return lo;
}

// block-wide exclusive scan of one unsigned per thread (1024 threads); returns total in `total`
__device__ __forceinline__ unsigned block_excl_scan(unsigned v, unsigned *wsum, unsigned &total)
{
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	unsigned inc = v;
	for (int o = 1; o < 64; o <<= 1) {
		const unsigned t = __shfl_up(inc, o);
		if (lane >= o)
			inc += t;
	}
	if (lane == 63)
		wsum[wv] = inc;
	__syncthreads();
	unsigned woff = 0, all = 0;
	for (int k = 0; k < 16; ++k) {
		const unsigned s = wsum[k];
		woff += k < wv ? s : 0u;
		all += s;
	}
	__syncthreads();
	total = all;
	return woff + inc - v;
}

// k_entries_count: one thread per entry, symbol counts from the histograms and the block-local
// exclusive prefixes of token slots / refinement bits; k_entries_blocks scans the block totals of
// each image; k_entries_finish adds the block offsets; k_entries_segs derives the per-segment values.

constexpr int ENT_BLOCK = 1024;

__global__ __launch_bounds__(ENT_BLOCK) void k_entries_count(PackGeom g, Work w)
{
	__shared__ unsigned wsum[16];
	const int img = blockIdx.y;
	const ImgInfo &I = w.info[img];
	const int K = I.K, E = I.E;
	if ((int)blockIdx.x * ENT_BLOCK >= E)
		return;
	const int *sd = w.seg_desc + (long)img * MAX_SEGS;
	const int *eb = w.seg_ebase + (long)img * (MAX_SEGS + 1);
	const int e = blockIdx.x * ENT_BLOCK + threadIdx.x;
	unsigned nt = 0, nr = 0;
	if (e < E) {
		const int k = seg_of_entry(eb, K, e);
		int c, l, p;
		seg_unpack(sd[k], c, l, p);
		const int j = e - eb[k];
		const int ntile = g.tile_first[l + 1] - g.tile_first[l];
		const long ring = (long)g.pixels[l + 1] - g.pixels[l];
		const long left = ring - (long)j * TILE;
		const int cnt = left < TILE ? (int)left : TILE;
		const unsigned short *cum = w.cum + ((long)(img * g.C + c) * w.NT + g.tile_first[l] + j) * NCUM;
		const int z = cum[p], upto = cum[p + 1];
		w.ent_zeros[img * w.ES + e] = (unsigned short)z;
		w.ent_ones[img * w.ES + e] = (unsigned short)(upto - z);
		w.ent_refs[img * w.ES + e] = (unsigned short)(cnt - upto);
		nt = (unsigned)(upto - z) + (j == ntile - 1 ? 1u : 0u);   // + the segment's break slot
		nr = (unsigned)(cnt - upto);
	}
	unsigned tt, rt;
	const unsigned tb = block_excl_scan(nt, wsum, tt);
	const unsigned rb = block_excl_scan(nr, wsum, rt);
	if (e < E) {
		w.ent_tokbase[img * (w.ES + 1) + e] = tb;
		w.ent_refscum[img * (w.ES + 1) + e] = rb;
	}
	if (threadIdx.x == 0) {
		w.ent_blk[(img * w.NCB + blockIdx.x) * 2] = tt;
		w.ent_blk[(img * w.NCB + blockIdx.x) * 2 + 1] = rt;
	}
}

__global__ __launch_bounds__(ENT_BLOCK) void k_entries_blocks(Work w)
{
	__shared__ unsigned wsum[16];
	const int img = blockIdx.x;
	ImgInfo &I = w.info[img];
	const int E = I.E;
	const int nb = (E + ENT_BLOCK - 1) / ENT_BLOCK;
	unsigned tok_run = 0, ref_run = 0;
	for (int b0 = 0; b0 < nb; b0 += ENT_BLOCK) {
		const int b = b0 + threadIdx.x;
		unsigned *slot = w.ent_blk + (img * w.NCB + b) * 2;
		const unsigned t = b < nb ? slot[0] : 0u, r = b < nb ? slot[1] : 0u;
		unsigned tt, rt;
		const unsigned tb = block_excl_scan(t, wsum, tt);
		const unsigned rb = block_excl_scan(r, wsum, rt);
		if (b < nb) {
			slot[0] = tok_run + tb;
			slot[1] = ref_run + rb;
		}
		tok_run += tt;
		ref_run += rt;
	}
	if (threadIdx.x == 0) {
		w.ent_tokbase[img * (w.ES + 1) + E] = tok_run;
		w.ent_refscum[img * (w.ES + 1) + E] = ref_run;
		I.T = tok_run + 1;   // + final flush (encode.c:221)
	}
}

__global__ __launch_bounds__(ENT_BLOCK) void k_entries_finish(Work w)
{
	const int img = blockIdx.y;
	const int e = blockIdx.x * ENT_BLOCK + threadIdx.x;
	if (e >= w.info[img].E)
		return;
	const unsigned *slot = w.ent_blk + (img * w.NCB + blockIdx.x) * 2;
	w.ent_tokbase[img * (w.ES + 1) + e] += slot[0];
	w.ent_refscum[img * (w.ES + 1) + e] += slot[1];
}

__global__ __launch_bounds__(ENT_BLOCK) void k_entries_segs(Work w)
{
	const int img = blockIdx.x;
	const int K = w.info[img].K;
	const int *eb = w.seg_ebase + (long)img * (MAX_SEGS + 1);
	const unsigned *tokbase = w.ent_tokbase + img * (w.ES + 1), *refscum = w.ent_refscum + img * (w.ES + 1);
	for (int k = threadIdx.x; k < K; k += ENT_BLOCK) {
		w.seg_refs[(long)img * MAX_SEGS + k] = refscum[eb[k + 1]] - refscum[eb[k]];
		const int last = eb[k + 1] - 1;
		w.brk_tok[(long)img * MAX_SEGS + k] = tokbase[last] + w.ent_ones[img * w.ES + last];
	}
}

// ---------------------------------------------------------------- k_tokens ---
// One wave per 1024-coefficient tile, all planes: the tile is read once into
// registers, then every plane that codes it classifies the same 16 rows.

__global__ __launch_bounds__(256) void k_tokens(PackGeom g, const int *__restrict__ lin, Work w)
{
	const int lane = threadIdx.x & 63;
	const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
	const int plane = blockIdx.y;
	if (tile >= w.NT)
		return;
	const int img = plane / g.C, c = plane - img * g.C;
	const ImgInfo &I = w.info[img];
	int l = 0;
	while (l + 1 < g.levels && tile >= g.tile_first[l + 1])
		++l;
	const int j = tile - g.tile_first[l];
	const long ring1 = g.pixels[l + 1];
	const long base = g.pixels[l] + (long)j * TILE;
	const int *src = lin + (long)plane * g.total;
	unsigned m[ROWS];       // magnitude, bit 31 = negative
	bool in[ROWS];
#pragma unroll
	for (int r = 0; r < ROWS; ++r) {
		const long i = base + r * 64 + lane;
		in[r] = i < ring1;
		const int v = in[r] ? src[i] : 0;
		m[r] = (unsigned)(v < 0 ? -v : v) | (v < 0 ? 0x80000000u : 0u);
	}
	const int *eb = w.seg_ebase + (long)img * (MAX_SEGS + 1);
	const int *sx = w.segidx + ((long)img * 48 + c * 16 + l) * MAX_PLANES;
	unsigned *tok_run = w.tok_run + img * w.TS;
	unsigned char *tok_flag = w.tok_flag + img * w.TS;
	const int pstart = I.planes[c] > 0 ? I.planes[c] - 1 : 0;
	// zeros before each one of the current row, by rank of the one: the run of a one is the difference
	// to its predecessor's count (LDS hand-over instead of per-lane 64-bit mask arithmetic)
	__shared__ unsigned zbefore[4][64];
	unsigned *zb = zbefore[threadIdx.x >> 6];
	for (int p = pstart; p >= 0; --p) {
		const int k1 = sx[p];
		if (!k1)
			continue;
		const int e = eb[k1 - 1] + j;
		if (!w.ent_ones[img * w.ES + e]) {   // no token from this tile at this plane: its zeros just pass through
			if (lane == 0)
				w.ent_tz[img * w.ES + e] = w.ent_zeros[img * w.ES + e];
			continue;
		}
		unsigned tb = w.ent_tokbase[img * (w.ES + 1) + e];
		unsigned pending = 0;   // zeros since the last one of this tile (uniform)
#pragma unroll
		for (int r = 0; r < ROWS; ++r) {
			const unsigned mag = m[r] & 0x7fffffffu;
			const bool refine = (mag >> (p + 1)) != 0;
			const bool one = in[r] && !refine && ((mag >> p) & 1u);
			const bool zero = in[r] && !refine && !one;
			const unsigned long long om = ballot64(one), zm = ballot64(zero);
			if (om) {
				const unsigned z = (unsigned)popc_below(zm), k = (unsigned)popc_below(om);
				if (one)
					zb[k] = z;
				__builtin_amdgcn_wave_barrier();
				if (one) {
					const unsigned run = k ? z - zb[k - 1] : pending + z;
					tok_run[tb + k] = run;
					tok_flag[tb + k] = (unsigned char)(F_HAS_SIGN | ((m[r] >> 31) ? F_SIGN : 0));
				}
				__builtin_amdgcn_wave_barrier();
				const int last = 63 - __builtin_clzll(om);
				pending = last == 63 ? 0u : (unsigned)__builtin_popcountll(zm >> (last + 1));
				tb += (unsigned)__builtin_popcountll(om);
			} else {
				pending += (unsigned)__builtin_popcountll(zm);
			}
		}
		if (lane == 0)
			w.ent_tz[img * w.ES + e] = (unsigned short)pending;
	}
}

// ----------------------------------------------------------------- k_carry ---
// Pending-run state s across entries: a tile maps s -> (has_one ? tz : s + tz);
// a segment end with refinement bits emits s as a phantom terminator (if s > 0)
// and resets it (rle.h:79-89); without refinement bits the run carries on.
// Maps are (keep, add): s -> add + (keep ? s : 0).

__device__ __forceinline__ RunMap compose(RunMap a, RunMap b)   // a then b
{
	RunMap r;
	r.keep = a.keep & b.keep;
	r.add = b.keep ? a.add + b.add : b.add;
	return r;
}

// The scan runs over all entries of a batch at once: k_carry_local reduces blocks of 1024
// entries to one map each, k_carry_blocks scans those per image (and writes the final flush
// token), k_carry_apply redoes the block-local scan from the block's entry state and patches
// the tokens.

constexpr int CARRY_BLOCK = ENT_BLOCK;   // both scans cut the entries into the same blocks (Work::NCB)

__device__ __forceinline__ RunMap carry_map_of(const Work &w, const ImgInfo &I, int img, int e, bool &has_one, bool &seg_end, bool &refs)
{
	RunMap t = { 1u, 0u };
	has_one = seg_end = refs = false;
	if (e < I.E) {
		const int *eb = w.seg_ebase + (long)img * (MAX_SEGS + 1);
		has_one = w.ent_ones[img * w.ES + e] != 0;
		t.keep = has_one ? 0u : 1u;
		t.add = w.ent_tz[img * w.ES + e];
		const int k = seg_of_entry(eb, I.K, e);
		seg_end = e == eb[k + 1] - 1;
		refs = seg_end && w.seg_refs[(long)img * MAX_SEGS + k] != 0;
	}
	return t;
}

// inclusive scan of one map per thread over a block of 1024; wagg = 16 maps of LDS
__device__ __forceinline__ RunMap block_scan_maps(RunMap m, RunMap *wagg, RunMap &total)
{
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	for (int o = 1; o < 64; o <<= 1) {
		RunMap a;
		a.keep = __shfl_up(m.keep, o);
		a.add = __shfl_up(m.add, o);
		if (lane >= o)
			m = compose(a, m);
	}
	if (lane == 63)
		wagg[wv] = m;
	__syncthreads();
	RunMap pre = { 1u, 0u }, all = { 1u, 0u };
	for (int k = 0; k < 16; ++k) {
		const RunMap a = wagg[k];
		if (k < wv)
			pre = compose(pre, a);
		all = compose(all, a);
	}
	__syncthreads();
	total = all;
	return compose(pre, m);
}

__global__ __launch_bounds__(CARRY_BLOCK) void k_carry_local(Work w)
{
	__shared__ RunMap wagg[16];
	const int img = blockIdx.y;
	const ImgInfo &I = w.info[img];
	if ((int)blockIdx.x * CARRY_BLOCK >= I.E)
		return;
	const int e = blockIdx.x * CARRY_BLOCK + threadIdx.x;
	bool has_one, seg_end, refs;
	RunMap t = carry_map_of(w, I, img, e, has_one, seg_end, refs);
	if (refs)
		t = RunMap{ 0u, 0u };   // the break slot takes the pending run, the next segment starts from 0
	RunMap total;
	block_scan_maps(t, wagg, total);
	if (threadIdx.x == 0)
		w.carry_agg[img * w.NCB + blockIdx.x] = total;
}

__global__ __launch_bounds__(CARRY_BLOCK) void k_carry_blocks(Work w)
{
	__shared__ RunMap wagg[16];
	const int img = blockIdx.x;
	const ImgInfo &I = w.info[img];
	const int nb = (I.E + CARRY_BLOCK - 1) / CARRY_BLOCK;
	unsigned s = 0;   // state 0 at stream start (rle.h:33)
	for (int b0 = 0; b0 < nb; b0 += CARRY_BLOCK) {
		const int b = b0 + threadIdx.x;
		const RunMap mine = b < nb ? w.carry_agg[img * w.NCB + b] : RunMap{ 1u, 0u };
		RunMap total;
		const RunMap inc = block_scan_maps(mine, wagg, total);
		// state entering block b = everything before it applied to s: inclusive minus own = shift by one
		RunMap ex;
		ex.keep = __shfl_up(inc.keep, 1);
		ex.add = __shfl_up(inc.add, 1);
		__shared__ RunMap edge[16];
		if ((threadIdx.x & 63) == 63)
			edge[threadIdx.x >> 6] = inc;
		__syncthreads();
		if ((threadIdx.x & 63) == 0)
			ex = threadIdx.x ? edge[(threadIdx.x >> 6) - 1] : RunMap{ 1u, 0u };
		if (b < nb)
			w.carry_in[img * w.NCB + b] = ex.add + (ex.keep ? s : 0u);
		s = total.add + (total.keep ? s : 0u);
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		w.tok_run[img * w.TS + I.T - 1] = s;                 // encode.c:221 rle_flush: always emitted
		w.tok_flag[img * w.TS + I.T - 1] = (unsigned char)F_FLUSH;
	}
}

__global__ __launch_bounds__(CARRY_BLOCK) void k_carry_apply(Work w)
{
	__shared__ RunMap wagg[16];
	const int img = blockIdx.y;
	const ImgInfo &I = w.info[img];
	if ((int)blockIdx.x * CARRY_BLOCK >= I.E)
		return;
	const int e = blockIdx.x * CARRY_BLOCK + threadIdx.x;
	bool has_one, seg_end, refs;
	const RunMap own = carry_map_of(w, I, img, e, has_one, seg_end, refs);
	const RunMap t = refs ? RunMap{ 0u, 0u } : own;
	RunMap total;
	const RunMap inc = block_scan_maps(t, wagg, total);
	RunMap ex;
	ex.keep = __shfl_up(inc.keep, 1);
	ex.add = __shfl_up(inc.add, 1);
	__shared__ RunMap edge[16];
	if ((threadIdx.x & 63) == 63)
		edge[threadIdx.x >> 6] = inc;
	__syncthreads();
	if ((threadIdx.x & 63) == 0)
		ex = threadIdx.x ? edge[(threadIdx.x >> 6) - 1] : RunMap{ 1u, 0u };
	if (e >= I.E)
		return;
	const unsigned s_blk = w.carry_in[img * w.NCB + blockIdx.x];
	const unsigned s_in = ex.add + (ex.keep ? s_blk : 0u);
	unsigned *tok_run = w.tok_run + img * w.TS;
	const unsigned tb = w.ent_tokbase[img * (w.ES + 1) + e];
	if (has_one)
		tok_run[tb] += s_in;
	if (seg_end) {   // the break slot
		const unsigned s = own.add + (own.keep ? s_in : 0u);
		const unsigned idx = tb + w.ent_ones[img * w.ES + e];
		tok_run[idx] = s;
		w.tok_flag[img * w.TS + idx] = (unsigned char)(F_BREAK | ((refs && s) ? 0u : F_VOID));
	}
}

// ------------------------------------------------------------------- k_lut ---
// Lanes 0..31 of each half-wave are the 32 possible VLI orders at the start of
// a 4096-token chunk; the half-wave walks the chunk once and every lane follows
// its own start state.  Snapshots at every 64-token boundary (sublut) let the
// next pass start each lane's 64 tokens from the right order.

__device__ __forceinline__ int vli_step(int o, unsigned v, bool skip)
{
	const int nx = vli_next(vli_top(o, v));
	return skip ? o : nx;
}

__device__ __forceinline__ void lut_body(const Work &w, long vbx, int img)
{
	const int lane = threadIdx.x & 63, half = lane >> 5, s = lane & 31;
	const long chunk = (vbx * 4 + (threadIdx.x >> 6)) * 2 + half;
	const unsigned T = w.info[img].T;
	const long nchunks = ((long)T + CHUNK - 1) / CHUNK;
	const long chunk_a = chunk - half;          // the wave's two chunks: a (lanes 0-31), a+1 (lanes 32-63)
	if (chunk_a >= nchunks)
		return;
	const bool live = chunk < nchunks;
	const unsigned *run = w.tok_run + img * w.TS;
	const unsigned char *flag = w.tok_flag + img * w.TS;
	unsigned char *sub = w.sublut + (img * w.NCS + chunk) * 64 * 32;
	// lane i holds token i of a 64-token row for both chunks; bit 31 = "void" (runs are < 2^31).
	// Tokens are broadcast with v_readlane (no LDS traffic), each half picks its own chunk's.
	// The next row is loaded while the current one is walked.
	auto fetch = [&](int q, unsigned &ra, unsigned &rb) {
		const long ta = chunk_a * CHUNK + q * SUB + lane, tb = ta + CHUNK;
		ra = ta < (long)T ? run[ta] | ((flag[ta] & F_VOID) ? 0x80000000u : 0u) : 0x80000000u;
		rb = tb < (long)T ? run[tb] | ((flag[tb] & F_VOID) ? 0x80000000u : 0u) : 0x80000000u;
	};
	int o = s;
	unsigned na, nb;
	fetch(0, na, nb);
	for (int q = 0; q < 64; ++q) {
		if (live)
			sub[q * 32 + s] = (unsigned char)o;
		const unsigned ra = na, rb = nb;
		if (q + 1 < 64)
			fetch(q + 1, na, nb);
#pragma unroll
		for (int t = 0; t < 64; ++t) {
			const unsigned a = __builtin_amdgcn_readlane(ra, t), b = __builtin_amdgcn_readlane(rb, t);
			const unsigned v = half ? b : a;
			o = vli_step(o, v & 0x7fffffffu, v >> 31);
		}
	}
	if (live)
		w.lut[(img * w.NCS + chunk) * 32 + s] = (unsigned char)o;
}

// The exact pass only runs for images the fast pass flagged: a small fixed grid that
// returns at once otherwise and strides over the virtual blocks when it has work.
__global__ __launch_bounds__(256) void k_lut(Work w)
{
	const int img = blockIdx.y;
	if (!w.slow[img])
		return;
	const long nvb = (w.NCS + 7) / 8;
	for (long vb = blockIdx.x; vb < nvb; vb += gridDim.x)
		lut_body(w, vb, img);
}

// group maps: 32 lanes (states) walk the 64 chunk maps of a group
__device__ __forceinline__ void chain_groups_body(const Work &w, long vbx, int img)
{
	const int lane = threadIdx.x & 63, half = lane >> 5, s = lane & 31;
	const long group = (vbx * 4 + (threadIdx.x >> 6)) * 2 + half;
	const unsigned T = w.info[img].T;
	const long nchunks = ((long)T + CHUNK - 1) / CHUNK;
	const long ngroups = (nchunks + GROUP - 1) / GROUP;
	if (group >= ngroups)
		return;
	const unsigned char *lut = w.lut + img * w.NCS * 32;
	int o = s;
	for (long c = group * GROUP; c < min((group + 1) * GROUP, nchunks); ++c)
		o = lut[c * 32 + o];
	w.glut[(img * w.NGS + group) * 32 + s] = (unsigned char)o;
}

__global__ __launch_bounds__(256) void k_chain_groups(Work w)
{
	const int img = blockIdx.y;
	if (!w.slow[img])
		return;
	const long nvb = (w.NGS + 7) / 8;
	for (long vb = blockIdx.x; vb < nvb; vb += gridDim.x)
		chain_groups_body(w, vb, img);
}

// per image: serial over groups, then every group's chunks in parallel
__global__ __launch_bounds__(1024) void k_chain_image(Work w)
{
	const int img = blockIdx.x;
	if (!w.slow[img])
		return;
	const ImgInfo &I = w.info[img];
	const long nchunks = ((long)I.T + CHUNK - 1) / CHUNK;
	const long ngroups = (nchunks + GROUP - 1) / GROUP;
	unsigned char *gentry = w.group_entry + img * w.NGS;
	if (threadIdx.x == 0) {
		int o = I.order0;
		for (long gq = 0; gq < ngroups; ++gq) {
			gentry[gq] = (unsigned char)o;
			o = w.glut[(img * w.NGS + gq) * 32 + o];
		}
	}
	__syncthreads();
	__threadfence_block();
	const unsigned char *lut = w.lut + img * w.NCS * 32;
	unsigned char *centry = w.chunk_entry + img * w.NCS;
	for (long gq = threadIdx.x; gq < ngroups; gq += 1024) {
		int o = gentry[gq];
		for (long c = gq * GROUP; c < min((gq + 1) * GROUP, nchunks); ++c) {
			centry[c] = (unsigned char)o;
			o = lut[c * 32 + o];
		}
	}
}

// ---------------------------------------------------------------- k_orders ---
// One lane per 64 tokens, now with the true start order: record each token's
// order and the bits it will occupy; wave-scan the lane totals.

__device__ __forceinline__ long find_break_seg(const unsigned *btok, int K, unsigned t)
{
	int lo = 0, hi = K - 1;
	while (lo < hi) {
		const int mid = (lo + hi) >> 1;
		if (btok[mid] < t)
			lo = mid + 1;
		else
			hi = mid;
	}
	return lo;
}

__device__ __forceinline__ unsigned long long wave_excl_scan64(unsigned long long v, unsigned long long &total)
{
	const int lane = lane_id();
	unsigned long long inc = v;
	for (int o = 1; o < 64; o <<= 1) {
		const unsigned long long t = __shfl_up(inc, o);
		if (lane >= o)
			inc += t;
	}
	total = __shfl(inc, 63);
	return inc - v;
}

// The token arrays are read as coalesced 64-token rows and transposed through LDS
// (row pitch 65 words: lane j then reads [j][t] conflict-free), because lane j
// needs the 64 CONSECUTIVE tokens j*64 .. j*64+63 of the wave's 4096-token chunk.
// Per token it records the VLI order it is coded with and its bit offset inside
// the lane's 64 tokens (refinement blocks not counted); k_emit then writes the
// tokens in parallel.
constexpr int ORD_WAVES = 2;

struct OrdTile {
	unsigned run[64][65];
	unsigned char flag[64][68];
	unsigned char ord[64][68];
	unsigned short off[64][66];
};

__device__ __forceinline__ void orders_body(const Work &w, OrdTile *tiles, long vbx, int img)
{
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const long chunk = vbx * ORD_WAVES + wv;
	const ImgInfo &I = w.info[img];
	const unsigned T = I.T;
	const long nchunks = ((long)T + CHUNK - 1) / CHUNK;
	if (chunk >= nchunks)
		return;
	const unsigned *run = w.tok_run + img * w.TS;
	const unsigned char *flag = w.tok_flag + img * w.TS;
	const unsigned *srefs = w.seg_refs + (long)img * MAX_SEGS;
	const unsigned *btok = w.brk_tok + (long)img * MAX_SEGS;
	OrdTile &tile = tiles[wv];
	const long tbase = chunk * CHUNK;
#pragma unroll 8
	for (int r = 0; r < 64; ++r) {
		const long t = tbase + r * 64 + lane;
		const bool in = t < (long)T;
		tile.run[r][lane] = in ? run[t] : 0u;
		tile.flag[r][lane] = in ? flag[t] : (unsigned char)F_VOID;
	}
	__builtin_amdgcn_wave_barrier();   // each wave only reads back its own tile

	// this lane's 64 tokens start at the order recorded for (chunk entry state, sub-chunk)
	int o = w.sublut[((img * w.NCS + chunk) * 64 + lane) * 32 + w.chunk_entry[img * w.NCS + chunk]];
	const long t0 = tbase + (long)lane * SUB;
	unsigned tokbits = 0;              // bits of this lane's tokens so far
	unsigned long long rawbits = 0;    // refinement blocks that follow break tokens of this lane
	for (int t = 0; t < SUB; ++t) {
		const unsigned f = tile.flag[lane][t];
		tile.ord[lane][t] = (unsigned char)o;
		tile.off[lane][t] = (unsigned short)tokbits;
		if (!(f & F_VOID)) {
			const int top = vli_top(o, tile.run[lane][t]);
			tokbits += (unsigned)(2 * top - o + 1) + ((f & F_HAS_SIGN) ? 1u : 0u);
			o = vli_next(top);
		}
		if (f & F_BREAK)
			rawbits += srefs[find_break_seg(btok, I.K, (unsigned)(t0 + t))];
	}
	unsigned long long total;
	const unsigned long long pre = wave_excl_scan64(tokbits + rawbits, total);
	w.lane_bits[(img * w.NCS + chunk) * 64 + lane] = pre;
	if (lane == 0)
		w.chunk_bits[img * w.NCS + chunk] = total;
	__builtin_amdgcn_wave_barrier();
	unsigned char *ord = w.tok_ord + img * w.TS;
	unsigned short *off = w.tok_off + img * w.TS;
#pragma unroll 8
	for (int r = 0; r < 64; ++r) {
		const long t = tbase + r * 64 + lane;
		if (t < (long)T) {
			ord[t] = tile.ord[r][lane];
			off[t] = tile.off[r][lane];
		}
	}
	__builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(64 * ORD_WAVES) void k_orders(Work w)
{
	__shared__ OrdTile tiles[ORD_WAVES];
	const int img = blockIdx.y;
	if (!w.slow[img])
		return;
	const long nvb = (w.NCS + ORD_WAVES - 1) / ORD_WAVES;
	for (long vb = blockIdx.x; vb < nvb; vb += gridDim.x)
		orders_body(w, tiles, vb, img);
}

// ----------------------------------------------------------- k_orders_fast ---
// The common case needs no 32-state machinery.  The order map of a token is
// monotone, so every start state ends between the chains started at 0 and at
// 31; over 64 tokens those two almost always meet (orders decay by 2 per small
// value), and then the group's exit order is a constant, whatever it was
// entered with.  One lane per 64-token group: walk the 0- and the 31-chain; if
// they met, the NEXT group's entry order is known, and a second walk records
// each token's order and bit offset.  Lane 0 only serves as the predecessor of
// lane 1 (63 groups of output per wave).  If any pair of chains did not meet
// the image is flagged and the exact hierarchical pass (k_lut ... k_orders)
// redoes it.  Tokens go through LDS in tiles of QT per group, so that several
// waves fit a SIMD.
constexpr int FSUBS = 63;
constexpr int QT = 32;                    // tokens per group staged at a time (128-byte rows; 8 -> 1440, 16 -> 970, 32 -> 870, 64 -> 1380 us per 16 frames)
constexpr int QLANES = QT / 4;            // lanes that move one group's QT tokens (four each)
constexpr int QSUBS = 64 / QLANES;        // groups per wave instruction

struct FastTile {
	unsigned run[64][QT + 1];
	unsigned char flag[64][QT + 4];
	unsigned char ord[64][QT + 4];
	unsigned short off[64][QT + 2];
};

__global__ __launch_bounds__(256) void k_orders_fast(Work w)
{
	__shared__ FastTile tiles[4];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const long wave = (long)blockIdx.x * 4 + wv;
	const int img = blockIdx.y;
	const ImgInfo &I = w.info[img];
	const long T = I.T;
	const long nsub = (T + SUB - 1) / SUB;
	if (wave * FSUBS >= nsub)
		return;
	const unsigned *run = w.tok_run + img * w.TS;
	const unsigned char *flag = w.tok_flag + img * w.TS;
	const unsigned *srefs = w.seg_refs + (long)img * MAX_SEGS;
	const unsigned *btok = w.brk_tok + (long)img * MAX_SEGS;
	FastTile &tile = tiles[wv];
	const long S = wave * FSUBS - 1 + lane;          // this lane's 64-token group (lane 0: predecessor only)
	const long tfirst = (wave * FSUBS - 1) * SUB;    // first token of the wave's window (-64 for wave 0)
	// a lane moves four consecutive tokens of one group at a time: 16 bytes of runs, 4 bytes of flags
	const int vsub = lane / QLANES, q4 = (lane % QLANES) * 4;
	auto load_tile = [&](int qt) {
#pragma unroll
		for (int k = 0; k < 64 / QSUBS; ++k) {
			const int sub = k * QSUBS + vsub;
			const long t = tfirst + (long)sub * SUB + qt * QT + q4;   // multiple of 4, like every image's token base
			uint4 r = make_uint4(0u, 0u, 0u, 0u);
			unsigned f = F_VOID * 0x01010101u;
			if (t >= 0 && t < T) {
				r = *reinterpret_cast<const uint4 *>(run + t);
				f = *reinterpret_cast<const unsigned *>(flag + t);
				const long left = T - t;   // tokens past T are stale scratch
				if (left < 4) {
					r.w = 0u;
					r.z = left > 2 ? r.z : 0u;
					r.y = left > 1 ? r.y : 0u;
					const unsigned keep = left > 2 ? 0x00ffffffu : left > 1 ? 0x0000ffffu : 0x000000ffu;
					f = (f & keep) | ((F_VOID * 0x01010101u) & ~keep);
				}
			}
			tile.run[sub][q4] = r.x;
			tile.run[sub][q4 + 1] = r.y;
			tile.run[sub][q4 + 2] = r.z;
			tile.run[sub][q4 + 3] = r.w;
			*reinterpret_cast<unsigned *>(&tile.flag[sub][q4]) = f;
		}
	};
	int lo = 0, hi = 31;
	for (int qt = 0; qt < SUB / QT; ++qt) {
		load_tile(qt);
		__builtin_amdgcn_wave_barrier();
#pragma unroll
		for (int t = 0; t < QT; ++t) {
			const unsigned f = tile.flag[lane][t];
			const unsigned v = tile.run[lane][t];
			lo = vli_step(lo, v, f & F_VOID);
			hi = vli_step(hi, v, f & F_VOID);
		}
		__builtin_amdgcn_wave_barrier();
	}
	const bool valid = S >= 0 && S < nsub;
	// A group whose two chains met leaves in a known order.  The rare other ones are resolved
	// exactly as soon as their own entry order is known (one more walk, only in waves that have
	// such a group); only if that chain of knowledge breaks (e.g. at the wave's predecessor lane)
	// is the image handed to the exact pass.
	int exitv = lo;
	bool exit_known = lo == hi || !valid;
	int o = 0;
	bool entry_known = false;
	for (int it = 0; it < 4; ++it) {
		o = __shfl_up(exitv, 1);
		entry_known = __shfl_up((int)exit_known, 1) != 0 && lane >= 1;
		if (S == 0) {
			o = I.order0;
			entry_known = true;
		}
		const bool resolve = valid && !exit_known && entry_known;
		if (!ballot64(resolve))
			break;
		int e = o;
		for (int qt = 0; qt < SUB / QT; ++qt) {
			load_tile(qt);
			__builtin_amdgcn_wave_barrier();
#pragma unroll
			for (int t = 0; t < QT; ++t)
				e = vli_step(e, tile.run[lane][t], tile.flag[lane][t] & F_VOID);
			__builtin_amdgcn_wave_barrier();
		}
		if (resolve) {
			exitv = e;
			exit_known = true;
		}
	}
	if (ballot64(lane >= 1 && valid && !entry_known)) {
		if (lane == 0)
			atomicOr(w.slow + img, 1);
		return;   // the exact pass takes the whole image
	}
	const bool produces = lane >= 1 && valid;
	const long t0 = S * SUB;
	unsigned tokbits = 0;
	unsigned long long rawbits = 0;
	unsigned char *ord = w.tok_ord + img * w.TS;
	unsigned short *off = w.tok_off + img * w.TS;
	for (int qt = 0; qt < SUB / QT; ++qt) {
		load_tile(qt);
		__builtin_amdgcn_wave_barrier();
#pragma unroll
		for (int t = 0; t < QT; ++t) {
			const unsigned f = tile.flag[lane][t];
			tile.ord[lane][t] = (unsigned char)o;
			tile.off[lane][t] = (unsigned short)tokbits;
			if (!(f & F_VOID)) {
				const int top = vli_top(o, tile.run[lane][t]);
				tokbits += (unsigned)(2 * top - o + 1) + ((f & F_HAS_SIGN) ? 1u : 0u);
				o = vli_next(top);
			}
			if ((f & F_BREAK) && produces)
				rawbits += srefs[find_break_seg(btok, I.K, (unsigned)(t0 + qt * QT + t))];
		}
		__builtin_amdgcn_wave_barrier();
#pragma unroll
		for (int k = 0; k < 64 / QSUBS; ++k) {
			const int sub = k * QSUBS + vsub;
			const long t = tfirst + (long)sub * SUB + qt * QT + q4;
			if (sub < 1 || t >= T)
				continue;
			const unsigned o4 = *reinterpret_cast<const unsigned *>(&tile.ord[sub][q4]);
			const unsigned f01 = *reinterpret_cast<const unsigned *>(&tile.off[sub][q4]);
			const unsigned f23 = *reinterpret_cast<const unsigned *>(&tile.off[sub][q4 + 2]);
			if (t + 4 <= T) {
				*reinterpret_cast<unsigned *>(ord + t) = o4;
				*reinterpret_cast<uint2 *>(off + t) = make_uint2(f01, f23);
			} else {
				for (int e = 0; e < (int)(T - t); ++e) {
					ord[t + e] = tile.ord[sub][q4 + e];
					off[t + e] = tile.off[sub][q4 + e];
				}
			}
		}
		__builtin_amdgcn_wave_barrier();
	}
	unsigned long long total;
	const unsigned long long pre = wave_excl_scan64(produces ? tokbits + rawbits : 0ull, total);
	if (produces)
		w.lane_bits[img * w.NCS * 64 + S] = pre;
	if (lane == 0)
		w.chunk_bits[img * w.NCS + wave] = total;
}

// ------------------------------------------------------------------ k_emit ---
// bits.h:58-78: one lane per four consecutive tokens, one wave per 256.  A token's position is
// chunk base + its 64-token group's offset + its own offset (+ the refinement blocks of earlier
// break tokens of the same group).  The four codes of a lane are adjacent in the stream unless a
// break lies between them, so they are glued into one bit string first; the strings of a wave
// are merged in an LDS window, and each stream word costs one global atomic per wave.

constexpr int COMB = 192;      // words of the LDS window (typical: 256 tokens of ~4 bits = 32 words)
constexpr int EMIT_TOK = 4;    // tokens per lane

__global__ __launch_bounds__(256) void k_emit(Work w, unsigned *out, long out_words)
{
	__shared__ unsigned comb[4][COMB];
	const int lane = threadIdx.x & 63;
	const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
	const int img = blockIdx.y;
	const ImgInfo &I = w.info[img];
	const long T = I.T;
	const long t0 = wave * (64 * EMIT_TOK) + lane * EMIT_TOK;   // multiple of 4, like the token arrays' bases
	if (wave * (64 * EMIT_TOK) >= T)
		return;
	const unsigned *srefs = w.seg_refs + (long)img * MAX_SEGS;
	const unsigned *btok = w.brk_tok + (long)img * MAX_SEGS;
	unsigned f4 = F_VOID * 0x01010101u, o4 = 0;
	uint4 r4 = make_uint4(0u, 0u, 0u, 0u);
	uint2 off4 = make_uint2(0u, 0u);
	unsigned long long gpos = 0;
	if (t0 < T) {
		f4 = *reinterpret_cast<const unsigned *>(w.tok_flag + img * w.TS + t0);
		o4 = *reinterpret_cast<const unsigned *>(w.tok_ord + img * w.TS + t0);
		r4 = *reinterpret_cast<const uint4 *>(w.tok_run + img * w.TS + t0);
		off4 = *reinterpret_cast<const uint2 *>(w.tok_off + img * w.TS + t0);
		const long left = T - t0;   // tokens past T are stale scratch
		if (left < EMIT_TOK) {
			const unsigned keep = left > 2 ? 0x00ffffffu : left > 1 ? 0x0000ffffu : 0x000000ffu;
			f4 = (f4 & keep) | ((F_VOID * 0x01010101u) & ~keep);
		}
		const long group = t0 / SUB;
		const long chunk = group / (w.slow[img] ? 64 : FSUBS);
		gpos = w.chunk_base[img * w.NCS + chunk] + w.lane_bits[img * w.NCS * 64 + group];
	}
	const unsigned run[EMIT_TOK] = { r4.x, r4.y, r4.z, r4.w };
	unsigned long long pos[EMIT_TOK], code[EMIT_TOK];
	int len[EMIT_TOK];
#pragma unroll
	for (int e = 0; e < EMIT_TOK; ++e) {
		const unsigned f = (f4 >> (8 * e)) & 255u;
		pos[e] = gpos + (((e & 1) ? (e & 2 ? off4.y : off4.x) >> 16 : (e & 2 ? off4.y : off4.x)) & 0xffffu);
		code[e] = 0;
		len[e] = 0;
		if (!(f & F_VOID)) {
			const int o = (int)((o4 >> (8 * e)) & 255u);
			const unsigned v = run[e];
			const int top = vli_top(o, v);
			const int z = top - o;
			code[e] = (1ull << z) | ((unsigned long long)(v + (1u << o) - (1u << top)) << (z + 1));
			len[e] = z + 1 + top;
			if (f & F_HAS_SIGN) {
				code[e] |= (unsigned long long)(f & F_SIGN) << len[e];
				++len[e];
			}
		}
	}
	// break tokens are followed by their segment's refinement block: later tokens of the same 64-token
	// group move up by its size (k_orders_fast / k_orders counted it for the groups after that)
	const unsigned brk4 = f4 & (F_BREAK * 0x01010101u);
	bool split = false;   // a refinement block lies between this lane's tokens
	if (ballot64(brk4 != 0)) {
#pragma unroll
		for (int e = 0; e < EMIT_TOK; ++e) {
			unsigned long long bm = ballot64(((brk4 >> (8 * e)) & F_BREAK) != 0);
			while (bm) {
				const int j = __builtin_ctzll(bm);
				bm &= bm - 1;
				const long tb = wave * (64 * EMIT_TOK) + j * EMIT_TOK + e;   // the break token
				const long k = find_break_seg(btok, I.K, (unsigned)tb);
				const unsigned refs = srefs[k];
#pragma unroll
				for (int q = 0; q < EMIT_TOK; ++q) {
					const long tq = t0 + q;
					if (tq > tb && tq / SUB == tb / SUB) {
						pos[q] += refs;
						split = split || (refs != 0 && lane == j);
					}
				}
			}
		}
#pragma unroll
		for (int e = 0; e < EMIT_TOK; ++e)   // every shift is in: where each break's refinement block starts
			if ((brk4 >> (8 * e)) & F_BREAK)
				w.seg_rawoff[(long)img * MAX_SEGS + find_break_seg(btok, I.K, (unsigned)(t0 + e))] = pos[e] + (unsigned)len[e];
	}
	unsigned *cw = comb[threadIdx.x >> 6];
	for (int i = lane; i < COMB; i += 64)
		cw[i] = 0;
	const long wbase = (long)(__shfl(pos[0], 0) >> 5);
	__builtin_amdgcn_wave_barrier();
	unsigned *dst = out + img * out_words;
	auto put = [&](unsigned long long p, unsigned long long c) {   // up to 64 code bits at bit position p
		const long w0 = (long)(p >> 5);
		const int sh = (int)(p & 31);
		const unsigned long long lo = c << sh;
		const unsigned part[3] = { (unsigned)lo, (unsigned)(lo >> 32), sh ? (unsigned)(c >> (64 - sh)) : 0u };
#pragma unroll
		for (int q = 0; q < 3; ++q) {
			const long wi = w0 + q;
			if (!part[q])
				continue;
			if (wi - wbase < COMB)
				atomicOr(cw + (wi - wbase), part[q]);
			else if (wi < out_words)
				atomicOr(dst + wi, part[q]);
		}
	};
	const int total = len[0] + len[1] + len[2] + len[3];
	// tokens of one 64-token group follow each other bit for bit (a lane never straddles two groups)
	if (!split && total <= 64) {
		unsigned long long c = 0;
		int at = 0;
#pragma unroll
		for (int e = 0; e < EMIT_TOK; ++e) {
			c |= len[e] ? code[e] << at : 0ull;
			at += len[e];
		}
		if (total)
			put(pos[0], c);
	} else {
#pragma unroll
		for (int e = 0; e < EMIT_TOK; ++e)
			if (len[e])
				put(pos[e], code[e]);
	}
	__builtin_amdgcn_wave_barrier();
	for (int i = lane; i < COMB; i += 64) {
		const unsigned v = cw[i];
		if (v && wbase + i < out_words)
			atomicOr(dst + wbase + i, v);
	}
}

// per image: exclusive scan of chunk bit totals
__global__ __launch_bounds__(1024) void k_bitscan(Work w, long capacity)
{
	__shared__ unsigned long long wsum[16];
	__shared__ unsigned long long carry;
	const int img = blockIdx.x;
	ImgInfo &I = w.info[img];
	const long nsub = ((long)I.T + SUB - 1) / SUB;
	const long nchunks = w.slow[img] ? ((long)I.T + CHUNK - 1) / CHUNK : (nsub + FSUBS - 1) / FSUBS;
	const unsigned long long *cb = w.chunk_bits + img * w.NCS;
	unsigned long long *base = w.chunk_base + img * w.NCS;
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	if (threadIdx.x == 0)
		carry = I.hdr_bits;
	__syncthreads();
	for (long b0 = 0; b0 < nchunks; b0 += 1024) {
		const long i = b0 + threadIdx.x;
		const unsigned long long v = i < nchunks ? cb[i] : 0ull;
		unsigned long long inc = v;
		for (int o = 1; o < 64; o <<= 1) {
			const unsigned long long t = __shfl_up(inc, o);
			if (lane >= o)
				inc += t;
		}
		if (lane == 63)
			wsum[wv] = inc;
		__syncthreads();
		unsigned long long woff = 0, all = 0;
		for (int k = 0; k < 16; ++k) {
			const unsigned long long s = wsum[k];
			woff += k < wv ? s : 0ull;
			all += s;
		}
		const unsigned long long c = carry;
		if (i < nchunks)
			base[i] = c + woff + inc - v;
		__syncthreads();
		if (threadIdx.x == 0)
			carry = c + all;
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		const unsigned long long bits = carry;
		w.stream_bits[img] = bits;
		unsigned long long bytes = (bits + 7) >> 3;
		// bytes.h:75-78: nothing is written past `capacity` bytes; the stream is a prefix
		I.total_bits = bits;
		if (capacity > 0 && bytes > (unsigned long long)capacity) {
			bytes = (unsigned long long)capacity;
			// the encoder only notices the limit when a complete byte is refused
			// (bits.h:61-66); a refused final padding byte (bits.h:51-56) leaves the count alone
			if (bits >= 8ull * ((unsigned long long)capacity + 1))
				I.total_bits = bytes * 8;
		}
		I.nbytes = bytes;
		I.pad = w.slow[img];   // 1: the exact 32-state order pass had to run for this image
	}
}

// The token and refinement writers OR their bits into the stream, so it has to start as zeros — but only
// the words the stream will occupy (its length is known after k_bitscan), not the whole output stride;
// the words k_plan filled with header, root image and plane counts stay.
__global__ __launch_bounds__(256) void k_clear_stream(Work w, unsigned *out, long out_words)
{
	const int img = blockIdx.y;
	const long first = ((long)w.info[img].hdr_bits + 31) >> 5;
	long last = (long)((w.stream_bits[img] + 31) >> 5) + 4;
	last = last < out_words ? last : out_words;
	const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
	if (i >= last || i + 4 <= first)
		return;
	unsigned *dst = out + img * out_words + i;
	if (i >= first && i + 4 <= last) {
		*reinterpret_cast<uint4 *>(dst) = make_uint4(0u, 0u, 0u, 0u);
	} else {
		for (int k = 0; k < 4; ++k)
			if (i + k >= first && i + k < last)
				dst[k] = 0u;
	}
}

// ---------------------------------------------------------------- k_refine ---
// encode.c:84-93 second pass: raw magnitude bits of already-significant
// coefficients, in coefficient order.  The k-th refinement coefficient of the
// segment owns bit (segment block offset + k).  One wave per tile, all planes,
// the tile's rows held in registers; the bits of one (tile, plane) are compacted
// through an LDS staging row and merged into the stream with atomicOr.

__global__ __launch_bounds__(256) void k_refine(PackGeom g, const int *__restrict__ lin, Work w, unsigned *out, long out_words)
{
	__shared__ unsigned stage[4][TILE / 32 + 2];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int tile = blockIdx.x * 4 + wv;
	const int plane = blockIdx.y;
	if (tile >= w.NT)
		return;
	const int img = plane / g.C, c = plane - img * g.C;
	const ImgInfo &I = w.info[img];
	int l = 0;
	while (l + 1 < g.levels && tile >= g.tile_first[l + 1])
		++l;
	const int j = tile - g.tile_first[l];
	const long ring1 = g.pixels[l + 1];
	const long base = g.pixels[l] + (long)j * TILE;
	const int *src = lin + (long)plane * g.total;
	unsigned m[ROWS];
	bool in[ROWS];
#pragma unroll
	for (int r = 0; r < ROWS; ++r) {
		const long i = base + r * 64 + lane;
		in[r] = i < ring1;
		const int v = in[r] ? src[i] : 0;
		m[r] = (unsigned)(v < 0 ? -v : v);
	}
	const int *eb = w.seg_ebase + (long)img * (MAX_SEGS + 1);
	const int *sx = w.segidx + ((long)img * 48 + c * 16 + l) * MAX_PLANES;
	const unsigned *refscum = w.ent_refscum + img * (w.ES + 1);
	unsigned *st = stage[wv];
	unsigned *dst = out + img * out_words;
	const unsigned long long below = (1ull << lane) - 1ull;
	for (int p = I.planes[c] - 2; p >= 0; --p) {   // the top plane of a channel has nothing to refine
		const int k1 = sx[p];
		if (!k1)
			continue;
		const int k = k1 - 1;
		const int e = eb[k] + j;
		const unsigned nref = w.ent_refs[img * w.ES + e];
		if (!nref)
			continue;
		const unsigned long long bit0 = w.seg_rawoff[(long)img * MAX_SEGS + k] + (refscum[e] - refscum[eb[k]]);
		if (lane < TILE / 32 + 2)
			st[lane] = 0;
		// each wave only touches its own stage row; wave-level ordering suffices
		__builtin_amdgcn_wave_barrier();
		const int shift = (int)(bit0 & 31);
		unsigned done = 0;
#pragma unroll
		for (int r = 0; r < ROWS; ++r) {
			const bool refine = in[r] && (m[r] >> (p + 1)) != 0;
			const unsigned long long rm = ballot64(refine);
			if (refine && ((m[r] >> p) & 1u)) {
				const unsigned pos = (unsigned)shift + done + (unsigned)__builtin_popcountll(rm & below);
				atomicOr(&st[pos >> 5], 1u << (pos & 31));
			}
			done += (unsigned)__builtin_popcountll(rm);
		}
		__builtin_amdgcn_wave_barrier();
		const long w0 = (long)(bit0 >> 5);
		const int nwords = (int)((shift + nref + 31) >> 5);
		if (lane < nwords) {
			const unsigned val = st[lane];
			if (val && w0 + lane < out_words)
				atomicOr(dst + w0 + lane, val);
		}
		__builtin_amdgcn_wave_barrier();
	}
}

} // namespace

// ------------------------------------------------------------------ driver ---

enum {
	SLOT_PK_CUM = 2, SLOT_PK_SMALL, SLOT_PK_ENT, SLOT_PK_TOKRUN, SLOT_PK_TOKB, SLOT_PK_LUT, SLOT_PK_CHUNK,
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

extern "C" int dwtx_encode_planes(dwtx_ctx *ctx, const int32_t *lin, int W, int H, int C, int n, long capacity,
	uint8_t *out, size_t out_stride, dwtx_stream_info *dev_info)
{
	if (!ctx || !lin || !out || !dev_info || (C != 1 && C != 3) || n < 1 || n > 65535 || (out_stride & 3) || out_stride < 8)
		return DWTX_ERR_ARG;
	DWTX_CHECK_DIMS(W, H);
	PackGeom g;
	{
		int lengths[DWTX_MAX_LEVELS], pixels[DWTX_MAX_LEVELS], widths[DWTX_MAX_LEVELS], heights[DWTX_MAX_LEVELS];
		g.levels = dwtx_compute_lengths(lengths, pixels, widths, heights, W, H, DWTX_MIN_LEN);
		for (int l = 0; l <= g.levels; ++l)
			g.pixels[l] = pixels[l];
	}
	g.C = C;
	g.W = W;
	g.H = H;
	g.total = (long)W * H;
	int NT = 0;
	for (int l = 0; l < g.levels; ++l) {
		g.tile_first[l] = NT;
		NT += (int)(((long)g.pixels[l + 1] - g.pixels[l] + TILE - 1) / TILE);
	}
	g.tile_first[g.levels] = NT;

	Work w;
	memset(&w, 0, sizeof(w));
	w.NT = NT;
	w.ES = (long)NT * C * MAX_PLANES + 16;
	w.TS = ((long)C * (g.total - g.pixels[0]) + MAX_SEGS + 8 + 63) / 64 * 64;   // multiple of 64: every image's token arrays start vector-aligned
	w.NCS = (w.TS / SUB + FSUBS - 1) / FSUBS + 2;   // waves of the fast order pass (>= chunks of the exact one)
	w.NGS = (w.NCS + GROUP - 1) / GROUP;
	w.NCB = (w.ES + CARRY_BLOCK - 1) / CARRY_BLOCK;
	const int nplanes = n * C;

	// carve scratch
	{
		size_t b = sizeof(unsigned short) * (size_t)nplanes * NT * NCUM;
		w.cum = (unsigned short *)dwtx_scratch(ctx, SLOT_PK_CUM, b);
		size_t off = 0;
		auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
		const size_t o_planes = take(sizeof(int) * nplanes);
		const size_t o_info = take(sizeof(ImgInfo) * n);
		const size_t o_slow = take(sizeof(int) * n);
		const size_t o_sbits = take(sizeof(unsigned long long) * n);
		const size_t o_sd = take(sizeof(int) * (size_t)n * MAX_SEGS);
		const size_t o_eb = take(sizeof(int) * (size_t)n * (MAX_SEGS + 1));
		const size_t o_sr = take(sizeof(unsigned) * (size_t)n * MAX_SEGS);
		const size_t o_ro = take(sizeof(unsigned long long) * (size_t)n * MAX_SEGS);
		const size_t o_bt = take(sizeof(unsigned) * (size_t)n * MAX_SEGS);
		const size_t o_sx = take(sizeof(int) * (size_t)n * 48 * MAX_PLANES);
		char *small = (char *)dwtx_scratch(ctx, SLOT_PK_SMALL, off);
		if (!w.cum || !small)
			return DWTX_ERR_NOMEM;
		w.planes_dev = (int *)(small + o_planes);
		w.info = (ImgInfo *)(small + o_info);
		w.slow = (int *)(small + o_slow);
		w.stream_bits = (unsigned long long *)(small + o_sbits);
		w.seg_desc = (int *)(small + o_sd);
		w.seg_ebase = (int *)(small + o_eb);
		w.seg_refs = (unsigned *)(small + o_sr);
		w.seg_rawoff = (unsigned long long *)(small + o_ro);
		w.brk_tok = (unsigned *)(small + o_bt);
		w.segidx = (int *)(small + o_sx);
		DWTX_HIP(hipMemsetAsync(small, 0, o_sd, ctx->stream));
		if (getenv("DWTX_FORCE_EXACT_ORDERS"))   // test hook: take the hierarchical 32-state pass for every image
			DWTX_HIP(hipMemsetAsync(small + o_slow, 1, sizeof(int) * n, ctx->stream));

		off = 0;
		const size_t o_on = take(sizeof(short) * (size_t)n * w.ES);
		const size_t o_ze = take(sizeof(short) * (size_t)n * w.ES);
		const size_t o_re = take(sizeof(short) * (size_t)n * w.ES);
		const size_t o_tz = take(sizeof(short) * (size_t)n * w.ES);
		const size_t o_tb = take(sizeof(unsigned) * (size_t)n * (w.ES + 1));
		const size_t o_rc = take(sizeof(unsigned) * (size_t)n * (w.ES + 1));
		const size_t o_ca = take(sizeof(RunMap) * (size_t)n * w.NCB);
		const size_t o_ci = take(sizeof(unsigned) * (size_t)n * w.NCB);
		const size_t o_ek = take(sizeof(unsigned) * 2 * (size_t)n * w.NCB);
		char *ent = (char *)dwtx_scratch(ctx, SLOT_PK_ENT, off);
		if (!ent)
			return DWTX_ERR_NOMEM;
		w.ent_ones = (unsigned short *)(ent + o_on);
		w.ent_zeros = (unsigned short *)(ent + o_ze);
		w.ent_refs = (unsigned short *)(ent + o_re);
		w.ent_tz = (unsigned short *)(ent + o_tz);
		w.ent_tokbase = (unsigned *)(ent + o_tb);
		w.ent_refscum = (unsigned *)(ent + o_rc);
		w.carry_agg = (RunMap *)(ent + o_ca);
		w.carry_in = (unsigned *)(ent + o_ci);
		w.ent_blk = (unsigned *)(ent + o_ek);

		w.tok_run = (unsigned *)dwtx_scratch(ctx, SLOT_PK_TOKRUN, sizeof(unsigned) * (size_t)n * w.TS);
		char *tb = (char *)dwtx_scratch(ctx, SLOT_PK_TOKB, 4 * (size_t)n * w.TS);
		if (!w.tok_run || !tb)
			return DWTX_ERR_NOMEM;
		w.tok_off = (unsigned short *)tb;
		w.tok_flag = (unsigned char *)tb + 2 * (size_t)n * w.TS;
		w.tok_ord = (unsigned char *)tb + 3 * (size_t)n * w.TS;

		w.sublut = (unsigned char *)dwtx_scratch(ctx, SLOT_PK_LUT, (size_t)n * w.NCS * 64 * 32);
		off = 0;
		const size_t o_lut = take((size_t)n * w.NCS * 32);
		const size_t o_gl = take((size_t)n * w.NGS * 32);
		const size_t o_ce = take((size_t)n * w.NCS);
		const size_t o_ge = take((size_t)n * w.NGS);
		const size_t o_cb = take(sizeof(unsigned long long) * (size_t)n * w.NCS);
		const size_t o_cs = take(sizeof(unsigned long long) * (size_t)n * w.NCS);
		const size_t o_lb = take(sizeof(unsigned long long) * (size_t)n * w.NCS * 64);
		char *ch = (char *)dwtx_scratch(ctx, SLOT_PK_CHUNK, off);
		if (!w.sublut || !ch)
			return DWTX_ERR_NOMEM;
		w.lut = (unsigned char *)(ch + o_lut);
		w.glut = (unsigned char *)(ch + o_gl);
		w.chunk_entry = (unsigned char *)(ch + o_ce);
		w.group_entry = (unsigned char *)(ch + o_ge);
		w.chunk_bits = (unsigned long long *)(ch + o_cb);
		w.chunk_base = (unsigned long long *)(ch + o_cs);
		w.lane_bits = (unsigned long long *)(ch + o_lb);
	}

	hipStream_t s = ctx->stream;
	const long out_words = (long)(out_stride / 4);
	unsigned *outw = (unsigned *)out;
	// k_plan stores the first words (header, root image, plane counts) outright; the rest of the stream is
	// cleared by k_clear_stream once its length is known

	hipLaunchKernelGGL(k_hist, dim3(dwtx_cdiv(NT, 4), nplanes), dim3(256), 0, s, g, lin, w);
	hipLaunchKernelGGL(k_plan, dim3(n), dim3(64), 0, s, g, lin, w, outw, out_words, capacity);
	hipLaunchKernelGGL(k_entries_count, dim3((unsigned)w.NCB, n), dim3(ENT_BLOCK), 0, s, g, w);
	hipLaunchKernelGGL(k_entries_blocks, dim3(n), dim3(ENT_BLOCK), 0, s, w);
	hipLaunchKernelGGL(k_entries_finish, dim3((unsigned)w.NCB, n), dim3(ENT_BLOCK), 0, s, w);
	hipLaunchKernelGGL(k_entries_segs, dim3(n), dim3(ENT_BLOCK), 0, s, w);
	hipLaunchKernelGGL(k_tokens, dim3(dwtx_cdiv(NT, 4), nplanes), dim3(256), 0, s, g, lin, w);
	hipLaunchKernelGGL(k_carry_local, dim3((unsigned)w.NCB, n), dim3(CARRY_BLOCK), 0, s, w);
	hipLaunchKernelGGL(k_carry_blocks, dim3(n), dim3(CARRY_BLOCK), 0, s, w);
	hipLaunchKernelGGL(k_carry_apply, dim3((unsigned)w.NCB, n), dim3(CARRY_BLOCK), 0, s, w);
	hipLaunchKernelGGL(k_orders_fast, dim3((int)((w.NCS + 3) / 4), n), dim3(256), 0, s, w);
	// exact pass: only images the fast pass flagged (their kernels return at once otherwise)
	hipLaunchKernelGGL(k_lut, dim3(512, n), dim3(256), 0, s, w);
	hipLaunchKernelGGL(k_chain_groups, dim3(64, n), dim3(256), 0, s, w);
	hipLaunchKernelGGL(k_chain_image, dim3(n), dim3(1024), 0, s, w);
	hipLaunchKernelGGL(k_orders, dim3(512, n), dim3(64 * ORD_WAVES), 0, s, w);
	hipLaunchKernelGGL(k_bitscan, dim3(n), dim3(1024), 0, s, w, capacity);
	hipLaunchKernelGGL(k_clear_stream, dim3((unsigned)((out_words / 4 + 256) / 256), n), dim3(256), 0, s, w, outw, out_words);
	hipLaunchKernelGGL(k_emit, dim3((unsigned)((w.TS / (64 * EMIT_TOK) + 1 + 3) / 4), n), dim3(256), 0, s, w, outw, out_words);
	hipLaunchKernelGGL(k_refine, dim3(dwtx_cdiv(NT, 4), nplanes), dim3(256), 0, s, g, lin, w, outw, out_words);
	DWTX_LAUNCH_CHECK();
	static_assert(sizeof(dwtx_stream_info) == sizeof(ImgInfo), "ImgInfo is the device image of dwtx_stream_info");
	DWTX_HIP(hipMemcpyAsync(dev_info, w.info, sizeof(ImgInfo) * (size_t)n, hipMemcpyDeviceToDevice, s));
	return DWTX_OK;
}
