// unpack.hip — the decoder's entropy stage on the GPU (decode.c:67-134,174-250
// over rle.h:66-103, vli.h:86-101, bits.h:80-106).
//
// The stream is sequential by construction: where a token starts depends on
// every token before it (adaptive VLI order, zero runs that straddle segments,
// raw bits whose count depends on what earlier planes made significant).  The
// work is split so that only the irreducible part stays serial:
//
//   k_tokenize  one lane per image: walks header, root image, plane counts and
//               the segment schedule, parsing VLI tokens with a register
//               look-ahead FIFO.  It touches no coefficient: per-(channel,
//               level) counters of not-yet-significant coefficients tell it how
//               many symbols each segment holds.  Output: for every pass-1
//               symbol that is a one, a bit in `onebits` (and its sign in
//               `signbits`) at (segment symbol base + symbol index); per
//               segment the stream offset of its refinement block.  Truncated
//               streams simply stop here; what was parsed stays valid
//               (decode.c:204-205).
//   k_rank      per (ring, plane): exclusive scan over 1024-coefficient tiles
//               of the number of not-yet-significant coefficients.
//   k_apply     one wave per tile and plane, planes descending: a coefficient
//               that is not yet significant is pass-1 symbol #rank -> read its
//               bit from `onebits`; a significant one is refinement bit #(index
//               - rank) -> read it straight from the stream.
//   k_finish    sign-magnitude -> two's complement (decode.c:102-117).
#include "dwtx_internal.h"

#include <string.h>

namespace {

constexpr int TILE = 1024;
constexpr int ROWS = TILE / 64;
constexpr int MAX_PLANES = 16;
constexpr int MAX_SEGS = 3 * 16 * MAX_PLANES;

struct UnpackGeom {
	int levels, C, W, H;
	long total;                               // W*H of the full image
	long lin_stride;                          // ints per plane in the lin buffer
	int pixels[DWTX_MAX_LEVELS + 1];
	int tile_first[DWTX_MAX_LEVELS + 1];
	int levels_max;                           // decode.c:163-171 PIXELS cap
};

struct DecInfo {
	int status;            // 0 ok, 1 = header / root image / plane counts unreadable (decode.c exits 1)
	int W, H, C;
	int levels;
	int planes[3];
	int pmax;
	int level;             // finest level any segment touched (decode.c:197,203,219,236); -1 none
	int nsegs;
	int truncated;         // a segment ended early (EOF) or the PIXELS cap stopped the walk
	int missing[48];       // decode.c:193-196: planes not fully decoded, [c*16 + l]
	unsigned long long bits_used;
};

struct DWork {
	DecInfo *info;                  // [n]
	int *seg_desc;                  // [n][MAX_SEGS]
	unsigned long long *seg_symbase; // [n][MAX_SEGS]
	unsigned long long *seg_b2;     // [n][MAX_SEGS]
	unsigned *seg_n2done;           // [n][MAX_SEGS]
	int *segidx;                    // [n][3][16][MAX_PLANES] -> k+1
	int *nonsig;                    // [n][3][16]
	unsigned *onebits, *signbits;   // [n][BW] words
	unsigned short *tile_sig;       // [nplanes][NT]
	unsigned *tile_rank;            // [nplanes][NT]
	long BW;                        // bitmap words per image
	int NT;
};

__device__ __forceinline__ int popc_below(unsigned long long m)
{
	return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
}

// ---- bits.h:80-106 as a look-ahead FIFO of 64-bit words in registers ---------
// f0..f5 cover stream bits [base, base+384).  A shift happens every 64 consumed
// bits and issues the load for the word 320 bits ahead, so by the time a word
// reaches the front its load has long completed.

struct BitReader {
	const unsigned long long *w;
	long n64;                       // readable 64-bit words
	unsigned long long end_bits;    // 8 * stream length
	unsigned long long b;           // read position
	unsigned long long base;
	unsigned long long f0, f1, f2, f3, f4, f5;

	__device__ __forceinline__ unsigned long long ld(unsigned long long bit) const
	{
		const long i = (long)(bit >> 6);
		return i < n64 ? w[i] : 0ull;
	}
	__device__ void seek(unsigned long long nb)
	{
		b = nb;
		base = nb & ~63ull;
		f0 = ld(base);
		f1 = ld(base + 64);
		f2 = ld(base + 128);
		f3 = ld(base + 192);
		f4 = ld(base + 256);
		f5 = ld(base + 320);
	}
	__device__ __forceinline__ unsigned long long peek()   // bits [b, b+64)
	{
		while (b - base >= 64) {
			f0 = f1;
			f1 = f2;
			f2 = f3;
			f3 = f4;
			f4 = f5;
			base += 64;
			f5 = ld(base + 320);
		}
		const int off = (int)(b - base);
		return off ? (f0 >> off) | (f1 << (64 - off)) : f0;
	}
	__device__ __forceinline__ unsigned long long avail() const { return end_bits > b ? end_bits - b : 0ull; }
	// n <= 32 raw bits, LSB first (bits.h:94-106); false at end of data
	__device__ __forceinline__ bool read(int n, unsigned &v)
	{
		if ((unsigned long long)n > avail())
			return false;
		const unsigned long long win = peek();
		v = n >= 32 ? (unsigned)win : (unsigned)win & ((1u << n) - 1u);
		b += n;
		return true;
	}
	// vli.h:86-101: (o*-o) zeros, a one, o* remainder bits; value = rem + 2^o* - 2^o
	__device__ __forceinline__ bool vli(int &order, unsigned &val)
	{
		const unsigned long long win = peek();
		const unsigned long long av = avail();
		const int z = win ? __builtin_ctzll(win) : 64;
		if ((unsigned long long)z >= av)
			return false;                       // ran out of data inside the unary part
		const int top = order + z;
		if (top > 31)
			return false;                       // not a stream this codec can have written
		const int need = z + 1 + top;
		if ((unsigned long long)need > av)
			return false;
		const unsigned rem = top ? (unsigned)(win >> (z + 1)) & (unsigned)((1ull << top) - 1ull) : 0u;
		val = rem + (1u << top) - (1u << order);
		order = top >= 2 ? top - 2 : 0;
		b += need;
		return true;
	}
};

struct BitmapWriter {
	unsigned *one, *sign;
	long cur;
	unsigned aone, asign;
	__device__ __forceinline__ void flush()
	{
		if (cur >= 0 && aone) {
			one[cur] = aone;
			if (asign)
				sign[cur] = asign;
		}
		aone = asign = 0;
	}
	__device__ __forceinline__ void set_one(unsigned long long pos)
	{
		const long wi = (long)(pos >> 5);
		if (wi != cur) {
			flush();
			cur = wi;
		}
		aone |= 1u << (pos & 31);
	}
	__device__ __forceinline__ void set_sign(unsigned long long pos) { asign |= 1u << (pos & 31); }
};

// --------------------------------------------------------------- k_tokenize ---

__global__ __launch_bounds__(64) void k_tokenize(UnpackGeom g, DWork w, const unsigned char *streams, long stream_stride,
	const unsigned long long *lens, int *lin, int n)
{
	const int img = blockIdx.x * blockDim.x + threadIdx.x;
	if (img >= n)
		return;
	DecInfo &I = w.info[img];
	I.status = 1;
	I.W = g.W;
	I.H = g.H;
	I.C = g.C;
	I.levels = g.levels;
	I.level = -1;
	I.nsegs = 0;
	I.truncated = 0;
	I.pmax = 0;
	for (int i = 0; i < 48; ++i)
		I.missing[i] = 0;
	const unsigned char *s8 = streams + img * stream_stride;
	const unsigned long long len = lens[img];
	// decode.c:142-159 header
	if (len < 6 || s8[0] != 'W' || s8[1] != (g.C == 3 ? '6' : '5') ||
		(s8[2] | (s8[3] << 8)) + 1 != g.W || (s8[4] | (s8[5] << 8)) + 1 != g.H)
		return;
	BitReader br;
	br.w = (const unsigned long long *)s8;
	br.n64 = stream_stride >> 3;
	br.end_bits = len * 8;
	br.seek(48);
	int order = 0;   // vli.h:24
	// decode.c:119-134 root image
	for (int c = 0; c < g.C; ++c) {
		unsigned cnt;
		if (!br.vli(order, cnt))
			return;
		int *dst = lin + (long)(img * g.C + c) * g.lin_stride;
		if (cnt)
			for (int i = 0; i < g.pixels[0]; ++i) {
				unsigned v, neg = 0;
				if (cnt > 32 || !br.read((int)cnt, v))
					return;
				if (v && !br.read(1, neg))
					return;
				dst[i] = neg ? -(int)v : (int)v;
			}
	}
	int planes[3] = { 0, 0, 0 };
	int pmax = 0;
	for (int c = 0; c < g.C; ++c) {   // decode.c:183-186
		unsigned p;
		if (!br.vli(order, p) || p > MAX_PLANES)
			return;
		planes[c] = (int)p;
		I.planes[c] = (int)p;
		pmax = (int)p > pmax ? (int)p : pmax;
	}
	I.pmax = pmax;
	I.status = 0;
	const int levels = g.levels;
	for (int c = 0; c < g.C; ++c)
		for (int l = 0; l < levels; ++l)
			I.missing[c * 16 + l] = planes[c];

	int *nonsig = w.nonsig + (long)img * 48;
	for (int c = 0; c < g.C; ++c)
		for (int l = 0; l < levels; ++l)
			nonsig[c * 16 + l] = g.pixels[l + 1] - g.pixels[l];
	int *sd = w.seg_desc + (long)img * MAX_SEGS;
	unsigned long long *ssym = w.seg_symbase + (long)img * MAX_SEGS;
	unsigned long long *sb2 = w.seg_b2 + (long)img * MAX_SEGS;
	unsigned *sn2 = w.seg_n2done + (long)img * MAX_SEGS;
	int *sidx = w.segidx + (long)img * 3 * 16 * MAX_PLANES;
	BitmapWriter bm;
	bm.one = w.onebits + img * w.BW;
	bm.sign = w.signbits + img * w.BW;
	bm.cur = -1;
	bm.aone = bm.asign = 0;

	unsigned cnt = 0;              // rle.h:25
	unsigned long long symtotal = 0;
	int nsegs = 0, level = -1;

	// decode.c:67-100 without touching coefficients; false = stop decoding (decode.c:204,221,238)
	auto segment = [&](int c, int l, int p) -> bool {
		const int num = g.pixels[l + 1] - g.pixels[l];
		const int n1 = p < 0 ? num : nonsig[c * 16 + l];
		const int n2 = num - n1;
		const int k = nsegs++;
		const unsigned long long sym0 = symtotal;
		symtotal += ((unsigned long long)num + 31) & ~31ull;
		sd[k] = c | (l << 4) | ((p + 1) << 8);
		ssym[k] = sym0;
		sb2[k] = 0;
		sn2[k] = 0;
		if (p >= 0)
			sidx[(c * 16 + l) * MAX_PLANES + p] = k + 1;
		int q = 0, ones = 0;
		bool ok = true;
		while (q < n1) {
			unsigned zr;
			if (cnt == 0) {   // rle.h:70-75
				unsigned v;
				if (!br.vli(order, v)) {
					ok = false;
					break;
				}
				zr = v;
			} else {
				zr = cnt - 1;
			}
			const unsigned rem = (unsigned)(n1 - q);
			if (zr >= rem) {            // the run outlives this segment's first pass
				q = n1;
				cnt = zr - rem + 1;
				break;
			}
			q += (int)zr;
			cnt = 0;
			if (p >= 0)
				bm.set_one(sym0 + (unsigned)q);
			++ones;
			unsigned neg;
			if (!br.read(1, neg)) {     // magnitude bit stays, sign unknown (decode.c:80-85)
				++q;
				ok = false;
				break;
			}
			if (neg && p >= 0)
				bm.set_sign(sym0 + (unsigned)q);
			++q;
		}
		if (p >= 0)
			nonsig[c * 16 + l] = n1 - ones;
		if (!ok)
			return false;
		if (p >= 0 && n2 > 0) {
			if (cnt > 0) {              // rle.h:95-101: a pending run must end exactly here
				if (cnt != 1)
					return false;
				cnt = 0;
			}
			sb2[k] = br.b;
			const unsigned long long av = br.avail();
			if ((unsigned long long)n2 > av) {
				sn2[k] = (unsigned)av;
				br.b = br.end_bits;
				return false;
			}
			sn2[k] = (unsigned)n2;
			br.seek(br.b + (unsigned)n2);
		}
		return true;
	};

	const int layers_max = 2 * (levels > pmax ? levels : pmax) - 1;
	bool stop = g.levels_max == 0;        // decode.c:199-200
	if (!stop && pmax == planes[0]) {     // decode.c:201-207
		level = 0;
		if (segment(0, 0, planes[0] - 1))
			--I.missing[0];
		else
			stop = true;
	}
	for (int layer = 0; !stop && layer < layers_max; ++layer) {   // decode.c:208-243
		for (int l = 0; !stop && l < levels && l <= layer + 1; ++l) {
			if (l >= g.levels_max) {
				stop = true;
				break;
			}
			const int p = pmax - 1 - (layer + 1 - l);
			if (p < 0 || p >= planes[0])
				continue;
			level = level < l ? l : level;
			if (segment(0, l, p))
				--I.missing[l];
			else
				stop = true;
		}
		for (int l = 0; !stop && l < levels && l <= layer; ++l) {
			if (l >= g.levels_max) {
				stop = true;
				break;
			}
			const int p = pmax - 1 - (layer - l);
			for (int c = 1; !stop && c < g.C; ++c) {
				if (p < 0 || p >= planes[c])
					continue;
				level = level < l ? l : level;
				if (segment(c, l, p))
					--I.missing[c * 16 + l];
				else
					stop = true;
			}
		}
	}
	bm.flush();
	I.level = level;
	I.nsegs = nsegs;
	I.truncated = stop ? 1 : 0;
	I.bits_used = br.b;
}

// ------------------------------------------------------------------ k_rank ---
// per (plane-of-image, ring): exclusive scan over tiles of the not-yet-significant count

__global__ __launch_bounds__(1024) void k_rank(UnpackGeom g, DWork w, int p)
{
	__shared__ unsigned wsum[16];
	__shared__ unsigned carry;
	const int l = blockIdx.x;
	const int plane = blockIdx.y;
	const int img = plane / g.C, c = plane - img * g.C;
	if (w.info[img].status || !w.segidx[((long)img * 48 + c * 16 + l) * MAX_PLANES + p])
		return;
	const int t0 = g.tile_first[l], nt = g.tile_first[l + 1] - t0;
	const long ring = (long)g.pixels[l + 1] - g.pixels[l];
	const unsigned short *sig = w.tile_sig + (long)plane * w.NT + t0;
	unsigned *rank = w.tile_rank + (long)plane * w.NT + t0;
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	if (threadIdx.x == 0)
		carry = 0;
	__syncthreads();
	for (int b0 = 0; b0 < nt; b0 += 1024) {
		const int i = b0 + threadIdx.x;
		unsigned v = 0;
		if (i < nt) {
			const long left = ring - (long)i * TILE;
			v = (unsigned)(left < TILE ? left : TILE) - sig[i];
		}
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
		const unsigned cy = carry;
		if (i < nt)
			rank[i] = cy + woff + inc - v;
		__syncthreads();
		if (threadIdx.x == 0)
			carry = cy + all;
		__syncthreads();
	}
}

// ----------------------------------------------------------------- k_apply ---

__global__ __launch_bounds__(256) void k_apply(UnpackGeom g, DWork w, const unsigned char *streams, long stream_stride,
	int *lin, int p)
{
	const int lane = threadIdx.x & 63;
	const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
	const int plane = blockIdx.y;
	if (tile >= w.NT)
		return;
	const int img = plane / g.C, c = plane - img * g.C;
	if (w.info[img].status)
		return;
	int l = 0;
	while (l + 1 < g.levels && tile >= g.tile_first[l + 1])
		++l;
	const int k1 = w.segidx[((long)img * 48 + c * 16 + l) * MAX_PLANES + p];
	if (!k1)
		return;
	const int k = k1 - 1;
	const int j = tile - g.tile_first[l];
	const long ring1 = g.pixels[l + 1];
	const long base = g.pixels[l] + (long)j * TILE;
	unsigned *v32 = (unsigned *)lin + (long)plane * g.lin_stride;
	const unsigned *one = w.onebits + img * w.BW;
	const unsigned *sgn = w.signbits + img * w.BW;
	const unsigned *stream = (const unsigned *)(streams + img * stream_stride);
	const unsigned long long sym0 = w.seg_symbase[(long)img * MAX_SEGS + k];
	const unsigned long long b2 = w.seg_b2[(long)img * MAX_SEGS + k];
	const unsigned n2done = w.seg_n2done[(long)img * MAX_SEGS + k];
	unsigned rank = w.tile_rank[(long)plane * w.NT + tile];   // not-yet-significant coefficients before this tile
	const unsigned long long below = (1ull << lane) - 1ull;
	unsigned newsig = 0;
	for (int r = 0; r < ROWS; ++r) {
		const long i = base + r * 64 + lane;
		const bool in = i < ring1;
		unsigned v = in ? v32[i] : 0u;
		const bool was_sig = in && (v & 0x7fffffffu) != 0;
		const unsigned long long nm = __ballot(in && !was_sig);
		const unsigned r1 = rank + (unsigned)__builtin_popcountll(nm & below);
		if (in && !was_sig) {
			const unsigned long long pos = sym0 + r1;
			if ((one[pos >> 5] >> (pos & 31)) & 1u) {
				v |= 1u << p;
				v |= ((sgn[pos >> 5] >> (pos & 31)) & 1u) << 31;
				v32[i] = v;
			}
		} else if (in) {
			const unsigned r2 = (unsigned)(i - g.pixels[l]) - r1;   // significant coefficients before this one
			if (r2 < n2done) {
				const unsigned long long pos = b2 + r2;
				const unsigned bit = (stream[pos >> 5] >> (pos & 31)) & 1u;
				if (bit)
					v32[i] = v | (bit << p);
			}
		}
		newsig += (unsigned)__builtin_popcountll(__ballot(in && !was_sig && (v & 0x7fffffffu) != 0));
		rank += (unsigned)__builtin_popcountll(nm);
	}
	if (lane == 0 && newsig)
		w.tile_sig[(long)plane * w.NT + tile] += (unsigned short)newsig;
}

// decode.c:102-117 process(): sign<<31 | magnitude -> two's complement, detail rings only
__global__ __launch_bounds__(256) void k_finish(UnpackGeom g, int *lin, int nplanes)
{
	const long per = g.lin_stride - g.pixels[0];
	const long totalw = per * nplanes;
	for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < totalw; i += (long)gridDim.x * blockDim.x) {
		const long plane = i / per, off = i - plane * per;
		unsigned *p = (unsigned *)lin + plane * g.lin_stride + g.pixels[0] + off;
		const unsigned v = *p;
		if (v) {
			const int mag = (int)(v & 0x1fffffffu);
			*p = (unsigned)((v >> 31) ? -mag : mag);
		}
	}
}

} // namespace

enum { SLOT_UP_SMALL = 12, SLOT_UP_BITS, SLOT_UP_TILES };

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

extern "C" int dwtx_decode_planes(dwtx_ctx *ctx, int32_t *lin, const uint8_t *streams, size_t stream_stride,
	const unsigned long long *dev_lens, int W, int H, int C, int n, int levels_max, dwtx_decode_info *host_info)
{
	if (!ctx || !lin || !streams || !dev_lens || !host_info || W < DWTX_MIN_LEN || H < DWTX_MIN_LEN || W > 65536 ||
		H > 65536 || (C != 1 && C != 3) || n < 1 || n > 65535 / 3 || (stream_stride & 7) || stream_stride < 64)
		return DWTX_ERR_ARG;
	UnpackGeom g;
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
	g.lin_stride = g.total;
	g.levels_max = levels_max < 0 || levels_max > g.levels ? g.levels : levels_max;
	int NT = 0;
	for (int l = 0; l < g.levels; ++l) {
		g.tile_first[l] = NT;
		NT += (int)(((long)g.pixels[l + 1] - g.pixels[l] + TILE - 1) / TILE);
	}
	g.tile_first[g.levels] = NT;
	const int nplanes = n * C;

	DWork w;
	memset(&w, 0, sizeof(w));
	w.NT = NT;
	// every segment owns ceil32(ring size) symbol slots; at most MAX_PLANES segments per (channel, level)
	w.BW = (long)((((unsigned long long)g.total + 32ull * g.levels) * C * MAX_PLANES) >> 5) + 64;
	{
		size_t off = 0;
		auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
		const size_t o_info = take(sizeof(DecInfo) * n);
		const size_t o_idx = take(sizeof(int) * (size_t)n * 48 * MAX_PLANES);
		const size_t o_zero_end = off;
		const size_t o_sd = take(sizeof(int) * (size_t)n * MAX_SEGS);
		const size_t o_sym = take(sizeof(unsigned long long) * (size_t)n * MAX_SEGS);
		const size_t o_b2 = take(sizeof(unsigned long long) * (size_t)n * MAX_SEGS);
		const size_t o_n2 = take(sizeof(unsigned) * (size_t)n * MAX_SEGS);
		const size_t o_ns = take(sizeof(int) * (size_t)n * 48);
		char *small = (char *)dwtx_scratch(ctx, SLOT_UP_SMALL, off);
		unsigned *bits = (unsigned *)dwtx_scratch(ctx, SLOT_UP_BITS, sizeof(unsigned) * 2 * (size_t)n * w.BW);
		off = 0;
		const size_t o_ts = take(sizeof(short) * (size_t)nplanes * NT);
		const size_t o_tr = take(sizeof(unsigned) * (size_t)nplanes * NT);
		char *tiles = (char *)dwtx_scratch(ctx, SLOT_UP_TILES, off);
		if (!small || !bits || !tiles)
			return DWTX_ERR_NOMEM;
		w.info = (DecInfo *)(small + o_info);
		w.segidx = (int *)(small + o_idx);
		w.seg_desc = (int *)(small + o_sd);
		w.seg_symbase = (unsigned long long *)(small + o_sym);
		w.seg_b2 = (unsigned long long *)(small + o_b2);
		w.seg_n2done = (unsigned *)(small + o_n2);
		w.nonsig = (int *)(small + o_ns);
		w.onebits = bits;
		w.signbits = bits + (size_t)n * w.BW;
		w.tile_sig = (unsigned short *)(tiles + o_ts);
		w.tile_rank = (unsigned *)(tiles + o_tr);
		DWTX_HIP(hipMemsetAsync(small, 0, o_zero_end, ctx->stream));
		DWTX_HIP(hipMemsetAsync(bits, 0, sizeof(unsigned) * 2 * (size_t)n * w.BW, ctx->stream));
		DWTX_HIP(hipMemsetAsync(tiles + o_ts, 0, sizeof(short) * (size_t)nplanes * NT, ctx->stream));
	}
	hipStream_t s = ctx->stream;
	DWTX_HIP(hipMemsetAsync(lin, 0, sizeof(int) * (size_t)nplanes * g.lin_stride, s));   // decode.c:177-179
	hipLaunchKernelGGL(k_tokenize, dim3(dwtx_cdiv(n, 64)), dim3(64), 0, s, g, w, streams, (long)stream_stride,
		dev_lens, lin, n);
	DWTX_LAUNCH_CHECK();
	static_assert(sizeof(dwtx_decode_info) == sizeof(DecInfo), "DecInfo is the device image of dwtx_decode_info");
	DWTX_HIP(hipMemcpyAsync(host_info, w.info, sizeof(DecInfo) * (size_t)n, hipMemcpyDeviceToHost, s));
	DWTX_HIP(hipStreamSynchronize(s));
	int pmax = 0;
	for (int i = 0; i < n; ++i)
		if (!host_info[i].status && host_info[i].pmax > pmax)
			pmax = host_info[i].pmax;
	for (int p = pmax - 1; p >= 0; --p) {
		hipLaunchKernelGGL(k_rank, dim3(g.levels, nplanes), dim3(1024), 0, s, g, w, p);
		hipLaunchKernelGGL(k_apply, dim3(dwtx_cdiv(NT, 4), nplanes), dim3(256), 0, s, g, w, streams, (long)stream_stride,
			lin, p);
	}
	hipLaunchKernelGGL(k_finish, dim3(2048), dim3(256), 0, s, g, lin, nplanes);
	DWTX_LAUNCH_CHECK();
	return DWTX_OK;
}
