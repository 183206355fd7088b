// linearize.hip — Hilbert-order linearisation of the wavelet pyramid and its
// inverse (encode.c:32-58 linearization, decode.c:32-65 reconstruction,
// hilbert.h:15-34).
//
// The reference walks d = 0..n*n-1 serially, calls hilbert(n, d) and keeps a
// running output index for the points that fall inside the level's L-shaped
// detail ring.  Here every point is independent: the Hilbert curve visits every
// aligned 2^k x 2^k square contiguously, so the curve is cut into blocks of
// 1024 consecutive d (one 32x32 square each); the number of ring points in a
// square is a closed-form rectangle intersection, an exclusive scan over the
// blocks (done once per image geometry, cached in the context) gives each
// block's first output slot, and inside a block the slot is a ballot/popcount
// rank.  One workgroup handles one block for one plane; lanes take consecutive
// d so the linear side of the copy is a coalesced stream and the pyramid side
// touches a compact 32x32 tile.
#include "hilbert_dev.h"

#include <stdlib.h>
#include <string.h>
#include <new>
#include <vector>

namespace {

constexpr int THREADS = 256;

struct LinGeom {
	int levels;             // levels of the full image
	int W, H;
	int widths[DWTX_MAX_LEVELS], heights[DWTX_MAX_LEVELS], pixels[DWTX_MAX_LEVELS], lengths[DWTX_MAX_LEVELS];
	int blk_first[DWTX_MAX_LEVELS + 1];  // first global block id of ring level l
	int blk_pts_log2[DWTX_MAX_LEVELS];   // log2(points per block) of ring level l (10, or less for tiny levels)
};

// hilbert.h:15-34 — curve index -> (x, y) on an n*n grid, branch-free.
// For x < s: s-1-x == (s-1) ^ x, and adding s is an OR.
__device__ __forceinline__ void hilbert_d2xy(int n, unsigned d, int &xo, int &yo)
{
	unsigned x = 0, y = 0;
	for (unsigned s = 1; s < (unsigned)n; s <<= 1) {
		const unsigned rx = (d >> 1) & 1u;
		const unsigned ry = (d ^ rx) & 1u;
		const unsigned flip = (rx & ~ry & 1u) ? (s - 1) : 0u;
		x ^= flip;
		y ^= flip;
		const unsigned sw = ry ? 0u : (x ^ y);
		x ^= sw;
		y ^= sw;
		x |= rx ? s : 0u;
		y |= ry ? s : 0u;
		d >>= 2;
	}
	xo = (int)x;
	yo = (int)y;
}

__device__ __forceinline__ int overlap(int lo, int len, int bound)
{
	// |[lo, lo+len) ∩ [0, bound)|
	int hi = lo + len;
	hi = hi < bound ? hi : bound;
	return hi > lo ? hi - lo : 0;
}

__device__ __forceinline__ int level_of_block(const LinGeom &g, int b)
{
	int l = 0;
	while (l + 1 < g.levels && b >= g.blk_first[l + 1])
		++l;
	return l;
}

// ring points inside curve block `lb` of ring level l
__device__ __forceinline__ int ring_points_in_block(const LinGeom &g, int l, int lb)
{
	const int n = g.lengths[l + 1];
	const int pl2 = g.blk_pts_log2[l];
	const int side = 1 << (pl2 >> 1);
	int x, y;
	hilbert_d2xy(n, (unsigned)lb << pl2, x, y);
	x &= ~(side - 1);
	y &= ~(side - 1);
	const int full = overlap(x, side, g.widths[l + 1]) * overlap(y, side, g.heights[l + 1]);
	const int ll = overlap(x, side, g.widths[l]) * overlap(y, side, g.heights[l]);
	return full - ll;
}

// one thread per block: counts[b] (scanned afterwards on one workgroup)
__global__ void k_block_counts(LinGeom g, int *__restrict__ counts, int nblocks)
{
	const int b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= nblocks)
		return;
	const int l = level_of_block(g, b);
	counts[b] = ring_points_in_block(g, l, b - g.blk_first[l]);
}

// exclusive scan of counts per level, single workgroup (geometry set-up only)
__global__ __launch_bounds__(1024) void k_block_scan(LinGeom g, int *__restrict__ counts)
{
	__shared__ int wsum[16];
	__shared__ int carry;
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	for (int l = 0; l < g.levels; ++l) {
		const int first = g.blk_first[l], nb = g.blk_first[l + 1] - first;
		if (threadIdx.x == 0)
			carry = 0;
		__syncthreads();
		for (int base = 0; base < nb; base += 1024) {
			const int i = base + threadIdx.x;
			const int v = i < nb ? counts[first + i] : 0;
			int inc = v;
			for (int o = 1; o < 64; o <<= 1) {
				const int t = __shfl_up(inc, o);
				if (lane >= o)
					inc += t;
			}
			if (lane == 63)
				wsum[wv] = inc;
			__syncthreads();
			int woff = 0;
			for (int k = 0; k < wv; ++k)
				woff += wsum[k];
			const int c = carry;
			if (i < nb)
				counts[first + i] = c + woff + inc - v;
			__syncthreads();
			if (threadIdx.x == 1023)
				carry = c + woff + inc;
			__syncthreads();
		}
	}
}

// Curve positions and output slots are the same for every plane: each workgroup works them out
// once for its 1024 curve points and then copies PLANES_PER_GROUP planes.
constexpr int PLANES_PER_GROUP = 16;
constexpr int PTS = 4;   // curve points per thread (1024 / THREADS)

template <bool INVERSE>
__global__ __launch_bounds__(THREADS) void k_ring_copy(LinGeom g, const int *__restrict__ blockbase,
	int *__restrict__ lin, long lin_ps, int *__restrict__ pyr, long pyr_ps, int ppitch,
	const int *__restrict__ missing, int C, int nplanes, unsigned skip_levels, const int *__restrict__ block_list,
	short *__restrict__ fine16, unsigned lv16)
{
	__shared__ int wcount[THREADS / 64];
	const int b = block_list ? block_list[blockIdx.x] : (int)blockIdx.x;
	const int l = level_of_block(g, b);
	const int lb = b - g.blk_first[l];
	const int n = g.lengths[l + 1];
	const int pl2 = g.blk_pts_log2[l];
	const int npts = 1 << pl2;
	const int total = ring_points_in_block(g, l, lb);   // uniform
	if (total == 0)
		return;
	const bool full = total == npts;
	if (((skip_levels >> l) & 1u) && full)
		return;   // on these levels the entropy stage reads / writes whole squares in the pyramid itself
	const int w0 = g.widths[l], h0 = g.heights[l], w1 = g.widths[l + 1], h1 = g.heights[l + 1];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const SquareMap smap = square_map(n, (unsigned)lb);   // uniform
	int slot[PTS], lidx[PTS];
	long offs[PTS], sq = 0;
	bool ok[PTS];
	int running = 0;
#pragma unroll
	for (int q = 0; q < PTS; ++q) {
		const int i = q * THREADS + threadIdx.x;
		int x = 0, y = 0;
		ok[q] = false;
		if (i < npts) {
			if (pl2 == 2 * BLK_LOG2)
				hilbert_in_square(smap, i, x, y);
			else
				hilbert_d2xy(n, ((unsigned)lb << pl2) + (unsigned)i, x, y);
			ok[q] = x < w1 && y < h1 && (x >= w0 || y >= h0);
		}
		offs[q] = (long)y * ppitch + x;
		lidx[q] = (y & 31) * 33 + (x & 31);
		sq = (long)(y & ~31) * ppitch + (x & ~31);
		if (full) {
			slot[q] = i;
		} else {
			const unsigned long long mask = ballot64(ok[q]);
			const int before = __builtin_popcountll(mask & ((1ull << lane) - 1ull));
			if (lane == 0)
				wcount[wv] = __builtin_popcountll(mask);
			__syncthreads();
			int woff = 0, all = 0;
			for (int k = 0; k < THREADS / 64; ++k) {
				const int c = wcount[k];
				woff += k < wv ? c : 0;
				all += c;
			}
			slot[q] = running + woff + before;
			running += all;
			__syncthreads();
		}
	}
	const int first = blockIdx.y * PLANES_PER_GROUP;
	const int last = min(first + PLANES_PER_GROUP, nplanes);
	if (full && pl2 == 2 * BLK_LOG2) {
		// A whole 32x32 square: stage it in LDS so that the pyramid side moves in full 128-byte rows
		// (two per wave instruction) instead of 8x8 patches of 32-byte pieces; two tiles alternate so
		// that one barrier per plane suffices.
		__shared__ int tile[2][32 * 33];
		for (int plane = first; plane < last; ++plane) {
			int *lp = lin + plane * lin_ps + g.pixels[l] + blockbase[b];
			int *pp = pyr + plane * pyr_ps + sq;
			int *t = tile[(plane - first) & 1];
			if (INVERSE) {
				int bias = 0;
				if (missing) {
					const int m = missing[(plane / C) * 48 + (plane % C) * 16 + l] - 2;
					bias = m >= 0 ? 1 << m : 0;
				}
#pragma unroll
				for (int q = 0; q < PTS; ++q) {
					int v = lp[q * THREADS + threadIdx.x];
					if (bias && v)
						v += v < 0 ? -bias : bias;
					t[lidx[q]] = v;
				}
				__syncthreads();
#pragma unroll
				for (int q = 0; q < PTS; ++q) {
					const int i = q * THREADS + threadIdx.x, r = i >> 5, cx = i & 31;
					pp[(long)r * ppitch + cx] = t[r * 33 + cx];
				}
			} else {
#pragma unroll
				for (int q = 0; q < PTS; ++q) {
					const int i = q * THREADS + threadIdx.x, r = i >> 5, cx = i & 31;
					t[r * 33 + cx] = pp[(long)r * ppitch + cx];
				}
				__syncthreads();
#pragma unroll
				for (int q = 0; q < PTS; ++q)
					lp[q * THREADS + threadIdx.x] = t[lidx[q]];
			}
		}
		return;
	}
	// (the finest ring of an 8-bit source may live in 16-bit planes of its own — dwtx_internal.h; its whole squares never come
	// this way, the blocks the ring's edges cut do)
	const bool f16 = fine16 != nullptr && ((lv16 >> l) & 1u);   // uniform
	for (int plane = first; plane < last; ++plane) {
		int *lp = lin + plane * lin_ps + g.pixels[l] + blockbase[b];
		int *pp = pyr + plane * pyr_ps;
		short *fp = f16 ? fine16 + plane * pyr_ps : nullptr;
		int bias = 0;
		if (INVERSE && missing) {
			// decode.c:51-58: planes never decoded leave a dead zone; recentre non-zero values
			const int m = missing[(plane / C) * 48 + (plane % C) * 16 + l] - 2;
			bias = m >= 0 ? 1 << m : 0;
		}
#pragma unroll
		for (int q = 0; q < PTS; ++q) {
			if (!ok[q])
				continue;
			if (INVERSE) {
				int v = lp[slot[q]];
				if (bias && v)
					v += v < 0 ? -bias : bias;
				if (f16)
					fp[offs[q]] = (short)v;
				else
					pp[offs[q]] = v;
			} else {
				lp[slot[q]] = f16 ? (int)fp[offs[q]] : pp[offs[q]];
			}
		}
	}
}

// root LL in raster order (encode.c:37-45 / decode.c:36-44)
template <bool INVERSE>
__global__ void k_root_copy(int w0, int h0, int *__restrict__ lin, long lin_ps, int *__restrict__ pyr, long pyr_ps, int ppitch)
{
	const int plane = blockIdx.y;
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= w0 * h0)
		return;
	const int y = i / w0, x = i - y * w0;
	int *lp = lin + plane * lin_ps + i;
	int *pp = pyr + plane * pyr_ps + (long)y * ppitch + x;
	if (INVERSE)
		*pp = *lp;
	else
		*lp = *pp;
}

} // namespace

// ---- geometry plans, cached per context --------------------------------------

struct dwtx_linplan {
	int W, H;
	LinGeom g;
	int nblocks;
	int *d_blockbase;
	dwtx_tiles tiles;     // the non-empty blocks as the entropy stage's tiles (device arrays in one allocation at tiles.base)
	// the blocks that are copied when the entropy stage takes the whole squares of every level it can (dwtx_square_levels):
	// blocks the ring's edges cut, and the small levels — global block ids, ascending; copy_upto[l] = how many lie below level l
	int *d_copy_list;
	int copy_upto[DWTX_MAX_LEVELS + 1];
	dwtx_linplan *next;
};

static void plan_destroy(dwtx_linplan *p)
{
	(void)hipFree(p->d_blockbase);
	(void)hipFree(const_cast<int *>(p->tiles.base));
	(void)hipFree(p->d_copy_list);
	(void)hipFree(const_cast<int *>(p->tiles.xy2tile));
	free(p);
}

// Fills a zero-initialised plan; on failure the caller destroys it (plan_destroy frees whatever was allocated so far).
// Host temporaries are std::vectors: every return path gives them back.
static int build_plan(dwtx_ctx *ctx, dwtx_linplan *p, int W, int H)
{
	p->W = W;
	p->H = H;
	LinGeom &g = p->g;
	g.W = W;
	g.H = H;
	g.levels = dwtx_compute_lengths(g.lengths, g.pixels, g.widths, g.heights, W, H, DWTX_MIN_LEN);
	int nb = 0;
	for (int l = 0; l < g.levels; ++l) {
		const int n = g.lengths[l + 1];
		int nl2 = 0;
		while ((1 << nl2) < n)
			++nl2;
		const int pl2 = 2 * nl2 < 2 * BLK_LOG2 ? 2 * nl2 : 2 * BLK_LOG2;
		g.blk_pts_log2[l] = pl2;
		g.blk_first[l] = nb;
		nb += (int)(((long)n * n) >> pl2);
	}
	g.blk_first[g.levels] = nb;
	p->nblocks = nb;
	if (hipMalloc((void **)&p->d_blockbase, sizeof(int) * (size_t)(nb > 0 ? nb : 1)) != hipSuccess) {
		dwtx_set_error("plan hipMalloc failed");
		return DWTX_ERR_NOMEM;
	}
	hipLaunchKernelGGL(k_block_counts, dim3(dwtx_cdiv(nb, 256)), dim3(256), 0, ctx->stream, g, p->d_blockbase, nb);
	hipLaunchKernelGGL(k_block_scan, dim3(1), dim3(1024), 0, ctx->stream, g, p->d_blockbase);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) {
		dwtx_set_error("plan kernels -> %s", hipGetErrorString(e));
		return DWTX_ERR_DEVICE;
	}
	// the tile table: every non-empty block, level by level (built on the host once per geometry)
	std::vector<int> hb((size_t)(nb > 0 ? nb : 1));
	if (hipMemcpyAsync(hb.data(), p->d_blockbase, sizeof(int) * (size_t)nb, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
		hipStreamSynchronize(ctx->stream) != hipSuccess) {
		dwtx_set_error("plan download failed");
		return DWTX_ERR_DEVICE;
	}
	// points of block bb that lie inside its level's ring
	auto in_ring = [&](int l, int bb) {
		const int last = g.blk_first[l + 1], ring = g.pixels[l + 1] - g.pixels[l];
		return (bb + 1 < last ? hb[(size_t)bb + 1] : ring) - hb[(size_t)bb];
	};
	int nt = 0;
	for (int l = 0; l < g.levels; ++l)
		for (int bb = g.blk_first[l]; bb < g.blk_first[l + 1]; ++bb)
			nt += in_ring(l, bb) > 0;
	{
		const size_t o_blk = sizeof(int) * (size_t)nt, o_cnt = 2 * sizeof(int) * (size_t)nt;
		const size_t bytes = o_cnt + sizeof(unsigned short) * (size_t)nt + 64;
		std::vector<char> host(bytes);
		int *tb = (int *)host.data(), *tk = (int *)(host.data() + o_blk);
		unsigned short *tc = (unsigned short *)(host.data() + o_cnt);
		int t = 0;
		for (int l = 0; l < g.levels; ++l) {
			p->tiles.tile_first[l] = t;
			for (int bb = g.blk_first[l]; bb < g.blk_first[l + 1]; ++bb) {
				const int c = in_ring(l, bb);
				if (c <= 0)
					continue;
				tb[t] = hb[(size_t)bb];
				tk[t] = bb - g.blk_first[l];
				tc[t] = (unsigned short)c;
				++t;
			}
		}
		p->tiles.tile_first[g.levels] = t;
		p->tiles.NT = t;
		char *dev = nullptr;
		if (hipMalloc((void **)&dev, bytes) != hipSuccess) {
			dwtx_set_error("tile table hipMalloc failed");
			return DWTX_ERR_NOMEM;
		}
		p->tiles.base = (const int *)dev;   // (plan_destroy frees the one allocation through this pointer)
		p->tiles.blk = (const int *)(dev + o_blk);
		p->tiles.cnt = (const unsigned short *)(dev + o_cnt);
		if (hipMemcpy(dev, host.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) {
			dwtx_set_error("tile table upload failed");
			return DWTX_ERR_DEVICE;
		}
	}
	{
		// block coordinates -> tile (for the forward transform's histograms): hilbert.h:15-34 at block granularity
		int total = 0;
		for (int l = 0; l < g.levels; ++l) {
			p->tiles.xy_first[l] = total;
			p->tiles.nbs[l] = g.lengths[l + 1] >= 64 ? g.lengths[l + 1] >> BLK_LOG2 : 0;
			total += p->tiles.nbs[l] * p->tiles.nbs[l];
		}
		p->tiles.xy_first[g.levels] = total;
		std::vector<int> tab((size_t)(total > 0 ? total : 1));
		for (int l = 0; l < g.levels; ++l) {
			const int nbs = p->tiles.nbs[l];
			if (!nbs)
				continue;
			const int first = g.blk_first[l], last = g.blk_first[l + 1];
			int t = p->tiles.tile_first[l];
			for (int bb = first; bb < last; ++bb) {
				const int c = in_ring(l, bb);
				unsigned x = 0, y = 0, d = (unsigned)(bb - first);   // the block's place on the nbs x nbs grid
				for (unsigned sd = 1; sd < (unsigned)nbs; sd <<= 1) {
					const unsigned rx = (d >> 1) & 1u, ry = (d ^ rx) & 1u;
					if (rx && !ry) {
						x ^= sd - 1;
						y ^= sd - 1;
					}
					if (!ry) {
						const unsigned tmp = x;
						x = y;
						y = tmp;
					}
					x |= rx ? sd : 0u;
					y |= ry ? sd : 0u;
					d >>= 2;
				}
				tab[(size_t)p->tiles.xy_first[l] + (size_t)y * nbs + x] = c > 0 ? t++ : -1;
			}
		}
		int *dtab = nullptr;
		if (hipMalloc((void **)&dtab, sizeof(int) * tab.size()) != hipSuccess) {
			dwtx_set_error("block table hipMalloc failed");
			return DWTX_ERR_NOMEM;
		}
		p->tiles.xy2tile = dtab;
		if (hipMemcpy(dtab, tab.data(), sizeof(int) * (size_t)total, hipMemcpyHostToDevice) != hipSuccess) {
			dwtx_set_error("block table upload failed");
			return DWTX_ERR_DEVICE;
		}
	}
	{
		const unsigned sq = dwtx_square_levels(W, H);
		std::vector<int> list((size_t)(nb > 0 ? nb : 1));
		int m = 0;
		for (int l = 0; l < g.levels; ++l) {
			p->copy_upto[l] = m;
			for (int bb = g.blk_first[l]; bb < g.blk_first[l + 1]; ++bb) {
				const int c = in_ring(l, bb);
				if (c > 0 && !(((sq >> l) & 1u) && c == (1 << g.blk_pts_log2[l])))
					list[(size_t)m++] = bb;
			}
		}
		p->copy_upto[g.levels] = m;
		if (hipMalloc((void **)&p->d_copy_list, sizeof(int) * (size_t)(m > 0 ? m : 1)) != hipSuccess) {
			dwtx_set_error("copy list hipMalloc failed");
			return DWTX_ERR_NOMEM;
		}
		if (hipMemcpy(p->d_copy_list, list.data(), sizeof(int) * (size_t)m, hipMemcpyHostToDevice) != hipSuccess) {
			dwtx_set_error("copy list upload failed");
			return DWTX_ERR_DEVICE;
		}
	}
	return DWTX_OK;
}

static int get_plan(dwtx_ctx *ctx, int W, int H, dwtx_linplan **out)
{
	for (dwtx_linplan *p = ctx->plans; p; p = p->next)
		if (p->W == W && p->H == H) {
			*out = p;
			return DWTX_OK;
		}
	dwtx_linplan *p = (dwtx_linplan *)calloc(1, sizeof(*p));
	if (!p)
		return DWTX_ERR_NOMEM;
	int rc;
	try {
		rc = build_plan(ctx, p, W, H);
	} catch (const std::bad_alloc &) {
		dwtx_set_error("plan tables: out of host memory");
		rc = DWTX_ERR_NOMEM;
	}
	if (rc) {
		plan_destroy(p);   // not linked into ctx->plans yet: nobody else would free it
		return rc;
	}
	p->next = ctx->plans;
	ctx->plans = p;
	*out = p;
	return DWTX_OK;
}

int dwtx_get_tiles(dwtx_ctx *ctx, int W, int H, dwtx_tiles *out)
{
	dwtx_linplan *p;
	const int rc = get_plan(ctx, W, H, &p);
	if (rc)
		return rc;
	*out = p->tiles;
	return DWTX_OK;
}

void dwtx_free_plans(dwtx_ctx *ctx)
{
	dwtx_linplan *p = ctx->plans;
	while (p) {
		dwtx_linplan *n = p->next;
		plan_destroy(p);
		p = n;
	}
	ctx->plans = nullptr;
}

// curve blocks up to the last level (below `levels`) that is copied at all: the blocks are numbered level by level
static int blocks_needed(const LinGeom &g, int levels, unsigned skip_levels)
{
	int nb = 0;
	for (int l = 0; l < levels && l < g.levels; ++l)   // (a level of whole squares only — a power-of-two square image — has nothing to copy)
		if (!((skip_levels >> l) & 1u) || g.widths[l + 1] != g.lengths[l + 1] || g.heights[l + 1] != g.lengths[l + 1])
			nb = g.blk_first[l + 1];
	return nb;
}

// levels that can hold whole 32x32 squares of ring coefficients (rows of a square are read / written as 16-byte pieces:
// the row pitch must keep them aligned)
unsigned dwtx_square_levels(int W, int H)
{
	dwtx_geom g;
	if (dwtx_geometry(&g, W, H) || (W & 3))
		return 0u;
	unsigned mask = 0;
	for (int l = 0; l < g.levels; ++l)
		if (g.lengths[l + 1] >= 64)
			mask |= 1u << l;
	return mask;
}

unsigned dwtx_levels16(int W, int H, unsigned sq_levels, int max_levels)
{
	dwtx_geom g;
	if (dwtx_geometry(&g, W, H) || g.levels < 1)
		return 0u;
	unsigned mask = 0u;
	for (int t = 0; t < max_levels && t < g.levels; ++t) {
		const int l = g.levels - 1 - t;           // ring level of lifting step t
		const int w = g.widths[l + 1], h = g.heights[l + 1];   // what that step transforms
		if (!((sq_levels >> l) & 1u) || w % 4 != 0 || (w <= 64 && h <= 64))   // (lift.hip: wide kernel, not the LDS tail)
			break;
		mask |= 1u << l;
	}
	return mask;
}

extern "C" int dwtx_linearization(dwtx_ctx *ctx, int32_t *lin, const int32_t *pyr, int W, int H, int nplanes)
{
	return dwtx_linearization_ex(ctx, lin, pyr, W, H, nplanes, 0u);
}

// skip_levels: ring levels that are NOT copied (their tiles are read from the pyramid by the entropy stage)
int dwtx_linearization_ex(dwtx_ctx *ctx, int32_t *lin, const int32_t *pyr, int W, int H, int nplanes, unsigned skip_levels,
	dwtx_p16 p16)
{
	if (!ctx || !lin || !pyr || nplanes < 1 || nplanes > 65535)
		return DWTX_ERR_ARG;
	DWTX_CHECK_DIMS(W, H);
	DWTX_ENTER(ctx);
	dwtx_linplan *p;
	int rc = get_plan(ctx, W, H, &p);
	if (rc)
		return rc;
	const LinGeom &g = p->g;
	if (p16.planes && (p16.levels & ~skip_levels))   // (those rings' whole squares are the entropy stage's to read)
		return DWTX_ERR_ARG;
	const long ps = (long)W * H;
	hipLaunchKernelGGL(k_root_copy<false>, dim3(dwtx_cdiv(g.pixels[0], 64), nplanes), dim3(64), 0, ctx->stream,
		g.widths[0], g.heights[0], lin, ps, const_cast<int *>(pyr), ps, W);
	// (a grid over blocks that are not copied would only start and end: the usual skip mask has its list of blocks)
	const bool listed = skip_levels && skip_levels == dwtx_square_levels(W, H);
	const int nb = listed ? p->copy_upto[g.levels] : blocks_needed(g, g.levels, skip_levels);
	if (nb)
		hipLaunchKernelGGL(k_ring_copy<false>, dim3(nb, dwtx_cdiv(nplanes, PLANES_PER_GROUP)), dim3(THREADS), 0, ctx->stream,
			g, p->d_blockbase, lin, ps, const_cast<int *>(pyr), ps, W, (const int *)nullptr, 1, nplanes, skip_levels,
			listed ? (const int *)p->d_copy_list : (const int *)nullptr, p16.planes, p16.planes ? p16.levels : 0u);
	DWTX_LAUNCH_CHECK();
	return DWTX_OK;
}

extern "C" int dwtx_reconstruction(dwtx_ctx *ctx, int32_t *pyr, const int32_t *lin, const int *dev_missing,
	int levels_out, int W, int H, int C, int n)
{
	return dwtx_reconstruction_ex(ctx, pyr, lin, dev_missing, levels_out, W, H, C, n, 0u);
}

// skip_levels: ring levels the decoder's entropy stage has already written into the pyramid (with their bias)
int dwtx_reconstruction_ex(dwtx_ctx *ctx, int32_t *pyr, const int32_t *lin, const int *dev_missing,
	int levels_out, int W, int H, int C, int n, unsigned skip_levels, dwtx_p16 p16)
{
	if (!ctx || !lin || !pyr || (C != 1 && C != 3) || n < 1 || n * C > 65535)
		return DWTX_ERR_ARG;
	DWTX_CHECK_DIMS(W, H);
	DWTX_ENTER(ctx);
	dwtx_linplan *p;
	int rc = get_plan(ctx, W, H, &p);
	if (rc)
		return rc;
	LinGeom g = p->g;
	if (levels_out < 0 || levels_out > g.levels)
		return DWTX_ERR_ARG;
	if (p16.planes && (levels_out != g.levels || (p16.levels & ~skip_levels)))   // whole pictures only
		return DWTX_ERR_ARG;
	const int ow = g.widths[levels_out], oh = g.heights[levels_out];
	const long lin_ps = (long)W * H;
	const long pyr_ps = (long)ow * oh;
	const int nplanes = n * C;
	hipLaunchKernelGGL(k_root_copy<true>, dim3(dwtx_cdiv(g.pixels[0], 64), nplanes), dim3(64), 0, ctx->stream,
		g.widths[0], g.heights[0], const_cast<int *>(lin), lin_ps, pyr, pyr_ps, ow);
	if (levels_out > 0) {
		g.levels = levels_out;   // only rings 0..levels_out-1 are rebuilt
		const bool listed = skip_levels && skip_levels == dwtx_square_levels(W, H);
		const int nb = listed ? p->copy_upto[levels_out] : blocks_needed(g, levels_out, skip_levels);
		if (nb)
			hipLaunchKernelGGL(k_ring_copy<true>, dim3(nb, dwtx_cdiv(nplanes, PLANES_PER_GROUP)), dim3(THREADS), 0,
				ctx->stream, g, p->d_blockbase, const_cast<int *>(lin), lin_ps, pyr, pyr_ps, ow, dev_missing, C, nplanes, skip_levels,
				listed ? (const int *)p->d_copy_list : (const int *)nullptr, p16.planes, p16.planes ? p16.levels : 0u);
	}
	DWTX_LAUNCH_CHECK();
	return DWTX_OK;
}
