// lift.hip — colour transform at the image edge and the multi-level 2-D CDF 5/3
// integer lifting transform (forward: encode.c:16-30 over cdf53.h:9-34; inverse:
// decode.c:16-30 over cdf53.h:36-61) as streaming gfx950 kernels.
//
// Layout: planar int32, plane p = image*C + channel, dense rows.
//
// Kernel shape (one level, both directions): a 64-lane wave owns 64 adjacent
// column PAIRS (x = 2k, 2k+1) and walks down a strip of row pairs.  The
// horizontal lifting step is done in registers with wave shuffles (the pair to
// the left supplies its detail value, the pair to the right its even sample),
// the vertical step is a 3-row sliding window held in registers.  No LDS, no
// re-reads: every input sample is loaded once per level (plus a 2+1 row halo
// per strip) and every output written once, each as a coalesced row segment.
//
// Edge rules (cdf53.h:13-21), expressed so the interior formula
// covers them:
//   predict at the last odd sample of an even-length line: x[N] := x[N-2]
//   update at sample 0: d[-1] := d[0]      (tdiv(2a,4) == tdiv(a,2))
//   odd-length line: the last even sample is NOT updated (not symmetric!)
#include "dwtx_internal.h"

namespace {

constexpr int WAVES = 4;          // waves per block, stacked along y
constexpr int ROWS_PER_WAVE = 32; // output row pairs per wave strip

struct LevelArgs {
	const int *src;  long src_ps;  int spitch;  // forward: input w*h      | inverse: LL (w2*h2)
	int *ll;         long ll_ps;   int llpitch; // forward: LL out (w2*h2) | inverse: output w*h
	int *det;        long det_ps;  int dpitch;  // Mallat pyramid: HL at (w2+x, y), LH at (x, h2+y), HH at (w2+x, h2+y)
	int w, h, w2, h2;
};

// ---------------------------------------------------------------- forward ---

struct FwdLane {
	int k;            // pair index
	int lane;
	bool valid;       // 2k   < w
	bool has_odd;     // 2k+1 < w
	bool right_in;    // 2k+2 < w
	bool frozen;      // w odd and 2k == w-1: even sample passes through
};

// horizontal lifting of one input row for this lane's pair -> (low, high)
__device__ __forceinline__ void fwd_row(const int *__restrict__ row, const FwdLane &L, int &lo, int &hi)
{
	int x0 = L.valid ? row[2 * L.k] : 0;
	int x1 = L.has_odd ? row[2 * L.k + 1] : 0;
	int xr = __shfl_down(x0, 1);
	if (L.lane == 63 && L.right_in)
		xr = row[2 * L.k + 2];
	if (!L.right_in)
		xr = x0;
	int d = x1 - tdiv2(x0 + xr);
	int dl = __shfl_up(d, 1);
	if (L.lane == 0 && L.k > 0) {
		int xm2 = row[2 * L.k - 2], xm1 = row[2 * L.k - 1];
		dl = xm1 - tdiv2(xm2 + x0);
	}
	if (L.k == 0)
		dl = d;
	lo = L.frozen ? x0 : x0 + tdiv4(dl + d);
	hi = d;
}

__global__ __launch_bounds__(64 * WAVES) void k_fwd_level(LevelArgs a)
{
	FwdLane L;
	L.lane = threadIdx.x & 63;
	const int wv = threadIdx.x >> 6;
	L.k = blockIdx.x * 64 + L.lane;
	const int j0 = (blockIdx.y * WAVES + wv) * ROWS_PER_WAVE;
	if (j0 >= a.h2)
		return;
	const int j1 = min(j0 + ROWS_PER_WAVE, a.h2);
	const int plane = blockIdx.z;
	L.valid = 2 * L.k < a.w;
	L.has_odd = 2 * L.k + 1 < a.w;
	L.right_in = 2 * L.k + 2 < a.w;
	L.frozen = (a.w & 1) && 2 * L.k == a.w - 1;

	const int *src = a.src + plane * a.src_ps;
	int *ll = a.ll + plane * a.ll_ps;
	int *det = a.det + plane * a.det_ps;

	int jj = j0 > 0 ? j0 - 1 : 0;
	int l0, h0;              // even row 2jj
	int pl = 0, ph = 0;      // vertical detail of the previous row pair
	fwd_row(src + (long)(2 * jj) * a.spitch, L, l0, h0);
	for (; jj < j1; ++jj) {
		const int r1 = 2 * jj + 1, r2 = r1 + 1;
		const bool odd_in = r1 < a.h;
		int l1 = 0, h1 = 0, l2 = l0, h2v = h0;
		if (odd_in)
			fwd_row(src + (long)r1 * a.spitch, L, l1, h1);
		if (r2 < a.h)
			fwd_row(src + (long)r2 * a.spitch, L, l2, h2v);
		const int dl = l1 - tdiv2(l0 + l2);
		const int dh = h1 - tdiv2(h0 + h2v);
		if (jj >= j0) {
			int sl = l0, sh = h0;
			if (odd_in) {    // an odd-height plane leaves its last even row untouched
				sl += tdiv4((jj ? pl : dl) + dl);
				sh += tdiv4((jj ? ph : dh) + dh);
			}
			if (L.valid)
				ll[(long)jj * a.llpitch + L.k] = sl;
			if (L.has_odd)
				det[(long)jj * a.dpitch + a.w2 + L.k] = sh;
			if (odd_in) {
				if (L.valid)
					det[(long)(a.h2 + jj) * a.dpitch + L.k] = dl;
				if (L.has_odd)
					det[(long)(a.h2 + jj) * a.dpitch + a.w2 + L.k] = dh;
			}
		}
		pl = dl;
		ph = dh;
		l0 = l2;
		h0 = h2v;
	}
}

// ---------------------------------------------------------------- inverse ---
//
// Columns are undone first, rows last (decode.c:21-29), so the horizontal step
// needs its neighbours' values AFTER their vertical step.  Each wave therefore
// carries one halo pair on either side: lane i works on pair kb-1+i, lanes
// 1..62 produce output (62 pairs = 124 columns per wave).

constexpr int INV_PAIRS = 62;

struct InvLane {
	int k;
	bool valid;      // 0 <= k and 2k < w
	bool has_odd;    // valid and 2k+1 < w
	bool right_in;   // 2k+2 < w
	bool frozen;     // w odd and 2k == w-1
	bool writes;     // lanes 1..62 and valid
};

// horizontal inverse of one row: (low, high) of this pair -> samples 2k, 2k+1
__device__ __forceinline__ void inv_row(int *__restrict__ row, const InvLane &L, int lo, int hi)
{
	int hl = __shfl_up(hi, 1);
	if (L.k <= 0)
		hl = hi;
	const int e = L.frozen ? lo : lo - tdiv4(hl + hi);
	int er = __shfl_down(e, 1);
	if (!L.right_in)
		er = e;
	const int o = hi + tdiv2(e + er);
	if (L.writes) {
		row[2 * L.k] = e;
		if (L.has_odd)
			row[2 * L.k + 1] = o;
	}
}

__global__ __launch_bounds__(64 * WAVES) void k_inv_level(LevelArgs a)
{
	InvLane L;
	const int lane = threadIdx.x & 63;
	const int wv = threadIdx.x >> 6;
	L.k = blockIdx.x * INV_PAIRS - 1 + lane;
	const int j0 = (blockIdx.y * WAVES + wv) * ROWS_PER_WAVE;
	if (j0 >= a.h2)
		return;
	const int j1 = min(j0 + ROWS_PER_WAVE, a.h2);
	const int plane = blockIdx.z;
	L.valid = L.k >= 0 && 2 * L.k < a.w;
	L.has_odd = L.k >= 0 && 2 * L.k + 1 < a.w;
	L.right_in = 2 * L.k + 2 < a.w;
	L.frozen = (a.w & 1) && 2 * L.k == a.w - 1;
	L.writes = L.valid && lane >= 1 && lane <= INV_PAIRS;

	const int *llp = a.src + plane * a.src_ps;
	const int *det = a.det + plane * a.det_ps;
	int *dst = a.ll + plane * a.ll_ps;
	const int kk = L.k < 0 ? 0 : L.k;

	// s/d of the current pair j, detail of pair j-1, even output row of pair j
	int sl, sh, dl = 0, dh = 0, pdl = 0, pdh = 0;
	int el, eh;
	const bool h_odd = a.h & 1;

	auto fetch = [&](int j, int &fsl, int &fsh, int &fdl, int &fdh) {
		fsl = L.valid ? llp[(long)j * a.spitch + kk] : 0;
		fsh = L.has_odd ? det[(long)j * a.dpitch + a.w2 + kk] : 0;
		const bool d_in = 2 * j + 1 < a.h;
		fdl = (d_in && L.valid) ? det[(long)(a.h2 + j) * a.dpitch + kk] : 0;
		fdh = (d_in && L.has_odd) ? det[(long)(a.h2 + j) * a.dpitch + a.w2 + kk] : 0;
	};

	// even row of pair j from s[j], d[j-1], d[j]
	auto even_of = [&](int j, int s, int dprev, int dcur) {
		if (h_odd && 2 * j == a.h - 1)
			return s;
		return s - tdiv4((j ? dprev : dcur) + dcur);
	};

	if (j0 > 0) {
		int t0, t1;
		fetch(j0 - 1, t0, t1, pdl, pdh);
	}
	fetch(j0, sl, sh, dl, dh);
	el = even_of(j0, sl, pdl, dl);
	eh = even_of(j0, sh, pdh, dh);
	for (int jj = j0; jj < j1; ++jj) {
		const int r0 = 2 * jj, r1 = r0 + 1;
		int nsl = 0, nsh = 0, ndl = 0, ndh = 0;
		int nel = el, neh = eh;          // mirror: x[h] := x[h-2]
		if (r1 + 1 < a.h) {
			fetch(jj + 1, nsl, nsh, ndl, ndh);
			nel = even_of(jj + 1, nsl, dl, ndl);
			neh = even_of(jj + 1, nsh, dh, ndh);
		}
		inv_row(dst + (long)r0 * a.llpitch, L, el, eh);
		if (r1 < a.h) {
			const int ol = dl + tdiv2(el + nel);
			const int oh = dh + tdiv2(eh + neh);
			inv_row(dst + (long)r1 * a.llpitch, L, ol, oh);
		}
		dl = ndl;
		dh = ndh;
		el = nel;
		eh = neh;
	}
}


// ------------------------------------------------------- pixels <-> planes ---

// pnm.h:69-74 (byte -> int) fused with image.h:52-65 rgb2ycocg.
__global__ __launch_bounds__(256) void k_planes_from_pixels(int *__restrict__ planes, const uint8_t *__restrict__ pix,
	long npix_per_image, int C, long total_pixels)
{
	long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
	const long stride = (long)gridDim.x * blockDim.x;
	for (; i < total_pixels; i += stride) {
		const long img = i / npix_per_image, off = i - img * npix_per_image;
		int *dst = planes + img * C * npix_per_image + off;
		if (C == 1) {
			dst[0] = pix[i];
		} else {
			const int r = pix[3 * i], g = pix[3 * i + 1], b = pix[3 * i + 2];
			const int co = r - b;
			const int t = b + tdiv2(co);
			const int cg = g - t;
			dst[0] = t + tdiv2(cg);
			dst[npix_per_image] = co;
			dst[2 * npix_per_image] = cg;
		}
	}
}

__device__ __forceinline__ int clampi(int v, int lo, int hi)
{
	return v < lo ? lo : v > hi ? hi : v;
}

// image.h:39-50 ycocg2rgb (input clamps included) + pnm.h:108 output clamp.
__global__ __launch_bounds__(256) void k_pixels_from_planes(uint8_t *__restrict__ pix, const int *__restrict__ planes,
	long npix_per_image, int C, long total_pixels)
{
	long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
	const long stride = (long)gridDim.x * blockDim.x;
	for (; i < total_pixels; i += stride) {
		const long img = i / npix_per_image, off = i - img * npix_per_image;
		const int *src = planes + img * C * npix_per_image + off;
		if (C == 1) {
			pix[i] = (uint8_t)clampi(src[0], 0, 255);
		} else {
			const int y = clampi(src[0], 0, 255);
			const int co = clampi(src[npix_per_image], -255, 255);
			const int cg = clampi(src[2 * npix_per_image], -255, 255);
			const int t = y - tdiv2(cg);
			const int g = cg + t;
			const int b = t - tdiv2(co);
			const int r = b + co;
			pix[3 * i] = (uint8_t)clampi(r, 0, 255);
			pix[3 * i + 1] = (uint8_t)clampi(g, 0, 255);
			pix[3 * i + 2] = (uint8_t)clampi(b, 0, 255);
		}
	}
}

// Integer-only synthetic frames: seed = frame index, so every box
// renders identical bytes.  kind 0 "smooth+noise", kind 1 uniform noise.
__global__ __launch_bounds__(256) void k_synth(uint8_t *__restrict__ pix, int W, int H, int C, long total,
	unsigned seed0, int kind)
{
	const long per = (long)W * H * C;
	for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
		const long img = i / per;
		long r = i - img * per;
		const unsigned k = (unsigned)(r % C);
		r /= C;
		const unsigned x = (unsigned)(r % W), y = (unsigned)(r / W);
		unsigned u = x * 0x9E3779B1u ^ y * 0x85EBCA77u ^ k * 0xC2B2AE3Du ^ (seed0 + (unsigned)img) * 0x27D4EB2Fu;
		u ^= u >> 15;
		u *= 0x2C1B3C6Du;
		u ^= u >> 12;
		u *= 0x297A2D39u;
		u ^= u >> 15;
		int p;
		if (kind) {
			p = (int)(u >> 24);
		} else {
			int tx = (int)(x % 192u) - 96, ty = (int)(y % 128u) - 64;
			p = 40 + (tx < 0 ? -tx : tx) + (ty < 0 ? -ty : ty) + 10 * (int)k + (int)(u >> 29);
		}
		pix[i] = (uint8_t)p;
	}
}

// transform steps of a W*H plane, fine to coarse: sizes[t] -> sizes[t+1]
// (encode.c:24-29: recurse while both halves are >= N0; the first step always runs)
int lift_steps(int W, int H, int *ws, int *hs)
{
	int n = 0;
	ws[0] = W;
	hs[0] = H;
	do {
		ws[n + 1] = (ws[n] + 1) >> 1;
		hs[n + 1] = (hs[n] + 1) >> 1;
		++n;
	} while (n < DWTX_MAX_LEVELS && ws[n] >= DWTX_MIN_LEN && hs[n] >= DWTX_MIN_LEN);
	return n;
}

} // namespace

extern "C" int dwtx_synth_pixels(dwtx_ctx *ctx, uint8_t *pix, int W, int H, int C, int n, unsigned seed0, int kind)
{
	if (!ctx || !pix || W < 1 || H < 1 || (C != 1 && C != 3) || n < 1)
		return DWTX_ERR_ARG;
	const long total = (long)W * H * C * n;
	const int blocks = (int)min((total + 255) / 256, (long)256 * 16);
	hipLaunchKernelGGL(k_synth, dim3(blocks), dim3(256), 0, ctx->stream, pix, W, H, C, total, seed0, kind);
	DWTX_LAUNCH_CHECK();
	return DWTX_OK;
}

extern "C" int dwtx_planes_from_pixels(dwtx_ctx *ctx, int32_t *planes, const uint8_t *pix, int W, int H, int C, int n)
{
	if (!ctx || !planes || !pix || W < 1 || H < 1 || (C != 1 && C != 3) || n < 1)
		return DWTX_ERR_ARG;
	const long npix = (long)W * H, total = npix * n;
	const int blocks = (int)min((total + 255) / 256, (long)256 * 16);
	hipLaunchKernelGGL(k_planes_from_pixels, dim3(blocks), dim3(256), 0, ctx->stream, planes, pix, npix, C, total);
	DWTX_LAUNCH_CHECK();
	return DWTX_OK;
}

extern "C" int dwtx_pixels_from_planes(dwtx_ctx *ctx, uint8_t *pix, const int32_t *planes, int W, int H, int C, int n)
{
	if (!ctx || !planes || !pix || W < 1 || H < 1 || (C != 1 && C != 3) || n < 1)
		return DWTX_ERR_ARG;
	const long npix = (long)W * H, total = npix * n;
	const int blocks = (int)min((total + 255) / 256, (long)256 * 16);
	hipLaunchKernelGGL(k_pixels_from_planes, dim3(blocks), dim3(256), 0, ctx->stream, pix, planes, npix, C, total);
	DWTX_LAUNCH_CHECK();
	return DWTX_OK;
}

extern "C" int dwtx_transformation_fwd(dwtx_ctx *ctx, int32_t *out, const int32_t *in, int W, int H, int nplanes)
{
	if (!ctx || !out || !in || W < 2 || H < 2 || nplanes < 1 || nplanes > 65535)
		return DWTX_ERR_ARG;
	int ws[DWTX_MAX_LEVELS + 2], hs[DWTX_MAX_LEVELS + 2];
	const int T = lift_steps(W, H, ws, hs);
	int *tmp[2] = { nullptr, nullptr };
	if (T > 1) {
		tmp[0] = (int *)dwtx_scratch(ctx, SLOT_LIFT_A, sizeof(int) * (size_t)ws[1] * hs[1] * nplanes);
		if (!tmp[0])
			return DWTX_ERR_NOMEM;
	}
	if (T > 2) {
		tmp[1] = (int *)dwtx_scratch(ctx, SLOT_LIFT_B, sizeof(int) * (size_t)ws[2] * hs[2] * nplanes);
		if (!tmp[1])
			return DWTX_ERR_NOMEM;
	}
	const long full_ps = (long)W * H;
	for (int t = 0; t < T; ++t) {
		LevelArgs a;
		a.w = ws[t];
		a.h = hs[t];
		a.w2 = ws[t + 1];
		a.h2 = hs[t + 1];
		if (t == 0) {
			a.src = in;
			a.src_ps = full_ps;
			a.spitch = W;
		} else {
			a.src = tmp[(t - 1) & 1];
			a.src_ps = (long)ws[t] * hs[t];
			a.spitch = ws[t];
		}
		if (t == T - 1) {
			a.ll = out;
			a.ll_ps = full_ps;
			a.llpitch = W;
		} else {
			a.ll = tmp[t & 1];
			a.ll_ps = (long)a.w2 * a.h2;
			a.llpitch = a.w2;
		}
		a.det = out;
		a.det_ps = full_ps;
		a.dpitch = W;
		dim3 grid(dwtx_cdiv(a.w2, 64), dwtx_cdiv(a.h2, WAVES * ROWS_PER_WAVE), nplanes);
		hipLaunchKernelGGL(k_fwd_level, grid, dim3(64 * WAVES), 0, ctx->stream, a);
		DWTX_LAUNCH_CHECK();
	}
	return DWTX_OK;
}

extern "C" int dwtx_transformation_inv(dwtx_ctx *ctx, int32_t *out, const int32_t *in, int W, int H, int nplanes)
{
	if (!ctx || !out || !in || W < 2 || H < 2 || nplanes < 1 || nplanes > 65535)
		return DWTX_ERR_ARG;
	int ws[DWTX_MAX_LEVELS + 2], hs[DWTX_MAX_LEVELS + 2];
	const int T = lift_steps(W, H, ws, hs);
	int *tmp[2] = { nullptr, nullptr };
	if (T > 1) {
		tmp[1] = (int *)dwtx_scratch(ctx, SLOT_LIFT_A, sizeof(int) * (size_t)ws[1] * hs[1] * nplanes);
		if (!tmp[1])
			return DWTX_ERR_NOMEM;
	}
	if (T > 2) {
		tmp[0] = (int *)dwtx_scratch(ctx, SLOT_LIFT_B, sizeof(int) * (size_t)ws[2] * hs[2] * nplanes);
		if (!tmp[0])
			return DWTX_ERR_NOMEM;
	}
	const long full_ps = (long)W * H;
	// step t rebuilds the ws[t]*hs[t] plane; its output goes to tmp[t&1] (t odd: the big one)
	for (int t = T - 1; t >= 0; --t) {
		LevelArgs a;
		a.w = ws[t];
		a.h = hs[t];
		a.w2 = ws[t + 1];
		a.h2 = hs[t + 1];
		if (t == T - 1) {
			a.src = in;
			a.src_ps = full_ps;
			a.spitch = W;
		} else {
			a.src = tmp[(t + 1) & 1];
			a.src_ps = (long)a.w2 * a.h2;
			a.spitch = a.w2;
		}
		if (t == 0) {
			a.ll = out;
			a.ll_ps = full_ps;
			a.llpitch = W;
		} else {
			a.ll = tmp[t & 1];
			a.ll_ps = (long)a.w * a.h;
			a.llpitch = a.w;
		}
		a.det = const_cast<int *>(in);
		a.det_ps = full_ps;
		a.dpitch = W;
		dim3 grid(dwtx_cdiv(a.w2, INV_PAIRS), dwtx_cdiv(a.h2, WAVES * ROWS_PER_WAVE), nplanes);
		hipLaunchKernelGGL(k_inv_level, grid, dim3(64 * WAVES), 0, ctx->stream, a);
		DWTX_LAUNCH_CHECK();
	}
	return DWTX_OK;
}
