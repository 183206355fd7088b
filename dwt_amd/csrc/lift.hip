// lift.hip — colour transform at the image edge and the multi-level 2-D CDF 5/3
// integer lifting transform (forward: encode.c:16-30 over cdf53.h:9-34; inverse:
// decode.c:16-30 over cdf53.h:36-61) as streaming gfx950 kernels.
//
// Layout: planar int32, plane p = image*C + channel, dense rows; the finest level can
// also read / write 8-bit pixels directly (gray or interleaved RGB with YCoCg-R fused).
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

#include <stdlib.h>

namespace {

constexpr int WAVES = 4;          // waves per block, stacked along y
// The wide forward kernel: 2^wx_log2 of a block's waves sit side by side, the rest on top of each other.  Side by side
// the strips' left / right halo columns are lines the neighbour wave on the same CU loads at the same time, and a block
// reads rows of up to 4 KB in one piece: forward 34.1 -> 31.9 us per 4096x4096 int32 plane, the RGB finest level 2.93 ->
// 2.31 ms per 256 frames of 1080p on two of three boxes, no change on the third (never slower).  (The inverse, whose waves overlap by eight lanes instead of loading halo columns,
// loses with the same arrangement — 36.1 -> 39.8 us — and keeps its four waves stacked.)
constexpr int MAX_ROWS_PER_WAVE = 64; // output row pairs per wave strip (fewer on small levels, to keep the chip full); 64 instead of 32: half the halo rows, forward 37 -> 34.5 us per 4096x4096 plane

struct LevelArgs {
	const int *src;  long src_ps;  int spitch;  // forward: input w*h      | inverse: LL (w2*h2)
	int *ll;         long ll_ps;   int llpitch; // forward: LL out (w2*h2) | inverse: output w*h
	int *det;        long det_ps;  int dpitch;  // Mallat pyramid: HL at (w2+x, y), LH at (x, h2+y), HH at (w2+x, h2+y)
	int w, h, w2, h2;
	int rpw;          // output row pairs per wave strip
	const uint8_t *src8;   // forward, finest level of a gray image: 8-bit pixels instead of src (pnm.h:69-74 widening fused)
	uint8_t *dst8;         // inverse, finest level of a gray image: clamped 8-bit pixels instead of ll (pnm.h:108 fused)
	short *det16;          // or null: this level's detail bands live here as 16-bit values (same positions, pitch and plane stride as det)
	const short *src16;    // forward, or null: the input band as 16-bit values (pitch and plane stride of src)
	short *ll16;           // forward, or null: the LL band goes out as 16-bit values (pitch and plane stride of ll)
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
	const int j0 = (blockIdx.y * WAVES + wv) * a.rpw;
	if (j0 >= a.h2)
		return;
	const int j1 = min(j0 + a.rpw, a.h2);
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
// wide inverse: 56 quads of output per wave = 896 bytes per row = seven whole 128-byte lines, so the
// row stores of neighbouring waves never share a line (lanes 0 and 57 are the halo, 58-63 idle)
constexpr int INV_QUADS = 56;

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
	const int j0 = (blockIdx.y * WAVES + wv) * a.rpw;
	if (j0 >= a.h2)
		return;
	const int j1 = min(j0 + a.rpw, a.h2);
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


// ------------------------------------------------------ wide (16-byte) path ---
// Same algorithm with two column pairs per lane: 16-byte loads of the input rows,
// 8-byte stores of each subband row (forward) / 8-byte subband loads and 16-byte
// stores (inverse), and the next rows' loads issued one loop iteration ahead of
// their use.  Needs w % 4 == 0 and 16-byte aligned rows; other shapes take the
// narrow kernels above.

struct I2 {
	int a, b;
};

// The tiles' magnitude histograms, dropped by the forward kernel while it holds the coefficients (dwtx_hist_sink):
// the level's blocks of 32x32 pyramid positions are the entropy stage's tiles.
struct HistArgs {
	unsigned *cum32;       // [plane][NT][16]
	unsigned *tile_mx;     // [plane][NTP]
	const int *xy2tile;    // this level's [by * nbs + bx] -> tile
	int NT, NTP, nbs;
};

struct LevelArgsW {
	LevelArgs a;
	int nquads;       // w / 4
	int wx_log2;      // forward: 1 << wx_log2 waves of a block side by side
	HistArgs hist;
};

// Per lane and subband: counts of "magnitude below 2^q" for q = 0..15 — nibbles of R for the last few coefficients
// (adding 0x1111.. << 4t per coefficient, t its bit count), folded into the bytes of ev (even q) and od (odd q) before a
// nibble can overflow; mx = OR of the magnitudes.  The same counts k_hist (pack.hip) makes from memory.
struct HistAcc {
	unsigned long long R, ev, od;
	unsigned mx;
};

__device__ __forceinline__ void hist_add(HistAcc &h, int v)
{
	const unsigned a = (unsigned)(v < 0 ? -v : v);
	h.mx |= a;
	// (a magnitude of 2^15 and more counts as 15 bits here: its plane has more than 16 bit planes — mx says so — and
	// is refused before any of these counts is used)
	const unsigned t = min(32u - (unsigned)__clz((int)a), 15u);
	h.R += 0x1111111111111111ull << (4u * t);
}

__device__ __forceinline__ void hist_fold(HistAcc &h)
{
	h.ev += h.R & 0x0f0f0f0f0f0f0f0full;
	h.od += (h.R >> 4) & 0x0f0f0f0f0f0f0f0full;
	h.R = 0;
}

// sum over the 16 lanes of a DPP row (its last lane ends up with the total)
__device__ __forceinline__ unsigned row16_add(unsigned v)
{
	v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
	v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
	v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
	v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
	return v;
}

__device__ __forceinline__ unsigned row16_or(unsigned v)
{
	v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
	v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
	v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
	v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
	return v;
}

// The 16 lanes of a row hold one block's columns: their counts, summed, are added to the block's tile (blocks that
// straddle two strips or two subbands get several such contributions).  All lanes of the wave take part.
__device__ __forceinline__ void hist_flush(HistAcc &h, const HistArgs &H, int plane, int bx, int by, int lane)
{
	hist_fold(h);
	unsigned D[8];
#pragma unroll
	for (int b = 0; b < 8; ++b)
		D[b] = row16_add(((unsigned)(h.ev >> (8 * b)) & 0xffu) | ((unsigned)(h.od >> (8 * b)) & 0xffu) << 16);
	const unsigned mx = row16_or(h.mx);
	if ((lane & 15) == 15 && bx < H.nbs && by < H.nbs) {
		const int tile = H.xy2tile[by * H.nbs + bx];
		if (tile >= 0) {
			unsigned *rec = H.cum32 + ((long)plane * H.NT + tile) * 16;
#pragma unroll
			for (int b = 0; b < 8; ++b)
				if (D[b])
					atomicAdd(rec + b, D[b]);
			if (mx)
				atomicOr(H.tile_mx + (long)plane * H.NTP + tile, mx);
		}
	}
	h.ev = h.od = 0;
	h.mx = 0;
}

struct FwdRaw {
	int4 x;
	int xr;           // x[4q+4] for lane 63
	int2 left;        // x[4q-2], x[4q-1] for lane 0
};

// Where the finest forward level may read its samples instead of int32 planes: 8-bit gray pixels (pnm.h:69-74 fused)
// or 8-bit interleaved RGB pixels, of which each plane's launch takes its own YCoCg-R channel (image.h:52-65 fused).
struct Rgb8 {};
template <typename SrcT>
struct SrcTag {};

// (8-bit rows wait for their turn as they were loaded — four pixels in a word, the neighbours' in another — and are
// widened when used)
struct FwdRaw8 {
	unsigned v;       // pixels 4q .. 4q+3
	unsigned edge;    // the word after them (lane 63: its first byte is x[4q+4]) or before them (lane 0: its last two are x[4q-2], x[4q-1])
};

// (RGB rows too: the four pixels' three words and two words of neighbours; the colour transform of image.h:52-65
// happens when the row is used)
struct FwdRawRgb {
	unsigned a, b, c;   // pixels 4q .. 4q+3
	unsigned e0, e1;    // lane 63: e0 = the word with pixel 4q+4; lane 0: bytes 12q-8 .. 12q-1 (pixels 4q-2 and 4q-1 are the last six)
};

__device__ __forceinline__ void fwd_lift_w(const FwdRaw &r, int q, int lane, int nquads, I2 &lo, I2 &hi)
{
	int xr = __shfl_down(r.x.x, 1);
	if (lane == 63)
		xr = r.xr;
	if (q + 1 >= nquads)
		xr = r.x.z;                       // x[w] := x[w-2]
	const int d0 = r.x.y - tdiv2(r.x.x + r.x.z);
	const int d1 = r.x.w - tdiv2(r.x.z + xr);
	int dl = __shfl_up(d1, 1);
	if (lane == 0)
		dl = r.left.y - tdiv2(r.left.x + r.x.x);
	if (q == 0)
		dl = d0;                          // d[-1] := d[0]
	lo.a = r.x.x + tdiv4(dl + d0);
	lo.b = r.x.z + tdiv4(d0 + d1);
	hi.a = d0;
	hi.b = d1;
}

__device__ __forceinline__ I2 i2_pred(I2 odd, I2 e0, I2 e2)
{
	I2 r = { odd.a - tdiv2(e0.a + e2.a), odd.b - tdiv2(e0.b + e2.b) };
	return r;
}

__device__ __forceinline__ I2 i2_upd(I2 even, I2 dprev, I2 d)
{
	I2 r = { even.a + tdiv4(dprev.a + d.a), even.b + tdiv4(dprev.b + d.b) };
	return r;
}

__device__ __forceinline__ void st2(int *p, I2 v)
{
	*reinterpret_cast<int2 *>(p) = make_int2(v.a, v.b);
}


// first sample of the plane's source and the channel a launch extracts (RGB only)
__device__ __forceinline__ const uint8_t *fwd_base(SrcTag<uint8_t>, const LevelArgs &a, int plane, int &ch)
{
	ch = 0;
	return a.src8 + plane * a.src_ps;
}
__device__ __forceinline__ const uint8_t *fwd_base(SrcTag<Rgb8>, const LevelArgs &a, int plane, int &ch)
{
	ch = plane % 3;
	return a.src8 + (plane / 3) * a.src_ps;   // src_ps = bytes per interleaved image, spitch = bytes per row
}

// Workgroups are dealt round-robin over the 8 XCDs (each with an L2 of its own): with the plain mapping the
// strips left and right of a strip — whose edge sectors it also loads as halo — sit on other XCDs and those
// sectors come from HBM a second time.  Remap so that an XCD owns a contiguous run of strips (whole bands of
// rows): workgroup D of a plane works on strip (D mod 8) * (N/8) + D / 8.  Speed only; any mapping is correct.
__device__ __forceinline__ void xcd_strip(int &bx, int &by)
{
	bx = blockIdx.x;
	by = blockIdx.y;
	const int n = gridDim.x * gridDim.y;
	if ((n & 7) == 0) {
		const int d = bx + gridDim.x * by;
		const int s = (d & 7) * (n >> 3) + (d >> 3);
		by = s / gridDim.x;
		bx = s - by * gridDim.x;
	}
}

// The RGB source: every plane's workgroup reads all three bytes of its pixels, so the three channels of a strip are
// made neighbours in time ON ONE XCD (grid.x holds three workgroups per strip, grid.z the images): the second and the
// third find the pixels in that XCD's L2 instead of fetching them from HBM again.
__device__ __forceinline__ void xcd_strip_rgb(int &bx, int &by, int &c)
{
	const int gx = gridDim.x / 3, n = gx * gridDim.y;
	const int d = blockIdx.x + gridDim.x * blockIdx.y;   // 0 .. 3n-1; workgroup d runs on XCD d mod 8
	int s;
	if ((n & 7) == 0) {
		const int slot = d >> 3;
		s = (d & 7) * (n >> 3) + slot / 3;
		c = slot % 3;
	} else {
		s = d / 3;
		c = d % 3;
	}
	by = s / gx;
	bx = s - by * gx;
}

template <typename SrcT>
struct IsRgb {
	static constexpr bool value = false;
};
template <>
struct IsRgb<Rgb8> {
	static constexpr bool value = true;
};

// A row moves from the registers it was loaded into to the registers it is used from: a move the register allocator
// cannot fold away, placed where the wave is to wait for the row (see the loop of k_fwd_level_w).
__device__ __forceinline__ unsigned hold(unsigned v)
{
	unsigned o;
	asm volatile("v_mov_b32 %0, %1" : "=v"(o) : "v"(v));
	return o;
}
__device__ __forceinline__ int hold(int v) { return (int)hold((unsigned)v); }

// One int32 row of the wave's strip as it is loaded: every lane loads — lanes beyond the row from the row's last quad,
// the edge pair from a clamped place — so that no load sits behind a branch or feeds a select (either would make the
// wave wait for it at once); what the extra lanes get is never used.
struct FwdRawI {
	int4 x;           // x[4q .. 4q+3]
	int2 e;           // lane 0: x[4q-2], x[4q-1]; the other lanes: e.x = x[4q+4]
};
struct LaneAtI {
	int main, edge;   // offsets in a row, in samples
};

__device__ __forceinline__ LaneAtI lane_at_i(int q, int lane, int nquads)
{
	const int qa = min(q, nquads - 1);
	LaneAtI o = { 4 * qa, lane == 0 ? max(4 * qa - 2, 0) : min(4 * qa + 4, 4 * nquads - 2) };
	return o;
}

__device__ __forceinline__ FwdRawI fwd_load_i(const int *__restrict__ row, const LaneAtI &at)
{
	FwdRawI r;
	r.x = *reinterpret_cast<const int4 *>(row + at.main);
	r.e = *reinterpret_cast<const int2 *>(row + at.edge);
	return r;
}

// the same row when the band is kept as 16-bit values (the levels of an 8-bit source whose range allows it)
struct FwdRawS {
	uint2 x;          // x[4q .. 4q+3]
	unsigned e;       // lane 0: x[4q-2], x[4q-1]; the other lanes: its low half = x[4q+4]
};

__device__ __forceinline__ FwdRawS fwd_load_i(const short *__restrict__ row, const LaneAtI &at)
{
	FwdRawS r;
	r.x = *reinterpret_cast<const uint2 *>(row + at.main);
	r.e = *reinterpret_cast<const unsigned *>(row + at.edge);
	return r;
}

__device__ __forceinline__ FwdRaw widen(const FwdRawI &r)
{
	FwdRaw o;
	o.x = r.x;
	o.xr = r.e.x;
	o.left = r.e;
	return o;
}

__device__ __forceinline__ FwdRaw widen(const FwdRawS &r)
{
	FwdRaw o;
	o.x = make_int4((int)(short)(r.x.x & 0xffffu), (int)r.x.x >> 16, (int)(short)(r.x.y & 0xffffu), (int)r.x.y >> 16);
	o.xr = (int)(short)(r.e & 0xffffu);
	o.left = make_int2(o.xr, (int)r.e >> 16);
	return o;
}

__device__ __forceinline__ FwdRaw hold(const FwdRawS &r)
{
	FwdRawS h;
	h.x = make_uint2(hold(r.x.x), hold(r.x.y));
	h.e = hold(r.e);
	return widen(h);
}

__device__ __forceinline__ void st2(short *p, I2 v)
{
	*reinterpret_cast<unsigned *>(p) = ((unsigned)v.a & 0xffffu) | ((unsigned)v.b << 16);
}

// the band a forward level reads, as the kernel variant sees it
template <bool P16>
struct SrcBand {
	typedef const int *ptr;
	typedef FwdRawI raw;
	static __device__ __forceinline__ ptr of(const LevelArgs &a, long plane) { return a.src + plane * a.src_ps; }
};
template <>
struct SrcBand<true> {
	typedef const short *ptr;
	typedef FwdRawS raw;
	static __device__ __forceinline__ ptr of(const LevelArgs &a, long plane) { return a.src16 + plane * a.src_ps; }
};

__device__ __forceinline__ FwdRaw hold(const FwdRawI &r)
{
	FwdRaw o;
	o.x = make_int4(hold(r.x.x), hold(r.x.y), hold(r.x.z), hold(r.x.w));
	o.xr = hold(r.e.x);
	o.left = make_int2(o.xr, hold(r.e.y));
	return o;
}

// Forward level on int32 planes (every level of dwtx_transformation_fwd; the levels below the finest in the codec).
// Memory operations retire in order on this part (one counter for loads and stores): a wave that waits for rows it
// loaded also waits for everything it issued before them, and a wait the compiler cannot count exactly waits for
// everything.  So the loop works in batches of S row pairs: wait once (where the rows are moved to the registers they
// are used from), send the previous batch's results out, ask for the next batch's rows, then compute S row pairs
// without touching memory — by the next wait both the stores and the loads are a whole batch of arithmetic old.
// P16: the level's input band and its detail bands are 16-bit values (levels 2..5 of an 8-bit source in the codec: with
// |x| <= 255 a sample of level k's input stays below 255 * 2.25^(k-1) and its details below four times that — the
// low-pass of cdf53.h:9-34 has an l1 norm of 1.5 per direction, the high-pass of 2 — i.e. 26 142 on the fifth level;
// that bound is loose: the composed five-level response has an l1 norm of 7.95, 2 028 for 8-bit samples, and
// tests/test_codec_gpu.py builds the picture that gets there); the arithmetic is int32 either way.
template <bool HIST, bool P16>
__global__ __launch_bounds__(64 * WAVES) void k_fwd_level_w(LevelArgsW A)
{
	const LevelArgs &a = A.a;
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	int bx, by;
	xcd_strip(bx, by);
	const int sx = (bx << A.wx_log2) + (wv & ((1 << A.wx_log2) - 1));   // the wave's strip of 64 quads
	const int q = sx * 64 + lane;
	const int j0 = (by * (WAVES >> A.wx_log2) + (wv >> A.wx_log2)) * a.rpw;
	if (j0 >= a.h2 || sx * 64 >= A.nquads)
		return;
	const int j1 = min(j0 + a.rpw, a.h2);
	const int plane = blockIdx.z;
	const bool valid = q < A.nquads;
	const typename SrcBand<P16>::ptr src = SrcBand<P16>::of(a, plane);
	int *ll = a.ll + plane * a.ll_ps;
	short *ll16 = a.ll16 ? a.ll16 + plane * a.ll_ps : nullptr;   // (uniform)
	int *det = a.det + plane * a.det_ps;
	short *det16 = P16 ? a.det16 + plane * a.det_ps : nullptr;

	constexpr int S = 2;
	const int jfirst = j0 > 0 ? j0 - 1 : 0;
	const LaneAtI at = lane_at_i(q, lane, A.nquads);
	I2 l0, h0, pl = { 0, 0 }, ph = { 0, 0 };
	fwd_lift_w(widen(fwd_load_i(src + (long)(2 * jfirst) * a.spitch, at)), q, lane, A.nquads, l0, h0);
	auto rowp = [&](int r) { return src + (long)min(r, a.h - 1) * a.spitch; };
	FwdRaw cur[2 * S];
	typename SrcBand<P16>::raw nxt[2 * S];
#pragma unroll
	for (int k = 0; k < 2 * S; ++k)
		nxt[k] = fwd_load_i(rowp(2 * jfirst + 1 + k), at);
	I2 osl[S], osh[S], odl[S], odh[S];   // a batch's results wait here for the next iteration's stores
	auto store_batch = [&](int jb) {
#pragma unroll
		for (int s = 0; s < S; ++s) {
			const int j = jb + s;
			if (j >= j0 && j < j1 && valid) {
				if (ll16)
					st2(ll16 + (long)j * a.llpitch + 2 * q, osl[s]);
				else
					st2(ll + (long)j * a.llpitch + 2 * q, osl[s]);
				if (P16) {
					st2(det16 + (long)j * a.dpitch + a.w2 + 2 * q, osh[s]);
					if (2 * j + 1 < a.h) {
						st2(det16 + (long)(a.h2 + j) * a.dpitch + 2 * q, odl[s]);
						st2(det16 + (long)(a.h2 + j) * a.dpitch + a.w2 + 2 * q, odh[s]);
					}
				} else {
					st2(det + (long)j * a.dpitch + a.w2 + 2 * q, osh[s]);
					if (2 * j + 1 < a.h) {
						st2(det + (long)(a.h2 + j) * a.dpitch + 2 * q, odl[s]);
						st2(det + (long)(a.h2 + j) * a.dpitch + a.w2 + 2 * q, odh[s]);
					}
				}
			}
		}
	};
	HistAcc hHL = { 0, 0, 0, 0 }, hLH = { 0, 0, 0, 0 }, hHH = { 0, 0, 0, 0 };
	for (int jb = jfirst; jb < j1; jb += S) {
#pragma unroll
		for (int k = 0; k < 2 * S; ++k)
			cur[k] = hold(nxt[k]);   // the one wait of the iteration: everything outstanding is a batch old
		if (jb > jfirst)
			store_batch(jb - S);
		if (jb + S < j1) {
#pragma unroll
			for (int k = 0; k < 2 * S; ++k)
				nxt[k] = fwd_load_i(rowp(2 * (jb + S) + 1 + k), at);
		}
#pragma unroll
		for (int s = 0; s < S; ++s) {
			const int jj = jb + s;
			if (jj >= j1)
				break;
			const int r1 = 2 * jj + 1, r2 = r1 + 1;
			const bool odd_in = r1 < a.h;
			I2 l1 = { 0, 0 }, h1 = { 0, 0 }, l2 = l0, h2v = h0;
			if (odd_in)
				fwd_lift_w(cur[2 * s], q, lane, A.nquads, l1, h1);
			if (r2 < a.h)
				fwd_lift_w(cur[2 * s + 1], q, lane, A.nquads, l2, h2v);
			const I2 dl = i2_pred(l1, l0, l2);
			const I2 dh = i2_pred(h1, h0, h2v);
			I2 sl = l0, sh = h0;
			if (odd_in) {
				sl = i2_upd(l0, jj ? pl : dl, dl);
				sh = i2_upd(h0, jj ? ph : dh, dh);
			}
			osl[s] = sl;
			osh[s] = sh;
			odl[s] = dl;
			odh[s] = dh;
			if (HIST && jj >= j0) {
				// the detail coefficients of this row pair (cdf53.h:9-34 output): HL row jj, LH and HH row h2 + jj
				if (valid) {
					hist_add(hHL, sh.a);
					hist_add(hHL, sh.b);
					if (odd_in) {
						hist_add(hLH, dl.a);
						hist_add(hLH, dl.b);
						hist_add(hHH, dh.a);
						hist_add(hHH, dh.b);
					}
				}
				if ((jj & 3) == 3) {   // eight coefficients per subband since the last fold: a nibble holds fifteen
					hist_fold(hHL);
					hist_fold(hLH);
					hist_fold(hHH);
				}
				// a block ends where its 32 rows end (or the strip does): the rows of HL are jj, those of LH / HH h2 + jj
				const bool last = jj == j1 - 1;
				const int bxl = (2 * q) >> 5, bxh = (a.w2 + 2 * q) >> 5;
				if (last || ((jj + 1) & 31) == 0)
					hist_flush(hHL, A.hist, plane, bxh, jj >> 5, lane);
				if (last || ((a.h2 + jj + 1) & 31) == 0) {
					hist_flush(hLH, A.hist, plane, bxl, (a.h2 + jj) >> 5, lane);
					hist_flush(hHH, A.hist, plane, bxh, (a.h2 + jj) >> 5, lane);
				}
			}
			pl = dl;
			ph = dh;
			l0 = l2;
			h0 = h2v;
		}
	}
	store_batch(jfirst + (j1 - 1 - jfirst) / S * S);   // the last batch (a strip has at least one row pair)
}

// ---- two levels per pass (forward, int32 planes) ----
// One launch per level moves every LL band twice more than the transform needs: written by level k, read by level k+1
// (16 B x sum 4^-k = 21.33 B per sample forward + inverse against 16 algorithmic).  Here a wave runs level k as above and
// hands each LL row — the lane's two LL samples are exactly one column PAIR of level k+1 — straight to a second lifting
// stage in registers: the LL band of level k never exists in memory.
// Seams.  Level k+1 of a lane needs the LL samples of the lanes beside it, which need the input samples beside theirs.
// Instead of halo loads the waves OVERLAP: lane 1's level-k results are right without any edge load (lane 0 supplies d,
// lane 2 its even sample), lane 2's level-k+1 results need lane 1's; on the right one lane is enough.  Vertically a
// strip's first row pair of level k+1 needs the pair before it: three row pairs of level k ahead of the strip, one behind.
// What decides the kernel's speed is how its STORES meet the 128-byte lines (round 4's ablation, 64 planes of 4096x4096:
// a pass that only reads takes 13 us per plane, the level-k detail stores add 12-16, level k+1's 4-byte stores 4-7 —
// reads and writes do not overlap, a written byte costs 1.5 read bytes, and owning 61 lanes of 64 (rows of 488 bytes
// at multiples of 488) was 17 % SLOWER than one launch per level although it moved 14 % fewer bytes): a wave owns the
// outputs of 48 lanes — rows of 384 bytes (level k) and 192 bytes (level k+1; the block's four waves side by side make
// whole lines of them) at multiples of themselves; lanes 8 .. 56 take part (the lanes before them keep the loads on
// whole lines, the lanes after them repeat lane 56's quad), and the stores bypass the L2's allocation (nontemporal: 5 %).
// Shapes: w % 4 == 0 (a lane's quad) and h % 4 == 0 (both levels have even heights: every row pair is whole);
// everything else takes one launch per level.
struct Level2Args {
	const int *src;  long src_ps;  int spitch;   // level k input, w x h
	int *ll2;        long ll2_ps;  int ll2pitch; // LL of level k+1, w/4 x h/4
	int *det;        long det_ps;  int dpitch;   // the pyramid: detail bands of both levels (Mallat layout)
	int w, h, nquads;
	int mpw;          // level k+1 row pairs per wave strip
};
// (strips of 8 row pairs of level k+1 — 32 input rows — measured best, against 4, 16, 32 and 64: the halo rows a strip shares with its
// neighbours are then still in the XCD's L2 when the neighbour asks for them, and there are eight times the waves to hide latency behind)
constexpr int F2_MPW = 8;
constexpr int F2_FIRST = 8, F2_OWN = 48, F2_ACTIVE = F2_FIRST + F2_OWN + 1;   // lanes F2_FIRST .. F2_FIRST + F2_OWN - 1 own a wave's outputs; lanes from F2_ACTIVE on only repeat the last active lane's loads

// cdf53.h:9-34 along the row for a lane's quad, neighbours by shuffle only (lanes 0 and 63 get wrong values where they
// would need a lane that is not there: lo.a of lane 0, lo.b / hi.b of lane 63 — never used, see above)
__device__ __forceinline__ void fwd_lift_q(const int4 &x, int q, int nquads, I2 &lo, I2 &hi)
{
	int xr = __shfl_down(x.x, 1);
	if (q + 1 >= nquads)
		xr = x.z;                         // x[w] := x[w-2]
	const int d0 = x.y - tdiv2(x.x + x.z);
	const int d1 = x.w - tdiv2(x.z + xr);
	int dl = __shfl_up(d1, 1);
	if (q <= 0)
		dl = d0;                          // d[-1] := d[0]
	lo.a = x.x + tdiv4(dl + d0);
	lo.b = x.z + tdiv4(d0 + d1);
	hi.a = d0;
	hi.b = d1;
}

// the same one level down: the lane's LL pair (x0, x1) of a row -> (low, high)
__device__ __forceinline__ void fwd_lift_pair(int x0, int x1, int q, int nquads, int &lo, int &hi)
{
	int xr = __shfl_down(x0, 1);
	if (q + 1 >= nquads)
		xr = x0;
	const int d = x1 - tdiv2(x0 + xr);
	int dl = __shfl_up(d, 1);
	if (q <= 0)
		dl = d;
	lo = x0 + tdiv4(dl + d);
	hi = d;
}

__device__ __forceinline__ int4 hold(const int4 &v) { return make_int4(hold(v.x), hold(v.y), hold(v.z), hold(v.w)); }
__device__ __forceinline__ void st2nt(int *p, I2 v)
{
	typedef int v2i __attribute__((ext_vector_type(2)));
	v2i x = { v.a, v.b };
	__builtin_nontemporal_store(x, reinterpret_cast<v2i *>(p));
}

__global__ __launch_bounds__(64 * WAVES) void k_fwd2_level_w(Level2Args a)
{
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	int bx, by;
	xcd_strip(bx, by);
	const int strip = bx * WAVES + wv;                      // the block's waves side by side
	if (strip * F2_OWN >= a.nquads)
		return;
	const int q = strip * F2_OWN - F2_FIRST + lane;         // the lane's quad of level k = its column pair of level k+1
	const int h2 = a.h >> 1, h4 = a.h >> 2, w2 = a.w >> 1, w4 = a.w >> 2;
	const int m0 = by * a.mpw;
	if (m0 >= h4)
		return;
	const int m1 = min(m0 + a.mpw, h4);
	const int plane = blockIdx.z;
	const bool own = lane >= F2_FIRST && lane < F2_FIRST + F2_OWN && q < a.nquads;
	// (every lane loads, from a clamped place: the lanes beyond the right-hand halo lane repeat its quad — no line of their own)
	const int *src = a.src + plane * a.src_ps + 4 * min(max(q - max(lane - (F2_ACTIVE - 1), 0), 0), a.nquads - 1);
	int *ll2 = a.ll2 + plane * a.ll2_ps;
	int *det = a.det + plane * a.det_ps;

	// LL rows of level k that the strip's row pairs m0 .. m1-1 of level k+1 need: 2 m0 - 2 (for the pair before: its
	// detail enters the first update) .. 2 m1 (the last predict; the plane's last row pair mirrors instead)
	const int r_lo = max(2 * m0 - 2, 0), r_hi = min(2 * m1, h2 - 1);
	const int jfirst = max(r_lo - 1, 0);        // level k: one row pair more, for its detail
	constexpr int S = 2;
	auto rowp = [&](int r) { return src + (long)min(r, a.h - 1) * a.spitch; };
	I2 l0, h0, pl = { 0, 0 }, ph = { 0, 0 };
	fwd_lift_q(*reinterpret_cast<const int4 *>(rowp(2 * jfirst)), q, a.nquads, l0, h0);
	int4 cur[2 * S], nxt[2 * S];
#pragma unroll
	for (int k = 0; k < 2 * S; ++k)
		nxt[k] = *reinterpret_cast<const int4 *>(rowp(2 * jfirst + 1 + k));
	// what a batch leaves for the next iteration's stores
	I2 osh[S], odl[S], odh[S];
	int o2[4] = { 0, 0, 0, 0 }, o2m = -1;   // a finished row pair of level k+1: LL, HL, LH, HH and its index
	auto store_batch = [&](int jb) {
		if (!own)
			return;
#pragma unroll
		for (int s = 0; s < S; ++s) {
			const int j = jb + s;
			if (j >= 2 * m0 && j < 2 * m1) {
				st2nt(det + (long)j * a.dpitch + w2 + 2 * q, osh[s]);
				st2nt(det + (long)(h2 + j) * a.dpitch + 2 * q, odl[s]);
				st2nt(det + (long)(h2 + j) * a.dpitch + w2 + 2 * q, odh[s]);
			}
		}
		if (o2m >= m0) {
			__builtin_nontemporal_store(o2[0], ll2 + (long)o2m * a.ll2pitch + q);
			__builtin_nontemporal_store(o2[1], det + (long)o2m * a.dpitch + w4 + q);
			__builtin_nontemporal_store(o2[2], det + (long)(h4 + o2m) * a.dpitch + q);
			__builtin_nontemporal_store(o2[3], det + (long)(h4 + o2m) * a.dpitch + w4 + q);
		}
		o2m = -1;
	};
	// level k+1, column direction: A = the pair's even row, B = its odd row, (pd*) the details of the pair before
	int Al = 0, Ah = 0, Bl = 0, Bh = 0, pdl = 0, pdh = 0;
	auto finish_pair = [&](int m, int Cl, int Ch) {
		const int dl2 = Bl - tdiv2(Al + Cl), dh2 = Bh - tdiv2(Ah + Ch);
		o2[0] = Al + tdiv4((m ? pdl : dl2) + dl2);
		o2[1] = Ah + tdiv4((m ? pdh : dh2) + dh2);
		o2[2] = dl2;
		o2[3] = dh2;
		o2m = m;          // (the pair before the strip, m0 - 1, is only here for its details: never stored)
		pdl = dl2;
		pdh = dh2;
	};
	for (int jb = jfirst; jb <= r_hi; jb += S) {
#pragma unroll
		for (int k = 0; k < 2 * S; ++k)
			cur[k] = hold(nxt[k]);   // the one wait of the iteration (see k_fwd_level_w)
		if (jb > jfirst)
			store_batch(jb - S);
		if (jb + S <= r_hi) {
#pragma unroll
			for (int k = 0; k < 2 * S; ++k)
				nxt[k] = *reinterpret_cast<const int4 *>(rowp(2 * (jb + S) + 1 + k));
		}
#pragma unroll
		for (int s = 0; s < S; ++s) {
			const int jj = jb + s;
			if (jj > r_hi)
				break;
			I2 l1, h1, l2 = l0, h2v = h0;
			fwd_lift_q(cur[2 * s], q, a.nquads, l1, h1);
			if (2 * jj + 2 < a.h)
				fwd_lift_q(cur[2 * s + 1], q, a.nquads, l2, h2v);
			const I2 dl = i2_pred(l1, l0, l2);
			const I2 dh = i2_pred(h1, h0, h2v);
			const I2 sl = i2_upd(l0, jj ? pl : dl, dl);
			osh[s] = i2_upd(h0, jj ? ph : dh, dh);
			odl[s] = dl;
			odh[s] = dh;
			pl = dl;
			ph = dh;
			l0 = l2;
			h0 = h2v;
			if (jj >= r_lo) {   // sl is LL row jj of level k: the lane's pair of level k+1
				int lo2, hi2;
				fwd_lift_pair(sl.a, sl.b, q, a.nquads, lo2, hi2);
				if (jj & 1) {
					Bl = lo2;
					Bh = hi2;
				} else {
					if (jj > r_lo)
						finish_pair((jj >> 1) - 1, lo2, hi2);
					Al = lo2;
					Ah = hi2;
				}
			}
		}
	}
	const int jlast = jfirst + (r_hi - jfirst) / S * S;
	if (m1 == h4) {
		// the plane's last row pair of level k+1 has no even row below it: x[N] := x[N-2] (its B row came with the last batch)
		store_batch(jlast);
		finish_pair(h4 - 1, Al, Ah);
		store_batch(jlast + S);   // (beyond the strip's rows of level k: only the finished pair goes out)
	} else {
		store_batch(jlast);
	}
}

// ---- the finest level from 8-bit pixels, in packed 16-bit arithmetic ----
// Samples of magnitude <= 255 cannot leave 16 bits anywhere in one level of cdf53.h:9-34 (|d| <= 510 after the row
// pass, <= 1020 after the column pass; every intermediate sum stays below 2^12): the lane's two column pairs ride in
// the halves of one register and every add / shift / subtract is a v_pk_* instruction on both.  Same results as
// the int32 arithmetic of k_fwd_level_w, with about half the vector instructions.
typedef short P2 __attribute__((ext_vector_type(2)));
typedef unsigned short U2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ P2 p2_of(unsigned u) { return __builtin_bit_cast(P2, u); }
__device__ __forceinline__ unsigned bits_of(P2 v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ P2 tdiv2p(P2 a) { return (a + (P2)((U2)a >> (U2)15)) >> (P2)1; }
__device__ __forceinline__ P2 tdiv4p(P2 a) { return (a + ((a >> (P2)15) & (P2)3)) >> (P2)2; }

// image.h:52-65 on two pixels at once: channel ch of (R, G, B) pairs
__device__ __forceinline__ P2 ycocg_p(P2 r, P2 g, P2 b, int ch)
{
	const P2 co = r - b;
	const P2 t = b + tdiv2p(co);
	const P2 cg = g - t;
	return ch == 0 ? t + tdiv2p(cg) : ch == 1 ? co : cg;
}

// one row's samples of this lane as packed pairs: E = (x[4q], x[4q+2]), O = (x[4q+1], x[4q+3]); xr = x[4q+4] in its low
// half (lane 63 only), left = (x[4q-2], x[4q-1]) (lane 0 only)
struct RowP {
	P2 E, O, xr, left;
};

__device__ __forceinline__ RowP row_p(const FwdRaw8 &r, int)
{
	RowP o;
	o.E = p2_of(r.v & 0x00ff00ffu);
	o.O = p2_of((r.v >> 8) & 0x00ff00ffu);
	o.xr = p2_of(r.edge & 255u);
	o.left = p2_of(__builtin_amdgcn_perm(0u, r.edge, 0x0c030c02u));   // the last two bytes of the word before
	return o;
}

__device__ __forceinline__ RowP row_p(const FwdRawRgb &r, int ch)
{
	// a = R0 G0 B0 R1, b = G1 B1 R2 G2, c = B2 R3 G3 B3 (v_perm_b32: selector bytes 0-3 pick from the second operand, 4-7 from the first, 0x0c is zero)
	RowP o;
	o.E = ycocg_p(p2_of(__builtin_amdgcn_perm(r.b, r.a, 0x0c060c00u)), p2_of(__builtin_amdgcn_perm(r.b, r.a, 0x0c070c01u)),
		p2_of(__builtin_amdgcn_perm(r.c, r.a, 0x0c040c02u)), ch);
	o.O = ycocg_p(p2_of(__builtin_amdgcn_perm(r.c, r.a, 0x0c050c03u)), p2_of(__builtin_amdgcn_perm(r.c, r.b, 0x0c060c00u)),
		p2_of(__builtin_amdgcn_perm(r.c, r.b, 0x0c070c01u)), ch);
	// lane 63: e0 = R4 G4 B4 ..; lane 0: e0, e1 = bytes 12q-8 .. 12q-1, pixels 4q-2 and 4q-1 are the last six
	o.xr = ycocg_p(p2_of(r.e0 & 255u), p2_of((r.e0 >> 8) & 255u), p2_of((r.e0 >> 16) & 255u), ch);
	o.left = ycocg_p(p2_of(__builtin_amdgcn_perm(r.e1, r.e0, 0x0c050c02u)), p2_of(__builtin_amdgcn_perm(r.e1, r.e0, 0x0c060c03u)),
		p2_of(__builtin_amdgcn_perm(r.e1, r.e0, 0x0c070c04u)), ch);
	return o;
}

// cdf53.h:9-34 along the row for the lane's two pairs: L = (s[2q], s[2q+1]), Hh = (d[2q], d[2q+1])
__device__ __forceinline__ void fwd_lift_p(const RowP &r, int q, int lane, int nquads, P2 &L, P2 &Hh)
{
	const unsigned e = bits_of(r.E);
	unsigned nx = (unsigned)__shfl_down((int)e, 1);          // the next lane's (x[4q+4], ..)
	if (lane == 63)
		nx = bits_of(r.xr);
	if (q + 1 >= nquads)
		nx = e >> 16;                                        // x[w] := x[w-2]
	const P2 En = p2_of(__builtin_amdgcn_alignbit(nx, e, 16));   // (x[4q+2], x[4q+4])
	const P2 D = r.O - tdiv2p(r.E + En);
	const unsigned d = bits_of(D);
	unsigned pv = (unsigned)__shfl_up((int)d, 1);            // high half: the previous lane's d[2q-1]
	if (lane == 0) {
		const P2 dm = r.left.yy - tdiv2p(r.left.xx + r.E.xx);   // d[2q-1] from the two samples before the strip
		pv = bits_of(dm) << 16;
	}
	if (q == 0)
		pv = d << 16;                                        // d[-1] := d[0]
	const P2 Dl = p2_of(__builtin_amdgcn_alignbit(d, pv, 16));   // (d[2q-1], d[2q])
	L = r.E + tdiv4p(Dl + D);
	Hh = D;
}

__device__ __forceinline__ void hist_add2(HistAcc &h, P2 v)
{
	const unsigned a = bits_of(__builtin_elementwise_max(v, -v));   // both magnitudes (below 2^12)
	h.mx |= a;
	const unsigned t0 = 32u - (unsigned)__clz((int)(a & 0xffffu)), t1 = 32u - (unsigned)__clz((int)(a >> 16));
	h.R += 0x1111111111111111ull << (4u * t0);
	h.R += 0x1111111111111111ull << (4u * t1);
}

__device__ __forceinline__ void st2(int *p, P2 v) { *reinterpret_cast<int2 *>(p) = make_int2((int)v.x, (int)v.y); }
__device__ __forceinline__ void st2(short *p, P2 v) { *reinterpret_cast<unsigned *>(p) = bits_of(v); }

// Rows for the batched loop below: every lane loads — lanes beyond the row from the row's last quad, the edge word from
// a clamped place — so that no load sits behind a branch or feeds a select (either would make the wave wait for it at
// once); what the extra lanes get is never used.  e8: byte offset of the lane's edge word in a row (the word before the
// lane's pixels for lane 0, the word after them for the others), the same for every row.
struct LaneAt {
	int main, edge;   // byte offsets in a source row
};

__device__ __forceinline__ LaneAt lane_at(SrcTag<uint8_t>, int q, int lane, int nquads)
{
	const int qa = min(q, nquads - 1);
	LaneAt o = { 4 * qa, lane == 0 ? max(4 * qa - 4, 0) : min(4 * qa + 4, 4 * (nquads - 1)) };
	return o;
}

__device__ __forceinline__ LaneAt lane_at(SrcTag<Rgb8>, int q, int lane, int nquads)
{
	const int qa = min(q, nquads - 1);
	LaneAt o = { 12 * qa, lane == 0 ? max(12 * qa - 8, 0) : min(12 * qa + 12, 12 * nquads - 8) };
	return o;
}

__device__ __forceinline__ FwdRaw8 fwd_load_p(SrcTag<uint8_t>, const uint8_t *__restrict__ row, const LaneAt &at)
{
	FwdRaw8 r;
	r.v = *reinterpret_cast<const unsigned *>(row + at.main);
	r.edge = *reinterpret_cast<const unsigned *>(row + at.edge);
	return r;
}

// (the RGB row as the two loads deliver it — three words and two words in consecutive registers; taking single words
// out of them at load time would be a use of the load)
typedef unsigned U32x3 __attribute__((ext_vector_type(3)));
typedef unsigned U32x2 __attribute__((ext_vector_type(2)));
struct FwdRawRgbP {
	U32x3 abc;   // pixels 4q .. 4q+3
	U32x2 e;     // lane 63: e.x = the word with pixel 4q+4; lane 0: bytes 12q-8 .. 12q-1
};

__device__ __forceinline__ FwdRawRgbP fwd_load_p(SrcTag<Rgb8>, const uint8_t *__restrict__ row, const LaneAt &at)
{
	FwdRawRgbP r;
	r.abc = *reinterpret_cast<const U32x3 *>(row + at.main);
	r.e = *reinterpret_cast<const U32x2 *>(row + at.edge);
	return r;
}

__device__ __forceinline__ FwdRaw8 hold(const FwdRaw8 &r)
{
	FwdRaw8 o = { hold(r.v), hold(r.edge) };
	return o;
}
__device__ __forceinline__ FwdRawRgb hold(const FwdRawRgbP &r)
{
	FwdRawRgb o = { hold(r.abc.x), hold(r.abc.y), hold(r.abc.z), hold(r.e.x), hold(r.e.y) };
	return o;
}
__device__ __forceinline__ FwdRawRgb as_used(const FwdRawRgbP &r)
{
	FwdRawRgb o = { r.abc.x, r.abc.y, r.abc.z, r.e.x, r.e.y };
	return o;
}
__device__ __forceinline__ FwdRaw8 as_used(const FwdRaw8 &r) { return r; }

// the row as it is loaded / as it is used
template <typename SrcT>
struct RowRegs {
	typedef FwdRaw8 Loaded;
	typedef FwdRaw8 Used;
};
template <>
struct RowRegs<Rgb8> {
	typedef FwdRawRgbP Loaded;
	typedef FwdRawRgb Used;
};

// CH: the YCoCg-R channel the launch's workgroup extracts, as a compile-time constant (RGB; the kernel below branches —
// uniformly — into the three instances: Co is one subtraction per pixel pair, and neither it nor Cg needs what only Y needs;
// with the channel as a run-time value every workgroup computed all three and selected: 1.84 -> 1.6 ms per 256 frames of 1080p)
template <typename SrcT, bool HIST, int CH>
__device__ __forceinline__ void fwd_pixels_body(const LevelArgsW &A, int bx, int by)
{
	const LevelArgs &a = A.a;
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int chan = CH;
	const int sx = (bx << A.wx_log2) + (wv & ((1 << A.wx_log2) - 1));
	const int q = sx * 64 + lane;
	const int j0 = (by * (WAVES >> A.wx_log2) + (wv >> A.wx_log2)) * a.rpw;
	if (j0 >= a.h2 || sx * 64 >= A.nquads)
		return;
	const int j1 = min(j0 + a.rpw, a.h2);
	const int plane = IsRgb<SrcT>::value ? (int)blockIdx.z * 3 + chan : (int)blockIdx.z;
	const bool valid = q < A.nquads;
	int ch_rt;
	const uint8_t *src = fwd_base(SrcTag<SrcT>(), a, plane, ch_rt);
	const int ch = CH;   // (== ch_rt)
	int *ll = a.ll + plane * a.ll_ps;
	short *ll16 = a.ll16 ? a.ll16 + plane * a.ll_ps : nullptr;      // (uniform)
	int *det = a.det + plane * a.det_ps;
	short *det16 = a.det16 ? a.det16 + plane * a.det_ps : nullptr;   // (uniform)

	// Memory operations retire in order on this part (one counter for loads and stores): a wave that waits for rows it
	// loaded also waits for everything it issued before them, and a wait the compiler cannot count exactly waits for
	// everything.  So the loop works in batches of S row pairs: wait once, send the previous batch's results out, ask
	// for the next batch's rows, then compute S row pairs without touching memory — by the next wait both the stores
	// and the loads are a whole batch of arithmetic old.
	constexpr int S = 2;
	const int jfirst = j0 > 0 ? j0 - 1 : 0;
	const P2 zero = p2_of(0u);
	P2 l0, h0, pl = zero, ph = zero;
	const LaneAt at = lane_at(SrcTag<SrcT>(), q, lane, A.nquads);
	{
		const typename RowRegs<SrcT>::Loaded r0 = fwd_load_p(SrcTag<SrcT>(), src + (long)(2 * jfirst) * a.spitch, at);
		fwd_lift_p(row_p(as_used(r0), ch), q, lane, A.nquads, l0, h0);
	}
	auto rowp = [&](int r) { return src + (long)min(r, a.h - 1) * a.spitch; };
	typename RowRegs<SrcT>::Used cur[2 * S];
	typename RowRegs<SrcT>::Loaded nxt[2 * S];
#pragma unroll
	for (int k = 0; k < 2 * S; ++k)
		nxt[k] = fwd_load_p(SrcTag<SrcT>(), rowp(2 * jfirst + 1 + k), at);
	P2 osl[S], osh[S], odl[S], odh[S];   // a batch's results wait here for the next iteration's stores
	auto store_batch = [&](int jb) {
#pragma unroll
		for (int s = 0; s < S; ++s) {
			const int j = jb + s;
			if (j >= j0 && j < j1 && valid) {
				const bool odd_in = 2 * j + 1 < a.h;
				if (ll16)
					st2(ll16 + (long)j * a.llpitch + 2 * q, osl[s]);
				else
					st2(ll + (long)j * a.llpitch + 2 * q, osl[s]);
				if (det16) {
					st2(det16 + (long)j * a.dpitch + a.w2 + 2 * q, osh[s]);
					if (odd_in) {
						st2(det16 + (long)(a.h2 + j) * a.dpitch + 2 * q, odl[s]);
						st2(det16 + (long)(a.h2 + j) * a.dpitch + a.w2 + 2 * q, odh[s]);
					}
				} else {
					st2(det + (long)j * a.dpitch + a.w2 + 2 * q, osh[s]);
					if (odd_in) {
						st2(det + (long)(a.h2 + j) * a.dpitch + 2 * q, odl[s]);
						st2(det + (long)(a.h2 + j) * a.dpitch + a.w2 + 2 * q, odh[s]);
					}
				}
			}
		}
	};
	HistAcc hHL = { 0, 0, 0, 0 }, hLH = { 0, 0, 0, 0 }, hHH = { 0, 0, 0, 0 };
	for (int jb = jfirst; jb < j1; jb += S) {
#pragma unroll
		for (int k = 0; k < 2 * S; ++k)
			cur[k] = hold(nxt[k]);   // the one wait of the iteration: everything outstanding is a batch old
		if (jb > jfirst)
			store_batch(jb - S);
		if (jb + S < j1) {
#pragma unroll
			for (int k = 0; k < 2 * S; ++k)
				nxt[k] = fwd_load_p(SrcTag<SrcT>(), rowp(2 * (jb + S) + 1 + k), at);
		}
#pragma unroll
		for (int s = 0; s < S; ++s) {
			const int jj = jb + s;
			if (jj >= j1)
				break;
			const int r1 = 2 * jj + 1, r2 = r1 + 1;
			const bool odd_in = r1 < a.h;
			P2 l1 = zero, h1 = zero, l2 = l0, h2v = h0;
			if (odd_in)
				fwd_lift_p(row_p(cur[2 * s], ch), q, lane, A.nquads, l1, h1);
			if (r2 < a.h)
				fwd_lift_p(row_p(cur[2 * s + 1], ch), q, lane, A.nquads, l2, h2v);
			const P2 dl = l1 - tdiv2p(l0 + l2);     // cdf53.h:13 down the columns
			const P2 dh = h1 - tdiv2p(h0 + h2v);
			P2 sl = l0, sh = h0;
			if (odd_in) {
				sl = l0 + tdiv4p((jj ? pl : dl) + dl);   // cdf53.h:20
				sh = h0 + tdiv4p((jj ? ph : dh) + dh);
			}
			osl[s] = sl;
			osh[s] = sh;
			odl[s] = dl;
			odh[s] = dh;
			if (HIST && jj >= j0) {
				// the detail coefficients of this row pair (cdf53.h:9-34 output): HL row jj, LH and HH row h2 + jj
				if (valid) {
					hist_add2(hHL, sh);
					if (odd_in) {
						hist_add2(hLH, dl);
						hist_add2(hHH, dh);
					}
				}
				if ((jj & 3) == 3) {   // eight coefficients per subband since the last fold: a nibble holds fifteen
					hist_fold(hHL);
					hist_fold(hLH);
					hist_fold(hHH);
				}
				// a block ends where its 32 rows end (or the strip does): the rows of HL are jj, those of LH / HH h2 + jj
				const bool last = jj == j1 - 1;
				const int bxl = (2 * q) >> 5, bxh = (a.w2 + 2 * q) >> 5;
				if (last || ((jj + 1) & 31) == 0) {
					hHL.mx = (hHL.mx | (hHL.mx >> 16)) & 0xffffu;
					hist_flush(hHL, A.hist, plane, bxh, jj >> 5, lane);
				}
				if (last || ((a.h2 + jj + 1) & 31) == 0) {
					hLH.mx = (hLH.mx | (hLH.mx >> 16)) & 0xffffu;
					hHH.mx = (hHH.mx | (hHH.mx >> 16)) & 0xffffu;
					hist_flush(hLH, A.hist, plane, bxl, (a.h2 + jj) >> 5, lane);
					hist_flush(hHH, A.hist, plane, bxh, (a.h2 + jj) >> 5, lane);
				}
			}
			pl = dl;
			ph = dh;
			l0 = l2;
			h0 = h2v;
		}
	}
	// the last batch (a strip has at least one row pair)
	store_batch(jfirst + (j1 - 1 - jfirst) / S * S);
}

template <typename SrcT, bool HIST>
__global__ __launch_bounds__(64 * WAVES) void k_fwd_pixels_w(LevelArgsW A)
{
	int bx, by, chan = 0;
	if (IsRgb<SrcT>::value) {
		xcd_strip_rgb(bx, by, chan);
		if (chan == 0)
			fwd_pixels_body<SrcT, HIST, 0>(A, bx, by);
		else if (chan == 1)
			fwd_pixels_body<SrcT, HIST, 1>(A, bx, by);
		else
			fwd_pixels_body<SrcT, HIST, 2>(A, bx, by);
	} else {
		xcd_strip(bx, by);
		fwd_pixels_body<SrcT, HIST, 0>(A, bx, by);
	}
}

// ---------------------------------------------------------------- inverse ---
// LL | HL | LH | HH samples of one row pair as they are loaded (F16: the detail bands are words of two 16-bit values,
// widened when used).  Every lane loads — lanes outside the row from a clamped quad, row pairs beyond the band from
// the last one — so that no load sits behind a branch or feeds a select; what those get is never used.
template <bool F16>
struct InvRawT {
	int2 sl, sh, dl, dh;
};
template <>
struct InvRawT<true> {
	int2 sl;
	unsigned sh, dl, dh;
};

__device__ __forceinline__ int2 ld2(const int *p) { return *reinterpret_cast<const int2 *>(p); }

// where a lane reads: its (clamped) column pair, and the clamps for the row pair index
struct InvAt {
	int col;          // 2 * clamped quad
	int jmax, jdmax;  // last row pair of the LL / HL bands, last of the LH / HH bands (an odd height has one row less there)
	int colr, coll;   // the column after the lane's pair and the one before it (clamped to the band)
};

__device__ __forceinline__ InvAt inv_at(const LevelArgs &a, int qd, int nquads)
{
	const int col = 2 * min(max(qd, 0), nquads - 1);
	InvAt o = { col, a.h2 - 1, a.h / 2 - 1, min(col + 2, 2 * nquads - 1), max(col - 1, 0) };
	return o;
}

// The columns next to a lane's pair, as loaded: the inverse undoes the columns first, so the row step of a wave's
// first and last lane needs its neighbours' samples AFTER their column step — every lane carries the column to the
// right of its pair (both bands) and the high band's column to its left through the column step as well; only lanes
// 0 and 63 use them (the others get the same values from their neighbours' registers), and what all 64 lanes load
// for them are lines their neighbours load anyway.  (Before: waves that overlapped by eight lanes of 64, i.e. 14 %
// more rows loaded than written.)
struct InvHalo {
	int sl_r, sh_r, dl_r, dh_r;   // LL | HL | LH | HH at column 2q+2
	int sh_l, dh_l;               // HL | HH at column 2q-1
};

__device__ __forceinline__ InvHalo inv_load_halo(const LevelArgs &a, const int *llp, const int *det, int j, const InvAt &at)
{
	InvHalo r;
	const int ja = min(j, at.jmax), jd = min(j, at.jdmax);
	r.sl_r = llp[(long)ja * a.spitch + at.colr];
	r.sh_r = det[(long)ja * a.dpitch + a.w2 + at.colr];
	r.dl_r = det[(long)(a.h2 + jd) * a.dpitch + at.colr];
	r.dh_r = det[(long)(a.h2 + jd) * a.dpitch + a.w2 + at.colr];
	r.sh_l = det[(long)ja * a.dpitch + a.w2 + at.coll];
	r.dh_l = det[(long)(a.h2 + jd) * a.dpitch + a.w2 + at.coll];
	return r;
}

__device__ __forceinline__ InvHalo inv_load_halo(const LevelArgs &a, const int *llp, const short *det16, int j, const InvAt &at)
{
	InvHalo r;   // (16-bit bands: the load itself widens)
	const int ja = min(j, at.jmax), jd = min(j, at.jdmax);
	r.sl_r = llp[(long)ja * a.spitch + at.colr];
	r.sh_r = det16[(long)ja * a.dpitch + a.w2 + at.colr];
	r.dl_r = det16[(long)(a.h2 + jd) * a.dpitch + at.colr];
	r.dh_r = det16[(long)(a.h2 + jd) * a.dpitch + a.w2 + at.colr];
	r.sh_l = det16[(long)ja * a.dpitch + a.w2 + at.coll];
	r.dh_l = det16[(long)(a.h2 + jd) * a.dpitch + a.w2 + at.coll];
	return r;
}

// the halo columns as one more pair per band: .a = the column to the right, .b = the column to the left (high band only)
struct HaloPairs {
	I2 sl, sh, dl, dh;
};

__device__ __forceinline__ InvRawT<false> inv_load_w(const LevelArgs &a, const int *llp, const int *det, int j, const InvAt &at)
{
	InvRawT<false> r;
	const int ja = min(j, at.jmax), jd = min(j, at.jdmax);
	r.sl = ld2(llp + (long)ja * a.spitch + at.col);
	r.sh = ld2(det + (long)ja * a.dpitch + a.w2 + at.col);
	r.dl = ld2(det + (long)(a.h2 + jd) * a.dpitch + at.col);
	r.dh = ld2(det + (long)(a.h2 + jd) * a.dpitch + a.w2 + at.col);
	return r;
}

__device__ __forceinline__ InvRawT<true> inv_load_w(const LevelArgs &a, const int *llp, const short *det16, int j, const InvAt &at)
{
	InvRawT<true> r;
	const int ja = min(j, at.jmax), jd = min(j, at.jdmax);
	r.sl = ld2(llp + (long)ja * a.spitch + at.col);
	r.sh = *reinterpret_cast<const unsigned *>(det16 + (long)ja * a.dpitch + a.w2 + at.col);
	r.dl = *reinterpret_cast<const unsigned *>(det16 + (long)(a.h2 + jd) * a.dpitch + at.col);
	r.dh = *reinterpret_cast<const unsigned *>(det16 + (long)(a.h2 + jd) * a.dpitch + a.w2 + at.col);
	return r;
}

__device__ __forceinline__ int2 hold(int2 v) { return make_int2(hold(v.x), hold(v.y)); }
__device__ __forceinline__ InvRawT<false> hold(const InvRawT<false> &r)
{
	InvRawT<false> o = { hold(r.sl), hold(r.sh), hold(r.dl), hold(r.dh) };
	return o;
}
__device__ __forceinline__ InvRawT<true> hold(const InvRawT<true> &r)
{
	InvRawT<true> o = { hold(r.sl), hold(r.sh), hold(r.dl), hold(r.dh) };
	return o;
}

__device__ __forceinline__ I2 to_i2(int2 v)
{
	I2 r = { v.x, v.y };
	return r;
}
__device__ __forceinline__ I2 to_i2(I2 v) { return v; }

__device__ __forceinline__ HaloPairs hold(const InvHalo &r)
{
	HaloPairs o;
	o.sl.a = hold(r.sl_r);
	o.sl.b = 0;
	o.sh.a = hold(r.sh_r);
	o.sh.b = hold(r.sh_l);
	o.dl.a = hold(r.dl_r);
	o.dl.b = 0;
	o.dh.a = hold(r.dh_r);
	o.dh.b = hold(r.dh_l);
	return o;
}
__device__ __forceinline__ HaloPairs as_pairs(const InvHalo &r)
{
	HaloPairs o = { { r.sl_r, 0 }, { r.sh_r, r.sh_l }, { r.dl_r, 0 }, { r.dh_r, r.dh_l } };
	return o;
}

__device__ __forceinline__ I2 to_i2(unsigned u)   // two 16-bit values
{
	I2 r = { (int)(short)(u & 0xffffu), (int)u >> 16 };
	return r;
}

// the plane's detail bands as the kernel variant reads them
template <bool F16>
struct DetPtr {
	typedef const int *type;
	static __device__ __forceinline__ type of(const LevelArgs &a, long plane) { return a.det + plane * a.det_ps; }
};
template <>
struct DetPtr<true> {
	typedef const short *type;
	static __device__ __forceinline__ type of(const LevelArgs &a, long plane) { return a.det16 + plane * a.det_ps; }
};

// horizontal inverse of one output row for this lane's two pairs: x[4qd .. 4qd+3]
struct Quad4 {
	int v[4];
};

__device__ __forceinline__ Quad4 inv_row_vals(int qd, int nquads, I2 lo, I2 hi)
{
	int hl = __shfl_up(hi.b, 1);
	if (qd <= 0)
		hl = hi.a;
	const int e0 = lo.a - tdiv4(hl + hi.a);
	const int e1 = lo.b - tdiv4(hi.a + hi.b);
	int er = __shfl_down(e0, 1);
	if (qd + 1 >= nquads)
		er = e1;
	Quad4 r;
	r.v[0] = e0;
	r.v[1] = hi.a + tdiv2(e0 + e1);
	r.v[2] = e1;
	r.v[3] = hi.b + tdiv2(e1 + er);
	return r;
}

// a finished output row in the registers it waits in for its store: four int32 samples, or four clamped 8-bit pixels (pnm.h:108)
template <typename DstT>
struct OutRow {
	typedef int4 type;
	static __device__ __forceinline__ type of(const Quad4 &v) { return make_int4(v.v[0], v.v[1], v.v[2], v.v[3]); }
	static __device__ __forceinline__ void store(int *__restrict__ row, int qd, const type &v) { *reinterpret_cast<int4 *>(row + 4 * qd) = v; }
};
template <>
struct OutRow<uint8_t> {
	typedef unsigned type;
	static __device__ __forceinline__ type of(const Quad4 &v)
	{
		auto c8 = [](int x) { return (unsigned)(x < 0 ? 0 : x > 255 ? 255 : x); };
		return c8(v.v[0]) | (c8(v.v[1]) << 8) | (c8(v.v[2]) << 16) | (c8(v.v[3]) << 24);
	}
	static __device__ __forceinline__ void store(uint8_t *__restrict__ row, int qd, const type &v) { *reinterpret_cast<unsigned *>(row + 4 * qd) = v; }
};

template <typename DstT>
__device__ __forceinline__ DstT *inv_dst(const LevelArgs &a);
template <>
__device__ __forceinline__ int *inv_dst<int>(const LevelArgs &a) { return a.ll; }
template <>
__device__ __forceinline__ uint8_t *inv_dst<uint8_t>(const LevelArgs &a) { return a.dst8; }

// The vertical state of one plane between row pairs: detail rows dl / dh and even rows el / eh of the current pair.
struct InvCols {
	I2 dl, dh, el, eh;
};

// cdf53.h:40-47 down the columns: the even row of pair j from its subband samples and the details around it
__device__ __forceinline__ I2 inv_even(const LevelArgs &a, int j, I2 s, I2 dprev, I2 dcur)
{
	if ((a.h & 1) && 2 * j == a.h - 1)
		return s;
	const I2 dp = j ? dprev : dcur;
	I2 r = { s.a - tdiv4(dp.a + dcur.a), s.b - tdiv4(dp.b + dcur.b) };
	return r;
}

// the strip's first row pair: what the loop carries
template <class Raw>
__device__ __forceinline__ InvCols inv_first(const LevelArgs &a, int j0, const Raw &before, const Raw &first)
{
	I2 pdl = { 0, 0 }, pdh = { 0, 0 };
	if (j0 > 0) {
		pdl = to_i2(before.dl);
		pdh = to_i2(before.dh);
	}
	InvCols c;
	c.dl = to_i2(first.dl);
	c.dh = to_i2(first.dh);
	c.el = inv_even(a, j0, to_i2(first.sl), pdl, c.dl);
	c.eh = inv_even(a, j0, to_i2(first.sh), pdh, c.dh);
	return c;
}

// one row pair, the column step: the (low, high) samples of its even and its odd row from the state and the next
// pair's samples `n`
struct RowLH {
	I2 lo, hi;
};

template <class Raw>
__device__ __forceinline__ void inv_cols(const LevelArgs &a, int jj, InvCols &c, const Raw &n, RowLH &even, RowLH &odd)
{
	const int r1 = 2 * jj + 1;
	I2 ndl = { 0, 0 }, ndh = { 0, 0 }, nel = c.el, neh = c.eh;   // mirror x[h] := x[h-2]
	if (r1 + 1 < a.h) {
		if (r1 + 2 < a.h) {
			ndl = to_i2(n.dl);
			ndh = to_i2(n.dh);
		}
		nel = inv_even(a, jj + 1, to_i2(n.sl), c.dl, ndl);
		neh = inv_even(a, jj + 1, to_i2(n.sh), c.dh, ndh);
	}
	even.lo = c.el;
	even.hi = c.eh;
	odd.lo.a = c.dl.a + tdiv2(c.el.a + nel.a);   // cdf53.h:49-56
	odd.lo.b = c.dl.b + tdiv2(c.el.b + nel.b);
	odd.hi.a = c.dh.a + tdiv2(c.eh.a + neh.a);
	odd.hi.b = c.dh.b + tdiv2(c.eh.b + neh.b);
	c.dl = ndl;
	c.dh = ndh;
	c.el = nel;
	c.eh = neh;
}

// (waves that overlap by a lane on each side: the RGB kernel)
template <class Raw>
__device__ __forceinline__ void inv_pair(const LevelArgs &a, int jj, int qd, int nquads, InvCols &c, const Raw &n, Quad4 &even, Quad4 &odd)
{
	RowLH e, o;
	inv_cols(a, jj, c, n, e, o);
	even = inv_row_vals(qd, nquads, e.lo, e.hi);
	odd = inv_row_vals(qd, nquads, o.lo, o.hi);
}

// the row step with the halo columns' samples x (x.lo.a, x.hi.a: the column to the right; x.hi.b: the one to the left)
__device__ __forceinline__ Quad4 inv_row_vals_h(int qd, int nquads, int lane, const RowLH &r, const RowLH &x)
{
	int hl = __shfl_up(r.hi.b, 1);
	if (lane == 0)
		hl = x.hi.b;
	if (qd <= 0)
		hl = r.hi.a;
	const int e0 = r.lo.a - tdiv4(hl + r.hi.a);
	const int e1 = r.lo.b - tdiv4(r.hi.a + r.hi.b);
	int er = __shfl_down(e0, 1);
	if (lane == 63)
		er = x.lo.a - tdiv4(r.hi.b + x.hi.a);
	if (qd + 1 >= nquads)
		er = e1;
	Quad4 q;
	q.v[0] = e0;
	q.v[1] = r.hi.a + tdiv2(e0 + e1);
	q.v[2] = e1;
	q.v[3] = r.hi.b + tdiv2(e1 + er);
	return q;
}

// Inverse level: int32 planes, or — the finest level of a gray image — clamped 8-bit pixels out (and, F16, the detail
// bands in as 16-bit values).  The loop is batched like the forward kernel's (see there): one wait per S row pairs.
template <typename DstT, bool F16>
__global__ __launch_bounds__(64 * WAVES) void k_inv_level_w(LevelArgsW A)
{
	const LevelArgs &a = A.a;
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	int bx, by;
	xcd_strip(bx, by);
	const int qd = bx * 64 + lane;   // all 64 lanes write: the halo columns ride along (InvHalo)
	const int j0 = (by * WAVES + wv) * a.rpw;
	if (j0 >= a.h2)
		return;
	const int j1 = min(j0 + a.rpw, a.h2);
	const int plane = blockIdx.z;
	const bool writes = qd < A.nquads;
	const int *llp = a.src + plane * a.src_ps;
	const typename DetPtr<F16>::type det = DetPtr<F16>::of(a, plane);
	typedef InvRawT<F16> InvRaw;
	typedef OutRow<DstT> Out;
	DstT *dst = inv_dst<DstT>(a) + plane * a.ll_ps;
	const InvAt at = inv_at(a, qd, A.nquads);

	constexpr int S = 2;
	const int jb4 = j0 > 0 ? j0 - 1 : 0;
	InvCols c = inv_first(a, j0, inv_load_w(a, llp, det, jb4, at), inv_load_w(a, llp, det, j0, at));
	InvCols cx = inv_first(a, j0, as_pairs(inv_load_halo(a, llp, det, jb4, at)), as_pairs(inv_load_halo(a, llp, det, j0, at)));
	InvRaw nxt[S], cur[S];
	InvHalo nxtx[S];
	HaloPairs curx[S];
#pragma unroll
	for (int s = 0; s < S; ++s) {
		nxt[s] = inv_load_w(a, llp, det, j0 + 1 + s, at);
		nxtx[s] = inv_load_halo(a, llp, det, j0 + 1 + s, at);
	}
	typename Out::type orow[2 * S];   // a batch's rows wait here for the next iteration's stores
	auto store_batch = [&](int jb) {
#pragma unroll
		for (int s = 0; s < S; ++s) {
			const int j = jb + s;
			if (j < j1 && writes) {
				Out::store(dst + (long)(2 * j) * a.llpitch, qd, orow[2 * s]);
				if (2 * j + 1 < a.h)
					Out::store(dst + (long)(2 * j + 1) * a.llpitch, qd, orow[2 * s + 1]);
			}
		}
	};
	for (int jb = j0; jb < j1; jb += S) {
#pragma unroll
		for (int s = 0; s < S; ++s) {
			cur[s] = hold(nxt[s]);   // the one wait of the iteration
			curx[s] = hold(nxtx[s]);
		}
		if (jb > j0)
			store_batch(jb - S);
		if (jb + S < j1) {
#pragma unroll
			for (int s = 0; s < S; ++s) {
				nxt[s] = inv_load_w(a, llp, det, jb + S + 1 + s, at);
				nxtx[s] = inv_load_halo(a, llp, det, jb + S + 1 + s, at);
			}
		}
#pragma unroll
		for (int s = 0; s < S; ++s) {
			const int jj = jb + s;
			if (jj >= j1)
				break;
			RowLH e, o, ex, ox;
			inv_cols(a, jj, c, cur[s], e, o);
			inv_cols(a, jj, cx, curx[s], ex, ox);
			orow[2 * s] = Out::of(inv_row_vals_h(qd, A.nquads, lane, e, ex));
			orow[2 * s + 1] = Out::of(inv_row_vals_h(qd, A.nquads, lane, o, ox));
		}
	}
	store_batch(j0 + (j1 - 1 - j0) / S * S);
}

// ---- two levels per pass (inverse, int32 planes): the mirror image of k_fwd2_level_w ----
// A wave undoes level k+1 for its column pairs — one pair per lane: the lane's LL sample and its three details — and
// hands every rebuilt LL row of level k (the lane's two samples of it) straight to the level-k stage, whose other
// inputs are that level's detail bands: the LL band of level k is neither written nor read.  Rows of the output leave
// as whole 128-byte lines: a wave owns the quads of lanes 4..59 (56 x 16 bytes = seven lines, at a multiple of seven
// lines), lanes 3 and 60 load their columns only for their neighbours' row steps (undoing the columns first leaves
// every lane's high-band samples right whatever its neighbours hold; cdf53.h:36-61), the other lanes repeat those
// lanes' loads.  Shapes as in k_fwd2_level_w: w % 4 == 0, h % 4 == 0.
struct Inv2Args {
	const int *ll2;  long ll2_ps;  int ll2pitch;   // LL of level k+1, w/4 x h/4
	const int *det;  long det_ps;  int dpitch;     // the pyramid: detail bands of both levels
	int *dst;        long dst_ps;  int opitch;     // output, w x h
	int w, h, nquads;
	int mpw;          // level k+1 row pairs per wave strip
	const short *det16;   // F16: the detail bands of BOTH levels as 16-bit values (positions, pitch and plane stride of det)
	uint8_t *dst8;        // 8-bit output (the finest level of a gray picture: pnm.h:108's clamp fused), dst_ps / opitch in bytes
};
constexpr int V2_FIRST = 4, V2_OWN = 56;

struct Raw2 {       // one row of level k+1 at the lane's pair: LL | HL | LH | HH
	int sl, sh, dl, dh;
};
// (one row pair of level k at the lane's two pairs — HL | LH | HH, its LL comes from level k+1 — is DetBand<F16>::raw1 below)
__device__ __forceinline__ Raw2 hold(const Raw2 &r)
{
	Raw2 o = { hold(r.sl), hold(r.sh), hold(r.dl), hold(r.dh) };
	return o;
}

// cdf53.h:36-61 along a row of level k+1 for the lane's pair (low, high) -> its two samples of level k's LL row
__device__ __forceinline__ I2 inv_row_pair(int q, int nquads, int lo, int hi)
{
	int hl = __shfl_up(hi, 1);
	if (q <= 0)
		hl = hi;
	const int e = lo - tdiv4(hl + hi);
	int er = __shfl_down(e, 1);
	if (q + 1 >= nquads)
		er = e;
	I2 r = { e, hi + tdiv2(e + er) };
	return r;
}

__device__ __forceinline__ I2 i2_sub4(I2 s, I2 dp, I2 d)   // cdf53.h:40-47: the even row from its low-pass sample and the details around it
{
	I2 r = { s.a - tdiv4(dp.a + d.a), s.b - tdiv4(dp.b + d.b) };
	return r;
}
__device__ __forceinline__ I2 i2_add2(I2 d, I2 e0, I2 e1)   // cdf53.h:49-56: the odd row
{
	I2 r = { d.a + tdiv2(e0.a + e1.a), d.b + tdiv2(e0.b + e1.b) };
	return r;
}

// the detail bands a two-level inverse step reads, and the rows it writes, as the kernel variant sees them
template <bool F16>
struct DetBand {
	typedef const int *ptr;
	struct raw1 {
		int2 sh, dl, dh;
	};
	static __device__ __forceinline__ ptr of(const Inv2Args &a, long plane) { return a.det + plane * a.det_ps; }
	static __device__ __forceinline__ int2 pair(const int *p) { return ld2(p); }
};
template <>
struct DetBand<true> {
	typedef const short *ptr;
	struct raw1 {
		unsigned sh, dl, dh;   // two 16-bit values each
	};
	static __device__ __forceinline__ ptr of(const Inv2Args &a, long plane) { return a.det16 + plane * a.det_ps; }
	static __device__ __forceinline__ unsigned pair(const short *p) { return *reinterpret_cast<const unsigned *>(p); }
};
__device__ __forceinline__ DetBand<false>::raw1 hold(const DetBand<false>::raw1 &r)
{
	DetBand<false>::raw1 o = { hold(r.sh), hold(r.dl), hold(r.dh) };
	return o;
}
__device__ __forceinline__ DetBand<true>::raw1 hold(const DetBand<true>::raw1 &r)
{
	DetBand<true>::raw1 o = { hold(r.sh), hold(r.dl), hold(r.dh) };
	return o;
}
template <typename DstT>
struct OutPlane {
	typedef int4 row;
	static __device__ __forceinline__ int *of(const Inv2Args &a, long plane) { return a.dst + plane * a.dst_ps; }
	static __device__ __forceinline__ void store(int *p, const int4 &v)
	{
		typedef int v4i __attribute__((ext_vector_type(4)));
		const v4i x = { v.x, v.y, v.z, v.w };
		__builtin_nontemporal_store(x, reinterpret_cast<v4i *>(p));
	}
};
template <>
struct OutPlane<uint8_t> {
	typedef unsigned row;
	static __device__ __forceinline__ uint8_t *of(const Inv2Args &a, long plane) { return a.dst8 + plane * a.dst_ps; }
	static __device__ __forceinline__ void store(uint8_t *p, unsigned v) { __builtin_nontemporal_store(v, reinterpret_cast<unsigned *>(p)); }
};

// DstT = int: int32 rows out (nontemporal 16-byte stores; the block's four waves on top of each other).  DstT = uint8_t: the finest
// level of a gray picture — clamped pixels, 4 bytes per lane: a wave's 224 bytes are seven 32-byte pieces, and the block's four waves
// sit SIDE BY SIDE so that together they write seven whole lines.  F16: both levels' detail bands are 16-bit values (the codec's
// pipelines, dwtx_p16); the arithmetic is int32 either way.
template <typename DstT, bool F16>
__global__ __launch_bounds__(64 * WAVES) void k_inv2_level_w(Inv2Args a)
{
	constexpr bool SIDE = sizeof(DstT) == 1;
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	int bx, by;
	xcd_strip(bx, by);
	const int strip = SIDE ? bx * WAVES + wv : bx;
	if (strip * V2_OWN >= a.nquads)
		return;
	const int q = strip * V2_OWN - V2_FIRST + lane;
	const int h2 = a.h >> 1, h4 = a.h >> 2, w2 = a.w >> 1, w4 = a.w >> 2;
	const int m0 = (SIDE ? by : by * WAVES + wv) * a.mpw;
	if (m0 >= h4)
		return;
	const int m1 = min(m0 + a.mpw, h4);
	const int plane = blockIdx.z;
	const bool own = lane >= V2_FIRST && lane < V2_FIRST + V2_OWN && q < a.nquads;
	// every lane loads: lanes 3 and 60 for their neighbours, the ones beyond them repeat those two lanes' columns
	const int qc = min(max(q + max(V2_FIRST - 1 - lane, 0) - max(lane - (V2_FIRST + V2_OWN), 0), 0), a.nquads - 1);
	const int *ll2 = a.ll2 + plane * a.ll2_ps + qc;
	const typename DetBand<F16>::ptr det = DetBand<F16>::of(a, plane);
	DstT *dst = OutPlane<DstT>::of(a, plane);
	typedef typename DetBand<F16>::raw1 Raw1;
	auto load2 = [&](int m) {
		const int mm = min(max(m, 0), h4 - 1);
		Raw2 r = { ll2[(long)mm * a.ll2pitch], (int)det[(long)mm * a.dpitch + w4 + qc], (int)det[(long)(h4 + mm) * a.dpitch + qc],
			(int)det[(long)(h4 + mm) * a.dpitch + w4 + qc] };
		return r;
	};
	auto load1 = [&](int j) {
		const int jc = min(max(j, 0), h2 - 1);
		Raw1 r = { DetBand<F16>::pair(det + (long)jc * a.dpitch + w2 + 2 * qc), DetBand<F16>::pair(det + (long)(h2 + jc) * a.dpitch + 2 * qc),
			DetBand<F16>::pair(det + (long)(h2 + jc) * a.dpitch + w2 + 2 * qc) };
		return r;
	};
	// level k+1, column direction: the state of pair m0
	int d2l, d2h, e2l, e2h;
	{
		const Raw2 before = load2(m0 - 1), first = load2(m0);
		d2l = first.dl;
		d2h = first.dh;
		e2l = first.sl - tdiv4((m0 ? before.dl : d2l) + d2l);
		e2h = first.sh - tdiv4((m0 ? before.dh : d2h) + d2h);
	}
	// level k: the state of row pair 2 m0 (its LL row is level k+1's even row of pair m0)
	I2 cdl, cdh, cel, ceh;
	{
		const Raw1 before = load1(2 * m0 - 1), first = load1(2 * m0);
		const I2 sl = inv_row_pair(q, a.nquads, e2l, e2h);
		cdl = to_i2(first.dl);
		cdh = to_i2(first.dh);
		cel = i2_sub4(sl, m0 ? to_i2(before.dl) : cdl, cdl);
		ceh = i2_sub4(to_i2(first.sh), m0 ? to_i2(before.dh) : cdh, cdh);
	}
	Raw2 n2 = load2(m0 + 1), c2;
	Raw1 n1a = load1(2 * m0 + 1), n1b = load1(2 * m0 + 2), c1a, c1b;
	typename OutPlane<DstT>::row orow[4];
	auto store_rows = [&](int m) {
		if (!own)
			return;
#pragma unroll
		for (int k = 0; k < 4; ++k)
			OutPlane<DstT>::store(dst + (long)(4 * m + k) * a.opitch + 4 * q, orow[k]);
	};
	// one row pair of level k: state (pair jj) + the next pair's samples -> its two output rows; the state moves on
	auto pair1 = [&](int jj, I2 nsl, const Raw1 &n, typename OutPlane<DstT>::row &even, typename OutPlane<DstT>::row &odd) {
		I2 ndl = { 0, 0 }, ndh = { 0, 0 }, nel = cel, neh = ceh;   // the plane's last pair mirrors: x[h] := x[h-2]
		if (jj + 1 < h2) {
			ndl = to_i2(n.dl);
			ndh = to_i2(n.dh);
			nel = i2_sub4(nsl, cdl, ndl);
			neh = i2_sub4(to_i2(n.sh), cdh, ndh);
		}
		const Quad4 ev = inv_row_vals(q, a.nquads, cel, ceh);
		const Quad4 od = inv_row_vals(q, a.nquads, i2_add2(cdl, cel, nel), i2_add2(cdh, ceh, neh));
		even = OutRow<DstT>::of(ev);
		odd = OutRow<DstT>::of(od);
		cdl = ndl;
		cdh = ndh;
		cel = nel;
		ceh = neh;
	};
	for (int m = m0; m < m1; ++m) {
		c2 = hold(n2);   // the one wait of the iteration (see k_fwd_level_w)
		c1a = hold(n1a);
		c1b = hold(n1b);
		if (m > m0)
			store_rows(m - 1);
		if (m + 1 < m1) {
			n2 = load2(m + 2);
			n1a = load1(2 * m + 3);
			n1b = load1(2 * m + 4);
		}
		// level k+1, pair m: its odd row, and the even row of the pair after it
		int nd2l = 0, nd2h = 0, ne2l = e2l, ne2h = e2h;
		if (m + 1 < h4) {
			nd2l = c2.dl;
			nd2h = c2.dh;
			ne2l = c2.sl - tdiv4(d2l + nd2l);
			ne2h = c2.sh - tdiv4(d2h + nd2h);
		}
		const I2 ll_odd = inv_row_pair(q, a.nquads, d2l + tdiv2(e2l + ne2l), d2h + tdiv2(e2h + ne2h));   // LL row 2m+1 of level k
		const I2 ll_next = inv_row_pair(q, a.nquads, ne2l, ne2h);                                           // LL row 2m+2
		d2l = nd2l;
		d2h = nd2h;
		e2l = ne2l;
		e2h = ne2h;
		pair1(2 * m, ll_odd, c1a, orow[0], orow[1]);
		pair1(2 * m + 1, ll_next, c1b, orow[2], orow[3]);
	}
	store_rows(m1 - 1);
}

// The finest inverse level of an RGB image: one wave carries the same columns of the three planes
// (Y, Co, Cg) and writes interleaved 8-bit pixels — image.h:39-50 ycocg2rgb with its input clamps and
// the output clamp of pnm.h:108 fused in.  grid.z = image; dst8 rows are 3*w bytes.
__device__ __forceinline__ int clamp_to(int v, int lo, int hi)
{
	return v < lo ? lo : v > hi ? hi : v;
}

struct Rgb12 {
	unsigned w[3];   // four pixels
};

__device__ __forceinline__ Rgb12 rgb_of(const Quad4 &y, const Quad4 &co, const Quad4 &cg)
{
	unsigned char px[12];
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const int yy = clamp_to(y.v[k], 0, 255), c0 = clamp_to(co.v[k], -255, 255), c1 = clamp_to(cg.v[k], -255, 255);
		const int t = yy - tdiv2(c1);
		const int g = c1 + t;
		const int b = t - tdiv2(c0);
		const int r = b + c0;
		px[3 * k] = (unsigned char)clamp_to(r, 0, 255);
		px[3 * k + 1] = (unsigned char)clamp_to(g, 0, 255);
		px[3 * k + 2] = (unsigned char)clamp_to(b, 0, 255);
	}
	Rgb12 o;
#pragma unroll
	for (int k = 0; k < 3; ++k)
		o.w[k] = px[4 * k] | (px[4 * k + 1] << 8) | (px[4 * k + 2] << 16) | ((unsigned)px[4 * k + 3] << 24);
	return o;
}

template <bool F16>
__global__ __launch_bounds__(64 * WAVES) void k_inv_level_w_rgb(LevelArgsW A)
{
	const LevelArgs &a = A.a;
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	int bx, by;
	xcd_strip(bx, by);
	const int qd = bx * INV_QUADS - 1 + lane;
	const int j0 = (by * WAVES + wv) * a.rpw;
	if (j0 >= a.h2)
		return;
	const int j1 = min(j0 + a.rpw, a.h2);
	const int image = blockIdx.z;
	const bool valid = qd >= 0 && qd < A.nquads;
	const bool writes = valid && lane >= 1 && lane <= INV_QUADS;
	const int *llp[3];
	typename DetPtr<F16>::type det[3];
	typedef InvRawT<F16> InvRaw;
#pragma unroll
	for (int ch = 0; ch < 3; ++ch) {
		llp[ch] = a.src + (long)(3 * image + ch) * a.src_ps;
		det[ch] = DetPtr<F16>::of(a, 3 * image + ch);
	}
	uint8_t *dst = a.dst8 + image * a.ll_ps;
	const InvAt at = inv_at(a, qd, A.nquads);

	// (one row pair per batch: three planes' worth of arithmetic lies between two waits as it is)
	InvCols c[3];
	InvRaw nxt[3], cur[3];
#pragma unroll
	for (int ch = 0; ch < 3; ++ch) {
		c[ch] = inv_first(a, j0, inv_load_w(a, llp[ch], det[ch], j0 > 0 ? j0 - 1 : 0, at), inv_load_w(a, llp[ch], det[ch], j0, at));
		nxt[ch] = inv_load_w(a, llp[ch], det[ch], j0 + 1, at);
	}
	Rgb12 orow[2];
	auto store_pair = [&](int j) {
		if (writes) {   // twelve bytes per lane in one store: the wave's 56 lanes write 672 consecutive bytes
			U32x3 v0 = { orow[0].w[0], orow[0].w[1], orow[0].w[2] }, v1 = { orow[1].w[0], orow[1].w[1], orow[1].w[2] };
			*reinterpret_cast<U32x3 *>(dst + (long)(2 * j) * a.llpitch + 12 * qd) = v0;
			if (2 * j + 1 < a.h)
				*reinterpret_cast<U32x3 *>(dst + (long)(2 * j + 1) * a.llpitch + 12 * qd) = v1;
		}
	};
	for (int jj = j0; jj < j1; ++jj) {
#pragma unroll
		for (int ch = 0; ch < 3; ++ch)
			cur[ch] = hold(nxt[ch]);   // the one wait of the iteration
		if (jj > j0)
			store_pair(jj - 1);
		if (jj + 1 < j1) {
#pragma unroll
			for (int ch = 0; ch < 3; ++ch)
				nxt[ch] = inv_load_w(a, llp[ch], det[ch], jj + 2, at);
		}
		Quad4 even[3], odd[3];
#pragma unroll
		for (int ch = 0; ch < 3; ++ch)
			inv_pair(a, jj, qd, A.nquads, c[ch], cur[ch], even[ch], odd[ch]);
		orow[0] = rgb_of(even[0], even[1], even[2]);
		orow[1] = rgb_of(odd[0], odd[1], odd[2]);
	}
	store_pair(j1 - 1);
}

// The two finest levels of an RGB picture in one pass: k_inv2_level_w's two stages for the same columns of the three planes
// (Y, Co, Cg; both levels' detail bands as 16-bit values), the colour transform and the clamps of image.h:39-50 / pnm.h:108 on
// the finished rows, interleaved 8-bit pixels out — twelve bytes per lane: a wave's 56 owning lanes write 672 bytes = 21
// 32-byte pieces, the block's four waves side by side 21 whole lines.  Three planes' state and rows in flight: ~150 vector
// registers, three waves per SIMD — and still the faster way: the int32 LL planes of the second level (3 B per pixel written,
// 3 B read) are gone.  grid.z = image.
template <bool F16>
__global__ __launch_bounds__(64 * WAVES) void k_inv2_level_w_rgb(Inv2Args a)
{
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	int bx, by;
	xcd_strip(bx, by);
	const int strip = bx * WAVES + wv;
	if (strip * V2_OWN >= a.nquads)
		return;
	const int q = strip * V2_OWN - V2_FIRST + lane;
	const int h2 = a.h >> 1, h4 = a.h >> 2, w2 = a.w >> 1, w4 = a.w >> 2;
	const int m0 = by * a.mpw;
	if (m0 >= h4)
		return;
	const int m1 = min(m0 + a.mpw, h4);
	const int image = blockIdx.z;
	const bool own = lane >= V2_FIRST && lane < V2_FIRST + V2_OWN && q < a.nquads;
	const int qc = min(max(q + max(V2_FIRST - 1 - lane, 0) - max(lane - (V2_FIRST + V2_OWN), 0), 0), a.nquads - 1);
	typedef typename DetBand<F16>::raw1 Raw1;
	const int *ll2[3];
	typename DetBand<F16>::ptr det[3];
#pragma unroll
	for (int ch = 0; ch < 3; ++ch) {
		ll2[ch] = a.ll2 + (long)(3 * image + ch) * a.ll2_ps + qc;
		det[ch] = DetBand<F16>::of(a, 3 * image + ch);
	}
	uint8_t *dst = a.dst8 + image * a.dst_ps;
	auto load2 = [&](int ch, int m) {
		const int mm = min(max(m, 0), h4 - 1);
		Raw2 r = { ll2[ch][(long)mm * a.ll2pitch], (int)det[ch][(long)mm * a.dpitch + w4 + qc], (int)det[ch][(long)(h4 + mm) * a.dpitch + qc],
			(int)det[ch][(long)(h4 + mm) * a.dpitch + w4 + qc] };
		return r;
	};
	auto load1 = [&](int ch, int j) {
		const int jc = min(max(j, 0), h2 - 1);
		Raw1 r = { DetBand<F16>::pair(det[ch] + (long)jc * a.dpitch + w2 + 2 * qc), DetBand<F16>::pair(det[ch] + (long)(h2 + jc) * a.dpitch + 2 * qc),
			DetBand<F16>::pair(det[ch] + (long)(h2 + jc) * a.dpitch + w2 + 2 * qc) };
		return r;
	};
	int d2l[3], d2h[3], e2l[3], e2h[3];
	I2 cdl[3], cdh[3], cel[3], ceh[3];
	Raw2 n2[3], c2[3];
	Raw1 n1a[3], n1b[3], c1a[3], c1b[3];
#pragma unroll
	for (int ch = 0; ch < 3; ++ch) {
		const Raw2 before2 = load2(ch, m0 - 1), first2 = load2(ch, m0);
		d2l[ch] = first2.dl;
		d2h[ch] = first2.dh;
		e2l[ch] = first2.sl - tdiv4((m0 ? before2.dl : d2l[ch]) + d2l[ch]);
		e2h[ch] = first2.sh - tdiv4((m0 ? before2.dh : d2h[ch]) + d2h[ch]);
		const Raw1 before1 = load1(ch, 2 * m0 - 1), first1 = load1(ch, 2 * m0);
		const I2 sl = inv_row_pair(q, a.nquads, e2l[ch], e2h[ch]);
		cdl[ch] = to_i2(first1.dl);
		cdh[ch] = to_i2(first1.dh);
		cel[ch] = i2_sub4(sl, m0 ? to_i2(before1.dl) : cdl[ch], cdl[ch]);
		ceh[ch] = i2_sub4(to_i2(first1.sh), m0 ? to_i2(before1.dh) : cdh[ch], cdh[ch]);
		n2[ch] = load2(ch, m0 + 1);
		n1a[ch] = load1(ch, 2 * m0 + 1);
		n1b[ch] = load1(ch, 2 * m0 + 2);
	}
	Rgb12 orow[4];
	auto store_rows = [&](int m) {
		if (!own)
			return;
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const U32x3 v = { orow[k].w[0], orow[k].w[1], orow[k].w[2] };
			__builtin_nontemporal_store(v, reinterpret_cast<U32x3 *>(dst + (long)(4 * m + k) * a.opitch + 12 * q));
		}
	};
	for (int m = m0; m < m1; ++m) {
#pragma unroll
		for (int ch = 0; ch < 3; ++ch) {
			c2[ch] = hold(n2[ch]);   // the one wait of the iteration (see k_fwd_level_w)
			c1a[ch] = hold(n1a[ch]);
			c1b[ch] = hold(n1b[ch]);
		}
		if (m > m0)
			store_rows(m - 1);
		if (m + 1 < m1) {
#pragma unroll
			for (int ch = 0; ch < 3; ++ch) {
				n2[ch] = load2(ch, m + 2);
				n1a[ch] = load1(ch, 2 * m + 3);
				n1b[ch] = load1(ch, 2 * m + 4);
			}
		}
		Quad4 rows[3][4];
#pragma unroll
		for (int ch = 0; ch < 3; ++ch) {
			// level k+1, pair m: its odd row, and the even row of the pair after it
			int nd2l = 0, nd2h = 0, ne2l = e2l[ch], ne2h = e2h[ch];
			if (m + 1 < h4) {
				nd2l = c2[ch].dl;
				nd2h = c2[ch].dh;
				ne2l = c2[ch].sl - tdiv4(d2l[ch] + nd2l);
				ne2h = c2[ch].sh - tdiv4(d2h[ch] + nd2h);
			}
			const I2 ll_odd = inv_row_pair(q, a.nquads, d2l[ch] + tdiv2(e2l[ch] + ne2l), d2h[ch] + tdiv2(e2h[ch] + ne2h));
			const I2 ll_next = inv_row_pair(q, a.nquads, ne2l, ne2h);
			d2l[ch] = nd2l;
			d2h[ch] = nd2h;
			e2l[ch] = ne2l;
			e2h[ch] = ne2h;
			// level k, row pairs 2m and 2m+1
#pragma unroll
			for (int half = 0; half < 2; ++half) {
				const int jj = 2 * m + half;
				const I2 nsl = half ? ll_next : ll_odd;
				const Raw1 &n = half ? c1b[ch] : c1a[ch];
				I2 ndl = { 0, 0 }, ndh = { 0, 0 }, nel = cel[ch], neh = ceh[ch];   // the plane's last pair mirrors: x[h] := x[h-2]
				if (jj + 1 < h2) {
					ndl = to_i2(n.dl);
					ndh = to_i2(n.dh);
					nel = i2_sub4(nsl, cdl[ch], ndl);
					neh = i2_sub4(to_i2(n.sh), cdh[ch], ndh);
				}
				rows[ch][2 * half] = inv_row_vals(q, a.nquads, cel[ch], ceh[ch]);
				rows[ch][2 * half + 1] = inv_row_vals(q, a.nquads, i2_add2(cdl[ch], cel[ch], nel), i2_add2(cdh[ch], ceh[ch], neh));
				cdl[ch] = ndl;
				cdh[ch] = ndh;
				cel[ch] = nel;
				ceh[ch] = neh;
			}
		}
#pragma unroll
		for (int k = 0; k < 4; ++k)
			orow[k] = rgb_of(rows[0][k], rows[1][k], rows[2][k]);
	}
	store_rows(m1 - 1);
}

// ------------------------------------------------------------ coarse tail ---
// Once a plane is at most 64x64 all remaining levels run in one workgroup.s LDS:
// one launch per direction replaces five to six latency-bound level launches.
// Two 64x64 int planes in LDS ping-pong between the row and the column pass.

constexpr int TAIL_MAX = 64;
constexpr int TAIL_THREADS = 1024;

struct TailArgs {
	const int *src;  long src_ps;  int spitch;   // fwd: input plane (w0*h0) | inv: unused
	int *dst;        long dst_ps;  int dpitch2;  // inv: output plane (w0*h0) | fwd: unused
	int *pyr;        long pyr_ps;  int ppitch;   // Mallat pyramid
	int nsteps;
	int ws[DWTX_MAX_LEVELS + 2], hs[DWTX_MAX_LEVELS + 2];   // ws[0]xhs[0] is the tail's finest size
};

// forward lifting of sample pair k of a line (stride st) of length n: low and high
__device__ __forceinline__ void tail_fwd_pair(const int *x, int st, int n, int k, int &lo, int &hi)
{
	const int x0 = x[2 * k * st];
	const bool has_odd = 2 * k + 1 < n;
	const int x1 = has_odd ? x[(2 * k + 1) * st] : 0;
	const int xr = 2 * k + 2 < n ? x[(2 * k + 2) * st] : x0;
	const int d = x1 - tdiv2(x0 + xr);
	const int dl = k ? x[(2 * k - 1) * st] - tdiv2(x[(2 * k - 2) * st] + x0) : d;
	lo = has_odd ? x0 + tdiv4(dl + d) : x0;   // odd length: last even sample passes through
	hi = d;
}

__global__ __launch_bounds__(TAIL_THREADS) void k_fwd_tail(TailArgs t)
{
	__shared__ int A[TAIL_MAX * TAIL_MAX];
	__shared__ int B[TAIL_MAX * TAIL_MAX];
	const int plane = blockIdx.x;
	const int *src = t.src + plane * t.src_ps;
	int *pyr = t.pyr + plane * t.pyr_ps;
	int w = t.ws[0], h = t.hs[0];
	for (int i = threadIdx.x; i < w * h; i += TAIL_THREADS) {
		const int y = i / w, x = i - y * w;
		A[y * TAIL_MAX + x] = src[(long)y * t.spitch + x];
	}
	__syncthreads();
	for (int s = 0; s < t.nsteps; ++s) {
		const int w2 = t.ws[s + 1], h2 = t.hs[s + 1];
		for (int i = threadIdx.x; i < h * w2; i += TAIL_THREADS) {   // rows: A -> B
			const int r = i / w2, k = i - r * w2;
			int lo, hi;
			tail_fwd_pair(A + r * TAIL_MAX, 1, w, k, lo, hi);
			B[r * TAIL_MAX + k] = lo;
			if (2 * k + 1 < w)
				B[r * TAIL_MAX + w2 + k] = hi;
		}
		__syncthreads();
		for (int i = threadIdx.x; i < h2 * w; i += TAIL_THREADS) {   // columns: B -> A
			const int j = i / w, c = i - j * w;
			int lo, hi;
			tail_fwd_pair(B + c, TAIL_MAX, h, j, lo, hi);
			A[j * TAIL_MAX + c] = lo;
			if (2 * j + 1 < h)
				A[(h2 + j) * TAIL_MAX + c] = hi;
		}
		__syncthreads();
		for (int i = threadIdx.x; i < w * h; i += TAIL_THREADS) {    // detail subbands out
			const int y = i / w, x = i - y * w;
			if (y >= h2 || x >= w2)
				pyr[(long)y * t.ppitch + x] = A[y * TAIL_MAX + x];
		}
		w = w2;
		h = h2;
	}
	for (int i = threadIdx.x; i < w * h; i += TAIL_THREADS) {        // root LL
		const int y = i / w, x = i - y * w;
		pyr[(long)y * t.ppitch + x] = A[y * TAIL_MAX + x];
	}
}

// inverse lifting: samples 2k and 2k+1 of a line from its low half s[] and high half d[]
__device__ __forceinline__ void tail_inv_pair(const int *s, const int *d, int st, int n, int k, int &e, int &o)
{
	auto even = [&](int j) {
		const int sj = s[j * st];
		if ((n & 1) && 2 * j == n - 1)
			return sj;
		const int dj = d[j * st];
		const int dp = j ? d[(j - 1) * st] : dj;
		return sj - tdiv4(dp + dj);
	};
	e = even(k);
	o = 0;
	if (2 * k + 1 < n) {
		const int en = 2 * k + 2 < n ? even(k + 1) : e;
		o = d[k * st] + tdiv2(e + en);
	}
}

__global__ __launch_bounds__(TAIL_THREADS) void k_inv_tail(TailArgs t)
{
	__shared__ int A[TAIL_MAX * TAIL_MAX];
	__shared__ int B[TAIL_MAX * TAIL_MAX];
	const int plane = blockIdx.x;
	const int *pyr = t.pyr + plane * t.pyr_ps;
	int *dst = t.dst + plane * t.dst_ps;
	{
		const int w = t.ws[t.nsteps], h = t.hs[t.nsteps];
		for (int i = threadIdx.x; i < w * h; i += TAIL_THREADS) {
			const int y = i / w, x = i - y * w;
			A[y * TAIL_MAX + x] = pyr[(long)y * t.ppitch + x];
		}
	}
	for (int s = t.nsteps - 1; s >= 0; --s) {
		const int w = t.ws[s], h = t.hs[s], w2 = t.ws[s + 1], h2 = t.hs[s + 1];
		for (int i = threadIdx.x; i < w * h; i += TAIL_THREADS) {    // detail subbands in
			const int y = i / w, x = i - y * w;
			if (y >= h2 || x >= w2)
				A[y * TAIL_MAX + x] = pyr[(long)y * t.ppitch + x];
		}
		__syncthreads();
		for (int i = threadIdx.x; i < h2 * w; i += TAIL_THREADS) {   // columns: A -> B
			const int j = i / w, c = i - j * w;
			int e, o;
			tail_inv_pair(A + c, A + h2 * TAIL_MAX + c, TAIL_MAX, h, j, e, o);
			B[(2 * j) * TAIL_MAX + c] = e;
			if (2 * j + 1 < h)
				B[(2 * j + 1) * TAIL_MAX + c] = o;
		}
		__syncthreads();
		for (int i = threadIdx.x; i < h * w2; i += TAIL_THREADS) {   // rows: B -> A
			const int r = i / w2, k = i - r * w2;
			int e, o;
			tail_inv_pair(B + r * TAIL_MAX, B + r * TAIL_MAX + w2, 1, w, k, e, o);
			A[r * TAIL_MAX + 2 * k] = e;
			if (2 * k + 1 < w)
				A[r * TAIL_MAX + 2 * k + 1] = o;
		}
		__syncthreads();
	}
	const int w = t.ws[0], h = t.hs[0];
	for (int i = threadIdx.x; i < w * h; i += TAIL_THREADS) {
		const int y = i / w, x = i - y * w;
		dst[(long)y * t.dpitch2 + x] = A[y * TAIL_MAX + x];
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
	DWTX_ENTER(ctx);
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
	DWTX_ENTER(ctx);
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
	DWTX_ENTER(ctx);
	const long npix = (long)W * H, total = npix * n;
	const int blocks = (int)min((total + 255) / 256, (long)256 * 16);
	hipLaunchKernelGGL(k_pixels_from_planes, dim3(blocks), dim3(256), 0, ctx->stream, pix, planes, npix, C, total);
	DWTX_LAUNCH_CHECK();
	return DWTX_OK;
}

static bool aligned_to(const void *p, size_t a) { return ((uintptr_t)p & (a - 1)) == 0; }

// fewer rows per wave on small levels so that a few thousand waves are in flight
static int pick_rpw(int strips_x, int h2, int nplanes)
{
	long waves32 = (long)strips_x * dwtx_cdiv(h2, MAX_ROWS_PER_WAVE) * nplanes;
	int rpw = MAX_ROWS_PER_WAVE;
	while (rpw > 4 && waves32 < 4096) {
		rpw >>= 1;
		waves32 <<= 1;
	}
	return rpw;
}

// in8 != nullptr: the source is 8-bit pixels, gray (in8_channels 1: plane p = image p) or interleaved RGB
// (in8_channels 3: plane p = channel p%3 of image p/3 after YCoCg-R); needs a finest level the wide kernel takes
static int lift_fwd(dwtx_ctx *ctx, int32_t *out, const int32_t *in, const uint8_t *in8, int in8_channels, int W, int H, int nplanes,
	const dwtx_hist_sink *sink = nullptr, unsigned *hist_levels = nullptr, dwtx_p16 p16 = dwtx_p16{ nullptr, 0u })
{
	if (hist_levels)
		*hist_levels = 0u;
	if (!ctx || !out || (!in && !in8) || W < 2 || H < 2 || nplanes < 1 || nplanes > 65535)
		return DWTX_ERR_ARG;
	DWTX_ENTER(ctx);
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
	// histograms ride along from the finest level down for as long as the levels allow it (the wide kernel, blocks
	// that the 16-lane rows cover exactly): pack.hip's k_hist counts the tiles of the levels below
	bool hist_on = sink != nullptr && hist_levels != nullptr;
	{
		dwtx_geom gg;
		if (hist_on && (dwtx_geometry(&gg, W, H) || gg.levels != T))
			hist_on = false;
	}
	int tail_from = T;   // first step that runs inside the LDS tail kernel
	for (int t = 0; t < T; ++t)
		if (ws[t] <= TAIL_MAX && hs[t] <= TAIL_MAX) {
			tail_from = t;
			break;
		}
	// the LL band ping-pongs between the two scratch planes (tmp[0] holds up to ws[1]*hs[1], tmp[1]
	// up to ws[2]*hs[2]); the last level writes it into the pyramid itself
	const int *src = in;
	long src_ps = full_ps;
	int spitch = W;
	if (in8 && tail_from == 0)
		return DWTX_ERR_ARG;
	auto ll_dest = [&](int k, int *&p, long &ps, int &pitch) {   // destination of the ws[k]*hs[k] LL band
		if (k == T) {
			p = out;
			ps = full_ps;
			pitch = W;
			return;
		}
		p = src == tmp[0] ? tmp[1] : tmp[0];   // never the plane being read; from tmp[0] only bands <= ws[2]*hs[2] follow
		ps = (long)ws[k] * hs[k];
		pitch = ws[k];
	};
	for (int t = 0; t < T;) {
		if (t == tail_from) {
			TailArgs ta;
			ta.src = src;
			ta.src_ps = src_ps;
			ta.spitch = spitch;
			ta.dst = nullptr;
			ta.dst_ps = 0;
			ta.dpitch2 = 0;
			ta.pyr = out;
			ta.pyr_ps = full_ps;
			ta.ppitch = W;
			ta.nsteps = T - t;
			for (int k = 0; k <= T - t; ++k) {
				ta.ws[k] = ws[t + k];
				ta.hs[k] = hs[t + k];
			}
			hipLaunchKernelGGL(k_fwd_tail, dim3(nplanes), dim3(TAIL_THREADS), 0, ctx->stream, ta);
			DWTX_LAUNCH_CHECK();
			break;
		}
		// two levels in one pass where the shapes allow it (k_fwd2_level_w): plain int32 planes, no histograms
		if (!in8 && !p16.planes && !hist_on && !ctx->opt[DWTX_OPT_NO_FUSED_LEVELS] && t + 1 < tail_from && t + 2 <= T &&
			ws[t] % 4 == 0 && hs[t] % 4 == 0 && spitch % 4 == 0 && src_ps % 4 == 0 && aligned_to(src, 16) && W % 2 == 0 && aligned_to(out, 8)) {
			Level2Args f;
			f.src = src;
			f.src_ps = src_ps;
			f.spitch = spitch;
			ll_dest(t + 2, f.ll2, f.ll2_ps, f.ll2pitch);
			f.det = out;
			f.det_ps = full_ps;
			f.dpitch = W;
			f.w = ws[t];
			f.h = hs[t];
			f.nquads = ws[t] / 4;
			const int strips = dwtx_cdiv(f.nquads, F2_OWN), h4 = hs[t] / 4;
			f.mpw = F2_MPW;
			while (f.mpw > 2 && (long)strips * dwtx_cdiv(h4, f.mpw) * nplanes < 4096)
				f.mpw >>= 1;
			hipLaunchKernelGGL(k_fwd2_level_w, dim3(dwtx_cdiv(strips, WAVES), dwtx_cdiv(h4, f.mpw), nplanes), dim3(64 * WAVES), 0, ctx->stream, f);
			DWTX_LAUNCH_CHECK();
			src = f.ll2;
			src_ps = f.ll2_ps;
			spitch = f.ll2pitch;
			t += 2;
			continue;
		}
		LevelArgs a;
		a.w = ws[t];
		a.h = hs[t];
		a.w2 = ws[t + 1];
		a.h2 = hs[t + 1];
		const bool bytes_in = in8 && t == 0;
		a.src = bytes_in ? nullptr : src;
		a.src8 = bytes_in ? in8 : nullptr;
		a.dst8 = nullptr;
		a.src_ps = bytes_in ? (long)in8_channels * full_ps : src_ps;
		a.spitch = bytes_in ? in8_channels * W : spitch;
		ll_dest(t + 1, a.ll, a.ll_ps, a.llpitch);
		a.det = out;
		a.det_ps = full_ps;
		a.dpitch = W;
		a.det16 = nullptr;
		a.src16 = nullptr;
		a.ll16 = nullptr;
		// 16-bit bands (dwtx_p16): the levels in the mask write their details there; between two such levels the LL
		// band travels as 16-bit values too (in the scratch planes, which are sized for int32).  Only from 8-bit pixels:
		// that is what bounds the magnitudes (see k_fwd_level_w).
		auto in_mask = [&](int step) { return p16.planes && step < T && step < tail_from && ((p16.levels >> (T - 1 - step)) & 1u); };
		if (in_mask(t)) {
			if (!in8 || (t == 0) != bytes_in || t > 4 || (t > 0 && !in_mask(t - 1)) || !aligned_to(p16.planes, 16))
				return DWTX_ERR_ARG;
			a.det16 = p16.planes;
			if (t > 0)
				a.src16 = reinterpret_cast<const short *>(src);
			if (in_mask(t + 1))
				a.ll16 = reinterpret_cast<short *>(a.ll);
		}
		const bool wide = a.w % 4 == 0 && a.spitch % 4 == 0 && a.src_ps % 4 == 0 &&
			(bytes_in ? aligned_to(a.src8, 4) : aligned_to(a.src, 16)) &&
			a.llpitch % 2 == 0 && a.ll_ps % 2 == 0 && aligned_to(a.ll, 8) &&
			a.dpitch % 2 == 0 && a.det_ps % 2 == 0 && aligned_to(a.det, 8);
		if ((bytes_in || a.det16) && !wide)
			return DWTX_ERR_ARG;   // callers check dwtx_gray8_ok() / dwtx_levels16() first
		const int level = T - 1 - t;   // the ring level this step's detail bands are
		// (the RGB kernel is bound by its own arithmetic — every plane's launch unpacks the pixels and does the colour
		// transform — and pays for the histogram in full: 3.2 -> 4.7 ms per 256 frames of 1080p against the 1.4 ms k_hist takes)
		const bool hist_here = hist_on && wide && a.w2 % 32 == 0 && sink->tiles.nbs[level] > 0;
		if (wide) {
			LevelArgsW A;
			A.nquads = a.w / 4;
			a.rpw = pick_rpw(dwtx_cdiv(A.nquads, 64), a.h2, nplanes);
			const int strips = dwtx_cdiv(A.nquads, 64);
			A.wx_log2 = strips >= 4 ? 2 : strips >= 2 ? 1 : 0;
			const int sx = dwtx_cdiv(strips, 1 << A.wx_log2);
			A.a = a;
			A.hist = HistArgs{ nullptr, nullptr, nullptr, 0, 0, 0 };
			dim3 grid(sx, dwtx_cdiv(a.h2, (WAVES >> A.wx_log2) * a.rpw), nplanes);
			dim3 rgb_grid(sx * 3, grid.y, nplanes / 3);   // the three channels of a strip side by side (xcd_strip_rgb)
			if (hist_here) {
				A.hist.cum32 = sink->cum32;
				A.hist.tile_mx = sink->tile_mx;
				A.hist.xy2tile = sink->tiles.xy2tile + sink->tiles.xy_first[level];
				A.hist.NT = sink->NT;
				A.hist.NTP = sink->NTP;
				A.hist.nbs = sink->tiles.nbs[level];
				*hist_levels |= 1u << level;
				if (bytes_in && in8_channels == 3)
					hipLaunchKernelGGL((k_fwd_pixels_w<Rgb8, true>), rgb_grid, dim3(64 * WAVES), 0, ctx->stream, A);
				else if (bytes_in)
					hipLaunchKernelGGL((k_fwd_pixels_w<uint8_t, true>), grid, dim3(64 * WAVES), 0, ctx->stream, A);
				else if (a.src16)
					hipLaunchKernelGGL((k_fwd_level_w<true, true>), grid, dim3(64 * WAVES), 0, ctx->stream, A);
				else
					hipLaunchKernelGGL((k_fwd_level_w<true, false>), grid, dim3(64 * WAVES), 0, ctx->stream, A);
			} else if (bytes_in && in8_channels == 3)
				hipLaunchKernelGGL((k_fwd_pixels_w<Rgb8, false>), rgb_grid, dim3(64 * WAVES), 0, ctx->stream, A);
			else if (bytes_in)
				hipLaunchKernelGGL((k_fwd_pixels_w<uint8_t, false>), grid, dim3(64 * WAVES), 0, ctx->stream, A);
			else if (a.src16)
				hipLaunchKernelGGL((k_fwd_level_w<false, true>), grid, dim3(64 * WAVES), 0, ctx->stream, A);
			else
				hipLaunchKernelGGL((k_fwd_level_w<false, false>), grid, dim3(64 * WAVES), 0, ctx->stream, A);
		} else {
			const int sx = dwtx_cdiv(a.w2, 64);
			a.rpw = pick_rpw(sx, a.h2, nplanes);
			dim3 grid(sx, dwtx_cdiv(a.h2, WAVES * a.rpw), nplanes);
			hipLaunchKernelGGL(k_fwd_level, grid, dim3(64 * WAVES), 0, ctx->stream, a);
		}
		DWTX_LAUNCH_CHECK();
		src = a.ll;
		src_ps = a.ll_ps;
		spitch = a.llpitch;
		++t;
	}
	return DWTX_OK;
}

extern "C" int dwtx_transformation_fwd(dwtx_ctx *ctx, int32_t *out, const int32_t *in, int W, int H, int nplanes)
{
	if (!in)
		return DWTX_ERR_ARG;
	return lift_fwd(ctx, out, in, nullptr, 0, W, H, nplanes);
}

// Can the finest level of a W*H gray image read / write 8-bit pixels directly?  (wide kernel, not the LDS tail)
bool dwtx_gray8_ok(int W, int H, const void *pix, size_t image_stride)
{
	return W % 4 == 0 && (W > TAIL_MAX || H > TAIL_MAX) && image_stride % 4 == 0 && aligned_to(pix, 4);
}

int dwtx_fwd_pixels8_hist(dwtx_ctx *ctx, int32_t *out, const uint8_t *pix, int W, int H, int C, int n, const dwtx_hist_sink *sink, unsigned *hist_levels,
	dwtx_p16 p16)
{
	if (!pix || (C != 1 && C != 3) || !dwtx_gray8_ok(W, H, pix, (size_t)W * H * C))
		return DWTX_ERR_ARG;
	return lift_fwd(ctx, out, nullptr, pix, C, W, H, n * C, sink, hist_levels, p16);
}

int dwtx_transformation_fwd_hist(dwtx_ctx *ctx, int32_t *out, const int32_t *in, int W, int H, int nplanes, const dwtx_hist_sink *sink,
	unsigned *hist_levels)
{
	if (!in)
		return DWTX_ERR_ARG;
	return lift_fwd(ctx, out, in, nullptr, 0, W, H, nplanes, sink, hist_levels);
}

int dwtx_fwd_pixels8(dwtx_ctx *ctx, int32_t *out, const uint8_t *pix, int W, int H, int C, int n)
{
	if (!pix || W < 2 || H < 2 || (C != 1 && C != 3) || !dwtx_gray8_ok(W, H, pix, (size_t)W * H * C))
		return DWTX_ERR_ARG;
	return lift_fwd(ctx, out, nullptr, pix, C, W, H, n * C);
}

// out8 != nullptr: the finest level writes clamped 8-bit pixels (gray, or interleaved RGB after the
// inverse colour transform when out8_channels == 3), image i at out8 + i*out8_ps
static int lift_inv(dwtx_ctx *ctx, int32_t *out, uint8_t *out8, long out8_ps, int out8_channels, const int32_t *in, int W, int H, int nplanes,
	const dwtx_p16 *p16 = nullptr)
{
	if (!ctx || (!out && !out8) || !in || W < 2 || H < 2 || nplanes < 1 || nplanes > 65535)
		return DWTX_ERR_ARG;
	DWTX_ENTER(ctx);
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
	// The LL band the next step reads: the pyramid's root, then the output of the step before.  Intermediate planes live in
	// two scratch planes — tmp[1] holds up to ws[1]*hs[1] samples, tmp[0] up to ws[2]*hs[2] — and a step never writes the
	// plane it reads: the ws[k] x hs[k] plane goes to tmp[(k + flip) & 1].  One step at a time alternates by itself; a
	// two-level step (k_inv2_level_w) reads plane k+2 and writes plane k, same parity, so it flips the assignment for
	// everything after it — allowed only if all later planes still fit (fuse_ok looks ahead).
	const int *cur = in;
	long cur_ps = full_ps;
	int cur_pitch = W;
	int flip = 0;
	auto plane_of = [&](int k, int fl) -> int * { return tmp[(k + fl) & 1]; };
	auto fits = [&](int k, int fl) { return ((k + fl) & 1) == 1 || k >= 2; };   // (tmp[0] is the small one)
	// two levels per pass (k_inv2_level_w): int32 planes throughout, or — the codec's pipelines — both levels' detail bands as 16-bit values
	// (dwtx_p16), the finest step then writing a gray picture's 8-bit pixels itself; an RGB picture's last step keeps its own kernel
	const bool have16 = p16 && p16->planes;
	const bool fusing = !ctx->opt[DWTX_OPT_NO_FUSED_LEVELS] && W % 4 == 0 && aligned_to(in, 8) && (out8 ? aligned_to(out8, 4) && out8_ps % 4 == 0 : aligned_to(out, 16)) &&
		(!have16 || aligned_to(p16->planes, 4));
	auto in16 = [&](int t) { return have16 && ((p16->levels >> (T - 1 - t)) & 1u) != 0; };   // step t's detail bands are 16-bit values
	auto can_fuse = [&](int t) {
		if (!fusing || t < 1 || ws[t - 1] % 4 != 0 || hs[t - 1] % 4 != 0 || in16(t) != in16(t - 1))
			return false;
		if (t - 1 == 0 && out8)   // (the 8-bit variants exist for 16-bit bands only: what the pipelines run; RGB: three planes per wave)
			return in16(0) && (out8_channels == 1 || (out8_channels == 3 && nplanes % 3 == 0));
		return true;
	};
	// Steps t .. 0 with the planes from here on assigned with `fl`: the most samples that can go through two-level steps (a pair is
	// worth the plane it writes: fusing the two finest levels saves sixteen times what the pair below them saves), -1 if the planes
	// do not fit.  At most 2^16 paths, a handful in practice.
	auto best = [&](auto &&self, int t, int fl) -> long {
		if (t < 0)
			return 0;
		long b = -1;
		if (can_fuse(t) && (t - 1 == 0 || fits(t - 1, fl ^ 1))) {
			const long r = self(self, t - 2, fl ^ 1);
			if (r >= 0)
				b = r + (long)ws[t - 1] * hs[t - 1];
		}
		if (t == 0 || fits(t, fl)) {
			const long r = self(self, t - 1, fl);
			if (r > b)
				b = r;
		}
		return b;
	};
	// take the two-level step at t if no plan that takes one step here does better
	auto fuse_ok = [&](int t) {
		if (!can_fuse(t) || cur == in || !(t - 1 == 0 || fits(t - 1, flip ^ 1)))
			return false;
		const long with = best(best, t - 2, flip ^ 1);
		if (with < 0)
			return false;
		const long without = (t == 0 || fits(t, flip)) ? best(best, t - 1, flip) : -1;
		return with + (long)ws[t - 1] * hs[t - 1] >= without;
	};
	int tail_from = T;
	for (int t = 0; t < T; ++t)
		if (ws[t] <= TAIL_MAX && hs[t] <= TAIL_MAX) {
			tail_from = t;
			break;
		}
	if (tail_from < T) {
		const int t = tail_from;
		TailArgs ta;
		ta.src = nullptr;
		ta.src_ps = 0;
		ta.spitch = 0;
		if (t == 0 && out8)
			return DWTX_ERR_ARG;
		ta.dst = t == 0 ? out : tmp[t & 1];
		ta.dst_ps = t == 0 ? full_ps : (long)ws[t] * hs[t];
		ta.dpitch2 = t == 0 ? W : ws[t];
		cur = ta.dst;
		cur_ps = ta.dst_ps;
		cur_pitch = ta.dpitch2;
		ta.pyr = const_cast<int *>(in);
		ta.pyr_ps = full_ps;
		ta.ppitch = W;
		ta.nsteps = T - t;
		for (int k = 0; k <= T - t; ++k) {
			ta.ws[k] = ws[t + k];
			ta.hs[k] = hs[t + k];
		}
		hipLaunchKernelGGL(k_inv_tail, dim3(nplanes), dim3(TAIL_THREADS), 0, ctx->stream, ta);
		DWTX_LAUNCH_CHECK();
	}
	// step t rebuilds the ws[t]*hs[t] plane; its output goes to tmp[t&1] (t odd: the big one)
	for (int t = tail_from - 1; t >= 0; --t) {
		// two levels in one pass (k_inv2_level_w): steps t and t-1 — plain int32 planes, shapes whose two levels have whole row pairs
		if (fuse_ok(t)) {
			Inv2Args f;
			f.ll2 = cur;           // the LL band step t reads: ws[t+1] x hs[t+1]
			f.ll2_ps = cur_ps;
			f.ll2pitch = cur_pitch;
			f.det = in;
			f.det_ps = full_ps;
			f.dpitch = W;
			f.det16 = in16(t) ? p16->planes : nullptr;
			f.dst8 = nullptr;
			if (t - 1 == 0 && out8) {
				f.dst = nullptr;
				f.dst8 = out8;
				f.dst_ps = out8_ps;
				f.opitch = W;
			} else if (t - 1 == 0) {
				f.dst = out;
				f.dst_ps = full_ps;
				f.opitch = W;
			} else {
				f.dst = plane_of(t - 1, flip ^ 1);
				f.dst_ps = (long)ws[t - 1] * hs[t - 1];
				f.opitch = ws[t - 1];
			}
			f.w = ws[t - 1];
			f.h = hs[t - 1];
			f.nquads = f.w / 4;
			{
				const int strips = dwtx_cdiv(f.nquads, V2_OWN), h4 = f.h / 4;
				f.mpw = F2_MPW;
				while (f.mpw > 2 && (long)strips * dwtx_cdiv(h4, f.mpw) * nplanes < 4096)
					f.mpw >>= 1;
				if (f.dst8 && out8_channels == 3) {
					f.opitch = 3 * W;
					hipLaunchKernelGGL(k_inv2_level_w_rgb<true>, dim3(dwtx_cdiv(strips, WAVES), dwtx_cdiv(h4, f.mpw), nplanes / 3), dim3(64 * WAVES), 0, ctx->stream, f);
				} else if (f.dst8)
					hipLaunchKernelGGL((k_inv2_level_w<uint8_t, true>), dim3(dwtx_cdiv(strips, WAVES), dwtx_cdiv(h4, f.mpw), nplanes), dim3(64 * WAVES), 0, ctx->stream, f);
				else if (f.det16)
					hipLaunchKernelGGL((k_inv2_level_w<int, true>), dim3(strips, dwtx_cdiv(h4, WAVES * f.mpw), nplanes), dim3(64 * WAVES), 0, ctx->stream, f);
				else
					hipLaunchKernelGGL((k_inv2_level_w<int, false>), dim3(strips, dwtx_cdiv(h4, WAVES * f.mpw), nplanes), dim3(64 * WAVES), 0, ctx->stream, f);
				DWTX_LAUNCH_CHECK();
				cur = f.dst;
				cur_ps = f.dst_ps;
				cur_pitch = f.opitch;
				flip ^= 1;
				--t;   // (the loop's own step takes the second)
				continue;
			}
		}
		LevelArgs a;
		a.w = ws[t];
		a.h = hs[t];
		a.w2 = ws[t + 1];
		a.h2 = hs[t + 1];
		a.src = cur;   // (without a tail the first step reads the root LL in the pyramid itself)
		a.src_ps = cur_ps;
		a.spitch = cur_pitch;
		const bool bytes_out = out8 && t == 0;
		a.src8 = nullptr;
		a.dst8 = nullptr;
		if (bytes_out) {
			a.ll = nullptr;
			a.dst8 = out8;
			a.ll_ps = out8_ps;
			a.llpitch = out8_channels * W;
		} else if (t == 0) {
			a.ll = out;
			a.ll_ps = full_ps;
			a.llpitch = W;
		} else {
			a.ll = plane_of(t, flip);
			if (!fits(t, flip) || a.ll == cur) {   // (fuse_ok's look-ahead keeps this from happening)
				dwtx_set_error("inverse transform: no scratch plane for step %d", t);
				return DWTX_ERR_ARG;
			}
			a.ll_ps = (long)a.w * a.h;
			a.llpitch = a.w;
		}
		a.det = const_cast<int *>(in);
		a.det_ps = full_ps;
		a.dpitch = W;
		a.det16 = nullptr;
		if (p16 && p16->planes && ((p16->levels >> (T - 1 - t)) & 1u)) {   // this level's detail bands are 16-bit values
			if (!out8 || !aligned_to(p16->planes, 4))
				return DWTX_ERR_ARG;
			a.det16 = p16->planes;
		}
		const bool wide = a.w % 4 == 0 && a.llpitch % 4 == 0 && a.ll_ps % 4 == 0 &&
			(bytes_out ? aligned_to(a.dst8, 4) : aligned_to(a.ll, 16)) &&
			a.spitch % 2 == 0 && a.src_ps % 2 == 0 && aligned_to(a.src, 8) &&
			a.dpitch % 2 == 0 && a.det_ps % 2 == 0 && aligned_to(a.det, 8);
		if ((bytes_out || a.det16) && !wide)
			return DWTX_ERR_ARG;
		if (wide) {
			LevelArgsW A;
			A.nquads = a.w / 4;
			const int per_wave = bytes_out && out8_channels == 3 ? INV_QUADS : 64;   // (the RGB kernel's waves overlap by a lane on each side)
			a.rpw = pick_rpw(dwtx_cdiv(A.nquads, per_wave), a.h2, nplanes);
			const int sx = dwtx_cdiv(A.nquads, per_wave);
			A.wx_log2 = 0;
			A.a = a;
			dim3 grid(sx, dwtx_cdiv(a.h2, WAVES * a.rpw), nplanes);
			if (bytes_out && out8_channels == 3) {
				grid.z = nplanes / 3;
				if (a.det16)
					hipLaunchKernelGGL(k_inv_level_w_rgb<true>, grid, dim3(64 * WAVES), 0, ctx->stream, A);
				else
					hipLaunchKernelGGL(k_inv_level_w_rgb<false>, grid, dim3(64 * WAVES), 0, ctx->stream, A);
			} else if (bytes_out && a.det16)
				hipLaunchKernelGGL((k_inv_level_w<uint8_t, true>), grid, dim3(64 * WAVES), 0, ctx->stream, A);
			else if (bytes_out)
				hipLaunchKernelGGL((k_inv_level_w<uint8_t, false>), grid, dim3(64 * WAVES), 0, ctx->stream, A);
			else if (a.det16)
				hipLaunchKernelGGL((k_inv_level_w<int, true>), grid, dim3(64 * WAVES), 0, ctx->stream, A);
			else
				hipLaunchKernelGGL((k_inv_level_w<int, false>), grid, dim3(64 * WAVES), 0, ctx->stream, A);
		} else {
			const int sx = dwtx_cdiv(a.w2, INV_PAIRS);
			a.rpw = pick_rpw(sx, a.h2, nplanes);
			dim3 grid(sx, dwtx_cdiv(a.h2, WAVES * a.rpw), nplanes);
			hipLaunchKernelGGL(k_inv_level, grid, dim3(64 * WAVES), 0, ctx->stream, a);
		}
		DWTX_LAUNCH_CHECK();
		cur = a.ll;   // (null after the last step, which may have written 8-bit pixels: nothing reads it)
		cur_ps = a.ll_ps;
		cur_pitch = a.llpitch;
	}
	return DWTX_OK;
}

extern "C" int dwtx_transformation_inv(dwtx_ctx *ctx, int32_t *out, const int32_t *in, int W, int H, int nplanes)
{
	if (!out)
		return DWTX_ERR_ARG;
	return lift_inv(ctx, out, nullptr, 0, 0, in, W, H, nplanes);
}

int dwtx_inv_pixels8(dwtx_ctx *ctx, uint8_t *pix, size_t image_stride, const int32_t *in, int W, int H, int C, int n, const dwtx_p16 *p16)
{
	if (!pix || W < 2 || H < 2 || (C != 1 && C != 3) || !dwtx_gray8_ok(W, H, pix, image_stride))
		return DWTX_ERR_ARG;
	return lift_inv(ctx, nullptr, pix, (long)image_stride, C, in, W, H, n * C, p16);
}
