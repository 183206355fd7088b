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
//             difference of two entries; max plane count per channel by atomicMax.
//   k_plan    per image (serial, tiny): header + root image + plane counts
//             written straight into the stream, the segment schedule, VLI order
//             after the header.
//   k_entries_* an "entry" is (segment, tile).  Symbol counts per entry from the
//             histograms, exclusive scans -> first token slot and refinement rank
//             of every entry, per segment its refinement block in the staging buffer.
//   k_code    one wave per tile, every lane owns 16 CONSECUTIVE coefficients, all
//             planes in one pass over the coefficients (read once): per-lane counts
//             of "magnitude below 2^q" packed into nibble fields, one cross-lane DPP
//             scan of the packed counts, then each non-zero coefficient knows the
//             plane it turns significant in, the zeros before it and its rank among
//             that plane's ones from two table look-ups -> 16-bit tokens (zero run
//             since the previous one of the tile, sign) written in stream order;
//             refinement bits compressed per lane and plane and written to a
//             staging bit buffer at their final rank (encode.c:84-93).
//   k_carry_* batch-wide segmented scan of pending zero runs across entries,
//             segment ends (phantom terminators, rle.h:79-89) and the final
//             flush; patches the first token of each entry.
//   k_gorder  the VLI order recurrence o' = max(ilog2(v+2^o)-2,0): one lane per
//             64-token group walks the chains started at order 0 and 31 (they
//             almost always meet: the group's exit order is then known whatever
//             it was entered with) and, with its entry order, the group's bit
//             count; k_lut/k_chain*/k_gorder_exact redo flagged images exactly as
//             a scan over monotone maps on 32 states.
//   k_bitscan per image: exclusive scan of chunk bit totals -> bit offsets.
//   k_clear_stream  zeroes the words the stream will occupy.
//   k_emit    one lane per 64-token group, entry order and bit position known:
//             walks its tokens once more and writes their codes (bits.h:58-78)
//             through an LDS window of the wave's stretch of the stream.
//   k_refcopy refinement blocks from the staging buffer to their place behind
//             each segment's tokens (a shifted copy).
#include "hilbert_dev.h"

#include <stdlib.h>
#include <string.h>

namespace {

constexpr int TILE = 1024;
constexpr int NCUM = 32;              // cum[0..31]
constexpr int MAX_PLANES = 16;        // 8-bit sources stay far below (checked in k_plan)
constexpr int MAX_SEGS = 3 * 16 * MAX_PLANES;
constexpr int SUB = 64;               // tokens per lane
constexpr int CHUNK = SUB * 64;       // tokens per wave
constexpr int GROUP = 64;             // chunks per group

// A token is 16 bits: zero run (0xfff: the run is in tok_big[t]), sign of the one that ends it, and what
// kind of slot it is.  Ones carry a sign; a segment's break slot (phantom terminator, rle.h:79-89) and the
// final flush (encode.c:221) do not; a void slot emits nothing (but a break still has its refinement block).
constexpr unsigned T_RUN = 0x0fffu, T_ESC = 0x0fffu, T_BREAK = 1u << 13, T_VOID = 1u << 14, T_NOSIGN = 1u << 15;   // sign: bit 12

struct PackGeom {
	int levels, C, W, H;
	long total;
	const int *pyr;        // wavelet pyramid of the same planes (pitch W), or null
	const short *fine16;   // or null: the rings of the levels in lv16 are not in pyr but here, as 16-bit coefficients (same pitch and positions)
	unsigned lv16;
	unsigned sq_levels;    // ring levels whose tiles are read from the pyramid's 32x32 squares instead of `lin` (hilbert_dev.h)
	int side[DWTX_MAX_LEVELS + 1];          // outer side of ring level l (lengths[l+1])
	int pixels[DWTX_MAX_LEVELS + 1];
	int tile_first[DWTX_MAX_LEVELS + 1];   // tile_first[levels] = tiles per plane
	// the tiles (dwtx_tiles): ring index of a tile's first coefficient, its coefficients, its block on the level's curve
	const int *tile_base;
	const unsigned short *tile_cnt;
	const int *tile_blk;
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
	unsigned cut;          // segments of the schedule that are not coded: they start beyond CAPACITY (k_cut)
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
	unsigned *tile_mx;          // [nplanes][NTP] OR of the tile's magnitudes: its own bit-plane count is 1 + ilog2 of it (k_hist or the forward
	                            // transform -> k_plan, k_code); NTP = NT rounded up to 4
	// per image
	ImgInfo *info;              // [n]
	int *seg_desc;              // [n][MAX_SEGS]   c | l<<4 | (p+1)<<8
	int *seg_ebase;             // [n][MAX_SEGS+1]
	unsigned *seg_refs;         // [n][MAX_SEGS]
	unsigned long long *seg_rawoff; // [n][MAX_SEGS] bit offset of the segment's refinement block in the stream
	unsigned *brk_tok;          // [n][MAX_SEGS] token index of the segment's break slot
	int *segidx;                // [n][3][16][MAX_PLANES] -> k+1 of the segment coding (channel, level, plane)
	unsigned *live;             // [n][3][16] bit p: plane p of (channel, level) is coded (its segment starts inside CAPACITY; k_cut)
	// per entry
	unsigned short *ent_ones, *ent_zeros, *ent_refs, *ent_tz;   // [n][ES]
	unsigned short *ent_seg;    // [n][ES] the entry's segment (k_entries_count looks it up once)
	unsigned *ent_tokbase;      // [n][ES+1]
	unsigned *ent_refscum;      // [n][ES+1]
	unsigned *ent_refw;         // [n][ES+1] first word of the entry's refinement bits in the staging buffer (every entry starts on a word)
	// per token
	unsigned short *tok16;      // [n][TS]
	unsigned *tok_big;          // [n][TS] only touched where tok16 says T_ESC
	// refinement bits of every segment, in coefficient order, each segment's block starting on a word
	unsigned *stage;            // [n][SW]
	// per 64-token group / per chunk of groups
	unsigned char *grp_ord;     // [n][NCS*64] VLI order the group is entered with
	unsigned long long *lane_bits;     // [n][NCS*64] bit offset of the group inside its chunk
	unsigned char *sublut;      // [n][NCS*64][32]
	unsigned char *lut;         // [n][NCS][32]
	unsigned char *glut;        // [n][NGS][32]
	unsigned char *chunk_entry; // [n][NCS]
	unsigned char *group_entry; // [n][NGS]
	unsigned long long *chunk_bits;    // [n][NCS]
	unsigned long long *chunk_base;    // [n][NCS]
	RunMap *carry_agg;                 // [n][NCB] map of each block of 1024 entries
	unsigned *carry_in;                // [n][NCB] pending run entering the block
	unsigned *ent_blk;                 // [n][NCB][3] token slots / refinement bits / staging words of each block of 1024 entries, then their scan
	unsigned long long *stream_bits;   // [n] bits of the whole stream before any capacity clip (k_bitscan -> k_clear_stream)
	int *slow;                         // [n] set when the fast order pass could not resolve an image
	long ES, TS, NCS, NGS, NCB, SW;
	int NT, NTP;
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

// Lanes of ONE wave hand data to each other through LDS: the wave's DS operations execute in order, so all
// that is needed is that the compiler keeps the accesses on their side of this point.
__device__ __forceinline__ void wave_sync()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// inclusive prefix sum over the 64 lanes in six DPP adds: row_shr 1,2,4,8 inside the rows of 16, then
// row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3
__device__ __forceinline__ unsigned wave_incl_add(unsigned v)
{
	v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
	v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
	v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
	v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
	v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
	v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
	return v;
}

// vli.h:67-84 in closed form: order o, value v -> o* 
// (1 << width) - 1 in one instruction (v_bfm_b32 takes five bits of the width; the compiler spells it with a shift and a not)
__device__ __forceinline__ unsigned bfm_mask(unsigned width)
{
	unsigned r;
	asm("v_bfm_b32 %0, %1, 0" : "=v"(r) : "v"(width));
	return r;
}
__device__ __forceinline__ int vli_top(int o, unsigned v) { return ilog2u(v + (1u << o)); }
__device__ __forceinline__ int vli_next(int top) { return top >= 2 ? top - 2 : 0; }

__device__ __forceinline__ void seg_unpack(int d, int &c, int &l, int &p)
{
	c = d & 15;
	l = (d >> 4) & 15;
	p = (d >> 8) - 1;
}

// The 16 coefficients of lane L (ring indices 16L .. 16L+15 of tile j of ring level l): from the linearised
// plane, or — for the levels flagged in sq_levels — from the pyramid's 32x32 square that holds exactly this tile.
struct __attribute__((packed, aligned(4))) Int4U {
	int x, y, z, w;
};

__device__ __forceinline__ void load_tile16(const PackGeom &g, const int *__restrict__ lin, int plane, int l, int tile, int lane, int nvalid,
	int nv, unsigned *lds, int (&val)[16])
{
	if (((g.sq_levels >> l) & 1u) && nvalid == TILE) {   // uniform
		if ((g.lv16 >> l) & 1u)
			load_square16(g.fine16 + (long)plane * g.total, g.W, g.side[l], g.tile_blk[tile], lane, lds, val);
		else
			load_square16(g.pyr + (long)plane * g.total, g.W, g.side[l], g.tile_blk[tile], lane, lds, val);
		return;
	}
	const int tbase = g.tile_base[tile];
	const int *src = lin + (long)plane * g.total + g.pixels[l] + tbase + 16 * lane;
	// A ring starts wherever the levels before it end: its tiles are 16-byte aligned only by luck (always for
	// power-of-two shapes).  A 16-byte load from a 4-byte aligned address is split up by the memory pipeline, so
	// an unaligned tile is read as five aligned quads around the lane's 16 coefficients and shifted in registers
	// by the ring's (uniform) misalignment.  The quad after a ring's last tile may lie outside the buffer: that
	// tile takes the scalar path.
	const int k = (int)(((uintptr_t)src >> 2) & 3);   // uniform: lanes are 64 bytes apart
	const bool last_of_ring = g.pixels[l] + (long)tbase + TILE >= g.pixels[l + 1];
	if (nvalid == TILE && (k == 0 || !last_of_ring)) {
		const int4 *A = reinterpret_cast<const int4 *>(src - k);
		int t[20];
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			const int4 v4 = A[q];
			t[4 * q] = v4.x;
			t[4 * q + 1] = v4.y;
			t[4 * q + 2] = v4.z;
			t[4 * q + 3] = v4.w;
		}
		t[16] = t[17] = t[18] = t[19] = 0;
		if (k) {
			const int4 v4 = A[4];
			t[16] = v4.x;
			t[17] = v4.y;
			t[18] = v4.z;
		}
		switch (k) {
		case 0:
#pragma unroll
			for (int i = 0; i < 16; ++i)
				val[i] = t[i];
			break;
		case 1:
#pragma unroll
			for (int i = 0; i < 16; ++i)
				val[i] = t[i + 1];
			break;
		case 2:
#pragma unroll
			for (int i = 0; i < 16; ++i)
				val[i] = t[i + 2];
			break;
		default:
#pragma unroll
			for (int i = 0; i < 16; ++i)
				val[i] = t[i + 3];
			break;
		}
	} else {
#pragma unroll
		for (int i = 0; i < 16; ++i)
			val[i] = i < nv ? src[i] : 0;
	}
}

// ------------------------------------------------------------------ k_hist ---
// Lane L owns coefficients 16L .. 16L+15 of the tile (four 16-byte loads).  With t = number of magnitude
// bits, adding 0x1111.. << 4t to a 64-bit register counts "t <= q" for all q = 0..15 at once in its
// nibbles (two registers of eight coefficients each: a nibble holds up to 8); the lane totals, widened
// to 16-bit fields, are summed over the wave with DPP adds.

// the loads of one tile (issued, not waited for); ok = which of the lane's 16 values exist
__device__ __forceinline__ void hist_load(const PackGeom &g, const int *__restrict__ lin, int plane, int tile, int lane, int (&val)[16],
	unsigned &ok, int &nvalid)
{
	int l = 0;
	while (l + 1 < g.levels && tile >= g.tile_first[l + 1])
		++l;
	const long base = g.pixels[l] + g.tile_base[tile];
	nvalid = g.tile_cnt[tile];
	ok = 0xffffu;
	if (((g.sq_levels >> l) & 1u) && nvalid == TILE) {
		// the tile is a 32x32 square of the pyramid (hilbert_dev.h); a histogram does not care about the order:
		// every lane takes four consecutive coefficients of four rows
		const SquareMap m = square_map(g.side[l], (unsigned)g.tile_blk[tile]);
		const long at = (long)plane * g.total + (long)(m.my & ~31u) * g.W + (m.mx & ~31u);
		if ((g.lv16 >> l) & 1u) {   // 16-bit rows: eight coefficients of two rows
			const short *sq = g.fine16 + at;
#pragma unroll
			for (int it = 0; it < 2; ++it) {
				const uint4 v4 = *reinterpret_cast<const uint4 *>(sq + (long)(it * 16 + (lane >> 2)) * g.W + (lane & 3) * 8);
				const unsigned u[4] = { v4.x, v4.y, v4.z, v4.w };
#pragma unroll
				for (int k = 0; k < 4; ++k) {
					val[8 * it + 2 * k] = (int)(short)(u[k] & 0xffffu);
					val[8 * it + 2 * k + 1] = (int)u[k] >> 16;
				}
			}
		} else {
			const int *sq = g.pyr + at;
#pragma unroll
			for (int it = 0; it < 4; ++it) {
				const int4 v4 = *reinterpret_cast<const int4 *>(sq + (long)(it * 8 + (lane >> 3)) * g.W + (lane & 7) * 4);
				val[4 * it] = v4.x;
				val[4 * it + 1] = v4.y;
				val[4 * it + 2] = v4.z;
				val[4 * it + 3] = v4.w;
			}
		}
	} else {
		// the same freedom on the linearised plane: the wave reads the tile front to back (lane-serial loads,
		// 64 bytes apart, touch every cache line of the tile four times).  ok = which of the 16 exist.
		const int *src = lin + (long)plane * g.total + base;
		if (nvalid == TILE && (((uintptr_t)src >> 2) & 3) == 0) {
#pragma unroll
			for (int it = 0; it < 4; ++it) {
				const int4 v4 = reinterpret_cast<const int4 *>(src)[it * 64 + lane];
				val[4 * it] = v4.x;
				val[4 * it + 1] = v4.y;
				val[4 * it + 2] = v4.z;
				val[4 * it + 3] = v4.w;
			}
		} else {
#pragma unroll
			for (int i = 0; i < 16; ++i) {
				const int e = i * 64 + lane;
				val[i] = e < nvalid ? src[e] : 0;
				ok &= ~((e < nvalid ? 0u : 1u) << i);
			}
		}
	}
}

__device__ __forceinline__ void hist_finish(const Work &w, int plane, int tile, int lane, const int (&val)[16], unsigned ok, int nvalid)
{
	constexpr unsigned long long ONES = 0x1111111111111111ull, M0F = 0x0f0f0f0f0f0f0f0full;
	unsigned long long Ra = 0, Rb = 0;
	unsigned mx = 0;
	if (ok == 0xffffu) {   // (uniform) a whole tile, nearly always: no per-coefficient masking
#pragma unroll
		for (int i = 0; i < 16; ++i) {
			const int v = val[i];
			const unsigned a = (unsigned)(v < 0 ? -v : v);
			mx |= a;
			const int t = a ? 32 - __builtin_clz(a) : 0;
			const unsigned long long m = t < 16 ? ONES << (4 * t) : 0ull;
			if (i < 8)
				Ra += m;
			else
				Rb += m;
		}
	} else {
#pragma unroll
		for (int i = 0; i < 16; ++i) {
			const int v = val[i];
			const unsigned a = (unsigned)(v < 0 ? -v : v);
			mx |= a;
			int t = a ? 32 - __builtin_clz(a) : 0;
			t = (ok >> i) & 1u ? t : 16;   // past the ring's end: counted nowhere
			const unsigned long long m = t < 16 ? ONES << (4 * t) : 0ull;
			if (i < 8)
				Ra += m;
			else
				Rb += m;
		}
	}
	const unsigned long long ev = (Ra & M0F) + (Rb & M0F);                 // byte b: #(t <= 2b) of this lane
	const unsigned long long od = ((Ra >> 4) & M0F) + ((Rb >> 4) & M0F);   // byte b: #(t <= 2b+1)
	unsigned D[8];   // D[b] = #(t <= 2b) | #(t <= 2b+1) << 16 over the whole tile (lane 63 after the scan)
#pragma unroll
	for (int b = 0; b < 8; ++b) {
		const unsigned x = ((unsigned)(ev >> (8 * b)) & 0xffu) | ((unsigned)(od >> (8 * b)) & 0xffu) << 16;
		D[b] = wave_incl_add(x);
	}
	for (int o = 32; o; o >>= 1)
		mx |= __shfl_xor(mx, o);
	unsigned short *cum = w.cum + ((long)plane * w.NT + tile) * NCUM;
	if (lane == 63) {
		*reinterpret_cast<uint4 *>(cum) = make_uint4(D[0], D[1], D[2], D[3]);
		*reinterpret_cast<uint4 *>(cum + 8) = make_uint4(D[4], D[5], D[6], D[7]);
	} else if (lane >= 16 && lane < NCUM) {
		// |v| < 2^16 for every stream this coder accepts (k_plan checks the plane count)
		// The last entry is the tile's own bit-plane count, 1 + ilog2(max |v|): k_plan takes the maximum over the
		// plane's tiles (encode.c:130, over the detail rings only, encode.c:165).  (An atomicMax per tile on one
		// word per plane, even filtered by a plain read first, serialises at the memory side: the eight XCDs' L2s
		// cannot hold a device-coherent word, and that cost 0.7 ms per 400 000 tiles.)
		cum[lane] = (unsigned short)(lane == NCUM - 1 ? (mx ? ilog2u(mx) + 1 : 0) : nvalid);
	}
	if (lane == 0)   // the magnitudes' OR once more in a dense array: k_plan reduces it, k_code asks it before loading the tile
		w.tile_mx[(long)plane * w.NTP + tile] = mx;
}

// Two tiles per wave: both tiles' loads are in flight before the first is counted (the kernel only waits for memory).
constexpr int HIST_TPW = 2;

// What the forward transform adds its histograms to (dwtx_hist_begin): a tile's record with the counts at zero, the
// fields past the last plane at the tile's size, and no magnitude seen yet.
__global__ __launch_bounds__(256) void k_hist_init(PackGeom g, Work w)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;   // dword i of the plane's records
	const int plane = blockIdx.y;
	if (i >= w.NT * (NCUM / 2))
		return;
	const int tile = i / (NCUM / 2), d = i - tile * (NCUM / 2);
	const unsigned cnt = g.tile_cnt[tile];
	reinterpret_cast<unsigned *>(w.cum + ((long)plane * w.NT + tile) * NCUM)[d] = d < 8 ? 0u : d < 15 ? cnt * 0x00010001u : cnt;
	if (d == 0)
		w.tile_mx[(long)plane * w.NTP + tile] = 0u;
}

// done_levels: ring levels whose tiles got their histograms from the forward transform (lift.hip): skipped here
__global__ __launch_bounds__(256) void k_hist(PackGeom g, const int *__restrict__ lin, Work w, unsigned done_levels)
{
	const int lane = threadIdx.x & 63;
	const int tile0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * HIST_TPW;
	const int plane = blockIdx.y;
	if (tile0 >= w.NT)
		return;
	if (done_levels) {   // (a wave's two tiles lie on one level, or the first is the last of its level: the pair is split then)
		int l = 0;
		while (l + 1 < g.levels && tile0 >= g.tile_first[l + 1])
			++l;
		const bool d0 = (done_levels >> l) & 1u;
		const bool same = tile0 + 1 < g.tile_first[l + 1];
		const bool d1 = same ? d0 : (l + 1 < g.levels ? ((done_levels >> (l + 1)) & 1u) != 0 : true);
		if (d0 && d1)
			return;
		if (d0 != d1) {   // one tile of the pair: counted alone
			const int t = d0 ? tile0 + 1 : tile0;
			if (t < w.NT) {
				int val1[16], nv1;
				unsigned ok1;
				hist_load(g, lin, plane, t, lane, val1, ok1, nv1);
				hist_finish(w, plane, t, lane, val1, ok1, nv1);
			}
			return;
		}
	}
	const int hist_tiles = w.NT;
	int val[HIST_TPW][16], nvalid[HIST_TPW];
	unsigned ok[HIST_TPW];
#pragma unroll
	for (int u = 0; u < HIST_TPW; ++u)
		if (tile0 + u < hist_tiles)   // uniform
			hist_load(g, lin, plane, tile0 + u, lane, val[u], ok[u], nvalid[u]);
#pragma unroll
	for (int u = 0; u < HIST_TPW; ++u)
		if (tile0 + u < hist_tiles)
			hist_finish(w, plane, tile0 + u, lane, val[u], ok[u], nvalid[u]);
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
		if (rc_cap <= 0)   // no limit: nothing is ever refused, rc_count() only needs the sum
			return true;
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

__global__ __launch_bounds__(1024) void k_plan(PackGeom g, const int *__restrict__ lin, Work w, unsigned *out, long out_words, long capacity)
{
	const int img = blockIdx.x;
	__shared__ int top_of[3];
	if (threadIdx.x < 3)
		top_of[threadIdx.x] = 0;
	__syncthreads();
	for (int c = 0; c < g.C; ++c) {   // the plane's bit-plane count from the OR of all its magnitudes (one word per tile, rows padded to 4 and zero there)
		const uint4 *tt = reinterpret_cast<const uint4 *>(w.tile_mx + (long)(img * g.C + c) * w.NTP);
		unsigned m = 0;
		for (int q = threadIdx.x; q * 4 < w.NT; q += blockDim.x) {
			const uint4 v = tt[q];
			m |= v.x | (q * 4 + 1 < w.NT ? v.y : 0u) | (q * 4 + 2 < w.NT ? v.z : 0u) | (q * 4 + 3 < w.NT ? v.w : 0u);
		}
		int top = m ? ilog2u(m) + 1 : 0;
		for (int o = 32; o; o >>= 1)
			top = max(top, __shfl_xor(top, o));
		if ((threadIdx.x & 63) == 0)
			atomicMax(&top_of[c], top);
	}
	// the rest is one thread's work (a few thousand dependent steps): what it reads comes to LDS first (its own
	// loads would each wait out the memory latency, having stores in between), what it would clear is cleared here
	constexpr int ROOT_MAX = 256;   // root images are at most 15x15 (the last level leaves 8..15 per side)
	__shared__ int root[3 * ROOT_MAX];
	const bool root_lds = g.pixels[0] <= ROOT_MAX;
	if (root_lds)
		for (int i = threadIdx.x; i < g.C * g.pixels[0]; i += blockDim.x) {
			const int c = i / g.pixels[0];
			root[c * ROOT_MAX + i - c * g.pixels[0]] = lin[(long)(img * g.C + c) * g.total + i - c * g.pixels[0]];
		}
	for (int i = threadIdx.x; i < 48 * MAX_PLANES; i += blockDim.x)
		w.segidx[(long)img * 48 * MAX_PLANES + i] = 0;
	__syncthreads();
	if (threadIdx.x)
		return;
	ImgInfo &I = w.info[img];
	int planes[3] = { 0, 0, 0 };
	int pmax = 0;
	for (int c = 0; c < g.C; ++c) {
		planes[c] = top_of[c];
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
		const int *r = root_lds ? root + c * ROOT_MAX : lin + (long)(img * g.C + c) * g.total;
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
	const int waves = (int)blockDim.x >> 6;
	for (int k = 0; k < waves; ++k) {
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
	__shared__ int eb[MAX_SEGS + 1];   // the entry -> segment search reads it seven times in a row
	for (int k = threadIdx.x; k <= K; k += ENT_BLOCK)
		eb[k] = w.seg_ebase[(long)img * (MAX_SEGS + 1) + k];
	__syncthreads();
	const int e = blockIdx.x * ENT_BLOCK + threadIdx.x;
	unsigned nt = 0, nr = 0;
	if (e < E) {
		const int k = seg_of_entry(eb, K, e);
		int c, l, p;
		seg_unpack(sd[k], c, l, p);
		const int j = e - eb[k];
		const int ntile = g.tile_first[l + 1] - g.tile_first[l];
		const int cnt = g.tile_cnt[g.tile_first[l] + j];
		const unsigned short *cum = w.cum + ((long)(img * g.C + c) * w.NT + g.tile_first[l] + j) * NCUM;
		const int z = cum[p], upto = cum[p + 1];
		w.ent_seg[img * w.ES + e] = (unsigned short)k;
		w.ent_zeros[img * w.ES + e] = (unsigned short)z;
		w.ent_ones[img * w.ES + e] = (unsigned short)(upto - z);
		w.ent_refs[img * w.ES + e] = (unsigned short)(cnt - upto);
		nt = (unsigned)(upto - z) + (j == ntile - 1 ? 1u : 0u);   // + the segment's break slot
		nr = (unsigned)(cnt - upto);
	}
	unsigned tt, rt, wt;
	const unsigned tb = block_excl_scan(nt, wsum, tt);
	const unsigned rb = block_excl_scan(nr, wsum, rt);
	const unsigned wb = block_excl_scan((nr + 31u) >> 5, wsum, wt);
	if (e < E) {
		w.ent_tokbase[img * (w.ES + 1) + e] = tb;
		w.ent_refscum[img * (w.ES + 1) + e] = rb;
		w.ent_refw[img * (w.ES + 1) + e] = wb;
	}
	if (threadIdx.x == 0) {
		w.ent_blk[(img * w.NCB + blockIdx.x) * 3] = tt;
		w.ent_blk[(img * w.NCB + blockIdx.x) * 3 + 1] = rt;
		w.ent_blk[(img * w.NCB + blockIdx.x) * 3 + 2] = wt;
	}
}

__global__ __launch_bounds__(ENT_BLOCK) void k_entries_blocks(Work w)
{
	__shared__ unsigned wsum[16];
	const int img = blockIdx.x;
	ImgInfo &I = w.info[img];
	const int E = I.E;
	const int nb = (E + ENT_BLOCK - 1) / ENT_BLOCK;
	unsigned tok_run = 0, ref_run = 0, word_run = 0;
	for (int b0 = 0; b0 < nb; b0 += ENT_BLOCK) {
		const int b = b0 + threadIdx.x;
		unsigned *slot = w.ent_blk + (img * w.NCB + b) * 3;
		const unsigned t = b < nb ? slot[0] : 0u, r = b < nb ? slot[1] : 0u, wd = b < nb ? slot[2] : 0u;
		unsigned tt, rt, wt;
		const unsigned tb = block_excl_scan(t, wsum, tt);
		const unsigned rb = block_excl_scan(r, wsum, rt);
		const unsigned wb = block_excl_scan(wd, wsum, wt);
		if (b < nb) {
			slot[0] = tok_run + tb;
			slot[1] = ref_run + rb;
			slot[2] = word_run + wb;
		}
		tok_run += tt;
		ref_run += rt;
		word_run += wt;
	}
	if (threadIdx.x == 0) {
		w.ent_tokbase[img * (w.ES + 1) + E] = tok_run;
		w.ent_refscum[img * (w.ES + 1) + E] = ref_run;
		w.ent_refw[img * (w.ES + 1) + E] = word_run;
		I.T = tok_run + 1;   // + final flush (encode.c:221)
	}
}

__global__ __launch_bounds__(ENT_BLOCK) void k_entries_finish(Work w)
{
	const int img = blockIdx.y;
	const int e = blockIdx.x * ENT_BLOCK + threadIdx.x;
	if (e >= w.info[img].E)
		return;
	const unsigned *slot = w.ent_blk + (img * w.NCB + blockIdx.x) * 3;
	w.ent_tokbase[img * (w.ES + 1) + e] += slot[0];
	w.ent_refscum[img * (w.ES + 1) + e] += slot[1];
	w.ent_refw[img * (w.ES + 1) + e] += slot[2];
}

__global__ __launch_bounds__(ENT_BLOCK) void k_entries_segs(Work w)
{
	const int img = blockIdx.x;
	const int K = w.info[img].K;
	const int *eb = w.seg_ebase + (long)img * (MAX_SEGS + 1);
	const unsigned *tokbase = w.ent_tokbase + img * (w.ES + 1), *refscum = w.ent_refscum + img * (w.ES + 1);
	static_assert(MAX_SEGS <= ENT_BLOCK, "one thread per segment");
	const int k = threadIdx.x;
	if (k < K) {
		const unsigned refs = refscum[eb[k + 1]] - refscum[eb[k]];
		w.seg_refs[(long)img * MAX_SEGS + k] = refs;
		const int last = eb[k + 1] - 1;
		w.brk_tok[(long)img * MAX_SEGS + k] = tokbase[last] + w.ent_ones[img * w.ES + last];
	}
}

// ------------------------------------------------------------------- k_cut ---
// encode.c:192,204,216: the reference leaves its plane loop at the first byte the sink refuses (bytes.h:75-78), so
// what follows that point in the schedule is never coded.  Here the stream's bits are only known at the end
// (k_bitscan), but a lower bound is known now: a one costs at least two bits (the bit that ends its VLI,
// vli.h:67-84, and its sign), a refinement bit costs itself.  A segment whose first bit — by that bound — lies at or
// beyond bit 8 * (CAPACITY + 1) cannot reach the output: it and everything after it is dropped from the work of
// k_code, the carry scan, the order pass and the emitter (K, E, T shrink; `live` tells k_code which planes of a
// (channel, level) are still coded: tiles with none are not even loaded).  The exact clip stays in k_bitscan; the
// bytes are the same prefix of the unlimited stream.
__global__ __launch_bounds__(ENT_BLOCK) void k_cut(Work w, long capacity, int cut_off)
{
	__shared__ unsigned long long need[MAX_SEGS];
	__shared__ int kc_sh;
	const int img = blockIdx.x;
	ImgInfo &I = w.info[img];
	const int K = I.K;
	const int *eb = w.seg_ebase + (long)img * (MAX_SEGS + 1);
	const unsigned *tokbase = w.ent_tokbase + img * (w.ES + 1);
	const int k = threadIdx.x;
	if (k < K) {
		const unsigned ones = tokbase[eb[k + 1]] - tokbase[eb[k]] - 1u;   // (the segment's break slot is not a one)
		need[k] = 2ull * ones + w.seg_refs[(long)img * MAX_SEGS + k];
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		int kc = K;
		if (capacity > 0 && !cut_off) {
			const unsigned long long limit = 8ull * ((unsigned long long)capacity + 1ull);
			unsigned long long at = I.hdr_bits;
			for (int j = 0; j < K; ++j) {
				if (at >= limit) {
					kc = j;
					break;
				}
				at += need[j];
			}
		}
		kc_sh = kc;
		I.cut = (unsigned)(K - kc);
		I.K = kc;
		I.E = eb[kc];
		I.T = tokbase[eb[kc]] + 1u;   // + the final flush's slot (encode.c:221; beyond the clip whenever something was cut)
	}
	__syncthreads();
	const int kc = kc_sh;
	if (threadIdx.x < 48) {
		const int *sx = w.segidx + ((long)img * 48 + threadIdx.x) * MAX_PLANES;
		unsigned m = 0;
		for (int p = 0; p < MAX_PLANES; ++p)
			m |= (sx[p] && sx[p] <= kc ? 1u : 0u) << p;
		w.live[(long)img * 48 + threadIdx.x] = m;
	}
}

// ------------------------------------------------------------------ k_code ---
// One wave per 1024-coefficient tile; lane L owns coefficients 16L .. 16L+15 of the tile, so everything
// that is sequential in the reference's scan order (encode.c:60-95) is sequential inside a lane and a
// prefix over lanes.  Let t = number of magnitude bits of a coefficient (0 for zero).  At plane p a
// coefficient is a zero symbol if t <= p, the one of that plane if t == p+1, a refinement bit if t >= p+2.
// With Z[q] = #(coefficients before me with t <= q):
//   zeros before me at my plane p = t-1:      Z[t-1]
//   my rank among the ones of plane p:        Z[t] - Z[t-1]
//   refinement bits before me at plane p:     (#coefficients before me) - Z[p+1]
// Z = (prefix over the lanes before me) + (count inside my lane).  The lane part is kept for all q at once
// as 16 nibbles of one 64-bit register (adding 0x1111.. << 4t per coefficient); the prefix over lanes is one
// DPP scan of the same counts widened to 16-bit fields.  Tokens of plane p take the tile's slots
// [#(t >= p+2), #(t >= p+1)): plane-major, in coefficient order — the order of the stream.

// The class table is laid out class by class, the 64 lanes of a class side by side: whatever class a lane asks for, its
// bank is its lane's — look-ups never conflict, and nothing is padded (4 KB per wave: six workgroups fit a CU).
__device__ __forceinline__ int tab8(int lane, int cls) { return cls * 128 + lane * 2; }    // up to 8 planes: two dwords per entry
__device__ __forceinline__ int tab16(int lane, int cls) { return cls * 64 + lane; }        // more: one
constexpr int ROWW = 34;             // words per staging row of one plane (31 + 1024 bits + slack)
// Token slots of a tile: its planes' tokens one plane after the other, highest plane first — each plane's first slot
// moved up by at most 7 so that it sits like the plane's first token in memory modulo 8 (the tokens then leave as
// 16-byte pieces, LDS and memory aligned alike) — plus 8 dummy slots for the zero coefficients' writes.
constexpr int ZS_DUMMY = TILE + 8 * MAX_PLANES, ZS_SLOTS = ZS_DUMMY + 8;

struct alignas(16) CodeLds {
	unsigned tab[1024];              // up to 8 planes: [t-1][lane][2] = { Z[t-1],  first slot of plane t-1 + Z[t] - Z[t-1] };
	                                 // more: [t-1][lane] = Z[t-1] | (Z[t]-Z[t-1]) << 10 | (first slot of plane t-1) << 20
	union {                          // (the token slots have left for memory before the refinement rows are gathered)
		unsigned short zs[ZS_SLOTS];     // token slots: zeros before (10 bits) | sign << 12, then turned into tokens in place
		unsigned rows[(MAX_PLANES - 1) * ROWW];
	};
	unsigned short cum[MAX_PLANES + 2];   // the tile's histogram: #(t <= q), q = 0..16
	unsigned short slot0[MAX_PLANES];     // plane p's first token slot
	unsigned gb[MAX_PLANES];         // (16-plane variant) plane p: token index of the entry's first token; ~0: plane not coded
	unsigned long long rb[MAX_PLANES];    // plane p: bit position of the entry's refinement bits in the staging buffer
	int ent[MAX_PLANES];             // plane p: entry index, -1 = the plane is not coded for this tile
};

// NQ = number of count fields q = 0 .. NQ-1 (one nibble each): 8 when the channel has at most 8 bit planes (the
// counts then live in 32-bit registers), else 16.  Z[NQ] needs no field: it counts every coefficient.
template <int NQ>
struct Nib;
template <>
struct Nib<8> {
	typedef unsigned T;
	static constexpr T ONES = 0x11111111u, M0F = 0x0f0f0f0fu;
};
template <>
struct Nib<16> {
	typedef unsigned long long T;
	static constexpr T ONES = 0x1111111111111111ull, M0F = 0x0f0f0f0f0f0f0f0full;
};

// Bit transpose of eight bytes (lo = bytes 0..3, hi = bytes 4..7): afterwards byte p holds bit p of the eight inputs,
// input j at bit j.
__device__ __forceinline__ void transpose8(unsigned &lo, unsigned &hi)
{
	unsigned t;
	t = (lo ^ (lo >> 7)) & 0x00AA00AAu;
	lo ^= t ^ (t << 7);
	t = (hi ^ (hi >> 7)) & 0x00AA00AAu;
	hi ^= t ^ (t << 7);
	t = (lo ^ (lo >> 14)) & 0x0000CCCCu;
	lo ^= t ^ (t << 14);
	t = (hi ^ (hi >> 14)) & 0x0000CCCCu;
	hi ^= t ^ (t << 14);
	const unsigned nlo = (lo & 0x0F0F0F0Fu) | ((hi << 4) & 0xF0F0F0F0u);
	hi = ((lo >> 4) & 0x0F0F0F0Fu) | (hi & 0xF0F0F0F0u);
	lo = nlo;
}

// PEXT4[mask << 4 | bits]: the bits that stand on the set positions of a 4-bit mask, pushed together (entry i is
// computed by thread i of the workgroup).  256 bytes: every LDS bank holds one word of it, look-ups never conflict.
__device__ __forceinline__ unsigned pext4_entry(unsigned i)
{
	const unsigned m = i >> 4, v = i & 15u;
	unsigned out = 0, k = 0;
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		out |= (((v & m) >> j) & 1u) << k;
		k += (m >> j) & 1u;
	}
	return out;
}

typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));

// two token slots at once: zero count minus the predecessor's (low 10 bits of each half), the sign bit (12) stays
__device__ __forceinline__ unsigned slots_to_tokens(unsigned a, unsigned before)
{
	const unsigned b = __builtin_amdgcn_alignbit(a, before, 16);   // the slots one place earlier
	const ushort2_t d = __builtin_bit_cast(ushort2_t, a & 0x03ff03ffu) - __builtin_bit_cast(ushort2_t, b & 0x03ff03ffu);
	return (a & 0x10001000u) | __builtin_bit_cast(unsigned, d);
}

template <int NQ, bool FULL>
__device__ __forceinline__ void code_tile(CodeLds &L, const unsigned char *pext4, const int (&val)[16], const Work &w, int img, int lane, int nvalid,
	int nv, int vb, int Pc, int tile_top, unsigned my_tokbase, int my_ent, unsigned long long my_rb)
{
	// Planes at or above the tile's own bit-plane count hold nothing but zeros here: all per-class work stops at P
	const int P = Pc < tile_top ? Pc : tile_top;
	typedef typename Nib<NQ>::T R_t;
	constexpr R_t ONES = Nib<NQ>::ONES, M0F = Nib<NQ>::M0F;
	constexpr int NB = NQ / 4;   // dwords of 16-bit fields per parity

	// ---- pass A: magnitudes, signs, per-lane counts of (t <= q) as nibbles (two halves: a nibble holds up to 8) ----
	unsigned mag[16];
	unsigned sgn = 0;
	R_t Ra = 0, Rb = 0;
#pragma unroll
	for (int i = 0; i < 16; ++i) {
		const int v = val[i];
		const unsigned a = (unsigned)(v < 0 ? -v : v);
		mag[i] = a;   // 0 for the coefficients past the ring's end
		sgn |= ((unsigned)v >> 31) << i;
		int t = 32 - __clz((int)a);
		t = t < NQ ? t : NQ;
		const int te = FULL || i < nv ? t : NQ;   // past the end: counted nowhere
		const R_t m = te < NQ ? ONES << (4 * te) : (R_t)0;
		if (i < 8)
			Ra += m;
		else
			Rb += m;
	}
	// bytes: ev byte b = #(t <= 2b), od byte b = #(t <= 2b+1) inside this lane (up to 16)
	const R_t ev = (Ra & M0F) + (Rb & M0F);
	const R_t od = ((Ra >> 4) & M0F) + ((Rb >> 4) & M0F);
	// 16-bit fields for the scan over lanes: E[b] = #(t <= 4b) | #(t <= 4b+2) << 16, O[b] = #(t <= 4b+1) | #(t <= 4b+3) << 16
	unsigned E[NB], O[NB], cE[NB], cO[NB];
	const int nq = (P >> 2) + 1 < NB ? (P >> 2) + 1 : NB;   // groups of four q that matter (q <= P); uniform
#pragma unroll
	for (int b = 0; b < NB; ++b) {
		const unsigned xe = (unsigned)(ev >> (16 * b)) & 0xffffu, xo = (unsigned)(od >> (16 * b)) & 0xffffu;
		const unsigned e0 = (xe & 0xffu) | ((xe & 0xff00u) << 8), o0 = (xo & 0xffu) | ((xo & 0xff00u) << 8);
		unsigned ei = 0, oi = 0;
		if (b < nq) {
			ei = wave_incl_add(e0);
			oi = wave_incl_add(o0);
		}
		cE[b] = (unsigned)__builtin_amdgcn_readlane((int)ei, 63);   // the tile's totals
		cO[b] = (unsigned)__builtin_amdgcn_readlane((int)oi, 63);
		E[b] = ei - e0;   // exclusive: the lanes before this one
		O[b] = oi - o0;
	}
	// Z over the lanes before me / the whole tile, q a compile-time constant after unrolling
#define ZL(q) ((((q) & 1 ? O[((q) >> 2) % NB] : E[((q) >> 2) % NB]) >> (16 * (((q) >> 1) & 1))) & 0xffffu)
#define CT(q) ((((q) & 1 ? cO[((q) >> 2) % NB] : cE[((q) >> 2) % NB]) >> (16 * (((q) >> 1) & 1))) & 0xffffu)
	// every plane's first slot (uniform; planes descending = slots ascending)
	unsigned first[NQ];
	{
		unsigned cur = 0;
#pragma unroll
		for (int p = NQ - 1; p >= 0; --p) {
			const unsigned c1 = p + 1 < P && p + 1 < NQ ? CT(p + 1) : (unsigned)nvalid, c0 = p < P ? CT(p) : (unsigned)nvalid;
			const unsigned tb = (unsigned)__builtin_amdgcn_readlane((int)my_tokbase, p);
			first[p] = cur + ((tb - cur) & 7u);
			cur = first[p] + (c1 - c0);
		}
	}
#pragma unroll
	for (int t = 1; t <= NQ; ++t) {
		if (t <= P) {
			const unsigned zlo = ZL(t - 1);
			const unsigned zhi = t < NQ && t < P ? ZL(t) : (unsigned)vb;   // (no coefficient of this tile has more than P bits)
			if (NQ == 8)
				*reinterpret_cast<uint2 *>(&L.tab[tab8(lane, t - 1)]) = make_uint2(zlo, first[t - 1] + zhi - zlo);
			else
				L.tab[tab16(lane, t - 1)] = zlo | (zhi - zlo) << 10 | first[t - 1] << 20;
		}
	}
	if (lane == 0) {
#pragma unroll
		for (int q = 0; q <= MAX_PLANES; ++q)
			L.cum[q] = (unsigned short)(q < P && q < NQ ? CT(q) : (unsigned)nvalid);
#pragma unroll
		for (int q = 0; q < MAX_PLANES; ++q)
			L.slot0[q] = (unsigned short)(q < NQ ? first[q < NQ ? q : 0] : 0u);
	}
	wave_sync();

	// per-plane facts of the tile for the plane loops below (lane p holds plane p's)
	if (lane < MAX_PLANES) {
		L.gb[lane] = my_ent >= 0 ? my_tokbase : ~0u;
		L.ent[lane] = my_ent;
		L.rb[lane] = my_rb;
	}
	wave_sync();

	auto cq = [&](int q) -> unsigned { return q < P && q < NQ ? CT(q) : (unsigned)nvalid; };   // q a constant

	// The refinement bits go first: their last step adds the rows' edge words to the staging buffer with global atomics
	// (the neighbouring tiles share those words), which take long to come back — issued here, they have the
	// whole token stage to do so; issued last they held the finished wave's slot (a quarter of this kernel's time).
	// ---- pass C: refinement bits (encode.c:84-93), per plane a <= 16-bit string per lane at rank (coefficients before) - Z[p+1] ----
	wave_sync();
	for (int i = lane; i < (NQ - 1) * ROWW; i += 64)
		L.rows[i] = 0u;
	wave_sync();
	// the lane's string of plane p at its place in the plane's row: zl = Z[p+1] of the lanes before, bit0 = where the entry's bits start
	auto deposit = [&](int p, unsigned acc, unsigned cnt, unsigned zl, unsigned bit0) {
		const unsigned pos = (bit0 & 31u) + ((unsigned)vb - zl);
		if (cnt) {
			unsigned *row = L.rows + p * ROWW;
			const unsigned sh = pos & 31u;
			atomicOr(&row[pos >> 5], acc << sh);
			if (sh + cnt > 32u)
				atomicOr(&row[(pos >> 5) + 1], acc >> (32u - sh));
		}
	};
	unsigned *stage = w.stage + img * w.SW;
	// a plane's row to the staging buffer: the entry's own words (every entry starts on a word there: no word is
	// shared, nothing is added atomically — a quarter of this kernel's time went into those two atomics per plane)
	auto row_out = [&](int p, int refs, unsigned long long bit0) {
		const int nw = (refs + 31) >> 5;
		if (lane < nw)
			stage[(bit0 >> 5) + lane] = L.rows[p * ROWW + lane];
	};
	if (NQ == 8) {
		// Magnitudes below 256: the 16 of them as bytes, bit-transposed, are the lane's sixteen bits of every plane
		// at once (byte p of A: coefficients 0..7, of B: 8..15).  A coefficient takes part in plane p's refinement
		// pass if a higher plane has a bit of it: the OR of the bytes above.  The string is the plane's bits on those
		// positions pushed together, a nibble per table look-up (PEXT4).
		unsigned A0 = mag[0] | mag[1] << 8 | mag[2] << 16 | mag[3] << 24, A1 = mag[4] | mag[5] << 8 | mag[6] << 16 | mag[7] << 24;
		unsigned B0 = mag[8] | mag[9] << 8 | mag[10] << 16 | mag[11] << 24, B1 = mag[12] | mag[13] << 8 | mag[14] << 16 | mag[15] << 24;
		transpose8(A0, A1);
		transpose8(B0, B1);
		const unsigned SA1 = (A1 >> 8) | (A1 >> 16) | (A1 >> 24), SB1 = (B1 >> 8) | (B1 >> 16) | (B1 >> 24);
		const unsigned SA0 = (A0 >> 8) | (A0 >> 16) | (A0 >> 24) | ((SA1 | A1) & 0xffu) * 0x01010101u;
		const unsigned SB0 = (B0 >> 8) | (B0 >> 16) | (B0 >> 24) | ((SB1 | B1) & 0xffu) * 0x01010101u;
		auto plane_string = [&](int p, unsigned Aw, unsigned SAw, unsigned Bw, unsigned SBw) {
			const int refs = nvalid - (int)L.cum[p + 1];
			if (refs <= 0 || L.ent[p] < 0)
				return;   // uniform
			const unsigned sh8 = 8u * ((unsigned)p & 3u);
			const unsigned nA = (Aw >> sh8) & 0xffu, sA = (SAw >> sh8) & 0xffu, nB = (Bw >> sh8) & 0xffu, sB = (SBw >> sh8) & 0xffu;
			const unsigned e0 = pext4[((sA & 15u) << 4) | (nA & 15u)], e1 = pext4[(sA & 0xf0u) | (nA >> 4)];
			const unsigned e2 = pext4[((sB & 15u) << 4) | (nB & 15u)], e3 = pext4[(sB & 0xf0u) | (nB >> 4)];
			const unsigned c0 = (unsigned)__builtin_popcount(sA & 15u), c1 = (unsigned)__builtin_popcount(sA);
			const unsigned c2 = c1 + (unsigned)__builtin_popcount(sB & 15u), c3 = c1 + (unsigned)__builtin_popcount(sB);
			deposit(p, e0 | e1 << c0 | e2 << c1 | e3 << c2, c3, L.tab[tab8(lane, p + 1)] & 0x3ffu, (unsigned)L.rb[p]);
		};
		for (int p = P - 2; p >= 4; --p)
			plane_string(p, A1, SA1, B1, SB1);
		for (int p = P - 2 < 3 ? P - 2 : 3; p >= 0; --p)
			plane_string(p, A0, SA0, B0, SB0);
		wave_sync();
		for (int p = P - 2; p >= 0; --p) {
			const int refs = nvalid - (int)L.cum[p + 1];
			if (refs <= 0 || L.ent[p] < 0)
				continue;
			row_out(p, refs, L.rb[p]);
		}
	} else {
		for (int p = P - 2; p >= 0; --p) {
			const int refs = nvalid - (int)L.cum[p + 1];
			if (refs <= 0 || L.ent[p] < 0)
				continue;   // uniform
			const unsigned thr = 2u << p;
			unsigned acc = 0, cnt = 0;
#pragma unroll
			for (int i = 0; i < 16; ++i) {
				const bool isref = mag[i] >= thr;
				acc |= (isref ? (mag[i] >> p) & 1u : 0u) << cnt;
				cnt += isref ? 1u : 0u;
			}
			deposit(p, acc, cnt, L.tab[tab16(lane, p + 1)] & 0x3ffu, (unsigned)L.rb[p]);
		}
		wave_sync();
		for (int p = P - 2; p >= 0; --p) {
			const int refs = nvalid - (int)L.cum[p + 1];
			if (refs <= 0 || L.ent[p] < 0)
				continue;
			row_out(p, refs, L.rb[p]);
		}
	}
	wave_sync();   // the rows have been read: the token slots take their place
	// ---- pass B: every non-zero coefficient drops its zero count into its token slot (zeros into a dummy slot) ----
	{
		R_t R = 0;
		constexpr int NB8 = NQ == 8 ? 8 : 2;   // look-ups in flight (registers: the 16-plane variant keeps 64-bit counts)
#pragma unroll
		for (int h = 0; h < 16 / NB8; ++h) {
			uint2 ent[NB8];
#pragma unroll
			for (int i = NB8 * h; i < NB8 * h + NB8; ++i) {   // a batch of table look-ups first: their latencies overlap
				const int t0 = min(32 - __clz((int)mag[i]), NQ);   // (the bit count again: sixteen registers less than keeping it)
				const int tm = t0 > 0 ? t0 - 1 : 0;
				if (NQ == 8) {
					ent[i % NB8] = *reinterpret_cast<const uint2 *>(&L.tab[tab8(lane, tm)]);
				} else {
					const unsigned e = L.tab[tab16(lane, tm)];
					ent[i % NB8] = make_uint2(e & 0x3ffu, (e >> 20) + ((e >> 10) & 0x3ffu));
				}
			}
#pragma unroll
			for (int i = NB8 * h; i < NB8 * h + NB8; ++i) {
				const int t = min(32 - __clz((int)mag[i]), NQ);
				const int tm = t > 0 ? t - 1 : 0;
				const unsigned x = (unsigned)(R >> (4 * tm));
				const unsigned lz_lo = x & 15u;
				const unsigned lz_hi = t < NQ ? (x >> 4) & 15u : (unsigned)i;
				const unsigned slot = mag[i] ? ent[i % NB8].y + lz_hi - lz_lo : (unsigned)(ZS_DUMMY + (lane & 7));
				L.zs[slot] = (unsigned short)(ent[i % NB8].x + lz_lo + ((sgn >> i) & 1u) * 0x1000u);
				const int te = FULL || i < nv ? t : NQ;
				R += te < NQ ? ONES << (4 * te) : (R_t)0;
			}
		}
	}
	wave_sync();
	// ---- tokens: run = zeros since the previous one of the same plane in this tile = a slot's zero count minus its
	//      predecessor's; the first slot of a plane keeps its count.  Eight slots per lane at a time, two per instruction,
	//      in place; then every plane's tokens — consecutive slots, consecutive in the stream — leave as they are ----
	{
		// (lanes 0..15 take a plane each) the plane's first slot as it is, and the zeros after its last one: what the
		// tile hands to the run counter (k_carry_*)
		unsigned first_tok = 0;
		int first_slot = -1;
		if (lane < MAX_PLANES) {
			const int p = lane;
			const unsigned c0 = L.cum[p], c1 = L.cum[p + 1];
			const unsigned ones = c1 - c0, slot0 = L.slot0[p];
			if (ones) {
				first_slot = (int)slot0;
				first_tok = L.zs[slot0];
			}
			if (my_ent >= 0)
				w.ent_tz[img * w.ES + my_ent] = (unsigned short)(ones ? c0 - (L.zs[slot0 + ones - 1] & 0x3ffu) : c0);
		}
		const int nslots = (int)(first[0] + (cq(1) - cq(0)));   // where the lowest plane's tokens end
		uint4 *zq = reinterpret_cast<uint4 *>(L.zs);
		// (a block's first lane needs the slot before the block as it was: those are read before any block is rewritten;
		// inside a block every lane has read before any lane writes — a wave's LDS operations execute in order)
		const int nblk = (nslots + 511) >> 9;   // uniform, <= 3
		unsigned before[3];
#pragma unroll
		for (int k = 0; k < 3; ++k)
			before[k] = k < nblk && 512 * k + 8 * lane > 0 && 512 * k + 8 * lane <= ZS_DUMMY ? (unsigned)L.zs[512 * k + 8 * lane - 1] << 16 : 0u;
		wave_sync();
#pragma unroll
		for (int k = 0; k < 3; ++k)
			if (k < nblk && 64 * k + lane < ZS_DUMMY / 8) {
				const uint4 a = zq[64 * k + lane];
				zq[64 * k + lane] = make_uint4(slots_to_tokens(a.x, before[k]), slots_to_tokens(a.y, a.x), slots_to_tokens(a.z, a.y), slots_to_tokens(a.w, a.z));
			}
		wave_sync();
		if (first_slot >= 0)
			L.zs[first_slot] = (unsigned short)first_tok;
		wave_sync();
		unsigned short *tok16 = w.tok16 + img * w.TS;
		// A plane's tokens to memory: slot and token sit alike modulo 8, so everything between the first and the last
		// 16-byte boundary leaves as 16-byte pieces (2-byte stores, 64 to an instruction, were what this kernel
		// waited for most); the up to seven tokens before and after go one by one, in one instruction.
		auto plane_out = [&](unsigned tb, unsigned slot0, int ones) {
			unsigned short *dst = tok16 + tb;
			const unsigned short *src = L.zs + slot0;
			const int head = min(ones, (int)((8u - (tb & 7u)) & 7u));
			const int chunks = (ones - head) >> 3, tail0 = head + 8 * chunks;
			const int e = lane < 8 ? lane : tail0 + lane - 8;   // lanes 0..7: the head, lanes 8..15: the tail
			if (lane < 8 ? lane < head : (lane < 16 && e < ones))
				dst[e] = src[e];
			for (int c = lane; c < chunks; c += 64)
				*reinterpret_cast<uint4 *>(dst + head + 8 * c) = *reinterpret_cast<const uint4 *>(src + head + 8 * c);
		};
		for (int p = P > 0 ? P - 1 : 0; p >= 0; --p) {
			const unsigned c0 = (unsigned)__builtin_amdgcn_readfirstlane((int)L.cum[p]), c1 = (unsigned)__builtin_amdgcn_readfirstlane((int)L.cum[p + 1]);
			const int ones = (int)(c1 - c0);
			const unsigned gb = (unsigned)__builtin_amdgcn_readfirstlane((int)L.gb[p]);
			if (ones <= 0 || gb == ~0u)
				continue;   // uniform
			plane_out(gb, (unsigned)__builtin_amdgcn_readfirstlane((int)L.slot0[p]), ones);
		}
	}

#undef ZL
#undef CT
}

// Two kernels: planes of up to 8 bit planes — every 8-bit picture — take the variant with 32-bit count registers,
// which alone needs 94 vector registers where the two variants in one kernel needed 126: five waves per SIMD instead
// of four (its LDS fits five workgroups per CU since the refinement rows share the token slots' words).  The wide
// variant runs a small grid that strides over the tiles, so that launching it for nothing costs nothing.
template <bool WIDE>
__device__ __forceinline__ void code_one(const PackGeom &g, const int *__restrict__ lin, const Work &w, CodeLds &L, const unsigned char *pext4, int tile,
	int plane, int lane)
{
	const int img = plane / g.C, c = plane - img * g.C;
	const ImgInfo &I = w.info[img];
	{
	int l = 0;
	while (l + 1 < g.levels && tile >= g.tile_first[l + 1])
		++l;
	const int j = tile - g.tile_first[l];
	const int nvalid = g.tile_cnt[tile];
	const int P = I.planes[c] < MAX_PLANES ? I.planes[c] : MAX_PLANES;
	const unsigned live = w.live[(long)img * 48 + c * 16 + l];   // planes of this ring that are coded (k_cut)
	if (!live)
		return;   // (uniform) everything this tile could add lies beyond CAPACITY
	// Coded planes at or above the tile's own bit-plane count see nothing but zeros: no tokens, no refinement bits, the
	// tile just hands its coefficients on to the run counter (k_carry_*).  When that is all there is — the finest
	// rings' high planes under a CAPACITY that cut the low ones off — the coefficients are not even loaded.
	const unsigned tile_or = w.tile_mx[(long)plane * w.NTP + tile];
	const int tile_top = tile_or ? ilog2u(tile_or) + 1 : 0;
	if (!(live & ((1u << (tile_top < 31 ? tile_top : 31)) - 1u))) {   // uniform
		if (lane < MAX_PLANES && ((live >> lane) & 1u)) {
			const int k = w.segidx[((long)img * 48 + c * 16 + l) * MAX_PLANES + lane] - 1;
			w.ent_tz[img * w.ES + w.seg_ebase[(long)img * (MAX_SEGS + 1) + k] + j] = (unsigned short)nvalid;
		}
		return;
	}
	const int first = 16 * lane;
	const int nv = nvalid - first < 0 ? 0 : nvalid - first > 16 ? 16 : nvalid - first;   // this lane's coefficients
	const int vb = first < nvalid ? first : nvalid;                                      // coefficients in the lanes before

	// ---- the coefficients, once (the class table's LDS words stage a pyramid square meanwhile) ----
	static_assert(sizeof(L.tab) >= sizeof(unsigned) * SQ_WORDS, "the class table doubles as the square's staging area");
	int val[16];
	load_tile16(g, lin, plane, l, tile, lane, nvalid, nv, L.tab, val);
	// per-plane bookkeeping of this tile's entries (lanes 0..15 take a plane each): three dependent look-ups
	// whose results are only needed after the first passes over the coefficients — they stay in registers till then
	unsigned my_tokbase = 0;
	int my_ent = -1;
	unsigned long long my_rb = 0;
	if (lane < MAX_PLANES) {
		const int p = lane;
		const int k1 = (live >> p) & 1u ? w.segidx[((long)img * 48 + c * 16 + l) * MAX_PLANES + p] : 0;
		if (k1) {
			const int k = k1 - 1;
			const int e0 = w.seg_ebase[(long)img * (MAX_SEGS + 1) + k];
			my_ent = e0 + j;
			my_rb = (unsigned long long)w.ent_refw[img * (w.ES + 1) + my_ent] << 5;   // (every entry's bits start on a word of the staging buffer)
			my_tokbase = w.ent_tokbase[img * (w.ES + 1) + my_ent];
		}
	}
	if (!WIDE) {
		if (nvalid == TILE)
			code_tile<8, true>(L, pext4, val, w, img, lane, nvalid, nv, vb, P, tile_top, my_tokbase, my_ent, my_rb);
		else
			code_tile<8, false>(L, pext4, val, w, img, lane, nvalid, nv, vb, P, tile_top, my_tokbase, my_ent, my_rb);
	} else {
		code_tile<16, false>(L, pext4, val, w, img, lane, nvalid, nv, vb, P, tile_top, my_tokbase, my_ent, my_rb);
	}
	}
}

template <bool WIDE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5))) void k_code(PackGeom g, const int *__restrict__ lin, Work w)
{
	__shared__ CodeLds lds[4];
	__shared__ unsigned char pext4[256];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int plane = blockIdx.y;
	const int img = plane / g.C, c = plane - img * g.C;
	if ((w.info[img].planes[c] > 8) != WIDE)
		return;   // (uniform over the workgroup) the other kernel's plane
	if (!WIDE) {
		pext4[threadIdx.x] = (unsigned char)pext4_entry(threadIdx.x);
		__syncthreads();   // the only point where the workgroup's waves meet
		const int tile = blockIdx.x * 4 + wv;
		if (tile < w.NT)   // whole wave; nothing below synchronises across waves
			code_one<false>(g, lin, w, lds[wv], pext4, tile, plane, lane);
	} else {
		for (int tile = blockIdx.x * 4 + wv; tile < w.NT; tile += gridDim.x * 4) {
			code_one<true>(g, lin, w, lds[wv], pext4, tile, plane, lane);
			wave_sync();   // the wave's next tile reuses its LDS
		}
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

__device__ __forceinline__ void token_store(const Work &w, int img, unsigned t, unsigned run, unsigned flags)
{
	if (run >= T_ESC) {
		w.tok_big[img * w.TS + t] = run;
		run = T_ESC;
	}
	w.tok16[img * w.TS + t] = (unsigned short)(run | flags);
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
		const int k = w.ent_seg[img * w.ES + e];
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
	const int waves = (int)blockDim.x >> 6;
	for (int k = 0; k < waves; ++k) {
		const RunMap a = wagg[k];
		if (k < wv)
			pre = compose(pre, a);
		all = compose(all, a);
	}
	__syncthreads();
	total = all;
	return compose(pre, m);
}

// The blocks of 1024 entries are scanned by 256 threads with four consecutive entries each: a quarter of the waves
// and of the barriers for the same entries (the kernels wait on both, not on memory).
constexpr int CARRY_PER = 4, CARRY_THREADS = CARRY_BLOCK / CARRY_PER;

__global__ __launch_bounds__(CARRY_THREADS) void k_carry_local(Work w)
{
	__shared__ RunMap wagg[16];
	const int img = blockIdx.y;
	const ImgInfo &I = w.info[img];
	if ((int)blockIdx.x * CARRY_BLOCK >= I.E)
		return;
	const int e0 = blockIdx.x * CARRY_BLOCK + threadIdx.x * CARRY_PER;
	RunMap t = { 1u, 0u };
#pragma unroll
	for (int j = 0; j < CARRY_PER; ++j) {
		bool has_one, seg_end, refs;
		const RunMap m = carry_map_of(w, I, img, e0 + j, has_one, seg_end, refs);
		t = compose(t, refs ? RunMap{ 0u, 0u } : m);   // the break slot takes the pending run, the next segment starts from 0
	}
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
	if (threadIdx.x == 0)
		token_store(w, img, I.T - 1, s, T_NOSIGN);   // encode.c:221 rle_flush: always emitted
}

__global__ __launch_bounds__(CARRY_THREADS) void k_carry_apply(Work w)
{
	__shared__ RunMap wagg[16];
	const int img = blockIdx.y;
	const ImgInfo &I = w.info[img];
	if ((int)blockIdx.x * CARRY_BLOCK >= I.E)
		return;
	const int e0 = blockIdx.x * CARRY_BLOCK + threadIdx.x * CARRY_PER;
	RunMap own[CARRY_PER];
	bool has_one[CARRY_PER], seg_end[CARRY_PER], refs[CARRY_PER];
	RunMap t = { 1u, 0u };
#pragma unroll
	for (int j = 0; j < CARRY_PER; ++j) {
		own[j] = carry_map_of(w, I, img, e0 + j, has_one[j], seg_end[j], refs[j]);
		t = compose(t, refs[j] ? RunMap{ 0u, 0u } : own[j]);
	}
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
	const unsigned s_blk = w.carry_in[img * w.NCB + blockIdx.x];
	unsigned s_in = ex.add + (ex.keep ? s_blk : 0u);   // the pending run entering this thread's first entry
#pragma unroll
	for (int j = 0; j < CARRY_PER; ++j) {
		const int e = e0 + j;
		if (e >= I.E)
			break;
		const unsigned tb = w.ent_tokbase[img * (w.ES + 1) + e];
		if (has_one[j] && s_in) {   // the entry's first token: its run began before this tile
			const unsigned tk = w.tok16[img * w.TS + tb];
			token_store(w, img, tb, (tk & T_RUN) + s_in, tk & ~T_RUN);
		}
		const unsigned s_out = own[j].add + (own[j].keep ? s_in : 0u);
		if (seg_end[j])   // the break slot
			token_store(w, img, tb + w.ent_ones[img * w.ES + e], s_out, T_BREAK | T_NOSIGN | ((refs[j] && s_out) ? 0u : T_VOID));
		s_in = refs[j] ? 0u : s_out;
	}
}

// ---------------------------------------------------------- token walks ---
// The order pass and the emitter both give every lane 64 CONSECUTIVE tokens (a "group") of a wave's
// window of 4096.  The window is read as coalesced 16-byte pieces and laid out in LDS as one row per
// group (pitch 34 dwords: lane j's ds_read_b64 of its q-th token quad hits bank pair 34j + 2q, all
// different over 32 lanes).  Tokens outside the image's [0, T) become void.

constexpr int WROW = 34;
constexpr unsigned VOID2 = (T_VOID | T_NOSIGN) * 0x00010001u;

__device__ __forceinline__ void stage_tokens(unsigned *rows, const unsigned short *tok16, long t0, long T, int lane)
{
	if (t0 >= 0 && t0 + CHUNK <= T) {
		// the whole window holds tokens (all but an image's first and last window): eight loads that feed nothing but
		// their LDS stores — no select on what they return, so they are all on their way before the first is waited for
		uint4 v[8];
#pragma unroll
		for (int it = 0; it < 8; ++it)
			v[it] = *reinterpret_cast<const uint4 *>(tok16 + t0 + it * 512 + lane * 8);
#pragma unroll
		for (int it = 0; it < 8; ++it) {
			const int off = it * 512 + lane * 8;
			unsigned *dst = rows + (off >> 6) * WROW + ((off & 63) >> 1);
			*reinterpret_cast<uint2 *>(dst) = make_uint2(v[it].x, v[it].y);
			*reinterpret_cast<uint2 *>(dst + 2) = make_uint2(v[it].z, v[it].w);
		}
		return;
	}
#pragma unroll
	for (int it = 0; it < 8; ++it) {
		const int off = it * 512 + lane * 8;
		const long t = t0 + off;
		uint4 v = make_uint4(VOID2, VOID2, VOID2, VOID2);
		if (t >= 0 && t + 8 <= T) {
			v = *reinterpret_cast<const uint4 *>(tok16 + t);
		} else if (t + 8 > 0 && t < T) {
			unsigned h[8];
#pragma unroll
			for (int e = 0; e < 8; ++e)
				h[e] = t + e >= 0 && t + e < T ? (unsigned)tok16[t + e] : (T_VOID | T_NOSIGN);
			v = make_uint4(h[0] | h[1] << 16, h[2] | h[3] << 16, h[4] | h[5] << 16, h[6] | h[7] << 16);
		}
		unsigned *dst = rows + (off >> 6) * WROW + ((off & 63) >> 1);
		*reinterpret_cast<uint2 *>(dst) = make_uint2(v.x, v.y);
		*reinterpret_cast<uint2 *>(dst + 2) = make_uint2(v.z, v.w);
	}
}

// The same for one half of every group: tokens [32 * half, 32 * half + 32) of the 64 groups of the wave's stretch, one
// row of 16 dwords per group (pitch 18: lane j's ds_read_b64 of its q-th quad hits bank pair 18j + 2q, all different
// over 32 lanes).  k_emit walks its tokens once, front to back: staging them half by half halves its LDS.
constexpr int HROW = 18;

// the loads of one half (in registers: the second half's are asked for before the first half is walked) ...
struct HalfTokens {
	uint4 v[4];
};

__device__ __forceinline__ HalfTokens load_tokens_half(const unsigned short *tok16, long t0, long T, int lane, int half)
{
	HalfTokens o;
	if (t0 >= 0 && t0 + CHUNK <= T) {   // (uniform) the whole window holds tokens: plain loads, nothing selected from what they return
#pragma unroll
		for (int it = 0; it < 4; ++it)
			o.v[it] = *reinterpret_cast<const uint4 *>(tok16 + t0 + (it * 16 + (lane >> 2)) * 64 + half * 32 + (lane & 3) * 8);
		return o;
	}
#pragma unroll
	for (int it = 0; it < 4; ++it) {
		const int grp = it * 16 + (lane >> 2), piece = lane & 3;   // 16 bytes = 8 tokens per lane
		const long t = t0 + grp * 64 + half * 32 + piece * 8;
		uint4 v = make_uint4(VOID2, VOID2, VOID2, VOID2);
		if (t >= 0 && t + 8 <= T) {
			v = *reinterpret_cast<const uint4 *>(tok16 + t);
		} else if (t + 8 > 0 && t < T) {
			unsigned h[8];
#pragma unroll
			for (int e = 0; e < 8; ++e)
				h[e] = t + e >= 0 && t + e < T ? (unsigned)tok16[t + e] : (T_VOID | T_NOSIGN);
			v = make_uint4(h[0] | h[1] << 16, h[2] | h[3] << 16, h[4] | h[5] << 16, h[6] | h[7] << 16);
		}
		o.v[it] = v;
	}
	return o;
}

// ... and their places in the rows
__device__ __forceinline__ void deposit_tokens_half(unsigned *rows, const HalfTokens &h, int lane)
{
#pragma unroll
	for (int it = 0; it < 4; ++it) {
		const int grp = it * 16 + (lane >> 2), piece = lane & 3;
		unsigned *dst = rows + grp * HROW + piece * 4;
		*reinterpret_cast<uint2 *>(dst) = make_uint2(h.v[it].x, h.v[it].y);
		*reinterpret_cast<uint2 *>(dst + 2) = make_uint2(h.v[it].z, h.v[it].w);
	}
}

// a token pair holds an escape (run field 0xfff) / a break slot
__device__ __forceinline__ bool pair_has_esc(unsigned x) { return (((x & 0x0fff0fffu) + 0x00010001u) & 0x10001000u) != 0u; }
__device__ __forceinline__ bool pair_has_break(unsigned x) { return (x & (T_BREAK * 0x00010001u)) != 0u; }
// ... an escape or a void slot (the padding of an image's last group): the pairs that leave the plain path of the order walks
__device__ __forceinline__ bool pair_is_special(unsigned x)
{
	return ((((x & 0x0fff0fffu) + 0x00010001u) & 0x10001000u) | (x & (T_VOID * 0x00010001u))) != 0u;
}
// the order after a token of run v coded at order o (vli.h:67-84)
__device__ __forceinline__ int vli_after(int o, unsigned v) { return (int)__builtin_elementwise_sub_sat((unsigned)vli_top(o, v), 2u); }

__device__ __forceinline__ int vli_step(int o, unsigned v, bool skip)
{
	const int nx = vli_next(vli_top(o, v));
	return skip ? o : nx;
}

__device__ __forceinline__ unsigned token_run(unsigned tk, const unsigned *big, long t)
{
	const unsigned r = tk & T_RUN;
	return r == T_ESC ? big[t] : r;
}

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

// the order chain of one group's 64 tokens from order o (tokens of this lane's row; tb = index of its first token)
__device__ __forceinline__ int walk_order(const unsigned *my, const unsigned *big, long tb, int o)
{
#pragma unroll 4
	for (int q = 0; q < 16; ++q) {
		const uint2 x2 = *reinterpret_cast<const uint2 *>(my + 2 * q);
		const unsigned xs[2] = { x2.x, x2.y };
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			const unsigned x = xs[h];
			if (pair_is_special(x)) {
				o = vli_step(o, token_run(x & 0xffffu, big, tb + 4 * q + 2 * h), x & T_VOID);
				o = vli_step(o, token_run(x >> 16, big, tb + 4 * q + 2 * h + 1), (x >> 16) & T_VOID);
			} else {   // (no select: a void slot would keep the order)
				o = vli_after(o, x & T_RUN);
				o = vli_after(o, (x >> 16) & T_RUN);
			}
		}
	}
	return o;
}

// bits of one token pair x (run values v) coded from order o — o moves on — and of the refinement blocks that follow
// its break slots (t = index of the pair's first token)
__device__ __forceinline__ void pair_bits(unsigned x, const unsigned (&v)[2], long t, int &o, bool count_raw,
	const unsigned *btok, const unsigned *srefs, int K, unsigned &tokbits, unsigned long long &rawbits)
{
#pragma unroll
	for (int e = 0; e < 2; ++e) {
		const unsigned tk = e ? x >> 16 : x & 0xffffu;
		const int top = vli_top(o, v[e]);
		const unsigned nb = (unsigned)(2 * top - o + 1) + ((tk & T_NOSIGN) ? 0u : 1u);
		tokbits += (tk & T_VOID) ? 0u : nb;
		o = (tk & T_VOID) ? o : vli_next(top);
	}
	if (pair_has_break(x) && count_raw) {
#pragma unroll
		for (int e = 0; e < 2; ++e)
			if ((e ? x >> 16 : x) & T_BREAK)
				rawbits += srefs[find_break_seg(btok, K, (unsigned)(t + e))];
	}
}

// the same for a pair without escapes and void slots: no selects
__device__ __forceinline__ void pair_bits_plain(unsigned x, long t, int &o, bool count_raw,
	const unsigned *btok, const unsigned *srefs, int K, unsigned &tokbits, unsigned long long &rawbits)
{
#pragma unroll
	for (int e = 0; e < 2; ++e) {
		const unsigned tk = e ? x >> 16 : x & 0xffffu;
		const int top = vli_top(o, tk & T_RUN);
		tokbits += (unsigned)(2 * top - o + 2) - ((tk >> 15) & 1u);   // T_NOSIGN: one bit less
		o = (int)__builtin_elementwise_sub_sat((unsigned)top, 2u);
	}
	if (pair_has_break(x) && count_raw) {
#pragma unroll
		for (int e = 0; e < 2; ++e)
			if ((e ? x >> 16 : x) & T_BREAK)
				rawbits += srefs[find_break_seg(btok, K, (unsigned)(t + e))];
	}
}

// two chains at once (start orders 0 and 31): see k_gorder.  Once the chains of every lane have met
// (typically after 10-20 tokens) the rest of the group is walked with one chain — whose orders are the group's
// whatever it is entered with, so the bits of those tokens are counted on the way (tokbits, rawbits).  Returns the
// token quad the single chain started at (16: the chains did not meet): walk_bits only has to do the quads before it.
__device__ __forceinline__ int walk_order2(const unsigned *my, const unsigned *big, long tb, int &lo, int &hi, bool count_raw,
	const unsigned *btok, const unsigned *srefs, int K, unsigned &tokbits, unsigned long long &rawbits)
{
	tokbits = 0;
	rawbits = 0;
	int q = 0;
	bool met = false;   // uniform
	for (; q < 16; q += 2) {
		if (q && !ballot64(lo != hi)) {
			met = true;
			break;
		}
#pragma unroll
		for (int qq = 0; qq < 2; ++qq) {
			const uint2 x2 = *reinterpret_cast<const uint2 *>(my + 2 * (q + qq));
			const unsigned xs[2] = { x2.x, x2.y };
#pragma unroll
			for (int h = 0; h < 2; ++h) {
				const unsigned x = xs[h];
				if (pair_is_special(x)) {
					const unsigned v0 = token_run(x & 0xffffu, big, tb + 4 * (q + qq) + 2 * h);
					const unsigned v1 = token_run(x >> 16, big, tb + 4 * (q + qq) + 2 * h + 1);
					lo = vli_step(lo, v0, x & T_VOID);
					hi = vli_step(hi, v0, x & T_VOID);
					lo = vli_step(lo, v1, (x >> 16) & T_VOID);
					hi = vli_step(hi, v1, (x >> 16) & T_VOID);
				} else {
					lo = vli_after(vli_after(lo, x & T_RUN), (x >> 16) & T_RUN);
					hi = vli_after(vli_after(hi, x & T_RUN), (x >> 16) & T_RUN);
				}
			}
		}
	}
	const int qmet = q;
	for (; q < 16; ++q) {
		const uint2 x2 = *reinterpret_cast<const uint2 *>(my + 2 * q);
		const unsigned xs[2] = { x2.x, x2.y };
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			const unsigned x = xs[h];
			if (pair_is_special(x)) {
				const unsigned v[2] = { token_run(x & 0xffffu, big, tb + 4 * q + 2 * h), token_run(x >> 16, big, tb + 4 * q + 2 * h + 1) };
				pair_bits(x, v, tb + 4 * q + 2 * h, lo, count_raw, btok, srefs, K, tokbits, rawbits);
			} else {
				pair_bits_plain(x, tb + 4 * q + 2 * h, lo, count_raw, btok, srefs, K, tokbits, rawbits);
			}
		}
	}
	if (met)
		hi = lo;   // hi was left behind in the single-chain part
	return qmet;
}

// bits of a group's first 4 * qend tokens coded from order o, added to tokbits / rawbits
__device__ __forceinline__ void walk_bits(const unsigned *my, const unsigned *big, long tb, int o, int qend, bool count_raw,
	const unsigned *btok, const unsigned *srefs, int K, unsigned &tokbits, unsigned long long &rawbits)
{
#pragma unroll 2
	for (int q = 0; q < qend; ++q) {
		const uint2 x2 = *reinterpret_cast<const uint2 *>(my + 2 * q);
		const unsigned xs[2] = { x2.x, x2.y };
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			const unsigned x = xs[h];
			if (pair_is_special(x)) {
				const unsigned v[2] = { token_run(x & 0xffffu, big, tb + 4 * q + 2 * h), token_run(x >> 16, big, tb + 4 * q + 2 * h + 1) };
				pair_bits(x, v, tb + 4 * q + 2 * h, o, count_raw, btok, srefs, K, tokbits, rawbits);
			} else {
				pair_bits_plain(x, tb + 4 * q + 2 * h, o, count_raw, btok, srefs, K, tokbits, rawbits);
			}
		}
	}
}

// ---------------------------------------------------------------- k_gorder ---
// The order map of a token is monotone, so every start state ends between the chains started at 0
// and at 31; over 64 tokens those two almost always meet (orders decay by 2 per small value), and then
// the group's exit order is a constant, whatever it was entered with.  One lane per 64-token group:
// walk the 0- and the 31-chain; if they met, the NEXT group's entry order is known, and a second walk
// counts the group's bits.  Lane 0 only serves as the predecessor of lane 1 (63 groups of output per
// wave).  A group whose chains did not meet is resolved exactly as soon as its own entry order is
// known; only if that chain of knowledge breaks is the image flagged and the exact hierarchical pass
// (k_lut ... k_gorder_exact) redoes it.
constexpr int FSUBS = 63;

struct WalkLds {
	unsigned tok[64 * WROW];
};

// (two waves per workgroup: a wave's window is 8.7 KB of LDS, and 160 KB take nine workgroups of two — 18 waves per CU —
// but only four of four)
constexpr int GO_WAVES = 2;

__global__ __launch_bounds__(64 * GO_WAVES) void k_gorder(Work w)
{
	__shared__ WalkLds lds[GO_WAVES];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const long wave = (long)blockIdx.x * GO_WAVES + wv;
	const int img = blockIdx.y;
	const ImgInfo &I = w.info[img];
	const long T = I.T;
	const long nsub = (T + SUB - 1) / SUB;
	if (wave * FSUBS >= nsub)
		return;
	const unsigned short *tok16 = w.tok16 + img * w.TS;
	const unsigned *big = w.tok_big + img * w.TS;
	unsigned *rows = lds[wv].tok;
	const long S = wave * FSUBS - 1 + lane;          // this lane's group (lane 0: predecessor only)
	const long tfirst = (wave * FSUBS - 1) * SUB;    // first token of the wave's window (-64 for wave 0)
	stage_tokens(rows, tok16, tfirst, T, lane);
	wave_sync();
	const unsigned *my = rows + lane * WROW;
	const long tb = S * SUB;
	int lo = 0, hi = 31;
	const bool valid = S >= 0 && S < nsub;
	const bool produces = lane >= 1 && valid;
	const unsigned *btok = w.brk_tok + (long)img * MAX_SEGS, *srefs = w.seg_refs + (long)img * MAX_SEGS;
	unsigned tokbits;
	unsigned long long rawbits;
	const int qmet = walk_order2(my, big, tb, lo, hi, produces, btok, srefs, I.K, tokbits, rawbits);
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
		const int e = walk_order(my, big, tb, o);
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
	walk_bits(my, big, tb, o, qmet, produces, btok, srefs, I.K, tokbits, rawbits);   // the tokens before the chains met
	unsigned long long total;
	const unsigned long long pre = wave_excl_scan64(produces ? tokbits + rawbits : 0ull, total);
	if (produces) {
		w.lane_bits[img * w.NCS * 64 + S] = pre;
		w.grp_ord[img * w.NCS * 64 + S] = (unsigned char)o;
	}
	if (lane == 0)
		w.chunk_bits[img * w.NCS + wave] = total;
}

// ------------------------------------------------------------------- k_lut ---
// Exact pass (flagged images only).  Lanes 0..31 of each half-wave are the 32 possible VLI orders at
// the start of a 4096-token chunk; the half-wave walks the chunk once and every lane follows its own
// start state.  Snapshots at every 64-token boundary (sublut) give every group its entry order once
// the chunk's own entry is known.

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
	const unsigned short *tok16 = w.tok16 + img * w.TS;
	const unsigned *big = w.tok_big + img * w.TS;
	unsigned char *sub = w.sublut + (img * w.NCS + chunk) * 64 * 32;
	// lane i holds token i of a 64-token row for both chunks; bit 31 = "void" (runs are < 2^31).
	// Tokens are broadcast with v_readlane (no LDS traffic), each half picks its own chunk's.
	auto one = [&](long t) -> unsigned {
		if (t >= (long)T)
			return 0x80000000u;
		const unsigned tk = tok16[t];
		return token_run(tk, big, t) | ((tk & T_VOID) ? 0x80000000u : 0u);
	};
	auto fetch = [&](int q, unsigned &ra, unsigned &rb) {
		const long ta = chunk_a * CHUNK + q * SUB + lane;
		ra = one(ta);
		rb = one(ta + CHUNK);
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

// One lane per 64-token group, now with the true start order (chunk entry state -> sublut): the group's bits.
__device__ __forceinline__ void gorder_exact_body(const Work &w, unsigned *rows, long chunk, int img)
{
	const int lane = threadIdx.x & 63;
	const ImgInfo &I = w.info[img];
	const long T = I.T;
	const long nchunks = (T + CHUNK - 1) / CHUNK;
	if (chunk >= nchunks)
		return;
	const unsigned short *tok16 = w.tok16 + img * w.TS;
	const unsigned *big = w.tok_big + img * w.TS;
	wave_sync();   // the previous virtual block's reads of the rows are done
	stage_tokens(rows, tok16, chunk * CHUNK, T, lane);
	wave_sync();
	const long S = chunk * 64 + lane;
	const int o = w.sublut[((img * w.NCS + chunk) * 64 + lane) * 32 + w.chunk_entry[img * w.NCS + chunk]];
	const bool produces = S * SUB < T;
	unsigned tokbits = 0;
	unsigned long long rawbits = 0;
	walk_bits(rows + lane * WROW, big, S * SUB, o, 16, produces, w.brk_tok + (long)img * MAX_SEGS, w.seg_refs + (long)img * MAX_SEGS, I.K,
		tokbits, rawbits);
	unsigned long long total;
	const unsigned long long pre = wave_excl_scan64(produces ? tokbits + rawbits : 0ull, total);
	if (produces) {
		w.lane_bits[img * w.NCS * 64 + S] = pre;
		w.grp_ord[img * w.NCS * 64 + S] = (unsigned char)o;
	}
	if (lane == 0)
		w.chunk_bits[img * w.NCS + chunk] = total;
}

__global__ __launch_bounds__(256) void k_gorder_exact(Work w)
{
	__shared__ WalkLds lds[4];
	const int img = blockIdx.y;
	if (!w.slow[img])
		return;
	const long nvb = (w.NCS + 3) / 4;
	for (long vb = blockIdx.x; vb < nvb; vb += gridDim.x)
		gorder_exact_body(w, lds[threadIdx.x >> 6].tok, vb * 4 + (threadIdx.x >> 6), img);
}

// ------------------------------------------------------------------ k_emit ---
// bits.h:58-78.  One lane per 64-token group: entry order and bit position are known, so the lane walks
// its tokens once more, builds each code (vli.h:67-84: top-o zeros, a one, top remainder bits, then the
// sign of the one that ended the run) and appends it to a 64-bit accumulator; full 32-bit words go to
// an LDS window that covers the wave's stretch of the stream (OR: neighbouring lanes share their edge
// words), the window goes to memory as whole words, only the wave's first and last word with atomicOr.
// A break slot is followed by its segment's refinement block (k_refcopy fills it in): the lane notes
// where it starts and skips it.  Words outside the window (a stretch made long by a refinement
// block, or very long codes) are OR-ed into memory directly.
constexpr int EWIN = 768;

struct EmitLds {
	unsigned tok[64 * HROW];   // half of every group's tokens at a time (stage_tokens_half)
	unsigned win[EWIN];
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5))) void k_emit(Work w, unsigned *out, long out_words)
{
	__shared__ EmitLds lds[4];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const long wave = (long)blockIdx.x * 4 + wv;
	const int img = blockIdx.y;
	const ImgInfo &I = w.info[img];
	const long T = I.T;
	const long ngroups = (T + SUB - 1) / SUB;
	if (wave * 64 >= ngroups)
		return;
	const unsigned short *tok16 = w.tok16 + img * w.TS;
	const unsigned *big = w.tok_big + img * w.TS;
	const unsigned *srefs = w.seg_refs + (long)img * MAX_SEGS;
	const unsigned *btok = w.brk_tok + (long)img * MAX_SEGS;
	unsigned *rows = lds[wv].tok, *win = lds[wv].win;
	for (int i = lane; i < EWIN; i += 64)
		win[i] = 0u;
	const long S = wave * 64 + lane;
	const bool live = S < ngroups;
	const int per = w.slow[img] ? 64 : FSUBS;
	unsigned long long pos = 0;
	int o = 0;
	if (live) {
		pos = w.chunk_base[img * w.NCS + S / per] + w.lane_bits[img * w.NCS * 64 + S];
		o = w.grp_ord[img * w.NCS * 64 + S];
	}
	const long wbase = (long)(__shfl(pos, 0) >> 5);   // lane 0 is always live
	unsigned *dst = out + img * out_words;
	wave_sync();
	auto put_word = [&](long wi, unsigned v) {
		if (!v)
			return;
		const unsigned long rel = (unsigned long)(wi - wbase);
		// (the LDS word through a pointer that says so: with two plain pointers the compiler joins the branches into one
		// flat atomic on a selected 64-bit address — ten vector instructions per word)
		if (rel < (unsigned long)EWIN)
			__hip_atomic_fetch_or((__attribute__((address_space(3))) unsigned *)win + rel, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		else if (wi < out_words)
			atomicOr(dst + wi, v);
	};
	long widx = (long)(pos >> 5);
	int fill = (int)(pos & 31);
	unsigned long long acc = 0;
	auto append = [&](unsigned c, int n) {   // n <= 32 bits
		acc |= (unsigned long long)c << fill;
		fill += n;
		if (fill >= 32) {
			put_word(widx, (unsigned)acc);
			++widx;
			acc >>= 32;
			fill -= 32;
		}
	};
	const unsigned *my = rows + lane * HROW;
	const long tb = S * SUB;
	// one token, the general way: up to 64 code bits, break slots, escapes
	auto slow_token = [&](unsigned tk, long t) {
		if (!(tk & T_VOID)) {
			const unsigned v = token_run(tk, big, t);
			const int top = vli_top(o, v);
			const int z = top - o;
			const unsigned rem = v + (1u << o) - (1u << top);
			unsigned long long code = (1ull << z) | ((unsigned long long)rem << (z + 1));
			int len = z + 1 + top;
			if (!(tk & T_NOSIGN)) {
				code |= (unsigned long long)((tk >> 12) & 1u) << len;
				++len;
			}
			o = vli_next(top);
			append((unsigned)code, len < 32 ? len : 32);
			if (len > 32)
				append((unsigned)(code >> 32), len - 32);
		}
		if (tk & T_BREAK) {
			// the segment's refinement block starts here (encode.c:84-93 follows the segment's first pass)
			put_word(widx, (unsigned)acc);
			acc = 0;
			const long k = find_break_seg(btok, I.K, (unsigned)t);
			const unsigned long long at = ((unsigned long long)widx << 5) + (unsigned)fill;
			w.seg_rawoff[(long)img * MAX_SEGS + k] = at;
			const unsigned long long next = at + srefs[k];
			widx = (long)(next >> 5);
			fill = (int)(next & 31);
		}
	};
	// one token whose code is at most 16 bits (order <= 7, run < 128, neither a break slot nor a void one — the padding of an
	// image's last group goes the general way): no branches, no selects
	auto code16 = [&](unsigned tk, unsigned &code, int &len) {
		const unsigned s = (tk & T_RUN) + (1u << o);
		const int top = 31 - __builtin_clz(s);
		const int z = top - o;
		const unsigned rem = s & bfm_mask((unsigned)top);
		const int l = z + 1 + top;
		code = (rem << (z + 1)) | (1u << z) | (((tk >> 12) & 1u) << l);
		len = l + ((tk & T_NOSIGN) ? 0 : 1);
		o = (int)__builtin_elementwise_sub_sat((unsigned)top, 2u);
	};
	HalfTokens staged = load_tokens_half(tok16, wave * CHUNK, T, lane, 0);
	for (int half = 0; half < 2; ++half) {   // (uniform: every lane of the wave takes part in the staging)
		if (half)
			wave_sync();   // the first half's rows have been read
		deposit_tokens_half(rows, staged, lane);
		if (!half)
			staged = load_tokens_half(tok16, wave * CHUNK, T, lane, 1);   // on its way while the first half is walked
		wave_sync();
		if (!live)
			continue;
#pragma unroll 2
		for (int q8 = 0; q8 < 8; ++q8) {
			const int q = half * 8 + q8;
			const uint2 x2 = *reinterpret_cast<const uint2 *>(my + 2 * q8);
			const unsigned xs[2] = { x2.x, x2.y };
#pragma unroll
			for (int h = 0; h < 2; ++h) {
				const unsigned x = xs[h];
				// run >= 128 (escapes too) or a break slot in the pair, or a high order: the general path
				if ((x & ((0x0f80u | T_BREAK | T_VOID) * 0x00010001u)) != 0u || o > 7) {
					slow_token(x & 0xffffu, tb + 4 * q + 2 * h);
					slow_token(x >> 16, tb + 4 * q + 2 * h + 1);
				} else {
					unsigned c0, c1;
					int l0, l1;
					code16(x & 0xffffu, c0, l0);
					code16(x >> 16, c1, l1);
					acc |= (unsigned long long)(c0 | (c1 << l0)) << fill;   // fill < 32, the pair at most 32 bits
					fill += l0 + l1;
					if (fill >= 32) {
						put_word(widx, (unsigned)acc);
						++widx;
						acc >>= 32;
						fill -= 32;
					}
				}
			}
		}
	}
	if (live)
		put_word(widx, (unsigned)acc);
	// the window to memory: the words strictly inside the wave's stretch are its own
	const long nlive = ngroups - wave * 64 < 64 ? ngroups - wave * 64 : 64;
	const long endw = __shfl(widx, (int)nlive - 1);   // word of the last bit position (shared with whatever follows)
	wave_sync();
	for (int i = lane; i < EWIN; i += 64) {
		const long wi = wbase + i;
		if (wi - i + (i & ~63) > endw)   // uniform: nothing of this wave beyond
			break;
		if (wi > endw || wi >= out_words)
			continue;
		const unsigned v = win[i];
		if (wi > wbase && wi < endw)
			dst[wi] = v;
		else if (v)
			atomicOr(dst + wi, v);
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

// The token and refinement writers OR their edge words into the stream, so it has to start as zeros — but only
// the words the stream will occupy (its length is known after k_bitscan), not the whole output stride;
// the words k_plan filled with header, root image and plane counts stay.
__global__ __launch_bounds__(256) void k_clear_stream(Work w, unsigned *out, long out_words)
{
	const int img = blockIdx.y;
	const long first = ((long)w.info[img].hdr_bits + 31) >> 5;
	long last = (long)((w.stream_bits[img] + 31) >> 5) + 4;
	last = last < out_words ? last : out_words;
	unsigned *base = out + img * out_words;
	const bool vec = ((uintptr_t)base & 15) == 0;   // (uniform) the stream's base is only word-aligned by contract: 16-byte stores when it allows them
	for (long i = ((first & ~3l) >> 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i * 4 < last; i += (long)gridDim.x * blockDim.x) {
		unsigned *dst = base + i * 4;
		if (vec && i * 4 >= first && i * 4 + 4 <= last) {
			*reinterpret_cast<uint4 *>(dst) = make_uint4(0u, 0u, 0u, 0u);
		} else {
			for (int k = 0; k < 4; ++k)
				if (i * 4 + k >= first && i * 4 + k < last)
					dst[k] = 0u;
		}
	}
}

// --------------------------------------------------------------- k_refcopy ---
// encode.c:84-93: a segment's raw refinement bits follow its first pass.  They wait in the staging buffer entry by
// entry (tile by tile), every entry's bits from a word boundary on; their place in the stream (seg_rawoff, noted by
// k_emit) is known now.  The segment's block is cut into windows of REF_WIN stream words; a workgroup gathers the
// entries that reach into its window in LDS — each at its bit position, words added with LDS atomics — and writes
// the window out as whole stream words; only the block's first and last word, which it shares with the tokens
// around it, are added to the stream atomically.
constexpr int REF_WIN = 2048;

__global__ __launch_bounds__(256) void k_refcopy(Work w, unsigned *out, long out_words)
{
	const int img = blockIdx.y;
	const int K = w.info[img].K;
	const int *eb = w.seg_ebase + (long)img * (MAX_SEGS + 1);
	const unsigned *refscum = w.ent_refscum + img * (w.ES + 1), *refw = w.ent_refw + img * (w.ES + 1);
	const unsigned short *erefs = w.ent_refs + img * w.ES;
	const unsigned *stage = w.stage + img * w.SW;
	unsigned *dst = out + img * out_words;
	__shared__ unsigned win[REF_WIN + 2];      // [1 + word]: one guard word on either side
	__shared__ unsigned wfirst[MAX_SEGS + 1];  // windows before segment k
	__shared__ unsigned wsum[4];
	// windows per segment, and their prefix (K <= 768: three segments per thread)
	{
		unsigned mine[3], sum = 0;
#pragma unroll
		for (int u = 0; u < 3; ++u) {
			const int k = 3 * (int)threadIdx.x + u;
			unsigned nwin = 0;
			if (k < K) {
				const unsigned n = w.seg_refs[(long)img * MAX_SEGS + k];
				if (n) {
					const unsigned long long D = w.seg_rawoff[(long)img * MAX_SEGS + k];
					const unsigned long words = (unsigned long)(((D + n - 1) >> 5) - (D >> 5) + 1);
					nwin = (unsigned)((words + REF_WIN - 1) / REF_WIN);
				}
			}
			mine[u] = nwin;
			sum += nwin;
		}
		unsigned total;
		const unsigned pre = block_excl_scan(sum, wsum, total);
		unsigned run = pre;
#pragma unroll
		for (int u = 0; u < 3; ++u) {
			const int k = 3 * (int)threadIdx.x + u;
			wfirst[k] = run;
			run += mine[u];
		}
		if (threadIdx.x == blockDim.x - 1)
			wfirst[MAX_SEGS] = run;
		__syncthreads();
	}
	const unsigned nwin_all = wfirst[K < MAX_SEGS ? K : MAX_SEGS];
	for (unsigned vw = blockIdx.x; vw < nwin_all; vw += gridDim.x) {
		// the window's segment: largest k with wfirst[k] <= vw among segments that have windows
		int lo = 0, hi = K - 1;
		while (lo < hi) {
			const int mid = (lo + hi + 1) >> 1;
			if (wfirst[mid] <= vw)
				lo = mid;
			else
				hi = mid - 1;
		}
		const int k = lo;
		const unsigned wi = vw - wfirst[k];
		const unsigned n = w.seg_refs[(long)img * MAX_SEGS + k];
		const unsigned long long D = w.seg_rawoff[(long)img * MAX_SEGS + k];
		const long d0 = (long)(D >> 5), dl = (long)((D + n - 1) >> 5);
		const long ws = d0 + (long)wi * REF_WIN;                          // the window's first stream word
		const int nwords = (int)(dl - ws + 1 < REF_WIN ? dl - ws + 1 : REF_WIN);
		// the window in the block's own bit coordinates (bit x of the block is stream bit D + x)
		const long long x0 = ((long long)ws << 5) - (long long)D, x1 = x0 + 32ll * nwords;
		for (int i = threadIdx.x; i < nwords + 2; i += blockDim.x)
			win[i] = 0u;
		__syncthreads();
		// entries that reach into [x0, x1): from the last one that starts at or before x0 up to the first that starts at or after x1
		const int e0 = eb[k], e1 = eb[k + 1];
		const unsigned cbase = refscum[e0];
		int a = e0, b = e1 - 1;
		while (a < b) {   // largest entry with start <= max(x0, 0)
			const int mid = (a + b + 1) >> 1;
			if ((long long)(refscum[mid] - cbase) <= (x0 > 0 ? x0 : 0))
				a = mid;
			else
				b = mid - 1;
		}
		for (int ea = a; ea < e1; ea += blockDim.x) {   // (uniform trip count: the stride leaves together, the entries' starts only grow)
			const int e = ea + (int)threadIdx.x;
			const long long c = e < e1 ? (long long)(refscum[e] - cbase) : x1;
			if (__syncthreads_and(c >= x1))
				break;
			const unsigned r = e < e1 ? erefs[e] : 0u;
			if (!r || c >= x1)
				continue;
			const unsigned *src = stage + refw[e];
			const int nsrc = (int)((r + 31u) >> 5);
			for (int jw = 0; jw < nsrc; ++jw) {
				const long long o = c + 32ll * jw - x0;   // the source word's first bit, relative to the window
				if (o <= -32 || o >= 32ll * nwords)
					continue;
				const unsigned v = src[jw];
				const long long ob = o + 32;              // relative to the guard word
				const int idx = (int)(ob >> 5), sh = (int)(ob & 31);
				if (v) {
					atomicOr(&win[idx], v << sh);
					if (sh && idx + 1 < nwords + 2)
						atomicOr(&win[idx + 1], v >> (32 - sh));
				}
			}
		}
		__syncthreads();
		for (int i = threadIdx.x; i < nwords; i += blockDim.x) {
			const long wd = ws + i;
			if (wd >= out_words)
				continue;
			const unsigned v = win[1 + i];
			if (wd == d0 || wd == dl) {   // shared with the tokens before / after the block
				if (v)
					atomicOr(dst + wd, v);
			} else {
				dst[wd] = v;
			}
		}
		__syncthreads();
	}
}

} // namespace

// ------------------------------------------------------------------ driver ---

enum {
	SLOT_PK_CUM = 2, SLOT_PK_SMALL, SLOT_PK_ENT, SLOT_PK_TOKBIG, SLOT_PK_TOK16, SLOT_PK_LUT, SLOT_PK_CHUNK, SLOT_PK_STAGE,
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

extern "C" int dwtx_encode_planes(dwtx_ctx *ctx, const int32_t *lin, int W, int H, int C, int n, long capacity,
	uint8_t *out, size_t out_stride, dwtx_stream_info *dev_info)
{
	return dwtx_encode_planes_ex(ctx, lin, nullptr, 0u, 0u, W, H, C, n, capacity, out, out_stride, dev_info);
}

// geometry, tiles and the histogram records of n images (what both the forward transform's histograms and the
// entropy stage start from)
static int pack_geometry(dwtx_ctx *ctx, int W, int H, int C, int n, PackGeom &g, Work &w, dwtx_tiles &tiles)
{
	{
		int lengths[DWTX_MAX_LEVELS], pixels[DWTX_MAX_LEVELS], widths[DWTX_MAX_LEVELS], heights[DWTX_MAX_LEVELS];
		g.levels = dwtx_compute_lengths(lengths, pixels, widths, heights, W, H, DWTX_MIN_LEN);
		for (int l = 0; l <= g.levels; ++l)
			g.pixels[l] = pixels[l];
		for (int l = 0; l < g.levels; ++l)
			g.side[l] = lengths[l + 1];
		g.side[g.levels] = 0;
	}
	g.pyr = nullptr;
	g.fine16 = nullptr;
	g.lv16 = 0u;
	g.sq_levels = 0;
	g.C = C;
	g.W = W;
	g.H = H;
	g.total = (long)W * H;
	const int rc = dwtx_get_tiles(ctx, W, H, &tiles);
	if (rc)
		return rc;
	for (int l = 0; l <= g.levels; ++l)
		g.tile_first[l] = tiles.tile_first[l];
	g.tile_base = tiles.base;
	g.tile_cnt = tiles.cnt;
	g.tile_blk = tiles.blk;
	memset(&w, 0, sizeof(w));
	w.NT = tiles.NT;
	w.NTP = (tiles.NT + 3) / 4 * 4;
	const size_t b = align_up(sizeof(unsigned short) * (size_t)n * C * w.NT * NCUM, 256);
	w.cum = (unsigned short *)dwtx_scratch(ctx, SLOT_PK_CUM, b + sizeof(unsigned) * (size_t)n * C * w.NTP);
	if (!w.cum)
		return DWTX_ERR_NOMEM;
	w.tile_mx = (unsigned *)((char *)w.cum + b);
	return DWTX_OK;
}

// The forward transform of the same n images, queued after this on the context's stream, adds the histograms of the
// levels it can take to these records (lift.hip); dwtx_encode_planes_ex is then told which levels are done.
int dwtx_hist_begin(dwtx_ctx *ctx, int W, int H, int C, int n, dwtx_hist_sink *sink)
{
	if (!ctx || !sink || (C != 1 && C != 3) || n < 1 || n > 65535)
		return DWTX_ERR_ARG;
	DWTX_ENTER(ctx);
	DWTX_CHECK_DIMS(W, H);
	PackGeom g;
	Work w;
	dwtx_tiles tiles;
	const int rc = pack_geometry(ctx, W, H, C, n, g, w, tiles);
	if (rc)
		return rc;
	hipLaunchKernelGGL(k_hist_init, dim3(dwtx_cdiv(w.NT * (NCUM / 2), 256), n * C), dim3(256), 0, ctx->stream, g, w);
	DWTX_LAUNCH_CHECK();
	sink->cum32 = reinterpret_cast<unsigned *>(w.cum);
	sink->tile_mx = w.tile_mx;
	sink->NT = w.NT;
	sink->NTP = w.NTP;
	sink->tiles = tiles;
	return DWTX_OK;
}

#ifdef DWTX_DEBUG_HOOKS
// Development builds only (tools/dbg_hist.py; not in include/dwtx.h, not in the product library): the tiles' histogram records
// as the last forward transform of this geometry left them — host_cum32 [n*C][NT][16], host_mx [n*C][NTP], tile_first [levels + 1].
extern "C" int dwtx_debug_hist_copy(dwtx_ctx *ctx, int W, int H, int C, int n, unsigned *host_cum32, unsigned *host_mx, int *NT, int *NTP, int *tile_first)
{
	PackGeom g;
	Work w;
	dwtx_tiles tiles;
	const int rc = pack_geometry(ctx, W, H, C, n, g, w, tiles);
	if (rc)
		return rc;
	*NT = w.NT;
	*NTP = w.NTP;
	for (int l = 0; l <= g.levels; ++l)
		tile_first[l] = tiles.tile_first[l];
	if (host_cum32) {
		DWTX_HIP(hipMemcpyAsync(host_cum32, w.cum, sizeof(unsigned) * 16 * (size_t)n * C * w.NT, hipMemcpyDeviceToHost, ctx->stream));
		DWTX_HIP(hipMemcpyAsync(host_mx, w.tile_mx, sizeof(unsigned) * (size_t)n * C * w.NTP, hipMemcpyDeviceToHost, ctx->stream));
		DWTX_HIP(hipStreamSynchronize(ctx->stream));
	}
	return DWTX_OK;
}
#endif

// pyr / sq_levels: the ring levels flagged in sq_levels are not in `lin`; their tiles are read from the
// 32x32 squares of the pyramid planes `pyr` (same plane order, pitch W) — see hilbert_dev.h
int dwtx_encode_planes_ex(dwtx_ctx *ctx, const int32_t *lin, const int32_t *pyr, unsigned sq_levels, unsigned hist_levels, int W, int H, int C,
	int n, long capacity, uint8_t *out, size_t out_stride, dwtx_stream_info *dev_info, dwtx_p16 p16)
{
	if (!ctx || !lin || !out || !dev_info || (C != 1 && C != 3) || n < 1 || n > 65535 || (out_stride & 3) || out_stride < 8)
		return DWTX_ERR_ARG;
	DWTX_ENTER(ctx);
	DWTX_CHECK_DIMS(W, H);
	if (sq_levels && (!pyr || (sq_levels & ~dwtx_square_levels(W, H)) || ((uintptr_t)pyr & 15)))
		return DWTX_ERR_ARG;
	PackGeom g;
	Work w;
	dwtx_tiles tiles;
	{
		const int rc = pack_geometry(ctx, W, H, C, n, g, w, tiles);   // (the records dwtx_hist_begin handed out, if it was called: same slot, same size)
		if (rc)
			return rc;
	}
	g.pyr = pyr;
	g.sq_levels = sq_levels;
	g.fine16 = p16.planes;
	g.lv16 = p16.planes ? p16.levels : 0u;
	if (p16.planes && ((p16.levels & ~sq_levels) || ((uintptr_t)p16.planes & 15)))   // (whole squares only: the cut blocks come through lin)
		return DWTX_ERR_ARG;
	const int NT = tiles.NT;
	// k_hist counts the tiles of the levels the forward transform has not done; when those are the coarse levels only (the
	// finest ones are a suffix of the tiles) its grid ends there
	int hist_tiles = NT;
	for (int l = g.levels - 1; l >= 0 && ((hist_levels >> l) & 1u); --l)
		hist_tiles = tiles.tile_first[l];
	w.ES = (long)NT * C * MAX_PLANES + 16;
	w.TS = ((long)C * (g.total - g.pixels[0]) + MAX_SEGS + 8 + 63) / 64 * 64;   // multiple of 64: every image's token arrays start vector-aligned
	w.NCS = (w.TS / SUB + FSUBS - 1) / FSUBS + 2;   // waves of the fast order pass (>= chunks of the exact one)
	w.NGS = (w.NCS + GROUP - 1) / GROUP;
	w.NCB = (w.ES + CARRY_BLOCK - 1) / CARRY_BLOCK;
	// refinement bits: at most MAX_PLANES-1 per coefficient, every entry's bits rounded up to a word
	w.SW = (long)(((unsigned long long)C * (unsigned long long)(g.total - g.pixels[0]) * (MAX_PLANES - 1) + 31) / 32) + w.ES + 64;
	const int nplanes = n * C;

	// carve scratch
	{
		size_t off = 0;
		auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
		const size_t o_info = take(sizeof(ImgInfo) * n);
		const size_t o_slow = take(sizeof(int) * n);
		const size_t o_sbits = take(sizeof(unsigned long long) * n);
		const size_t o_sd = take(sizeof(int) * (size_t)n * MAX_SEGS);
		const size_t o_eb = take(sizeof(int) * (size_t)n * (MAX_SEGS + 1));
		const size_t o_sr = take(sizeof(unsigned) * (size_t)n * MAX_SEGS);
		const size_t o_ro = take(sizeof(unsigned long long) * (size_t)n * MAX_SEGS);
		const size_t o_bt = take(sizeof(unsigned) * (size_t)n * MAX_SEGS);
		const size_t o_sx = take(sizeof(int) * (size_t)n * 48 * MAX_PLANES);
		const size_t o_lv = take(sizeof(unsigned) * (size_t)n * 48);
		char *small = (char *)dwtx_scratch(ctx, SLOT_PK_SMALL, off);
		if (!small)
			return DWTX_ERR_NOMEM;
		w.info = (ImgInfo *)(small + o_info);
		w.slow = (int *)(small + o_slow);
		w.stream_bits = (unsigned long long *)(small + o_sbits);
		w.seg_desc = (int *)(small + o_sd);
		w.seg_ebase = (int *)(small + o_eb);
		w.seg_refs = (unsigned *)(small + o_sr);
		w.seg_rawoff = (unsigned long long *)(small + o_ro);
		w.brk_tok = (unsigned *)(small + o_bt);
		w.segidx = (int *)(small + o_sx);
		w.live = (unsigned *)(small + o_lv);
		DWTX_HIP(hipMemsetAsync(small, 0, o_sd, ctx->stream));
		if (ctx->opt[DWTX_OPT_EXACT_ORDERS])   // test hook: take the hierarchical 32-state pass for every image
			DWTX_HIP(hipMemsetAsync(small + o_slow, 1, sizeof(int) * n, ctx->stream));

		off = 0;
		const size_t o_on = take(sizeof(short) * (size_t)n * w.ES);
		const size_t o_ze = take(sizeof(short) * (size_t)n * w.ES);
		const size_t o_re = take(sizeof(short) * (size_t)n * w.ES);
		const size_t o_tz = take(sizeof(short) * (size_t)n * w.ES);
		const size_t o_sg = take(sizeof(short) * (size_t)n * w.ES);
		const size_t o_tb = take(sizeof(unsigned) * (size_t)n * (w.ES + 1));
		const size_t o_rc = take(sizeof(unsigned) * (size_t)n * (w.ES + 1));
		const size_t o_rw = take(sizeof(unsigned) * (size_t)n * (w.ES + 1));
		const size_t o_ca = take(sizeof(RunMap) * (size_t)n * w.NCB);
		const size_t o_ci = take(sizeof(unsigned) * (size_t)n * w.NCB);
		const size_t o_ek = take(sizeof(unsigned) * 3 * (size_t)n * w.NCB);
		char *ent = (char *)dwtx_scratch(ctx, SLOT_PK_ENT, off);
		if (!ent)
			return DWTX_ERR_NOMEM;
		w.ent_ones = (unsigned short *)(ent + o_on);
		w.ent_zeros = (unsigned short *)(ent + o_ze);
		w.ent_refs = (unsigned short *)(ent + o_re);
		w.ent_tz = (unsigned short *)(ent + o_tz);
		w.ent_seg = (unsigned short *)(ent + o_sg);
		w.ent_tokbase = (unsigned *)(ent + o_tb);
		w.ent_refscum = (unsigned *)(ent + o_rc);
		w.ent_refw = (unsigned *)(ent + o_rw);
		w.carry_agg = (RunMap *)(ent + o_ca);
		w.carry_in = (unsigned *)(ent + o_ci);
		w.ent_blk = (unsigned *)(ent + o_ek);

		w.tok_big = (unsigned *)dwtx_scratch(ctx, SLOT_PK_TOKBIG, sizeof(unsigned) * (size_t)n * w.TS);
		w.tok16 = (unsigned short *)dwtx_scratch(ctx, SLOT_PK_TOK16, sizeof(unsigned short) * (size_t)n * w.TS);
		w.stage = (unsigned *)dwtx_scratch(ctx, SLOT_PK_STAGE, sizeof(unsigned) * (size_t)n * w.SW);
		if (!w.tok_big || !w.tok16 || !w.stage)
			return DWTX_ERR_NOMEM;

		w.sublut = (unsigned char *)dwtx_scratch(ctx, SLOT_PK_LUT, (size_t)n * w.NCS * 64 * 32);
		off = 0;
		const size_t o_lut = take((size_t)n * w.NCS * 32);
		const size_t o_gl = take((size_t)n * w.NGS * 32);
		const size_t o_ce = take((size_t)n * w.NCS);
		const size_t o_ge = take((size_t)n * w.NGS);
		const size_t o_go = take((size_t)n * w.NCS * 64);
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
		w.grp_ord = (unsigned char *)(ch + o_go);
		w.chunk_bits = (unsigned long long *)(ch + o_cb);
		w.chunk_base = (unsigned long long *)(ch + o_cs);
		w.lane_bits = (unsigned long long *)(ch + o_lb);
	}

	hipStream_t s = ctx->stream;
	const long out_words = (long)(out_stride / 4);
	unsigned *outw = (unsigned *)out;
	// k_plan stores the first words (header, root image, plane counts) outright; the rest of the stream is
	// cleared by k_clear_stream once its length is known

	if (hist_tiles > 0)
		hipLaunchKernelGGL(k_hist, dim3(dwtx_cdiv(hist_tiles, 4 * HIST_TPW), nplanes), dim3(256), 0, s, g, lin, w, hist_levels);
	hipLaunchKernelGGL(k_plan, dim3(n), dim3(1024), 0, s, g, lin, w, outw, out_words, capacity);
	hipLaunchKernelGGL(k_entries_count, dim3((unsigned)w.NCB, n), dim3(ENT_BLOCK), 0, s, g, w);
	hipLaunchKernelGGL(k_entries_blocks, dim3(n), dim3(ENT_BLOCK), 0, s, w);
	hipLaunchKernelGGL(k_entries_finish, dim3((unsigned)w.NCB, n), dim3(ENT_BLOCK), 0, s, w);
	hipLaunchKernelGGL(k_entries_segs, dim3(n), dim3(ENT_BLOCK), 0, s, w);
	hipLaunchKernelGGL(k_cut, dim3(n), dim3(ENT_BLOCK), 0, s, w, capacity, (int)(ctx->opt[DWTX_OPT_NO_CAPACITY_CUT] != 0));
	hipLaunchKernelGGL(k_code<false>, dim3(dwtx_cdiv(NT, 4), nplanes), dim3(256), 0, s, g, lin, w);
	hipLaunchKernelGGL(k_code<true>, dim3(dwtx_cdiv(NT, 4) < 64 ? dwtx_cdiv(NT, 4) : 64, nplanes), dim3(256), 0, s, g, lin, w);
	hipLaunchKernelGGL(k_carry_local, dim3((unsigned)w.NCB, n), dim3(CARRY_THREADS), 0, s, w);
	hipLaunchKernelGGL(k_carry_blocks, dim3(n), dim3(CARRY_BLOCK), 0, s, w);
	hipLaunchKernelGGL(k_carry_apply, dim3((unsigned)w.NCB, n), dim3(CARRY_THREADS), 0, s, w);
	hipLaunchKernelGGL(k_gorder, dim3((int)((w.NCS + GO_WAVES - 1) / GO_WAVES), n), dim3(64 * GO_WAVES), 0, s, w);
	// exact pass: only images the fast pass flagged (their kernels return at once otherwise)
	hipLaunchKernelGGL(k_lut, dim3(512, n), dim3(256), 0, s, w);
	hipLaunchKernelGGL(k_chain_groups, dim3(64, n), dim3(256), 0, s, w);
	hipLaunchKernelGGL(k_chain_image, dim3(n), dim3(1024), 0, s, w);
	hipLaunchKernelGGL(k_gorder_exact, dim3(512, n), dim3(256), 0, s, w);
	hipLaunchKernelGGL(k_bitscan, dim3(n), dim3(1024), 0, s, w, capacity);
	{
		const long cb = (out_words / 4 + 256) / 256;
		hipLaunchKernelGGL(k_clear_stream, dim3((unsigned)(cb < 1024 ? cb : 1024), n), dim3(256), 0, s, w, outw, out_words);
	}
	hipLaunchKernelGGL(k_emit, dim3((unsigned)((w.TS / CHUNK + 1 + 3) / 4), n), dim3(256), 0, s, w, outw, out_words);
	hipLaunchKernelGGL(k_refcopy, dim3(256, n), dim3(256), 0, s, w, outw, out_words);
	DWTX_LAUNCH_CHECK();
	static_assert(sizeof(dwtx_stream_info) == sizeof(ImgInfo), "ImgInfo is the device image of dwtx_stream_info");
	DWTX_HIP(hipMemcpyAsync(dev_info, w.info, sizeof(ImgInfo) * (size_t)n, hipMemcpyDeviceToDevice, s));
	return DWTX_OK;
}
