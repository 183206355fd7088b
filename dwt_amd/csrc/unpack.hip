// unpack.hip — the decoder's entropy stage on the GPU (decode.c:67-134,174-250
// over rle.h:66-103, vli.h:86-101, bits.h:80-106).
//
// The stream is sequential by construction: where a token starts depends on
// every token before it (adaptive VLI order, zero runs that straddle segments,
// raw bits whose count depends on what earlier planes made significant).  The
// work is split so that only the irreducible part stays serial:
//
//   k_nch       chunks of every stream that hold bytes (the tables below are laid
//               out for the stream stride but worked on only that far).
//   k_peek + k_clear_bitmaps  the plane counts of every stream's preamble bound how much of
//               its symbol bitmap can be used: only that much is cleared.
//   k_link_first/k_link_work/k_scan_*  speculative 128-bit chunk parse (a thread parses a run of eight chunks
//               one after the other, then relaxation rounds repair the runs' seams) that lets the walker jump
//               over stitched stretches of the stream (see below): one family of
//               recorded paths, two (even / odd start) for a part of the batch whose
//               walk gives up on a parity-locked stretch (k_part_reset).
//   k_tokenize  one wave per image (all lanes on the same uniform values): walks
//               header, root image, plane counts and the segment schedule.  It
//               touches no coefficient: per-(channel, level) counters of
//               not-yet-significant coefficients tell it how many symbols each
//               segment holds.  On a stitched path it hops over whole runs of
//               chunks (64-way search in the prefix sums); otherwise it parses one
//               chunk at a time with chunk_scan, counting only.  Output: hop
//               records, per segment the stream offset of its refinement block,
//               and for the few tokens it reads bit by bit two bits in `symbits`
//               (one flag, sign) at (segment symbol base + symbol index).
//               Truncated streams simply stop here; what was parsed stays valid
//               (decode.c:204-205).
//   k_tokenize<true> + k_segprep / k_segjoin  the same walk with one wave per SEGMENT
//               when a sidecar index (include/dwtx.h dwtx_index: the walk's state at
//               every segment start, recorded by the serial walk) is offered; k_segjoin
//               checks that the segments fit together, else the serial walk runs.
//   k_hopbits   re-parses every chunk (piece) the walker accounted for and sets
//               its symbol bits.
//   k_rank + k_count  per plane, descending, on per-tile COUNTS only: first pass-1
//               symbol index of every 1024-coefficient tile (exclusive scan of the
//               tiles' insignificant counts) and the ones each tile gains (popcount
//               of its slice of `symbits`; also flags planes that turn coefficients
//               of the tile on).
//   k_apply_all one wave per tile, all planes in registers: an insignificant
//               coefficient is pass-1 symbol #rank -> bits from `symbits`; a
//               significant one is refinement bit #(index - rank) -> read straight
//               from the stream.  Written once, in two's complement
//               (decode.c:102-117).
// Batches run as two to four parts on streams of their own (the second also clears
// `symbits` while the first builds its tables); see dwtx_decode_planes_ex.
#include "hilbert_dev.h"

#include <stdlib.h>
#include <string.h>
#include <vector>

namespace {

constexpr int TILE = 1024;
constexpr int MAX_PLANES = 16;
constexpr int MAX_SEGS = 3 * 16 * MAX_PLANES;
constexpr int CH_LOG2 = 7;             // speculative-parse chunk: 128 stream bits
constexpr int CH_BITS = 1 << CH_LOG2;
constexpr int SCAN_BLOCK = 1024;       // chunks per scan workgroup
constexpr int LINK_ROUNDS = 12;         // relaxation rounds: fewer leave more chunks unstitched, more let paths that ran through refinement blocks take over (both cost the walker; measured optimum 10-12)
constexpr int FAM = 2;                 // speculative path families: start at bit 0 / bit 1 of a chunk (see k_link_first)

struct UnpackGeom {
	int levels, C, W, H;
	long total;                               // W*H of the full image
	long lin_stride;                          // ints per plane in the lin buffer
	int pixels[DWTX_MAX_LEVELS + 1];
	int tile_first[DWTX_MAX_LEVELS + 1];
	int levels_max;                           // decode.c:163-171 PIXELS cap
	int *pyr;                                 // pyramid planes (pitch W) that take the tiles of the levels in sq_levels, or null
	short *fine16;                            // or null: the squares of the ring levels in lv16 go here as 16-bit coefficients instead (same positions and pitch)
	unsigned lv16;
	unsigned sq_levels;                       // ring levels written as 32x32 squares of the pyramid instead of into `lin` (hilbert_dev.h)
	int side[DWTX_MAX_LEVELS + 1];            // outer side of ring level l (lengths[l+1])
	// the tiles (dwtx_tiles): ring index of a tile's first coefficient, its coefficients, its block on the level's curve
	const int *tile_base;
	const unsigned short *tile_cnt;
	const int *tile_blk;
};

struct DecInfo {
	int status;            // 0 ok, 1 = header / root image / plane counts unreadable (decode.c exits 1), 2 = more than MAX_PLANES bit planes
	int W, H, C;
	int levels;
	int planes[3];
	int pmax;
	int level;             // finest level any segment touched (decode.c:197,203,219,236); -1 none
	int nsegs;
	int truncated;         // bit 0: the walk stopped early; bit 1: because a read ran past the end of the data (bytes.h:99-103)
	int missing[48];       // decode.c:193-196: planes not fully decoded, [c*16 + l]
	unsigned long long bits_used;
	unsigned hops, hopped_chunks;      // walker statistics: jumps over stitched chunks
	unsigned walked_tokens;            // tokens the walker had to parse itself
	unsigned zeros_left;               // the run-length reader's counter at the end (rle.h:43-46 reports it when > 1)
};

// The sidecar index (never part of the .dwt): the token walk's state where each segment's
// first pass begins.  With it every segment can be walked by a wave of its own (k_tokenize<true>), and because a
// segment walked from entry k's state must arrive exactly in entry k+1's, the index is checked for free
// (k_segjoin): a wrong or foreign index only costs the fallback to the serial walk.
struct SegIndex {
	unsigned long long bit;       // stream position of the segment's first pass
	unsigned long long sym_base;  // first symbol slot of the segment in the image's bitmap
	unsigned n1;                  // symbols of the first pass (coefficients still insignificant)
	unsigned cnt;                 // rle.h:25 zero-run counter on entry
	unsigned desc;                // c | l << 4 | (p + 1) << 8
	unsigned order;               // vli.h:24 order on entry
};

struct SegResult {
	unsigned long long bit;       // position after the segment (first pass and refinement block)
	unsigned order, cnt, ones;
	unsigned nhops, hopped, walked;
	unsigned ok;                  // 1: walked to its end without a stop condition
	unsigned pad;
};

struct DWork {
	DecInfo *info;                  // [n]
	int *seg_desc;                  // [n][MAX_SEGS]
	unsigned long long *seg_symbase; // [n][MAX_SEGS]
	unsigned long long *seg_b2;     // [n][MAX_SEGS]
	unsigned *seg_n2done;           // [n][MAX_SEGS]
	int *segidx;                    // [n][3][16][MAX_PLANES] -> k+1
	int *nonsig;                    // [n][3][16]
	unsigned *symbits;              // [n][BW] words, 2 bits per pass-1 symbol (one, sign)
	unsigned short *tile_nonsig;    // [nplanes][NT] coefficients of the tile that are still insignificant
	unsigned *tile_rank;            // [nplanes][MAX_PLANES][NT] insignificant coefficients before the tile, per plane
	unsigned long long *count_base; // [nplanes][16] k_rank -> k_count: 1 + 2 * symbol base of the (plane, level)'s segment at the current bit plane, 0 = none
	long BW;                        // bitmap words per image
	int NT;
	// speculative chunk parse (see k_link_first): per 128-bit chunk of every stream
	unsigned short *exitX;          // [n*FAM][NCH] rel | order<<8: state in which the chunk's recorded path leaves it; 0xffff dead
	unsigned short *entryE;         // [n*FAM][NCH] state in which that recorded path entered (valid if == exitX[chunk-1])
	unsigned long long *cs;         // [n][NCH+1] exclusive prefix of symbols (run+1) along the arriving paths
	unsigned *ct;                   // [n][NCH+1] exclusive prefix of tokens
	unsigned *cg;                   // [n][NCH+1] exclusive prefix of "arriving path does not rejoin" flags
	unsigned long long *part_s;     // [n][NB]
	unsigned *part_t, *part_g;      // [n][NB]
	int *hop_seg;                   // [n][w.MAX_HOPS]
	unsigned *hop_first, *hop_last, *hop_q0;   // [n][w.MAX_HOPS]
	unsigned *hop_entry;            // [n][MAX_HOPS] 0xffffffff = stitched run (enter at exitX[first-1]); else off | order<<8
	unsigned *hop_ntok;             // [n][MAX_HOPS] tokens to apply (walker-parsed chunk pieces)
	unsigned *breaks;               // [n*FAM][NCH] chunk indices whose arriving path does not rejoin, ascending
	unsigned *todo[2];              // [n*FAM][LINK_SHARDS][todo_cap] chunks to re-parse, this round / next round
	unsigned *todo_count;           // [LINK_ROUNDS + 1][n*FAM][LINK_SHARDS]: entries queued for round r, zeroed once per call
	long todo_round;                // elements per round of todo_count
	long todo_cap;
	unsigned long long *dbg;
	int *nhops;                     // [n]
	int *nch;                       // [n] chunks that hold stream bytes (rounded so that nch+1 is a multiple of 4), <= NCH
	long NCH, NB;
	long MAX_HOPS;
	SegIndex *idx;                  // [n][MAX_SEGS] state at the start of every segment: written by the serial walk, read by the indexed one
	SegResult *segres;              // [n][MAX_SEGS] indexed walk: where each segment's own wave ended up
	int *idx_nsegs;                 // [n] segments in idx (indexed walk: 0 = no usable index for this image)
	unsigned *seg_slot;             // [n][MAX_SEGS + 1] indexed walk: first hop record of every segment's private stretch
	int fam;                        // families in use this pass: 1 (the usual case) or FAM (see k_link_first); the tables keep FAM rows per image either way
	unsigned streak_max, scans_base;   // the one-family walk's patience with chunks it parses by hand (k_tokenize)
};

// grid row -> virtual stream (image * FAM + family) when only w.fam of the FAM families run
__device__ __forceinline__ int vstream(const DWork &w, int row) { return (row / w.fam) * FAM + row % w.fam; }

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
	// bits.h:94-106 for any n, as the reference's binary does it: bit i lands at position i modulo 32
	__device__ bool read_any(unsigned n, unsigned &v)
	{
		if (n <= 32u)
			return read((int)n, v);
		unsigned a = 0;
		for (unsigned i = 0; i < n; ++i) {
			unsigned bit;
			if (!read(1, bit))
				return false;
			a |= bit << (i & 31u);
		}
		v = a;
		return true;
	}
	// The same for any bits at all, as the reference's binary reads them (only damaged streams get here: the
	// order passes 31).  vli.h:90-91 adds 1 << order per zero and bits.h:100 ORs bit << i: x86 takes both shift
	// counts modulo 32 and the sums wrap, and a negative result is an error to every caller (rle.h:72,
	// decode.c:122) — the decode stops there, quietly, while running out of data prints bytes.h:101.
	// 0: value read, 1: end of data, 2: the reference's reader returns a negative number.
	__device__ int vli_any(int &order, unsigned &val)
	{
		if (vli(order, val))
			return 0;
		unsigned ord = (unsigned)order, sum = 0;
		for (;;) {
			if (avail() >= 64 && !peek()) {   // 64 zeros: every shift count modulo 32 occurs twice
				sum += 2u * 0xffffffffu;
				ord += 64;
				b += 64;
				continue;
			}
			unsigned bit;
			if (!read(1, bit))
				return 1;
			if (bit)
				break;
			sum += 1u << (ord & 31u);
			++ord;
		}
		unsigned a = 0;
		for (unsigned i = 0; i < ord; ++i) {
			unsigned bit;
			if (!read(1, bit))
				return 1;
			a |= bit << (i & 31u);
		}
		order = ord >= 2u ? (int)(ord - 2u) : 0;
		const int v = (int)(a + sum);
		if (v < 0)
			return 2;
		val = (unsigned)v;
		return 0;
	}
};

// Pass-1 symbols that are ones, two bits per symbol in one array: bit 2P = "symbol P is a one",
// bit 2P+1 = its sign (P = segment symbol base + index in the segment's first pass).
struct BitmapWriter {
	unsigned *sym;
	long cur;
	unsigned acc;
	bool lead;   // the walker runs with all lanes of its wave in step: only one of them may add to memory
	__device__ __forceinline__ void flush()
	{
		if (cur >= 0 && acc && lead)   // hop chunks add their bits to the same words later (k_hopbits)
			atomicOr(sym + cur, acc);
		acc = 0;
	}
	__device__ __forceinline__ void set_one(unsigned long long pos)
	{
		const long wi = (long)(pos >> 4);
		if (wi != cur) {
			flush();
			cur = wi;
		}
		acc |= 1u << ((pos & 15) * 2);
	}
	__device__ __forceinline__ void set_sign(unsigned long long pos) { acc |= 2u << ((pos & 15) * 2); }
};

// ------------------------------------------------- speculative chunk parse ---
// Under the pass-1 grammar (VLI token + sign bit, rle.h:56-64 / vli.h:67-84) the
// parser state at a token start is (bit position b, order o) and the next state
// is a pure function of it: f(b,o) = (b + 2z + o + 2, max(o+z-2, 0)), z = zeros
// before the next one bit.  Paths that ever share a state coincide from there
// on, and in practice they do merge within a few tokens.  So every 128-bit chunk
// is parsed from (chunk start, order 0) [path P, k_link_first] and again from the
// state in which P of the previous chunk arrives [path Q, k_link].  Where Q
// leaves the chunk in the same state as P, the stream is "stitched": a walker
// that enters a chunk in the state P of the previous chunk left it in follows Q
// chunk after chunk, and prefix sums of Q's token and symbol counts let it jump
// over any number of stitched chunks with a binary search instead of parsing.

struct ChunkWin {
	unsigned long long w0, w1, w2;
};

__device__ __forceinline__ ChunkWin chunk_load(const unsigned long long *w64, long n64, long chunk)
{
	ChunkWin c;
	const long i = chunk * (CH_BITS / 64);
	c.w0 = i < n64 ? w64[i] : 0ull;
	c.w1 = i + 1 < n64 ? w64[i + 1] : 0ull;
	c.w2 = i + 2 < n64 ? w64[i + 2] : 0ull;
	return c;
}

__device__ __forceinline__ unsigned long long chunk_win(const ChunkWin &c, int off)   // off in [0, 128)
{
	const int r = off & 63;
	const unsigned long long lo = off >> 6 ? c.w1 : c.w0, hi = off >> 6 ? c.w2 : c.w1;
	return r ? (lo >> r) | (hi << (64 - r)) : lo;
}

// ((1 << width) - 1) << offset in one instruction (v_bfm_b32 takes five bits of each; the compiler spells the expression
// with two shifts and a not)
__device__ __forceinline__ unsigned bfm(unsigned width, unsigned offset)
{
	unsigned r;
	asm("v_bfm_b32 %0, %1, %2" : "=v"(r) : "v"(width), "v"(offset));
	return r;
}

// one token of the pass-1 grammar at order o; false if it cannot be a token this codec wrote
__device__ __forceinline__ bool token_at(unsigned long long win, int o, int &len, unsigned &run, unsigned &neg, int &next)
{
	if (!win)
		return false;
	const int z = __builtin_ctzll(win);
	const int top = o + z;
	if (top > 31)
		return false;
	len = z + top + 2;
	run = (top ? (unsigned)(win >> (z + 1)) & (unsigned)((1ull << top) - 1ull) : 0u) + (1u << top) - (1u << o);
	neg = (unsigned)(win >> (z + 1 + top)) & 1u;
	next = top >= 2 ? top - 2 : 0;
	return true;
}

// Walk the tokens of one chunk from (off, o) until off leaves the chunk.  visit(run, neg) is called
// for every token before it is consumed and may return false to stop there (off/o then still
// describe that token).  Returns false if some position cannot hold a token (dead path).
// The chunk is cut into four 32-bit segments with compile-time register indices; almost every
// token fits the 32-bit window of its segment (v_alignbit), the rare long one takes the 64-bit path.
// Round 4: the parse kernels issued as many SCALAR instructions as vector ones (k_link_first 1 731 against 1 620 per
// wave, profiles/r04_pmc_kernels.txt) — and the scalar unit, one instruction per cycle for the CU's four SIMDs, has
// exactly the capacity of the vector units at 4.1 cycles per instruction (profiles/r04_valu_peak.json): the nested
// divergent branches around every token (fits the window? can be a token at all? goes on?) cost 16 scalar
// instructions per token.  So the token loop has ONE condition now (`off < bound`): a lane that meets a long token,
// stops or dies sets its bound to 0, the long token is dealt with outside the loop (a uniform test per segment) and
// the lane then re-enters it; visit(run, neg, counts) is called for every token the loop looks at and must do nothing
// when `counts` is false (the same token comes again from the long path).
// (the segment loop is unrolled — the registers of a segment are compile-time indices — and the compiler then reports that it did not unroll the
// token loops inside it as well, which nobody asked for: that diagnostic is switched off for this file's kernels)
#pragma clang diagnostic ignored "-Wpass-failed"
template <class F>
__device__ __forceinline__ bool chunk_walk(const ChunkWin &c, int &off, int &o, F &&visit)
{
	const unsigned d[6] = { (unsigned)c.w0, (unsigned)(c.w0 >> 32), (unsigned)c.w1, (unsigned)(c.w1 >> 32),
		(unsigned)c.w2, (unsigned)(c.w2 >> 32) };
	bool dead = false, stop = false;
#pragma unroll
	for (int seg = 0; seg < CH_BITS / 32; ++seg) {
		for (;;) {   // (uniform: once, and once more for every lane-token that did not fit its window)
			const int bound = 32 * (seg + 1);
			bool fits = !(stop || dead), go = true;
			// the token loop: no branch inside — a token that does not fit its window, or at which the visitor stops, only
			// takes the lane out of the loop (visit's third argument says whether the token counts)
			while (off < bound && fits && go) {
				// (v_alignbit takes the low five bits of the count: off itself will do; the sentinel keeps the bit search
				// inside the word and sends an empty window down the long path)
				const unsigned w32 = __builtin_amdgcn_alignbit(d[seg + 1], d[seg], (unsigned)off);
				const int z = __builtin_ctz(w32 | 0x80000000u);
				const int top = o + z;
				const int len = z + top + 2;
				fits = len <= 32;
				// top remainder bits after the one, plus 2^top - 2^o = (2^z - 1) << o (bit-field extract / mask instructions)
				const unsigned run = __builtin_amdgcn_ubfe(w32, (unsigned)(z + 1), (unsigned)top) + bfm((unsigned)z, (unsigned)o);
				const unsigned neg = __builtin_amdgcn_ubfe(w32, (unsigned)(len - 1), 1u);
				go = visit(run, neg, fits);
				const bool adv = fits && go;
				off += adv ? len : 0;
				o = adv ? (int)__builtin_elementwise_sub_sat((unsigned)top, 2u) : o;
			}
			stop = stop || !go;
			const bool slow = !fits && !stop && !dead;
			if (!ballot64(slow))
				break;
			if (slow) {
				const int r = off & 31;
				const unsigned long long lo64 = d[seg] | ((unsigned long long)d[seg + 1] << 32);
				const unsigned long long w64 = r ? (lo64 >> r) | ((unsigned long long)d[seg + 2] << (64 - r)) : lo64;
				int len, next;
				unsigned run, neg;
				if (!token_at(w64, o, len, run, neg, next)) {
					dead = true;
				} else if (visit(run, neg, true)) {
					off += len;
					o = next;
				} else {
					stop = true;
				}
			}
		}
	}
	return !dead;
}

// chunk_walk for visitors that never stop the walk (the table kernels, k_hopbits' stitched stretches).  The token loop has
// no select at all: a lane whose token does not fit the 32-bit window still "takes" it — that carries it out of the
// segment, hence out of the loop — and what the loop did for that token is taken back from copies before the 64-bit path
// reads the token again: keep() copies the visitor's registers at the start of every token (moves: the cheap class of
// profiles/r04_valu_peak.json, where the selects and carries they replace are of the dear one), undo() puts them back.
// visit(run, neg, fits): whatever it does to memory must do nothing when `fits` is false; what it does to registers
// need not care.  18 vector instructions per token for a counting visitor against 22.  Returns false for a dead path.
template <class V, class K, class U>
__device__ __forceinline__ bool chunk_walk_all(const ChunkWin &c, int &off, int &o, V &&visit, K &&keep, U &&undo)
{
	const unsigned d[6] = { (unsigned)c.w0, (unsigned)(c.w0 >> 32), (unsigned)c.w1, (unsigned)(c.w1 >> 32),
		(unsigned)c.w2, (unsigned)(c.w2 >> 32) };
	bool dead = false;
#pragma unroll
	for (int seg = 0; seg < CH_BITS / 32; ++seg) {
		for (;;) {   // (uniform: once, and once more for every lane-token that did not fit its window)
			const int bound = 32 * (seg + 1);
			int len = 0, o_prev = o;
			while (off < bound) {
				o_prev = o;
				keep();
				const unsigned w32 = __builtin_amdgcn_alignbit(d[seg + 1], d[seg], (unsigned)off);
				const int z = __builtin_ctz(w32 | 0x80000000u);
				const int top = o + z;
				len = z + top + 2;
				visit(__builtin_amdgcn_ubfe(w32, (unsigned)(z + 1), (unsigned)top) + bfm((unsigned)z, (unsigned)o),
					__builtin_amdgcn_ubfe(w32, (unsigned)(len - 1), 1u), len <= 32);
				off += len;
				o = (int)__builtin_elementwise_sub_sat((unsigned)top, 2u);
			}
			const bool slow = len > 32;
			if (!ballot64(slow))
				break;
			if (slow) {
				off -= len;
				o = o_prev;
				undo();
				const int r = off & 31;
				const unsigned long long lo64 = d[seg] | ((unsigned long long)d[seg + 1] << 32);
				const unsigned long long w64 = r ? (lo64 >> r) | ((unsigned long long)d[seg + 2] << (64 - r)) : lo64;
				int len64, next;
				unsigned run, neg;
				if (!token_at(w64, o, len64, run, neg, next)) {
					dead = true;
					off = 2 * CH_BITS + 64;   // past every bound: the lane only waits for the others now
				} else {
					visit(run, neg, true);
					off += len64;
					o = next;
				}
			}
		}
	}
	return !dead;
}

// counting only: tokens and symbols (32 bits, saturating: a run is below 2^32, vli.h:86-101, and no segment asks for 2^31
// symbols — a ring has at most 2^30 coefficients — so a chunk at the ceiling is simply never hopped over)
__device__ __forceinline__ bool chunk_count(const ChunkWin &c, int &off, int &o, unsigned &tok, unsigned &sym)
{
	unsigned sym_prev = sym;
	return chunk_walk_all(c, off, o,
		[&](unsigned run, unsigned, bool) {
			sym = __builtin_elementwise_add_sat(sym, run + 1u);
			++tok;
		},
		[&]() { sym_prev = sym; },
		[&]() {
			--tok;
			sym = sym_prev;
		});
}

// inclusive prefix sum over the 64 lanes (DPP row shifts and broadcasts)
__device__ __forceinline__ unsigned wave_incl_add_u(unsigned v)
{
	v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
	v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
	v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
	v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
	v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15
	v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31
	return v;
}

// The walker's own parse of the rest of one chunk: counts tokens and symbols from (off, o) until the
// chunk ends, the next token's run would pass `need` symbols, or a token does not fit a 32-bit
// window (the careful path takes that one).  All 64 lanes of the wave must be active and every
// argument uniform: lane i holds the 32 stream bits from chunk offsets i and 64+i, so the serial
// token chain fetches its window with one v_readlane and otherwise runs on the scalar unit.
struct ChunkScan {
	unsigned tok, sym;
	int off, o;
	// the token at (off, o) is the one whose run outlives the segment (rle.h:95-101 carries it on): its run, the bits of its
	// code without the sign (the one it ends lies in a later segment) and the order after it; cross_bits = 0: not known
	unsigned cross_run;
	int cross_bits, cross_o;
};

__device__ __forceinline__ ChunkScan chunk_scan(const ChunkWin &c, int off, int o, unsigned need)
{
	const int lane = (int)threadIdx.x & 63;
	const unsigned d0 = (unsigned)c.w0, d1 = (unsigned)(c.w0 >> 32), d2 = (unsigned)c.w1, d3 = (unsigned)(c.w1 >> 32),
		d4 = (unsigned)c.w2;
	const bool up = lane >= 32;
	const unsigned winA = __builtin_amdgcn_alignbit(up ? d2 : d1, up ? d1 : d0, (unsigned)lane & 31u);
	const unsigned winB = __builtin_amdgcn_alignbit(up ? d4 : d3, up ? d3 : d2, (unsigned)lane & 31u);
	// First try without any per-token checks and with the serial part cut down to the state chain
	// (b,o) -> (b + 2z + o + 2, max(o + z - 2, 0)): it only needs the zero count at b (one v_readlane) and
	// leaves each token's order in the lane of its offset.  The lanes then work out their tokens' run
	// lengths together and a DPP reduction adds them up.  Offsets only grow, so the loops end whatever
	// the bits are; a token longer than the 32-bit window, a run the 32-bit sum cannot hold or an
	// overshoot of the segment (once per segment) just sends the chunk to the checked loop below.
	{
		const int zA = winA ? __builtin_ctz(winA) : 32, zB = winB ? __builtin_ctz(winB) : 32;
		// (the chain's state is uniform and is told so: left to itself the compiler kept it in vector registers — a
		// v_readfirstlane and a v_readlane per token, 16 instructions of which every one waits for the one before; from
		// scalar registers the chain is one v_readlane and four scalar instructions long)
		// Eleven instructions per token: the state carried is the order plus two (the token's length is 2z + o + 2), and whether
		// every token fits its window is looked at afterwards, by all lanes at once.  (A v_writelane for the order of the
		// token's lane would save two more, but takes its lane through m0, which is not ours to clobber.)
		int ordA = -1, ordB = -1, offf = __builtin_amdgcn_readfirstlane(off), of2 = __builtin_amdgcn_readfirstlane(o) + 2;
		while (offf < 64) {
			const int z = __builtin_amdgcn_readlane(zA, offf);
			ordA = lane == offf ? of2 : ordA;
			const int top2 = of2 + z;
			offf += top2 + z;
			of2 = max(top2 - 2, 2);
		}
		while (offf < 128) {
			const int z = __builtin_amdgcn_readlane(zB, offf - 64);
			ordB = lane == offf - 64 ? of2 : ordB;
			const int top2 = of2 + z;
			offf += top2 + z;
			of2 = max(top2 - 2, 2);
		}
		const int of = of2 - 2;
		ordA = ordA < 0 ? -1 : ordA - 2;
		ordB = ordB < 0 ? -1 : ordB - 2;
		// z + top of the longest token met (tokens of lanes never visited do not count)
		const bool longA = ordA >= 0 && 2 * zA + ordA > 30, longB = ordB >= 0 && 2 * zB + ordB > 30;
		const int worst = ballot64(longA || longB) ? 31 : 0;
		if (worst <= 30) {
			constexpr unsigned CAP = 1u << 24;   // 128 tokens of at most this many symbols cannot overflow
			auto cost = [&](unsigned win, int z, int ord) -> unsigned {
				const int top = ord + z;
				const unsigned run = ((win >> ((z + 1) & 31)) & ((1u << (top & 31)) - 1u)) + (1u << (top & 31)) - (1u << (ord & 31));
				return ord < 0 ? 0u : min(run + 1u, CAP);
			};
			const unsigned cA = cost(winA, zA, ordA), cB = cost(winB, zB, ordB);
			const unsigned long long visA = ballot64(ordA >= 0), visB = ballot64(ordB >= 0);
			const bool capped = ballot64(cA == CAP || cB == CAP) != 0;
			int v = (int)(cA + cB);
			v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);    // quad_perm [1,0,3,2]
			v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);    // quad_perm [2,3,0,1]
			v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);   // row_half_mirror
			v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);   // row_mirror: every lane has its row's sum
			const unsigned symf = (unsigned)(__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) +
				__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
			if (!capped && symf <= need) {
				ChunkScan r = { (unsigned)(__builtin_popcountll(visA) + __builtin_popcountll(visB)), symf, offf, of, 0u, 0, 0 };
				return r;
			}
			if (!capped) {
				// The chunk holds more than the segment still takes (its last chunk, once per segment): the tokens that
				// fit are those whose running sum stays within `need` — two prefix sums over the lanes instead of the
				// checked loop's second, serial walk — and the walk goes on at the first token that does not fit
				// (its offset is its lane, its order is in that lane).
				const unsigned need_s = (unsigned)__builtin_amdgcn_readfirstlane((int)need);
				const unsigned pA = wave_incl_add_u(cA);
				const unsigned totA = (unsigned)__builtin_amdgcn_readlane((int)pA, 63);
				const unsigned pB = totA + wave_incl_add_u(cB);
				const unsigned long long fitA = ballot64(ordA >= 0 && pA <= need_s), fitB = ballot64(ordB >= 0 && pB <= need_s);
				const unsigned long long restA = visA & ~fitA, restB = visB & ~fitB;
				// (the sums only grow: the fitting tokens come first, so fitB is empty unless all of A's fit)
				unsigned sym = 0;
				if (fitB)
					sym = (unsigned)__builtin_amdgcn_readlane((int)pB, 63 - __builtin_clzll(fitB));
				else if (fitA)
					sym = (unsigned)__builtin_amdgcn_readlane((int)pA, 63 - __builtin_clzll(fitA));
				int off2, o2, z2;
				unsigned c2;
				if (restA) {
					const int at = __builtin_ctzll(restA);
					off2 = at;
					o2 = __builtin_amdgcn_readlane(ordA, at);
					z2 = __builtin_amdgcn_readlane(zA, at);
					c2 = (unsigned)__builtin_amdgcn_readlane((int)cA, at);
				} else {   // (restB is not empty: the whole chunk did not fit)
					const int at = __builtin_ctzll(restB);
					off2 = 64 + at;
					o2 = __builtin_amdgcn_readlane(ordB, at);
					z2 = __builtin_amdgcn_readlane(zB, at);
					c2 = (unsigned)__builtin_amdgcn_readlane((int)cB, at);
				}
				// that token is the one that runs past the segment's end (its cost is its run + 1, not capped here)
				const int top2 = o2 + z2;
				ChunkScan r = { (unsigned)(__builtin_popcountll(fitA) + __builtin_popcountll(fitB)), sym, off2, o2, c2 - 1u, z2 + top2 + 1, max(top2, 2) - 2 };
				return r;
			}
		}
	}
	// (the checked loop takes the last chunk of every segment — its tokens run past what the segment needs — so it gets the
	// same treatment: its state is told to be uniform and stays on the scalar unit)
	const unsigned need_s = (unsigned)__builtin_amdgcn_readfirstlane((int)need);
	unsigned tok = 0, left = need_s;   // left = symbols the segment still takes
	off = __builtin_amdgcn_readfirstlane(off);
	o = __builtin_amdgcn_readfirstlane(o);
	// branch-free token step (single-exit loops keep the scalar code tight); false = stop at this token
	auto token = [&](unsigned w32) -> bool {
		const int z = w32 ? __builtin_ctz(w32) : 32;
		const int top = o + z;
		const unsigned run = ((w32 >> ((z + 1) & 31)) & ((1u << (top & 31)) - 1u)) + (1u << (top & 31)) - (1u << o);
		const bool ok = z + top <= 30 && run < left;   // fits the window, and the run stays inside the segment
		left -= ok ? run + 1u : 0u;
		tok += ok ? 1u : 0u;
		off += ok ? z + top + 2 : 0;
		o = ok ? max(top, 2) - 2 : o;
		return ok;
	};
	bool ok = true;
	while (ok && off < 64)
		ok = token((unsigned)__builtin_amdgcn_readlane((int)winA, off));
	while (ok && off < 128)
		ok = token((unsigned)__builtin_amdgcn_readlane((int)winB, off - 64));
	ChunkScan r = { tok, need_s - left, off, o, 0u, 0, 0 };
	return r;
}

// The chunk tables are laid out for the stream stride, the streams are usually much shorter: the chunk
// kernels run a capped grid whose workgroups stride over the virtual blocks of 256 chunks that hold data.
constexpr int CHUNK_GRID = 2048;

// Refinement by relaxation: (re-)parse chunk `ch` from the state its predecessor's recorded path
// currently leaves in; if that moves this chunk's own exit, the successor is queued for the next
// round.  The first round (k_link_first) touches every chunk — in runs of LINK_RUN, one after the other —
// and queues the runs' first chunks whose warm-up had not met the run before; afterwards a percent of
// the chunks, then a few per mille.
constexpr int LINK_SHARDS = 64;   // work lists are sharded: one counter would serialise the appends in L2
constexpr int LINK_RUN = 8;       // chunks a thread of k_link_first parses one after the other (runs start at multiples of it)
constexpr int RUN_W = 2 * LINK_RUN + 2;   // 64-bit LDS words per thread there

// Returns true if the chunk's exit moved (its successor must be re-parsed).
// (Round 4 tried to go on with the successor in the same thread while the exit moves, up to the end of the run of
// LINK_RUN chunks: fewer rounds for the same reach — but reach is what hurts: paths that come out of the raw refinement
// blocks then run 8 times as far into the first-pass stretches and replace records that were right; the walker's hops
// went from 230 to 300 per frame and its time from 13.6 to 21.5 us.)
__device__ __forceinline__ bool link_parse(const DWork &w, const unsigned char *streams, long stream_stride, int vs, long ch)
{
	const int img = vs / FAM;
	const long ci = vs * (w.NCH + 1) + ch;
	const unsigned short in = w.exitX[vs * w.NCH + ch - 1];
	unsigned sym = 0, tok = 0;
	unsigned short out = 0xffff;
	if (in != 0xffff) {
		const ChunkWin c = chunk_load((const unsigned long long *)(streams + img * stream_stride), stream_stride >> 3, ch);
		int off = in & 0xff, o = in >> 8;
		if (chunk_count(c, off, o, tok, sym))
			out = (unsigned short)((off - CH_BITS) | (o << 8));
	}
	const unsigned short old = w.exitX[vs * w.NCH + ch];
	w.entryE[vs * w.NCH + ch] = in;
	w.cs[ci] = sym;
	w.ct[ci] = tok;
	w.exitX[vs * w.NCH + ch] = out;
	return out != old && ch + 1 < w.nch[img];
}

// queue chunk `ch` for the next round; one atomic per (wave, shard) instead of one per lane.
// Every lane of the wave must call this (want = false for lanes with nothing to queue).
__device__ __forceinline__ void link_push(const DWork &w, int vs, long ch, bool want, unsigned *next_list, unsigned *next_count)
{
	const int lane = threadIdx.x & 63;
	const int shard = (int)((ch >> 8) % LINK_SHARDS);
	unsigned long long todo = ballot64(want);
	while (todo) {
		const int leader = __builtin_ctzll(todo);
		const int sh = __shfl(shard, leader);
		const unsigned long long same = ballot64(want && shard == sh) & todo;
		unsigned base = 0;
		if (lane == leader)
			base = atomicAdd(next_count + vs * LINK_SHARDS + sh, (unsigned)__builtin_popcountll(same));
		base = __shfl(base, leader);
		if (want && shard == sh)
			next_list[((long)vs * LINK_SHARDS + sh) * w.todo_cap + base +
				(unsigned)__builtin_popcountll(same & ((1ull << lane) - 1ull))] = (unsigned)ch;
		todo &= ~same;
	}
}

// Speculation and the first relaxation round in one kernel.  Until round 4 a thread parsed ONE chunk twice: from the
// speculative start (the chunk's first bit at order 0) and again from the exit its left-hand neighbour speculated —
// two parses per chunk, of which the first is thrown away.  Now a thread parses a RUN of LINK_RUN chunks one after
// the other, carrying the state along, after one chunk of warm-up (the chunk before its run, from the speculative
// start; nothing of it is recorded): 1 + 1/LINK_RUN parses per chunk, and inside a run every record is made from
// the state its predecessor leaves in by construction — also in the raw refinement blocks (most of a stream's bits),
// where every parse is arbitrary and the one-chunk scheme kept the later rounds busy re-parsing records nobody uses.
// Only a run's first chunk can be wrong (its warm-up had not met the path of the run before): it goes to the work list
// (round 2) when the warm-up's exit is not the exit the run before recorded; a workgroup's first run always does.
// With two families the speculative start is the chunk's middle plus the family's parity: at order 0 every token has
// even length (2z + o + 2), so two parses that start an odd number of bits apart cannot meet while the order stays 0.
// A path that dies (no token this codec writes fits there) starts again from the speculative start in the next chunk;
// the record's entry then differs from its predecessor's exit, which is what flags it (k_scan_local).
// The workgroup's stretch of the stream (32 KB) is staged in LDS with coalesced loads, a thread's run 18 words apart
// (two words of padding: the 64 lanes' 8-byte reads spread over all banks).
// A chunk's record as one word on its way through LDS: symbols (31 bits, saturating: a count no segment can ask for — a
// ring has at most 2^30 coefficients — so such a chunk is simply never hopped over), tokens (7 bits: at most 64 start in
// 128 bits), entry and exit state (13 bits each: offset 7, order 5; all ones = dead).  All ones = no record.
__device__ __forceinline__ unsigned link_state13(unsigned short s) { return s == 0xffff ? 0x1fffu : ((unsigned)s & 0x7fu) | ((unsigned)s >> 8) << 7; }
__device__ __forceinline__ unsigned short link_state16(unsigned s) { return s == 0x1fffu ? (unsigned short)0xffff : (unsigned short)((s & 0x7fu) | (s >> 7) << 8); }
__device__ __forceinline__ unsigned long long link_record(unsigned sym, unsigned tok, unsigned short in, unsigned short out)
{
	const unsigned long long s31 = sym < 0x7fffffffu ? sym : 0x7fffffffu;
	return s31 | (unsigned long long)tok << 31 | (unsigned long long)link_state13(in) << 38 | (unsigned long long)link_state13(out) << 51;
}


__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(64), amdgpu_waves_per_eu(4))) void k_link_first(DWork w, const unsigned char *streams, long stream_stride)
{
	__shared__ unsigned long long words[256 * RUN_W + 4];
	__shared__ unsigned short sx[256];
	const int vs = vstream(w, blockIdx.y), img = vs / FAM;
	const long nch = w.nch[img];
	const unsigned short start = (unsigned short)((w.fam == 1 ? 0 : CH_BITS / 2) + vs % FAM);
	const unsigned long long *w64 = (const unsigned long long *)(streams + img * stream_stride);
	const long n64 = stream_stride >> 3;
	constexpr long SPAN = 256 * LINK_RUN;   // chunks per workgroup and stride
	for (long base = (long)blockIdx.x * SPAN; base < nch; base += (long)gridDim.x * SPAN) {   // uniform
		// words [2 (base - 1), 2 (base + SPAN) + 1) of the stream: the chunk before the stretch, the stretch, one word of look-ahead
		const long g0 = 2 * (base - 1);
#pragma unroll
		for (int k = 0; k < (2 * (int)SPAN + 3 + 255) / 256; ++k) {
			const int L = k * 256 + (int)threadIdx.x;
			const long gi = g0 + L;
			if (L < 2 * (int)SPAN + 3)
				words[L + (L / (2 * LINK_RUN)) * 2] = gi >= 0 && gi < n64 ? w64[gi] : 0ull;
		}
		__syncthreads();
		const long c0 = base + (long)threadIdx.x * LINK_RUN;
		unsigned long long *mine = words + threadIdx.x * RUN_W;
		auto word = [&](int i) { return mine[i + (i >= 2 * LINK_RUN ? 2 : 0)]; };   // (words 2 LINK_RUN.. are the next thread's first)
		unsigned short st = start, wexit = 0xffff;
		unsigned long long carry = word(0);
		for (int j = 0; j <= LINK_RUN; ++j) {   // j = 0: the warm-up chunk
			const long chunk = c0 - 1 + j;
			ChunkWin c = { carry, word(2 * j + 1), word(2 * j + 2) };
			carry = c.w2;
			unsigned long long rec = ~0ull;
			if (chunk >= 0 && chunk < nch) {
				const unsigned short in = j == 0 || chunk == 0 || st == 0xffff ? start : st;
				int off = in & 0xff, o = in >> 8;
				unsigned sym = 0, tok = 0;
				const bool alive = chunk_count(c, off, o, tok, sym);
				st = alive ? (unsigned short)((off - CH_BITS) | (o << 8)) : (unsigned short)0xffff;
				rec = link_record(sym, tok, in, st);
			}
			// The records leave through LDS (one word each, over stream words this thread has consumed and no other thread
			// reads: 3 .. LINK_RUN + 2 of its stretch), so that they go to memory side by side: written from here, a wave's
			// store would touch 64 cache lines, LINK_RUN times over.
			if (j == 0)
				wexit = st;
			else
				mine[2 + j] = rec;
		}
		sx[threadIdx.x] = st;
		__syncthreads();
#pragma unroll
		for (int k = 0; k < LINK_RUN; ++k) {
			const int q = k * 256 + (int)threadIdx.x;
			const long chunk = base + q;
			const unsigned long long rec = words[(q / LINK_RUN) * RUN_W + 3 + q % LINK_RUN];
			if (rec != ~0ull) {   // (a chunk past the stream's end has none)
				if (chunk > 0) {
					w.entryE[vs * w.NCH + chunk] = link_state16((unsigned)(rec >> 38) & 0x1fffu);
					w.cs[vs * (w.NCH + 1) + chunk] = rec & 0x7fffffffull;
					w.ct[vs * (w.NCH + 1) + chunk] = (unsigned)(rec >> 31) & 0x7fu;
				}
				w.exitX[vs * w.NCH + chunk] = link_state16((unsigned)(rec >> 51));
			}
		}
		// the run's first chunk was parsed from the warm-up's exit: right only if that is what the run before recorded as its last exit
		const bool redo = c0 > 0 && c0 < nch && (threadIdx.x == 0 || sx[threadIdx.x - 1] != wexit);
		link_push(w, vs, c0, redo, w.todo[1], w.todo_count + 2 * w.todo_round);
		__syncthreads();   // words and sx are reused by the next stride
	}
}

// later rounds: the chunks queued by the previous one.  Several percent after round 1, then slowly fewer: inside
// the raw refinement blocks (most of a stream's bits) every parse is arbitrary and the paths there keep moving;
// those records are never used, so the rounds are simply cut off (LINK_ROUNDS)
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(64))) void k_link_work(DWork w, const unsigned char *streams, long stream_stride, int cur, int round)
{
	const int vs = vstream(w, blockIdx.y), shard = blockIdx.x % LINK_SHARDS, part = blockIdx.x / LINK_SHARDS,
		parts = gridDim.x / LINK_SHARDS;
	const unsigned count = w.todo_count[round * w.todo_round + vs * LINK_SHARDS + shard];
	const unsigned *list = w.todo[cur] + ((long)vs * LINK_SHARDS + shard) * w.todo_cap;
	for (unsigned q0 = part * blockDim.x; q0 < count; q0 += parts * blockDim.x) {   // uniform trip count per wave
		const unsigned q = q0 + threadIdx.x;
		bool moved = false;
		long ch = 0;
		if (q < count) {
			ch = list[q];
			moved = link_parse(w, streams, stream_stride, vs, ch);
		}
		link_push(w, vs, ch + 1, moved, w.todo[cur ^ 1], w.todo_count + (round + 1) * w.todo_round);
	}
}

// three-kernel exclusive scan of (cs, ct, cg) over the NCH+1 elements of every image
struct Tri {
	unsigned long long s;
	unsigned t, g;
};

__device__ __forceinline__ Tri tri_add(Tri a, Tri b)
{
	Tri r = { a.s + b.s, a.t + b.t, a.g + b.g };
	return r;
}

__device__ __forceinline__ Tri tri_shfl_up(Tri v, int o)
{
	Tri r;
	r.s = __shfl_up(v.s, o);
	r.t = __shfl_up(v.t, o);
	r.g = __shfl_up(v.g, o);
	return r;
}

// block-wide (256 threads) exclusive scan; returns this thread's prefix and the block total
__device__ __forceinline__ Tri block_scan_tri(Tri v, Tri *wsum, Tri &total)
{
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	Tri inc = v;
	for (int o = 1; o < 64; o <<= 1) {
		const Tri t = tri_shfl_up(inc, o);
		if (lane >= o)
			inc = tri_add(inc, t);
	}
	if (lane == 63)
		wsum[wv] = inc;
	__syncthreads();
	Tri off = { 0, 0, 0 }, all = { 0, 0, 0 };
	for (int k = 0; k < 4; ++k) {
		if (k < wv)
			off = tri_add(off, wsum[k]);
		all = tri_add(all, wsum[k]);
	}
	__syncthreads();
	total = all;
	Tri ex = { off.s + inc.s - v.s, off.t + inc.t - v.t, off.g + inc.g - v.g };
	return ex;
}

// Four consecutive elements per thread (NCH+1 is a multiple of 4: whole vectors are in or out)
struct Quad {
	ulonglong2 s01, s23;
	uint4 t, g;
};

__device__ __forceinline__ Quad quad_load(const DWork &w, long at)
{
	Quad q;
	q.s01 = *reinterpret_cast<const ulonglong2 *>(w.cs + at);
	q.s23 = *reinterpret_cast<const ulonglong2 *>(w.cs + at + 2);
	q.t = *reinterpret_cast<const uint4 *>(w.ct + at);
	q.g = *reinterpret_cast<const uint4 *>(w.cg + at);
	return q;
}

__device__ __forceinline__ void quad_store(const DWork &w, long at, const Quad &q)
{
	*reinterpret_cast<ulonglong2 *>(w.cs + at) = q.s01;
	*reinterpret_cast<ulonglong2 *>(w.cs + at + 2) = q.s23;
	*reinterpret_cast<uint4 *>(w.ct + at) = q.t;
	*reinterpret_cast<uint4 *>(w.cg + at) = q.g;
}

// A chunk's record is usable iff it was made from the state its predecessor now leaves in: cg = 1 marks
// the others (never hopped over; their counts are zeroed to keep the prefix sums small and monotone).  Chunk
// 0 has no predecessor; element nch is the scan's sentinel.  The flag is left in entryE for k_scan_add.
constexpr int SCAN_GRID = 1024;

__global__ __launch_bounds__(256) void k_scan_local(DWork w)
{
	__shared__ Tri wsum[4];
	const int vs = vstream(w, blockIdx.y);   // virtual stream
	const long n = w.NCH + 1, nch = w.nch[vs / FAM], used = nch + 1;
	unsigned short *exitX = w.exitX + (long)vs * w.NCH, *entryE = w.entryE + (long)vs * w.NCH;
	for (long vb = blockIdx.x; vb * SCAN_BLOCK < used; vb += gridDim.x) {
		const long i = vb * SCAN_BLOCK + 4 * threadIdx.x;
		const bool in = i < used;
		Quad q;
		q.s01 = q.s23 = make_ulonglong2(0ull, 0ull);
		q.t = q.g = make_uint4(0u, 0u, 0u, 0u);
		if (in) {
			q = quad_load(w, vs * n + i);
			unsigned long long *sv[4] = { &q.s01.x, &q.s01.y, &q.s23.x, &q.s23.y };
			unsigned *tv[4] = { &q.t.x, &q.t.y, &q.t.z, &q.t.w }, *gv[4] = { &q.g.x, &q.g.y, &q.g.z, &q.g.w };
			unsigned short prev = i >= 1 && i - 1 < nch ? exitX[i - 1] : (unsigned short)0xffff;
#pragma unroll
			for (int e = 0; e < 4; ++e) {
				const long c = i + e;
				unsigned flag;
				if (c == 0 || c >= nch) {
					flag = c == 0 ? 1u : 0u;
					*sv[e] = 0;
					*tv[e] = 0;
				} else {
					const unsigned short x = exitX[c];
					const bool bad = prev == 0xffff || entryE[c] != prev || x == 0xffff;
					flag = bad ? 1u : 0u;
					if (bad) {
						*sv[e] = 0;
						*tv[e] = 0;
					}
					prev = x;
				}
				if (c == 0)
					prev = nch > 0 ? exitX[0] : (unsigned short)0xffff;
				*gv[e] = flag;
				if (c < nch)
					entryE[c] = (unsigned short)flag;
			}
		}
		const Tri mine = { q.s01.x + q.s01.y + q.s23.x + q.s23.y, q.t.x + q.t.y + q.t.z + q.t.w, q.g.x + q.g.y + q.g.z + q.g.w };
		Tri total;
		const Tri pre = block_scan_tri(mine, wsum, total);
		if (in) {
			Quad o;
			o.s01.x = pre.s;
			o.s01.y = o.s01.x + q.s01.x;
			o.s23.x = o.s01.y + q.s01.y;
			o.s23.y = o.s23.x + q.s23.x;
			o.t.x = pre.t;
			o.t.y = o.t.x + q.t.x;
			o.t.z = o.t.y + q.t.y;
			o.t.w = o.t.z + q.t.z;
			o.g.x = pre.g;
			o.g.y = o.g.x + q.g.x;
			o.g.z = o.g.y + q.g.y;
			o.g.w = o.g.z + q.g.z;
			quad_store(w, vs * n + i, o);
		}
		if (threadIdx.x == 0) {
			w.part_s[vs * w.NB + vb] = total.s;
			w.part_t[vs * w.NB + vb] = total.t;
			w.part_g[vs * w.NB + vb] = total.g;
		}
		__syncthreads();   // wsum is reused by the next virtual block
	}
}

__global__ __launch_bounds__(256) void k_scan_parts(DWork w)
{
	__shared__ Tri wsum[4];
	const int img = vstream(w, blockIdx.x);   // virtual stream
	const long nb = (w.nch[img / FAM] + 1 + SCAN_BLOCK - 1) / SCAN_BLOCK;   // blocks k_scan_local ran
	Tri carry = { 0, 0, 0 };
	for (long b0 = 0; b0 < nb; b0 += 256) {
		const long i = b0 + threadIdx.x;
		Tri v = { 0, 0, 0 };
		if (i < nb) {
			v.s = w.part_s[img * w.NB + i];
			v.t = w.part_t[img * w.NB + i];
			v.g = w.part_g[img * w.NB + i];
		}
		Tri total;
		const Tri pre = block_scan_tri(v, wsum, total);
		if (i < nb) {
			w.part_s[img * w.NB + i] = carry.s + pre.s;
			w.part_t[img * w.NB + i] = carry.t + pre.t;
			w.part_g[img * w.NB + i] = carry.g + pre.g;
		}
		carry = tri_add(carry, total);
	}
}

// adds the block offsets and files the unjoined chunks: the one with rank r (= cg[i], exclusive prefix) goes to breaks[r]
__global__ __launch_bounds__(256) void k_scan_add(DWork w)
{
	const int vs = vstream(w, blockIdx.y);   // virtual stream
	const long n = w.NCH + 1, nch = w.nch[vs / FAM], used = nch + 1;
	const unsigned short *flags = w.entryE + (long)vs * w.NCH;
	unsigned *breaks = w.breaks + (long)vs * w.NCH;
	for (long vb = blockIdx.x; vb * SCAN_BLOCK < used; vb += gridDim.x) {
		const long i = vb * SCAN_BLOCK + 4 * threadIdx.x;
		if (i >= used)
			continue;
		const unsigned long long ps = w.part_s[vs * w.NB + vb];
		const unsigned pt = w.part_t[vs * w.NB + vb], pg = w.part_g[vs * w.NB + vb];
		Quad q = quad_load(w, vs * n + i);
		q.s01.x += ps;
		q.s01.y += ps;
		q.s23.x += ps;
		q.s23.y += ps;
		q.t.x += pt;
		q.t.y += pt;
		q.t.z += pt;
		q.t.w += pt;
		q.g.x += pg;
		q.g.y += pg;
		q.g.z += pg;
		q.g.w += pg;
		if (vb)   // the first block's prefixes are final already
			quad_store(w, vs * n + i, q);
		const unsigned g4[4] = { q.g.x, q.g.y, q.g.z, q.g.w };
#pragma unroll
		for (int e = 0; e < 4; ++e)
			if (i + e < nch && flags[i + e])
				breaks[g4[e]] = (unsigned)(i + e);
	}
}

// the tokens of every chunk (piece) the walker did not set itself -> symbits
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8))) void k_hopbits(DWork w, const unsigned char *streams, long stream_stride)
{
	constexpr int HB_WORDS = 16;   // 256 symbols
	constexpr int HB_WIN = 2048;   // the workgroup's window: 32768 symbols
	// One stretch of LDS serves both ways of gathering bits (per-thread rows / one window for the workgroup) and is
	// all zeros whenever a way starts: both hand back what they used as zeros (17 instead of 25 KB per workgroup:
	// eight waves per SIMD instead of six).
	__shared__ unsigned hb[256 * (HB_WORDS + 1)];
	static_assert(HB_WIN <= 256 * (HB_WORDS + 1), "the workgroup's window lies inside the per-thread rows");
	unsigned *win = hb;
	__shared__ unsigned win_last;
	for (int i = 0; i <= HB_WORDS; ++i)
		hb[threadIdx.x * (HB_WORDS + 1) + i] = 0u;   // each thread only ever touches its own row
	const int img = blockIdx.y;
	const int nh = w.nhops[img];
	const long nchunks = w.nch[img];
	const unsigned *hf = w.hop_first + (long)img * w.MAX_HOPS, *hl = w.hop_last + (long)img * w.MAX_HOPS;
	for (long vblock = blockIdx.x; vblock * blockDim.x < nchunks; vblock += gridDim.x) {
	const long chunk = vblock * blockDim.x + threadIdx.x;
	// records are in stream order; the search is done once per workgroup (uniform, scalar loads) for
	// its first chunk, every thread then steps forward to its own chunk
	const unsigned first_chunk = (unsigned)(vblock * blockDim.x);
	int lo = 0, hi = nh;   // first h with hl[h] >= first_chunk
	while (lo < hi) {
		const int mid = (lo + hi) >> 1;
		if (hl[mid] < first_chunk)
			lo = mid + 1;
		else
			hi = mid;
	}
	unsigned *sym = w.symbits + img * w.BW;
	const long n = w.NCH + 1;
	ChunkWin c;
	bool loaded = false;
	unsigned *mybuf = hb + threadIdx.x * (HB_WORDS + 1);
	// the tokens record h accounts for in this thread's chunk
	auto piece = [&](int h) {
		const int hs = w.hop_seg[(long)img * w.MAX_HOPS + h];
		const int k = hs & 0xffff, vs = img * FAM + (hs >> 16);
		if ((w.seg_desc[(long)img * MAX_SEGS + k] >> 8) == 0)
			return;   // plane -1 (flat image): symbols carry no bits
		if (!loaded) {
			c = chunk_load((const unsigned long long *)(streams + img * stream_stride), stream_stride >> 3, chunk);
			loaded = true;
		}
		unsigned long long pos = w.seg_symbase[(long)img * MAX_SEGS + k] + w.hop_q0[(long)img * w.MAX_HOPS + h];
		const unsigned entry = w.hop_entry[(long)img * w.MAX_HOPS + h];
		unsigned left = w.hop_ntok[(long)img * w.MAX_HOPS + h];
		int off, o;
		if (entry == 0xffffffffu) {
			pos += w.cs[vs * n + chunk] - w.cs[vs * n + hf[h]];
			const unsigned short in = w.exitX[vs * w.NCH + chunk - 1];
			off = in & 0xff;
			o = in >> 8;
		} else {
			off = (int)(entry & 0xff);
			o = (int)(entry >> 8);
		}
		// The piece's ones go to a window of HB_WORDS bitmap words that the thread keeps in LDS (a chunk of dense
		// tokens spans about half of it; ones beyond the window — long zero runs — go to memory directly).  The
		// words strictly inside the piece's symbol range belong to it alone (neighbouring pieces and the walker's
		// own tokens can only share its first and last word): plain stores, only those two need atomics.
		// A piece stays inside its segment (< 2^28 symbols): positions are 32-bit offsets from its first word.
		unsigned *wp = sym + (pos >> 4);
		const unsigned rp0 = (unsigned)(pos & 15);
		unsigned rp = rp0;
		bool beyond = false;   // some one fell outside the LDS window
		const int off0 = off, o0 = o;
		// (branch-free visitors, see chunk_walk: a token that does not count ORs a zero into the row; a stitched chunk's
		// `left` of 0xffffffff never runs out: all its tokens count)
		chunk_walk(c, off, o, [&](unsigned run, unsigned neg, bool counts) {
			const bool take = counts && left != 0u;
			rp += take ? run : 0u;
			const unsigned wi = rp >> 4;
			beyond = beyond || (take && wi >= (unsigned)HB_WORDS);
			atomicOr(&mybuf[wi < (unsigned)HB_WORDS ? wi : (unsigned)HB_WORDS], take ? (1u | (neg << 1)) << ((rp & 15u) * 2u) : 0u);
			rp += take ? 1u : 0u;
			left -= take ? 1u : 0u;
			return left != 0u || !counts;   // (stop at the first token that is no longer this piece's)
		});
		if (beyond) {   // rare (long zero runs): the ones beyond the window go to memory one by one
			unsigned rq = rp0, lq = w.hop_ntok[(long)img * w.MAX_HOPS + h];
			int off1 = off0, o1 = o0;
			chunk_walk(c, off1, o1, [&](unsigned run, unsigned neg, bool counts) {
				const bool take = counts && lq != 0u;
				rq += take ? run : 0u;
				if (take && (rq >> 4) >= (unsigned)HB_WORDS)
					atomicOr(wp + (rq >> 4), (1u | (neg << 1)) << ((rq & 15u) * 2u));
				rq += take ? 1u : 0u;
				lq -= take ? 1u : 0u;
				return lq != 0u || !counts;
			});
			mybuf[HB_WORDS] = 0u;
		}
		const unsigned lastw = rp > rp0 ? (rp - 1) >> 4 : 0u;
#pragma unroll
		for (int i = 0; i < HB_WORDS; ++i) {
			const unsigned v = mybuf[i];
			if (v) {
				if (i == 0 || (unsigned)i >= lastw)
					atomicOr(wp + i, v);
				else
					wp[i] = v;
				mybuf[i] = 0u;
			}
		}
	};
	const bool mine = chunk < w.nch[img] && chunk >= 1;
	// Most workgroups lie inside one long hop: its record is then the same for every thread and is
	// fetched through the scalar unit; only the chunk's own table rows are per-thread loads.
	if (lo < nh && hf[lo] <= first_chunk && hl[lo] >= first_chunk + blockDim.x - 1 && (lo + 1 >= nh || hf[lo + 1] > first_chunk + blockDim.x - 1)) {
		const int hs = w.hop_seg[(long)img * w.MAX_HOPS + lo];
		const int k = hs & 0xffff, vs = img * FAM + (hs >> 16);
		const bool stitched = w.hop_entry[(long)img * w.MAX_HOPS + lo] == 0xffffffffu && first_chunk >= 1;
		if (!stitched) {
			if (mine)
				piece(lo);
			continue;
		}
		if ((w.seg_desc[(long)img * MAX_SEGS + k] >> 8) == 0)
			continue;   // plane -1 (flat image): symbols carry no bits
		// All 256 chunks lie inside one stitched run: their symbols are one gapless stretch of the segment.  The
		// workgroup gathers the ones in an LDS window over that stretch (dense planes: ~100 symbols per chunk) and
		// writes it out as whole words; everything strictly inside the stretch is this workgroup's alone, its
		// first and last word are shared with the neighbours (atomics).  Ones beyond the window (long zero runs)
		// go to memory directly.
		const unsigned long long seg0 = w.seg_symbase[(long)img * MAX_SEGS + k] + w.hop_q0[(long)img * w.MAX_HOPS + lo] -
			w.cs[vs * n + hf[lo]];
		const unsigned long long base_w = (seg0 + w.cs[vs * n + first_chunk]) >> 4;   // uniform
		if (threadIdx.x == 0)
			win_last = 0u;
		__syncthreads();
		{
			const ChunkWin cw = chunk_load((const unsigned long long *)(streams + img * stream_stride), stream_stride >> 3, chunk);
			const unsigned short in = w.exitX[vs * w.NCH + chunk - 1];
			int off = in & 0xff, o = in >> 8;
			unsigned roff = (unsigned)(seg0 + w.cs[vs * n + chunk] - (base_w << 4));   // symbols from the window's first one
			const unsigned roff0 = roff;
			unsigned *gp = sym + base_w;
			unsigned roff_prev = roff;
			chunk_walk_all(cw, off, o,
				[&](unsigned run, unsigned neg, bool fits) {
					roff += run;
					const unsigned wi = roff >> 4;
					const unsigned bits = fits ? (1u | (neg << 1)) << ((roff & 15u) * 2u) : 0u;
					if (wi < (unsigned)HB_WIN)
						atomicOr(&win[wi], bits);
					else if (fits)
						atomicOr(gp + wi, bits);
					++roff;
				},
				[&]() { roff_prev = roff; }, [&]() { roff = roff_prev; });
			if (roff > roff0) {
				const unsigned lw = (roff - 1) >> 4;
				atomicMax(&win_last, lw < (unsigned)HB_WIN - 1u ? lw : (unsigned)HB_WIN - 1u);
			}
		}
		__syncthreads();
		{
			const unsigned lastw = win_last;
			unsigned *gp = sym + base_w;
			for (unsigned i = threadIdx.x; i <= lastw; i += 256) {
				const unsigned v = win[i];
				win[i] = 0u;
				if (i > 0 && i < lastw)
					gp[i] = v;
				else if (v)
					atomicOr(gp + i, v);
			}
		}
		__syncthreads();   // the window is reused by the next virtual block
		continue;
	}
	if (!mine)
		continue;
	int h = lo;
	while (h < nh && hl[h] < (unsigned)chunk)
		++h;
	for (; h < nh && hf[h] <= (unsigned)chunk; ++h)
		piece(h);
	}
}

// A part of the batch is walked again (with both families): what the first attempt left behind and the call's
// initial clears covered goes back to zero — the work-list counters of every round, the hop count, the walker's
// per-image records.  (The symbol bitmap is cleared beside this.)
__global__ __launch_bounds__(256) void k_part_reset(DWork h, int cnt)
{
	const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
	const long per_round = (long)cnt * FAM * LINK_SHARDS;
	if (t < per_round)
		for (int r = 0; r < LINK_ROUNDS + 2; ++r)
			h.todo_count[r * h.todo_round + t] = 0u;
	if (t < cnt)
		h.nhops[t] = 0;
	if (t < (long)cnt * 48 * MAX_PLANES)
		h.segidx[t] = 0;
	if (t < (long)cnt * (long)(sizeof(DecInfo) / sizeof(int)))
		reinterpret_cast<int *>(h.info)[t] = 0;
}

// decode.c:142-159 header, decode.c:119-134 root image (channel c's coefficients go to root + c * root_stride if
// root is given), decode.c:183-186 plane counts.  0: read; 1: unreadable (decode.c exits 1); 2: a plane count above
// MAX_PLANES — only damage can produce one (8-bit sources stay below 12): refused, where the reference goes on and
// decodes garbage (decode.c:183-186 accepts any get_vli() value) — the one documented difference, DESIGN.md section 7.
__device__ int read_preamble(BitReader &br, const UnpackGeom &g, const unsigned char *s8, unsigned long long len, long stream_stride,
	int &order, int (&planes)[3], int *root, long root_stride)
{
	if (len < 6 || s8[0] != 'W' || s8[1] != (g.C == 3 ? '6' : '5') ||
		(s8[2] | (s8[3] << 8)) + 1 != g.W || (s8[4] | (s8[5] << 8)) + 1 != g.H)
		return 1;
	br.w = (const unsigned long long *)s8;
	br.n64 = stream_stride >> 3;
	br.end_bits = len * 8;
	br.seek(48);
	order = 0;
	for (int c = 0; c < g.C; ++c) {
		unsigned cnt;
		if (br.vli_any(order, cnt))
			return 1;
		if (cnt)
			for (int i = 0; i < g.pixels[0]; ++i) {
				unsigned v, neg = 0;
				if (!br.read_any(cnt, v))
					return 1;
				if (v && !br.read(1, neg))
					return 1;
				if (root)
					root[c * root_stride + i] = neg ? -(int)v : (int)v;
			}
	}
	for (int c = 0; c < g.C; ++c) {
		unsigned p;
		if (br.vli_any(order, p))
			return 1;
		if (p > MAX_PLANES)
			return 2;
		planes[c] = (int)p;
	}
	return 0;
}

// How much of its symbol bitmap an image can use follows from its plane counts: every segment owns ceil32(ring
// size) symbols, a (channel, level) has as many segments as the channel has planes (plus the flat image's
// "plane -1" segment on level 0).  Clearing only that much — 8-bit pictures have 8 to 11 planes of the 16 the
// bitmap is laid out for — takes 40 % off the one big clear of the decoder.
__global__ __launch_bounds__(64) void k_peek(UnpackGeom g, const unsigned char *streams, long stream_stride, const unsigned long long *lens,
	long BW, unsigned *clear_words, int n)
{
	const int img = blockIdx.x * blockDim.x + threadIdx.x;
	if (img >= n)
		return;
	const unsigned char *s8 = streams + img * stream_stride;
	const unsigned long long len = lens[img] < (unsigned long long)stream_stride ? lens[img] : (unsigned long long)stream_stride;
	BitReader br;
	int order, planes[3] = { 0, 0, 0 };
	unsigned long long words = 0;
	if (!read_preamble(br, g, s8, len, stream_stride, order, planes, nullptr, 0)) {
		unsigned long long per = 0;
		for (int l = 0; l < g.levels; ++l)
			per += ((unsigned long long)(g.pixels[l + 1] - g.pixels[l]) + 31) & ~31ull;
		unsigned long long sym = ((unsigned long long)(g.pixels[1] - g.pixels[0]) + 31) & ~31ull;
		for (int c = 0; c < g.C; ++c)
			sym += (unsigned long long)planes[c] * per;
		words = ((sym >> 4) + 128 + 3) & ~3ull;   // the same slack as BW has
	}
	clear_words[img] = (unsigned)(words < (unsigned long long)BW ? words : (unsigned long long)BW);
}

__global__ __launch_bounds__(256) void k_clear_bitmaps(unsigned *bits, long BW, const unsigned *clear_words)
{
	uint4 *dst = reinterpret_cast<uint4 *>(bits + (long)blockIdx.y * BW);
	const unsigned quads = clear_words[blockIdx.y] >> 2;
	for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += gridDim.x * blockDim.x)
		dst[i] = make_uint4(0u, 0u, 0u, 0u);
}

// --------------------------------------------------------------- k_tokenize ---

// decode.c:198-243: the order in which (channel, level, plane) segments follow one another.  seg(c, l, p) decodes
// one and says whether decoding goes on; `level` and `missing` (through one_less_missing(channel * 16 + level): the caller
// says where its counters live) are kept as decode.c:197,203,219,236 keeps them.
// Returns whether the walk stopped before the schedule's end (decode.c:199-200,204,221,238).
template <class F, class M>
__device__ __forceinline__ bool walk_schedule(const UnpackGeom &g, const int (&planes)[3], int pmax, int &level, M &&one_less_missing, F &&seg)
{
	const int levels = g.levels;
	const int layers_max = 2 * (levels > pmax ? levels : pmax) - 1;
	bool stop = g.levels_max == 0;        // decode.c:199-200
	if (!stop && pmax == planes[0]) {     // decode.c:201-207
		level = 0;
		if (seg(0, 0, planes[0] - 1))
			one_less_missing(0);
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
			if (seg(0, l, p))
				one_less_missing(l);
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
				if (seg(c, l, p))
					one_less_missing(c * 16 + l);
				else
					stop = true;
			}
		}
	}
	return stop;
}

// Hand-parsed chunks in a row / in all (plus 16 per segment and 1 per 512 chunks) before the one-family walk gives up.  A chunk
// parsed by hand costs the walk 1.7 us, the second walk of a 4096x4096 frame 1.6 ms: patience up to a few hundred chunks in a row is
// cheaper than giving up (round 4: two of the benchmark's first 256 synthetic frames have a stretch of 49..128 such chunks and
// took the second walk under the earlier limit of 48; tools/walk_patience.py).
constexpr unsigned WALK_STREAK_MAX = 256u, WALK_SCANS_BASE = 256u;
constexpr unsigned WALK_GAVE_UP = 0xffffffffu;   // DecInfo::hops of an image whose one-family walk was abandoned

__device__ __forceinline__ void info_begin(DecInfo &I, const UnpackGeom &g)
{
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
}

// SEG = false: one wave per image walks the whole stream (grid: images).
// SEG = true: one wave per segment (grid: MAX_SEGS x images) starts from the state the sidecar index gives for it;
// k_segjoin afterwards checks that the segments fit together and puts their hop records in a row.
template <bool SEG>
__global__ __launch_bounds__(64) void k_tokenize(UnpackGeom g, DWork w, const unsigned char *streams, long stream_stride,
	const unsigned long long *lens, int *lin, int n)
{
	const int img = SEG ? blockIdx.y : blockIdx.x;
	if (img >= n)
		return;
	const int seg_k = SEG ? (int)blockIdx.x : 0;
	if (SEG && seg_k >= w.idx_nsegs[img])
		return;
	// All 64 lanes walk in step on uniform values (the compiler keeps them on the scalar unit), so that
	// chunk_scan can use the lanes; plain stores just repeat the same value, atomics are lane 0's.
	DecInfo &I = w.info[img];
	const unsigned char *s8 = streams + img * stream_stride;
	// a length beyond the stride cannot be real data: only the bytes inside the stride are read
	const unsigned long long len = lens[img] < (unsigned long long)stream_stride ? lens[img] : (unsigned long long)stream_stride;
	BitReader br;
	int order = 0;   // vli.h:24
	int planes[3] = { 0, 0, 0 };
	int pmax = 0;
	int nonsig_own[1] = { 0 };   // SEG: the one counter this segment needs
	// The serial walk's 48 counters (channel x level) live in the lanes of one register: read with v_readlane, written with a
	// select.  (In memory — until round 4 — every segment began by waiting for a load of its counter: 70 round trips per frame.)
	int nonsig_v = 0;
	SegIndex *idx = w.idx + (long)img * MAX_SEGS;
	if (!SEG) {
		info_begin(I, g);
		if (const int bad = read_preamble(br, g, s8, len, stream_stride, order, planes, lin + (long)img * g.C * g.lin_stride, g.lin_stride)) {
			I.status = bad;
			return;
		}
		for (int c = 0; c < g.C; ++c) {
			I.planes[c] = planes[c];
			pmax = planes[c] > pmax ? planes[c] : pmax;
		}
		I.pmax = pmax;
		I.status = 0;
		for (int c = 0; c < g.C; ++c)
			for (int l = 0; l < g.levels; ++l)
				I.missing[c * 16 + l] = planes[c];
		{
			const int lv = (int)threadIdx.x & 15, ch = (int)threadIdx.x >> 4;   // this lane's (channel, level)
			nonsig_v = ch < g.C && lv < g.levels ? g.pixels[lv + 1] - g.pixels[lv] : 0;
		}
	} else {
		br.w = (const unsigned long long *)s8;
		br.n64 = stream_stride >> 3;
		br.end_bits = len * 8;
		br.b = idx[seg_k].bit < br.end_bits ? idx[seg_k].bit : br.end_bits;
		br.base = 0;
		br.f0 = br.f1 = br.f2 = br.f3 = br.f4 = br.f5 = 0;
		order = (int)(idx[seg_k].order & 31u);
		nonsig_own[0] = (int)idx[seg_k].n1;
	}
	int *sd = w.seg_desc + (long)img * MAX_SEGS;
	unsigned long long *ssym = w.seg_symbase + (long)img * MAX_SEGS;
	unsigned long long *sb2 = w.seg_b2 + (long)img * MAX_SEGS;
	unsigned *sn2 = w.seg_n2done + (long)img * MAX_SEGS;
	int *sidx = w.segidx + (long)img * 3 * 16 * MAX_PLANES;
	BitmapWriter bm;
	bm.sym = w.symbits + img * w.BW;
	bm.cur = -1;
	bm.acc = 0;
	bm.lead = threadIdx.x == 0;
	const int wl = (int)threadIdx.x;   // lane, for the few places where the lanes share work

	unsigned cnt = SEG ? idx[seg_k].cnt : 0u;              // rle.h:25
	unsigned long long symtotal = SEG ? idx[seg_k].sym_base : 0ull;
	int nsegs = seg_k, level = -1;

	// stitched-chunk tables of this stream (k_link_* / k_scan_*)
	const unsigned short *exitX0 = w.exitX + (long)img * FAM * w.NCH;
	const unsigned long long *CS0 = w.cs + (long)img * FAM * (w.NCH + 1);
	const unsigned *CT0 = w.ct + (long)img * FAM * (w.NCH + 1), *CG0 = w.cg + (long)img * FAM * (w.NCH + 1);
	const unsigned *BR0 = w.breaks + (long)img * FAM * w.NCH;
	const unsigned long long *s64 = (const unsigned long long *)s8;
	// chunk i is safe to parse blindly if every token starting in it ends inside the data
	const long lastsafe = br.end_bits >= 64 + CH_BITS ? (long)((br.end_bits - 64) >> CH_LOG2) - 1 : -1;
#ifdef DWTX_DEBUG_HOOKS   // cycle counters for tools/dbg_walker.py (s_memtime waits on the scalar memory counter: not in the product)
	unsigned long long t_hop = 0, t_fast = 0, t_load = 0, t_scan = 0, t_care = 0, n_care = 0, t_all0 = __builtin_readcyclecounter(), t_mark = 0;
#define WALK_MARK() t_mark = __builtin_readcyclecounter()
#define WALK_ADD(acc) do { const unsigned long long t_now = __builtin_readcyclecounter(); acc += t_now - t_mark; t_mark = t_now; } while (0)
#else
#define WALK_MARK()
#define WALK_ADD(acc)
#endif
	// With one family the walk can meet a stretch that the recorded paths do not follow at all (k_link_first): it gives
	// up when the chunks it parses by hand pile up, and the host repeats the part with both families.
	unsigned scans = 0, streak = 0;
	bool giveup = false;
	long checked = -1;
	int nhops = 0;
	unsigned hopped = 0, walked = 0;
	bool br_synced = !SEG;   // br's look-ahead registers match br.b
	bool eof = false;        // a read ran past the end of the data (the reference prints bytes.h:101 once)
	// SEG: the segment's records go to a stretch of the image's list that is its own (k_segprep sized it)
	const long hop0 = (long)img * w.MAX_HOPS + (SEG ? (long)w.seg_slot[(long)img * (MAX_SEGS + 1) + seg_k] : 0l);
	const long hop_cap = SEG ? (long)(w.seg_slot[(long)img * (MAX_SEGS + 1) + seg_k + 1] - w.seg_slot[(long)img * (MAX_SEGS + 1) + seg_k]) : w.MAX_HOPS;
	int *hop_seg = w.hop_seg + hop0;
	unsigned *hop_first = w.hop_first + hop0, *hop_last = w.hop_last + hop0, *hop_q0 = w.hop_q0 + hop0, *hop_entry = w.hop_entry + hop0,
		*hop_ntok = w.hop_ntok + hop0;

	// decode.c:67-100 without touching coefficients; false = stop decoding (decode.c:204,221,238)
	auto segment = [&](int c, int l, int p) -> bool {
		const int num = g.pixels[l + 1] - g.pixels[l];
		// coefficients of this (channel, level) still insignificant
		const int n1 = p < 0 ? num : SEG ? nonsig_own[0] : __builtin_amdgcn_readlane(nonsig_v, c * 16 + l);
		const int n2 = num - n1;
		const int k = nsegs++;
		const unsigned long long sym0 = symtotal;
		if (!SEG) {   // what a wave of its own would have to know to start here (the sidecar index)
			SegIndex e;
			e.bit = br.b;
			e.sym_base = sym0;
			e.n1 = (unsigned)n1;
			e.cnt = cnt;
			e.desc = (unsigned)(c | (l << 4) | ((p + 1) << 8));
			e.order = (unsigned)order;
			idx[k] = e;
		}
		symtotal += ((unsigned long long)num + 31) & ~31ull;
		sd[k] = c | (l << 4) | ((p + 1) << 8);
		ssym[k] = sym0;
		sb2[k] = 0;
		sn2[k] = 0;
		if (p >= 0)
			sidx[(c * 16 + l) * MAX_PLANES + p] = k + 1;
		int q = 0, ones = 0;
		bool ok = true;
		streak = 0;
		while (q < n1) {
			unsigned zr;
			if (cnt == 0) {   // rle.h:70-75: a token starts here
				const long ci = (long)(br.b >> CH_LOG2);
				if (ci >= 1 && ci <= lastsafe && nhops < hop_cap) {
					const unsigned need = (unsigned)(n1 - q);
					const int rel = (int)(br.b - ((unsigned long long)ci << CH_LOG2));
					bool moved = false;
					WALK_MARK();
					if (ci != checked) {
						checked = ci;
						for (int fam = 0; fam < w.fam && !moved; ++fam) {
							const unsigned ep = exitX0[fam * w.NCH + ci - 1];
							if (ep == 0xffffu || (int)(ep & 0xffu) != rel || (int)(ep >> 8) != order)
								continue;
							// We stand exactly where this family's recorded path leaves chunk ci-1.  If chunk ci's
							// record was made from that very state (it is not flagged), the stream IS the recorded
							// path from here up to the next flagged chunk `nb`: take all of it if the segment
							// needs that many symbols, else binary-search the prefix sums for the last chunk
							// that still fits.
							const unsigned long long *CS = CS0 + fam * (w.NCH + 1);
							const unsigned *CT = CT0 + fam * (w.NCH + 1), *CG = CG0 + fam * (w.NCH + 1);
							const unsigned g0 = CG[ci], g1 = CG[ci + 1], gtot = CG[w.nch[img]];
							if (g1 != g0)
								continue;   // chunk ci itself was recorded from another entry state
							const unsigned long long s0 = CS[ci];
							long hi = g0 < gtot ? (long)BR0[fam * w.NCH + g0] - 1 : lastsafe;
							hi = hi < lastsafe ? hi : lastsafe;
							long lo = ci - 1;
							if (hi >= ci && CS[hi + 1] - s0 <= (unsigned long long)need) {
								lo = hi;
							} else {
								// 64-way search: lane j probes lo + (j+1)*stride; the predicate is true up to the answer
								hi = hi - 1;
								while (lo < hi) {
									const long span = hi - lo, stride = (span + 63) >> 6;
									const long step = (long)(wl + 1) * stride;
									const long probe = lo + (step < span ? step : span);
									const int t = __builtin_popcountll(ballot64(CS[probe + 1] - s0 <= (unsigned long long)need));
									const long reach = (long)t * stride, next = (long)(t + 1) * stride;
									hi = t == 64 ? hi : lo + (next < span ? next : span) - 1;
									lo = lo + (reach < span ? reach : span);
								}
							}
							if (lo >= ci) {
								hop_seg[nhops] = k | (fam << 16);
								hop_first[nhops] = (unsigned)ci;
								hop_last[nhops] = (unsigned)lo;
								hop_q0[nhops] = (unsigned)q;
								hop_entry[nhops] = 0xffffffffu;
								hop_ntok[nhops] = 0xffffffffu;
								++nhops;
								hopped += (unsigned)(lo - ci + 1);
								q += (int)(CS[lo + 1] - s0);
								ones += (int)(CT[lo + 1] - CT[ci]);
								const unsigned eq = exitX0[fam * w.NCH + lo];
								order = (int)(eq >> 8);
								br.b = ((unsigned long long)(lo + 1) << CH_LOG2) + (eq & 0xffu);
								br_synced = false;
								moved = true;
							}
						}
					}
					WALK_ADD(t_hop);
					if (moved)
						streak = 0;
					if (!moved) {
						// parse the rest of this chunk ourselves, counting only; k_hopbits sets the bits later
						++scans;
						++streak;
						if (w.fam < FAM && (streak > w.streak_max || scans > w.scans_base + 16u * (unsigned)nsegs + (unsigned)(lastsafe >> 9))) {
							giveup = true;
							return false;
						}
						const ChunkWin cwv = chunk_load(s64, br.n64, ci);
#ifdef DWTX_DEBUG_HOOKS
						asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
						WALK_ADD(t_load);
#endif
						const ChunkScan cs = chunk_scan(cwv, rel, order, need);
#ifdef DWTX_DEBUG_HOOKS
						WALK_ADD(t_scan);
#endif
						const unsigned tok = cs.tok, sym = cs.sym;
						const int off = cs.off, o = cs.o;
						if (tok) {
							hop_seg[nhops] = k;
							hop_first[nhops] = (unsigned)ci;
							hop_last[nhops] = (unsigned)ci;
							hop_q0[nhops] = (unsigned)q;
							hop_entry[nhops] = (unsigned)rel | ((unsigned)order << 8);
							hop_ntok[nhops] = tok;
							++nhops;
							walked += tok;
							q += (int)sym;
							ones += (int)tok;
							order = o;
							br.b = ((unsigned long long)ci << CH_LOG2) + (unsigned)off;
							br_synced = false;
							moved = true;
						}
						if (cs.cross_bits && q < n1) {   // (q == n1: the tokens that fit end the pass exactly, what follows is not this segment's)
							// The next token's run outlives this segment's first pass, and chunk_scan has it: what the careful path
							// below does with such a token (it reads the code, not the sign: the one that ends the run lies in a
							// later segment), without filling the bit reader again for it.
							const unsigned rem = (unsigned)(n1 - q);
							++walked;
							order = cs.cross_o;
							br.b = ((unsigned long long)ci << CH_LOG2) + (unsigned)off + (unsigned)cs.cross_bits;
							br_synced = false;
							q = n1;
							cnt = cs.cross_run - rem + 1;
							WALK_ADD(t_fast);
							break;
						}
					}
					WALK_ADD(t_fast);
					if (moved)
						continue;
				}
#ifdef DWTX_DEBUG_HOOKS
				WALK_MARK();
				++n_care;
#endif
				if (!br_synced) {
					br.seek(br.b);
					br_synced = true;
				}
				unsigned v;
				const int bad = br.vli_any(order, v);
				if (bad) {
					ok = false;
					eof = bad == 1;
					break;
				}
				++walked;
				zr = v;
#ifdef DWTX_DEBUG_HOOKS
				WALK_ADD(t_care);
#endif
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
			if (!br_synced) {
				br.seek(br.b);
				br_synced = true;
			}
			unsigned neg;
			if (!br.read(1, neg)) {     // magnitude bit stays, sign unknown (decode.c:80-85)
				++q;
				ok = false;
				eof = true;
				break;
			}
			if (neg && p >= 0)
				bm.set_sign(sym0 + (unsigned)q);
			++q;
		}
		if (p >= 0) {
			if (SEG)
				nonsig_own[0] = n1 - ones;
			else
				nonsig_v = wl == c * 16 + l ? n1 - ones : nonsig_v;
		}
		if (!ok)
			return false;
		if (p >= 0 && n2 > 0) {
			if (cnt > 0) {              // rle.h:95-101: a pending run must end exactly here
				if (cnt != 1) {
					--cnt;   // get_rle() has taken one zero before rle_get_bit() gives up
					return false;
				}
				cnt = 0;
			}
			sb2[k] = br.b;
			const unsigned long long av = br.avail();
			if ((unsigned long long)n2 > av) {
				sn2[k] = (unsigned)av;
				br.b = br.end_bits;
				eof = true;
				return false;
			}
			sn2[k] = (unsigned)n2;
			br.b += (unsigned)n2;
			br_synced = false;
		}
		return true;
	};

	bool stop = false;
	if (SEG) {
		const unsigned d = idx[seg_k].desc;
		const int c = (int)(d & 15u), l = (int)((d >> 4) & 15u), p = (int)(d >> 8) - 1;
		// (an index is untrusted input: the bounds are compared without sums that could wrap, and an entry order the
		// serial walk can only reach on a damaged stream — 32 and more — sends the image to the serial walk)
		const unsigned long long ring = (unsigned long long)(g.pixels[l < g.levels ? l + 1 : 1] - g.pixels[l < g.levels ? l : 0]);
		const unsigned long long sym_cap = (unsigned long long)w.BW * 16ull;
		const bool sane = c < g.C && l < g.levels && p < MAX_PLANES && (unsigned long long)idx[seg_k].n1 <= ring && idx[seg_k].order <= 31u &&
			ring + 32ull <= sym_cap && idx[seg_k].sym_base <= sym_cap - ring - 32ull;
		const bool went = sane && segment(c, l, p);
		bm.flush();
		SegResult r;
		r.bit = br.b;
		r.order = (unsigned)order;
		r.cnt = cnt;
		r.ones = sane ? idx[seg_k].n1 - (unsigned)nonsig_own[0] : 0u;
		r.nhops = (unsigned)nhops;
		r.hopped = hopped;
		r.walked = walked;
		r.ok = went && !giveup && !eof ? 1u : 0u;
		r.pad = 0u;
		w.segres[(long)img * MAX_SEGS + seg_k] = r;
		return;
	}
	// (decode.c:193-196's `missing` in the lanes of a register like the counters above: in memory every segment ended with a
	// load, a decrement and a store of its entry)
	int missing_v = wl < 48 && (wl >> 4) < g.C && (wl & 15) < g.levels ? planes[wl >> 4 < 3 ? wl >> 4 : 0] : 0;
	stop = walk_schedule(g, planes, pmax, level, [&](int i) { missing_v = wl == i ? missing_v - 1 : missing_v; }, segment);
	if (wl < 48 && (wl >> 4) < g.C && (wl & 15) < g.levels)
		I.missing[wl] = missing_v;
	bm.flush();
	if (giveup) {   // nothing of this walk is used: no records for k_hopbits, the marker for the host
		w.nhops[img] = 0;
		I.hops = WALK_GAVE_UP;
		return;
	}
	w.nhops[img] = nhops;
	I.hops = (unsigned)nhops;
	I.hopped_chunks = hopped;
	I.walked_tokens = walked;
	I.zeros_left = cnt;
#ifdef DWTX_DEBUG_HOOKS
	if (w.dbg) { w.dbg[img * 8 + 0] = __builtin_readcyclecounter() - t_all0; w.dbg[img * 8 + 1] = t_hop; w.dbg[img * 8 + 2] = t_fast; w.dbg[img * 8 + 3] = scans;
		w.dbg[img * 8 + 4] = t_load; w.dbg[img * 8 + 5] = t_scan; w.dbg[img * 8 + 6] = t_care; w.dbg[img * 8 + 7] = n_care; }
#endif
	I.level = level;
	I.nsegs = nsegs;
	I.truncated = stop ? (eof ? 3 : 1) : 0;
	I.bits_used = br.b;
}

// ------------------------------------------------------- indexed walk: before / after ---
// k_segprep (one thread per image): every segment's private stretch of hop records, 8 + one per eight chunks
// between its start and the next segment's (the serial walk's MAX_HOPS is the sum of exactly these).  An index
// whose positions are not ascending inside the stream is dropped here (idx_nsegs = 0: the host walks serially).
__global__ __launch_bounds__(64) void k_segprep(DWork w, const unsigned long long *lens, long stream_stride, int n)
{
	const int img = blockIdx.x * blockDim.x + threadIdx.x;
	if (img >= n)
		return;
	const int K = w.idx_nsegs[img];
	const SegIndex *idx = w.idx + (long)img * MAX_SEGS;
	unsigned *slot = w.seg_slot + (long)img * (MAX_SEGS + 1);
	const unsigned long long end_bits = 8ull * (lens[img] < (unsigned long long)stream_stride ? lens[img] : (unsigned long long)stream_stride);
	bool good = K > 0 && K <= MAX_SEGS;
	unsigned long long at = 0;
	for (int k = 0; good && k < K; ++k) {
		const unsigned long long b = idx[k].bit, nb = k + 1 < K ? idx[k + 1].bit : end_bits;
		good = b <= nb && nb <= end_bits && idx[k].sym_base < (unsigned long long)w.BW * 16ull && idx[k].order <= 31u;
		slot[k] = (unsigned)at;
		at += 8ull + ((nb - b) >> (CH_LOG2 + 3));
	}
	slot[K > 0 && K <= MAX_SEGS ? K : 0] = (unsigned)at;
	if (!good || at > (unsigned long long)w.MAX_HOPS)
		w.idx_nsegs[img] = 0;
}

// k_segjoin (one wave per image, uniform code): the serial walk's bookkeeping around the segments that
// k_tokenize<true> walked on their own.  It reads the preamble, replays the schedule and checks every link of the
// chain — segment k must be the schedule's k-th segment, start where segment k-1 arrived (position, VLI order, run
// counter), hold the symbols the counters of its (channel, level) say and own the next symbol slots — so an index
// that passes describes exactly the walk the serial kernel would have made.  Then the segments' hop records move
// together into one ascending list.  Anything else: the give-up marker, and the host walks the part serially.
__global__ __launch_bounds__(64) void k_segjoin(UnpackGeom g, DWork w, const unsigned char *streams, long stream_stride,
	const unsigned long long *lens, int *lin, int n)
{
	const int img = blockIdx.x;
	if (img >= n)
		return;
	const int lane = threadIdx.x;
	DecInfo &I = w.info[img];
	info_begin(I, g);
	const unsigned char *s8 = streams + img * stream_stride;
	const unsigned long long len = lens[img] < (unsigned long long)stream_stride ? lens[img] : (unsigned long long)stream_stride;
	BitReader br;
	int order = 0, planes[3] = { 0, 0, 0 };
	if (const int bad = read_preamble(br, g, s8, len, stream_stride, order, planes, lin + (long)img * g.C * g.lin_stride, g.lin_stride)) {
		I.status = bad;   // as from the serial walk
		return;
	}
	int pmax = 0;
	for (int c = 0; c < g.C; ++c) {
		I.planes[c] = planes[c];
		pmax = planes[c] > pmax ? planes[c] : pmax;
	}
	I.pmax = pmax;
	I.status = 0;
	for (int c = 0; c < g.C; ++c)
		for (int l = 0; l < g.levels; ++l)
			I.missing[c * 16 + l] = planes[c];
	const int K = w.idx_nsegs[img];
	const SegIndex *idx = w.idx + (long)img * MAX_SEGS;
	const SegResult *res = w.segres + (long)img * MAX_SEGS;
	int nonsig[48];
	for (int c = 0; c < g.C; ++c)
		for (int l = 0; l < g.levels; ++l)
			nonsig[c * 16 + l] = g.pixels[l + 1] - g.pixels[l];
	unsigned long long at = br.b, symtotal = 0;
	unsigned cnt = 0, hopped = 0, walked = 0;
	int k = 0, level = -1;
	bool good = K > 0 && g.levels_max >= g.levels;
	const bool stop = walk_schedule(g, planes, pmax, level, [&](int i) { --I.missing[i]; }, [&](int c, int l, int p) {
		const int num = g.pixels[l + 1] - g.pixels[l];
		const int n1 = p < 0 ? num : nonsig[c * 16 + l];
		good = good && k < K;
		if (good) {
			const SegIndex e = idx[k];
			const SegResult r = res[k];
			good = e.desc == (unsigned)(c | (l << 4) | ((p + 1) << 8)) && e.bit == at && e.order == (unsigned)order && e.cnt == cnt &&
				e.n1 == (unsigned)n1 && e.sym_base == symtotal && r.ok != 0u && r.ones <= (unsigned)n1;
			if (good) {
				at = r.bit;
				order = (int)r.order;
				cnt = r.cnt;
				if (p >= 0)
					nonsig[c * 16 + l] = n1 - (int)r.ones;
				symtotal += ((unsigned long long)num + 31) & ~31ull;
				hopped += r.hopped;
				walked += r.walked;
				++k;
			}
		}
		return good;
	});
	good = good && !stop && k == K;
	if (!good) {
		w.nhops[img] = 0;
		I.hops = WALK_GAVE_UP;
		return;
	}
	// the hop records of all segments in one row (a segment's stretch never starts before the row's end: forward copy)
	const unsigned *slot = w.seg_slot + (long)img * (MAX_SEGS + 1);
	const long h0 = (long)img * w.MAX_HOPS;
	unsigned total = 0;
	for (int j = 0; j < K; ++j) {
		const unsigned nh = res[j].nhops, from = slot[j];
		if (from != total)
			for (unsigned i0 = 0; i0 < nh; i0 += 64) {
				const unsigned i = i0 + (unsigned)lane;
				int a = 0;
				unsigned b = 0, c2 = 0, d = 0, e = 0, f = 0;
				if (i < nh) {
					a = w.hop_seg[h0 + from + i];
					b = w.hop_first[h0 + from + i];
					c2 = w.hop_last[h0 + from + i];
					d = w.hop_q0[h0 + from + i];
					e = w.hop_entry[h0 + from + i];
					f = w.hop_ntok[h0 + from + i];
				}
				if (i < nh) {
					w.hop_seg[h0 + total + i] = a;
					w.hop_first[h0 + total + i] = b;
					w.hop_last[h0 + total + i] = c2;
					w.hop_q0[h0 + total + i] = d;
					w.hop_entry[h0 + total + i] = e;
					w.hop_ntok[h0 + total + i] = f;
				}
			}
		total += nh;
	}
	w.nhops[img] = (int)total;
	I.hops = total;
	I.hopped_chunks = hopped;
	I.walked_tokens = walked;
	I.zeros_left = cnt;
	I.level = level;
	I.nsegs = K;
	I.truncated = 0;
	I.bits_used = at;
}

// ------------------------------------------------------------------ k_rank ---
// Which pass-1 symbol a coefficient is depends on how many coefficients before
// it are still insignificant.  Per tile that count only needs the symbol bitmaps:
// for planes descending, k_rank scans the tiles' insignificant counts of a ring
// (-> first symbol index of every tile at this plane) and k_count subtracts the
// ones the plane finds in each tile.  No coefficient is touched yet.

__global__ __launch_bounds__(1024) void k_rank(UnpackGeom g, DWork w, int p)
{
	__shared__ unsigned wsum[16];
	__shared__ unsigned carry;
	const int l = blockIdx.x;
	const int plane = blockIdx.y;
	const int img = plane / g.C, c = plane - img * g.C;
	const int k1 = w.info[img].status ? 0 : w.segidx[((long)img * 48 + c * 16 + l) * MAX_PLANES + p];
	if (threadIdx.x == 0)   // k_count's tiles find their segment's symbols with one look-up
		w.count_base[(long)plane * 16 + l] = k1 ? 1ull + 2ull * w.seg_symbase[(long)img * MAX_SEGS + k1 - 1] : 0ull;
	if (!k1)
		return;
	const int t0 = g.tile_first[l], nt = g.tile_first[l + 1] - t0;
	const unsigned short *ns = w.tile_nonsig + (long)plane * w.NT + t0;
	unsigned *rank = w.tile_rank + ((long)plane * MAX_PLANES + p) * w.NT + t0;
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	if (threadIdx.x == 0)
		carry = 0;
	__syncthreads();
	for (int b0 = 0; b0 < nt; b0 += 1024) {
		const int i = b0 + threadIdx.x;
		const unsigned v = i < nt ? ns[i] : 0u;
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

// ones among the tile's symbols at plane p = popcount of its slice of the bitmap (at most 65 words).  One lane
// per tile: its up to 17 16-byte loads are all in flight together, so a wave of 64 tiles pays one memory round
// trip (16 lanes per tile meant sixteen times the waves, each waiting out two dependent round trips).  Bit 31 of
// the tile's rank entry records "this plane turns coefficients of this tile on" for k_apply_all.
__global__ __launch_bounds__(256) void k_count(UnpackGeom g, DWork w, int p)
{
	const int tile = blockIdx.x * blockDim.x + threadIdx.x;
	const int plane = blockIdx.y;
	const int img = plane / g.C;
	if (tile >= w.NT)
		return;
	int l = 0;
	while (l + 1 < g.levels && tile >= g.tile_first[l + 1])
		++l;
	const unsigned long long cb = w.count_base[(long)plane * 16 + l];
	if (!cb)
		return;
	unsigned short *ns = w.tile_nonsig + (long)plane * w.NT + tile;
	unsigned *rk = w.tile_rank + ((long)plane * MAX_PLANES + p) * w.NT + tile;
	const unsigned n = *ns;
	if (!n)
		return;
	const unsigned *sym = w.symbits + img * w.BW;
	const unsigned long long a = cb - 1ull + 2ull * *rk;
	const unsigned long long e = a + 2ull * n;   // bit range [a, e), "one" flags on the even bits
	const long first = (long)(a >> 5), last = (long)((e - 1) >> 5);
	unsigned ones = 0;
#pragma unroll 4
	for (long w0 = first & ~3l; w0 <= last; w0 += 4) {
		const uint4 v = *reinterpret_cast<const uint4 *>(sym + w0);
		const unsigned x[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			const long wi = w0 + q;
			unsigned m = x[q] & 0x55555555u;
			if (wi < first || wi > last)
				m = 0u;
			if (wi == first)
				m &= ~0u << (a & 31);
			if (wi == last && (e & 31))
				m &= (1u << (e & 31)) - 1u;
			ones += (unsigned)__builtin_popcount(m);
		}
	}
	if (ones) {
		*ns = (unsigned short)(n - ones);
		*rk |= 0x80000000u;
	}
}

// Bit transpose of eight bytes (lo = bytes 0..3, hi = bytes 4..7): afterwards byte p holds bit p of the eight inputs, input j at
// bit j (the network pack.hip uses the other way round)
__device__ __forceinline__ void bit_transpose8(unsigned &lo, unsigned &hi)
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

// Handing bits out to the coefficients of a lane, four coefficients per table look-up.
// DEP_SYM[m][s]: m = 4-bit mask of coefficients that take a pass-1 symbol, s = the next four symbols (two bits
// each: one flag, sign): the symbols go, in order, to the set bits of m.  Entry = ones | (signs of those ones) << 4.
// DEP_REF[m][b]: the same for plain bits (refinement).
struct DepositTables {
	unsigned char sym[16 * 256];
	unsigned char ref[16 * 16];
};

constexpr DepositTables make_deposit_tables()
{
	DepositTables t{};
	for (unsigned m = 0; m < 16; ++m) {
		for (unsigned s = 0; s < 256; ++s) {
			unsigned ones = 0, signs = 0, k = 0;
			for (unsigned j = 0; j < 4; ++j)
				if ((m >> j) & 1u) {
					const unsigned one = (s >> (2 * k)) & 1u, sg = (s >> (2 * k + 1)) & 1u;
					ones |= one << j;
					signs |= (one & sg) << j;
					++k;
				}
			t.sym[m * 256 + s] = (unsigned char)(ones | signs << 4);
		}
		for (unsigned b = 0; b < 16; ++b) {
			unsigned out = 0, k = 0;
			for (unsigned j = 0; j < 4; ++j)
				if ((m >> j) & 1u) {
					out |= ((b >> k) & 1u) << j;
					++k;
				}
			t.ref[m * 16 + b] = (unsigned char)out;
		}
	}
	return t;
}

__device__ const DepositTables DEPOSIT = make_deposit_tables();

struct __attribute__((packed, aligned(4))) Int4S {
	int x, y, z, w;
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6))) void k_apply_all(UnpackGeom g, DWork w, const unsigned char *streams, long stream_stride, int *lin)
{
	// the deposit tables in LDS (4352 bytes, 17 per thread), before any wave leaves
	__shared__ __attribute__((aligned(16))) unsigned char dep[sizeof(DepositTables)];
	static_assert(sizeof(DepositTables) == 17 * 256, "one 16-byte piece and one byte per thread");
	*reinterpret_cast<uint4 *>(dep + 16 * threadIdx.x) = *reinterpret_cast<const uint4 *>(DEPOSIT.sym + 16 * threadIdx.x);
	dep[4096 + threadIdx.x] = DEPOSIT.ref[threadIdx.x];
	__syncthreads();
	const unsigned char *dsym = dep, *dref = dep + 4096;
	const int lane = threadIdx.x & 63;
	const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
	const int plane = blockIdx.y;
	if (tile >= w.NT)
		return;
	const int img = plane / g.C, c = plane - img * g.C;
	const DecInfo &I = w.info[img];
	if (I.status)
		return;
	int l = 0;
	while (l + 1 < g.levels && tile >= g.tile_first[l + 1])
		++l;
	const unsigned tbase = (unsigned)g.tile_base[tile];   // ring index of the tile's first coefficient
	const long base = g.pixels[l] + (long)tbase;
	const int nvalid = g.tile_cnt[tile];
	const int first = 16 * lane;
	const int nv = nvalid - first < 0 ? 0 : nvalid - first > 16 ? 16 : nvalid - first;   // this lane's coefficients
	const int vb = first < nvalid ? first : nvalid;                                      // coefficients in the lanes before
	const unsigned *sym = w.symbits + img * w.BW;
	const unsigned *stream = (const unsigned *)(streams + img * stream_stride);
	const long stream_words = stream_stride >> 2;
	// The planes' bits of the lane's 16 coefficients — one 16-bit row per plane, rows coming in from the highest plane down —
	// are kept as a 256-bit shift register (eight registers: every plane pushes its row in with eight v_alignbit) and turned
	// into magnitudes once, at the end, by two 8x8 bit transposes per eight planes.  (Until round 4 every plane set its bit in
	// 16 magnitude registers — extract, shift, or: 48 of the ~60 vector instructions a plane costs a lane.)
	unsigned R[8] = { 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u };
	int p_last = 0, rows_in = 0;            // the plane of the row that came in last (halfword 0), rows so far; uniform
	// (only the registers that hold rows move: rows_in is uniform, and with the eight or nine planes of an 8-bit picture the
	// upper half of the register never does)
	auto push_row = [&](unsigned row16) {
		if (rows_in >= 8) {
#pragma unroll
			for (int k = 7; k >= 4; --k)
				R[k] = __builtin_amdgcn_alignbit(R[k], R[k - 1], 16);
		}
		if (rows_in >= 6)
			R[3] = __builtin_amdgcn_alignbit(R[3], R[2], 16);
		if (rows_in >= 4)
			R[2] = __builtin_amdgcn_alignbit(R[2], R[1], 16);
		if (rows_in >= 2)
			R[1] = __builtin_amdgcn_alignbit(R[1], R[0], 16);
		R[0] = (R[0] << 16) | row16;
		++rows_in;
	};
	unsigned neg = 0;                       // bit i: coefficient i of this lane is negative
	const unsigned valid16 = nv >= 16 ? 0xffffu : (1u << nv) - 1u;
	unsigned ins = valid16;                 // bit i: coefficient i is still insignificant

	// What a plane needs from memory does not depend on the planes before it: the tile's slice of the symbol
	// bitmap (its insignificant coefficients are consecutive symbols from the tile's rank: at most 65 words) and
	// its slice of the segment's refinement block (at most 33 words).  Lane p fetches plane p's record, then the
	// wave fetches the slices of up to AP_BATCH planes as coalesced rows and keeps them in LDS: all those round
	// trips overlap instead of two dependent ones per plane.
	constexpr int AP_BATCH = 8, SYMW = 66, REFW = 34;
	// (one 4 KB stretch of LDS per wave holds these slices and, once all planes are in, the square that is staged for
	// the store: 20 KB per workgroup instead of 34, six waves per SIMD instead of four)
	static_assert(AP_BATCH * (SYMW + REFW) <= SQ_WORDS, "the plane slices share the wave's square staging area");
	__shared__ __attribute__((aligned(16))) unsigned ap_mem[4][SQ_WORDS];
	unsigned (*ap_sym)[SYMW] = reinterpret_cast<unsigned (*)[SYMW]>(ap_mem[threadIdx.x >> 6]);
	unsigned (*ap_ref)[REFW] = reinterpret_cast<unsigned (*)[REFW]>(ap_mem[threadIdx.x >> 6] + AP_BATCH * SYMW);
	__shared__ unsigned ap_rank[4][MAX_PLANES], ap_n2[4][MAX_PLANES], ap_refbit[4][MAX_PLANES];
	const int wv = threadIdx.x >> 6;
	unsigned my_tr = 0, my_n2 = 0;
	unsigned long long my_sb = 0, my_b2 = 0;
	bool my_live = false;
	if (lane < MAX_PLANES && lane < I.planes[c]) {
		const int k1 = w.segidx[((long)img * 48 + c * 16 + l) * MAX_PLANES + lane];
		if (k1) {
			my_live = true;
			my_tr = w.tile_rank[((long)plane * MAX_PLANES + lane) * w.NT + tile];
			my_sb = w.seg_symbase[(long)img * MAX_SEGS + k1 - 1];
			my_b2 = w.seg_b2[(long)img * MAX_SEGS + k1 - 1];
			my_n2 = w.seg_n2done[(long)img * MAX_SEGS + k1 - 1];
		}
	}
	// planes that do something: from the first one that turns a coefficient of this tile on (k_count's flag) downwards
	const unsigned long long livem = ballot64(my_live), flagm = ballot64(my_live && (my_tr >> 31));
	const int ptop = flagm ? 63 - __builtin_clzll(flagm) : -1;
	unsigned long long todo = ptop >= 0 ? livem & ((2ull << ptop) - 1ull) : 0ull;
	while (todo) {
		// ---- fetch the slices of the next AP_BATCH planes (descending) ----
		int pl[AP_BATCH];
		unsigned long long t2 = todo;
#pragma unroll
		for (int q = 0; q < AP_BATCH; ++q) {
			pl[q] = t2 ? 63 - __builtin_clzll(t2) : -1;
			t2 &= pl[q] >= 0 ? ~(1ull << pl[q]) : ~0ull;
		}
		unsigned sv0[AP_BATCH], sv1[AP_BATCH], rv[AP_BATCH];
#pragma unroll
		for (int q = 0; q < AP_BATCH; ++q) {
			sv0[q] = sv1[q] = rv[q] = 0u;
			if (pl[q] < 0)
				continue;   // uniform
			const unsigned rank = (unsigned)__builtin_amdgcn_readlane((int)my_tr, pl[q]) & 0x7fffffffu;
			const unsigned long long sb = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(my_sb >> 32), pl[q]) << 32) |
				(unsigned)__builtin_amdgcn_readlane((int)my_sb, pl[q]);
			const unsigned long long b2 = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(my_b2 >> 32), pl[q]) << 32) |
				(unsigned)__builtin_amdgcn_readlane((int)my_b2, pl[q]);
			const unsigned n2 = (unsigned)__builtin_amdgcn_readlane((int)my_n2, pl[q]);
			const unsigned *sp = sym + (sb >> 4) + (rank >> 4);                    // a segment's symbols start on a word boundary
			sv0[q] = sp[lane];
			if (lane < SYMW - 64)
				sv1[q] = sp[64 + lane];
			const unsigned r2t = tbase - rank;                                     // refinement index of the tile's first significant coefficient
			const unsigned long long rbit = b2 + r2t;
			const long rw = (long)(rbit >> 5) + lane;
			if (lane < REFW && r2t < n2 && rw < stream_words)
				rv[q] = stream[rw];
			if (lane == 0) {
				ap_rank[wv][pl[q]] = rank;
				ap_n2[wv][pl[q]] = n2;
				ap_refbit[wv][pl[q]] = (unsigned)(rbit & 31);
			}
		}
#pragma unroll
		for (int q = 0; q < AP_BATCH; ++q) {
			if (pl[q] < 0)
				continue;
			ap_sym[q][lane] = sv0[q];
			if (lane < SYMW - 64)
				ap_sym[q][64 + lane] = sv1[q];
			if (lane < REFW)
				ap_ref[q][lane] = rv[q];
		}
		sq_wave_sync();
		// ---- the planes of the batch, in order ----
		const int nbatch = __builtin_popcountll(todo) < AP_BATCH ? __builtin_popcountll(todo) : AP_BATCH;
#pragma unroll 1
		for (int q = 0; q < nbatch; ++q) {
			const int p = 63 - __builtin_clzll(todo);
			todo &= ~(1ull << p);
			const unsigned rank = ap_rank[wv][p], n2done = ap_n2[wv][p];
			const unsigned ci = (unsigned)__builtin_popcount(ins);
			const unsigned nb = wave_incl_add_u(ci) - ci;                 // insignificant coefficients in the lanes before
			// symbols: two bits each (one flag, sign), this lane's start `nb` symbols after the tile's
			const unsigned srel = (rank & 15u) + nb;
			const unsigned *sw = ap_sym[q] + (srel >> 4);
			unsigned s32 = __builtin_amdgcn_alignbit(sw[1], sw[0], (srel & 15u) * 2u);
			// refinement bits that the stream still holds for this lane (a truncated stream ends inside some block)
			const unsigned sb4 = (unsigned)vb - nb;                          // significant coefficients in the lanes before
			const unsigned r2 = tbase - rank + sb4;                          // refinement index of this lane's first significant one
			const unsigned cs = (unsigned)nv - ci;
			const unsigned avail = r2 < n2done ? (n2done - r2 < cs ? n2done - r2 : cs) : 0u;
			const unsigned rrel = ap_refbit[wv][p] + sb4;
			const unsigned *rwp = ap_ref[q] + (rrel >> 5);
			unsigned r16 = __builtin_amdgcn_alignbit(rwp[1], rwp[0], rrel & 31u) & bfm(avail, 0u);   // avail <= 16
			// four coefficients per look-up: symbols to the insignificant ones, refinement bits to the others, in order
			const unsigned sig = valid16 & ~ins;
			unsigned ones16 = 0, sgn16 = 0, ref16 = 0;
#pragma unroll
			for (int n4 = 0; n4 < 4; ++n4) {
				const unsigned mi = (ins >> (4 * n4)) & 15u, ms = (sig >> (4 * n4)) & 15u;
				const unsigned es = dsym[mi * 256u + (s32 & 255u)], er = dref[ms * 16u + (r16 & 15u)];
				ones16 |= (es & 15u) << (4 * n4);
				sgn16 |= (es >> 4) << (4 * n4);
				ref16 |= er << (4 * n4);
				s32 >>= 2 * __builtin_popcount(mi);
				r16 >>= __builtin_popcount(ms);
			}
			const unsigned bits16 = ones16 | ref16;
			neg |= sgn16;                            // the sign follows a pass-1 one (decode.c:80-85)
			ins &= ~ones16;
			// (a (channel, level)'s planes come in descending order without gaps: every stream codes them that way and a cut
			// stream loses the lowest ones; should one ever be missing, its row is all zeros)
			for (int gap = rows_in ? p_last - p - 1 : 0; gap > 0; --gap)
				push_row(0u);
			push_row(bits16);
			p_last = p;
		}
		sq_wave_sync();   // the next batch overwrites the slices
	}
	// magnitudes: halfword m of the shift register is the row of plane p_last + m; bytes of eight rows, bit-transposed, are the
	// eight planes' bits of eight coefficients
	unsigned mag[16];
	{
		unsigned lo[2], hi[2];
#pragma unroll
		for (int gg = 0; gg < 2; ++gg) {
			const unsigned sel = (unsigned)gg * 0x01010101u + 0x06040200u;   // byte gg of each of four halfwords
			lo[gg] = __builtin_amdgcn_perm(R[1], R[0], sel);
			hi[gg] = __builtin_amdgcn_perm(R[3], R[2], sel);
			bit_transpose8(lo[gg], hi[gg]);
		}
#pragma unroll
		for (int i = 0; i < 16; ++i)
			mag[i] = __builtin_amdgcn_ubfe((i & 4) ? hi[i >> 3] : lo[i >> 3], 8u * (i & 3), 8u);
		if (rows_in > 8) {   // uniform: more than eight planes (no 8-bit picture has them on its finest levels)
#pragma unroll
			for (int gg = 0; gg < 2; ++gg) {
				const unsigned sel = (unsigned)gg * 0x01010101u + 0x06040200u;
				lo[gg] = __builtin_amdgcn_perm(R[5], R[4], sel);
				hi[gg] = __builtin_amdgcn_perm(R[7], R[6], sel);
				bit_transpose8(lo[gg], hi[gg]);
			}
#pragma unroll
			for (int i = 0; i < 16; ++i)
				mag[i] |= __builtin_amdgcn_ubfe((i & 4) ? hi[i >> 3] : lo[i >> 3], 8u * (i & 3), 8u) << 8;
		}
#pragma unroll
		for (int i = 0; i < 16; ++i)
			mag[i] <<= p_last;
	}
	if (((g.sq_levels >> l) & 1u) && nvalid == TILE) {
		// the tile is a whole 32x32 square of the pyramid: decode.c:32-65 reconstruction() for it right here,
		// with the dead-zone bias of planes that were never decoded (decode.c:51-58)
		const int m = I.missing[c * 16 + l] - 2;
		const int bias = m >= 0 ? 1 << m : 0;
		int val[16];
#pragma unroll
		for (int i = 0; i < 16; ++i) {
			int v = ((neg >> i) & 1u) ? -(int)mag[i] : (int)mag[i];
			if (bias && v)
				v += v < 0 ? -bias : bias;
			val[i] = v;
		}
		if ((g.lv16 >> l) & 1u)
			store_square16(g.fine16 + (long)plane * g.lin_stride, g.W, g.side[l], g.tile_blk[tile], lane, ap_mem[threadIdx.x >> 6], val);
		else
			store_square16(g.pyr + (long)plane * g.lin_stride, g.W, g.side[l], g.tile_blk[tile], lane, ap_mem[threadIdx.x >> 6], val);
		return;
	}
	int *dst = lin + (long)plane * g.lin_stride + base + first;
	if (nvalid == TILE) {
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			Int4S v;
			v.x = ((neg >> (4 * q)) & 1u) ? -(int)mag[4 * q] : (int)mag[4 * q];
			v.y = ((neg >> (4 * q + 1)) & 1u) ? -(int)mag[4 * q + 1] : (int)mag[4 * q + 1];
			v.z = ((neg >> (4 * q + 2)) & 1u) ? -(int)mag[4 * q + 2] : (int)mag[4 * q + 2];
			v.w = ((neg >> (4 * q + 3)) & 1u) ? -(int)mag[4 * q + 3] : (int)mag[4 * q + 3];
			*reinterpret_cast<Int4S *>(dst + 4 * q) = v;
		}
	} else {
#pragma unroll
		for (int i = 0; i < 16; ++i)
			if (i < nv)
				dst[i] = ((neg >> i) & 1u) ? -(int)mag[i] : (int)mag[i];
	}
}

// tile_nonsig starts as the tile's coefficient count
__global__ __launch_bounds__(256) void k_tiles_init(UnpackGeom g, DWork w, int nplanes)
{
	const int tile = blockIdx.x * blockDim.x + threadIdx.x;
	const int plane = blockIdx.y;
	if (tile >= w.NT || plane >= nplanes)
		return;
	int l = 0;
	while (l + 1 < g.levels && tile >= g.tile_first[l + 1])
		++l;
	w.tile_nonsig[(long)plane * w.NT + tile] = g.tile_cnt[tile];
}

// The chunk tables are laid out for the stream stride, but only the chunks that hold stream bytes are
// worked on: nch[i] (+1 sentinel row) is a multiple of 4 like NCH + 1, for the vectorised scans.
__global__ __launch_bounds__(256) void k_nch(DWork w, const unsigned long long *lens, int n)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
		return;
	const unsigned long long bits = lens[i] * 8ull;
	long c = (long)((bits + CH_BITS - 1) / CH_BITS);
	c = (c + 1 + 3) / 4 * 4 - 1;
	w.nch[i] = (int)(c < w.NCH ? c : w.NCH);
}

} // namespace

enum { SLOT_UP_SMALL = 12, SLOT_UP_BITS, SLOT_UP_TILES, SLOT_UP_CHUNKS };

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// `done(user, first, count)` (optional) is called on the host as soon as host_info[first..first+count)
// is valid and every kernel writing those images' planes has been enqueued on ctx->stream (or
// ordered before it): the caller can queue its own follow-up work for that part of the batch
// there while the rest of the batch is still being decoded.
int dwtx_decode_planes_ex(dwtx_ctx *ctx, int32_t *lin, int32_t *pyr, const uint8_t *streams, size_t stream_stride,
	const unsigned long long *dev_lens, int W, int H, int C, int n, int levels_max, dwtx_decode_info *host_info,
	int (*done)(void *user, int first, int count, unsigned fused_levels), void *user, dwtx_p16 p16)
{
	if (!ctx || !lin || !streams || !dev_lens || !host_info || (C != 1 && C != 3) || n < 1 || n > 65535 / 3 ||
		(stream_stride & 7) || stream_stride < 64)
		return DWTX_ERR_ARG;
	DWTX_ENTER(ctx);
	DWTX_CHECK_DIMS(W, H);
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
	g.pyr = nullptr;
	g.fine16 = nullptr;
	g.lv16 = 0u;
	g.sq_levels = 0;
	{
		dwtx_geom gg;
		dwtx_geometry(&gg, W, H);
		for (int l = 0; l <= g.levels; ++l)
			g.side[l] = l < g.levels ? gg.lengths[l + 1] : 0;
	}
	// the levels that are full power-of-two squares can go straight into the pyramid (k_apply_all)
	const unsigned sq_all = pyr && !((uintptr_t)pyr & 15) && !ctx->opt[DWTX_OPT_NO_SQUARE_TILES] ? dwtx_square_levels(W, H) : 0u;
	constexpr int MAX_PARTS = 4;
	unsigned part_mask[MAX_PARTS] = { 0u, 0u, 0u, 0u };
	int part_first[MAX_PARTS + 1] = { 0, 0, 0, 0, 0 };   // images [part_first[k], part_first[k+1]) are part k
	auto part_of = [&](int i0) { int k = 0; while (k + 1 < MAX_PARTS && part_first[k + 1] <= i0 && part_first[k + 1] > 0) ++k; return k; };
	dwtx_tiles tiles;
	{
		const int rc_tiles = dwtx_get_tiles(ctx, W, H, &tiles);
		if (rc_tiles)
			return rc_tiles;
	}
	const int NT = tiles.NT;
	for (int l = 0; l <= g.levels; ++l)
		g.tile_first[l] = tiles.tile_first[l];
	g.tile_base = tiles.base;
	g.tile_cnt = tiles.cnt;
	g.tile_blk = tiles.blk;
	const int nplanes = n * C;

	DWork w;
	unsigned *clear_words = nullptr;   // [n] bitmap words each image can use (k_peek)
	memset(&w, 0, sizeof(w));
	w.NT = NT;
#ifdef DWTX_DEBUG_HOOKS   // tools/dbg_walker.py: device address for the walker's cycle counters (never in the shipped build)
	w.dbg = (unsigned long long *)getenv("DWTX_DBG_PTR") ? (unsigned long long *)strtoull(getenv("DWTX_DBG_PTR"), 0, 0) : nullptr;
#endif
	int link_rounds = LINK_ROUNDS;
	w.streak_max = WALK_STREAK_MAX;
	w.scans_base = WALK_SCANS_BASE;
#ifdef DWTX_DEBUG_HOOKS   // tools/find_second_walk.py: other limits to try
	if (getenv("DWTX_DBG_STREAK"))
		w.streak_max = (unsigned)strtoul(getenv("DWTX_DBG_STREAK"), 0, 0);
	if (getenv("DWTX_DBG_SCANS"))
		w.scans_base = (unsigned)strtoul(getenv("DWTX_DBG_SCANS"), 0, 0);
	if (getenv("DWTX_DBG_ROUNDS") && atoi(getenv("DWTX_DBG_ROUNDS")) >= 1 && atoi(getenv("DWTX_DBG_ROUNDS")) <= LINK_ROUNDS)
		link_rounds = atoi(getenv("DWTX_DBG_ROUNDS"));
#endif
	// every segment owns ceil32(ring size) symbol slots; at most MAX_PLANES segments per (channel, level)
	w.BW = ((long)((((unsigned long long)g.total + 32ull * g.levels) * C * MAX_PLANES) >> 4) + 128 + 3) & ~3l;   // 2 bits per symbol; whole 16-byte groups per image
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
		unsigned *bits = (unsigned *)dwtx_scratch(ctx, SLOT_UP_BITS, sizeof(unsigned) * (size_t)n * w.BW);
		off = 0;
		const size_t o_ts = take(sizeof(short) * (size_t)nplanes * NT);
		const size_t o_tr = take(sizeof(unsigned) * (size_t)nplanes * MAX_PLANES * NT);
		const size_t o_cb = take(sizeof(unsigned long long) * (size_t)nplanes * 16);
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
		w.symbits = bits;
		w.tile_nonsig = (unsigned short *)(tiles + o_ts);
		w.tile_rank = (unsigned *)(tiles + o_tr);
		w.count_base = (unsigned long long *)(tiles + o_cb);
		// speculative chunk tables
		w.NCH = (long)((stream_stride * 8 + CH_BITS - 1) / CH_BITS);
		w.NCH = (w.NCH + 1 + 3) / 4 * 4 - 1;   // NCH+1 table rows per stream, a multiple of 4 for the vectorised scans
		w.NB = (w.NCH + 1 + SCAN_BLOCK - 1) / SCAN_BLOCK;
		off = 0;
		w.MAX_HOPS = 8 * MAX_SEGS + w.NCH / 8;
		w.fam = 1;
		const size_t o_ep = take(sizeof(short) * (size_t)n * FAM * w.NCH);
		const size_t o_eq = take(sizeof(short) * (size_t)n * FAM * w.NCH);
		const size_t o_cs = take(sizeof(unsigned long long) * (size_t)n * FAM * (w.NCH + 1));
		const size_t o_ct = take(sizeof(unsigned) * (size_t)n * FAM * (w.NCH + 1));
		const size_t o_cg = take(sizeof(unsigned) * (size_t)n * FAM * (w.NCH + 1));
		const size_t o_ps = take(sizeof(unsigned long long) * (size_t)n * FAM * w.NB);
		const size_t o_pt = take(sizeof(unsigned) * (size_t)n * FAM * w.NB);
		const size_t o_pg = take(sizeof(unsigned) * (size_t)n * FAM * w.NB);
		const size_t o_hs = take(sizeof(int) * (size_t)n * w.MAX_HOPS);
		const size_t o_hf = take(sizeof(unsigned) * (size_t)n * w.MAX_HOPS);
		const size_t o_hl = take(sizeof(unsigned) * (size_t)n * w.MAX_HOPS);
		const size_t o_hq = take(sizeof(unsigned) * (size_t)n * w.MAX_HOPS);
		const size_t o_he = take(sizeof(unsigned) * (size_t)n * w.MAX_HOPS);
		const size_t o_hn = take(sizeof(unsigned) * (size_t)n * w.MAX_HOPS);
		const size_t o_br = take(sizeof(unsigned) * (size_t)n * FAM * w.NCH);
		w.todo_cap = ((w.NCH + 256) / 256 + 63) / 64 * 256 + 256;   // chunks whose workgroup maps to one shard
		const size_t o_td = take(sizeof(unsigned) * 2 * (size_t)n * FAM * 64 * w.todo_cap);
		const size_t o_tc = take(sizeof(unsigned) * (LINK_ROUNDS + 2) * (size_t)n * FAM * LINK_SHARDS);
		const size_t o_nh = take(sizeof(int) * (size_t)n);
		const size_t o_nc = take(sizeof(int) * (size_t)n);
		const size_t o_cw = take(sizeof(unsigned) * (size_t)n);
		const size_t o_ix = take(sizeof(SegIndex) * (size_t)n * MAX_SEGS);
		const size_t o_sr = take(sizeof(SegResult) * (size_t)n * MAX_SEGS);
		const size_t o_in = take(sizeof(int) * (size_t)n);
		const size_t o_sl = take(sizeof(unsigned) * (size_t)n * (MAX_SEGS + 1));
		char *chunks = (char *)dwtx_scratch(ctx, SLOT_UP_CHUNKS, off);
		if (!chunks)
			return DWTX_ERR_NOMEM;
		w.exitX = (unsigned short *)(chunks + o_ep);
		w.entryE = (unsigned short *)(chunks + o_eq);
		w.cs = (unsigned long long *)(chunks + o_cs);
		w.ct = (unsigned *)(chunks + o_ct);
		w.cg = (unsigned *)(chunks + o_cg);
		w.part_s = (unsigned long long *)(chunks + o_ps);
		w.part_t = (unsigned *)(chunks + o_pt);
		w.part_g = (unsigned *)(chunks + o_pg);
		w.hop_seg = (int *)(chunks + o_hs);
		w.hop_first = (unsigned *)(chunks + o_hf);
		w.hop_last = (unsigned *)(chunks + o_hl);
		w.hop_q0 = (unsigned *)(chunks + o_hq);
		w.hop_entry = (unsigned *)(chunks + o_he);
		w.hop_ntok = (unsigned *)(chunks + o_hn);
		w.breaks = (unsigned *)(chunks + o_br);
		w.todo[0] = (unsigned *)(chunks + o_td);
		w.todo[1] = w.todo[0] + (size_t)n * FAM * 64 * w.todo_cap;
		w.todo_count = (unsigned *)(chunks + o_tc);
		w.todo_round = (long)n * FAM * LINK_SHARDS;
		DWTX_HIP(hipMemsetAsync(w.todo_count, 0, sizeof(unsigned) * (LINK_ROUNDS + 2) * (size_t)w.todo_round, ctx->stream));
		w.nhops = (int *)(chunks + o_nh);
		w.nch = (int *)(chunks + o_nc);
		clear_words = (unsigned *)(chunks + o_cw);
		w.idx = (SegIndex *)(chunks + o_ix);
		w.segres = (SegResult *)(chunks + o_sr);
		w.idx_nsegs = (int *)(chunks + o_in);
		w.seg_slot = (unsigned *)(chunks + o_sl);
		DWTX_HIP(hipMemsetAsync(w.nhops, 0, sizeof(int) * (size_t)n, ctx->stream));
		DWTX_HIP(hipMemsetAsync(small, 0, o_zero_end, ctx->stream));
		// The symbol bitmap (the one big clear, ~64 MB per 4096x4096 plane) is only needed by the token walk:
		// it is cleared on the second stream while the chunk tables are built on the first.
		{
			const int rc_side = dwtx_need_side_streams(ctx, false);
			if (rc_side)
				return rc_side;
		}
		DWTX_HIP(hipEventRecord(ctx->ev[2], ctx->stream));            // earlier work on the main stream may still read the bitmap
		DWTX_HIP(hipStreamWaitEvent(ctx->aux, ctx->ev[2], 0));
		hipLaunchKernelGGL(k_peek, dim3(dwtx_cdiv(n, 64)), dim3(64), 0, ctx->aux, g, streams, (long)stream_stride, dev_lens, w.BW, clear_words, n);
		hipLaunchKernelGGL(k_clear_bitmaps, dim3(64, n), dim3(256), 0, ctx->aux, bits, w.BW, clear_words);
		// (the per-tile counters are first used after the token walk too: beside the clear, on the main stream, this
		// trivial kernel waited 0.5 ms for a free slot and held the chunk kernels up)
		hipLaunchKernelGGL(k_tiles_init, dim3(dwtx_cdiv(NT, 256), nplanes), dim3(256), 0, ctx->aux, g, w, nplanes);
		DWTX_HIP(hipEventRecord(ctx->ev[3], ctx->aux));
	}
	hipStream_t s = ctx->stream;
	// decode.c:177-179 zeroes everything; here the rings are written exactly once by k_apply_all, so only
	// the root image (written by the token walker when it has any bits) needs clearing
	DWTX_HIP(hipMemset2DAsync(lin, sizeof(int) * (size_t)g.lin_stride, 0, sizeof(int) * (size_t)g.pixels[0], nplanes, s));
	hipLaunchKernelGGL(k_nch, dim3(dwtx_cdiv(n, 256)), dim3(256), 0, s, w, dev_lens, n);

	// everything below works on a range of images [i0, i0+cnt): all tables are per image
	auto slice = [&](int i0) {
		DWork h = w;
		h.info += i0;
		h.segidx += (size_t)i0 * 48 * MAX_PLANES;
		h.seg_desc += (size_t)i0 * MAX_SEGS;
		h.seg_symbase += (size_t)i0 * MAX_SEGS;
		h.seg_b2 += (size_t)i0 * MAX_SEGS;
		h.seg_n2done += (size_t)i0 * MAX_SEGS;
		h.nonsig += (size_t)i0 * 48;
		h.symbits += (size_t)i0 * w.BW;
		h.tile_nonsig += (size_t)i0 * C * NT;
		h.tile_rank += (size_t)i0 * C * MAX_PLANES * NT;
		h.count_base += (size_t)i0 * C * 16;
		h.exitX += (size_t)i0 * FAM * w.NCH;
		h.entryE += (size_t)i0 * FAM * w.NCH;
		h.cs += (size_t)i0 * FAM * (w.NCH + 1);
		h.ct += (size_t)i0 * FAM * (w.NCH + 1);
		h.cg += (size_t)i0 * FAM * (w.NCH + 1);
		h.part_s += (size_t)i0 * FAM * w.NB;
		h.part_t += (size_t)i0 * FAM * w.NB;
		h.part_g += (size_t)i0 * FAM * w.NB;
		h.hop_seg += (size_t)i0 * w.MAX_HOPS;
		h.hop_first += (size_t)i0 * w.MAX_HOPS;
		h.hop_last += (size_t)i0 * w.MAX_HOPS;
		h.hop_q0 += (size_t)i0 * w.MAX_HOPS;
		h.hop_entry += (size_t)i0 * w.MAX_HOPS;
		h.hop_ntok += (size_t)i0 * w.MAX_HOPS;
		h.breaks += (size_t)i0 * FAM * w.NCH;
		for (int k = 0; k < 2; ++k)
			h.todo[k] += (size_t)i0 * FAM * 64 * w.todo_cap;
		h.todo_count += (size_t)i0 * FAM * LINK_SHARDS;
		h.nhops += i0;
		h.nch += i0;
		h.idx += (size_t)i0 * MAX_SEGS;
		h.segres += (size_t)i0 * MAX_SEGS;
		h.idx_nsegs += i0;
		h.seg_slot += (size_t)i0 * (MAX_SEGS + 1);
		if (h.dbg)
			h.dbg += (size_t)i0 * 8;
		return h;
	};
	// chunk tables; then (walk) token walk and symbol bits of the hopped-over chunks
	auto pre = [&](hipStream_t st, int i0, int cnt, int fam) -> int {
		DWork h = slice(i0);
		h.fam = fam;
		const unsigned char *str = streams + (size_t)i0 * stream_stride;
		const unsigned cblocks = (unsigned)((w.NCH + 1 + 255) / 256);
		const dim3 cg(cblocks < CHUNK_GRID ? cblocks : CHUNK_GRID, cnt * h.fam);
		const unsigned fblocks = (unsigned)((w.NCH + 1 + 256 * LINK_RUN - 1) / (256 * LINK_RUN));
		hipLaunchKernelGGL(k_link_first, dim3(fblocks < CHUNK_GRID ? fblocks : CHUNK_GRID, cnt * h.fam), dim3(256), 0, st, h, str,
			(long)stream_stride);   // speculation and round 1 in one, fills the list of chunks to redo
		int cur = 1;
		for (int r = 2; r <= link_rounds; ++r) {   // the lists shrink: fewer workgroups per shard after the first rounds
			hipLaunchKernelGGL(k_link_work, dim3(LINK_SHARDS * (r <= 3 ? 4 : 1), cnt * h.fam), dim3(256), 0, st, h, str, (long)stream_stride, cur, r);
			cur ^= 1;
		}
		const unsigned sblocks = (unsigned)(w.NB < SCAN_GRID ? w.NB : SCAN_GRID);
		hipLaunchKernelGGL(k_scan_local, dim3(sblocks, cnt * h.fam), dim3(256), 0, st, h);
		hipLaunchKernelGGL(k_scan_parts, dim3(cnt * h.fam), dim3(256), 0, st, h);
		hipLaunchKernelGGL(k_scan_add, dim3(sblocks, cnt * h.fam), dim3(256), 0, st, h);
		DWTX_LAUNCH_CHECK();
		return DWTX_OK;
	};
	// One family of recorded paths is enough for almost every stream (k_link_first); DWTX_TWO_FAMILIES starts with both
	// (a test hook: it is the path a walk that gave up falls back to).
	// One or two images leave most of the chip idle anyway: both families then, for the shorter walk.
	const int fam0 = n <= 2 || ctx->opt[DWTX_OPT_TWO_FAMILIES] ? FAM : 1;
	// Sidecar indices (dwtx_ctx_set_index): a part of the batch whose images all come with a plausible index is
	// walked segment-parallel; k_segjoin proves the index on the way or hands the part back to the serial walk.
	const dwtx_index *ix_in = ctx->index_in ? ctx->index_in + ctx->index_base : nullptr;
	dwtx_index *ix_out = ctx->index_out ? ctx->index_out + ctx->index_base : nullptr;
	static_assert(sizeof(dwtx_seg_index) == sizeof(SegIndex) && DWTX_INDEX_MAX_SEGS == MAX_SEGS, "SegIndex is the device image of dwtx_seg_index");
	auto indexed = [&](int i0, int cnt) -> bool {
		if (!ix_in || g.levels_max < g.levels || ctx->opt[DWTX_OPT_NO_INDEX])
			return false;
		for (int i = i0; i < i0 + cnt; ++i)
			if (ix_in[i].magic != DWTX_INDEX_MAGIC || ix_in[i].W != W || ix_in[i].H != H || ix_in[i].C != C || ix_in[i].nsegs <= 0 ||
				ix_in[i].nsegs > MAX_SEGS)
				return false;
		return true;
	};
	bool part_indexed[MAX_PARTS] = { false, false, false, false };
	std::vector<int> index_segs(ix_in ? (size_t)n : 0u);
	auto walk = [&](hipStream_t st, int i0, int cnt, int fam, bool use_index) -> int {
		DWork h = slice(i0);
		h.fam = fam;
		const unsigned char *str = streams + (size_t)i0 * stream_stride;
		DWTX_HIP(hipStreamWaitEvent(st, ctx->ev[3], 0));   // the bitmap is clear
		part_indexed[part_of(i0)] = use_index;
		if (use_index) {
			int maxk = 0;
			for (int i = 0; i < cnt; ++i) {
				const int k = ix_in[i0 + i].nsegs;
				index_segs[(size_t)(i0 + i)] = k;   // (lives as long as this call: the copy below may still be reading it)
				maxk = k > maxk ? k : maxk;
				DWTX_HIP(hipMemcpyAsync(h.idx + (size_t)i * MAX_SEGS, ix_in[i0 + i].seg, sizeof(SegIndex) * (size_t)k, hipMemcpyHostToDevice, st));
			}
			DWTX_HIP(hipMemcpyAsync(h.idx_nsegs, index_segs.data() + i0, sizeof(int) * (size_t)cnt, hipMemcpyHostToDevice, st));
			hipLaunchKernelGGL(k_segprep, dim3(dwtx_cdiv(cnt, 64)), dim3(64), 0, st, h, dev_lens + i0, (long)stream_stride, cnt);
			hipLaunchKernelGGL(k_tokenize<true>, dim3(maxk, cnt), dim3(64), 0, st, g, h, str, (long)stream_stride, dev_lens + i0,
				lin + (size_t)i0 * C * g.lin_stride, cnt);
			hipLaunchKernelGGL(k_segjoin, dim3(cnt), dim3(64), 0, st, g, h, str, (long)stream_stride, dev_lens + i0,
				lin + (size_t)i0 * C * g.lin_stride, cnt);
		} else
		hipLaunchKernelGGL(k_tokenize<false>, dim3(cnt), dim3(64), 0, st, g, h, str, (long)stream_stride, dev_lens + i0,
			lin + (size_t)i0 * C * g.lin_stride, cnt);
		const unsigned hblocks = (unsigned)((w.NCH + 255) / 256);
		hipLaunchKernelGGL(k_hopbits, dim3(hblocks < CHUNK_GRID ? hblocks : CHUNK_GRID, cnt), dim3(256), 0, st, h, str, (long)stream_stride);
		DWTX_LAUNCH_CHECK();
		return DWTX_OK;
	};
	static_assert(sizeof(dwtx_decode_info) == sizeof(DecInfo), "DecInfo is the device image of dwtx_decode_info");
	// walker results to the host (synchronises the stream), then the plane scatter
	auto post = [&](hipStream_t st, int i0, int cnt) -> int {
		const DWork h = slice(i0);
		DWTX_HIP(hipMemcpyAsync(host_info + i0, h.info, sizeof(DecInfo) * (size_t)cnt, hipMemcpyDeviceToHost, st));
		DWTX_HIP(hipStreamSynchronize(st));
		// A walk that did not come through leaves the marker: an indexed walk whose index does not fit the stream is
		// repeated serially (same tables), a one-family serial walk that gave up (k_tokenize) is repeated with the
		// part parsed again for both families.
		auto gave_up = [&]() {
			bool any = false;
			for (int i = i0; i < i0 + cnt; ++i)
				any = any || host_info[i].hops == WALK_GAVE_UP;
			return any;
		};
		// Only the images whose walk gave up are done again, run by run of consecutive ones (the tables are per image;
		// the other images of the part keep what their walks found).  Round 4: one frame in 200 of the benchmark's
		// synthetic ones takes the second walk — repeating its whole part of 48 frames cost the batch 40 % of its
		// decoding time (18.5 -> 27.2 ms for 160 -> 192 frames), repeating the one frame is lost in it.
		auto again = [&](int fam, bool tables) -> int {
			int rc3 = DWTX_OK;
			for (int i = i0; i < i0 + cnt && !rc3;) {
				if (host_info[i].hops != WALK_GAVE_UP) {
					++i;
					continue;
				}
				int j = i + 1;
				while (j < i0 + cnt && host_info[j].hops == WALK_GAVE_UP)
					++j;
				const int c = j - i;
				const DWork hs = slice(i);
				const long items = (long)c * FAM * LINK_SHARDS > (long)c * 48 * MAX_PLANES ? (long)c * FAM * LINK_SHARDS : (long)c * 48 * MAX_PLANES;
				hipLaunchKernelGGL(k_part_reset, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, hs, c);
				DWTX_HIP(hipMemsetAsync(hs.symbits, 0, sizeof(unsigned) * (size_t)c * w.BW, st));
				if (tables)
					rc3 = pre(st, i, c, fam);
				if (!rc3)
					rc3 = walk(st, i, c, fam, false);
				i = j;
			}
			if (rc3)
				return rc3;
			DWTX_HIP(hipMemcpyAsync(host_info + i0, h.info, sizeof(DecInfo) * (size_t)cnt, hipMemcpyDeviceToHost, st));
			DWTX_HIP(hipStreamSynchronize(st));
			return DWTX_OK;
		};
		int rc2;
		if (part_indexed[part_of(i0)] && gave_up()) {
			if (ctx->opt[DWTX_OPT_NO_INDEX_FALLBACK]) {   // test hook: shows that an index was turned down
				dwtx_set_error("the sidecar index does not fit the stream (DWTX_OPT_NO_INDEX_FALLBACK forbids the serial walk)");
				return DWTX_ERR_DEVICE;
			}
			if ((rc2 = again(fam0, false)))
				return rc2;
		}
		if (gave_up()) {
			if (ctx->opt[DWTX_OPT_NO_SECOND_WALK]) {   // test hook: shows that a stream takes this path
				dwtx_set_error("the one-family token walk gave up (DWTX_OPT_NO_SECOND_WALK forbids the second)");
				return DWTX_ERR_DEVICE;
			}
			if ((rc2 = again(FAM, true)))
				return rc2;
		}
		if (ix_out)   // the index of every stream that was decoded to its end (the serial walk wrote it, the indexed one proved it)
			for (int i = i0; i < i0 + cnt; ++i) {
				dwtx_index &X = ix_out[i];
				const DecInfo &D = reinterpret_cast<const DecInfo *>(host_info)[i];
				X.magic = DWTX_INDEX_MAGIC;
				X.W = W;
				X.H = H;
				X.C = C;
				X.reserved = 0;
				X.stream_bits = D.bits_used;
				X.nsegs = !D.status && !D.truncated && g.levels_max >= g.levels && D.nsegs > 0 && D.nsegs <= MAX_SEGS ? D.nsegs : 0;
				if (X.nsegs)
					DWTX_HIP(hipMemcpy(X.seg, h.idx + (size_t)(i - i0) * MAX_SEGS, sizeof(SegIndex) * (size_t)X.nsegs, hipMemcpyDeviceToHost));
			}
		int pmax = 0;
		for (int i = i0; i < i0 + cnt; ++i)
			if (!host_info[i].status && host_info[i].pmax > pmax)
				pmax = host_info[i].pmax;
		for (int p = pmax - 1; p >= 0; --p) {
			hipLaunchKernelGGL(k_rank, dim3(g.levels, cnt * C), dim3(1024), 0, st, g, h, p);
			hipLaunchKernelGGL(k_count, dim3(dwtx_cdiv(NT, 256), cnt * C), dim3(256), 0, st, g, h, p);
		}
		// Whole-resolution images only (decode.c:251-254: a stream that ends early gives a smaller picture, whose
		// pyramid has another pitch): then reconstruction() of the square levels happens inside k_apply_all.
		bool whole = sq_all != 0;
		for (int i = i0; i < i0 + cnt; ++i)
			whole = whole && !host_info[i].status && host_info[i].level == g.levels - 1;
		UnpackGeom ga = g;
		ga.sq_levels = whole ? sq_all : 0u;
		ga.pyr = whole ? pyr + (size_t)i0 * C * g.lin_stride : nullptr;
		// 16-bit planes for the finest ring if the caller keeps them and no stream of the part claims coefficients
		// beyond 15 bits (an 8-bit source never does; a damaged stream may: then the part stays in the int32 pyramid)
		const bool fine = whole && p16.planes && p16.levels && !((uintptr_t)p16.planes & 15) && !(p16.levels & ~sq_all) && pmax <= 15;
		ga.fine16 = fine ? p16.planes + (size_t)i0 * C * g.lin_stride : nullptr;
		ga.lv16 = fine ? p16.levels : 0u;
		part_mask[part_of(i0)] = ga.sq_levels | (fine ? DWTX_FUSED_FINE16 : 0u);
		hipLaunchKernelGGL(k_apply_all, dim3(dwtx_cdiv(NT, 4), cnt * C), dim3(256), 0, st, ga, h,
			streams + (size_t)i0 * stream_stride, (long)stream_stride, lin + (size_t)i0 * C * g.lin_stride);
		DWTX_LAUNCH_CHECK();
		return DWTX_OK;
	};
	int rc;
	if (n < 4 || ctx->opt[DWTX_OPT_ONE_STREAM]) {
		if ((rc = pre(s, 0, n, fam0)) || (rc = walk(s, 0, n, fam0, indexed(0, n))) || (rc = post(s, 0, n)))
			return rc;
		return done ? done(user, 0, n, part_mask[0]) : DWTX_OK;
	}
	// The token walk is one wave per image and leaves the chip idle: the batch runs as parts, one stream each, the
	// parts' chunk-table kernels one after the other (each fills the chip) and every part's walk beside the tables
	// of the parts after it and the scatter of the parts before it.  Four parts from 24 images on (measured on 64
	// frames: 4096x4096 gray 10.6 -> 10.3 ms, 16 x 4096x4096 RGB 11.0 -> 10.8, 1080p RGB the same 6.7: there the
	// walk itself, 2.5 ms whatever the part, and the last part's scatter are the critical path), else two.
	const long want_parts = ctx->opt[DWTX_OPT_DECODE_PARTS];
	const int K = want_parts ? (want_parts < 2 ? 2 : want_parts > MAX_PARTS ? MAX_PARTS : (int)want_parts) : (n >= 24 ? 4 : 2);
	for (int k = 0; k <= K; ++k)
		part_first[k] = (int)((long)n * k / K);
	for (int k = K + 1; k <= MAX_PARTS; ++k)
		part_first[k] = 0;
	if (K > 2 && (rc = dwtx_need_side_streams(ctx, true)))
		return rc;
	auto stream_of = [&](int k) { return k == 0 ? s : k == 1 ? ctx->aux : ctx->more[k - 2]; };
	auto run_parts = [&]() -> int {
		for (int k = 0; k < K; ++k) {
			const int i0 = part_first[k], cnt = part_first[k + 1] - part_first[k];
			hipStream_t st = stream_of(k);
			if (k == 1)
				DWTX_HIP(hipStreamWaitEvent(st, ctx->ev[0], 0));
			else if (k > 1)
				DWTX_HIP(hipStreamWaitEvent(st, ctx->pev[k - 1], 0));   // after the tables of the part before (and, through them, after everything earlier on the main stream)
			if ((rc = pre(st, i0, cnt, fam0)))
				return rc;
			if (k + 1 < K)
				DWTX_HIP(hipEventRecord(k == 0 ? ctx->ev[0] : ctx->pev[k], st));
			if ((rc = walk(st, i0, cnt, fam0, indexed(i0, cnt))))
				return rc;
		}
		for (int k = 0; k < K; ++k) {   // each part's scatter as soon as its walk is over, then the caller's follow-up on the main stream
			const int i0 = part_first[k], cnt = part_first[k + 1] - part_first[k];
			if ((rc = post(stream_of(k), i0, cnt)))
				return rc;
			if (k > 0) {
				hipEvent_t ev = k == 1 ? ctx->ev[1] : ctx->pev[4 + k];
				DWTX_HIP(hipEventRecord(ev, stream_of(k)));
				DWTX_HIP(hipStreamWaitEvent(s, ev, 0));
			}
			if (done && (rc = done(user, i0, cnt, part_mask[k])))
				return rc;
		}
		return DWTX_OK;
	};
	rc = run_parts();
	if (rc)   // a part failed: what the other parts still have queued on their streams uses the shared tables — it
		for (int k = 1; k < K; ++k)   // must be over before the caller (or the next call's clears) touches them
			(void)hipStreamSynchronize(stream_of(k));
	return rc;
}

extern "C" int dwtx_decode_planes(dwtx_ctx *ctx, int32_t *lin, const uint8_t *streams, size_t stream_stride,
	const unsigned long long *dev_lens, int W, int H, int C, int n, int levels_max, dwtx_decode_info *host_info)
{
	return dwtx_decode_planes_ex(ctx, lin, nullptr, streams, stream_stride, dev_lens, W, H, C, n, levels_max, host_info, nullptr, nullptr);
}
