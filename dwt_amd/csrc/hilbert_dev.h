// hilbert_dev.h — the Hilbert curve inside aligned 32x32 squares (hilbert.h:15-34), shared by the
// linearisation kernels and by the entropy-stage kernels that read / write coefficient tiles straight
// from / to the wavelet pyramid.
//
// The curve visits every aligned 2^k x 2^k square contiguously.  Inside a 32x32 square it is the same
// for every square up to the flips and swaps its position on the coarser levels imposes: a table
// holds the first five levels of the recursion for the 1024 points of a square (x | y << 8); what
// the levels above do to a square's points is one swap and two XOR masks, uniform per square.
//
// The entropy stage's tiles are the curve's blocks (dwtx_tiles, dwtx_internal.h): a block that lies wholly inside a
// level's ring holds 1024 consecutive ring coefficients of encode.c:46-56, in curve order — it IS a tile, and it is
// read from / written to the pyramid as the 32x32 square it is (blocks cut by the image border or the LL quadrant
// go through the linearised copy).
#pragma once

#include "dwtx_internal.h"

namespace {

constexpr int BLK_LOG2 = 5;                 // 32x32 squares
constexpr int SQ_WORDS = 1024;              // LDS words of a staged square

struct HilbertLow {
	unsigned short xy[1 << (2 * BLK_LOG2)];
};

constexpr HilbertLow make_hilbert_low()
{
	HilbertLow t{};
	for (unsigned i = 0; i < (1u << (2 * BLK_LOG2)); ++i) {
		unsigned x = 0, y = 0, d = i;
		for (unsigned s = 1; s < (1u << BLK_LOG2); s <<= 1) {
			const unsigned rx = (d >> 1) & 1u;
			const unsigned ry = (d ^ rx) & 1u;
			if (rx && !ry) {
				x ^= s - 1;
				y ^= s - 1;
			}
			if (!ry) {
				const unsigned tmp = x;
				x = y;
				y = tmp;
			}
			x |= rx ? s : 0u;
			y |= ry ? s : 0u;
			d >>= 2;
		}
		t.xy[i] = (unsigned short)(x | (y << 8));
	}
	return t;
}

__device__ const HilbertLow HILBERT_LOW = make_hilbert_low();

// x = (sw ? yl : xl) ^ mx, y = (sw ? xl : yl) ^ my  (OR-ing in a level's bit is an XOR too, the bit is still clear)
struct SquareMap {
	bool sw;
	unsigned mx, my;
};

__device__ __forceinline__ SquareMap square_map(int n, unsigned sq)   // square index = curve index >> 10, n >= 32
{
	SquareMap m = { false, 0u, 0u };
	for (unsigned s = 1u << BLK_LOG2; s < (unsigned)n; s <<= 1) {
		const unsigned rx = (sq >> 1) & 1u;
		const unsigned ry = (sq ^ rx) & 1u;
		if (rx && !ry) {
			m.mx ^= s - 1;
			m.my ^= s - 1;
		}
		if (!ry) {
			const unsigned t = m.mx;
			m.mx = m.my;
			m.my = t;
			m.sw = !m.sw;
		}
		m.mx ^= rx ? s : 0u;
		m.my ^= ry ? s : 0u;
		sq >>= 2;
	}
	return m;
}

__device__ __forceinline__ void hilbert_in_square(const SquareMap &m, int i, int &xo, int &yo)
{
	const unsigned e = HILBERT_LOW.xy[i];
	const unsigned xl = e & 255u, yl = e >> 8;
	xo = (int)((m.sw ? yl : xl) ^ m.mx);
	yo = (int)((m.sw ? xl : yl) ^ m.my);
}

// Lanes of ONE wave hand data to each other through LDS: the wave's DS operations execute in order, so all
// that is needed is that the compiler keeps the accesses on their side of this point.
__device__ __forceinline__ void sq_wave_sync()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A staged square lives in LDS as 32 rows of 32 words, the 16-byte groups of a row XOR-swizzled with the row
// (eight consecutive rows hit different banks): word (y, x) sits at (y << 5) | (x ^ ((y & 7) << 2)).  With
// x = X ^ a, y = Y ^ b (X, Y the table's coordinates, swapped or not; a, b the low five bits of the square's
// XOR masks) that position is  [(Y << 5) | (X ^ ((Y & 7) << 2))] ^ [(b << 5) | (a ^ ((b & 7) << 2))]:
// a per-point constant XOR a per-square key.  SQUARE_POS holds the constants as byte offsets, for both swap states.
struct SquarePos {
	unsigned short off[2][1 << (2 * BLK_LOG2)];
};

constexpr SquarePos make_square_pos()
{
	SquarePos t{};
	const HilbertLow h = make_hilbert_low();
	for (unsigned i = 0; i < (1u << (2 * BLK_LOG2)); ++i) {
		const unsigned xl = h.xy[i] & 255u, yl = h.xy[i] >> 8;
		t.off[0][i] = (unsigned short)(((yl << 5) | (xl ^ ((yl & 7u) << 2))) << 2);
		t.off[1][i] = (unsigned short)(((xl << 5) | (yl ^ ((xl & 7u) << 2))) << 2);
	}
	return t;
}

__device__ const SquarePos SQUARE_POS = make_square_pos();

// byte offsets (inside the staged square) of the 16 curve points 16*lane .. 16*lane+15 of the square `m` describes
__device__ __forceinline__ void square_positions16(const SquareMap &m, int lane, unsigned (&pos)[16])
{
	const unsigned a = m.mx & 31u, b = m.my & 31u;
	const unsigned key = (((b << 5) | (a ^ ((b & 7u) << 2))) << 2) * 0x00010001u;
	const unsigned short *tab = SQUARE_POS.off[m.sw ? 1 : 0];   // uniform
	const uint4 e0 = *reinterpret_cast<const uint4 *>(tab + 16 * lane);
	const uint4 e1 = *reinterpret_cast<const uint4 *>(tab + 16 * lane + 8);
	const unsigned e[8] = { e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w };
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		const unsigned w = e[k] ^ key;
		pos[2 * k] = w & 0xffffu;
		pos[2 * k + 1] = w >> 16;
	}
}

// Curve block `blk` of a ring level (outer side n) of one plane, a whole square: the wave reads it as whole
// 128-byte rows, stages it in LDS (SQ_WORDS words, 16-byte aligned) and every lane picks up its 16 consecutive curve points.
__device__ __forceinline__ void load_square16(const int *__restrict__ plane_pyr, int ppitch, int n, int blk, int lane,
	unsigned *lds, int (&val)[16])
{
	const SquareMap m = square_map(n, (unsigned)blk);
	const unsigned X0 = m.mx & ~31u, Y0 = m.my & ~31u;
#pragma unroll
	for (int it = 0; it < 4; ++it) {
		const int row = it * 8 + (lane >> 3), c4 = (lane & 7) * 4;
		const int4 v = *reinterpret_cast<const int4 *>(plane_pyr + (long)(Y0 + row) * ppitch + X0 + c4);
		*reinterpret_cast<int4 *>(lds + ((row << 5) | (c4 ^ ((row & 7) << 2)))) = v;
	}
	unsigned pos[16];
	square_positions16(m, lane, pos);
	sq_wave_sync();
	const char *base = reinterpret_cast<const char *>(lds);
#pragma unroll
	for (int k = 0; k < 16; ++k)
		val[k] = *reinterpret_cast<const int *>(base + pos[k]);
	sq_wave_sync();   // the caller may reuse the LDS words
}

// the inverse: 16 consecutive curve points per lane -> the pyramid's 32x32 square
__device__ __forceinline__ void store_square16(int *__restrict__ plane_pyr, int ppitch, int n, int blk, int lane, unsigned *lds,
	const int (&val)[16])
{
	const SquareMap m = square_map(n, (unsigned)blk);
	const unsigned X0 = m.mx & ~31u, Y0 = m.my & ~31u;
	unsigned pos[16];
	square_positions16(m, lane, pos);
	char *base = reinterpret_cast<char *>(lds);
#pragma unroll
	for (int k = 0; k < 16; ++k)
		*reinterpret_cast<int *>(base + pos[k]) = val[k];
	sq_wave_sync();
#pragma unroll
	for (int it = 0; it < 4; ++it) {
		const int row = it * 8 + (lane >> 3), c4 = (lane & 7) * 4;
		*reinterpret_cast<int4 *>(plane_pyr + (long)(Y0 + row) * ppitch + X0 + c4) =
			*reinterpret_cast<const int4 *>(lds + ((row << 5) | (c4 ^ ((row & 7) << 2))));
	}
	sq_wave_sync();
}

// The same for a plane kept as 16-bit coefficients (the finest ring of an 8-bit source, lift.hip): 64-byte rows,
// widened on the way into LDS / narrowed on the way out, so that the staged square is the one above.
__device__ __forceinline__ void load_square16(const short *__restrict__ plane_pyr, int ppitch, int n, int blk, int lane,
	unsigned *lds, int (&val)[16])
{
	const SquareMap m = square_map(n, (unsigned)blk);
	const unsigned X0 = m.mx & ~31u, Y0 = m.my & ~31u;
#pragma unroll
	for (int it = 0; it < 2; ++it) {
		const int row = it * 16 + (lane >> 2), c8 = (lane & 3) * 8;
		const uint4 v = *reinterpret_cast<const uint4 *>(plane_pyr + (long)(Y0 + row) * ppitch + X0 + c8);
		const unsigned u[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			const int c4 = c8 + 4 * h;
			const int4 w = make_int4((int)(short)(u[2 * h] & 0xffffu), (int)u[2 * h] >> 16, (int)(short)(u[2 * h + 1] & 0xffffu), (int)u[2 * h + 1] >> 16);
			*reinterpret_cast<int4 *>(lds + ((row << 5) | (c4 ^ ((row & 7) << 2)))) = w;
		}
	}
	unsigned pos[16];
	square_positions16(m, lane, pos);
	sq_wave_sync();
	const char *base = reinterpret_cast<const char *>(lds);
#pragma unroll
	for (int k = 0; k < 16; ++k)
		val[k] = *reinterpret_cast<const int *>(base + pos[k]);
	sq_wave_sync();
}

__device__ __forceinline__ void store_square16(short *__restrict__ plane_pyr, int ppitch, int n, int blk, int lane, unsigned *lds,
	const int (&val)[16])
{
	const SquareMap m = square_map(n, (unsigned)blk);
	const unsigned X0 = m.mx & ~31u, Y0 = m.my & ~31u;
	unsigned pos[16];
	square_positions16(m, lane, pos);
	char *base = reinterpret_cast<char *>(lds);
#pragma unroll
	for (int k = 0; k < 16; ++k)
		*reinterpret_cast<int *>(base + pos[k]) = val[k];
	sq_wave_sync();
#pragma unroll
	for (int it = 0; it < 2; ++it) {
		const int row = it * 16 + (lane >> 2), c8 = (lane & 3) * 8;
		unsigned u[4];
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			const int c4 = c8 + 4 * h;
			const int4 w = *reinterpret_cast<const int4 *>(lds + ((row << 5) | (c4 ^ ((row & 7) << 2))));
			u[2 * h] = ((unsigned)w.x & 0xffffu) | ((unsigned)w.y << 16);
			u[2 * h + 1] = ((unsigned)w.z & 0xffffu) | ((unsigned)w.w << 16);
		}
		*reinterpret_cast<uint4 *>(plane_pyr + (long)(Y0 + row) * ppitch + X0 + c8) = make_uint4(u[0], u[1], u[2], u[3]);
	}
	sq_wave_sync();
}

} // namespace

