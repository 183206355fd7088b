/*
 * dwt_oracle.h — CPU ORACLE for the xdsopl/dwt encode/decode hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke
 * check in __graft_entry__.py and the cpu_baseline leg of bench.py may link or
 * call anything in oracle/.  The product (dwt_amd/, include/dwtx.h) never does.
 *
 * It is a from-scratch restatement, in plain single-threaded C, of what the
 * reference computes (reference = /root/reference, cited as file:line in the
 * .c file).  Parity status: PINNED — the restatement is byte-compared against
 * the real reference built from its own sources into oracle/_ref/ (see
 * oracle/Makefile, target `ref`) on the fixture sweep in tests/, and against
 * the committed goldens under tests/golden/ that the same reference produced.
 */
#ifndef DWT_ORACLE_H
#define DWT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_LEVELS 16
/* Largest side the oracle (and the product) accepts.  encode.c:140 lets sides up to 65536 through, but for a side above
 * 32768 the finest level's Hilbert square is 65536 wide and encode.c:45 / decode.c:47 compute `lengths * lengths` in
 * int: the product wraps to 0, the finest ring is never visited, and the reference binary writes a stream that its own
 * decoder does not turn back into the picture (pinned in tests/test_oracle.py).  "What the reference computes" is
 * therefore only defined up to 32768 per side; above that both the oracle and the library refuse (return 1 /
 * DWTX_ERR_ARG) — the second documented difference, DESIGN.md section 7. */
#define ORC_MAX_SIDE 32768

/* Level geometry (utils.h:17-40).  Index 0 = root LL, index `levels` = full image. */
typedef struct {
	int levels;
	int widths[ORC_MAX_LEVELS];
	int heights[ORC_MAX_LEVELS];
	int pixels[ORC_MAX_LEVELS];
	int lengths[ORC_MAX_LEVELS];
} orc_geom;

typedef struct {
	int meta_bits;      /* encode.c:175-176 */
	int root_bits;      /* encode.c:179-180 */
	int total_bits;     /* encode.c:226,230 (bit count before padding) */
	int kib;            /* encode.c:228 */
	int planes[3];
	int levels;
	long tokens;        /* VLI tokens emitted (diagnostic) */
	long raw_bits;      /* raw sign/refinement bits emitted (diagnostic) */
} orc_stats;

int orc_ilog2(int x);
int orc_geometry(orc_geom *g, int W, int H, int min_len);

/* 1-D lifting on a contiguous line of n samples (cdf53.h:9-61). */
void orc_fwd53_line(int *x, int n, int *scratch);
void orc_inv53_line(int *x, int n, int *scratch);

/* Multi-level 2-D transform, interleaved [H][W*C] int image, in place.
 * fwd: pixels -> Mallat pyramid (encode.c:16-30); inv: mirror (decode.c:16-30). */
void orc_forward(int *img, int W, int H, int C, int min_len);
void orc_inverse(int *img, int W, int H, int C, int min_len);

/* Colour transform (image.h:39-65), interleaved triples, n pixels. */
void orc_rgb_to_ycocg(int *img, long n);
void orc_ycocg_to_rgb(int *img, long n);   /* with the decoder's clamps */

void orc_hilbert(int n, int d, int *x, int *y);   /* hilbert.h:15-34 */

/* Pyramid (interleaved) -> planar Hilbert-linearised [C][total] (encode.c:32-58). */
void orc_linearize(int *lin, const int *pyr, const orc_geom *g, int C);
/* Inverse, with truncation bias (decode.c:32-65).  `levels` may be < g->levels. */
void orc_reconstruct(int *pyr, int *const *lin, const int *missing, const orc_geom *g, int levels, int C);

/* Whole-file encode: 8-bit interleaved pixels -> .dwt bytes (incl. 6-byte header).
 * capacity <= 0 means unlimited.  *out is malloc'ed.  Returns 0, or 1 on bad size. */
int orc_encode(const uint8_t *pix, int W, int H, int C, long capacity,
	uint8_t **out, size_t *out_len, orc_stats *st);
/* the entropy stage of orc_encode alone, on linearised coefficient planes lin[C][W*H] */
int orc_encode_lin(const int *lin, int W, int H, int C, long capacity,
	uint8_t **out, size_t *out_len, orc_stats *st);

/* Whole-file decode.  pixels_max < 0 means "no PIXELS argument".  *pix is
 * malloc'ed (already clamped to 0..255 like pnm.h:108).  Returns 0 or 1. */
int orc_decode(const uint8_t *dwt, size_t len, long pixels_max,
	uint8_t **pix, int *W, int *H, int *C);

/* Entropy stage of the decoder only (decode.c:174-250): lin = int[C][W*H]
 * two's complement, zero beyond what the stream covered; *level = finest level
 * touched (-1 none); missing = int[48] ([c*16+l]); planes = int[3]. */
int orc_decode_stage(const uint8_t *dwt, size_t len, long pixels_max, int *lin, int *level, int *missing, int *planes);

/* Stage dumps used by the per-kernel parity tests.
 * coef: int[C*W*H] interleaved pyramid after colour+forward transform.
 * lin : int[C][W*H] planar, raw two's-complement values (before sign-magnitude). */
int orc_stage_dump(const uint8_t *pix, int W, int H, int C, int *coef, int *lin, int *planes);

/* Integer-only synthetic image generator. kind 0 = smooth+noise, 1 = uniform noise. */
void orc_synth(uint8_t *pix, int W, int H, int C, uint32_t seed, int kind);

#ifdef __cplusplus
}
#endif
#endif
