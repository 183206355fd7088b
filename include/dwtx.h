/*
 * dwtx.h — C ABI of the MI355X-native encode/decode hot path of xdsopl/dwt.
 *
 * The reference (/root/reference) has no FFI layer: its hot path is a set of
 * header-defined C functions called from two main()s.  This header declares the
 * entry points a maintainer would bind in their place; each cites the
 * reference call site / function it replaces.  Plain C types only: device and
 * host buffers are raw pointers, sizes are ints/size_t, errors are negative
 * ints like the reference's (-1 I/O or EOF, -2 capacity; see bytes.h:75-105).
 *
 * Conventions
 *   - `dev` pointers are HIP device pointers (from dwtx_malloc or any HIP
 *     allocator, e.g. a torch tensor's data_ptr()).  `host` pointers are plain
 *     host memory.
 *   - Images inside the library are PLANAR int32: plane p = image*C + channel,
 *     each plane H rows of W ints, row pitch W (dense).  The reference keeps
 *     interleaved int buffers (image.h:12-15); dwtx_planes_from_pixels /
 *     dwtx_pixels_from_planes convert at the edge, fused with the colour
 *     transform.
 *   - All functions are asynchronous on the context's stream unless they
 *     return data to the host; dwtx_sync() waits.
 *   - Thread-compatible: one context per host thread.
 */
#ifndef DWTX_H
#define DWTX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DWTX_OK            0
#define DWTX_ERR_IO       -1   /* bytes.h:79-83,99-103 */
#define DWTX_ERR_CAPACITY -2   /* bytes.h:77-78 */
#define DWTX_ERR_ARG      -3   /* encode.c:139-146, decode.c:145-159 (exit code 1 there) */
#define DWTX_ERR_DEVICE   -4   /* HIP runtime failure; dwtx_last_error() has the text */
#define DWTX_ERR_NOMEM    -5

#define DWTX_MAX_LEVELS 16
#define DWTX_MIN_LEN     8     /* encode.c:144, decode.c:157 */
/* Largest image side.  encode.c:140 lets sides up to 65536 through, but above 32768 the finest level's Hilbert square is
 * 65536 wide and encode.c:45 / decode.c:47 compute `lengths[l+1] * lengths[l+1]` in int: it wraps to 0, the finest ring
 * is never visited and the reference binary writes a stream its own decoder does not turn back into the picture.  There
 * is nothing to be bit-exact with beyond this side, so every entry point refuses it (DWTX_ERR_ARG; the CLIs exit 1 with
 * a message) — a documented difference, DESIGN.md section 7. */
#define DWTX_MAX_SIDE    32768

typedef struct dwtx_ctx dwtx_ctx;

/* Level geometry, utils.h:17-40.  Index 0 = root LL, index `levels` = full image. */
typedef struct dwtx_geom {
	int levels;
	int widths[DWTX_MAX_LEVELS];
	int heights[DWTX_MAX_LEVELS];
	int pixels[DWTX_MAX_LEVELS];
	int lengths[DWTX_MAX_LEVELS];
} dwtx_geom;

/* encode.c:175-180,226-230 stderr counters */
typedef struct dwtx_stats {
	int meta_bits;
	int root_bits;
	int total_bits;
	int kib;
	int levels;
	int planes[3];
} dwtx_stats;

/* Per-image result record of the entropy stage (device or host memory). */
typedef struct dwtx_stream_info {
	int planes[3];                 /* encode.c:163-165 */
	int pmax;
	int segments;                  /* (channel, level, plane) segments coded, encode.c:183-221 (all of the schedule unless CAPACITY cut it) */
	int entries;
	unsigned tokens;               /* VLI token slots (incl. void ones) */
	int order0;                    /* VLI order after header, root image and plane counts */
	unsigned hdr_bits;             /* 48 header bits + root image + plane counts */
	unsigned root_bits;            /* encode.c:179-180, as the reference counts it (CAPACITY can cut into the root image) */
	unsigned meta_bits;            /* encode.c:175-176: 48 unless CAPACITY < 6 */
	unsigned segments_cut;         /* segments of the schedule left uncoded because they start beyond CAPACITY (encode.c:192,204,216) */
	unsigned long long total_bits; /* encode.c:226: bit count before padding (8*capacity when truncated) */
	unsigned long long nbytes;     /* bytes of the .dwt stream: min(capacity, ceil(total_bits/8)) */
	int error;                     /* non-zero: unsupported data (more than 16 bit planes) */
	int exact_orders;              /* 1: the fast VLI-order pass did not resolve this image, the exact one ran */
} dwtx_stream_info;

/* Per-image result record of the decoder's entropy stage (host memory). */
typedef struct dwtx_decode_info {
	int status;                    /* 0 ok; 1 = header, root image or plane counts unreadable (decode.c exits 1);
	                                * 2 = the stream claims more than 16 bit planes: refused (decode.c:183-186 would go on;
	                                * only damage produces such a count — the one documented difference) */
	int W, H, C;
	int levels;
	int planes[3];                 /* decode.c:183-186 */
	int pmax;
	int level;                     /* finest level any segment touched (decode.c:197,203,219,236); -1 = none */
	int nsegs;
	int truncated;                 /* bit 0: the walk stopped early (end of data or PIXELS cap); bit 1: a read ran past
	                                * the end of the data (where decode prints bytes.h:101 "reached end of file") */
	int missing[48];               /* decode.c:193-196: planes not fully decoded, [channel*16 + level] */
	unsigned long long bits_used;
	unsigned hops, hopped_chunks;  /* token walker statistics: jumps over stitched 128-bit chunks */
	unsigned walked_tokens;        /* tokens the walker parsed itself */
	unsigned zeros_left;           /* run-length counter at the end; decode prints rle.h:45 "%d zeros not read." if > 1 */
} dwtx_decode_info;

/* ---- sidecar index ---------------------------------
 * Not part of the reference and never inside a .dwt: an optional companion that records, for every
 * (channel, level, plane) segment of decode.c:198-243's schedule, the decoder's state where the segment's first
 * pass begins (decode.c:67-100: stream position, vli.h:24 order, rle.h:25 run counter, how many coefficients
 * are still insignificant).  A decode that is given the index walks all segments at once instead of one after
 * the other; it checks on the way that the segments fit together, so a wrong, stale or foreign index cannot
 * change the result: the decoder then falls back to the plain walk.  A decode produces the index of every stream
 * it decodes to its end. */
#define DWTX_INDEX_MAGIC 0x49545744u   /* "DWTI" */
#define DWTX_INDEX_MAX_SEGS 768         /* 3 channels x 16 levels x 16 planes */
typedef struct dwtx_seg_index {
	unsigned long long bit;            /* stream position of the segment's first pass */
	unsigned long long sym_base;       /* decoder-internal: first symbol slot of the segment */
	unsigned n1;                       /* symbols of the first pass */
	unsigned cnt;                      /* rle.h:25 on entry */
	unsigned desc;                     /* channel | level << 4 | (plane + 1) << 8 */
	unsigned order;                    /* vli.h:24 on entry */
} dwtx_seg_index;
typedef struct dwtx_index {
	unsigned magic;                    /* DWTX_INDEX_MAGIC */
	int W, H, C;
	int nsegs;                         /* 0: no index (stream unreadable, cut short, or decoded with a PIXELS cap) */
	int reserved;
	unsigned long long stream_bits;    /* bits of the stream the decoder used */
	dwtx_seg_index seg[DWTX_INDEX_MAX_SEGS];
} dwtx_index;

/* ---- context / memory ---------------------------------------------------- */

/* Create a context on HIP device `device` with a stream of its own. */
int dwtx_ctx_create(int device, dwtx_ctx **ctx);
/* Same, but run on an existing hipStream_t (e.g. torch's current stream);
 * NULL means the device's default stream. */
int dwtx_ctx_create_on_stream(int device, void *stream, dwtx_ctx **ctx);
void dwtx_ctx_destroy(dwtx_ctx *ctx);
const char *dwtx_last_error(void);
int dwtx_sync(dwtx_ctx *ctx);
/* Sidecar indices for the decode calls that follow on this context (dwtx_decode_planes / _device / _images):
 * entry i of `in` (may be NULL) is offered for image i of a call, entry i of `out` (may be NULL) receives the
 * index of image i (nsegs = 0 if it has none).  Both are host arrays that must stay valid until replaced;
 * dwtx_ctx_set_index(ctx, NULL, NULL) ends it. */
int dwtx_ctx_set_index(dwtx_ctx *ctx, const dwtx_index *in, dwtx_index *out);
void *dwtx_stream(dwtx_ctx *ctx);

/* Diagnostic switches of a context, all off (0) by default.  They exist for the tests, the profiling tools and
 * the CLIs' debugging aids: none of them changes a result, they choose between code paths that must agree
 * (DESIGN.md section 8).  The library itself reads no environment variables. */
enum dwtx_option {
	DWTX_OPT_EXACT_ORDERS = 0,     /* encoder: every image takes the exact 32-state VLI-order pass */
	DWTX_OPT_NO_SQUARE_TILES,      /* no tiles straight from / to the pyramid: everything through the linearised copy */
	DWTX_OPT_PART_IMAGES,          /* host-buffer pipelines: images per part (0 = automatic) */
	DWTX_OPT_ONE_STREAM,           /* decoder: the whole batch on one HIP stream (clean per-kernel profiles) */
	DWTX_OPT_DECODE_PARTS,         /* decoder: parts a batch is cut into, 2..4 (0 = automatic) */
	DWTX_OPT_TWO_FAMILIES,         /* decoder: both speculative path families from the start */
	DWTX_OPT_NO_SECOND_WALK,       /* decoder: a token walk that gives up is an error instead of being repeated */
	DWTX_OPT_NO_INDEX,             /* decoder: offered sidecar indices are ignored */
	DWTX_OPT_NO_INDEX_FALLBACK,    /* decoder: an index that is turned down is an error instead of the serial walk */
	DWTX_OPT_NO_CAPACITY_CUT,      /* encoder: CAPACITY only clips the finished stream (all segments are coded) */
	DWTX_OPT_NO_FINE16,            /* the finest ring stays in the int32 pyramid (no 16-bit planes for it) */
	DWTX_OPT_NO_FUSED_LEVELS,      /* transforms on int32 planes: one launch per level (no two-levels-per-pass kernels) */
	DWTX_OPT_COUNT
};
int dwtx_ctx_set_option(dwtx_ctx *ctx, int option, long value);
long dwtx_ctx_get_option(dwtx_ctx *ctx, int option);

void *dwtx_malloc(dwtx_ctx *ctx, size_t bytes);
void dwtx_free(dwtx_ctx *ctx, void *dev);
/* Page-locked host memory for the host-buffer entry points below: their transfers then overlap the
 * kernels of the previous part of the batch (pageable buffers work too, only slower). */
void *dwtx_host_alloc(dwtx_ctx *ctx, size_t bytes);
void dwtx_host_free(dwtx_ctx *ctx, void *host);
int dwtx_upload(dwtx_ctx *ctx, void *dev, const void *host, size_t bytes);
int dwtx_download(dwtx_ctx *ctx, void *host, const void *dev, size_t bytes);

/* ---- host-side geometry --------------------------------------------------- */

/* utils.h:28-40 compute_lengths(): same argument order and return value. */
int dwtx_compute_lengths(int *lengths, int *pixels, int *widths, int *heights, int W, int H, int N0);
int dwtx_geometry(dwtx_geom *g, int W, int H);

/* ---- stage kernels (device buffers, batches of n images) ------------------ */

/* Benchmark/test utility (no reference counterpart): render n synthetic 8-bit
 * frames [n][H][W][C] on the device with the integer-only generator of
 * SURVEY.md §8d; frame i uses seed seed0+i.  kind 0 = smooth+noise, 1 = noise. */
int dwtx_synth_pixels(dwtx_ctx *ctx, uint8_t *dev_pix, int W, int H, int C, int n, unsigned seed0, int kind);

/* pnm.h:69-74 widening + image.h:67-72 ycocg_from_rgb (C==3): interleaved
 * 8-bit pixels [n][H][W][C] -> planar int32 [n*C][H][W]. */
int dwtx_planes_from_pixels(dwtx_ctx *ctx, int32_t *dev_planes, const uint8_t *dev_pix, int W, int H, int C, int n);
/* image.h:74-79 rgb_from_ycocg (with its clamps, image.h:41-43) + pnm.h:108 clamp. */
int dwtx_pixels_from_planes(dwtx_ctx *ctx, uint8_t *dev_pix, const int32_t *dev_planes, int W, int H, int C, int n);

/* encode.c:16-30 transformation(): multi-level forward CDF 5/3 of `nplanes`
 * planar W*H images.  dev_in is preserved; dev_out receives the Mallat pyramid. */
int dwtx_transformation_fwd(dwtx_ctx *ctx, int32_t *dev_out, const int32_t *dev_in, int W, int H, int nplanes);
/* decode.c:16-30 transformation(): inverse.  dev_in (pyramid) is preserved. */
int dwtx_transformation_inv(dwtx_ctx *ctx, int32_t *dev_out, const int32_t *dev_in, int W, int H, int nplanes);

/* The same two transforms as the whole-image pipelines run them when the pictures are 8-bit pixels with W % 4 == 0 and more
 * than 64 pixels on a side (else DWTX_ERR_ARG: use the two calls above): encode.c:155-159 — widening, ycocg_from_rgb and
 * transformation() — in one pass over interleaved pixels [n][H][W][C], the finest level in packed 16-bit arithmetic; and
 * decode.c:258-264 — transformation(), rgb_from_ycocg with its clamps and write_pnm's clamp.  The detail rings of the up to
 * five finest levels (those in *levels16 / levels16, bit l = ring level l) are kept as int16 in dev_rings16 [n*C][H][W] —
 * same positions and pitch as in the pyramid, whose positions for those rings are then not touched; results are the
 * int32 transform's (an 8-bit source cannot leave 16 bits there: DESIGN.md section 4.1).  dev_rings16 == NULL: everything
 * in dev_pyr [n*C][H][W] int32.  What bench.py's `roofline_codec` times. */
int dwtx_transformation_fwd_pixels(dwtx_ctx *ctx, int32_t *dev_pyr, int16_t *dev_rings16, unsigned *levels16,
	const uint8_t *dev_pix, int W, int H, int C, int n);
int dwtx_transformation_inv_pixels(dwtx_ctx *ctx, uint8_t *dev_pix, const int32_t *dev_pyr, const int16_t *dev_rings16,
	unsigned levels16, int W, int H, int C, int n);

/* encode.c:32-58 linearization(): Mallat pyramid planes [nplanes][H][W] ->
 * Hilbert-linearised planes [nplanes][W*H] (root raster first, then the detail
 * ring of each level in curve order, hilbert.h:15-34). */
int dwtx_linearization(dwtx_ctx *ctx, int32_t *dev_lin, const int32_t *dev_pyr, int W, int H, int nplanes);
/* decode.c:32-65 reconstruction(): the inverse scatter, for the first
 * `levels_out` levels only (output planes are widths[levels_out] x
 * heights[levels_out], dense).  dev_missing is NULL or int[n][3][16] as in
 * decode.c:193-196 (planes not decoded per channel and level -> dequantisation
 * bias, decode.c:51-58). */
int dwtx_reconstruction(dwtx_ctx *ctx, int32_t *dev_pyr, const int32_t *dev_lin, const int *dev_missing,
	int levels_out, int W, int H, int C, int n);

/* encode.c:166-221: header, root image, plane counts, bit-plane segments in
 * schedule order, final run flush — for n images at once.  dev_lin is the
 * output of dwtx_linearization ([n*C][W*H], two's complement).  Image i's
 * stream is written to dev_out + i*out_stride (out_stride a multiple of 4; at
 * most out_stride bytes are ever written, so it should be >= capacity when
 * capacity > 0).  capacity <= 0 means unlimited (encode.c:150-152).
 * dev_info[i].nbytes is the length of stream i. */
int dwtx_encode_planes(dwtx_ctx *ctx, const int32_t *dev_lin, int W, int H, int C, int n, long capacity,
	uint8_t *dev_out, size_t out_stride, dwtx_stream_info *dev_info);

/* decode.c:174-250: root image, plane counts and all bit-plane segments of n
 * streams of identical geometry (W, H, C as in their headers) into linearised
 * two's-complement planes dev_lin [n*C][W*H] (zero where the stream ended
 * early).  Stream i occupies dev_streams + i*stream_stride (stride a multiple
 * of 8), its byte length is dev_lens[i].  levels_max < 0 = all levels
 * (decode.c:163-171 computes it from the PIXELS argument).  Synchronous:
 * host_info[i] is filled on return (level and missing[] drive
 * dwtx_reconstruction / dwtx_transformation_inv). */
int dwtx_decode_planes(dwtx_ctx *ctx, int32_t *dev_lin, const uint8_t *dev_streams, size_t stream_stride,
	const unsigned long long *dev_lens, int W, int H, int C, int n, int levels_max, dwtx_decode_info *host_info);

/* ---- whole-image pipelines (batches of n same-geometry images) -------------- */

/* A safe out_stride for unlimited-capacity encodes of W*H*C images (multiple of 8). */
size_t dwtx_encode_bound(int W, int H, int C);

/* encode.c:155-221 with everything resident in HBM: 8-bit interleaved pixels
 * [n][H][W][C] -> n streams at dev_out + i*out_stride.  Asynchronous. */
int dwtx_encode_device(dwtx_ctx *ctx, const uint8_t *dev_pix, int W, int H, int C, int n, long capacity,
	uint8_t *dev_out, size_t out_stride, dwtx_stream_info *dev_info);

/* decode.c:174-264 with everything resident in HBM.  Image i is written densely
 * at dev_pix + i*pix_stride with the size the stream supports
 * (widths/heights[host_info[i].level + 1], decode.c:251-254).  Synchronises once
 * (after the token walk) to learn that size. */
int dwtx_decode_device(dwtx_ctx *ctx, const uint8_t *dev_streams, size_t stream_stride,
	const unsigned long long *dev_lens, int W, int H, int C, int n, int levels_max,
	uint8_t *dev_pix, size_t pix_stride, dwtx_decode_info *host_info);

/* The sender's side of the one exchange step between GPUs (the encoded streams of a step travel to one
 * rank; no reference counterpart — the reference writes one file per process): the n streams of a batch, stream i at
 * dev_streams + i*stream_stride with dev_lens[i] bytes (as dwtx_encode_device leaves them), are moved together into ONE
 * contiguous buffer, stream i at byte offset sum over j < i of round8(dev_lens[j]) — so that a step's streams travel as one
 * message per peer instead of one per frame.  dev_offsets (optional, [n + 1]) receives the offsets, [n] = the total; the
 * receiver computes the same offsets from the gathered lengths.  A length beyond the stride is clamped to it; nothing is
 * written beyond out_bytes (size it from the lengths: the sum of the rounded lengths, at most n * stream_stride).
 * Asynchronous on the context's stream. */
int dwtx_pack_streams(dwtx_ctx *ctx, uint8_t *dev_out, size_t out_bytes, unsigned long long *dev_offsets,
	const uint8_t *dev_streams, size_t stream_stride, const unsigned long long *dev_lens, int n);

/* Host-buffer wrappers: what encode.c:133-232 / decode.c:136-268 do between
 * read_pnm/write_pnm and the byte sink.  pixels_max < 0 = no PIXELS argument.
 * dwtx_decode_images returns DWTX_ERR_ARG for a bad header and DWTX_ERR_IO when
 * the root image or plane counts cannot be read (both exit code 1 in decode.c); a
 * single stream that claims more than 16 bit planes is DWTX_ERR_ARG too (status 2). */
int dwtx_encode_images(dwtx_ctx *ctx, const uint8_t *host_pix, int W, int H, int C, int n, long capacity,
	uint8_t *host_out, size_t out_stride, size_t *out_lens, dwtx_stats *stats);
int dwtx_decode_images(dwtx_ctx *ctx, const uint8_t *host_streams, size_t stream_stride, const size_t *lens, int n,
	int pixels_max, uint8_t *host_pix, size_t pix_stride, int *outW, int *outH, int *outC);
/* Same, and copies the n decoder records to `infos` (may be NULL): what decode.c needs for its
 * stderr diagnostics (bytes.h:101, rle.h:45). */
int dwtx_decode_images_info(dwtx_ctx *ctx, const uint8_t *host_streams, size_t stream_stride, const size_t *lens, int n,
	int pixels_max, uint8_t *host_pix, size_t pix_stride, int *outW, int *outH, int *outC, dwtx_decode_info *infos);

#ifdef __cplusplus
}
#endif
#endif
