// dwtx_internal.h — shared internals of libdwtx (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/dwtx.h"

#define DWTX_SCRATCH_SLOTS 24
#define DWTX_ENC_PARTS 4

struct dwtx_linplan;

struct dwtx_ctx {
	int device;
	hipStream_t stream;
	bool own_stream;
	void *scratch[DWTX_SCRATCH_SLOTS];
	size_t scratch_bytes[DWTX_SCRATCH_SLOTS];
	dwtx_linplan *plans;   // per-geometry Hilbert block tables (linearize.hip)
	hipStream_t aux;       // second stream: half of a decode batch runs here so that one half's serial
	hipEvent_t ev[4];      // token walk overlaps the other half's parallel kernels (unpack.hip); [2],[3]: bitmap clear
	bool have_aux;
	hipStream_t more[2];   // decode batches of small pictures run as four parts, one stream each (unpack.hip)
	hipEvent_t pev[8];     // [k] part k's chunk tables done, [4+k] part k's scatter done
	bool have_more;
	hipStream_t copy;      // host-buffer wrappers: transfers of one part of a batch overlap the kernels of another (codec.hip)
	hipEvent_t cev[6];
	bool have_copy;
	const dwtx_index *index_in;   // dwtx_ctx_set_index: sidecar indices offered to / asked from the decode calls
	dwtx_index *index_out;
	size_t index_base;            // entry of the current call's first image (the host pipeline decodes a batch in parts)
	long opt[DWTX_OPT_COUNT];     // dwtx_ctx_set_option: diagnostic switches (tests, tools), all 0 by default
	// dwtx_encode_device cuts a batch into parts that run on contexts of their own (stream + scratch each): one part's
	// memory-bound lifting overlaps the instruction-bound entropy stage of the part before (codec.hip)
	dwtx_ctx *enc_part[DWTX_ENC_PARTS];
	hipEvent_t enc_ev[2 * DWTX_ENC_PARTS + 1];   // [k] part k's transform is queued, [PARTS + k] part k is done, [2 * PARTS] the call's start
	bool have_enc_ev;
};

// Every entry point that allocates, launches or copies makes the context's device the calling thread's current
// one first: a host that drives several contexts (one per GPU) from one thread gets its kernels and scratch on the
// right device.  On a single-GPU process it is one hipGetDevice().
int dwtx_enter(dwtx_ctx *ctx);
#define DWTX_ENTER(ctx)              \
	do {                             \
		const int rc_ = dwtx_enter(ctx); \
		if (rc_)                     \
			return rc_;              \
	} while (0)
#ifdef DWTX_DEBUG_HOOKS
// debug builds: every kernel-launching function checks that it runs on its context's device
int dwtx_debug_check_device(dwtx_ctx *ctx, const char *file, int line);
#endif

void dwtx_free_plans(dwtx_ctx *ctx);
int dwtx_need_side_streams(dwtx_ctx *ctx, bool more);            // ctx.hip: creates aux / ev (and more / pev) on first use
// ctx.hip: part k's context (made on first use) with the parent's options.  Parts run on the context's own streams — the
// caller's, aux, more[0], more[1]: the runtime maps streams onto a handful of hardware queues, and streams that share a
// queue run one after the other; the encoder's and the decoder's parts therefore use the same four.
int dwtx_encoder_part(dwtx_ctx *ctx, int k, dwtx_ctx **part);
int dwtx_need_copy_stream(dwtx_ctx *ctx);   // creates ctx->copy / ctx->cev on first use

void dwtx_set_error(const char *fmt, ...);
// grow-only per-slot device scratch; contents undefined after a grow
void *dwtx_scratch(dwtx_ctx *ctx, int slot, size_t bytes);

#define DWTX_HIP(call)                                                          \
	do {                                                                        \
		hipError_t e_ = (call);                                                 \
		if (e_ != hipSuccess) {                                                 \
			dwtx_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
			return DWTX_ERR_DEVICE;                                             \
		}                                                                       \
	} while (0)

#ifdef DWTX_DEBUG_HOOKS
#define DWTX_LAUNCH_CHECK()                                            \
	do {                                                               \
		if (dwtx_debug_check_device(ctx, __FILE__, __LINE__))          \
			return DWTX_ERR_DEVICE;                                    \
		DWTX_HIP(hipGetLastError());                                   \
	} while (0)
#else
#define DWTX_LAUNCH_CHECK() DWTX_HIP(hipGetLastError())
#endif

static inline int dwtx_cdiv(int a, int b) { return (a + b - 1) / b; }

// Image sizes: sides of 8..DWTX_MAX_SIDE (above it the reference's own arithmetic overflows, include/dwtx.h), and the
// kernels' int indices want one plane (W*H) below 2^31 — which 32768 x 32768 = 2^30 always is; the check stays for
// whoever raises DWTX_MAX_SIDE (the reference itself indexes with int, encode.c:40 `channels*(width*y+x)`).
static inline bool dwtx_dims_ok(int W, int H)
{
	return W >= DWTX_MIN_LEN && H >= DWTX_MIN_LEN && W <= DWTX_MAX_SIDE && H <= DWTX_MAX_SIDE && (long)W * H <= 0x7fffffffL - 4096;
}
#define DWTX_CHECK_DIMS(W, H)                                                              \
	do {                                                                                    \
		if (!dwtx_dims_ok(W, H)) {                                                          \
			dwtx_set_error("unsupported image size %dx%d (8..%d per side)", W, H, DWTX_MAX_SIDE); \
			return DWTX_ERR_ARG;                                                            \
		}                                                                                   \
	} while (0)

// scratch slot assignment
enum {
	SLOT_LIFT_A = 0,
	SLOT_LIFT_B = 1,
};

// lift.hip: the finest lifting level reads / writes 8-bit pixels itself (the widening of pnm.h:69-74, the
// clamp of pnm.h:108 and, for RGB, the YCoCg-R colour transform of image.h:39-65 fused into it).  dwtx_gray8_ok says whether the
// shape allows it; image i of the inverse is written at pix + i*image_stride.
bool dwtx_gray8_ok(int W, int H, const void *pix, size_t image_stride);
int dwtx_fwd_pixels8(dwtx_ctx *ctx, int32_t *out, const uint8_t *pix, int W, int H, int C, int n);   // C = 3: YCoCg-R fused too (image.h:52-65)
struct dwtx_p16;
int dwtx_inv_pixels8(dwtx_ctx *ctx, uint8_t *pix, size_t image_stride, const int32_t *in, int W, int H, int C, int n, const dwtx_p16 *p16 = nullptr);   // C = 3: image.h:39-50 fused too

// The entropy stage's tiles (linearize.hip): the Hilbert curve of ring level l visits every aligned 32x32 square of
// its lengths[l+1]-sided square contiguously ("curve block"), so the ring's coefficients, in the order of
// encode.c:46-56, are the blocks' points one block after the other.  A tile is one non-empty curve block: `base` =
// ring index of its first coefficient, `cnt` = its coefficients (1024 for a block that lies wholly inside the ring),
// `blk` = the block's index on the curve.  Levels below 32x32 are one block each.  Tiles are numbered level by level.
struct dwtx_tiles {
	int NT;
	int tile_first[DWTX_MAX_LEVELS + 1];
	const int *base;              // device, [NT]
	const unsigned short *cnt;    // device, [NT]
	const int *blk;               // device, [NT]
	// block (bx, by) of level l's curve square (32x32 pyramid positions each, lengths[l+1] / 32 = nbs[l] blocks per side)
	// -> its tile, -1 for a block without ring coefficients: xy2tile[xy_first[l] + by * nbs[l] + bx]; levels below 64 have none
	const int *xy2tile;           // device
	int xy_first[DWTX_MAX_LEVELS + 1];
	int nbs[DWTX_MAX_LEVELS];
};

// Where the forward transform drops the tiles' magnitude histograms while it still holds the coefficients in registers
// (lift.hip k_fwd_level_w; what k_hist would otherwise read them from memory again for).  cum32: [plane][NT][16] words,
// word b of a tile = #(|v| < 2^(2b)) | #(|v| < 2^(2b+1)) << 16, zero before the transform adds to them; tile_mx:
// [plane][NTP] OR of the tile's magnitudes.
struct dwtx_hist_sink {
	unsigned *cum32;
	unsigned *tile_mx;
	int NT, NTP;
	dwtx_tiles tiles;
};
int dwtx_hist_begin(dwtx_ctx *ctx, int W, int H, int C, int n, dwtx_hist_sink *sink);   // pack.hip
// lift.hip: the forward transform with the histograms of the levels it can take (returned in *hist_levels, bit l = ring level l)
// p16 (optional, everywhere below): the detail coefficients of the ring levels in `levels` (a mask; the finest levels)
// are kept as 16-bit values in planes of their own — [n*C][H][W] int16, the positions of the pyramid — instead of in the
// int32 pyramid, whose positions for those rings are then never touched.  An 8-bit source cannot leave 16 bits on
// its finest ring (|HL|, |LH|, |HH| <= 1020: cdf53.h:13-21 on samples of magnitude <= 255) — three quarters of
// everything the entropy stage reads and writes; a decoder may keep every ring that way whose streams claim at most 15 bit planes.
struct dwtx_p16 {
	int16_t *planes;
	unsigned levels;
};
int dwtx_fwd_pixels8_hist(dwtx_ctx *ctx, int32_t *out, const uint8_t *pix, int W, int H, int C, int n, const dwtx_hist_sink *sink, unsigned *hist_levels,
	dwtx_p16 p16 = dwtx_p16{ nullptr, 0u });
int dwtx_transformation_fwd_hist(dwtx_ctx *ctx, int32_t *out, const int32_t *in, int W, int H, int nplanes, const dwtx_hist_sink *sink,
	unsigned *hist_levels);
int dwtx_get_tiles(dwtx_ctx *ctx, int W, int H, dwtx_tiles *out);

// Tiles straight from / to the pyramid (hilbert_dev.h): on the ring levels in the mask, tiles that are whole 32x32
// squares need no linearised copy — the entropy stage reads (pack.hip) / writes (unpack.hip) them in the pyramid itself;
// only the blocks the ring's edges cut (image border, LL quadrant) still go through `lin`.
unsigned dwtx_square_levels(int W, int H);
// the up to `max_levels` finest ring levels that may live in 16-bit planes: whole squares read / written in place (in
// sq_levels) and transformed by the 16-byte-per-lane lifting kernels; 0 if the finest one does not qualify
unsigned dwtx_levels16(int W, int H, unsigned sq_levels, int max_levels);
int dwtx_linearization_ex(dwtx_ctx *ctx, int32_t *lin, const int32_t *pyr, int W, int H, int nplanes, unsigned skip_levels,
	dwtx_p16 p16 = dwtx_p16{ nullptr, 0u });
int dwtx_reconstruction_ex(dwtx_ctx *ctx, int32_t *pyr, const int32_t *lin, const int *dev_missing, int levels_out, int W, int H,
	int C, int n, unsigned skip_levels, dwtx_p16 p16 = dwtx_p16{ nullptr, 0u });
// hist_levels: ring levels whose tile histograms the forward transform has already written (dwtx_hist_begin)
int dwtx_encode_planes_ex(dwtx_ctx *ctx, const int32_t *lin, const int32_t *pyr, unsigned sq_levels, unsigned hist_levels, int W, int H, int C, int n,
	long capacity, uint8_t *out, size_t out_stride, dwtx_stream_info *dev_info, dwtx_p16 p16 = dwtx_p16{ nullptr, 0u });

// unpack.hip: dwtx_decode_planes with a host callback per finished part of the batch (see there)
// `pyr` (optional): pyramid planes [n*C][H][W]; for parts of the batch that decode at full resolution the tiles of
// the full-square ring levels are written there (bias included) and `done` is told which levels (fused_levels).
constexpr unsigned DWTX_FUSED_FINE16 = 1u << 31;   // in `fused_levels`: the part's rings of p16.levels were written to the 16-bit planes, not to pyr
int dwtx_decode_planes_ex(dwtx_ctx *ctx, int32_t *lin, int32_t *pyr, const uint8_t *streams, size_t stream_stride,
	const unsigned long long *dev_lens, int W, int H, int C, int n, int levels_max, dwtx_decode_info *host_info,
	int (*done)(void *user, int first, int count, unsigned fused_levels), void *user, dwtx_p16 p16 = dwtx_p16{ nullptr, 0u });

// The wave's lanes for which `pred` holds.  (HIP's __ballot() takes an int: the condition would be turned
// into 0/1 in a register and compared again — two extra instructions per use in the ballot-heavy kernels.)
__device__ __forceinline__ unsigned long long ballot64(bool pred) { return __builtin_amdgcn_ballot_w64(pred); }

// C truncating division by 2 and 4 on the device (cdf53.h:13,20 use `/`)
__device__ __forceinline__ int tdiv2(int a) { return (a + (int)((unsigned)a >> 31)) >> 1; }
__device__ __forceinline__ int tdiv4(int a) { return (a + ((a >> 31) & 3)) >> 2; }
