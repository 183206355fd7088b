// codec.hip — whole-image entry points: the device pipelines behind the
// reference's two main()s (encode.c:133-232, decode.c:136-268 minus file I/O),
// for batches of same-geometry images.
#include "dwtx_internal.h"

#include <stdlib.h>
#include <string.h>

enum { SLOT_CD_A = 16, SLOT_CD_B, SLOT_CD_INFO, SLOT_CD_IO, SLOT_CD_IO2, SLOT_CD_LENS, SLOT_CD_F16 };

extern "C" size_t dwtx_encode_bound(int W, int H, int C)
{
	// 8-bit sources: at most 11 bit planes per coefficient (9 bits of pixel range, +1 for YCoCg chroma, +1
	// for the HH gain of the 5/3 lifting), each plane costs a coefficient at most 2 bits (a raw refinement
	// bit, or a pass-1 symbol: VLI(0) at order 0 is one bit, plus the sign) -> below 3 bytes per sample;
	// uniform noise measures ~9 bit/sample
	size_t b = (size_t)3 * W * H * C + 4096;
	return (b + 7) / 8 * 8;
}

// pixels (device) -> streams (device) for one part of a batch, on the part's context; `lifted` (optional) is recorded
// once the part's transform and linearisation are queued
static int encode_part(dwtx_ctx *ctx, const uint8_t *dev_pix, int W, int H, int C, int n, long capacity,
	uint8_t *dev_out, size_t out_stride, dwtx_stream_info *dev_info, hipEvent_t lifted)
{
	const size_t bytes = sizeof(int) * (size_t)W * H * C * n;
	int *a = (int *)dwtx_scratch(ctx, SLOT_CD_A, bytes);
	int *b = (int *)dwtx_scratch(ctx, SLOT_CD_B, bytes);
	if (!a || !b)
		return DWTX_ERR_NOMEM;
	int rc;
	// the forward transform leaves the tiles' magnitude histograms behind where it can (encode.c:112-131's maximum and
	// everything the entropy stage counts before it codes): the coefficients are not read a second time for them
	dwtx_hist_sink sink;
	unsigned hist_levels = 0;
	if ((rc = dwtx_hist_begin(ctx, W, H, C, n, &sink)))
		return rc;
	// encode.c:160: levels that are full power-of-two squares stay in the pyramid (the coder reads their tiles there)
	const unsigned sq = ctx->opt[DWTX_OPT_NO_SQUARE_TILES] ? 0u : dwtx_square_levels(W, H);
	// The finest rings (up to five levels) as 16-bit values in planes of their own when the transform starts from 8-bit
	// pixels — the finest ring, three quarters of all coefficients, cannot leave 11 bits then, the fifth level's not 15
	// (lift.hip k_fwd_level_w) — and the coder reads their squares in place: the transform writes, and the coder reads,
	// half the bytes for them.
	dwtx_p16 fine16 = { nullptr, 0u };
	const bool from_pixels = dwtx_gray8_ok(W, H, dev_pix, (size_t)W * H * C);
	if (from_pixels && sq && !ctx->opt[DWTX_OPT_NO_FINE16] && (fine16.levels = dwtx_levels16(W, H, sq, 5))) {
		fine16.planes = (int16_t *)dwtx_scratch(ctx, SLOT_CD_F16, sizeof(int16_t) * (size_t)W * H * C * n);
		if (!fine16.planes)
			return DWTX_ERR_NOMEM;
	}
	if (from_pixels) {
		if ((rc = dwtx_fwd_pixels8_hist(ctx, b, dev_pix, W, H, C, n, &sink, &hist_levels, fine16)))   // encode.c:155-159 in one pass
			return rc;
	} else {
		if ((rc = dwtx_planes_from_pixels(ctx, a, dev_pix, W, H, C, n)))       // encode.c:155-156
			return rc;
		if ((rc = dwtx_transformation_fwd_hist(ctx, b, a, W, H, n * C, &sink, &hist_levels)))   // encode.c:159
			return rc;
	}
	if ((rc = dwtx_linearization_ex(ctx, a, b, W, H, n * C, sq, fine16)))
		return rc;
	if (lifted)
		DWTX_HIP(hipEventRecord(lifted, ctx->stream));
	return dwtx_encode_planes_ex(ctx, a, b, sq, hist_levels, W, H, C, n, capacity, dev_out, out_stride, dev_info, fine16);   // encode.c:163-221
}

// The pipelines' transforms on their own (include/dwtx.h): what encode_part / dwtx_decode_device's `finish` run around the
// entropy stage, with the tiles' histograms riding along in the forward direction as they do there.
extern "C" int dwtx_transformation_fwd_pixels(dwtx_ctx *ctx, int32_t *dev_pyr, int16_t *dev_rings16, unsigned *levels16,
	const uint8_t *dev_pix, int W, int H, int C, int n)
{
	if (!ctx || !dev_pyr || !dev_pix || (C != 1 && C != 3) || n < 1 || (dev_rings16 && !levels16))
		return DWTX_ERR_ARG;
	DWTX_ENTER(ctx);
	DWTX_CHECK_DIMS(W, H);
	if (levels16)
		*levels16 = 0u;
	if (!dwtx_gray8_ok(W, H, dev_pix, (size_t)W * H * C)) {
		dwtx_set_error("the pixel transforms need W %% 4 == 0, more than 64 pixels on a side and 4-byte aligned pixels (%dx%d)", W, H);
		return DWTX_ERR_ARG;
	}
	int rc;
	dwtx_hist_sink sink;
	unsigned hist_levels = 0;
	if ((rc = dwtx_hist_begin(ctx, W, H, C, n, &sink)))
		return rc;
	dwtx_p16 fine16 = { nullptr, 0u };
	const unsigned sq = dwtx_square_levels(W, H);
	if (dev_rings16 && sq && (fine16.levels = dwtx_levels16(W, H, sq, 5)))
		fine16.planes = dev_rings16;
	if (levels16)
		*levels16 = fine16.planes ? fine16.levels : 0u;
	return dwtx_fwd_pixels8_hist(ctx, dev_pyr, dev_pix, W, H, C, n, &sink, &hist_levels, fine16);
}

extern "C" int dwtx_transformation_inv_pixels(dwtx_ctx *ctx, uint8_t *dev_pix, const int32_t *dev_pyr, const int16_t *dev_rings16,
	unsigned levels16, int W, int H, int C, int n)
{
	if (!ctx || !dev_pyr || !dev_pix || (C != 1 && C != 3) || n < 1 || (levels16 && !dev_rings16))
		return DWTX_ERR_ARG;
	DWTX_ENTER(ctx);
	DWTX_CHECK_DIMS(W, H);
	if (!dwtx_gray8_ok(W, H, dev_pix, (size_t)W * H * C)) {
		dwtx_set_error("the pixel transforms need W %% 4 == 0, more than 64 pixels on a side and 4-byte aligned pixels (%dx%d)", W, H);
		return DWTX_ERR_ARG;
	}
	if (levels16 && levels16 != dwtx_levels16(W, H, dwtx_square_levels(W, H), 5))
		return DWTX_ERR_ARG;   // (the mask the forward call reported for this geometry, or none)
	const dwtx_p16 f16 = { levels16 ? const_cast<int16_t *>(dev_rings16) : nullptr, levels16 };
	return dwtx_inv_pixels8(ctx, dev_pix, (size_t)W * H * C, dev_pyr, W, H, C, n, &f16);
}

// pixels (device) -> streams (device); async on the context's stream.
// The transform is bound by memory, the entropy stage by vector-instruction issue: a batch runs as parts on streams of
// their own, staggered so that part k's transform runs beside part k-1's entropy stage (the transforms follow one
// another: each fills the memory system by itself).  The caller's stream waits for all parts.
extern "C" int dwtx_encode_device(dwtx_ctx *ctx, const uint8_t *dev_pix, int W, int H, int C, int n, long capacity,
	uint8_t *dev_out, size_t out_stride, dwtx_stream_info *dev_info)
{
	if (!ctx || !dev_pix || !dev_out || !dev_info || (C != 1 && C != 3) || n < 1)
		return DWTX_ERR_ARG;
	DWTX_ENTER(ctx);
	DWTX_CHECK_DIMS(W, H);
	// (parts pay from about 32 images each: measured at the end of round 3, when the transform no longer waits on memory
	// the way it did — 1024 frames of 1080p RGB 43.2 -> 41.9 ms, 256 frames 11.0 -> 10.9, but 64 frames of 4096x4096 gray
	// 6.57 -> 6.72 and 16 frames 1.98 -> 2.17 the wrong way)
	const int K = ctx->opt[DWTX_OPT_ONE_STREAM] || n < 32 * DWTX_ENC_PARTS ? 1 : DWTX_ENC_PARTS;
	if (K == 1)
		return encode_part(ctx, dev_pix, W, H, C, n, capacity, dev_out, out_stride, dev_info, nullptr);
	dwtx_ctx *part[DWTX_ENC_PARTS];
	int rc;
	for (int k = 0; k < K; ++k)
		if ((rc = dwtx_encoder_part(ctx, k, &part[k])))
			return rc;
	hipEvent_t *lifted = ctx->enc_ev, *done = ctx->enc_ev + DWTX_ENC_PARTS, start = ctx->enc_ev[2 * DWTX_ENC_PARTS];
	DWTX_HIP(hipEventRecord(start, ctx->stream));   // the pixels are the caller's earlier work on its stream
	const size_t img_bytes = (size_t)W * H * C;
	rc = DWTX_OK;
	int queued = 0;
	for (int k = 0; k < K && !rc; ++k) {
		const int i0 = (int)((long)n * k / K), cnt = (int)((long)n * (k + 1) / K) - i0;
		hipStream_t st = part[k]->stream;
		// (a failure here must not leave the function: parts already queued read the caller's buffers, the join loop below waits for them)
		if (hipStreamWaitEvent(st, start, 0) != hipSuccess || (k && hipStreamWaitEvent(st, lifted[k - 1], 0) != hipSuccess)) {
			dwtx_set_error("%s:%d hipStreamWaitEvent failed for encoder part %d", __FILE__, __LINE__, k);
			rc = DWTX_ERR_DEVICE;
			break;
		}
		rc = encode_part(part[k], dev_pix + img_bytes * i0, W, H, C, cnt, capacity, dev_out + out_stride * (size_t)i0, out_stride,
			dev_info + i0, lifted[k]);
		if (hipEventRecord(done[k], st) != hipSuccess && !rc)
			rc = DWTX_ERR_DEVICE;
		queued = k + 1;
	}
	for (int k = 0; k < queued; ++k)   // (also after a failure: what was queued reads the caller's buffers)
		if (hipStreamWaitEvent(ctx->stream, done[k], 0) != hipSuccess && !rc)
			rc = DWTX_ERR_DEVICE;
	return rc;
}

// streams (device) -> pixels (device).  Image i is written densely (ow*oh*C bytes)
// at dev_pix + i*pix_stride; its size is widths/heights[info[i].level + 1].
extern "C" int dwtx_decode_device(dwtx_ctx *ctx, const uint8_t *dev_streams, size_t stream_stride,
	const unsigned long long *dev_lens, int W, int H, int C, int n, int levels_max,
	uint8_t *dev_pix, size_t pix_stride, dwtx_decode_info *host_info)
{
	if (!ctx || !dev_streams || !dev_lens || !dev_pix || !host_info || n < 1)
		return DWTX_ERR_ARG;
	DWTX_ENTER(ctx);
	DWTX_CHECK_DIMS(W, H);
	dwtx_geom g;
	int rc = dwtx_geometry(&g, W, H);
	if (rc)
		return rc;
	const size_t bytes = sizeof(int) * (size_t)W * H * C * n;
	int *a = (int *)dwtx_scratch(ctx, SLOT_CD_A, bytes);
	int *b = (int *)dwtx_scratch(ctx, SLOT_CD_B, bytes);
	if (!a || !b)
		return DWTX_ERR_NOMEM;
	const size_t plane_ints = (size_t)W * H;
	// 16-bit planes for the finest ring of whole pictures (see encode_part; the decoder checks the streams' plane counts)
	// (every ring the 16-byte-per-lane inverse kernels take, up to five levels: what a stream holds is bounded by its
	// plane counts on every level; the LL bands between the levels are sums of those and stay int32)
	dwtx_p16 fine16 = { nullptr, 0u };
	{
		const unsigned sq = ctx->opt[DWTX_OPT_NO_SQUARE_TILES] ? 0u : dwtx_square_levels(W, H);
		if (sq && !ctx->opt[DWTX_OPT_NO_FINE16] && dwtx_gray8_ok(W, H, dev_pix, pix_stride) && (fine16.levels = dwtx_levels16(W, H, sq, 5))) {
			fine16.planes = (int16_t *)dwtx_scratch(ctx, SLOT_CD_F16, sizeof(int16_t) * (size_t)W * H * C * n);
			if (!fine16.planes)
				return DWTX_ERR_NOMEM;
		}
	}
	// scratch both parts of the batch will ask for, sized once for the larger request
	{
		const size_t part_planes = (size_t)(n < 4 ? n : n - n / 2) * C;
		const size_t w1 = (W + 1) / 2, h1 = (H + 1) / 2, w2 = (w1 + 1) / 2, h2 = (h1 + 1) / 2;
		if (!dwtx_scratch(ctx, SLOT_CD_INFO, sizeof(int) * 48 * (size_t)n) ||
			!dwtx_scratch(ctx, SLOT_LIFT_A, sizeof(int) * w1 * h1 * part_planes) ||
			!dwtx_scratch(ctx, SLOT_LIFT_B, sizeof(int) * w2 * h2 * part_planes))
			return DWTX_ERR_NOMEM;
	}
	// reconstruction -> inverse transform -> pixels for images [first, first+count), queued on ctx->stream
	auto finish = [&](int first, int count, unsigned fused) -> int {
		const dwtx_decode_info &I = host_info[first];
		const int lo = I.level + 1;                                          // decode.c:251
		const int ow = g.widths[lo], oh = g.heights[lo];
		if ((size_t)ow * oh * C > pix_stride)
			return DWTX_ERR_ARG;
		int *miss = nullptr;
		bool biased = false;
		for (int k = 0; k < 48; ++k)
			biased = biased || I.missing[k] >= 2;
		if (biased) {
			miss = (int *)dwtx_scratch(ctx, SLOT_CD_INFO, sizeof(int) * 48 * (size_t)n) + 48 * (size_t)first;
			for (int i = 0; i < count; ++i) {
				int r = (int)hipMemcpyAsync(miss + 48 * i, host_info[first + i].missing, sizeof(int) * 48,
					hipMemcpyHostToDevice, ctx->stream);
				if (r)
					return DWTX_ERR_DEVICE;
			}
		}
		int *lin = a + plane_ints * C * first;
		int *pyr = b + plane_ints * C * first;
		int *img = a + plane_ints * C * first;   // lin is dead once reconstructed
		dwtx_p16 f16 = { nullptr, 0u };
		if (fused & DWTX_FUSED_FINE16) {   // the decoder put the part's finest rings there
			f16.planes = fine16.planes + plane_ints * C * first;
			f16.levels = fine16.levels;
		}
		fused &= ~DWTX_FUSED_FINE16;
		int r;
		if ((r = dwtx_reconstruction_ex(ctx, pyr, lin, miss, lo, W, H, C, count, fused, f16)))    // decode.c:257 (the rest of it)
			return r;
		if (dwtx_gray8_ok(ow, oh, dev_pix + pix_stride * first, pix_stride))
			return dwtx_inv_pixels8(ctx, dev_pix + pix_stride * first, pix_stride, pyr, ow, oh, C, count, &f16);   // decode.c:258-264
		if ((r = dwtx_transformation_inv(ctx, img, pyr, ow, oh, count * C)))                 // decode.c:258
			return r;
		if (count == 1 || (size_t)ow * oh * C == pix_stride)
			return dwtx_pixels_from_planes(ctx, dev_pix + pix_stride * first, img, ow, oh, C, count);   // decode.c:262-264
		for (int i = 0; i < count; ++i)
			if ((r = dwtx_pixels_from_planes(ctx, dev_pix + pix_stride * (first + i),
					img + (size_t)ow * oh * C * i, ow, oh, C, 1)))
				return r;
		return DWTX_OK;
	};
	// called by the decoder for each part of the batch as soon as its coefficients are on their way
	auto part = [&](int first, int count, unsigned fused) -> int {
		bool uniform = true;
		for (int i = first; i < first + count; ++i)
			uniform = uniform && !host_info[i].status && host_info[i].level == host_info[first].level &&
				memcmp(host_info[i].missing, host_info[first].missing, sizeof(host_info[first].missing)) == 0;
		if (uniform)
			return finish(first, count, fused);
		for (int i = first; i < first + count; ++i) {   // (a fused part's images all come out whole; their square levels are in the pyramid already)
			int r;
			if (!host_info[i].status && (r = finish(i, 1, fused)))
				return r;
		}
		return DWTX_OK;
	};
	using Part = decltype(part);
	return dwtx_decode_planes_ex(ctx, a, b, dev_streams, stream_stride, dev_lens, W, H, C, n, levels_max, host_info,
		[](void *user, int first, int count, unsigned fused) { return (*(Part *)user)(first, count, fused); }, &part, fine16);
}

// ---- dwtx_pack_streams: a step's streams as one message (include/dwtx.h) -------------------------------------------
namespace {
constexpr int PK_THREADS = 256;
constexpr int PK_PIECE = 1 << 16;   // bytes of one stream a workgroup moves

__device__ __forceinline__ unsigned long long pk_round8(unsigned long long len, unsigned long long stride)
{
	return ((len < stride ? len : stride) + 7ull) & ~7ull;
}

// grid (pieces of the longest possible stream, n): every workgroup adds up the rounded lengths before its stream
// (n is a batch size: a few hundred 8-byte loads from L2) and moves its piece with 8-byte accesses — rows and offsets are
// multiples of 8 bytes
__global__ __launch_bounds__(PK_THREADS) void k_pack_streams(uint8_t *out, unsigned long long out_bytes, unsigned long long *offsets,
	const uint8_t *streams, unsigned long long stride, const unsigned long long *lens, int n)
{
	__shared__ unsigned long long part[PK_THREADS / 64];
	const int i = blockIdx.y;
	unsigned long long mine = 0;
	for (int j = threadIdx.x; j < i; j += PK_THREADS)
		mine += pk_round8(lens[j], stride);
	for (int o = 32; o; o >>= 1)
		mine += __shfl_down(mine, o);
	if ((threadIdx.x & 63) == 0)
		part[threadIdx.x >> 6] = mine;
	__syncthreads();
	unsigned long long off = 0;
	for (int k = 0; k < PK_THREADS / 64; ++k)
		off += part[k];
	const unsigned long long len8 = pk_round8(lens[i], stride);
	if (offsets && blockIdx.x == 0 && threadIdx.x == 0) {
		offsets[i] = off;
		if (i == n - 1)
			offsets[n] = off + len8;
	}
	const unsigned long long first = (unsigned long long)blockIdx.x * PK_PIECE;
	if (first >= len8)
		return;
	const unsigned long long last = first + PK_PIECE < len8 ? first + PK_PIECE : len8;
	const unsigned long long *src = reinterpret_cast<const unsigned long long *>(streams + (unsigned long long)i * stride);
	unsigned long long *dst = reinterpret_cast<unsigned long long *>(out + off);
	for (unsigned long long b = first + 8ull * threadIdx.x; b < last; b += 8ull * PK_THREADS)
		if (off + b + 8 <= out_bytes)
			dst[b >> 3] = src[b >> 3];
}
} // namespace

extern "C" int dwtx_pack_streams(dwtx_ctx *ctx, uint8_t *dev_out, size_t out_bytes, unsigned long long *dev_offsets,
	const uint8_t *dev_streams, size_t stream_stride, const unsigned long long *dev_lens, int n)
{
	if (!ctx || !dev_out || !dev_streams || !dev_lens || n < 1 || n > 65535 || (stream_stride & 7) || !stream_stride ||
		((uintptr_t)dev_out & 7) || ((uintptr_t)dev_streams & 7))
		return DWTX_ERR_ARG;
	DWTX_ENTER(ctx);
	const unsigned pieces = (unsigned)((stream_stride + PK_PIECE - 1) / PK_PIECE);
	hipLaunchKernelGGL(k_pack_streams, dim3(pieces, n), dim3(PK_THREADS), 0, ctx->stream, dev_out, (unsigned long long)out_bytes, dev_offsets,
		dev_streams, (unsigned long long)stream_stride, dev_lens, n);
	DWTX_LAUNCH_CHECK();
	return DWTX_OK;
}

// ---- host-buffer wrappers (what the CLIs call) ---------------------------------
// A batch is cut into parts of at most part_size() images.  Part k's kernels run on the context's
// stream while part k+1's input is on its way in and part k-1's output on its way out (a second
// stream, two device staging buffers each way): device memory stays bounded for any n, and with
// page-locked host buffers (dwtx_host_alloc) the PCIe transfers hide behind the kernels.

static int part_size(const dwtx_ctx *ctx, int W, int H, int C, int n)
{
	const size_t samples = (size_t)W * H * C;
	size_t p = ((size_t)64 << 20) / (samples ? samples : 1);   // about 64 M samples per part (16 frames of 4096x4096 gray)
	p = p < 4 ? 4 : p > 256 ? 256 : p;
	if (ctx->opt[DWTX_OPT_PART_IMAGES] > 0)   // test hook: force small parts
		p = (size_t)ctx->opt[DWTX_OPT_PART_IMAGES];
	return (size_t)n < p ? n : (int)p;
}

static int sync_all(dwtx_ctx *ctx)
{
	hipError_t a = hipStreamSynchronize(ctx->stream);
	hipError_t b = ctx->have_copy ? hipStreamSynchronize(ctx->copy) : hipSuccess;
	return a == hipSuccess && b == hipSuccess ? DWTX_OK : DWTX_ERR_DEVICE;
}

extern "C" int dwtx_encode_images(dwtx_ctx *ctx, const uint8_t *pix, int W, int H, int C, int n, long capacity,
	uint8_t *out, size_t out_stride, size_t *out_lens, dwtx_stats *stats)
{
	if (!ctx || !pix || !out || !out_lens || (out_stride & 7) || n < 1 || (C != 1 && C != 3))
		return DWTX_ERR_ARG;
	DWTX_ENTER(ctx);
	DWTX_CHECK_DIMS(W, H);
	const int P = part_size(ctx, W, H, C, n), parts = (n + P - 1) / P;
	const size_t img_bytes = (size_t)W * H * C;
	int rc = dwtx_need_copy_stream(ctx);
	if (rc)
		return rc;
	uint8_t *dpix = (uint8_t *)dwtx_scratch(ctx, SLOT_CD_IO, 2 * img_bytes * P);
	uint8_t *dout = (uint8_t *)dwtx_scratch(ctx, SLOT_CD_IO2, 2 * out_stride * (size_t)P);
	dwtx_stream_info *dinfo = (dwtx_stream_info *)dwtx_scratch(ctx, SLOT_CD_LENS, 2 * sizeof(dwtx_stream_info) * (size_t)P);
	dwtx_stream_info *hinfo = nullptr;   // page-locked: its copy must not block the host
	if (hipHostMalloc((void **)&hinfo, 2 * sizeof(dwtx_stream_info) * (size_t)P, hipHostMallocDefault) != hipSuccess)
		hinfo = nullptr;
	if (!dpix || !dout || !dinfo || !hinfo) {
		if (hinfo)
			(void)hipHostFree(hinfo);
		return DWTX_ERR_NOMEM;
	}
	hipStream_t ms = ctx->stream, cs = ctx->copy;
	hipEvent_t *ev_in = ctx->cev, *ev_enc = ctx->cev + 2, *ev_out = ctx->cev + 4;   // per staging slot
	hipError_t e = hipSuccess;
	rc = DWTX_OK;
	auto first_of = [&](int k) { return k * P; };
	auto count_of = [&](int k) { return k == parts - 1 ? n - k * P : P; };
	// part k's streams to the host, once its lengths are known
	auto drain = [&](int k) {
		const int slot = k & 1, i0 = first_of(k), cnt = count_of(k);
		const dwtx_stream_info *hi = hinfo + (size_t)slot * P;
		e = hipEventSynchronize(ev_enc[slot]);
		for (int i = 0; e == hipSuccess && rc == DWTX_OK && i < cnt; ++i) {
			if (hi[i].error) {
				dwtx_set_error("image %d needs more than 16 bit planes", i0 + i);
				rc = DWTX_ERR_ARG;
				break;
			}
			if (hi[i].nbytes > out_stride) {
				dwtx_set_error("image %d: stream of %llu bytes exceeds out_stride %zu", i0 + i, hi[i].nbytes, out_stride);
				rc = DWTX_ERR_CAPACITY;
				break;
			}
			out_lens[i0 + i] = (size_t)hi[i].nbytes;
			e = hipMemcpyAsync(out + out_stride * (i0 + i), dout + out_stride * ((size_t)slot * P + i), (size_t)hi[i].nbytes,
				hipMemcpyDeviceToHost, cs);
			if (stats) {
				dwtx_stats &st = stats[i0 + i];
				st.meta_bits = (int)hi[i].meta_bits;                 // encode.c:175
				st.root_bits = (int)hi[i].root_bits;                 // encode.c:179
				st.total_bits = (int)hi[i].total_bits;               // encode.c:226 (int there too)
				st.kib = (int)((hi[i].nbytes + 512) / 1024);         // encode.c:228
				st.levels = 0;
				for (int c = 0; c < 3; ++c)
					st.planes[c] = hi[i].planes[c];
			}
		}
		if (e == hipSuccess)
			e = hipEventRecord(ev_out[slot], cs);
	};
	for (int k = 0; k < parts && e == hipSuccess && rc == DWTX_OK; ++k) {
		const int slot = k & 1, i0 = first_of(k), cnt = count_of(k);
		if (k >= 2) {
			e = hipStreamWaitEvent(cs, ev_enc[slot], 0);     // part k-2 has read this pixel buffer
			if (e == hipSuccess)
				e = hipStreamWaitEvent(ms, ev_out[slot], 0); // and its streams have left this output buffer
		}
		if (e == hipSuccess)
			e = hipMemcpyAsync(dpix + (size_t)slot * P * img_bytes, pix + (size_t)i0 * img_bytes, img_bytes * cnt,
				hipMemcpyHostToDevice, cs);
		if (e == hipSuccess)
			e = hipEventRecord(ev_in[slot], cs);
		if (e == hipSuccess)
			e = hipStreamWaitEvent(ms, ev_in[slot], 0);
		if (e != hipSuccess)
			break;
		rc = dwtx_encode_device(ctx, dpix + (size_t)slot * P * img_bytes, W, H, C, cnt, capacity,
			dout + out_stride * (size_t)slot * P, out_stride, dinfo + (size_t)slot * P);
		if (rc)
			break;
		e = hipMemcpyAsync(hinfo + (size_t)slot * P, dinfo + (size_t)slot * P, sizeof(dwtx_stream_info) * (size_t)cnt,
			hipMemcpyDeviceToHost, ms);
		if (e == hipSuccess)
			e = hipEventRecord(ev_enc[slot], ms);
		if (k >= 1 && e == hipSuccess)
			drain(k - 1);   // overlaps part k's kernels
	}
	if (e == hipSuccess && rc == DWTX_OK)
		drain(parts - 1);
	const int s = sync_all(ctx);
	(void)hipHostFree(hinfo);
	if (e != hipSuccess || s) {
		dwtx_set_error("encode_images transfer -> %s", hipGetErrorString(e));
		return DWTX_ERR_DEVICE;
	}
	return rc;
}

extern "C" int dwtx_decode_images(dwtx_ctx *ctx, const uint8_t *streams, size_t stream_stride, const size_t *lens, int n,
	int pixels_max, uint8_t *pix, size_t pix_stride, int *outW, int *outH, int *outC)
{
	return dwtx_decode_images_info(ctx, streams, stream_stride, lens, n, pixels_max, pix, pix_stride, outW, outH, outC, nullptr);
}

extern "C" int dwtx_decode_images_info(dwtx_ctx *ctx, const uint8_t *streams, size_t stream_stride, const size_t *lens, int n,
	int pixels_max, uint8_t *pix, size_t pix_stride, int *outW, int *outH, int *outC, dwtx_decode_info *infos)
{
	if (!ctx || !streams || !lens || !pix || !outW || !outH || !outC || n < 1 || (stream_stride & 7))
		return DWTX_ERR_ARG;
	DWTX_ENTER(ctx);
	// decode.c:142-159: geometry comes from the first stream's header; all streams of a batch share it
	if (lens[0] < 6 || streams[0] != 'W' || (streams[1] != '5' && streams[1] != '6'))
		return DWTX_ERR_ARG;
	for (int i = 0; i < n; ++i)
		if (lens[i] > stream_stride) {
			dwtx_set_error("stream %d: %zu bytes do not fit the stream stride %zu", i, lens[i], stream_stride);
			return DWTX_ERR_ARG;
		}
	const int C = streams[1] == '6' ? 3 : 1;
	const int W = (streams[2] | (streams[3] << 8)) + 1, H = (streams[4] | (streams[5] << 8)) + 1;
	DWTX_CHECK_DIMS(W, H);
	dwtx_geom g;
	dwtx_geometry(&g, W, H);
	int levels_max = -1;
	if (pixels_max >= 0) {   // decode.c:165-171
		levels_max = g.levels;
		while (levels_max > 0 && g.pixels[levels_max] > pixels_max)
			--levels_max;
	}
	const int P = part_size(ctx, W, H, C, n), parts = (n + P - 1) / P;
	const size_t img_bytes = (size_t)W * H * C;
	int rc = dwtx_need_copy_stream(ctx);
	if (rc)
		return rc;
	uint8_t *dstr = (uint8_t *)dwtx_scratch(ctx, SLOT_CD_IO, 2 * stream_stride * (size_t)P + 64);
	uint8_t *dpix = (uint8_t *)dwtx_scratch(ctx, SLOT_CD_IO2, 2 * img_bytes * P);
	unsigned long long *dlens = (unsigned long long *)dwtx_scratch(ctx, SLOT_CD_LENS, 2 * sizeof(unsigned long long) * (size_t)P);
	unsigned long long *hl = (unsigned long long *)malloc(sizeof(unsigned long long) * (size_t)n);
	dwtx_decode_info *info = (dwtx_decode_info *)malloc(sizeof(dwtx_decode_info) * (size_t)n);
	if (!dstr || !dpix || !dlens || !hl || !info) {
		free(hl);
		free(info);
		return DWTX_ERR_NOMEM;
	}
	for (int i = 0; i < n; ++i)
		hl[i] = lens[i];
	hipStream_t ms = ctx->stream, cs = ctx->copy;
	hipEvent_t *ev_in = ctx->cev, *ev_dec = ctx->cev + 2, *ev_out = ctx->cev + 4;   // per staging slot
	hipError_t e = hipSuccess;
	rc = DWTX_OK;
	auto count_of = [&](int k) { return k == parts - 1 ? n - k * P : P; };
	// part k's streams and lengths to the device (copy stream)
	auto feed = [&](int k) {
		const int slot = k & 1, i0 = k * P, cnt = count_of(k);
		if (k >= 2)
			e = hipStreamWaitEvent(cs, ev_dec[slot], 0);   // part k-2 has read this staging buffer
		if (e == hipSuccess)
			e = hipMemcpyAsync(dstr + stream_stride * (size_t)slot * P, streams + stream_stride * (size_t)i0, stream_stride * (size_t)cnt,
				hipMemcpyHostToDevice, cs);
		if (e == hipSuccess)
			e = hipMemcpyAsync(dlens + (size_t)slot * P, hl + i0, sizeof(unsigned long long) * (size_t)cnt, hipMemcpyHostToDevice, cs);
		if (e == hipSuccess)
			e = hipEventRecord(ev_in[slot], cs);
	};
	feed(0);
	for (int k = 0; k < parts && e == hipSuccess; ++k) {
		const int slot = k & 1, i0 = k * P, cnt = count_of(k);
		if (k + 1 < parts)
			feed(k + 1);   // travels while part k is decoded
		if (e == hipSuccess)
			e = hipStreamWaitEvent(ms, ev_in[slot], 0);
		if (e == hipSuccess && k >= 2)
			e = hipStreamWaitEvent(ms, ev_out[slot], 0);   // part k-2's pixels have left this buffer
		if (e != hipSuccess)
			break;
		ctx->index_base = (size_t)i0;   // sidecar index entries follow the images (dwtx_ctx_set_index)
		const int r = dwtx_decode_device(ctx, dstr + stream_stride * (size_t)slot * P, stream_stride, dlens + (size_t)slot * P, W, H, C, cnt,
			levels_max, dpix + img_bytes * (size_t)slot * P, img_bytes, info + i0);
		ctx->index_base = 0;
		if (r) {
			rc = r;
			break;
		}
		e = hipEventRecord(ev_dec[slot], ms);
		if (e == hipSuccess)
			e = hipStreamWaitEvent(cs, ev_dec[slot], 0);
		for (int i = 0; e == hipSuccess && i < cnt; ++i) {
			const dwtx_decode_info &I = info[i0 + i];
			outC[i0 + i] = C;
			if (I.status) {
				outW[i0 + i] = outH[i0 + i] = 0;
				if (n == 1 && I.status == 2) {   // refused, not unreadable: the caller is told why
					dwtx_set_error("the stream claims more than 16 bit planes: damaged, refused (the reference would decode garbage)");
					rc = DWTX_ERR_ARG;
				} else {
					rc = n == 1 ? DWTX_ERR_IO : rc;   // decode.c:181,185: unreadable root/planes -> exit 1
				}
				continue;
			}
			const int lo = I.level + 1;
			outW[i0 + i] = g.widths[lo];
			outH[i0 + i] = g.heights[lo];
			e = hipMemcpyAsync(pix + pix_stride * (size_t)(i0 + i), dpix + img_bytes * ((size_t)slot * P + i),
				(size_t)outW[i0 + i] * outH[i0 + i] * C, hipMemcpyDeviceToHost, cs);
		}
		if (e == hipSuccess)
			e = hipEventRecord(ev_out[slot], cs);
	}
	const int s = sync_all(ctx);
	if (infos && e == hipSuccess && !s && (rc == DWTX_OK || rc == DWTX_ERR_IO || rc == DWTX_ERR_ARG))
		memcpy(infos, info, sizeof(dwtx_decode_info) * (size_t)n);
	free(hl);
	free(info);
	if (e != hipSuccess || s) {
		dwtx_set_error("decode_images transfer -> %s", hipGetErrorString(e));
		return DWTX_ERR_DEVICE;
	}
	return rc;
}
