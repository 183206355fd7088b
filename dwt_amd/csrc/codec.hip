// codec.hip — whole-image entry points: the device pipelines behind the
// reference's two main()s (encode.c:133-232, decode.c:136-268 minus file I/O),
// for batches of same-geometry images.
#include "dwtx_internal.h"

#include <stdlib.h>
#include <string.h>

enum { SLOT_CD_A = 16, SLOT_CD_B, SLOT_CD_INFO, SLOT_CD_IO, SLOT_CD_IO2, SLOT_CD_LENS };

extern "C" size_t dwtx_encode_bound(int W, int H, int C)
{
	// generous: noise costs ~9 bit/sample, allow 2 bytes per sample
	size_t b = (size_t)2 * W * H * C + 4096;
	return (b + 7) / 8 * 8;
}

// pixels (device) -> streams (device); async on the context's stream
extern "C" int dwtx_encode_device(dwtx_ctx *ctx, const uint8_t *dev_pix, int W, int H, int C, int n, long capacity,
	uint8_t *dev_out, size_t out_stride, dwtx_stream_info *dev_info)
{
	if (!ctx || !dev_pix || !dev_out || !dev_info || W < DWTX_MIN_LEN || H < DWTX_MIN_LEN || W > 65536 || H > 65536 ||
		(C != 1 && C != 3) || n < 1)
		return DWTX_ERR_ARG;
	const size_t bytes = sizeof(int) * (size_t)W * H * C * n;
	int *a = (int *)dwtx_scratch(ctx, SLOT_CD_A, bytes);
	int *b = (int *)dwtx_scratch(ctx, SLOT_CD_B, bytes);
	if (!a || !b)
		return DWTX_ERR_NOMEM;
	int rc;
	if (dwtx_gray8_ok(W, H, dev_pix, (size_t)W * H * C)) {
		if ((rc = dwtx_fwd_pixels8(ctx, b, dev_pix, W, H, C, n)))              // encode.c:155-159 in one pass
			return rc;
	} else {
		if ((rc = dwtx_planes_from_pixels(ctx, a, dev_pix, W, H, C, n)))       // encode.c:155-156
			return rc;
		if ((rc = dwtx_transformation_fwd(ctx, b, a, W, H, n * C)))            // encode.c:159
			return rc;
	}
	if ((rc = dwtx_linearization(ctx, a, b, W, H, n * C)))                 // encode.c:160
		return rc;
	return dwtx_encode_planes(ctx, a, W, H, C, n, capacity, dev_out, out_stride, dev_info);   // encode.c:163-221
}

// streams (device) -> pixels (device).  Image i is written densely (ow*oh*C bytes)
// at dev_pix + i*pix_stride; its size is widths/heights[info[i].level + 1].
extern "C" int dwtx_decode_device(dwtx_ctx *ctx, const uint8_t *dev_streams, size_t stream_stride,
	const unsigned long long *dev_lens, int W, int H, int C, int n, int levels_max,
	uint8_t *dev_pix, size_t pix_stride, dwtx_decode_info *host_info)
{
	if (!ctx || !dev_streams || !dev_lens || !dev_pix || !host_info || n < 1)
		return DWTX_ERR_ARG;
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
	auto finish = [&](int first, int count) -> int {
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
		int r;
		if ((r = dwtx_reconstruction(ctx, pyr, lin, miss, lo, W, H, C, count)))              // decode.c:257
			return r;
		if (dwtx_gray8_ok(ow, oh, dev_pix + pix_stride * first, pix_stride))
			return dwtx_inv_pixels8(ctx, dev_pix + pix_stride * first, pix_stride, pyr, ow, oh, C, count);   // decode.c:258-264
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
	auto part = [&](int first, int count) -> int {
		bool uniform = true;
		for (int i = first; i < first + count; ++i)
			uniform = uniform && !host_info[i].status && host_info[i].level == host_info[first].level &&
				memcmp(host_info[i].missing, host_info[first].missing, sizeof(host_info[first].missing)) == 0;
		if (uniform)
			return finish(first, count);
		for (int i = first; i < first + count; ++i) {
			int r;
			if (!host_info[i].status && (r = finish(i, 1)))
				return r;
		}
		return DWTX_OK;
	};
	using Part = decltype(part);
	return dwtx_decode_planes_ex(ctx, a, dev_streams, stream_stride, dev_lens, W, H, C, n, levels_max, host_info,
		[](void *user, int first, int count) { return (*(Part *)user)(first, count); }, &part);
}

// ---- host-buffer convenience wrappers (what the CLIs call) --------------------

extern "C" int dwtx_encode_images(dwtx_ctx *ctx, const uint8_t *pix, int W, int H, int C, int n, long capacity,
	uint8_t *out, size_t out_stride, size_t *out_lens, dwtx_stats *stats)
{
	if (!ctx || !pix || !out || !out_lens || (out_stride & 7))
		return DWTX_ERR_ARG;
	const size_t in_bytes = (size_t)W * H * C * n;
	uint8_t *dpix = (uint8_t *)dwtx_scratch(ctx, SLOT_CD_IO, in_bytes);
	uint8_t *dout = (uint8_t *)dwtx_scratch(ctx, SLOT_CD_IO2, out_stride * (size_t)n);
	dwtx_stream_info *dinfo = (dwtx_stream_info *)dwtx_scratch(ctx, SLOT_CD_LENS, sizeof(dwtx_stream_info) * (size_t)n);
	if (!dpix || !dout || !dinfo)
		return DWTX_ERR_NOMEM;
	DWTX_HIP(hipMemcpyAsync(dpix, pix, in_bytes, hipMemcpyHostToDevice, ctx->stream));
	int rc = dwtx_encode_device(ctx, dpix, W, H, C, n, capacity, dout, out_stride, dinfo);
	if (rc)
		return rc;
	dwtx_stream_info *hinfo = (dwtx_stream_info *)malloc(sizeof(dwtx_stream_info) * (size_t)n);
	if (!hinfo)
		return DWTX_ERR_NOMEM;
	hipError_t e = hipMemcpyAsync(hinfo, dinfo, sizeof(dwtx_stream_info) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream);
	if (e == hipSuccess)
		e = hipStreamSynchronize(ctx->stream);
	rc = DWTX_OK;
	for (int i = 0; e == hipSuccess && i < n; ++i) {
		if (hinfo[i].error) {
			dwtx_set_error("image %d needs more than 16 bit planes", i);
			rc = DWTX_ERR_ARG;
			break;
		}
		if (hinfo[i].nbytes > out_stride) {
			dwtx_set_error("image %d: stream of %llu bytes exceeds out_stride %zu", i, hinfo[i].nbytes, out_stride);
			rc = DWTX_ERR_CAPACITY;
			break;
		}
		out_lens[i] = (size_t)hinfo[i].nbytes;
		e = hipMemcpyAsync(out + out_stride * i, dout + out_stride * i, (size_t)hinfo[i].nbytes, hipMemcpyDeviceToHost,
			ctx->stream);
		if (stats) {
			stats[i].meta_bits = 48;                                   // encode.c:175
			stats[i].root_bits = (int)hinfo[i].root_bits;              // encode.c:179
			stats[i].total_bits = (int)hinfo[i].total_bits;            // encode.c:226 (int there too)
			stats[i].kib = (int)((hinfo[i].nbytes + 512) / 1024);      // encode.c:228
			stats[i].levels = 0;
			for (int c = 0; c < 3; ++c)
				stats[i].planes[c] = hinfo[i].planes[c];
		}
	}
	if (e == hipSuccess)
		e = hipStreamSynchronize(ctx->stream);
	free(hinfo);
	if (e != hipSuccess) {
		dwtx_set_error("encode_images copy -> %s", hipGetErrorString(e));
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
	// decode.c:142-159: geometry comes from the first stream's header; all streams of a batch share it
	if (lens[0] < 6 || streams[0] != 'W' || (streams[1] != '5' && streams[1] != '6'))
		return DWTX_ERR_ARG;
	const int C = streams[1] == '6' ? 3 : 1;
	const int W = (streams[2] | (streams[3] << 8)) + 1, H = (streams[4] | (streams[5] << 8)) + 1;
	if (W < DWTX_MIN_LEN || H < DWTX_MIN_LEN)
		return DWTX_ERR_ARG;
	dwtx_geom g;
	dwtx_geometry(&g, W, H);
	int levels_max = -1;
	if (pixels_max >= 0) {   // decode.c:165-171
		levels_max = g.levels;
		while (levels_max > 0 && g.pixels[levels_max] > pixels_max)
			--levels_max;
	}
	uint8_t *dstr = (uint8_t *)dwtx_scratch(ctx, SLOT_CD_IO, stream_stride * (size_t)n + 64);
	uint8_t *dpix = (uint8_t *)dwtx_scratch(ctx, SLOT_CD_IO2, (size_t)W * H * C * n);
	unsigned long long *dlens = (unsigned long long *)dwtx_scratch(ctx, SLOT_CD_LENS, sizeof(unsigned long long) * (size_t)n);
	unsigned long long *hl = (unsigned long long *)malloc(sizeof(unsigned long long) * (size_t)n);
	dwtx_decode_info *info = (dwtx_decode_info *)malloc(sizeof(dwtx_decode_info) * (size_t)n);
	if (!dstr || !dpix || !dlens || !hl || !info) {
		free(hl);
		free(info);
		return DWTX_ERR_NOMEM;
	}
	for (int i = 0; i < n; ++i)
		hl[i] = lens[i];
	int rc = DWTX_OK;
	hipError_t e = hipMemcpyAsync(dstr, streams, stream_stride * (size_t)n, hipMemcpyHostToDevice, ctx->stream);
	if (e == hipSuccess)
		e = hipMemcpyAsync(dlens, hl, sizeof(unsigned long long) * (size_t)n, hipMemcpyHostToDevice, ctx->stream);
	if (e == hipSuccess)
		rc = dwtx_decode_device(ctx, dstr, stream_stride, dlens, W, H, C, n, levels_max, dpix, (size_t)W * H * C, info);
	for (int i = 0; e == hipSuccess && rc == DWTX_OK && i < n; ++i) {
		if (info[i].status) {
			outW[i] = outH[i] = 0;
			outC[i] = C;
			rc = n == 1 ? DWTX_ERR_IO : rc;   // decode.c:181,185: unreadable root/planes -> exit 1
			continue;
		}
		const int lo = info[i].level + 1;
		outW[i] = g.widths[lo];
		outH[i] = g.heights[lo];
		outC[i] = C;
		e = hipMemcpyAsync(pix + pix_stride * i, dpix + (size_t)W * H * C * i, (size_t)outW[i] * outH[i] * C,
			hipMemcpyDeviceToHost, ctx->stream);
	}
	if (e == hipSuccess)
		e = hipStreamSynchronize(ctx->stream);
	if (infos && e == hipSuccess && (rc == DWTX_OK || rc == DWTX_ERR_IO))
		memcpy(infos, info, sizeof(dwtx_decode_info) * (size_t)n);
	free(hl);
	free(info);
	if (e != hipSuccess) {
		dwtx_set_error("decode_images copy -> %s", hipGetErrorString(e));
		return DWTX_ERR_DEVICE;
	}
	return rc;
}
