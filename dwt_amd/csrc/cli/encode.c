/*
 * encode — drop-in for the reference's ./encode (encode.c:133-232):
 *   encode input.pnm output.dwt [CAPACITY]
 * Same argv, "-" for stdin/stdout, exit codes (0 ok, 1 on bad args / unreadable
 * input / size out of range / unwritable output), stderr statistics lines and
 * .dwt bytes.  The transform and the coder run on the GPU through libdwtx.
 *
 * Beyond the reference: with DWTX_WRITE_INDEX set in the environment the stream just written is walked once by the
 * decoder and its sidecar index (include/dwtx.h dwtx_index) is left in "output.dwt.idx", for later decodes to find.
 */
#include "../../../include/dwtx.h"
#include "pnm_io.h"

int main(int argc, char **argv)
{
	if (argc != 3 && argc != 4) {
		fprintf(stderr, "usage: %s input.pnm output.dwt [CAPACITY]\n", argv[0]);
		return 1;
	}
	int W, H, C;
	uint8_t *pix = pnm_read(argv[1], &W, &H, &C);
	if (!pix || W > 65536 || H > 65536)          /* encode.c:140 */
		return 1;
	if (W < DWTX_MIN_LEN || H < DWTX_MIN_LEN)    /* encode.c:145 */
		return 1;
	if (W > DWTX_MAX_SIDE || H > DWTX_MAX_SIDE) {
		/* the reference goes on and writes a stream that does not decode to the picture (int overflow at encode.c:45) */
		fprintf(stderr, "image sides above %d are not supported (%dx%d)\n", DWTX_MAX_SIDE, W, H);
		return 1;
	}
	long capacity = argc >= 4 ? atoi(argv[3]) : 0;   /* encode.c:150-152 */
	dwtx_ctx *ctx;
	if (dwtx_ctx_create(0, &ctx)) {
		fprintf(stderr, "%s\n", dwtx_last_error());
		return 1;
	}
	/* a CAPACITY beyond what the image can need must not size the buffers */
	size_t stride = dwtx_encode_bound(W, H, C);
	if (capacity > 0 && ((size_t)capacity + 15) / 8 * 8 < stride)
		stride = ((size_t)capacity + 15) / 8 * 8;
	uint8_t *out = (uint8_t *)malloc(stride);
	size_t len = 0;
	dwtx_stats st;
	int rc = dwtx_encode_images(ctx, pix, W, H, C, 1, capacity, out, stride, &len, &st);
	if (rc) {
		fprintf(stderr, "%s\n", dwtx_last_error());
		return 1;
	}
	/* bytes.h:40-57: the sink is opened after the transform, before the first byte */
	const char *fname = std_name(argv[2], "/dev/stdout");
	FILE *f = open_stream(argv[2], fname, 1);
	if (!f) {
		fprintf(stderr, "could not open \"%s\" file to write\n", fname);
		return 1;
	}
	fprintf(stderr, "%d bits for meta data\n", st.meta_bits);     /* encode.c:176 */
	fprintf(stderr, "%d bits for root image\n", st.root_bits);    /* encode.c:180 */
	if (fwrite(out, 1, len, f) != len)
		fprintf(stderr, "could not write to file \"%s\"\n", argv[2]);   /* bytes.h:80 */
	fclose(f);
	fprintf(stderr, "%d bits (%d KiB) encoded\n", st.total_bits, st.kib);   /* encode.c:230 */
	if (getenv("DWTX_WRITE_INDEX") && strcmp(argv[2], "-")) {
		/* the index is what a decode of the stream finds out (a stream cut by CAPACITY has none) */
		dwtx_index *ix = (dwtx_index *)calloc(1, sizeof(dwtx_index));
		const size_t dstride = (len + 64 + 7) / 8 * 8;
		uint8_t *padded = (uint8_t *)calloc(dstride, 1);
		uint8_t *back = (uint8_t *)malloc((size_t)W * H * C);
		memcpy(padded, out, len);
		int ow, oh, oc;
		dwtx_decode_info info;
		dwtx_ctx_set_index(ctx, NULL, ix);
		if (!dwtx_decode_images_info(ctx, padded, dstride, &len, 1, -1, back, (size_t)W * H * C, &ow, &oh, &oc, &info) && ix->nsegs > 0) {
			char *iname = (char *)malloc(strlen(argv[2]) + 5);
			sprintf(iname, "%s.idx", argv[2]);
			FILE *fo = fopen(iname, "wb");
			if (fo) {
				fwrite(ix, 1, sizeof(dwtx_index) - sizeof(ix->seg), fo);
				fwrite(ix->seg, sizeof(dwtx_seg_index), (size_t)ix->nsegs, fo);
				fclose(fo);
			}
			free(iname);
		}
		dwtx_ctx_set_index(ctx, NULL, NULL);
		free(back);
		free(padded);
		free(ix);
	}
	dwtx_ctx_destroy(ctx);
	free(out);
	free(pix);
	return 0;
}
