/*
 * decode — drop-in for the reference's ./decode (decode.c:136-268):
 *   decode input.dwt output.pnm [PIXELS]
 * Same argv, "-" for stdin/stdout, exit codes and output bytes, including
 * truncated streams (resolution drop, dequantisation bias) and the PIXELS cap.
 *
 * Beyond the reference (never needed, never changes the output): if a file "input.dwt.idx" lies beside the
 * stream it is offered to the decoder as the stream's sidecar index (include/dwtx.h dwtx_index: all segments
 * are then walked at once; an index that does not fit is noticed and ignored), and with DWTX_WRITE_INDEX set
 * in the environment the decode leaves that file behind.
 */
#include "../../../include/dwtx.h"
#include "pnm_io.h"

int main(int argc, char **argv)
{
	if (argc < 3 || argc > 4) {
		fprintf(stderr, "usage: %s input.dwt output.pnm [PIXELS]\n", argv[0]);
		return 1;
	}
	const char *fname = std_name(argv[1], "/dev/stdin");
	FILE *f = open_stream(argv[1], fname, 0);
	if (!f) {
		fprintf(stderr, "could not open \"%s\" file to read\n", fname);   /* bytes.h:31 */
		return 1;
	}
	size_t len;
	uint8_t *raw = read_all(f, &len);
	fclose(f);
	if (!raw)
		return 1;
	/* decode.c:145-155 reads the header byte by byte: a wrong magic byte ends the program silently,
	 * only a MISSING byte prints get_byte()'s message (bytes.h:101) */
	if (len >= 1 && raw[0] != 'W')
		return 1;
	if (len >= 2 && raw[1] != '5' && raw[1] != '6')
		return 1;
	if (len < 6) {
		fprintf(stderr, "reached end of file \"%s\"\n", argv[1]);
		return 1;
	}
	int W = (raw[2] | (raw[3] << 8)) + 1, H = (raw[4] | (raw[5] << 8)) + 1, C = raw[1] == '6' ? 3 : 1;
	if (W < DWTX_MIN_LEN || H < DWTX_MIN_LEN)                           /* decode.c:158 */
		return 1;
	if (W > DWTX_MAX_SIDE || H > DWTX_MAX_SIDE) {
		/* the reference goes on and skips the finest level (int overflow at decode.c:47): nothing to match */
		fprintf(stderr, "image sides above %d are not supported (%dx%d)\n", DWTX_MAX_SIDE, W, H);
		return 1;
	}
	int pixels_max = -1;
	if (argc >= 4) {                                                    /* decode.c:165-166 */
		pixels_max = atoi(argv[3]);
		if (pixels_max < 0)
			pixels_max = 0;   /* any negative value drops every level, like 0 */
	}
	size_t stride = (len + 64 + 7) / 8 * 8;
	uint8_t *padded = (uint8_t *)calloc(stride, 1);
	memcpy(padded, raw, len);
	free(raw);
	uint8_t *pix = (uint8_t *)malloc((size_t)W * H * C);
	dwtx_ctx *ctx;
	if (dwtx_ctx_create(0, &ctx)) {
		fprintf(stderr, "%s\n", dwtx_last_error());
		return 1;
	}
	/* the sidecar index: 32 header bytes + 32 per segment, as they lie in dwtx_index */
	dwtx_index *ix_in = NULL, *ix_out = NULL;
	char *iname = NULL;
	if (strcmp(argv[1], "-")) {
		iname = (char *)malloc(strlen(argv[1]) + 5);
		sprintf(iname, "%s.idx", argv[1]);
		FILE *fi = fopen(iname, "rb");
		if (fi) {
			ix_in = (dwtx_index *)calloc(1, sizeof(dwtx_index));
			const size_t head = sizeof(dwtx_index) - sizeof(ix_in->seg);
			if (fread(ix_in, 1, head, fi) != head || ix_in->magic != DWTX_INDEX_MAGIC || ix_in->nsegs <= 0 ||
				ix_in->nsegs > DWTX_INDEX_MAX_SEGS || fread(ix_in->seg, sizeof(dwtx_seg_index), (size_t)ix_in->nsegs, fi) != (size_t)ix_in->nsegs) {
				free(ix_in);
				ix_in = NULL;
			}
			fclose(fi);
		}
		if (getenv("DWTX_WRITE_INDEX"))
			ix_out = (dwtx_index *)calloc(1, sizeof(dwtx_index));
	}
	dwtx_ctx_set_index(ctx, ix_in, ix_out);
	/* debugging aids of this tool (the library reads no environment): ignore the sidecar index / refuse to fall back from one */
	if (getenv("DWTX_NO_INDEX"))
		dwtx_ctx_set_option(ctx, DWTX_OPT_NO_INDEX, 1);
	if (getenv("DWTX_NO_INDEX_FALLBACK"))
		dwtx_ctx_set_option(ctx, DWTX_OPT_NO_INDEX_FALLBACK, 1);
	int ow, oh, oc;
	dwtx_decode_info info;
	int rc = dwtx_decode_images_info(ctx, padded, stride, &len, 1, pixels_max, pix, (size_t)W * H * C, &ow, &oh, &oc, &info);
	if (rc == DWTX_ERR_IO) {   /* decode.c:181,185: root image or plane counts cut off */
		fprintf(stderr, "reached end of file \"%s\"\n", argv[1]);
		return 1;
	}
	if (rc) {
		fprintf(stderr, "%s\n", dwtx_last_error());
		return 1;
	}
	if (info.truncated & 2)    /* bytes.h:99-103: get_byte() ran into the end of a cut-off stream; decoding goes on with what it has */
		fprintf(stderr, "reached end of file \"%s\"\n", argv[1]);
	if (info.zeros_left > 1)   /* rle.h:43-46: the PIXELS cap (or the end of data) left part of a zero run unread */
		fprintf(stderr, "%u zeros not read.\n", info.zeros_left);
	if (!pnm_write(argv[2], pix, ow, oh, oc))
		return 1;
	if (ix_out && ix_out->nsegs > 0) {
		FILE *fo = fopen(iname, "wb");
		if (fo) {
			fwrite(ix_out, 1, sizeof(dwtx_index) - sizeof(ix_out->seg), fo);
			fwrite(ix_out->seg, sizeof(dwtx_seg_index), (size_t)ix_out->nsegs, fo);
			fclose(fo);
		}
	}
	free(ix_in);
	free(ix_out);
	free(iname);
	dwtx_ctx_destroy(ctx);
	free(padded);
	free(pix);
	return 0;
}
