/*
 * orc_cli.c — command-line front end of the CPU ORACLE (test infrastructure only).
 *
 *   orc_cli encode in.pnm out.dwt [CAPACITY]     (same argv/stderr as encode.c:133-232)
 *   orc_cli decode in.dwt out.pnm [PIXELS]       (same argv as decode.c:136-268)
 *   orc_cli synth  W H C SEED KIND out.pnm       (SURVEY §8d generator)
 *   orc_cli time   W H C SEED KIND REPS          (single-thread round-trip timing, prints JSON)
 */
#include "dwt_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static uint8_t *slurp(const char *name, size_t *len)
{
	FILE *f = strcmp(name, "-") ? fopen(name, "rb") : stdin;
	if (!f) {
		fprintf(stderr, "could not open \"%s\" file to read\n", name);
		return 0;
	}
	size_t cap = 1 << 16, n = 0;
	uint8_t *b = malloc(cap);
	for (;;) {
		size_t r = fread(b + n, 1, cap - n, f);
		n += r;
		if (r == 0)
			break;
		if (n == cap)
			b = realloc(b, cap *= 2);
	}
	if (f != stdin)
		fclose(f);
	*len = n;
	return b;
}

static int spill(const char *name, const void *hdr, size_t hlen, const void *data, size_t len)
{
	FILE *f = strcmp(name, "-") ? fopen(name, "wb") : stdout;
	if (!f) {
		fprintf(stderr, "could not open \"%s\" file to write\n", name);
		return 1;
	}
	if (hlen)
		fwrite(hdr, 1, hlen, f);
	fwrite(data, 1, len, f);
	if (f != stdout)
		fclose(f);
	return 0;
}

/* pnm.h:14-90: P5/P6, '#' comments, maxval 255 */
static int parse_pnm(const uint8_t *b, size_t len, int *W, int *H, int *C, size_t *off)
{
	if (len < 3 || b[0] != 'P' || (b[1] != '5' && b[1] != '6'))
		return 1;
	*C = b[1] == '5' ? 1 : 3;
	size_t p = 2;
	int v[3];
	for (int i = 0; i < 3; ++i) {
		for (;;) {
			if (p >= len)
				return 1;
			if (b[p] == '#') {
				while (p < len && b[p] != '\n')
					++p;
			} else if (b[p] >= '0' && b[p] <= '9') {
				break;
			} else {
				++p;
			}
		}
		long a = 0;
		while (p < len && b[p] >= '0' && b[p] <= '9')
			a = a * 10 + (b[p++] - '0');
		v[i] = (int)a;
	}
	++p; /* the single whitespace byte after maxval */
	if (v[2] != 255 || v[0] <= 0 || v[1] <= 0)
		return 1;
	if (p + (size_t)v[0] * v[1] * *C > len)
		return 1;
	*W = v[0];
	*H = v[1];
	*off = p;
	return 0;
}

static double now(void)
{
	struct timespec t;
	clock_gettime(CLOCK_MONOTONIC, &t);
	return t.tv_sec + 1e-9 * t.tv_nsec;
}

int main(int argc, char **argv)
{
	if (argc >= 4 && !strcmp(argv[1], "encode")) {
		size_t len, off;
		uint8_t *f = slurp(argv[2], &len);
		int W, H, C;
		if (!f || parse_pnm(f, len, &W, &H, &C, &off))
			return 1;
		long cap = argc >= 5 ? atoi(argv[4]) : 0;
		uint8_t *out;
		size_t n;
		orc_stats st;
		if (orc_encode(f + off, W, H, C, cap, &out, &n, &st))
			return 1;
		fprintf(stderr, "%d bits for meta data\n", st.meta_bits);
		fprintf(stderr, "%d bits for root image\n", st.root_bits);
		int rc = spill(argv[3], 0, 0, out, n);
		fprintf(stderr, "%d bits (%d KiB) encoded\n", st.total_bits, st.kib);
		return rc;
	}
	if (argc >= 4 && !strcmp(argv[1], "decode")) {
		size_t len;
		uint8_t *f = slurp(argv[2], &len);
		if (!f)
			return 1;
		long px = -1;
		if (argc >= 5) {
			px = atoi(argv[4]);
			if (px < 0)
				px = 0;
		}
		uint8_t *pix;
		int W, H, C;
		if (orc_decode(f, len, px, &pix, &W, &H, &C))
			return 1;
		char hdr[64];
		int hl = snprintf(hdr, sizeof(hdr), "P%d %d %d 255\n", C == 1 ? 5 : 6, W, H);
		return spill(argv[3], hdr, (size_t)hl, pix, (size_t)W * H * C);
	}
	if (argc == 8 && !strcmp(argv[1], "synth")) {
		int W = atoi(argv[2]), H = atoi(argv[3]), C = atoi(argv[4]);
		uint8_t *pix = malloc((size_t)W * H * C);
		orc_synth(pix, W, H, C, (uint32_t)strtoul(argv[5], 0, 10), atoi(argv[6]));
		char hdr[64];
		int hl = snprintf(hdr, sizeof(hdr), "P%d %d %d 255\n", C == 1 ? 5 : 6, W, H);
		return spill(argv[7], hdr, (size_t)hl, pix, (size_t)W * H * C);
	}
	if (argc == 8 && !strcmp(argv[1], "time")) {
		int W = atoi(argv[2]), H = atoi(argv[3]), C = atoi(argv[4]);
		int reps = atoi(argv[7]);
		uint8_t *pix = malloc((size_t)W * H * C);
		double te = 0, td = 0;
		size_t bytes = 0;
		int ok = 1;
		for (int r = 0; r < reps; ++r) {
			orc_synth(pix, W, H, C, (uint32_t)strtoul(argv[5], 0, 10) + (uint32_t)r, atoi(argv[6]));
			uint8_t *out, *back;
			size_t n;
			int w, h, c;
			double t0 = now();
			orc_encode(pix, W, H, C, 0, &out, &n, 0);
			double t1 = now();
			orc_decode(out, n, -1, &back, &w, &h, &c);
			double t2 = now();
			te += t1 - t0;
			td += t2 - t1;
			bytes += n;
			ok &= w == W && h == H && !memcmp(pix, back, (size_t)W * H * C);
			free(out);
			free(back);
		}
		printf("{\"frames\": %d, \"encode_s\": %.6f, \"decode_s\": %.6f, \"bytes\": %zu, \"lossless\": %s, \"mpix_per_s\": %.4f}\n",
			reps, te, td, bytes, ok ? "true" : "false", 1e-6 * W * H * reps / (te + td));
		return !ok;
	}
	fprintf(stderr, "usage: %s encode|decode|synth|time ...\n", argv[0]);
	return 1;
}
