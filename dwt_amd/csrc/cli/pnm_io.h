/* pnm_io.h — PNM (P5/P6, maxval 255) and whole-file byte I/O for the CLIs.
 * Host-side edge of the pipeline: pnm.h:14-117 and bytes.h:24-118 of the
 * reference do this one fgetc/fputc at a time; here it is bulk fread/fwrite
 * with the same accepted inputs, messages and "-" = stdin/stdout rule. */
#ifndef DWTX_PNM_IO_H
#define DWTX_PNM_IO_H

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include <string.h>

static const char *std_name(const char *name, const char *fallback)
{
	/* bytes.h:26-28,42-44 / pnm.h:16-18,93-95: exactly "-" selects the standard stream */
	return (name[0] == '-' && !name[1]) ? fallback : name;
}

/* "-" means the process's own standard stream (bytes.h:26-28,42-44, pnm.h:16-18,93-95 open
 * /dev/stdin and /dev/stdout by path; a duplicate of the descriptor is the same file without the
 * path lookup, which some sandboxes refuse for pipes). */
static FILE *open_stream(const char *name, const char *fname, int writing)
{
	if (fname != name) {
		int fd = dup(writing ? 1 : 0);
		FILE *f = fd >= 0 ? fdopen(fd, writing ? "wb" : "rb") : 0;
		if (f)
			return f;
		if (fd >= 0)
			close(fd);
	}
	return fopen(fname, writing ? "wb" : "rb");
}

static uint8_t *read_all(FILE *f, size_t *len)
{
	size_t cap = 1 << 20, n = 0;
	uint8_t *b = (uint8_t *)malloc(cap);
	while (b) {
		size_t r = fread(b + n, 1, cap - n, f);
		n += r;
		if (!r)
			break;
		if (n == cap) {
			cap *= 2;
			b = (uint8_t *)realloc(b, cap);
		}
	}
	*len = n;
	return b;
}

/* pnm.h:14-90.  Returns malloc'ed pixel payload or NULL (message printed). */
static uint8_t *pnm_read(const char *name, int *W, int *H, int *C)
{
	const char *fname = std_name(name, "/dev/stdin");
	FILE *f = open_stream(name, fname, 0);
	if (!f) {
		fprintf(stderr, "could not open \"%s\" file to read.\n", fname);
		return 0;
	}
	size_t len;
	uint8_t *b = read_all(f, &len);
	fclose(f);
	if (!b || len < 2 || b[0] != 'P' || (b[1] != '5' && b[1] != '6')) {
		fprintf(stderr, "file \"%s\" neither P5 nor P6 image.\n", fname);
		free(b);
		return 0;
	}
	*C = b[1] == '5' ? 1 : 3;
	size_t p = 3;           /* pnm.h:33: one byte after the magic is consumed unconditionally */
	long v[3] = { 0, 0, 0 };
	for (int i = 0; i < 3; ++i) {
		/* pnm.h:37-43: comments, then anything that is not a digit */
		while (p < len && b[p] == '#') {
			while (p < len && b[p] != '\n')
				++p;
			++p;
		}
		while (p < len && (b[p] < '0' || b[p] > '9'))
			++p;
		int digits = 0;
		while (p < len && b[p] >= '0' && b[p] <= '9' && digits < 15) {
			v[i] = v[i] * 10 + (b[p] - '0');
			++p;
			++digits;
		}
		if (p >= len)
			goto eof;
		++p;                /* pnm.h:46-49: the byte that ended the number is consumed */
	}
	if (!(v[0] && v[1] && v[2])) {
		fprintf(stderr, "could not read image file \"%s\".\n", fname);
		free(b);
		return 0;
	}
	if (v[2] != 255) {
		fprintf(stderr, "cant read \"%s\", only 8 bit per channel SRGB supported at the moment.\n", fname);
		free(b);
		return 0;
	}
	{
		size_t need = (size_t)v[0] * (size_t)v[1] * (size_t)*C;
		if (p + need > len)
			goto eof;
		uint8_t *pix = (uint8_t *)malloc(need ? need : 1);
		memcpy(pix, b + p, need);
		free(b);
		*W = (int)v[0];
		*H = (int)v[1];
		return pix;
	}
eof:
	fprintf(stderr, "EOF while reading from \"%s\".\n", fname);
	free(b);
	return 0;
}

/* pnm.h:92-117 (values are already clamped to 0..255 by the device kernel) */
static int pnm_write(const char *name, const uint8_t *pix, int W, int H, int C)
{
	const char *fname = std_name(name, "/dev/stdout");
	FILE *f = open_stream(name, fname, 1);
	if (!f) {
		fprintf(stderr, "could not open \"%s\" file to write.\n", fname);
		return 0;
	}
	if (fprintf(f, "P%d %d %d 255\n", C == 1 ? 5 : 6, W, H) < 0 ||
		fwrite(pix, 1, (size_t)W * H * C, f) != (size_t)W * H * C) {
		fprintf(stderr, "EOF while writing to \"%s\".\n", fname);
		fclose(f);
		return 0;
	}
	fclose(f);
	return 1;
}

#endif
