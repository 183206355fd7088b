/*
 * dwt_oracle.c — CPU ORACLE (test infrastructure only; see dwt_oracle.h).
 *
 * Plain-C restatement of the xdsopl/dwt codec.  Every function cites the
 * reference lines (under /root/reference) whose behaviour it restates.  The
 * structure is deliberately different from the reference (memory sinks instead
 * of FILE*, closed-form VLI, flag-free plane classification, line-buffer
 * lifting) — what must match is the output, byte for byte.
 *
 * Parity: PINNED against oracle/_ref (the real reference compiled from its own
 * sources by oracle/Makefile) and tests/golden/.
 */
#include "dwt_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ geometry */

/* utils.h:9-15 — floor(log2(x)) for x>0, -1 for x<=0. */
int orc_ilog2(int x)
{
	int l = -1;
	while (x > 0) {
		x >>= 1;
		++l;
	}
	return l;
}

static int pow2_cover(int v)
{
	/* utils.h:35-36: 1 << (ilog2(v-1)+1) = smallest power of two >= v (v>=1) */
	return 1 << (orc_ilog2(v - 1) + 1);
}

/* utils.h:17-40.  Sizes halve (rounding up) while both halves stay >= min_len;
 * entry 0 is the coarsest LL, entry `levels` the full image. */
int orc_geometry(orc_geom *g, int W, int H, int min_len)
{
	int ws[ORC_MAX_LEVELS + 1], hs[ORC_MAX_LEVELS + 1];
	int n = 0;
	ws[0] = W;
	hs[0] = H;
	/* the first halving always happens; a further one only while the current
	 * halves are still >= min_len (lengths_helper tests W2,H2 before recursing) */
	do {
		ws[n + 1] = (ws[n] + 1) / 2;
		hs[n + 1] = (hs[n] + 1) / 2;
		++n;
	} while (n < ORC_MAX_LEVELS - 1 && ws[n] >= min_len && hs[n] >= min_len);
	g->levels = n;
	for (int l = 0; l <= n; ++l) {
		int w = ws[n - l], h = hs[n - l];
		g->widths[l] = w;
		g->heights[l] = h;
		g->pixels[l] = w * h;
		int a = pow2_cover(w), b = pow2_cover(h);
		g->lengths[l] = a > b ? a : b;
	}
	return n;
}

/* ------------------------------------------------------------------- lifting */

/* cdf53.h:9-34 on one contiguous line.  C `/` truncates toward zero, which is
 * what the reference relies on. */
void orc_fwd53_line(int *x, int n, int *scratch)
{
	int even_end = n & ~1;
	for (int i = 1; i + 1 < n; i += 2)
		x[i] -= (x[i - 1] + x[i + 1]) / 2;
	if ((n & 1) == 0)
		x[n - 1] -= x[n - 2];
	x[0] += x[1] / 2;
	for (int i = 2; i < even_end; i += 2)
		x[i] += (x[i - 1] + x[i + 1]) / 4;
	int nlow = (n + 1) / 2;
	for (int i = 0; i < n; ++i)
		scratch[(i & 1) ? nlow + (i >> 1) : (i >> 1)] = x[i];
	memcpy(x, scratch, sizeof(int) * (size_t)n);
}

/* cdf53.h:36-61 */
void orc_inv53_line(int *x, int n, int *scratch)
{
	int even_end = n & ~1;
	int nlow = (n + 1) / 2;
	for (int i = 0; i < n; ++i)
		scratch[i] = x[(i & 1) ? nlow + (i >> 1) : (i >> 1)];
	scratch[0] -= scratch[1] / 2;
	for (int i = 2; i < even_end; i += 2)
		scratch[i] -= (scratch[i - 1] + scratch[i + 1]) / 4;
	for (int i = 1; i + 1 < n; i += 2)
		scratch[i] += (scratch[i - 1] + scratch[i + 1]) / 2;
	if ((n & 1) == 0)
		scratch[n - 1] += scratch[n - 2];
	memcpy(x, scratch, sizeof(int) * (size_t)n);
}

static void level_rows(int *img, int pitch, int w, int h, int C, int inverse, int *line, int *scratch)
{
	for (int j = 0; j < h; ++j) {
		for (int c = 0; c < C; ++c) {
			int *row = img + (size_t)j * pitch + c;
			for (int i = 0; i < w; ++i)
				line[i] = row[(size_t)i * C];
			if (inverse)
				orc_inv53_line(line, w, scratch);
			else
				orc_fwd53_line(line, w, scratch);
			for (int i = 0; i < w; ++i)
				row[(size_t)i * C] = line[i];
		}
	}
}

static void level_cols(int *img, int pitch, int w, int h, int C, int inverse, int *line, int *scratch)
{
	for (int i = 0; i < w * C; ++i) {
		int *col = img + i;
		for (int j = 0; j < h; ++j)
			line[j] = col[(size_t)j * pitch];
		if (inverse)
			orc_inv53_line(line, h, scratch);
		else
			orc_fwd53_line(line, h, scratch);
		for (int j = 0; j < h; ++j)
			col[(size_t)j * pitch] = line[j];
	}
}

/* encode.c:16-30: per level rows first, then columns, then recurse on the LL. */
void orc_forward(int *img, int W, int H, int C, int min_len)
{
	int n = W > H ? W : H;
	int *line = malloc(sizeof(int) * (size_t)n * 2);
	int *scratch = line + n;
	int pitch = W * C;
	int w = W, h = H;
	for (;;) {
		level_rows(img, pitch, w, h, C, 0, line, scratch);
		level_cols(img, pitch, w, h, C, 0, line, scratch);
		w = (w + 1) / 2;
		h = (h + 1) / 2;
		if (w < min_len || h < min_len)
			break;
	}
	free(line);
}

/* decode.c:16-30: coarsest level first; per level columns first, then rows. */
void orc_inverse(int *img, int W, int H, int C, int min_len)
{
	int ws[ORC_MAX_LEVELS + 2], hs[ORC_MAX_LEVELS + 2];
	int n = 0;
	ws[0] = W;
	hs[0] = H;
	while ((ws[n] + 1) / 2 >= min_len && (hs[n] + 1) / 2 >= min_len) {
		ws[n + 1] = (ws[n] + 1) / 2;
		hs[n + 1] = (hs[n] + 1) / 2;
		++n;
	}
	int m = W > H ? W : H;
	int *line = malloc(sizeof(int) * (size_t)m * 2);
	int *scratch = line + m;
	int pitch = W * C;
	for (int k = n; k >= 0; --k) {
		level_cols(img, pitch, ws[k], hs[k], C, 1, line, scratch);
		level_rows(img, pitch, ws[k], hs[k], C, 1, line, scratch);
	}
	free(line);
}

/* -------------------------------------------------------------------- colour */

/* image.h:52-65 (rgb2ycocg) over image.h:67-72 */
void orc_rgb_to_ycocg(int *img, long n)
{
	for (long i = 0; i < n; ++i) {
		int *p = img + 3 * i;
		int r = p[0], g = p[1], b = p[2];
		int co = r - b;
		int t = b + co / 2;
		int cg = g - t;
		p[0] = t + cg / 2;
		p[1] = co;
		p[2] = cg;
	}
}

static int clampi(int v, int lo, int hi)
{
	return v < lo ? lo : v > hi ? hi : v;
}

/* image.h:39-50 (ycocg2rgb, including its input clamps) over image.h:74-79 */
void orc_ycocg_to_rgb(int *img, long n)
{
	for (long i = 0; i < n; ++i) {
		int *p = img + 3 * i;
		int y = clampi(p[0], 0, 255);
		int co = clampi(p[1], -255, 255);
		int cg = clampi(p[2], -255, 255);
		int t = y - cg / 2;
		int g = cg + t;
		int b = t - co / 2;
		int r = b + co;
		p[0] = r;
		p[1] = g;
		p[2] = b;
	}
}

/* ------------------------------------------------------------------- hilbert */

/* hilbert.h:15-34: curve index -> (x,y) on an n*n grid, n a power of two.
 * Two bits of d are consumed per doubling of the cell size. */
void orc_hilbert(int n, int d, int *px, int *py)
{
	int x = 0, y = 0;
	for (int cell = 1; cell < n; cell <<= 1) {
		int hi = (d >> 1) & 1;
		int lo = (d ^ hi) & 1;
		if (!lo) {
			if (hi) {
				x = cell - 1 - x;
				y = cell - 1 - y;
			}
			int t = x;
			x = y;
			y = t;
		}
		x += hi ? cell : 0;
		y += lo ? cell : 0;
		d >>= 2;
	}
	*px = x;
	*py = y;
}

static int in_ring(const orc_geom *g, int l, int x, int y)
{
	return x < g->widths[l + 1] && y < g->heights[l + 1] && (x >= g->widths[l] || y >= g->heights[l]);
}

/* The curve visits every aligned s*s square (s a power of two) on s*s consecutive indices starting at a multiple
 * of s*s.  A square that holds no point of ring l — wholly outside the level's w*h rectangle, or wholly inside the
 * LL quadrant — contributes nothing to the loops of encode.c:45-56 / decode.c:47-63, so those indices can be stepped
 * over in one go; the points that ARE visited come in the reference's order.  (The reference walks all n*n indices:
 * minutes for a thin 32768-wide frame; this keeps such frames testable.)  Returns how many indices from d on are
 * known to be empty (0: look at d itself). */
static long empty_span(const orc_geom *g, int l, int n, long d)
{
	for (int s = n; s >= 8; s >>= 1) {
		long ss = (long)s * s;
		if (d & (ss - 1))
			continue;
		int x, y;
		orc_hilbert(n, (int)d, &x, &y);
		x &= ~(s - 1);
		y &= ~(s - 1);
		if (x >= g->widths[l + 1] || y >= g->heights[l + 1] || (x + s <= g->widths[l] && y + s <= g->heights[l]))
			return ss;
	}
	return 0;
}

/* encode.c:32-58 */
void orc_linearize(int *lin, const int *pyr, const orc_geom *g, int C)
{
	int L = g->levels;
	int W = g->widths[L];
	size_t total = (size_t)g->pixels[L];
	size_t k = 0;
	for (int y = 0; y < g->heights[0]; ++y)
		for (int x = 0; x < g->widths[0]; ++x, ++k)
			for (int c = 0; c < C; ++c)
				lin[c * total + k] = pyr[((size_t)y * W + x) * C + c];
	for (int l = 0; l < L; ++l) {
		int n = g->lengths[l + 1];
		long nn = (long)n * n;
		for (long d = 0; d < nn; ++d) {
			if (!(d & 63)) {
				long skip = empty_span(g, l, n, d);
				if (skip) {
					d += skip - 1;
					continue;
				}
			}
			int x, y;
			orc_hilbert(n, (int)d, &x, &y);
			if (!in_ring(g, l, x, y))
				continue;
			for (int c = 0; c < C; ++c)
				lin[c * total + k] = pyr[((size_t)y * W + x) * C + c];
			++k;
		}
	}
}

/* decode.c:32-65.  `levels` is the number of levels actually reconstructed;
 * the output pitch is widths[levels]. */
void orc_reconstruct(int *pyr, int *const *lin, const int *missing, const orc_geom *g, int levels, int C)
{
	int W = g->widths[levels];
	size_t k = 0;
	for (int y = 0; y < g->heights[0]; ++y)
		for (int x = 0; x < g->widths[0]; ++x, ++k)
			for (int c = 0; c < C; ++c)
				pyr[((size_t)y * W + x) * C + c] = lin[c][k];
	for (int l = 0; l < levels; ++l) {
		int n = g->lengths[l + 1];
		long nn = (long)n * n;
		for (long d = 0; d < nn; ++d) {
			if (!(d & 63)) {
				long skip = empty_span(g, l, n, d);
				if (skip) {
					d += skip - 1;
					continue;
				}
			}
			int x, y;
			orc_hilbert(n, (int)d, &x, &y);
			if (!in_ring(g, l, x, y))
				continue;
			for (int c = 0; c < C; ++c) {
				int v = lin[c][k];
				int m = missing[c * 16 + l] - 2;
				if (m >= 0 && v != 0)
					v += v < 0 ? -(1 << m) : (1 << m);
				pyr[((size_t)y * W + x) * C + c] = v;
			}
			++k;
		}
	}
}

/* ------------------------------------------------------------ bit sink (enc) */

/* bytes.h:75-85 + bits.h:58-78 as a growing memory buffer.  A byte is refused
 * (-2) once `cap` bytes exist (cap > 0); the field being written stops at that
 * bit, exactly like write_bits() returning early. */
typedef struct {
	uint8_t *buf;
	size_t len, alloc;
	long cap;
	uint64_t acc;
	int nacc;
	int err;
	int order;      /* vli.h:33  */
	long run;       /* rle.h:33  */
	long tokens, raw;
} sink;

static int sink_byte(sink *s, int b)
{
	if (s->cap > 0 && (long)s->len >= s->cap)
		return s->err = -2;
	if (s->len == s->alloc) {
		s->alloc = s->alloc ? s->alloc * 2 : 4096;
		s->buf = realloc(s->buf, s->alloc);
	}
	s->buf[s->len++] = (uint8_t)b;
	return 0;
}

static int sink_bits(sink *s, uint32_t v, int n)
{
	if (n <= 0)
		return 0;
	if (n < 32)
		v &= (1u << n) - 1u;
	s->acc |= (uint64_t)v << s->nacc;
	s->nacc += n;
	while (s->nacc >= 8) {
		if (sink_byte(s, (int)(s->acc & 255))) {
			s->acc = 0;
			s->nacc = 0;
			return s->err;
		}
		s->acc >>= 8;
		s->nacc -= 8;
	}
	return 0;
}

/* vli.h:67-84 in closed form: with order o and value v,
 * o* = ilog2(v + 2^o); (o*-o) zeros, a one, then v + 2^o - 2^o* in o* bits. */
static int sink_vli(sink *s, long v)
{
	int o = s->order;
	long biased = v + (1L << o);
	int top = -1;
	for (long t = biased; t > 0; t >>= 1)
		++top;
	int r;
	++s->tokens;
	if ((r = sink_bits(s, 0, top - o)))
		return r;
	if ((r = sink_bits(s, 1, 1)))
		return r;
	if ((r = sink_bits(s, (uint32_t)(biased - (1L << top)), top)))
		return r;
	s->order = top >= 2 ? top - 2 : 0;
	return 0;
}

/* rle.h:56-64 put_rle(1) and rle.h:35-38 rle_flush: emit the pending zero run */
static int sink_run(sink *s)
{
	int r = sink_vli(s, s->run);
	s->run = 0;
	return r;
}

/* rle.h:79-89 rle_put_bit: a raw bit first terminates a pending (>0) run */
static int sink_raw(sink *s, int bit)
{
	int r;
	if (s->run > 0 && (r = sink_run(s)))
		return r;
	++s->raw;
	return sink_bits(s, bit ? 1u : 0u, 1);
}

/* encode.c:97-110 — note: errors are ignored there, so we do not stop either */
static void put_root(sink *s, const int *val, int num)
{
	int max = 0;
	for (int i = 0; i < num; ++i) {
		int a = val[i] < 0 ? -val[i] : val[i];
		if (a > max)
			max = a;
	}
	int cnt = 1 + orc_ilog2(max);
	sink_vli(s, cnt);
	if (!cnt)
		return;
	for (int i = 0; i < num; ++i) {
		int a = val[i] < 0 ? -val[i] : val[i];
		sink_bits(s, (uint32_t)a, cnt);
		if (val[i])
			sink_bits(s, val[i] < 0, 1);
	}
}

/* encode.c:60-95 restated without the sig/ref flag bits: at plane p a
 * coefficient of magnitude m is in the refinement class iff (m >> (p+1)) != 0
 * (SURVEY §5.5).  v[] holds sign<<31 | magnitude (encode.c:112-131).
 * plane < 0 only happens for all-zero luma: every symbol is 0. */
static int put_plane(sink *s, const uint32_t *v, int num, int plane)
{
	if (plane < 0) {
		s->run += num;
		return 0;
	}
	int r;
	for (int i = 0; i < num; ++i) {
		uint32_t m = v[i] & 0x1fffffffu;
		if (m >> (plane + 1))
			continue;
		if ((m >> plane) & 1) {
			if ((r = sink_run(s)))
				return r;
			if ((r = sink_raw(s, (int)(v[i] >> 31))))
				return r;
		} else {
			++s->run;
		}
	}
	for (int i = 0; i < num; ++i) {
		uint32_t m = v[i] & 0x1fffffffu;
		if (!(m >> (plane + 1)))
			continue;
		if ((r = sink_raw(s, (int)((m >> plane) & 1))))
			return r;
	}
	return 0;
}

/* encode.c:112-131: two's complement -> sign<<31 | (mag & 0x1fffffff); returns plane count */
static int to_sign_magnitude(uint32_t *dst, const int *src, size_t num)
{
	int max = 0;
	for (size_t i = 0; i < num; ++i) {
		int sgn = src[i] < 0;
		int mag = sgn ? -src[i] : src[i];
		if (mag > max)
			max = mag;
		dst[i] = ((uint32_t)sgn << 31) | ((uint32_t)mag & 0x1fffffffu);
	}
	return 1 + orc_ilog2(max);
}

static int *pixels_to_coefs(const uint8_t *pix, int W, int H, int C)
{
	size_t n = (size_t)W * H * C;
	int *img = malloc(sizeof(int) * n);
	for (size_t i = 0; i < n; ++i)
		img[i] = pix[i];
	if (C == 3)
		orc_rgb_to_ycocg(img, (long)W * H);   /* encode.c:155-156 */
	orc_forward(img, W, H, C, 8);             /* encode.c:159 */
	return img;
}

int orc_stage_dump(const uint8_t *pix, int W, int H, int C, int *coef, int *lin, int *planes)
{
	if (W < 8 || H < 8 || W > ORC_MAX_SIDE || H > ORC_MAX_SIDE || (C != 1 && C != 3))
		return 1;
	orc_geom g;
	orc_geometry(&g, W, H, 8);
	size_t total = (size_t)W * H;
	int *img = pixels_to_coefs(pix, W, H, C);
	int *tmp = lin ? lin : malloc(sizeof(int) * total * C);
	orc_linearize(tmp, img, &g, C);
	if (coef)
		memcpy(coef, img, sizeof(int) * total * C);
	if (planes) {
		for (int c = 0; c < C; ++c) {
			int max = 0;
			for (size_t i = g.pixels[0]; i < total; ++i) {
				int a = tmp[c * total + i];
				a = a < 0 ? -a : a;
				if (a > max)
					max = a;
			}
			planes[c] = 1 + orc_ilog2(max);
		}
	}
	if (!lin)
		free(tmp);
	free(img);
	return 0;
}

/* encode.c:133-232 minus file I/O */
int orc_encode(const uint8_t *pix, int W, int H, int C, long capacity,
	uint8_t **out, size_t *out_len, orc_stats *st)
{
	if (W < 8 || H < 8 || W > ORC_MAX_SIDE || H > ORC_MAX_SIDE || (C != 1 && C != 3))
		return 1;                                   /* encode.c:140-146 */
	orc_geom g;
	orc_geometry(&g, W, H, 8);
	size_t total = (size_t)W * H;
	int *img = pixels_to_coefs(pix, W, H, C);
	int *lin = malloc(sizeof(int) * total * C);
	orc_linearize(lin, img, &g, C);               /* encode.c:160 */
	free(img);
	int rc = orc_encode_lin(lin, W, H, C, capacity, out, out_len, st);
	free(lin);
	return rc;
}

/* encode.c:163-230 on linearised coefficient planes lin[C][W*H] (what encode.c:160 leaves in `buffer`): the entropy
 * stage alone, for tests that feed it coefficients no 8-bit picture produces (up to 16 bit planes) */
int orc_encode_lin(const int *lin, int W, int H, int C, long capacity,
	uint8_t **out, size_t *out_len, orc_stats *st)
{
	if (W < 8 || H < 8 || W > ORC_MAX_SIDE || H > ORC_MAX_SIDE || (C != 1 && C != 3))
		return 1;
	orc_geom g;
	int levels = orc_geometry(&g, W, H, 8);
	size_t total = (size_t)W * H;
	uint32_t *sm = malloc(sizeof(uint32_t) * total * C);
	int planes[3] = { 0, 0, 0 };
	for (int c = 0; c < C; ++c)                   /* encode.c:163-165 */
		planes[c] = to_sign_magnitude(sm + c * total + g.pixels[0],
			lin + c * total + g.pixels[0], total - g.pixels[0]);

	sink s;
	memset(&s, 0, sizeof(s));
	s.cap = capacity;
	sink_byte(&s, 'W');                           /* encode.c:169-172 */
	sink_byte(&s, C == 3 ? '6' : '5');
	sink_byte(&s, (W - 1) & 255);
	sink_byte(&s, ((W - 1) >> 8) & 255);
	sink_byte(&s, (H - 1) & 255);
	sink_byte(&s, ((H - 1) >> 8) & 255);
	int meta = s.nacc + 8 * (int)s.len;
	for (int c = 0; c < C; ++c)                   /* encode.c:177-178 */
		put_root(&s, lin + c * total, g.pixels[0]);
	int root = s.nacc + 8 * (int)s.len;
	for (int c = 0; c < C; ++c)                   /* encode.c:181-182 */
		sink_vli(&s, planes[c]);
	int pmax = 0;
	for (int c = 0; c < C; ++c)
		if (planes[c] > pmax)
			pmax = planes[c];
	int layers_max = 2 * (levels > pmax ? levels : pmax) - 1;
	int stopped = 0;
	/* encode.c:189-220: luma runs one layer ahead of chroma, coarse levels one plane ahead of fine */
	if (pmax == planes[0])
		stopped = put_plane(&s, sm + g.pixels[0], g.pixels[1] - g.pixels[0], planes[0] - 1) != 0;
	for (int layer = 0; !stopped && layer < layers_max; ++layer) {
		for (int l = 0; !stopped && l < levels && l <= layer + 1; ++l) {
			int p = pmax - 1 - (layer + 1 - l);
			if (p < 0 || p >= planes[0])
				continue;
			stopped = put_plane(&s, sm + g.pixels[l], g.pixels[l + 1] - g.pixels[l], p) != 0;
		}
		for (int l = 0; !stopped && l < levels && l <= layer; ++l) {
			int p = pmax - 1 - (layer - l);
			for (int c = 1; !stopped && c < C; ++c) {
				if (p < 0 || p >= planes[c])
					continue;
				stopped = put_plane(&s, sm + c * total + g.pixels[l], g.pixels[l + 1] - g.pixels[l], p) != 0;
			}
		}
	}
	if (!stopped)
		sink_run(&s);                             /* encode.c:221 rle_flush */
	int bits = s.nacc + 8 * (int)s.len;           /* encode.c:226 */
	if (s.nacc)                                   /* bits.h:51-56 */
		sink_byte(&s, (int)(s.acc & 255));
	if (st) {
		st->meta_bits = meta;
		st->root_bits = root - meta;
		st->total_bits = bits;
		st->kib = ((int)s.len + 512) / 1024;
		st->levels = levels;
		st->tokens = s.tokens;
		st->raw_bits = s.raw;
		for (int c = 0; c < 3; ++c)
			st->planes[c] = planes[c];
	}
	free(sm);
	*out = s.buf ? s.buf : malloc(1);
	*out_len = s.len;
	return 0;
}

/* ---------------------------------------------------------- bit source (dec) */

typedef struct {
	const uint8_t *p;
	size_t len, pos;
	unsigned acc;
	int nacc;
	int order;  /* vli.h:24 */
	int run;    /* rle.h:25: 0 = need a new VLI, k>0 = k-1 zeros then a one */
} source;

/* bits.h:80-92 over bytes.h:97-105: -1 at end of data */
static int src_bit(source *s)
{
	if (!s->nacc) {
		if (s->pos >= s->len)
			return -1;
		s->acc = s->p[s->pos++];
		s->nacc = 8;
	}
	int b = s->acc & 1;
	s->acc >>= 1;
	--s->nacc;
	return b;
}

/* bits.h:94-106 */
static int src_bits(source *s, int n, int *out)
{
	int a = 0;
	for (int i = 0; i < n; ++i) {
		int b = src_bit(s);
		if (b < 0)
			return b;
		a |= b << i;
	}
	*out = a;
	return 0;
}

/* optional diagnostics hook (tools/merge_sim.c): bit position and order at every VLI token start */
void (*orc_trace_vli)(size_t bitpos, int order) = 0;

/* vli.h:86-101 */
static int src_vli(source *s)
{
	int sum = 0, b, rem = 0;
	if (orc_trace_vli)
		orc_trace_vli(s->pos * 8 - (size_t)s->nacc, s->order);
	while ((b = src_bit(s)) == 0) {
		sum += 1 << s->order;
		++s->order;
	}
	if (b < 0)
		return b;
	if (src_bits(s, s->order, &rem))
		return -1;
	s->order = s->order >= 2 ? s->order - 2 : 0;
	return sum + rem;
}

/* rle.h:66-77 get_rle */
static int src_symbol(source *s)
{
	if (s->run < 0)
		return s->run;
	if (!s->run) {
		s->run = src_vli(s);
		if (s->run < 0)
			return s->run;
		return !s->run;
	}
	return s->run-- == 1;
}

/* rle.h:91-103 rle_get_bit: a pending run must end exactly here (phantom one) */
static int src_raw(source *s)
{
	if (s->run < 0)
		return s->run;
	if (s->run > 0) {
		int one = src_symbol(s);
		if (one < 0)
			return one;
		if (one != 1)
			return -1;
	}
	return src_bit(s);
}

/* decode.c:119-134 */
static int get_root(source *s, int *val, int num)
{
	int cnt = src_vli(s);
	if (cnt < 0)
		return cnt;
	if (!cnt)
		return 0;
	for (int i = 0; i < num; ++i) {
		if (src_bits(s, cnt, &val[i]))
			return -1;
		if (val[i]) {
			int neg = src_bit(s);
			if (neg < 0)
				return neg;
			if (neg)
				val[i] = -val[i];
		}
	}
	return 0;
}

/* decode.c:67-100 without the flag bits: refinement class iff a higher bit is
 * already set.  v[] is sign<<31 | magnitude.  Whatever was decoded before an
 * error stays in v[] (the caller keeps partial planes, decode.c:204-205). */
static int get_plane(source *s, uint32_t *v, int num, int plane)
{
	int sh = plane & 31;   /* plane -1 (flat image) shifts zeros only */
	for (int i = 0; i < num; ++i) {
		uint32_t m = v[i] & 0x7fffffffu;
		if (plane >= 0 && (m >> (plane + 1)))
			continue;
		int b = src_symbol(s);
		if (b < 0)
			return b;
		if (!b)
			continue;
		v[i] |= 1u << sh;
		int neg = src_raw(s);
		if (neg < 0)
			return neg;
		v[i] |= (uint32_t)neg << 31;
	}
	if (plane < 0)
		return 0;
	for (int i = 0; i < num; ++i) {
		uint32_t m = v[i] & 0x7fffffffu;
		/* a coefficient that became significant in THIS plane is not refined now */
		if (!(m >> (plane + 1)))
			continue;
		int b = src_raw(s);
		if (b < 0)
			return b;
		v[i] |= (uint32_t)b << plane;
	}
	return 0;
}

/* decode.c:136-250: everything up to (and including) process().  On success
 * buf[c] (calloc'ed, g->pixels[levels_max] ints each) holds two's-complement
 * linearised coefficients, *level the finest level touched, missing[] the
 * per-(channel,level) count of planes not fully decoded. */
static int decode_entropy(const uint8_t *dwt, size_t len, long pixels_max,
	orc_geom *g, int *Cout, int **buf, int *level_out, int *missing, int *planes)
{
	if (len < 6 || dwt[0] != 'W' || (dwt[1] != '5' && dwt[1] != '6'))
		return 1;                                              /* decode.c:145-156 */
	int C = dwt[1] == '6' ? 3 : 1;
	int W = (dwt[2] | (dwt[3] << 8)) + 1;
	int H = (dwt[4] | (dwt[5] << 8)) + 1;
	if (W < 8 || H < 8 || W > ORC_MAX_SIDE || H > ORC_MAX_SIDE)
		return 1;                                              /* see ORC_MAX_SIDE in dwt_oracle.h */
	int levels = orc_geometry(g, W, H, 8);
	int levels_max = levels;
	if (pixels_max >= 0)                                       /* decode.c:165-171 */
		while (levels_max > 0 && g->pixels[levels_max] > pixels_max)
			--levels_max;
	size_t total = (size_t)g->pixels[levels_max];
	source s;
	memset(&s, 0, sizeof(s));
	s.p = dwt;
	s.len = len;
	s.pos = 6;
	for (int c = 0; c < 3; ++c)
		buf[c] = 0;
	for (int c = 0; c < C; ++c)
		buf[c] = calloc(total, sizeof(int));
	planes[0] = planes[1] = planes[2] = 0;
	for (int c = 0; c < C; ++c)                                /* decode.c:180-182 */
		if (get_root(&s, buf[c], g->pixels[0]))
			goto fail;
	for (int c = 0; c < C; ++c)                                /* decode.c:183-186 */
		if ((planes[c] = src_vli(&s)) < 0)
			goto fail;
	int pmax = 0;
	for (int c = 0; c < C; ++c)
		if (planes[c] > pmax)
			pmax = planes[c];
	int layers_max = 2 * (levels > pmax ? levels : pmax) - 1;
	memset(missing, 0, sizeof(int) * 48);
	for (int c = 0; c < C; ++c)
		for (int l = 0; l < levels; ++l)
			missing[c * 16 + l] = planes[c];
	int level = -1;
	int stop = levels_max == 0;                                /* decode.c:199-200 */
	if (!stop && pmax == planes[0]) {                          /* decode.c:201-207 */
		level = 0;
		if (get_plane(&s, (uint32_t *)buf[0] + g->pixels[0], g->pixels[1] - g->pixels[0], planes[0] - 1))
			stop = 1;
		else
			--missing[0];
	}
	for (int layer = 0; !stop && layer < layers_max; ++layer) { /* decode.c:208-243 */
		for (int l = 0; !stop && l < levels && l <= layer + 1; ++l) {
			if (l >= levels_max) {
				stop = 1;
				break;
			}
			int p = pmax - 1 - (layer + 1 - l);
			if (p < 0 || p >= planes[0])
				continue;
			if (level < l)
				level = l;
			if (get_plane(&s, (uint32_t *)buf[0] + g->pixels[l], g->pixels[l + 1] - g->pixels[l], p))
				stop = 1;
			else
				--missing[l];
		}
		for (int l = 0; !stop && l < levels && l <= layer; ++l) {
			if (l >= levels_max) {
				stop = 1;
				break;
			}
			int p = pmax - 1 - (layer - l);
			for (int c = 1; !stop && c < C; ++c) {
				if (p < 0 || p >= planes[c])
					continue;
				if (level < l)
					level = l;
				if (get_plane(&s, (uint32_t *)buf[c] + g->pixels[l], g->pixels[l + 1] - g->pixels[l], p))
					stop = 1;
				else
					--missing[c * 16 + l];
			}
		}
	}
	/* decode.c:249-250 process(): sign-magnitude -> two's complement on what was touched */
	for (int c = 0; c < C; ++c) {
		uint32_t *v = (uint32_t *)buf[c];
		for (int i = g->pixels[0]; i < g->pixels[level + 1]; ++i) {
			int mag = (int)(v[i] & 0x1fffffffu);
			buf[c][i] = (v[i] >> 31) ? -mag : mag;
		}
	}
	*level_out = level;
	*Cout = C;
	return 0;
fail:
	for (int c = 0; c < C; ++c)
		free(buf[c]);
	return 1;
}

/* entropy stage only: lin is int[C][W*H] (zero-filled beyond what was decoded) */
int orc_decode_stage(const uint8_t *dwt, size_t len, long pixels_max, int *lin, int *level, int *missing, int *planes)
{
	orc_geom g;
	int C, *buf[3];
	if (decode_entropy(dwt, len, pixels_max, &g, &C, buf, level, missing, planes))
		return 1;
	size_t total = (size_t)g.pixels[g.levels];
	memset(lin, 0, sizeof(int) * total * C);
	for (int c = 0; c < C; ++c) {
		memcpy(lin + c * total, buf[c], sizeof(int) * (size_t)g.pixels[*level + 1]);
		free(buf[c]);
	}
	return 0;
}

/* decode.c:136-268 minus file I/O */
int orc_decode(const uint8_t *dwt, size_t len, long pixels_max,
	uint8_t **pix, int *Wo, int *Ho, int *Co)
{
	orc_geom g;
	int C, *buf[3], level, missing[48], planes[3];
	if (decode_entropy(dwt, len, pixels_max, &g, &C, buf, &level, missing, planes))
		return 1;
	int out_levels = level + 1;                                /* decode.c:251-254 */
	int ow = g.widths[out_levels], oh = g.heights[out_levels];
	size_t on = (size_t)ow * oh;
	int *img = malloc(sizeof(int) * on * C);
	orc_reconstruct(img, buf, missing, &g, out_levels, C);     /* decode.c:257 */
	orc_inverse(img, ow, oh, C, 8);                            /* decode.c:258 */
	if (C == 3)
		orc_ycocg_to_rgb(img, (long)on);                       /* decode.c:262-263 */
	uint8_t *o = malloc(on * C + 1);
	for (size_t i = 0; i < on * C; ++i)
		o[i] = (uint8_t)clampi(img[i], 0, 255);                /* pnm.h:108 */
	free(img);
	for (int c = 0; c < C; ++c)
		free(buf[c]);
	*pix = o;
	*Wo = ow;
	*Ho = oh;
	*Co = C;
	return 0;
}

/* ---------------------------------------------------------------- synthetic */

static uint32_t mix32(uint32_t x, uint32_t y, uint32_t k, uint32_t seed)
{
	uint32_t u = x * 0x9E3779B1u ^ y * 0x85EBCA77u ^ k * 0xC2B2AE3Du ^ seed * 0x27D4EB2Fu;
	u ^= u >> 15;
	u *= 0x2C1B3C6Du;
	u ^= u >> 12;
	u *= 0x297A2D39u;
	u ^= u >> 15;
	return u;
}

static int tri(int t, int P)
{
	int r = t % (2 * P) - P;
	return r < 0 ? -r : r;
}

/* SURVEY.md §8d generator; integer-only so every box renders identical bytes. */
void orc_synth(uint8_t *pix, int W, int H, int C, uint32_t seed, int kind)
{
	for (int y = 0; y < H; ++y)
		for (int x = 0; x < W; ++x)
			for (int k = 0; k < C; ++k) {
				uint32_t u = mix32((uint32_t)x, (uint32_t)y, (uint32_t)k, seed);
				int p = kind ? (int)(u >> 24) : 40 + tri(x, 96) + tri(y, 64) + 10 * k + (int)(u >> 29);
				pix[((size_t)y * W + x) * C + k] = (uint8_t)p;
			}
}
