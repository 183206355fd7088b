/* chunk_stats.c — CPU study for unpack.hip's chunk records and speculation start (development aid; the oracle gives the
 * true token path):  how many pass-1 symbols the tokens that START in a 128-bit chunk hold (does a record of N symbols cover
 * the chunk?), tokens per chunk, and how often a speculative parse started `s` bits into a chunk (order 0) leaves the chunk
 * in the true state.     gcc -O2 -o /tmp/chunk_stats tools/chunk_stats.c && /tmp/chunk_stats 4096 1 */
#include "../oracle/dwt_oracle.c"
#include <stdio.h>

static uint8_t *truth;      /* per bit: 0 = not a token start, else order+1 */
static size_t nbits;
static void hook(size_t b, int o) { if (b < nbits) truth[b] = (uint8_t)(o + 1); }
static const uint8_t *S;
static uint64_t win(size_t b) { uint64_t lo = 0; for (int i = 0; i < 8; ++i) lo |= (uint64_t)S[(b >> 3) + i] << (8 * i);
	int r = b & 7; uint64_t hi = S[(b >> 3) + 8]; return r ? (lo >> r) | (hi << (64 - r)) : lo; }
static int step(size_t *b, int *o, uint64_t *run) { uint64_t w = win(*b); if (!w) return 0; int z = __builtin_ctzll(w); int top = *o + z; if (top > 31) return 0;
	if (run) *run = ((top ? (w >> (z + 1)) & ((1ull << top) - 1) : 0) + (1ull << top) - (1ull << *o));
	*b += z + top + 2; *o = top >= 2 ? top - 2 : 0; return 1; }

int main(int argc, char **argv)
{
	int W = argc > 1 ? atoi(argv[1]) : 1024, H = W, C = argc > 2 ? atoi(argv[2]) : 1;
	const int CH = 128;
	uint8_t *pix = malloc((size_t)W * H * C);
	orc_synth(pix, W, H, C, 0, argc > 3 ? atoi(argv[3]) : 0);
	uint8_t *dwt; size_t len;
	orc_encode(pix, W, H, C, 0, &dwt, &len, 0);
	nbits = len * 8;
	uint8_t *padded = calloc(len + 64, 1); memcpy(padded, dwt, len); S = padded;
	truth = calloc(nbits + 64, 1);
	orc_trace_vli = hook;
	uint8_t *back; int w, h, c;
	orc_decode(dwt, len, -1, &back, &w, &h, &c);
	size_t nch = nbits / CH;
	size_t hist_sym[12] = { 0 }, hist_tok[9] = { 0 }, p1 = 0, toks = 0;
	double syms = 0;
	for (size_t i = 1; i + 1 < nch; ++i) {
		uint64_t s = 0; int t = 0;
		for (size_t q = i * CH; q < (i + 1) * CH; ++q)
			if (truth[q]) { size_t b = q; int o = truth[q] - 1; uint64_t run = 0; step(&b, &o, &run); s += run + 1; ++t; }
		if (t < 2) continue;   /* (a chunk with one true token start is the edge of a refinement block) */
		++p1; toks += t; syms += (double)s;
		int k = 0; while (k < 11 && s > (16ull << k)) ++k;   /* buckets: <=16, <=32, ... <=16384, more */
		++hist_sym[k];
		int kt = t <= 8 ? 0 : t <= 12 ? 1 : t <= 16 ? 2 : t <= 24 ? 3 : t <= 32 ? 4 : t <= 40 ? 5 : t <= 48 ? 6 : t <= 56 ? 7 : 8;
		++hist_tok[kt];
	}
	printf("%dx%dx%d: %zu bytes, %zu chunks, %zu of them hold pass-1 tokens (%.1f%%): %.1f tokens, %.0f symbols per such chunk\n", W, H, C, len, nch, p1,
		100.0 * p1 / nch, (double)toks / p1, syms / p1);
	size_t acc = 0;
	for (int k = 0; k < 12; ++k) { acc += hist_sym[k]; printf("  symbols <= %6d : %5.1f%% (cumulative %5.1f%%)\n", k < 11 ? 16 << k : 1 << 30, 100.0 * hist_sym[k] / p1, 100.0 * acc / p1); }
	const char *tn[9] = { "<=8", "<=12", "<=16", "<=24", "<=32", "<=40", "<=48", "<=56", "<=64" };
	for (int k = 0; k < 9; ++k) printf("  tokens %-5s : %5.1f%%\n", tn[k], 100.0 * hist_tok[k] / p1);
	/* speculation started s bits into the chunk at order 0: is the exit state the true one? */
	for (int s0 = 0; s0 < CH; s0 += 16) {
		size_t good = 0, cons = 0;
		for (size_t i = 1; i + 2 < nch; i += 3) {
			size_t bnd = (i + 1) * CH, t = bnd; while (t < bnd + 64 && !truth[t]) ++t; if (t >= bnd + 64) continue;
			int any = 0; for (size_t q = i * CH; q < bnd; ++q) any |= truth[q] != 0; if (!any) continue;
			++cons;
			size_t b = i * CH + s0; int o = 0, al = 1; while (al && b < bnd) al = step(&b, &o, 0);
			good += al && b == t && truth[t] == o + 1;
		}
		printf("  speculation from bit %3d of the chunk: exit on the true path in %.2f%% of %zu chunks\n", s0, 100.0 * good / cons, cons);
	}
	return 0;
}
