/* merge_sim.c — CPU study of how fast speculative VLI parses rejoin the true token path
 * (development aid for dwt_amd/csrc/unpack.hip; uses the oracle for ground truth). */
#include "../oracle/dwt_oracle.c"
#include <stdio.h>

static uint8_t *truth;      /* per bit: 0 = not a token start, else order+1 */
static size_t nbits;
static void hook(size_t b, int o) { if (b < nbits) truth[b] = (uint8_t)(o + 1); }

static const uint8_t *S;
static uint64_t win(size_t b) { uint64_t v = 0; for (int i = 0; i < 9; ++i) { size_t k = (b >> 3) + i; if (k * 8 < nbits + 64) v |= (i < 8) ? (uint64_t)S[k] << (8 * i) : 0; }
	uint64_t lo = v; int r = b & 7; uint64_t hi = S[(b >> 3) + 8]; return r ? (lo >> r) | (hi << (64 - r)) : lo; }
static int step(size_t *b, int *o) { uint64_t w = win(*b); if (!w) return 0; int z = __builtin_ctzll(w); int top = *o + z; if (top > 31) return 0;
	*b += z + top + 2; *o = top >= 2 ? top - 2 : 0; return 1; }

int main(int argc, char **argv)
{
	int W = argc > 1 ? atoi(argv[1]) : 1024, H = W, C = argc > 2 ? atoi(argv[2]) : 1;
	int CH = argc > 3 ? atoi(argv[3]) : 128;
	uint8_t *pix = malloc((size_t)W * H * C);
	orc_synth(pix, W, H, C, 0, 0);
	uint8_t *dwt; size_t len;
	orc_encode(pix, W, H, C, 0, &dwt, &len, 0);
	nbits = len * 8;
	uint8_t *padded = calloc(len + 64, 1); memcpy(padded, dwt, len); S = padded;
	truth = calloc(nbits + 64, 1);
	orc_trace_vli = hook;
	uint8_t *back; int w, h, c;
	orc_decode(dwt, len, -1, &back, &w, &h, &c);
	size_t ntok = 0; for (size_t i = 0; i < nbits; ++i) ntok += truth[i] != 0;
	printf("%dx%dx%d: %zu bytes, %zu true VLI tokens\n", W, H, C, len, ntok);
	/* for each true token start t (sampled), start spec parses at the chunk boundary before it with
	 * various seeds and measure how many bits until a spec path lands on a true token start with the true order */
	size_t nch = nbits / CH;
	for (int nfam = 1; nfam <= 4; nfam *= 2) for (int lookback = 1; lookback <= 8; lookback *= 2) {
		size_t joined = 0, considered = 0;
		for (size_t ci = lookback + 1; ci + 2 < nch; ci += 7) {
			/* is there a true token path through the boundary ci*CH?  find first true token start >= ci*CH within 64 bits,
			   and require the chunk before to be in a pass-1 region (a true start within it) */
			size_t bnd = ci * CH, t = bnd; while (t < bnd + 64 && !truth[t]) ++t; if (t >= bnd + 64) continue;
			/* true path continuity check: the previous chunk has token starts too */
			size_t u = bnd - CH; int any = 0; for (size_t q = u; q < bnd; ++q) any |= truth[q] != 0; if (!any) continue;
			++considered;
			int ok = 0;
			for (int f = 0; f < nfam && !ok; ++f) {
				size_t b = (ci - lookback) * CH + (f & 1); int o = (f >> 1);   /* families: parity x start order 0/1 */
				int alive = 1;
				while (alive && b < bnd) alive = step(&b, &o);
				ok = alive && b == t && truth[t] == o + 1;
			}
			joined += ok;
		}
		printf("chunk %d fam %d lookback %d chunks: spec state == true state at boundary in %.3f%% of %zu boundaries\n",
			CH, nfam, lookback, 100.0 * joined / considered, considered);
	}
	/* iterated linking exactly as k_spec/k_link do it: out[t][i] = exit of the parse of chunk i entered at out[t-1][i-1] */
	for (int fam = 0; fam < 2; ++fam) {
		uint32_t *prev = malloc(4 * nch), *cur = malloc(4 * nch);
		for (size_t i = 0; i < nch; ++i) { size_t b = i * CH + fam; int o = 0, al = 1; while (al && b < (i + 1) * CH) al = step(&b, &o);
			prev[i] = al ? (uint32_t)((b - (i + 1) * CH) | (o << 8)) : 0xffff; }
		for (int t = 1; t <= 12; ++t) {
			size_t changed = 0, considered = 0, truthmatch = 0;
			cur[0] = 0xffff;
			for (size_t i = 1; i < nch; ++i) {
				uint32_t in = prev[i - 1];
				if (in == 0xffff) { cur[i] = 0xffff; } else {
					size_t b = i * CH + (in & 0xff); int o = in >> 8, al = 1; while (al && b < (i + 1) * CH) al = step(&b, &o);
					cur[i] = al ? (uint32_t)((b - (i + 1) * CH) | (o << 8)) : 0xffff; }
				int any = 0; for (size_t q = i * CH; q < (i + 1) * CH; q += 4) any |= truth[q] | truth[q+1] | truth[q+2] | truth[q+3];
				if (!any) continue;
				++considered; changed += cur[i] != prev[i];
				if (cur[i] != 0xffff) { size_t eb = (i + 1) * CH + (cur[i] & 0xff); truthmatch += truth[eb] == (cur[i] >> 8) + 1; }
			}
			printf("fam %d round %2d: exits that still moved %.4f%% ; exit on the true path %.3f%% (of %zu chunks with true tokens)\n",
				fam, t, 100.0 * changed / considered, 100.0 * truthmatch / considered, considered);
			uint32_t *tmp = prev; prev = cur; cur = tmp;
		}
	}
	return 0;
}
