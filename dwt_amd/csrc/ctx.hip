// ctx.hip — context, device memory helpers, host-side geometry.
#include "dwtx_internal.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static thread_local char g_err[512] = "";

void dwtx_set_error(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
}

extern "C" const char *dwtx_last_error(void)
{
	return g_err;
}

static int ctx_create(int device, void *stream, bool make_stream, dwtx_ctx **out)
{
	if (!out)
		return DWTX_ERR_ARG;
	int count = 0;
	DWTX_HIP(hipGetDeviceCount(&count));
	if (device < 0 || device >= count) {
		dwtx_set_error("device %d out of range (%d visible)", device, count);
		return DWTX_ERR_ARG;
	}
	DWTX_HIP(hipSetDevice(device));
	dwtx_ctx *c = (dwtx_ctx *)calloc(1, sizeof(dwtx_ctx));
	if (!c)
		return DWTX_ERR_NOMEM;
	c->device = device;
	if (!make_stream) {
		c->stream = (hipStream_t)stream;
		c->own_stream = false;
	} else {
		hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
		if (e != hipSuccess) {
			free(c);
			dwtx_set_error("hipStreamCreate -> %s", hipGetErrorString(e));
			return DWTX_ERR_DEVICE;
		}
		c->own_stream = true;
	}
	*out = c;
	return DWTX_OK;
}

extern "C" int dwtx_ctx_create(int device, dwtx_ctx **out)
{
	return ctx_create(device, nullptr, true, out);
}

extern "C" int dwtx_ctx_create_on_stream(int device, void *stream, dwtx_ctx **out)
{
	return ctx_create(device, stream, false, out);
}

// Makes the context's device current for the calling thread (the HIP runtime keeps one current device per thread;
// streams, events and allocations belong to the device they were made on).
int dwtx_enter(dwtx_ctx *c)
{
	if (!c)
		return DWTX_ERR_ARG;
	int cur = -1;
	if (hipGetDevice(&cur) == hipSuccess && cur == c->device)
		return DWTX_OK;
	DWTX_HIP(hipSetDevice(c->device));
	return DWTX_OK;
}

#ifdef DWTX_DEBUG_HOOKS
int dwtx_debug_check_device(dwtx_ctx *c, const char *file, int line)
{
	int cur = -1;
	if (hipGetDevice(&cur) != hipSuccess || cur != c->device) {
		dwtx_set_error("%s:%d kernels launched with device %d current, the context lives on device %d", file, line, cur, c->device);
		return DWTX_ERR_DEVICE;
	}
	return DWTX_OK;
}
#endif

extern "C" int dwtx_ctx_set_option(dwtx_ctx *c, int option, long value)
{
	if (!c || option < 0 || option >= DWTX_OPT_COUNT)
		return DWTX_ERR_ARG;
	c->opt[option] = value;
	return DWTX_OK;
}

extern "C" long dwtx_ctx_get_option(dwtx_ctx *c, int option)
{
	return c && option >= 0 && option < DWTX_OPT_COUNT ? c->opt[option] : 0;
}

extern "C" int dwtx_ctx_set_index(dwtx_ctx *c, const dwtx_index *in, dwtx_index *out)
{
	if (!c)
		return DWTX_ERR_ARG;
	c->index_in = in;
	c->index_out = out;
	c->index_base = 0;
	return DWTX_OK;
}

int dwtx_need_side_streams(dwtx_ctx *c, bool more)
{
	if (!c->have_aux) {
		DWTX_HIP(hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking));
		for (int i = 0; i < 4; ++i)
			DWTX_HIP(hipEventCreateWithFlags(&c->ev[i], hipEventDisableTiming));
		c->have_aux = true;
	}
	if (more && !c->have_more) {
		for (int i = 0; i < 2; ++i)
			DWTX_HIP(hipStreamCreateWithFlags(&c->more[i], hipStreamNonBlocking));
		for (int i = 0; i < 8; ++i)
			DWTX_HIP(hipEventCreateWithFlags(&c->pev[i], hipEventDisableTiming));
		c->have_more = true;
	}
	return DWTX_OK;
}

int dwtx_encoder_part(dwtx_ctx *c, int k, dwtx_ctx **part)
{
	if (k < 0 || k >= DWTX_ENC_PARTS)
		return DWTX_ERR_ARG;
	if (!c->have_enc_ev) {
		for (int i = 0; i < 2 * DWTX_ENC_PARTS + 1; ++i)
			DWTX_HIP(hipEventCreateWithFlags(&c->enc_ev[i], hipEventDisableTiming));
		c->have_enc_ev = true;
	}
	if (k == 0) {   // part 0 runs on the caller's stream: the context itself, with the scratch it already has for small batches
		*part = c;
		return DWTX_OK;
	}
	if (!c->enc_part[k]) {
		int rc = dwtx_need_side_streams(c, true);
		if (rc)
			return rc;
		hipStream_t st = k == 1 ? c->aux : c->more[k - 2];
		if ((rc = ctx_create(c->device, (void *)st, false, &c->enc_part[k])))
			return rc;
	}
	memcpy(c->enc_part[k]->opt, c->opt, sizeof(c->opt));
	*part = c->enc_part[k];
	return DWTX_OK;
}

extern "C" void dwtx_ctx_destroy(dwtx_ctx *c)
{
	if (!c)
		return;
	(void)hipSetDevice(c->device);
	(void)hipStreamSynchronize(c->stream);
	for (int k = 0; k < DWTX_ENC_PARTS; ++k)
		if (c->enc_part[k])
			dwtx_ctx_destroy(c->enc_part[k]);
	if (c->have_enc_ev)
		for (int i = 0; i < 2 * DWTX_ENC_PARTS + 1; ++i)
			(void)hipEventDestroy(c->enc_ev[i]);
	dwtx_free_plans(c);
	for (int i = 0; i < DWTX_SCRATCH_SLOTS; ++i)
		if (c->scratch[i])
			(void)hipFree(c->scratch[i]);
	if (c->have_more) {
		for (int i = 0; i < 2; ++i)
			(void)hipStreamDestroy(c->more[i]);
		for (int i = 0; i < 8; ++i)
			(void)hipEventDestroy(c->pev[i]);
	}
	if (c->have_aux) {
		(void)hipStreamDestroy(c->aux);
		for (int i = 0; i < 4; ++i)
			(void)hipEventDestroy(c->ev[i]);
	}
	if (c->have_copy) {
		(void)hipStreamDestroy(c->copy);
		for (int i = 0; i < 6; ++i)
			(void)hipEventDestroy(c->cev[i]);
	}
	if (c->own_stream)
		(void)hipStreamDestroy(c->stream);
	free(c);
}

int dwtx_need_copy_stream(dwtx_ctx *c)
{
	if (c->have_copy)
		return DWTX_OK;
	DWTX_ENTER(c);
	DWTX_HIP(hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking));
	for (int i = 0; i < 6; ++i)
		DWTX_HIP(hipEventCreateWithFlags(&c->cev[i], hipEventDisableTiming));
	c->have_copy = true;
	return DWTX_OK;
}

// Page-locked host memory: transfers from/to it run asynchronously, so the host-buffer entry points can
// overlap them with kernels (pageable buffers work too, the runtime then stages them).
extern "C" void *dwtx_host_alloc(dwtx_ctx *c, size_t bytes)
{
	void *p = nullptr;
	if (dwtx_enter(c))
		return nullptr;
	hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
	if (e != hipSuccess) {
		dwtx_set_error("hipHostMalloc(%zu) -> %s", bytes, hipGetErrorString(e));
		return nullptr;
	}
	return p;
}

extern "C" void dwtx_host_free(dwtx_ctx *c, void *host)
{
	if (!host || dwtx_enter(c))
		return;
	(void)hipStreamSynchronize(c->stream);
	if (c->have_copy)
		(void)hipStreamSynchronize(c->copy);
	(void)hipHostFree(host);
}

extern "C" int dwtx_sync(dwtx_ctx *c)
{
	DWTX_ENTER(c);
	DWTX_HIP(hipStreamSynchronize(c->stream));
	return DWTX_OK;
}

extern "C" void *dwtx_stream(dwtx_ctx *c)
{
	return (void *)c->stream;
}

extern "C" void *dwtx_malloc(dwtx_ctx *c, size_t bytes)
{
	void *p = nullptr;
	if (dwtx_enter(c))
		return nullptr;
	hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
	if (e != hipSuccess) {
		dwtx_set_error("hipMalloc(%zu) -> %s", bytes, hipGetErrorString(e));
		return nullptr;
	}
	return p;
}

extern "C" void dwtx_free(dwtx_ctx *c, void *dev)
{
	if (!dev || dwtx_enter(c))
		return;
	(void)hipStreamSynchronize(c->stream);
	(void)hipFree(dev);
}

extern "C" int dwtx_upload(dwtx_ctx *c, void *dev, const void *host, size_t bytes)
{
	DWTX_ENTER(c);
	DWTX_HIP(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, c->stream));
	DWTX_HIP(hipStreamSynchronize(c->stream));
	return DWTX_OK;
}

extern "C" int dwtx_download(dwtx_ctx *c, void *host, const void *dev, size_t bytes)
{
	DWTX_ENTER(c);
	DWTX_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->stream));
	DWTX_HIP(hipStreamSynchronize(c->stream));
	return DWTX_OK;
}

void *dwtx_scratch(dwtx_ctx *c, int slot, size_t bytes)
{
	if (slot < 0 || slot >= DWTX_SCRATCH_SLOTS)
		return nullptr;
	if (c->scratch_bytes[slot] >= bytes && c->scratch[slot])
		return c->scratch[slot];
	if (dwtx_enter(c))   // the allocation below belongs to the context's device, whatever the thread had current
		return nullptr;
	// kernels still in flight may use the old buffer
	(void)hipStreamSynchronize(c->stream);
	if (c->scratch[slot])
		(void)hipFree(c->scratch[slot]);
	c->scratch[slot] = nullptr;
	c->scratch_bytes[slot] = 0;
	size_t want = bytes + bytes / 8 + 256;
	void *p = nullptr;
	if (hipMalloc(&p, want) != hipSuccess) {
		dwtx_set_error("scratch hipMalloc(%zu) failed", want);
		return nullptr;
	}
	c->scratch[slot] = p;
	c->scratch_bytes[slot] = want;
	return p;
}

// ---- geometry (utils.h:9-40) ------------------------------------------------

static int floor_log2(int v)
{
	int l = -1;
	for (; v > 0; v >>= 1)
		++l;
	return l;
}

extern "C" int dwtx_compute_lengths(int *lengths, int *pixels, int *widths, int *heights, int W, int H, int N0)
{
	// sizes from fine to coarse: halve (round up) once, then again while the
	// halves are still >= N0 (utils.h:17-26)
	int ws[DWTX_MAX_LEVELS + 1], hs[DWTX_MAX_LEVELS + 1];
	int n = 0;
	ws[0] = W;
	hs[0] = H;
	do {
		ws[n + 1] = (ws[n] + 1) >> 1;
		hs[n + 1] = (hs[n] + 1) >> 1;
		++n;
	} while (n < DWTX_MAX_LEVELS - 1 && ws[n] >= N0 && hs[n] >= N0);
	for (int l = 0; l <= n; ++l) {
		int w = ws[n - l], h = hs[n - l];
		widths[l] = w;
		heights[l] = h;
		pixels[l] = w * h;
		int a = 1 << (floor_log2(w - 1) + 1);
		int b = 1 << (floor_log2(h - 1) + 1);
		lengths[l] = a > b ? a : b;
	}
	return n;
}

extern "C" int dwtx_geometry(dwtx_geom *g, int W, int H)
{
	if (!g || W < 1 || H < 1)
		return DWTX_ERR_ARG;
	g->levels = dwtx_compute_lengths(g->lengths, g->pixels, g->widths, g->heights, W, H, DWTX_MIN_LEN);
	return DWTX_OK;
}
