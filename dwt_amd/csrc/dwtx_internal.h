// dwtx_internal.h — shared internals of libdwtx (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/dwtx.h"

#define DWTX_SCRATCH_SLOTS 24

struct dwtx_linplan;

struct dwtx_ctx {
	int device;
	hipStream_t stream;
	bool own_stream;
	void *scratch[DWTX_SCRATCH_SLOTS];
	size_t scratch_bytes[DWTX_SCRATCH_SLOTS];
	dwtx_linplan *plans;   // per-geometry Hilbert block tables (linearize.hip)
	hipStream_t aux;       // second stream: half of a decode batch runs here so that one half's serial
	hipEvent_t ev[2];      // token walk overlaps the other half's parallel kernels (unpack.hip)
	bool have_aux;
};

void dwtx_free_plans(dwtx_ctx *ctx);

void dwtx_set_error(const char *fmt, ...);
// grow-only per-slot device scratch; contents undefined after a grow
void *dwtx_scratch(dwtx_ctx *ctx, int slot, size_t bytes);

#define DWTX_HIP(call)                                                          \
	do {                                                                        \
		hipError_t e_ = (call);                                                 \
		if (e_ != hipSuccess) {                                                 \
			dwtx_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
			return DWTX_ERR_DEVICE;                                             \
		}                                                                       \
	} while (0)

#define DWTX_LAUNCH_CHECK() DWTX_HIP(hipGetLastError())

static inline int dwtx_cdiv(int a, int b) { return (a + b - 1) / b; }

// scratch slot assignment
enum {
	SLOT_LIFT_A = 0,
	SLOT_LIFT_B = 1,
};

// C truncating division by 2 and 4 on the device (cdf53.h:13,20 use `/`)
__device__ __forceinline__ int tdiv2(int a) { return (a + (int)((unsigned)a >> 31)) >> 1; }
__device__ __forceinline__ int tdiv4(int a) { return (a + ((a >> 31) & 3)) >> 2; }
