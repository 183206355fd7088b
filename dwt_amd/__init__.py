"""dwt_amd — host-side mirror of the xdsopl/dwt encode/decode hot path on MI355X.

Everything here is plumbing over the C ABI in include/dwtx.h (libdwtx.so,
hand-written HIP for gfx950): torch supplies device memory and streams, the
library does the work.  Function names follow the reference's
(`transformation`, `linearization`, `reconstruction`, ... in encode.c/decode.c).
There is no CPU fallback: without the built library or without a GPU the calls
raise.
"""
import ctypes as C

from . import _lib
from ._lib import Geom, Stats, StreamInfo, DecodeInfo, Index, SegIndex, INDEX_MAGIC, INDEX_MAX_SEGS, LIB_PATH  # noqa: F401

__all__ = ["Context", "DwtxError", "compute_lengths", "geometry", "Geom", "Stats"]


class DwtxError(RuntimeError):
    def __init__(self, rc, what):
        msg = _lib.load().dwtx_last_error().decode(errors="replace")
        super().__init__(f"{what} failed with {rc}: {msg}")
        self.rc = rc


def _check(rc, what):
    if rc != 0:
        raise DwtxError(rc, what)


def compute_lengths(W, H, N0=8):
    """utils.h:28 compute_lengths -> (levels, lengths, pixels, widths, heights)."""
    lib = _lib.load()
    arr = [(C.c_int * 16)() for _ in range(4)]
    levels = lib.dwtx_compute_lengths(arr[0], arr[1], arr[2], arr[3], W, H, N0)
    return (levels,) + tuple(list(a[: levels + 1]) for a in arr)


def geometry(W, H):
    g = Geom()
    _check(_lib.load().dwtx_geometry(C.byref(g), W, H), "dwtx_geometry")
    return g


def _ptr(t):
    return C.c_void_p(t.data_ptr())


class Context:
    """One HIP device + stream + scratch arena (dwtx_ctx)."""

    def __init__(self, device=0, stream=None):
        import torch

        self.torch = torch
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("dwt_amd needs a HIP device; there is no CPU path")
        self.device = torch.device("cuda", device)
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        h = C.c_void_p()
        _check(self.lib.dwtx_ctx_create_on_stream(device, C.c_void_p(stream), C.byref(h)), "dwtx_ctx_create_on_stream")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.dwtx_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name, value=1):
        """Diagnostic switch of the context (enum dwtx_option in include/dwtx.h; tests and tools only)."""
        _check(self.lib.dwtx_ctx_set_option(self.h, _lib.OPTIONS[name], int(value)), "dwtx_ctx_set_option")

    def get_option(self, name):
        return self.lib.dwtx_ctx_get_option(self.h, _lib.OPTIONS[name])

    def set_index(self, offered=None, wanted=0):
        """Sidecar indices for the decode calls that follow (dwtx_ctx_set_index): `offered` is an array of Index
        made by new_indices()/an earlier decode (entry i goes with image i of a call), `wanted` the number of
        entries to collect.  Returns the array that will receive them (or None).  set_index() ends it."""
        out = (Index * wanted)() if wanted else None
        self._index = (offered, out)   # the library keeps the pointers
        _check(self.lib.dwtx_ctx_set_index(self.h, C.cast(offered, C.c_void_p) if offered is not None else None,
                                           C.cast(out, C.c_void_p) if out is not None else None), "dwtx_ctx_set_index")
        return out

    def sync(self):
        _check(self.lib.dwtx_sync(self.h), "dwtx_sync")

    def synth_pixels(self, n, H, W, C_, seed0=0, kind=0):
        """Synthetic uint8 frames [n,H,W,C] rendered on the device (SURVEY.md §8d generator)."""
        out = self.torch.empty((n, H, W, C_), dtype=self.torch.uint8, device=self.device)
        _check(self.lib.dwtx_synth_pixels(self.h, _ptr(out), W, H, C_, n, seed0, kind), "dwtx_synth_pixels")
        return out

    # -- stage kernels -------------------------------------------------------

    def planes_from_pixels(self, pix):
        """uint8 [n,H,W,C] interleaved -> int32 [n*C,H,W] planar (YCoCg-R if C==3)."""
        torch = self.torch
        n, H, W, Cn = pix.shape
        assert pix.dtype == torch.uint8 and pix.is_contiguous() and pix.device == self.device
        out = torch.empty((n * Cn, H, W), dtype=torch.int32, device=self.device)
        _check(self.lib.dwtx_planes_from_pixels(self.h, _ptr(out), _ptr(pix), W, H, Cn, n), "dwtx_planes_from_pixels")
        return out

    def pixels_from_planes(self, planes, C_):
        torch = self.torch
        nC, H, W = planes.shape
        n = nC // C_
        assert planes.dtype == torch.int32 and planes.is_contiguous()
        out = torch.empty((n, H, W, C_), dtype=torch.uint8, device=self.device)
        _check(self.lib.dwtx_pixels_from_planes(self.h, _ptr(out), _ptr(planes), W, H, C_, n), "dwtx_pixels_from_planes")
        return out

    def transformation_fwd(self, planes, out=None):
        """encode.c:16 transformation: int32 [P,H,W] -> Mallat pyramid [P,H,W]."""
        torch = self.torch
        P, H, W = planes.shape
        assert planes.dtype == torch.int32 and planes.is_contiguous()
        if out is None:
            out = torch.empty_like(planes)
        _check(self.lib.dwtx_transformation_fwd(self.h, _ptr(out), _ptr(planes), W, H, P), "dwtx_transformation_fwd")
        return out

    def transformation_inv(self, pyr, out=None):
        """decode.c:16 transformation (inverse)."""
        torch = self.torch
        P, H, W = pyr.shape
        assert pyr.dtype == torch.int32 and pyr.is_contiguous()
        if out is None:
            out = torch.empty_like(pyr)
        _check(self.lib.dwtx_transformation_inv(self.h, _ptr(out), _ptr(pyr), W, H, P), "dwtx_transformation_inv")
        return out

    def transformation_fwd_pixels(self, pix, rings16=True, out=None):
        """encode.c:155-159 in one pass as dwtx_encode_device runs it: uint8 [n,H,W,C] -> (int32 pyramid [n*C,H,W],
        int16 planes [n*C,H,W] or None, mask of the ring levels that live in the int16 planes)."""
        torch = self.torch
        n, H, W, Cn = pix.shape
        assert pix.dtype == torch.uint8 and pix.is_contiguous()
        pyr, r16 = out if out is not None else (None, None)
        if pyr is None:
            pyr = torch.zeros((n * Cn, H, W), dtype=torch.int32, device=pix.device)
        if rings16 and r16 is None:
            r16 = torch.zeros((n * Cn, H, W), dtype=torch.int16, device=pix.device)
        mask = C.c_uint(0)
        _check(self.lib.dwtx_transformation_fwd_pixels(self.h, _ptr(pyr), _ptr(r16) if rings16 else None, C.byref(mask), _ptr(pix),
                                                       W, H, Cn, n), "dwtx_transformation_fwd_pixels")
        return pyr, (r16 if rings16 else None), mask.value

    def transformation_inv_pixels(self, pyr, r16, mask, Cn, out=None):
        """decode.c:258-264 as dwtx_decode_device runs it: (pyramid, int16 ring planes, mask) -> uint8 [n,H,W,C]."""
        torch = self.torch
        P, H, W = pyr.shape
        n = P // Cn
        if out is None:
            out = torch.empty((n, H, W, Cn), dtype=torch.uint8, device=pyr.device)
        _check(self.lib.dwtx_transformation_inv_pixels(self.h, _ptr(out), _ptr(pyr), _ptr(r16) if mask else None, mask, W, H, Cn, n),
               "dwtx_transformation_inv_pixels")
        return out

    def linearization(self, pyr):
        """encode.c:32 linearization: pyramid [P,H,W] -> Hilbert-linearised [P,H*W]."""
        torch = self.torch
        P, H, W = pyr.shape
        assert pyr.dtype == torch.int32 and pyr.is_contiguous()
        out = torch.empty((P, H * W), dtype=torch.int32, device=self.device)
        _check(self.lib.dwtx_linearization(self.h, _ptr(out), _ptr(pyr), W, H, P), "dwtx_linearization")
        return out

    def reconstruction(self, lin, W, H, C_, levels_out=None, missing=None):
        """decode.c:32 reconstruction: [n*C, W*H] -> pyramid [n*C, h', w'] of the first levels_out levels."""
        torch = self.torch
        P = lin.shape[0]
        n = P // C_
        g = geometry(W, H)
        if levels_out is None:
            levels_out = g.levels
        ow, oh = g.widths[levels_out], g.heights[levels_out]
        out = torch.empty((P, oh, ow), dtype=torch.int32, device=self.device)
        mp = C.c_void_p(0)
        if missing is not None:
            assert missing.dtype == torch.int32 and missing.numel() == n * 48 and missing.is_contiguous()
            mp = _ptr(missing)
        _check(self.lib.dwtx_reconstruction(self.h, _ptr(out), _ptr(lin), mp, levels_out, W, H, C_, n),
               "dwtx_reconstruction")
        return out

    def encode_planes(self, lin, W, H, C_, capacity=0, out_stride=None):
        """encode.c:166-221 on linearised planes [n*C, W*H] -> (list of bytes, list of StreamInfo)."""
        import numpy as np

        torch = self.torch
        n = lin.shape[0] // C_
        assert lin.dtype == torch.int32 and lin.is_contiguous()
        if out_stride is None:
            out_stride = capacity if capacity > 0 else 2 * W * H * C_ + 4096
            out_stride = (out_stride + 8 + 3) // 4 * 4
        out = torch.empty((n, out_stride), dtype=torch.uint8, device=self.device)
        info = torch.empty((n, C.sizeof(StreamInfo)), dtype=torch.uint8, device=self.device)
        _check(self.lib.dwtx_encode_planes(self.h, _ptr(lin), W, H, C_, n, capacity, _ptr(out), out_stride, _ptr(info)),
               "dwtx_encode_planes")
        raw = info.cpu().numpy()
        infos = [StreamInfo.from_buffer_copy(raw[i].tobytes()) for i in range(n)]
        host = out.cpu().numpy()
        streams = []
        for i in range(n):
            if infos[i].error:
                raise DwtxError(-3, "dwtx_encode_planes (more than 16 bit planes)")
            if infos[i].nbytes > out_stride:
                raise DwtxError(-2, "dwtx_encode_planes (out_stride too small)")
            streams.append(host[i, : infos[i].nbytes].tobytes())
        return streams, infos

    def decode_planes(self, streams, W, H, C_, levels_max=-1):
        """decode.c:174-250 on a list of .dwt byte strings of one geometry ->
        (lin int32 [n*C, W*H] two's complement, list of DecodeInfo)."""
        import numpy as np

        torch = self.torch
        n = len(streams)
        stride = (max(len(s) for s in streams) + 64 + 7) // 8 * 8
        host = np.zeros((n, stride), dtype=np.uint8)
        for i, s in enumerate(streams):
            host[i, : len(s)] = np.frombuffer(s, dtype=np.uint8)
        dev = torch.from_numpy(host).to(self.device)
        lens = torch.tensor([len(s) for s in streams], dtype=torch.int64, device=self.device)
        lin = torch.empty((n * C_, W * H), dtype=torch.int32, device=self.device)
        infos = (DecodeInfo * n)()
        _check(self.lib.dwtx_decode_planes(self.h, _ptr(lin), _ptr(dev), stride, _ptr(lens), W, H, C_, n, levels_max,
                                           C.cast(infos, C.c_void_p)), "dwtx_decode_planes")
        self._keep = (dev, lens)   # kernels after the internal sync still read the streams
        return lin, list(infos)

    # -- whole images (host numpy in/out; what the CLIs do) ---------------------

    def encode(self, pix, capacity=0):
        """uint8 numpy [n,H,W,C] (or [H,W,C]) -> list of .dwt byte strings, list of Stats."""
        import numpy as np

        single = pix.ndim == 3
        pix = np.ascontiguousarray(pix[None] if single else pix, dtype=np.uint8)
        n, H, W, Cn = pix.shape
        stride = self.lib.dwtx_encode_bound(W, H, Cn) if capacity <= 0 else (capacity + 15) // 8 * 8
        out = np.empty((n, stride), dtype=np.uint8)
        lens = (C.c_size_t * n)()
        stats = (Stats * n)()
        _check(self.lib.dwtx_encode_images(self.h, pix.ctypes.data, W, H, Cn, n, capacity, out.ctypes.data, stride,
                                           C.cast(lens, C.c_void_p), C.cast(stats, C.c_void_p)), "dwtx_encode_images")
        streams = [out[i, : lens[i]].tobytes() for i in range(n)]
        return (streams[0], stats[0]) if single else (streams, list(stats))

    def decode(self, streams, pixels_max=-1):
        """.dwt byte string (or list of same-geometry ones) -> uint8 numpy [h,w,C] (or list); None if unreadable."""
        import numpy as np

        single = isinstance(streams, (bytes, bytearray))
        lst = [streams] if single else list(streams)
        n = len(lst)
        if len(lst[0]) < 6:
            raise DwtxError(-3, "dwtx_decode_images (short header)")
        W = (lst[0][2] | (lst[0][3] << 8)) + 1
        H = (lst[0][4] | (lst[0][5] << 8)) + 1
        Cn = 3 if lst[0][1:2] == b"6" else 1
        stride = (max(len(s) for s in lst) + 64 + 7) // 8 * 8
        host = np.zeros((n, stride), dtype=np.uint8)
        for i, s in enumerate(lst):
            host[i, : len(s)] = np.frombuffer(bytes(s), dtype=np.uint8)
        lens = (C.c_size_t * n)(*[len(s) for s in lst])
        pstride = W * H * Cn
        pix = np.empty((n, pstride), dtype=np.uint8)
        ow, oh, oc = (C.c_int * n)(), (C.c_int * n)(), (C.c_int * n)()
        rc = self.lib.dwtx_decode_images(self.h, host.ctypes.data, stride, C.cast(lens, C.c_void_p), n, pixels_max,
                                         pix.ctypes.data, pstride, ow, oh, oc)
        if rc == -1 and n == 1:   # one unreadable stream is an error code (decode.c exits 1), in a batch it is a missing picture
            return None if single else [None]
        _check(rc, "dwtx_decode_images")
        outs = [pix[i, : ow[i] * oh[i] * oc[i]].reshape(oh[i], ow[i], oc[i]).copy() if ow[i] else None for i in range(n)]
        return outs[0] if single else outs

    # -- whole images, device resident (what bench.py times) --------------------

    def encode_device(self, pix, capacity=0, out=None, info=None):
        """uint8 device tensor [n,H,W,C] -> (streams uint8 [n,stride], info uint8 [n,sizeof(StreamInfo)]) on device; async."""
        torch = self.torch
        n, H, W, Cn = pix.shape
        assert pix.dtype == torch.uint8 and pix.is_contiguous()
        stride = self.lib.dwtx_encode_bound(W, H, Cn) if capacity <= 0 else (capacity + 15) // 8 * 8
        if out is None:
            out = torch.empty((n, stride), dtype=torch.uint8, device=self.device)
        if info is None:
            info = torch.empty((n, C.sizeof(StreamInfo)), dtype=torch.uint8, device=self.device)
        _check(self.lib.dwtx_encode_device(self.h, _ptr(pix), W, H, Cn, n, capacity, _ptr(out), out.shape[1], _ptr(info)),
               "dwtx_encode_device")
        return out, info

    def stream_lengths(self, info):
        """int64 device tensor of stream byte lengths from the info records of encode_device."""
        off = StreamInfo.nbytes.offset
        return info[:, off:off + 8].contiguous().view(self.torch.int64).view(-1)

    def pack_streams(self, streams, lens, out, offsets=None):
        """dwtx_pack_streams: the n streams of a batch (uint8 [n, stride], int64 lens on the device) moved together into the
        flat uint8 tensor `out`, stream i at sum(round8(lens[:i])); offsets: optional int64 [n + 1] device tensor."""
        torch = self.torch
        n, stride = streams.shape
        assert streams.dtype == torch.uint8 and streams.is_contiguous() and out.dtype == torch.uint8 and out.is_contiguous()
        assert lens.dtype == torch.int64 and lens.numel() == n and lens.is_contiguous()
        assert offsets is None or (offsets.dtype == torch.int64 and offsets.numel() == n + 1)
        _check(self.lib.dwtx_pack_streams(self.h, _ptr(out), out.numel(), _ptr(offsets) if offsets is not None else None,
                                          _ptr(streams), stride, _ptr(lens), n), "dwtx_pack_streams")
        return out

    def decode_device(self, streams, lens, W, H, C_, levels_max=-1, out=None):
        """device streams [n,stride] + int64 lens -> (uint8 [n, W*H*C] pixels, list of DecodeInfo); syncs once."""
        torch = self.torch
        n, stride = streams.shape
        assert streams.dtype == torch.uint8 and streams.is_contiguous() and stride % 8 == 0
        assert lens.dtype == torch.int64 and lens.numel() == n
        if out is None:
            out = torch.empty((n, W * H * C_), dtype=torch.uint8, device=self.device)
        infos = (DecodeInfo * n)()
        _check(self.lib.dwtx_decode_device(self.h, _ptr(streams), stride, _ptr(lens), W, H, C_, n, levels_max,
                                           _ptr(out), W * H * C_, C.cast(infos, C.c_void_p)), "dwtx_decode_device")
        return out, list(infos)
