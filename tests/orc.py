"""numpy wrapper over oracle/liboracle.so — the CPU checker (test infrastructure only)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")
GOLDEN = os.path.join(ROOT, "tests", "golden")


class Geom(C.Structure):
    _fields_ = [("levels", C.c_int), ("widths", C.c_int * 16), ("heights", C.c_int * 16),
                ("pixels", C.c_int * 16), ("lengths", C.c_int * 16)]


class Stats(C.Structure):
    _fields_ = [("meta_bits", C.c_int), ("root_bits", C.c_int), ("total_bits", C.c_int), ("kib", C.c_int),
                ("planes", C.c_int * 3), ("levels", C.c_int), ("tokens", C.c_long), ("raw_bits", C.c_long)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(os.path.join(ORACLE_DIR, "liboracle.so"))
        L.orc_geometry.argtypes = [C.POINTER(Geom), C.c_int, C.c_int, C.c_int]
        L.orc_encode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_long,
                                 C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(Stats)]
        L.orc_encode_lin.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_long,
                                     C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(Stats)]
        L.orc_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_long, C.POINTER(C.c_void_p),
                                 C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_stage_dump.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_synth.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int]
        L.orc_forward.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_inverse.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_hilbert.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_decode_stage.argtypes = [C.c_void_p, C.c_size_t, C.c_long, C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.c_void_p]
        L.orc_reconstruct.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(Geom), C.c_int, C.c_int]
        L.orc_linearize.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Geom), C.c_int]
        _lib = L
    return _lib


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def geometry(W, H):
    g = Geom()
    lib().orc_geometry(C.byref(g), W, H, 8)
    return g


def synth(W, H, Cn, seed=0, kind=0):
    pix = np.empty((H, W, Cn), dtype=np.uint8)
    lib().orc_synth(pix.ctypes.data, W, H, Cn, seed, kind)
    return pix


def encode(pix, capacity=0):
    H, W, Cn = pix.shape
    pix = np.ascontiguousarray(pix, dtype=np.uint8)
    out, n, st = C.c_void_p(), C.c_size_t(), Stats()
    rc = lib().orc_encode(pix.ctypes.data, W, H, Cn, capacity, C.byref(out), C.byref(n), C.byref(st))
    if rc:
        raise ValueError("orc_encode rejected the image")
    data = C.string_at(out, n.value)
    _libc.free(out)
    return data, st


def encode_lin(lin, W, H, capacity=0):
    """The entropy stage alone on linearised coefficient planes int32 [C, W*H] -> (.dwt bytes, Stats)."""
    lin = np.ascontiguousarray(lin, dtype=np.int32)
    Cn = lin.shape[0]
    out, n, st = C.c_void_p(), C.c_size_t(), Stats()
    rc = lib().orc_encode_lin(lin.ctypes.data, W, H, Cn, capacity, C.byref(out), C.byref(n), C.byref(st))
    if rc:
        raise ValueError("orc_encode_lin rejected the planes")
    data = C.string_at(out, n.value)
    _libc.free(out)
    return data, st


def decode(data, pixels_max=-1):
    out, W, H, Cn = C.c_void_p(), C.c_int(), C.c_int(), C.c_int()
    buf = C.create_string_buffer(data, len(data))
    rc = lib().orc_decode(buf, len(data), pixels_max, C.byref(out), C.byref(W), C.byref(H), C.byref(Cn))
    if rc:
        return None
    arr = np.frombuffer(C.string_at(out, W.value * H.value * Cn.value), dtype=np.uint8)
    _libc.free(out)
    return arr.reshape(H.value, W.value, Cn.value).copy()


def decode_stage(data, W, H, Cn, pixels_max=-1):
    """-> (lin int32 [C, W*H], level, missing int32[48], planes) or None if the stream is unreadable."""
    lin = np.empty((Cn, W * H), dtype=np.int32)
    level = C.c_int()
    missing = np.zeros(48, dtype=np.int32)
    planes = (C.c_int * 3)()
    buf = C.create_string_buffer(data, len(data))
    rc = lib().orc_decode_stage(buf, len(data), pixels_max, lin.ctypes.data, C.byref(level), missing.ctypes.data, planes)
    if rc:
        return None
    return lin, level.value, missing, list(planes)[:Cn]


def stage_dump(pix):
    """-> (coef int32 [H,W,C] interleaved pyramid, lin int32 [C,H*W], planes list)."""
    H, W, Cn = pix.shape
    pix = np.ascontiguousarray(pix, dtype=np.uint8)
    coef = np.empty((H, W, Cn), dtype=np.int32)
    lin = np.empty((Cn, H * W), dtype=np.int32)
    planes = (C.c_int * 3)()
    rc = lib().orc_stage_dump(pix.ctypes.data, W, H, Cn, coef.ctypes.data, lin.ctypes.data, planes)
    assert rc == 0
    return coef, lin, list(planes)[:Cn]


def forward(img):
    """int32 [H,W,C] interleaved -> pyramid (copy)."""
    a = np.ascontiguousarray(img, dtype=np.int32).copy()
    H, W, Cn = a.shape
    lib().orc_forward(a.ctypes.data, W, H, Cn, 8)
    return a


def inverse(pyr):
    a = np.ascontiguousarray(pyr, dtype=np.int32).copy()
    H, W, Cn = a.shape
    lib().orc_inverse(a.ctypes.data, W, H, Cn, 8)
    return a


def linearize(pyr):
    """int32 [H,W,C] interleaved pyramid -> [C, H*W]."""
    H, W, Cn = pyr.shape
    g = geometry(W, H)
    pyr = np.ascontiguousarray(pyr, dtype=np.int32)
    lin = np.empty((Cn, H * W), dtype=np.int32)
    lib().orc_linearize(lin.ctypes.data, pyr.ctypes.data, C.byref(g), Cn)
    return lin


def reconstruct(lin, W, H, levels_out, missing=None):
    """[C, W*H] -> interleaved pyramid [h', w', C] of the first levels_out levels (decode.c:32-65)."""
    Cn = lin.shape[0]
    g = geometry(W, H)
    ow, oh = g.widths[levels_out], g.heights[levels_out]
    rows = [np.ascontiguousarray(lin[c], dtype=np.int32) for c in range(Cn)]
    ptrs = (C.c_void_p * Cn)(*[r.ctypes.data for r in rows])
    miss = np.zeros(48, dtype=np.int32) if missing is None else np.ascontiguousarray(missing, dtype=np.int32)
    out = np.empty((oh, ow, Cn), dtype=np.int32)
    lib().orc_reconstruct(out.ctypes.data, ptrs, miss.ctypes.data, C.byref(g), levels_out, Cn)
    return out


def hilbert(n, d):
    x, y = C.c_int(), C.c_int()
    lib().orc_hilbert(n, d, C.byref(x), C.byref(y))
    return x.value, y.value


def read_pnm(path):
    b = open(path, "rb").read()
    assert b[:2] in (b"P5", b"P6")
    Cn = 1 if b[:2] == b"P5" else 3
    pos, vals = 2, []
    while len(vals) < 3:
        while b[pos:pos + 1].isspace():
            pos += 1
        if b[pos:pos + 1] == b"#":
            pos = b.index(b"\n", pos) + 1
            continue
        end = pos
        while b[end:end + 1].isdigit():
            end += 1
        vals.append(int(b[pos:end]))
        pos = end
    pos += 1
    W, H, _ = vals
    return np.frombuffer(b[pos:pos + W * H * Cn], dtype=np.uint8).reshape(H, W, Cn).copy()


def write_pnm(path, pix):
    H, W, Cn = pix.shape
    with open(path, "wb") as f:
        f.write(b"P%d %d %d 255\n" % (5 if Cn == 1 else 6, W, H))
        f.write(np.ascontiguousarray(pix, dtype=np.uint8).tobytes())


def have_ref():
    return os.path.exists(os.path.join(REF_DIR, "encode")) and os.path.exists(os.path.join(REF_DIR, "decode"))


def many_plane_stream(W, H, Cn, planes, payload_seed=1, payload_len=600):
    """A .dwt whose preamble is well-formed — header, all-zero root images, plane counts `planes` (one per channel) —
    followed by seeded random bytes: plane counts above 16 are what only damage can produce (8-bit sources stay
    below 12).  decode.c:183-186 accepts any get_vli() value."""
    bits = []
    order = 0

    def vli(v):   # vli.h:67-84 in closed form
        nonlocal order
        top = (v + (1 << order)).bit_length() - 1
        bits.extend([0] * (top - order) + [1])
        rem = v + (1 << order) - (1 << top)
        bits.extend((rem >> i) & 1 for i in range(top))
        order = max(top - 2, 0)

    for _ in range(Cn):
        vli(0)            # encode.c:97-110: cnt = 0, no root values follow
    for p in planes[:Cn]:
        vli(p)
    body = bytearray(b"W" + (b"6" if Cn == 3 else b"5") + bytes([(W - 1) & 255, (W - 1) >> 8, (H - 1) & 255, (H - 1) >> 8]))
    acc = 0
    for i, b in enumerate(bits):
        acc |= b << (i & 7)
        if (i & 7) == 7:
            body.append(acc)
            acc = 0
    rng = np.random.default_rng(payload_seed)
    tail = rng.integers(0, 256, payload_len, dtype=np.uint8)
    if len(bits) & 7:   # the payload starts inside the last preamble byte
        body.append(acc | (int(tail[0]) << (len(bits) & 7)) & 255)
    return bytes(body) + tail.tobytes()
