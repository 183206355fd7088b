"""ctypes loader for libdwtx.so (the C ABI declared in include/dwtx.h).

The library is hand-written HIP for gfx950; there is no CPU or PyTorch
fallback.  If the shared object is missing this module raises at import of the
symbol table, and every compute entry point fails loudly without a GPU.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdwtx.so")

MAX_LEVELS = 16


class Geom(C.Structure):
    _fields_ = [
        ("levels", C.c_int),
        ("widths", C.c_int * MAX_LEVELS),
        ("heights", C.c_int * MAX_LEVELS),
        ("pixels", C.c_int * MAX_LEVELS),
        ("lengths", C.c_int * MAX_LEVELS),
    ]


class StreamInfo(C.Structure):
    _fields_ = [
        ("planes", C.c_int * 3), ("pmax", C.c_int), ("segments", C.c_int), ("entries", C.c_int),
        ("tokens", C.c_uint), ("order0", C.c_int), ("hdr_bits", C.c_uint), ("root_bits", C.c_uint), ("meta_bits", C.c_uint), ("segments_cut", C.c_uint),
        ("total_bits", C.c_ulonglong), ("nbytes", C.c_ulonglong), ("error", C.c_int), ("exact_orders", C.c_int),
    ]


class DecodeInfo(C.Structure):
    _fields_ = [
        ("status", C.c_int), ("W", C.c_int), ("H", C.c_int), ("C", C.c_int), ("levels", C.c_int),
        ("planes", C.c_int * 3), ("pmax", C.c_int), ("level", C.c_int), ("nsegs", C.c_int),
        ("truncated", C.c_int), ("missing", C.c_int * 48), ("bits_used", C.c_ulonglong),
        ("hops", C.c_uint), ("hopped_chunks", C.c_uint), ("walked_tokens", C.c_uint), ("zeros_left", C.c_uint),
    ]


class SegIndex(C.Structure):
    _fields_ = [("bit", C.c_ulonglong), ("sym_base", C.c_ulonglong), ("n1", C.c_uint), ("cnt", C.c_uint),
                ("desc", C.c_uint), ("order", C.c_uint)]


INDEX_MAGIC = 0x49545744
INDEX_MAX_SEGS = 768


class Index(C.Structure):
    """dwtx_index: the sidecar index of one stream (include/dwtx.h)."""
    _fields_ = [("magic", C.c_uint), ("W", C.c_int), ("H", C.c_int), ("C", C.c_int), ("nsegs", C.c_int),
                ("reserved", C.c_int), ("stream_bits", C.c_ulonglong), ("seg", SegIndex * INDEX_MAX_SEGS)]


class Stats(C.Structure):
    _fields_ = [
        ("meta_bits", C.c_int),
        ("root_bits", C.c_int),
        ("total_bits", C.c_int),
        ("kib", C.c_int),
        ("levels", C.c_int),
        ("planes", C.c_int * 3),
    ]


# enum dwtx_option (include/dwtx.h): diagnostic switches of a context
OPTIONS = {name: i for i, name in enumerate((
    "exact_orders", "no_square_tiles", "part_images", "one_stream", "decode_parts", "two_families", "no_second_walk",
    "no_index", "no_index_fallback", "no_capacity_cut", "no_fine16", "no_fused_levels"))}

# name -> (restype, argtypes); must list every symbol include/dwtx.h declares
_vp, _i, _sz = C.c_void_p, C.c_int, C.c_size_t
SYMBOLS = {
    "dwtx_ctx_create": (_i, [_i, C.POINTER(_vp)]),
    "dwtx_ctx_create_on_stream": (_i, [_i, _vp, C.POINTER(_vp)]),
    "dwtx_ctx_destroy": (None, [_vp]),
    "dwtx_last_error": (C.c_char_p, []),
    "dwtx_sync": (_i, [_vp]),
    "dwtx_ctx_set_index": (_i, [_vp, _vp, _vp]),
    "dwtx_stream": (_vp, [_vp]),
    "dwtx_ctx_set_option": (_i, [_vp, _i, C.c_long]),
    "dwtx_ctx_get_option": (C.c_long, [_vp, _i]),
    "dwtx_malloc": (_vp, [_vp, _sz]),
    "dwtx_free": (None, [_vp, _vp]),
    "dwtx_host_alloc": (_vp, [_vp, _sz]),
    "dwtx_host_free": (None, [_vp, _vp]),
    "dwtx_upload": (_i, [_vp, _vp, _vp, _sz]),
    "dwtx_download": (_i, [_vp, _vp, _vp, _sz]),
    "dwtx_compute_lengths": (_i, [C.POINTER(_i)] * 4 + [_i, _i, _i]),
    "dwtx_geometry": (_i, [C.POINTER(Geom), _i, _i]),
    "dwtx_synth_pixels": (_i, [_vp, _vp, _i, _i, _i, _i, C.c_uint, _i]),
    "dwtx_planes_from_pixels": (_i, [_vp, _vp, _vp, _i, _i, _i, _i]),
    "dwtx_pixels_from_planes": (_i, [_vp, _vp, _vp, _i, _i, _i, _i]),
    "dwtx_transformation_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i]),
    "dwtx_transformation_inv": (_i, [_vp, _vp, _vp, _i, _i, _i]),
    "dwtx_transformation_fwd_pixels": (_i, [_vp, _vp, _vp, C.POINTER(C.c_uint), _vp, _i, _i, _i, _i]),
    "dwtx_transformation_inv_pixels": (_i, [_vp, _vp, _vp, _vp, C.c_uint, _i, _i, _i, _i]),
    "dwtx_pack_streams": (_i, [_vp, _vp, C.c_size_t, _vp, _vp, C.c_size_t, _vp, _i]),
    "dwtx_linearization": (_i, [_vp, _vp, _vp, _i, _i, _i]),
    "dwtx_reconstruction": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i]),
    "dwtx_encode_bound": (_sz, [_i, _i, _i]),
    "dwtx_encode_device": (_i, [_vp, _vp, _i, _i, _i, _i, C.c_long, _vp, _sz, _vp]),
    "dwtx_decode_device": (_i, [_vp, _vp, _sz, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "dwtx_encode_images": (_i, [_vp, _vp, _i, _i, _i, _i, C.c_long, _vp, _sz, _vp, _vp]),
    "dwtx_decode_images": (_i, [_vp, _vp, _sz, _vp, _i, _i, _vp, _sz, _vp, _vp, _vp]),
    "dwtx_decode_images_info": (_i, [_vp, _vp, _sz, _vp, _i, _i, _vp, _sz, _vp, _vp, _vp, _vp]),
    "dwtx_decode_planes": (_i, [_vp, _vp, _vp, _sz, _vp, _i, _i, _i, _i, _i, _vp]),
    "dwtx_encode_planes": (_i, [_vp, _vp, _i, _i, _i, _i, C.c_long, _vp, _sz, _vp]),
}

_lib = None


def load():
    """Load libdwtx.so once and type its symbols.  Raises OSError if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError(
                f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C dwt_amd/csrc` (there is no fallback path)")
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib
