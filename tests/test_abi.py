"""CPU tests of the drop-in boundary: libdwtx.so loads and exports every symbol
include/dwtx.h declares; host-side geometry matches the oracle.  No compute calls."""
import os
import re

import pytest

import orc

ROOT = orc.ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dwtx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dwtx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from dwt_amd import _lib

    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"{n} declared in dwtx.h but not exported"
        assert n in _lib.SYMBOLS, f"{n} not typed in dwt_amd/_lib.py"
    assert set(_lib.SYMBOLS) <= set(names)


@pytest.mark.parametrize("wh", [(8, 8), (320, 240), (1920, 1080), (4096, 4096), (16384, 16384), (131, 77),
                                (17, 300), (65536, 9), (9, 65536), (32768, 9), (9, 32768), (32768, 32768), (20001, 9)])
def test_compute_lengths_matches_oracle(wh):
    import dwt_amd

    W, H = wh
    levels, lengths, pixels, widths, heights = dwt_amd.compute_lengths(W, H, 8)
    g = orc.geometry(W, H)
    assert levels == g.levels
    assert lengths == list(g.lengths[: levels + 1])
    assert widths == list(g.widths[: levels + 1])
    assert heights == list(g.heights[: levels + 1])
    assert pixels == list(g.pixels[: levels + 1])


def test_no_gpu_fails_loudly():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import dwt_amd

    with pytest.raises(RuntimeError):
        dwt_amd.Context(0)
