"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/raytracer_amd.h declares; the host-side structs match the C layouts; no compute call can
silently succeed without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

import raytracer_amd as ra
from raytracer_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "raytracer_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pm_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported():
    lib = ra.load_library()
    syms = declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/raytracer_amd.h but not exported"
    assert set(capi.EXPORTS) == set(syms)


def test_struct_layouts():
    assert ra.prim_dtype.itemsize == 144 and ra.photon_dtype.itemsize == 32
    assert ctypes.sizeof(ra.RenderParams) == 80
    assert ctypes.sizeof(ra.Stats) == 136
    assert ra.photon_dtype.fields["power"][1] == 12 and ra.photon_dtype.fields["dx"][1] == 16 and ra.photon_dtype.fields["bounce"][1] == 28


def test_default_scene_matches_reference_scene():
    from .conftest import golden
    assert ra.default_scene().tobytes() == golden("scene_default.bin", ra.prim_dtype).tobytes()


def test_no_silent_cpu_fallback():
    """Without a gfx950 device pm_ctx_create must fail loudly (this test runs where no GPU is visible)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    with pytest.raises(ra.PMError):
        ra.Context(0)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "raytracer_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "orc_py" not in text and "pm_oracle" not in text and "liboracle" not in text, os.path.join(dirpath, f)
