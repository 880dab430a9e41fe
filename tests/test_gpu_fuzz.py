"""A short run of the randomised parity sweep (tools/fuzz_parity.py): random frame sizes, photon counts, seeds, cameras,
light colours and flags on three scenes, panes and photon arrays bit for bit against the oracle.  (Round 1 ran 4 300
such cases without a mismatch; 60 of them are kept in the suite.)"""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_randomised_parity_sweep():
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "tools", "fuzz_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.sweep(60, 2026, verbose=False) == []


def test_randomised_knn_and_gather_sweep():
    """tools/fuzz_knn.py: random maps of both kinds (1 .. 400 000 photons), query points on photons, near them, in mid air and
    far outside the scene, arbitrary and axis-aligned normals: pm_knn_gather and pm_knn_query against the oracle."""
    spec = importlib.util.spec_from_file_location("fuzz_knn", os.path.join(ROOT, "tools", "fuzz_knn.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.sweep(40, 99, verbose=False) == []


def test_randomised_shard_sweep():
    """tools/fuzz_shards.py: 1 .. 8 ranks' shards cut by the frame's own rule, assembled on the device, and ranges cut at
    random through pm_emit_range, against the map one rank emits alone (three scenes, both kinds)."""
    spec = importlib.util.spec_from_file_location("fuzz_shards", os.path.join(ROOT, "tools", "fuzz_shards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.sweep(30, 5, verbose=False) == []
