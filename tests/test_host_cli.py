"""The reference's command line (src/Main.cpp:14-41) and image front-end (src/Image.cpp:14-76) on the
GPU backend: argument checks (CPU), and the PNG it writes against the oracle's pane (GPU)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "raytracer_amd", "host", "raytracer")


def run(args, cwd, env=None):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([EXE] + [str(a) for a in args], cwd=cwd, env=e, capture_output=True, text=True, timeout=300)


@pytest.mark.skipif(not os.path.exists(EXE), reason="host CLI not built (make -C raytracer_amd/host)")
def test_cli_argument_checks(tmp_path):
    """Same messages and exit code 0 as the reference; none of these paths touches the GPU."""
    r = run([1, 2, 3], tmp_path)
    assert r.returncode == 0 and r.stdout == "Usage: ./raytracer [# cores] [width] [height] [samples]\n"
    r = run([8, 1440, 1440, 1], tmp_path)     # Main refuses non-16:9 sizes
    assert r.returncode == 0 and r.stdout == "Image dimensions must be a 16:9 ratio.\n"
    r = run([8, 1280, 720, -1], tmp_path)
    assert r.returncode == 0 and r.stdout == "Dimensions and samples must be positive\n"
    # options the reference lacks: a dangling option is a usage error, a well-formed one does not disturb the checks
    r = run(["--seed"], tmp_path)
    assert r.returncode == 0 and r.stdout.startswith("Usage:")
    r = run(["--photons", 10, "--out", "x.png", 8, 1440, 1440, 1], tmp_path)
    assert r.returncode == 0 and r.stdout == "Image dimensions must be a 16:9 ratio.\n"
    assert not os.path.exists(tmp_path / "raytraced.png") and not os.path.exists(tmp_path / "x.png")
    # an option of ours that cannot work is an error a launcher can see: stderr, non-zero, before the GPU is touched
    for bad in ("spheres:-3", "spheres:99999999", "spheres:x"):
        r = run(["--scene", bad, 8, 1280, 720, 1], tmp_path)
        assert r.returncode == 2 and r.stdout == "" and "spheres:N" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("samples", [1, 2])
def test_cli_image_matches_oracle(tmp_path, orc, samples):
    from PIL import Image
    w, h, n, seed = 160, 90, 2048, 4711
    r = run([8, w, h, samples], tmp_path, {"RAYTRACER_SEED": str(seed), "RAYTRACER_PHOTONS": str(n)})
    assert r.returncode == 0, r.stderr
    assert "Rendering scene" in r.stdout and "Scene rendered in" in r.stdout
    img = np.array(Image.open(tmp_path / "raytraced.png"))
    assert img.shape == (h, w, 4)
    scene = orc.default_scene()
    g, c = orc.Map(orc.emit(scene, 0, seed, n)[0]), orc.Map(orc.emit(scene, 1, seed, n)[0])
    pane, _ = orc.render(scene, g, c, orc.params(w * samples, h * samples, seed), nthreads=4)
    # s x s box filter with integer division, vertical flip (src/Image.cpp:36-57)
    ch = [((pane >> sh) & 0xFF).astype(np.int64).reshape(h, samples, w, samples).sum(axis=(1, 3)) // (samples * samples) for sh in (16, 8, 0)]
    red, green, blue = [x[::-1] for x in ch]
    # file bytes are the little-endian 0xAARRGGBB words: B, G, R, A (red and blue swapped in the file)
    assert np.array_equal(img[..., 0], blue) and np.array_equal(img[..., 1], green) and np.array_equal(img[..., 2], red)
    assert (img[..., 3] == 255).all()
    g.close(); c.close()
