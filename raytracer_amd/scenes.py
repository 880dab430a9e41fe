"""Synthetic scenes of BASELINE configs 4 and 5 as pm_prim arrays.

Neither scene exists in the reference (its only scene is the Cornell box of src/Image.cpp:20-28, ``default_scene``);
both are built from the reference's own primitives -- ``PlaneObject`` walls and ``SphereObject`` -- with the material
argument order of its constructors (col, d, s, t, a).  The random numbers come from a counter-based stream (mix64 of
seed and index, the same mixer as the RNG contract of the render path), so that any host language reproduces the
scene from the seed alone.
"""
import numpy as np

from .capi import default_scene, prim_dtype

_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _mix64(z):
    z = z.astype(np.uint64)
    z ^= z >> np.uint64(30); z *= np.uint64(0xBF58476D1CE4E5B9)
    z ^= z >> np.uint64(27); z *= np.uint64(0x94D049BB133111EB)
    z ^= z >> np.uint64(31)
    return z


def uniform_stream(seed, count):
    """count doubles in [0,1): u_j = (mix64(mix64(seed) + (j+1)*GOLD) >> 11) * 2^-53."""
    with np.errstate(over="ignore"):
        base = _mix64(np.array([seed], dtype=np.uint64))[0]
        j = np.arange(1, count + 1, dtype=np.uint64)
        return (_mix64(base + j * _GOLD) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def many_spheres_scene(n_spheres=10000, seed=2024):
    """BASELINE config 5: the five Cornell walls + n diffuse spheres (d = 0.4, a = 1), centres uniform in
    x, y in [-4.5, 4.5], z in [0.5, 9.5], radius uniform in [0.02, 0.08], colour bytes uniform in 64..255."""
    s = np.zeros(5 + n_spheres, dtype=prim_dtype)
    s[:5] = default_scene()[:5]
    u = uniform_stream(seed, 7 * n_spheres).reshape(n_spheres, 7)
    sp = s[5:]
    sp["kind"] = 0
    sp["x"] = u[:, 0] * 9.0 - 4.5
    sp["y"] = u[:, 1] * 9.0 - 4.5
    sp["z"] = u[:, 2] * 9.0 + 0.5
    sp["radius"] = 0.02 + u[:, 3] * 0.06
    for k, name in enumerate(("red", "green", "blue")):
        byte = (64 + np.floor(u[:, 4 + k] * 192.0)).astype(np.float32)
        sp[name] = byte / np.float32(255.0)              # SceneObject's ctor: byte / 255.0f, src/Scene.cpp:54-56
    sp["diffuse_chance"], sp["absorb_chance"] = 0.4, 1.0
    return s


# centre x, y, z and radius of the five glass spheres of the caustics scene
GLASS_SPHERES = ((2.0, -3.5, 3.0, 1.5), (-2.0, -3.5, 5.0, 1.5), (0.0, -4.0, 7.5, 1.0), (-3.0, -1.0, 8.0, 0.8), (2.5, 0.5, 6.5, 1.0))


def caustics_scene():
    """BASELINE config 4: the five Cornell walls + five glass spheres with the material of the reference's glass
    sphere (d = 0, s = 0.1, t = 0.9, a = 0, refraction 2.5; src/Image.cpp:27): every sphere hit spawns a refracted and
    a reflected ray, and most of the caustic map's photons have passed through glass."""
    s = np.zeros(5 + len(GLASS_SPHERES), dtype=prim_dtype)
    s[:5] = default_scene()[:5]
    glass = default_scene()[5]
    for i, (x, y, z, r) in enumerate(GLASS_SPHERES):
        s[5 + i] = glass
        s[5 + i]["x"], s[5 + i]["y"], s[5 + i]["z"], s[5 + i]["radius"] = x, y, z, r
    return s
