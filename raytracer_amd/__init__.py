"""raytracer_amd -- MI355X (gfx950) HIP backend for the photon-mapping hot path of Deamon5550/raytracer.

Only what the hot path needs lives here: ``csrc/`` (HIP kernels + the C-ABI of
``libraytracer_amd.so``), ``capi`` (ctypes binding of that C-ABI) and ``frame`` (host-side frame
orchestration incl. the multi-GPU split).  There is no CPU fallback: importing works anywhere,
but every compute call needs the built library and a gfx950 device and fails loudly otherwise.
"""
from .capi import (  # noqa: F401
    PM_MAP_GLOBAL, PM_MAP_CAUSTIC, PM_FLAG_NO_GLOBAL, PM_FLAG_NO_CAUSTIC, PM_FLAG_NO_DIRECT, PM_FLAG_COUNT_TESTS,
    PMError, Context, PhotonMap, RenderParams, Stats, default_scene, prim_dtype, photon_dtype, load_library,
)
from .scenes import many_spheres_scene, caustics_scene  # noqa: F401,E402
