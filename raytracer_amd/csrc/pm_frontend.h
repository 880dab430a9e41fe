// pm_frontend.h -- the two small kernels either side of the frame pipeline:
//   k_intersect_rays  one lane per caller-given ray: SceneObject::intersect (one primitive, reference src/Scene.cpp:87-193)
//                     or Scene::intersect (closest hit over the scene, src/Scene.cpp:25-47) through the very device
//                     functions the frame kernels use (prim_hit / prim_normal / scene_intersect): the boundary's
//                     intersect calls and the parity gate of the primitives on gfx950
//   k_downsample      the image front-end of raytrace::render (src/Image.cpp:36-57): s x s box filter with integer
//                     division per channel, vertical flip, words left as 0xAARRGGBB
#pragma once
#include "pm_device.h"

namespace pm {

// prim[i] >= 0: test that primitive alone; hit[i] = 1 / 0, res / nrm written on a hit only (the reference leaves its
// outputs alone on a miss).  prim[i] < 0 (or prim == NULL): closest hit over the scene, skipping exclude[i] (exclude ==
// NULL: nothing skipped); hit[i] = object id or -1.
template <int MODE>
__global__ __launch_bounds__(256) void k_intersect_rays(SceneView gs, long long n, const double *src, const double *dir, const double *dt, const int32_t *prim,
                                                         const int32_t *exclude, int32_t *hit, double *res, double *nrm) {
    extern __shared__ __align__(16) unsigned char smem[];
    SceneView sv = gs;
    if (MODE == SCENE_LDS) { sv = stage_scene(gs, smem); __syncthreads(); }
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const V3 o = {src[3 * i], src[3 * i + 1], src[3 * i + 2]}, d = {dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]};
    const double t = dt ? dt[i] : 0.0;
    const int p = prim ? prim[i] : -1;
    V3 r = {0, 0, 0}, nn = {0, 0, 0};
    int h;
    if (p >= 0) {
        h = prim_hit(sv, p, o, d, t, r) ? 1 : 0;
        if (h) prim_normal(sv, p, t, r, nn);
    } else {
        // The sphere grid narrows the scan down to the spheres along the ray, which is the reference's scan for the unit
        // directions every ray of the frame has.  A caller's direction of another length makes SphereObject::intersect
        // "hit" spheres the ray never comes near (b = d . l is scaled, d2 = l . l - b^2 is not a distance any more): such
        // rays take the scan itself.
        const double dd = d.x * d.x + d.y * d.y + d.z * d.z;
        if (MODE == SCENE_GRID && !(fabs(dd - 1.0) <= 1e-12)) h = scene_intersect<SCENE_GLOBAL>(sv, o, d, exclude ? exclude[i] : -1, t, r, nn);
        else h = scene_intersect<MODE>(sv, o, d, exclude ? exclude[i] : -1, t, r, nn);
    }
    hit[i] = h;
    if (p >= 0 ? h != 0 : h >= 0) {
        res[3 * i] = r.x; res[3 * i + 1] = r.y; res[3 * i + 2] = r.z;
        nrm[3 * i] = nn.x; nrm[3 * i + 1] = nn.y; nrm[3 * i + 2] = nn.z;
    }
}

// out is w x h, pane is (w*s) x (h*s).  One lane per output pixel, x fastest: a wave reads s consecutive runs of 64*s
// words per pane row (coalesced) and writes one run of 64.  Sums are exact in 32 bits for s <= 4096.
__global__ __launch_bounds__(256) void k_downsample(const uint32_t *pane, int w, int h, int s, int flip, uint32_t *out) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    uint32_t red = 0, green = 0, blue = 0;
    const size_t pw = (size_t)w * s;
    for (int y0 = 0; y0 < s; y0++) {
        const uint32_t *row = pane + ((size_t)y * s + y0) * pw + (size_t)x * s;
        for (int x0 = 0; x0 < s; x0++) {
            const uint32_t v = row[x0];
            red += (v >> 16) & 0xFFu; green += (v >> 8) & 0xFFu; blue += v & 0xFFu;
        }
    }
    const uint32_t ss = (uint32_t)s * (uint32_t)s;
    red /= ss; green /= ss; blue /= ss;
    out[(size_t)x + (size_t)(flip ? h - y - 1 : y) * w] = (0xFFu << 24) | (red << 16) | (green << 8) | blue;
}

}  // namespace pm
