// pm_device.h -- device-side building blocks shared by the gfx950 kernels:
// RNG contract v1, IEEE-exact FP64 vector helpers, primitive intersection.
// Everything is FP64 with contraction off (-ffp-contract=off): primary hit ids and
// the direct-lighting pass must match the CPU path bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pm {

// ---------------------------------------------------------------- RNG contract v1
// replaces randutil::nextDouble (reference src/Random.cpp:25-27); see DESIGN.md
constexpr uint64_t GOLD = 0x9E3779B97F4A7C15ULL;
constexpr uint64_t ODD1 = 0xD1B54A32D192ED03ULL;
constexpr uint64_t ODD2 = 0x8CB92BA72F3D8DD7ULL;
enum : uint32_t { KIND_PIXEL = 1, KIND_EMIT_GLOBAL = 2, KIND_EMIT_CAUSTIC = 3 };

__host__ __device__ inline uint64_t mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27; z *= 0x94D049BB133111EBULL;
    z ^= z >> 31; return z;
}
__host__ __device__ inline uint64_t stream_key(uint64_t seed, uint32_t kind) { return mix64(seed + GOLD * (uint64_t)kind); }
__host__ __device__ inline uint64_t item_key(uint64_t K, uint64_t item) { return mix64(K ^ (item * ODD1 + ODD2)); }
__host__ __device__ inline double draw(uint64_t IK, uint32_t j) {
    return (double)(mix64(IK + ((uint64_t)j + 1ULL) * GOLD) >> 11) * (1.0 / 9007199254740992.0);
}

// ---------------------------------------------------------------- scalar helpers
// fastfloor, reference src/Vector.cpp:13-16; x86 cvttsd2si yields INT_MIN on NaN / overflow
__device__ inline int32_t fastfloor(double x) {
    int32_t xi = (x > -2147483649.0 && x < 2147483648.0) ? (int32_t)x : INT32_MIN;
    return x < (double)xi ? xi - 1 : xi;
}
__device__ inline int32_t imin(int32_t a, int32_t b) { return a < b ? a : b; }  // src/Vector.cpp:18-20

struct V3 { double x, y, z; };
// Vec3::normalize, src/Vector.cpp:95-100: three divisions by the length
__device__ inline void normalize(V3 &v) {
    double len = sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
    v.x /= len; v.y /= len; v.z /= len;
}

// Deterministic sin/cos for psi in [0, 2*pi): quadrant reduction + fdlibm-style polynomial
// kernels in a fixed operation order (no FMA), so that CPU and GPU agree bit for bit.  The
// reference calls libm (src/PhotonMap.cpp:299,412); the difference is bounded in tests.
__device__ inline void det_sincos(double x, double &sn, double &cs) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double INVPIO2 = 6.36619772367581382433e-01;
    const double PIO2_1 = 1.57079632673412561417e+00;
    const double PIO2_1T = 6.07710050650619224932e-11;
    int q = (int)(x * INVPIO2 + 0.5);
    double fq = (double)q;
    double r = (x - fq * PIO2_1) - fq * PIO2_1T;
    double z = r * r;
    double ps = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    double s = r + (z * r) * (S1 + z * ps);
    double pc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    double c = 1.0 - (0.5 * z - z * pc);
    switch (q & 3) {
    case 0: sn = s; cs = c; break;
    case 1: sn = c; cs = -s; break;
    case 2: sn = -s; cs = -c; break;
    default: sn = -c; cs = s; break;
    }
}

// ---------------------------------------------------------------- scene (SoA)
// Geometry and material of all primitives as structure-of-arrays.  Kernels stage the arrays
// into LDS when they fit (the per-lane loop then reads wave-uniform addresses = LDS broadcast)
// and read the same layout from global memory otherwise.
enum { G_X = 0, G_Y, G_Z, G_RADIUS, G_DX, G_DY, G_DZ, G_MIN, G_MAX, G_ABSORB, G_DIFFUSE, G_SPECULAR,
       G_TRANSMISSION, G_REFRACTION, G_NFIELDS };
enum { PK_SPHERE = 0, PK_PLANE_X = 1, PK_PLANE_Y = 2, PK_PLANE_Z = 3, PK_NEVER = 4 };

struct SceneView {
    int n;            // primitives
    int stride;       // elements per field array
    const double *g;  // [G_NFIELDS][stride]
    const float *rgb; // [3][stride]
    const int *kind;  // [stride]  PK_*
    __device__ double f(int field, int i) const { return g[field * stride + i]; }
};
// bytes of one staged scene with `stride` slots
__host__ __device__ inline size_t scene_bytes(int stride) {
    return (size_t)stride * (G_NFIELDS * sizeof(double) + 3 * sizeof(float) + sizeof(int));
}

// SphereObject::intersect, src/Scene.cpp:87-116
__device__ inline bool sphere_intersect(const SceneView &s, int i, const V3 &o, const V3 &d, double dt, V3 &res, V3 &nrm) {
    double x0 = s.f(G_X, i) + dt * s.f(G_DX, i);
    double y0 = s.f(G_Y, i) + dt * s.f(G_DY, i);
    double z0 = s.f(G_Z, i) + dt * s.f(G_DZ, i);
    double lx = x0 - o.x, ly = y0 - o.y, lz = z0 - o.z;
    double b = d.x * lx + d.y * ly + d.z * lz;
    if (b < 0) return false;
    double d2 = (lx * lx + ly * ly + lz * lz) - b * b;
    double radius = s.f(G_RADIUS, i);
    if (d2 > radius * radius) return false;
    double t = sqrt(radius * radius - d2);
    if (t < 0) t = b + t; else t = b - t;
    if (t < 0) return false;
    res.x = d.x * t + o.x; res.y = d.y * t + o.y; res.z = d.z * t + o.z;
    nrm.x = res.x - x0; nrm.y = res.y - y0; nrm.z = res.z - z0;
    normalize(nrm);
    return true;
}
// PlaneObject::intersect, src/Scene.cpp:135-193 (IEEE inf/NaN behaviour is load-bearing)
__device__ inline bool plane_intersect(const SceneView &s, int kind, int i, const V3 &c, const V3 &r, V3 &res, V3 &nrm) {
    double mn = s.f(G_MIN, i), mx = s.f(G_MAX, i);
    if (kind == PK_PLANE_X) {
        double px = s.f(G_X, i);
        double dx = px - c.x;
        double mul = dx / r.x;
        if (mul < 0) return false;
        double hz = c.z + mul * r.z;
        if (hz < -0.01) return false;
        double hy = c.y + mul * r.y;
        if (hy > mx || hy < mn) return false;
        res.x = px; res.y = hy; res.z = hz;
        nrm.x = px < 0 ? 1 : -1; nrm.y = 0; nrm.z = 0;
        return true;
    } else if (kind == PK_PLANE_Y) {
        double py = s.f(G_Y, i);
        double dy = py - c.y;
        double mul = dy / r.y;
        if (mul < 0) return false;
        double hz = c.z + mul * r.z;
        if (hz < -0.01) return false;
        double hx = c.x + mul * r.x;
        if (hx < mn || hx > mx) return false;
        res.x = hx; res.y = py; res.z = hz;
        nrm.x = 0; nrm.y = py < 0 ? 1 : -1; nrm.z = 0;
        return true;
    } else if (kind == PK_PLANE_Z) {
        double pz = s.f(G_Z, i);
        double dz = pz - c.z;
        double mul = dz / r.z;
        if (mul < 0) return false;
        double hx = c.x + mul * r.x;
        if (hx < -5 || hx > 5) return false;
        double hy = c.y + mul * r.y;
        if (hy < mn || hy > mx) return false;
        res.x = hx; res.y = hy; res.z = pz;
        nrm.x = 0; nrm.y = 0; nrm.z = pz < 0 ? 1 : -1;
        return true;
    }
    return false;
}
__device__ inline bool prim_intersect(const SceneView &s, int i, const V3 &o, const V3 &d, double dt, V3 &res, V3 &nrm) {
    int kind = s.kind[i];
    if (kind == PK_SPHERE) return sphere_intersect(s, i, o, d, dt, res, nrm);
    return plane_intersect(s, kind, i, o, d, res, nrm);
}
// Scene::intersect, src/Scene.cpp:25-47: closest hit by squared distance of the reconstructed
// hit point, strict '<' so the lowest index wins ties; returns object id or -1
__device__ inline int scene_intersect(const SceneView &s, const V3 &src, const V3 &dir, int exclude, double dt, V3 &fres, V3 &fnrm) {
    double nearest = 1024 * 1024;
    int nearest_obj = -1;
    V3 res = {0, 0, 0}, nrm = {0, 0, 0};
    for (int i = 0; i < s.n; i++) {
        if (i == exclude) continue;
        if (prim_intersect(s, i, src, dir, dt, res, nrm)) {
            double dist = (res.x - src.x) * (res.x - src.x);
            dist += (res.y - src.y) * (res.y - src.y);
            dist += (res.z - src.z) * (res.z - src.z);
            if (dist < nearest) { nearest = dist; nearest_obj = i; fres = res; fnrm = nrm; }
        }
    }
    return nearest_obj;
}

// Stage the SoA scene into LDS (all threads of the block call this, then __syncthreads()).
__device__ inline SceneView stage_scene(const SceneView &gs, unsigned char *lds) {
    size_t bytes = scene_bytes(gs.stride);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(gs.g);  // g, rgb, kind are one contiguous block
    uint32_t *dst = reinterpret_cast<uint32_t *>(lds);
    for (size_t w = threadIdx.x + (size_t)threadIdx.y * blockDim.x; w < bytes / 4; w += (size_t)blockDim.x * blockDim.y) dst[w] = src[w];
    SceneView v;
    v.n = gs.n; v.stride = gs.stride;
    v.g = reinterpret_cast<const double *>(lds);
    v.rgb = reinterpret_cast<const float *>(lds + (size_t)gs.stride * G_NFIELDS * sizeof(double));
    v.kind = reinterpret_cast<const int *>(lds + (size_t)gs.stride * (G_NFIELDS * sizeof(double) + 3 * sizeof(float)));
    return v;
}

// primary ray, loop body of render_task, src/Raytrace.cpp:284-293
__device__ inline V3 primary_ray(int width, int height, const V3 &cam, uint64_t K, uint64_t pixel, int x, int y) {
    uint64_t IK0 = item_key(K, pixel * 32u);
    double fov = (width / 1280.0) * 64.0;
    double x1 = draw(IK0, 0) * 0.6 - 0.3;
    double y1 = draw(IK0, 1) * 0.6 - 0.3;
    double x0 = (x - width / 2 + x1) / fov - cam.x;
    double y0 = (y - height / 2 + y1) / fov - cam.y;
    V3 r = {x0, y0, -cam.z};
    normalize(r);
    return r;
}

}  // namespace pm
