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

// Uniform grid over the spheres of a large scene (built on the host at pm_scene_upload).  It only narrows down WHICH
// primitives a ray is tested against; the tests themselves and the choice among the hits are those of the linear scan.
struct SceneGrid {
    int nloose;                  // primitives tested one by one: planes, spheres too large for the grid or not finite
    int rx, ry, rz;              // cells per axis
    const int *loose;            // ascending primitive ids
    const uint32_t *cell_start;  // [rx*ry*rz + 1] into refs
    const int *refs;             // sphere ids of every cell whose padded box (swept over dt in [0,1]) touches the cell
    const double *packed;        // [refs][8]: x y z radius dx dy dz of the sphere of each ref, and its id (as an integer in the
                                 // last slot): one 64-byte line per test instead of seven lines of the field arrays
    double lo[3], hi[3], cs[3], inv_cs[3];
};
enum { SCENE_LDS = 0, SCENE_GLOBAL = 1, SCENE_GRID = 2 };   // how a kernel reads the scene

struct SceneView {
    int n;            // primitives
    int stride;       // elements per field array
    const double *g;  // [G_NFIELDS][stride]
    const float *rgb; // [3][stride]
    const int *kind;  // [stride]  PK_*
    SceneGrid grid;   // SCENE_GRID only
    __device__ double f(int field, int i) const { return g[field * stride + i]; }
};
// bytes of one staged scene with `stride` slots
__host__ __device__ inline size_t scene_bytes(int stride) {
    return (size_t)stride * (G_NFIELDS * sizeof(double) + 3 * sizeof(float) + sizeof(int));
}

// SphereObject::intersect, src/Scene.cpp:87-116, in two parts: the hit point, and the normal at it.  The reference
// computes the normal of every hit; only the normal of the hit that wins is ever read, so it is computed once, for
// the winner, from the same operands (same bits).
__device__ inline bool sphere_hit(const SceneView &s, int i, const V3 &o, const V3 &d, double dt, V3 &res) {
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
    return true;
}
// the same test on a packed record (SceneGrid::packed): same operands, same operations
__device__ inline bool sphere_hit_packed(const double *P, const V3 &o, const V3 &d, double dt, V3 &res) {
    const double4 a = *reinterpret_cast<const double4 *>(P), b4 = *reinterpret_cast<const double4 *>(P + 4);   // x y z radius | dx dy dz id
    double x0 = a.x + dt * b4.x;
    double y0 = a.y + dt * b4.y;
    double z0 = a.z + dt * b4.z;
    double lx = x0 - o.x, ly = y0 - o.y, lz = z0 - o.z;
    double b = d.x * lx + d.y * ly + d.z * lz;
    if (b < 0) return false;
    double d2 = (lx * lx + ly * ly + lz * lz) - b * b;
    double radius = a.w;
    if (d2 > radius * radius) return false;
    double t = sqrt(radius * radius - d2);
    if (t < 0) t = b + t; else t = b - t;
    if (t < 0) return false;
    res.x = d.x * t + o.x; res.y = d.y * t + o.y; res.z = d.z * t + o.z;
    return true;
}
__device__ inline void sphere_normal(const SceneView &s, int i, double dt, const V3 &res, V3 &nrm) {
    double x0 = s.f(G_X, i) + dt * s.f(G_DX, i);
    double y0 = s.f(G_Y, i) + dt * s.f(G_DY, i);
    double z0 = s.f(G_Z, i) + dt * s.f(G_DZ, i);
    nrm.x = res.x - x0; nrm.y = res.y - y0; nrm.z = res.z - z0;
    normalize(nrm);
}
// PlaneObject::intersect, src/Scene.cpp:135-193 (IEEE inf/NaN behaviour is load-bearing)
__device__ inline bool plane_hit(const SceneView &s, int kind, int i, const V3 &c, const V3 &r, V3 &res) {
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
        return true;
    }
    return false;
}
__device__ inline void plane_normal(const SceneView &s, int kind, int i, V3 &nrm) {   // towards the origin, src/Scene.cpp:155,172,189
    nrm.x = 0; nrm.y = 0; nrm.z = 0;
    if (kind == PK_PLANE_X) nrm.x = s.f(G_X, i) < 0 ? 1 : -1;
    else if (kind == PK_PLANE_Y) nrm.y = s.f(G_Y, i) < 0 ? 1 : -1;
    else nrm.z = s.f(G_Z, i) < 0 ? 1 : -1;
}
// primitive tests a lane has made (the bench's intersection roofline counts them: 22 flop per sphere test, 6 per plane test,
// SURVEY section 8d); only the counting variants of the kernels pass one, everywhere else the pointer is a compile-time null
struct TestCount { uint32_t spheres, planes, cells, wsteps, wtests; };
__device__ inline bool wave_first_lane() { return (__ffsll((long long)__ballot(1)) - 1) == (int)(threadIdx.x & 63); }
__device__ inline bool prim_hit(const SceneView &s, int i, const V3 &o, const V3 &d, double dt, V3 &res, TestCount *tc = nullptr) {
    int kind = s.kind[i];
    if (kind == PK_SPHERE) { if (tc) tc->spheres++; return sphere_hit(s, i, o, d, dt, res); }
    if (tc) tc->planes++;
    return plane_hit(s, kind, i, o, d, res);
}
__device__ inline void prim_normal(const SceneView &s, int i, double dt, const V3 &res, V3 &nrm) {
    int kind = s.kind[i];
    if (kind == PK_SPHERE) sphere_normal(s, i, dt, res, nrm);
    else plane_normal(s, kind, i, nrm);
}
// one axis of the clip against the grid's box; false = the ray misses the box
__device__ inline bool grid_clip(double lo, double hi, double o, double d, double &inv, double &t0, double &t1) {
    if (d != 0) {
        inv = 1.0 / d;
        double ta = (lo - o) * inv, tb = (hi - o) * inv;
        double tn = ta < tb ? ta : tb, tf = ta < tb ? tb : ta;
        if (tn > t0) t0 = tn;
        if (tf < t1) t1 = tf;
        return true;
    }
    inv = __builtin_inf();
    return !(o < lo || o > hi);
}
// cell index of the entry point along one axis (NaN -> 0) and the parameter at which the ray leaves that cell
__device__ inline int grid_enter(double lo, double cs, double inv_cs, int res, double o, double d, double inv, double t0, double &tnext) {
    double f = ((o + d * t0) - lo) * inv_cs;
    int i = f > 0 ? (f < (double)res ? (int)f : res - 1) : 0;
    tnext = d != 0 ? (lo + (double)(i + (d > 0 ? 1 : 0)) * cs - o) * inv : __builtin_inf();
    return i;
}
template <class F>
__device__ inline void grid_walk(const SceneGrid &G, const V3 &o, const V3 &d, F &&visit) {
    double t0 = 0, t1 = __builtin_inf(), invx, invy, invz;
    if (!grid_clip(G.lo[0], G.hi[0], o.x, d.x, invx, t0, t1)) return;
    if (!grid_clip(G.lo[1], G.hi[1], o.y, d.y, invy, t0, t1)) return;
    if (!grid_clip(G.lo[2], G.hi[2], o.z, d.z, invz, t0, t1)) return;
    if (!(t0 <= t1)) return;
    double tx, ty, tz;
    int ix = grid_enter(G.lo[0], G.cs[0], G.inv_cs[0], G.rx, o.x, d.x, invx, t0, tx);
    int iy = grid_enter(G.lo[1], G.cs[1], G.inv_cs[1], G.ry, o.y, d.y, invy, t0, ty);
    int iz = grid_enter(G.lo[2], G.cs[2], G.inv_cs[2], G.rz, o.z, d.z, invz, t0, tz);
    // One step = one axis advanced by one cell; branch-free, so lanes stepping along different axes stay together.  The
    // crossing parameters are advanced by a constant per axis (rounding drifts by a few ulp per step, far inside the
    // padding of the sphere boxes).
    const int sx = d.x > 0 ? 1 : -1, sy = d.y > 0 ? 1 : -1, sz = d.z > 0 ? 1 : -1;
    const double dtx = G.cs[0] * fabs(invx), dty = G.cs[1] * fabs(invy), dtz = G.cs[2] * fabs(invz);
    const int cx = sx, cy = sy * G.rx, cz = sz * G.rx * G.ry;
    int c = (iz * G.ry + iy) * G.rx + ix;
    double t_in = t0;
    for (int guard = G.rx + G.ry + G.rz + 3; guard > 0; guard--) {
        uint32_t b = G.cell_start[c], e = G.cell_start[c + 1];
        if (!visit(b, e, t_in)) return;
        const bool bx = tx <= ty && tx <= tz, by = !bx && ty <= tz, bz = !bx && !by;
        t_in = bx ? tx : (by ? ty : tz);
        if (!(t_in <= t1)) return;
        ix += bx ? sx : 0; iy += by ? sy : 0; iz += bz ? sz : 0;
        tx += bx ? dtx : 0.0; ty += by ? dty : 0.0; tz += bz ? dtz : 0.0;
        c += bx ? cx : (by ? cy : cz);
        if ((unsigned)ix >= (unsigned)G.rx || (unsigned)iy >= (unsigned)G.ry || (unsigned)iz >= (unsigned)G.rz) return;
    }
}

// Scene::intersect, src/Scene.cpp:25-47: closest hit by squared distance of the reconstructed
// hit point, strict '<' so the lowest index wins ties; returns object id or -1
template <int MODE>
__device__ inline int scene_intersect(const SceneView &s, const V3 &src, const V3 &dir, int exclude, double dt, V3 &fres, V3 &fnrm, TestCount *tc = nullptr) {
    double nearest = 1024 * 1024;
    int nearest_obj = -1;
    V3 res = {0, 0, 0};
    if (MODE != SCENE_GRID) {
        for (int i = 0; i < s.n; i++) {
            if (i == exclude) continue;
            if (prim_hit(s, i, src, dir, dt, res, tc)) {
                double dist = (res.x - src.x) * (res.x - src.x);
                dist += (res.y - src.y) * (res.y - src.y);
                dist += (res.z - src.z) * (res.z - src.z);
                if (dist < nearest) { nearest = dist; nearest_obj = i; fres = res; }
            }
        }
        if (nearest_obj >= 0) prim_normal(s, nearest_obj, dt, fres, fnrm);
        return nearest_obj;
    }
    // The same minimum over (dist, index), with the candidates in another order: a later candidate wins only with a
    // smaller distance, or the same distance and a lower index -- what the strict '<' of the index-order scan yields.
    auto candidate = [&](int i) {
        if (i == exclude) return;
        if (prim_hit(s, i, src, dir, dt, res, tc)) {
            double dist = (res.x - src.x) * (res.x - src.x);
            dist += (res.y - src.y) * (res.y - src.y);
            dist += (res.z - src.z) * (res.z - src.z);
            if (dist < nearest || (dist == nearest && i < nearest_obj)) { nearest = dist; nearest_obj = i; fres = res; }
        }
    };
    const SceneGrid &G = s.grid;
    for (int k = 0; k < G.nloose; k++) candidate(G.loose[k]);
    const double dd = dir.x * dir.x + dir.y * dir.y + dir.z * dir.z;    // 1 within rounding; kept for rays that are not
    // every hit in a cell entered at parameter t lies at squared distance >= t^2 |dir|^2: beyond t_stop nothing can beat
    // `nearest` (margin for the rounding of both sides)
    auto stop_at = [&](double lim) { return sqrt(lim / dd) * (1.0 + 1e-9) + 1e-9; };
    double t_stop = stop_at(nearest);
    grid_walk(G, src, dir, [&](uint32_t b, uint32_t e, double t_in) {
        if (t_in > t_stop) return false;
        for (uint32_t r = b; r < e; r++) {
            const double *P = G.packed + (size_t)r * 8;
            const int i = (int)__double_as_longlong(P[7]);
            if (i == exclude) continue;
            if (tc) tc->spheres++;
            if (sphere_hit_packed(P, src, dir, dt, res)) {
                double dist = (res.x - src.x) * (res.x - src.x);
                dist += (res.y - src.y) * (res.y - src.y);
                dist += (res.z - src.z) * (res.z - src.z);
                if (dist < nearest || (dist == nearest && i < nearest_obj)) { nearest = dist; nearest_obj = i; fres = res; t_stop = stop_at(nearest); }
            }
        }
        return true;
    });
    if (nearest_obj >= 0) prim_normal(s, nearest_obj, dt, fres, fnrm);
    return nearest_obj;
}

// One shadow ray of the loop at src/Raytrace.cpp:190-203: is there a primitive other than `obj` with a hit whose squared
// distance from p is not beyond max_dist (the first such primitive in index order ends the reference's loop; which one
// it is does not matter).
template <int MODE>
__device__ inline bool shadow_blocked(const SceneView &s, const V3 &p, const V3 &ray, double dt, int obj, double max_dist, TestCount *tc = nullptr) {
    V3 result = {0, 0, 0};
    auto blocks = [&](int k) {
        if (k == obj) return false;
        if (!prim_hit(s, k, p, ray, dt, result, tc)) return false;
        double ds = (p.x - result.x) * (p.x - result.x) + (p.y - result.y) * (p.y - result.y) + (p.z - result.z) * (p.z - result.z);
        return !(ds > max_dist);
    };
    if (MODE != SCENE_GRID) {
        for (int k = 0; k < s.n; k++) if (blocks(k)) return true;
        return false;
    }
    const SceneGrid &G = s.grid;
    for (int k = 0; k < G.nloose; k++) if (blocks(G.loose[k])) return true;
    const double dd = ray.x * ray.x + ray.y * ray.y + ray.z * ray.z;
    const double t_stop = sqrt(max_dist / dd) * (1.0 + 1e-9) + 1e-9;    // hits in cells entered later are beyond max_dist
    bool hit = false;
    grid_walk(G, p, ray, [&](uint32_t b, uint32_t e, double t_in) {
        if (t_in > t_stop) return false;
        if (tc) { tc->cells++; if (wave_first_lane()) tc->wsteps++; }
        for (uint32_t r = b; r < e; r++) {
            const double *P = G.packed + (size_t)r * 8;
            if (tc && wave_first_lane()) tc->wtests++;
            if ((int)__double_as_longlong(P[7]) == obj) continue;
            if (tc) tc->spheres++;
            if (!sphere_hit_packed(P, p, ray, dt, result)) continue;
            double ds = (p.x - result.x) * (p.x - result.x) + (p.y - result.y) * (p.y - result.y) + (p.z - result.z) * (p.z - result.z);
            if (!(ds > max_dist)) { hit = true; return false; }
        }
        return true;
    });
    return hit;
}

__device__ inline void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Stage the SoA scene into LDS (all threads of the block call this, then __syncthreads()).
__device__ inline SceneView stage_scene(const SceneView &gs, unsigned char *lds) {
    size_t bytes = scene_bytes(gs.stride);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(gs.g);  // g, rgb, kind are one contiguous block
    uint32_t *dst = reinterpret_cast<uint32_t *>(lds);
    for (size_t w = threadIdx.x + (size_t)threadIdx.y * blockDim.x; w < bytes / 4; w += (size_t)blockDim.x * blockDim.y) dst[w] = src[w];
    SceneView v = gs;
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
