// pm_trace.h -- rays through the sphere grid of a large scene (SCENE_GRID), traced by a wave as a stream.
//
// Lane by lane (grid_walk in pm_device.h) a wave of rays is slow for two reasons, both measured on the 10 000 sphere scene:
// a cell holds 0.7 sphere references on average, so the test loop of a step runs as often as the fullest cell of the 64
// lanes asks for (18 % of the lanes busy in the tests of the shadow pass); and rays end at different times -- a shadow ray
// at its blocker, a photon at the bounce where it is absorbed -- while the wave goes on to its longest ray (56 % of the
// lanes busy in the steps of the shadow pass, 40 % over the bounces of an emission attempt).
//
// WaveTracer separates the three things a lane does and lets the kernel decide when each happens:
//   start   a lane is given a ray: the planes and the spheres outside the grid are tested at once (they are few), the
//           ray's operands go to the lane's slot in LDS, the walk is set up
//   pump    one action of the wave: every walking lane queues (ray, reference) pairs of the cell it is in -- a ring in LDS
//           shared by the wave; when 64 pairs are queued (or few lanes are still walking) the wave tests 64 pairs, one per
//           lane, operands read from the slot of the pair's ray: full rounds whatever cells the pairs came from; otherwise
//           every walking lane whose cell is queued steps into its next cell
//   finished / result   a ray is complete when its walk has ended and the ring has moved past its last pair; the kernel
//           takes the result and may give the lane its next ray (a bounce, or a new work item) while the other lanes go on
//
// The answers are those of scene_intersect / shadow_blocked (pm_device.h), bit for bit.  The tests are the same functions on
// the same operands; the closest hit is the minimum over (distance, index) of the hits and an any-hit answer is the OR over
// them, both independent of the order of the tests; and testing a pair too many -- a ray runs ahead of its queued tests by a
// few cells, and a pair may be tested against the ray that took over its lane -- adds only hits that are real hits of that
// ray, which change neither a minimum over all of the ray's hits nor an OR.
#pragma once
#include "pm_device.h"
#include "pm_knn.h"

namespace pm {

constexpr int TQ_CAP = 256;                 // pairs in the ring (a power of two)
constexpr uint32_t TR_NONE = 0x7FFFFFFFu;
struct TraceBuf {                           // of one wave
    double ox[64], oy[64], oz[64], dx[64], dy[64], dz[64], dt[64], lim[64];
    unsigned long long near[64];            // closest hit: bits of the smallest squared distance so far
    uint32_t win[64];                       // closest hit: the primitive that has it (TR_NONE: none); any hit: 1 = blocked
    int skip[64];                           // the primitive the ray leaves from
    uint32_t qref[TQ_CAP];
    unsigned char qsrc[TQ_CAP];
};
__device__ inline unsigned long long f64_bits(double v) { return (unsigned long long)__double_as_longlong(v); }
__device__ inline double bits_f64(unsigned long long v) { return __longlong_as_double((long long)v); }

#ifdef PM_TRACE_STATS
// diagnostics (builds with -DPM_TRACE_STATS only): wave-level and lane-level counts of the tracer's actions
static __device__ unsigned long long g_trace_stats[16];
#define TRACE_STAT(k, v) do { const unsigned long long v_ = (unsigned long long)(v); const unsigned long long m_ = __ballot(1); if ((int)(threadIdx.x & 63) == __ffsll((long long)m_) - 1) atomicAdd(&g_trace_stats[k], v_); } while (0)
#else
#define TRACE_STAT(k, v) do { } while (0)
#endif

// EAGER: with few lanes still walking the ring fills slowly; test what is queued early, so that the lanes that wait for
// their result (and for their next ray) do not wait for it
template <bool ANYHIT, bool EAGER>
struct WaveTracer {
    const SceneView &s;
    TraceBuf &wb;
    TestCount *tc;
    const int lane;
    const unsigned long long below;
    // lane
    bool pending = false;                   // the lane has a ray whose result has not been taken
    bool walking = false;                   // ... which is still entering cells
    uint32_t r = 0, e = 0;                  // references of its cell that are not queued yet
    uint32_t done_at = 0;                   // ring position behind the ray's last pair
    double tx = 0, ty = 0, tz = 0, dtx = 0, dty = 0, dtz = 0, t1 = 0, t_stop = 0, dd = 1;
    int ix = 0, iy = 0, iz = 0, sx = 1, sy = 1, sz = 1, c = 0, guard = 0;
    unsigned long long seen = 0;            // closest hit: the distance t_stop was derived from
    // wave
    uint32_t head = 0, tail = 0;

    __device__ WaveTracer(const SceneView &scene, TraceBuf &buf, TestCount *count = nullptr)
        : s(scene), wb(buf), tc(count), lane((int)(threadIdx.x & 63)), below((1ULL << (threadIdx.x & 63)) - 1ULL) {}

    // every hit in a cell entered at parameter t lies at squared distance >= t^2 |d|^2 (scene_intersect's bound)
    __device__ double stop_at(double lim) const { return sqrt(lim / dd) * (1.0 + 1e-9) + 1e-9; }

    // Lanes with fresh = true (they hold no ray) take the ray (o, d, dt) leaving primitive `skip`; any hit: blocked when a hit
    // lies within squared distance `lim`.  Every lane of the wave calls it.
    __device__ void start(bool fresh, const V3 &o, const V3 &d, double dt, int skip, double lim) {
        const SceneGrid &G = s.grid;
        if (fresh) {
            double nearest = 1024 * 1024;
            int nobj = -1;
            bool blocked = false;
            V3 res = {0, 0, 0};
            for (int k = 0; k < G.nloose && !blocked; k++) {
                const int i = G.loose[k];
                if (i == skip || !prim_hit(s, i, o, d, dt, res, tc)) continue;
                double dist = (res.x - o.x) * (res.x - o.x);
                dist += (res.y - o.y) * (res.y - o.y);
                dist += (res.z - o.z) * (res.z - o.z);
                if (ANYHIT) blocked = !(dist > lim);
                else if (dist < nearest || (dist == nearest && i < nobj)) { nearest = dist; nobj = i; }
            }
            wb.ox[lane] = o.x; wb.oy[lane] = o.y; wb.oz[lane] = o.z; wb.dx[lane] = d.x; wb.dy[lane] = d.y; wb.dz[lane] = d.z;
            wb.dt[lane] = dt; wb.lim[lane] = lim; wb.skip[lane] = skip;
            wb.near[lane] = f64_bits(nearest);
            wb.win[lane] = ANYHIT ? (blocked ? 1u : 0u) : (nobj < 0 ? TR_NONE : (uint32_t)nobj);
            seen = f64_bits(nearest);
            pending = true;
            r = e = 0;
            // grid_walk's set-up
            bool ok = !blocked;
            double t0 = 0, invx = 0, invy = 0, invz = 0;
            t1 = __builtin_inf();
            if (ok) ok = grid_clip(G.lo[0], G.hi[0], o.x, d.x, invx, t0, t1);
            if (ok) ok = grid_clip(G.lo[1], G.hi[1], o.y, d.y, invy, t0, t1);
            if (ok) ok = grid_clip(G.lo[2], G.hi[2], o.z, d.z, invz, t0, t1);
            if (ok) ok = t0 <= t1;
            if (ok) {
                dd = d.x * d.x + d.y * d.y + d.z * d.z;
                t_stop = stop_at(ANYHIT ? lim : nearest);
                ok = !(t0 > t_stop);
            }
            if (ok) {
                ix = grid_enter(G.lo[0], G.cs[0], G.inv_cs[0], G.rx, o.x, d.x, invx, t0, tx);
                iy = grid_enter(G.lo[1], G.cs[1], G.inv_cs[1], G.ry, o.y, d.y, invy, t0, ty);
                iz = grid_enter(G.lo[2], G.cs[2], G.inv_cs[2], G.rz, o.z, d.z, invz, t0, tz);
                sx = d.x > 0 ? 1 : -1; sy = d.y > 0 ? 1 : -1; sz = d.z > 0 ? 1 : -1;
                dtx = G.cs[0] * fabs(invx); dty = G.cs[1] * fabs(invy); dtz = G.cs[2] * fabs(invz);
                c = (iz * G.ry + iy) * G.rx + ix;
                guard = G.rx + G.ry + G.rz + 3;
                r = G.cell_start[c]; e = G.cell_start[c + 1];
                if (tc) tc->cells++;
            }
            walking = ok;
            done_at = head;                  // nothing queued for this ray yet
        }
        wave_lds_fence();
    }

    __device__ bool finished() const { return pending && !walking && r >= e && (int)(head - done_at) >= 0; }

    // any hit: the answer of a finished ray; the lane is free afterwards
    __device__ bool blocked() { pending = false; return wb.win[lane] != 0u; }
    // closest hit of a finished ray (object id or -1, hit point, normal); the lane is free afterwards
    __device__ int result(V3 &res, V3 &nrm) {
        pending = false;
        const uint32_t w = wb.win[lane];
        if (w == TR_NONE) return -1;
        const V3 o = {wb.ox[lane], wb.oy[lane], wb.oz[lane]}, d = {wb.dx[lane], wb.dy[lane], wb.dz[lane]};
        const double dt = wb.dt[lane];
        (void)prim_hit(s, (int)w, o, d, dt, res);      // the winner's hit point once more: same operands, same bits
        prim_normal(s, (int)w, dt, res, nrm);
        return (int)w;
    }

    __device__ void test_round(uint32_t avail) {
        const SceneGrid &G = s.grid;
        wave_lds_fence();
        const uint32_t take = avail < 64u ? avail : 64u;
        if (tc && wave_first_lane()) tc->wtests++;
        TRACE_STAT(5, 1); TRACE_STAT(6, take);
        bool hit = false;
        double dist = 0;
        int i = 0, src = 0;
        if ((uint32_t)lane < take) {
            const uint32_t at = (head + (uint32_t)lane) & (TQ_CAP - 1);
            src = wb.qsrc[at];
            const double *P = G.packed + (size_t)wb.qref[at] * 8;
            i = (int)__double_as_longlong(P[7]);
            if (i != wb.skip[src]) {
                if (tc) tc->spheres++;
                const V3 o = {wb.ox[src], wb.oy[src], wb.oz[src]}, d = {wb.dx[src], wb.dy[src], wb.dz[src]};
                V3 res = {0, 0, 0};
                if (sphere_hit_packed(P, o, d, wb.dt[src], res)) {
                    dist = (res.x - o.x) * (res.x - o.x);
                    dist += (res.y - o.y) * (res.y - o.y);
                    dist += (res.z - o.z) * (res.z - o.z);
                    if (ANYHIT) { if (!(dist > wb.lim[src])) wb.win[src] = 1u; }
                    else hit = dist < 1024.0 * 1024.0 || (dist == 1024.0 * 1024.0 && wb.win[src] != TR_NONE);
                }
            }
        }
        if (!ANYHIT) {
            // minimum over (distance, index) per ray: the distance first, then the index among the hits at that distance
            const unsigned long long bits = f64_bits(dist);
            unsigned long long before = 0;
            if (hit) before = wb.near[src];
            wave_lds_fence();
            if (hit) atomicMin(&wb.near[src], bits);
            wave_lds_fence();
            const bool top = hit && wb.near[src] == bits;
            if (top && bits < before) wb.win[src] = TR_NONE;     // a new distance: the holder of the old one is out
            wave_lds_fence();
            if (top) atomicMin(&wb.win[src], (uint32_t)i);
        }
        head += take;
        wave_lds_fence();
        if (ANYHIT) {
            if (pending && wb.win[lane] != 0u) { walking = false; r = e; done_at = head; }
        } else if (walking) {
            const unsigned long long nb = wb.near[lane];
            if (nb != seen) { seen = nb; t_stop = stop_at(bits_f64(nb)); }
        }
    }

    __device__ void step(bool can_step) {
        const SceneGrid &G = s.grid;
        if (tc && wave_first_lane()) tc->wsteps++;
        TRACE_STAT(3, 1); TRACE_STAT(4, __popcll(__ballot(can_step)));
        if (can_step) {
            const bool bx = tx <= ty && tx <= tz, by = !bx && ty <= tz, bz = !bx && !by;
            const double t_in = bx ? tx : (by ? ty : tz);
            ix += bx ? sx : 0; iy += by ? sy : 0; iz += bz ? sz : 0;
            tx += bx ? dtx : 0.0; ty += by ? dty : 0.0; tz += bz ? dtz : 0.0;
            c += bx ? sx : (by ? sy * G.rx : sz * G.rx * G.ry);
            guard--;
            walking = t_in <= t1 && guard > 0 && !(t_in > t_stop) &&
                      !((unsigned)ix >= (unsigned)G.rx || (unsigned)iy >= (unsigned)G.ry || (unsigned)iz >= (unsigned)G.rz);
            if (walking) { r = G.cell_start[c]; e = G.cell_start[c + 1]; if (tc) tc->cells++; }
            else done_at = tail;
        }
    }

    // one action of the wave
    __device__ void pump() {
        // queue the references of the cells just entered: every lane its own run of slots behind the tail (a prefix sum over
        // the lanes' counts), or, when the ring cannot take them all, one reference per lane and turn as far as it goes
        const uint32_t cnt = walking ? e - r : 0u;
        if (__ballot(cnt != 0u)) {
            const uint32_t incl = wave_scan_incl_u32(cnt);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (tail - head + total <= (uint32_t)TQ_CAP) {
                uint32_t at = tail + incl - cnt;
                for (uint32_t k = r; k < e; k++, at++) { wb.qref[at & (TQ_CAP - 1)] = k; wb.qsrc[at & (TQ_CAP - 1)] = (unsigned char)lane; }
                r = e;
                tail += total;
            } else {
                for (;;) {
                    const bool has = walking && r < e;
                    const unsigned long long m = __ballot(has);
                    if (!m) break;
                    const uint32_t n = (uint32_t)__popcll(m);
                    if (tail - head + n > (uint32_t)TQ_CAP) break;
                    if (has) {
                        const uint32_t at = (tail + (uint32_t)__popcll(m & below)) & (TQ_CAP - 1);
                        wb.qref[at] = r; wb.qsrc[at] = (unsigned char)lane;
                        r++;
                    }
                    tail += n;
                }
            }
        }
        const bool can_step = walking && r >= e;
        const bool any_step = __ballot(can_step) != 0ULL;
        const uint32_t nwalk = (uint32_t)__popcll(__ballot(walking)), avail = tail - head;
        const uint32_t want = EAGER && 2u * nwalk < 64u ? 2u * nwalk : 64u;
        if (avail > 0u && (avail >= want || !any_step)) test_round(avail);
        else if (any_step) step(can_step);
    }
};

}  // namespace pm
