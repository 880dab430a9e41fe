// pm_build.h -- photon emission (one attempt per lane, ordered compaction), the assembly of a map from the ranks'
// shards, and the GPU build of the photon index: Morton keys, the linear octree's cell tables and tight cell boxes.
#pragma once
#include "pm_device.h"
#include "pm_knn.h"
#include "pm_morton.h"
#include "pm_grid.h"
#include "pm_sort.h"
#include "pm_trace.h"
#include "../../include/raytracer_amd.h"

namespace pm {

// ------------------------------------------------------------------ emission
// One pass of the outer loop of createPhotonMap (src/PhotonMap.cpp:288-367) or
// createCausticPhotonMap (:400-484) per lane.  flag[t] = 1 and tmp[t] = record when the attempt
// stores a photon.  block_counts[b] = successes in block b (for the ordered compaction).
constexpr int EMIT_MAX_BOUNCES = 64;   // the reference loops until absorb/miss; bounded here so every wave exits

template <int MODE>
__global__ __launch_bounds__(256) void k_emit(SceneView gs, int map_kind, uint64_t K, V3 light_color, uint64_t a0, uint64_t count,
                                               pm_photon *tmp, uint8_t *flag, uint32_t *block_counts) {
    extern __shared__ __align__(16) unsigned char smem[];
    SceneView sv = gs;
    if (MODE == SCENE_LDS) { sv = stage_scene(gs, smem); __syncthreads(); }
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool stored = false;
    if (t < count) {
        uint64_t IK = item_key(K, a0 + t);
        uint32_t j = 0;
        V3 power = light_color;
        double x0 = draw(IK, j++) * 2 - 1;
        double z0 = draw(IK, j++) * 2 + 3;
        V3 src = {x0, 4.95, z0};
        double sin_theta = sqrt(draw(IK, j++));
        double cos_theta = sqrt(1 - sin_theta * sin_theta);
        double psi = draw(IK, j++) * 6.2831853;
        double sp, cp;
        det_sincos(psi, sp, cp);
        V3 dir = {sin_theta * cp, map_kind == PM_MAP_GLOBAL ? -cos_theta : -2 * cos_theta, sin_theta * sp};
        normalize(dir);
        int bounces = 0, exclude = -1;
        bool specular_bounce = false;
        V3 res = {0, 0, 0}, nrm = {0, 0, 0};
        while (bounces < EMIT_MAX_BOUNCES) {
            bounces++;
            int obj = scene_intersect<MODE>(sv, src, dir, exclude, 0.0, res, nrm);
            if (obj < 0) break;
            double chance = draw(IK, j++);
            if (bounces > 3) chance = 1;
            double kd = sv.f(G_DIFFUSE, obj), ks = sv.f(G_SPECULAR, obj), kt = sv.f(G_TRANSMISSION, obj);
            if (chance < kd + ks) {
                V3 n1 = nrm;
                double k = n1.x * dir.x + n1.y * dir.y + n1.z * dir.z;
                n1.x *= k; n1.y *= k; n1.z *= k;
                n1.x *= 2; n1.y *= 2; n1.z *= 2;
                src = res;
                dir.x = dir.x - n1.x; dir.y = dir.y - n1.y; dir.z = dir.z - n1.z;
                normalize(dir);
                power.x *= (double)sv.rgb[0 * sv.stride + obj]; power.y *= (double)sv.rgb[1 * sv.stride + obj]; power.z *= (double)sv.rgb[2 * sv.stride + obj];
                if (map_kind == PM_MAP_GLOBAL) exclude = obj;   // :327 ; the caustic loop always passes nullptr (:420)
                specular_bounce = true;
                continue;
            } else if (chance < kd + ks + kt) {
                double n = 1 / sv.f(G_REFRACTION, obj);
                double d = nrm.x * dir.x + nrm.y * dir.y + nrm.z * dir.z;
                V3 n1 = {nrm.x * d, nrm.y * d, nrm.z * d};
                n1.x = dir.x - n1.x; n1.y = dir.y - n1.y; n1.z = dir.z - n1.z;
                n1.x *= n; n1.y *= n; n1.z *= n;
                double s = 1 - (n * n) * (1 - d * d);
                double sq = sqrt(s);
                V3 n2 = {nrm.x * sq, nrm.y * sq, nrm.z * sq};
                n1.x += -n2.x; n1.y += -n2.y; n1.z += -n2.z;
                src.x = res.x + n1.x * 0.01; src.y = res.y + n1.y * 0.01; src.z = res.z + n1.z * 0.01;
                dir = n1;
                normalize(dir);
                exclude = -1;
                specular_bounce = true;
                continue;
            } else {
                if (map_kind == PM_MAP_CAUSTIC && !specular_bounce) break;   // :463-466
                pm_photon ph;
                ph.x = (float)res.x; ph.y = (float)res.y; ph.z = (float)res.z;
                ph.power[0] = (uint8_t)fastfloor(power.x * 0xFF);
                ph.power[1] = (uint8_t)fastfloor(power.y * 0xFF);
                ph.power[2] = (uint8_t)fastfloor(power.z * 0xFF);
                ph.power[3] = 255;
                ph.dx = (float)dir.x; ph.dy = (float)dir.y; ph.dz = (float)dir.z;
                ph.bounce = (int8_t)bounces;
                ph.pad[0] = ph.pad[1] = ph.pad[2] = 0;
                tmp[t] = ph;
                stored = true;
            }
            break;
        }
        flag[t] = stored ? 1 : 0;
    }
    int c = __syncthreads_count(stored ? 1 : 0);
    if (threadIdx.x == 0) block_counts[blockIdx.x] = (uint32_t)c;
}

// The same attempts with refilled lanes (round 3): a photon makes one to four segments (a bounce is taken with the surface's
// k_d + k_s, the fourth hit stores), so a wave of k_emit runs four rounds of intersections with 47 % of its lanes.  Here a wave
// owns a run of EMIT_RUN consecutive attempts; a lane whose photon is stored, lost or absorbed takes the next attempt of the run in
// the same round.  The first rays of the attempts are set up 64 at a time by the whole wave (the set-up is a third of an
// attempt's instructions: done for the few lanes that fall idle in a round it would cost as much as for 64) and wait in LDS.  flag[t] / tmp[t] are indexed by the attempt, so
// the photon order does not depend on which lane traced what; the arithmetic of an attempt is k_emit's, statement by statement.
#ifndef PM_EMIT_RUN_LEN
#define PM_EMIT_RUN_LEN 256
#endif
constexpr uint32_t EMIT_RUN = PM_EMIT_RUN_LEN;
static_assert(EMIT_RUN % 256 == 0 && EMIT_RUN <= 1024, "a run is a whole number of the compaction's blocks of 256 attempts");
// LDS of one wave: the first rays of up to 64 attempts, set up by the whole wave at once (one attempt per lane) and handed to
// lanes as they fall idle.  [field][entry]: x0, z0, dir x y z, and the attempt's key.
struct EmitPool { double f[5][64]; uint64_t ik[64]; };
template <int MODE>
__global__ __launch_bounds__(256) void k_emit_run(SceneView gs, int map_kind, uint64_t K, V3 light_color, uint64_t a0, uint64_t count,
                                                   pm_photon *tmp, uint8_t *flag, uint32_t *block_counts) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ EmitPool pools[4];
    SceneView sv = gs;
    if (MODE == SCENE_LDS) { sv = stage_scene(gs, smem); __syncthreads(); }
    const int lane = threadIdx.x & 63;
    EmitPool &pool = pools[threadIdx.x >> 6];
    const uint64_t r0 = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * EMIT_RUN;
    if (r0 >= count) return;                                   // (wave-uniform, behind the block's only barrier)
    const uint32_t len = (uint32_t)(count - r0 < EMIT_RUN ? count - r0 : EMIT_RUN);
    uint32_t next = 0;                                         // attempts of the run set up so far
    uint32_t pool_base = 0, pool_n = 0, pool_at = 0;           // the pool holds attempts pool_base .. pool_base + pool_n, pool_at of them handed out
    uint32_t nstored[EMIT_RUN / 256];
#pragma unroll
    for (uint32_t b = 0; b < EMIT_RUN / 256; b++) nstored[b] = 0;
    bool active = false, specular_bounce = false;
    uint32_t off = 0, j = 0;
    uint64_t IK = 0;
    V3 power = light_color, src = {0, 0, 0}, dir = {0, 0, 0};
    int bounces = 0, exclude = -1;
    for (;;) {
        const unsigned long long idle = ~__ballot(active);
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        if (n_idle != 0u && (pool_at < pool_n || next < len)) {
            if (pool_at == pool_n) {
                // the pool is empty: the next 64 attempts of the run are set up, one per lane (src/PhotonMap.cpp:290-301, :402-413)
                wave_lds_fence();
                const uint32_t o = next + (uint32_t)lane;
                if (o < len) {
                    const uint64_t ik = item_key(K, a0 + r0 + o);
                    uint32_t jj = 0;
                    double x0 = draw(ik, jj++) * 2 - 1;
                    double z0 = draw(ik, jj++) * 2 + 3;
                    double sin_theta = sqrt(draw(ik, jj++));
                    double cos_theta = sqrt(1 - sin_theta * sin_theta);
                    double psi = draw(ik, jj++) * 6.2831853;
                    double sp, cp;
                    det_sincos(psi, sp, cp);
                    V3 d0 = {sin_theta * cp, map_kind == PM_MAP_GLOBAL ? -cos_theta : -2 * cos_theta, sin_theta * sp};
                    normalize(d0);
                    pool.f[0][lane] = x0; pool.f[1][lane] = z0; pool.f[2][lane] = d0.x; pool.f[3][lane] = d0.y; pool.f[4][lane] = d0.z;
                    pool.ik[lane] = ik;
                }
                pool_base = next;
                pool_n = len - next < 64u ? len - next : 64u;
                pool_at = 0;
                next += pool_n;
                wave_lds_fence();
            }
            const uint32_t e = pool_at + (uint32_t)lane_rank(idle);
            if (!active && e < pool_n) {
                off = pool_base + e;
                IK = pool.ik[e];
                j = 4;
                power = light_color;
                src = {pool.f[0][e], 4.95, pool.f[1][e]};
                dir = {pool.f[2][e], pool.f[3][e], pool.f[4][e]};
                bounces = 0; exclude = -1; specular_bounce = false;
                active = true;
            }
            pool_at = pool_at + n_idle < pool_n ? pool_at + n_idle : pool_n;
        }
        if (__ballot(active) == 0ULL) {
            if (pool_at == pool_n && next >= len) break;      // nothing in flight, nothing left to hand out
            continue;                                         // (the pool ran dry half way through a hand-out: it is filled next)
        }
        bool done = false, stored = false;
        if (active) {
            V3 res = {0, 0, 0}, nrm = {0, 0, 0};
            bounces++;
            int obj = scene_intersect<MODE>(sv, src, dir, exclude, 0.0, res, nrm);
            if (obj < 0) done = true;
            else {
                double chance = draw(IK, j++);
                if (bounces > 3) chance = 1;
                double kd = sv.f(G_DIFFUSE, obj), ks = sv.f(G_SPECULAR, obj), kt = sv.f(G_TRANSMISSION, obj);
                if (chance < kd + ks) {
                    V3 n1 = nrm;
                    double k = n1.x * dir.x + n1.y * dir.y + n1.z * dir.z;
                    n1.x *= k; n1.y *= k; n1.z *= k;
                    n1.x *= 2; n1.y *= 2; n1.z *= 2;
                    src = res;
                    dir.x = dir.x - n1.x; dir.y = dir.y - n1.y; dir.z = dir.z - n1.z;
                    normalize(dir);
                    power.x *= (double)sv.rgb[0 * sv.stride + obj]; power.y *= (double)sv.rgb[1 * sv.stride + obj]; power.z *= (double)sv.rgb[2 * sv.stride + obj];
                    if (map_kind == PM_MAP_GLOBAL) exclude = obj;
                    specular_bounce = true;
                } else if (chance < kd + ks + kt) {
                    double n = 1 / sv.f(G_REFRACTION, obj);
                    double d = nrm.x * dir.x + nrm.y * dir.y + nrm.z * dir.z;
                    V3 n1 = {nrm.x * d, nrm.y * d, nrm.z * d};
                    n1.x = dir.x - n1.x; n1.y = dir.y - n1.y; n1.z = dir.z - n1.z;
                    n1.x *= n; n1.y *= n; n1.z *= n;
                    double s = 1 - (n * n) * (1 - d * d);
                    double sq = sqrt(s);
                    V3 n2 = {nrm.x * sq, nrm.y * sq, nrm.z * sq};
                    n1.x += -n2.x; n1.y += -n2.y; n1.z += -n2.z;
                    src.x = res.x + n1.x * 0.01; src.y = res.y + n1.y * 0.01; src.z = res.z + n1.z * 0.01;
                    dir = n1;
                    normalize(dir);
                    exclude = -1;
                    specular_bounce = true;
                } else {
                    done = true;
                    if (!(map_kind == PM_MAP_CAUSTIC && !specular_bounce)) {
                        pm_photon ph;
                        ph.x = (float)res.x; ph.y = (float)res.y; ph.z = (float)res.z;
                        ph.power[0] = (uint8_t)fastfloor(power.x * 0xFF);
                        ph.power[1] = (uint8_t)fastfloor(power.y * 0xFF);
                        ph.power[2] = (uint8_t)fastfloor(power.z * 0xFF);
                        ph.power[3] = 255;
                        ph.dx = (float)dir.x; ph.dy = (float)dir.y; ph.dz = (float)dir.z;
                        ph.bounce = (int8_t)bounces;
                        ph.pad[0] = ph.pad[1] = ph.pad[2] = 0;
                        tmp[r0 + off] = ph;
                        stored = true;
                    }
                }
            }
            if (!done && bounces >= EMIT_MAX_BOUNCES) done = true;
            if (done) { flag[r0 + off] = stored ? 1 : 0; active = false; }
        }
#pragma unroll
        for (uint32_t b = 0; b < EMIT_RUN / 256; b++) nstored[b] += (uint32_t)__popcll(__ballot(stored && (off >> 8) == b));
    }
    if (lane == 0) {
#pragma unroll
        for (uint32_t b = 0; b < EMIT_RUN / 256; b++)
            if (r0 + 256ull * b < count) block_counts[(r0 >> 8) + b] = nstored[b];
    }
}

// The same attempts through the sphere grid of a large scene, as a stream (pm_trace.h): a wave keeps 64 attempts in flight.
// A lane whose photon is absorbed, lost or stored takes the next attempt from the launch's counter while its neighbours go
// on with theirs; a lane whose photon bounces starts the next segment in place.  Lanes are served in batches (16 waiting, or
// nobody walking), since serving runs the bounce arithmetic and the set-up of a walk for the whole wave.  flag[t] / tmp[t]
// are indexed by the attempt, so the order of the photons does not depend on which lane traced what; the successes per
// block of 256 attempts are counted afterwards from the flags (k_flag_counts).
constexpr int TRACE_SERVE_MIN = 32;     // 8: 6.1 ms for both maps of the 10 000 sphere scene, 16: 3.9, 24: 3.5, 32: 3.4, 48: 3.5
// four waves per SIMD: 128 registers without a spill (3: 135 registers, 4.05 ms for both maps; 4: 3.9; 5 and 6 spill in the walk: 5.5, 8.2)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_emit_stream(SceneView sv, int map_kind, uint64_t K, V3 light_color, uint64_t a0, uint64_t count,
                                                      pm_photon *tmp, uint8_t *flag, uint32_t *next) {
    __shared__ TraceBuf bufs[4];
    WaveTracer<false, true> tr(sv, bufs[threadIdx.x >> 6]);
    const int lane = threadIdx.x & 63;
    bool have = false, more = true, specular_bounce = false;
    uint64_t t = 0, IK = 0;
    uint32_t j = 0;
    V3 power = light_color;
    int bounces = 0, exclude = -1;
    for (;;) {
        const bool fin = tr.finished();
        const unsigned long long m_fin = __ballot(fin), m_have = __ballot(have);
        if (!m_have && !more) break;
        const int n_busy = __popcll(m_have & ~m_fin), n_serve = __popcll(m_fin) + (more ? 64 - __popcll(m_have) : 0);
        TRACE_STAT(8, 1);
        if (n_serve > 0 && (n_serve >= TRACE_SERVE_MIN || n_busy == 0)) {
            TRACE_STAT(0, 1); TRACE_STAT(1, __popcll(m_fin));
            bool fresh = false;
            V3 src = {0, 0, 0}, dir = {0, 0, 0};
            if (fin) {
                // the body of the reference's bounce loop behind the intersection, src/PhotonMap.cpp:300-367 / :412-484
                V3 res = {0, 0, 0}, nrm = {0, 0, 0};
                dir = {tr.wb.dx[lane], tr.wb.dy[lane], tr.wb.dz[lane]};
                const int obj = tr.result(res, nrm);
                bool stored = false, again = false;
                if (obj >= 0) {
                    double chance = draw(IK, j++);
                    if (bounces > 3) chance = 1;
                    double kd = sv.f(G_DIFFUSE, obj), ks = sv.f(G_SPECULAR, obj), kt = sv.f(G_TRANSMISSION, obj);
                    if (chance < kd + ks) {
                        V3 n1 = nrm;
                        double k = n1.x * dir.x + n1.y * dir.y + n1.z * dir.z;
                        n1.x *= k; n1.y *= k; n1.z *= k;
                        n1.x *= 2; n1.y *= 2; n1.z *= 2;
                        src = res;
                        dir.x = dir.x - n1.x; dir.y = dir.y - n1.y; dir.z = dir.z - n1.z;
                        normalize(dir);
                        power.x *= (double)sv.rgb[0 * sv.stride + obj]; power.y *= (double)sv.rgb[1 * sv.stride + obj]; power.z *= (double)sv.rgb[2 * sv.stride + obj];
                        if (map_kind == PM_MAP_GLOBAL) exclude = obj;
                        specular_bounce = true;
                        again = true;
                    } else if (chance < kd + ks + kt) {
                        double n = 1 / sv.f(G_REFRACTION, obj);
                        double d = nrm.x * dir.x + nrm.y * dir.y + nrm.z * dir.z;
                        V3 n1 = {nrm.x * d, nrm.y * d, nrm.z * d};
                        n1.x = dir.x - n1.x; n1.y = dir.y - n1.y; n1.z = dir.z - n1.z;
                        n1.x *= n; n1.y *= n; n1.z *= n;
                        double sq = sqrt(1 - (n * n) * (1 - d * d));
                        V3 n2 = {nrm.x * sq, nrm.y * sq, nrm.z * sq};
                        n1.x += -n2.x; n1.y += -n2.y; n1.z += -n2.z;
                        src.x = res.x + n1.x * 0.01; src.y = res.y + n1.y * 0.01; src.z = res.z + n1.z * 0.01;
                        dir = n1;
                        normalize(dir);
                        exclude = -1;
                        specular_bounce = true;
                        again = true;
                    } else if (!(map_kind == PM_MAP_CAUSTIC && !specular_bounce)) {
                        pm_photon ph;
                        ph.x = (float)res.x; ph.y = (float)res.y; ph.z = (float)res.z;
                        ph.power[0] = (uint8_t)fastfloor(power.x * 0xFF);
                        ph.power[1] = (uint8_t)fastfloor(power.y * 0xFF);
                        ph.power[2] = (uint8_t)fastfloor(power.z * 0xFF);
                        ph.power[3] = 255;
                        ph.dx = (float)dir.x; ph.dy = (float)dir.y; ph.dz = (float)dir.z;
                        ph.bounce = (int8_t)bounces;
                        ph.pad[0] = ph.pad[1] = ph.pad[2] = 0;
                        tmp[t] = ph;
                        stored = true;
                    }
                }
                if (again && bounces < EMIT_MAX_BOUNCES) { bounces++; fresh = true; }
                else { flag[t] = stored ? 1 : 0; have = false; }
            }
            if (more) {
                // the lanes without an attempt take the next ones of the launch, in lane order
                const unsigned long long m_need = __ballot(!have);
                const uint32_t n = (uint32_t)__popcll(m_need);
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(next, n);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                const uint64_t mine = (uint64_t)base + (uint64_t)__popcll(m_need & tr.below);
                TRACE_STAT(2, n);
                if (!have && mine < count) {
                    t = mine;
                    IK = item_key(K, a0 + t);
                    j = 0;
                    power = light_color;
                    double x0 = draw(IK, j++) * 2 - 1;
                    double z0 = draw(IK, j++) * 2 + 3;
                    src = {x0, 4.95, z0};
                    double sin_theta = sqrt(draw(IK, j++));
                    double cos_theta = sqrt(1 - sin_theta * sin_theta);
                    double psi = draw(IK, j++) * 6.2831853;
                    double sp, cp;
                    det_sincos(psi, sp, cp);
                    dir = {sin_theta * cp, map_kind == PM_MAP_GLOBAL ? -cos_theta : -2 * cos_theta, sin_theta * sp};
                    normalize(dir);
                    bounces = 1; exclude = -1; specular_bounce = false;
                    have = true; fresh = true;
                }
                if ((uint64_t)base + n >= count) more = false;
            }
            tr.start(fresh, src, dir, 0.0, exclude, 0.0);
            continue;
        }
        tr.pump();
    }
}
// successes per block of 256 attempts, from the flags
__global__ __launch_bounds__(256) void k_flag_counts(const uint8_t *flag, uint64_t count, uint32_t *block_counts) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = __syncthreads_count(t < count && flag[t] != 0);
    if (threadIdx.x == 0) block_counts[blockIdx.x] = (uint32_t)c;
}

// exclusive scan of nb block counts by one 1024-thread block; offs[nb] = total.  running != NULL: a map emitted in several
// ranges chains them on the device -- the scan starts from *running (the successes of the ranges before) and leaves
// *running at the new total, so the scatter that follows appends behind the earlier ranges without the host knowing a count.
// Every thread sums a run of consecutive counts, the 1024 sums are scanned with wave shuffles, the runs are written out.
// bounds_init != NULL: the six words of the index build's bounds are reset here (minima all ones, maxima zero), which
// saves the build a copy from the host.
__global__ __launch_bounds__(1024) void k_scan_blocks(const uint32_t *counts, uint64_t *offs, uint32_t nb, uint64_t *running, uint32_t *bounds_init) {
    __shared__ uint64_t wsum[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (bounds_init && threadIdx.x < 6) bounds_init[threadIdx.x] = threadIdx.x < 3 ? 0xFFFFFFFFu : 0u;
    const uint32_t per = (nb + 1023u) / 1024u, lo = min(threadIdx.x * per, nb), hi = min(lo + per, nb);
    uint64_t mine = 0;
    for (uint32_t i = lo; i < hi; i++) mine += counts[i];
    uint64_t inc = mine;
    for (int s = 1; s < 64; s <<= 1) { uint64_t y = __shfl_up(inc, s); if (lane >= s) inc += y; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint64_t before = running ? *running : 0;
    for (int w = 0; w < wave; w++) before += wsum[w];
    uint64_t run = before + inc - mine;
    for (uint32_t i = lo; i < hi; i++) { offs[i] = run; run += counts[i]; }
    __syncthreads();                                  // every thread has read *running
    if (threadIdx.x == 1023) { offs[nb] = run; if (running) *running = run; }
}

__global__ __launch_bounds__(256) void k_emit_scatter(const pm_photon *tmp, const uint8_t *flag, const uint64_t *offs, uint64_t count,
                                                       pm_photon *out, uint64_t cap) {
    __shared__ uint32_t wave_cnt[4];
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bool f = t < count && flag[t] != 0;
    unsigned long long m = __ballot(f);
    if (lane == 0) wave_cnt[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t before = 0;
    for (int w = 0; w < wave; w++) before += wave_cnt[w];
    if (f) {
        uint64_t dst = offs[blockIdx.x] + before + (uint32_t)__popcll(m & ((1ULL << lane) - 1ULL));
        if (dst < cap) out[dst] = tmp[t];
    }
}

// ------------------------------------------------------------------ shards of a map emitted by several ranks
// A shard: 32-byte header {count, cap, 0, 0} + cap records.  The map is the first n records in rank order.
// status[0] = photons available (sum of counts), status[1] = 1 when a shard overflowed its cap.
__global__ __launch_bounds__(256) void k_assemble_shards(const unsigned char *shards, int world, unsigned long long stride, long long n, pm_photon *out,
                                                          unsigned long long *status) {
    __shared__ unsigned long long first[65];
    if (threadIdx.x == 0) {
        unsigned long long run = 0, over = 0;
        for (int r = 0; r < world; r++) {
            const unsigned long long *h = reinterpret_cast<const unsigned long long *>(shards + (size_t)r * stride);
            first[r] = run;
            if (h[0] > h[1]) over = 1;
            run += h[0] < h[1] ? h[0] : h[1];
        }
        first[world] = run;
        if (blockIdx.x == 0) { status[0] = run; status[1] = over; }
    }
    __syncthreads();
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || (unsigned long long)i >= first[world]) return;
    int r = 0;
    while (first[r + 1] <= (unsigned long long)i) r++;
    const uint4 *src = reinterpret_cast<const uint4 *>(shards + (size_t)r * stride + 32) + 2 * ((unsigned long long)i - first[r]);
    uint4 *dst = reinterpret_cast<uint4 *>(out + i);
    dst[0] = src[0]; dst[1] = src[1];
}

// ------------------------------------------------------------------ index build
__device__ inline uint32_t f2ord(float f) { uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ inline float ord2f(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }

// bounds[0..2] = min xyz, bounds[3..5] = max xyz as order-preserving uints (init: 0xFFFFFFFF / 0).
// Wave reduction, then block reduction through LDS, then six atomics per BLOCK (six addresses shared by
// every wave of the grid would serialise at the memory side).
__global__ __launch_bounds__(256) void k_bounds(const pm_photon *ph, long long n, uint32_t *bounds) {
    __shared__ uint32_t part[4][6];
    uint32_t mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[3] = {0, 0, 0};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        uint32_t v[3] = {f2ord(ph[i].x), f2ord(ph[i].y), f2ord(ph[i].z)};
        for (int a = 0; a < 3; a++) { mn[a] = min(mn[a], v[a]); mx[a] = max(mx[a], v[a]); }
    }
    for (int a = 0; a < 3; a++) {
        for (int s = 1; s < 64; s <<= 1) { mn[a] = min(mn[a], (uint32_t)__shfl_xor((int)mn[a], s)); mx[a] = max(mx[a], (uint32_t)__shfl_xor((int)mx[a], s)); }
        if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6][a] = mn[a]; part[threadIdx.x >> 6][3 + a] = mx[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        uint32_t v = part[0][threadIdx.x];
        for (int w = 1; w < 4; w++) v = threadIdx.x < 3 ? min(v, part[w][threadIdx.x]) : max(v, part[w][threadIdx.x]);
        if (threadIdx.x < 3) atomicMin(&bounds[threadIdx.x], v); else atomicMax(&bounds[threadIdx.x], v);
    }
}

// Keys of the photons (blocks of SORT_TILE, the sort's own tiling), the sort's first histogram (digit 0 of every block:
// k_sort_count of pass 0, pm_sort.h) and, by the first thread, the quantisation grid the queries need (pm_knn.h, MapView).
__global__ __launch_bounds__(256) void k_morton(const pm_photon *ph, long long n, const uint32_t *bounds, uint64_t *keys, uint32_t *vals, uint32_t nblk,
                                                 uint32_t *table, GridParams *g) {
    __shared__ uint32_t hist[SORT_DIGITS];
    for (int d = threadIdx.x; d < SORT_DIGITS; d += 256) hist[d] = 0u;
    // one cubic grid over the largest extent (anisotropic scaling would distort the curve's locality)
    float lo[3], ext = 0.f;
    for (int a = 0; a < 3; a++) { lo[a] = ord2f(bounds[a]); ext = fmaxf(ext, ord2f(bounds[3 + a]) - lo[a]); }
    if (blockIdx.x == 0 && threadIdx.x == 0) { g->glo[0] = lo[0]; g->glo[1] = lo[1]; g->glo[2] = lo[2]; g->gext = ext; g->gu = (double)ext / 2097152.0; }
    __syncthreads();
    const long long base = (long long)blockIdx.x * SORT_TILE, end = min(base + (long long)SORT_TILE, n);
    for (long long i = base + threadIdx.x; i < end; i += 256) {
        const uint64_t k = morton63(quant21(ph[i].x, lo[0], ext), quant21(ph[i].y, lo[1], ext), quant21(ph[i].z, lo[2], ext));
        keys[i] = k;                                // linear octree: aligned cells are contiguous key ranges
        vals[i] = (uint32_t)i;
        atomicAdd(&hist[sort_digit(k, SORT_FIRST_BIT)], 1u);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < SORT_DIGITS; d += 256) table[(size_t)d * nblk + blockIdx.x] = hist[d];
}

// ---- the cell table of the linear octree: cells[c] = first sorted photon of cell c of the FINEST table level lmax (8^lmax + 1
// entries; the last one is n).  A cell of a coarser level l is the aligned run of 8^(lmax - l) fine cells, so its range is
// read from the same table (pm_grid.h, cell_range).  Built from the photon side: the table is preset to all ones, the first
// photon of every occupied cell writes its index (k_permute), and the empty cells take the start of the next occupied one:
// a suffix minimum, in two launches (block minima; then every block looks right for the first block that has one).
constexpr int FILL_TILE = 2048;                          // table entries per block of the fill (256 threads x 8)
__global__ __launch_bounds__(256) void k_fill_mins(const uint32_t *cells, long long n_entries, uint32_t *bmin) {
    __shared__ uint32_t part[4];
    const long long base = (long long)blockIdx.x * FILL_TILE;
    uint32_t m = 0xFFFFFFFFu;
    for (int j = threadIdx.x; j < FILL_TILE; j += 256) { const long long i = base + j; if (i < n_entries) m = min(m, cells[i]); }
    for (int s = 1; s < 64; s <<= 1) m = min(m, (uint32_t)__shfl_xor((int)m, s));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) bmin[blockIdx.x] = min(min(part[0], part[1]), min(part[2], part[3]));
}
__global__ __launch_bounds__(256) void k_fill_cells(uint32_t *cells, long long n_entries, const uint32_t *bmin, uint32_t nblk) {
    __shared__ uint32_t part[4], wmin[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the first occupied entry to the right of this block (starts only grow to the right: the minimum is the nearest)
    uint32_t m = 0xFFFFFFFFu;
    for (uint32_t b0 = blockIdx.x + 1; b0 < nblk; b0 += 256) {
        const uint32_t b = b0 + threadIdx.x;
        m = b < nblk ? bmin[b] : 0xFFFFFFFFu;
        if (__syncthreads_or(m != 0xFFFFFFFFu)) break;
    }
    for (int s = 1; s < 64; s <<= 1) m = min(m, (uint32_t)__shfl_xor((int)m, s));
    if (lane == 0) part[wave] = m;
    // this thread's eight consecutive entries; the minimum of everything to its right inside the block by wave shuffles
    const long long at = (long long)blockIdx.x * FILL_TILE + (long long)threadIdx.x * 8;
    uint32_t v[8], mine = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < 8; j++) { v[j] = at + j < n_entries ? cells[at + j] : 0xFFFFFFFFu; mine = min(mine, v[j]); }
    uint32_t suf = mine;                                 // inclusive suffix minimum over the lanes of the wave
    for (int s = 1; s < 64; s <<= 1) { const uint32_t y = (uint32_t)__shfl_down((int)suf, s); if (lane + s < 64) suf = min(suf, y); }
    if (lane == 0) wmin[wave] = suf;
    uint32_t right = (uint32_t)__shfl_down((int)suf, 1);
    if (lane == 63) right = 0xFFFFFFFFu;
    __syncthreads();
    for (int w = wave + 1; w < 4; w++) right = min(right, wmin[w]);
    right = min(right, min(min(part[0], part[1]), min(part[2], part[3])));
#pragma unroll
    for (int j = 7; j >= 0; j--) {
        if (v[j] == 0xFFFFFFFFu) v[j] = right; else right = v[j];
        if (at + j < n_entries) cells[at + j] = v[j];
    }
}

// ---- tight boxes of the octree cells (pm_grid.h): per cell and axis the lowest and highest occupied 1/1024 slice,
// 6 x 10 bits in one 64-bit word; all ones for an empty cell.  Photon sets are flat (they lie on surfaces) while cells are
// cubes: the box of a floor cell is a slab, and a query ball that dips into the cube without reaching the floor can drop the cell.
constexpr uint64_t BOX_EMPTY = ~0ULL;
__device__ inline uint64_t pack_box(const uint32_t lo[3], const uint32_t hi[3]) {
    uint64_t w = 0;
    for (int a = 0; a < 3; a++) w |= ((uint64_t)lo[a] << (20 * a)) | ((uint64_t)hi[a] << (20 * a + 10));
    return w;
}
// a cell's box from its eight children's: union of the non-empty ones, halved into the parent's 1/1024 slices
__device__ inline uint64_t box_of_children(const uint64_t *child) {
    uint32_t lo[3] = {1023u, 1023u, 1023u}, hi[3] = {0u, 0u, 0u};
    bool any = false;
    for (int j = 0; j < 8; j++) {
        const uint64_t w = child[j];
        if (w == BOX_EMPTY) continue;
        any = true;
        for (int a = 0; a < 3; a++) {
            uint32_t half = ((uint32_t)(j >> a) & 1u) * 512u;
            lo[a] = min(lo[a], half + (((uint32_t)(w >> (20 * a)) & 1023u) >> 1));
            hi[a] = max(hi[a], half + (((uint32_t)(w >> (20 * a + 10)) & 1023u) >> 1));
        }
    }
    return any ? pack_box(lo, hi) : BOX_EMPTY;
}
// The finest level and the `depth` (<= 3) levels above it in one launch: a block takes one cell of level lmax - depth, works
// out the boxes of its 8^depth fine cells from their photons (one lane per cell -- most cells hold a handful of photons
// or none; a cell with more than 32 is reduced by the whole wave, one with more than 512 is a dense blob that queries
// split further anyway: its box is the whole cube) and folds them up level by level through LDS.
__global__ __launch_bounds__(256) void k_cell_boxes_tree(const float4 *pos4, const uint32_t *cells, int lmax, int depth, const uint32_t *bounds, uint64_t *cbox) {
    __shared__ uint64_t lbox[512 + 64 + 8 + 1];
    const int lane = threadIdx.x & 63;
    const int nfine = 1 << (3 * depth);
    float glo[3] = {ord2f(bounds[0]), ord2f(bounds[1]), ord2f(bounds[2])}, ext = 0.f;
    for (int a = 0; a < 3; a++) ext = fmaxf(ext, ord2f(bounds[3 + a]) - glo[a]);
    const int sh = 21 - lmax;   // >= 13
    auto slice = [&](const float4 &p, uint32_t lo[3], uint32_t hi[3]) {
        float pa[3] = {p.x, p.y, p.z};
        for (int a = 0; a < 3; a++) {
            uint32_t sub = (quant21(pa[a], glo[a], ext) & ((1u << sh) - 1u)) >> (sh - 10);
            lo[a] = min(lo[a], sub); hi[a] = max(hi[a], sub);
        }
    };
    for (int j0 = 0; j0 < nfine; j0 += 256) {            // (a wave runs a round whole or not at all: the ballots below are whole)
        const int j = j0 + threadIdx.x;
        if (j0 + (int)(threadIdx.x & ~63u) >= nfine) break;
        const long long c = (long long)blockIdx.x * nfine + j;
        uint32_t first = 0, end = 0;
        if (j < nfine) { first = cells[c]; end = cells[c + 1]; }
        uint64_t w = BOX_EMPTY;
        const bool huge = end - first > 512u, big = !huge && end - first > 32u;
        if (huge) { const uint32_t lo[3] = {0u, 0u, 0u}, hi[3] = {1023u, 1023u, 1023u}; w = pack_box(lo, hi); }
        else if (end > first && !big) {
            uint32_t lo[3] = {1023u, 1023u, 1023u}, hi[3] = {0u, 0u, 0u};
            for (uint32_t i = first; i < end; i++) slice(pos4[i], lo, hi);
            w = pack_box(lo, hi);
        }
        for (unsigned long long todo = __ballot(big); todo; todo &= todo - 1) {
            const int src = __ffsll((long long)todo) - 1;
            const uint32_t f = (uint32_t)__shfl((int)first, src), e = (uint32_t)__shfl((int)end, src);
            uint32_t lo[3] = {1023u, 1023u, 1023u}, hi[3] = {0u, 0u, 0u};
            for (uint32_t i = f + (uint32_t)lane; i < e; i += 64u) slice(pos4[i], lo, hi);
            for (int a = 0; a < 3; a++)
                for (int s = 1; s < 64; s <<= 1) { lo[a] = min(lo[a], (uint32_t)__shfl_xor((int)lo[a], s)); hi[a] = max(hi[a], (uint32_t)__shfl_xor((int)hi[a], s)); }
            if (lane == src) w = pack_box(lo, hi);
        }
        if (j < nfine) { cbox[box_off(lmax) + c] = w; lbox[j] = w; }
    }
    // levels lmax - 1 .. lmax - depth of this block's subtree
    int below = 0, count = nfine;                        // offset and size of the level below in lbox
    for (int d = 1; d <= depth; d++) {
        __syncthreads();
        const int mine = count >> 3, at = below + count;
        for (int j = threadIdx.x; j < mine; j += 256) {
            const uint64_t w = box_of_children(&lbox[below + 8 * j]);
            lbox[at + j] = w;
            cbox[box_off(lmax - d) + (long long)blockIdx.x * mine + j] = w;
        }
        below = at; count = mine;
    }
}
// levels ltop, ltop - 1, ..., 0 in one block (the levels above the subtrees of k_cell_boxes_tree): one launch instead of one per
// level.  A small block: beside wide kernels of the frame a 16-wave block waits a long time for a compute unit with room for all of it.
__global__ __launch_bounds__(256) void k_cell_boxes_top(int ltop, uint64_t *cbox) {
    for (int l = ltop; l >= 0; l--) {
        for (long long c = threadIdx.x; c < (1LL << (3 * l)); c += blockDim.x) cbox[box_off(l) + c] = box_of_children(&cbox[box_off(l + 1) + 8 * c]);
        __threadfence_block();
        __syncthreads();
    }
}

// sorted order -> the two 16-byte streams the gather reads; and the first photon of every occupied cell of the finest
// table level enters its index into the cell table (preset to all ones; the last entry is n)
__global__ __launch_bounds__(256) void k_permute(const pm_photon *ph, const uint32_t *order, const uint64_t *keys, long long n, int lmax, float4 *pos4, uint4 *rest,
                                                  uint32_t *cells) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t o = order[i];
    const uint4 *rec = reinterpret_cast<const uint4 *>(&ph[o]);
    uint4 a = rec[0], b = rec[1];   // a = x y z power ; b = dx dy dz bounce
    pos4[i] = make_float4(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(o));
    rest[i] = make_uint4(b.x, b.y, b.z, a.w);
    const int ks = 63 - 3 * lmax;
    const uint64_t c = lmax ? keys[i] >> ks : 0ULL;
    if (i == 0 || (lmax ? keys[i - 1] >> ks : 0ULL) != c) cells[c] = (uint32_t)i;
    if (i == 0) cells[1LL << (3 * lmax)] = (uint32_t)n;
}

}  // namespace pm
