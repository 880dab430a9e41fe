// pm_build.h -- photon emission (one attempt per lane, ordered compaction), the assembly of a map from the ranks'
// shards, and the GPU build of the photon index: Morton keys, the linear octree's cell tables and tight cell boxes.
#pragma once
#include "pm_device.h"
#include "pm_knn.h"
#include "pm_morton.h"
#include "pm_grid.h"
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

// exclusive scan of nb block counts by one 1024-thread block; offs[nb] = total.  running != NULL: a map emitted in several
// ranges chains them on the device -- the scan starts from *running (the successes of the ranges before) and leaves
// *running at the new total, so the scatter that follows appends behind the earlier ranges without the host knowing a count.
__global__ __launch_bounds__(1024) void k_scan_blocks(const uint32_t *counts, uint64_t *offs, uint32_t nb, uint64_t *running) {
    __shared__ uint64_t buf[1024];
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = running ? *running : 0;
    __syncthreads();
    for (uint32_t base = 0; base < nb; base += 1024) {
        uint32_t i = base + threadIdx.x;
        uint64_t v = i < nb ? counts[i] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int s = 1; s < 1024; s <<= 1) {
            uint64_t add = threadIdx.x >= (unsigned)s ? buf[threadIdx.x - s] : 0;
            __syncthreads();
            buf[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < nb) offs[i] = carry + buf[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += buf[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) { offs[nb] = carry; if (running) *running = carry; }
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

__global__ __launch_bounds__(256) void k_morton(const pm_photon *ph, long long n, const uint32_t *bounds, uint64_t *keys, uint32_t *vals) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float p[3] = {ph[i].x, ph[i].y, ph[i].z};
    // one cubic grid over the largest extent (anisotropic scaling would distort the curve's locality)
    float ext = 0.f;
    for (int a = 0; a < 3; a++) ext = fmaxf(ext, ord2f(bounds[3 + a]) - ord2f(bounds[a]));
    uint32_t qv[3];
    for (int a = 0; a < 3; a++) qv[a] = quant21(p[a], ord2f(bounds[a]), ext);
    keys[i] = morton63(qv[0], qv[1], qv[2]);     // linear octree: aligned cells are contiguous key ranges
    vals[i] = (uint32_t)i;
}

// the quantisation grid of a map, from the bounds: what the queries need (pm_knn.h, MapView)
__global__ void k_grid_params(const uint32_t *bounds, GridParams *g) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float ext = 0.f;
    for (int a = 0; a < 3; a++) { g->glo[a] = ord2f(bounds[a]); ext = fmaxf(ext, ord2f(bounds[3 + a]) - ord2f(bounds[a])); }
    g->gext = ext;
    g->gu = (double)ext / 2097152.0;
}

// cell start tables of the linear octree: entry c of level l = first sorted photon whose key prefix is >= c
__global__ __launch_bounds__(256) void k_cell_tables(const uint64_t *keys, long long n, uint32_t *cells, int lmax) {
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= cell_off(lmax + 1)) return;
    int l = 0;
    while (l < lmax && t >= cell_off(l + 1)) l++;
    unsigned long long c = (unsigned long long)(t - cell_off(l));
    long long lo = 0, hi = n;
    if (c >> (3 * l)) lo = n;   // the sentinel entry 8^l
    else {
        unsigned long long target = l == 0 ? 0ULL : c << (63 - 3 * l);
        while (lo < hi) { long long mid = (lo + hi) >> 1; if (keys[mid] < target) lo = mid + 1; else hi = mid; }
    }
    cells[t] = (uint32_t)lo;
}

// ---- tight boxes of the octree cells (pm_grid.h): per cell and axis the lowest and highest occupied 1/1024 slice,
// 6 x 10 bits in one 64-bit word.  Photon sets are flat (they lie on surfaces) while cells are cubes: the box of a
// floor cell is a slab, and a query ball that dips into the cube without reaching the floor can drop the cell.
__device__ inline uint64_t pack_box(const uint32_t lo[3], const uint32_t hi[3]) {
    uint64_t w = 0;
    for (int a = 0; a < 3; a++) w |= ((uint64_t)lo[a] << (20 * a)) | ((uint64_t)hi[a] << (20 * a + 10));
    return w;
}
// finest table level: one lane per cell (most cells hold a handful of photons or none); a cell with more than 32
// photons is then reduced by the whole wave
__global__ __launch_bounds__(256) void k_cell_boxes_fine(const float4 *pos4, const uint32_t *cells, int lmax, const uint32_t *bounds, uint64_t *cbox) {
    const long long ncell = 1LL << (3 * lmax);
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const uint32_t *tab = cells + cell_off(lmax);
    uint32_t first = 0, end = 0;
    if (c < ncell) { first = tab[c]; end = tab[c + 1]; }
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
    // a cell holding many hundreds of photons is a dense blob that queries split further anyway: its box is the whole cube
    const bool huge = end - first > 512u;
    if (huge) { const uint32_t lo[3] = {0u, 0u, 0u}, hi[3] = {1023u, 1023u, 1023u}; cbox[box_off(lmax) + c] = pack_box(lo, hi); }
    const bool big = !huge && end - first > 32u;
    if (end > first && !big && !huge) {
        uint32_t lo[3] = {1023u, 1023u, 1023u}, hi[3] = {0u, 0u, 0u};
        for (uint32_t i = first; i < end; i++) slice(pos4[i], lo, hi);
        cbox[box_off(lmax) + c] = pack_box(lo, hi);
    }
    for (unsigned long long todo = __ballot(big); todo; todo &= todo - 1) {
        const int src = __ffsll((long long)todo) - 1;
        const uint32_t f = (uint32_t)__shfl((int)first, src), e = (uint32_t)__shfl((int)end, src);
        uint32_t lo[3] = {1023u, 1023u, 1023u}, hi[3] = {0u, 0u, 0u};
        for (uint32_t i = f + (uint32_t)lane; i < e; i += 64u) slice(pos4[i], lo, hi);
        for (int a = 0; a < 3; a++)
            for (int s = 1; s < 64; s <<= 1) { lo[a] = min(lo[a], (uint32_t)__shfl_xor((int)lo[a], s)); hi[a] = max(hi[a], (uint32_t)__shfl_xor((int)hi[a], s)); }
        if (lane == src) cbox[box_off(lmax) + c] = pack_box(lo, hi);
    }
}
// level l from level l + 1: union of the non-empty children, halved into the parent's 1/1024 slices
__device__ inline void cell_box_up(const uint32_t *cells, int l, long long c, uint64_t *cbox) {
    const uint32_t *tab = cells + cell_off(l + 1);
    uint32_t lo[3] = {1023u, 1023u, 1023u}, hi[3] = {0u, 0u, 0u};
    bool any = false;
    for (int j = 0; j < 8; j++) {
        long long k = 8 * c + j;
        if (tab[k] == tab[k + 1]) continue;
        any = true;
        uint64_t w = cbox[box_off(l + 1) + k];
        for (int a = 0; a < 3; a++) {
            uint32_t half = ((uint32_t)(j >> a) & 1u) * 512u;
            lo[a] = min(lo[a], half + (((uint32_t)(w >> (20 * a)) & 1023u) >> 1));
            hi[a] = max(hi[a], half + (((uint32_t)(w >> (20 * a + 10)) & 1023u) >> 1));
        }
    }
    if (any) cbox[box_off(l) + c] = pack_box(lo, hi);
}
__global__ __launch_bounds__(256) void k_cell_boxes_up(const uint32_t *cells, int l, uint64_t *cbox) {
    long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < (1LL << (3 * l))) cell_box_up(cells, l, c, cbox);
}
// levels ltop, ltop - 1, ..., 0 in one block (8^ltop <= 512 cells): one launch instead of one per level.  A small block:
// beside wide kernels of the frame a 16-wave block waits a long time for a compute unit with room for all of it.
__global__ __launch_bounds__(256) void k_cell_boxes_top(const uint32_t *cells, int ltop, uint64_t *cbox) {
    for (int l = ltop; l >= 0; l--) {
        for (long long c = threadIdx.x; c < (1LL << (3 * l)); c += blockDim.x) cell_box_up(cells, l, c, cbox);
        __threadfence_block();
        __syncthreads();
    }
}

// sorted order -> the two 16-byte streams the gather reads
__global__ __launch_bounds__(256) void k_permute(const pm_photon *ph, const uint32_t *order, long long n, float4 *pos4, uint4 *rest) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t o = order[i];
    const uint4 *rec = reinterpret_cast<const uint4 *>(&ph[o]);
    uint4 a = rec[0], b = rec[1];   // a = x y z power ; b = dx dy dz bounce
    pos4[i] = make_float4(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(o));
    rest[i] = make_uint4(b.x, b.y, b.z, a.w);
}

}  // namespace pm
