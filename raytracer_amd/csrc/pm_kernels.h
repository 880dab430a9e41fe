// pm_kernels.h -- the gfx950 kernels of the frame pipeline.
//
//   k_primary        one lane per pixel (8x8 pixel tile per wave): primary ray + closest hit,
//                    classifies the pixel, wave-ballot compaction of absorb points ("shade points")
//                    and of pixels that bounce (reflect / refract) into a second queue
//   k_secondary      one lane per bounced pixel: the traceRay recursion as an explicit-stack state
//                    machine (shared mutable (src,ray) pair); emits further shade points
//   k_shadow         one lane per shade point: 25 shadow any-hit rays + direct term
//   k_sp_*           the bounced pixels' shade points in Morton order, cut into groups by octree cell
//   k_gather_tile<KIND>  one wave per GROUP of neighbouring shade points, one lane per query (pm_tile.h)
//   k_gather<KIND>   one WAVE per shade point, for what the group kernel leaves: exact 63-NN + radiance estimate
//                    (pm_grid.h, pm_knn.h)
//   k_shade          one lane per shade point: clamp, colour, 8-bit pack; final pixel for simple pixels
//   k_resolve        one lane per bounced pixel: folds its node table bottom-up into the final pixel
#pragma once
#include "pm_device.h"
#include "pm_knn.h"
#include "pm_grid.h"
#include "pm_tile.h"
#include "pm_pair.h"
#include "pm_trace.h"

namespace pm {

struct FrameParams {
    int width, height;
    V3 cam;
    V3 light;
    uint64_t K;          // stream key (seed, KIND_PIXEL)
    int flags;
    int rank, world;
};

struct Counters { uint32_t n_sp, n_groups, n_complex, n_secondary, n_left[2][4], pad, n_sgroups, n_early, n_pieces, overflow, n_sp0, n_bounced, done_primary, done_secondary, n_loose, n_sorted; unsigned long long tests[4]; unsigned long long walk[4]; uint32_t why[8]; unsigned long long tphase[8]; uint32_t ghist[2][32]; };   // n_left[k][p]: shade points that part p of map k's group launches left to the per-query gather; n_sp and n_groups are one 64-bit word for k_primary (one atomic per block takes slots and group entries together)

// The frame's counts live on the device: nothing of a frame waits for the host.  n_sp0 = shade points of primary hits (slots
// 0 .. n_sp0, grouped by pixel tile), written by k_mark_primary behind k_primary; n_bounced = shade points of bounced pixels
// (slots n_sp0 .. n_sp0 + n_bounced), written by k_mark_secondary behind k_secondary.  Every later kernel is launched for the
// capacity of the queues and reads what there really is from here.  overflow: bit 0 = a shade point did not fit, bit 1 = more
// bounced pixels than node tables, bit 2 = more bounced pixels' shade points than the spatial sort was sized for; the host
// looks at it once, when the frame is over, grows what was too small and renders the frame again.
// n_loose: shade points of primary hits whose pixel tile is spread out in depth (k_primary lists their slots instead of making
// the tile a group); n_sorted = n_bounced + n_loose: what goes through the spatial sort.
enum { OVF_SP = 1u, OVF_COMPLEX = 2u, OVF_BOUNCED = 4u };
__device__ inline uint32_t sp_total(const Counters *ctr, uint32_t cap) { return min(ctr->n_sp0 + ctr->n_bounced, cap); }
// (A counter of finished blocks would let the last block of k_primary write n_sp0 itself, but 8 100 blocks adding to one
// address serialise at about 90 ns each: 0.7 ms.  Two one-wave kernels between the passes cost 5 us each.)
// shade point queue (SoA)
struct SPQueue {
    double *geo;         // [6][cap]: px py pz nx ny nz
    int32_t *obj;        // [cap]
    uint32_t *pixel;     // [cap]
    uint32_t *node;      // [cap] node id | 0x80000000 when the point belongs to a bounced pixel's table
    uint32_t *tabslot;   // [cap] index into the node table for bounced pixels
    double *direct;      // [cap]
    double *rad_g;       // [3][cap]
    double *rad_c;       // [3][cap]
    uint32_t *color;     // [cap] absorb_res
    uint32_t *overflow;  // Counters::overflow: set when a shade point did not fit
    uint32_t cap;
};

constexpr uint32_t NODE_COMPLEX = 0x80000000u;

__device__ inline bool is_light_patch(const V3 &r) {   // src/Raytrace.cpp:43
    return r.y > 4.95 && r.x > -1 && r.x < 1 && r.z > 3 && r.z < 5;
}
__device__ inline uint32_t light_colour(const V3 &lc) { // src/Raytrace.cpp:46
    return (0xFFu << 24) | ((uint32_t)imin(fastfloor(lc.x * 0xFF) + 50, 0xFF) << 16) |
           ((uint32_t)imin(fastfloor(lc.y * 0xFF) + 50, 0xFF) << 8) | (uint32_t)imin(fastfloor(lc.z * 0xFF) + 50, 0xFF);
}
// the mix at src/Raytrace.cpp:245-257
__device__ inline uint32_t mix_colour(uint32_t reflect_res, uint32_t refract_res, uint32_t absorb_res, double ks, double kt, double ka) {
    float sred = ((reflect_res >> 16) & 0xFF) / 255.0f, sgreen = ((reflect_res >> 8) & 0xFF) / 255.0f, sblue = (reflect_res & 0xFF) / 255.0f;
    float tred = ((refract_res >> 16) & 0xFF) / 255.0f, tgreen = ((refract_res >> 8) & 0xFF) / 255.0f, tblue = (refract_res & 0xFF) / 255.0f;
    float ared = ((absorb_res >> 16) & 0xFF) / 255.0f, agreen = ((absorb_res >> 8) & 0xFF) / 255.0f, ablue = (absorb_res & 0xFF) / 255.0f;
    uint32_t red = (uint32_t)fastfloor((sred * ks + tred * kt + ared * ka) * 255);
    uint32_t green = (uint32_t)fastfloor((sgreen * ks + tgreen * kt + agreen * ka) * 255);
    uint32_t blue = (uint32_t)fastfloor((sblue * ks + tblue * kt + ablue * ka) * 255);
    return (0xFFu << 24) | (red << 16) | (green << 8) | blue;
}

__device__ inline void sp_store(const SPQueue &q, uint32_t slot, const V3 &res, const V3 &nrm, int obj, uint32_t pixel, uint32_t node, uint32_t tabslot) {
    // capacity is sized for the worst case on the host; should it ever not be, the frame is reported as failed (PM_ESTATE)
    // rather than delivered with pixels missing
    if (slot >= q.cap) { atomicOr(q.overflow, (uint32_t)OVF_SP); return; }
    q.geo[0 * (size_t)q.cap + slot] = res.x; q.geo[1 * (size_t)q.cap + slot] = res.y; q.geo[2 * (size_t)q.cap + slot] = res.z;
    q.geo[3 * (size_t)q.cap + slot] = nrm.x; q.geo[4 * (size_t)q.cap + slot] = nrm.y; q.geo[5 * (size_t)q.cap + slot] = nrm.z;
    q.obj[slot] = obj; q.pixel[slot] = pixel; q.node[slot] = node; q.tabslot[slot] = tabslot;
}

// pixel of this lane: 16x16 tile per 256-thread block, 8x8 per wave; blockIdx.y walks owned row blocks
__device__ inline bool tile_pixel(const FrameParams &fp, int &x, int &y) {
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    x = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7);
    int yb = blockIdx.y * fp.world + fp.rank;
    y = yb * 16 + (wave >> 1) * 8 + (lane >> 3);
    return x < fp.width && y < fp.height;
}

// sum of a lane counter over the wave, added to a device counter by one lane (the counting variants of the kernels)
__device__ inline void wave_add_count(unsigned long long *to, uint32_t mine) {
    uint32_t v = mine;
    for (int s = 1; s < 64; s <<= 1) v += (uint32_t)__shfl_xor((int)v, s);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(to, (unsigned long long)v);
}
// loose_k > 0: a tile whose shade points lie at squared distances from the camera further apart than this factor is no group
// of neighbours (a sphere in front of a wall, several small spheres one behind the other): its points are listed in `loose`
// and grouped with the bounced pixels' points by where they lie, not by the pixel they are seen through.
template <int MODE, bool COUNT>
__device__ inline void primary_body(const SceneView &gs, const FrameParams &fp, uint32_t *pane, Counters *ctr, const SPQueue &q, uint32_t *complex_list, uint2 *groups,
                                    uint32_t *loose, double loose_k) {
    extern __shared__ __align__(16) unsigned char smem[];
    TestCount tc = {0u, 0u, 0u, 0u, 0u};
    SceneView sv = gs;
    if (MODE == SCENE_LDS) { sv = stage_scene(gs, smem); __syncthreads(); }
    int x, y;
    bool valid = tile_pixel(fp, x, y);
    int lane = threadIdx.x & 63;
    bool emit_sp = false, emit_cx = false;
    V3 res = {0, 0, 0}, nrm = {0, 0, 0};
    int obj = -1;
    uint32_t pixel = 0;
    if (valid) {
        pixel = (uint32_t)x + (uint32_t)y * (uint32_t)fp.width;
        V3 ray = primary_ray(fp.width, fp.height, fp.cam, fp.K, pixel, x, y);
        double dt = draw(item_key(fp.K, (uint64_t)pixel * 32u + 1u), 0);   // traceRay's own draw, src/Raytrace.cpp:39
        obj = scene_intersect<MODE>(sv, fp.cam, ray, -1, dt, res, nrm, COUNT ? &tc : nullptr);
        if (obj < 0) pane[pixel] = 0xFF000000u;
        else if (is_light_patch(res)) pane[pixel] = light_colour(fp.light);
        else if (sv.f(G_TRANSMISSION, obj) > 0 || sv.f(G_SPECULAR, obj) > 0) emit_cx = true;
        else if (sv.f(G_ABSORB, obj) > 0) emit_sp = true;
        else pane[pixel] = mix_colour(0, 0, 0, sv.f(G_SPECULAR, obj), sv.f(G_TRANSMISSION, obj), sv.f(G_ABSORB, obj));
    }
    // wavefront-ballot compaction, one atomic per BLOCK for the shade point slots and the group entries of its four tiles
    // together ({n_sp, n_groups} is one 64-bit word): an atomic per wave and counter -- 16 000 on one cache line at 1440 x 1440,
    // a quarter of a million at 7680 x 4320 -- was what the pass waited for
    __shared__ uint32_t blk_cnt[4], blk_grp[4], blk_cx[4], blk_cx_base;
    __shared__ unsigned long long blk_base;
    const int wave = threadIdx.x >> 6;
    unsigned long long m = __ballot(emit_sp);
    const uint32_t count = (uint32_t)__popcll(m);
    bool spread = false;
    if (m && loose_k > 0) {
        const double d2 = (res.x - fp.cam.x) * (res.x - fp.cam.x) + (res.y - fp.cam.y) * (res.y - fp.cam.y) + (res.z - fp.cam.z) * (res.z - fp.cam.z);
        const double near2 = wave_min_f64(emit_sp ? d2 : 1e300), far2 = wave_max_f64(emit_sp ? d2 : 0.0);
        spread = far2 > near2 * loose_k;
    }
    const unsigned long long mcx = __ballot(emit_cx);
    if (lane == 0) { blk_cnt[wave] = count; blk_grp[wave] = count && !spread ? 1u : 0u; blk_cx[wave] = (uint32_t)__popcll(mcx); }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t total = blk_cnt[0] + blk_cnt[1] + blk_cnt[2] + blk_cnt[3], ng = blk_grp[0] + blk_grp[1] + blk_grp[2] + blk_grp[3];
        blk_base = total ? atomicAdd(reinterpret_cast<unsigned long long *>(&ctr->n_sp), (unsigned long long)total | ((unsigned long long)ng << 32)) : 0ULL;
    }
    if (threadIdx.x == 64) {
        const uint32_t total = blk_cx[0] + blk_cx[1] + blk_cx[2] + blk_cx[3];
        blk_cx_base = total ? atomicAdd(&ctr->n_complex, total) : 0u;
    }
    __syncthreads();
    if (m) {
        uint32_t base = (uint32_t)blk_base, gslot = (uint32_t)(blk_base >> 32);
        for (int w = 0; w < wave; w++) { base += blk_cnt[w]; gslot += blk_grp[w]; }
        const int leader = __ffsll((long long)m) - 1;
        uint32_t lbase = 0;
        if (lane == leader) {
            if (spread) lbase = atomicAdd(&ctr->n_loose, count);
            else groups[gslot] = make_uint2(base, count);   // the tile's shade points: one gather group
        }
        lbase = __shfl(lbase, leader);
        const uint32_t rank = (uint32_t)__popcll(m & ((1ULL << lane) - 1ULL));
        if (emit_sp) sp_store(q, base + rank, res, nrm, obj, pixel, 1u, 0u);
        if (emit_sp && spread && lbase + rank < q.cap) loose[lbase + rank] = base + rank;
    }
    if (emit_cx) {
        uint32_t base = blk_cx_base;
        for (int w = 0; w < wave; w++) base += blk_cx[w];
        complex_list[base + (uint32_t)__popcll(mcx & ((1ULL << lane) - 1ULL))] = pixel;
    }
    if (COUNT) { wave_add_count(&ctr->tests[0], tc.spheres); wave_add_count(&ctr->tests[1], tc.planes); }
}
template <int MODE>
__global__ __launch_bounds__(256) void k_primary(SceneView gs, FrameParams fp, uint32_t *pane, Counters *ctr, SPQueue q, uint32_t *complex_list, uint2 *groups, uint32_t *loose, double loose_k) {
    primary_body<MODE, false>(gs, fp, pane, ctr, q, complex_list, groups, loose, loose_k);
}
// the same pass counting the primitive tests it makes (PM_FLAG_COUNT_TESTS: the bench's intersection roofline; never the timed frame)
template <int MODE>
__global__ __launch_bounds__(256) void k_primary_count(SceneView gs, FrameParams fp, uint32_t *pane, Counters *ctr, SPQueue q, uint32_t *complex_list, uint2 *groups, uint32_t *loose, double loose_k) {
    primary_body<MODE, true>(gs, fp, pane, ctr, q, complex_list, groups, loose, loose_k);
}
// the counts of the pass that just ended, for the kernels behind it (one thread)
__global__ void k_mark_primary(Counters *ctr, uint32_t cap) {
    if (threadIdx.x == 0 && blockIdx.x == 0) ctr->n_sp0 = min(ctr->n_sp, cap);
}
__global__ void k_mark_secondary(Counters *ctr, uint32_t cap, uint32_t b_cap) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const uint32_t total = min(ctr->n_sp, cap), nb = total - ctr->n_sp0, nl = min(ctr->n_loose, cap);
    if ((unsigned long long)nb + nl > b_cap) atomicOr(&ctr->overflow, (uint32_t)OVF_BOUNCED);
    ctr->n_bounced = min(nb, b_cap);     // what the kernels behind may touch; the flag has the frame rendered again
    ctr->n_sorted = ctr->n_bounced + min(nl, b_cap - ctr->n_bounced);
}

// hit ids / points / normals only (parity gate for primary rays)
template <int MODE>
__global__ __launch_bounds__(256) void k_primary_hits(SceneView gs, FrameParams fp, int y0, int y1, int32_t *ids, double *pts, double *nrms) {
    extern __shared__ __align__(16) unsigned char smem[];
    SceneView sv = gs;
    if (MODE == SCENE_LDS) { sv = stage_scene(gs, smem); __syncthreads(); }
    long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)(y1 - y0) * fp.width;
    if (t >= total) return;
    int x = (int)(t % fp.width), y = y0 + (int)(t / fp.width);
    uint32_t pixel = (uint32_t)x + (uint32_t)y * (uint32_t)fp.width;
    V3 ray = primary_ray(fp.width, fp.height, fp.cam, fp.K, pixel, x, y);
    double dt = draw(item_key(fp.K, (uint64_t)pixel * 32u + 1u), 0);
    V3 res = {0, 0, 0}, nrm = {0, 0, 0};
    int obj = scene_intersect<MODE>(sv, fp.cam, ray, -1, dt, res, nrm);
    ids[t] = obj;
    if (pts) { pts[3 * t] = res.x; pts[3 * t + 1] = res.y; pts[3 * t + 2] = res.z; }
    if (nrms) { nrms[3 * t] = nrm.x; nrms[3 * t + 1] = nrm.y; nrms[3 * t + 2] = nrm.z; }
}

// node table of a bounced pixel: 16 entries (node ids 1..15), two words each
//   a: 0 = not visited, 1 = constant colour in b, 2 = object (obj id in bits 8..) with absorb slot in b (0xFFFFFFFF = none)
struct NodeEntry { uint32_t a, b; };

template <int MODE>
__global__ __launch_bounds__(256) void k_secondary(SceneView gs, FrameParams fp, Counters *ctr, SPQueue q, const uint32_t *complex_list,
                                                    uint32_t c_cap, NodeEntry *tab) {
    extern __shared__ __align__(16) unsigned char smem[];
    SceneView sv = gs;
    if (MODE == SCENE_LDS) { sv = stage_scene(gs, smem); __syncthreads(); }
    // launched for the capacity of the node tables; the number of bounced pixels is on the device
    uint32_t n_complex = ctr->n_complex;
    if (n_complex > c_cap) { if (threadIdx.x == 0 && blockIdx.x == 0) atomicOr(&ctr->overflow, (uint32_t)OVF_COMPLEX); n_complex = c_cap; }
    uint32_t nsec = 0;
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < n_complex; c += gridDim.x * blockDim.x) {
        uint32_t pixel = complex_list[c];
        int x = (int)(pixel % (uint32_t)fp.width), y = (int)(pixel / (uint32_t)fp.width);
        NodeEntry *my = tab + (size_t)c * 16;
        for (int i = 0; i < 16; i++) { my[i].a = 0; my[i].b = 0; }

        struct Frame { V3 hit, nrm; int obj; int stage; };
        Frame fr[4];
        V3 src = fp.cam;
        V3 ray = primary_ray(fp.width, fp.height, fp.cam, fp.K, pixel, x, y);
        int depth = 0, exclude = -1;
        uint32_t node = 1;
        for (;;) {
            // ---- enter traceRay(src, ray, exclude, bounce = depth) : src/Raytrace.cpp:32-47
            bool returned = false;
            if (depth > 3) returned = true;   // bounce > MAX_BOUNCES: opaque black, no draw, no table entry
            else {
                double dt = draw(item_key(fp.K, (uint64_t)pixel * 32u + node), 0);
                if (depth > 0) nsec++;
                V3 res = {0, 0, 0}, nrm = {0, 0, 0};
                int obj = scene_intersect<MODE>(sv, src, ray, exclude, dt, res, nrm);
                if (obj < 0) { my[node].a = 1; my[node].b = 0xFF000000u; returned = true; }
                else if (is_light_patch(res)) { my[node].a = 1; my[node].b = light_colour(fp.light); returned = true; }
                else { fr[depth].hit = res; fr[depth].nrm = nrm; fr[depth].obj = obj; fr[depth].stage = 0; }
            }
            if (returned) { depth--; node >>= 1; }
            // ---- resume frames until one of them makes a recursive call
            bool descend = false;
            while (depth >= 0 && !descend) {
                Frame &f = fr[depth];
                int obj = f.obj;
                if (f.stage == 0) {
                    f.stage = 1;
                    if (sv.f(G_TRANSMISSION, obj) > 0) {   // refract, src/Raytrace.cpp:52-71
                        double n = 1 / sv.f(G_REFRACTION, obj);
                        double d = f.nrm.x * ray.x + f.nrm.y * ray.y + f.nrm.z * ray.z;
                        V3 n1 = {f.nrm.x * d, f.nrm.y * d, f.nrm.z * d};
                        n1.x = ray.x - n1.x; n1.y = ray.y - n1.y; n1.z = ray.z - n1.z;
                        n1.x *= n; n1.y *= n; n1.z *= n;
                        double s = 1 - (n * n) * (1 - d * d);
                        double sq = sqrt(s);
                        V3 n2 = {f.nrm.x * sq, f.nrm.y * sq, f.nrm.z * sq};
                        n1.x += -n2.x; n1.y += -n2.y; n1.z += -n2.z;
                        src.x = f.hit.x + n1.x * 0.01; src.y = f.hit.y + n1.y * 0.01; src.z = f.hit.z + n1.z * 0.01;
                        ray = n1;
                        normalize(ray);
                        exclude = -1; node = node * 2; depth++; descend = true;
                    }
                } else if (f.stage == 1) {
                    f.stage = 2;
                    if (sv.f(G_SPECULAR, obj) > 0) {       // reflect whatever `ray` now holds, src/Raytrace.cpp:73-82
                        V3 n1 = f.nrm;
                        double k = n1.x * ray.x + n1.y * ray.y + n1.z * ray.z;
                        n1.x *= k; n1.y *= k; n1.z *= k;
                        n1.x *= 2; n1.y *= 2; n1.z *= 2;
                        src = f.hit;
                        ray.x = ray.x - n1.x; ray.y = ray.y - n1.y; ray.z = ray.z - n1.z;
                        normalize(ray);
                        exclude = obj; node = node * 2 + 1; depth++; descend = true;
                    }
                } else {
                    uint32_t slot = 0xFFFFFFFFu;
                    if (sv.f(G_ABSORB, obj) > 0) {          // absorb branch is evaluated later, from the queue
                        // (the lanes that are here together take their slots with one atomic)
                        const unsigned long long here = __ballot(true);
                        const int first = __ffsll((long long)here) - 1;
                        uint32_t base = 0;
                        if ((int)(threadIdx.x & 63) == first) base = atomicAdd(&ctr->n_sp, (uint32_t)__popcll(here));
                        slot = (uint32_t)__shfl((int)base, first) + (uint32_t)__popcll(here & ((1ULL << (threadIdx.x & 63)) - 1ULL));
                        sp_store(q, slot, f.hit, f.nrm, obj, pixel, node | NODE_COMPLEX, c * 16 + node);
                    }
                    my[node].a = 2u | ((uint32_t)obj << 8);
                    my[node].b = slot;
                    depth--; node >>= 1;
                }
            }
            if (depth < 0) break;
        }
    }
    for (int sh = 1; sh < 64; sh <<= 1) nsec += (uint32_t)__shfl_xor((int)nsec, sh);
    if ((threadIdx.x & 63) == 0 && nsec) atomicAdd(&ctr->n_secondary, nsec);
}

// shadow loop + direct term, src/Raytrace.cpp:174-212
template <int MODE>
__global__ __launch_bounds__(256) void k_shadow(SceneView gs, FrameParams fp, SPQueue q, const Counters *ctr) {
    extern __shared__ __align__(16) unsigned char smem[];
    SceneView sv = gs;
    if (MODE == SCENE_LDS) { sv = stage_scene(gs, smem); __syncthreads(); }
    const uint32_t n_sp = sp_total(ctr, q.cap);
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n_sp; s += gridDim.x * blockDim.x) {
        V3 p = {q.geo[0 * (size_t)q.cap + s], q.geo[1 * (size_t)q.cap + s], q.geo[2 * (size_t)q.cap + s]};
        V3 nrm = {q.geo[3 * (size_t)q.cap + s], q.geo[4 * (size_t)q.cap + s], q.geo[5 * (size_t)q.cap + s]};
        int obj = q.obj[s];
        uint64_t IK = item_key(fp.K, (uint64_t)q.pixel[s] * 32u + (q.node[s] & 31u));
        int light_count = 0;
        V3 shadow_ray = {0, 0, 0};
        for (int i = 0; i < 25; i++) {
            double x0 = draw(IK, 1 + 3 * i) * 2 - 1;
            double z0 = draw(IK, 2 + 3 * i) * 2 + 3;
            double y0 = 4.95;
            shadow_ray.x = x0 - p.x; shadow_ray.y = y0 - p.y; shadow_ray.z = z0 - p.z;
            double max_dist = shadow_ray.x * shadow_ray.x + shadow_ray.y * shadow_ray.y + shadow_ray.z * shadow_ray.z;
            normalize(shadow_ray);
            double dt = draw(IK, 3 + 3 * i);
            if (shadow_blocked<MODE>(sv, p, shadow_ray, dt, obj, max_dist)) light_count++;
        }
        double direct = 0;
        if (light_count < 25) {
            double d = nrm.x * shadow_ray.x + nrm.y * shadow_ray.y + nrm.z * shadow_ray.z;
            if (d > 0) direct += d * 0.2 * (1 - (light_count / (double)25));
            // the Blinn term at src/Raytrace.cpp:215-227 only feeds a variable that is never read
        }
        if (fp.flags & 4) direct = 0;
        q.direct[s] = direct;
    }
}

// ---- spatial order for the shade points of bounced pixels.  They come out of k_secondary in pixel order of the bounce
// trees, which says nothing about where they lie; sorted along a Morton curve over their own bounds, every 64 consecutive
// ones are neighbours on some surface and are gathered as a group like the shade points of a pixel tile.
__device__ inline uint32_t sp_f2ord(float f) { uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ inline float sp_ord2f(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }
// slot of the i-th shade point that is sorted: the bounced pixels' points first, then the listed points of spread-out tiles
// (clamped to the queue: after an overflow -- the frame is rendered again -- a listed slot may lie beyond it or not have been written)
__device__ inline uint32_t sp_sorted_slot(const Counters *ctr, const uint32_t *loose, uint32_t i, uint32_t cap) {
    return min(i < ctr->n_bounced ? ctr->n_sp0 + i : loose[i - ctr->n_bounced], cap - 1u);
}
__global__ __launch_bounds__(256) void k_sp_bounds(SPQueue q, const Counters *ctr, const uint32_t *loose, uint32_t *bounds) {
    __shared__ uint32_t part[4][6];
    const uint32_t n = ctr->n_sorted;
    uint32_t mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[3] = {0, 0, 0};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t slot = sp_sorted_slot(ctr, loose, i, q.cap);
        for (int a = 0; a < 3; a++) {
            uint32_t v = sp_f2ord((float)q.geo[a * (size_t)q.cap + slot]);
            mn[a] = min(mn[a], v); mx[a] = max(mx[a], v);
        }
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
__global__ __launch_bounds__(256) void k_sp_morton(SPQueue q, const Counters *ctr, const uint32_t *loose, const uint32_t *bounds, uint64_t *keys, uint32_t *vals) {
    const uint32_t n = ctr->n_sorted;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t slot = sp_sorted_slot(ctr, loose, i, q.cap);
    float ext = 0.f;
    for (int a = 0; a < 3; a++) ext = fmaxf(ext, sp_ord2f(bounds[3 + a]) - sp_ord2f(bounds[a]));
    uint32_t qv[3];
    for (int a = 0; a < 3; a++) qv[a] = quant21((float)q.geo[a * (size_t)q.cap + slot], sp_ord2f(bounds[a]), ext);
    keys[i] = morton63(qv[0], qv[1], qv[2]);
    vals[i] = slot;
}

// Groups of the sorted shade points.  k_sp_pieces: the runs that share an octree cell of about `cell_rel` times the extent
// of their bounds (a few 63-NN radii), cut into pieces of at most 64; one thread per run (the thread of its first point;
// the end of the run by binary search).  k_sp_classify: one wave per piece.  A piece goes to the group kernel when a shared
// candidate set is likely to pay: enough points, and an expected set size (the group kernel's own estimate, with the mean
// photon density in place of a centre query) within `est_max`.  The others go to the `early` list: the per-query kernel
// takes them on a stream of its own right away, beside the group kernels, instead of after them.
__global__ __launch_bounds__(256) void k_sp_pieces(const uint64_t *keys, double cell_rel, uint2 *pieces, Counters *ctr) {
    const uint32_t n = ctr->n_sorted;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int level = 1;
    while (level < 10 && 1.0 / (double)(1 << (level + 1)) >= cell_rel) level++;
    const int shift = 63 - 3 * level;
    const uint64_t cell = keys[i] >> shift;
    if (i > 0 && (keys[i - 1] >> shift) == cell) return;
    uint32_t lo = i + 1, hi = n;                      // first index beyond the run
    while (lo < hi) { uint32_t mid = lo + (hi - lo) / 2; if ((keys[mid] >> shift) == cell) lo = mid + 1; else hi = mid; }
    const uint32_t cnt = lo - i, np = (cnt + 63u) / 64u, per = (cnt + np - 1u) / np;
    uint32_t g = atomicAdd(&ctr->n_pieces, np);
    for (uint32_t c = 0; c < cnt; c += per, g++) pieces[g] = make_uint2(i + c, min(per, cnt - c));
}
__global__ __launch_bounds__(256) void k_sp_classify(SPQueue q, const uint32_t *sorted, const uint2 *pieces, const uint32_t *bounds, double photons_per_ext2,
                                                      double est_max, uint32_t min_run, uint2 *sgroups, Counters *ctr, uint32_t *early) {
    const int lane = threadIdx.x & 63;
    const uint32_t n_pieces = ctr->n_pieces, waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < n_pieces; w += waves) {
        const uint2 pc = pieces[w];
        const uint32_t s = sorted[pc.x + min((uint32_t)lane, pc.y - 1u)];      // idle lanes repeat the last point
        bool grouped = pc.y >= min_run;
        if (grouped) {
            float ext = 0.f;
            for (int a = 0; a < 3; a++) ext = fmaxf(ext, sp_ord2f(bounds[3 + a]) - sp_ord2f(bounds[a]));
            const double rho = photons_per_ext2 / fmax((double)ext * (double)ext, 1e-300);   // photons per unit of surface
            const double rad = sqrt(PM_TILE_BETA2 * 63.0 / (3.14159265 * rho));
            double e[3];
            for (int a = 0; a < 3; a++) { double v = q.geo[a * (size_t)q.cap + s]; e[a] = wave_max_f64(v) - wave_min_f64(v) + 2 * rad; }
            grouped = e[0] * e[1] * e[2] / fmin(e[0], fmin(e[1], e[2])) * rho <= est_max;
        }
        if (grouped) { if (lane == 0) sgroups[atomicAdd(&ctr->n_sgroups, 1u)] = pc; }
        else {
            uint32_t b0 = 0;
            if (lane == 0) b0 = atomicAdd(&ctr->n_early, pc.y);
            b0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)b0);
            if ((uint32_t)lane < pc.y) early[b0 + (uint32_t)lane] = s;
        }
    }
}

// The same loop with one shadow RAY per lane (10 shade points x 25 rays per 256-thread block): the launch that is used.
// The work is cut 25 times finer, so every SIMD stays busy until the end and the pass shares the device better with the
// map builds beside it (default scene: the pass itself 0.24 -> 0.28 ms, the step 3.10 -> 3.05 ms; 10 000 spheres, where a
// ray ends at its first blocker or walks cells all the way to the light: step 13.3 -> 11.8 ms).
template <int MODE, bool COUNT, bool QUEUED>
__device__ inline void shadow_rays_body(const SceneView &gs, const FrameParams &fp, const SPQueue &q, Counters *ctr) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int blocked[10];
    constexpr bool WAVE_WALK = MODE == SCENE_GRID && QUEUED;     // the wave's rays share a queue of tests (pm_trace.h)
    __shared__ TraceBuf walk[WAVE_WALK ? 4 : 1];
    TestCount tc = {0u, 0u, 0u, 0u, 0u};
    SceneView sv = gs;
    if (MODE == SCENE_LDS) { sv = stage_scene(gs, smem); }
    const uint32_t n_sp = sp_total(ctr, q.cap);      // launched for the queue's capacity: blocks take groups of 10 shade points in turn
    const uint32_t local = threadIdx.x / 25u, i = threadIdx.x % 25u;
    for (uint32_t g = blockIdx.x; g * 10u < n_sp; g += gridDim.x) {
        if (threadIdx.x < 10) blocked[threadIdx.x] = 0;
        __syncthreads();
        const uint32_t s = g * 10u + local;
        const bool live = threadIdx.x < 250u && s < n_sp;
        V3 p = {0, 0, 0}, nrm = {0, 0, 0}, shadow_ray = {0, 0, 0};
        double dt = 0, max_dist = 0;
        int obj = -1;
        if (live) {
            p = {q.geo[0 * (size_t)q.cap + s], q.geo[1 * (size_t)q.cap + s], q.geo[2 * (size_t)q.cap + s]};
            obj = q.obj[s];
            const uint64_t IK = item_key(fp.K, (uint64_t)q.pixel[s] * 32u + (q.node[s] & 31u));
            const double x0 = draw(IK, 1 + 3 * i) * 2 - 1;
            const double z0 = draw(IK, 2 + 3 * i) * 2 + 3;
            const double y0 = 4.95;
            shadow_ray.x = x0 - p.x; shadow_ray.y = y0 - p.y; shadow_ray.z = z0 - p.z;
            max_dist = shadow_ray.x * shadow_ray.x + shadow_ray.y * shadow_ray.y + shadow_ray.z * shadow_ray.z;
            normalize(shadow_ray);
            dt = draw(IK, 3 + 3 * i);
        }
        bool dark;
        if (WAVE_WALK) {
            WaveTracer<true, false> tr(sv, walk[threadIdx.x >> 6], COUNT ? &tc : nullptr);
            tr.start(live, p, shadow_ray, dt, obj, max_dist);
            while (__ballot(tr.pending && !tr.finished()) != 0ULL) tr.pump();
            dark = live && tr.blocked();
        } else dark = live && shadow_blocked<MODE>(sv, p, shadow_ray, dt, obj, max_dist, COUNT ? &tc : nullptr);
        if (dark) atomicAdd(&blocked[local], 1);
        __syncthreads();
        if (live && i == 24u) {        // the direct term uses the direction of the LAST shadow ray, src/Raytrace.cpp:207-212
            nrm = {q.geo[3 * (size_t)q.cap + s], q.geo[4 * (size_t)q.cap + s], q.geo[5 * (size_t)q.cap + s]};
            const int light_count = blocked[local];
            double direct = 0;
            if (light_count < 25) {
                double d = nrm.x * shadow_ray.x + nrm.y * shadow_ray.y + nrm.z * shadow_ray.z;
                if (d > 0) direct += d * 0.2 * (1 - (light_count / (double)25));
            }
            if (fp.flags & 4) direct = 0;
            q.direct[s] = direct;
        }
        __syncthreads();               // `blocked` is cleared for the next group
    }
    if (COUNT) { wave_add_count(&ctr->tests[2], tc.spheres); wave_add_count(&ctr->tests[3], tc.planes); wave_add_count(&ctr->walk[0], tc.cells); wave_add_count(&ctr->walk[1], tc.wsteps); wave_add_count(&ctr->walk[2], tc.wtests); }
}
template <int MODE>
// (through the sphere grid the pass waits for memory 47 % of its wave cycles: at least six waves per SIMD -- 80 VGPRs, 10 dwords of
// scratch per lane, and as many as the tracer's 25 KB of LDS per block allow -- 7.28 -> 7.09 ms per step on 10 000 spheres; the
// passes of small scenes need 52 registers and are not affected)
#ifndef PM_SHADOW_WAVES
#define PM_SHADOW_WAVES 6
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PM_SHADOW_WAVES))) void k_shadow_rays(SceneView gs, FrameParams fp, SPQueue q, Counters *ctr) { shadow_rays_body<MODE, false, true>(gs, fp, q, ctr); }
template <int MODE>
__global__ __launch_bounds__(256) void k_shadow_rays_count(SceneView gs, FrameParams fp, SPQueue q, Counters *ctr) { shadow_rays_body<MODE, true, true>(gs, fp, q, ctr); }
// every ray walks and tests on its own (PM_SHADOW_QUEUE=0: A/B runs of the sphere grid's walk)
template <int MODE>
__global__ __launch_bounds__(256) void k_shadow_rays_lane(SceneView gs, FrameParams fp, SPQueue q, Counters *ctr) { shadow_rays_body<MODE, false, false>(gs, fp, q, ctr); }
template <int MODE>
__global__ __launch_bounds__(256) void k_shadow_rays_lane_count(SceneView gs, FrameParams fp, SPQueue q, Counters *ctr) { shadow_rays_body<MODE, true, false>(gs, fp, q, ctr); }

// One draw from a work counter per wave.  Out of line on purpose: inlined into the scheduling loop, the
// lane-0 branch was threaded into the loop's control flow and lanes 1..63 were sent round the loop on their own.
__device__ __noinline__ uint32_t wave_draw(uint32_t *ctr) {
    uint32_t g = 0;
    if ((threadIdx.x & 63) == 0) g = atomicAdd(ctr, 1u);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)g);
}

#ifndef PM_GATHER_GROUP
#define PM_GATHER_GROUP 2      // consecutive shade points (neighbours) per draw: the second starts from the first one's ball
#endif
#ifndef PM_SCHED_CHUNK
#define PM_SCHED_CHUNK 256
#endif
constexpr uint32_t SCHED_CHUNK = PM_SCHED_CHUNK;   // items per chunk; chunk c belongs to counter c % 64
constexpr uint32_t SCHED_STRIDE = 32;    // one counter per 128-byte line
constexpr uint32_t SCHED_WORDS = 64 * SCHED_STRIDE;
static_assert(SCHED_CHUNK % PM_GATHER_GROUP == 0, "a draw must not straddle two chunks");
// Work distribution of the gather kernels: items are cut into chunks of SCHED_CHUNK consecutive items (a strip of the
// image, so a compact part of the photon map); chunk c is served by counter c % 64 and counter k by the blocks of XCD
// k % 8 (blocks b, b+8, ... share an XCD and its L2).  A wave starts on one of its XCD's 8 counters and moves to the next
// when that runs dry, so a cluster of expensive items never becomes one wave's tail and no counter sees more than a few
// thousand draws.  Returns false when all 8 counters are exhausted; else [i0, i1) is a run of up to `group` items.
// Chunk sizes (round 3): an XCD never draws from another XCD's counters, so with chunks of 1 024 groups a 1440 x 1440 frame
// (8 126 groups) gave every XCD ONE strip of the image, and the strips differ in cost by up to 60 % (a group of the caustic
// map: 221 us in the top strip, 138 us in the seventh).  128 groups per chunk (PM_TILE_CHUNK) deals eight strips to every
// XCD: default scene 2.23 -> 2.19 ms per step, five glass spheres 2.63 -> 2.53, config 3's sizes 35.0 -> 34.6; 256 items for
// the per-query kernels (PM_SCHED_CHUNK) another 0.01.  32 groups per chunk is slower again (an XCD's L2 then sees the whole map).
struct WorkDraw {
    uint32_t xcd, sub, tries;
    __device__ WorkDraw() : xcd(blockIdx.x & 7), sub((blockIdx.x >> 3) & 7), tries(0) {}
    __device__ bool next(uint32_t *sched, uint32_t n_items, uint32_t group, uint32_t &i0, uint32_t &i1, const uint32_t chunk = SCHED_CHUNK) {   // items 0 .. n_items
        while (tries < 8) {
            const uint32_t c = xcd + 8 * sub;
            uint32_t g = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_draw(&sched[c * SCHED_STRIDE]));
            uint32_t li = g * group;
            uint64_t s0 = ((uint64_t)c + 64ull * (li / chunk)) * chunk + li % chunk;
            if (s0 >= n_items) { tries++; sub = (sub + 1) & 7; continue; }   // this counter is exhausted (s0 grows with every draw)
            i0 = (uint32_t)s0;
            i1 = i0 + group < n_items ? i0 + group : n_items;
            return true;
        }
        return false;
    }
};

// One wave per shade point (pm_grid.h).  Items are the shade points the group kernel left: `left[0 .. *n_left)`, then the
// slots [sec_lo, sec_lo + sec_n) of the bounced pixels' shade points, which come in no spatial order at all.
// sec_mode: which slots follow the list: 0 none, 1 the bounced pixels' shade points [n_sp0, n_sp0 + n_bounced), 2 every shade point
template <int KIND>
__global__ __launch_bounds__(256) void k_gather(MapView m_in, SPQueue q, const uint32_t *left, const uint32_t *n_left, const Counters *ctr, int sec_mode,
                                                uint32_t *sched) {
    MapView m = m_in;
    load_grid(m);
    __shared__ KnnBuf bufs[4];
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform, and the compiler should know
    KnnBuf *buf = &bufs[wave_in_block];
    int lane = threadIdx.x & 63;
    double *out = KIND == 0 ? q.rad_g : q.rad_c;
    const uint32_t sec_lo = sec_mode == 1 ? ctr->n_sp0 : 0u;
    const uint32_t sec_n = sec_mode == 1 ? ctr->n_bounced : (sec_mode == 2 ? sp_total(ctr, q.cap) : 0u);
    const uint32_t nl = *n_left, n_items = nl + sec_n;
    WorkDraw wd;
#ifdef PM_WAVE_TIMES
    unsigned long long wt0 = __builtin_amdgcn_s_memtime(); uint32_t wq = 0;
#endif
    uint32_t i0, i1;
    while (wd.next(sched, n_items, (uint32_t)PM_GATHER_GROUP, i0, i1)) {
        KnnCarry carry;
        carry.full = false;
        for (uint32_t i = i0; i < i1; i++) {
            const uint32_t s = i < nl ? left[i] : sec_lo + (i - nl);
            V3 p = {q.geo[0 * (size_t)q.cap + s], q.geo[1 * (size_t)q.cap + s], q.geo[2 * (size_t)q.cap + s]};
            V3 nrm = {q.geo[3 * (size_t)q.cap + s], q.geo[4 * (size_t)q.cap + s], q.geo[5 * (size_t)q.cap + s]};
            double R, G, B;
            knn_gather<KIND>(buf, m, p, nrm, R, G, B, carry);
            if (lane == 0) { out[0 * (size_t)q.cap + s] = R; out[1 * (size_t)q.cap + s] = G; out[2 * (size_t)q.cap + s] = B; }
#ifdef PM_WAVE_TIMES
            wq++;
#endif
        }
    }
#ifdef PM_WAVE_TIMES
    // diagnostic build: per wave {start, end, queries} after the counters
    if (lane == 0) {
        uint32_t w = blockIdx.x * 4 + wave_in_block;
        unsigned long long wt1 = __builtin_amdgcn_s_memtime();
        uint32_t *o = sched + 4 * SCHED_WORDS + (size_t)KIND * 8192 * 4 + (size_t)w * 4;
        o[0] = (uint32_t)wt0; o[1] = (uint32_t)wt1; o[2] = wq; o[3] = wd.xcd;
    }
#endif
}

// The same items, two per wave (pm_pair.h): each half-wave answers one shade point with a block query; a shade point the block
// query cannot answer (far from the photons, a dense cluster next door) is then taken by the whole wave with the octree walk,
// right here.  The kernel the frame uses (PM_PAIR=0: k_gather, for A/B runs).
template <int KIND>
#ifndef PM_PAIR_WAVES
#define PM_PAIR_WAVES 4
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PM_PAIR_WAVES, PM_PAIR_WAVES))) void k_gather_pair(MapView m_in, SPQueue q, const uint32_t *left, const uint32_t *n_left, const Counters *ctr, int sec_mode,
                                                     uint32_t *sched) {
    MapView m = m_in;
    load_grid(m);
    union PairOrKnn { PairBuf pair; KnnBuf knn; };
    __shared__ PairOrKnn bufs[4];
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    PairOrKnn *buf = &bufs[wave_in_block];
    const int lane = threadIdx.x & 63, half = lane >> 5;
    double *out = KIND == 0 ? q.rad_g : q.rad_c;
    const uint32_t sec_lo = sec_mode == 1 ? ctr->n_sp0 : 0u;
    const uint32_t sec_n = sec_mode == 1 ? ctr->n_bounced : (sec_mode == 2 ? sp_total(ctr, q.cap) : 0u);
    const uint32_t nl = *n_left, n_items = nl + sec_n;
    WorkDraw wd;
    uint32_t i0, i1;
    while (wd.next(sched, n_items, 2u, i0, i1)) {
        const bool have = i0 + (uint32_t)half < i1;
        const uint32_t i = have ? i0 + (uint32_t)half : i0;
        const uint32_t s = i < nl ? left[i] : sec_lo + (i - nl);
        const V3 p = {q.geo[0 * (size_t)q.cap + s], q.geo[1 * (size_t)q.cap + s], q.geo[2 * (size_t)q.cap + s]};
        const V3 nrm = {q.geo[3 * (size_t)q.cap + s], q.geo[4 * (size_t)q.cap + s], q.geo[5 * (size_t)q.cap + s]};
        double R, G, B;
        const bool ok = pair_gather<KIND>(&buf->pair, m, p, nrm, have, R, G, B);
#ifdef PM_PAIR_STAMPS
        if (lane == 0) atomicAdd(&g_pair_ticks[7], 1ULL);
#endif
        if (ok && (lane & 31) == 0) { out[0 * (size_t)q.cap + s] = R; out[1 * (size_t)q.cap + s] = G; out[2 * (size_t)q.cap + s] = B; }
#ifdef PM_PAIR_STATS
        if ((lane & 31) == 0 && have) { atomicAdd(const_cast<uint32_t *>(&ctr->ghist[KIND][0]), 1u); if (ok) atomicAdd(const_cast<uint32_t *>(&ctr->ghist[KIND][1]), 1u); }
#endif
        // what a half could not answer: the whole wave, with the walk
        unsigned long long todo = __ballot(have && !ok && (lane & 31) == 0);
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const uint32_t s2 = (uint32_t)__shfl((int)s, src);
            const V3 p2 = {__shfl(p.x, src), __shfl(p.y, src), __shfl(p.z, src)}, n2 = {__shfl(nrm.x, src), __shfl(nrm.y, src), __shfl(nrm.z, src)};
            KnnCarry carry;
            carry.full = false;
            double R2, G2, B2;
            wave_lds_fence();
#ifdef PM_PAIR_STAMPS
            const unsigned long long tw0_ = __builtin_amdgcn_s_memtime();
#endif
            knn_gather<KIND>(&buf->knn, m, p2, n2, R2, G2, B2, carry);
#ifdef PM_PAIR_STAMPS
            if (lane == 0) { atomicAdd(&g_pair_ticks[5], __builtin_amdgcn_s_memtime() - tw0_); atomicAdd(&g_pair_ticks[6], 1ULL); }
#endif
            if (lane == 0) { out[0 * (size_t)q.cap + s2] = R2; out[1 * (size_t)q.cap + s2] = G2; out[2 * (size_t)q.cap + s2] = B2; }
            wave_lds_fence();
        }
    }
}

#ifdef PM_WAVE_TIMES
// diagnostics (builds with -DPM_WAVE_TIMES only): per wave of the latest k_gather_tile launch of each map {first tick, last tick, groups, XCD}
static __device__ uint32_t g_tile_waves[2][8192][4];
static __device__ uint32_t g_tile_groups[2][16384][2];     // per group {start, end} on the device-wide clock
#endif
#ifndef PM_TILE_WAVES
#define PM_TILE_WAVES 4
#endif
#ifndef PM_TILE_CHUNK
#define PM_TILE_CHUNK 128
#endif
// One wave per GROUP of shade points (pm_tile.h): the shade points of one 8x8 pixel tile, one per lane.  A group that
// cannot be gathered as a group appends its slots to `left` for k_gather.
template <int KIND>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PM_TILE_WAVES, PM_TILE_WAVES))) void k_gather_tile(MapView m_in, SPQueue q, const uint2 *groups, const Counters *ctr, uint32_t *left, uint32_t *n_left,
                                                    uint32_t *sched, float4 *strips, int dbg, const uint32_t *sorted, const uint2 *sgroups, uint32_t part, uint32_t nparts) {
    MapView m = m_in;
    load_grid(m);
    __shared__ TileBuf tb;
    const int lane = threadIdx.x;
    const Strip S(reinterpret_cast<float *>(strips + (size_t)blockIdx.x * TS_STRIDE));   // this wave's strip for the shared candidate set
    double *out = KIND == 0 ? q.rad_g : q.rad_c;
    // the groups of the pixel tiles, then the groups of `sorted` (shade points of bounced pixels in spatial order)
    // The groups are cut into `nparts` launches: what one part leaves is gathered per query (on another stream) while the next
    // part runs, so that only the last part's leftovers are gathered with nothing beside them.
    const uint32_t n_tiles = ctr->n_groups, n_groups = n_tiles + ctr->n_sgroups;
    const uint32_t g_lo = (uint32_t)((unsigned long long)n_groups * part / nparts), g_hi = (uint32_t)((unsigned long long)n_groups * (part + 1u) / nparts);
    WorkDraw wd;
    uint32_t i0, i1;
#ifdef PM_WAVE_TIMES
    const unsigned long long wt0 = __builtin_amdgcn_s_memrealtime();      // (the device-wide 100 MHz clock: s_memtime counts per XCD)
    uint32_t wgroups = 0;
    if (lane == 0 && blockIdx.x < 8192) { uint32_t *o = g_tile_waves[KIND][blockIdx.x]; o[0] = (uint32_t)wt0; o[1] = (uint32_t)wt0; o[2] = 0; o[3] = wd.xcd | ((uint32_t)__builtin_amdgcn_s_getreg(20 | (3 << 11)) << 8); }   // (hardware register 20: XCC_ID, 4 bits)
#endif
    while (wd.next(sched, g_hi - g_lo, 1u, i0, i1, (uint32_t)PM_TILE_CHUNK)) {
        i0 += g_lo;
#ifdef PM_WAVE_TIMES
        const uint32_t gt0 = (uint32_t)__builtin_amdgcn_s_memrealtime();
        wgroups++;
        if (lane == 0 && blockIdx.x < 8192) { uint32_t *o = g_tile_waves[KIND][blockIdx.x]; o[2] = wgroups; }
#endif
#ifdef PM_TILE_STAMPS
        const unsigned long long tg0 = __builtin_amdgcn_s_memtime();
#endif
        bool active;
        uint32_t s;
        if (i0 < n_tiles) {
            const uint2 g = groups[i0];
            if (g.x >= q.cap) continue;                       // a queue that overflowed (the frame is rendered again): nothing of it is touched
            active = (uint32_t)lane < g.y && g.x + (uint32_t)lane < q.cap;
            s = g.x + (active ? (uint32_t)lane : 0u);
        } else {
            const uint2 g = sgroups[i0 - n_tiles];
            active = (uint32_t)lane < g.y;
            s = sorted[g.x + (active ? (uint32_t)lane : 0u)];
        }
        const V3 p = {q.geo[0 * (size_t)q.cap + s], q.geo[1 * (size_t)q.cap + s], q.geo[2 * (size_t)q.cap + s]};
        const V3 nrm = {q.geo[3 * (size_t)q.cap + s], q.geo[4 * (size_t)q.cap + s], q.geo[5 * (size_t)q.cap + s]};
        // dead: the lane goes to the per-query kernel (its group is too spread out for one shared set, or the lane itself
        // came up short)
        double R, G, B, r_scale = 0.0;
        int why = 0;
        const bool ok = tile_gather<KIND>(&tb, S, m, p, nrm, active, r_scale, R, G, B, why, const_cast<unsigned long long *>(ctr->tphase));
        if (ok) { out[0 * (size_t)q.cap + s] = R; out[1 * (size_t)q.cap + s] = G; out[2 * (size_t)q.cap + s] = B; }
        bool dead = active && !ok;
        if (dbg) {   // diagnostic: lanes handed over, by reason (7 = the lane's own shortfall)
            const int lost = __popcll(__ballot(dead));
            if (lane == 0 && lost) atomicAdd(const_cast<uint32_t *>(&ctr->why[why != 0 ? (why & 7) : 7]), (uint32_t)lost);
        }
#ifdef PM_TILE_STAMPS
        if (lane == 0) { unsigned long long dt = __builtin_amdgcn_s_memtime() - tg0; atomicAdd(const_cast<uint32_t *>(&ctr->ghist[KIND][63 - __clzll((long long)(dt | 1ULL))]), 1u); }
#endif
        const unsigned long long lm = __ballot(dead);
        if (lm) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(n_left, (uint32_t)__popcll(lm));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (dead) left[base + (uint32_t)lane_rank(lm)] = s;
        }
#ifdef PM_WAVE_TIMES
        if (lane == 0 && blockIdx.x < 8192) g_tile_waves[KIND][blockIdx.x][1] = (uint32_t)__builtin_amdgcn_s_memrealtime();
        if (lane == 0 && i0 < 16384) { g_tile_groups[KIND][i0][0] = gt0; g_tile_groups[KIND][i0][1] = (uint32_t)__builtin_amdgcn_s_memrealtime(); }
#endif
    }
}

// standalone gather on caller-provided points (pm_knn_gather): points/normals/radiance are [m][3]
template <int KIND>
__global__ __launch_bounds__(256) void k_gather_points(MapView m_in, const double *pts, const double *nrms, double *rad, long long count) {
    MapView m = m_in;
    load_grid(m);
    __shared__ KnnBuf bufs[4];
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    KnnBuf *buf = &bufs[wave_in_block];
    int lane = threadIdx.x & 63;
    long long xcd = blockIdx.x & 7, nx = gridDim.x >> 3;
    long long lo = (count * xcd) >> 3, hi = (count * (xcd + 1)) >> 3;
    long long wl = (blockIdx.x >> 3) * 4 + wave_in_block, wpx = nx * 4;
    for (long long s = lo + wl; s < hi; s += wpx) {
        V3 p = {pts[3 * s], pts[3 * s + 1], pts[3 * s + 2]};
        V3 nrm = {nrms[3 * s], nrms[3 * s + 1], nrms[3 * s + 2]};
        double R, G, B;
        KnnCarry carry;
        carry.full = false;
        knn_gather<KIND>(buf, m, p, nrm, R, G, B, carry);
        if (lane == 0) { rad[3 * s] = R; rad[3 * s + 1] = G; rad[3 * s + 2] = B; }
    }
}

// raw k-NN (pm_knn_query): idx = emission index, ascending (d2, idx)
__global__ __launch_bounds__(256) void k_knn_query(MapView m_in, const double *pts, long long count, int32_t *idx, double *d2, int32_t *found, uint32_t *work) {
    MapView m = m_in;
    load_grid(m);
    __shared__ KnnBuf bufs[4];
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    KnnBuf *buf = &bufs[wave_in_block];
    int lane = threadIdx.x & 63;
    long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + wave_in_block, waves_total = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long s = wave; s < count; s += waves_total) {
        V3 p = {pts[3 * s], pts[3 * s + 1], pts[3 * s + 2]};
        KnnState st;
        double r;
#ifdef PM_STAMPS
        unsigned long long tq0 = __builtin_amdgcn_s_memtime();
#endif
        int f = grid_query(st, buf, m, p, r, MAX_R2);
#ifdef PM_STAMPS
        // diagnostic build: the four work words carry cycles (total, node sections, leaf sections, reduces) instead
        unsigned long long tq1 = __builtin_amdgcn_s_memtime();
        st.c_nodes = (uint32_t)(tq1 - tq0); st.c_leaves = (uint32_t)st.t_node; st.c_reduces = (uint32_t)st.t_leaf; st.c_cands = (uint32_t)st.t_red;
#endif
        wave_lds_fence();
        Best b;
        b.d = __longlong_as_double(0x7FF0000000000000LL); b.o = 0xFFFFFFFFu; b.i = 0;
        if (lane < f) {
            b.d = __longlong_as_double((long long)buf->d[lane]);
            b.i = buf->i[lane];
            b.o = __float_as_uint(m.pos4[b.i].w);
        }
        bitonic_sort_asc(b, lane);
        if (lane < KNN_K) {
            bool ok = lane < f;
            idx[s * KNN_K + lane] = ok ? (int32_t)b.o : -1;
            d2[s * KNN_K + lane] = ok ? b.d : 0.0;
        }
        if (lane == 0) found[s] = f;
        if (work && lane == 0) { work[4 * s] = st.c_nodes; work[4 * s + 1] = st.c_leaves; work[4 * s + 2] = st.c_reduces; work[4 * s + 3] = st.c_cands; }
        wave_lds_fence();
    }
}

// clamp / colour / pack, src/Raytrace.cpp:230-242 ; simple pixels also get their final mix (:245-257)
__device__ inline void shade_point(const SceneView &sv, const FrameParams &fp, const SPQueue &q, uint32_t s, uint32_t *pane, NodeEntry *tab) {
    int obj = q.obj[s];
    double direct = q.direct[s];
    double gr = 0, gg = 0, gb = 0, cr = 0, cg = 0, cb = 0;
    if (!(fp.flags & 1)) { gr = q.rad_g[0 * (size_t)q.cap + s]; gg = q.rad_g[1 * (size_t)q.cap + s]; gb = q.rad_g[2 * (size_t)q.cap + s]; }
    if (!(fp.flags & 2)) { cr = q.rad_c[0 * (size_t)q.cap + s]; cg = q.rad_c[1 * (size_t)q.cap + s]; cb = q.rad_c[2 * (size_t)q.cap + s]; }
    double rr = direct + cr + gr; if (rr > 1) rr = 1;
    double rg = direct + cg + gg; if (rg > 1) rg = 1;
    double rb = direct + cb + gb; if (rb > 1) rb = 1;
    uint32_t red = (uint32_t)fastfloor((double)sv.rgb[0 * sv.stride + obj] * rr * 0xFF);
    uint32_t green = (uint32_t)fastfloor((double)sv.rgb[1 * sv.stride + obj] * rg * 0xFF);
    uint32_t blue = (uint32_t)fastfloor((double)sv.rgb[2 * sv.stride + obj] * rb * 0xFF);
    uint32_t absorb_res = (0xFFu << 24) | (red << 16) | (green << 8) | blue;
    q.color[s] = absorb_res;
    uint32_t node = q.node[s];
    if (node & NODE_COMPLEX) tab[q.tabslot[s]].b = absorb_res;   // slot -> colour, consumed by k_resolve
    else pane[q.pixel[s]] = mix_colour(0, 0, absorb_res, sv.f(G_SPECULAR, obj), sv.f(G_TRANSMISSION, obj), sv.f(G_ABSORB, obj));
}
template <int MODE>
__global__ __launch_bounds__(256) void k_shade(SceneView gs, FrameParams fp, SPQueue q, const Counters *ctr, uint32_t *pane, NodeEntry *tab) {
    extern __shared__ __align__(16) unsigned char smem[];
    SceneView sv = gs;
    if (MODE == SCENE_LDS) { sv = stage_scene(gs, smem); __syncthreads(); }
    const uint32_t n_sp = sp_total(ctr, q.cap);
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < n_sp; s += gridDim.x * blockDim.x) shade_point(sv, fp, q, s, pane, tab);
}

// folds the node table of one bounced pixel bottom-up into its final colour (the mixes of traceRay's returns, src/Raytrace.cpp:245-257)
__device__ inline uint32_t resolve_pixel(const SceneView &sv, const NodeEntry *my) {
    uint32_t col[32];
#pragma unroll
    for (int i = 0; i < 32; i++) col[i] = 0;   // a branch that was not taken contributes 0 (refract_res / reflect_res initial value)
#pragma unroll
    for (int node = 15; node >= 1; node--) {
        NodeEntry e = my[node];
        uint32_t v = 0;
        if (e.a == 1) v = e.b;
        else if ((e.a & 0xFF) == 2) {
            int obj = (int)(e.a >> 8);
            double ks = sv.f(G_SPECULAR, obj), kt = sv.f(G_TRANSMISSION, obj), ka = sv.f(G_ABSORB, obj);
            // children at depth 4 return opaque black (src/Raytrace.cpp:33-35); channels are 0 either way
            uint32_t refract_res = (kt > 0) ? (node < 8 ? col[2 * node] : 0xFF000000u) : 0u;
            uint32_t reflect_res = (ks > 0) ? (node < 8 ? col[2 * node + 1] : 0xFF000000u) : 0u;
            uint32_t absorb_res = (ka > 0) ? e.b : 0u;
            v = mix_colour(reflect_res, refract_res, absorb_res, ks, kt, ka);
        }
        col[node] = v;
    }
    return col[1];
}
template <int MODE>
__global__ __launch_bounds__(256) void k_resolve(SceneView gs, const uint32_t *complex_list, const Counters *ctr, uint32_t c_cap, const NodeEntry *tab, uint32_t *pane) {
    extern __shared__ __align__(16) unsigned char smem[];
    SceneView sv = gs;
    if (MODE == SCENE_LDS) { sv = stage_scene(gs, smem); __syncthreads(); }
    const uint32_t n_complex = min(ctr->n_complex, c_cap);
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < n_complex; c += gridDim.x * blockDim.x)
        pane[complex_list[c]] = resolve_pixel(sv, tab + (size_t)c * 16);
}

}  // namespace pm
