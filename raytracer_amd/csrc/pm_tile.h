// pm_tile.h -- the radiance gather for a GROUP of up to 64 neighbouring shade points (the shade points of one
// 8x8 pixel tile, or a run of bounced pixels' shade points that share an octree cell), one query per lane.  Same
// results, bit for bit, as one wave per query (pm_grid.h): the exact 63rd-smallest squared distance of every query
// and the canonical order-independent radiance sum.
//
// Neighbouring points on a wall search nearly the same photons, so the group shares the work that one wave per
// query repeats 64 times:
//   1. one ordinary 63-NN query (pm_grid.h) at the centre c of the group's bounding box gives the scale r_c; the group
//      shares the squared radius L = 1.6 r_c;
//   2. one octree range walk collects S = every photon closer than sqrt(L) to the bounding box (a few hundred
//      photons: every photon closer than sqrt(L) to any of the group's points) into the wave's strip of device
//      memory, as float positions relative to the box centre + photon indices;
//   3. every lane finds ITS exact 63rd distance in S in two passes over S (staged through LDS in chunks, packed FP32):
//      a 64-bin histogram of d2 over [0, L) brackets it (pass 1); the photons surely below the bracket are listed as
//      the lane's members and the few photons inside it are ranked by their exact double d2 (pass 2 + ranking), which
//      gives the exact r_q and the remaining members;
//   4. every lane sums the radiance terms of its own members in 64-bit fixed point.
// A lane with fewer than 63 photons inside the shared radius, or a whole group that is not coherent (points on different
// surfaces, far from every photon, |S| over the strip's capacity), is handed to the one-wave-per-query kernel through a
// list of shade point slots.
#pragma once
#include "pm_grid.h"

namespace pm {

#ifndef PM_TILE_BETA2
#define PM_TILE_BETA2 1.6     // shared squared radius of a group, in units of the centre query's 63rd squared distance
#endif
#ifndef PM_TILE_EST_MAX
#define PM_TILE_EST_MAX 1600   // expected size of the shared set beyond which a group is left to the per-query kernel: one wave
                              // would sit on it for many times the average while independent queries spread over the device
#endif
#ifndef PM_TILE_FENCE
#define PM_TILE_FENCE 1
#endif
#ifndef PM_TS_CAP
#define PM_TS_CAP 2048
#endif
constexpr int TS_CAP = PM_TS_CAP;  // candidates a group may share (16-bit indices)
constexpr int TS_BINS = 64;        // histogram bins per lane (8-bit counters, four per word; a lane whose bin overflows is handed over)
constexpr int TS_BINL = 8;         // photons of the crossing bin a lane can rank
#ifndef PM_TS_BATCH
#define PM_TS_BATCH 8
#endif
constexpr int TS_BATCH = PM_TS_BATCH;   // candidates fetched ahead in the selection passes
constexpr int TS_MROW = 68;        // member list row stride in BYTES: 17 words, conflict-free by row and by column
constexpr int TS_PAGE = 256;       // a member is stored as its candidate index modulo TS_PAGE (one byte); the lane remembers where its pages begin
static_assert(TS_CAP <= 8 * TS_PAGE, "eight pages of candidates: their first members fit two registers, a byte each");

// per-wave LDS of the selection; the shared set S itself lives in a per-wave strip of device memory (it is read
// front to back by all lanes at once, so every read is one broadcast cache line).
// 9.25 KB: sixteen waves per compute unit (the member lists were 16-bit entries once, 13.25 KB and eleven waves).
struct TileSel {
    uint32_t hist[(TS_BINS / 4) * 64];             // [word][lane]
    uint8_t member[64 * TS_MROW];                  // [lane][k]: candidate index & 255 of the lane's k-th member for sure, in ascending order
    uint16_t binl[TS_BINL * 64];                   // [k][lane]: the uncertain band; after the ranking, those of it that are members
};
struct alignas(16) TileBuf {
    union { KnnBuf knn; TileSel sel; };    // the centre query and the range walk are over before the selection starts
};

__device__ inline double wave_min_f64(double v) { for (int s = 1; s < 64; s <<= 1) v = fmin(v, __shfl_xor(v, s)); return v; }
__device__ inline double wave_max_f64(double v) { for (int s = 1; s < 64; s <<= 1) v = fmax(v, __shfl_xor(v, s)); return v; }

// The shared set of one wave, in its strip of device memory: photon positions RELATIVE to the centre of the group's
// box, rounded to float (good to a few 1e-8: enough to sort photons into distance bins, never used for a result), and
// the photons' indices.  One array per component so that two neighbouring candidates load as a register pair.
constexpr int TS_CHUNK = 128;                  // candidates staged in LDS at a time by the selection passes
constexpr int TS_STRIDE = TS_CAP + TS_CHUNK;   // spare entries: the passes fetch whole chunks, the tail holds sentinels
struct Strip {
    float *x, *y, *z;
    uint32_t *idx;
    __device__ explicit Strip(float *base) : x(base), y(base + TS_STRIDE), z(base + 2 * TS_STRIDE), idx((uint32_t *)(base + 3 * TS_STRIDE)) {}
};

// Range walk: every photon whose squared distance to the box [qlo, qhi] is below lim goes to S.
// Returns the number collected, or -1 when there are more than TS_CAP.
__device__ inline int tile_collect(TileBuf *tb, const Strip &S, const MapView &m, const V3 &qlo, const V3 &qhi, double lim) {
    const int lane = threadIdx.x & 63;
    const V3 qc = {0.5 * (qlo.x + qhi.x), 0.5 * (qlo.y + qhi.y), 0.5 * (qlo.z + qhi.z)};
    const uint32_t qf[3] = {quant21((float)qc.x, m.glo[0], m.gext), quant21((float)qc.y, m.glo[1], m.gext), quant21((float)qc.z, m.glo[2], m.gext)};
    // the same exact cover of space as grid_query: the 7 siblings of the centre's cell at each table level + that cell
    const int L = m.lmax;
    int lev = 0;
    uint32_t cx = 0, cy = 0, cz = 0, first = 0, cnt = 0;
    double dmin2 = 0.0, dmax2 = 0.0;
    if (lane <= 7 * L) {
        int j = 0;
        if (lane == 7 * L) lev = L;
        else { lev = lane / 7 + 1; j = lane % 7 + 1; }
        const int sh = 21 - lev;
        cx = (lev ? qf[0] >> sh : 0u) ^ (uint32_t)(j & 1);
        cy = (lev ? qf[1] >> sh : 0u) ^ (uint32_t)((j >> 1) & 1);
        cz = (lev ? qf[2] >> sh : 0u) ^ (uint32_t)(j >> 2);
        uint64_t key = cell_key(lev, cx, cy, cz);
        cell_range(m, lev, key, 0u, 0u, first, cnt);
        if (cnt > 0) cell_d2(m, qlo, qhi, lev, cx, cy, cz, key, true, dmin2, dmax2);
    }
    int ns = 0;
    for (;;) {
        const bool alive = cnt > 0 && dmin2 < lim;
        if (!alive) cnt = 0;
        const unsigned long long am = __ballot(alive);
        if (am == 0ULL) break;
        const bool big = alive && cnt > (uint32_t)PM_GRID_SPLIT && lev < GRID_MAX_LEVEL;
        unsigned long long todo = __ballot(big);
        unsigned long long freem = ~am;
        if (todo && __popcll(freem) + 1 >= 8) {
            // split as many big cells as there are idle lanes for their children
            bool fresh = false;
            int plev = 0; uint32_t px = 0, py = 0, pz = 0, pfirst = 0, pcnt = 0; int cj = 0;
            while (todo) {
                int bl = __ffsll((long long)todo) - 1;
                unsigned long long fm = freem | (1ULL << bl);
                if (__popcll(fm) < 8) break;
                int j = lane_rank(fm);
                bool mine = ((fm >> lane) & 1ULL) && j < 8;
                int l_ = __builtin_amdgcn_readlane(lev, bl);
                uint32_t x_ = (uint32_t)__builtin_amdgcn_readlane((int)cx, bl), y_ = (uint32_t)__builtin_amdgcn_readlane((int)cy, bl),
                         z_ = (uint32_t)__builtin_amdgcn_readlane((int)cz, bl);
                uint32_t f_ = (uint32_t)__builtin_amdgcn_readlane((int)first, bl), c_ = (uint32_t)__builtin_amdgcn_readlane((int)cnt, bl);
                if (mine) { fresh = true; plev = l_; px = x_; py = y_; pz = z_; pfirst = f_; pcnt = c_; cj = j; }
                if (lane == bl && !mine) cnt = 0;
                freem = fm & ~__ballot(mine);
                todo &= ~(1ULL << bl);
            }
            if (fresh) {
                lev = plev + 1;
                cx = 2 * px + (uint32_t)(cj & 1); cy = 2 * py + (uint32_t)((cj >> 1) & 1); cz = 2 * pz + (uint32_t)(cj >> 2);
                uint64_t key = cell_key(lev, cx, cy, cz);
                cell_range(m, lev, key, pfirst, pfirst + pcnt, first, cnt);
                if (cnt > 0) cell_d2(m, qlo, qhi, lev, cx, cy, cz, key, lev <= m.lmax, dmin2, dmax2);
            }
            continue;
        }
        // read the small cells; with no room to split and no small cell left, one big cell as it is
        bool take = alive && !big;
        if (__ballot(take) == 0ULL) take = lane == __ffsll((long long)todo) - 1;
        uint32_t c = take ? cnt : 0u;
        uint32_t incl = wave_scan_incl_u32(c);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        uint32_t *pre = tb->knn.cellpre;
        unsigned long long nonempty = __ballot(c > 0);
        const int ncell = __popcll(nonempty);
        int k = lane_rank(nonempty);
        wave_lds_fence();
        pre[lane] = 0xFFFFFFFFu;
        wave_lds_fence();
        if (c > 0) { pre[k] = incl; pre[64 + k] = first - (incl - c); }
        wave_lds_fence();
        const int s0 = ncell <= 1 ? 0 : 1 << (31 - __clz(ncell - 1));
        for (uint32_t base = 0; base < total; base += 64) {
            uint32_t t = base + (uint32_t)lane;
            int j = 0;
            for (int step = s0; step > 0; step >>= 1)
                if (pre[j + step - 1] <= t) j += step;
            bool valid = t < total;
            uint32_t i = valid ? pre[64 + j] + t : 0u;
            float4 p = valid ? m.pos4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            double x = (double)p.x, y = (double)p.y, z = (double)p.z;
            double gx = x < qlo.x ? qlo.x - x : (x > qhi.x ? x - qhi.x : 0.0);
            double gy = y < qlo.y ? qlo.y - y : (y > qhi.y ? y - qhi.y : 0.0);
            double gz = z < qlo.z ? qlo.z - z : (z > qhi.z ? z - qhi.z : 0.0);
            bool pass = valid && gx * gx + gy * gy + gz * gz < lim;
            unsigned long long mask = __ballot(pass);
            int np = __popcll(mask);
            if (ns + np > TS_CAP) return -1;
            if (pass) {
                int at = ns + lane_rank(mask);
                S.x[at] = (float)(x - qc.x); S.y[at] = (float)(y - qc.y); S.z[at] = (float)(z - qc.z); S.idx[at] = i;
            }
            ns += np;
        }
        if (take) cnt = 0;
    }
    // far-away sentinels behind the last candidate: the passes work in whole batches and need no index test
#pragma unroll
    for (int u = 0; u < TS_CHUNK / 64; u++) S.x[ns + u * 64 + lane] = 3.0e18f;
    // The strip is read back through the vector caches by this same wave: its writes must have reached the L2 (the wave
    // waits for them) and the lines of the previous group's strip must be gone from the CU's vector cache (buffer_inv).
    // An agent-scope fence does both -- and, on gfx950, a write-back of the XCD's whole L2 before it (buffer_wbl2), once
    // per group: 2.72 -> 2.55 ms per step without that (PM_TILE_FENCE=0 brings the fence back for A/B runs; same panes).
#if PM_TILE_FENCE == 0
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
#else
    asm volatile("s_waitcnt vmcnt(0)\n\tbuffer_inv sc1" ::: "memory");
#endif
    wave_lds_fence();
    return ns;
}

__device__ inline double tile_d2(const float4 &s, const V3 &p) {   // src/PhotonMap.cpp:211-214, as in grid_scan
    double ddx = (double)s.x - p.x, ddy = (double)s.y - p.y, ddz = (double)s.z - p.z;
    return ddx * ddx + ddy * ddy + ddz * ddz;
}

// One pass over the shared set: fn(j, d2f) for every candidate j < ns (and for some sentinels behind them, whose d2f is
// astronomically large), d2f the float squared distance to this lane's point (relative coordinates).  The set is staged
// through LDS TS_CHUNK candidates at a time -- the next chunk is on its way from the L2 while this one is consumed --
// and read from there by all lanes at once (broadcast reads), TS_BATCH candidates per step.
// stage: 2 * 3 * TS_CHUNK floats of LDS not otherwise in use during the pass.
typedef float f2 __attribute__((ext_vector_type(2)));
template <class F, class C>
__device__ inline void tile_pass(float *stage, const Strip &S, int ns, float pxf, float pyf, float pzf, F fn, C chunk_fn) {
    static_assert(TS_CHUNK == 128 && TS_BATCH == 8, "two candidates per lane and chunk; batches of eight");
    const int lane = threadIdx.x & 63;
    const f2 px2 = {pxf, pxf}, py2 = {pyf, pyf}, pz2 = {pzf, pzf};
    const int nch = (ns + TS_CHUNK - 1) / TS_CHUNK;
    f2 rx = *reinterpret_cast<const f2 *>(S.x + 2 * lane), ry = *reinterpret_cast<const f2 *>(S.y + 2 * lane), rz = *reinterpret_cast<const f2 *>(S.z + 2 * lane);
    for (int c = 0; c < nch; c++) {
        float *b = stage + (c & 1) * 3 * TS_CHUNK;
        *reinterpret_cast<f2 *>(b + 2 * lane) = rx;
        *reinterpret_cast<f2 *>(b + TS_CHUNK + 2 * lane) = ry;
        *reinterpret_cast<f2 *>(b + 2 * TS_CHUNK + 2 * lane) = rz;
        wave_lds_fence();
        if (c + 1 < nch) {
            const int o = (c + 1) * TS_CHUNK + 2 * lane;
            rx = *reinterpret_cast<const f2 *>(S.x + o); ry = *reinterpret_cast<const f2 *>(S.y + o); rz = *reinterpret_cast<const f2 *>(S.z + o);
        }
        const int jn = min(ns - c * TS_CHUNK, TS_CHUNK);
        chunk_fn(c);
        for (int j0 = 0; j0 < jn; j0 += TS_BATCH) {
            const float4 xa = *reinterpret_cast<const float4 *>(b + j0), xb = *reinterpret_cast<const float4 *>(b + j0 + 4);
            const float4 ya = *reinterpret_cast<const float4 *>(b + TS_CHUNK + j0), yb = *reinterpret_cast<const float4 *>(b + TS_CHUNK + j0 + 4);
            const float4 za = *reinterpret_cast<const float4 *>(b + 2 * TS_CHUNK + j0), zb = *reinterpret_cast<const float4 *>(b + 2 * TS_CHUNK + j0 + 4);
            const f2 x[4] = {{xa.x, xa.y}, {xa.z, xa.w}, {xb.x, xb.y}, {xb.z, xb.w}};
            const f2 y[4] = {{ya.x, ya.y}, {ya.z, ya.w}, {yb.x, yb.y}, {yb.z, yb.w}};
            const f2 z[4] = {{za.x, za.y}, {za.z, za.w}, {zb.x, zb.y}, {zb.z, zb.w}};
            const int j = c * TS_CHUNK + j0;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                f2 dx = x[u] - px2, dy = y[u] - py2, dz = z[u] - pz2;
                f2 d2f = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
                fn(j + 2 * u, d2f[0]);
                fn(j + 2 * u + 1, d2f[1]);
            }
        }
    }
    wave_lds_fence();
}

// One group.  p, nrm: this lane's shade point; active: lane takes part (idle lanes must hold the point of
// some active lane, so that the bounding box is that of the active ones).  r_scale: the 63rd squared distance of a query
// in the neighbourhood, or <= 0 to have one made at the centre of the box (then returned).
// Returns, per lane, whether R, G, B hold the lane's radiance estimate.  why = 2 or 4 on return says that the group is
// too spread out for one shared set (nothing was computed): all of it goes to the per-query kernel.  A lane that
// gets false otherwise (fewer than 63 photons inside the shared radius, a crossing bin too full to rank) goes to the
// per-query kernel.
#ifdef PM_TILE_STAMPS
#define TS_LAP(k) { unsigned long long now_ = __builtin_amdgcn_s_memtime(); if (lane == 0) atomicAdd(&tphase[k], now_ - ts_); ts_ = now_; }
#else
#define TS_LAP(k)
#endif
template <int KIND>
__device__ inline bool tile_gather(TileBuf *tb, const Strip &S, const MapView &m, const V3 &p, const V3 &nrm, bool active, double &r_scale, double &R, double &G,
                                   double &B, int &why, unsigned long long *tphase) {
    const int lane = threadIdx.x & 63;
    R = G = B = 0.0;
    why = 0;
#ifdef PM_TILE_STAMPS
    unsigned long long ts_ = __builtin_amdgcn_s_memtime();
#endif
    if (m.n < KNN_K) { why = 1; return false; }
    // 1. bounding box; an ordinary query at its centre gives the scale: the shared radius is a little more than the
    //    centre's own 63-NN radius (photon density changes slowly across a tile)
    const V3 qlo = {wave_min_f64(p.x), wave_min_f64(p.y), wave_min_f64(p.z)};
    const V3 qhi = {wave_max_f64(p.x), wave_max_f64(p.y), wave_max_f64(p.z)};
    if (!(r_scale > 0.0)) {
        const V3 c = {0.5 * (qlo.x + qhi.x), 0.5 * (qlo.y + qhi.y), 0.5 * (qlo.z + qhi.z)};
        KnnState st;
        double r_c;
        int found = grid_query(st, &tb->knn, m, c, r_c, MAX_R2);
        if (found < KNN_K || !(r_c > 0.0)) { why = 1; return false; }
        r_scale = r_c;
    }
    TS_LAP(0)
    const double L = r_scale * PM_TILE_BETA2;
    if (!(L < MAX_R2)) { why = 3; return false; }   // the reference never looks beyond MAX_R2: leave that case to the per-query path
    {
        // photons lie on surfaces: the shared set is about the two largest extents of the box, each grown by the radius
        // on both sides, times the photon density 63 / (pi r)
        double e[3] = {qhi.x - qlo.x, qhi.y - qlo.y, qhi.z - qlo.z};
        double rad = sqrt(L), e_min = fmin(e[0], fmin(e[1], e[2]));
        double area = (e[0] + 2 * rad) * (e[1] + 2 * rad) * (e[2] + 2 * rad) / (e_min + 2 * rad);
        if (area * (double)KNN_K > (double)PM_TILE_EST_MAX * 3.14159 * r_scale) { why = 2; return false; }
    }
    // 2. the shared candidate set: everything closer than sqrt(L) to the box, so everything closer than that to any lane's
    //    point (collected with a hair more than L: the two distances are rounded differently)
    const int ns = __builtin_amdgcn_readfirstlane(tile_collect(tb, S, m, qlo, qhi, L * (1.0 + 1e-12)));
    TS_LAP(1)
    if (ns < 0) { why = 4; return false; }
#ifdef PM_TILE_STAMPS
    if (lane == 0) { atomicAdd(&tphase[6], (unsigned long long)ns); atomicAdd(&tphase[7], 1ULL); }
#endif
    // 3. The two passes over S work on FLOAT distances d2f computed from the relative positions; g bounds |d2f - d2| for
    //    every candidate with d2 up to about L (dx is off by at most eta = 4 * 2^-24 * M per axis, M the largest relative
    //    coordinate; then the float sum of squares; the bin edges themselves are float products), times 4 for comfort.
    const V3 qc = {0.5 * (qlo.x + qhi.x), 0.5 * (qlo.y + qhi.y), 0.5 * (qlo.z + qhi.z)};
    const double radL = sqrt(L);
    const double M = 0.5 * fmax(qhi.x - qlo.x, fmax(qhi.y - qlo.y, qhi.z - qlo.z)) + radL;
    const double eta = M * (1.0 / 4194304.0);
    const double g = 4.0 * (3.5 * eta * radL + 3.0 * eta * eta + L * (1.0 / 524288.0));
    const float pxf = (float)(p.x - qc.x), pyf = (float)(p.y - qc.y), pzf = (float)(p.z - qc.z);
    // pass 1: histogram of d2f over [0, Lf), Lf = L - g rounded down: d2f < Lf is certain to mean d2 < L
    TileSel &sel = tb->sel;
#pragma unroll
    for (int w = 0; w < TS_BINS / 4; w++) sel.hist[w * 64 + lane] = 0u;
    wave_lds_fence();
    const float Lf = __double2float_rd(L - g);
    const float fscale = (float)TS_BINS / (float)L;
    const float Lq = active ? Lf : -1.0f;
    uint32_t n_in = 0;
    tile_pass(reinterpret_cast<float *>(sel.member), S, ns, pxf, pyf, pzf, [&](int, float d2f) {   // (the member lists are not in use yet)
        if (d2f < Lq) {
            int bin = min(TS_BINS - 1, (int)(d2f * fscale));
            atomicAdd(&sel.hist[(bin >> 2) * 64 + lane], 1u << (8 * (bin & 3)));
            n_in++;
        }
    }, [](int) {});
    wave_lds_fence();
    int bstar = -1;
    uint32_t cum = 0;
#pragma unroll
    for (int w = 0; w < TS_BINS / 4; w++) {
        uint32_t h = sel.hist[w * 64 + lane];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            uint32_t c = (h >> (8 * b)) & 0xFFu;
            if (bstar < 0 && cum + c >= (uint32_t)KNN_K) bstar = 4 * w + b;
            cum += c;
        }
    }
    wave_lds_fence();
    TS_LAP(2)
    // The 63rd-smallest d2 lies in [lo_e - g, hi_e + g], lo_e and hi_e the edges of the bin in which the running count
    // crosses 63.  So a photon with d2f < lo_e - 2g is below it for sure, one with d2f > hi_e + 2g above it for sure, and
    // the photons in between are ranked by their exact d2.  A lane is handed over when it has fewer than 63 photons
    // certainly inside the shared radius, when its 8-bit counters overflowed (the unpacked bins then do not add up), or
    // when the band of uncertain photons reaches the rim of the shared radius.
    const double lo_e = (double)bstar * (L * (1.0 / TS_BINS)), hi_e = (double)(bstar + 1) * (L * (1.0 / TS_BINS));
    bool ok = active && bstar >= 0 && cum == n_in && hi_e + 2.0 * g < (double)Lf;
    if (__ballot(ok) == 0ULL) { why = 5; return false; }
    const float lo_t = ok ? __double2float_rd(lo_e - 2.0 * g) : -1.0f, hi_t = ok ? __double2float_ru(hi_e + 2.0 * g) : -1.0f;
    // pass 2: members for sure, list of the uncertain band.  A member is stored as one byte, its index in S modulo TS_PAGE;
    // candidates come in ascending order, so all it takes to get the index back is the number of members the lane had when
    // each page of S began (pg_lo / pg_hi: a byte per page, 255 = the page was never reached; at most 62 members are sure).
    int mc = 0, bc = 0;
    uint32_t pg_lo = 0xFFFFFFFFu, pg_hi = 0xFFFFFFFFu;
    tile_pass(reinterpret_cast<float *>(sel.hist), S, ns, pxf, pyf, pzf, [&](int j, float d2f) {   // (the histogram is done)
        // (the two counters are advanced outside the branches: merged into one conditional store they end up in memory)
        const bool sure = d2f < lo_t, band = !sure && d2f <= hi_t;
        if (sure) sel.member[lane * TS_MROW + mc] = (uint8_t)j;
        if (band && bc < TS_BINL) sel.binl[bc * 64 + lane] = (uint16_t)j;
        mc += sure ? 1 : 0;
        bc += band ? 1 : 0;
    }, [&](int c) {
        constexpr int per_page = TS_PAGE / TS_CHUNK;
        if (c % per_page == 0) {
            const int pgn = c / per_page, sh = 8 * (pgn & 3);
            if (pgn < 4) pg_lo = (pg_lo & ~(0xFFu << sh)) | ((uint32_t)mc << sh);
            else pg_hi = (pg_hi & ~(0xFFu << sh)) | ((uint32_t)mc << sh);
        }
    });
    TS_LAP(3)
    // exact d2 of the uncertain band (positions from the map itself), then the exact rank: r = the entry with fewer than
    // `need` entries below it and at least `need` not above it
    const int need = KNN_K - mc;   // 1-based rank of the answer inside the band
    if (bc > TS_BINL || need < 1 || need > bc) { ok = false; bc = 0; mc = 0; }
    if (__ballot(ok) == 0ULL) { why = 6; return false; }
    wave_lds_fence();
    double bd[TS_BINL];
    uint16_t bj[TS_BINL];
#pragma unroll
    for (int a = 0; a < TS_BINL; a++) {
        bd[a] = __longlong_as_double(0x7FF0000000000000LL); bj[a] = 0;
        if (a < bc) { bj[a] = sel.binl[a * 64 + lane]; bd[a] = tile_d2(m.pos4[S.idx[bj[a]]], p); }
    }
    double r = 0.0;
#pragma unroll
    for (int a = 0; a < TS_BINL; a++) {
        int less = 0, leq = 0;
#pragma unroll
        for (int b = 0; b < TS_BINL; b++) { less += bd[b] < bd[a]; leq += bd[b] <= bd[a]; }
        if (a < bc && less < need && need <= leq) r = bd[a];
    }
    // those of the band that are members follow the sure ones, listed by their full index (the band's own list is reused:
    // entry bm is never beyond the entry it came from)
    int bm = 0;
#pragma unroll
    for (int a = 0; a < TS_BINL; a++)
        if (a < bc && bd[a] < r) { sel.binl[bm * 64 + lane] = bj[a]; bm++; }   // d2 == r has weight 0
    const int mc_sure = mc;
    mc += bm;
    wave_lds_fence();
    TS_LAP(4)
    // 4. radiance: every lane sums its own members (canonical sum: 64-bit fixed point, 2^-56 resolution)
    int mcm = mc;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) mcm = max(mcm, __shfl_xor(mcm, s));
    long long aR = 0, aG = 0, aB = 0;
    const RDiv rd = rdiv_make(r);
    // three stages in flight: photon index of members k+4, k+5; the two records of members k+2, k+3; the terms of k, k+1
    // (index_of is called for k = 0, 1, 2, ... in turn: the page of the k-th sure member is found by walking on)
    int page = 0;
    const unsigned long long pages = ((unsigned long long)pg_hi << 32) | pg_lo;
    int page_end = (int)((pages >> 8) & 255u);            // members the lane had when the next page began
    auto index_of = [&](int k) -> uint32_t {
        if (k >= mc) return 0u;
        if (k >= mc_sure) return S.idx[sel.binl[(k - mc_sure) * 64 + lane]];
        while (k >= page_end) { page++; page_end = page < 7 ? (int)((pages >> (8 * (page + 1))) & 255u) : 255; }
        return S.idx[page * TS_PAGE + (int)sel.member[lane * TS_MROW + k]];
    };
    auto term = [&](int k, const float4 &s, const uint4 &rr) {
        if (k < mc) {
            long long fr, fg, fb;
            // d2 as everywhere (photon - point, three squares summed left to right); the term wants point - photon: the negation is exact
            const double ddx = (double)s.x - p.x, ddy = (double)s.y - p.y, ddz = (double)s.z - p.z;
            photon_term_fx<KIND>(-ddx, -ddy, -ddz, rr, nrm, ddx * ddx + ddy * ddy + ddz * ddz, rd, fr, fg, fb);
            aR += fr; aG += fg; aB += fb;
        }
    };
    uint32_t i0 = index_of(0), i1 = index_of(1), i2 = index_of(2), i3 = index_of(3);
    float4 s0 = m.pos4[i0], s1 = m.pos4[i1];
    uint4 r0 = m.rest[i0], r1 = m.rest[i1];
    for (int k = 0; k < mcm; k += 2) {
        uint32_t i4 = index_of(k + 4), i5 = index_of(k + 5);
        float4 s2 = m.pos4[i2], s3 = m.pos4[i3];
        uint4 r2 = m.rest[i2], r3 = m.rest[i3];
        term(k, s0, r0); term(k + 1, s1, r1);
        s0 = s2; r0 = r2; s1 = s3; r1 = r3; i2 = i4; i3 = i5;
    }
    const double norm_k = KIND == 0 ? (3.141592653589 * 2) : (3.141592653589 * 8);
    R = ((double)aR * (1.0 / FX_SCALE)) / (norm_k * r);
    G = ((double)aG * (1.0 / FX_SCALE)) / (norm_k * r);
    B = ((double)aB * (1.0 / FX_SCALE)) / (norm_k * r);
    wave_lds_fence();
    TS_LAP(5)
    return ok;
}

}  // namespace pm
