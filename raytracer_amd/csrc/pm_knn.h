// pm_knn.h -- wave-per-query exact k-NN (k = 63) over a 64-ary BVH of Morton-sorted photons.
//
// Replaces find_nearest_photons + insert (reference src/PhotonMap.cpp:142-255).  The reference
// keeps a 63-entry max-heap per query; on a 64-lane wavefront the k-best list instead lives one
// entry per lane, sorted ascending by (d2, emission index), and is maintained with wave-wide
// insertion / bitonic networks.  Every tree node has up to 64 children, so one coalesced wave
// load fetches all child boxes (or all 64 photons of a leaf) of a node.
#pragma once
#include "pm_device.h"

namespace pm {

constexpr int KNN_K = 63;
constexpr int MAX_LEVELS = 5;        // 64^5 photons
constexpr double MAX_R2 = 100.0;     // MAX_PHOTON_RADIUS, reference src/Raytrace.cpp:13 (compared with squared distance)

struct BoxRec { float lox, loy, loz, pad0, hix, hiy, hiz, pad1; };  // 32 B

struct MapView {
    long long n;                 // photons
    const float4 *pos4;          // sorted order: x, y, z, bits(emission index)
    const uint4 *rest;           // sorted order: bits(dx), bits(dy), bits(dz), power[0..3]
    int top;                     // index of the top level (its nodes are the root's children)
    const BoxRec *lv[MAX_LEVELS];// lv[0] = leaf boxes (64 photons each), lv[t] = boxes of 64 lv[t-1] nodes
    int cnt[MAX_LEVELS];
};

__device__ inline float wave_min_f32(float v) {
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) v = fminf(v, __shfl_xor(v, s));
    return v;
}

struct Best {
    double d;      // squared distance (+inf = empty)
    uint32_t o;    // emission index (tie break)
    uint32_t i;    // index in sorted order
};
__device__ inline bool key_less(double da, uint32_t oa, double db, uint32_t ob) { return da < db || (da == db && oa < ob); }

// compare-exchange with lane^j; keepmin lanes keep the smaller key
__device__ inline void cmpx(Best &b, int j, bool keepmin) {
    double pd = __shfl_xor(b.d, j);
    uint32_t po = __shfl_xor(b.o, j);
    uint32_t pi = __shfl_xor(b.i, j);
    bool take = keepmin ? key_less(pd, po, b.d, b.o) : key_less(b.d, b.o, pd, po);
    if (take) { b.d = pd; b.o = po; b.i = pi; }
}
__device__ inline void bitonic_sort_asc(Best &b, int lane) {
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            bool up = (lane & k) == 0 || k == 64;
            bool lower = (lane & j) == 0;
            cmpx(b, j, lower == up);
        }
    }
}
__device__ inline void bitonic_merge_asc(Best &b, int lane) {
#pragma unroll
    for (int j = 32; j > 0; j >>= 1) cmpx(b, j, (lane & j) == 0);
}

struct KnnState {
    Best best;            // this lane's entry of the sorted k-best list
    double tau_d;         // key of entry k-1 (wave-uniform): candidates must sort before it
    uint32_t tau_o;
    int lane;
};

__device__ inline void knn_update_tau(KnnState &st) {
    st.tau_d = __shfl(st.best.d, KNN_K - 1);
    st.tau_o = __shfl(st.best.o, KNN_K - 1);
}

#ifndef PM_INSERT_MAX
#define PM_INSERT_MAX 6
#endif

// offer the (up to) 64 photons of one leaf
__device__ inline void knn_leaf(KnnState &st, const MapView &m, const V3 &q, int leaf) {
    long long i = (long long)leaf * 64 + st.lane;
    bool valid = i < m.n;
    float4 p = valid ? m.pos4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    // find_nearest_photons, src/PhotonMap.cpp:211-214
    double dx = (double)p.x - q.x, dy = (double)p.y - q.y, dz = (double)p.z - q.z;
    double d2 = dx * dx + dy * dy + dz * dz;
    uint32_t o = __float_as_uint(p.w);
    bool pass = valid && d2 < MAX_R2 && key_less(d2, o, st.tau_d, st.tau_o);
    unsigned long long mask = __ballot(pass);
    if (mask == 0ULL) return;
    Best c;
    c.d = pass ? d2 : __longlong_as_double(0x7FF0000000000000LL);
    c.o = pass ? o : 0xFFFFFFFFu;
    c.i = (uint32_t)i;
    if (__popcll(mask) <= PM_INSERT_MAX) {
        // few survivors: wave-wide insertion, one candidate at a time
        while (mask) {
            int src = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            double cd = __shfl(c.d, src);
            uint32_t co = __shfl(c.o, src), ci = __shfl(c.i, src);
            // tau may have tightened since the ballot
            if (!key_less(cd, co, st.tau_d, st.tau_o)) continue;
            bool before = key_less(st.best.d, st.best.o, cd, co);
            int pos = __popcll(__ballot(before));
            double pd = __shfl_up(st.best.d, 1);
            uint32_t po = __shfl_up(st.best.o, 1), pi = __shfl_up(st.best.i, 1);
            if (st.lane > pos) { st.best.d = pd; st.best.o = po; st.best.i = pi; }
            else if (st.lane == pos) { st.best.d = cd; st.best.o = co; st.best.i = ci; }
            knn_update_tau(st);
        }
    } else {
        // many survivors: sort them, keep the 64 smallest of the union, re-sort
        bitonic_sort_asc(c, st.lane);
        int rev = 63 - st.lane;
        double rd = __shfl(c.d, rev);
        uint32_t ro = __shfl(c.o, rev), ri = __shfl(c.i, rev);
        if (key_less(rd, ro, st.best.d, st.best.o)) { st.best.d = rd; st.best.o = ro; st.best.i = ri; }
        bitonic_merge_asc(st.best, st.lane);
        knn_update_tau(st);
    }
}

// squared distance from q to a box, rounded so that it never exceeds the d2 of a photon inside it
__device__ inline double box_d2(const BoxRec &b, const V3 &q) {
    double gx = q.x < (double)b.lox ? (double)b.lox - q.x : (q.x > (double)b.hix ? (double)b.hix - q.x : 0.0);
    double gy = q.y < (double)b.loy ? (double)b.loy - q.y : (q.y > (double)b.hiy ? (double)b.hiy - q.y : 0.0);
    double gz = q.z < (double)b.loz ? (double)b.loz - q.z : (q.z > (double)b.hiz ? (double)b.hiz - q.z : 0.0);
    return gx * gx + gy * gy + gz * gz;
}

// visit the children of `node` at level LV+1 (they live in m.lv[LV]); nearest child first
template <int LV>
__device__ inline void knn_visit(KnnState &st, const MapView &m, const V3 &q, int node) {
    const float INF = __int_as_float(0x7F800000);
    long long c0 = (long long)node * 64;
    long long ci = c0 + st.lane;
    double dm = __longlong_as_double(0x7FF0000000000000LL);
    if (ci < (long long)m.cnt[LV]) {
        const float4 *bp = reinterpret_cast<const float4 *>(&m.lv[LV][ci]);
        float4 lo = bp[0], hi = bp[1];
        BoxRec b = {lo.x, lo.y, lo.z, 0.f, hi.x, hi.y, hi.z, 0.f};
        dm = box_d2(b, q);
    }
    float key = (dm < MAX_R2) ? (float)dm : INF;
    for (;;) {
        if (!(dm <= st.tau_d)) key = INF;   // pruned by the current k-th distance (ties stay: index tie-break)
        float mn = wave_min_f32(key);
        if (mn == INF) break;
        unsigned long long eq = __ballot(key == mn);
        int c = __ffsll((long long)eq) - 1;
        if (st.lane == c) key = INF;
        if constexpr (LV == 0) knn_leaf(st, m, q, (int)(c0 + c));
        else knn_visit<LV - 1>(st, m, q, (int)(c0 + c));
    }
}

// full query; on return st.best holds the sorted result list (entries >= found are +inf)
__device__ inline void knn_query(KnnState &st, const MapView &m, const V3 &q) {
    st.lane = threadIdx.x & 63;
    st.best.d = __longlong_as_double(0x7FF0000000000000LL);
    st.best.o = 0xFFFFFFFFu;
    st.best.i = 0;
    st.tau_d = st.best.d;
    st.tau_o = 0xFFFFFFFFu;
    if (m.n <= 0) return;
    switch (m.top) {
    case 0: knn_visit<0>(st, m, q, 0); break;
    case 1: knn_visit<1>(st, m, q, 0); break;
    case 2: knn_visit<2>(st, m, q, 0); break;
    case 3: knn_visit<3>(st, m, q, 0); break;
    default: knn_visit<4>(st, m, q, 0); break;
    }
}

// One photon's term of the radiance sums: src/Raytrace.cpp:100-125 (global), :139-164 (caustic)
template <int KIND>
__device__ inline void photon_term(const float4 &pp, const uint4 &rr, const V3 &p, const V3 &nrm, double pd2, double r,
                                   double &tr, double &tg, double &tb) {
    tr = tg = tb = 0.0;
    float pdx = __uint_as_float(rr.x), pdy = __uint_as_float(rr.y), pdz = __uint_as_float(rr.z);
    double d = -(nrm.x * (double)pdx + nrm.y * (double)pdy + nrm.z * (double)pdz);
    if (d <= 0) return;
    double sx = p.x - (double)pp.x, sy = p.y - (double)pp.y, sz = p.z - (double)pp.z;
    // the plane filter with its dist.z quirk: NaN compares false -> photon kept
    double k = (sx * nrm.x + sy * nrm.y + sz * nrm.z) / (sz * sz + sz * sz + sz * sz);
    sx *= k; sy *= k; sz *= k;
    if (sx * sx + sy * sy + sz * sz > 0.1) return;
    double filter = (1 - pd2 / r);
    uint32_t pw = rr.w;
    int p0 = pw & 0xFF, p1 = (pw >> 8) & 0xFF, p2 = (pw >> 16) & 0xFF;
    if (KIND == 0) {
        filter = filter * filter;
        tr = d * (p0 / 255.0) * filter;
        tg = d * (p1 / 255.0) * filter;
        tb = d * (p2 / 255.0) * filter;
    } else {
        filter = filter * filter * filter * filter;
        // raytrace::min(int32,int32) on a truncated double: factor is 0 or 1
        tr = d * filter * imin((int32_t)((double)(p0 / 255.0f) + 0.5), 1);
        tg = d * filter * imin((int32_t)((double)(p1 / 255.0f) + 0.5), 1);
        tb = d * filter * imin((int32_t)((double)(p2 / 255.0f) + 0.5), 1);
    }
}

__device__ inline double wave_tree_sum(double v) {
    // canonical summation order: xor-butterfly over the 64 sorted slots (same tree in the oracle)
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) v = v + __shfl_xor(v, s);
    return v;
}

// query + radiance estimate; result valid in every lane
template <int KIND>
__device__ inline void knn_gather(const MapView &m, const V3 &p, const V3 &nrm, double &R, double &G, double &B) {
    KnnState st;
    knn_query(st, m, p);
    int found = __popcll(__ballot(st.best.d < __longlong_as_double(0x7FF0000000000000LL)));
    int nsel = found < KNN_K ? found : KNN_K;
    R = G = B = 0.0;
    if (nsel == 0) return;
    double r = __shfl(st.best.d, nsel - 1);   // distances[0] of the reference's max-heap
    double tr = 0, tg = 0, tb = 0;
    if (st.lane < nsel) {
        float4 pp = m.pos4[st.best.i];
        uint4 rr = m.rest[st.best.i];
        photon_term<KIND>(pp, rr, p, nrm, st.best.d, r, tr, tg, tb);
    }
    tr = wave_tree_sum(tr); tg = wave_tree_sum(tg); tb = wave_tree_sum(tb);
    const double norm_k = KIND == 0 ? (3.141592653589 * 2) : (3.141592653589 * 8);
    R = tr / (norm_k * r); G = tg / (norm_k * r); B = tb / (norm_k * r);
}

}  // namespace pm
