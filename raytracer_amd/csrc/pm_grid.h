// pm_grid.h -- wave-per-query exact k-NN (k = 63) on a linear octree: photons sorted by 63-bit Morton
// key, so every aligned cell of every level is ONE contiguous range of the sorted array, looked up in
// per-level start tables.  Replaces find_nearest_photons + insert (reference src/PhotonMap.cpp:142-255).
//
// A query picks the finest level whose cells still hold enough photons around it, takes the 3x3x3
// block of cells around its own cell (27 lanes: one cell each), and scans the photons of those cells
// 64 at a time (coalesced 16-byte loads).  Everything closer than g, the distance from the query to the
// nearest face of the block, is guaranteed to be in the block, so the 63 nearest found are exact as soon
// as the 63rd distance is below g; otherwise the query is repeated one level coarser.  Candidates go to
// the per-wave LDS buffer and the histogram + radix select of pm_knn.h picks the exact 63rd distance.
#pragma once
#include "pm_knn.h"
#include "pm_hilbert.h"

namespace pm {

#ifndef PM_GRID_T_OWN
#define PM_GRID_T_OWN 12       // finest level whose own cell holds at least this many photons
#endif
#ifndef PM_GRID_EST
#define PM_GRID_EST 80         // expected photons inside the guaranteed ball needed to try a level
#endif

__device__ inline uint32_t quant21(float p, float lo, float ext) {   // shared by build and query
    float t = ext > 0.f ? (p - lo) / ext : 0.f;
    return (uint32_t)fminf(fmaxf(t * 2097152.0f, 0.f), 2097151.0f);
}
__device__ inline uint64_t morton63(uint32_t x, uint32_t y, uint32_t z) { return spread21(x) | (spread21(y) << 1) | (spread21(z) << 2); }

__device__ inline uint32_t wave_sum_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl_u32(v), 63); }

// Scan the photons of the 3x3x3 block at `level`; returns true when the result in the buffer is exact.
#ifndef PM_GRID_SPLIT
#define PM_GRID_SPLIT 192      // a cell of the scan list holding more photons than this is split into its 8 children
#endif

PM_WAVE_REDUCE(wave_max_u32, uint32_t, umax_op)

// photon range [first, first + cnt) of the cell with Morton key `key` at `level`; below the finest table the
// cell is a sub-range of [plo, pend) (its parent's or ancestor's range): two binary searches on the sorted keys
__device__ inline void cell_range(const MapView &m, int level, uint64_t key, uint32_t plo, uint32_t pend, uint32_t &first, uint32_t &cnt) {
    if (level <= m.lmax) {
        const uint32_t *tab = m.cells + cell_off(level);
        first = tab[key];
        cnt = tab[key + 1] - first;
        return;
    }
    const int ks = 63 - 3 * level;
    uint32_t a = plo, b = pend;
    while (a < b) { uint32_t mid = (a + b) >> 1; if ((m.keys[mid] >> ks) < key) a = mid + 1; else b = mid; }
    first = a;
    b = pend;
    while (a < b) { uint32_t mid = (a + b) >> 1; if ((m.keys[mid] >> ks) <= key) a = mid + 1; else b = mid; }
    cnt = a - first;
}
// squared distance from q to the cell (cx,cy,cz) of `level`, shrunk by the two-fine-unit binning margin
__device__ inline double cell_d2(const MapView &m, const V3 &q, int level, uint32_t cx, uint32_t cy, uint32_t cz) {
    double h = (double)(1u << (21 - level)) * m.gu, d2 = 0.0;
    const uint32_t cc[3] = {cx, cy, cz};
#pragma unroll
    for (int a = 0; a < 3; a++) {
        double qa = a == 0 ? q.x : (a == 1 ? q.y : q.z);
        double lo_face = (double)m.glo[a] + (double)cc[a] * h - 2.0 * m.gu, hi_face = lo_face + h + 4.0 * m.gu;
        double gap = qa < lo_face ? lo_face - qa : (qa > hi_face ? qa - hi_face : 0.0);
        d2 += gap * gap;
    }
    return d2;
}

// verdict: 0 = scanned (return value says whether the result is exact), -1 = too sparse, go coarser, +1 = far denser than
// needed, go finer (only offered while allow_finer)
__device__ inline bool grid_attempt(KnnState &st, const MapView &m, const V3 &q, const uint32_t qf[3], int level, bool allow_finer, int &verdict) {
    const int lane = st.lane;
    const int sh = 21 - level;
    const uint32_t ncell = 1u << level;
    // this lane's cell of the block (lanes 0..26); every lane later holds one cell of the scan list
    int dx = lane % 3 - 1, dy = (lane / 3) % 3 - 1, dz = lane / 9 - 1;
    long long bx = (long long)(qf[0] >> sh) + dx, by = (long long)(qf[1] >> sh) + dy, bz = (long long)(qf[2] >> sh) + dz;
    bool cell_ok = lane < 27 && bx >= 0 && by >= 0 && bz >= 0 && bx < ncell && by < ncell && bz < ncell;
    int clev = level;
    uint32_t cx = cell_ok ? (uint32_t)bx : 0u, cy = cell_ok ? (uint32_t)by : 0u, cz = cell_ok ? (uint32_t)bz : 0u;
    uint32_t first = 0, cnt = 0;
    if (cell_ok) {
        uint64_t key = morton63(cx, cy, cz);   // 3*level bits
        uint32_t plo = 0, pend = 0;
        if (level > m.lmax) {
            const uint32_t *tab = m.cells + cell_off(m.lmax);
            uint64_t anc = key >> (3 * (level - m.lmax));
            plo = tab[anc]; pend = tab[anc + 1];
        }
        cell_range(m, level, key, plo, pend, first, cnt);
    }
    uint32_t total = wave_sum_u32(cnt);
    // g: everything closer than this to q is inside the block (photons were binned with float arithmetic:
    // two fine units of margin).  A side on which the block reaches the border of the grid does not limit g.
    double g2 = MAX_R2;
    bool covers = true;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        long long c = (long long)(qf[a] >> sh);
        double qa = a == 0 ? q.x : (a == 1 ? q.y : q.z);
        if (c - 1 > 0) {
            double face = (double)m.glo[a] + ((double)((c - 1) << sh) + 2.0) * m.gu;
            double g = qa - face;
            covers = false;
            g2 = fmin(g2, g > 0 ? g * g : 0.0);
        }
        if (c + 2 < (long long)ncell) {
            double face = (double)m.glo[a] + ((double)((c + 2) << sh) - 2.0) * m.gu;
            double g = face - qa;
            covers = false;
            g2 = fmin(g2, g > 0 ? g * g : 0.0);
        }
    }
    if (!covers) {
        // cheap rejection before any photon is read: expected photons inside the guaranteed ball, taking the
        // photons of the block to lie on a surface through it (area 9 h^2)
        double h = (double)(1u << sh) * m.gu;
        double est = (double)total * (3.14159 * g2) / (9.0 * h * h);
        if (est < (double)PM_GRID_EST) { verdict = -1; return false; }
        if (allow_finer && level < 21 && est >= 4.0 * (double)PM_GRID_EST) { verdict = 1; return false; }
    }
    verdict = 0;
    // Only d2 < lim is useful for THIS attempt: beyond g the block is not complete.
    st.lim = covers ? MAX_R2 : fmin(MAX_R2, g2);
    st.count = 0;
    st.reduced = false;
    if (cnt > 0 && cell_d2(m, q, clev, cx, cy, cz) >= st.lim) cnt = 0;   // cell entirely beyond lim
    // Refine the scan list: while some cell is much fuller than a query needs (a caustic focus next to a sparse
    // spot, a dense floor below an off-surface query), replace it by those of its 8 children that still reach
    // into the lim ball.  Children are contiguous sub-ranges of the parent (Morton order).
    for (int round = 0; round < 16; round++) {
        uint32_t mx = wave_max_u32(cnt);
        if (mx <= (uint32_t)PM_GRID_SPLIT) break;
        unsigned long long freem = __ballot(cnt == 0 || cnt == mx);
        int big = __ffsll((long long)__ballot(cnt == mx)) - 1;
        freem |= 1ULL << big;
        // only ONE cell with cnt == mx is split; other cells with the same count keep their lane
        unsigned long long same = __ballot(cnt == mx) & ~(1ULL << big);
        freem &= ~same;
        if (__popcll(freem) < 8) break;
        int plev = __builtin_amdgcn_readlane(clev, big);
        if (plev >= 21) break;
        uint32_t px = (uint32_t)__builtin_amdgcn_readlane((int)cx, big), py = (uint32_t)__builtin_amdgcn_readlane((int)cy, big),
                 pz = (uint32_t)__builtin_amdgcn_readlane((int)cz, big);
        uint32_t pfirst = (uint32_t)__builtin_amdgcn_readlane((int)first, big), pcnt = (uint32_t)__builtin_amdgcn_readlane((int)cnt, big);
        int j = lane_rank(freem);
        bool is_child = ((freem >> lane) & 1ULL) && j < 8;
        if (lane == big) cnt = 0;
        if (is_child) {
            clev = plev + 1;
            cx = 2 * px + (uint32_t)(j & 1); cy = 2 * py + (uint32_t)((j >> 1) & 1); cz = 2 * pz + (uint32_t)(j >> 2);
            cell_range(m, clev, morton63(cx, cy, cz), pfirst, pfirst + pcnt, first, cnt);
            if (cnt > 0 && cell_d2(m, q, clev, cx, cy, cz) >= st.lim) cnt = 0;
        }
    }
    uint32_t incl = wave_scan_incl_u32(cnt);
    total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    uint32_t *pre = st.buf->cellpre;         // [0..63] inclusive prefix of the non-empty cells, [64..127] index offset of each
    unsigned long long nonempty = __ballot(cnt > 0);
    int k = lane_rank(nonempty);
    wave_lds_fence();
    pre[lane] = 0xFFFFFFFFu;
    wave_lds_fence();
    if (cnt > 0) { pre[k] = incl; pre[64 + k] = first - (incl - cnt); }   // second word: add the global position t to get the photon index
    wave_lds_fence();
    for (uint32_t base = 0; base < total; base += 64) {
        uint32_t t = base + (uint32_t)lane;
        // number of non-empty cells whose inclusive prefix is <= t  (binary search, 64 slots)
        int j = 0;
#pragma unroll
        for (int step = 32; step > 0; step >>= 1)
            if (pre[j + step - 1] <= t) j += step;
        bool valid = t < total;
        uint32_t i = valid ? pre[64 + j] + t : 0u;
        st.c_leaves++;
        float4 p = valid ? m.pos4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        double ddx = (double)p.x - q.x, ddy = (double)p.y - q.y, ddz = (double)p.z - q.z;   // src/PhotonMap.cpp:211-214
        double d2 = ddx * ddx + ddy * ddy + ddz * ddz;
        bool pass = valid && d2 < st.lim;
        unsigned long long mask = __ballot(pass);
        if (mask == 0ULL) continue;
        if (pass) {
            int pos = st.count + lane_rank(mask);
            st.buf->d[pos] = (unsigned long long)__double_as_longlong(d2);
            st.buf->i[pos] = i;
        }
        int np = __popcll(mask);
        st.count += np;
        st.c_cands += np;
        if (st.count >= KBUF - 64) knn_reduce(st);
    }
    if (st.count > KNN_K) knn_reduce(st);
    // exact if the block covers everything within reach, or the 63rd distance lies inside the guaranteed ball
    // (lim was capped at g2, so 63 candidates found means exactly that)
    return covers || st.count >= KNN_K;
}

// Full query; same contract as knn_query (pm_knn.h).
__device__ inline int grid_query(KnnState &st, KnnBuf *buf, const MapView &m, const V3 &q, double &r) {
    st.lane = threadIdx.x & 63;
    st.buf = buf;
    st.lim = MAX_R2;
    st.count = 0;
    st.reduced = false;
    st.any_reduce = false;
    st.c_nodes = st.c_leaves = st.c_reduces = st.c_cands = 0;
#ifdef PM_STAMPS
    st.t_node = st.t_leaf = st.t_red = 0;
#endif
    r = 0.0;
    if (m.n <= 0) return 0;
    uint32_t qf[3] = {quant21((float)q.x, m.glo[0], m.gext), quant21((float)q.y, m.glo[1], m.gext), quant21((float)q.z, m.glo[2], m.gext)};
    // level choice: lane l looks up the population of the query's own cell at level l
    uint64_t qkey = morton63(qf[0], qf[1], qf[2]);
    uint32_t own = 0;
    if (st.lane <= m.lmax) {
        uint64_t key = st.lane == 0 ? 0ULL : (qkey >> (3 * (21 - st.lane)));
        const uint32_t *tab = m.cells + cell_off(st.lane);
        own = tab[key + 1] - tab[key];
    }
    unsigned long long okm = __ballot(own >= (uint32_t)PM_GRID_T_OWN);
    int level = okm ? 63 - __clzll((long long)okm) : 0;
    // a query off the photon-bearing surfaces has an empty own cell long before its 3x3x3 block runs dry:
    // start two levels finer and let the density estimate walk back up (a rejected level costs 27 table reads)
    if (level < m.lmax) level = min(m.lmax, level + 2);
    else {
        // denser than the finest table resolves: photons on a surface thin out 4x per level
        uint32_t own_f = (uint32_t)__builtin_amdgcn_readlane((int)own, m.lmax) / (uint32_t)PM_GRID_T_OWN;
        if (own_f >= 4u) level = min(21, m.lmax + (31 - __clz((int)own_f)) / 2);
    }
    bool allow_finer = true;
    for (;;) {
        st.c_nodes++;
        int verdict = 0;
        if (grid_attempt(st, m, q, qf, level, allow_finer, verdict)) break;
        if (verdict > 0) { level++; continue; }
        allow_finer = false;     // once a level was too sparse (or failed) the walk only goes coarser: no oscillation
        if (level == 0) break;   // cannot happen: level 0 always covers; belt and braces
        level--;
    }
    if (st.count == 0) return 0;
    wave_lds_fence();
    if (st.reduced) { r = st.lim; return st.count; }
    double d = st.lane < st.count ? __longlong_as_double((long long)st.buf->d[st.lane]) : 0.0;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) d = fmax(d, __shfl_xor(d, s));
    r = d;
    return st.count;
}

}  // namespace pm
