// pm_grid.h -- wave-per-query exact k-NN (k = 63) on a linear octree: photons sorted by 63-bit Morton
// key, so every aligned cell of every level is ONE contiguous range of the sorted array, looked up in
// per-level start tables.  Replaces find_nearest_photons + insert (reference src/PhotonMap.cpp:142-255).
//
// A query is a best-first walk over octree cells with the 64 lanes of the wave as the queue (one cell per
// lane).  The queue starts as an exact cover of space: the query's own cell at the finest table level and
// the 7 siblings of its ancestor at every level above (one table lookup per lane).  Each round either
// scans, 64 photons at a time (coalesced 16-byte loads), the small cells that are nearer than every cell
// still too big to scan, or splits the nearest big cells into their 8 children.  Cells whose distance
// from the query is not below the running bound `lim` are dropped; `lim` falls to the exact 63rd-smallest
// distance each time the candidate buffer is reduced (histogram + radix select of pm_knn.h).  The walk
// ends when no cell is left, so the result is exact wherever the query lies -- on a photon-bearing
// surface (a handful of cells), or far from every photon (only the cells its ball reaches).
#pragma once
#include "pm_knn.h"
#include "pm_morton.h"

namespace pm {

__device__ inline uint32_t quant21(float p, float lo, float ext) {   // shared by build and query
    float t = ext > 0.f ? (p - lo) / ext : 0.f;
    return (uint32_t)fminf(fmaxf(t * 2097152.0f, 0.f), 2097151.0f);
}
__device__ inline uint64_t morton63(uint32_t x, uint32_t y, uint32_t z) { return spread21(x) | (spread21(y) << 1) | (spread21(z) << 2); }

constexpr int GRID_MAX_LEVEL = 10;   // cells are not split below this level: photons are sorted on the top 30 key bits only (pm_sort.hip)
#ifndef PM_GRID_SPLIT
#define PM_GRID_SPLIT 128      // a queued cell holding more photons than this is split into its 8 children before it is read
#endif
#ifndef PM_GRID_SLACK
#define PM_GRID_SLACK 24       // candidates beyond 63 that make a reduce worth its cost (a tighter lim prunes cells)
#endif
#ifndef PM_GRID_MULTI
#define PM_GRID_MULTI 4        // big cells split per round once lim is an actual k-th distance
#endif
#ifndef PM_BLOCK_MIN
#define PM_BLOCK_MIN 24        // block query: the level whose cell around the query holds at least this many photons ...
#endif
#ifndef PM_BLOCK_MAX
#define PM_BLOCK_MAX 1536      // ... and whose 3x3x3 block holds at most this many (a dense cluster next door: the walk prunes it, the block cannot)
#endif

// photon range [first, first + cnt) of the cell with Morton key `key` at `level`; below the finest table the
// cell is a sub-range of [plo, pend) (its parent's range): two binary searches on the sorted keys
__device__ inline void cell_range(const MapView &m, int level, uint64_t key, uint32_t plo, uint32_t pend, uint32_t &first, uint32_t &cnt) {
    if (level <= m.lmax) {
        // one table, of the finest level: a coarser cell is an aligned run of its entries
        const int up = 3 * (m.lmax - level);
        first = m.cells[key << up];
        cnt = m.cells[(key + 1) << up] - first;
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
// key of a cell: 3 * level bits.  Table levels fit 32-bit arithmetic.
__device__ inline uint32_t spread10(uint32_t v) {   // 10 bits -> every third bit
    v &= 0x3FFu;
    v = (v | v << 16) & 0x030000FFu;
    v = (v | v << 8) & 0x0300F00Fu;
    v = (v | v << 4) & 0x030C30C3u;
    v = (v | v << 2) & 0x09249249u;
    return v;
}
__device__ inline uint64_t cell_key(int level, uint32_t cx, uint32_t cy, uint32_t cz) {
    if (level <= 10) return (uint64_t)(spread10(cx) | spread10(cy) << 1 | spread10(cz) << 2);
    return morton63(cx, cy, cz);
}

// squared distances from the query box [qlo, qhi] (a point when both are equal) to the nearest and the farthest point of
// the box of cell (cx,cy,cz) of `level`: the occupied slices of the cell where the map stores them (table levels), the
// whole cube below; either way grown by the two-fine-unit binning margin (photons were binned with float arithmetic).
// dmin2 is never above, dmax2 never below the d2 of a photon filed in the cell (dmax2 is meant for point queries).
__device__ inline void cell_d2(const MapView &m, const V3 &qlo, const V3 &qhi, int level, uint32_t cx, uint32_t cy, uint32_t cz, uint64_t key,
                               bool boxed, double &dmin2, double &dmax2) {
    const int sh = 21 - level;
    uint32_t lo_s[3] = {0u, 0u, 0u}, hi_s[3] = {1u << sh, 1u << sh, 1u << sh};   // in fine units from the cell's corner
    if (boxed) {   // level <= lmax <= 8: a slice is 2^(sh-10) >= 8 fine units
        uint64_t w = m.cbox[box_off(level) + (long long)key];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            lo_s[a] = ((uint32_t)(w >> (20 * a)) & 1023u) << (sh - 10);
            hi_s[a] = (((uint32_t)(w >> (20 * a + 10)) & 1023u) + 1u) << (sh - 10);
        }
    }
    dmin2 = dmax2 = 0.0;
    const uint32_t cc[3] = {cx, cy, cz};
#pragma unroll
    for (int a = 0; a < 3; a++) {
        double ql = a == 0 ? qlo.x : (a == 1 ? qlo.y : qlo.z), qh = a == 0 ? qhi.x : (a == 1 ? qhi.y : qhi.z);
        uint32_t o = cc[a] << sh;
        double lo_face = (double)m.glo[a] + ((double)(o + lo_s[a]) - 2.0) * m.gu, hi_face = (double)m.glo[a] + ((double)(o + hi_s[a]) + 2.0) * m.gu;
        double gap = qh < lo_face ? lo_face - qh : (ql > hi_face ? ql - hi_face : 0.0);
        double far = fmax(fabs(ql - lo_face), fabs(hi_face - qh));
        dmin2 += gap * gap;
        dmax2 += far * far;
    }
}

// Offer every photon of the cells flagged `take` (ranges [first, first + cnt) per lane): the ranges are laid end to
// end and read 64 photons at a time; candidates closer than lim go to the buffer.
__device__ inline void grid_scan(KnnState &st, const MapView &m, const V3 &q, bool take, uint32_t first, uint32_t cnt) {
    const int lane = st.lane;
    uint32_t c = take ? cnt : 0u;
    uint32_t incl = wave_scan_incl_u32(c);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    uint32_t *pre = st.buf->cellpre;         // [0..63] inclusive prefix of the taken cells, [64..127] index offset of each
    unsigned long long nonempty = __ballot(c > 0);
    const int ncell = __popcll(nonempty);
    int k = lane_rank(nonempty);
    wave_lds_fence();
    pre[lane] = 0xFFFFFFFFu;
    wave_lds_fence();
    if (c > 0) { pre[k] = incl; pre[64 + k] = first - (incl - c); }   // second word: add the position t in the list to get the photon index
    wave_lds_fence();
    // search depth for `ncell` slots: the largest first step s0 with 2 * s0 >= ncell (0 for a single cell)
    const int s0 = ncell <= 1 ? 0 : 1 << (31 - __clz(ncell - 1));
    // two chunks of 64 photons per iteration: both index searches and both loads are issued before either is used
    for (uint32_t base = 0; base < total; base += 128) {
        uint32_t t[2] = {base + (uint32_t)lane, base + 64u + (uint32_t)lane};
        uint32_t i[2];
        bool valid[2];
        float4 p[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            // number of taken cells whose inclusive prefix is <= t
            int j = 0;
            for (int step = s0; step > 0; step >>= 1)
                if (pre[j + step - 1] <= t[h]) j += step;
            valid[h] = t[h] < total;
            i[h] = valid[h] ? pre[64 + j] + t[h] : 0u;
            p[h] = valid[h] ? m.pos4[i[h]] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if (h == 1 && base + 64u >= total) break;
            st.c_leaves++;
            double ddx = (double)p[h].x - q.x, ddy = (double)p[h].y - q.y, ddz = (double)p[h].z - q.z;   // src/PhotonMap.cpp:211-214
            double d2 = ddx * ddx + ddy * ddy + ddz * ddz;
            bool pass = valid[h] && d2 < st.lim;
            unsigned long long mask = __ballot(pass);
            if (mask == 0ULL) continue;
            if (pass) {
                int pos = st.count + lane_rank(mask);
                st.buf->d[pos] = (unsigned long long)__double_as_longlong(d2);
                st.buf->i[pos] = i[h];
            }
            int np = __popcll(mask);
            st.count += np;
            st.c_cands += np;
            if (st.count >= KBUF - 64) knn_reduce(st);
        }
    }
}

// Block query: the common case answered without a walk.  A query that lies among photons (a shade point on a lit surface)
// finds its 63 neighbours inside the 3x3x3 block of cells around it at the level where a cell is about as wide as the 63rd
// distance.  One round of table lookups picks that level (the finest one whose cell around the query holds PM_BLOCK_MIN
// photons) and the block's 27 ranges, one scan reads them, one reduce selects.  The answer is exact whenever the 63rd
// distance turns out smaller than the distance m from the query to the nearest face of the block beyond which photons can
// lie: every photon outside the block is at least m away, and every photon of the block closer than m was offered.
// Returns false when it does not apply or did not succeed (fewer than 63 photons within m): the walk below then answers,
// from scratch.
__device__ inline bool block_query(KnnState &st, const MapView &m, const V3 &q, const uint32_t qf[3]) {
    const int lane = st.lane;
    const int L = m.lmax;
    if (L < 1) return false;
    const double qa[3] = {q.x, q.y, q.z};
    // the query must lie inside the grid (its fine coordinates are clamped otherwise and say nothing about distances)
    for (int a = 0; a < 3; a++) if (!(qa[a] >= (double)m.glo[a] && qa[a] <= (double)m.glo[a] + (double)m.gext)) return false;
    // photons in the query's own cell, level by level (lane l: level l)
    uint32_t first = 0, cnt = 0;
    if (lane >= 1 && lane <= L) {
        const int sh = 21 - lane;
        cell_range(m, lane, cell_key(lane, qf[0] >> sh, qf[1] >> sh, qf[2] >> sh), 0u, 0u, first, cnt);
    }
    const unsigned long long enough = __ballot(lane >= 1 && lane <= L && cnt >= (uint32_t)PM_BLOCK_MIN);
    if (enough == 0ULL) return false;
    const int lev = 63 - __clzll((long long)enough);     // the finest such level
    const int sh = 21 - lev;
    const uint32_t side = 1u << lev;
    const uint32_t c0[3] = {qf[0] >> sh, qf[1] >> sh, qf[2] >> sh};
    // distance to the nearest face of the block that has cells beyond it (the grid's own faces have nothing beyond them),
    // less the binning margin of cell_d2
    double mdist = __builtin_inf();
#pragma unroll
    for (int a = 0; a < 3; a++) {
        if (c0[a] > 0u) mdist = fmin(mdist, qa[a] - ((double)m.glo[a] + (double)((c0[a] - 1u) << sh) * m.gu));
        if (c0[a] + 1u < side) mdist = fmin(mdist, ((double)m.glo[a] + (double)((c0[a] + 2u) << sh) * m.gu) - qa[a]);
    }
    mdist -= 3.0 * m.gu;
    if (!(mdist > 0.0)) return false;
    const double lim = fmin(mdist * mdist, MAX_R2);
    // the block's cells, one per lane
    first = 0; cnt = 0;
    double dmin2 = 0.0, dmax2 = 0.0;
    if (lane < 27) {
        const int dx = lane % 3 - 1, dy = (lane / 3) % 3 - 1, dz = lane / 9 - 1;
        const uint32_t cx = c0[0] + (uint32_t)dx, cy = c0[1] + (uint32_t)dy, cz = c0[2] + (uint32_t)dz;   // wraps below zero: caught by the test
        if (cx < side && cy < side && cz < side) {
            const uint64_t key = cell_key(lev, cx, cy, cz);
            cell_range(m, lev, key, 0u, 0u, first, cnt);
            if (cnt > 0) { cell_d2(m, q, q, lev, cx, cy, cz, key, true, dmin2, dmax2); if (!(dmin2 < lim)) cnt = 0; }
        }
    }
    uint32_t total = wave_scan_incl_u32(cnt);
    total = (uint32_t)__builtin_amdgcn_readlane((int)total, 63);
    if (total < (uint32_t)KNN_K || total > (uint32_t)PM_BLOCK_MAX) return false;
    st.lim = lim;
    st.count = 0;
    st.reduced = false;
    grid_scan(st, m, q, cnt > 0, first, cnt);
    if (st.count < KNN_K) { st.count = 0; return false; }
    if (st.count > KNN_K || !st.reduced) knn_reduce(st);
    return true;      // st.lim is the exact 63rd distance, below m^2 by construction of the scan
}

// Full query; same contract as knn_query (pm_knn.h).
// lim0: a proven bound on the 63rd-smallest d2 (63 photons or more lie closer than it), or MAX_R2 when nothing is
// known.  Runs of neighbouring queries pass (sqrt(r_prev) + |q - q_prev|)^2: the previous ball lies inside that one.
__device__ inline int grid_query(KnnState &st, KnnBuf *buf, const MapView &m, const V3 &q, double &r, double lim0) {
    const float INF = __int_as_float(0x7F800000);
    const int lane = threadIdx.x & 63;
    st.lane = lane;
    st.buf = buf;
    st.lim = fmin(lim0, MAX_R2);
    st.count = 0;
    st.reduced = false;
    st.any_reduce = false;
    st.c_nodes = st.c_leaves = st.c_reduces = st.c_cands = 0;
#ifdef PM_STAMPS
    st.t_node = st.t_leaf = st.t_red = 0;
#endif
    r = 0.0;
    if (m.n <= 0) return 0;
    const uint32_t qf[3] = {quant21((float)q.x, m.glo[0], m.gext), quant21((float)q.y, m.glo[1], m.gext), quant21((float)q.z, m.glo[2], m.gext)};
#ifndef PM_NO_BLOCK_QUERY
    if (block_query(st, m, q, qf)) { st.c_nodes = 0x80000000u; wave_lds_fence(); r = st.lim; return st.count; }   // (work counters: the top bit of `nodes` says the block query answered)
    st.lim = fmin(lim0, MAX_R2);
    st.count = 0;
    st.reduced = false;
#endif
    // The queue, one cell per lane: lanes 7(l-1) .. 7(l-1)+6 hold the 7 siblings of the query's level-l cell, lane 7*lmax
    // the query's own cell at level lmax.  Together they tile the whole grid (lmax <= 8: 57 lanes).
    const int L = m.lmax;
#ifdef PM_STAMPS
    unsigned long long ts_ = __builtin_amdgcn_s_memtime();
#define PM_LAP(field) { unsigned long long now_ = __builtin_amdgcn_s_memtime(); st.field += now_ - ts_; ts_ = now_; }
#else
#define PM_LAP(field)
#endif
    int lev = 0;
    uint32_t cx = 0, cy = 0, cz = 0, first = 0, cnt = 0;
    double dmin2 = 0.0, dmax2 = 0.0;
    bool fresh_cells = true;
    bool tight = lim0 < MAX_R2;   // lim is already close to the answer: reduce only when the buffer has grown
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
        if (cnt > 0) cell_d2(m, q, q, lev, cx, cy, cz, key, true, dmin2, dmax2);
    }
    for (;;) {
        if (fresh_cells) {
            // a cell holding 63 photons or more bounds the 63rd distance by its farthest corner, before any photon is read
            float best = wave_min_f32(cnt >= (uint32_t)KNN_K ? __double2float_ru(dmax2) : INF);
            if (best < INF) {
                double bound = __longlong_as_double(__double_as_longlong((double)best) + 1);   // one ulp above: the test is '<'
                if (bound < st.lim) { st.lim = bound; st.reduced = false; }
            }
            fresh_cells = false;
        }
        const bool alive = cnt > 0 && dmin2 < st.lim;
        if (!alive) cnt = 0;
        const unsigned long long am = __ballot(alive);
        if (am == 0ULL) break;
        const bool big = alive && cnt > (uint32_t)PM_GRID_SPLIT && lev < GRID_MAX_LEVEL;
        const unsigned long long bm = __ballot(big);
        const float fkey = (float)dmin2;      // order of the walk only; pruning uses dmin2 itself
        const float fb = bm ? wave_min_f32(big ? fkey : INF) : INF;
        // small cells no farther than the nearest big one are read now
        bool take = alive && !big && fkey <= fb;
        const int nfree = 64 - __popcll(am);
        if (__ballot(take) == 0ULL) {
            if (nfree + 1 >= 8) {
                // split the nearest big cells: the parent's lane and the idle lanes receive the children
                st.c_nodes++;
                int nsplit = st.lim < MAX_R2 ? PM_GRID_MULTI : 1;
                unsigned long long todo = bm;
                unsigned long long freem = ~am;
                bool fresh = false;
                int plev = 0; uint32_t px = 0, py = 0, pz = 0, pfirst = 0, pcnt = 0; int cj = 0;
                float fnear = fb;
                while (nsplit-- > 0 && todo) {
                    int bl = __ffsll((long long)(__ballot(big && fkey == fnear) & todo)) - 1;
                    if (bl < 0) break;
                    unsigned long long fm = freem | (1ULL << bl);
                    if (__popcll(fm) < 8) break;
                    int j = lane_rank(fm);
                    bool mine = ((fm >> lane) & 1ULL) && j < 8;
                    int l_ = __builtin_amdgcn_readlane(lev, bl);
                    uint32_t x_ = (uint32_t)__builtin_amdgcn_readlane((int)cx, bl), y_ = (uint32_t)__builtin_amdgcn_readlane((int)cy, bl),
                             z_ = (uint32_t)__builtin_amdgcn_readlane((int)cz, bl);
                    uint32_t f_ = (uint32_t)__builtin_amdgcn_readlane((int)first, bl), c_ = (uint32_t)__builtin_amdgcn_readlane((int)cnt, bl);
                    if (mine) { fresh = true; plev = l_; px = x_; py = y_; pz = z_; pfirst = f_; pcnt = c_; cj = j; }
                    // the 8 lowest lanes of fm are used up
                    if (lane == bl && !mine) cnt = 0;   // the parent is replaced by its children
                    unsigned long long used = __ballot(mine);
                    freem = fm & ~used;
                    todo &= ~(1ULL << bl);
                    if (nsplit > 0 && todo) {
                        bool rest = big && ((todo >> lane) & 1ULL);
                        fnear = wave_min_f32(rest ? fkey : INF);
                    }
                }
                if (fresh) {
                    lev = plev + 1;
                    cx = 2 * px + (uint32_t)(cj & 1); cy = 2 * py + (uint32_t)((cj >> 1) & 1); cz = 2 * pz + (uint32_t)(cj >> 2);
                    uint64_t key = cell_key(lev, cx, cy, cz);
                    cell_range(m, lev, key, pfirst, pfirst + pcnt, first, cnt);
                    if (cnt > 0) cell_d2(m, q, q, lev, cx, cy, cz, key, lev <= m.lmax, dmin2, dmax2);
                }
                fresh_cells = true;
                PM_LAP(t_node)
                continue;
            }
            // no room to split: read every small cell to free lanes, or failing that the nearest big cell as it is
            take = alive && !big;
            if (__ballot(take) == 0ULL) take = lane == __ffsll((long long)__ballot(big && fkey == fb)) - 1;
        }
        PM_LAP(t_node)
        grid_scan(st, m, q, take, first, cnt);
        if (take) cnt = 0;
        PM_LAP(t_leaf)
        if (st.count >= KNN_K && (!tight || st.count >= KNN_K + PM_GRID_SLACK)) { knn_reduce(st); tight = true; }
#ifdef PM_STAMPS
        ts_ = __builtin_amdgcn_s_memtime();
#endif
    }
    PM_LAP(t_node)
    if (st.count > KNN_K) knn_reduce(st);
    if (st.count == 0) return 0;
    wave_lds_fence();
    if (st.reduced) { r = st.lim; return st.count; }
    double d = lane < st.count ? __longlong_as_double((long long)st.buf->d[lane]) : 0.0;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) d = fmax(d, __shfl_xor(d, s));
    r = d;
    return st.count;
}

}  // namespace pm
