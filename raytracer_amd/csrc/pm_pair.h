// pm_pair.h -- the per-query gather with TWO queries per wave, one per half-wave of 32 lanes.
//
// One wave per query (pm_grid.h) spends most of its instructions on work that is the same for every lane: picking
// the level, looking up the block's cells, the histogram select.  A 3x3x3 block has 27 cells: it fits a half-wave, and
// every wave-wide instruction of the block query can serve two queries at once -- a 64-bit ballot is two 32-bit masks,
// one per query; DPP scans run per row; LDS holds one buffer per half.  This file restates block_query, grid_scan,
// knn_reduce and the radiance sum of pm_grid.h / pm_knn.h in that form.  Same exactness argument as block_query: a query is
// answered here only when its 63rd distance turns out smaller than the distance to the nearest face of the block that has
// cells beyond it.  A query that cannot be answered this way (no level fits, a dense cluster next door, fewer than 63
// photons in reach, a crossing bin of more than 32 entries) is appended to a list for the one-wave-per-query kernel.
// Results are bit-identical to that kernel's: the same exact 63rd distance, the same canonical fixed-point sum.
#pragma once
#include "pm_grid.h"

namespace pm {

constexpr int PAIR_BUF = 160;                  // candidates a half-wave buffers before it selects (5 slots of 32)
constexpr int PAIR_SLOTS = PAIR_BUF / 32;
#ifndef PM_PAIR_AHEAD
#define PM_PAIR_AHEAD 1
#endif
constexpr int PAIR_AHEAD = PM_PAIR_AHEAD;          // rounds of the block scan fetched together

struct PairHalf {
    unsigned long long d[PAIR_BUF];            // squared distances as IEEE bit patterns
    uint32_t i[PAIR_BUF];                      // photon index in sorted order
    uint32_t hist[64];
    unsigned long long sel[32];                // entries of the crossing bin
    uint32_t pre[32], off[32];                 // inclusive prefix of the block's non-empty cells; photon index = off + position
    unsigned long long acc[4];
};
struct PairBuf { PairHalf h[2]; };

#ifdef PM_PAIR_STAMPS
// diagnostics (builds with -DPM_PAIR_STAMPS only): clock ticks of a wave per phase of pair_gather, summed over all waves
static __device__ unsigned long long g_pair_ticks[8];
#define PAIR_LAP(k) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63) == 0) atomicAdd(&g_pair_ticks[k], now_ - ts_); ts_ = now_; }
#define PAIR_T0 unsigned long long ts_ = __builtin_amdgcn_s_memtime();
#else
#define PAIR_LAP(k)
#define PAIR_T0
#endif

// this lane's half of a 64-bit ballot
__device__ inline uint32_t half_mask(unsigned long long m, int half) { return half ? (uint32_t)(m >> 32) : (uint32_t)m; }
// inclusive prefix sum over the lanes of this lane's half, and the half's total
__device__ inline uint32_t half_scan_incl(uint32_t x, int half, uint32_t &total) {
    const uint32_t incl = wave_scan_incl_u32(x);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)incl, 31), all = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    total = half ? all - lo : lo;
    return half ? incl - lo : incl;
}

// Exact 63rd-smallest of the first `count` (>= 63, <= PAIR_BUF) entries of the half's buffer; compacts the buffer to the 63
// entries that matter and returns tau as a bit pattern.  `active`: this half takes part (uniform per half).  ok is cleared
// for a half whose crossing bin holds more than 32 entries.
__device__ inline unsigned long long pair_reduce(PairHalf *hb, int half, int hl, bool active, int count, double lim, bool &ok) {
    hb->hist[hl] = 0u; hb->hist[hl + 32] = 0u;
    wave_lds_fence();
    unsigned long long v[PAIR_SLOTS];
    uint32_t idx[PAIR_SLOTS], bin[PAIR_SLOTS];
    const double scale = (double)(64.0f * __frcp_rn((float)lim));
#pragma unroll
    for (int s = 0; s < PAIR_SLOTS; s++) {
        const bool valid = active && hl + 32 * s < count;
        v[s] = valid ? hb->d[hl + 32 * s] : 0ULL;
        idx[s] = valid ? hb->i[hl + 32 * s] : 0u;
        bin[s] = 0u;
        if (valid) {
            bin[s] = (uint32_t)fmin(__longlong_as_double((long long)v[s]) * scale, 63.0);
            atomicAdd(&hb->hist[bin[s]], 1u);
        }
    }
    wave_lds_fence();
    // running count over the 64 bins, two bins per lane
    const uint32_t h0 = hb->hist[2 * hl], h1 = hb->hist[2 * hl + 1];
    uint32_t tot;
    const uint32_t inc1 = half_scan_incl(h0 + h1, half, tot), inc0 = inc1 - h1;
    const uint32_t crossm = half_mask(__ballot(active && inc1 >= (uint32_t)KNN_K), half);       // exists: count >= 63
    const int fl = crossm ? __ffs((int)crossm) - 1 : 0;
    const bool in_first = inc0 >= (uint32_t)KNN_K;
    const int my_bstar = in_first ? 2 * hl : 2 * hl + 1, my_below = (int)(in_first ? inc0 - h0 : inc1 - h1);
    const int src = (half << 5) + fl;
    const uint32_t bstar = (uint32_t)__shfl(my_bstar, src);
    const int need = KNN_K - __shfl(my_below, src);
    // the crossing bin's entries, gathered; every lane ranks one by counting
    int run = 0;
#pragma unroll
    for (int s = 0; s < PAIR_SLOTS; s++) {
        const bool inbin = active && hl + 32 * s < count && bin[s] == bstar;
        const uint32_t mh = half_mask(__ballot(inbin), half);
        const int at = run + __popc(mh & ((1u << hl) - 1u));
        if (inbin && at < 32) hb->sel[at] = v[s];
        run += __popc(mh);
    }
    if (run > 32) ok = false;
    const int remaining = run > 32 ? 0 : run;
    wave_lds_fence();
    const unsigned long long x = hl < remaining ? hb->sel[hl] : ~0ULL;
    int less = 0, leq = 0;
    const int rounds = max(__builtin_amdgcn_readlane(remaining, 0), __builtin_amdgcn_readlane(remaining, 32));
    for (int k = 0; k < rounds; k++) {
        const unsigned long long y = k < remaining ? hb->sel[k] : ~0ULL;
        less += y < x ? 1 : 0;
        leq += y <= x ? 1 : 0;
    }
    const uint32_t hitm = half_mask(__ballot(hl < remaining && less < need && need <= leq), half);
    const unsigned long long tau = __shfl(x, (half << 5) + (hitm ? __ffs((int)hitm) - 1 : 0));
    // compact: all entries < tau, then enough entries == tau to make 63
    int n_lt = 0;
#pragma unroll
    for (int s = 0; s < PAIR_SLOTS; s++) n_lt += __popc(half_mask(__ballot(active && hl + 32 * s < count && v[s] < tau), half));
    int run_lt = 0, run_eq = 0;
    const uint32_t below = (1u << hl) - 1u;
#pragma unroll
    for (int s = 0; s < PAIR_SLOTS; s++) {
        const bool valid = active && hl + 32 * s < count;
        const uint32_t lt = half_mask(__ballot(valid && v[s] < tau), half), eq = half_mask(__ballot(valid && v[s] == tau), half);
        int pos = -1;
        if (valid && v[s] < tau) pos = run_lt + __popc(lt & below);
        else if (valid && v[s] == tau) { const int e = n_lt + run_eq + __popc(eq & below); if (e < KNN_K) pos = e; }
        if (pos >= 0 && remaining > 0) { hb->d[pos] = v[s]; hb->i[pos] = idx[s]; }
        run_lt += __popc(lt);
        run_eq += __popc(eq);
    }
    wave_lds_fence();
    return tau;
}

// Two queries per wave: lanes 0..31 answer the query of slot s of the low half, lanes 32..63 that of the high half (have:
// this lane's half has a query).  Returns per lane whether its half's query was answered (R, G, B valid in every lane of the half).
template <int KIND>
__device__ inline bool pair_gather(PairBuf *pb, const MapView &m, const V3 &q, const V3 &nrm, bool have, double &R, double &G, double &B) {
    const int lane = threadIdx.x & 63, half = lane >> 5, hl = lane & 31;
    PairHalf *hb = &pb->h[half];
    R = G = B = 0.0;
    PAIR_T0
    const int L = m.lmax;
    bool ok = have && m.n >= KNN_K && L >= 1;
    const double qa[3] = {q.x, q.y, q.z};
    for (int a = 0; a < 3; a++) ok = ok && qa[a] >= (double)m.glo[a] && qa[a] <= (double)m.glo[a] + (double)m.gext;
    const uint32_t qf[3] = {quant21((float)q.x, m.glo[0], m.gext), quant21((float)q.y, m.glo[1], m.gext), quant21((float)q.z, m.glo[2], m.gext)};
    // the level: photons in the query's own cell, level by level (lane l of the half: level l)
    uint32_t first = 0, cnt = 0;
    if (ok && hl >= 1 && hl <= L) {
        const int sh = 21 - hl;
        cell_range(m, hl, cell_key(hl, qf[0] >> sh, qf[1] >> sh, qf[2] >> sh), 0u, 0u, first, cnt);
    }
    const uint32_t enough = half_mask(__ballot(ok && hl >= 1 && hl <= L && cnt >= (uint32_t)PM_BLOCK_MIN), half);
    ok = ok && enough != 0u;
    const int lev = enough ? 31 - __clz((int)enough) : 1;
    const int sh = 21 - lev;
    const uint32_t side = 1u << lev;
    const uint32_t c0[3] = {qf[0] >> sh, qf[1] >> sh, qf[2] >> sh};
    double mdist = __builtin_inf();
#pragma unroll
    for (int a = 0; a < 3; a++) {
        if (c0[a] > 0u) mdist = fmin(mdist, qa[a] - ((double)m.glo[a] + (double)((c0[a] - 1u) << sh) * m.gu));
        if (c0[a] + 1u < side) mdist = fmin(mdist, ((double)m.glo[a] + (double)((c0[a] + 2u) << sh) * m.gu) - qa[a]);
    }
    mdist -= 3.0 * m.gu;
    ok = ok && mdist > 0.0;
    double lim = fmin(mdist * mdist, MAX_R2);
    PAIR_LAP(0)
    // the block's cells, one per lane of the half
    first = 0; cnt = 0;
    if (ok && hl < 27) {
        const int dx = hl % 3 - 1, dy = (hl / 3) % 3 - 1, dz = hl / 9 - 1;
        const uint32_t cx = c0[0] + (uint32_t)dx, cy = c0[1] + (uint32_t)dy, cz = c0[2] + (uint32_t)dz;
        if (cx < side && cy < side && cz < side) {
            const uint64_t key = cell_key(lev, cx, cy, cz);
            cell_range(m, lev, key, 0u, 0u, first, cnt);
            double dmin2 = 0.0, dmax2 = 0.0;
            if (cnt > 0) { cell_d2(m, q, q, lev, cx, cy, cz, key, true, dmin2, dmax2); if (!(dmin2 < lim)) cnt = 0; }
        }
    }
    uint32_t total;
    const uint32_t incl = half_scan_incl(cnt, half, total);
    ok = ok && total >= (uint32_t)KNN_K && total <= (uint32_t)PM_BLOCK_MAX;
    if (!ok) { cnt = 0; total = 0; }
    // the half's scan list: inclusive prefixes of its non-empty cells, padded with all ones for the search
    hb->pre[hl] = 0xFFFFFFFFu;
    wave_lds_fence();
    {
        const uint32_t nonempty = half_mask(__ballot(cnt > 0), half);
        const int k = __popc(nonempty & ((1u << hl) - 1u));
        if (cnt > 0) { hb->pre[k] = incl; hb->off[k] = first - (incl - cnt); }
    }
    wave_lds_fence();
    PAIR_LAP(1)
    int count = 0;
    const uint32_t below = (1u << hl) - 1u;
    const uint32_t longest = max((uint32_t)__builtin_amdgcn_readlane((int)total, 0), (uint32_t)__builtin_amdgcn_readlane((int)total, 32));
    // PAIR_AHEAD rounds of 32 photons are looked up and fetched together, then taken in turn.  (The scan is 58 % of a wave's
    // time in this function and it is waiting for memory -- tools/pair_phases.py -- but 2 or 4 rounds in flight changed no
    // step time: the gather phase is bound by the device's vector issue, which a wave that waits does not use.  1 = no look-ahead.)
    for (uint32_t base = 0; base < longest; base += 32 * PAIR_AHEAD) {
        uint32_t ia[PAIR_AHEAD];
        float4 pa[PAIR_AHEAD];
#pragma unroll
        for (int u = 0; u < PAIR_AHEAD; u++) {
            const uint32_t t = base + 32u * (uint32_t)u + (uint32_t)hl;
            int j = 0;
#pragma unroll
            for (int step = 16; step > 0; step >>= 1)
                if (hb->pre[j + step - 1] <= t) j += step;
            const bool valid = t < total;
            ia[u] = valid ? hb->off[j] + t : 0u;
            pa[u] = valid ? m.pos4[ia[u]] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < PAIR_AHEAD; u++) {
            const uint32_t t = base + 32u * (uint32_t)u + (uint32_t)hl;
            const bool valid = t < total;                    // (total is 0 for a half that has given up meanwhile)
            const uint32_t i = ia[u];
            const float4 p = pa[u];
            const double ddx = (double)p.x - q.x, ddy = (double)p.y - q.y, ddz = (double)p.z - q.z;   // src/PhotonMap.cpp:211-214
            const double d2 = ddx * ddx + ddy * ddy + ddz * ddz;
            const bool pass = valid && d2 < lim;
            const uint32_t mh = half_mask(__ballot(pass), half);
            if (pass) {
                const int pos = count + __popc(mh & below);
                hb->d[pos] = (unsigned long long)__double_as_longlong(d2);
                hb->i[pos] = i;
            }
            count += __popc(mh);
            // a half whose buffer could not take another round selects now (its bound falls to the 63rd distance so far)
            const bool full = count > PAIR_BUF - 32;
            if (__ballot(full)) {
                wave_lds_fence();
                PAIR_LAP(2)
                const unsigned long long tau = pair_reduce(hb, half, hl, full, count, lim, ok);
                PAIR_LAP(3)
                if (full) { count = KNN_K; lim = __longlong_as_double((long long)tau); }
                if (!ok) { total = 0; }          // (gives up: nothing more is read for this half)
            }
        }
    }
    wave_lds_fence();
    PAIR_LAP(2)
    ok = ok && count >= KNN_K;
    {
        const bool sel = ok;                  // (every answered query selects once more: the buffer holds >= 63 candidates below lim)
        if (__ballot(sel)) {
            const unsigned long long tau = pair_reduce(hb, half, hl, sel, count, lim, ok);
            if (sel) { count = KNN_K; lim = __longlong_as_double((long long)tau); }
        }
    }
    PAIR_LAP(3)
    if (__ballot(ok) == 0ULL) return false;
    // radiance: the half's 63 entries, two per lane, summed in 64-bit fixed point (canonical order-independent sum)
    const double r = lim;
    if (hl < 3) hb->acc[hl] = 0ULL;
    wave_lds_fence();
    const RDiv rd = rdiv_make(r);
#pragma unroll
    for (int e0 = 0; e0 < 64; e0 += 32) {
        const int e = e0 + hl;
        if (ok && e < KNN_K) {
            const double d2 = __longlong_as_double((long long)hb->d[e]);
            if (d2 < r) {   // d2 == r has weight 0
                const uint32_t i = hb->i[e];
                const float4 pp = m.pos4[i];
                const uint4 rr = m.rest[i];
                long long fr, fg, fb;
                photon_term_fx<KIND>(q.x - (double)pp.x, q.y - (double)pp.y, q.z - (double)pp.z, rr, nrm, d2, rd, fr, fg, fb);
                if (fr != 0) atomicAdd(&hb->acc[0], (unsigned long long)fr);
                if (fg != 0) atomicAdd(&hb->acc[1], (unsigned long long)fg);
                if (fb != 0) atomicAdd(&hb->acc[2], (unsigned long long)fb);
            }
        }
    }
    wave_lds_fence();
    const double norm_k = KIND == 0 ? (3.141592653589 * 2) : (3.141592653589 * 8);
    R = ((double)(long long)hb->acc[0] * (1.0 / FX_SCALE)) / (norm_k * r);
    G = ((double)(long long)hb->acc[1] * (1.0 / FX_SCALE)) / (norm_k * r);
    B = ((double)(long long)hb->acc[2] * (1.0 / FX_SCALE)) / (norm_k * r);
    wave_lds_fence();
    PAIR_LAP(4)
    return ok;
}

}  // namespace pm
