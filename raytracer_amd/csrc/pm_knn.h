// pm_knn.h -- shared pieces of the k-NN radiance gather (k = 63): the per-wave candidate buffer, the exact
// 63rd-smallest select, the per-photon radiance term, and the one-wave-per-query gather.
//
// Replaces find_nearest_photons + insert (reference src/PhotonMap.cpp:142-255) and the radiance loops that
// consume their result (src/Raytrace.cpp:97-131, 136-170).  The search itself is in pm_grid.h (one wave per
// query, linear octree) and pm_tile.h (one lane per query, groups of neighbouring queries).
//
// The reference keeps a 63-entry max-heap per query.  Here:
//   * photons closer than the current bound `lim` are appended, ballot-compacted, to a per-wave LDS
//     candidate buffer (no ordering is maintained);
//   * when the buffer holds >= 63 entries for the first time, and whenever it fills up, the exact
//     63rd-smallest squared distance is found with a histogram + wave-wide binary radix select on the IEEE
//     bit patterns (mostly scalar-unit work: ballots + popcounts), `lim` drops to it and the buffer is
//     compacted to the 63 survivors;
//   * the radiance sum is order independent: each term is truncated to a multiple of 2^-56 and summed
//     as 64-bit integers (the canonical sum defined in DESIGN.md), so no sort is needed.
// Photons at exactly the k-th distance have weight (1 - d2/r)^2 == 0, so ties cannot change the sum.
#pragma once
#include "pm_device.h"

namespace pm {

constexpr int KNN_K = 63;
constexpr double MAX_R2 = 100.0;     // MAX_PHOTON_RADIUS, reference src/Raytrace.cpp:13 (compared with squared distance)
#ifndef PM_KBUF
#define PM_KBUF 256
#endif
constexpr int KBUF = PM_KBUF;        // candidate buffer entries per wave

struct MapView {
    long long n;                 // photons
    const float4 *pos4;          // sorted order: x, y, z, bits(emission index)
    const uint4 *rest;           // sorted order: bits(dx), bits(dy), bits(dz), power[0..3]
    // linear octree (pm_grid.h): per-level cell start tables over the key-sorted photons
    const uint64_t *keys;        // sorted 63-bit Morton keys (levels below lmax are resolved by binary search)
    const uint32_t *cells;       // first photon of every cell of level lmax (8^lmax + 1 entries); coarser cells are aligned runs of it
    const uint64_t *cbox;        // level l occupies [box_off(l), box_off(l) + 8^l): occupied 1/1024 slices of each cell (all ones: empty)
    int lmax;                    // finest level
    float glo[3], gext;          // cubic quantisation grid: fine index = (p - glo) / gext * 2^21
    double gu;                   // size of a fine unit: gext / 2^21
    const struct GridParams *grid;  // the same five numbers in device memory, written by the build; kernels copy them in on entry
};
struct GridParams { float glo[3], gext; double gu; };
// a kernel's by-value MapView arrives without the grid numbers (the host never reads them back): fetch them (scalar loads)
__device__ inline void load_grid(MapView &m) {
    if (m.n <= 0) return;
    m.glo[0] = m.grid->glo[0]; m.glo[1] = m.grid->glo[1]; m.glo[2] = m.grid->glo[2]; m.gext = m.grid->gext; m.gu = m.grid->gu;
}
__host__ __device__ inline long long box_off(int l) { return ((1LL << (3 * l)) - 1) / 7; }

// per-wave LDS working set
struct KnnBuf {
    unsigned long long d[KBUF];  // squared distances as IEEE bit patterns (>= 0, so they order like integers)
    uint32_t i[KBUF];            // photon index in sorted order
    unsigned long long acc[4];   // fixed-point radiance accumulators (3 used)
    uint32_t hist[64];           // d2 histogram of the select
    unsigned long long sel[64];  // the entries of the histogram bin in which the count crosses 63
    uint32_t cellpre[128];       // scan list of the grid query (pm_grid.h)
};

// ---- wave-wide reductions on the DPP network (no LDS round trips); result is wave-uniform
#define PM_DPP_STEP(T, OP, CTRL, RM)                                                                    \
    { int t_ = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, RM, 0xf, false); \
      v = OP(v, __builtin_bit_cast(T, t_)); }
__device__ inline float fmin_op(float a, float b) { return fminf(a, b); }
__device__ inline uint32_t or_op(uint32_t a, uint32_t b) { return a | b; }
__device__ inline uint32_t and_op(uint32_t a, uint32_t b) { return a & b; }
__device__ inline uint32_t umax_op(uint32_t a, uint32_t b) { return a > b ? a : b; }
#define PM_WAVE_REDUCE(NAME, T, OP)                                                     \
    __device__ inline T NAME(T v) {                                                     \
        PM_DPP_STEP(T, OP, 0xb1, 0xf)  /* quad_perm [1,0,3,2] */                       \
        PM_DPP_STEP(T, OP, 0x4e, 0xf)  /* quad_perm [2,3,0,1] */                       \
        PM_DPP_STEP(T, OP, 0x124, 0xf) /* row_ror:4 */                                 \
        PM_DPP_STEP(T, OP, 0x128, 0xf) /* row_ror:8 */                                 \
        PM_DPP_STEP(T, OP, 0x142, 0xa) /* row_bcast:15 -> rows 1,3 */                  \
        PM_DPP_STEP(T, OP, 0x143, 0xc) /* row_bcast:31 -> rows 2,3 */                  \
        return __builtin_bit_cast(T, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63)); \
    }
PM_WAVE_REDUCE(wave_min_f32, float, fmin_op)
PM_WAVE_REDUCE(wave_or_u32, uint32_t, or_op)
PM_WAVE_REDUCE(wave_and_u32, uint32_t, and_op)

__device__ inline int lane_rank(unsigned long long mask) {   // set bits of mask below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
__device__ inline unsigned long long readlane_u64(unsigned long long v, int l) {
    uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}

struct KnnState {
    double lim;          // wave-uniform: only photons with d2 < lim are collected (starts at MAX_R2)
    int count;           // entries in the buffer
    bool reduced;        // lim is an exact 63rd-smallest (set by a reduce, cleared when a cell bound lowers lim)
    bool any_reduce;     // some reduce has happened: lim is tight from here on
    int lane;
    KnnBuf *buf;
    // wave-uniform work counters (scalar registers): node visits, leaf visits, reduces, appended candidates
    uint32_t c_nodes, c_leaves, c_reduces, c_cands;
#ifdef PM_STAMPS
    unsigned long long t_node, t_leaf, t_red;   // diagnostic build only: shader cycles per section
#endif
};
#ifdef PM_STAMPS
#define PM_T0 unsigned long long t0_ = __builtin_amdgcn_s_memtime();
#define PM_T1(field) st.field += __builtin_amdgcn_s_memtime() - t0_;
#else
#define PM_T0
#define PM_T1(field)
#endif

// Exact 63rd-smallest of the first `count` (>= 63) buffer entries; compacts the buffer to the 63
// entries that matter (all with d2 < tau plus as many d2 == tau as needed) and lowers lim to tau.
// Plain values in and out (not the state struct): the function is deliberately not inlined -- it runs
// 2-3 times per query but would otherwise be cloned into every level of the traversal and drive its
// register count -- and a by-reference state would be forced into scratch memory.
__device__ inline uint32_t wave_scan_incl_u32(uint32_t x) {   // inclusive prefix sum over the 64 lanes (DPP)
    uint32_t r = x;
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);   // row_shr:1
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);   // row_shr:2
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x113, 0xf, 0xf, false);   // row_shr:3
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x114, 0xf, 0xe, false);   // row_shr:4, banks 1-3
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x118, 0xf, 0xc, false);   // row_shr:8, banks 2-3
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1,3
    r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2,3
    return r;
}

__device__ __noinline__ unsigned long long knn_reduce_impl(KnnBuf *buf_in, int count_in, int lim_lo_in, int lim_hi_in) {
    constexpr int NS = KBUF / 64;
    const int lane = threadIdx.x & 63;
    // arguments of a real call arrive in vector registers; all are wave-uniform, and saying so keeps the
    // whole select (masks, counters, loop control) on the scalar unit
    const int count = __builtin_amdgcn_readfirstlane(count_in);
    const double lim = __hiloint2double(__builtin_amdgcn_readfirstlane(lim_hi_in), __builtin_amdgcn_readfirstlane(lim_lo_in));
    // the buffer lives in LDS: the low 32 bits of its generic address are the LDS offset; going back to an
    // LDS-qualified pointer turns the flat loads/stores of this out-of-line function into ds_read / ds_write
    typedef __attribute__((address_space(3))) KnnBuf LdsKnnBuf;
    LdsKnnBuf *buf = (LdsKnnBuf *)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(unsigned long long)buf_in);
    buf->hist[lane] = 0u;
    wave_lds_fence();
    unsigned long long v[NS];
    uint32_t idx[NS], bin[NS];
    unsigned long long act[NS];
    // Stage 1: 64-bin histogram of d2 over [0, lim).  The bin is a monotone function of d2, so the bin in
    // which the running count crosses 63 holds the answer and only its few entries need an exact select.
    // (Photons spread evenly over a surface make d2 itself uniform, so the bins fill evenly.)
    const double scale = (double)(64.0f * __frcp_rn((float)lim));
#pragma unroll
    for (int s = 0; s < NS; s++) {
        bool valid = lane + 64 * s < count;
        v[s] = 0ULL; idx[s] = 0u; act[s] = 0ULL; bin[s] = 0u;
        if (64 * s >= count) continue;   // wave-uniform: unused slots cost nothing below either
        v[s] = valid ? buf->d[lane + 64 * s] : 0ULL;
        idx[s] = valid ? buf->i[lane + 64 * s] : 0u;
        if (valid) {
            bin[s] = (uint32_t)fmin(__longlong_as_double((long long)v[s]) * scale, 63.0);
            atomicAdd((uint32_t *)&buf->hist[bin[s]], 1u);
        }
    }
    wave_lds_fence();
    uint32_t h = buf->hist[lane];
    uint32_t inc = wave_scan_incl_u32(h);
    int bstar = __ffsll((long long)__ballot(inc >= (uint32_t)KNN_K)) - 1;   // exists: count >= 63
    int need = KNN_K - (int)((uint32_t)__builtin_amdgcn_readlane((int)inc, bstar) - (uint32_t)__builtin_amdgcn_readlane((int)h, bstar));
    int remaining = 0;
#pragma unroll
    for (int s = 0; s < NS; s++) {
        if (64 * s >= count) continue;
        act[s] = __ballot(lane + 64 * s < count && bin[s] == (uint32_t)bstar);
        remaining += __popcll(act[s]);
    }
    // Stage 2: exact select of the need-th smallest among the entries of that bin.  The bin holds a handful of entries
    // (a 64th of the buffer when the distances are spread evenly): they are gathered into LDS, every lane takes one and
    // counts how many are smaller and how many are not larger -- one broadcast read and two compares per entry -- and
    // the entry whose rank interval holds `need` is the answer.  (A bin of more than 64 entries -- hundreds of equal
    // distances -- falls back to a binary radix select on the IEEE bit patterns: ballots + scalar popcounts.)
    unsigned long long tau = 0ULL;
    if (remaining <= 64) {
        int run = 0;
#pragma unroll
        for (int s = 0; s < NS; s++) {
            if (act[s] == 0ULL) continue;
            if ((act[s] >> lane) & 1ULL) buf->sel[run + lane_rank(act[s])] = v[s];
            run += __popcll(act[s]);
        }
        wave_lds_fence();
        const unsigned long long x = lane < remaining ? buf->sel[lane] : ~0ULL;
        int less = 0, leq = 0;
        for (int k = 0; k < remaining; k++) {
            const unsigned long long y = buf->sel[k];
            less += y < x ? 1 : 0;
            leq += y <= x ? 1 : 0;
        }
        const unsigned long long hitm = __ballot(lane < remaining && less < need && need <= leq);   // every such lane holds the same value
        tau = readlane_u64(x, __ffsll((long long)hitm) - 1);
    } else {
        uint32_t o_hi = 0, o_lo = 0, a_hi = 0xFFFFFFFFu, a_lo = 0xFFFFFFFFu;
        unsigned long long diff = 0ULL;
#pragma unroll
        for (int s = 0; s < NS; s++)
            if ((act[s] >> lane) & 1ULL) { o_hi |= (uint32_t)(v[s] >> 32); o_lo |= (uint32_t)v[s]; a_hi &= (uint32_t)(v[s] >> 32); a_lo &= (uint32_t)v[s]; }
        diff = (((unsigned long long)(wave_or_u32(o_hi) ^ wave_and_u32(a_hi))) << 32) | (wave_or_u32(o_lo) ^ wave_and_u32(a_lo));
        if (diff != 0ULL) {
            for (int b = 63 - __clzll((long long)diff); b >= 0 && remaining > 1; b--) {
                uint32_t bit = 1u << (b & 31);
                unsigned long long z[NS];
                int zero_cnt = 0;
#pragma unroll
                for (int s = 0; s < NS; s++) {
                    z[s] = 0ULL;
                    if (act[s] == 0ULL) continue;
                    uint32_t w = b >= 32 ? (uint32_t)(v[s] >> 32) : (uint32_t)v[s];
                    z[s] = __ballot((w & bit) == 0u) & act[s];
                    zero_cnt += __popcll(z[s]);
                }
                if (need <= zero_cnt) {
#pragma unroll
                    for (int s = 0; s < NS; s++) act[s] = z[s];
                    remaining = zero_cnt;
                } else {
                    need -= zero_cnt;
#pragma unroll
                    for (int s = 0; s < NS; s++) act[s] &= ~z[s];
                    remaining -= zero_cnt;
                }
            }
        }
        // every still-active entry equals the answer
#pragma unroll
        for (int s = NS - 1; s >= 0; s--)
            if (act[s]) tau = readlane_u64(v[s], __ffsll((long long)act[s]) - 1);
    }
    // compact: all entries < tau, then enough entries == tau to make 63
    unsigned long long lt[NS], eq[NS];
    int n_lt = 0;
#pragma unroll
    for (int s = 0; s < NS; s++) {
        bool valid = lane + 64 * s < count;
        lt[s] = eq[s] = 0ULL;
        if (64 * s >= count) continue;
        lt[s] = __ballot(valid && v[s] < tau);
        eq[s] = __ballot(valid && v[s] == tau);
        n_lt += __popcll(lt[s]);
    }
    int run_lt = 0, run_eq = 0;
#pragma unroll
    for (int s = 0; s < NS; s++) {
        bool valid = lane + 64 * s < count;
        if (64 * s >= count) continue;
        int pos = -1;
        if (valid && v[s] < tau) pos = run_lt + lane_rank(lt[s]);
        else if (valid && v[s] == tau) { int e = n_lt + run_eq + lane_rank(eq[s]); if (e < KNN_K) pos = e; }
        if (pos >= 0) { buf->d[pos] = v[s]; buf->i[pos] = idx[s]; }
        run_lt += __popcll(lt[s]);
        run_eq += __popcll(eq[s]);
    }
    wave_lds_fence();
    return tau;
}
__device__ inline void knn_reduce(KnnState &st) {
    PM_T0
    st.c_reduces++;
    unsigned long long tau = knn_reduce_impl(st.buf, st.count, __double2loint(st.lim), __double2hiint(st.lim));
    tau = readlane_u64(tau, 0);   // the result of a real call comes back in vector registers; it is wave-uniform
    st.count = KNN_K;
    st.lim = __longlong_as_double((long long)tau);
    st.reduced = true;
    st.any_reduce = true;
    PM_T1(t_red)
}

// x / 255.0 for an integer 0 <= x <= 255, correctly rounded, without the division: the product with the rounded
// reciprocal is off by one ulp for 24 of the 256 values, one Newton step with exact residual (fma) repairs all of them
// (checked exhaustively in exact arithmetic: tests/test_oracle_golden.py::test_byte_over_255_identity)
__device__ inline double byte_over_255(int x) {
    const double inv = 1.0 / 255.0, xd = (double)x;
    double q = xd * inv;
    double res = __builtin_fma(-q, 255.0, xd);
    return __builtin_fma(res, inv, q);
}

// x / d for one d and many x.  The compiler's IEEE division is v_div_scale (both operands), v_rcp_f64, two Newton steps on
// the reciprocal, then per quotient a product, an exact residual and a final fma (v_div_fmas), and v_div_fixup; the scaling
// and the fix-up are the identity for ordinary operands.  Here the part that depends on d alone is done once; the quotient
// is then the compiler's own, bit for bit, for d in [2^-100, 2^100] and 2^-900 <= x <= d; for smaller x it may differ from
// it below 2^-800, which a caller that forms 1 - x / d cannot see.  Any other d takes the plain division.
struct RDiv { double d, y; bool ok; };
__device__ inline RDiv rdiv_make(double d) {
    RDiv r;
    r.d = d;
    r.ok = d >= 0x1p-100 && d <= 0x1p100;
    double y = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-d, y, 1.0);
    r.y = __builtin_fma(y, e, y);
    return r;
}
__device__ inline double rdiv(const RDiv &r, double x) {
    if (__builtin_expect(!r.ok, 0)) return x / r.d;
    const double q = x * r.y;
    return __builtin_fma(__builtin_fma(-r.d, q, x), r.y, q);
}

// One photon's term of the radiance sums: src/Raytrace.cpp:100-125 (global), :139-164 (caustic).
// s = shade point - photon position (the reference's `dist`), pd2 = |s|^2, r = the 63rd-smallest pd2 (rd: its RDiv).
template <int KIND>
__device__ inline void photon_term_s(double sx, double sy, double sz, const uint4 &rr, const V3 &nrm, double pd2, const RDiv &rd,
                                     double &tr, double &tg, double &tb) {
    tr = tg = tb = 0.0;
    float pdx = __uint_as_float(rr.x), pdy = __uint_as_float(rr.y), pdz = __uint_as_float(rr.z);
    double d = -(nrm.x * (double)pdx + nrm.y * (double)pdy + nrm.z * (double)pdz);
    if (d <= 0) return;
    // The plane filter with its dist.z quirk: s is scaled by k = (s . n) / (3 s_z^2) and the photon dropped
    // when |k s|^2 > 0.1; NaN compares false -> photon kept.  Only the verdict is used.  In real numbers |k s|^2 =
    // num^2 |s|^2 / den^2, and the reference's rounded value is within a few 2^-53 of that (relative) as long as no
    // intermediate overflows, underflows, is infinite or NaN; so is the comparison of A = num^2 |s|^2 with 0.1 B, B = den^2,
    // when num^2, A and B are normal numbers (`usual`: the bounds below keep every product in range).  A verdict that
    // clears a 2^-30 band is therefore the reference's, without the division; inside the band, and for every unusual
    // operand, the reference's own arithmetic decides.  Two unusual operands are common and settled by rule:
    //   s_z == 0 exactly (shade point and photon both on the back wall): k is a NaN or an infinity, k s_z a NaN: kept;
    //   num == 0 exactly (shade point and photon on the same axis-aligned wall): k is a zero or a NaN, and so is k s: kept.
    const double num = sx * nrm.x + sy * nrm.y + sz * nrm.z, den = sz * sz + sz * sz + sz * sz;
    const double m = num * num, A = m * pd2, B = den * den;
    // (bitwise, not short-circuit: one rare branch instead of a ladder of them)
    const bool usual = (den >= 0x1p-300) & (B <= 0x1p600) & (m >= 0x1p-500) & (A >= 0x1p-960) & (A <= 0x1p600);
    const bool keep = (sz == 0.0) | (num == 0.0) | (usual & (A < B * (0.1 * (1.0 - 0x1p-30))));
    bool drop = usual & (A > B * (0.1 * (1.0 + 0x1p-30)));
    if (__builtin_expect(!(keep | drop), 0)) {
        const double k = num / den;
        const double kx = sx * k, ky = sy * k, kz = sz * k;
        drop = kx * kx + ky * ky + kz * kz > 0.1;
    }
    if (drop) return;
    double filter = (1 - rdiv(rd, pd2));
    uint32_t pw = rr.w;
    int p0 = pw & 0xFF, p1 = (pw >> 8) & 0xFF, p2 = (pw >> 16) & 0xFF;
    if (KIND == 0) {
        filter = filter * filter;
        tr = d * byte_over_255(p0) * filter;     // d * (power[0] / 255.0) * filter
        tg = d * byte_over_255(p1) * filter;
        tb = d * byte_over_255(p2) * filter;
    } else {
        filter = filter * filter * filter * filter;
        // raytrace::min(int32,int32) on a truncated double: min((int)((double)(p / 255.0f) + 0.5), 1) is 1 for
        // p >= 128 and 0 below (checked for all 256 values, same test)
        tr = d * filter * (double)(p0 >= 128);
        tg = d * filter * (double)(p1 >= 128);
        tb = d * filter * (double)(p2 >= 128);
    }
}
template <int KIND>
__device__ inline void photon_term(const float4 &pp, const uint4 &rr, const V3 &p, const V3 &nrm, double pd2, double r,
                                   double &tr, double &tg, double &tb) {
    photon_term_s<KIND>(p.x - (double)pp.x, p.y - (double)pp.y, p.z - (double)pp.z, rr, nrm, pd2, rdiv_make(r), tr, tg, tb);
}

constexpr double FX_SCALE = 72057594037927936.0;   // 2^56

// The same term as its three addends of the canonical sum: trunc(term * 2^56), per channel.  For the caustic map the three
// channels are one value or nothing (the power factor is 0 or 1), so it is converted once.
template <int KIND>
__device__ inline void photon_term_fx(double sx, double sy, double sz, const uint4 &rr, const V3 &nrm, double pd2, const RDiv &rd,
                                      long long &fr, long long &fg, long long &fb) {
    double tr, tg, tb;
    photon_term_s<KIND>(sx, sy, sz, rr, nrm, pd2, rd, tr, tg, tb);
    if (KIND == 0) {
        fr = (long long)(tr * FX_SCALE); fg = (long long)(tg * FX_SCALE); fb = (long long)(tb * FX_SCALE);
    } else {
        // tr, tg, tb are each v = d * filter (times 1.0) or +0.0; max picks v when any channel has it
        const long long v = (long long)(fmax(tr, fmax(tg, tb)) * FX_SCALE);
        fr = tr != 0.0 ? v : 0; fg = tg != 0.0 ? v : 0; fb = tb != 0.0 ? v : 0;
    }
}

__device__ inline int grid_query(KnnState &st, KnnBuf *buf, const MapView &m, const V3 &q, double &r, double lim0);   // pm_grid.h

// A run of neighbouring queries hands the ball of one query to the next: 63 photons lie within sqrt(r) of `at`.
struct KnnCarry { V3 at; double r; bool full; };
__device__ inline double carry_bound(const KnnCarry &c, const V3 &p) {
    if (!c.full) return MAX_R2;
    double dx = p.x - c.at.x, dy = p.y - c.at.y, dz = p.z - c.at.z;
    double reach = sqrt(c.r) + sqrt(dx * dx + dy * dy + dz * dz);
    return reach * reach * (1.0 + 1e-12) + 1e-300;   // rounded up: the collection test is '<'
}

// query + radiance estimate; result valid in every lane
template <int KIND>
__device__ inline void knn_gather(KnnBuf *buf, const MapView &m, const V3 &p, const V3 &nrm, double &R, double &G, double &B, KnnCarry &carry) {
    KnnState st;
    double r;
    int found = grid_query(st, buf, m, p, r, carry_bound(carry, p));
    carry.at = p; carry.r = r; carry.full = found == KNN_K;
    R = G = B = 0.0;
    if (found == 0) return;
    if (st.lane < 3) buf->acc[st.lane] = 0ULL;
    wave_lds_fence();
    const RDiv rd = rdiv_make(r);
    if (st.lane < found) {
        double d2 = __longlong_as_double((long long)buf->d[st.lane]);
        if (d2 < r) {   // d2 == r has weight 0
            uint32_t i = buf->i[st.lane];
            float4 pp = m.pos4[i];
            uint4 rr = m.rest[i];
            long long fr, fg, fb;
            photon_term_fx<KIND>(p.x - (double)pp.x, p.y - (double)pp.y, p.z - (double)pp.z, rr, nrm, d2, rd, fr, fg, fb);
            // canonical order-independent sum: 64-bit fixed point in LDS
            if (fr != 0) atomicAdd(&buf->acc[0], (unsigned long long)fr);
            if (fg != 0) atomicAdd(&buf->acc[1], (unsigned long long)fg);
            if (fb != 0) atomicAdd(&buf->acc[2], (unsigned long long)fb);
        }
    }
    wave_lds_fence();
    const double norm_k = KIND == 0 ? (3.141592653589 * 2) : (3.141592653589 * 8);
    R = ((double)(long long)buf->acc[0] * (1.0 / FX_SCALE)) / (norm_k * r);
    G = ((double)(long long)buf->acc[1] * (1.0 / FX_SCALE)) / (norm_k * r);
    B = ((double)(long long)buf->acc[2] * (1.0 / FX_SCALE)) / (norm_k * r);
    wave_lds_fence();
}

// ---- sorted output for the pm_knn_query entry point only: ascending (d2, emission index)
struct Best { double d; uint32_t o; uint32_t i; };
__device__ inline bool key_less(double da, uint32_t oa, double db, uint32_t ob) { return da < db || (da == db && oa < ob); }
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

}  // namespace pm
