// pm_sort.h -- stable LSD radix sort of (63-bit Morton key, 32-bit index) pairs on the key's top 30 bits, hand written for
// gfx950: three passes of 10-bit digits, three launches per pass, every launch wide.  It orders the photons of a map (index
// build, pm_build.h) and the bounced pixels' shade points (pm_kernels.h); the octree is never walked below level 10
// (pm_grid.h), so the order inside a level-10 cell is free and the low 33 key bits are not sorted on.
//
//   pass p (digit = key bits 33 + 10 p .. 42 + 10 p), blocks of SORT_TILE consecutive pairs:
//     k_sort_count    LDS histogram of the block's digits          -> table[digit][block]
//     k_sort_rows     one wave per digit: exclusive scan of its row -> table[digit][block] = pairs of that digit in earlier
//                     blocks, rowsum[digit] = pairs of that digit
//     k_sort_scatter  every block scans rowsum (1024 values) for the digits' bases, ranks its own pairs stably -- per
//                     wave by ballot matching (the lanes that hold the same digit find each other with ten ballots), across
//                     the block's waves by per-wave digit counters in LDS -- and writes each pair to its final place
//
// The count may live in device memory (n_dev): the shade-point sort is queued before the host knows how many there are.
// Then the grid is sized for the capacity and blocks beyond the count leave at once.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pm {

constexpr int SORT_BITS = 10, SORT_DIGITS = 1 << SORT_BITS, SORT_FIRST_BIT = 33, SORT_PASSES = 3;
constexpr int SORT_WAVES = 4, SORT_ROUNDS = 8;                  // a wave ranks 8 rounds of 64 consecutive pairs
constexpr int SORT_TILE = SORT_WAVES * SORT_ROUNDS * 64;         // 2048 pairs per block

__device__ inline uint32_t sort_count(uint32_t n_host, const uint32_t *n_dev) { return n_dev ? min(*n_dev, n_host) : n_host; }
__device__ inline uint32_t sort_digit(uint64_t key, int shift) { return (uint32_t)(key >> shift) & (SORT_DIGITS - 1); }

__global__ __launch_bounds__(256) void k_sort_count(const uint64_t *keys, uint32_t n_host, const uint32_t *n_dev, int shift, uint32_t nblk, uint32_t *table) {
    __shared__ uint32_t hist[SORT_DIGITS];
    const uint32_t n = sort_count(n_host, n_dev);
    const uint32_t base = blockIdx.x * SORT_TILE;
    // blocks beyond the count still clear their column: k_sort_rows reads the whole table
    for (int d = threadIdx.x; d < SORT_DIGITS; d += 256) hist[d] = 0u;
    __syncthreads();
    if (base < n) {
        const uint32_t end = min(base + (uint32_t)SORT_TILE, n);
        for (uint32_t i = base + threadIdx.x; i < end; i += 256) atomicAdd(&hist[sort_digit(keys[i], shift)], 1u);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < SORT_DIGITS; d += 256) table[(size_t)d * nblk + blockIdx.x] = hist[d];
}

__device__ inline uint32_t sort_wave_scan_incl(uint32_t x) {
    for (int s = 1; s < 64; s <<= 1) { uint32_t y = (uint32_t)__shfl_up((int)x, s); if ((int)(threadIdx.x & 63) >= s) x += y; }
    return x;
}

// one wave per digit (4 digits per block)
__global__ __launch_bounds__(256) void k_sort_rows(uint32_t *table, uint32_t nblk, uint32_t *rowsum) {
    const int lane = threadIdx.x & 63;
    const uint32_t d = blockIdx.x * 4 + (threadIdx.x >> 6);
    uint32_t *row = table + (size_t)d * nblk;
    uint32_t run = 0;
    for (uint32_t b0 = 0; b0 < nblk; b0 += 64) {
        const uint32_t b = b0 + (uint32_t)lane;
        const uint32_t v = b < nblk ? row[b] : 0u;
        const uint32_t inc = sort_wave_scan_incl(v);
        if (b < nblk) row[b] = run + inc - v;
        run += (uint32_t)__shfl((int)inc, 63);
    }
    if (lane == 0) rowsum[d] = run;
}

__global__ __launch_bounds__(256) void k_sort_scatter(const uint64_t *keys_in, const uint32_t *vals_in, uint32_t n_host, const uint32_t *n_dev, int shift,
                                                       uint32_t nblk, const uint32_t *table, const uint32_t *rowsum, uint64_t *keys_out, uint32_t *vals_out) {
    __shared__ uint32_t dbase[SORT_DIGITS];                 // first output slot of each digit (exclusive scan of rowsum), then + earlier blocks
    __shared__ uint32_t wcount[SORT_WAVES][SORT_DIGITS];    // pairs of each digit ranked so far by each wave; then the waves' bases
    __shared__ uint32_t part[4];
    const uint32_t n = sort_count(n_host, n_dev);
    const uint32_t base = blockIdx.x * SORT_TILE;
    if (base >= n) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // digits' bases: thread t owns digits 4t .. 4t+3
    {
        const uint4 r = reinterpret_cast<const uint4 *>(rowsum)[threadIdx.x];
        const uint32_t mine = r.x + r.y + r.z + r.w;
        const uint32_t inc = sort_wave_scan_incl(mine);
        if (lane == 63) part[wave] = inc;
        __syncthreads();
        uint32_t before = inc - mine;
        for (int w = 0; w < wave; w++) before += part[w];
        dbase[4 * threadIdx.x] = before; dbase[4 * threadIdx.x + 1] = before + r.x;
        dbase[4 * threadIdx.x + 2] = before + r.x + r.y; dbase[4 * threadIdx.x + 3] = before + r.x + r.y + r.z;
    }
    for (int d = threadIdx.x; d < SORT_DIGITS; d += 256) { wcount[0][d] = 0u; wcount[1][d] = 0u; wcount[2][d] = 0u; wcount[3][d] = 0u; }
    __syncthreads();
    // stable rank of every pair among the pairs of its digit in this wave: rounds in order, lanes in order
    uint64_t key[SORT_ROUNDS];
    uint32_t rank[SORT_ROUNDS];
    const unsigned long long below = (1ULL << lane) - 1ULL;
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS; r++) {
        const uint32_t i = base + (uint32_t)((wave * SORT_ROUNDS + r) * 64 + lane);
        const bool valid = i < n;
        key[r] = valid ? keys_in[i] : ~0ULL;
        const uint32_t d = sort_digit(key[r], shift);
        unsigned long long match = __ballot(valid);
#pragma unroll
        for (int b = 0; b < SORT_BITS; b++) {
            const unsigned long long set = __ballot((d >> b) & 1u);
            match &= ((d >> b) & 1u) ? set : ~set;
        }
        // every lane of a match group reads the counter, then the group's first lane advances it (one wave: lock step)
        const uint32_t before = wcount[wave][d];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (valid && (match & below) == 0ULL) wcount[wave][d] = before + (uint32_t)__popcll(match);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        rank[r] = before + (uint32_t)__popcll(match & below);
    }
    __syncthreads();
    // per digit: the waves' counts become their bases (earlier waves first), on top of the digit's base and the earlier blocks
    for (int d = threadIdx.x; d < SORT_DIGITS; d += 256) {
        uint32_t run = dbase[d] + table[(size_t)d * nblk + blockIdx.x];
#pragma unroll
        for (int w = 0; w < SORT_WAVES; w++) { const uint32_t c = wcount[w][d]; wcount[w][d] = run; run += c; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SORT_ROUNDS; r++) {
        const uint32_t i = base + (uint32_t)((wave * SORT_ROUNDS + r) * 64 + lane);
        if (i < n) {
            const uint32_t at = wcount[wave][sort_digit(key[r], shift)] + rank[r];
            keys_out[at] = key[r];
            vals_out[at] = vals_in[i];
        }
    }
}

// scratch of one sort: table[SORT_DIGITS][nblk] + rowsum[SORT_DIGITS], in 32-bit words
inline size_t sort_scratch_words(uint32_t n_host) { return (size_t)SORT_DIGITS * ((n_host + SORT_TILE - 1) / SORT_TILE) + SORT_DIGITS; }

// Sorts (keys_a, vals_a)[0 .. n) into (keys_b, vals_b); both pairs of buffers are overwritten.  n = n_host, or *n_dev when
// n_dev is not NULL (then n_host is the capacity the launches are sized for).  Nothing waits for the device.
// first_counted: the histogram of pass 0 is in the table already (the kernel that made the keys counted them: k_morton).
inline hipError_t radix_sort_pairs_30(hipStream_t stream, uint64_t *keys_a, uint32_t *vals_a, uint64_t *keys_b, uint32_t *vals_b, uint32_t *scratch,
                                      uint32_t n_host, const uint32_t *n_dev, bool first_counted = false) {
    if (n_host == 0) return hipSuccess;
    const uint32_t nblk = (n_host + SORT_TILE - 1) / SORT_TILE;
    uint32_t *table = scratch, *rowsum = scratch + (size_t)SORT_DIGITS * nblk;
    uint64_t *kin = keys_a, *kout = keys_b;
    uint32_t *vin = vals_a, *vout = vals_b;
    for (int p = 0; p < SORT_PASSES; p++) {
        const int shift = SORT_FIRST_BIT + SORT_BITS * p;
        if (p > 0 || !first_counted) hipLaunchKernelGGL(k_sort_count, dim3(nblk), dim3(256), 0, stream, kin, n_host, n_dev, shift, nblk, table);
        hipLaunchKernelGGL(k_sort_rows, dim3(SORT_DIGITS / 4), dim3(256), 0, stream, table, nblk, rowsum);
        hipLaunchKernelGGL(k_sort_scatter, dim3(nblk), dim3(256), 0, stream, kin, vin, n_host, n_dev, shift, nblk, table, rowsum, kout, vout);
        uint64_t *tk = kin; kin = kout; kout = tk;
        uint32_t *tv = vin; vin = vout; vout = tv;
    }
    // three passes: a -> b -> a -> b
    return hipGetLastError();
}

}  // namespace pm
