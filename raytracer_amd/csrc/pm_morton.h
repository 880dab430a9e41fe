// pm_morton.h -- 63-bit Morton keys (21 bits per axis).  Photons are sorted by this key, so every aligned
// cell of every octree level is ONE contiguous range of the sorted array (pm_grid.h).
#pragma once
#include <stdint.h>
#ifdef __HIPCC__
#define PM_HD __host__ __device__
#else
#define PM_HD
#endif

namespace pm {

PM_HD inline uint64_t spread21(uint64_t v) {   // 21 bits -> every third bit
    v &= 0x1FFFFFULL;
    v = (v | v << 32) & 0x1F00000000FFFFULL;
    v = (v | v << 16) & 0x1F0000FF0000FFULL;
    v = (v | v << 8) & 0x100F00F00F00F00FULL;
    v = (v | v << 4) & 0x10C30C30C30C30C3ULL;
    v = (v | v << 2) & 0x1249249249249249ULL;
    return v;
}

}  // namespace pm
