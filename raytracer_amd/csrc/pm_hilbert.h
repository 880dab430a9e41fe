// pm_hilbert.h -- 3-D Hilbert curve index (21 bits per axis -> 63-bit key), after J. Skilling,
// "Programming the Hilbert curve" (AIP Conf. Proc. 707, 2004): axes -> transposed index -> key.
// Photons are sorted by this key: unlike Morton order, consecutive keys are always neighbouring
// cells, so every run of 64 consecutive photons (a leaf) is spatially connected and its box compact.
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

template <int BITS>
PM_HD inline uint64_t hilbert3(uint32_t x, uint32_t y, uint32_t z) {
    uint32_t X[3] = {x, y, z};
    const uint32_t M = 1u << (BITS - 1);
    for (uint32_t Q = M; Q > 1; Q >>= 1) {
        uint32_t P = Q - 1;
        for (int i = 0; i < 3; i++) {
            if (X[i] & Q) X[0] ^= P;
            else { uint32_t t = (X[0] ^ X[i]) & P; X[0] ^= t; X[i] ^= t; }
        }
    }
    for (int i = 1; i < 3; i++) X[i] ^= X[i - 1];
    uint32_t t = 0;
    for (uint32_t Q = M; Q > 1; Q >>= 1) if (X[2] & Q) t ^= Q - 1;
    for (int i = 0; i < 3; i++) X[i] ^= t;
    return (spread21(X[0]) << 2) | (spread21(X[1]) << 1) | spread21(X[2]);
}

}  // namespace pm
