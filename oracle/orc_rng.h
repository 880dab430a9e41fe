/*
 * oracle/orc_rng.h -- TEST INFRASTRUCTURE (oracle). Not part of the product.
 *
 * Counter-based random stream contract ("RNG contract v1").
 *
 * The reference draws every random number from one global, sequential,
 * unsynchronised engine (src/Random.cpp:8-27) -- SURVEY F7.  That cannot be
 * parallelised, so the product defines counter-based streams and the oracle
 * (and, by tape replay, the compiled reference) consume exactly the same
 * values in the reference's draw order (src/Raytrace.cpp:39,182-189,288-289;
 * src/PhotonMap.cpp:290-298,312).
 *
 *   mix64       = SplitMix64 finaliser
 *   stream key  K  = mix64(seed + GOLD * kind)
 *   item key    IK = mix64(K ^ (item * ODD1 + ODD2))
 *   draw j      u  = mix64(IK + (j + 1) * GOLD);  double = (u >> 11) * 2^-53
 *
 * kinds:  1 = pixel streams        item = pixel_index * 32 + node_id
 *                                   node 0        : j=0 x-jitter, j=1 y-jitter
 *                                   node n>=1     : j=0 dt of that traceRay call,
 *                                                   j=1+3i,2+3i,3+3i = shadow ray i (x0, z0, dt)
 *                                   node ids      : root 1, refract child 2n, reflect child 2n+1
 *         2 = global emission      item = attempt index; j=0 x0, 1 z0, 2 u(sin^2), 3 psi, 4+h = chance at hit h
 *         3 = caustic emission     same layout
 *         4 = kd pivot (oracle only; tree shape is not part of the contract)
 */
#ifndef ORC_RNG_H
#define ORC_RNG_H
#include <stdint.h>

#define ORC_GOLD 0x9E3779B97F4A7C15ULL
#define ORC_ODD1 0xD1B54A32D192ED03ULL
#define ORC_ODD2 0x8CB92BA72F3D8DD7ULL

#define ORC_KIND_PIXEL 1u
#define ORC_KIND_EMIT_GLOBAL 2u
#define ORC_KIND_EMIT_CAUSTIC 3u
#define ORC_KIND_KD_PIVOT 4u

static inline uint64_t orc_mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27; z *= 0x94D049BB133111EBULL;
    z ^= z >> 31; return z;
}
static inline uint64_t orc_stream_key(uint64_t seed, uint32_t kind) {
    return orc_mix64(seed + ORC_GOLD * (uint64_t)kind);
}
static inline uint64_t orc_item_key(uint64_t K, uint64_t item) {
    return orc_mix64(K ^ (item * ORC_ODD1 + ORC_ODD2));
}
static inline uint64_t orc_draw_u64(uint64_t IK, uint32_t j) {
    return orc_mix64(IK + ((uint64_t)j + 1ULL) * ORC_GOLD);
}
static inline double orc_draw(uint64_t IK, uint32_t j) {
    return (double)(orc_draw_u64(IK, j) >> 11) * (1.0 / 9007199254740992.0);
}
#endif
