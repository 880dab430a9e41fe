/*
 * oracle/ref_randutil.cpp -- TEST INFRASTRUCTURE (oracle/_ref build only).
 *
 * Substitute for the reference's src/Random.cpp translation unit (which is NOT
 * linked into oracle/_ref/ref_harness): the three randutil functions declared in
 * the reference's src/Random.h:7-11 serve draws from a tape recorded by the C
 * restatement (tape replay) and kd pivots from stream kind 4.
 * A mismatch in the number or order of draws shows up as tape under/overrun.
 */
#include "Random.h"      /* the reference's own header, from -I$(REF)/src */
#include "orc_rng.h"
#include "pm_oracle.h"
#include "ref_randutil.h"

static const double *g_tape = nullptr;
static size_t g_n = 0, g_pos = 0, g_overrun = 0;
static uint64_t g_pivot_seed = 0, g_pivot_counter = 0;

void ref_tape_set(const double *v, size_t n) { g_tape = v; g_n = n; g_pos = 0; }
size_t ref_tape_pos() { return g_pos; }
size_t ref_tape_len() { return g_n; }
size_t ref_tape_overrun() { return g_overrun; }
void ref_pivot_set(uint64_t seed) { g_pivot_seed = seed; g_pivot_counter = 0; }

namespace randutil {
    void init(int64) {}
    int nextInt(int32 min, int32 max) {
        return (int) orc_pivot_next(g_pivot_seed, &g_pivot_counter, min, max);
    }
    double nextDouble() {
        if (g_pos >= g_n) { g_overrun++; return 0.5; }
        return g_tape[g_pos++];
    }
}
