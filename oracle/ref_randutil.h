/* oracle/ref_randutil.h -- TEST INFRASTRUCTURE: controls for the tape-replay randutil. */
#pragma once
#include <stddef.h>
#include <stdint.h>
void ref_tape_set(const double *v, size_t n);
size_t ref_tape_pos();
size_t ref_tape_len();
size_t ref_tape_overrun();
void ref_pivot_set(uint64_t seed);
