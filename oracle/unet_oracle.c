/*
 * TEST INFRASTRUCTURE — see unet_oracle_body.h.  Builds liboracle.so with two
 * instantiations of every function: *_f32 (float arithmetic, the reference's dtype)
 * and *_f64 (double arithmetic, the ground truth that fp32 results are judged by;
 * the reference's own fp32 CPU path is not bit-deterministic across
 * thread counts, so tolerances are normalised and anchored on the f64 run).
 */
#include <math.h>
#include <stdlib.h>
#include <stddef.h>
#include <string.h>

#define REAL float
#define SUFFIX _f32
#include "unet_oracle_body.h"
#undef REAL
#undef SUFFIX

#define REAL double
#define SUFFIX _f64
#include "unet_oracle_body.h"
#undef REAL
#undef SUFFIX

/* functions.input_size_compute (functions.py:121-146): smallest even L >= 20 with
 * 16L-124 >= original; input = 16L+60, output = 16L-124. */
void oracle_input_size_compute(int original, int *input_size, int *output_size)
{
    int L = 20;
    while (16 * L - 124 < original) L += 2;
    *input_size = 16 * L + 60;
    *output_size = 16 * L - 124;
}
