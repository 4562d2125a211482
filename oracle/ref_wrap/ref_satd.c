/*
 * ref_satd.c -- wrapper that compiles the REAL src_tb/satd.c in place
 * (separate TU from ref_dct32.c: both reference files define `static mat`).
 * TEST INFRASTRUCTURE ONLY; output to oracle/_ref/ (git-ignored).
 */
#include "satd.c"           /* resolved through -I$(REF)/src_tb */

void ref_satd8x8_batch(const short *diff, unsigned int *out, unsigned long n_blocks)
{
    for (unsigned long b = 0; b < n_blocks; b++) out[b] = (unsigned int)satd8x8(diff + 64 * b);
}

const short *ref_satd8x8_last_input(void) { return mat; }
