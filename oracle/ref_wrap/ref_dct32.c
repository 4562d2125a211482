/*
 * ref_dct32.c -- thin wrapper that compiles the REAL reference golden model
 * (src_tb/dct32.c) from where it lies under /root/reference.  No reference
 * source is copied: the file is pulled in by #include with -I<ref>/src_tb.
 * Output goes to oracle/_ref/ only (git-ignored).  TEST INFRASTRUCTURE ONLY.
 *
 * The reference's workers are `static` and print debug lines for every row of
 * the second pass (src_tb/dct32.c:84-106), hence inclusion + a muted printf.
 */
#include <stdio.h>
#include <stdlib.h>
#undef printf
#define printf(...) ((void)0)
#include "dct32.c"          /* resolved through -I$(REF)/src_tb */
#undef printf

/* n blocks of 32x32 int16, exactly dct32_genNew()'s two calls (dct32.c:197-198) */
void ref_dct32_fwd(const short *in, short *out, unsigned long n_blocks)
{
    short coef[32 * 32];
    for (unsigned long b = 0; b < n_blocks; b++) {
        partialButterfly32(in + b * 1024, coef, 4, 32);
        partialButterfly32(coef, out + b * 1024, 11, 32);
    }
}

void ref_dct32_pass(const short *src, short *dst, int shift, int line)
{
    partialButterfly32(src, dst, shift, line);
}

const short *ref_dct32_table(void) { return &g_t32[0][0]; }

/* expose the BDPI stimulus block so fixtures can record it */
const short *ref_dct32_last_input(void)  { return mat; }
const short *ref_dct32_last_output(void) { return dct; }
