/*
 * tb_protocol.c -- plain-C host that drives libx266hip.so exactly the way the
 * Bluespec testbenches drive the BDPI golden model:
 *   DCT  (src/mkDct32.bsv:430-470): dct32_genNew; 16 x dct32_getDiff (two rows
 *        each); 256 x dct32_getDct (four coefficients each, column-major)
 *   SATD (src/mkSatd.bsv:215-252):  satd8x8_genNew; 8 x satd8x8_getDiff (one row
 *        each); satd8x8_getSatd
 * and prints every word, so a checker can compare the stream with the one the
 * real reference produces (tests/golden/bdpi_*.npz).  bsc/Bluesim is not
 * available in this image; this program stands in for mkTb's call sequence.
 *
 *   usage: tb_protocol dct <n_blocks> | satd <n_blocks>
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/x266hip.h"

int main(int argc, char **argv)
{
    if (argc != 3) {
        fprintf(stderr, "usage: %s dct|satd <n_blocks>\n", argv[0]);
        return -1;
    }
    const int n = atoi(argv[2]);
    if (!strcmp(argv[1], "dct")) {
        for (int b = 0; b < n; b++) {
            unsigned int res[32];
            dct32_genNew();
            for (int i = 0; i < 16; i++) {
                dct32_getDiff(res);
                for (int w = 0; w < 32; w++) printf("D %08X\n", res[w]);
            }
            for (int i = 0; i < 256; i++) printf("C %016llX\n", dct32_getDct());
        }
    } else if (!strcmp(argv[1], "satd")) {
        for (int b = 0; b < n; b++) {
            unsigned int res[4];
            satd8x8_genNew();
            for (int i = 0; i < 8; i++) {
                satd8x8_getDiff(res);
                for (int w = 0; w < 4; w++) printf("D %08X\n", res[w]);
            }
            printf("S %u\n", satd8x8_getSatd());
        }
    } else {
        return -1;
    }
    return 0;
}
