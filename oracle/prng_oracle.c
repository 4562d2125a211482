/*
 * prng_oracle.c -- synthetic residual stream + checksums (TEST INFRASTRUCTURE).
 *
 * The reference draws each stimulus sample as (rand()&0xFF) - (rand()&0xFF)
 * (src_tb/dct32.c:191-193, src_tb/satd.c:132-134): a triangular distribution
 * on [-255, 255], i.e. a 9-bit residual.  glibc rand() is neither portable nor
 * parallel, so batches use a counter-based SplitMix64 with the same
 * distribution; libx266hip's xFillResidual kernel implements the identical
 * function on the device.
 */
#include "x266_oracle.h"

static inline uint64_t splitmix64_at(uint64_t seed, uint64_t index)
{
    uint64_t z = seed + (index + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void orc_fill_residual(int16_t *dst, size_t n_samples, uint64_t seed, uint64_t first_index)
{
    for (size_t i = 0; i < n_samples; i++) {
        const uint64_t r = splitmix64_at(seed, first_index + i);
        dst[i] = (int16_t)((int)(r & 0xFF) - (int)((r >> 8) & 0xFF));
    }
}

uint64_t orc_checksum64(const void *data, size_t n_bytes)
{
    const uint8_t *p = (const uint8_t *)data;
    uint64_t h = 0xCBF29CE484222325ull;
    size_t i = 0;
    for (; i + 8 <= n_bytes; i += 8) {
        uint64_t w = 0;
        for (int k = 0; k < 8; k++) w |= (uint64_t)p[i + k] << (8 * k);
        h = (h ^ w) * 0x100000001B3ull;
    }
    for (; i < n_bytes; i++) h = (h ^ p[i]) * 0x100000001B3ull;
    return h;
}

uint64_t orc_sum_u16(const int16_t *data, size_t n)
{
    uint64_t s = 0;
    for (size_t i = 0; i < n; i++) s += (uint16_t)data[i];
    return s;
}
