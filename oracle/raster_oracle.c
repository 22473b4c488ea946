/*
 * ORACLE (test infrastructure only -- never linked or imported by the product path).
 * CPU restatement of the third-party rasterizer behind SinGS' render path; see the
 * header of raster_core.inc.c for provenance and the "parity unpinned" statement.
 *
 * Build:  make -C oracle     (gcc -O2 -ffp-contract=off, no -ffast-math)
 * Exports the *_f32 functions (the checker) and *_f64 twins (finite-difference truth).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define REAL float
#define FN(n) n##_f32
#define SQRT sqrtf
#define CEIL ceilf
#define EXP expf
#define FABS fabsf
#include "raster_core.inc.c"
#undef REAL
#undef FN
#undef SQRT
#undef CEIL
#undef EXP
#undef FABS
#undef SG_BLOCK

#define REAL double
#define FN(n) n##_f64
#define SQRT sqrt
#define CEIL ceil
#define EXP exp
#define FABS fabs
#define SG_C0 SG_C0_d
#define SG_C1 SG_C1_d
#define SG_C2 SG_C2_d
#define SG_C3 SG_C3_d
#include "raster_core.inc.c"

/* ------------------------------------------------------------------ A.2 binning (integer) */

/* inclusive prefix sum of tiles_touched; returns R = num_rendered */
uint32_t sgo_scan(int P, const uint32_t *tiles_touched, uint32_t *offsets)
{
    uint32_t s = 0;
    for (int i = 0; i < P; i++) { s += tiles_touched[i]; offsets[i] = s; }
    return s;
}

/* key = (tile_id << 32) | float_bits(depth), value = Gaussian index; y outer / x inner */
void sgo_duplicate_with_keys(int P, const float *xy, const float *depths, const uint32_t *offsets,
                             const int32_t *radii, int gx, int gy,
                             uint64_t *keys, uint32_t *vals)
{
    for (int i = 0; i < P; i++) {
        if (!(radii[i] > 0)) continue;
        uint32_t off = i == 0 ? 0 : offsets[i - 1];
        int mr = radii[i];
        float px = xy[2 * i], py = xy[2 * i + 1];
        int rminx = (int)((px - (float)mr) / 16.0f), rminy = (int)((py - (float)mr) / 16.0f);
        int rmaxx = (int)((px + (float)mr + 15.0f) / 16.0f), rmaxy = (int)((py + (float)mr + 15.0f) / 16.0f);
        rminx = rminx < 0 ? 0 : rminx; rminx = rminx > gx ? gx : rminx;
        rminy = rminy < 0 ? 0 : rminy; rminy = rminy > gy ? gy : rminy;
        rmaxx = rmaxx < 0 ? 0 : rmaxx; rmaxx = rmaxx > gx ? gx : rmaxx;
        rmaxy = rmaxy < 0 ? 0 : rmaxy; rmaxy = rmaxy > gy ? gy : rmaxy;
        uint32_t dbits;
        memcpy(&dbits, &depths[i], 4);
        for (int y = rminy; y < rmaxy; y++)
            for (int x = rminx; x < rmaxx; x++) {
                uint64_t key = (uint64_t)(uint32_t)(y * gx + x);
                key <<= 32;
                key |= dbits;
                keys[off] = key; vals[off] = (uint32_t)i; off++;
            }
    }
}

/* getHigherMsb of upstream: binary search for the MSB starting at 16, +1 if n >> msb */
uint32_t sgo_higher_msb(uint32_t n)
{
    uint32_t msb = 16, step = 16;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

/* stable LSD radix sort (8-bit digits) of (key,value) pairs on bits [0, end_bit) */
void sgo_sort_pairs(uint32_t R, const uint64_t *keys_in, const uint32_t *vals_in,
                    uint64_t *keys_out, uint32_t *vals_out, int end_bit)
{
    uint64_t *ka = (uint64_t *)malloc((size_t)(R + 1) * 8), *kb = (uint64_t *)malloc((size_t)(R + 1) * 8);
    uint32_t *va = (uint32_t *)malloc((size_t)(R + 1) * 4), *vb = (uint32_t *)malloc((size_t)(R + 1) * 4);
    memcpy(ka, keys_in, (size_t)R * 8); memcpy(va, vals_in, (size_t)R * 4);
    for (int shift = 0; shift < end_bit; shift += 8) {
        int bits = end_bit - shift < 8 ? end_bit - shift : 8;
        uint32_t mask = (1u << bits) - 1;
        size_t cnt[257]; memset(cnt, 0, sizeof cnt);
        for (uint32_t i = 0; i < R; i++) cnt[((ka[i] >> shift) & mask) + 1]++;
        for (int d = 0; d < 256; d++) cnt[d + 1] += cnt[d];
        for (uint32_t i = 0; i < R; i++) {
            size_t dst = cnt[(ka[i] >> shift) & mask]++;
            kb[dst] = ka[i]; vb[dst] = va[i];
        }
        uint64_t *tk = ka; ka = kb; kb = tk;
        uint32_t *tv = va; va = vb; vb = tv;
    }
    memcpy(keys_out, ka, (size_t)R * 8); memcpy(vals_out, va, (size_t)R * 4);
    free(ka); free(kb); free(va); free(vb);
}

/* ranges[tile] = [first, last+1) in the sorted list; untouched tiles keep (0,0) */
void sgo_identify_ranges(uint32_t R, const uint64_t *keys_sorted, int ntiles, uint32_t *ranges)
{
    memset(ranges, 0, (size_t)ntiles * 2 * 4);
    for (uint32_t i = 0; i < R; i++) {
        uint32_t t = (uint32_t)(keys_sorted[i] >> 32);
        if (i == 0) ranges[2 * t] = 0;
        else {
            uint32_t pt = (uint32_t)(keys_sorted[i - 1] >> 32);
            if (t != pt) { ranges[2 * pt + 1] = i; ranges[2 * t] = i; }
        }
        if (i == R - 1) ranges[2 * t + 1] = R;
    }
}
