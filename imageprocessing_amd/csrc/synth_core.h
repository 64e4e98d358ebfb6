// synth_core.h -- integer-only recipe of the synthetic PISCO-like frames.
// Shared by the device renderer (piscoseg.hip) and mirrored in NumPy (imageprocessing_amd/synth.py);
// both must give identical bytes.  No counterpart in the reference (it has no data generator).
#pragma once
#include <stdint.h>
#include "pseg_core.h"

PSEG_HD uint64_t synth_hash(uint64_t seed, uint64_t frame, uint64_t pix)
{
    uint64_t z = seed * 0x9E3779B97F4A7C15ull + frame * 0xBF58476D1CE4E5B9ull + pix * 0x94D049BB133111EBull + 0x2545F4914F6CDD1Dull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}

// bright field with a smooth radial roll-off inside R_in, dark vignette outside R_out, per-frame
// gain of +-1 %, noise in [-6, 6] (sum of four hash bytes)
PSEG_HD int synth_background(int x, int y, int H, int W, uint64_t seed, int frame)
{
    int64_t dx = x - W / 2, dy = y - H / 2;
    int64_t r2 = dx * dx + dy * dy;
    int64_t rin = (1230 * (int64_t)W) / 2560, rout = (1240 * (int64_t)W) / 2560;
    int64_t rin2 = rin * rin, rout2 = rout * rout;
    int64_t base;
    if (r2 <= rin2) base = 170 - (15 * r2) / rin2;
    else if (r2 >= rout2) base = 30;
    else base = 155 - (125 * (r2 - rin2)) / (rout2 - rin2);
    int64_t g = 990 + (int64_t)(synth_hash(seed, (uint64_t)frame, 0xFFFFFFFFull) % 21);
    base = base * g / 1000;
    uint64_t z = synth_hash(seed, (uint64_t)frame, (uint64_t)y * (uint64_t)W + (uint64_t)x);
    int t = (int)((z & 255) + ((z >> 8) & 255) + ((z >> 16) & 255) + ((z >> 24) & 255)) * 3;
    return (int)base + t / 255 - 6;
}

// objs row: cx, cy, a, b, cosQ12, sinQ12, contrast, kind (0 ellipse, 1 ring, 2 ring + core blob)
PSEG_HD int synth_object_hit(const int32_t *o, int x, int y)
{
    int64_t dx = x - o[0], dy = y - o[1];
    int64_t a = o[2], b = o[3];
    int64_t ext = (a > b ? a : b) + 1;
    if (dx > ext || dx < -ext || dy > ext || dy < -ext) return 0;
    int64_t u = dx * o[4] + dy * o[5], v = -dx * o[5] + dy * o[4];
    int64_t e = u * u * b * b + v * v * a * a;
    int64_t lim = a * a * b * b * (int64_t)(1 << 24);
    if (e > lim) return 0;
    if (o[7] == 0) return o[6];
    if (e > lim / 4) return o[6];                       // ring body (outside the half-scale ellipse)
    if (o[7] == 2 && e <= lim / 25) return o[6];        // core blob nested in the hole
    // semi-axes are capped at 400 by the generator so that e stays below 2^63
    return 0;
}
