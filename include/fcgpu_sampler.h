/*
 * fcgpu_sampler.h -- the counter-based sample stream shared by the GPU path, the CPU oracle and
 * the reference harness (so that all three consume IDENTICAL samples).
 *
 * Replaces the reference's engine-based samplers behind the `fc::sampler` protocol
 * (src/core/sampler.hpp:15-34; src/samplers/random_sampler.hpp:26-54): a 2-D sample is a pure
 * function of (seed, pixel x, pixel y, sample index, dimension index). `get()` returns the draw
 * of the current dimension and advances it by one; `advance_dimension(n)` adds n.
 *
 * Every draw is a 24-bit fraction k * 2^-24, k in [0, 2^24), so the value is exactly
 * representable in fp32 and fp64 alike: the fp64 reference and the fp32 device path see the
 * same real numbers.
 *
 * Dimension map of one Forward-MIS sample (forward_mis_integrator.hpp; the reference draws
 * inside argument lists, and the oracle build fixes g++'s right-to-left evaluation, so the
 * FIRST draw feeds the LAST parameter):
 *   dim 0            camera sample_direction (pixel jitter)        :17 (2nd argument)
 *   dim 1            camera sample_point (lens)                    :17 (1st argument)
 *   per bounce b = 0.., base = 2 + 6 b:
 *   base + 0  .x     lobe pick  bsdf::sample_bxdf                  :48
 *   base + 1  .x     light pick light_distribution::sample         :54
 *   base + 2         light sample_point                            :77 (3rd argument)
 *   base + 3  .x     light sample_primitive                        :77 (2nd argument)
 *   base + 4         bsdf u2                                       :106 (2nd get())
 *   base + 5         bsdf u1                                       :106 (1st get())
 *   delta lobes skip base+1..base+3 (advance_dimension(3), :155) and use base+4 / base+5 alike.
 */
#ifndef FCGPU_SAMPLER_H
#define FCGPU_SAMPLER_H

#include <stdint.h>

#if defined(__HIPCC__)
#define FCGPU_SAMPLER_FN static inline __host__ __device__
#else
#define FCGPU_SAMPLER_FN static inline
#endif

/* splitmix64 finalizer (Steele, Lea, Flood 2014) */
FCGPU_SAMPLER_FN uint64_t fcgpu_mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* key of one (pixel, sample): px, py < 2^20, sample < 2^24 */
FCGPU_SAMPLER_FN uint64_t fcgpu_sample_key(uint64_t seed, uint32_t px, uint32_t py, uint32_t sample)
{
    uint64_t id = (uint64_t)px | ((uint64_t)py << 20) | ((uint64_t)sample << 40);
    return fcgpu_mix64(fcgpu_mix64(id) ^ (seed * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull));
}

/* raw 64 bits of dimension `dim`; x = high word, y = low word */
FCGPU_SAMPLER_FN uint64_t fcgpu_sample_bits(uint64_t key, uint32_t dim)
{
    return fcgpu_mix64(key + ((uint64_t)dim + 1u) * 0x9E3779B97F4A7C15ull);
}

FCGPU_SAMPLER_FN uint32_t fcgpu_sample_x24(uint64_t bits) { return (uint32_t)(bits >> 40); }
FCGPU_SAMPLER_FN uint32_t fcgpu_sample_y24(uint64_t bits) { return ((uint32_t)bits) >> 8; }

/* 2^-24 */
#define FCGPU_SAMPLER_SCALE 5.9604644775390625e-08

/* ---------------------------------------------------------------------------------------------
 * Stream interface: what the integrators draw from. Two modes behind it.
 *
 * FCGPU_SAMPLER_RANDOM      independent draws: the functions above (the analogue of
 *                           random_sampler, src/samplers/random_sampler.hpp:26-54).
 * FCGPU_SAMPLER_STRATIFIED  the analogue of stratified_sampler (src/samplers/stratified_sampler.hpp:
 *                           94-150): per (pixel, dimension) the n*n samples of a pixel are jittered
 *                           points of an n x n grid (:128-141: x = (j + u) / n, y = (i + u') / n, cell
 *                           index = i * n + j), handed to the sample indices in a pseudo-random order
 *                           (:144-149 shuffle). The reference draws grid, jitter and shuffle from one
 *                           pcg32 stream per pixel; here all three are pure functions of
 *                           (seed, pixel, dimension, sample): the order is a hashed permutation of
 *                           [0, n*n) (cycle-walking, Kensler 2013 "Correlated Multi-Jittered Sampling",
 *                           listing 3, without its final rotation), the jitter a hash. Statistically
 *                           the same estimator, not the same numbers. n = floor(sqrt(sample_count))
 *                           as the reference (:15-16); sample indices >= n*n start a new,
 *                           independently permuted and jittered grid ("epoch").
 *                           Draws stay 24-bit fractions: cell j spans [b(j), b(j+1)) with
 *                           b(j) = (j * floor(2^32 / n)) >> 8, and x24 = b(j) + ((u24 * width) >> 24).
 * ------------------------------------------------------------------------------------------- */
#define FCGPU_SAMPLER_RANDOM 0
#define FCGPU_SAMPLER_STRATIFIED 1

typedef struct fcgpu_sampler_config {
    uint64_t seed;
    uint32_t mode;
    uint32_t n;          /* strata per axis (stratified) */
    uint32_t count;      /* n * n */
    uint32_t cell_mul;   /* floor(2^32 / n) (n = 1: 2^32 - 1) */
    uint32_t count_mul;  /* multiply-high division by count: see fcgpu_sampler_div */
    uint32_t count_sh1;
    uint32_t n_mul;      /* ... and by n */
    uint32_t n_sh1;
} fcgpu_sampler_config;

typedef struct fcgpu_sample_stream {
    uint64_t key;      /* RANDOM: key of (pixel, sample); STRATIFIED: key of (pixel, epoch) */
    uint32_t index;    /* STRATIFIED: sample index within its epoch */
    uint32_t reserved;
} fcgpu_sample_stream;

/* floor(v / d) for 32-bit v through the magic numbers of fcgpu_sampler_magic (Granlund & Montgomery) */
FCGPU_SAMPLER_FN uint32_t fcgpu_sampler_div(uint32_t v, uint32_t d, uint32_t mul, uint32_t sh1)
{
    uint32_t t = (uint32_t)(((uint64_t)v * mul) >> 32);
    uint32_t q = (t + ((v - t) >> 1)) >> sh1;
    return d <= 1u ? v : q;
}

static inline void fcgpu_sampler_magic(uint32_t d, uint32_t* mul, uint32_t* sh1)
{
    uint32_t s = 0;
    *mul = 0;
    *sh1 = 0;
    if(d <= 1u) return;
    while(((uint64_t)1 << s) < d) ++s;
    *mul = (uint32_t)(((((uint64_t)1 << (32 + s)) + d - 1) / d) - ((uint64_t)1 << 32));
    *sh1 = s - 1u;
}

static inline fcgpu_sampler_config fcgpu_sampler_random(uint64_t seed)
{
    fcgpu_sampler_config c = {seed, FCGPU_SAMPLER_RANDOM, 1u, 1u, 0xFFFFFFFFu, 0u, 0u, 0u, 0u};
    return c;
}

/* stratified_sampler{sample_count, seed} (stratified_sampler.hpp:12-19); sample_count in [1, 2^24] */
static inline fcgpu_sampler_config fcgpu_sampler_stratified(uint64_t seed, uint32_t sample_count)
{
    fcgpu_sampler_config c = fcgpu_sampler_random(seed);
    uint32_t n = 1;
    while((uint64_t)(n + 1u) * (n + 1u) <= sample_count) ++n;
    c.mode = FCGPU_SAMPLER_STRATIFIED;
    c.n = n;
    c.count = n * n;
    c.cell_mul = n <= 1u ? 0xFFFFFFFFu : (uint32_t)(((uint64_t)1 << 32) / n);
    fcgpu_sampler_magic(c.count, &c.count_mul, &c.count_sh1);
    fcgpu_sampler_magic(c.n, &c.n_mul, &c.n_sh1);
    return c;
}

/* hashed permutation of [0, l): position i -> element, keyed by p (Kensler 2013, cycle walking) */
FCGPU_SAMPLER_FN uint32_t fcgpu_permute(uint32_t i, uint32_t l, uint32_t p)
{
    uint32_t w = l - 1u;
    w |= w >> 1;
    w |= w >> 2;
    w |= w >> 4;
    w |= w >> 8;
    w |= w >> 16;
    do
    {
        i ^= p;
        i *= 0xe170893du;
        i ^= p >> 16;
        i ^= (i & w) >> 4;
        i ^= p >> 8;
        i *= 0x0929eb3fu;
        i ^= p >> 23;
        i ^= (i & w) >> 1;
        i *= 1u | p >> 27;
        i *= 0x6935fa69u;
        i ^= (i & w) >> 11;
        i *= 0x74dcb303u;
        i ^= (i & w) >> 2;
        i *= 0x9e501cc3u;
        i ^= (i & w) >> 2;
        i *= 0xc860a3dfu;
        i &= w;
        i ^= i >> 5;
    } while(i >= l);
    return i;
}

FCGPU_SAMPLER_FN fcgpu_sample_stream fcgpu_stream_begin(const fcgpu_sampler_config* c, uint32_t px, uint32_t py, uint32_t sample)
{
    fcgpu_sample_stream s;
    s.reserved = 0u;
    if(c->mode == FCGPU_SAMPLER_STRATIFIED)
    {
        const uint32_t epoch = fcgpu_sampler_div(sample, c->count, c->count_mul, c->count_sh1);
        s.index = sample - epoch * c->count;
        /* sample field 0xFFFFFF - epoch: no collision with the per-sample keys below 2^24 - 2^12 epochs */
        s.key = fcgpu_sample_key(c->seed ^ 0x5354524154494649ull, px, py, 0xFFFFFFu - epoch);
    }
    else
    {
        s.index = sample;
        s.key = fcgpu_sample_key(c->seed, px, py, sample);
    }
    return s;
}

/* the 2-D draw of dimension `dim`, packed like fcgpu_sample_bits (x24 = bits >> 40, y24 = low word >> 8) */
FCGPU_SAMPLER_FN uint64_t fcgpu_stream_bits(const fcgpu_sampler_config* c, fcgpu_sample_stream s, uint32_t dim)
{
    if(c->mode != FCGPU_SAMPLER_STRATIFIED) return fcgpu_sample_bits(s.key, dim);
    const uint64_t dkey = fcgpu_sample_bits(s.key, dim);                       /* per (pixel, epoch, dimension) */
    const uint32_t cell = fcgpu_permute(s.index, c->count, (uint32_t)(dkey >> 32));
    const uint32_t i = fcgpu_sampler_div(cell, c->n, c->n_mul, c->n_sh1);     /* row    -> y (:131-139) */
    const uint32_t j = cell - i * c->n;                                       /* column -> x           */
    const uint64_t jit = fcgpu_mix64(dkey ^ ((uint64_t)(s.index + 1u) * 0xD6E8FEB86659FD93ull));
    const uint32_t ux = (uint32_t)(jit >> 40), uy = ((uint32_t)jit) >> 8;
    const uint32_t bx0 = (uint32_t)(((uint64_t)j * c->cell_mul) >> 8), bx1 = (uint32_t)(((uint64_t)(j + 1u) * c->cell_mul) >> 8);
    const uint32_t by0 = (uint32_t)(((uint64_t)i * c->cell_mul) >> 8), by1 = (uint32_t)(((uint64_t)(i + 1u) * c->cell_mul) >> 8);
    const uint32_t x24 = bx0 + (uint32_t)(((uint64_t)ux * (bx1 - bx0)) >> 24);
    const uint32_t y24 = by0 + (uint32_t)(((uint64_t)uy * (by1 - by0)) >> 24);
    return ((uint64_t)x24 << 40) | ((uint64_t)y24 << 8);
}

#endif
