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

#if defined(__HIPCC__) || defined(__CUDACC__)
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

#endif
