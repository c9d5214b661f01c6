/* test shim: include/fcgpu_sampler.h behind a flat C entry for ctypes */
#include "../../include/fcgpu_sampler.h"
void shim_draws(uint64_t seed, uint32_t stratified, uint32_t px, uint32_t py, uint32_t dim, uint32_t first, uint32_t n, uint32_t* x24, uint32_t* y24)
{
    fcgpu_sampler_config c = stratified ? fcgpu_sampler_stratified(seed, stratified) : fcgpu_sampler_random(seed);
    for(uint32_t k = 0; k < n; ++k)
    {
        fcgpu_sample_stream s = fcgpu_stream_begin(&c, px, py, first + k);
        uint64_t b = fcgpu_stream_bits(&c, s, dim);
        x24[k] = fcgpu_sample_x24(b);
        y24[k] = fcgpu_sample_y24(b);
    }
}
void shim_config(uint32_t stratified, uint32_t* n, uint32_t* count, uint32_t* cell_mul)
{
    fcgpu_sampler_config c = fcgpu_sampler_stratified(0, stratified);
    *n = c.n; *count = c.count; *cell_mul = c.cell_mul;
}
uint64_t shim_random_bits(uint64_t seed, uint32_t px, uint32_t py, uint32_t s, uint32_t dim) { return fcgpu_sample_bits(fcgpu_sample_key(seed, px, py, s), dim); }
