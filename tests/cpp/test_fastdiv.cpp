// CPU check of fcg::fast_div (csrc/fc_types.h) against the hardware divide.
#include <cstdio>
#include <cstdint>
#include <random>
#include "../../ray-tracing_amd/csrc/fc_types.h"
int main()
{
    std::mt19937_64 rng(7);
    const uint32_t ds[] = {1, 2, 3, 5, 7, 64, 255, 256, 257, 641, 1000, 1920, 1080, 2073600, 33554432 - 1, 33554432, 33177600, 0x7FFFFFFFu, 0x40000001u};
    long bad = 0, n_checked = 0;
    auto check = [&](uint32_t d, uint32_t n) {
        const fcg::FastDiv f = fcg::make_fastdiv(d);
        ++n_checked;
        if(fcg::fast_div(n, f) != n / d) { if(bad < 5) std::printf("mismatch n=%u d=%u\n", n, d); ++bad; }
    };
    for(uint32_t d : ds)
    {
        for(uint32_t k = 0; k < 70000; ++k) { check(d, k); check(d, 0xFFFFFFFFu - k); check(d, d * (k % 3000u) + (k & 1u ? d - 1 : 0)); }
        for(int k = 0; k < 200000; ++k) check(d, static_cast<uint32_t>(rng()));
    }
    for(int k = 0; k < 2000000; ++k) { uint32_t d = static_cast<uint32_t>(rng() % 0x7FFFFFFFull) + 1u; check(d, static_cast<uint32_t>(rng())); check(d, d * static_cast<uint32_t>(rng() % (0xFFFFFFFFull / d + 1)) ); }
    std::printf("fast_div: %ld checks, %ld mismatches\n", n_checked, bad);
    return bad != 0;
}
