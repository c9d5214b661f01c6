// Development microbenchmark: what a grid of 256-thread blocks streams on this GPU with 16-byte accesses per lane, by
// mix of loads and stores (the ceiling k_generate / k_shade / k_accumulate are measured against in DESIGN.md section 5).
// build: hipcc --offload-arch=gfx950 -O3 -o stream_bw stream_bw.hip ; run: ./stream_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
template<int NR, int NW, bool NT, bool TILED>
__global__ __launch_bounds__(256) void k(const u4* __restrict__ src, u4* __restrict__ dst, size_t n)
{
    // NR source columns and NW destination columns of n 16-byte records each
    for(size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull)
    {
        u4 acc{1u, 2u, 3u, 4u};
#pragma unroll
        for(int r = 0; r < NR; ++r)
        {
            const size_t a = TILED ? ((i >> 6) * NR + r) * 64 + (i & 63) : r * n + i;
            u4 v = NT ? __builtin_nontemporal_load(src + a) : src[a];
            acc += v;
        }
#pragma unroll
        for(int w = 0; w < NW; ++w)
        {
            u4 v = acc + (unsigned)w;
            const size_t a = TILED ? ((i >> 6) * NW + w) * 64 + (i & 63) : w * n + i;
            if(NT) __builtin_nontemporal_store(v, dst + a); else dst[a] = v;
        }
        if(NW == 0 && acc.x == 0x12345u) dst[i] = acc;  // keep the loads
    }
}
template<int NR, int NW, bool NT, bool TILED = false> void run(const char* name, const u4* s, u4* d, size_t n, int grid)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<NR, NW, NT, TILED>), dim3(grid), dim3(256), 0, 0, s, d, n);
    hipEventRecord(a);
    for(int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<NR, NW, NT, TILED>), dim3(grid), dim3(256), 0, 0, s, d, n);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    double bytes = 16.0 * n * (NR + NW);
    printf("%-34s grid %5d: %7.3f ms  %6.2f TB/s (read %d x 16 B, write %d x 16 B per lane)\n", name, grid, ms, bytes / ms / 1e9, NR, NW);
}
// k_shade's queues as column groups: each group is its own buffer, tiled by 64 records (TILED) or plain columns
template<bool TILED>
__global__ __launch_bounds__(256) void kg(const u4* __restrict__ src, u4* __restrict__ dst, size_t n)
{
    constexpr int RG[2] = {2, 3}, WG[3] = {2, 3, 3};
    for(size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull)
    {
        u4 acc{1u, 2u, 3u, 4u};
        size_t base = 0;
#pragma unroll
        for(int g = 0; g < 2; ++g)
        {
#pragma unroll
            for(int c = 0; c < RG[g]; ++c)
            {
                const size_t a = base + (TILED ? ((i >> 6) * RG[g] + c) * 64 + (i & 63) : c * n + i);
                acc += __builtin_nontemporal_load(src + a);
            }
            base += RG[g] * n;
        }
        base = 0;
#pragma unroll
        for(int g = 0; g < 3; ++g)
        {
#pragma unroll
            for(int c = 0; c < WG[g]; ++c)
            {
                const size_t a = base + (TILED ? ((i >> 6) * WG[g] + c) * 64 + (i & 63) : c * n + i);
                __builtin_nontemporal_store(acc + (unsigned)c, dst + a);
            }
            base += WG[g] * n;
        }
    }
}
template<bool TILED> void rung(const char* name, const u4* s, u4* d, size_t n, int grid)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((kg<TILED>), dim3(grid), dim3(256), 0, 0, s, d, n);
    hipEventRecord(a);
    for(int r = 0; r < 5; ++r) hipLaunchKernelGGL((kg<TILED>), dim3(grid), dim3(256), 0, 0, s, d, n);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    printf("%-34s grid %5d: %7.3f ms  %6.2f TB/s\n", name, grid, ms, 16.0 * n * 13 / ms / 1e9);
}
// plain columns, column c shifted by c * skew records (does the relative alignment of the columns matter?)
__global__ __launch_bounds__(256) void ks(const u4* __restrict__ src, u4* __restrict__ dst, size_t n, size_t skew)
{
    for(size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull)
    {
        u4 acc{1u, 2u, 3u, 4u};
#pragma unroll
        for(int c = 0; c < 5; ++c) acc += __builtin_nontemporal_load(src + c * (n + skew) + i);
#pragma unroll
        for(int c = 0; c < 8; ++c) __builtin_nontemporal_store(acc + (unsigned)c, dst + c * (n + skew) + i);
    }
}
void runs(const u4* s, u4* d, size_t n, int grid, size_t skew)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(ks, dim3(grid), dim3(256), 0, 0, s, d, n, skew);
    hipEventRecord(a);
    for(int r = 0; r < 5; ++r) hipLaunchKernelGGL(ks, dim3(grid), dim3(256), 0, 0, s, d, n, skew);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    printf("k_shade mix, columns skewed by %7zu B      grid %5d: %7.3f ms  %6.2f TB/s\n", skew * 16, grid, ms, 16.0 * n * 13 / ms / 1e9);
}
// k_shade's mix with its queue append: the 8 output columns are written at slots a block obtains with one atomic per
// output queue and trip (two queues: columns 0-4 and 5-7), as block_append2 does; ATOMIC = false writes at the
// record's own index (k_generate's pattern). Every record appends, so the slots stay below n.
template<bool ATOMIC>
__global__ __launch_bounds__(256) void ka(const u4* __restrict__ src, u4* __restrict__ dst, size_t n, unsigned* ctr)
{
    __shared__ unsigned base_lds[2];
    for(size_t i0 = blockIdx.x * 256ull; i0 < n; i0 += gridDim.x * 256ull)
    {
        const size_t i = i0 + threadIdx.x;
        u4 acc{1u, 2u, 3u, 4u};
        if(i < n)
        {
#pragma unroll
            for(int c = 0; c < 5; ++c) acc += __builtin_nontemporal_load(src + c * n + i);
        }
        size_t slot_a = i, slot_b = i;
        if(ATOMIC)
        {
            const unsigned count = static_cast<unsigned>(n - i0 < 256 ? n - i0 : 256);
            __syncthreads();
            if(threadIdx.x < 2) base_lds[threadIdx.x] = atomicAdd(&ctr[threadIdx.x * 32], count);
            __syncthreads();
            slot_a = base_lds[0] + threadIdx.x;
            slot_b = base_lds[1] + threadIdx.x;
            __syncthreads();
        }
        if(i < n)
        {
#pragma unroll
            for(int c = 0; c < 5; ++c) __builtin_nontemporal_store(acc + (unsigned)c, dst + c * n + slot_a);
#pragma unroll
            for(int c = 5; c < 8; ++c) __builtin_nontemporal_store(acc + (unsigned)c, dst + c * n + slot_b);
        }
    }
}
template<bool ATOMIC> void runa(const char* name, const u4* s, u4* d, size_t n, int grid, unsigned* ctr)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float total = 0;
    for(int r = 0; r < 6; ++r)
    {
        hipMemsetAsync(ctr, 0, 256, 0);  // the slots restart at 0 for every launch: all of them stay below n
        hipEventRecord(a);
        hipLaunchKernelGGL((ka<ATOMIC>), dim3(grid), dim3(256), 0, 0, s, d, n, ctr);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if(r > 0) total += ms;
    }
    const float ms = total / 5;
    printf("%-34s grid %5d: %7.3f ms  %6.2f TB/s  (%.0f atomics/us per counter)\n", name, grid, ms, 16.0 * n * 13 / ms / 1e9, ATOMIC ? (n / 256.0) / (ms * 1e3) : 0.0);
}
int main()
{
    const size_t n = (1ull << 27) - (1ull << 20);  // leaves room for the skewed variant inside the same allocations  // records per column: 2 GiB per column
    u4 *s, *d;
    if(hipMalloc(&s, 16 * (n + (1ull << 20)) * 6) != hipSuccess || hipMalloc(&d, 16 * (n + (1ull << 20)) * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    unsigned* ctr; hipMalloc(&ctr, 256);
    hipMemset(s, 1, 16 * n * 6); hipMemset(d, 0, 16 * n * 8);
    for(int grid : {2048, 8192, 32768})
    {
        run<5, 0, true>("read only, nt", s, d, n, grid);
        run<0, 5, true>("write only, nt (k_generate-like)", s, d, n, grid);
        run<0, 5, false>("write only, plain", s, d, n, grid);
        run<4, 4, true>("copy, nt", s, d, n, grid);
        run<4, 4, false>("copy, plain", s, d, n, grid);
        run<5, 8, true>("k_shade mix (80 B in, 128 B out)", s, d, n, grid);
        run<6, 1, true>("k_shadow-like (96 B in, 16 B out)", s, d, n, grid);
        run<0, 5, true, true>("write only, nt, 64-record tiles", s, d, n, grid);
        run<5, 8, true, true>("k_shade mix, 64-record tiles", s, d, n, grid);
        run<5, 8, false, true>("k_shade mix, tiles, plain", s, d, n, grid);
        run<5, 0, true, true>("read only, nt, tiles", s, d, n, grid);
        rung<false>("k_shade groups {2,3}->{2,3,3} cols", s, d, n, grid);
        rung<true>("k_shade groups {2,3}->{2,3,3} tiled", s, d, n, grid);
        runa<false>("k_shade mix, own index", s, d, n, grid, ctr);
        runa<true>("k_shade mix, block-atomic append", s, d, n, grid, ctr);
        for(size_t skew : {0ull, 8ull, 16ull, 64ull, 256ull, 1024ull, 4096ull, 16384ull + 16ull, 65536ull + 4096ull + 64ull}) runs(s, d, n, grid, skew);
    }
    return 0;
}
