// Development microbenchmark: dependent random 64-byte "node" fetches, one chain per lane, as BVH
// traversal issues them. Reports lines/us/CU for several table sizes, waves per CU and fetch shapes.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// MODE 0: four dwordx4 loads per lane; MODE 1: one dwordx4 per lane (16 of the 64 bytes); MODE 2: quad-cooperative LDS-DMA
template<int MODE>
__global__ __launch_bounds__(256) void k(const uint4* table, uint32_t mask, int steps, uint32_t* out)
{
    __shared__ __attribute__((aligned(16))) unsigned char buf[4][4 * 1040];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t idx = mix(blockIdx.x * 256 + threadIdx.x) & mask, acc = 0;
    unsigned char* wb = buf[wave];
    for(int s = 0; s < steps; ++s)
    {
        uint32_t v;
        if(MODE == 0)
        {
            const uint4* p = table + (size_t)idx * 4;
            uint4 a = p[0], b = p[1], c = p[2], d = p[3];
            v = a.x ^ b.y ^ c.z ^ d.w;
        }
        else if(MODE == 1)
        {
            uint4 a = table[(size_t)idx * 4];
            v = a.x ^ a.w;
        }
        else
        {
#define FETCH(K)                                                                                                         \
    {                                                                                                                    \
        uint32_t ck = __builtin_amdgcn_mov_dpp(idx, K * 0x55, 0xF, 0xF, true);                                           \
        const unsigned char* g = reinterpret_cast<const unsigned char*>(table) + (size_t)ck * 64 + (lane & 3) * 16;      \
        __builtin_amdgcn_global_load_lds((glb_void*)g, (lds_void*)(wb + K * 1040), 16, 0, 0);                            \
    }
            FETCH(0) FETCH(1) FETCH(2) FETCH(3)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint4* m = reinterpret_cast<const uint4*>(wb + (lane & 3) * 1040 + (lane >> 2) * 64);
            uint4 a = m[0], b = m[1], c = m[2], d = m[3];
            v = a.x ^ b.y ^ c.z ^ d.w;
        }
        acc += v;
        idx = mix(idx + v + s) & mask;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main()
{
    hipDeviceProp_t prop;
    if(hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
    const int cus = prop.multiProcessorCount;
    const size_t max_lines = 1u << 21;  // 128 MB
    std::vector<uint4> h(max_lines * 4);
    for(size_t i = 0; i < h.size(); ++i) h[i] = uint4{(uint32_t)(i * 2654435761u), (uint32_t)(i * 40503u), (uint32_t)i, (uint32_t)(i * 7u)};
    uint4* d;
    uint32_t* out;
    if(hipMalloc(&d, h.size() * sizeof(uint4)) != hipSuccess || hipMalloc(&out, 4u * 256 * cus * 8) != hipSuccess) return 1;
    (void)hipMemcpy(d, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int steps = 2000;
    printf("CUs %d; lines/us/CU (64-byte random dependent fetches)\n", cus);
    for(uint32_t lines : {1u << 12, 1u << 13, 1u << 16, 1u << 21})  // 256 KB, 512 KB, 4 MB, 128 MB
        for(int bpc : {1, 2, 3, 4, 6, 8})
        {
            float ms[3];
            for(int mode = 0; mode < 3; ++mode)
            {
                dim3 g(cus * bpc), b(256);
                for(int rep = 0; rep < 2; ++rep)
                {
                    (void)hipEventRecord(e0, 0);
                    if(mode == 0) hipLaunchKernelGGL(k<0>, g, b, 0, 0, d, lines - 1, steps, out);
                    if(mode == 1) hipLaunchKernelGGL(k<1>, g, b, 0, 0, d, lines - 1, steps, out);
                    if(mode == 2) hipLaunchKernelGGL(k<2>, g, b, 0, 0, d, lines - 1, steps, out);
                    (void)hipEventRecord(e1, 0);
                    (void)hipEventSynchronize(e1);
                    (void)hipEventElapsedTime(&ms[mode], e0, e1);
                }
            }
            auto rate = [&](float m) { return (double)bpc * 256 * steps / (m * 1e3); };
            printf("table %6u KB  blocks/CU %d (waves %2d): 4xdwordx4 %7.1f  1xdwordx4 %7.1f  quad-glds %7.1f   ns/step %6.0f %6.0f %6.0f\n", lines / 16, bpc, bpc * 4,
                   rate(ms[0]), rate(ms[1]), rate(ms[2]), ms[0] * 1e6 / steps, ms[1] * 1e6 / steps, ms[2] * 1e6 / steps);
        }
    return 0;
}
