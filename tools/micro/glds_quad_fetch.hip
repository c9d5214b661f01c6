#include <hip/hip_runtime.h>
#include <cstdint>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
template<int K> __device__ __forceinline__ void fetch(const uint4* nodes, uint32_t cur, unsigned char* wb, int lane)
{
    uint32_t ck = __builtin_amdgcn_mov_dpp(cur, K * 0x55, 0xF, 0xF, true);
    if((int)ck >= 0)
    {
        const unsigned char* g = reinterpret_cast<const unsigned char*>(nodes) + (size_t)ck * 64 + (lane & 3) * 16;
        __builtin_amdgcn_global_load_lds((glb_void*)g, (lds_void*)(wb + K * 1040), 16, 0, 0);
    }
}
__global__ void k(const uint4* nodes, const uint32_t* idx, uint4* out)
{
    __shared__ __attribute__((aligned(16))) unsigned char buf[4][4 * 1040];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t cur = idx[blockIdx.x * 256 + threadIdx.x];
    unsigned char* wb = buf[wave];
    fetch<0>(nodes, cur, wb, lane); fetch<1>(nodes, cur, wb, lane); fetch<2>(nodes, cur, wb, lane); fetch<3>(nodes, cur, wb, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint4* mine = reinterpret_cast<const uint4*>(wb + (lane & 3) * 1040 + (lane >> 2) * 64);
    uint4 a = mine[0], b = mine[1], c = mine[2], d = mine[3];
    out[blockIdx.x * 256 + threadIdx.x] = uint4{a.x + b.x, a.y + c.y, a.z + d.z, a.w ^ b.w ^ c.w ^ d.w};
}

// Host check (development microtest): every lane holding a node index must see its own 64-byte
// node after the quad-cooperative LDS-DMA fetch; lanes holding a negative index fetch nothing.
#include <cstdio>
#include <vector>
#include <random>
int main()
{
    const int nodes_n = 1 << 16, blocks = 512, n = blocks * 256;
    std::vector<uint4> nodes(nodes_n * 4);
    for(int i = 0; i < nodes_n * 4; ++i) nodes[i] = uint4{(unsigned)i * 7u + 1u, (unsigned)i * 13u + 2u, (unsigned)i * 17u + 3u, (unsigned)i * 19u + 4u};
    std::vector<uint32_t> idx(n);
    std::mt19937 rng(1);
    for(int i = 0; i < n; ++i) idx[i] = (rng() % 3 == 0) ? 0x80000000u | (rng() & 0xFFFF) : rng() % nodes_n;
    uint4 *dn, *dout;
    uint32_t* di;
    hipMalloc(&dn, nodes.size() * sizeof(uint4));
    hipMalloc(&di, n * 4);
    hipMalloc(&dout, n * sizeof(uint4));
    hipMemcpy(dn, nodes.data(), nodes.size() * sizeof(uint4), hipMemcpyHostToDevice);
    hipMemcpy(di, idx.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, dn, di, dout);
    std::vector<uint4> out(n);
    if(hipMemcpy(out.data(), dout, n * sizeof(uint4), hipMemcpyDeviceToHost) != hipSuccess) { printf("FAIL hip\n"); return 1; }
    long bad = 0, checked = 0;
    for(int i = 0; i < n; ++i)
    {
        if((int)idx[i] < 0) continue;
        const uint4 *p = &nodes[(size_t)idx[i] * 4];
        uint4 e{p[0].x + p[1].x, p[0].y + p[2].y, p[0].z + p[3].z, p[0].w ^ p[1].w ^ p[2].w ^ p[3].w};
        ++checked;
        if(e.x != out[i].x || e.y != out[i].y || e.z != out[i].z || e.w != out[i].w) ++bad;
    }
    printf("glds quad fetch: checked %ld bad %ld -> %s\n", checked, bad, bad ? "FAIL" : "OK");
    return bad != 0;
}
