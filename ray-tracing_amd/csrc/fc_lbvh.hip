// fc_lbvh.hip -- binary BVH hierarchy built on the device: the alternative to the host's
// reference-order SAH build (bvh_acceleration_structure.hpp:228-365) for meshes large enough that the host
// recursion dominates scene upload. Linear BVH: 63-bit Morton codes of the primitive centroids, device radix
// sort, the Karras 2012 radix-tree construction (one thread per internal node), bottom-up bounds by the
// second-arriver rule. The result goes back to the host as (left, right, box) per internal node; the host
// turns it into its depth-first BinNode array and runs the same 4-wide collapse + quantisation on it.
//
// Any valid BVH returns the same closest hit (exact t ties aside), so parity with the reference holds; the
// tree is of lower quality than the SAH one (more node steps per ray) -- DESIGN.md has the numbers.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdint>
#include <string>
#include <vector>

#include "fc_host.h"

namespace fcg
{
    namespace
    {
        constexpr int LB = 256;

        __device__ __forceinline__ uint64_t spread21(uint32_t v)  // 21 bits -> every third bit of 63
        {
            uint64_t x = v & 0x1FFFFFu;
            x = (x | (x << 32)) & 0x1F00000000FFFFull;
            x = (x | (x << 16)) & 0x1F0000FF0000FFull;
            x = (x | (x << 8)) & 0x100F00F00F00F00Full;
            x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
            x = (x | (x << 2)) & 0x1249249249249249ull;
            return x;
        }

        // boxes: [n][6] = lo.xyz, hi.xyz; scene: centroid bounds lo.xyz + 1 / extent per axis (0 for a flat axis)
        __global__ __launch_bounds__(LB) void k_morton(const float* __restrict__ boxes, uint32_t n, float lx, float ly, float lz, float sx,
                                                       float sy, float sz, uint64_t* __restrict__ keys, uint32_t* __restrict__ ids)
        {
            const uint32_t i = blockIdx.x * LB + threadIdx.x;
            if(i >= n) return;
            const float* b = boxes + 6ull * i;
            const float cx = 0.5f * b[0] + 0.5f * b[3], cy = 0.5f * b[1] + 0.5f * b[4], cz = 0.5f * b[2] + 0.5f * b[5];
            auto cell = [](float c, float lo, float scale) {
                float f = (c - lo) * scale * 2097152.0f;
                f = f < 0.0f ? 0.0f : (f > 2097151.0f ? 2097151.0f : f);
                return static_cast<uint32_t>(f);
            };
            keys[i] = (spread21(cell(cx, lx, sx)) << 2) | (spread21(cell(cy, ly, sy)) << 1) | spread21(cell(cz, lz, sz));
            ids[i] = i;
        }

        // length of the common prefix of keys i and j, ties broken by the index (Karras 2012, section 4)
        __device__ __forceinline__ int prefix(const uint64_t* keys, int n, int i, int j)
        {
            if(j < 0 || j >= n) return -1;
            const uint64_t a = keys[i], b = keys[j];
            if(a != b) return __clzll(static_cast<long long>(a ^ b));
            return 64 + __clz(i ^ j);
        }

        // children: >= 0 internal node, < 0 leaf ~k (k = position in sorted order)
        __global__ __launch_bounds__(LB) void k_hierarchy(const uint64_t* __restrict__ keys, int n, int32_t* __restrict__ left,
                                                          int32_t* __restrict__ right, int32_t* __restrict__ parent_of_internal,
                                                          int32_t* __restrict__ parent_of_leaf)
        {
            const int i = blockIdx.x * LB + threadIdx.x;
            if(i >= n - 1) return;
            const int d = prefix(keys, n, i, i + 1) - prefix(keys, n, i, i - 1) >= 0 ? 1 : -1;
            const int d_min = prefix(keys, n, i, i - d);
            int l_max = 2;
            while(prefix(keys, n, i, i + l_max * d) > d_min) l_max *= 2;
            int l = 0;
            for(int t = l_max / 2; t >= 1; t /= 2)
                if(prefix(keys, n, i, i + (l + t) * d) > d_min) l += t;
            const int j = i + l * d;
            const int d_node = prefix(keys, n, i, j);
            int s = 0;
            for(int t = (l + 1) / 2;; t = (t + 1) / 2)
            {
                if(prefix(keys, n, i, i + (s + t) * d) > d_node) s += t;
                if(t == 1) break;
            }
            const int gamma = i + s * d + (d < 0 ? -1 : 0);
            const int lo = i < j ? i : j, hi = i < j ? j : i;
            if(lo == gamma)
            {
                left[i] = ~gamma;
                parent_of_leaf[gamma] = i;
            }
            else
            {
                left[i] = gamma;
                parent_of_internal[gamma] = i;
            }
            if(hi == gamma + 1)
            {
                right[i] = ~(gamma + 1);
                parent_of_leaf[gamma + 1] = i;
            }
            else
            {
                right[i] = gamma + 1;
                parent_of_internal[gamma + 1] = i;
            }
            if(i == 0) parent_of_internal[0] = -1;
        }

        // One thread per leaf walks up; the second thread to reach a node unites its children's boxes. Node boxes
        // are written by one thread and read by another, possibly on another XCD (own L2): they go through
        // agent-scope atomic stores / loads, ordered by the fences around the visit counter.
        __device__ __forceinline__ float load_shared(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        __device__ __forceinline__ void store_shared(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

        __global__ __launch_bounds__(LB) void k_bounds(const float* __restrict__ boxes, const uint32_t* __restrict__ ids, int n,
                                                       const int32_t* __restrict__ left, const int32_t* __restrict__ right,
                                                       const int32_t* __restrict__ parent_of_internal, const int32_t* __restrict__ parent_of_leaf,
                                                       uint32_t* visits, float* node_boxes)
        {
            const int k = blockIdx.x * LB + threadIdx.x;
            if(k >= n) return;
            int node = parent_of_leaf[k];
            while(node >= 0)
            {
                __threadfence();                                                  // my child's box before my visit
                if(atomicAdd(&visits[node], 1u) == 0u) return;                    // the sibling subtree is not finished yet
                __threadfence();                                                  // the sibling's box after its visit
                float box[6];
                for(int side = 0; side < 2; ++side)
                {
                    const int32_t c = side ? right[node] : left[node];
                    for(int a = 0; a < 3; ++a)
                    {
                        float lo, hi;
                        if(c < 0)
                        {
                            const float* src = boxes + 6ull * ids[~c];
                            lo = src[a];
                            hi = src[3 + a];
                        }
                        else
                        {
                            lo = load_shared(node_boxes + 6ull * c + a);
                            hi = load_shared(node_boxes + 6ull * c + 3 + a);
                        }
                        box[a] = side ? fminf(box[a], lo) : lo;
                        box[3 + a] = side ? fmaxf(box[3 + a], hi) : hi;
                    }
                }
                for(int a = 0; a < 6; ++a) store_shared(node_boxes + 6ull * node + a, box[a]);
                node = parent_of_internal[node];
            }
        }

        struct DeviceBuffers
        {
            std::vector<void*> ptrs;
            ~DeviceBuffers()
            {
                for(void* p : ptrs) (void)hipFree(p);
            }
            template<typename T> hipError_t alloc(T** p, size_t count)
            {
                hipError_t e = hipMalloc(reinterpret_cast<void**>(p), sizeof(T) * (count ? count : 1));
                if(e == hipSuccess) ptrs.push_back(*p);
                return e;
            }
        };
    }

    int lbvh_build(hipStream_t stream, const std::vector<float>& boxes, LbvhTree& out, std::string& err)
    {
        const size_t n = boxes.size() / 6;
        if(n < 2 || n > (1u << 30))
        {
            err = "lbvh_build needs between 2 and 2^30 primitives";
            return FCGPU_E_INVALID;
        }
        // centroid bounds on the host (one pass over data that is already here)
        float lo[3]{3.4e38f, 3.4e38f, 3.4e38f}, hi[3]{-3.4e38f, -3.4e38f, -3.4e38f};
        for(size_t i = 0; i < n; ++i)
            for(int a = 0; a < 3; ++a)
            {
                const float c = 0.5f * boxes[6 * i + a] + 0.5f * boxes[6 * i + 3 + a];
                lo[a] = c < lo[a] ? c : lo[a];
                hi[a] = c > hi[a] ? c : hi[a];
            }
        float scale[3];
        for(int a = 0; a < 3; ++a) scale[a] = hi[a] > lo[a] ? 1.0f / (hi[a] - lo[a]) : 0.0f;

#define LBVH_CHECK(call)                                                  \
    do                                                                    \
    {                                                                     \
        hipError_t e_ = (call);                                           \
        if(e_ != hipSuccess)                                              \
        {                                                                 \
            err = std::string("lbvh_build: ") + hipGetErrorString(e_);    \
            return e_ == hipErrorOutOfMemory ? FCGPU_E_NOMEM : FCGPU_E_HIP; \
        }                                                                 \
    } while(0)

        DeviceBuffers dev;
        float *d_boxes, *d_node_boxes;
        uint64_t *d_keys, *d_keys_sorted;
        uint32_t *d_ids, *d_ids_sorted, *d_visits;
        int32_t *d_left, *d_right, *d_parent_int, *d_parent_leaf;
        LBVH_CHECK(dev.alloc(&d_boxes, 6 * n));
        LBVH_CHECK(dev.alloc(&d_node_boxes, 6 * (n - 1)));
        LBVH_CHECK(dev.alloc(&d_keys, n));
        LBVH_CHECK(dev.alloc(&d_keys_sorted, n));
        LBVH_CHECK(dev.alloc(&d_ids, n));
        LBVH_CHECK(dev.alloc(&d_ids_sorted, n));
        LBVH_CHECK(dev.alloc(&d_visits, n - 1));
        LBVH_CHECK(dev.alloc(&d_left, n - 1));
        LBVH_CHECK(dev.alloc(&d_right, n - 1));
        LBVH_CHECK(dev.alloc(&d_parent_int, n - 1));
        LBVH_CHECK(dev.alloc(&d_parent_leaf, n));
        LBVH_CHECK(hipMemcpyAsync(d_boxes, boxes.data(), sizeof(float) * 6 * n, hipMemcpyHostToDevice, stream));
        LBVH_CHECK(hipMemsetAsync(d_visits, 0, sizeof(uint32_t) * (n - 1), stream));

        const uint32_t blocks = static_cast<uint32_t>((n + LB - 1) / LB);
        hipLaunchKernelGGL(k_morton, dim3(blocks), dim3(LB), 0, stream, d_boxes, static_cast<uint32_t>(n), lo[0], lo[1], lo[2], scale[0], scale[1],
                           scale[2], d_keys, d_ids);
        size_t temp_bytes = 0;
        LBVH_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, d_keys, d_keys_sorted, d_ids, d_ids_sorted, static_cast<int>(n), 0, 63, stream));
        void* d_temp;
        LBVH_CHECK(hipMalloc(&d_temp, temp_bytes ? temp_bytes : 1));
        dev.ptrs.push_back(d_temp);
        LBVH_CHECK(hipcub::DeviceRadixSort::SortPairs(d_temp, temp_bytes, d_keys, d_keys_sorted, d_ids, d_ids_sorted, static_cast<int>(n), 0, 63, stream));
        hipLaunchKernelGGL(k_hierarchy, dim3(blocks), dim3(LB), 0, stream, d_keys_sorted, static_cast<int>(n), d_left, d_right, d_parent_int, d_parent_leaf);
        hipLaunchKernelGGL(k_bounds, dim3(blocks), dim3(LB), 0, stream, d_boxes, d_ids_sorted, static_cast<int>(n), d_left, d_right, d_parent_int,
                           d_parent_leaf, d_visits, d_node_boxes);
        LBVH_CHECK(hipGetLastError());

        out.order.resize(n);
        out.left.resize(n - 1);
        out.right.resize(n - 1);
        out.node_boxes.resize(6 * (n - 1));
        LBVH_CHECK(hipMemcpyAsync(out.order.data(), d_ids_sorted, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, stream));
        LBVH_CHECK(hipMemcpyAsync(out.left.data(), d_left, sizeof(int32_t) * (n - 1), hipMemcpyDeviceToHost, stream));
        LBVH_CHECK(hipMemcpyAsync(out.right.data(), d_right, sizeof(int32_t) * (n - 1), hipMemcpyDeviceToHost, stream));
        LBVH_CHECK(hipMemcpyAsync(out.node_boxes.data(), d_node_boxes, sizeof(float) * 6 * (n - 1), hipMemcpyDeviceToHost, stream));
        LBVH_CHECK(hipStreamSynchronize(stream));
#undef LBVH_CHECK
        return FCGPU_OK;
    }
}
