// tools/widestats/wide_stats.cpp -- DEVELOPMENT TOOL (not part of the library, not a render path): walks the flattened
// 4-wide or 8-wide hierarchy of fc_host.cpp on the CPU for a set of rays and counts node steps, child-box tests and
// primitive tests per ray, the way the device kernels visit them (4-wide: hit children sorted by entry distance; 8-wide:
// leaf children of a node first, interior children in octant order), and returns the closest hits so that the
// hierarchy can be checked against the 4-wide one before any kernel runs on it.
// Build: see tools/widestats/build.sh. Use: tools/wide_stats.py.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../include/fcgpu.h"
#include "../../ray-tracing_amd/csrc/fc_host.h"

using namespace fcg;

namespace
{
    struct Ray
    {
        double o[3], d[3], inv[3];
        double t_max;
    };

    bool tri_test(const TriRec& T, const Ray& r, double t_max, double& t)
    {
        // Moeller-Trumbore in double: statistics and id checks only
        double e1[3], e2[3], p[3], s[3], q[3];
        for(int i = 0; i < 3; ++i)
        {
            e1[i] = double(T.v1[i]) - T.v0[i];
            e2[i] = double(T.v2[i]) - T.v0[i];
            s[i] = r.o[i] - T.v0[i];
        }
        p[0] = r.d[1] * e2[2] - r.d[2] * e2[1];
        p[1] = r.d[2] * e2[0] - r.d[0] * e2[2];
        p[2] = r.d[0] * e2[1] - r.d[1] * e2[0];
        const double det = e1[0] * p[0] + e1[1] * p[1] + e1[2] * p[2];
        if(det == 0.0) return false;
        const double u = (s[0] * p[0] + s[1] * p[1] + s[2] * p[2]) / det;
        if(u < 0.0 || u > 1.0) return false;
        q[0] = s[1] * e1[2] - s[2] * e1[1];
        q[1] = s[2] * e1[0] - s[0] * e1[2];
        q[2] = s[0] * e1[1] - s[1] * e1[0];
        const double v = (r.d[0] * q[0] + r.d[1] * q[1] + r.d[2] * q[2]) / det;
        if(v < 0.0 || u + v > 1.0) return false;
        t = (e2[0] * q[0] + e2[1] * q[1] + e2[2] * q[2]) / det;
        return t > 0.0 && t <= t_max;
    }

    bool xprim_test(const XPrim<double>& P, uint32_t kind, const Ray& r, double t_max, double& t)
    {
        double o[3], d[3];
        for(int i = 0; i < 3; ++i)
        {
            o[i] = P.inv[4 * i] * r.o[0] + P.inv[4 * i + 1] * r.o[1] + P.inv[4 * i + 2] * r.o[2] + P.inv[4 * i + 3];
            d[i] = P.inv[4 * i] * r.d[0] + P.inv[4 * i + 1] * r.d[1] + P.inv[4 * i + 2] * r.d[2];
        }
        if(kind == KIND_SPHERE)
        {
            const double a = d[0] * d[0] + d[1] * d[1] + d[2] * d[2], b = 2 * (o[0] * d[0] + o[1] * d[1] + o[2] * d[2]);
            const double c = o[0] * o[0] + o[1] * o[1] + o[2] * o[2] - P.a * P.a, disc = b * b - 4 * a * c;
            if(disc < 0) return false;
            const double sq = std::sqrt(disc), t0 = (-b - sq) / (2 * a), t1 = (-b + sq) / (2 * a);
            t = t0 >= 0 ? t0 : t1;
            return t >= 0 && t <= t_max;
        }
        t = -o[1] / d[1];
        if(!(t >= 0 && t <= t_max)) return false;
        const double x = o[0] + d[0] * t, z = o[2] + d[2] * t;
        return std::fabs(x) <= P.a / 2 && std::fabs(z) <= P.b / 2;
    }

    struct Walker
    {
        const FlatScene& F;
        uint64_t steps{}, child_tests{}, prim_tests{}, max_stack{};

        void leaf_run(uint32_t kind, uint32_t first, uint32_t count, const Ray& r, double& t_max, uint32_t& pk)
        {
            for(uint32_t j = 0; j < count; ++j)
            {
                double t;
                prim_tests += 1;
                const TriRec& T = F.tris[first + j];
                if(kind == KIND_TRI && T.attr == XPRIM_PROXY)
                {
                    if(xprim_test(F.f64.xprims[T.prim], T.entity, r, t_max, t)) t_max = t, pk = T.entity << 30 | T.prim;
                }
                else if(kind == KIND_TRI)
                {
                    if(tri_test(T, r, t_max, t)) t_max = t, pk = first + j;
                }
                else if(xprim_test(F.f64.xprims[first], kind, r, t_max, t))
                    t_max = t, pk = kind << 30 | first;
            }
        }

        static bool slab(const float* org, const double* step, const uint32_t* ql, const uint32_t* qh, const Ray& r, double t_max, double& tn)
        {
            tn = 0.0;
            double tf = t_max;
            for(int a = 0; a < 3; ++a)
            {
                const double lo = org[a] + ql[a] * step[a], hi = org[a] + qh[a] * step[a];
                double t0 = (lo - r.o[a]) * r.inv[a], t1 = (hi - r.o[a]) * r.inv[a];
                if(t0 > t1) std::swap(t0, t1);
                tn = std::max(tn, t0);
                tf = std::min(tf, t1);
            }
            return tn <= tf * 1.0000005;
        }

        // 4-wide: position of every node in level (breadth-first) order, the order an LDS copy of the top of the tree would
        // hold (fc_host.cpp: treelet_first); steps on the first 85 / 341 nodes = levels 0..3 / 0..4 of a full tree
        std::vector<uint32_t> level_rank;
        uint64_t steps_top85 = 0, steps_top341 = 0;
        void rank_levels()
        {
            level_rank.assign(F.nodes.size(), 0xFFFFFFFFu);
            std::vector<uint32_t> order{0u};
            for(size_t k = 0; k < order.size(); ++k)
            {
                level_rank[order[k]] = static_cast<uint32_t>(k);
                for(int c = 0; c < 4; ++c)
                {
                    const uint32_t ch = F.nodes[order[k]].child[c];
                    if(ch != INVALID_U32 && !child_is_leaf(ch)) order.push_back(ch);
                }
            }
        }

        uint32_t walk4(const Ray& r, double& t_max)
        {
            uint32_t pk = INVALID_U32;
            std::vector<uint32_t> stack{0u};
            while(!stack.empty())
            {
                max_stack = std::max<uint64_t>(max_stack, stack.size());
                const uint32_t cur = stack.back();
                stack.pop_back();
                if(child_is_leaf(cur))
                {
                    leaf_run(leaf_kind(cur), leaf_first(cur), leaf_count(cur), r, t_max, pk);
                    continue;
                }
                const Node4& N = F.nodes[cur];
                steps += 1;
                if(!level_rank.empty())
                {
                    steps_top85 += level_rank[cur] < 85u ? 1 : 0;
                    steps_top341 += level_rank[cur] < 341u ? 1 : 0;
                }
                const float org[3]{N.ox, N.oy, N.oz};
                const double step[3]{N.sx, N.sy, N.sz};
                std::pair<double, uint32_t> hit[4];
                int nh = 0;
                for(int c = 0; c < 4; ++c)
                {
                    if(N.child[c] == INVALID_U32) continue;
                    child_tests += 1;
                    uint32_t ql[3], qh[3];
                    for(int a = 0; a < 3; ++a) ql[a] = N.qlo[a] >> (8 * c) & 255u, qh[a] = N.qhi[a] >> (8 * c) & 255u;
                    double tn;
                    if(slab(org, step, ql, qh, r, t_max, tn)) hit[nh++] = {tn, N.child[c]};
                }
                static const bool partial = std::getenv("WIDESTATS_PARTIAL_ORDER") != nullptr;  // what-if: only the nearest child is found, the others keep slot order
                static const bool slot_order = std::getenv("WIDESTATS_SLOT_ORDER") != nullptr;  // what-if: no ordering at all (first hit child in slot order is visited first)
                if(slot_order)
                    std::reverse(hit, hit + nh);
                else if(partial)
                {
                    int best = 0;
                    for(int k = 1; k < nh; ++k) if(hit[k].first < hit[best].first) best = k;
                    if(nh > 0) std::swap(hit[best], hit[nh - 1]);
                }
                else
                std::sort(hit, hit + nh, [](auto& a, auto& b) { return a.first > b.first; });  // far first: the nearest is popped first
                for(int k = 0; k < nh; ++k) stack.push_back(hit[k].second);
            }
            return pk;
        }

        uint32_t walk8(const Ray& r, double& t_max)
        {
            uint32_t pk = INVALID_U32;
            const uint32_t oct = (r.d[0] < 0 ? 1u : 0u) | (r.d[1] < 0 ? 2u : 0u) | (r.d[2] < 0 ? 4u : 0u);
            struct Group
            {
                uint32_t base, hits_key, imask;
            };
            std::vector<Group> stack;
            uint32_t cur = 0;
            for(;;)
            {
                const Node8& N = F.nodes8[cur];
                steps += 1;
                const float org[3]{N.ox, N.oy, N.oz};
                double step[3];
                for(int a = 0; a < 3; ++a) step[a] = std::ldexp(1.0, int(N.exps >> (8 * a) & 255u) - 127);
                const uint32_t imask = N.exps >> 24, valid = N.child_base >> 24;
                uint32_t int_key = 0, leaf_key = 0;
                for(uint32_t sl = 0; sl < 8; ++sl)
                {
                    if(!(valid >> sl & 1u)) continue;
                    child_tests += 1;
                    uint32_t ql[3], qh[3];
                    for(int a = 0; a < 3; ++a) ql[a] = N.qlo[a][sl >> 2] >> (8 * (sl & 3)) & 255u, qh[a] = N.qhi[a][sl >> 2] >> (8 * (sl & 3)) & 255u;
                    double tn;
                    if(slab(org, step, ql, qh, r, t_max, tn)) ((imask >> sl & 1u) ? int_key : leaf_key) |= 1u << (sl ^ oct);
                }
                // leaf children of this node first, in octant order
                const uint32_t leaf_slots = valid & ~imask;
                for(uint32_t k = 0; k < 8; ++k)
                {
                    if(!(leaf_key >> k & 1u)) continue;
                    const uint32_t sl = k ^ oct, below = (1u << sl) - 1u;
                    const uint32_t offset = __builtin_popcount(leaf_slots & below) + __builtin_popcount(N.two_mask & below);
                    leaf_run(KIND_TRI, N.tri_base + offset, (N.two_mask >> sl & 1u) ? 2u : 1u, r, t_max, pk);
                }
                if(int_key) stack.push_back(Group{N.child_base & 0x00FFFFFFu, int_key, imask});
                max_stack = std::max<uint64_t>(max_stack, stack.size());
                if(stack.empty()) break;
                Group& g = stack.back();
                const uint32_t k = static_cast<uint32_t>(__builtin_ctz(g.hits_key)), sl = k ^ oct;
                cur = g.base + static_cast<uint32_t>(__builtin_popcount(g.imask & ((1u << sl) - 1u)));
                g.hits_key &= g.hits_key - 1u;
                if(!g.hits_key) stack.pop_back();
            }
            return pk;
        }
    };
}

// stats: [0] node steps, [1] child-box tests, [2] primitive tests, [3] deepest stack, [4] nodes, [5] max_stack the builder states,
// [6] / [7] 4-wide: node steps on the first 85 / 341 nodes in level order
extern "C" int wide_stats(const fcgpu_scene_desc* scene, int width, uint64_t n, const fcgpu_ray* rays, uint32_t* pk_out, double* t_out, uint64_t* stats)
{
    FlatScene F;
    std::string err;
    if(flatten_scene(*scene, F, err, nullptr, false, width) != FCGPU_OK) return -1;
    Walker W{F};
    if(width != 8) W.rank_levels();
    for(uint64_t i = 0; i < n; ++i)
    {
        Ray r;
        for(int a = 0; a < 3; ++a)
        {
            r.o[a] = rays[i].origin[a];
            r.d[a] = rays[i].direction[a];
            r.inv[a] = 1.0 / r.d[a];
            if(std::isinf(r.inv[a])) r.inv[a] = std::copysign(1e300, r.inv[a]);
        }
        double t_max = rays[i].t_max;
        const uint32_t pk = width == 8 ? W.walk8(r, t_max) : W.walk4(r, t_max);
        // report the global primitive id so that the two hierarchies (whose triangle orders differ) can be compared
        uint32_t prim = INVALID_U32;
        if(pk != INVALID_U32) prim = (pk >> 30) == KIND_TRI ? F.tris[pk & 0x3FFFFFFFu].prim : F.f64.xprims[pk & 0x3FFFFFFFu].prim;
        if(pk_out) pk_out[i] = prim;
        if(t_out) t_out[i] = pk != INVALID_U32 ? t_max : std::numeric_limits<double>::infinity();
    }
    stats[0] = W.steps, stats[1] = W.child_tests, stats[2] = W.prim_tests, stats[3] = W.max_stack;
    stats[4] = width == 8 ? F.nodes8.size() : F.nodes.size();
    stats[5] = F.max_stack;
    stats[6] = W.steps_top85, stats[7] = W.steps_top341;
    return 0;
}
