// fc_host.cpp -- host-side scene flattening and BVH construction.
//
// What it replaces in the reference (all on the host, one-off, outside the timed region):
//   * pr_transform / prs_transform matrix construction      src/core/transform.hpp:11-17, 72-82
//   * mesh_surface's world-space vertex pre-transform        src/surfaces/mesh_surface.hpp:16-48
//   * entity_scene's entity -> primitive flattening          src/core/scene.hpp:39-62
//   * bvh_acceleration_structure's top-down SAH build        src/acceleration_structures/bvh_acceleration_structure.hpp:10-24, 228-365
//   * material -> lobe list (the arena-allocated bsdf)        src/materials/*.hpp, src/core/bsdf.hpp:13-44
//   * distribution_1d over emitter triangle areas            src/core/distribution.hpp:32-60
// The binary tree is built with the reference's algorithm (same buckets, same costs, same
// std::nth_element / std::partition calls) and then collapsed into the 4-wide quantised
// layout of fc_types.h, which is what the device traverses.
#include "fc_host.h"
#include "fc_material.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <thread>
#include <limits>

namespace fcg
{
    namespace
    {
        // ------------------------------------------------------------ double-precision transforms
        struct M4
        {
            double m[4][4];
        };

        M4 identity()
        {
            M4 r{};
            for(int i = 0; i < 4; ++i) r.m[i][i] = 1.0;
            return r;
        }
        M4 mul(const M4& a, const M4& b)
        {
            M4 c{};
            for(int i = 0; i < 4; ++i)
                for(int j = 0; j < 4; ++j)
                    c.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j] + a.m[i][3] * b.m[3][j];
            return c;
        }
        M4 translate(const double* t, double sign)
        {
            M4 r = identity();
            r.m[0][3] = sign * t[0];
            r.m[1][3] = sign * t[1];
            r.m[2][3] = sign * t[2];
            return r;
        }
        M4 scale3(double x, double y, double z)
        {
            M4 r = identity();
            r.m[0][0] = x;
            r.m[1][1] = y;
            r.m[2][2] = z;
            return r;
        }
        M4 rot_x(double th)
        {
            double s = std::sin(th), c = std::cos(th);
            M4 r = identity();
            r.m[1][1] = c; r.m[1][2] = -s;
            r.m[2][1] = s; r.m[2][2] = c;
            return r;
        }
        M4 rot_y(double th)
        {
            double s = std::sin(th), c = std::cos(th);
            M4 r = identity();
            r.m[0][0] = c; r.m[0][2] = s;
            r.m[2][0] = -s; r.m[2][2] = c;
            return r;
        }
        M4 rot_z(double th)
        {
            double s = std::sin(th), c = std::cos(th);
            M4 r = identity();
            r.m[0][0] = c; r.m[0][1] = -s;
            r.m[1][0] = s; r.m[1][1] = c;
            return r;
        }

        struct Xf
        {
            M4 t, inv;
        };

        // transform.hpp:14-17 (pr) and :79-82 (prs). The reference's single-argument and
        // two-argument constructors produce the same matrices as the general product with
        // zero rotation / unit scale up to the sign of zero, so one routine serves all three.
        Xf make_xf(const double* pos, const double* rot, const double* scl)
        {
            Xf x;
            bool has_rot = rot[0] != 0.0 || rot[1] != 0.0 || rot[2] != 0.0;
            bool has_scl = scl != nullptr && (scl[0] != 1.0 || scl[1] != 1.0 || scl[2] != 1.0);
            if(!has_rot && !has_scl)
            {
                x.t = translate(pos, 1.0);
                x.inv = translate(pos, -1.0);
                return x;
            }
            x.t = mul(mul(mul(translate(pos, 1.0), rot_y(rot[1])), rot_x(rot[0])), rot_z(rot[2]));
            x.inv = mul(mul(mul(rot_z(-rot[2]), rot_x(-rot[0])), rot_y(-rot[1])), translate(pos, -1.0));
            if(has_scl)
            {
                x.t = mul(x.t, scale3(scl[0], scl[1], scl[2]));
                x.inv = mul(scale3(1.0 / scl[0], 1.0 / scl[1], 1.0 / scl[2]), x.inv);
            }
            return x;
        }

        struct D3
        {
            double x, y, z;
        };
        D3 xpoint(const M4& m, D3 p)
        {
            return {m.m[0][0] * p.x + m.m[0][1] * p.y + m.m[0][2] * p.z + m.m[0][3],
                    m.m[1][0] * p.x + m.m[1][1] * p.y + m.m[1][2] * p.z + m.m[1][3],
                    m.m[2][0] * p.x + m.m[2][1] * p.y + m.m[2][2] * p.z + m.m[2][3]};
        }
        // prs_transform::transform_normal, transform.hpp:100-106 (transpose of the inverse, normalised)
        D3 xnormal(const M4& inv, D3 n)
        {
            double x = inv.m[0][0] * n.x + inv.m[1][0] * n.y + inv.m[2][0] * n.z;
            double y = inv.m[0][1] * n.x + inv.m[1][1] * n.y + inv.m[2][1] * n.z;
            double z = inv.m[0][2] * n.x + inv.m[1][2] * n.y + inv.m[2][2] * n.z;
            double len = std::sqrt(x * x + y * y + z * z);
            return {x / len, y / len, z / len};
        }

        // ------------------------------------------------------------ float bounds (bounds3f)
        struct BoxF
        {
            float lo[3]{std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
            float hi[3]{std::numeric_limits<float>::lowest(), std::numeric_limits<float>::lowest(), std::numeric_limits<float>::lowest()};

            void grow(const float* p)
            {
                for(int i = 0; i < 3; ++i)
                {
                    lo[i] = std::min(lo[i], p[i]);
                    hi[i] = std::max(hi[i], p[i]);
                }
            }
            void grow(const BoxF& b)
            {
                for(int i = 0; i < 3; ++i)
                {
                    lo[i] = std::min(lo[i], b.lo[i]);
                    hi[i] = std::max(hi[i], b.hi[i]);
                }
            }
            float area() const  // math.hpp:607-611
            {
                float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
                return 2.0f * (dx * dy + dx * dz + dy * dz);
            }
            int max_extent() const  // math.hpp:590-605
            {
                float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
                if(dx > dy && dx > dz) return 0;
                if(dy > dz) return 1;
                return 2;
            }
        };

        // pr_transform::transform_bounds of a local box, rounded to float (math.hpp:544-547)
        BoxF transformed_box(const M4& t, D3 lo, D3 hi)
        {
            double mn[3]{1e300, 1e300, 1e300}, mx[3]{-1e300, -1e300, -1e300};
            for(int c = 0; c < 8; ++c)
            {
                // TBounds3::Corner, math.hpp:575-578
                D3 p{(c & 1) ? hi.x : lo.x, (c & 4) ? hi.y : lo.y, (c & 2) ? hi.z : lo.z};
                D3 q = xpoint(t, p);
                mn[0] = std::min(mn[0], q.x); mn[1] = std::min(mn[1], q.y); mn[2] = std::min(mn[2], q.z);
                mx[0] = std::max(mx[0], q.x); mx[1] = std::max(mx[1], q.y); mx[2] = std::max(mx[2], q.z);
            }
            BoxF b;
            for(int i = 0; i < 3; ++i)
            {
                b.lo[i] = static_cast<float>(mn[i]);
                b.hi[i] = static_cast<float>(mx[i]);
            }
            return b;
        }

        // ------------------------------------------------------------ binary SAH build
        struct PrimRef  // primitive_info, bvh_acceleration_structure.hpp:200-226
        {
            uint32_t index;
            BoxF box;
            float centroid[3];
        };

        struct BinNode
        {
            BoxF box;
            uint32_t a;         // leaf: first primitive in `order`; interior: second child
            uint16_t b;         // leaf: primitive count; interior: split axis
            uint16_t interior;
        };

        struct BinaryBvh
        {
            std::vector<BinNode> nodes;
            std::vector<uint32_t> order;  // primitive indices in leaf order
        };

        class BinaryBuilder
        {
        public:
            BinaryBuilder(std::vector<PrimRef>& refs, BinaryBvh& out, bool reference_rule) : refs_{refs}, out_{out}, reference_rule_{reference_rule} {}

            // Builds the whole tree into out.nodes (depth-first, first child at index + 1) and out.order. The recursion is the
            // reference's (:228-365); large inputs are built by several threads: the top of the tree sequentially down to
            // ranges of `grain` primitives, those ranges in parallel (disjoint slices of refs), then one splice into
            // depth-first order. The tree does not depend on the thread count: a split looks at its own range only.
            void run()
            {
                const uint32_t n = static_cast<uint32_t>(refs_.size());
                unsigned threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
                if(const char* e = std::getenv("FCGPU_BUILD_THREADS")) threads = static_cast<unsigned>(std::min(64, std::max(1, std::atoi(e))));
                if(n < (1u << 16) || threads < 2)
                {
                    nodes_ = &out_.nodes;
                    build(0, n);
                }
                else
                {
                    std::vector<BinNode> top;
                    nodes_ = &top;
                    grain_ = std::max(4096u, n / (8u * threads));
                    build(0, n);
                    grain_ = 0;
                    std::vector<std::vector<BinNode>> sub(tasks_.size());
                    std::atomic<size_t> next{0};
                    std::atomic<bool> failed{false};
                    auto worker = [&]() {
                        try
                        {
                            for(size_t t = next++; t < tasks_.size(); t = next++)
                            {
                                BinaryBuilder b{refs_, out_, reference_rule_};
                                b.nodes_ = &sub[t];
                                b.build(tasks_[t].first, tasks_[t].second);
                            }
                        }
                        catch(...)
                        {
                            failed = true;  // out of memory in a worker: reported by the calling thread below
                        }
                    };
                    std::vector<std::thread> pool;
                    for(unsigned t = 1; t < threads; ++t) pool.emplace_back(worker);
                    worker();
                    for(auto& t : pool) t.join();
                    if(failed) throw std::bad_alloc{};
                    size_t total = top.size();
                    for(auto& v : sub) total += v.size();
                    out_.nodes.reserve(total);
                    splice(top, sub, 0);
                }
                out_.order.resize(n);
                for(uint32_t i = 0; i < n; ++i) out_.order[i] = refs_[i].index;  // leaves cover refs in order: first == begin
            }

        private:
            std::vector<PrimRef>& refs_;
            BinaryBvh& out_;
            std::vector<BinNode>* nodes_{};                      // where this builder writes (indices relative to it)
            uint32_t grain_{0};                                  // > 0: ranges of at most this many primitives become tasks
            std::vector<std::pair<uint32_t, uint32_t>> tasks_;   // deferred ranges; their placeholder has interior == 2, a = task
            static constexpr uint16_t DEFERRED = 2;

            uint32_t build(uint32_t begin, uint32_t end)  // :228-245
            {
                if(grain_ && end - begin <= grain_)
                {
                    const uint32_t index = static_cast<uint32_t>(nodes_->size());
                    nodes_->push_back(BinNode{BoxF{}, static_cast<uint32_t>(tasks_.size()), 0, DEFERRED});
                    tasks_.emplace_back(begin, end);
                    return index;
                }
                BoxF box = refs_[begin].box;
                for(uint32_t i = begin + 1; i < end; ++i) box.grow(refs_[i].box);
                if(end - begin == 1) return leaf(begin, end, box);
                return interior(begin, end, box);
            }

            // appends the subtree rooted at top[t] to out_.nodes in depth-first order
            void splice(const std::vector<BinNode>& top, const std::vector<std::vector<BinNode>>& sub, uint32_t t)
            {
                const BinNode& node = top[t];
                if(node.interior == DEFERRED)
                {
                    const uint32_t offset = static_cast<uint32_t>(out_.nodes.size());
                    for(BinNode b : sub[node.a])
                    {
                        if(b.interior) b.a += offset;
                        out_.nodes.push_back(b);
                    }
                    return;
                }
                const uint32_t index = static_cast<uint32_t>(out_.nodes.size());
                out_.nodes.push_back(node);
                if(!node.interior) return;
                splice(top, sub, t + 1);
                out_.nodes[index].a = static_cast<uint32_t>(out_.nodes.size());
                splice(top, sub, node.a);
            }
            // reference_rule_: the reference's split search -- 12 buckets along the longest centroid axis (:289-326).
            // Otherwise the three axes with 32 buckets each: 17 % fewer node steps per ray on the room scene.
            bool reference_rule_;

            uint32_t leaf(uint32_t begin, uint32_t end, const BoxF& box)  // :247-260
            {
                uint32_t index = static_cast<uint32_t>(nodes_->size());
                nodes_->push_back(BinNode{box, begin, static_cast<uint16_t>(end - begin), 0});
                return index;
            }

            uint32_t interior(uint32_t begin, uint32_t end, const BoxF& box)  // :262-365
            {
                BoxF cb;
                for(int k = 0; k < 3; ++k) cb.lo[k] = cb.hi[k] = refs_[begin].centroid[k];
                for(uint32_t i = begin + 1; i < end; ++i) cb.grow(refs_[i].centroid);

                int axis = cb.max_extent();
                float axis_length = cb.hi[axis] - cb.lo[axis];
                if(axis_length == 0.0f) return leaf(begin, end, box);

                uint32_t count = end - begin;
                uint32_t middle = begin + count / 2;
                if(count <= 4)
                {
                    std::nth_element(refs_.begin() + begin, refs_.begin() + middle, refs_.begin() + end,
                                     [axis](const PrimRef& a, const PrimRef& b) { return a.centroid[axis] < b.centroid[axis]; });
                }
                else
                {
                    const int bucket_count = reference_rule_ ? 12 : 32;
                    struct Bucket
                    {
                        uint32_t count{};
                        BoxF box{};
                    };
                    double min_cost = 0.0;
                    int min_index = -1, min_axis = axis;
                    for(int a = 0; a < 3; ++a)
                    {
                        if(reference_rule_ && a != axis) continue;
                        const float len = cb.hi[a] - cb.lo[a];
                        if(len == 0.0f) continue;
                        Bucket buckets[32]{};
                        for(uint32_t i = begin; i < end; ++i)
                        {
                            float offset = (refs_[i].centroid[a] - cb.lo[a]) / len;
                            int bi = std::min(static_cast<int>(offset * bucket_count), bucket_count - 1);
                            buckets[bi].count += 1;
                            buckets[bi].box.grow(refs_[i].box);
                        }
                        // right[i] = union of buckets i+1 .. (min / max are exact: the order of the unions does not matter)
                        Bucket right[32]{};
                        for(int i = bucket_count - 2; i >= 0; --i)
                        {
                            right[i] = right[i + 1];
                            right[i].box.grow(buckets[i + 1].box);
                            right[i].count += buckets[i + 1].count;
                        }
                        Bucket b0{};
                        for(int i = 0; i < bucket_count - 1; ++i)
                        {
                            b0.box.grow(buckets[i].box);
                            b0.count += buckets[i].count;
                            const Bucket& b1 = right[i];
                            const double cost = 0.125 + (b0.count * b0.box.area() + b1.count * b1.box.area()) / box.area();
                            if(min_index < 0 || cost < min_cost)
                            {
                                min_cost = cost;
                                min_index = i;
                                min_axis = a;
                            }
                        }
                    }
                    double leaf_cost = static_cast<double>(count);
                    if(min_cost < leaf_cost)
                    {
                        axis = min_axis;
                        axis_length = cb.hi[axis] - cb.lo[axis];
                        float partition_point = cb.lo[axis] + axis_length / bucket_count * (min_index + 1);
                        auto it = std::partition(refs_.begin() + begin, refs_.begin() + end,
                                                 [axis, partition_point](const PrimRef& a) { return a.centroid[axis] < partition_point; });
                        middle = static_cast<uint32_t>(std::distance(refs_.begin(), it));
                        if(middle == begin || middle == end)
                        {
                            // the float partition point fell outside the occupied buckets (a centroid range of a few
                            // ulps): split at the median instead, or the recursion would never end
                            middle = begin + count / 2;
                            std::nth_element(refs_.begin() + begin, refs_.begin() + middle, refs_.begin() + end,
                                             [axis](const PrimRef& a, const PrimRef& b) { return a.centroid[axis] < b.centroid[axis]; });
                        }
                    }
                    else
                    {
                        return leaf(begin, end, box);
                    }
                }

                uint32_t index = static_cast<uint32_t>(nodes_->size());
                nodes_->emplace_back();
                build(begin, middle);
                uint32_t right = build(middle, end);
                (*nodes_)[index] = BinNode{box, right, static_cast<uint16_t>(axis), 1};
                return index;
            }
        };

        // ------------------------------------------------------------ hierarchy from the device builder
        // Lays the device's (left, right, box) hierarchy out as the depth-first BinNode array the host builder
        // produces (first child at index + 1), one primitive per leaf, and leaves refs in leaf order.
        int build_on_device(const HierarchyBuilder& builder, std::vector<PrimRef>& refs, BinaryBvh& out, std::string& err)
        {
            const size_t n = refs.size();
            std::vector<float> boxes(6 * n);
            for(size_t i = 0; i < n; ++i)
                for(int a = 0; a < 3; ++a)
                {
                    boxes[6 * i + a] = refs[i].box.lo[a];
                    boxes[6 * i + 3 + a] = refs[i].box.hi[a];
                }
            LbvhTree tree;
            int rc = builder(boxes, tree, err);
            if(rc) return rc;
            if(tree.order.size() != n || tree.left.size() != n - 1 || tree.right.size() != n - 1 || tree.node_boxes.size() != 6 * (n - 1))
            {
                err = "device hierarchy has the wrong size";
                return FCGPU_E_HIP;
            }
            std::vector<PrimRef> sorted(n);
            std::vector<uint8_t> seen(n, 0);
            for(size_t k = 0; k < n; ++k)
            {
                if(tree.order[k] >= n || seen[tree.order[k]])
                {
                    err = "device hierarchy: leaf order is not a permutation";
                    return FCGPU_E_HIP;
                }
                seen[tree.order[k]] = 1;
                sorted[k] = refs[tree.order[k]];
            }
            refs.swap(sorted);
            out.nodes.clear();
            out.nodes.reserve(2 * n - 1);
            out.order.resize(n);
            for(size_t k = 0; k < n; ++k) out.order[k] = refs[k].index;
            // explicit stack: (child reference, parent whose `a` is to receive this node's index or INVALID_U32)
            std::vector<std::pair<int32_t, uint32_t>> stack;
            stack.emplace_back(0, INVALID_U32);
            size_t emitted_internal = 0;
            while(!stack.empty())
            {
                auto [ref, parent] = stack.back();
                stack.pop_back();
                const uint32_t index = static_cast<uint32_t>(out.nodes.size());
                if(parent != INVALID_U32) out.nodes[parent].a = index;
                if(ref < 0)
                {
                    const uint32_t k = static_cast<uint32_t>(~ref);
                    if(k >= n)
                    {
                        err = "device hierarchy: leaf out of range";
                        return FCGPU_E_HIP;
                    }
                    out.nodes.push_back(BinNode{refs[k].box, k, 1, 0});
                    continue;
                }
                if(static_cast<size_t>(ref) >= n - 1 || ++emitted_internal > n - 1)
                {
                    err = "device hierarchy: not a tree";
                    return FCGPU_E_HIP;
                }
                BinNode node{};
                for(int a = 0; a < 3; ++a)
                {
                    node.box.lo[a] = tree.node_boxes[6 * static_cast<size_t>(ref) + a];
                    node.box.hi[a] = tree.node_boxes[6 * static_cast<size_t>(ref) + 3 + a];
                }
                node.interior = 1;
                node.b = static_cast<uint16_t>(node.box.max_extent());
                out.nodes.push_back(node);
                stack.emplace_back(tree.right[ref], index);      // second child: its index lands in `a`
                stack.emplace_back(tree.left[ref], INVALID_U32);  // first child: next in depth-first order
            }
            if(out.nodes.size() != 2 * n - 1)
            {
                err = "device hierarchy: node count mismatch";
                return FCGPU_E_HIP;
            }
            return FCGPU_OK;
        }

        // ------------------------------------------------------------ 4-wide collapse + quantisation
        struct WideChild
        {
            BoxF box;
            bool leaf;
            uint32_t bin;    // interior child: binary node, or INVALID_U32 for a folded group (`first` = pending index)
            uint32_t kind;   // leaf: primitive kind
            uint32_t first;  // leaf: index into tris / xprims
            uint32_t count;
        };

        // Collapses the binary tree into 4-wide nodes. Binary subtrees holding at most MAX_LEAF_PRIMS
        // triangles become ONE leaf (a run of consecutive TriRec), which removes the bottom level of
        // box tests and dependent loads; spheres and planes are single-primitive leaves.
        class WideBuilder
        {
        public:
            // order_box[k] / order_ki[k]: bounds and kind|index of the k-th primitive in binary leaf order
            WideBuilder(const BinaryBvh& bin, const std::vector<BoxF>& order_box, const std::vector<uint32_t>& order_ki, FlatScene& out, int width = 4)
                : bin_{bin}, order_box_{order_box}, order_ki_{order_ki}, out_{out}, width_{static_cast<size_t>(width)}
            {
            }

            // 8-wide: the triangle records in the order the nodes reference them (indices into the leaf-ordered array the
            // builder was given; a sphere / plane leaf is XPRIM_PROXY | kind << 24 | its xprims index)
            std::vector<uint32_t> tri_order8;
            std::string error;

            void run()
            {
                // per binary node: its range in leaf order and whether it holds triangles only
                range_.assign(bin_.nodes.size(), {0u, 0u});
                tri_only_.assign(bin_.nodes.size(), 0);
                for(size_t n = bin_.nodes.size(); n-- > 0;)
                {
                    const BinNode& b = bin_.nodes[n];
                    if(!b.interior)
                    {
                        range_[n] = {b.a, static_cast<uint32_t>(b.b)};
                        bool t = true;
                        for(uint32_t k = b.a; k < b.a + b.b; ++k) t = t && (order_ki_[k] >> 30) == KIND_TRI;
                        tri_only_[n] = t;
                    }
                    else
                    {
                        auto l = range_[n + 1], r = range_[b.a];
                        range_[n] = {l.first, l.second + r.second};
                        tri_only_[n] = tri_only_[n + 1] && tri_only_[b.a];
                    }
                }
                std::vector<WideChild> kids;
                gather(0, kids);
                if(width_ == 8)
                {
                    out_.nodes8.assign(1, Node8{});
                    out_.max_stack = emit8(0, kids);
                    if(out_.nodes8.size() > NODE8_MAX_NODES) error = "8-wide hierarchy: more than 2^24 nodes";
                }
                else
                    out_.max_stack = emit(kids);
            }

        private:
            const BinaryBvh& bin_;
            const std::vector<BoxF>& order_box_;
            const std::vector<uint32_t>& order_ki_;
            FlatScene& out_;
            std::vector<std::pair<uint32_t, uint32_t>> range_;
            std::vector<char> tri_only_;
            size_t width_;
            uint32_t max_leaf_ = leaf_limit();

            static uint32_t leaf_limit()  // FCGPU_MAX_LEAF (development knob): 1 .. MAX_LEAF_PRIMS
            {
                const char* e = std::getenv("FCGPU_MAX_LEAF");
                const int v = e ? std::atoi(e) : MAX_LEAF_PRIMS;
                return static_cast<uint32_t>(std::min(std::max(v, 1), MAX_LEAF_PRIMS));
            }

            bool is_leafish(uint32_t b) const
            {
                return !bin_.nodes[b].interior || (tri_only_[b] && range_[b].second <= max_leaf_);
            }

            // leaf runs covering the primitives of binary node b, inserted at `at`
            size_t push_runs(uint32_t b, std::vector<WideChild>& kids, size_t at)
            {
                uint32_t k = range_[b].first, end = k + range_[b].second;
                size_t added = 0;
                while(k < end)
                {
                    uint32_t kind = order_ki_[k] >> 30;
                    uint32_t n = 1;
                    if(kind == KIND_TRI)
                        while(k + n < end && n < max_leaf_ && (order_ki_[k + n] >> 30) == KIND_TRI) ++n;
                    BoxF box;
                    for(uint32_t q = k; q < k + n; ++q) box.grow(order_box_[q]);
                    kids.insert(kids.begin() + at + added, WideChild{box, true, 0u, kind, order_ki_[k] & 0x3FFFFFFFu, n});
                    added += 1;
                    k += n;
                }
                return added;
            }

            size_t push_bin(uint32_t b, std::vector<WideChild>& kids, size_t at)
            {
                if(is_leafish(b)) return push_runs(b, kids, at);
                kids.insert(kids.begin() + at, WideChild{bin_.nodes[b].box, false, b, 0u, 0u, 0u});
                return 1;
            }

            // children of binary node b, opened until four (largest surface area first)
            void gather(uint32_t b, std::vector<WideChild>& kids)
            {
                kids.clear();
                if(is_leafish(b))
                {
                    push_runs(b, kids, 0);
                    return;
                }
                const BinNode& n = bin_.nodes[b];
                push_bin(b + 1, kids, 0);
                push_bin(n.a, kids, kids.size());
                while(kids.size() < width_)
                {
                    int best = -1;
                    float best_area = -1.0f;
                    for(size_t i = 0; i < kids.size(); ++i)
                    {
                        if(!kids[i].leaf && kids[i].box.area() > best_area)
                        {
                            best_area = kids[i].box.area();
                            best = static_cast<int>(i);
                        }
                    }
                    if(best < 0) break;
                    uint32_t bb = kids[best].bin;
                    kids.erase(kids.begin() + best);
                    const BinNode& nb = bin_.nodes[bb];
                    size_t added = push_bin(bb + 1, kids, best);
                    push_bin(nb.a, kids, best + added);
                }
            }

            // writes one wide node for `kids` (folding surplus children under extra nodes) and
            // returns the worst-case stack depth needed below it
            uint32_t emit(std::vector<WideChild> kids)
            {
                while(kids.size() > 4)
                {
                    std::vector<WideChild> tail(kids.end() - 4, kids.end());
                    kids.erase(kids.end() - 4, kids.end());
                    BoxF b;
                    for(auto& k : tail) b.grow(k.box);
                    pending_.push_back(std::move(tail));
                    kids.push_back(WideChild{b, false, INVALID_U32, 0u, static_cast<uint32_t>(pending_.size() - 1), 0u});
                }

                uint32_t index = static_cast<uint32_t>(out_.nodes.size());
                out_.nodes.emplace_back();
                BoxF box;
                for(auto& k : kids) box.grow(k.box);

                Node4 node{};
                quantise(box, kids, node);
                uint32_t deepest = 0;
                for(size_t c = 0; c < 4; ++c)
                {
                    if(c >= kids.size())
                    {
                        node.child[c] = INVALID_U32;
                        continue;
                    }
                    const WideChild& k = kids[c];
                    if(k.leaf)
                    {
                        node.child[c] = make_leaf(k.kind, k.first, k.count);
                    }
                    else
                    {
                        node.child[c] = static_cast<uint32_t>(out_.nodes.size());
                        std::vector<WideChild> sub;
                        if(k.bin == INVALID_U32)
                            sub = pending_[k.first];
                        else
                            gather(k.bin, sub);
                        deepest = std::max(deepest, emit(std::move(sub)));
                    }
                }
                out_.nodes[index] = node;
                // while one child is being traversed at most (n - 1) siblings wait on the stack
                return static_cast<uint32_t>(kids.size() - 1) + deepest;
            }

            // ---- 8-wide emission. Writes node `index` (already allocated) for `kids`, allocates its interior children as
            // consecutive nodes, appends the records of its leaf children to tri_order8, recurses; returns the worst-case
            // number of stack entries below it (the traversal pushes at most one group of pending siblings per level).
            uint32_t emit8(uint32_t index, std::vector<WideChild> kids)
            {
                while(kids.size() > 8)
                {
                    std::vector<WideChild> tail(kids.end() - 8, kids.end());
                    kids.erase(kids.end() - 8, kids.end());
                    BoxF b;
                    for(auto& k : tail) b.grow(k.box);
                    pending_.push_back(std::move(tail));
                    kids.push_back(WideChild{b, false, INVALID_U32, 0u, static_cast<uint32_t>(pending_.size() - 1), 0u});
                }
                BoxF box;
                for(auto& k : kids) box.grow(k.box);
                // slot assignment: child c prefers the slot whose corner its centre points at (bit a of the slot = the
                // child lies on the + side of axis a), greedily by the largest projection onto the slot's diagonal
                int slot_of[8], child_in[8];
                for(int i = 0; i < 8; ++i) slot_of[i] = child_in[i] = -1;
                float centre[3];
                for(int a = 0; a < 3; ++a) centre[a] = 0.5f * (box.lo[a] + box.hi[a]);
                for(size_t round = 0; round < kids.size(); ++round)
                {
                    float best = -std::numeric_limits<float>::max();
                    int bc = -1, bs = -1;
                    for(size_t c = 0; c < kids.size(); ++c)
                    {
                        if(slot_of[c] >= 0) continue;
                        for(int sl = 0; sl < 8; ++sl)
                        {
                            if(child_in[sl] >= 0) continue;
                            float score = 0.0f;
                            for(int a = 0; a < 3; ++a)
                                score += (0.5f * (kids[c].box.lo[a] + kids[c].box.hi[a]) - centre[a]) * ((sl >> a & 1) ? 1.0f : -1.0f);
                            if(score > best)
                            {
                                best = score;
                                bc = static_cast<int>(c);
                                bs = sl;
                            }
                        }
                    }
                    slot_of[bc] = bs;
                    child_in[bs] = bc;
                }
                Node8 node{};
                node.ox = box.lo[0];
                node.oy = box.lo[1];
                node.oz = box.lo[2];
                BoxF slot_box[8];
                bool used[8];
                for(int sl = 0; sl < 8; ++sl)
                {
                    used[sl] = child_in[sl] >= 0;
                    if(used[sl]) slot_box[sl] = kids[static_cast<size_t>(child_in[sl])].box;
                }
                uint32_t lo_words[3][2], hi_words[3][2];
                int exps[3];
                quantise_slots(box, slot_box, used, 8, lo_words, hi_words, exps);
                uint32_t imask = 0, valid = 0, two = 0, interior = 0;
                for(int sl = 0; sl < 8; ++sl)
                {
                    if(!used[sl]) continue;
                    valid |= 1u << sl;
                    if(!kids[static_cast<size_t>(child_in[sl])].leaf)
                    {
                        imask |= 1u << sl;
                        interior += 1;
                    }
                }
                for(int a = 0; a < 3; ++a)
                    for(int h = 0; h < 2; ++h)
                    {
                        node.qlo[a][h] = lo_words[a][h];
                        node.qhi[a][h] = hi_words[a][h];
                    }
                node.exps = static_cast<uint32_t>(exps[0] + 127) | static_cast<uint32_t>(exps[1] + 127) << 8 | static_cast<uint32_t>(exps[2] + 127) << 16 | imask << 24;
                const uint32_t child_base = static_cast<uint32_t>(out_.nodes8.size());
                out_.nodes8.resize(out_.nodes8.size() + interior);
                node.tri_base = static_cast<uint32_t>(tri_order8.size());
                for(int sl = 0; sl < 8; ++sl)
                {
                    if(!used[sl] || (imask >> sl & 1u)) continue;
                    const WideChild& k = kids[static_cast<size_t>(child_in[sl])];
                    if(k.kind == KIND_TRI)
                    {
                        if(k.count > 2) error = "8-wide hierarchy: a leaf of more than two triangles (FCG_MAX_LEAF_PRIMS must be <= 2)";
                        for(uint32_t q = 0; q < k.count; ++q) tri_order8.push_back(k.first + q);
                        if(k.count == 2) two |= 1u << sl;
                    }
                    else
                        tri_order8.push_back(XPRIM_PROXY_TAG | k.kind << 24 | k.first);
                }
                node.child_base = (child_base & 0x00FFFFFFu) | valid << 24;
                node.two_mask = two;
                out_.nodes8[index] = node;
                uint32_t deepest = 0, rank = 0;
                for(int sl = 0; sl < 8; ++sl)
                {
                    if(!used[sl] || !(imask >> sl & 1u)) continue;
                    const WideChild& k = kids[static_cast<size_t>(child_in[sl])];
                    std::vector<WideChild> sub;
                    if(k.bin == INVALID_U32)
                        sub = pending_[k.first];
                    else
                        gather(k.bin, sub);
                    deepest = std::max(deepest, emit8(child_base + rank, std::move(sub)));
                    rank += 1;
                }
                return interior ? 1u + deepest : 0u;
            }

        public:
            static constexpr uint32_t XPRIM_PROXY_TAG = 0x80000000u;  // in tri_order8: kind << 24 | xprims index follow

        private:
            // quantised bounds of up to `n` child slots on the node's power-of-two grid (step 2^e per axis), conservative
            // under the device's fp32 decode fma(q, step, origin); unused slots get the inverted interval [255, 0].
            // words[axis][h]: byte c of word h = slot 4 h + c
            static void quantise_slots(const BoxF& box, const BoxF* slot_box, const bool* used, int n, uint32_t (*lo_words)[2], uint32_t (*hi_words)[2], int* exps)
            {
                for(int a = 0; a < 3; ++a)
                {
                    float extent = box.hi[a] - box.lo[a];
                    int e = -100;
                    if(extent > 0.0f) e = static_cast<int>(std::ceil(std::log2(static_cast<double>(extent) / 255.0)));
                    e = std::max(e, -120);
                    for(;;)
                    {
                        float step = std::ldexp(1.0f, e);
                        bool ok = true;
                        uint32_t lw[2]{0, 0}, hw[2]{0, 0};
                        for(int c = 0; c < n && ok; ++c)
                        {
                            uint32_t ql = 255, qh = 0;
                            if(used[c])
                            {
                                double rl = (static_cast<double>(slot_box[c].lo[a]) - box.lo[a]) / step;
                                double rh = (static_cast<double>(slot_box[c].hi[a]) - box.lo[a]) / step;
                                long l = std::max(0l, static_cast<long>(std::floor(rl)));
                                long h = static_cast<long>(std::ceil(rh));
                                while(l > 0 && std::fmaf(static_cast<float>(l), step, box.lo[a]) > slot_box[c].lo[a]) --l;
                                while(h <= 255 && std::fmaf(static_cast<float>(h), step, box.lo[a]) < slot_box[c].hi[a]) ++h;
                                if(h > 255 || l > 255) ok = false;
                                ql = static_cast<uint32_t>(l);
                                qh = static_cast<uint32_t>(h);
                            }
                            lw[c >> 2] |= ql << (8 * (c & 3));
                            hw[c >> 2] |= qh << (8 * (c & 3));
                        }
                        if(ok)
                        {
                            for(int h = 0; h < 2; ++h)
                            {
                                lo_words[a][h] = lw[h];
                                hi_words[a][h] = hw[h];
                            }
                            exps[a] = e;
                            break;
                        }
                        ++e;
                    }
                }
            }

            std::vector<std::vector<WideChild>> pending_;

            static void quantise(const BoxF& box, const std::vector<WideChild>& kids, Node4& node)
            {
                node.ox = box.lo[0];
                node.oy = box.lo[1];
                node.oz = box.lo[2];
                float* steps[3]{&node.sx, &node.sy, &node.sz};
                for(int a = 0; a < 3; ++a)
                {
                    float extent = box.hi[a] - box.lo[a];
                    int e = -100;
                    if(extent > 0.0f)
                    {
                        e = static_cast<int>(std::ceil(std::log2(static_cast<double>(extent) / 255.0)));
                    }
                    e = std::max(e, -120);
                    for(;;)
                    {
                        float step = std::ldexp(1.0f, e);
                        bool ok = true;
                        uint32_t lo_word = 0, hi_word = 0;
                        for(size_t c = 0; c < 4 && ok; ++c)
                        {
                            uint32_t ql = 255, qh = 0;  // empty slot: inverted interval
                            if(c < kids.size())
                            {
                                double rl = (static_cast<double>(kids[c].box.lo[a]) - box.lo[a]) / step;
                                double rh = (static_cast<double>(kids[c].box.hi[a]) - box.lo[a]) / step;
                                long l = static_cast<long>(std::floor(rl));
                                long h = static_cast<long>(std::ceil(rh));
                                l = std::max(0l, l);
                                // conservative under the device's fp32 decode (fma(q, step, origin))
                                while(l > 0 && std::fmaf(static_cast<float>(l), step, box.lo[a]) > kids[c].box.lo[a]) --l;
                                while(h <= 255 && std::fmaf(static_cast<float>(h), step, box.lo[a]) < kids[c].box.hi[a]) ++h;
                                if(h > 255 || l > 255) ok = false;
                                ql = static_cast<uint32_t>(l);
                                qh = static_cast<uint32_t>(h);
                            }
                            lo_word |= ql << (8 * c);
                            hi_word |= qh << (8 * c);
                        }
                        if(ok)
                        {
                            node.qlo[a] = lo_word;
                            node.qhi[a] = hi_word;
                            *steps[a] = step;  // 2^e, e >= -120: a normal float
                            break;
                        }
                        ++e;
                    }
                }
            }
        };

#if FCG_TREELET
        // Experiment (FCG_TREELET = n > 0, tools/build_variant.sh): the top of the hierarchy, the first n wide nodes in
        // breadth-first order, is moved to indices 0 .. n-1, which the traversal kernels copy into LDS per block
        // (fc_traverse.h); all other nodes keep their depth-first order behind them.
        void treelet_first(std::vector<Node4>& nodes, uint32_t n)
        {
            std::vector<uint32_t> bfs{0u};
            for(size_t head = 0; head < bfs.size() && bfs.size() < n; ++head)
                for(uint32_t c : nodes[bfs[head]].child)
                    if(c != INVALID_U32 && !child_is_leaf(c) && bfs.size() < n) bfs.push_back(c);
            std::vector<uint32_t> new_index(nodes.size(), INVALID_U32);
            for(size_t k = 0; k < bfs.size(); ++k) new_index[bfs[k]] = static_cast<uint32_t>(k);
            uint32_t next = static_cast<uint32_t>(bfs.size());
            for(size_t k = 0; k < nodes.size(); ++k)
                if(new_index[k] == INVALID_U32) new_index[k] = next++;
            std::vector<Node4> moved(nodes.size());
            for(size_t k = 0; k < nodes.size(); ++k)
            {
                Node4 node = nodes[k];
                for(uint32_t& c : node.child)
                    if(c != INVALID_U32 && !child_is_leaf(c)) c = new_index[c];
                moved[new_index[k]] = node;
            }
            nodes.swap(moved);
        }
#endif

        // ------------------------------------------------------------ typed conversion
        template<typename R> void put12(R* dst, const M4& m)
        {
            for(int i = 0; i < 3; ++i)
                for(int j = 0; j < 4; ++j) dst[i * 4 + j] = static_cast<R>(m.m[i][j]);
        }

        using LobeD = LobeRec<double>;  // lobe in double, converted per precision

        MatValues<double> material_constants(const fcgpu_material& m)
        {
            MatValues<double> v{};
            for(int i = 0; i < 3; ++i)
            {
                v.color_a[i] = m.color_a[i];
                v.color_b[i] = m.color_b[i];
            }
            v.roughness[0] = m.roughness[0];
            v.roughness[1] = m.roughness[1];
            v.ior = m.ior;
            v.metalness = m.metalness;
            return v;
        }

        // materials -> lobes through fc_material.h (material::evaluate of each class + the bsdf constructor);
        // for a textured material the constants stand in, only to learn which lobe types it can produce
        int material_lobes(const fcgpu_material& m, std::vector<LobeD>& lobes, std::string& err)
        {
            if(m.kind < FCGPU_MATERIAL_DIFFUSE || m.kind > FCGPU_MATERIAL_TRANSMISSION)
            {
                err = "unknown material kind";
                return FCGPU_E_INVALID;
            }
            const MatValues<double> v = material_constants(m);
            int types[MAX_LOBES];
            double scales[MAX_LOBES], weights[MAX_LOBES], factor[MAX_LOBES], cdf[MAX_LOBES];
            const int n = material_layout(m.kind, v, types, scales, weights);
            lobe_pick_table(n, scales, weights, factor, cdf);
            for(int i = 0; i < n; ++i) lobes.push_back(material_lobe(m.kind, v, i, types[i], factor[i], cdf[i]));
            return n;
        }

        // TriShade records in arithmetic R, with the expressions make_surface_point used per hit (mesh_surface.hpp:211-248)
        template<typename R>
        void make_tri_shade(const std::vector<TriRec>& tris, const std::vector<TriAttr>& attrs, std::vector<TriShade<R>>& dst)
        {
            dst.resize(tris.size());
            for(size_t k = 0; k < tris.size(); ++k)
            {
                const TriRec& T = tris[k];
                vec3<R> p0{R(T.v0[0]), R(T.v0[1]), R(T.v0[2])}, p1{R(T.v1[0]), R(T.v1[1]), R(T.v1[2])}, p2{R(T.v2[0]), R(T.v2[1]), R(T.v2[2])};
                vec3<R> dp02 = p0 - p2, dp12 = p1 - p2;
                vec3<R> ng = normalize(cross(dp02, dp12));
                vec3<R> dpdu;
                if(T.attr == INVALID_U32 || T.attr >= attrs.size())  // (the placeholder record of a scene without triangles has attr 0)
                {
                    // default uvs (0,0),(1,0),(1,1) (:348-351): det = 1, dpdu = dp12 - dp02
                    dpdu = (R(-1) * dp02 - R(-1) * dp12) / R(1);
                }
                else
                {
                    const TriAttr& A = attrs[T.attr];
                    R d02x = R(A.uv0[0]) - R(A.uv2[0]), d02y = R(A.uv0[1]) - R(A.uv2[1]);
                    R d12x = R(A.uv1[0]) - R(A.uv2[0]), d12y = R(A.uv1[1]) - R(A.uv2[1]);
                    R det = d02x * d12y - d02y * d12x;
                    dpdu = (d12y * dp02 - d02y * dp12) / det;
                }
                vec3<R> t = normalize(dpdu);
                TriShade<R>& o = dst[k];
                o.ng[0] = ng.x; o.ng[1] = ng.y; o.ng[2] = ng.z;
                o.t[0] = t.x; o.t[1] = t.y; o.t[2] = t.z;
                o.entity = T.entity & ENTITY_MASK;
                o.attr = T.attr;
            }
        }

        template<typename R>
        void convert_lobes(const std::vector<LobeD>& src, std::vector<LobeRec<R>>& dst)
        {
            for(const LobeD& l : src)
            {
                LobeRec<R> r{};
                r.type = l.type;
                r.fresnel = l.fresnel;
                for(int i = 0; i < 3; ++i)
                {
                    r.color[i] = static_cast<R>(l.color[i]);
                    r.color2[i] = static_cast<R>(l.color2[i]);
                }
                r.alpha[0] = static_cast<R>(l.alpha[0]);
                r.alpha[1] = static_cast<R>(l.alpha[1]);
                r.ior = static_cast<R>(l.ior);
                r.factor = static_cast<R>(l.factor);
                r.cdf = l.cdf;
                dst.push_back(r);
            }
        }

        struct XPrimD
        {
            Xf xf;
            double a, b, inv_area;
            uint32_t prim, entity;
        };
        template<typename R>
        void convert_xprims(const std::vector<XPrimD>& src, std::vector<XPrim<R>>& dst)
        {
            for(const XPrimD& s : src)
            {
                XPrim<R> x{};
                put12(x.inv, s.xf.inv);
                put12(x.fwd, s.xf.t);
                x.a = static_cast<R>(s.a);
                x.b = static_cast<R>(s.b);
                x.prim = s.prim;
                x.entity = s.entity;
                x.inv_area = static_cast<R>(s.inv_area);
                dst.push_back(x);
            }
        }

        struct LightD
        {
            int32_t entity;
            uint32_t kind, index, first_prim, tri_count, cdf_offset;
            double Le[3], inv_area;
        };
        template<typename R>
        void convert_lights(const std::vector<LightD>& src, std::vector<LightRec<R>>& dst)
        {
            for(const LightD& s : src)
            {
                LightRec<R> l{};
                l.entity = s.entity;
                l.kind = s.kind;
                l.index = s.index;
                l.first_prim = s.first_prim;
                l.tri_count = s.tri_count;
                l.cdf_offset = s.cdf_offset;
                for(int i = 0; i < 3; ++i) l.Le[i] = static_cast<R>(s.Le[i]);
                l.inv_area = static_cast<R>(s.inv_area);
                dst.push_back(l);
            }
        }

        template<typename R>
        void convert_camera(const fcgpu_camera& c, CameraRec<R>& out)
        {
            const double zero[3]{0, 0, 0};
            (void)zero;
            Xf xf = make_xf(c.position, c.rotation, nullptr);
            put12(out.fwd, xf.t);
            // perspective_camera.hpp:19-27
            double focus = c.lens_radius == 0.0 ? 1.0 : c.focus_distance;
            double pixel = 2.0 * focus * std::tan(c.fov / 2.0) / static_cast<double>(c.height);
            out.plane_w = static_cast<R>(c.width * pixel);
            out.plane_h = static_cast<R>(c.height * pixel);
            out.focus = static_cast<R>(focus);
            out.lens_radius = static_cast<R>(c.lens_radius);
            out.W = c.width;
            out.H = c.height;
        }
        // ------------------------------------------------------------ images and image textures
        // texel decode of the reference's pixel classes: r8_image.hpp, rgb8_image.hpp, srgb8_image.hpp
        // (core/color.hpp:13-26), rgb32_image.hpp
        bool decode_image(const fcgpu_image& im, FlatImage& out, std::string& err)
        {
            if(im.width <= 0 || im.height <= 0 || im.pixels == nullptr || im.width > 32768 || im.height > 32768)
            {
                err = "image: bad resolution or null pixels";
                return false;
            }
            const size_t n = static_cast<size_t>(im.width) * static_cast<size_t>(im.height);
            out.width = im.width;
            out.height = im.height;
            out.rgb.assign(3 * n, 0.0);
            const uint8_t* b = static_cast<const uint8_t*>(im.pixels);
            auto srgb = [](uint8_t v) {
                double x = v / 255.0;
                return x <= 0.04045 ? x / 12.92 : std::pow((x + 0.055) / 1.055, 2.4);
            };
            switch(im.format)
            {
            case FCGPU_IMAGE_R8:
                for(size_t i = 0; i < n; ++i) out.rgb[3 * i] = b[i] / 255.0;
                break;
            case FCGPU_IMAGE_RGB8:
                for(size_t i = 0; i < 3 * n; ++i) out.rgb[i] = b[i] / 255.0;
                break;
            case FCGPU_IMAGE_SRGB8:
                for(size_t i = 0; i < 3 * n; ++i) out.rgb[i] = srgb(b[i]);
                break;
            case FCGPU_IMAGE_RGB32:
                for(size_t i = 0; i < 3 * n; ++i)
                {
                    float v;
                    std::memcpy(&v, b + 4 * i, 4);
                    out.rgb[i] = static_cast<double>(v);
                }
                break;
            default:
                err = "image: unknown format";
                return false;
            }
            return true;
        }

        struct TexD  // image_texture_2d_rgb in double, as the reference evaluates it on the host
        {
            const FlatImage* im;
            int filter;
            D3 texel(int x, int y) const
            {
                const double* p = &im->rgb[3 * (static_cast<size_t>(im->width) * y + x)];
                return {p[0], p[1], p[2]};
            }
            D3 evaluate(double u, double v) const  // image_texture.hpp:22-33, 88-124
            {
                const int W = im->width, H = im->height;
                if(filter == FCGPU_FILTER_BILINEAR)
                {
                    double ax = u * W - 0.5, ay = v * H - 0.5;
                    int x0 = static_cast<int>(std::floor(ax)), y0 = static_cast<int>(std::floor(ay));
                    int px0 = std::clamp(x0, 0, W - 1), px1 = std::clamp(x0 + 1, 0, W - 1);
                    int py0 = std::clamp(y0, 0, H - 1), py1 = std::clamp(y0 + 1, 0, H - 1);
                    D3 v00 = texel(px0, py0), v10 = texel(px1, py0), v01 = texel(px0, py1), v11 = texel(px1, py1);
                    double wx = ax - x0, wy = ay - y0;
                    D3 a{v00.x * (1.0 - wx) + v10.x * wx, v00.y * (1.0 - wx) + v10.y * wx, v00.z * (1.0 - wx) + v10.z * wx};
                    D3 b{v01.x * (1.0 - wx) + v11.x * wx, v01.y * (1.0 - wx) + v11.y * wx, v01.z * (1.0 - wx) + v11.z * wx};
                    return {a.x * (1.0 - wy) + b.x * wy, a.y * (1.0 - wy) + b.y * wy, a.z * (1.0 - wy) + b.z * wy};
                }
                int px = std::min(static_cast<int>(u * W), W - 1), py = std::min(static_cast<int>(v * H), H - 1);
                return texel(px, py);
            }
            // image_texture_2d_rgb::integrate over [a, b], image_texture.hpp:35-78 (midpoint rule per texel)
            D3 integrate(double au, double av, double bu, double bv, int samples) const
            {
                const int W = im->width, H = im->height;
                int p0x = static_cast<int>(std::floor(au * W)), p0y = static_cast<int>(std::floor(av * H));
                int p1x = static_cast<int>(std::ceil(bu * W)), p1y = static_cast<int>(std::ceil(bv * H));
                D3 value{0, 0, 0};
                for(int i = p0y; i < p1y; ++i)
                    for(int j = p0x; j < p1x; ++j)
                    {
                        double afx = std::max(au, static_cast<double>(j) / W), afy = std::max(av, static_cast<double>(i) / H);
                        double bfx = std::min(bu, static_cast<double>(j + 1) / W), bfy = std::min(bv, static_cast<double>(i + 1) / H);
                        double delta_u = (bfx - afx) / static_cast<double>(samples), delta_v = (bfy - afy) / static_cast<double>(samples);
                        double area = delta_u * delta_v;
                        D3 acc{0, 0, 0};
                        for(int ii = 0; ii < samples; ++ii)
                        {
                            double v = afy + (ii + 0.5) * delta_v;
                            for(int jj = 0; jj < samples; ++jj)
                            {
                                double u = afx + (jj + 0.5) * delta_u;
                                D3 e = evaluate(u, v);
                                acc.x += e.x; acc.y += e.y; acc.z += e.z;
                            }
                        }
                        value.x += acc.x * area; value.y += acc.y * area; value.z += acc.z * area;
                    }
                return value;
            }
        };

        // distribution_1d's constructor, distribution.hpp:31-59: returns the function integral; `f` may be rewritten
        double build_dist1d(std::vector<double>& f, double* cdf)
        {
            const size_t n = f.size();
            cdf[0] = 0.0;
            for(size_t i = 1; i <= n; ++i) cdf[i] = cdf[i - 1] + f[i - 1];
            double integral = cdf[n] / static_cast<double>(n);
            if(integral != 0.0)
            {
                for(size_t i = 1; i <= n; ++i) cdf[i] /= static_cast<double>(n) * integral;
            }
            else
            {
                integral = 1.0;
                for(size_t i = 1; i <= n; ++i)
                {
                    f[i - 1] = 1.0;
                    cdf[i] = static_cast<double>(i) / static_cast<double>(n);
                }
            }
            return integral;
        }

        // texture_infinity_area_light's constructor, texture_infinity_area_light.hpp:15-38
        int build_env_texture(const fcgpu_scene_desc& d, FlatScene& out, std::string& err)
        {
            const fcgpu_env_texture& t = d.env_texture;
            if(t.image < 0 || static_cast<uint32_t>(t.image) >= d.image_count)
            {
                err = "env_texture.image out of range";
                return FCGPU_E_INVALID;
            }
            if(t.distribution_width <= 0 || t.distribution_height <= 0 || t.distribution_width > 16384 || t.distribution_height > 16384 ||
               t.integral_sample_count <= 0 || (t.filter != FCGPU_FILTER_BOX && t.filter != FCGPU_FILTER_BILINEAR))
            {
                err = "env_texture: bad distribution resolution, integral_sample_count or filter";
                return FCGPU_E_INVALID;
            }
            FlatEnvTexture& e = out.env_texture;
            e.image = t.image;
            e.filter = t.filter;
            e.nx = t.distribution_width;
            e.ny = t.distribution_height;
            e.strength = t.strength;
            const double zero[3]{0.0, 0.0, 0.0};
            Xf xf = make_xf(zero, t.rotation, nullptr);
            for(int r = 0; r < 3; ++r)
                for(int c = 0; c < 4; ++c)
                {
                    e.fwd[4 * r + c] = xf.t.m[r][c];
                    e.inv[4 * r + c] = xf.inv.m[r][c];
                }
            const TexD tex{&out.images[t.image], t.filter};
            const double delta_u = static_cast<double>(e.nx), delta_v = static_cast<double>(e.ny);
            e.func_x.resize(static_cast<size_t>(e.nx) * e.ny);
            e.cdf_x.resize(static_cast<size_t>(e.nx + 1) * e.ny);
            e.int_x.resize(e.ny);
            e.cdf_y.resize(e.ny + 1);
            std::vector<double> row(e.nx);
            for(int i = 0; i < e.ny; ++i)
            {
                double sin_theta = std::sin(3.14159265358979323846 * (i + 0.5) / e.ny);
                for(int j = 0; j < e.nx; ++j)
                {
                    D3 integral = tex.integrate(j / delta_u, i / delta_v, (j + 1) / delta_u, (i + 1) / delta_v, t.integral_sample_count);
                    row[j] = (0.212671 * integral.x + 0.715160 * integral.y + 0.072169 * integral.z) * sin_theta;  // core/color.hpp:8-11
                }
                e.int_x[i] = build_dist1d(row, &e.cdf_x[static_cast<size_t>(i) * (e.nx + 1)]);
                std::copy(row.begin(), row.end(), e.func_x.begin() + static_cast<size_t>(i) * e.nx);
            }
            e.func_y = e.int_x;  // distribution_2d: the y function is the row integrals (distribution.hpp:134-140)
            e.int_y = build_dist1d(e.func_y, e.cdf_y.data());
            return FCGPU_OK;
        }
    }  // namespace

    // The lobes of one constant material exactly as a scene upload precomputes them (fcgpu_bsdf_table parity hook)
    int material_lobe_table(const fcgpu_material& m, std::vector<LobeRec<float>>& f32, std::vector<LobeRec<double>>& f64, std::string& err)
    {
        std::vector<LobeD> lobes;
        const int n = material_lobes(m, lobes, err);
        if(n < 0) return n;
        f32.clear();
        f64.clear();
        convert_lobes(lobes, f32);
        convert_lobes(lobes, f64);
        return n;
    }

    int flatten_scene(const fcgpu_scene_desc& d, FlatScene& out, std::string& err, const HierarchyBuilder* device_builder, bool reference_split_rule,
                      int bvh_width)
    {
        out = FlatScene{};
        if(d.entity_count == 0 || d.surface_count == 0 || d.material_count == 0)
        {
            err = "scene needs at least one entity, surface and material";
            return FCGPU_E_INVALID;
        }
        if(d.camera.width <= 0 || d.camera.height <= 0 || d.camera.width >= (1 << 20) || d.camera.height >= (1 << 20))
        {
            err = "camera resolution out of range";
            return FCGPU_E_INVALID;
        }
        if(d.medium_count > static_cast<uint32_t>(MAX_MEDIA))
        {
            err = "more than 15 media";
            return FCGPU_E_LIMIT;
        }
        out.width = d.camera.width;
        out.height = d.camera.height;

        // ---- materials
        // image textures of material parameters (image_texture_2d_rgb / _r, textures/image_texture.hpp)
        for(uint32_t i = 0; i < d.texture_count; ++i)
        {
            const fcgpu_texture& t = d.textures[i];
            if(t.kind == FCGPU_TEXTURE_CHECKER)
            {
                if(!(t.scale == t.scale))
                {
                    err = "texture " + std::to_string(i) + ": checker scale is NaN";
                    return FCGPU_E_INVALID;
                }
            }
            else if(t.kind != FCGPU_TEXTURE_IMAGE || t.image < 0 || static_cast<uint32_t>(t.image) >= d.image_count ||
                    (t.filter != FCGPU_FILTER_BOX && t.filter != FCGPU_FILTER_BILINEAR))
            {
                err = "texture " + std::to_string(i) + ": kind, image index or filter out of range";
                return FCGPU_E_INVALID;
            }
            out.textures.push_back(t);
        }
        std::vector<LobeD> lobes;
        for(uint32_t i = 0; i < d.material_count; ++i)
        {
            const fcgpu_material& m = d.materials[i];
            int first = static_cast<int>(lobes.size());
            int n = material_lobes(m, lobes, err);
            if(n < 0) return n;
            MaterialRec rec{};
            rec.first_lobe = first;
            rec.lobe_count = n;
            rec.kind = m.kind;
            // which parameters each material class has (its constructor's texture arguments, src/materials/*.hpp);
            // roughness is a texture_2d_rg everywhere but in standard_material and the reference has no rg image texture
            const bool std_mat = m.kind == FCGPU_MATERIAL_STANDARD;
            const bool allowed[6]{
                m.kind != FCGPU_MATERIAL_TRANSMISSION,                                                                   // TEX_COLOR_A
                m.kind == FCGPU_MATERIAL_GLASS || m.kind == FCGPU_MATERIAL_PLASTIC || m.kind == FCGPU_MATERIAL_TRANSMISSION,  // TEX_COLOR_B
                std_mat,                                                                                                 // TEX_ROUGHNESS
                std_mat,                                                                                                 // TEX_METALNESS
                std_mat || m.kind == FCGPU_MATERIAL_PLASTIC,                                                             // TEX_IOR
                std_mat || m.kind == FCGPU_MATERIAL_DIFFUSE || m.kind == FCGPU_MATERIAL_MIRROR,                          // TEX_NORMAL (normal_mapping<T>)
            };
            const int32_t handles[6]{m.tex_color_a, m.tex_color_b, m.tex_roughness, m.tex_metalness, m.tex_ior, m.tex_normal};
            for(int s = 0; s < 6; ++s)
            {
                if(handles[s] == 0) continue;
                if(handles[s] < 0 || static_cast<uint32_t>(handles[s]) > d.texture_count || !allowed[s])
                {
                    err = "material " + std::to_string(i) + ": texture slot " + std::to_string(s) +
                          (allowed[s] ? " names no texture" : " is not a texture parameter of this material class");
                    return FCGPU_E_INVALID;
                }
                if(d.textures[handles[s] - 1].kind == FCGPU_TEXTURE_CHECKER && (s == TEX_ROUGHNESS || s == TEX_METALNESS || s == TEX_IOR))
                {
                    err = "material " + std::to_string(i) + ": the checker texture is an rgb texture (checker_texture_2d_rgb), slot " + std::to_string(s) + " takes a texture_2d_r";
                    return FCGPU_E_INVALID;
                }
                rec.tex[s] = handles[s];
                rec.textured = 1;
            }
            if(rec.textured)
            {
                // lobe types are decided per hit: smooth vs rough by the roughness texel, the standard material's
                // lobe set by its metalness texel
                const bool rough_tex = rec.tex[TEX_ROUGHNESS] != 0;
                const bool smooth_const = m.roughness[0] == 0.0 && (std_mat || m.roughness[1] == 0.0);
                auto reflective = [&]() { out.lobe_mask |= (rough_tex || !smooth_const ? 1u << L_GGX_REFL : 0u) | (rough_tex || smooth_const ? 1u << L_SPEC_REFL : 0u); };
                if(std_mat)
                {
                    out.lobe_mask |= 1u << L_LAMBERT;
                    reflective();
                }
            }
            out.materials.push_back(rec);
            const MatValues<double> v = material_constants(m);
            MatParam<float> vf{};
            for(int k = 0; k < 3; ++k)
            {
                vf.color_a[k] = static_cast<float>(v.color_a[k]);
                vf.color_b[k] = static_cast<float>(v.color_b[k]);
            }
            vf.roughness[0] = static_cast<float>(v.roughness[0]);
            vf.roughness[1] = static_cast<float>(v.roughness[1]);
            vf.ior = static_cast<float>(v.ior);
            vf.metalness = static_cast<float>(v.metalness);
            out.f32.matparams.push_back(vf);
            out.f64.matparams.push_back(v);
        }
        for(const LobeD& l : lobes) out.lobe_mask |= 1u << l.type;
        convert_lobes(lobes, out.f32.lobes);
        convert_lobes(lobes, out.f64.lobes);

        // ---- media (index 0 = the helper's dummy vacuum, integrator.hpp:115)
        auto push_medium = [&](int prio, double ior, double sx, double sy, double sz)
        {
            MediumRec<float> a{prio, 0, (float)ior, {(float)sx, (float)sy, (float)sz}};
            MediumRec<double> b{prio, 0, ior, {sx, sy, sz}};
            out.f32.media.push_back(a);
            out.f64.media.push_back(b);
        };
        push_medium(-1, 1.0, 0.0, 0.0, 0.0);
        for(uint32_t i = 0; i < d.medium_count; ++i)
        {
            const fcgpu_medium& m = d.media[i];
            // -color * density folded; the sign is applied on the device (medium.hpp:49)
            push_medium(m.priority, m.ior, m.color[0] * m.density, m.color[1] * m.density, m.color[2] * m.density);
        }

        // ---- surfaces -> primitives (entity order, scene.hpp:49-55)
        std::vector<XPrimD> xprims;
        std::vector<LightD> lights;
        std::vector<PrimRef> refs;
        std::vector<uint32_t> kind_index;  // per global primitive
        std::vector<TriRec> tris_orig;     // per triangle, original order; re-ordered after the build
        std::vector<TriAttr> attrs;

        for(uint32_t ei = 0; ei < d.entity_count; ++ei)
        {
            const fcgpu_entity& e = d.entities[ei];
            if(e.surface < 0 || static_cast<uint32_t>(e.surface) >= d.surface_count || e.material < 0 ||
               static_cast<uint32_t>(e.material) >= d.material_count || e.medium >= static_cast<int32_t>(d.medium_count))
            {
                err = "entity references out of range";
                return FCGPU_E_INVALID;
            }
            const fcgpu_surface& s = d.surfaces[e.surface];
            EntityRec er{};
            er.material = e.material;
            er.medium = e.medium < 0 ? 0 : e.medium + 1;
            er.light = -1;
            er.flags = (s.kind == FCGPU_SURFACE_MESH && s.has_normals ? 1u : 0u) | (s.kind == FCGPU_SURFACE_MESH && s.has_uvs ? 2u : 0u);

            // TriRec::entity carries, above ENTITY_MASK, the queue key of a hit when nothing else decides it: no medium
            // boundary, one constant lobe (k_extend then skips the entity -> material -> lobe loads)
            if(ei > ENTITY_MASK)
            {
                err = "more than 2^24 entities";
                return FCGPU_E_LIMIT;
            }
            uint32_t key_hint = KEY_HINT_NONE;
            {
                const MaterialRec& m = out.materials[static_cast<size_t>(e.material)];
                if(er.medium == 0 && !m.textured && m.lobe_count == 1) key_hint = static_cast<uint32_t>(lobes[static_cast<size_t>(m.first_lobe)].type);
            }

            uint32_t first_prim = static_cast<uint32_t>(kind_index.size());
            LightD light{};
            light.entity = static_cast<int32_t>(ei);
            light.first_prim = first_prim;
            for(int k = 0; k < 3; ++k) light.Le[k] = e.emission_color[k] * e.emission_strength;

            if(s.kind == FCGPU_SURFACE_SPHERE || s.kind == FCGPU_SURFACE_PLANE)
            {
                if(s.scale[0] != 1.0 || s.scale[1] != 1.0 || s.scale[2] != 1.0)
                {
                    err = "sphere / plane surfaces take a pr_transform (no scale)";
                    return FCGPU_E_INVALID;
                }
                XPrimD x{};
                x.xf = make_xf(s.position, s.rotation, nullptr);
                x.prim = first_prim;
                x.entity = ei;
                BoxF box;
                if(s.kind == FCGPU_SURFACE_SPHERE)
                {
                    x.a = s.radius;
                    x.inv_area = 1.0 / (4.0 * 3.141592653589793 * s.radius * s.radius);  // sphere_surface.hpp:31-34,170
                    box = transformed_box(x.xf.t, {-s.radius, -s.radius, -s.radius}, {s.radius, s.radius, s.radius});
                    light.kind = KIND_SPHERE;
                }
                else
                {
                    x.a = s.size[0];
                    x.b = s.size[1];
                    x.inv_area = 1.0 / (s.size[0] * s.size[1]);  // plane_surface.hpp:30-33,129
                    box = transformed_box(x.xf.t, {-s.size[0] / 2.0, 0.0, -s.size[1] / 2.0}, {s.size[0] / 2.0, 0.0, s.size[1] / 2.0});
                    light.kind = KIND_PLANE;
                }
                uint32_t xi = static_cast<uint32_t>(xprims.size());
                light.index = xi;
                light.inv_area = x.inv_area;
                xprims.push_back(x);
                uint32_t kind = s.kind == FCGPU_SURFACE_SPHERE ? KIND_SPHERE : KIND_PLANE;
                kind_index.push_back((kind << 30) | xi);
                PrimRef r{};
                r.index = first_prim;
                r.box = box;
                for(int k = 0; k < 3; ++k) r.centroid[k] = box.lo[k] * 0.5f + box.hi[k] * 0.5f;  // math.hpp:580-583
                refs.push_back(r);
                out.prim_info.push_back(PrimInfo{(kind << 30) | xi, ei});
            }
            else if(s.kind == FCGPU_SURFACE_MESH)
            {
                if(s.index_count % 3 != 0 || s.index_count == 0 || s.vertex_count == 0 ||
                   static_cast<uint64_t>(s.vertex_offset) + s.vertex_count > d.vertex_total ||
                   static_cast<uint64_t>(s.index_offset) + s.index_count > d.index_total || d.positions == nullptr || d.indices == nullptr)
                {
                    err = "mesh ranges out of bounds";
                    return FCGPU_E_INVALID;
                }
                Xf xf = make_xf(s.position, s.rotation, s.scale);
                // mesh_surface.hpp:23-38: transform in double, store float
                std::vector<float> wp(3 * static_cast<size_t>(s.vertex_count));
                std::vector<float> wn;
                for(uint32_t v = 0; v < s.vertex_count; ++v)
                {
                    const float* p = d.positions + 3 * static_cast<size_t>(s.vertex_offset + v);
                    D3 q = xpoint(xf.t, {p[0], p[1], p[2]});
                    wp[3 * v + 0] = static_cast<float>(q.x);
                    wp[3 * v + 1] = static_cast<float>(q.y);
                    wp[3 * v + 2] = static_cast<float>(q.z);
                }
                if(s.has_normals)
                {
                    if(d.normals == nullptr)
                    {
                        err = "mesh has_normals but no normals pool";
                        return FCGPU_E_INVALID;
                    }
                    wn.resize(wp.size());
                    for(uint32_t v = 0; v < s.vertex_count; ++v)
                    {
                        const float* p = d.normals + 3 * static_cast<size_t>(s.vertex_offset + v);
                        D3 q = xnormal(xf.inv, {p[0], p[1], p[2]});
                        wn[3 * v + 0] = static_cast<float>(q.x);
                        wn[3 * v + 1] = static_cast<float>(q.y);
                        wn[3 * v + 2] = static_cast<float>(q.z);
                    }
                }
                if(s.has_uvs && d.uvs == nullptr)
                {
                    err = "mesh has_uvs but no uv pool";
                    return FCGPU_E_INVALID;
                }
                uint32_t ntri = s.index_count / 3;
                double area_total = 0.0;
                std::vector<double> areas(ntri);
                for(uint32_t t = 0; t < ntri; ++t)
                {
                    const uint32_t* ix = d.indices + s.index_offset + 3 * static_cast<size_t>(t);
                    if(ix[0] >= s.vertex_count || ix[1] >= s.vertex_count || ix[2] >= s.vertex_count)
                    {
                        err = "mesh index out of range";
                        return FCGPU_E_INVALID;
                    }
                    TriRec tr{};
                    const float* v[3]{&wp[3 * ix[0]], &wp[3 * ix[1]], &wp[3 * ix[2]]};
                    std::memcpy(tr.v0, v[0], 12);
                    std::memcpy(tr.v1, v[1], 12);
                    std::memcpy(tr.v2, v[2], 12);
                    uint32_t prim = first_prim + t;
                    tr.prim = prim;
                    tr.entity = ei | (key_hint << 24);
                    tr.attr = INVALID_U32;
                    if(s.has_normals || s.has_uvs)
                    {
                        TriAttr a{};
                        // defaults of mesh_surface.hpp:348-351 when the mesh has no uvs
                        a.uv0[0] = 0; a.uv0[1] = 0; a.uv1[0] = 1; a.uv1[1] = 0; a.uv2[0] = 1; a.uv2[1] = 1;
                        if(s.has_normals)
                        {
                            std::memcpy(a.n0, &wn[3 * ix[0]], 12);
                            std::memcpy(a.n1, &wn[3 * ix[1]], 12);
                            std::memcpy(a.n2, &wn[3 * ix[2]], 12);
                        }
                        if(s.has_uvs)
                        {
                            const float* u = d.uvs + 2 * static_cast<size_t>(s.vertex_offset);
                            std::memcpy(a.uv0, u + 2 * ix[0], 8);
                            std::memcpy(a.uv1, u + 2 * ix[1], 8);
                            std::memcpy(a.uv2, u + 2 * ix[2], 8);
                        }
                        tr.attr = static_cast<uint32_t>(attrs.size());
                        attrs.push_back(a);
                    }
                    uint32_t ti = static_cast<uint32_t>(tris_orig.size());
                    tris_orig.push_back(tr);
                    kind_index.push_back((KIND_TRI << 30) | ti);
                    out.prim_info.push_back(PrimInfo{(KIND_TRI << 30) | ti, ei});

                    PrimRef r{};
                    r.index = prim;
                    r.box.grow(v[0]);
                    r.box.grow(v[1]);
                    r.box.grow(v[2]);
                    for(int k = 0; k < 3; ++k) r.centroid[k] = r.box.lo[k] * 0.5f + r.box.hi[k] * 0.5f;
                    refs.push_back(r);

                    // mesh_surface::get_area(i), :72-76 (double on widened float vertices)
                    double ax = (double)v[1][0] - v[0][0], ay = (double)v[1][1] - v[0][1], az = (double)v[1][2] - v[0][2];
                    double bx = (double)v[2][0] - v[0][0], by = (double)v[2][1] - v[0][1], bz = (double)v[2][2] - v[0][2];
                    double cx = ay * bz - az * by, cy = az * bx - ax * bz, cz = ax * by - ay * bx;
                    areas[t] = 0.5 * std::sqrt(cx * cx + cy * cy + cz * cz);
                    area_total += areas[t];
                }
                light.kind = KIND_TRI;
                light.tri_count = ntri;
                light.inv_area = 1.0 / area_total;
                if(e.emissive)
                {
                    // distribution_1d, distribution.hpp:32-60
                    light.cdf_offset = static_cast<uint32_t>(out.light_cdf.size());
                    std::vector<double> cdf(ntri + 1, 0.0);
                    for(uint32_t t = 1; t <= ntri; ++t) cdf[t] = cdf[t - 1] + areas[t - 1];
                    double integral = cdf[ntri] / static_cast<double>(ntri);
                    if(integral != 0.0)
                    {
                        for(uint32_t t = 1; t <= ntri; ++t) cdf[t] /= static_cast<double>(ntri) * integral;
                    }
                    else
                    {
                        for(uint32_t t = 1; t <= ntri; ++t) cdf[t] = static_cast<double>(t) / static_cast<double>(ntri);
                    }
                    out.light_cdf.insert(out.light_cdf.end(), cdf.begin(), cdf.end());
                }
            }
            else
            {
                err = "unknown surface kind";
                return FCGPU_E_INVALID;
            }

            if(e.emissive)
            {
                er.light = static_cast<int32_t>(lights.size());
                lights.push_back(light);
            }
            out.entities.push_back(er);
        }
        if(kind_index.size() >= (1u << 26))
        {
            err = "too many primitives (limit 2^26)";
            return FCGPU_E_LIMIT;
        }

        if(d.image_count > 0 && d.images == nullptr)
        {
            err = "image_count > 0 with null images";
            return FCGPU_E_INVALID;
        }
        out.images.resize(d.image_count);
        for(uint32_t k = 0; k < d.image_count; ++k)
            if(!decode_image(d.images[k], out.images[k], err)) return FCGPU_E_INVALID;
        if(d.env_kind != FCGPU_ENV_NONE && d.env_kind != FCGPU_ENV_CONST && d.env_kind != FCGPU_ENV_TEXTURE)
        {
            err = "unknown env_kind";
            return FCGPU_E_INVALID;
        }
        if(d.env_kind == FCGPU_ENV_TEXTURE)
        {
            int rc = build_env_texture(d, out, err);
            if(rc) return rc;
        }
        out.light_count = static_cast<uint32_t>(lights.size()) + (d.env_kind != FCGPU_ENV_NONE ? 1u : 0u);
        out.env_kind = d.env_kind;
        for(int k = 0; k < 3; ++k)
        {
            out.f32.env[k] = static_cast<float>(d.env_color[k]);
            out.f64.env[k] = d.env_color[k];
        }
        if(out.light_cdf.empty()) out.light_cdf.push_back(0.0);

        // ---- BVH: binary build (host SAH, either split rule, or the device's hierarchy), then the wide collapse
        BinaryBvh bin;
        if(device_builder && refs.size() >= 2)
        {
            int rc = build_on_device(*device_builder, refs, bin, err);
            if(rc) return rc;
        }
        else
        {
            BinaryBuilder bb{refs, bin, reference_split_rule};
            bb.run();
        }
        out.binary_nodes = static_cast<uint32_t>(bin.nodes.size());

        // triangles re-ordered to leaf order so that leaf ranges read consecutive records
        std::vector<uint32_t> new_tri_slot(tris_orig.size(), INVALID_U32);
        for(uint32_t prim : bin.order)
        {
            uint32_t ki = kind_index[prim];
            if((ki >> 30) == KIND_TRI)
            {
                uint32_t old = ki & 0x3FFFFFFFu;
                new_tri_slot[old] = static_cast<uint32_t>(out.tris.size());
                out.tris.push_back(tris_orig[old]);
            }
        }
        for(size_t p = 0; p < kind_index.size(); ++p)
        {
            if((kind_index[p] >> 30) == KIND_TRI)
            {
                uint32_t slot = new_tri_slot[kind_index[p] & 0x3FFFFFFFu];
                kind_index[p] = (KIND_TRI << 30) | slot;
                out.prim_info[p].kind_index = kind_index[p];
            }
        }
        out.tri_attrs = std::move(attrs);
        if(out.tris.empty()) out.tris.push_back(TriRec{});
        make_tri_shade(out.tris, out.tri_attrs, out.f32.tri_shade);
        make_tri_shade(out.tris, out.tri_attrs, out.f64.tri_shade);
        if(out.tri_attrs.empty()) out.tri_attrs.push_back(TriAttr{});

        std::vector<BoxF> order_box(bin.order.size());
        std::vector<uint32_t> order_ki(bin.order.size());
        for(size_t k = 0; k < bin.order.size(); ++k)
        {
            order_box[k] = refs[k].box;  // the build leaves refs in leaf order
            order_ki[k] = kind_index[bin.order[k]];
        }
        if(bvh_width != 4 && bvh_width != 8)
        {
            err = "bvh_width must be 4 or 8";
            return FCGPU_E_INVALID;
        }
        WideBuilder wb{bin, order_box, order_ki, out, bvh_width};
        wb.run();
        if(!wb.error.empty())
        {
            err = wb.error;
            return FCGPU_E_LIMIT;
        }
        if(bvh_width == 8)
        {
            // the triangle records move once more, into the order the 8-wide nodes reference them (each node's leaf
            // children consecutive), with a proxy record per sphere / plane leaf; everything keyed by triangle index follows
            std::vector<TriRec> moved(wb.tri_order8.size());
            std::vector<uint32_t> new_slot(out.tris.size(), INVALID_U32);
            for(size_t k = 0; k < wb.tri_order8.size(); ++k)
            {
                const uint32_t w = wb.tri_order8[k];
                if(w & WideBuilder::XPRIM_PROXY_TAG)
                {
                    TriRec proxy{};
                    proxy.prim = w & 0x00FFFFFFu;
                    proxy.entity = (w >> 24) & 0x3u;
                    proxy.attr = XPRIM_PROXY;
                    moved[k] = proxy;
                }
                else
                {
                    moved[k] = out.tris[w];
                    new_slot[w] = static_cast<uint32_t>(k);
                }
            }
            for(size_t p = 0; p < kind_index.size(); ++p)
            {
                if((kind_index[p] >> 30) == KIND_TRI)
                {
                    kind_index[p] = (KIND_TRI << 30) | new_slot[kind_index[p] & 0x3FFFFFFFu];
                    out.prim_info[p].kind_index = kind_index[p];
                }
            }
            out.tris.swap(moved);
            if(out.tris.empty()) out.tris.push_back(TriRec{});
            make_tri_shade(out.tris, out.tri_attrs, out.f32.tri_shade);
            make_tri_shade(out.tris, out.tri_attrs, out.f64.tri_shade);
            out.nodes.assign(1, Node4{});  // unused: the 4-wide array stays a valid (empty-rooted) allocation
            for(uint32_t& c : out.nodes[0].child) c = INVALID_U32;
        }
#if FCG_TREELET
        treelet_first(out.nodes, FCG_TREELET);
#endif
        if(out.max_stack > static_cast<uint32_t>(STACK_CAP + STACK_SPILL_MAX))
        {
            err = "BVH needs a deeper traversal stack than STACK_CAP + STACK_SPILL_MAX";
            return FCGPU_E_LIMIT;
        }

        convert_xprims(xprims, out.f32.xprims);
        convert_xprims(xprims, out.f64.xprims);
        if(out.f32.xprims.empty())
        {
            out.f32.xprims.emplace_back();
            out.f64.xprims.emplace_back();
        }
        convert_lights(lights, out.f32.lights);
        convert_lights(lights, out.f64.lights);
        if(out.f32.lights.empty())
        {
            out.f32.lights.emplace_back();
            out.f64.lights.emplace_back();
        }
        convert_camera(d.camera, out.f32.camera);
        convert_camera(d.camera, out.f64.camera);
        return FCGPU_OK;
    }
}
