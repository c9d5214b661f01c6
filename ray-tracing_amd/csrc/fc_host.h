// fc_host.h -- host-side scene flattening: POD scene description -> device-ready arrays.
#pragma once
#include <cstdint>
#include <functional>
#include <string>
#include <utility>
#include <vector>

#include "../../include/fcgpu.h"
#include "fc_types.h"

namespace fcg
{
    struct FlatImage  // decoded as image::rgb() returns it
    {
        int32_t width{}, height{};
        std::vector<double> rgb;  // 3 per texel, row-major, top row first
    };

    struct FlatEnvTexture  // texture_infinity_area_light: texture parameters + radiance distribution
    {
        int32_t image{-1}, filter{}, nx{}, ny{};
        double strength{};
        double fwd[12]{}, inv[12]{};
        std::vector<double> func_x, cdf_x, int_x, func_y, cdf_y;
        double int_y{};
    };

    template<typename R>
    struct FlatTyped  // arrays whose element type depends on the arithmetic type
    {
        std::vector<XPrim<R>> xprims;
        std::vector<TriShade<R>> tri_shade;  // by triangle index, same order as FlatScene::tris
        std::vector<LobeRec<R>> lobes;
        std::vector<MatParam<R>> matparams;
        std::vector<MediumRec<R>> media;
        std::vector<LightRec<R>> lights;
        CameraRec<R> camera{};
        R env[3]{};
    };

    struct FlatScene
    {
        std::vector<Node4> nodes;
        std::vector<Node8> nodes8;  // instead of `nodes` when the 8-wide hierarchy was asked for (bvh_width 8)
        std::vector<TriRec> tris;
        std::vector<TriAttr> tri_attrs;
        std::vector<PrimInfo> prim_info;
        std::vector<EntityRec> entities;
        std::vector<MaterialRec> materials;
        std::vector<double> light_cdf;
        FlatTyped<float> f32;
        FlatTyped<double> f64;
        uint32_t light_count{};   // incl. the environment light when present (scene.hpp:66-70)
        int32_t env_kind{};
        std::vector<FlatImage> images;
        std::vector<fcgpu_texture> textures;  // validated copies of the description's
        FlatEnvTexture env_texture;
        uint32_t max_stack{};     // worst-case traversal stack depth of the wide BVH
        uint32_t binary_nodes{};
        uint32_t lobe_mask{};     // bit t set when some material can pick lobe type t
        int32_t width{}, height{};
    };

    // A binary hierarchy over n primitives as the device builder (fc_lbvh.hip) returns it: n - 1 internal nodes, root 0;
    // a child >= 0 is an internal node, < 0 the leaf ~k holding primitive order[k].
    struct LbvhTree
    {
        std::vector<uint32_t> order;
        std::vector<int32_t> left, right;
        std::vector<float> node_boxes;  // [n - 1][6] lo.xyz, hi.xyz
    };
    // boxes: [n][6] primitive bounds. Returns FCGPU_OK or an error code with a message in `err`.
    using HierarchyBuilder = std::function<int(const std::vector<float>& boxes, LbvhTree& out, std::string& err)>;

    // Returns FCGPU_OK or a negative error code with a message in `err`. `device_builder`: build the binary hierarchy
    // there instead of with the host's SAH builder (scenes of fewer than 2 primitives always use the host).
    // `reference_split_rule`: the host builder searches splits exactly as the reference does (FCGPU_BVH_HOST_REFERENCE).
    // `bvh_width`: 4 (Node4, the default) or 8 (Node8).
    int flatten_scene(const fcgpu_scene_desc& d, FlatScene& out, std::string& err, const HierarchyBuilder* device_builder = nullptr,
                      bool reference_split_rule = false, int bvh_width = 4);

    // Lobes of one constant material (material::evaluate + the bsdf constructor) as an upload precomputes them: double
    // arithmetic, converted per precision. Returns the lobe count or a negative error code.
    int material_lobe_table(const fcgpu_material& m, std::vector<LobeRec<float>>& f32, std::vector<LobeRec<double>>& f64, std::string& err);
}
