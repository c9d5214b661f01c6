// fc_material.h -- material::evaluate of the six material classes (src/materials/*.hpp) followed
// by the bsdf constructor (src/core/bsdf.hpp:13-44): parameter values -> lobe list. Shared by the
// host (constant materials: lobes precomputed at upload, in double) and the device (materials
// with image textures: lobes built per hit from the texel values).
#pragma once
#include "../../include/fcgpu.h"
#include "fc_types.h"

namespace fcg
{
    constexpr int MAX_LOBES = 3;  // standard_material emits at most three (standard_material.hpp:31-34)

    template<typename R> using MatValues = MatParam<R>;  // a material's parameters at one uv: texels where textured, else its constants

    template<typename R> FC_HD R rough_to_alpha(R r)  // smith_ggx_microfacet_model, microfacet.hpp:76-81
    {
        R x = r > R(0.002) ? r : R(0.002);
        return x * x;
    }

    // Which lobes a material has and how they are picked (types, pick cdf): all k_extend needs.
    // Returns the lobe count; weight / scale follow each material's evaluate().
    template<typename R>
    FC_HD int material_layout(int kind, const MatValues<R>& v, int* types, double* scales, double* weights)
    {
        const bool smooth = v.roughness[0] == R(0) && v.roughness[1] == R(0);
        int n = 0;
        auto add = [&](int type, double scale, double weight) {
            types[n] = type;
            scales[n] = scale;
            weights[n] = weight;
            n += 1;
        };
        switch(kind)
        {
        case FCGPU_MATERIAL_DIFFUSE: add(L_LAMBERT, 1.0, 1.0); break;                              // diffuse_material.hpp:29-37
        case FCGPU_MATERIAL_MIRROR: add(smooth ? L_SPEC_REFL : L_GGX_REFL, 1.0, 1.0); break;       // mirror_material.hpp:38-55
        case FCGPU_MATERIAL_GLASS: add(smooth ? L_SPEC_GLASS : L_GGX_GLASS, 1.0, 1.0); break;      // glass_material.hpp:29-40
        case FCGPU_MATERIAL_PLASTIC:                                                               // plastic_material.hpp:26-51
            add(L_LAMBERT, 1.0, 1.0);
            add(smooth ? L_SPEC_REFL : L_GGX_REFL, 1.0, 1.0);
            break;
        case FCGPU_MATERIAL_STANDARD:                                                              // standard_material.hpp:49-103
        {
            const bool sm = v.roughness[0] == R(0);
            const double metal = static_cast<double>(v.metalness);
            if(metal < 1.0)
            {
                add(L_LAMBERT, 1.0 - metal, (1.0 - metal) / 2.0);
                add(sm ? L_SPEC_REFL : L_GGX_REFL, 1.0 - metal, (1.0 - metal) / 2.0);
            }
            if(metal > 0.0) add(sm ? L_SPEC_REFL : L_GGX_REFL, metal, metal);
            break;
        }
        case FCGPU_MATERIAL_TRANSMISSION: add(smooth ? L_SPEC_TRANS : L_GGX_TRANS, 1.0, 1.0); break;  // transmission_material.hpp:28-39
        default: break;
        }
        return n;
    }

    // bsdf constructor, bsdf.hpp:30-43: pick pdfs from the weights, exclusive prefix as the pick cdf
    // (accumulated left to right as the reference does), factor = scale / pdf
    FC_HD void lobe_pick_table(int n, const double* scales, const double* weights, double* factor, double* cdf)
    {
        double weight_sum = 0.0;
        for(int i = 0; i < n; ++i) weight_sum += weights[i];
        double prefix = 0.0;
        for(int i = 0; i < n; ++i)
        {
            const double pdf = weights[i] / weight_sum;
            factor[i] = scales[i] / pdf;
            cdf[i] = i == 0 ? 0.0 : prefix;
            prefix = (i == 0 ? 0.0 : prefix) + pdf;
        }
    }

    // lobe `index` of the material: parameters of the bxdf objects each evaluate() constructs
    template<typename R>
    FC_HD LobeRec<R> material_lobe(int kind, const MatValues<R>& v, int index, int type, double factor, double cdf)
    {
        LobeRec<R> l{};
        l.type = type;
        l.factor = static_cast<R>(factor);
        l.cdf = cdf;
        auto set = [&](int fresnel, const R* c, const R* c2, R rx, R ry, R ior) {
            l.fresnel = fresnel;
            for(int i = 0; i < 3; ++i)
            {
                l.color[i] = c ? c[i] : R(0);
                l.color2[i] = c2 ? c2[i] : R(0);
            }
            l.alpha[0] = rough_to_alpha(rx);
            l.alpha[1] = rough_to_alpha(ry);
            l.ior = ior;
        };
        const R one[3]{R(1), R(1), R(1)};
        switch(kind)
        {
        case FCGPU_MATERIAL_DIFFUSE: set(0, v.color_a, nullptr, R(0), R(0), R(0)); break;
        case FCGPU_MATERIAL_MIRROR: set(0, v.color_a, nullptr, v.roughness[0], v.roughness[1], R(0)); break;
        case FCGPU_MATERIAL_GLASS: set(0, v.color_a, v.color_b, v.roughness[0], v.roughness[1], R(0)); break;
        case FCGPU_MATERIAL_PLASTIC:
            if(index == 0)
                set(0, v.color_a, nullptr, R(0), R(0), R(0));
            else
                set(1, v.color_b, nullptr, v.roughness[0], v.roughness[1], v.ior);  // fresnel_dielectric with the material's ior
            break;
        case FCGPU_MATERIAL_STANDARD:
        {
            const bool dielectric_part = static_cast<double>(v.metalness) < 1.0;
            if(dielectric_part && index == 0)
                set(0, v.color_a, nullptr, R(0), R(0), R(0));
            else if(dielectric_part && index == 1)
                set(1, one, nullptr, v.roughness[0], v.roughness[0], v.ior);
            else
                set(0, v.color_a, nullptr, v.roughness[0], v.roughness[0], R(0));  // fresnel_one, ior 0
            break;
        }
        case FCGPU_MATERIAL_TRANSMISSION: set(0, v.color_b, nullptr, v.roughness[0], v.roughness[1], R(0)); break;
        default: break;
        }
        return l;
    }
}
