// fc_oracle.hpp -- TEST INFRASTRUCTURE ONLY. CPU restatement of the reference's Forward-MIS /
// Forward-BSDF hot path, one sample at a time, exactly in the order the reference executes it.
//
// It is the checker for the HIP path where the real reference (oracle/_ref) cannot go, and is
// itself pinned against the real reference: tests/test_oracle_vs_reference.py (when
// /root/reference is present) and the committed golden fixtures under tests/golden/ that
// oracle/make_golden.py generated from the real reference.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it; the product
// (ray-tracing_amd/) never includes, links or calls anything in this directory.
//
// Every function cites the reference lines it restates. Template parameter R is the arithmetic
// type: double reproduces the reference; float gives the numbers an fp32 device should produce.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <memory>
#include <vector>

#include "fcgpu.h"
#include "fcgpu_sampler.h"

namespace fco
{
    template<typename R>
    struct V3
    {
        R x{}, y{}, z{};
        R operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    };
    template<typename R> V3<R> operator+(V3<R> a, V3<R> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
    template<typename R> V3<R> operator-(V3<R> a, V3<R> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
    template<typename R> V3<R> operator-(V3<R> a) { return {-a.x, -a.y, -a.z}; }
    template<typename R> V3<R> operator*(V3<R> a, V3<R> b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
    template<typename R> V3<R> operator*(V3<R> a, R s) { return {a.x * s, a.y * s, a.z * s}; }
    template<typename R> V3<R> operator*(R s, V3<R> a) { return {a.x * s, a.y * s, a.z * s}; }
    template<typename R> V3<R> operator/(V3<R> a, R s) { return {a.x / s, a.y / s, a.z / s}; }
    template<typename R> R dot(V3<R> a, V3<R> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
    template<typename R> V3<R> cross(V3<R> a, V3<R> b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
    template<typename R> R length(V3<R> a) { return std::sqrt(dot(a, a)); }
    template<typename R> V3<R> normalize(V3<R> a) { return a / length(a); }
    template<typename R> bool nonzero(V3<R> a) { return a.x != R(0) || a.y != R(0) || a.z != R(0); }  // math.hpp:124-127

    // 4x4 matrices in double (scene set-up is host work in both precisions)
    struct Mat
    {
        double m[4][4]{};
    };

    template<typename R>
    struct Transform  // pr_transform / prs_transform, transform.hpp
    {
        R t[3][4]{}, inv[3][4]{};
        V3<R> point(V3<R> p) const { return apply(t, p, R(1)); }
        V3<R> dir(V3<R> d) const { return apply(t, d, R(0)); }
        V3<R> inv_point(V3<R> p) const { return apply(inv, p, R(1)); }
        V3<R> inv_dir(V3<R> d) const { return apply(inv, d, R(0)); }
        static V3<R> apply(const R (&m)[3][4], V3<R> v, R w)
        {
            if(w != R(0))
                return {m[0][0] * v.x + m[0][1] * v.y + m[0][2] * v.z + m[0][3], m[1][0] * v.x + m[1][1] * v.y + m[1][2] * v.z + m[1][3],
                        m[2][0] * v.x + m[2][1] * v.y + m[2][2] * v.z + m[2][3]};
            return {m[0][0] * v.x + m[0][1] * v.y + m[0][2] * v.z, m[1][0] * v.x + m[1][1] * v.y + m[1][2] * v.z,
                    m[2][0] * v.x + m[2][1] * v.y + m[2][2] * v.z};
        }
    };

    struct BoxF  // bounds3f
    {
        float lo[3]{std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
        float hi[3]{std::numeric_limits<float>::lowest(), std::numeric_limits<float>::lowest(), std::numeric_limits<float>::lowest()};
    };

    template<typename R>
    struct Point  // surface_point.hpp:48-62, pointers replaced by indices
    {
        V3<R> position{}, normal{}, tangent{}, shading_normal{}, bitangent{};
        R u{}, v{};
        int entity{-1};
        int surface{-1};
    };

    enum LobeKind { LAMBERT, GGX_REFLECTION, SPECULAR_REFLECTION, GGX_GLASS, SPECULAR_GLASS, GGX_TRANSMISSION, SPECULAR_TRANSMISSION };

    // normal_mapping<T> (bsdfs/normal_mapping.hpp): a microsurface of two facets, one with the mapped normal p and a
    // vertical "tangent" facet t that closes the profile and reflects specularly. All vectors live in the
    // shading frame (y up). off = the skip_ path (p within 0.001 of the shading normal, :745-749).
    template<typename R>
    struct NormalMap
    {
        bool on{};
        V3<R> p{}, t{}, tangent{}, bitangent{};
        V3<R> to_p(V3<R> w) const { return {dot(w, tangent), dot(w, p), dot(w, bitangent)}; }  // frame.hpp:22-25
        V3<R> from_p(V3<R> w) const                                                            // frame.hpp:27-34
        {
            return {tangent.x * w.x + p.x * w.y + bitangent.x * w.z, tangent.y * w.x + p.y * w.y + bitangent.y * w.z,
                    tangent.z * w.x + p.z * w.y + bitangent.z * w.z};
        }
        V3<R> mirror(V3<R> w) const { return w - R(2) * dot(w, t) * t; }  // reflection in the tangent facet
        // which facets a direction sees: 0 = only the tangent facet, 1 = both, 2 = only the mapped facet (:42-43)
        int facets(V3<R> w) const { return dot(w, p) <= R(0) ? 0 : dot(w, t) > R(0) ? 1 : 2; }
        R sine() const { return std::sqrt(R(1) - p.y * p.y); }
        R hits_p(V3<R> w) const  // probability that a ray along w meets the mapped facet first (:66-69)
        {
            R a_p = dot(w, p) / p.y, a_t = dot(w, t) * sine() / p.y;
            return a_p / (a_p + a_t);
        }
        R open_p(V3<R> w) const { return w.y * p.y / dot(w, p); }             // part of facet p not hidden behind t (gp_*)
        R open_t(V3<R> w) const { return w.y * p.y / (dot(w, t) * sine()); }  // same for the tangent facet (gt_*)
    };

    template<typename R>
    struct Lobe
    {
        NormalMap<R> nm{};
        LobeKind kind{};
        bool fresnel_dielectric{};
        V3<R> reflectance{}, transmittance{};
        R alpha_x{}, alpha_y{};
        R ior{};
        R scale{}, pdf{}, weight{};  // bsdf::bsdf_bxdf, bsdf.hpp:180-186
        bool is_delta() const { return kind == SPECULAR_REFLECTION || kind == SPECULAR_GLASS || kind == SPECULAR_TRANSMISSION; }
    };

    template<typename R>
    struct Bsdf
    {
        V3<R> t{}, ns{}, b{}, ng{};
        int count{};
        Lobe<R> lobes[4];
    };

    template<typename R>
    struct Medium
    {
        int priority{};
        R ior{};
        V3<R> color{};
        R density{};
        bool vacuum{};
    };

    struct RayRecord  // same layout as the reference harness' log
    {
        double o[3], d[3], t_max, t;
        uint32_t prim, kind;
    };

    struct Stats
    {
        uint64_t closest_rays, any_rays, any_occluded, nan_samples, samples;
        double seconds;
    };

    template<typename R> class Scene;
}
