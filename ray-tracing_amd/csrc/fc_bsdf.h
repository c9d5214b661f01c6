// fc_bsdf.h -- device-side BSDF evaluators (generic over R).
//
// Restates, for the GPU, the lobe classes of src/bsdfs/ and the two adapter layers above them:
//   lobe (one-sided, written for the adjoint direction)      src/bsdfs/*.hpp
//   bxdf_adapter<T> (side flip, eta swap, eta^2 scaling)      src/core/bxdf.hpp:57-127
//   bsdf (shading frame, ng/ns consistency, lobe factor)      src/core/bsdf.hpp:62-100,131-141
// Local shading space is y-up: local = (w.t, w.ns, w.b) (bsdf.hpp:156-172).
// normal_mapping<T> (normal_mapping.hpp:32-734) around the lambertian / specular / microfacet reflection lobes is the
// NM = true instantiation below (nm_combine); without a normal-map texture its `skip_` path (:34-35, 745-750) is the
// bare lobe.
#pragma once
#include "fc_types.h"

#ifndef FCG_FP32_FRESNEL
#define FCG_FP32_FRESNEL 1
#endif
#ifndef FCG_FP32_GLASS
#define FCG_FP32_GLASS 1
#endif

namespace fcg
{
    // ---------------------------------------------------------------- common.hpp
    template<typename R> FC_HD R fr_dielectric(R cos_i, R eta_i, R eta_t)  // common.hpp:8-20
    {
        cos_i = clampr(cos_i, R(-1), R(1));
        R cos_t;
        if constexpr(sizeof(R) == 4 && FCG_FP32_FRESNEL)
        {
            // fp32: cos_t^2 = 1 - eta^2 (1 - cos_i^2) regrouped as (1 - eta)(1 + eta) + eta^2 cos_i^2. The reference's
            // order (via sin_i, sin_t) cancels at grazing incidence between nearly equal indices: in fp32 it returned
            // F = 7e-4 instead of 0 at cos_i = 1e-3, eta = 1, which value = F / cos turns into 0.56 (tests/test_gpu_parity.py:
            // device bsdf tables). One square root less as well.
            const R eta = eta_i / eta_t;
            const R cos2_t = (R(1) - eta) * (R(1) + eta) + eta * eta * cos_i * cos_i;
            if(!(cos2_t > R(0))) return R(1);
            cos_t = r_sqrt(cos2_t);
        }
        else
        {
            R sin_i = r_sqrt(r_max(R(0), R(1) - cos_i * cos_i));
            R sin_t = eta_i / eta_t * sin_i;
            if(sin_t >= R(1)) return R(1);
            cos_t = r_sqrt(r_max(R(0), R(1) - sin_t * sin_t));
        }
        R r_parl = ((eta_t * cos_i) - (eta_i * cos_t)) / ((eta_t * cos_i) + (eta_i * cos_t));
        R r_perp = ((eta_i * cos_i) - (eta_t * cos_t)) / ((eta_i * cos_i) + (eta_t * cos_t));
        return (r_parl * r_parl + r_perp * r_perp) / R(2);
    }

    template<typename R> FC_HD vec3<R> reflect(vec3<R> w, vec3<R> n)  // common.hpp:22-25
    {
        return -w + (R(2) * dot(w, n)) * n;
    }

    template<typename R> FC_HD bool refract(vec3<R> i, vec3<R> m, R eta, vec3<R>& o)  // common.hpp:27-39
    {
        R cos_i = dot(i, m);
        R sin2_i = r_max(R(0), R(1) - cos_i * cos_i);
        R sin2_t = eta * eta * sin2_i;
        if(sin2_t >= R(1)) return false;
        R cos_t = r_sqrt(R(1) - sin2_t);
        o = eta * (-i) + (eta * cos_i - cos_t) * m;
        return true;
    }

    // ---------------------------------------------------------------- sampling.hpp
    template<typename R> FC_HD void sample_disk_concentric(R ux, R uy, R& dx, R& dy)  // sampling.hpp:6-29
    {
        R ox = ux * R(2) - R(1), oy = uy * R(2) - R(1);
        if(ox == R(0) && oy == R(0))
        {
            dx = R(0);
            dy = R(0);
            return;
        }
        R theta, r;
        if(r_abs(ox) > r_abs(oy))
        {
            r = ox;
            theta = consts<R>::pi * oy / (R(4) * ox);
        }
        else
        {
            r = oy;
            theta = consts<R>::pi / R(2) - consts<R>::pi * ox / (R(4) * oy);
        }
        dx = r * r_cos(theta);
        dy = r * r_sin(theta);
    }

    template<typename R> FC_HD vec3<R> sample_hemisphere_cosine(R ux, R uy)  // sampling.hpp:31-35
    {
        R dx, dy;
        sample_disk_concentric(ux, uy, dx, dy);
        return {dx, r_sqrt(r_max(R(0), R(1) - dx * dx - dy * dy)), dy};
    }

    template<typename R> FC_HD vec3<R> sample_sphere_uniform(R ux, R uy)  // sampling.hpp:49-55
    {
        R z = R(1) - R(2) * ux;
        R r = r_sqrt(r_max(R(0), R(1) - z * z));
        R phi = R(2) * consts<R>::pi * uy;
        return {r * r_cos(phi), z, r * r_sin(phi)};
    }

    // ---------------------------------------------------------------- microfacet.hpp (Smith GGX)
    template<typename R>
    struct Ggx
    {
        R ax, ay;

        FC_HD vec3<R> sample(vec3<R> i, R ux, R uy) const  // microfacet.hpp:28-44
        {
            vec3<R> ih = normalize(vec3<R>{ax * i.x, i.y, ay * i.z});
            R lensq = ih.x * ih.x + ih.z * ih.z;
            vec3<R> T1 = lensq > R(0) ? vec3<R>{-ih.z, R(0), ih.x} / r_sqrt(lensq) : vec3<R>{R(1), R(0), R(0)};
            vec3<R> T2 = cross(T1, ih);
            R r = r_sqrt(ux);
            R phi = R(2) * consts<R>::pi * uy;
            R t1 = r * r_cos(phi);
            R t2 = r * r_sin(phi);
            R s = R(0.5) * (R(1) + ih.y);
            t2 = (R(1) - s) * r_sqrt(R(1) - t1 * t1) + s * t2;
            vec3<R> Nh = t1 * T1 + t2 * T2 + r_sqrt(r_max(R(0), R(1) - t1 * t1 - t2 * t2)) * ih;
            return normalize(vec3<R>{ax * Nh.x, r_max(R(0), Nh.y), ay * Nh.z});
        }
        FC_HD R D(vec3<R> m) const  // :51-55
        {
            R x = m.x * m.x / (ax * ax) + m.y * m.y + m.z * m.z / (ay * ay);
            return R(1) / (consts<R>::pi * ax * ay * x * x);
        }
        FC_HD R lambda(vec3<R> i) const  // :70-74
        {
            R x = (ax * ax * i.x * i.x + ay * ay * i.z * i.z) / (i.y * i.y);
            return (R(-1) + r_sqrt(R(1) + x)) / R(2);
        }
        FC_HD R G1(vec3<R> i) const { return R(1) / (R(1) + lambda(i)); }                  // :57-60
        FC_HD R G2(vec3<R> i, vec3<R> o) const { return R(1) / (R(1) + lambda(i) + lambda(o)); }  // :62-65
        FC_HD R pdf(vec3<R> i, vec3<R> m) const { return G1(i) * r_max(R(0), dot(i, m)) * D(m) / i.y; }  // :46-49
    };

    template<typename R> FC_HD vec3<R> rgb(const R* c) { return {c[0], c[1], c[2]}; }

    template<typename R> FC_HD R lobe_fresnel(const LobeRec<R>& L, R cos_i, R eta_a)  // microfacet_reflection.hpp:18-35
    {
        return L.fresnel ? fr_dielectric(cos_i, eta_a, L.ior) : R(1);
    }

    // half vector of the transmission lobes, oriented as microfacet_glass.hpp:38-51; false = rejected
    template<typename R> FC_HD bool refraction_half(vec3<R> i, vec3<R> o, R eta_a, R eta_b, vec3<R>& h, R& i_dot_h, R& o_dot_h)
    {
        h = normalize(-(eta_a * i + eta_b * o));
        if(eta_a <= eta_b)
        {
            if(h.y <= R(0)) return false;
        }
        else
        {
            if(h.y >= R(0)) return false;
            h = -h;
        }
        i_dot_h = dot(i, h);
        o_dot_h = dot(o, h);
        return !(i_dot_h <= R(0) || o_dot_h >= R(0));
    }

    // ---------------------------------------------------------------- one-sided lobes
    // T::evaluate(i, o, eta_a, eta_b)
    template<int TYPE, typename R>
    FC_HD vec3<R> lobe_evaluate(const LobeRec<R>& L, vec3<R> i, vec3<R> o, R eta_a, R eta_b)
    {
        const vec3<R> zero{R(0), R(0), R(0)};
        if constexpr(TYPE == L_LAMBERT)  // lambertian_reflection.hpp:20-24
        {
            if(i.y <= R(0) || o.y <= R(0)) return zero;
            return rgb(L.color) * consts<R>::inv_pi;
        }
        else if constexpr(TYPE == L_GGX_REFL)  // microfacet_reflection.hpp:49-59
        {
            if(o.y <= R(0)) return zero;
            Ggx<R> g{L.alpha[0], L.alpha[1]};
            vec3<R> h = normalize(i + o);
            R G = g.G2(i, o);
            R D = g.D(h);
            R F = lobe_fresnel(L, dot(i, h), eta_a);
            return rgb(L.color) * (F * (G * D / (R(4) * i.y * o.y)));
        }
        else if constexpr(TYPE == L_GGX_GLASS)  // microfacet_glass.hpp:22-61
        {
            if(i.y == R(0) || o.y == R(0)) return zero;
            Ggx<R> g{L.alpha[0], L.alpha[1]};
            if(o.y > R(0))
            {
                vec3<R> h = normalize(i + o);
                R G = g.G2(i, o);
                R D = g.D(h);
                R F = fr_dielectric(dot(i, h), eta_a, eta_b);
                return rgb(L.color) * (G * D * F / (R(4) * i.y * o.y));
            }
            vec3<R> h;
            R idh, odh;
            if(!refraction_half(i, o, eta_a, eta_b, h, idh, odh)) return zero;
            R eta = eta_a / eta_b;
            R J = -odh / sqr(eta * idh + odh);
            R G = g.G2(i, o);
            R D = g.D(h);
            R F = fr_dielectric(idh, eta_a, eta_b);
            return rgb(L.color2) * (idh * G * D * J * (R(1) - F) / (i.y * -o.y));
        }
        else if constexpr(TYPE == L_GGX_TRANS)  // microfacet_transmission.hpp:21-46
        {
            if(o.y >= R(0)) return zero;
            Ggx<R> g{L.alpha[0], L.alpha[1]};
            vec3<R> h;
            R idh, odh;
            if(!refraction_half(i, o, eta_a, eta_b, h, idh, odh)) return zero;
            R eta = eta_a / eta_b;
            R J = -odh / sqr(eta * idh + odh);
            return rgb(L.color) * (idh * g.G2(i, o) * g.D(h) * J / (i.y * -o.y));
        }
        else  // delta lobes: specular_*.hpp evaluate == 0
        {
            return zero;
        }
    }

    // T::pdf(i, o, eta_a, eta_b)
    template<int TYPE, typename R>
    FC_HD R lobe_pdf(const LobeRec<R>& L, vec3<R> i, vec3<R> o, R eta_a, R eta_b)
    {
        if constexpr(TYPE == L_LAMBERT)  // lambertian_reflection.hpp:42-46
        {
            if(i.y <= R(0) || o.y <= R(0)) return R(0);
            return o.y * consts<R>::inv_pi;
        }
        else if constexpr(TYPE == L_GGX_REFL)  // microfacet_reflection.hpp:88-95
        {
            if(o.y <= R(0)) return R(0);
            Ggx<R> g{L.alpha[0], L.alpha[1]};
            vec3<R> h = normalize(i + o);
            return g.pdf(i, h) * (R(1) / (R(4) * dot(i, h)));
        }
        else if constexpr(TYPE == L_GGX_GLASS)  // microfacet_glass.hpp:123-160
        {
            if(i.y == R(0) || o.y == R(0)) return R(0);
            Ggx<R> g{L.alpha[0], L.alpha[1]};
            if(o.y > R(0))
            {
                vec3<R> h = normalize(i + o);
                R idh = dot(i, h);
                R F = fr_dielectric(idh, eta_a, eta_b);
                return g.pdf(i, h) * (R(1) / (R(4) * idh)) * F;
            }
            vec3<R> h;
            R idh, odh;
            if(!refraction_half(i, o, eta_a, eta_b, h, idh, odh)) return R(0);
            R eta = eta_a / eta_b;
            R F = fr_dielectric(idh, eta_a, eta_b);
            R J = -odh / sqr(eta * idh + odh);
            return g.pdf(i, h) * J * (R(1) - F);
        }
        else if constexpr(TYPE == L_GGX_TRANS)  // microfacet_transmission.hpp:85-108
        {
            if(o.y >= R(0)) return R(0);
            Ggx<R> g{L.alpha[0], L.alpha[1]};
            vec3<R> h;
            R idh, odh;
            if(!refraction_half(i, o, eta_a, eta_b, h, idh, odh)) return R(0);
            R eta = eta_a / eta_b;
            R J = -odh / sqr(eta * idh + odh);
            return g.pdf(i, h) * J;
        }
        else
        {
            return R(0);
        }
    }

    // T::sample(i, eta_a, eta_b, u1, u2, &o, &value, &pdf_o)
    template<int TYPE, typename R>
    FC_HD bool lobe_sample(const LobeRec<R>& L, vec3<R> i, R eta_a, R eta_b, R u1x, R u1y, R u2x, R u2y,
                           vec3<R>& o, vec3<R>& value, R& pdf_o)
    {
        (void)u2y;
        if constexpr(TYPE == L_LAMBERT)  // lambertian_reflection.hpp:26-40
        {
            if(i.y <= R(0)) return false;
            o = sample_hemisphere_cosine(u1x, u1y);
            if(o.y == R(0)) return false;
            value = rgb(L.color) * consts<R>::inv_pi;
            pdf_o = o.y * consts<R>::inv_pi;
            return true;
        }
        else if constexpr(TYPE == L_GGX_REFL)  // microfacet_reflection.hpp:61-86
        {
            if(i.y == R(0)) return false;
            Ggx<R> g{L.alpha[0], L.alpha[1]};
            vec3<R> h = g.sample(i, u1x, u1y);
            R idh = dot(i, h);
            if(idh <= R(0)) return false;
            o = reflect(i, h);
            if(o.y <= R(0)) return false;
            R G = g.G2(i, o);
            R D = g.D(h);
            R F = lobe_fresnel(L, idh, eta_a);
            value = rgb(L.color) * (F * (G * D / (R(4) * i.y * o.y)));
            pdf_o = g.pdf(i, h) * (R(1) / (R(4) * idh));
            return true;
        }
        else if constexpr(TYPE == L_SPEC_REFL)  // specular_reflection.hpp:25-40
        {
            if(i.y == R(0)) return false;
            R F = lobe_fresnel(L, i.y, eta_a);
            o = vec3<R>{-i.x, i.y, -i.z};
            value = rgb(L.color) * F / o.y;
            pdf_o = R(1);
            return true;
        }
        else if constexpr(TYPE == L_GGX_GLASS)  // microfacet_glass.hpp:63-121
        {
            if(i.y == R(0)) return false;
            Ggx<R> g{L.alpha[0], L.alpha[1]};
            vec3<R> h = g.sample(i, u1x, u1y);
            R idh = dot(i, h);
            if(idh <= R(0)) return false;
            R F = fr_dielectric(idh, eta_a, eta_b);
            if(u2x < F)
            {
                o = reflect(i, h);
                if(o.y <= R(0)) return false;
                R G = g.G2(i, o);
                R D = g.D(h);
                value = rgb(L.color) * (G * D * F / (R(4) * i.y * o.y));
                pdf_o = g.pdf(i, h) * (R(1) / (R(4) * idh)) * F;
                return true;
            }
            R eta = eta_a / eta_b;
            if(!refract(i, h, eta, o)) return false;
            if(o.y >= R(0)) return false;
            R odh = dot(o, h);
            R J = -odh / sqr(eta * idh + odh);
            R G = g.G2(i, o);
            R D = g.D(h);
            value = rgb(L.color2) * (idh * G * D * J * (R(1) - F) / (i.y * -o.y));
            pdf_o = g.pdf(i, h) * J * (R(1) - F);
            return true;
        }
        else if constexpr(TYPE == L_SPEC_GLASS)  // specular_glass.hpp:25-74
        {
            R cos_i = i.y;
            R eta = eta_a / eta_b;
            R cos_t;
            bool transmits;
            if constexpr(sizeof(R) == 4 && FCG_FP32_GLASS)
            {
                // fp32: the regrouped cos_t^2 of fr_dielectric (no cancellation at grazing incidence / close indices).
                // i is not a unit vector where the shading frame is not orthonormal (a rotated sphere: its tangent stays
                // unrotated, sphere_surface.hpp:130-140); the reference then takes sqrt(1 - cos_i^2) of a negative
                // number and its NaN fails `sin_t < 1`, i.e. it reflects with F = 1: the second test keeps that.
                const R cos2_t = (R(1) - eta) * (R(1) + eta) + eta * eta * cos_i * cos_i;
                transmits = cos2_t > R(0) && cos_i * cos_i <= R(1);
                cos_t = r_sqrt(r_max(cos2_t, R(0)));
            }
            else
            {
                R sin2_i = R(1) - cos_i * cos_i;
                R sin_i = r_sqrt(sin2_i);
                R sin_t = eta * sin_i;
                cos_t = r_sqrt(R(1) - sin_t * sin_t);
                transmits = sin_t < R(1);
            }
            R F = R(1);
            if(transmits)
            {
                R r_parl = ((eta_b * cos_i) - (eta_a * cos_t)) / ((eta_b * cos_i) + (eta_a * cos_t));
                R r_perp = ((eta_a * cos_i) - (eta_b * cos_t)) / ((eta_a * cos_i) + (eta_b * cos_t));
                F = (r_parl * r_parl + r_perp * r_perp) / R(2);
            }
            if(u1x < F)
            {
                o = vec3<R>{-i.x, i.y, -i.z};
                value = rgb(L.color) * (F / o.y);
                pdf_o = F;
                return true;
            }
            o = eta * (-i);
            o.y += eta * cos_i - cos_t;
            value = rgb(L.color2) * ((R(1) - F) * (eta_b * eta_b) / (eta_a * eta_a * -o.y));
            pdf_o = R(1) - F;
            return true;
        }
        else if constexpr(TYPE == L_GGX_TRANS)  // microfacet_transmission.hpp:48-83
        {
            if(i.y == R(0)) return false;
            Ggx<R> g{L.alpha[0], L.alpha[1]};
            vec3<R> h = g.sample(i, u1x, u1y);
            R idh = dot(i, h);
            if(idh <= R(0)) return false;
            R eta = eta_a / eta_b;
            if(!refract(i, h, eta, o)) return false;
            if(o.y >= R(0)) return false;
            R odh = dot(o, h);
            R J = -odh / sqr(eta * idh + odh);
            value = rgb(L.color) * (idh * g.G2(i, o) * g.D(h) * J / (i.y * -o.y));
            pdf_o = g.pdf(i, h) * J;
            return true;
        }
        else  // L_SPEC_TRANS, specular_transmission.hpp:25-37
        {
            if(!refract(i, vec3<R>{R(0), R(1), R(0)}, eta_a / eta_b, o)) return false;
            value = rgb(L.color) * ((eta_b * eta_b) / (eta_a * eta_a * -o.y));
            pdf_o = R(1);
            return true;
        }
    }

    // ---------------------------------------------------------------- bsdf + bxdf_adapter
    template<typename R>
    struct ShadingFrame
    {
        vec3<R> t, ns, b, ng;
        FC_HD vec3<R> to_local(vec3<R> w) const { return {dot(w, t), dot(w, ns), dot(w, b)}; }  // bsdf.hpp:156-163
        FC_HD vec3<R> to_world(vec3<R> w) const                                                  // bsdf.hpp:165-172
        {
            return {t.x * w.x + ns.x * w.y + b.x * w.z, t.y * w.x + ns.y * w.y + b.y * w.z, t.z * w.x + ns.z * w.y + b.z * w.z};
        }
    };

    // ---------------------------------------------------------------- normal_mapping<T>, bsdfs/normal_mapping.hpp
    // The mapped normal p and a vertical "tangent" facet t (which closes the profile and reflects specularly)
    // form a two-facet microsurface in the shading frame (y up). Light goes i -> o along at most three routes:
    //   P  : meets facet p, scatters by the inner lobe straight to o                         f(i, o)
    //   TP : meets facet t first, is mirrored (ri), then scatters on p to o                   f(ri, o)
    //   PT : scatters on p towards ro, the mirror image of o; facet t turns that into o        f(i, ro)
    // evaluate (:32-145), pdf (:629-731) and sample (:147-626) enumerate the 3 x 3 facet classes of (i, o); nm_combine
    // is those tables written once. on = false is the skip_ path (p within 0.001 of the shading normal, :745-749).
    template<typename R>
    struct NormalMapRec
    {
        bool on = false;
        vec3<R> p, t, tangent, bitangent;
        FC_HD vec3<R> to_p(vec3<R> w) const { return {dot(w, tangent), dot(w, p), dot(w, bitangent)}; }  // frame.hpp:22-25
        FC_HD vec3<R> from_p(vec3<R> w) const                                                            // frame.hpp:27-34
        {
            return {tangent.x * w.x + p.x * w.y + bitangent.x * w.z, tangent.y * w.x + p.y * w.y + bitangent.y * w.z,
                    tangent.z * w.x + p.z * w.y + bitangent.z * w.z};
        }
        FC_HD vec3<R> mirror(vec3<R> w) const { return w - t * (R(2) * dot(w, t)); }
        // facets a direction sees: 0 = only t, 1 = both, 2 = only p (:42-43)
        FC_HD int facets(vec3<R> w) const { return dot(w, p) <= R(0) ? 0 : (dot(w, t) > R(0) ? 1 : 2); }
        FC_HD R sine() const { return r_sqrt(R(1) - p.y * p.y); }
        FC_HD R hits_p(vec3<R> w) const  // probability that a ray along w meets facet p first (:66-69)
        {
            R a_p = dot(w, p) / p.y, a_t = dot(w, t) * sine() / p.y;
            return a_p / (a_p + a_t);
        }
        FC_HD R open_p(vec3<R> w) const { return w.y * p.y / dot(w, p); }             // gp_*: part of p not hidden behind t
        FC_HD R open_t(vec3<R> w) const { return w.y * p.y / (dot(w, t) * sine()); }  // gt_*
    };

    // the material's decode of a normal-map texel (diffuse_material.hpp:20-27 and twins) + normal_mapping::precompute (:743-760)
    template<typename R> FC_HD NormalMapRec<R> make_normal_map(vec3<R> texel)
    {
        NormalMapRec<R> N;
        vec3<R> n{texel.x * R(2) - R(1), texel.z * R(2) - R(1), texel.y * R(2) - R(1)};  // y <-> z: blue is up
        n = normalize(n);
        if(n.y < R(0)) n = -n;
        if(r_abs(n.x) < R(0.001) && r_abs(n.z) < R(0.001)) return N;
        N.on = true;
        N.p = n;
        N.t = normalize(vec3<R>{-n.x, R(0), -n.z});
        N.bitangent = normalize(cross(vec3<R>{R(1), R(0), R(0)}, n));
        N.tangent = cross(n, N.bitangent);
        return N;
    }

    enum NmKnown { NM_NONE = 0, NM_I_O = 1, NM_RI_O = 2, NM_I_RO = 3 };  // the inner pair sample() already holds

    template<int TYPE, typename R>
    FC_HD void nm_combine(const LobeRec<R>& L, const NormalMapRec<R>& N, vec3<R> i, vec3<R> o, int co, R eta_a, R eta_b, int known,
                          vec3<R> f_known, R pdf_known, bool want_f, bool want_pdf, vec3<R>& value, R& pdf)
    {
        const int ci = N.facets(i);
        value = vec3<R>{R(0), R(0), R(0)};
        pdf = R(0);
        if(ci == 0 && co == 0) return;
        const R lambda = ci == 0 ? R(0) : (ci == 2 ? R(1) : N.hits_p(i));
        auto inner = [&](int which, vec3<R> a, vec3<R> b, vec3<R>& f, R& q) {
            if(which == known)
            {
                f = f_known;
                q = pdf_known;
                return;
            }
            const vec3<R> la = N.to_p(a), lb = N.to_p(b);
            f = want_f ? lobe_evaluate<TYPE>(L, la, lb, eta_a, eta_b) : vec3<R>{R(0), R(0), R(0)};
            q = want_pdf ? lobe_pdf<TYPE>(L, la, lb, eta_a, eta_b) : R(0);
        };
        vec3<R> sum{R(0), R(0), R(0)};
        R q_sum = R(0);
        if(co != 0)  // o leaves facet p directly: routes P and TP
        {
            const R open = co == 2 ? N.open_p(o) : R(1);
            const R o_dot_p = dot(o, N.p);
            vec3<R> f;
            R q;
            if(ci != 0)
            {
                inner(NM_I_O, i, o, f, q);
                sum = sum + f * (lambda * o_dot_p * open);
                q_sum += lambda * q * open;
            }
            if(ci != 2)
            {
                inner(NM_RI_O, N.mirror(i), o, f, q);
                sum = sum + f * ((R(1) - lambda) * o_dot_p * open);
                q_sum += (R(1) - lambda) * q;  // no visibility term on this route in pdf (:640-648, :690-693)
            }
        }
        if(co != 2 && ci != 0)  // route PT
        {
            const vec3<R> ro = N.mirror(o);
            const R ro_dot_p = dot(ro, N.p);
            const R hidden = R(1) - N.open_p(ro);
            vec3<R> f;
            R q;
            inner(NM_I_RO, i, ro, f, q);
            sum = sum + f * (lambda * hidden * ro_dot_p * (co == 0 ? N.open_t(o) : R(1)));
            q_sum += lambda * q * hidden;
        }
        value = sum / o.y;
        pdf = q_sum;
    }

    template<int TYPE, typename R>
    FC_HD vec3<R> mapped_evaluate(const LobeRec<R>& L, const NormalMapRec<R>& N, vec3<R> i, vec3<R> o, R eta_a, R eta_b)
    {
        if(!N.on) return lobe_evaluate<TYPE>(L, i, o, eta_a, eta_b);
        vec3<R> f;
        R q;
        nm_combine<TYPE>(L, N, i, o, N.facets(o), eta_a, eta_b, NM_NONE, vec3<R>{R(0), R(0), R(0)}, R(0), true, false, f, q);
        return f;
    }
    template<int TYPE, typename R>
    FC_HD R mapped_pdf(const LobeRec<R>& L, const NormalMapRec<R>& N, vec3<R> i, vec3<R> o, R eta_a, R eta_b)
    {
        if(!N.on) return lobe_pdf<TYPE>(L, i, o, eta_a, eta_b);
        vec3<R> f;
        R q;
        nm_combine<TYPE>(L, N, i, o, N.facets(o), eta_a, eta_b, NM_NONE, vec3<R>{R(0), R(0), R(0)}, R(0), false, true, f, q);
        return q;
    }
    template<int TYPE, typename R>
    FC_HD bool mapped_sample(const LobeRec<R>& L, const NormalMapRec<R>& N, vec3<R> i, R eta_a, R eta_b, R u1x, R u1y, R u2x, R u2y,
                             vec3<R>& o, vec3<R>& value, R& pdf_o)
    {
        if(!N.on) return lobe_sample<TYPE>(L, i, eta_a, eta_b, u1x, u1y, u2x, u2y, o, value, pdf_o);
        const int ci = N.facets(i);
        // first encounter: facet t when i cannot see p, by the projected areas when it sees both (:232)
        const bool via_t = ci == 0 || (ci == 1 && !(u2x < N.hits_p(i)));
        const vec3<R> start = via_t ? N.mirror(i) : i;
        vec3<R> local_m, f_m;
        R pdf_m;
        if(!lobe_sample<TYPE>(L, N.to_p(start), eta_a, eta_b, u1x, u1y, u2x, u2y, local_m, f_m, pdf_m)) return false;
        const vec3<R> m = N.from_p(local_m);
        if(m.y <= R(0)) return false;
        const int cm = dot(m, N.t) > R(0) ? 1 : 2;
        int known = via_t ? NM_RI_O : NM_I_O;
        int co = cm;
        o = m;
        if(!via_t && cm == 2 && u2y > N.open_p(m))  // leaves p behind facet t: mirrored once more (:306-372, :551-603)
        {
            o = N.mirror(m);
            co = dot(o, N.p) > R(0) ? 1 : 0;
            known = NM_I_RO;
        }
        nm_combine<TYPE>(L, N, i, o, co, eta_a, eta_b, known, f_m, pdf_m, true, true, value, pdf_o);
        return true;
    }

    // bsdf::evaluate, bsdf.hpp:62-72 over bxdf_adapter::evaluate, bxdf.hpp:57-67. NM: the lobe sits inside a
    // normal_mapping<T> (diffuse / mirror / standard materials with a normal map)
    template<int TYPE, bool NM = false, typename R>
    FC_HD vec3<R> bsdf_evaluate(const LobeRec<R>& L, const ShadingFrame<R>& f, vec3<R> wo, vec3<R> wi, R eta_a, R eta_b,
                                const NormalMapRec<R>* N = nullptr)
    {
        R wo_wg = dot(wo, f.ng), wo_ws = dot(wo, f.ns), wi_wg = dot(wi, f.ng), wi_ws = dot(wi, f.ns);
        if(wo_wg * wo_ws <= R(0) || wi_wg * wi_ws <= R(0)) return vec3<R>{R(0), R(0), R(0)};
        R c = r_abs(wi_ws) * L.factor / r_abs(wi_wg);
        vec3<R> lo = f.to_local(wo), li = f.to_local(wi);
        vec3<R> v;
        if constexpr(NM)
            v = li.y >= R(0) ? mapped_evaluate<TYPE>(L, *N, li, lo, eta_a, eta_b) : mapped_evaluate<TYPE>(L, *N, -li, -lo, eta_b, eta_a);
        else
            v = li.y >= R(0) ? lobe_evaluate<TYPE>(L, li, lo, eta_a, eta_b) : lobe_evaluate<TYPE>(L, -li, -lo, eta_b, eta_a);
        return c * v;
    }

    // bsdf::pdf_wi, bsdf.hpp:131-141 over bxdf_adapter::pdf_wi, bxdf.hpp:117-127
    template<int TYPE, bool NM = false, typename R>
    FC_HD R bsdf_pdf_wi(const LobeRec<R>& L, const ShadingFrame<R>& f, vec3<R> wo, vec3<R> wi, R eta_a, R eta_b,
                        const NormalMapRec<R>* N = nullptr)
    {
        R wo_wg = dot(wo, f.ng), wo_ws = dot(wo, f.ns), wi_wg = dot(wi, f.ng), wi_ws = dot(wi, f.ns);
        if(wo_wg * wo_ws <= R(0) || wi_wg * wi_ws <= R(0)) return R(0);
        vec3<R> lo = f.to_local(wo), li = f.to_local(wi);
        if constexpr(NM)
            return lo.y >= R(0) ? mapped_pdf<TYPE>(L, *N, lo, li, eta_a, eta_b) : mapped_pdf<TYPE>(L, *N, -lo, -li, eta_b, eta_a);
        else
            return lo.y >= R(0) ? lobe_pdf<TYPE>(L, lo, li, eta_a, eta_b) : lobe_pdf<TYPE>(L, -lo, -li, eta_b, eta_a);
    }

    // bsdf::sample_wi, bsdf.hpp:74-100 over bxdf_adapter::sample_wi, bxdf.hpp:69-97
    template<int TYPE, bool NM = false, typename R>
    FC_HD bool bsdf_sample_wi(const LobeRec<R>& L, const ShadingFrame<R>& f, vec3<R> wo, R eta_a, R eta_b,
                              R u1x, R u1y, R u2x, R u2y, vec3<R>& wi, vec3<R>& value, R& pdf, const NormalMapRec<R>* N = nullptr)
    {
        R wo_wg = dot(wo, f.ng), wo_ws = dot(wo, f.ns);
        if(wo_wg * wo_ws <= R(0)) return false;
        vec3<R> lo = f.to_local(wo);
        vec3<R> li;
        if(lo.y >= R(0))
        {
            bool ok;
            if constexpr(NM)
                ok = mapped_sample<TYPE>(L, *N, lo, eta_a, eta_b, u1x, u1y, u2x, u2y, li, value, pdf);
            else
                ok = lobe_sample<TYPE>(L, lo, eta_a, eta_b, u1x, u1y, u2x, u2y, li, value, pdf);
            if(!ok) return false;
            if(li.y <= R(0)) value = value * ((eta_a * eta_a) / (eta_b * eta_b));
        }
        else
        {
            bool ok;
            if constexpr(NM)
                ok = mapped_sample<TYPE>(L, *N, -lo, eta_b, eta_a, u1x, u1y, u2x, u2y, li, value, pdf);
            else
                ok = lobe_sample<TYPE>(L, -lo, eta_b, eta_a, u1x, u1y, u2x, u2y, li, value, pdf);
            if(!ok) return false;
            li = -li;
            if(li.y >= R(0)) value = value * ((eta_b * eta_b) / (eta_a * eta_a));
        }
        wi = f.to_world(li);
        R wi_wg = dot(wi, f.ng), wi_ws = dot(wi, f.ns);
        if(wi_wg * wi_ws <= R(0)) return false;
        value = value * (r_abs(wi_ws) * L.factor / r_abs(wi_wg));
        return true;
    }

    template<typename R> FC_HD R power_heuristic(R primary, R alternative)  // forward_mis_integrator.hpp:215-219
    {
        R x = alternative / primary;
        return R(1) / (R(1) + x * x);
    }
}
