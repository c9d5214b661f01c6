// fc_env.h -- the environment light on the device: constant (const_infinity_area_light.hpp) or
// image-textured with importance sampling (texture_infinity_area_light.hpp:15-131 over
// image_texture_2d_rgb, image_texture.hpp:88-124, and distribution_2d, distribution.hpp:23-166).
#pragma once
#include "../../include/fcgpu.h"
#include "fc_bsdf.h"
#include "fc_types.h"

namespace fcg
{
    template<typename R> FC_D vec3<R> tex_texel(const TexRef<R>& E, int x, int y)
    {
        const Vec4<R> t = E.texels[static_cast<size_t>(E.width) * static_cast<size_t>(y) + static_cast<size_t>(x)];
        return vec3<R>{t.x, t.y, t.z};
    }

    FC_D int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

    // image_texture_2d_rgb::evaluate
    template<typename R> FC_D vec3<R> tex_evaluate(const TexRef<R>& E, R u, R v)
    {
        if(E.kind == FCGPU_TEXTURE_CHECKER)
        {
            // checker_texture.hpp:14-20 (int arithmetic: a negative odd sum has remainder -1, which is "not 0" too)
            const int x = static_cast<int>(r_floor(u * E.checker_scale)), y = static_cast<int>(r_floor(v * E.checker_scale));
            const R* c = (x + y) % 2 == 0 ? E.checker_a : E.checker_b;
            return vec3<R>{c[0], c[1], c[2]};
        }
        if(E.filter == FCGPU_FILTER_BILINEAR)
        {
            // :99-124: texel centres at (i + 0.5) / resolution, clamp addressing
            const R ax = u * R(E.width) - R(0.5), ay = v * R(E.height) - R(0.5);
            const int x0 = static_cast<int>(r_floor(ax)), y0 = static_cast<int>(r_floor(ay));
            const int px0 = clampi(x0, 0, E.width - 1), px1 = clampi(x0 + 1, 0, E.width - 1);
            const int py0 = clampi(y0, 0, E.height - 1), py1 = clampi(y0 + 1, 0, E.height - 1);
            const vec3<R> v00 = tex_texel(E, px0, py0), v10 = tex_texel(E, px1, py0), v01 = tex_texel(E, px0, py1), v11 = tex_texel(E, px1, py1);
            const R wx = ax - R(x0), wy = ay - R(y0);
            const vec3<R> v0 = v00 * (R(1) - wx) + v10 * wx;
            const vec3<R> v1 = v01 * (R(1) - wx) + v11 * wx;
            return v0 * (R(1) - wy) + v1 * wy;
        }
        // box, :88-97
        int px = static_cast<int>(u * R(E.width)), py = static_cast<int>(v * R(E.height));
        // the reference clamps only from above (uv below 0 indexes out of bounds there); clamped on both sides here
        px = clampi(px, 0, E.width - 1);
        py = clampi(py, 0, E.height - 1);
        return tex_texel(E, px, py);
    }

    // distribution_1d::sample_continuous (distribution.hpp:62-77) over n cells
    FC_D void dist1d_sample(const double* func, const double* cdf, double integral, int n, double u, double& x, double& pdf, int& index)
    {
        const double one_minus_eps = 0x1.fffffffffffffp-1;
        u = u < 0.0 ? 0.0 : (u > one_minus_eps ? one_minus_eps : u);
        int lo = 0, hi = n + 1;  // upper_bound over cdf[0..n]
        while(lo < hi)
        {
            const int mid = (lo + hi) >> 1;
            if(cdf[mid] > u)
                hi = mid;
            else
                lo = mid + 1;
        }
        const int upper = lo, lower = lo - 1;
        const double du = (u - cdf[lower]) / (cdf[upper] - cdf[lower]);
        x = (static_cast<double>(lower) + du) / static_cast<double>(n);
        pdf = func[lower] / integral;
        index = lower;
    }

    // distribution_1d::pdf_continuous (:79-87)
    FC_D double dist1d_pdf(const double* func, double integral, int n, double x, int& index)
    {
        double fi = x * static_cast<double>(n);
        // static_cast<size_t> of a negative value is not reached: x comes from acos / atan2 ranges >= 0
        long long i = static_cast<long long>(fi);
        i = i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
        index = static_cast<int>(i);
        return func[i] / integral;
    }

    // direction -> (u, v) of the latitude-longitude map, texture_infinity_area_light.hpp:58-65
    template<typename R> FC_D void env_uv(const EnvTex<R>& E, vec3<R> wi, R& u, R& v, R& theta)
    {
        const vec3<R> w = xf_dir(E.inv, wi);
        const R wy = w.y < R(-1) ? R(-1) : (w.y > R(1) ? R(1) : w.y);
        theta = r_acos(wy);
        const R p = r_atan2(w.z, w.x);
        const R phi = p < R(0) ? p + R(2) * consts<R>::pi : p;
        v = theta / consts<R>::pi;
        u = R(1) - phi / (R(2) * consts<R>::pi);
    }

    // infinity_area_light::get_Li
    template<typename R> FC_D vec3<R> env_Li(const SceneDev<R>& S, vec3<R> wi)
    {
        if(S.env_kind != FCGPU_ENV_TEXTURE) return rgb(S.env);
        R u, v, theta;
        env_uv(S.envtex, wi, u, v, theta);
        return tex_evaluate(S.envtex.tex, u, v) * S.envtex.strength;
    }

    // infinity_area_light::pdf_wi (solid angle)
    template<typename R> FC_D R env_pdf(const SceneDev<R>& S, vec3<R> wi)
    {
        if(S.env_kind != FCGPU_ENV_TEXTURE) return R(1) / (R(4) * consts<R>::pi);  // const_infinity_area_light.hpp
        const EnvTex<R>& E = S.envtex;
        R u, v, theta;
        env_uv(E, wi, u, v, theta);
        const R sin_theta = r_sin(theta);
        if(sin_theta == R(0)) return R(0);
        int iy, ix;
        const double py = dist1d_pdf(E.func_y, E.int_y, E.ny, static_cast<double>(v), iy);
        const double px = dist1d_pdf(E.func_x + static_cast<size_t>(iy) * E.nx, E.int_x[iy], E.nx, static_cast<double>(u), ix);
        return R(px * py) / (R(2) * consts<R>::pi * consts<R>::pi * sin_theta);
    }

    // infinity_area_light::sample_wi: false = no sample (the estimator term is skipped)
    template<typename R> FC_D bool env_sample(const SceneDev<R>& S, R ux, R uy, vec3<R>& wi, R& pdf_wi, vec3<R>& Li)
    {
        if(S.env_kind != FCGPU_ENV_TEXTURE)
        {
            // const_infinity_area_light::sample_wi, const_infinity_area_light.hpp:33-44
            Li = rgb(S.env);
            if(Li.x == R(0) && Li.y == R(0) && Li.z == R(0)) return false;
            wi = sample_sphere_uniform(ux, uy);
            pdf_wi = R(1) / (R(4) * consts<R>::pi);
            return true;
        }
        const EnvTex<R>& E = S.envtex;
        // distribution_2d::sample_continuous (distribution.hpp:143-152): y from u.y, then x from u.x in that row
        double sy, pdf_y, sx, pdf_x;
        int iy, ix;
        dist1d_sample(E.func_y, E.cdf_y, E.int_y, E.ny, static_cast<double>(uy), sy, pdf_y, iy);
        dist1d_sample(E.func_x + static_cast<size_t>(iy) * E.nx, E.cdf_x + static_cast<size_t>(iy) * (E.nx + 1), E.int_x[iy], E.nx,
                      static_cast<double>(ux), sx, pdf_x, ix);
        const R su = R(sx), sv = R(sy);
        const R theta = sv * consts<R>::pi;
        const R phi = (R(1) - su) * R(2) * consts<R>::pi;
        const R cos_theta = r_cos(theta), sin_theta = r_sin(theta);
        if(sin_theta == R(0)) return false;
        Li = tex_evaluate(E.tex, su, sv) * E.strength;
        if(Li.x == R(0) && Li.y == R(0) && Li.z == R(0)) return false;
        const vec3<R> w{sin_theta * r_cos(phi), cos_theta, sin_theta * r_sin(phi)};
        wi = xf_dir(E.fwd, w);
        pdf_wi = R(pdf_x * pdf_y) / (R(2) * consts<R>::pi * consts<R>::pi * sin_theta);
        return true;
    }
}
