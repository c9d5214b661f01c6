// fc_math.h -- small vector / scalar helpers, generic over the arithmetic type R (float: the
// fast path, double: reference precision). Host + device.
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FC_HD __host__ __device__ __forceinline__
#define FC_D __device__ __forceinline__
#else
#define FC_HD inline
#define FC_D inline
#endif

namespace fcg
{
    template<typename R>
    struct vec3
    {
        R x, y, z;
    };

    template<typename R> FC_HD vec3<R> mk(R x, R y, R z) { return vec3<R>{x, y, z}; }
    template<typename R> FC_HD vec3<R> operator+(vec3<R> a, vec3<R> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
    template<typename R> FC_HD vec3<R> operator-(vec3<R> a, vec3<R> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
    template<typename R> FC_HD vec3<R> operator-(vec3<R> a) { return {-a.x, -a.y, -a.z}; }
    template<typename R> FC_HD vec3<R> operator*(vec3<R> a, vec3<R> b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
    template<typename R> FC_HD vec3<R> operator*(vec3<R> a, R s) { return {a.x * s, a.y * s, a.z * s}; }
    template<typename R> FC_HD vec3<R> operator*(R s, vec3<R> a) { return {a.x * s, a.y * s, a.z * s}; }
    template<typename R> FC_HD vec3<R> operator/(vec3<R> a, R s) { return {a.x / s, a.y / s, a.z / s}; }
    template<typename R> FC_HD R dot(vec3<R> a, vec3<R> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
    template<typename R> FC_HD vec3<R> cross(vec3<R> a, vec3<R> b)
    {
        return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
    }
    template<typename R> FC_HD R sqr(R a) { return a * a; }

    FC_HD float r_sqrt(float a) { return sqrtf(a); }
    FC_HD double r_sqrt(double a) { return sqrt(a); }
    FC_HD float r_abs(float a) { return fabsf(a); }
    FC_HD double r_abs(double a) { return fabs(a); }
    FC_HD float r_sin(float a) { return sinf(a); }
    FC_HD double r_sin(double a) { return sin(a); }
    FC_HD float r_cos(float a) { return cosf(a); }
    FC_HD double r_cos(double a) { return cos(a); }
    FC_HD float r_acos(float a) { return acosf(a); }
    FC_HD double r_acos(double a) { return acos(a); }
    FC_HD float r_atan2(float a, float b) { return atan2f(a, b); }
    FC_HD double r_atan2(double a, double b) { return atan2(a, b); }
    FC_HD float r_floor(float a) { return floorf(a); }
    FC_HD double r_floor(double a) { return floor(a); }
    FC_HD float r_exp(float a) { return expf(a); }
    FC_HD double r_exp(double a) { return exp(a); }
    FC_HD float r_min(float a, float b) { return fminf(a, b); }
    FC_HD double r_min(double a, double b) { return fmin(a, b); }
    FC_HD float r_max(float a, float b) { return fmaxf(a, b); }
    FC_HD double r_max(double a, double b) { return fmax(a, b); }
    FC_HD bool r_isfinite(float a) { return __builtin_isfinite(a); }
    FC_HD bool r_isfinite(double a) { return __builtin_isfinite(a); }

    template<typename R> FC_HD R length(vec3<R> a) { return r_sqrt(dot(a, a)); }
    template<typename R> FC_HD vec3<R> normalize(vec3<R> a) { return a / length(a); }
    template<typename R> FC_HD R comp(vec3<R> a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
    template<typename R> FC_HD R clampr(R v, R lo, R hi) { return v < lo ? lo : (v > hi ? hi : v); }

    // 3x4 row-major affine transform, applied as transform.hpp:19-49 does
    template<typename R> FC_HD vec3<R> xf_point(const R* m, vec3<R> p)
    {
        return {m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3],
                m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
                m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]};
    }
    template<typename R> FC_HD vec3<R> xf_dir(const R* m, vec3<R> d)
    {
        return {m[0] * d.x + m[1] * d.y + m[2] * d.z,
                m[4] * d.x + m[5] * d.y + m[6] * d.z,
                m[8] * d.x + m[9] * d.y + m[10] * d.z};
    }

    template<typename R> struct consts;
    template<> struct consts<float>
    {
        static constexpr float pi = 3.14159265358979323846f;
        static constexpr float inv_pi = 0.31830988618379067154f;
        static constexpr float inf = __builtin_huge_valf();
    };
    template<> struct consts<double>
    {
        static constexpr double pi = 3.141592653589793;
        static constexpr double inv_pi = 1.0 / 3.141592653589793;
        static constexpr double inf = __builtin_huge_val();
    };
}
