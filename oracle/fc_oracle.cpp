// fc_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see fc_oracle.hpp). CPU restatement of the
// reference's Forward-MIS / Forward-BSDF path, sample by sample, in the reference's own order.
// Parity pinning: oracle/make_golden.py + tests/golden/* (vectors produced by the REAL reference)
// and tests/test_oracle_vs_reference.py. Each block cites the reference lines it follows.
#include "fc_oracle.hpp"

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <thread>

namespace fco
{
    // 0 = random stream; else stratified_sampler{count, seed} analogue (fcgpu_sampler.h); set through *_set_stratified
    static uint32_t g_stratified_count = 0;

    // Conditioning probe (tools/ulp_sensitivity.py): with a non-zero seed every transcendental result is moved by -1, 0 or +1
    // unit in the last place, chosen by a hash of the argument and the seed. The device's libm and glibc differ by that
    // much; a pixel whose value moves by 1e-9 under this noise is ill-conditioned, not wrong. 0 (default) = off: the plain libm.
    static std::atomic<uint64_t> g_ulp_noise{0};

    namespace lm
    {
        template<typename R> inline R nudge(R y, R x)
        {
            const uint64_t seed = g_ulp_noise.load(std::memory_order_relaxed);
            if(seed == 0 || !(y == y) || y == R(0)) return y;
            double xd = static_cast<double>(x);
            uint64_t h;
            std::memcpy(&h, &xd, sizeof h);
            h = (h ^ seed) * 0x9E3779B97F4A7C15ull;
            h ^= h >> 29;
            h *= 0xBF58476D1CE4E5B9ull;
            h ^= h >> 32;
            const int k = static_cast<int>(h % 3u) - 1;
            return k == 0 ? y : std::nextafter(y, k > 0 ? std::numeric_limits<R>::infinity() : -std::numeric_limits<R>::infinity());
        }
        template<typename R> inline R sin(R x) { return nudge(std::sin(x), x); }
        template<typename R> inline R cos(R x) { return nudge(std::cos(x), x); }
        template<typename R> inline R tan(R x) { return nudge(std::tan(x), x); }
        template<typename R> inline R exp(R x) { return nudge(std::exp(x), x); }
        template<typename R> inline R acos(R x) { return nudge(std::acos(x), x); }
        template<typename R> inline R pow(R x, R y) { return nudge(std::pow(x, y), x); }
        template<typename R> inline R atan2(R y, R x) { return nudge(std::atan2(y, x), y + x); }
    }

    namespace
    {
        constexpr double PI = 3.141592653589793;  // math.hpp:13

        // ------------------------------------------------------------ transforms (transform.hpp)
        Mat identity()
        {
            Mat r;
            for(int i = 0; i < 4; ++i) r.m[i][i] = 1.0;
            return r;
        }
        Mat operator*(const Mat& a, const Mat& b)  // math.hpp:513-527
        {
            Mat c;
            for(int i = 0; i < 4; ++i)
                for(int j = 0; j < 4; ++j)
                    c.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j] + a.m[i][3] * b.m[3][j];
            return c;
        }
        Mat translate(double x, double y, double z)  // math.hpp:453-461
        {
            Mat r = identity();
            r.m[0][3] = x; r.m[1][3] = y; r.m[2][3] = z;
            return r;
        }
        Mat scaling(double x, double y, double z)
        {
            Mat r = identity();
            r.m[0][0] = x; r.m[1][1] = y; r.m[2][2] = z;
            return r;
        }
        Mat rotate_x(double t)  // math.hpp:473-484
        {
            Mat r = identity();
            r.m[1][1] = lm::cos(t); r.m[1][2] = -lm::sin(t);
            r.m[2][1] = lm::sin(t); r.m[2][2] = lm::cos(t);
            return r;
        }
        Mat rotate_y(double t)  // math.hpp:486-497
        {
            Mat r = identity();
            r.m[0][0] = lm::cos(t); r.m[0][2] = lm::sin(t);
            r.m[2][0] = -lm::sin(t); r.m[2][2] = lm::cos(t);
            return r;
        }
        Mat rotate_z(double t)  // math.hpp:499-510
        {
            Mat r = identity();
            r.m[0][0] = lm::cos(t); r.m[0][1] = -lm::sin(t);
            r.m[1][0] = lm::sin(t); r.m[1][1] = lm::cos(t);
            return r;
        }

        struct MatPair
        {
            Mat t, inv;
        };
        // pr_transform(position, rotation) transform.hpp:14-17; prs_transform(+scale) :79-82;
        // the translation-only constructors (:11-13, :72-74) when rotation (and scale) are trivial
        MatPair make_transform(const double* p, const double* r, const double* s)
        {
            bool rot = r[0] != 0.0 || r[1] != 0.0 || r[2] != 0.0;
            bool scl = s && (s[0] != 1.0 || s[1] != 1.0 || s[2] != 1.0);
            MatPair m;
            if(!rot && !scl)
            {
                m.t = translate(p[0], p[1], p[2]);
                m.inv = translate(-p[0], -p[1], -p[2]);
                return m;
            }
            m.t = translate(p[0], p[1], p[2]) * rotate_y(r[1]) * rotate_x(r[0]) * rotate_z(r[2]);
            m.inv = rotate_z(-r[2]) * rotate_x(-r[0]) * rotate_y(-r[1]) * translate(-p[0], -p[1], -p[2]);
            if(scl)
            {
                m.t = m.t * scaling(s[0], s[1], s[2]);
                m.inv = scaling(1.0 / s[0], 1.0 / s[1], 1.0 / s[2]) * m.inv;
            }
            return m;
        }
        template<typename R> Transform<R> convert(const MatPair& m)
        {
            Transform<R> t;
            for(int i = 0; i < 3; ++i)
                for(int j = 0; j < 4; ++j)
                {
                    t.t[i][j] = static_cast<R>(m.t.m[i][j]);
                    t.inv[i][j] = static_cast<R>(m.inv.m[i][j]);
                }
            return t;
        }
        V3<double> mat_point(const Mat& m, V3<double> p)
        {
            return {m.m[0][0] * p.x + m.m[0][1] * p.y + m.m[0][2] * p.z + m.m[0][3], m.m[1][0] * p.x + m.m[1][1] * p.y + m.m[1][2] * p.z + m.m[1][3],
                    m.m[2][0] * p.x + m.m[2][1] * p.y + m.m[2][2] * p.z + m.m[2][3]};
        }

        void grow(BoxF& b, const float* p)
        {
            for(int i = 0; i < 3; ++i)
            {
                b.lo[i] = std::min(b.lo[i], p[i]);
                b.hi[i] = std::max(b.hi[i], p[i]);
            }
        }
        void grow(BoxF& b, const BoxF& o)
        {
            for(int i = 0; i < 3; ++i)
            {
                b.lo[i] = std::min(b.lo[i], o.lo[i]);
                b.hi[i] = std::max(b.hi[i], o.hi[i]);
            }
        }
        float area(const BoxF& b)  // math.hpp:607-611
        {
            float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
            return 2.0f * (dx * dy + dx * dz + dy * dz);
        }
        BoxF transformed_bounds(const Mat& t, V3<double> lo, V3<double> hi)  // transform.hpp:51-59 then bounds3f{...}
        {
            double mn[3]{1e300, 1e300, 1e300}, mx[3]{-1e300, -1e300, -1e300};
            for(int c = 0; c < 8; ++c)
            {
                V3<double> q = mat_point(t, {(c & 1) ? hi.x : lo.x, (c & 4) ? hi.y : lo.y, (c & 2) ? hi.z : lo.z});  // math.hpp:575-578
                for(int k = 0; k < 3; ++k)
                {
                    mn[k] = std::min(mn[k], q[k]);
                    mx[k] = std::max(mx[k], q[k]);
                }
            }
            BoxF b;
            for(int k = 0; k < 3; ++k)
            {
                b.lo[k] = static_cast<float>(mn[k]);
                b.hi[k] = static_cast<float>(mx[k]);
            }
            return b;
        }
    }

    // ================================================================ scene
    // ---------------------------------------------------------------- image texture + environment light
    // image::rgb() of the four pixel classes: r8_image.hpp:22-40, rgb8_image.hpp, srgb8_image.hpp with
    // srgb_to_rgb (core/color.hpp:13-26), rgb32_image.hpp; raw bytes as assets::load_image reads them
    struct ImageD
    {
        int w{}, h{};
        std::vector<double> rgb;
        void decode(const fcgpu_image& im)
        {
            w = im.width;
            h = im.height;
            const size_t n = static_cast<size_t>(w) * static_cast<size_t>(h);
            rgb.assign(3 * n, 0.0);
            const unsigned char* b = static_cast<const unsigned char*>(im.pixels);
            for(size_t i = 0; i < n; ++i)
                for(int c = 0; c < 3; ++c)
                {
                    double v = 0.0;
                    if(im.format == FCGPU_IMAGE_R8)
                        v = c == 0 ? b[i] / 255.0 : 0.0;
                    else if(im.format == FCGPU_IMAGE_RGB8)
                        v = b[3 * i + c] / 255.0;
                    else if(im.format == FCGPU_IMAGE_SRGB8)
                    {
                        double x = b[3 * i + c] / 255.0;
                        v = x <= 0.04045 ? x / 12.92 : lm::pow((x + 0.055) / 1.055, 2.4);
                    }
                    else
                    {
                        float f;
                        std::memcpy(&f, b + 4 * (3 * i + c), 4);
                        v = f;
                    }
                    rgb[3 * i + c] = v;
                }
        }
        V3<double> texel(int x, int y) const
        {
            const double* p = &rgb[3 * (static_cast<size_t>(w) * y + x)];
            return {p[0], p[1], p[2]};
        }
    };

    // image_texture_2d_rgb::evaluate, image_texture.hpp:22-33 with box :88-97 and bilinear :99-124
    // (image_texture_2d_r, :131-141 / :193-233, is the same arithmetic on image::r() = rgb().x)
    template<typename T> V3<T> image_evaluate(const ImageD& image, int filter, T u, T v)
    {
        const int W = image.w, H = image.h;
        auto tx = [&](int x, int y) {
            V3<double> t = image.texel(x, y);
            return V3<T>{static_cast<T>(t.x), static_cast<T>(t.y), static_cast<T>(t.z)};
        };
        if(filter == FCGPU_FILTER_BILINEAR)
        {
            T ax = u * T(W) - T(0.5), ay = v * T(H) - T(0.5);
            int x0 = static_cast<int>(std::floor(ax)), y0 = static_cast<int>(std::floor(ay));
            int px0 = std::clamp(x0, 0, W - 1), px1 = std::clamp(x0 + 1, 0, W - 1);
            int py0 = std::clamp(y0, 0, H - 1), py1 = std::clamp(y0 + 1, 0, H - 1);
            V3<T> v00 = tx(px0, py0), v10 = tx(px1, py0), v01 = tx(px0, py1), v11 = tx(px1, py1);
            T wx = ax - T(x0), wy = ay - T(y0);
            V3<T> a = v00 * (T(1) - wx) + v10 * wx;
            V3<T> b = v01 * (T(1) - wx) + v11 * wx;
            return a * (T(1) - wy) + b * wy;
        }
        // (negative uv would index out of bounds in the reference; clamped on both sides here and on the device)
        int px = std::clamp(static_cast<int>(u * T(W)), 0, W - 1), py = std::clamp(static_cast<int>(v * T(H)), 0, H - 1);
        return tx(px, py);
    }

    // distribution_1d, distribution.hpp:23-121
    struct Dist1
    {
        std::vector<double> f, cdf;
        double integral{};
        void build(std::vector<double> function)  // :31-59
        {
            f = std::move(function);
            const size_t n = f.size();
            cdf.assign(n + 1, 0.0);
            for(size_t i = 1; i <= n; ++i) cdf[i] = cdf[i - 1] + f[i - 1];
            integral = cdf[n] / static_cast<double>(n);
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
        }
        void sample_continuous(double u, double& x, double& pdf, size_t& index) const  // :62-77
        {
            u = std::clamp(u, 0.0, 0x1.fffffffffffffp-1);
            size_t upper = static_cast<size_t>(std::upper_bound(cdf.begin(), cdf.end(), u) - cdf.begin());
            size_t lower = upper - 1;
            double du = (u - cdf[lower]) / (cdf[upper] - cdf[lower]);
            x = (static_cast<double>(lower) + du) / static_cast<double>(f.size());
            pdf = f[lower] / integral;
            index = lower;
        }
        double pdf_continuous(double x, size_t& index) const  // :79-87
        {
            index = std::clamp(static_cast<size_t>(x * static_cast<double>(f.size())), size_t{0}, f.size() - 1);
            return f[index] / integral;
        }
    };

    template<typename R>
    struct EnvTexture  // texture_infinity_area_light.hpp:15-131
    {
        bool on{};
        ImageD image;
        int filter{}, integral_samples{};
        R strength{};
        Transform<R> xf{};
        std::vector<Dist1> rows;  // distribution_2d, distribution.hpp:124-166
        Dist1 marginal;

        template<typename T> V3<T> evaluate(T u, T v) const { return image_evaluate<T>(image, filter, u, v); }

        // image_texture_2d_rgb::integrate, image_texture.hpp:35-78 (always double: host-side set-up)
        V3<double> integrate(double au, double av, double bu, double bv) const
        {
            const int W = image.w, H = image.h;
            int p0x = static_cast<int>(std::floor(au * W)), p0y = static_cast<int>(std::floor(av * H));
            int p1x = static_cast<int>(std::ceil(bu * W)), p1y = static_cast<int>(std::ceil(bv * H));
            V3<double> value{};
            for(int i = p0y; i < p1y; ++i)
                for(int j = p0x; j < p1x; ++j)
                {
                    double afx = std::max(au, static_cast<double>(j) / W), afy = std::max(av, static_cast<double>(i) / H);
                    double bfx = std::min(bu, static_cast<double>(j + 1) / W), bfy = std::min(bv, static_cast<double>(i + 1) / H);
                    double du = (bfx - afx) / static_cast<double>(integral_samples), dv = (bfy - afy) / static_cast<double>(integral_samples);
                    V3<double> acc{};
                    for(int ii = 0; ii < integral_samples; ++ii)
                    {
                        double v = afy + (ii + 0.5) * dv;
                        for(int jj = 0; jj < integral_samples; ++jj) acc = acc + evaluate<double>(afx + (jj + 0.5) * du, v);
                    }
                    value = value + acc * (du * dv);
                }
            return value;
        }

        void build(const fcgpu_scene_desc& d)  // constructor, :15-38
        {
            const fcgpu_env_texture& t = d.env_texture;
            on = true;
            image.decode(d.images[t.image]);
            filter = t.filter;
            integral_samples = t.integral_sample_count;
            strength = static_cast<R>(t.strength);
            const double zero[3]{0.0, 0.0, 0.0};
            xf = convert<R>(make_transform(zero, t.rotation, nullptr));
            const int nx = t.distribution_width, ny = t.distribution_height;
            std::vector<double> yf;
            rows.resize(ny);
            for(int i = 0; i < ny; ++i)
            {
                std::vector<double> row;
                double sin_theta = lm::sin(PI * (i + 0.5) / ny);
                for(int j = 0; j < nx; ++j)
                {
                    V3<double> in = integrate(j / static_cast<double>(nx), i / static_cast<double>(ny), (j + 1) / static_cast<double>(nx),
                                              (i + 1) / static_cast<double>(ny));
                    row.push_back((0.212671 * in.x + 0.715160 * in.y + 0.072169 * in.z) * sin_theta);  // luminance, core/color.hpp:8-11
                }
                rows[i].build(std::move(row));
                yf.push_back(rows[i].integral);  // distribution.hpp:134-140
            }
            marginal.build(std::move(yf));
        }

        void uv_of(V3<R> wi, R& u, R& v, R& theta) const  // :58-65
        {
            V3<R> w = xf.inv_dir(wi);
            theta = lm::acos(std::clamp(w.y, R(-1), R(1)));
            R p = lm::atan2(w.z, w.x);
            R phi = p < R(0) ? p + R(2) * R(PI) : p;
            v = theta / R(PI);
            u = R(1) - phi / (R(2) * R(PI));
        }
        V3<R> get_Li(V3<R> wi) const  // :56-68
        {
            R u, v, theta;
            uv_of(wi, u, v, theta);
            return evaluate<R>(u, v) * strength;
        }
        R pdf_wi(V3<R> wi) const  // :124-139
        {
            R u, v, theta;
            uv_of(wi, u, v, theta);
            R sin_theta = lm::sin(theta);
            if(sin_theta == R(0)) return R(0);
            size_t iy, ix;
            double py = marginal.pdf_continuous(static_cast<double>(v), iy);  // distribution_2d::pdf_continuous, :154-160
            double px = rows[iy].pdf_continuous(static_cast<double>(u), ix);
            return R(px * py) / (R(2) * R(PI) * R(PI) * sin_theta);
        }
        bool sample_wi(R ux, R uy, V3<R>& wi, R& pdf, V3<R>& Li) const  // :70-96
        {
            double sy, sx, pdf_y, pdf_x;
            size_t iy, ix;
            marginal.sample_continuous(static_cast<double>(uy), sy, pdf_y, iy);  // distribution_2d::sample_continuous, :143-152
            rows[iy].sample_continuous(static_cast<double>(ux), sx, pdf_x, ix);
            R su = R(sx), sv = R(sy);
            R theta = sv * R(PI), phi = (R(1) - su) * R(2) * R(PI);
            R cos_theta = lm::cos(theta), sin_theta = lm::sin(theta);
            if(sin_theta == R(0)) return false;
            Li = evaluate<R>(su, sv) * strength;
            if(!nonzero(Li)) return false;
            wi = xf.dir(V3<R>{sin_theta * lm::cos(phi), cos_theta, sin_theta * lm::sin(phi)});
            pdf = R(pdf_x * pdf_y) / (R(2) * R(PI) * R(PI) * sin_theta);
            return true;
        }
    };

    template<typename R>
    class Scene
    {
    public:
        struct Surface
        {
            int kind{};
            Transform<R> xf{};
            R radius{}, size_x{}, size_y{};
            // mesh (world-space float vertices, mesh_surface.hpp:23-38)
            std::vector<float> positions, normals, uvs;
            std::vector<uint32_t> indices;
            bool has_normals{}, has_uvs{};
            uint32_t prim_count{1};
            double area{};
            std::vector<double> cdf;  // distribution_1d over triangle areas (emitters)
        };
        struct Entity
        {
            int surface{}, material{}, medium{-1}, light{-1};
            uint32_t first_prim{};
        };
        struct Light
        {
            int entity{};
            V3<R> color{};
            R strength{};
        };
        struct Prim
        {
            int entity;
            uint32_t index;
        };
        struct Node  // bvh_acceleration_structure::node, :141-195
        {
            BoxF box;
            uint32_t a;
            uint16_t b, interior;
        };

        std::vector<Surface> surfaces;
        std::vector<fcgpu_material> materials;
        std::vector<ImageD> images;            // decoded, for the image textures of materials
        std::vector<fcgpu_texture> textures;
        std::vector<Medium<R>> media;
        std::vector<Entity> entities;
        std::vector<Light> lights;
        std::vector<Prim> prims;        // entity order (brute force order)
        std::vector<Prim> bvh_prims;    // BVH leaf order
        std::vector<Node> nodes;
        fcgpu_camera cam{};
        Transform<R> cam_xf{};
        R pixel_size{}, plane_w{}, plane_h{}, focus{}, lens_radius{};
        bool has_env{};
        V3<R> env{};
        EnvTexture<R> envtex{};  // texture_infinity_area_light when envtex.on
        R epsilon{R(0.000001)};  // scene.hpp:182
        bool offset_along_ray{false};  // fp32 device mode (include/fcgpu.h FCGPU_OFFSET_ALONG_RAY); false = the reference's offset
        Medium<R> vacuum{};      // helper::dummy_medium_, integrator.hpp:115

        explicit Scene(const fcgpu_scene_desc& d)
        {
            vacuum.priority = -1;
            vacuum.ior = R(1);
            vacuum.vacuum = true;
            for(uint32_t i = 0; i < d.surface_count; ++i) surfaces.push_back(make_surface(d, d.surfaces[i]));
            materials.assign(d.materials, d.materials + d.material_count);
            if(d.texture_count)
            {
                textures.assign(d.textures, d.textures + d.texture_count);
                images.resize(d.image_count);
                for(uint32_t i = 0; i < d.image_count; ++i) images[i].decode(d.images[i]);
            }
            for(uint32_t i = 0; i < d.medium_count; ++i)
            {
                Medium<R> m;
                m.priority = d.media[i].priority;
                m.ior = static_cast<R>(d.media[i].ior);
                m.color = {static_cast<R>(d.media[i].color[0]), static_cast<R>(d.media[i].color[1]), static_cast<R>(d.media[i].color[2])};
                m.density = static_cast<R>(d.media[i].density);
                media.push_back(m);
            }
            // entity_scene constructor, scene.hpp:35-74
            for(uint32_t i = 0; i < d.entity_count; ++i)
            {
                const fcgpu_entity& e = d.entities[i];
                Entity ent;
                ent.surface = e.surface;
                ent.material = e.material;
                ent.medium = e.medium;
                ent.first_prim = static_cast<uint32_t>(prims.size());
                for(uint32_t p = 0; p < surfaces[e.surface].prim_count; ++p) prims.push_back({static_cast<int>(i), p});
                if(e.emissive)
                {
                    prepare_for_sampling(surfaces[e.surface]);
                    ent.light = static_cast<int>(lights.size());
                    Light l;
                    l.entity = static_cast<int>(i);
                    l.color = {static_cast<R>(e.emission_color[0]), static_cast<R>(e.emission_color[1]), static_cast<R>(e.emission_color[2])};
                    l.strength = static_cast<R>(e.emission_strength);
                    lights.push_back(l);
                }
                entities.push_back(ent);
            }
            has_env = d.env_kind == FCGPU_ENV_CONST || d.env_kind == FCGPU_ENV_TEXTURE;
            env = {static_cast<R>(d.env_color[0]), static_cast<R>(d.env_color[1]), static_cast<R>(d.env_color[2])};
            if(d.env_kind == FCGPU_ENV_TEXTURE) envtex.build(d);
            build_bvh();
            // perspective_camera constructor, perspective_camera.hpp:19-27
            cam = d.camera;
            cam_xf = convert<R>(make_transform(cam.position, cam.rotation, nullptr));
            double f = cam.lens_radius == 0.0 ? 1.0 : cam.focus_distance;
            double px = 2.0 * f * lm::tan(cam.fov / 2.0) / static_cast<double>(cam.height);
            pixel_size = static_cast<R>(px);
            plane_w = static_cast<R>(cam.width * px);
            plane_h = static_cast<R>(cam.height * px);
            focus = static_cast<R>(f);
            lens_radius = static_cast<R>(cam.lens_radius);
        }

        int light_count() const { return static_cast<int>(lights.size()) + (has_env ? 1 : 0); }  // scene.hpp:66-73

        // ------------------------------------------------------------ surfaces
        static Surface make_surface(const fcgpu_scene_desc& d, const fcgpu_surface& s)
        {
            Surface out;
            out.kind = s.kind;
            MatPair mp = make_transform(s.position, s.rotation, s.kind == FCGPU_SURFACE_MESH ? s.scale : nullptr);
            out.xf = convert<R>(mp);
            if(s.kind == FCGPU_SURFACE_SPHERE)
            {
                out.radius = static_cast<R>(s.radius);
                out.area = 4.0 * PI * s.radius * s.radius;  // sphere_surface.hpp:31-34
            }
            else if(s.kind == FCGPU_SURFACE_PLANE)
            {
                out.size_x = static_cast<R>(s.size[0]);
                out.size_y = static_cast<R>(s.size[1]);
                out.area = s.size[0] * s.size[1];  // plane_surface.hpp:30-33
            }
            else
            {
                // mesh_surface constructor, mesh_surface.hpp:16-48
                out.has_normals = s.has_normals != 0;
                out.has_uvs = s.has_uvs != 0;
                out.prim_count = s.index_count / 3;
                out.positions.resize(3 * static_cast<size_t>(s.vertex_count));
                for(uint32_t v = 0; v < s.vertex_count; ++v)
                {
                    const float* p = d.positions + 3 * static_cast<size_t>(s.vertex_offset + v);
                    V3<double> q = mat_point(mp.t, {p[0], p[1], p[2]});
                    out.positions[3 * v] = static_cast<float>(q.x);
                    out.positions[3 * v + 1] = static_cast<float>(q.y);
                    out.positions[3 * v + 2] = static_cast<float>(q.z);
                }
                if(out.has_normals)
                {
                    out.normals.resize(out.positions.size());
                    for(uint32_t v = 0; v < s.vertex_count; ++v)
                    {
                        const float* n = d.normals + 3 * static_cast<size_t>(s.vertex_offset + v);
                        // prs_transform::transform_normal, transform.hpp:100-106
                        double x = mp.inv.m[0][0] * n[0] + mp.inv.m[1][0] * n[1] + mp.inv.m[2][0] * n[2];
                        double y = mp.inv.m[0][1] * n[0] + mp.inv.m[1][1] * n[1] + mp.inv.m[2][1] * n[2];
                        double z = mp.inv.m[0][2] * n[0] + mp.inv.m[1][2] * n[1] + mp.inv.m[2][2] * n[2];
                        double len = std::sqrt(x * x + y * y + z * z);
                        out.normals[3 * v] = static_cast<float>(x / len);
                        out.normals[3 * v + 1] = static_cast<float>(y / len);
                        out.normals[3 * v + 2] = static_cast<float>(z / len);
                    }
                }
                if(out.has_uvs) out.uvs.assign(d.uvs + 2 * static_cast<size_t>(s.vertex_offset), d.uvs + 2 * static_cast<size_t>(s.vertex_offset + s.vertex_count));
                out.indices.assign(d.indices + s.index_offset, d.indices + s.index_offset + s.index_count);
                for(uint32_t t = 0; t < out.prim_count; ++t) out.area += tri_area(out, t);
            }
            return out;
        }

        static void tri_positions(const Surface& s, uint32_t t, V3<double>& a, V3<double>& b, V3<double>& c)  // mesh_surface.hpp:317-325
        {
            const float* p0 = &s.positions[3 * s.indices[3 * t]];
            const float* p1 = &s.positions[3 * s.indices[3 * t + 1]];
            const float* p2 = &s.positions[3 * s.indices[3 * t + 2]];
            a = {p0[0], p0[1], p0[2]};
            b = {p1[0], p1[1], p1[2]};
            c = {p2[0], p2[1], p2[2]};
        }
        static double tri_area(const Surface& s, uint32_t t)  // mesh_surface.hpp:72-76
        {
            V3<double> a, b, c;
            tri_positions(s, t, a, b, c);
            return 0.5 * length(cross(b - a, c - a));
        }
        static void prepare_for_sampling(Surface& s)  // mesh_surface.hpp:258-269 + distribution.hpp:32-60
        {
            if(s.kind != FCGPU_SURFACE_MESH || !s.cdf.empty()) return;
            size_t n = s.prim_count;
            s.cdf.assign(n + 1, 0.0);
            for(size_t i = 1; i <= n; ++i) s.cdf[i] = s.cdf[i - 1] + tri_area(s, static_cast<uint32_t>(i - 1));
            double integral = s.cdf[n] / static_cast<double>(n);
            if(integral != 0.0)
                for(size_t i = 1; i <= n; ++i) s.cdf[i] /= static_cast<double>(n) * integral;
            else
                for(size_t i = 1; i <= n; ++i) s.cdf[i] = static_cast<double>(i) / static_cast<double>(n);
        }

        BoxF prim_bounds(const Prim& p) const
        {
            const Surface& s = surfaces[entities[p.entity].surface];
            const fcgpu_surface* unused = nullptr;
            (void)unused;
            BoxF b;
            if(s.kind == FCGPU_SURFACE_MESH)  // mesh_surface.hpp:60-65
            {
                grow(b, &s.positions[3 * s.indices[3 * p.index]]);
                grow(b, &s.positions[3 * s.indices[3 * p.index + 1]]);
                grow(b, &s.positions[3 * s.indices[3 * p.index + 2]]);
                return b;
            }
            return surface_box[entities[p.entity].surface];
        }

        std::vector<BoxF> surface_box;  // sphere / plane bounds, filled by set_boxes
        void set_boxes(const fcgpu_scene_desc& d)
        {
            surface_box.resize(d.surface_count);
            for(uint32_t i = 0; i < d.surface_count; ++i)
            {
                const fcgpu_surface& s = d.surfaces[i];
                MatPair mp = make_transform(s.position, s.rotation, nullptr);
                if(s.kind == FCGPU_SURFACE_SPHERE)  // sphere_surface.hpp:20-24
                    surface_box[i] = transformed_bounds(mp.t, {-s.radius, -s.radius, -s.radius}, {s.radius, s.radius, s.radius});
                else if(s.kind == FCGPU_SURFACE_PLANE)  // plane_surface.hpp:19-23
                    surface_box[i] = transformed_bounds(mp.t, {-s.size[0] / 2.0, 0.0, -s.size[1] / 2.0}, {s.size[0] / 2.0, 0.0, s.size[1] / 2.0});
            }
        }

        // ------------------------------------------------------------ BVH build (bvh_acceleration_structure.hpp:228-365)
        struct Info
        {
            uint32_t index;
            BoxF box;
            float centroid[3];
        };
        std::vector<Info> infos_;
        const fcgpu_scene_desc* desc_{};

        void build_bvh() {}  // deferred until set_boxes (needs the description); see finish()

        void finish(const fcgpu_scene_desc& d)
        {
            set_boxes(d);
            infos_.clear();
            for(uint32_t i = 0; i < prims.size(); ++i)
            {
                Info in;
                in.index = i;
                in.box = prim_bounds(prims[i]);
                for(int k = 0; k < 3; ++k) in.centroid[k] = in.box.lo[k] * 0.5f + in.box.hi[k] * 0.5f;  // math.hpp:580-583
                infos_.push_back(in);
            }
            nodes.clear();
            bvh_prims.clear();
            build(0, static_cast<uint32_t>(infos_.size()));
        }

        uint32_t build(uint32_t begin, uint32_t end)
        {
            BoxF box = infos_[begin].box;
            for(uint32_t i = begin + 1; i < end; ++i) grow(box, infos_[i].box);
            if(end - begin == 1) return build_leaf(begin, end, box);
            return build_interior(begin, end, box);
        }
        uint32_t build_leaf(uint32_t begin, uint32_t end, const BoxF& box)
        {
            uint32_t first = static_cast<uint32_t>(bvh_prims.size());
            for(uint32_t i = begin; i < end; ++i) bvh_prims.push_back(prims[infos_[i].index]);
            nodes.push_back(Node{box, first, static_cast<uint16_t>(end - begin), 0});
            return static_cast<uint32_t>(nodes.size() - 1);
        }
        uint32_t build_interior(uint32_t begin, uint32_t end, const BoxF& box)
        {
            BoxF cb;
            for(uint32_t i = begin; i < end; ++i) grow(cb, infos_[i].centroid);
            float dx = cb.hi[0] - cb.lo[0], dy = cb.hi[1] - cb.lo[1], dz = cb.hi[2] - cb.lo[2];
            int axis = (dx > dy && dx > dz) ? 0 : (dy > dz ? 1 : 2);  // math.hpp:590-605
            float axis_len = cb.hi[axis] - cb.lo[axis];
            if(axis_len == 0.0f) return build_leaf(begin, end, box);
            uint32_t count = end - begin;
            uint32_t middle = begin + count / 2;
            if(count <= 4)
            {
                std::nth_element(infos_.begin() + begin, infos_.begin() + middle, infos_.begin() + end,
                                 [axis](const Info& a, const Info& b) { return a.centroid[axis] < b.centroid[axis]; });
            }
            else
            {
                constexpr int NB = 12;
                struct Bucket
                {
                    uint32_t count{};
                    BoxF box{};
                };
                Bucket buckets[NB]{};
                for(uint32_t i = begin; i < end; ++i)
                {
                    float offset = (infos_[i].centroid[axis] - cb.lo[axis]) / axis_len;
                    int bi = std::min(static_cast<int>(offset * NB), NB - 1);
                    buckets[bi].count += 1;
                    grow(buckets[bi].box, infos_[i].box);
                }
                double costs[NB - 1]{};
                for(int i = 0; i < NB - 1; ++i)
                {
                    Bucket b0{}, b1{};
                    for(int j = 0; j <= i; ++j)
                    {
                        grow(b0.box, buckets[j].box);
                        b0.count += buckets[j].count;
                    }
                    for(int j = i + 1; j < NB; ++j)
                    {
                        grow(b1.box, buckets[j].box);
                        b1.count += buckets[j].count;
                    }
                    costs[i] = 0.125 + (b0.count * area(b0.box) + b1.count * area(b1.box)) / area(box);
                }
                double min_cost = costs[0];
                int min_i = 0;
                for(int i = 1; i < NB - 1; ++i)
                    if(costs[i] < min_cost)
                    {
                        min_cost = costs[i];
                        min_i = i;
                    }
                if(min_cost < static_cast<double>(count))
                {
                    float split = cb.lo[axis] + axis_len / NB * (min_i + 1);
                    auto it = std::partition(infos_.begin() + begin, infos_.begin() + end, [axis, split](const Info& a) { return a.centroid[axis] < split; });
                    middle = static_cast<uint32_t>(it - infos_.begin());
                }
                else
                {
                    return build_leaf(begin, end, box);
                }
            }
            uint32_t index = static_cast<uint32_t>(nodes.size());
            nodes.emplace_back();
            build(begin, middle);
            uint32_t right = build(middle, end);
            nodes[index] = Node{box, right, static_cast<uint16_t>(axis), 1};
            return index;
        }

        // ------------------------------------------------------------ primitive intersection
        // returns t (and fills the surface point when `sp` != nullptr)
        bool hit_prim(const Prim& p, V3<R> o, V3<R> d, R t_max, R& t_out, Point<R>* sp) const
        {
            const Entity& e = entities[p.entity];
            const Surface& s = surfaces[e.surface];
            bool ok;
            if(s.kind == FCGPU_SURFACE_SPHERE)
                ok = hit_sphere(s, o, d, t_max, t_out, sp);
            else if(s.kind == FCGPU_SURFACE_PLANE)
                ok = hit_plane(s, o, d, t_max, t_out, sp);
            else
                ok = hit_triangle(s, p.index, o, d, t_max, t_out, sp);
            if(ok && sp)
            {
                sp->entity = p.entity;
                sp->surface = e.surface;
            }
            return ok;
        }

        static bool hit_sphere(const Surface& s, V3<R> ro, V3<R> rd, R t_max, R& t_out, Point<R>* sp)  // sphere_surface.hpp:41-83 / 85-147
        {
            V3<R> o = s.xf.inv_point(ro), d = s.xf.inv_dir(rd);
            R a = dot(d, d), b = R(2) * dot(o, d), c = dot(o, o) - s.radius * s.radius;
            R disc = b * b - R(4) * a * c;
            if(disc < R(0)) return false;
            R sq = std::sqrt(disc);
            R q = b < R(0) ? R(-0.5) * (b - sq) : R(-0.5) * (b + sq);
            R t0 = q / a, t1 = c / q;
            if(t0 > t1) std::swap(t0, t1);
            R t = t0;
            if(t < R(0)) t = t1;
            if(t < R(0) || t > t_max) return false;
            t_out = t;
            if(sp)
            {
                V3<R> pos = o + d * t;
                *sp = Point<R>{};
                sp->position = s.xf.point(pos);
                sp->normal = s.xf.dir(normalize(pos));
                R phi = lm::atan2(pos.z, pos.x);
                if(phi <= R(0)) phi += R(PI) * R(2);
                V3<R> tangent{-lm::sin(phi), R(0), lm::cos(phi)};
                sp->shading_normal = sp->normal;
                sp->tangent = tangent;
                sp->bitangent = cross(tangent, sp->normal);
            }
            return true;
        }

        static bool hit_plane(const Surface& s, V3<R> ro, V3<R> rd, R t_max, R& t_out, Point<R>* sp)  // plane_surface.hpp:40-64 / 66-106
        {
            V3<R> o = s.xf.inv_point(ro), d = s.xf.inv_dir(rd);
            R t = -o.y / d.y;
            if(t < R(0) || t > t_max || std::isinf(t) || std::isnan(t)) return false;
            V3<R> pos = o + d * t;
            R hx = s.size_x / R(2), hy = s.size_y / R(2);
            if(pos.x < -hx || pos.x > hx || pos.z < -hy || pos.z > hy) return false;
            t_out = t;
            if(sp)
            {
                *sp = Point<R>{};
                sp->position = s.xf.point(pos);
                sp->normal = s.xf.dir({R(0), R(1), R(0)});
                sp->u = (pos.x + hx) / s.size_x;
                sp->v = R(1) - (pos.z + hy) / s.size_y;
                sp->shading_normal = sp->normal;
                sp->tangent = s.xf.dir({R(1), R(0), R(0)});
                sp->bitangent = s.xf.dir({R(0), R(0), R(1)});
            }
            return true;
        }

        static V3<R> vtx(const std::vector<float>& a, uint32_t i) { return {R(a[3 * i]), R(a[3 * i + 1]), R(a[3 * i + 2])}; }

        static bool hit_triangle(const Surface& s, uint32_t prim, V3<R> ro, V3<R> rd, R t_max, R& t_out, Point<R>* sp)  // mesh_surface.hpp:78-143 / 145-256
        {
            uint32_t i0 = s.indices[3 * prim], i1 = s.indices[3 * prim + 1], i2 = s.indices[3 * prim + 2];
            V3<R> p0 = vtx(s.positions, i0), p1 = vtx(s.positions, i1), p2 = vtx(s.positions, i2);
            V3<R> p0t = p0 - ro, p1t = p1 - ro, p2t = p2 - ro;
            V3<R> ad{std::abs(rd.x), std::abs(rd.y), std::abs(rd.z)};
            int kz = (ad.x > ad.y && ad.x > ad.z) ? 0 : (ad.y > ad.z ? 1 : 2);
            int kx = kz + 1;
            if(kx == 3) kx = 0;
            int ky = kx + 1;
            if(ky == 3) ky = 0;
            V3<R> d{rd[kx], rd[ky], rd[kz]};
            p0t = {p0t[kx], p0t[ky], p0t[kz]};
            p1t = {p1t[kx], p1t[ky], p1t[kz]};
            p2t = {p2t[kx], p2t[ky], p2t[kz]};
            R sx = -d.x / d.z, sy = -d.y / d.z, sz = R(1) / d.z;
            p0t.x += sx * p0t.z; p0t.y += sy * p0t.z;
            p1t.x += sx * p1t.z; p1t.y += sy * p1t.z;
            p2t.x += sx * p2t.z; p2t.y += sy * p2t.z;
            R e0 = p1t.x * p2t.y - p1t.y * p2t.x;
            R e1 = p2t.x * p0t.y - p2t.y * p0t.x;
            R e2 = p0t.x * p1t.y - p0t.y * p1t.x;
            if((e0 < R(0) || e1 < R(0) || e2 < R(0)) && (e0 > R(0) || e1 > R(0) || e2 > R(0))) return false;
            R det = e0 + e1 + e2;
            if(det == R(0)) return false;
            p0t.z *= sz; p1t.z *= sz; p2t.z *= sz;
            R t_scaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
            if(det < R(0) && (t_scaled >= R(0) || t_scaled < t_max * det)) return false;
            if(det > R(0) && (t_scaled <= R(0) || t_scaled > t_max * det)) return false;
            R inv_det = R(1) / det;
            R b0 = e0 * inv_det, b1 = e1 * inv_det, b2 = e2 * inv_det;
            t_out = t_scaled * inv_det;
            if(sp)
            {
                *sp = Point<R>{};
                sp->position = b0 * p0 + b1 * p1 + b2 * p2;
                V3<R> dp02 = p0 - p2, dp12 = p1 - p2;
                R u0 = 0, v0 = 0, u1 = 1, v1 = 0, u2 = 1, v2 = 1;  // :348-351
                if(s.has_uvs)
                {
                    u0 = R(s.uvs[2 * i0]); v0 = R(s.uvs[2 * i0 + 1]);
                    u1 = R(s.uvs[2 * i1]); v1 = R(s.uvs[2 * i1 + 1]);
                    u2 = R(s.uvs[2 * i2]); v2 = R(s.uvs[2 * i2 + 1]);
                }
                sp->u = b0 * u0 + b1 * u1 + b2 * u2;
                sp->v = b0 * v0 + b1 * v1 + b2 * v2;
                R d02x = u0 - u2, d02y = v0 - v2, d12x = u1 - u2, d12y = v1 - v2;
                R duv_det = d02x * d12y - d02y * d12x;
                V3<R> dpdu = (d12y * dp02 - d02y * dp12) / duv_det;
                sp->normal = normalize(cross(dp02, dp12));
                if(s.has_normals)
                    sp->shading_normal = normalize(b0 * vtx(s.normals, i0) + b1 * vtx(s.normals, i1) + b2 * vtx(s.normals, i2));
                else
                    sp->shading_normal = sp->normal;
                V3<R> tangent = normalize(dpdu);
                V3<R> bitangent = cross(tangent, sp->shading_normal);
                tangent = cross(sp->shading_normal, bitangent);
                sp->tangent = tangent;
                sp->bitangent = bitangent;
            }
            return true;
        }

        // ------------------------------------------------------------ acceleration structures
        static bool box_hit(const BoxF& bx, V3<R> o, R t_max, V3<R> inv, const int neg[3])  // math.hpp:671-689
        {
            R lo[3]{R(bx.lo[0]), R(bx.lo[1]), R(bx.lo[2])}, hi[3]{R(bx.hi[0]), R(bx.hi[1]), R(bx.hi[2])};
            auto pick = [&](int side, int axis) { return side ? hi[axis] : lo[axis]; };
            R t0 = (pick(neg[0], 0) - o.x) * inv.x, t1 = (pick(1 - neg[0], 0) - o.x) * inv.x;
            R ty0 = (pick(neg[1], 1) - o.y) * inv.y, ty1 = (pick(1 - neg[1], 1) - o.y) * inv.y;
            if(t0 > ty1 || ty0 > t1) return false;
            if(ty0 > t0) t0 = ty0;
            if(ty1 < t1) t1 = ty1;
            R tz0 = (pick(neg[2], 2) - o.z) * inv.z, tz1 = (pick(1 - neg[2], 2) - o.z) * inv.z;
            if(t0 > tz1 || tz0 > t1) return false;
            if(tz0 > t0) t0 = tz0;
            if(tz1 < t1) t1 = tz1;
            return (t0 < t_max) && (t1 > R(0));
        }

        // closest hit, BVH (:31-91) or brute force (brute_force_acceleration_structure.hpp:25-49)
        bool closest(V3<R> o, V3<R> d, R t_max, bool brute, Point<R>& out, uint32_t& prim_id, R& t_hit) const
        {
            bool found = false;
            Point<R> sp;
            if(brute)
            {
                for(uint32_t i = 0; i < prims.size(); ++i)
                {
                    R t;
                    if(hit_prim(prims[i], o, d, t_max, t, &sp))
                    {
                        t_max = t;
                        out = sp;
                        prim_id = i;
                        found = true;
                    }
                }
                t_hit = t_max;
                return found;
            }
            V3<R> inv{R(1) / d.x, R(1) / d.y, R(1) / d.z};
            int neg[3]{inv.x < 0, inv.y < 0, inv.z < 0};
            uint32_t stack[64];
            int n = 0;
            stack[n++] = 0;
            while(n > 0)
            {
                uint32_t ni = stack[--n];
                const Node& node = nodes[ni];
                if(!box_hit(node.box, o, t_max, inv, neg)) continue;
                if(!node.interior)
                {
                    for(uint32_t i = node.a; i < node.a + node.b; ++i)
                    {
                        R t;
                        if(hit_prim(bvh_prims[i], o, d, t_max, t, &sp))
                        {
                            t_max = t;
                            out = sp;
                            prim_id = entities[bvh_prims[i].entity].first_prim + bvh_prims[i].index;
                            found = true;
                        }
                    }
                }
                else if(neg[node.b])
                {
                    stack[n++] = ni + 1;
                    stack[n++] = node.a;
                }
                else
                {
                    stack[n++] = node.a;
                    stack[n++] = ni + 1;
                }
            }
            t_hit = t_max;
            return found;
        }

        bool any(V3<R> o, V3<R> d, R t_max, bool brute) const  // :93-138 / brute :51-63
        {
            R t;
            if(brute)
            {
                for(const Prim& p : prims)
                    if(hit_prim(p, o, d, t_max, t, nullptr)) return true;
                return false;
            }
            V3<R> inv{R(1) / d.x, R(1) / d.y, R(1) / d.z};
            int neg[3]{inv.x < 0, inv.y < 0, inv.z < 0};
            uint32_t stack[64];
            int n = 0;
            stack[n++] = 0;
            while(n > 0)
            {
                uint32_t ni = stack[--n];
                const Node& node = nodes[ni];
                if(!box_hit(node.box, o, t_max, inv, neg)) continue;
                if(!node.interior)
                {
                    for(uint32_t i = node.a; i < node.a + node.b; ++i)
                        if(hit_prim(bvh_prims[i], o, d, t_max, t, nullptr)) return true;
                }
                else if(neg[node.b])
                {
                    stack[n++] = ni + 1;
                    stack[n++] = node.a;
                }
                else
                {
                    stack[n++] = node.a;
                    stack[n++] = ni + 1;
                }
            }
            return false;
        }
    };

    // ================================================================ BSDF layer
    template<typename R> R fr_dielectric(R cos_i, R eta_i, R eta_t)  // common.hpp:8-20
    {
        cos_i = std::clamp(cos_i, R(-1), R(1));
        R sin_i = std::sqrt(std::max(R(0), R(1) - cos_i * cos_i));
        R sin_t = eta_i / eta_t * sin_i;
        if(sin_t >= R(1)) return R(1);
        R cos_t = std::sqrt(std::max(R(0), R(1) - sin_t * sin_t));
        R r_parl = ((eta_t * cos_i) - (eta_i * cos_t)) / ((eta_t * cos_i) + (eta_i * cos_t));
        R r_perp = ((eta_i * cos_i) - (eta_t * cos_t)) / ((eta_i * cos_i) + (eta_t * cos_t));
        return (r_parl * r_parl + r_perp * r_perp) / R(2);
    }
    template<typename R> V3<R> reflect(V3<R> w, V3<R> n) { return -w + R(2) * dot(w, n) * n; }  // common.hpp:22-25
    template<typename R> bool refract(V3<R> i, V3<R> m, R eta, V3<R>* o)                        // common.hpp:27-39
    {
        R cos_i = dot(i, m);
        R sin2_i = std::max(R(0), R(1) - cos_i * cos_i);
        R sin2_t = eta * eta * sin2_i;
        if(sin2_t >= R(1)) return false;
        R cos_t = std::sqrt(R(1) - sin2_t);
        *o = eta * -i + (eta * cos_i - cos_t) * m;
        return true;
    }

    template<typename R> void disk_concentric(R ux, R uy, R& x, R& y)  // sampling.hpp:6-29
    {
        R ox = ux * R(2) - R(1), oy = uy * R(2) - R(1);
        if(ox == R(0) && oy == R(0))
        {
            x = y = R(0);
            return;
        }
        R theta, r;
        if(std::abs(ox) > std::abs(oy))
        {
            r = ox;
            theta = R(PI) * oy / (R(4) * ox);
        }
        else
        {
            r = oy;
            theta = R(PI) / R(2) - R(PI) * ox / (R(4) * oy);
        }
        x = r * lm::cos(theta);
        y = r * lm::sin(theta);
    }
    template<typename R> V3<R> hemisphere_cosine(R ux, R uy)  // sampling.hpp:31-35
    {
        R x, y;
        disk_concentric(ux, uy, x, y);
        return {x, std::sqrt(std::max(R(0), R(1) - x * x - y * y)), y};
    }
    template<typename R> V3<R> sphere_uniform(R ux, R uy)  // sampling.hpp:49-55
    {
        R z = R(1) - R(2) * ux;
        R r = std::sqrt(std::max(R(0), R(1) - z * z));
        R phi = R(2) * R(PI) * uy;
        return {r * lm::cos(phi), z, r * lm::sin(phi)};
    }

    // smith_ggx_microfacet_model, microfacet.hpp:21-82
    template<typename R>
    struct Ggx
    {
        R ax, ay;
        V3<R> sample(V3<R> i, R ux, R uy) const
        {
            V3<R> ih = normalize(V3<R>{ax * i.x, i.y, ay * i.z});
            R lensq = ih.x * ih.x + ih.z * ih.z;
            V3<R> T1 = lensq > R(0) ? V3<R>{-ih.z, R(0), ih.x} / std::sqrt(lensq) : V3<R>{R(1), R(0), R(0)};
            V3<R> T2 = cross(T1, ih);
            R r = std::sqrt(ux);
            R phi = R(2) * R(PI) * uy;
            R t1 = r * lm::cos(phi), t2 = r * lm::sin(phi);
            R s = R(0.5) * (R(1) + ih.y);
            t2 = (R(1) - s) * std::sqrt(R(1) - t1 * t1) + s * t2;
            V3<R> Nh = t1 * T1 + t2 * T2 + std::sqrt(std::max(R(0), R(1) - t1 * t1 - t2 * t2)) * ih;
            return normalize(V3<R>{ax * Nh.x, std::max(R(0), Nh.y), ay * Nh.z});
        }
        R distribution(V3<R> m) const
        {
            R x = m.x * m.x / (ax * ax) + m.y * m.y + m.z * m.z / (ay * ay);
            return R(1) / (R(PI) * ax * ay * x * x);
        }
        R lambda(V3<R> i) const
        {
            R x = (ax * ax * i.x * i.x + ay * ay * i.z * i.z) / (i.y * i.y);
            return (R(-1) + std::sqrt(R(1) + x)) / R(2);
        }
        R masking(V3<R> i) const { return R(1) / (R(1) + lambda(i)); }
        R masking(V3<R> i, V3<R> o) const { return R(1) / (R(1) + lambda(i) + lambda(o)); }
        R pdf(V3<R> i, V3<R> m) const { return masking(i) * std::max(R(0), dot(i, m)) * distribution(m) / i.y; }
    };

    template<typename R> R lobe_fresnel(const Lobe<R>& L, R cos_i, R eta_a)  // microfacet_reflection.hpp:18-35
    {
        return L.fresnel_dielectric ? fr_dielectric(cos_i, eta_a, L.ior) : R(1);
    }

    // transmission half vector with the orientation tests shared by microfacet_glass.hpp:38-51 and
    // microfacet_transmission.hpp:25-38
    template<typename R> bool transmission_half(V3<R> i, V3<R> o, R ea, R eb, V3<R>& h, R& idh, R& odh)
    {
        h = normalize(-(ea * i + eb * o));
        if(ea <= eb)
        {
            if(h.y <= R(0)) return false;
        }
        else
        {
            if(h.y >= R(0)) return false;
            h = -h;
        }
        idh = dot(i, h);
        odh = dot(o, h);
        if(idh <= R(0) || odh >= R(0)) return false;
        return true;
    }

    template<typename R> V3<R> lobe_evaluate(const Lobe<R>& L, V3<R> i, V3<R> o, R ea, R eb)
    {
        Ggx<R> g{L.alpha_x, L.alpha_y};
        switch(L.kind)
        {
        case LAMBERT:  // lambertian_reflection.hpp:20-24
            if(i.y <= R(0) || o.y <= R(0)) return {};
            return L.reflectance * R(1.0 / PI);
        case GGX_REFLECTION:  // microfacet_reflection.hpp:49-59
        {
            if(o.y <= R(0)) return {};
            V3<R> h = normalize(i + o);
            R G = g.masking(i, o), D = g.distribution(h);
            R F = lobe_fresnel(L, dot(i, h), ea);
            return L.reflectance * F * (G * D / (R(4) * i.y * o.y));
        }
        case GGX_GLASS:  // microfacet_glass.hpp:22-61
        {
            if(i.y == R(0) || o.y == R(0)) return {};
            if(o.y > R(0))
            {
                V3<R> h = normalize(i + o);
                R G = g.masking(i, o), D = g.distribution(h);
                R F = fr_dielectric(dot(i, h), ea, eb);
                return L.reflectance * (G * D * F / (R(4) * i.y * o.y));
            }
            V3<R> h;
            R idh, odh;
            if(!transmission_half(i, o, ea, eb, h, idh, odh)) return {};
            R eta = ea / eb;
            R J = -odh / ((eta * idh + odh) * (eta * idh + odh));
            R G = g.masking(i, o), D = g.distribution(h);
            R F = fr_dielectric(idh, ea, eb);
            return L.transmittance * (idh * G * D * J * (R(1) - F) / (i.y * -o.y));
        }
        case GGX_TRANSMISSION:  // microfacet_transmission.hpp:21-46
        {
            if(o.y >= R(0)) return {};
            V3<R> h;
            R idh, odh;
            if(!transmission_half(i, o, ea, eb, h, idh, odh)) return {};
            R eta = ea / eb;
            R J = -odh / ((eta * idh + odh) * (eta * idh + odh));
            return L.transmittance * (idh * g.masking(i, o) * g.distribution(h) * J / (i.y * -o.y));
        }
        default:  // delta lobes evaluate to zero (specular_*.hpp)
            return {};
        }
    }

    template<typename R> R lobe_pdf(const Lobe<R>& L, V3<R> i, V3<R> o, R ea, R eb)
    {
        Ggx<R> g{L.alpha_x, L.alpha_y};
        switch(L.kind)
        {
        case LAMBERT:  // lambertian_reflection.hpp:42-46
            if(i.y <= R(0) || o.y <= R(0)) return R(0);
            return o.y * R(1.0 / PI);
        case GGX_REFLECTION:  // microfacet_reflection.hpp:88-95
        {
            if(o.y <= R(0)) return R(0);
            V3<R> h = normalize(i + o);
            R J = R(1) / (R(4) * dot(i, h));
            return g.pdf(i, h) * J;
        }
        case GGX_GLASS:  // microfacet_glass.hpp:123-160
        {
            if(i.y == R(0) || o.y == R(0)) return R(0);
            if(o.y > R(0))
            {
                V3<R> h = normalize(i + o);
                R idh = dot(i, h);
                R F = fr_dielectric(idh, ea, eb);
                R J = R(1) / (R(4) * dot(i, h));
                return g.pdf(i, h) * J * F;
            }
            V3<R> h;
            R idh, odh;
            if(!transmission_half(i, o, ea, eb, h, idh, odh)) return R(0);
            R eta = ea / eb;
            R F = fr_dielectric(idh, ea, eb);
            R J = -odh / ((eta * idh + odh) * (eta * idh + odh));
            return g.pdf(i, h) * J * (R(1) - F);
        }
        case GGX_TRANSMISSION:  // microfacet_transmission.hpp:85-108
        {
            if(o.y >= R(0)) return R(0);
            V3<R> h;
            R idh, odh;
            if(!transmission_half(i, o, ea, eb, h, idh, odh)) return R(0);
            R eta = ea / eb;
            R J = -odh / ((eta * idh + odh) * (eta * idh + odh));
            return g.pdf(i, h) * J;
        }
        default:
            return R(0);
        }
    }

    template<typename R> bool lobe_sample(const Lobe<R>& L, V3<R> i, R ea, R eb, R u1x, R u1y, R u2x, R u2y, V3<R>* o, V3<R>* value, R* pdf_o)
    {
        (void)u2y;
        Ggx<R> g{L.alpha_x, L.alpha_y};
        switch(L.kind)
        {
        case LAMBERT:  // lambertian_reflection.hpp:26-40
            if(i.y <= R(0)) return false;
            *o = hemisphere_cosine(u1x, u1y);
            if(o->y == R(0)) return false;
            *value = L.reflectance * R(1.0 / PI);
            *pdf_o = o->y * R(1.0 / PI);
            return true;
        case GGX_REFLECTION:  // microfacet_reflection.hpp:61-86
        {
            if(i.y == R(0)) return false;
            V3<R> h = g.sample(i, u1x, u1y);
            R idh = dot(i, h);
            if(idh <= R(0)) return false;
            *o = reflect(i, h);
            if(o->y <= R(0)) return false;
            R G = g.masking(i, *o), D = g.distribution(h);
            R F = lobe_fresnel(L, idh, ea);
            *value = L.reflectance * F * (G * D / (R(4) * i.y * o->y));
            *pdf_o = g.pdf(i, h) * (R(1) / (R(4) * idh));
            return true;
        }
        case SPECULAR_REFLECTION:  // specular_reflection.hpp:25-40
        {
            if(i.y == R(0)) return false;
            R F = lobe_fresnel(L, i.y, ea);
            *o = {-i.x, i.y, -i.z};
            *value = F * L.reflectance / o->y;
            *pdf_o = R(1);
            return true;
        }
        case GGX_GLASS:  // microfacet_glass.hpp:63-121
        {
            if(i.y == R(0)) return false;
            V3<R> h = g.sample(i, u1x, u1y);
            R idh = dot(i, h);
            if(idh <= R(0)) return false;
            R F = fr_dielectric(idh, ea, eb);
            if(u2x < F)
            {
                *o = reflect(i, h);
                if(o->y <= R(0)) return false;
                R G = g.masking(i, *o), D = g.distribution(h);
                *value = L.reflectance * (G * D * F / (R(4) * i.y * o->y));
                *pdf_o = g.pdf(i, h) * (R(1) / (R(4) * idh)) * F;
                return true;
            }
            R eta = ea / eb;
            if(!refract(i, h, eta, o)) return false;
            if(o->y >= R(0)) return false;
            R odh = dot(*o, h);
            R J = -odh / ((eta * idh + odh) * (eta * idh + odh));
            R G = g.masking(i, *o), D = g.distribution(h);
            *value = L.transmittance * (idh * G * D * J * (R(1) - F) / (i.y * -o->y));
            *pdf_o = g.pdf(i, h) * J * (R(1) - F);
            return true;
        }
        case SPECULAR_GLASS:  // specular_glass.hpp:25-74
        {
            R cos_i = i.y;
            R sin_i = std::sqrt(R(1) - cos_i * cos_i);
            R eta = ea / eb;
            R sin_t = eta * sin_i;
            R cos_t = std::sqrt(R(1) - sin_t * sin_t);
            R F = R(1);
            if(sin_t < R(1))
            {
                R r_parl = ((eb * cos_i) - (ea * cos_t)) / ((eb * cos_i) + (ea * cos_t));
                R r_perp = ((ea * cos_i) - (eb * cos_t)) / ((ea * cos_i) + (eb * cos_t));
                F = (r_parl * r_parl + r_perp * r_perp) / R(2);
            }
            if(u1x < F)
            {
                *o = {-i.x, i.y, -i.z};
                *value = L.reflectance * (F / o->y);
                *pdf_o = F;
                return true;
            }
            *o = eta * -i;
            o->y += eta * cos_i - cos_t;
            *value = L.transmittance * ((R(1) - F) * (eb * eb) / (ea * ea * -o->y));
            *pdf_o = R(1) - F;
            return true;
        }
        case GGX_TRANSMISSION:  // microfacet_transmission.hpp:48-83
        {
            if(i.y == R(0)) return false;
            V3<R> h = g.sample(i, u1x, u1y);
            R idh = dot(i, h);
            if(idh <= R(0)) return false;
            R eta = ea / eb;
            if(!refract(i, h, eta, o)) return false;
            if(o->y >= R(0)) return false;
            R odh = dot(*o, h);
            R J = -odh / ((eta * idh + odh) * (eta * idh + odh));
            *value = L.transmittance * (idh * g.masking(i, *o) * g.distribution(h) * J / (i.y * -o->y));
            *pdf_o = g.pdf(i, h) * J;
            return true;
        }
        case SPECULAR_TRANSMISSION:  // specular_transmission.hpp:25-37
            if(!refract(i, V3<R>{R(0), R(1), R(0)}, ea / eb, o)) return false;
            *value = L.transmittance * ((eb * eb) / (ea * ea * -o->y));
            *pdf_o = R(1);
            return true;
        }
        return false;
    }

    // material::evaluate for each class (const textures; normal_mapping on its skip_ path)
    // texture_2d_rgb::evaluate of a material parameter: image_texture_2d_rgb (image_texture.hpp:22-33) or
    // checker_texture_2d_rgb (checker_texture.hpp:14-20)
    template<typename R> V3<R> texture_evaluate(const Scene<R>& scene, const fcgpu_texture& t, R u, R v)
    {
        if(t.kind == FCGPU_TEXTURE_CHECKER)
        {
            int x = static_cast<int>(std::floor(u * R(t.scale))), y = static_cast<int>(std::floor(v * R(t.scale)));
            const double* c = (x + y) % 2 == 0 ? t.color_a : t.color_b;
            return V3<R>{R(c[0]), R(c[1]), R(c[2])};
        }
        return image_evaluate<R>(scene.images[static_cast<size_t>(t.image)], t.filter, u, v);
    }

    // Image textures (image_texture_2d_rgb / _r) are evaluated at p.get_uv() like each evaluate() does.
    template<typename R> Bsdf<R> material_bsdf(const fcgpu_material& m_in, const Point<R>& p, const Scene<R>* scene = nullptr)
    {
        fcgpu_material m = m_in;
        if(scene && !scene->textures.empty())
        {
            auto tex = [&](int32_t handle) { return texture_evaluate<R>(*scene, scene->textures[static_cast<size_t>(handle - 1)], p.u, p.v); };
            auto set3 = [](double* dst, V3<R> c) {
                dst[0] = static_cast<double>(c.x);
                dst[1] = static_cast<double>(c.y);
                dst[2] = static_cast<double>(c.z);
            };
            if(m.tex_color_a > 0) set3(m.color_a, tex(m.tex_color_a));
            if(m.tex_color_b > 0) set3(m.color_b, tex(m.tex_color_b));
            if(m.tex_roughness > 0) m.roughness[0] = m.roughness[1] = static_cast<double>(tex(m.tex_roughness).x);
            if(m.tex_metalness > 0) m.metalness = static_cast<double>(tex(m.tex_metalness).x);
            if(m.tex_ior > 0) m.ior = static_cast<double>(tex(m.tex_ior).x);
        }
        // the normal map of diffuse / mirror / standard materials (diffuse_material.hpp:20-27, mirror_material.hpp:29-36,
        // standard_material.hpp:39-46) and normal_mapping's precompute (normal_mapping.hpp:743-760)
        NormalMap<R> nm{};
        if(scene && m.tex_normal > 0)
        {
            V3<R> c = texture_evaluate<R>(*scene, scene->textures[static_cast<size_t>(m.tex_normal - 1)], p.u, p.v);
            V3<R> n = c * R(2) - V3<R>{R(1), R(1), R(1)};
            std::swap(n.y, n.z);
            n = normalize(n);
            if(n.y < R(0)) n = -n;
            if(!(std::abs(n.x) < R(0.001) && std::abs(n.z) < R(0.001)))
            {
                nm.on = true;
                nm.p = n;
                nm.t = normalize(V3<R>{-n.x, R(0), -n.z});
                nm.bitangent = normalize(cross(V3<R>{R(1), R(0), R(0)}, n));
                nm.tangent = cross(n, nm.bitangent);
            }
        }
        const bool maps_normals = m.kind == FCGPU_MATERIAL_DIFFUSE || m.kind == FCGPU_MATERIAL_MIRROR || m.kind == FCGPU_MATERIAL_STANDARD;
        Bsdf<R> b;
        b.t = p.tangent;
        b.ns = p.shading_normal;
        b.b = p.bitangent;
        b.ng = p.normal;
        double scales[4]{}, weights[4]{};
        auto rgb = [](const double* c) { return V3<R>{R(c[0]), R(c[1]), R(c[2])}; };
        auto alpha = [](double r)  // microfacet.hpp:76-81
        {
            double x = std::max(r, 0.002);
            return R(x * x);
        };
        auto add = [&](LobeKind kind, bool fd, V3<R> refl, V3<R> trans, double rx, double ry, double ior, double scale, double weight)
        {
            Lobe<R>& L = b.lobes[b.count];
            if(maps_normals) L.nm = nm;
            L.kind = kind;
            L.fresnel_dielectric = fd;
            L.reflectance = refl;
            L.transmittance = trans;
            L.alpha_x = alpha(rx);
            L.alpha_y = alpha(ry);
            L.ior = R(ior);
            scales[b.count] = scale;
            weights[b.count] = weight;
            b.count += 1;
        };
        bool smooth = m.roughness[0] == 0.0 && m.roughness[1] == 0.0;
        V3<R> one{R(1), R(1), R(1)};
        switch(m.kind)
        {
        case FCGPU_MATERIAL_DIFFUSE:  // diffuse_material.hpp:18-38
            add(LAMBERT, false, rgb(m.color_a), {}, 0, 0, 0, 1.0, 1.0);
            break;
        case FCGPU_MATERIAL_MIRROR:  // mirror_material.hpp:20-56
            add(smooth ? SPECULAR_REFLECTION : GGX_REFLECTION, false, rgb(m.color_a), {}, m.roughness[0], m.roughness[1], 0.0, 1.0, 1.0);
            break;
        case FCGPU_MATERIAL_GLASS:  // glass_material.hpp:19-41
            add(smooth ? SPECULAR_GLASS : GGX_GLASS, false, rgb(m.color_a), rgb(m.color_b), m.roughness[0], m.roughness[1], 0.0, 1.0, 1.0);
            break;
        case FCGPU_MATERIAL_PLASTIC:  // plastic_material.hpp:24-52
            add(LAMBERT, false, rgb(m.color_a), {}, 0, 0, 0, 1.0, 1.0);
            add(smooth ? SPECULAR_REFLECTION : GGX_REFLECTION, true, rgb(m.color_b), {}, m.roughness[0], m.roughness[1], m.ior, 1.0, 1.0);
            break;
        case FCGPU_MATERIAL_STANDARD:  // standard_material.hpp:29-107
        {
            bool sm = m.roughness[0] == 0.0;
            if(m.metalness < 1.0)
            {
                double s = 1.0 - m.metalness, w = (1.0 - m.metalness) / 2.0;
                add(LAMBERT, false, rgb(m.color_a), {}, 0, 0, 0, s, w);
                add(sm ? SPECULAR_REFLECTION : GGX_REFLECTION, true, one, {}, m.roughness[0], m.roughness[0], m.ior, s, w);
            }
            if(m.metalness > 0.0)
                add(sm ? SPECULAR_REFLECTION : GGX_REFLECTION, false, rgb(m.color_a), {}, m.roughness[0], m.roughness[0], 0.0, m.metalness, m.metalness);
            break;
        }
        case FCGPU_MATERIAL_TRANSMISSION:  // transmission_material.hpp:19-40
            add(smooth ? SPECULAR_TRANSMISSION : GGX_TRANSMISSION, false, {}, rgb(m.color_b), m.roughness[0], m.roughness[1], 0.0, 1.0, 1.0);
            break;
        }
        // bsdf constructor, bsdf.hpp:30-43 (in double, as the reference)
        double sum = 0.0;
        for(int i = 0; i < b.count; ++i) sum += weights[i];
        double pdf[4]{}, w[4]{};
        for(int i = 0; i < b.count; ++i) pdf[i] = weights[i] / sum;
        w[0] = 0.0;
        for(int i = 1; i < b.count; ++i) w[i] += w[i - 1] + pdf[i - 1];
        for(int i = 0; i < b.count; ++i)
        {
            b.lobes[i].scale = R(scales[i]);
            b.lobes[i].pdf = R(pdf[i]);
            b.lobes[i].weight = R(w[i]);
        }
        return b;
    }

    // ---------------------------------------------------------------- normal_mapping<T>, bsdfs/normal_mapping.hpp
    // Light travels i -> (facets) -> o along at most three routes (second-order scattering):
    //   P  : meets the mapped facet, scatters by the inner bxdf straight to o            f(i, o)
    //   TP : meets the tangent facet first, is mirrored (ri), then scatters on p to o     f(ri, o)
    //   PT : scatters on p towards ro, the mirror image of o, and the tangent facet turns it into o   f(i, ro)
    // evaluate (:32-145), pdf (:629-731) and sample (:147-626) enumerate the 3x3 facet classes of (i, o); the sums
    // below are those tables written once. `Known` marks the inner pair that sample() already holds (delta inner
    // lobes have no evaluate(), and the reference reuses the sampled value for that term as well).
    enum NmKnown { NM_NONE, NM_I_O, NM_RI_O, NM_I_RO };
    template<typename R>
    void nm_combine(const Lobe<R>& L, V3<R> i, V3<R> o, int co, R ea, R eb, NmKnown known, V3<R> f_known, R pdf_known, V3<R>* value, R* pdf)
    {
        const NormalMap<R>& N = L.nm;
        const int ci = N.facets(i);
        *value = V3<R>{};
        *pdf = R(0);
        if(ci == 0 && co == 0) return;
        const R lambda = ci == 0 ? R(0) : ci == 2 ? R(1) : N.hits_p(i);
        auto inner = [&](NmKnown which, V3<R> a, V3<R> b, V3<R>& f, R& q) {
            if(which == known)
            {
                f = f_known;
                q = pdf_known;
                return;
            }
            V3<R> la = N.to_p(a), lb = N.to_p(b);
            f = lobe_evaluate(L, la, lb, ea, eb);
            q = lobe_pdf(L, la, lb, ea, eb);
        };
        V3<R> sum{};
        R q_sum = R(0);
        if(co != 0)  // o leaves the mapped facet directly: routes P and TP
        {
            const R open = co == 2 ? N.open_p(o) : R(1);
            const R o_dot_p = dot(o, N.p);
            V3<R> f;
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
            V3<R> ro = N.mirror(o);
            const R ro_dot_p = dot(ro, N.p);
            const R hidden = R(1) - N.open_p(ro);
            V3<R> f;
            R q;
            inner(NM_I_RO, i, ro, f, q);
            sum = sum + f * (lambda * hidden * ro_dot_p * (co == 0 ? N.open_t(o) : R(1)));
            q_sum += lambda * q * hidden;
        }
        *value = sum / o.y;
        *pdf = q_sum;
    }
    template<typename R> V3<R> mapped_evaluate(const Lobe<R>& L, V3<R> i, V3<R> o, R ea, R eb)
    {
        if(!L.nm.on) return lobe_evaluate(L, i, o, ea, eb);
        V3<R> f;
        R q;
        nm_combine(L, i, o, L.nm.facets(o), ea, eb, NM_NONE, V3<R>{}, R(0), &f, &q);
        return f;
    }
    template<typename R> R mapped_pdf(const Lobe<R>& L, V3<R> i, V3<R> o, R ea, R eb)
    {
        if(!L.nm.on) return lobe_pdf(L, i, o, ea, eb);
        V3<R> f;
        R q;
        nm_combine(L, i, o, L.nm.facets(o), ea, eb, NM_NONE, V3<R>{}, R(0), &f, &q);
        return q;
    }
    template<typename R> bool mapped_sample(const Lobe<R>& L, V3<R> i, R ea, R eb, R u1x, R u1y, R u2x, R u2y, V3<R>* o, V3<R>* value, R* pdf_o)
    {
        if(!L.nm.on) return lobe_sample(L, i, ea, eb, u1x, u1y, u2x, u2y, o, value, pdf_o);
        const NormalMap<R>& N = L.nm;
        const int ci = N.facets(i);
        // first encounter: the tangent facet when i cannot see p, by the projected areas when it sees both (:232)
        const bool via_t = ci == 0 || (ci == 1 && !(u2x < N.hits_p(i)));
        const V3<R> start = via_t ? N.mirror(i) : i;
        V3<R> local_m, f_m;
        R pdf_m;
        if(!lobe_sample(L, N.to_p(start), ea, eb, u1x, u1y, u2x, u2y, &local_m, &f_m, &pdf_m)) return false;
        V3<R> m = N.from_p(local_m);
        if(m.y <= R(0)) return false;
        const int cm = dot(m, N.t) > R(0) ? 1 : 2;
        NmKnown known = via_t ? NM_RI_O : NM_I_O;
        int co = cm;
        *o = m;
        if(!via_t && cm == 2 && u2y > N.open_p(m))  // leaving p behind the tangent facet: mirrored once more (:306-372, :551-603)
        {
            *o = N.mirror(m);
            co = dot(*o, N.p) > R(0) ? 1 : 0;
            known = NM_I_RO;
        }
        nm_combine(L, i, *o, co, ea, eb, known, f_m, pdf_m, value, pdf_o);
        return true;
    }

    template<typename R> int sample_bxdf(const Bsdf<R>& b, R u)  // bsdf.hpp:46-55
    {
        for(int i = b.count - 1; i >= 0; --i)
            if(u > b.lobes[i].weight) return i;
        return 0;
    }
    template<typename R> V3<R> to_local(const Bsdf<R>& b, V3<R> w) { return {dot(w, b.t), dot(w, b.ns), dot(w, b.b)}; }  // bsdf.hpp:156-163
    template<typename R> V3<R> to_world(const Bsdf<R>& b, V3<R> w)                                                          // bsdf.hpp:165-172
    {
        return {b.t.x * w.x + b.ns.x * w.y + b.b.x * w.z, b.t.y * w.x + b.ns.y * w.y + b.b.y * w.z, b.t.z * w.x + b.ns.z * w.y + b.b.z * w.z};
    }

    template<typename R> V3<R> bsdf_evaluate(const Bsdf<R>& b, int l, V3<R> wo, V3<R> wi, R ea, R eb)  // bsdf.hpp:62-72 + bxdf.hpp:57-67
    {
        R wo_wg = dot(wo, b.ng), wo_ws = dot(wo, b.ns), wi_wg = dot(wi, b.ng), wi_ws = dot(wi, b.ns);
        if(wo_wg * wo_ws <= R(0) || wi_wg * wi_ws <= R(0)) return {};
        const Lobe<R>& L = b.lobes[l];
        R c = std::abs(wi_ws) * L.scale / (std::abs(wi_wg) * L.pdf);
        V3<R> lo = to_local(b, wo), li = to_local(b, wi);
        V3<R> f = li.y >= R(0) ? mapped_evaluate(L, li, lo, ea, eb) : mapped_evaluate(L, -li, -lo, eb, ea);
        return c * f;
    }
    template<typename R> R bsdf_pdf_wi(const Bsdf<R>& b, int l, V3<R> wo, V3<R> wi, R ea, R eb)  // bsdf.hpp:131-141 + bxdf.hpp:117-127
    {
        R wo_wg = dot(wo, b.ng), wo_ws = dot(wo, b.ns), wi_wg = dot(wi, b.ng), wi_ws = dot(wi, b.ns);
        if(wo_wg * wo_ws <= R(0) || wi_wg * wi_ws <= R(0)) return R(0);
        V3<R> lo = to_local(b, wo), li = to_local(b, wi);
        return lo.y >= R(0) ? mapped_pdf(b.lobes[l], lo, li, ea, eb) : mapped_pdf(b.lobes[l], -lo, -li, eb, ea);
    }
    template<typename R>
    bool bsdf_sample_wi(const Bsdf<R>& b, int l, V3<R> wo, R ea, R eb, R u1x, R u1y, R u2x, R u2y, V3<R>* wi, V3<R>* value, R* pdf)  // bsdf.hpp:74-100 + bxdf.hpp:69-97
    {
        R wo_wg = dot(wo, b.ng), wo_ws = dot(wo, b.ns);
        if(wo_wg * wo_ws <= R(0)) return false;
        const Lobe<R>& L = b.lobes[l];
        V3<R> lo = to_local(b, wo), li;
        if(lo.y >= R(0))
        {
            if(!mapped_sample(L, lo, ea, eb, u1x, u1y, u2x, u2y, &li, value, pdf)) return false;
            if(li.y <= R(0)) *value = *value * ((ea * ea) / (eb * eb));
        }
        else
        {
            if(!mapped_sample(L, -lo, eb, ea, u1x, u1y, u2x, u2y, &li, value, pdf)) return false;
            li = -li;
            if(li.y >= R(0)) *value = *value * ((eb * eb) / (ea * ea));
        }
        *wi = to_world(b, li);
        R wi_wg = dot(*wi, b.ng), wi_ws = dot(*wi, b.ns);
        if(wi_wg * wi_ws <= R(0)) return false;
        *value = *value * (std::abs(wi_ws) * L.scale / (std::abs(wi_wg) * L.pdf));
        return true;
    }

    // ================================================================ integrators
    struct ExternalSampler  // fc::sampler::get / advance_dimension of a host, as plain callbacks
    {
        void (*get)(void* user, double* x, double* y);
        void (*advance)(void* user, int count);
        void* user;
    };

    template<typename R>
    class Tracer
    {
    public:
        const Scene<R>& S;
        bool brute{};
        uint64_t closest_rays{}, any_rays{}, any_occluded{};
        std::vector<RayRecord>* log{};

        explicit Tracer(const Scene<R>& s) : S{s} {}

        // counter-based sampler behind the fc::sampler protocol (sampler.hpp:15-24)
        fcgpu_sampler_config sampler{fcgpu_sampler_random(0)};
        fcgpu_sample_stream stream{};
        uint32_t dim{};
        void set_sample(const fcgpu_sampler_config& cfg, int px, int py, uint32_t sample)
        {
            sampler = cfg;
            stream = fcgpu_stream_begin(&sampler, static_cast<uint32_t>(px), static_cast<uint32_t>(py), sample);
            dim = 0;
        }
        // ... or any fc::sampler handed in by a host (run_once_from below): get() / advance_dimension() forwarded
        const ExternalSampler* external{};
        void get(R& x, R& y)
        {
            if(external)
            {
                double ex, ey;
                external->get(external->user, &ex, &ey);
                x = R(ex);
                y = R(ey);
                return;
            }
            uint64_t bits = fcgpu_stream_bits(&sampler, stream, dim++);
            x = R(fcgpu_sample_x24(bits) * FCGPU_SAMPLER_SCALE);
            y = R(fcgpu_sample_y24(bits) * FCGPU_SAMPLER_SCALE);
        }
        double get_x_double()
        {
            if(external)
            {
                double ex, ey;
                external->get(external->user, &ex, &ey);
                return ex;
            }
            uint64_t bits = fcgpu_stream_bits(&sampler, stream, dim++);
            return fcgpu_sample_x24(bits) * FCGPU_SAMPLER_SCALE;
        }
        void advance(int n)
        {
            if(external)
                external->advance(external->user, n);
            else
                dim += static_cast<uint32_t>(n);
        }

        void record(V3<R> o, V3<R> d, R t_max, uint32_t kind, uint32_t prim, double t)
        {
            if(!log) return;
            RayRecord r{};
            r.o[0] = o.x; r.o[1] = o.y; r.o[2] = o.z;
            r.d[0] = d.x; r.d[1] = d.y; r.d[2] = d.z;
            r.t_max = t_max;
            r.t = t;
            r.prim = prim;
            r.kind = kind;
            log->push_back(r);
        }

        // origin offset of scene.hpp:85-93, 118-131, 146-152: epsilon along the geometric normal, to the side `positive` names.
        // offset_along_ray restates what the fp32 DEVICE does instead (csrc/fc_kernels.hip: lift_origin): advance along the
        // unit direction w until epsilon above the surface, at most 100 epsilon, the rest along the normal.
        V3<R> lift(V3<R> p, V3<R> n, V3<R> w, bool positive) const
        {
            if(!S.offset_along_ray) return positive ? p + n * S.epsilon : p - n * S.epsilon;
            const R c = std::abs(dot(n, w));
            const R a = std::min(S.epsilon / c, R(100) * S.epsilon);
            const R rest = S.epsilon - a * c;
            const V3<R> q = p + w * a;
            return positive ? q + n * rest : q - n * rest;
        }
        // entity_scene::raycast, scene.hpp:81-108
        bool scene_raycast(const Point<R>& p, V3<R> w, Point<R>& out)
        {
            V3<R> o = lift(p.position, p.normal, w, dot(p.normal, w) > R(0));
            uint32_t prim = ~0u;
            R t;
            closest_rays += 1;
            bool hit = S.closest(o, w, std::numeric_limits<R>::infinity(), brute, out, prim, t);
            record(o, w, std::numeric_limits<R>::infinity(), 0, hit ? prim : ~0u, hit ? static_cast<double>(dot(out.position - o, w) / dot(w, w)) : std::numeric_limits<double>::infinity());
            return hit;
        }
        // entity_scene::visibility(p0, p1), scene.hpp:110-142
        bool visibility(const Point<R>& a, const Point<R>& b)
        {
            V3<R> p0 = a.position, p1 = b.position, n0 = a.normal, n1 = b.normal;
            V3<R> to1 = p1 - p0;
            const V3<R> w01 = S.offset_along_ray ? to1 / length(to1) : to1;  // the direction only matters in the device mode
            p0 = lift(p0, n0, w01, dot(to1, n0) > R(0));
            p1 = lift(p1, n1, w01 * R(-1), dot(to1, n1) < R(0));
            to1 = p1 - p0;
            R len = length(to1);
            V3<R> w = to1 / len;
            any_rays += 1;
            bool occ = S.any(p0, w, len, brute);
            if(occ) any_occluded += 1;
            record(p0, w, len, 1, ~0u, occ ? 1.0 : 0.0);
            return !occ;
        }
        // entity_scene::visibility(p, w), scene.hpp:144-157
        bool visibility_dir(const Point<R>& p, V3<R> w)
        {
            V3<R> o = lift(p.position, p.normal, w, dot(p.normal, w) > R(0));
            any_rays += 1;
            bool occ = S.any(o, w, std::numeric_limits<R>::infinity(), brute);
            if(occ) any_occluded += 1;
            record(o, w, std::numeric_limits<R>::infinity(), 1, ~0u, occ ? 1.0 : 0.0);
            return !occ;
        }

        // ---- helper (priority-medium stack), integrator.hpp:22-119
        const Medium<R>* buffer[10];
        int buffer_size{};
        void helper_reset()
        {
            buffer[0] = &S.vacuum;
            buffer_size = 1;
        }
        const Medium<R>* medium_of(const Point<R>& p) const
        {
            if(p.entity < 0) return nullptr;
            int m = S.entities[p.entity].medium;
            return m < 0 ? nullptr : &S.media[m];
        }
        bool helper_raycast(const Point<R>& p, V3<R> w, const Medium<R>** above, const Medium<R>** below, Point<R>& out)
        {
            const Medium<R>* pm = medium_of(p);
            if(pm != nullptr)
            {
                bool entering = dot(w, p.normal) <= R(0);
                if(entering) buffer[buffer_size++] = pm;
            }
            Point<R> p1;
            if(!scene_raycast(p, w, p1)) return false;
            const Medium<R>* top = buffer[0];
            for(int i = 1; i < buffer_size; ++i)
                if(buffer[i]->priority > top->priority) top = buffer[i];
            const Medium<R>* m1 = medium_of(p1);
            if(m1 == nullptr)
            {
                *above = top;
                *below = top;
                out = p1;
                return true;
            }
            bool entering = dot(w, p1.normal) <= R(0);
            if(entering)
            {
                if(m1->priority <= top->priority) return helper_raycast(p1, w, above, below, out);
                *above = top;
                *below = m1;
                out = p1;
                return true;
            }
            int index = 1;
            for(; index < buffer_size; ++index)
                if(buffer[index] == m1) break;
            if(index == buffer_size) return false;
            std::swap(buffer[index], buffer[buffer_size - 1]);
            buffer_size -= 1;
            const Medium<R>* new_top = buffer[0];
            for(int i = 1; i < buffer_size; ++i)
                if(buffer[i]->priority > new_top->priority) new_top = buffer[i];
            if(new_top->priority == top->priority) return helper_raycast(p1, w, above, below, out);
            *above = new_top;
            *below = top;
            out = p1;
            return true;
        }

        static V3<R> transmittance(const Medium<R>& m, V3<R> a, V3<R> b)  // medium.hpp:34-37, 47-50
        {
            if(m.vacuum) return {R(1), R(1), R(1)};
            V3<R> e = -m.color * m.density * length(b - a);
            return {lm::exp(e.x), lm::exp(e.y), lm::exp(e.z)};
        }

        // const_diffuse_area_light::get_Le, const_diffuse_area_light.hpp:20-26
        V3<R> get_Le(int light, const Point<R>& p, int p_light, V3<R> wo) const
        {
            const auto& L = S.lights[light];
            if(p_light != light || p.surface != S.entities[L.entity].surface) return {};
            if(dot(p.normal, wo) <= R(0)) return {};
            return L.color * L.strength;
        }
        R light_pdf_p(int light) const { return R(1.0 / S.surfaces[S.entities[S.lights[light].entity].surface].area); }

        // surface::sample_p for each kind
        Point<R> sample_surface(const typename Scene<R>::Surface& s, int surface_index, double u_prim, R ux, R uy) const
        {
            Point<R> p;
            p.surface = surface_index;
            if(s.kind == FCGPU_SURFACE_MESH)  // mesh_surface.hpp:276-295
            {
                double u = std::clamp(u_prim, 0.0, 0x1.fffffffffffffp-1);  // distribution.hpp:94-105
                auto it = std::upper_bound(s.cdf.begin(), s.cdf.end(), u);
                size_t tri = static_cast<size_t>(it - s.cdf.begin()) - 1;
                uint32_t i0 = s.indices[3 * tri], i1 = s.indices[3 * tri + 1], i2 = s.indices[3 * tri + 2];
                V3<R> p0 = Scene<R>::vtx(s.positions, i0), p1 = Scene<R>::vtx(s.positions, i1), p2 = Scene<R>::vtx(s.positions, i2);
                R su0 = std::sqrt(ux);  // sampling.hpp:62-66
                R bx = R(1) - su0, by = uy * su0;
                p.position = p0 * bx + p1 * by + p2 * (R(1) - bx - by);
                p.normal = normalize(cross(p1 - p0, p2 - p0));
            }
            else if(s.kind == FCGPU_SURFACE_SPHERE)  // sphere_surface.hpp:157-172
            {
                V3<R> n = sphere_uniform(ux, uy);
                p.position = s.xf.point(n * s.radius);
                p.normal = s.xf.dir(n);
            }
            else  // plane_surface.hpp:116-132
            {
                p.position = s.xf.point({(ux - R(0.5)) * s.size_x, R(0), (uy - R(0.5)) * s.size_y});
                p.normal = s.xf.dir({R(0), R(1), R(0)});
            }
            return p;
        }

        static R power_heuristics(R primary, R alternative)  // forward_mis_integrator.hpp:215-219
        {
            R x = alternative / primary;
            return R(1) / (R(1) + x * x);
        }

        // one sample. Returns Li with beta0 = 1; the caller applies the film semantics.
        // infinity_area_light interface: const_infinity_area_light.hpp / texture_infinity_area_light.hpp
        V3<R> env_Li(V3<R> wi) const { return S.envtex.on ? S.envtex.get_Li(wi) : S.env; }
        R env_pdf(V3<R> wi) const { return S.envtex.on ? S.envtex.pdf_wi(wi) : R(1.0 / (4.0 * PI)); }
        bool env_sample(R ux, R uy, V3<R>& wi, R& pdf, V3<R>& Li) const
        {
            if(S.envtex.on) return S.envtex.sample_wi(ux, uy, wi, pdf, Li);
            if(!nonzero(S.env)) return false;  // const_infinity_area_light::sample_wi, :33-44
            wi = sphere_uniform(ux, uy);
            pdf = R(1.0 / (4.0 * PI));
            Li = S.env;
            return true;
        }

        V3<R> run_once(bool mis, int max_path_length, bool visible_env, int px, int py)
        {
            // perspective_camera::sample_p_and_wi, perspective_camera.hpp:29-75; g++ evaluates the two
            // sampler.get() arguments right to left: the first draw is sample_direction
            R dx, dy, lx, ly;
            get(dx, dy);
            get(lx, ly);
            V3<R> lens{};
            if(S.lens_radius != R(0))
            {
                R a, b;
                disk_concentric(lx, ly, a, b);
                lens.x = a * S.lens_radius;
                lens.y = b * S.lens_radius;
            }
            Point<R> cam;
            cam.position = S.cam_xf.point(lens);
            cam.normal = S.cam_xf.dir({R(0), R(0), R(1)});
            V3<R> plane{((R(px) + dx) / R(S.cam.width) - R(0.5)) * S.plane_w, (R(0.5) - (R(py) + dy) / R(S.cam.height)) * S.plane_h, S.focus};
            V3<R> dir = plane - lens;
            V3<R> wi = S.cam_xf.dir(dir / std::sqrt(dot(dir, dir)));
            return trace_path(mis, max_path_length, visible_env, cam, wi);
        }

        // run_once after measurement.sample_p_and_wi (forward_mis_integrator.hpp:21-210, forward_bsdf_integrator.hpp:
        // 23-99): the bounce loop from a camera point along wi, beta0 = 1
        V3<R> trace_path(bool mis, int max_path_length, bool visible_env, const Point<R>& cam, V3<R> wi)
        {
            V3<R> Li{};
            V3<R> beta{R(1), R(1), R(1)};
            helper_reset();
            const Medium<R>* above = nullptr;
            const Medium<R>* below = nullptr;
            Point<R> p1;
            if(!helper_raycast(cam, wi, &above, &below, p1))
            {
                if(S.has_env && (visible_env || !mis)) Li = Li + beta * env_Li(wi);  // :29-35 / bsdf :31-37
                return Li;
            }
            V3<R> w10 = -wi;
            int l1 = S.entities[p1.entity].light;
            if(l1 >= 0) Li = Li + beta * get_Le(l1, p1, l1, w10);  // :40-43

            for(int i = 2; i <= max_path_length; ++i)
            {
                Bsdf<R> bsdf = material_bsdf(S.materials[S.entities[p1.entity].material], p1, &S);
                int bx = sample_bxdf(bsdf, R(get_x_double()));  // :48
                bool delta = bsdf.lobes[bx].is_delta();
                R ea = above->ior, eb = below->ior;
                if(mis && !delta)
                {
                    // light strategy, :52-99
                    int n = S.light_count();
                    double u_pick = get_x_double();
                    int li = static_cast<int>(std::min(static_cast<size_t>(u_pick * static_cast<double>(n)), static_cast<size_t>(n - 1)));
                    R pdf_light = R(1.0 / static_cast<double>(n));
                    if(S.has_env && li == n - 1)
                    {
                        R ux, uy;
                        get(ux, uy);
                        V3<R> wl, Le;
                        R pdf_wi;
                        if(env_sample(ux, uy, wl, pdf_wi, Le))  // infinity_area_light::sample_wi
                        {
                            V3<R> f = bsdf_evaluate(bsdf, bx, w10, wl, ea, eb);
                            if(nonzero(f) && visibility_dir(p1, wl))
                            {
                                R pdf_bsdf = bsdf_pdf_wi(bsdf, bx, w10, wl, ea, eb);
                                R pdf_l = pdf_light * pdf_wi;
                                R weight = power_heuristics(pdf_l, pdf_bsdf);
                                Li = Li + (beta * f * Le) * (weight * std::abs(dot(p1.normal, wl)) / pdf_l);
                            }
                        }
                        advance(1);
                    }
                    else
                    {
                        // arguments right to left: sample_point first, then sample_primitive
                        R ux, uy;
                        get(ux, uy);
                        double u_prim = get_x_double();
                        const auto& L = S.lights[li];
                        int surf = S.entities[L.entity].surface;
                        Point<R> pL = sample_surface(S.surfaces[surf], surf, u_prim, ux, uy);
                        pL.entity = L.entity;
                        V3<R> Le = get_Le(li, pL, li, p1.position - pL.position);
                        V3<R> d1L = pL.position - p1.position;
                        V3<R> w1L = normalize(d1L);
                        V3<R> f = bsdf_evaluate(bsdf, bx, w10, w1L, ea, eb);
                        if(nonzero(f) && visibility(p1, pL))
                        {
                            R x = std::abs(dot(pL.normal, w1L)) / dot(d1L, d1L);
                            R G = std::abs(dot(p1.normal, w1L)) * x;
                            R pdf_bsdf = bsdf_pdf_wi(bsdf, bx, w10, w1L, ea, eb) * x;
                            R pdf_l = pdf_light * light_pdf_p(li);
                            R weight = power_heuristics(pdf_l, pdf_bsdf);
                            Li = Li + (beta * f * G * Le) * (weight / pdf_l);
                        }
                    }
                }
                else if(mis)
                {
                    advance(3);  // :155
                }

                // bsdf strategy: u2 is the first draw, u1 the second (right-to-left arguments), :106 / :162
                R u2x, u2y, u1x, u1y;
                get(u2x, u2y);
                get(u1x, u1y);
                V3<R> w12, value;
                R pdf_w12;
                if(!bsdf_sample_wi(bsdf, bx, w10, ea, eb, u1x, u1y, u2x, u2y, &w12, &value, &pdf_w12)) break;
                beta = beta * (value * (std::abs(dot(p1.normal, w12)) / pdf_w12));
                Point<R> p2;
                if(!helper_raycast(p1, w12, &above, &below, p2))
                {
                    if(S.has_env)
                    {
                        if(mis && !delta)
                        {
                            R pdf_light = R(1.0 / static_cast<double>(S.light_count()));
                            R pdf_l = pdf_light * env_pdf(w12);
                            Li = Li + power_heuristics(pdf_w12, pdf_l) * beta * env_Li(w12);  // :117-123
                        }
                        else
                        {
                            Li = Li + beta * env_Li(w12);  // :173-176, bsdf :67-70
                        }
                    }
                    break;
                }
                V3<R> w21 = -w12;
                bool entering = dot(p2.normal, w12) <= R(0);
                beta = beta * transmittance(entering ? *above : *below, p1.position, p2.position);  // :130-138
                int l2 = S.entities[p2.entity].light;
                if(l2 >= 0)
                {
                    if(mis && !delta)
                    {
                        R pdf_light = R(1.0 / static_cast<double>(S.light_count()));
                        R pdf_light_p2 = pdf_light * light_pdf_p(l2);
                        V3<R> d = p2.position - p1.position;
                        R pdf_bsdf_p2 = pdf_w12 * std::abs(dot(p2.normal, w12)) / dot(d, d);
                        R weight = power_heuristics(pdf_bsdf_p2, pdf_light_p2);
                        Li = Li + weight * beta * get_Le(l2, p2, l2, w21);  // :140-147
                    }
                    else
                    {
                        Li = Li + beta * get_Le(l2, p2, l2, w21);  // :193-196, bsdf :87-90
                    }
                }
                p1 = p2;
                w10 = w21;
            }
            return Li;
        }
    };

    // ================================================================ C entry points
    template<typename R>
    struct Handle
    {
        std::unique_ptr<Scene<R>> scene;
    };

    template<typename R> void* scene_create(const fcgpu_scene_desc* d)
    {
        auto* h = new Handle<R>{};
        h->scene.reset(new Scene<R>{*d});
        h->scene->finish(*d);
        return h;
    }

    template<typename R>
    int render(void* handle, int integrator, int max_path_length, int visible_inf, uint64_t seed, uint32_t sample_begin, uint32_t sample_count,
               int x0, int y0, int w, int h, int threads, double* rgb_sum, uint64_t* total_count, Stats* stats, RayRecord* ray_log,
               uint64_t ray_log_capacity, uint64_t* ray_log_size)
    {
        const Scene<R>& S = *static_cast<Handle<R>*>(handle)->scene;
        const bool mis = integrator == FCGPU_INTEGRATOR_MIS;
        const fcgpu_sampler_config sampler_cfg = g_stratified_count ? fcgpu_sampler_stratified(seed, g_stratified_count) : fcgpu_sampler_random(seed);
        if(ray_log) threads = 1;
        threads = std::max(1, threads);
        const int W = S.cam.width, H = S.cam.height;
        std::vector<RayRecord> log;
        std::atomic<int> next{0};
        std::vector<uint64_t> nan_counts(threads, 0), closest(threads, 0), anyr(threads, 0), occ(threads, 0);
        auto start = std::chrono::steady_clock::now();
        // film: one private accumulator per pixel is enough here because pixels are disjoint across workers
        std::vector<double> film(static_cast<size_t>(w) * h * 3, 0.0);
        auto body = [&](int index)
        {
            Tracer<R> tr{S};
            if(ray_log) tr.log = &log;
            for(;;)
            {
                int idx = next.fetch_add(1, std::memory_order_relaxed);
                if(idx >= w * h) break;
                int px = x0 + idx % w, py = y0 + idx / w;
                double acc[3]{0, 0, 0};
                for(uint32_t s = 0; s < sample_count; ++s)
                {
                    tr.set_sample(sampler_cfg, px, py, sample_begin + s);
                    V3<R> Li = tr.run_once(mis, max_path_length, visible_inf != 0, px, py);
                    // beta0 = W*H (perspective_camera.hpp:64-68) then the NaN/Inf guard (:126-131)
                    double v[3]{static_cast<double>(Li.x) * W * H, static_cast<double>(Li.y) * W * H, static_cast<double>(Li.z) * W * H};
                    if(!std::isfinite(v[0]) || !std::isfinite(v[1]) || !std::isfinite(v[2]))
                    {
                        nan_counts[index] += 1;
                        continue;
                    }
                    acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2];
                }
                double* out = &film[3 * static_cast<size_t>(idx)];
                out[0] = acc[0]; out[1] = acc[1]; out[2] = acc[2];
            }
            closest[index] = tr.closest_rays;
            anyr[index] = tr.any_rays;
            occ[index] = tr.any_occluded;
        };
        if(threads == 1)
            body(0);
        else
        {
            std::vector<std::thread> pool;
            for(int i = 0; i < threads; ++i) pool.emplace_back(body, i);
            for(auto& t : pool) t.join();
        }
        double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
        if(rgb_sum)
            for(int y = 0; y < h; ++y)
                for(int x = 0; x < w; ++x)
                    for(int c = 0; c < 3; ++c) rgb_sum[3 * (static_cast<size_t>(y0 + y) * W + (x0 + x)) + c] += film[3 * (static_cast<size_t>(y) * w + x) + c];
        uint64_t count = static_cast<uint64_t>(w) * h * sample_count;
        if(total_count) *total_count += count;
        if(stats)
        {
            *stats = Stats{};
            for(int i = 0; i < threads; ++i)
            {
                stats->closest_rays += closest[i];
                stats->any_rays += anyr[i];
                stats->any_occluded += occ[i];
                stats->nan_samples += nan_counts[i];
            }
            stats->samples = count;
            stats->seconds = seconds;
        }
        if(ray_log)
        {
            uint64_t n = std::min<uint64_t>(log.size(), ray_log_capacity);
            std::memcpy(ray_log, log.data(), n * sizeof(RayRecord));
            if(ray_log_size) *ray_log_size = log.size();
        }
        (void)H;
        return 0;
    }

    // One integrator::run_once (integrator.hpp:13-19) after the measurement sample: the host owns camera, film and
    // sampler (its fc::measurement / fc::sampler), this runs the bounce loop. rays[0..2] += closest, any, occluded.
    template<typename R>
    void run_once_from(void* handle, int integrator, int max_path_length, int visible_inf, const double* origin, const double* normal,
                       const double* wi, const ExternalSampler* sampler, double* Li_out, uint64_t* rays)
    {
        const Scene<R>& S = *static_cast<Handle<R>*>(handle)->scene;
        Tracer<R> tr{S};
        tr.external = sampler;
        Point<R> cam;
        cam.position = V3<R>{R(origin[0]), R(origin[1]), R(origin[2])};
        cam.normal = V3<R>{R(normal[0]), R(normal[1]), R(normal[2])};
        V3<R> Li = tr.trace_path(integrator == FCGPU_INTEGRATOR_MIS, max_path_length, visible_inf != 0, cam, V3<R>{R(wi[0]), R(wi[1]), R(wi[2])});
        Li_out[0] = static_cast<double>(Li.x);
        Li_out[1] = static_cast<double>(Li.y);
        Li_out[2] = static_cast<double>(Li.z);
        if(rays)
        {
            rays[0] += tr.closest_rays;
            rays[1] += tr.any_rays;
            rays[2] += tr.any_occluded;
        }
    }

    template<typename R> int trace_closest(void* handle, uint64_t n, const fcgpu_ray* rays, fcgpu_hit* hits, int brute)
    {
        const Scene<R>& S = *static_cast<Handle<R>*>(handle)->scene;
        for(uint64_t i = 0; i < n; ++i)
        {
            V3<R> o{R(rays[i].origin[0]), R(rays[i].origin[1]), R(rays[i].origin[2])}, d{R(rays[i].direction[0]), R(rays[i].direction[1]), R(rays[i].direction[2])};
            Point<R> p;
            uint32_t prim = ~0u;
            R t;
            hits[i] = fcgpu_hit{};
            if(S.closest(o, d, R(rays[i].t_max), brute != 0, p, prim, t))
            {
                hits[i].prim = prim;
                hits[i].t = static_cast<double>(dot(p.position - o, d) / dot(d, d));
            }
            else
            {
                hits[i].prim = ~0u;
                hits[i].t = std::numeric_limits<double>::infinity();
            }
        }
        return 0;
    }
    template<typename R> int trace_any(void* handle, uint64_t n, const fcgpu_ray* rays, uint8_t* out, int brute)
    {
        const Scene<R>& S = *static_cast<Handle<R>*>(handle)->scene;
        for(uint64_t i = 0; i < n; ++i)
        {
            V3<R> o{R(rays[i].origin[0]), R(rays[i].origin[1]), R(rays[i].origin[2])}, d{R(rays[i].direction[0]), R(rays[i].direction[1]), R(rays[i].direction[2])};
            out[i] = S.any(o, d, R(rays[i].t_max), brute != 0) ? 1 : 0;
        }
        return 0;
    }

    struct BsdfQuery
    {
        double wo[3], wi[3], u_lobe, u1[2], u2[2];
    };
    struct BsdfResult
    {
        double f[3], pdf, s_wi[3], s_value[3], s_pdf;
        int32_t lobe, is_delta, s_ok, reserved;
    };
    template<typename R>
    int bsdf_table(const fcgpu_material* m, const double* ng, const double* t, const double* ns, const double* b, double ea, double eb, uint64_t n,
                   const BsdfQuery* q, BsdfResult* out)
    {
        auto v = [](const double* p) { return V3<R>{R(p[0]), R(p[1]), R(p[2])}; };
        for(uint64_t i = 0; i < n; ++i)
        {
            Point<R> p;
            p.normal = v(ng);
            p.tangent = v(t);
            p.shading_normal = v(ns);
            p.bitangent = v(b);
            Bsdf<R> bs = material_bsdf(*m, p);
            int l = sample_bxdf(bs, R(q[i].u_lobe));
            BsdfResult& r = out[i];
            r = BsdfResult{};
            r.lobe = l;
            r.is_delta = bs.lobes[l].is_delta();
            V3<R> f = bsdf_evaluate(bs, l, v(q[i].wo), v(q[i].wi), R(ea), R(eb));
            r.f[0] = f.x; r.f[1] = f.y; r.f[2] = f.z;
            r.pdf = bsdf_pdf_wi(bs, l, v(q[i].wo), v(q[i].wi), R(ea), R(eb));
            V3<R> wi, value;
            R pdf;
            r.s_ok = bsdf_sample_wi(bs, l, v(q[i].wo), R(ea), R(eb), R(q[i].u1[0]), R(q[i].u1[1]), R(q[i].u2[0]), R(q[i].u2[1]), &wi, &value, &pdf);
            if(r.s_ok)
            {
                r.s_wi[0] = wi.x; r.s_wi[1] = wi.y; r.s_wi[2] = wi.z;
                r.s_value[0] = value.x; r.s_value[1] = value.y; r.s_value[2] = value.z;
                r.s_pdf = pdf;
            }
        }
        return 0;
    }
}

#define FCO_EXPORT(PREFIX, R)                                                                                                                     \
    extern "C" void* PREFIX##_scene_create(const fcgpu_scene_desc* d) { return fco::scene_create<R>(d); }                                          \
    extern "C" void PREFIX##_scene_destroy(void* h) { delete static_cast<fco::Handle<R>*>(h); }                                                    \
    extern "C" void PREFIX##_set_epsilon(void* h, double e) { static_cast<fco::Handle<R>*>(h)->scene->epsilon = static_cast<R>(e); }               \
    extern "C" void PREFIX##_set_offset_along_ray(void* h, int on) { static_cast<fco::Handle<R>*>(h)->scene->offset_along_ray = on != 0; }         \
    extern "C" void PREFIX##_set_stratified(uint32_t count) { fco::g_stratified_count = count; }                                                  \
    extern "C" void PREFIX##_set_ulp_noise(uint64_t seed) { fco::g_ulp_noise.store(seed); }                                                       \
    extern "C" int PREFIX##_render(void* h, int integ, int mpl, int vis, uint64_t seed, uint32_t sb, uint32_t sc, int x0, int y0, int w, int hh,     \
                                   int threads, double* rgb, uint64_t* cnt, fco::Stats* st, fco::RayRecord* log, uint64_t cap, uint64_t* nlog)       \
    {                                                                                                                                             \
        return fco::render<R>(h, integ, mpl, vis, seed, sb, sc, x0, y0, w, hh, threads, rgb, cnt, st, log, cap, nlog);                             \
    }                                                                                                                                             \
    extern "C" void PREFIX##_run_once_from(void* h, int integ, int mpl, int vis, const double* origin, const double* normal, const double* wi,     \
                                           const fco::ExternalSampler* sampler, double* Li, uint64_t* rays)                                      \
    {                                                                                                                                             \
        fco::run_once_from<R>(h, integ, mpl, vis, origin, normal, wi, sampler, Li, rays);                                                        \
    }                                                                                                                                             \
    extern "C" int PREFIX##_trace_closest(void* h, uint64_t n, const fcgpu_ray* r, fcgpu_hit* o, int b) { return fco::trace_closest<R>(h, n, r, o, b); } \
    extern "C" int PREFIX##_trace_any(void* h, uint64_t n, const fcgpu_ray* r, uint8_t* o, int b) { return fco::trace_any<R>(h, n, r, o, b); }       \
    extern "C" int PREFIX##_bsdf_table(const fcgpu_material* m, const double* ng, const double* t, const double* ns, const double* b, double ea,     \
                                       double eb, uint64_t n, const fco::BsdfQuery* q, fco::BsdfResult* out)                                        \
    {                                                                                                                                             \
        return fco::bsdf_table<R>(m, ng, t, ns, b, ea, eb, n, q, out);                                                                             \
    }

FCO_EXPORT(fcoracle_f64, double)
FCO_EXPORT(fcoracle_f32, float)
