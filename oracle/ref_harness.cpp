// ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
//
// Thin C wrapper that drives the REAL reference (the unmodified headers under
// /root/reference/src, compiled where they lie by oracle/Makefile into oracle/_ref/libfcref.so)
// from the POD scene description of include/fcgpu.h:
//   * builds fc::entity_scene exactly as src/example_scenes.hpp does (same constructors),
//   * plugs the counter-based sample stream of include/fcgpu_sampler.h in as an
//     fc::sampler_source (src/core/sampler.hpp:27-34),
//   * runs fc::forward_mis_integrator / fc::forward_bsdf_integrator::run_once per pixel sample
//     with the loop of renderer::worker_thread (src/renderer/renderer.hpp:154-179),
//   * exposes the BVH and the brute-force acceleration structures for ray/hit parity,
//   * exposes bsdf::evaluate / pdf_wi / sample_wi for per-lobe tables.
//
// Accommodations needed to compile the reference with g++:
//   -fpermissive  : members named like their type (acceleration_structure.hpp:14-28 ...)
//   `explicit` is defined away ONLY while src/core/frame.hpp is included, because
//   src/bsdfs/normal_mapping.hpp:758 copy-list-initialises a frame through its explicit
//   constructor (accepted by MSVC only). No reference source is copied or edited.
//
// g++ evaluates function arguments right-to-left; the dimension map in fcgpu_sampler.h records
// the resulting draw -> parameter assignment.

#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <optional>
#include <thread>
#include <vector>

#include "core/math.hpp"
#include "core/assets.hpp"
#include "images/r8_image.hpp"
#include "images/rgb8_image.hpp"
#include "images/rgb32_image.hpp"
#include "images/srgb8_image.hpp"
#include <filesystem>
#define explicit
#include "core/frame.hpp"
#undef explicit

#include "acceleration_structures/brute_force_acceleration_structure.hpp"
#include "acceleration_structures/bvh_acceleration_structure.hpp"
#include "allocators/paged_allocator.hpp"
#include "core/scene.hpp"
#include "integrators/forward_bsdf_integrator.hpp"
#include "integrators/forward_mis_integrator.hpp"
#include "light_distributions/uniform_light_distribution.hpp"
#include "lights/const_diffuse_area_light.hpp"
#include "lights/const_infinity_area_light.hpp"
#include "lights/texture_infinity_area_light.hpp"
#include "textures/image_texture.hpp"
#include "textures/checker_texture.hpp"
#include "materials/diffuse_material.hpp"
#include "materials/glass_material.hpp"
#include "materials/mirror_material.hpp"
#include "materials/plastic_material.hpp"
#include "materials/standard_material.hpp"
#include "materials/transmission_material.hpp"
#include "renderer/cameras/perspective_camera.hpp"
#include "surfaces/mesh_surface.hpp"
#include "surfaces/plane_surface.hpp"
#include "surfaces/sphere_surface.hpp"
#include "textures/const_texture.hpp"

#include "fcgpu.h"
#include "fcgpu_sampler.h"

namespace
{
    using namespace fc;

    // ---------------------------------------------------------------- sampler
    // 0 = random stream; else the stratified mode of fcgpu_sampler.h over that many samples (fcref_set_stratified)
    std::uint32_t g_stratified_count{0};

    class counter_sampler : public sampler_source
    {
    public:
        counter_sampler(int sample_count, fcgpu_sampler_config const& config) : sample_count_{sample_count}, config_{config} {}

        std::unique_ptr<sampler_source> clone() const override
        {
            return std::unique_ptr<sampler_source>(new counter_sampler{sample_count_, config_});
        }
        int get_sample_count() const override { return sample_count_; }
        void set_sample(vector2i const& pixel, int sample) override
        {
            stream_ = fcgpu_stream_begin(&config_, static_cast<std::uint32_t>(pixel.x), static_cast<std::uint32_t>(pixel.y),
                                         static_cast<std::uint32_t>(sample));
            dim_ = 0;
        }
        vector2 get() override
        {
            std::uint64_t bits{fcgpu_stream_bits(&config_, stream_, dim_++)};
            return {fcgpu_sample_x24(bits) * FCGPU_SAMPLER_SCALE, fcgpu_sample_y24(bits) * FCGPU_SAMPLER_SCALE};
        }
        void advance_dimension(int count = 1) override { dim_ += static_cast<std::uint32_t>(count); }
        void set_dimension(int dimension_index) override { dim_ = static_cast<std::uint32_t>(dimension_index); }

    private:
        int sample_count_{};
        fcgpu_sampler_config config_{};
        fcgpu_sample_stream stream_{};
        std::uint32_t dim_{};
    };

    // ---------------------------------------------------------------- recording accel
    struct ray_record
    {
        double o[3];
        double d[3];
        double t_max;
        double t;            // closest: hit distance (inf on miss); any: 1 / 0
        std::uint32_t prim;  // closest: primitive id or ~0u
        std::uint32_t kind;  // 0 closest, 1 any
    };

    struct shared_state
    {
        entity const* entity_base{};
        std::vector<std::uint32_t> prim_offset{};  // per entity
        std::atomic<std::uint64_t> closest_rays{};
        std::atomic<std::uint64_t> any_rays{};
        std::atomic<std::uint64_t> any_occluded{};
        bool count{};
        std::vector<ray_record>* log{};  // single-threaded recording only

        std::uint32_t prim_id(entity_primitive const& ep) const
        {
            return prim_offset[static_cast<std::size_t>(ep.entity - entity_base)] + ep.primitive;
        }
    };

    class wrapping_accel : public acceleration_structure
    {
    public:
        wrapping_accel(std::vector<entity_primitive> eps, shared_state* shared)
            : shared_{shared}
        {
            shared_->entity_base = eps.empty() ? nullptr : eps.front().entity;
            // entity_scene pushes entities in order, so the first primitive belongs to entity 0
            bvh_.reset(new bvh_acceleration_structure{eps});
            brute_.reset(new brute_force_acceleration_structure{std::move(eps)});
        }

        bounds3 get_bounds() const override { return bvh_->get_bounds(); }

        std::optional<acceleration_structure_raycast_surface_point_result> raycast_surface_point(ray3 const& ray, double t_max, allocator_wrapper& allocator) const override
        {
            auto result{bvh_->raycast_surface_point(ray, t_max, allocator)};
            if(shared_->count) shared_->closest_rays.fetch_add(1, std::memory_order_relaxed);
            if(shared_->log != nullptr)
            {
                ray_record r{};
                fill(r, ray, t_max);
                r.kind = 0;
                if(result)
                {
                    r.prim = shared_->prim_id(result->entity_primitive);
                    r.t = dot(result->p->get_position() - ray.origin, ray.direction) / dot(ray.direction, ray.direction);
                }
                else
                {
                    r.prim = ~0u;
                    r.t = std::numeric_limits<double>::infinity();
                }
                shared_->log->push_back(r);
            }
            return result;
        }

        bool raycast(ray3 const& ray, double t_max) const override
        {
            bool result{bvh_->raycast(ray, t_max)};
            if(shared_->count)
            {
                shared_->any_rays.fetch_add(1, std::memory_order_relaxed);
                if(result) shared_->any_occluded.fetch_add(1, std::memory_order_relaxed);
            }
            if(shared_->log != nullptr)
            {
                ray_record r{};
                fill(r, ray, t_max);
                r.kind = 1;
                r.prim = ~0u;
                r.t = result ? 1.0 : 0.0;
                shared_->log->push_back(r);
            }
            return result;
        }

        acceleration_structure const& bvh() const { return *bvh_; }
        acceleration_structure const& brute() const { return *brute_; }

    private:
        static void fill(ray_record& r, ray3 const& ray, double t_max)
        {
            r.o[0] = ray.origin.x; r.o[1] = ray.origin.y; r.o[2] = ray.origin.z;
            r.d[0] = ray.direction.x; r.d[1] = ray.direction.y; r.d[2] = ray.direction.z;
            r.t_max = t_max;
        }

        shared_state* shared_{};
        std::unique_ptr<acceleration_structure> bvh_{};
        std::unique_ptr<acceleration_structure> brute_{};
    };

    class wrapping_accel_factory : public acceleration_structure_factory
    {
    public:
        explicit wrapping_accel_factory(shared_state* shared, wrapping_accel const** out) : shared_{shared}, out_{out} {}
        std::unique_ptr<acceleration_structure> create(std::vector<entity_primitive> eps) const override
        {
            auto* a{new wrapping_accel{std::move(eps), shared_}};
            *out_ = a;
            return std::unique_ptr<acceleration_structure>{a};
        }

    private:
        shared_state* shared_{};
        wrapping_accel const** out_{};
    };

    // ---------------------------------------------------------------- counting measurement
    // Delegates to the reference camera; counts the samples its NaN/Inf guard zeroes
    // (perspective_camera.hpp:126-131) instead of relying on the printed message.
    class counting_measurement : public measurement
    {
    public:
        counting_measurement(camera* cam, std::uint64_t* nan_count) : cam_{cam}, nan_count_{nan_count} {}

        std::optional<measurement_sample_p_and_wi_result> sample_p_and_wi(vector2 const& a, vector2 const& b, allocator_wrapper& al) const override
        {
            return cam_->sample_p_and_wi(a, b, al);
        }
        std::optional<measurement_sample_p> sample_p(surface_point const& v, vector2 const& s, allocator_wrapper& al) const override
        {
            return cam_->sample_p(v, s, al);
        }
        std::optional<measurement_sample_p> sample_p(vector3 const& wi, vector2 const& s, allocator_wrapper& al) const override
        {
            return cam_->sample_p(wi, s, al);
        }
        double pdf_wi(surface_point const& p, vector3 const& wi) const override { return cam_->pdf_wi(p, wi); }
        void add_sample(surface_point const& p, vector3 Li) const override
        {
            if(std::isinf(Li.x) || std::isinf(Li.y) || std::isinf(Li.z) || std::isnan(Li.x) || std::isnan(Li.y) || std::isnan(Li.z))
            {
                *nan_count_ += 1;
                Li = {0.0, 0.0, 0.0};  // same replacement the camera makes, without its std::cout line
            }
            cam_->add_sample(p, Li);
        }
        void add_sample_count(int value) const override { cam_->add_sample_count(value); }

    private:
        camera* cam_{};
        std::uint64_t* nan_count_{};
    };

    // ---------------------------------------------------------------- scene construction
    struct in_memory_mesh
    {
        std::shared_ptr<mesh> m{};
    };

    vector3 v3(double const* p) { return {p[0], p[1], p[2]}; }

    struct ref_scene
    {
        shared_state shared{};
        wrapping_accel const* accel{};
        std::shared_ptr<entity_scene> scene{};
        std::vector<std::shared_ptr<surface>> surfaces{};
        std::vector<std::shared_ptr<material>> materials{};
        std::vector<std::shared_ptr<medium>> media{};
        fcgpu_camera camera{};
        std::uint32_t primitive_total{};
    };

    // const_texture_2d_* for constants, image_texture_2d_rgb / image_texture_2d_r (textures/image_texture.hpp)
    // where the description names an image texture (handle k > 0 = textures[k - 1])
    std::shared_ptr<material> make_material(fcgpu_scene_desc const& d, std::vector<std::shared_ptr<image>> const& images, fcgpu_material const& m)
    {
        auto filter_of = [](fcgpu_texture const& t) { return t.filter == FCGPU_FILTER_BILINEAR ? reconstruction_filter::bilinear : reconstruction_filter::box; };
        auto rgb = [&](double const* c, std::int32_t handle) -> std::shared_ptr<texture_2d_rgb> {
            if(handle > 0)
            {
                fcgpu_texture const& t{d.textures[handle - 1]};
                if(t.kind == FCGPU_TEXTURE_CHECKER)
                    return std::make_shared<checker_texture_2d_rgb>(v3(t.color_a), v3(t.color_b), t.scale);
                return std::shared_ptr<texture_2d_rgb>{new image_texture_2d_rgb{images.at(t.image), filter_of(t), 1}};
            }
            return std::make_shared<const_texture_2d_rgb>(vector3{c[0], c[1], c[2]});
        };
        auto normal_map = [&](std::int32_t handle) -> std::shared_ptr<texture_2d_rgb> {
            if(handle <= 0) return nullptr;
            fcgpu_texture const& t{d.textures[handle - 1]};
            if(t.kind == FCGPU_TEXTURE_CHECKER) return std::make_shared<checker_texture_2d_rgb>(v3(t.color_a), v3(t.color_b), t.scale);
            return std::shared_ptr<texture_2d_rgb>{new image_texture_2d_rgb{images.at(t.image), filter_of(t), 1}};
        };
        auto rg = [](double const* c) { return std::make_shared<const_texture_2d_rg>(vector2{c[0], c[1]}); };
        auto r = [&](double c, std::int32_t handle) -> std::shared_ptr<texture_2d_r> {
            if(handle > 0)
            {
                fcgpu_texture const& t{d.textures[handle - 1]};
                return std::shared_ptr<texture_2d_r>{new image_texture_2d_r{images.at(t.image), filter_of(t), 1}};
            }
            return std::make_shared<const_texture_2d_r>(c);
        };
        switch(m.kind)
        {
        case FCGPU_MATERIAL_DIFFUSE:
            return std::make_shared<diffuse_material>(rgb(m.color_a, m.tex_color_a), normal_map(m.tex_normal));
        case FCGPU_MATERIAL_MIRROR:
            return std::make_shared<mirror_material>(rgb(m.color_a, m.tex_color_a), rg(m.roughness), normal_map(m.tex_normal));
        case FCGPU_MATERIAL_GLASS:
            return std::make_shared<glass_material>(rgb(m.color_a, m.tex_color_a), rgb(m.color_b, m.tex_color_b), rg(m.roughness));
        case FCGPU_MATERIAL_PLASTIC:
            return std::make_shared<plastic_material>(rgb(m.color_a, m.tex_color_a), rgb(m.color_b, m.tex_color_b), rg(m.roughness), r(m.ior, m.tex_ior));
        case FCGPU_MATERIAL_STANDARD:
            return std::make_shared<standard_material>(rgb(m.color_a, m.tex_color_a), r(m.metalness, m.tex_metalness), r(m.roughness[0], m.tex_roughness),
                                                       r(m.ior, m.tex_ior), normal_map(m.tex_normal));
        case FCGPU_MATERIAL_TRANSMISSION:
            return std::make_shared<transmission_material>(rgb(m.color_b, m.tex_color_b), rg(m.roughness));
        default:
            return nullptr;
        }
    }

    std::shared_ptr<surface> make_surface(fcgpu_scene_desc const& d, fcgpu_surface const& s)
    {
        switch(s.kind)
        {
        case FCGPU_SURFACE_SPHERE:
            return std::make_shared<sphere_surface>(pr_transform{v3(s.position), v3(s.rotation)}, s.radius);
        case FCGPU_SURFACE_PLANE:
            return std::make_shared<plane_surface>(pr_transform{v3(s.position), v3(s.rotation)}, vector2{s.size[0], s.size[1]});
        case FCGPU_SURFACE_MESH:
        {
            std::unique_ptr<vector3f[]> positions{new vector3f[s.vertex_count]};
            std::unique_ptr<vector3f[]> normals{};
            std::unique_ptr<vector2f[]> uvs{};
            for(std::uint32_t i{}; i < s.vertex_count; ++i)
            {
                float const* p{d.positions + 3 * static_cast<std::size_t>(s.vertex_offset + i)};
                positions[i] = vector3f{p[0], p[1], p[2]};
            }
            if(s.has_normals)
            {
                normals.reset(new vector3f[s.vertex_count]);
                for(std::uint32_t i{}; i < s.vertex_count; ++i)
                {
                    float const* p{d.normals + 3 * static_cast<std::size_t>(s.vertex_offset + i)};
                    normals[i] = vector3f{p[0], p[1], p[2]};
                }
            }
            if(s.has_uvs)
            {
                uvs.reset(new vector2f[s.vertex_count]);
                for(std::uint32_t i{}; i < s.vertex_count; ++i)
                {
                    float const* p{d.uvs + 2 * static_cast<std::size_t>(s.vertex_offset + i)};
                    uvs[i] = vector2f{p[0], p[1]};
                }
            }
            std::unique_ptr<std::uint32_t[]> indices{new std::uint32_t[s.index_count]};
            std::memcpy(indices.get(), d.indices + s.index_offset, sizeof(std::uint32_t) * s.index_count);
            auto m{std::make_shared<default_mesh>(s.vertex_count, std::move(positions), std::move(normals), std::move(uvs), s.index_count, std::move(indices))};
            return std::make_shared<mesh_surface>(prs_transform{v3(s.position), v3(s.rotation), v3(s.scale)}, m);
        }
        default:
            return nullptr;
        }
    }

    struct render_worker
    {
        std::shared_ptr<render_target> target{};
        std::unique_ptr<camera> cam{};
        std::uint64_t nan_count{};
    };
}

namespace
{
    template<typename P> std::shared_ptr<fc::image> make_raw_image(fcgpu_image const& im)
    {
        std::size_t const n{static_cast<std::size_t>(im.width) * static_cast<std::size_t>(im.height)};
        std::vector<P> pixels{};
        pixels.resize(n);
        std::memcpy(reinterpret_cast<char*>(pixels.data()), im.pixels, n * sizeof(P));
        return std::shared_ptr<fc::raw_image<P>>{new fc::raw_image<P>{{im.width, im.height}, std::move(pixels)}};
    }

    std::shared_ptr<fc::image> make_image(fcgpu_image const& im)
    {
        switch(im.format)
        {
        case FCGPU_IMAGE_R8: return make_raw_image<fc::r8_pixel>(im);
        case FCGPU_IMAGE_RGB8: return make_raw_image<fc::rgb8_pixel>(im);
        case FCGPU_IMAGE_SRGB8: return make_raw_image<fc::srgb8_pixel>(im);
        default: return make_raw_image<fc::rgb32_pixel>(im);
        }
    }
}

// ---------------------------------------------------------------- the drop-in integrator
// fc::integrator whose run_once hands the bounce loop to this repository's restatement of the device algorithm, on the
// reference's OWN measurement (camera + film) and sampler objects: single-sample debugging against the stock
// integrators, and the proof that the integrator interface is honoured. The loop arrives as a function pointer
// (liboracle.so: fcoracle_f64_run_once_from), so the harness links nothing of ours.
namespace
{
    struct external_sampler
    {
        void (*get)(void* user, double* x, double* y);
        void (*advance)(void* user, int count);
        void* user;
    };
    using run_once_from_fn = void (*)(void* scene, int integrator, int max_path_length, int visible_inf, double const* origin,
                                      double const* normal, double const* wi, external_sampler const* sampler, double* Li, std::uint64_t* rays);

    class gpu_forward_mis_integrator : public fc::integrator
    {
    public:
        gpu_forward_mis_integrator(run_once_from_fn loop, void* restated_scene, int kind, int max_path_length, bool visible_infinity_area_light,
                                   shared_state* counters)
            : loop_{loop}, scene_{restated_scene}, kind_{kind}, max_path_length_{max_path_length}, visible_{visible_infinity_area_light}, counters_{counters}
        { }

        void run_once(fc::measurement& measurement, fc::scene const&, fc::sampler& sampler, fc::allocator_wrapper& allocator) const override
        {
            using namespace fc;
            measurement.add_sample_count(1);
            auto ms{measurement.sample_p_and_wi(sampler.get(), sampler.get(), allocator)};  // as forward_mis_integrator.hpp:17
            if(!ms) return;
            vector3 beta{ms->Wo * (std::abs(dot(ms->p->get_normal(), ms->wi)) / (ms->pdf_p * ms->pdf_wi))};
            external_sampler ext{
                [](void* user, double* x, double* y) {
                    vector2 v{static_cast<fc::sampler*>(user)->get()};
                    *x = v.x;
                    *y = v.y;
                },
                [](void* user, int count) { static_cast<fc::sampler*>(user)->advance_dimension(count); }, &sampler};
            vector3 o{ms->p->get_position()}, n{ms->p->get_normal()};
            double origin[3]{o.x, o.y, o.z}, normal[3]{n.x, n.y, n.z}, wi[3]{ms->wi.x, ms->wi.y, ms->wi.z}, Li[3]{};
            std::uint64_t rays[3]{};
            loop_(scene_, kind_, max_path_length_, visible_ ? 1 : 0, origin, normal, wi, &ext, Li, rays);
            if(counters_ != nullptr && counters_->count)
            {
                counters_->closest_rays += rays[0];
                counters_->any_rays += rays[1];
                counters_->any_occluded += rays[2];
            }
            measurement.add_sample(*ms->p, beta * vector3{Li[0], Li[1], Li[2]});
        }

    private:
        run_once_from_fn loop_{};
        void* scene_{};
        int kind_{};
        int max_path_length_{};
        bool visible_{};
        shared_state* counters_{};
    };

    run_once_from_fn g_shim_loop{};
    void* g_shim_scene{};
}

extern "C"
{
    struct fcref_stats
    {
        std::uint64_t closest_rays;
        std::uint64_t any_rays;
        std::uint64_t any_occluded;
        std::uint64_t nan_samples;
        std::uint64_t samples;
        double seconds;
    };

    void* fcref_scene_create(fcgpu_scene_desc const* d)
    {
        auto* rs{new ref_scene{}};
        rs->camera = d->camera;

        for(std::uint32_t i{}; i < d->surface_count; ++i)
        {
            rs->surfaces.push_back(make_surface(*d, d->surfaces[i]));
            if(rs->surfaces.back() == nullptr) { delete rs; return nullptr; }
        }
        // images exactly as assets::load_image builds them from the raw file bytes (assets.cpp:205-231)
        std::vector<std::shared_ptr<image>> images{};
        for(std::uint32_t k{0}; k < d->image_count; ++k) images.push_back(make_image(d->images[k]));

        for(std::uint32_t i{}; i < d->material_count; ++i)
        {
            rs->materials.push_back(make_material(*d, images, d->materials[i]));
            if(rs->materials.back() == nullptr) { delete rs; return nullptr; }
        }
        for(std::uint32_t i{}; i < d->medium_count; ++i)
        {
            auto const& m{d->media[i]};
            rs->media.push_back(std::make_shared<uniform_medium>(m.priority, m.ior, v3(m.color), m.density));
        }

        std::vector<entity> entities{};
        std::uint32_t offset{};
        for(std::uint32_t i{}; i < d->entity_count; ++i)
        {
            auto const& e{d->entities[i]};
            entity ent{};
            ent.surface = rs->surfaces[e.surface];
            ent.material = rs->materials[e.material];
            if(e.emissive)
            {
                ent.area_light = std::make_shared<const_diffuse_area_light>(ent.surface.get(), v3(e.emission_color), e.emission_strength);
            }
            if(e.medium >= 0) ent.medium = rs->media[e.medium];
            rs->shared.prim_offset.push_back(offset);
            offset += ent.surface->get_primitive_count();
            entities.push_back(std::move(ent));
        }
        rs->primitive_total = offset;

        std::shared_ptr<infinity_area_light> env{};
        if(d->env_kind == FCGPU_ENV_CONST)
        {
            env = std::make_shared<const_infinity_area_light>(v3(d->env_color), 1.0);
        }
        else if(d->env_kind == FCGPU_ENV_TEXTURE)
        {
            // example_scenes.hpp:339-341
            fcgpu_env_texture const& t{d->env_texture};
            std::shared_ptr<image_texture_2d_rgb> texture{new image_texture_2d_rgb{
                images.at(t.image), t.filter == FCGPU_FILTER_BILINEAR ? reconstruction_filter::bilinear : reconstruction_filter::box,
                t.integral_sample_count}};
            env = std::shared_ptr<infinity_area_light>{new texture_infinity_area_light{
                pr_transform{{}, v3(t.rotation)}, texture, t.strength, vector2i{t.distribution_width, t.distribution_height}}};
        }

        wrapping_accel_factory factory{&rs->shared, &rs->accel};
        uniform_light_distribution_factory uldf{};
        uniform_spatial_light_distribution_factory usldf{};
        rs->scene = std::make_shared<entity_scene>(std::move(entities), env, factory, uldf, usldf);
        return rs;
    }

    // Route the following fcref_render calls through gpu_forward_mis_integrator (loop = null: back to the stock integrators).
    void fcref_set_drop_in_integrator(void* loop, void* restated_scene)
    {
        g_shim_loop = reinterpret_cast<run_once_from_fn>(loop);
        g_shim_scene = restated_scene;
    }

    void fcref_scene_destroy(void* scene) { delete static_cast<ref_scene*>(scene); }

    std::uint32_t fcref_primitive_count(void* scene) { return static_cast<ref_scene*>(scene)->primitive_total; }

    // Renders samples [sample_begin, sample_begin+sample_count) of every pixel of the tile.
    // rgb_sum (W*H*3 doubles, may be NULL) receives the SUM over workers of
    // render_target::get_pixel_sample_sum, total_count the sum of get_sample_count -- the two
    // quantities renderer::export_image divides (renderer.hpp:118-135). Values are ADDED to
    // whatever rgb_sum holds so passes can be accumulated.
    int fcref_render(void* scene_ptr, int integrator_kind, int max_path_length, int visible_inf, std::uint64_t seed,
                     std::uint32_t sample_begin, std::uint32_t sample_count, int x0, int y0, int w, int h, int threads,
                     double* rgb_sum, std::uint64_t* total_count, fcref_stats* stats, ray_record* ray_log, std::uint64_t ray_log_capacity,
                     std::uint64_t* ray_log_size)
    {
        auto* rs{static_cast<ref_scene*>(scene_ptr)};
        std::shared_ptr<integrator> integ{};
        if(g_shim_loop != nullptr)
            integ.reset(new gpu_forward_mis_integrator{g_shim_loop, g_shim_scene, integrator_kind, max_path_length, visible_inf != 0, &rs->shared});
        else if(integrator_kind == FCGPU_INTEGRATOR_MIS)
            integ.reset(new forward_mis_integrator{max_path_length, visible_inf != 0});
        else
            integ.reset(new forward_bsdf_integrator{max_path_length});

        vector2i resolution{rs->camera.width, rs->camera.height};
        perspective_camera_factory cam_factory{pr_transform{v3(rs->camera.position), v3(rs->camera.rotation)}, rs->camera.fov, rs->camera.lens_radius, rs->camera.focus_distance};

        std::vector<ray_record> log{};
        if(ray_log != nullptr) threads = 1;
        threads = std::max(1, threads);

        rs->shared.count = stats != nullptr;
        rs->shared.closest_rays = 0;
        rs->shared.any_rays = 0;
        rs->shared.any_occluded = 0;
        rs->shared.log = ray_log != nullptr ? &log : nullptr;

        std::vector<render_worker> workers(threads);
        for(auto& wk : workers)
        {
            wk.target.reset(new render_target{resolution});
            wk.cam = cam_factory.create(wk.target);
        }

        std::atomic<int> next_pixel{};
        int pixel_total{w * h};
        auto start{std::chrono::steady_clock::now()};

        auto body = [&](int index)
        {
            // renderer::worker_thread, renderer.hpp:154-179
            paged_allocator pages{1024 * 1024};
            allocator_wrapper sample_allocator{&pages};
            counter_sampler sampler{static_cast<int>(sample_count),
                                    g_stratified_count ? fcgpu_sampler_stratified(seed, g_stratified_count) : fcgpu_sampler_random(seed)};
            render_worker& wk{workers[index]};
            counting_measurement meas{wk.cam.get(), &wk.nan_count};
            while(true)
            {
                int idx{next_pixel.fetch_add(1, std::memory_order_relaxed)};
                if(idx >= pixel_total) break;
                vector2i pixel{x0 + idx % w, y0 + idx / w};
                wk.cam->set_pixel(pixel);
                for(std::uint32_t s{}; s < sample_count; ++s)
                {
                    sampler.set_sample(pixel, static_cast<int>(sample_begin + s));
                    integ->run_once(meas, *rs->scene, sampler, sample_allocator);
                    sample_allocator.clear();
                }
            }
        };

        if(threads == 1)
        {
            body(0);
        }
        else
        {
            std::vector<std::thread> pool{};
            for(int i{}; i < threads; ++i) pool.emplace_back(body, i);
            for(auto& t : pool) t.join();
        }
        double seconds{std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count()};

        std::uint64_t count{};
        std::uint64_t nan_count{};
        for(auto& wk : workers)
        {
            count += wk.target->get_sample_count();
            nan_count += wk.nan_count;
        }
        if(rgb_sum != nullptr)
        {
            for(int y{y0}; y < y0 + h; ++y)
                for(int x{x0}; x < x0 + w; ++x)
                {
                    vector3 c{};
                    for(auto& wk : workers) c += wk.target->get_pixel_sample_sum({x, y});
                    double* out{rgb_sum + 3 * (static_cast<std::size_t>(y) * resolution.x + x)};
                    out[0] += c.x; out[1] += c.y; out[2] += c.z;
                }
        }
        if(total_count != nullptr) *total_count += count;
        if(stats != nullptr)
        {
            stats->closest_rays = rs->shared.closest_rays;
            stats->any_rays = rs->shared.any_rays;
            stats->any_occluded = rs->shared.any_occluded;
            stats->nan_samples = nan_count;
            stats->samples = count;
            stats->seconds = seconds;
        }
        if(ray_log != nullptr)
        {
            std::uint64_t n{std::min<std::uint64_t>(log.size(), ray_log_capacity)};
            std::memcpy(ray_log, log.data(), n * sizeof(ray_record));
            if(ray_log_size != nullptr) *ray_log_size = log.size();
        }
        rs->shared.count = false;
        rs->shared.log = nullptr;
        return 0;
    }

    // closest-hit through the reference's BVH (brute_force = 0) or its brute-force structure
    int fcref_trace_closest(void* scene_ptr, std::uint64_t n, fcgpu_ray const* rays, fcgpu_hit* hits, int brute_force)
    {
        auto* rs{static_cast<ref_scene*>(scene_ptr)};
        acceleration_structure const& accel{brute_force ? rs->accel->brute() : rs->accel->bvh()};
        paged_allocator pages{1024 * 1024};
        allocator_wrapper al{&pages};
        for(std::uint64_t i{}; i < n; ++i)
        {
            ray3 ray{v3(rays[i].origin), v3(rays[i].direction)};
            auto r{accel.raycast_surface_point(ray, rays[i].t_max, al)};
            fcgpu_hit& h{hits[i]};
            h = fcgpu_hit{};
            if(r)
            {
                h.prim = rs->shared.prim_id(r->entity_primitive);
                h.t = dot(r->p->get_position() - ray.origin, ray.direction) / dot(ray.direction, ray.direction);
            }
            else
            {
                h.prim = ~0u;
                h.t = std::numeric_limits<double>::infinity();
            }
            al.clear();
        }
        return 0;
    }

    int fcref_trace_any(void* scene_ptr, std::uint64_t n, fcgpu_ray const* rays, std::uint8_t* occluded, int brute_force)
    {
        auto* rs{static_cast<ref_scene*>(scene_ptr)};
        acceleration_structure const& accel{brute_force ? rs->accel->brute() : rs->accel->bvh()};
        for(std::uint64_t i{}; i < n; ++i)
        {
            ray3 ray{v3(rays[i].origin), v3(rays[i].direction)};
            occluded[i] = accel.raycast(ray, rays[i].t_max) ? 1 : 0;
        }
        return 0;
    }

    // Per-lobe tables: material::evaluate on a synthetic hit (geometric normal ng, shading frame
    // t/ns/b, uv 0), then for query i: lobe = bsdf::sample_bxdf(u_lobe[i]);
    //   f[i]     = bsdf::evaluate(lobe, wo[i], wi[i], eta_a, eta_b)
    //   pdf[i]   = bsdf::pdf_wi(...)
    //   sample   = bsdf::sample_wi(lobe, wo[i], eta_a, eta_b, u1[i], u2[i]) -> ok, wi, value, pdf
    struct fcref_bsdf_query
    {
        double wo[3];
        double wi[3];
        double u_lobe;
        double u1[2];
        double u2[2];
    };
    struct fcref_bsdf_result
    {
        double f[3];
        double pdf;
        double s_wi[3];
        double s_value[3];
        double s_pdf;
        std::int32_t lobe;
        std::int32_t is_delta;
        std::int32_t s_ok;
        std::int32_t reserved;
    };

    int fcref_bsdf_table(fcgpu_material const* m, double const* ng, double const* tangent, double const* ns, double const* bitangent,
                         double eta_a, double eta_b, std::uint64_t n, fcref_bsdf_query const* q, fcref_bsdf_result* out)
    {
        fcgpu_material constants{*m};  // the table is over one constant material: no image textures here
        constants.tex_color_a = constants.tex_color_b = constants.tex_roughness = 0;
        constants.tex_metalness = constants.tex_ior = constants.tex_normal = 0;
        auto mat{make_material(fcgpu_scene_desc{}, {}, constants)};
        if(mat == nullptr) return -1;
        paged_allocator pages{1024 * 1024};
        allocator_wrapper al{&pages};
        for(std::uint64_t i{}; i < n; ++i)
        {
            surface_point p{};
            p.set_normal(v3(ng));
            p.set_shading_tangent(v3(tangent));
            p.set_shading_normal(v3(ns));
            p.set_shading_bitangent(v3(bitangent));
            bsdf const* b{mat->evaluate(p, al)};
            int lobe{b->sample_bxdf(q[i].u_lobe)};
            fcref_bsdf_result& r{out[i]};
            r = fcref_bsdf_result{};
            r.lobe = lobe;
            r.is_delta = b->get_type(lobe) == bxdf_type::delta ? 1 : 0;
            vector3 wo{v3(q[i].wo)};
            vector3 wi{v3(q[i].wi)};
            vector3 f{b->evaluate(lobe, wo, wi, eta_a, eta_b)};
            r.f[0] = f.x; r.f[1] = f.y; r.f[2] = f.z;
            r.pdf = b->pdf_wi(lobe, wo, wi, eta_a, eta_b);
            vector3 swi{};
            vector3 value{};
            double pdf{};
            auto ok{b->sample_wi(lobe, wo, eta_a, eta_b, vector2{q[i].u1[0], q[i].u1[1]}, vector2{q[i].u2[0], q[i].u2[1]}, &swi, &value, &pdf)};
            r.s_ok = ok == sample_result::success ? 1 : 0;
            if(r.s_ok)
            {
                r.s_wi[0] = swi.x; r.s_wi[1] = swi.y; r.s_wi[2] = swi.z;
                r.s_value[0] = value.x; r.s_value[1] = value.y; r.s_value[2] = value.z;
                r.s_pdf = pdf;
            }
            al.clear();
        }
        return 0;
    }

    std::uint64_t fcref_ray_record_size() { return sizeof(ray_record); }

    void fcref_set_stratified(std::uint32_t count) { g_stratified_count = count; }

    // ---- the reference's own asset loader (src/core/assets.cpp, compiled where it lies): reads
    // <root>/assets/<name>.mesh and <root>/assets/images/<name>.{metadata,asset}. fc::assets resolves
    // paths against the current directory, so these calls change it for their duration (tests only).
    // Pass null arrays to query the counts first. Returns 0, or -1 when the loader rejects the file
    // (it terminates on a bad file: `throw;` without an active exception, assets.cpp:108 -- the
    // caller is expected to hand over files that exist).
    int fcref_load_mesh(char const* root, char const* name, std::uint32_t* vertex_count, std::uint32_t* index_count, int* has_normals,
                        int* has_uvs, float* positions, float* normals, float* uvs, std::uint32_t* indices)
    {
        std::filesystem::path const back{std::filesystem::current_path()};
        std::filesystem::current_path(root);
        int rc{0};
        {
            fc::assets loader{};
            std::shared_ptr<fc::mesh> m{loader.get_mesh(name)};
            if(!m)
                rc = -1;
            else
            {
                std::uint32_t const nv{m->get_vertex_count()}, ni{m->get_index_count()};
                *vertex_count = nv;
                *index_count = ni;
                *has_normals = m->get_normals() != nullptr;
                *has_uvs = m->get_uvs() != nullptr;
                if(positions) std::memcpy(positions, m->get_positions(), sizeof(float) * 3 * nv);
                if(normals && m->get_normals()) std::memcpy(normals, m->get_normals(), sizeof(float) * 3 * nv);
                if(uvs && m->get_uvs()) std::memcpy(uvs, m->get_uvs(), sizeof(float) * 2 * nv);
                if(indices) std::memcpy(indices, m->get_indices(), sizeof(std::uint32_t) * ni);
            }
        }
        std::filesystem::current_path(back);
        return rc;
    }

    // Texel decode by the reference's pixel classes (images/*.hpp): the raw bytes of an .asset file
    // (format 0 r8, 1 rgb8, 2 srgb8, 3 rgb32) are memcpy'd into the reference's pixel vector exactly as
    // assets::load_image reads them (assets.cpp:205-231), then image::rgb() is taken per texel.
    int fcref_decode_image(int format, int width, int height, void const* raw, double* rgb)
    {
        std::size_t const n{static_cast<std::size_t>(width) * static_cast<std::size_t>(height)};
        std::shared_ptr<fc::image> img{};
        auto make{[&]<typename P>() {
            std::vector<P> pixels{};
            pixels.resize(n);
            std::memcpy(reinterpret_cast<char*>(pixels.data()), raw, n * sizeof(P));
            img = std::shared_ptr<fc::raw_image<P>>{new fc::raw_image<P>{{width, height}, std::move(pixels)}};
        }};
        if(format == 0) make.template operator()<fc::r8_pixel>();
        else if(format == 1) make.template operator()<fc::rgb8_pixel>();
        else if(format == 2) make.template operator()<fc::srgb8_pixel>();
        else if(format == 3) make.template operator()<fc::rgb32_pixel>();
        else return -1;
        for(int y{0}; y < height; ++y)
            for(int x{0}; x < width; ++x)
            {
                fc::vector3 const c{img->rgb({x, y})};
                double* o{rgb + 3 * (static_cast<std::size_t>(y) * width + x)};
                o[0] = c.x;
                o[1] = c.y;
                o[2] = c.z;
            }
        return 0;
    }

    // The file path through fc::assets::get_image. NOTE: built with g++ this returns -1 for every
    // file: assets.cpp:168 brace-initialises a json from the parsed document, which g++ turns into a
    // one-element ARRAY (an nlohmann/compiler interaction; MSVC copies), and from_json's at("type")
    // then throws type_error.304. The image file format is therefore pinned by its description in
    // assets.cpp only; the texel decode is pinned by fcref_decode_image above.
    int fcref_load_image(char const* root, char const* name, int* width, int* height, double* rgb)
    {
        std::filesystem::path const back{std::filesystem::current_path()};
        std::filesystem::current_path(root);
        int rc{0};
        {
            fc::assets loader{};
            std::shared_ptr<fc::image> img{};
            try
            {
                img = loader.get_image(name);
            }
            catch(std::exception const& e)
            {
                std::fprintf(stderr, "fcref_load_image: %s\n", e.what());
            }
            if(!img)
                rc = -1;
            else
            {
                fc::vector2i const res{img->get_resolution()};
                *width = res.x;
                *height = res.y;
                if(rgb)
                    for(int y{0}; y < res.y; ++y)
                        for(int x{0}; x < res.x; ++x)
                        {
                            fc::vector3 const c{img->rgb({x, y})};
                            double* o{rgb + 3 * (static_cast<std::size_t>(y) * res.x + x)};
                            o[0] = c.x;
                            o[1] = c.y;
                            o[2] = c.z;
                        }
            }
        }
        std::filesystem::current_path(back);
        return rc;
    }
}
