// fc_renderer.hpp -- C++ host-side mirror of the reference's renderer over the C-ABI.
//
// `fcgpu::gpu_renderer` is shaped like `fc::renderer` (src/renderer/renderer.hpp:20-141):
//     gpu_renderer r{scene_desc, integrator, sample_count, seed, device};  r.run();  r.export_image("room");
// where `integrator` carries what `forward_mis_integrator{max_path_length, visible_infinity_area_light}`
// (forward_mis_integrator.hpp:9-11) or `forward_bsdf_integrator{max_path_length}` carries.
// Errors surface as std::runtime_error with the library's message (the C-ABI itself never throws).
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/fcgpu.h"
#include "../../include/fcgpu_sampler.h"

namespace fcgpu
{
    struct integrator_desc
    {
        int kind{FCGPU_INTEGRATOR_MIS};
        int max_path_length{10};
        bool visible_infinity_area_light{true};
    };

    class gpu_renderer
    {
    public:
        // `stratified`: stratified_sampler{sample_count, seed} instead of random_sampler{sample_count, seed}
        gpu_renderer(const fcgpu_scene_desc& scene, integrator_desc integrator, std::uint32_t sample_count, std::uint64_t seed = 0,
                     int device = 0, int precision = FCGPU_PRECISION_F32, bool stratified = false)
            : integrator_{integrator}, sample_count_{sample_count}, width_{scene.camera.width}, height_{scene.camera.height}
        {
            if(fcgpu_create(device, &ctx_) != FCGPU_OK) throw std::runtime_error(fcgpu_last_error(nullptr));
            fcgpu_options opt{};
            opt.precision = precision;
            opt.sort_materials = 1;
            check(fcgpu_set_options(ctx_, &opt));
            check(fcgpu_upload_scene(ctx_, &scene));
            check(stratified ? fcgpu_set_sampler_stratified(ctx_, seed, sample_count) : fcgpu_set_sampler(ctx_, seed));
            // renderer::run_pixel loops over sampler.get_sample_count() (renderer.hpp:159), which for stratified_sampler is
            // floor(sqrt(sample_count))^2 (stratified_sampler.hpp:15-16, 26-29), not the number it was constructed with
            if(stratified) sample_count_ = fcgpu_sampler_stratified(seed, sample_count).count;
        }
        gpu_renderer(const gpu_renderer&) = delete;
        gpu_renderer& operator=(const gpu_renderer&) = delete;
        ~gpu_renderer()
        {
            if(ctx_) fcgpu_destroy(ctx_);
        }

        // renderer::run (renderer.hpp:65-113): all pixels, all samples; `tile` restricts to a rectangle
        void run(const fcgpu_tile* tile = nullptr)
        {
            check(fcgpu_render_pass(ctx_, tile, 0, sample_count_, integrator_.kind, integrator_.max_path_length,
                                    integrator_.visible_infinity_area_light ? 1 : 0));
            check(fcgpu_sync(ctx_));
            verify();
        }

        // a traversal-stack push that found no room means dropped BVH subtrees, i.e. a wrong image: the upload sizes the
        // stack so that this cannot happen, and a host must not ship the frame if it did
        void verify()
        {
            const fcgpu_counters c = counters();
            if(c.stack_overflows != 0) throw std::runtime_error("fcgpu: traversal stack overflowed (stack_overflows != 0): the image is not valid");
            // more than the reference's 9 nested media on one path (integrator.hpp:24): the push was dropped
            if(c.medium_overflows != 0) throw std::runtime_error("fcgpu: a path nested more than 9 media (medium_overflows != 0): the image is not valid");
        }

        // this renderer's share of a frame rendered by several GPUs: the bands of `band_rows` film rows b with
        // b % band_stride == band_first, all samples, one pass (fcgpu_render_bands); asynchronous, see multi_gpu_renderer
        void start_bands(std::uint32_t band_rows, std::uint32_t band_first, std::uint32_t band_stride)
        {
            check(fcgpu_render_bands(ctx_, band_rows, band_first, band_stride, 0, sample_count_, integrator_.kind, integrator_.max_path_length,
                                     integrator_.visible_infinity_area_light ? 1 : 0));
        }
        void sync() { check(fcgpu_sync(ctx_)); }

        // renderer::export_image (renderer.hpp:115-141): `<name>.raw`, float32 RGB, top row first
        void export_image(const std::string& name) { check(fcgpu_export_image(ctx_, name.c_str())); }

        // the operands of export_image's division
        std::vector<double> film(std::uint64_t* sample_count)
        {
            std::vector<double> sum(3u * static_cast<std::size_t>(width_) * static_cast<std::size_t>(height_));
            check(fcgpu_read_film(ctx_, sum.data(), sample_count));
            return sum;
        }

        fcgpu_counters counters()
        {
            fcgpu_counters c{};
            check(fcgpu_get_counters(ctx_, &c));
            return c;
        }

        fcgpu_ctx* context() { return ctx_; }

    private:
        void check(int rc)
        {
            if(rc != FCGPU_OK) throw std::runtime_error(std::string("fcgpu error ") + std::to_string(rc) + ": " + fcgpu_last_error(ctx_));
        }

        fcgpu_ctx* ctx_{};
        integrator_desc integrator_{};
        std::uint32_t sample_count_{};
        int width_{}, height_{};
    };

    // fc::renderer with worker_count = number of GPUs (renderer.hpp:65-81: N workers share the frame; :125-135: their films
    // are summed at export). One context per device in ONE process: every GPU renders its bands of film rows for all
    // samples (full-width tiles dealt round-robin: each pixel has one owner, so the summed film is bit-identical to one
    // GPU's), the launches of all GPUs run concurrently, fcgpu_reduce_film sums the films into the first context over xGMI.
    // `devices` may name a device more than once (two contexts on one GPU: how the one-GPU tests exercise it).
    class multi_gpu_renderer
    {
    public:
        multi_gpu_renderer(const fcgpu_scene_desc& scene, integrator_desc integrator, std::uint32_t sample_count, const std::vector<int>& devices,
                           std::uint64_t seed = 0, int precision = FCGPU_PRECISION_F32, bool stratified = false, std::uint32_t band_rows = 4)
            : band_rows_{band_rows}
        {
            if(devices.empty()) throw std::runtime_error("multi_gpu_renderer: no devices");
            for(int d : devices) parts_.emplace_back(new gpu_renderer(scene, integrator, sample_count, seed, d, precision, stratified));
        }
        ~multi_gpu_renderer()
        {
            for(gpu_renderer* p : parts_) delete p;
        }
        multi_gpu_renderer(const multi_gpu_renderer&) = delete;
        multi_gpu_renderer& operator=(const multi_gpu_renderer&) = delete;

        void run()
        {
            const std::uint32_t n = static_cast<std::uint32_t>(parts_.size());
            for(std::uint32_t i = 0; i < n; ++i) parts_[i]->start_bands(band_rows_, i, n);  // asynchronous: all GPUs render at once
            for(gpu_renderer* p : parts_)
            {
                p->sync();
                p->verify();
            }
            std::vector<fcgpu_ctx*> ctxs;
            for(gpu_renderer* p : parts_) ctxs.push_back(p->context());
            const int rc = fcgpu_reduce_film(ctxs.data(), static_cast<int>(ctxs.size()), 0);
            if(rc != FCGPU_OK) throw std::runtime_error(std::string("fcgpu_reduce_film: ") + fcgpu_last_error(ctxs[0]));
        }
        gpu_renderer& root() { return *parts_[0]; }  // holds the whole film after run(): export_image / film / counters
        void export_image(const std::string& name) { parts_[0]->export_image(name); }
        std::uint64_t rays()
        {
            std::uint64_t n = 0;
            for(gpu_renderer* p : parts_) n += p->counters().closest_rays + p->counters().shadow_rays;
            return n;
        }

    private:
        std::vector<gpu_renderer*> parts_;
        std::uint32_t band_rows_;
    };
}
