// fc_renderer.hpp -- C++ host-side mirror of the reference's renderer over the C-ABI.
//
// `fcgpu::gpu_renderer` is shaped like `fc::renderer` (src/renderer/renderer.hpp:20-141):
//     gpu_renderer r{scene_desc, integrator, sample_count, seed, device};  r.run();  r.export_image("room");
// where `integrator` carries what `forward_mis_integrator{max_path_length, visible_infinity_area_light}`
// (forward_mis_integrator.hpp:9-11) or `forward_bsdf_integrator{max_path_length}` carries.
// Errors surface as std::runtime_error with the library's message (the C-ABI itself never throws).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
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

        // Part of the job: samples [begin, begin + count) of every pixel. With save_checkpoint / load_checkpoint a render can
        // stop between two such passes and go on later, in another process: the reference writes its image once, at the end
        // (renderer.hpp:115-141), but its film is running sums + a count (render_target.hpp:15-23) and so is this one.
        void run_samples(std::uint32_t begin, std::uint32_t count)
        {
            check(fcgpu_render_pass(ctx_, nullptr, begin, count, integrator_.kind, integrator_.max_path_length,
                                    integrator_.visible_infinity_area_light ? 1 : 0));
            check(fcgpu_sync(ctx_));
            verify();
        }

        // `<path>`: 32-byte header {"FCFILM1", width, height, sample count, next sample index} + W*H*3 raw doubles
        // (fcgpu_save_film: the sums as the device holds them, so that a resumed render is bit-identical to an uninterrupted one)
        void save_checkpoint(const std::string& path, std::uint32_t next_sample)
        {
            std::vector<double> raw(3u * static_cast<std::size_t>(width_) * static_cast<std::size_t>(height_));
            checkpoint_header h{};
            std::memcpy(h.magic, "FCFILM1", 8);
            h.width = static_cast<std::uint32_t>(width_);
            h.height = static_cast<std::uint32_t>(height_);
            h.next_sample = next_sample;
            check(fcgpu_save_film(ctx_, raw.data(), &h.sample_count));
            std::FILE* f = std::fopen(path.c_str(), "wb");
            const bool ok = f && std::fwrite(&h, sizeof(h), 1, f) == 1 && std::fwrite(raw.data(), sizeof(double), raw.size(), f) == raw.size();
            if(f && std::fclose(f) != 0) throw std::runtime_error("gpu_renderer: cannot write " + path);
            if(!ok) throw std::runtime_error("gpu_renderer: cannot write " + path);
        }

        // returns the index of the first sample that is still to render
        std::uint32_t load_checkpoint(const std::string& path)
        {
            std::vector<double> raw(3u * static_cast<std::size_t>(width_) * static_cast<std::size_t>(height_));
            checkpoint_header h{};
            std::FILE* f = std::fopen(path.c_str(), "rb");
            if(!f) throw std::runtime_error("gpu_renderer: cannot read " + path);
            const bool ok = std::fread(&h, sizeof(h), 1, f) == 1 && std::memcmp(h.magic, "FCFILM1", 8) == 0 &&
                            h.width == static_cast<std::uint32_t>(width_) && h.height == static_cast<std::uint32_t>(height_) &&
                            std::fread(raw.data(), sizeof(double), raw.size(), f) == raw.size() && std::fgetc(f) == EOF;
            std::fclose(f);
            if(!ok) throw std::runtime_error("gpu_renderer: " + path + " is not a checkpoint of a " + std::to_string(width_) + "x" + std::to_string(height_) + " film");
            check(fcgpu_restore_film(ctx_, raw.data(), h.sample_count));
            return static_cast<std::uint32_t>(h.next_sample);
        }

        std::uint32_t sample_count() const { return sample_count_; }

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
        struct checkpoint_header
        {
            char magic[8];
            std::uint32_t width, height;
            std::uint64_t sample_count, next_sample;
        };
        static_assert(sizeof(checkpoint_header) == 32, "checkpoint header layout");

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
