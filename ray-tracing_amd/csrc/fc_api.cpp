// fc_api.cpp -- implementation of the C-ABI of include/fcgpu.h: context, scene upload, pass
// scheduling (the host side of the wavefront loop), film read-out.
//
// Host-side counterpart of renderer::run / worker_thread / export_image
// (src/renderer/renderer.hpp:65-113, 154-179, 115-141): instead of threads pulling pixels it
// cuts a pass into batches of at most `max_paths_in_flight` paths and, per batch, enqueues
// generate -> { extend -> shade<lobe>... -> shadow } x max_path_length -> accumulate on one HIP
// stream. Nothing here computes radiance on the CPU.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <new>
#include <chrono>
#include <string>
#include <vector>

#include "../../include/fcgpu.h"
#include "fc_host.h"
#include "fc_kernels.h"

using namespace fcg;

namespace
{
    thread_local std::string g_create_error;

    // Paths per batch. Every kernel of a bounce ends with a tail in which the slowest rays drain while
    // most of the chip idles (~70 us at 1080p whatever the batch size), so batches are made large:
    // 2^27 paths = 64 samples of a 1080p frame, ~27 GB of queues in fp32 and twice that in fp64 (288 GB of HBM per
    // GPU); queues are only ever as large as the largest pass asked for. Measured on MI355X: 2^21 -> 199, 2^23 -> 262,
    // 2^25 -> 288 Msamples/s with the first kernels; 2^24 -> 408, 2^25 -> 436, 2^26 -> 449, 2^27 -> 454 with the last.
    constexpr uint32_t DEFAULT_PATHS_IN_FLIGHT = 1u << 27;
}

struct fcgpu_ctx
{
    int device{};
    hipStream_t stream{};
    std::string err;
    fcgpu_options opt{};
    fcgpu_sampler_config sampler{fcgpu_sampler_random(0)};
    bool has_scene{};
    uint64_t flatten_us{};  // host time of the last scene flattening incl. the BVH build (either builder)
    FlatScene flat;  // host copy (typed arrays are small; nodes/tris are kept for introspection)
    std::vector<void*> scene_allocs;
    std::vector<void*> queue_allocs;
    SceneDev<float> s32{};
    SceneDev<double> s64{};
    Wavefront<float> q32{}, q32b{};    // b: second batch slot (two batches in flight, FCGPU_STREAMS=2)
    Wavefront<double> q64{}, q64b{};
    int slots{1};
    size_t stack_spill_words{};        // words of one batch slot's spill area
    uint32_t stack_spill_entries{};    // entries per lane column of it
    uint32_t* stack_spill{};           // scene_allocs entry: traversal stack entries beyond STACK_CAP, per grid lane
    hipStream_t stream_b{};
    hipStream_t span_stream{};         // stream the next StageSpan records on
    QueueCounters* counters_b{};
    hipEvent_t ev_fork{}, ev_join{}, ev_acc[2]{};
    int queue_precision{-1};
    uint32_t queue_capacity{};
    uint32_t queue_limit{};
    bool queue_shrunk{};               // the last allocation had to halve the batch size (device memory)
    double* film{};
    uint64_t film_elems{};
    QueueCounters* counters{};
    unsigned long long* stats{};
    bool decision_log{};               // fcgpu_set_decision_log
    uint32_t *dec_hit{}, *dec_flags{};  // [dec_bounces][dec_paths], owned (not in a pool: sized per pass)
    uint32_t dec_bounces{};
    uint64_t dec_paths{};
    unsigned char* noalloc_scratch{};  // this device's copy of the static scratch array (fc_kernels.hip), set at create
    // fcgpu_reduce_film: which film rows hold anything since the last clear (so that a contributor that rendered bands
    // sends its bands, not the frame), the landing zone of the peer copies on a root, an event to order streams of
    // different devices, and the bytes the copies moved
    struct Coverage
    {
        bool any{}, banded{};
        uint32_t rows{}, first{}, stride{1};
    } cover;
    double* reduce_staging{};
    hipEvent_t ev_reduce{};
    uint64_t reduce_bytes{};
    uint64_t samples{};
    uint64_t launches{};
    int grid{};
    int stream_grid{};  // grid of the streaming kernels (LaunchCfg::stream_grid)
    unsigned long long* timeline{};  // FCGPU_TIMELINE development aid
    size_t timeline_words{};
    int lds_pad_extend{0}, lds_pad_shadow{0};  // FCGPU_LDS_PAD_EXTEND / _SHADOW (bytes)
    int finish_min{16}, descend_min{12};  // traversal tuning, overridable through FCGPU_FINISH_MIN / FCGPU_DESCEND_MIN
    int finish_min_shadow{16}, descend_min_shadow{24};  // ... FCGPU_FINISH_MIN_SHADOW / FCGPU_DESCEND_MIN_SHADOW
    // stage profiling: a pool of event pairs drained at sync points
    bool profiling{};
    struct TimedSpan
    {
        hipEvent_t a, b;
        int stage;
    };
    std::vector<TimedSpan> spans_free, spans_busy;
    fcgpu_stage_times stage_times{};
};

namespace
{
    int fail(fcgpu_ctx* c, int code, const std::string& msg)
    {
        if(c) c->err = msg;
        return code;
    }

    int hip_fail(fcgpu_ctx* c, hipError_t e, const char* what)
    {
        if(e == hipErrorOutOfMemory) (void)hipGetLastError();  // the sticky "last error" must not fail the next launch check
        return fail(c, e == hipErrorOutOfMemory ? FCGPU_E_NOMEM : FCGPU_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
    }

    // No C++ exception crosses the ABI (include/fcgpu.h): the entry points that allocate host memory run inside this.
    template<typename F> int guarded(fcgpu_ctx* c, F&& body) noexcept
    {
        int code = FCGPU_E_INVALID;
        const char* what = "unexpected exception";
        try
        {
            return body();
        }
        catch(const std::bad_alloc&)
        {
            code = FCGPU_E_NOMEM;
            what = "out of host memory";
        }
        catch(...)
        {
        }
        try
        {
            if(c) c->err = what;
        }
        catch(...)
        {
        }
        return code;
    }

#define HIP_TRY(c, call)                                 \
    do                                                   \
    {                                                    \
        hipError_t e_ = (call);                          \
        if(e_ != hipSuccess) return hip_fail(c, e_, #call); \
    } while(0)

    template<typename T> int upload(fcgpu_ctx* c, std::vector<void*>& pool, const std::vector<T>& v, const T** out)
    {
        void* p = nullptr;
        size_t bytes = std::max<size_t>(sizeof(T) * v.size(), 16);
        HIP_TRY(c, hipMalloc(&p, bytes));
        pool.push_back(p);
        if(!v.empty()) HIP_TRY(c, hipMemcpy(p, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
        *out = static_cast<const T*>(p);
        return FCGPU_OK;
    }

    void free_pool(std::vector<void*>& pool)
    {
        for(void* p : pool) (void)hipFree(p);
        pool.clear();
    }

    template<typename R> int fill_scene(fcgpu_ctx* c, SceneDev<R>& S, const FlatTyped<R>& T)
    {
        int rc;
        if((rc = upload(c, c->scene_allocs, T.xprims, &S.xprims))) return rc;
        if((rc = upload(c, c->scene_allocs, T.tri_shade, &S.tri_shade))) return rc;
        if((rc = upload(c, c->scene_allocs, T.lobes, &S.lobes))) return rc;
        if((rc = upload(c, c->scene_allocs, T.media, &S.media))) return rc;
        if((rc = upload(c, c->scene_allocs, T.lights, &S.lights))) return rc;
        S.camera = T.camera;
        for(int k = 0; k < 3; ++k) S.env[k] = T.env[k];
        S.light_count = c->flat.light_count;
        S.env_kind = c->flat.env_kind;
        S.stack_spill = c->stack_spill;
        S.stack_spill_stride = static_cast<uint32_t>(c->grid) * 256u;
        S.stack_spill_cap = c->stack_spill_entries;
        S.max_stack = c->flat.max_stack;
        S.node_count = static_cast<uint32_t>(c->flat.nodes.size());
        S.textured_materials = 0;
        for(const MaterialRec& m : c->flat.materials) S.textured_materials |= static_cast<uint32_t>(m.textured);
        S.envtex = EnvTex<R>{};
        // decoded images -> texel arrays of this precision; image textures refer to them
        std::vector<TexRef<R>> image_ref(c->flat.images.size());
        for(size_t k = 0; k < c->flat.images.size(); ++k)
        {
            const FlatImage& im = c->flat.images[k];
            std::vector<Vec4<R>> texels(static_cast<size_t>(im.width) * im.height);
            for(size_t i = 0; i < texels.size(); ++i)
                texels[i] = Vec4<R>{static_cast<R>(im.rgb[3 * i]), static_cast<R>(im.rgb[3 * i + 1]), static_cast<R>(im.rgb[3 * i + 2]), R(0)};
            if((rc = upload(c, c->scene_allocs, texels, &image_ref[k].texels))) return rc;
            image_ref[k].width = im.width;
            image_ref[k].height = im.height;
            image_ref[k].filter = FCGPU_FILTER_BILINEAR;
            image_ref[k].kind = FCGPU_TEXTURE_IMAGE;
        }
        std::vector<TexRef<R>> tex_refs;
        for(const fcgpu_texture& t : c->flat.textures)
        {
            TexRef<R> r{};
            r.kind = t.kind;
            if(t.kind == FCGPU_TEXTURE_CHECKER)
            {
                for(int k = 0; k < 3; ++k)
                {
                    r.checker_a[k] = static_cast<R>(t.color_a[k]);
                    r.checker_b[k] = static_cast<R>(t.color_b[k]);
                }
                r.checker_scale = static_cast<R>(t.scale);
            }
            else
            {
                r = image_ref[t.image];
                r.kind = FCGPU_TEXTURE_IMAGE;
                r.filter = t.filter;
            }
            tex_refs.push_back(r);
        }
        if((rc = upload(c, c->scene_allocs, tex_refs, &S.textures))) return rc;
        if((rc = upload(c, c->scene_allocs, T.matparams, &S.matparams))) return rc;
        if(c->flat.env_kind == FCGPU_ENV_TEXTURE)
        {
            const FlatEnvTexture& e = c->flat.env_texture;
            EnvTex<R>& E = S.envtex;
            E.tex = image_ref[e.image];
            E.tex.filter = e.filter;
            if((rc = upload(c, c->scene_allocs, e.func_x, &E.func_x))) return rc;
            if((rc = upload(c, c->scene_allocs, e.cdf_x, &E.cdf_x))) return rc;
            if((rc = upload(c, c->scene_allocs, e.int_x, &E.int_x))) return rc;
            if((rc = upload(c, c->scene_allocs, e.func_y, &E.func_y))) return rc;
            if((rc = upload(c, c->scene_allocs, e.cdf_y, &E.cdf_y))) return rc;
            E.nx = e.nx;
            E.ny = e.ny;
            E.strength = static_cast<R>(e.strength);
            E.int_y = e.int_y;
            for(int k = 0; k < 12; ++k)
            {
                E.fwd[k] = static_cast<R>(e.fwd[k]);
                E.inv[k] = static_cast<R>(e.inv[k]);
            }
        }
        return FCGPU_OK;
    }

    void drain_spans(fcgpu_ctx* c)
    {
        if(c->spans_busy.empty()) return;
        (void)hipStreamSynchronize(c->stream);
        if(c->stream_b) (void)hipStreamSynchronize(c->stream_b);
        for(auto& s : c->spans_busy)
        {
            float ms = 0.0f;
            if(hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess)
            {
                c->stage_times.ms[s.stage] += ms;
                c->stage_times.launches[s.stage] += 1;
            }
            c->spans_free.push_back(s);
        }
        c->spans_busy.clear();
    }

    // RAII span: records an event pair around the launches issued during its lifetime
    struct StageSpan
    {
        fcgpu_ctx* c;
        fcgpu_ctx::TimedSpan s{};
        bool on{};
        StageSpan(fcgpu_ctx* ctx, int stage) : c{ctx}
        {
            if(!c->profiling) return;
            if(c->spans_busy.size() >= 8192) drain_spans(c);
            if(!c->spans_free.empty())
            {
                s = c->spans_free.back();
                c->spans_free.pop_back();
            }
            else if(hipEventCreate(&s.a) != hipSuccess || hipEventCreate(&s.b) != hipSuccess)
            {
                return;
            }
            s.stage = stage;
            on = hipEventRecord(s.a, c->span_stream ? c->span_stream : c->stream) == hipSuccess;
        }
        ~StageSpan()
        {
            if(!on) return;
            (void)hipEventRecord(s.b, c->span_stream ? c->span_stream : c->stream);
            c->spans_busy.push_back(s);
        }
    };

    bool offset_along_ray(const fcgpu_ctx* c)
    {
        if(c->opt.origin_offset == FCGPU_OFFSET_DEFAULT) return c->opt.precision == FCGPU_PRECISION_F32;
        return c->opt.origin_offset == FCGPU_OFFSET_ALONG_RAY;
    }

    double effective_epsilon(const fcgpu_ctx* c)
    {
        if(c->opt.epsilon > 0.0) return c->opt.epsilon;
        return c->opt.precision == FCGPU_PRECISION_F64 ? 1e-6 : 1e-4;
    }

    template<typename R> int alloc_wavefront(fcgpu_ctx* c, Wavefront<R>& Q, uint32_t cap)
    {
        auto col = [&](Vec4<R>** out) -> int
        {
            void* p = nullptr;
            HIP_TRY(c, hipMalloc(&p, sizeof(Vec4<R>) * static_cast<size_t>(cap)));
            c->queue_allocs.push_back(p);
            *out = static_cast<Vec4<R>*>(p);
            return FCGPU_OK;
        };
        int rc;
        if((rc = col(&Q.ray_c0)) || (rc = col(&Q.ray_c1))) return rc;
        if((rc = col(&Q.sh_c0)) || (rc = col(&Q.sh_c1)) || (rc = col(&Q.sh_c2))) return rc;
        if((rc = col(&Q.hit_c0)) || (rc = col(&Q.hit_c1)) || (rc = col(&Q.li_shadow))) return rc;
        void* p = nullptr;
        HIP_TRY(c, hipMalloc(&p, static_cast<size_t>(cap)));
        c->queue_allocs.push_back(p);
        Q.hit_key = static_cast<uint8_t*>(p);
        // one index list per key that can occur in this scene (lobe types present, last vertex, miss)
        for(int k = 0; k < Q_COUNT; ++k)
        {
            bool needed = k == Q_FINAL || (k == Q_MISS && c->flat.env_kind != FCGPU_ENV_NONE) || (k < L_COUNT && (c->flat.lobe_mask >> k & 1u));
            Q.hit_idx[k] = nullptr;
            if(!needed) continue;
            HIP_TRY(c, hipMalloc(&p, sizeof(uint32_t) * static_cast<size_t>(cap)));
            c->queue_allocs.push_back(p);
            Q.hit_idx[k] = static_cast<uint32_t*>(p);
        }
        for(int k = 0; k < 2; ++k)
            if((rc = col(&Q.st_c0[k])) || (rc = col(&Q.st_c1[k])) || (rc = col(&Q.st_c2[k]))) return rc;
        Q.counters = c->counters;
        Q.stats = c->stats;
        Q.timeline = nullptr;
        if(std::getenv("FCGPU_TIMELINE"))
        {
            c->timeline_words = static_cast<size_t>(c->grid) * 4u * TIMELINE_WORDS;
            HIP_TRY(c, hipMalloc(&p, sizeof(unsigned long long) * c->timeline_words));
            c->queue_allocs.push_back(p);
            HIP_TRY(c, hipMemset(p, 0, sizeof(unsigned long long) * c->timeline_words));
            Q.timeline = static_cast<unsigned long long*>(p);
            c->timeline = Q.timeline;
        }
        Q.dec_hit = Q.dec_flags = nullptr;
        Q.dec_stride = 0;
        Q.capacity = cap;
        return FCGPU_OK;
    }

    // Queues are sized to what the pass needs (at most the configured cap) and only ever grow. When the device cannot
    // hold them (other tenants, fp64 at the default cap: ~57 GB) the batch size is halved down to 2^16 paths before
    // FCGPU_E_NOMEM is returned: smaller batches cost speed (longer tails), never correctness.
    int ensure_queues(fcgpu_ctx* c, uint64_t paths_in_pass)
    {
        const uint32_t limit = c->opt.max_paths_in_flight ? c->opt.max_paths_in_flight : DEFAULT_PATHS_IN_FLIGHT;
        if(c->slots > 1) paths_in_pass = (paths_in_pass + 1) / 2;  // two batches in flight share the pass
        uint32_t cap = static_cast<uint32_t>(std::min<uint64_t>(limit, std::max<uint64_t>(paths_in_pass, 256)));
        if(c->queue_precision == c->opt.precision && c->queue_limit == limit && (c->queue_capacity >= cap || c->queue_shrunk)) return FCGPU_OK;
        c->queue_limit = limit;
        c->queue_shrunk = false;
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if(c->stream_b) HIP_TRY(c, hipStreamSynchronize(c->stream_b));
        free_pool(c->queue_allocs);
        c->queue_precision = -1;
        int rc;
        for(;;)
        {
            rc = c->opt.precision == FCGPU_PRECISION_F64 ? alloc_wavefront(c, c->q64, cap) : alloc_wavefront(c, c->q32, cap);
            if(!rc && c->slots > 1)
            {
                rc = c->opt.precision == FCGPU_PRECISION_F64 ? alloc_wavefront(c, c->q64b, cap) : alloc_wavefront(c, c->q32b, cap);
                c->q64b.counters = c->counters_b;
                c->q32b.counters = c->counters_b;
            }
            if(rc != FCGPU_E_NOMEM || cap <= (1u << 16)) break;
            free_pool(c->queue_allocs);
            cap = std::max(cap / 2, 1u << 16);
            c->queue_shrunk = true;  // the device is full: later passes reuse these queues instead of asking for more
        }
        if(rc)
        {
            free_pool(c->queue_allocs);
            return rc;
        }
        c->queue_precision = c->opt.precision;
        c->queue_capacity = cap;
        return FCGPU_OK;
    }

    // Rows of a pass dealt in bands (fcgpu_render_bands): band b of `rows` film rows belongs to this pass when
    // b % stride == first. rows == 0: a plain tile.
    struct Bands
    {
        uint32_t rows{0}, first{0}, stride{1};
    };

    template<typename R>
    int run_batches(fcgpu_ctx* c, SceneDev<R> S, const Wavefront<R>& Qa, const Wavefront<R>& Qb, const fcgpu_tile& tile, const Bands& bands,
                    uint32_t sample_begin, uint32_t sample_count, int integrator, int max_len, int visible_env)
    {
        S.epsilon = static_cast<R>(effective_epsilon(c));
        S.offset_along_ray = offset_along_ray(c) ? 1u : 0u;
        const uint64_t P = static_cast<uint64_t>(tile.w) * static_cast<uint64_t>(tile.h);
        const uint64_t cap = Qa.capacity;
        Wavefront<R> Qlog = Qa;
        if(c->decision_log)
        {
            // parity hook: one word pair per (bounce, path) of THIS pass, which must be one batch so that path ids are unique
            if(P * sample_count > cap || c->slots > 1) return fail(c, FCGPU_E_LIMIT, "decision log: the pass must fit one batch (and FCGPU_STREAMS=1)");
            if(c->dec_hit) (void)hipFree(c->dec_hit);
            if(c->dec_flags) (void)hipFree(c->dec_flags);
            c->dec_hit = c->dec_flags = nullptr;
            c->dec_bounces = static_cast<uint32_t>(max_len);
            c->dec_paths = P * sample_count;
            const size_t bytes = sizeof(uint32_t) * c->dec_paths * c->dec_bounces;
            HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->dec_hit), bytes));
            if(const hipError_t e2 = hipMalloc(reinterpret_cast<void**>(&c->dec_flags), bytes); e2 != hipSuccess)
            {
                (void)hipFree(c->dec_hit);  // both or none: fcgpu_read_decision_log tests dec_hit
                c->dec_hit = c->dec_flags = nullptr;
                return hip_fail(c, e2, "hipMalloc(decision log)");
            }
            HIP_TRY(c, hipMemsetAsync(c->dec_hit, 0xFF, bytes, c->stream));  // 0xFFFFFFFF = the path had ended before
            HIP_TRY(c, hipMemsetAsync(c->dec_flags, 0, bytes, c->stream));
            Qlog.dec_hit = c->dec_hit;
            Qlog.dec_flags = c->dec_flags;
            Qlog.dec_stride = static_cast<uint32_t>(c->dec_paths);
        }
        // Two batch slots (FCGPU_STREAMS=2): consecutive batches run on two streams with their own queues,
        // so the memory-bound shade kernels of one batch can share the chip with the VALU-bound traversal
        // of the other. Film adds stay ordered: every accumulate waits for the previous batch's.
        uint32_t batch_index = 0;
        if(c->slots > 1)
        {
            HIP_TRY(c, hipEventRecord(c->ev_fork, c->stream));
            HIP_TRY(c, hipStreamWaitEvent(c->stream_b, c->ev_fork, 0));
        }

        Batch B{};
        B.x0 = tile.x0;
        B.y0 = tile.y0;
        B.tw = tile.w;
        B.div_tw = make_fastdiv(static_cast<uint32_t>(tile.w));
        B.div_band = make_fastdiv(bands.rows ? bands.rows : 1u);
        B.band_skip = bands.rows ? (bands.stride - 1u) * bands.rows : 0u;
        B.sampler = c->sampler;
        B.max_len = max_len;
        B.integrator = integrator;
        B.visible_env = visible_env;
        B.W = c->flat.width;
        B.film = c->film;
        B.finish_min = c->finish_min;
        B.descend_min = c->descend_min;
        B.finish_min_shadow = c->finish_min_shadow;
        B.descend_min_shadow = c->descend_min_shadow;

        auto run_one = [&](uint32_t pix0, uint32_t npix, uint32_t s0, uint32_t nsamp)
        {
            const int slot = c->slots > 1 ? static_cast<int>(batch_index & 1u) : 0;
            const Wavefront<R>& Q = slot ? Qb : Qlog;
            LaunchCfg L{slot ? c->stream_b : c->stream, c->grid, c->lds_pad_extend, c->lds_pad_shadow, c->stream_grid};
            c->span_stream = L.stream;
            // two batches in flight traverse at the same time: each slot has its own spill columns
            if(c->stack_spill) S.stack_spill = c->stack_spill + static_cast<size_t>(slot) * c->stack_spill_words;
            B.pix0 = pix0;
            B.npix = npix;
            B.div_npix = make_fastdiv(npix);
            B.s0 = s0;
            B.nsamp = nsamp;
            B.parity = 0;
            B.h = 1;
            {
                StageSpan sp{c, FCGPU_STAGE_GENERATE};
                launch_reset_counters(L, Q.counters, 0);
                launch_generate(L, S, Q, B);
            }
            c->launches += 2;
            for(int h = 1; h <= max_len; ++h)
            {
                B.h = h;
                if(h > 1)
                {
                    launch_reset_counters(L, Q.counters, B.parity);
                    c->launches += 1;
                }
                {
                    StageSpan sp{c, FCGPU_STAGE_EXTEND};
                    launch_extend(L, S, Q, B);
                }
                {
                    StageSpan sp{c, FCGPU_STAGE_SORT};
                    launch_sort(L, Q, B);
                }
                c->launches += 2;
                {
                    StageSpan sp{c, FCGPU_STAGE_SHADE};
                    if(h < max_len)
                    {
                        for(int k = 0; k < L_COUNT; ++k)
                        {
                            if(c->flat.lobe_mask >> k & 1u)
                            {
                                launch_shade(L, S, Q, B, k);
                                c->launches += 1;
                            }
                        }
                    }
                    else
                    {
                        launch_shade(L, S, Q, B, Q_FINAL);
                        c->launches += 1;
                    }
                    if(c->flat.env_kind != FCGPU_ENV_NONE)
                    {
                        launch_shade(L, S, Q, B, Q_MISS);
                        c->launches += 1;
                    }
                }
                if(h < max_len && integrator == FCGPU_INTEGRATOR_MIS)
                {
                    StageSpan sp{c, FCGPU_STAGE_SHADOW};
                    launch_shadow(L, S, Q, B);
                    c->launches += 1;
                }
                B.parity ^= 1;
            }
            if(c->slots > 1 && batch_index > 0) (void)hipStreamWaitEvent(L.stream, c->ev_acc[slot ^ 1], 0);
            {
                StageSpan sp{c, FCGPU_STAGE_ACCUMULATE};
                launch_accumulate(L, S, Q, B);
            }
            if(c->slots > 1) (void)hipEventRecord(c->ev_acc[slot], L.stream);
            c->launches += 1;
            batch_index += 1;
        };

        if(P >= cap)
        {
            // pixel ranges of one sample
            for(uint32_t s = 0; s < sample_count; ++s)
                for(uint64_t p0 = 0; p0 < P; p0 += cap)
                    run_one(static_cast<uint32_t>(p0), static_cast<uint32_t>(std::min<uint64_t>(cap, P - p0)), sample_begin + s, 1);
        }
        else
        {
            // whole samples over all pixels of the tile
            uint32_t per = static_cast<uint32_t>(cap / P);
            for(uint32_t s = 0; s < sample_count; s += per)
                run_one(0, static_cast<uint32_t>(P), sample_begin + s, std::min(per, sample_count - s));
        }
        c->span_stream = nullptr;
        if(c->slots > 1)
        {
            HIP_TRY(c, hipEventRecord(c->ev_join, c->stream_b));
            HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
        }
        hipError_t e = hipGetLastError();
        if(e != hipSuccess) return hip_fail(c, e, "kernel launch");
        c->samples += P * sample_count;
        return FCGPU_OK;
    }

    template<typename R>
    int run_trace(fcgpu_ctx* c, SceneDev<R> S, uint64_t n, const fcgpu_ray* rays, fcgpu_hit* hits, uint8_t* occ)
    {
        S.epsilon = static_cast<R>(effective_epsilon(c));
        LaunchCfg L{c->stream, c->grid};
        launch_trace(L, S, n, rays, hits, occ, c->stats);
        c->launches += 1;
        hipError_t e = hipGetLastError();
        if(e != hipSuccess) return hip_fail(c, e, "trace launch");
        return FCGPU_OK;
    }

    int trace_device(fcgpu_ctx* c, uint64_t n, const fcgpu_ray* rays, fcgpu_hit* hits, uint8_t* occ)
    {
        return guarded(c, [&]() -> int {
            if(!c) return FCGPU_E_INVALID;
            if(!c->has_scene) return fail(c, FCGPU_E_NOSCENE, "no scene uploaded");
            if(n == 0) return FCGPU_OK;
            if(!rays || (!hits && !occ)) return fail(c, FCGPU_E_INVALID, "null buffer");
            HIP_TRY(c, hipSetDevice(c->device));
            return c->opt.precision == FCGPU_PRECISION_F64 ? run_trace(c, c->s64, n, rays, hits, occ) : run_trace(c, c->s32, n, rays, hits, occ);
        });
    }

    int trace_host(fcgpu_ctx* c, uint64_t n, const fcgpu_ray* rays, fcgpu_hit* hits, uint8_t* occ)
    {
        return guarded(c, [&]() -> int {
            if(!c) return FCGPU_E_INVALID;
            if(!c->has_scene) return fail(c, FCGPU_E_NOSCENE, "no scene uploaded");
            if(n == 0) return FCGPU_OK;
            if(!rays || (!hits && !occ)) return fail(c, FCGPU_E_INVALID, "null buffer");
            HIP_TRY(c, hipSetDevice(c->device));
            fcgpu_ray* d_rays = nullptr;
            void* d_out = nullptr;
            size_t out_bytes = hits ? sizeof(fcgpu_hit) * n : n;
            HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&d_rays), sizeof(fcgpu_ray) * n));
            hipError_t e = hipMalloc(&d_out, out_bytes);
            if(e != hipSuccess)
            {
                (void)hipFree(d_rays);
                return hip_fail(c, e, "hipMalloc");
            }
            int rc = FCGPU_OK;
            e = hipMemcpyAsync(d_rays, rays, sizeof(fcgpu_ray) * n, hipMemcpyHostToDevice, c->stream);
            if(e == hipSuccess) rc = trace_device(c, n, d_rays, hits ? static_cast<fcgpu_hit*>(d_out) : nullptr, hits ? nullptr : static_cast<uint8_t*>(d_out));
            if(e == hipSuccess && rc == FCGPU_OK) e = hipMemcpyAsync(hits ? static_cast<void*>(hits) : static_cast<void*>(occ), d_out, out_bytes, hipMemcpyDeviceToHost, c->stream);
            if(e == hipSuccess && rc == FCGPU_OK) e = hipStreamSynchronize(c->stream);
            (void)hipFree(d_rays);
            (void)hipFree(d_out);
            if(rc != FCGPU_OK) return rc;
            if(e != hipSuccess) return hip_fail(c, e, "trace copy");
            return FCGPU_OK;
        });
    }
    // shared tail of fcgpu_render_pass / fcgpu_render_bands: `tile` is validated by the caller (with bands: x0 = 0, full
    // width, y0 = first owned row, h = number of owned rows)
    int render_rows(fcgpu_ctx* c, const fcgpu_tile& tile, const Bands& bands, uint32_t sample_begin, uint32_t sample_count, int integrator,
                    int max_path_length, int visible_env)
    {
        if(integrator != FCGPU_INTEGRATOR_BSDF && integrator != FCGPU_INTEGRATOR_MIS) return fail(c, FCGPU_E_INVALID, "unknown integrator");
        if(max_path_length < 1 || max_path_length > 1000) return fail(c, FCGPU_E_INVALID, "max_path_length out of range");
        if(static_cast<uint64_t>(sample_begin) + sample_count > (1u << 24)) return fail(c, FCGPU_E_LIMIT, "sample index >= 2^24");
        if(sample_count == 0) return FCGPU_OK;
        HIP_TRY(c, hipSetDevice(c->device));
        int rc = ensure_queues(c, static_cast<uint64_t>(tile.w) * static_cast<uint64_t>(tile.h) * sample_count);
        if(rc) return rc;
        // rows of the film this context has written since its last clear (fcgpu_reduce_film sends only those)
        fcgpu_ctx::Coverage& cov = c->cover;
        const bool same_bands = cov.banded && cov.rows == bands.rows && cov.first == bands.first && cov.stride == bands.stride;
        cov.banded = bands.rows != 0 && (!cov.any || same_bands);
        cov.any = true;
        if(cov.banded)
        {
            cov.rows = bands.rows;
            cov.first = bands.first;
            cov.stride = bands.stride;
        }
        if(c->opt.precision == FCGPU_PRECISION_F64)
            return run_batches(c, c->s64, c->q64, c->q64b, tile, bands, sample_begin, sample_count, integrator, max_path_length, visible_env);
        return run_batches(c, c->s32, c->q32, c->q32b, tile, bands, sample_begin, sample_count, integrator, max_path_length, visible_env);
    }
}

extern "C"
{
    int fcgpu_version(void) { return 1005; }

    int fcgpu_check_scene(const fcgpu_scene_desc* d, uint32_t bvh_builder, fcgpu_scene_stats* out)
    {
        try
        {
            const bool wide8 = (bvh_builder & FCGPU_BVH_WIDE8) != 0;
            bvh_builder &= ~static_cast<uint32_t>(FCGPU_BVH_WIDE8);
            if(!d || (bvh_builder != FCGPU_BVH_HOST_SAH && bvh_builder != FCGPU_BVH_HOST_REFERENCE))
            {
                g_create_error = d ? "fcgpu_check_scene builds on the host: FCGPU_BVH_HOST_SAH or FCGPU_BVH_HOST_REFERENCE" : "null scene";
                return FCGPU_E_INVALID;
            }
            FlatScene flat;
            std::string err;
            const int rc = flatten_scene(*d, flat, err, nullptr, bvh_builder == FCGPU_BVH_HOST_REFERENCE, wide8 ? 8 : 4);
            if(rc != FCGPU_OK)
            {
                g_create_error = err;
                return rc;
            }
            if(out)
            {
                *out = fcgpu_scene_stats{};
                out->primitives = flat.prim_info.size();
                out->triangles = flat.tris.size();
                out->binary_nodes = flat.binary_nodes;
                out->wide_nodes = wide8 ? flat.nodes8.size() : flat.nodes.size();
                out->max_stack = flat.max_stack;
                out->lobe_mask = flat.lobe_mask;
            }
            return FCGPU_OK;
        }
        catch(const std::bad_alloc&)
        {
            return FCGPU_E_NOMEM;
        }
        catch(...)
        {
            return FCGPU_E_INVALID;
        }
    }

    int fcgpu_device_count(int* count)
    {
        if(!count) return FCGPU_E_INVALID;
        *count = 0;
        const hipError_t e = hipGetDeviceCount(count);
        if(e != hipSuccess)
        {
            g_create_error = std::string("hipGetDeviceCount: ") + hipGetErrorString(e);
            return FCGPU_E_HIP;
        }
        return FCGPU_OK;
    }

    int fcgpu_create(int device, fcgpu_ctx** out)
    {
        if(!out)
        {
            g_create_error = "null output pointer";
            return FCGPU_E_INVALID;
        }
        *out = nullptr;
        int count = 0;
        hipError_t e = hipGetDeviceCount(&count);
        if(e != hipSuccess || device < 0 || device >= count)
        {
            g_create_error = e != hipSuccess ? std::string("hipGetDeviceCount: ") + hipGetErrorString(e) : "no such HIP device";
            return e != hipSuccess ? FCGPU_E_HIP : FCGPU_E_INVALID;
        }
        fcgpu_ctx* c = new(std::nothrow) fcgpu_ctx{};
        if(!c)
        {
            g_create_error = "out of host memory";
            return FCGPU_E_NOMEM;
        }
        c->device = device;
        c->opt.precision = FCGPU_PRECISION_F32;
        c->opt.sort_materials = 1;
        auto bail = [&](hipError_t err, const char* what)
        {
            g_create_error = std::string(what) + ": " + hipGetErrorString(err);
            if(c->counters) (void)hipFree(c->counters);
            if(c->stats) (void)hipFree(c->stats);
            if(c->stream) (void)hipStreamDestroy(c->stream);
            delete c;
            return FCGPU_E_HIP;
        };
        if((e = hipSetDevice(device)) != hipSuccess) return bail(e, "hipSetDevice");
        if((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
        if((e = hipMalloc(reinterpret_cast<void**>(&c->counters), sizeof(QueueCounters))) != hipSuccess) return bail(e, "hipMalloc");
        if((e = hipMalloc(reinterpret_cast<void**>(&c->stats), sizeof(unsigned long long) * ST_COUNT)) != hipSuccess) return bail(e, "hipMalloc");
        if((e = hipMemset(c->counters, 0, sizeof(QueueCounters))) != hipSuccess) return bail(e, "hipMemset");
        if((e = hipMemset(c->stats, 0, sizeof(unsigned long long) * ST_COUNT)) != hipSuccess) return bail(e, "hipMemset");
        // no scratch address (a device index beyond the table, a failed symbol lookup) only disables fcgpu_trace_closest_noalloc,
        // which then reports it; every other entry point works without
        if(noalloc_scratch_for_device(device, &c->noalloc_scratch) != hipSuccess)
        {
            c->noalloc_scratch = nullptr;
            (void)hipGetLastError();
        }
        hipDeviceProp_t prop{};
        if((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return bail(e, "hipGetDeviceProperties");
        c->grid = std::max(1, prop.multiProcessorCount) * 8;
        // the grid-stride streaming kernels run 128 blocks per CU: blocks that come and go keep loads and stores of many
        // trips in flight at once (a plain column copy of the queues' shape streams 4.4 TB/s from 8 blocks per CU and
        // 5.05 TB/s from 128, tools/micro/stream_bw.hip; k_shade -3.5 %, k_generate -25 %)
        c->stream_grid = std::max(1, prop.multiProcessorCount) * 128;
        if(const char* e = std::getenv("FCGPU_LDS_PAD_EXTEND")) c->lds_pad_extend = std::max(0, std::atoi(e));
        if(const char* e = std::getenv("FCGPU_LDS_PAD_SHADOW")) c->lds_pad_shadow = std::max(0, std::atoi(e));
        if(const char* e = std::getenv("FCGPU_FINISH_MIN")) c->finish_min = std::max(1, std::atoi(e));
        if(const char* e = std::getenv("FCGPU_DESCEND_MIN")) c->descend_min = std::max(0, std::atoi(e));
        if(const char* e = std::getenv("FCGPU_FINISH_MIN_SHADOW")) c->finish_min_shadow = std::max(1, std::atoi(e));
        if(const char* e = std::getenv("FCGPU_DESCEND_MIN_SHADOW")) c->descend_min_shadow = std::max(0, std::atoi(e));
        if(const char* e = std::getenv("FCGPU_GRID_MULT")) c->grid = std::max(1, prop.multiProcessorCount) * std::max(1, std::atoi(e));
        if(const char* e = std::getenv("FCGPU_STREAM_GRID_MULT")) c->stream_grid = std::max(1, prop.multiProcessorCount) * std::max(1, std::atoi(e));
        if(const char* e = std::getenv("FCGPU_STREAMS")) c->slots = std::atoi(e) >= 2 ? 2 : 1;
        if(c->slots > 1)
        {
            if((e = hipStreamCreateWithFlags(&c->stream_b, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
            if((e = hipMalloc(reinterpret_cast<void**>(&c->counters_b), sizeof(QueueCounters))) != hipSuccess) return bail(e, "hipMalloc");
            if((e = hipMemset(c->counters_b, 0, sizeof(QueueCounters))) != hipSuccess) return bail(e, "hipMemset");
            for(hipEvent_t* ev : {&c->ev_fork, &c->ev_join, &c->ev_acc[0], &c->ev_acc[1]})
                if((e = hipEventCreateWithFlags(ev, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
        }
        *out = c;
        return FCGPU_OK;
    }

    int fcgpu_destroy(fcgpu_ctx* c)
    {
        if(!c) return FCGPU_E_INVALID;
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        if(c->timeline)
        {
            if(const char* path = std::getenv("FCGPU_TIMELINE"))
            {
                std::vector<unsigned long long> host(c->timeline_words);
                if(hipMemcpy(host.data(), c->timeline, sizeof(unsigned long long) * host.size(), hipMemcpyDeviceToHost) == hipSuccess)
                    if(FILE* f = std::fopen(path, "wb"))
                    {
                        std::fwrite(host.data(), sizeof(unsigned long long), host.size(), f);
                        std::fclose(f);
                    }
            }
        }
        drain_spans(c);
        for(auto& s : c->spans_free)
        {
            (void)hipEventDestroy(s.a);
            (void)hipEventDestroy(s.b);
        }
        free_pool(c->scene_allocs);
        free_pool(c->queue_allocs);
        if(c->film) (void)hipFree(c->film);
        if(c->reduce_staging) (void)hipFree(c->reduce_staging);
        if(c->ev_reduce) (void)hipEventDestroy(c->ev_reduce);
        if(c->dec_hit) (void)hipFree(c->dec_hit);
        if(c->dec_flags) (void)hipFree(c->dec_flags);
        (void)hipFree(c->counters);
        if(c->slots > 1)
        {
            (void)hipStreamSynchronize(c->stream_b);
            (void)hipFree(c->counters_b);
            (void)hipStreamDestroy(c->stream_b);
            for(hipEvent_t ev : {c->ev_fork, c->ev_join, c->ev_acc[0], c->ev_acc[1]}) (void)hipEventDestroy(ev);
        }
        (void)hipFree(c->stats);
        (void)hipStreamDestroy(c->stream);
        delete c;
        return FCGPU_OK;
    }

    const char* fcgpu_last_error(const fcgpu_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

    int fcgpu_set_options(fcgpu_ctx* c, const fcgpu_options* opt)
    {
        return guarded(c, [&]() -> int {
            if(!c || !opt) return fail(c, FCGPU_E_INVALID, "null argument");
            if(opt->precision != FCGPU_PRECISION_F32 && opt->precision != FCGPU_PRECISION_F64) return fail(c, FCGPU_E_INVALID, "unknown precision");
            if(opt->max_paths_in_flight != 0 && opt->max_paths_in_flight < 256) return fail(c, FCGPU_E_INVALID, "max_paths_in_flight < 256");
            if(opt->bvh_builder > FCGPU_BVH_HOST_REFERENCE) return fail(c, FCGPU_E_INVALID, "unknown bvh_builder");
            if(opt->bvh_width != 0 && opt->bvh_width != 4 && opt->bvh_width != 8) return fail(c, FCGPU_E_INVALID, "bvh_width must be 0 (default), 4 or 8");
            if(opt->origin_offset > FCGPU_OFFSET_ALONG_RAY) return fail(c, FCGPU_E_INVALID, "unknown origin_offset");
            if(opt->sort_materials != 1) return fail(c, FCGPU_E_UNSUPPORTED, "sort_materials must be 1: the shade kernels are one instantiation per lobe type");
            c->opt = *opt;
            return FCGPU_OK;
        });
    }

    int fcgpu_set_sampler(fcgpu_ctx* c, uint64_t seed)
    {
        if(!c) return FCGPU_E_INVALID;
        c->sampler = fcgpu_sampler_random(seed);
        return FCGPU_OK;
    }

    int fcgpu_set_sampler_stratified(fcgpu_ctx* c, uint64_t seed, uint32_t sample_count)
    {
        return guarded(c, [&]() -> int {
            if(!c) return FCGPU_E_INVALID;
            if(sample_count == 0 || sample_count > (1u << 24)) return fail(c, FCGPU_E_INVALID, "stratified sample_count must be in [1, 2^24]");
            c->sampler = fcgpu_sampler_stratified(seed, sample_count);
            return FCGPU_OK;
        });
    }

    int fcgpu_upload_scene(fcgpu_ctx* c, const fcgpu_scene_desc* d)
    {
        return guarded(c, [&]() -> int {
            if(!c || !d) return fail(c, FCGPU_E_INVALID, "null argument");
            HIP_TRY(c, hipSetDevice(c->device));
            FlatScene flat;
            std::string err;
            const HierarchyBuilder on_device = [c](const std::vector<float>& boxes, LbvhTree& tree, std::string& e) {
                return lbvh_build(c->stream, boxes, tree, e);
            };
            const auto t_build = std::chrono::steady_clock::now();
            const int width = c->opt.bvh_width == 8 ? 8 : 4;
            int rc = flatten_scene(*d, flat, err, c->opt.bvh_builder == FCGPU_BVH_DEVICE_LBVH ? &on_device : nullptr,
                                   c->opt.bvh_builder == FCGPU_BVH_HOST_REFERENCE, width);
            if(rc == FCGPU_OK && width == 8 && flat.max_stack > static_cast<uint32_t>(STACK8_CAP))
            {
                rc = FCGPU_E_LIMIT;
                err = "bvh_width 8: the hierarchy needs more than " + std::to_string(STACK8_CAP) + " group entries on the traversal stack; use bvh_width 4";
            }
            c->flatten_us = static_cast<uint64_t>(std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_build).count());
            if(rc != FCGPU_OK) return fail(c, rc, err);

            HIP_TRY(c, hipStreamSynchronize(c->stream));
            free_pool(c->scene_allocs);
            free_pool(c->queue_allocs);  // hit queues depend on the lobe types of the scene
            c->queue_precision = -1;
            c->has_scene = false;
            c->flat = std::move(flat);

            c->stack_spill = nullptr;
            c->stack_spill_words = 0;
            c->stack_spill_entries = 0;
            // The kernels without spill code stop pushing once fewer than three free LDS entries are left (one test per
            // node step instead of one per push), which is exact only while the worst case is <= STACK_CAP - 2: any
            // deeper hierarchy, including the ones that would just fit, runs the SPILL instantiations.
            // The fp32 shadow kernel keeps SHADOW_STACK_CAP <= STACK_CAP entries in LDS (fc_types.h), so the columns are
            // allocated and sized for it; which kernels run their SPILL instantiation is decided per kernel from
            // SceneDev::max_stack (fc_kernels.hip: launch_*).
            if(width == 4 && c->flat.max_stack > static_cast<uint32_t>(SHADOW_STACK_CAP - 2))
            {
                // deep BVH: the part of the worst-case stack beyond the LDS entries lives in global memory,
                // one column per lane of the persistent grid (and per batch slot)
                void* p = nullptr;
                const size_t entries = c->flat.max_stack > static_cast<uint32_t>(SHADOW_STACK_CAP) ? c->flat.max_stack - static_cast<uint32_t>(SHADOW_STACK_CAP) : 1u;
                c->stack_spill_entries = static_cast<uint32_t>(entries);
                c->stack_spill_words = entries * static_cast<size_t>(c->grid) * 256u;
                HIP_TRY(c, hipMalloc(&p, c->stack_spill_words * sizeof(uint32_t) * static_cast<size_t>(c->slots)));
                c->scene_allocs.push_back(p);
                c->stack_spill = static_cast<uint32_t*>(p);
            }
            SceneDev<float> a{};
            if((rc = upload(c, c->scene_allocs, c->flat.nodes, &a.nodes))) return rc;
            if(width == 8 && (rc = upload(c, c->scene_allocs, c->flat.nodes8, &a.nodes8))) return rc;
            if((rc = upload(c, c->scene_allocs, c->flat.tris, &a.tris))) return rc;
            if((rc = upload(c, c->scene_allocs, c->flat.tri_attrs, &a.tri_attrs))) return rc;
            if((rc = upload(c, c->scene_allocs, c->flat.prim_info, &a.prim_info))) return rc;
            if((rc = upload(c, c->scene_allocs, c->flat.entities, &a.entities))) return rc;
            if((rc = upload(c, c->scene_allocs, c->flat.materials, &a.materials))) return rc;
            if((rc = upload(c, c->scene_allocs, c->flat.light_cdf, &a.light_cdf))) return rc;
            SceneDev<double> b{};
            b.nodes = a.nodes;
            b.nodes8 = a.nodes8;
            b.tris = a.tris;
            b.tri_attrs = a.tri_attrs;
            b.prim_info = a.prim_info;
            b.entities = a.entities;
            b.materials = a.materials;
            b.light_cdf = a.light_cdf;
            if((rc = fill_scene(c, a, c->flat.f32))) return rc;
            if((rc = fill_scene(c, b, c->flat.f64))) return rc;
            c->s32 = a;
            c->s64 = b;

            if(c->film) (void)hipFree(c->film);
            c->film = nullptr;
            c->film_elems = 3ull * static_cast<uint64_t>(c->flat.width) * static_cast<uint64_t>(c->flat.height);
            HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->film), sizeof(double) * c->film_elems));
            HIP_TRY(c, hipMemset(c->film, 0, sizeof(double) * c->film_elems));
            if(c->reduce_staging) (void)hipFree(c->reduce_staging);  // sized to the previous film
            c->reduce_staging = nullptr;
            c->cover = fcgpu_ctx::Coverage{};
            c->samples = 0;
            c->has_scene = true;
            return FCGPU_OK;
        });
    }

    int fcgpu_render_pass(fcgpu_ctx* c, const fcgpu_tile* tile, uint32_t sample_begin, uint32_t sample_count, int integrator,
                          int max_path_length, int visible_infinity_area_light)
    {
        return guarded(c, [&]() -> int {
            if(!c) return FCGPU_E_INVALID;
            if(!c->has_scene) return fail(c, FCGPU_E_NOSCENE, "no scene uploaded");
            fcgpu_tile full{0, 0, c->flat.width, c->flat.height};
            if(!tile) tile = &full;
            if(tile->w <= 0 || tile->h <= 0 || tile->x0 < 0 || tile->y0 < 0 || tile->x0 + tile->w > c->flat.width || tile->y0 + tile->h > c->flat.height)
                return fail(c, FCGPU_E_INVALID, "tile outside the film");
            return render_rows(c, *tile, Bands{}, sample_begin, sample_count, integrator, max_path_length, visible_infinity_area_light);
        });
    }

    int fcgpu_render_bands(fcgpu_ctx* c, uint32_t band_rows, uint32_t band_first, uint32_t band_stride, uint32_t sample_begin,
                           uint32_t sample_count, int integrator, int max_path_length, int visible_infinity_area_light)
    {
        return guarded(c, [&]() -> int {
            if(!c) return FCGPU_E_INVALID;
            if(!c->has_scene) return fail(c, FCGPU_E_NOSCENE, "no scene uploaded");
            if(band_rows == 0 || band_stride == 0 || band_first >= band_stride || band_rows > (1u << 20) || band_stride > (1u << 20))
                return fail(c, FCGPU_E_INVALID, "band_rows and band_stride must be positive, band_first < band_stride");
            // rows this context owns: band b = y / band_rows with b % band_stride == band_first, packed in order
            const uint64_t H = static_cast<uint64_t>(c->flat.height);
            uint64_t owned = 0;
            for(uint64_t y0 = static_cast<uint64_t>(band_first) * band_rows; y0 < H; y0 += static_cast<uint64_t>(band_stride) * band_rows)
                owned += std::min<uint64_t>(band_rows, H - y0);
            if(owned == 0) return FCGPU_OK;  // more GPUs than bands: nothing to do here
            const fcgpu_tile rows{0, static_cast<int32_t>(band_first * band_rows), c->flat.width, static_cast<int32_t>(owned)};
            return render_rows(c, rows, Bands{band_rows, band_first, band_stride}, sample_begin, sample_count, integrator, max_path_length,
                               visible_infinity_area_light);
        });
    }

    int fcgpu_trace_closest(fcgpu_ctx* c, uint64_t n, const fcgpu_ray* rays, fcgpu_hit* hits)
    {
        if(c && !hits && n) return fail(c, FCGPU_E_INVALID, "null hits");
        return trace_host(c, n, rays, hits, nullptr);
    }
    int fcgpu_trace_any(fcgpu_ctx* c, uint64_t n, const fcgpu_ray* rays, uint8_t* occluded)
    {
        if(c && !occluded && n) return fail(c, FCGPU_E_INVALID, "null output");
        return trace_host(c, n, rays, nullptr, occluded);
    }
    int fcgpu_trace_closest_device(fcgpu_ctx* c, uint64_t n, const fcgpu_ray* rays_dev, fcgpu_hit* hits_dev)
    {
        if(c && !hits_dev && n) return fail(c, FCGPU_E_INVALID, "null hits");
        return trace_device(c, n, rays_dev, hits_dev, nullptr);
    }
    int fcgpu_trace_any_device(fcgpu_ctx* c, uint64_t n, const fcgpu_ray* rays_dev, uint8_t* occluded_dev)
    {
        if(c && !occluded_dev && n) return fail(c, FCGPU_E_INVALID, "null output");
        return trace_device(c, n, rays_dev, nullptr, occluded_dev);
    }

    // The no-allocation variant of fcgpu_trace_closest_device (include/fcgpu.h): same kernel, its scratch in the static
    // device array whose per-device address fcgpu_create looked up. No hipMalloc / hipFree on any path.
    int fcgpu_trace_closest_noalloc(fcgpu_ctx* c, uint64_t n, const fcgpu_ray* rays_dev, fcgpu_hit* hits_dev)
    {
        if(!c) return FCGPU_E_INVALID;
        if(!c->has_scene) return fail(c, FCGPU_E_NOSCENE, "no scene uploaded");
        if(n == 0) return FCGPU_OK;
        if(!rays_dev || !hits_dev) return fail(c, FCGPU_E_INVALID, "null buffer");
        if(n > FCGPU_NOALLOC_MAX_RAYS) return fail(c, FCGPU_E_LIMIT, "fcgpu_trace_closest_noalloc: more than FCGPU_NOALLOC_MAX_RAYS (2^30) rays in one call");
        const bool deep = c->flat.max_stack > static_cast<uint32_t>(STACK_CAP - 2);  // k_trace runs its SPILL instantiation (fc_kernels.hip: spills())
        if(c->flat.max_stack > static_cast<uint32_t>(STACK_CAP) + NOALLOC_SPILL_ENTRIES)
            return fail(c, FCGPU_E_LIMIT, "fcgpu_trace_closest_noalloc: the hierarchy needs more than FCGPU_NOALLOC_MAX_STACK (62) traversal-stack entries; use fcgpu_trace_closest_device");
        if(!c->noalloc_scratch) return fail(c, FCGPU_E_HIP, "no scratch address for this device");
        HIP_TRY(c, hipSetDevice(c->device));
        unsigned long long* stats = reinterpret_cast<unsigned long long*>(c->noalloc_scratch);
        // the array keeps the last call's contents: the counters must start at zero (the spill columns need not)
        HIP_TRY(c, hipMemsetAsync(stats, 0, NOALLOC_STATS_BYTES, c->stream));
        const LaunchCfg L{c->stream, static_cast<int>(NOALLOC_LANES / 256u)};
        auto run = [&](auto S) {
            S.epsilon = static_cast<decltype(S.epsilon)>(effective_epsilon(c));
            S.stack_spill = deep ? reinterpret_cast<uint32_t*>(c->noalloc_scratch + NOALLOC_STATS_BYTES) : nullptr;
            S.stack_spill_stride = NOALLOC_LANES;
            S.stack_spill_cap = NOALLOC_SPILL_ENTRIES;  // the static array's columns, not the context's own spill area
            launch_trace(L, S, n, rays_dev, hits_dev, static_cast<uint8_t*>(nullptr), stats);
        };
        if(c->opt.precision == FCGPU_PRECISION_F64)
            run(c->s64);
        else
            run(c->s32);
        launch_fold_stats(L, c->stats, stats);
        c->launches += 2;
        const hipError_t e = hipGetLastError();
        if(e != hipSuccess) return hip_fail(c, e, "trace launch");
        return FCGPU_OK;
    }

    // ------------------------------------------------------------------ parity hook: per-path decision log
    int fcgpu_set_decision_log(fcgpu_ctx* c, int enabled)
    {
        if(!c) return FCGPU_E_INVALID;
        c->decision_log = enabled != 0;
        return FCGPU_OK;
    }

    int fcgpu_read_decision_log(fcgpu_ctx* c, uint32_t* bounces, uint64_t* paths, uint32_t* hit_words, uint32_t* flag_words)
    {
        return guarded(c, [&]() -> int {
            if(!c) return FCGPU_E_INVALID;
            if(!c->dec_hit) return fail(c, FCGPU_E_INVALID, "no decision log: enable it before the render pass");
            if(bounces) *bounces = c->dec_bounces;
            if(paths) *paths = c->dec_paths;
            HIP_TRY(c, hipSetDevice(c->device));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            const size_t bytes = sizeof(uint32_t) * c->dec_paths * c->dec_bounces;
            if(hit_words) HIP_TRY(c, hipMemcpy(hit_words, c->dec_hit, bytes, hipMemcpyDeviceToHost));
            if(flag_words) HIP_TRY(c, hipMemcpy(flag_words, c->dec_flags, bytes, hipMemcpyDeviceToHost));
            return FCGPU_OK;
        });
    }

    // ------------------------------------------------------------------ multi-GPU film reduce (renderer.hpp:125-135)
    // Contributors whose film holds only their bands (fcgpu_render_bands with one pattern since the last clear) send those
    // rows, all at once: their copies run on their own streams into disjoint rows of one staging film on the root's
    // device, each followed by an add over just those rows. Contributors that rendered anything else send the frame, one
    // after the other through the same staging film. Between different devices the copies are direct peer accesses
    // (hipDeviceEnablePeerAccess: xGMI on an MI355X node) where the devices allow it, hipMemcpyPeerAsync otherwise.
    int fcgpu_reduce_film(fcgpu_ctx* const* ctxs, int count, int root)
    {
        fcgpu_ctx* r = (ctxs && count > 0 && root >= 0 && root < count) ? ctxs[root] : nullptr;
        return guarded(r, [&]() -> int {
            if(!r) return FCGPU_E_INVALID;
            for(int i = 0; i < count; ++i)
            {
                if(!ctxs[i]) return fail(r, FCGPU_E_INVALID, "null context in the list");
                if(!ctxs[i]->has_scene) return fail(r, FCGPU_E_NOSCENE, "a context has no scene");
                if(ctxs[i]->film_elems != r->film_elems || ctxs[i]->flat.width != r->flat.width) return fail(r, FCGPU_E_INVALID, "films differ in resolution");
                for(int j = 0; j < i; ++j)
                    if(ctxs[j] == ctxs[i]) return fail(r, FCGPU_E_INVALID, "a context is listed twice");
            }
            if(count == 1) return FCGPU_OK;
            const size_t bytes = sizeof(double) * r->film_elems;
            const uint64_t H = static_cast<uint64_t>(r->flat.height), row_elems = 3ull * static_cast<uint64_t>(r->flat.width);
            const size_t row_bytes = sizeof(double) * row_elems;
            // every contributor has finished rendering before its film is read
            for(int i = 0; i < count; ++i)
            {
                fcgpu_ctx* s = ctxs[i];
                HIP_TRY(r, hipSetDevice(s->device));
                HIP_TRY(r, hipStreamSynchronize(s->stream));
                if(s->stream_b) HIP_TRY(r, hipStreamSynchronize(s->stream_b));
                if(!s->ev_reduce) HIP_TRY(r, hipEventCreateWithFlags(&s->ev_reduce, hipEventDisableTiming));
            }
            // direct access from each contributor's device to the root's memory (the copy runs on the contributor's stream)
            std::vector<char> peer(static_cast<size_t>(count), 1);
            for(int i = 0; i < count; ++i)
            {
                fcgpu_ctx* s = ctxs[i];
                if(i == root || s->device == r->device) continue;
                int can = 0;
                HIP_TRY(r, hipDeviceCanAccessPeer(&can, s->device, r->device));
                peer[static_cast<size_t>(i)] = 0;
                if(!can) continue;
                HIP_TRY(r, hipSetDevice(s->device));
                const hipError_t e = hipDeviceEnablePeerAccess(r->device, 0);
                if(e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled)
                    peer[static_cast<size_t>(i)] = 1;
                (void)hipGetLastError();
            }
            HIP_TRY(r, hipSetDevice(r->device));
            if(!r->reduce_staging) HIP_TRY(r, hipMalloc(reinterpret_cast<void**>(&r->reduce_staging), bytes));  // kept: the next reduce reuses it
            double* staging = r->reduce_staging;

            // copy of `n` bytes at the same offset of two films, on the contributor's stream
            auto copy = [&](fcgpu_ctx* s, bool direct, uint64_t elem_off, size_t n) -> hipError_t
            {
                r->reduce_bytes += n;
                if(direct) return hipMemcpyAsync(staging + elem_off, s->film + elem_off, n, hipMemcpyDeviceToDevice, s->stream);
                return hipMemcpyPeerAsync(staging + elem_off, r->device, s->film + elem_off, s->device, n, s->stream);
            };
            // banded contributors that can go together: one band pattern (rows, stride), distinct `first`
            std::vector<int> together, alone;
            for(int i = 0; i < count; ++i)
            {
                if(i == root) continue;
                const fcgpu_ctx::Coverage& c = ctxs[i]->cover;
                if(!c.any) continue;  // rendered nothing since its last clear: only its sample count (0) counts
                bool ok = c.banded;
                for(int j : together)
                    ok = ok && ctxs[j]->cover.rows == c.rows && ctxs[j]->cover.stride == c.stride && ctxs[j]->cover.first != c.first;
                (ok ? together : alone).push_back(i);
            }
            int rc = FCGPU_OK;
            auto check = [&](hipError_t e, const char* what)
            {
                if(e != hipSuccess && rc == FCGPU_OK) rc = hip_fail(r, e, what);
                return e == hipSuccess;
            };
            for(int i : together)
            {
                fcgpu_ctx* s = ctxs[static_cast<size_t>(i)];
                const fcgpu_ctx::Coverage& c = s->cover;
                const bool direct = s->device == r->device || peer[static_cast<size_t>(i)];
                if(!check(hipSetDevice(s->device), "hipSetDevice")) break;
                const uint64_t period = static_cast<uint64_t>(c.stride) * c.rows;
                uint64_t owned_rows = 0, full_bands = 0, tail_y = 0, tail_rows = 0;
                for(uint64_t y0 = static_cast<uint64_t>(c.first) * c.rows; y0 < H; y0 += period)
                {
                    const uint64_t n = std::min<uint64_t>(c.rows, H - y0);
                    owned_rows += n;
                    if(n == c.rows)
                        full_bands += 1;
                    else
                    {
                        tail_y = y0;
                        tail_rows = n;
                    }
                }
                const uint64_t first_off = static_cast<uint64_t>(c.first) * c.rows * row_elems;
                bool ok = true;
                if(direct && full_bands > 0)
                {
                    // all whole bands of this contributor in one strided copy
                    r->reduce_bytes += full_bands * c.rows * row_bytes;
                    ok = check(hipMemcpy2DAsync(staging + first_off, period * row_bytes, s->film + first_off, period * row_bytes, c.rows * row_bytes,
                                                full_bands, hipMemcpyDeviceToDevice, s->stream), "film band copy");
                }
                else
                {
                    for(uint64_t b = 0; b < full_bands && ok; ++b) ok = check(copy(s, false, first_off + b * period * row_elems, c.rows * row_bytes), "film band copy");
                }
                if(ok && tail_rows) ok = check(copy(s, direct, tail_y * row_elems, tail_rows * row_bytes), "film band copy");
                if(!ok) break;
                check(hipEventRecord(s->ev_reduce, s->stream), "hipEventRecord");
                check(hipMemsetAsync(s->film, 0, bytes, s->stream), "film clear");  // after its copy, in stream order
                if(!check(hipSetDevice(r->device), "hipSetDevice")) break;
                check(hipStreamWaitEvent(r->stream, s->ev_reduce, 0), "hipStreamWaitEvent");
                launch_film_add_bands(LaunchCfg{r->stream, r->grid}, r->film, staging, row_elems, owned_rows, c.rows, c.first, c.stride);
            }
            for(int i : alone)
            {
                if(rc != FCGPU_OK) break;
                fcgpu_ctx* s = ctxs[static_cast<size_t>(i)];
                const bool direct = s->device == r->device || peer[static_cast<size_t>(i)];
                // the staging film is free once the root's previous add has run
                if(!check(hipSetDevice(r->device), "hipSetDevice")) break;
                check(hipEventRecord(r->ev_reduce, r->stream), "hipEventRecord");
                if(!check(hipSetDevice(s->device), "hipSetDevice")) break;
                check(hipStreamWaitEvent(s->stream, r->ev_reduce, 0), "hipStreamWaitEvent");
                if(!check(copy(s, direct, 0, bytes), "film copy")) break;
                check(hipEventRecord(s->ev_reduce, s->stream), "hipEventRecord");
                check(hipMemsetAsync(s->film, 0, bytes, s->stream), "film clear");
                if(!check(hipSetDevice(r->device), "hipSetDevice")) break;
                check(hipStreamWaitEvent(r->stream, s->ev_reduce, 0), "hipStreamWaitEvent");
                launch_film_add(LaunchCfg{r->stream, r->grid}, r->film, staging, r->film_elems);
            }
            // everything has landed before the call returns; the contributors start over: a later render + reduce adds only what is new
            for(int i = 0; i < count; ++i)
            {
                fcgpu_ctx* s = ctxs[i];
                if(hipSetDevice(s->device) == hipSuccess) check(hipStreamSynchronize(s->stream), "hipStreamSynchronize");
                if(i == root || rc != FCGPU_OK) continue;
                r->samples += s->samples;
                s->samples = 0;
                s->cover = fcgpu_ctx::Coverage{};
            }
            (void)hipSetDevice(r->device);
            check(hipGetLastError(), "film add launch");
            r->cover.any = true;
            r->cover.banded = false;  // the root's film now holds rows of every contributor
            return rc;
        });
    }

    // ------------------------------------------------------------------ per-lobe parity hook
    int fcgpu_bsdf_table(fcgpu_ctx* c, const fcgpu_material* material, const double* ng, const double* tangent, const double* ns,
                         const double* bitangent, double eta_a, double eta_b, uint64_t n, const fcgpu_bsdf_query* queries,
                         fcgpu_bsdf_result* results)
    {
        return guarded(c, [&]() -> int {
            if(!c) return FCGPU_E_INVALID;
            if(!material || !ng || !tangent || !ns || !bitangent) return fail(c, FCGPU_E_INVALID, "null argument");
            if(n == 0) return FCGPU_OK;
            if(!queries || !results) return fail(c, FCGPU_E_INVALID, "null buffer");
            if(material->tex_color_a || material->tex_color_b || material->tex_roughness || material->tex_metalness || material->tex_ior || material->tex_normal)
                return fail(c, FCGPU_E_UNSUPPORTED, "fcgpu_bsdf_table takes constant materials (no texture handles)");
            std::vector<LobeRec<float>> l32;
            std::vector<LobeRec<double>> l64;
            std::string err;
            const int lobes = material_lobe_table(*material, l32, l64, err);
            if(lobes <= 0) return fail(c, lobes < 0 ? lobes : FCGPU_E_INVALID, lobes < 0 ? err : "material without lobes");
            HIP_TRY(c, hipSetDevice(c->device));
            std::vector<void*> pool;
            struct Free
            {
                std::vector<void*>& p;
                ~Free() { free_pool(p); }
            } on_exit{pool};
            const bool f64 = c->opt.precision == FCGPU_PRECISION_F64;
            void *d_lobes = nullptr, *d_q = nullptr, *d_out = nullptr;
            const size_t lobe_bytes = f64 ? sizeof(LobeRec<double>) * l64.size() : sizeof(LobeRec<float>) * l32.size();
            HIP_TRY(c, hipMalloc(&d_lobes, lobe_bytes));
            pool.push_back(d_lobes);
            HIP_TRY(c, hipMalloc(&d_q, sizeof(fcgpu_bsdf_query) * n));
            pool.push_back(d_q);
            HIP_TRY(c, hipMalloc(&d_out, sizeof(fcgpu_bsdf_result) * n));
            pool.push_back(d_out);
            HIP_TRY(c, hipMemcpyAsync(d_lobes, f64 ? static_cast<const void*>(l64.data()) : static_cast<const void*>(l32.data()), lobe_bytes, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(c, hipMemcpyAsync(d_q, queries, sizeof(fcgpu_bsdf_query) * n, hipMemcpyHostToDevice, c->stream));
            double frame[12];
            for(int k = 0; k < 3; ++k)
            {
                frame[k] = ng[k];
                frame[3 + k] = tangent[k];
                frame[6 + k] = ns[k];
                frame[9 + k] = bitangent[k];
            }
            const LaunchCfg L{c->stream, c->grid};
            if(f64)
                launch_bsdf_table<double>(L, static_cast<const LobeRec<double>*>(d_lobes), lobes, frame, eta_a, eta_b, n, static_cast<const fcgpu_bsdf_query*>(d_q), static_cast<fcgpu_bsdf_result*>(d_out));
            else
                launch_bsdf_table<float>(L, static_cast<const LobeRec<float>*>(d_lobes), lobes, frame, eta_a, eta_b, n, static_cast<const fcgpu_bsdf_query*>(d_q), static_cast<fcgpu_bsdf_result*>(d_out));
            hipError_t e = hipGetLastError();
            if(e != hipSuccess) return hip_fail(c, e, "bsdf table launch");
            HIP_TRY(c, hipMemcpyAsync(results, d_out, sizeof(fcgpu_bsdf_result) * n, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            return FCGPU_OK;
        });
    }

    int fcgpu_clear_film(fcgpu_ctx* c)
    {
        return guarded(c, [&]() -> int {
            if(!c) return FCGPU_E_INVALID;
            if(!c->has_scene) return fail(c, FCGPU_E_NOSCENE, "no scene uploaded");
            HIP_TRY(c, hipSetDevice(c->device));
            HIP_TRY(c, hipMemsetAsync(c->film, 0, sizeof(double) * c->film_elems, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            c->cover = fcgpu_ctx::Coverage{};
            c->samples = 0;
            return FCGPU_OK;
        });
    }

    int fcgpu_read_film(fcgpu_ctx* c, double* rgb_sum, uint64_t* sample_count)
    {
        return guarded(c, [&]() -> int {
            if(!c) return FCGPU_E_INVALID;
            if(!c->has_scene) return fail(c, FCGPU_E_NOSCENE, "no scene uploaded");
            HIP_TRY(c, hipSetDevice(c->device));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            if(rgb_sum)
            {
                HIP_TRY(c, hipMemcpy(rgb_sum, c->film, sizeof(double) * c->film_elems, hipMemcpyDeviceToHost));
                // beta0 = W*H (perspective_camera.hpp:64-68) is applied here instead of per sample
                const double scale = static_cast<double>(c->flat.width) * static_cast<double>(c->flat.height);
                for(uint64_t i = 0; i < c->film_elems; ++i) rgb_sum[i] *= scale;
            }
            if(sample_count) *sample_count = c->samples;
            return FCGPU_OK;
        });
    }

    int fcgpu_save_film(fcgpu_ctx* c, double* raw_sum, uint64_t* sample_count)
    {
        return guarded(c, [&]() -> int {
            if(!c) return FCGPU_E_INVALID;
            if(!c->has_scene) return fail(c, FCGPU_E_NOSCENE, "no scene uploaded");
            if(!raw_sum || !sample_count) return fail(c, FCGPU_E_INVALID, "null argument");
            HIP_TRY(c, hipSetDevice(c->device));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            HIP_TRY(c, hipMemcpy(raw_sum, c->film, sizeof(double) * c->film_elems, hipMemcpyDeviceToHost));
            *sample_count = c->samples;
            return FCGPU_OK;
        });
    }

    int fcgpu_restore_film(fcgpu_ctx* c, const double* raw_sum, uint64_t sample_count)
    {
        return guarded(c, [&]() -> int {
            if(!c) return FCGPU_E_INVALID;
            if(!c->has_scene) return fail(c, FCGPU_E_NOSCENE, "no scene uploaded");
            if(!raw_sum) return fail(c, FCGPU_E_INVALID, "null argument");
            HIP_TRY(c, hipSetDevice(c->device));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            HIP_TRY(c, hipMemcpy(c->film, raw_sum, sizeof(double) * c->film_elems, hipMemcpyHostToDevice));
            // what the rows hold is no longer known to be one band pattern: a later fcgpu_reduce_film sends the whole frame
            c->cover = fcgpu_ctx::Coverage{};
            c->cover.any = true;
            c->samples = sample_count;
            return FCGPU_OK;
        });
    }

    int fcgpu_film_device_ptr(fcgpu_ctx* c, void** rgb_dev, uint64_t* bytes)
    {
        if(!c) return FCGPU_E_INVALID;
        if(!c->has_scene) return fail(c, FCGPU_E_NOSCENE, "no scene uploaded");
        if(rgb_dev) *rgb_dev = c->film;
        if(bytes) *bytes = sizeof(double) * c->film_elems;
        return FCGPU_OK;
    }

    int fcgpu_add_sample_count(fcgpu_ctx* c, int64_t delta)
    {
        if(!c) return FCGPU_E_INVALID;
        c->samples = static_cast<uint64_t>(static_cast<int64_t>(c->samples) + delta);
        return FCGPU_OK;
    }

    int fcgpu_get_counters(fcgpu_ctx* c, fcgpu_counters* out)
    {
        return guarded(c, [&]() -> int {
            if(!c || !out) return fail(c, FCGPU_E_INVALID, "null argument");
            HIP_TRY(c, hipSetDevice(c->device));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            unsigned long long st[ST_COUNT];
            HIP_TRY(c, hipMemcpy(st, c->stats, sizeof(st), hipMemcpyDeviceToHost));
            std::memset(out, 0, sizeof(*out));
            out->samples = c->samples;
            out->closest_rays = st[ST_CLOSEST];
            out->shadow_rays = st[ST_SHADOW];
            out->shadow_occluded = st[ST_OCCLUDED];
            out->nan_samples = st[ST_NAN];
            out->stack_overflows = st[ST_STACK_OVERFLOW];
            out->bvh_nodes = c->flat.nodes8.empty() ? c->flat.nodes.size() : c->flat.nodes8.size();
            out->primitives = c->flat.prim_info.size();
            out->kernel_launches = c->launches;
#ifdef FCG_SHADE_CLOCKS
            {
                static const char* names[8] = {"wait for the prefetched records", "surface point + entity", "arrival terms", "material + lobe record",
                                               "light strategy", "bsdf strategy", "queue append", "next loads + stores issued"};
                unsigned long long total = 0;
                for(int k = 0; k < 8; ++k) total += st[ST_SHADE_PH0 + k];
                if(total)
                    for(int k = 0; k < 8; ++k)
                        std::fprintf(stderr, "k_shade<Lambert> phase %d (%s): %.1f %% of the wave cycles, %.0f cycles per trip\n", k, names[k],
                                     100.0 * st[ST_SHADE_PH0 + k] / total, double(st[ST_SHADE_PH0 + k]) / double(st[ST_SHADE_PH0 + 8] ? st[ST_SHADE_PH0 + 8] : 1));
            }
#endif
            out->reserved[0] = st[ST_MEDIUM_OVERFLOW];
            out->reserved[1] = c->flat.max_stack;
            out->reserved[2] = c->flat.binary_nodes;
            out->reserved[3] = st[ST_NODES];
            out->reserved[4] = st[ST_PRIMS];
            out->reserved[5] = st[ST_STACK_SPILL];
            out->reserved[6] = c->flatten_us;
            out->medium_overflows = st[ST_MEDIUM_OVERFLOW];
            out->medium_deep_pushes = st[ST_MEDIUM_DEEP];
            out->reduce_bytes_copied = c->reduce_bytes;
            return FCGPU_OK;
        });
    }

    int fcgpu_reset_counters(fcgpu_ctx* c)
    {
        if(!c) return FCGPU_E_INVALID;
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, hipMemsetAsync(c->stats, 0, sizeof(unsigned long long) * ST_COUNT, c->stream));
        c->launches = 0;
        return FCGPU_OK;
    }

    int fcgpu_sync(fcgpu_ctx* c)
    {
        return guarded(c, [&]() -> int {
            if(!c) return FCGPU_E_INVALID;
            HIP_TRY(c, hipSetDevice(c->device));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            return FCGPU_OK;
        });
    }

    int fcgpu_set_profiling(fcgpu_ctx* c, int enabled)
    {
        return guarded(c, [&]() -> int {
            if(!c) return FCGPU_E_INVALID;
            HIP_TRY(c, hipSetDevice(c->device));
            drain_spans(c);
            c->profiling = enabled != 0;
            return FCGPU_OK;
        });
    }

    int fcgpu_get_stage_times(fcgpu_ctx* c, fcgpu_stage_times* out)
    {
        return guarded(c, [&]() -> int {
            if(!c || !out) return fail(c, FCGPU_E_INVALID, "null argument");
            HIP_TRY(c, hipSetDevice(c->device));
            drain_spans(c);
            *out = c->stage_times;
            c->stage_times = fcgpu_stage_times{};
            return FCGPU_OK;
        });
    }

    int fcgpu_stream(fcgpu_ctx* c, void** stream)
    {
        if(!c || !stream) return fail(c, FCGPU_E_INVALID, "null argument");
        *stream = c->stream;
        return FCGPU_OK;
    }

    int fcgpu_export_image(fcgpu_ctx* c, const char* name)
    {
        return guarded(c, [&]() -> int {
            if(!c || !name) return fail(c, FCGPU_E_INVALID, "null argument");
            if(!c->has_scene) return fail(c, FCGPU_E_NOSCENE, "no scene uploaded");
            std::vector<double> sum(c->film_elems);
            uint64_t count = 0;
            int rc = fcgpu_read_film(c, sum.data(), &count);
            if(rc) return rc;
            if(count == 0) return fail(c, FCGPU_E_INVALID, "nothing rendered yet: the image would be 0 / 0");
            std::string path = std::string(name) + ".raw";
            FILE* f = std::fopen(path.c_str(), "wb");
            if(!f) return fail(c, FCGPU_E_INVALID, "cannot open " + path);
            std::vector<float> row(3 * static_cast<size_t>(c->flat.width));
            for(int y = 0; y < c->flat.height; ++y)
            {
                for(size_t i = 0; i < row.size(); ++i) row[i] = static_cast<float>(sum[static_cast<size_t>(y) * row.size() + i] / static_cast<double>(count));
                std::fwrite(row.data(), sizeof(float), row.size(), f);
            }
            std::fclose(f);
            return FCGPU_OK;
        });
    }
}
