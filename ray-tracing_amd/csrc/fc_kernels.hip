// fc_kernels.hip -- the wavefront kernels of the Forward-MIS / Forward-BSDF integrator for gfx950.
//
// One sample of forward_mis_integrator::run_once (forward_mis_integrator.hpp:13-210) is cut into
// stages that each run over a whole queue of paths:
//
//   generate   :17-21   camera sample -> first extension ray, path state init
//   extend     :27,113  helper::raycast = closest hit + priority-medium stack (integrator.hpp:33-109);
//                       picks the lobe (:48) and files the hit in that lobe's queue (material sort)
//   shade<L>   :130-147 (arrival: transmittance, emitter + MIS)  then :47-112 (light strategy,
//                       BSDF sample) for lobe type L; emits one shadow ray and one extension ray
//   shadow     :84      scene.visibility = any-hit; adds the light-strategy term when unoccluded
//   accumulate :209     film add with the NaN/Inf guard (perspective_camera.hpp:113-134)
//
// Queues are compacted with wave64 ballots + popcount prefix sums, aggregated per 256-thread block
// through LDS so that a block issues one global atomic per queue.
// No MFMA: nothing here is a dense contraction.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>
#include <type_traits>

#include "../../include/fcgpu_sampler.h"
#include "fc_bsdf.h"
#include "fc_env.h"
#include "fc_kernels.h"
#include "fc_material.h"
#include "fc_traverse.h"


namespace fcg
{
    constexpr int BLOCK = 256;
    constexpr int WAVES = BLOCK / 64;

    // ------------------------------------------------------------------ helpers
    template<typename R> FC_D void draw(const Batch& B, const fcgpu_sample_stream& st, uint32_t dim, R& x, R& y)
    {
        uint64_t bits = fcgpu_stream_bits(&B.sampler, st, dim);
        x = R(fcgpu_sample_x24(bits)) * R(FCGPU_SAMPLER_SCALE);
        y = R(fcgpu_sample_y24(bits)) * R(FCGPU_SAMPLER_SCALE);
    }

    FC_D void slot_to_pixel(const Batch& B, uint32_t slot, uint32_t& px, uint32_t& py, uint32_t& sample)
    {
        uint32_t sl = fast_div(slot, B.div_npix);
        uint32_t pit = B.pix0 + (slot - sl * B.npix);
        uint32_t row = fast_div(pit, B.div_tw);
        px = static_cast<uint32_t>(B.x0) + (pit - row * static_cast<uint32_t>(B.tw));
        py = static_cast<uint32_t>(B.y0) + row + fast_div(row, B.div_band) * B.band_skip;  // band_skip = 0 unless rows are dealt in bands
        sample = B.s0 + sl;
    }

    FC_D fcgpu_sample_stream slot_stream(const Batch& B, uint32_t slot)
    {
        uint32_t px, py, s;
        slot_to_pixel(B, slot, px, py, s);
        return fcgpu_stream_begin(&B.sampler, px, py, s);
    }

    // Block-aggregated multi-queue append. Every thread of the block calls it (uniform control
    // flow); `key` < NK selects the queue, key < 0 appends nothing. Returns the slot.
    template<int NK>
    FC_D uint32_t block_append(int key, uint32_t* const* counters, uint32_t (*lds)[WAVES])
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const uint64_t lt = (1ull << lane) - 1ull;
        uint32_t prefix = 0;
#pragma unroll
        for(int k = 0; k < NK; ++k)
        {
            uint64_t m = __ballot(key == k);
            if(key == k) prefix = __popcll(m & lt);
            if(lane == 0) lds[k][wave] = __popcll(m);
        }
        __syncthreads();
        if(threadIdx.x < NK)
        {
            uint32_t total = 0;
#pragma unroll
            for(int w = 0; w < WAVES; ++w) total += lds[threadIdx.x][w];
            uint32_t base = total ? atomicAdd(counters[threadIdx.x], total) : 0u;
#pragma unroll
            for(int w = 0; w < WAVES; ++w)
            {
                uint32_t c = lds[threadIdx.x][w];
                lds[threadIdx.x][w] = base;
                base += c;
            }
        }
        __syncthreads();
        uint32_t slot = key >= 0 ? lds[key][wave] + prefix : INVALID_U32;
        __syncthreads();
        return slot;
    }

    // Two independent appends (flag a -> queue 0, flag b -> queue 1) in one pass: two barriers and
    // two atomics per block instead of six barriers.
    FC_D void block_append2(bool a, bool b, uint32_t* const* counters, uint32_t (*lds)[WAVES], uint32_t& slot_a, uint32_t& slot_b)
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const uint64_t lt = (1ull << lane) - 1ull;
        const uint64_t ma = __ballot(a), mb = __ballot(b);
        if(lane == 0)
        {
            lds[0][wave] = __popcll(ma);
            lds[1][wave] = __popcll(mb);
        }
        __syncthreads();
        if(threadIdx.x < 2)
        {
            uint32_t total = 0;
#pragma unroll
            for(int w = 0; w < WAVES; ++w) total += lds[threadIdx.x][w];
            uint32_t base = total ? atomicAdd(counters[threadIdx.x], total) : 0u;
#pragma unroll
            for(int w = 0; w < WAVES; ++w)
            {
                uint32_t c = lds[threadIdx.x][w];
                lds[threadIdx.x][w] = base;
                base += c;
            }
        }
        __syncthreads();
        slot_a = lds[0][wave] + __popcll(ma & lt);
        slot_b = lds[1][wave] + __popcll(mb & lt);
        __syncthreads();  // lds is reused by the next trip
    }

    FC_D void wave_stat_add(unsigned long long* stats, int index, uint32_t v)
    {
#pragma unroll
        for(int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        if((threadIdx.x & 63) == 0 && v) atomicAdd(&stats[index], static_cast<unsigned long long>(v));
    }

    // v is already a per-wave total held by every lane
    FC_D void wave_stat_add_uniform(unsigned long long* stats, int index, uint32_t v)
    {
        if((threadIdx.x & 63) == 0 && v) atomicAdd(&stats[index], static_cast<unsigned long long>(v));
    }

    template<typename R> FC_D uint32_t medium_top(const SceneDev<R>& S, const MStack& ms)  // integrator.hpp:51-56
    {
        uint32_t top = 0;
        int top_prio = S.media[0].priority;
        uint32_t n = ms_count(ms.w);
        for(uint32_t i = 0; i < n; ++i)
        {
            uint32_t id = ms_get(ms, i);
            int p = S.media[id].priority;
            if(p > top_prio)
            {
                top_prio = p;
                top = id;
            }
        }
        return top;
    }

    // integrator.hpp:41-42. The reference holds the vacuum + 9 media and does not check (a tenth push writes past its
    // array); here a push beyond MEDIUM_STACK_CAP is dropped and counted: the counter is an error the hosts refuse
    // (fcgpu_counters.medium_overflows). `deep` counts the pushes that land in the extension word (development statistic).
    FC_D void medium_push(MStack& ms, uint32_t id, uint32_t& overflow, uint32_t& deep)
    {
        const uint32_t n = ms_count(ms.w);
        if(n >= static_cast<uint32_t>(MEDIUM_STACK_CAP))
        {
            overflow += 1;
            return;
        }
        if(n >= static_cast<uint32_t>(MEDIUM_STACK_INLINE)) deep += 1;
        ms_set(ms, n, id);
        ms.w = ms_with_count(ms.w, n + 1);
    }

    // Ray-origin offset of entity_scene::raycast / visibility (scene.hpp:85-93, 118-131, 146-152). The reference lifts the
    // origin epsilon = 1e-6 off the surface along the geometric normal, to the side the ray leaves on (`positive` =
    // its test on dot(n, w)). fp32 cannot resolve 1e-6 at scene coordinates, and the 1e-4 it needs, applied along the
    // normal, moves the whole ray sideways by up to 1e-4: measured at the headline configuration that alone, in fp64
    // arithmetic, takes 12 % of the pixels more than 1 % away from the reference at 16 spp.
    // S.offset_along_ray (the fp32 default) therefore advances the origin ALONG THE RAY until it is epsilon above the
    // surface: the ray stays on the reference's line, only its first epsilon / |cos| are skipped; at grazing angles the
    // advance is capped at 100 epsilon and the rest of the height comes from the normal as before. w: unit direction of travel.
    template<typename R> FC_D vec3<R> lift_origin(const SceneDev<R>& S, vec3<R> p, vec3<R> n, vec3<R> w, bool positive)
    {
        if(!S.offset_along_ray) return positive ? p + n * S.epsilon : p - n * S.epsilon;
        const R c = r_abs(dot(n, w));
        const R a = r_min(S.epsilon / c, R(100) * S.epsilon);
        const R rest = S.epsilon - a * c;
        const vec3<R> q = p + w * a;
        return positive ? q + n * rest : q - n * rest;
    }

    template<typename R> FC_D PathState<R> state_unpack(const Vec4<R>& a, const Vec4<R>& b, const Vec4<R>& c)
    {
        PathState<R> st;
        st.beta[0] = a.x; st.beta[1] = a.y; st.beta[2] = a.z; st.pdf = a.w;
        st.p1[0] = b.x; st.p1[1] = b.y; st.p1[2] = b.z; st.flags = unpack_word(b.w);
        st.w[0] = c.x; st.w[1] = c.y; st.w[2] = c.z; st.ms_ext = c.w;
        return st;
    }
    // path state columns (Wavefront::st_c*), streamed like the other queue records
    template<typename R> FC_D PathState<R> state_load(const Wavefront<R>& Q, int set, uint32_t i)
    {
        return state_unpack(ld_stream(&Q.st_c0[set][i]), ld_stream(&Q.st_c1[set][i]), ld_stream(&Q.st_c2[set][i]));
    }
    template<typename R> FC_D void state_store(const Wavefront<R>& Q, int set, uint32_t i, const PathState<R>& st)
    {
        st_stream(&Q.st_c0[set][i], Vec4<R>{st.beta[0], st.beta[1], st.beta[2], st.pdf});
        st_stream(&Q.st_c1[set][i], Vec4<R>{st.p1[0], st.p1[1], st.p1[2], pack_word(R(0), st.flags)});
        st_stream(&Q.st_c2[set][i], Vec4<R>{st.w[0], st.w[1], st.w[2], st.ms_ext});
    }

    // ------------------------------------------------------------------ generate
    template<typename R>
    __global__ __launch_bounds__(BLOCK) void k_generate(SceneDev<R> S, Wavefront<R> Q, Batch B)
    {
        const uint32_t n = B.npix * B.nsamp;
        for(uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK)
        {
            uint32_t px, py, s;
            slot_to_pixel(B, i, px, py, s);
            const fcgpu_sample_stream key = fcgpu_stream_begin(&B.sampler, px, py, s);
            R dx, dy, lx, ly;
            draw(B, key, 0u, dx, dy);  // sample_direction
            draw(B, key, 1u, lx, ly);  // sample_point (lens)
            const CameraRec<R>& C = S.camera;
            // perspective_camera::sample_p_and_wi, perspective_camera.hpp:29-75
            vec3<R> lens{R(0), R(0), R(0)};
            if(C.lens_radius != R(0))
            {
                R a, b;
                sample_disk_concentric(lx, ly, a, b);
                lens.x = a * C.lens_radius;
                lens.y = b * C.lens_radius;
            }
            vec3<R> plane{((R(px) + dx) / R(C.W) - R(0.5)) * C.plane_w, (R(0.5) - (R(py) + dy) / R(C.H)) * C.plane_h, C.focus};
            vec3<R> dir = plane - lens;
            vec3<R> wi_l = dir / r_sqrt(dot(dir, dir));
            vec3<R> wi = xf_dir(C.fwd, wi_l);
            vec3<R> origin = xf_point(C.fwd, lens);
            vec3<R> n_cam{C.fwd[2], C.fwd[6], C.fwd[10]};
            // entity_scene::raycast origin offset, scene.hpp:85-93
            vec3<R> o = lift_origin(S, origin, n_cam, wi, dot(n_cam, wi) > R(0));

            PathState<R> st{};
            st.beta[0] = st.beta[1] = st.beta[2] = R(1);  // beta0 = W*H is folded into the film scale
            st.flags = 1u;
            st.p1[0] = origin.x; st.p1[1] = origin.y; st.p1[2] = origin.z;
            st.w[0] = wi.x; st.w[1] = wi.y; st.w[2] = wi.z;
            state_store(Q, B.parity, i, st);  // ray i of the first queue
            Q.li_shadow[i] = Vec4<R>{R(0), R(0), R(0), R(0)};

            Vec4<R> c0{o.x, o.y, o.z, wi.x};
            Vec4<R> c1{wi.y, wi.z, R(0), R(0)};
            pack_ids(c1, i, 0u);
            Q.ray_c0[i] = c0;
            Q.ray_c1[i] = c1;
        }
        if(blockIdx.x == 0 && threadIdx.x == 0) Q.counters->ray[B.parity][0] = n;
    }

    // Experiment FCG_TREELET: every block copies the top of the hierarchy into its LDS (fc_traverse.h fetches from there)
    template<typename R> FC_D void treelet_fill(const SceneDev<R>& S, unsigned char* lds)
    {
#if FCG_TREELET
        const uint4* src = reinterpret_cast<const uint4*>(S.nodes);
        uint4* dst = reinterpret_cast<uint4*>(lds);
        const uint32_t have = S.node_count < static_cast<uint32_t>(FCG_TREELET) ? S.node_count : static_cast<uint32_t>(FCG_TREELET);
        for(uint32_t i = threadIdx.x; i < 4u * have; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
#endif
    }

    // WIDE8 instantiations traverse the 8-wide hierarchy (Node8, fcgpu_options.bvh_width = 8) instead of the 4-wide one
    template<typename R, bool WIDE8> struct TravSel { typedef Trav<R> type; };
    template<typename R> struct TravSel<R, true> { typedef Trav8<R> type; };

    // ------------------------------------------------------------------ extend
    // Persistent waves with dynamic ray fetch: a lane whose ray is finished does not idle until the
    // slowest ray of its wave ends; once FINISH_MIN lanes of the wave have finished (or nothing is
    // left to traverse) they resolve their hits together and take new rays (work_take).

    // ------------------------------------------------------------------ textured materials
    // surface_point::get_uv of a hit: mesh_surface.hpp:215-216 (+ :337-352 default uvs), plane_surface.hpp:91-95;
    // sphere_surface never sets it (surface_point.hpp:50: {0, 0}). k_extend (lobe pick) and k_shade (lobe
    // parameters) call this with the same stored p, b0, b1, so that both see the same texels.
    template<typename R>
    FC_D void surface_uv(const SceneDev<R>& S, uint32_t kind, uint32_t idx, vec3<R> p, R b0, R b1, R& u, R& v)
    {
        u = R(0);
        v = R(0);
        if(kind == KIND_TRI)
        {
            const uint32_t attr = S.tris[idx].attr;
            const R b2 = R(1) - b0 - b1;
            if(attr == INVALID_U32)
            {
                u = b0 * R(0) + b1 * R(1) + b2 * R(1);
                v = b0 * R(0) + b1 * R(0) + b2 * R(1);
            }
            else
            {
                const TriAttr& A = S.tri_attrs[attr];
                u = b0 * R(A.uv0[0]) + b1 * R(A.uv1[0]) + b2 * R(A.uv2[0]);
                v = b0 * R(A.uv0[1]) + b1 * R(A.uv1[1]) + b2 * R(A.uv2[1]);
            }
        }
        else if(kind == KIND_PLANE)
        {
            const XPrim<R>& P = S.xprims[idx];
            const vec3<R> q = xf_point(P.inv, p);
            u = (q.x + P.a / R(2)) / P.a;
            v = R(1) - (q.z + P.b / R(2)) / P.b;
        }
    }

#ifdef FCG_TRI_REFINE
    // Experiment (tools/build_variant.sh -DFCG_TRI_REFINE): (t, b0, b1) of the ACCEPTED triangle hit recomputed in double
    // from the same float ray and float vertices, once per ray (see DESIGN.md section 4 for what it measured)
    FC_D void refine_hit_f64(const SceneDev<float>& S, Trav<float>& T)
    {
        if(T.pk == INVALID_U32 || (T.pk >> 30) != KIND_TRI) return;
        const TriRec& tr = S.tris[T.pk & 0x3FFFFFFFu];
        const RayCtx<double> rc = make_ray(vec3<double>{double(T.rc.o.x), double(T.rc.o.y), double(T.rc.o.z)},
                                           vec3<double>{double(T.rc.d.x), double(T.rc.d.y), double(T.rc.d.z)});
        double t, b0, b1;
        if(tri_intersect<double>(tr, rc, consts<double>::inf, t, b0, b1))
        {
            T.t_max = static_cast<float>(t);
            T.b0 = static_cast<float>(b0);
            T.b1 = static_cast<float>(b1);
        }
    }
    template<typename TT> FC_D void refine_hit_f64(const SceneDev<double>&, TT&) {}
    FC_D void refine_hit_f64(const SceneDev<float>&, Trav8<float>&) {}
#endif

    // the material's parameters at (u, v): image textures where it has them (x of the texel for the scalar
    // ones, image_texture_2d_r), its constants elsewhere
    template<typename R>
    FC_D MatValues<R> material_values(const SceneDev<R>& S, uint32_t material, const MaterialRec& m, R u, R v)
    {
        MatValues<R> val = S.matparams[material];
        if(m.tex[TEX_COLOR_A])
        {
            const vec3<R> c = tex_evaluate(S.textures[m.tex[TEX_COLOR_A] - 1], u, v);
            val.color_a[0] = c.x, val.color_a[1] = c.y, val.color_a[2] = c.z;
        }
        if(m.tex[TEX_COLOR_B])
        {
            const vec3<R> c = tex_evaluate(S.textures[m.tex[TEX_COLOR_B] - 1], u, v);
            val.color_b[0] = c.x, val.color_b[1] = c.y, val.color_b[2] = c.z;
        }
        if(m.tex[TEX_ROUGHNESS]) val.roughness[0] = val.roughness[1] = tex_evaluate(S.textures[m.tex[TEX_ROUGHNESS] - 1], u, v).x;
        if(m.tex[TEX_METALNESS]) val.metalness = tex_evaluate(S.textures[m.tex[TEX_METALNESS] - 1], u, v).x;
        if(m.tex[TEX_IOR]) val.ior = tex_evaluate(S.textures[m.tex[TEX_IOR] - 1], u, v).x;
        return val;
    }

    // TEX: the scene has materials with image textures (per-hit lobe construction compiled in: +5 % shade time, so
    // scenes without them run the instantiations without it)
    // FCG_EXTEND_WAVES (experiment, tools/build_variant.sh): makes the compiler fit k_extend into the registers of that many waves per SIMD
#ifndef FCG_EXTEND_SPEC
#define FCG_EXTEND_SPEC 0
#endif
#ifndef FCG_SHADOW_SPEC
#define FCG_SHADOW_SPEC 0
#endif
#ifndef FCG_SHADOW_DENSE
#define FCG_SHADOW_DENSE 1  // 0: the fp32 shadow rays run k_shadow<float, ...> with STACK_CAP entries like every other traversal kernel (A/B)
#endif
#ifndef FCG_SHADOW_DENSE_WAVES
#define FCG_SHADOW_DENSE_WAVES 7  // waves per SIMD k_shadow_dense is compiled for (with FCG_SHADOW_STACK_CAP 22: seven blocks per CU)
#endif
#ifdef FCG_SHADOW_WAVES
#define FCG_SHADOW_ATTR __attribute__((amdgpu_waves_per_eu(FCG_SHADOW_WAVES, FCG_SHADOW_WAVES)))
#else
#define FCG_SHADOW_ATTR
#endif
    // The spill-capable k_extend of untextured scenes (every hierarchy deeper than 28 entries) is held to the 96 registers of
    // five waves per SIMD like the plain one (the compiler takes 106 = four waves otherwise; with the limit it spills 8 B per
    // lane to scratch): k_extend -7.4 % on the room at tess 128. 0 = no limit (the textured instantiations, 120+ registers,
    // lose more to scratch than a fifth wave gives them and are left alone).
#ifndef FCG_EXTEND_SPILL_WAVES
#define FCG_EXTEND_SPILL_WAVES 5
#endif
#ifdef FCG_EXTEND_WAVES
#define FCG_EXTEND_ATTR __attribute__((amdgpu_waves_per_eu(FCG_EXTEND_WAVES, FCG_EXTEND_WAVES)))
#else
#define FCG_EXTEND_ATTR __attribute__((amdgpu_waves_per_eu(FCG_EXTEND_SPILL_WAVES && SPILL && !TEX && !PROF && !WIDE8 && sizeof(R) == 4 ? FCG_EXTEND_SPILL_WAVES : 1)))
#endif
    template<typename R, bool PROF, bool SPILL, bool TEX, bool WIDE8 = false>
    __global__ __launch_bounds__(BLOCK) FCG_EXTEND_ATTR void k_extend(SceneDev<R> S, Wavefront<R> Q, Batch B)
    {
        // PROF: descent, leaf cycles, trips, leaf phases, finish cycles, finish phases, leaf lanes, finish lanes, cycles of the finish phase's
        // three parts (hit record, work_take, ray load + set-up), per-lane node steps that found nothing, of those emptied by t_max,
        // lanes waiting with a leaf / with a finished ray summed over the descent trips
        unsigned long long ph[15] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        __shared__ uint32_t stack_lds[STACK_CAP * BLOCK];
        __shared__ __attribute__((aligned(16))) unsigned char nodebuf_lds[FCG_TREELET ? FCG_TREELET * 64 : WAVES * NODEBUF_WAVE];
        uint32_t* stack = stack_lds + threadIdx.x;
        unsigned char* nodebuf = FCG_TREELET ? nodebuf_lds : nodebuf_lds + (threadIdx.x >> 6) * NODEBUF_WAVE;
        treelet_fill(S, nodebuf_lds);
        const uint32_t n = Q.counters->ray[B.parity][0];
        uint32_t n_cast = 0, n_moverflow = 0, n_mdeep = 0, n_nodes = 0, n_prims = 0;
        StackStats n_overflow;
        const bool last = B.h >= B.max_len;
        // dimensions per bounce: 6 for forward_mis (fcgpu_sampler.h), 3 for forward_bsdf
        // (forward_bsdf_integrator.hpp:50,56: lobe pick, then u2, u1)
        const uint32_t dim_lobe = 2u + (B.integrator == FCGPU_INTEGRATOR_MIS ? 6u : 3u) * static_cast<uint32_t>(B.h - 1);

        WorkCursor W = work_begin(n);
        typename TravSel<R, WIDE8>::type T;
        trav_idle(T);
        uint32_t slot = INVALID_U32;  // ray-queue index this lane is working on
        uint32_t path = 0, ms = 0;
        vec3<R> dir{};
        int guard = 0;
        unsigned long long t_start = 0;
        uint32_t n_rounds = 0;
        if(PROF) t_start = wall_clock64();

        for(;;)
        {
            const bool finished = slot != INVALID_U32 && trav_done(T);
            const bool traversing = slot != INVALID_U32 && !trav_done(T);
            const uint64_t any_trav = __ballot(traversing);
            if(__popcll(__ballot(finished)) >= B.finish_min || any_trav == 0ull)
            {
                unsigned long long cf = 0;
                if(PROF)
                {
                    cf = clock64();
                    ph[7] += __popcll(__ballot(finished));
                }
                if(finished)
                {
                    // helper::raycast after scene.raycast returned, integrator.hpp:47-108 (recursion -> re-cast)
                    int key = 0xFF;
                    uint32_t dec_lobe = 0;
                    Vec4<R> h0{}, h1{};
                    bool recast = false;
                    bool alive = false;
                    uint32_t prim = INVALID_U32, entity = 0, key_hint = KEY_HINT_NONE;
                    vec3<R> p{}, nrm{};
                    // the path's medium stack: its word came with the ray; the extension word (entries 5..8) stays in the
                    // path-state column at the ray's index and is fetched only by paths that deep, written back only when changed
                    MStack M{ms, 0u};
                    if(ms_deep(ms)) M.ext = unpack_word(Q.st_c2[B.parity][slot].w);
                    const uint32_t ext_in = M.ext;
                    if(T.pk != INVALID_U32)
                    {
#ifdef FCG_TRI_REFINE
                        refine_hit_f64(S, T);
#endif
                        hit_point(S, T.rc, T.pk, T.t_max, T.b0, T.b1, p, nrm, prim, entity, key_hint);
                        // a hinted triangle is no medium boundary and has one constant lobe: the record just loaded decides
                        const uint32_t med = key_hint != KEY_HINT_NONE ? 0u : static_cast<uint32_t>(S.entities[entity].medium);
                        const uint32_t top = medium_top(S, M);
                        if(med == 0u)
                        {
                            M.w = ms_with_media(M.w, top, top);
                            alive = true;
                        }
                        else if(dot(dir, nrm) <= R(0))  // entering
                        {
                            if(S.media[med].priority <= S.media[top].priority)
                            {
                                medium_push(M, med, n_moverflow, n_mdeep);  // the re-cast leaves p INTO its medium, :36-44
                                p = lift_origin(S, p, nrm, dir, false);  // scene.hpp:86-93 with dot(n, w) <= 0
                                recast = true;
                            }
                            else
                            {
                                M.w = ms_with_media(M.w, top, med);
                                alive = true;
                            }
                        }
                        else
                        {
                            uint32_t cnt = ms_count(M.w), idx = 0;
                            for(; idx < cnt; ++idx)
                                if(ms_get(M, idx) == med) break;
                            if(idx < cnt)  // else: leaving a medium never entered, the path dies (:88)
                            {
                                ms_set(M, idx, ms_get(M, cnt - 1));  // swap with last, :90-91
                                M.w = ms_with_count(M.w, cnt - 1);
                                const uint32_t new_top = medium_top(S, M);
                                if(S.media[new_top].priority == S.media[top].priority)
                                {
                                    p = lift_origin(S, p, nrm, dir, true);
                                    recast = true;
                                }
                                else
                                {
                                    M.w = ms_with_media(M.w, new_top, top);
                                    alive = true;
                                }
                            }
                        }
                    }
                    if(M.ext != ext_in) Q.st_c2[B.parity][slot].w = pack_word(R(0), M.ext);
                    ms = M.w;
                    if(recast && ++guard < 32)
                    {
                        trav_begin(T, p, dir, consts<R>::inf);
                        n_cast += 1;
                    }
                    else
                    {
                        if(alive)
                        {
                            uint32_t lobe = 0;
                            if(last)
                            {
                                // the last vertex only gathers emitted light (forward_mis_integrator.hpp:128-147): a hit
                                // on anything else ends the path here, without a hit record
                                key = S.entities[entity].light >= 0 ? Q_FINAL : 0xFF;
                            }
                            else if(key_hint != KEY_HINT_NONE)
                            {
                                key = static_cast<int>(key_hint);  // single constant lobe: any draw picks lobe 0
                            }
                            else
                            {
                                // bsdf::sample_bxdf, bsdf.hpp:46-55, on draw `dim_lobe`.x
                                const MaterialRec m = S.materials[S.entities[entity].material];
                                // the draw decides nothing for a single-lobe material (any u picks lobe 0): its two hashes
                                // (stream key + dimension) are skipped there; the dimension stays reserved either way
                                double u = 0.0;
                                if(m.lobe_count > 1 || (TEX && m.textured))
                                {
                                    uint64_t bits = fcgpu_stream_bits(&B.sampler, slot_stream(B, path), dim_lobe);
                                    u = fcgpu_sample_x24(bits) * FCGPU_SAMPLER_SCALE;
                                }
                                if(TEX && m.textured)
                                {
                                    R tu, tv;
                                    surface_uv(S, T.pk >> 30, T.pk & 0x3FFFFFFFu, p, T.b0, T.b1, tu, tv);
                                    const MatValues<R> val = material_values(S, S.entities[entity].material, m, tu, tv);
                                    int types[MAX_LOBES];
                                    double scales[MAX_LOBES], weights[MAX_LOBES], factor[MAX_LOBES], cdf[MAX_LOBES];
                                    const int count = material_layout(m.kind, val, types, scales, weights);
                                    lobe_pick_table(count, scales, weights, factor, cdf);
                                    for(int l = count - 1; l >= 0; --l)
                                    {
                                        if(u > cdf[l])
                                        {
                                            lobe = static_cast<uint32_t>(l);
                                            break;
                                        }
                                    }
                                    key = types[lobe];
                                }
                                else
                                {
                                    for(int l = m.lobe_count - 1; l >= 0; --l)
                                    {
                                        if(u > S.lobes[m.first_lobe + l].cdf)
                                        {
                                            lobe = static_cast<uint32_t>(l);
                                            break;
                                        }
                                    }
                                    key = S.lobes[m.first_lobe + lobe].type;
                                }
                            }
                            dec_lobe = lobe;
                            h0 = Vec4<R>{p.x, p.y, p.z, T.b0};
                            h1.x = T.b1;
                            h1.y = pack_word(R(0), ms);
                            pack_ids(h1, hit_word(T.pk, lobe), path);
                        }
                        else if(S.env_kind != FCGPU_ENV_NONE)
                        {
                            key = Q_MISS;
                            h1.y = pack_word(R(0), ms);
                            pack_ids(h1, INVALID_U32, path);
                        }
                        if(key != 0xFF)
                        {
                            st_stream(&Q.hit_c0[slot], h0);
                            st_stream(&Q.hit_c1[slot], h1);
                        }
                        Q.hit_key[slot] = static_cast<uint8_t>(key);
                        if(Q.dec_hit)  // parity hook: what this path hit at vertex h
                            Q.dec_hit[static_cast<size_t>(B.h - 1) * Q.dec_stride + path] = alive ? hit_word(T.pk, dec_lobe) : (T.pk == INVALID_U32 ? DEC_MISS : DEC_DEAD);  // MISS: the ray left the scene (with or without an environment); DEAD: the path ended at a boundary (integrator.hpp:88)
                        slot = INVALID_U32;
                    }
                }
                // idle lanes take new rays
                unsigned long long c1 = 0, c2 = 0;
                if(PROF) c1 = clock64();
                uint32_t fresh = work_take(W, slot == INVALID_U32, Q.counters->work_extend, n);
                if(PROF) c2 = clock64();
                if(fresh != INVALID_U32)
                {
                    slot = fresh;
                    Vec4<R> c0 = ld_stream(&Q.ray_c0[slot]), c1 = ld_stream(&Q.ray_c1[slot]);
                    unpack_ids(c1, path, ms);
                    dir = vec3<R>{c0.w, c1.x, c1.y};
                    trav_begin(T, vec3<R>{c0.x, c0.y, c0.z}, dir, consts<R>::inf);
                    n_cast += 1;
                    guard = 0;
                }
                if(__ballot(slot != INVALID_U32) == 0ull) break;
                if(PROF)
                {
                    const unsigned long long c3 = clock64();
                    ph[4] += c3 - cf;
                    ph[5] += 1;
                    ph[8] += c1 - cf;
                    ph[9] += c2 - c1;
                    ph[10] += c3 - c2;
                }
            }
            if constexpr(WIDE8)
                trav_round8<R, false, BLOCK>(S, T, stack, n_overflow, B.descend_min, n_nodes, n_prims);
            else
                trav_round<R, false, BLOCK, SPILL, PROF, FCG_EXTEND_SPEC != 0>(S, T, stack, nodebuf, n_overflow, B.descend_min, n_nodes, n_prims, ph);
            n_rounds += 1;
        }
        if(PROF)
        {
            uint32_t rays = n_cast, nodes = n_nodes, prims = n_prims, nodes_max = n_nodes, prims_max = n_prims;
#pragma unroll
            for(int o = 32; o > 0; o >>= 1)
            {
                rays += __shfl_down(rays, o);
                nodes += __shfl_down(nodes, o);
                prims += __shfl_down(prims, o);
                nodes_max = max(nodes_max, static_cast<uint32_t>(__shfl_down(nodes_max, o)));
                ph[11] += __shfl_down(ph[11], o);
                ph[12] += __shfl_down(ph[12], o);
                prims_max = max(prims_max, static_cast<uint32_t>(__shfl_down(prims_max, o)));
            }
            if((threadIdx.x & 63) == 0)
            {
                unsigned long long* t = Q.timeline + (static_cast<size_t>(blockIdx.x) * WAVES + (threadIdx.x >> 6)) * TIMELINE_WORDS;
                t[0] = t_start;
                t[1] = wall_clock64();
                t[2] = rays;
                t[3] = n_rounds;
                t[4] = nodes;
                t[5] = nodes_max;
                t[6] = prims;
                t[7] = prims_max;
                for(int k = 0; k < 15; ++k) t[8 + k] = ph[k];
            }
        }
        wave_stat_add(Q.stats, ST_CLOSEST, n_cast);
        // per-lane counts are summed over the wave; the counts trav_round keeps per wave (fc_traverse.h) are added once
        if(SPILL) wave_stat_add(Q.stats, ST_STACK_OVERFLOW, n_overflow.lost); else wave_stat_add_uniform(Q.stats, ST_STACK_OVERFLOW, n_overflow.lost);
        wave_stat_add(Q.stats, ST_STACK_SPILL, n_overflow.spilled);
        wave_stat_add(Q.stats, ST_MEDIUM_OVERFLOW, n_moverflow);
        wave_stat_add(Q.stats, ST_MEDIUM_DEEP, n_mdeep);
        if(PROF) wave_stat_add(Q.stats, ST_NODES, n_nodes); else wave_stat_add_uniform(Q.stats, ST_NODES, n_nodes);
        wave_stat_add(Q.stats, ST_PRIMS, n_prims);
    }

    // material sort: index lists per queue key from hit_key[]. A tile is BLOCK * SORT_ITEMS hits (SORT_ITEMS per thread,
    // BLOCK apart, so that the byte loads coalesce); a block compacts up to FCG_SORT_SUB_MAX tiles with ONE global atomic per key.
#ifndef FCG_SORT_ITEMS
#define FCG_SORT_ITEMS 16
#endif
    constexpr int SORT_ITEMS = FCG_SORT_ITEMS;

    // A block takes `sub` consecutive tiles of BLOCK * SORT_ITEMS hits per iteration and asks for its list space ONCE per key
    // for all of them. The kernel's time followed its number of block iterations (8 / 16 / 32 hits per thread and iteration:
    // 7.6 / 4.3 / 3.0 ms per step): an iteration is a dependent sequence (key bytes from memory, ballots, barrier, one returning
    // atomic per key, barrier, index stores) that the few resident blocks of a CU do not overlap well. Two passes over the
    // block's tiles: count per wave and key, one atomic per key, then the same ballots again to place the indices (the key bytes
    // of the second pass come from the L2). `sub` follows the queue length, so that short queues still spread over the grid.
#ifndef FCG_SORT_SUB_MAX
#define FCG_SORT_SUB_MAX 16
#endif
    template<typename R>
    __global__ __launch_bounds__(BLOCK) void k_sort(Wavefront<R> Q, Batch B)
    {
        __shared__ uint32_t cnt_lds[Q_COUNT][WAVES];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const uint64_t lt = (1ull << lane) - 1ull;
        const uint32_t n = Q.counters->ray[B.parity][0];
        constexpr uint32_t TILE = BLOCK * SORT_ITEMS;
        uint32_t sub = n / (gridDim.x * TILE);
        sub = sub < 1u ? 1u : (sub > static_cast<uint32_t>(FCG_SORT_SUB_MAX) ? static_cast<uint32_t>(FCG_SORT_SUB_MAX) : sub);
        const uint64_t chunk = static_cast<uint64_t>(TILE) * sub;
        for(uint64_t base64 = blockIdx.x * chunk; base64 < n; base64 += gridDim.x * chunk)
        {
            const uint32_t base = static_cast<uint32_t>(base64);
            // pass 1: lane k (< Q_COUNT) accumulates the wave's count of key k over the block's tiles
            uint32_t wave_cnt = 0;
            for(uint32_t t = 0; t < sub; ++t)
            {
                const uint32_t tile = base + t * TILE;
                if(tile >= n) break;
                // item j of this thread is hit  tile + j * BLOCK + threadIdx.x  (coalesced byte loads)
                uint32_t key[SORT_ITEMS];
#pragma unroll
                for(int j = 0; j < SORT_ITEMS; ++j)
                {
                    const uint32_t i = tile + j * BLOCK + threadIdx.x;
                    key[j] = i < n ? Q.hit_key[i] : 0xFFu;
                }
#pragma unroll
                for(int k = 0; k < Q_COUNT; ++k)
                {
                    if(Q.hit_idx[k] == nullptr) continue;  // a key this scene cannot produce (no such lobe, no environment) has no list
                    uint32_t run = 0;
#pragma unroll
                    for(int j = 0; j < SORT_ITEMS; ++j) run += __popcll(__ballot(key[j] == static_cast<uint32_t>(k)));
                    if(lane == k) wave_cnt += run;
                }
            }
            if(lane < Q_COUNT) cnt_lds[lane][wave] = wave_cnt;
            __syncthreads();
            if(threadIdx.x < Q_COUNT)
            {
                uint32_t total = 0;
#pragma unroll
                for(int w = 0; w < WAVES; ++w) total += cnt_lds[threadIdx.x][w];
                uint32_t start = total ? atomicAdd(&Q.counters->hit[threadIdx.x][0], total) : 0u;
#pragma unroll
                for(int w = 0; w < WAVES; ++w)
                {
                    uint32_t c = cnt_lds[threadIdx.x][w];
                    cnt_lds[threadIdx.x][w] = start;
                    start += c;
                }
            }
            __syncthreads();
            // pass 2: lane k holds where the wave's next index of key k goes
            uint32_t next = lane < Q_COUNT ? cnt_lds[lane][wave] : 0u;
            for(uint32_t t = 0; t < sub; ++t)
            {
                const uint32_t tile = base + t * TILE;
                if(tile >= n) break;
                uint32_t key[SORT_ITEMS];
                uint32_t pos[SORT_ITEMS];  // list position of the item
#pragma unroll
                for(int j = 0; j < SORT_ITEMS; ++j)
                {
                    const uint32_t i = tile + j * BLOCK + threadIdx.x;
                    key[j] = i < n ? Q.hit_key[i] : 0xFFu;
                    pos[j] = 0;
                }
#pragma unroll
                for(int k = 0; k < Q_COUNT; ++k)
                {
                    if(Q.hit_idx[k] == nullptr) continue;
                    const uint32_t first = static_cast<uint32_t>(__shfl(static_cast<int>(next), k));
                    uint32_t run = 0;
#pragma unroll
                    for(int j = 0; j < SORT_ITEMS; ++j)
                    {
                        const uint64_t m = __ballot(key[j] == static_cast<uint32_t>(k));
                        if(key[j] == static_cast<uint32_t>(k)) pos[j] = first + run + __popcll(m & lt);
                        run += __popcll(m);
                    }
                    if(lane == k) next += run;
                }
#pragma unroll
                for(int j = 0; j < SORT_ITEMS; ++j)
                {
                    if(key[j] != 0xFFu) Q.hit_idx[key[j]][pos[j]] = tile + j * BLOCK + threadIdx.x;
                }
            }
            __syncthreads();  // cnt_lds is reused by the next iteration
        }
    }

    // ------------------------------------------------------------------ surface point reconstruction
    template<typename R>
    struct SurfacePoint
    {
        vec3<R> p;
        ShadingFrame<R> f;
        uint32_t entity;
        uint32_t kind_index;
    };

    template<typename R>
    FC_D SurfacePoint<R> make_surface_point(const SceneDev<R>& S, uint32_t kind, uint32_t idx, vec3<R> p, R b0, R b1)
    {
        SurfacePoint<R> sp;
        sp.p = p;
        sp.kind_index = (kind << 30) | idx;
        if(kind == KIND_TRI)
        {
            // mesh_surface.hpp:211-248; normal and tangent are constants of the triangle, computed at upload (TriShade)
            const TriShade<R> T = S.tri_shade[idx];
            sp.entity = T.entity;
            sp.f.ng = vec3<R>{T.ng[0], T.ng[1], T.ng[2]};
            sp.f.ns = sp.f.ng;
            if(T.attr != INVALID_U32 && (S.entities[sp.entity].flags & 1u))
            {
                const TriAttr& A = S.tri_attrs[T.attr];
                R b2 = R(1) - b0 - b1;
                sp.f.ns = normalize(b0 * ldv<R>(A.n0) + b1 * ldv<R>(A.n1) + b2 * ldv<R>(A.n2));
            }
            const vec3<R> tangent{T.t[0], T.t[1], T.t[2]};
            sp.f.b = cross(tangent, sp.f.ns);
            sp.f.t = cross(sp.f.ns, sp.f.b);
        }
        else
        {
            const XPrim<R>& P = S.xprims[idx];
            sp.entity = P.entity;
            if(kind == KIND_SPHERE)
            {
                // sphere_surface.hpp:123-140; tangent (-sin phi, 0, cos phi), phi = atan2(z, x), is
                // (-z, 0, x) / hypot(x, z) -- built from the OBJECT-space position and left unrotated
                vec3<R> pl = xf_point(P.inv, p);
                sp.f.ng = xf_dir(P.fwd, normalize(pl));
                R hyp = r_sqrt(pl.x * pl.x + pl.z * pl.z);
                vec3<R> tangent = hyp > R(0) ? vec3<R>{-pl.z / hyp, R(0), pl.x / hyp} : vec3<R>{R(0), R(0), R(1)};
                sp.f.ns = sp.f.ng;
                sp.f.t = tangent;
                sp.f.b = cross(tangent, sp.f.ng);
            }
            else
            {
                // plane_surface.hpp:88-99
                sp.f.ng = vec3<R>{P.fwd[1], P.fwd[5], P.fwd[9]};
                sp.f.ns = sp.f.ng;
                sp.f.t = vec3<R>{P.fwd[0], P.fwd[4], P.fwd[8]};
                sp.f.b = vec3<R>{P.fwd[2], P.fwd[6], P.fwd[10]};
            }
        }
        return sp;
    }

    // uniform-area point on an emitter: surface::sample_p
    template<typename R>
    FC_D void light_sample_point(const SceneDev<R>& S, const LightRec<R>& Lt, double u_prim, R ux, R uy, vec3<R>& pL, vec3<R>& nL)
    {
        if(Lt.kind == KIND_TRI)
        {
            // mesh_surface.hpp:276-295 + distribution_1d::sample_discrete, distribution.hpp:94-105
            const double* cdf = S.light_cdf + Lt.cdf_offset;
            double u = u_prim;  // already in [0, 1 - 2^-24]
            // upper_bound(cdf, u) - 1 = the last k with cdf[k] <= u. The search is a chain of dependent loads on this
            // kernel's critical path, so first try the cell a uniform-area mesh would give: two independent loads
            // confirm it (cdf[g] <= u < cdf[g + 1] holds for exactly that k); any other case takes the search.
            uint32_t tri = static_cast<uint32_t>(u * static_cast<double>(Lt.tri_count));
            if(tri >= Lt.tri_count) tri = Lt.tri_count - 1;
            // One round trip for the three: both cells are requested before the first is compared (`&`, not `&&`: the
            // short-circuit form made the second load wait for the first comparison), and the primitive of the guessed
            // cell with them (used unless the search below picks another triangle).
            const double c0 = cdf[tri], c1 = cdf[tri + 1];
            uint32_t record = S.prim_info[Lt.first_prim + tri].kind_index;
            if(!(static_cast<int>(c0 <= u) & static_cast<int>(c1 > u)))
            {
                uint32_t lo = 0, hi = Lt.tri_count + 1;  // upper_bound over cdf[0..tri_count]
                while(lo < hi)
                {
                    uint32_t mid = (lo + hi) >> 1;
                    if(cdf[mid] > u)
                        hi = mid;
                    else
                        lo = mid + 1;
                }
                tri = lo - 1;
                if(tri >= Lt.tri_count) tri = Lt.tri_count - 1;
                record = S.prim_info[Lt.first_prim + tri].kind_index;
            }
            const TriRec& T = S.tris[record & 0x3FFFFFFFu];
            vec3<R> p0 = ldv<R>(T.v0), p1 = ldv<R>(T.v1), p2 = ldv<R>(T.v2);
            R su0 = r_sqrt(ux);  // sample_triangle_uniform, sampling.hpp:62-66
            R bx = R(1) - su0, by = uy * su0;
            pL = p0 * bx + p1 * by + p2 * (R(1) - bx - by);
            nL = normalize(cross(p1 - p0, p2 - p0));
        }
        else
        {
            const XPrim<R>& P = S.xprims[Lt.index];
            if(Lt.kind == KIND_SPHERE)
            {
                vec3<R> n = sample_sphere_uniform(ux, uy);  // sphere_surface.hpp:157-172
                pL = xf_point(P.fwd, n * P.a);
                nL = xf_dir(P.fwd, n);
            }
            else
            {
                vec3<R> q{(ux - R(0.5)) * P.a, R(0), (uy - R(0.5)) * P.b};  // plane_surface.hpp:116-132
                pL = xf_point(P.fwd, q);
                nL = vec3<R>{P.fwd[1], P.fwd[5], P.fwd[9]};
            }
        }
    }

    // ------------------------------------------------------------------ shade
    // KEY in [0, L_COUNT): a path arrives at a hit, then scatters with lobe type KEY.
    // KEY == Q_FINAL: the last vertex (path length == max_path_length): arrival terms only.
    // Where in a trip k_shade requests the coming trip's records (software pipeline, see the kernel): 0 after the queue append,
    // 1 before it, 2 before the BSDF strategy, 3 before the light strategy. 1 is the default: the five loads then fly during
    // the append's three barriers and its returning atomic, with the same registers live as after it; measured against 0 on one
    // box, five alternating pairs: k_shade 70.1 -> 67.6 ms per step (-3.5 %), films bit-identical. 2 and 3 keep 21 more
    // registers live through the strategies and gain nothing.
#ifndef FCG_SHADE_LOADS_AT
#define FCG_SHADE_LOADS_AT 1
#endif
#ifndef FCG_SHADE_LIGHT_FIRST
#define FCG_SHADE_LIGHT_FIRST 0  // 1: k_shade samples the emitter before it reconstructs the hit (A/B of DESIGN.md section 5)
#endif
    // The untextured fp32 Lambert instantiation (a quarter of the headline step) is held to the 128 registers of four waves per
    // SIMD, where it sat by itself before the loads moved (the allocator takes 129 with them in front of the append; the limit
    // costs no scratch beyond the 20 B every instantiation has), its textured twin to the 168 of three (167 before the light
    // record was loaded by value, 169 after). 0 = no limit.
#ifndef FCG_SHADE_WAVES
#define FCG_SHADE_WAVES 4
#endif
#if FCG_SHADE_WAVES
    // (the textured microfacet-transmission instantiation sat at 168 too and is held there like the textured Lambert one)
    constexpr int shade_waves(bool f32, int key, bool tex)
    {
        return !f32 ? 1 : key == L_LAMBERT ? (tex ? FCG_SHADE_WAVES - 1 : FCG_SHADE_WAVES) : (tex && key == L_GGX_TRANS ? FCG_SHADE_WAVES - 1 : 1);
    }
#define FCG_SHADE_ATTR __attribute__((amdgpu_waves_per_eu(shade_waves(sizeof(R) == 4, KEY, TEX))))
#else
#define FCG_SHADE_ATTR
#endif
    // FCG_SHADE_CLOCKS (development build): wave cycles between marks of a k_shade<float, Lambert> trip, summed into the
    // statistics words behind ST_SHADE_PH0 and printed by fcgpu_get_counters. A mark waits for the value named (or, WAIT = 1,
    // for every outstanding memory operation) so that the time is booked where the wave really sat.
#ifdef FCG_SHADE_CLOCKS
#define SHADE_MARK(k, WAIT, value)                                                  \
    do                                                                              \
    {                                                                               \
        if constexpr(KEY == L_LAMBERT && !TEX && sizeof(R) == 4)                    \
        {                                                                           \
            asm volatile("" ::"v"(value));                                          \
            if(WAIT) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");    \
            const unsigned long long now_ = clock64();                              \
            sc_ph[k] += now_ - sc_t;                                                \
            sc_t = now_;                                                            \
        }                                                                           \
    } while(0)
#else
#define SHADE_MARK(k, WAIT, value)
#endif
    template<typename R, int KEY, bool TEX>
    __global__ __launch_bounds__(BLOCK) FCG_SHADE_ATTR void k_shade(SceneDev<R> S, Wavefront<R> Q, Batch B)
    {
#ifdef FCG_SHADE_CLOCKS
        unsigned long long sc_ph[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, sc_t = clock64();
#endif
        __shared__ uint32_t q_lds[2][WAVES];
        __shared__ uint32_t* q_ctr[2];
        if(threadIdx.x == 0)
        {
            q_ctr[0] = &Q.counters->ray[B.parity ^ 1][0];
            q_ctr[1] = &Q.counters->shadow[0];
        }
        __syncthreads();

        const uint32_t n = Q.counters->hit[KEY][0];
        const bool mis = B.integrator == FCGPU_INTEGRATOR_MIS;
        const uint32_t dim = 2u + (mis ? 6u : 3u) * static_cast<uint32_t>(B.h - 1);
        const uint32_t dim_u2 = dim + (mis ? 4u : 1u), dim_u1 = dim + (mis ? 5u : 2u);
        const uint32_t* hit_idx = Q.hit_idx[KEY];

        // Software pipeline over the trips of the grid-stride loop. A trip's critical path used to be three memory round
        // trips in a row (the acknowledgement of the previous trip's stores, its index hit_idx[i], then the records at that
        // index) before the first gather of the shading itself could start. The records of trip k + 1 (and the index of trip
        // k + 2) are now requested at the end of trip k, after the queue append and before the stores: few values are
        // live there, so the 21 registers in flight do not add to the kernel's peak, and the loads share their wait
        // with the stores' instead of following it.
        const uint32_t stride = gridDim.x * BLOCK;
        uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
        Vec4<R> nx_h0{}, nx_h1{}, nx_s0{}, nx_s1{}, nx_s2{};  // records of the coming trip
        uint32_t hi_next = 0;                                  // index of the trip after it
        if(i < n)
        {
            const uint32_t hi = hit_idx[i];
            nx_h0 = ld_stream(&Q.hit_c0[hi]);
            nx_h1 = ld_stream(&Q.hit_c1[hi]);
            nx_s0 = ld_stream(&Q.st_c0[B.parity][hi]);
            nx_s1 = ld_stream(&Q.st_c1[B.parity][hi]);
            nx_s2 = ld_stream_xyz(&Q.st_c2[B.parity][hi]);
        }
        if(static_cast<uint64_t>(i) + stride < n) hi_next = hit_idx[i + stride];  // else 0: a valid record that is dropped

        for(uint32_t base = blockIdx.x * BLOCK; base < n; base += stride, i += stride)
        {
            bool emit_ray = false, emit_shadow = false;
            Vec4<R> r0{}, r1{}, s0{}, s1{}, s2{};
            PathState<R> nst{};  // the state that travels on with the extension ray (written beside it, below)
            // the coming trip's records and the index of the one after it (see the top of the loop); unconditional (a trip past
            // the end re-reads record 0 and drops it): a conditional request would keep the old values alive through the whole
            // trip. Where in the trip it is issued: FCG_SHADE_LOADS_AT.
            auto request_next = [&]()
            {
                const uint64_t i2 = static_cast<uint64_t>(i) + stride + stride;
                nx_h0 = ld_stream(&Q.hit_c0[hi_next]);
                nx_h1 = ld_stream(&Q.hit_c1[hi_next]);
                nx_s0 = ld_stream(&Q.st_c0[B.parity][hi_next]);
                nx_s1 = ld_stream(&Q.st_c1[B.parity][hi_next]);
                nx_s2 = ld_stream_xyz(&Q.st_c2[B.parity][hi_next]);
                hi_next = hit_idx[i2 < n ? static_cast<uint32_t>(i2) : 0u];
            };
            if(i < n)
            {
                const Vec4<R> h0 = nx_h0, h1 = nx_h1;
                SHADE_MARK(0, 0, nx_s2.z + nx_s1.z + nx_s0.z + nx_h0.z + nx_h1.x);
                uint32_t hw, path;  // hw = hit_word(kind | index, lobe)
                unpack_ids(h1, hw, path);
                uint32_t ms = unpack_word(h1.y);
                PathState<R> st = state_unpack(nx_s0, nx_s1, nx_s2);
                vec3<R> beta{st.beta[0], st.beta[1], st.beta[2]};
                const vec3<R> p1{st.p1[0], st.p1[1], st.p1[2]};
                const vec3<R> w12{st.w[0], st.w[1], st.w[2]};  // direction that reached this hit
#if FCG_SHADE_LIGHT_FIRST
                // The light pick and the point on the emitter depend on the path's sample stream only, not on the hit: their
                // chain of dependent gathers (light record -> CDF cells -> primitive -> triangle) is started here, so that it
                // runs beside the hit's own chain (shading record -> entity -> material -> lobe) instead of after it.
                // Same arithmetic, same results; the light strategy below uses what is prepared here.
                constexpr bool LIGHT_FIRST = KEY != Q_FINAL && !(KEY == L_SPEC_REFL || KEY == L_SPEC_GLASS || KEY == L_SPEC_TRANS);
                uint32_t lf_li = 0;
                R lf_ux = R(0), lf_uy = R(0);
                vec3<R> lf_pL{}, lf_nL{};
                if constexpr(LIGHT_FIRST)
                {
                    if(mis && S.light_count > 0u)
                    {
                        const fcgpu_sample_stream lkey = slot_stream(B, path);
                        const uint64_t bits = fcgpu_stream_bits(&B.sampler, lkey, dim + 1u);
                        lf_li = static_cast<uint32_t>((static_cast<uint64_t>(fcgpu_sample_x24(bits)) * S.light_count) >> 24);
                        if(lf_li >= S.light_count) lf_li = S.light_count - 1u;  // uniform_light_distribution.hpp:13-17
                        draw(B, lkey, dim + 2u, lf_ux, lf_uy);
                        if(!(S.env_kind != FCGPU_ENV_NONE && lf_li == S.light_count - 1u))
                        {
                            const uint64_t pbits = fcgpu_stream_bits(&B.sampler, lkey, dim + 3u);
                            const LightRec<R> lf_Lt = S.lights[lf_li];
                            light_sample_point(S, lf_Lt, fcgpu_sample_x24(pbits) * FCGPU_SAMPLER_SCALE, lf_ux, lf_uy, lf_pL, lf_nL);
                        }
                    }
                }
#endif
                SurfacePoint<R> sp = make_surface_point(S, hit_word_kind(hw), hit_word_index(hw), vec3<R>{h0.x, h0.y, h0.z}, h0.w, h1.x);
                const EntityRec ent = S.entities[sp.entity];
                const vec3<R> ng = sp.f.ng;
                const uint32_t above = ms_above(ms), below = ms_below(ms);
                SHADE_MARK(1, 0, ent.material);

                // ---- arrival at p2 = sp.p along w12 (forward_mis_integrator.hpp:128-147 / :181-196)
                if(B.h > 1)
                {
                    const bool entering = dot(ng, w12) <= R(0);
                    const MediumRec<R> M = S.media[entering ? above : below];  // by value: the three coefficients in one round trip
                    // vacuum_medium (medium.hpp:34-37) and a uniform_medium without density: exp(-0 * d) = 1 exactly, so
                    // the three exponentials and the distance are only computed inside a medium that absorbs
                    if(static_cast<int>(M.sigma[0] != R(0)) | static_cast<int>(M.sigma[1] != R(0)) | static_cast<int>(M.sigma[2] != R(0)))
                    {
                        R dist = length(sp.p - p1);
                        beta = beta * vec3<R>{r_exp(-M.sigma[0] * dist), r_exp(-M.sigma[1] * dist), r_exp(-M.sigma[2] * dist)};  // medium.hpp:47-50
                    }
                }
                if(ent.light >= 0)
                {
                    // an emitter was hit: its term goes straight to the path's radiance column (a rare, exclusive RMW:
                    // a path is in one queue entry per launch)
                    const LightRec<R>& Lt = S.lights[ent.light];
                    vec3<R> Le = dot(ng, -w12) > R(0) ? rgb(Lt.Le) : vec3<R>{R(0), R(0), R(0)};  // const_diffuse_area_light.hpp:20-26
                    vec3<R> add;
                    if((st.flags & 1u) || !mis)
                    {
                        add = beta * Le;  // :40-43, :193-196
                    }
                    else
                    {
                        R pdf_light = R(1) / R(S.light_count);
                        R pdf_light_p2 = pdf_light * Lt.inv_area;
                        vec3<R> d12 = sp.p - p1;
                        R pdf_bsdf_p2 = st.pdf * r_abs(dot(ng, w12)) / dot(d12, d12);
                        R weight = power_heuristic(pdf_bsdf_p2, pdf_light_p2);
                        add = weight * (beta * Le);  // :140-147
                    }
                    Vec4<R> acc = Q.li_shadow[path];
                    acc.x += add.x;
                    acc.y += add.y;
                    acc.z += add.z;
                    Q.li_shadow[path] = acc;
                }

                SHADE_MARK(2, 1, beta.x);
                bool alive = false;
                R pdf_w12 = R(0);
                vec3<R> wn{};
                uint32_t ms_ext_out = 0;
                if constexpr(KEY != Q_FINAL)
                {
                    // ---- scatter at p1 := sp.p (forward_mis_integrator.hpp:47-112 / :153-168)
                    const vec3<R> w10 = -w12;
                    const MaterialRec mat = S.materials[ent.material];
                    LobeRec<R> L;
                    // normal_mapping<T> wraps the lobes of diffuse / mirror / standard materials only
                    constexpr bool NM = TEX && (KEY == L_LAMBERT || KEY == L_GGX_REFL || KEY == L_SPEC_REFL);
                    NormalMapRec<R> nmap;
                    if(TEX && mat.textured)
                    {
                        // material::evaluate at this hit's uv (fc_material.h); k_extend picked the lobe from the same values
                        R tu, tv;
                        surface_uv(S, hit_word_kind(hw), hit_word_index(hw), sp.p, h0.w, h1.x, tu, tv);
                        const MatValues<R> val = material_values(S, static_cast<uint32_t>(ent.material), mat, tu, tv);
                        int types[MAX_LOBES];
                        double scales[MAX_LOBES], weights[MAX_LOBES], factor[MAX_LOBES], cdf[MAX_LOBES];
                        const int count = material_layout(mat.kind, val, types, scales, weights);
                        lobe_pick_table(count, scales, weights, factor, cdf);
                        const int l = static_cast<int>(hit_word_lobe(hw));
                        L = material_lobe(mat.kind, val, l, types[l], factor[l], cdf[l]);
                        if constexpr(NM)
                            if(mat.tex[TEX_NORMAL]) nmap = make_normal_map(tex_evaluate(S.textures[mat.tex[TEX_NORMAL] - 1], tu, tv));
                    }
                    else
                    {
                        L = S.lobes[mat.first_lobe + hit_word_lobe(hw)];
                    }
                    const R eta_a = S.media[above].ior, eta_b = S.media[below].ior;
                    SHADE_MARK(3, 1, eta_a + eta_b);
                    const fcgpu_sample_stream key = slot_stream(B, path);
                    constexpr bool DELTA = KEY == L_SPEC_REFL || KEY == L_SPEC_GLASS || KEY == L_SPEC_TRANS;

                    if constexpr(FCG_SHADE_LOADS_AT == 3 && !DELTA) request_next();
                    if(!DELTA && mis && S.light_count > 0u)
                    {
                        // light strategy, :52-99
#if FCG_SHADE_LIGHT_FIRST
                        const uint32_t li = lf_li;
                        const R ux = lf_ux, uy = lf_uy;
#else
                        uint64_t bits = fcgpu_stream_bits(&B.sampler, key, dim + 1u);
                        uint32_t li = static_cast<uint32_t>((static_cast<uint64_t>(fcgpu_sample_x24(bits)) * S.light_count) >> 24);
                        if(li >= S.light_count) li = S.light_count - 1u;  // uniform_light_distribution.hpp:13-17
                        R ux, uy;
                        draw(B, key, dim + 2u, ux, uy);
#endif
                        R pdf_light = R(1) / R(S.light_count);
                        const bool is_env = S.env_kind != FCGPU_ENV_NONE && li == S.light_count - 1u;
                        if(is_env)
                        {
                            // infinity_area_light::sample_wi: const_infinity_area_light.hpp:33-44 /
                            // texture_infinity_area_light.hpp:70-96; then forward_mis_integrator.hpp:56-72
                            vec3<R> wi, Le;
                            R pdf_wi;
                            if(env_sample(S, ux, uy, wi, pdf_wi, Le))
                            {
                                vec3<R> f = bsdf_evaluate<KEY, NM>(L, sp.f, w10, wi, eta_a, eta_b, &nmap);
                                if(f.x != R(0) || f.y != R(0) || f.z != R(0))
                                {
                                    R pdf_bsdf = bsdf_pdf_wi<KEY, NM>(L, sp.f, w10, wi, eta_a, eta_b, &nmap);
                                    R pdf_l = pdf_light * pdf_wi;
                                    R weight = power_heuristic(pdf_l, pdf_bsdf);
                                    vec3<R> c = (beta * f * Le) * (weight * r_abs(dot(ng, wi)) / pdf_l);  // :68
                                    vec3<R> o = lift_origin(S, sp.p, ng, wi, dot(ng, wi) > R(0));  // scene.hpp:144-157
                                    emit_shadow = true;
                                    s0 = Vec4<R>{o.x, o.y, o.z, wi.x};
                                    s1 = Vec4<R>{wi.y, wi.z, consts<R>::inf, pack_word(R(0), path)};
                                    s2 = Vec4<R>{c.x, c.y, c.z, R(0)};
                                }
                            }
                        }
                        else
                        {
                            // standard_light::sample_p, const_diffuse_area_light.hpp:28-42
                            // by value: every field in one round trip (through a reference each field was fetched where it
                            // is first used, i.e. one dependent round trip per basic block: kind, counts, radiance, area)
                            const LightRec<R> Lt = S.lights[li];
#if FCG_SHADE_LIGHT_FIRST
                            const vec3<R> pL = lf_pL, nL = lf_nL;
#else
                            uint64_t pbits = fcgpu_stream_bits(&B.sampler, key, dim + 3u);
                            double u_prim = fcgpu_sample_x24(pbits) * FCGPU_SAMPLER_SCALE;
                            vec3<R> pL, nL;
                            light_sample_point(S, Lt, u_prim, ux, uy, pL, nL);
#endif
                            vec3<R> LeL = dot(nL, sp.p - pL) > R(0) ? rgb(Lt.Le) : vec3<R>{R(0), R(0), R(0)};
                            vec3<R> d1L = pL - sp.p;
                            vec3<R> w1L = normalize(d1L);
                            vec3<R> f = bsdf_evaluate<KEY, NM>(L, sp.f, w10, w1L, eta_a, eta_b, &nmap);
                            if(f.x != R(0) || f.y != R(0) || f.z != R(0))
                            {
                                R x = r_abs(dot(nL, w1L)) / dot(d1L, d1L);
                                R G = r_abs(dot(ng, w1L)) * x;
                                R pdf_bsdf_pL = bsdf_pdf_wi<KEY, NM>(L, sp.f, w10, w1L, eta_a, eta_b, &nmap) * x;
                                R pdf_light_pL = pdf_light * Lt.inv_area;
                                R weight = power_heuristic(pdf_light_pL, pdf_bsdf_pL);
                                vec3<R> c = (beta * f * G * LeL) * (weight / pdf_light_pL);  // :86-91
                                // entity_scene::visibility(p0, p1), scene.hpp:110-142
                                vec3<R> q0 = lift_origin(S, sp.p, ng, w1L, dot(d1L, ng) > R(0));
                                vec3<R> q1 = lift_origin(S, pL, nL, -w1L, dot(d1L, nL) < R(0));
                                vec3<R> to1 = q1 - q0;
                                R len = length(to1);
                                vec3<R> w01 = to1 / len;
                                emit_shadow = true;
                                s0 = Vec4<R>{q0.x, q0.y, q0.z, w01.x};
                                s1 = Vec4<R>{w01.y, w01.z, len, pack_word(R(0), path)};
                                s2 = Vec4<R>{c.x, c.y, c.z, R(0)};
                            }
                        }
                    }

                    SHADE_MARK(4, 1, s2.x + s0.x);
                    if constexpr(FCG_SHADE_LOADS_AT == 2 && !DELTA) request_next();
                    // BSDF strategy, :101-113 / :157-169 (u2 = first draw, u1 = second: right-to-left arguments)
                    R u1x, u1y, u2x, u2y;
                    draw(B, key, dim_u2, u2x, u2y);
                    draw(B, key, dim_u1, u1x, u1y);
                    vec3<R> value;
                    if(bsdf_sample_wi<KEY, NM>(L, sp.f, w10, eta_a, eta_b, u1x, u1y, u2x, u2y, wn, value, pdf_w12, &nmap))
                    {
                        beta = beta * (value * (r_abs(dot(ng, wn)) / pdf_w12));  // :112
                        alive = true;
                        // helper::raycast entry: leaving p into its own medium pushes it, integrator.hpp:36-44
                        // (a path nested deeper than five media keeps entries 5..8 beside its state: MStack)
                        MStack M{ms, 0u};
                        if(ms_deep(ms)) M.ext = unpack_word(Q.st_c2[B.parity][hit_idx[i]].w);
                        uint32_t overflow = 0, deep = 0;
                        if(ent.medium != 0 && dot(wn, ng) <= R(0)) medium_push(M, static_cast<uint32_t>(ent.medium), overflow, deep);
                        if(overflow) atomicAdd(&Q.stats[ST_MEDIUM_OVERFLOW], 1ull);
                        if(deep) atomicAdd(&Q.stats[ST_MEDIUM_DEEP], 1ull);
                        vec3<R> o = lift_origin(S, sp.p, ng, wn, dot(ng, wn) > R(0));
                        emit_ray = true;
                        r0 = Vec4<R>{o.x, o.y, o.z, wn.x};
                        r1 = Vec4<R>{wn.y, wn.z, R(0), R(0)};
                        pack_ids(r1, path, ms_strip_media(M.w));
                        ms_ext_out = M.ext;
                        st.flags = DELTA ? 1u : 0u;
                    }
                }

                if(Q.dec_flags)  // parity hook: what was decided at this hit
                {
                    uint32_t f = DEC_SHADED | (static_cast<uint32_t>(KEY) << DEC_KEY_SHIFT) | (alive ? DEC_SAMPLED : 0u) | (emit_shadow ? DEC_SHADOW_CAST : 0u);
                    if(alive && dot(wn, ng) * dot(w12, ng) > R(0)) f |= DEC_TRANSMITTED;  // w12 arrives, wn leaves: same side of ng = went through
                    Q.dec_flags[static_cast<size_t>(B.h - 1) * Q.dec_stride + path] = f;
                }
                if(alive)  // a path that ends here leaves nothing behind: what it gathered is in the radiance column
                {
                    nst.beta[0] = beta.x; nst.beta[1] = beta.y; nst.beta[2] = beta.z;
                    nst.pdf = pdf_w12;
                    nst.p1[0] = sp.p.x; nst.p1[1] = sp.p.y; nst.p1[2] = sp.p.z;
                    nst.w[0] = wn.x; nst.w[1] = wn.y; nst.w[2] = wn.z;
                    nst.flags = st.flags;
                    nst.ms_ext = pack_word(R(0), ms_ext_out);
                }
            }
            if constexpr(FCG_SHADE_LOADS_AT >= 2 && KEY != Q_FINAL && !(KEY == L_SPEC_REFL || KEY == L_SPEC_GLASS || KEY == L_SPEC_TRANS))
                if(i >= n) request_next();  // lanes without a hit in this (last) trip: their records are dropped either way
            SHADE_MARK(5, 1, r0.x + nst.beta[0]);
            uint32_t slot_r = 0, slot_s = 0;
            if constexpr(FCG_SHADE_LOADS_AT == 1 || KEY == Q_FINAL || (FCG_SHADE_LOADS_AT >= 2 && (KEY == L_SPEC_REFL || KEY == L_SPEC_GLASS || KEY == L_SPEC_TRANS))) request_next();
            // two independent appends share one pass: queue 0 = extension rays, queue 1 = shadow rays
            if constexpr(KEY != Q_FINAL) block_append2(emit_ray, emit_shadow, q_ctr, q_lds, slot_r, slot_s);
            SHADE_MARK(6, 0, slot_r + slot_s);
            if constexpr(FCG_SHADE_LOADS_AT == 0 && KEY != Q_FINAL) request_next();
            if constexpr(KEY != Q_FINAL)
            {
                if(emit_ray)
                {
                    st_stream(&Q.ray_c0[slot_r], r0);
                    st_stream(&Q.ray_c1[slot_r], r1);
                    state_store(Q, B.parity ^ 1, slot_r, nst);
                }
                if(emit_shadow)
                {
                    st_stream(&Q.sh_c0[slot_s], s0);
                    st_stream(&Q.sh_c1[slot_s], s1);
                    st_stream(&Q.sh_c2[slot_s], s2);
                }
            }
            SHADE_MARK(7, 0, slot_r);
#ifdef FCG_SHADE_CLOCKS
            sc_ph[8] += 1;
#endif
        }
#ifdef FCG_SHADE_CLOCKS
        if constexpr(KEY == L_LAMBERT && !TEX && sizeof(R) == 4)
            if((threadIdx.x & 63) == 0)
                for(int k = 0; k < 9; ++k) atomicAdd(&Q.stats[ST_SHADE_PH0 + k], sc_ph[k]);
#endif
    }

    // paths that left the scene: environment term (forward_mis_integrator.hpp:29-35, 115-125, 171-178)
    template<typename R>
    __global__ __launch_bounds__(BLOCK) void k_miss(SceneDev<R> S, Wavefront<R> Q, Batch B)
    {
        const uint32_t n = Q.counters->hit[Q_MISS][0];
        const bool mis = B.integrator == FCGPU_INTEGRATOR_MIS;
        for(uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK)
        {
            const uint32_t hi = Q.hit_idx[Q_MISS][i];
            Vec4<R> h1 = Q.hit_c1[hi];
            uint32_t prim, path;
            unpack_ids(h1, prim, path);
            const PathState<R> st = state_load(Q, B.parity, hi);
            vec3<R> beta{st.beta[0], st.beta[1], st.beta[2]};
            const vec3<R> w{st.w[0], st.w[1], st.w[2]};  // direction of the ray that left the scene
            vec3<R> env = env_Li(S, w);
            R weight = R(1);
            if(B.h == 1)
            {
                if(mis && !B.visible_env) weight = R(0);
            }
            else if(mis && !(st.flags & 1u))
            {
                R pdf_light = R(1) / R(S.light_count);
                R pdf_light_w12 = pdf_light * env_pdf(S, w);
                weight = power_heuristic(st.pdf, pdf_light_w12);
            }
            if(weight != R(0) || weight != weight)
            {
                Vec4<R> acc = Q.li_shadow[path];
                acc.x += weight * beta.x * env.x;
                acc.y += weight * beta.y * env.y;
                acc.z += weight * beta.z * env.z;
                Q.li_shadow[path] = acc;
            }
        }
    }

    // ------------------------------------------------------------------ shadow
    // CAP: entries of a lane's stack column in `stack_lds`
    template<typename R, bool SPILL, bool WIDE8, int CAP>
    FC_D void shadow_body(const SceneDev<R>& S, const Wavefront<R>& Q, const Batch& B, uint32_t* stack_lds, unsigned char* nodebuf_lds)
    {
        uint32_t* stack = stack_lds + threadIdx.x;
        unsigned char* nodebuf = FCG_TREELET ? nodebuf_lds : nodebuf_lds + (threadIdx.x >> 6) * NODEBUF_WAVE;
        treelet_fill(S, nodebuf_lds);
        const uint32_t n = Q.counters->shadow[0];
        uint32_t n_cast = 0, n_occ = 0, n_nodes = 0, n_prims = 0;
        StackStats n_overflow;
        WorkCursor W = work_begin(n);
        typename TravSel<R, WIDE8>::type T;
        trav_idle(T);
        uint32_t slot = INVALID_U32, path = 0;
        for(;;)
        {
            const bool finished = slot != INVALID_U32 && trav_done(T);
            const bool traversing = slot != INVALID_U32 && !trav_done(T);
            if(__popcll(__ballot(finished)) >= B.finish_min_shadow || __ballot(traversing) == 0ull)
            {
                if(finished)
                {
                    if(T.pk != INVALID_U32)
                    {
                        n_occ += 1;
                        if(Q.dec_flags) Q.dec_flags[static_cast<size_t>(B.h - 1) * Q.dec_stride + path] |= DEC_OCCLUDED;  // this lane holds the path's only shadow ray
                    }
                    else
                    {
                        // unoccluded: the light-strategy term of forward_mis_integrator.hpp:84-91
                        Vec4<R> c2 = ld_stream(&Q.sh_c2[slot]);
                        Vec4<R> acc = Q.li_shadow[path];
                        acc.x += c2.x;
                        acc.y += c2.y;
                        acc.z += c2.z;
                        Q.li_shadow[path] = acc;
                    }
                    slot = INVALID_U32;
                }
                uint32_t fresh = work_take(W, slot == INVALID_U32, Q.counters->work_shadow, n);
                if(fresh != INVALID_U32)
                {
                    slot = fresh;
                    Vec4<R> c0 = ld_stream(&Q.sh_c0[slot]), c1 = ld_stream(&Q.sh_c1[slot]);
                    path = unpack_word(c1.w);
                    trav_begin(T, vec3<R>{c0.x, c0.y, c0.z}, vec3<R>{c0.w, c1.x, c1.y}, c1.z);
                    n_cast += 1;
                }
                if(__ballot(slot != INVALID_U32) == 0ull) break;
            }
            if constexpr(WIDE8)
                trav_round8<R, true, BLOCK>(S, T, stack, n_overflow, B.descend_min_shadow, n_nodes, n_prims);
            else
                trav_round<R, true, BLOCK, SPILL, false, FCG_SHADOW_SPEC != 0, CAP>(S, T, stack, nodebuf, n_overflow, B.descend_min_shadow, n_nodes, n_prims);
        }
        wave_stat_add(Q.stats, ST_SHADOW, n_cast);
        wave_stat_add(Q.stats, ST_OCCLUDED, n_occ);
        if(SPILL) wave_stat_add(Q.stats, ST_STACK_OVERFLOW, n_overflow.lost); else wave_stat_add_uniform(Q.stats, ST_STACK_OVERFLOW, n_overflow.lost);
        wave_stat_add(Q.stats, ST_STACK_SPILL, n_overflow.spilled);
        wave_stat_add_uniform(Q.stats, ST_NODES, n_nodes);
        wave_stat_add(Q.stats, ST_PRIMS, n_prims);
    }

    template<typename R, bool SPILL, bool WIDE8 = false>
    __global__ __launch_bounds__(BLOCK) FCG_SHADOW_ATTR void k_shadow(SceneDev<R> S, Wavefront<R> Q, Batch B)
    {
        __shared__ uint32_t stack_lds[STACK_CAP * BLOCK];
        __shared__ __attribute__((aligned(16))) unsigned char nodebuf_lds[FCG_TREELET ? FCG_TREELET * 64 : WAVES * NODEBUF_WAVE];
        shadow_body<R, SPILL, WIDE8, STACK_CAP>(S, Q, B, stack_lds, nodebuf_lds);
    }

#if !defined(FCG_ONLY_F64) && FCG_SHADOW_DENSE
    // The product path's shadow kernel (fp32, 4-wide hierarchy): SHADOW_STACK_CAP entries in LDS and the registers of seven
    // waves per SIMD (fc_types.h). Same code, same results.
    template<bool SPILL>
    __global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(FCG_SHADOW_DENSE_WAVES))) void k_shadow_dense(SceneDev<float> S, Wavefront<float> Q, Batch B)
    {
        __shared__ uint32_t stack_lds[SHADOW_STACK_CAP * BLOCK];
        __shared__ __attribute__((aligned(16))) unsigned char nodebuf_lds[FCG_TREELET ? FCG_TREELET * 64 : WAVES * NODEBUF_WAVE];
        shadow_body<float, SPILL, false, SHADOW_STACK_CAP>(S, Q, B, stack_lds, nodebuf_lds);
    }
#endif

    // ------------------------------------------------------------------ accumulate
    // One thread per pixel of the batch, looping over the batch's samples: one writer per pixel,
    // so the film sum is deterministic (no float atomics).
    template<typename R>
    __global__ __launch_bounds__(BLOCK) void k_accumulate(SceneDev<R> S, Wavefront<R> Q, Batch B)
    {
        uint32_t n_nan = 0;
        for(uint32_t pp = blockIdx.x * BLOCK + threadIdx.x; pp < B.npix; pp += gridDim.x * BLOCK)
        {
            uint32_t pit = B.pix0 + pp;
            uint32_t row = fast_div(pit, B.div_tw);
            uint32_t px = static_cast<uint32_t>(B.x0) + (pit - row * static_cast<uint32_t>(B.tw));
            uint32_t py = static_cast<uint32_t>(B.y0) + row + fast_div(row, B.div_band) * B.band_skip;
            double* out = B.film + 3ull * (static_cast<uint64_t>(py) * static_cast<uint64_t>(B.W) + px);
            double r = out[0], g = out[1], b = out[2];
            for(uint32_t sl = 0; sl < B.nsamp; ++sl)
            {
                const Vec4<R> sh = Q.li_shadow[sl * B.npix + pp];  // everything the path gathered
                R x = sh.x, y = sh.y, z = sh.z;
                if(!r_isfinite(x) || !r_isfinite(y) || !r_isfinite(z))  // perspective_camera.hpp:126-131
                {
                    n_nan += 1;
                    continue;
                }
                r += static_cast<double>(x);
                g += static_cast<double>(y);
                b += static_cast<double>(z);
            }
            out[0] = r;
            out[1] = g;
            out[2] = b;
        }
        wave_stat_add(Q.stats, ST_NAN, n_nan);
    }

#ifndef FCG_ONLY_F64
    __global__ void k_reset_counters(QueueCounters* c, int parity_to_keep)
    {
        if(threadIdx.x < Q_COUNT) c->hit[threadIdx.x][0] = 0;
        if(threadIdx.x == 0)
        {
            c->ray[parity_to_keep ^ 1][0] = 0;
            c->shadow[0] = 0;
        }
        if(threadIdx.x < WORK_RANGES)
        {
            c->work_extend[threadIdx.x][0] = 0;
            c->work_shadow[threadIdx.x][0] = 0;
        }
    }
#endif

    // ------------------------------------------------------------------ parity hooks
    // whole-wave wrapper over the 8-wide rounds (like fc_traverse.h: traverse)
    template<typename R, bool ANY>
    FC_D bool traverse8(const SceneDev<R>& S, bool active, vec3<R> o, vec3<R> d, R t_max, uint32_t* stack, R& t_out, uint32_t& pk_out, R& b0_out, R& b1_out,
                        StackStats& overflow)
    {
        Trav8<R> T;
        trav_begin(T, o, d, t_max);
        if(!active) trav_idle(T);
        uint32_t nn = 0, np = 0;
        while(__ballot(!trav_done(T)) != 0ull) trav_round8<R, ANY, BLOCK>(S, T, stack, overflow, 12, nn, np);
        t_out = T.t_max;
        pk_out = T.pk;
        b0_out = T.b0;
        b1_out = T.b1;
        return T.pk != INVALID_U32;
    }

    template<typename R, bool SPILL, bool WIDE8 = false>
    __global__ __launch_bounds__(BLOCK) void k_trace(SceneDev<R> S, uint64_t n, const fcgpu_ray* rays, fcgpu_hit* hits,
                                                     uint8_t* occluded, unsigned long long* stats)
    {
        __shared__ uint32_t stack[STACK_CAP * BLOCK];
        __shared__ __attribute__((aligned(16))) unsigned char nodebuf_lds[FCG_TREELET ? FCG_TREELET * 64 : WAVES * NODEBUF_WAVE];
        unsigned char* nodebuf = FCG_TREELET ? nodebuf_lds : nodebuf_lds + (threadIdx.x >> 6) * NODEBUF_WAVE;
        treelet_fill(S, nodebuf_lds);
        StackStats n_overflow, dense_spill, dense_plain;
        // block-uniform trip count: every lane of a wave takes part in the cooperative node fetches
        for(uint64_t base = static_cast<uint64_t>(blockIdx.x) * BLOCK; base < n; base += static_cast<uint64_t>(gridDim.x) * BLOCK)
        {
            const uint64_t i = base + threadIdx.x;
            const bool active = i < n;
            fcgpu_ray r{};
            if(active) r = rays[i];
            RayCtx<R> rc = make_ray(vec3<R>{R(r.origin[0]), R(r.origin[1]), R(r.origin[2])},
                                    vec3<R>{R(r.direction[0]), R(r.direction[1]), R(r.direction[2])});
            R t, b0, b1;
            uint32_t pk;
            if(hits != nullptr)
            {
                fcgpu_hit h{};
                bool hit;
                if constexpr(WIDE8)
                    hit = traverse8<R, false>(S, active, rc.o, rc.d, R(r.t_max), stack + threadIdx.x, t, pk, b0, b1, n_overflow);
                else
                    hit = traverse<R, false, BLOCK, SPILL>(S, active, rc, R(r.t_max), stack + threadIdx.x, nodebuf, t, pk, b0, b1, n_overflow);
                if(hit)
                {
                    vec3<R> p, nrm;
                    uint32_t prim, entity, key_hint;
                    hit_point(S, rc, pk, t, b0, b1, p, nrm, prim, entity, key_hint);
                    h.t = static_cast<double>(t);
                    h.b0 = static_cast<double>(b0);
                    h.b1 = static_cast<double>(b1);
                    h.prim = prim;
                }
                else
                {
                    h.t = __builtin_huge_val();
                    h.prim = INVALID_U32;
                }
                if(active) hits[i] = h;
            }
            else
            {
                bool hit;
                if constexpr(WIDE8)
                    hit = traverse8<R, true>(S, active, rc.o, rc.d, R(r.t_max), stack + threadIdx.x, t, pk, b0, b1, n_overflow);
#if FCG_SHADOW_DENSE
                else if constexpr(sizeof(R) == 4)
                {
                    // occlusion rays of the fp32 path use the LDS stack the way k_shadow_dense does (SHADOW_STACK_CAP entries,
                    // the rest in the spill columns), so that fcgpu_trace_any tests that kernel's traversal
                    if(S.stack_spill && S.max_stack > static_cast<uint32_t>(SHADOW_STACK_CAP - 2))
                        hit = traverse<R, true, BLOCK, true, SHADOW_STACK_CAP>(S, active, rc, R(r.t_max), stack + threadIdx.x, nodebuf, t, pk, b0, b1, dense_spill);
                    else
                        hit = traverse<R, true, BLOCK, false, SHADOW_STACK_CAP>(S, active, rc, R(r.t_max), stack + threadIdx.x, nodebuf, t, pk, b0, b1, dense_plain);
                }
#endif
                else
                    hit = traverse<R, true, BLOCK, SPILL>(S, active, rc, R(r.t_max), stack + threadIdx.x, nodebuf, t, pk, b0, b1, n_overflow);
                if(active) occluded[i] = hit ? 1 : 0;
            }
        }
        if(SPILL) wave_stat_add(stats, ST_STACK_OVERFLOW, n_overflow.lost); else wave_stat_add_uniform(stats, ST_STACK_OVERFLOW, n_overflow.lost);
        wave_stat_add(stats, ST_STACK_SPILL, n_overflow.spilled);
        // the fp32 occlusion rays' own counts (per lane with spill code, per wave without: StackStats)
        wave_stat_add(stats, ST_STACK_OVERFLOW, dense_spill.lost);
        wave_stat_add(stats, ST_STACK_SPILL, dense_spill.spilled);
        wave_stat_add_uniform(stats, ST_STACK_OVERFLOW, dense_plain.lost);
    }

    // ------------------------------------------------------------------ scratch of fcgpu_trace_closest_noalloc
#ifndef FCG_ONLY_F64
    // One copy per GPU the library runs on, at that GPU's own address. In host code the NAME is only a stand-in for the
    // symbol: the address kernels may use comes from hipGetSymbolAddress, below, and nothing else is ever passed on.
    static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];

    hipError_t noalloc_scratch_for_device(int device, unsigned char** out)
    {
        constexpr int MAX_DEVICES = 64;  // beyond it: no scratch address, i.e. no fcgpu_trace_closest_noalloc on that device (everything else works)
        static_assert(STACK_CAP != 30 || STACK_CAP + static_cast<int>(NOALLOC_SPILL_ENTRIES) == FCGPU_NOALLOC_MAX_STACK, "include/fcgpu.h states the limit the shipped kernel has (other -DFCG_STACK_CAP values are tools/build_variant.sh experiments)");
        static_assert(static_cast<int>(NOALLOC_SPILL_ENTRIES) <= STACK_SPILL_MAX, "the kernel bounds spill pushes by STACK_SPILL_MAX: the static columns must hold that many");
        static std::mutex lock;
        static unsigned char* address[MAX_DEVICES] = {};
        *out = nullptr;
        if(device < 0 || device >= MAX_DEVICES) return hipErrorInvalidDevice;
        std::lock_guard<std::mutex> guard(lock);
        if(!address[device])
        {
            void* p = nullptr;
            const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));  // of the CURRENT device: the caller made `device` current
            if(e != hipSuccess) return e;
            address[device] = static_cast<unsigned char*>(p);
        }
        *out = address[device];
        return hipSuccess;
    }

    __global__ void k_fold_stats(unsigned long long* dst, const unsigned long long* src)
    {
        if(threadIdx.x < ST_COUNT && src[threadIdx.x]) atomicAdd(&dst[threadIdx.x], src[threadIdx.x]);
    }
    void launch_fold_stats(const LaunchCfg& L, unsigned long long* dst, const unsigned long long* src)
    {
        hipLaunchKernelGGL(k_fold_stats, dim3(1), dim3(64), 0, L.stream, dst, src);
    }
#endif

    // ------------------------------------------------------------------ multi-GPU film sum (fcgpu_reduce_film)
#ifndef FCG_ONLY_F64
    __global__ __launch_bounds__(BLOCK) void k_film_add(double* dst, const double* src, uint64_t n)
    {
        for(uint64_t i = static_cast<uint64_t>(blockIdx.x) * BLOCK + threadIdx.x; i < n; i += static_cast<uint64_t>(gridDim.x) * BLOCK) dst[i] += src[i];
    }
    void launch_film_add(const LaunchCfg& L, double* dst, const double* src, uint64_t n)
    {
        hipLaunchKernelGGL(k_film_add, dim3(L.grid), dim3(BLOCK), 0, L.stream, dst, src, n);
    }
    // the same over the rows of the bands b = first, first + stride, ... of `rows` film rows (row_elems doubles per row)
    __global__ __launch_bounds__(BLOCK) void k_film_add_bands(double* dst, const double* src, uint64_t row_elems, uint64_t owned_rows, uint32_t rows,
                                                              uint32_t first, uint32_t stride)
    {
        const uint64_t n = owned_rows * row_elems;
        for(uint64_t i = static_cast<uint64_t>(blockIdx.x) * BLOCK + threadIdx.x; i < n; i += static_cast<uint64_t>(gridDim.x) * BLOCK)
        {
            const uint64_t r = i / row_elems, band = r / rows;
            const uint64_t y = (first + band * stride) * rows + (r - band * rows);
            const uint64_t off = y * row_elems + (i - r * row_elems);
            dst[off] += src[off];
        }
    }
    void launch_film_add_bands(const LaunchCfg& L, double* dst, const double* src, uint64_t row_elems, uint64_t owned_rows, uint32_t rows,
                               uint32_t first, uint32_t stride)
    {
        hipLaunchKernelGGL(k_film_add_bands, dim3(L.grid), dim3(BLOCK), 0, L.stream, dst, src, row_elems, owned_rows, rows, first, stride);
    }
#endif

    // ------------------------------------------------------------------ per-lobe parity hook (fcgpu_bsdf_table)
    // One query per thread through the very functions k_extend / k_shade call: the lobe pick of bsdf::sample_bxdf
    // (bsdf.hpp:46-55), then bsdf::evaluate / pdf_wi / sample_wi (bsdf.hpp:62-100, 131-141) of the picked lobe.
    template<typename R>
    __global__ __launch_bounds__(BLOCK) void k_bsdf_table(const LobeRec<R>* lobes, int lobe_count, ShadingFrame<R> frame, R eta_a, R eta_b, uint64_t n,
                                                          const fcgpu_bsdf_query* queries, fcgpu_bsdf_result* results)
    {
        for(uint64_t i = static_cast<uint64_t>(blockIdx.x) * BLOCK + threadIdx.x; i < n; i += static_cast<uint64_t>(gridDim.x) * BLOCK)
        {
            const fcgpu_bsdf_query q = queries[i];
            int lobe = 0;
            for(int l = lobe_count - 1; l >= 0; --l)
            {
                if(q.u_lobe > lobes[l].cdf)
                {
                    lobe = l;
                    break;
                }
            }
            const LobeRec<R> L = lobes[lobe];
            const vec3<R> wo{R(q.wo[0]), R(q.wo[1]), R(q.wo[2])}, wi{R(q.wi[0]), R(q.wi[1]), R(q.wi[2])};
            vec3<R> f{}, s_wi{}, s_value{};
            R pdf = R(0), s_pdf = R(0);
            bool ok = false;
            auto run = [&](auto key) {
                constexpr int KEY = decltype(key)::value;
                f = bsdf_evaluate<KEY>(L, frame, wo, wi, eta_a, eta_b);
                pdf = bsdf_pdf_wi<KEY>(L, frame, wo, wi, eta_a, eta_b);
                ok = bsdf_sample_wi<KEY>(L, frame, wo, eta_a, eta_b, R(q.u1[0]), R(q.u1[1]), R(q.u2[0]), R(q.u2[1]), s_wi, s_value, s_pdf);
            };
            switch(L.type)
            {
            case L_LAMBERT: run(std::integral_constant<int, L_LAMBERT>{}); break;
            case L_GGX_REFL: run(std::integral_constant<int, L_GGX_REFL>{}); break;
            case L_SPEC_REFL: run(std::integral_constant<int, L_SPEC_REFL>{}); break;
            case L_GGX_GLASS: run(std::integral_constant<int, L_GGX_GLASS>{}); break;
            case L_SPEC_GLASS: run(std::integral_constant<int, L_SPEC_GLASS>{}); break;
            case L_GGX_TRANS: run(std::integral_constant<int, L_GGX_TRANS>{}); break;
            default: run(std::integral_constant<int, L_SPEC_TRANS>{}); break;
            }
            fcgpu_bsdf_result r{};
            r.lobe = lobe;
            r.is_delta = lobe_is_delta(L.type) ? 1 : 0;
            r.f[0] = static_cast<double>(f.x); r.f[1] = static_cast<double>(f.y); r.f[2] = static_cast<double>(f.z);
            r.pdf = static_cast<double>(pdf);
            r.s_ok = ok ? 1 : 0;
            if(ok)
            {
                r.s_wi[0] = static_cast<double>(s_wi.x); r.s_wi[1] = static_cast<double>(s_wi.y); r.s_wi[2] = static_cast<double>(s_wi.z);
                r.s_value[0] = static_cast<double>(s_value.x); r.s_value[1] = static_cast<double>(s_value.y); r.s_value[2] = static_cast<double>(s_value.z);
                r.s_pdf = static_cast<double>(s_pdf);
            }
            results[i] = r;
        }
    }
    template<typename R>
    void launch_bsdf_table(const LaunchCfg& L, const LobeRec<R>* lobes, int lobe_count, const double* frame, double eta_a, double eta_b, uint64_t n,
                           const fcgpu_bsdf_query* queries, fcgpu_bsdf_result* results)
    {
        auto v = [&](int k) { return vec3<R>{R(frame[3 * k]), R(frame[3 * k + 1]), R(frame[3 * k + 2])}; };
        ShadingFrame<R> f;
        f.ng = v(0);
        f.t = v(1);
        f.ns = v(2);
        f.b = v(3);
        const int grid = static_cast<int>(std::min<uint64_t>(static_cast<uint64_t>(L.grid), (n + BLOCK - 1) / BLOCK));
        hipLaunchKernelGGL(k_bsdf_table<R>, dim3(grid), dim3(BLOCK), 0, L.stream, lobes, lobe_count, f, R(eta_a), R(eta_b), n, queries, results);
    }

    // ------------------------------------------------------------------ launchers
    template<typename R> void launch_generate(const LaunchCfg& L, const SceneDev<R>& S, const Wavefront<R>& Q, const Batch& B)
    {
        hipLaunchKernelGGL(k_generate<R>, dim3(L.streaming()), dim3(BLOCK), 0, L.stream, S, Q, B);
    }
    // the kernels with STACK_CAP entries in LDS run their SPILL instantiation (fc_api.cpp: upload allocates the columns)
    template<typename R> static bool spills(const SceneDev<R>& S) { return S.stack_spill && S.max_stack > static_cast<uint32_t>(STACK_CAP - 2); }
    template<typename R> void launch_extend(const LaunchCfg& L, const SceneDev<R>& S, const Wavefront<R>& Q, const Batch& B)
    {
        // SPILL variants only for scenes whose BVH needs more stack than the LDS holds (S.stack_spill set at upload),
        // TEX variants only for scenes with textured materials
        const dim3 g(L.grid), b(BLOCK);
        if(S.nodes8)
        {
            if(S.textured_materials)
                hipLaunchKernelGGL((k_extend<R, false, false, true, true>), g, b, L.lds_pad_extend, L.stream, S, Q, B);
            else
                hipLaunchKernelGGL((k_extend<R, false, false, false, true>), g, b, L.lds_pad_extend, L.stream, S, Q, B);
        }
        else if(S.textured_materials)
        {
            if(spills(S))
                hipLaunchKernelGGL((k_extend<R, false, true, true>), g, b, L.lds_pad_extend, L.stream, S, Q, B);
            else
                hipLaunchKernelGGL((k_extend<R, false, false, true>), g, b, L.lds_pad_extend, L.stream, S, Q, B);
        }
        else if(spills(S))
            hipLaunchKernelGGL((k_extend<R, false, true, false>), g, b, L.lds_pad_extend, L.stream, S, Q, B);
        else if(Q.timeline)
            hipLaunchKernelGGL((k_extend<R, true, false, false>), g, b, L.lds_pad_extend, L.stream, S, Q, B);
        else
            hipLaunchKernelGGL((k_extend<R, false, false, false>), g, b, L.lds_pad_extend, L.stream, S, Q, B);
    }
    template<typename R> void launch_sort(const LaunchCfg& L, const Wavefront<R>& Q, const Batch& B)
    {
        hipLaunchKernelGGL(k_sort<R>, dim3(L.grid), dim3(BLOCK), 0, L.stream, Q, B);
    }
    template<typename R, bool TEX> void launch_shade_lobe(const LaunchCfg& L, const SceneDev<R>& S, const Wavefront<R>& Q, const Batch& B, int key)
    {
        const dim3 g(L.streaming()), b(BLOCK);
        switch(key)
        {
        case L_LAMBERT: hipLaunchKernelGGL((k_shade<R, L_LAMBERT, TEX>), g, b, 0, L.stream, S, Q, B); break;
        case L_GGX_REFL: hipLaunchKernelGGL((k_shade<R, L_GGX_REFL, TEX>), g, b, 0, L.stream, S, Q, B); break;
        case L_SPEC_REFL: hipLaunchKernelGGL((k_shade<R, L_SPEC_REFL, TEX>), g, b, 0, L.stream, S, Q, B); break;
        case L_GGX_GLASS: hipLaunchKernelGGL((k_shade<R, L_GGX_GLASS, TEX>), g, b, 0, L.stream, S, Q, B); break;
        case L_SPEC_GLASS: hipLaunchKernelGGL((k_shade<R, L_SPEC_GLASS, TEX>), g, b, 0, L.stream, S, Q, B); break;
        case L_GGX_TRANS: hipLaunchKernelGGL((k_shade<R, L_GGX_TRANS, TEX>), g, b, 0, L.stream, S, Q, B); break;
        case L_SPEC_TRANS: hipLaunchKernelGGL((k_shade<R, L_SPEC_TRANS, TEX>), g, b, 0, L.stream, S, Q, B); break;
        default: break;
        }
    }
    template<typename R> void launch_shade(const LaunchCfg& L, const SceneDev<R>& S, const Wavefront<R>& Q, const Batch& B, int key)
    {
        const dim3 g(L.streaming()), b(BLOCK);
        if(key == Q_FINAL)
            hipLaunchKernelGGL((k_shade<R, Q_FINAL, false>), g, b, 0, L.stream, S, Q, B);  // arrival terms only: no lobe is built
        else if(key == Q_MISS)
            hipLaunchKernelGGL(k_miss<R>, g, b, 0, L.stream, S, Q, B);
        else if(S.textured_materials)
            launch_shade_lobe<R, true>(L, S, Q, B, key);
        else
            launch_shade_lobe<R, false>(L, S, Q, B, key);
    }
    template<typename R> void launch_shadow(const LaunchCfg& L, const SceneDev<R>& S, const Wavefront<R>& Q, const Batch& B)
    {
        if(S.nodes8)
        {
            hipLaunchKernelGGL((k_shadow<R, false, true>), dim3(L.grid), dim3(BLOCK), L.lds_pad_shadow, L.stream, S, Q, B);
            return;
        }
#if !defined(FCG_ONLY_F64) && FCG_SHADOW_DENSE
        if constexpr(sizeof(R) == 4)
        {
            if(S.stack_spill && S.max_stack > static_cast<uint32_t>(SHADOW_STACK_CAP - 2))
                hipLaunchKernelGGL((k_shadow_dense<true>), dim3(L.grid), dim3(BLOCK), L.lds_pad_shadow, L.stream, S, Q, B);
            else
                hipLaunchKernelGGL((k_shadow_dense<false>), dim3(L.grid), dim3(BLOCK), L.lds_pad_shadow, L.stream, S, Q, B);
            return;
        }
#endif
        if(spills(S))
            hipLaunchKernelGGL((k_shadow<R, true>), dim3(L.grid), dim3(BLOCK), L.lds_pad_shadow, L.stream, S, Q, B);
        else
            hipLaunchKernelGGL((k_shadow<R, false>), dim3(L.grid), dim3(BLOCK), L.lds_pad_shadow, L.stream, S, Q, B);
    }
    template<typename R> void launch_accumulate(const LaunchCfg& L, const SceneDev<R>& S, const Wavefront<R>& Q, const Batch& B)
    {
        hipLaunchKernelGGL(k_accumulate<R>, dim3(L.streaming()), dim3(BLOCK), 0, L.stream, S, Q, B);
    }
#ifndef FCG_ONLY_F64
    void launch_reset_counters(const LaunchCfg& L, QueueCounters* c, int parity_to_keep)
    {
        hipLaunchKernelGGL(k_reset_counters, dim3(1), dim3(64), 0, L.stream, c, parity_to_keep);
    }
#endif
    template<typename R> void launch_trace(const LaunchCfg& L, const SceneDev<R>& S, uint64_t n, const fcgpu_ray* rays,
                                           fcgpu_hit* hits, uint8_t* occluded, unsigned long long* stats)
    {
        if(S.nodes8)
            hipLaunchKernelGGL((k_trace<R, false, true>), dim3(L.grid), dim3(BLOCK), 0, L.stream, S, n, rays, hits, occluded, stats);
        else if(spills(S))
            hipLaunchKernelGGL((k_trace<R, true>), dim3(L.grid), dim3(BLOCK), 0, L.stream, S, n, rays, hits, occluded, stats);
        else
            hipLaunchKernelGGL((k_trace<R, false>), dim3(L.grid), dim3(BLOCK), 0, L.stream, S, n, rays, hits, occluded, stats);
    }

#define FCG_INSTANTIATE(R)                                                                                             \
    template void launch_generate<R>(const LaunchCfg&, const SceneDev<R>&, const Wavefront<R>&, const Batch&);          \
    template void launch_extend<R>(const LaunchCfg&, const SceneDev<R>&, const Wavefront<R>&, const Batch&);            \
    template void launch_sort<R>(const LaunchCfg&, const Wavefront<R>&, const Batch&);                                  \
    template void launch_shade<R>(const LaunchCfg&, const SceneDev<R>&, const Wavefront<R>&, const Batch&, int);        \
    template void launch_shadow<R>(const LaunchCfg&, const SceneDev<R>&, const Wavefront<R>&, const Batch&);            \
    template void launch_accumulate<R>(const LaunchCfg&, const SceneDev<R>&, const Wavefront<R>&, const Batch&);        \
    template void launch_trace<R>(const LaunchCfg&, const SceneDev<R>&, uint64_t, const fcgpu_ray*, fcgpu_hit*, uint8_t*, unsigned long long*); \
    template void launch_bsdf_table<R>(const LaunchCfg&, const LobeRec<R>*, int, const double*, double, double, uint64_t, const fcgpu_bsdf_query*, fcgpu_bsdf_result*);
    // The library compiles this file twice (Makefile): the fp32 product kernels with the compiler's default fused
    // multiply-adds, the fp64 parity kernels with -ffp-contract=off, so that decisions a rounding can flip (a sign test at
    // grazing incidence, a zero test on a BSDF value) fall as they do in the reference, which is built without FMA.
#ifndef FCG_ONLY_F64
    FCG_INSTANTIATE(float)
#endif
#ifndef FCG_ONLY_F32
    FCG_INSTANTIATE(double)
#endif
}
