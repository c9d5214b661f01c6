// fc_kernels.h -- host-callable launchers of the wavefront kernels (defined in fc_kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>

#include "../../include/fcgpu.h"
#include "../../include/fcgpu_sampler.h"
#include "fc_host.h"
#include "fc_types.h"

namespace fcg
{
    struct Batch
    {
        int32_t x0, y0, tw;          // tile origin and width
        uint32_t pix0, npix;         // pixel-in-tile range of this batch
        FastDiv div_npix, div_tw;    // slot -> (sample, pixel) and pixel -> (row, column)
        // band sharding (fcgpu_render_bands): row r of the pass is film row y0 + r + (r / band_rows) * band_skip, i.e. the
        // bands of band_rows rows that this GPU owns, packed; band_skip = 0 (and div_band = 1) for a plain tile
        FastDiv div_band;
        uint32_t band_skip;
        uint32_t s0, nsamp;          // sample range of this batch; slots = npix * nsamp
        fcgpu_sampler_config sampler;  // include/fcgpu_sampler.h: seed + mode (random / stratified)
        int32_t h;                   // hit index being produced / shaded: 1 = camera ray's hit
        int32_t max_len;             // max_path_length
        int32_t integrator;          // FCGPU_INTEGRATOR_*
        int32_t visible_env;
        int32_t parity;              // which ray counter extend reads (shade writes parity ^ 1)
        int32_t W;                   // film width
        int32_t finish_min;          // traversal tuning (see fc_traverse.h / k_extend)
        int32_t descend_min;
        int32_t finish_min_shadow, descend_min_shadow;  // the same two for k_shadow
        double* film;
    };

    struct LaunchCfg
    {
        hipStream_t stream;
        int grid;       // persistent grid size (blocks of 256)
        int lds_pad_extend = 0, lds_pad_shadow = 0;  // extra dynamic LDS per block: caps the blocks per CU (tuning)
        int stream_grid = 0;  // grid of the grid-stride streaming kernels (generate, shade, miss, accumulate); 0 = `grid`
        int streaming() const { return stream_grid > 0 ? stream_grid : grid; }
    };

    // fc_lbvh.hip: binary hierarchy over primitive bounds ([n][6]) built on the device (Morton sort + Karras tree + bounds)
    int lbvh_build(hipStream_t stream, const std::vector<float>& boxes, LbvhTree& out, std::string& err);

    template<typename R> void launch_generate(const LaunchCfg& L, const SceneDev<R>& S, const Wavefront<R>& Q, const Batch& B);
    template<typename R> void launch_extend(const LaunchCfg& L, const SceneDev<R>& S, const Wavefront<R>& Q, const Batch& B);
    template<typename R> void launch_sort(const LaunchCfg& L, const Wavefront<R>& Q, const Batch& B);
    // key in [0, L_COUNT): lobe shade; Q_FINAL: last vertex; Q_MISS: environment term
    template<typename R> void launch_shade(const LaunchCfg& L, const SceneDev<R>& S, const Wavefront<R>& Q, const Batch& B, int key);
    template<typename R> void launch_shadow(const LaunchCfg& L, const SceneDev<R>& S, const Wavefront<R>& Q, const Batch& B);
    template<typename R> void launch_accumulate(const LaunchCfg& L, const SceneDev<R>& S, const Wavefront<R>& Q, const Batch& B);
    void launch_reset_counters(const LaunchCfg& L, QueueCounters* c, int parity_to_keep);

    // dst[i] += src[i] over the film of the root context (fcgpu_reduce_film)
    void launch_film_add(const LaunchCfg& L, double* dst, const double* src, uint64_t n);
    void launch_film_add_bands(const LaunchCfg& L, double* dst, const double* src, uint64_t row_elems, uint64_t owned_rows, uint32_t rows,
                               uint32_t first, uint32_t stride);
    // frame = ng, tangent, ns, bitangent (3 doubles each); device pointers (fcgpu_bsdf_table)
    template<typename R>
    void launch_bsdf_table(const LaunchCfg& L, const LobeRec<R>* lobes, int lobe_count, const double* frame, double eta_a, double eta_b, uint64_t n,
                           const fcgpu_bsdf_query* queries, fcgpu_bsdf_result* results);

    // ---- fcgpu_trace_closest_noalloc: the variant of fcgpu_trace_closest_device for callers that cannot allocate.
    // Its scratch (the traversal kernel's statistics words and, for deep hierarchies, the spill columns of the traversal
    // stack) is a file-scope __device__ array in fc_kernels.hip instead of hipMalloc'ed memory: 256 bytes of counters +
    // NOALLOC_SPILL_ENTRIES x NOALLOC_LANES words = 8 MiB + 256 B per GPU, held until the process ends.
    constexpr uint32_t NOALLOC_LANES = 256u * 256u;       // the variant's grid: 256 blocks of 256 lanes
    constexpr uint32_t NOALLOC_SPILL_ENTRIES = 32u;       // stack entries beyond STACK_CAP per lane
    constexpr size_t NOALLOC_STATS_BYTES = 256;
    constexpr size_t NOALLOC_SCRATCH_BYTES = NOALLOC_STATS_BYTES + sizeof(uint32_t) * NOALLOC_SPILL_ENTRIES * NOALLOC_LANES;
    // device address of this device's copy of the scratch array (hipGetSymbolAddress, once per device; the device must
    // be current). Returns hipSuccess or the HIP error.
    hipError_t noalloc_scratch_for_device(int device, unsigned char** out);
    // adds the statistics words a no-alloc launch left in the scratch to the context's counters
    void launch_fold_stats(const LaunchCfg& L, unsigned long long* dst, const unsigned long long* src);

    template<typename R> void launch_trace(const LaunchCfg& L, const SceneDev<R>& S, uint64_t n, const fcgpu_ray* rays,
                                           fcgpu_hit* hits, uint8_t* occluded, unsigned long long* stats);
}
