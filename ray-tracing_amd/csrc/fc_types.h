// fc_types.h -- flattened, device-resident scene + wavefront queue layouts (host + device).
//
// HBM layout:
//   nodes      Node4[ ]     64-B 4-wide BVH nodes, child boxes quantised to 8 bit on a
//                           per-node power-of-two grid, SoA inside the node (x/y/z planes as
//                           the four bytes of one dword -> v_cvt_f32_ubyte0..3 decode).
//   tris       TriRec[ ]    48-B pre-gathered triangles in BVH leaf order (float vertices, as
//                           mesh_surface stores them: mesh_surface.hpp:306)
//   xprims     XPrim<R>[ ]  spheres / planes with their object<->world transforms
//   queues     SoA columns of Vec4<R>; path state AoS (one cache line per path in fp32)
#pragma once
#include <cstdint>

#include "fc_math.h"

namespace fcg
{
    constexpr int TIMELINE_WORDS = 24;  // per wave, FCGPU_TIMELINE development aid (k_extend's PROF instantiation)

    constexpr uint32_t INVALID_U32 = 0xFFFFFFFFu;

    // Experiment: the first FCG_TREELET wide nodes (breadth-first top of the hierarchy, fc_host.cpp: treelet_first) are
    // copied into LDS by every block of a traversal kernel and fetched from there (0 = off, the default)
#ifndef FCG_TREELET
#define FCG_TREELET 0
#endif

    // ---- BVH
    constexpr uint32_t NODE_LEAF_BIT = 0x80000000u;
#ifndef FCG_MAX_LEAF_PRIMS
#define FCG_MAX_LEAF_PRIMS 2
#endif
    // triangles per leaf run (spheres / planes: one per leaf). Measured on the room scene: 4 -> 362, 3 -> 373, 2 -> 373,
    // 1 -> 336 Msamples/s: a test is 3 record loads + ~70 instructions for the few lanes that hold a leaf
    constexpr int MAX_LEAF_PRIMS = FCG_MAX_LEAF_PRIMS;
#ifndef FCG_STACK_CAP
#define FCG_STACK_CAP 30
#endif
    constexpr int STACK_CAP = FCG_STACK_CAP;
    // The fp32 4-wide shadow kernel keeps a shorter column in LDS and fits 72 registers: 22 entries x 256 lanes x 4 B = 22 KB per
    // block, seven blocks per CU = seven waves per SIMD instead of five (k_shadow -4 %). No ray of the bench scene goes deeper
    // than 7 entries; hierarchies whose worst case exceeds SHADOW_STACK_CAP - 2 run its SPILL instantiation, whose entries
    // beyond the LDS part live in the same global columns as k_extend's.
#ifndef FCG_SHADOW_STACK_CAP
#define FCG_SHADOW_STACK_CAP 22
#endif
    constexpr int SHADOW_STACK_CAP = FCG_SHADOW_STACK_CAP;
    static_assert(SHADOW_STACK_CAP <= STACK_CAP, "the spill columns are sized for the shorter of the two LDS stacks");
    // entries beyond STACK_CAP spill to a per-lane column in global memory (allocated only for scenes
    // whose worst case needs them): deep trees work, shallow ones pay nothing
    constexpr int STACK_SPILL_MAX = 226;  // LDS traversal stack entries per lane: 30 KB per 256-thread block, 5 blocks per CU

    struct alignas(16) Node4
    {
        float ox, oy, oz;       // grid origin (= node box minimum)
        float sx;               // grid step along x: a power of two, stored as the float the node step multiplies with
        uint32_t qlo[3];        // per axis: byte c = quantised lower bound of child c
        uint32_t qhi[3];        // per axis: byte c = quantised upper bound of child c
        float sy, sz;           // grid steps along y and z (kept as floats, not exponent bytes: six instructions fewer per node step)
        uint32_t child[4];      // interior: node index; leaf: NODE_LEAF_BIT | kind<<29 | (count-1)<<26 | first; empty: INVALID_U32
    };
    static_assert(sizeof(Node4) == 64, "Node4 must be one 64-byte line");

    // The 8-wide node (fcgpu_options.bvh_width = 8; Ylitie, Karras, Laine 2017 in spirit): 80 bytes = five 16-byte pieces.
    // Children sit in SLOTS chosen by the builder from where they lie in the node (slot bit 0 / 1 / 2 = the +x / +y / +z
    // side), so that a ray visits the hit children in order of (slot XOR the ray's octant) without sorting distances;
    // interior children are consecutive nodes (in slot order) and the triangles of the leaf children consecutive records
    // (in slot order), so one index each and a few mask bits stand for all eight children. A sphere or plane is a leaf
    // of one PROXY record in the triangle array (TriRec::attr = XPRIM_PROXY, ::prim = index into xprims, ::entity = kind).
    struct alignas(16) Node8
    {
        float ox, oy, oz;     // grid origin (= node box minimum)
        uint32_t exps;        // bytes 0..2: biased exponents of the grid steps along x, y, z (step = 2^(e - 127)); byte 3: imask, bit s = slot s is an interior node
        uint32_t child_base;  // bits 0..23: first interior child; bits 24..31: valid mask, bit s = slot s holds a child
        uint32_t tri_base;    // first record of the leaf children's triangles
        uint32_t two_mask;    // bits 0..7: leaf slot s holds two records (else one); rest 0
        uint32_t pad;
        uint32_t qlo[3][2];   // [axis][half]: byte c of half h = quantised lower bound of slot 4 h + c
        uint32_t qhi[3][2];
    };
    static_assert(sizeof(Node8) == 80, "Node8 is five 16-byte pieces");
    constexpr uint32_t XPRIM_PROXY = 0xFFFFFFFEu;
    constexpr uint32_t NODE8_MAX_NODES = 1u << 24;
    constexpr int STACK8_CAP = STACK_CAP / 2;  // group entries of the 8-wide traversal: two words each in the same LDS array

    FC_HD bool child_is_leaf(uint32_t c) { return (c & NODE_LEAF_BIT) != 0; }
    // leaf word: first = index into tris (KIND_TRI, `count` consecutive records) or xprims (count 1)
    FC_HD uint32_t leaf_first(uint32_t c) { return c & 0x03FFFFFFu; }
    FC_HD uint32_t leaf_count(uint32_t c) { return ((c >> 26) & 0x7u) + 1u; }
    FC_HD uint32_t leaf_kind(uint32_t c) { return (c >> 29) & 0x3u; }
    FC_HD uint32_t make_leaf(uint32_t kind, uint32_t first, uint32_t count) { return NODE_LEAF_BIT | (kind << 29) | ((count - 1u) << 26) | first; }

    constexpr uint32_t KIND_TRI = 0, KIND_SPHERE = 1, KIND_PLANE = 2;

    struct alignas(16) TriRec
    {
        float v0[3]; uint32_t prim;    // global primitive id
        float v1[3]; uint32_t entity;  // entity index (ENTITY_MASK) | key hint << 24 (KEY_HINT_NONE = look the material up)
        float v2[3]; uint32_t attr;    // index of vertex attributes (normals/uvs) or INVALID_U32
    };
    static_assert(sizeof(TriRec) == 48, "TriRec is 3 x 16 B");
    constexpr uint32_t ENTITY_MASK = 0x00FFFFFFu, KEY_HINT_NONE = 0xFFu;

    // What shading needs of a triangle, by triangle index, computed once at upload in the arithmetic of the kernels that
    // read it: geometric normal and tangent (mesh_surface.hpp:212-213, 221-248 are constants of the triangle). 32 bytes in
    // fp32, so a record never straddles a 128-byte line (a 48-byte TriRec does 3 times in 8) and k_shade neither fetches
    // the vertices a second time nor normalises twice per hit.
    template<typename R>
    struct alignas(16) TriShade
    {
        R ng[3];
        uint32_t entity;  // entity index (no key hint)
        R t[3];           // normalize(dpdu)
        uint32_t attr;    // as TriRec::attr
    };
    static_assert(sizeof(TriShade<float>) == 32 && sizeof(TriShade<double>) == 64, "TriShade is two 16-byte (32-byte) pieces");

    struct TriAttr  // per-triangle vertex attributes, only for meshes that carry them
    {
        float n0[3], n1[3], n2[3];  // world-space vertex normals (mesh_surface.hpp:30-38)
        float uv0[2], uv1[2], uv2[2];
        float pad;
    };
    static_assert(sizeof(TriAttr) == 64, "TriAttr is one line");

    template<typename R>
    struct alignas(16) XPrim  // sphere or plane
    {
        R inv[12];   // world -> object (pr_transform::inv_t_)
        R fwd[12];   // object -> world
        R a, b;      // sphere: a = radius; plane: a = size.x, b = size.y
        uint32_t prim, entity;
        R inv_area;
        R pad;
    };

    struct PrimInfo  // by global primitive id
    {
        uint32_t kind_index;  // kind(2) | index(30) into tris / xprims
        uint32_t entity;
    };

    struct EntityRec
    {
        int32_t material;
        int32_t medium;   // 0 = none, else medium id (1-based; medium 0 is the implicit vacuum)
        int32_t light;    // -1 or light index
        uint32_t flags;   // bit0 has_normals, bit1 has_uvs
    };

    // ---- shading
    enum LobeType : int32_t
    {
        L_LAMBERT = 0,     // lambertian_reflection.hpp
        L_GGX_REFL = 1,    // microfacet_reflection.hpp
        L_SPEC_REFL = 2,   // specular_reflection.hpp
        L_GGX_GLASS = 3,   // microfacet_glass.hpp
        L_SPEC_GLASS = 4,  // specular_glass.hpp
        L_GGX_TRANS = 5,   // microfacet_transmission.hpp
        L_SPEC_TRANS = 6,  // specular_transmission.hpp
        L_COUNT = 7
    };
    // wavefront queue keys: one hit queue per lobe type + the last-vertex queue + the miss queue
    constexpr int Q_FINAL = L_COUNT;
    constexpr int Q_MISS = L_COUNT + 1;
    constexpr int Q_COUNT = L_COUNT + 2;

    FC_HD bool lobe_is_delta(int t) { return t == L_SPEC_REFL || t == L_SPEC_GLASS || t == L_SPEC_TRANS; }

    template<typename R>
    struct LobeRec
    {
        int32_t type;
        int32_t fresnel;   // 0 = fresnel_one, 1 = fresnel_dielectric (microfacet_reflection.hpp:18-35)
        R color[3];        // reflectance (or transmittance for the *_TRANS lobes)
        R color2[3];       // transmittance of the glass lobes
        R alpha[2];        // max(roughness, 0.002)^2, microfacet.hpp:76-81
        R ior;             // material ior used by the reflection lobes' Fresnel
        R factor;          // bsdf scale / lobe pick pdf (bsdf.hpp:70,96)
        double cdf;        // exclusive prefix of the pick pdfs (bsdf.hpp:39-43), compared in double
    };

    struct MaterialRec
    {
        int32_t first_lobe;   // constant materials: their lobes are precomputed (lobes[first_lobe ..])
        int32_t lobe_count;
        int32_t kind;         // FCGPU_MATERIAL_*
        int32_t textured;     // 1 = some parameter is an image texture: lobes are built per hit (fc_material.h)
        int32_t tex[6];       // TEX_* slots: 0 = constant (no normal map), k > 0 = textures[k - 1]
    };
    enum TexSlot { TEX_COLOR_A = 0, TEX_COLOR_B = 1, TEX_ROUGHNESS = 2, TEX_METALNESS = 3, TEX_IOR = 4, TEX_NORMAL = 5 };

    template<typename R>
    struct MatParam  // the constants of fcgpu_material, by material
    {
        R color_a[3], color_b[3], roughness[2], ior, metalness;
    };

    template<typename R> struct Vec4;
    template<typename R>
    struct TexRef  // image_texture_2d_*: decoded texels of image::rgb() + reconstruction filter
    {
        const Vec4<R>* texels;  // row-major, top row first
        int32_t width, height;
        int32_t filter;         // FCGPU_FILTER_*
        int32_t kind;           // FCGPU_TEXTURE_*: image (above) or checker_texture_2d_rgb (below)
        R checker_a[3], checker_b[3], checker_scale;
    };

    constexpr int MAX_MEDIA = 15;       // 4-bit medium ids, 0 = vacuum
    constexpr int MEDIUM_STACK_CAP = 9; // nested media beyond the vacuum: the reference's capacity (integrator.hpp:24: buffer_capacity = 10)
    constexpr int MEDIUM_STACK_INLINE = 5;  // of those, the entries that travel in the ray / hit records' own word (MStack)

    template<typename R>
    struct MediumRec
    {
        int32_t priority;
        int32_t pad;
        R ior;
        R sigma[3];  // colour * density (medium.hpp:49)
    };

    template<typename R>
    struct LightRec
    {
        int32_t entity;
        uint32_t kind;        // KIND_TRI (mesh) / KIND_SPHERE / KIND_PLANE
        uint32_t index;       // xprims index (sphere / plane)
        uint32_t first_prim;  // mesh: global id of its first triangle
        uint32_t tri_count;
        uint32_t cdf_offset;  // mesh: offset into light_cdf (tri_count + 1 doubles)
        R Le[3];
        R inv_area;           // pdf_p = 1 / area (mesh_surface.hpp:292, sphere :170, plane :129)
    };

    // texture_infinity_area_light (texture_infinity_area_light.hpp:15-131): the image texture
    // (decoded texels, image_texture.hpp:88-124) and its radiance distribution (distribution.hpp:
    // 23-166; kept in double like the light CDFs so that discrete picks equal the reference's).
    template<typename R>
    struct EnvTex
    {
        TexRef<R> tex;
        int32_t nx, ny;           // distribution resolution
        R strength;
        R fwd[12], inv[12];       // the light's pr_transform (rotation only) and its inverse
        const double* func_x;     // [ny][nx]   row functions (luminance(integral) * sin theta), after distribution_1d's zero-row rule
        const double* cdf_x;      // [ny][nx+1]
        const double* int_x;      // [ny]       function integrals of the rows
        const double* func_y;     // [ny]
        const double* cdf_y;      // [ny+1]
        double int_y;
    };

    template<typename R>
    struct CameraRec
    {
        R fwd[12];
        R plane_w, plane_h;   // sample_plane_size_ (perspective_camera.hpp:24-26)
        R focus, lens_radius;
        int32_t W, H;
    };

    template<typename R>
    struct SceneDev
    {
        const Node4* nodes;
        const Node8* nodes8;  // non-null: the scene was flattened 8-wide (bvh_width 8) and the kernels run their WIDE8 instantiations
        const TriRec* tris;
        const TriAttr* tri_attrs;
        const XPrim<R>* xprims;
        const PrimInfo* prim_info;
        const EntityRec* entities;
        const MaterialRec* materials;
        const LobeRec<R>* lobes;
        const MatParam<R>* matparams;
        const TexRef<R>* textures;
        const MediumRec<R>* media;
        const LightRec<R>* lights;
        const double* light_cdf;
        const TriShade<R>* tri_shade;  // by triangle index
        uint32_t* stack_spill;     // [depth - LDS entries of the kernel][grid lanes] or nullptr when the BVH fits every kernel's LDS stack
        uint32_t stack_spill_stride;
        uint32_t node_count;       // wide nodes
        uint32_t stack_spill_cap;  // entries per lane column of stack_spill: pushes beyond a kernel's LDS entries + this are counted as lost
        uint32_t max_stack;        // worst-case traversal stack of the hierarchy: selects the SPILL instantiations per kernel
        uint32_t textured_materials;  // 1 = some material has image textures (selects the TEX kernel instantiations)
        CameraRec<R> camera;
        uint32_t light_count;
        int32_t env_kind;
        R env[3];
        R epsilon;
        uint32_t offset_along_ray;  // 1: origins leave their surface along the ray (fp32 default), 0: along the normal like the reference (lift_origin)
        EnvTex<R> envtex;     // env_kind == FCGPU_ENV_TEXTURE
    };

    // ---- wavefront queues
    template<typename R>
    struct alignas(16) Vec4
    {
        R x, y, z, w;
    };

    // two 32-bit ids travel in the tail of an R column: two float slots or one double slot
    FC_HD void pack_ids(Vec4<float>& v, uint32_t a, uint32_t b)
    {
        v.z = __builtin_bit_cast(float, a);
        v.w = __builtin_bit_cast(float, b);
    }
    FC_HD void unpack_ids(const Vec4<float>& v, uint32_t& a, uint32_t& b)
    {
        a = __builtin_bit_cast(uint32_t, v.z);
        b = __builtin_bit_cast(uint32_t, v.w);
    }
    FC_HD void pack_ids(Vec4<double>& v, uint32_t a, uint32_t b)
    {
        v.z = __builtin_bit_cast(double, (static_cast<uint64_t>(a) << 32) | b);
        v.w = 0.0;
    }
    FC_HD void unpack_ids(const Vec4<double>& v, uint32_t& a, uint32_t& b)
    {
        uint64_t u = __builtin_bit_cast(uint64_t, v.z);
        a = static_cast<uint32_t>(u >> 32);
        b = static_cast<uint32_t>(u);
    }

    // Division of any 32-bit n by a launch-invariant d (1 <= d < 2^31) with a multiply-high instead of
    // the ~35-instruction software divide: q = floor(n * M / 2^(32+s)), M = ceil(2^(32+s) / d),
    // s = ceil(log2 d) (Granlund & Montgomery 1994); mul holds M - 2^32.
    struct FastDiv
    {
        uint32_t d, mul, sh1;  // sh1 = s - 1 (d > 1)
    };
    inline FastDiv make_fastdiv(uint32_t d)
    {
        FastDiv f{d, 0u, 0u};
        if(d <= 1u) return f;
        uint32_t s = 0;
        while((1ull << s) < d) ++s;
        const uint64_t m = ((1ull << (32 + s)) + d - 1) / d;
        f.mul = static_cast<uint32_t>(m - (1ull << 32));
        f.sh1 = s - 1u;
        return f;
    }
    FC_HD uint32_t fast_div(uint32_t n, const FastDiv& f)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        const uint32_t t = __umulhi(n, f.mul);
#else
        const uint32_t t = static_cast<uint32_t>((static_cast<uint64_t>(n) * f.mul) >> 32);
#endif
        const uint32_t q = (t + ((n - t) >> 1)) >> f.sh1;
        return f.d <= 1u ? n : q;
    }

    // hit-queue word: lobe index within the material (2) | primitive kind (2) | index into tris / xprims (28)
    FC_HD uint32_t hit_word(uint32_t pk, uint32_t lobe) { return (lobe << 30) | ((pk >> 30) << 28) | (pk & 0x0FFFFFFFu); }
    FC_HD uint32_t hit_word_lobe(uint32_t w) { return w >> 30; }
    FC_HD uint32_t hit_word_kind(uint32_t w) { return (w >> 28) & 3u; }
    FC_HD uint32_t hit_word_index(uint32_t w) { return w & 0x0FFFFFFFu; }

    // one 32-bit word in a single R slot
    FC_HD float pack_word(float, uint32_t a) { return __builtin_bit_cast(float, a); }
    FC_HD double pack_word(double, uint32_t a) { return __builtin_bit_cast(double, static_cast<uint64_t>(a)); }
    FC_HD uint32_t unpack_word(float v) { return __builtin_bit_cast(uint32_t, v); }
    FC_HD uint32_t unpack_word(double v) { return static_cast<uint32_t>(__builtin_bit_cast(uint64_t, v)); }

    // Priority-medium stack of a path (helper::buffer_, integrator.hpp:24-31: the vacuum + at most 9 media).
    //   w   = count bits 0..2 | 5 x id(4) | above(4) | below(4) | count bit 3: the word of the ray and hit records
    //   ext = ids 5..8 (4 x 4 bits): only paths nested deeper than five media ever touch it. It travels in the fourth
    //         word of the path-state column st_c2 (same index as the ray / hit record), which k_extend and k_shade read
    //         only when the count in `w` says so; everybody else moves one word as before.
    struct MStack
    {
        uint32_t w, ext;
    };
    FC_HD uint32_t ms_count(uint32_t w) { return (w & 7u) | ((w >> 31) << 3); }
    FC_HD bool ms_deep(uint32_t w) { return ms_count(w) > static_cast<uint32_t>(MEDIUM_STACK_INLINE); }
    FC_HD uint32_t ms_with_count(uint32_t w, uint32_t n) { return (w & 0x7FFFFFF8u) | (n & 7u) | ((n >> 3) << 31); }
    FC_HD uint32_t ms_get(const MStack& m, uint32_t i)
    {
        return i < static_cast<uint32_t>(MEDIUM_STACK_INLINE) ? (m.w >> (3 + 4 * i)) & 15u : (m.ext >> (4 * (i - MEDIUM_STACK_INLINE))) & 15u;
    }
    FC_HD void ms_set(MStack& m, uint32_t i, uint32_t id)
    {
        if(i < static_cast<uint32_t>(MEDIUM_STACK_INLINE))
            m.w = (m.w & ~(15u << (3 + 4 * i))) | (id << (3 + 4 * i));
        else
            m.ext = (m.ext & ~(15u << (4 * (i - MEDIUM_STACK_INLINE)))) | (id << (4 * (i - MEDIUM_STACK_INLINE)));
    }
    FC_HD uint32_t ms_above(uint32_t w) { return (w >> 23) & 15u; }
    FC_HD uint32_t ms_below(uint32_t w) { return (w >> 27) & 15u; }
    FC_HD uint32_t ms_with_media(uint32_t w, uint32_t above, uint32_t below)
    {
        return (w & 0x807FFFFFu) | (above << 23) | (below << 27);
    }
    FC_HD uint32_t ms_strip_media(uint32_t w) { return w & 0x807FFFFFu; }  // what travels with a ray: the stack without above / below

    // per-path persistent state (the record the Wavefront's st_c0..2 columns hold, 16 bytes of it each): 48 B in fp32, 96 B in fp64. The radiance a path has gathered is NOT part of it: it
    // lives in the Wavefront's `li_shadow` column, which only the hits that add to it touch (emitter and environment
    // hits, unoccluded shadow rays), so a bounce moves 2 x 48 instead of 2 x 64 bytes of state.
    template<typename R>
    struct alignas(16) PathState
    {
        R beta[3];
        R pdf;        // pdf_w12 of the last BSDF sample (forward_mis_integrator.hpp:104)
        R p1[3];      // previous vertex (un-offset), for transmittance and the area-measure pdf
        uint32_t flags;  // bit0: last lobe was delta (or camera segment): emitter hits un-weighted (in fp64 the compiler pads to 8)
        R w[3];       // direction of the segment that produced the current hit (w12)
        R ms_ext;     // MStack::ext as a packed word (0 unless the path is nested deeper than five media)
    };

    static_assert(sizeof(PathState<float>) == 48 && sizeof(PathState<double>) == 96, "one path state = three 16-byte (32-byte) pieces");

    constexpr uint32_t DEC_MISS = 0xFFFFFFFEu, DEC_DEAD = 0xFFFFFFFDu;
    constexpr uint32_t DEC_SHADED = 1u, DEC_SAMPLED = 2u, DEC_TRANSMITTED = 4u, DEC_SHADOW_CAST = 8u, DEC_OCCLUDED = 16u, DEC_KEY_SHIFT = 8u;

    constexpr int WORK_RANGES = 8;

    struct QueueCounters  // device-resident, zeroed per bounce
    {
        // every append counter on its own 128-byte line: atomics on one line are served one after the other
        uint32_t hit[Q_COUNT][32];
        uint32_t ray[2][32];     // [parity][0]: ping-pong: extension rays of this / the next bounce
        uint32_t shadow[32];     // [0]
        // dynamic work cursors of the two traversal kernels: one per ray-queue range, each on its own
        // 128-byte line (a single word serves only ~88 atomics/us)
        uint32_t work_extend[WORK_RANGES][32];
        uint32_t work_shadow[WORK_RANGES][32];
    };

    enum StatIndex
    {
        ST_CLOSEST = 0, ST_SHADOW = 1, ST_OCCLUDED = 2, ST_NAN = 3, ST_STACK_OVERFLOW = 4, ST_MEDIUM_OVERFLOW = 5, ST_NODES = 6, ST_PRIMS = 7, ST_STACK_SPILL = 8, ST_MEDIUM_DEEP = 9,
#ifdef FCG_SHADE_CLOCKS  // development build (tools/build_variant.sh): wave cycles per phase of a k_shade<Lambert> trip, [8] = trips
        ST_SHADE_PH0 = 10, ST_COUNT = 20
#else
        ST_COUNT = 10
#endif
    };

    template<typename R>
    struct Wavefront
    {
        // extension rays: c0 = {o.xyz, d.x}, c1 = {d.y, d.z, ids(path, mstack)}
        Vec4<R>* ray_c0;
        Vec4<R>* ray_c1;
        // hits, in ray-queue order: c0 = {p.xyz, b0}, c1 = {b1, word(mstack), ids(prim|lobe<<30, path)};
        // hit_key[i] = queue key of hit i (0xFF = path ended); hit_idx[k] = indices of the hits with key k
        Vec4<R>* hit_c0;
        Vec4<R>* hit_c1;
        uint8_t* hit_key;
        uint32_t* hit_idx[Q_COUNT];
        // shadow rays: c0 = {o.xyz, d.x}, c1 = {d.y, d.z, tmax, path}, c2 = {contribution rgb, -}
        Vec4<R>* sh_c0;
        Vec4<R>* sh_c1;
        Vec4<R>* sh_c2;
        // path state in RAY-QUEUE order, one set of columns per ray-queue parity: k_shade reads set [parity] at the index of
        // the hit it shades (= the index of the ray that found it) and writes set [parity ^ 1] at the slot of the ray it
        // appends, so both are streams next to the hit / ray records instead of 48-byte gathers and scatters by path id
        // (measured: k_shade<Lambert> went from 6.6 ms at the first bounce to 9.9 ms at the ninth as path ids in the queue
        // lost their order). c0 = {beta.xyz, pdf}, c1 = {p1.xyz, flags}, c2 = {w.xyz, -}: the fields of PathState.
        Vec4<R>* st_c0[2];
        Vec4<R>* st_c1[2];
        Vec4<R>* st_c2[2];
        // light-strategy radiance by path id, kept apart from the path state: the shadow kernel is its
        // only writer (16 B read-modify-write per visible ray instead of a 64-B state line) and
        // accumulate adds it to state.Li
        Vec4<R>* li_shadow;
        QueueCounters* counters;
        unsigned long long* stats;
        // development aid (FCGPU_TIMELINE=file): per wave of the last k_extend launch {start, end (100 MHz
        // ticks), rays, rounds, node steps sum / lane max, primitive tests sum / lane max}; nullptr in normal operation
        unsigned long long* timeline;  // FCGPU_TIMELINE: TIMELINE_WORDS words per wave of the grid
        // parity hook (fcgpu_set_decision_log): per bounce and path, the discrete decisions of the path, or nullptr.
        // dec_hit[(h - 1) * dec_stride + path] = hit word (lobe | kind | record index) of hit h, DEC_MISS or DEC_DEAD;
        // dec_flags[...] = DEC_* bits of what k_shade / k_shadow decided at that hit
        uint32_t* dec_hit;
        uint32_t* dec_flags;
        uint32_t dec_stride;
        uint32_t capacity;
    };

#if defined(__HIPCC__)
    // Queue records are written once and read once, gigabytes apart: streaming ("nt") accesses keep them from evicting
    // the BVH nodes and triangle records that every wave re-reads out of L1 / L2 (measured: k_extend -8.8 %). Only for
    // records a wave touches as whole contiguous columns (ray, hit and shadow queues): the 64-byte path state is read
    // as four 16-byte pieces of one line per lane, and without L1 allocation each piece becomes its own L2 request
    // (k_shade +31 % when tried), and nt stores of it lose write combining (k_generate x4). FCG_STREAM_HINTS=0 builds without.
#ifndef FCG_STREAM_HINTS
#define FCG_STREAM_HINTS 1
#endif
    typedef unsigned int stream_u32x4 __attribute__((ext_vector_type(4)));
    template<typename T> FC_D T ld_stream(const T* p)
    {
        static_assert(sizeof(T) % 16 == 0, "whole 16-byte pieces");
#if FCG_STREAM_HINTS
        stream_u32x4 piece[sizeof(T) / 16];
        const stream_u32x4* src = reinterpret_cast<const stream_u32x4*>(p);
#pragma unroll
        for(size_t i = 0; i < sizeof(T) / 16; ++i) piece[i] = __builtin_nontemporal_load(src + i);
        T out;
        __builtin_memcpy(&out, piece, sizeof(T));
        return out;
#else
        return *p;
#endif
    }
    // x, y, z of a Vec4 column whose w is unused (w reads as 0): for a load that stays in flight across other work, so that
    // no register of it is dead on arrival (the compiler hands a dead destination register to the next instruction that
    // needs one, and that instruction then waits for the whole load)
    template<typename R> struct stream_xyz { typedef R type __attribute__((ext_vector_type(3))); };
    template<typename R> FC_D Vec4<R> ld_stream_xyz(const Vec4<R>* p)
    {
        typedef typename stream_xyz<R>::type V;
#if FCG_STREAM_HINTS
        const V v = __builtin_nontemporal_load(reinterpret_cast<const V*>(p));
#else
        const V v = *reinterpret_cast<const V*>(p);
#endif
        return Vec4<R>{v.x, v.y, v.z, R(0)};
    }
    template<typename T> FC_D void st_stream(T* p, const T& v)
    {
        static_assert(sizeof(T) % 16 == 0, "whole 16-byte pieces");
#if FCG_STREAM_HINTS
        stream_u32x4 piece[sizeof(T) / 16];
        __builtin_memcpy(piece, &v, sizeof(T));
        stream_u32x4* dst = reinterpret_cast<stream_u32x4*>(p);
#pragma unroll
        for(size_t i = 0; i < sizeof(T) / 16; ++i) __builtin_nontemporal_store(piece[i], dst + i);
#else
        *p = v;
#endif
    }
#endif
}
