// fc_traverse.h -- device-side primitive intersectors and 4-wide BVH traversal (generic over R).
//
// Replaces, on the device:
//   bvh_acceleration_structure::raycast_surface_point / raycast   bvh_acceleration_structure.hpp:31-91, 93-138
//   TBounds3::Raycast (slab test)                                 src/core/math.hpp:671-689
//   mesh_surface::raycast[_surface_point] (watertight triangle)   src/surfaces/mesh_surface.hpp:78-143, 145-209
//   sphere_surface::raycast[_surface_point]                       src/surfaces/sphere_surface.hpp:41-83, 85-121
//   plane_surface::raycast[_surface_point]                        src/surfaces/plane_surface.hpp:40-64, 66-84
// One ray per lane; the per-lane traversal stack lives in LDS as stack[depth][lane] (lane-major
// rows: conflict-free ds_read/ds_write_b32).
#pragma once
#include "fc_types.h"

namespace fcg
{
    template<typename R>
    struct RayCtx
    {
        vec3<R> o, d, inv_d;
        bool negx, negy, negz;  // sign of each direction component: which slab plane is the near one
        int kx, ky, kz;  // mesh_surface.hpp:89-93
        R sx, sy, sz;    // mesh_surface.hpp:100-102
    };

    template<typename R> struct huge_inv;
    template<> struct huge_inv<float> { static constexpr float v = 1e30f; };
    template<> struct huge_inv<double> { static constexpr double v = 1e300; };

    template<typename R> FC_D R safe_inv(R d)
    {
        R inv = R(1) / d;
        // axis-parallel rays: +-inf would turn (plane - origin) * inv into NaN when the origin lies on the plane
        R h = huge_inv<R>::v;
        return inv > h ? h : (inv < -h ? -h : inv);
    }

    template<typename R> FC_D RayCtx<R> make_ray(vec3<R> o, vec3<R> d)
    {
        RayCtx<R> rc;
        rc.o = o;
        rc.d = d;
        rc.inv_d = vec3<R>{safe_inv(d.x), safe_inv(d.y), safe_inv(d.z)};
        rc.negx = rc.inv_d.x < R(0);
        rc.negy = rc.inv_d.y < R(0);
        rc.negz = rc.inv_d.z < R(0);
        R ax = r_abs(d.x), ay = r_abs(d.y), az = r_abs(d.z);
        int kz = (ax > ay && ax > az) ? 0 : (ay > az ? 1 : 2);  // max_dimension, math.hpp:242-256
        int kx = kz + 1;
        if(kx == 3) kx = 0;
        int ky = kx + 1;
        if(ky == 3) ky = 0;
        rc.kx = kx;
        rc.ky = ky;
        rc.kz = kz;
        R dz = comp(d, kz);
        rc.sx = -comp(d, kx) / dz;
        rc.sy = -comp(d, ky) / dz;
        rc.sz = R(1) / dz;
        return rc;
    }

    template<typename R> FC_D vec3<R> ldv(const float* p) { return vec3<R>{R(p[0]), R(p[1]), R(p[2])}; }

    // watertight triangle test; accepts t == t_max like the reference (t_scaled > t_max*det rejects)
    template<typename R>
    FC_D bool tri_intersect(const TriRec& T, const RayCtx<R>& rc, R t_max, R& t_hit, R& b0, R& b1)
    {
#pragma clang fp contract(off)
        vec3<R> p0 = ldv<R>(T.v0) - rc.o, p1 = ldv<R>(T.v1) - rc.o, p2 = ldv<R>(T.v2) - rc.o;
        R p0x = comp(p0, rc.kx), p0y = comp(p0, rc.ky), p0z = comp(p0, rc.kz);
        R p1x = comp(p1, rc.kx), p1y = comp(p1, rc.ky), p1z = comp(p1, rc.kz);
        R p2x = comp(p2, rc.kx), p2y = comp(p2, rc.ky), p2z = comp(p2, rc.kz);
        p0x += rc.sx * p0z; p0y += rc.sy * p0z;
        p1x += rc.sx * p1z; p1y += rc.sy * p1z;
        p2x += rc.sx * p2z; p2y += rc.sy * p2z;
        R e0 = p1x * p2y - p1y * p2x;
        R e1 = p2x * p0y - p2y * p0x;
        R e2 = p0x * p1y - p0y * p1x;
        if((e0 < R(0) || e1 < R(0) || e2 < R(0)) && (e0 > R(0) || e1 > R(0) || e2 > R(0))) return false;
        R det = e0 + e1 + e2;
        if(det == R(0)) return false;
        p0z *= rc.sz;
        p1z *= rc.sz;
        p2z *= rc.sz;
        R t_scaled = e0 * p0z + e1 * p1z + e2 * p2z;
        if(det < R(0) && (t_scaled >= R(0) || t_scaled < t_max * det)) return false;
        if(det > R(0) && (t_scaled <= R(0) || t_scaled > t_max * det)) return false;
        R inv_det = R(1) / det;
        b0 = e0 * inv_det;
        b1 = e1 * inv_det;
        t_hit = t_scaled * inv_det;
        return true;
    }

    template<typename R>
    FC_D bool sphere_intersect(const XPrim<R>& P, const RayCtx<R>& rc, R t_max, R& t_hit)
    {
        vec3<R> o = xf_point(P.inv, rc.o);
        vec3<R> d = xf_dir(P.inv, rc.d);
        R a = dot(d, d);
        R b = R(2) * dot(o, d);
        R c = dot(o, o) - P.a * P.a;
        R disc = b * b - R(4) * a * c;
        if(disc < R(0)) return false;
        R sq = r_sqrt(disc);
        R q = b < R(0) ? R(-0.5) * (b - sq) : R(-0.5) * (b + sq);
        R t0 = q / a;
        R t1 = c / q;
        if(t0 > t1)
        {
            R tmp = t0;
            t0 = t1;
            t1 = tmp;
        }
        R t = t0;
        if(t < R(0)) t = t1;
        if(t < R(0) || t > t_max) return false;  // NaN (q == 0) falls through like the reference: comparisons false
        t_hit = t;
        return true;
    }

    template<typename R>
    FC_D bool plane_intersect(const XPrim<R>& P, const RayCtx<R>& rc, R t_max, R& t_hit)
    {
        vec3<R> o = xf_point(P.inv, rc.o);
        vec3<R> d = xf_dir(P.inv, rc.d);
        R t = -o.y / d.y;
        if(t < R(0) || t > t_max || !r_isfinite(t)) return false;
        vec3<R> p = o + d * t;
        R hx = P.a / R(2), hz = P.b / R(2);
        if(p.x < -hx || p.x > hx || p.z < -hz || p.z > hz) return false;
        t_hit = t;
        return true;
    }

    FC_D uint32_t order_key(float t) { return __builtin_bit_cast(uint32_t, t); }  // t >= 0: bit order == value order
    FC_D uint32_t order_key(double t) { return order_key(static_cast<float>(t)); }

    // compare-exchange of (key, child) pairs: the smaller key ends in (ka, ca)
    FC_D void cswap(uint32_t& ka, uint32_t& kb, uint32_t& ca, uint32_t& cb)
    {
        bool s = kb < ka;
        uint32_t k0 = s ? kb : ka, k1 = s ? ka : kb;
        uint32_t c0 = s ? cb : ca, c1 = s ? ca : cb;
        ka = k0;
        kb = k1;
        ca = c0;
        cb = c1;
    }

    struct StackStats
    {
        // SPILL kernels: per lane. Non-SPILL kernels: `lost` is a per-WAVE count held by every lane (node_step), which
        // stat_add_lost() below adds once per wave.
        uint32_t lost{0};     // pushes beyond every capacity: a wrong image, must stay 0 (the host sizes the spill area)
        uint32_t spilled{0};  // pushes that went to the global spill area
    };

    template<typename R> FC_D R byte_to_real(uint32_t q, int c) { return R((q >> (8 * c)) & 0xFFu); }

    // Entry / exit distances of the four child boxes: plane = q * a + b per axis (a = cell size / d,
    // b = (grid origin - ray origin) / d), clipped to [0, t_max].
    template<typename R>
    struct slab4
    {
        static FC_D void run(uint32_t qnx, uint32_t qny, uint32_t qnz, uint32_t qfx, uint32_t qfy, uint32_t qfz, R ax, R ay, R az,
                             R bx, R by, R bz, R t_max, R* tn, R* tf)
        {
#pragma unroll
            for(int c = 0; c < 4; ++c)
            {
                R nx = byte_to_real<R>(qnx, c) * ax + bx, fx = byte_to_real<R>(qfx, c) * ax + bx;
                R ny = byte_to_real<R>(qny, c) * ay + by, fy = byte_to_real<R>(qfy, c) * ay + by;
                R nz = byte_to_real<R>(qnz, c) * az + bz, fz = byte_to_real<R>(qfz, c) * az + bz;
                tn[c] = r_max(r_max(nx, ny), r_max(nz, R(0)));
                tf[c] = r_min(r_min(fx, fy), r_min(fz, t_max));
            }
        }
    };

    // fp32: two children per v_pk_fma_f32 (packed fp32 issues two FMAs per lane per instruction on CDNA3/4)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    template<>
    struct slab4<float>
    {
        static FC_D f32x2 planes(uint32_t q, int c, float a, float b)
        {
            f32x2 v{byte_to_real<float>(q, c), byte_to_real<float>(q, c + 1)};
            return __builtin_elementwise_fma(v, f32x2{a, a}, f32x2{b, b});
        }
        static FC_D void run(uint32_t qnx, uint32_t qny, uint32_t qnz, uint32_t qfx, uint32_t qfy, uint32_t qfz, float ax, float ay,
                             float az, float bx, float by, float bz, float t_max, float* tn, float* tf)
        {
#pragma unroll
            for(int c = 0; c < 4; c += 2)
            {
                f32x2 nx = planes(qnx, c, ax, bx), fx = planes(qfx, c, ax, bx);
                f32x2 ny = planes(qny, c, ay, by), fy = planes(qfy, c, ay, by);
                f32x2 nz = planes(qnz, c, az, bz), fz = planes(qfz, c, az, bz);
                tn[c] = r_max(r_max(nx.x, ny.x), r_max(nz.x, 0.0f));
                tn[c + 1] = r_max(r_max(nx.y, ny.y), r_max(nz.y, 0.0f));
                tf[c] = r_min(r_min(fx.x, fy.x), r_min(fz.x, t_max));
                tf[c + 1] = r_min(r_min(fx.y, fy.y), r_min(fz.y, t_max));
            }
        }
    };

#ifndef FCG_ANY_UNSORTED
#define FCG_ANY_UNSORTED 1  // 0: occlusion rays sort their children like closest-hit rays (the A/B of DESIGN.md section 5)
#endif
    // One node step: slab-tests the four quantised child boxes of `cur`, pushes the hit children that
    // are not taken (far ones first) and returns the nearest one (or INVALID_U32 when none is hit).
    // The near / far plane of every axis is picked by the sign of the ray direction (rc.neg*), which
    // is what min/max of the two plane distances would give.
    // this lane's column of the global spill area, computed where it is used (rarely) instead of living in two registers
    FC_D size_t spill_column() { return static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; }
    // CAP: entries of the lane's LDS stack column (STACK_CAP, or the shorter one of the dense shadow kernel)
    template<typename R, bool ANY, int STRIDE, bool SPILL, int CAP = STACK_CAP>
    FC_D uint32_t node_step(const RayCtx<R>& rc, R t_max, uint4 n0, uint4 n1, uint4 n2, uint4 n3, uint32_t* stack, uint32_t* spill,
                            uint32_t spill_stride, int spill_cap, int& sp, StackStats& overflow, bool& lost_push)
    {
        R ox = R(__builtin_bit_cast(float, n0.x)), oy = R(__builtin_bit_cast(float, n0.y)), oz = R(__builtin_bit_cast(float, n0.z));
        R ax = R(__builtin_bit_cast(float, n0.w)) * rc.inv_d.x;  // grid steps: powers of two (Node4::sx, sy, sz)
        R ay = R(__builtin_bit_cast(float, n2.z)) * rc.inv_d.y;
        R az = R(__builtin_bit_cast(float, n2.w)) * rc.inv_d.z;
        R bx = (ox - rc.o.x) * rc.inv_d.x, by = (oy - rc.o.y) * rc.inv_d.y, bz = (oz - rc.o.z) * rc.inv_d.z;
        uint32_t qnx = rc.negx ? n1.w : n1.x, qfx = rc.negx ? n1.x : n1.w;
        uint32_t qny = rc.negy ? n2.x : n1.y, qfy = rc.negy ? n1.y : n2.x;
        uint32_t qnz = rc.negz ? n2.y : n1.z, qfz = rc.negz ? n1.z : n2.y;
        uint32_t ch[4]{n3.x, n3.y, n3.z, n3.w};
        uint32_t key[4];
        R tn[4], tf[4];
        slab4<R>::run(qnx, qny, qnz, qfx, qfy, qfz, ax, ay, az, bx, by, bz, t_max, tn, tf);
#if FCG_ANY_UNSORTED
        if constexpr(ANY)
        {
            // Occlusion rays: whether anything is hit does not depend on the visiting order, so the hit children are taken
            // in slot order, the first one to descend into and the others pushed: no keys, no sorting network (113 instead of
            // 132 VALU instructions per step, k_shadow -3 %). `room`: as below, one test per step; a SPILL kernel whose LDS
            // part is nearly full takes the general path below instead.
            const bool room = sp <= CAP - 3;
            if(!SPILL || room)
            {
                uint32_t next = INVALID_U32;
                bool lost = false;
#pragma unroll
                for(int c = 0; c < 4; ++c)
                {
                    const bool hit = (tn[c] <= tf[c] * R(1.0000005)) && (ch[c] != INVALID_U32);
                    const bool take = hit && next == INVALID_U32;
                    const bool push = hit && !take && room;
                    lost = lost || (hit && !take && !room);
                    next = take ? ch[c] : next;
                    if(push) stack[sp * STRIDE] = ch[c];
                    sp += push ? 1 : 0;
                }
                lost_push = lost;
                return next;
            }
        }
#endif
#pragma unroll
        for(int c = 0; c < 4; ++c)
        {
            bool hit = (tn[c] <= tf[c] * R(1.0000005)) && (ch[c] != INVALID_U32);
            key[c] = hit ? order_key(tn[c]) : INVALID_U32;
        }
        // sorting network on (entry distance, child) pairs; missed children (INVALID key) sink to the end.
        // (Occlusion rays take the unsorted path above unless FCG_ANY_UNSORTED is 0.)
        cswap(key[0], key[1], ch[0], ch[1]);
        cswap(key[2], key[3], ch[2], ch[3]);
        cswap(key[0], key[2], ch[0], ch[2]);
        cswap(key[1], key[3], ch[1], ch[3]);
        cswap(key[1], key[2], ch[1], ch[2]);
        uint32_t next = key[0] != INVALID_U32 ? ch[0] : INVALID_U32;
        if(!SPILL)
        {
            // one test per step instead of one per push: exact while the BVH's worst case is <= STACK_CAP - 2 entries (a step
            // that has anything to push then starts at sp <= CAP - 3); the upload gives every deeper hierarchy,
            // also one that would just fit, the SPILL instantiation, so `room` is always true here
            // a lost push (cannot happen, see above) is reported as a predicate: the caller counts it per wave on the
            // scalar unit (the ballot of a comparison is its mask), which costs the vector unit nothing
            const bool room = sp <= CAP - 3;
            lost_push = !room && key[1] != INVALID_U32;
#pragma unroll
            for(int k = 3; k >= 1; --k)
            {
                const bool push = room && key[k] != INVALID_U32;
                if(push) stack[sp * STRIDE] = ch[k];
                sp += push ? 1 : 0;
            }
        }
        else
        {
            // deep tree: entries beyond STACK_CAP go to this lane's column of the spill buffer; while three entries are free
            // in LDS (nearly always: the worst case is a bound over all rays) the pushes are those of the plain kernel
            if(sp <= CAP - 3)
            {
#pragma unroll
                for(int k = 3; k >= 1; --k)
                {
                    const bool push = key[k] != INVALID_U32;
                    if(push) stack[sp * STRIDE] = ch[k];
                    sp += push ? 1 : 0;
                }
            }
            else
#pragma unroll
            for(int k = 3; k >= 1; --k)
            {
                if(key[k] != INVALID_U32)
                {
                    if(sp < CAP)
                        stack[sp * STRIDE] = ch[k];
                    else if(sp < CAP + spill_cap)  // spill_cap: entries of the column actually supplied (SceneDev::stack_spill_cap)
                    {
                        spill[static_cast<size_t>(sp - CAP) * spill_stride + spill_column()] = ch[k];
                        overflow.spilled += 1;
                    }
                    else
                        overflow.lost += 1;
                    sp += sp < CAP + spill_cap ? 1 : 0;
                }
            }
        }
        return next;
    }

    template<int STRIDE, bool SPILL, int CAP = STACK_CAP> FC_D uint32_t stack_pop(const uint32_t* stack, const uint32_t* spill, uint32_t spill_stride, int& sp)
    {
        sp -= 1;
        if(SPILL && sp >= CAP) return spill[static_cast<size_t>(sp - CAP) * spill_stride + spill_column()];
        return stack[sp * STRIDE];
    }

    // Quad-cooperative node fetch (FCG_COOP_FETCH=1; off by default). A lane that loads its own
    // 64-byte node with four 16-byte loads costs the L1 (TA/TCP) four tag cycles for one line. Here the
    // four lanes of a quad load the four 16-byte pieces of ONE node per instruction (one line, one L1
    // transaction), for each of the quad's four owners in turn, straight into LDS
    // (global_load_lds_dwordx4: destination = wave base + lane * 16), and every owner then reads its
    // node back with ds_read_b128. Measured (tools/micro/node_gather.hip): a saturated CU gathers
    // 2200 random lines/us this way against 1000 with per-lane loads -- but traversal only asks for
    // ~220 lines/us/CU, so it is latency that counts there, and the extra DPP + LDS round trip and the
    // 16 KB of LDS per block made k_extend 3 % slower. Kept for scenes/hardware where the L1 saturates.
    // Layout of a wave's buffer: slot k (owner = lane 4q+k) at k * NODEBUF_SLOT, quad q at + q * 64;
    // the 16-byte skew between slots makes the read-back conflict-free.
    // Every lane of the wave must execute this (wave-uniform control flow): a lane that holds no
    // interior node still loads pieces for its quad mates.
#ifndef FCG_COOP_FETCH
#define FCG_COOP_FETCH 0
#endif
    constexpr int NODEBUF_SLOT = 64 * 16 + 16;
    constexpr int NODEBUF_WAVE = FCG_COOP_FETCH ? 4 * NODEBUF_SLOT : 16;
    typedef __attribute__((address_space(3))) void lds_void_t;
    typedef const __attribute__((address_space(1))) void global_void_t;

    template<int K>
    FC_D void coop_fetch_slot(const Node4* nodes, uint32_t cur, unsigned char* nodebuf)
    {
        const uint32_t owner_cur = static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(cur), K * 0x55, 0xF, 0xF, true));
        if(!child_is_leaf(owner_cur))
        {
            const unsigned char* g = reinterpret_cast<const unsigned char*>(nodes) + static_cast<size_t>(owner_cur) * sizeof(Node4) + (threadIdx.x & 3u) * 16u;
            __builtin_amdgcn_global_load_lds((global_void_t*)g, (lds_void_t*)(nodebuf + K * NODEBUF_SLOT), 16, 0, 0);
        }
    }

    FC_D void coop_fetch_nodes(const Node4* nodes, uint32_t cur, unsigned char* nodebuf)
    {
        coop_fetch_slot<0>(nodes, cur, nodebuf);
        coop_fetch_slot<1>(nodes, cur, nodebuf);
        coop_fetch_slot<2>(nodes, cur, nodebuf);
        coop_fetch_slot<3>(nodes, cur, nodebuf);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    FC_D const uint4* coop_my_node(const unsigned char* nodebuf)
    {
        const uint32_t lane = threadIdx.x & 63u;
        return reinterpret_cast<const uint4*>(nodebuf + (lane & 3u) * NODEBUF_SLOT + (lane >> 2) * 64u);
    }

    // Resumable per-lane traversal state. pk = kind(2) | index(30) of the best primitive so far.
    template<typename R>
    struct Trav
    {
        RayCtx<R> rc;
        R t_max;
        R b0, b1;
        uint32_t pk;
        uint32_t cur;  // interior node index, leaf word, or INVALID_U32 = nothing left (leaf bit set)
        uint32_t post; // SPEC rounds: a leaf reached but not tested yet (the lane went on descending), else INVALID_U32
        int sp;
    };

    template<typename R> FC_D void trav_begin(Trav<R>& T, vec3<R> o, vec3<R> d, R t_max)
    {
        T.rc = make_ray(o, d);
        T.t_max = t_max;
        T.b0 = R(0);
        T.b1 = R(0);
        T.pk = INVALID_U32;
        T.cur = 0;  // root is always a wide node
        T.post = INVALID_U32;
        T.sp = 0;
    }
    template<typename R> FC_D bool trav_done(const Trav<R>& T) { return T.cur == INVALID_U32 && T.post == INVALID_U32; }

    // One round of the while-while traversal for the whole wave: every lane that holds an interior
    // node descends until it holds a leaf (the descent loop ends early once fewer than DESCEND_MIN
    // lanes are still descending while others wait with a leaf), then the lanes holding a leaf test
    // its primitives together. Lanes with cur == INVALID_U32 idle through it.
    // Closest hit (ANY = false): T.t_max / T.pk / T.b0 / T.b1 track the best hit; accepts t == t_max.
    // Any hit (ANY = true): the first primitive hit sets T.pk and ends the lane's traversal.
    // `stack` points at this lane's column: entry k is stack[k * STRIDE]; `nodebuf` is the wave's
    // NODEBUF_WAVE-byte LDS staging area. All 64 lanes of the wave must call it together.

    // n_nodes: PROF: this lane's node steps; otherwise the WAVE's node steps, the same value in every lane (add it once
    // per wave: wave_stat_add_uniform). n_prims: this lane's primitive tests.
    // PROF (development aid): ph[0] += clock cycles in the descent loop, ph[1] += cycles in the leaf
    // phase, ph[2] += wave-level descent trips, ph[3] += leaf phases entered, ph[6] += lanes holding a leaf in them.
    // SPEC (speculative descent, after Aila & Laine 2009): a lane that reaches a leaf parks it in T.post and goes on with the
    // next entry of its stack instead of idling until the wave's leaf phase; it waits only once it holds a second leaf (or its
    // stack is empty). The primitives tested are the same for an ANY ray that is not occluded (nothing shrinks its t_max).
    template<typename R, bool ANY, int STRIDE, bool SPILL, bool PROF = false, bool SPEC = false, int CAP = STACK_CAP>
    FC_D void trav_round(const SceneDev<R>& S, Trav<R>& T, uint32_t* stack, unsigned char* nodebuf, StackStats& overflow, int descend_min,
                         uint32_t& n_nodes, uint32_t& n_prims, unsigned long long* ph = nullptr)
    {
        unsigned long long c0 = 0;
        if(PROF) c0 = clock64();
        uint32_t* spill = SPILL ? S.stack_spill : nullptr;  // uniform: a lane's column is added where an entry is touched (spill_column)
        uint32_t cur = T.cur;
        uint32_t post = SPEC ? T.post : INVALID_U32;
        int sp = T.sp;
        bool leaf_waiting = SPEC ? __ballot(post != INVALID_U32) != 0ull : __ballot(child_is_leaf(cur) && cur != INVALID_U32) != 0ull;
        for(;;)
        {
            const bool descending = !child_is_leaf(cur);
            const uint64_t dm = __ballot(descending);
            if(dm == 0ull || (leaf_waiting && __popcll(dm) < descend_min)) break;
            if(PROF)
            {
                ph[2] += 1;
                ph[13] += __popcll(__ballot(child_is_leaf(cur) && cur != INVALID_U32));  // lanes waiting with a leaf through this trip
                ph[14] += __popcll(__ballot(cur == INVALID_U32));                        // lanes with a finished ray (or none)
            }
#if FCG_COOP_FETCH
            coop_fetch_nodes(S.nodes, cur, nodebuf);
#endif
            bool lost_push = false;
            if(descending)
            {
#if FCG_COOP_FETCH
                const uint4* np = coop_my_node(nodebuf);
#else
                const uint4* np = reinterpret_cast<const uint4*>(S.nodes + cur);
#endif
#if FCG_TREELET
                // top of the tree from the block's LDS copy (`nodebuf` is that copy in this build), the rest from L1 / L2
                uint4 nq[4];
                if(cur < static_cast<uint32_t>(FCG_TREELET))
                {
                    const uint4* lp = reinterpret_cast<const uint4*>(nodebuf) + 4u * cur;
                    nq[0] = lp[0], nq[1] = lp[1], nq[2] = lp[2], nq[3] = lp[3];
                }
                else
                {
                    nq[0] = np[0], nq[1] = np[1], nq[2] = np[2], nq[3] = np[3];
                }
                np = nq;
#endif
                uint32_t next = node_step<R, ANY, STRIDE, SPILL, CAP>(T.rc, T.t_max, np[0], np[1], np[2], np[3], stack, spill, S.stack_spill_stride, static_cast<int>(S.stack_spill_cap), sp, overflow, lost_push);
                if(PROF) n_nodes += 1;  // per lane (the timeline wants the lane maximum); otherwise per wave, below
                if(PROF && next == INVALID_U32)
                {
                    // steps that found nothing: ph[11]; of those, the ones a shrunken t_max emptied (a stack entry pushed
                    // before the hit was found: what keeping entry distances on the stack could skip): ph[12]
                    ph[11] += 1;
                    int sp2 = sp;
                    StackStats dummy;
                    bool dummy_lost = false;
                    if(!SPILL && sp <= CAP - 3 &&
                       node_step<R, ANY, STRIDE, SPILL, CAP>(T.rc, consts<R>::inf, np[0], np[1], np[2], np[3], stack, spill, S.stack_spill_stride, static_cast<int>(S.stack_spill_cap), sp2, dummy, dummy_lost) != INVALID_U32)
                        ph[12] += 1;
                }
                if(next == INVALID_U32 && sp > 0) next = stack_pop<STRIDE, SPILL, CAP>(stack, spill, S.stack_spill_stride, sp);
                cur = next;
            }
            if(!PROF) n_nodes += __popcll(dm);  // node steps of the whole wave, counted on the scalar unit
            if(!SPILL) overflow.lost += __popcll(__ballot(lost_push));
            if(SPEC)
            {
                if(post == INVALID_U32 && child_is_leaf(cur) && cur != INVALID_U32)
                {
                    post = cur;
                    cur = sp > 0 ? stack_pop<STRIDE, SPILL, CAP>(stack, spill, S.stack_spill_stride, sp) : INVALID_U32;
                }
                leaf_waiting = leaf_waiting || __ballot(post != INVALID_U32) != 0ull;
            }
            else
                leaf_waiting = leaf_waiting || __ballot(child_is_leaf(cur) && cur != INVALID_U32) != 0ull;
        }
        unsigned long long c1 = 0;
        if(PROF)
        {
            c1 = clock64();
            ph[0] += c1 - c0;
            const uint64_t lm = SPEC ? __ballot(post != INVALID_U32) : __ballot(child_is_leaf(cur) && cur != INVALID_U32);
            ph[3] += lm != 0ull ? 1 : 0;
            ph[6] += __popcll(lm);
        }
        const uint32_t leaf = SPEC ? post : cur;
        if(child_is_leaf(leaf) && leaf != INVALID_U32)
        {
            const uint32_t first = leaf_first(leaf), kind = leaf_kind(leaf);
            bool stop = false;
            if(kind == KIND_TRI)
            {
                const uint32_t count = leaf_count(leaf);
                n_prims += count;
                // all records of the leaf are requested before the first test: one memory latency per
                // leaf instead of one per triangle
                TriRec tr[MAX_LEAF_PRIMS];
#pragma unroll
                for(uint32_t j = 0; j < static_cast<uint32_t>(MAX_LEAF_PRIMS); ++j)
                    if(j < count) tr[j] = S.tris[first + j];
#pragma unroll
                for(uint32_t j = 0; j < static_cast<uint32_t>(MAX_LEAF_PRIMS); ++j)
                {
                    R t, b0, b1;
                    if(j < count && !stop && tri_intersect(tr[j], T.rc, T.t_max, t, b0, b1))
                    {
                        T.t_max = t;
                        T.pk = (KIND_TRI << 30) | (first + j);
                        T.b0 = b0;
                        T.b1 = b1;
                        if(ANY) stop = true;
                    }
                }
            }
            else
            {
                R t;
                n_prims += 1;
                bool hit = kind == KIND_SPHERE ? sphere_intersect(S.xprims[first], T.rc, T.t_max, t) : plane_intersect(S.xprims[first], T.rc, T.t_max, t);
                if(hit)
                {
                    T.t_max = t;
                    T.pk = (kind << 30) | first;
                    T.b0 = R(0);
                    T.b1 = R(0);
                    if(ANY) stop = true;
                }
            }
            if(SPEC)
            {
                // the parked leaf is done; a second leaf the lane was waiting with takes its place
                post = INVALID_U32;
                if(stop)
                    cur = INVALID_U32;
                else if(child_is_leaf(cur) && cur != INVALID_U32)
                {
                    post = cur;
                    cur = sp > 0 ? stack_pop<STRIDE, SPILL, CAP>(stack, spill, S.stack_spill_stride, sp) : INVALID_U32;
                }
            }
            else
                cur = (!stop && sp > 0) ? stack_pop<STRIDE, SPILL, CAP>(stack, spill, S.stack_spill_stride, sp) : INVALID_U32;
        }
        T.cur = cur;
        if(SPEC) T.post = post;
        T.sp = sp;
        if(PROF) ph[1] += clock64() - c1;
    }

    // ------------------------------------------------------------------ 8-wide hierarchy (Node8, fc_types.h)
    // Per-lane state. `word` != 0: the lane holds the hit LEAF children of the node it stepped last (`cur` = that node's
    // tri_base; word = hits in visiting order (8) | leaf slots (8) | two-record slots (8) | bit 31) and waits for the
    // wave's leaf phase. word == 0: `cur` is the node to step next, or INVALID_U32 = finished. The groups of hit
    // INTERIOR children that still wait live on the LDS stack, two words each: {child_base, hits in visiting order (8) |
    // imask (8)}; visiting order = slot XOR the ray's octant, lowest first.
    template<typename R>
    struct Trav8
    {
        RayCtx<R> rc;
        R t_max;
        R b0, b1;
        uint32_t pk;
        uint32_t cur, word;
        uint32_t oct;  // bit a: the ray travels towards -a
        int sp;        // group entries on the stack
    };

    template<typename R> FC_D void trav_begin(Trav8<R>& T, vec3<R> o, vec3<R> d, R t_max)
    {
        T.rc = make_ray(o, d);
        T.t_max = t_max;
        T.b0 = R(0);
        T.b1 = R(0);
        T.pk = INVALID_U32;
        T.cur = 0;
        T.word = 0;
        T.oct = (T.rc.negx ? 1u : 0u) | (T.rc.negy ? 2u : 0u) | (T.rc.negz ? 4u : 0u);
        T.sp = 0;
    }
    template<typename R> FC_D bool trav_done(const Trav8<R>& T) { return T.cur == INVALID_U32; }
    template<typename R> FC_D void trav_idle(Trav8<R>& T)
    {
        T.cur = INVALID_U32;
        T.word = 0;
        T.sp = 0;
    }
    template<typename R> FC_D void trav_idle(Trav<R>& T)
    {
        T.cur = INVALID_U32;
        T.post = INVALID_U32;
        T.sp = 0;
    }

    // next node of the topmost group, or INVALID_U32 when the stack is empty
    template<int STRIDE> FC_D uint32_t pop8(uint32_t* stack, int& sp, uint32_t oct)
    {
        if(sp == 0) return INVALID_U32;
        const uint32_t base = stack[(2 * (sp - 1)) * STRIDE];
        uint32_t w = stack[(2 * (sp - 1) + 1) * STRIDE];
        const uint32_t k = static_cast<uint32_t>(__builtin_ctz(w)), slot = k ^ oct;  // (w & 0xFF) != 0 for every entry on the stack
        const uint32_t next = base + static_cast<uint32_t>(__builtin_popcount((w >> 8) & ((1u << slot) - 1u)));
        w &= w - 1u;
        if(w & 0xFFu)
            stack[(2 * (sp - 1) + 1) * STRIDE] = w;
        else
            sp -= 1;
        return next;
    }

    // entry / exit distances of two children per call: byte c and c + 1 of the near / far plane words
    template<typename R> struct planes8
    {
        static FC_D void run(uint32_t q, int c, R a, R b, R& p0, R& p1)
        {
            p0 = byte_to_real<R>(q, c) * a + b;
            p1 = byte_to_real<R>(q, c + 1) * a + b;
        }
    };
    template<> struct planes8<float>
    {
        static FC_D void run(uint32_t q, int c, float a, float b, float& p0, float& p1)
        {
            const f32x2 v = slab4<float>::planes(q, c, a, b);
            p0 = v.x;
            p1 = v.y;
        }
    };

    // One step on an 8-wide node: which children the ray hits, as masks in visiting order; pushes the interior group,
    // leaves the leaf group in (cur, word) or pops the next node. Returns false when a push found no room (counted by the caller).
    template<typename R, int STRIDE>
    FC_D bool node_step8(const RayCtx<R>& rc, uint32_t oct, R t_max, const uint4 n0, const uint4 n1, const uint4 n2, const uint4 n3, const uint4 n4,
                         uint32_t* stack, int& sp, uint32_t& cur, uint32_t& word)
    {
        const R ox = R(__builtin_bit_cast(float, n0.x)), oy = R(__builtin_bit_cast(float, n0.y)), oz = R(__builtin_bit_cast(float, n0.z));
        const uint32_t ex = n0.w;
        const R ax = R(__builtin_bit_cast(float, (ex & 0xFFu) << 23)) * rc.inv_d.x;
        const R ay = R(__builtin_bit_cast(float, (ex & 0xFF00u) << 15)) * rc.inv_d.y;
        const R az = R(__builtin_bit_cast(float, (ex & 0xFF0000u) << 7)) * rc.inv_d.z;
        const R bx = (ox - rc.o.x) * rc.inv_d.x, by = (oy - rc.o.y) * rc.inv_d.y, bz = (oz - rc.o.z) * rc.inv_d.z;
        // n2 = qlo x (2), qlo y (2); n3 = qlo z (2), qhi x (2); n4 = qhi y (2), qhi z (2)
        const uint32_t nx[2]{rc.negx ? n3.z : n2.x, rc.negx ? n3.w : n2.y}, fx[2]{rc.negx ? n2.x : n3.z, rc.negx ? n2.y : n3.w};
        const uint32_t ny[2]{rc.negy ? n4.x : n2.z, rc.negy ? n4.y : n2.w}, fy[2]{rc.negy ? n2.z : n4.x, rc.negy ? n2.w : n4.y};
        const uint32_t nz[2]{rc.negz ? n4.z : n3.x, rc.negz ? n4.w : n3.y}, fz[2]{rc.negz ? n3.x : n4.z, rc.negz ? n3.y : n4.w};
        uint32_t hits = 0;  // slot order
#pragma unroll
        for(int h = 0; h < 2; ++h)
        {
#pragma unroll
            for(int c = 0; c < 4; c += 2)
            {
                R tnx0, tnx1, tny0, tny1, tnz0, tnz1, tfx0, tfx1, tfy0, tfy1, tfz0, tfz1;
                planes8<R>::run(nx[h], c, ax, bx, tnx0, tnx1);
                planes8<R>::run(ny[h], c, ay, by, tny0, tny1);
                planes8<R>::run(nz[h], c, az, bz, tnz0, tnz1);
                planes8<R>::run(fx[h], c, ax, bx, tfx0, tfx1);
                planes8<R>::run(fy[h], c, ay, by, tfy0, tfy1);
                planes8<R>::run(fz[h], c, az, bz, tfz0, tfz1);
                const R tn0 = r_max(r_max(tnx0, tny0), r_max(tnz0, R(0))), tf0 = r_min(r_min(tfx0, tfy0), r_min(tfz0, t_max));
                const R tn1 = r_max(r_max(tnx1, tny1), r_max(tnz1, R(0))), tf1 = r_min(r_min(tfx1, tfy1), r_min(tfz1, t_max));
                hits |= (tn0 <= tf0 * R(1.0000005) ? 1u : 0u) << (4 * h + c);
                hits |= (tn1 <= tf1 * R(1.0000005) ? 1u : 0u) << (4 * h + c + 1);
            }
        }
        const uint32_t valid = n1.x >> 24, imask = ex >> 24;
        hits &= valid;
        // interior hits in byte 0, leaf hits in byte 1: both into visiting order (bit k = slot k XOR oct)
        uint32_t x = (hits & imask) | (hits & ~imask) << 8;
        const uint32_t x1 = ((x & 0x5555u) << 1) | ((x >> 1) & 0x5555u);
        x = (oct & 1u) ? x1 : x;
        const uint32_t x2 = ((x & 0x3333u) << 2) | ((x >> 2) & 0x3333u);
        x = (oct & 2u) ? x2 : x;
        const uint32_t x4 = ((x & 0x0F0Fu) << 4) | ((x >> 4) & 0x0F0Fu);
        x = (oct & 4u) ? x4 : x;
        const uint32_t int_key = x & 0xFFu, leaf_key = x >> 8;
        bool room = true;
        if(int_key)
        {
            room = sp < STACK8_CAP;
            if(room)
            {
                stack[(2 * sp) * STRIDE] = n1.x & 0x00FFFFFFu;
                stack[(2 * sp + 1) * STRIDE] = int_key | imask << 8;
                sp += 1;
            }
        }
        if(leaf_key)
        {
            cur = n1.y;
            word = leaf_key | (valid & ~imask) << 8 | (n1.z & 0xFFu) << 16 | 0x80000000u;
        }
        else
            cur = pop8<STRIDE>(stack, sp, oct);
        return room;
    }

    // One round for the whole wave, like trav_round: lanes holding a node step until enough of them hold leaves, then the
    // lanes holding leaves test ONE hit leaf child (at most two records) each.
    template<typename R, bool ANY, int STRIDE>
    FC_D void trav_round8(const SceneDev<R>& S, Trav8<R>& T, uint32_t* stack, StackStats& overflow, int descend_min, uint32_t& n_nodes, uint32_t& n_prims)
    {
        uint32_t cur = T.cur, word = T.word;
        int sp = T.sp;
        bool leaf_waiting = __ballot(word != 0u) != 0ull;
        for(;;)
        {
            const bool descending = word == 0u && cur != INVALID_U32;
            const uint64_t dm = __ballot(descending);
            if(dm == 0ull || (leaf_waiting && __popcll(dm) < descend_min)) break;
            bool lost_push = false;
            if(descending)
            {
                const uint4* np = reinterpret_cast<const uint4*>(S.nodes8 + cur);
                lost_push = !node_step8<R, STRIDE>(T.rc, T.oct, T.t_max, np[0], np[1], np[2], np[3], np[4], stack, sp, cur, word);
            }
            n_nodes += __popcll(dm);
            overflow.lost += __popcll(__ballot(lost_push));
            leaf_waiting = leaf_waiting || __ballot(word != 0u) != 0ull;
        }
        if(word != 0u)
        {
            const uint32_t k = static_cast<uint32_t>(__builtin_ctz(word)), slot = k ^ T.oct, below = (1u << slot) - 1u;
            const uint32_t leaf_slots = (word >> 8) & 0xFFu, two = (word >> 16) & 0xFFu;
            const uint32_t first = cur + static_cast<uint32_t>(__builtin_popcount(leaf_slots & below)) + static_cast<uint32_t>(__builtin_popcount(two & below));
            const uint32_t count = 1u + ((two >> slot) & 1u);
            n_prims += count;
            TriRec tr[2];
            tr[0] = S.tris[first];
            if(count > 1u) tr[1] = S.tris[first + 1u];
            bool stop = false;
#pragma unroll
            for(uint32_t j = 0; j < 2u; ++j)
            {
                if(j < count && !stop)
                {
                    R t, b0 = R(0), b1 = R(0);
                    bool hit;
                    uint32_t pk;
                    if(tr[j].attr == XPRIM_PROXY)
                    {
                        const uint32_t kind = tr[j].entity, idx = tr[j].prim;
                        hit = kind == KIND_SPHERE ? sphere_intersect(S.xprims[idx], T.rc, T.t_max, t) : plane_intersect(S.xprims[idx], T.rc, T.t_max, t);
                        pk = (kind << 30) | idx;
                    }
                    else
                    {
                        hit = tri_intersect(tr[j], T.rc, T.t_max, t, b0, b1);
                        pk = (KIND_TRI << 30) | (first + j);
                    }
                    if(hit)
                    {
                        T.t_max = t;
                        T.pk = pk;
                        T.b0 = b0;
                        T.b1 = b1;
                        if(ANY) stop = true;
                    }
                }
            }
            word &= word - 1u;  // this leaf is done
            if(stop)
            {
                cur = INVALID_U32;
                word = 0u;
            }
            else if((word & 0xFFu) == 0u)
            {
                word = 0u;
                cur = pop8<STRIDE>(stack, sp, T.oct);
            }
        }
        T.cur = cur;
        T.word = word;
        T.sp = sp;
    }

    // Whole-wave convenience wrapper (parity hooks): runs rounds until every ray of the wave is finished.
    // Lanes without a ray pass active = false (they still take part in the cooperative fetches).
    template<typename R, bool ANY, int STRIDE, bool SPILL, int CAP = STACK_CAP>
    FC_D bool traverse(const SceneDev<R>& S, bool active, const RayCtx<R>& rc, R t_max, uint32_t* stack, unsigned char* nodebuf, R& t_out,
                       uint32_t& pk_out, R& b0_out, R& b1_out, StackStats& overflow)
    {
        Trav<R> T;
        T.rc = rc;
        T.t_max = t_max;
        T.b0 = R(0);
        T.b1 = R(0);
        T.pk = INVALID_U32;
        T.cur = active ? 0u : INVALID_U32;
        T.post = INVALID_U32;
        T.sp = 0;
        uint32_t nn = 0, np = 0;
        while(__ballot(!trav_done(T)) != 0ull) trav_round<R, ANY, STRIDE, SPILL, false, false, CAP>(S, T, stack, nodebuf, overflow, 12, nn, np);
        t_out = T.t_max;
        pk_out = T.pk;
        b0_out = T.b0;
        b1_out = T.b1;
        return T.pk != INVALID_U32;
    }

    // Wave-level dynamic work distribution. The ray queue is cut into WORK_RANGES equal ranges, each
    // with its own cursor word; a wave starts in the range of its block (blockIdx & 7: blocks are dealt
    // round-robin to the 8 XCDs, so a range is mostly served by one XCD) and moves on to the next range
    // when its own is exhausted. Idle lanes take ray indices from the wave's current chunk; a new chunk
    // costs ONE atomic per wave. The chunk size follows the queue (work_chunk): large queues are dealt in
    // chunks of 512 consecutive rays, which a wave then works through on its own -- measured at 132 M rays per
    // launch against chunks of 64: k_extend -2.7 %, k_shadow -5 % (neighbours in the queue were emitted by
    // neighbouring shade threads and, for shadow rays, all aim at the lights) --; small queues keep chunks of 64
    // so that the end-of-launch tail stays short: when the cursors run dry every wave still has to finish at most one chunk.
#ifndef FCG_WORK_CHUNK_MIN
#define FCG_WORK_CHUNK_MIN 64
#endif
#ifndef FCG_WORK_CHUNK_MAX
#define FCG_WORK_CHUNK_MAX 512
#endif
    FC_D uint32_t work_chunk(uint32_t n)
    {
        const uint32_t c = n >> 16;  // ~1/8 of a wave's fair share of a queue served by 8192 waves
        return c < FCG_WORK_CHUNK_MIN ? FCG_WORK_CHUNK_MIN : (c > FCG_WORK_CHUNK_MAX ? FCG_WORK_CHUNK_MAX : c);
    }

    struct WorkCursor
    {
        uint32_t cur, end;     // wave-uniform: rest of the current chunk
        uint32_t range, tried; // current range; ranges found exhausted (WORK_RANGES = nothing left)
        uint32_t chunk;        // rays per chunk for this launch (work_chunk)
    };

    FC_D WorkCursor work_begin(uint32_t n) { return WorkCursor{0u, 0u, blockIdx.x & (WORK_RANGES - 1), 0u, work_chunk(n)}; }
    FC_D uint32_t work_range_lo(uint32_t n, uint32_t r) { return static_cast<uint32_t>((static_cast<uint64_t>(n) * r) / WORK_RANGES); }

    // returns this lane's new ray index or INVALID_U32; `want` = lane is idle
    FC_D uint32_t work_take(WorkCursor& W, bool want, uint32_t (*counters)[32], uint32_t n)
    {
        uint32_t mine = INVALID_U32;
        for(;;)
        {
            uint64_t m = __ballot(want && mine == INVALID_U32);
            uint32_t cnt = __popcll(m);
            if(cnt == 0u) break;
            if(W.cur >= W.end)
            {
                if(W.tried >= static_cast<uint32_t>(WORK_RANGES)) break;
                const uint32_t lo = work_range_lo(n, W.range), hi = work_range_lo(n, W.range + 1u);
                uint32_t base = 0;
                if((threadIdx.x & 63) == 0) base = atomicAdd(&counters[W.range][0], W.chunk);
                base = __builtin_amdgcn_readfirstlane(base);
                if(base >= hi - lo)
                {
                    W.range = (W.range + 1u) & (WORK_RANGES - 1);
                    W.tried += 1u;
                    continue;
                }
                W.cur = lo + base;
                W.end = hi - W.cur < W.chunk ? hi : W.cur + W.chunk;
            }
            uint32_t avail = W.end - W.cur;
            uint32_t rank = __popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull));
            if(want && mine == INVALID_U32 && rank < avail) mine = W.cur + rank;
            W.cur += cnt < avail ? cnt : avail;
        }
        return mine;
    }

    // object-space point where the float ray meets the sphere P, in double arithmetic (same root choice as sphere_intersect;
    // `t32` = the root the fp32 test accepted, used when the double solve finds none: a grazing hit)
    FC_D vec3<float> sphere_point_f64(const XPrim<float>& P, const RayCtx<float>& rc, float t32)
    {
        // one object-space component at a time (and once more for the point): few doubles live at once, so that the finish
        // phase of k_extend stays within the registers of five waves per SIMD
        auto comp_o = [&](int i) { return double(P.inv[4 * i]) * double(rc.o.x) + double(P.inv[4 * i + 1]) * double(rc.o.y) + double(P.inv[4 * i + 2]) * double(rc.o.z) + double(P.inv[4 * i + 3]); };
        auto comp_d = [&](int i) { return double(P.inv[4 * i]) * double(rc.d.x) + double(P.inv[4 * i + 1]) * double(rc.d.y) + double(P.inv[4 * i + 2]) * double(rc.d.z); };
        double a = 0.0, hb = 0.0, c = -double(P.a) * double(P.a);
#pragma unroll
        for(int i = 0; i < 3; ++i)
        {
            const double o = comp_o(i), d = comp_d(i);
            a += d * d;
            hb += o * d;
            c += o * o;
        }
        const double b = 2.0 * hb;
        const double disc = b * b - 4.0 * a * c;
        double t = double(t32);
        if(disc >= 0.0)
        {
            const double sq = sqrt(disc);
            const double q = b < 0.0 ? -0.5 * (b - sq) : -0.5 * (b + sq);
            const double t0 = q / a, t1 = c / q;
            // the root nearest to the one the fp32 test took (it chose the first non-negative one)
            const double pick = fabs(t0 - double(t32)) <= fabs(t1 - double(t32)) ? t0 : t1;
            if(pick == pick) t = pick;
        }
        return vec3<float>{static_cast<float>(comp_o(0) + comp_d(0) * t), static_cast<float>(comp_o(1) + comp_d(1) * t), static_cast<float>(comp_o(2) + comp_d(2) * t)};
    }
    template<typename R> FC_D vec3<R> sphere_point_f64(const XPrim<R>&, const RayCtx<R>&, R) { return vec3<R>{}; }  // (never called for R = double)

    // geometry of a hit: world position and geometric normal, as the reference's raycast_surface_point builds them
    template<typename R>
    FC_D void hit_point(const SceneDev<R>& S, const RayCtx<R>& rc, uint32_t pk, R t, R b0, R b1, vec3<R>& p, vec3<R>& n,
                        uint32_t& prim, uint32_t& entity, uint32_t& key_hint)
    {
        key_hint = KEY_HINT_NONE;
        uint32_t kind = pk >> 30, idx = pk & 0x3FFFFFFFu;
        if(kind == KIND_TRI)
        {
            const TriRec& T = S.tris[idx];
            vec3<R> p0 = ldv<R>(T.v0), p1 = ldv<R>(T.v1), p2 = ldv<R>(T.v2);
            R b2 = R(1) - b0 - b1;
            p = b0 * p0 + b1 * p1 + b2 * p2;              // mesh_surface.hpp:211
            n = normalize(cross(p0 - p2, p1 - p2));       // :212-213, 227
            prim = T.prim;
            entity = T.entity & ENTITY_MASK;
            key_hint = T.entity >> 24;
        }
        else
        {
            const XPrim<R>& P = S.xprims[idx];
            vec3<R> ol = xf_point(P.inv, rc.o), dl = xf_dir(P.inv, rc.d);
            vec3<R> pl = ol + dl * t;
            if constexpr(sizeof(R) == 4)
            {
                // fp32: the object-space hit point of a SPHERE solved once more in double from the same float ray (the
                // accepted hit only, once per ray). In fp32 the quadratic cancels (b^2 - 4ac = 1024 - 1008 for the room's
                // spheres seen from the camera) and leaves 4e-6 in t, i.e. 2e-6 in the normal of a radius-2 sphere, which a
                // mirror bounce carries to the walls: measured on the 1080p smooth-mirror crop.
                if(kind == KIND_SPHERE) pl = sphere_point_f64(P, rc, t);
            }
            p = xf_point(P.fwd, pl);                      // sphere_surface.hpp:123-126, plane_surface.hpp:79-88
            if(kind == KIND_SPHERE)
                n = xf_dir(P.fwd, normalize(pl));         // sphere_surface.hpp:127
            else
                n = vec3<R>{P.fwd[1], P.fwd[5], P.fwd[9]};  // plane_surface.hpp:89
            prim = P.prim;
            entity = P.entity;
        }
    }
}
