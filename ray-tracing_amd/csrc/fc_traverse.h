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

    template<typename R> FC_D R exp2_from_biased(uint32_t e);
    template<> FC_D float exp2_from_biased<float>(uint32_t e) { return __builtin_bit_cast(float, e << 23); }
    template<> FC_D double exp2_from_biased<double>(uint32_t e)
    {
        return __builtin_bit_cast(double, static_cast<uint64_t>(e + (1023u - 127u)) << 52);
    }

    FC_D uint32_t order_key(float t, uint32_t c) { return (__builtin_bit_cast(uint32_t, t) & ~3u) | c; }
    FC_D uint32_t order_key(double t, uint32_t c) { return order_key(static_cast<float>(t), c); }

    FC_D void cswap(uint32_t& a, uint32_t& b)
    {
        uint32_t lo = a < b ? a : b;
        uint32_t hi = a < b ? b : a;
        a = lo;
        b = hi;
    }

    // One node step: slab-tests the four quantised child boxes of `cur`, pushes the hit children that
    // are not taken (far ones first) and returns the nearest one (or INVALID_U32 when none is hit).
    template<typename R, bool ANY, int STRIDE>
    FC_D uint32_t node_step(const SceneDev<R>& S, const RayCtx<R>& rc, R t_max, uint32_t cur, uint32_t* stack, int& sp, uint32_t& overflow)
    {
        const uint4* np = reinterpret_cast<const uint4*>(S.nodes + cur);
        uint4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
        R ox = R(__builtin_bit_cast(float, n0.x)), oy = R(__builtin_bit_cast(float, n0.y)), oz = R(__builtin_bit_cast(float, n0.z));
        R ax = exp2_from_biased<R>(n0.w & 0xFFu) * rc.inv_d.x;
        R ay = exp2_from_biased<R>((n0.w >> 8) & 0xFFu) * rc.inv_d.y;
        R az = exp2_from_biased<R>((n0.w >> 16) & 0xFFu) * rc.inv_d.z;
        R bx = (ox - rc.o.x) * rc.inv_d.x, by = (oy - rc.o.y) * rc.inv_d.y, bz = (oz - rc.o.z) * rc.inv_d.z;
        uint32_t qlx = n1.x, qly = n1.y, qlz = n1.z, qhx = n1.w, qhy = n2.x, qhz = n2.y;
        uint32_t ch[4]{n3.x, n3.y, n3.z, n3.w};
        uint32_t key[4];
#pragma unroll
        for(int c = 0; c < 4; ++c)
        {
            R lx = R((qlx >> (8 * c)) & 0xFFu) * ax + bx, hx = R((qhx >> (8 * c)) & 0xFFu) * ax + bx;
            R ly = R((qly >> (8 * c)) & 0xFFu) * ay + by, hy = R((qhy >> (8 * c)) & 0xFFu) * ay + by;
            R lz = R((qlz >> (8 * c)) & 0xFFu) * az + bz, hz = R((qhz >> (8 * c)) & 0xFFu) * az + bz;
            R tn = r_max(r_max(r_min(lx, hx), r_min(ly, hy)), r_max(r_min(lz, hz), R(0)));
            R tf = r_min(r_min(r_max(lx, hx), r_max(ly, hy)), r_min(r_max(lz, hz), t_max));
            bool hit = (tn <= tf * R(1.0000005)) && (ch[c] != INVALID_U32);
            key[c] = hit ? order_key(tn, static_cast<uint32_t>(c)) : INVALID_U32;
        }
        if(!ANY)
        {
            cswap(key[0], key[1]);
            cswap(key[2], key[3]);
            cswap(key[0], key[2]);
            cswap(key[1], key[3]);
            cswap(key[1], key[2]);
        }
        uint32_t next = INVALID_U32;
#pragma unroll
        for(int k = 3; k >= 0; --k)
        {
            if(key[k] != INVALID_U32)
            {
                uint32_t c = key[k] & 3u;
                uint32_t child = c == 0 ? ch[0] : (c == 1 ? ch[1] : (c == 2 ? ch[2] : ch[3]));
                if(next != INVALID_U32)
                {
                    if(sp < STACK_CAP)
                        stack[(sp++) * STRIDE] = next;
                    else
                        overflow += 1;
                }
                next = child;
            }
        }
        return next;
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
        T.sp = 0;
    }
    template<typename R> FC_D bool trav_done(const Trav<R>& T) { return T.cur == INVALID_U32; }

    // One round of the while-while traversal for the whole wave: every lane that holds an interior
    // node descends until it holds a leaf (the descent loop ends early once fewer than DESCEND_MIN
    // lanes are still descending while others wait with a leaf), then the lanes holding a leaf test
    // its primitives together. Lanes with cur == INVALID_U32 idle through it.
    // Closest hit (ANY = false): T.t_max / T.pk / T.b0 / T.b1 track the best hit; accepts t == t_max.
    // Any hit (ANY = true): the first primitive hit sets T.pk and ends the lane's traversal.
    // `stack` points at this lane's column: entry k is stack[k * STRIDE].

    template<typename R, bool ANY, int STRIDE>
    FC_D void trav_round(const SceneDev<R>& S, Trav<R>& T, uint32_t* stack, uint32_t& overflow, int descend_min, uint32_t& n_nodes, uint32_t& n_prims)
    {
        uint32_t cur = T.cur;
        int sp = T.sp;
        bool leaf_waiting = __ballot(child_is_leaf(cur) && cur != INVALID_U32) != 0ull;
        while(!child_is_leaf(cur))
        {
            uint32_t next = node_step<R, ANY, STRIDE>(S, T.rc, T.t_max, cur, stack, sp, overflow);
            n_nodes += 1;
            if(next == INVALID_U32 && sp > 0) next = stack[(--sp) * STRIDE];
            cur = next;
            leaf_waiting = leaf_waiting || __ballot(child_is_leaf(cur) && cur != INVALID_U32) != 0ull;
            if(leaf_waiting && __popcll(__ballot(!child_is_leaf(cur))) < descend_min) break;
        }
        if(child_is_leaf(cur) && cur != INVALID_U32)
        {
            const uint32_t first = leaf_first(cur), kind = leaf_kind(cur);
            bool stop = false;
            if(kind == KIND_TRI)
            {
                const uint32_t count = leaf_count(cur);
                n_prims += count;
                for(uint32_t j = 0; j < count; ++j)
                {
                    R t, b0, b1;
                    if(tri_intersect(S.tris[first + j], T.rc, T.t_max, t, b0, b1))
                    {
                        T.t_max = t;
                        T.pk = (KIND_TRI << 30) | (first + j);
                        T.b0 = b0;
                        T.b1 = b1;
                        if(ANY)
                        {
                            stop = true;
                            break;
                        }
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
            cur = (!stop && sp > 0) ? stack[(--sp) * STRIDE] : INVALID_U32;
        }
        T.cur = cur;
        T.sp = sp;
    }

    // Whole-ray convenience wrapper (parity hooks): runs rounds until this lane's ray is finished.
    template<typename R, bool ANY, int STRIDE>
    FC_D bool traverse(const SceneDev<R>& S, const RayCtx<R>& rc, R t_max, uint32_t* stack, R& t_out, uint32_t& pk_out,
                       R& b0_out, R& b1_out, uint32_t& overflow)
    {
        Trav<R> T;
        T.rc = rc;
        T.t_max = t_max;
        T.b0 = R(0);
        T.b1 = R(0);
        T.pk = INVALID_U32;
        T.cur = 0;
        T.sp = 0;
        uint32_t nn = 0, np = 0;
        while(!trav_done(T)) trav_round<R, ANY, STRIDE>(S, T, stack, overflow, 12, nn, np);
        t_out = T.t_max;
        pk_out = T.pk;
        b0_out = T.b0;
        b1_out = T.b1;
        return T.pk != INVALID_U32;
    }

    // Wave-level dynamic work distribution: idle lanes take the next ray indices from the wave's
    // current chunk; an exhausted chunk is replaced with ONE global atomic (WORK_CHUNK rays), so a
    // launch issues n / WORK_CHUNK atomics on the work counter (a single word serves ~88 atomics/us).
    constexpr uint32_t WORK_CHUNK = 256;

    struct WorkCursor
    {
        uint32_t cur, end;  // wave-uniform
        bool dry;           // wave-uniform: the global counter ran past n
    };

    // returns this lane's new ray index or INVALID_U32; `want` = lane is idle
    FC_D uint32_t work_take(WorkCursor& W, bool want, uint32_t* counter, uint32_t n)
    {
        uint32_t mine = INVALID_U32;
        for(;;)
        {
            uint64_t m = __ballot(want && mine == INVALID_U32);
            uint32_t cnt = __popcll(m);
            if(cnt == 0u) break;
            if(W.cur >= W.end)
            {
                if(W.dry) break;
                uint32_t base = 0;
                if((threadIdx.x & 63) == 0) base = atomicAdd(counter, WORK_CHUNK);
                base = __builtin_amdgcn_readfirstlane(base);
                if(base >= n)
                {
                    W.dry = true;
                    break;
                }
                W.cur = base;
                W.end = base + WORK_CHUNK < n ? base + WORK_CHUNK : n;
            }
            uint32_t avail = W.end - W.cur;
            uint32_t rank = __popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull));
            if(want && mine == INVALID_U32 && rank < avail) mine = W.cur + rank;
            W.cur += cnt < avail ? cnt : avail;
        }
        return mine;
    }

    // geometry of a hit: world position and geometric normal, as the reference's raycast_surface_point builds them
    template<typename R>
    FC_D void hit_point(const SceneDev<R>& S, const RayCtx<R>& rc, uint32_t pk, R t, R b0, R b1, vec3<R>& p, vec3<R>& n,
                        uint32_t& prim, uint32_t& entity)
    {
        uint32_t kind = pk >> 30, idx = pk & 0x3FFFFFFFu;
        if(kind == KIND_TRI)
        {
            const TriRec& T = S.tris[idx];
            vec3<R> p0 = ldv<R>(T.v0), p1 = ldv<R>(T.v1), p2 = ldv<R>(T.v2);
            R b2 = R(1) - b0 - b1;
            p = b0 * p0 + b1 * p1 + b2 * p2;              // mesh_surface.hpp:211
            n = normalize(cross(p0 - p2, p1 - p2));       // :212-213, 227
            prim = T.prim;
            entity = T.entity;
        }
        else
        {
            const XPrim<R>& P = S.xprims[idx];
            vec3<R> ol = xf_point(P.inv, rc.o), dl = xf_dir(P.inv, rc.d);
            vec3<R> pl = ol + dl * t;
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
