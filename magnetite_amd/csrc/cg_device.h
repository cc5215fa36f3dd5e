// Device helpers shared by the CG kernels (cg.hip) and the on-chip CG kernel (persist.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace magk {

// 16-byte store, optionally write-through (sc1): a launch that leaves its output dirty in the XCD L2s pays
// the write-back at the kernel boundary (MI355X_MICROARCH.md, price-list row "boundary": + B / 6 TB/s), on the
// critical path of the next launch; write-through moves it under the launch's own streaming.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <bool WT>
__device__ inline void store2(double2 *base, int64_t n_nodes, int64_t idx, double2 v)
{
    if (WT) {
        auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)base, (short)0, (int)(n_nodes * 16), 0x00020000);
        u32x4 d;
        __builtin_memcpy(&d, &v, 16);
        __builtin_amdgcn_raw_buffer_store_b128(d, rsrc, (int)(idx * 16), 0, 16);
    } else {
        base[idx] = v;
    }
}

// 1/a to full fp64 precision from the hardware seed (two Newton steps); the
// IEEE division sequence is ~3x the instructions and this is per (node, element).
__device__ inline double fast_rcp(double a)
{
    double y = __builtin_amdgcn_rcp(a);
    double e = fma(-a, y, 1.0);
    y = fma(y, e, y);
    e = fma(-a, y, 1.0);
    y = fma(y, e, y);
    return y;
}

// One word of the ring table (symbolic.hip, k_ring16): two 16-bit entries -- tile-local id in bits 0-11, bit 15 = no
// triangle between the previous entry and this one, 0xffff = end.  Every other consecutive pair (prev, cur) is the
// triangle (a, prev, cur): one LDS gather per entry instead of two per triangle.
// The walk carries everything RELATIVE to the centre node a (d = c - c_a, u = p - p_a), computed once per neighbour.
template <class V2, class Tri>
__device__ inline void ring_word(uint32_t ww, const V2 *s_xy, const V2 *s_p, const V2 ca, const V2 pa, V2 &pd, V2 &pu,
                                 Tri &&tri)
{
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const uint32_t e = half ? (ww >> 16) : (ww & 0xffffu);
        if (e != 0xffffu) {
            const uint32_t id = e & 0xfffu;
            const V2 cxy = s_xy[id], cp = s_p[id];
            V2 d, u;
            d.x = cxy.x - ca.x;
            d.y = cxy.y - ca.y;
            u.x = cp.x - pa.x;
            u.y = cp.y - pa.y;
            if (!(e & 0x8000u)) tri(pd, pu, d, u);
            pd = d;
            pu = u;
        }
    }
}

// corner_force for the triangle (a, b, c) given relative to a: db = c_b - c_a, ub = p_b - p_a, ...  The B-matrix rows
// sum to zero (beta_a = -(beta_b + beta_c)), so the strain needs only the relative values:
//   2A = db x dc ;  2A eps_x = dc.y ub.x - db.y uc.x ;  2A eps_y = db.x uc.y - dc.x ub.y ;
//   2A gamma = (dc.y ub.y - dc.x ub.x) + (db.x uc.x - db.y uc.y)
// -- a quarter fewer operations than corner_force, and differences of p are formed before they are multiplied.
template <class R>
__device__ inline R ring_rcp(R a);
template <>
__device__ inline double ring_rcp<double>(double a) { return fast_rcp(a); }
template <>
__device__ inline float ring_rcp<float>(float a) { return 1.0f / a; }

template <class V2, class R>
__device__ inline void fan_force(const V2 db, const V2 ub, const V2 dc, const V2 uc, R c0, R nu, R h, R &fx, R &fy)
{
    const R ba = db.y - dc.y, ga = dc.x - db.x;
    const R twoA = db.x * dc.y - dc.x * db.y;
    const R ex = dc.y * ub.x - db.y * uc.x;
    const R ey = db.x * uc.y - dc.x * ub.y;
    const R g = (dc.y * ub.y - dc.x * ub.x) + (db.x * uc.x - db.y * uc.y);
    const R w = c0 * ring_rcp<R>(twoA);
    const R sx = ex + nu * ey, sy = nu * ex + ey, tq = h * g;
    fx += w * (ba * sx + ga * tq);
    fy += w * (ga * sy + ba * tq);
}

} // namespace magk
