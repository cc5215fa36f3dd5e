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
    // (every multiply-add is written as the FMA it is: the rounding of the CG kernels is defined here, not by the compiler's
    // contraction choices -- persist.hip is compiled with -ffp-contract=off)
    const R ba = db.y - dc.y, ga = dc.x - db.x;
    const R twoA = fma(db.x, dc.y, -(dc.x * db.y));
    const R ex = fma(dc.y, ub.x, -(db.y * uc.x));
    const R ey = fma(db.x, uc.y, -(dc.x * ub.y));
    const R g = fma(dc.y, ub.y, -(dc.x * ub.x)) + fma(db.x, uc.x, -(db.y * uc.y));
    const R w = c0 * ring_rcp<R>(twoA);
    const R sx = fma(nu, ey, ex), sy = fma(nu, ex, ey), tq = h * g;
    fx = fma(w, fma(ba, sx, ga * tq), fx);
    fy = fma(w, fma(ga, sy, ba * tq), fy);
}

// Ring walk without per-entry tests.  k_ring16 pads every row to the tile's row length with entries that repeat the
// last neighbour and carry the break bit, so every entry below 2 * nwords is a real LDS slot: the walk is a chain of
// unconditional gathers, the loop bound is a scalar (tile-uniform) branch, and only the ADDITION of a triangle's force
// is selected by the break bit (a repeated entry spans no area: its force is NaN/inf and is selected out, never
// multiplied in).  With two waves per SIMD the exec-mask bookkeeping of the branchy walk is pure issue-slot cost.
// NU: entries walked as ONE straight-line block when the tile's rows have at least that many (a closed fan of valence 6
// is 7 entries: the rule on the benchmark meshes).  A scalar test per entry makes every entry its own basic block, and the
// instruction scheduler works per block: the chain  2A -> v_rcp_f64 -> two Newton steps -> weight -> force  of one
// entry (~12 dependent fp64 operations) then cannot overlap the next entry's, and in-kernel stamps showed a wave stalled
// on its own dependencies half of the time even when it wins every arbitration (profiles/r03_persist_phases.json).  In
// one block the gathers of all NU entries issue up front and the NU - 1 chains interleave.
// A second block, entries NU .. NU2 - 1, is walked the same way when the rows have at least NU2 entries (ROCm 7.2's
// iterative-ilp scheduler, the fastest for this kernel, crashes on blocks of more than four triangles).
// IDMASK / toff: the on-chip kernel rewrites the entries in its registers to WORKGROUP-wide LDS slots (15 bits); entries of
// longer rows still come from the table in memory, tile-local, and are moved by the tile's slot offset `toff`.
template <int NW, int NU = 1, int NU2 = NU, uint32_t IDMASK = 0xfffu>
__device__ inline void ring_walk_uniform(const uint32_t (&w)[NW], const uint32_t *more, int32_t stride, int32_t nent,
                                         const double2 *s_xy, const double2 *s_p, const double2 ca, const double2 pa,
                                         double c0, double nu, double h, double &fx, double &fy, uint32_t toff = 0)
{
    double2 pd, pu;
    auto step = [&](uint32_t e, bool seed) {
        const uint32_t id = e & IDMASK;
        const double2 cxy = s_xy[id], cp = s_p[id];
        const double2 d = make_double2(cxy.x - ca.x, cxy.y - ca.y), u = make_double2(cp.x - pa.x, cp.y - pa.y);
        if (!seed) {
            double dfx = 0.0, dfy = 0.0;
            fan_force<double2, double>(pd, pu, d, u, c0, nu, h, dfx, dfy);
            const bool closes = !(e & 0x8000u);
            fx += closes ? dfx : 0.0;
            fy += closes ? dfy : 0.0;
        }
        pd = d;
        pu = u;
    };
    auto entry = [&](int k) { return (k & 1) ? (w[k >> 1] >> 16) : (w[k >> 1] & 0xffffu); };
    step(w[0] & 0xffffu, true);
    if (NU > 1 && nent >= NU) { // nent: a scalar
#pragma unroll
        for (int k = 1; k < NU; ++k) step(entry(k), false);
    } else {
#pragma unroll
        for (int k = 1; k < NU; ++k)
            if (k < nent) step(entry(k), false);
    }
    if (NU2 > NU && nent >= NU2) {
#pragma unroll
        for (int k = NU; k < NU2; ++k) step(entry(k), false);
    } else {
#pragma unroll
        for (int k = NU; k < NU2; ++k)
            if (k < nent) step(entry(k), false);
    }
#pragma unroll
    for (int k = NU2; k < 2 * NW; ++k)
        if (k < nent) step(entry(k), false);
    for (int32_t k = 2 * NW; k < nent; ++k) {
        const uint32_t ww = more[(int64_t)(k >> 1) * stride];
        const uint32_t e = (k & 1) ? (ww >> 16) : (ww & 0xffffu);
        step(((e & 0xfffu) + toff) | (e & 0x8000u), false);
    }
}

// The same walk with the triangles' weights c0 / (2A) held in registers (the on-chip CG kernel): 2A depends on the
// coordinates only, so the cross product, the v_rcp_f64, its two Newton steps and the scaling -- 8 of the ~48
// instructions of a ring step, the only quarter-rate one among them -- are paid once per solve instead of once per
// iteration, by ring_weights() below with the very operations ring_walk_uniform uses (same bits).  The break bit is
// folded into the weight (0 for an entry that closes no triangle: its bracket is finite, 0 x finite adds nothing), so
// the per-entry selects go as well.  NC weights per node (a closed fan of valence 6 has 6 triangles); triangles beyond
// NC are evaluated as before.
template <int NW, int NC, uint32_t IDMASK = 0xfffu>
__device__ inline void ring_weights(const uint32_t (&w)[NW], int32_t nent, const double2 *s_xy, const double2 ca, double c0,
                                    double (&wgt)[NC])
{
    double2 pd;
    auto entry = [&](int k) { return (k & 1) ? (w[k >> 1] >> 16) : (w[k >> 1] & 0xffffu); };
    {
        const double2 cxy = s_xy[entry(0) & IDMASK];
        pd = make_double2(cxy.x - ca.x, cxy.y - ca.y);
    }
#pragma unroll
    for (int k = 1; k <= NC; ++k) {
        wgt[k - 1] = 0.0;
        if (k < nent && k < 2 * NW) {
            const uint32_t e = entry(k);
            const double2 cxy = s_xy[e & IDMASK];
            const double2 d = make_double2(cxy.x - ca.x, cxy.y - ca.y);
            const double twoA = fma(pd.x, d.y, -(d.x * pd.y));
            if (!(e & 0x8000u)) wgt[k - 1] = c0 * fast_rcp(twoA);
            pd = d;
        }
    }
}

template <class V2, class R>
__device__ inline void fan_force_w(const V2 db, const V2 ub, const V2 dc, const V2 uc, R wt, R nu, R h, R &fx, R &fy)
{
    const R ba = db.y - dc.y, ga = dc.x - db.x;
    const R ex = fma(dc.y, ub.x, -(db.y * uc.x));
    const R ey = fma(db.x, uc.y, -(dc.x * ub.y));
    const R g = fma(dc.y, ub.y, -(dc.x * ub.x)) + fma(db.x, uc.x, -(db.y * uc.y));
    const R sx = fma(nu, ey, ex), sy = fma(nu, ex, ey), tq = h * g;
    fx = fma(wt, fma(ba, sx, ga * tq), fx);
    fy = fma(wt, fma(ga, sy, ba * tq), fy);
}

template <int NW, int NC, uint32_t IDMASK = 0xfffu>
__device__ inline void ring_walk_cached(const uint32_t (&w)[NW], const uint32_t *more, int32_t stride, int32_t nent,
                                        const double2 *s_xy, const double2 *s_p, const double2 ca, const double2 pa,
                                        double c0, double nu, double h, const double (&wgt)[NC], double &fx, double &fy,
                                        uint32_t toff = 0)
{
    double2 pd, pu;
    auto entry = [&](int k) { return (k & 1) ? (w[k >> 1] >> 16) : (w[k >> 1] & 0xffffu); };
    auto gather = [&](uint32_t e, double2 &d, double2 &u) {
        const uint32_t id = e & IDMASK;
        const double2 cxy = s_xy[id], cp = s_p[id];
        d = make_double2(cxy.x - ca.x, cxy.y - ca.y);
        u = make_double2(cp.x - pa.x, cp.y - pa.y);
    };
    gather(entry(0), pd, pu);
#pragma unroll
    for (int k = 1; k < 2 * NW; ++k)
        if (k < nent) { // nent: a scalar
            const uint32_t e = entry(k);
            double2 d, u;
            gather(e, d, u);
            if (k <= NC) {
                fan_force_w<double2, double>(pd, pu, d, u, wgt[k - 1], nu, h, fx, fy);
            } else {
                double dfx = 0.0, dfy = 0.0;
                fan_force<double2, double>(pd, pu, d, u, c0, nu, h, dfx, dfy);
                const bool closes = !(e & 0x8000u);
                fx += closes ? dfx : 0.0;
                fy += closes ? dfy : 0.0;
            }
            pd = d;
            pu = u;
        }
    for (int32_t k = 2 * NW; k < nent; ++k) {
        const uint32_t ww = more[(int64_t)(k >> 1) * stride];
        const uint32_t e0 = (k & 1) ? (ww >> 16) : (ww & 0xffffu);
        const uint32_t e = ((e0 & 0xfffu) + toff) | (e0 & 0x8000u);
        double2 d, u;
        gather(e, d, u);
        double dfx = 0.0, dfy = 0.0;
        fan_force<double2, double>(pd, pu, d, u, c0, nu, h, dfx, dfy);
        const bool closes = !(e & 0x8000u);
        fx += closes ? dfx : 0.0;
        fy += closes ? dfy : 0.0;
        pd = d;
        pu = u;
    }
}

// ---- edge blocks (the on-chip CG kernel, round 3) ----
// A triangle's force on its corner a is linear in the two other corners' RELATIVE values: f_a = K_ab u_b + K_ac u_c, and the
// 2 x 2 blocks depend on the coordinates only.  Folded per ring ENTRY -- entry j takes K_ab of the triangle after it and
// K_ac of the triangle before it -- a node's triangles become blocks, and a ring step is one LDS gather of p, two
// subtractions and four FMAs instead of the ~23 fp64 operations and two gathers of fan_force_w.  NB block entries per
// node: the NB - 1 triangles between them, and triangle NB as well when it CLOSES the fan onto entry 0 (a closed fan of
// valence NB -- the rule on structured meshes -- is then NB blocks and nothing else).  Rows this does not cover (several
// fans at a node, more entries) never get here: k_ring16 flags the mesh and it keeps the triangle walk.
// Only the SYMMETRIC part of a block is kept (three numbers): the antisymmetric part of one triangle's contribution is
//     -kappa J for the entry before it, +kappa J for the entry after it,   kappa = (h - nu) c0 / 2,  J = [[0, 1], [-1, 0]]
// whatever its shape (K12 - K21 = (h - nu) w (ba dc.x + ga dc.y) = -(h - nu) w 2A, and w = c0 / 2A), so over a fan the
// antisymmetric parts telescope to kappa J (u_last - u_first): nothing for a closed fan, two subtractions and two FMAs per
// NODE otherwise (tests/test_edge_block_algebra.py restates this in numpy).
// The blocks are filled by applying fan_force_w, with the weight ring_weights computes, to unit vectors: the same
// arithmetic as the triangle walk up to the order of the additions.
// XY: coordinates of a ring entry's node, by its (masked) id
template <int NW, int NB, uint32_t IDMASK = 0xfffu, class XY>
__device__ inline void ring_blocks(const uint32_t (&w)[NW], int32_t nent, XY &&xy_of, const double2 ca, double c0, double nu,
                                   double h, double (&kb)[3 * NB])
{
    static_assert(NB + 1 <= 2 * NW, "block entries come from the ring words held in registers");
    auto entry = [&](int k) { return (k & 1) ? (w[k >> 1] >> 16) : (w[k >> 1] & 0xffffu); };
#pragma unroll
    for (int i = 0; i < 3 * NB; ++i) kb[i] = 0.0;
    double2 pd;
    {
        const double2 cxy = xy_of(entry(0) & IDMASK);
        pd = make_double2(cxy.x - ca.x, cxy.y - ca.y);
    }
    const double2 z = make_double2(0.0, 0.0), ex = make_double2(1.0, 0.0), ey = make_double2(0.0, 1.0);
#pragma unroll
    for (int k = 1; k <= NB; ++k)
        if (k < nent) {
            const uint32_t e = entry(k);
            const double2 cxy = xy_of(e & IDMASK);
            const double2 d = make_double2(cxy.x - ca.x, cxy.y - ca.y);
            const bool closes = !(e & 0x8000u);
            const bool fold = k < NB || (e & IDMASK) == (entry(0) & IDMASK); // triangle NB only onto entry 0
            if (closes && fold) {
                const double twoA = fma(pd.x, d.y, -(d.x * pd.y));
                const double wt = c0 * fast_rcp(twoA);
                double b00 = 0.0, b10 = 0.0, b01 = 0.0, b11 = 0.0, c00 = 0.0, c10 = 0.0, c01 = 0.0, c11 = 0.0;
                fan_force_w<double2, double>(pd, ex, d, z, wt, nu, h, b00, b10); // K_ab, first column
                fan_force_w<double2, double>(pd, ey, d, z, wt, nu, h, b01, b11);
                fan_force_w<double2, double>(pd, z, d, ex, wt, nu, h, c00, c10); // K_ac
                fan_force_w<double2, double>(pd, z, d, ey, wt, nu, h, c01, c11);
                const int jc = k < NB ? k : 0; // (k == NB: the closing entry is entry 0)
                kb[3 * (k - 1) + 0] += b00;
                kb[3 * (k - 1) + 1] += 0.5 * (b01 + b10);
                kb[3 * (k - 1) + 2] += b11;
                kb[3 * jc + 0] += c00;
                kb[3 * jc + 1] += 0.5 * (c01 + c10);
                kb[3 * jc + 2] += c11;
            }
            pd = d;
        }
}

// Block entries the tile's rows do not reach repeat the last one (as padding does inside the rows, with a zero block):
// ring_walk_blocks then gathers all NB entries without a test -- every one is a slot somebody initialised.
template <int NW, int NB, uint32_t IDMASK = 0xfffu>
__device__ inline void ring_pad_entries(uint32_t (&w)[NW], int32_t nent)
{
    auto entry = [&](int k) { return (k & 1) ? (w[k >> 1] >> 16) : (w[k >> 1] & 0xffffu); };
    if (nent > 0 && nent < NB) {
        const uint32_t last = (entry(nent - 1) & IDMASK) | 0x8000u;
#pragma unroll
        for (int k = 1; k < NB; ++k)
            if (k >= nent) w[k >> 1] = (k & 1) ? ((w[k >> 1] & 0xffffu) | (last << 16)) : ((w[k >> 1] & 0xffff0000u) | last);
    }
}

// The walk over a node's NB blocks: all NB gathers of p in one straight-line block (ring_pad_entries made every entry a
// valid slot; a zero block adds nothing), one FMA per term.  folded: the fan closes inside the blocks (nothing to telescope).
template <int NW, int NB, uint32_t IDMASK = 0xfffu>
__device__ inline void ring_walk_blocks(const uint32_t (&w)[NW], const double2 *s_p, const double2 pa, double kappa,
                                        bool folded, const double (&kb)[3 * NB], double &fx, double &fy)
{
    auto entry = [&](int k) { return (k & 1) ? (w[k >> 1] >> 16) : (w[k >> 1] & 0xffffu); };
    double2 u0, ul;
    {
        const double2 cp = s_p[entry(0) & IDMASK];
        u0 = make_double2(cp.x - pa.x, cp.y - pa.y);
        fx = fma(kb[1], u0.y, fma(kb[0], u0.x, fx)); // one FMA per term (a sum of two products first costs a third operation)
        fy = fma(kb[2], u0.y, fma(kb[1], u0.x, fy));
        ul = u0;
    }
#pragma unroll
    for (int k = 1; k < NB; ++k) {
        const double2 cp = s_p[entry(k) & IDMASK];
        const double2 u = make_double2(cp.x - pa.x, cp.y - pa.y);
        fx = fma(kb[3 * k + 1], u.y, fma(kb[3 * k], u.x, fx));
        fy = fma(kb[3 * k + 2], u.y, fma(kb[3 * k + 1], u.x, fy));
        ul = u;
    }
    const double kap = folded ? 0.0 : kappa; // a fan closed inside the blocks: its antisymmetric parts cancel
    fx = fma(kap, ul.y - u0.y, fx);          // otherwise they telescope to kappa J (u_last - u_first)
    fy = fma(-kap, ul.x - u0.x, fy);
}

// The same walk for rows with MORE than NB blocks (on-chip kernel, EBM == 2: gmsh-type meshes, where a quarter of the nodes
// have seven neighbours).  The blocks beyond the registers are 32-byte records {k11, k12 | k22, slot} in the workgroup's
// LDS pool: this lane's node owns records off .. off + cnt - 1; nmax = the most records any lane of this wave has for this
// node slot (a scalar: the trip count).  No lane sits out: a lane with fewer records reads record 0 of the pool, a zero
// block on a valid slot, so a step is straight-line code -- two record reads, one gather, six fp64 operations -- and two
// steps go together (their four record reads issue at once, then the two gathers).  u_first and u_last of an OPEN fan are
// always register entries (k_edge_blocks_ovf puts an open row's last entry into block NB - 1 and its middle ones into
// the pool), so the telescoped antisymmetric part needs nothing from the pool.  Same sums as the short rows', block after
// block.
template <int NW, int NB, uint32_t IDMASK = 0xfffu>
__device__ inline void ring_walk_blocks_ovf(const uint32_t (&w)[NW], const double2 *s_p, const double2 pa, double kappa,
                                            bool folded, const double (&kb)[3 * NB], const double2 *pool, uint32_t off,
                                            uint32_t cnt, int32_t nmax, double &fx, double &fy)
{
    auto entry = [&](int k) { return (k & 1) ? (w[k >> 1] >> 16) : (w[k >> 1] & 0xffffu); };
    double2 u0, ul;
    {
        const double2 cp = s_p[entry(0) & IDMASK];
        u0 = make_double2(cp.x - pa.x, cp.y - pa.y);
        fx = fma(kb[1], u0.y, fma(kb[0], u0.x, fx));
        fy = fma(kb[2], u0.y, fma(kb[1], u0.x, fy));
        ul = u0;
    }
#pragma unroll
    for (int k = 1; k < NB; ++k) {
        const double2 cp = s_p[entry(k) & IDMASK];
        const double2 u = make_double2(cp.x - pa.x, cp.y - pa.y);
        fx = fma(kb[3 * k + 1], u.y, fma(kb[3 * k], u.x, fx));
        fy = fma(kb[3 * k + 2], u.y, fma(kb[3 * k + 1], u.x, fy));
        ul = u;
    }
    const double kap = folded ? 0.0 : kappa;
    fx = fma(kap, ul.y - u0.y, fx);
    fy = fma(-kap, ul.x - u0.x, fy);
    auto rec_of = [&](int32_t k) { return 2u * ((uint32_t)k < cnt ? off + (uint32_t)k : 0u); };
    auto apply = [&](const double2 a, const double2 b) {
        const double2 cp = s_p[(uint32_t)__double2loint(b.y)];
        const double2 u = make_double2(cp.x - pa.x, cp.y - pa.y);
        fx = fma(a.y, u.y, fma(a.x, u.x, fx));
        fy = fma(b.x, u.y, fma(a.y, u.x, fy));
    };
    // nmax is a scalar: one record (most waves of a frontal mesh: some lane has a seventh neighbour), two together (some lane
    // has an eighth), the rare rest one at a time -- not unrolled: an unrolled loop here takes its registers from the blocks
#ifndef MAG_PERSIST_OVF_PAIR
#define MAG_PERSIST_OVF_PAIR 0 // 1: the first two records together (four record reads in flight) -- measured slower, 7.51 against 7.10 us per iteration on the 1M frontal mesh: the pair takes six block registers to scratch, and their reload waits for the node slot's granule stores
#endif
#if MAG_PERSIST_OVF_PAIR
    if (nmax == 1) {
        const uint32_t i0 = rec_of(0);
        const double2 a0 = pool[i0], b0 = pool[i0 + 1];
        apply(a0, b0);
    } else if (nmax >= 2) {
        const uint32_t i0 = rec_of(0), i1 = rec_of(1);
        const double2 a0 = pool[i0], b0 = pool[i0 + 1], a1 = pool[i1], b1 = pool[i1 + 1];
        apply(a0, b0);
        apply(a1, b1);
#pragma clang loop unroll(disable)
        for (int32_t k = 2; k < nmax; ++k) {
            const uint32_t i2 = rec_of(k);
            const double2 a2 = pool[i2], b2 = pool[i2 + 1];
            apply(a2, b2);
        }
    }
#else
    // MAG_PERSIST_OVF_UNCOND records are taken by EVERY wave, in the same basic block as the register blocks (their record
    // reads issue alongside the six gathers; a wave none of whose lanes has that many reads the zero record): on a frontal
    // mesh nearly every wave has a lane with a seventh neighbour.  The rest in a loop that is not unrolled.
#ifndef MAG_PERSIST_OVF_UNCOND
#define MAG_PERSIST_OVF_UNCOND 0
#endif
#pragma unroll
    for (int32_t k = 0; k < MAG_PERSIST_OVF_UNCOND; ++k) {
        const uint32_t i2 = rec_of(k);
        const double2 a2 = pool[i2], b2 = pool[i2 + 1];
        apply(a2, b2);
    }
#pragma clang loop unroll(disable)
    for (int32_t k = MAG_PERSIST_OVF_UNCOND; k < nmax; ++k) {
        const uint32_t i2 = rec_of(k);
        const double2 a2 = pool[i2], b2 = pool[i2 + 1];
        apply(a2, b2);
    }
#endif
}

} // namespace magk
