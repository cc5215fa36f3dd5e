// Matrix-free element-loop operator and the fused CG iteration for gfx950.
//
// Replaces the per-iteration work of solver.rs:31-36 (CSR SpMV) and of argmin's
// ConjugateGradient::next_iter (solver.rs:142-157; recurrences in SURVEY 3.3):
//     q = A p; alpha = rtr/(p.q); x += alpha p; r += alpha q;
//     rtr' = r.r; beta = rtr'/rtr; p = -r + beta p; cost = f(rtr')
// with r = A x - b (argmin's sign).  A = M K M acts on full-length vectors
// (M zeroes prescribed-displacement DOFs), i.e. K_ff embedded in 2N.
//
// Layout: nodes are in Hilbert order; workgroup tile t owns nodes
// [t*B, (t+1)*B).  One thread owns one node (both DOFs, double2).  The tile's
// coordinates and p are staged in LDS; every incident element of a node is
// visited through the tile's ELL table (slot-major => coalesced int2 reads),
// and the thread accumulates only ITS corner's force  f_a = t*A * B_a^T D B u_e
// -- owner-computes, so there is no scatter, no atomic and no colouring pass,
// and the summation order per node is fixed (ascending element index).
// Corners owned by another tile (the tile's halo) are staged in LDS next to the
// owned nodes (k_operator_lds, k_cg_fused*) or, when a tile's halo would not
// fit, read from global memory (k_operator, the gather fallback); their p is
// recomputed from their previous state, which is stable during the launch.
//
// Kernels in this file, in the order they appear:
//   k_operator / k_operator_lds   y = [M] K [M] v, or (CG mode) the operator launch of the two-launch iteration
//   k_update                      r += alpha q, r.r          (two-launch iteration, cg_variant 0)
//   k_csr_*                       the reference's literal iteration on K_ff in CSR (MAG_OP_CSR)
//   k_cg_fused / k_cg_fused_dma   ONE launch per CG iteration (cg_variant 1, default; _dma = LDS-DMA staging)
//   k_cg_fused32                  the same in fp32 (BASELINE config 5's sweep)
// Dot products are reduced in a fixed order from <= kMaxGrid per-workgroup
// partials by every workgroup of the NEXT launch (the kernel boundary is the
// grid-wide sync), so results are bitwise reproducible run to run.
#include <cstdlib>
#include <cstring>

#include <hip/hip_runtime.h>

#include "cg_device.h"
#include "kernels.h"

namespace magk {

int cg_grid(int32_t T) { return T < kMaxGrid ? (T < 1 ? 1 : T) : kMaxGrid; }

// ------------------------------------------------------------ reductions ---
template <int B>
__device__ inline double block_sum(double v, double *s_red)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < B / 64; ++i) t += s_red[i];
    return t; // identical in every thread
}

template <int B>
__device__ inline double sum_partials(const double *part, int n, double *s_red)
{
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += B) v += part[i];
    return block_sum<B>(v, s_red);
}

// Force on corner a of the CST (a, b, c) for nodal values pa, pb, pc:
//   2A eps = (sum beta_i u_i, sum gamma_i v_i, sum gamma_i u_i + beta_i v_i)     solver.rs:213-225
//   sigma  = D eps                                                                solver.rs:241-247
//   f_a    = t A B_a^T sigma = c0/(2A) * [beta_a sx' + gamma_a t', gamma_a sy' + beta_a t']
// with c0 = E t / (2 (1 - nu^2)) and primes denoting the unscaled (2A eps) quantities.
// 2A is taken from coordinate differences (signed, like solver.rs:192).
__device__ inline void corner_force(const double2 ca, const double2 pa, const double2 cb, const double2 pb,
                                    const double2 cc, const double2 pc, double c0, double nu, double h, double &fx,
                                    double &fy)
{
    const double ba = cb.y - cc.y, bb = cc.y - ca.y, bc = ca.y - cb.y;
    const double ga = cc.x - cb.x, gb = ca.x - cc.x, gc = cb.x - ca.x;
    // (explicit FMAs: this file is compiled with -ffp-contract=off since round 4 -- the rounding of the matrix-free kernels is
    // defined here, not by the compiler; the kernels that restate the reference's iteration to the letter, the two-launch
    // variant and the CSR operator, keep their separate multiplications and additions: Rust never fuses)
    const double twoA = fma(gc, bb, -(gb * bc));
    const double ex = fma(bc, pc.x, fma(bb, pb.x, ba * pa.x));
    const double ey = fma(gc, pc.y, fma(gb, pb.y, ga * pa.y));
    const double g = fma(bc, pc.y, fma(gc, pc.x, fma(bb, pb.y, fma(gb, pb.x, fma(ba, pa.y, ga * pa.x)))));
    const double w = c0 * fast_rcp(twoA);
    const double sx = fma(nu, ey, ex), sy = fma(nu, ex, ey), tq = h * g;
    fx = fma(w, fma(ba, sx, ga * tq), fx);
    fy = fma(w, fma(ga, sy, ba * tq), fy);
}

// State machine of the two-launch iteration (operator launch): judge iterate k from its exact r.r, record the verdict,
// derive beta.  Returns 0 to go on, 1 when the solve is over (x still lacks alpha_{k-1} p_{k-1}: the caller applies it
// unless `broke`).
__device__ inline int cg_step(CgState *st, long long k, double rr, double rrh0, double rrh1, double *hist, int hist_len,
                              double &beta, bool &broke)
{
    const double cost = st->stop_mode == 1 ? fabs(rr) : sqrt(rr);
    const bool finished = (k >= 1) && (cost <= st->target);
    broke = !(fabs(rr) <= 1.79769313486231570e308); // NaN or inf
    const bool maxed = k >= st->max_iter;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (k >= 1 && k - 1 < hist_len) hist[k - 1] = cost;
        if (k >= 1 && cost < st->best_cost) { // argmin's state.update(): best_param follows the lowest cost
            st->best_cost = cost;
            st->best_iter = k;
        }
        if (finished || broke || maxed) {
            st->iterations = k;
            st->final_cost = cost;
            st->converged = finished ? 1 : 0;
            st->breakdown = broke ? 1 : 0;
            st->done = 1;
        } else {
            st->rr_hist[k & 1] = rr;
            st->iterB = k;
        }
    }
    if (finished || broke || maxed) return 1;
    const double rr_prev = (k == 0) ? rr : ((k & 1) ? rrh0 : rrh1);
    beta = rr / rr_prev;
    return 0;
}

// ------------------------------------------------------ operator kernel ---
template <int B, bool CG>
__global__ void __launch_bounds__(B) k_operator(const OpParams P)
{
    __shared__ double2 s_xy[B];
    __shared__ double2 s_p[B];
    __shared__ double s_red[B / 64];

    double beta = 0.0, alpha = 0.0;
    if (CG) {
        CgState *st = P.st;
        const long long k = st->iterA;
        const int was_done = st->done;
        alpha = st->alpha_last;
        const double rrh0 = st->rr_hist[0], rrh1 = st->rr_hist[1];
        const double rr = sum_partials<B>(P.partRR, P.nPart, s_red);
        if (was_done) return;
        bool broke = false;
        if (cg_step(st, k, rr, rrh0, rrh1, P.hist, P.hist_len, beta, broke)) {
            if (!broke) {
                for (int32_t t = P.t0 + blockIdx.x; t < P.t1; t += gridDim.x) {
                    const int64_t nd = (int64_t)t * B + threadIdx.x;
                    if (nd < P.N) {
                        const double2 pp = P.pprev[nd];
                        double2 xx = P.x[nd];
                        xx.x += alpha * pp.x;
                        xx.y += alpha * pp.y;
                        P.x[nd] = xx;
                    }
                }
            }
            return;
        }
    }

    const double c0 = P.c0, nu = P.nu, h = P.h;
    double acc = 0.0;
    for (int32_t t = P.t0 + blockIdx.x; t < P.t1; t += gridDim.x) {
        const int64_t base = (int64_t)t * B;
        const int64_t node = base + threadIdx.x;
        const bool valid = node < P.N;
        double2 ca = make_double2(0.0, 0.0), pa = make_double2(0.0, 0.0);
        uint8_t m = 3;
        if (valid) {
            ca = P.xyP[node];
            m = P.maskP[node];
            if (CG) {
                const double2 r2 = P.r[node], pp = P.pprev[node];
                pa.x = -r2.x + beta * pp.x;
                pa.y = -r2.y + beta * pp.y;
                P.pnew[node] = pa;
                double2 xx = P.x[node];
                xx.x += alpha * pp.x;
                xx.y += alpha * pp.y;
                P.x[node] = xx;
            } else {
                pa = P.v[node];
                if (P.masked) {
                    if (m & 1) pa.x = 0.0;
                    if (m & 2) pa.y = 0.0;
                }
            }
        }
        __syncthreads(); // previous tile's readers are done with the LDS images
        s_xy[threadIdx.x] = ca;
        s_p[threadIdx.x] = pa;
        __syncthreads();

        const int32_t deg = P.tile_deg[t];
        const int2 *ell = P.ell + P.tile_off[t] + threadIdx.x;
        double fx = 0.0, fy = 0.0;
        for (int32_t k = 0; k < deg; ++k) {
            const int2 bc = ell[(int64_t)k * B];
            if (bc.x < 0) continue;
            double2 cb, pb, cc, pc;
            const uint32_t lb = (uint32_t)(bc.x - (int32_t)base), lc = (uint32_t)(bc.y - (int32_t)base);
            if (lb < (uint32_t)B) {
                cb = s_xy[lb];
                pb = s_p[lb];
            } else {
                cb = P.xyP[bc.x];
                if (CG) {
                    const double2 r2 = P.r[bc.x], pp = P.pprev[bc.x];
                    pb.x = -r2.x + beta * pp.x;
                    pb.y = -r2.y + beta * pp.y;
                } else {
                    pb = P.v[bc.x];
                    if (P.masked) {
                        const uint8_t mb = P.maskP[bc.x];
                        if (mb & 1) pb.x = 0.0;
                        if (mb & 2) pb.y = 0.0;
                    }
                }
            }
            if (lc < (uint32_t)B) {
                cc = s_xy[lc];
                pc = s_p[lc];
            } else {
                cc = P.xyP[bc.y];
                if (CG) {
                    const double2 r2 = P.r[bc.y], pp = P.pprev[bc.y];
                    pc.x = -r2.x + beta * pp.x;
                    pc.y = -r2.y + beta * pp.y;
                } else {
                    pc = P.v[bc.y];
                    if (P.masked) {
                        const uint8_t mc = P.maskP[bc.y];
                        if (mc & 1) pc.x = 0.0;
                        if (mc & 2) pc.y = 0.0;
                    }
                }
            }
            corner_force(ca, pa, cb, pb, cc, pc, c0, nu, h, fx, fy);
        }
        if (valid) {
            if (CG || P.masked) {
                if (m & 1) fx = 0.0;
                if (m & 2) fy = 0.0;
            }
            if (CG) {
                P.q[node] = make_double2(fx, fy);
                acc += pa.x * fx + pa.y * fy;
            } else {
                P.y[node] = make_double2(fx, fy);
            }
        }
    }
    if (CG) {
        for (int32_t k = blockIdx.x * B + threadIdx.x; k < P.n_iface; k += gridDim.x * B) {
            const int32_t g = P.iface[k];
            if (g < P.own0 || g >= P.own1) {
                const double2 r2 = P.r[g], pp = P.pprev[g];
                P.pnew[g] = make_double2(-r2.x + beta * pp.x, -r2.y + beta * pp.y);
            }
        }
        const double tot = block_sum<B>(acc, s_red);
        if (threadIdx.x == 0) P.partPQ[blockIdx.x] = tot;
    }
}

// ----------------------------------------- operator kernel, LDS halo ---
// Same operator on the tile-local table (fill_ell16): the tile's halo nodes are staged in LDS next to
// its owned nodes, so the incident-element loop is LDS + ALU only and every global access of the launch
// is issued up front.  Slot words of the first 8 incident elements are prefetched into registers.
constexpr int kSlotRegs = 5; // ring words kept in registers: 10 entries, a closed fan of valence <= 9

template <int B, bool CG, bool WT>
__global__ void __launch_bounds__(B) k_operator_lds(const OpParams P)
{
    extern __shared__ __attribute__((aligned(16))) double2 smem[];
    double2 *s_xy = smem;
    double2 *s_p = smem + P.cap;
    double *s_red = (double *)(smem + 2 * P.cap);

    const int tid = threadIdx.x;
    // registers of the tile in flight
    int64_t node = 0;
    bool valid = false, hvalid = false;
    double2 ca, a0, a1, xo, hc, h0, h1;
    uint8_t m = 3, hm = 0;
    int32_t deg = 0, nh = 0, hoff = 0;
    uint32_t w[kSlotRegs];
    const uint32_t *ell = nullptr;

    auto load_tile = [&](int32_t t) {
        node = (int64_t)t * B + tid;
        valid = node < P.N;
        ca = a0 = a1 = xo = make_double2(0.0, 0.0);
        m = 3;
        if (valid) {
            ca = P.xyP[node];
            m = P.maskP[node];
            if (CG) {
                a0 = P.r[node];
                a1 = P.pprev[node];
                xo = P.x[node];
            } else {
                a0 = P.v[node];
            }
        }
        deg = P.tile_deg[t];
        ell = P.ell16 + P.tile_off[t] + tid;
#pragma unroll
        for (int k = 0; k < kSlotRegs; ++k) w[k] = k < deg ? ell[(int64_t)k * B] : 0xffffffffu;
        hoff = P.tile_hoff[t];
        nh = P.tile_hoff[t + 1] - hoff;
        hvalid = tid < nh;
        hc = h0 = h1 = make_double2(0.0, 0.0);
        hm = 0;
        if (hvalid) {
            const int32_t g = P.halo_g[hoff + tid];
            hc = P.halo_xy[hoff + tid]; // static per-tile copy: coalesced, independent of the index load
            if (CG) {
                h0 = P.r[g];
                h1 = P.pprev[g];
            } else {
                h0 = P.v[g];
                hm = P.maskP[g];
            }
        }
    };

    // the whole CG state line is read before anything waits, together with the first tile's operands
    CgState *st = P.st;
    long long k = 0;
    int was_done = 0;
    double rrh0 = 0.0, rrh1 = 0.0, alpha = 0.0;
    if (CG) {
        k = st->iterA;
        was_done = st->done;
        rrh0 = st->rr_hist[0];
        rrh1 = st->rr_hist[1];
        alpha = st->alpha_last;
    }
    load_tile(P.t0 + blockIdx.x);

    double beta = 0.0;
    if (CG) {
        const double rr = sum_partials<B>(P.partRR, P.nPart, s_red);
        if (was_done) return;
        bool broke = false;
        if (cg_step(st, k, rr, rrh0, rrh1, P.hist, P.hist_len, beta, broke)) {
            // x still lacks the last step alpha_{k-1} p_{k-1} (the update is folded into this kernel)
            if (!broke) {
                for (int32_t t = P.t0 + blockIdx.x; t < P.t1; t += gridDim.x) {
                    const int64_t nd = (int64_t)t * B + tid;
                    if (nd < P.N) {
                        const double2 pp = P.pprev[nd];
                        double2 xx = P.x[nd];
                        xx.x += alpha * pp.x;
                        xx.y += alpha * pp.y;
                        P.x[nd] = xx;
                    }
                }
            }
            return;
        }
    }

    const double c0 = P.c0, nu = P.nu, h = P.h;
    double acc = 0.0;
    int32_t t = P.t0 + blockIdx.x;
    for (;;) {
        double2 pa;
        if (CG) {
            pa.x = -a0.x + beta * a1.x;
            pa.y = -a0.y + beta * a1.y;
            if (valid) {
                store2<WT>(P.pnew, P.N, node, pa);
                // x += alpha_{k-1} p_{k-1}: argmin's param.scaled_add, one launch late, p_{k-1} is already here
                xo.x += alpha * a1.x;
                xo.y += alpha * a1.y;
                store2<WT>(P.x, P.N, node, xo);
            }
        } else {
            pa = a0;
            if (P.masked) {
                if (m & 1) pa.x = 0.0;
                if (m & 2) pa.y = 0.0;
            }
        }
        __syncthreads(); // previous tile's readers are done with the LDS images
        s_xy[tid] = ca;
        s_p[tid] = pa;
        if (hvalid) {
            double2 hp;
            if (CG) {
                hp.x = -h0.x + beta * h1.x;
                hp.y = -h0.y + beta * h1.y;
            } else {
                hp = h0;
                if (P.masked) {
                    if (hm & 1) hp.x = 0.0;
                    if (hm & 2) hp.y = 0.0;
                }
            }
            s_xy[B + tid] = hc;
            s_p[B + tid] = hp;
        }
        for (int32_t hh = tid + B; hh < nh; hh += B) { // tiles with more halo nodes than threads (rare)
            const int32_t g = P.halo_g[hoff + hh];
            double2 hp;
            if (CG) {
                const double2 r2 = P.r[g], pp = P.pprev[g];
                hp.x = -r2.x + beta * pp.x;
                hp.y = -r2.y + beta * pp.y;
            } else {
                hp = P.v[g];
                if (P.masked) {
                    const uint8_t mm = P.maskP[g];
                    if (mm & 1) hp.x = 0.0;
                    if (mm & 2) hp.y = 0.0;
                }
            }
            s_xy[B + hh] = P.xyP[g];
            s_p[B + hh] = hp;
        }
        __syncthreads();

        double fx = 0.0, fy = 0.0;
        {
            double2 rd = ca, ru = pa; // previous ring entry (relative to this node); every fan starts with the break bit
            auto tri = [&](const double2 db, const double2 ub, const double2 dc, const double2 uc) {
                fan_force<double2, double>(db, ub, dc, uc, c0, nu, h, fx, fy);
            };
#pragma unroll
            for (int k = 0; k < kSlotRegs; ++k) ring_word(w[k], s_xy, s_p, ca, pa, rd, ru, tri);
            for (int32_t k = kSlotRegs; k < deg; ++k) ring_word(ell[(int64_t)k * B], s_xy, s_p, ca, pa, rd, ru, tri);
        }
        if (valid) {
            if (CG || P.masked) {
                if (m & 1) fx = 0.0;
                if (m & 2) fy = 0.0;
            }
            if (CG) {
                store2<WT>(P.q, P.N, node, make_double2(fx, fy));
                acc += pa.x * fx + pa.y * fy;
            } else {
                P.y[node] = make_double2(fx, fy);
            }
        }
        t += gridDim.x;
        if (t >= P.t1) break;
        load_tile(t);
    }
    if (CG) {
        // multi-GPU: p of the interface nodes other ranks own is advanced locally from their (exchanged) r
        for (int32_t k = blockIdx.x * B + tid; k < P.n_iface; k += gridDim.x * B) {
            const int32_t g = P.iface[k];
            if (g < P.own0 || g >= P.own1) {
                const double2 r2 = P.r[g], pp = P.pprev[g];
                P.pnew[g] = make_double2(-r2.x + beta * pp.x, -r2.y + beta * pp.y);
            }
        }
        const double tot = block_sum<B>(acc, s_red);
        if (tid == 0) P.partPQ[blockIdx.x] = tot;
    }
}

size_t op_lds_bytes(int32_t cap, int32_t B) { return (size_t)cap * 32 + (size_t)(B / 64) * 8 + 16; }

// plain SpMV launches need no dot partials, so their grid is free: one persistent round (all workgroups co-resident)
static int op_plain_grid(int32_t B, int32_t cap, int32_t tiles)
{
    // the occupancy query is a host API call: do it once per (B, cap), not per launch
    static thread_local int32_t c_B = 0, c_cap = 0;
    static thread_local long c_resident = 0;
    if (c_B == B && c_cap == cap && c_resident > 0) return (int)(c_resident > tiles ? tiles : c_resident);
    int dev = 0, cus = 256, per_cu = 1;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const size_t lds = op_lds_bytes(cap, B);
    hipError_t e;
    if (B == 256)
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_operator_lds<256, false, false>, 256, lds);
    else if (B == 1024)
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_operator_lds<1024, false, false>, 1024, lds);
    else
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_operator_lds<512, false, false>, 512, lds);
    if (e != hipSuccess || per_cu < 1) per_cu = 1;
    long g = (long)per_cu * cus;
    c_B = B;
    c_cap = cap;
    c_resident = g;
    if (g > tiles) g = tiles;
    return g < 1 ? 1 : (int)g;
}

void op_launch(const OpParams &P, int32_t B, bool cg_mode, hipStream_t s)
{
    int grid = cg_grid(P.t1 - P.t0);
    if (P.ell16) {
        if (!cg_mode) grid = op_plain_grid(B, P.cap, P.t1 - P.t0);
        const size_t lds = op_lds_bytes(P.cap, B);
#define MAG_OPL(BB)                                                       \
    if (!cg_mode)                                                         \
        k_operator_lds<BB, false, false><<<grid, BB, lds, s>>>(P);        \
    else if (P.wt)                                                        \
        k_operator_lds<BB, true, true><<<grid, BB, lds, s>>>(P);          \
    else                                                                  \
        k_operator_lds<BB, true, false><<<grid, BB, lds, s>>>(P);
        if (B == 256) {
            MAG_OPL(256)
        } else if (B == 1024) {
            MAG_OPL(1024)
        } else {
            MAG_OPL(512)
        }
#undef MAG_OPL
        return;
    }
#define MAG_OP(BB)                                                     \
    if (cg_mode)                                                       \
        k_operator<BB, true><<<grid, BB, 0, s>>>(P);                   \
    else                                                               \
        k_operator<BB, false><<<grid, BB, 0, s>>>(P);
    if (B == 256) {
        MAG_OP(256)
    } else if (B == 1024) {
        MAG_OP(1024)
    } else {
        MAG_OP(512)
    }
#undef MAG_OP
}

// -------------------------------------------------------- update kernel ---
// alpha = rtr/(p.q); x += alpha p; r += alpha q; partial r.r   (argmin next_iter)
template <int B, bool WT>
__global__ void __launch_bounds__(B) k_update(const UpdParams P)
{
    __shared__ double s_red[B / 64];
    CgState *st = P.st;
    const int done = st->done;
    const long long k = st->iterB;
    const double rrh0 = st->rr_hist[0], rrh1 = st->rr_hist[1];
    // first tile's operands are in flight while the p.q partials are reduced
    int32_t t = P.t0 + blockIdx.x;
    int64_t node = (int64_t)t * B + threadIdx.x;
    double2 q = make_double2(0.0, 0.0), r = q;
    if (node < P.N) {
        q = P.q[node];
        r = P.r[node];
    }
    const double pq = sum_partials<B>(P.partPQ, P.nPart, s_red);
    if (done) return;
    const double alpha = ((k & 1) ? rrh1 : rrh0) / pq;
    double acc = 0.0;
    for (;;) {
        if (node < P.N) {
            r.x += alpha * q.x;
            r.y += alpha * q.y;
            store2<WT>(P.r, P.N, node, r);
            acc += r.x * r.x + r.y * r.y;
        }
        t += gridDim.x;
        if (t >= P.t1) break;
        node = (int64_t)t * B + threadIdx.x;
        if (node < P.N) {
            q = P.q[node];
            r = P.r[node];
        }
    }
    const double tot = block_sum<B>(acc, s_red);
    if (threadIdx.x == 0) {
        P.partRR[blockIdx.x] = tot;
        if (blockIdx.x == 0) {
            st->iterA = k + 1;
            st->alpha_last = alpha; // consumed by the next operator launch for x += alpha p
        }
    }
}

void upd_launch(const UpdParams &P, int32_t B, hipStream_t s)
{
    const int grid = cg_grid(P.t1 - P.t0);
#define MAG_UPD(BB)                                  \
    if (P.wt)                                        \
        k_update<BB, true><<<grid, BB, 0, s>>>(P);   \
    else                                             \
        k_update<BB, false><<<grid, BB, 0, s>>>(P);
    if (B == 256) {
        MAG_UPD(256)
    } else if (B == 1024) {
        MAG_UPD(1024)
    } else {
        MAG_UPD(512)
    }
#undef MAG_UPD
}

// ------------------------------------------------------------ init/setup ---
// argmin init: r0 = -(b - A x0) with x0 = 0 (solver.rs:143) => r0 = -b; p0 = -r0 comes out of the
// first operator launch (beta = 1, p_prev = 0).
template <int B>
__global__ void __launch_bounds__(B) k_cg_init(const double2 *bP, double2 *r, int64_t N, int32_t T, int32_t t0,
                                               int32_t t1, double *partRR)
{
    __shared__ double s_red[B / 64];
    double acc = 0.0;
    // every rank holds the full right-hand side, so ghost residuals start correct without an exchange
    for (int32_t t = blockIdx.x; t < T; t += gridDim.x) {
        const int64_t node = (int64_t)t * B + threadIdx.x;
        if (node < N) {
            const double2 b = bP[node];
            const double2 v = make_double2(-b.x, -b.y);
            r[node] = v;
            if (t >= t0 && t < t1) acc += v.x * v.x + v.y * v.y;
        }
    }
    const double tot = block_sum<B>(acc, s_red);
    if (threadIdx.x == 0) partRR[blockIdx.x] = tot;
}

void cg_init(const double2 *bP, double2 *r, int64_t N, int32_t B, int32_t T, int32_t t0, int32_t t1, double *partRR,
             hipStream_t s)
{
    const int grid = cg_grid(T);
    if (B == 256)
        k_cg_init<256><<<grid, 256, 0, s>>>(bP, r, N, T, t0, t1, partRR);
    else if (B == 1024)
        k_cg_init<1024><<<grid, 1024, 0, s>>>(bP, r, N, T, t0, t1, partRR);
    else
        k_cg_init<512><<<grid, 512, 0, s>>>(bP, r, N, T, t0, t1, partRR);
}

// ---------------------------------------------- multi-GPU pack / unpack ---
// buf[0] = sum of this rank's dot partials; buf[1 + 2k ..] = value at interface node k if this rank owns it, else 0.
// Summed over ranks (one all-reduce) it carries the global dot product and every interface value.
__global__ void __launch_bounds__(256) k_iface_pack(const double *part, int nPart, const double2 *v,
                                                    const int32_t *iface, int32_t n_iface, int32_t own0,
                                                    int32_t own1, double *buf)
{
    __shared__ double s_red[4];
    if (blockIdx.x == 0) {
        const double tot = sum_partials<256>(part, nPart, s_red);
        if (threadIdx.x == 0) buf[0] = tot;
    }
    for (int32_t k = blockIdx.x * 256 + threadIdx.x; k < n_iface; k += gridDim.x * 256) {
        const int32_t g = iface[k];
        const bool mine = g >= own0 && g < own1;
        const double2 val = mine ? v[g] : make_double2(0.0, 0.0);
        buf[1 + 2 * k] = val.x;
        buf[2 + 2 * k] = val.y;
    }
}

void iface_pack(const double *part, int nPart, const double2 *v, const int32_t *iface, int32_t n_iface, int32_t own0,
                int32_t own1, double *buf, hipStream_t s)
{
    int grid = (n_iface + 255) / 256;
    grid = grid < 1 ? 1 : (grid > 256 ? 256 : grid);
    k_iface_pack<<<grid, 256, 0, s>>>(part, nPart, v, iface, n_iface, own0, own1, buf);
}

__global__ void __launch_bounds__(256) k_iface_unpack(const double *buf, const int32_t *iface, int32_t n_iface,
                                                      int32_t own0, int32_t own1, double2 *v)
{
    for (int32_t k = blockIdx.x * 256 + threadIdx.x; k < n_iface; k += gridDim.x * 256) {
        const int32_t g = iface[k];
        if (g < own0 || g >= own1) v[g] = make_double2(buf[1 + 2 * k], buf[2 + 2 * k]);
    }
}

void iface_unpack(const double *buf, const int32_t *iface, int32_t n_iface, int32_t own0, int32_t own1, double2 *v,
                  hipStream_t s)
{
    if (n_iface <= 0) return;
    int grid = (n_iface + 255) / 256;
    grid = grid > 256 ? 256 : grid;
    k_iface_unpack<<<grid, 256, 0, s>>>(buf, iface, n_iface, own0, own1, v);
}

__global__ void __launch_bounds__(256) k_cg_setup(const double *partRR, int nPart, int stop_mode, double tol,
                                                  long long max_iter, CgState *st)
{
    __shared__ double s_red[4];
    const double bb = sum_partials<256>(partRR, nPart, s_red);
    if (threadIdx.x == 0) {
        st->rr_hist[0] = bb;
        st->rr_hist[1] = bb;
        st->bb = bb;
        st->tol = tol;
        st->target = stop_mode == 2 ? tol * sqrt(bb) : tol;
        st->stop_mode = stop_mode;
        st->final_cost = stop_mode == 1 ? bb : sqrt(bb);
        st->alpha_last = 0.0;
        st->iterA = 0;
        st->iterB = 0;
        st->iterations = 0;
        st->max_iter = max_iter;
        st->breakdown = 0;
        st->best_cost = __builtin_inf();
        st->best_iter = 0;
        // b == 0: the first alpha would be 0/0; return x = 0 (documented deviation, see oracle orc_cg)
        st->done = (bb == 0.0) ? 1 : 0;
        st->converged = (bb == 0.0) ? 1 : 0;
        if (bb == 0.0) st->final_cost = 0.0;
    }
}

void cg_setup(const double *partRR, int32_t nPart, int stop_mode, double tol, long long max_iter, CgState *st,
              hipStream_t s)
{
    k_cg_setup<<<1, 256, 0, s>>>(partRR, nPart, stop_mode, tol, max_iter, st);
}

// ============================================ reference-faithful CSR CG ===
// MAG_OP_CSR: what the reference runs per iteration (solver.rs:23-37 + argmin's next_iter), on K_ff in CSR with
// the unknowns in ascending DOF order (solver.rs:443-450): three launches per iteration, same device-side state
// machine as the two-launch matrix-free variant.  Kept for A/B against the matrix-free operator, not for speed.
int csr_grid(int64_t n)
{
    int64_t g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > kMaxGrid ? kMaxGrid : g));
}

__global__ void __launch_bounds__(256) k_csr_p(const CsrCgParams P)
{
    __shared__ double s_red[4];
    CgState *st = P.st;
    const long long k = st->iterA;
    const int was_done = st->done;
    const double alpha = st->alpha_last;
    const double rrh0 = st->rr_hist[0], rrh1 = st->rr_hist[1];
    const double rr = sum_partials<256>(P.partRR, P.nPart, s_red);
    if (was_done) return;
    bool broke = false;
    double beta = 0.0;
    const bool stop = cg_step(st, k, rr, rrh0, rrh1, P.hist, P.hist_len, beta, broke) != 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < P.n; i += (int64_t)gridDim.x * 256) {
        const double pp = P.pprev[i];
        if (!broke) P.x[i] += alpha * pp; // x += alpha_{k-1} p_{k-1}
        if (!stop) P.pnew[i] = -P.r[i] + beta * pp;
    }
}

void csr_p_launch(const CsrCgParams &P, hipStream_t s) { k_csr_p<<<csr_grid(P.n), 256, 0, s>>>(P); }

__global__ void __launch_bounds__(256) k_csr_spmv(const CsrCgParams P)
{
    __shared__ double s_red[4];
    const int done = P.st->done;
    double acc = 0.0;
    if (!done) {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < P.n; i += (int64_t)gridDim.x * 256) {
            double sum = 0.0;
            for (int32_t e = P.rowptr[i]; e < P.rowptr[i + 1]; ++e) sum += P.val[e] * P.pnew[P.col[e]];
            P.q[i] = sum;
            acc += P.pnew[i] * sum;
        }
    }
    const double tot = block_sum<256>(acc, s_red);
    if (threadIdx.x == 0) P.partPQ[blockIdx.x] = tot;
}

void csr_spmv_launch(const CsrCgParams &P, hipStream_t s) { k_csr_spmv<<<csr_grid(P.n), 256, 0, s>>>(P); }

__global__ void __launch_bounds__(256) k_csr_update(int64_t n, double *r, const double *q, const double *partPQ,
                                                    int nPart, double *partRR, CgState *st)
{
    __shared__ double s_red[4];
    const int done = st->done;
    const long long k = st->iterB;
    const double rrh0 = st->rr_hist[0], rrh1 = st->rr_hist[1];
    const double pq = sum_partials<256>(partPQ, nPart, s_red);
    if (done) return;
    const double alpha = ((k & 1) ? rrh1 : rrh0) / pq;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double v = r[i] + alpha * q[i];
        r[i] = v;
        acc += v * v;
    }
    const double tot = block_sum<256>(acc, s_red);
    if (threadIdx.x == 0) {
        partRR[blockIdx.x] = tot;
        if (blockIdx.x == 0) {
            st->iterA = k + 1;
            st->alpha_last = alpha;
        }
    }
}

void csr_update_launch(int64_t n, double *r, const double *q, const double *partPQ, int32_t nPart, double *partRR,
                       CgState *st, hipStream_t s)
{
    k_csr_update<<<csr_grid(n), 256, 0, s>>>(n, r, q, partPQ, nPart, partRR, st);
}

__global__ void __launch_bounds__(256) k_csr_init(const double *b, double *r, int64_t n, double *partRR)
{
    __shared__ double s_red[4];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double v = -b[i];
        r[i] = v;
        acc += v * v;
    }
    const double tot = block_sum<256>(acc, s_red);
    if (threadIdx.x == 0) partRR[blockIdx.x] = tot;
}

void csr_init(const double *b, double *r, int64_t n, double *partRR, hipStream_t s)
{
    k_csr_init<<<csr_grid(n), 256, 0, s>>>(b, r, n, partRR);
}

// solver.rs:443-454: unknown slots filled in ascending DOF order from the CG solution
__global__ void __launch_bounds__(256) k_expand_free(const double *xf, const int32_t *fidx, const uint8_t *u_known,
                                                     const double *u_in, int64_t n2, double *u)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n2) u[i] = u_known[i] ? u_in[i] : xf[fidx[i]];
}

void expand_free(const double *xf, const int32_t *fidx, const uint8_t *u_known, const double *u_in, int64_t n2,
                 double *u, hipStream_t s)
{
    k_expand_free<<<(int)((n2 + 255) / 256), 256, 0, s>>>(xf, fidx, u_known, u_in, n2, u);
}

// ======================================================= fused CG iteration ===
// ONE launch per CG iteration.  Launch j produces iterate j from iterate j-1 and the four dot products of
// iterate j-1 (r.r, p.q, r.q, q.q, reduced from the previous launch's per-workgroup partials):
//     alpha = r.r / p.q                      (argmin: rtr / p.dot(Ap), true dots)
//     r_j = r + alpha q ; x_j = x + alpha p  (argmin: scaled_add)
//     beta = (r.r + 2 alpha r.q + alpha^2 q.q) / r.r      <- |r_j|^2 one step ahead of its own reduction
//     p_j = -r_j + beta p ; q_j = A p_j
// The only departure from argmin's recurrences is the NUMERATOR of beta: |r_j|^2 expanded from exact dots of
// iterate j-1 instead of a second grid-wide reduction.  The expansion restarts from the true r.r every launch, so
// its error does not accumulate (one-step relative error ~ eps * |r_{j-1}|^2 / |r_j|^2); alpha, the stop test and
// the reported cost all use the true r.r of the iterate, available one launch later.  Owner-computes as in
// k_operator_lds; a tile recomputes r_j and p_j of its halo nodes from their previous record.
// State machine shared by the fused iteration kernels: judge iterate j-1 from its exact dots S (identical in every
// workgroup), let workgroup 0 record the verdict, and derive this launch's alpha and beta.  Returns false when the
// solve is over (converged / iteration cap / non-finite residual): x and r of iterate j-1 are then already in place.
// NS = 5 is the preconditioned iteration (k_cg_fused<.., PRE>): S = {r.r, p.q, r.z, q.z, q.Minv q} with z = Minv r;
// rho = r.z takes the place of r.r in alpha and beta, the stop test stays on the true residual norm.
template <int NS>
__device__ inline bool fused_step(FusedState *st, int par, long long j, double target, long long max_iter, int stop_mode,
                                  const double (&S)[NS], double *hist, int hist_len, double &alpha, double &beta)
{
    const double rr = S[0]; // |r_{j-1}|^2, exact
    const double cost = stop_mode == 1 ? fabs(rr) : sqrt(rr);
    const long long it_done = j - 1; // argmin iterations completed when this launch starts
    const bool finished = (it_done >= 1) && (cost <= target);
    const bool broke = !(fabs(rr) <= 1.79769313486231570e308);
    const bool maxed = it_done >= max_iter;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (it_done >= 1 && it_done - 1 < hist_len) hist[it_done - 1] = cost;
        if (it_done >= 1 && cost < st->best_cost) { // argmin's state.update(): best_param follows the lowest cost
            st->best_cost = cost;
            st->best_iter = it_done;
        }
        if (finished || broke || maxed) {
            st->iterations = it_done < 0 ? 0 : it_done;
            st->final_cost = cost;
            st->converged = finished ? 1 : 0;
            st->breakdown = broke ? 1 : 0;
            st->done = 1;
        } else {
            st->jslot[par ^ 1] = j + 1;
        }
    }
    if (finished || broke || maxed) return false;
    const double rho = NS == 5 ? S[2] : rr;
    alpha = rho / S[1];
    beta = fma(alpha * alpha, S[NS - 1], fma(2.0 * alpha, S[NS - 2], rho)) / rho;
    return true;
}

template <int B, int NS>
__device__ inline void block_sumN(double (&v)[NS], double *s_red)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int c = 0; c < NS; ++c) v[c] += __shfl_down(v[c], off);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int c = 0; c < NS; ++c) s_red[c * (B / 64) + (threadIdx.x >> 6)] = v[c];
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NS; ++c) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < B / 64; ++i) t += s_red[c * (B / 64) + i];
        v[c] = t;
    }
}

template <int B>
__device__ inline void block_sum4(double (&v)[4], double *s_red)
{
    block_sumN<B, 4>(v, s_red);
}

// z = Minv r for one node: Minv = (i00, i01, i11) in fp32 (symmetric 2x2; i01 = 0 for plain Jacobi)
__device__ inline double2 apply_minv(const float4 mi, const double2 r)
{
    double2 z;
    z.x = fma((double)mi.y, r.y, (double)mi.x * r.x);
    z.y = fma((double)mi.z, r.y, (double)mi.y * r.x);
    return z;
}

// COMM (multi-GPU): the launch reads the all-reduced exchange buffer of the previous iteration directly (dot partials
// summed over ranks slot by slot, q of interface nodes other ranks own) and writes its own contribution to the next
// one (partials, q of the interface nodes it owns, zeros elsewhere), so a distributed iteration is this kernel plus
// ONE in-place all-reduce -- no pack/unpack launches.
// PRE (opt-in, mag_options.preconditioner): Jacobi / 2x2 block-Jacobi preconditioned CG, p_j = -Minv r_j + beta p;
// every tile applies Minv to its owned AND halo nodes (inverse blocks in fp32, 16 bytes per node), five sums.
template <int B, bool WT, bool COMM, bool PRE>
__global__ void __launch_bounds__(B) k_cg_fused(const FusedParams P)
{
    constexpr int NS = PRE ? 5 : 4;
    extern __shared__ __attribute__((aligned(16))) double2 smem[];
    double2 *s_xy = smem;
    double2 *s_p = smem + P.cap;
    double *s_red = (double *)(smem + 2 * P.cap);
    const int tid = threadIdx.x;

    // ---- everything this launch reads first is issued before anything waits
    FusedState *st = P.st;
    const long long j = st->jslot[P.par];
    const int was_done = st->done;
    const double target = st->target;
    const long long max_iter = st->max_iter;
    const int stop_mode = st->stop_mode;

    int64_t node = 0;
    bool valid = false, hvalid = false;
    double2 ca, ar, aq, ap, xo, hc, hr, hq, hp;
    uint8_t m = 3;
    int32_t deg = 0, nh = 0, hoff = 0, hg = 0, oslot = -1;
    float4 mi = make_float4(0.f, 0.f, 0.f, 0.f), hmi = mi;
    uint32_t w[kSlotRegs];
    const uint32_t *ell = nullptr;
    auto load_tile = [&](int32_t t) {
        const TileMeta tm = P.meta[t];
        node = (int64_t)t * B + tid;
        valid = node < P.N;
        ca = ar = aq = ap = xo = make_double2(0.0, 0.0);
        m = 3;
        oslot = -1;
        if (valid) {
            const Rqp rec = P.in[node];
            ar = rec.r;
            aq = rec.q;
            ap = rec.p;
            xo = P.x[node];
            ca = P.xyP[node];
            m = P.maskP[node];
            if (COMM) oslot = P.own_qslot[node];
            if (PRE) mi = P.minvP[node];
        }
        deg = tm.deg;
        ell = P.ell16 + tm.ell_off + tid;
#pragma unroll
        for (int k = 0; k < kSlotRegs; ++k) w[k] = k < deg ? ell[(int64_t)k * B] : 0xffffffffu;
        hoff = tm.hoff;
        nh = tm.nh;
        hvalid = tid < nh;
        hc = hr = hq = hp = make_double2(0.0, 0.0);
        if (hvalid) {
            hg = P.halo_g[hoff + tid];
            hc = P.halo_xy[hoff + tid];
            if (PRE) hmi = P.halo_minv[hoff + tid];
            const Rqp rec = P.in[hg];
            hr = rec.r;
            hq = rec.q;
            hp = rec.p;
            if (COMM) {
                const int32_t hs = P.halo_qslot[hoff + tid];
                if (hs >= 0) hq = P.comm_in_q[hs];
            }
        }
    };
    load_tile(P.t0 + blockIdx.x);

    // ---- dots of iterate j-1
    double S[NS] = {};
    for (int i = tid; i < P.nPart; i += B) {
#pragma unroll
        for (int c = 0; c < NS; ++c) S[c] += P.part_in[c * P.part_stride_in + i];
    }
    block_sumN<B, NS>(S, s_red);
    if (was_done) return;
    double alpha = 0.0, beta = 0.0;
    if (!fused_step<NS>(st, P.par, j, target, max_iter, stop_mode, S, P.hist, P.hist_len, alpha, beta)) return;

    const double c0 = P.c0, nu = P.nu, h = P.h;
    double acc[NS] = {};
    int32_t t = P.t0 + blockIdx.x;
    for (;;) {
        // r_j, x_j, p_j of the owned node
        double2 rn, pn;
        rn.x = fma(alpha, aq.x, ar.x);
        rn.y = fma(alpha, aq.y, ar.y);
        const double2 zn = PRE ? apply_minv(mi, rn) : rn;
        pn.x = fma(beta, ap.x, -zn.x);
        pn.y = fma(beta, ap.y, -zn.y);
        xo.x = fma(alpha, ap.x, xo.x);
        xo.y = fma(alpha, ap.y, xo.y);
        __syncthreads(); // previous tile's readers are done with the LDS images
        s_xy[tid] = ca;
        s_p[tid] = pn;
        if (hvalid) {
            double2 hrn, hpn;
            hrn.x = fma(alpha, hq.x, hr.x);
            hrn.y = fma(alpha, hq.y, hr.y);
            const double2 hzn = PRE ? apply_minv(hmi, hrn) : hrn;
            hpn.x = fma(beta, hp.x, -hzn.x);
            hpn.y = fma(beta, hp.y, -hzn.y);
            s_xy[B + tid] = hc;
            s_p[B + tid] = hpn;
        }
        for (int32_t hh = tid + B; hh < nh; hh += B) { // more halo nodes than threads (rare)
            const int32_t g = P.halo_g[hoff + hh];
            const Rqp rec = P.in[g];
            double2 hq2 = rec.q;
            if (COMM) {
                const int32_t hs = P.halo_qslot[hoff + hh];
                if (hs >= 0) hq2 = P.comm_in_q[hs];
            }
            double2 hrn, hpn;
            hrn.x = fma(alpha, hq2.x, rec.r.x);
            hrn.y = fma(alpha, hq2.y, rec.r.y);
            const double2 hzn = PRE ? apply_minv(P.halo_minv[hoff + hh], hrn) : hrn;
            hpn.x = fma(beta, rec.p.x, -hzn.x);
            hpn.y = fma(beta, rec.p.y, -hzn.y);
            s_xy[B + hh] = P.halo_xy[hoff + hh];
            s_p[B + hh] = hpn;
        }
        __syncthreads();

        double fx = 0.0, fy = 0.0;
        {
            double2 rd = ca, ru = pn; // previous ring entry (relative to this node); every fan starts with the break bit
            auto tri = [&](const double2 db, const double2 ub, const double2 dc, const double2 uc) {
                fan_force<double2, double>(db, ub, dc, uc, c0, nu, h, fx, fy);
            };
#pragma unroll
            for (int k = 0; k < kSlotRegs; ++k) ring_word(w[k], s_xy, s_p, ca, pn, rd, ru, tri);
            for (int32_t k = kSlotRegs; k < deg; ++k) ring_word(ell[(int64_t)k * B], s_xy, s_p, ca, pn, rd, ru, tri);
        }
        if (valid) {
            if (m & 1) fx = 0.0;
            if (m & 2) fy = 0.0;
            // 48-byte record, three 16-byte stores
            store2<WT>((double2 *)P.out, 3 * P.N, 3 * node, rn);
            store2<WT>((double2 *)P.out, 3 * P.N, 3 * node + 1, make_double2(fx, fy));
            store2<WT>((double2 *)P.out, 3 * P.N, 3 * node + 2, pn);
            store2<WT>(P.x, P.N, node, xo);
            if (COMM && oslot >= 0) P.comm_out_q[oslot] = make_double2(fx, fy);
            acc[0] = fma(rn.y, rn.y, fma(rn.x, rn.x, acc[0]));
            acc[1] = fma(pn.y, fy, fma(pn.x, fx, acc[1]));
            if (PRE) {
                const double2 zq = apply_minv(mi, make_double2(fx, fy));
                acc[2] = fma(rn.y, zn.y, fma(rn.x, zn.x, acc[2]));
                acc[NS - 2] = fma(fy, zn.y, fma(fx, zn.x, acc[NS - 2]));
                acc[NS - 1] = fma(fy, zq.y, fma(fx, zq.x, acc[NS - 1]));
            } else {
                acc[2] = fma(rn.y, fy, fma(rn.x, fx, acc[2]));
                acc[3] = fma(fy, fy, fma(fx, fx, acc[3]));
            }
        }
        t += gridDim.x;
        if (t >= P.t1) break;
        load_tile(t);
    }
    if (COMM) {
        // records of interface nodes other ranks own are advanced locally: q_{j-1} from the exchange buffer, r and p by
        // the same recurrences their owner runs; their slot of the outgoing buffer is this rank's zero
        for (int32_t k = blockIdx.x * B + tid; k < P.n_iface; k += gridDim.x * B) {
            const int32_t g = P.iface[k];
            if (g < P.own0 || g >= P.own1) {
                const Rqp rec = P.in[g];
                const double2 qg = P.comm_in_q[k];
                double2 rn, pn;
                rn.x = fma(alpha, qg.x, rec.r.x);
                rn.y = fma(alpha, qg.y, rec.r.y);
                const double2 zg = PRE ? apply_minv(P.minvP[g], rn) : rn;
                pn.x = fma(beta, rec.p.x, -zg.x);
                pn.y = fma(beta, rec.p.y, -zg.y);
                P.out[g].r = rn;
                P.out[g].p = pn;
                P.comm_out_q[k] = make_double2(0.0, 0.0);
            }
        }
        // partial slots other ranks fill but this (smaller) grid does not
        if (blockIdx.x == 0) {
            for (int32_t i = gridDim.x + tid; i < P.part_stride; i += B) {
#pragma unroll
                for (int c = 0; c < NS; ++c) P.part_out[c * P.part_stride + i] = 0.0;
            }
        }
    }
    block_sumN<B, NS>(acc, s_red);
    if (tid == 0) {
#pragma unroll
        for (int c = 0; c < NS; ++c) P.part_out[c * P.part_stride + blockIdx.x] = acc[c];
    }
}

template <int B, bool WT, bool COMM, bool PRE>
__global__ void __launch_bounds__(B, 4) k_cg_fused_dma(const FusedParams P)
{
    constexpr int NS = PRE ? 5 : 4;
    extern __shared__ __attribute__((aligned(16))) double2 smem[];
    double2 *s_xy = smem;
    double2 *s_p = smem + P.cap;
    double *s_red = (double *)(smem + 2 * P.cap);
    // LDS-DMA staging (global_load_lds_dwordx4: HBM -> LDS without passing through VGPRs, so nothing of the tile is
    // held in registers while the dots are reduced).  A wave moves its 64 records (3 KiB) as three fully coalesced
    // 1-KiB pieces into a wave-private stage and each lane picks its 48-byte record out of LDS (stride 3 x 16 B:
    // conflict-free); a per-lane 48-byte-stride global access streams ~20 % slower.  Halo records are gathered by
    // DMA too (per-lane source address, lane-linear destination).
    double2 *s_stage_all = (double2 *)(s_red + (PRE ? 6 : 4) * (B / 64)); // 6, not 5: keeps the 16-byte alignment
    double2 *s_stage = s_stage_all + (threadIdx.x >> 6) * 192;
    double2 *s_hr = s_stage_all + 3 * B, *s_hq = s_hr + B, *s_hp = s_hq + B;
    typedef __attribute__((address_space(3))) void lds_void;
    typedef __attribute__((address_space(1))) const void glb_void;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const double2 *in2 = (const double2 *)P.in;
    double2 *out2 = (double2 *)P.out;
    const int64_t n3 = 3 * P.N;

    // ---- everything this launch reads first is issued before anything waits
    FusedState *st = P.st;
    const long long j = st->jslot[P.par];
    const int was_done = st->done;
    const double target = st->target;
    const long long max_iter = st->max_iter;
    const int stop_mode = st->stop_mode;

    int64_t node = 0, wbase = 0;
    bool valid = false, hvalid = false;
    double2 xo;
    uint8_t m = 3;
    int32_t deg = 0, nh = 0, hoff = 0, oslot = -1;
    float4 mi = make_float4(0.f, 0.f, 0.f, 0.f), hmi = mi;
    uint32_t w[kSlotRegs];
    const uint32_t *ell = nullptr;
    // callers guarantee that no wave still reads s_xy / the stages of the previous tile (barrier before the call)
    auto load_tile = [&](int32_t t) {
        const TileMeta tm = P.meta[t];
        node = (int64_t)t * B + tid;
        wbase = 3 * ((int64_t)t * B + (tid & ~63));
        valid = node < P.N;
        hoff = tm.hoff;
        nh = tm.nh;
        hvalid = tid < nh;
        int32_t hg = 0;
        if (hvalid) hg = P.halo_g[hoff + tid];
#pragma unroll
        for (int c = 0; c < 3; ++c)
            if (wbase + 64 * c + lane < n3)
                __builtin_amdgcn_global_load_lds((glb_void *)(in2 + wbase + 64 * c + lane), (lds_void *)(s_stage + 64 * c),
                                                 16, 0, 0);
        if (valid)
            __builtin_amdgcn_global_load_lds((glb_void *)(P.xyP + node), (lds_void *)(s_xy + wv * 64), 16, 0, 0);
        xo = make_double2(0.0, 0.0);
        m = 3;
        oslot = -1;
        if (valid) {
            xo = P.x[node];
            m = P.maskP[node];
            if (COMM) oslot = P.own_qslot[node];
            if (PRE) mi = P.minvP[node];
        }
        deg = tm.deg;
        ell = P.ell16 + tm.ell_off + tid;
#pragma unroll
        for (int k = 0; k < kSlotRegs; ++k) w[k] = k < deg ? ell[(int64_t)k * B] : 0xffffffffu;
        if (hvalid) {
            if (PRE) hmi = P.halo_minv[hoff + tid];
            __builtin_amdgcn_global_load_lds((glb_void *)(P.halo_xy + hoff + tid), (lds_void *)(s_xy + B + wv * 64), 16, 0, 0);
            const double2 *src = (const double2 *)(P.in + hg);
            const double2 *qsrc = src + 1;
            if (COMM) {
                const int32_t hs = P.halo_qslot[hoff + tid];
                if (hs >= 0) qsrc = P.comm_in_q + hs;
            }
            __builtin_amdgcn_global_load_lds((glb_void *)(src), (lds_void *)(s_hr + wv * 64), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void *)(qsrc), (lds_void *)(s_hq + wv * 64), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_void *)(src + 2), (lds_void *)(s_hp + wv * 64), 16, 0, 0);
        }
    };
    load_tile(P.t0 + blockIdx.x);

    // ---- dots of iterate j-1
    double S[NS] = {};
    for (int i = tid; i < P.nPart; i += B) {
#pragma unroll
        for (int c = 0; c < NS; ++c) S[c] += P.part_in[c * P.part_stride_in + i];
    }
    block_sumN<B, NS>(S, s_red);
    if (was_done) return;
    double alpha = 0.0, beta = 0.0;
    if (!fused_step<NS>(st, P.par, j, target, max_iter, stop_mode, S, P.hist, P.hist_len, alpha, beta)) return;

    const double c0 = P.c0, nu = P.nu, h = P.h;
    double acc[NS] = {};
    int32_t t = P.t0 + blockIdx.x;
    for (;;) {
        // this wave's DMA has landed: pick the lane's record (r, q, p of iterate j-1) and coordinates out of LDS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const double2 ar = s_stage[3 * lane], aq = s_stage[3 * lane + 1], ap = s_stage[3 * lane + 2];
        const double2 ca = s_xy[tid];
        // r_j, x_j, p_j of the owned node
        double2 rn, pn;
        rn.x = fma(alpha, aq.x, ar.x);
        rn.y = fma(alpha, aq.y, ar.y);
        const double2 zn = PRE ? apply_minv(mi, rn) : rn;
        pn.x = fma(beta, ap.x, -zn.x);
        pn.y = fma(beta, ap.y, -zn.y);
        xo.x = fma(alpha, ap.x, xo.x);
        xo.y = fma(alpha, ap.y, xo.y);
        s_p[tid] = pn;
        if (hvalid) {
            const double2 hr = s_hr[tid], hq = s_hq[tid], hp = s_hp[tid];
            double2 hrn, hpn;
            hrn.x = fma(alpha, hq.x, hr.x);
            hrn.y = fma(alpha, hq.y, hr.y);
            const double2 hzn = PRE ? apply_minv(hmi, hrn) : hrn;
            hpn.x = fma(beta, hp.x, -hzn.x);
            hpn.y = fma(beta, hp.y, -hzn.y);
            s_p[B + tid] = hpn;
        }
        for (int32_t hh = tid + B; hh < nh; hh += B) { // more halo nodes than threads (rare)
            const int32_t g = P.halo_g[hoff + hh];
            const Rqp rec = P.in[g];
            double2 hq2 = rec.q;
            if (COMM) {
                const int32_t hs = P.halo_qslot[hoff + hh];
                if (hs >= 0) hq2 = P.comm_in_q[hs];
            }
            double2 hrn, hpn;
            hrn.x = fma(alpha, hq2.x, rec.r.x);
            hrn.y = fma(alpha, hq2.y, rec.r.y);
            const double2 hzn = PRE ? apply_minv(P.halo_minv[hoff + hh], hrn) : hrn;
            hpn.x = fma(beta, rec.p.x, -hzn.x);
            hpn.y = fma(beta, rec.p.y, -hzn.y);
            s_xy[B + hh] = P.halo_xy[hoff + hh];
            s_p[B + hh] = hpn;
        }
        __syncthreads();

        double fx = 0.0, fy = 0.0;
        {
            double2 rd = ca, ru = pn; // previous ring entry (relative to this node); every fan starts with the break bit
            auto tri = [&](const double2 db, const double2 ub, const double2 dc, const double2 uc) {
                fan_force<double2, double>(db, ub, dc, uc, c0, nu, h, fx, fy);
            };
#pragma unroll
            for (int k = 0; k < kSlotRegs; ++k) ring_word(w[k], s_xy, s_p, ca, pn, rd, ru, tri);
            for (int32_t k = kSlotRegs; k < deg; ++k) ring_word(ell[(int64_t)k * B], s_xy, s_p, ca, pn, rd, ru, tri);
        }
        if (m & 1) fx = 0.0;
        if (m & 2) fy = 0.0;
        // record of iterate j -> wave stage -> three coalesced 1-KiB stores
        s_stage[3 * lane] = rn;
        s_stage[3 * lane + 1] = make_double2(fx, fy);
        s_stage[3 * lane + 2] = pn;
        __builtin_amdgcn_wave_barrier();
        {
            const double2 o0 = s_stage[lane], o1 = s_stage[64 + lane], o2 = s_stage[128 + lane];
            if (wbase + lane < n3) store2<WT>(out2, n3, wbase + lane, o0);
            if (wbase + 64 + lane < n3) store2<WT>(out2, n3, wbase + 64 + lane, o1);
            if (wbase + 128 + lane < n3) store2<WT>(out2, n3, wbase + 128 + lane, o2);
        }
        if (valid) {
            store2<WT>(P.x, P.N, node, xo);
            if (COMM && oslot >= 0) P.comm_out_q[oslot] = make_double2(fx, fy);
            acc[0] = fma(rn.y, rn.y, fma(rn.x, rn.x, acc[0]));
            acc[1] = fma(pn.y, fy, fma(pn.x, fx, acc[1]));
            if (PRE) {
                const double2 zq = apply_minv(mi, make_double2(fx, fy));
                acc[2] = fma(rn.y, zn.y, fma(rn.x, zn.x, acc[2]));
                acc[NS - 2] = fma(fy, zn.y, fma(fx, zn.x, acc[NS - 2]));
                acc[NS - 1] = fma(fy, zq.y, fma(fx, zq.x, acc[NS - 1]));
            } else {
                acc[2] = fma(rn.y, fy, fma(rn.x, fx, acc[2]));
                acc[3] = fma(fy, fy, fma(fx, fx, acc[3]));
            }
        }
        t += gridDim.x;
        if (t >= P.t1) break;
        __syncthreads(); // every wave is done with this tile's LDS images before the next tile's DMA overwrites them
        load_tile(t);
    }
    if (COMM) {
        // records of interface nodes other ranks own are advanced locally: q_{j-1} from the exchange buffer, r and p by
        // the same recurrences their owner runs; their slot of the outgoing buffer is this rank's zero
        for (int32_t k = blockIdx.x * B + tid; k < P.n_iface; k += gridDim.x * B) {
            const int32_t g = P.iface[k];
            if (g < P.own0 || g >= P.own1) {
                const Rqp rec = P.in[g];
                const double2 qg = P.comm_in_q[k];
                double2 rn, pn;
                rn.x = fma(alpha, qg.x, rec.r.x);
                rn.y = fma(alpha, qg.y, rec.r.y);
                const double2 zg = PRE ? apply_minv(P.minvP[g], rn) : rn;
                pn.x = fma(beta, rec.p.x, -zg.x);
                pn.y = fma(beta, rec.p.y, -zg.y);
                P.out[g].r = rn;
                P.out[g].p = pn;
                P.comm_out_q[k] = make_double2(0.0, 0.0);
            }
        }
        // partial slots other ranks fill but this (smaller) grid does not
        if (blockIdx.x == 0) {
            for (int32_t i = gridDim.x + tid; i < P.part_stride; i += B) {
#pragma unroll
                for (int c = 0; c < NS; ++c) P.part_out[c * P.part_stride + i] = 0.0;
            }
        }
    }
    block_sumN<B, NS>(acc, s_red);
    if (tid == 0) {
#pragma unroll
        for (int c = 0; c < NS; ++c) P.part_out[c * P.part_stride + blockIdx.x] = acc[c];
    }
}

static size_t fused_lds_bytes(int32_t cap, int32_t B, bool dma, bool pre)
{
    return (size_t)cap * 32 + (size_t)(B / 64) * (pre ? 48 : 32) + 16 + (dma ? (size_t)B * 96 : 0);
}

static size_t fused_lds_bytes(int32_t cap, int32_t B, bool dma, bool pre);
static bool fused_dma(int32_t B, int32_t cap, bool pre)
{
    const char *e = getenv("MAG_TUNE_DMA"); // read per call: tests switch kernel families inside one process
    if (e && atoi(e) == 0) return false;
    // one 1024-node tile per CU: the stages need 96 KiB on top of the node images (measured at 1M triangles: 23.1 us
    // per iteration against 32.3 us with per-lane records; 512-node tiles remain the faster choice at 19.1 us)
    if (B == 1024) return fused_lds_bytes(cap, B, true, pre) <= 160 * 1024;
    return true;
}

// one table for the occupancy query and the launch: instantiation by (kernel family, B, write-through, COMM, PRE)
typedef void (*fused_fn)(const FusedParams);
template <int B, bool WT>
static fused_fn fused_pick_dma(bool comm, bool pre)
{
    if (comm) return pre ? k_cg_fused_dma<B, WT, true, true> : k_cg_fused_dma<B, WT, true, false>;
    return pre ? k_cg_fused_dma<B, WT, false, true> : k_cg_fused_dma<B, WT, false, false>;
}
template <int B>
static fused_fn fused_pick_aos(bool comm, bool pre)
{
    // per-lane 48-byte-stride records: 16-byte write-through pieces measured slower than plain stores here
    if (comm) return pre ? k_cg_fused<B, false, true, true> : k_cg_fused<B, false, true, false>;
    return pre ? k_cg_fused<B, false, false, true> : k_cg_fused<B, false, false, false>;
}
static fused_fn fused_pick(int32_t B, bool dma, bool wt, bool comm, bool pre)
{
    if (dma) {
        if (B == 256) return wt ? fused_pick_dma<256, true>(comm, pre) : fused_pick_dma<256, false>(comm, pre);
        if (B == 1024) return wt ? fused_pick_dma<1024, true>(comm, pre) : fused_pick_dma<1024, false>(comm, pre);
        return wt ? fused_pick_dma<512, true>(comm, pre) : fused_pick_dma<512, false>(comm, pre);
    }
    if (B == 256) return fused_pick_aos<256>(comm, pre);
    if (B == 1024) return fused_pick_aos<1024>(comm, pre);
    return fused_pick_aos<512>(comm, pre);
}

int fused_grid(int32_t B, int32_t cap, int32_t tiles, bool comm, bool pre)
{
    int dev = 0, cus = 256, per_cu = 1;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const bool dma = fused_dma(B, cap, pre);
    const size_t lds = fused_lds_bytes(cap, B, dma, pre);
    const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fused_pick(B, dma, false, comm, pre),
                                                                      B == 256 || B == 1024 ? B : 512, lds);
    if (e != hipSuccess || per_cu < 1) per_cu = 1;
    long g = (long)per_cu * cus;
    if (g > kMaxGrid) g = kMaxGrid;
    if (const char *cap_env = getenv("MAG_TUNE_GRID")) // tests: few workgroups, so each walks several tiles
        if (atoi(cap_env) > 0 && g > atoi(cap_env)) g = atoi(cap_env);
    if (g > tiles) g = tiles;
    return g < 1 ? 1 : (int)g;
}

void fused_launch(const FusedParams &P, int32_t B, int32_t grid, hipStream_t s)
{
    const bool comm = P.comm_out_q != nullptr, pre = P.minvP != nullptr;
    const bool dma = fused_dma(B, P.cap, pre);
    const int32_t threads = B == 256 || B == 1024 ? B : 512;
    const size_t lds = fused_lds_bytes(P.cap, B, dma, pre);
    fused_pick(B, dma, P.wt != 0, comm, pre)<<<grid, threads, lds, s>>>(P);
}

__global__ void __launch_bounds__(256) k_tile_meta(const int32_t *tile_deg, const int32_t *tile_ent,
                                                   const int64_t *tile_off, const int32_t *tile_hoff, int32_t T,
                                                   TileMeta *meta)
{
    const int32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    TileMeta m;
    m.ell_off = tile_off[t];
    m.deg = tile_deg[t];
    m.hoff = tile_hoff[t];
    m.nh = tile_hoff[t + 1] - tile_hoff[t];
    m.ent = tile_ent[t];
    m.pad2 = 0;
    meta[t] = m;
}

void tile_meta(const int32_t *tile_deg, const int32_t *tile_ent, const int64_t *tile_off, const int32_t *tile_hoff,
               int32_t T, TileMeta *meta, hipStream_t s)
{
    k_tile_meta<<<(T + 255) / 256, 256, 0, s>>>(tile_deg, tile_ent, tile_off, tile_hoff, T, meta);
}

template <int B>
__global__ void __launch_bounds__(B) k_fused_init(const double2 *bP, const float4 *minvP, Rqp *in, Rqp *out, int64_t N,
                                                  int32_t T, int32_t t0, int32_t t1, double *part, int32_t stride)
{
    __shared__ double s_red[B / 64];
    double acc = 0.0, rho = 0.0;
    const double2 z = make_double2(0.0, 0.0);
    for (int32_t t = blockIdx.x; t < T; t += gridDim.x) {
        const int64_t node = (int64_t)t * B + threadIdx.x;
        if (node < N) {
            const double2 b = bP[node];
            Rqp rec;
            rec.r = make_double2(-b.x, -b.y); // argmin init: r0 = -(b - A x0), x0 = 0
            rec.q = z;
            rec.p = z;
            in[node] = rec;
            out[node] = rec;
            if (t >= t0 && t < t1) {
                acc = fma(b.y, b.y, fma(b.x, b.x, acc));
                if (minvP) {
                    const double2 zb = apply_minv(minvP[node], b);
                    rho = fma(b.y, zb.y, fma(b.x, zb.x, rho));
                }
            }
        }
    }
    const double tot = block_sum<B>(acc, s_red);
    const double tot_rho = minvP ? block_sum<B>(rho, s_red) : 0.0;
    if (threadIdx.x == 0) {
        part[blockIdx.x] = tot;
        part[stride + blockIdx.x] = blockIdx.x == 0 ? 1.0 : 0.0; // "p.q" > 0: alpha finite, multiplies q = 0
        part[2 * stride + blockIdx.x] = tot_rho;                  // preconditioned: rho_0 = r0.Minv r0 (else r.q = 0)
        part[3 * stride + blockIdx.x] = 0.0;
        if (minvP) part[4 * stride + blockIdx.x] = 0.0;
    }
}

void fused_init(const double2 *bP, const float4 *minvP, Rqp *in, Rqp *out, int64_t N, int32_t B, int32_t T, int32_t t0,
                int32_t t1, double *part, int32_t stride, int32_t grid, hipStream_t s)
{
    if (B == 256)
        k_fused_init<256><<<grid, 256, 0, s>>>(bP, minvP, in, out, N, T, t0, t1, part, stride);
    else if (B == 1024)
        k_fused_init<1024><<<grid, 1024, 0, s>>>(bP, minvP, in, out, N, T, t0, t1, part, stride);
    else
        k_fused_init<512><<<grid, 512, 0, s>>>(bP, minvP, in, out, N, T, t0, t1, part, stride);
}

// halo copies of the inverse node blocks, contiguous per tile like halo_xy
__global__ void __launch_bounds__(256) k_halo_minv(const int32_t *halo_g, const float4 *minvP, int64_t halo_total,
                                                   float4 *halo_minv)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < halo_total) halo_minv[i] = minvP[halo_g[i]];
}

void halo_minv(const int32_t *halo_g, const float4 *minvP, int64_t halo_total, float4 *out, hipStream_t s)
{
    if (halo_total > 0)
        k_halo_minv<<<(unsigned)((halo_total + 255) / 256), 256, 0, s>>>(halo_g, minvP, halo_total, out);
}

__global__ void __launch_bounds__(256) k_fused_setup(const double *part, int nPart, int32_t stride, int stop_mode,
                                                     double tol, long long max_iter, FusedState *st)
{
    __shared__ double s_red[4];
    (void)stride;
    const double bb = sum_partials<256>(part, nPart, s_red);
    if (threadIdx.x == 0) {
        st->jslot[0] = 0;
        st->jslot[1] = 0;
        st->bb = bb;
        st->tol = tol;
        st->target = stop_mode == 2 ? tol * sqrt(bb) : tol;
        st->stop_mode = stop_mode;
        st->final_cost = stop_mode == 1 ? bb : sqrt(bb);
        st->iterations = 0;
        st->max_iter = max_iter;
        st->breakdown = 0;
        st->best_cost = __builtin_inf();
        st->best_iter = 0;
        st->exchange_timeout = 0;
        st->done = (bb == 0.0) ? 1 : 0; // b == 0: x = 0 (documented deviation)
        st->converged = (bb == 0.0) ? 1 : 0;
        if (bb == 0.0) st->final_cost = 0.0;
    }
}

void fused_setup(const double *part, int32_t nPart, int32_t stride, int stop_mode, double tol, long long max_iter,
                 FusedState *st, hipStream_t s)
{
    k_fused_setup<<<1, 256, 0, s>>>(part, nPart, stride, stop_mode, tol, max_iter, st);
}

// exchange-buffer slots of the COMM iteration.  own_qslot[node] = k if this rank owns interface node iface[k];
// halo_qslot[i] = k if halo entry i is interface node iface[k] owned by another rank (its q comes from the buffer);
// -1 otherwise.  iface is sorted ascending.
__global__ void __launch_bounds__(256) k_own_qslot(const int32_t *iface, int32_t n_iface, int32_t own0, int32_t own1,
                                                   int32_t *own_qslot)
{
    const int32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_iface) return;
    const int32_t g = iface[k];
    if (g >= own0 && g < own1) own_qslot[g] = k;
}

__global__ void __launch_bounds__(256) k_halo_qslot(const int32_t *halo_g, int64_t halo_total, const int32_t *iface,
                                                    int32_t n_iface, int32_t own0, int32_t own1, int32_t *halo_qslot)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= halo_total) return;
    const int32_t g = halo_g[i];
    int32_t slot = -1;
    if (g < own0 || g >= own1) {
        int32_t lo = 0, hi = n_iface;
        while (lo < hi) {
            const int32_t mid = (lo + hi) >> 1;
            if (iface[mid] < g)
                lo = mid + 1;
            else
                hi = mid;
        }
        if (lo < n_iface && iface[lo] == g) slot = lo;
    }
    halo_qslot[i] = slot;
}

void comm_slots(const int32_t *iface, int32_t n_iface, int32_t own0, int32_t own1, const int32_t *halo_g,
                int64_t halo_total, int64_t N, int32_t *own_qslot, int32_t *halo_qslot, hipStream_t s)
{
    (void)hipMemsetAsync(own_qslot, 0xff, 4 * (size_t)N, s);
    if (n_iface > 0) k_own_qslot<<<(n_iface + 255) / 256, 256, 0, s>>>(iface, n_iface, own0, own1, own_qslot);
    if (halo_total > 0)
        k_halo_qslot<<<(unsigned)((halo_total + 255) / 256), 256, 0, s>>>(halo_g, halo_total, iface, n_iface, own0, own1,
                                                                          halo_qslot);
}

// ================================================================= fp32 leg ===
// BASELINE config 5 asks for an fp64-vs-fp32 CG tolerance sweep.  Same fused iteration as k_cg_fused, with r, q, p, x,
// the LDS images and the element arithmetic in fp32 and the four dot products accumulated in fp64.  Coordinates are
// stored relative to the reading tile's first node: element-size differences of O(1) coordinates would keep only
// ~3 digits in fp32, tile-relative ones keep ~6.  Opt-in (mag_options.precision = 1); it cannot meet the 1e-8
// parity bar and is never the default.
template <int B, bool COMM>
__global__ void __launch_bounds__(B) k_cg_fused32(const Fused32Params P)
{
    extern __shared__ __attribute__((aligned(16))) double2 smem[];
    float2 *s_xy = (float2 *)smem;
    float2 *s_p = s_xy + P.cap;
    double *s_red = (double *)(s_p + P.cap);
    const int tid = threadIdx.x;

    FusedState *st = P.st;
    const long long j = st->jslot[P.par];
    const int was_done = st->done;
    const double target = st->target;
    const long long max_iter = st->max_iter;
    const int stop_mode = st->stop_mode;

    int64_t node = 0;
    bool valid = false, hvalid = false;
    float2 ca, ar, aq, ap, xo, hc, hr, hq, hp;
    uint8_t m = 3;
    int32_t deg = 0, nh = 0, hoff = 0, oslot = -1;
    uint32_t w[kSlotRegs];
    const uint32_t *ell = nullptr;
    const float2 z = make_float2(0.f, 0.f);
    auto iface_q = [&](int32_t slot) { // q of an interface node another rank owns: from the all-reduced exchange buffer
        const double2 v = P.comm_in_q[slot];
        return make_float2((float)v.x, (float)v.y);
    };
    auto load_tile = [&](int32_t t) {
        const TileMeta tm = P.meta[t];
        node = (int64_t)t * B + tid;
        valid = node < P.N;
        ca = ar = aq = ap = xo = z;
        m = 3;
        oslot = -1;
        if (valid) {
            const Rqp32 rec = P.in[node];
            ar = rec.r;
            aq = rec.q;
            ap = rec.p;
            xo = P.x[node];
            ca = P.xyP32[node];
            m = P.maskP[node];
            if (COMM) oslot = P.own_qslot[node];
        }
        deg = tm.deg;
        ell = P.ell16 + tm.ell_off + tid;
#pragma unroll
        for (int k = 0; k < kSlotRegs; ++k) w[k] = k < deg ? ell[(int64_t)k * B] : 0xffffffffu;
        hoff = tm.hoff;
        nh = tm.nh;
        hvalid = tid < nh;
        hc = hr = hq = hp = z;
        if (hvalid) {
            const int32_t hg = P.halo_g[hoff + tid];
            hc = P.halo_xy32[hoff + tid];
            const Rqp32 rec = P.in[hg];
            hr = rec.r;
            hq = rec.q;
            hp = rec.p;
            if (COMM) {
                const int32_t hs = P.halo_qslot[hoff + tid];
                if (hs >= 0) hq = iface_q(hs);
            }
        }
    };
    const int32_t t_first = COMM ? P.t0 : 0, t_end = COMM ? P.t1 : P.T;
    load_tile(t_first + blockIdx.x);

    double S[4] = {0.0, 0.0, 0.0, 0.0};
    const int32_t stride_in = COMM ? P.part_stride_in : P.part_stride;
    for (int i = tid; i < P.nPart; i += B) {
#pragma unroll
        for (int c = 0; c < 4; ++c) S[c] += P.part_in[c * stride_in + i];
    }
    block_sum4<B>(S, s_red);
    if (was_done) return;
    double alpha_d = 0.0, beta_d = 0.0;
    if (!fused_step(st, P.par, j, target, max_iter, stop_mode, S, P.hist, P.hist_len, alpha_d, beta_d)) return;
    const float alpha = (float)alpha_d, beta = (float)beta_d;

    const float c0 = P.c0, nu = P.nu, h = P.h;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    int32_t t = t_first + blockIdx.x;
    for (;;) {
        float2 rn, pn;
        rn.x = fma(alpha, aq.x, ar.x);
        rn.y = fma(alpha, aq.y, ar.y);
        pn.x = fma(beta, ap.x, -rn.x);
        pn.y = fma(beta, ap.y, -rn.y);
        xo.x = fma(alpha, ap.x, xo.x);
        xo.y = fma(alpha, ap.y, xo.y);
        __syncthreads();
        s_xy[tid] = ca;
        s_p[tid] = pn;
        for (int32_t hh = tid; hh < nh; hh += B) {
            float2 c2 = hc, r2 = hr, q2 = hq, p2 = hp;
            if (hh >= B) { // more halo nodes than threads (rare)
                const int32_t g = P.halo_g[hoff + hh];
                const Rqp32 rec = P.in[g];
                c2 = P.halo_xy32[hoff + hh];
                r2 = rec.r;
                q2 = rec.q;
                p2 = rec.p;
                if (COMM) {
                    const int32_t hs = P.halo_qslot[hoff + hh];
                    if (hs >= 0) q2 = iface_q(hs);
                }
            }
            float2 hrn, hpn;
            hrn.x = fma(alpha, q2.x, r2.x);
            hrn.y = fma(alpha, q2.y, r2.y);
            hpn.x = fma(beta, p2.x, -hrn.x);
            hpn.y = fma(beta, p2.y, -hrn.y);
            s_xy[B + hh] = c2;
            s_p[B + hh] = hpn;
        }
        __syncthreads();

        float fx = 0.f, fy = 0.f;
        {
            float2 rd = ca, ru = pn; // previous ring entry (relative to this node); every fan starts with the break bit
            auto tri = [&](const float2 db, const float2 ub, const float2 dc, const float2 uc) {
                fan_force<float2, float>(db, ub, dc, uc, c0, nu, h, fx, fy);
            };
#pragma unroll
            for (int k = 0; k < kSlotRegs; ++k) ring_word(w[k], s_xy, s_p, ca, pn, rd, ru, tri);
            for (int32_t k = kSlotRegs; k < deg; ++k) ring_word(ell[(int64_t)k * B], s_xy, s_p, ca, pn, rd, ru, tri);
        }
        if (valid) {
            if (m & 1) fx = 0.f;
            if (m & 2) fy = 0.f;
            Rqp32 o;
            o.r = rn;
            o.q = make_float2(fx, fy);
            o.p = pn;
            P.out[node] = o;
            P.x[node] = xo;
            if (COMM && oslot >= 0) P.comm_out_q[oslot] = make_double2((double)fx, (double)fy);
            acc[0] = fma((double)rn.y, (double)rn.y, fma((double)rn.x, (double)rn.x, acc[0]));
            acc[1] = fma((double)pn.y, (double)fy, fma((double)pn.x, (double)fx, acc[1]));
            acc[2] = fma((double)rn.y, (double)fy, fma((double)rn.x, (double)fx, acc[2]));
            acc[3] = fma((double)fy, (double)fy, fma((double)fx, (double)fx, acc[3]));
        }
        t += gridDim.x;
        if (t >= t_end) break;
        load_tile(t);
    }
    if (COMM) {
        // as k_cg_fused: records of interface nodes other ranks own are advanced locally, their slot of the outgoing
        // buffer is this rank's zero; partial slots this (smaller) grid does not fill are zeroed
        for (int32_t k = blockIdx.x * B + tid; k < P.n_iface; k += gridDim.x * B) {
            const int32_t g = P.iface[k];
            if (g < P.own0 || g >= P.own1) {
                const Rqp32 rec = P.in[g];
                const float2 qg = iface_q(k);
                float2 rn, pn;
                rn.x = fma(alpha, qg.x, rec.r.x);
                rn.y = fma(alpha, qg.y, rec.r.y);
                pn.x = fma(beta, rec.p.x, -rn.x);
                pn.y = fma(beta, rec.p.y, -rn.y);
                P.out[g].r = rn;
                P.out[g].p = pn;
                P.comm_out_q[k] = make_double2(0.0, 0.0);
            }
        }
        if (blockIdx.x == 0) {
            for (int32_t i = gridDim.x + tid; i < P.part_stride; i += B) {
#pragma unroll
                for (int c = 0; c < 4; ++c) P.part_out[c * P.part_stride + i] = 0.0;
            }
        }
    }
    block_sum4<B>(acc, s_red);
    if (tid == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) P.part_out[c * P.part_stride + blockIdx.x] = acc[c];
    }
}

static size_t fused32_lds_bytes(int32_t cap, int32_t B) { return (size_t)cap * 16 + (size_t)(B / 64) * 32 + 32; }

int fused32_grid(int32_t B, int32_t cap, int32_t tiles)
{
    int dev = 0, cus = 256, per_cu = 1;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const size_t lds = fused32_lds_bytes(cap, B);
    hipError_t e = B == 256 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_cg_fused32<256, true>, 256, lds)
                            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_cg_fused32<512, true>, 512, lds);
    if (e != hipSuccess || per_cu < 1) per_cu = 1;
    long g = (long)per_cu * cus;
    if (g > kMaxGrid) g = kMaxGrid;
    if (g > tiles) g = tiles;
    return g < 1 ? 1 : (int)g;
}

void fused32_launch(const Fused32Params &P, int32_t B, int32_t grid, hipStream_t s)
{
    const size_t lds = fused32_lds_bytes(P.cap, B);
    const bool comm = P.comm_out_q != nullptr;
    if (B == 256) {
        if (comm)
            k_cg_fused32<256, true><<<grid, 256, lds, s>>>(P);
        else
            k_cg_fused32<256, false><<<grid, 256, lds, s>>>(P);
    } else if (comm)
        k_cg_fused32<512, true><<<grid, 512, lds, s>>>(P);
    else
        k_cg_fused32<512, false><<<grid, 512, lds, s>>>(P);
}

__global__ void __launch_bounds__(256) k_coords32(const double2 *xyP, const int32_t *halo_g, const int32_t *tile_hoff,
                                                  int64_t N, int32_t B, int32_t T, float2 *xyP32, float2 *halo_xy32)
{
    const int32_t t = blockIdx.x;
    const double2 o = xyP[(int64_t)t * B]; // the tile's first node is its origin
    for (int l = threadIdx.x; l < B; l += 256) {
        const int64_t i = (int64_t)t * B + l;
        if (i < N) {
            const double2 c = xyP[i];
            xyP32[i] = make_float2((float)(c.x - o.x), (float)(c.y - o.y));
        }
    }
    for (int32_t k = tile_hoff[t] + threadIdx.x; k < tile_hoff[t + 1]; k += 256) {
        const double2 c = xyP[halo_g[k]];
        halo_xy32[k] = make_float2((float)(c.x - o.x), (float)(c.y - o.y));
    }
}

void coords32(const double *xyP, const int32_t *halo_g, const int32_t *tile_hoff, int64_t N, int32_t B, int32_t T,
              float *xyP32, float *halo_xy32, hipStream_t s)
{
    k_coords32<<<T, 256, 0, s>>>((const double2 *)xyP, halo_g, tile_hoff, N, B, T, (float2 *)xyP32, (float2 *)halo_xy32);
}

template <int B>
__global__ void __launch_bounds__(B) k_fused32_init(const double2 *bP, Rqp32 *in, Rqp32 *out, float2 *x, int64_t N,
                                                    int32_t T, int32_t t0, int32_t t1, double *part, int32_t stride)
{
    __shared__ double s_red[B / 64];
    double acc = 0.0;
    const float2 z = make_float2(0.f, 0.f);
    for (int32_t t = blockIdx.x; t < T; t += gridDim.x) {
        const int64_t node = (int64_t)t * B + threadIdx.x;
        if (node < N) {
            const double2 b = bP[node];
            Rqp32 rec;
            rec.r = make_float2((float)-b.x, (float)-b.y);
            rec.q = z;
            rec.p = z;
            in[node] = rec;
            out[node] = rec;
            x[node] = z;
            if (t >= t0 && t < t1) acc = fma((double)rec.r.y, (double)rec.r.y, fma((double)rec.r.x, (double)rec.r.x, acc)); // this rank's tiles
        }
    }
    const double tot = block_sum<B>(acc, s_red);
    if (threadIdx.x == 0) {
        part[blockIdx.x] = tot;
        part[stride + blockIdx.x] = blockIdx.x == 0 ? 1.0 : 0.0;
        part[2 * stride + blockIdx.x] = 0.0;
        part[3 * stride + blockIdx.x] = 0.0;
    }
}

void fused32_init(const double2 *bP, Rqp32 *in, Rqp32 *out, float2 *x, int64_t N, int32_t B, int32_t T, int32_t t0,
                  int32_t t1, double *part, int32_t stride, int32_t grid, hipStream_t s)
{
    if (B == 256)
        k_fused32_init<256><<<grid, 256, 0, s>>>(bP, in, out, x, N, T, t0, t1, part, stride);
    else
        k_fused32_init<512><<<grid, 512, 0, s>>>(bP, in, out, x, N, T, t0, t1, part, stride);
}

__global__ void __launch_bounds__(256) k_x32_to_f64(const float2 *x32, int64_t N, double2 *x)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < N) x[i] = make_double2((double)x32[i].x, (double)x32[i].y);
}

void x32_to_f64(const float2 *x32, int64_t N, double2 *x, hipStream_t s)
{
    k_x32_to_f64<<<(int)((N + 255) / 256), 256, 0, s>>>(x32, N, x);
}

// --------------------------------------------------- numbering helpers ---
static inline int blocks_for(int64_t n, int threads)
{
    int64_t b = (n + threads - 1) / threads;
    return (int)(b < 1 ? 1 : b);
}

__global__ void __launch_bounds__(256) k_to_hilbert(const double *v, const int32_t *iperm, const uint8_t *u_known,
                                                    int32_t masked, int64_t N, double *vP)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= 2 * N) return;
    vP[2 * (int64_t)iperm[r >> 1] + (r & 1)] = (masked && u_known[r]) ? 0.0 : v[r];
}

void to_hilbert(const double *v, const int32_t *iperm, const uint8_t *u_known, int32_t masked, int64_t N, double *vP,
                hipStream_t s)
{
    k_to_hilbert<<<blocks_for(2 * N, 256), 256, 0, s>>>(v, iperm, u_known, masked, N, vP);
}

__global__ void __launch_bounds__(256) k_from_hilbert(const double *yP, const int32_t *iperm, int64_t N, double *y)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= 2 * N) return;
    y[r] = yP[2 * (int64_t)iperm[r >> 1] + (r & 1)];
}

void from_hilbert(const double *yP, const int32_t *iperm, int64_t N, double *y, hipStream_t s)
{
    k_from_hilbert<<<blocks_for(2 * N, 256), 256, 0, s>>>(yP, iperm, N, y);
}

__global__ void __launch_bounds__(256) k_known_to_hilbert(const double *u_in, const uint8_t *u_known,
                                                          const int32_t *iperm, int64_t N, double *uP)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= 2 * N) return;
    uP[2 * (int64_t)iperm[r >> 1] + (r & 1)] = u_known[r] ? u_in[r] : 0.0;
}

void known_to_hilbert(const double *u_in, const uint8_t *u_known, const int32_t *iperm, int64_t N, double *uP,
                      hipStream_t s)
{
    k_known_to_hilbert<<<blocks_for(2 * N, 256), 256, 0, s>>>(u_in, u_known, iperm, N, uP);
}

// b = f_known - K_fk u_known on free rows (solver.rs:427-432), matrix-free: yP = K * u_ext
__global__ void __launch_bounds__(256) k_rhs_from_apply(const double *yP, const double *f_in, const uint8_t *u_known,
                                                        const uint32_t *perm, int64_t N, double *bP)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; // Hilbert DOF index
    if (r >= 2 * N) return;
    const int64_t o = 2 * (int64_t)perm[r >> 1] + (r & 1);
    bP[r] = u_known[o] ? 0.0 : (-yP[r] + f_in[o]);
}

void rhs_from_apply(const double *yP, const double *f_in, const uint8_t *u_known, const uint32_t *perm, int64_t N,
                    double *bP, hipStream_t s)
{
    k_rhs_from_apply<<<blocks_for(2 * N, 256), 256, 0, s>>>(yP, f_in, u_known, perm, N, bP);
}


} // namespace magk
